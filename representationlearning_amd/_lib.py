"""ctypes binding of librssf.so (the C ABI declared in include/rssf.h).

Fails loudly if the shared library is missing: there is deliberately no fallback path.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: it brings in the HIP runtime librssf links against)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSSF_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "librssf.so")

RSSF_F32, RSSF_BF16 = 0, 1
WGRAD_NO_DRAW = 0x200         # rssf.h RSSF_WGRAD_NO_DRAW: the caller of rssf_conv_wgrad_bnapply will not read `draw`
CONV_GENERIC = 0x100
          # rssf.h RSSF_CONV_GENERIC: OR-ed into the dtype argument of rssf_conv_gather* (generic kernels only)

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class WinAttnFwdParams(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "x", "y", "stats_x", "stats_y", "omega", "ln_gamma", "ln_beta",
        "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "out")] + [
        (n, c_int) for n in ("B", "H", "W", "C", "heads", "window", "dtype")]


class WinAttnBwdParams(ctypes.Structure):
    _fields_ = [("f", WinAttnFwdParams)] + [(n, c_void_p) for n in (
        "dout", "dxhat", "dyhat", "domega", "dwq", "dbq", "dwk", "dbk", "dwv", "dbv", "dwo", "dbo", "prod_ws")]


MAX_TAPS = 19      # RSSF_MAX_TAPS
P2P_MAX_ITEMS, P2P_MAX_FLOATS = 8, 4096      # RSSF_P2P_MAX_ITEMS, RSSF_P2P_MAX_FLOATS


class PackJob(ctypes.Structure):       # rssf_pack_job
    _fields_ = [("w", c_void_p * 3), ("out", c_void_p), ("ks", c_int * 3)] + [
        (n, c_int) for n in ("nsrc", "ntaps", "cout", "cin", "rows_p", "cols_p", "transpose")] + [
        ("src_of_tap", c_int * MAX_TAPS), ("kpos_of_tap", c_int * MAX_TAPS), ("alias_of_tap", (c_int * 4) * MAX_TAPS)]


class WgradReduceJob(ctypes.Structure):       # rssf_wgrad_reduce_job
    _fields_ = [("partial", c_void_p), ("dw", c_void_p * 3), ("ks", c_int * 3)] + [(n, c_int) for n in ("ntaps", "cout", "cin", "ksplit")] + [
        ("src_of_tap", c_int * MAX_TAPS), ("kpos_of_tap", c_int * MAX_TAPS), ("alias_of_tap", (c_int * 4) * MAX_TAPS)]


GROUP_MAX = 4      # RSSF_GROUP_MAX
LOSS_ACC_ELEMS = 192      # RSSF_LOSS_ACC_ELEMS: fp32 scratch per sample of rssf_cgfl_loss_fwd (tests/test_abi.py holds it against rssf.h)
c_double = ctypes.c_double


class Conv3x3Item(ctypes.Structure):       # rssf_conv3x3_item
    _fields_ = [(n, c_void_p) for n in ("in_", "wpk", "out", "stats", "addend", "bn_raw", "bn_res", "bn_ss", "bn_sums", "pre_stats", "pre_gamma",
                                        "pre_beta", "pre_running_mean", "pre_running_var", "pre_mean_invstd", "pre_ss")] + [
        ("pre_n", c_double), ("pre_momentum", c_float), ("pre_eps", c_float)] + [
        (n, c_int) for n in ("pre_training", "pre_act", "bn_act", "B", "H", "W", "Cin", "Cout")]


class Wgrad3x3Item(ctypes.Structure):       # rssf_wgrad3x3_item
    _fields_ = [(n, c_void_p) for n in ("dout", "in_", "dw", "workspace", "defer_reduce", "bn_dy", "bn_raw", "bn_ss", "bn_mi", "bn_sums", "bn_res",
                                        "draw", "dres", "dgamma", "dbeta", "in_ss")] + [
        ("bn_n", c_double), ("pscale", c_float)] + [(n, c_int) for n in ("bn_act", "bn_training", "in_act", "B", "H", "W", "Cin", "Cout")]


class BnApplyItem(ctypes.Structure):       # rssf_bn_apply_item
    _fields_ = [(n, c_void_p) for n in ("raw", "stats", "gamma", "beta", "running_mean", "running_var", "mean_invstd", "scale_shift", "res_pre",
                                        "res_post", "y")] + [
        ("rows", c_int64), ("n", c_double), ("momentum", c_float), ("eps", c_float), ("C", c_int), ("act", c_int), ("training", c_int)]


class BnReduceItem(ctypes.Structure):       # rssf_bn_reduce_item
    _fields_ = [(n, c_void_p) for n in ("dy", "raw", "scale_shift", "res_pre", "sums")] + [("rows", c_int64), ("C", c_int), ("act", c_int)]


class BnBwdApplyItem(ctypes.Structure):     # rssf_bn_bwd_apply_item
    _fields_ = ([(n, c_void_p) for n in ("dy", "raw", "scale_shift", "mean_invstd", "sums", "res_pre", "draw", "dres", "dgamma", "dbeta")]
                + [("rows", c_int64), ("n", ctypes.c_double), ("C", c_int), ("act", c_int), ("training", c_int), ("param_grad_scale", c_float)])


# name -> (restype, argtypes); every symbol include/rssf.h declares must appear here (tests check it)
SIGNATURES = {
    "rssf_version": (ctypes.c_char_p, []),
    "rssf_arch": (ctypes.c_char_p, []),
    "rssf_last_error": (ctypes.c_char_p, []),
    "rssf_layernorm_fwd": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_float, c_int, c_void_p]),
    "rssf_layernorm_bwd": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int, c_void_p]),
    "rssf_gate_pool_fwd": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_void_p]),
    "rssf_ln_gate_pool_fwd_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "rssf_gate_pool_ln_bwd_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "rssf_gate_pool_ln_bwd": (c_int, [c_void_p] * 14 + [c_int, c_int, c_int, c_int, c_void_p]),
    "rssf_ln_gate_pool_fwd": (c_int, [c_void_p] * 4 + [c_float] + [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p]),
    "rssf_gate_weights_fwd": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    "rssf_gate_weights_bwd": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_void_p]),
    "rssf_gate_pool_bwd": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p]),
    "rssf_winattn_fwd": (c_int, [ctypes.POINTER(WinAttnFwdParams), c_void_p]),
    "rssf_winattn_bwd_workspace_elems": (c_int64, [c_int] * 4),
    "rssf_winattn_bwd": (c_int, [ctypes.POINTER(WinAttnBwdParams), c_void_p]),
    "rssf_conv_tile_n": (c_int, [c_int]),
    "rssf_conv_packed_elems": (c_int64, [c_int, c_int, c_int, c_int]),
    "rssf_conv_packed_rows": (c_int, [c_int]),
    "rssf_conv_packed_cols": (c_int, [c_int, c_int]),
    "rssf_conv_pack_job_blocks": (c_int, [c_int, c_int, c_int, c_int]),
    "rssf_conv_pack_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rssf_conv_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                               c_void_p, c_int, c_void_p]),
    "rssf_conv_gather": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv_stats_workspace_elems": (c_int64, [c_int] * 4),
    "rssf_conv_gather_add": (c_int, [c_void_p] * 7 + [c_int] * 10 + [c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv_gather_bnbwd": (c_int, [c_void_p] * 7 + [c_int] + [c_void_p] + [c_int] * 10 + [c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv_wgrad_workspace_elems": (c_int64, [c_int] * 6),
    "rssf_conv_wgrad_planes_supported": (c_int, [c_int] * 7 + [c_void_p, c_void_p, c_int, c_int]),
    "rssf_conv_wgrad_planes": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv_wgrad": (c_int, [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv_wgrad_preact": (c_int, [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv_wgrad_preact_dgrad_supported": (c_int, [c_int] * 6),
    "rssf_conv_wgrad_preact_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 6 + [c_int] * 5 + [c_void_p, c_int, c_void_p]),
    "rssf_conv_wgrad_bnapply_dgrad_supported": (c_int, [c_int] * 9 + [c_void_p, c_void_p, c_int, c_int]),
    "rssf_conv_wgrad_bnapply_dgrad": (c_int, [c_void_p] * 8 + [c_int, ctypes.c_double, c_int, c_float] + [c_void_p] * 6 + [c_int] * 5 + [c_void_p, c_int, c_void_p]),
    "rssf_conv_wgrad_bnapply": (c_int, [c_void_p] * 10 + [c_int, ctypes.c_double, c_int, c_float] + [c_void_p, c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv_wgrad_preact_supported": (c_int, [c_int] * 10 + [c_void_p, c_void_p, c_int, c_int]),
    "rssf_conv_gather_preact_supported": (c_int, [c_int] * 10 + [c_void_p, c_void_p, c_int]),
    "rssf_conv_gather_preact": (c_int, [c_void_p] * 8 + [ctypes.c_double, c_float, c_float, c_int, c_int] + [c_void_p] * 5 + [c_int] * 10 + [c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv_wgrad_reduce_blocks": (c_int, [c_void_p]),
    "rssf_conv_wgrad_reduce_batch": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "rssf_conv3x3_group": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "rssf_conv3x3_wgrad_group": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "rssf_bn_finalize_apply_group": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "rssf_bn_bwd_reduce_group": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "rssf_bn_bwd_apply_group": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "rssf_bn_finalize": (c_int, [c_void_p] * 7 + [c_int, ctypes.c_double, c_float, c_float, c_int, c_void_p]),
    "rssf_bn_apply": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "rssf_bn_finalize_apply_planes_supported": (c_int, [c_int] * 6),
    "rssf_bn_finalize_apply_planes": (c_int, [c_void_p] * 10 + [c_int] * 6 + [ctypes.c_double, c_float, c_float, c_int, c_int, c_void_p]),
    "rssf_bn_finalize_apply": (c_int, [c_void_p] * 11 + [c_int64, c_int, c_int, ctypes.c_double, c_float, c_float, c_int, c_int, c_void_p]),
    "rssf_bn_bwd_reduce_workspace_elems": (c_int64, [c_int64, c_int]),
    "rssf_bn_bwd_reduce": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "rssf_bn_bwd_reduce_post": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "rssf_bn_bwd_apply_post": (c_int, [c_void_p] * 12 + [c_int64, c_int, c_int, ctypes.c_double, c_int, c_float, c_int, c_void_p]),
    "rssf_bn_bwd_apply": (c_int, [c_void_p] * 10 + [c_int64, c_int, c_int, ctypes.c_double, c_int, c_float, c_int, c_void_p]),
    "rssf_input_pipeline": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "rssf_upsample_bilinear": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "rssf_upsample_bilinear_slice": (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "rssf_head_upsample_softmax": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "rssf_upsample_nearest_add": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "rssf_upsample_nearest_sum": (c_int, [c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 6 + [c_void_p]),
    "rssf_aux_head_workspace_elems": (c_int64, [c_int, c_int]),
    "rssf_aux_head_fwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "rssf_maxpool3x3s2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rssf_mha_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_float, c_int, c_void_p]),
    "rssf_dwconv3x3": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "rssf_attn_proj_sigmoid": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int64, c_void_p]),
    "rssf_attn_pred": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    "rssf_resize_bilinear": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "rssf_cam_merge": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "rssf_cam_normalize": (c_int, [c_void_p, c_int, c_int64, c_void_p]),
    "rssf_cgfl_loss_fwd": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "rssf_cgfl_loss_bwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "rssf_argmax_confusion": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "rssf_zero_f32": (c_int, [c_void_p, c_int64, c_void_p]),
    "rssf_vec_sum3": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "rssf_vec_add_to3": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "rssf_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rssf_add3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rssf_pad_channels": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "rssf_add_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rssf_image_to_nhwc": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_int64] * 4 + [c_int, c_void_p]),
    "rssf_grad_sqnorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "rssf_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float, c_void_p, c_float, c_float,
                              c_float, c_int, c_void_p]),
    "rssf_comm_unique_id": (c_int, [c_void_p, ctypes.c_char_p]),
    "rssf_comm_init": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p, ctypes.c_char_p]),
    "rssf_comm_rank": (c_int, [c_void_p]),
    "rssf_comm_world": (c_int, [c_void_p]),
    "rssf_comm_nranks": (c_int, [c_void_p]),
    "rssf_allreduce_bucket": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "rssf_syncbn_exchange": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "rssf_comm_destroy": (c_int, [c_void_p]),
    "rssf_p2p_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "rssf_p2p_connect": (c_int, [c_void_p, c_int, c_void_p]),
    "rssf_p2p_exchange": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rssf_p2p_connect_local": (c_int, [c_void_p, c_int, c_void_p]),
    "rssf_p2p_set_timeout_ms": (c_int, [c_void_p, c_int]),
    "rssf_p2p_status": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "rssf_p2p_wait_us": (c_int, [c_void_p, c_int, ctypes.POINTER(c_double), ctypes.POINTER(ctypes.c_int64), c_int]),
    "rssf_p2p_destroy": (c_int, [c_void_p]),
    "rssf_debug_lane_reduce": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rssf_debug_trread": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rssf_debug_poison_lds": (c_int, [ctypes.c_uint, c_void_p, c_void_p]),
    "rssf_debug_mma": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
}

_lib = None


def load():
    """Load librssf.so once; raise (never fall back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"librssf.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  representationlearning_amd has no CPU/eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    arch = lib.rssf_arch()
    if arch != b"gfx950":
        raise RuntimeError("librssf.so holds gfx950 code objects only; the current device reports %r" % arch.decode())
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rssf_status {rc}): {load().rssf_last_error().decode()}")


def dtype_code(t):
    if t.dtype == torch.float32:
        return RSSF_F32
    if t.dtype == torch.bfloat16:
        return RSSF_BF16
    raise TypeError(f"librssf supports float32/bfloat16 activations, got {t.dtype}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    """Raw handle of torch's current HIP stream (the cheap C accessor: this is called ~3 000 times per training step)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("librssf ops run on the MI355X only (got a CPU tensor); there is no CPU fallback")
