/* librssf — C ABI of the MI355X-native RSSFormer training-step hot path (gfx950 only).
 *
 * The reference (Rongtao-Xu/RepresentationLearning, RSSFormer-TIP2023) has no FFI on this path:
 * it is Python nn.Modules over ATen.  This header is the boundary the build adds
 * under that Python surface; each entry point names the reference function it replaces
 * (paths relative to /root/reference/RSSFormer-TIP2023/).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; all tensor pointers are DEVICE pointers owned by the
 *     caller; the library never allocates/frees device memory and keeps no global mutable state;
 *   - asynchronous on the given hipStream_t (passed as void*), no implicit device sync, re-entrant;
 *   - activations are channels-last: tokens [B, N=H*W, C] == NHWC; `dtype` selects activation storage
 *     (RSSF_F32 | RSSF_BF16); parameters, statistics and parameter gradients are always fp32;
 *     parameter gradients are ACCUMULATED (+=) into the caller's fp32 buffers;
 *   - return 0 (RSSF_OK) or a negative rssf_status; rssf_last_error() is a thread-local string.
 */
#ifndef RSSF_H
#define RSSF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { RSSF_F32 = 0, RSSF_BF16 = 1 } rssf_dtype;
/* Kernel selection is a function of the call's arguments only (shape, dtype, which optional operands are present): no environment
 * variable or other process state steers it.  The one explicit knob: OR RSSF_CONV_GENERIC into the `dtype` argument of the
 * rssf_conv_gather* / rssf_conv_wgrad* entry points to run the generic gather / halo / tiled kernels instead of a shape-specialised one
 * (the point-wise stream kernels of the forward pass and of the weight gradient, the 128-channel many-tap kernel, the 32 -> 32 channel
 * 3x3 row-stream kernel of rssf_conv_gather* / rssf_conv_gather_preact / rssf_conv_gather_bnbwd / rssf_conv3x3_group: conv_rows32.hip)
 * - same results up to the summation order (bit-identical for the 3x3 forward without a bias); the parity tests hold the two against
 * each other. */
#define RSSF_CONV_GENERIC 0x100
/* OR-ed into the `dtype` argument of rssf_conv_wgrad_bnapply: the caller will not read `draw` after the call (the layer needs no data
 * gradient - the stem's first convolution): a kernel that forms draw on the fly may skip writing it.  `draw` must still be a valid buffer. */
#define RSSF_WGRAD_NO_DRAW 0x200
/* per-channel BatchNorm statistics are accumulated into this many interleaved copies (slot = block index mod slots) to
 * spread same-address atomics; every statistics buffer below is [RSSF_BN_SLOTS][2][C] fp32 and consumers sum the slots */
#define RSSF_BN_SLOTS 16
#define RSSF_BN_BWD_SLOTS 8
typedef enum {
  RSSF_OK = 0,
  RSSF_ERR_BAD_ARG = -1,
  RSSF_ERR_UNSUPPORTED = -2,
  RSSF_ERR_LAUNCH = -3
} rssf_status;

const char* rssf_version(void);
const char* rssf_arch(void);        /* gcnArchName of the current device ("gfx950" is the only one the kernels exist for);
                                     * the compiled-for architecture when no device is present */
const char* rssf_last_error(void);  /* thread-local */

/* ---- LayerNorm over C (eps 1e-6): modules/MTFM.py:64,80-81,107,109 ------------------------------ */
/* stats[row] = {mean, rstd}; y may be NULL (stats only).  rows = B*N. */
int rssf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                       int64_t rows, int C, float eps, int dtype, void* stream);
/* dx = LN'(dy); dgamma/dbeta accumulated (+=).  dx_add (optional) is added to dx (residual gradient). */
int rssf_layernorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma, const void* dx_add,
                       void* dx, float* dgamma, float* dbeta, int64_t rows, int C, int dtype, void* stream);

/* ---- Saliency gate: modules/multihead_isa_pool_attention.py:101-115 (SpatialAttention), :148-167 -- */
/* pool: for each of X (s=0) and Y (s=1): mean / max over the C "view channels" of the (B,N,C)->(B,C,H,W)
 * reinterpretation of LN1(x).  pooled [B][4][N] = {aX, mX, aY, mY}; argmax [B][2][N] (int32 view-channel). */
int rssf_gate_pool_fwd(const void* x, const void* y, const float* stats_x, const float* stats_y,
                       const float* gamma, const float* beta, float* pooled, int32_t* argmax,
                       int B, int N, int C, int dtype, void* stream);
/* the same launch FORMING the LayerNorm statistics of x and y instead of reading them (norm1 of both streams, MTFM.py:104-105: the two
 * statistics-only rssf_layernorm_fwd passes are not needed): stats_x / stats_y [B*N][2] = {mean, rstd} are OUTPUTS, rssf_layernorm_fwd(eps)'s
 * arithmetic (equal to the last bit or one ulp).  Shapes: N % C == 0, C a multiple of the 16-byte vector, a power of two of <= 16 vectors per row
 * (rssf_ln_gate_pool_fwd_supported == 1; Base C = 32 in both dtypes). */
int rssf_ln_gate_pool_fwd_supported(int B, int N, int C, int dtype);
int rssf_ln_gate_pool_fwd(const void* x, const void* y, const float* gamma, const float* beta, float eps, float* stats_x, float* stats_y,
                          float* pooled, int32_t* argmax, int B, int N, int C, int dtype, void* stream);
/* weights: g_s = sigmoid(conv7x7(pooled_s; k_s)); omega = softmax_2(Wl [g0;g1] + bl).
 * k [2][2][7][7], wl [2][2], bl [2]; gsig [B][2][N] (saved for bwd), omega [B][2][N], logits [B][2][N] (optional). */
int rssf_gate_weights_fwd(const float* pooled, const float* k, const float* wl, const float* bl, float* gsig,
                          float* omega, float* logits, int B, int H, int W, void* stream);
/* backward of gate_weights: domega [B][2][N] -> dpooled; dk, dwl, dbl accumulated (+=).
 * dk [2][2][7][7]; with `dk_stream1` non-null the gradient of k[1] (98 values) is accumulated THERE and dk takes k[0]'s 98
 * values only - the two 7x7 kernels are separate parameters of the reference module (multihead_isa_pool_attention.py:30-31),
 * so a caller can hand in the two .grad buffers and needs no staging copy.
 * `dpooled` must hold B*6*N + RSSF_GATE_SLOTS*202 floats: the first B*4*N are the result ([B][4][N]), the rest is
 * scratch ([B][2][N] pre-sigmoid gradients, then slotted partial sums of the 202 parameter gradients). */
#define RSSF_GATE_SLOTS 16
int rssf_gate_weights_bwd(const float* domega, const float* pooled, const float* gsig, const float* omega,
                          const float* k, const float* wl, float* dpooled, float* dk, float* dk_stream1, float* dwl,
                          float* dbl, int B, int H, int W, void* stream);
/* backward of gate_pool + merge: dxhat[b,n,c] (+)= gate-path gradient (mean: /C, max: argmax-routed).
 * dxhat/dyhat hold the attention-path gradient w.r.t. LN1 outputs on entry and the total on exit. */
int rssf_gate_pool_bwd(const float* dpooled, const int32_t* argmax, void* dxhat, void* dyhat, int B, int N, int C,
                       int dtype, void* stream);
/* the same merge AND the LayerNorm backward of both token streams (norm1, MTFM.py:104-105) in one pass: what rssf_gate_pool_bwd followed
 * by rssf_layernorm_bwd(dxhat, x, stats_x, gamma, dx_add, dx, ..) and rssf_layernorm_bwd(dyhat, y, stats_y, gamma, NULL, dy, ..) compute
 * (the merged gradient is rounded to the activation dtype as the three launches round it; dgamma / dbeta accumulate, both streams share
 * the LayerNorm).  dxhat / dyhat are NOT modified.  dx_add (optional): the residual-path gradient added to dx.  Shapes as
 * rssf_ln_gate_pool_fwd's, C <= 128 (rssf_gate_pool_ln_bwd_supported == 1). */
int rssf_gate_pool_ln_bwd_supported(int B, int N, int C, int dtype);
int rssf_gate_pool_ln_bwd(const float* dpooled, const int32_t* argmax, const void* dxhat, const void* dyhat, const void* x, const void* y,
                          const float* stats_x, const float* stats_y, const float* gamma, const void* dx_add, void* dx, void* dy,
                          float* dgamma, float* dbeta, int B, int N, int C, int dtype, void* stream);

/* ---- Fused 7x7-window cross attention: InterlacedPoolAttention2.forward :168-188, PadBlock /
 *      LocalPermuteModule (multihead_isa_attention.py:364-426), Mhca (DAL.py:785-1030) and the
 *      residual of GeneralTransformerBlock.forward (MTFM.py:107) ------------------------------------- */
typedef struct {
  const void* x;            /* low tokens  [B,N,C] (un-normalised) */
  const void* y;            /* high tokens [B,N,C] */
  const float* stats_x;     /* LN1 {mean,rstd} per token of x */
  const float* stats_y;
  const float* omega;       /* [B][2][N] gate weights */
  const float* ln_gamma;    /* norm1 */
  const float* ln_beta;
  const float* wq; const float* bq; const float* wk; const float* bk;
  const float* wv; const float* bv; const float* wo; const float* bo;   /* [C][C], [C] */
  void* out;                /* [B,N,C] = x + attn */
  int B, H, W, C, heads, window;
  int dtype;
} rssf_winattn_fwd_params;
int rssf_winattn_fwd(const rssf_winattn_fwd_params* p, void* stream);

typedef struct {
  rssf_winattn_fwd_params f;   /* same inputs as forward (out unused) */
  const void* dout;            /* [B,N,C] gradient w.r.t. the attention term (the residual is the caller's) */
  void* dxhat;                 /* [B,N,C] grad w.r.t. LN1(x) through the attention path (gate applied) */
  void* dyhat;                 /* [B,N,C] */
  float* domega;               /* [B][2][N].  With prod_ws: overwritten.  Without: must be zeroed by the caller, accumulated
                                * with one fp32 atomic per activation element (18 M per Base launch: 43 % of the kernel) */
  float* dwq; float* dbq; float* dwk; float* dbk; float* dwv; float* dbv; float* dwo; float* dbo;  /* += */
  void* prod_ws;               /* optional scratch [2][B][N][C] of the ACTIVATION dtype (rssf_winattn_bwd_workspace_elems()): the
                                * per-element products d(x~)*LN(x), d(y~)*LN(y) are stored there and a second launch sums, in
                                * fp32, the C elements that share a gate weight */
} rssf_winattn_bwd_params;
int64_t rssf_winattn_bwd_workspace_elems(int B, int H, int W, int C);      /* elements of prod_ws */
int rssf_winattn_bwd(const rssf_winattn_bwd_params* p, void* stream);

/* ---- Convolution (implicit GEMM, MFMA) on channels-last activations ------------------------------------------
 *      replaces aten::convolution at every nn.Conv2d of the path: HRNet 3x3 / 1x1 convs
 *      (_hrnet_rssformer.py:216-287, 361-405, 512-546), neck + head (hrnet_aux.py:45-49, 78-81) and MlpDWBN's
 *      fc1 / fc2 / fused {dw + dw6 + dw12} (ffn_block.py:219-228, 246-257).
 * A convolution is a list of <= 19 "taps" (dy, dx), each with its own [Cout][Cin] slab of packed weights:
 *   out(b, oy, ox, :) = bias + sum_t W_t * in(b, (oy*mul + dy_t)/div, (ox*mul + dx_t)/div, :)      (zero outside)
 * forward:  mul = stride, div = 1, dy = ky*dil - pad.   dgrad: mul = 1, div = stride, dy = pad - ky*dil, with the
 * transposed packing (`transpose` = 1) and in/out swapped. */
/* rows of a packed slab are padded to this multiple (= the N tile the kernel will use for `cout` channels) */
int rssf_conv_tile_n(int cout);
/* elements of the packed weight buffer [ntaps][rowsP][colsP] for `dtype` */
int64_t rssf_conv_packed_elems(int ntaps, int rows, int cols, int dtype);
/* pack up to 3 torch-layout fp32 weights [Cout][Cin][k][k] (ksizes[i] = k) into per-tap slabs; tap t takes kernel
 * position kpos_of_tap[t] (= ky*k + kx) of source src_of_tap[t].  transpose = 0: rows = Cout (forward);
 * transpose = 1: rows = Cin (data gradient). */
int rssf_conv_pack(const float* w0, const float* w1, const float* w2, const int* ksizes, int nsrc, const int* src_of_tap,
                   const int* kpos_of_tap, const int* alias_of_tap, int ntaps, int Cout, int Cin, int transpose, void* out,
                   int dtype, void* stream);
/* alias_of_tap (optional, [ntaps][4] = {src, kpos, src, kpos}, -1 = none): further kernel positions that sample the SAME
 * input pixel as tap t (the centres of MlpDWBN's 1x1 + two dilated 3x3 convolutions, ffn_block.py:219-228).  Packing
 * sums their weights into tap t's slab and the weight gradient writes tap t's gradient to each of them, so that a sum of
 * convolutions runs with one tap per DISTINCT offset (17 instead of 19 for the MLP). */
/* Batched packing: ONE launch re-packs every convolution of the model (both layouts) after an optimizer step.
 * `jobs` is a DEVICE array of rssf_pack_job; `block_map` a DEVICE array of nblocks {job index, tile index} pairs, one per
 * (output channel, input channel) tile of each job (tiles of a job: rssf_conv_pack_job_blocks(); a block reads its tile's
 * source rows once, with all kernel positions, and writes every tap's slab). */
#define RSSF_MAX_TAPS 19
typedef struct rssf_pack_job {
  const float* w[3];              /* torch-layout fp32 sources (w[i] unused for i >= nsrc) */
  void* out;                      /* packed slabs [ntaps][rows_p][cols_p] of the batch dtype */
  int ks[3];
  int nsrc, ntaps, cout, cin, rows_p, cols_p, transpose;
  int src_of_tap[RSSF_MAX_TAPS];
  int kpos_of_tap[RSSF_MAX_TAPS];
  int alias_of_tap[RSSF_MAX_TAPS][4];   /* see rssf_conv_pack; {-1,-1,-1,-1} when a tap has no aliases */
} rssf_pack_job;
/* rows_p / cols_p of the packed slabs for a (rows, cols) weight matrix (rows = cout, or cin when transposed) */
int rssf_conv_packed_rows(int rows);
int rssf_conv_packed_cols(int cols, int dtype);
/* kk_total = sum over the job's sources of ks * ks (the tile shrinks with the number of kernel positions it holds) */
int rssf_conv_pack_job_blocks(int rows_p, int cols_p, int transpose, int kk_total);
/* (a job's packed image and its sources must stay below 2^31 elements: the kernel indexes them with 32-bit arithmetic) */
int rssf_conv_pack_batch(const rssf_pack_job* jobs, const int* block_map, int nblocks, int dtype, void* stream);
/* the gather convolution itself.  bias [Cout] optional; stats [RSSF_BN_SLOTS][2][Cout] optional: per-channel sum and sum of squares of
 * the OUTPUT (incl. bias) atomically accumulated for the BatchNorm that follows (fused statistics). */
int rssf_conv_gather(const void* in, const void* wpk, void* out, const float* bias, float* stats, int B, int IH, int IW,
                     int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx, int dtype,
                     void* stream);
/* the same with `addend` [B][OH][OW][Cout] (optional) added to the result in the epilogue: the data-gradient launch of a
 * residual block folds the skip-path gradient in instead of leaving a separate elementwise add to autograd */
/* stats_ws (optional, needs `stats`): DETERMINISTIC statistics (the reference trains with cuDNN off for the same
 * reason, train.py:71-73) - instead of slotted atomics every pixel tile stores its partial {sum, sumsq} into
 * stats_ws[tile][2][Cout] and a second launch adds the tiles in a fixed order into slot 0 of `stats` (which the caller zeroed):
 * two runs are bit-identical.  Size it with rssf_conv_stats_workspace_elems(). */
int64_t rssf_conv_stats_workspace_elems(int B, int OH, int OW, int Cout);
int rssf_conv_gather_add(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend,
                         float* stats_ws, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps,
                         const int* dy, const int* dx, int dtype, void* stream);
/* DATA-GRADIENT launch that also produces the BatchNorm-backward statistics of the layer whose output gradient it computes
 * (torch autograd: native_batch_norm_backward's reduction over dy, reference chain _hrnet_rssformer.py:216-246 conv -> bn -> relu ->
 * conv): `out` = d(loss)/d(act(bn(raw) + res_pre)) of the PREVIOUS layer, so
 *     bn_sums[slot][2][Cout] += { sum dz, sum dz * raw },   dz = out * act'(raw * scale + shift + res_pre)
 * exactly what rssf_bn_bwd_reduce(out, bn_raw, ...) would add - computed on the bf16 values being stored, in the epilogue that
 * already holds them, instead of a separate pass that reads `out` and `bn_raw` again.  bn_sums is [RSSF_BN_BWD_SLOTS][2][Cout],
 * zeroed by the caller; bn_res_pre optional; bn_act as in rssf_bn_*.  Shapes without a statistics epilogue run the plain
 * convolution followed by rssf_bn_bwd_reduce: the result is the same either way.  Only valid when `out` (+ addend) is the
 * COMPLETE gradient of that activation (a single consumer). */
int rssf_conv_gather_bnbwd(const void* in, const void* wpk, void* out, const void* addend, const void* bn_raw, const void* bn_res_pre,
                           const float* bn_scale_shift, int bn_act, float* bn_sums, int B, int IH, int IW, int Cin, int OH, int OW,
                           int Cout, int mul, int div, int ntaps, const int* dy, const int* dx, int dtype, void* stream);
/* FORWARD convolution whose input is still the RAW output of the producing convolution: the operand convolved is
 * act(bn(in_raw)) of the producer's BatchNorm, formed while the 3x3 halo tile is staged and zero outside the image like the
 * padding of the activation - the producer's finalize+apply pass (rssf_bn_finalize_apply) never runs and its activation is
 * never stored (conv -> bn -> relu -> conv, _hrnet_rssformer.py:216-246, when the activation has no other consumer).  The
 * launch FINALIZES that BatchNorm too: the pre_* arguments are rssf_bn_finalize's (statistics from the producer's epilogue,
 * all-reduced under SyncBN; running statistics updated in place; mean / invstd and scale / shift written for the backward pass
 * and for rssf_conv_wgrad_bnapply(in_scale_shift)).  Results are those of rssf_bn_finalize_apply + rssf_conv_gather.  Only shapes
 * for which rssf_conv_gather_preact_supported() returns 1 (bf16; 3x3 / stride 1 / "same", channels multiples of 8, Cin <= 256; or the
 * 128 -> 32 point-wise stream kernel: MlpDWBN's fc2 applying norm2 + GELU, ffn_block.py:229-236). */
int rssf_conv_gather_preact_supported(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps,
                                      const int* dy, const int* dx, int dtype);
int rssf_conv_gather_preact(const void* in_raw, const float* pre_stats, const float* pre_gamma, const float* pre_beta,
                            float* pre_running_mean, float* pre_running_var, float* pre_mean_invstd, float* pre_scale_shift, double pre_n,
                            float pre_momentum, float pre_eps, int pre_training, int pre_act, const void* wpk, void* out,
                            const float* bias, float* stats, float* stats_ws, int B, int IH, int IW, int Cin, int OH, int OW, int Cout,
                            int mul, int div, int ntaps, const int* dy, const int* dx, int dtype, void* stream);
/* weight gradient, accumulated (+=) into the torch-layout fp32 gradients of the source convs; dbias optional (+=).
 * workspace: fp32 scratch of rssf_conv_wgrad_workspace_elems() elements for the split-K partials (two-stage
 * reduction); NULL selects the slower atomic path. */
int64_t rssf_conv_wgrad_workspace_elems(int B, int OH, int OW, int Cin, int Cout, int ntaps);
/* The second (reduction) stage of the split-K gradient can be DEFERRED: with defer_reduce != NULL only the first stage is
 * launched and *defer_reduce (host memory) receives the description of the pending reduction - the workspace must then stay
 * untouched until the caller has run it through rssf_conv_wgrad_reduce_batch, which serves any number of pending
 * reductions (e.g. all ~330 convolutions of a training step) in ONE launch.  jobs / block_map are DEVICE arrays; block_map
 * holds, for each of the nblocks workgroups, {job index, block index within the job}, a job contributing
 * rssf_conv_wgrad_reduce_blocks(job) consecutive blocks (0 when nothing is pending: partial == NULL). */
typedef struct rssf_wgrad_reduce_job {
  const float* partial;           /* [ksplit][ntaps][cout][cin] fp32 */
  float* dw[3];                   /* torch-layout fp32 gradients of the source convolutions (+=) */
  int ks[3];
  int ntaps, cout, cin, ksplit;
  int src_of_tap[RSSF_MAX_TAPS];
  int kpos_of_tap[RSSF_MAX_TAPS];
  int alias_of_tap[RSSF_MAX_TAPS][4];
} rssf_wgrad_reduce_job;
int rssf_conv_wgrad(const void* dout, const void* in, float* dw0, float* dw1, float* dw2, const int* ksizes, int nsrc,
                    const int* src_of_tap, const int* kpos_of_tap, const int* alias_of_tap, float* dbias, float* workspace,
                    int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy,
                    const int* dx, rssf_wgrad_reduce_job* defer_reduce, int dtype, void* stream);
/* First stage of the same weight gradient from a TRANSPOSED copy of the input (rssf_bn_finalize_apply_planes): 128 -> 128 channels,
 * stride 1, "same" size, >= 8 taps with even dx, |dy|, |dx| <= pad, W a multiple of 128 (MlpDWBN's fused sum, ffn_block.py:226-228).
 * The input operand goes coalesced into registers, only dout passes through LDS, shared by the taps of one dy (csrc/
 * conv_wgrad_planes.hip).  Same workspace (rssf_conv_wgrad_workspace_elems covers both), same partial layout, same second stage,
 * same results up to the summation order as rssf_conv_wgrad(dout, in, ...). */
int rssf_conv_wgrad_planes_supported(int B, int H, int W, int Cin, int Cout, int stride, int ntaps, const int* dy, const int* dx, int pad,
                                     int dtype);
int rssf_conv_wgrad_planes(const void* dout, const void* in_planes, int pad, float* dw0, float* dw1, float* dw2, const int* ksizes, int nsrc,
                           const int* src_of_tap, const int* kpos_of_tap, const int* alias_of_tap, float* dbias, float* workspace, int B,
                           int H, int W, int C, int ntaps, const int* dy, const int* dx, rssf_wgrad_reduce_job* defer_reduce, int dtype,
                           void* stream);
/* WEIGHT-GRADIENT launch that also performs the BatchNorm-backward apply of its own layer (torch autograd:
 * native_batch_norm_backward's input gradient + the activation mask, then convolution_backward's weight gradient): the
 * output-gradient operand of the weight gradient is draw = rssf_bn_bwd_apply(bn_dy, bn_raw, ...), so the halo-tiled 3x3 kernel
 * forms it while staging that operand (reads bn_dy and bn_raw instead of draw) and also writes it to `draw` for the data-gradient
 * launch that follows (+ dz to `dres`, + dgamma / dbeta): the separate apply pass (read dy, raw; write draw) disappears.  Same
 * operation order as rssf_bn_bwd_apply: bit-identical `draw`, `dres`, parameter gradients and weight gradient.  Shapes without
 * such a kernel run rssf_bn_bwd_apply followed by rssf_conv_wgrad(draw, ...).  The BatchNorm arguments are rssf_bn_bwd_apply's,
 * the rest rssf_conv_wgrad's (its `dout` is `draw`).
 * in_scale_shift (optional, [2][Cin]) / in_act: `in` is then the RAW output of the producing convolution and the operand
 * contracted is act(in * scale + shift) - the layer was run forward by rssf_conv_gather_preact; only where
 * rssf_conv_wgrad_preact_supported() says so: 1 = a kernel that takes the pre-activation operand AND the fused apply (3x3 / stride 1,
 * bias-free), 2 = the pre-activation operand only (MlpDWBN's fc2, 32 <- 128 point-wise, ffn_block.py:236: with BatchNorm arguments the
 * apply runs as its separate pass first), 0 = none.
 * rssf_conv_wgrad_preact: rssf_conv_wgrad with such an input operand and no BatchNorm of its own in the call (a layer whose backward
 * apply is a different pass, e.g. rssf_bn_bwd_apply_post). */
int rssf_conv_wgrad_preact_supported(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, int nsrc,
                                     const int* dy, const int* dx, int has_bias, int dtype);
int rssf_conv_wgrad_preact(const void* dout, const void* in_raw, const float* in_scale_shift, int in_act, float* dw0, float* dw1, float* dw2,
                           const int* ksizes, int nsrc, const int* src_of_tap, const int* kpos_of_tap, const int* alias_of_tap, float* dbias,
                           float* workspace, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy,
                           const int* dx, rssf_wgrad_reduce_job* defer_reduce, int dtype, void* stream);
int rssf_conv_wgrad_bnapply(const void* bn_dy, const void* bn_raw, const float* bn_scale_shift, const float* bn_mean_invstd,
                            const float* bn_sums, const void* bn_res_pre, void* draw, void* dres, float* dgamma, float* dbeta,
                            int bn_act, double bn_n, int bn_training, float param_grad_scale, const void* in,
                            const float* in_scale_shift, int in_act, float* dw0, float* dw1,
                            float* dw2, const int* ksizes, int nsrc, const int* src_of_tap, const int* kpos_of_tap,
                            const int* alias_of_tap, float* dbias, float* workspace, int B, int IH, int IW, int Cin, int OH, int OW,
                            int Cout, int stride, int ntaps, const int* dy, const int* dx, rssf_wgrad_reduce_job* defer_reduce,
                            int dtype, void* stream);
/* A point-wise (1x1, stride 1, one bias) layer's WHOLE backward in one launch: rssf_conv_wgrad_bnapply (BatchNorm-backward apply on the way:
 * draw, dgamma, dbeta; weight gradient into the split-K workspace; bias gradient) plus its data gradient dx[p][ci] = sum_co draw[p][co] * W[co][ci]
 * (`weight`: the fp32 master weights [Cout][Cin], rounded to bf16 as rssf_conv_pack does) - the draw tile a block has formed in LDS is the
 * data gradient's operand, so dx costs no further pass over the 4 C-channel tensors (MlpDWBN's fc1, ffn_block.py:218-224: the launch that
 * would re-read draw to form dx disappears).  Only where rssf_conv_wgrad_bnapply_dgrad_supported() returns 1 (bf16, 128 <- 32 channels,
 * no residual before the activation); same results as rssf_conv_wgrad_bnapply + rssf_conv_gather (transposed pack) up to the summation order. */
int rssf_conv_wgrad_bnapply_dgrad_supported(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy,
                                            const int* dx, int has_res_pre, int dtype);
int rssf_conv_wgrad_bnapply_dgrad(const void* bn_dy, const void* bn_raw, const float* bn_scale_shift, const float* bn_mean_invstd,
                                  const float* bn_sums, void* draw, float* dgamma, float* dbeta, int bn_act, double bn_n, int bn_training,
                                  float param_grad_scale, const void* in, const float* weight, void* dx_out, float* dw, float* dbias,
                                  float* workspace, int B, int H, int W, int Cin, int Cout, rssf_wgrad_reduce_job* defer_reduce, int dtype,
                                  void* stream);
/* The same for a point-wise layer whose INPUT is the raw output of the producing convolution (run forward by rssf_conv_gather_preact) and
 * whose own BatchNorm-backward apply is a different pass: weight gradient (rssf_conv_wgrad_preact), bias gradient, the data gradient
 * dx = dout * W AND the producer's BatchNorm-backward statistics {sum dz, sum dz * in_raw}, dz = dx * act'(in_raw * scale + shift)
 * (rssf_conv_gather_bnbwd's epilogue) into in_bn_sums [RSSF_BN_BWD_SLOTS][2][Cin] (zeroed by the caller) - one pass over the 4 C-channel
 * raw tensor instead of two (MlpDWBN's fc2 behind norm2, ffn_block.py:229-236).  bf16, 32 <- 128 channels. */
int rssf_conv_wgrad_preact_dgrad_supported(int B, int H, int W, int Cin, int Cout, int dtype);
int rssf_conv_wgrad_preact_dgrad(const void* dout, const void* in_raw, const float* in_scale_shift, int in_act, const float* weight, void* dx_out,
                                 float* in_bn_sums, float* dw, float* dbias, float* workspace, int B, int H, int W, int Cin, int Cout,
                                 rssf_wgrad_reduce_job* defer_reduce, int dtype, void* stream);
int rssf_conv_wgrad_reduce_blocks(const rssf_wgrad_reduce_job* job);
int rssf_conv_wgrad_reduce_batch(const rssf_wgrad_reduce_job* jobs, const int* block_map, int nblocks, void* stream);

/* ---- BatchNorm2d (+ activation + residual adds), channels-last: nn.BatchNorm2d / nn.SyncBatchNorm call sites of
 *      _hrnet_rssformer.py, hrnet_aux.py:47 and ffn_block.py:222-234 (momentum 0.1, eps 1e-5) -------------------- */
/* act: 0 none, 1 ReLU, 2 GELU(erf).  stats = [RSSF_BN_SLOTS][2][C] {sum, sumsq} over n samples (from rssf_conv_gather; all-reduced by
 * the host for SyncBN).  Writes mean_invstd [2][C], scale_shift [2][C]; updates running stats when training. */
int rssf_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float* mean_invstd, float* scale_shift, int C, double n, float momentum, float eps, int training,
                     void* stream);
/* y = act(raw*scale + shift + res_pre) + res_post   (res_* optional, same shape as raw).
 * act | RSSF_ACT_POST_RELU: y = relu(act(..) + res_post) - `self.relu(self.transformer(..))` of HighResolutionModule.forward
 * (_hrnet_rssformer.py:435) inside the last BatchNorm pass of MlpDWBN; its backward: rssf_bn_bwd_reduce_post / _apply_post */
#define RSSF_ACT_POST_RELU 16
int rssf_bn_apply(const void* raw, const float* scale_shift, const void* res_pre, const void* res_post, void* y,
                  int64_t rows, int C, int act, int dtype, void* stream);
/* rssf_bn_finalize followed by rssf_bn_apply as ONE launch (same arguments, same results) */
int rssf_bn_finalize_apply(const void* raw, const float* stats, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* mean_invstd, float* scale_shift, const void* res_pre, const void* res_post,
                           void* y, int64_t rows, int C, int act, double n, float momentum, float eps, int training, int dtype,
                           void* stream);
/* the same (no residuals) that ALSO writes the activation TRANSPOSED and zero-padded, y_planes[C][B][H + 2 pad][W + 2 pad] (pixels
 * contiguous; only the interior is written: zero the buffer once) - the input operand of rssf_conv_wgrad_planes, formed in the pass
 * that holds every value anyway (reference: the BatchNorm + GELU between MlpDWBN's fc1 and its depth-wise sum, ffn_block.py:219-228).
 * y and the BatchNorm results are those of rssf_bn_finalize_apply. */
int rssf_bn_finalize_apply_planes_supported(int B, int H, int W, int C, int pad, int dtype);
int rssf_bn_finalize_apply_planes(const void* raw, const float* stats, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float* mean_invstd, float* scale_shift, void* y, void* y_planes, int B, int H, int W,
                                  int C, int pad, int act, double n, float momentum, float eps, int training, int dtype, void* stream);
/* sums [RSSF_BN_BWD_SLOTS][2][C] fp32, zeroed by the caller: slot (block mod slots) += { sum dz, sum dz*raw },
 * dz = dy * act'(raw*scale + shift + res_pre).  rssf_bn_bwd_apply sums the slots (all-reduce the whole buffer for SyncBN) */
/* det_ws (optional): deterministic mode - per-block partials [blocks][2][C] (rssf_bn_bwd_reduce_workspace_elems() floats),
 * added in block order into slot 0 of `sums` by a second launch; NULL: slotted atomics */
int64_t rssf_bn_bwd_reduce_workspace_elems(int64_t rows, int C);
int rssf_bn_bwd_reduce(const void* dy, const void* raw, const float* scale_shift, const void* res_pre, float* sums,
                       int64_t rows, int C, int act, float* det_ws, int dtype, void* stream);
/* draw = d(loss)/d(raw); dres (optional) = dz = gradient of res_pre; dgamma/dbeta (optional) accumulated:
 * dgamma += param_grad_scale * sum(dz*xhat), dbeta += param_grad_scale * sum(dz).  param_grad_scale is 1, or 1/world when
 * `sums` were all-reduced for SyncBN: the data-parallel mean of the LOCAL parameter gradients (torch SyncBatchNorm + DDP
 * semantics, configs/base/loveda.py:106-108) equals the global sums / world. */
int rssf_bn_bwd_apply(const void* dy, const void* raw, const float* scale_shift, const float* mean_invstd, const float* sums,
                      const void* res_pre, void* draw, void* dres, float* dgamma, float* dbeta, int64_t rows, int C, int act,
                      double n, int training, float param_grad_scale, int dtype, void* stream);

/* ---- Grouped launches: the parallel branches of a HighResolutionModule (_hrnet_rssformer.py:216-246 BasicBlock, :410-423
 *      `x[i] = self.branches[i](x[i])`) run the SAME step at the same moment on independent tensors - 128^2 x 32, 64^2 x 64,
 *      32^2 x 128, 16^2 x 256 for Base.  Each of those launches is bound by the latency chain of its blocks, not by a roofline;
 *      the entry points below take the n problems of such a step and run them as ONE grid (blocks of the long-chain problems
 *      first, the others fill their bubbles; one launch ramp and tail instead of n).  Results per problem are those of the
 *      single-problem entry point named with each struct (same kernels' block bodies).  n > RSSF_GROUP_MAX, mixed kinds or
 *      shapes without a grouped kernel run problem by problem through those entry points - always valid, never faster. */
#define RSSF_GROUP_MAX 4
/* rssf_conv_gather_add / _bnbwd / _preact of a 3x3, stride-1, "same" convolution (taps in forward order; mirrored = 1:
 * data-gradient order, i.e. dy/dx negated, wpk the transposed pack).  Unused optional parts are NULL. */
typedef struct rssf_conv3x3_item {
  const void* in; const void* wpk; void* out;
  float* stats;                                        /* [RSSF_BN_SLOTS][2][Cout] of the following BatchNorm, or NULL */
  const void* addend;                                  /* added to out, or NULL */
  const void* bn_raw; const void* bn_res; const float* bn_ss; float* bn_sums;   /* rssf_conv_gather_bnbwd (bn_sums != NULL) */
  const float* pre_stats; const float* pre_gamma; const float* pre_beta;        /* rssf_conv_gather_preact (pre_ss != NULL) */
  float* pre_running_mean; float* pre_running_var; float* pre_mean_invstd; float* pre_ss;
  double pre_n;
  float pre_momentum, pre_eps;
  int pre_training, pre_act, bn_act;
  int B, H, W, Cin, Cout;
} rssf_conv3x3_item;
int rssf_conv3x3_group(const rssf_conv3x3_item* items, int n, int mirrored, int dtype, void* stream);
/* rssf_conv_wgrad / rssf_conv_wgrad_bnapply of a bias-free 3x3, stride-1, "same" convolution (bn_dy != NULL: the fused
 * BatchNorm-backward apply, `draw` is then an OUTPUT and `dout` is ignored; in_ss != NULL: pre-activation input operand).
 * workspace: rssf_conv_wgrad_workspace_elems(..., 9) floats per item; defer_reduce as in rssf_conv_wgrad. */
typedef struct rssf_wgrad3x3_item {
  const void* dout; const void* in; float* dw; float* workspace; struct rssf_wgrad_reduce_job* defer_reduce;
  const void* bn_dy; const void* bn_raw; const float* bn_ss; const float* bn_mi; const float* bn_sums; const void* bn_res;
  void* draw; void* dres; float* dgamma; float* dbeta;
  const float* in_ss;
  double bn_n;
  float pscale;
  int bn_act, bn_training, in_act;
  int B, H, W, Cin, Cout;
} rssf_wgrad3x3_item;
int rssf_conv3x3_wgrad_group(const rssf_wgrad3x3_item* items, int n, int dtype, void* stream);
/* rssf_bn_finalize_apply per item */
typedef struct rssf_bn_apply_item {
  const void* raw; const float* stats; const float* gamma; const float* beta; float* running_mean; float* running_var;
  float* mean_invstd; float* scale_shift; const void* res_pre; const void* res_post; void* y;
  int64_t rows;
  double n;
  float momentum, eps;
  int C, act, training;
} rssf_bn_apply_item;
int rssf_bn_finalize_apply_group(const rssf_bn_apply_item* items, int n, int dtype, void* stream);
/* rssf_bn_bwd_reduce per item (slotted atomics; the deterministic workspace form is not grouped) */
typedef struct rssf_bn_reduce_item {
  const void* dy; const void* raw; const float* scale_shift; const void* res_pre; float* sums;
  int64_t rows;
  int C, act;
} rssf_bn_reduce_item;
int rssf_bn_bwd_reduce_group(const rssf_bn_reduce_item* items, int n, int dtype, void* stream);
/* rssf_bn_bwd_apply per item: the BatchNorm-backward apply of the layers whose weight gradient cannot carry it (the 1x1 and strided
 * fuse convolutions of one depth of HighResolutionModule._make_fuse_layers, _hrnet_rssformer.py:361-405) as one grid */
typedef struct rssf_bn_bwd_apply_item {
  const void* dy; const void* raw; const float* scale_shift; const float* mean_invstd; const float* sums; const void* res_pre;
  void* draw; void* dres; float* dgamma; float* dbeta;
  int64_t rows;
  double n;
  int C, act, training;
  float param_grad_scale;
} rssf_bn_bwd_apply_item;
int rssf_bn_bwd_apply_group(const rssf_bn_bwd_apply_item* items, int n, int dtype, void* stream);

/* rssf_bn_bwd_reduce / rssf_bn_bwd_apply of a layer run forward with res_post (and possibly RSSF_ACT_POST_RELU in `act`): the
 * gradient that enters the activation is g = dy where y > 0 (all of dy without the flag); dpost (optional) = g is the gradient of
 * res_post.  det_ws as in rssf_bn_bwd_reduce. */
int rssf_bn_bwd_reduce_post(const void* dy, const void* raw, const float* scale_shift, const void* res_pre, const void* res_post,
                            float* sums, int64_t rows, int C, int act, float* det_ws, int dtype, void* stream);
int rssf_bn_bwd_apply_post(const void* dy, const void* raw, const float* scale_shift, const float* mean_invstd, const float* sums,
                           const void* res_pre, const void* res_post, void* draw, void* dres, void* dpost, float* dgamma,
                           float* dbeta, int64_t rows, int C, int act, double n, int training, float param_grad_scale, int dtype,
                           void* stream);

/* ---- Device input pipeline: RandomCrop -> OneOf(HorizontalFlip, VerticalFlip, RandomRotate90) ->
 *      ShiftScaleRotate -> Normalize -> ToTensor and the LoveDA `mask - 1` shift (configs/base/loveda.py:18-36,
 *      data/loveda.py:82-91) as one gather over a device-resident uint8 dataset img [nsrc][SH][SW][3], mask [nsrc][SH][SW]
 *      (optional).  params [B][4] (device, int32) = {source image, crop y0, crop x0, op}; affine (optional) [B][6] (device, double):
 *      the INVERSE 2x3 matrix of the image's ShiftScaleRotate warp (cv2.invertAffineTransform of getRotationMatrix2D + shift), a NaN
 *      in its first element = "this image is not warped"; the warp follows cv2.warpAffine on 8-bit data (fixed-point coordinates,
 *      1/32-step bilinear taps, nearest for the mask, BORDER_REFLECT_101).  out_img [B][OH][OW][3] channels-last of `dtype`;
 *      out_mask [B][OH][OW] int64.  Normalize as albumentations does it: (float32(v) - mean*max_pixel_value) *
 *      reciprocal(std*max_pixel_value).  The rot90 ops need a square crop. ---------------------------------------------------- */
#define RSSF_AUG_NONE 0
#define RSSF_AUG_HFLIP 1
#define RSSF_AUG_VFLIP 2
#define RSSF_AUG_ROT90 3        /* + k, k = 0..3 quarter turns counter-clockwise (np.rot90) */
int rssf_input_pipeline(const uint8_t* img, const uint8_t* mask, const int* params, const double* affine, void* out_img,
                        int64_t* out_mask, int B, int nsrc, int SH, int SW, int OH, int OW, const float* mean3, const float* std3,
                        float max_pixel_value, int dtype, void* stream);

/* ---- Up-sampling, channels-last ------------------------------------------------------------------------------
 * bilinear with align_corners=True: F.interpolate x3 in SimpleFusion8 (hrnet_aux.py:61-65), UpsamplingBilinear2d(x4)
 * of the head (:80).  backward = 0: in [B,IH,IW,C] -> out [B,OH,OW,C];  backward = 1: `in` is the gradient
 * [B,OH,OW,C], `out` receives the input gradient [B,IH,IW,C] (gather form, no atomics). */
int rssf_upsample_bilinear(const void* in, void* out, int B, int IH, int IW, int OH, int OW, int C, int backward, int dtype,
                           void* stream);
/* The same with the full-resolution tensor addressed as a CHANNEL SLICE of a wider channels-last tensor (pixel stride
 * ld_wide >= C elements, pointer already offset to the slice): forward writes straight into a concatenation buffer
 * (SimpleFusion8's torch.cat, hrnet_aux.py:64), backward reads the slice of the concatenated gradient - no cat / split
 * copies.  IH == OH, IW == OW is the identity (the un-resized branch 0). */
int rssf_upsample_bilinear_slice(const void* in, void* out, int B, int IH, int IW, int OH, int OW, int C, int ld_wide, int backward,
                                 int dtype, void* stream);
/* Evaluation head in one pass: bilinear (align_corners=True) resize of the class logits [B,IH,IW,K] (K <= 32) to [B,OH,OW], softmax
 * over K and/or argmax - nn.UpsamplingBilinear2d + `logit.softmax(dim=1)` of HRNetFusion.forward (hrnet_aux.py:80, 103-104) and
 * `pred.argmax(dim=1)` of eval.py:66 / predict.py:42.  probs [B,OH,OW,K] fp32 channels-last (optional), pred [B,OH,OW] int32
 * (optional; the first maximum wins, as torch.argmax); at least one of them. */
int rssf_head_upsample_softmax(const void* logits, float* probs, int32_t* pred, int B, int IH, int IW, int OH, int OW, int K,
                               int dtype, void* stream);
/* nearest up-sampling by an integer factor fused with the running branch sum of the HRNet fuse layers
 * (_hrnet_rssformer.py:380, 424-427).  backward = 0: out [B,IH*s,IW*s,C] = (acc ? acc : 0) + up(in);
 * backward = 1: `in` is the output gradient, `out` [B,IH,IW,C] = its s x s block sums (acc ignored). */
int rssf_upsample_nearest_add(const void* acc, const void* in, void* out, int B, int IH, int IW, int scale, int C, int backward,
                              int dtype, void* stream);
/* The whole fuse sum of one HighResolutionModule output in one pass (_hrnet_rssformer.py:424-435: `y = y + fuse[i][j](x[j])` over
 * j): backward = 0: io [B,OH,OW,C] = sum_k up(terms[k] [B,OH/s_k,OW/s_k,C], s_k), scales[k] = 1 for the terms at the output's
 * resolution; fp32 accumulation in the order given, rounded once.  backward = 1: io is the output gradient, terms[k] receives its
 * s_k x s_k block sums for every s_k > 1 (all of them in one grid); terms[k] may be null for s_k = 1 (that gradient is io itself).
 * nterms <= 4; C a multiple of the 16-byte vector.  terms / scales are HOST arrays. */
int rssf_upsample_nearest_sum(const void* const* terms, const int* scales, int nterms, void* io, int B, int OH, int OW, int C,
                              int backward, int dtype, void* stream);

/* ---- image-level auxiliary head: `self.headaux(self.avg_pool(f0).flatten(1))` (module/baseline/hrnet_aux.py:86-87, 99-100):
 *      global average pool of the channels-last branch-0 feature [B, HW, C] (C <= 64) -> Linear(C, K) (K <= 16), forward only (the
 *      loss reads it under no_grad).  workspace: rssf_aux_head_workspace_elems() floats; out [B][K] fp32. ---- */
int64_t rssf_aux_head_workspace_elems(int B, int C);
int rssf_aux_head_fwd(const void* feat, const float* weight, const float* bias, float* workspace, float* out, int B, int HW, int C,
                      int K, int dtype, void* stream);

/* ---- nn.MaxPool2d(3, stride 2, padding 1), channels-last, forward only: the stem pooling of the ResNet-50 CAM inference path
 *      (WaveCAM-TMM2023/net/resnet50.py:68,88 - BASELINE config 5's conv-only relative).  out [B, (IH-1)/2+1, (IW-1)/2+1, C]. ---- */
int rssf_maxpool3x3s2(const void* in, void* out, int B, int IH, int IW, int C, int dtype, void* stream);

/* ---- CGFL loss: SegmentationLossaux.forward (module/CGFL.py:201-227) -> MCTransAuxLoss (losses/auxloss.py:257-305)
 *      -> softmax_focalloss (module/CGFL.py:72-101), on channels-last logits [B, HW, K] and int64 labels [B, HW] ---- */
/* acc: fp32 scratch of RSSF_LOSS_ACC_ELEMS floats per sample (zeroed inside; the six partial values of a sample lie on six cache
 * lines of their own - atomics from many workgroups to one line serialise); aux [B][KA] fp32 image-level scores (KA = 7 in the reference);
 * KA == 1: aux [B] IS the per-sample gamma (the l1 vector MCTransAuxLoss returns) - the call shape of softmax_focalloss(.., gamma=l1), CGFL.py:72, 221;
 * out[0] = loss, out[1] = backward coefficient (detached modulating bracket / n_valid).  A label outside [0, K) that is not
 * ignore_index (F.cross_entropy asserts on it) makes both NaN: the failure is loud, no out-of-range read happens. */
/* deterministic != 0: one block per sample instead of up to 64 (no cross-block float atomics): bit-identical loss */
#define RSSF_LOSS_ACC_ELEMS 192
int rssf_cgfl_loss_fwd(const void* logits, const int64_t* labels, const float* aux, float* acc, float* out, int B, int HW, int K,
                       int KA, int ignore_index, int deterministic, int dtype, void* stream);
/* dlogits = dloss * out[1] * (softmax(logits) - onehot(label)) on valid pixels, 0 on ignored ones; dloss may be NULL (=1) */
int rssf_cgfl_loss_bwd(const void* logits, const int64_t* labels, const float* out, const float* dloss, void* dlogits, int B, int HW,
                       int K, int ignore_index, int dtype, void* stream);

/* ---- evaluation: eval.py:66-71 (`pred.argmax(dim=1)`, ignore(-1) mask, er.metric.PixelMetric.forward) ----------------
 * scores [npix][K] channels-last class scores (logits or probabilities: same argmax), labels [npix] int64 (optional),
 * pred [npix] int32 (optional), cm [K][K] int64 += bincount(label*K + pred) over pixels whose label != ignore_index. */
int rssf_argmax_confusion(const void* scores, const int64_t* labels, int32_t* pred, int64_t* cm, int64_t npix, int K,
                          int ignore_index, int dtype, void* stream);

/* ---- optimizer over flat fp32 buffers: the external `ever` trainer's clip_grad_norm_(35) + SGD(momentum .9,
 *      wd 1e-4) of configs/base/loveda.py:68-77 as two launches over all parameters ----------------------- */
/* p[0..n) = 0 as a KERNEL launch.  Inside a captured hipGraph a memset NODE (hipMemsetAsync, which is also what some framework
 * zero-fills lower to) stopped taking effect after a device synchronisation between replays on ROCm 7.0 / MI355X; every buffer the
 * training step clears per replay (flat gradient, statistics pool) goes through this instead. */
int rssf_zero_f32(float* p, int64_t n, void* stream);
/* Small glue launches (csrc/small.hip).  out = a + b (+ c): the ONE epilogue bias of MlpDWBN's three summed convolutions
 * (reference modules/ffn_block.py:226-228, 250-257); d0 (d1, d2) += src: their common bias gradient.  fp32, c / d1 / d2 may be null. */
int rssf_vec_sum3(const float* a, const float* b, const float* c, float* out, int n, void* stream);
int rssf_vec_add_to3(const float* src, float* d0, float* d1, float* d2, int n, void* stream);
/* out = a + b over n elements of `dtype` (out may alias a or b; 16-byte aligned): `fuse(x)` running sums of
 * HighResolutionModule.forward (_hrnet_rssformer.py:424-435) and gradient sums no convolution epilogue carries */
int rssf_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
/* out = a + b + c (fp32 sum, rounded once; out may alias an operand): the gradient of a HighResolutionModule branch output x[j],
 * which feeds its own fuse sum, the 1x1 convolution towards output 0 and the accumulating convolutions towards the other outputs
 * (_hrnet_rssformer.py:424-435) - what autograd's own accumulation would sum with two element-wise framework launches */
int rssf_add3(const void* a, const void* b, const void* c, void* out, int64_t n, int dtype, void* stream);
/* Channel padding of the operands whose channel count is no multiple of the 16-byte vector (the 6-class head gradient of
 * SimpleFusion8 / hrnet_aux.py:51-68, the 3-channel stem input): dst[r][0..Cp) = src[r][0..C), zeros behind; and the way back for
 * the parameter gradients of such a layer: dst[r][c] += src[r][c], c < cols, with row pitches (fp32) - what F.pad and a sliced
 * add_ did with two framework launches each. */
int rssf_pad_channels(const void* src, void* dst, int64_t rows, int C, int Cp, int dtype, void* stream);
int rssf_add_rows(float* dst, const float* src, int rows, int cols, int ld_dst, int ld_src, void* stream);
/* fp32 image [B,C,H,W] given by its element strides (NCHW or channels-last memory) -> channels-last [B,H,W,Cp] of `dtype`, channels
 * zero-padded to the 16-byte vector (Cp = 8 bf16 / 4 fp32): the network input of HighResolutionNet.forward
 * (_hrnet_rssformer.py:605-613) in one launch. */
int rssf_image_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                       int dtype, void* stream);
/* out[0] = sum g^2; `out` holds 1 + RSSF_SQNORM_BLOCKS floats (out[1..] = per-block partials, added in a fixed order:
 * bit-identical on every data-parallel replica, no float atomics). */
#define RSSF_SQNORM_BLOCKS 2048
int rssf_grad_sqnorm(const float* g, int64_t n, float* out, void* stream);
/* g' = grad_scale*g*clip + wd*p ; buf = first_step ? g' : mu*buf + g' ; p -= lr*buf
 * clip = max_norm > 0 ? min(1, max_norm / (grad_scale*sqrt(*sqnorm) + 1e-6)) : 1   (device-side, no host sync). */
/* lr_dev (optional): device pointer to the learning rate, read at execution time (hipGraph replay); else `lr`. */
int rssf_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, const float* sqnorm, float grad_scale,
                  float max_norm, const float* lr_dev, float lr, float momentum, float weight_decay, int first_step,
                  void* stream);

/* ---- data-parallel exchange: what the reference gets from DistributedDataParallel + nn.SyncBatchNorm
 *      (`ever` th_amp_ddp trainer, train.py:79; configs/base/loveda.py:106-108 sync_bn; modules/ffn_block.py:222-234) as
 *      RCCL collectives enqueued on the CALLER's stream (capturable into the step's hipGraph).  RCCL is bound at run time:
 *      rccl_path names the library to dlopen (the process's existing RCCL, e.g. <torch>/lib/librccl.so), NULL = default
 *      search.  One process per GPU; the 128-byte id comes from rank 0 (rssf_comm_unique_id) over the caller's rendezvous. -- */
typedef struct rssf_comm rssf_comm;
int rssf_comm_unique_id(void* id128, const char* rccl_path);
int rssf_comm_init(rssf_comm** comm, int rank, int world, const void* id128, const char* rccl_path);
int rssf_comm_rank(const rssf_comm* comm);
int rssf_comm_world(const rssf_comm* comm);
int rssf_comm_nranks(const rssf_comm* comm);     /* ncclCommCount: the rank count RCCL itself reports for the communicator (0: none, -1: error) */
/* in-place sum over ranks of one flat gradient bucket (`count` elements of `dtype`); the 1/world factor of the mean is
 * folded into rssf_sgd_step's grad_scale */
int rssf_allreduce_bucket(void* buf, int64_t count, int dtype, rssf_comm* comm, void* stream);
/* in-place sum over ranks of a BatchNorm statistics buffer ([slots][2][C] fp32: forward {sum, sumsq}, backward
 * {sum dz, sum dz*raw}); the caller multiplies the sample count by rssf_comm_world() */
int rssf_syncbn_exchange(float* stats, int64_t count, rssf_comm* comm, void* stream);
int rssf_comm_destroy(rssf_comm* comm);

/* Peer-to-peer SyncBN statistics exchange (csrc/p2p.hip): what nn.SyncBatchNorm's all_gather /
 * all_reduce of per-layer statistics (modules/ffn_block.py:222-234, configs/base/loveda.py:106-108) costs as ONE single-workgroup
 * kernel per rank - peer writes into hipIpc-mapped fine-grained windows + flags over xGMI, rank-ordered sum (bit-identical on every
 * rank), replay-safe inside a captured hipGraph.  The object owns its window (like an RCCL communicator owns its buffers);
 * `ipc_handle64` = the 64 bytes of its hipIpcMemHandle_t, carried to the peers by the caller's rendezvous; a CHANNEL is an
 * independent exchange sequence (one per stream that issues exchanges).  `stats` holds, per layer i, an [nslots][item_n[i]] block
 * at element offset item_off[i] (item_n = 2C): on return slot 0 holds the sum over slots AND ranks, the other slots are zero.
 * A peer that does not show up within the time-out (rssf_p2p_set_timeout_ms) sets the error word rssf_p2p_status reports. */
#define RSSF_P2P_MAX_FLOATS 4096
#define RSSF_P2P_MAX_ITEMS 8
typedef struct rssf_p2p rssf_p2p;
int rssf_p2p_create(rssf_p2p** p2p, int rank, int world, int channels, void* ipc_handle64);
int rssf_p2p_connect(rssf_p2p* p2p, int peer, const void* ipc_handle64);
/* the same for a peer that lives in THIS process (one rank per stream or per device of one process; a hipIpc handle cannot be opened
 * by the process that exported it): rank `peer` is `other`, which must outlive every exchange of `p2p` */
int rssf_p2p_connect_local(rssf_p2p* p2p, int peer, rssf_p2p* other);
int rssf_p2p_exchange(rssf_p2p* p2p, int channel, float* stats, const int* item_off, const int* item_n, int nitems, int nslots,
                      void* stream);
/* bound of the wait for a peer in the exchanges launched from now on; 0 = unbounded (what a collective does).  A new object starts
 * with 10 000 ms. */
int rssf_p2p_set_timeout_ms(rssf_p2p* p2p, int ms);
int rssf_p2p_status(rssf_p2p* p2p, int* timed_out);
/* diagnosis of a multi-GPU run: the time the exchanges of `channel` have spent WAITING for their peers' words since the object was
 * created (or since the last call with reset != 0), in microseconds, and how many exchanges that was - counted on the device by the
 * exchange kernel itself (it rides in the captured step), read here with a synchronising copy.  What a rank waits is the skew of the
 * slowest peer plus the xGMI write / poll latency: `bench.py --gpus N` prints it per rank. */
int rssf_p2p_wait_us(rssf_p2p* p2p, int channel, double* wait_us, int64_t* exchanges, int reset);
int rssf_p2p_destroy(rssf_p2p* p2p);

/* ---- Mix-Transformer (SegFormer MiT) inference operators of the SCD class-activation-map path: BASELINE config 5 as worded
 *      (SCD-AAAI2023/network/mix_transformer.py:93-131 Attention.forward, :377-388 DWConv, :45-52 Mlp.forward;
 *      network/TSCD_model.py:66-79; utils/camutils.py:85-113 multi_scale_cam).  Forward only (the reference extracts CAMs under
 *      no_grad). ---- */
/* out[b, n, h*d + :] = softmax_m(q[b, n, h] . k[b, m, h] * scale) v[b, m, h]; q [B, N, heads*d], kv [B, M, 2*heads*d] (the
 * reference's `kv` Linear output: k of head h at channel h*d, v at heads*d + h*d), d = head_dim in {32, 64}.  logits (optional)
 * [B, heads, N, M] fp32 receives the RAW q.k products (Attention.forward's `attn_`, returned to the caller by the reference). */
int rssf_mha_fwd(const void* q, const void* kv, void* out, float* logits, int B, int N, int M, int heads, int head_dim, float scale,
                 int dtype, void* stream);
/* y = act(depthwise_conv3x3(x) + bias), channels-last [B, H, W, C], w [C][3][3] fp32 (nn.Conv2d(C, C, 3, 1, 1, groups=C)),
 * act 0 = none, 2 = GELU (exact erf form) */
int rssf_dwconv3x3(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int act, int dtype,
                   void* stream);
/* out[b][p] = sigmoid(bias + sum_h w[h] a0[b][h][p] + w[heads + h] a1[b][h][p]), p < plane: TSCD.forward's
 * sigmoid(attn_proj(cat(attns[-2:], 1)))[:, 0] on the raw logits of the last two blocks */
int rssf_attn_proj_sigmoid(const float* a0, const float* a1, const float* w, const float* bias, float* out, int B, int heads,
                           int64_t plane, void* stream);
/* the same prediction straight from the projections of the two blocks (q_s [B, N, heads*d], kv_s [B, M, 2*heads*d]):
 * out[b][n][m] = sigmoid(bias + sum_{s,h} w[s*heads + h] * q_s[b,n,h,:] . k_s[b,m,h,:]); the per-head logit tensors never exist */
int rssf_attn_pred(const void* q0, const void* kv0, const void* q1, const void* kv1, const float* w, const float* bias, float* out,
                   int B, int N, int M, int heads, int head_dim, int dtype, void* stream);
/* F.interpolate(size=(OH, OW), mode='bilinear', align_corners=False), channels-last [B, IH, IW, C] (C = 1 with B = planes for
 * planar tensors) */
int rssf_resize_bilinear(const void* in, void* out, int B, int IH, int IW, int OH, int OW, int C, int dtype, void* stream);
/* one scale of multi_scale_cam: cam [2B, CH, CW, K] channels-last (images, then their horizontal flips) ->
 * acc[b][k][y][x] (+)= relu(max(up(cam[b])(y, x), up(cam[B + b])(y, W-1-x))), up = align_corners=False bilinear to H x W */
int rssf_cam_merge(const void* cam, float* acc, int B, int K, int CH, int CW, int H, int W, int accumulate, int dtype, void* stream);
/* per plane of n elements: x <- (x - min x) / (max x - min x + 1e-5) */
int rssf_cam_normalize(float* cam, int planes, int64_t n, void* stream);

/* ---- test hooks -------------------------------------------------------------------------------------- */
/* D[16][16] = A[16][K] * B[16][K]^T through the library's MFMA tile helper (layout self-check). */
int rssf_debug_mma(const void* a, const void* b, float* d, int K, int dtype, void* stream);
/* probe of the DPP / v_permlane{16,32}_swap lane reductions: in[64] -> out[10][64] = per lane {xor-16 pair sum, xor-32 pair sum,
 * 4-row sum, 4-row max, wave sum, wave max}, then the two results of swap16 and of swap32 on (1000 + lane, 2000 + lane) */
int rssf_debug_lane_reduce(const float* in, float* out, void* stream);
/* probe of the LDS transpose read (ds_read_b64_tr_b16): lds[i] = i, lane l reads at element address addr[l] */
int rssf_debug_trread(const int* addr, short* out, void* stream);
/* TEST HOOK, not part of any training / inference path (tests/test_gpu_trainer.py::test_step_does_not_read_uninitialised_lds is its one
 * caller): fills the LDS of every CU with `pattern` (e.g. 0x7fc07fc0: bf16 / fp32 NaNs), so that a kernel launched next that reads LDS it
 * never wrote sees it (scratch4: 4 bytes of device memory) */
int rssf_debug_poison_lds(unsigned pattern, void* scratch4, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSSF_H */
