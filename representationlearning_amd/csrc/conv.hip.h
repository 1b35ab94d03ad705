// Implicit-GEMM convolution on channels-last activations (gfx950 MFMA), shared definitions.
#pragma once
#include "common.hip.h"

namespace rssf {
namespace cv {

constexpr int MAX_TAPS = 19;   // fused {1x1 + 3x3 dil 6 + 3x3 dil 12} of MlpDWBN (ffn_block.py:226-228,250-257)

// One "tap" = one (dy, dx) displacement with its own [Cout][Cin] weight slab.  out(oy,ox) += W_t * in(oy*s + dy, ox*s + dx)
struct Taps {
  int n;
  int dy[MAX_TAPS];
  int dx[MAX_TAPS];
};

// GEMM-K MFMA (full-rate on gfx950): bf16 16x16x32, f32 16x16x4 (exact fp32 for the parity mode)
template <typename T> struct MmaK;
template <> struct MmaK<bf16_t> {
  static constexpr int KSTEP = 32, KPL = 8, BK = 32;       // BK: channels per LDS stage (64-byte rows)
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag load(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaK<float> {
  static constexpr int KSTEP = 4, KPL = 1, BK = 16;
  typedef float frag;
  static __device__ __forceinline__ frag load(const float* p) { return *p; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// Workgroups are dispatched round-robin over the 8 XCDs (observed: block b runs on XCD b % 8), each with a private 4 MiB
// L2.  Work items that share operands (the taps of one pixel range, the channel tiles of one pixel tile, neighbouring
// pixel tiles) are therefore numbered so that they land on ONE XCD back to back: logical = (b % 8) * per + b / 8, with
// per = ceil(total / 8) and a grid of 8 * per blocks (logical >= total: the block exits).  Speed only, never correctness.
#ifdef __HIPCC__
__device__ __forceinline__ int64_t xcd_logical(unsigned b, int per) { return (int64_t)(b & 7u) * per + (b >> 3); }
#endif
inline int xcd_per(int64_t total) { return (int)((total + 7) / 8); }

// halo-tiled 3x3 / stride-1 bf16 kernel (conv_halo.hip), dispatched from rssf_conv_gather
struct HaloArgs {
  const bf16_t* in;      // [B, H, W, Cin]
  const bf16_t* wpk;     // [9][CoutP][CinP]
  bf16_t* out;           // [B, H, W, Cout]
  const float* bias;
  float* stats;          // [RSSF_BN_SLOTS][2][Cout] or null
  float* stats_ws;       // deterministic mode: per-tile partials [tiles][2][Cout] or null (see ConvArgs in conv_fwd.hip)
  const bf16_t* addend;  // [B, H, W, Cout] added to the output (fused gradient accumulation) or null
  // BatchNorm-backward statistics of the layer whose OUTPUT gradient this (data-gradient) launch produces, or bn_sums == null:
  // sums[block % RSSF_BN_BWD_SLOTS][2][Cout] += { sum dz, sum dz * raw },  dz = out * act'(raw * scale + shift + res_pre)
  const bf16_t* bn_raw;  // [B, H, W, Cout] pre-normalisation output of that layer's convolution
  const bf16_t* bn_res;  // [B, H, W, Cout] residual added before the activation, or null
  const float* bn_ss;    // [2][Cout] scale, shift
  float* bn_sums;
  int bn_act;
  // PRE-activation input (forward launches, pre_ss != null): `in` is the RAW output of the producing convolution and the
  // operand actually convolved is act(in * scale + shift), formed while the halo tile is staged (zero outside the image, as
  // padding of the activation would be) - the producer's BatchNorm apply pass never runs and its activation is never stored
  // The producer's BatchNorm is FINALIZED here as well (no launch of its own): every block folds the statistics slots of the Cin
  // channels into scale / shift (LDS), block 0 publishes mean / invstd / scale / shift and updates the running statistics -
  // the arithmetic of bn_finalize_kernel (bn.hip).
  const float* pre_stats;   // [RSSF_BN_SLOTS][2][Cin] (training) or null (eval: running statistics)
  const float* pre_gamma; const float* pre_beta;
  float* pre_rmean; float* pre_rvar;      // running statistics (updated in training mode; the source in eval mode)
  float* pre_mi; float* pre_ss;           // OUT [2][Cin] each: mean / invstd, scale / shift; pre_ss != null selects the PRE kernel
  float pre_n, pre_momentum, pre_eps;
  int pre_training;
  int pre_act;
  int B, H, W, Cin, Cout, CinP, CoutP;
  int tiles_y, tiles_x, ntiles_n, xcd_per;
  int ntiles;            // pixel tiles of the problem
  int64_t total;         // blocks of the launch
  int dy[9], dx[9];
};

bool halo_eligible(int IH, int IW, int Cin, int OH, int OW, int mul, int div, int ntaps, const int* dy, const int* dx);
int launch_halo(HaloArgs a, hipStream_t st);
int launch_halo_group(HaloArgs* items, int n, hipStream_t st);      // n <= RSSF_GROUP_MAX problems as one grid
// the same convolution at 32 -> 32 channels as a row stream: weights in registers, no LDS staging (conv_rows32.hip); identity / ReLU only
struct PwPre;
bool rows32_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx);
int launch_rows32(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend, const void* bn_raw,
                  const void* bn_res, const float* bn_ss, float* bn_sums, int bn_act, const PwPre* pre, int B, int H, int W, bool mirror,
                  hipStream_t st);
// many-tap 128 -> 128 channel convolutions with the pixel operand in registers (conv_taps128.hip): MlpDWBN's fused 17-tap sum, forward
// and data gradient (bf16)
bool taps128_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps);
int launch_taps128(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend, const void* bn_raw,
                   const void* bn_res, const float* bn_ss, float* bn_sums, int bn_act, int B, int H, int W, int Cin, int Cout, int CinP,
                   int CoutP, int ntaps, const int* dy, const int* dx, hipStream_t st);
// point-wise 32 -> 128 / 128 -> 32 channel convolutions as a stream (conv_pw.hip): MlpDWBN's fc1 / fc2, forward and data gradient (bf16)
bool pw_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx);
// the producer's BatchNorm for a launch that applies it on load (rssf_conv_gather_preact): the arguments of rssf_bn_finalize
struct PwPre {
  const float* stats; const float* gamma; const float* beta; float* rmean; float* rvar; float* mi; float* ss;
  float n, momentum, eps; int training, act;
};
// the stem's first convolution forward (8 <- 3 padded channels -> 64, 3x3 / stride 2) with the im2col row as the K axis (conv_stem_fwd.hip)
bool stem_fwd_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx);
int launch_stem_fwd(const void* in, const void* wpk, void* out, float* stats, int B, int IH, int IW, int OH, int OW, int CinP, int CoutP, hipStream_t st);
// data gradient of a 3x3 / stride-2 / padding-1 convolution (narrow channel counts) by output parity, a stream over the output-gradient rows (conv_dgrad_s2.hip)
bool dgrad_s2_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx);
int launch_dgrad_s2(const void* dout, const void* wpk, void* dx, const void* addend, const void* bn_raw, const void* bn_res, const float* bn_ss,
                    float* bn_sums, int bn_act, int B, int IH, int IW, int Cin, int Cout, int CinP, int CoutP, hipStream_t st);
bool pw_preact_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx);
int launch_pw(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* bn_raw, const void* bn_res,
              const float* bn_ss, float* bn_sums, int bn_act, int B, int H, int W, int Cin, int Cout, int CinP, int CoutP, hipStream_t st,
              const PwPre* pre = nullptr, const void* addend = nullptr);
bool pw_addend_eligible(int Cin, int Cout);      // the stream kernel adds `addend` for this shape (64 -> 256: layer1's conv1 data gradients)
int launch_stats_fold(const float* ws, int64_t tiles, int C, float* stats, hipStream_t st);
int launch_wgrad_reduce(const rssf_wgrad_reduce_job& j, hipStream_t st);      // second stage of a split-K weight gradient (conv_wgrad.hip)
int64_t wgrad_planes_workspace_elems(int B, int H, int W, int Cin, int Cout, int ntaps);     // 0: conv_wgrad_planes.hip does not serve the shape
// arguments of rssf_bn_bwd_apply for a weight-gradient launch that performs the apply on the way (rssf_conv_wgrad_bnapply)
struct WgradBn {
  const void* dy; const void* raw; const float* ss; const float* mi; const float* sums; const void* res; void* draw; void* dres;
  float* dgamma; float* dbeta; double n; int act, training; float pscale;
};
// narrow point-wise weight gradients, one block per pixel range (conv_wgrad_pw.hip); wgrad_pw_ksplit() partial planes [Cout][Cin]
bool wgrad_pw_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx);
int wgrad_pw_ksplit(int B, int OH, int OW, int Cin, int Cout);
// 3x3 / stride-2 convolutions with few input channels (the stem: 8 <- 3 padded -> 64; the fuse layers' 32 -> 32 / 64 / 128): a block owns
// every (co, tap, ci) of a pixel range; apply + weight gradient in one pass (conv_wgrad_stem.hip)
bool wgrad_stem_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx);
int wgrad_stem_ksplit(int B, int OH, int OW, int Cin, int Cout);
int launch_wgrad_stem(const void* dout, const void* in, float* partial, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int ksplit,
                      const WgradBn* bn, bool write_draw, hipStream_t st);
bool wgrad_pw_preact_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx);
bool wgrad_pw_dgrad_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx);
int launch_wgrad_pw(const void* dout, const void* in, float* partial, float* dbias, int B, int OH, int OW, int Cin, int Cout, int ksplit,
                    const WgradBn* bn, hipStream_t st, const float* x_ss = nullptr, int x_act = 0, const float* w_dg = nullptr,
                    void* dx_dg = nullptr, float* st_sums = nullptr);

template <typename T> struct LdsPad;
template <> struct LdsPad<bf16_t> { static constexpr int X = 8; };   // +16 B per row: conflict-free ds_read_b128
template <> struct LdsPad<float> { static constexpr int X = 4; };

}  // namespace cv
}  // namespace rssf
