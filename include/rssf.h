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
typedef enum {
  RSSF_OK = 0,
  RSSF_ERR_BAD_ARG = -1,
  RSSF_ERR_UNSUPPORTED = -2,
  RSSF_ERR_LAUNCH = -3
} rssf_status;

const char* rssf_version(void);
const char* rssf_arch(void);        /* always "gfx950" */
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
/* weights: g_s = sigmoid(conv7x7(pooled_s; k_s)); omega = softmax_2(Wl [g0;g1] + bl).
 * k [2][2][7][7], wl [2][2], bl [2]; gsig [B][2][N] (saved for bwd), omega [B][2][N], logits [B][2][N] (optional). */
int rssf_gate_weights_fwd(const float* pooled, const float* k, const float* wl, const float* bl, float* gsig,
                          float* omega, float* logits, int B, int H, int W, void* stream);
/* backward of gate_weights: domega [B][2][N] -> dpooled [B][4][N]; dk, dwl, dbl accumulated. */
int rssf_gate_weights_bwd(const float* domega, const float* pooled, const float* gsig, const float* omega,
                          const float* k, const float* wl, float* dpooled, float* dk, float* dwl, float* dbl,
                          int B, int H, int W, void* stream);
/* backward of gate_pool + merge: dxhat[b,n,c] (+)= gate-path gradient (mean: /C, max: argmax-routed).
 * dxhat/dyhat hold the attention-path gradient w.r.t. LN1 outputs on entry and the total on exit. */
int rssf_gate_pool_bwd(const float* dpooled, const int32_t* argmax, void* dxhat, void* dyhat, int B, int N, int C,
                       int dtype, void* stream);

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
  float* domega;               /* [B][2][N], must be zeroed by the caller; accumulated with atomics */
  float* dwq; float* dbq; float* dwk; float* dbk; float* dwv; float* dbv; float* dwo; float* dbo;  /* += */
} rssf_winattn_bwd_params;
int rssf_winattn_bwd(const rssf_winattn_bwd_params* p, void* stream);

/* ---- test hooks -------------------------------------------------------------------------------------- */
/* D[16][16] = A[16][K] * B[16][K]^T through the library's MFMA tile helper (layout self-check). */
int rssf_debug_mma(const void* a, const void* b, float* d, int K, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSSF_H */
