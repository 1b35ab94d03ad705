// BatchNorm2d (train: batch statistics; eval: running statistics) fused with the activation and the residual
// adds that surround it on the RSSFormer path, on channels-last activations.  Reference: nn.BatchNorm2d /
// nn.SyncBatchNorm call sites in _hrnet_rssformer.py:216-287, 361-405, 512-546, hrnet_aux.py:45-49 and
// ffn_block.py:222-234 (momentum 0.1, eps 1e-5, biased variance for normalisation, unbiased for running_var).
//
// The per-channel sums come from the producing convolution's epilogue (conv_fwd.hip), so forward is
//   finalize (C threads)  ->  apply: y = act(raw*scale + shift + res_pre) + res_post      (one HBM pass)
// and backward is
//   reduce: s1 = sum dz, s2 = sum dz*raw   (dz = dy * act'(z))                            (one pass)
//   apply : draw = scale * (dz - s1/n - xhat * sum(dz*xhat)/n),  dres_pre = dz            (one pass)
// Cross-rank SyncBN = all-reduce of the tiny [2][C] buffers between the two launches (host side, RCCL).
#include "common.hip.h"
using namespace rssf;

// fixed-order second level of the deterministic statistics (conv_fwd.hip)
namespace rssf { namespace cv { int launch_stats_fold(const float* ws, int64_t tiles, int C, float* stats, hipStream_t st); } }

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

// act = activation (bits 0..1) | RSSF_ACT_POST_RELU: y = relu(act(z) + res_post) - the ReLU that closes a HighResolutionModule's
// transformer output (_hrnet_rssformer.py:435) rides in the last BatchNorm pass of MlpDWBN
__device__ __forceinline__ float act_fwd(float z, int act) {
  const int a = act & 3;
  return a == ACT_RELU ? fmaxf(z, 0.f) : a == ACT_GELU ? gelu_erf(z) : z;
}
__device__ __forceinline__ float act_bwd(float z, int act) {
  const int a = act & 3;
  return a == ACT_RELU ? (z > 0.f ? 1.f : 0.f) : a == ACT_GELU ? gelu_erf_grad(z) : 1.f;
}
__host__ __device__ __forceinline__ bool act_ok(int act) { return (act & ~RSSF_ACT_POST_RELU) >= 0 && (act & ~RSSF_ACT_POST_RELU) <= 2; }

// stats [2][C] = {sum, sumsq} over n samples  ->  mean/invstd, scale/shift; running stats updated in place.
__global__ void bn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean_invstd,
                                   float* __restrict__ scale_shift, int C, float n, float momentum, float eps, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < RSSF_BN_SLOTS; ++k) { s1 += stats[k * 2 * C + c]; s2 += stats[k * 2 * C + C + c]; }
    mean = s1 / n;
    var = fmaxf(s2 / n - mean * mean, 0.f);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float invstd = rsqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  mean_invstd[c] = mean; mean_invstd[C + c] = invstd;
  scale_shift[c] = sc; scale_shift[C + c] = beta[c] - mean * sc;
}

// elementwise over [rows][C].  Thread layout: blockDim = (cols, rpb) with cols = min(C/VEC, 256) vector columns, so a
// thread keeps ONE channel group for its whole grid-stride loop over rows: scale/shift sit in registers and there is
// no per-element index arithmetic; consecutive threads still touch consecutive 16-byte chunks (fully coalesced).
template <typename T, int VEC>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ raw, const float* __restrict__ ss, const T* __restrict__ res_pre,
                                                       const T* __restrict__ res_post, T* __restrict__ y, int64_t rows, int C, int act) {
  const int allcols = C / VEC;
  const int colbase = blockIdx.y * 256;
  const int cols = allcols - colbase < 256 ? allcols - colbase : 256;
  const int rpb = 256 / cols;
  const int col = threadIdx.x % cols, rlocal = threadIdx.x / cols;
  if (rlocal >= rpb) return;
  const int c0 = (colbase + col) * VEC;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sc[e] = ss[c0 + e]; sh[e] = ss[C + c0 + e]; }
  for (int64_t r = (int64_t)blockIdx.x * rpb + rlocal; r < rows; r += (int64_t)gridDim.x * rpb) {
    const int64_t off = r * C + c0;
    if constexpr (VEC > 1) {
      Vec<T> v, rp, rq, o;
      v.load(raw + off);
      if (res_pre) rp.load(res_pre + off);
      if (res_post) rq.load(res_post + off);
      float ov[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float z = fmaf(v.get(e), sc[e], sh[e]);               // explicit: conv_halo.hip's pre-activation staging rounds alike
        if (res_pre) z += rp.get(e);
        float t = act_fwd(z, act);
        if (res_post) t += rq.get(e);
        if (act & RSSF_ACT_POST_RELU) t = fmaxf(t, 0.f);
        ov[e] = t;
      }
      o.set_all(ov);
      o.store(y + off);
    } else {
      float z = ldf(raw + off) * sc[0] + sh[0];
      if (res_pre) z += ldf(res_pre + off);
      float t = act_fwd(z, act);
      if (res_post) t += ldf(res_post + off);
      if (act & RSSF_ACT_POST_RELU) t = fmaxf(t, 0.f);
      stf(y + off, t);
    }
  }
}

// bn_finalize + bn_apply in one launch (330 BatchNorm layers per step: one launch and ~4 us of latency less each).  Every
// block folds the statistics slots of ITS channels into scale/shift (cooperatively, through LDS, under the first row's
// loads); block row 0 also publishes mean/invstd, scale/shift and the running statistics for the backward pass.
// (a __device__ body with explicit block coordinates: bn_finapply_kernel runs it for one layer, bn_finapply_group_kernel for the
// layers of a lock-step group in ONE grid - rssf_bn_finalize_apply_group)
template <typename T, int VEC>
__device__ __forceinline__ void bn_finapply_block(const T* __restrict__ raw, const float* __restrict__ stats,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ running_mean, float* __restrict__ running_var,
                                                  float* __restrict__ mean_invstd, float* __restrict__ scale_shift,
                                                  const T* __restrict__ res_pre, const T* __restrict__ res_post, T* __restrict__ y,
                                                  int64_t rows, int C, int act, float n, float momentum, float eps, int training,
                                                  const unsigned bx, const unsigned by, const unsigned gx, float* lds) {
  const int allcols = C / VEC;
  const int colbase = by * 256;
  const int cols = allcols - colbase < 256 ? allcols - colbase : 256;
  const int nch = cols * VEC, ch0 = colbase * VEC;
  const int rpb = 256 / cols;
  const int col = threadIdx.x % cols, rlocal = threadIdx.x / cols;
  const bool active = rlocal < rpb;
  const int c0 = (colbase + col) * VEC;
  const int64_t stride = (int64_t)gx * rpb;
  int64_t r = (int64_t)bx * rpb + rlocal;
  Vec<T> v, rp, rq;
  bool have = active && r < rows;
  if constexpr (VEC > 1) {
    if (have) {
      v.load(raw + r * C + c0);
      if (res_pre) rp.load(res_pre + r * C + c0);
      if (res_post) rq.load(res_post + r * C + c0);
    }
  }
  float* scsh = lds;
  for (int i = threadIdx.x; i < nch; i += blockDim.x) {
    const int c = ch0 + i;
    float mean, var;
    if (training) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < RSSF_BN_SLOTS; ++k) { s1 += stats[(size_t)k * 2 * C + c]; s2 += stats[(size_t)k * 2 * C + C + c]; }
      mean = s1 / n;
      var = fmaxf(s2 / n - mean * mean, 0.f);
    } else {
      mean = running_mean[c];
      var = running_var[c];
    }
    const float invstd = rsqrtf(var + eps);
    const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
    scsh[i] = sc; scsh[nch + i] = sh;
    if (bx == 0) {
      mean_invstd[c] = mean; mean_invstd[C + c] = invstd;
      scale_shift[c] = sc; scale_shift[C + c] = sh;
      if (training && running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
      }
    }
  }
  __syncthreads();
  if (!active) return;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sc[e] = scsh[col * VEC + e]; sh[e] = scsh[nch + col * VEC + e]; }
  if constexpr (VEC > 1) {
    while (have) {
      const int64_t off = r * C + c0, rn = r + stride;
      const bool have_next = rn < rows;
      Vec<T> nv, np, nq;
      if (have_next) {
        nv.load(raw + rn * C + c0);
        if (res_pre) np.load(res_pre + rn * C + c0);
        if (res_post) nq.load(res_post + rn * C + c0);
      }
      float ov[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float z = fmaf(v.get(e), sc[e], sh[e]);               // explicit: conv_halo.hip's pre-activation staging rounds alike
        if (res_pre) z += rp.get(e);
        float t = act_fwd(z, act);
        if (res_post) t += rq.get(e);
        if (act & RSSF_ACT_POST_RELU) t = fmaxf(t, 0.f);
        ov[e] = t;
      }
      Vec<T> o;
      o.set_all(ov);
      o.store(y + off);
      v = nv; rp = np; rq = nq; r = rn; have = have_next;
    }
  } else {
    for (; r < rows; r += stride) {
      const int64_t off = r * C + c0;
      float z = ldf(raw + off) * sc[0] + sh[0];
      if (res_pre) z += ldf(res_pre + off);
      float t = act_fwd(z, act);
      if (res_post) t += ldf(res_post + off);
      if (act & RSSF_ACT_POST_RELU) t = fmaxf(t, 0.f);
      stf(y + off, t);
    }
  }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256) bn_finapply_kernel(const T* __restrict__ raw, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ mean_invstd, float* __restrict__ scale_shift,
                                                          const T* __restrict__ res_pre, const T* __restrict__ res_post, T* __restrict__ y,
                                                          int64_t rows, int C, int act, float n, float momentum, float eps, int training) {
  extern __shared__ float lds[];                          // scale/shift [2][nch]
  bn_finapply_block<T, VEC>(raw, stats, gamma, beta, running_mean, running_var, mean_invstd, scale_shift, res_pre, res_post, y, rows, C, act, n,
                            momentum, eps, training, blockIdx.x, blockIdx.y, gridDim.x, lds);
}
// the layers of a lock-step group in one grid: item i owns blocks [start[i], start[i+1]) laid out as its own (gx, gy) grid
struct FinApplyItem {
  const void* raw; const float* stats; const float* gamma; const float* beta; float* rm; float* rv; float* mi; float* ss;
  const void* rp; const void* rq; void* y;
  int64_t rows;
  int C, act, training, gx;
  float n, momentum, eps;
};
struct FinApplyGroup { FinApplyItem it[RSSF_GROUP_MAX]; int start[RSSF_GROUP_MAX + 1]; int n; };
template <typename T, int VEC>
__global__ void __launch_bounds__(256) bn_finapply_group_kernel(FinApplyGroup g) {
  extern __shared__ float lds[];
  int i = 0;
#pragma unroll
  for (int k = 1; k < RSSF_GROUP_MAX; ++k)
    if (k < g.n && blockIdx.x >= (unsigned)g.start[k]) i = k;
  const FinApplyItem& a = g.it[i];
  const unsigned r = blockIdx.x - (unsigned)g.start[i], gx = (unsigned)a.gx;
  bn_finapply_block<T, VEC>((const T*)a.raw, a.stats, a.gamma, a.beta, a.rm, a.rv, a.mi, a.ss, (const T*)a.rp, (const T*)a.rq, (T*)a.y, a.rows, a.C, a.act,
                            a.n, a.momentum, a.eps, a.training, r % gx, r / gx, gx, lds);
}

// s[0][c] += sum dz ; s[1][c] += sum dz*raw ; thread owns a fixed vector column and strides over rows.
// DET (deterministic mode): no shuffle/atomic folding - every thread parks its partials in LDS ([row group][2][channels of
// this column block]), they are summed in row-group order and the block's totals go to det_ws[block][2][C] with plain stores;
// cv::stats_fold_kernel then adds the blocks in order into slot 0 of `sums`.
template <typename T, int VEC, bool DET>
__device__ __forceinline__ void bn_bwd_reduce_block(const T* __restrict__ dy, const T* __restrict__ raw, const float* __restrict__ ss,
                                                    const T* __restrict__ res_pre, float* __restrict__ sums, int64_t rows, int C,
                                                    int act, float* __restrict__ det_ws, const unsigned bx, const unsigned by,
                                                    const unsigned gx, float* sacc, const T* __restrict__ res_post = nullptr) {
  if constexpr (!DET) {
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sacc[i] = 0.f;
    __syncthreads();
  }
  const int allcols = C / VEC;
  const int colbase = by * 256;                           // column blocks of <= 256 vector columns
  const int cols = allcols - colbase < 256 ? allcols - colbase : 256;
  const int rpb = blockDim.x / cols;                      // rows handled per block per pass (>= 1)
  const int col = threadIdx.x % cols, rlocal = threadIdx.x / cols;
  float a1[VEC], a2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
  const bool active = rlocal < rpb;
  const int c0 = (colbase + col) * VEC;
  if (active) {
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = ss[c0 + e]; sh[e] = ss[C + c0 + e]; }
    const int64_t stride = (int64_t)gx * rpb;
    const bool post = (act & RSSF_ACT_POST_RELU) != 0;         // y = relu(act(z) + res_post): the gradient passes where y > 0
    auto body = [&](const Vec<T>& vd, const Vec<T>& vr, const Vec<T>& vp, const Vec<T>& vq) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float x = vr.get(e);
        float z = fmaf(x, sc[e], sh[e]);
        if (res_pre) z += vp.get(e);
        float g = vd.get(e);
        if (post && !(act_fwd(z, act) + (res_post ? vq.get(e) : 0.f) > 0.f)) g = 0.f;
        const float dz = g * act_bwd(z, act);
        a1[e] += dz; a2[e] += dz * x;
      }
    };
    int64_t r = (int64_t)bx * rpb + rlocal;
    if constexpr (VEC > 1) {
      for (; r + stride < rows; r += 2 * stride) {          // two rows in flight per thread
        const int64_t o0 = r * C + c0, o1 = (r + stride) * C + c0;
        Vec<T> d0, r0, p0, q0, d1, r1, p1, q1;
        d0.load(dy + o0); r0.load(raw + o0); d1.load(dy + o1); r1.load(raw + o1);
        if (res_pre) { p0.load(res_pre + o0); p1.load(res_pre + o1); }
        if (res_post) { q0.load(res_post + o0); q1.load(res_post + o1); }
        body(d0, r0, p0, q0); body(d1, r1, p1, q1);
      }
      for (; r < rows; r += stride) {
        const int64_t o0 = r * C + c0;
        Vec<T> d0, r0, p0, q0;
        d0.load(dy + o0); r0.load(raw + o0);
        if (res_pre) p0.load(res_pre + o0);
        if (res_post) q0.load(res_post + o0);
        body(d0, r0, p0, q0);
      }
    } else {
      for (; r < rows; r += stride) {
        const int64_t off = r * C + c0;
        const float x = ldf(raw + off);
        float z = x * sc[0] + sh[0];
        if (res_pre) z += ldf(res_pre + off);
        float g = ldf(dy + off);
        if (post && !(act_fwd(z, act) + (res_post ? ldf(res_post + off) : 0.f) > 0.f)) g = 0.f;
        const float dz = g * act_bwd(z, act);
        a1[0] += dz; a2[0] += dz * x;
      }
    }
  }
  if constexpr (DET) {
    const int nch = cols * VEC;
    if (active) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { sacc[(rlocal * 2) * nch + col * VEC + e] = a1[e]; sacc[(rlocal * 2 + 1) * nch + col * VEC + e] = a2[e]; }
    }
    __syncthreads();
    float* part = det_ws + (size_t)bx * 2 * C;
    for (int i = threadIdx.x; i < 2 * nch; i += blockDim.x) {
      const int half = i / nch, ch = i % nch;
      float t = 0.f;
      for (int g = 0; g < rpb; ++g) t += sacc[(g * 2 + half) * nch + ch];
      part[half * C + colbase * VEC + ch] = t;
    }
    return;
  }
  // lanes that share a column inside a wave (cols divides 64) are folded with shuffles before touching LDS
  const bool fold = (64 % cols) == 0;
  if (fold) {
    for (int o = 32; o >= cols; o >>= 1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { a1[e] += __shfl_xor(a1[e], o, 64); a2[e] += __shfl_xor(a2[e], o, 64); }
    }
  }
  if (active && (!fold || (threadIdx.x & 63) < cols)) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) { atomicAdd(&sacc[c0 + e], a1[e]); atomicAdd(&sacc[C + c0 + e], a2[e]); }
  }
  __syncthreads();
  float* slot = sums + (size_t)(bx % RSSF_BN_BWD_SLOTS) * 2 * C;
  for (int i = threadIdx.x; i < cols * VEC; i += blockDim.x) {
    const int c = colbase * VEC + i;
    atomicAdd(&slot[c], sacc[c]);
    atomicAdd(&slot[C + c], sacc[C + c]);
  }
}

template <typename T, int VEC, bool DET>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ raw, const float* __restrict__ ss,
                                                            const T* __restrict__ res_pre, float* __restrict__ sums, int64_t rows, int C,
                                                            int act, float* __restrict__ det_ws, const T* __restrict__ res_post) {
  extern __shared__ float sacc[];                         // [2][C]  (DET: [256 / cols][2][cols * VEC])
  bn_bwd_reduce_block<T, VEC, DET>(dy, raw, ss, res_pre, sums, rows, C, act, det_ws, blockIdx.x, blockIdx.y, gridDim.x, sacc, res_post);
}
struct ReduceItem { const void* dy; const void* raw; const float* ss; const void* rp; float* sums; int64_t rows; int C, act, gx; };
struct ReduceGroup { ReduceItem it[RSSF_GROUP_MAX]; int start[RSSF_GROUP_MAX + 1]; int n; };
template <typename T, int VEC>
__global__ void __launch_bounds__(256) bn_bwd_reduce_group_kernel(ReduceGroup g) {
  extern __shared__ float sacc[];
  int i = 0;
#pragma unroll
  for (int k = 1; k < RSSF_GROUP_MAX; ++k)
    if (k < g.n && blockIdx.x >= (unsigned)g.start[k]) i = k;
  const ReduceItem& a = g.it[i];
  const unsigned r = blockIdx.x - (unsigned)g.start[i], gx = (unsigned)a.gx;
  bn_bwd_reduce_block<T, VEC, false>((const T*)a.dy, (const T*)a.raw, a.ss, (const T*)a.rp, a.sums, a.rows, a.C, a.act, nullptr, r % gx, r / gx, gx, sacc);
}

// draw = scale*(dz - k1 - xhat*k2) (training) or scale*dz (eval);  dres (optional) = dz.   Same thread layout as
// bn_apply_kernel: per-channel constants (scale, shift, mean, invstd, k1, k2) are computed once per thread.
template <typename T, int VEC>
__device__ __forceinline__ void bn_bwd_apply_block(const T* __restrict__ dy, const T* __restrict__ raw, const float* __restrict__ ss,
                                                           const float* __restrict__ mi, const float* __restrict__ sums,
                                                           const T* __restrict__ res_pre, T* __restrict__ draw, T* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C,
                                                           int act, float n, int training, float pscale, const T* __restrict__ res_post,
                                                           T* __restrict__ dpost, const unsigned bx, const unsigned by, const unsigned gx, float* tot) {
  const int allcols = C / VEC;
  const int colbase = by * 256;
  const int cols = allcols - colbase < 256 ? allcols - colbase : 256;
  const int rpb = 256 / cols;
  const int col = threadIdx.x % cols, rlocal = threadIdx.x / cols;
  const bool active = rlocal < rpb;
  const int c0 = (colbase + col) * VEC;
  const int64_t stride = (int64_t)gx * rpb;
  int64_t r = (int64_t)bx * rpb + rlocal;
  // the first row's operands and the per-channel constants are requested before the slot totals are folded, so that
  // the fold's load -> LDS -> barrier chain overlaps with them (each thread only sees a handful of rows)
  Vec<T> vd, vr, vp, vq;
  const bool post = (act & RSSF_ACT_POST_RELU) != 0;        // y = relu(act(z) + res_post): dpost = dy where y > 0, and dz from that
  bool have = active && r < rows;
  if constexpr (VEC > 1) {
    if (have) {
      vd.load(dy + r * C + c0); vr.load(raw + r * C + c0);
      if (res_pre) vp.load(res_pre + r * C + c0);
      if (res_post) vq.load(res_post + r * C + c0);
    }
  }
  float sc[VEC], sh[VEC], mean[VEC], istd[VEC], k1[VEC], k2[VEC], cb[VEC], cc[VEC];
  if (active) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = ss[c0 + e]; sh[e] = ss[C + c0 + e]; mean[e] = mi[c0 + e]; istd[e] = mi[C + c0 + e]; }
  }
  for (int i = threadIdx.x; i < 2 * cols * VEC; i += blockDim.x) {
    const int half = i / (cols * VEC), c = colbase * VEC + i % (cols * VEC);
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < RSSF_BN_BWD_SLOTS; ++k) t += sums[(size_t)k * 2 * C + half * C + c];
    tot[i] = t;
  }
  __syncthreads();
  if (!active) return;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int c = c0 + e;
    const float s1 = tot[col * VEC + e], s2 = tot[cols * VEC + col * VEC + e];
    float dot;
    bn_bwd_constants(sc[e], mean[e], istd[e], s1, s2, n, dot, cb[e], cc[e]);
    k1[e] = s1 / n;
    k2[e] = dot / n;
    // one writer per channel.  pscale = 1/world under SyncBN: `sums` are then the GLOBAL totals, while DDP averages the LOCAL
    // parameter gradients (torch SyncBatchNorm takes grad_weight/grad_bias from the local sums) - global/world == that mean
    if (dgamma && bx == 0 && rlocal == 0) { dgamma[c] += dot * pscale; dbeta[c] += s1 * pscale; }
    // draw = sc*(dz - k1 - (x - mean)*istd*k2) = sc*dz + cb*x + cc : two FMAs per element instead of six operations (the pass
    // runs 8 waves per SIMD at 22 % VALU-active each: instruction issue, not the fabric, was its limit)
  }
  if constexpr (VEC > 1) {
    while (have) {
      const int64_t off = r * C + c0, rn = r + stride;
      const bool have_next = rn < rows;
      Vec<T> nd, nr, np, nq;
      if (have_next) {
        nd.load(dy + rn * C + c0); nr.load(raw + rn * C + c0);
        if (res_pre) np.load(res_pre + rn * C + c0);
        if (res_post) nq.load(res_post + rn * C + c0);
      }
      float o1[VEC], o2[VEC], o3[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float x = vr.get(e);
        float z = fmaf(x, sc[e], sh[e]);                       // explicit: the fused forms (conv_wgrad.hip, conv_halo.hip) round alike
        if (res_pre) z += vp.get(e);
        float g = vd.get(e);
        if (post && !(act_fwd(z, act) + (res_post ? vq.get(e) : 0.f) > 0.f)) g = 0.f;
        o3[e] = g;
        const float dz = g * act_bwd(z, act);
        o2[e] = dz;
        o1[e] = training ? fmaf(sc[e], dz, fmaf(cb[e], x, cc[e])) : sc[e] * dz;
      }
      Vec<T> w1, w2;
      w1.set_all(o1); w2.set_all(o2);
      w1.store(draw + off);
      if (dres) w2.store(dres + off);
      if (dpost) { Vec<T> w3; w3.set_all(o3); w3.store(dpost + off); }
      vd = nd; vr = nr; vp = np; vq = nq; r = rn; have = have_next;
    }
  } else {
    for (; r < rows; r += stride) {
      const int64_t off = r * C + c0;
      const float x = ldf(raw + off);
      float z = fmaf(x, sc[0], sh[0]);
      if (res_pre) z += ldf(res_pre + off);
      float g = ldf(dy + off);
      if (post && !(act_fwd(z, act) + (res_post ? ldf(res_post + off) : 0.f) > 0.f)) g = 0.f;
      const float dz = g * act_bwd(z, act);
      stf(draw + off, training ? sc[0] * (dz - k1[0] - (x - mean[0]) * istd[0] * k2[0]) : sc[0] * dz);
      if (dres) stf(dres + off, dz);
      if (dpost) stf(dpost + off, g);
    }
  }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ raw, const float* __restrict__ ss,
                                                           const float* __restrict__ mi, const float* __restrict__ sums,
                                                           const T* __restrict__ res_pre, T* __restrict__ draw, T* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C,
                                                           int act, float n, int training, float pscale, const T* __restrict__ res_post,
                                                           T* __restrict__ dpost) {
  extern __shared__ float tot[];                           // [2][cols*VEC] slot totals of this block's channels
  bn_bwd_apply_block<T, VEC>(dy, raw, ss, mi, sums, res_pre, draw, dres, dgamma, dbeta, rows, C, act, n, training, pscale, res_post, dpost,
                             blockIdx.x, blockIdx.y, gridDim.x, tot);
}
// the layers of a lock-step group in one grid (see bn_finapply_group_kernel): the 1x1 / strided fuse convolutions of one depth
struct BwdApplyItem {
  const void* dy; const void* raw; const float* ss; const float* mi; const float* sums; const void* rp; void* draw; void* dres;
  float* dgamma; float* dbeta;
  int64_t rows;
  int C, act, training, gx;
  float n, pscale;
};
struct BwdApplyGroup { BwdApplyItem it[RSSF_GROUP_MAX]; int start[RSSF_GROUP_MAX + 1]; int n; };
template <typename T, int VEC>
__global__ void __launch_bounds__(256) bn_bwd_apply_group_kernel(BwdApplyGroup g) {
  extern __shared__ float tot[];
  int i = 0;
#pragma unroll
  for (int k = 1; k < RSSF_GROUP_MAX; ++k)
    if (k < g.n && blockIdx.x >= (unsigned)g.start[k]) i = k;
  const BwdApplyItem& a = g.it[i];
  const unsigned r = blockIdx.x - (unsigned)g.start[i], gx = (unsigned)a.gx;
  bn_bwd_apply_block<T, VEC>((const T*)a.dy, (const T*)a.raw, a.ss, a.mi, a.sums, (const T*)a.rp, (T*)a.draw, (T*)a.dres, a.dgamma, a.dbeta, a.rows,
                             a.C, a.act, a.n, a.training, a.pscale, nullptr, nullptr, r % gx, r / gx, gx, tot);
}

// (row blocks, column blocks of <= 256 vector columns) for the fixed-column thread layout
dim3 grid2d(int64_t rows, int C, int vec) {
  const int cols = C / vec;
  const int cblocks = (cols + 255) / 256;
  const int rpb = 256 / (cols < 256 ? cols : 256);
  int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > 2048) blocks = 2048;
  return dim3((unsigned)blocks, (unsigned)cblocks);
}

template <typename T>
int apply_launch(const void* raw, const float* ss, const void* rp, const void* rq, void* y, int64_t rows, int C, int act, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  const dim3 grid = grid2d(rows, C, (C % V == 0) ? V : 1);
  if (C % V == 0) bn_apply_kernel<T, V><<<grid, 256, 0, st>>>((const T*)raw, ss, (const T*)rp, (const T*)rq, (T*)y, rows, C, act);
  else bn_apply_kernel<T, 1><<<grid, 256, 0, st>>>((const T*)raw, ss, (const T*)rp, (const T*)rq, (T*)y, rows, C, act);
  return check_launch("bn_apply");
}

template <typename T>
int finapply_launch(const void* raw, const float* stats, const float* gamma, const float* beta, float* rm, float* rv, float* mi, float* ss,
                    const void* rp, const void* rq, void* y, int64_t rows, int C, int act, float n, float momentum, float eps, int training,
                    hipStream_t st) {
  constexpr int V = Vec<T>::N;
  const int vec = (C % V == 0) ? V : 1;
  const dim3 grid = grid2d(rows, C, vec);
  const int nch = (C / vec < 256 ? C / vec : 256) * vec;
  const size_t sh = 2 * nch * sizeof(float);
  if (vec == V)
    bn_finapply_kernel<T, V><<<grid, 256, sh, st>>>((const T*)raw, stats, gamma, beta, rm, rv, mi, ss, (const T*)rp, (const T*)rq, (T*)y, rows,
                                                    C, act, n, momentum, eps, training);
  else
    bn_finapply_kernel<T, 1><<<grid, 256, sh, st>>>((const T*)raw, stats, gamma, beta, rm, rv, mi, ss, (const T*)rp, (const T*)rq, (T*)y, rows,
                                                    C, act, n, momentum, eps, training);
  return check_launch("bn_finalize_apply");
}

constexpr int REDUCE_MAX_BLOCKS = 512;

template <typename T>
int reduce_launch(const void* dy, const void* raw, const float* ss, const void* rp, float* sums, int64_t rows, int C, int act, float* det_ws,
                  hipStream_t st, const void* rq = nullptr) {
  constexpr int V = Vec<T>::N;
  const size_t sh = 2 * C * sizeof(float);
  const int vec = (C % V == 0) ? V : 1;
  const int cols = C / vec;
  const int cblocks = (cols + 255) / 256;
  const int rpb = 256 / (cols < 256 ? cols : 256);
  int64_t blocks = (rows + 4 * rpb - 1) / (4 * rpb);      // >= 4 rows per thread; measured best cap on MI355X: 512
  if (blocks > REDUCE_MAX_BLOCKS) blocks = REDUCE_MAX_BLOCKS;        // every block ends with 2C global atomics, spread over RSSF_BN_BWD_SLOTS copies
  dim3 grid((unsigned)blocks, (unsigned)cblocks);
  if (det_ws) {
    const size_t shd = (size_t)rpb * 2 * (cols < 256 ? cols : 256) * vec * sizeof(float);
    if (vec == V)
      bn_bwd_reduce_kernel<T, V, true><<<grid, 256, shd, st>>>((const T*)dy, (const T*)raw, ss, (const T*)rp, sums, rows, C, act, det_ws, (const T*)rq);
    else
      bn_bwd_reduce_kernel<T, 1, true><<<grid, 256, shd, st>>>((const T*)dy, (const T*)raw, ss, (const T*)rp, sums, rows, C, act, det_ws, (const T*)rq);
    const int rc = check_launch("bn_bwd_reduce(det)");
    return rc ? rc : rssf::cv::launch_stats_fold(det_ws, blocks, C, sums, st);
  }
  if (vec == V)
    bn_bwd_reduce_kernel<T, V, false><<<grid, 256, sh, st>>>((const T*)dy, (const T*)raw, ss, (const T*)rp, sums, rows, C, act, nullptr, (const T*)rq);
  else
    bn_bwd_reduce_kernel<T, 1, false><<<grid, 256, sh, st>>>((const T*)dy, (const T*)raw, ss, (const T*)rp, sums, rows, C, act, nullptr, (const T*)rq);
  return check_launch("bn_bwd_reduce");
}


template <typename T>
int bwd_apply_launch(const void* dy, const void* raw, const float* ss, const float* mi, const float* sums, const void* rp, void* draw,
                     void* dres, float* dgamma, float* dbeta, int64_t rows, int C, int act, float n, int training, float pscale,
                     hipStream_t st, const void* rq = nullptr, void* dpost = nullptr) {
  constexpr int V = Vec<T>::N;
  const dim3 grid = grid2d(rows, C, (C % V == 0) ? V : 1);
  const size_t sh = 2 * sizeof(float) * (C < 256 * V ? C : 256 * V);
  if (C % V == 0)
    bn_bwd_apply_kernel<T, V><<<grid, 256, sh, st>>>((const T*)dy, (const T*)raw, ss, mi, sums, (const T*)rp, (T*)draw, (T*)dres, dgamma,
                                                    dbeta, rows, C, act, n, training, pscale, (const T*)rq, (T*)dpost);
  else
    bn_bwd_apply_kernel<T, 1><<<grid, 256, sh, st>>>((const T*)dy, (const T*)raw, ss, mi, sums, (const T*)rp, (T*)draw, (T*)dres, dgamma,
                                                    dbeta, rows, C, act, n, training, pscale, (const T*)rq, (T*)dpost);
  return check_launch("bn_bwd_apply");
}
}  // namespace

extern "C" int rssf_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float* mean_invstd, float* scale_shift, int C, double n, float momentum, float eps, int training,
                                void* stream) {
  RSSF_REQUIRE(gamma && beta && mean_invstd && scale_shift && C > 0, "bn_finalize: bad arguments");
  RSSF_REQUIRE(training ? (stats != nullptr && n >= 1) : (running_mean && running_var), "bn_finalize: missing statistics");
  bn_finalize_kernel<<<(C + 127) / 128, 128, 0, (hipStream_t)stream>>>(stats, gamma, beta, running_mean, running_var, mean_invstd,
                                                                      scale_shift, C, (float)n, momentum, eps, training);
  return check_launch("bn_finalize");
}

extern "C" int rssf_bn_apply(const void* raw, const float* scale_shift, const void* res_pre, const void* res_post, void* y, int64_t rows,
                             int C, int act, int dtype, void* stream) {
  RSSF_REQUIRE(raw && scale_shift && y && rows > 0 && C > 0 && act_ok(act), "bn_apply: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return apply_launch<float>(raw, scale_shift, res_pre, res_post, y, rows, C, act, st);
  if (dtype == RSSF_BF16) return apply_launch<bf16_t>(raw, scale_shift, res_pre, res_post, y, rows, C, act, st);
  set_error("bn_apply: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_bn_finalize_apply(const void* raw, const float* stats, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float* mean_invstd, float* scale_shift, const void* res_pre,
                                      const void* res_post, void* y, int64_t rows, int C, int act, double n, float momentum, float eps,
                                      int training, int dtype, void* stream) {
  RSSF_REQUIRE(raw && gamma && beta && mean_invstd && scale_shift && y && rows > 0 && C > 0 && act_ok(act),
               "bn_finalize_apply: bad arguments");
  RSSF_REQUIRE(training ? (stats != nullptr && n >= 1) : (running_mean && running_var), "bn_finalize_apply: missing statistics");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32)
    return finapply_launch<float>(raw, stats, gamma, beta, running_mean, running_var, mean_invstd, scale_shift, res_pre, res_post, y, rows,
                                  C, act, (float)n, momentum, eps, training, st);
  if (dtype == RSSF_BF16)
    return finapply_launch<bf16_t>(raw, stats, gamma, beta, running_mean, running_var, mean_invstd, scale_shift, res_pre, res_post, y, rows,
                                   C, act, (float)n, momentum, eps, training, st);
  set_error("bn_finalize_apply: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

// ---- finalize + apply that ALSO writes the activation TRANSPOSED: y_planes[c][b][H + 2 pad][W + 2 pad] (pixels contiguous, the
// border is never written: the caller zeroed the buffer once) - the input operand of rssf_conv_wgrad_planes (conv_wgrad_planes.hip),
// produced in the pass that has every value in registers anyway instead of a transposing pass of its own (read 67 MB, write 94 MB
// per MlpDWBN block).  bf16, C = 128 (the hidden width of MlpDWBN at Base), W a multiple of 64: a workgroup walks tiles of 64
// pixels of one image row x 128 channels; rows go to y as they come (16-byte stores), the tile turns around through LDS and leaves
// as 64-byte runs of one channel.
namespace {
constexpr int PL_C = 128, PL_PX = 64, PL_PITCH = PL_C + 8;
__global__ void __launch_bounds__(256) bn_finapply_planes_kernel(const bf16_t* __restrict__ raw, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 float* __restrict__ mean_invstd, float* __restrict__ scale_shift,
                                                                 bf16_t* __restrict__ y, bf16_t* __restrict__ planes, int B, int H, int W, int pad,
                                                                 int act, float n, float momentum, float eps, int training) {
  constexpr int C = PL_C;
  __shared__ float scsh[2 * C];
  __shared__ __attribute__((aligned(16))) bf16_t tile[PL_PX * PL_PITCH];
  const int tid = threadIdx.x;
  if (tid < C) {                       // the arithmetic of bn_finapply_block
    const int c = tid;
    float mean, var;
    if (training) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < RSSF_BN_SLOTS; ++k) { s1 += stats[(size_t)k * 2 * C + c]; s2 += stats[(size_t)k * 2 * C + C + c]; }
      mean = s1 / n;
      var = fmaxf(s2 / n - mean * mean, 0.f);
    } else {
      mean = running_mean[c];
      var = running_var[c];
    }
    const float invstd = rsqrtf(var + eps);
    const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
    scsh[c] = sc; scsh[C + c] = sh;
    if (blockIdx.x == 0) {
      mean_invstd[c] = mean; mean_invstd[C + c] = invstd;
      scale_shift[c] = sc; scale_shift[C + c] = sh;
      if (training && running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
      }
    }
  }
  __syncthreads();
  const int chunk = tid & 15, prow = tid >> 4;          // apply phase: 16-byte channel chunk, pixel row within a pass of 16
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = scsh[chunk * 8 + e]; sh[e] = scsh[C + chunk * 8 + e]; }
  const int tpr = W / PL_PX, ntiles = B * H * tpr;
  const int HP = H + 2 * pad, WP = W + 2 * pad;
  const int tch = tid & 127, thalf = tid >> 7;          // transposed phase: channel, half of the tile's pixels
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int row = t / tpr, x0 = (t - row * tpr) * PL_PX;
    const int b = row / H, yy = row - b * H;
    const size_t pix0 = (size_t)row * W + x0;
    Vec<bf16_t> v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p].load(raw + (pix0 + p * 16 + prow) * C + chunk * 8);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = act_fwd(fmaf(v[p].get(e), sc[e], sh[e]), act);
      Vec<bf16_t> w;
      w.set_all(o);
      w.store(y + (pix0 + p * 16 + prow) * C + chunk * 8);
      w.store(tile + (p * 16 + prow) * PL_PITCH + chunk * 8);
    }
    __syncthreads();
    // channel tch, pixels 32 thalf .. + 31 of the tile: 64 contiguous bytes of its plane row
    uint32_t pk[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t lo = tile[(thalf * 32 + 2 * k) * PL_PITCH + tch].v, hi = tile[(thalf * 32 + 2 * k + 1) * PL_PITCH + tch].v;
      pk[k] = lo | (hi << 16);
    }
    bf16_t* dst = planes + ((size_t)tch * B + b) * HP * WP + (size_t)(yy + pad) * WP + pad + x0 + thalf * 32;
#pragma unroll
    for (int k = 0; k < 8; ++k) {              // 8-byte stores: the run starts 2 * pad bytes into the (16-byte aligned) plane row
      typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
      *reinterpret_cast<u32x2_t*>(dst + 4 * k) = u32x2_t{pk[2 * k], pk[2 * k + 1]};
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int rssf_bn_finalize_apply_planes_supported(int B, int H, int W, int C, int pad, int dtype) {
  return dtype == RSSF_BF16 && C == PL_C && B > 0 && H > 0 && W > 0 && (W % PL_PX) == 0 && pad >= 0 && (pad % 4) == 0 && ((W + 2 * pad) % 4) == 0 &&      /* pad % 4: the 8-byte stores start pad elements into a 16-byte-aligned plane row */
                 (int64_t)PL_C * B * (H + 2 * pad) * (W + 2 * pad) < ((int64_t)1 << 30)
             ? 1 : 0;
}

extern "C" int rssf_bn_finalize_apply_planes(const void* raw, const float* stats, const float* gamma, const float* beta, float* running_mean,
                                             float* running_var, float* mean_invstd, float* scale_shift, void* y, void* y_planes, int B,
                                             int H, int W, int C, int pad, int act, double n, float momentum, float eps, int training,
                                             int dtype, void* stream) {
  RSSF_REQUIRE(raw && gamma && beta && mean_invstd && scale_shift && y && y_planes && act >= 0 && act <= 2, "bn_finalize_apply_planes: bad arguments");
  RSSF_REQUIRE(training ? (stats != nullptr && n >= 1) : (running_mean && running_var), "bn_finalize_apply_planes: missing statistics");
  RSSF_REQUIRE(rssf_bn_finalize_apply_planes_supported(B, H, W, C, pad, dtype), "bn_finalize_apply_planes: unsupported shape (ask _supported)");
  const int ntiles = B * H * (W / PL_PX);
  const int blocks = ntiles < 2048 ? ntiles : 2048;
  bn_finapply_planes_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)raw, stats, gamma, beta, running_mean, running_var, mean_invstd,
                                                                     scale_shift, (bf16_t*)y, (bf16_t*)y_planes, B, H, W, pad, act, (float)n,
                                                                     momentum, eps, training);
  return check_launch("bn_finalize_apply_planes");
}

extern "C" int64_t rssf_bn_bwd_reduce_workspace_elems(int64_t rows, int C) { return (int64_t)REDUCE_MAX_BLOCKS * 2 * C; }

extern "C" int rssf_bn_bwd_reduce(const void* dy, const void* raw, const float* scale_shift, const void* res_pre, float* sums,
                                  int64_t rows, int C, int act, float* det_ws, int dtype, void* stream) {
  RSSF_REQUIRE(dy && raw && scale_shift && sums && rows > 0 && C > 0, "bn_bwd_reduce: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return reduce_launch<float>(dy, raw, scale_shift, res_pre, sums, rows, C, act, det_ws, st);
  if (dtype == RSSF_BF16) return reduce_launch<bf16_t>(dy, raw, scale_shift, res_pre, sums, rows, C, act, det_ws, st);
  set_error("bn_bwd_reduce: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_bn_bwd_reduce_post(const void* dy, const void* raw, const float* scale_shift, const void* res_pre, const void* res_post,
                                       float* sums, int64_t rows, int C, int act, float* det_ws, int dtype, void* stream) {
  RSSF_REQUIRE(dy && raw && scale_shift && sums && rows > 0 && C > 0 && act_ok(act), "bn_bwd_reduce_post: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return reduce_launch<float>(dy, raw, scale_shift, res_pre, sums, rows, C, act, det_ws, st, res_post);
  if (dtype == RSSF_BF16) return reduce_launch<bf16_t>(dy, raw, scale_shift, res_pre, sums, rows, C, act, det_ws, st, res_post);
  set_error("bn_bwd_reduce_post: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_bn_bwd_apply_post(const void* dy, const void* raw, const float* scale_shift, const float* mean_invstd, const float* sums,
                                      const void* res_pre, const void* res_post, void* draw, void* dres, void* dpost, float* dgamma,
                                      float* dbeta, int64_t rows, int C, int act, double n, int training, float param_grad_scale, int dtype,
                                      void* stream) {
  RSSF_REQUIRE(dy && raw && scale_shift && mean_invstd && sums && draw && rows > 0 && C > 0 && act_ok(act), "bn_bwd_apply_post: bad arguments");
  RSSF_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_bwd_apply_post: dgamma and dbeta go together");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32)
    return bwd_apply_launch<float>(dy, raw, scale_shift, mean_invstd, sums, res_pre, draw, dres, dgamma, dbeta, rows, C, act, (float)n,
                                   training, param_grad_scale, st, res_post, dpost);
  if (dtype == RSSF_BF16)
    return bwd_apply_launch<bf16_t>(dy, raw, scale_shift, mean_invstd, sums, res_pre, draw, dres, dgamma, dbeta, rows, C, act, (float)n,
                                    training, param_grad_scale, st, res_post, dpost);
  set_error("bn_bwd_apply_post: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_bn_bwd_apply(const void* dy, const void* raw, const float* scale_shift, const float* mean_invstd, const float* sums,
                                 const void* res_pre, void* draw, void* dres, float* dgamma, float* dbeta, int64_t rows, int C, int act,
                                 double n, int training, float param_grad_scale, int dtype, void* stream) {
  RSSF_REQUIRE(dy && raw && scale_shift && mean_invstd && sums && draw && rows > 0 && C > 0, "bn_bwd_apply: bad arguments");
  RSSF_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_bwd_apply: dgamma and dbeta go together");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32)
    return bwd_apply_launch<float>(dy, raw, scale_shift, mean_invstd, sums, res_pre, draw, dres, dgamma, dbeta, rows, C, act, (float)n,
                                   training, param_grad_scale, st);
  if (dtype == RSSF_BF16)
    return bwd_apply_launch<bf16_t>(dy, raw, scale_shift, mean_invstd, sums, res_pre, draw, dres, dgamma, dbeta, rows, C, act, (float)n,
                                    training, param_grad_scale, st);
  set_error("bn_bwd_apply: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

// ---- grouped BatchNorm passes (rssf.h "Grouped launches") ---------------------------------------------------------------------
namespace {
template <typename T>
int finapply_group_launch(const rssf_bn_apply_item* items, int n, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  FinApplyGroup g;
  g.n = n;
  int blocks = 0, maxc = 0;
  for (int i = 0; i < n; ++i) {
    const rssf_bn_apply_item& it = items[i];
    const dim3 grid = grid2d(it.rows, it.C, V);
    g.it[i] = {it.raw, it.stats, it.gamma, it.beta, it.running_mean, it.running_var, it.mean_invstd, it.scale_shift, it.res_pre, it.res_post, it.y,
               it.rows, it.C, it.act, it.training, (int)grid.x, (float)it.n, it.momentum, it.eps};
    g.start[i] = blocks;
    blocks += (int)(grid.x * grid.y);
    const int nch = (it.C / V < 256 ? it.C / V : 256) * V;
    if (nch > maxc) maxc = nch;
  }
  for (int k = n; k <= RSSF_GROUP_MAX; ++k) g.start[k] = blocks;
  bn_finapply_group_kernel<T, V><<<(unsigned)blocks, 256, 2 * maxc * sizeof(float), st>>>(g);
  return check_launch("bn_finalize_apply_group");
}
template <typename T>
int reduce_group_launch(const rssf_bn_reduce_item* items, int n, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  ReduceGroup g;
  g.n = n;
  int blocks = 0, maxc = 0;
  for (int i = 0; i < n; ++i) {
    const rssf_bn_reduce_item& it = items[i];
    const int cols = it.C / V, cblocks = (cols + 255) / 256, rpb = 256 / (cols < 256 ? cols : 256);
    int64_t gx = (it.rows + 4 * rpb - 1) / (4 * rpb);
    if (gx > REDUCE_MAX_BLOCKS) gx = REDUCE_MAX_BLOCKS;
    g.it[i] = {it.dy, it.raw, it.scale_shift, it.res_pre, it.sums, it.rows, it.C, it.act, (int)gx};
    g.start[i] = blocks;
    blocks += (int)gx * cblocks;
    if (it.C > maxc) maxc = it.C;
  }
  for (int k = n; k <= RSSF_GROUP_MAX; ++k) g.start[k] = blocks;
  bn_bwd_reduce_group_kernel<T, V><<<(unsigned)blocks, 256, 2 * maxc * sizeof(float), st>>>(g);
  return check_launch("bn_bwd_reduce_group");
}
template <typename T>
int bwd_apply_group_launch(const rssf_bn_bwd_apply_item* items, int n, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  BwdApplyGroup g;
  g.n = n;
  int blocks = 0, maxc = 0;
  for (int i = 0; i < n; ++i) {
    const rssf_bn_bwd_apply_item& it = items[i];
    const dim3 grid = grid2d(it.rows, it.C, V);
    g.it[i] = {it.dy, it.raw, it.scale_shift, it.mean_invstd, it.sums, it.res_pre, it.draw, it.dres, it.dgamma, it.dbeta, it.rows, it.C, it.act,
               it.training, (int)grid.x, (float)it.n, it.param_grad_scale};
    g.start[i] = blocks;
    blocks += (int)(grid.x * grid.y);
    const int nch = it.C < 256 * V ? it.C : 256 * V;
    if (nch > maxc) maxc = nch;
  }
  for (int k = n; k <= RSSF_GROUP_MAX; ++k) g.start[k] = blocks;
  bn_bwd_apply_group_kernel<T, V><<<(unsigned)blocks, 256, 2 * maxc * sizeof(float), st>>>(g);
  return check_launch("bn_bwd_apply_group");
}
}  // namespace

extern "C" int rssf_bn_finalize_apply_group(const rssf_bn_apply_item* items, int n, int dtype, void* stream) {
  RSSF_REQUIRE(items && n >= 1 && (dtype == RSSF_F32 || dtype == RSSF_BF16), "bn_finalize_apply_group: bad arguments");
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  bool grouped = n >= 2 && n <= RSSF_GROUP_MAX;
  for (int i = 0; i < n; ++i) {
    const rssf_bn_apply_item& it = items[i];
    RSSF_REQUIRE(it.raw && it.gamma && it.beta && it.mean_invstd && it.scale_shift && it.y && it.rows > 0 && it.C > 0 && act_ok(it.act),
                 "bn_finalize_apply_group: bad item %d", i);
    RSSF_REQUIRE(it.training ? (it.stats != nullptr && it.n >= 1) : (it.running_mean && it.running_var),
                 "bn_finalize_apply_group: missing statistics (item %d)", i);
    grouped = grouped && (it.C % V) == 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (grouped) return dtype == RSSF_BF16 ? finapply_group_launch<bf16_t>(items, n, st) : finapply_group_launch<float>(items, n, st);
  for (int i = 0; i < n; ++i) {
    const rssf_bn_apply_item& it = items[i];
    const int rc = rssf_bn_finalize_apply(it.raw, it.stats, it.gamma, it.beta, it.running_mean, it.running_var, it.mean_invstd, it.scale_shift,
                                          it.res_pre, it.res_post, it.y, it.rows, it.C, it.act, it.n, it.momentum, it.eps, it.training, dtype, stream);
    if (rc) return rc;
  }
  return RSSF_OK;
}

extern "C" int rssf_bn_bwd_reduce_group(const rssf_bn_reduce_item* items, int n, int dtype, void* stream) {
  RSSF_REQUIRE(items && n >= 1 && (dtype == RSSF_F32 || dtype == RSSF_BF16), "bn_bwd_reduce_group: bad arguments");
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  bool grouped = n >= 2 && n <= RSSF_GROUP_MAX;
  for (int i = 0; i < n; ++i) {
    const rssf_bn_reduce_item& it = items[i];
    RSSF_REQUIRE(it.dy && it.raw && it.scale_shift && it.sums && it.rows > 0 && it.C > 0 && it.act >= 0 && it.act <= 2, "bn_bwd_reduce_group: bad item %d", i);
    grouped = grouped && (it.C % V) == 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (grouped) return dtype == RSSF_BF16 ? reduce_group_launch<bf16_t>(items, n, st) : reduce_group_launch<float>(items, n, st);
  for (int i = 0; i < n; ++i) {
    const rssf_bn_reduce_item& it = items[i];
    const int rc = rssf_bn_bwd_reduce(it.dy, it.raw, it.scale_shift, it.res_pre, it.sums, it.rows, it.C, it.act, nullptr, dtype, stream);
    if (rc) return rc;
  }
  return RSSF_OK;
}

extern "C" int rssf_bn_bwd_apply_group(const rssf_bn_bwd_apply_item* items, int n, int dtype, void* stream) {
  RSSF_REQUIRE(items && n >= 1 && (dtype == RSSF_F32 || dtype == RSSF_BF16), "bn_bwd_apply_group: bad arguments");
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  bool grouped = n >= 2 && n <= RSSF_GROUP_MAX;
  for (int i = 0; i < n; ++i) {
    const rssf_bn_bwd_apply_item& it = items[i];
    RSSF_REQUIRE(it.dy && it.raw && it.scale_shift && it.mean_invstd && it.sums && it.draw && it.rows > 0 && it.C > 0 && it.act >= 0 && it.act <= 2,
                 "bn_bwd_apply_group: bad item %d", i);
    RSSF_REQUIRE((it.dgamma == nullptr) == (it.dbeta == nullptr), "bn_bwd_apply_group: dgamma and dbeta go together (item %d)", i);
    grouped = grouped && (it.C % V) == 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (grouped) return dtype == RSSF_BF16 ? bwd_apply_group_launch<bf16_t>(items, n, st) : bwd_apply_group_launch<float>(items, n, st);
  for (int i = 0; i < n; ++i) {
    const rssf_bn_bwd_apply_item& it = items[i];
    const int rc = rssf_bn_bwd_apply(it.dy, it.raw, it.scale_shift, it.mean_invstd, it.sums, it.res_pre, it.draw, it.dres, it.dgamma, it.dbeta, it.rows,
                                     it.C, it.act, it.n, it.training, it.param_grad_scale, dtype, stream);
    if (rc) return rc;
  }
  return RSSF_OK;
}
