// Inference operators of the SCD class-activation-map path (BASELINE config 5 as it is worded: Mix-Transformer backbone + CAM).
// Reference: SCD-AAAI2023/network/mix_transformer.py (Attention.forward :93-131, DWConv :377-388, Mlp.forward :45-52),
// SCD-AAAI2023/network/TSCD_model.py:66-79 (attn_proj + sigmoid, CAM from the classifier weights) and
// SCD-AAAI2023/utils/camutils.py:85-113 (multi_scale_cam).  Forward only: the reference extracts CAMs under no_grad.
//
//   rssf_mha_fwd            softmax(q k^T * scale) v for a SHORT key/value sequence (spatial-reduction attention: 100..121 keys;
//                           stage 4 of MiT: all 441..961 tokens), optionally the raw q k^T logits (the reference returns them)
//   rssf_dwconv3x3          depth-wise 3x3 convolution + bias (+ GELU), channels-last
//   rssf_attn_proj_sigmoid  sigmoid(Conv2d(16 -> 1, 1x1)) over the concatenated logits of the last two blocks
//   rssf_attn_pred          the same prediction from the q / kv projections of the two blocks, no logit tensors
//   rssf_resize_bilinear    F.interpolate(mode='bilinear', align_corners=False)
//   rssf_cam_merge          interpolate the CAMs of an image and of its flip to the image size, max, ReLU, (+=)
//   rssf_cam_normalize      per-plane (x - min) / (max - min + 1e-5)
#include "common.hip.h"
using namespace rssf;

namespace {
constexpr int ACT_NONE = 0, ACT_GELU = 2;      // the activation codes of the BatchNorm passes (bn.hip)

// ---- attention --------------------------------------------------------------------------------------------------------------
// One wave owns 16 queries of one head, a block (4 waves) 64; keys / values pass through LDS in chunks of 64 and are shared by
// the four waves.  The products run TRANSPOSED so that every MFMA result is the next MFMA's operand in registers:
//   S^T (keys x queries) = K Q^T            A = K tile from LDS (k-contiguous), B = the wave's Q fragments (registers)
//   O^T (d x queries)   += V^T P^T          B = exp2(S^T - max) as it sits in the accumulator layout (lane: keys 4g..4g+3 of
//                                           query l&15), A = V^T: bf16 from the TRANSPOSED LDS copy of the value chunk (one
//                                           8-byte read), f32 (K-step 4) straight from the row-major copy
// A query is a COLUMN of both results, i.e. lanes l&15 = n: its running maximum / sum are per-lane values (4 lane groups hold
// the same query: two lane swaps per chunk), the rescaling of O^T is a per-lane multiply, and the output / logit rows are
// contiguous 4-element stores.
constexpr int MHA_KC = 64;
template <typename T, int D> struct MhaLds {
  static constexpr bool TR = sizeof(T) == 2;                       // bf16: V chunk stored transposed
  static constexpr int LDK = D + (TR ? 8 : 4);
  static constexpr int LDV = TR ? MHA_KC + 8 : D + 4;
  static constexpr int K_ELEMS = MHA_KC * LDK;
  static constexpr int V_ELEMS = TR ? D * LDV : MHA_KC * LDV;
};

template <typename T, int D>
__global__ void __launch_bounds__(256) mha_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ out,
                                                      float* __restrict__ logits, int N, int M, int heads, float c) {
  using LY = MhaLds<T, D>;
  using MM = Mma<T>;
  constexpr int VEC = Vec<T>::N, KSTEP = MM::KSTEP, KPL = MM::KPL, NKK = D / KSTEP, NT = D / 16, NKT = MHA_KC / 16;
  __shared__ __attribute__((aligned(16))) T lds[LY::K_ELEMS + LY::V_ELEMS];
  T* Ks = lds;
  T* Vs = lds + LY::K_ELEMS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, C = heads * D;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int qrow = q0 + n < N ? q0 + n : N - 1;
  const bool qlive = q0 + n < N;
  typename MM::frag qf[NKK];
  {
    const T* qp = q + ((int64_t)b * N + qrow) * C + h * D + g * KPL;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = MM::load(qp + kk * KSTEP);
  }
  f32x4 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const T* kbase = kv + (int64_t)b * M * 2 * C + h * D;
  float* lrow = logits ? logits + (((int64_t)b * heads + h) * N + qrow) * M : nullptr;

  for (int kc0 = 0; kc0 < M; kc0 += MHA_KC) {
    __syncthreads();
    // ---- stage the chunk: keys row-major; values transposed (bf16) or row-major (f32); rows past M are zero
    for (int i = threadIdx.x; i < MHA_KC * (D / VEC); i += 256) {
      const int r = i / (D / VEC), cv = i % (D / VEC);
      Vec<T> vk, vv;
      if (kc0 + r < M) {
        const T* p = kbase + (int64_t)(kc0 + r) * 2 * C + cv * VEC;
        vk.load(p);
        vv.load(p + C);
      } else {
        vk.clear();
        vv.clear();
      }
      vk.store(Ks + r * LY::LDK + cv * VEC);
      if constexpr (LY::TR) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) stf(Vs + (cv * VEC + e) * LY::LDV + r, vv.get(e));
      } else {
        vv.store(Vs + r * LY::LDV + cv * VEC);
      }
    }
    __syncthreads();
    const int nkt = (M - kc0 + 15) / 16 < NKT ? (M - kc0 + 15) / 16 : NKT;       // live key tiles of this chunk (block-uniform)
    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
        const T* ka = Ks + (kt * 16 + n) * LY::LDK + g * KPL;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) s[kt] = MM::mma(MM::load(ka + kk * KSTEP), qf[kk], s[kt]);
        const int key = kc0 + kt * 16 + g * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (lrow && qlive && key + r < M) lrow[key + r] = s[kt][r];
          s[kt][r] = key + r < M ? s[kt][r] * c : -INFINITY;
          mx = fmaxf(mx, s[kt][r]);
        }
      }
    }
    mx = rows_reduce<OpMax>(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kt][r] = exp2f(s[kt][r] - m_new);
          lsum += s[kt][r];
        }
      }
    l_run = l_run * alpha + lsum;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t][r] *= alpha;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        if (kt < nkt) {
          if constexpr (LY::TR) {
            o[t] = MM::mma(MM::load(Vs + (t * 16 + n) * LY::LDV + kt * 16 + g * 4), pack_bf16x4(s[kt]), o[t]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)          // K-step 4: slot g of step r <-> key 4g + r on both operands
              o[t] = MM::mma(Vs[(kt * 16 + g * 4 + r) * LY::LDV + t * 16 + n], s[kt][r], o[t]);
          }
        }
    }
  }
  const float inv = 1.f / rows_reduce<OpSum>(l_run);
  if (qlive) {
    T* op = out + ((int64_t)b * N + qrow) * C + h * D + g * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) stf(op + t * 16 + r, o[t][r] * inv);
    }
  }
}

template <typename T>
int mha_launch(const void* q, const void* kv, void* out, float* logits, int B, int N, int M, int heads, int d, float scale, hipStream_t st) {
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)heads, (unsigned)B);
  const float c = scale * 1.4426950408889634f;
  if (d == 64) mha_fwd_kernel<T, 64><<<grid, 256, 0, st>>>((const T*)q, (const T*)kv, (T*)out, logits, N, M, heads, c);
  else if (d == 32) mha_fwd_kernel<T, 32><<<grid, 256, 0, st>>>((const T*)q, (const T*)kv, (T*)out, logits, N, M, heads, c);
  else { set_error("mha_fwd: head_dim %d is not built (32 and 64 are: MiT-B0 / B1..B5)", d); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("mha_fwd");
}

// ---- depth-wise 3x3 + bias (+ GELU) -------------------------------------------------------------------------------------------
// A thread owns VEC channels of a run of DW_SEG pixels of one image row: its 9 x VEC weights and the bias sit in registers, the
// 3 x 3 window slides along the row (three 16-byte loads per output pixel instead of nine), consecutive threads own consecutive
// channel vectors of the same pixels (coalesced).  HBM-bound: one read and one write of the activation (the halo rows come from
// L2); the first form of this kernel - nine loads and 9 x VEC scalar weight loads per output - ran at 0.25 TB/s.
constexpr int DW_SEG = 16;
template <typename T, int VEC>
__global__ void __launch_bounds__(256) dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ y, int B, int H, int W, int C, int act) {
  const int cols = C / VEC, segs = (W + DW_SEG - 1) / DW_SEG;
  const int64_t total = (int64_t)B * H * segs * cols;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cv = (int)(i % cols);
  int64_t p = i / cols;
  const int seg = (int)(p % segs); p /= segs;
  const int yy = (int)(p % H);
  const int b = (int)(p / H);
  const int c0 = cv * VEC, x0 = seg * DW_SEG;
  float wt[9][VEC], bs[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    bs[e] = bias ? bias[c0 + e] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t][e] = w[(c0 + e) * 9 + t];
  }
  const T* img = x + (int64_t)b * H * W * C + c0;
  float win[3][3][VEC];                       // [row][column slot][channel]: columns xx-1, xx, xx+1
  auto load_col = [&](int slot, int ix) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = yy + r - 1;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      if constexpr (VEC > 1) {
        Vec<T> v;
        if (ok) v.load(img + ((int64_t)iy * W + ix) * C); else v.clear();
#pragma unroll
        for (int e = 0; e < VEC; ++e) win[r][slot][e] = v.get(e);
      } else {
        win[r][slot][0] = ok ? ldf(img + ((int64_t)iy * W + ix) * C) : 0.f;
      }
    }
  };
  load_col(0, x0 - 1);
  load_col(1, x0);
  T* dst = y + (((int64_t)b * H + yy) * W + x0) * C + c0;
#pragma unroll
  for (int k = 0; k < DW_SEG; ++k) {
    const int xx = x0 + k;
    if (xx >= W) break;
    load_col(2, xx + 1);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float a = bs[e];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) a = fmaf(win[r][cidx][e], wt[r * 3 + cidx][e], a);
      acc[e] = act == ACT_GELU ? gelu_erf(a) : a;
    }
    if constexpr (VEC > 1) {
      Vec<T> o;
      o.set_all(acc);
      o.store(dst + (int64_t)k * C);
    } else {
      stf(dst + (int64_t)k * C, acc[0]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { win[r][0][e] = win[r][1][e]; win[r][1][e] = win[r][2][e]; }
  }
}

int grid_of(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

template <typename T>
int dwconv_launch(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int act, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  const int64_t runs = (int64_t)B * H * ((W + DW_SEG - 1) / DW_SEG);
  if (C % V == 0) dwconv3x3_kernel<T, V><<<(unsigned)((runs * (C / V) + 255) / 256), 256, 0, st>>>((const T*)x, w, bias, (T*)y, B, H, W, C, act);
  else dwconv3x3_kernel<T, 1><<<(unsigned)((runs * C + 255) / 256), 256, 0, st>>>((const T*)x, w, bias, (T*)y, B, H, W, C, act);
  return check_launch("dwconv3x3");
}

// ---- sigmoid(1x1 conv over the 2 x heads logit planes) ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) attn_proj_sigmoid_kernel(const float* __restrict__ a0, const float* __restrict__ a1,
                                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                                float* __restrict__ out, int B, int heads, int64_t plane) {
  const int64_t total = (int64_t)B * plane;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / plane, p = i - b * plane;
    float acc = bias ? bias[0] : 0.f;
    for (int h = 0; h < heads; ++h) {
      acc = fmaf(w[h], a0[(b * heads + h) * plane + p], acc);
      acc = fmaf(w[heads + h], a1[(b * heads + h) * plane + p], acc);
    }
    out[i] = sigmoidf(acc);
  }
}

// ---- the attention prediction without the logit tensors -------------------------------------------------------------------------
// out[b][n][m] = sigmoid(bias + sum over the two blocks s and their heads h of w[s * heads + h] * (q_s[b,n,h,:] . k_s[b,m,h,:])):
// TSCD.forward's sigmoid(attn_proj(cat(attns[-2:]))) computed from the q / kv projections directly.  The [B, heads, N, M] fp32
// logits of the two blocks (1.9 GB written and read again at B = 32, 31 x 31 tokens) never exist; per (block, head) the 64-key tile
// of k is staged once per workgroup, the head's products are formed in fp32 accumulators and folded into the running sum with
// that head's weight - the arithmetic of the reference (fp32 logits times weights), in another order.
template <typename T, int D>
__global__ void __launch_bounds__(256) attn_pred_kernel(const T* __restrict__ q0, const T* __restrict__ kv0, const T* __restrict__ q1,
                                                        const T* __restrict__ kv1, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, int N, int M, int heads) {
  using MM = Mma<T>;
  constexpr int VEC = Vec<T>::N, KSTEP = MM::KSTEP, KPL = MM::KPL, NKK = D / KSTEP, LDK = D + (sizeof(T) == 2 ? 8 : 4);
  __shared__ __attribute__((aligned(16))) T Ks[64 * LDK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, C = heads * D, kc0 = blockIdx.y * 64;
  const int qa = blockIdx.x * 64 + wave * 16 + n;
  const int qrow = qa < N ? qa : N - 1;
  f32x4 tot[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) tot[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sh = 0; sh < 2 * heads; ++sh) {
    const int src = sh / heads, h = sh - src * heads;
    const T* qs = src ? q1 : q0;
    const T* ks = (src ? kv1 : kv0) + (int64_t)b * M * 2 * C + h * D;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * (D / VEC); i += 256) {
      const int r = i / (D / VEC), cv = i % (D / VEC);
      Vec<T> vk;
      if (kc0 + r < M) vk.load(ks + (int64_t)(kc0 + r) * 2 * C + cv * VEC); else vk.clear();
      vk.store(Ks + r * LDK + cv * VEC);
    }
    typename MM::frag qf[NKK];
    const T* qp = qs + ((int64_t)b * N + qrow) * C + h * D + g * KPL;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = MM::load(qp + kk * KSTEP);
    __syncthreads();
    const float wh = w[sh];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
      const T* ka = Ks + (kt * 16 + n) * LDK + g * KPL;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) s = MM::mma(MM::load(ka + kk * KSTEP), qf[kk], s);
#pragma unroll
      for (int r = 0; r < 4; ++r) tot[kt][r] = fmaf(wh, s[r], tot[kt][r]);
    }
  }
  if (qa < N) {
    const float b0 = bias ? bias[0] : 0.f;
    float* orow = out + ((int64_t)b * N + qa) * M;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kc0 + kt * 16 + g * 4 + r;
        if (key < M) orow[key] = sigmoidf(b0 + tot[kt][r]);
      }
  }
}

template <typename T>
int attn_pred_launch(const void* q0, const void* kv0, const void* q1, const void* kv1, const float* w, const float* bias, float* out, int B, int N,
                     int M, int heads, int d, hipStream_t st) {
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)B);
  if (d == 64) attn_pred_kernel<T, 64><<<grid, 256, 0, st>>>((const T*)q0, (const T*)kv0, (const T*)q1, (const T*)kv1, w, bias, out, N, M, heads);
  else if (d == 32) attn_pred_kernel<T, 32><<<grid, 256, 0, st>>>((const T*)q0, (const T*)kv0, (const T*)q1, (const T*)kv1, w, bias, out, N, M, heads);
  else { set_error("attn_pred: head_dim %d is not built (32 and 64 are)", d); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("attn_pred");
}

// ---- half-pixel bilinear sampling (align_corners=False; PyTorch's upsample_bilinear2d with `size=`) ---------------------------
struct Tap2 { int i0, i1; float w; };
__device__ __forceinline__ Tap2 hp_tap(int o, float scale, int in) {
  float s = ((float)o + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  Tap2 t;
  t.i0 = (int)s;
  if (t.i0 > in - 1) t.i0 = in - 1;
  t.i1 = t.i0 + 1 < in ? t.i0 + 1 : in - 1;
  t.w = s - (float)t.i0;
  return t;
}

template <typename T>
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int IH, int IW, int OH, int OW,
                                                              int C, float sy, float sx) {
  const int64_t total = (int64_t)B * OH * OW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % C);
    int64_t p = i / C;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const Tap2 ty = hp_tap(oy, sy, IH), tx = hp_tap(ox, sx, IW);
    const T* base = in + (int64_t)b * IH * IW * C + ch;
    const float v00 = ldf(base + ((int64_t)ty.i0 * IW + tx.i0) * C), v01 = ldf(base + ((int64_t)ty.i0 * IW + tx.i1) * C);
    const float v10 = ldf(base + ((int64_t)ty.i1 * IW + tx.i0) * C), v11 = ldf(base + ((int64_t)ty.i1 * IW + tx.i1) * C);
    // PyTorch's order: the two rows are interpolated along x, then blended along y
    const float top = v00 * (1.f - tx.w) + v01 * tx.w, bot = v10 * (1.f - tx.w) + v11 * tx.w;
    stf(out + i, top * (1.f - ty.w) + bot * ty.w);
  }
}

// acc[b][k][y][x] (+)= relu(max(up(cam[b])(y, x), up(cam[b + B])(y, W-1-x)))      (camutils.py:93-96, 105-108)
template <typename T>
__global__ void __launch_bounds__(256) cam_merge_kernel(const T* __restrict__ cam, float* __restrict__ acc, int B, int K, int CH, int CW, int H,
                                                        int W, float sy, float sx, int accumulate) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const Tap2 ty = hp_tap(y, sy, CH), tx = hp_tap(x, sx, CW), tf = hp_tap(W - 1 - x, sx, CW);
    const T* c0 = cam + (int64_t)b * CH * CW * K;
    const T* c1 = cam + (int64_t)(b + B) * CH * CW * K;
    auto sample = [&](const T* base, const Tap2& tq, int k) {
      const float v00 = ldf(base + ((int64_t)ty.i0 * CW + tq.i0) * K + k), v01 = ldf(base + ((int64_t)ty.i0 * CW + tq.i1) * K + k);
      const float v10 = ldf(base + ((int64_t)ty.i1 * CW + tq.i0) * K + k), v11 = ldf(base + ((int64_t)ty.i1 * CW + tq.i1) * K + k);
      const float top = v00 * (1.f - tq.w) + v01 * tq.w, bot = v10 * (1.f - tq.w) + v11 * tq.w;
      return top * (1.f - ty.w) + bot * ty.w;
    };
    for (int k = 0; k < K; ++k) {
      const float v = fmaxf(fmaxf(sample(c0, tx, k), sample(c1, tf, k)), 0.f);
      float* dst = acc + (((int64_t)b * K + k) * H + y) * W + x;
      *dst = accumulate ? *dst + v : v;
    }
  }
}

// per plane: x <- (x - min) / (max - min + 1e-5)      (camutils.py:111-112: cam + max(-cam); cam / (max(cam) + 1e-5))
__global__ void __launch_bounds__(1024) cam_normalize_kernel(float* __restrict__ cam, int64_t n) {
  __shared__ float smin[16], smax[16];
  float* p = cam + (int64_t)blockIdx.x * n;
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { const float v = p[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
  mn = -wave_max(-mn);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = smin[0]; mx = smax[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { mn = fminf(mn, smin[w]); mx = fmaxf(mx, smax[w]); }
  const float shift = -mn;                         // = adaptive_max_pool2d(-cam)
  const float denom = (mx + shift) + 1e-5f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = (p[i] + shift) / denom;
}
}  // namespace

extern "C" int rssf_mha_fwd(const void* q, const void* kv, void* out, float* logits, int B, int N, int M, int heads, int head_dim, float scale,
                            int dtype, void* stream) {
  RSSF_REQUIRE(q && kv && out && B > 0 && N > 0 && M > 0 && heads > 0 && B <= 65535 && heads <= 65535, "mha_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return mha_launch<float>(q, kv, out, logits, B, N, M, heads, head_dim, scale, st);
  if (dtype == RSSF_BF16) return mha_launch<bf16_t>(q, kv, out, logits, B, N, M, heads, head_dim, scale, st);
  set_error("mha_fwd: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_dwconv3x3(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int act, int dtype,
                              void* stream) {
  RSSF_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && (act == ACT_NONE || act == ACT_GELU), "dwconv3x3: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return dwconv_launch<float>(x, w, bias, y, B, H, W, C, act, st);
  if (dtype == RSSF_BF16) return dwconv_launch<bf16_t>(x, w, bias, y, B, H, W, C, act, st);
  set_error("dwconv3x3: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_attn_proj_sigmoid(const float* a0, const float* a1, const float* w, const float* bias, float* out, int B, int heads,
                                      int64_t plane, void* stream) {
  RSSF_REQUIRE(a0 && a1 && w && out && B > 0 && heads > 0 && plane > 0, "attn_proj_sigmoid: bad arguments");
  attn_proj_sigmoid_kernel<<<grid_of((int64_t)B * plane), 256, 0, (hipStream_t)stream>>>(a0, a1, w, bias, out, B, heads, plane);
  return check_launch("attn_proj_sigmoid");
}

extern "C" int rssf_attn_pred(const void* q0, const void* kv0, const void* q1, const void* kv1, const float* w, const float* bias, float* out,
                             int B, int N, int M, int heads, int head_dim, int dtype, void* stream) {
  RSSF_REQUIRE(q0 && kv0 && q1 && kv1 && w && out && B > 0 && N > 0 && M > 0 && heads > 0 && B <= 65535 && (M + 63) / 64 <= 65535,
               "attn_pred: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return attn_pred_launch<float>(q0, kv0, q1, kv1, w, bias, out, B, N, M, heads, head_dim, st);
  if (dtype == RSSF_BF16) return attn_pred_launch<bf16_t>(q0, kv0, q1, kv1, w, bias, out, B, N, M, heads, head_dim, st);
  set_error("attn_pred: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_resize_bilinear(const void* in, void* out, int B, int IH, int IW, int OH, int OW, int C, int dtype, void* stream) {
  RSSF_REQUIRE(in && out && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0, "resize_bilinear: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const int g = grid_of((int64_t)B * OH * OW * C);
  if (dtype == RSSF_F32) resize_bilinear_kernel<float><<<g, 256, 0, st>>>((const float*)in, (float*)out, B, IH, IW, OH, OW, C, sy, sx);
  else if (dtype == RSSF_BF16) resize_bilinear_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)in, (bf16_t*)out, B, IH, IW, OH, OW, C, sy, sx);
  else { set_error("resize_bilinear: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("resize_bilinear");
}

extern "C" int rssf_cam_merge(const void* cam, float* acc, int B, int K, int CH, int CW, int H, int W, int accumulate, int dtype, void* stream) {
  RSSF_REQUIRE(cam && acc && B > 0 && K > 0 && CH > 0 && CW > 0 && H > 0 && W > 0, "cam_merge: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const float sy = (float)CH / (float)H, sx = (float)CW / (float)W;
  const int g = grid_of((int64_t)B * H * W);
  if (dtype == RSSF_F32) cam_merge_kernel<float><<<g, 256, 0, st>>>((const float*)cam, acc, B, K, CH, CW, H, W, sy, sx, accumulate);
  else if (dtype == RSSF_BF16) cam_merge_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)cam, acc, B, K, CH, CW, H, W, sy, sx, accumulate);
  else { set_error("cam_merge: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("cam_merge");
}

extern "C" int rssf_cam_normalize(float* cam, int planes, int64_t n, void* stream) {
  RSSF_REQUIRE(cam && planes > 0 && n > 0, "cam_normalize: bad arguments");
  cam_normalize_kernel<<<(unsigned)planes, 1024, 0, (hipStream_t)stream>>>(cam, n);
  return check_launch("cam_normalize");
}
