// Inference operators of the SCD class-activation-map path (BASELINE config 5 as it is worded: Mix-Transformer backbone + CAM).
// Reference: SCD-AAAI2023/network/mix_transformer.py (Attention.forward :93-131, DWConv :377-388, Mlp.forward :45-52),
// SCD-AAAI2023/network/TSCD_model.py:66-79 (attn_proj + sigmoid, CAM from the classifier weights) and
// SCD-AAAI2023/utils/camutils.py:85-113 (multi_scale_cam).  Forward only: the reference extracts CAMs under no_grad.
//
//   rssf_mha_fwd            softmax(q k^T * scale) v for a SHORT key/value sequence (spatial-reduction attention: 100..121 keys;
//                           stage 4 of MiT: all 441..961 tokens), optionally the raw q k^T logits (the reference returns them)
//   rssf_dwconv3x3          depth-wise 3x3 convolution + bias (+ GELU), channels-last
//   rssf_attn_proj_sigmoid  sigmoid(Conv2d(16 -> 1, 1x1)) over the concatenated logits of the last two blocks
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
template <typename T, int VEC>
__global__ void __launch_bounds__(256) dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ y, int B, int H, int W, int C, int act) {
  const int cols = C / VEC;
  const int64_t total = (int64_t)B * H * W * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cols);
    int64_t p = i / cols;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const int b = (int)(p / H);
    const int c0 = cv * VEC;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = bias ? bias[c0 + e] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = yy + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = xx + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const T* src = x + (((int64_t)b * H + iy) * W + ix) * C + c0;
        if constexpr (VEC > 1) {
          Vec<T> v;
          v.load(src);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] = fmaf(v.get(e), w[(c0 + e) * 9 + ky * 3 + kx], acc[e]);
        } else {
          acc[0] = fmaf(ldf(src), w[c0 * 9 + ky * 3 + kx], acc[0]);
        }
      }
    }
    T* dst = y + (((int64_t)b * H + yy) * W + xx) * C + c0;
    if constexpr (VEC > 1) {
      Vec<T> o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.set(e, act == ACT_GELU ? gelu_erf(acc[e]) : acc[e]);
      o.store(dst);
    } else {
      stf(dst, act == ACT_GELU ? gelu_erf(acc[0]) : acc[0]);
    }
  }
}

int grid_of(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

template <typename T>
int dwconv_launch(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int act, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  if (C % V == 0) dwconv3x3_kernel<T, V><<<grid_of((int64_t)B * H * W * (C / V)), 256, 0, st>>>((const T*)x, w, bias, (T*)y, B, H, W, C, act);
  else dwconv3x3_kernel<T, 1><<<grid_of((int64_t)B * H * W * C), 256, 0, st>>>((const T*)x, w, bias, (T*)y, B, H, W, C, act);
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
