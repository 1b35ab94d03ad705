// Saliency gate of InterlacedPoolAttention2 (modules/multihead_isa_pool_attention.py:148-167) with the two
// SpatialAttention(7) gates (:101-115).  Bug-compatible with the reference's (B,N,C)->view(B,C,H,W)
// reinterpretation: for flat offset f = n*C + c inside one image, view-pixel p = f mod N, view-channel
// c' = f div N.  All maps are tiny ([B][k][N] fp32) and stay L2 resident;
// the only full-tensor traffic is one read of x and y (pool) and one read-modify-write (pool backward).
#include "common.hip.h"
using namespace rssf;

namespace {

// ---- pool: mean/max over view-channels of LN1(x), LN1(y) ------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) gate_pool_fwd_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                            const float* __restrict__ stx, const float* __restrict__ sty,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ pooled, int32_t* __restrict__ argmax,
                                                            int B, int N, int C) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * 2 * N;
  if (gid >= total) return;
  const int p = (int)(gid % N);
  const int s = (int)((gid / N) % 2);
  const int b = (int)(gid / (2 * (int64_t)N));
  const T* src = (s == 0 ? x : y) + (int64_t)b * N * C;
  const float* st = (s == 0 ? stx : sty) + (int64_t)b * N * 2;
  float sum = 0.f, mx = -INFINITY;
  int am = 0;
  for (int cp = 0; cp < C; ++cp) {
    const int64_t f = (int64_t)cp * N + p;
    const int n = (int)(f / C), c = (int)(f % C);
    const float v = (ldf(src + f) - st[n * 2]) * st[n * 2 + 1] * gamma[c] + beta[c];
    sum += v;
    if (v > mx) { mx = v; am = cp; }
  }
  float* pb = pooled + (int64_t)b * 4 * N;
  pb[(2 * s) * N + p] = sum / C;
  pb[(2 * s + 1) * N + p] = mx;
  argmax[((int64_t)b * 2 + s) * N + p] = am;
}

// Vector form (N and C multiples of the 16-byte vector): a thread owns VEC consecutive view-pixels; for each
// view-channel its VEC elements are one 16-byte load that lies inside ONE token row (so one {mean, rstd} pair), and the
// (token, channel) of the next view-channel follows by adding N/C and N%C with a carry - no division in the loop.
// PARTS waves share a run of 64 pixel groups, each walking a contiguous range of C / PARTS view-channels; the partial {sum, max,
// argmax} are folded through LDS in channel order (strictly greater wins: the FIRST maximum, as the one-thread walk - and
// torch.max - picks it).  With one thread per pixel group the launch was 256 blocks of 32 dependent 16-byte loads each: 23 us for
// 33 MB at B = 16 (1.5 TB/s).
template <typename T, int PARTS>
__global__ void __launch_bounds__(64 * PARTS) gate_pool_fwd_vec_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                                       const float* __restrict__ stx, const float* __restrict__ sty,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                       float* __restrict__ pooled, int32_t* __restrict__ argmax,
                                                                       int B, int N, int C) {
  constexpr int V = Vec<T>::N;
  __shared__ float ssum[PARTS > 1 ? PARTS - 1 : 1][64][V], smax[PARTS > 1 ? PARTS - 1 : 1][64][V];
  __shared__ int sarg[PARTS > 1 ? PARTS - 1 : 1][64][V];
  const int nv = N / V;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int64_t gid = (int64_t)blockIdx.x * 64 + lane;
  const bool live = gid < (int64_t)B * 2 * nv;
  const int64_t g = live ? gid : 0;
  const int p0 = (int)(g % nv) * V;
  const int s = (int)((g / nv) % 2);
  const int b = (int)(g / (2 * (int64_t)nv));
  const T* src = (s == 0 ? x : y) + (int64_t)b * N * C;
  const float2* st = reinterpret_cast<const float2*>((s == 0 ? stx : sty) + (int64_t)b * N * 2);
  const int qn = N / C, rn = N % C;
  const int cper = C / PARTS, cp0 = part * cper;
  // (token, channel) of view-channel cp0 at view-pixel p0: flat index cp0 * N + p0 of the [N][C] token tensor
  const int64_t f0 = (int64_t)cp0 * N + p0;
  int n = (int)(f0 / C), c = (int)(f0 % C);
  float sum[V], mx[V];
  int am[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { sum[i] = 0.f; mx[i] = -INFINITY; am[i] = cp0; }
#pragma unroll 8
  for (int cp = cp0; cp < cp0 + cper; ++cp) {
    Vec<T> v;
    v.load(src + (int64_t)cp * N + p0);
    const float2 ms = st[n];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float t = (v.get(i) - ms.x) * ms.y * gamma[c + i] + beta[c + i];
      sum[i] += t;
      if (t > mx[i]) { mx[i] = t; am[i] = cp; }
    }
    n += qn; c += rn;
    if (c >= C) { c -= C; ++n; }
  }
  if constexpr (PARTS > 1) {
    if (part > 0) {
#pragma unroll
      for (int i = 0; i < V; ++i) { ssum[part - 1][lane][i] = sum[i]; smax[part - 1][lane][i] = mx[i]; sarg[part - 1][lane][i] = am[i]; }
    }
    __syncthreads();
    if (part > 0) return;
#pragma unroll
    for (int k = 0; k < PARTS - 1; ++k)
#pragma unroll
      for (int i = 0; i < V; ++i) {
        sum[i] += ssum[k][lane][i];
        if (smax[k][lane][i] > mx[i]) { mx[i] = smax[k][lane][i]; am[i] = sarg[k][lane][i]; }
      }
  }
  if (!live) return;
  float* pb = pooled + (int64_t)b * 4 * N;
  int32_t* ab = argmax + ((int64_t)b * 2 + s) * N;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    pb[(2 * s) * N + p0 + i] = sum[i] / C;
    pb[(2 * s + 1) * N + p0 + i] = mx[i];
    ab[p0 + i] = am[i];
  }
}

// ---- the two 7x7 convolutions run on 16x16-pixel tiles staged (with a 3-pixel halo, zero outside the image) in LDS ---
constexpr int GT = 16, GH = GT + 6, GLD = GH + 1;
constexpr int GATE_SLOT_ELEMS = 196 + 6;            // dk [2][2][7][7], dwl [2][2], dbl [2]
__device__ __forceinline__ void gate_load_tile(const float* __restrict__ plane, int H, int W, int h0, int w0, float* sm) {
  for (int i = threadIdx.x; i < GH * GH; i += blockDim.x) {
    const int r = i / GH, c = i % GH, hh = h0 + r - 3, ww = w0 + c - 3;
    sm[r * GLD + c] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? plane[hh * W + ww] : 0.f;
  }
}

// ---- weights: 7x7 conv (2->1, pad 3, no bias) + sigmoid per stream, 1x1 conv 2->2 + softmax over the 2 streams
__global__ void __launch_bounds__(256) gate_weights_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ k,
                                                               const float* __restrict__ wl, const float* __restrict__ bl,
                                                               float* __restrict__ gsig, float* __restrict__ omega,
                                                               float* __restrict__ logits, int B, int H, int W) {
  __shared__ float sk[196];
  __shared__ float sm[4][GH * GLD];
  const int N = H * W;
  const int tw = (W + GT - 1) / GT, th = (H + GT - 1) / GT;
  const int b = blockIdx.x / (tw * th), t = blockIdx.x % (tw * th);
  const int h0 = (t / tw) * GT, w0 = (t % tw) * GT;
  for (int i = threadIdx.x; i < 196; i += blockDim.x) sk[i] = k[i];
#pragma unroll
  for (int pl = 0; pl < 4; ++pl) gate_load_tile(pooled + ((int64_t)b * 4 + pl) * N, H, W, h0, w0, sm[pl]);
  __syncthreads();
  const int lh = threadIdx.x / GT, lw = threadIdx.x % GT, h = h0 + lh, w = w0 + lw;
  if (h >= H || w >= W) return;
  float g[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float acc = 0.f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const float* m = sm[2 * s + ch] + lh * GLD + lw;
      const float* kk = sk + (s * 2 + ch) * 49;
#pragma unroll
      for (int u = 0; u < 7; ++u)
#pragma unroll
        for (int v = 0; v < 7; ++v) acc += kk[u * 7 + v] * m[u * GLD + v];
    }
    g[s] = sigmoidf(acc);
  }
  const float l0 = wl[0] * g[0] + wl[1] * g[1] + bl[0];
  const float l1 = wl[2] * g[0] + wl[3] * g[1] + bl[1];
  const float m = fmaxf(l0, l1);
  const float e0 = __expf(l0 - m), e1 = __expf(l1 - m);
  const float inv = 1.f / (e0 + e1);
  const int64_t o = (int64_t)b * 2 * N;
  const int p = h * W + w;
  gsig[o + p] = g[0]; gsig[o + N + p] = g[1];
  omega[o + p] = e0 * inv; omega[o + N + p] = e1 * inv;
  if (logits) { logits[o + p] = l0; logits[o + N + p] = l1; }
}

// ---- weights backward, stage 1: domega -> dpre (gradient at the 7x7 conv outputs, pre-sigmoid); dwl/dbl --------
__global__ void __launch_bounds__(256) gate_weights_bwd1_kernel(const float* __restrict__ domega, const float* __restrict__ gsig,
                                                                const float* __restrict__ omega, const float* __restrict__ wl,
                                                                float* __restrict__ dpre, float* __restrict__ slots,
                                                                int B, int N) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dwl[4], dbl[2]
  if (gid < (int64_t)B * N) {
    const int p = (int)(gid % N), b = (int)(gid / N);
    const int64_t o = (int64_t)b * 2 * N;
    const float w0 = omega[o + p], w1 = omega[o + N + p];
    const float d0 = domega[o + p], d1 = domega[o + N + p];
    const float dot = d0 * w0 + d1 * w1;
    const float dl0 = w0 * (d0 - dot), dl1 = w1 * (d1 - dot);   // softmax backward
    const float g0 = gsig[o + p], g1 = gsig[o + N + p];
    a[0] = dl0 * g0; a[1] = dl0 * g1; a[2] = dl1 * g0; a[3] = dl1 * g1; a[4] = dl0; a[5] = dl1;
    const float dg0 = wl[0] * dl0 + wl[2] * dl1, dg1 = wl[1] * dl0 + wl[3] * dl1;
    dpre[o + p] = dg0 * g0 * (1.f - g0);
    dpre[o + N + p] = dg1 * g1 * (1.f - g1);
  }
  __shared__ float red[6];
  if (threadIdx.x < 6) red[threadIdx.x] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 6; ++i) a[i] = wave_sum(a[i]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&red[i], a[i]);
  }
  __syncthreads();
  // one global atomic per value per block, spread over RSSF_GATE_SLOTS copies (1024 blocks on one address cost ~40 us)
  if (threadIdx.x < 6) atomicAdd(&slots[(blockIdx.x % RSSF_GATE_SLOTS) * GATE_SLOT_ELEMS + 196 + threadIdx.x], red[threadIdx.x]);
}

// stage 2: dpooled = conv^T(dpre, k) ; dk += sum_pixels dpre * shifted(pooled).  Same 16x16 tiles: the dpre halo tile
// serves the transposed convolution, the pooled halo tiles the kernel gradient; per-tap partial sums are folded over the
// wave by shuffles and over the block through LDS, one global atomic per tap per block.
__global__ void __launch_bounds__(256) gate_weights_bwd2_kernel(const float* __restrict__ dpre, const float* __restrict__ pooled,
                                                                const float* __restrict__ k, float* __restrict__ dpooled,
                                                                float* __restrict__ slots, int B, int H, int W) {
  __shared__ float sk[196];
  __shared__ float sdk[196];
  __shared__ float sd[2][GH * GLD];
  __shared__ float sm[4][GH * GLD];
  __shared__ float red[49][65];                          // per-tap partial sums of the 64 lane quads of the block
  const int N = H * W;
  const int tw = (W + GT - 1) / GT, th = (H + GT - 1) / GT;
  const int b = blockIdx.x / (tw * th), t = blockIdx.x % (tw * th);
  const int h0 = (t / tw) * GT, w0 = (t % tw) * GT;
  for (int i = threadIdx.x; i < 196; i += blockDim.x) { sk[i] = k[i]; sdk[i] = 0.f; }
#pragma unroll
  for (int s = 0; s < 2; ++s) gate_load_tile(dpre + ((int64_t)b * 2 + s) * N, H, W, h0, w0, sd[s]);
#pragma unroll
  for (int pl = 0; pl < 4; ++pl) gate_load_tile(pooled + ((int64_t)b * 4 + pl) * N, H, W, h0, w0, sm[pl]);
  __syncthreads();
  const int lh = threadIdx.x / GT, lw = threadIdx.x % GT, h = h0 + lh, w = w0 + lw;
  const bool ok = h < H && w < W;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* dp = sd[s] + lh * GLD + lw;             // dp[(3+a)*GLD + 3+c] = dpre(h+a, w+c)
    const float mine = ok ? dp[3 * GLD + 3] : 0.f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const float* m = sm[2 * s + ch] + lh * GLD + lw;
      const float* kk = sk + (s * 2 + ch) * 49;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 7; ++u)
#pragma unroll
        for (int v = 0; v < 7; ++v) {
          // forward: out[h,w] += k[u,v] * in[h+u-3, w+v-3]  ->  this input was read by output (h-(u-3), w-(v-3))
          acc += kk[u * 7 + v] * dp[(6 - u) * GLD + (6 - v)];
          // kernel-gradient partial: fold the lane quad with two DPP adds (no LDS traffic), one LDS write per quad
          float tq = mine * m[u * GLD + v];
          tq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tq), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
          tq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tq), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
          red[u * 7 + v][threadIdx.x >> 2] = tq;           // the four lanes of a quad store the same value
        }
      if (ok) dpooled[((int64_t)b * 4 + 2 * s + ch) * N + h * W + w] = acc;
      __syncthreads();
      if (threadIdx.x < 245) {                            // 5 threads per tap, ~13 quads each
        const int tap = threadIdx.x % 49, part = threadIdx.x / 49;
        float sum = 0.f;
        for (int q = part * 13; q < (part * 13 + 13 < 64 ? part * 13 + 13 : 64); ++q) sum += red[tap][q];
        atomicAdd(&sdk[(s * 2 + ch) * 49 + tap], sum);
      }
      __syncthreads();
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 196; i += blockDim.x) atomicAdd(&slots[(blockIdx.x % RSSF_GATE_SLOTS) * GATE_SLOT_ELEMS + i], sdk[i]);
}

// folds the slot copies into the parameter gradients: dk [196], dwl [4], dbl [2]
__global__ void __launch_bounds__(256) gate_weights_fold_kernel(const float* __restrict__ slots, float* __restrict__ dk, float* __restrict__ dk1,
                                                               float* __restrict__ dwl, float* __restrict__ dbl) {
  const int i = threadIdx.x;
  if (i >= GATE_SLOT_ELEMS) return;
  float s = 0.f;
  for (int k = 0; k < RSSF_GATE_SLOTS; ++k) s += slots[k * GATE_SLOT_ELEMS + i];
  if (i < 98) dk[i] += s;
  else if (i < 196) dk1[i - 98] += s;          // stream 1's kernel: dk + 98, or a buffer of its own (two separate parameters)
  else if (i < 200) dwl[i - 196] += s;
  else dbl[i - 200] += s;
}

// ---- pool backward: add the gate-path gradient into d(LN1 output) ----------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) gate_pool_bwd_kernel(const float* __restrict__ dpooled, const int32_t* __restrict__ argmax,
                                                            T* __restrict__ dxhat, T* __restrict__ dyhat, int B, int N, int C) {
  const int64_t per = (int64_t)N * C;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * 2 * per) return;
  const int64_t f = gid % per;
  const int s = (int)((gid / per) % 2);
  const int b = (int)(gid / (2 * per));
  const int p = (int)(f % N), cp = (int)(f / N);
  const float* dpb = dpooled + (int64_t)b * 4 * N;
  float g = dpb[(2 * s) * N + p] / C;
  if (argmax[((int64_t)b * 2 + s) * N + p] == cp) g += dpb[(2 * s + 1) * N + p];
  T* dst = (s == 0 ? dxhat : dyhat) + (int64_t)b * per + f;
  stf(dst, ldf(dst) + g);
}

// Vector form: a thread owns V consecutive view-pixels of one stream, keeps their {d mean, d max, argmax} in registers and
// walks the C view-channels (one 16-byte read-modify-write each): the [B][k][N] maps are read once, not C times.
template <typename T, int PARTS>
__global__ void __launch_bounds__(64 * PARTS) gate_pool_bwd_vec_kernel(const float* __restrict__ dpooled, const int32_t* __restrict__ argmax,
                                                                       T* __restrict__ dxhat, T* __restrict__ dyhat, int B, int N, int C) {
  constexpr int V = Vec<T>::N;
  const int nv = N / V;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;      // PARTS waves per run of 64 pixel groups: a channel range each
  const int64_t gid = (int64_t)blockIdx.x * 64 + lane;
  if (gid >= (int64_t)B * 2 * nv) return;
  const int p0 = (int)(gid % nv) * V;
  const int s = (int)((gid / nv) % 2);
  const int b = (int)(gid / (2 * (int64_t)nv));
  const float* dm = dpooled + ((int64_t)b * 4 + 2 * s) * N + p0;
  const int32_t* am = argmax + ((int64_t)b * 2 + s) * N + p0;
  float gmean[V], gmax[V];
  int a[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { gmean[i] = dm[i] / C; gmax[i] = dm[N + i]; a[i] = am[i]; }
  T* dst = (s == 0 ? dxhat : dyhat) + (int64_t)b * N * C + p0;
  const int cper = (C + PARTS - 1) / PARTS, cp0 = part * cper, cp1 = cp0 + cper < C ? cp0 + cper : C;
#pragma unroll 8
  for (int cp = cp0; cp < cp1; ++cp) {
    Vec<T> v;
    v.load(dst + (int64_t)cp * N);
    float o[V];
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = v.get(i) + gmean[i] + (a[i] == cp ? gmax[i] : 0.f);
    v.set_all(o);
    v.store(dst + (int64_t)cp * N);
  }
}

}  // namespace

extern "C" int rssf_gate_pool_fwd(const void* x, const void* y, const float* stats_x, const float* stats_y,
                                  const float* gamma, const float* beta, float* pooled, int32_t* argmax, int B, int N,
                                  int C, int dtype, void* stream) {
  RSSF_REQUIRE(x && y && stats_x && stats_y && gamma && beta && pooled && argmax && B > 0 && N > 0 && C > 0,
               "gate_pool_fwd: bad arguments");
  const int64_t total = (int64_t)B * 2 * N;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  if ((dtype == RSSF_F32 || dtype == RSSF_BF16) && N % V == 0 && C % V == 0) {
    dim3 gv((unsigned)((total / V + 63) / 64));                // 64 pixel groups per block
    const bool split = C % 4 == 0;                               // four waves, a quarter of the view-channels each
    if (dtype == RSSF_F32) {
      if (split) gate_pool_fwd_vec_kernel<float, 4><<<gv, 256, 0, st>>>((const float*)x, (const float*)y, stats_x, stats_y, gamma, beta, pooled, argmax, B, N, C);
      else gate_pool_fwd_vec_kernel<float, 1><<<gv, 64, 0, st>>>((const float*)x, (const float*)y, stats_x, stats_y, gamma, beta, pooled, argmax, B, N, C);
    } else {
      if (split) gate_pool_fwd_vec_kernel<bf16_t, 4><<<gv, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)y, stats_x, stats_y, gamma, beta, pooled, argmax, B, N, C);
      else gate_pool_fwd_vec_kernel<bf16_t, 1><<<gv, 64, 0, st>>>((const bf16_t*)x, (const bf16_t*)y, stats_x, stats_y, gamma, beta, pooled, argmax, B, N, C);
    }
  } else if (dtype == RSSF_F32)
    gate_pool_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)x, (const float*)y, stats_x, stats_y, gamma, beta,
                                                       pooled, argmax, B, N, C);
  else if (dtype == RSSF_BF16)
    gate_pool_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)y, stats_x, stats_y, gamma, beta,
                                                        pooled, argmax, B, N, C);
  else { set_error("gate_pool_fwd: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("gate_pool_fwd");
}

extern "C" int rssf_gate_weights_fwd(const float* pooled, const float* k, const float* wl, const float* bl, float* gsig,
                                     float* omega, float* logits, int B, int H, int W, void* stream) {
  RSSF_REQUIRE(pooled && k && wl && bl && gsig && omega && B > 0 && H > 0 && W > 0, "gate_weights_fwd: bad arguments");
  const int tiles = ((H + GT - 1) / GT) * ((W + GT - 1) / GT);
  gate_weights_fwd_kernel<<<dim3((unsigned)(B * tiles)), 256, 0, (hipStream_t)stream>>>(pooled, k, wl, bl, gsig, omega, logits, B, H, W);
  return check_launch("gate_weights_fwd");
}

extern "C" int rssf_gate_weights_bwd(const float* domega, const float* pooled, const float* gsig, const float* omega,
                                     const float* k, const float* wl, float* dpooled, float* dk, float* dk_stream1, float* dwl,
                                     float* dbl, int B, int H, int W, void* stream) {
  RSSF_REQUIRE(domega && pooled && gsig && omega && k && wl && dpooled && dk && dwl && dbl && B > 0 && H > 0 && W > 0,
               "gate_weights_bwd: bad arguments");
  const int N = H * W;
  const int64_t total = (int64_t)B * N;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  // dpre is staged in the first half of dpooled's [B][4][N] buffer?  No: stage 2 reads dpre of neighbours while
  // writing dpooled, so it needs its own storage: reuse the tail of `dpooled` is unsafe -> use domega's sibling:
  // the caller passes dpooled sized [B][6][N]; planes 4..5 hold dpre.
  float* dpre = dpooled + (int64_t)B * 4 * N;
  float* slots = dpooled + (int64_t)B * 6 * N;
  if (int rcz = zero_floats(slots, (int64_t)RSSF_GATE_SLOTS * GATE_SLOT_ELEMS, st)) return rcz;     // a kernel, not a memset node (common.hip.h)
  gate_weights_bwd1_kernel<<<grid, 256, 0, st>>>(domega, gsig, omega, wl, dpre, slots, B, N);
  int rc = check_launch("gate_weights_bwd1");
  if (rc) return rc;
  // note: dpre layout is [B][2][N] contiguous after the 4N planes of ALL batches
  const int tiles = ((H + GT - 1) / GT) * ((W + GT - 1) / GT);
  gate_weights_bwd2_kernel<<<dim3((unsigned)(B * tiles)), 256, 0, st>>>(dpre, pooled, k, dpooled, slots, B, H, W);
  rc = check_launch("gate_weights_bwd2");
  if (rc) return rc;
  gate_weights_fold_kernel<<<1, 256, 0, st>>>(slots, dk, dk_stream1 ? dk_stream1 : dk + 98, dwl, dbl);
  return check_launch("gate_weights_fold");
}

extern "C" int rssf_gate_pool_bwd(const float* dpooled, const int32_t* argmax, void* dxhat, void* dyhat, int B, int N,
                                  int C, int dtype, void* stream) {
  RSSF_REQUIRE(dpooled && argmax && dxhat && dyhat && B > 0 && N > 0 && C > 0, "gate_pool_bwd: bad arguments");
  const int64_t total = (int64_t)B * 2 * N * C;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  if ((dtype == RSSF_F32 || dtype == RSSF_BF16) && N % V == 0) {
    dim3 gv((unsigned)(((int64_t)B * 2 * (N / V) + 63) / 64));   // 64 pixel groups per block, four waves with a channel range each
    if (dtype == RSSF_F32) gate_pool_bwd_vec_kernel<float, 4><<<gv, 256, 0, st>>>(dpooled, argmax, (float*)dxhat, (float*)dyhat, B, N, C);
    else gate_pool_bwd_vec_kernel<bf16_t, 4><<<gv, 256, 0, st>>>(dpooled, argmax, (bf16_t*)dxhat, (bf16_t*)dyhat, B, N, C);
  } else if (dtype == RSSF_F32)
    gate_pool_bwd_kernel<float><<<grid, 256, 0, st>>>(dpooled, argmax, (float*)dxhat, (float*)dyhat, B, N, C);
  else if (dtype == RSSF_BF16)
    gate_pool_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>(dpooled, argmax, (bf16_t*)dxhat, (bf16_t*)dyhat, B, N, C);
  else { set_error("gate_pool_bwd: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("gate_pool_bwd");
}
