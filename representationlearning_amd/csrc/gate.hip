// Saliency gate of InterlacedPoolAttention2 (modules/multihead_isa_pool_attention.py:148-167) with the two
// SpatialAttention(7) gates (:101-115).  Bug-compatible with the reference's (B,N,C)->view(B,C,H,W)
// reinterpretation: for flat offset f = n*C + c inside one image, view-pixel p = f mod N, view-channel
// c' = f div N.  All maps are tiny ([B][k][N] fp32) and stay L2 resident;
// the only full-tensor traffic is one read of x and y (pool) and one read-modify-write (pool backward).
#include "common.cuh"
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

// ---- weights: 7x7 conv (2->1, pad 3, no bias) + sigmoid per stream, 1x1 conv 2->2 + softmax over the 2 streams
__global__ void __launch_bounds__(256) gate_weights_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ k,
                                                               const float* __restrict__ wl, const float* __restrict__ bl,
                                                               float* __restrict__ gsig, float* __restrict__ omega,
                                                               float* __restrict__ logits, int B, int H, int W) {
  __shared__ float sk[196];
  for (int i = threadIdx.x; i < 196; i += blockDim.x) sk[i] = k[i];
  __syncthreads();
  const int N = H * W;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * N) return;
  const int p = (int)(gid % N), b = (int)(gid / N);
  const int h = p / W, w = p % W;
  float g[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float acc = 0.f;
    for (int ch = 0; ch < 2; ++ch) {
      const float* m = pooled + ((int64_t)b * 4 + 2 * s + ch) * N;
      const float* kk = sk + (s * 2 + ch) * 49;
      for (int u = 0; u < 7; ++u) {
        const int hh = h + u - 3;
        if (hh < 0 || hh >= H) continue;
        for (int v = 0; v < 7; ++v) {
          const int ww = w + v - 3;
          if (ww < 0 || ww >= W) continue;
          acc += kk[u * 7 + v] * m[hh * W + ww];
        }
      }
    }
    g[s] = sigmoidf(acc);
  }
  const float l0 = wl[0] * g[0] + wl[1] * g[1] + bl[0];
  const float l1 = wl[2] * g[0] + wl[3] * g[1] + bl[1];
  const float m = fmaxf(l0, l1);
  const float e0 = __expf(l0 - m), e1 = __expf(l1 - m);
  const float inv = 1.f / (e0 + e1);
  const int64_t o = (int64_t)b * 2 * N;
  gsig[o + p] = g[0]; gsig[o + N + p] = g[1];
  omega[o + p] = e0 * inv; omega[o + N + p] = e1 * inv;
  if (logits) { logits[o + p] = l0; logits[o + N + p] = l1; }
}

// ---- weights backward, stage 1: domega -> dpre (gradient at the 7x7 conv outputs, pre-sigmoid); dwl/dbl --------
__global__ void __launch_bounds__(256) gate_weights_bwd1_kernel(const float* __restrict__ domega, const float* __restrict__ gsig,
                                                                const float* __restrict__ omega, const float* __restrict__ wl,
                                                                float* __restrict__ dpre, float* __restrict__ dwl,
                                                                float* __restrict__ dbl, int B, int N) {
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
  if (threadIdx.x < 4) atomicAdd(&dwl[threadIdx.x], red[threadIdx.x]);          // one global atomic per block
  else if (threadIdx.x < 6) atomicAdd(&dbl[threadIdx.x - 4], red[threadIdx.x]);
}

// stage 2: dpooled = conv^T(dpre, k) ; dk += sum_pixels dpre * shifted(pooled)
__global__ void __launch_bounds__(256) gate_weights_bwd2_kernel(const float* __restrict__ dpre, const float* __restrict__ pooled,
                                                                const float* __restrict__ k, float* __restrict__ dpooled,
                                                                float* __restrict__ dk, int B, int H, int W) {
  __shared__ float sk[196];
  __shared__ float sdk[196];
  for (int i = threadIdx.x; i < 196; i += blockDim.x) { sk[i] = k[i]; sdk[i] = 0.f; }
  __syncthreads();
  const int N = H * W;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = gid < (int64_t)B * N;
  const int p = ok ? (int)(gid % N) : 0, b = ok ? (int)(gid / N) : 0;
  const int h = p / W, w = p % W;
  for (int s = 0; s < 2; ++s) {
    const float* dp = dpre + ((int64_t)b * 2 + s) * N;
    const float mine = ok ? dp[p] : 0.f;
    for (int ch = 0; ch < 2; ++ch) {
      const float* m = pooled + ((int64_t)b * 4 + 2 * s + ch) * N;
      const float* kk = sk + (s * 2 + ch) * 49;
      float acc = 0.f;
      for (int u = 0; u < 7; ++u) {
        for (int v = 0; v < 7; ++v) {
          // forward: out[h,w] += k[u,v] * in[h+u-3, w+v-3]
          const int ho = h - (u - 3), wo = w - (v - 3);          // output pixel that read THIS input through tap (u,v)
          if (ok && ho >= 0 && ho < H && wo >= 0 && wo < W) acc += kk[u * 7 + v] * dp[ho * W + wo];
          const int hi = h + u - 3, wi = w + v - 3;              // input pixel THIS output read through tap (u,v)
          float t = 0.f;
          if (ok && hi >= 0 && hi < H && wi >= 0 && wi < W) t = mine * m[hi * W + wi];
          t = wave_sum(t);
          if ((threadIdx.x & 63) == 0) atomicAdd(&sdk[(s * 2 + ch) * 49 + u * 7 + v], t);
        }
      }
      if (ok) dpooled[((int64_t)b * 4 + 2 * s + ch) * N + p] = acc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 196; i += blockDim.x) atomicAdd(&dk[i], sdk[i]);
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

}  // namespace

extern "C" int rssf_gate_pool_fwd(const void* x, const void* y, const float* stats_x, const float* stats_y,
                                  const float* gamma, const float* beta, float* pooled, int32_t* argmax, int B, int N,
                                  int C, int dtype, void* stream) {
  RSSF_REQUIRE(x && y && stats_x && stats_y && gamma && beta && pooled && argmax && B > 0 && N > 0 && C > 0,
               "gate_pool_fwd: bad arguments");
  const int64_t total = (int64_t)B * 2 * N;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32)
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
  const int64_t total = (int64_t)B * H * W;
  gate_weights_fwd_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      pooled, k, wl, bl, gsig, omega, logits, B, H, W);
  return check_launch("gate_weights_fwd");
}

extern "C" int rssf_gate_weights_bwd(const float* domega, const float* pooled, const float* gsig, const float* omega,
                                     const float* k, const float* wl, float* dpooled, float* dk, float* dwl, float* dbl,
                                     int B, int H, int W, void* stream) {
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
  gate_weights_bwd1_kernel<<<grid, 256, 0, st>>>(domega, gsig, omega, wl, dpre, dwl, dbl, B, N);
  int rc = check_launch("gate_weights_bwd1");
  if (rc) return rc;
  // note: dpre layout is [B][2][N] contiguous after the 4N planes of ALL batches
  gate_weights_bwd2_kernel<<<grid, 256, 0, st>>>(dpre, pooled, k, dpooled, dk, B, H, W);
  return check_launch("gate_weights_bwd2");
}

extern "C" int rssf_gate_pool_bwd(const float* dpooled, const int32_t* argmax, void* dxhat, void* dyhat, int B, int N,
                                  int C, int dtype, void* stream) {
  RSSF_REQUIRE(dpooled && argmax && dxhat && dyhat && B > 0 && N > 0 && C > 0, "gate_pool_bwd: bad arguments");
  const int64_t total = (int64_t)B * 2 * N * C;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32)
    gate_pool_bwd_kernel<float><<<grid, 256, 0, st>>>(dpooled, argmax, (float*)dxhat, (float*)dyhat, B, N, C);
  else if (dtype == RSSF_BF16)
    gate_pool_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>(dpooled, argmax, (bf16_t*)dxhat, (bf16_t*)dyhat, B, N, C);
  else { set_error("gate_pool_bwd: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("gate_pool_bwd");
}
