// LayerNorm over the channel axis of channels-last tokens [rows, C] (eps 1e-6).
// Reference: modules/MTFM.py:64,80-81 (nn.LayerNorm(C, eps=1e-6)), applied at :107 (norm1 on both
// streams) and :109 (norm2).  HBM-bound: one read (+ one write); 16-byte lane accesses, G lanes per row.
#include "common.hip.h"
using namespace rssf;

namespace {

// ---- forward ------------------------------------------------------------------------------------------
// G lanes (a power of two) serve one row, the first C / VEC of them hold VEC channels each - C = 48 bf16 is six live lanes of
// eight (Large: the one-thread-per-row fallback took 117 us forward / 868 us backward per launch at 4 x 1024 x 1024);
// fallback (C not a multiple of VEC, or more than 16 lanes): one thread per row, scalar loop.
// sum over the G lanes of a row's lane group (G = 1, 2, 4, 8, 16 consecutive lanes inside one 16-lane DPP row), result in every lane
// of the group: pure VALU (the __shfl_xor form is ds_bpermute - an LDS instruction with ~100 cycles of latency - twice per row sum
// on the dependency chain of every row)
template <int G> __device__ __forceinline__ float group_sum(float v) {
  if (G >= 2) v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
  if (G >= 4) v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_mov<0x141>(v);       // row_half_mirror
  if (G >= 16) v += dpp_mov<0x140>(v);      // row_mirror
  return v;
}
// sum over the lanes of a wave that own the same channels (same lane % G), result in every such lane
template <int G> __device__ __forceinline__ float same_sub_sum(float v) {
  if (G <= 8) v += dpp_mov<0x128>(v);       // row_ror:8
  if (G <= 4) v += dpp_mov<0x124>(v);       // row_ror:4
  if (G <= 2) v += dpp_mov<0x122>(v);       // row_ror:2
  if (G <= 1) v += dpp_mov<0x121>(v);       // row_ror:1
  return rows_reduce<OpSum>(v);             // the four 16-lane rows
}

template <typename T, int G>
__global__ void __launch_bounds__(256) ln_fwd_vec(const T* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, T* __restrict__ y,
                                                  float* __restrict__ stats, int64_t rows, int C, float eps) {
  constexpr int VEC = Vec<T>::N;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / G;
  const int sub = (int)(gid % G);
  const bool ok = row < rows && sub * VEC < C;
  Vec<T> v;
  float s = 0.f;
  if (ok) {
    v.load(x + row * C + sub * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += v.get(i);
  }
  s = group_sum<G>(s);
  const float mean = s / C;
  float q = 0.f;
  if (ok) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { float d = v.get(i) - mean; q += d * d; }
  }
  q = group_sum<G>(q);
  const float rstd = rsqrtf(q / C + eps);
  if (!ok) return;
  if (stats && sub == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
  if (y) {
    Vec<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int c = sub * VEC + i;
      o.set(i, (v.get(i) - mean) * rstd * gamma[c] + beta[c]);
    }
    o.store(y + row * C + sub * VEC);
  }
}

// One wave per row, NP 16-byte vectors per lane: the Mix-Transformer widths of the SCD CAM path (C = 320, 512; up to 64 * NP * VEC)
template <typename T, int NP>
__global__ void __launch_bounds__(256) ln_fwd_wave(const T* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, T* __restrict__ y,
                                                   float* __restrict__ stats, int64_t rows, int C, float eps) {
  constexpr int VEC = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;                                  // wave-uniform
  Vec<T> v[NP];
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c0 = (lane + 64 * p) * VEC;
    if (c0 < C) {
      v[p].load(x + row * C + c0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s += v[p].get(i);
    }
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p)
    if ((lane + 64 * p) * VEC < C) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float d = v[p].get(i) - mean; q += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
  if (stats && lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
  if (y) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c0 = (lane + 64 * p) * VEC;
      if (c0 < C) {
        Vec<T> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.set(i, (v[p].get(i) - mean) * rstd * gamma[c0 + i] + beta[c0 + i]);
        o.store(y + row * C + c0);
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) ln_fwd_scalar(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int64_t rows, int C, float eps) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const T* xr = x + row * C;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += ldf(xr + c);
  const float mean = s / C;
  float q = 0.f;
  for (int c = 0; c < C; ++c) { float d = ldf(xr + c) - mean; q += d * d; }
  const float rstd = rsqrtf(q / C + eps);
  if (stats) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
  if (y) for (int c = 0; c < C; ++c) stf(y + row * C + c, (ldf(xr + c) - mean) * rstd * gamma[c] + beta[c]);
}

// ---- backward -------------------------------------------------------------------------------------------
// dx = rstd * (g - mean_c(g) - xhat * mean_c(g*xhat)),  g = dy*gamma ;  dgamma += sum_rows dy*xhat ; dbeta += sum dy
// G lanes per row (VEC channels each, register-resident); each lane keeps its own dgamma/dbeta partials over a
// grid-stride loop of rows, flushed once through LDS -> one global atomicAdd per channel per block.
#ifndef RSSF_LN_BWD_THREADS
#define RSSF_LN_BWD_THREADS 512      // 8 waves per CU on the 256-block grid (256: 16.8 us, 512: 15.7 us, 1024: 37.8 us at 262 144 x 32 bf16)
#endif
template <typename T, int G>
__global__ void __launch_bounds__(RSSF_LN_BWD_THREADS) ln_bwd_vec(const T* __restrict__ dy, const T* __restrict__ x,
                                                  const float* __restrict__ stats, const float* __restrict__ gamma,
                                                  const T* __restrict__ dx_add, T* __restrict__ dx,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                  int64_t rows, int C) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sg = reinterpret_cast<float*>(smem_raw);
  float* sb = sg + C;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sg[i] = 0.f;
  __syncthreads();
  const bool live = (int)(threadIdx.x % G) * VEC < C;       // lanes past C / VEC of a group read lane 0's channels and contribute nothing
  const int sub = live ? threadIdx.x % G : 0;
  float gam[VEC], ag[VEC], ab[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { gam[i] = gamma[sub * VEC + i]; ag[i] = 0.f; ab[i] = 0.f; }
  const int64_t rows_per_pass = (int64_t)gridDim.x * (blockDim.x / G);
  const int64_t niter = (rows + rows_per_pass - 1) / rows_per_pass;
  const int64_t row0 = (int64_t)blockIdx.x * (blockDim.x / G) + threadIdx.x / G;
  auto one = [&](int64_t row, bool ok, const Vec<T>& vx, const Vec<T>& vd, const Vec<T>& va, float mean, float rstd) {
    float s1 = 0.f, s2 = 0.f, xh[VEC], g[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      xh[i] = (vx.get(i) - mean) * rstd;
      const float d = ok ? vd.get(i) : 0.f;
      g[i] = d * gam[i];
      s1 += g[i]; s2 += g[i] * xh[i];
      ag[i] += d * xh[i]; ab[i] += d;
    }
    s1 = group_sum<G>(s1); s2 = group_sum<G>(s2);
    s1 /= C; s2 /= C;
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      o[i] = rstd * (g[i] - s1 - xh[i] * s2);
      if (dx_add) o[i] += va.get(i);
    }
    if (ok) {
      Vec<T> w;
      w.set_all(o);
      w.store(dx + row * C + sub * VEC);
    }
  };
  // RF rows in flight per lane group (the loop is a chain of memory round trips: 2 in flight left it latency-bound at
  // 2.5 TB/s); loads issued unconditionally (a row past the end re-reads row 0 and is discarded)
  constexpr int RF = 4;
  for (int64_t it = 0; it < niter; it += RF) {
    int64_t rr[RF], qq[RF];
    bool ok[RF];
    Vec<T> vx[RF], vd[RF], va[RF];
    float mm[RF], ss[RF];
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      rr[u] = (it + u) * rows_per_pass + row0;
      const bool inr = (it + u < niter) && rr[u] < rows;
      ok[u] = inr && live;
      qq[u] = inr ? rr[u] : 0;
      vx[u].load(x + qq[u] * C + sub * VEC); vd[u].load(dy + qq[u] * C + sub * VEC);
      if (dx_add) va[u].load(dx_add + qq[u] * C + sub * VEC);
      mm[u] = stats[qq[u] * 2]; ss[u] = stats[qq[u] * 2 + 1];
    }
#pragma unroll
    for (int u = 0; u < RF; ++u) one(rr[u], ok[u], vx[u], vd[u], va[u], mm[u], ss[u]);
  }
  // lanes that own the same channels (same `sub`) inside a wave are folded by shuffles before the LDS atomics
#pragma unroll
  for (int i = 0; i < VEC; ++i) { ag[i] = same_sub_sum<G>(ag[i]); ab[i] = same_sub_sum<G>(ab[i]); }
  if ((threadIdx.x & 63) < G && live) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { atomicAdd(&sg[sub * VEC + i], ag[i]); atomicAdd(&sb[sub * VEC + i], ab[i]); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += blockDim.x) { atomicAdd(&dgamma[i], sg[i]); atomicAdd(&dbeta[i], sb[i]); }
}

template <typename T>
__global__ void __launch_bounds__(256) ln_bwd_scalar(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const T* __restrict__ dx_add, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int64_t rows, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sg = reinterpret_cast<float*>(smem_raw);   // [C] dgamma, [C] dbeta
  float* sb = sg + C;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sg[i] = 0.f;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += stride) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    const T* xr = x + row * C;
    const T* gr = dy + row * C;
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float xh = (ldf(xr + c) - mean) * rstd;
      const float d = ldf(gr + c);
      const float g = d * gamma[c];
      s1 += g; s2 += g * xh;
      atomicAdd(&sg[c], d * xh);
      atomicAdd(&sb[c], d);
    }
    s1 /= C; s2 /= C;
    for (int c = 0; c < C; ++c) {
      const float xh = (ldf(xr + c) - mean) * rstd;
      float v = rstd * (ldf(gr + c) * gamma[c] - s1 - xh * s2);
      if (dx_add) v += ldf(dx_add + row * C + c);
      stf(dx + row * C + c, v);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    atomicAdd(&dgamma[i], sg[i]);
    atomicAdd(&dbeta[i], sb[i]);
  }
}

int lane_group(int n) {            // the smallest power of two >= n lanes (0: more than 16, the scalar kernels run)
  int g = 1;
  while (g < n) g <<= 1;
  return g <= 16 ? g : 0;
}

#ifndef RSSF_LN_BWD_BLOCKS
#define RSSF_LN_BWD_BLOCKS 256
#endif
template <typename T>
int ln_bwd_launch(const void* dy, const void* x, const float* stats, const float* gamma, const void* dx_add, void* dx,
                  float* dgamma, float* dbeta, int64_t rows, int C, hipStream_t st) {
  constexpr int VEC = Vec<T>::N;
  const int G = (C % VEC == 0) ? lane_group(C / VEC) : 0;
  const size_t sh = 2 * C * sizeof(float);
  const T* a = (const T*)dy; const T* b = (const T*)x; const T* c = (const T*)dx_add; T* d = (T*)dx;
  // 256 blocks: every block ends with 2C same-address global atomics (~40 ns each, serialised per address)
  auto grid = [&](int g) { int64_t n = (rows * g + RSSF_LN_BWD_THREADS - 1) / RSSF_LN_BWD_THREADS; return dim3((unsigned)(n > RSSF_LN_BWD_BLOCKS ? RSSF_LN_BWD_BLOCKS : n)); };
  switch (G) {
    case 1: ln_bwd_vec<T, 1><<<grid(1), RSSF_LN_BWD_THREADS, sh, st>>>(a, b, stats, gamma, c, d, dgamma, dbeta, rows, C); break;
    case 2: ln_bwd_vec<T, 2><<<grid(2), RSSF_LN_BWD_THREADS, sh, st>>>(a, b, stats, gamma, c, d, dgamma, dbeta, rows, C); break;
    case 4: ln_bwd_vec<T, 4><<<grid(4), RSSF_LN_BWD_THREADS, sh, st>>>(a, b, stats, gamma, c, d, dgamma, dbeta, rows, C); break;
    case 8: ln_bwd_vec<T, 8><<<grid(8), RSSF_LN_BWD_THREADS, sh, st>>>(a, b, stats, gamma, c, d, dgamma, dbeta, rows, C); break;
    case 16: ln_bwd_vec<T, 16><<<grid(16), RSSF_LN_BWD_THREADS, sh, st>>>(a, b, stats, gamma, c, d, dgamma, dbeta, rows, C); break;
    default: ln_bwd_scalar<T><<<grid(1), 256, sh, st>>>(a, b, stats, gamma, c, d, dgamma, dbeta, rows, C); break;
  }
  return check_launch("layernorm_bwd");
}

template <typename T>
int ln_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* stats, int64_t rows, int C,
                  float eps, hipStream_t st) {
  constexpr int VEC = Vec<T>::N;
  const T* xp = (const T*)x; T* yp = (T*)y;
  const int G = (C % VEC == 0) ? lane_group(C / VEC) : 0;
  auto grid = [&](int g) { return dim3((unsigned)((rows * g + 255) / 256)); };
  switch (G) {
    case 1: ln_fwd_vec<T, 1><<<grid(1), 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps); break;
    case 2: ln_fwd_vec<T, 2><<<grid(2), 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps); break;
    case 4: ln_fwd_vec<T, 4><<<grid(4), 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps); break;
    case 8: ln_fwd_vec<T, 8><<<grid(8), 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps); break;
    case 16: ln_fwd_vec<T, 16><<<grid(16), 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps); break;
    default: {
      const int np = (C % VEC == 0) ? (C / VEC + 63) / 64 : 0;      // wide rows: a wave per row
      const dim3 wg((unsigned)((rows + 3) / 4));
      if (np == 1) ln_fwd_wave<T, 1><<<wg, 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps);
      else if (np == 2) ln_fwd_wave<T, 2><<<wg, 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps);
      else if (np >= 3 && np <= 4) ln_fwd_wave<T, 4><<<wg, 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps);
      else ln_fwd_scalar<T><<<grid(1), 256, 0, st>>>(xp, gamma, beta, yp, stats, rows, C, eps);
      break;
    }
  }
  return check_launch("layernorm_fwd");
}
}  // namespace

extern "C" int rssf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                  int64_t rows, int C, float eps, int dtype, void* stream) {
  RSSF_REQUIRE(x && rows > 0 && C > 0 && C <= 1024, "layernorm_fwd: bad shape rows=%lld C=%d", (long long)rows, C);
  RSSF_REQUIRE(y == nullptr || (gamma && beta), "layernorm_fwd: gamma/beta required when y is requested");
  RSSF_REQUIRE(y || stats, "layernorm_fwd: nothing to compute");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return ln_fwd_launch<float>(x, gamma, beta, y, stats, rows, C, eps, st);
  if (dtype == RSSF_BF16) return ln_fwd_launch<bf16_t>(x, gamma, beta, y, stats, rows, C, eps, st);
  set_error("layernorm_fwd: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_layernorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma,
                                  const void* dx_add, void* dx, float* dgamma, float* dbeta, int64_t rows, int C,
                                  int dtype, void* stream) {
  RSSF_REQUIRE(dy && x && stats && gamma && dx && dgamma && dbeta && rows > 0 && C > 0 && C <= 1024,
               "layernorm_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return ln_bwd_launch<float>(dy, x, stats, gamma, dx_add, dx, dgamma, dbeta, rows, C, st);
  if (dtype == RSSF_BF16) return ln_bwd_launch<bf16_t>(dy, x, stats, gamma, dx_add, dx, dgamma, dbeta, rows, C, st);
  set_error("layernorm_bwd: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}
