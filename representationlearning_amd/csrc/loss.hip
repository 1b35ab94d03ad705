// CGFL segmentation loss, forward and backward, on channels-last logits [B, H*W, K].
// Reference: SegmentationLossaux.forward (module/CGFL.py:201-227) -> MCTransAuxLoss.forward (losses/auxloss.py:257-305)
// -> softmax_focalloss (module/CGFL.py:72-101).  Closed form:
//     fg      = (y > 0) & (y != ignore)
//     lab_b   = [any non-fg pixel in sample b, any fg pixel in sample b, 0, 0, ...]             (one-hot sum of unique(fg_b))
//     l1_b    = sum_c 1 / (1 + exp|aux_bc - lab_bc|) / (2B)                                      (local batch size)
//     CE      = mean over valid pixels of -log softmax(logit)[y]
//     loss    = CE * [ sum_{all pixels} (1 - p[max(y,0)]) * (1 - l1_b/7) / (n_valid + B) ]        (bracket detached)
// One pass over the logits for the forward (per-sample partial sums via block reduction + atomics), a 1-block
// finalize, one pass for the backward (softmax recomputed: nothing but the logits is kept).  HBM-bound.
#include "common.hip.h"
using namespace rssf;

namespace {
constexpr int MAXK = 32;

// acc[b] = { ce_sum, n_valid, sum(1-p_y), any_fg, any_nonfg, n_bad }   n_bad: labels outside [0, K) that are not ignore_index.
// F.cross_entropy asserts on those; a kernel cannot raise, so the loss (and with it every gradient) becomes NaN instead.
// The logits of one pixel into registers.  KC > 0: the class count is a compile-time constant (6 / 7: the tiles of BASELINE.json /
// LoveDA) - fully unrolled, the row as 4-byte words where it is a whole number of them (bf16, even K), no indexed register array
// (the run-time form keeps v[] in scratch: 66 us forward / 47 us backward for 83 MB at 16 x 512 x 512 x 6).
template <typename T, int KC>
__device__ __forceinline__ void load_logits(const T* lg, int K, float* v) {
  if constexpr (KC == 0) {
    for (int k = 0; k < K; ++k) v[k] = ldf(lg + k);
  } else if constexpr (sizeof(T) == 2 && KC % 2 == 0) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(lg);
#pragma unroll
    for (int u = 0; u < KC / 2; ++u) { const uint32_t r = w[u]; v[2 * u] = __uint_as_float(r << 16); v[2 * u + 1] = __uint_as_float(r & 0xffff0000u); }
  } else {
#pragma unroll
    for (int k = 0; k < KC; ++k) v[k] = ldf(lg + k);
  }
}
template <typename T, int KC>
__device__ __forceinline__ void store_logits(T* dg, int K, const float* v) {
  if constexpr (KC == 0) {
    for (int k = 0; k < K; ++k) stf(dg + k, v[k]);
  } else if constexpr (sizeof(T) == 2 && KC % 2 == 0) {
    uint32_t* w = reinterpret_cast<uint32_t*>(dg);
#pragma unroll
    for (int u = 0; u < KC / 2; ++u) w[u] = f2bf2(v[2 * u], v[2 * u + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < KC; ++k) stf(dg + k, v[k]);
  }
}

// acc layout: RSSF_LOSS_ACC_ELEMS floats per sample, each of the six values at the head of a 128-byte line of its own.  Atomics of many
// workgroups to ONE cache line serialise (~14 ns per wave-level request, whatever the address inside the line): with the packed
// [B][6] layout the launch time grew linearly with the block count (32 blocks per sample 31 us, 64: 42, 128: 91, 256: 171 at
// 16 x 512 x 512 x 6); spread over lines 64 blocks per sample take 22.5 us (128: 23.4, 256: 27.2) - DESIGN.md lesson 54.
#define RSSF_LOSS_SS RSSF_LOSS_ACC_ELEMS
#define RSSF_LOSS_VS 32
#define RSSF_LOSS_BX 64
#define RSSF_LOSS_UNR 4
static_assert(RSSF_LOSS_ACC_ELEMS >= 6 * RSSF_LOSS_VS, "six lines per sample");
template <typename T, int KC = 0>
__global__ void __launch_bounds__(256) loss_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ acc,
                                                       int HW, int K_, int ignore_index) {
  constexpr int NV = KC ? KC : MAXK;
  const int K = KC ? KC : K_;
  const int b = blockIdx.y;
  float ce = 0.f, nv = 0.f, sm = 0.f, fg = 0.f, bg = 0.f, bad = 0.f;
  // UNR pixels of a thread in flight (compile-time class counts: the loop is a chain of memory round trips otherwise); the pixels
  // are consumed in the order of the plain loop, so the per-thread sums are the same numbers
  constexpr int UNR = KC ? RSSF_LOSS_UNR : 1;
  const int S = gridDim.x * blockDim.x;
  for (int p0 = blockIdx.x * blockDim.x + threadIdx.x; p0 < HW; p0 += UNR * S) {
    float vv[UNR][NV];
    int ys[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int p = p0 + u * S < HW ? p0 + u * S : p0;
      const int64_t pix = (int64_t)b * HW + p;
      ys[u] = (int)labels[pix];
      load_logits<T, KC>(logits + pix * K, K, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (p0 + u * S >= HW) break;
      float* v = vv[u];
      const int y = ys[u];
      const bool valid = y != ignore_index;
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (k < K) mx = fmaxf(mx, v[k]);
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (k < K) se += __expf(v[k] - mx);
      const bool oob = valid && (y < 0 || y >= K);
      if (oob) bad += 1.f;
      const int yy = (valid && !oob) ? y : 0;
      float vy = v[0];
      if constexpr (KC == 0) vy = v[yy];
      else {
#pragma unroll
        for (int k = 1; k < NV; ++k) vy = yy == k ? v[k] : vy;
      }
      const float logp = vy - mx - __logf(se);
      if (valid) { ce -= logp; nv += 1.f; }
      sm += 1.f - __expf(logp);
      if (valid && y > 0) fg = 1.f; else bg = 1.f;
    }
  }
  ce = wave_sum(ce); nv = wave_sum(nv); sm = wave_sum(sm); fg = wave_max(fg); bg = wave_max(bg); bad = wave_max(bad);
  __shared__ float red[4][6];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = ce; red[w][1] = nv; red[w][2] = sm; red[w][3] = fg; red[w][4] = bg; red[w][5] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* a = acc + b * RSSF_LOSS_SS;
    atomicAdd(a + 0, red[0][0] + red[1][0] + red[2][0] + red[3][0]);
    atomicAdd(a + 1 * RSSF_LOSS_VS, red[0][1] + red[1][1] + red[2][1] + red[3][1]);
    atomicAdd(a + 2 * RSSF_LOSS_VS, red[0][2] + red[1][2] + red[2][2] + red[3][2]);
    // flags: 0 (the zeroed buffer) or 1 - every block that sets one stores the same bits, so a plain store does (returning atomics on
    // one address from every block of a sample were the tail of this kernel: 32 us -> see DESIGN.md lesson 54)
    if (fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3])) > 0.f) a[3 * RSSF_LOSS_VS] = 1.f;
    if (fmaxf(fmaxf(red[0][4], red[1][4]), fmaxf(red[2][4], red[3][4])) > 0.f) a[4 * RSSF_LOSS_VS] = 1.f;
    if (fmaxf(fmaxf(red[0][5], red[1][5]), fmaxf(red[2][5], red[3][5])) > 0.f) a[5 * RSSF_LOSS_VS] = 1.f;
  }
}

// out[0] = loss, out[1] = gradient coefficient = bracket / n_valid  (d loss / d logit = coef * (p - onehot) on valid pixels)
__global__ void __launch_bounds__(64) loss_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ aux, float* __restrict__ out,
                                                           int B, int KA) {
  // one wave, a sample per lane (the one-thread form walked the samples serially: 20 us of dependent loads and exponentials at
  // the end of every forward pass)
  float ce = 0.f, nv = 0.f, mf = 0.f, bad = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) {
    const float* a = acc + b * RSSF_LOSS_SS;
    bad += a[5 * RSSF_LOSS_VS];
    float l1 = 0.f;
    if (KA == 1) {
      l1 = aux[b];                                           // the caller's own per-sample gamma (softmax_focalloss(.., gamma=l1), CGFL.py:72, 221)
    } else {
      for (int c = 0; c < KA; ++c) {
        const float lab = c == 0 ? a[4 * RSSF_LOSS_VS] : (c == 1 ? a[3 * RSSF_LOSS_VS] : 0.f);
        l1 += 1.f / (1.f + expf(fabsf(aux[b * KA + c] - lab)));
      }
      l1 /= (2.f * B);
    }
    ce += a[0]; nv += a[1 * RSSF_LOSS_VS];
    mf += a[2 * RSSF_LOSS_VS] * (1.f - l1 / 7.f);
  }
  ce = wave_sum(ce); nv = wave_sum(nv); mf = wave_sum(mf); bad = wave_sum(bad);
  if (threadIdx.x != 0) return;
  const float bracket = mf / (nv + (float)B);
  out[0] = bad > 0.f ? NAN : (ce / nv) * bracket;
  out[1] = bad > 0.f ? NAN : bracket / nv;
}

template <typename T, int KC = 0>
__global__ void __launch_bounds__(256) loss_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       const float* __restrict__ coef, const float* __restrict__ dloss, T* __restrict__ dlogits,
                                                       int64_t npix, int K_, int ignore_index) {
  constexpr int NV = KC ? KC : MAXK;
  const int K = KC ? KC : K_;
  const float g = coef[1] * (dloss ? dloss[0] : 1.f);
  constexpr int UNR = KC ? RSSF_LOSS_UNR : 1;
  const int64_t S = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < npix; p0 += UNR * S) {
    float vv[UNR][NV];
    int ys[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t pix = p0 + u * S < npix ? p0 + u * S : p0;
      ys[u] = (int)labels[pix];
      load_logits<T, KC>(logits + pix * K, K, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t pix = p0 + u * S;
      if (pix >= npix) break;
      float* v = vv[u];
      const int y = ys[u];
      T* dg = dlogits + pix * K;
      if (y == ignore_index) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = 0.f;
        store_logits<T, KC>(dg, K, v);
        continue;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (k < K) mx = fmaxf(mx, v[k]);
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (k < K) { v[k] = __expf(v[k] - mx); se += v[k]; }
      const float inv = 1.f / se;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (k < K) v[k] = g * (v[k] * inv - (k == y ? 1.f : 0.f));
      store_logits<T, KC>(dg, K, v);
    }
  }
}
}  // namespace

namespace {
// Image-level auxiliary head: AdaptiveAvgPool2d(1) of branch 0 -> Linear(C, KA)  (hrnet_aux.py:86-87, 99-100).  Its output only
// modulates the loss under no_grad (CGFL.py:75-97), so there is no backward.  Two small launches with a fixed summation order
// instead of ATen's mean-reduce + a hipBLASLt GEMM: the library GEMM was the one foreign kernel left inside the captured training
// step, and it did not survive a replay after a device synchronisation (its output turned to garbage / NaN: the argument buffer it
// reads at run time is filled at enqueue time, outside the graph).
constexpr int AUX_CHUNKS = 32, AUX_MAXC = 64, AUX_MAXK = 16;
template <typename T>
__global__ void __launch_bounds__(256) aux_pool_kernel(const T* __restrict__ f, float* __restrict__ partial, int HW, int C) {
  __shared__ float red[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int groups = 256 / C, c = threadIdx.x % C, g = threadIdx.x / C;          // `groups` pixel lanes x C channels
  const int per = (HW + AUX_CHUNKS - 1) / AUX_CHUNKS;
  const int p0 = chunk * per, p1 = p0 + per < HW ? p0 + per : HW;
  float acc = 0.f;
  if (g < groups)
    for (int p = p0 + g; p < p1; p += groups) acc += ldf(f + ((int64_t)b * HW + p) * C + c);
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < groups; ++k) t += red[k * C + threadIdx.x];
    partial[((int64_t)b * AUX_CHUNKS + chunk) * C + threadIdx.x] = t;
  }
}
// Vector form (C a multiple of the 16-byte vector): a thread owns V channels, 256 / (C / V) pixel lanes per block, four loads in flight.
// (The scalar form loads one 2-byte element per thread and iteration in a chain of 64 round trips: 29 us for the 16.8 MB of the
// benchmark's feature map.)
template <typename T>
__global__ void __launch_bounds__(256) aux_pool_vec_kernel(const T* __restrict__ f, float* __restrict__ partial, int HW, int C) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[256][V + 1];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cpr = C / V, groups = 256 / cpr, cv = threadIdx.x % cpr, g = threadIdx.x / cpr;
  const int per = (HW + AUX_CHUNKS - 1) / AUX_CHUNKS;
  const int p0 = chunk * per, p1 = p0 + per < HW ? p0 + per : HW;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  if (g < groups) {
    const T* base = f + (int64_t)b * HW * C + cv * V;
    constexpr int RF = 4;
    for (int p = p0 + g; p < p1; p += RF * groups) {
      Vec<T> v[RF];
#pragma unroll
      for (int u = 0; u < RF; ++u) { const int q = p + u * groups < p1 ? p + u * groups : p; v[u].load(base + (int64_t)q * C); }
#pragma unroll
      for (int u = 0; u < RF; ++u)
        if (p + u * groups < p1) {
#pragma unroll
          for (int e = 0; e < V; ++e) acc[e] += v[u].get(e);
        }
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x, cvv = c / V, e = c % V;
    float t = 0.f;
    for (int k = 0; k < groups; ++k) t += red[k * cpr + cvv][e];
    partial[((int64_t)b * AUX_CHUNKS + chunk) * C + c] = t;
  }
}
__global__ void aux_linear_kernel(const float* __restrict__ partial, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ out, int HW, int C, int K) {
  __shared__ float mean[AUX_MAXC];
  const int b = blockIdx.x;
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < AUX_CHUNKS; ++k) t += partial[((int64_t)b * AUX_CHUNKS + k) * C + threadIdx.x];
    mean[threadIdx.x] = t / (float)HW;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    float t = bias ? bias[threadIdx.x] : 0.f;
    for (int c = 0; c < C; ++c) t += w[threadIdx.x * C + c] * mean[c];
    out[b * K + threadIdx.x] = t;
  }
}
}  // namespace

extern "C" int64_t rssf_aux_head_workspace_elems(int B, int C) { return (int64_t)B * AUX_CHUNKS * C; }

extern "C" int rssf_aux_head_fwd(const void* feat, const float* weight, const float* bias, float* workspace, float* out, int B, int HW,
                                 int C, int K, int dtype, void* stream) {
  RSSF_REQUIRE(feat && weight && workspace && out && B > 0 && HW > 0 && C > 0 && C <= AUX_MAXC && K > 0 && K <= AUX_MAXK,
               "aux_head_fwd: bad arguments (C <= %d, K <= %d)", AUX_MAXC, AUX_MAXK);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(AUX_CHUNKS, (unsigned)B);
  if (dtype == RSSF_F32) {
    if (C % 4 == 0) aux_pool_vec_kernel<float><<<grid, 256, 0, st>>>((const float*)feat, workspace, HW, C);
    else aux_pool_kernel<float><<<grid, 256, 0, st>>>((const float*)feat, workspace, HW, C);
  } else if (dtype == RSSF_BF16) {
    if (C % 8 == 0) aux_pool_vec_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)feat, workspace, HW, C);
    else aux_pool_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)feat, workspace, HW, C);
  }
  else { set_error("aux_head_fwd: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  int rc = check_launch("aux_head_pool");
  if (rc) return rc;
  aux_linear_kernel<<<(unsigned)B, 64, 0, st>>>(workspace, weight, bias, out, HW, C, K);
  return check_launch("aux_head_linear");
}

extern "C" int rssf_cgfl_loss_fwd(const void* logits, const int64_t* labels, const float* aux, float* acc, float* out, int B, int HW, int K,
                                  int KA, int ignore_index, int deterministic, int dtype, void* stream) {
  RSSF_REQUIRE(logits && labels && aux && acc && out && B > 0 && HW > 0 && K > 0 && K <= MAXK && KA >= 1, "cgfl_loss_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (int rcz = zero_floats(acc, (int64_t)RSSF_LOSS_SS * B, st)) return rcz;      // a kernel, not a memset node (common.hip.h)
  int bx = (HW + 255) / 256;
  if (bx > RSSF_LOSS_BX) bx = RSSF_LOSS_BX;                // blocks per sample (see RSSF_LOSS_SS above)
  if (deterministic) bx = 1;           // one block per sample: wave shuffles + an ordered 4-way sum, a single add into the zeroed acc
  dim3 grid((unsigned)bx, (unsigned)B);
#define RSSF_LOSS_FWD(Tt, KCv) loss_fwd_kernel<Tt, KCv><<<grid, 256, 0, st>>>((const Tt*)logits, labels, acc, HW, K, ignore_index)
  if (dtype == RSSF_F32) { if (K == 6) RSSF_LOSS_FWD(float, 6); else if (K == 7) RSSF_LOSS_FWD(float, 7); else RSSF_LOSS_FWD(float, 0); }
  else if (dtype == RSSF_BF16) { if (K == 6) RSSF_LOSS_FWD(bf16_t, 6); else if (K == 7) RSSF_LOSS_FWD(bf16_t, 7); else RSSF_LOSS_FWD(bf16_t, 0); }
  else { set_error("cgfl_loss_fwd: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
#undef RSSF_LOSS_FWD
  int rc = check_launch("cgfl_loss_fwd");
  if (rc) return rc;
  loss_finalize_kernel<<<1, 64, 0, st>>>(acc, aux, out, B, KA);
  return check_launch("cgfl_loss_finalize");
}

extern "C" int rssf_cgfl_loss_bwd(const void* logits, const int64_t* labels, const float* out, const float* dloss, void* dlogits, int B,
                                  int HW, int K, int ignore_index, int dtype, void* stream) {
  RSSF_REQUIRE(logits && labels && out && dlogits && B > 0 && HW > 0 && K > 0 && K <= MAXK, "cgfl_loss_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t npix = (int64_t)B * HW;
  int64_t blocks = (npix + 255) / 256;
  if (blocks > 4096) blocks = 4096;
#define RSSF_LOSS_BWD(Tt, KCv) \
  loss_bwd_kernel<Tt, KCv><<<(unsigned)blocks, 256, 0, st>>>((const Tt*)logits, labels, out, dloss, (Tt*)dlogits, npix, K, ignore_index)
  if (dtype == RSSF_F32) { if (K == 6) RSSF_LOSS_BWD(float, 6); else if (K == 7) RSSF_LOSS_BWD(float, 7); else RSSF_LOSS_BWD(float, 0); }
  else if (dtype == RSSF_BF16) { if (K == 6) RSSF_LOSS_BWD(bf16_t, 6); else if (K == 7) RSSF_LOSS_BWD(bf16_t, 7); else RSSF_LOSS_BWD(bf16_t, 0); }
  else { set_error("cgfl_loss_bwd: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
#undef RSSF_LOSS_BWD
  return check_launch("cgfl_loss_bwd");
}
