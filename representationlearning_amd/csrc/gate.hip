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
// STATS = G > 0 (rssf_ln_gate_pool_fwd): the LayerNorm statistics are FORMED here instead of read.  With N % C == 0 a token row is the
// V-element pieces of G = C / V consecutive, G-aligned lanes at EVERY view-channel (the element offset cp * N + p0 keeps p0 mod C),
// every (view-channel, lane group) is a different token and every token occurs exactly once in the grid: the row sums are quad /
// half-row DPP folds (norm.hip::ln_fwd_vec's arithmetic, statement by statement), lane 0 of a group writes {mean, rstd}
// for the attention kernels - the two statistics-only LayerNorm passes over x and y (2 x 7.3 us per block) are not launched.
template <int G> __device__ __forceinline__ float gate_group_sum(float v) {
  if (G >= 2) v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
  if (G >= 4) v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_mov<0x141>(v);       // row_half_mirror
  if (G >= 16) v += dpp_mov<0x140>(v);      // row_mirror
  return v;
}
template <typename T, int PARTS, int G = 0>
__global__ void __launch_bounds__(64 * PARTS) gate_pool_fwd_vec_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                                       float* __restrict__ stx, float* __restrict__ sty,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                       float* __restrict__ pooled, int32_t* __restrict__ argmax,
                                                                       int B, int N, int C, float eps) {
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
  float2* st = reinterpret_cast<float2*>((s == 0 ? stx : sty) + (int64_t)b * N * 2);
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
    float2 ms;
    if constexpr (G > 0) {
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) sm += v.get(i);
      sm = gate_group_sum<G>(sm);
      const float mean = sm / C;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) { const float d = v.get(i) - mean; q += d * d; }
      q = gate_group_sum<G>(q);
      ms = float2{mean, rsqrtf(q / C + eps)};
      if (live && (lane & (G - 1)) == 0) st[n] = ms;
    } else {
      ms = st[n];
    }
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

// ---- the two 7x7 convolutions (and their backward) run on pixel tiles staged with their halo, zero outside the image, in LDS ---
constexpr int GATE_SLOT_ELEMS = 196 + 6;            // dk [2][2][7][7], dwl [2][2], dbl [2]
// 32 x 32-pixel tiles with their halo (3 rows above / below, 4 columns left / right: 16-byte aligned rows) for the 7x7 kernels below
constexpr int G2T = 32, G2H = G2T + 6, G2W = G2T + 8, G2LD = 44, G2PLANE = G2H * G2LD;
// stages NPL planes of one tile: tile row i = image row h0 - 3 + i, tile column j = image column w0 - 4 + j, zero outside the image.
// All of a thread's loads are in flight together (a loop of load -> LDS store round trips cost 6 of the kernel's 15 us).
template <int NPL, typename F>
__device__ __forceinline__ void gate2_stage(float* sp, int tid, int h0, int w0, int H, int W, F plane_of) {
  if ((W & 3) == 0) {
    constexpr int UNITS = NPL * G2H * (G2W / 4), NIT = (UNITS + 255) / 256;
    f32x4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * 256;
      const int pl = e / (G2H * (G2W / 4)), rem = e % (G2H * (G2W / 4)), r = rem / (G2W / 4), c = (rem % (G2W / 4)) * 4;
      const int hh = h0 - 3 + r, ww = w0 - 4 + c;
      v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < UNITS && hh >= 0 && hh < H && ww >= 0 && ww < W) v[it] = *reinterpret_cast<const f32x4*>(plane_of(pl) + (int64_t)hh * W + ww);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * 256;
      const int pl = e / (G2H * (G2W / 4)), rem = e % (G2H * (G2W / 4)), r = rem / (G2W / 4), c = (rem % (G2W / 4)) * 4;
      if (e < UNITS) *reinterpret_cast<f32x4*>(sp + pl * G2PLANE + r * G2LD + c) = v[it];
    }
  } else {
    for (int e = tid; e < NPL * G2H * G2W; e += 256) {
      const int pl = e / (G2H * G2W), rem = e % (G2H * G2W), r = rem / G2W, c = rem % G2W;
      const int hh = h0 - 3 + r, ww = w0 - 4 + c;
      sp[pl * G2PLANE + r * G2LD + c] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? plane_of(pl)[(int64_t)hh * W + ww] : 0.f;
    }
  }
}

// ---- weights: 7x7 conv (2->1, pad 3, no bias) + sigmoid per stream, 1x1 conv 2->2 + softmax over the 2 streams
// Round 5: 32 x 32 tiles, a thread owns FOUR pixels of a row - per kernel row and map it reads the 12 values under them once (three
// 16-byte LDS reads for 28 FMAs; a thread per pixel read LDS once per FMA) and the tile's loads are all in flight together.
__global__ void __launch_bounds__(256) gate_weights_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ k,
                                                               const float* __restrict__ wl, const float* __restrict__ bl,
                                                               float* __restrict__ gsig, float* __restrict__ omega,
                                                               float* __restrict__ logits, int B, int H, int W) {
  __shared__ __attribute__((aligned(16))) float sp[4 * G2PLANE];
  const int N = H * W;
  const int tw = (W + G2T - 1) / G2T, th = (H + G2T - 1) / G2T;
  const int b = blockIdx.x / (tw * th), t = blockIdx.x % (tw * th);
  const int h0 = (t / tw) * G2T, w0 = (t % tw) * G2T;
  const int tid = threadIdx.x;
  gate2_stage<4>(sp, tid, h0, w0, H, W, [&](int pl) { return pooled + ((int64_t)b * 4 + pl) * N; });
  __syncthreads();
  const int row = tid >> 3, c4 = (tid & 7) * 4;
  const int h = h0 + row, w = w0 + c4;
  if (h >= H || w >= W) return;
  float g[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        // out[h, w] += k[u, v] * in[h + u - 3, w + v - 3]: tile row row + u, tile column c4 + j + v + 1
        const float* m = sp + (2 * s + ch) * G2PLANE + (row + u) * G2LD + c4;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(m), x1 = *reinterpret_cast<const f32x4*>(m + 4), x2 = *reinterpret_cast<const f32x4*>(m + 8);
        const float x[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
#pragma unroll
        for (int v = 0; v < 7; ++v) {
          const float kv = k[((s * 2 + ch) * 7 + u) * 7 + v];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(kv, x[j + v + 1], acc[j]);
        }
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) g[s][j] = sigmoidf(acc[j]);
  }
  float o0[4], o1[4], l0[4], l1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    l0[j] = wl[0] * g[0][j] + wl[1] * g[1][j] + bl[0];
    l1[j] = wl[2] * g[0][j] + wl[3] * g[1][j] + bl[1];
    const float m = fmaxf(l0[j], l1[j]);
    const float e0 = __expf(l0[j] - m), e1 = __expf(l1[j] - m);
    const float inv = 1.f / (e0 + e1);
    o0[j] = e0 * inv; o1[j] = e1 * inv;
  }
  const int64_t o = (int64_t)b * 2 * N + (int64_t)h * W + w;
  auto put = [&](float* dst, const float (&v)[4]) {
    if ((W & 3) == 0) *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (w + j < W) dst[j] = v[j];
    }
  };
  put(gsig + o, g[0]); put(gsig + o + N, g[1]);
  put(omega + o, o0); put(omega + o + N, o1);
  if (logits) { put(logits + o, l0); put(logits + o + N, l1); }
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

// stage 2: dpooled = conv^T(dpre, k) ; dk += sum_pixels dpre * shifted(pooled), on 32 x 32-pixel tiles staged with their halo in LDS
// (six planes: dpre of the two streams, the four pooled maps).
//
// Round 5.  The first form (16 x 16 tiles, a thread per pixel) formed the kernel gradient as 196 per-pixel products per thread, each
// folded over a lane quad by two DPP adds, written to LDS and summed there behind a barrier per (stream, channel): 3 LDS reads + 1 LDS
// write per tap and pixel, 34 us per launch at B = 16 x 128 x 128 for 0.1 GFLOP.  Now both halves are register-blocked and nothing
// but the final 196 sums crosses lanes:
//   * dpooled: a thread owns FOUR pixels of a row; per kernel row it reads the 12 dpre values under them once (three 16-byte LDS
//     reads) for 4 x 7 taps x 2 channels of FMAs;
//   * dk: a thread owns ONE kernel row u and one tile row: it walks the 32 pixels of the row with dpre's row and pooled's row (shifted by u - 3)
//     in registers - 7 accumulators (the taps v of its row), 18 16-byte LDS reads per 224 FMAs; the tile's 32 rows are summed through LDS
//     once at the end (one barrier), one global atomic per tap and block - a quarter of the blocks of the old form.
// Row pitch 44 words: the 15 rows a wave's dk threads touch at one column lie in 15 different bank quads.
// (gate.hip is built without SLP vectorisation: tests/test_build_isa.py.)
__global__ void __launch_bounds__(256) gate_weights_bwd2_kernel(const float* __restrict__ dpre, const float* __restrict__ pooled,
                                                                const float* __restrict__ k, float* __restrict__ dpooled,
                                                                float* __restrict__ slots, int B, int H, int W) {
  __shared__ __attribute__((aligned(16))) float sp[6 * G2PLANE];        // [0..1] dpre s, [2..5] pooled (2 s + ch); later the row sums
  const int N = H * W;
  const int tw = (W + G2T - 1) / G2T, th = (H + G2T - 1) / G2T;
  const int b = blockIdx.x / (tw * th), t = blockIdx.x % (tw * th);
  const int h0 = (t / tw) * G2T, w0 = (t % tw) * G2T;
  const int tid = threadIdx.x;
  // ---- stage the six planes: tile row i = image row h0 - 3 + i, tile column j = image column w0 - 4 + j; zero outside the image
  gate2_stage<6>(sp, tid, h0, w0, H, W, [&](int pl) { return pl < 2 ? dpre + ((int64_t)b * 2 + pl) * N : pooled + ((int64_t)b * 4 + (pl - 2)) * N; });
  __syncthreads();
  // ---- dpooled: pixels (row, c4 .. c4 + 3).  forward: out[h, w] += k[u, v] * in[h + u - 3, w + v - 3]  ->  input (h, w) was read by
  //      output (h + 3 - u, w + 3 - v): tile row row + 6 - u, tile column c4 + j + 7 - v
  {
    const int row = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const float* dp = sp + s * G2PLANE + (row + 6 - u) * G2LD + c4;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(dp), x1 = *reinterpret_cast<const f32x4*>(dp + 4), x2 = *reinterpret_cast<const f32x4*>(dp + 8);
        const float x[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int v = 0; v < 7; ++v) {
            const float kv = k[((s * 2 + ch) * 7 + u) * 7 + v];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ch][j] = fmaf(kv, x[j + 7 - v], acc[ch][j]);
          }
      }
      const int h = h0 + row, w = w0 + c4;
      if (h < H) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          float* dst = dpooled + ((int64_t)b * 4 + 2 * s + ch) * N + (int64_t)h * W + w;
          if ((W & 3) == 0) {
            if (w < W) *reinterpret_cast<f32x4*>(dst) = f32x4{acc[ch][0], acc[ch][1], acc[ch][2], acc[ch][3]};
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (w + j < W) dst[j] = acc[ch][j];
          }
        }
      }
    }
  }
  // ---- dk: kernel row u, tile row `row`: dk[s][ch][u][v] += dpre_s(row, col) * pooled_{s,ch}(row + u - 3, col + v - 3); dpre is zero
  //      outside the image, so the tile's dead pixels add nothing
  const int ku = tid & 7, krow = tid >> 3;
  float dkacc[4][7];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int v = 0; v < 7; ++v) dkacc[q][v] = 0.f;
  if (ku < 7) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float dv[32];
      const float* dr = sp + s * G2PLANE + (krow + 3) * G2LD + 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(dr + 4 * i);
        dv[4 * i] = q[0]; dv[4 * i + 1] = q[1]; dv[4 * i + 2] = q[2]; dv[4 * i + 3] = q[3];
      }
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        float mv[40];
        const float* mr = sp + (2 + 2 * s + ch) * G2PLANE + (krow + ku) * G2LD;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(mr + 4 * i);
          mv[4 * i] = q[0]; mv[4 * i + 1] = q[1]; mv[4 * i + 2] = q[2]; mv[4 * i + 3] = q[3];
        }
#pragma unroll
        for (int col = 0; col < 32; ++col)
#pragma unroll
          for (int v = 0; v < 7; ++v) dkacc[2 * s + ch][v] = fmaf(dv[col], mv[col + v + 1], dkacc[2 * s + ch][v]);
      }
    }
  }
  __syncthreads();                                       // every wave is done with the planes: their storage takes the row sums
  constexpr int RLD = 197;
  static_assert(32 * RLD <= 6 * G2PLANE, "row sums do not fit the plane storage");
  if (ku < 7) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int v = 0; v < 7; ++v) sp[krow * RLD + q * 49 + ku * 7 + v] = dkacc[q][v];
  }
  __syncthreads();
  if (tid < 196) {
    float sum = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) sum += sp[r * RLD + tid];
    atomicAdd(&slots[(blockIdx.x % RSSF_GATE_SLOTS) * GATE_SLOT_ELEMS + tid], sum);
  }
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

// Pool backward AND the LayerNorm backward of both token streams in one walk (rssf_gate_pool_ln_bwd): the same thread -> (view pixels,
// view-channel range) map as the pooling kernels, so a thread's {d mean, d max, argmax} and its eight LayerNorm weights (the channel of
// a view pixel does not depend on the view-channel when N % C == 0) live in registers, a token row is the pieces of G aligned lanes, and
// what rssf_gate_pool_bwd wrote back to d(xhat) only for rssf_layernorm_bwd to read it again never leaves the registers (it is rounded
// to the activation dtype as the three-launch form rounds it).  LayerNorm backward: norm.hip::ln_bwd_vec's arithmetic.  Persistent
// grid (<= RSSF_GLB_BLOCKS blocks: every block ends in 2C same-address atomics).
template <typename T, int G>
__global__ void __launch_bounds__(256) gate_pool_ln_bwd_kernel(const float* __restrict__ dpooled, const int32_t* __restrict__ argmax,
                                                               const T* __restrict__ dxhat, const T* __restrict__ dyhat, const T* __restrict__ x,
                                                               const T* __restrict__ y, const float* __restrict__ stx, const float* __restrict__ sty,
                                                               const float* __restrict__ gamma, const T* __restrict__ dx_add, T* __restrict__ dx,
                                                               T* __restrict__ dy, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int N,
                                                               int C) {
  constexpr int V = Vec<T>::N, PARTS = 4;
  __shared__ float sg[PARTS][2][64 * 2];            // per wave: dgamma / dbeta partials of its lanes' channel pieces [G][V] (G * V = C <= 128)
  const int nv = N / V;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int sub = lane & (G - 1);                   // this lane's piece of a token row: channels sub * V .. + V - 1
  const int qn = N / C;
  const int cper = C / PARTS, cp0 = part * cper;
  float gam[V], ag[V], ab[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { gam[i] = gamma[sub * V + i]; ag[i] = 0.f; ab[i] = 0.f; }
  const int64_t total = (int64_t)B * 2 * nv;
  for (int64_t g0 = (int64_t)blockIdx.x * 64; g0 < total; g0 += (int64_t)gridDim.x * 64) {
    const int64_t gid = g0 + lane;
    const bool live = gid < total;                  // (total is a multiple of G: a lane group is live or dead as a whole)
    const int64_t gq = live ? gid : 0;
    const int p0 = (int)(gq % nv) * V;
    const int s = (int)((gq / nv) % 2);
    const int b = (int)(gq / (2 * (int64_t)nv));
    const float* dm = dpooled + ((int64_t)b * 4 + 2 * s) * N + p0;
    const int32_t* am = argmax + ((int64_t)b * 2 + s) * N + p0;
    float gmean[V], gmax[V];
    int a[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { gmean[i] = dm[i] / C; gmax[i] = dm[N + i]; a[i] = am[i]; }
    const int64_t img = (int64_t)b * N * C;
    const T* src = (s == 0 ? x : y) + img + p0;
    const T* dsrc = (s == 0 ? dxhat : dyhat) + img + p0;
    const T* asrc = (s == 0 && dx_add) ? dx_add + img + p0 : nullptr;
    T* dst = (s == 0 ? dx : dy) + img + p0;
    const float2* st = reinterpret_cast<const float2*>((s == 0 ? stx : sty) + (int64_t)b * N * 2);
    int n = (int)(((int64_t)cp0 * N + p0) / C);
    for (int cp = cp0; cp < cp0 + cper; ++cp) {
      Vec<T> vx, vd, va;
      vx.load(src + (int64_t)cp * N); vd.load(dsrc + (int64_t)cp * N);
      if (asrc) va.load(asrc + (int64_t)cp * N);
      const float2 ms = st[n];
      float s1 = 0.f, s2 = 0.f, xh[V], gg[V];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float d = vd.get(i) + gmean[i] + (a[i] == cp ? gmax[i] : 0.f);
        if constexpr (sizeof(T) == 2) d = bf2f(f2bf(d));          // what rssf_gate_pool_bwd stores and rssf_layernorm_bwd reads back
        xh[i] = (vx.get(i) - ms.x) * ms.y;
        gg[i] = d * gam[i];
        s1 += gg[i]; s2 += gg[i] * xh[i];
        if (live) { ag[i] += d * xh[i]; ab[i] += d; }
      }
      s1 = gate_group_sum<G>(s1); s2 = gate_group_sum<G>(s2);
      s1 /= C; s2 /= C;
      float o[V];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        o[i] = ms.y * (gg[i] - s1 - xh[i] * s2);
        if (asrc) o[i] += va.get(i);
      }
      if (live) {
        Vec<T> w;
        w.set_all(o);
        w.store(dst + (int64_t)cp * N);
      }
      n += qn;
    }
  }
  // lanes of a wave that own the same channel piece (same lane % G): rotations inside the 16-lane rows, then the four rows
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float u = ag[i], w = ab[i];
    if (G <= 8) { u += dpp_mov<0x128>(u); w += dpp_mov<0x128>(w); }
    if (G <= 4) { u += dpp_mov<0x124>(u); w += dpp_mov<0x124>(w); }
    if (G <= 2) { u += dpp_mov<0x122>(u); w += dpp_mov<0x122>(w); }
    if (G <= 1) { u += dpp_mov<0x121>(u); w += dpp_mov<0x121>(w); }
    ag[i] = rows_reduce<OpSum>(u); ab[i] = rows_reduce<OpSum>(w);
  }
  if (lane < G) {
#pragma unroll
    for (int i = 0; i < V; ++i) { sg[part][0][lane * V + i] = ag[i]; sg[part][1][lane * V + i] = ab[i]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int w = i / C, c = i % C;
    const float t = (sg[0][w][c] + sg[1][w][c]) + (sg[2][w][c] + sg[3][w][c]);
    atomicAdd((w == 0 ? dgamma : dbeta) + c, t);
  }
}

}  // namespace

// lanes per token row of the fused statistics + pooling launch, or 0: N % C == 0 (a row keeps its lanes at every view-channel), whole
// 16-byte vectors, a power of two of at most 16 lanes per row, four waves over the view-channels
static int ln_gate_pool_group(int N, int C, int dtype) {
  const int V = dtype == RSSF_BF16 ? 8 : dtype == RSSF_F32 ? 4 : 0;
  if (!V || C % V || N % C || C % 4) return 0;
  const int g = C / V;
  return (g == 1 || g == 2 || g == 4 || g == 8 || g == 16) ? g : 0;
}
extern "C" int rssf_ln_gate_pool_fwd_supported(int B, int N, int C, int dtype) { return (B > 0 && N > 0 && C > 0 && ln_gate_pool_group(N, C, dtype)) ? 1 : 0; }

extern "C" int rssf_ln_gate_pool_fwd(const void* x, const void* y, const float* gamma, const float* beta, float eps, float* stats_x,
                                     float* stats_y, float* pooled, int32_t* argmax, int B, int N, int C, int dtype, void* stream) {
  RSSF_REQUIRE(x && y && stats_x && stats_y && gamma && beta && pooled && argmax && B > 0 && N > 0 && C > 0, "ln_gate_pool_fwd: bad arguments");
  const int G = ln_gate_pool_group(N, C, dtype);
  RSSF_REQUIRE(G, "ln_gate_pool_fwd: unsupported shape (ask rssf_ln_gate_pool_fwd_supported)");
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  const int64_t total = (int64_t)B * 2 * N;
  const dim3 gv((unsigned)((total / V + 63) / 64));
  hipStream_t st = (hipStream_t)stream;
#define RSSF_LGP(Tt, Gv) gate_pool_fwd_vec_kernel<Tt, 4, Gv><<<gv, 256, 0, st>>>((const Tt*)x, (const Tt*)y, stats_x, stats_y, gamma, beta, pooled, argmax, B, N, C, eps)
  if (dtype == RSSF_BF16) { if (G == 1) RSSF_LGP(bf16_t, 1); else if (G == 2) RSSF_LGP(bf16_t, 2); else if (G == 4) RSSF_LGP(bf16_t, 4); else if (G == 8) RSSF_LGP(bf16_t, 8); else RSSF_LGP(bf16_t, 16); }
  else { if (G == 1) RSSF_LGP(float, 1); else if (G == 2) RSSF_LGP(float, 2); else if (G == 4) RSSF_LGP(float, 4); else if (G == 8) RSSF_LGP(float, 8); else RSSF_LGP(float, 16); }
#undef RSSF_LGP
  return check_launch("ln_gate_pool_fwd");
}

extern "C" int rssf_gate_pool_ln_bwd_supported(int B, int N, int C, int dtype) { return (B > 0 && N > 0 && C > 0 && C <= 128 && ln_gate_pool_group(N, C, dtype)) ? 1 : 0; }

extern "C" int rssf_gate_pool_ln_bwd(const float* dpooled, const int32_t* argmax, const void* dxhat, const void* dyhat, const void* x, const void* y,
                                     const float* stats_x, const float* stats_y, const float* gamma, const void* dx_add, void* dx, void* dy,
                                     float* dgamma, float* dbeta, int B, int N, int C, int dtype, void* stream) {
  RSSF_REQUIRE(dpooled && argmax && dxhat && dyhat && x && y && stats_x && stats_y && gamma && dx && dy && dgamma && dbeta,
               "gate_pool_ln_bwd: bad arguments");
  RSSF_REQUIRE(rssf_gate_pool_ln_bwd_supported(B, N, C, dtype) == 1, "gate_pool_ln_bwd: unsupported shape (ask rssf_gate_pool_ln_bwd_supported)");
  const int G = ln_gate_pool_group(N, C, dtype);
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  int64_t blocks = ((int64_t)B * 2 * (N / V) + 63) / 64;
#ifndef RSSF_GLB_BLOCKS
#define RSSF_GLB_BLOCKS 512      // (16 x 128 x 128 x 32 bf16: 256 blocks 33.4 us, 512: 25.8, 1 024: 29.8 - every block ends in 2C same-address atomics)
#endif
  if (blocks > RSSF_GLB_BLOCKS) blocks = RSSF_GLB_BLOCKS;
  hipStream_t st = (hipStream_t)stream;
#define RSSF_GLB(Tt, Gv) gate_pool_ln_bwd_kernel<Tt, Gv><<<(unsigned)blocks, 256, 0, st>>>(dpooled, argmax, (const Tt*)dxhat, (const Tt*)dyhat, (const Tt*)x, \
    (const Tt*)y, stats_x, stats_y, gamma, (const Tt*)dx_add, (Tt*)dx, (Tt*)dy, dgamma, dbeta, B, N, C)
  if (dtype == RSSF_BF16) { if (G == 1) RSSF_GLB(bf16_t, 1); else if (G == 2) RSSF_GLB(bf16_t, 2); else if (G == 4) RSSF_GLB(bf16_t, 4); else if (G == 8) RSSF_GLB(bf16_t, 8); else RSSF_GLB(bf16_t, 16); }
  else { if (G == 1) RSSF_GLB(float, 1); else if (G == 2) RSSF_GLB(float, 2); else if (G == 4) RSSF_GLB(float, 4); else if (G == 8) RSSF_GLB(float, 8); else RSSF_GLB(float, 16); }
#undef RSSF_GLB
  return check_launch("gate_pool_ln_bwd");
}

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
      if (split) gate_pool_fwd_vec_kernel<float, 4><<<gv, 256, 0, st>>>((const float*)x, (const float*)y, const_cast<float*>(stats_x), const_cast<float*>(stats_y), gamma, beta, pooled, argmax, B, N, C, 0.f);
      else gate_pool_fwd_vec_kernel<float, 1><<<gv, 64, 0, st>>>((const float*)x, (const float*)y, const_cast<float*>(stats_x), const_cast<float*>(stats_y), gamma, beta, pooled, argmax, B, N, C, 0.f);
    } else {
      if (split) gate_pool_fwd_vec_kernel<bf16_t, 4><<<gv, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)y, const_cast<float*>(stats_x), const_cast<float*>(stats_y), gamma, beta, pooled, argmax, B, N, C, 0.f);
      else gate_pool_fwd_vec_kernel<bf16_t, 1><<<gv, 64, 0, st>>>((const bf16_t*)x, (const bf16_t*)y, const_cast<float*>(stats_x), const_cast<float*>(stats_y), gamma, beta, pooled, argmax, B, N, C, 0.f);
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
  const int tiles = ((H + G2T - 1) / G2T) * ((W + G2T - 1) / G2T);
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
  const int tiles = ((H + G2T - 1) / G2T) * ((W + G2T - 1) / G2T);
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
