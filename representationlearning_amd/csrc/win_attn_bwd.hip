// Fused 7x7-window cross attention, backward (autograd of the forward in win_attn_fwd.hip; reference:
// torch autograd through InterlacedPoolAttention2.forward :164-188 and Mhca, modules/DAL.py:873-1020,
// incl. the three gradient paths into q/k — softmax, mean(M), argmax-routed max(M) — SURVEY App. C).
//
// One PAIR of wavefronts per window, one attention head each (they share the six LDS tiles, stage the inputs together and
// exchange their shares of the input gradient through LDS at the end; workgroup barriers separate those phases, so all
// pairs of a workgroup walk their windows in lockstep).  The forward is recomputed from (x, y, LN stats, omega) so nothing but the
// block inputs is saved; HBM traffic = read x, y, dout + write dxhat, dyhat.  Every operand is staged in LDS ONCE,
// token-major; contractions over the token axis read it through the LDS transpose read (RowFrag, win_attn.hip.h), which
// keeps the per-wave footprint at 6 tiles (was 11 + transposed weight copies) so that 4 waves fit a CU instead of 2;
// O and dU never touch LDS: they are formed in both register orientations by swapping MFMA operands.
// Softmax / dS tiles stay in registers and are computed in both orientations instead of being transposed through LDS.
// Weight gradients are accumulated in LDS per workgroup and flushed once.
#include <mutex>
#include <type_traits>
#include "win_attn.hip.h"
using namespace rssf;
using namespace rssf::wa;

#ifndef RSSF_BWD_DBG
#define RSSF_BWD_DBG 0      // timing experiments only: 1 = no S1 loads, 2 = no S9 (results are garbage)
#endif
#ifndef RSSF_BWD_COMPACT_BF16
#define RSSF_BWD_COMPACT_BF16 0
#endif

namespace {

template <typename T, typename DM, bool ACC_LDS> struct BwdLayout {
  static constexpr int P = Pad<T>::X;
  static constexpr int LDX = DM::CP + P;     // XS/YS/GS   [token][in channel]
  static constexpr int LDV = DM::CV + P;     // QS/KS/VS   [token][virtual channel]
  static constexpr int LDW = DM::CP + P;     // Wq/Wk/Wv/WoT rows = virtual channel, k = real channel
  static constexpr int LDD = DM::DP + P;     // dM / dM^T scratch
  static constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
  // fp32 (parity mode) must squeeze C = 48 into 160 KiB: tile regions sized per kind and ONE dM scratch read both ways
  // (straight, and through the transposing fragment read as dM^T).  bf16 keeps the uniform regions and the two dM / dM^T
  // copies of round 1: its C = 48 instantiation (512 VGPRs + ~80 spilled under divergent pair/head branches) is sensitive to
  // code changes - the compact form produced garbage there on hardware while C = 32 / 18 and fp32 C = 48 were fine.
  static constexpr bool COMPACT = RSSF_BWD_COMPACT_BF16 || sizeof(T) == 4;
  static constexpr int REGU = (max3(LP * LDX, LP * LDV, 0) + 7) / 8 * 8;
  static constexpr int REGX = COMPACT ? (LP * LDX + 7) / 8 * 8 : REGU;       // XS YS GS          [token][in channel]
  static constexpr int REGV = COMPACT ? (LP * LDV + 7) / 8 * 8 : REGU;       // QS KS VS (later dq dk dv)   [token][virtual channel]
  static constexpr int SCRATCH_DM = ((COMPACT ? 1 : 2) * DM::DP * LDD + 7) / 8 * 8;
  static constexpr int SCRATCH_Q = (int)((3 * LP * sizeof(float) + sizeof(T) - 1) / sizeof(T) + 7) / 8 * 8;      // per-query softmax statistics (S4 -> S5)
  static constexpr int SCRATCH = SCRATCH_DM > SCRATCH_Q ? SCRATCH_DM : SCRATCH_Q;
  static constexpr int W_ELEMS = 4 * DM::CV * LDW;
  static constexpr int F_ELEMS = 3 * DM::CV + 2 * DM::CP;                       // bq bk bv, gamma beta
  static constexpr int A_ELEMS = ACC_LDS ? 4 * DM::CV * DM::CP + 3 * DM::CV + DM::CP : 0;
  static constexpr size_t SHARED_OFF = (sizeof(T) * W_ELEMS + sizeof(float) * (F_ELEMS + A_ELEMS) + 15) / 16 * 16;
  // A window is served by a PAIR of waves (one head each) that share the six tiles; each wave has its own dM scratch.
  static_assert(DM::HEADS == 2, "the backward kernel maps one head to each wave of a pair");
  static constexpr int PAIR_ELEMS = 3 * REGX + 3 * REGV + 2 * SCRATCH;
  static constexpr size_t WAVE_BYTES = sizeof(T) * PAIR_ELEMS;            // per pair
  static constexpr size_t LIM = 160 * 1024 - 64;     // 64 B: the statically allocated pair-barrier counters
#ifndef RSSF_BWD_MAX_PAIRS
#define RSSF_BWD_MAX_PAIRS 4       // timing builds only: fewer window pairs per workgroup (= waves per SIMD halved at 2)
#endif
  static constexpr int PAIRS_FIT = (SHARED_OFF + 4 * WAVE_BYTES <= LIM) ? 4 : (SHARED_OFF + 3 * WAVE_BYTES <= LIM) ? 3
                                 : (SHARED_OFF + 2 * WAVE_BYTES <= LIM) ? 2 : 1;
  static constexpr int PAIRS = PAIRS_FIT < RSSF_BWD_MAX_PAIRS ? PAIRS_FIT : RSSF_BWD_MAX_PAIRS;
  static constexpr int WAVES = 2 * PAIRS;
  static constexpr size_t BYTES = SHARED_OFF + WAVE_BYTES * PAIRS;
  static constexpr bool FITS = BYTES <= LIM;
  // head-sum exchange of the input-gradient tiles: CT x 2 token tiles x {x, y} f32x4 per lane, in two tile regions
  static_assert(2 * REGX * sizeof(T) >= (size_t)DM::CT * 2 * 2 * 64 * 16, "exchange buffer does not fit two tile regions");
};

// plain (no LN / gate) token-major tile load, zero for pad / dead slots
template <typename T, typename DM>
__device__ __forceinline__ void load_plain_tile(const Geom& g, const T* src, int64_t img, int qh, int qw, T* dst, int ldx,
                                                int lane, int part) {          // two cooperating waves: chunks part, part + 2, ..
  constexpr int V = Vec<T>::N;
  if constexpr (DM::C % V == 0) {
    constexpr int cpr = DM::CP / V;
    constexpr int ITERS = ((LP * cpr + 63) / 64 + 1) / 2;
    Vec<T> v[ITERS];
    bool ok[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {          // branch-free: all loads in flight together, dead slots read token 0
      const int e = lane + (part + 2 * it) * 64, t = e / cpr, c0 = (e % cpr) * V;
      const int n = slot_token(g, qh, qw, t);
      ok[it] = e < LP * cpr && n >= 0 && c0 < DM::C;
      v[it].load(src + (img + (ok[it] ? n : 0)) * DM::C + (ok[it] ? c0 : 0));
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int e = lane + (part + 2 * it) * 64, t = e / cpr, c0 = (e % cpr) * V;
      if (!ok[it]) v[it].raw = {0, 0, 0, 0};
      if (e < LP * cpr) v[it].store(dst + t * ldx + c0);
    }
  } else {
    for (int e = lane + 64 * part; e < LP * DM::CP; e += 128) {
      const int t = e / DM::CP, c = e % DM::CP;
      const int n = slot_token(g, qh, qw, t);
      stf(dst + t * ldx + c, (n >= 0 && c < DM::C) ? ldf(src + (img + n) * DM::C + c) : 0.f);
    }
  }
}

// C-layout tile (rows = virtual channel mt*16+4g+r, col = token tt*16+l15) -> token-major buffer [token][ld]: the four
// values of a lane are contiguous, one 8-byte (bf16) / 16-byte (f32) LDS store; dead tokens (>= L) are written as zero
// so that contractions over the token axis see only the 49 live slots.
// 4 consecutive elements of a global row: one 8-byte (bf16) / 16-byte (f32) access
template <typename T> struct Quad4;
template <> struct Quad4<bf16_t> {
  typedef uint2 raw;
  static __device__ __forceinline__ raw load(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ f32x4 unpack(const raw& u) {
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
  }
};
template <> struct Quad4<float> {
  typedef f32x4 raw;
  static __device__ __forceinline__ raw load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ f32x4 unpack(const raw& u) { return u; }
};
template <typename T>
__device__ __forceinline__ void store_tok_major(T* buf, int ld, int mt, int tt, const f32x4& v, int l15, int grp, int L) {
  const int tok = tt * 16 + l15;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  store4(buf + tok * ld + mt * 16 + grp * 4, tok < L ? v : z);
}

// Barrier of the TWO waves that serve one window (a monotonic LDS counter: both add 1 per barrier, both wait for 2 x phase).  The
// phases of a window used to be separated by workgroup barriers, which marched all pairs of a CU in lockstep: every wave of the CU
// sat in the same global-memory phase at the same time, and nothing overlapped its latency (49 % of the wave cycles were parked).
// Pairs now drift apart and cover each other.  Both waves of a pair are resident in the same workgroup, so the spin cannot
// deadlock; LDS is coherent within the CU once the writer's lgkmcnt has drained.
__device__ __forceinline__ void pair_barrier(unsigned* ctr, unsigned& phase, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  phase += 2;
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < phase) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <typename T, typename DM, bool ACC_LDS>
__global__ void __launch_bounds__((BwdLayout<T, DM, ACC_LDS>::WAVES * 64))
winattn_bwd_kernel(rssf_winattn_bwd_params bp, Geom g) {
  using LY = BwdLayout<T, DM, ACC_LDS>;
  constexpr int LDX = LY::LDX, LDV = LY::LDV, LDW = LY::LDW, LDD = LY::LDD;
  constexpr int C = DM::C, CP = DM::CP, CV = DM::CV, MT = DM::MT, CT = DM::CT, TPH = DM::TPH, D = DM::D, DP = DM::DP;
  const rssf_winattn_fwd_params& p = bp.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ unsigned pbar[4];                   // pair barriers (LY::PAIRS <= 4)
  T* sWq = reinterpret_cast<T*>(smem_raw);       // [CV][LDW]
  T* sWk = sWq + CV * LDW;
  T* sWv = sWk + CV * LDW;
  T* sWoT = sWv + CV * LDW;                      // [CV][LDW]  WoT[m][c] = Wo[c][m]
  float* sB = reinterpret_cast<float*>(sWoT + CV * LDW);   // bq bk bv [CV]
  float* sLn = sB + 3 * CV;                      // gamma, beta [CP]
  float* aW = sLn + 2 * CP;                      // accumulators: dWq dWk dWv [CV][CP], dWo^T [CV][CP], dbq dbk dbv [CV], dbo [CP]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform: scalar branches on pair / hw
  const int l15 = lane & 15, grp = lane >> 4;
  const int pair = wave >> 1, hw = wave & 1;               // wave hw of a pair computes head hw of the pair's window
  T* base = reinterpret_cast<T*>(smem_raw + LY::SHARED_OFF) + (size_t)pair * LY::PAIR_ELEMS;
  T* XS = base;                 T* YS = XS + LY::REGX;     T* GS = YS + LY::REGX;
  T* QS = GS + LY::REGX;        T* KS = QS + LY::REGV;     T* VS = KS + LY::REGV;      // later dq, dk, dv (head by head)
  T* dMs = VS + LY::REGV + hw * LY::SCRATCH;
#if defined(RSSF_BWD_ZERO_LDS)      // debug builds (tools/ab_lib.sh): does anything read LDS it never wrote?
  for (int i = threadIdx.x; i < (int)(LY::BYTES / 4); i += blockDim.x) reinterpret_cast<unsigned*>(smem_raw)[i] = 0u;
  __syncthreads();
#endif

  if constexpr (CV == C && C % 4 == 0) {
    // head width = its padded width (C = 32): the staged images are the row-major matrices themselves - one 16-byte load per
    // thread and matrix instead of ~1 500 dynamic instructions of element-wise index arithmetic per wave and launch.  The pad
    // columns k >= C of a row are never read (every contraction stops at CP = C).
    for (int i = threadIdx.x; i < C * C / 4; i += blockDim.x) {
      const int m = (i * 4) / C, k = (i * 4) % C;
      store4(sWq + m * LDW + k, reinterpret_cast<const f32x4*>(p.wq)[i]);
      store4(sWk + m * LDW + k, reinterpret_cast<const f32x4*>(p.wk)[i]);
      store4(sWv + m * LDW + k, reinterpret_cast<const f32x4*>(p.wv)[i]);
      const f32x4 o4 = reinterpret_cast<const f32x4*>(p.wo)[i];      // Wo[c = m][m' = k..k+3] -> WoT[m'][c]
#pragma unroll
      for (int j = 0; j < 4; ++j) stf(sWoT + (k + j) * LDW + m, o4[j]);
    }
  } else {
    for (int i = threadIdx.x; i < CV * LDW; i += blockDim.x) {
      const int m = i / LDW, k = i % LDW;
      const int rc = real_ch<DM>(m);
      const bool ok = rc >= 0 && k < C;
      stf(sWq + i, ok ? p.wq[rc * C + k] : 0.f);
      stf(sWk + i, ok ? p.wk[rc * C + k] : 0.f);
      stf(sWv + i, ok ? p.wv[rc * C + k] : 0.f);
      stf(sWoT + i, ok ? p.wo[k * C + rc] : 0.f);
    }
  }
  // biases and LayerNorm affine: the five loads of a thread in flight together (clamped indices, selected afterwards)
  for (int i = threadIdx.x; i < (CV > CP ? CV : CP); i += blockDim.x) {
    const int rc = i < CV ? real_ch<DM>(i) : -1;
    const int rq = rc >= 0 ? rc : 0, ci = i < C ? i : 0;
    const float vq = p.bq[rq], vk = p.bk[rq], vv = p.bv[rq], vg = p.ln_gamma[ci], vb = p.ln_beta[ci];
    if (i < CV) {
      sB[i] = rc >= 0 ? vq : 0.f;
      sB[CV + i] = rc >= 0 ? vk : 0.f;
      sB[2 * CV + i] = rc >= 0 ? vv : 0.f;
    }
    if (i < CP) {
      sLn[i] = i < C ? vg : 0.f;
      sLn[CP + i] = i < C ? vb : 0.f;
    }
  }
  if (ACC_LDS) for (int i = threadIdx.x; i < LY::A_ELEMS; i += blockDim.x) aW[i] = 0.f;
  if (threadIdx.x < 4) pbar[threadIdx.x] = 0u;
  unsigned bar_phase = 0u;
  __syncthreads();

  // accumulate one weight-gradient element: which = 0 q, 1 k, 2 v (index [m][c]); 3 = o (index [m][c] meaning dWo[c][m])
  auto acc_w = [&](int which, int m, int c, float v) {
    if (ACC_LDS) { atomicAdd(&aW[which * CV * CP + m * CP + c], v); return; }
    const int rc = real_ch<DM>(m);
    if (rc < 0 || c >= C) return;
    float* dst = which == 0 ? bp.dwq : which == 1 ? bp.dwk : which == 2 ? bp.dwv : bp.dwo;
    atomicAdd(which == 3 ? dst + c * C + rc : dst + rc * C + c, v);
  };
  auto acc_b = [&](int which, int idx, float v) {   // 0..2: dbq/dbk/dbv[virtual m]; 3: dbo[c]
    if (ACC_LDS) { atomicAdd(&aW[4 * CV * CP + (which < 3 ? which * CV + idx : 3 * CV + idx)], v); return; }
    if (which == 3) { if (idx < C) atomicAdd(bp.dbo + idx, v); return; }
    const int rc = real_ch<DM>(idx);
    if (rc < 0) return;
    atomicAdd((which == 0 ? bp.dbq : which == 1 ? bp.dbk : bp.dbv) + rc, v);
  };

  const float scale = rsqrtf((float)D);
  const T* X = reinterpret_cast<const T*>(p.x);
  const T* Y = reinterpret_cast<const T*>(p.y);
  const T* DOUT = reinterpret_cast<const T*>(bp.dout);
  T* DXH = reinterpret_cast<T*>(bp.dxhat);
  T* DYH = reinterpret_cast<T*>(bp.dyhat);
  T* PW = reinterpret_cast<T*>(bp.prod_ws);          // products in the activation dtype (bf16 mode: half the scratch traffic)
  const int wpi = g.QH * g.QW;

  // Parameter gradients live in registers across ALL windows of this wave (the kernel is LDS-bound to one wave per
  // SIMD, so VGPRs are free) and are folded into the workgroup's LDS accumulators once at the end; measured before:
  // ~60 LDS float atomics per lane per window kept the LDS pipe busy 20x longer than all other LDS traffic together.
  f32x4 gWq[TPH][CT], gWk[TPH][CT], gWv[TPH][CT], gWo[TPH][CT];       // rows of THIS wave's head: virtual channel (hw*TPH + i)*16 ..
  f32x4 gbq[TPH], gbk[TPH], gbv[TPH];
  float gbo = 0.f;
  const typename Packed<T>::type ones = Packed<T>::pack(f32x4{1.f, 1.f, 1.f, 1.f});
#pragma unroll
  for (int mt = 0; mt < TPH; ++mt) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      gWq[mt][ct] = {0.f, 0.f, 0.f, 0.f}; gWk[mt][ct] = gWq[mt][ct]; gWv[mt][ct] = gWq[mt][ct]; gWo[mt][ct] = gWq[mt][ct];
    }
    gbq[mt] = {0.f, 0.f, 0.f, 0.f}; gbk[mt] = gbq[mt]; gbv[mt] = gbq[mt];
  }

  // All pairs of the block run the same number of iterations: the phases are separated by workgroup barriers.
  const int nslots = gridDim.x * LY::PAIRS;
  const int iters = (g.nWin + nslots - 1) / nslots;
  for (int iter = 0; iter < iters; ++iter) {
    // (workgroups numbered XCD-major: the windows of one round that share 128-byte lines meet in one L2 - see win_attn_fwd.hip)
    const int lb = g.xcd_major ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int wi_raw = iter * nslots + lb * LY::PAIRS + pair;
    const bool active = wi_raw < g.nWin;
    const int wi = active ? wi_raw : 0;
    const int b = wi / wpi, qh = (wi % wpi) / g.QW, qw = wi % g.QW;
    const int64_t img = (int64_t)b * g.N;
    const float* om0 = p.omega + (int64_t)b * 2 * g.N;
    float* dom0 = bp.domega + (int64_t)b * 2 * g.N;

    f32x4 dxt[CT][NT], dyt[CT][NT];     // d(x~)^T, d(y~)^T accumulators (this head's share): rows = real channel, col = token
    // ---- S1: gated LN'ed inputs and dout tile, token-major; the two waves of the pair stage half of the chunks each ----
    pair_barrier(pbar + pair, bar_phase, lane);                    // the previous window's exchange buffers (which alias these tiles) have been read
#if RSSF_BWD_DBG != 1
    if (active) {
      load_gated_tiles<T, DM, 2>(p, g, sLn, X, Y, om0, img, qh, qw, XS, YS, LDX, lane, hw);
      load_plain_tile<T, DM>(g, DOUT, img, qh, qw, GS, LDX, lane, hw);
    }
#endif
    pair_barrier(pbar + pair, bar_phase, lane);
    if (active) {
    // ---- S2: projections of THIS wave's head; q,k,v staged token-major (K = tokens contractions use RowFrag) -----------
#pragma unroll
    for (int mi0 = 0; mi0 < TPH; ++mi0) {
      const int mt = hw * TPH + mi0;
      const int mrow = mt * 16 + grp * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
        aq = mma_tile<T>(sWq + mt * 16 * LDW, LDW, XS + tt * 16 * LDX, LDX, CP, aq);
        ak = mma_tile<T>(sWk + mt * 16 * LDW, LDW, YS + tt * 16 * LDX, LDX, CP, ak);
        av = mma_tile<T>(sWv + mt * 16 * LDW, LDW, YS + tt * 16 * LDX, LDX, CP, av);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          aq[r] = (aq[r] + sB[mrow + r]) * scale;
          ak[r] += sB[CV + mrow + r];
          av[r] += sB[2 * CV + mrow + r];
        }
        store_tok_major<T>(QS, LDV, mt, tt, aq, l15, grp, g.L);
        store_tok_major<T>(KS, LDV, mt, tt, ak, l15, grp, g.L);
        store_tok_major<T>(VS, LDV, mt, tt, av, l15, grp, g.L);
      }
    }
    wave_sync();

    // dbo += sum_tokens dout   (columns of GS)
    static_assert(CP <= 64, "dbo: one lane per padded channel");
    if (hw == 0 && lane < CP)
      for (int t = 0; t < g.L; ++t) gbo += ldf(GS + t * LDX + lane);

    {
      const int h = hw;
      const int hoff = h * DP;
      // ---- S3: alpha = sigmoid(mean(M)+max(M)), M = q_h^T k_h, with its argmax ---------------------------------
      float msum = 0.f, mmax = -INFINITY;
      int marg = 0;
#pragma unroll
      for (int it = 0; it < TPH; ++it)
#pragma unroll
        for (int jt = 0; jt < TPH; ++jt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k0 = 0; k0 < LP; k0 += 16) acc = mma_row_row<T>(QS, LDV, hoff + it * 16, KS, LDV, hoff + jt * 16, k0, acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + grp * 4 + r, j = jt * 16 + l15;
            if (i < D && j < D) {
              msum += acc[r];
              if (acc[r] > mmax) { mmax = acc[r]; marg = i * DP + j; }
            }
          }
        }
      msum = wave_reduce_dpp<OpSum>(msum);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mmax, o, 64);
        const int oa = __shfl_xor(marg, o, 64);
        if (om > mmax || (om == mmax && oa < marg)) { mmax = om; marg = oa; }
      }
      const float alpha = sigmoidf(msum / (float)(D * D) + mmax);

      // dO^T (head rows) = WoT * dout^T ; dU = alpha * dO
      f32x4 dU[TPH][NT];
#pragma unroll
      for (int mi = 0; mi < TPH; ++mi)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          dU[mi][qt] = mma_tile<T>(sWoT + (hoff + mi * 16) * LDW, LDW, GS + qt * 16 * LDX, LDX, CP, acc);
        }

      // ---- S4: orientation 1 (rows = keys, col = query): P, U = P v, dalpha, dS -> dq -------------------------
      // U, O = alpha U and dU are ALSO formed with rows = query, col = channel (operand order swapped): in that
      // orientation they chain straight into dWo (here) and dv (S5) from registers - no LDS copies of O / dU.
      f32x4 dq[TPH][NT], dk[TPH][NT], dv[TPH][NT];
      typename Packed<T>::type dUq[NT][TPH];               // alpha * dO, rows = query, col = head channel
      // FOLD: dk / dv accumulate inside the query-tile loop below from TRANSPOSED P / dS tiles.  Only for one channel tile per head
      // (C = 32 / 18).  The C = 48 instantiations (two tiles per head, 512 VGPRs + 100-200 spilled registers) keep the validated
      // second pass: built with the folded form, the fp32 C = 48 kernel returned v_proj gradients 60x too large at -O3 / -O2 and
      // correct ones at -O1 (same source; tools/attn_c48_dbg.py) - a code-generation problem at that register pressure, not
      // worth chasing for the parity-mode kernel.
#ifndef RSSF_BWD_FOLD_ALL
#define RSSF_BWD_FOLD_ALL 0        // 1: the folded form for C = 48 too (A/B builds: tools/ab_lib.sh; see the note above)
#endif
      constexpr bool FOLD = TPH == 1 || RSSF_BWD_FOLD_ALL;
      if constexpr (FOLD)
#pragma unroll
      for (int mi = 0; mi < TPH; ++mi)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) { dk[mi][kt] = {0.f, 0.f, 0.f, 0.f}; dv[mi][kt] = dk[mi][kt]; }
      // identity tile as an MFMA B operand (chain slot order): X^T = mma(A = X read as a chained C-layout tile, B = I).  dk and dv
      // contract over the QUERY axis, i.e. need P and dS with rows = queries: one MFMA per 16x16 tile turns the (keys x queries)
      // tiles of this loop around - the second orientation used to be RECOMPUTED (scores, 256 exponentials and ~1 500 VALU
      // instructions per window and wave, plus the per-query statistics through LDS).
      f32x4 idv;
#pragma unroll
      for (int r = 0; r < 4; ++r) idv[r] = (grp * 4 + r == l15) ? 1.f : 0.f;
      const typename Packed<T>::type ident = Packed<T>::pack(idv);

      float smx[NT], sinv[NT], srs[NT];     // !FOLD only
      float dalpha = 0.f;
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        __builtin_amdgcn_sched_barrier(0);
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          s[kt] = mma_tile<T>(KS + kt * 16 * LDV + hoff, LDV, QS + qt * 16 * LDV + hoff, LDV, DP, acc);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (kt * 16 + grp * 4 + r >= g.L) s[kt][r] = -INFINITY;
            mx = fmaxf(mx, s[kt][r]);
          }
        mx = rows_reduce<OpMax>(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[kt][r] = __expf(s[kt][r] - mx); sum += s[kt][r]; }
        sum = rows_reduce<OpSum>(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] *= inv;
        // U[q][m] = sum_key P[q][key] v[key][m] ; dalpha += <dO, U> ; O = alpha U -> dWo ; dU = alpha dO (both orientations)
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 u = {0.f, 0.f, 0.f, 0.f}, dO = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NT; ++kt)
            u = Packed<T>::mma(Packed<T>::pack(s[kt]), RowFrag<T>::load(VS, LDV, kt * 16, hoff + mi * 16), u);
          dO = mma_tile<T>(GS + qt * 16 * LDX, LDX, sWoT + (hoff + mi * 16) * LDW, LDW, CP, dO);    // rows = query, col = channel
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dalpha += dO[r] * u[r];
            o[r] = alpha * u[r];
            dO[r] *= alpha;
            dU[mi][qt][r] *= alpha;
          }
          dUq[qt][mi] = Packed<T>::pack(dO);
          const typename Packed<T>::type op = Packed<T>::pack(o);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)      // dWo^T[m][c] += sum_q O[q][m] dout[q][c]   (dout rows of dead slots are zero)
            gWo[mi][ct] = Packed<T>::mma(op, RowFrag<T>::load(GS, LDX, qt * 16, ct * 16), gWo[mi][ct]);
        }
        // dP^T[key][query] = sum_m v[key][m] dU[query][m]
        f32x4 dp[NT];
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mi = 0; mi < TPH; ++mi) acc = mma_lds_chain<T>(VS + kt * 16 * LDV, LDV, hoff + mi * 16, dU[mi][qt], acc);
          dp[kt] = acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) rs += s[kt][r] * acc[r];
        }
        rs = rows_reduce<OpSum>(rs);
        if constexpr (!FOLD) { smx[qt] = mx; sinv[qt] = inv; srs[qt] = rs; }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[kt][r] = s[kt][r] * (dp[kt][r] - rs);       // dS^T
        // dq^T[m][query] = sum_key k^T[m][key] dS^T[key][query]
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) acc = mma_row_chain<T>(KS, LDV, hoff + mi * 16, kt * 16, dp[kt], acc);
          dq[mi][qt] = acc;
        }
        // rows = queries: dk^T[m][key] += sum_q q^T[m][q] dS[q][key] ; dv^T[m][key] += sum_q dU[q][m] P[q][key]
        // (the transposed tiles hold exactly the bf16 values that fed dq / U above: re-packing them is exact)
        if constexpr (FOLD)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          const f32x4 pt = Packed<T>::mma(Packed<T>::pack(s[kt]), ident, z);
          const f32x4 dst = Packed<T>::mma(Packed<T>::pack(dp[kt]), ident, z);
          const typename Packed<T>::type ptp = Packed<T>::pack(pt);
#pragma unroll
          for (int mi = 0; mi < TPH; ++mi) {
            dk[mi][kt] = mma_row_chain<T>(QS, LDV, hoff + mi * 16, qt * 16, dst, dk[mi][kt]);
            dv[mi][kt] = Packed<T>::mma(dUq[qt][mi], ptp, dv[mi][kt]);
          }
        }
      }
      dalpha = wave_reduce_dpp<OpSum>(dalpha);            // each tile element lives in exactly one lane: plain sum
      if constexpr (!FOLD) {
      // per-query softmax statistics for orientation 2 (there a lane needs the values of queries 4g..4g+3, which live in lanes
      // 4g..4g+3 of this orientation): parked in the dM scratch (free until S6) and read back as one 16-byte LDS read per tile,
      // instead of 192 variable-lane shuffles per window (ds_bpermute + a v_readlane waterfall: 536 instructions)
      float* qstat = reinterpret_cast<float*>(dMs);        // [3][LP]: max, 1/sum, rowsum(P dP)
      if (grp == 0) {
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
          qstat[qt * 16 + l15] = smx[qt]; qstat[LP + qt * 16 + l15] = sinv[qt]; qstat[2 * LP + qt * 16 + l15] = srs[qt];
        }
      }
      wave_sync();

      // ---- S5: orientation 2 (rows = queries, col = key): dS -> dk, P -> dv ---------------------------------------
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        f32x4 p2[NT], ds2[NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = mma_tile<T>(QS + qt * 16 * LDV + hoff, LDV, KS + kt * 16 * LDV + hoff, LDV, DP, acc);
          f32x4 dpp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mi = 0; mi < TPH; ++mi) dpp = mma_chain_lds<T>(dU[mi][qt], VS + kt * 16 * LDV, LDV, hoff + mi * 16, dpp);
          const bool keylive = kt * 16 + l15 < g.L;
          const f32x4 mxq = *reinterpret_cast<const f32x4*>(qstat + qt * 16 + grp * 4);             // queries qt*16 + 4g + r
          const f32x4 invq = *reinterpret_cast<const f32x4*>(qstat + LP + qt * 16 + grp * 4);
          const f32x4 rsq = *reinterpret_cast<const f32x4*>(qstat + 2 * LP + qt * 16 + grp * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = keylive ? __expf(acc[r] - mxq[r]) * invq[r] : 0.f;
            p2[qt][r] = pv;
            ds2[qt][r] = pv * (dpp[r] - rsq[r]);
          }
        }
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = ak;
#pragma unroll
          for (int qt = 0; qt < NT; ++qt) {
            ak = mma_row_chain<T>(QS, LDV, hoff + mi * 16, qt * 16, ds2[qt], ak);
            av = Packed<T>::mma(dUq[qt][mi], Packed<T>::pack(p2[qt]), av);      // dv^T[m][key] = sum_q dU[q][m] P[q][key]
          }
          dk[mi][kt] = ak; dv[mi][kt] = av;
        }
      }

      wave_sync();      // the per-query statistics (float view of the dM scratch) have been read: S6 overwrites the scratch
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- S6: alpha path: dM = du * (1/d^2 + onehot(argmax)) -> dq += k dM^T, dk += q dM ---------------------------
      const float du = dalpha * alpha * (1.f - alpha);
#pragma unroll
      for (int it = 0; it < TPH; ++it)
#pragma unroll
        for (int jt = 0; jt < TPH; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + grp * 4 + r, j = jt * 16 + l15;
            float v = 0.f;
            if (i < D && j < D) v = du / (float)(D * D) + ((i * DP + j) == marg ? du : 0.f);
            stf(dMs + i * LDD + j, v);
            if constexpr (!LY::COMPACT) stf(dMs + DP * LDD + j * LDD + i, v);
          }
      wave_sync();
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 aq = dq[mi][tt], ak = dk[mi][tt];
#pragma unroll
          for (int mj = 0; mj < TPH; ++mj) {
            // dq^T += dM k^T (both operands k-contiguous in LDS, K = head channels mj*16..), dk^T += dM^T q^T (dM^T rows =
            // columns of the one dM buffer, through the transposing fragment read)
            aq = mma_tile<T>(dMs + mi * 16 * LDD + mj * 16, LDD, KS + tt * 16 * LDV + hoff + mj * 16, LDV, 16, aq);
            if constexpr (LY::COMPACT)
              ak = Packed<T>::mma(RowFrag<T>::load(dMs, LDD, mj * 16, mi * 16), chain_frag_lds<T>(QS + tt * 16 * LDV, LDV, hoff + mj * 16), ak);
            else
              ak = mma_tile<T>(dMs + DP * LDD + mi * 16 * LDD + mj * 16, LDD, QS + tt * 16 * LDV + hoff + mj * 16, LDV, 16, ak);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) aq[r] *= scale;       // q = (W x + b) * scale
          dq[mi][tt] = aq; dk[mi][tt] = ak;
        }
      wave_sync();   // every read of the QS/KS/VS columns of this head is done -> reuse them for dq, dk, dv

      __builtin_amdgcn_sched_barrier(0);
      // ---- S7: stage gradients of the projection outputs; bias grads; input grads --------------------------------
#pragma unroll
      for (int mi = 0; mi < TPH; ++mi) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          store_tok_major<T>(QS, LDV, h * TPH + mi, tt, dq[mi][tt], l15, grp, g.L);
          store_tok_major<T>(KS, LDV, h * TPH + mi, tt, dk[mi][tt], l15, grp, g.L);
          store_tok_major<T>(VS, LDV, h * TPH + mi, tt, dv[mi][tt], l15, grp, g.L);
          const bool live = tt * 16 + l15 < g.L;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!live) { dq[mi][tt][r] = 0.f; dk[mi][tt][r] = 0.f; dv[mi][tt][r] = 0.f; }
        }
        // d(x~)^T[c][t] += sum_m Wq[m][c] dq[m][t] ;  d(y~)^T += Wk^T dk + Wv^T dv
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            // first contribution starts from zero here: the 2 x CT x NT accumulators are not live during S3..S6
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            dxt[ct][tt] = mma_row_chain<T>(sWq, LDW, ct * 16, hoff + mi * 16, dq[mi][tt], mi == 0 ? z : dxt[ct][tt]);
            dyt[ct][tt] = mma_row_chain<T>(sWk, LDW, ct * 16, hoff + mi * 16, dk[mi][tt], mi == 0 ? z : dyt[ct][tt]);
            dyt[ct][tt] = mma_row_chain<T>(sWv, LDW, ct * 16, hoff + mi * 16, dv[mi][tt], dyt[ct][tt]);
          }
      }
    }
    wave_sync();

    __builtin_amdgcn_sched_barrier(0);
    // ---- S8: weight gradients dW[m][c] = sum_t d(proj)[t][m] * in[t][c]; bias gradients = the same contraction against
    //      a column of ones (every column of the gb* tiles holds the row sum; column 0 is flushed) ------------------------
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      typename Packed<T>::type fx[CT], fy[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        fx[ct] = RowFrag<T>::load(XS, LDX, tt * 16, ct * 16);
        fy[ct] = RowFrag<T>::load(YS, LDX, tt * 16, ct * 16);
      }
#pragma unroll
      for (int mi = 0; mi < TPH; ++mi) {
        const int mc = (hw * TPH + mi) * 16;
        const typename Packed<T>::type fq = RowFrag<T>::load(QS, LDV, tt * 16, mc), fk = RowFrag<T>::load(KS, LDV, tt * 16, mc),
                                       fv = RowFrag<T>::load(VS, LDV, tt * 16, mc);
        gbq[mi] = Packed<T>::mma(fq, ones, gbq[mi]);
        gbk[mi] = Packed<T>::mma(fk, ones, gbk[mi]);
        gbv[mi] = Packed<T>::mma(fv, ones, gbv[mi]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          gWq[mi][ct] = Packed<T>::mma(fq, fx[ct], gWq[mi][ct]);
          gWk[mi][ct] = Packed<T>::mma(fk, fy[ct], gWk[mi][ct]);
          gWv[mi][ct] = Packed<T>::mma(fv, fy[ct], gWv[mi][ct]);
        }
      }
    }
    }   // active: S2 .. S8

    // ---- S9's global operands: raw x / y, LN statistics, gate weights of the two token tiles this wave finishes (L2 hits: S1 read
    //      them).  Requesting them ahead of the two exchange barriers was measured and dropped: the ~60 registers they hold across
    //      the exchange spill (58 at C = 32 bf16) and the kernel went from 173 to 218 us ---------------------------------------
    const bool vec9 = (C % 4 == 0) && (g.N % 4 == 0);
    float2 s9sx[2], s9sy[2];
    typename Quad4<T>::raw s9x[2][CT], s9y[2][CT];
    f32x4 s9w0[2][CT], s9w1[2][CT];
    int s9n[2], s9pp[2][CT];
    auto s9_load = [&](int tl) {
      {
        const int n = slot_token(g, qh, qw, (2 * hw + tl) * 16 + l15);
        const int nn = n >= 0 ? n : 0;
        s9n[tl] = n;
        s9sx[tl] = *reinterpret_cast<const float2*>(p.stats_x + (img + nn) * 2);
        s9sy[tl] = *reinterpret_cast<const float2*>(p.stats_y + (img + nn) * 2);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c0 = ct * 16 + grp * 4;
          const int64_t f = (int64_t)nn * C + (c0 < C ? c0 : 0);
          s9pp[tl][ct] = (int)((unsigned)f % (unsigned)g.N);
          s9x[tl][ct] = Quad4<T>::load(X + img * C + f);
          s9y[tl][ct] = Quad4<T>::load(Y + img * C + f);
          s9w0[tl][ct] = *reinterpret_cast<const f32x4*>(om0 + s9pp[tl][ct]);
          s9w1[tl][ct] = *reinterpret_cast<const f32x4*>(om0 + g.N + s9pp[tl][ct]);
        }
      }
    };

    // ---- head sum of d(x~), d(y~): wave hw keeps token tiles 2hw, 2hw+1 and receives the partner's share of them through
    //      LDS (the six tiles are dead now): partner 0's inbox = {XS, YS}, partner 1's = {GS, QS} -------------------------
    pair_barrier(pbar + pair, bar_phase, lane);
    f32x4* inbox0 = reinterpret_cast<f32x4*>(XS);
    f32x4* inbox1 = reinterpret_cast<f32x4*>(GS);
    auto send = [&](auto TT0, f32x4* box) {               // TT0: first token tile of the RECEIVER
      constexpr int t0 = decltype(TT0)::value;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          box[((ct * 2 + tl) * 2 + 0) * 64 + lane] = dxt[ct][t0 + tl];
          box[((ct * 2 + tl) * 2 + 1) * 64 + lane] = dyt[ct][t0 + tl];
        }
    };
    if (active) {
      if (hw == 0) send(std::integral_constant<int, 2>{}, inbox1);
      else send(std::integral_constant<int, 0>{}, inbox0);
    }
    pair_barrier(pbar + pair, bar_phase, lane);
    if (active) {
    auto finish = [&](auto TT0, const f32x4* box) {
      constexpr int t0 = decltype(TT0)::value;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          const f32x4 ox = box[((ct * 2 + tl) * 2 + 0) * 64 + lane], oy = box[((ct * 2 + tl) * 2 + 1) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) { dxt[ct][t0 + tl][r] += ox[r]; dyt[ct][t0 + tl][r] += oy[r]; }
        }

    // ---- S9: dxhat = d(x~) * omega0, dyhat = d(y~) * omega1 ; domega += d(x~) * LN(x) ----------------------------------
    // A lane owns 4 consecutive channels (4g..4g+3 of tile ct) of token tt*16 + l15.  Vector path (C and N multiples of
    // 4: the 4 gate weights are contiguous and never wrap): all global loads of a token tile are issued branch-free
    // (dead / padded slots read token 0) before anything is consumed; 8-byte stores; only the stores / atomics are
    // predicated.
    if (vec9) {
#pragma unroll
      for (int tt = t0; tt < t0 + 2; ++tt) {
        const int tl = tt - t0;
        s9_load(tl);
        const int n = s9n[tl];
        const float2 sx = s9sx[tl], sy = s9sy[tl];
        const auto& rx = s9x[tl]; const auto& ry = s9y[tl];
        const auto& w0 = s9w0[tl]; const auto& w1 = s9w1[tl];
        const auto& pp = s9pp[tl];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c0 = ct * 16 + grp * 4;
          if (n < 0 || c0 >= C) continue;
          const f32x4 xv = Quad4<T>::unpack(rx[ct]), yv = Quad4<T>::unpack(ry[ct]);
          f32x4 ox, oy, gx, gy;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ga = sLn[c0 + r], be = sLn[CP + c0 + r];
            ox[r] = dxt[ct][tt][r] * w0[ct][r];
            oy[r] = dyt[ct][tt][r] * w1[ct][r];
            gx[r] = dxt[ct][tt][r] * ((xv[r] - sx.x) * sx.y * ga + be);
            gy[r] = dyt[ct][tt][r] * ((yv[r] - sy.x) * sy.y * ga + be);
          }
          const int64_t off = img * C + (int64_t)n * C + c0;
          store4(DXH + off, ox);
          store4(DYH + off, oy);
          if (PW) {                                       // products to scratch; domega_reduce_kernel sums them
            const int64_t po = ((int64_t)b * g.N + n) * C + c0;
            store4(PW + po, gx);
            store4(PW + (int64_t)g.B * g.N * C + po, gy);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              atomicAdd(dom0 + pp[ct] + r, gx[r]);
              atomicAdd(dom0 + g.N + pp[ct] + r, gy[r]);
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int tt = t0; tt < t0 + 2; ++tt) {
        const int n = slot_token(g, qh, qw, tt * 16 + l15);
        if (n < 0) continue;
        const float2 sx = *reinterpret_cast<const float2*>(p.stats_x + (img + n) * 2);
        const float2 sy = *reinterpret_cast<const float2*>(p.stats_y + (img + n) * 2);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = ct * 16 + grp * 4 + r;
            if (c >= C) continue;
            const int64_t f = (int64_t)n * C + c;
            const int pp = (int)((unsigned)f % (unsigned)g.N);
            const int64_t off = img * C + f;
            const float xh = (ldf(X + off) - sx.x) * sx.y * sLn[c] + sLn[CP + c];
            const float yh = (ldf(Y + off) - sy.x) * sy.y * sLn[c] + sLn[CP + c];
            stf(DXH + off, dxt[ct][tt][r] * om0[pp]);
            stf(DYH + off, dyt[ct][tt][r] * om0[g.N + pp]);
            if (PW) {
              stf(PW + off, dxt[ct][tt][r] * xh);
              stf(PW + (int64_t)g.B * g.N * C + off, dyt[ct][tt][r] * yh);
            } else {
              atomicAdd(dom0 + pp, dxt[ct][tt][r] * xh);
              atomicAdd(dom0 + g.N + pp, dyt[ct][tt][r] * yh);
            }
          }
      }
    }
    };   // finish
    static_assert(NT == 4, "two token tiles per wave of a pair");
#if RSSF_BWD_DBG != 2
    if (hw == 0) finish(std::integral_constant<int, 0>{}, inbox0);
    else finish(std::integral_constant<int, 2>{}, inbox1);
#endif
    }   // active: exchange + S9
  }

  // ---- fold this wave's register accumulators into the workgroup accumulators (or straight into HBM) -------------------
#pragma unroll
  for (int mt = 0; mt < TPH; ++mt) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (hw * TPH + mt) * 16 + grp * 4 + r, c = ct * 16 + l15;
        acc_w(0, m, c, gWq[mt][ct][r]); acc_w(1, m, c, gWk[mt][ct][r]); acc_w(2, m, c, gWv[mt][ct][r]);
        acc_w(3, m, c, gWo[mt][ct][r]);
      }
    if (l15 == 0)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (hw * TPH + mt) * 16 + grp * 4 + r;
        acc_b(0, m, gbq[mt][r]); acc_b(1, m, gbk[mt][r]); acc_b(2, m, gbv[mt][r]);
      }
  }
  if (hw == 0 && lane < CP) acc_b(3, lane, gbo);

  if (ACC_LDS) {
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * CV * CP; i += blockDim.x) {
      const int which = i / (CV * CP), m = (i / CP) % CV, c = i % CP;
      const int rc = real_ch<DM>(m);
      if (rc < 0 || c >= C) continue;
      float* dst = which == 0 ? bp.dwq : which == 1 ? bp.dwk : which == 2 ? bp.dwv : bp.dwo;
      atomicAdd(which == 3 ? dst + c * C + rc : dst + rc * C + c, aW[i]);
    }
    for (int i = threadIdx.x; i < 3 * CV; i += blockDim.x) {
      const int which = i / CV, rc = real_ch<DM>(i % CV);
      if (rc >= 0) atomicAdd((which == 0 ? bp.dbq : which == 1 ? bp.dbk : bp.dbv) + rc, aW[4 * CV * CP + i]);
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) atomicAdd(bp.dbo + i, aW[4 * CV * CP + 3 * CV + i]);
  }
}

// domega[b][s][pp] = sum_j prod_s[b][pp + j*N], j = 0..C-1 : the C activation elements whose flat offset is pp modulo N
// share gate weight pp (the reference's (B,N,C) -> view(B,C,H,W) scramble).  Coalesced over pp, C strided reads.
template <typename T>
__global__ void __launch_bounds__(256) domega_reduce_kernel(const T* __restrict__ prod, float* __restrict__ domega, int B, int N, int C) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * 2 * N) return;
  const int pp = (int)(gid % N), s = (int)((gid / N) % 2), b = (int)(gid / (2 * (int64_t)N));
  const T* src = prod + ((int64_t)s * B + b) * N * C + pp;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int j = 0;
  for (; j + 3 < C; j += 4) {
    a0 += ldf(src + (int64_t)j * N); a1 += ldf(src + (int64_t)(j + 1) * N); a2 += ldf(src + (int64_t)(j + 2) * N); a3 += ldf(src + (int64_t)(j + 3) * N);
  }
  for (; j < C; ++j) a0 += ldf(src + (int64_t)j * N);
  domega[gid] = (a0 + a1) + (a2 + a3);
}

template <typename T, typename DM, bool ACC_LDS>
int launch_bwd(const rssf_winattn_bwd_params* p, const Geom& g, hipStream_t st) {
  using LY = BwdLayout<T, DM, ACC_LDS>;
  int blocks = (g.nWin + LY::PAIRS - 1) / LY::PAIRS;
  auto kern = winattn_bwd_kernel<T, DM, ACC_LDS>;
  // once per (instantiation, device), outside any replayed hipGraph capture: the dynamic-LDS limit and the CU count
  constexpr int MAX_DEVICES = 32;
  static std::once_flag once[MAX_DEVICES];
  static int cus[MAX_DEVICES];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAX_DEVICES) dev = 0;
  hipError_t e = hipSuccess;
  std::call_once(once[dev], [&] {
    if (LY::BYTES > 64 * 1024) e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LY::BYTES);
    int v = 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    (void)hipGetLastError();
    cus[dev] = v;
  });
  if (e != hipSuccess) { set_error("winattn_bwd: cannot raise LDS limit: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  if (blocks > cus[dev]) blocks = cus[dev];   // one resident workgroup per CU (157 KB LDS): persistent, one gradient flush each
  Geom gx = g;
  gx.xcd_major = (blocks % 8 == 0 && blocks >= 8) ? 1 : 0;
  kern<<<blocks, LY::WAVES * 64, LY::BYTES, st>>>(*p, gx);
  int rc = check_launch("winattn_bwd");
  if (rc || !p->prod_ws) return rc;
  const int64_t n = (int64_t)g.B * 2 * g.N;
  domega_reduce_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(reinterpret_cast<const T*>(p->prod_ws), p->domega, g.B, g.N, DM::C);
  return check_launch("winattn_bwd(domega reduce)");
}

// LDS-resident gradient accumulators (one flush per workgroup) when they do not cost a pair of waves; else straight atomics
template <typename T, typename DM>
int pick_bwd(const rssf_winattn_bwd_params* p, const Geom& g, hipStream_t st) {
  using LA = BwdLayout<T, DM, true>;
  using LN = BwdLayout<T, DM, false>;
  // (C = 48 bf16: the LDS accumulators leave room for one pair only; the two-pair global-atomics variant aborted on
  // hardware - 512 VGPRs + 90 spilled under divergent pair/head branches - so the validated one-pair form stays)
  if constexpr (LA::FITS) return launch_bwd<T, DM, true>(p, g, st);
  else if constexpr (LN::FITS) return launch_bwd<T, DM, false>(p, g, st);
  else {
    set_error("winattn_bwd: C=%d in this dtype exceeds the 160 KiB LDS of a CU (use bf16 activations)", DM::C);
    return RSSF_ERR_UNSUPPORTED;
  }
}

template <typename T>
int dispatch_bwd(const rssf_winattn_bwd_params* p, const Geom& g, hipStream_t st) {
  if (p->f.heads == 2 && p->f.C == 32) return pick_bwd<T, Dims<32, 2>>(p, g, st);
  if (p->f.heads == 2 && p->f.C == 18) return pick_bwd<T, Dims<18, 2>>(p, g, st);
  if (p->f.heads == 2 && p->f.C == 48) return pick_bwd<T, Dims<48, 2>>(p, g, st);
  set_error("winattn_bwd: no kernel instantiated for C=%d heads=%d (built: 18/32/48 x 2 heads)", p->f.C, p->f.heads);
  return RSSF_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int64_t rssf_winattn_bwd_workspace_elems(int B, int H, int W, int C) { return (int64_t)2 * B * H * W * C; }

extern "C" int rssf_winattn_bwd(const rssf_winattn_bwd_params* p, void* stream) {
  RSSF_REQUIRE(p, "winattn_bwd: null params");
  const rssf_winattn_fwd_params& f = p->f;
  RSSF_REQUIRE(f.x && f.y && f.stats_x && f.stats_y && f.omega && f.ln_gamma && f.ln_beta && f.wq && f.bq && f.wk && f.bk &&
                   f.wv && f.bv && f.wo && f.bo,
               "winattn_bwd: null forward tensor pointer");
  RSSF_REQUIRE(p->dout && p->dxhat && p->dyhat && p->domega && p->dwq && p->dbq && p->dwk && p->dbk && p->dwv && p->dbv &&
                   p->dwo && p->dbo,
               "winattn_bwd: null gradient pointer");
  RSSF_REQUIRE(f.B > 0 && f.H > 0 && f.W > 0 && f.C > 0 && f.heads > 0 && f.window == 7, "winattn_bwd: bad shape");
  RSSF_REQUIRE(f.C % f.heads == 0, "winattn_bwd: embed_dim must be divisible by num_heads");
  RSSF_REQUIRE((int64_t)f.H * f.W * f.C < ((int64_t)1 << 31), "winattn_bwd: H*W*C must stay below 2^31");
  const Geom g = make_geom(f.B, f.H, f.W, f.window);
  hipStream_t st = (hipStream_t)stream;
  if (f.dtype == RSSF_F32) return dispatch_bwd<float>(p, g, st);
  if (f.dtype == RSSF_BF16) return dispatch_bwd<bf16_t>(p, g, st);
  set_error("winattn_bwd: unsupported dtype %d", f.dtype);
  return RSSF_ERR_UNSUPPORTED;
}
