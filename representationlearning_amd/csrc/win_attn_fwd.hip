// Fused 7x7-window cross attention, forward.
//
// Replaces, in one kernel and one HBM round trip (read low, read high, write out):
//   InterlacedPoolAttention2.forward :164-188  (gate multiply, pad, permute, attention, rev-permute, depad)
//   PadBlock / LocalPermuteModule              (modules/multihead_isa_attention.py:364-426)   -> index math only
//   Mhca.multi_head_attention_forward          (modules/DAL.py:873-1020)
//   the attention residual of GeneralTransformerBlock.forward (modules/MTFM.py:107)
//
// Mapping: one 64-lane wavefront owns one window (49 live tokens padded to 64 = 4 MFMA tiles); 4 waves per
// workgroup share the projection weights in LDS.  Everything between the single global read of the two
// 49xC tiles and the single global write stays in LDS / registers.  All matmuls run "transposed"
// (channels x tokens) so that every MFMA result lands in exactly the register layout the next MFMA wants
// as an operand (q,k -> S^T -> P^T -> O^T -> out^T).  q and k are projected in both orientations (one for S^T, one for
// M = q^T k and v for O^T = v^T P^T), which costs 16 extra MFMAs per window and removes every LDS round trip after the
// tile load: LDS holds only the two gated input tiles and the weights.
// The kernel is HBM-bound (0.27 GFLOP vs 3.1 MB per image-block); MFMA just keeps the
// arithmetic out of the way.
#include <mutex>
#include "win_attn.hip.h"
using namespace rssf;
using namespace rssf::wa;

// tuning knobs (defaults = the measured best on MI355X; tools/attn_variants.sh builds the alternatives)
#ifndef RSSF_FWD_PIPE
#define RSSF_FWD_PIPE 1            // bf16: next window's loads one window ahead + raw-x tile in LDS (0: two tiles per wave, three workgroups per CU)
#endif
#ifndef RSSF_FWD_OCC
#define RSSF_FWD_OCC 2             // waves per SIMD the register allocator makes room for (bf16)
#endif
#ifndef RSSF_FWD_DBG
#define RSSF_FWD_DBG 0             // timing builds only (tools/ab_lib_flags.sh): 1 no softmax arithmetic, 2 no attention core (S / softmax / PV),
#endif                             // 4 no LayerNorm / gate arithmetic in the tile staging, 8 no projections either (tiles -> out-projection)
#ifndef RSSF_FWD_PREFETCH_STATS
#define RSSF_FWD_PREFETCH_STATS 1  // LayerNorm statistics travel with the prefetched tiles (1) or are fetched at use (0)
#endif

namespace {

template <typename T, typename DM> struct FwdLayout {
  static constexpr int LDX = DM::CP + Pad<T>::X;     // xs/ys rows  [token][channel]
  static constexpr int LDW = DM::CP + Pad<T>::X;     // Wq/Wk/Wv rows [virtual channel][in channel]
  static constexpr int LDO = DM::CV + Pad<T>::X;     // Wo rows [out channel][virtual channel]
  static constexpr int REGION = (LP * LDX + 7) / 8 * 8;
  static constexpr int W_ELEMS = 3 * DM::CV * LDW + DM::CP * LDO;
  static constexpr int F_ELEMS = 3 * DM::CV + 3 * DM::CP;   // biases + LN affine (fp32)
  static constexpr size_t SHARED_OFF = (sizeof(T) * W_ELEMS + sizeof(float) * F_ELEMS + 15) / 16 * 16;
  // bf16: a third tile per wave keeps the RAW x tokens for the residual (they are in registers when the tile is staged; re-reading
  // them from global cost 9 MB of fabric traffic per launch on top of the L2 hits)
  static constexpr int NREG = (sizeof(T) == 2 && RSSF_FWD_PIPE) ? 3 : 2;
  static constexpr size_t WAVE_BYTES = sizeof(T) * NREG * REGION;
  // waves per workgroup: as many (<= 4) as fit the 160 KiB LDS of one CU
  static constexpr int WAVES = (SHARED_OFF + 4 * WAVE_BYTES <= 160 * 1024) ? 4 : (SHARED_OFF + 2 * WAVE_BYTES <= 160 * 1024) ? 2 : 1;
  static constexpr size_t BYTES = SHARED_OFF + WAVE_BYTES * WAVES;
};

// projection tile: D(16x16) = A[16 rows][K] * B[16 rows][K]^T with the widest MFMA the dtype / K allow
// `acc` = the bias tile (C-layout), so that no separate bias add follows the MFMA
template <typename T>
__device__ __forceinline__ f32x4 proj_tile(const T* A, int lda, const T* B, int ldb, int K, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  if constexpr (sizeof(T) == 2) {
    if (K % 32 == 0) {                       // v_mfma_f32_16x16x32_bf16: half the instructions of the K=16 form
      const T* pa = A + (lane & 15) * lda + (lane >> 4) * 8;
      const T* pb = B + (lane & 15) * ldb + (lane >> 4) * 8;
      for (int k = 0; k < K; k += 32)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(pa + k), *reinterpret_cast<const bf16x8*>(pb + k),
                                                      acc, 0, 0, 0);
      return acc;
    }
  }
  return mma_tile<T>(A, lda, B, ldb, K, acc);
}

// Address = wave-uniform base + 32-bit BYTE offset per lane: the form the backend turns into `global_load ... v_off, s[base:base+1]`
// (scalar base, zero-extended VGPR offset).  An element index scaled by the pointer type does not qualify - the shift may carry out
// of 32 bits as far as the compiler knows - and costs a 64-bit VALU add per access.
template <typename U> __device__ __forceinline__ const U* at_bytes(const void* base, unsigned byte_off) {
  return reinterpret_cast<const U*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename U> __device__ __forceinline__ U* at_bytes(void* base, unsigned byte_off) {
  return reinterpret_cast<U*>(reinterpret_cast<char*>(base) + byte_off);
}

// {a.x - b.x, a.y - b.x} as ONE packed instruction (the compiler emits two v_sub_f32 for scores that come out of an MFMA)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub_lo(f32x2 a, f32x2 b) {
#ifdef RSSF_FWD_NO_PK
  return f32x2{a[0] - b[0], a[1] - b[0]};
#endif
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// 8-byte (bf16) / 16-byte (f32) access to 4 consecutive channels of one token
template <typename T> struct Quad;
template <> struct Quad<bf16_t> {
  typedef uint2 raw;
  static __device__ __forceinline__ raw load_raw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ void unpack(const raw& u, float (&v)[4]) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    uint2 u;
    u.x = f2bf2(v[0], v[1]);
    u.y = f2bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
  }
};
template <> struct Quad<float> {
  typedef float4 raw;
  static __device__ __forceinline__ raw load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void unpack(const raw& u, float (&v)[4]) { v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; }
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 u = *reinterpret_cast<const float4*>(p);
    v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};

// Global-load stage of one window's two 49xC tiles (vector path): issued one window AHEAD of its use, so that the HBM
// latency of window i+1 is paid under the MFMA/softmax work of window i (the kernel used to park ~half of its wave cycles on
// s_waitcnt: one wave per window is one long dependency chain, and 2-3 waves per SIMD do not cover an HBM round trip).
template <typename T, typename DM> struct TileRegs {
  static constexpr int V = Vec<T>::N, CPR = DM::CP / V, ITERS = (LP * CPR + 63) / 64;
  Vec<T> vx[ITERS], vy[ITERS];
#if RSSF_FWD_PREFETCH_STATS
  float2 sx[ITERS], sy[ITERS];
#else
  int tok[ITERS];                    // LN statistics are fetched when the tile is finished (L2-resident: written by the pass before)
#endif
  int pp[ITERS];                     // gate-weight index of the chunk's first element, -1: dead slot / padding chunk
  // gate weights of the chunk (x tile / y tile).  L2-resident (128 KB per image) but still a vector-memory round trip: they are
  // requested at the END of the previous window (gates_issue), so that between "finish this window's tiles" and "request the next
  // window's tokens" no load is outstanding whose arrival somebody waits for - vmcnt counts in order, and a wait for a gate weight
  // issued after the next window's token loads is a wait for those HBM loads too (46 % of the wave cycles were parked that way).
  float4 wa[ITERS][V / 4], wb[ITERS][V / 4];
};

// n mod d for 0 <= n < 2^24 with a precomputed float reciprocal: one multiply, a truncation and one correction step (the
// compiler's 32-bit integer division is ~20 instructions and sits behind a branch)
__device__ __forceinline__ int fast_mod(int n, int d, float rcp) {
  int r = n - (int)((float)n * rcp) * d;
  r = r < 0 ? r + d : r;
  return r >= d ? r - d : r;
}

// window-dependent SCALARS of the tile addressing; everything per lane is 32-bit arithmetic against them
struct WinBase {
  int uq, vq;          // image row / column of window slot 0 (may be negative: the centre padding)
  int nc;              // N / C when N % C == 0 (gate weights of a token row are contiguous), else 0
  float rcp_nc;
};

template <typename T, typename DM>
__device__ __forceinline__ void tiles_issue(TileRegs<T, DM>& R, const rssf_winattn_fwd_params& p, const Geom& g, const T* X, const T* Y,
                                            int wi, int lane, int nc, float rcp_nc) {
  using TR = TileRegs<T, DM>;
  const int wpi = g.QH * g.QW;
  const int b = wi / wpi, qh = (wi % wpi) / g.QW, qw = wi % g.QW;           // scalar (wi is wave-uniform)
  const int uq = qh * WIN - g.padT, vq = qw * WIN - g.padL;
  const T* Xi = X + (int64_t)b * g.N * DM::C;                                // uniform bases: saddr + 32-bit lane offsets
  const T* Yi = Y + (int64_t)b * g.N * DM::C;
  const float* Sx = p.stats_x + (int64_t)b * g.N * 2;
  const float* Sy = p.stats_y + (int64_t)b * g.N * 2;
#pragma unroll
  for (int it = 0; it < TR::ITERS; ++it) {          // branch-free: dead / padded slots read token 0 and are zeroed later
    const int e = lane + it * 64;
    const int t = e / TR::CPR, c0 = (e % TR::CPR) * TR::V;
    const int u = uq + t / WIN, v = vq + t % WIN;
    const bool ok = t < WIN * WIN && (unsigned)u < (unsigned)g.H && (unsigned)v < (unsigned)g.W && c0 < DM::C && e < LP * TR::CPR;
    const int nn = ok ? u * g.W + v : 0;
    const unsigned f = (unsigned)(nn * DM::C + (ok ? c0 : 0));                // N*C < 2^31 (checked by the entry points); unsigned:
    R.vx[it].load(at_bytes<T>(Xi, f * (unsigned)sizeof(T)));                  // scalar base + 32-bit lane offset, no 64-bit VALU
    R.vy[it].load(at_bytes<T>(Yi, f * (unsigned)sizeof(T)));
#if RSSF_FWD_PREFETCH_STATS
    R.sx[it] = *at_bytes<float2>(Sx, (unsigned)nn * 8u);
    R.sy[it] = *at_bytes<float2>(Sy, (unsigned)nn * 8u);
#else
    R.tok[it] = nn;
#endif
    // gate-weight index (n*C + c) mod N: with N = nc*C it is (n mod nc)*C + c
    const int pp = nc ? fast_mod(nn, nc, rcp_nc) * DM::C + c0 : (int)(f % (unsigned)g.N);
    R.pp[it] = ok ? pp : -1;
  }
}

// gate weights of the window whose tokens `R` holds (CONTIG geometry: the V weights of a chunk are contiguous)
template <typename T, typename DM>
__device__ __forceinline__ void gates_issue(TileRegs<T, DM>& R, const float* om0, int N) {
  using TR = TileRegs<T, DM>;
  const float* om1 = om0 + N;
#pragma unroll
  for (int it = 0; it < TR::ITERS; ++it) {
    const unsigned q = R.pp[it] < 0 ? 0u : (unsigned)R.pp[it];
#pragma unroll
    for (int i = 0; i < TR::V / 4; ++i) {
      R.wa[it][i] = *at_bytes<float4>(om0, q * 4u + 16u * i);
      R.wb[it][i] = *at_bytes<float4>(om1, q * 4u + 16u * i);
    }
  }
}

// LayerNorm (given stats) * gate weight omega[(n*C+c) mod N] -> LDS as T, zero rows for padded / dead slots
template <typename T, typename DM, bool CONTIG>
__device__ __forceinline__ void tiles_finish(const TileRegs<T, DM>& R, const rssf_winattn_fwd_params& p, int64_t img, const Geom& g,
                                             const float* sLn, const float* om0, T* xs, T* ys, T* xr, int ldx, int lane) {
  using TR = TileRegs<T, DM>;
  constexpr int V = TR::V;
#if !RSSF_FWD_PREFETCH_STATS
  float2 sxv[TR::ITERS], syv[TR::ITERS];
#pragma unroll
  for (int it = 0; it < TR::ITERS; ++it) {
    sxv[it] = *reinterpret_cast<const float2*>(p.stats_x + img * 2 + (unsigned)(R.tok[it] * 2));
    syv[it] = *reinterpret_cast<const float2*>(p.stats_y + img * 2 + (unsigned)(R.tok[it] * 2));
  }
#endif
  static_assert(CONTIG, "tiles_finish: the pipelined path is the N % C == 0 one");
#pragma unroll
  for (int it = 0; it < TR::ITERS; ++it) {
    const int e = lane + it * 64;
    if (e >= LP * TR::CPR) break;
    const int t = e / TR::CPR, c0 = (e % TR::CPR) * V;
    const bool ok = R.pp[it] >= 0;
    float w0[V], w1[V];
#pragma unroll
    for (int i = 0; i < V / 4; ++i) {
      const float4 a4 = R.wa[it][i], b4 = R.wb[it][i];
      w0[4 * i] = a4.x; w0[4 * i + 1] = a4.y; w0[4 * i + 2] = a4.z; w0[4 * i + 3] = a4.w;
      w1[4 * i] = b4.x; w1[4 * i + 1] = b4.y; w1[4 * i + 2] = b4.z; w1[4 * i + 3] = b4.w;
    }
    float fx[V], fy[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float ga = sLn[c0 + i], be = sLn[DM::CP + c0 + i];
#if RSSF_FWD_PREFETCH_STATS
      const float2 sx = R.sx[it], sy = R.sy[it];
#else
      const float2 sx = sxv[it], sy = syv[it];
#endif
      if (RSSF_FWD_DBG & 4) { fx[i] = R.vx[it].get(i) + sx.x + w0[i] + ga; fy[i] = R.vy[it].get(i) + sy.y + w1[i] + be; continue; }
      fx[i] = ((R.vx[it].get(i) - sx.x) * sx.y * ga + be) * w0[i];
      fy[i] = ((R.vy[it].get(i) - sy.x) * sy.y * ga + be) * w1[i];
    }
    Vec<T> ox, oy;
    ox.set_all(fx); oy.set_all(fy);
    if (!ok) { ox.clear(); oy.clear(); }          // dead / padded slots: zero rows (selected on the PACKED words, not per element)
    ox.store(xs + t * ldx + c0);
    oy.store(ys + t * ldx + c0);
    R.vx[it].store(xr + t * ldx + c0);            // raw tokens for the residual of step 4 (dead slots: never read back)
  }
}

template <typename T, typename DM> struct FwdOcc {
  // waves per SIMD the register allocator is asked to make room for: what the LDS footprint allows anyway
  static constexpr int WG_PER_CU = (int)((160 * 1024) / FwdLayout<T, DM>::BYTES);
  static constexpr int WAVES_PER_SIMD = (WG_PER_CU * FwdLayout<T, DM>::WAVES + 3) / 4;
  static constexpr int VALUE = sizeof(T) == 2 ? (WAVES_PER_SIMD > RSSF_FWD_OCC ? RSSF_FWD_OCC : (WAVES_PER_SIMD < 1 ? 1 : WAVES_PER_SIMD)) : 1;
};

// CONTIG: N % C == 0 (every HRNet geometry of the path: the gate weights of a token row are contiguous).  A compile-time flag so
// that the general-geometry gather (one scalar gate weight per element) is not carried as dead code through the hot variant.
template <typename T, typename DM, bool CONTIG>
__global__ void __launch_bounds__((FwdLayout<T, DM>::WAVES * 64), (FwdOcc<T, DM>::VALUE)) winattn_fwd_kernel(rssf_winattn_fwd_params p, Geom g) {
  using LY = FwdLayout<T, DM>;
  constexpr int LDX = LY::LDX, LDW = LY::LDW, LDO = LY::LDO;
  constexpr int C = DM::C, CP = DM::CP, CV = DM::CV, MT = DM::MT, CT = DM::CT, TPH = DM::TPH, D = DM::D;
  constexpr int LW = WIN * WIN;                      // live tokens of a window (the entry point checked window == 7)
  constexpr bool PIPE = RSSF_FWD_PIPE && CONTIG && sizeof(T) == 2 && (C % Vec<T>::N) == 0;      // next window's global loads issued one window ahead
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // ---- workgroup-shared: weights as T (MFMA operands), biases / LN affine fp32 ---------------------------------
  T* sWq = reinterpret_cast<T*>(smem_raw);           // [CV][LDW]  virtual rows, k = real input channel
  T* sWk = sWq + CV * LDW;
  T* sWv = sWk + CV * LDW;
  T* sWo = sWv + CV * LDW;                           // [CP][LDO]  rows = output channel, k = virtual channel
  float* sB = reinterpret_cast<float*>(sWo + CP * LDO);   // bq[CV] bk[CV] bv[CV] bo[CP]
  float* sLn = sB + 3 * CV + CP;                     // gamma[CP] beta[CP]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform: window arithmetic on the SALU, scalar-base global loads
  const int l15 = lane & 15, grp = lane >> 4;
  T* xs = reinterpret_cast<T*>(smem_raw + LY::SHARED_OFF) + (size_t)wave * LY::NREG * LY::REGION;   // [LP][LDX] gated LN(x)
  T* ys = xs + LY::REGION;                                                                    // [LP][LDX] gated LN(y)
  T* xr = ys + LY::REGION;                                                                    // [LP][LDX] raw x (bf16 pipeline only)

  const T* X = reinterpret_cast<const T*>(p.x);
  const T* Y = reinterpret_cast<const T*>(p.y);
  T* OUT = reinterpret_cast<T*>(p.out);
  const int wpi = g.QH * g.QW;
  const int stride = gridDim.x * LY::WAVES;
  // Workgroups go round-robin over the 8 XCDs (conv.hip.h): numbered XCD-major, the workgroups of ONE XCD take a contiguous run of
  // windows per round - row neighbours, whose 448-byte token rows share 128-byte lines, then meet in one L2 (g.xcd_major, set by
  // the launcher when the grid is a multiple of 8; the window -> result map does not depend on who computes a window)
  const int lb = g.xcd_major ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  int wi = lb * LY::WAVES + wave;
  // softmax in base 2 with the scale folded into the q projection: Wq and bq are staged multiplied by log2(e)/sqrt(d), so q needs no
  // epilogue arithmetic at all (v_exp_f32 is a base-2 exponential).  M = q^T k is linear in q: alpha reads mean(M) + max(M) back
  // through 1/log2(e).
  const float scale2 = rsqrtf((float)D) * 1.4426950408889634f;
  TileRegs<T, DM> R;
  const int nc = CONTIG ? g.N / C : 0;
  const float rcp_nc = nc ? 1.0f / (float)nc : 0.f;
  if constexpr (PIPE) {                                                                                      // under the weight staging below
    const int w0 = wi < g.nWin ? wi : g.nWin - 1;
    tiles_issue<T, DM>(R, p, g, X, Y, w0, lane, nc, rcp_nc);
    gates_issue<T, DM>(R, p.omega + (int64_t)(w0 / wpi) * 2 * g.N, g.N);
  }

  if constexpr (CV == C && C % 4 == 0) {
    // head width = its padded width (C = 32): the staged images are the row-major matrices themselves - one 16-byte load per
    // thread and matrix instead of ~1 500 dynamic instructions of element-wise index arithmetic per wave and launch (3 us of a
    // 37 us launch).  The pad columns of a row are never read (every contraction stops at CP = C = CV).
    for (int i = threadIdx.x; i < C * C / 4; i += blockDim.x) {
      const int m = (i * 4) / C, k = (i * 4) % C;
      f32x4 q4 = reinterpret_cast<const f32x4*>(p.wq)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) q4[j] *= scale2;
      store4(sWq + m * LDW + k, q4);
      store4(sWk + m * LDW + k, reinterpret_cast<const f32x4*>(p.wk)[i]);
      store4(sWv + m * LDW + k, reinterpret_cast<const f32x4*>(p.wv)[i]);
      store4(sWo + m * LDO + k, reinterpret_cast<const f32x4*>(p.wo)[i]);
    }
  } else {
    for (int i = threadIdx.x; i < CV * LDW; i += blockDim.x) {
      const int m = i / LDW, k = i % LDW;
      const int rc = real_ch<DM>(m);
      const bool ok = rc >= 0 && k < C;
      stf(sWq + i, ok ? p.wq[rc * C + k] * scale2 : 0.f);
      stf(sWk + i, ok ? p.wk[rc * C + k] : 0.f);
      stf(sWv + i, ok ? p.wv[rc * C + k] : 0.f);
    }
    for (int i = threadIdx.x; i < CP * LDO; i += blockDim.x) {
      const int co = i / LDO, m = i % LDO;
      const int rc = m < CV ? real_ch<DM>(m) : -1;
      stf(sWo + i, (co < C && rc >= 0) ? p.wo[co * C + rc] : 0.f);
    }
  }
  // biases and LayerNorm affine: all six loads of a thread are in flight together (clamped indices, selected afterwards) - as
  // conditional loads they were six serial L2 round trips in front of the workgroup barrier
  for (int i = threadIdx.x; i < (CV > CP ? CV : CP); i += blockDim.x) {
    const int rc = i < CV ? real_ch<DM>(i) : -1;
    const int rq = rc >= 0 ? rc : 0, ci = i < C ? i : 0;
    const float vq = p.bq[rq], vk = p.bk[rq], vv = p.bv[rq], vo = p.bo[ci], vg = p.ln_gamma[ci], vb = p.ln_beta[ci];
    if (i < CV) {
      sB[i] = rc >= 0 ? vq * scale2 : 0.f;
      sB[CV + i] = rc >= 0 ? vk : 0.f;
      sB[2 * CV + i] = rc >= 0 ? vv : 0.f;
    }
    if (i < CP) {
      sB[3 * CV + i] = i < C ? vo : 0.f;
      sLn[i] = i < C ? vg : 0.f;
      sLn[CP + i] = i < C ? vb : 0.f;
    }
  }
  __syncthreads();

  // bias tiles, loop-invariant: the MFMAs of the projections start from them.  Transposed tiles (rows = channels): four different
  // row biases per lane; straight tiles (column = channel): one bias per lane in all four rows.
  f32x4 bqT[MT], bqN[MT], bkN[MT], bvN[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int mrow = mt * 16 + grp * 4, mcol = mt * 16 + l15;
    bqT[mt] = f32x4{sB[mrow], sB[mrow + 1], sB[mrow + 2], sB[mrow + 3]};
    bqN[mt] = f32x4{sB[mcol], sB[mcol], sB[mcol], sB[mcol]};
    bkN[mt] = f32x4{sB[CV + mcol], sB[CV + mcol], sB[CV + mcol], sB[CV + mcol]};
    bvN[mt] = f32x4{sB[2 * CV + mcol], sB[2 * CV + mcol], sB[2 * CV + mcol], sB[2 * CV + mcol]};
  }

  for (; wi < g.nWin; wi += stride) {
    const int b = wi / wpi, qh = (wi % wpi) / g.QW, qw = wi % g.QW;
    const int64_t img = (int64_t)b * g.N;
    const float* om0 = p.omega + (int64_t)b * 2 * g.N;
    const T* Ximg = X + img * C;
    T* OUTimg = OUT + img * C;

    // ---- 1. both 49xC tiles: LayerNorm (given stats) * gate weight -> LDS as T, zero padded ------------------------------
    wave_sync();
    if constexpr (PIPE) {
      tiles_finish<T, DM, CONTIG>(R, p, img, g, sLn, om0, xs, ys, xr, LDX, lane);
      __builtin_amdgcn_sched_barrier(0);          // no token load of the next window above the last use of this window's registers
      const int nx = wi + stride < g.nWin ? wi + stride : g.nWin - 1;     // clamped, never branched around: a conditional load
      tiles_issue<T, DM>(R, p, g, X, Y, nx, lane, nc, rcp_nc);                         // makes every later s_waitcnt drain to zero
      __builtin_amdgcn_sched_barrier(0);
    } else {
      load_gated_tiles<T, DM>(p, g, sLn, X, Y, om0, img, qh, qw, xs, ys, LDX, lane);
    }
    int ntok[NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) ntok[qt] = slot_token(g, qh, qw, qt * 16 + l15);
    wave_sync();

    // ---- 2./3. head by head (only ONE head's q, k, v tiles are live at a time).  Everything downstream of the projections is
    //      register-chained, so q and k are produced in BOTH orientations:
    //         transposed  q^T,k^T [channel][token] (k-slot = channel)  -> operands of S^T = k q^T
    //         straight    q,k,v   [token][channel] (k-slot = token)    -> operands of M = q^T k and of O^T = v^T P^T
    using PK = Packed<T>;
    const typename PK::type ones = PK::pack(f32x4{1.f, 1.f, 1.f, 1.f});
    typename PK::type o[MT][NT];
#pragma unroll
    for (int h = 0; h < DM::HEADS; ++h) {
      typename PK::type qT[TPH][NT], kT[TPH][NT], vN[NT][TPH];
      f32x4 Macc[TPH][TPH];
#pragma unroll
      for (int i = 0; i < TPH; ++i)
#pragma unroll
        for (int j = 0; j < TPH; ++j) Macc[i][j] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        typename PK::type qN[TPH], kN[TPH];
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          const int mt = h * TPH + mi;
          const f32x4 aq = proj_tile<T>(sWq + mt * 16 * LDW, LDW, xs + tt * 16 * LDX, LDX, CP, bqT[mt]);
          // k^T enters only S = q.k: its bias adds q.bk to every key of a query - a shift the softmax does not see - and is left out
          const f32x4 ak = proj_tile<T>(sWk + mt * 16 * LDW, LDW, ys + tt * 16 * LDX, LDX, CP, f32x4{0.f, 0.f, 0.f, 0.f});
          f32x4 nq = proj_tile<T>(xs + tt * 16 * LDX, LDX, sWq + mt * 16 * LDW, LDW, CP, bqN[mt]);
          const f32x4 nk = proj_tile<T>(ys + tt * 16 * LDX, LDX, sWk + mt * 16 * LDW, LDW, CP, bkN[mt]);
          const f32x4 nv = proj_tile<T>(ys + tt * 16 * LDX, LDX, sWv + mt * 16 * LDW, LDW, CP, bvN[mt]);
          if (tt * 16 + 15 >= LW) {                                  // compile-time: only the last token tile has dead slots
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (tt * 16 + grp * 4 + r >= LW) nq[r] = 0.f;           // tokens >= 49 must not enter M (DAL.py:1003)
          }
          qT[mi][tt] = PK::pack(aq); kT[mi][tt] = PK::pack(ak); qN[mi] = PK::pack(nq); kN[mi] = PK::pack(nk);
          vN[tt][mi] = PK::pack(nv);
        }
        // M_h += q_h^T k_h over this token tile (k-slot = token: both operands are C-layout rows)
#pragma unroll
        for (int i = 0; i < TPH; ++i)
#pragma unroll
          for (int j = 0; j < TPH; ++j) Macc[i][j] = PK::mma(qN[i], kN[j], Macc[i][j]);
      }

      // channel alpha = sigmoid(mean(M) + max(M)),  M = q_h^T k_h  (d x d)   (DAL.py:1003-1010)
      float msum = 0.f, mmax = -INFINITY;
#pragma unroll
      for (int it = 0; it < TPH; ++it)
#pragma unroll
        for (int jt = 0; jt < TPH; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + grp * 4 + r, j = jt * 16 + l15;
            if (i < D && j < D) { msum += Macc[it][jt][r]; mmax = fmaxf(mmax, Macc[it][jt][r]); }
          }
      msum = wave_reduce_dpp<OpSum>(msum);
      mmax = wave_reduce_dpp<OpMax>(mmax);
      // M was formed from q * log2(e): undo it on the two statistics.  sigmoid through v_rcp_f32 (1 ulp)
      const float alpha = __builtin_amdgcn_rcpf(1.0f + __expf(-(msum * (1.0f / (float)(D * D)) + mmax) * 0.6931471805599453f));

      // per query tile: S^T = k q^T, softmax over keys, O^T = v^T P^T
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        f32x4 s[NT];              // [key tile]: lane holds key kt*16+4*grp+r for query qt*16+l15   (base-2 logits)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mi = 0; mi < TPH; ++mi) acc = PK::mma(kT[mi][kt], qT[mi][qt], acc);
          s[kt] = acc;
        }
        // softmax over the 49 live keys (no mask, no bias: DAL.py:959,996); LW is a constant: only key tile 3 has dead rows
        // Of key tile 3 (keys 48..63) only key 48 is live - element r = 0 of lane group 0: the other three rows of that tile are
        // constants (-inf / 0) and cost no arithmetic.
        static_assert(LW == 3 * 16 + 1 && NT == 4, "softmax: 49 live keys in four 16-key tiles");
        if (RSSF_FWD_DBG & 2) {
#pragma unroll
          for (int mi = 0; mi < TPH; ++mi) o[h * TPH + mi][qt] = PK::pack(s[mi]);
          continue;
        }
        s[3][0] = grp == 0 ? s[3][0] : -INFINITY;
        float mx = s[3][0];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = rows_reduce<OpMax>(mx);
        const f32x2 mx2 = {mx, mx};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          if (RSSF_FWD_DBG & 1) continue;
          const f32x2 lo = pk_sub_lo(f32x2{s[kt][0], s[kt][1]}, mx2), hi = pk_sub_lo(f32x2{s[kt][2], s[kt][3]}, mx2);
          s[kt][0] = __builtin_amdgcn_exp2f(lo[0]); s[kt][1] = __builtin_amdgcn_exp2f(lo[1]);
          s[kt][2] = __builtin_amdgcn_exp2f(hi[0]); s[kt][3] = __builtin_amdgcn_exp2f(hi[1]);
        }
        s[3][0] = __builtin_amdgcn_exp2f(s[3][0] - mx);
        s[3][1] = 0.f; s[3][2] = 0.f; s[3][3] = 0.f;
        typename PK::type pk[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) pk[kt] = PK::pack(s[kt]);
        // Row sums on the matrix pipe (the VALU is the busy one): ones[16][key] * P^T[key][query] puts sum_key P[query][key] into
        // all four accumulator rows of the lane that owns `query` - no adds, no cross-lane reduction, and the sum is over exactly
        // the (bf16-rounded) probabilities that multiply v below.
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) sacc = PK::mma(ones, pk[kt], sacc);
        // P stays UN-normalised (values in (0, 1]): 1/sum - and alpha, o = alpha * (P v) - scale the 4 values of the O tile of this
        // lane's query instead of the 16 probabilities
        const float inv = alpha * __builtin_amdgcn_rcpf(sacc[0]);
        // O^T[dcol][query] = sum_key v[key][dcol] P^T[key][query]   (A = straight v tile: row index = dcol lane, k-slot = key)
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) acc = PK::mma(vN[kt][mi], pk[kt], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] *= inv;
          o[h * TPH + mi][qt] = PK::pack(acc);
        }
      }
    }

    // ---- 4. out-projection (transposed) + bias + residual; each lane stores 4 consecutive channels of one token ----------
    // residual values: from the raw-x tile in LDS (bf16 pipeline) or re-read from global just before the out-projection MFMAs
    // that cover their latency - not carried in registers across steps 2-3
    typename Quad<T>::raw xres[NT][CT];
    if constexpr (PIPE) {
      // the next window's gate weights, one out-projection ahead of their use (64 registers that are free from here to the next
      // tiles_finish); the stores below queue up behind them and are never waited for
      const int nx = wi + stride < g.nWin ? wi + stride : g.nWin - 1;
      __builtin_amdgcn_sched_barrier(0);
      gates_issue<T, DM>(R, p.omega + (int64_t)(nx / wpi) * 2 * g.N, g.N);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c0 = ct * 16 + grp * 4;
          xres[qt][ct] = Quad<T>::load_raw(xr + (qt * 16 + l15) * LDX + (c0 < C ? c0 : 0));       // LDS
        }
    } else if constexpr (C % 4 == 0) {
#pragma unroll
      for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c0 = ct * 16 + grp * 4;
          xres[qt][ct] = Quad<T>::load_raw(Ximg + (unsigned)((ntok[qt] >= 0 ? ntok[qt] : 0) * C + (c0 < C ? c0 : 0)));
        }
    }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      const int n = ntok[qt];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc = PK::mma_lds_a(sWo + ct * 16 * LDO, LDO, mt * 16, o[mt][qt], acc);
        const int c0 = ct * 16 + grp * 4;
        const unsigned off = (unsigned)(n * C + c0);                     // 32-bit, against the image's uniform base pointer
        if constexpr (C % 4 == 0) {
          float xr[4];
          Quad<T>::unpack(xres[qt][ct], xr);
#pragma unroll
          for (int r = 0; r < 4; ++r) xr[r] += acc[r] + sB[3 * CV + (c0 < C ? c0 : 0) + r];
          if (n >= 0 && c0 < C) Quad<T>::store(at_bytes<T>(OUTimg, off * (unsigned)sizeof(T)), xr);
        } else {
          if (n < 0) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (c0 + r < C) stf(OUTimg + off + r, ldf(Ximg + off + r) + acc[r] + sB[3 * CV + c0 + r]);
        }
      }
    }
  }
}

// device properties / per-kernel attributes are looked up once per device and cached in immutable-after-publication slots
// (no first-device-wins statics: one entry per device ordinal, written under a once_flag)
constexpr int MAX_DEVICES = 32;
int device_cus() {
  static std::once_flag once[MAX_DEVICES];
  static int cus[MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
  std::call_once(once[dev], [dev] {
    int v = 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev] = v;
  });
  return cus[dev];
}

template <typename T, typename DM>
int launch_fwd(const rssf_winattn_fwd_params* p, const Geom& g, hipStream_t st) {
  using LY = FwdLayout<T, DM>;
  static_assert(LY::BYTES <= 160 * 1024, "LDS budget");
  int blocks = (g.nWin + LY::WAVES - 1) / LY::WAVES;
  const bool contig = g.N % DM::C == 0 && g.N < (1 << 24);
  auto kern = contig ? winattn_fwd_kernel<T, DM, true> : winattn_fwd_kernel<T, DM, false>;
  // once per (instantiation, device): raise the dynamic-LDS limit, then ask the runtime how many workgroups a CU really holds
  // (registers AND LDS).  Persistent grid = exactly that many: a larger grid leaves a second, half-empty round of workgroups
  // (the LDS-only estimate of 3 per CU against the 2 the registers allow cost a third of the machine in the tail).
  static std::once_flag once[MAX_DEVICES];
  static int resident[MAX_DEVICES];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAX_DEVICES) dev = 0;
  hipError_t e = hipSuccess;
  std::call_once(once[dev], [&] {
    if (LY::BYTES > 64 * 1024) {
      e = hipFuncSetAttribute((const void*)winattn_fwd_kernel<T, DM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LY::BYTES);
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)winattn_fwd_kernel<T, DM, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LY::BYTES);
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)winattn_fwd_kernel<T, DM, true>, LY::WAVES * 64, LY::BYTES) != hipSuccess ||
        per_cu < 1)
      per_cu = 1;
    (void)hipGetLastError();
    resident[dev] = per_cu * device_cus();
  });
  if (e != hipSuccess) { set_error("winattn_fwd: cannot raise LDS limit: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  if (blocks > resident[dev]) blocks = resident[dev];
#ifdef RSSF_FWD_GRID_DIV           // timing builds only: a 1 / n grid (one workgroup per CU at n = 2: the window chain of a wave ALONE on its SIMD)
  if (blocks > resident[dev] / RSSF_FWD_GRID_DIV) blocks = resident[dev] / RSSF_FWD_GRID_DIV;
#endif
  Geom gx = g;
  gx.xcd_major = (blocks % 8 == 0 && blocks >= 8) ? 1 : 0;
  kern<<<blocks, LY::WAVES * 64, LY::BYTES, st>>>(*p, gx);
  return check_launch("winattn_fwd");
}

template <typename T>
int dispatch_fwd(const rssf_winattn_fwd_params* p, const Geom& g, hipStream_t st) {
  if (p->heads == 2 && p->C == 32) return launch_fwd<T, Dims<32, 2>>(p, g, st);   // Base  (hrnetv2_w32)
  if (p->heads == 2 && p->C == 18) return launch_fwd<T, Dims<18, 2>>(p, g, st);   // Tiny  (hrnetv2_w32s table)
  if (p->heads == 2 && p->C == 48) return launch_fwd<T, Dims<48, 2>>(p, g, st);   // Large (hrnetv2_w48)
  set_error("winattn_fwd: no kernel instantiated for C=%d heads=%d (built: 18/32/48 x 2 heads)", p->C, p->heads);
  return RSSF_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int rssf_winattn_fwd(const rssf_winattn_fwd_params* p, void* stream) {
  RSSF_REQUIRE(p, "winattn_fwd: null params");
  RSSF_REQUIRE(p->x && p->y && p->stats_x && p->stats_y && p->omega && p->ln_gamma && p->ln_beta && p->wq && p->bq &&
                   p->wk && p->bk && p->wv && p->bv && p->wo && p->bo && p->out,
               "winattn_fwd: null tensor pointer");
  RSSF_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->heads > 0 && p->window == 7,
               "winattn_fwd: bad shape B=%d H=%d W=%d C=%d heads=%d window=%d", p->B, p->H, p->W, p->C, p->heads, p->window);
  RSSF_REQUIRE(p->C % p->heads == 0, "winattn_fwd: embed_dim must be divisible by num_heads");   // DAL.py:700-702
  RSSF_REQUIRE((int64_t)p->H * p->W * p->C < ((int64_t)1 << 31), "winattn_fwd: H*W*C must stay below 2^31");
  const Geom g = make_geom(p->B, p->H, p->W, p->window);
  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == RSSF_F32) return dispatch_fwd<float>(p, g, st);
  if (p->dtype == RSSF_BF16) return dispatch_fwd<bf16_t>(p, g, st);
  set_error("winattn_fwd: unsupported dtype %d", p->dtype);
  return RSSF_ERR_UNSUPPORTED;
}
