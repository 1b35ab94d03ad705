// Implicit-GEMM convolution, "gather" form, channels-last (NHWC) activations, MFMA.
//
// One kernel serves the forward convolution AND the data gradient (dgrad = the same gather with mirrored taps,
// transposed weight slabs and, for stride-2 layers, a divisibility predicate on the source pixel).  Covers every
// convolution on the RSSFormer path: HRNet 3x3 (s1/s2) and 1x1 convs (_hrnet_rssformer.py:216-287, 361-405,
// 512-546), the neck/head 1x1 convs (hrnet_aux.py:45-49, 78-81) and MlpDWBN's fc1/fc2 and the fused
// {1x1 + 3x3 dil 6 + 3x3 dil 12} sum (ffn_block.py:219-228, 246-257) as ONE 19-tap launch.
//
// GEMM view: M = B*OH*OW output pixels, N = Cout, K = taps * Cin.  Block tile 128 (pixels) x BN (channels),
// 4 waves each 32 x BN; K is walked tap by tap in 64-byte channel chunks staged through LDS, with the next
// chunk's global loads issued before the MFMAs of the current one (register double buffering).  Epilogue:
// + bias, per-channel sum / sum-of-squares partials for the following BatchNorm (fused statistics: the
// activation is not re-read for the mean/var pass), tile transposed through LDS for 16-byte coalesced stores.
#include <string.h>
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

#ifndef RSSF_BM256_MINK
#define RSSF_BM256_MINK 1024      // smallest GEMM K (taps x padded input channels) for which a 128-wide layer takes the 256-pixel tile
#endif

namespace {

struct ConvArgs {
  const void* in;        // [B, IH, IW, Cin]
  const void* wpk;       // [ntaps][CoutP][CinP]  (k = input channel contiguous, zero padded)
  void* out;             // [B, OH, OW, Cout]
  const float* bias;     // [Cout] or null
  const void* addend;    // [B, OH, OW, Cout] added to the output (fused gradient accumulation) or null
  float* stats;          // [RSSF_BN_SLOTS][2][Cout] sum, sumsq (atomically accumulated, slot = block % slots) or null
  float* stats_ws;       // deterministic mode: per-pixel-tile partials [tiles_m][2][Cout] (plain stores; folded in tile order by
                         // stats_fold_kernel into slot 0 of `stats`) or null
  // BatchNorm-backward statistics of the layer whose output gradient this data-gradient launch produces (bf16; see HaloArgs)
  const void* bn_raw; const void* bn_res; const float* bn_ss; float* bn_sums; int bn_act;
  int B, IH, IW, Cin, OH, OW, Cout, CinP, CoutP;
  int ntiles_n, xcd_per;
  int64_t total;
  int mul, div;          // source row = (oy*mul + dy) / div   (div > 1: only when divisible)
  Taps taps;
};

// VOK: Cin is a multiple of the 16-byte vector -> branch-free staging (loads from a clamped address, invalid rows zeroed
// when they are written to LDS) so that all of a step's global loads are in flight together; any control flow around a
// load makes the compiler drain vmcnt at the join, which serialises the loads (measured: 8 dependent round trips per
// step in the weight-gradient kernel).  !VOK (the 3-channel stem, the 18-channel Small variant): element-wise gather.
// LDS staging layout of an [R rows][BK] operand tile, addressed by (row, 16-byte k-slot q = 0..3, + element within).
// bf16: two planes (k-slots {0,1} and {2,3}), rows 32 B apart inside a plane, planes 64 B further apart.  A
// ds_read_b128 is serviced in four 16-lane groups that pair rows {0-3,12-15} of one k-slot with rows {4-11} of the next
// (MI355X_MICROARCH.md, LDS): here the first set lands on the even 16-byte bank slots and the second on the odd ones, and
// the 8-lane groups of the staging ds_write_b128 (two rows x four k-slots) hit eight different slots as well - conflict
// free both ways with no row padding.  (Measured on the padded row-major tile: 43 % of the LDS cycles were conflicts.)
// f32 (parity mode): padded row-major rows, scalar fragment reads.
template <typename T> struct StageLay;
template <> struct StageLay<bf16_t> {
  static constexpr int elems(int R) { return 2 * (R * 16 + 32); }
  static __device__ __forceinline__ int off(int r, int q, int R) { return (q >> 1) * (R * 16 + 32) + r * 16 + (q & 1) * 8; }
  static __device__ __forceinline__ int frag(int r, int ks, int grp, int R) { return off(r, grp, R); }      // ks == 0 (BK = KSTEP)
};
template <> struct StageLay<float> {
  static constexpr int LDA = 16 + 4;
  static constexpr int elems(int R) { return R * LDA; }
  static __device__ __forceinline__ int off(int r, int q, int R) { return r * LDA + q * 4; }
  static __device__ __forceinline__ int frag(int r, int ks, int grp, int R) { return r * LDA + ks + grp; }
};

// DIV: data gradient of a strided convolution (source pixel = (oy + dy) / div when divisible); a template parameter so
// that the common case carries no integer division in the staging loop.
template <typename T, int BM, int BNT, bool VOK, bool DIV>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM == 256 ? 2 : (BM == 128 && BNT == 128) ? 3 : 1, 8))) conv_gather_kernel(ConvArgs a) {
  using MK = MmaK<T>;
  using SL = StageLay<T>;
  constexpr int BK = MK::BK, V = Vec<T>::N, CPR = BK / V;      // 16-byte chunks per staged row (= 4)
  // wave grid: WM along pixels x WN along channels.  The 128 x 128 tile uses 2 x 2 waves of 64 x 64 (8 fragment reads per 16
  // MFMAs instead of 10 for 32 x 128 waves); narrower tiles keep 32-pixel waves.
  constexpr int WM = (BM >= 128 && BNT == 128) ? 2 : BM / 32, WN = 4 / WM;     // 256 x 128: 2 x 2 waves of 128 x 64
  constexpr int MI = BM / (16 * WM);                             // 16-row tiles per wave along the pixels
  constexpr int WCOLS = BNT / WN;                                // channels per wave
  constexpr int NI = WCOLS / 16;
  static_assert(NI >= 1, "tile too narrow for the wave grid");
  constexpr int A_CHUNKS = BM * CPR / 256;                       // per thread
  constexpr int B_CHUNKS = (BNT * CPR + 255) / 256;
  constexpr int LDC = BNT + LdsPad<T>::X;
  constexpr int STAGE_ELEMS = SL::elems(BM) + SL::elems(BNT);        // one staging buffer (A tile + B tile); two of them
  constexpr int OUT_ELEMS = BM * LDC;
  constexpr int LDS_ELEMS = 2 * STAGE_ELEMS > OUT_ELEMS ? 2 * STAGE_ELEMS : OUT_ELEMS;
  __shared__ __attribute__((aligned(16))) T lds[LDS_ELEMS];
  __shared__ float sstat[WM * 2 * BNT];                  // one row of partials per wave row: summed in a fixed order, no LDS atomics

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave_m = wave % WM, wave_n = wave / WM;
  const int64_t M = (int64_t)a.B * a.OH * a.OW;
  const int64_t q = xcd_logical(blockIdx.x, a.xcd_per);   // channel tiles of a pixel tile, then neighbouring pixel tiles, on one XCD
  if (q >= a.total) return;
  // (the logical block index fits 32 bits - it is bounded by the grid size; 64-bit scalar divisions are a ~60-instruction serial
  // chain each, in front of the first global load of the block)
  const unsigned q32 = (unsigned)q, qt = q32 / (unsigned)a.ntiles_n;
  const int64_t m0 = (int64_t)qt * BM;
  const int n0 = (int)(q32 - qt * (unsigned)a.ntiles_n) * BNT;
  const T* IN = reinterpret_cast<const T*>(a.in);
  const T* W = reinterpret_cast<const T*>(a.wpk);

  // per-thread fixed A rows (pixels) and channel sub-chunk
  int pb[A_CHUNKS], py[A_CHUNKS], px[A_CHUNKS];      // pb: element offset of the image (32-bit: host rejects >= 2^31 elements)
  bool pv[A_CHUNKS];
#pragma unroll
  for (int i = 0; i < A_CHUNKS; ++i) {
    const int row = (tid + i * 256) / CPR;
    const int64_t m = m0 + row;
    pv[i] = m < M;
    const int mm = pv[i] ? (int)m : 0;
    pb[i] = (mm / (a.OH * a.OW)) * a.IH * a.IW * a.Cin;
    const int rem = mm % (a.OH * a.OW);
    py[i] = (rem / a.OW) * a.mul;
    px[i] = (rem % a.OW) * a.mul;
  }
  const int sub = (tid % CPR) * V;
  // FAST (bf16, vector-aligned channels, no stride predicate): hardware-bounds-checked buffer loads.  Per thread and chunk
  // the byte offset of (its pixel, tap (0,0), channel sub) and a bit mask of the taps whose source pixel lies inside the
  // image are fixed; a step then costs one add (the tap's scalar byte offset) and one select (out-of-range sentinel ->
  // the load returns zeros) per 16-byte chunk instead of ~25 VALU instructions of coordinate arithmetic and masking
  // (measured before: 4.7 VALU per MFMA, the SIMD issue port saturated at 30 % MFMA utilisation).
  constexpr bool FAST = VOK && !DIV && sizeof(T) == 2;
  constexpr unsigned OOB = 0x80000000u;                 // >= num_records (the host keeps the tensors below 2^31 bytes)
  unsigned abase[A_CHUNKS], amask[A_CHUNKS];
  unsigned bbase[B_CHUNKS];
  __amdgpu_buffer_rsrc_t rin, rw;
  if constexpr (FAST) {
    rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(IN), 0, (int)((int64_t)a.B * a.IH * a.IW * a.Cin * sizeof(T)), 0x00020000);
    rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W), 0, (int)((int64_t)a.taps.n * a.CoutP * a.CinP * sizeof(T)), 0x00020000);
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      abase[i] = (unsigned)(pb[i] + (py[i] * a.IW + px[i]) * a.Cin + sub) * (unsigned)sizeof(T);
      unsigned m = 0;
      for (int t = 0; t < a.taps.n; ++t) {
        const int sy = py[i] + a.taps.dy[t], sx = px[i] + a.taps.dx[t];
        if (pv[i] && sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW) m |= 1u << t;
      }
      amask[i] = m;
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      const int c = tid + i * 256;
      bbase[i] = (unsigned)((((c / CPR) % BNT) * a.CinP + (c % CPR) * V) * (int)sizeof(T));
    }
  }
  // byte offset of every tap, read one step ahead of its use (a scalar load from the kernel arguments at the point of use
  // put ~200 cycles of SMEM latency in front of every step's global loads)
  __shared__ int s_toff[MAX_TAPS + 2];
  if constexpr (FAST) {
    if (tid < MAX_TAPS + 2) s_toff[tid] = tid < a.taps.n ? (a.taps.dy[tid] * a.IW + a.taps.dx[tid]) * a.Cin * (int)sizeof(T) : 0;
    __syncthreads();
  }
  int toff_pref = FAST ? s_toff[0] : 0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = {0.f, 0.f, 0.f, 0.f};

  const int kchunks = a.CinP / BK;
  // Strided data gradient: a tap contributes to an output row only when (oy + dy) is divisible by the stride.  When the
  // whole pixel tile lies in ONE output row (the usual case: OW is a multiple of the tile), the taps of the wrong row
  // parity are dropped for the whole block (3 or 6 of a 3x3's 9 taps remain) instead of being staged as zeros.
  __shared__ int s_tap[MAX_TAPS];
  __shared__ int s_ntap;
  int ntap_act = a.taps.n;
  if constexpr (DIV) {
    if (tid == 0) {
      const int64_t mlast = (m0 + BM - 1 < M ? m0 + BM - 1 : M - 1);
      const int64_t r0 = m0 / a.OW, r1 = mlast / a.OW;               // global output-row index (batch folded in)
      int n = 0;
      for (int t = 0; t < a.taps.n; ++t) {
        bool any = r0 != r1;
        if (!any) { const int sy = (int)(r0 % a.OH) * a.mul + a.taps.dy[t]; any = sy >= 0 && sy % a.div == 0 && sy / a.div < a.IH; }
        if (any) s_tap[n++] = t;
      }
      s_ntap = n;
    }
    __syncthreads();
    ntap_act = s_ntap;
  }
  const int nsteps = ntap_act * kchunks;
  // Software pipeline, prefetch distance 2: the activations stream from HBM / the infinity cache (~1-2 us under load) and a
  // step is only 16 MFMAs per wave, so with one step of lookahead every step waited out a full memory latency (measured:
  // 2400 cycles per step with three resident blocks, 28 % MFMA utilisation at the MLP shape).  Two register sets hold the
  // loads of steps s+1 and s+2 while step s computes; two LDS staging buffers leave one barrier per step.
  struct StepRegs { Vec<T> ra[A_CHUNKS], rb[B_CHUNKS]; bool rok[A_CHUNKS]; };
  StepRegs R0, R1;
  int lt = 0, lkc = 0;                                    // (active tap index, channel chunk) of the next step to load
  auto load_step = [&](StepRegs& R) {
    const int t = DIV ? s_tap[lt] : lt, kc = lkc;
    if (++lkc == kchunks) { lkc = 0; ++lt; }
    if constexpr (FAST) {
      // steps past the end (the pipeline always runs an even number of steps and never branches around a load, so that the
      // compiler can count the outstanding loads exactly: with conditional loads it drained vmcnt to 0 before every LDS
      // store, i.e. it also waited for the step prefetched last) load from the out-of-range sentinel: zeros, never used.
      const bool live = t < ntap_act;
      const int toff = toff_pref + kc * BK * (int)sizeof(T);                                        // may be negative
      const int crem = a.Cin - kc * BK;                                                              // channels left in this chunk
#pragma unroll
      for (int i = 0; i < A_CHUNKS; ++i) {
        const bool ok = ((amask[i] >> t) & 1u) && sub < crem;                                        // mask has no bits >= ntaps
        R.ra[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? abase[i] + (unsigned)toff : OOB, 0, 0));
      }
      const int woff = live ? (((t * a.CoutP + n0) * a.CinP) + kc * BK) * (int)sizeof(T) : 0;
#pragma unroll
      for (int i = 0; i < B_CHUNKS; ++i)
        R.rb[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, live ? bbase[i] : OOB, woff, 0));
      toff_pref = s_toff[lt < MAX_TAPS + 1 ? lt : MAX_TAPS + 1];                                   // next step's tap
      return;
    }
    const int dy = a.taps.dy[t], dx = a.taps.dx[t];
    const int c0 = kc * BK + sub;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      int sy = py[i] + dy, sx = px[i] + dx;
      bool ok = pv[i] && sy >= 0 && sx >= 0 && c0 < a.Cin;
      if constexpr (DIV) {
        if (a.div == 2) {                                  // the only stride on this path: no integer division
          ok = ok && ((sy | sx) & 1) == 0;
          sy >>= 1; sx >>= 1;
        } else {
          ok = ok && (sy % a.div == 0) && (sx % a.div == 0);
          sy /= a.div; sx /= a.div;
        }
      }
      ok = ok && sy < a.IH && sx < a.IW;
      const int off = ok ? pb[i] + (sy * a.IW + sx) * a.Cin + c0 : 0;
      if constexpr (VOK) {
        R.ra[i].load(IN + off);                           // unconditional; masked in store_step
        R.rok[i] = ok;
      } else {
        R.ra[i].raw = {0, 0, 0, 0};
        if (ok)
#pragma unroll
          for (int e = 0; e < V; ++e) if (c0 + e < a.Cin) R.ra[i].set(e, ldf(IN + off + e));
        R.rok[i] = true;
      }
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      const int c = tid + i * 256;
      const int row = (c / CPR) % BNT;                     // surplus threads re-read a valid row (not stored)
      R.rb[i].load(W + ((int64_t)t * a.CoutP + n0 + row) * a.CinP + kc * BK + (c % CPR) * V);
    }
  };
  auto store_step = [&](const StepRegs& R, T* As, T* Bs) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      Vec<T> v = R.ra[i];
      if constexpr (!FAST) { if (!R.rok[i]) v.raw = {0, 0, 0, 0}; }     // FAST: the buffer load already returned zeros
      v.store(As + SL::off((tid + i * 256) / CPR, tid % CPR, BM));
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      const int c = tid + i * 256;
      if ((BNT * CPR) % 256 == 0 || c / CPR < BNT) R.rb[i].store(Bs + SL::off(c / CPR, c % CPR, BNT));
    }
  };
  auto compute_step = [&](const T* As, const T* Bs) {
#pragma unroll
    for (int ks = 0; ks < BK; ks += MK::KSTEP) {
      typename MK::frag fa[MI], fb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi] = MK::load(As + SL::frag(wave_m * (16 * MI) + mi * 16 + l15, ks, grp, BM));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fb[ni] = MK::load(Bs + SL::frag(wave_n * WCOLS + ni * 16 + l15, ks, grp, BNT));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = MK::mma(fa[mi], fb[ni], acc[mi][ni]);
    }
  };
  T* As0 = lds;                  T* Bs0 = As0 + SL::elems(BM);
  T* As1 = lds + STAGE_ELEMS;    T* Bs1 = As1 + SL::elems(BM);

  const int nrun = FAST ? (nsteps + 1) & ~1 : nsteps;
  if (FAST || nsteps > 0) load_step(R0);
  if (FAST || nsteps > 1) load_step(R1);
  for (int step = 0; step < nrun; step += 2) {
    store_step(R0, As0, Bs0);
    __syncthreads();             // buffer 0 complete (and every wave is done reading buffer 1's previous contents)
    if (FAST || step + 2 < nsteps) load_step(R0);
    compute_step(As0, Bs0);
    if (FAST || step + 1 < nsteps) {
      store_step(R1, As1, Bs1);  // safe: all waves passed the barrier above, i.e. finished computing step-1 out of buffer 1
      __syncthreads();
      if (FAST || step + 3 < nsteps) load_step(R1);
      compute_step(As1, Bs1);
    }
  }
  __syncthreads();               // the staging buffers become the output tile

  // ---- epilogue -------------------------------------------------------------------------------------------
  T* Cs = lds;                                            // [BM][LDC]
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int lcol = wave_n * WCOLS + ni * 16 + l15;
    const int col = n0 + lcol;
    const float bv = (a.bias && col < a.Cout) ? a.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave_m * (16 * MI) + mi * 16 + grp * 4 + r;
        const float v = acc[mi][ni][r] + bv;
        stf(Cs + row * LDC + lcol, v);
        if (m0 + row < M) { s1 += v; s2 += v * v; }
      }
    if (a.stats) {
      s1 = rows_reduce<OpSum>(s1);                         // over the four 16-lane groups: lane swaps, no LDS round trip
      s2 = rows_reduce<OpSum>(s2);
      if (grp == 0) { sstat[wave_m * 2 * BNT + lcol] = s1; sstat[(wave_m * 2 + 1) * BNT + lcol] = s2; }     // one writer per entry
    }
  }
  __syncthreads();
  if (a.stats) {
    float* slot = a.stats + (size_t)(blockIdx.x % RSSF_BN_SLOTS) * 2 * a.Cout;      // spread the same-address atomics
    float* part = a.stats_ws ? a.stats_ws + (size_t)(q / a.ntiles_n) * 2 * a.Cout : nullptr;
    for (int i = tid; i < BNT; i += 256) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += sstat[w * 2 * BNT + i]; t2 += sstat[(w * 2 + 1) * BNT + i]; }
      if (n0 + i < a.Cout) {
        if (part) { part[n0 + i] = t1; part[a.Cout + n0 + i] = t2; }
        else { atomicAdd(slot + n0 + i, t1); atomicAdd(slot + a.Cout + n0 + i, t2); }
      }
    }
  }
  T* OUT = reinterpret_cast<T*>(a.out);
  constexpr int OCPR = BNT / V;                          // 16-byte chunks per output row of the tile
  const bool ovec = (a.Cout % V) == 0;
  // fused BatchNorm-backward statistics (bf16 data-gradient launches, Cout % 8 == 0 - the dispatcher checks): a thread's channel
  // chunk is the same in every pass (256 % OCPR == 0), so its 2 x 8 partial sums stay in registers
  constexpr bool BNB = sizeof(T) == 2 && (256 % OCPR) == 0 && OCPR <= 16;
  const bool bnb = BNB && a.bn_sums != nullptr;
  const int ccl = (tid % OCPR) * V;
  float bsc[V], bsh[V], t1[V], t2[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const bool cok = bnb && n0 + ccl + e < a.Cout;
    bsc[e] = cok ? a.bn_ss[n0 + ccl + e] : 0.f;
    bsh[e] = cok ? a.bn_ss[a.Cout + n0 + ccl + e] : 0.f;
    t1[e] = 0.f; t2[e] = 0.f;
  }
  for (int c = tid; c < BM * OCPR; c += 256) {
    const int row = c / OCPR, cc = (c % OCPR) * V;
    const int64_t m = m0 + row;
    const int col = n0 + cc;
    if (m >= M || col >= a.Cout) continue;
    T* dst = OUT + m * a.Cout + col;
    const T* add = a.addend ? reinterpret_cast<const T*>(a.addend) + m * a.Cout + col : nullptr;
    if (ovec) {
      Vec<T> v, xr, xp;
      if (bnb) {
        xr.load(reinterpret_cast<const T*>(a.bn_raw) + m * a.Cout + col);
        if (a.bn_res) xp.load(reinterpret_cast<const T*>(a.bn_res) + m * a.Cout + col);
      }
      v.load(Cs + row * LDC + cc);
      if (add) {
        Vec<T> w;
        w.load(add);
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = v.get(e) + w.get(e);
        v.set_all(o);
      }
      v.store(dst);
      if (bnb) {                                           // on the values just stored: what a separate pass would read
        auto accumulate = [&](auto ACT) {                  // block-uniform activation: one specialised loop runs
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const float x = xr.get(e);
            float z = fmaf(x, bsc[e], bsh[e]);
            if (a.bn_res) z += xp.get(e);
            const float g = v.get(e);
            const float dz = decltype(ACT)::value == 1 ? (z > 0.f ? g : 0.f) : decltype(ACT)::value == 2 ? g * gelu_erf_grad(z) : g;
            t1[e] += dz; t2[e] = fmaf(dz, x, t2[e]);
          }
        };
        if (a.bn_act == 1) accumulate(std::integral_constant<int, 1>{});
        else if (a.bn_act == 2) accumulate(std::integral_constant<int, 2>{});
        else accumulate(std::integral_constant<int, 0>{});
      }
    } else {
      for (int e = 0; e < V && col + e < a.Cout; ++e) stf(dst + e, ldf(Cs + row * LDC + cc + e) + (add ? ldf(add + e) : 0.f));
    }
  }
  if constexpr (BNB) {
    if (bnb) {
      // lanes with the same channel chunk: every OCPR-th lane of a 16-lane row (rotations inside the row), then the four rows; one
      // row of partials per wave in the (now idle) output tile, summed in a fixed order, one global atomic per channel and sum
      __syncthreads();
      float* sbn = reinterpret_cast<float*>(lds);            // [4 waves][2][BNT]
      static_assert(sizeof(float) * 4 * 2 * BNT <= sizeof(T) * BM * LDC, "partials fit the output tile");
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if (OCPR <= 4) { t1[e] += dpp_mov<0x124>(t1[e]); t2[e] += dpp_mov<0x124>(t2[e]); }      // row_ror:4
        if (OCPR <= 8) { t1[e] += dpp_mov<0x128>(t1[e]); t2[e] += dpp_mov<0x128>(t2[e]); }      // row_ror:8
        t1[e] = rows_reduce<OpSum>(t1[e]); t2[e] = rows_reduce<OpSum>(t2[e]);
      }
      if (lane < OCPR) {
#pragma unroll
        for (int e = 0; e < V; ++e) { sbn[wave * 2 * BNT + ccl + e] = t1[e]; sbn[(wave * 2 + 1) * BNT + ccl + e] = t2[e]; }
      }
      __syncthreads();
      float* slot = a.bn_sums + (size_t)(blockIdx.x % RSSF_BN_BWD_SLOTS) * 2 * a.Cout;
      for (int i = tid; i < BNT; i += 256) {
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { u1 += sbn[w * 2 * BNT + i]; u2 += sbn[(w * 2 + 1) * BNT + i]; }
        if (n0 + i < a.Cout) { atomicAdd(slot + n0 + i, u1); atomicAdd(slot + a.Cout + n0 + i, u2); }
      }
    }
  }
}

// ---- weight packing: torch [Cout][Cin][kh][kw] fp32  ->  [tap][RowsP][ColsP] T, k-contiguous, zero padded ------------
// transpose == 0 (forward):  rows = Cout, cols = Cin, tap t <- kernel position kpos[t] of source conv src[t]
// transpose == 1 (dgrad):    rows = Cin,  cols = Cout
struct PackArgs {
  const float* w[3];     // up to 3 source convs (the fused MLP sum); w[s] has kernel size ks[s] x ks[s]
  int ks[3];
  int nsrc;
  int src_of_tap[MAX_TAPS];
  int kpos_of_tap[MAX_TAPS];
  int alias[MAX_TAPS][4];      // {src, kpos} x 2 further kernel positions summed into the tap's slab (-1: none)
  int ntaps, Cout, Cin, RowsP, ColsP, transpose;
};
template <typename T>
__global__ void __launch_bounds__(256) pack_weights_kernel(PackArgs a, T* out) {
  const int64_t total = (int64_t)a.ntaps * a.RowsP * a.ColsP;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % a.ColsP), row = (int)((i / a.ColsP) % a.RowsP), t = (int)(i / ((int64_t)a.ColsP * a.RowsP));
    const int co = a.transpose ? col : row, ci = a.transpose ? row : col;
    float v = 0.f;
    if (co < a.Cout && ci < a.Cin) {
      const int s = a.src_of_tap[t], kk = a.ks[s] * a.ks[s];
      v = a.w[s][((int64_t)co * a.Cin + ci) * kk + a.kpos_of_tap[t]];
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const int s2 = a.alias[t][e];
        if (s2 >= 0) v += a.w[s2][((int64_t)co * a.Cin + ci) * (a.ks[s2] * a.ks[s2]) + a.alias[t][e + 1]];
      }
    }
    stf(out + i, v);
  }
}

// Batched packing, one launch for every convolution of the model.  A block owns a (co, ci) TILE of one job with ALL its taps: it reads
// the tile's source rows once - for a fixed output channel the `ci x k x k` floats of the tile are contiguous - into LDS and writes
// the tap slabs from there.  (The first form was indexed by OUTPUT element: consecutive threads read the source with a 36-byte
// stride and the nine taps of a pixel sat in nine different blocks - 412 MB of line fetches for 120 MB of weights, 250 us per step.)
// Tile: 32 channels along the packed row (ci for the forward layout, co for the transposed one: 64-byte runs) x as many of the
// other as fit 10 KB of LDS with all kernel positions (16 for 1x1, 8 for 3x3, 4 for the 19 positions of MlpDWBN's sum): the pass
// is latency-bound - a global round trip, a barrier, the stores - and lives on resident blocks per CU (a 39 KB tile of 512 pairs
// for every job left four: 221 us).  Pad rows / columns of the packed image lie inside the tiles of the last row / column: zeros.
constexpr int PACK_LDS_FLOATS = 2560;
// (co, ci) pairs of a tile: 512 for 1x1, 256 for 3x3, 128 for the 19 positions of the MLP sum, ... down to one 32-channel row (<= 80 positions)
__host__ __device__ inline int pack_tile_pairs(int kk_total) {
  int p = 512;
  while (p > 32 && p * kk_total > PACK_LDS_FLOATS) p >>= 1;
  return p;
}
template <typename T>
__global__ void __launch_bounds__(256) pack_batch_kernel(const rssf_pack_job* __restrict__ jobs, const int* __restrict__ block_map) {
  const rssf_pack_job& j = jobs[block_map[2 * blockIdx.x]];
  const unsigned tile = (unsigned)block_map[2 * blockIdx.x + 1];
  unsigned kk[3], kk_total = 0;
  for (int s = 0; s < 3; ++s) { kk[s] = (unsigned)(j.ks[s] * j.ks[s]); if (s < j.nsrc) kk_total += kk[s]; }
  const unsigned pairs = (unsigned)pack_tile_pairs((int)kk_total), other = pairs >> 5;       // 32 x `other` channels
  const unsigned TC = j.transpose ? 32u : other, TI = j.transpose ? other : 32u;
  const unsigned ci_p = (unsigned)(j.transpose ? j.rows_p : j.cols_p);
  const unsigned tiles_i = (ci_p + TI - 1) / TI;
  const unsigned co0 = (tile / tiles_i) * TC, ci0 = (tile % tiles_i) * TI;
  const unsigned cout = (unsigned)j.cout, cin = (unsigned)j.cin;
  __shared__ float sw[PACK_LDS_FLOATS + 3 * 32];
  // the job's tap tables, once per block (read through the job pointer inside the tap loop they were two dependent scalar-memory
  // round trips per tap in front of the stores)
  __shared__ int stap[RSSF_MAX_TAPS][6];
  if (threadIdx.x < RSSF_MAX_TAPS * 6) {
    const int t = threadIdx.x / 6, f = threadIdx.x % 6;
    stap[t][f] = f == 0 ? j.src_of_tap[t] : f == 1 ? j.kpos_of_tap[t] : j.alias_of_tap[t][f - 2];
  }
  // per source: [co][ci][k] with an ODD row pitch (the transposed write phase walks co: an even pitch would put 32 lanes on few banks)
  unsigned sbase[3], pitch[3];
  unsigned base = 0;
  for (int s = 0; s < 3; ++s) {
    pitch[s] = (TI * kk[s]) | 1u;
    sbase[s] = base;
    if (s < j.nsrc) base += TC * pitch[s];
  }
  for (int s = 0; s < j.nsrc; ++s) {
    const unsigned span = TI * kk[s];
    const float* __restrict__ w = j.w[s];
    // e -> (co, r = ci * kk + k): quotients of small integers through a float reciprocal (exact below 2^22: + 0.5 keeps the product
    // off the integer boundaries); all of a thread's loads are in flight together (a loop over co serialised the round trips: 339 us)
    const float inv_span = 1.0f / (float)span, inv_kk = 1.0f / (float)kk[s];
    // all of a thread's loads first, then its LDS stores: as one loop (load, store, next) the round trips ran one after the other
    constexpr int NV = (PACK_LDS_FLOATS + 255) / 256;
    float v[NV];
    unsigned dst[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const unsigned e = threadIdx.x + (unsigned)k * 256u;
      const unsigned co = (unsigned)(((float)e + 0.5f) * inv_span), r = e - co * span;
      const unsigned ci = (unsigned)(((float)r + 0.5f) * inv_kk);
      const unsigned gco = co0 + co;
      const bool in = e < TC * span;
      dst[k] = in ? sbase[s] + co * pitch[s] + r : 0xffffffffu;
      v[k] = (in && gco < cout && ci0 + ci < cin) ? w[(gco * cin + ci0) * kk[s] + r] : 0.f;      // contiguous in r
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (dst[k] != 0xffffffffu) sw[dst[k]] = v[k];
  }
  __syncthreads();
  T* out = reinterpret_cast<T*>(j.out);
  const unsigned rows_p = (unsigned)j.rows_p, cols_p = (unsigned)j.cols_p;
  for (int t = 0; t < j.ntaps; ++t) {
    const int s = stap[t][0];
    const unsigned kp = (unsigned)stap[t][1];
    const int s2 = stap[t][2], s3 = stap[t][4];
    const unsigned kp2 = (unsigned)stap[t][3], kp3 = (unsigned)stap[t][5];
    for (unsigned e = threadIdx.x; e < pairs; e += 256) {
      // consecutive threads: consecutive columns of the packed row
      const unsigned co = j.transpose ? (e & 31u) : (e >> 5), ci = j.transpose ? (e >> 5) : (e & 31u);
      const unsigned row = j.transpose ? ci0 + ci : co0 + co, col = j.transpose ? co0 + co : ci0 + ci;
      float v = sw[sbase[s] + co * pitch[s] + ci * kk[s] + kp];
      if (s2 >= 0) v += sw[sbase[s2] + co * pitch[s2] + ci * kk[s2] + kp2];
      if (s3 >= 0) v += sw[sbase[s3] + co * pitch[s3] + ci * kk[s3] + kp3];
      if (row < rows_p && col < cols_p) stf(out + ((unsigned)t * rows_p + row) * cols_p + col, v);
    }
  }
}

int pick_bn(int cout) { return cout <= 32 ? 32 : cout <= 64 ? 64 : 128; }
}  // namespace

// Deterministic statistics, second level: out[col] = sum over the `tiles` rows of ws[tiles][ncol] in a fixed order (each
// thread walks its rows in sequence, the 8 row groups are combined in sequence).  ncol = 2*C; out = slot 0 of a zeroed
// slotted statistics buffer, so the consumers (which sum the slots) are unchanged.
namespace rssf {
namespace cv {
__global__ void __launch_bounds__(256) stats_fold_kernel(const float* __restrict__ ws, int tiles, int ncol, float* __restrict__ out) {
  __shared__ float part[8][32];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  float acc = 0.f;
  if (col < ncol)
    for (int t = rg; t < tiles; t += 8) acc += ws[(size_t)t * ncol + col];
  part[rg][threadIdx.x & 31] = acc;
  __syncthreads();
  if (rg == 0 && col < ncol) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += part[g][threadIdx.x];
    out[col] = t;
  }
}
int launch_stats_fold(const float* ws, int64_t tiles, int C, float* stats, hipStream_t st) {
  stats_fold_kernel<<<(2 * C + 31) / 32, 256, 0, st>>>(ws, (int)tiles, 2 * C, stats);
  return check_launch("stats_fold");
}
}  // namespace cv
}  // namespace rssf

namespace {

// Tile choice: the widest tile that still gives the chip >= 2 blocks per CU; small feature maps (32x32, 16x16 at
// B=16) otherwise run on a quarter of the CUs.
void pick_tile(int64_t M, int cout, int& bm, int& bnt) {
  const int bmax = pick_bn(cout);
  const int cand[4][2] = {{128, bmax}, {64, bmax}, {64, bmax > 32 ? bmax / 2 : 32}, {64, 32}};
  int64_t best = -1;
  for (int i = 0; i < 4; ++i) {
    const int64_t blocks = ((M + cand[i][0] - 1) / cand[i][0]) * ((cout + cand[i][1] - 1) / cand[i][1]);
    if (blocks >= 512) { bm = cand[i][0]; bnt = cand[i][1]; return; }
    if (blocks > best) { best = blocks; bm = cand[i][0]; bnt = cand[i][1]; }
  }
}

template <typename T>
int launch_conv(ConvArgs a, hipStream_t st) {
  const int64_t M = (int64_t)a.B * a.OH * a.OW;
  int bm, bnt;
  pick_tile(M, a.Cout, bm, bnt);
  // many-tap 128-channel layers (the MLP's 17-tap sum): a 256-pixel tile halves the weight-slab traffic per MFMA and
  // makes the wave tile 128 x 64 (LDS fragment reads per MFMA 0.375 instead of 0.5)
  constexpr int bm256_mink = RSSF_BM256_MINK;
  if (sizeof(T) == 2 && bm == 128 && bnt == 128 && a.taps.n * a.CinP >= bm256_mink && a.div == 1 && (a.Cin % Vec<T>::N) == 0 &&
      (M + 255) / 256 * ((a.Cout + 127) / 128) >= 512)
    bm = 256;
  a.ntiles_n = (a.Cout + bnt - 1) / bnt;
  a.total = ((M + bm - 1) / bm) * a.ntiles_n;
  a.xcd_per = xcd_per(a.total);
  dim3 grid((unsigned)a.xcd_per * 8);
  const bool vok = (a.Cin % Vec<T>::N) == 0;
#define RSSF_CONV(BMv, BNv)                                                   \
  do {                                                                        \
    if (a.div > 1) {                                                          \
      if (vok) conv_gather_kernel<T, BMv, BNv, true, true><<<grid, 256, 0, st>>>(a);    \
      else conv_gather_kernel<T, BMv, BNv, false, true><<<grid, 256, 0, st>>>(a);       \
    } else {                                                                  \
      if (vok) conv_gather_kernel<T, BMv, BNv, true, false><<<grid, 256, 0, st>>>(a);   \
      else conv_gather_kernel<T, BMv, BNv, false, false><<<grid, 256, 0, st>>>(a);      \
    }                                                                         \
  } while (0)
  if (bm == 256) {                                    // deep-K 128-channel layers only (bf16, vector-aligned, unit stride)
    if (vok && a.div == 1) conv_gather_kernel<T, 256, 128, true, false><<<grid, 256, 0, st>>>(a);
    else { set_error("conv_gather: 256-pixel tile chosen for an unsupported layer"); return RSSF_ERR_LAUNCH; }
  } else
  if (bm == 128) { if (bnt == 32) RSSF_CONV(128, 32); else if (bnt == 64) RSSF_CONV(128, 64); else RSSF_CONV(128, 128); }
  else           { if (bnt == 32) RSSF_CONV(64, 32);  else if (bnt == 64) RSSF_CONV(64, 64);  else RSSF_CONV(64, 128); }
#undef RSSF_CONV
  const int rc = check_launch("conv_gather");
  if (rc || !(a.stats && a.stats_ws)) return rc;
  return launch_stats_fold(a.stats_ws, a.total / a.ntiles_n, a.Cout, a.stats, st);
}

}  // namespace

extern "C" int rssf_conv_tile_n(int cout) { return pick_bn(cout); }

extern "C" int rssf_conv_pack(const float* w0, const float* w1, const float* w2, const int* ksizes, int nsrc,
                              const int* src_of_tap, const int* kpos_of_tap, const int* alias_of_tap, int ntaps, int Cout, int Cin, int transpose,
                              void* out, int dtype, void* stream) {
  RSSF_REQUIRE(w0 && ksizes && src_of_tap && kpos_of_tap && out && nsrc >= 1 && nsrc <= 3 && ntaps >= 1 && ntaps <= MAX_TAPS,
               "conv_pack: bad arguments");
  PackArgs a;
  a.w[0] = w0; a.w[1] = w1; a.w[2] = w2;
  for (int i = 0; i < 3; ++i) a.ks[i] = i < nsrc ? ksizes[i] : 1;
  a.nsrc = nsrc;
  for (int t = 0; t < ntaps; ++t) {
    a.src_of_tap[t] = src_of_tap[t]; a.kpos_of_tap[t] = kpos_of_tap[t];
    for (int e = 0; e < 4; ++e) {
      a.alias[t][e] = alias_of_tap ? alias_of_tap[t * 4 + e] : -1;
      RSSF_REQUIRE((e & 1) || a.alias[t][e] < nsrc, "conv_pack: alias source out of range");
    }
  }
  a.ntaps = ntaps; a.Cout = Cout; a.Cin = Cin; a.transpose = transpose;
  const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
  const int bk = dtype == RSSF_BF16 ? MmaK<bf16_t>::BK : MmaK<float>::BK;
  a.RowsP = (rows + pick_bn(rows) - 1) / pick_bn(rows) * pick_bn(rows);
  a.ColsP = (cols + bk - 1) / bk * bk;
  const int64_t total = (int64_t)ntaps * a.RowsP * a.ColsP;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) pack_weights_kernel<float><<<blocks, 256, 0, st>>>(a, (float*)out);
  else if (dtype == RSSF_BF16) pack_weights_kernel<bf16_t><<<blocks, 256, 0, st>>>(a, (bf16_t*)out);
  else { set_error("conv_pack: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("conv_pack");
}

extern "C" int rssf_conv_packed_rows(int rows) { return (rows + pick_bn(rows) - 1) / pick_bn(rows) * pick_bn(rows); }
extern "C" int rssf_conv_packed_cols(int cols, int dtype) {
  const int bk = dtype == RSSF_BF16 ? MmaK<bf16_t>::BK : MmaK<float>::BK;
  return (cols + bk - 1) / bk * bk;
}
extern "C" int rssf_conv_pack_job_blocks(int rows_p, int cols_p, int transpose, int kk_total) {
  if (kk_total < 1 || 32 * kk_total > PACK_LDS_FLOATS) return 0;      // too many kernel positions for the batched form: pack it singly
  const int co_p = transpose ? cols_p : rows_p, ci_p = transpose ? rows_p : cols_p;
  const int other = pack_tile_pairs(kk_total) / 32;
  const int tc = transpose ? 32 : other, ti = transpose ? other : 32;
  return ((co_p + tc - 1) / tc) * ((ci_p + ti - 1) / ti);
}
extern "C" int rssf_conv_pack_batch(const rssf_pack_job* jobs, const int* block_map, int nblocks, int dtype, void* stream) {
  RSSF_REQUIRE(jobs && block_map && nblocks > 0, "conv_pack_batch: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) pack_batch_kernel<float><<<nblocks, 256, 0, st>>>(jobs, block_map);
  else if (dtype == RSSF_BF16) pack_batch_kernel<bf16_t><<<nblocks, 256, 0, st>>>(jobs, block_map);
  else { set_error("conv_pack_batch: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("conv_pack_batch");
}

extern "C" int64_t rssf_conv_packed_elems(int ntaps, int rows, int cols, int dtype) {
  const int bk = dtype == RSSF_BF16 ? MmaK<bf16_t>::BK : MmaK<float>::BK;
  const int64_t rp = (rows + pick_bn(rows) - 1) / pick_bn(rows) * pick_bn(rows);
  const int64_t cp = (cols + bk - 1) / bk * bk;
  return (int64_t)ntaps * rp * cp;
}

extern "C" int rssf_conv_gather(const void* in, const void* wpk, void* out, const float* bias, float* stats, int B, int IH,
                                int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy,
                                const int* dx, int dtype, void* stream) {
  return rssf_conv_gather_add(in, wpk, out, bias, stats, nullptr, nullptr, B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx, dtype,
                              stream);
}

// upper bound over every tile shape the launchers may choose: 64-pixel GEMM tiles, 4 x 16-pixel halo tiles
extern "C" int64_t rssf_conv_stats_workspace_elems(int B, int OH, int OW, int Cout) {
  const int64_t gemm_tiles = ((int64_t)B * OH * OW + 63) / 64;
  const int64_t halo_tiles = (int64_t)B * ((OH + 3) / 4) * ((OW + 15) / 16);
  return (gemm_tiles > halo_tiles ? gemm_tiles : halo_tiles) * 2 * Cout;
}

namespace {
struct BnBwdStats {            // see HaloArgs::bn_* (conv.hip.h)
  const void* raw; const void* res; const float* ss; float* sums; int act;
};
bool halo_path(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx, int dtype) {
  return dtype == RSSF_BF16 && (int64_t)B * OH * OW * Cout < ((int64_t)1 << 30) && halo_eligible(IH, IW, Cin, OH, OW, mul, div, ntaps, dy, dx);
}
struct PreAct {              // see HaloArgs::pre_* (conv.hip.h)
  const float* stats; const float* gamma; const float* beta; float* rmean; float* rvar; float* mi; float* ss;
  float n, momentum, eps; int training, act;
};
HaloArgs make_halo(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend, float* stats_ws,
                   const BnBwdStats* bn, const PreAct* pre, int B, int H, int W, int Cin, int Cout, int CinP, int CoutP, const int* dy, const int* dx) {
  HaloArgs h;
  memset(&h, 0, sizeof(h));
  if (pre) {
    h.pre_stats = pre->stats; h.pre_gamma = pre->gamma; h.pre_beta = pre->beta; h.pre_rmean = pre->rmean; h.pre_rvar = pre->rvar;
    h.pre_mi = pre->mi; h.pre_ss = pre->ss; h.pre_n = pre->n; h.pre_momentum = pre->momentum; h.pre_eps = pre->eps;
    h.pre_training = pre->training; h.pre_act = pre->act;
  }
  h.in = (const bf16_t*)in; h.wpk = (const bf16_t*)wpk; h.out = (bf16_t*)out; h.bias = bias; h.stats = stats; h.stats_ws = stats_ws;
  h.addend = (const bf16_t*)addend;
  if (bn) { h.bn_raw = (const bf16_t*)bn->raw; h.bn_res = (const bf16_t*)bn->res; h.bn_ss = bn->ss; h.bn_sums = bn->sums; h.bn_act = bn->act; }
  h.B = B; h.H = H; h.W = W; h.Cin = Cin; h.Cout = Cout; h.CinP = CinP; h.CoutP = CoutP;
  for (int t = 0; t < 9; ++t) { h.dy[t] = dy[t]; h.dx[t] = dx[t]; }
  return h;
}
int conv_gather_impl(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend, float* stats_ws,
                     const BnBwdStats* bn, const PreAct* pre, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps,
                     const int* dy, const int* dx, int dtype_flags, void* stream) {
  // RSSF_CONV_GENERIC OR-ed into the dtype argument: the generic gather / halo kernels only (what the shape-specialised kernels are
  // held against in the parity tests); an explicit argument of the call, not process state
  const bool generic = (dtype_flags & RSSF_CONV_GENERIC) != 0;
  const int dtype = dtype_flags & ~RSSF_CONV_GENERIC;
  RSSF_REQUIRE(in && wpk && out && dy && dx && B > 0 && IH > 0 && IW > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0 &&
                   ntaps >= 1 && ntaps <= MAX_TAPS && mul >= 1 && div >= 1,
               "conv_gather: bad arguments");
  RSSF_REQUIRE((int64_t)B * IH * IW * Cin < ((int64_t)1 << 30) && (int64_t)B * OH * OW < ((int64_t)1 << 31),
               "conv_gather: input tensors of 2^30 or more elements are not supported (32-bit byte offsets, buffer bounds)");
  ConvArgs a;
  a.in = in; a.wpk = wpk; a.out = out; a.bias = bias; a.stats = stats; a.addend = addend; a.stats_ws = stats ? stats_ws : nullptr;
  a.B = B; a.IH = IH; a.IW = IW; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
  a.mul = mul; a.div = div;
  a.taps.n = ntaps;
  for (int t = 0; t < ntaps; ++t) { a.taps.dy[t] = dy[t]; a.taps.dx[t] = dx[t]; }
  const int bnt = pick_bn(Cout);
  const int bk = dtype == RSSF_BF16 ? MmaK<bf16_t>::BK : MmaK<float>::BK;
  a.CoutP = (Cout + bnt - 1) / bnt * bnt;
  a.CinP = (Cin + bk - 1) / bk * bk;
  hipStream_t st = (hipStream_t)stream;
#ifndef RSSF_ROWS32_DISABLE        // (A/B builds: tools/ab_lib_flags.sh)
  // 32 -> 32 channels (the full-resolution branch's BasicBlocks): the row stream with the weights in registers (conv_rows32.hip)
  if (!generic && dtype == RSSF_BF16 && !a.stats_ws && (!pre || pre->act <= 1) && (!bn || bn->act <= 1) &&
      rows32_eligible(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx)) {
    const bool mirror = dy[0] > 0;
    if (mirror ? (!pre && !stats && !bias) : (!addend && !bn)) {
      PwPre pp;
      if (pre) pp = PwPre{pre->stats, pre->gamma, pre->beta, pre->rmean, pre->rvar, pre->mi, pre->ss, pre->n, pre->momentum, pre->eps, pre->training, pre->act};
      return launch_rows32(in, wpk, out, bias, stats, addend, bn ? bn->raw : nullptr, bn ? bn->res : nullptr, bn ? bn->ss : nullptr,
                           bn ? bn->sums : nullptr, bn ? bn->act : 0, pre ? &pp : nullptr, B, IH, IW, mirror, st);
    }
  }
#endif
  if (halo_path(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx, dtype)) {
    const bool fused = bn && (Cout % 8) == 0;               // the 16-byte-row epilogue carries the statistics
    const HaloArgs h = make_halo(in, wpk, out, bias, stats, addend, a.stats_ws, fused ? bn : nullptr, pre, B, IH, IW, Cin, Cout, a.CinP, a.CoutP, dy, dx);
    const int rc = launch_halo(h, st);
    if (rc || !bn || fused) return rc;
    return rssf_bn_bwd_reduce(out, bn->raw, bn->ss, bn->res, bn->sums, (int64_t)B * OH * OW, Cout, bn->act, nullptr, dtype, stream);
  }
#ifndef RSSF_STEM_FWD_DISABLE      // (A/B builds: tools/ab_lib_flags.sh)
  if (!generic && dtype == RSSF_BF16 && !pre && !bias && !addend && !bn && !a.stats_ws && stem_fwd_eligible(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx))
    return launch_stem_fwd(in, wpk, out, stats, B, IH, IW, OH, OW, a.CinP, a.CoutP, st);
#endif
#ifndef RSSF_DGRAD_S2_DISABLE      // (A/B builds: tools/ab_lib_flags.sh)
  if (!generic && dtype == RSSF_BF16 && !pre && !bias && !stats && (!bn || (Cout % 8) == 0) &&
      dgrad_s2_eligible(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx))
    return launch_dgrad_s2(in, wpk, out, addend, bn ? bn->raw : nullptr, bn ? bn->res : nullptr, bn ? bn->ss : nullptr, bn ? bn->sums : nullptr, bn ? bn->act : 0,
                           B, IH, IW, Cin, Cout, a.CinP, a.CoutP, st);
#endif
  if (dtype == RSSF_BF16 && pre && !bn && !a.stats_ws && !addend && pw_preact_eligible(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx)) {
    // (the stream kernel is the only one with a pre-activation operand for this shape: RSSF_CONV_GENERIC does not apply)
    const PwPre pp = {pre->stats, pre->gamma, pre->beta, pre->rmean, pre->rvar, pre->mi, pre->ss, pre->n, pre->momentum, pre->eps, pre->training, pre->act};
    return launch_pw(in, wpk, out, bias, stats, nullptr, nullptr, nullptr, nullptr, 0, B, IH, IW, Cin, Cout, a.CinP, a.CoutP, st, &pp);
  }
  if (!generic && dtype == RSSF_BF16 && !pre && !a.stats_ws && (!addend || pw_addend_eligible(Cin, Cout)) &&
      pw_eligible(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx))
    return launch_pw(in, wpk, out, bias, stats, bn ? bn->raw : nullptr, bn ? bn->res : nullptr, bn ? bn->ss : nullptr, bn ? bn->sums : nullptr,
                     bn ? bn->act : 0, B, IH, IW, Cin, Cout, a.CinP, a.CoutP, st, nullptr, addend);
  if (!generic && dtype == RSSF_BF16 && !pre && !a.stats_ws && taps128_eligible(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps))
    return launch_taps128(in, wpk, out, bias, stats, addend, bn ? bn->raw : nullptr, bn ? bn->res : nullptr, bn ? bn->ss : nullptr,
                          bn ? bn->sums : nullptr, bn ? bn->act : 0, B, IH, IW, Cin, Cout, a.CinP, a.CoutP, ntaps, dy, dx, st);
  if (pre) { set_error("conv_gather_preact: no kernel with a pre-activation input for this shape (ask rssf_conv_gather_preact_supported)"); return RSSF_ERR_UNSUPPORTED; }
  const bool gfused = bn && dtype == RSSF_BF16 && (Cout % 8) == 0;          // the gather kernels' 16-byte-row epilogue carries them too
  a.bn_raw = gfused ? bn->raw : nullptr; a.bn_res = gfused ? bn->res : nullptr; a.bn_ss = gfused ? bn->ss : nullptr;
  a.bn_sums = gfused ? bn->sums : nullptr; a.bn_act = gfused ? bn->act : 0;
  int rc;
  if (dtype == RSSF_F32) rc = launch_conv<float>(a, st);
  else if (dtype == RSSF_BF16) rc = launch_conv<bf16_t>(a, st);
  else { set_error("conv_gather: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  if (rc || !bn || gfused) return rc;
  // no kernel with a statistics epilogue for this shape: the separate pass
  return rssf_bn_bwd_reduce(out, bn->raw, bn->ss, bn->res, bn->sums, (int64_t)B * OH * OW, Cout, bn->act, nullptr, dtype, stream);
}
}  // namespace

extern "C" int rssf_conv_gather_add(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend,
                                    float* stats_ws, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps,
                                    const int* dy, const int* dx, int dtype, void* stream) {
  return conv_gather_impl(in, wpk, out, bias, stats, addend, stats_ws, nullptr, nullptr, B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx, dtype,
                          stream);
}

extern "C" int rssf_conv_gather_bnbwd(const void* in, const void* wpk, void* out, const void* addend, const void* bn_raw,
                                      const void* bn_res_pre, const float* bn_scale_shift, int bn_act, float* bn_sums, int B, int IH,
                                      int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx,
                                      int dtype, void* stream) {
  RSSF_REQUIRE(bn_raw && bn_scale_shift && bn_sums && bn_act >= 0 && bn_act <= 2, "conv_gather_bnbwd: bad BatchNorm arguments");
  const BnBwdStats bn = {bn_raw, bn_res_pre, bn_scale_shift, bn_sums, bn_act};
  return conv_gather_impl(in, wpk, out, nullptr, nullptr, addend, nullptr, &bn, nullptr, B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx, dtype,
                          stream);
}

extern "C" int rssf_conv_gather_preact_supported(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps,
                                                 const int* dy, const int* dx, int dtype) {
  if (!dy || !dx) return 0;
  if (Cin <= 256 && halo_path(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx, dtype)) return 1;
  return dtype == RSSF_BF16 && pw_preact_eligible(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx) ? 1 : 0;      // MlpDWBN's fc2
}

extern "C" int rssf_conv_gather_preact(const void* in_raw, const float* pre_stats, const float* pre_gamma, const float* pre_beta,
                                       float* pre_running_mean, float* pre_running_var, float* pre_mean_invstd, float* pre_scale_shift,
                                       double pre_n, float pre_momentum, float pre_eps, int pre_training, int pre_act, const void* wpk,
                                       void* out, const float* bias, float* stats, float* stats_ws, int B, int IH, int IW, int Cin, int OH,
                                       int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx, int dtype, void* stream) {
  RSSF_REQUIRE(pre_gamma && pre_beta && pre_mean_invstd && pre_scale_shift && pre_act >= 0 && pre_act <= 2,
               "conv_gather_preact: bad pre-activation arguments");
  RSSF_REQUIRE(pre_training ? (pre_stats != nullptr && pre_n >= 1) : (pre_running_mean && pre_running_var),
               "conv_gather_preact: missing statistics of the producer's BatchNorm");
  const PreAct pre = {pre_stats, pre_gamma, pre_beta, pre_running_mean, pre_running_var, pre_mean_invstd, pre_scale_shift, (float)pre_n,
                      pre_momentum, pre_eps, pre_training, pre_act};
  return conv_gather_impl(in_raw, wpk, out, bias, stats, nullptr, stats_ws, nullptr, &pre, B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx,
                          dtype, stream);
}

// ---- grouped 3x3 launches (rssf.h "Grouped launches") ----------------------------------------------------------------------
extern "C" int rssf_conv3x3_group(const rssf_conv3x3_item* items, int n, int mirrored, int dtype, void* stream) {
  RSSF_REQUIRE(items && n >= 1 && (mirrored == 0 || mirrored == 1), "conv3x3_group: bad arguments");
  int dy[9], dx[9];
  for (int t = 0; t < 9; ++t) { dy[t] = (mirrored ? -1 : 1) * (t / 3 - 1); dx[t] = (mirrored ? -1 : 1) * (t % 3 - 1); }
  bool grouped = n >= 2 && n <= RSSF_GROUP_MAX && dtype == RSSF_BF16;
  for (int i = 0; i < n; ++i) {
    const rssf_conv3x3_item& it = items[i];
    RSSF_REQUIRE(it.in && it.wpk && it.out && it.B > 0 && it.H > 0 && it.W > 0 && it.Cin > 0 && it.Cout > 0, "conv3x3_group: bad item %d", i);
    RSSF_REQUIRE(!it.bn_sums || (it.bn_raw && it.bn_ss && it.bn_act >= 0 && it.bn_act <= 2), "conv3x3_group: bad BatchNorm arguments (item %d)", i);
    RSSF_REQUIRE(!it.pre_ss || (it.pre_gamma && it.pre_beta && it.pre_mean_invstd && it.pre_act >= 0 && it.pre_act <= 2 && !it.addend && !it.bn_sums &&
                                (it.pre_training ? (it.pre_stats != nullptr && it.pre_n >= 1) : (it.pre_running_mean && it.pre_running_var))),
                 "conv3x3_group: bad pre-activation arguments (item %d)", i);
    grouped = grouped && (it.Cout % 8) == 0 && halo_path(it.B, it.H, it.W, it.Cin, it.H, it.W, it.Cout, 1, 1, 9, dy, dx, dtype) &&
              (int64_t)it.B * it.H * it.W * it.Cin < ((int64_t)1 << 30) && (it.pre_ss != nullptr) == (items[0].pre_ss != nullptr) &&
              (!it.pre_ss || (!mirrored && it.Cin <= 256));
  }
  if (grouped) {
    HaloArgs hs[RSSF_GROUP_MAX];
    int nh = 0;
    for (int i = 0; i < n; ++i) {
      const rssf_conv3x3_item& it = items[i];
      const BnBwdStats bn = {it.bn_raw, it.bn_res, it.bn_ss, it.bn_sums, it.bn_act};
      const PreAct pre = {it.pre_stats, it.pre_gamma, it.pre_beta, it.pre_running_mean, it.pre_running_var, it.pre_mean_invstd, it.pre_ss,
                          (float)it.pre_n, it.pre_momentum, it.pre_eps, it.pre_training, it.pre_act};
#ifndef RSSF_ROWS32_DISABLE
      // a 32 -> 32 channel member runs on the row-stream kernel, in a launch of its own (conv_gather_impl picks it), ahead of the group
      if (rows32_eligible(it.B, it.H, it.W, it.Cin, it.H, it.W, it.Cout, 1, 1, 9, dy, dx) && (!it.pre_ss || it.pre_act <= 1) && (!it.bn_sums || it.bn_act <= 1) &&
          (mirrored ? !it.stats : (!it.addend && !it.bn_sums))) {
        const int rc = conv_gather_impl(it.in, it.wpk, it.out, nullptr, it.stats, it.addend, nullptr, it.bn_sums ? &bn : nullptr,
                                        it.pre_ss ? &pre : nullptr, it.B, it.H, it.W, it.Cin, it.H, it.W, it.Cout, 1, 1, 9, dy, dx, dtype, stream);
        if (rc) return rc;
        continue;
      }
#endif
      const int bnt = pick_bn(it.Cout);
      hs[nh++] = make_halo(it.in, it.wpk, it.out, nullptr, it.stats, it.addend, nullptr, it.bn_sums ? &bn : nullptr, it.pre_ss ? &pre : nullptr, it.B,
                           it.H, it.W, it.Cin, it.Cout, (it.Cin + 31) / 32 * 32, (it.Cout + bnt - 1) / bnt * bnt, dy, dx);
    }
    if (nh == 0) return RSSF_OK;
    if (nh == 1) return launch_halo(hs[0], (hipStream_t)stream);
    return launch_halo_group(hs, nh, (hipStream_t)stream);
  }
  for (int i = 0; i < n; ++i) {
    const rssf_conv3x3_item& it = items[i];
    const BnBwdStats bn = {it.bn_raw, it.bn_res, it.bn_ss, it.bn_sums, it.bn_act};
    const PreAct pre = {it.pre_stats, it.pre_gamma, it.pre_beta, it.pre_running_mean, it.pre_running_var, it.pre_mean_invstd, it.pre_ss,
                        (float)it.pre_n, it.pre_momentum, it.pre_eps, it.pre_training, it.pre_act};
    const int rc = conv_gather_impl(it.in, it.wpk, it.out, nullptr, it.stats, it.addend, nullptr, it.bn_sums ? &bn : nullptr,
                                    it.pre_ss ? &pre : nullptr, it.B, it.H, it.W, it.Cin, it.H, it.W, it.Cout, 1, 1, 9, dy, dx, dtype, stream);
    if (rc) return rc;
  }
  return RSSF_OK;
}
