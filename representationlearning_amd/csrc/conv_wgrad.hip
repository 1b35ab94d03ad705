// Weight gradient of the channels-last implicit-GEMM convolution (MFMA, split-K over pixels).
//
//   dW[tap][co][ci] = sum_p dout[p][co] * in[src(p, tap)][ci]          (autograd of conv_fwd.hip)
//
// GEMM view per tap: M = Cout, N = Cin, K = B*OH*OW pixels.  Both operands live pixel-major in HBM (channels
// contiguous) while the MFMA wants the contraction axis (pixels) contiguous per lane.  gfx950's LDS transpose read
// (ds_read_b64_tr_b16) does that for free: slabs of 64 pixels are staged ROW-MAJOR ([pixel][channel], plain 16-byte
// copies, coalesced) and each lane's 8 K-values come from two transpose reads.  Semantics measured on MI355X
// (tests/test_gpu_conv.py::test_lds_transpose_read): within a 16-lane group, lane i receives element (i % 4) of the
// 8-byte chunks addressed by lanes 4j + i/4, j = 0..3; so when lanes 4j..4j+3 address the four quarters of row j of a
// [4][16] sub-tile, lane i gets column i of it.  fp32 (parity mode) needs no transpose: the 16x16x4 f32 MFMA takes one
// K value per lane, a plain ds_read_b32 of the row-major slab.
//
// A block owns one (co, ci) tile and a pixel range and keeps the accumulators of ALL taps of a tap group in registers
// (<= 9), so the dout slab is staged once per 64 pixels, the shifted input slab once per tap, with the next tap's
// global loads in flight under the current tap's MFMAs.  Split-K partials go to a caller-provided fp32 workspace and a
// second tiny kernel adds their sum into the torch-layout gradient [Cout][Cin][kh][kw] (the fused 19-tap MLP conv
// scatters to its three source convs); measured: per-element atomics from ~1000 blocks cost 300-470 us per call,
// ~10x the GEMM itself.  (Without a workspace the kernel falls back to atomics.)
// Optional bias gradient (sum_p dout) for convolutions without a following BatchNorm.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace {

struct WgradArgs {
  const void* dout;      // [B, OH, OW, Cout]
  const void* in;        // [B, IH, IW, Cin]
  float* dw[3];          // torch-layout fp32 grads of up to 3 source convs
  int ks[3];
  float* dbias;          // [Cout] or null
  float* partial;        // [ksplit][ntaps][Cout][Cin] fp32 split-K partials (two-stage reduction) or null (atomics)
  int ntaps_total;
  int src_of_tap[MAX_TAPS];
  int kpos_of_tap[MAX_TAPS];
  int alias[MAX_TAPS][4];      // further {src, kpos} x 2 that receive the tap's gradient (-1: none), see rssf_conv_pack
  int B, IH, IW, Cin, OH, OW, Cout, stride, ksplit, ctiles_m, ctiles_n, tap0, ntap, inner, xcd_per;
  Taps taps;
};

constexpr int KP = 64;   // pixels per staged slab

typedef __attribute__((ext_vector_type(4))) short v4s;

// A/B fragment of the K-step starting at pixel row `k0` for the 16 channels starting at column `c0` of a row-major
// slab [KP][ld]; bf16: two transpose reads (rows k0+8g..+3 and +4..+7); f32: one element, row k0+g.
template <typename T> struct SlabFrag;
template <> struct SlabFrag<bf16_t> {
  static __device__ __forceinline__ bf16x8 load(const bf16_t* slab, int ld, int k0, int c0, int lane) {
    const int grp = lane >> 4, i = lane & 15;
    // K-slot -> pixel-row map of a 32-row K-step: lane group g takes rows 4g..4g+3 (first read) and 16+4g..16+4g+3
    // (second read).  Any bijection works as long as both operands use it; this one makes the 32 lanes an LDS cycle
    // services touch rows 0..7 (resp. 16..23): eight consecutive 32-byte bank groups with the +32 B row pad.  (With rows
    // 8g..8g+3 the two lane groups of a cycle sit 8 rows = 0 mod 256 B apart: a 2-way conflict on every read.)
    const bf16_t* p = slab + (k0 + grp * 4 + (i >> 2)) * ld + c0 + (i & 3) * 4;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 16 * ld));
    union { struct { v4s a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
  }
};
template <> struct SlabFrag<float> {
  static __device__ __forceinline__ float load(const float* slab, int ld, int k0, int c0, int lane) {
    return slab[(k0 + (lane >> 4)) * ld + c0 + (lane & 15)];
  }
};

// VOK: Cout and Cin are multiples of the 16-byte vector -> branch-free staging (see conv_fwd.hip): loads are issued
// unconditionally from a clamped address and invalid rows are zeroed when they are written to LDS.
template <typename T, int TMN, int TG, bool VOK>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradArgs a) {
  using MK = MmaK<T>;
  constexpr int V = Vec<T>::N;
  // +32 B per bf16 row: the four rows a 16-lane group touches in one transpose read land in disjoint 32-byte bank
  // groups (with +16 B they overlap pairwise: measured 33 % of LDS cycles lost to conflicts)
  constexpr int LD = TMN + (sizeof(T) == 2 ? 16 : LdsPad<T>::X);
  constexpr int CPR = TMN / V;                           // 16-byte chunks per slab row
  constexpr int CHUNKS = (KP * CPR + 255) / 256;         // per thread per operand
  constexpr int WI = TMN / 32;                           // 16x16 tiles per wave along each axis (wave grid 2x2)
  __shared__ __attribute__((aligned(16))) T DS[KP * LD];  // dout slab [pixel][co]
  __shared__ __attribute__((aligned(16))) T XS[KP * LD];  // in   slab [pixel][ci]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  // logical order: the (tile, tap) blocks of one pixel range are consecutive on one XCD (they share the dout slab and,
  // shifted, the input rows), successive ranges of an XCD are adjacent in memory
  const int64_t q = xcd_logical(blockIdx.x, a.xcd_per);
  if (q >= (int64_t)a.inner * a.ksplit) return;
  const unsigned q32 = (unsigned)q;                         // bounded by the grid size: 32-bit divisions
  const int range = (int)(q32 / (unsigned)a.inner);
  int bid = (int)(q32 - (unsigned)range * (unsigned)a.inner);
  const int cit = bid % a.ctiles_n; bid /= a.ctiles_n;
  const int cot = bid % a.ctiles_m; bid /= a.ctiles_m;
  const int tbase = a.tap0 + (TG == 1 ? bid : 0);        // TG == 1: the taps of the group are spread over blockIdx.x
  const int co0 = cot * TMN, ci0 = cit * TMN;
  const int64_t M = (int64_t)a.B * a.OH * a.OW;
  const int64_t per = ((M + a.ksplit - 1) / a.ksplit + KP - 1) / KP * KP;
  const int64_t kbeg = (int64_t)range * per, kend = kbeg + per < M ? kbeg + per : M;
  const T* DO = reinterpret_cast<const T*>(a.dout);
  const T* IN = reinterpret_cast<const T*>(a.in);
  const int wm = wave >> 1, wn = wave & 1;
  // bf16 vector path: hardware-bounds-checked buffer loads (32-bit byte offsets; an invalid row gets the out-of-range sentinel
  // and the load returns zeros: no zeroing at the LDS store), and for the one-tap-per-block tiles NO branch around the loads
  // of the next slab, so that the compiler counts the outstanding loads exactly (see conv_fwd.hip)
  constexpr bool FASTW = VOK && sizeof(T) == 2;
  constexpr unsigned OOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t rdo, rin;
  if constexpr (FASTW) {
    rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(DO), 0, (int)((int64_t)a.B * a.OH * a.OW * a.Cout * sizeof(T)), 0x00020000);
    rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(IN), 0, (int)((int64_t)a.B * a.IH * a.IW * a.Cin * sizeof(T)), 0x00020000);
  }

  f32x4 acc[TG][WI][WI];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < WI; ++j) acc[t][i][j] = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = a.dbias && tbase == 0 && cit == 0;

  // per-thread staging slots: chunk c -> (pixel row, 16-byte column)
  int srow[CHUNKS], scol[CHUNKS];
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) { const int id = tid + c * 256; srow[c] = id / CPR; scol[c] = (id % CPR) * V; }

  Vec<T> rx[CHUNKS], rd[CHUNKS];
  int pb[CHUNKS], py[CHUNKS], px[CHUNKS];
  bool pv[CHUNKS], dok[CHUNKS], xok[CHUNKS];

  // Pixel coordinates of this thread's slab rows are carried from slab to slab (+KP pixels with carries) instead of being
  // re-derived by integer division: the index arithmetic was ~11 VALU instructions per MFMA (a third of the wave's cycles).
  // Offsets are 32-bit (the host entry rejects tensors of >= 2^31 elements).
  int pix[CHUNKS];
  auto first_slab = [&](int64_t k0) {
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      pix[c] = (int)k0 + srow[c];
      pb[c] = pix[c] / (a.OH * a.OW);
      const int rem = pix[c] % (a.OH * a.OW);
      py[c] = rem / a.OW; px[c] = rem % a.OW;
    }
  };
  auto next_slab = [&]() {
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      pix[c] += KP; px[c] += KP;
      while (px[c] >= a.OW) { px[c] -= a.OW; ++py[c]; }
      while (py[c] >= a.OH) { py[c] -= a.OH; ++pb[c]; }
    }
  };
  auto load_d = [&]() {               // dout rows of the current slab
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      pv[c] = srow[c] < KP && pix[c] < (int)kend;
      const int ch = co0 + scol[c];
      dok[c] = pv[c] && ch < a.Cout;
      const int off = dok[c] ? pix[c] * a.Cout + ch : 0;
      if constexpr (FASTW) {
        rd[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, dok[c] ? (unsigned)off * 2u : OOB, 0, 0));
      } else if constexpr (VOK) {
        rd[c].load(DO + off);
      } else {
        rd[c].raw = {0, 0, 0, 0};
        if (dok[c])
          for (int e = 0; e < V; ++e) if (ch + e < a.Cout) rd[c].set(e, ldf(DO + off + e));
      }
    }
  };
  auto load_x = [&](int t) {      // gather the input slab of tap t into registers
    const int dy = a.taps.dy[tbase + t], dx = a.taps.dx[tbase + t];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const int sy = py[c] * a.stride + dy, sx = px[c] * a.stride + dx;
      const int ch = ci0 + scol[c];
      xok[c] = pv[c] && sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW && ch < a.Cin;
      const int off = xok[c] ? ((pb[c] * a.IH + sy) * a.IW + sx) * a.Cin + ch : 0;
      if constexpr (FASTW) {
        rx[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, xok[c] ? (unsigned)off * 2u : OOB, 0, 0));
      } else if constexpr (VOK) {
        rx[c].load(IN + off);
      } else {
        rx[c].raw = {0, 0, 0, 0};
        if (xok[c])
          for (int e = 0; e < V; ++e) if (ch + e < a.Cin) rx[c].set(e, ldf(IN + off + e));
      }
    }
  };

  // Software pipeline over (slab, tap) steps: the operands of the NEXT step (next tap of this slab, or the dout rows and
  // first tap of the next slab) are requested right after the barrier that publishes the current step, so the global
  // latency runs under the MFMAs instead of in front of them.
  if (kbeg < kend) { first_slab(kbeg); load_d(); load_x(0); }
  for (int64_t k0 = kbeg; k0 < kend; k0 += KP) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (TG == 1 || t < a.ntap) {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
          if (srow[c] < KP) {
            if (t == 0) {
              Vec<T> v = rd[c];
              if (VOK && !FASTW && !dok[c]) v.raw = {0, 0, 0, 0};
              v.store(DS + srow[c] * LD + scol[c]);
            }
            Vec<T> v = rx[c];
            if (VOK && !FASTW && !xok[c]) v.raw = {0, 0, 0, 0};
            v.store(XS + srow[c] * LD + scol[c]);
          }
        __syncthreads();
        if (t + 1 < TG && t + 1 < a.ntap) load_x(t + 1);
        else if ((FASTW && TG == 1) || k0 + KP < kend) { next_slab(); load_d(); load_x(0); }    // past the end: pv false -> zeros, unused
#pragma unroll
        for (int ks = 0; ks < KP; ks += MK::KSTEP) {
          typename MK::frag fa[WI], fb[WI];
#pragma unroll
          for (int i = 0; i < WI; ++i) {
            fa[i] = SlabFrag<T>::load(DS, LD, ks, (wm * WI + i) * 16, lane);
            fb[i] = SlabFrag<T>::load(XS, LD, ks, (wn * WI + i) * 16, lane);
          }
#pragma unroll
          for (int i = 0; i < WI; ++i)
#pragma unroll
            for (int j = 0; j < WI; ++j) acc[t][i][j] = MK::mma(fa[i], fb[j], acc[t][i][j]);
        }
        if (t == 0 && do_bias && tid < TMN) {
          float s = 0.f;
          for (int k = 0; k < KP; ++k) s += ldf(DS + k * LD + tid);
          bsum += s;
        }
        __syncthreads();
      }
    }
  }

  // ---- flush: rows = co (4*grp + r), cols = ci (l15) -----------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    if (TG > 1 && t >= a.ntap) continue;
    const int tap = tbase + t;
    const int s = a.src_of_tap[tap], kk = a.ks[s] * a.ks[s], kpos = a.kpos_of_tap[tap];
    float* dw = a.dw[s];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < WI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + (wm * WI + i) * 16 + grp * 4 + r, ci = ci0 + (wn * WI + j) * 16 + l15;
          if (co >= a.Cout || ci >= a.Cin) continue;
          if (a.partial) a.partial[(((int64_t)range * a.ntaps_total + tap) * a.Cout + co) * a.Cin + ci] = acc[t][i][j][r];
          else {
            atomicAdd(dw + ((int64_t)co * a.Cin + ci) * kk + kpos, acc[t][i][j][r]);
            for (int e = 0; e < 4; e += 2) {
              const int s2 = a.alias[tap][e];
              if (s2 >= 0) atomicAdd(a.dw[s2] + ((int64_t)co * a.Cin + ci) * (a.ks[s2] * a.ks[s2]) + a.alias[tap][e + 1], acc[t][i][j][r]);
            }
          }
        }
  }
  if (do_bias && tid < TMN && co0 + tid < a.Cout) atomicAdd(a.dbias + co0 + tid, bsum);
}

// The 128 x 128 (co, ci) tile of the wide layers (MlpDWBN's 17-tap sum: 8 launches of 146 GFLOP per training step) with EIGHT
// waves and a two-slab pipeline.  The four-wave kernel above keeps one slab in flight per block (register staging of the next
// slab under the MFMAs of the current one, two barriers per slab) and two blocks per CU: with ~2 us of memory latency under load
// and ~0.6 us of work per slab it waited out a round trip per slab (measured: 298 us per launch = 19 % of the MFMA peak, the
// forward kernel of the same FLOPs runs 191 us).  Here a wave owns a 64 x 32 tile (32 accumulator registers instead of 64) and 2
// staging chunks per operand (instead of 4), which pays for TWO register sets and two LDS buffers: slabs s+1 and s+2 are in
// flight while slab s computes, one barrier per slab, no branch around a load (exact vmcnt bookkeeping, see conv_fwd.hip).
__global__ void __launch_bounds__(512) conv_wgrad8_kernel(WgradArgs a) {
  using T = bf16_t;
  using MK = MmaK<T>;
  constexpr int TMN = 128, V = 8, LD = TMN + 16, CPR = TMN / V, CHUNKS = KP * CPR / 512;       // 2 chunks per thread and operand
  static_assert(CHUNKS * 512 == KP * CPR, "staging covers the slab exactly");
  __shared__ __attribute__((aligned(16))) T DS[2][KP * LD];
  __shared__ __attribute__((aligned(16))) T XS[2][KP * LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int64_t q = xcd_logical(blockIdx.x, a.xcd_per);
  if (q >= (int64_t)a.inner * a.ksplit) return;
  const unsigned q32 = (unsigned)q;
  const int range = (int)(q32 / (unsigned)a.inner);
  int bid = (int)(q32 - (unsigned)range * (unsigned)a.inner);
  const int cit = bid % a.ctiles_n; bid /= a.ctiles_n;
  const int cot = bid % a.ctiles_m; bid /= a.ctiles_m;
  const int tap = a.tap0 + bid;
  const int co0 = cot * TMN, ci0 = cit * TMN;
  const int64_t M = (int64_t)a.B * a.OH * a.OW;
  const int64_t per = ((M + a.ksplit - 1) / a.ksplit + KP - 1) / KP * KP;
  const int64_t kbeg = (int64_t)range * per, kend = kbeg + per < M ? kbeg + per : M;
  const int wm = wave >> 2, wn = wave & 3;                   // 2 x 4 waves of 64 (co) x 32 (ci)
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dout), 0, (int)((int64_t)a.B * a.OH * a.OW * a.Cout * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)((int64_t)a.B * a.IH * a.IW * a.Cin * 2), 0x00020000);
  const int dy = a.taps.dy[tap], dx = a.taps.dx[tap];

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  int srow[CHUNKS], scol[CHUNKS], pix[CHUNKS], pb[CHUNKS], py[CHUNKS], px[CHUNKS];
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int id = tid + c * 512;
    srow[c] = id / CPR; scol[c] = (id % CPR) * V;
    pix[c] = (int)kbeg + srow[c];
    pb[c] = pix[c] / (a.OH * a.OW);
    const int rem = pix[c] % (a.OH * a.OW);
    py[c] = rem / a.OW; px[c] = rem % a.OW;
  }
  struct Regs { u32x4 d[CHUNKS], x[CHUNKS]; };
  Regs R0, R1;
  auto load = [&](Regs& R) {         // the slab the coordinate state points at, then advance the state by one slab
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const bool pv = pix[c] < (int)kend;                                                  // past the range: zeros, never used
      const int cho = co0 + scol[c], chi = ci0 + scol[c];
      const int sy = py[c] * a.stride + dy, sx = px[c] * a.stride + dx;
      const bool dok = pv && cho < a.Cout;
      const bool xok = pv && sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW && chi < a.Cin;
      const unsigned doff = (unsigned)(pix[c] * a.Cout + cho) * 2u;
      const unsigned xoff = (unsigned)(((pb[c] * a.IH + sy) * a.IW + sx) * a.Cin + chi) * 2u;
      R.d[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, dok ? doff : OOB, 0, 0));
      R.x[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, xok ? xoff : OOB, 0, 0));
      pix[c] += KP; px[c] += KP;
      while (px[c] >= a.OW) { px[c] -= a.OW; ++py[c]; }
      while (py[c] >= a.OH) { py[c] -= a.OH; ++pb[c]; }
    }
  };
  auto store = [&](const Regs& R, int b) {
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      *reinterpret_cast<u32x4*>(&DS[b][srow[c] * LD + scol[c]]) = R.d[c];
      *reinterpret_cast<u32x4*>(&XS[b][srow[c] * LD + scol[c]]) = R.x[c];
    }
  };
  auto compute = [&](int b) {
#pragma unroll
    for (int ks = 0; ks < KP; ks += MK::KSTEP) {
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = SlabFrag<T>::load(DS[b], LD, ks, (wm * 4 + i) * 16, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = SlabFrag<T>::load(XS[b], LD, ks, (wn * 2 + j) * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MK::mma(fa[i], fb[j], acc[i][j]);
    }
  };
  // bias gradient (MlpDWBN's convolutions carry a bias, ffn_block.py:219-228): column sums of
  // the dout slab, by the blocks of the first tap / first ci tile only (block-uniform; LDS reads, no global load in the branch)
  const bool do_bias = a.dbias && tap == a.tap0 && cit == 0;
  float bsum = 0.f;
  auto bias_rows = [&](int b) {
    const int col = tid & 127, r0 = (tid >> 7) * (KP / 4);
#pragma unroll
    for (int k = 0; k < KP / 4; ++k) bsum += ldf(&DS[b][(r0 + k) * LD + col]);
  };
  const int nslabs = (int)((kend - kbeg + KP - 1) / KP), nrun = (nslabs + 1) & ~1;
  load(R0); load(R1);
  for (int s = 0; s < nrun; s += 2) {
    store(R0, 0);
    __syncthreads();               // buffer 0 complete; every wave is done computing out of buffer 1's previous contents
    load(R0);
    compute(0);
    if (do_bias) bias_rows(0);
    store(R1, 1);
    __syncthreads();
    load(R1);
    compute(1);
    if (do_bias) bias_rows(1);
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(&XS[0][0]);          // [4][128]
    red[tid] = bsum;
    __syncthreads();
    if (tid < 128 && co0 + tid < a.Cout) atomicAdd(a.dbias + co0 + tid, (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]));
  }
  if (!a.partial) return;          // (the dispatcher sends only workspace launches here)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wm * 4 + i) * 16 + grp * 4 + r, ci = ci0 + (wn * 2 + j) * 16 + l15;
        if (co < a.Cout && ci < a.Cin) a.partial[(((int64_t)range * a.ntaps_total + tap) * a.Cout + co) * a.Cin + ci] = acc[i][j][r];
      }
}

// TWO taps per block (round 4).  With one tap per block every tap streams all of dout and of the input through its blocks: 17 x
// 134 MB = 2.3 GB per launch of the MLP sum, L2 -> CU at 10 TB/s - that stream, not the transpose reads, is what the 223 us are
// (the lattice kernel of conv_lattice.hip hit the same ceiling at 8 TB/s with its weight fragments).  Here the dout slab of a block
// serves two taps (1.7 GB per launch), a wave keeps both taps' 64 x 32 accumulators (64 registers), and a K-step is 16 transpose
// reads for 16 MFMAs instead of 12 for 8.  LDS: 2 x (1 + 2) slabs = 110 KB, one block of eight waves per CU; the grid is one round
// of the chip (ksplit = CUs / tap groups).  An odd last tap runs with a second tap whose loads are out of range (zeros).
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_wgrad8x2_kernel(WgradArgs a) {
  using T = bf16_t;
  using MK = MmaK<T>;
  constexpr int TMN = 128, V = 8, LD = TMN + 16, CPR = TMN / V, CHUNKS = KP * CPR / 512, SLAB = KP * LD;
  extern __shared__ __attribute__((aligned(16))) char smem_w[];
  T* const S = reinterpret_cast<T*>(smem_w);               // [2 buffers][dout, x tap 0, x tap 1][KP * LD]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int64_t q = xcd_logical(blockIdx.x, a.xcd_per);
  if (q >= (int64_t)a.inner * a.ksplit) return;
  const unsigned q32 = (unsigned)q;
  const int range = (int)(q32 / (unsigned)a.inner);
  int bid = (int)(q32 - (unsigned)range * (unsigned)a.inner);
  const int cit = bid % a.ctiles_n; bid /= a.ctiles_n;
  const int cot = bid % a.ctiles_m; bid /= a.ctiles_m;
  const int tapA = a.tap0 + 2 * bid, tapB = tapA + 1;
  const bool hasB = tapB < a.tap0 + a.ntap;
  const int co0 = cot * TMN, ci0 = cit * TMN;
  const int64_t M = (int64_t)a.B * a.OH * a.OW;
  const int64_t per = ((M + a.ksplit - 1) / a.ksplit + KP - 1) / KP * KP;
  const int64_t kbeg = (int64_t)range * per, kend = kbeg + per < M ? kbeg + per : M;
  const int wm = wave >> 2, wn = wave & 3;                   // 2 x 4 waves of 64 (co) x 32 (ci)
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dout), 0, (int)((int64_t)a.B * a.OH * a.OW * a.Cout * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)((int64_t)a.B * a.IH * a.IW * a.Cin * 2), 0x00020000);
  const int dyA = a.taps.dy[tapA], dxA = a.taps.dx[tapA];
  const int dyB = a.taps.dy[hasB ? tapB : tapA], dxB = a.taps.dx[hasB ? tapB : tapA];

  f32x4 acc[2][4][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = {0.f, 0.f, 0.f, 0.f};

  int srow[CHUNKS], scol[CHUNKS], pix[CHUNKS], pb[CHUNKS], py[CHUNKS], px[CHUNKS];
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int id = tid + c * 512;
    srow[c] = id / CPR; scol[c] = (id % CPR) * V;
    pix[c] = (int)kbeg + srow[c];
    pb[c] = pix[c] / (a.OH * a.OW);
    const int rem = pix[c] % (a.OH * a.OW);
    py[c] = rem / a.OW; px[c] = rem % a.OW;
  }
  struct Regs { u32x4 d[CHUNKS], xa[CHUNKS], xb[CHUNKS]; };
  Regs R0, R1;
  auto load = [&](Regs& R) {         // the slab the coordinate state points at, then advance the state by one slab
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const bool pv = pix[c] < (int)kend;                                                  // past the range: zeros, never used
      const int cho = co0 + scol[c], chi = ci0 + scol[c];
      const int yy = py[c] * a.stride, xx = px[c] * a.stride;
      const int syA = yy + dyA, sxA = xx + dxA, syB = yy + dyB, sxB = xx + dxB;
      const bool dok = pv && cho < a.Cout;
      const bool xokA = pv && syA >= 0 && syA < a.IH && sxA >= 0 && sxA < a.IW && chi < a.Cin;
      const bool xokB = pv && hasB && syB >= 0 && syB < a.IH && sxB >= 0 && sxB < a.IW && chi < a.Cin;
      const unsigned doff = (unsigned)(pix[c] * a.Cout + cho) * 2u;
      const unsigned xoffA = (unsigned)(((pb[c] * a.IH + syA) * a.IW + sxA) * a.Cin + chi) * 2u;
      const unsigned xoffB = (unsigned)(((pb[c] * a.IH + syB) * a.IW + sxB) * a.Cin + chi) * 2u;
      R.d[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, dok ? doff : OOB, 0, 0));
      R.xa[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, xokA ? xoffA : OOB, 0, 0));
      R.xb[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, xokB ? xoffB : OOB, 0, 0));
      pix[c] += KP; px[c] += KP;
      while (px[c] >= a.OW) { px[c] -= a.OW; ++py[c]; }
      while (py[c] >= a.OH) { py[c] -= a.OH; ++pb[c]; }
    }
  };
  auto store = [&](const Regs& R, int b) {
    T* base = S + b * 3 * SLAB;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      *reinterpret_cast<u32x4*>(base + srow[c] * LD + scol[c]) = R.d[c];
      *reinterpret_cast<u32x4*>(base + SLAB + srow[c] * LD + scol[c]) = R.xa[c];
      *reinterpret_cast<u32x4*>(base + 2 * SLAB + srow[c] * LD + scol[c]) = R.xb[c];
    }
  };
  auto compute = [&](int b) {
    const T* base = S + b * 3 * SLAB;
#pragma unroll
    for (int ks = 0; ks < KP; ks += MK::KSTEP) {
      bf16x8 fa[4], fb[2][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = SlabFrag<T>::load(base, LD, ks, (wm * 4 + i) * 16, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[t][j] = SlabFrag<T>::load(base + (1 + t) * SLAB, LD, ks, (wn * 2 + j) * 16, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[t][i][j] = MK::mma(fa[i], fb[t][j], acc[t][i][j]);
    }
  };
  const bool do_bias = a.dbias && tapA == a.tap0 && cit == 0;
  float bsum = 0.f;
  auto bias_rows = [&](int b) {
    const T* base = S + b * 3 * SLAB;
    const int col = tid & 127, r0 = (tid >> 7) * (KP / 4);
#pragma unroll
    for (int k = 0; k < KP / 4; ++k) bsum += ldf(base + (r0 + k) * LD + col);
  };
  const int nslabs = (int)((kend - kbeg + KP - 1) / KP), nrun = (nslabs + 1) & ~1;
  load(R0); load(R1);
  for (int s = 0; s < nrun; s += 2) {
    store(R0, 0);
    __syncthreads();               // buffer 0 complete; every wave is done computing out of buffer 1's previous contents
    load(R0);
    compute(0);
    if (do_bias) bias_rows(0);
    store(R1, 1);
    __syncthreads();
    load(R1);
    compute(1);
    if (do_bias) bias_rows(1);
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(S);                  // [4][128]
    red[tid] = bsum;
    __syncthreads();
    if (tid < 128 && co0 + tid < a.Cout) atomicAdd(a.dbias + co0 + tid, (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]));
  }
  if (!a.partial) return;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t == 1 && !hasB) break;
    const int tap = tapA + t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + (wm * 4 + i) * 16 + grp * 4 + r, ci = ci0 + (wn * 2 + j) * 16 + l15;
          if (co < a.Cout && ci < a.Cin) a.partial[(((int64_t)range * a.ntaps_total + tap) * a.Cout + co) * a.Cin + ci] = acc[t][i][j][r];
        }
  }
}
constexpr size_t WG8X2_LDS = (size_t)2 * 3 * KP * (128 + 16) * 2;

// second stage of the split-K reduction: dw[torch layout] += sum_k partial[k][tap][co][ci].  A block owns RI consecutive
// gradient elements (rows of the partial planes); its 8 thread groups walk interleaved k planes, fold through LDS, and
// one thread per element does the (non-atomic) read-modify-write.
#ifndef RSSF_REDUCE_COLS
#define RSSF_REDUCE_COLS 256
#endif
constexpr int RI = RSSF_REDUCE_COLS, RG = 8, RT = 256 / RG, RV = RI / (4 * RT);      // columns per block, plane groups, threads per group, 16-byte vectors per thread
static_assert(RV >= 1 && RV * 4 * RT == RI && RI % 256 == 0 || RI == 128, "column tiling");
// second stage of the split-K weight gradient: dw += sum over the ksplit planes of `partial`, for one job (= one convolution).
// `blk` = block index within the job (RI columns of the [ntaps*Cout*Cin] plane each).  A thread owns RV x 4 columns and
// every RG-th plane, 4 RV 16-byte loads in flight (the pass reads 2.5 GB per training step: with 4-byte loads it ran at 2.9 TB/s,
// with 128 columns per block - 248 000 blocks of ~10 KB each - at 4.1 TB/s).
__device__ __forceinline__ void wgrad_reduce_body(const rssf_wgrad_reduce_job& a, int blk) {
  __shared__ float red[RG][RI];
  const int64_t per = (int64_t)a.ntaps * a.cout * a.cin;
  const int it = threadIdx.x % RT, kg = threadIdx.x / RT;
  f32x4 tot[RV];
#pragma unroll
  for (int v = 0; v < RV; ++v) {
    const int64_t i = (int64_t)blk * RI + (v * RT + it) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if ((per & 3) == 0) {                                      // 16-byte aligned rows (every layer of the path)
      if (i < per) {
        const float* p = a.partial + i;
        auto ld = [&](int k) { return *reinterpret_cast<const f32x4*>(p + (int64_t)k * per); };
        int k = kg;
        for (; k + 3 * RG < a.ksplit; k += 4 * RG) {
          const f32x4 v0 = ld(k), v1 = ld(k + RG), v2 = ld(k + 2 * RG), v3 = ld(k + 3 * RG);
          s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; k < a.ksplit; k += RG) s0 += ld(k);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i + e < per)
          for (int k = kg; k < a.ksplit; k += RG) s0[e] += a.partial[(int64_t)k * per + i + e];
    }
    tot[v] = (s0 + s1) + (s2 + s3);
  }
#pragma unroll
  for (int v = 0; v < RV; ++v) *reinterpret_cast<f32x4*>(&red[kg][(v * RT + it) * 4]) = tot[v];
  __syncthreads();
  for (int c = threadIdx.x; c < RI; c += 256) {               // one thread per column does the (non-atomic) read-modify-write
    const int64_t j = (int64_t)blk * RI + c;
    if (j >= per) break;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < RG; ++g) s += red[g][c];
    const int ci = (int)(j % a.cin), co = (int)((j / a.cin) % a.cout), tap = (int)(j / ((int64_t)a.cin * a.cout));
    const int sc = a.src_of_tap[tap], kk = a.ks[sc] * a.ks[sc];
    a.dw[sc][((int64_t)co * a.cin + ci) * kk + a.kpos_of_tap[tap]] += s;
    for (int e = 0; e < 4; e += 2) {
      const int s2 = a.alias_of_tap[tap][e];
      if (s2 >= 0) a.dw[s2][((int64_t)co * a.cin + ci) * (a.ks[s2] * a.ks[s2]) + a.alias_of_tap[tap][e + 1]] += s;
    }
  }
}
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(rssf_wgrad_reduce_job a) { wgrad_reduce_body(a, blockIdx.x); }
// every deferred reduction of a training step in ONE launch: block b serves job block_map[2b], block block_map[2b+1] of it
// (331 launches of ~5.6 us - each a latency-bound walk over ~9 MB of partials - become one bandwidth-bound pass)
__global__ void __launch_bounds__(256) wgrad_reduce_batch_kernel(const rssf_wgrad_reduce_job* __restrict__ jobs, const int* __restrict__ block_map) {
  wgrad_reduce_body(jobs[block_map[2 * blockIdx.x]], block_map[2 * blockIdx.x + 1]);
}

rssf_wgrad_reduce_job make_job(const WgradArgs& a) {
  rssf_wgrad_reduce_job j;
  memset(&j, 0, sizeof(j));            // padding bytes too: callers compare job descriptions bytewise
  j.partial = a.partial;
  for (int i = 0; i < 3; ++i) { j.dw[i] = a.dw[i]; j.ks[i] = a.ks[i]; }
  j.ntaps = a.ntaps_total; j.cout = a.Cout; j.cin = a.Cin; j.ksplit = a.ksplit;
  for (int t = 0; t < MAX_TAPS; ++t) {
    const bool live = t < a.ntaps_total;
    j.src_of_tap[t] = live ? a.src_of_tap[t] : 0; j.kpos_of_tap[t] = live ? a.kpos_of_tap[t] : 0;
    for (int e = 0; e < 4; ++e) j.alias_of_tap[t][e] = live ? a.alias[t][e] : -1;
  }
  return j;
}
// run the second stage now, or hand its description to the caller (who batches it: rssf_conv_wgrad_reduce_batch)
int finish_reduce(const WgradArgs& a, rssf_wgrad_reduce_job* defer, hipStream_t st) {
  const rssf_wgrad_reduce_job j = make_job(a);
  if (defer) { *defer = j; return RSSF_OK; }
  const int64_t per = (int64_t)j.ntaps * j.cout * j.cin;
  wgrad_reduce_kernel<<<(unsigned)((per + RI - 1) / RI), 256, 0, st>>>(j);
  return check_launch("conv_wgrad_reduce");
}

}  // namespace
namespace rssf { namespace cv {
// the second stage for a first stage that lives in another translation unit (conv_wgrad_planes.hip)
int launch_wgrad_reduce(const rssf_wgrad_reduce_job& j, hipStream_t st) {
  const int64_t per = (int64_t)j.ntaps * j.cout * j.cin;
  wgrad_reduce_kernel<<<(unsigned)((per + RI - 1) / RI), 256, 0, st>>>(j);
  return check_launch("conv_wgrad_reduce");
}
} }
namespace {
// ---- halo-tiled weight gradient of the 3x3 / stride-1 / "same" bf16 convolutions (HRNet BasicBlock / Bottleneck) -------
// A block owns one 32 x 32 (co, ci) tile and a run of 8 x 16-pixel spatial tiles.  Per spatial tile it stages the dout tile
// [128 px][32 co] and the input HALO [10 x 18 px][32 ci] once (pixel-major, plain 16-byte copies, next tile's global loads
// in flight under the MFMAs) and all nine taps contract out of LDS: the tap only shifts the row index of the input
// operand's transpose reads.  Each wave owns one 16 x 16 sub-tile of every tap (9 accumulators, no cross-wave reduction).
// Versus the generic kernel (a block per tap, the dout slab and a shifted input slab re-staged per 64 pixels): 4x less
// L2 -> LDS traffic, two barriers per 36 MFMAs per wave instead of per 2, and the split-K factor (= partial planes the
// second stage has to fold) drops to ntiles / 8.
struct WgradHaloArgs {
  const bf16_t* dout;    // [B, H, W, Cout]
  const bf16_t* in;      // [B, H, W, Cin]
  float* partial;        // [ksplit][9][Cout][Cin]
  int B, H, W, Cin, Cout, tiles_y, tiles_x, ntiles, tiles_per_block, npairs, ptiles_n, xcd_per;
  int64_t total;
  int dy[9], dx[9];
  // Fused BatchNorm-backward APPLY (bn_dy != null; `dout` is then not read): the output-gradient tile of the convolution is
  // COMPUTED while it is staged, draw = scale * (dz - k1 - xhat * k2), dz = bn_dy * act'(bn_raw * scale + shift + bn_res) - the
  // arithmetic of bn_bwd_apply_kernel (bn.hip), same operation order - and the blocks of input-channel tile 0 also write it to
  // `draw_out` (the data-gradient launch reads it) and dz to `dres_out` (optional); the blocks of pixel range 0 add the
  // parameter gradients.  One tensor pass (read dy, raw; write draw) and a launch less per layer.
  const bf16_t* bn_dy; const bf16_t* bn_raw; const bf16_t* bn_res;
  const float* bn_ss; const float* bn_mi; const float* bn_sums;       // [2][Cout] x 2, [RSSF_BN_BWD_SLOTS][2][Cout]
  bf16_t* draw_out; bf16_t* dres_out;
  float* dgamma; float* dbeta;
  float bn_n, bn_pscale;
  int bn_act, bn_training;
  // PRE-activation input operand (x_ss != null): `in` is the RAW output of the producing convolution, the operand contracted is
  // act(in * scale + shift) formed while the halo tile is staged (see HaloArgs::pre_ss, conv.hip.h)
  const float* x_ss;     // [2][Cin]
  int x_act;
};
constexpr int HTH = 8, HTW = 16, HHP = (HTH + 2) * (HTW + 2), HNPX = HTH * HTW, HCT = 32, HLD = HCT + 16;

// 8 waves: two groups of four split the four 32-pixel K-steps of every tile (twice the waves per CU on the same LDS
// footprint - the block is latency-bound, one resident block per CU); group 1's accumulators are folded into group 0's
// through LDS once, after the last tile.
constexpr int HWG_THREADS = 512;
// FUSE / RES are template parameters: a run-time branch around the staging loads would make the compiler drain vmcnt at the join,
// i.e. wait for a tile's loads where they are issued instead of one tile later (measured: 16 -> 27 us per launch)
template <bool FUSE, bool RES, bool XPRE>
__device__ __forceinline__ void wgrad_halo_block(const WgradHaloArgs& a, const int64_t q) {
  constexpr int NT_ = HWG_THREADS;
  constexpr int X_LOADS = (HHP * 4 + NT_ - 1) / NT_, D_LOADS = HNPX * 4 / NT_;
  constexpr int STAGE_ELEMS = (HHP + HNPX) * HLD;
  static_assert(STAGE_ELEMS * 2 >= 5 * 4 * 64 * 4 * 4, "fold buffer (5 taps x 4 waves x 64 lanes x f32x4) must fit the staging LDS");
  __shared__ __attribute__((aligned(16))) bf16_t lds_raw[STAGE_ELEMS];
  bf16_t* XH = lds_raw;
  bf16_t* DS = lds_raw + HHP * HLD;
  const unsigned q32 = (unsigned)q;                         // bounded by the grid size: 32-bit divisions
  const int range = (int)(q32 / (unsigned)a.npairs), pair = (int)(q32 - (unsigned)range * (unsigned)a.npairs);
  const int co0 = (pair / a.ptiles_n) * HCT, ci0 = (pair % a.ptiles_n) * HCT;
  const int t_begin = range * a.tiles_per_block;
  const int t_end = t_begin + a.tiles_per_block < a.ntiles ? t_begin + a.tiles_per_block : a.ntiles;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wg = wave >> 2, w4 = wave & 3;                  // wave group (K-steps 2wg, 2wg+1), wave inside the group
  const int wm = w4 >> 1, wn = w4 & 1;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};

  // ---- fused BatchNorm-backward apply: per-channel constants of this block's 32 output channels through LDS -----------------
  constexpr bool fuse = FUSE;
  __shared__ __attribute__((aligned(16))) float sbn[4][HCT];                               // scale, shift, cb, cc (bn_bwd_apply_kernel's names)
  static_assert(D_LOADS == 1, "a thread stages ONE 8-channel chunk of the dout tile: its constants are fixed");
  // (the constants of a thread's 8 channels are READ BACK from LDS where a tile is staged - eight 16-byte reads per tile - instead
  // of living in 32 + 16 registers: with them the fused variants needed 166..186 VGPRs = ONE resident block per CU; at <= 128 two
  // blocks share a CU, which is what lets the problems of a grouped launch overlap, see conv3x3_wgrad_halo_group_kernel)
  if (fuse) {
    if (tid < HCT) {
      const int c = co0 + tid;
      float sc = 0.f, sh = 0.f, cb = 0.f, cc = 0.f;
      if (c < a.Cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < RSSF_BN_BWD_SLOTS; ++k) { s1 += a.bn_sums[(size_t)k * 2 * a.Cout + c]; s2 += a.bn_sums[(size_t)k * 2 * a.Cout + a.Cout + c]; }
        const float mean = a.bn_mi[c], istd = a.bn_mi[a.Cout + c];
        sc = a.bn_ss[c]; sh = a.bn_ss[a.Cout + c];
        float dot;
        bn_bwd_constants(sc, mean, istd, s1, s2, a.bn_n, dot, cb, cc);
        if (a.dgamma && range == 0 && ci0 == 0) { a.dgamma[c] += dot * a.bn_pscale; a.dbeta[c] += s1 * a.bn_pscale; }      // one writer per channel
      }
      sbn[0][tid] = sc; sbn[1][tid] = sh; sbn[2][tid] = cb; sbn[3][tid] = cc;
    }
  }
  const bool writer = fuse && ci0 == 0;                       // the (co, ci = 0) blocks own the global copy of draw / dres
  // XPRE: scale / shift of the block's 32 input channels (a thread's channel group inside the tile is fixed: tid & 3)
  __shared__ __attribute__((aligned(16))) float sxs[2][HCT];
  if constexpr (XPRE) {
    if (tid >= 64 && tid < 64 + HCT) {
      const int c = ci0 + tid - 64;
      const bool cok = c < a.Cin;
      sxs[0][tid - 64] = cok ? a.x_ss[c] : 0.f;
      sxs[1][tid - 64] = cok ? a.x_ss[a.Cin + c] : 0.f;
    }
  }
  if (fuse || XPRE) __syncthreads();
  const int cg8 = (tid & 3) * 8;                              // this thread's 8 channels inside a 32-channel tile

  Vec<bf16_t> rx[X_LOADS], rd[D_LOADS], rr[D_LOADS], rq[D_LOADS], vdk[D_LOADS], vzk[D_LOADS];
  // every load is a hardware-bounds-checked buffer load with a 32-bit byte offset (the entry point keeps the tensors below 2^31
  // bytes): a halo pixel outside the image / a channel chunk past the tensor gets the out-of-range offset and loads zeros - no
  // 64-bit address pairs, no validity flags held across the tile (VGPRs are what limits this kernel to one block per CU)
  constexpr unsigned SOOB = 0x80000000u;                    // >= num_records
  unsigned xoff[X_LOADS], doff[D_LOADS], soff[D_LOADS];
  const int dbytes = (int)((int64_t)a.B * a.H * a.W * a.Cout * 2), xbytes = (int)((int64_t)a.B * a.H * a.W * a.Cin * 2);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(fuse ? a.bn_dy : a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rraw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(fuse ? a.bn_raw : a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(fuse && RES ? a.bn_res : a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdraw = __builtin_amdgcn_make_buffer_rsrc(fuse ? a.draw_out : const_cast<bf16_t*>(a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdres = __builtin_amdgcn_make_buffer_rsrc(fuse && a.dres_out ? a.dres_out : (fuse ? a.draw_out : const_cast<bf16_t*>(a.dout)), 0, dbytes, 0x00020000);
  auto load_tile = [&](int t) {
    const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
    const int y0 = ty * HTH, x0 = tx * HTW;
#pragma unroll
    for (int i = 0; i < X_LOADS; ++i) {
      const int idx = tid + i * NT_, p = idx >> 2, ch = ci0 + (idx & 3) * 8;
      const int gy = y0 - 1 + p / (HTW + 2), gx = x0 - 1 + p % (HTW + 2);
      const bool ok = idx < HHP * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ch < a.Cin;
      xoff[i] = ok ? (unsigned)((((b * a.H + gy) * a.W + gx) * a.Cin + ch) * 2) : SOOB;
      rx[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, xoff[i], 0, 0));
    }
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      const int idx = tid + i * NT_, p = idx >> 2, ch = co0 + (idx & 3) * 8;
      const int gy = y0 + p / HTW, gx = x0 + p % HTW;
      const bool ok = gy < a.H && gx < a.W && ch < a.Cout;
      doff[i] = ok ? (unsigned)((((b * a.H + gy) * a.W + gx) * a.Cout + ch) * 2) : SOOB;
      rd[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, doff[i], 0, 0));
      if constexpr (fuse) {
        rr[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rraw, doff[i], 0, 0));
        if constexpr (RES) rq[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, doff[i], 0, 0));
      }
    }
  };
  // draw / dz of one staged chunk (see WgradHaloArgs): block-uniform activation, one specialised loop runs
  auto apply_chunk = [&](auto ACT, int i, Vec<bf16_t>& vdraw, Vec<bf16_t>& vdz) {
    float o1[8], o2[8], bsc[8], bsh[8], bcb[8], bcc[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&sbn[0][cg8 + 4 * h]), v1 = *reinterpret_cast<const f32x4*>(&sbn[1][cg8 + 4 * h]);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(&sbn[2][cg8 + 4 * h]), v3 = *reinterpret_cast<const f32x4*>(&sbn[3][cg8 + 4 * h]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bsc[4 * h + e] = v0[e]; bsh[4 * h + e] = v1[e]; bcb[4 * h + e] = v2[e]; bcc[4 * h + e] = v3[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = rr[i].get(e);
      float z = fmaf(x, bsc[e], bsh[e]);
      if constexpr (RES) z += rq[i].get(e);
      const float g = rd[i].get(e);
      const float dz = decltype(ACT)::value == 1 ? g * (z > 0.f ? 1.f : 0.f) : decltype(ACT)::value == 2 ? g * gelu_erf_grad(z) : g;
      o2[e] = dz;
      o1[e] = a.bn_training ? fmaf(bsc[e], dz, fmaf(bcb[e], x, bcc[e])) : bsc[e] * dz;
    }
    vdraw.set_all(o1); vdz.set_all(o2);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < X_LOADS; ++i) {
      const int idx = tid + i * NT_;
      Vec<bf16_t> v = rx[i];
      if constexpr (XPRE) {
        float xsc[8], xsh[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(&sxs[0][cg8 + 4 * h]), v1 = *reinterpret_cast<const f32x4*>(&sxs[1][cg8 + 4 * h]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { xsc[4 * h + e] = v0[e]; xsh[4 * h + e] = v1[e]; }
        }
        auto apply = [&](auto ACT) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z = fmaf(v.get(e), xsc[e], xsh[e]);
            o[e] = decltype(ACT)::value == 1 ? fmaxf(z, 0.f) : decltype(ACT)::value == 2 ? gelu_erf(z) : z;
          }
          v.set_all(o);
        };
        if (a.x_act == 1) apply(std::integral_constant<int, 1>{});
        else if (a.x_act == 2) apply(std::integral_constant<int, 2>{});
        else apply(std::integral_constant<int, 0>{});
      }
      if constexpr (XPRE) { if (xoff[i] == SOOB) v.raw = {0, 0, 0, 0}; }      // (plain operand: an out-of-range load returned zeros)
      if (idx < HHP * 4) v.store(XH + (idx >> 2) * HLD + (idx & 3) * 8);
    }
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      const int idx = tid + i * NT_;
      Vec<bf16_t> v = rd[i];
      if constexpr (fuse) {
        if (a.bn_act == 1) apply_chunk(std::integral_constant<int, 1>{}, i, v, vzk[i]);
        else if (a.bn_act == 2) apply_chunk(std::integral_constant<int, 2>{}, i, v, vzk[i]);
        else apply_chunk(std::integral_constant<int, 0>{}, i, v, vzk[i]);
        vdk[i] = v;
        soff[i] = writer ? doff[i] : SOOB;
        if (doff[i] == SOOB) v.raw = {0, 0, 0, 0};             // (draw of an out-of-range pixel is the constant term, not zero)
      }
      v.store(DS + (idx >> 2) * HLD + (idx & 3) * 8);
    }
  };
  // draw / dz of the tile just staged go to global memory AFTER the next tile's loads have been issued: vmcnt counts in order, so
  // stores issued before those loads would have to complete before the loads can be waited for.  Unconditional bounds-checked
  // buffer stores (non-writer blocks, pixels outside the image and a null dres get the out-of-range offset: dropped).
  auto flush_stores = [&]() {
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, vdk[i].raw), rdraw, soff[i], 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, vzk[i].raw), rdres,
                                             a.dres_out ? soff[i] : SOOB, 0, 0);
    }
  };
  // this lane's transpose-read base rows: dout tile row (pixel) / halo row of tap (0,0) for the 32-pixel k-step 0
  const int i4 = l15 >> 2, c4 = (l15 & 3) * 4;
  // K-slot map of a 32-pixel step (see SlabFrag): first read = pixels 4g..4g+3 of image row 2ks, second = same of row 2ks+1
  // (the wave group's K-steps 2wg, 2wg + 1 are folded into the base rows; the taps are in the standard order - halo_wgrad_eligible -
  // so every tap / K-step is an IMMEDIATE offset of the transpose reads: with run-time dy / dx the compiler hoisted 18 address
  // registers out of the tile loop)
  const bf16_t* dbase = DS + (grp * 4 + i4 + 2 * wg * 32) * HLD + wm * 16 + c4;
  const bf16_t* xbase = XH + ((HTW + 2) * (1 + 4 * wg) + grp * 4 + 1 + i4) * HLD + wn * 16 + c4;
  auto tr8 = [&](const bf16_t* p, int hi_rows) {   // 8 K values of this lane's column: two transpose reads `hi_rows` rows apart
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + hi_rows * HLD));
    union { struct { v4s a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
  };

  if (t_begin < t_end) load_tile(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    store_tile();
    __syncthreads();
    if (t + 1 < t_end) load_tile(t + 1);
    if constexpr (fuse) flush_stores();
#pragma unroll
    for (int kk = 0; kk < HNPX / 64; ++kk) {                  // 32 pixels = image rows 2ks, 2ks+1 of the tile
      const bf16x8 fa = tr8(dbase + kk * 32 * HLD, HTW);
#ifdef RSSF_HALO_WG_DBG_ONEREAD      // timing builds only: ONE input fragment read per K-step instead of nine (what the transposing reads cost)
      const bf16x8 fb0 = tr8(xbase + ((2 * kk) * (HTW + 2)) * HLD, HTW + 2);
#endif
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
#ifdef RSSF_HALO_WG_DBG_ONEREAD
        const bf16x8 fb = fb0;
#else
        const bf16x8 fb = tr8(xbase + ((2 * kk + tap / 3 - 1) * (HTW + 2) + tap % 3 - 1) * HLD, HTW + 2);
#endif
        acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[tap], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // fold group 1 into group 0 (two passes of <= 5 taps through the now idle staging LDS), then group 0 writes the partials
  f32x4* fold = reinterpret_cast<f32x4*>(lds_raw);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int t0 = pass * 5, t1 = pass == 0 ? 5 : 9;
    if (wg == 1)
      for (int tap = t0; tap < t1; ++tap) fold[((tap - t0) * 4 + w4) * 64 + lane] = acc[tap];
    __syncthreads();
    if (wg == 0)
      for (int tap = t0; tap < t1; ++tap) {
        const f32x4 o = fold[((tap - t0) * 4 + w4) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tap][r] += o[r];
      }
    __syncthreads();
  }
  if (wg == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * 16 + grp * 4 + r, ci = ci0 + wn * 16 + l15;
        if (co < a.Cout && ci < a.Cin) a.partial[(((int64_t)range * 9 + tap) * a.Cout + co) * a.Cin + ci] = acc[tap][r];
      }
  }
}

// two resident blocks per CU (<= 128 VGPRs): a block alone keeps ONE tile's loads in flight under its MFMAs, i.e. it runs at the
// latency of its loads; the second block - of the same problem, or of another problem of a grouped launch - fills that time
template <bool FUSE, bool RES, bool XPRE = false>
__global__ void __launch_bounds__(HWG_THREADS, 4) conv3x3_wgrad_halo_kernel(WgradHaloArgs a) {
  const int64_t q = xcd_logical(blockIdx.x, a.xcd_per);       // the (co, ci) tiles of one pixel range share an XCD's L2
  if (q >= a.total) return;
  wgrad_halo_block<FUSE, RES, XPRE>(a, q);
}
// GROUPED launch (rssf_conv3x3_wgrad_group, see conv3x3_halo_group_kernel in conv_halo.hip): problem i owns the block indices
// [start[i], start[i+1]) of every XCD.  Every problem's blocks do the same amount of work (a run of tiles_per_block spatial tiles of
// one 32 x 32 channel pair), whatever its resolution.
struct WgradHaloGroupArgs {
  WgradHaloArgs it[RSSF_GROUP_MAX];
  int start[RSSF_GROUP_MAX + 1];
  int n;
};
template <bool FUSE, bool RES, bool XPRE>
__global__ void __launch_bounds__(HWG_THREADS, 4) conv3x3_wgrad_halo_group_kernel(WgradHaloGroupArgs g) {
  const unsigned xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  int i = 0;
#pragma unroll
  for (int k = 1; k < RSSF_GROUP_MAX; ++k)
    if (k < g.n && idx >= (unsigned)g.start[k]) i = k;
  const WgradHaloArgs& a = g.it[i];
  const int64_t q = (int64_t)xcd * a.xcd_per + (idx - (unsigned)g.start[i]);
  if (q >= a.total) return;
  wgrad_halo_block<FUSE, RES, XPRE>(a, q);
}

// spatial tiles per block: 8 when that still yields >= 256 blocks, fewer for small problems
int halo_tiles_per_block(int64_t ntiles, int npairs, int max_tpb = 8, int min_blocks = 256) {
  int tpb = max_tpb;
  while (tpb > 1 && ((ntiles + tpb - 1) / tpb) * npairs < min_blocks) tpb >>= 1;
  return tpb;
}
int halo_ksplit(int B, int H, int W, int Cout, int Cin) {
  const int64_t ntiles = (int64_t)B * ((H + HTH - 1) / HTH) * ((W + HTW - 1) / HTW);
  const int npairs = ((Cout + HCT - 1) / HCT) * ((Cin + HCT - 1) / HCT);
  const int tpb = halo_tiles_per_block(ntiles, npairs);
  return (int)((ntiles + tpb - 1) / tpb);
}
bool halo_wgrad_eligible(int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, int nsrc, const int* dy, const int* dx) {
  if (stride != 1 || IH != OH || IW != OW || ntaps != 9 || nsrc != 1 || (Cin % 8) != 0 || (Cout % 8) != 0) return false;
  for (int t = 0; t < 9; ++t)
    if (dy[t] != t / 3 - 1 || dx[t] != t % 3 - 1) return false;       // the kernel's tap offsets are compile-time
  return true;
}

// Tiling: 32x32 output tile with all 9 taps of a 3x3 in registers for the 32-channel layers (few FLOPs per pixel:
// stage dout once per slab); 64x64 / 128x128 tiles with ONE tap per block for wider layers (light registers -> many
// resident blocks hide the gather latency; 128x128 doubles the FLOPs per staged byte for the MLP's 128-channel convs).
int tile_of(int cout, int cin) {
  const int m = cout < cin ? cout : cin;
  return (cout <= 32 && cin <= 32) ? 32 : (m >= 128 ? 128 : 64);
}
bool taps_in_registers(int cout, int cin) { return tile_of(cout, cin) == 32; }

// split-K factor: 512..1024 blocks in flight, at least 4 slabs of 64 pixels each
int pick_ksplit(int cout, int cin, int ntaps, int64_t M) {
  const int t = tile_of(cout, cin);
  int64_t par = (int64_t)((cout + t - 1) / t) * ((cin + t - 1) / t);
  if (!taps_in_registers(cout, cin)) par *= ntaps;
  constexpr int blocks128 = 512, blocks64 = 1024;           // swept in the step, round 4: 1 024 blocks 31.70 ms, 768: 31.66, 512: 31.84, 256: 32.83
  int64_t ks = (t == 128 ? blocks128 : blocks64) / par;  // measured on MI355X (tools/wgrad_bench.py sweep)
  const int64_t maxks = (M + 4 * KP - 1) / (4 * KP);
  if (ks > maxks) ks = maxks;
  if (ks < 1) ks = 1;
  return (int)ks;
}

template <typename T, int TMN>
int launch_group(WgradArgs& a, int tap0, int ntap, hipStream_t st) {
  a.tap0 = tap0; a.ntap = ntap;
  a.ctiles_m = (a.Cout + TMN - 1) / TMN; a.ctiles_n = (a.Cin + TMN - 1) / TMN;
  const int tiles = a.ctiles_m * a.ctiles_n;
  a.inner = TMN == 32 ? tiles : tiles * ntap;
  a.xcd_per = xcd_per((int64_t)a.inner * a.ksplit);
  dim3 grid((unsigned)a.xcd_per * 8);
  const bool vok = (a.Cout % Vec<T>::N) == 0 && (a.Cin % Vec<T>::N) == 0;
  if (TMN == 32 && ntap != 1) {
    if (vok) conv_wgrad_kernel<T, TMN, (TMN == 32 ? 9 : 1), true><<<grid, 256, 0, st>>>(a);
    else conv_wgrad_kernel<T, TMN, (TMN == 32 ? 9 : 1), false><<<grid, 256, 0, st>>>(a);
  } else {
    if (TMN == 128 && sizeof(T) == 2 && vok && a.partial && ntap >= 2) {
      // one round of the chip: tap groups x ksplit <= CUs (never more planes than the workspace was sized for)
      static const int cus = [] { int d = 0, v = 256; (void)hipGetDevice(&d); if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256; return v; }();
      static hipError_t e2 = hipFuncSetAttribute((const void*)conv_wgrad8x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG8X2_LDS);
      if (e2 != hipSuccess) { set_error("conv_wgrad8x2: cannot raise the LDS limit: %s", hipGetErrorString(e2)); return RSSF_ERR_LAUNCH; }
      const int groups = (ntap + 1) / 2;
      int ks = cus / (tiles * groups);
      if (ks > a.ksplit) ks = a.ksplit;
      if (ks < 1) ks = 1;
      a.ksplit = ks;
      a.inner = tiles * groups;
      a.xcd_per = xcd_per((int64_t)a.inner * a.ksplit);
      conv_wgrad8x2_kernel<<<dim3((unsigned)a.xcd_per * 8), 512, WG8X2_LDS, st>>>(a);
    } else if (TMN == 128 && sizeof(T) == 2 && vok && a.partial) conv_wgrad8_kernel<<<grid, 512, 0, st>>>(a);
    else if (vok) conv_wgrad_kernel<T, TMN, 1, true><<<grid, 256, 0, st>>>(a);
    else conv_wgrad_kernel<T, TMN, 1, false><<<grid, 256, 0, st>>>(a);
  }
  return check_launch("conv_wgrad");
}

template <typename T>
int launch_all(WgradArgs& a, int ntaps, rssf_wgrad_reduce_job* defer, hipStream_t st) {
  // tap groups: consecutive taps of one source conv (1 for a 1x1, 9 for a 3x3)
  int t = 0;
  const int tile = tile_of(a.Cout, a.Cin);
  if (tile != 32) {          // one tap per block: all taps (of all source convs) in a single launch
    const int rc = tile == 64 ? launch_group<T, 64>(a, 0, ntaps, st) : launch_group<T, 128>(a, 0, ntaps, st);
    if (rc) return rc;
    t = ntaps;
  }
  while (t < ntaps) {
    int n = 1;
    while (t + n < ntaps && n < 9 && a.src_of_tap[t + n] == a.src_of_tap[t]) ++n;
    const int rc = tile == 32 ? launch_group<T, 32>(a, t, n, st) : tile == 64 ? launch_group<T, 64>(a, t, n, st)
                                                                              : launch_group<T, 128>(a, t, n, st);
    if (rc) return rc;
    t += n;
  }
  if (a.partial) return finish_reduce(a, defer, st);
  if (defer) defer->partial = nullptr;      // atomics path: nothing left to do
  return RSSF_OK;
}

}  // namespace

extern "C" int64_t rssf_conv_wgrad_workspace_elems(int B, int OH, int OW, int Cin, int Cout, int ntaps) {
  int ks = pick_ksplit(Cout, Cin, ntaps, (int64_t)B * OH * OW);
  if (ntaps == 9) { const int hk = halo_ksplit(B, OH, OW, Cout, Cin); if (hk > ks) ks = hk; }   // either kernel may run
  static const int dy0[1] = {0}, dx0[1] = {0};
  if (ntaps == 1 && wgrad_pw_eligible(B, OH, OW, Cin, OH, OW, Cout, 1, 1, dy0, dx0)) {       // (a stride-1 point-wise layer or not: the bound holds)
    const int pk = wgrad_pw_ksplit(B, OH, OW, Cin, Cout);
    if (pk > ks) ks = pk;
  }
  {
    int sdy[9], sdx[9];
    for (int t = 0; t < 9; ++t) { sdy[t] = t / 3 - 1; sdx[t] = t % 3 - 1; }
    if (ntaps == 9 && wgrad_stem_eligible(B, 2 * OH, 2 * OW, Cin, OH, OW, Cout, 2, 9, sdy, sdx)) {     // (a stride-2 stem layer or not: the bound holds)
      const int sk = wgrad_stem_ksplit(B, OH, OW, Cin, Cout);
      if (sk > ks) ks = sk;
    }
  }
  const int64_t generic = (int64_t)ks * ntaps * Cout * Cin, planes = wgrad_planes_workspace_elems(B, OH, OW, Cin, Cout, ntaps);
  return generic > planes ? generic : planes;       // either first stage may run (rssf_conv_wgrad_planes takes the same workspace)
}

namespace {
using BnApply = rssf::cv::WgradBn;               // arguments of rssf_bn_bwd_apply (see WgradHaloArgs::bn_*)
struct XPreAct { const float* ss; int act; };
// arguments of the halo-tiled kernel for one problem; ksplit_out = partial planes the second stage has to fold
WgradHaloArgs make_wgrad_halo(const void* dout, const void* in, float* workspace, int B, int H, int W, int Cin, int Cout, const BnApply* bn,
                              const XPreAct* xpre, int& ksplit_out, int max_tpb = 8, int min_blocks = 256) {
  WgradHaloArgs h;
  memset(&h, 0, sizeof(h));
  h.dout = (const bf16_t*)dout; h.in = (const bf16_t*)in; h.partial = workspace;
  h.B = B; h.H = H; h.W = W; h.Cin = Cin; h.Cout = Cout;
  h.tiles_y = (H + HTH - 1) / HTH; h.tiles_x = (W + HTW - 1) / HTW; h.ntiles = B * h.tiles_y * h.tiles_x;
  h.ptiles_n = (Cin + HCT - 1) / HCT; h.npairs = ((Cout + HCT - 1) / HCT) * h.ptiles_n;
  h.tiles_per_block = halo_tiles_per_block(h.ntiles, h.npairs, max_tpb, min_blocks);
  ksplit_out = (h.ntiles + h.tiles_per_block - 1) / h.tiles_per_block;
  h.total = (int64_t)ksplit_out * h.npairs;
  h.xcd_per = xcd_per(h.total);
  for (int t = 0; t < 9; ++t) { h.dy[t] = t / 3 - 1; h.dx[t] = t % 3 - 1; }
  if (bn) {                  // the BatchNorm-backward apply rides in this kernel's staging of the output-gradient tile
    h.bn_dy = (const bf16_t*)bn->dy; h.bn_raw = (const bf16_t*)bn->raw; h.bn_res = (const bf16_t*)bn->res;
    h.bn_ss = bn->ss; h.bn_mi = bn->mi; h.bn_sums = bn->sums;
    h.draw_out = (bf16_t*)bn->draw; h.dres_out = (bf16_t*)bn->dres; h.dgamma = bn->dgamma; h.dbeta = bn->dbeta;
    h.bn_n = (float)bn->n; h.bn_pscale = bn->pscale; h.bn_act = bn->act; h.bn_training = bn->training;
  }
  h.x_ss = xpre ? xpre->ss : nullptr; h.x_act = xpre ? xpre->act : 0;
  return h;
}
int conv_wgrad_impl(const void* dout, const void* in, float* dw0, float* dw1, float* dw2, const int* ksizes,
                    int nsrc, const int* src_of_tap, const int* kpos_of_tap, const int* alias_of_tap, float* dbias,
                    float* workspace, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps,
                    const int* dy, const int* dx, rssf_wgrad_reduce_job* defer_reduce, const BnApply* bn, const XPreAct* xpre, int dtype_flags,
                    void* stream, const float* w_dg = nullptr, void* dx_dg = nullptr, float* st_sums = nullptr) {
  const int dtype = dtype_flags & 0xff;
  const bool generic = (dtype_flags & RSSF_CONV_GENERIC) != 0;       // the narrow point-wise kernel off (rssf.h): parity tests, A/B runs
  RSSF_REQUIRE(dout && in && dw0 && ksizes && src_of_tap && kpos_of_tap && dy && dx && nsrc >= 1 && nsrc <= 3 && ntaps >= 1 &&
                   ntaps <= MAX_TAPS && B > 0 && IH > 0 && IW > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0 && stride >= 1,
               "conv_wgrad: bad arguments");
  RSSF_REQUIRE((int64_t)B * OH * OW * Cout < ((int64_t)1 << 30) && (int64_t)B * IH * IW * Cin < ((int64_t)1 << 30),
               "conv_wgrad: activation tensors of 2^30 or more elements are not supported (32-bit byte offsets, buffer bounds)");
  WgradArgs a;
  a.dout = dout; a.in = in; a.dw[0] = dw0; a.dw[1] = dw1; a.dw[2] = dw2; a.dbias = dbias;
  for (int i = 0; i < 3; ++i) a.ks[i] = i < nsrc ? ksizes[i] : 1;
  a.taps.n = ntaps;
  for (int t = 0; t < ntaps; ++t) {
    a.src_of_tap[t] = src_of_tap[t]; a.kpos_of_tap[t] = kpos_of_tap[t];
    a.taps.dy[t] = dy[t]; a.taps.dx[t] = dx[t];
    for (int e = 0; e < 4; ++e) {
      a.alias[t][e] = alias_of_tap ? alias_of_tap[t * 4 + e] : -1;
      RSSF_REQUIRE((e & 1) || a.alias[t][e] < nsrc, "conv_wgrad: alias source out of range");
    }
  }
  a.B = B; a.IH = IH; a.IW = IW; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.stride = stride;
  a.partial = workspace; a.ntaps_total = ntaps;
  a.ksplit = pick_ksplit(Cout, Cin, ntaps, (int64_t)B * OH * OW);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_BF16 && workspace && !dbias && halo_wgrad_eligible(IH, IW, Cin, OH, OW, Cout, stride, ntaps, nsrc, dy, dx)) {
#ifndef RSSF_HALO_WG_TPB
#define RSSF_HALO_WG_TPB 8
#endif
#ifndef RSSF_HALO_WG_MIN
#define RSSF_HALO_WG_MIN 256
#endif
    constexpr int stpb = RSSF_HALO_WG_TPB, smin = RSSF_HALO_WG_MIN;          // tiles per block / fewest blocks (swept in round 4: tools/wgrad_bench.py)
    const WgradHaloArgs h = make_wgrad_halo(dout, in, workspace, B, IH, IW, Cin, Cout, bn, xpre, a.ksplit, stpb, smin);
    const dim3 hgrid((unsigned)h.xcd_per * 8);
    if (xpre) {
      if (!bn) conv3x3_wgrad_halo_kernel<false, false, true><<<hgrid, HWG_THREADS, 0, st>>>(h);
      else if (bn->res) conv3x3_wgrad_halo_kernel<true, true, true><<<hgrid, HWG_THREADS, 0, st>>>(h);
      else conv3x3_wgrad_halo_kernel<true, false, true><<<hgrid, HWG_THREADS, 0, st>>>(h);
    } else {
      if (!bn) conv3x3_wgrad_halo_kernel<false, false><<<hgrid, HWG_THREADS, 0, st>>>(h);
      else if (bn->res) conv3x3_wgrad_halo_kernel<true, true><<<hgrid, HWG_THREADS, 0, st>>>(h);
      else conv3x3_wgrad_halo_kernel<true, false><<<hgrid, HWG_THREADS, 0, st>>>(h);
    }
    if (int rc = check_launch("conv3x3_wgrad_halo")) return rc;
    return finish_reduce(a, defer_reduce, st);
  }
  if (xpre && dtype == RSSF_BF16 && workspace && wgrad_pw_preact_eligible(B, IH, IW, Cin, OH, OW, Cout, stride, ntaps, dy, dx)) {
    if (bn) {                  // (this kernel has no form with both: the layer's own apply as the separate pass, `dout` is its `draw`)
      const int rc = rssf_bn_bwd_apply(bn->dy, bn->raw, bn->ss, bn->mi, bn->sums, bn->res, bn->draw, bn->dres, bn->dgamma, bn->dbeta,
                                       (int64_t)B * OH * OW, Cout, bn->act, bn->n, bn->training, bn->pscale, dtype, stream);
      if (rc) return rc;
    }
    a.ksplit = wgrad_pw_ksplit(B, OH, OW, Cin, Cout);
    if (int rc = launch_wgrad_pw(dout, in, workspace, dbias, B, OH, OW, Cin, Cout, a.ksplit, nullptr, st, xpre->ss, xpre->act, w_dg, dx_dg, st_sums))
      return rc;
    return finish_reduce(a, defer_reduce, st);
  }
  if (xpre) { set_error("conv_wgrad: no kernel with a pre-activation input operand for this shape (ask rssf_conv_wgrad_preact_supported)"); return RSSF_ERR_UNSUPPORTED; }
#ifndef RSSF_WGRAD_STEM_DISABLE     // (A/B builds: tools/ab_lib_flags.sh)
  if (dtype == RSSF_BF16 && workspace && !generic && !dbias && !xpre && !w_dg && (!bn || (!bn->res && !bn->dres)) &&
      wgrad_stem_eligible(B, IH, IW, Cin, OH, OW, Cout, stride, ntaps, dy, dx)) {
    // the stem's first convolution: apply + weight gradient in one pass over dy / raw, `draw` only on request (conv_wgrad_stem.hip)
    a.ksplit = wgrad_stem_ksplit(B, OH, OW, Cin, Cout);
    if (int rc = launch_wgrad_stem(dout, in, workspace, B, IH, IW, Cin, OH, OW, Cout, a.ksplit, bn, (dtype_flags & RSSF_WGRAD_NO_DRAW) == 0, st)) return rc;
    return finish_reduce(a, defer_reduce, st);
  }
#endif
#ifndef RSSF_WGRAD_PW_DISABLE       // (A/B builds: tools/ab_lib_flags.sh)
  if (dtype == RSSF_BF16 && workspace && !generic && wgrad_pw_eligible(B, IH, IW, Cin, OH, OW, Cout, stride, ntaps, dy, dx)) {
    // narrow point-wise layers: one block per pixel range streams both operands once (conv_wgrad_pw.hip), the apply rides along
    a.ksplit = wgrad_pw_ksplit(B, OH, OW, Cin, Cout);
    if (int rc = launch_wgrad_pw(dout, in, workspace, dbias, B, OH, OW, Cin, Cout, a.ksplit, bn, st, nullptr, 0, w_dg, dx_dg)) return rc;
    return finish_reduce(a, defer_reduce, st);
  }
#endif
  if (w_dg || dx_dg) { set_error("conv_wgrad: no kernel with a fused data gradient for this call (ask rssf_conv_wgrad_bnapply_dgrad_supported)"); return RSSF_ERR_UNSUPPORTED; }
  if (bn) {                    // no kernel with a fused apply for this shape: the separate pass, then the plain weight gradient
    const int rc = rssf_bn_bwd_apply(bn->dy, bn->raw, bn->ss, bn->mi, bn->sums, bn->res, bn->draw, bn->dres, bn->dgamma, bn->dbeta,
                                     (int64_t)B * OH * OW, Cout, bn->act, bn->n, bn->training, bn->pscale, dtype, stream);
    if (rc) return rc;
  }
  if (dtype == RSSF_F32) return launch_all<float>(a, ntaps, defer_reduce, st);
  if (dtype == RSSF_BF16) return launch_all<bf16_t>(a, ntaps, defer_reduce, st);
  set_error("conv_wgrad: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}
}  // namespace

extern "C" int rssf_conv_wgrad(const void* dout, const void* in, float* dw0, float* dw1, float* dw2, const int* ksizes,
                               int nsrc, const int* src_of_tap, const int* kpos_of_tap, const int* alias_of_tap, float* dbias,
                               float* workspace, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps,
                               const int* dy, const int* dx, rssf_wgrad_reduce_job* defer_reduce, int dtype, void* stream) {
  return conv_wgrad_impl(dout, in, dw0, dw1, dw2, ksizes, nsrc, src_of_tap, kpos_of_tap, alias_of_tap, dbias, workspace, B, IH, IW, Cin, OH,
                         OW, Cout, stride, ntaps, dy, dx, defer_reduce, nullptr, nullptr, dtype, stream);
}

extern "C" int rssf_conv_wgrad_preact(const void* dout, const void* in_raw, const float* in_scale_shift, int in_act, float* dw0, float* dw1,
                                      float* dw2, const int* ksizes, int nsrc, const int* src_of_tap, const int* kpos_of_tap,
                                      const int* alias_of_tap, float* dbias, float* workspace, int B, int IH, int IW, int Cin, int OH, int OW,
                                      int Cout, int stride, int ntaps, const int* dy, const int* dx, rssf_wgrad_reduce_job* defer_reduce,
                                      int dtype, void* stream) {
  RSSF_REQUIRE(in_scale_shift && in_act >= 0 && in_act <= 2, "conv_wgrad_preact: bad pre-activation arguments");
  const XPreAct xp = {in_scale_shift, in_act};
  return conv_wgrad_impl(dout, in_raw, dw0, dw1, dw2, ksizes, nsrc, src_of_tap, kpos_of_tap, alias_of_tap, dbias, workspace, B, IH, IW, Cin, OH,
                         OW, Cout, stride, ntaps, dy, dx, defer_reduce, nullptr, &xp, dtype, stream);
}

extern "C" int rssf_conv_wgrad_bnapply(const void* bn_dy, const void* bn_raw, const float* bn_scale_shift, const float* bn_mean_invstd,
                                       const float* bn_sums, const void* bn_res_pre, void* draw, void* dres, float* dgamma, float* dbeta,
                                       int bn_act, double bn_n, int bn_training, float param_grad_scale, const void* in,
                                       const float* in_scale_shift, int in_act, float* dw0,
                                       float* dw1, float* dw2, const int* ksizes, int nsrc, const int* src_of_tap, const int* kpos_of_tap,
                                       const int* alias_of_tap, float* dbias, float* workspace, int B, int IH, int IW, int Cin, int OH,
                                       int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx,
                                       rssf_wgrad_reduce_job* defer_reduce, int dtype, void* stream) {
  RSSF_REQUIRE(bn_dy && bn_raw && bn_scale_shift && bn_mean_invstd && bn_sums && draw && bn_act >= 0 && bn_act <= 2,
               "conv_wgrad_bnapply: bad BatchNorm arguments");
  RSSF_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "conv_wgrad_bnapply: dgamma and dbeta go together");
  const BnApply bn = {bn_dy, bn_raw, bn_scale_shift, bn_mean_invstd, bn_sums, bn_res_pre, draw, dres, dgamma, dbeta, bn_n, bn_act,
                      bn_training, param_grad_scale};
  // `draw` doubles as the weight gradient's output-gradient operand: it is complete when this call returns to the stream
  const XPreAct xp = {in_scale_shift, in_act};
  return conv_wgrad_impl(draw, in, dw0, dw1, dw2, ksizes, nsrc, src_of_tap, kpos_of_tap, alias_of_tap, dbias, workspace, B, IH, IW, Cin, OH,
                         OW, Cout, stride, ntaps, dy, dx, defer_reduce, &bn, in_scale_shift ? &xp : nullptr, dtype, stream);
}

extern "C" int rssf_conv_wgrad_preact_dgrad_supported(int B, int H, int W, int Cin, int Cout, int dtype) {
  static const int z1[1] = {0};
  return dtype == RSSF_BF16 && wgrad_pw_preact_eligible(B, H, W, Cin, H, W, Cout, 1, 1, z1, z1) ? 1 : 0;
}

extern "C" int rssf_conv_wgrad_preact_dgrad(const void* dout, const void* in_raw, const float* in_scale_shift, int in_act, const float* weight,
                                            void* dx_out, float* in_bn_sums, float* dw, float* dbias, float* workspace, int B, int H, int W,
                                            int Cin, int Cout, rssf_wgrad_reduce_job* defer_reduce, int dtype, void* stream) {
  RSSF_REQUIRE(dout && in_raw && in_scale_shift && in_act >= 0 && in_act <= 2 && weight && dx_out && in_bn_sums && dw && workspace,
               "conv_wgrad_preact_dgrad: bad arguments");
  RSSF_REQUIRE(rssf_conv_wgrad_preact_dgrad_supported(B, H, W, Cin, Cout, dtype) == 1, "conv_wgrad_preact_dgrad: unsupported shape (ask _supported)");
  static const int ks1[1] = {1}, z1[1] = {0};
  const XPreAct xp = {in_scale_shift, in_act};
  return conv_wgrad_impl(dout, in_raw, dw, nullptr, nullptr, ks1, 1, z1, z1, nullptr, dbias, workspace, B, H, W, Cin, H, W, Cout, 1, 1, z1, z1,
                         defer_reduce, nullptr, &xp, dtype, stream, weight, dx_out, in_bn_sums);
}

extern "C" int rssf_conv_wgrad_bnapply_dgrad_supported(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps,
                                                       const int* dy, const int* dx, int has_res_pre, int dtype) {
#ifdef RSSF_WGRAD_PW_DISABLE
  return 0;
#else
  return dtype == RSSF_BF16 && dy && dx && !has_res_pre && wgrad_pw_dgrad_eligible(B, IH, IW, Cin, OH, OW, Cout, stride, ntaps, dy, dx) ? 1 : 0;
#endif
}

extern "C" int rssf_conv_wgrad_bnapply_dgrad(const void* bn_dy, const void* bn_raw, const float* bn_scale_shift, const float* bn_mean_invstd,
                                             const float* bn_sums, void* draw, float* dgamma, float* dbeta, int bn_act, double bn_n,
                                             int bn_training, float param_grad_scale, const void* in, const float* weight, void* dx_out,
                                             float* dw, float* dbias, float* workspace, int B, int H, int W, int Cin, int Cout,
                                             rssf_wgrad_reduce_job* defer_reduce, int dtype, void* stream) {
  RSSF_REQUIRE(bn_dy && bn_raw && bn_scale_shift && bn_mean_invstd && bn_sums && draw && bn_act >= 0 && bn_act <= 2 && weight && dx_out && dw &&
                   workspace, "conv_wgrad_bnapply_dgrad: bad arguments");
  RSSF_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "conv_wgrad_bnapply_dgrad: dgamma and dbeta go together");
  static const int ks1[1] = {1}, z1[1] = {0};
  const BnApply bn = {bn_dy, bn_raw, bn_scale_shift, bn_mean_invstd, bn_sums, nullptr, draw, nullptr, dgamma, dbeta, bn_n, bn_act, bn_training,
                      param_grad_scale};
  return conv_wgrad_impl(draw, in, dw, nullptr, nullptr, ks1, 1, z1, z1, nullptr, dbias, workspace, B, H, W, Cin, H, W, Cout, 1, 1, z1, z1,
                         defer_reduce, &bn, nullptr, dtype, stream, weight, dx_out);
}

extern "C" int rssf_conv_wgrad_reduce_blocks(const rssf_wgrad_reduce_job* job) {
  if (!job || !job->partial) return 0;
  return (int)(((int64_t)job->ntaps * job->cout * job->cin + RI - 1) / RI);
}

extern "C" int rssf_conv_wgrad_reduce_batch(const rssf_wgrad_reduce_job* jobs, const int* block_map, int nblocks, void* stream) {
  RSSF_REQUIRE(jobs && block_map && nblocks > 0, "conv_wgrad_reduce_batch: bad arguments");
  wgrad_reduce_batch_kernel<<<(unsigned)nblocks, 256, 0, (hipStream_t)stream>>>(jobs, block_map);
  return check_launch("conv_wgrad_reduce_batch");
}

extern "C" int rssf_conv_wgrad_preact_supported(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, int nsrc,
                                                const int* dy, const int* dx, int has_bias, int dtype) {
  if (dtype != RSSF_BF16 || !dy || !dx) return 0;
  if (!has_bias && halo_wgrad_eligible(IH, IW, Cin, OH, OW, Cout, stride, ntaps, nsrc, dy, dx)) return 1;
  // MlpDWBN's fc2 (plain weight gradient only - its own BatchNorm is not applied in the same launch: rssf_conv_wgrad_preact)
  return nsrc == 1 && wgrad_pw_preact_eligible(B, IH, IW, Cin, OH, OW, Cout, stride, ntaps, dy, dx) ? 2 : 0;
}

// ---- grouped 3x3 weight gradients (rssf.h "Grouped launches") ----------------------------------------------------------------
extern "C" int rssf_conv3x3_wgrad_group(const rssf_wgrad3x3_item* items, int n, int dtype, void* stream) {
  RSSF_REQUIRE(items && n >= 1, "conv3x3_wgrad_group: bad arguments");
  int dy[9], dx[9];
  for (int t = 0; t < 9; ++t) { dy[t] = t / 3 - 1; dx[t] = t % 3 - 1; }
  static const int ks9[1] = {3}, src9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, kpos9[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
  bool grouped = n >= 2 && n <= RSSF_GROUP_MAX && dtype == RSSF_BF16;
  for (int i = 0; i < n; ++i) {
    const rssf_wgrad3x3_item& it = items[i];
    RSSF_REQUIRE(it.in && it.dw && it.B > 0 && it.H > 0 && it.W > 0 && it.Cin > 0 && it.Cout > 0 && (it.bn_dy ? it.draw != nullptr : it.dout != nullptr),
                 "conv3x3_wgrad_group: bad item %d", i);
    RSSF_REQUIRE(!it.bn_dy || (it.bn_raw && it.bn_ss && it.bn_mi && it.bn_sums && it.bn_act >= 0 && it.bn_act <= 2 && (it.dgamma == nullptr) == (it.dbeta == nullptr)),
                 "conv3x3_wgrad_group: bad BatchNorm arguments (item %d)", i);
    RSSF_REQUIRE(!it.in_ss || it.bn_dy, "conv3x3_wgrad_group: a pre-activation input operand needs the fused apply (item %d)", i);
    grouped = grouped && it.workspace && halo_wgrad_eligible(it.H, it.W, it.Cin, it.H, it.W, it.Cout, 1, 9, 1, dy, dx) &&
              (int64_t)it.B * it.H * it.W * it.Cout < ((int64_t)1 << 30) && (int64_t)it.B * it.H * it.W * it.Cin < ((int64_t)1 << 30) &&
              (it.bn_dy != nullptr) == (items[0].bn_dy != nullptr) && (it.bn_res != nullptr) == (items[0].bn_res != nullptr) &&
              (it.in_ss != nullptr) == (items[0].in_ss != nullptr);
  }
  hipStream_t st = (hipStream_t)stream;
  if (grouped) {
    WgradHaloGroupArgs g;
    g.n = n;
    int idx = 0;
    WgradArgs ra[RSSF_GROUP_MAX];
    for (int i = 0; i < n; ++i) {
      const rssf_wgrad3x3_item& it = items[i];
      const BnApply bn = {it.bn_dy, it.bn_raw, it.bn_ss, it.bn_mi, it.bn_sums, it.bn_res, it.draw, it.dres, it.dgamma, it.dbeta, it.bn_n, it.bn_act,
                          it.bn_training, it.pscale};
      const XPreAct xp = {it.in_ss, it.in_act};
      int ksplit = 1;
      // the problems of a group fill the chip TOGETHER: each may run longer tile runs per block (fewer split-K partial planes for
      // the second stage to fold: 3.5 GB per step with runs of 8) as long as it keeps >= gmin blocks
#ifndef RSSF_HALO_WG_GTPB
#define RSSF_HALO_WG_GTPB 16
#endif
#ifndef RSSF_HALO_WG_GMIN
#define RSSF_HALO_WG_GMIN 128
#endif
      constexpr int gtpb = RSSF_HALO_WG_GTPB, gmin = RSSF_HALO_WG_GMIN;
      g.it[i] = make_wgrad_halo(it.bn_dy ? it.draw : it.dout, it.in, it.workspace, it.B, it.H, it.W, it.Cin, it.Cout, it.bn_dy ? &bn : nullptr,
                                it.in_ss ? &xp : nullptr, ksplit, n >= 2 ? gtpb : 8, n >= 2 ? gmin : 256);
      g.start[i] = idx;
      idx += g.it[i].xcd_per;
      WgradArgs& a = ra[i];                          // what the second stage needs to know (make_job)
      memset(&a, 0, sizeof(a));
      a.partial = it.workspace; a.dw[0] = it.dw; a.ks[0] = 3; a.ks[1] = a.ks[2] = 1;
      a.ntaps_total = 9; a.Cout = it.Cout; a.Cin = it.Cin; a.ksplit = ksplit;
      for (int t = 0; t < MAX_TAPS; ++t) { a.src_of_tap[t] = 0; a.kpos_of_tap[t] = t < 9 ? t : 0; for (int e = 0; e < 4; ++e) a.alias[t][e] = -1; }
    }
    for (int k = n; k <= RSSF_GROUP_MAX; ++k) g.start[k] = idx;
    const dim3 grid((unsigned)idx * 8);
    const bool fuse = items[0].bn_dy != nullptr, res = items[0].bn_res != nullptr, xp = items[0].in_ss != nullptr;
    if (xp) {
      if (!fuse) conv3x3_wgrad_halo_group_kernel<false, false, true><<<grid, HWG_THREADS, 0, st>>>(g);
      else if (res) conv3x3_wgrad_halo_group_kernel<true, true, true><<<grid, HWG_THREADS, 0, st>>>(g);
      else conv3x3_wgrad_halo_group_kernel<true, false, true><<<grid, HWG_THREADS, 0, st>>>(g);
    } else {
      if (!fuse) conv3x3_wgrad_halo_group_kernel<false, false, false><<<grid, HWG_THREADS, 0, st>>>(g);
      else if (res) conv3x3_wgrad_halo_group_kernel<true, true, false><<<grid, HWG_THREADS, 0, st>>>(g);
      else conv3x3_wgrad_halo_group_kernel<true, false, false><<<grid, HWG_THREADS, 0, st>>>(g);
    }
    if (int rc = check_launch("conv3x3_wgrad_halo_group")) return rc;
    for (int i = 0; i < n; ++i)
      if (int rc = finish_reduce(ra[i], items[i].defer_reduce, st)) return rc;
    return RSSF_OK;
  }
  for (int i = 0; i < n; ++i) {
    const rssf_wgrad3x3_item& it = items[i];
    const BnApply bn = {it.bn_dy, it.bn_raw, it.bn_ss, it.bn_mi, it.bn_sums, it.bn_res, it.draw, it.dres, it.dgamma, it.dbeta, it.bn_n, it.bn_act,
                        it.bn_training, it.pscale};
    const XPreAct xp = {it.in_ss, it.in_act};
    const int rc = conv_wgrad_impl(it.bn_dy ? it.draw : it.dout, it.in, it.dw, nullptr, nullptr, ks9, 1, src9, kpos9, nullptr, nullptr, it.workspace, it.B,
                                   it.H, it.W, it.Cin, it.H, it.W, it.Cout, 1, 9, dy, dx, it.defer_reduce, it.bn_dy ? &bn : nullptr,
                                   it.in_ss ? &xp : nullptr, dtype, stream);
    if (rc) return rc;
  }
  return RSSF_OK;
}
