// Dilated multi-tap convolution on its pixel LATTICE (gfx950 MFMA, bf16): MlpDWBN's fused {1x1 + 3x3 dil 6 + 3x3 dil 12} sum
// (ffn_block.py:226-228, 250-257), forward and data gradient.
//
// Every tap offset of that sum is a multiple of g = 6 pixels, so the pixels with (y mod g, x mod g) = (ry, rx) form a closed
// sub-problem: on that lattice (22 x 22 points of a 128 x 128 map) the 17 distinct taps are the centre, the eight neighbours at
// distance 1 and the eight at distance 2 - a dense 5 x 5-footprint convolution.  The generic gather kernel (conv_fwd.hip) fetches and
// stages every activation tile once PER TAP (17 global gathers and 17 LDS images of each pixel, one workgroup barrier per tap and
// channel chunk; it sits at 0.34 of the MFMA peak, bound by LDS traffic and the staging itself).  Here a workgroup owns half a
// lattice class of one image (<= 11 x 22 points = 16 MFMA row tiles): per 32-channel chunk it stages the haloed lattice region ONCE
// (15 x 26 points, 96-byte rows: conflict-free ds_read_b128) and runs all 17 taps out of LDS - a tap is an IMMEDIATE offset of the
// A-fragment reads.  The weights never pass through LDS: a wave's B fragments are 16-byte buffer loads straight from the packed
// slabs (L2-resident, 557 KB), one tap ahead of their use - the vector-memory path is otherwise idle - so the tap loop has NO
// workgroup barrier: one per channel chunk.  The next chunk's region is fetched piece by piece under the tap loop into the second
// LDS buffer.  Per wave and tap: 8 ds_read_b128 + 4 buffer loads + 32 MFMAs (64 x 64 wave tile of a 256 x 128 block tile), ~0 VALU.
//
// Epilogue as in the gather kernel: + bias, fused BatchNorm statistics (forward) or BatchNorm-backward statistics of the producer
// (data gradient, rssf_conv_gather_bnbwd), output tile through LDS for 16-byte stores (a pixel's 128 channels are 256 contiguous bytes).
#include <mutex>
#include <type_traits>
#include <utility>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace rssf {
namespace cv {

namespace {

constexpr int LT_TAPS = 17;
constexpr int LYT = 11, LXT = 22;                       // lattice points of one tile (rows x columns)
constexpr int RY = LYT + 4, RP = LXT + 4;               // haloed region: two lattice points on every side
constexpr int ROWB = 96;                                // bytes per staged point (32 channels + 32 B pad)
constexpr int NPTS = RY * RP, REGION_BYTES = NPTS * ROWB;
constexpr int A_PIECES = 8;                    // 16-byte pieces per thread and chunk (see the staging map in the kernel)
static_assert(RP * 4 <= 128 && RP * 8 >= 128 && RY <= 2 * A_PIECES && RY >= 2 * A_PIECES - 1, "staging map: a region row is at most 128 pieces, 15 or 16 rows");
constexpr int BMT = 16, BNT = 128, LDC = BNT + 8;       // block tile: 16 row tiles (256 positions) x 128 channels
constexpr size_t LDS_BYTES = 2 * (size_t)REGION_BYTES;
static_assert(LDS_BYTES >= (size_t)BMT * 16 * LDC * 2, "the output tile reuses the staging buffers");
static_assert(LYT * LXT <= BMT * 16, "a tile's positions fit the block tile");

// canonical tap order: centre, ring 1, ring 2 (in lattice units)
__host__ __device__ constexpr int tap_ty(int c) { return c == 0 ? 0 : (((c - 1) % 8 < 3) ? -1 : ((c - 1) % 8 < 5) ? 0 : 1) * (c > 8 ? 2 : 1); }
__host__ __device__ constexpr int tap_tx(int c) {
  const int k = (c - 1) % 8;
  return c == 0 ? 0 : ((k == 0 || k == 3 || k == 5) ? -1 : (k == 1 || k == 6) ? 0 : 1) * (c > 8 ? 2 : 1);
}

template <typename F, int... I> __device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct LatticeArgs {
  const bf16_t* in; const bf16_t* wpk; bf16_t* out; const float* bias; float* stats; const bf16_t* addend;
  const bf16_t* bn_raw; const bf16_t* bn_res; const float* bn_ss; float* bn_sums; int bn_act;
  int B, H, W, Cin, Cout, CinP, CoutP;
  int g, nyt, nxt, xcd_per, total;
  int slab[LT_TAPS];                                    // packed-weight slab of canonical tap c
};

// ABL: timing experiments only (tools/lattice_bench.py; results are garbage): 1 no weight loads in the loop, 2 no A-fragment reads,
// 4 no region staging, 8 no MFMAs
template <int ABL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_lattice_kernel(LatticeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float sstat[2 * 2 * BNT];
  __shared__ int s_wof[LT_TAPS];
  __shared__ __attribute__((aligned(16))) int s_tab[2 * LT_TAPS * 4];      // per (row parity, tap): see the step loop
  const int64_t q64 = xcd_logical(blockIdx.x, a.xcd_per);
  if (q64 >= a.total) return;
  unsigned q = (unsigned)q64;
  const int txi = (int)(q % (unsigned)a.nxt); q /= (unsigned)a.nxt;
  const int tyi = (int)(q % (unsigned)a.nyt); q /= (unsigned)a.nyt;
  const int rx = (int)(q % (unsigned)a.g); q /= (unsigned)a.g;
  const int ry = (int)(q % (unsigned)a.g);
  const int b = (int)(q / (unsigned)a.g);
  const int ny = (a.H - ry + a.g - 1) / a.g, nx = (a.W - rx + a.g - 1) / a.g;      // lattice points of this class (<= 0: none)
  const int ly0 = tyi * LYT, lx0 = txi * LXT;
  const int rows_t = ny - ly0 < LYT ? ny - ly0 : LYT, cols_t = nx - lx0 < LXT ? nx - lx0 : LXT;
  if (rows_t <= 0 || cols_t <= 0) return;
  const int npos = rows_t * cols_t;

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wave_m = wave & 1, wave_n = wave >> 1;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, (int)((int64_t)a.B * a.H * a.W * a.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wpk), 0, LT_TAPS * a.CoutP * a.CinP * 2, 0x00020000);

  // staging map: thread -> (16-byte column piece cp = tid % 128 of a region row: point cp / 4, channel piece cp % 4; row parity
  // rp = tid / 128, wave-uniform); its piece i (0..7) is region row 2 i + rp.  Global offset and LDS address are AFFINE in i (one
  // VGPR each) and the row test is scalar.  Lanes past the region's 26 columns mirror a valid column, the sixteenth row mirrors
  // row 13: same data to the same address - no conditional store in the loop.
  const int cp = (tid & 127) < RP * 4 ? (tid & 127) : (tid & 127) - RP * 4, rp = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int scol = cp >> 2, slx = lx0 - 2 + scol;
  const bool col_ok = slx >= 0 && slx < nx;
  const unsigned aoff0 = col_ok ? (unsigned)((((b * a.H + ry + a.g * (ly0 - 2 + rp)) * a.W + rx + a.g * slx) * a.Cin + (cp & 3) * 8) * 2) : OOB;
  const unsigned astride = (unsigned)(2 * a.g * a.W * a.Cin * 2);            // two lattice rows
  const int alds = (rp * RP + scol) * ROWB + (cp & 3) * 16;                   // + i * 2 * RP * ROWB
  auto piece_row = [&](int i) { return 2 * i + rp < RY ? i : i - 1; };        // scalar
  auto piece_off = [&](int i) -> unsigned {
    const int ly = ly0 - 2 + 2 * i + rp;                                      // scalar
    return ((int)(ly >= 0) & (int)(ly < ny)) ? aoff0 + (unsigned)i * astride : OOB;
  };
  // A-fragment base of every row tile of this wave: output position o -> region point (row, col) of its CENTRE minus the halo origin;
  // a tap adds ((2 + ty) * RP + 2 + tx) * ROWB
  int abase[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int o = (wave_m * 8 + mi) * 16 + l15, oo = o < npos ? o : 0;
    const int row = oo / cols_t, col = oo - row * cols_t;
    abase[mi] = (row * RP + col) * ROWB + grp * 16;
  }
  const unsigned bvoff = (unsigned)((l15 * a.CinP + grp * 8) * 2);
  // per (staging row parity, canonical tap c) the four wave-uniform values a step needs, so that a step reads ONE 16-byte LDS word
  // (a step ahead) instead of deriving them on the SALU (65 scalar instructions per step in front of 32 MFMAs):
  //   x  LDS offset of the A operand of tap c             y  byte offset of the weight slab of tap c + 2 (requested in step c)
  //   z  LDS offset of the region row the piece requested three steps ago belongs to
  //   w  global offset of the region row this step requests (0x40000000: outside the image - the sum with a lane's offset stays
  //      out of range, the load returns zeros; lattice_eligible keeps the tensors below 2^30 bytes)
  if (tid < 2 * LT_TAPS) {
    const int trp = tid / LT_TAPS, tc = tid - trp * LT_TAPS;
    const int t2 = (tc + 2) % LT_TAPS;
    auto prow = [&](int i) { return 2 * i + trp < RY ? i : i - 1; };
    const int ist = prow((tc >= 3 ? tc - 3 : tc + LT_TAPS - 3) & 7), ild = prow(tc & 7);
    const int ly = ly0 - 2 + 2 * ild + trp;
    int* e = s_tab + tid * 4;
    e[0] = ((2 + tap_ty(tc)) * RP + 2 + tap_tx(tc)) * ROWB;
    e[1] = a.slab[t2] * a.CoutP * a.CinP * 2;
    e[2] = ist * 2 * RP * ROWB;
    e[3] = (ly >= 0 && ly < ny) ? ild * (int)astride : 0x40000000;
  }
  if (tid < LT_TAPS) s_wof[tid] = a.slab[tid] * a.CoutP * a.CinP * 2;

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = {0.f, 0.f, 0.f, 0.f};

  const int nchunks = a.Cin / 32;
  // prologue: chunk 0 into buffer 0
  {
    Vec<bf16_t> r[A_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) r[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, piece_off(piece_row(i)), 0, 0));
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) *reinterpret_cast<u32x4*>(smem + alds + piece_row(i) * 2 * RP * ROWB) = r[i].raw;
  }
  __syncthreads();

  // One STEP = one canonical tap of one 32-channel chunk: 32 MFMAs per wave.  Three register sets rotate through three unrolled steps:
  // the B fragments (weights) of step s + 2 are requested at the top of step s (an L2 round trip under load is longer than one
  // step), and every step moves ONE 16-byte piece of the next chunk's region: it stores the piece requested three steps earlier
  // (HBM latency: with one step of slack every step waited for its piece) and requests piece (tap % 8) - pieces repeat, same data
  // to the same address, so the loop has no branch around a load (the compiler then counts vmcnt exactly) and no conditional store.
  const int wbase = wave_n * 64 * a.CinP * 2, wni = 16 * a.CinP * 2;
  const int tabrow = rp * LT_TAPS;
  int c = 0, kc = 0, cur = 0;
  int pst_base = alds + REGION_BYTES;                     // LDS address of this thread's piece row 0 in the buffer being FILLED
  int soff_next = 64;                                     // channel-chunk byte offset of the region being fetched
  i32x4 tn;                                               // table entry of the current tap, read one step ahead
  auto load_b = [&](u32x4 (&dst)[4], int kc_, int wof) {
    // past the end (the step count is padded to whole trips of three): out of range - ZEROS, the padding steps add nothing
    const bool live = kc_ < nchunks;
    const int s0 = live ? wof + wbase + kc_ * 64 : 0;
    const unsigned v = bvoff + (live ? 0u : OOB);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) dst[ni] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, v, s0 + ni * wni, 0));
  };
  auto step = [&](const u32x4 (&fbc)[4], u32x4 (&fbn)[4], Vec<bf16_t>& pc) {
    const int imm = __builtin_amdgcn_readfirstlane(tn.x), wof2 = __builtin_amdgcn_readfirstlane(tn.y);
    const int sto = __builtin_amdgcn_readfirstlane(tn.z), lda = __builtin_amdgcn_readfirstlane(tn.w);
    const int c1 = c + 1 < LT_TAPS ? c + 1 : 0, kc2 = c + 2 < LT_TAPS ? kc : kc + 1;
    if constexpr (!(ABL & 1)) load_b(fbn, kc2, wof2);     // step s + 2
    tn = *reinterpret_cast<const i32x4*>(s_tab + (tabrow + c1) * 4);
    if constexpr (!(ABL & 4)) {
      *reinterpret_cast<u32x4*>(smem + pst_base + sto) = pc.raw;
      pc.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, aoff0 + (unsigned)lda, soff_next, 0));
    }
    // (left to itself the scheduler sinks the weight loads to the END of the step - right in front of their use)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fa[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(smem + abase[mi] + ((ABL & 2) ? 0 : imm));
    if constexpr (ABL & 8) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) asm volatile("" ::"v"(fa[mi]));
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) asm volatile("" ::"v"(fbc[ni]));
    } else {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi], __builtin_bit_cast(bf16x8, fbc[ni]), acc[mi][ni], 0, 0, 0);
    }
    if constexpr (!(ABL & 2)) {
      // A fragments THREE reads ahead of the MFMAs that consume them (the compiler's own order keeps one ahead: 4 MFMAs = 64 cycles
      // do not cover an LDS round trip)
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (++c == LT_TAPS) {                                 // chunk boundary (uniform): the next region is complete, everybody is done with this one
      c = 0; ++kc;
      soff_next += 64;
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) abase[mi] += REGION_BYTES - 2 * cur;
      pst_base -= REGION_BYTES - 2 * cur;
      cur = REGION_BYTES - cur;
    }
  };
  u32x4 fbA[4], fbB[4], fbC[4];
  Vec<bf16_t> pA, pB, pC;
  load_b(fbA, 0, __builtin_amdgcn_readfirstlane(s_wof[0]));
  load_b(fbB, 0, __builtin_amdgcn_readfirstlane(s_wof[1]));
  if constexpr (ABL & 1) load_b(fbC, 0, __builtin_amdgcn_readfirstlane(s_wof[2]));
  // the ring starts as if a chunk had run before, with pieces (14, 15, 16) % 8 of chunk 1 in flight
  tn = *reinterpret_cast<const i32x4*>(s_tab + tabrow * 4);
  {
    const int w6 = *(s_tab + (tabrow + 6) * 4 + 3), w7 = *(s_tab + (tabrow + 7) * 4 + 3), w0 = *(s_tab + (tabrow + 0) * 4 + 3);
    pA.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, aoff0 + (unsigned)w6, 64, 0));
    pB.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, aoff0 + (unsigned)w7, 64, 0));
    pC.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, aoff0 + (unsigned)w0, 64, 0));
  }
  const int nsteps = nchunks * LT_TAPS;
  for (int s = 0; s < nsteps; s += 3) {                   // no branch around a step: a load under any condition makes the vmcnt bookkeeping conservative
    step(fbA, fbC, pA);
    step(fbB, fbA, pB);
    step(fbC, fbB, pC);
  }
  __syncthreads();

  // ---- epilogue: bias, BatchNorm partial statistics, output tile through LDS ----------------------------------------------
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);           // [256][LDC]
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int col = wave_n * 64 + ni * 16 + l15;
    const float bv = (a.bias && col < a.Cout) ? a.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (wave_m * 8 + mi) * 16 + grp * 4 + r;
        const float v = acc[mi][ni][r] + bv;
        stf(Cs + o * LDC + col, v);
        if (o < npos) { s1 += v; s2 += v * v; }
      }
    if (a.stats) {
      s1 = rows_reduce<OpSum>(s1);
      s2 = rows_reduce<OpSum>(s2);
      if (grp == 0) { sstat[wave_m * 2 * BNT + col] = s1; sstat[(wave_m * 2 + 1) * BNT + col] = s2; }
    }
  }
  __syncthreads();
  if (a.stats) {
    float* slot = a.stats + (size_t)(blockIdx.x % RSSF_BN_SLOTS) * 2 * a.Cout;
    for (int i = tid; i < BNT; i += 256) {
      const float t1 = sstat[i] + sstat[2 * BNT + i], t2 = sstat[BNT + i] + sstat[3 * BNT + i];
      if (i < a.Cout) { atomicAdd(slot + i, t1); atomicAdd(slot + a.Cout + i, t2); }
    }
  }
  constexpr int OCPR = BNT / 8;                           // 16-byte chunks per output row: a thread's channel chunk is fixed (256 % 16 == 0)
  const bool bnb = a.bn_sums != nullptr;
  const int cc = (tid % OCPR) * 8;
  float bsc[8], bsh[8], t1[8], t2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool cok = bnb && cc + e < a.Cout;
    bsc[e] = cok ? a.bn_ss[cc + e] : 0.f;
    bsh[e] = cok ? a.bn_ss[a.Cout + cc + e] : 0.f;
    t1[e] = 0.f; t2[e] = 0.f;
  }
  for (int o = tid / OCPR; o < npos; o += 256 / OCPR) {
    const int prow = o / cols_t, pcol = o - prow * cols_t;
    const int64_t m = ((int64_t)b * a.H + ry + a.g * (ly0 + prow)) * a.W + rx + a.g * (lx0 + pcol);
    if (cc >= a.Cout) continue;
    bf16_t* dst = a.out + m * a.Cout + cc;
    Vec<bf16_t> v, xr, xp;
    if (bnb) {
      xr.load(a.bn_raw + m * a.Cout + cc);
      if (a.bn_res) xp.load(a.bn_res + m * a.Cout + cc);
    }
    v.load(Cs + o * LDC + cc);
    if (a.addend) {
      Vec<bf16_t> w;
      w.load(a.addend + m * a.Cout + cc);
      float s[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = v.get(e) + w.get(e);
      v.set_all(s);
    }
    v.store(dst);
    if (bnb) {                                             // on the values just stored: what a separate pass would read
      auto accumulate = [&](auto ACT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = xr.get(e);
          float z = fmaf(x, bsc[e], bsh[e]);
          if (a.bn_res) z += xp.get(e);
          const float gv = v.get(e);
          const float dz = decltype(ACT)::value == 1 ? (z > 0.f ? gv : 0.f) : decltype(ACT)::value == 2 ? gv * gelu_erf_grad(z) : gv;
          t1[e] += dz; t2[e] = fmaf(dz, x, t2[e]);
        }
      };
      if (a.bn_act == 1) accumulate(std::integral_constant<int, 1>{});
      else if (a.bn_act == 2) accumulate(std::integral_constant<int, 2>{});
      else accumulate(std::integral_constant<int, 0>{});
    }
  }
  if (bnb) {
    // a 16-lane row holds the 16 channel chunks once each: fold the four rows of a wave, one LDS row of partials per wave, summed
    // in a fixed order, one global atomic per channel and sum
    __syncthreads();
    float* sbn = reinterpret_cast<float*>(smem);            // [4 waves][2][BNT]
#pragma unroll
    for (int e = 0; e < 8; ++e) { t1[e] = rows_reduce<OpSum>(t1[e]); t2[e] = rows_reduce<OpSum>(t2[e]); }
    if (lane < OCPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sbn[wave * 2 * BNT + cc + e] = t1[e]; sbn[(wave * 2 + 1) * BNT + cc + e] = t2[e]; }
    }
    __syncthreads();
    float* slot = a.bn_sums + (size_t)(blockIdx.x % RSSF_BN_BWD_SLOTS) * 2 * a.Cout;
    for (int i = tid; i < BNT; i += 256) {
      float u1 = 0.f, u2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { u1 += sbn[w * 2 * BNT + i]; u2 += sbn[(w * 2 + 1) * BNT + i]; }
      if (i < a.Cout) { atomicAdd(slot + i, u1); atomicAdd(slot + a.Cout + i, u2); }
    }
  }
}

}  // namespace

// The lattice pitch g (>= 2) when the taps are exactly {0, +-g, +-2g}^2 restricted to the centre and the two rings of eight, in any
// order; 0 otherwise.  slab[c] receives the index of canonical tap c.
static int lattice_pitch(int ntaps, const int* dy, const int* dx, int* slab) {
  if (ntaps != LT_TAPS) return 0;
  int g = 0;
  for (int t = 0; t < ntaps; ++t) {
    const int m = abs(dy[t]) > abs(dx[t]) ? abs(dy[t]) : abs(dx[t]);
    if (m && (!g || m < g)) g = m;
  }
  if (g < 2) return 0;
  for (int c = 0; c < LT_TAPS; ++c) {
    slab[c] = -1;
    for (int t = 0; t < ntaps; ++t)
      if (dy[t] == g * tap_ty(c) && dx[t] == g * tap_tx(c)) slab[c] = t;
    if (slab[c] < 0) return 0;
  }
  return g;
}

bool lattice_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx) {
  const char* sw = getenv("RSSF_LATTICE");                  // A/B switch, read per call (tests hold the two kernels against each other)
  const bool enabled = sw && sw[0] == '1';                  // off by default while it only ties the gather kernel at B = 16
  int slab[LT_TAPS];
  return enabled && mul == 1 && div == 1 && IH == OH && IW == OW && Cout == BNT && Cin >= 32 && (Cin % 32) == 0 &&
         (int64_t)B * IH * IW * (Cin > Cout ? Cin : Cout) < ((int64_t)1 << 29) && lattice_pitch(ntaps, dy, dx, slab) > 0;
}

int launch_lattice(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend, const void* bn_raw,
                   const void* bn_res, const float* bn_ss, float* bn_sums, int bn_act, int B, int H, int W, int Cin, int Cout, int CinP,
                   int CoutP, int ntaps, const int* dy, const int* dx, hipStream_t st) {
  LatticeArgs a;
  a.in = (const bf16_t*)in; a.wpk = (const bf16_t*)wpk; a.out = (bf16_t*)out; a.bias = bias; a.stats = stats; a.addend = (const bf16_t*)addend;
  a.bn_raw = (const bf16_t*)bn_raw; a.bn_res = (const bf16_t*)bn_res; a.bn_ss = bn_ss; a.bn_sums = bn_sums; a.bn_act = bn_act;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CinP = CinP; a.CoutP = CoutP;
  a.g = lattice_pitch(ntaps, dy, dx, a.slab);
  if (a.g < 2) { set_error("conv_lattice: the taps are not a two-ring lattice"); return RSSF_ERR_UNSUPPORTED; }
  a.nyt = ((H + a.g - 1) / a.g + LYT - 1) / LYT;
  a.nxt = ((W + a.g - 1) / a.g + LXT - 1) / LXT;
  const int64_t total = (int64_t)B * a.g * a.g * a.nyt * a.nxt;
  if (total >= ((int64_t)1 << 28)) { set_error("conv_lattice: %lld tiles", (long long)total); return RSSF_ERR_UNSUPPORTED; }
  a.total = (int)total;
  a.xcd_per = xcd_per(total);
  static std::once_flag once;
  static hipError_t e = hipSuccess;
  static int abl = 0;
  std::call_once(once, [] {
    abl = getenv("RSSF_LATTICE_ABL") ? atoi(getenv("RSSF_LATTICE_ABL")) : 0;
    for (const void* f : {(const void*)conv_lattice_kernel<0>, (const void*)conv_lattice_kernel<1>, (const void*)conv_lattice_kernel<2>,
                          (const void*)conv_lattice_kernel<4>, (const void*)conv_lattice_kernel<8>, (const void*)conv_lattice_kernel<7>})
      if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (getenv("RSSF_LATTICE_DEBUG")) {
      int per_cu = -1;
      hipFuncAttributes fa;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_lattice_kernel<0>, 256, LDS_BYTES);
      (void)hipFuncGetAttributes(&fa, (const void*)conv_lattice_kernel<0>);
      fprintf(stderr, "[rssf] conv_lattice: %d workgroups per CU (occupancy API), %d registers, %zu B static + %zu B dynamic LDS\n", per_cu, fa.numRegs,
              fa.sharedSizeBytes, (size_t)LDS_BYTES);
    }
  });
  if (e != hipSuccess) { set_error("conv_lattice: cannot raise the LDS limit: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  const dim3 grid((unsigned)a.xcd_per * 8);
  switch (abl) {
    case 1: conv_lattice_kernel<1><<<grid, 256, LDS_BYTES, st>>>(a); break;
    case 2: conv_lattice_kernel<2><<<grid, 256, LDS_BYTES, st>>>(a); break;
    case 4: conv_lattice_kernel<4><<<grid, 256, LDS_BYTES, st>>>(a); break;
    case 8: conv_lattice_kernel<8><<<grid, 256, LDS_BYTES, st>>>(a); break;
    case 7: conv_lattice_kernel<7><<<grid, 256, LDS_BYTES, st>>>(a); break;
    default: conv_lattice_kernel<0><<<grid, 256, LDS_BYTES, st>>>(a);
  }
  return check_launch("conv_lattice");
}

}  // namespace cv
}  // namespace rssf
