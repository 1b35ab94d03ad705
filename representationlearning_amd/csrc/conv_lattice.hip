// Dilated multi-tap convolution on its pixel LATTICE (gfx950 MFMA, bf16): MlpDWBN's fused {1x1 + 3x3 dil 6 + 3x3 dil 12} sum
// (ffn_block.py:226-228, 250-257), forward and data gradient.
//
// Every tap offset of that sum is a multiple of g = 6 pixels, so the pixels with (y mod g, x mod g) = (ry, rx) form a closed
// sub-problem: on that lattice (22 x 22 points of a 128 x 128 map) the 17 distinct taps are the centre, the eight neighbours at
// distance 1 and the eight at distance 2 - a dense 5 x 5-footprint convolution.  The generic gather kernel (conv_fwd.hip) fetches and
// stages every activation tile once PER TAP (17 global gathers and 17 LDS images of each pixel; 0.34 of the MFMA peak, bound by its
// LDS traffic and the staging itself).  Here a workgroup of EIGHT waves (4 along the positions x 2 along the channels, 64 x 64 wave
// tiles) owns a whole class of one image (<= 22 x 22 points = 31 MFMA row tiles): per 32-channel chunk the haloed lattice region
// (26 x 26 points, 96-byte rows: conflict-free ds_read_b128) is staged ONCE and all 17 taps run out of LDS - a tap is an offset of
// the A-fragment reads; the next chunk's region arrives piece by piece under the tap loop in a second buffer.  The weight slab of a
// step ([128][32], 8 KB) is fetched once per workgroup - one 16-byte piece per thread - into a double-buffered LDS tile: 0.32 GB of
// weight traffic per launch at B = 16 (a first version with four waves per half class and the weights as register fragments
// straight from the L2 moved 1.1 GB and was bound by exactly that stream: without its MFMAs it took 145 of its 158 us).
// LDS: two regions + two weight tiles = 151 KB, one workgroup per CU (two waves per SIMD).  Per wave and step: 12 ds_read_b128,
// 2 ds_write_b128, 2 buffer loads, 32 MFMAs, one workgroup barrier.
//
// Tail: 16 images x 36 classes = 576 workgroups are 2.25 rounds of 256 CUs; the remainder past the last full round runs as
// QUARTER workgroups (32 of the 128 output channels each, one row tile of weights per wave) so that the last round is a quarter long.
//
// Epilogue as in the gather kernel: + bias, fused BatchNorm statistics (forward) or BatchNorm-backward statistics of the producer
// (data gradient, rssf_conv_gather_bnbwd), output tile through LDS for 16-byte stores.
#include <mutex>
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace rssf {
namespace cv {

namespace {

constexpr int LT_TAPS = 17;
constexpr int LYT = 22, LXT = 22;                       // lattice points of one tile (rows x columns)
constexpr int RY = LYT + 4, RP = LXT + 4;               // haloed region: two lattice points on every side
constexpr int ROWB = 96;                                // bytes per staged row (32 channels + 32 B pad)
constexpr int REGION = RY * RP * ROWB;
constexpr int BNT = 128;                                // output channels of a full workgroup
constexpr int BTILE = BNT * ROWB;                       // one weight tile: 128 rows of 32 channels
constexpr size_t LDS_BYTES = 2 * (size_t)REGION + 2 * (size_t)BTILE;
constexpr int A_PIECES = 8;                             // per thread and chunk: region row 4 i + (tid / 128)
static_assert(LDS_BYTES >= (size_t)512 * (BNT + 8) * 2 + 8 * 2 * BNT * 4, "the output tile and the statistics partials reuse the staging buffers");
static_assert(LYT * LXT <= 512 && RY <= 4 * A_PIECES && RP * 4 <= 128, "a class fits the block tile; the staging map covers the region");

// canonical tap order: centre, ring 1, ring 2 (in lattice units)
__host__ __device__ constexpr int tap_ty(int c) { return c == 0 ? 0 : (((c - 1) % 8 < 3) ? -1 : ((c - 1) % 8 < 5) ? 0 : 1) * (c > 8 ? 2 : 1); }
__host__ __device__ constexpr int tap_tx(int c) {
  const int k = (c - 1) % 8;
  return c == 0 ? 0 : ((k == 0 || k == 3 || k == 5) ? -1 : (k == 1 || k == 6) ? 0 : 1) * (c > 8 ? 2 : 1);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct LatticeArgs {
  const bf16_t* in; const bf16_t* wpk; bf16_t* out; const float* bias; float* stats; const bf16_t* addend;
  const bf16_t* bn_raw; const bf16_t* bn_res; const float* bn_ss; float* bn_sums; int bn_act;
  int B, H, W, Cin, Cout, CinP, CoutP;
  int g, nyt, nxt, xcd_per, total;
  int nfull;                                            // tiles [0, nfull) run as whole workgroups, the others as four quarter workgroups
  int slab[LT_TAPS];                                    // packed-weight slab of canonical tap c
};

// One workgroup: lattice tile q, output channels [n0, n0 + 32 NI).  NI = 16-channel tiles per wave (4: whole, 1: quarter).
template <int NI>
__device__ __forceinline__ void lattice_block(const LatticeArgs& a, unsigned q, const int n0, char* smem, int* s_wof, int* s_tab) {
  constexpr int BN = 32 * NI, LDC = BN + 8, OCPR = BN / 8;
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wave_m = wave & 3, wave_n = wave >> 2;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, (int)((int64_t)a.B * a.H * a.W * a.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wpk), 0, LT_TAPS * a.CoutP * a.CinP * 2, 0x00020000);
  char* const Abuf = smem;                                 // two regions
  char* const Bbuf = smem + 2 * REGION;                    // two weight tiles

  // region staging map: thread -> (16-byte column piece cp = tid % 128 of a region row: point cp / 4, channel piece cp % 4; row group
  // rq = tid / 128, wave-uniform); its piece i is region row 4 i + rq.  Global offset and LDS address are AFFINE in i (one VGPR each)
  // and the row test is scalar.  Lanes past the region's 26 columns mirror a valid column, rows past the region mirror the group's
  // last row: same data to the same address - no conditional store in the loop.
  const int cp = (tid & 127) < RP * 4 ? (tid & 127) : (tid & 127) - RP * 4, rq = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int scol = cp >> 2, slx = lx0 - 2 + scol;
  const bool col_ok = slx >= 0 && slx < nx;
  const unsigned aoff0 = col_ok ? (unsigned)((((b * a.H + ry + a.g * (ly0 - 2 + rq)) * a.W + rx + a.g * slx) * a.Cin + (cp & 3) * 8) * 2) : OOB;
  const unsigned astride = (unsigned)(4 * a.g * a.W * a.Cin * 2);            // four lattice rows
  const int alds = (rq * RP + scol) * ROWB + (cp & 3) * 16;                   // + i * 4 * RP * ROWB
  // weight staging map: thread -> row (tid / 4) % BN (output channel n0 + row), 16-byte piece tid % 4 of the step's [BN][32] slab
  // (a quarter workgroup has four threads per piece: same data to the same address)
  const int brow = (tid >> 2) & (BN - 1);
  const unsigned bvoff = (unsigned)((((n0 + brow) * a.CinP) + (tid & 3) * 8) * 2);
  const int blds = brow * ROWB + (tid & 3) * 16;
  // A-fragment base of every row tile of this wave: output position o -> region point (row, col) of its CENTRE minus the halo origin;
  // a tap adds ((2 + ty) * RP + 2 + tx) * ROWB
  int abase[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int o = (wave_m * 8 + mi) * 16 + l15, oo = o < npos ? o : 0;
    const int row = oo / cols_t, col = oo - row * cols_t;
    abase[mi] = (row * RP + col) * ROWB + grp * 16;
  }
  int bfrag = 2 * REGION + (wave_n * 16 * NI + l15) * ROWB + grp * 16;        // + ni * 16 * ROWB; toggles between the two tiles
  // per (row group, canonical tap c) the four wave-uniform values a step needs, so that a step reads ONE 16-byte LDS word (a step
  // ahead) instead of deriving them on the SALU (65 scalar instructions per step in front of 32 MFMAs at first):
  //   x  LDS offset of tap c's A operand                  y  byte offset of the weight slab of tap c + 2 (requested in step c)
  //   z  LDS offset of the region row stored in step c (requested in the step before)
  //   w  global offset of the region row requested in step c (0x40000000: outside the image - the sum with a lane's offset stays
  //      out of range, the load returns zeros; lattice_eligible keeps the tensors below 2^30 bytes)
  if (tid < 4 * LT_TAPS) {
    const int trq = tid / LT_TAPS, tc = tid - trq * LT_TAPS;
    auto prow = [&](int i) { const int im = (RY - 1 - trq) / 4; return i < im ? i : im; };
    const int ist = prow((tc >= 1 ? tc - 1 : LT_TAPS - 1) & 7), ild = prow(tc & 7);
    const int ly = ly0 - 2 + 4 * ild + trq;
    int* e = s_tab + tid * 4;
    e[0] = ((2 + tap_ty(tc)) * RP + 2 + tap_tx(tc)) * ROWB;
    e[1] = a.slab[(tc + 2) % LT_TAPS] * a.CoutP * a.CinP * 2;
    e[2] = ist * 4 * RP * ROWB;
    e[3] = (ly >= 0 && ly < ny) ? ild * (int)astride : 0x40000000;
  }
  if (tid < LT_TAPS) s_wof[tid] = a.slab[tid] * a.CoutP * a.CinP * 2;

  f32x4 acc[8][NI];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = {0.f, 0.f, 0.f, 0.f};

  const int nchunks = a.Cin / 32;
  __syncthreads();                                         // tables
  const int tabrow = rq * LT_TAPS;
  // prologue: region of chunk 0 into buffer 0, weight tile of step 0 into tile 0
  {
    Vec<bf16_t> r[A_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i)
      r[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, aoff0 + (unsigned)s_tab[(tabrow + i) * 4 + 3], 0, 0));
    const u32x4 w0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, bvoff, __builtin_amdgcn_readfirstlane(s_wof[0]), 0));
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int im = (RY - 1 - rq) / 4, ii = i < im ? i : im;
      *reinterpret_cast<u32x4*>(Abuf + alds + ii * 4 * RP * ROWB) = r[i].raw;
    }
    *reinterpret_cast<u32x4*>(Bbuf + blds) = w0;
  }

  // Step s = (chunk, canonical tap): store the weight piece of step s + 1 into the OTHER tile and the region piece into the OTHER
  // region (both requested in the step before: a step of 32 MFMAs per wave, two waves per SIMD, is longer than an HBM round trip);
  // request the weight piece of step s + 2 and a region piece; 8 + NI fragment reads, 8 NI MFMAs; barrier (the tile written in this
  // step is read in the next; the tile read in this step is written in the next).  Requests repeat cyclically (region pieces:
  // tap % 8, same data to the same address): no branch around a load - the compiler then counts vmcnt exactly - no conditional store.
  int c = 0, kc = 0, cur = 0;
  int pst_base = alds + REGION, bst = blds + 2 * REGION + BTILE;      // where this step's pieces go
  int soff_next = 64;
  i32x4 tn;
  auto load_w = [&](int kc_, int wof) -> u32x4 {            // past the end: zeros, never used
    const bool live = kc_ < nchunks;
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, bvoff + (live ? 0u : OOB), live ? wof + kc_ * 64 : 0, 0));
  };
  u32x4 wq = load_w(0, __builtin_amdgcn_readfirstlane(s_wof[1]));
  Vec<bf16_t> pc;
  pc.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, aoff0 + (unsigned)s_tab[(tabrow + 0) * 4 + 3], 64, 0));
  tn = *reinterpret_cast<const i32x4*>(s_tab + tabrow * 4);
  __syncthreads();                                         // region 0 and weight tile 0 are complete
  const int nsteps = nchunks * LT_TAPS;
  for (int s = 0; s < nsteps; ++s) {
    const int imm = __builtin_amdgcn_readfirstlane(tn.x), wof2 = __builtin_amdgcn_readfirstlane(tn.y);
    const int sto = __builtin_amdgcn_readfirstlane(tn.z), lda = __builtin_amdgcn_readfirstlane(tn.w);
    const int c1 = c + 1 < LT_TAPS ? c + 1 : 0, kc2 = c + 2 < LT_TAPS ? kc : kc + 1;
    *reinterpret_cast<u32x4*>(smem + bst) = wq;                                 // weights of step s + 1
    *reinterpret_cast<u32x4*>(smem + pst_base + sto) = pc.raw;
    wq = load_w(kc2, wof2);                                                     // weights of step s + 2
    pc.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, aoff0 + (unsigned)lda, soff_next, 0));
    tn = *reinterpret_cast<const i32x4*>(s_tab + (tabrow + c1) * 4);
    // (left to itself the scheduler sinks the requests to the END of the step - right in front of their use)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fa[8], fb[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const bf16x8*>(smem + bfrag + ni * 16 * ROWB);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(smem + abase[mi] + imm);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    if constexpr (NI == 4) {
      // fragments TWO reads ahead of the MFMAs that consume them (the compiler's own order keeps one ahead: 4 MFMAs = 64 cycles do
      // not cover an LDS round trip)
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the weight tiles swap every step
    bfrag += (bfrag >= 2 * REGION + BTILE) ? -BTILE : BTILE;
    bst += (bst >= 2 * REGION + BTILE) ? -BTILE : BTILE;
    if (++c == LT_TAPS) {                                   // chunk boundary: the regions swap
      c = 0; ++kc;
      soff_next += 64;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) abase[mi] += REGION - 2 * cur;
      pst_base -= REGION - 2 * cur;
      cur = REGION - cur;
    }
    __syncthreads();
  }

  // ---- epilogue: bias, BatchNorm partial statistics, output tile through LDS ----------------------------------------------------
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);           // [512][LDC]
  float* sstat = reinterpret_cast<float*>(smem + (size_t)512 * LDC * 2);      // [4 wave rows][2][BN], later [8 waves][2][BN]
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int lcol = wave_n * 16 * NI + ni * 16 + l15, col = n0 + lcol;
    const float bv = (a.bias && col < a.Cout) ? a.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (wave_m * 8 + mi) * 16 + grp * 4 + r;
        const float v = acc[mi][ni][r] + bv;
        stf(Cs + o * LDC + lcol, v);
        if (o < npos) { s1 += v; s2 += v * v; }
      }
    if (a.stats) {
      s1 = rows_reduce<OpSum>(s1);                         // over the four 16-lane groups: lane swaps, no LDS round trip
      s2 = rows_reduce<OpSum>(s2);
      if (grp == 0) { sstat[wave_m * 2 * BN + lcol] = s1; sstat[(wave_m * 2 + 1) * BN + lcol] = s2; }
    }
  }
  __syncthreads();
  if (a.stats) {
    float* slot = a.stats + (size_t)(blockIdx.x % RSSF_BN_SLOTS) * 2 * a.Cout;
    for (int i = tid; i < BN; i += 512) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += sstat[w * 2 * BN + i]; t2 += sstat[(w * 2 + 1) * BN + i]; }
      if (n0 + i < a.Cout) { atomicAdd(slot + n0 + i, t1); atomicAdd(slot + a.Cout + n0 + i, t2); }
    }
  }
  // 16-byte rows; a thread's channel chunk is the same in every pass (512 % OCPR == 0), so the 2 x 8 partial sums of the fused
  // BatchNorm-backward statistics stay in registers
  const bool bnb = a.bn_sums != nullptr;
  const int ccl = (tid % OCPR) * 8, cc = n0 + ccl;
  float bsc[8], bsh[8], t1[8], t2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool cok = bnb && cc + e < a.Cout;
    bsc[e] = cok ? a.bn_ss[cc + e] : 0.f;
    bsh[e] = cok ? a.bn_ss[a.Cout + cc + e] : 0.f;
    t1[e] = 0.f; t2[e] = 0.f;
  }
  for (int o = tid / OCPR; o < npos; o += 512 / OCPR) {
    const int prow = o / cols_t, pcol = o - prow * cols_t;
    const int64_t m = ((int64_t)b * a.H + ry + a.g * (ly0 + prow)) * a.W + rx + a.g * (lx0 + pcol);
    if (cc >= a.Cout) continue;
    bf16_t* dst = a.out + m * a.Cout + cc;
    Vec<bf16_t> v, xr, xp;
    if (bnb) {
      xr.load(a.bn_raw + m * a.Cout + cc);
      if (a.bn_res) xp.load(a.bn_res + m * a.Cout + cc);
    }
    v.load(Cs + o * LDC + ccl);
    if (a.addend) {
      Vec<bf16_t> w;
      w.load(a.addend + m * a.Cout + cc);
      float sm[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sm[e] = v.get(e) + w.get(e);
      v.set_all(sm);
    }
    v.store(dst);
    if (bnb) {                                             // on the values just stored: what a separate pass would read
      auto accumulate = [&](auto ACT) {                    // block-uniform activation: one specialised loop runs
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
    // lanes with the same channel chunk: every OCPR-th lane of a 16-lane row (rotations inside the row), then the four rows; one
    // LDS row of partials per wave, summed in a fixed order, one global atomic per channel and sum
    __syncthreads();
    float* sbn = sstat;                                     // [8 waves][2][BN]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (OCPR <= 4) { t1[e] += dpp_mov<0x124>(t1[e]); t2[e] += dpp_mov<0x124>(t2[e]); }      // row_ror:4
      if (OCPR <= 8) { t1[e] += dpp_mov<0x128>(t1[e]); t2[e] += dpp_mov<0x128>(t2[e]); }      // row_ror:8
      t1[e] = rows_reduce<OpSum>(t1[e]); t2[e] = rows_reduce<OpSum>(t2[e]);
    }
    if (lane < OCPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sbn[wave * 2 * BN + ccl + e] = t1[e]; sbn[(wave * 2 + 1) * BN + ccl + e] = t2[e]; }
    }
    __syncthreads();
    float* slot = a.bn_sums + (size_t)(blockIdx.x % RSSF_BN_BWD_SLOTS) * 2 * a.Cout;
    for (int i = tid; i < BN; i += 512) {
      float u1 = 0.f, u2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { u1 += sbn[w * 2 * BN + i]; u2 += sbn[(w * 2 + 1) * BN + i]; }
      if (n0 + i < a.Cout) { atomicAdd(slot + n0 + i, u1); atomicAdd(slot + a.Cout + n0 + i, u2); }
    }
  }
}

// grid: 8 * xcd_per blocks map XCD-major onto the whole tiles [0, nfull) (tiles of one image share lines of the input only through
// the channel chunks, but they do share the XCD's copy of the weights); then four quarter workgroups per remaining tile
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_lattice_kernel(LatticeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_wof[LT_TAPS];
  __shared__ __attribute__((aligned(16))) int s_tab[4 * LT_TAPS * 4];
  const unsigned nmain = (unsigned)a.xcd_per * 8u;
  if (blockIdx.x < nmain) {
    const int64_t q = xcd_logical(blockIdx.x, a.xcd_per);
    if (q >= a.nfull) return;
    lattice_block<4>(a, (unsigned)q, 0, smem, s_wof, s_tab);
  } else {
    const unsigned t = blockIdx.x - nmain;
    lattice_block<1>(a, (unsigned)a.nfull + (t >> 2), (int)(t & 3u) * 32, smem, s_wof, s_tab);
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
  // OPT-IN (RSSF_LATTICE=1): measured in the training step it does not beat the gather kernel, whose operands are cache-resident
  // there (forward 178 against 169 us, data gradient 215 against 205 us; stand-alone on cold operands 167 against 193 us) - DESIGN.md
  const bool enabled = sw && sw[0] == '1';
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
  // one workgroup per CU: what is left past the last full round of the chip runs as quarter workgroups when that shortens the tail
  // (a quarter workgroup takes ~0.4 of a whole one: up to 2 rounds of quarters beat one round of whole workgroups)
  static const int cus = [] {
    int dev = 0, v = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  const char* tsw = getenv("RSSF_LATTICE_TAIL");              // tuning: 0 = never split the tail
  const char* csw = getenv("RSSF_LATTICE_CUS");               // tests: pretend a smaller chip, so that small maps reach the quarter path
  const int ncu = csw && atoi(csw) > 0 ? atoi(csw) : cus;
  const int rem = a.total % ncu;
  a.nfull = (a.total > ncu && rem > 0 && rem * 4 <= 2 * ncu && !(tsw && tsw[0] == '0')) ? a.total - rem : a.total;
  a.xcd_per = xcd_per(a.nfull);
  static std::once_flag once;
  static hipError_t e = hipSuccess;
  std::call_once(once, [] {
    e = hipFuncSetAttribute((const void*)conv_lattice_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (getenv("RSSF_LATTICE_DEBUG")) {
      int per_cu = -1;
      hipFuncAttributes fa;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_lattice_kernel, 512, LDS_BYTES);
      (void)hipFuncGetAttributes(&fa, (const void*)conv_lattice_kernel);
      fprintf(stderr, "[rssf] conv_lattice: %d workgroups per CU (occupancy API), %d registers, %zu B static + %zu B dynamic LDS\n", per_cu, fa.numRegs,
              fa.sharedSizeBytes, (size_t)LDS_BYTES);
    }
  });
  if (e != hipSuccess) { set_error("conv_lattice: cannot raise the LDS limit: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  const unsigned grid = (unsigned)a.xcd_per * 8u + (unsigned)(a.total - a.nfull) * 4u;
  conv_lattice_kernel<<<dim3(grid), 512, LDS_BYTES, st>>>(a);
  return check_launch("conv_lattice");
}

}  // namespace cv
}  // namespace rssf
