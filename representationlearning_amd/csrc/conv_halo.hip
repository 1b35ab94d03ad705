// 3x3 / stride-1 / "same" convolution on channels-last bf16 activations, halo-tiled (gfx950 MFMA).
//
// The HRNet BasicBlock / Bottleneck / fuse 3x3 convolutions (_hrnet_rssformer.py:216-287) are small-channel
// (32..256) and at B=16 all cost the same 4.8 GFLOP; in the generic gather kernel (conv_fwd.hip) they are bound by
// the per-tap global->LDS staging (every input pixel is fetched nine times, two barriers per four MFMAs).  Here a
// block owns a TH x 16 pixel tile: per 32-channel chunk it stages the (TH+2) x 18 input halo ONCE plus the nine
// [BN][32] weight slabs, and then runs all nine taps out of LDS with shifted A-operand addresses - 9 x MI x NI
// MFMAs per barrier pair, the next chunk's global loads in flight underneath.  The data gradient is the same kernel
// with mirrored taps and transposed weight slabs (the host passes those, exactly as for rssf_conv_gather).
//
// Epilogue identical to the gather kernel: + bias, fused BatchNorm statistics (slotted atomics), LDS transpose
// for 16-byte coalesced stores.
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace rssf {
namespace cv {

namespace {

#ifndef RSSF_HALO_LDK_PAD
#define RSSF_HALO_LDK_PAD 8
#endif
constexpr int TW = 16, KC = 32, LDK = KC + RSSF_HALO_LDK_PAD;      // 80-byte LDS rows (2-way conflicts on the non-contiguous 16-lane groups of ds_read_b128; 96-byte rows are conflict-free but cost a resident block per CU: measured equal, round 4)

// MIRROR: the nine taps in data-gradient order (offset of tap t = -(t/3 - 1, t%3 - 1)) instead of forward order; the offsets
// are compile-time, so a tap only changes the IMMEDIATE offset of the A-operand LDS reads (with run-time dy/dx every tap cost
// two scalar loads and ~6 VALU of address arithmetic per fragment in an issue-bound kernel).
// One block's work: logical tile q of problem a.  A __device__ function so that the single-problem kernel and the grouped kernel
// (several independent problems in ONE launch, conv3x3_halo_group_kernel below) share it.
template <int TH, int BN, bool MIRROR, bool PRE>
__device__ __forceinline__ void halo_block(const HaloArgs& a, const unsigned q) {
  constexpr int MI = TH / 4, NI = BN / 16, HP = (TH + 2) * (TW + 2), BMP = TH * TW;
  constexpr int A_ELEMS = HP * LDK, B_ELEMS = 9 * BN * LDK, LDC = BN + 8;
  constexpr int A_LOADS = (HP * 4 + 255) / 256, B_LOADS = (9 * BN * 4 + 255) / 256;
  constexpr int LDS_ELEMS = (A_ELEMS + B_ELEMS) > BMP * LDC ? (A_ELEMS + B_ELEMS) : BMP * LDC;
  __shared__ __attribute__((aligned(16))) bf16_t lds[LDS_ELEMS];
  __shared__ float sstat[4 * 2 * BN];                    // one row of partials per wave: summed in a fixed order, no LDS atomics
  bf16_t* As = lds;
  bf16_t* Bs = lds + A_ELEMS;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  // 32-bit unsigned tile arithmetic (launch_halo keeps the block count below 2^31): the 64-bit scalar divisions were a serial
  // ~150-instruction chain in front of the first global load of every block
  const unsigned ntn = (unsigned)a.ntiles_n, ntx = (unsigned)a.tiles_x, nty = (unsigned)a.tiles_y;
  unsigned t = q / ntn;
  const int tile_id = (int)t;
  const int n0 = (int)(q - t * ntn) * BN;
  const unsigned t1 = t / ntx;
  const int tx = (int)(t - t1 * ntx);
  const int b = (int)(t1 / nty);
  const int ty = (int)(t1 - (unsigned)b * nty);
  const int y0 = ty * TH, x0 = tx * TW;

  // fixed per-thread staging slots.  All loads are hardware-bounds-checked buffer loads with 32-bit byte offsets: an
  // out-of-image halo pixel or an out-of-range channel chunk gets the out-of-range sentinel and the load returns zeros (no
  // 64-bit address arithmetic, no clamps, no zeroing at the LDS store).  The kernel is instruction-issue-bound (4 waves per
  // SIMD, each issuing 35 % of its cycles: 727 VALU + 486 SALU around 36 MFMAs before this change), so instructions are time.
  constexpr unsigned OOB = 0x80000000u;                   // >= num_records: the host keeps both tensors below 2^31 bytes
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0,
                                                                         (int)((int64_t)a.B * a.H * a.W * a.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wpk), 0, 9 * a.CoutP * a.CinP * 2, 0x00020000);
  unsigned aoff[A_LOADS], boff[B_LOADS];
  int asub[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int idx = tid + i * 256, p = idx >> 2;
    const int py = p / (TW + 2), gy = y0 - 1 + py, gx = x0 - 1 + (p - py * (TW + 2));
    const bool ok = idx < HP * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    asub[i] = (idx & 3) * 8;
    aoff[i] = ok ? (unsigned)((((b * a.H + gy) * a.W + gx) * a.Cin + asub[i]) * 2) : OOB;
  }
  {
    // row = tap*BN + n; chunk i of a thread is 64 rows = 64/BN taps further on, same n: one multiply for all of them
    static_assert(64 % BN == 0, "a thread's weight chunks must keep their column");
    const int row0 = tid >> 2;
    const unsigned b0 = (unsigned)((((row0 / BN) * a.CoutP + (row0 % BN)) * a.CinP + (tid & 3) * 8) * 2);
    const unsigned bstep = (unsigned)((64 / BN) * a.CoutP * a.CinP * 2);
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) boff[i] = ((9 * BN * 4) % 256 == 0 || tid + i * 256 < 9 * BN * 4) ? b0 + i * bstep : OOB;
  }
  Vec<bf16_t> ra[A_LOADS], rb[B_LOADS];
  // PRE: finalize the producer's BatchNorm (see HaloArgs::pre_*): scale / shift of all Cin channels into LDS, under the first
  // chunk's loads below; the constants of a thread's 8 channels of a chunk (its channel group inside a chunk is fixed: tid & 3)
  // are read back when the chunk is staged
  constexpr int PRE_MAXC = 256;
  __shared__ float spre[PRE ? 2 * PRE_MAXC : 1];
  auto finalize_producer = [&]() {
    for (int c = tid; c < a.Cin; c += 256) {
      float mean, var;
      if (a.pre_training) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < RSSF_BN_SLOTS; ++k) { s1 += a.pre_stats[(size_t)k * 2 * a.Cin + c]; s2 += a.pre_stats[(size_t)k * 2 * a.Cin + a.Cin + c]; }
        mean = s1 / a.pre_n;
        var = fmaxf(s2 / a.pre_n - mean * mean, 0.f);
      } else {
        mean = a.pre_rmean[c];
        var = a.pre_rvar[c];
      }
      const float invstd = rsqrtf(var + a.pre_eps);
      const float sc = a.pre_gamma[c] * invstd, sh = a.pre_beta[c] - mean * sc;
      spre[c] = sc; spre[PRE_MAXC + c] = sh;
      if (q == 0) {                                                          // one block publishes for the backward pass
        a.pre_mi[c] = mean; a.pre_mi[a.Cin + c] = invstd;
        a.pre_ss[c] = sc; a.pre_ss[a.Cin + c] = sh;
        if (a.pre_training && a.pre_rmean) {
          a.pre_rmean[c] = (1.f - a.pre_momentum) * a.pre_rmean[c] + a.pre_momentum * mean;
          a.pre_rvar[c] = (1.f - a.pre_momentum) * a.pre_rvar[c] + a.pre_momentum * var * (a.pre_n > 1.f ? a.pre_n / (a.pre_n - 1.f) : 1.f);
        }
      }
    }
  };
  auto load_chunk = [&](int kc) {
    const int crem = a.Cin - kc * KC;                                        // channels left in this chunk (scalar)
    const int soff = kc * KC * 2, swoff = (n0 * a.CinP + kc * KC) * 2;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      ra[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, asub[i] < crem ? aoff[i] : OOB, soff, 0));
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      rb[i].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, boff[i], swoff, 0));
  };
  auto store_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int idx = tid + i * 256;
      if constexpr (PRE) {
        // act(raw * scale + shift) of the producer, zero for halo pixels outside the image / channels past Cin (the padding of the
        // ACTIVATION is zero, not act(shift)); block-uniform activation: one specialised loop runs
        const bool valid = aoff[i] != OOB && asub[i] < a.Cin - kc * KC;
        const int c0 = valid ? kc * KC + asub[i] : 0;
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(spre + c0), sc1 = *reinterpret_cast<const f32x4*>(spre + c0 + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(spre + PRE_MAXC + c0), sh1 = *reinterpret_cast<const f32x4*>(spre + PRE_MAXC + c0 + 4);
        auto apply = [&](auto ACT) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z = fmaf(ra[i].get(e), e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
            o[e] = decltype(ACT)::value == 1 ? fmaxf(z, 0.f) : decltype(ACT)::value == 2 ? gelu_erf(z) : z;
          }
          ra[i].set_all(o);
        };
        if (a.pre_act == 1) apply(std::integral_constant<int, 1>{});
        else if (a.pre_act == 2) apply(std::integral_constant<int, 2>{});
        else apply(std::integral_constant<int, 0>{});
        if (!valid) ra[i].clear();
      }
      if ((HP * 4) % 256 == 0 || idx < HP * 4) ra[i].store(As + (idx >> 2) * LDK + (idx & 3) * 8);
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      const int idx = tid + i * 256;
      if ((9 * BN * 4) % 256 == 0 || idx < 9 * BN * 4) rb[i].store(Bs + (idx >> 2) * LDK + (idx & 3) * 8);
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = {0.f, 0.f, 0.f, 0.f};

  const int nchunks = a.CinP / KC;
  load_chunk(0);
  if constexpr (PRE) { finalize_producer(); __syncthreads(); }
  for (int kc = 0; kc < nchunks; ++kc) {
    store_chunk(kc);
    __syncthreads();
    if (kc + 1 < nchunks) load_chunk(kc + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = MIRROR ? 1 - tap / 3 : tap / 3 - 1, dx = MIRROR ? 1 - tap % 3 : tap % 3 - 1;
      bf16x8 fa[MI], fb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        fa[mi] = *reinterpret_cast<const bf16x8*>(As + ((wave * MI + mi + 1 + dy) * (TW + 2) + l15 + 1 + dx) * LDK + grp * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        fb[ni] = *reinterpret_cast<const bf16x8*>(Bs + (tap * BN + ni * 16 + l15) * LDK + grp * 8);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: bias, BatchNorm partial statistics, transposed store ------------------------------------------------
  bf16_t* Cs = lds;                                       // [TH*16][LDC]
  // (the statistics are only wanted by the forward launches, and only edge tiles need the per-pixel validity test: both are
  // block-uniform branches around ~40 VALU instructions of an issue-bound kernel)
  const bool full = y0 + TH <= a.H && x0 + TW <= a.W;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int lcol = ni * 16 + l15, col = n0 + lcol;
    const float bv = (a.bias && col < a.Cout) ? a.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int py = wave * MI + mi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[mi][ni][r] += bv;
        stf(Cs + (py * TW + grp * 4 + r) * LDC + lcol, acc[mi][ni][r]);
      }
    }
    if (a.stats) {
      if (full) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = acc[mi][ni][r]; s1 += v; s2 += v * v; }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const bool yok = y0 + wave * MI + mi < a.H;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[mi][ni][r];
            if (yok && x0 + grp * 4 + r < a.W) { s1 += v; s2 += v * v; }
          }
        }
      }
      s1 = rows_reduce<OpSum>(s1);                         // over the four 16-lane groups: lane swaps, no LDS round trip
      s2 = rows_reduce<OpSum>(s2);
      if (grp == 0) { sstat[wave * 2 * BN + lcol] = s1; sstat[(wave * 2 + 1) * BN + lcol] = s2; }
    }
  }
  __syncthreads();
  if (a.stats) {
    float* slot = a.stats + (size_t)(blockIdx.x % RSSF_BN_SLOTS) * 2 * a.Cout;
    float* part = a.stats_ws ? a.stats_ws + (size_t)tile_id * 2 * a.Cout : nullptr;
    for (int i = tid; i < BN; i += 256) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += sstat[w * 2 * BN + i]; t2 += sstat[(w * 2 + 1) * BN + i]; }
      if (n0 + i < a.Cout) {
        if (part) { part[n0 + i] = t1; part[a.Cout + n0 + i] = t2; }
        else { atomicAdd(slot + n0 + i, t1); atomicAdd(slot + a.Cout + n0 + i, t2); }
      }
    }
  }
  constexpr int OCPR = BN / 8;
  if ((a.Cout % 8) == 0) {
    // 16-byte rows through bounds-checked buffer accesses with 32-bit byte offsets (the dispatcher keeps the output below 2^31
    // bytes): a pixel outside the image or a channel chunk past Cout gets the out-of-range offset - the store is dropped, the
    // addend load returns zeros - so there is no branch and no 64-bit address arithmetic in the loop
    const int out_bytes = (int)((int64_t)a.B * a.H * a.W * a.Cout * 2);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.addend ? a.addend : a.out), 0, out_bytes, 0x00020000);
    static_assert((BMP * OCPR) % 256 == 0 && 256 % OCPR == 0, "whole passes of the 256 threads over the output chunks, fixed channel chunk");
    // fused BatchNorm-backward statistics (data-gradient launches): this thread's 8 channels are the same in every pass
    const bool bnb = a.bn_sums != nullptr;
    const int ccl = (tid % OCPR) * 8;                         // channel chunk of this thread inside the block's BN columns
    const __amdgpu_buffer_rsrc_t rraw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(bnb ? a.bn_raw : a.out), 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(bnb && a.bn_res ? a.bn_res : a.out), 0, out_bytes, 0x00020000);
    float bsc[8], bsh[8], t1[8], t2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool cok = bnb && n0 + ccl + e < a.Cout;
      bsc[e] = cok ? a.bn_ss[n0 + ccl + e] : 0.f;
      bsh[e] = cok ? a.bn_ss[a.Cout + n0 + ccl + e] : 0.f;
      t1[e] = 0.f; t2[e] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < BMP * OCPR / 256; ++it) {
      const int c = tid + it * 256;
      const int pix = c / OCPR, cc = (c % OCPR) * 8;
      const int gy = y0 + pix / TW, gx = x0 + pix % TW, col = n0 + cc;
      const bool ok = gy < a.H && gx < a.W && col < a.Cout;
      const unsigned off = ok ? (unsigned)((((b * a.H + gy) * a.W + gx) * a.Cout + col) * 2) : OOB;
      Vec<bf16_t> v, xr, xp;
      if (bnb) {
        xr.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rraw, off, 0, 0));
        if (a.bn_res) xp.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, off, 0, 0));
      }
      v.load(Cs + pix * LDC + cc);
      if (a.addend) {
        Vec<bf16_t> w;
        w.raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, off, 0, 0));
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v.get(e) + w.get(e);
        v.set_all(o);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v.raw), rout, off, 0, 0);
      if (bnb) {                                               // on the bf16 values just stored: what a separate pass would read
        if (!ok) v.clear();                                    // (an out-of-range chunk loaded zeros for raw: dz * raw is 0 anyway)
        auto accumulate = [&](auto ACT) {                      // block-uniform activation: one specialised loop runs
#pragma unroll
          for (int e = 0; e < 8; ++e) {
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
    }
    if (bnb) {
      // lanes with the same channel chunk: every OCPR-th lane of a 16-lane row (rotations by 4 / 8 inside the row), then the four
      // rows; one LDS row of partials per wave, summed in a fixed order, one global atomic per channel and sum per block
      __syncthreads();                                         // sstat may still be read by the forward statistics above
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (OCPR == 4) { t1[e] += dpp_mov<0x124>(t1[e]); t2[e] += dpp_mov<0x124>(t2[e]); }      // row_ror:4
        t1[e] += dpp_mov<0x128>(t1[e]); t2[e] += dpp_mov<0x128>(t2[e]);                          // row_ror:8
        t1[e] = rows_reduce<OpSum>(t1[e]); t2[e] = rows_reduce<OpSum>(t2[e]);
      }
      if (lane < OCPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sstat[wave * 2 * BN + ccl + e] = t1[e]; sstat[(wave * 2 + 1) * BN + ccl + e] = t2[e]; }
      }
      __syncthreads();
      float* slot = a.bn_sums + (size_t)(blockIdx.x % RSSF_BN_BWD_SLOTS) * 2 * a.Cout;
      for (int i = tid; i < BN; i += 256) {
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { u1 += sstat[w * 2 * BN + i]; u2 += sstat[(w * 2 + 1) * BN + i]; }
        if (n0 + i < a.Cout) { atomicAdd(slot + n0 + i, u1); atomicAdd(slot + a.Cout + n0 + i, u2); }
      }
    }
  } else {
    for (int c = tid; c < BMP * OCPR; c += 256) {
      const int pix = c / OCPR, cc = (c % OCPR) * 8;
      const int gy = y0 + pix / TW, gx = x0 + pix % TW, col = n0 + cc;
      if (gy >= a.H || gx >= a.W || col >= a.Cout) continue;
      bf16_t* dst = a.out + (((int64_t)b * a.H + gy) * a.W + gx) * a.Cout + col;
      const bf16_t* add = a.addend ? a.addend + (dst - a.out) : nullptr;
      for (int e = 0; e < 8 && col + e < a.Cout; ++e) stf(dst + e, ldf(Cs + pix * LDC + cc + e) + (add ? ldf(add + e) : 0.f));
    }
  }
}

template <int TH, int BN, bool MIRROR, bool PRE = false>
__global__ void __launch_bounds__(256) conv3x3_halo_kernel(HaloArgs a) {
  const int64_t q64 = xcd_logical(blockIdx.x, a.xcd_per);
  if (q64 >= a.total) return;
  halo_block<TH, BN, MIRROR, PRE>(a, (unsigned)q64);
}

// GROUPED launch: up to RSSF_GROUP_MAX independent problems (the parallel branches of a HighResolutionModule run the same
// BasicBlock step at 128^2 x 32, 64^2 x 64, 32^2 x 128, 16^2 x 256: _hrnet_rssformer.py:216-246, 410-423) as ONE grid.  Each of
// these launches is bound by the latency chain of its blocks (load -> LDS -> 9 taps -> epilogue, DESIGN.md lesson 32), not by a
// roofline: side by side in one grid the blocks of the four problems fill each other's bubbles and the launch ramp / tail is paid
// once.  Block b: XCD b & 7 (the hardware's round-robin), index b >> 3; problem i owns the indices [start[i], start[i+1]) and maps
// them XCD-major onto its own tiles exactly as the single-problem kernel does (neighbouring tiles of one problem share an L2).
// The host orders the problems by DESCENDING channel count: the long chains (8 chunks at 256 channels) start first, the short
// 32-channel blocks fill the tail.  One tile shape (8 x 16 pixels x 32 output channels) for every problem: the LDS footprint of a
// launch is static, the 64-column shape would halve the resident blocks of all problems.
struct HaloGroupArgs {
  HaloArgs it[RSSF_GROUP_MAX];
  int start[RSSF_GROUP_MAX + 1];
  int n;
};
template <bool MIRROR, bool PRE, int TH = 8>
__global__ void __launch_bounds__(256) conv3x3_halo_group_kernel(HaloGroupArgs g) {
  const unsigned xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  int i = 0;
#pragma unroll
  for (int k = 1; k < RSSF_GROUP_MAX; ++k)
    if (k < g.n && idx >= (unsigned)g.start[k]) i = k;
  const HaloArgs& a = g.it[i];
  const int64_t q = (int64_t)xcd * a.xcd_per + (idx - (unsigned)g.start[i]);
  if (q >= a.total) return;
  halo_block<TH, 32, MIRROR, PRE>(a, (unsigned)q);
}

}  // namespace

// true when the halo kernel covers this call (bf16 only; the caller checked the dtype)
// +1: forward tap order, -1: mirrored (data-gradient) order, 0: neither
static int tap_order(const int* dy, const int* dx) {
  int sign = 0;
  for (int s = 1; s >= -1 && !sign; s -= 2) {
    bool ok = true;
    for (int t = 0; t < 9; ++t) ok = ok && dy[t] == s * (t / 3 - 1) && dx[t] == s * (t % 3 - 1);
    if (ok) sign = s;
  }
  return sign;
}

bool halo_eligible(int IH, int IW, int Cin, int OH, int OW, int mul, int div, int ntaps, const int* dy, const int* dx) {
  if (mul != 1 || div != 1 || IH != OH || IW != OW || ntaps != 9 || (Cin % 8) != 0) return false;
  return tap_order(dy, dx) != 0;
}

static int tile_problem(HaloArgs& a, int th, int bn) {
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + th - 1) / th;
  a.ntiles_n = (a.Cout + bn - 1) / bn;
  const int64_t ntiles = (int64_t)a.B * a.tiles_y * a.tiles_x;
  if (ntiles >= ((int64_t)1 << 31)) { set_error("conv3x3_halo: %lld tiles exceed the 32-bit tile arithmetic", (long long)ntiles); return RSSF_ERR_UNSUPPORTED; }
  a.ntiles = (int)ntiles;
  a.total = ntiles * a.ntiles_n;                               // blocks: pixel tiles x output-channel columns
  if (a.total >= ((int64_t)1 << 31)) { set_error("conv3x3_halo: %lld blocks exceed the 32-bit tile arithmetic", (long long)a.total); return RSSF_ERR_UNSUPPORTED; }
  a.xcd_per = xcd_per(a.total);
  return RSSF_OK;
}

// n <= RSSF_GROUP_MAX problems in one grid (see conv3x3_halo_group_kernel); all forward-order or all mirrored, all with or all
// without a pre-activation input.  Deterministic-mode statistics workspaces are not served here (the caller launches one by one).
int launch_halo_group(HaloArgs* items, int n, hipStream_t st) {
  if (n < 1 || n > RSSF_GROUP_MAX) { set_error("conv3x3_halo_group: %d problems (1..%d)", n, RSSF_GROUP_MAX); return RSSF_ERR_BAD_ARG; }
  const bool mirror = tap_order(items[0].dy, items[0].dx) < 0, pre = items[0].pre_ss != nullptr;
  int order[RSSF_GROUP_MAX];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i)                                   // descending input channels (= chunks per block): long chains first
    for (int j = i; j > 0 && items[order[j]].Cin > items[order[j - 1]].Cin; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  HaloGroupArgs g;
  g.n = n;
  int idx = 0;
  constexpr int gth = 8;           // tile height of the grouped form (16-row tiles measured equal: round 4)
  for (int k = 0; k < n; ++k) {
    HaloArgs& a = items[order[k]];
    if ((tap_order(a.dy, a.dx) < 0) != mirror || (a.pre_ss != nullptr) != pre || a.stats_ws) {
      set_error("conv3x3_halo_group: the problems of one launch must share tap order and input kind");
      return RSSF_ERR_BAD_ARG;
    }
    if (pre && (mirror || a.Cin > 256)) { set_error("conv3x3_halo_group: a pre-activation input is a forward-launch feature (<= 256 channels)"); return RSSF_ERR_UNSUPPORTED; }
    if (const int rc = tile_problem(a, gth, 32)) return rc;
    g.it[k] = a;
    g.start[k] = idx;
    idx += a.xcd_per;
  }
  for (int k = n; k <= RSSF_GROUP_MAX; ++k) g.start[k] = idx;
  const dim3 grid((unsigned)idx * 8);
  if (mirror) conv3x3_halo_group_kernel<true, false><<<grid, 256, 0, st>>>(g);
  else if (pre) conv3x3_halo_group_kernel<false, true><<<grid, 256, 0, st>>>(g);
  else conv3x3_halo_group_kernel<false, false><<<grid, 256, 0, st>>>(g);
  return check_launch("conv3x3_halo_group");
}

int launch_halo(HaloArgs a, hipStream_t st) {
  const int tx = (a.W + TW - 1) / TW;
  const int64_t tiles8 = (int64_t)a.B * ((a.H + 7) / 8) * tx;
  int th = 8, bn = 32;
  if (a.Cout >= 64 && tiles8 * ((a.Cout + 63) / 64) >= 512) bn = 64;
  else if (tiles8 * ((a.Cout + 31) / 32) < 512) th = 4;
  const bool mirror = tap_order(a.dy, a.dx) < 0;
  if (const int rc = tile_problem(a, th, bn)) return rc;
  dim3 grid((unsigned)a.xcd_per * 8);
  if (a.pre_ss && (mirror || a.Cin > 256)) { set_error("conv3x3_halo: a pre-activation input is a forward-launch feature (<= 256 channels)"); return RSSF_ERR_UNSUPPORTED; }
#define RSSF_HALO(THv, BNv)                                                      \
  do {                                                                           \
    if (mirror) conv3x3_halo_kernel<THv, BNv, true><<<grid, 256, 0, st>>>(a);    \
    else if (a.pre_ss) conv3x3_halo_kernel<THv, BNv, false, true><<<grid, 256, 0, st>>>(a);  \
    else conv3x3_halo_kernel<THv, BNv, false><<<grid, 256, 0, st>>>(a);          \
  } while (0)
  if (th == 8 && bn == 64) RSSF_HALO(8, 64);
  else if (th == 8) RSSF_HALO(8, 32);
  else RSSF_HALO(4, 32);
#undef RSSF_HALO
  const int rc = check_launch("conv3x3_halo");
  if (rc || !(a.stats && a.stats_ws)) return rc;
  return launch_stats_fold(a.stats_ws, a.total / a.ntiles_n, a.Cout, a.stats, st);
}

}  // namespace cv
}  // namespace rssf
