// 3x3 / stride-1 / "same" convolution with 32 input and 32 output channels on channels-last bf16 activations as a ROW STREAM
// (gfx950 MFMA): the BasicBlock convolutions of HRNet's full-resolution branch (_hrnet_rssformer.py:216-246: 64 forward and 64
// data-gradient launches of a training step, all on the step's dependency chain beside the transformer blocks).
//
// The halo kernel (conv_halo.hip) stages a 10 x 18 pixel tile AND the nine weight slabs of a block through LDS and reads both MFMA
// operands back from there: at 32 channels the weights (18 KB) outweigh the pixels (11.5 KB), every MFMA costs two 1 KB fragment
// reads, and the kernel issues 20 VALU + 13 SALU instructions per MFMA.  Here nothing is staged:
//   * a wave owns a strip of 16 pixels x R32_R output rows of one image and keeps ALL 9 x 32 x 32 weights as 18 register fragments
//     (72 VGPRs; the transposed product: rows = output channels, column = pixel, as in conv_pw.hip);
//   * an input row of the strip is ONE coalesced 1 KB load that IS the MFMA operand (lane = pixel x 8-channel group) plus a
//     two-pixel edge load; the dx = -1 / +1 operands are the same registers shifted by one lane inside each 16-lane row (DPP
//     row_shr / row_shl, the edge pixel entering through the `old` operand): 8 VALU moves per input row;
//   * an input row feeds the three output rows around it (18 MFMAs), whose accumulators rotate through three register sets: every
//     input pixel is loaded once per wave, every output row is complete two iterations after its first tap and leaves as one
//     16-byte store per lane (fragment rows paired into eight consecutive channels, conv_pw.hip's permutation);
//   * the row loop is fully unrolled and branch-free around memory operations (out-of-image rows / columns get the out-of-range
//     buffer offset: loads return zeros, stores are dropped), input rows are requested R32_D rows ahead, the operands of a row's
//     epilogue (addend, BatchNorm-backward raw / residual) two iterations ahead: the compiler's vmcnt bookkeeping is exact.
// No LDS in the main loop, no barrier (one in the pre-activation prologue, one in the statistics fold).
// Variants as the halo kernel's: PRE (the producer's BatchNorm finalize + ReLU applied on load, rssf_conv_gather_preact), the
// following BatchNorm's statistics (forward), MIRROR (data gradient: mirrored taps on transposed weight slabs) with the skip
// gradient added before rounding (ADD) and the producer's BatchNorm-backward statistics (BNB: 1 without, 2 with a pre-activation
// residual; rssf_conv_gather_bnbwd).  Arithmetic of the fused parts: conv_halo.hip's, statement by statement.
#include <cstring>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace rssf {
namespace cv {

namespace {

constexpr int R32_R = 8;      // output rows of a wave
#ifndef RSSF_R32_D
#define RSSF_R32_D 5
#endif
constexpr int R32_D = RSSF_R32_D;      // input rows requested ahead of the one being consumed

struct R32Args {
  const bf16_t* in; const bf16_t* wpk; bf16_t* out; const float* bias; float* stats;
  const bf16_t* addend; const bf16_t* bn_raw; const bf16_t* bn_res; const float* bn_ss; float* bn_sums; int bn_act;
  const float* pre_stats; const float* pre_gamma; const float* pre_beta; float* pre_rmean; float* pre_rvar; float* pre_mi; float* pre_ss;
  float pre_n, pre_momentum, pre_eps; int pre_training, pre_act;
  int B, H, W, strips, bands, units, nblocks, xcd_per;
};

__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bfel(const u32x4& v, int e) { return (e & 1) ? bfhi(v[e >> 1]) : bflo(v[e >> 1]); }

// one lane to the right / left inside every 16-lane row; the lane without a source keeps `old`
__device__ __forceinline__ u32x4 shr1(const u32x4& old, const u32x4& v) {
  u32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)old[k], (int)v[k], 0x111, 0xf, 0xf, false);      // row_shr:1
  return r;
}
__device__ __forceinline__ u32x4 shl1(const u32x4& old, const u32x4& v) {
  u32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)old[k], (int)v[k], 0x101, 0xf, 0xf, false);      // row_shl:1
  return r;
}
__device__ __forceinline__ float r16sum(float v) {
  v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);       // row_half_mirror
  v += dpp_mov<0x140>(v);       // row_mirror
  return v;
}

template <bool MIRROR, bool PRE, int BNB, bool ADD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) conv3x3_rows32_kernel(R32Args a) {
  constexpr int R = R32_R, NI = R + 2;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ float sred[4][2][32];
  __shared__ __attribute__((aligned(16))) float spre[PRE ? 64 : 4];
  const int q = (int)xcd_logical(blockIdx.x, a.xcd_per);
  if (q >= a.nblocks) return;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this wave's unit: image b, row band, 16-pixel strip (a block: four neighbouring strips)
  const int u = q * 4 + wave;
  const bool live = u < a.units;
  const int us = live ? u : 0;
  const int strip = us % a.strips, t1 = us / a.strips;
  const int band = t1 % a.bands, b = t1 / a.bands;
  const int x0 = live ? strip * 16 : a.W, y0 = band * R;
  const int H = a.H, W = a.W;
  const int bytes = (int)((int64_t)a.B * H * W * 64);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ADD ? a.addend : a.in), 0, ADD ? bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rraw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(BNB ? a.bn_raw : a.in), 0, BNB ? bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(BNB == 2 ? a.bn_res : a.in), 0, BNB == 2 ? bytes : 0, 0x00020000);

  // per-lane column parts of the byte offsets (out of the image: the out-of-range sentinel, which survives the row term below)
  const bool colok = x0 + l15 < W;
  const unsigned moff = colok ? (unsigned)((x0 + l15) * 64 + grp * 16) : OOB;
  const bool eok = (l15 == 0 && x0 > 0 && x0 < W) || (l15 == 15 && x0 + 16 < W);
  const unsigned eoff = eok ? (unsigned)((l15 == 0 ? x0 - 1 : x0 + 16) * 64 + grp * 16) : OOB;
  auto rowbase = [&](int r) -> unsigned { return (unsigned)((b * H + r) * W) * 64u; };      // scalar
  auto rowok = [&](int r) -> bool { return r >= 0 && r < H; };

  // ---- weights: fragment (tap, j): row l15 = output channel 8 (l15 >> 2) + 4 j + (l15 & 3), input channels 8 grp .. + 7 ----------
  bf16x8 fw[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      fw[t][j] = *reinterpret_cast<const bf16x8*>(a.wpk + (size_t)((t * 32 + 8 * (l15 >> 2) + 4 * j + (l15 & 3)) * 32 + grp * 8));

  // ---- input rows y0 - 1 .. y0 + R: requested R32_D rows ahead ---------------------------------------------------------------------
  u32x4 xm[NI], xe[NI];
  auto request = [&](int i) {
    const int r = y0 - 1 + i;
    const bool ok = rowok(r);
    const unsigned base = rowbase(r);
    xm[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? moff + base : OOB, 0, 0));
    xe[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? eoff + base : OOB, 0, 0));
  };
#pragma unroll
  for (int i = 0; i <= R32_D && i < NI; ++i) request(i);

  // epilogue operands of output row y0 + k (addend, BatchNorm-backward raw / residual): requested two iterations before their use
  u32x4 ea[ADD ? R : 1], er[BNB ? R : 1], ep[BNB == 2 ? R : 1];
  auto out_off = [&](int k) -> unsigned {
    const int y = y0 + k;
    return (y < H) ? moff + rowbase(y) : OOB;
  };
  auto request_epi = [&](int k) {
    const unsigned off = out_off(k);
    if constexpr (ADD) ea[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, off, 0, 0));
    if constexpr (BNB != 0) er[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rraw, off, 0, 0));
    if constexpr (BNB == 2) ep[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, off, 0, 0));
  };

  // ---- per-lane constants of this lane's eight channels 8 grp .. 8 grp + 7 ----------------------------------------------------------
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = (!MIRROR && a.bias) ? a.bias[grp * 8 + e] : 0.f;
  float bsc[BNB ? 8 : 1], bsh[BNB ? 8 : 1];
  if constexpr (BNB != 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { bsc[e] = a.bn_ss[grp * 8 + e]; bsh[e] = a.bn_ss[32 + grp * 8 + e]; }
  }
  const float bthr = (BNB != 0 && a.bn_act == 1) ? 0.f : -__builtin_inff();      // dz = z > thr ? g : 0  (ReLU / identity)

  // PRE: finalize the producer's BatchNorm (conv_halo.hip::finalize_producer, bn_finalize_kernel's arithmetic) under the loads above
  float psc[PRE ? 8 : 1], psh[PRE ? 8 : 1];
  float plo = 0.f;
  if constexpr (PRE) {
    if (tid < 32) {
      const int c = tid;
      float mean, var;
      if (a.pre_training) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < RSSF_BN_SLOTS; ++k) { s1 += a.pre_stats[(size_t)k * 64 + c]; s2 += a.pre_stats[(size_t)k * 64 + 32 + c]; }
        mean = s1 / a.pre_n;
        var = fmaxf(s2 / a.pre_n - mean * mean, 0.f);
      } else {
        mean = a.pre_rmean[c];
        var = a.pre_rvar[c];
      }
      const float invstd = rsqrtf(var + a.pre_eps);
      const float sc = a.pre_gamma[c] * invstd, sh = a.pre_beta[c] - mean * sc;
      spre[c] = sc; spre[32 + c] = sh;
      if (q == 0) {                                                          // one block publishes for the backward pass
        a.pre_mi[c] = mean; a.pre_mi[32 + c] = invstd;
        a.pre_ss[c] = sc; a.pre_ss[32 + c] = sh;
        if (a.pre_training && a.pre_rmean) {
          a.pre_rmean[c] = (1.f - a.pre_momentum) * a.pre_rmean[c] + a.pre_momentum * mean;
          a.pre_rvar[c] = (1.f - a.pre_momentum) * a.pre_rvar[c] + a.pre_momentum * var * (a.pre_n > 1.f ? a.pre_n / (a.pre_n - 1.f) : 1.f);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { psc[e] = spre[grp * 8 + e]; psh[e] = spre[32 + grp * 8 + e]; }
    plo = a.pre_act == 1 ? 0.f : -__builtin_inff();                          // act(z) = max(z, lo): ReLU / identity
  }
  // act(raw * scale + shift) of the producer on an operand register set; zero where the ACTIVATION is padding (outside the image)
  auto preact = [&](u32x4& v, bool ok) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaxf(fmaf(bfel(v, e), psc[PRE ? e : 0], psh[PRE ? e : 0]), plo);
    const u32x4 p = {f2bf2(o[0], o[1]), f2bf2(o[2], o[3]), f2bf2(o[4], o[5]), f2bf2(o[6], o[7])};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ok ? p[k] : 0u;
  };

  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  const bool want = BNB != 0 || a.stats != nullptr;

  // ---- the rows ------------------------------------------------------------------------------------------------------------------
  // input row r = y0 - 1 + i meets output rows r + 1 (role 0: its first taps), r (role 1) and r - 1 (role 2: complete afterwards);
  // output row y0 + k accumulates in acc[k % 3]
  f32x4 acc[3][2];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (i + R32_D + 1 < NI) request(i + R32_D + 1);
    if (i < R) request_epi(i);
    if constexpr (PRE) {
      const bool rk = rowok(y0 - 1 + i);
      preact(xm[i], rk && colok);
      preact(xe[i], rk && eok);
    }
    const u32x4 f0 = xm[i];
    const u32x4 fl = shr1(xe[i], f0);       // pixel x - 1 in lane x
    const u32x4 fr = shl1(xe[i], f0);       // pixel x + 1 in lane x
#pragma unroll
    for (int s = 0; s < 3; ++s) {           // dx = s - 1
      const bf16x8 fx = __builtin_bit_cast(bf16x8, s == 0 ? fl : s == 1 ? f0 : fr);
      const int kx = MIRROR ? 2 - s : s;
#pragma unroll
      for (int role = 2; role >= 0; --role) {
        const int k = i - role;             // output row y0 + k
        if (k < 0 || k >= R) continue;
        const int ky = MIRROR ? 2 - role : role;
        const int t = ky * 3 + kx;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 c;
          if (role == 0 && s == 0) c = f32x4{bias8[4 * j], bias8[4 * j + 1], bias8[4 * j + 2], bias8[4 * j + 3]};
          else c = acc[k % 3][j];
          acc[k % 3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t][j], fx, c, 0, 0, 0);
        }
      }
    }
    // ---- output row y0 + i - 2 is complete ---------------------------------------------------------------------------------------
    if (i >= 2) {
      const int k = i - 2;
      const bool ok = colok && (y0 + k < H);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = acc[k % 3][e >> 2][e & 3];
        if constexpr (ADD) v[e] += bfel(ea[k], e);
        v[e] = ok ? v[e] : 0.f;
      }
      const u32x4 o = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7])};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, o), rout, out_off(k), 0, 0);
      if constexpr (BNB != 0) {
        // on the bf16 values just stored: what a separate pass would read (conv_halo.hip's epilogue)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = bfel(er[k], e);
          float z = fmaf(x, bsc[e], bsh[e]);
          if constexpr (BNB == 2) z += bfel(ep[k], e);
          const float g = bfel(o, e);
          const float dz = z > bthr ? g : 0.f;
          s1[e] += dz; s2[e] = fmaf(dz, x, s2[e]);
        }
      } else {
        if (want) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
        }
      }
    }
  }
  if (!want) return;
  // fold: the 16 pixel lanes of a row, the block's waves through LDS, one atomic per channel and sum
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = r16sum(s1[e]); s2[e] = r16sum(s2[e]); }
  if (l15 == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sred[wave][0][grp * 8 + e] = s1[e]; sred[wave][1][grp * 8 + e] = s2[e]; }
  }
  __syncthreads();
  if (tid < 64) {
    const int w = tid >> 5, c = tid & 31;
    const float t = (sred[0][w][c] + sred[1][w][c]) + (sred[2][w][c] + sred[3][w][c]);
    float* slot = BNB != 0 ? a.bn_sums + (size_t)(blockIdx.x % RSSF_BN_BWD_SLOTS) * 64 : a.stats + (size_t)(blockIdx.x % RSSF_BN_SLOTS) * 64;
    atomicAdd(slot + w * 32 + c, t);
  }
}

}  // namespace

bool rows32_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx) {
  if (Cin != 32 || Cout != 32 || (int64_t)B * IH * IW * 32 >= ((int64_t)1 << 30)) return false;
  return halo_eligible(IH, IW, Cin, OH, OW, mul, div, ntaps, dy, dx);
}

// pre_act / bn_act: identity or ReLU (the GELU forms stay on the halo kernel)
int launch_rows32(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend, const void* bn_raw,
                  const void* bn_res, const float* bn_ss, float* bn_sums, int bn_act, const PwPre* pre, int B, int H, int W, bool mirror,
                  hipStream_t st) {
  R32Args a;
  memset(&a, 0, sizeof(a));
  a.in = (const bf16_t*)in; a.wpk = (const bf16_t*)wpk; a.out = (bf16_t*)out; a.bias = bias; a.stats = stats;
  a.addend = (const bf16_t*)addend; a.bn_raw = (const bf16_t*)bn_raw; a.bn_res = (const bf16_t*)bn_res; a.bn_ss = bn_ss; a.bn_sums = bn_sums;
  a.bn_act = bn_act;
  if (pre) {
    a.pre_stats = pre->stats; a.pre_gamma = pre->gamma; a.pre_beta = pre->beta; a.pre_rmean = pre->rmean; a.pre_rvar = pre->rvar;
    a.pre_mi = pre->mi; a.pre_ss = pre->ss; a.pre_n = pre->n; a.pre_momentum = pre->momentum; a.pre_eps = pre->eps;
    a.pre_training = pre->training; a.pre_act = pre->act;
  }
  a.B = B; a.H = H; a.W = W;
  a.strips = (W + 15) / 16;
  a.bands = (H + R32_R - 1) / R32_R;
  a.units = B * a.bands * a.strips;
  a.nblocks = (a.units + 3) / 4;
  a.xcd_per = xcd_per(a.nblocks);
  const dim3 grid((unsigned)a.xcd_per * 8);
  if (!mirror) {
    if (addend || bn_sums) { set_error("conv3x3_rows32: addend / BatchNorm-backward statistics are data-gradient features"); return RSSF_ERR_UNSUPPORTED; }
    if (pre) conv3x3_rows32_kernel<false, true, 0, false><<<grid, 256, 0, st>>>(a);
    else conv3x3_rows32_kernel<false, false, 0, false><<<grid, 256, 0, st>>>(a);
  } else {
    if (pre || stats || bias) { set_error("conv3x3_rows32: pre-activation input / statistics / bias are forward features"); return RSSF_ERR_UNSUPPORTED; }
    const int bnb = !bn_sums ? 0 : bn_res ? 2 : 1;
#define RSSF_R32(BNBv)                                                                         \
  do {                                                                                         \
    if (addend) conv3x3_rows32_kernel<true, false, BNBv, true><<<grid, 256, 0, st>>>(a);       \
    else conv3x3_rows32_kernel<true, false, BNBv, false><<<grid, 256, 0, st>>>(a);             \
  } while (0)
    if (bnb == 0) RSSF_R32(0); else if (bnb == 1) RSSF_R32(1); else RSSF_R32(2);
#undef RSSF_R32
  }
  return check_launch("conv3x3_rows32");
}

}  // namespace cv
}  // namespace rssf
