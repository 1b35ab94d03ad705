// Point-wise (1 x 1) convolution 32 -> 128 channels on channels-last bf16 activations as a STREAM (gfx950 MFMA): MlpDWBN's fc1
// forward and fc2's data gradient (ffn_block.py:219-225, 246-249: Conv2d(C, 4C, 1) / the transpose of Conv2d(4C, C, 1)), 16 launches
// of a training step.
//
// In the generic gather kernel (conv_fwd.hip) this shape is ONE K-step per block: a 128 x 128 tile runs its tap tables, two LDS
// stagings, barriers and the LDS-transposed epilogue around 16 MFMAs per wave - 50 us (85 us with the fused BatchNorm-backward
// statistics) for a pass that reads 17 MB and writes 67 MB.  Here nothing is staged: a wave keeps the whole [128][32] weight matrix
// as eight register fragments and the bias as accumulator seeds, and walks 16-pixel tiles - a tile's 16 x 64 bytes are one
// coalesced 1 KB load that IS the MFMA operand (lane = pixel x 8-channel group), eight MFMAs form the transposed result
// (rows = output channels, column = pixel), so a lane holds four consecutive channels of one pixel per fragment.  The ROWS of a
// fragment pair are a permutation of 32 channels chosen so that a lane's 4 + 4 values are EIGHT consecutive channels (fragment
// 2q: rows 4 g + r = channel 32 q + 8 g + r, fragment 2q + 1: channel 32 q + 8 g + 4 + r - only the weight rows are loaded in that
// order): 16-byte stores, and 16-byte loads of everything the epilogue reads beside (BatchNorm-backward operands, an addend: with
// 8-byte pieces a wave instruction touched 16 cache lines for 32 bytes each, and the addend form ran slower than the generic kernel).
// The next tile's load is in flight under the current tile's MFMAs and stores.  Epilogue per wave: the per-channel sum / sum of
// squares for the following BatchNorm (forward) or {sum dz, sum dz * raw} of the producer's BatchNorm backward (data gradient,
// rssf_conv_gather_bnbwd) accumulate in registers across a wave's tiles, are folded over the 16 pixel lanes by DPP, over the
// block's waves through LDS, and leave as one atomic per channel and block.
#include <cstring>
#include <mutex>
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace rssf {
namespace cv {

namespace {


struct PwArgs {
  const bf16_t* in; const bf16_t* wpk; bf16_t* out; const float* bias; float* stats;
  const bf16_t* bn_raw; const bf16_t* bn_res; const float* bn_ss; float* bn_sums; int bn_act;
  const bf16_t* addend;    // [M][Cout] added to the result before rounding (ADD), or null
  int64_t M;            // pixels
  int CoutP, CinP, Cout;   // Cout: channels of a pixel row of `out`; a block computes the slice [16 NT blockIdx.y, + 16 NT)
  // PRE (rssf_conv_gather_preact): `in` is the RAW output of the producing convolution; its BatchNorm is finalized here (the arguments
  // of rssf_bn_finalize, see HaloArgs::pre_* in conv.hip.h) and act(in * scale + shift) is formed on the operand registers
  const float* pre_stats; const float* pre_gamma; const float* pre_beta; float* pre_rmean; float* pre_rvar; float* pre_mi; float* pre_ss;
  float pre_n, pre_momentum, pre_eps; int pre_training, pre_act;
};

// sum over the 16 lanes of a row (the pixels of a tile), result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);       // row_half_mirror
  v += dpp_mov<0x140>(v);       // row_mirror
  return v;
}

// channel (inside the block's slice, less 8 grp) of value r of fragment j
__device__ __forceinline__ constexpr int pw_co(int j, int r) { return 32 * (j >> 1) + 4 * (j & 1) + r; }

// KS = input channels / 32 (K-steps), NT = output channels / 16 (fragments): (1, 8) = 32 -> 128, (4, 2) = 128 -> 32
// ADD: out = conv + addend (an accumulating data gradient; the addend may be `out` itself), added before rounding
template <int KS, int NT, bool BNB, bool PRE = false, bool ADD = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) conv_pw_kernel(PwArgs a) {
  constexpr int PW_K = 32 * KS, PW_N = 16 * NT, PW_NT = NT, NQ = NT / 2;
  static_assert(NT % 2 == 0, "fragments come in pairs");
  __shared__ float sred[4][2][PW_N];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = (int)blockIdx.y * PW_N, CO = a.Cout;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, (int)(a.M * PW_K * 2), 0x00020000);
  // weights: row l15 of fragment j = output channel pw_co(j, l15 & 3) + 8 (l15 >> 2) (lane: that channel, input channels 8 grp .. 8 grp + 7)
  bf16x8 fw[PW_NT][KS];
#pragma unroll
  for (int j = 0; j < PW_NT; ++j)
#pragma unroll
    for (int k = 0; k < KS; ++k)
      fw[j][k] = *reinterpret_cast<const bf16x8*>(a.wpk + (size_t)(n0 + pw_co(j, l15 & 3) + 8 * (l15 >> 2)) * a.CinP + k * 32 + grp * 8);
  // accumulator seeds: the bias of this lane's four channels of every fragment
  f32x4 seed[PW_NT];
#pragma unroll
  for (int j = 0; j < PW_NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) seed[j][r] = (!BNB && a.bias) ? a.bias[n0 + pw_co(j, r) + 8 * grp] : 0.f;      // (a data gradient has no bias)
  // BatchNorm-backward statistics: scale / shift of the 128 channels in LDS, read per fragment (64 registers otherwise)
  __shared__ __attribute__((aligned(16))) float sss[BNB ? 2 * PW_N : 4];
  if constexpr (BNB) {
    if (tid < 2 * PW_N) sss[tid] = a.bn_ss[(tid / PW_N) * CO + n0 + tid % PW_N];      // [scale][shift] of this slice
    __syncthreads();
  }
  // PRE: scale / shift of the PW_K input channels (the arithmetic of bn_finalize_kernel / conv_halo.hip's finalize_producer), read back
  // per tile: a lane's channels are 32 k + 8 grp .. + 7 of every K-step k
  __shared__ __attribute__((aligned(16))) float spre[PRE ? 2 * PW_K : 4];
  if constexpr (PRE) {
    if (tid < PW_K) {
      const int c = tid;
      float mean, var;
      if (a.pre_training) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < RSSF_BN_SLOTS; ++k) { t1 += a.pre_stats[(size_t)k * 2 * PW_K + c]; t2 += a.pre_stats[(size_t)k * 2 * PW_K + PW_K + c]; }
        mean = t1 / a.pre_n;
        var = fmaxf(t2 / a.pre_n - mean * mean, 0.f);
      } else {
        mean = a.pre_rmean[c];
        var = a.pre_rvar[c];
      }
      const float invstd = rsqrtf(var + a.pre_eps);
      const float sc = a.pre_gamma[c] * invstd, sh = a.pre_beta[c] - mean * sc;
      spre[c] = sc; spre[PW_K + c] = sh;
      if (blockIdx.x == 0 && blockIdx.y == 0) {                              // one block publishes for the backward pass
        a.pre_mi[c] = mean; a.pre_mi[PW_K + c] = invstd;
        a.pre_ss[c] = sc; a.pre_ss[PW_K + c] = sh;
        if (a.pre_training && a.pre_rmean) {
          a.pre_rmean[c] = (1.f - a.pre_momentum) * a.pre_rmean[c] + a.pre_momentum * mean;
          a.pre_rvar[c] = (1.f - a.pre_momentum) * a.pre_rvar[c] + a.pre_momentum * var * (a.pre_n > 1.f ? a.pre_n / (a.pre_n - 1.f) : 1.f);
        }
      }
    }
    __syncthreads();
  }
  const bool want = BNB || a.stats != nullptr;
  float s1[PW_NT * 4], s2[PW_NT * 4];
#pragma unroll
  for (int e = 0; e < PW_NT * 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

  const int64_t ntiles = (a.M + 15) / 16;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  constexpr unsigned OOB = 0x80000000u;
  auto tile_off = [&](int64_t tt) -> unsigned {            // byte offset of this lane's first 16-byte piece; past the end: zeros
    const int64_t pix = tt * 16 + l15;
    return (tt < ntiles && pix < a.M) ? (unsigned)(pix * PW_K * 2 + grp * 16) : OOB;
  };
  // the addend rides one tile ahead like the input (a lane's pieces: 16 bytes per fragment pair, 64 bytes apart)
  const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ADD ? a.addend : a.in), 0, (int)(ADD ? a.M * CO * 2 : 0), 0x00020000);
  auto add_off = [&](int64_t tt) -> unsigned {
    const int64_t pix = tt * 16 + l15;
    return (tt < ntiles && pix < a.M) ? (unsigned)((pix * CO + n0 + grp * 8) * 2) : OOB;
  };
  u32x4 xa[KS], ada[ADD ? NQ : 1];
  {
    const unsigned o0 = tile_off(t);
#pragma unroll
    for (int k = 0; k < KS; ++k) xa[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, o0, k * 64, 0));
    if constexpr (ADD) {
      const unsigned q0 = add_off(t);
#pragma unroll
      for (int q = 0; q < NQ; ++q) ada[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, q0, q * 64, 0));
    }
  }
  for (; t < ntiles; t += stride) {
    u32x4 xc[KS], adc[ADD ? NQ : 1];
    const unsigned o1 = tile_off(t + stride);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      xc[k] = xa[k];
      xa[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, o1, k * 64, 0));      // unconditional: exact vmcnt
    }
    if constexpr (ADD) {
      const unsigned q1 = add_off(t + stride);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        adc[q] = ada[q];
        ada[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, q1, q * 64, 0));
      }
    }
    const int64_t pix = t * 16 + l15;
    const bool pok = pix < a.M;
    if constexpr (PRE) {
      // act(raw * scale + shift) on the operand registers (a pixel past the end loaded zeros and is never stored)
      auto apply = [&](auto ACT) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const f32x4 sc0 = *reinterpret_cast<const f32x4*>(spre + k * 32 + grp * 8), sc1 = *reinterpret_cast<const f32x4*>(spre + k * 32 + grp * 8 + 4);
          const f32x4 sh0 = *reinterpret_cast<const f32x4*>(spre + PW_K + k * 32 + grp * 8), sh1 = *reinterpret_cast<const f32x4*>(spre + PW_K + k * 32 + grp * 8 + 4);
          Vec<bf16_t> v;
          v.raw = xc[k];
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z = fmaf(v.get(e), e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
            o[e] = decltype(ACT)::value == 1 ? fmaxf(z, 0.f) : decltype(ACT)::value == 2 ? gelu_erf(z) : z;
          }
          v.set_all(o);
          xc[k] = v.raw;
        }
      };
      if (a.pre_act == 1) apply(std::integral_constant<int, 1>{});
      else if (a.pre_act == 2) apply(std::integral_constant<int, 2>{});
      else apply(std::integral_constant<int, 0>{});
    }
    f32x4 acc[PW_NT];
#pragma unroll
    for (int j = 0; j < PW_NT; ++j) {
      acc[j] = seed[j];
#pragma unroll
      for (int k = 0; k < KS; ++k) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][k], __builtin_bit_cast(bf16x8, xc[k]), acc[j], 0, 0, 0);
    }
    if constexpr (ADD) {
#pragma unroll
      for (int j = 0; j < PW_NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned w = adc[j >> 1][2 * (j & 1) + (r >> 1)];
          acc[j][r] += (r & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
        }
    }
    bf16_t* orow = a.out + pix * CO + n0 + grp * 8;
    u32x4 rawv[BNB ? NQ : 1], resv[BNB ? NQ : 1];
    if constexpr (BNB) {
      if (pok) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          rawv[q] = *reinterpret_cast<const u32x4*>(a.bn_raw + pix * CO + n0 + grp * 8 + q * 32);
          if (a.bn_res) resv[q] = *reinterpret_cast<const u32x4*>(a.bn_res + pix * CO + n0 + grp * 8 + q * 32);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const u32x4 o = {f2bf2(acc[2 * q][0], acc[2 * q][1]), f2bf2(acc[2 * q][2], acc[2 * q][3]),
                       f2bf2(acc[2 * q + 1][0], acc[2 * q + 1][1]), f2bf2(acc[2 * q + 1][2], acc[2 * q + 1][3])};
      if (pok) *reinterpret_cast<u32x4*>(orow + q * 32) = o;
      if (want && pok) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * q + h;
          if constexpr (BNB) {
            // on the bf16 values just stored: what a separate pass would read
            const f32x4 bsc4 = *reinterpret_cast<const f32x4*>(sss + pw_co(j, 0) + grp * 8), bsh4 = *reinterpret_cast<const f32x4*>(sss + PW_N + pw_co(j, 0) + grp * 8);
            auto accumulate = [&](auto ACT) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const unsigned ow = o[2 * h + (r >> 1)], rw = rawv[q][2 * h + (r >> 1)];
                const float g = (r & 1) ? __uint_as_float(ow & 0xffff0000u) : __uint_as_float(ow << 16);
                const float x = (r & 1) ? __uint_as_float(rw & 0xffff0000u) : __uint_as_float(rw << 16);
                float z = fmaf(x, bsc4[r], bsh4[r]);
                if (a.bn_res) { const unsigned pw = resv[q][2 * h + (r >> 1)]; z += (r & 1) ? __uint_as_float(pw & 0xffff0000u) : __uint_as_float(pw << 16); }
                const float dz = decltype(ACT)::value == 1 ? (z > 0.f ? g : 0.f) : decltype(ACT)::value == 2 ? g * gelu_erf_grad(z) : g;
                s1[j * 4 + r] += dz; s2[j * 4 + r] = fmaf(dz, x, s2[j * 4 + r]);
              }
            };
            if (a.bn_act == 1) accumulate(std::integral_constant<int, 1>{});
            else if (a.bn_act == 2) accumulate(std::integral_constant<int, 2>{});
            else accumulate(std::integral_constant<int, 0>{});
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float v = acc[j][r]; s1[j * 4 + r] += v; s2[j * 4 + r] = fmaf(v, v, s2[j * 4 + r]); }
          }
        }
      }
    }
  }
  if (!want) return;
  // fold: the 16 pixel lanes of a row, then the block's waves, then one atomic per channel and sum
#pragma unroll
  for (int e = 0; e < PW_NT * 4; ++e) { s1[e] = row16_sum(s1[e]); s2[e] = row16_sum(s2[e]); }
  if (l15 == 0) {
#pragma unroll
    for (int j = 0; j < PW_NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sred[wave][0][pw_co(j, r) + 8 * grp] = s1[j * 4 + r]; sred[wave][1][pw_co(j, r) + 8 * grp] = s2[j * 4 + r]; }
  }
  __syncthreads();
  if (tid < PW_N) {
    const float u1 = (sred[0][0][tid] + sred[1][0][tid]) + (sred[2][0][tid] + sred[3][0][tid]);
    const float u2 = (sred[0][1][tid] + sred[1][1][tid]) + (sred[2][1][tid] + sred[3][1][tid]);
    float* slot = BNB ? a.bn_sums + (size_t)(blockIdx.x % RSSF_BN_BWD_SLOTS) * 2 * CO : a.stats + (size_t)(blockIdx.x % RSSF_BN_SLOTS) * 2 * CO;
    atomicAdd(slot + n0 + tid, u1);
    atomicAdd(slot + CO + n0 + tid, u2);
  }
}

}  // namespace

bool pw_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx) {
  return ntaps == 1 && dy[0] == 0 && dx[0] == 0 && mul == 1 && div == 1 && IH == OH && IW == OW &&
         ((Cin == 32 && Cout == 128) || (Cin == 128 && Cout == 32) || (Cin == 64 && Cout == 256)) &&
         (int64_t)B * IH * IW * (Cin > Cout ? Cin : Cout) < ((int64_t)1 << 30);
}

bool pw_addend_eligible(int Cin, int Cout) { return Cin == 64 && Cout == 256; }

bool pw_preact_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx) {
  return Cin == 128 && Cout == 32 && pw_eligible(B, IH, IW, Cin, OH, OW, Cout, mul, div, ntaps, dy, dx);
}

int launch_pw(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* bn_raw, const void* bn_res,
              const float* bn_ss, float* bn_sums, int bn_act, int B, int H, int W, int Cin, int Cout, int CinP, int CoutP, hipStream_t st,
              const PwPre* pre, const void* addend) {
  PwArgs a;
  memset(&a, 0, sizeof(a));
  if (addend && Cin != 64) { set_error("conv_pw: an addend is a feature of the 64 -> 256 kernel"); return RSSF_ERR_UNSUPPORTED; }
  a.addend = (const bf16_t*)addend;
  if (pre) {
    if (bn_sums || !(Cin == 128 && Cout == 32)) { set_error("conv_pw: a pre-activation input is a forward feature of the 128 -> 32 kernel"); return RSSF_ERR_UNSUPPORTED; }
    a.pre_stats = pre->stats; a.pre_gamma = pre->gamma; a.pre_beta = pre->beta; a.pre_rmean = pre->rmean; a.pre_rvar = pre->rvar;
    a.pre_mi = pre->mi; a.pre_ss = pre->ss; a.pre_n = pre->n; a.pre_momentum = pre->momentum; a.pre_eps = pre->eps;
    a.pre_training = pre->training; a.pre_act = pre->act;
  }
  a.in = (const bf16_t*)in; a.wpk = (const bf16_t*)wpk; a.out = (bf16_t*)out; a.bias = bias; a.stats = stats;
  a.bn_raw = (const bf16_t*)bn_raw; a.bn_res = (const bf16_t*)bn_res; a.bn_ss = bn_ss; a.bn_sums = bn_sums; a.bn_act = bn_act;
  a.M = (int64_t)B * H * W; a.CoutP = CoutP; a.CinP = CinP; a.Cout = Cout;
  const int64_t ntiles = (a.M + 15) / 16;
  int64_t blocks = (ntiles + 3) / 4;
  constexpr int maxb = 512;                                    // (measured at 16 x 128^2: 512 blocks 17 / 36 us, 1 024: 20 / 39, 2 048: 25 / 45)
  if (blocks > maxb) blocks = maxb;
  if (Cin == 64) {                                             // 64 -> 256 (layer1's Bottleneck expansions): two slices of 128 channels
    const dim3 grid2((unsigned)blocks, 2);
    if (addend) {                                              // layer1's conv1 data gradients accumulate onto the residual path's gradient
      if (bn_sums) conv_pw_kernel<2, 8, true, false, true><<<grid2, 256, 0, st>>>(a);
      else conv_pw_kernel<2, 8, false, false, true><<<grid2, 256, 0, st>>>(a);
    } else if (bn_sums) conv_pw_kernel<2, 8, true><<<grid2, 256, 0, st>>>(a);
    else conv_pw_kernel<2, 8, false><<<grid2, 256, 0, st>>>(a);
    return check_launch("conv_pw");
  }
  const dim3 grid((unsigned)blocks);
  if (Cin == 32) {
    if (bn_sums) conv_pw_kernel<1, 8, true><<<grid, 256, 0, st>>>(a);
    else conv_pw_kernel<1, 8, false><<<grid, 256, 0, st>>>(a);
  } else {
    if (bn_sums) conv_pw_kernel<4, 2, true><<<grid, 256, 0, st>>>(a);
    else if (pre) conv_pw_kernel<4, 2, false, true><<<grid, 256, 0, st>>>(a);
    else conv_pw_kernel<4, 2, false><<<grid, 256, 0, st>>>(a);
  }
  return check_launch("conv_pw");
}

}  // namespace cv
}  // namespace rssf
