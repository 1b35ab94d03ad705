// Forward of the STEM's first convolution (reference _hrnet_rssformer.py:407-413, 441-447: Conv2d(3, 64, 3, stride 2, padding 1, bias=False) on the
// 512 x 512 image, channels padded 3 -> 8: one pixel = one 16-byte piece) as a STREAM with the following BatchNorm's statistics in the
// epilogue - the first convolution of the step, nothing runs beside it.
//
// In the generic gather kernel the layer is nine K-steps of 32 channels of which 8 are real (3/4 of every staged slab and of every MFMA
// are channel padding), each with its gather, LDS staging and barrier: 97 us for a pass that reads 67 MB and writes 134 MB.  Here the
// im2col row of an output pixel - 9 taps x 8 channels = 72 values - is the K axis directly: three K-steps of four taps; lane
// (pixel, tap-in-step) loads ITS tap's 16-byte pixel straight into the MFMA operand register (out of the image, or tap 9..11: the
// sentinel offset -> zeros), the [64][72] weights live in registers as twelve fragments per wave, and the MFMAs form the transposed
// result - a lane holds four consecutive output channels of one pixel per fragment, eight per fragment pair (the weight rows of a pair
// are fetched as conv_pw.hip's 32-channel permutation): 16-byte stores.  Nothing is staged; the next
// tile's three loads are in flight under the current tile's twelve MFMAs and its stores.
#include <cstring>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace {

struct StemFwdArgs {
  const bf16_t* in; const bf16_t* wpk; bf16_t* out; float* stats;
  int B, IH, IW, OH, OW, CoutP, CinP;
  int64_t ntiles;          // B * OH * OW / 16
};

// channel (less 8 grp) of value r of result fragment j (conv_pw.hip's pairing)
__device__ __forceinline__ constexpr int sf_co(int j, int r) { return 32 * (j >> 1) + 4 * (j & 1) + r; }

__device__ __forceinline__ float row16_sum_sf(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) conv_stem_fwd_kernel(StemFwdArgs a) {
  constexpr int CO = 64, NT = 4, KS = 3;
  __shared__ float sred[4][2][CO];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t xbytes = (int64_t)a.B * a.IH * a.IW * 16;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, (int)xbytes, 0x00020000);
  // this lane's tap of every K-step and its weight fragments: channel 16 j + l15, kernel position 4 ks + grp (positions 9 .. 11: zeros)
  int dyk[KS], dxk[KS];
  bool tapok[KS];
  bf16x8 fw[NT][KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int tap = ks * 4 + grp;
    tapok[ks] = tap < 9;
    dyk[ks] = tap / 3 - 1; dxk[ks] = tap % 3 - 1;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      u32x4 w = {0u, 0u, 0u, 0u};
      if (tapok[ks]) w = *reinterpret_cast<const u32x4*>(a.wpk + ((size_t)tap * a.CoutP + sf_co(j, l15 & 3) + 8 * (l15 >> 2)) * a.CinP);
      fw[j][ks] = __builtin_bit_cast(bf16x8, w);
    }
  }
  const bool want = a.stats != nullptr;
  float s1[NT * 4], s2[NT * 4];
#pragma unroll
  for (int e = 0; e < NT * 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

  const int tpr = a.OW / 16;
  constexpr unsigned OOB = 0x80000000u;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  auto offsets = [&](int64_t tt, unsigned (&off)[KS]) {
    const int64_t row = tt / tpr;
    const int ox = (int)(tt - row * tpr) * 16 + l15;
    const int oy = (int)(row % a.OH), b = (int)(row / a.OH);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int iy = 2 * oy + dyk[ks], ix = 2 * ox + dxk[ks];
      const bool ok = tt < a.ntiles && tapok[ks] && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
      off[ks] = ok ? (unsigned)(((b * a.IH + iy) * a.IW + ix) * 16) : OOB;
    }
  };
  u32x4 xa[KS];
  {
    unsigned off[KS];
    offsets(t, off);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xa[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off[ks], 0, 0));
  }
  for (; t < a.ntiles; t += stride) {
    u32x4 xc[KS];
    {
      unsigned off[KS];
      offsets(t + stride, off);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        xc[ks] = xa[ks];
        xa[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off[ks], 0, 0));      // unconditional: exact vmcnt
      }
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[j] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][ks], __builtin_bit_cast(bf16x8, xc[ks]), acc[j], 0, 0, 0);
    }
    bf16_t* orow = a.out + (t * 16 + l15) * CO + grp * 8;
#pragma unroll
    for (int q = 0; q < NT / 2; ++q) {
      const u32x4 o = {f2bf2(acc[2 * q][0], acc[2 * q][1]), f2bf2(acc[2 * q][2], acc[2 * q][3]),
                       f2bf2(acc[2 * q + 1][0], acc[2 * q + 1][1]), f2bf2(acc[2 * q + 1][2], acc[2 * q + 1][3])};
      *reinterpret_cast<u32x4*>(orow + q * 32) = o;
    }
    if (want) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[j][r]; s1[j * 4 + r] += v; s2[j * 4 + r] = fmaf(v, v, s2[j * 4 + r]); }
    }
  }
  if (!want) return;
#pragma unroll
  for (int e = 0; e < NT * 4; ++e) { s1[e] = row16_sum_sf(s1[e]); s2[e] = row16_sum_sf(s2[e]); }
  if (l15 == 0) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sred[wave][0][sf_co(j, r) + grp * 8] = s1[j * 4 + r]; sred[wave][1][sf_co(j, r) + grp * 8] = s2[j * 4 + r]; }
  }
  __syncthreads();
  if (tid < CO) {
    const float u1 = (sred[0][0][tid] + sred[1][0][tid]) + (sred[2][0][tid] + sred[3][0][tid]);
    const float u2 = (sred[0][1][tid] + sred[1][1][tid]) + (sred[2][1][tid] + sred[3][1][tid]);
    float* slot = a.stats + (size_t)(blockIdx.x % RSSF_BN_SLOTS) * 2 * CO;
    atomicAdd(slot + tid, u1);
    atomicAdd(slot + CO + tid, u2);
  }
}

}  // namespace

namespace rssf { namespace cv {

bool stem_fwd_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx) {
  if (Cin != 8 || Cout != 64 || mul != 2 || div != 1 || ntaps != 9 || IH != 2 * OH || IW != 2 * OW || (OW % 16) != 0) return false;
  for (int t = 0; t < 9; ++t)
    if (dy[t] != t / 3 - 1 || dx[t] != t % 3 - 1) return false;
  return (int64_t)B * IH * IW * 16 < ((int64_t)1 << 31) && (int64_t)B * OH * OW * 64 < ((int64_t)1 << 30);
}

int launch_stem_fwd(const void* in, const void* wpk, void* out, float* stats, int B, int IH, int IW, int OH, int OW, int CinP, int CoutP, hipStream_t st) {
  StemFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.in = (const bf16_t*)in; a.wpk = (const bf16_t*)wpk; a.out = (bf16_t*)out; a.stats = stats;
  a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.CoutP = CoutP; a.CinP = CinP;
  a.ntiles = (int64_t)B * OH * OW / 16;
  int64_t blocks = (a.ntiles + 3) / 4;
  if (blocks > 512) blocks = 512;                         // (conv_pw.hip: more blocks were slower for these streams)
  conv_stem_fwd_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(a);
  return check_launch("conv_stem_fwd");
}

} }
