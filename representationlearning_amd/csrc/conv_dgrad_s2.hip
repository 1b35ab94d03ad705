// Data gradient of a 3 x 3 / stride-2 / padding-1 convolution as a STREAM over its output-gradient rows (gfx950 MFMA): the stem's second
// convolution (reference _hrnet_rssformer.py:409-413, 443-447: Conv2d(64, 64, 3, stride 2, padding 1) on the 256 x 256 map - the last big
// launch of the step's backward, nothing runs beside it) with the BatchNorm-backward statistics of the layer before it in the epilogue
// (rssf_conv_gather_bnbwd).
//
//   dx[iy][ix][ci] = sum over the kernel positions (ky, kx) with (iy + 1 - ky), (ix + 1 - kx) even of
//                    dout[(iy + 1 - ky) / 2][(ix + 1 - kx) / 2][co] * W[co][ci][ky][kx]
//
// In the generic gather kernel (conv_fwd.hip, DIV) a tile of 128 input pixels drops the taps of the wrong ROW parity and stages the rest
// with half of their lanes masked (the column parity): 4.5 K-steps x 2 chunks of gathers, LDS stagings and barriers per tile, 8 192 tiles:
// 212 us for a pass that reads 34 + 134 MB and writes 134 MB.  By parity the problem is four dense ones on the grid of dout: around the
// output-gradient pixel (i, j)
//   dx(2i,   2j  ) = d(i,j) W11                                   dx(2i,   2j+1) = d(i,j) W12 + d(i,j+1) W10
//   dx(2i+1, 2j  ) = d(i,j) W21 + d(i+1,j) W01                    dx(2i+1, 2j+1) = d(i,j) W22 + d(i,j+1) W20 + d(i+1,j) W02 + d(i+1,j+1) W00
// so a wave takes 16 consecutive (i, j..j+15): the four shifted dout tiles are coalesced-enough 16-byte loads that ARE the MFMA operands
// (lane = pixel x 8-channel group, as in conv_pw.hip; out-of-range rows / columns load zeros), the nine weight slabs sit in LDS as bf16
// (shared by the block's eight waves, read as fragments), and the MFMAs form the TRANSPOSED results: a lane holds four consecutive
// input channels of one dx pixel per fragment, and the rows of a fragment PAIR are a permutation of 32 channels (the slabs are staged in
// that row order) that makes a lane's 4 + 4 values eight consecutive channels (conv_pw.hip): 16-byte stores, and the addend and the
// producer's raw values in 16-byte pieces of the same layout.
// The next tile's loads are in flight under the current tile's 72 MFMAs.  No masks, no staging, no barrier in the loop.
#include <cstring>
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace {

struct S2Args {
  const bf16_t* dout;      // [B][OH][OW][CO]   (the convolution's output gradient)
  const bf16_t* wpk;       // transposed pack [9][CiP][CoP]: slab t = kernel position (t / 3, t % 3), rows = input channels, K = output channels
  bf16_t* dx;              // [B][2 OH][2 OW][CI]
  const bf16_t* addend;    // added before rounding (an accumulating consumer: may be dx itself), or null
  const bf16_t* bn_raw; const bf16_t* bn_res; const float* bn_ss; float* bn_sums; int bn_act;      // BNB
  int B, OH, OW, CiP, CoP;
  int64_t ntiles;          // B * OH * OW / 16
};

// channel (less 8 grp) of value r of result fragment n; LDS row of input channel ci inside a slab (conv_pw.hip's pairing)
__device__ __forceinline__ constexpr int s2_co(int n, int r) { return 32 * (n >> 1) + 4 * (n & 1) + r; }
__device__ __forceinline__ int s2_row(int ci) { return ((ci >> 5) * 2 + ((ci >> 2) & 1)) * 16 + ((ci >> 3) & 3) * 4 + (ci & 3); }

__device__ __forceinline__ float row16_sum_s2(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// KS = output channels of the convolution / 32 (K-steps), NT = its input channels / 16 (result fragments)
template <int KS, int NT, bool BNB>
__global__ void __launch_bounds__(512) conv_dgrad_s2_kernel(S2Args a) {
  constexpr int CO = 32 * KS, CI = 16 * NT, LDW = CO + 8, NW = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* WS = reinterpret_cast<bf16_t*>(smem);                               // [9][CI][LDW]
  float* sred = reinterpret_cast<float*>(smem + (size_t)9 * CI * LDW * 2);    // [NW][2][CI]
  float* sss = sred + NW * 2 * CI;                                            // [2][CI] scale / shift (BNB)
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the nine slabs: 16-byte pieces, (tap, ci) rows of CO channels
  for (int i = tid; i < 9 * CI * (CO / 8); i += 512) {
    const int row = i / (CO / 8), c8 = (i % (CO / 8)) * 8, t = row / CI, ci = row % CI;
    *reinterpret_cast<u32x4*>(WS + (size_t)(t * CI + s2_row(ci)) * LDW + c8) = *reinterpret_cast<const u32x4*>(a.wpk + ((size_t)t * a.CiP + ci) * a.CoP + c8);
  }
  if constexpr (BNB) {
    if (tid < 2 * CI) sss[tid] = a.bn_ss[tid];
  }
  __syncthreads();

  const int IW = 2 * a.OW;
  const int tpr = a.OW / 16;                                    // tiles per dout row
  const int64_t dbytes = (int64_t)a.B * a.OH * a.OW * CO * 2;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dout), 0, (int)dbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int64_t stride = (int64_t)gridDim.x * NW;
  int64_t t = (int64_t)blockIdx.x * NW + wave;
  // the four shifted tiles of a tile index: byte offsets of this lane's first piece (pixel j0 + l15 (+1), channels 8 grp ..), or OOB
  auto offsets = [&](int64_t tt, unsigned (&off)[4]) {
    const int jt = (int)(tt % tpr);
    const int64_t r = tt / tpr;                                 // global dout row (b * OH + i)
    const int i = (int)(r % a.OH);
    const int j = jt * 16 + l15;
    const bool live = tt < a.ntiles;
    const unsigned base = (unsigned)((r * a.OW + j) * (CO * 2) + grp * 16);
    const bool jn = j + 1 < a.OW, in = i + 1 < a.OH;
    off[0] = live ? base : OOB;                                             // d(i, j)
    off[1] = live && jn ? base + CO * 2 : OOB;                              // d(i, j+1)
    off[2] = live && in ? base + (unsigned)a.OW * CO * 2 : OOB;             // d(i+1, j)
    off[3] = live && in && jn ? base + (unsigned)a.OW * CO * 2 + CO * 2 : OOB;      // d(i+1, j+1)
  };
  u32x4 xa[4][KS];
  {
    unsigned off[4];
    offsets(t, off);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int k = 0; k < KS; ++k) xa[p][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, off[p], k * 64, 0));
  }
  float s1[NT * 4], s2[NT * 4];
#pragma unroll
  for (int e = 0; e < NT * 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

  for (; t < a.ntiles; t += stride) {
    u32x4 xc[4][KS];
    {
      unsigned off[4];
      offsets(t + stride, off);
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          xc[p][k] = xa[p][k];
          xa[p][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, off[p], k * 64, 0));      // unconditional: exact vmcnt
        }
    }
    // (the weight fragments are the same LDS words for every tile: left alone the compiler hoists all 72 reads - 288 registers - out of the
    // loop and spills; an opaque per-iteration offset keeps them where they are used)
    int wlane = (l15 * LDW + grp * 8) * 2;
    asm volatile("" : "+v"(wlane));
    const char* wsl = reinterpret_cast<const char*>(WS) + wlane;
    const int jt = (int)(t % tpr);
    const int64_t r = t / tpr;
    const int i = (int)(r % a.OH), b = (int)(r / a.OH);
    const int j = jt * 16 + l15;
    // dx pixel (2 i + pi, 2 j + pj): element offset of this lane's 8 channels of fragment pair 0
    const int64_t px00 = (((int64_t)b * 2 * a.OH + 2 * i) * IW + 2 * j) * CI + grp * 8;
    // one parity class: acc = sum over its (tap, shifted tile) pairs; store; statistics
    auto cls = [&](int pi, int pj, auto... pairs) {            // pairs: std::integral_constant<int, 4 * tap + shifted tile>: compile-time indices
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = {0.f, 0.f, 0.f, 0.f};
      auto one = [&](auto TP) {
        constexpr int tap = decltype(TP)::value >> 2, p = decltype(TP)::value & 3;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const bf16x8 fb = __builtin_bit_cast(bf16x8, xc[p][k]);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const bf16x8 fw = *reinterpret_cast<const bf16x8*>(wsl + ((tap * CI + n * 16) * LDW + k * 32) * 2);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fb, acc[n], 0, 0, 0);
          }
        }
      };
      (one(pairs), ...);
      const int64_t po = px00 + ((int64_t)pi * IW + pj) * CI;
      if (a.addend) {                                           // block-uniform
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
          const u32x4 av = *reinterpret_cast<const u32x4*>(a.addend + po + q * 32);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4& c = acc[2 * q + h];
            c[0] += __uint_as_float(av[2 * h] << 16); c[1] += __uint_as_float(av[2 * h] & 0xffff0000u);
            c[2] += __uint_as_float(av[2 * h + 1] << 16); c[3] += __uint_as_float(av[2 * h + 1] & 0xffff0000u);
          }
        }
      }
      u32x4 rawv[BNB ? NT / 2 : 1], resv[BNB ? NT / 2 : 1];
      if constexpr (BNB) {
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
          rawv[q] = *reinterpret_cast<const u32x4*>(a.bn_raw + po + q * 32);
          if (a.bn_res) resv[q] = *reinterpret_cast<const u32x4*>(a.bn_res + po + q * 32);
        }
      }
#pragma unroll
      for (int q = 0; q < NT / 2; ++q) {
        const u32x4 o = {f2bf2(acc[2 * q][0], acc[2 * q][1]), f2bf2(acc[2 * q][2], acc[2 * q][3]),
                         f2bf2(acc[2 * q + 1][0], acc[2 * q + 1][1]), f2bf2(acc[2 * q + 1][2], acc[2 * q + 1][3])};
        *reinterpret_cast<u32x4*>(a.dx + po + q * 32) = o;
        if constexpr (BNB) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n = 2 * q + h;
            const f32x4 bsc4 = *reinterpret_cast<const f32x4*>(sss + s2_co(n, 0) + grp * 8), bsh4 = *reinterpret_cast<const f32x4*>(sss + CI + s2_co(n, 0) + grp * 8);
            auto accumulate = [&](auto ACT) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned ow = o[2 * h + (e >> 1)], rw = rawv[q][2 * h + (e >> 1)];
                const float g = (e & 1) ? __uint_as_float(ow & 0xffff0000u) : __uint_as_float(ow << 16);
                const float x = (e & 1) ? __uint_as_float(rw & 0xffff0000u) : __uint_as_float(rw << 16);
                float z = fmaf(x, bsc4[e], bsh4[e]);
                if (a.bn_res) { const unsigned pw = resv[q][2 * h + (e >> 1)]; z += (e & 1) ? __uint_as_float(pw & 0xffff0000u) : __uint_as_float(pw << 16); }
                const float dz = decltype(ACT)::value == 1 ? (z > 0.f ? g : 0.f) : decltype(ACT)::value == 2 ? g * gelu_erf_grad(z) : g;
                s1[n * 4 + e] += dz; s2[n * 4 + e] = fmaf(dz, x, s2[n * 4 + e]);
              }
            };
            if (a.bn_act == 1) accumulate(std::integral_constant<int, 1>{});
            else if (a.bn_act == 2) accumulate(std::integral_constant<int, 2>{});
            else accumulate(std::integral_constant<int, 0>{});
          }
        }
      }
    };
#define S2P(tap, p) std::integral_constant<int, 4 * (tap) + (p)>{}      // (kernel position ky * 3 + kx, shifted tile)
    cls(0, 0, S2P(4, 0));
    cls(0, 1, S2P(5, 0), S2P(3, 1));
    cls(1, 0, S2P(7, 0), S2P(1, 2));
    cls(1, 1, S2P(8, 0), S2P(6, 1), S2P(2, 2), S2P(0, 3));
#undef S2P
  }
  if constexpr (!BNB) return;
  // fold: the 16 pixel lanes of a row, the block's waves, one atomic per channel and sum (conv_pw.hip)
#pragma unroll
  for (int e = 0; e < NT * 4; ++e) { s1[e] = row16_sum_s2(s1[e]); s2[e] = row16_sum_s2(s2[e]); }
  if (l15 == 0) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) { sred[(wave * 2 + 0) * CI + s2_co(n, q) + grp * 8] = s1[n * 4 + q]; sred[(wave * 2 + 1) * CI + s2_co(n, q) + grp * 8] = s2[n * 4 + q]; }
  }
  __syncthreads();
  if (tid < 2 * CI) {
    const int which = tid / CI, c = tid % CI;
    float u = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) u += sred[(w * 2 + which) * CI + c];
    atomicAdd(a.bn_sums + ((size_t)(blockIdx.x % RSSF_BN_BWD_SLOTS) * 2 + which) * CI + c, u);
  }
}

template <int KS, int NT>
constexpr size_t s2_lds() { return (size_t)9 * (16 * NT) * (32 * KS + 8) * 2 + (size_t)(8 * 2 + 2) * (16 * NT) * 4; }

template <int KS, int NT, bool BNB>
int s2_launch(const S2Args& a, int blocks, hipStream_t st) {
  static hipError_t e = hipFuncSetAttribute((const void*)conv_dgrad_s2_kernel<KS, NT, BNB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s2_lds<KS, NT>());
  if (e != hipSuccess) { set_error("conv_dgrad_s2: cannot raise the LDS limit: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  conv_dgrad_s2_kernel<KS, NT, BNB><<<dim3((unsigned)blocks), 512, s2_lds<KS, NT>(), st>>>(a);
  return check_launch("conv_dgrad_s2");
}

// (output-gradient channels, dx channels) with an instantiation: 9 weight slabs of [dx channels][output-gradient channels] must fit the LDS
bool s2_shape(int cin, int cout) {
  return (cin == 64 && cout == 64) || (cin == 64 && cout == 32) || (cin == 32 && cout == 32) || (cin == 128 && cout == 32) || (cin == 32 && cout == 64);
}

}  // namespace

namespace rssf { namespace cv {

// the data gradient of a 3x3 / stride-2 / padding-1 convolution in conv_gather_impl's terms: `in` = the output gradient [B, IH, IW, Cin],
// `out` = dx [B, 2 IH, 2 IW, Cout], mul 1, div 2, the nine mirrored taps.  The stem's convolution (64 -> 64) and the down-sampling fuse
// convolutions of the HighResolutionModules whose weight slabs fit (reference _hrnet_rssformer.py:380-405)
bool dgrad_s2_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps, const int* dy, const int* dx) {
  if (mul != 1 || div != 2 || ntaps != 9 || OH != 2 * IH || OW != 2 * IW || (IW % 16) != 0 || !s2_shape(Cin, Cout)) return false;
  for (int t = 0; t < 9; ++t)
    if (dy[t] != 1 - t / 3 || dx[t] != 1 - t % 3) return false;
  return (int64_t)B * OH * OW * Cout < ((int64_t)1 << 30) && (int64_t)B * IH * IW * Cin * 2 < ((int64_t)1 << 31);
}

int launch_dgrad_s2(const void* dout, const void* wpk, void* dx, const void* addend, const void* bn_raw, const void* bn_res, const float* bn_ss,
                    float* bn_sums, int bn_act, int B, int IH, int IW, int Cin, int Cout, int CinP, int CoutP, hipStream_t st) {
  S2Args a;
  memset(&a, 0, sizeof(a));
  a.dout = (const bf16_t*)dout; a.wpk = (const bf16_t*)wpk; a.dx = (bf16_t*)dx; a.addend = (const bf16_t*)addend;
  a.bn_raw = (const bf16_t*)bn_raw; a.bn_res = (const bf16_t*)bn_res; a.bn_ss = bn_ss; a.bn_sums = bn_sums; a.bn_act = bn_act;
  a.B = B; a.OH = IH; a.OW = IW; a.CiP = CoutP; a.CoP = CinP;       // (the pack's rows = dx channels, its K = output-gradient channels)
  a.ntiles = (int64_t)B * IH * IW / 16;
  static const int cus = [] { int d = 0, v = 256; (void)hipGetDevice(&d); if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256; return v; }();
  // one resident workgroup per CU (its weights live in LDS), persistent; a small map gets fewer workgroups so that a wave still walks
  // >= 2 tiles behind its share of the weight fill
  int64_t blocks = (a.ntiles + 15) / 16;
  if (blocks > cus) blocks = cus;
  if (blocks < 1) blocks = 1;
  const bool bnb = bn_sums != nullptr;
  const int nb = (int)blocks;
  if (Cin == 64 && Cout == 64) return bnb ? s2_launch<2, 4, true>(a, nb, st) : s2_launch<2, 4, false>(a, nb, st);
  if (Cin == 64 && Cout == 32) return bnb ? s2_launch<2, 2, true>(a, nb, st) : s2_launch<2, 2, false>(a, nb, st);
  if (Cin == 32 && Cout == 32) return bnb ? s2_launch<1, 2, true>(a, nb, st) : s2_launch<1, 2, false>(a, nb, st);
  if (Cin == 128 && Cout == 32) return bnb ? s2_launch<4, 2, true>(a, nb, st) : s2_launch<4, 2, false>(a, nb, st);
  if (Cin == 32 && Cout == 64) return bnb ? s2_launch<1, 4, true>(a, nb, st) : s2_launch<1, 4, false>(a, nb, st);
  set_error("conv_dgrad_s2: no kernel for %d -> %d channels", Cin, Cout);
  return RSSF_ERR_UNSUPPORTED;
}

} }
