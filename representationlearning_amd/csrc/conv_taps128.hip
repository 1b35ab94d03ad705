// Many-tap 128 -> 128 channel convolution on channels-last bf16 activations with the PIXEL operand in registers (gfx950 MFMA):
// MlpDWBN's fused {1x1 + 3x3 dil 6 + 3x3 dil 12} sum (ffn_block.py:226-228, 250-257), forward and data gradient - 17 taps of a
// [262 144 x 128] x [128 x 128] product at the benchmark geometry, 16 launches and 46 % of the FLOPs of a training step.
//
// Why not the gather kernel (conv_fwd.hip): at N = 128 every activation element feeds only 256 FLOPs, so staging the pixel tiles
// through LDS costs 32 bytes per clock and CU at the MFMA peak - and ds_write_b128 moves 79 (MI355X_MICROARCH.md, LDS): writes +
// fragment reads of the pixel operand alone are as long as the MFMA work, behind one barrier per 32 MFMAs.  Loading the operand
// straight into fragment layout does not work either: a fragment's 16 lanes of one k-group are 16 different pixels = 16 different
// cache lines per quarter wave, and the texture addresser then takes ~83 cycles per load (measured, tools/mlp_direct_proto.hip:
// 150 us of loads alone).
//
// Here a wave owns 64 consecutive pixels of one image row and ALL 128 output channels (a 64 x 128 tile, 128 accumulator registers):
//   * per tap and 16-pixel tile FOUR coalesced 1 KB loads (load J: pixels 4J..4J+3, each pixel's whole 256-byte channel row;
//     lane = [k5 k4 | p1 p0 | k3 k2], 16 bytes = dwords [k1 k0]) land in registers;
//   * two butterfly stages on the VALU (lane bit 1 <-> J bit 1, lane bit 0 <-> J bit 0: v_mov_b32_dpp quad_perm + v_cndmask per
//     dword) turn them into lane = [k5 k4 | p1 p0 p3 p2], register J' = [k3 k2]: register J' IS the MFMA operand of K-step J'
//     (channels 32 G + 8 J' + e for lane group G = [k5 k4]; the weight fragments are read to match);
//   * the image row of a tap is a buffer descriptor of its own (base = row start, num_records = row bytes, 0 when the row lies
//     outside the image): x + dx outside the row is answered with zeros by the hardware range check - no masks, no selects, one
//     scalar row computation per tap and wave;
//   * only the WEIGHTS go through LDS: one [128][128] tile per tap (36 KB, double-buffered), ONE barrier per tap = per 128 MFMAs
//     of a wave; rows padded to 288 bytes with the 16-byte chunks ordered 4 kappa + G so that a fragment read is
//     lane_const + ct * 4608 + kappa * 64 bytes and conflict-free (the 16-lane service groups of ds_read_b128 pair rows {0-3,12-15}
//     of G with rows {4-11} of G ^ 1: chunk slots 2 r + G mod 16 are the evens / the odds);
//   * the products run transposed (rows = output channels, columns = pixels): a lane holds four consecutive channels of one
//     pixel, the epilogue passes 16 pixels at a time through a wave-private LDS tile into 16-byte row stores, with bias, addend,
//     the BatchNorm statistics of the output (forward) or the producer's BatchNorm-backward statistics (rssf_conv_gather_bnbwd)
//     on the values being stored.
// Requests are pinned with sched_barriers (left alone the scheduler sinks every load in front of its use and the prefetch distance
// collapses) and ordered alike in the prologue and the loop so that the wait counts the compiler derives at the loop header are exact.
// Measured stand-alone at 16 x 128 x 128 x 128 (tools/mlp_direct_proto.hip, same process, same buffers): 145 us against 165 us of
// the gather kernel; without the transposes 127 us, without activation loads 118 us, without MFMAs 97 us.
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace rssf {
namespace cv {

namespace {

constexpr int T_C = 128;                      // channels in and out
constexpr int T_NW = 8;                       // waves per workgroup (64 pixels each)
constexpr int T_PITCH = 144;                  // weight row pitch in LDS (elements): 256 + 32 bytes
constexpr int T_WELEMS = 128 * T_PITCH;
#ifndef RSSF_T128_PAIRS
#define RSSF_T128_PAIRS 1                     // 1: "pair" loads (two lanes per pixel and load, one exchange stage); 0: quad loads, two stages
#endif

constexpr int T_MAXS = 40;                    // K-steps of a launch: (tap, 128-channel chunk of the input) pairs
struct T128Args {
  const bf16_t* in; const bf16_t* wpk; bf16_t* out; const float* bias; float* stats; const bf16_t* addend;
  const bf16_t* bn_raw; const bf16_t* bn_res; const float* bn_ss; float* bn_sums; int bn_act;
  int B, H, W, nsteps, per;
  int Cin, Cout, CinP, CoutP, ntn;            // general form: channels of a pixel row in / out, packed slab sizes, 128-channel output tiles
  int dy[T_MAXS], dx[T_MAXS];                 // per K-step: the tap's displacement,
  int cin0[T_MAXS], woff[T_MAXS];             // first input channel of the chunk, BYTE offset of slab element (row 0, that channel)
};

template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// exchange the lane bit selected by CTRL (quad_perm xor 1 / xor 2) with the register bit that tells a from b
template <int CTRL> __device__ __forceinline__ void lane_exchange(u32x4& a, u32x4& b, bool hi) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t ta = dpp_u32<CTRL>(b[d]), tb = dpp_u32<CTRL>(a[d]);
    const uint32_t na = hi ? ta : a[d], nb = hi ? b[d] : tb;
    a[d] = na; b[d] = nb;
  }
}

// LP: pair loads - lane = [k5 k4 | p2 p1 p0 | k2], load J = [p3 k3]: one exchange stage (lane bit 0 <-> p3) at twice the cache lines per
// load instruction.  BNB: the BatchNorm-backward statistics epilogue (data-gradient launches).
// GEN: the general form - any number of input / output channels (multiples of 32 / 8): a K-step is a (tap, 128-channel chunk of the
// input) pair, a workgroup computes one 128-channel tile of the output (the tiles of a pixel range are neighbours on one XCD); past
// the last input channel the slab columns are staged as zeros and the pixel operand reads the next pixel's (finite) values or - at
// the end of the image row - the range check's zeros.  !GEN: 128 -> 128 channels with compile-time strides (MlpDWBN).
template <int LP, bool BNB, bool GEN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_taps128_kernel(T128Args a) {
  constexpr int C = T_C, NW = T_NW;
  const int CI = GEN ? a.Cin : C, CO = GEN ? a.Cout : C, CIP = GEN ? a.CinP : C, COP = GEN ? a.CoutP : C;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * T_WELEMS];
  __shared__ float sred[NW][2][C];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned q = (blockIdx.x & 7u) * (unsigned)a.per + (blockIdx.x >> 3);      // XCD-major: neighbouring pixel tiles share an L2
  const int M = a.B * a.H * a.W;
  const unsigned qm = GEN ? q / (unsigned)a.ntn : q;
  const int n0 = GEN ? (int)(q - qm * (unsigned)a.ntn) * C : 0;   // first output channel of this workgroup
  const int m0 = (int)qm * (NW * 64);
  if (m0 >= M) return;
  const int mw = m0 + wave * 64;                                  // wave-uniform: 64 consecutive pixels of one image row (W % 64 == 0)
  const int x0 = mw % a.W, yrow = (mw / a.W) % a.H, img = mw / (a.W * a.H);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wpk), 0, 0x7fffffff, 0x00020000);
  const unsigned lane_base = LP ? (unsigned)((x0 + ((lane >> 1) & 7)) * (CI * 2) + grp * 64 + (lane & 1) * 16)
                                : (unsigned)((x0 + ((lane >> 2) & 3)) * (CI * 2) + grp * 64 + (lane & 3) * 16);
  // weight staging: thread -> row tid / 16 (+ 32 i), LDS chunk position qp = tid % 16 = 4 kappa + G, i.e. global chunk 4 G + kappa
  const int qp = tid & 15;
  const int gq8 = (((qp & 3) << 2) | (qp >> 2)) << 3;            // first channel (within the 128-channel chunk) of this thread's slab chunk
  const unsigned bsrc = (unsigned)((n0 + (tid >> 4)) * (CIP * 2) + gq8 * 2);
  const int bdst = (tid >> 4) * T_PITCH + qp * 8;
  const int foff = l15 * T_PITCH + grp * 8;                       // + ct * 16 * T_PITCH + kappa * 32 elements

  f32x4 acc[4][8];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[mi][ct] = {0.f, 0.f, 0.f, 0.f};
  u32x4 RA[4][4];                  // [tile mi][load J]; after the exchanges [mi][register of a K-step]
  u32x4 RB[4];
  // descriptor of the image row K-step t reads for this wave, from the chunk's first channel on (scalar arithmetic): a pixel past the
  // end of the row - or, GEN, a channel past the end of its last pixel - is out of range
  auto row_rsrc = [&](int t) {
    const int ts = t < a.nsteps ? t : 0;
    const int r = yrow + a.dy[ts], c0 = GEN ? a.cin0[ts] : 0;
    const bool ok = t < a.nsteps && r >= 0 && r < a.H;
    const bf16_t* p = a.in + (size_t)((img * a.H + (ok ? r : 0)) * a.W) * CI + c0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p), 0, ok ? (a.W * CI - c0) * 2 : 0, 0x00020000);
  };
  auto load_A = [&](int mi, const __amdgpu_buffer_rsrc_t& rs, int dxb) {
    const unsigned v = lane_base + (unsigned)(mi * 16 * CI * 2 + dxb);            // may wrap below zero: out of range -> zeros
#pragma unroll
    for (int J = 0; J < 4; ++J)
      RA[mi][J] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, v + (unsigned)(LP ? (J >> 1) * 8 * CI * 2 + (J & 1) * 32 : J * 4 * CI * 2), 0, 0));
  };
  auto load_B = [&](int t) {       // K-steps past the end, slab columns past the last input channel: zeros
    const int ts = t < a.nsteps ? t : 0;
    const int woff = GEN ? a.woff[ts] : ts * (C * C * 2);
    const bool ok = t < a.nsteps && (!GEN || a.cin0[ts] + gq8 < CIP);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      RB[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? bsrc : 0x80000000u, woff + i * 32 * CIP * 2, 0));
  };
  auto store_B = [&](bf16_t* Bs) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(Bs + bdst + i * 32 * T_PITCH) = RB[i];
  };
  // Pipeline.  Weights: tile t+2 is requested at the start of tap t, written to the other LDS buffer at the start of tap t+1 (after the
  // barrier that retires that buffer's readers), read as fragments in tap t+2: one register set.  Pixels: the two tiles of a pair are
  // re-requested for the next tap right after the pair's 64 MFMAs.  The request order of the prologue is the loop's.
  load_B(0);
  store_B(lds);
  load_B(1);
  __builtin_amdgcn_sched_barrier(0);
  {
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(0);
    const int dxb = a.dx[0] * (CI * 2);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) load_A(mi, rs, dxb);
  }
  __builtin_amdgcn_sched_barrier(0);

  const bool hi1 = lane & 2, hi0 = lane & 1;
  for (int t = 0; t < a.nsteps; ++t) {
    const bf16_t* Bs = lds + (t & 1) * T_WELEMS;
    __syncthreads();
    store_B(lds + ((t + 1) & 1) * T_WELEMS);
    load_B(t + 2);
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(t + 1);
    const int dxb = (t + 1 < a.nsteps ? a.dx[t + 1] : 0) * (CI * 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
      for (int mi = 2 * pr; mi < 2 * pr + 2; ++mi) {
        if (LP) {
          lane_exchange<0xB1>(RA[mi][0], RA[mi][2], hi0);
          lane_exchange<0xB1>(RA[mi][1], RA[mi][3], hi0);
        } else {
          lane_exchange<0x4E>(RA[mi][0], RA[mi][2], hi1);
          lane_exchange<0x4E>(RA[mi][1], RA[mi][3], hi1);
          lane_exchange<0xB1>(RA[mi][0], RA[mi][1], hi0);
          lane_exchange<0xB1>(RA[mi][2], RA[mi][3], hi0);
        }
      }
#pragma unroll
      for (int kp = 0; kp < 4; ++kp)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          bf16x8 fb[4];
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) fb[c4] = *reinterpret_cast<const bf16x8*>(Bs + foff + (hf * 4 + c4) * 16 * T_PITCH + kp * 32);
          const int rg = LP ? ((kp & 1) << 1) | (kp >> 1) : kp;        // the register that holds K-step kp
#pragma unroll
          for (int mi = 2 * pr; mi < 2 * pr + 2; ++mi)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
              acc[mi][hf * 4 + c4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[c4], __builtin_bit_cast(bf16x8, RA[mi][rg]), acc[mi][hf * 4 + c4], 0, 0, 0);
        }
      load_A(2 * pr, rs, dxb);
      load_A(2 * pr + 1, rs, dxb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                 // the weight buffers become the waves' output tiles

  // ---- epilogue -------------------------------------------------------------------------------------------------------------------
  // column `col` of an accumulator tile is pixel prow(col) of the 16-pixel tile; a lane holds channels ct * 16 + grp * 4 + r
  bf16_t* Cs = lds + wave * (16 * 136);
  const int cc = lane & 15;                 // this lane's 16-byte channel chunk in the store phase
  const int prow = LP ? 8 * (l15 & 1) + (l15 >> 1) : 4 * (l15 & 3) + (l15 >> 2);
  const bool want = BNB || a.stats != nullptr;
  float s1[8], s2[8], bsc[8], bsh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s1[e] = 0.f; s2[e] = 0.f;
    const bool cok = BNB && n0 + cc * 8 + e < CO;
    bsc[e] = cok ? a.bn_ss[n0 + cc * 8 + e] : 0.f;
    bsh[e] = cok ? a.bn_ss[CO + n0 + cc * 8 + e] : 0.f;
  }
  const bool cok8 = n0 + cc * 8 < CO;         // this lane's channel chunk exists (Cout is a multiple of 8)
  f32x4 bv[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct)
    bv[ct] = (a.bias && n0 + ct * 16 + grp * 4 < CO) ? *reinterpret_cast<const f32x4*>(a.bias + n0 + ct * 16 + grp * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    // the rows this lane stores: pixels (lane >> 4) + 4 j of the tile; their addend / raw / residual rows are requested first
    Vec<bf16_t> va[4], xr[4], xp[4];
    const size_t mrow = (size_t)(mw + mi * 16 + (lane >> 4)) * CO + (cok8 ? n0 + cc * 8 : 0);       // (a chunk past Cout: loaded from channel 0, never stored)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.addend) va[j].load(a.addend + mrow + (size_t)j * 4 * CO);
      if (BNB) {
        xr[j].load(a.bn_raw + mrow + (size_t)j * 4 * CO);
        if (a.bn_res) xp[j].load(a.bn_res + mrow + (size_t)j * 4 * CO);
      }
    }
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const f32x4 v = acc[mi][ct] + bv[ct];
      typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
      const u32x2_t pk = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(Cs + prow * 136 + ct * 16 + grp * 4) = pk;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = (lane >> 4) + j * 4;
      Vec<bf16_t> v;
      v.load(Cs + px * 136 + cc * 8);
      if (a.addend) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v.get(e) + va[j].get(e);
        v.set_all(o);
      }
      if (!GEN || cok8) v.store(a.out + mrow + (size_t)j * 4 * CO);
      if (BNB) {                             // on the values just stored: what rssf_bn_bwd_reduce would read
        auto accumulate = [&](auto ACT) {    // block-uniform activation: one specialised loop runs
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = xr[j].get(e);
            float z = fmaf(x, bsc[e], bsh[e]);
            if (a.bn_res) z += xp[j].get(e);
            const float g = v.get(e);
            const float dz = decltype(ACT)::value == 1 ? (z > 0.f ? g : 0.f) : decltype(ACT)::value == 2 ? g * gelu_erf_grad(z) : g;
            s1[e] += dz; s2[e] = fmaf(dz, x, s2[e]);
          }
        };
        if (a.bn_act == 1) accumulate(std::integral_constant<int, 1>{});
        else if (a.bn_act == 2) accumulate(std::integral_constant<int, 2>{});
        else accumulate(std::integral_constant<int, 0>{});
      } else if (want) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = v.get(e); s1[e] += f; s2[e] = fmaf(f, f, s2[e]); }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (want) {
    // the four lanes of a channel chunk (one per 16-lane row), then the waves in a fixed order, one atomic per channel, sum and block
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = rows_reduce<OpSum>(s1[e]); s2[e] = rows_reduce<OpSum>(s2[e]); }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sred[wave][0][cc * 8 + e] = s1[e]; sred[wave][1][cc * 8 + e] = s2[e]; }
    }
    __syncthreads();
    if (tid < 2 * C && n0 + tid % C < CO) {
      float tsum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tsum += sred[w][tid / C][tid % C];
      float* dst = BNB ? a.bn_sums + (size_t)(blockIdx.x % RSSF_BN_BWD_SLOTS) * 2 * CO : a.stats + (size_t)(blockIdx.x % RSSF_BN_SLOTS) * 2 * CO;
      atomicAdd(dst + (tid / C) * CO + n0 + tid % C, tsum);
    }
  }
}

}  // namespace

bool taps128_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int mul, int div, int ntaps) {
#ifdef RSSF_T128_DISABLE       // A/B builds only (tools/ab_lib.sh): the step without this kernel
  return false;
#endif
  if (mul != 1 || div != 1 || IH != OH || IW != OW || (IW % 64) != 0 || ((int64_t)B * IH * IW) % (T_NW * 64) != 0) return false;
  if ((int64_t)B * IH * IW * (Cin > Cout ? Cin : Cout) >= ((int64_t)1 << 30)) return false;
  // 128 -> 128 channels (MlpDWBN's sum): at least 8 taps - below that the [128][128] weight tile per barrier does not pay (the 3x3
  // layers take the halo kernel anyway)
  if (Cin == T_C && Cout == T_C) return ntaps >= 8;
  // general form: wide layers only - every output tile re-reads the pixels, and a tile narrower than 96 channels wastes the 64 x 128
  // wave tile (the neck's 480 -> 480 point-wise convolution, hrnet_aux.py:45-49, is the case on the training path)
  const int steps = ntaps * ((Cin + T_C - 1) / T_C);
  return Cin >= T_C && (Cin % 32) == 0 && (Cout % 8) == 0 && Cout >= 96 && (Cout % T_C == 0 || Cout % T_C >= 96) && steps >= 4 && steps <= T_MAXS;
}

int launch_taps128(const void* in, const void* wpk, void* out, const float* bias, float* stats, const void* addend, const void* bn_raw,
                   const void* bn_res, const float* bn_ss, float* bn_sums, int bn_act, int B, int H, int W, int Cin, int Cout, int CinP,
                   int CoutP, int ntaps, const int* dy, const int* dx, hipStream_t st) {
  T128Args a;
  a.in = (const bf16_t*)in; a.wpk = (const bf16_t*)wpk; a.out = (bf16_t*)out; a.bias = bias; a.stats = stats; a.addend = (const bf16_t*)addend;
  a.bn_raw = (const bf16_t*)bn_raw; a.bn_res = (const bf16_t*)bn_res; a.bn_ss = bn_ss; a.bn_sums = bn_sums; a.bn_act = bn_act;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CinP = CinP; a.CoutP = CoutP;
  const int kcn = (Cin + T_C - 1) / T_C;
  a.nsteps = ntaps * kcn;
  a.ntn = (Cout + T_C - 1) / T_C;
  if (a.nsteps > T_MAXS || (int64_t)ntaps * CoutP * CinP * 2 >= ((int64_t)1 << 31)) { set_error("conv_taps128: %d K-steps / slab size out of range", a.nsteps); return RSSF_ERR_UNSUPPORTED; }
  for (int s = 0; s < T_MAXS; ++s) {
    const int t = s < a.nsteps ? s / kcn : 0, kc = s < a.nsteps ? s % kcn : 0;
    a.dy[s] = s < a.nsteps ? dy[t] : 0; a.dx[s] = s < a.nsteps ? dx[t] : 0;
    a.cin0[s] = kc * T_C;
    a.woff[s] = (int)(((int64_t)t * CoutP * CinP + kc * T_C) * 2);
  }
  const bool gen = !(Cin == T_C && Cout == T_C && CinP == T_C && CoutP == T_C);
  const int64_t total = (int64_t)B * H * W / (T_NW * 64) * a.ntn;
  a.per = xcd_per(total);
  const dim3 grid((unsigned)a.per * 8u);
  if (gen) {
    if (bn_sums) conv_taps128_kernel<RSSF_T128_PAIRS, true, true><<<grid, 512, 0, st>>>(a);
    else conv_taps128_kernel<RSSF_T128_PAIRS, false, true><<<grid, 512, 0, st>>>(a);
  } else {
    if (bn_sums) conv_taps128_kernel<RSSF_T128_PAIRS, true, false><<<grid, 512, 0, st>>>(a);
    else conv_taps128_kernel<RSSF_T128_PAIRS, false, false><<<grid, 512, 0, st>>>(a);
  }
  return check_launch("conv_taps128");
}

}  // namespace cv
}  // namespace rssf
