// Weight gradient of the NARROW point-wise convolutions (1x1, stride 1, bf16): MlpDWBN's fc1 / fc2 (reference ffn_block.py:218, 236: 32 <-> 128
// channels at 1/4 resolution), layer1's Bottleneck reductions / expansions (_hrnet_rssformer.py:146-156: 64 <-> 256) and the 1x1 fuse
// convolutions of the HighResolutionModules.
//
//   dW[co][ci] = sum_p dout[p][co] * in[p][ci],   K = B*H*W pixels (262 144 at the benchmark geometry), Cout * Cin <= 16 384
//
// A GEMM that is ALL operand traffic (84 MB of activations for 2 GFLOP): the job is to stream both operands once at the fabric's rate.
// The generic kernel (conv_wgrad.hip) tiles the channels 64 x 64 - for 32 channels half of every staged slab is padding, for 256 the
// narrow operand is re-read four times - keeps one 64-pixel slab per block in flight and ran these layers at 2.4 TB/s (36 launches
// of 35 us per step).  Here a block owns EVERY (co, ci) pair of a pixel range: the two operand chunks of 32 / 64 pixels are plain
// contiguous spans of memory (16-byte loads, thread t takes bytes 16 t, 16 (t + 256), ...), staged pixel-major in LDS, and the MFMA
// fragments come out of gfx950's transposing LDS reads (SlabFrag, see conv_wgrad.hip); the next chunk's loads are in flight
// under the current chunk's MFMAs, several blocks per CU.  Split-K partials + the common second stage as everywhere else.
//
// FUSE: the BatchNorm-backward APPLY of the layer rides in the launch (rssf_conv_wgrad_bnapply): the block owns all channels of its
// pixels, so it computes draw = sc * dz + cb * raw + cc for its chunk exactly once - from dy and raw instead of loading draw -
// writes it out for the data-gradient launch (and dz for a residual branch) and contracts it from LDS.  One tensor pass
// (read dy + raw, write draw) and one launch less per layer: bn_bwd_apply_kernel's arithmetic, bit for bit (common.hip.h).
#include <cstring>
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace {

typedef __attribute__((ext_vector_type(4))) short v4s;

struct PwWgradArgs {
  const bf16_t* dout;     // [M][CO]  (FUSE: not read)
  const bf16_t* in;       // [M][CI]
  float* partial;         // [ksplit][CO][CI]
  float* dbias;           // [CO] or null
  int64_t M;
  int per;                // pixels per block (a multiple of the chunk)
  // fused BatchNorm-backward apply (see WgradHaloArgs in conv_wgrad.hip)
  const bf16_t* bn_dy; const bf16_t* bn_raw; const bf16_t* bn_res;
  const float* bn_ss; const float* bn_mi; const float* bn_sums;
  bf16_t* draw_out; bf16_t* dres_out;
  float* dgamma; float* dbeta;
  float bn_n, bn_pscale;
  int bn_act, bn_training;
  // XPRE: `in` is the RAW output of the producing convolution, the operand contracted is act(in * scale + shift) (x_ss: [2][CI])
  const float* x_ss; int x_act;
  // DG: the layer's DATA gradient in the same launch, dx[p][ci] = sum_co draw[p][co] * W[co][ci] - the draw chunk is in LDS anyway and a
  // 1x1 convolution's data gradient needs nothing else (w: the fp32 master weights [CO][CI], rounded to bf16 as the weight pack does)
  const float* w; bf16_t* dx_out;
  // DGS (with XPRE): the data gradient of a layer whose input is the producer's RAW output, plus that producer's BatchNorm-backward
  // statistics {sum dz, sum dz * raw}, dz = dx * act'(raw * scale + shift) (rssf_conv_gather_bnbwd's epilogue), into st_sums
  float* st_sums;            // [RSSF_BN_BWD_SLOTS][2][CI]
};

// MFMA fragment of a K-step (32 pixel rows from k0) for 16 channels from column c0 of a pixel-major tile: conv_wgrad.hip's SlabFrag
__device__ __forceinline__ bf16x8 frag(const bf16_t* tile, int ld, int k0, int c0, int lane) {
  const int grp = lane >> 4, i = lane & 15;
  const bf16_t* p = tile + (k0 + grp * 4 + (i >> 2)) * ld + c0 + (i & 3) * 4;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 16 * ld));
  union { struct { v4s a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// GM x GN waves (GN = 4 / GM), each WM x WN tiles of 16 x 16: CO = 16 GM WM output channels, CI = 16 GN WN input channels; KPX pixels per chunk
template <int GM, int WM, int WN, int KPX, bool FUSE, bool RES, bool XPRE = false, bool DG = false, bool DGS = false>
__global__ void __launch_bounds__(256) conv_wgrad_pw_kernel(PwWgradArgs a) {
  constexpr int GN = 4 / GM, CO = 16 * GM * WM, CI = 16 * GN * WN;
  constexpr int LDD = CO + 16, LDX = CI + 16;                       // +32 B per row: conflict-free transposing reads (conv_wgrad.hip)
  constexpr int DV = KPX * CO / 8 / 256, XV = KPX * CI / 8 / 256;   // 16-byte vectors per thread and chunk
  static_assert(DV * 256 * 8 == KPX * CO && XV * 256 * 8 == KPX * CI && DV >= 1 && XV >= 1, "the chunk divides over the block exactly");
  static_assert(KPX % 32 == 0 && 256 % (CO / 8) == 0 && 256 % (CI / 8) == 0, "a thread's channel group is the same in every vector");
  __shared__ __attribute__((aligned(16))) bf16_t DS[KPX * LDD];
  __shared__ __attribute__((aligned(16))) bf16_t XS[KPX * LDX];
  __shared__ __attribute__((aligned(16))) float sbn[4][FUSE ? CO : 4];      // scale, shift, cb, cc
  __shared__ float sbias[CO];
  __shared__ __attribute__((aligned(16))) float sxs[2][XPRE ? CI : 4];
  constexpr int LDW = CO + 8;                                       // W^T [ci][co] bf16, 16-byte aligned rows
  static_assert(!DG || (KPX == 64 && CI % 16 == 0 && CO % 32 == 0), "DG: one 16-pixel tile per wave, whole MFMA tiles");
  __shared__ __attribute__((aligned(16))) bf16_t WT[(DG || DGS) ? CI * LDW : 8];
  static_assert(!DGS || (XPRE && !FUSE && !DG && KPX == 64 && CO == 32 && CI % 16 == 0), "DGS: the 32 <- 128 kernel with a pre-activation input");
  constexpr int LDY = CI + 8;                                       // dx tile [px][ci] bf16 (+16 B per row)
  __shared__ __attribute__((aligned(16))) bf16_t YS[DGS ? KPX * LDY : 8];
  static_assert(!DGS || KPX * LDY * 2 >= 2 * 16 * CI * 4, "the statistics fold reuses the dx tile");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wm = wave / GN, wn = wave % GN;
  const int range = blockIdx.x;
  const int64_t kbeg = (int64_t)range * a.per;
  const int64_t kend = kbeg + a.per < a.M ? kbeg + a.per : a.M;
  const bool do_bias = a.dbias != nullptr;
  if (tid < CO) sbias[tid] = 0.f;
  if constexpr (XPRE) {
    for (int c = tid; c < 2 * CI; c += 256) sxs[c / CI][c % CI] = a.x_ss[c];
  }
  if constexpr (DG || DGS) {
    for (int i = tid; i < CO * CI; i += 256) WT[(i % CI) * LDW + i / CI].v = f2bf(a.w[i]);
  }

  if constexpr (FUSE) {
    if (tid < CO) {
      const int c = tid;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < RSSF_BN_BWD_SLOTS; ++k) { s1 += a.bn_sums[(size_t)k * 2 * CO + c]; s2 += a.bn_sums[(size_t)k * 2 * CO + CO + c]; }
      const float mean = a.bn_mi[c], istd = a.bn_mi[CO + c];
      const float sc = a.bn_ss[c], sh = a.bn_ss[CO + c];
      float dot, cb, cc;
      bn_bwd_constants(sc, mean, istd, s1, s2, a.bn_n, dot, cb, cc);
      if (a.dgamma && range == 0) { a.dgamma[c] += dot * a.bn_pscale; a.dbeta[c] += s1 * a.bn_pscale; }       // one writer per channel
      sbn[0][c] = sc; sbn[1][c] = sh; sbn[2][c] = cb; sbn[3][c] = cc;
    }
  }
  __syncthreads();

  // hardware-bounds-checked buffer accesses, 32-bit byte offsets (the entry point keeps the tensors below 2^31 bytes)
  const int dbytes = (int)(a.M * CO * 2), xbytes = (int)(a.M * CI * 2);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(FUSE ? a.bn_dy : a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rraw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(FUSE ? a.bn_raw : a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(FUSE && RES ? a.bn_res : a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdraw = __builtin_amdgcn_make_buffer_rsrc(FUSE ? a.draw_out : const_cast<bf16_t*>(a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdres = __builtin_amdgcn_make_buffer_rsrc(FUSE && a.dres_out ? a.dres_out : (FUSE ? a.draw_out : const_cast<bf16_t*>(a.dout)), 0, dbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 bs2[4];                                    // dbias of the thread's 8 output channels: (e0, e1) .. (e6, e7)
#pragma unroll
  for (int j = 0; j < 4; ++j) bs2[j] = f32x2{0.f, 0.f};

  Vec<bf16_t> rd[DV], rx[XV], rr[FUSE ? DV : 1], rq[FUSE && RES ? DV : 1], vdz[FUSE ? DV : 1], rawk[DGS ? XV : 1];
  float st1[8], st2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }
  unsigned soff[DV];
  auto load_chunk = [&](int64_t k0) {
    const unsigned db = (unsigned)(k0 * CO * 2) + (unsigned)tid * 16u, xb = (unsigned)(k0 * CI * 2) + (unsigned)tid * 16u;
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      rd[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, db + c * 4096u, 0, 0));
      if constexpr (FUSE) {
        rr[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rraw, db + c * 4096u, 0, 0));
        if constexpr (RES) rq[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, db + c * 4096u, 0, 0));
      }
    }
#pragma unroll
    for (int c = 0; c < XV; ++c) rx[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, xb + c * 4096u, 0, 0));
  };
  const int cg8 = (tid % (CO / 8)) * 8;           // this thread's 8 output channels (the same in every vector of every chunk)
  // draw / dz of the staged chunk: block-uniform activation, one specialised loop runs
  auto apply_chunk = [&](auto ACT) {
    float bsc[8], bsh[8], bcb[8], bcc[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&sbn[0][cg8 + 4 * h]), v1 = *reinterpret_cast<const f32x4*>(&sbn[1][cg8 + 4 * h]);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(&sbn[2][cg8 + 4 * h]), v3 = *reinterpret_cast<const f32x4*>(&sbn[3][cg8 + 4 * h]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bsc[4 * h + e] = v0[e]; bsh[4 * h + e] = v1[e]; bcb[4 * h + e] = v2[e]; bcc[4 * h + e] = v3[e]; }
    }
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      float o1[8], o2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = rr[c].get(e);
        float z = fmaf(x, bsc[e], bsh[e]);
        if constexpr (RES) z += rq[c].get(e);
        const float g = rd[c].get(e);
        const float dz = decltype(ACT)::value == 1 ? g * (z > 0.f ? 1.f : 0.f) : decltype(ACT)::value == 2 ? g * gelu_erf_grad(z) : g;
        o2[e] = dz;
        o1[e] = a.bn_training ? fmaf(bsc[e], dz, fmaf(bcb[e], x, bcc[e])) : bsc[e] * dz;
      }
      rd[c].set_all(o1); vdz[c].set_all(o2);
    }
  };
  auto stage = [&](int64_t k0) {                   // registers -> LDS (+ the apply)
    if constexpr (FUSE) {
      if (a.bn_act == 1) apply_chunk(std::integral_constant<int, 1>{});
      else if (a.bn_act == 2) apply_chunk(std::integral_constant<int, 2>{});
      else apply_chunk(std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      const int id = tid + c * 256, row = id / (CO / 8), col = (id % (CO / 8)) * 8;
      rd[c].store(DS + row * LDD + col);
      soff[c] = (unsigned)(k0 * CO * 2) + (unsigned)id * 16u;
      if (do_bias) {
        // the two halves of each 32-bit word as ONE float pair, added as a pair: left to itself the SLP vectoriser paired (e1, e2) of
        // two words into a `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` - the crossed-halves form whose LOW half comes back wrong
        // on gfx950 when another wave of the SIMD issues MFMAs (DESIGN.md lesson 59; tools/pk_crossed_repro.hip)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = rd[c].raw[j];
          bs2[j] += f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
        }
      }
    }
    if constexpr (DGS) {
#pragma unroll
      for (int c = 0; c < XV; ++c) rawk[c] = rx[c];
    }
    if constexpr (XPRE) {                            // the producer's BatchNorm + activation on this thread's 8 input channels
      const int xg8 = (tid % (CI / 8)) * 8;
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(&sxs[0][xg8]), sc1 = *reinterpret_cast<const f32x4*>(&sxs[0][xg8 + 4]);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(&sxs[1][xg8]), sh1 = *reinterpret_cast<const f32x4*>(&sxs[1][xg8 + 4]);
      auto apply = [&](auto ACT) {
#pragma unroll
        for (int c = 0; c < XV; ++c) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z = fmaf(rx[c].get(e), e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
            o[e] = decltype(ACT)::value == 1 ? fmaxf(z, 0.f) : decltype(ACT)::value == 2 ? gelu_erf(z) : z;
          }
          rx[c].set_all(o);
        }
      };
      if (a.x_act == 1) apply(std::integral_constant<int, 1>{});
      else if (a.x_act == 2) apply(std::integral_constant<int, 2>{});
      else apply(std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int c = 0; c < XV; ++c) {
      const int id = tid + c * 256, row = id / (CI / 8), col = (id % (CI / 8)) * 8;
      rx[c].store(XS + row * LDX + col);
    }
  };
  // the chunk's draw / dz go out AFTER the next chunk's loads were issued (vmcnt counts in order: stores in front of those loads
  // would have to complete before the loads can be waited for)
  Vec<bf16_t> kd[FUSE ? DV : 1], kz[FUSE ? DV : 1];
  auto flush_stores = [&]() {
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, kd[c].raw), rdraw, soff[c], 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, kz[c].raw), rdres,
                                             a.dres_out ? soff[c] : OOB, 0, 0);
    }
  };

  if (kbeg < kend) load_chunk(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += KPX) {
    stage(k0);
    if constexpr (FUSE) {
#pragma unroll
      for (int c = 0; c < DV; ++c) { kd[c] = rd[c]; kz[c] = vdz[c]; }
    }
    __syncthreads();
    if (k0 + KPX < kend) load_chunk(k0 + KPX);
    if constexpr (FUSE) flush_stores();
#pragma unroll
    for (int ks = 0; ks < KPX; ks += 32) {
      bf16x8 fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[i] = frag(DS, LDD, ks, (wm * WM + i) * 16, lane);
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[j] = frag(XS, LDX, ks, (wn * WN + j) * 16, lane);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (DG) {
      // dx^T[ci][px] = W^T[ci][co] * draw^T[co][px] for the wave's 16 pixels: A = rows of W^T, B = rows of the draw tile (both K = co
      // contiguous: plain 16-byte LDS reads); the result has a lane's four values = four consecutive ci of ONE pixel: 8-byte stores
      f32x4 dacc[CI / 16];
#pragma unroll
      for (int j = 0; j < CI / 16; ++j) dacc[j] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kq = 0; kq < CO / 32; ++kq) {
        const bf16x8 fp = *reinterpret_cast<const bf16x8*>(DS + (wave * 16 + l15) * LDD + kq * 32 + grp * 8);
#pragma unroll
        for (int j = 0; j < CI / 16; ++j) {
          const bf16x8 fw = *reinterpret_cast<const bf16x8*>(WT + (j * 16 + l15) * LDW + kq * 32 + grp * 8);
          dacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fp, dacc[j], 0, 0, 0);
        }
      }
      bf16_t* drow = a.dx_out + (k0 + wave * 16 + l15) * CI + grp * 4;
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#pragma unroll
      for (int j = 0; j < CI / 16; ++j) {
        const u32x2 o = {f2bf2(dacc[j][0], dacc[j][1]), f2bf2(dacc[j][2], dacc[j][3])};
        *reinterpret_cast<u32x2*>(drow + j * 16) = o;
      }
    }
    if constexpr (DGS) {
      // dx^T[ci][px] for the wave's 16 pixels (K = the 32 output channels: one MFMA per 16 input channels) -> the dx tile in LDS
      const bf16x8 fp = *reinterpret_cast<const bf16x8*>(DS + (wave * 16 + l15) * LDD + grp * 8);
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#pragma unroll
      for (int j = 0; j < CI / 16; ++j) {
        const bf16x8 fw = *reinterpret_cast<const bf16x8*>(WT + (j * 16 + l15) * LDW + grp * 8);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fp, zero, 0, 0, 0);
        const u32x2 o = {f2bf2(d[0], d[1]), f2bf2(d[2], d[3])};
        *reinterpret_cast<u32x2*>(YS + (wave * 16 + l15) * LDY + j * 16 + grp * 4) = o;
      }
    }
    __syncthreads();
    if constexpr (DGS) {
      // the thread's own 16-byte pieces again (its channel group is fixed): dx out, and the producer's statistics on the bf16 values
      // stored, with the raw input kept from staging - the arithmetic of conv_pw_kernel's BNB epilogue
      const int xg8 = (tid % (CI / 8)) * 8;
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(&sxs[0][xg8]), sc1 = *reinterpret_cast<const f32x4*>(&sxs[0][xg8 + 4]);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(&sxs[1][xg8]), sh1 = *reinterpret_cast<const f32x4*>(&sxs[1][xg8 + 4]);
      auto accumulate = [&](auto ACT) {
#pragma unroll
        for (int c = 0; c < XV; ++c) {
          const int id = tid + c * 256, row = id / (CI / 8), col = (id % (CI / 8)) * 8;
          Vec<bf16_t> g;
          g.load(YS + row * LDY + col);
          *reinterpret_cast<u32x4*>(a.dx_out + (k0 + row) * CI + col) = g.raw;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = rawk[c].get(e);
            const float z = fmaf(x, e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
            const float gv = g.get(e);
            const float dz = decltype(ACT)::value == 1 ? (z > 0.f ? gv : 0.f) : decltype(ACT)::value == 2 ? gv * gelu_erf_grad(z) : gv;
            st1[e] += dz; st2[e] = fmaf(dz, x, st2[e]);
          }
        }
      };
      if (a.x_act == 1) accumulate(std::integral_constant<int, 1>{});
      else if (a.x_act == 2) accumulate(std::integral_constant<int, 2>{});
      else accumulate(std::integral_constant<int, 0>{});
    }
  }
  if constexpr (DGS) {
    // fold the 16 row groups of the block through LDS (the dx tile is dead), then one atomic per channel and sum
    __syncthreads();
    float* red = reinterpret_cast<float*>(YS);                     // [2][16][CI]
    const int rg = tid / (CI / 8), xg8 = (tid % (CI / 8)) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[(0 * 16 + rg) * CI + xg8 + e] = st1[e]; red[(1 * 16 + rg) * CI + xg8 + e] = st2[e]; }
    __syncthreads();
    for (int i = tid; i < 2 * CI; i += 256) {
      const int which = i / CI, c = i % CI;
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[(which * 16 + g) * CI + c];
      atomicAdd(a.st_sums + ((size_t)(range % RSSF_BN_BWD_SLOTS) * 2 + which) * CI + c, t);
    }
  }

  // partial plane of this pixel range: rows = co (4 grp + r), columns = ci (l15)
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = (wm * WM + i) * 16 + grp * 4 + r, ci = (wn * WN + j) * 16 + l15;
        a.partial[((int64_t)range * CO + co) * CI + ci] = acc[i][j][r];
      }
  if (do_bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&sbias[cg8 + e], bs2[e >> 1][e & 1]);
    __syncthreads();
    if (tid < CO) atomicAdd(a.dbias + tid, sbias[tid]);
  }
}

struct Shape { int co, ci, kpx, kpx_fuse; };
// the instantiated channel pairs (Cout, Cin) -> pixels per chunk, without / with the fused apply (the chunk that keeps the three staged
// tensors + the apply's temporaries under ~160 registers).  (Measured: a plain chunk of 128 pixels - twice the loads in flight per block -
// left the 32 <- 128 layers at 25 us and cost the low-resolution layers blocks; what helps is MORE BLOCKS per CU, see wgrad_pw_ksplit.)
constexpr Shape SHAPES[] = {{128, 32, 64, 64}, {32, 128, 64, 64}, {256, 64, 32, 32}, {64, 256, 32, 32}, {32, 64, 64, 64}, {64, 32, 64, 64},
                            {64, 128, 64, 64}, {128, 64, 64, 64}, {64, 64, 64, 64}};
const Shape* shape_of(int cout, int cin) {
  for (const Shape& s : SHAPES)
    if (s.co == cout && s.ci == cin) return &s;
  return nullptr;
}

template <int GM, int WM, int WN, int KPX, int KPF>
int launch_shape(const PwWgradArgs& a, int ksplit, bool fuse, bool res, hipStream_t st) {
  const dim3 grid((unsigned)ksplit);
  if (!fuse) conv_wgrad_pw_kernel<GM, WM, WN, KPX, false, false><<<grid, 256, 0, st>>>(a);
  else if (res) conv_wgrad_pw_kernel<GM, WM, WN, KPF, true, true><<<grid, 256, 0, st>>>(a);
  else conv_wgrad_pw_kernel<GM, WM, WN, KPF, true, false><<<grid, 256, 0, st>>>(a);
  return check_launch("conv_wgrad_pw");
}

}  // namespace

namespace rssf { namespace cv {

// (Cout, Cin) = (32, 128) - MlpDWBN's fc2 - has the variant with a pre-activation input operand
bool wgrad_pw_preact_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx) {
  return Cout == 32 && Cin == 128 && wgrad_pw_eligible(B, IH, IW, Cin, OH, OW, Cout, stride, ntaps, dy, dx);
}

// MlpDWBN's fc1 (128 <- 32): weight gradient + BatchNorm-backward apply + data gradient in one launch
bool wgrad_pw_dgrad_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx) {
  static const int z[1] = {0};
  return Cout == 128 && Cin == 32 && wgrad_pw_eligible(B, IH, IW, Cin, OH, OW, Cout, stride, ntaps, dy ? dy : z, dx ? dx : z);
}

bool wgrad_pw_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx) {
  if (ntaps != 1 || stride != 1 || IH != OH || IW != OW || dy[0] != 0 || dx[0] != 0) return false;
  const Shape* s = shape_of(Cout, Cin);
  const int64_t M = (int64_t)B * OH * OW;
  return s != nullptr && M % s->kpx == 0 && M >= 16 * s->kpx;            // (kpx_fuse divides kpx)
}

// split-K factor: every block a whole number of chunks, >= 2 of them.  Swept in the step (tools/ab_lib_flags.sh): see DESIGN.md
#ifndef RSSF_PW_KS_SMALL
#define RSSF_PW_KS_SMALL 512
#endif
#ifndef RSSF_PW_KS_BIG
#define RSSF_PW_KS_BIG 256
#endif
int wgrad_pw_ksplit(int B, int OH, int OW, int Cin, int Cout) {
  const Shape* s = shape_of(Cout, Cin);
  const int64_t chunks = (int64_t)B * OH * OW / s->kpx;
  int64_t ks = (int64_t)Cout * Cin > 8192 ? RSSF_PW_KS_BIG : RSSF_PW_KS_SMALL;
  if (ks > chunks / 2) ks = chunks / 2;
  if (ks < 1) ks = 1;
  const int64_t cpb = (chunks + ks - 1) / ks;           // chunks per block
  return (int)((chunks + cpb - 1) / cpb);
}

int launch_wgrad_pw(const void* dout, const void* in, float* partial, float* dbias, int B, int OH, int OW, int Cin, int Cout, int ksplit,
                    const WgradBn* bn, hipStream_t st, const float* x_ss, int x_act, const float* w_dg, void* dx_dg, float* st_sums) {
  const Shape* s = shape_of(Cout, Cin);
  if (!s) { set_error("conv_wgrad_pw: no kernel for %d -> %d channels", Cin, Cout); return RSSF_ERR_UNSUPPORTED; }
  PwWgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dout = (const bf16_t*)dout; a.in = (const bf16_t*)in; a.partial = partial; a.dbias = dbias;
  a.M = (int64_t)B * OH * OW;
  const int64_t chunks = a.M / s->kpx;
  a.per = (int)((chunks + ksplit - 1) / ksplit) * s->kpx;
  bool res = false;
  if (bn) {
    a.bn_dy = (const bf16_t*)bn->dy; a.bn_raw = (const bf16_t*)bn->raw; a.bn_res = (const bf16_t*)bn->res;
    a.bn_ss = bn->ss; a.bn_mi = bn->mi; a.bn_sums = bn->sums;
    a.draw_out = (bf16_t*)bn->draw; a.dres_out = (bf16_t*)bn->dres; a.dgamma = bn->dgamma; a.dbeta = bn->dbeta;
    a.bn_n = (float)bn->n; a.bn_pscale = bn->pscale; a.bn_act = bn->act; a.bn_training = bn->training;
    res = bn->res != nullptr;
  }
  const bool fuse = bn != nullptr;
  const int co = Cout, ci = Cin;
  if (st_sums) {             // the 32 <- 128 layer's whole backward behind a pre-activation input: dW, dbias, dx and the producer's statistics
    if (!w_dg || !dx_dg || !x_ss || fuse || !(co == 32 && ci == 128) || (a.M / s->kpx) * (int64_t)s->kpx != a.M) {
      set_error("conv_wgrad_pw: data gradient + statistics are a feature of the plain 32 <- 128 kernel with a pre-activation input");
      return RSSF_ERR_UNSUPPORTED;
    }
    a.x_ss = x_ss; a.x_act = x_act; a.w = w_dg; a.dx_out = (bf16_t*)dx_dg; a.st_sums = st_sums;
    conv_wgrad_pw_kernel<1, 2, 2, 64, false, false, true, false, true><<<dim3((unsigned)ksplit), 256, 0, st>>>(a);
    return check_launch("conv_wgrad_pw");
  }
  if (w_dg || dx_dg) {
    if (!w_dg || !dx_dg || !wgrad_pw_dgrad_eligible(B, OH, OW, Cin, OH, OW, Cout, 1, 1, nullptr, nullptr) || !fuse || res || x_ss) {
      set_error("conv_wgrad_pw: the fused data gradient is a feature of the 128 <- 32 kernel with the fused apply (no residual)");
      return RSSF_ERR_UNSUPPORTED;
    }
    a.w = w_dg; a.dx_out = (bf16_t*)dx_dg;
    conv_wgrad_pw_kernel<4, 2, 2, 64, true, false, false, true><<<dim3((unsigned)ksplit), 256, 0, st>>>(a);
    return check_launch("conv_wgrad_pw");
  }
  if (x_ss) {
    if (fuse || !(co == 32 && ci == 128)) { set_error("conv_wgrad_pw: the pre-activation input operand is a feature of the plain 32 <- 128 kernel"); return RSSF_ERR_UNSUPPORTED; }
    a.x_ss = x_ss; a.x_act = x_act;
    conv_wgrad_pw_kernel<1, 2, 2, 64, false, false, true><<<dim3((unsigned)ksplit), 256, 0, st>>>(a);
    return check_launch("conv_wgrad_pw");
  }
  if (co == 128 && ci == 32) return launch_shape<4, 2, 2, 64, 64>(a, ksplit, fuse, res, st);
  if (co == 32 && ci == 128) return launch_shape<1, 2, 2, 64, 64>(a, ksplit, fuse, res, st);
  if (co == 256 && ci == 64) return launch_shape<4, 4, 4, 32, 32>(a, ksplit, fuse, res, st);
  if (co == 64 && ci == 256) return launch_shape<1, 4, 4, 32, 32>(a, ksplit, fuse, res, st);
  if (co == 32 && ci == 64) return launch_shape<1, 2, 1, 64, 64>(a, ksplit, fuse, res, st);
  if (co == 64 && ci == 32) return launch_shape<4, 1, 2, 64, 64>(a, ksplit, fuse, res, st);
  if (co == 64 && ci == 128) return launch_shape<2, 2, 4, 64, 64>(a, ksplit, fuse, res, st);
  if (co == 128 && ci == 64) return launch_shape<2, 4, 2, 64, 64>(a, ksplit, fuse, res, st);
  return launch_shape<2, 2, 2, 64, 64>(a, ksplit, fuse, res, st);       // 64 x 64
}

} }
