// Backward of the STEM's first convolution (reference _hrnet_rssformer.py:407-413, 441-447: Conv2d(3, 64, 3, stride 2, padding 1, bias=False) ->
// BatchNorm -> ReLU on the 512 x 512 image, channels padded 3 -> 8 for 16-byte pixels): the layer's BatchNorm-backward apply and its weight
// gradient in ONE pass over dy and raw, at the END of the step's backward where nothing runs beside it.
//
//   dW[co][tap][ci] = sum_p draw[p][co] * x[2 oy + dy_t][2 ox + dx_t][ci],   p = (b, oy, ox): 1 048 576 output pixels at the benchmark geometry
//
// Before: bn_bwd_apply_kernel (read dy, raw: 134 MB each, write draw: 134 MB, 66 us) + the generic weight-gradient kernel with one tap
// per block (each of its nine tap blocks re-stages the dout slab - 1.2 GB through the L2 - and an input slab of which 7/8 are channel
// padding: 171 us).  The image needs no gradient, so `draw` has NO other reader: here a block owns every (co, tap, ci) of a range of
// output pixels, forms draw for a chunk of 64 pixels from dy and raw on the way to LDS (bn_bwd_apply_kernel's arithmetic,
// common.hip.h), gathers the chunk's im2col patches - 9 taps x 8 channels = nine 16-byte pieces per pixel, zeros outside the image - into
// a [64][72] LDS tile and contracts the two tiles over the pixels with transposing LDS reads (conv_wgrad.hip): a GEMM with
// M = 64, N = 72 (padded to 80), K = pixels.  draw is written only if the caller asks for it.  Partials in the common split-K layout
// [ksplit][9][64][8], the common second stage.
//
// The same kernel, templated on the channel counts, serves the down-sampling 3x3 / stride-2 fuse convolutions with 32 input channels
// (_hrnet_rssformer.py:380-405: 32 -> 32 / 64 / 128; four 16-byte pieces per tap and pixel, N = 288 patch columns): the generic kernel
// gives every tap a block of its own and re-stages the dout slab nine times (25 launches of 15-26 us per step, on the side stream).
#include <cstring>
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace {

typedef __attribute__((ext_vector_type(4))) short v4s;

constexpr int S_TAPS = 9, S_KPX = 64;

struct StemArgs {
  const bf16_t* dout;      // [M][CO] (FUSE: not read)
  const bf16_t* in;        // [B][IH][IW][CI]
  float* partial;          // [ksplit][9][CO][CI]
  int64_t M;
  int B, IH, IW, OH, OW, per;
  const bf16_t* bn_dy; const bf16_t* bn_raw;
  const float* bn_ss; const float* bn_mi; const float* bn_sums;
  bf16_t* draw_out;        // or null
  float* dgamma; float* dbeta;
  float bn_n, bn_pscale;
  int bn_act, bn_training;
};

__device__ __forceinline__ bf16x8 sfrag(const bf16_t* tile, int ld, int k0, int c0, int lane) {      // conv_wgrad.hip's SlabFrag
  const int grp = lane >> 4, i = lane & 15;
  const bf16_t* p = tile + (k0 + grp * 4 + (i >> 2)) * ld + c0 + (i & 3) * 4;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 16 * ld));
  union { struct { v4s a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// S_CI input channels (8: the channel-padded image; 32), S_CO output channels (32 / 64 / 128).  N = 9 S_CI patch columns (padded to whole MFMA
// tiles); waves: one (or two) output-channel tiles each and all patch-column tiles - for 32 output channels two waves share a tile's columns
template <int S_CI, int S_CO, bool FUSE>
__global__ void __launch_bounds__(256) conv_wgrad_stem_kernel(StemArgs a) {
  constexpr int S_N = S_TAPS * S_CI, S_NP = (S_N + 15) / 16 * 16, S_LDD = S_CO + 16, S_LDX = S_NP + 16;
  constexpr int PPP = S_CI / 8;                              // 16-byte pieces per (pixel, tap)
  constexpr int COT = S_CO / 16, CW = COT >= 4 ? COT / 4 : 1, NSPLIT = COT >= 4 ? 1 : 4 / COT, NT = S_NP / 16, NTW = NT / NSPLIT;
  static_assert(NT % NSPLIT == 0 && 256 % (S_CO / 8) == 0 && S_CI % 8 == 0, "tiling");
  constexpr int DV = S_KPX * S_CO / 8 / 256;                 // vectors of dout per thread and chunk
  constexpr int XP = (S_KPX * S_TAPS * PPP + 255) / 256;     // patch pieces per thread and chunk
  static_assert(DV >= 1 && DV * 256 * 8 == S_KPX * S_CO, "the dout chunk divides over the block");
  __shared__ __attribute__((aligned(16))) bf16_t DS[S_KPX * S_LDD];
  __shared__ __attribute__((aligned(16))) bf16_t XS[S_KPX * S_LDX];
  __shared__ __attribute__((aligned(16))) float sbn[4][FUSE ? S_CO : 4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int range = blockIdx.x;
  const int64_t kbeg = (int64_t)range * a.per;
  const int64_t kend = kbeg + a.per < a.M ? kbeg + a.per : a.M;
  // the pad columns N .. NP - 1 of the patch tile stay zero (they are never written again)
  if constexpr (S_NP > S_N) {
    for (int i = tid; i < S_KPX * (S_NP - S_N); i += 256) XS[(i / (S_NP - S_N)) * S_LDX + S_N + i % (S_NP - S_N)].v = 0;
  }
  if constexpr (FUSE) {
    if (tid < S_CO) {
      const int c = tid;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < RSSF_BN_BWD_SLOTS; ++k) { s1 += a.bn_sums[(size_t)k * 2 * S_CO + c]; s2 += a.bn_sums[(size_t)k * 2 * S_CO + S_CO + c]; }
      const float mean = a.bn_mi[c], istd = a.bn_mi[S_CO + c];
      const float sc = a.bn_ss[c], sh = a.bn_ss[S_CO + c];
      float dot, cb, cc;
      bn_bwd_constants(sc, mean, istd, s1, s2, a.bn_n, dot, cb, cc);
      if (a.dgamma && range == 0) { a.dgamma[c] += dot * a.bn_pscale; a.dbeta[c] += s1 * a.bn_pscale; }
      sbn[0][c] = sc; sbn[1][c] = sh; sbn[2][c] = cb; sbn[3][c] = cc;
    }
  }
  __syncthreads();

  const int dbytes = (int)(a.M * S_CO * 2), xbytes = (int)((int64_t)a.B * a.IH * a.IW * S_CI * 2);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(FUSE ? a.bn_dy : a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rraw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(FUSE ? a.bn_raw : a.dout), 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdraw = __builtin_amdgcn_make_buffer_rsrc(FUSE && a.draw_out ? a.draw_out : const_cast<bf16_t*>(FUSE ? a.bn_dy : a.dout), 0, dbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  const int cot0 = COT >= 4 ? wave * CW : wave % COT;        // this wave's first output-channel tile
  const int nt0 = COT >= 4 ? 0 : (wave / COT) * NTW;         // ... and its first patch-column tile
  f32x4 acc[CW][NTW];
#pragma unroll
  for (int i = 0; i < CW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  // this thread's patch pieces: slot -> (pixel of the chunk, tap); fixed for all chunks
  int ppx[XP], pcol[XP], pdy[XP], pdx[XP];                  // pixel of the chunk, first patch column (= 8 x piece of the pixel's row), tap offsets
#pragma unroll
  for (int c = 0; c < XP; ++c) {
    const int id = tid + c * 256;
    ppx[c] = id / (S_TAPS * PPP);
    const int rem = id % (S_TAPS * PPP), tap = rem / PPP;
    pcol[c] = rem * 8;                                       // tap * S_CI + (rem % PPP) * 8
    pdy[c] = tap / 3 - 1; pdx[c] = tap % 3 - 1;
  }
  Vec<bf16_t> rd[DV], rr[FUSE ? DV : 1], rx[XP];
  unsigned soff[DV];
  const int ohw = a.OH * a.OW;
  auto load_chunk = [&](int64_t k0) {
    const unsigned db = (unsigned)(k0 * S_CO * 2) + (unsigned)tid * 16u;
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      rd[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, db + c * 4096u, 0, 0));
      if constexpr (FUSE) rr[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rraw, db + c * 4096u, 0, 0));
    }
#pragma unroll
    for (int c = 0; c < XP; ++c) {
      const int p = (int)k0 + ppx[c];                          // output pixel (the entry point keeps M below 2^31)
      const int b = p / ohw, rem = p - b * ohw;
      const int oy = rem / a.OW, ox = rem - oy * a.OW;
      const int iy = 2 * oy + pdy[c], ix = 2 * ox + pdx[c];
      const bool ok = ppx[c] < S_KPX && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
      const unsigned off = ok ? (unsigned)(((b * a.IH + iy) * a.IW + ix) * (S_CI * 2) + (pcol[c] % S_CI) * 2) : OOB;
      rx[c].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0));      // out of the image: zeros
    }
  };
  const int cg8 = (tid % (S_CO / 8)) * 8;
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
      float o1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = rr[c].get(e);
        const float z = fmaf(x, bsc[e], bsh[e]);
        const float g = rd[c].get(e);
        const float dz = decltype(ACT)::value == 1 ? g * (z > 0.f ? 1.f : 0.f) : decltype(ACT)::value == 2 ? g * gelu_erf_grad(z) : g;
        o1[e] = a.bn_training ? fmaf(bsc[e], dz, fmaf(bcb[e], x, bcc[e])) : bsc[e] * dz;
      }
      rd[c].set_all(o1);
    }
  };
  Vec<bf16_t> kd[FUSE ? DV : 1];
  auto stage = [&](int64_t k0) {
    if constexpr (FUSE) {
      if (a.bn_act == 1) apply_chunk(std::integral_constant<int, 1>{});
      else if (a.bn_act == 2) apply_chunk(std::integral_constant<int, 2>{});
      else apply_chunk(std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      const int id = tid + c * 256, row = id / (S_CO / 8), col = (id % (S_CO / 8)) * 8;
      rd[c].store(DS + row * S_LDD + col);
      soff[c] = (unsigned)(k0 * S_CO * 2) + (unsigned)id * 16u;
      if constexpr (FUSE) kd[c] = rd[c];
    }
#pragma unroll
    for (int c = 0; c < XP; ++c)
      if (ppx[c] < S_KPX) rx[c].store(XS + ppx[c] * S_LDX + pcol[c]);
  };

  if (kbeg < kend) load_chunk(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += S_KPX) {
    stage(k0);
    __syncthreads();
    if (k0 + S_KPX < kend) load_chunk(k0 + S_KPX);
    if constexpr (FUSE) {                                        // (after the next chunk's loads: vmcnt counts in order)
#pragma unroll
      for (int c = 0; c < DV; ++c)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, kd[c].raw), rdraw,
                                               a.draw_out ? soff[c] : OOB, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < S_KPX; ks += 32) {
      bf16x8 fa[CW];
#pragma unroll
      for (int i = 0; i < CW; ++i) fa[i] = sfrag(DS, S_LDD, ks, (cot0 + i) * 16, lane);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const bf16x8 fb = sfrag(XS, S_LDX, ks, (nt0 + j) * 16, lane);
#pragma unroll
        for (int i = 0; i < CW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // partial[range][tap][co][ci]: rows = co (4 grp + r of the tile), columns n = 16 j + l15 = S_CI tap + ci
#pragma unroll
  for (int i = 0; i < CW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = (nt0 + j) * 16 + l15;
      if (n >= S_N) continue;
      const int tap = n / S_CI, ci = n % S_CI;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = (cot0 + i) * 16 + grp * 4 + r;
        a.partial[(((int64_t)range * S_TAPS + tap) * S_CO + co) * S_CI + ci] = acc[i][j][r];
      }
    }
}

template <int CI, int CO>
int stem_launch(const StemArgs& a, int ksplit, bool fuse, hipStream_t st) {
  if (fuse) conv_wgrad_stem_kernel<CI, CO, true><<<dim3((unsigned)ksplit), 256, 0, st>>>(a);
  else conv_wgrad_stem_kernel<CI, CO, false><<<dim3((unsigned)ksplit), 256, 0, st>>>(a);
  return check_launch("conv_wgrad_stem");
}

// (32 -> 128 is instantiated and tested but not dispatched: on its 32 x 32 output map - 256 chunks, 64 workgroups - it ran 19.5 us against
// the generic kernel's 15)
bool stem_shape(int cin, int cout) { return (cin == 8 && cout == 64) || (cin == 32 && (cout == 32 || cout == 64)); }

}  // namespace

namespace rssf { namespace cv {

bool wgrad_stem_eligible(int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx) {
  if (!stem_shape(Cin, Cout) || stride != 2 || ntaps != S_TAPS || IH != 2 * OH || IW != 2 * OW) return false;
  for (int t = 0; t < S_TAPS; ++t)
    if (dy[t] != t / 3 - 1 || dx[t] != t % 3 - 1) return false;
  const int64_t M = (int64_t)B * OH * OW;
  return M % S_KPX == 0 && M >= 8 * S_KPX && M * Cout * 2 < ((int64_t)1 << 31) && (int64_t)B * IH * IW * Cin * 2 < ((int64_t)1 << 31);
}

// split-K factor: whole chunks per block, >= 4 of them; the partial planes are 18 KB (the stem) .. 147 KB (32 -> 128) each
int wgrad_stem_ksplit(int B, int OH, int OW, int Cin, int Cout) {
  const int64_t chunks = (int64_t)B * OH * OW / S_KPX;
  int64_t ks = Cin == 8 ? 512 : (Cout >= 128 ? 128 : 256);
  if (ks > chunks / 4) ks = chunks / 4;
  if (ks < 1) ks = 1;
  const int64_t cpb = (chunks + ks - 1) / ks;
  return (int)((chunks + cpb - 1) / cpb);
}

int launch_wgrad_stem(const void* dout, const void* in, float* partial, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int ksplit,
                      const WgradBn* bn, bool write_draw, hipStream_t st) {
  StemArgs a;
  memset(&a, 0, sizeof(a));
  a.dout = (const bf16_t*)dout; a.in = (const bf16_t*)in; a.partial = partial;
  a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW;
  a.M = (int64_t)B * OH * OW;
  const int64_t chunks = a.M / S_KPX;
  a.per = (int)((chunks + ksplit - 1) / ksplit) * S_KPX;
  if (bn) {
    if (bn->res || bn->dres) { set_error("conv_wgrad_stem: no residual form"); return RSSF_ERR_UNSUPPORTED; }
    a.bn_dy = (const bf16_t*)bn->dy; a.bn_raw = (const bf16_t*)bn->raw; a.bn_ss = bn->ss; a.bn_mi = bn->mi; a.bn_sums = bn->sums;
    a.draw_out = write_draw ? (bf16_t*)bn->draw : nullptr; a.dgamma = bn->dgamma; a.dbeta = bn->dbeta;
    a.bn_n = (float)bn->n; a.bn_pscale = bn->pscale; a.bn_act = bn->act; a.bn_training = bn->training;
  }
  const bool fuse = bn != nullptr;
  if (Cin == 8 && Cout == 64) return stem_launch<8, 64>(a, ksplit, fuse, st);
  if (Cin == 32 && Cout == 32) return stem_launch<32, 32>(a, ksplit, fuse, st);
  if (Cin == 32 && Cout == 64) return stem_launch<32, 64>(a, ksplit, fuse, st);
  if (Cin == 32 && Cout == 128) return stem_launch<32, 128>(a, ksplit, fuse, st);
  set_error("conv_wgrad_stem: no kernel for %d -> %d channels", Cin, Cout);
  return RSSF_ERR_UNSUPPORTED;
}

} }
