// Weight gradient of a many-tap 128 -> 128 channel convolution with the INPUT operand in registers (gfx950 MFMA): MlpDWBN's fused
// {1x1 + 3x3 dil 6 + 3x3 dil 12} sum (ffn_block.py:226-228, 250-257; torch: convolution_backward's weight gradient), first stage.
//     dW[t][co][ci] = sum_p dout[p][co] * x[p + delta_t][ci]            K = pixels (262 144 at the benchmark geometry)
// Both operands of this product are K-major in memory.  conv_wgrad8x2_kernel (conv_wgrad.hip) stages both pixel-major in LDS - the
// dout slab and one shifted input slab per tap: 24 KB of LDS writes per 32 pixels and two taps - and reads them through
// ds_read_b64_tr_b16: 0.25 of the MFMA peak, LDS-bound.  Here the input comes from a TRANSPOSED, zero-padded copy
//     xT[ci][b][H + 2 pad][W + 2 pad]      (pixels contiguous; written beside x by the BatchNorm apply that produces x:
//                                           rssf_bn_finalize_apply_planes, bn.hip)
// so that for a K-step of 128 pixels of one image row and a tap, a wave's operand is 16 rows (ci) x 256 contiguous bytes per tile,
// shifted by the tap ALONG the contiguous axis: the coalesced-load + DPP-butterfly path of conv_taps128.hip with "pixel := input
// channel, channel := pixel" - no LDS, no masks (the border is the copy's zero padding).  Only dout goes through LDS, pixel-major as
// it lies in memory: ONE [128 p][128 co] tile per K-step, shared by the three taps of a workgroup, read as MFMA fragments by the
// transposing LDS read.  A workgroup = one K-range x one group of up to three taps with the SAME dy (they read the same rows of xT,
// shifted by dx inside the row: the same cache lines); its 24 tiles of 16 input channels go three to a wave.  The groups of a
// K-range are neighbours on one XCD.  Partials leave in the layout of the generic first stage ([ksplit][ntaps][co][ci] fp32) and
// take the same second stage (rssf_conv_wgrad_reduce_batch / wgrad_reduce_kernel).
// Measured stand-alone at 16 x 128 x 128 x 128, 17 taps (tools/mlp_wgrad_proto.hip): 158-170 us against 229 us.
#include <string.h>
#include <type_traits>
#include "conv.hip.h"
using namespace rssf;
using namespace rssf::cv;

namespace rssf {
namespace cv {

namespace {

constexpr int P_C = 128;
constexpr int P_PITCH = 144;                  // LDS row pitch of the dout tile (elements): 256 + 32 bytes
constexpr int P_TILE = 128 * P_PITCH;
constexpr int P_NTL = 3;                      // tiles per wave = taps per group
constexpr int P_MAXG = 10;                    // tap groups of a launch
typedef __attribute__((ext_vector_type(4))) short v4s;

struct PlanesArgs {
  const bf16_t* dout;     // [B*H*W][128]
  const bf16_t* xT;       // [128][B][H + 2 pad][W + 2 pad]
  float* partial;         // [ksplit][ntaps][128 co][128 ci]
  float* dbias;           // [128] (+=) or null
  int B, H, W, pad, ntaps, ksplit, spb, per, ngroups;      // spb: K-steps (128-pixel row segments) per K-range
  int gtap[P_MAXG][P_NTL];                                 // group -> its taps (a slot < 0: its first tap again, computed and discarded)
  int dy[MAX_TAPS], dx[MAX_TAPS];
};

template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ void lane_exchange(u32x4& a, u32x4& b, bool hi) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t ta = dpp_u32<CTRL>(b[d]), tb = dpp_u32<CTRL>(a[d]);
    const uint32_t na = hi ? ta : a[d], nb = hi ? b[d] : tb;
    a[d] = na; b[d] = nb;
  }
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_wgrad_planes_kernel(PlanesArgs a) {
  constexpr int C = P_C, NTL = P_NTL, PITCH = P_PITCH, TILE = P_TILE;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned q = (blockIdx.x & 7u) * (unsigned)a.per + (blockIdx.x >> 3);      // XCD-major: the groups of a K-range share an L2
  if (q >= (unsigned)(a.ngroups * a.ksplit)) return;
  const int tg = (int)(q % (unsigned)a.ngroups), ks = (int)(q / (unsigned)a.ngroups);
  const int HP = a.H + 2 * a.pad, WP = a.W + 2 * a.pad;
  const int PP = a.B * HP * WP;                                   // plane pitch (elements)
  const int segs = a.W / 128, nsteps = a.B * a.H * segs;
  const int s0 = ks * a.spb, s1 = s0 + a.spb < nsteps ? s0 + a.spb : nsteps;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.xT), 0, (int)((int64_t)C * PP * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dout), 0, (int)((int64_t)nsteps * 128 * C * 2), 0x00020000);
  // this wave's tiles: tile 3 wave + j of the group's 24 -> tap slot (tile / 8), input-channel tile (tile % 8)
  int tci[NTL], ttap[NTL], tdx2[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int g = wave * NTL + j;
    tci[j] = g & 7;
    const int t = a.gtap[tg][g >> 3];
    ttap[j] = t;
    tdx2[j] = a.dx[t < 0 ? a.gtap[tg][0] : t] * 2;
  }
  const int gdy = a.dy[a.gtap[tg][0]];
  // pair loads of conv_taps128.hip: lane = [k5 k4 | p2 p1 p0 | k2], load J = [p3 k3]; "pixel" = ci (a plane of xT), "channel" = pixel
  const unsigned lane_base = (unsigned)(((lane >> 1) & 7) * PP * 2 + grp * 64 + (lane & 1) * 16);
  auto step_off = [&](int s) {                                    // byte offset (in a plane) of the 128-pixel run K-step s reads at dx = 0
    const int r = s / segs, seg = s - r * segs;
    const int b = r / a.H, y = r - b * a.H;
    return ((b * HP + y + a.pad + gdy) * WP + a.pad + seg * 128) * 2;
  };
  // dout tile staging: thread -> pixel pl = tid / 16 (+ 32 i), 16-byte chunk tid % 16; LDS row R(p) = 8 Q + (p & 3) + 4 ((p >> 5) & 1),
  // Q = ((p >> 2) & 1) + 2 ((p >> 3) & 3) + 8 (p >> 6): the eight 4-row pieces a transposing read of 32 lanes touches (rows
  // 32 g + 8 kappa + j, j = 0..3, g = two lane groups) are LDS rows with eight different residues mod 8 = eight 32-byte bank groups
  const int pl = tid >> 4;
  const unsigned dsrc = (unsigned)(pl * (C * 2) + (tid & 15) * 16);
  const int ddst = (8 * (((pl >> 2) & 1) + 2 * ((pl >> 3) & 3)) + (pl & 3)) * PITCH + (tid & 15) * 8;
  // fragment (ct, kappa): first read rows 32 g + 8 kappa + (i >> 2) -> LDS rows 64 (g >> 1) + 4 (g & 1) + (i >> 2) + 16 kappa, second + 8;
  // K-slot e of lane group g = pixel 32 g + 8 kappa + e: what register kappa of the butterflied input operand holds for that group
  const int foff = (64 * (grp >> 1) + 4 * (grp & 1) + (l15 >> 2)) * PITCH + (l15 & 3) * 4;

  f32x4 acc[NTL][8];
#pragma unroll
  for (int mi = 0; mi < NTL; ++mi)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[mi][ct] = {0.f, 0.f, 0.f, 0.f};
  u32x4 RA[NTL][4], RB[4];
  auto load_A = [&](int mi, int soff) {
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      const unsigned v = lane_base + (unsigned)((tci[mi] * 16 + 8 * (J >> 1)) * PP * 2 + (J & 1) * 32);
      RA[mi][J] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, v, soff + tdx2[mi], 0));
    }
  };
  auto load_B = [&](int s) {
    const int soff = s < s1 ? s * (128 * C * 2) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      RB[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, s < s1 ? dsrc : 0x80000000u, soff + i * 32 * C * 2, 0));
  };
  auto store_B = [&](bf16_t* Bs) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(Bs + ddst + (64 * (i >> 1) + 4 * (i & 1)) * PITCH) = RB[i];
  };
  // the request order of the prologue is the loop's (exact wait counts at the loop header, conv_taps128.hip)
  load_B(s0);
  store_B(lds);
  load_B(s0 + 1);
  __builtin_amdgcn_sched_barrier(0);
  {
    const int so = step_off(s0);
#pragma unroll
    for (int mi = 0; mi < NTL; ++mi) load_A(mi, so);
  }
  __builtin_amdgcn_sched_barrier(0);
  const bool hi0 = lane & 1;
  auto pass = [&](const bf16_t* Bs, auto M0, auto M1) {          // tiles [M0, M1) against all 32 fragments of the dout tile
    constexpr int m0 = decltype(M0)::value, m1 = decltype(M1)::value;
#pragma unroll
    for (int mi = m0; mi < m1; ++mi) {
      lane_exchange<0xB1>(RA[mi][0], RA[mi][2], hi0);
      lane_exchange<0xB1>(RA[mi][1], RA[mi][3], hi0);
    }
#pragma unroll
    for (int kp = 0; kp < 4; ++kp)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        bf16x8 fb[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const bf16_t* p = Bs + foff + kp * 16 * PITCH + (hf * 4 + c4) * 16;
          const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
          const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 8 * PITCH));
          union { struct { v4s a, b; } s; bf16x8 v; } u;
          u.s.a = lo; u.s.b = hi;
          fb[c4] = u.v;
        }
        const int rg = ((kp & 1) << 1) | (kp >> 1);
#pragma unroll
        for (int mi = m0; mi < m1; ++mi)
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4)
            acc[mi][hf * 4 + c4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[c4], __builtin_bit_cast(bf16x8, RA[mi][rg]), acc[mi][hf * 4 + c4], 0, 0, 0);
      }
  };
  // bias gradient = column sums of dout: the workgroups of tap group 0 add up the tile that is in LDS anyway (thread: one output
  // channel, a quarter of the tile's rows - in LDS order, the sum does not care)
  const bool do_bias = a.dbias != nullptr && tg == 0;
  float bsum = 0.f;
  for (int s = s0; s < s1; ++s) {
    const int t = s - s0;
    const bf16_t* Bs = lds + (t & 1) * TILE;
    __syncthreads();
    store_B(lds + ((t + 1) & 1) * TILE);
    load_B(s + 2);
    const int so = step_off(s + 1 < s1 ? s + 1 : s);
    __builtin_amdgcn_sched_barrier(0);
    pass(Bs, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    load_A(0, so);
    load_A(1, so);
    __builtin_amdgcn_sched_barrier(0);
    pass(Bs, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
    load_A(2, so);
    if (do_bias) {
      const bf16_t* col = Bs + (tid >> 7) * 32 * PITCH + (tid & 127);
#pragma unroll 8
      for (int k = 0; k < 32; ++k) bsum += ldf(col + k * PITCH);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  if (do_bias) {
    float* red = reinterpret_cast<float*>(lds);                  // [4][128]
    red[tid] = bsum;
    __syncthreads();
    if (tid < 128) atomicAdd(a.dbias + tid, (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]));
    __syncthreads();
  }
  // ---- partials: acc[mi][ct][r] = dW[tap of tile mi][co = 16 ct + 4 grp + r][ci = 16 tci + prow(l15)]; a tile at a time through a
  //      wave-private [128 co][16 ci] fp32 LDS tile -> 64-byte runs of partial[ks][tap][co][ci]
  float* Ts = reinterpret_cast<float*>(lds) + wave * (128 * 16);
  const int prow = 8 * (l15 & 1) + (l15 >> 1);
#pragma unroll
  for (int mi = 0; mi < NTL; ++mi) {
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ts[(ct * 16 + grp * 4 + r) * 16 + prow] = acc[mi][ct][r];
    __builtin_amdgcn_wave_barrier();
    if (ttap[mi] >= 0) {
      float* dst = a.partial + ((size_t)(ks * a.ntaps + ttap[mi]) * C) * C + tci[mi] * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j, co = c >> 2, part = c & 3;
        *reinterpret_cast<f32x4*>(dst + (size_t)co * C + part * 4) = *reinterpret_cast<const f32x4*>(Ts + co * 16 + part * 4);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// tap groups: the taps of one dy, three at a time.  Returns the number of groups (0: more than P_MAXG)
int make_groups(int ntaps, const int* dy, int (*gtap)[P_NTL]) {
  bool used[MAX_TAPS] = {false};
  int ng = 0;
  for (int t0 = 0; t0 < ntaps; ++t0) {
    if (used[t0]) continue;
    int list[MAX_TAPS], m = 0;
    for (int t = t0; t < ntaps; ++t)
      if (!used[t] && dy[t] == dy[t0]) { list[m++] = t; used[t] = true; }
    for (int i = 0; i < m; i += P_NTL, ++ng) {
      if (ng >= P_MAXG) return 0;
      for (int j = 0; j < P_NTL; ++j) gtap[ng][j] = i + j < m ? list[i + j] : -1;
    }
  }
  return ng;
}
int planes_ksplit(int nsteps, int ngroups, int& spb) {
  int ks = 256 / ngroups;                     // one round of the chip: a workgroup per CU
  if (ks < 1) ks = 1;
  if (ks > nsteps) ks = nsteps;
  spb = (nsteps + ks - 1) / ks;
  return (nsteps + spb - 1) / spb;
}
bool planes_shape_ok(int B, int H, int W, int Cin, int Cout, int ntaps) {
  return Cin == P_C && Cout == P_C && ntaps >= 8 && ntaps <= MAX_TAPS && (W % 128) == 0 && B > 0 && H > 0;
}

}  // namespace

int64_t wgrad_planes_workspace_elems(int B, int H, int W, int Cin, int Cout, int ntaps) {
  if (!planes_shape_ok(B, H, W, Cin, Cout, ntaps)) return 0;
  // upper bound over the tap sets (the grouping depends on the taps' dy): one group per tap at worst is refused by _supported;
  // with at least ceil(ntaps / 3) groups ksplit <= 256 / that
  const int gmin = (ntaps + P_NTL - 1) / P_NTL;
  return (int64_t)(256 / gmin + 1) * ntaps * P_C * P_C;
}

}  // namespace cv
}  // namespace rssf

extern "C" int rssf_conv_wgrad_planes_supported(int B, int H, int W, int Cin, int Cout, int stride, int ntaps, const int* dy, const int* dx,
                                                int pad, int dtype) {
  if (dtype != RSSF_BF16 || stride != 1 || !dy || !dx || !planes_shape_ok(B, H, W, Cin, Cout, ntaps)) return 0;
  for (int t = 0; t < ntaps; ++t)
    if (dy[t] < -pad || dy[t] > pad || dx[t] < -pad || dx[t] > pad || (dx[t] & 1)) return 0;      // (odd dx: a 2-byte-aligned 16-byte load)
  if ((int64_t)P_C * B * (H + 2 * pad) * (W + 2 * pad) * 2 >= ((int64_t)1 << 31) || (int64_t)B * H * W * P_C * 2 >= ((int64_t)1 << 31)) return 0;
  int gtap[P_MAXG][P_NTL];
  return make_groups(ntaps, dy, gtap) > 0 ? 1 : 0;
}

extern "C" int rssf_conv_wgrad_planes(const void* dout, const void* in_planes, int pad, float* dw0, float* dw1, float* dw2, const int* ksizes,
                                      int nsrc, const int* src_of_tap, const int* kpos_of_tap, const int* alias_of_tap, float* dbias,
                                      float* workspace, int B, int H, int W, int C, int ntaps, const int* dy, const int* dx,
                                      rssf_wgrad_reduce_job* defer_reduce, int dtype, void* stream) {
  RSSF_REQUIRE(dout && in_planes && dw0 && ksizes && src_of_tap && kpos_of_tap && workspace && dy && dx && nsrc >= 1 && nsrc <= 3,
               "conv_wgrad_planes: bad arguments");
  RSSF_REQUIRE(rssf_conv_wgrad_planes_supported(B, H, W, C, C, 1, ntaps, dy, dx, pad, dtype),
               "conv_wgrad_planes: unsupported shape (ask rssf_conv_wgrad_planes_supported)");
  PlanesArgs a;
  memset(&a, 0, sizeof(a));
  a.dout = (const bf16_t*)dout; a.xT = (const bf16_t*)in_planes; a.partial = workspace; a.dbias = dbias;
  a.B = B; a.H = H; a.W = W; a.pad = pad; a.ntaps = ntaps;
  for (int t = 0; t < ntaps; ++t) { a.dy[t] = dy[t]; a.dx[t] = dx[t]; }
  a.ngroups = make_groups(ntaps, dy, a.gtap);
  const int nsteps = B * H * (W / 128);
  a.ksplit = planes_ksplit(nsteps, a.ngroups, a.spb);
  RSSF_REQUIRE((int64_t)a.ksplit * ntaps * P_C * P_C <= wgrad_planes_workspace_elems(B, H, W, C, C, ntaps), "conv_wgrad_planes: workspace bound");
  a.per = xcd_per((int64_t)a.ngroups * a.ksplit);
  hipStream_t st = (hipStream_t)stream;
  conv_wgrad_planes_kernel<<<dim3((unsigned)a.per * 8u), 512, 0, st>>>(a);
  if (int rc = check_launch("conv_wgrad_planes")) return rc;
  rssf_wgrad_reduce_job j;
  memset(&j, 0, sizeof(j));            // padding bytes too: callers compare job descriptions bytewise
  j.partial = workspace;
  j.dw[0] = dw0; j.dw[1] = dw1; j.dw[2] = dw2;
  for (int i = 0; i < 3; ++i) j.ks[i] = i < nsrc ? ksizes[i] : 1;
  j.ntaps = ntaps; j.cout = C; j.cin = C; j.ksplit = a.ksplit;
  for (int t = 0; t < MAX_TAPS; ++t) {
    const bool live = t < ntaps;
    j.src_of_tap[t] = live ? src_of_tap[t] : 0; j.kpos_of_tap[t] = live ? kpos_of_tap[t] : 0;
    for (int e = 0; e < 4; ++e) {
      j.alias_of_tap[t][e] = (live && alias_of_tap) ? alias_of_tap[t * 4 + e] : -1;
      RSSF_REQUIRE((e & 1) || j.alias_of_tap[t][e] < nsrc, "conv_wgrad_planes: alias source out of range");
    }
  }
  if (defer_reduce) { *defer_reduce = j; return RSSF_OK; }
  return launch_wgrad_reduce(j, st);
}
