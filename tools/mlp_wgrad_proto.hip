// Stand-alone prototype (no PyTorch): weight gradient of MlpDWBN's 17-tap convolution with the INPUT operand in registers.
//   dW[t][co][ci] = sum_p dout[p][co] * x[p + delta_t][ci]        (K = pixels: 262 144 at the benchmark geometry)
// The shipping kernel (conv_wgrad8x2_kernel) stages both operands pixel-major in LDS, per tap pair, and reads them through
// ds_read_b64_tr_b16: 24 KB of LDS writes per 32 pixels and two taps.  Here the input comes from a TRANSPOSED, zero-padded copy
// xT[ci][b][H + 24][W + 24] (pixels contiguous: what the BatchNorm apply that produces x would write beside x): for a K-step of one
// image row (128 pixels) and a tap, a wave's operand is 64 rows (ci) x 256 contiguous bytes, shifted by the tap along the contiguous
// axis - the coalesced-load + DPP-butterfly path of conv_taps128.hip with "pixel := ci, channel := pixel".  Only dout goes through
// LDS, pixel-major as it lies in memory, ONE [128 p][128 co] tile per K-step shared by the four taps (eight waves) of a workgroup,
// read as MFMA fragments by the transposing LDS read.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mlp_wgrad_proto.hip -Lrepresentationlearning_amd/lib -lrssf -o tools/bin/mlp_wgrad_proto
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <type_traits>
#include "../representationlearning_amd/csrc/common.hip.h"
using namespace rssf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int MAXT = 19, PAD = 12, C = 128;
constexpr int PITCH = 144;                    // LDS row pitch of the dout tile (elements): 256 + 32 bytes
constexpr int TILE = 128 * PITCH;
typedef __attribute__((ext_vector_type(4))) short v4s;

struct WArgs {
  const bf16_t* dout;     // [B*H*W][128]
  const bf16_t* xT;       // [128][B][H + 2 PAD][W + 2 PAD]
  float* partial;         // [ksplit][ntaps][128 co][128 ci]
  int B, H, W, ntaps, ksplit, rpb, per, ngroups;      // rpb: image rows per K-range
  int gtap[8][3];         // tap group (all of one dy) -> its three taps (a slot < 0: repeat of slot 0, computed and discarded)
  int dy[MAXT], dx[MAXT];
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

// A workgroup = one K-range (a run of image rows) x one tap group of THREE taps with the same dy: 24 tiles of 16 input channels,
// three per wave.  The taps of a group read the same rows of xT (shifted by dx inside the row: the same cache lines), and the
// groups of a K-range are neighbours on one XCD.  MODE 0 full, 1 no MFMA, 2 no xT loads
constexpr int NTL = 3;
template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) wgrad_rtr_kernel(WArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned q = (blockIdx.x & 7u) * (unsigned)a.per + (blockIdx.x >> 3);      // XCD-major: the groups of a K-range share an L2
  if (q >= (unsigned)(a.ngroups * a.ksplit)) return;
  const int tg = (int)(q % (unsigned)a.ngroups), ks = (int)(q / (unsigned)a.ngroups);
  const int HP = a.H + 2 * PAD, WP = a.W + 2 * PAD;
  const int PP = a.B * HP * WP;                                   // plane pitch (elements)
  const int nrows = a.B * a.H;
  const int r0 = ks * a.rpb, r1 = r0 + a.rpb < nrows ? r0 + a.rpb : nrows;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.xT), 0, (int)((int64_t)C * PP * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dout), 0, (int)((int64_t)nrows * a.W * C * 2), 0x00020000);
  // this wave's tiles: global tile 3 wave + j -> tap slot (tile / 8), input-channel tile (tile % 8)
  int tslot[NTL], tci[NTL], ttap[NTL], tdx2[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int g = wave * NTL + j;
    tslot[j] = g >> 3; tci[j] = g & 7;
    const int t = a.gtap[tg][tslot[j]];
    ttap[j] = t;
    tdx2[j] = a.dx[t < 0 ? a.gtap[tg][0] : t] * 2;
  }
  const int gdy = a.dy[a.gtap[tg][0]];
  // pair loads of conv_taps128.hip: lane = [k5 k4 | p2 p1 p0 | k2], load J = [p3 k3]; "pixel" = ci (a plane of xT), "channel" = pixel
  const unsigned lane_base = (unsigned)(((lane >> 1) & 7) * PP * 2 + grp * 64 + (lane & 1) * 16);
  auto row_off = [&](int r) {                                     // byte offset (in a plane) of the 128-pixel run image row r reads at dx = 0
    const int b = r / a.H, y = r - b * a.H;
    return ((b * HP + y + PAD + gdy) * WP + PAD) * 2;
  };
  const int pl = tid >> 4;
  const unsigned dsrc = (unsigned)(pl * (C * 2) + (tid & 15) * 16);
  const int ddst = (8 * (((pl >> 2) & 1) + 2 * ((pl >> 3) & 3)) + (pl & 3)) * PITCH + (tid & 15) * 8;
  const int foff = (64 * (grp >> 1) + 4 * (grp & 1) + (l15 >> 2)) * PITCH + (l15 & 3) * 4;

  f32x4 acc[NTL][8];
#pragma unroll
  for (int mi = 0; mi < NTL; ++mi)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[mi][ct] = {0.f, 0.f, 0.f, 0.f};
  u32x4 RA[NTL][4], RB[4];
  auto load_A = [&](int mi, int roff) {
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      const unsigned v = lane_base + (unsigned)((tci[mi] * 16 + 8 * (J >> 1)) * PP * 2 + (J & 1) * 32);
      if (MODE != 2) RA[mi][J] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, v, roff + tdx2[mi], 0));
    }
  };
  auto load_B = [&](int r) {
    const int roff = r < r1 ? r * (a.W * C * 2) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      RB[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, r < r1 ? dsrc : 0x80000000u, roff + i * 32 * C * 2, 0));
  };
  auto store_B = [&](bf16_t* Bs) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(Bs + ddst + (64 * (i >> 1) + 4 * (i & 1)) * PITCH) = RB[i];
  };
  if (MODE == 2) {
#pragma unroll
    for (int mi = 0; mi < NTL; ++mi)
#pragma unroll
      for (int J = 0; J < 4; ++J) RA[mi][J] = u32x4{lane_base, 0x3f803f80u, (unsigned)mi, (unsigned)J};
  }
  load_B(r0);
  store_B(lds);
  load_B(r0 + 1);
  __builtin_amdgcn_sched_barrier(0);
  {
    const int ro = row_off(r0);
#pragma unroll
    for (int mi = 0; mi < NTL; ++mi) load_A(mi, ro);
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
          for (int c4 = 0; c4 < 4; ++c4) {
            if (MODE != 1) acc[mi][hf * 4 + c4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[c4], __builtin_bit_cast(bf16x8, RA[mi][rg]), acc[mi][hf * 4 + c4], 0, 0, 0);
            else { asm volatile("" ::"v"(RA[mi][rg])); asm volatile("" ::"v"(fb[c4])); }
          }
      }
  };
  for (int r = r0; r < r1; ++r) {
    const int t = r - r0;
    const bf16_t* Bs = lds + (t & 1) * TILE;
    __syncthreads();
    store_B(lds + ((t + 1) & 1) * TILE);
    load_B(r + 2);
    const int ro = row_off(r + 1 < r1 ? r + 1 : r);
    __builtin_amdgcn_sched_barrier(0);
    pass(Bs, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    load_A(0, ro);
    load_A(1, ro);
    __builtin_amdgcn_sched_barrier(0);
    pass(Bs, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
    load_A(2, ro);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
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

// helpers: transposed padded copy, partial fold, naive reference
__global__ void make_xT(const bf16_t* x, bf16_t* xT, int B, int H, int W) {
  const int HP = H + 2 * PAD, WP = W + 2 * PAD;
  const int64_t n = (int64_t)B * H * W * C, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ci = (int)(i % C);
  const int64_t p = i / C;
  const int xx = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
  xT[(int64_t)ci * B * HP * WP + ((int64_t)b * HP + y + PAD) * WP + xx + PAD] = x[i];
}
__global__ void fold(const float* partial, float* dw, int ksplit, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < ksplit; ++k) s += partial[(size_t)k * n + i];
  dw[i] = s;
}
// reference: one block per (tap, co), threads over ci, pixels in a loop (sampled K range for speed: rows [0, refrows))
__global__ void ref_kernel(const bf16_t* dout, const bf16_t* x, float* dw, int B, int H, int W, int ntaps, const int* dyx) {
  const int t = blockIdx.x / C, co = blockIdx.x % C, ci = threadIdx.x;
  const int dy = dyx[t], dx = dyx[MAXT + t];
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y) {
      const int sy = y + dy;
      if (sy < 0 || sy >= H) continue;
      for (int xx = 0; xx < W; ++xx) {
        const int sx = xx + dx;
        if (sx < 0 || sx >= W) continue;
        s += bf2f(dout[(((int64_t)b * H + y) * W + xx) * C + co].v) * bf2f(x[(((int64_t)b * H + sy) * W + sx) * C + ci].v);
      }
    }
  dw[((size_t)t * C + co) * C + ci] = s;
}

static uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, H = 128, W = 128, NT = 17;
  const int HP = H + 2 * PAD, WP = W + 2 * PAD;
  const size_t M = (size_t)B * H * W, n = M * C, nT = (size_t)C * B * HP * WP;
  int dy[MAXT] = {0}, dx[MAXT] = {0}, k = 0;
  dy[k] = 0; dx[k++] = 0;
  for (int d = 6; d <= 12; d += 6)
    for (int iy = -1; iy <= 1; ++iy)
      for (int ix = -1; ix <= 1; ++ix)
        if (iy || ix) { dy[k] = iy * d; dx[k++] = ix * d; }
  std::vector<uint16_t> hx(n), hd(n);
  srand(2);
  for (auto& v : hx) v = h_f2bf((rand() / (float)RAND_MAX - 0.4f) * 2.f);
  for (auto& v : hd) v = h_f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.02f);
  bf16_t *dx_, *dd, *dxT; float *dpart, *ddw, *dref; int* ddyx;
  const int nrows = B * H;
  // tap groups: the taps of one dy, three at a time (a short group repeats its first tap in a slot marked -1)
  int gtap[8][3], ntg = 0;
  for (int d = -12; d <= 12; d += 6) {
    int list[8], m = 0;
    for (int t = 0; t < NT; ++t) if (dy[t] == d) list[m++] = t;
    for (int i = 0; i < m; i += 3, ++ntg)
      for (int j = 0; j < 3; ++j) gtap[ntg][j] = i + j < m ? list[i + j] : -1;
  }
  int ksplit = argc > 2 ? atoi(argv[2]) : 42;
  int rpb = (nrows + ksplit - 1) / ksplit;
  ksplit = (nrows + rpb - 1) / rpb;
  CK(hipMalloc(&dx_, n * 2)); CK(hipMalloc(&dd, n * 2)); CK(hipMalloc(&dxT, nT * 2));
  CK(hipMemcpy(dx_, hx.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, hd.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dxT, 0, nT * 2));
  const int ndw = NT * C * C;
  CK(hipMalloc(&dpart, (size_t)ksplit * ndw * 4)); CK(hipMalloc(&ddw, ndw * 4)); CK(hipMalloc(&dref, ndw * 4)); CK(hipMalloc(&ddyx, 2 * MAXT * 4));
  int hdyx[2 * MAXT]; for (int t = 0; t < MAXT; ++t) { hdyx[t] = dy[t]; hdyx[MAXT + t] = dx[t]; }
  CK(hipMemcpy(ddyx, hdyx, sizeof(hdyx), hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](const char* name, auto&& fn, int nrep) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < nrep; ++i) fn();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / nrep, fl = 2.0 * M * C * C * NT;
    printf("time  %-34s %8.1f us   %6.0f TFLOP/s (%.3f of 2.5 PF)\n", name, us, fl / us / 1e6, fl / us / 1e6 / 2500.0);
    CK(hipGetLastError());
  };
  timed("make_xT (stand-in for the producer)", [&] { make_xT<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(dx_, dxT, B, H, W); }, 5);
  WArgs a; a.dout = dd; a.xT = dxT; a.partial = dpart; a.B = B; a.H = H; a.W = W; a.ntaps = NT; a.ksplit = ksplit; a.rpb = rpb;
  for (int t = 0; t < MAXT; ++t) { a.dy[t] = dy[t]; a.dx[t] = dx[t]; }
  a.ngroups = ntg; a.per = (ntg * ksplit + 7) / 8;
  for (int g = 0; g < 8; ++g) for (int j = 0; j < 3; ++j) a.gtap[g][j] = g < ntg ? gtap[g][j] : -1;
  const int grid = a.per * 8;
  printf("grid %d workgroups (%d tap groups x %d K-ranges of %d image rows), partials %.1f MB\n", ntg * ksplit, ntg, ksplit, rpb, (double)ksplit * ndw * 4 / 1e6);
  wgrad_rtr_kernel<0><<<grid, 512, 0, st>>>(a);
  fold<<<(ndw + 255) / 256, 256, 0, st>>>(dpart, ddw, ksplit, ndw);
  ref_kernel<<<NT * C, C, 0, st>>>(dd, dx_, dref, B, H, W, NT, ddyx);
  CK(hipStreamSynchronize(st)); CK(hipGetLastError());
  std::vector<float> hw(ndw), hr(ndw);
  CK(hipMemcpy(hw.data(), ddw, ndw * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), dref, ndw * 4, hipMemcpyDeviceToHost));
  double num = 0, den = 0, mx = 0; int worst = 0;
  for (int i = 0; i < ndw; ++i) { const double d = (double)hw[i] - hr[i]; num += d * d; den += (double)hr[i] * hr[i]; if (fabs(d) > mx) { mx = fabs(d); worst = i; } }
  printf("check: rel err %.3e  max |diff| %.4g at tap %d co %d ci %d (ref %.5g got %.5g)\n", sqrt(num / den), mx, worst / (C * C), (worst / C) % C, worst % C, hr[worst], hw[worst]);
  for (int rep = 0; rep < 2; ++rep) {
    timed("wgrad_rtr (stage 1, partials)", [&] { wgrad_rtr_kernel<0><<<grid, 512, 0, st>>>(a); }, 20);
    timed("fold of the partials", [&] { fold<<<(ndw + 255) / 256, 256, 0, st>>>(dpart, ddw, ksplit, ndw); }, 20);
    // the shipping stage 1 on the same operands (its own workspace, reduction deferred into a job struct that is never run)
    {
      const int ks3[3] = {1, 3, 3};
      int src[MAXT], kpos[MAXT];
      src[0] = 0; kpos[0] = 0; int q = 1;
      for (int s = 1; s <= 2; ++s) for (int kp = 0; kp < 9; ++kp) if (kp != 4) { src[q] = s; kpos[q] = kp; ++q; }
      static float *w0 = nullptr, *w1, *w2, *ws;
      static rssf_wgrad_reduce_job job;
      if (!w0) {
        CK(hipMalloc(&w0, C * C * 4)); CK(hipMalloc(&w1, C * C * 9 * 4)); CK(hipMalloc(&w2, C * C * 9 * 4));
        CK(hipMalloc(&ws, (size_t)rssf_conv_wgrad_workspace_elems(B, H, W, C, C, NT) * 4));
      }
      timed("librssf rssf_conv_wgrad (stage 1)", [&] {
        if (rssf_conv_wgrad(dd, dx_, w0, w1, w2, ks3, 3, src, kpos, nullptr, nullptr, ws, B, H, W, C, H, W, C, 1, NT, dy, dx, &job, RSSF_BF16, st)) {
          fprintf(stderr, "rssf_conv_wgrad: %s\n", rssf_last_error()); exit(1); }
      }, 20);
    }
  }
  timed("wgrad_rtr no MFMA", [&] { wgrad_rtr_kernel<1><<<grid, 512, 0, st>>>(a); }, 10);
  timed("wgrad_rtr no xT loads", [&] { wgrad_rtr_kernel<2><<<grid, 512, 0, st>>>(a); }, 10);
  return 0;
}
