// Stand-alone prototype + micro-benchmark (no PyTorch): MlpDWBN's 17-tap dilated convolution (ffn_block.py:226-228,250-257) at the
// benchmark geometry with the ACTIVATION operand loaded straight from global memory in MFMA fragment layout (no LDS staging of the
// pixels: each wave owns its 64 pixels alone) and only the weight slab shared through LDS.  Links against librssf.so to time the
// shipping gather kernel in the same process on the same buffers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mlp_direct_proto.hip -Lrepresentationlearning_amd/lib -lrssf -o gpurun_out/mlp_proto
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "../representationlearning_amd/csrc/common.hip.h"
using namespace rssf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int MAXT = 19;
struct PArgs {
  const bf16_t* in; const bf16_t* wpk; bf16_t* out; const float* bias; float* stats;
  int B, H, W, ntaps, per;      // C = 128 in and out
  int dy[MAXT], dx[MAXT];
};

// LDS image of one [128 rows][32 k] weight sub-tile (the layout of conv_fwd.hip's StageLay<bf16>: conflict-free ds_read_b128 / ds_write_b128)
__device__ __forceinline__ int sl_off(int r, int q) { return (q >> 1) * (128 * 16 + 32) + r * 16 + (q & 1) * 8; }
constexpr int SL_ELEMS = 2 * (128 * 16 + 32);

// NW waves (64 pixels x 128 channels each), KC channels per barrier step (32 or 64), DEPTH = activation fragment groups in flight,
// MODE 0 full, 1 no MFMA (loads + LDS only), 2 no activation loads (MFMA + weights only)
template <int NW, int KC, int DEPTH, int MODE>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_adirect_kernel(PArgs a) {
  constexpr int NT = NW * 64, NH = KC / 32, C = 128, GPT = C / 32;       // GPT fragment groups per tap
  constexpr int SPT = GPT / NH;                                            // barrier steps per tap
  constexpr int BCH = (128 * 4 * NH) / NT;                                 // 16-byte weight chunks per thread and step
  static_assert(BCH >= 1 && GPT % DEPTH == 0 && SPT % 2 == 0, "shape");
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * NH * SL_ELEMS > NW * 16 * 136 ? 2 * NH * SL_ELEMS : NW * 16 * 136];
  __shared__ float sred[NW][2][C];
  __shared__ int s_toff[MAXT + 2];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned q = (blockIdx.x & 7u) * (unsigned)a.per + (blockIdx.x >> 3);      // XCD-major numbering
  const int M = a.B * a.H * a.W;
  const int m0 = (int)q * (NW * 64);
  if (m0 >= M) return;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, (int)((int64_t)M * C * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wpk), 0, a.ntaps * C * C * 2, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  if (tid < MAXT + 2) s_toff[tid] = tid < a.ntaps ? (a.dy[tid] * a.W + a.dx[tid]) * C * 2 : 0;
  unsigned abase[4], amask[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wave * 64 + mi * 16 + l15;
    const int x = m % a.W, y = (m / a.W) % a.H;
    abase[mi] = (unsigned)m * (C * 2) + grp * (16 * NH);
    unsigned mk = 0;
    for (int t = 0; t < a.ntaps; ++t) {
      const int sy = y + a.dy[t], sx = x + a.dx[t];
      if (m < M && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W) mk |= 1u << t;
    }
    amask[mi] = mk;
  }
  // weight staging: chunk c -> sub-tile h, row, k-slot q' ; global bytes = row * 256 + kc * KC * 2 + q' * 16 * NH + h * 16
  unsigned bbase[BCH]; int bdst[BCH];
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int c = tid + i * NT;
    const int h = c / 512, r = (c % 512) / 4, qq = c % 4;
    bbase[i] = (unsigned)(r * (C * 2) + qq * (16 * NH) + h * 16);
    bdst[i] = h * SL_ELEMS + sl_off(r, qq);
  }
  __syncthreads();

  f32x4 acc[4][8];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[mi][ct] = {0.f, 0.f, 0.f, 0.f};

  bf16x8 A[DEPTH][4];
  const int ngroups = a.ntaps * GPT;
  // activation fragment group G = tap * GPT + gi: channels (gi / NH) * KC + {grp * 8 NH + h * 8 + e}, h = gi % NH
  auto load_A = [&](bf16x8 (&dst)[4], int t, int gi, int toff) {
    const int cb = (gi / NH) * (KC * 2) + (gi % NH) * 16;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const bool ok = (amask[mi] >> t) & 1u;
      if (MODE != 2) dst[mi] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? abase[mi] + (unsigned)(toff + cb) : OOB, 0, 0));
      else dst[mi] = __builtin_bit_cast(bf16x8, u32x4{abase[mi], amask[mi], (unsigned)toff, 0x3f803f80u});
    }
  };
  auto load_A1 = [&](bf16x8& dst, int mi, int t, int gi, int toff) {
    const int cb = (gi / NH) * (KC * 2) + (gi % NH) * 16;
    const bool ok = (amask[mi] >> t) & 1u;
    if (MODE != 2) dst = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? abase[mi] + (unsigned)(toff + cb) : OOB, 0, 0));
  };
  auto load_B = [&](u32x4 (&dst)[BCH], int t, int st) {      // step st of tap t (taps past the end: zeros)
    const int woff = t < a.ntaps ? t * (C * C * 2) + st * (KC * 2) : 0;
#pragma unroll
    for (int i = 0; i < BCH; ++i)
      dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, t < a.ntaps ? bbase[i] : OOB, woff, 0));
  };
  auto store_B = [&](const u32x4 (&src)[BCH], bf16_t* Bs) {
#pragma unroll
    for (int i = 0; i < BCH; ++i) *reinterpret_cast<u32x4*>(Bs + bdst[i]) = src[i];
  };
  bf16_t* Bs0 = lds;
  bf16_t* Bs1 = lds + NH * SL_ELEMS;

  // Pipeline.  Weights: tile s+2 is requested (global -> registers) at the start of step s, written to the other LDS buffer at the
  // start of step s+1 (after the barrier that retires that buffer's readers), read as fragments in step s+2: ONE register set.
  // Activations: a fragment (mi) is re-requested for the group DEPTH ahead right after its eight MFMAs; sched_barriers pin the
  // requests there (left alone the scheduler sinks every load in front of its use and the prefetch distance collapses).
  int toff_cur = s_toff[0], toff_nxt = s_toff[1];
  // (the order of the requests matches the steady state's - weights of the step after next, then the activation groups - so that
  //  the wait counts the compiler derives at the loop header are the same from the pre-header and from the back edge)
  u32x4 RBs[BCH];
  load_B(RBs, 0, 0);
  store_B(RBs, Bs0);
  load_B(RBs, 0, 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_A(A[d], 0, d, toff_cur);
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0; t < a.ntaps; ++t) {
    const int toff_n2 = s_toff[t + 2 < MAXT + 2 ? t + 2 : MAXT + 1];
#pragma unroll
    for (int st = 0; st < SPT; ++st) {
      bf16_t* Bs = (st & 1) ? Bs1 : Bs0;
      bf16_t* Bn = (st & 1) ? Bs0 : Bs1;
      __syncthreads();                   // tile (t, st) complete in Bs; every wave is done with Bn's previous tile
      store_B(RBs, Bn);                  // tile (t, st + 1)
      { const int s2 = st + 2; load_B(RBs, s2 < SPT ? t : t + 1, s2 < SPT ? s2 : s2 - SPT); }
      bf16x8 fb[8];
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bs + sl_off(ct * 16 + l15, grp));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int gi = st * NH + h, slot = gi % DEPTH;
        const int gn = gi + DEPTH;            // the group that takes this slot next
        const int tn = gn < GPT ? t : t + 1, gin = gn < GPT ? gn : gn - GPT;
        const int toffn = gn < GPT ? toff_cur : toff_nxt;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
          for (int ct = 0; ct < 8; ++ct) {
            if (MODE != 1) acc[mi][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ct], A[slot][mi], acc[mi][ct], 0, 0, 0);
            else { asm volatile("" ::"v"(A[slot][mi])); asm volatile("" ::"v"(fb[ct])); }
            if (mi == 3 && h + 1 < NH) fb[ct] = *reinterpret_cast<const bf16x8*>(Bs + (h + 1) * SL_ELEMS + sl_off(ct * 16 + l15, grp));
          }
          load_A1(A[slot][mi], mi, tn, gin, toffn);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    toff_cur = toff_nxt; toff_nxt = toff_n2;
  }
  __syncthreads();

  // ---- epilogue: + bias, bf16, 16 pixels at a time through a wave-private LDS tile -> 16-byte row stores; BatchNorm statistics --------
  bf16_t* Cs = lds + wave * (16 * 136);
  const int cc = lane & 15;                 // this lane's 16-byte channel chunk in the store phase
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  f32x4 bv[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) bv[ct] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ct * 16 + grp * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const f32x4 v = acc[mi][ct] + bv[ct];
      typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
      const u32x2_t pk = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(Cs + l15 * 136 + ct * 16 + grp * 4) = pk;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = (lane >> 4) + j * 4;
      Vec<bf16_t> v;
      v.load(Cs + px * 136 + cc * 8);
      const int m = m0 + wave * 64 + mi * 16 + px;
      v.store(a.out + (size_t)m * C + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = v.get(e); s1[e] += f; s2[e] = fmaf(f, f, s2[e]); }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (a.stats) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = rows_reduce<OpSum>(s1[e]); s2[e] = rows_reduce<OpSum>(s2[e]); }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sred[wave][0][cc * 8 + e] = s1[e]; sred[wave][1][cc * 8 + e] = s2[e]; }
    }
    __syncthreads();
    if (tid < 2 * C) {
      float tsum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tsum += sred[w][tid / C][tid % C];
      atomicAdd(a.stats + (size_t)(blockIdx.x % 16) * 2 * C + tid, tsum);
    }
  }
}


// ---- variant 2: coalesced activation loads, transposed into MFMA fragment layout IN REGISTERS (DPP exchanges), no LDS for pixels ----
// A wave owns 64 consecutive pixels of one image row.  Per tap and 16-pixel tile: four 1 KB loads (instruction J: pixels 4J..4J+3 of
// the tile, the whole 256-byte channel row of each: lane = [k5 k4 | p1 p0 | k3 k2], 16 bytes = dwords [k1 k0]); two exchange stages
// (lane bit 1 <-> J bit 1, lane bit 0 <-> J bit 0) leave lane = [k5 k4 | p1 p0 p3 p2] and register J' = [k3 k2]: register J' IS the
// operand of K-step J' (channels 32 G + 8 J' + e for lane group G).  The image row of a tap is a buffer descriptor of its own
// (base = row start, num_records = row bytes, or 0 when the row lies outside the image): x + dx outside the row -> the load returns
// zeros by the hardware range check, no masks.  Weights: one [128][128] tile per tap in LDS (XOR-swizzled 16-byte chunks), one
// barrier per tap.
constexpr int WT_PITCH = 144;                 // weight row pitch in LDS (elements): 256 + 32 bytes
constexpr int WT_ELEMS = 128 * WT_PITCH;
// LDS image of a tap's [128][128] weight tile: row pitch 288 bytes, 16-byte chunk position 4 kappa + G holds channels 32 G + 8 kappa ..:
// a fragment read (rows ct * 16 + l15, lane group G, K-step kappa) is lane_const + ct * 4608 + kappa * 64 bytes (immediates) and is
// conflict-free (the 16-lane service groups of ds_read_b128 pair rows {0-3,12-15} of G with rows {4-11} of G ^ 1: chunk slots
// 2 r + G mod 16 are the evens / the odds).
template <int CTRL> __device__ __forceinline__ uint32_t dppu(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ void lane_exchange(u32x4& a, u32x4& b, bool hi) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t ta = dppu<CTRL>(b[d]), tb = dppu<CTRL>(a[d]);
    const uint32_t na = hi ? ta : a[d], nb = hi ? b[d] : tb;
    a[d] = na; b[d] = nb;
  }
}
// the same exchange as ONE v_cndmask_b32_dpp per dword (VOP2 with a DPP source: dst = vcc ? src1 : perm(src0)); the compiler does not
// fold v_mov_b32_dpp into the select.  lo / hi: 64-bit lane masks of the exchanged lane bit clear / set.
#define RSSF_XCHG_ASM(QP)                                                                                                     \
  asm volatile("s_nop 1\n\ts_mov_b64 vcc, %[lo]\n\t"                                                                          \
               "v_cndmask_b32_dpp %[n0], %[b0], %[a0], vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"                    \
               "v_cndmask_b32_dpp %[n1], %[b1], %[a1], vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"                    \
               "v_cndmask_b32_dpp %[n2], %[b2], %[a2], vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"                    \
               "v_cndmask_b32_dpp %[n3], %[b3], %[a3], vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"                    \
               "s_mov_b64 vcc, %[hi]\n\t"                                                                                     \
               "v_cndmask_b32_dpp %[m0], %[a0], %[b0], vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"                    \
               "v_cndmask_b32_dpp %[m1], %[a1], %[b1], vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"                    \
               "v_cndmask_b32_dpp %[m2], %[a2], %[b2], vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"                    \
               "v_cndmask_b32_dpp %[m3], %[a3], %[b3], vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf"                         \
               : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [m0] "=&v"(m[0]), [m1] "=&v"(m[1]),    \
                 [m2] "=&v"(m[2]), [m3] "=&v"(m[3])                                                                             \
               : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), \
                 [b3] "v"(b[3]), [lo] "s"(lo), [hi] "s"(hi)                                                                     \
               : "vcc")
template <int BIT> __device__ __forceinline__ void lane_exchange_asm(u32x4& a, u32x4& b) {
  constexpr uint64_t hi = BIT == 1 ? 0xCCCCCCCCCCCCCCCCull : 0xAAAAAAAAAAAAAAAAull, lo = ~hi;
  uint32_t n[4], m[4];
  if constexpr (BIT == 1) RSSF_XCHG_ASM("[2,3,0,1]"); else RSSF_XCHG_ASM("[1,0,3,2]");
  a = u32x4{n[0], n[1], n[2], n[3]};
  b = u32x4{m[0], m[1], m[2], m[3]};
}
// MODE 0 full, 1 no MFMA, 2 no activation loads, 3 no transposition, 4 no barrier in the loop (wrong results: timing only),
// 5 transposition through v_mov_b32_dpp + v_cndmask_b32 (compiler-generated)
// LP = 1: "pair" loads - lane = [k5 k4 | p2 p1 p0 | k2] (two lanes per pixel and load: 32 contiguous bytes), load J = [p3 k3]: ONE
// exchange stage (lane bit 0 <-> p3) at twice the lines per load instruction
template <int MODE, int LP = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_rtr_kernel(PArgs a) {
  constexpr int NW = 8, C = 128;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * WT_ELEMS];
  __shared__ float sred[NW][2][C];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned q = (blockIdx.x & 7u) * (unsigned)a.per + (blockIdx.x >> 3);
  const int M = a.B * a.H * a.W;
  const int m0 = (int)q * (NW * 64);
  if (m0 >= M) return;
  const int mw = m0 + wave * 64;                                  // wave-uniform: 64 consecutive pixels of one image row (W % 64 == 0)
  const int x0 = mw % a.W, yrow = (mw / a.W) % a.H, img = mw / (a.W * a.H);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wpk), 0, a.ntaps * C * C * 2, 0x00020000);
  const unsigned lane_base = LP ? (unsigned)((x0 + ((lane >> 1) & 7)) * (C * 2) + grp * 64 + (lane & 1) * 16)
                                : (unsigned)((x0 + ((lane >> 2) & 3)) * (C * 2) + grp * 64 + (lane & 3) * 16);
  // weight staging: thread -> row tid / 16 (+ 32 i), LDS chunk position qp = tid % 16 = 4 kappa + G, i.e. global chunk 4 G + kappa
  const int qp = tid & 15;
  const unsigned bsrc = (unsigned)((tid >> 4) * (C * 2) + ((((qp & 3) << 2) | (qp >> 2)) << 4));
  const int bdst = (tid >> 4) * WT_PITCH + qp * 8;
  const int foff = l15 * WT_PITCH + grp * 8;                      // + ct * 16 * WT_PITCH + kappa * 32 elements

  f32x4 acc[4][8];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[mi][ct] = {0.f, 0.f, 0.f, 0.f};
  u32x4 RA[4][4];                  // [tile mi][load J], after the exchanges [mi][K-step]
  u32x4 RB[4];
  auto row_rsrc = [&](int t) {
    const int r = yrow + a.dy[t < a.ntaps ? t : 0];
    const bool ok = t < a.ntaps && r >= 0 && r < a.H;
    const bf16_t* p = a.in + (size_t)((img * a.H + (ok ? r : 0)) * a.W) * C;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p), 0, ok ? a.W * C * 2 : 0, 0x00020000);
  };
  auto load_A = [&](int mi, const __amdgpu_buffer_rsrc_t& rs, int dxb) {
    const unsigned v = lane_base + (unsigned)(mi * 16 * C * 2 + dxb);
#pragma unroll
    for (int J = 0; J < 4; ++J)
      if (MODE != 2) RA[mi][J] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, v + (unsigned)(LP ? (J >> 1) * 8 * C * 2 + (J & 1) * 32 : J * 4 * C * 2), 0, 0));
  };
  auto load_B = [&](int t) {
    const int woff = t < a.ntaps ? t * (C * C * 2) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      RB[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, t < a.ntaps ? bsrc : 0x80000000u, woff + i * 32 * C * 2, 0));
  };
  auto store_B = [&](bf16_t* Bs) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(Bs + bdst + i * 32 * WT_PITCH) = RB[i];
  };
  if (MODE == 2) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int J = 0; J < 4; ++J) RA[mi][J] = u32x4{lane_base, 0x3f803f80u, (unsigned)mi, (unsigned)J};
  }
  load_B(0);
  store_B(lds);
  load_B(1);
  __builtin_amdgcn_sched_barrier(0);
  {
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(0);
    const int dxb = a.dx[0] * (C * 2);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) load_A(mi, rs, dxb);
  }
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0; t < a.ntaps; ++t) {
    const bf16_t* Bs = lds + (t & 1) * WT_ELEMS;
    if (MODE != 4) __syncthreads();
    store_B(lds + ((t + 1) & 1) * WT_ELEMS);
    load_B(t + 2);
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(t + 1);
    const int dxb = (t + 1 < a.ntaps ? a.dx[t + 1] : 0) * (C * 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      if (MODE != 3) {
        const bool hi1 = lane & 2, hi0 = lane & 1;
#pragma unroll
        for (int mi = 2 * pr; mi < 2 * pr + 2; ++mi) {
          if (LP) {
            lane_exchange<0xB1>(RA[mi][0], RA[mi][2], hi0);
            lane_exchange<0xB1>(RA[mi][1], RA[mi][3], hi0);
          } else if (MODE == 5) {
            lane_exchange<0x4E>(RA[mi][0], RA[mi][2], hi1);
            lane_exchange<0x4E>(RA[mi][1], RA[mi][3], hi1);
            lane_exchange<0xB1>(RA[mi][0], RA[mi][1], hi0);
            lane_exchange<0xB1>(RA[mi][2], RA[mi][3], hi0);
          } else {
            lane_exchange_asm<1>(RA[mi][0], RA[mi][2]);
            lane_exchange_asm<1>(RA[mi][1], RA[mi][3]);
            lane_exchange_asm<0>(RA[mi][0], RA[mi][1]);
            lane_exchange_asm<0>(RA[mi][2], RA[mi][3]);
          }
        }
      }
#pragma unroll
      for (int kp = 0; kp < 4; ++kp)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          bf16x8 fb[4];
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) fb[c4] = *reinterpret_cast<const bf16x8*>(Bs + foff + (hf * 4 + c4) * 16 * WT_PITCH + kp * 32);
#pragma unroll
          for (int mi = 2 * pr; mi < 2 * pr + 2; ++mi)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
              const int ct = hf * 4 + c4;
              constexpr int dummy = 0; (void)dummy;
              const int rg = LP ? ((kp & 1) << 1) | (kp >> 1) : kp;        // register that holds K-step kp
              if (MODE != 1) acc[mi][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[c4], __builtin_bit_cast(bf16x8, RA[mi][rg]), acc[mi][ct], 0, 0, 0);
              else { asm volatile("" ::"v"(RA[mi][rg])); asm volatile("" ::"v"(fb[c4])); }
            }
        }
      load_A(2 * pr, rs, dxb);
      load_A(2 * pr + 1, rs, dxb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();

  // epilogue: column col of an accumulator tile is pixel 4 (col & 3) + (col >> 2) of the 16-pixel tile
  bf16_t* Cs = lds + wave * (16 * 136);
  const int cc = lane & 15;
  const int prow = LP ? 8 * (l15 & 1) + (l15 >> 1) : 4 * (l15 & 3) + (l15 >> 2);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  f32x4 bv[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) bv[ct] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ct * 16 + grp * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
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
      const int m = mw + mi * 16 + px;
      v.store(a.out + (size_t)m * C + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = v.get(e); s1[e] += f; s2[e] = fmaf(f, f, s2[e]); }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (a.stats) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = rows_reduce<OpSum>(s1[e]); s2[e] = rows_reduce<OpSum>(s2[e]); }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sred[wave][0][cc * 8 + e] = s1[e]; sred[wave][1][cc * 8 + e] = s2[e]; }
    }
    __syncthreads();
    if (tid < 2 * C) {
      float tsum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tsum += sred[w][tid / C][tid % C];
      atomicAdd(a.stats + (size_t)(blockIdx.x % 16) * 2 * C + tid, tsum);
    }
  }
}
template <int MODE, int LP = 0>
static void launch_rtr(PArgs a, hipStream_t st) {
  const int M = a.B * a.H * a.W, total = (M + 511) / 512;
  a.per = (total + 7) / 8;
  conv_rtr_kernel<MODE, LP><<<a.per * 8, 512, 0, st>>>(a);
}

// naive reference (fp32 accumulate)
__global__ void ref_kernel(PArgs a, float* out) {
  const int M = a.B * a.H * a.W, C = 128;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C) return;
  const int m = idx / C, co = idx % C, x = m % a.W, y = (m / a.W) % a.H;
  float s = a.bias ? a.bias[co] : 0.f;
  for (int t = 0; t < a.ntaps; ++t) {
    const int sy = y + a.dy[t], sx = x + a.dx[t];
    if (sy < 0 || sy >= a.H || sx < 0 || sx >= a.W) continue;
    const bf16_t* ip = a.in + (size_t)(m + a.dy[t] * a.W + a.dx[t]) * C;
    const bf16_t* wp = a.wpk + ((size_t)t * C + co) * C;
    for (int ci = 0; ci < C; ++ci) s += bf2f(ip[ci].v) * bf2f(wp[ci].v);
  }
  out[idx] = s;
}

static uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float h_bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int NW, int KC, int DEPTH, int MODE>
static void launch(PArgs a, hipStream_t st) {
  const int M = a.B * a.H * a.W, total = (M + NW * 64 - 1) / (NW * 64);
  a.per = (total + 7) / 8;
  conv_adirect_kernel<NW, KC, DEPTH, MODE><<<a.per * 8, NW * 64, 0, st>>>(a);
}
typedef void (*launch_fn)(PArgs, hipStream_t);

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, H = 128, W = 128, C = 128, NT = 17;
  const size_t M = (size_t)B * H * W, n = M * C;
  int dy[MAXT], dx[MAXT], k = 0;
  dy[k] = 0; dx[k++] = 0;
  for (int d = 6; d <= 12; d += 6)
    for (int iy = -1; iy <= 1; ++iy)
      for (int ix = -1; ix <= 1; ++ix)
        if (iy || ix) { dy[k] = iy * d; dx[k++] = ix * d; }
  const int NRING = 6;
  std::vector<uint16_t> hin(n), hw((size_t)NT * C * C);
  srand(1);
  for (auto& v : hin) v = h_f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& v : hw) v = h_f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
  std::vector<float> hb(C);
  for (auto& v : hb) v = rand() / (float)RAND_MAX - 0.5f;
  bf16_t* din[NRING]; bf16_t* dout[NRING]; bf16_t* dw; float *dbias, *dstats, *dref;
  for (int i = 0; i < NRING; ++i) { CK(hipMalloc(&din[i], n * 2)); CK(hipMalloc(&dout[i], n * 2)); CK(hipMemcpy(din[i], hin.data(), n * 2, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&dbias, C * 4)); CK(hipMemcpy(dbias, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dstats, 16 * 2 * C * 4)); CK(hipMemset(dstats, 0, 16 * 2 * C * 4));
  CK(hipMalloc(&dref, n * 4));
  PArgs a; a.wpk = dw; a.bias = dbias; a.stats = dstats; a.B = B; a.H = H; a.W = W; a.ntaps = NT; a.per = 0;
  for (int t = 0; t < NT; ++t) { a.dy[t] = dy[t]; a.dx[t] = dx[t]; }
  hipStream_t st; CK(hipStreamCreate(&st));

  // ---- correctness: every full-MFMA variant against the naive kernel (and the shipping kernel likewise) ---------------------------
  a.in = din[0]; a.out = dout[0];
  ref_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(a, dref);
  CK(hipStreamSynchronize(st));
  std::vector<float> href(n); CK(hipMemcpy(href.data(), dref, n * 4, hipMemcpyDeviceToHost));
  std::vector<uint16_t> hout(n);
  auto check = [&](const char* name) {
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hout.data(), dout[0], n * 2, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0; size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
      const double e = fabs((double)h_bf2f(hout[i]) - href[i]);
      if (e > maxerr) maxerr = e;
      if (fabs(href[i]) > maxref) maxref = fabs(href[i]);
      if (e > 0.02 + 0.01 * fabs(href[i])) ++bad;
    }
    std::vector<float> hs(16 * 2 * C); CK(hipMemcpy(hs.data(), dstats, hs.size() * 4, hipMemcpyDeviceToHost));
    double ssum = 0, rsum = 0;
    for (int s = 0; s < 16; ++s) for (int c = 0; c < C; ++c) ssum += hs[(s * 2) * C + c];
    for (size_t i = 0; i < n; ++i) rsum += href[i];
    printf("check %-28s max|err| %.4f (max|ref| %.2f) bad %zu   stats sum %.1f vs %.1f\n", name, maxerr, maxref, bad, ssum, rsum);
    CK(hipMemset(dstats, 0, 16 * 2 * C * 4));
  };
  struct Var { const char* name; launch_fn fn; };
  const Var vars[] = {
      {"adirect NW4 KC32 D2", launch<4, 32, 2, 0>}, {"adirect NW8 KC32 D2", launch<8, 32, 2, 0>}, {"adirect NW8 KC32 D4", launch<8, 32, 4, 0>},
      {"adirect NW4 KC64 D2", launch<4, 64, 2, 0>}, {"adirect NW8 KC64 D2", launch<8, 64, 2, 0>}, {"adirect NW8 KC64 D4", launch<8, 64, 4, 0>},
      {"regtranspose NW8", launch_rtr<0>}, {"regtranspose NW8 mov+sel", launch_rtr<5>}, {"regtranspose NW8 pairs", launch_rtr<5, 1>},
  };
  const Var diag[] = {
      {"rtr noMFMA", launch_rtr<1>}, {"rtr noAload", launch_rtr<2>}, {"rtr noTranspose", launch_rtr<3>}, {"rtr noBarrier", launch_rtr<4>}, {"rtr pairs noMFMA", launch_rtr<1, 1>}, {"rtr pairs noTranspose", launch_rtr<3, 1>},
      {"noMFMA  NW8 KC64 D4", launch<8, 64, 4, 1>}, {"noMFMA  NW8 KC32 D2", launch<8, 32, 2, 1>}, {"noMFMA  NW4 KC64 D2", launch<4, 64, 2, 1>},
      {"noAload NW8 KC64 D4", launch<8, 64, 4, 2>}, {"noAload NW8 KC32 D2", launch<8, 32, 2, 2>}, {"noAload NW4 KC64 D2", launch<4, 64, 2, 2>},
  };
  for (const Var& v : vars) { CK(hipMemsetAsync(dout[0], 0, n * 2, st)); v.fn(a, st); CK(hipGetLastError()); check(v.name); }
  int* ddy; int* ddx_;      // (host arrays are fine for the C ABI: it copies them into the launch arguments)
  (void)ddy; (void)ddx_;
  {
    CK(hipMemsetAsync(dout[0], 0, n * 2, st));
    const int rc = rssf_conv_gather(din[0], dw, dout[0], dbias, dstats, B, H, W, C, H, W, C, 1, 1, NT, dy, dx, RSSF_BF16, st);
    if (rc) { fprintf(stderr, "rssf_conv_gather: %s\n", rssf_last_error()); return 1; }
    check("librssf gather");
  }

  // ---- timing: back-to-back launches, (a) a ring of inputs / outputs (operands miss the L2), (b) one input / output --------------
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flops = 2.0 * M * C * C * NT;
  auto timeit = [&](const char* name, auto&& fn) {
    for (int mode = 0; mode < 2; ++mode) {
      const int nrep = 24;
      for (int i = 0; i < 6; ++i) { a.in = din[mode ? 0 : i % NRING]; a.out = dout[mode ? 0 : i % NRING]; fn(); }
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < nrep; ++i) { a.in = din[mode ? 0 : i % NRING]; a.out = dout[mode ? 0 : i % NRING]; fn(); }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / nrep;
      printf("time  %-28s %s  %7.1f us  %6.0f TFLOP/s executed (%.3f of 2.5 PF; 19-position accounting %.3f)\n", name, mode ? "same buffers" : "ring of 6   ", us,
             flops / us / 1e6, flops / us / 1e6 / 2500.0, flops * 19 / 17 / us / 1e6 / 2500.0);
    }
  };
  for (int rep = 0; rep < 2; ++rep) {
    timeit("librssf gather", [&] { rssf_conv_gather(a.in, dw, a.out, dbias, dstats, B, H, W, C, H, W, C, 1, 1, NT, dy, dx, RSSF_BF16, st); });
    for (const Var& v : vars) timeit(v.name, [&] { v.fn(a, st); });
  }
  for (const Var& v : diag) timeit(v.name, [&] { v.fn(a, st); });
  CK(hipGetLastError());
  return 0;
}
