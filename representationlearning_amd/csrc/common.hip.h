// Shared device helpers for the RSSFormer HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/rssf.h"

#define RSSF_WAVE 64

namespace rssf {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// ---- bf16 <-> f32 (bf16 carried as raw uint16) -------------------------------------------------
struct bf16_t { uint16_t v; };

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round-to-nearest-even: the compiler's own fptrunc, which gfx950 lowers to v_cvt_pk_bf16_f32
// (measured on the window-attention kernel: the bit-twiddling form cost ~7 VALU per element and made the kernel
// VALU-issue-bound: 5.9 k VALU instructions per window).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {   // two values -> one dword (lo in bits 0..15)
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float cvt(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(p->v); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { p->v = f2bf(v); }
};

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return Elem<T>::ld(p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { Elem<T>::st(p, v); }

// Vector access: VEC elements (16 bytes for bf16 x8, 16 bytes for f32 x4).  `raw` is a native ext-vector (not HIP's
// uint4/float4 struct-of-union): arrays of Vec then always scalarise into VGPRs (the struct form was seen demoted to
// scratch memory once a staging array was written under a branch).
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  f32x4 raw;
  __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const f32x4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<f32x4*>(p) = raw; }
  __device__ __forceinline__ float get(int i) const { return raw[i]; }
  __device__ __forceinline__ void set(int i, float v) { raw[i] = v; }
  __device__ __forceinline__ void set_all(const float (&v)[4]) { raw = f32x4{v[0], v[1], v[2], v[3]}; }
  __device__ __forceinline__ void clear() { raw = f32x4{0.f, 0.f, 0.f, 0.f}; }
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  u32x4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<u32x4*>(p) = raw; }
  __device__ __forceinline__ float get(int i) const {
    const uint32_t w = raw[i >> 1];
    return (i & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
  }
  __device__ __forceinline__ void set(int i, float v) {
    const uint32_t w = raw[i >> 1], h = f2bf(v);
    raw[i >> 1] = (i & 1) ? ((w & 0x0000ffffu) | (h << 16)) : ((w & 0xffff0000u) | h);
  }
  __device__ __forceinline__ void set_all(const float (&v)[8]) {     // 4 x v_cvt_pk_bf16_f32
    raw = u32x4{f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7])};
  }
  __device__ __forceinline__ void clear() { raw = u32x4{0u, 0u, 0u, 0u}; }
};

// ---- wave reductions ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// The same reductions without LDS traffic: ds_bpermute (what __shfl_xor lowers to) is an LDS instruction with ~100 cycles of
// latency, and these reductions sit on the dependency chain of every softmax row.  Within a 16-lane row: DPP (quad_perm,
// row_half_mirror, row_mirror); across rows: the gfx950 lane-swap instructions.  v_permlane16_swap(a, b) exchanges the odd
// rows of a with the even rows of b, v_permlane32_swap(a, b) the upper half of a with the lower half of b; called with a = b = v
// the two results hold, in every lane, the two values of the lane's xor-16 (xor-32) pair.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
struct OpSum { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };
// Measured (tests/test_gpu_conv.py::test_lane_reductions_without_lds): swap16(a, b) -> a = [a.row0, b.row0, a.row2, b.row2],
// b = [a.row1, b.row1, a.row3, b.row3]; swap32(a, b) -> a = [a.lo, b.lo], b = [a.hi, b.hi].  Issued through inline asm with two
// read-write operands (two distinct registers by construction): handed the SAME value twice, the compiler merges the two results
// of the __builtin_amdgcn_permlane*_swap builtins into one (observed in the ISA: v_add_f32 v7, v7, v7 after the swap).  The
// s_nop covers the VALU-write -> lane-swap-read hazard the compiler would otherwise schedule for.
template <typename OP> __device__ __forceinline__ float xor16_reduce(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return OP::f(a, b);
}
template <typename OP> __device__ __forceinline__ float xor32_reduce(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return OP::f(a, b);
}
// all four rows (lanes l, l^16, l^32, l^48)
template <typename OP> __device__ __forceinline__ float rows_reduce(float v) { return xor32_reduce<OP>(xor16_reduce<OP>(v)); }
// the whole wave
template <typename OP> __device__ __forceinline__ float wave_reduce_dpp(float v) {
  v = OP::f(v, dpp_mov<0xB1>(v));       // quad_perm [1,0,3,2]
  v = OP::f(v, dpp_mov<0x4E>(v));       // quad_perm [2,3,0,1]
  v = OP::f(v, dpp_mov<0x141>(v));      // row_half_mirror: quad q <-> quad q^1
  v = OP::f(v, dpp_mov<0x140>(v));      // row_mirror: half <-> half
  return rows_reduce<OP>(v);
}

// Per-channel constants of the BatchNorm-backward apply, draw = sc*dz + cb*x + cc (= sc*(dz - k1 - xhat*k2)), from the channel's
// sums s1 = sum dz, s2 = sum dz*raw.  ONE definition with explicit fused multiply-adds: bn_bwd_apply_kernel (bn.hip) and the
// weight-gradient kernel that applies on the fly (conv_wgrad.hip) must round alike for bit-identical results.
__device__ __forceinline__ void bn_bwd_constants(float sc, float mean, float istd, float s1, float s2, float n, float& dot, float& cb,
                                                 float& cc) {
  dot = fmaf(-mean, s1, s2) * istd;                        // sum dz * xhat
  const float k1 = s1 / n, k2 = dot / n;
  cb = -(sc * istd) * k2;
  cc = fmaf(-cb, mean, -(sc * k1));
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level) given e = exp(-u*u): one v_rcp + 6 FMA.
// The library erff costs ~35 VALU instructions with two range branches and made the GELU BatchNorm passes (128 channels
// at 1/4 resolution, 33 M elements) VALU-bound at 2.4 TB/s; the exponential is shared with the Gaussian of the gradient.
__device__ __forceinline__ float erf_from_exp(float u, float e) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, fabsf(u), 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  return copysignf(fmaf(-poly, e, 1.f), u);
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float u = x * 0.70710678118654752f;
  return 0.5f * x * (1.0f + erf_from_exp(u, __expf(-u * u)));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float u = x * 0.70710678118654752f;
  const float e = __expf(-u * u);                       // = exp(-x^2/2): also the Gaussian density's exponential
  return 0.5f * (1.0f + erf_from_exp(u, e)) + x * 0.39894228040143268f * e;
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---- MFMA 16x16 tile helpers -----------------------------------------------------------------------
// One wave computes D(16x16) += A(16xK) * B(Kx16).  Operands are addressed "k-contiguous":
//   A element (m,k) at A[m*lda + k],  B element (k,n) at B[n*ldb + k]   (i.e. B is stored as B^T rows).
// C/D layout (all 16x16 shapes, gfx950): lane l holds D[row = (l>>4)*4 + r][col = l&15], r = 0..3.
// A/B fragment layout: lane l supplies row/col (l&15) and k-slots (l>>4)*KPL .. +KPL of each K-step.
template <typename T> struct Mma;
template <> struct Mma<float> {                 // v_mfma_f32_16x16x4_f32 : exact f32, K-step 4, KPL 1
  static constexpr int KSTEP = 4, KPL = 1;
  typedef float frag;
  static __device__ __forceinline__ frag load(const float* p) { return *p; }
  static __device__ __forceinline__ frag zero() { return 0.f; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t> {                // v_mfma_f32_16x16x16_bf16 : K-step 16, KPL 4
  static constexpr int KSTEP = 16, KPL = 4;
  typedef s16x4 frag;
  static __device__ __forceinline__ frag load(const bf16_t* p) { return *reinterpret_cast<const s16x4*>(p); }
  static __device__ __forceinline__ frag zero() { return s16x4{0, 0, 0, 0}; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
};

// D += A[rowA0.., :K] * B[rowB0.., :K]^T ; K multiple of KSTEP; A,B in LDS (or global), k-contiguous.
template <typename T>
__device__ __forceinline__ f32x4 mma_tile(const T* A, int lda, const T* B, int ldb, int K, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const T* pa = A + (lane & 15) * lda + (lane >> 4) * Mma<T>::KPL;
  const T* pb = B + (lane & 15) * ldb + (lane >> 4) * Mma<T>::KPL;
  for (int k = 0; k < K; k += Mma<T>::KSTEP) acc = Mma<T>::mma(Mma<T>::load(pa + k), Mma<T>::load(pb + k), acc);
  return acc;
}

// Pack 4 accumulator values (k-slots (l>>4)*4 + r of a 16-wide K tile) into a B/A fragment for the NEXT
// mma whose K axis is this tile's row axis (register chaining, no LDS).  bf16: one frag; f32: 4 frags.
__device__ __forceinline__ s16x4 pack_bf16x4(f32x4 v) {
  // two v_cvt_pk_bf16_f32.  (The 4-wide __builtin_convertvector lowers to four single conversions + two v_perm_b32 on gfx950:
  // 6 instructions per packed tile, 13 % of the window-attention forward's VALU stream.)
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
  const u32x2_t r = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3])};
  return __builtin_bit_cast(s16x4, r);
}

}  // namespace rssf

// ---- zero-fill as a KERNEL -------------------------------------------------------------------------------------------------
// hipMemsetAsync nodes inside a captured training step stopped taking effect after a device synchronisation between two replays
// (MI355X, ROCm 7.0: the loss accumulators then carried the previous replay's sums - loss 1.75 -> 0.99 / NaN; found with
// tools/graph_vs_eager.py).  Scratch the library must clear is cleared by a kernel node instead.
namespace rssf {
int zero_floats(float* p, int64_t n, hipStream_t st);
}

// ---- host-side error plumbing ------------------------------------------------------------------------
namespace rssf {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
}
#define RSSF_REQUIRE(cond, ...)                          \
  do {                                                   \
    if (!(cond)) {                                       \
      rssf::set_error(__VA_ARGS__);                      \
      return RSSF_ERR_BAD_ARG;                           \
    }                                                    \
  } while (0)
