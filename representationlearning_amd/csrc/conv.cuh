// Implicit-GEMM convolution on channels-last activations (gfx950 MFMA), shared definitions.
#pragma once
#include "common.cuh"

namespace rssf {
namespace cv {

constexpr int MAX_TAPS = 19;   // fused {1x1 + 3x3 dil 6 + 3x3 dil 12} of MlpDWBN (ffn_block.py:226-228,250-257)

// One "tap" = one (dy, dx) displacement with its own [Cout][Cin] weight slab.  out(oy,ox) += W_t * in(oy*s + dy, ox*s + dx)
struct Taps {
  int n;
  int dy[MAX_TAPS];
  int dx[MAX_TAPS];
};

// GEMM-K MFMA (full-rate on gfx950): bf16 16x16x32, f32 16x16x4 (exact fp32 for the parity mode)
template <typename T> struct MmaK;
template <> struct MmaK<bf16_t> {
  static constexpr int KSTEP = 32, KPL = 8, BK = 32;       // BK: channels per LDS stage (64-byte rows)
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag load(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaK<float> {
  static constexpr int KSTEP = 4, KPL = 1, BK = 16;
  typedef float frag;
  static __device__ __forceinline__ frag load(const float* p) { return *p; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

template <typename T> struct LdsPad;
template <> struct LdsPad<bf16_t> { static constexpr int X = 8; };   // +16 B per row: conflict-free ds_read_b128
template <> struct LdsPad<float> { static constexpr int X = 4; };

}  // namespace cv
}  // namespace rssf
