// Test hooks: exercise the MFMA tile helpers in isolation so layout mistakes show up as a unit-test failure
// rather than inside a fused kernel.
#include "win_attn.cuh"
using namespace rssf;

namespace {
// d[0:256]   = A * B^T through mma_tile (both operands from memory)
// d[256:512] = same product, but B passed through the register-chaining path:
//              E = A*B^T is first produced in C layout, then F = Bsq * E  (Bsq = first 16 cols of B) is computed
//              with E chained as the B operand (k-slot = E's row) -> checks mma_lds_chain's slot convention.
template <typename T>
__global__ void debug_mma_kernel(const T* a, const T* b, float* d, int K) {
  const int lane = threadIdx.x & 63;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mma_tile<T>(a, K, b, K, K, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) d[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
  // F[i][j] = sum_k b[i][k] * E[k][j],  k in 0..15
  f32x4 f = {0.f, 0.f, 0.f, 0.f};
  f = wa::mma_lds_chain<T>(b, K, 0, acc, f);
#pragma unroll
  for (int r = 0; r < 4; ++r) d[256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = f[r];
  // G[i][j] = sum_k E[k][i] * E[k][j]  (both chained)
  f32x4 gq = {0.f, 0.f, 0.f, 0.f};
  gq = wa::mma_chain<T>(acc, acc, gq);
#pragma unroll
  for (int r = 0; r < 4; ++r) d[512 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = gq[r];
}
}  // namespace

extern "C" int rssf_debug_mma(const void* a, const void* b, float* d, int K, int dtype, void* stream) {
  RSSF_REQUIRE(a && b && d && K >= 16 && K % 16 == 0, "debug_mma: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) debug_mma_kernel<float><<<1, 64, 0, st>>>((const float*)a, (const float*)b, d, K);
  else if (dtype == RSSF_BF16) debug_mma_kernel<bf16_t><<<1, 64, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, d, K);
  else { set_error("debug_mma: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("debug_mma");
}
