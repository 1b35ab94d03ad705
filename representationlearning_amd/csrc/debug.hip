// Test hooks: exercise the MFMA tile helpers in isolation so layout mistakes show up as a unit-test failure
// rather than inside a fused kernel.
#include "win_attn.hip.h"
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

namespace {
// probe of ds_read_b64_tr_b16: LDS holds lds[i] = i; lane l passes element address addr[l]; out[l*4+j] = result elem j
__global__ void debug_trread_kernel(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  typedef __attribute__((ext_vector_type(4))) short v4s;
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + addr[lane]));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
}  // namespace

extern "C" int rssf_debug_trread(const int* addr, short* out, void* stream) {
  RSSF_REQUIRE(addr && out, "debug_trread: bad arguments");
  debug_trread_kernel<<<1, 64, 0, (hipStream_t)stream>>>(addr, out);
  return check_launch("debug_trread");
}

namespace {
// probe of the LDS-free lane reductions of common.hip.h: out[0..5][lane] = xor16 sum, xor32 sum, 4-row sum, 4-row max, wave sum, wave max
__global__ void debug_lane_reduce_kernel(const float* in, float* out) {
  const int lane = threadIdx.x;
  const float v = in[lane];
  out[0 * 64 + lane] = xor16_reduce<OpSum>(v);
  out[1 * 64 + lane] = xor32_reduce<OpSum>(v);
  out[2 * 64 + lane] = rows_reduce<OpSum>(v);
  out[3 * 64 + lane] = rows_reduce<OpMax>(v);
  out[4 * 64 + lane] = wave_reduce_dpp<OpSum>(v);
  out[5 * 64 + lane] = wave_reduce_dpp<OpMax>(v);
  // raw semantics probe: a = 1000 + lane, b = 2000 + lane
  const auto r16 = __builtin_amdgcn_permlane16_swap(1000u + lane, 2000u + lane, false, false);
  const auto r32 = __builtin_amdgcn_permlane32_swap(1000u + lane, 2000u + lane, false, false);
  out[6 * 64 + lane] = (float)r16[0]; out[7 * 64 + lane] = (float)r16[1];
  out[8 * 64 + lane] = (float)r32[0]; out[9 * 64 + lane] = (float)r32[1];
}
}  // namespace

extern "C" int rssf_debug_lane_reduce(const float* in, float* out, void* stream) {
  RSSF_REQUIRE(in && out, "debug_lane_reduce: bad arguments");
  debug_lane_reduce_kernel<<<1, 64, 0, (hipStream_t)stream>>>(in, out);
  return check_launch("debug_lane_reduce");
}

extern "C" int rssf_debug_mma(const void* a, const void* b, float* d, int K, int dtype, void* stream) {
  RSSF_REQUIRE(a && b && d && K >= 16 && K % 16 == 0, "debug_mma: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) debug_mma_kernel<float><<<1, 64, 0, st>>>((const float*)a, (const float*)b, d, K);
  else if (dtype == RSSF_BF16) debug_mma_kernel<bf16_t><<<1, 64, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, d, K);
  else { set_error("debug_mma: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("debug_mma");
}

// Fill the LDS of every CU with a bit pattern (one 160 KB workgroup per CU at a time, four rounds): what a kernel that reads LDS it never
// wrote would see next.  tests/test_gpu_trainer.py runs a training step before and after poisoning with NaNs: a result that depends on
// uninitialised LDS (a pad column that enters an MFMA, a fold buffer read past what was written) turns up as a different loss or gradient.
__global__ void __launch_bounds__(256) debug_poison_lds_kernel(unsigned pattern, unsigned* sink) {
  extern __shared__ unsigned pl[];
  const int n = 160 * 1024 / 4;
  for (int i = threadIdx.x; i < n; i += 256) pl[i] = pattern;
  __syncthreads();
  if (sink && pl[(threadIdx.x * 37) % n] != pattern) sink[0] = 1u;       // (keeps the stores alive)
}
extern "C" int rssf_debug_poison_lds(unsigned pattern, void* scratch4, void* stream) {
  static hipError_t e = hipFuncSetAttribute((const void*)debug_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { set_error("debug_poison_lds: cannot raise the LDS limit: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  debug_poison_lds_kernel<<<dim3((unsigned)cus * 4), 256, 160 * 1024, (hipStream_t)stream>>>(pattern, (unsigned*)scratch4);
  return check_launch("debug_poison_lds");
}

