// Stand-alone probe for DESIGN.md lesson 23 (no PyTorch): the exact instruction pair the SLP vectoriser produced in
// gate_weights_bwd2_kernel - two ds_read2_b32 feeding v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (crossed halves) - executed by every
// wave of a 256-CU grid while a second stream keeps LDS, VALU and the memory pipes of the same CUs busy.  Counts wrong results.
//   hipcc --offload-arch=gfx950 -O3 tools/pkmul_repro.hip -o /tmp/pkmul_repro && /tmp/pkmul_repro [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) probe(unsigned* bad, unsigned* first, int iters, int waitn) {
  __shared__ float lds[2 * 512];
  const int t = threadIdx.x;
  unsigned nbad = 0;
  for (int i = 0; i < iters; ++i) {
    lds[2 * t] = 1.0f + (float)((t * 7 + i) & 255);            // non-zero operands: a zero half is always an error
    lds[2 * t + 1] = 2.0f + (float)((t * 3 + i) & 127);
    lds[512 + 2 * t] = 3.0f + (float)((t + 5 * i) & 63);
    lds[512 + 2 * t + 1] = 0.5f + (float)((t * 11 + i) & 31);
    __syncthreads();
    const unsigned a0 = (unsigned)(2 * ((t + i) & 255)) * 4u, a1 = 2048u + (unsigned)(2 * ((t + 3 * i) & 255)) * 4u;
    f2 a, b, r;
    asm volatile("ds_read2_b32 %1, %3 offset1:1\n ds_read2_b32 %2, %4 offset1:1\n s_waitcnt lgkmcnt(0)\n"
                 "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r), "=&v"(a), "=&v"(b) : "v"(a0), "v"(a1) : "memory");
    float e0 = a.x * b.y, e1 = a.y * b.x, g0 = r.x, g1 = r.y;
    if (waitn) {          // the consumer of the original kernel: the lane quad folded by two DPP adds, straight behind the product
      g0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, g0), 0xB1, 0xF, 0xF, true));
      g0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, g0), 0x4E, 0xF, 0xF, true));
      e0 += __shfl_xor(e0, 1, 64); e0 += __shfl_xor(e0, 2, 64);
    }
    if (g0 != e0 || g1 != e1) { if (!nbad) atomicCAS(first, 0u, 1u + (unsigned)(blockIdx.x * 256 + t)); ++nbad; }
    __syncthreads();
  }
  if (nbad) atomicAdd(bad, nbad);
}

__global__ void __launch_bounds__(256) noise(float* sink, int iters) {      // LDS + VALU + HBM traffic from another stream
  __shared__ float s[8192];
  float acc = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int k = threadIdx.x; k < 8192; k += 256) s[k] = acc + k;
    __syncthreads();
    for (int k = 0; k < 32; ++k) acc = fmaf(acc, 1.0001f, s[(threadIdx.x * 33 + k * 257 + i) & 8191]);
    __syncthreads();
    acc += sink[(blockIdx.x * 256 + threadIdx.x + i * 4099) & ((1 << 22) - 1)];
  }
  sink[(blockIdx.x * 256 + threadIdx.x) & ((1 << 22) - 1)] = acc;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 400;
  unsigned *bad, *first; float* sink;
  hipMalloc(&bad, 8); first = bad + 1; hipMalloc(&sink, sizeof(float) << 22);
  hipMemset(bad, 0, 8); hipMemset(sink, 0, sizeof(float) << 22);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  for (int mode = 0; mode < 2; ++mode) {
    for (int l = 0; l < launches; ++l) {
      noise<<<512, 256, 0, s2>>>(sink, 40);
      probe<<<1024, 256, 0, s1>>>(bad, first, 64, mode);
      if (l % 3 == 0) noise<<<256, 256, 0, s2>>>(sink, 15);
    }
    hipDeviceSynchronize();
    unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %d launches x 1024 blocks x 256 lanes x 64 products under load: %u wrong (first thread %d)  [%s]\n", mode, launches, h[0],
           (int)h[1] - 1, hipGetErrorString(hipGetLastError()));
    hipMemset(bad, 0, 8);
  }
  return 0;
}
