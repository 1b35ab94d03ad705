// Does gfx950 hardware need more wait states between an MFMA and a dependent read of its result than hipcc inserts?
// (DESIGN.md lesson 21: the fp32 C = 48 attention backward - 512 VGPRs, ~200 spills, i.e. MFMA results stored to scratch right after
// they are produced - returns garbage that changes with every scheduling flag.)  For each MFMA form and each consumer kind
// (VALU add / global store / another MFMA's SrcC) the kernel runs MFMA -> k x s_nop -> consumer from inline asm for k = 0..18 and
// reports the smallest k from which the result is right; the compiler's own choice for the same pair is read from the disassembly:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_hazard.hip -o tools/bin/mfma_hazard ; llvm-objdump -d ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

#define NOPS(k) ".rept " #k "\n s_nop 0\n .endr\n"

// The accumulator tuple is v[20:23] by name (a 32-bit lane of an asm operand tuple cannot be named); a first MFMA leaves OLD values
// in it, the second overwrites them; the consumer must see the second.
#define LOADC "v_mov_b32 v20, %[c0]\n v_mov_b32 v21, %[c1]\n v_mov_b32 v22, %[c2]\n v_mov_b32 v23, %[c3]\n s_nop 4\n"
template <int K> __global__ void probe_f32_valu(const float* in, float* out) {
  const int t = threadIdx.x;
  float a = in[t], b = in[64 + t], c0 = in[128 + t], c1 = in[192 + t], c2 = in[256 + t], c3 = in[320 + t];
  float r;
  asm volatile(LOADC "v_mfma_f32_16x16x4_f32 v[20:23], %[a], %[b], v[20:23]\n s_nop 15\n s_nop 15\n"
               "v_mfma_f32_16x16x4_f32 v[20:23], %[a], %[b], v[20:23]\n" NOPS(%c[k]) "v_add_f32 %[r], v20, v23\n s_nop 15\n s_nop 15"
               : [r] "=v"(r) : [a] "v"(a), [b] "v"(b), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [k] "n"(K) : "v20", "v21", "v22", "v23");
  out[t] = r;
}
template <int K> __global__ void probe_f32_store(const float* in, f4* out) {
  const int t = threadIdx.x;
  float a = in[t], b = in[64 + t], c0 = in[128 + t], c1 = in[192 + t], c2 = in[256 + t], c3 = in[320 + t];
  f4* p = out + t;
  asm volatile(LOADC "v_mfma_f32_16x16x4_f32 v[20:23], %[a], %[b], v[20:23]\n s_nop 15\n s_nop 15\n"
               "v_mfma_f32_16x16x4_f32 v[20:23], %[a], %[b], v[20:23]\n" NOPS(%c[k]) "global_store_dwordx4 %[p], v[20:23], off\n s_waitcnt vmcnt(0)\n s_nop 15"
               :: [a] "v"(a), [b] "v"(b), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [p] "v"(p), [k] "n"(K) : "v20", "v21", "v22", "v23", "memory");
}
template <int K> __global__ void probe_bf16_store(const float* in, f4* out) {
  const int t = threadIdx.x;
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[(t * 8 + i) & 511]; b[i] = (__bf16)in[(t * 5 + i * 3) & 511]; }
  float c0 = in[128 + t], c1 = in[192 + t], c2 = in[256 + t], c3 = in[320 + t];
  f4* p = out + t;
  asm volatile(LOADC "v_mfma_f32_16x16x32_bf16 v[20:23], %[a], %[b], v[20:23]\n s_nop 15\n s_nop 15\n"
               "v_mfma_f32_16x16x32_bf16 v[20:23], %[a], %[b], v[20:23]\n" NOPS(%c[k]) "global_store_dwordx4 %[p], v[20:23], off\n s_waitcnt vmcnt(0)\n s_nop 15"
               :: [a] "v"(a), [b] "v"(b), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [p] "v"(p), [k] "n"(K) : "v20", "v21", "v22", "v23", "memory");
}
// what the compiler does with the same dependent pairs (read the s_nop it puts between them in the disassembly)
__global__ void cc_f32_store(const float* in, f4* out) {
  const int t = threadIdx.x;
  f4 c = {in[128 + t], in[192 + t], in[256 + t], in[320 + t]};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(in[t], in[64 + t], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(in[t], in[64 + t], c, 0, 0, 0);
  out[t] = c;
}
__global__ void cc_f32_valu(const float* in, float* out) {
  const int t = threadIdx.x;
  f4 c = {in[128 + t], in[192 + t], in[256 + t], in[320 + t]};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(in[t], in[64 + t], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(in[t], in[64 + t], c, 0, 0, 0);
  out[t] = c[0] + c[3];
}

template <int K> void run_all(const float* din, float* dout, float* ref_valu, f4* ref_store, f4* ref_bf, int* first) {
  float h[64]; f4 h4[64];
  probe_f32_valu<K><<<1, 64>>>(din, dout); hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  bool ok0 = true; for (int i = 0; i < 64; ++i) ok0 = ok0 && h[i] == ref_valu[i];
  probe_f32_store<K><<<1, 64>>>(din, (f4*)dout); hipMemcpy(h4, dout, sizeof(h4), hipMemcpyDeviceToHost);
  bool ok1 = true; for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) ok1 = ok1 && h4[i][r] == ref_store[i][r];
  probe_bf16_store<K><<<1, 64>>>(din, (f4*)dout); hipMemcpy(h4, dout, sizeof(h4), hipMemcpyDeviceToHost);
  bool ok2 = true; for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) ok2 = ok2 && h4[i][r] == ref_bf[i][r];
  printf("nops %2d: f32 MFMA -> VALU %s   f32 MFMA -> store %s   bf16 MFMA -> store %s\n", K, ok0 ? "ok " : "BAD", ok1 ? "ok " : "BAD", ok2 ? "ok " : "BAD");
  if (ok0 && first[0] < 0) first[0] = K; if (!ok0) first[0] = -1;
  if (ok1 && first[1] < 0) first[1] = K; if (!ok1) first[1] = -1;
  if (ok2 && first[2] < 0) first[2] = K; if (!ok2) first[2] = -1;
}

int main() {
  float hin[512];
  for (int i = 0; i < 512; ++i) hin[i] = 0.25f * ((i * 37) % 19) - 1.5f;
  float *din, *dout; hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, 64 * 16); hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  // references: the same sequences with plenty of wait states
  float ref_valu[64]; f4 ref_store[64], ref_bf[64];
  probe_f32_valu<18><<<1, 64>>>(din, dout); hipMemcpy(ref_valu, dout, sizeof(ref_valu), hipMemcpyDeviceToHost);
  probe_f32_store<18><<<1, 64>>>(din, (f4*)dout); hipMemcpy(ref_store, dout, sizeof(ref_store), hipMemcpyDeviceToHost);
  probe_bf16_store<18><<<1, 64>>>(din, (f4*)dout); hipMemcpy(ref_bf, dout, sizeof(ref_bf), hipMemcpyDeviceToHost);
  // cross-check the references against the compiler's own code
  f4 h4[64]; float h[64];
  cc_f32_store<<<1, 64>>>(din, (f4*)dout); hipMemcpy(h4, dout, sizeof(h4), hipMemcpyDeviceToHost);
  bool same = true; for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) same = same && h4[i][r] == ref_store[i][r];
  cc_f32_valu<<<1, 64>>>(din, dout); hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  bool same2 = true; for (int i = 0; i < 64; ++i) same2 = same2 && h[i] == ref_valu[i];
  printf("compiler-generated pairs equal the 18-nop reference: store %s, valu %s\n", same ? "yes" : "NO", same2 ? "yes" : "NO");
  int first[3] = {-1, -1, -1};
  run_all<0>(din, dout, ref_valu, ref_store, ref_bf, first); run_all<1>(din, dout, ref_valu, ref_store, ref_bf, first);
  run_all<2>(din, dout, ref_valu, ref_store, ref_bf, first); run_all<3>(din, dout, ref_valu, ref_store, ref_bf, first);
  run_all<4>(din, dout, ref_valu, ref_store, ref_bf, first); run_all<5>(din, dout, ref_valu, ref_store, ref_bf, first);
  run_all<6>(din, dout, ref_valu, ref_store, ref_bf, first); run_all<7>(din, dout, ref_valu, ref_store, ref_bf, first);
  run_all<8>(din, dout, ref_valu, ref_store, ref_bf, first); run_all<9>(din, dout, ref_valu, ref_store, ref_bf, first);
  run_all<10>(din, dout, ref_valu, ref_store, ref_bf, first); run_all<11>(din, dout, ref_valu, ref_store, ref_bf, first);
  run_all<12>(din, dout, ref_valu, ref_store, ref_bf, first); run_all<14>(din, dout, ref_valu, ref_store, ref_bf, first);
  run_all<16>(din, dout, ref_valu, ref_store, ref_bf, first);
  printf("smallest number of s_nop 0 from which every larger count tested is right: f32->VALU %d, f32->store %d, bf16->store %d\n", first[0], first[1], first[2]);
  return 0;
}
