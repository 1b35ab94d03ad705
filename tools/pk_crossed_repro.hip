// Stand-alone reproducer for DESIGN.md lesson 59 (no PyTorch, no library): v_pk_add_f32 with the SECOND source crossed
// (op_sel:[0,1] op_sel_hi:[1,0]: low result += src1.hi, high result += src1.lo) accumulating in place, with MFMAs in flight.
// Exact integer-valued operands; every thread checks its packed sums against scalar adds of the same values.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_crossed_repro.hip -o tools/bin/pk_crossed_repro && tools/bin/pk_crossed_repro [launches] [blocks]
// A row of the table = one combination of: the crossed form (second source / first source / none: natural selects on swapped data),
// an MFMA in the loop (same wave / only in the odd waves of a block, which do no packed adds / none), wait states (s_nop) between the
// MFMA and the packed add, a natural packed add beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f4;
enum { SRC1 = 0, SRC0 = 1, SWAPPED = 2 };                 // form of the checked add
enum { MF_NONE = 0, MF_SAME = 1, MF_OTHER_WAVES = 2 };    // where the MFMAs run

template <int FORM, int MF, int NOPS, bool NAT, int PAD = 0>
__global__ void __launch_bounds__(256) probe(unsigned* bad, unsigned* first, int iters, float* sink) {
  const int t = threadIdx.x, wave = t >> 6;
  f2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
  float rA0 = 0.f, rA1 = 0.f, rB0 = 0.f, rB1 = 0.f;
  f4 macc = {0.f, 0.f, 0.f, 0.f};
  bf16x8_t ma, mb;
  for (int i = 0; i < 8; ++i) { ma[i] = (__bf16)(0.001f * (t + i)); mb[i] = (__bf16)(0.002f * (t ^ i)); }
  const bool adds = MF != MF_OTHER_WAVES || (wave & 1) == 0, mfmas = MF == MF_SAME || (MF == MF_OTHER_WAVES && (wave & 1));
  for (int i = 0; i < iters; ++i) {
    if (mfmas) macc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, macc, 0, 0, 0);
    if (NOPS > 0) { for (int k = 0; k < NOPS; ++k) asm volatile("s_nop 15"); }
    if (adds) {
      f2 x = {(float)((t * 3 + i * 5) & 127), (float)((t * 7 + i) & 127)};
      if (FORM == SRC1) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(accA) : "v"(x));
      else if (FORM == SRC0) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(accA) : "v"(x));
      else { f2 y = {x.y, x.x}; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(accA) : "v"(y)); }
      rA0 += x.y; rA1 += x.x;
      if (NAT) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(accB) : "v"(x)); rB0 += x.x; rB1 += x.y; }
      for (int k = 0; k < PAD; ++k) asm volatile("v_nop");       // VALU issue slots between the packed add and the loop's next MFMA
    }
  }
  unsigned nbad = 0;
  if (accA.x != rA0) nbad |= 1;            // the LOW result (read the source's high register in the crossed forms)
  if (accA.y != rA1) nbad |= 2;
  if (NAT && (accB.x != rB0 || accB.y != rB1)) nbad |= 4;
  if (nbad) { atomicCAS(first, 0u, (nbad << 28) | (1u + (unsigned)((blockIdx.x * 256 + t) & 0xfffffff))); atomicAdd(bad, 1u); }
  if (MF != MF_NONE) sink[(blockIdx.x * 256 + t) & 0xfffff] = macc[0] + macc[3];
}

// Other packed instructions / selects beside MFMAs of the same wave: tmp = op(a, x[, c]) with the selects under test, both halves summed by
// scalar adds and held against the scalar arithmetic of what the selects mean.
enum { ADD_BHI = 0, ADD_BLO = 1, MUL_X1 = 2, MUL_BHI = 3, MUL_BHI0 = 4, FMA_X1 = 5, FMA_X2 = 6, FMA_BLO2 = 7, MUL_X0 = 8 };
template <int OP>
__global__ void __launch_bounds__(256) probe2(unsigned* bad, unsigned* first, int iters, float* sink) {
  const int t = threadIdx.x;
  float s0 = 0.f, s1 = 0.f, r0 = 0.f, r1 = 0.f;
  f4 macc = {0.f, 0.f, 0.f, 0.f};
  bf16x8_t ma, mb;
  for (int i = 0; i < 8; ++i) { ma[i] = (__bf16)(0.001f * (t + i)); mb[i] = (__bf16)(0.002f * (t ^ i)); }
  for (int i = 0; i < iters; ++i) {
    macc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, macc, 0, 0, 0);
    const f2 a = {(float)((t + i) & 15), (float)((t * 5 + i * 3) & 15)}, x = {(float)((t * 3 + i * 5) & 31), (float)((t * 7 + i) & 31)}, c = {(float)((t + 2 * i) & 63), (float)((3 * t + i) & 63)};
    f2 y; float e0, e1;
    if (OP == ADD_BHI)       { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(y) : "v"(a), "v"(x)); e0 = a.x + x.y; e1 = a.y + x.y; }
    else if (OP == ADD_BLO)  { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(y) : "v"(a), "v"(x)); e0 = a.x + x.x; e1 = a.y + x.x; }
    else if (OP == MUL_X1)   { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(a), "v"(x)); e0 = a.x * x.y; e1 = a.y * x.x; }
    else if (OP == MUL_X0)   { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(y) : "v"(a), "v"(x)); e0 = a.y * x.x; e1 = a.x * x.y; }
    else if (OP == MUL_BHI)  { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(y) : "v"(a), "v"(x)); e0 = a.x * x.y; e1 = a.y * x.y; }
    else if (OP == MUL_BHI0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(y) : "v"(a), "v"(x)); e0 = a.y * x.x; e1 = a.y * x.y; }
    else if (OP == FMA_X1)   { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(y) : "v"(a), "v"(x), "v"(c)); e0 = a.x * x.y + c.x; e1 = a.y * x.x + c.y; }
    else if (OP == FMA_X2)   { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(y) : "v"(a), "v"(x), "v"(c)); e0 = a.x * x.x + c.y; e1 = a.y * x.y + c.x; }
    else                     { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(y) : "v"(a), "v"(x), "v"(c)); e0 = a.x * x.x + c.x; e1 = a.y * x.y + c.x; }
    s0 += y.x; s1 += y.y; r0 += e0; r1 += e1;
  }
  unsigned nbad = (s0 != r0 ? 1u : 0u) | (s1 != r1 ? 2u : 0u);
  if (nbad) { atomicCAS(first, 0u, (nbad << 28) | (1u + (unsigned)((blockIdx.x * 256 + t) & 0xfffffff))); atomicAdd(bad, 1u); }
  sink[(blockIdx.x * 256 + t) & 0xfffff] = macc[0] + macc[3];
}

static unsigned* bad; static float* sink; static int launches, blocks;
template <int OP> void row2(const char* what) {
  hipMemset(bad, 0, 8);
  for (int l = 0; l < launches; ++l) probe2<OP><<<blocks, 256>>>(bad, bad + 1, 2048, sink);
  (void)hipDeviceSynchronize();
  unsigned h[2]; (void)hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
  printf("%-98s %8u threads wrong of %.2e", what, h[0], (double)launches * blocks * 256);
  if (h[0]) printf("   (first: which halves = %u: 1 low, 2 high)", h[1] >> 28);
  printf("  [%s]\n", hipGetErrorString(hipGetLastError()));
}
template <int FORM, int MF, int NOPS, bool NAT, int PAD = 0> void row(const char* what) {
  hipMemset(bad, 0, 8);
  for (int l = 0; l < launches; ++l) probe<FORM, MF, NOPS, NAT, PAD><<<blocks, 256>>>(bad, bad + 1, 2048, sink);
  (void)hipDeviceSynchronize();
  unsigned h[2]; (void)hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
  printf("%-98s %8u threads wrong of %.2e", what, h[0], (double)launches * blocks * 256);
  if (h[0]) printf("   (first: which sums = %u: 1 low result, 2 high result, 4 the natural add)", h[1] >> 28);
  printf("  [%s]\n", hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  launches = argc > 1 ? atoi(argv[1]) : 100; blocks = argc > 2 ? atoi(argv[2]) : 2048;
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&sink, sizeof(float) << 20);
  printf("%d launches x %d blocks x 256 threads, 2048 packed adds per thread\n", launches, blocks);
  row<SRC1, MF_NONE, 0, false>("second source crossed, no MFMA");
  row<SRC1, MF_NONE, 0, true>("second source crossed + a natural packed add, no MFMA");
  row<SRC1, MF_SAME, 0, false>("second source crossed, an MFMA per add in the SAME wave");
  row<SRC1, MF_SAME, 0, true>("second source crossed + a natural packed add, an MFMA per add in the same wave");
  row<SRC1, MF_SAME, 1, false>("second source crossed, MFMA in the same wave, s_nop 15 (16 wait states) between them");
  row<SRC1, MF_SAME, 4, false>("second source crossed, MFMA in the same wave, 4 x s_nop 15 (64 wait states) between them");
  row<SRC1, MF_SAME, 0, false, 1>("second source crossed, MFMA in the same wave, 1 v_nop between the add and the next MFMA");
  row<SRC1, MF_SAME, 0, false, 2>("second source crossed, MFMA in the same wave, 2 v_nop between the add and the next MFMA");
  row<SRC1, MF_SAME, 0, false, 4>("second source crossed, MFMA in the same wave, 4 v_nop between the add and the next MFMA");
  row<SRC1, MF_SAME, 0, false, 8>("second source crossed, MFMA in the same wave, 8 v_nop between the add and the next MFMA");
  row<SRC1, MF_SAME, 0, false, 16>("second source crossed, MFMA in the same wave, 16 v_nop between the add and the next MFMA");
  row<SRC1, MF_OTHER_WAVES, 0, false>("second source crossed in the even waves, MFMAs only in the ODD waves of the block");
  row<SRC1, MF_OTHER_WAVES, 0, false, 8>("second source crossed in the even waves + 8 v_nop, MFMAs only in the odd waves");
  row<SRC0, MF_SAME, 0, false>("FIRST source crossed, an MFMA per add in the same wave");
  row<SRC0, MF_OTHER_WAVES, 0, false>("first source crossed in the even waves, MFMAs only in the odd waves");
  row<SWAPPED, MF_SAME, 0, false>("natural selects on swapped data, an MFMA per add in the same wave");
  row<SWAPPED, MF_OTHER_WAVES, 0, false>("natural selects on swapped data in the even waves, MFMAs only in the odd waves");
  printf("other instructions and selects, an MFMA per packed instruction in the same wave:\n");
  row2<MUL_X1>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]     (second source crossed: lesson 23's instruction)");
  row2<MUL_X0>("v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]     (first source crossed)");
  row2<FMA_X1>("v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (second source crossed)");
  row2<FMA_X2>("v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0] (third source crossed)");
  row2<ADD_BHI>("v_pk_add_f32 op_sel:[0,1]                     (second source: high half to both results)");
  row2<ADD_BLO>("v_pk_add_f32 op_sel_hi:[1,0]                  (second source: low half to both results)");
  row2<MUL_BHI>("v_pk_mul_f32 op_sel:[0,1]                     (second source: high half to both results)");
  row2<MUL_BHI0>("v_pk_mul_f32 op_sel:[1,0]                     (first source: high half to both results)");
  row2<FMA_BLO2>("v_pk_fma_f32 op_sel_hi:[1,1,0]                (third source: low half to both results)");
  return 0;
}
