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
enum { SRC1 = 0, SRC0 = 1, SWAPPED = 2, ADD_HI = 3, ADD_LO = 4, FMA_S1X = 5, FMA_S1HI = 6, FMA_S1LO = 7, FMA_S0X = 8, MUL_S1X = 9, NIP_ADD_HI = 10, NIP_MUL_X = 11, MUL_S0HI = 12, FMA_S2X = 13, FMA_S2HI = 14, MUL_S1HI = 15 };   // form of the checked in-place instruction
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
      else if (FORM == SWAPPED) { f2 y = {x.y, x.x}; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(accA) : "v"(y)); }
      const f2 a = {(float)((t + i) & 3), (float)((t * 5 + i * 3) & 3)};
      if (FORM <= SWAPPED) { rA0 += x.y; rA1 += x.x; }
      else if (FORM == ADD_HI) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(accA) : "v"(x)); rA0 += x.y; rA1 += x.y; }
      else if (FORM == ADD_LO) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(accA) : "v"(x)); rA0 += x.x; rA1 += x.x; }
      else if (FORM == FMA_S1X) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(accA) : "v"(a), "v"(x)); rA0 += a.x * x.y; rA1 += a.y * x.x; }
      else if (FORM == FMA_S1HI) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(accA) : "v"(a), "v"(x)); rA0 += a.x * x.y; rA1 += a.y * x.y; }
      else if (FORM == FMA_S1LO) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accA) : "v"(a), "v"(x)); rA0 += a.x * x.x; rA1 += a.y * x.x; }
      else if (FORM == FMA_S0X) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(accA) : "v"(x), "v"(a)); rA0 += x.y * a.x; rA1 += x.x * a.y; }
      else if (FORM == NIP_ADD_HI) { f2 y; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(y) : "v"(a), "v"(x)); accA.x += y.x; accA.y += y.y; rA0 += a.x + x.y; rA1 += a.y + x.y; }
      else if (FORM == NIP_MUL_X) { f2 y; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(a), "v"(x)); accA.x += y.x; accA.y += y.y; rA0 += a.x * x.y; rA1 += a.y * x.x; }
      else if (FORM == FMA_S2X) { const f2 one = {1.f, 1.f}; asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "+v"(accA) : "v"(one), "v"(x)); rA0 += x.y; rA1 += x.x; }
      else if (FORM == FMA_S2HI) { const f2 one = {1.f, 1.f}; asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1]" : "+v"(accA) : "v"(one), "v"(x)); rA0 += x.y; rA1 += x.y; }
      else if (FORM == MUL_S0HI || FORM == MUL_S1HI) {
        const f2 p = {(i & 1) ? 0.5f : 2.0f, ((i + (t & 1)) & 1) ? 2.0f : 0.5f};
        if (i == 0) { accA = f2{1.f, 1.f}; rA0 = rA1 = 1.f; }
        if (FORM == MUL_S0HI) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[1,0]" : "+v"(accA) : "v"(p));
        else asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(accA) : "v"(p));
        rA0 *= p.y; rA1 *= p.y;
      }
      else if (FORM == MUL_S1X) {            // products of powers of two that cancel over two iterations: exact, bounded
        const f2 p = {(i & 1) ? 0.5f : 2.0f, ((i + (t & 1)) & 1) ? 2.0f : 0.5f};
        if (i == 0) { accA = f2{1.f, 1.f}; rA0 = rA1 = 1.f; }
        asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(accA) : "v"(p)); rA0 *= p.y; rA1 *= p.x;
      }
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
  printf("other IN-PLACE forms, even waves; MFMAs only in the odd waves of the block:\n");
  row<ADD_HI, MF_OTHER_WAVES, 0, false>("v_pk_add_f32 acc, acc, x op_sel:[0,1]                      (second source: high half to both)");
  row<ADD_LO, MF_OTHER_WAVES, 0, false>("v_pk_add_f32 acc, acc, x op_sel_hi:[1,0]                   (second source: low half to both)");
  row<MUL_S1X, MF_OTHER_WAVES, 0, false>("v_pk_mul_f32 acc, acc, x op_sel:[0,1] op_sel_hi:[1,0]      (second source crossed)");
  row<FMA_S1X, MF_OTHER_WAVES, 0, false>("v_pk_fma_f32 acc, a, x, acc op_sel:[0,1,0] op_sel_hi:[1,0,1] (second source crossed)");
  row<FMA_S0X, MF_OTHER_WAVES, 0, false>("v_pk_fma_f32 acc, x, a, acc op_sel:[1,0,0] op_sel_hi:[0,1,1] (first source crossed)");
  row<FMA_S1HI, MF_OTHER_WAVES, 0, false>("v_pk_fma_f32 acc, a, x, acc op_sel:[0,1,0]                (second source: high half to both)");
  row<FMA_S1LO, MF_OTHER_WAVES, 0, false>("v_pk_fma_f32 acc, a, x, acc op_sel_hi:[1,0,1]             (second source: low half to both)");
  row<MUL_S1HI, MF_OTHER_WAVES, 0, false>("v_pk_mul_f32 acc, acc, x op_sel:[0,1]                      (second source: high half to both)");
  row<MUL_S0HI, MF_OTHER_WAVES, 0, false>("v_pk_mul_f32 acc, x, acc op_sel:[1,0]                      (FIRST source: high half to both)");
  row<FMA_S2X, MF_OTHER_WAVES, 0, false>("v_pk_fma_f32 acc, acc, 1, x op_sel:[0,0,1] op_sel_hi:[1,1,0] (THIRD source crossed)");
  row<FMA_S2HI, MF_OTHER_WAVES, 0, false>("v_pk_fma_f32 acc, acc, 1, x op_sel:[0,0,1]                 (third source: high half to both)");
  printf("NOT in place (fresh destination, summed by scalar adds), even waves; MFMAs only in the odd waves:\n");
  row<NIP_ADD_HI, MF_OTHER_WAVES, 0, false>("v_pk_add_f32 y, a, x op_sel:[0,1]                          (second source: high half to both)");
  row<NIP_MUL_X, MF_OTHER_WAVES, 0, false>("v_pk_mul_f32 y, a, x op_sel:[0,1] op_sel_hi:[1,0]          (second source crossed: lesson 23's instruction)");
  printf("other instructions and selects (not in place), an MFMA per packed instruction in the same wave:\n");
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
