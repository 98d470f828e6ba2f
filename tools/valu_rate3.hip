// Issue cost of the packed-f16 / mixed-precision VALU ops on gfx950 (wave64); same harness as valu_rate2.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
#define OP8(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m))
#define OP8T(INS) asm volatile(INS " %0, %0, %8, %8\n" INS " %1, %1, %8, %8\n" INS " %2, %2, %8, %8\n" INS " %3, %3, %8, %8\n" INS " %4, %4, %8, %8\n" INS " %5, %5, %8, %8\n" INS " %6, %6, %8, %8\n" INS " %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m))
#define OP8ACC(INS, SFX) asm volatile(INS " %0, %8, %8, %0" SFX "\n" INS " %1, %8, %8, %1" SFX "\n" INS " %2, %8, %8, %2" SFX "\n" INS " %3, %8, %8, %3" SFX "\n" INS " %4, %8, %8, %4" SFX "\n" INS " %5, %8, %8, %5" SFX "\n" INS " %6, %8, %8, %6" SFX "\n" INS " %7, %8, %8, %7" SFX : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m))
template <int MODE> __global__ void k(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float m = 1.0001f;
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 0) OP8T("v_fma_f32");
    if (MODE == 1) OP8T("v_pk_fma_f16");
    if (MODE == 2) OP8("v_pk_mul_f16");
    if (MODE == 3) OP8("v_pk_max_f16");
    if (MODE == 4) OP8("v_pk_min_f16");
    if (MODE == 5) OP8("v_pk_add_f16");
    if (MODE == 6) OP8("v_cvt_pkrtz_f16_f32");
    if (MODE == 7) OP8ACC("v_fma_mix_f32", " op_sel_hi:[1,1,0]");
    if (MODE == 8) OP8ACC("v_dot2_f32_f16", "");
    if (MODE == 11) OP8ACC("v_fma_mix_f32", " op_sel:[1,0,0] op_sel_hi:[1,1,0]");
    if (MODE == 12) OP8ACC("v_dot4_i32_i8", "");
    if (MODE == 13) OP8ACC("v_dot4_u32_u8", "");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
template <int MODE> void run(const char* name) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256 * 8, 256>>>(out, 1.0f); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<256 * 8, 256>>>(out, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per = ms * 1e6 / (8.0 * ITERS * 8);
  printf("%-28s %7.3f ms  %.2f ns/wave-instr/SIMD  (%.2f cycles @2.4GHz)\n", name, ms, per, per * 2.4);
  hipFree(out);
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f16"); run<2>("v_pk_mul_f16"); run<5>("v_pk_add_f16");
  run<3>("v_pk_max_f16"); run<4>("v_pk_min_f16"); run<6>("v_cvt_pkrtz_f16_f32"); run<7>("v_fma_mix_f32 (lo)"); run<11>("v_fma_mix_f32 (hi)");
  run<8>("v_dot2_f32_f16"); run<12>("v_dot4_i32_i8"); run<13>("v_dot4_u32_u8");
  return 0;
}
