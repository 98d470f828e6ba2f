// Per-instruction issue cost on gfx950 (wave64), measured with inline asm so the compiler cannot
// substitute sequences.  8 independent chains, 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
#define OP8(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m))
#define OP8U(INS) asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define OP8T(INS) asm volatile(INS " %0, %0, %8, %8\n" INS " %1, %1, %8, %8\n" INS " %2, %2, %8, %8\n" INS " %3, %3, %8, %8\n" INS " %4, %4, %8, %8\n" INS " %5, %5, %8, %8\n" INS " %6, %6, %8, %8\n" INS " %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m))
template <int MODE> __global__ void k(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float m = 1.0001f;
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 0) OP8("v_mul_f32");
    if (MODE == 1) OP8("v_add_f32");
    if (MODE == 2) OP8("v_max_f32");
    if (MODE == 3) OP8("v_and_b32");
    if (MODE == 4) OP8U("v_cvt_f32_ubyte0");
    if (MODE == 5) OP8U("v_cvt_f32_u32");
    if (MODE == 6) OP8T("v_fma_f32");
    if (MODE == 7) OP8T("v_med3_f32");
    if (MODE == 8) OP8U("v_exp_f32");
    if (MODE == 9) OP8U("v_rsq_f32");
    if (MODE == 10) OP8U("v_floor_f32");
    if (MODE == 11) OP8("v_add_u32");
    if (MODE == 12) OP8U("v_cvt_f32_f16");
    if (MODE == 13) OP8("v_cndmask_b32");
    if (MODE == 14) OP8U("v_mov_b32");
    if (MODE == 15) OP8U("v_sqrt_f32");
    if (MODE == 16) OP8("v_mul_lo_u32");
    if (MODE == 17) OP8("v_min_i32");
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
  printf("%-20s %7.3f ms  %.2f ns/wave-instr/SIMD  (%.2f cycles @2.4GHz)\n", name, ms, per, per * 2.4);
  hipFree(out);
}
int main() {
  run<0>("v_mul_f32"); run<1>("v_add_f32"); run<6>("v_fma_f32"); run<2>("v_max_f32"); run<7>("v_med3_f32"); run<3>("v_and_b32");
  run<11>("v_add_u32"); run<17>("v_min_i32"); run<13>("v_cndmask_b32"); run<14>("v_mov_b32"); run<16>("v_mul_lo_u32");
  run<4>("v_cvt_f32_ubyte0"); run<5>("v_cvt_f32_u32"); run<12>("v_cvt_f32_f16"); run<10>("v_floor_f32");
  run<8>("v_exp_f32"); run<9>("v_rsq_f32"); run<15>("v_sqrt_f32");
  return 0;
}
