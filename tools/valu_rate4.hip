// Issue cost of packed-fp32 ops on gfx950 as a function of how many distinct 64-bit VGPR operands they read
// (eight independent accumulator chains per wave, 8 waves per SIMD: throughput, not latency).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 2048
#define ACC8(BODY) asm volatile(BODY(0) BODY(1) BODY(2) BODY(3) BODY(4) BODY(5) BODY(6) BODY(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(n))
#define PKFMA3(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"          /* three distinct VGPR pairs */
#define PKFMA2(i) "v_pk_fma_f32 %" #i ", %8, %8, %" #i "\n"          /* two distinct */
#define PKFMA1(i) "v_pk_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n"  /* one */
#define PKMUL2(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define PKADD2(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define PKFMASEL(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " op_sel_hi:[1,0,1]\n"
template <int MODE> __global__ void k(float* out, float seed) {
  v2f a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
  v2f m = {1.0001f, 0.9999f}, n = {1e-3f, 2e-3f};
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 0) ACC8(PKFMA3);
    if (MODE == 1) ACC8(PKFMA2);
    if (MODE == 2) ACC8(PKFMA1);
    if (MODE == 3) ACC8(PKMUL2);
    if (MODE == 4) ACC8(PKADD2);
    if (MODE == 5) ACC8(PKFMASEL);
  }
  v2f s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <int MODE> void run(const char* name) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256 * 8, 256>>>(out, 1.0f); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<256 * 8, 256>>>(out, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per = ms * 1e6 / (8.0 * ITERS * 8);
  printf("%-44s %7.3f ms  %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, ms, per * 2.4);
  hipFree(out);
}
int main() {
  run<0>("v_pk_fma_f32, 3 distinct VGPR pairs"); run<5>("v_pk_fma_f32, 3 pairs, op_sel broadcast"); run<1>("v_pk_fma_f32, 2 distinct"); run<2>("v_pk_fma_f32, 1 distinct");
  run<3>("v_pk_mul_f32"); run<4>("v_pk_add_f32");
  return 0;
}
