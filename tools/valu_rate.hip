// Microbenchmark: issue rate of scalar vs packed fp32 VALU ops on gfx950 (wave64), per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE> __global__ void k(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float m = 1.0001f, c = 0.5f;
  const v2f m2 = {m, m}, c2 = {c, c};
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 0) {  // 8 independent scalar fma
      a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
      a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
    } else if (MODE == 1) {  // 8 independent packed fma (16 flops-pairs)
      p0 = __builtin_elementwise_fma(p0, m2, c2); p1 = __builtin_elementwise_fma(p1, m2, c2); p2 = __builtin_elementwise_fma(p2, m2, c2); p3 = __builtin_elementwise_fma(p3, m2, c2);
      p4 = __builtin_elementwise_fma(p4, m2, c2); p5 = __builtin_elementwise_fma(p5, m2, c2); p6 = __builtin_elementwise_fma(p6, m2, c2); p7 = __builtin_elementwise_fma(p7, m2, c2);
    } else if (MODE == 2) {  // 8 scalar mul + 8 scalar add (no contraction)
      a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
    } else if (MODE == 3) {  // 8 v_rcp
      a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
      a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
    } else if (MODE == 4) {  // 8 IEEE divisions
      a0 = c / a0; a1 = c / a1; a2 = c / a2; a3 = c / a3; a4 = c / a4; a5 = c / a5; a6 = c / a6; a7 = c / a7;
    } else if (MODE == 5) {  // 8 v_max_f32
      a0 = fmaxf(a0, c); a1 = fmaxf(a1, m); a2 = fmaxf(a2, c); a3 = fmaxf(a3, m); a4 = fmaxf(a4, c); a5 = fmaxf(a5, m); a6 = fmaxf(a6, c); a7 = fmaxf(a7, m);
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 6) {  // 8 f64 fma
      double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
      for (int k = 0; k < 2; k++) { d0 = __builtin_fma(d0, 1.0001, 0.5); d1 = __builtin_fma(d1, 1.0001, 0.5); d2 = __builtin_fma(d2, 1.0001, 0.5); d3 = __builtin_fma(d3, 1.0001, 0.5); }
      a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
    }
  }
  float r = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  r += ((p0.x + p0.y) + (p1.x + p1.y)) + ((p2.x + p2.y) + (p3.x + p3.y)) + ((p4.x + p4.y) + (p5.x + p5.y)) + ((p6.x + p6.y) + (p7.x + p7.y));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, int waves_per_simd, double ops_per_iter) {
  float* out; hipMalloc(&out, 256 * 1024 * 64 * sizeof(float));
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 per SIMD of a CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: waves_per_simd waves, each ITERS * ops_per_iter wave-instructions
  double wave_instr_per_simd = (double)waves_per_simd * ITERS * ops_per_iter;
  printf("%-34s waves/SIMD %d  %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32 x8", w, 8);
    run<1>("v_pk_fma_f32 x8", w, 8);
    run<2>("v_mul_f32 x8 + v_add_f32 x8", w, 16);
    run<3>("v_rcp_f32 x8", w, 8);
    run<4>("IEEE div x8 (per division)", w, 8);
    run<5>("v_max_f32 x8", w, 8);
    run<6>("v_fma_f64 x8 (+4 cvt pairs)", w, 8);
  }
  return 0;
}
