// micro-benchmark: FP64 VALU issue rates on gfx950 (used to price the WENO kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void k(double* out, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c); }
    if (OP == 1) { a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m; }
    if (OP == 2) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
    if (OP == 3) { a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3); a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7); }
    if (OP == 4) { a0 = a0 > a1 ? a2 : a3; a1 = a1 > a2 ? a3 : a4; a2 = a2 > a3 ? a4 : a5; a3 = a3 > a4 ? a5 : a6; a4 = a4 > a5 ? a6 : a7; a5 = a5 > a6 ? a7 : a0; a6 = a6 > a7 ? a0 : a1; a7 = a7 > a0 ? a1 : a2; }
    if (OP == 5) { float f0 = (float)a0; f0 = fmaf(f0, 1.0001f, 0.5f); a0 = f0; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd, threads = 256, iters = 20000;   // 256 threads = 4 waves = 1 per SIMD
  double* d; hipMalloc(&d, sizeof(double) * blocks * threads);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<blocks, threads>>>(d, 100, 1.0); hipDeviceSynchronize();
  hipEventRecord(a); k<OP><<<blocks, threads>>>(d, iters, 1.0); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double wave_instr = (double)blocks * 4 * iters * 8;          // per-wave instructions of the measured op
  double per_simd = wave_instr / 1024.0;                       // per SIMD
  printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / per_simd);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("fma_f64", w); run<1>("mul_f64", w); run<2>("add_f64", w); run<3>("rcp_f64", w); run<4>("cmp+cndmask", w); }
  return 0;
}
