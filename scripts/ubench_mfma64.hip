// Microbenchmark: what v_mfma_f64_16x16x4_f64 sustains on an MI355X, by independent accumulator
// chains per wave (1..16) and waves per SIMD (1, 2).  Build: hipcc --offload-arch=gfx950 -O3
// scripts/ubench_mfma64.hip -o scripts/ubench_mfma64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

template <int NACC>
void run(int wgs_per_cu, double *out) {
  const int iters = 2000 / NACC * 4;
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 2.0);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 2.0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mf = 5.0 * grid * 4.0 * iters * 8 * NACC;
  printf("chains %2d  waves/SIMD %d : %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC, wgs_per_cu,
         mf * 2048 / (ms * 1e-3) / 1e12, (ms * 1e-3 / 5) * 2.4e9 / (iters * 8.0 * NACC * wgs_per_cu));
}

int main() {
  double *out;
  hipMalloc(&out, 64);
  for (int w = 1; w <= 2; w++) {
    run<1>(w, out);
    run<2>(w, out);
    run<4>(w, out);
    run<8>(w, out);
    run<16>(w, out);
  }
  return 0;
}
