// Development micro-benchmark: cycles of factor32 (chol_factor32.h) on one workgroup, per 4-column panel,
// with the timing-only experiments of F32_EXP.
//   for e in 0 1 2 3 4 7; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DF32_EXP=$e scripts/ubench_f32.hip -o scripts/ubench_f32_$e; done
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../psba_amd/csrc/chol_factor32.h"
using namespace psba;

__global__ __launch_bounds__(256) void k(const double *A, double *out, long long *tim, int reps) {
  __shared__ Factor32Lds s;
  const int tid = threadIdx.x;
  for (int r = 0; r < reps; r++) {
    if (tid < 4) s.flag[tid] = 0;
    if (tid == 4) s.fail = 0;
    for (int t = tid; t < GB * GB; t += 256) s.D[t / GB][t % GB] = A[t];
    __syncthreads();
    if (tid == 0) tim[0] = (long long)__builtin_amdgcn_s_memtime();
    factor32<true>(s, tid, tim);
    if (tid == 0) tim[2] = (long long)__builtin_amdgcn_s_memtime();
    __syncthreads();
  }
  for (int t = tid; t < GB * GB; t += 256) out[t] = f32_L(s, t / GB, t % GB);
  for (int t = tid; t < GB * GB; t += 256) out[GB * GB + t] = f32_Linv(s, t / GB, t % GB);
}

int main() {
  std::vector<double> B(GB * GB), A(GB * GB, 0.0), out(2 * GB * GB);
  srand(3);
  for (auto &v : B) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < GB; i++)
    for (int j = 0; j < GB; j++) {
      double t = i == j ? 4.0 : 0.0;
      for (int k = 0; k < GB; k++) t += B[i * GB + k] * B[j * GB + k];
      A[i * GB + j] = t;
    }
  double *dA, *dO;
  long long *dT, hT[32];
  hipMalloc(&dA, 8 * GB * GB); hipMalloc(&dO, 16 * GB * GB); hipMalloc(&dT, 32 * 8);
  hipMemcpy(dA, A.data(), 8 * GB * GB, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(dT, 0, 256);
    k<<<1, 256>>>(dA, dO, dT, 20);
    hipDeviceSynchronize();
    hipMemcpy(hT, dT, 256, hipMemcpyDeviceToHost);
    printf("F32_EXP=%d  factor32 %lld memtime ticks (~ core cycles); pivot wave done %lld, inverse waves done %lld; per panel:",
           F32_EXP, hT[2] - hT[0], hT[3] - hT[0], hT[4] - hT[0]);
    for (int q = 0; q < 8; q++) printf(" %lld", hT[5 + q] - (q ? hT[4 + q] : hT[0]));
    printf("\n   inverse wave, per panel (saw the panel at, rows stored at; cycles since entry):");
    for (int q = 0; q < 8; q++) printf(" %lld/%lld", hT[16 + 2 * q] - hT[0], hT[17 + 2 * q] - hT[0]);
    printf("\n   pivot wave published at:");
    for (int q = 0; q < 8; q++) printf(" %lld", hT[5 + q] - hT[0]);
    printf("\n");
  }
  hipMemcpy(out.data(), dO, 16 * GB * GB, hipMemcpyDeviceToHost);
  // check L L^T = A (only meaningful for F32_EXP = 0)
  double err = 0;
  for (int i = 0; i < GB; i++)
    for (int j = 0; j <= i; j++) {
      double t = 0;
      for (int k = 0; k <= j; k++) t += out[i * GB + k] * out[j * GB + k];
      err = fmax(err, fabs(t - A[i * GB + j]));
    }
  double err2 = 0;  // L Linv = I
  for (int i = 0; i < GB; i++)
    for (int j = 0; j < GB; j++) {
      double t = 0;
      for (int k = 0; k < GB; k++) t += out[i * GB + k] * out[GB * GB + k * GB + j];
      err2 = fmax(err2, fabs(t - (i == j)));
    }
  printf("max |L L^T - A| = %.3e   max |L Linv - I| = %.3e\n", err, err2);
  return 0;
}
