// Development micro-benchmarks for fp64 / LDS latencies on gfx950 (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__global__ void k_fma_chain(double *out, long long *t, double x) {
  double a = x, b = 1.0000001;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) a = __builtin_fma(a, b, 1e-9);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_fma_indep(double *out, long long *t, double x) {
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4=x+4,a5=x+5,a6=x+6,a7=x+7, b = 1.0000001;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; i += 8) {
    a0 = __builtin_fma(a0, b, 1e-9); a1 = __builtin_fma(a1, b, 1e-9); a2 = __builtin_fma(a2, b, 1e-9); a3 = __builtin_fma(a3, b, 1e-9);
    a4 = __builtin_fma(a4, b, 1e-9); a5 = __builtin_fma(a5, b, 1e-9); a6 = __builtin_fma(a6, b, 1e-9); a7 = __builtin_fma(a7, b, 1e-9);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_rsq_chain(double *out, long long *t, double x) {
  double a = x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) a = __builtin_amdgcn_rsq(a) + 1.0;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_lds_rt(double *out, long long *t, double x) {
  __shared__ double s[64];
  double a = x + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) { s[threadIdx.x] = a; __builtin_amdgcn_wave_barrier(); a = s[(threadIdx.x + 1) & 63] + 1.0; __builtin_amdgcn_wave_barrier(); }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_readlane(double *out, long long *t, double x) {
  double a = x + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int lo = __builtin_amdgcn_readlane(__double2loint(a), u), hi = __builtin_amdgcn_readlane(__double2hiint(a), u);
      a = a * 0.5 + __hiloint2double(hi, lo);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_lds_atomic(double *out, long long *t, int stride, int reps) {
  extern __shared__ double sa[];
  for (int i = threadIdx.x; i < 12288; i += blockDim.x) sa[i] = 0;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  int idx = (threadIdx.x * stride) % 12288;
  for (int i = 0; i < reps; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) atomicAdd(&sa[(idx + u * 37) % 12288], 1.0);
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
  out[threadIdx.x] = sa[threadIdx.x];
}
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma_f64(double *out, long long *t, double x, int nacc) {
  d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  double a = x + threadIdx.x, b = x * 0.5;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; i += 4) {
    if (nacc == 1) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3]; if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  double *out; long long *t; hipMalloc(&out, 8 * 1024); hipMalloc(&t, 8 * 1024);
  long long h[8];
  auto rep = [&](const char *name, double per) { hipMemcpy(h, t, 8, hipMemcpyDeviceToHost); printf("%-28s %8lld cycles total, %.1f per op\n", name, h[0], h[0] / per); };
  for (int w = 0; w < 2; w++) {
    k_fma_chain<<<1, 64>>>(out, t, 1.0); hipDeviceSynchronize(); rep("dependent v_fma_f64", N);
    k_fma_indep<<<1, 64>>>(out, t, 1.0); hipDeviceSynchronize(); rep("independent v_fma_f64 (x8)", N);
    k_rsq_chain<<<1, 64>>>(out, t, 2.0); hipDeviceSynchronize(); rep("dependent rsq+add", N);
    k_lds_rt<<<1, 64>>>(out, t, 1.0); hipDeviceSynchronize(); rep("lds write->read roundtrip", N);
    k_readlane<<<1, 64>>>(out, t, 1.0); hipDeviceSynchronize(); rep("readlane x2 + fma", N);
    k_mfma_f64<<<1, 64>>>(out, t, 1.0, 1); hipDeviceSynchronize(); rep("mfma f64 16x16x4 dependent, 1 wave", N);
    k_mfma_f64<<<1, 64>>>(out, t, 1.0, 4); hipDeviceSynchronize(); rep("mfma f64 16x16x4 4 acc, 1 wave", N);
    k_mfma_f64<<<1, 512>>>(out, t, 1.0, 4); hipDeviceSynchronize(); rep("mfma f64 16x16x4 4 acc, 8 waves", N);
    k_mfma_f64<<<1, 1024>>>(out, t, 1.0, 4); hipDeviceSynchronize(); rep("mfma f64 16x16x4 4 acc, 16 waves", N);
    for (int threads : {1024}) for (int stride : {37}) {
      k_lds_atomic<<<1, threads, 12288 * 8>>>(out, t, stride, 64); hipDeviceSynchronize();
      hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
      printf("lds atomicAdd f64 threads=%4d stride=%2d: %8lld cycles, %.2f cycles per wave-instr\n", threads, stride, h[0], (double)h[0] / (64 * 8 * (threads / 64)));
    }
  }
  return 0;
}
