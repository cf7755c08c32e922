// Development check: accuracy of v_rsq_f64 and of one / two Newton steps on it.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_rsq.hip -o scripts/ubench_rsq
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const double *d, double *y0, double *y1, double *y2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = d[i];
  double y = __builtin_amdgcn_rsq(x);
  y0[i] = y;
  y = y * (1.5 - 0.5 * x * y * y);
  y1[i] = y;
  y = y * (1.5 - 0.5 * x * y * y);
  y2[i] = y;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), a(n), b(n), c(n);
  srand(7);
  for (int i = 0; i < n; i++) h[i] = exp(((double)rand() / RAND_MAX) * 40.0 - 20.0);
  double *d, *y0, *y1, *y2;
  hipMalloc(&d, n * 8); hipMalloc(&y0, n * 8); hipMalloc(&y1, n * 8); hipMalloc(&y2, n * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, y0, y1, y2, n);
  hipMemcpy(a.data(), y0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), y1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), y2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) {
    long double t = 1.0L / sqrtl((long double)h[i]);
    e0 = fmax(e0, (double)fabsl((a[i] - t) / t));
    e1 = fmax(e1, (double)fabsl((b[i] - t) / t));
    e2 = fmax(e2, (double)fabsl((c[i] - t) / t));
  }
  printf("max rel err: rsq %.3e  +1 Newton %.3e  +2 Newton %.3e (eps = %.3e)\n", e0, e1, e2, 2.22e-16);
  return 0;
}
