// Development microbenchmark: the rate of the LDS atomic unit for
// the accumulate patterns K2 could use instead of 36 ds_add_f64 per product.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/ubench_ldsatom.hip -o scripts/ubench_ldsatom
// Every kernel runs on all 256 CUs with 1024 threads; times are HIP-event times of the whole launch,
// reported as cycles (at 2.4 GHz nominal) per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NCU = 256;
constexpr double GHZ = 2.4;

// KIND 0: ds_add_f64  1: ds_add_u64  2: ds_add_u32  3: ds_add_f32  4: read-add-write b64 (not atomic)
// 5: ds_add_f64 with only the even lanes active  6: ds_add_f64, lanes of the upper half-wave 18 doubles on
// 7: ds_add_rtn_u64  8: ds_max_u64 (another 64-bit integer op)
// 9: ds_read_b128 + two adds + ds_write_b128 (not atomic; block stride 38 doubles for the alignment), 18 per turn
template <int KIND, int NA>
__global__ __launch_bounds__(1024) void k_add(const int *pos, int iters, int nblk, double *out) {
  extern __shared__ double sm[];
  for (int t = threadIdx.x; t < 38 * nblk; t += blockDim.x) sm[t] = 0;
  __syncthreads();
  const int *r = pos + ((size_t)blockIdx.x * iters) * blockDim.x + threadIdx.x;
  unsigned long long keep = 0;
  for (int it = 0; it < iters; it++) {
    double *b = sm + (KIND == 9 ? 38 : 37) * r[(size_t)it * blockDim.x];
    if (KIND == 6) b += (threadIdx.x & 32) ? 18 : 0;
    if (KIND == 5 && (threadIdx.x & 1)) continue;
#pragma unroll
    for (int k = 0; k < NA; k++) {
      if (KIND == 0 || KIND == 5 || KIND == 6) atomicAdd(&b[k], 1.0 + k);
      if (KIND == 1) atomicAdd(reinterpret_cast<unsigned long long *>(&b[k]), (unsigned long long)(k + 1 + it));
      if (KIND == 2) atomicAdd(reinterpret_cast<unsigned *>(&b[k]), (unsigned)(k + 1 + it));
      if (KIND == 3) atomicAdd(reinterpret_cast<float *>(&b[k]), 1.0f + k);
      if (KIND == 4) b[k] += 1.0 + k;
      if (KIND == 7) keep += atomicAdd(reinterpret_cast<unsigned long long *>(&b[k]), (unsigned long long)(k + 1 + it));
      if (KIND == 8) atomicMax(reinterpret_cast<unsigned long long *>(&b[k]), (unsigned long long)(k + 1 + it));
      if (KIND == 9) {
        double2 *q = reinterpret_cast<double2 *>(b) + k;
        double2 x = *q;
        x.x += 1.0 + k;
        x.y += 2.0 + k;
        *q = x;
      }
    }
  }
  __syncthreads();
  if (sm[threadIdx.x] == 1.2345 || keep == 12345) out[0] = 1;
}

int main() {
  std::mt19937 rng(12345);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double *out; CK(hipMalloc(&out, 64));
  const int nblk = 272, threads = 1024, iters = 24;
  const size_t n = (size_t)NCU * iters * threads;
  std::vector<int> pos(n);
  int *d; CK(hipMalloc(&d, n * 4));
  const char *pats[] = {"16 lanes distinct mod 16", "random blocks", "32 lanes distinct mod 32"};
  const char *kinds[] = {"ds_add_f64", "ds_add_u64", "ds_add_u32", "ds_add_f32", "read+add+write b64", "ds_add_f64 even lanes only",
                         "ds_add_f64 upper half +18", "ds_add_rtn_u64", "ds_max_u64", "read+2 adds+write b128 (x18)"};
  for (int pat = 0; pat < 3; pat++) {
    for (size_t base = 0; base < n; base += 32) {
      int perm[32];
      std::iota(perm, perm + 32, 0); std::shuffle(perm, perm + 32, rng);
      int p16[16];
      std::iota(p16, p16 + 16, 0); std::shuffle(p16, p16 + 16, rng);
      int q16[16];
      std::iota(q16, q16 + 16, 0); std::shuffle(q16, q16 + 16, rng);
      for (int l = 0; l < 32; l++) {
        int v;
        if (pat == 0) v = (l < 16 ? p16[l] : q16[l - 16]) + 16 * (rng() % (nblk / 16));
        else if (pat == 1) v = rng() % nblk;
        else v = perm[l] + 32 * (rng() % (nblk / 32));
        pos[base + l] = v;
      }
    }
    CK(hipMemcpy(d, pos.data(), n * 4, hipMemcpyHostToDevice));
    for (int kind = 0; kind < 10; kind++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        const size_t lds = nblk * 38 * 8;
        switch (kind) {
          case 0: k_add<0, 36><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 1: k_add<1, 36><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 2: k_add<2, 36><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 3: k_add<3, 36><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 4: k_add<4, 36><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 5: k_add<5, 36><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 6: k_add<6, 18><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 7: k_add<7, 36><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 8: k_add<8, 36><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
          case 9: k_add<9, 18><<<NCU, threads, lds>>>(d, iters, nblk, out); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_cu = (double)iters * (kind == 6 || kind == 9 ? 18 : 36) * (threads / 64);
      printf("%-28s %-26s %8.1f us  %6.2f cyc/instr/CU\n", kinds[kind], pats[pat], 1e3 * ms, ms * 1e-3 * GHZ * 1e9 / instr_per_cu);
    }
  }
  return 0;
}
