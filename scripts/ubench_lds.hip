// Development microbenchmark: cost of one ds_add_f64 wave-instruction under different
// lane -> address patterns (which lanes may share a bank without serialising?).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/ubench_lds.hip -o scripts/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const int *slot, double *out, long long *t, int reps) {
  extern __shared__ double sa[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) sa[i] = 0;
  __syncthreads();
  const int s = slot[threadIdx.x];
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) atomicAdd(&sa[s + u], 1.0);
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) t[0] = t1 - t0;
  out[threadIdx.x] = sa[threadIdx.x];
}
int main() {
  int *d; double *out; long long *t;
  hipMalloc(&d, 4096); hipMalloc(&out, 8192); hipMalloc(&t, 8);
  const char *names[] = {"lane*37 (reference)", "distinct mod 32 per 32 lanes, 2-way mod 16 per 16", "random blocks *37",
                         "pairs of lanes same address", "distinct mod 16 per 16 lanes, same bank across 16-groups",
                         "distinct mod 32 per 32 lanes, same bank across halves", "random distinct-mod-32 per 32 lanes",
                         "random distinct-mod-16 per 16 lanes", "4 lanes same address", "all distinct blocks, random (no mod constraint)"};
  for (int threads : {64, 1024}) for (int p = 0; p < 10; p++) {
    std::vector<int> s(1024);
    srand(1);
    for (int w = 0; w < 16; w++) {
      int perm32[64];
      for (int l = 0; l < 64; l++) {
        int v = 0;
        switch (p) {
          case 0: v = l * 37; break;
          case 1: v = 37 * (((l * 2) % 32 + ((l % 32) >= 16)) + 32 * (l / 32)); break;
          case 2: v = 37 * (rand() % 400); break;
          case 3: v = 37 * (l / 2); break;
          case 4: v = 37 * ((l % 16) + 32 * (l / 16)); break;
          case 5: v = 37 * ((l % 32) + 32 * (l / 32)); break;
          case 6: { if (l % 32 == 0) { for (int k = 0; k < 32; k++) perm32[k] = k; for (int k = 31; k > 0; k--) { int r = rand() % (k + 1); int x = perm32[k]; perm32[k] = perm32[r]; perm32[r] = x; } }
                    v = 37 * (perm32[l % 32] + 32 * (rand() % 12)); break; }
          case 7: { if (l % 16 == 0) { for (int k = 0; k < 16; k++) perm32[k] = k; for (int k = 15; k > 0; k--) { int r = rand() % (k + 1); int x = perm32[k]; perm32[k] = perm32[r]; perm32[r] = x; } }
                    v = 37 * (perm32[l % 16] + 16 * (rand() % 24)); break; }
          case 8: v = 37 * (l / 4); break;
          case 9: { if (l == 0) { for (int k = 0; k < 64; k++) perm32[k] = 0; }
                    int b; bool ok; do { b = rand() % 400; ok = true; for (int k = 0; k < l; k++) if (perm32[k] == b) ok = false; } while (!ok); perm32[l] = b; v = 37 * b; break; }
        }
        s[w * 64 + l] = v;
      }
    }
    hipMemcpy(d, s.data(), 4096, hipMemcpyHostToDevice);
    k<<<1, threads, 16384 * 8>>>(d, out, t, 64); hipDeviceSynchronize();
    k<<<1, threads, 16384 * 8>>>(d, out, t, 64); hipDeviceSynchronize();
    long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("threads=%4d %-60s %.2f ticks per wave-instr\n", threads, names[p], (double)h / (64 * 8 * (threads / 64)));
  }
  return 0;
}
