// Development microbenchmark behind K2's ring route: how fast can a CU fill its LDS with
// runs of 144-byte W records, by LDS-DMA and by register staging (global_load_dwordx4 + ds_write_b128)?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_fill.hip -o scripts/ubench_fill
// One workgroup per CU (256); workgroups b, b + 8, ... share an XCD and a 1/16 stretch of a 50 MB
// table pairwise, as the kernel's workgroups do.  Every mover wave walks its own list of runs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const double2 *src, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory");
}

// mode 0: LDS-DMA, wait vmcnt(0) every `batch` ops; mode 1: register staging, `batch` loads in flight then writes
template <int MODE, int BATCH, int RUN7>
__global__ __launch_bounds__(1024) void k_fill(const double2 *W2, int nrec, int nops_per_wave, double *out) {
  extern __shared__ double2 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int stretch = (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) % 2);
  const int my_lo = (int)((long long)nrec * stretch / 16), my_span = nrec / 16 - 80;
  double acc = 0;
  for (int i = 0; i < nops_per_wave; i += BATCH) {
    double2 r[BATCH];
    int dsts[BATCH], ns[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; b++) {
      // (the run is computed, not loaded: a descriptor load would sit in the same in-order queue as the
      // record loads and serialise them; the kernel reads its lists three steps ahead)
      const int it = i + b;
      const int lo = my_lo, span = my_span;
      const int a0 = __builtin_amdgcn_readfirstlane(lo + (int)(((long long)span * it) / nops_per_wave) + ((it * 37 + wave * 11) & 63));
      const int n = RUN7 ? 7 : 1 + ((it * 5 + wave) % 7);
      const int slot = __builtin_amdgcn_readfirstlane((it * 131 + wave * 57) % 960);
      ns[b] = n;
      dsts[b] = slot * 9;
      if (MODE == 0) {
        if (lane < 9 * n) dma16(W2 + (size_t)a0 * 9 + lane, (unsigned)slot * 144u);
      } else {
        r[b] = make_double2(0, 0);
        if (lane < 9 * n) r[b] = W2[(size_t)a0 * 9 + lane];
      }
    }
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int b = 0; b < BATCH; b++)
        if (lane < 9 * ns[b]) lds[dsts[b] + lane] = r[b];
    }
  }
  __syncthreads();
  acc = lds[threadIdx.x].x;
  if (acc == 1.2345) out[0] = acc;
}

int main() {
  const int NCU = 256, NREC = 345364;  // records of 144 B
  double2 *W2; CK(hipMalloc(&W2, (size_t)NREC * 144 + 4096)); CK(hipMemset(W2, 0, (size_t)NREC * 144 + 4096));
  double *out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::mt19937 rng(7);
  const int lds_bytes = 140 * 1024, nslots = lds_bytes / 144;
  for (int runlen : {0, 7}) {            // 0: geometric-ish runs of 1..7 (mean ~4.4), 7: always 7
    for (int nw : {4, 8, 16}) {
      const int bytes_per_cu = 900 * 1024;
      // ops per wave so that a CU moves ~900 KB
      std::vector<int2> ops;
      int per_wave = 0;
      {
        const double mean = runlen ? 7.0 : 4.4;
        per_wave = (int)(bytes_per_cu / (mean * 144) / nw);
        per_wave = (per_wave + 15) / 16 * 16;
      }
      ops.resize((size_t)NCU * nw * per_wave);
      long long recs = 0;
      for (int b = 0; b < NCU; b++) {
        const int stretch = (b % 8) + 8 * ((b / 8) % 2);  // 16 stretches; workgroups of one XCD share two
        const int lo = (int)((long long)NREC * stretch / 16), hi = (int)((long long)NREC * (stretch + 1) / 16) - 8;
        for (int w = 0; w < nw; w++)
          for (int i = 0; i < per_wave; i++) {
            // sweep the stretch front to back in the course of the list, with a random offset (what the kernel does)
            const int pos = lo + (int)((long long)(hi - lo) * i / per_wave) + (int)(rng() % 64);
            int n = runlen ? 7 : 1 + (int)(rng() % 8);
            if (n > 7) n = 7;
            const int slot = (int)(rng() % (nslots - 8));
            ops[((size_t)b * nw + w) * per_wave + i] = make_int2(pos < hi ? pos : hi, slot | (n << 16));
            recs += n;
          }
      }
      int2 *d; CK(hipMalloc(&d, ops.size() * 8)); CK(hipMemcpy(d, ops.data(), ops.size() * 8, hipMemcpyHostToDevice));
      for (int mode = 0; mode < 2; mode++)
        for (int batch : {2, 4, 8}) {
          float ms = 0;
          for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
#define LAUNCH(M, B) do { if (runlen) hipLaunchKernelGGL((k_fill<M, B, 1>), dim3(NCU), dim3(64 * nw), lds_bytes, 0, W2, NREC, per_wave, out); else hipLaunchKernelGGL((k_fill<M, B, 0>), dim3(NCU), dim3(64 * nw), lds_bytes, 0, W2, NREC, per_wave, out); } while (0)
            if (mode == 0) { if (batch == 2) LAUNCH(0, 2); else if (batch == 4) LAUNCH(0, 4); else LAUNCH(0, 8); }
            else { if (batch == 2) LAUNCH(1, 2); else if (batch == 4) LAUNCH(1, 4); else LAUNCH(1, 8); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
          }
          const double bytes = (double)NCU * nw * per_wave * (runlen ? 7.0 : 4.0) * 144;
          printf("runs %s  waves %2d  %s batch %d: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip  %5.0f cycles per op per wave (2.4 GHz)\n",
                 runlen ? "7      " : "1..7   ", nw, mode ? "reg-staged" : "LDS-DMA   ", batch, 1e3 * ms, bytes / NCU / (ms * 1e-3) / 1e9,
                 bytes / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / per_wave);
        }
      CK(hipFree(d));
    }
  }
  return 0;
}
