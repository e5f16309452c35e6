// Micro-benchmark (development): f64 MFMA 16x16x4 issue rate, f64 VALU FMA rate, and whether the two overlap on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 fp64_peaks.hip -o fp64_peaks ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: MFMA only, 1: VALU only, 2: both interleaved (1 MFMA : 8 FMA), 3: both, clustered (10 MFMA then 80 FMA)
__global__ __launch_bounds__(256, 1) void k(double* out, int iters, unsigned long long* cyc) {
  v4d acc[10];
  for (int t = 0; t < 10; t++) acc[t] = (v4d){0, 0, 0, 0};
  double f[16];
  for (int q = 0; q < 16; q++) f[q] = threadIdx.x * 1e-3 + q;
  const double a = threadIdx.x * 1e-6 + 1.0, b = 1.0 - threadIdx.x * 1e-7;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
      for (int t = 0; t < 10; t++) {
        if (MODE != 1) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
          for (int q = 0; q < 8; q++) f[(t * 8 + q) & 15] = fma(f[(t * 8 + q) & 15], b, a);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (MODE == 1 || MODE == 3) {
      if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 80; q++) f[q & 15] = fma(f[q & 15], b, a);
      if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < 10; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int q = 0; q < 16; q++) s += f[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = (MODE == 1) ? 0 : 10.0 * iters, vf = (MODE == 0) ? 0 : 80.0 * iters;
  const double flops = ((double)blocks * 4) * (mf * 2048 + vf * 128);
  printf("%-28s blocks %4d  %8.3f ms  %10llu cycles/wave  per-iter %7.1f cycles  %7.2f TFLOP/s  (clock %.2f GHz)\n", name, blocks, ms, c,
         (double)c / iters, flops / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {256, 512}) {
    run<0>("MFMA f64 16x16x4 x10", blocks);
    run<1>("VALU f64 FMA x80", blocks);
    run<2>("interleaved 1 MFMA : 8 FMA", blocks);
    run<3>("clustered 10 MFMA + 80 FMA", blocks);
  }
  return 0;
}
