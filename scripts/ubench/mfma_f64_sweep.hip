// Micro-benchmark (development): v_mfma_f64_16x16x4_f64 rate vs independent accumulators and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(double* out, int iters, unsigned long long* cyc) {
  v4d acc[NACC];
  for (int t = 0; t < NACC; t++) acc[t] = (v4d){0, 0, 0, 0};
  double a[4], b[4];
  for (int q = 0; q < 4; q++) { a[q] = threadIdx.x * 1e-6 + 1.0 + q; b[q] = 1.0 - threadIdx.x * 1e-7 * (q + 1); }
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t & 3], b[(t >> 2) & 3], acc[t], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < NACC; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
void run(int blocks, int threads) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, threads>>>(out, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<NACC><<<blocks, threads>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nwaves = (double)blocks * threads / 64, nm = (double)NACC * iters;
  printf("NACC %2d  waves/SIMD %.1f  %8.3f ms  cycles per MFMA per wave %6.1f   chip %6.2f TFLOP/s\n", NACC, nwaves / 1024.0, ms, (double)c / nm,
         nwaves * nm * 2048 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int wps : {1, 2, 4, 8}) {
    run<2>(256 * wps, 256);
    run<4>(256 * wps, 256);
    run<10>(256 * wps, 256);
    run<20>(256 * wps, 256);
  }
  return 0;
}
