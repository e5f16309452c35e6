// Micro-benchmark (development): do f64 MFMA and f64 VALU FMA overlap within ONE wave on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE, int VPM>  // MODE 0 MFMA only, 1 VALU only, 2 interleaved (1 MFMA : VPM FMA, pinned)
__global__ void k(double* out, int iters, unsigned long long* cyc) {
  v4d acc[10];
  for (int t = 0; t < 10; t++) acc[t] = (v4d){0, 0, 0, 0};
  double f[8];
  for (int q = 0; q < 8; q++) f[q] = threadIdx.x * 1e-3 + q;
  const double a = threadIdx.x * 1e-6 + 1.0, b = 1.0 - threadIdx.x * 1e-7;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 10; t++) {
      if (MODE != 1) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      if (MODE != 0) {
#pragma unroll
        for (int q = 0; q < VPM; q++) f[q & 7] = fma(f[q & 7], b, a);
      }
      if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < 10; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int q = 0; q < 8; q++) s += f[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int VPM>
void run(const char* name) {
  const int blocks = 256, threads = 256;
  double* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8);
  const int iters = 2000;
  k<MODE, VPM><<<blocks, threads>>>(out, 10, cyc);
  hipDeviceSynchronize();
  k<MODE, VPM><<<blocks, threads>>>(out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s VPM %2d  cycles per group (1 MFMA + VPM FMA): %7.1f\n", name, VPM, (double)c / (10.0 * iters));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0>("MFMA only");
  run<1, 4>("VALU only"); run<1, 8>("VALU only"); run<1, 12>("VALU only"); run<1, 16>("VALU only");
  run<2, 4>("interleaved"); run<2, 8>("interleaved"); run<2, 12>("interleaved"); run<2, 16>("interleaved");
  return 0;
}
