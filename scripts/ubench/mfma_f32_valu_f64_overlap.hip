// Micro-benchmark (development): does an f32 MFMA (v_mfma_f32_16x16x4_f32) overlap with f64 VALU FMAs inside ONE wave on
// gfx950?  (f64 MFMA and f64 VALU do not: mfma_valu_overlap.hip.)  Also probes the f32 C/D lane map.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int VPM>  // MODE 0 MFMA only, 1 VALU only, 2 interleaved (1 MFMA : VPM f64 FMA, pinned)
__global__ void k(double* out, int iters, unsigned long long* cyc) {
  v4f acc[10];
  for (int t = 0; t < 10; t++) acc[t] = (v4f){0, 0, 0, 0};
  double f[8];
  for (int q = 0; q < 8; q++) f[q] = threadIdx.x * 1e-3 + q;
  const double a = threadIdx.x * 1e-6 + 1.0, b = 1.0 - threadIdx.x * 1e-7;
  const float af = (float)a, bf = (float)b;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 10; t++) {
      if (MODE != 1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[t], 0, 0, 0);
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
  printf("%-34s VPM %2d  cycles per group (1 f32 MFMA + VPM f64 FMA): %7.1f\n", name, VPM, (double)c / (10.0 * iters));
  hipFree(out); hipFree(cyc);
}

// layout probe: A[i][k] = 100 i + k, B[k][j] = (k == 0) ? 1 : 0 restricted ... -> D[i][j] = A[i][0]: prints row index per (lane, reg)
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  // operand convention assumed: lane l supplies A[row l%16][k = l/16] and B[k = l/16][col l%16]
  const float a = (float)(100 * (l % 16) + (l / 16));
  const float b = (l / 16 == 0) ? (float)(1 + (l % 16)) : 0.0f;      // B[0][j] = 1 + j
  v4f d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[4 * l + r] = d[r];                    // = (100 i + 0) * (1 + j)
}

int main() {
  run<0, 0>("f32 MFMA only");
  run<1, 4>("f64 VALU only"); run<1, 8>("f64 VALU only"); run<1, 16>("f64 VALU only");
  run<2, 4>("interleaved"); run<2, 8>("interleaved"); run<2, 16>("interleaved");
  float* d; hipMalloc(&d, 256 * sizeof(float));
  probe<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("C/D map of v_mfma_f32_16x16x4_f32 (lane: [row,col] per reg):\n");
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) {
    printf(" lane %2d:", l);
    for (int r = 0; r < 4; r++) {
      // value = 100 i (1 + j): with j in 0..15 and i in 0..15 -> i = round(v / (100 (1+j))) needs j: try all j for integrality
      int fi = -1, fj = -1;
      for (int j = 0; j < 16 && fi < 0; j++) { float q = h[4 * l + r] / (100.0f * (1 + j)); int qi = (int)(q + 0.5f); if (fabsf(q - qi) < 1e-4f && qi < 16 && (qi > 0 || h[4*l+r]==0)) { if (j == l % 16) { fi = qi; fj = j; } } }
      printf(" r%d=[%d,%d](%.0f)", r, fi, fj, h[4 * l + r]);
    }
    printf("\n");
  }
  return 0;
}
