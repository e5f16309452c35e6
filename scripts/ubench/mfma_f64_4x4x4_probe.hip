// Micro-benchmark + layout probe (development, round 2) for v_mfma_f64_4x4x4_4b_f64 on gfx950:
//  (1) throughput with 1..4 waves per SIMD, every wave reported (is a partner wave starved, as with the 16x16x4 form?);
//  (2) does a partner wave's f64 VALU stream survive beside it;
//  (3) the operand / result lane layout, found by setting one A (or B) lane to 1 and everything else of that operand to 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int THREADS>
__global__ __launch_bounds__(THREADS) void rate_k(double* out, int iters, unsigned long long* cyc, int valu_partner) {
  const int wave = threadIdx.x >> 6;
  double acc[16];
  for (int t = 0; t < 16; t++) acc[t] = 0;
  double f[16];
  for (int q = 0; q < 16; q++) f[q] = threadIdx.x * 1e-3 + q;
  const double a = threadIdx.x * 1e-6 + 1.0, b = 1.0 - threadIdx.x * 1e-7;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (valu_partner && wave >= 4) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 16; q++) f[q] = fma(f[q], b, a);
    }
  } else {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int t = 0; t < 16; t++) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < 16; t++) s += acc[t] + f[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

__global__ void probe_k(double* D) {   // D[(which * 64 + src) * 64 + lane]: which 0 = A one-hot, 1 = B one-hot
  const int lane = threadIdx.x;
  for (int which = 0; which < 2; which++)
    for (int src = 0; src < 64; src++) {
      const double a = which == 0 ? (lane == src ? 1.0 : 0.0) : 1.0 + lane;      // the other operand: distinct values 1 + lane
      const double b = which == 1 ? (lane == src ? 1.0 : 0.0) : 1.0 + lane;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      D[(which * 64 + src) * 64 + lane] = d;
    }
}

int main() {
  double* out; unsigned long long* cyc; double* D;
  hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&cyc, 256); hipMalloc(&D, 8 * 2 * 64 * 64);
  const int iters = 2000;
  auto report = [&](int waves, const char* tag) {
    unsigned long long c[16]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%-44s", tag);
    for (int w = 0; w < waves; w++) printf(" w%d %6.2f", w, (double)c[w] / (16.0 * iters));
    printf("   (cycles per instruction, per wave)\n");
  };
  rate_k<256><<<256, 256>>>(out, 10, cyc, 0); rate_k<256><<<256, 256>>>(out, iters, cyc, 0); hipDeviceSynchronize(); report(4, "4x4x4_4b, 1 wave / SIMD");
  rate_k<512><<<256, 512>>>(out, 10, cyc, 0); rate_k<512><<<256, 512>>>(out, iters, cyc, 0); hipDeviceSynchronize(); report(8, "4x4x4_4b, 2 waves / SIMD");
  rate_k<768><<<256, 768>>>(out, 10, cyc, 0); rate_k<768><<<256, 768>>>(out, iters, cyc, 0); hipDeviceSynchronize(); report(12, "4x4x4_4b, 3 waves / SIMD");
  rate_k<1024><<<256, 1024>>>(out, 10, cyc, 0); rate_k<1024><<<256, 1024>>>(out, iters, cyc, 0); hipDeviceSynchronize(); report(16, "4x4x4_4b, 4 waves / SIMD");
  rate_k<512><<<256, 512>>>(out, 10, cyc, 1); rate_k<512><<<256, 512>>>(out, iters, cyc, 1); hipDeviceSynchronize(); report(8, "waves 0-3 4x4x4_4b | waves 4-7 f64 FMA");
  probe_k<<<1, 64>>>(D); hipDeviceSynchronize();
  std::vector<double> h(2 * 64 * 64); hipMemcpy(h.data(), D, h.size() * 8, hipMemcpyDeviceToHost);
  // which output lanes does A lane s feed, and with which B lane's value?
  for (int which = 0; which < 2; which++) {
    printf("%s one-hot source lane -> (output lane : partner operand lane) ...\n", which == 0 ? "A" : "B");
    for (int src = 0; src < 64; src += (src < 20 ? 1 : 7)) {
      printf("  src %2d:", src);
      for (int l = 0; l < 64; l++) { const double v = h[(which * 64 + src) * 64 + l]; if (v != 0.0) printf(" %d:%d", l, (int)(v - 1.0 + 0.5)); }
      printf("\n");
    }
  }
  return 0;
}
