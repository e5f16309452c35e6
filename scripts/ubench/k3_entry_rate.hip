// Micro-benchmark (development, round 2): cycles per vxm::k3_entry<false> call (phase A's arithmetic: ~232 f64 VALU instructions) with one and
// with two waves per SIMD, inputs in registers, nothing else in the loop.  Tells how much of phase A's measured time is the arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 -I voxel-slam_amd/csrc -o k3_entry_rate k3_entry_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include "vxba_math.hpp"

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k(const double* in, double* out, int iters, unsigned long long* cyc) {
  double P[6], v[3], R[9], p[3], acc[27], rows[3][6];
  const int t = threadIdx.x;
  for (int q = 0; q < 6; q++) P[q] = in[t + q];
  for (int q = 0; q < 3; q++) v[q] = in[t + 6 + q];
  for (int q = 0; q < 9; q++) R[q] = in[t + 9 + q];
  for (int q = 0; q < 3; q++) p[q] = in[t + 18 + q];
  for (int q = 0; q < 27; q++) acc[q] = 0;
  vxm::VoxelCache vc;
  for (int q = 0; q < 3; q++) { vc.u0[q] = in[t + 21 + q]; vc.u1[q] = in[t + 24 + q]; vc.u2[q] = in[t + 27 + q]; vc.vbar[q] = in[t + 30 + q]; }
  vc.s1 = in[t + 33]; vc.s2 = in[t + 34]; vc.invN = in[t + 35]; vc.coe = in[t + 36]; vc.sc = in[t + 37];
  double n = in[t + 38];
  double sink = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    vxm::k3_entry<false>(P, v, n, R, p, vc, rows, acc);
    // feed something back so that iterations are not collapsed, without adding fp64 work worth mentioning
    for (int r = 0; r < 3; r++) sink += rows[r][0] + rows[r][3];
    asm volatile("" : "+v"(P[0]), "+v"(v[0]), "+v"(n));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = sink;
  for (int q = 0; q < 27; q++) s += acc[q];
  out[blockIdx.x * blockDim.x + t] = s;
  if ((t & 63) == 0 && blockIdx.x == 0) cyc[t >> 6] = t1 - t0;
}

int main() {
  double *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 8 * 1024); hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&cyc, 128);
  hipMemset(in, 0, 8 * 1024);
  const int iters = 500;
  auto run = [&](auto tag) {
    constexpr int threads = decltype(tag)::value;
    k<threads><<<256, threads>>>(in, out, 5, cyc); hipDeviceSynchronize();
    k<threads><<<256, threads>>>(in, out, iters, cyc); hipDeviceSynchronize();
    unsigned long long c[16]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%d wave(s)/SIMD: %7.1f cycles per k3_entry call per wave (wave 0), %7.1f (last wave) -> %7.1f per SIMD-call\n", threads / 256, (double)c[0] / iters,
           (double)c[threads / 64 - 1] / iters, (double)c[0] / iters / (threads / 256));
  };
  run(std::integral_constant<int, 256>{});
  run(std::integral_constant<int, 512>{});
  run(std::integral_constant<int, 768>{});
  return 0;
}
