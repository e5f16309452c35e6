// Round-5 probe for the K3 rebuild: v_mfma_f64_4x4x4_4b_f64 with its four BLOCKS used as four K-slices of ONE 4 x 4 output block.
//
// The instruction multiplies block t of A (4 x 4) with block t of B (4 x 4) for t = 0..3 independently.  If the operand register of
// column group g holds  X_g[lane] = T[row(lane)][4 g + col(lane)]  of a 16-row slab T of the SYRK operand (rows = 16 consecutive K
// rows, i.e. block t = K rows 4t .. 4t+3), then  D = mfma(X_I, X_J)  leaves in block t the partial product over K rows 4t..4t+3 of
// the 4 x 4 output block (I, J), and the sum over the four blocks (once, at the end of the kernel) is T[:, I]^T T[:, J].  One register
// per column group serves as A and as B operand of every pair it takes part in: 15 LDS reads feed 120 instructions.
//
// (1) layout: one-hot tables for A and B -> which lane is (block, k, i); printed compactly and checked against the hypothesis
//     lane = 16 t + 4 k + i  (row = lane >> 2, col = lane & 3).
// (2) numerics: S = T^T T for a random 16 x 60 slab through the scheme above against the host.
// (3) rate: the phase-M inner loop of the planned kernel -- 8 waves, each owning 15 (I, J) pairs, all nine 16-row slabs of a
//     144 x 60 tile in LDS, operands read with ds_read_b64 -- cycles per step, beside today's 16x16x4 arrangement (45 instructions
//     per wave and step, four operand slots).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

__global__ void probe_k(double* D) {   // D[(which * 64 + src) * 64 + lane]
  const int lane = threadIdx.x;
  for (int which = 0; which < 2; which++)
    for (int src = 0; src < 64; src++) {
      const double a = which == 0 ? (lane == src ? 1.0 : 0.0) : 1.0 + lane;
      const double b = which == 1 ? (lane == src ? 1.0 : 0.0) : 1.0 + lane;
      D[(which * 64 + src) * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    }
}

constexpr int RS = 60;   // row stride of the tile in doubles (== 4 mod 8: the eight rows a half-wave reads land on distinct 32-byte slots)
__global__ void syrk_k(const double* T, double* S) {   // T[16][60] -> S[15][15][64] raw accumulators
  __shared__ double t[16 * RS];
  for (int k = threadIdx.x; k < 16 * RS; k += 64) t[k] = T[k];
  __syncthreads();
  const int lane = threadIdx.x, row = lane >> 2, col = lane & 3;
  double x[15];
  for (int g = 0; g < 15; g++) x[g] = t[row * RS + 4 * g + col];
  for (int I = 0; I < 15; I++)
    for (int J = 0; J < 15; J++) S[(I * 15 + J) * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[I], x[J], 0.0, 0, 0, 0);
}

// ---- rate ------------------------------------------------------------------------------------------------------------------
// pair patterns: waves 0-2 "intra" (5 groups, 15 pairs i <= j), waves 3-7 "cross" (3 x 5 groups)
template <bool PREFETCH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void rate_new(double* out, int steps, unsigned long long* cyc) {
  extern __shared__ double tile[];   // 144 x RS
  for (int k = threadIdx.x; k < 144 * RS; k += 512) tile[k] = 1e-3 * (k % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane >> 2, col = lane & 3;
  double acc[15];
  for (int j = 0; j < 15; j++) acc[j] = 0.0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 3) {
    const double* bp = tile + row * RS + 4 * (5 * wave) + col;
    for (int s = 0; s < steps; s++) {
      double x[5], xn[5];
#pragma unroll
      for (int g = 0; g < 5; g++) x[g] = bp[4 * g];
#pragma unroll
      for (int q = 0; q < 9; q++) {
        if (PREFETCH && q + 1 < 9) {
#pragma unroll
          for (int g = 0; g < 5; g++) xn[g] = bp[(q + 1) * 16 * RS + 4 * g];
          __builtin_amdgcn_sched_barrier(0);
        }
        int j = 0;
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
          for (int b = a; b < 5; b++) { acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[a], x[b], acc[j], 0, 0, 0); j++; }
        if (PREFETCH) {
#pragma unroll
          for (int g = 0; g < 5; g++) x[g] = xn[g];
        } else if (q + 1 < 9) {
#pragma unroll
          for (int g = 0; g < 5; g++) x[g] = bp[(q + 1) * 16 * RS + 4 * g];
        }
      }
      __syncthreads();
    }
  } else {
    const int a0 = (wave - 3) % 3 * 3, b0 = 5 + (wave & 1) * 5;
    const double* bpa = tile + row * RS + 4 * a0 + col;
    const double* bpb = tile + row * RS + 4 * b0 + col;
    for (int s = 0; s < steps; s++) {
      double xa[3], xb[5], xan[3], xbn[5];
#pragma unroll
      for (int g = 0; g < 3; g++) xa[g] = bpa[4 * g];
#pragma unroll
      for (int g = 0; g < 5; g++) xb[g] = bpb[4 * g];
#pragma unroll
      for (int q = 0; q < 9; q++) {
        if (PREFETCH && q + 1 < 9) {
#pragma unroll
          for (int g = 0; g < 3; g++) xan[g] = bpa[(q + 1) * 16 * RS + 4 * g];
#pragma unroll
          for (int g = 0; g < 5; g++) xbn[g] = bpb[(q + 1) * 16 * RS + 4 * g];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = 0; b < 5; b++) acc[a * 5 + b] = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[a], xb[b], acc[a * 5 + b], 0, 0, 0);
        if (PREFETCH) {
#pragma unroll
          for (int g = 0; g < 3; g++) xa[g] = xan[g];
#pragma unroll
          for (int g = 0; g < 5; g++) xb[g] = xbn[g];
        } else if (q + 1 < 9) {
#pragma unroll
          for (int g = 0; g < 3; g++) xa[g] = bpa[(q + 1) * 16 * RS + 4 * g];
#pragma unroll
          for (int g = 0; g < 5; g++) xb[g] = bpb[(q + 1) * 16 * RS + 4 * g];
        }
      }
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double sum = 0;
  for (int j = 0; j < 15; j++) sum += acc[j];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void rate_old(double* out, int steps, unsigned long long* cyc) {
  extern __shared__ double tile[];   // 144 x 64, pair-interleaved layout not reproduced: plain rows, 16-lane contiguous reads
  for (int k = threadIdx.x; k < 144 * 64; k += 512) tile[k] = 1e-3 * (k % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lrow = lane >> 4, lcol = lane & 15;
  v4d acc[5];
  for (int j = 0; j < 5; j++) acc[j] = (v4d){0, 0, 0, 0};
  const int kq = wave >> 1;
  const double* bp = tile + (4 * 9 * kq + lrow) * 64 + lcol;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; s++) {
    double x[4], xn[4];
#pragma unroll
    for (int g = 0; g < 4; g++) x[g] = bp[16 * g];
#pragma unroll
    for (int q = 0; q < 9; q++) {
      if (q + 1 < 9) {
#pragma unroll
        for (int g = 0; g < 4; g++) xn[g] = bp[(q + 1) * 4 * 64 + 16 * g];
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], x[1], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], x[2], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], x[3], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], x[1], acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], x[2], acc[4], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; g++) x[g] = xn[g];
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double sum = 0;
  for (int j = 0; j < 5; j++) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
  double* D; hipMalloc(&D, 8 * 2 * 64 * 64);
  probe_k<<<1, 64>>>(D); hipDeviceSynchronize();
  std::vector<double> h(2 * 64 * 64); hipMemcpy(h.data(), D, h.size() * 8, hipMemcpyDeviceToHost);
  // hypothesis: A lane s = (t, k, i) = (s >> 4, (s >> 2) & 3, s & 3) feeds D lanes (t, i, j) for j = 0..3 with B lane (t, k, j); D lane = 16 t + 4 ? ...
  // print for every A source lane: output lanes and partner B lanes
  for (int which = 0; which < 2; which++) {
    printf("%s one-hot source lane -> output lane:partner lane\n", which == 0 ? "A" : "B");
    for (int src = 0; src < 64; src++) {
      printf("  %2d:", src);
      for (int l = 0; l < 64; l++) { const double v = h[(which * 64 + src) * 64 + l]; if (v != 0.0) printf(" %d:%d", l, (int)(v - 1.0 + 0.5)); }
      printf("\n");
    }
  }
  // numerics
  {
    std::vector<double> T(16 * RS), S(15 * 15 * 64);
    srand(7);
    for (auto& v : T) v = (rand() % 2001 - 1000) * 1e-3;
    double *dT, *dS; hipMalloc(&dT, T.size() * 8); hipMalloc(&dS, S.size() * 8);
    hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice);
    syrk_k<<<1, 64>>>(dT, dS); hipDeviceSynchronize();
    hipMemcpy(S.data(), dS, S.size() * 8, hipMemcpyDeviceToHost);
    // try the output maps lane = 16 t + 4 i + j  and  lane = 16 t + 4 j + i
    for (int map = 0; map < 2; map++) {
      double worst = 0;
      for (int I = 0; I < 15; I++)
        for (int J = 0; J < 15; J++)
          for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
              double ref = 0;
              for (int r = 0; r < 16; r++) ref += T[r * RS + 4 * I + i] * T[r * RS + 4 * J + j];
              double got = 0;
              for (int t = 0; t < 4; t++) got += S[(I * 15 + J) * 64 + 16 * t + (map == 0 ? 4 * i + j : 4 * j + i)];
              worst = fmax(worst, fabs(got - ref));
            }
      printf("SYRK through blocks-as-K, output map %s: max |err| = %.3e\n", map == 0 ? "16t+4i+j" : "16t+4j+i", worst);
    }
  }
  // rates
  double* out; unsigned long long* cyc; hipMalloc(&out, 8 * 512 * 256); hipMalloc(&cyc, 256);
  const int steps = 200;
  auto report = [&](const char* tag) {
    unsigned long long c[8]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%-52s", tag);
    for (int w = 0; w < 8; w++) printf(" %6.0f", (double)c[w] / steps);
    printf("   cycles per step, per wave\n");
  };
  hipFuncSetAttribute((const void*)rate_new<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * RS * 8);
  hipFuncSetAttribute((const void*)rate_new<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * RS * 8);
  hipFuncSetAttribute((const void*)rate_old, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 64 * 8);
  for (int rep = 0; rep < 2; rep++) {
    rate_new<true><<<256, 512, 144 * RS * 8>>>(out, steps, cyc); hipDeviceSynchronize(); if (rep) report("new: 4x4x4_4b, 15 pairs / wave, prefetched operands");
    rate_new<false><<<256, 512, 144 * RS * 8>>>(out, steps, cyc); hipDeviceSynchronize(); if (rep) report("new: 4x4x4_4b, 15 pairs / wave, no prefetch");
    rate_old<<<256, 512, 144 * 64 * 8>>>(out, steps, cyc); hipDeviceSynchronize(); if (rep) report("old: 16x16x4, 5 tiles x 9 K-steps / wave");
  }
  printf("expected: new 135 x 17 = 2295 per wave -> 4590 per SIMD (two waves); old 45 x 64 = 2880 -> 5760\n");
  return 0;
}
