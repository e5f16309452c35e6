// Micro-benchmark (development, round 2): what does an f64 VALU instruction cost on gfx950, by opcode, with one and with two waves
// per SIMD, and what happens to one wave's f64 VALU stream while the OTHER wave of the SIMD streams f64 MFMAs?
// Build: hipcc --offload-arch=gfx950 -O3 -o fp64_valu_rates fp64_valu_rates.hip ; one workgroup per CU, cycles from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

// OP: 0 fma, 1 mul, 2 add, 3 mixed (mul, fma, add round robin), 4 dependent fma chain, 5 v_mov_b64 (non-fp64 VALU), 6 32-bit integer add
template <int OP>
__global__ void valu_k(double* out, int iters, unsigned long long* cyc) {
  double f[16];
  for (int q = 0; q < 16; q++) f[q] = threadIdx.x * 1e-3 + q;
  int g[16];
  for (int q = 0; q < 16; q++) g[q] = threadIdx.x + q;
  const double a = threadIdx.x * 1e-6 + 1.0, b = 1.0 - threadIdx.x * 1e-7;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      if (OP == 0) f[q] = fma(f[q], b, a);
      if (OP == 1) f[q] = f[q] * b;
      if (OP == 2) f[q] = f[q] + a;
      if (OP == 3) f[q] = (q % 3 == 0) ? f[q] * b : ((q % 3 == 1) ? fma(f[q], b, a) : f[q] + a);
      if (OP == 4) f[0] = fma(f[0], b, a);
      if (OP == 5) asm volatile("v_mov_b64 %0, %1" : "=v"(f[q]) : "v"(f[(q + 1) & 15]));
      if (OP == 6) g[q] = g[q] + g[(q + 1) & 15];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int q = 0; q < 16; q++) s += f[q] + g[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// Waves 0..3 (one per SIMD) stream MFMAs, waves 4..7 (their SIMD partners) stream f64 FMAs (MODE 1) / v_mov_b64 (MODE 2) / nothing (MODE 0);
// MODE 3: waves 0..3 idle, waves 4..7 FMAs (reference).
template <int MODE>
__global__ void cross_k(double* out, int iters, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  v4d acc[5];
  for (int t = 0; t < 5; t++) acc[t] = (v4d){0, 0, 0, 0};
  double f[16];
  for (int q = 0; q < 16; q++) f[q] = threadIdx.x * 1e-3 + q;
  const double a = threadIdx.x * 1e-6 + 1.0, b = 1.0 - threadIdx.x * 1e-7;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (MODE != 3)
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 5; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
  } else {
    if (MODE == 1 || MODE == 3)
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) f[q] = fma(f[q], b, a);
      }
    if (MODE == 2)
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) asm volatile("v_mov_b64 %0, %1" : "=v"(f[q]) : "v"(f[(q + 1) & 15]));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < 5; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int q = 0; q < 16; q++) s += f[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction (256 multiply-adds against 1024 of the 16x16x4 form)
__global__ void mfma4_k(double* out, int iters, unsigned long long* cyc) {
  double acc[8];
  for (int t = 0; t < 8; t++) acc[t] = 0;
  const double a = threadIdx.x * 1e-6 + 1.0, b = 1.0 - threadIdx.x * 1e-7;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[t], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < 8; t++) s += acc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

static double* out;
static unsigned long long* cyc;

template <int OP>
void run_valu(const char* name) {
  const int iters = 2000;
  for (int threads : {256, 512}) {
    valu_k<OP><<<256, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    valu_k<OP><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long c[8];
    hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%-28s %d wave(s)/SIMD: %6.2f cycles per wave-instruction (wave 0), %6.2f per SIMD-instruction\n", name, threads / 256, (double)c[0] / (16.0 * iters),
           (double)c[0] / (16.0 * iters) / (threads / 256));
  }
}
template <int MODE>
void run_cross(const char* name) {
  const int iters = 2000;
  cross_k<MODE><<<256, 512>>>(out, 10, cyc);
  hipDeviceSynchronize();
  cross_k<MODE><<<256, 512>>>(out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c[8];
  hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
  printf("%-60s MFMA wave: %7.1f cycles per MFMA | VALU wave: %6.2f cycles per instruction\n", name, (double)c[0] / (5.0 * iters), (double)c[4] / (16.0 * iters));
}

int main() {
  hipMalloc(&out, sizeof(double) * 256 * 512);
  hipMalloc(&cyc, 64);
  run_valu<0>("v_fma_f64 (independent)");
  run_valu<1>("v_mul_f64");
  run_valu<2>("v_add_f64");
  run_valu<3>("mul / fma / add mix");
  run_valu<4>("v_fma_f64 dependent chain");
  run_valu<5>("v_mov_b64");
  run_valu<6>("v_add_u32");
  for (int threads : {256, 512}) {
    mfma4_k<<<256, threads>>>(out, 10, cyc); hipDeviceSynchronize();
    mfma4_k<<<256, threads>>>(out, 2000, cyc); hipDeviceSynchronize();
    unsigned long long c[8]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("v_mfma_f64_4x4x4_4b_f64          %d wave(s)/SIMD: %6.2f cycles per wave-instruction (256 multiply-adds; the 16x16x4 form does 1024 in 64)\n", threads / 256,
           (double)c[0] / (8.0 * 2000));
  }
  run_cross<0>("MFMA stream alone (partner idle)");
  run_cross<3>("f64 FMA stream alone (partner idle)");
  run_cross<1>("MFMA stream | partner wave: f64 FMA stream");
  run_cross<2>("MFMA stream | partner wave: v_mov_b64 stream");
  return 0;
}
