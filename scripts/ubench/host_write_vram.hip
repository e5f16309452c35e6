// Micro-benchmark (development, round 4): can the host write straight into device memory on this box (fine-grained hipExtMallocWithFlags
// memory through the large BAR), how long does a 30 KB record take that way, and what does a kernel pay to read it -- against the same
// record in mapped host memory, which is what the LiDAR-inertial shell's in-launch solve reads today (csrc/vxba_capi_li.hip: zc_lirec)?
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/host_write_vram.hip -o scripts/ubench/host_write_vram && scripts/ubench/host_write_vram
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void reader(const double* __restrict__ rec, int n, double* out, unsigned long long* cyc) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += rec[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = red[0] + red[1] + red[2] + red[3]; cyc[0] = __builtin_readcyclecounter() - t0; }
}
int main() {
  const int n = 3841;   // li_rec_len(10) doubles = 30 KB
  double *vram = nullptr, *hostm = nullptr, *hostm_dev = nullptr, *out = nullptr; unsigned long long* cyc = nullptr;
  CK(hipExtMallocWithFlags((void**)&vram, n * sizeof(double), hipDeviceMallocFinegrained));
  CK(hipHostMalloc((void**)&hostm, n * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&hostm_dev, hostm, 0));
  CK(hipMalloc((void**)&out, 8)); CK(hipMalloc((void**)&cyc, 8));
  std::vector<double> src(n);
  for (int i = 0; i < n; i++) src[i] = 1.0 + i;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, vram) == hipSuccess) printf("fine-grained device allocation: type %d, hostPointer %p, devicePointer %p\n", (int)at.type, at.hostPointer, at.devicePointer);
  for (int mode = 0; mode < 2; mode++) {
    double* dst = mode ? hostm : vram; const double* dsrc = mode ? hostm_dev : vram;
    double best_w = 1e9, best_k = 1e9; unsigned long long best_c = ~0ull; double got = 0;
    for (int rep = 0; rep < 50; rep++) {
      for (int i = 0; i < n; i++) src[i] = rep + i;
      auto t0 = std::chrono::steady_clock::now();
      std::memcpy(dst, src.data(), n * sizeof(double));       // mode 0: CPU stores into VRAM through the BAR (posted writes)
      __sync_synchronize();
      auto t1 = std::chrono::steady_clock::now();
      reader<<<1, 256>>>(dsrc, n, out, cyc);
      CK(hipDeviceSynchronize());
      auto t2 = std::chrono::steady_clock::now();
      unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&got, out, 8, hipMemcpyDeviceToHost));
      best_w = std::min(best_w, std::chrono::duration<double, std::micro>(t1 - t0).count());
      best_k = std::min(best_k, std::chrono::duration<double, std::micro>(t2 - t1).count());
      best_c = std::min(best_c, c);
      const double want = (double)n * rep + (double)n * (n - 1) / 2;
      if (got != want) { printf("mode %d rep %d: WRONG sum %.1f vs %.1f (stale read)\n", mode, rep, got, want); break; }
    }
    printf("%s: host write of 30 KB %.2f us | kernel reads it in %llu cycles (launch + sync %.1f us) | sums right\n", mode ? "mapped host memory (today)" : "fine-grained DEVICE memory, written by the host",
           best_w, best_c, best_k);
  }
  return 0;
}
