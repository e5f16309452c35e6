// Device scratch for the stand-alone (handle-less) entry points: one grow-only allocation per device, leased for the duration of a
// call.  These calls are small batched kernels run once per scan (plane fits, cluster builds, plane covariances); paying a handful of
// hipMalloc / hipFree pairs each time costs ten times the kernel.  Calls on one device take turns; requests above 1 GiB get a private
// allocation that is freed again.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>

namespace vxs {

struct Slot {
  std::mutex mtx;
  char* base = nullptr;
  size_t cap = 0;
};
inline Slot* slots() {
  static Slot s[16];
  return s;
}

class Lease {
 public:
  // after construction ok() tells whether `bytes` of device memory are available at base(); the current device must be `device`
  Lease(int device, size_t bytes) {
    if (bytes == 0) bytes = 256;
    if (device < 0 || device >= 16 || bytes > ((size_t)1 << 30)) {
      own_ = hipMalloc((void**)&base_, bytes) == hipSuccess;
      ok_ = own_;
      return;
    }
    slot_ = &slots()[device];
    slot_->mtx.lock();
    if (bytes > slot_->cap) {
      if (slot_->base) { hipDeviceSynchronize(); hipFree(slot_->base); }
      slot_->base = nullptr; slot_->cap = 0;
      const size_t want = bytes + bytes / 2;
      if (hipMalloc((void**)&slot_->base, want) != hipSuccess) return;
      slot_->cap = want;
    }
    base_ = slot_->base;
    ok_ = true;
  }
  ~Lease() {
    if (own_ && base_) hipFree(base_);
    if (slot_) slot_->mtx.unlock();
  }
  Lease(const Lease&) = delete;
  Lease& operator=(const Lease&) = delete;
  bool ok() const { return ok_; }
  // carve the next `bytes` (256-byte aligned) out of the lease
  template <class T>
  T* take(size_t bytes) {
    char* r = base_ + used_;
    used_ += (bytes + 255) / 256 * 256;
    return (T*)r;
  }
  static size_t padded(size_t bytes) { return (bytes + 255) / 256 * 256; }

 private:
  Slot* slot_ = nullptr;
  char* base_ = nullptr;
  size_t used_ = 0;
  bool own_ = false, ok_ = false;
};

}  // namespace vxs
