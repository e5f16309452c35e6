// C-ABI implementation, multi-GPU attachment (include/vxba.h): the caller's all-reduce hook, RCCL communicators created from the
// library the process already has, and the hipIpc mailboxes of the one-shot peer all-reduce (kernel: vxba_capi_core.hip).
// Reference analogue of what is exchanged: the thread fan-in of voxel_map.hpp:298-335, across GPUs.
#include "vxba_capi_internal.hpp"

#include <dlfcn.h>

using namespace vxc;

extern "C" {

int vxba_set_allreduce(vxba_factor* f, vxba_allreduce_fn fn, void* ctx) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  f->allreduce = fn;
  f->allreduce_ctx = ctx;
  return VXBA_OK;
}

namespace {
// NULL / "": the RCCL the loader finds ("librccl.so", already mapped when the process uses one), else ROCm's own copy
void* open_rccl(const char* path) {
  if (path && path[0]) return dlopen(path, RTLD_NOW | RTLD_LOCAL);
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  return h;
}
}  // namespace

int vxba_rccl_unique_id(const char* librccl_path, void* out) {
  if (!out) return VXBA_ERR_ARG;
  void* lib = open_rccl(librccl_path);
  if (!lib) return VXBA_ERR_STATE;
  auto fn = (ncclResult_t(*)(ncclUniqueId*))dlsym(lib, "ncclGetUniqueId");
  if (!fn) return VXBA_ERR_STATE;
  ncclUniqueId id;
  if (fn(&id) != ncclSuccess) return VXBA_ERR_STATE;
  std::memcpy(out, &id, sizeof id);
  return VXBA_OK;
}

int vxba_rccl_attach(vxba_factor* f, const char* librccl_path, int nranks, int rank, const void* unique_id) {
  VX_LOCK(f);
  if (!f || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return fail(f, VXBA_ERR_ARG, "rccl_attach: bad argument");
  if (f->rccl_comm) return fail(f, VXBA_ERR_STATE, "rccl_attach: already attached");
  hipSetDevice(f->device);
  void* lib = open_rccl(librccl_path);
  if (!lib) return fail(f, VXBA_ERR_STATE, "rccl_attach: cannot dlopen librccl");
  auto init = (ncclResult_t(*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(lib, "ncclCommInitRank");
  f->p_ncclAllReduce = (decltype(f->p_ncclAllReduce))dlsym(lib, "ncclAllReduce");
  f->p_ncclCommDestroy = (decltype(f->p_ncclCommDestroy))dlsym(lib, "ncclCommDestroy");
  if (!init || !f->p_ncclAllReduce || !f->p_ncclCommDestroy) return fail(f, VXBA_ERR_STATE, "rccl_attach: missing RCCL symbols");
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  if (init(&f->rccl_comm, nranks, id, rank) != ncclSuccess) { f->rccl_comm = nullptr; return fail(f, VXBA_ERR_STATE, "ncclCommInitRank failed"); }
  f->rccl_lib = lib;
  return VXBA_OK;
}

// The same with the id exchange done through a caller-supplied broadcast (MPI_Bcast, a socket, a file ...): rank 0 creates the id,
// bcast(ctx, buf, 128, root = 0) must leave rank 0's bytes in every rank's buf.  No torch, no Python.
int vxba_rccl_attach_bcast(vxba_factor* f, const char* librccl_path, int nranks, int rank, vxba_bcast_fn bcast, void* ctx) {
  if (!f || !bcast || nranks < 1 || rank < 0 || rank >= nranks) return fail(f, VXBA_ERR_ARG, "rccl_attach_bcast: bad argument");
  unsigned char id[128];
  std::memset(id, 0, sizeof id);
  if (rank == 0) {
    int rc = vxba_rccl_unique_id(librccl_path, id);
    if (rc != VXBA_OK) return fail(f, rc, "rccl_attach_bcast: ncclGetUniqueId failed (librccl not found?)");
  }
  if (bcast(ctx, id, sizeof id, 0) != 0) return fail(f, VXBA_ERR_STATE, "rccl_attach_bcast: the broadcast callback failed");
  return vxba_rccl_attach(f, librccl_path, nranks, rank, id);
}

int vxba_rccl_detach(vxba_factor* f) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  if (f->rccl_comm) {
    hipSetDevice(f->device);
    hipStreamSynchronize(f->stream);
    f->p_ncclCommDestroy(f->rccl_comm);
    f->rccl_comm = nullptr;
  }
  return VXBA_OK;
}

// ---- vxba_peer_*: the mailbox, its IPC handle, the peers' mappings ------------------------------------------------------------
static size_t peer_box_bytes(size_t len) { return (2 * len + 2 * PEER_WGS + 1) * sizeof(double); }
int vxba_peer_export(vxba_factor* f, void* handle_out) {
  VX_LOCK(f);
  if (!f || !handle_out) return fail(f, VXBA_ERR_ARG, "peer_export: null argument");
  hipSetDevice(f->device);
  if (!f->peer.box) {
    const size_t len = vxba_packed_len(f) + 1;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, peer_box_bytes(len), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      return fail(f, VXBA_ERR_HIP, "peer_export: cannot allocate fine-grained device memory for the mailbox");
    }
    VX_HIP(f, hipMemset(p, 0, peer_box_bytes(len)));
    f->peer.box = (double*)p;
    f->peer.len = len;
  }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, f->peer.box) != hipSuccess) { (void)hipGetLastError(); return fail(f, VXBA_ERR_HIP, "peer_export: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"); }
  static_assert(sizeof h == VXBA_PEER_HANDLE_BYTES, "IPC handle size");
  std::memcpy(handle_out, &h, sizeof h);
  return VXBA_OK;
}
int vxba_peer_attach(vxba_factor* f, int nranks, int rank, const void* handles) {
  VX_LOCK(f);
  if (!f || !handles || nranks < 1 || nranks > VXBA_PEER_MAX || rank < 0 || rank >= nranks) return fail(f, VXBA_ERR_ARG, "peer_attach: bad argument");
  if (!f->peer.box) return fail(f, VXBA_ERR_STATE, "peer_attach: call vxba_peer_export first");
  if (f->peer.nranks) return fail(f, VXBA_ERR_STATE, "peer_attach: already attached");
  if (is_wide(f)) return fail(f, VXBA_ERR_UNSUPPORTED, "peer_attach: windows wider than 10 frames use RCCL (2.9 MB buffers are bandwidth-bound)");
  hipSetDevice(f->device);
  for (int p = 0; p < nranks; p++) {
    if (p == rank) { f->peer.boxes[p] = f->peer.box; continue; }
    hipIpcMemHandle_t h;
    std::memcpy(&h, (const char*)handles + (size_t)p * VXBA_PEER_HANDLE_BYTES, sizeof h);
    void* q = nullptr;
    if (hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      for (int k = 0; k < p; k++) if (f->peer.opened[k]) { hipIpcCloseMemHandle(f->peer.opened[k]); f->peer.opened[k] = nullptr; }
      return fail(f, VXBA_ERR_HIP, "peer_attach: hipIpcOpenMemHandle failed (peer not reachable / IPC disabled)");
    }
    f->peer.opened[p] = q;
    f->peer.boxes[p] = (double*)q;
  }
  // The mailbox outlives detach: its per-slot flags still hold the call numbers of the previous attachment and its status word a
  // timeout that may have ended it.  With the sequence restarting at 0 those stale flags would satisfy the first waits (stale slices
  // summed silently, the self-test passing on the old pattern), so they are cleared here -- the caller's barrier between attach
  // and the first collective (vxba.h) orders the clearing before any peer's first read.
  if (hipMemsetAsync(f->peer.box + 2 * f->peer.len, 0, (2 * PEER_WGS + 1) * sizeof(double), f->stream) != hipSuccess || hipStreamSynchronize(f->stream) != hipSuccess) {
    (void)hipGetLastError();
    for (int p = 0; p < nranks; p++) if (f->peer.opened[p]) { hipIpcCloseMemHandle(f->peer.opened[p]); f->peer.opened[p] = nullptr; }
    return fail(f, VXBA_ERR_HIP, "peer_attach: cannot reset the mailbox flags");
  }
  f->peer.nranks = nranks; f->peer.rank = rank; f->peer.seq = 0;
  return VXBA_OK;
}
int vxba_peer_detach(vxba_factor* f) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  if (f->stream) hipStreamSynchronize(f->stream);
  for (int p = 0; p < VXBA_PEER_MAX; p++) {
    if (f->peer.opened[p]) hipIpcCloseMemHandle(f->peer.opened[p]);
    f->peer.opened[p] = nullptr; f->peer.boxes[p] = nullptr;
  }
  f->peer.nranks = 0;
  return VXBA_OK;
}
// Collective self-test (call on every rank after vxba_peer_attach, before trusting the link): every rank contributes rank + 1 in
// every element of a full-length buffer and must read back N (N + 1) / 2.  *ok = 0 on a wrong sum or a peer that never arrived.
int vxba_peer_selftest(vxba_factor* f, int* ok) {
  VX_LOCK(f);
  if (!f || !ok) return VXBA_ERR_ARG;
  *ok = 0;
  if (!has_peer(f)) return fail(f, VXBA_ERR_STATE, "peer_selftest: not attached");
  hipSetDevice(f->device);
  const size_t n = vxba_packed_len(f) + 1;
  std::vector<double> h(n, (double)(f->peer.rank + 1));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  VX_HIP(f, hipMemcpyAsync(f->own_packed, h.data(), n * sizeof(double), hipMemcpyHostToDevice, f->stream));
  int rc = shard_allreduce(f, f->own_packed, n);
  if (rc) return rc;
  VX_HIP(f, hipMemcpyAsync(h.data(), f->own_packed, n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  int st = 0;
  rc = vxba_peer_status(f, &st);
  if (rc) return rc;
  const double want = 0.5 * f->peer.nranks * (f->peer.nranks + 1);
  bool good = st == 0;
  for (size_t i = 0; i < n && good; i++) good = h[i] == want;
  *ok = good ? 1 : 0;
  return VXBA_OK;
}
// 0: fine; 1: a peer never raised its flag within the spin bound (results of that call are not a sum: the caller must stop)
int vxba_peer_status(vxba_factor* f, int* status) {
  VX_LOCK(f);
  if (!f || !status) return VXBA_ERR_ARG;
  *status = 0;
  if (!f->peer.box) return VXBA_OK;
  hipSetDevice(f->device);
  VX_HIP(f, hipStreamSynchronize(f->stream));
  unsigned long long st = 0;
  VX_HIP(f, hipMemcpy(&st, f->peer.box + 2 * f->peer.len + 2 * PEER_WGS, sizeof st, hipMemcpyDeviceToHost));
  *status = (int)st;
  return VXBA_OK;
}

int vxba_use_external_buffers(vxba_factor* f, double* d_packed, double* d_scalar) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  VX_HIP(f, hipStreamSynchronize(f->stream));
  f->d_packed = d_packed ? d_packed : f->own_packed;
  f->d_scalar = d_scalar ? d_scalar : f->own_scalar;
  return VXBA_OK;
}

}  // extern "C"
