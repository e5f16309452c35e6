// C-ABI implementation (include/vxba.h): device memory, streams, launch sequencing and the host part of
// the LM shell.  No CPU fallback anywhere: without a gfx950 device every entry point fails loudly.
#include "vxba_factor.hpp"

namespace vxc {

int fail(vxba_factor* f, int code, const char* msg) {
  if (f) f->err = msg;
  return code;
}

int n_planes(const vxba_factor* f) { return (f->W > VXBA_MAX_WIN ? 0 : 10 * f->W) + N_META_PLANES; }   // wide factors keep their clusters in f->wstore
// win_size above VXBA_MAX_WIN: the sparse-incidence sweeps of vxba_wide.hip and the host-side LM shell
bool is_wide(const vxba_factor* f) { return f->W > VXBA_MAX_WIN; }

// exchange buffers sized for the current win_size: packed [Hess | JacT | residual] and, directly behind it, the scalar of the
// residual sweep -- contiguous so that the sharded loop can reduce both with one collective
int ensure_exchange(vxba_factor* f) {
  const size_t plen = (size_t)36 * f->W * f->W + 6 * f->W + 1;
  if (plen + 1 <= f->xlen) {
    if (f->d_scalar == f->own_scalar) f->d_scalar = f->own_packed + plen;
    f->own_scalar = f->own_packed + plen;
    return VXBA_OK;
  }
  if (f->stream) VX_HIP(f, hipStreamSynchronize(f->stream));
  const bool own_p = !f->d_packed || f->d_packed == f->own_packed;
  const bool own_s = !f->d_scalar || f->d_scalar == f->own_scalar;
  if (f->own_packed) VX_HIP(f, hipFree(f->own_packed));
  if (f->h_packed) VX_HIP(f, hipHostFree(f->h_packed));
  f->own_packed = nullptr; f->own_scalar = nullptr; f->h_packed = nullptr; f->zc_packed = nullptr; f->xlen = 0;
  VX_HIP(f, hipMalloc((void**)&f->own_packed, (plen + 1) * sizeof(double)));
  VX_HIP(f, hipHostMalloc((void**)&f->h_packed, (plen + 1) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: see li_damping_iter_queued
  VX_HIP(f, hipHostGetDevicePointer((void**)&f->zc_packed, f->h_packed, 0));
  f->own_scalar = f->own_packed + plen;
  if (own_p) f->d_packed = f->own_packed;
  if (own_s) f->d_scalar = f->own_scalar;
  f->xlen = plen + 1;
  return VXBA_OK;
}

FactorView view(const vxba_factor* f) {
  FactorView fv;
  const size_t VS = (size_t)f->VS;
  double* p = f->planes;
  fv.clb = f->clb;
  fv.cl = p;                     p += f->W > VXBA_MAX_WIN ? 0 : (size_t)10 * f->W * VS;   // wide: no cluster planes (never dereferenced)
  fv.fix = p;                    p += 10 * VS;
  fv.coe = p;                    p += VS;
  fv.eigval = p;                 p += 3 * VS;
  fv.eigvec = p;                 p += 9 * VS;
  fv.merged = p;                 p += 10 * VS;
  fv.aux = p;
  fv.cl32 = nullptr;
  fv.VS = f->VS;
  fv.W = f->W;
  return fv;
}

int residual_view(vxba_factor* f, FactorView& fv) {
  fv = view(f);
  if (f->precision != VXBA_PRECISION_MIXED_F32_CLUSTERS || is_wide(f) || f->V == 0) return VXBA_OK;
  if (!f->cl32 || f->cl32_vs != f->VS) {
    if (f->cl32) { VX_HIP(f, hipStreamSynchronize(f->stream)); VX_HIP(f, hipFree(f->cl32)); f->cl32 = nullptr; }
    VX_HIP(f, hipMalloc((void**)&f->cl32, (size_t)10 * f->W * f->VS * sizeof(float)));
    f->cl32_vs = f->VS;
    f->cl32_built = 0;
  }
  fv.cl32 = f->cl32;
  if (f->cl32_built < f->V) {
    vxk::launch_build_cl32(fv, f->cl32_built, f->V - f->cl32_built, f->stream);
    VX_HIP(f, hipGetLastError());
    f->cl32_built = f->V;
  }
  return VXBA_OK;
}
// cluster planes of voxels >= v0 were (re)written: their f32 copies are stale
static inline void clusters_written(vxba_factor* f, int v0) { if (v0 < f->cl32_built) f->cl32_built = v0; }

vxw::WideView wview(const vxba_factor* f) { return vxw::wide_view(view(f), f->wstore); }

int ensure_staging(vxba_factor* f, size_t len) {
  if (len <= f->staging_len) return VXBA_OK;
  if (f->staging) { VX_HIP(f, hipStreamSynchronize(f->stream)); VX_HIP(f, hipFree(f->staging)); f->staging = nullptr; f->staging_len = 0; }
  len = std::max(len, (size_t)1 << 16);
  VX_HIP(f, hipMalloc((void**)&f->staging, len * sizeof(double)));
  f->staging_len = len;
  return VXBA_OK;
}

int ensure_capacity(vxba_factor* f, int n_total) {
  if (n_total <= f->VS) return VXBA_OK;
  int want = std::max(n_total, 2 * f->VS);
  want = (want + 63) / 64 * 64;
  double* np = nullptr;
  const size_t bytes = (size_t)n_planes(f) * want * sizeof(double);
  VX_HIP(f, hipMalloc((void**)&np, bytes));
  VX_HIP(f, hipMemsetAsync(np, 0, bytes, f->stream));
  if (f->planes && f->V > 0) vxk::launch_copy_planes(f->planes, f->VS, np, want, n_planes(f), f->V, f->stream);
  double* nclb = nullptr;
  if (!is_wide(f)) {   // the batch-major copy feeds the MFMA sweep only
    const size_t clb_bytes = vxk::k3_clb_len(f->W, want) * sizeof(double);
    VX_HIP(f, hipMalloc((void**)&nclb, clb_bytes));
    VX_HIP(f, hipMemsetAsync(nclb, 0, clb_bytes, f->stream));
    if (f->clb && f->V > 0)   // batches are absolute, so the old copy is a prefix of the new one
      VX_HIP(f, hipMemcpyAsync(nclb, f->clb, vxk::k3_clb_len(f->W, f->V) * sizeof(double), hipMemcpyDeviceToDevice, f->stream));
  }
  if (f->planes) { VX_HIP(f, hipStreamSynchronize(f->stream)); VX_HIP(f, hipFree(f->planes)); if (f->clb) VX_HIP(f, hipFree(f->clb)); }
  f->planes = np;
  f->clb = nclb;
  f->VS = want;
  if (is_wide(f)) {
    const char* emsg = nullptr;
    if (vxw::store_reserve(f->wstore, want, f->wstore.ES, f->V, f->stream, &emsg) != 0) return fail(f, VXBA_ERR_HIP, emsg ? emsg : "wide store: allocation failed");
  }
  const size_t p2 = (size_t)want / 32 + 2;   // one partial per workgroup of 32..64 voxels (vxk::k2_voxels_per_block)
  if (p2 > f->partial2_len) {
    if (f->d_partial2) VX_HIP(f, hipFree(f->d_partial2));
    VX_HIP(f, hipMalloc((void**)&f->d_partial2, p2 * sizeof(double)));
    if (f->h_partial2) VX_HIP(f, hipHostFree(f->h_partial2));
    f->h_partial2 = nullptr; f->zc_partial2 = nullptr;
    VX_HIP(f, hipHostMalloc((void**)&f->h_partial2, p2 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: a partial is visible to the host when its workgroup has written it
    VX_HIP(f, hipHostGetDevicePointer((void**)&f->zc_partial2, f->h_partial2, 0));
    f->partial2_len = p2;
  }
  return VXBA_OK;
}

int ensure_partials3(vxba_factor* f) {
  const size_t need = (size_t)vxk::k3_grid_blocks(f->cus) * vxk::k3_partial_len(f->W);
  if (need <= f->partial3_len) return VXBA_OK;
  if (f->d_partial3) VX_HIP(f, hipFree(f->d_partial3));
  VX_HIP(f, hipMalloc((void**)&f->d_partial3, need * sizeof(double)));
  f->partial3_len = need;
  return VXBA_OK;
}

int check_range(vxba_factor* f, int head, int end) {
  if (head < 0 || end < head || end > f->V) return fail(f, VXBA_ERR_ARG, "voxel range [head,end) outside the factor");
  return VXBA_OK;
}

void fill_poses(const vxba_factor* f, const double* Rp, PoseArg& pa) {
  std::memset(&pa, 0, sizeof pa);
  std::memcpy(pa.Rp, Rp, sizeof(double) * 12 * f->W);
}

// ---- profiling helpers ----
hipEvent_t get_event(vxba_factor* f) {
  if (!f->free_events.empty()) { hipEvent_t e = f->free_events.back(); f->free_events.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
struct ScopedKernelTimer {
  vxba_factor* f; int kind; int on = 0; hipEvent_t a = nullptr, b = nullptr;
  ScopedKernelTimer(vxba_factor* f_, int kind_) : f(f_), kind(kind_) {
    on = (f->profiling >> kind_) & 1;
    if (on) { a = get_event(f); b = get_event(f); if (a) hipEventRecord(a, f->stream); }
  }
  ~ScopedKernelTimer() {
    if (on && a && b) { hipEventRecord(b, f->stream); f->pending.push_back({a, b, kind}); }
  }
};
int drain_events(vxba_factor* f) {
  if (f->pending.empty()) return VXBA_OK;
  VX_HIP(f, hipStreamSynchronize(f->stream));
  for (auto& ep : f->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) { f->ms_sum[ep.kind] += ms; f->calls[ep.kind]++; }
    f->free_events.push_back(ep.a);
    f->free_events.push_back(ep.b);
  }
  f->pending.clear();
  return VXBA_OK;
}

// ---- one-shot all-reduce over the peers' mailboxes ---------------------------------------------------------------------------
// The exchange buffer of the sharded LM loop is 29 KB: under a ring collective that is pure latency (2 (N - 1) hops).  Here every
// rank publishes its buffer in its own mailbox and reads the N - 1 others directly over xGMI -- one hop, all links at once.
// Workgroup w owns slice w of the buffer on every rank: it copies its slice into the local mailbox (slot = call parity), fences,
// raises flag[slot][w] = call number, then waits for the same flag of every peer and adds the peers' slices in RANK ORDER (own
// slice included, read back from the mailbox), so all ranks compute bit-identical sums.  Double buffering is enough: a rank can
// only reach call k + 2 after every peer raised its flags for call k + 1, i.e. finished reading call k.  Mailboxes are fine-grained
// device memory and are read with system-scope loads (no stale lines of call k - 2 from a non-coherent cache).
constexpr int PEER_WGS = 8, PEER_THREADS = 256;
struct PeerArgs { double* boxes[VXBA_PEER_MAX]; int nranks, rank; unsigned long long len; };
__device__ __forceinline__ unsigned long long* peer_flags(double* box, unsigned long long len) { return reinterpret_cast<unsigned long long*>(box + 2 * len); }
__global__ __launch_bounds__(PEER_THREADS) void peer_allreduce_kernel(PeerArgs a, double* __restrict__ buf, unsigned long long count, unsigned long long seq) {
  const int w = blockIdx.x, tid = threadIdx.x;
  const unsigned slot = (unsigned)(seq & 1);
  const unsigned long long per = (count + PEER_WGS - 1) / PEER_WGS, lo = per * w, hi = lo + per < count ? lo + per : count;
  double* mine = a.boxes[a.rank] + slot * a.len;
  for (unsigned long long i = lo + tid; i < hi; i += PEER_THREADS) mine[i] = buf[i];
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(peer_flags(a.boxes[a.rank], a.len) + slot * PEER_WGS + w, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __shared__ int failed;
  if (tid == 0) {
    int bad = 0;
    for (int p = 0; p < a.nranks && !bad; p++) {
      const unsigned long long* fl = peer_flags(a.boxes[p], a.len) + slot * PEER_WGS + w;
      long long spins = 0;
      while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1ll << 23)) { bad = 1; break; }       // a peer that never arrives: a few seconds, then give up loudly
      }
    }
    failed = bad;
    if (bad) __hip_atomic_store(peer_flags(a.boxes[a.rank], a.len) + 2 * PEER_WGS, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // status word
  }
  __syncthreads();
  if (failed) return;
  for (unsigned long long i = lo + tid; i < hi; i += PEER_THREADS) {
    double s = 0.0;
    for (int p = 0; p < a.nranks; p++) {
      const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(a.boxes[p] + slot * a.len + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      s += __longlong_as_double((long long)bits);
    }
    buf[i] = s;
  }
}
bool has_peer(const vxba_factor* f) { return f->peer.nranks > 1; }

// Sum `count` f64 across the voxel shards, stream-ordered: the peers' mailboxes if attached, else direct RCCL, else the caller's hook.
bool has_collective(const vxba_factor* f) { return has_peer(f) || f->rccl_comm != nullptr || f->allreduce != nullptr; }
int shard_allreduce(vxba_factor* f, double* d_buf, size_t count) {
  ScopedKernelTimer timer(f, 4);   // profiling bit 16: events around the collective on the factor's stream (kernel + the wait for the peers)
  if (has_peer(f)) {
    if (count > f->peer.len) return fail(f, VXBA_ERR_STATE, "peer all-reduce: buffer larger than the mailbox");
    PeerArgs a;
    for (int p = 0; p < VXBA_PEER_MAX; p++) a.boxes[p] = f->peer.boxes[p];
    a.nranks = f->peer.nranks; a.rank = f->peer.rank; a.len = f->peer.len;
    peer_allreduce_kernel<<<PEER_WGS, PEER_THREADS, 0, f->stream>>>(a, d_buf, count, ++f->peer.seq);
    return VXBA_OK;
  }
  if (f->rccl_comm) {
    if (f->p_ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, f->rccl_comm, f->stream) != ncclSuccess)
      return fail(f, VXBA_ERR_STATE, "ncclAllReduce failed");
    return VXBA_OK;
  }
  if (f->allreduce && f->allreduce(f->allreduce_ctx, d_buf, count, (void*)f->stream) != 0) return fail(f, VXBA_ERR_STATE, "all-reduce hook failed");
  return VXBA_OK;
}

bool fused_solve(const vxba_factor* f) { return f->opt[VXBA_OPT_FUSED_SOLVE] != 0; }
void options_from_env(vxba_factor* f) {   // initial values only; vxba_set_option is the interface
  auto flag = [](const char* name, int dflt) { const char* e = getenv(name); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : dflt; };
  f->opt[VXBA_OPT_FUSED_SOLVE] = flag("VXBA_FUSED_SOLVE", 1);
  f->opt[VXBA_OPT_SPEC_COLLECTIVE] = flag("VXBA_SPEC_COLLECTIVE", 1);
  f->opt[VXBA_OPT_WIDE_DEVICE_SOLVE] = flag("VXBA_WIDE_DEVICE_SOLVE", 1);
  f->opt[VXBA_OPT_LI_DEVICE_LOOP] = flag("VXBA_LI_DEVICE", 0);
  const char* e = getenv("VXBA_K2_VPB");
  const int v = e ? atoi(e) : 64;
  f->opt[VXBA_OPT_K2_VOXELS_PER_BLOCK] = (v >= 32 && v <= 64) ? v : 64;
}

// The sweeps are asynchronous: two calls in a row with different poses must not share one staging buffer (the second memcpy
// would overwrite it before the first H2D copy has run).  Eight pinned slots, each reused only after its copy has completed.
int upload_poses(vxba_factor* f, const double* Rp) {
  const unsigned slot = f->pose_slot++ & 7u;
  if (!f->pose_ev[slot]) VX_HIP(f, hipEventCreateWithFlags(&f->pose_ev[slot], hipEventDisableTiming));
  else VX_HIP(f, hipEventSynchronize(f->pose_ev[slot]));
  double* h = f->h_poses + (size_t)slot * 12 * VXBA_MAX_WIN_WIDE;
  std::memcpy(h, Rp, sizeof(double) * 12 * f->W);
  VX_HIP(f, hipMemcpyAsync(f->d_poses, h, sizeof(double) * 12 * f->W, hipMemcpyHostToDevice, f->stream));
  VX_HIP(f, hipEventRecord(f->pose_ev[slot], f->stream));
  return VXBA_OK;
}

// ---- sweeps (asynchronous on f->stream; results in device memory) ----
// Stand-alone mode: poses by value (Rp, host pointer -> kernel argument), lm == nullptr.
// LM mode (lm != nullptr): the sweep's prologue takes the pending accept/reject decision from ctl[*c] (and flips *c),
// reads the poses from the control block and skips the work when the loop does not need it; Rp carries the restart poses.
int sweep_hess_device(vxba_factor* f, const double* Rp, vxk::LMState* lm, int* c, const vxk::LMPending* pend, int head, int end,
                      double* d_out, const double* cache_src) {
  const size_t plen = vxba_packed_len(f);
  if (end == head) { VX_HIP(f, hipMemsetAsync(d_out, 0, plen * sizeof(double), f->stream)); return VXBA_OK; }
  if (is_wide(f)) {   // sparse-incidence sweep, host-driven LM only (lm == nullptr)
    if (lm || !Rp) return fail(f, VXBA_ERR_UNSUPPORTED, "device-resident LM loop: only for win_size <= VXBA_MAX_WIN");
    int rcw = upload_poses(f, Rp);
    if (rcw) return rcw;
    if (f->wide_dirty || f->wide.V != f->V) {
      const char* emsg = nullptr;
      if (vxw::build_index(wview(f), f->V, f->wide, f->stream, &emsg) != 0) return fail(f, VXBA_ERR_HIP, emsg ? emsg : "wide index build failed");
      f->wide_dirty = false;
    }
    {
      ScopedKernelTimer t(f, 0);
      vxw::launch_k3_wide(wview(f), f->d_poses, f->wide, head, end, d_out, f->d_partial2, f->stream);
    }
    VX_HIP(f, hipGetLastError());
    return shard_allreduce(f, d_out, plen);
  }
  int rc = ensure_partials3(f);
  if (rc) return rc;
  PoseArg pa;
  if (Rp) fill_poses(f, Rp, pa); else std::memset(&pa, 0, sizeof pa);
  const FactorView fv = view(f);
  // one wave per batch of NV voxels, eight waves per workgroup; never launch more workgroups than there are steps of eight batches
  const int nv = vxk::k3_nv(f->W);
  const int nbatches = (end - 1) / nv - head / nv + 1;
  const int nblocks = vxk::k3_blocks_for(nbatches, vxk::k3_grid_blocks(f->cus));
  vxk::LMPending none;
  std::memset(&none, 0, sizeof none);
  const vxk::LMPending& pd = pend ? *pend : none;
  const int c_in = c ? *c : 0;
  if (lm && pd.pending) *c ^= 1;   // the prologue persists the decision into the other control block
  const int c_now = c ? *c : 0;
  if (f->profiling & 1) {   // events bound to the dispatch itself: same interval as the profiler's kernel duration
    hipEvent_t a = get_event(f), b = get_event(f);
    vxk::launch_k3_hessian(fv, pa, lm, c_in, pd, cache_src, head, end, f->d_partial3, nblocks, f->precision, f->stream, a, b);
    if (a && b) f->pending.push_back({a, b, 0});
  } else {
    vxk::launch_k3_hessian(fv, pa, lm, c_in, pd, cache_src, head, end, f->d_partial3, nblocks, f->precision, f->stream);
  }
  {
    ScopedKernelTimer t(f, 2);
    // with a collective the LM state is filled after the all-reduce, from the reduced buffer
    vxk::launch_k3_finalize(f->d_partial3, nblocks, f->W, lm, c_now, has_collective(f) ? 0 : 1, d_out, f->stream);
  }
  VX_HIP(f, hipGetLastError());
  rc = shard_allreduce(f, d_out, plen);
  if (rc) return rc;
  if (lm && has_collective(f)) vxk::launch_lm_unpack(lm, c_now, d_out, f->W, f->stream);
  return VXBA_OK;
}

// partials_to_host: the block partials go straight to mapped host memory (h_partial2) and no sum is launched -- the caller adds them up
// with host_sum_partials once the sweep is done (d_out is ignored).
int sweep_residual_device(vxba_factor* f, const double* Rp, vxk::LMState* lm, int c, int head, int end, double* d_out,
                          int* nparts_out, unsigned fused_seq, bool partials_to_host) {
  if (end == head) { if (d_out) VX_HIP(f, hipMemsetAsync(d_out, 0, sizeof(double), f->stream)); return VXBA_OK; }
  if (is_wide(f)) {
    if (lm || !Rp || !d_out) return fail(f, VXBA_ERR_UNSUPPORTED, "device-resident LM loop: only for win_size <= VXBA_MAX_WIN");
    int rcw = upload_poses(f, Rp);
    if (rcw) return rcw;
    int np;
    {
      ScopedKernelTimer t(f, 1);
      np = vxw::launch_k2_wide(wview(f), f->d_poses, head, end, f->d_partial2, f->stream);
    }
    if (nparts_out) *nparts_out = np;
    vxk::launch_sum_partials(f->d_partial2, np, d_out, f->stream);
    VX_HIP(f, hipGetLastError());
    return shard_allreduce(f, d_out, 1);
  }
  PoseArg pa;
  if (Rp) fill_poses(f, Rp, pa); else std::memset(&pa, 0, sizeof pa);
  FactorView fv;
  { const int rcv = residual_view(f, fv); if (rcv) return rcv; }
  int nparts;
  double* const part = (partials_to_host && !is_wide(f)) ? f->zc_partial2 : f->d_partial2;
  if (partials_to_host) d_out = nullptr;
  if (f->profiling & 2) {
    hipEvent_t a = get_event(f), b = get_event(f);
    nparts = vxk::launch_k2_residual(fv, pa, lm, c, fused_seq, head, end, part, f->opt[VXBA_OPT_K2_VOXELS_PER_BLOCK] | (f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] ? 0x10000 : 0), f->stream, a, b);
    if (a && b) f->pending.push_back({a, b, 1});
  } else {
    nparts = vxk::launch_k2_residual(fv, pa, lm, c, fused_seq, head, end, part, f->opt[VXBA_OPT_K2_VOXELS_PER_BLOCK] | (f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] ? 0x10000 : 0), f->stream);
  }
  if (nparts_out) *nparts_out = nparts;
  if (d_out) {
    vxk::launch_sum_partials(f->d_partial2, nparts, d_out, f->stream);
    VX_HIP(f, hipGetLastError());
    return shard_allreduce(f, d_out, 1);
  }
  return VXBA_OK;
}

// Sharded (collective) LM loop, speculative form: ONE all-reduce per iteration.  The Hessian sweep of iteration i+1 linearises at
// the trial poses of iteration i before anybody knows whether they are accepted; its reduction also carries the trial residual
// (the residual sweep's partial sums) in the slot behind the packed buffer; after the single all-reduce a small kernel takes the
// accept/reject decision from the reduced residual and adopts the reduced system if the step was accepted.  A rejected step wastes
// that sweep (the reference recomputes nothing then); in exchange every iteration saves one latency-bound collective and the
// partial-sum kernel.  Needs the scalar exchange buffer directly behind the packed one (true for the factor's own buffers and for
// dist.attach_allreduce's tensor); VXBA_SPEC_COLLECTIVE=0 falls back to the two-collective loop.
bool spec_collective(const vxba_factor* f) {
  return f->opt[VXBA_OPT_SPEC_COLLECTIVE] != 0 && has_collective(f) && !is_wide(f) && f->d_scalar == f->d_packed + vxba_packed_len(f);
}

int spec_hess_phase(vxba_factor* f, const double* Rp0, int* c, bool first_of_solve, bool has_pending, bool restart, const double* cache_src,
                    int k2_nparts) {
  int rc = ensure_partials3(f);
  if (rc) return rc;
  PoseArg pa;
  fill_poses(f, Rp0, pa);
  const FactorView fv = view(f);
  const int nv = vxk::k3_nv(f->W);
  const int nbatches = (f->V - 1) / nv + 1;
  const int nblocks = vxk::k3_blocks_for(nbatches, vxk::k3_grid_blocks(f->cus));
  vxk::LMPending pd;
  std::memset(&pd, 0, sizeof pd);
  pd.pending = first_of_solve ? 3 : 2;
  if (f->profiling & 1) {
    hipEvent_t a = get_event(f), b = get_event(f);
    vxk::launch_k3_hessian(fv, pa, f->d_lm, *c, pd, cache_src, 0, f->V, f->d_partial3, nblocks, f->precision, f->stream, a, b);
    if (a && b) f->pending.push_back({a, b, 0});
  } else {
    vxk::launch_k3_hessian(fv, pa, f->d_lm, *c, pd, cache_src, 0, f->V, f->d_partial3, nblocks, f->precision, f->stream);
  }
  {
    ScopedKernelTimer t(f, 2);
    vxk::launch_k3_finalize(f->d_partial3, nblocks, f->W, f->d_lm, *c, 0, f->d_packed, f->stream, 1, has_pending ? f->d_partial2 : nullptr, k2_nparts);
  }
  VX_HIP(f, hipGetLastError());
  rc = shard_allreduce(f, f->d_packed, vxba_packed_len(f) + 1);
  if (rc) return rc;
  vxk::launch_lm_spec_unpack(f->d_lm, *c, f->d_packed, f->W, has_pending ? 1 : 0, restart ? 1 : 0, pa, f->stream);
  if (has_pending) *c ^= 1;
  return VXBA_OK;
}

// closes a speculative loop: the last trial's residual still needs its own (scalar) all-reduce and decision
int spec_final_decision(vxba_factor* f, const double* Rp0, int* c, int k2_nparts) {
  vxk::launch_sum_partials(f->d_partial2, k2_nparts, f->d_scalar, f->stream);
  VX_HIP(f, hipGetLastError());
  int rc = shard_allreduce(f, f->d_scalar, 1);
  if (rc) return rc;
  PoseArg pa;
  fill_poses(f, Rp0, pa);
  vxk::LMPending pend;
  std::memset(&pend, 0, sizeof pend);
  pend.pending = 1;
  pend.d_scalar = f->d_scalar;
  vxk::launch_lm_update(f->d_lm, *c, pend, pa, f->W, f->stream);
  *c ^= 1;
  return VXBA_OK;
}

int sweep_hess_host(vxba_factor* f, const double* Rp, int head, int end) {
  int rc = sweep_hess_device(f, Rp, nullptr, nullptr, nullptr, head, end, f->d_packed);
  if (rc) return rc;
  VX_HIP(f, hipMemcpyAsync(f->h_packed, f->d_packed, vxba_packed_len(f) * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  return VXBA_OK;
}
int sweep_residual_host(vxba_factor* f, const double* Rp, int head, int end, double* residual) {
  int rc = sweep_residual_device(f, Rp, nullptr, 0, head, end, f->d_scalar);
  if (rc) return rc;
  VX_HIP(f, hipMemcpyAsync(f->h_scalar, f->d_scalar, sizeof(double), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  *residual = f->h_scalar[0];
  return VXBA_OK;
}

// Append per-voxel metadata rows (fix, coe and optionally the cache) for n voxels at offset v0.
int append_meta(vxba_factor* f, int v0, int n, const double* fix, const double* coe, const double* eig_val, const double* eig_vec,
                const double* merged) {
  const FactorView fv = view(f);
  const size_t per = 10 + 1 + 3 + 9 + 10;
  int rc = ensure_staging(f, (size_t)n * per);
  if (rc) return rc;
  double* s = f->staging;
  std::vector<double> tmp;
  if (!fix) tmp.assign((size_t)n * 10, 0.0);
  VX_HIP(f, hipMemcpyAsync(s, fix ? fix : tmp.data(), sizeof(double) * n * 10, hipMemcpyHostToDevice, f->stream));
  vxk::launch_scatter_rows(s, fv.fix, f->VS, v0, n, 10, f->stream);
  s += (size_t)n * 10;
  std::vector<double> ones;
  if (!coe) ones.assign(n, 1.0);
  VX_HIP(f, hipMemcpyAsync(s, coe ? coe : ones.data(), sizeof(double) * n, hipMemcpyHostToDevice, f->stream));
  vxk::launch_scatter_rows(s, fv.coe, f->VS, v0, n, 1, f->stream);
  s += n;
  if (eig_val && eig_vec && merged) {
    VX_HIP(f, hipMemcpyAsync(s, eig_val, sizeof(double) * n * 3, hipMemcpyHostToDevice, f->stream));
    vxk::launch_scatter_rows(s, fv.eigval, f->VS, v0, n, 3, f->stream);
    s += (size_t)n * 3;
    VX_HIP(f, hipMemcpyAsync(s, eig_vec, sizeof(double) * n * 9, hipMemcpyHostToDevice, f->stream));
    vxk::launch_scatter_rows(s, fv.eigvec, f->VS, v0, n, 9, f->stream);
    s += (size_t)n * 9;
    VX_HIP(f, hipMemcpyAsync(s, merged, sizeof(double) * n * 10, hipMemcpyHostToDevice, f->stream));
    vxk::launch_scatter_rows(s, fv.merged, f->VS, v0, n, 10, f->stream);
    vxk::launch_seed_aux(fv, v0, v0 + n, f->stream);
  }
  // host temporaries (tmp/ones) and the caller's arrays must outlive the async copies
  VX_HIP(f, hipStreamSynchronize(f->stream));
  VX_HIP(f, hipGetLastError());
  return VXBA_OK;
}

}  // namespace vxc
using namespace vxc;

extern "C" {

int vxba_create(int win_size, int device, vxba_factor** out) {
  if (!out) return VXBA_ERR_ARG;
  *out = nullptr;
  if (win_size < 1 || win_size > VXBA_MAX_WIN_WIDE) return VXBA_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VXBA_ERR_NODEV;
  if (device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return VXBA_ERR_NODEV;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VXBA_ERR_NODEV;  // kernels are built for gfx950 only
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  vxba_factor* f = new vxba_factor();
  f->W = win_size;
  f->device = device;
  f->cus = prop.multiProcessorCount;
  options_from_env(f);
  auto bail = [&](hipError_t) { vxba_destroy(f); return VXBA_ERR_HIP; };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&f->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
  f->stream = f->own_stream;
  if (ensure_exchange(f) != VXBA_OK) return bail(hipErrorOutOfMemory);
  if ((e = hipMalloc((void**)&f->d_count, sizeof(unsigned long long))) != hipSuccess) return bail(e);
  if ((e = hipMalloc((void**)&f->d_poses, sizeof(double) * 12 * VXBA_MAX_WIN_WIDE)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc((void**)&f->h_poses, sizeof(double) * 8 * 12 * VXBA_MAX_WIN_WIDE, hipHostMallocDefault)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc((void**)&f->h_scalar, 2 * sizeof(double), hipHostMallocDefault)) != hipSuccess) return bail(e);
  if ((e = hipMalloc((void**)&f->d_lm, sizeof(vxk::LMState))) != hipSuccess) return bail(e);
  if ((e = hipMemset(f->d_lm, 0, sizeof(vxk::LMState))) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc((void**)&f->h_lm, sizeof(vxk::LMState), hipHostMallocDefault)) != hipSuccess) return bail(e);
  *out = f;
  return VXBA_OK;
}

int vxba_destroy(vxba_factor* f) {
  if (!f) return VXBA_OK;
  hipSetDevice(f->device);
  if (f->stream) hipStreamSynchronize(f->stream);
  vxba_rccl_detach(f);
  vxba_peer_detach(f);
  if (f->peer.box) hipFree(f->peer.box);
  for (auto& ep : f->pending) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
  for (auto e : f->free_events) hipEventDestroy(e);
  hipFree(f->planes); hipFree(f->clb); hipFree(f->cl32); hipFree(f->snapshot); hipFree(f->staging); hipFree(f->d_partial3); hipFree(f->d_partial2); if (f->h_partial2) hipHostFree(f->h_partial2);
  if (f->h_feed) (void)hipHostFree(f->h_feed);
  if (f->li_ev2) (void)hipEventDestroy(f->li_ev2);
  if (f->li_ev3) (void)hipEventDestroy(f->li_ev3);
  vxw::free_index(f->wide);
  vxw::store_free(f->wstore);
  vxw::wide_solver_free(f->wide_solver);
  hipFree(f->own_packed); hipFree(f->d_count); hipFree(f->d_poses);
  if (f->h_poses) hipHostFree(f->h_poses);
  for (auto& ev : f->pose_ev) if (ev) hipEventDestroy(ev);
  if (f->li_ev) hipEventDestroy(f->li_ev);
  if (f->h_packed) hipHostFree(f->h_packed);
  if (f->h_scalar) hipHostFree(f->h_scalar);
  hipFree(f->d_lm); hipFree(f->d_li); hipFree(f->d_li_hess); hipFree(f->d_scratch);
  if (f->h_lm) hipHostFree(f->h_lm);
  if (f->own_stream) hipStreamDestroy(f->own_stream);
  delete f;
  return VXBA_OK;
}

int vxba_clear(vxba_factor* f) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  f->V = 0;
  f->cl32_built = 0;
  f->wstore.nnz = 0;
  f->wide_dirty = true;
  f->snapshot_v = 0;
  return VXBA_OK;
}

int vxba_set_win_size(vxba_factor* f, int win_size) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  if (win_size < 1 || win_size > VXBA_MAX_WIN_WIDE) return fail(f, VXBA_ERR_UNSUPPORTED, "win_size outside [1, VXBA_MAX_WIN_WIDE]");
  if (win_size == f->W) return VXBA_OK;
  if (f->V != 0) return fail(f, VXBA_ERR_STATE, "win_size can only change on an empty factor");
  hipSetDevice(f->device);
  if (f->planes) { VX_HIP(f, hipStreamSynchronize(f->stream)); VX_HIP(f, hipFree(f->planes)); if (f->clb) VX_HIP(f, hipFree(f->clb)); f->planes = nullptr; f->clb = nullptr; }
  if (f->cl32) { VX_HIP(f, hipFree(f->cl32)); f->cl32 = nullptr; f->cl32_vs = 0; f->cl32_built = 0; }
  f->VS = 0;
  vxw::store_free(f->wstore);
  vxw::free_index(f->wide);
  f->W = win_size;
  return ensure_exchange(f);
}

int vxba_win_size(const vxba_factor* f) { return f ? f->W : 0; }
int vxba_size(const vxba_factor* f) { return f ? f->V : 0; }
size_t vxba_packed_len(const vxba_factor* f) { return f ? (size_t)36 * f->W * f->W + 6 * f->W + 1 : 0; }
const char* vxba_last_error(const vxba_factor* f) { return f ? f->err.c_str() : "null factor"; }

int vxba_set_stream(vxba_factor* f, void* hip_stream) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  VX_HIP(f, hipStreamSynchronize(f->stream));
  f->stream = hip_stream ? (hipStream_t)hip_stream : f->own_stream;
  return VXBA_OK;
}

int vxba_reserve(vxba_factor* f, int n_voxels) {
  VX_LOCK(f);
  if (!f || n_voxels < 0) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  return ensure_capacity(f, n_voxels);
}

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

namespace {
// Wide factors are filled by a few big pushes (a top-level window of the hierarchical BA), not by a stream of small ones: the
// grow-only staging / scratch buffers of the push calls are handed back afterwards when they are large (they would otherwise
// outweigh the compressed-row store itself).
void release_push_buffers(vxba_factor* f) {
  if (!is_wide(f)) return;
  const size_t keep = (size_t)16 << 20;
  if (f->staging && f->staging_len * sizeof(double) > keep) { (void)hipStreamSynchronize(f->stream); (void)hipFree(f->staging); f->staging = nullptr; f->staging_len = 0; }
  if (f->d_scratch && f->scratch_cap > keep) { (void)hipStreamSynchronize(f->stream); (void)hipFree(f->d_scratch); f->d_scratch = nullptr; f->scratch_cap = 0; }
}
// wide factors: n voxels whose clusters sit densely on the device (d_dense[n][W][10], N == 0 = unobserved) -> appended to the store
int wide_append_dense(vxba_factor* f, int n, const double* d_dense, int frame_major = 0) {
  const char* emsg = nullptr;
  const long long added = vxw::store_append_dense(f->wstore, f->V, n, d_dense, f->W, f->V, f->stream, &emsg, frame_major);
  if (added < 0) return fail(f, VXBA_ERR_HIP, emsg ? emsg : "wide store: append failed");
  f->wstore.nnz += added;
  return VXBA_OK;
}
}  // namespace

int vxba_push_voxels(vxba_factor* f, int n, const double* clusters, const double* fix, const double* coe, const double* eig_val,
                     const double* eig_vec, const double* merged) {
  VX_LOCK(f);
  if (!f || n < 0 || (n > 0 && (!clusters || !fix || !coe))) return fail(f, VXBA_ERR_ARG, "push_voxels: null input");
  if (n == 0) return VXBA_OK;
  for (int a = 0; a < n; a++)
    if (!(coe[a] >= 0.0)) return fail(f, VXBA_ERR_ARG, "push_voxels: coe must be >= 0");
  hipSetDevice(f->device);
  int rc = ensure_capacity(f, f->V + n);
  if (rc) return rc;
  const size_t ncl = (size_t)n * f->W * 10;
  rc = ensure_staging(f, ncl);
  if (rc) return rc;
  VX_HIP(f, hipMemcpyAsync(f->staging, clusters, ncl * sizeof(double), hipMemcpyHostToDevice, f->stream));
  const long long nnz0 = f->wstore.nnz;
  if (is_wide(f)) { rc = wide_append_dense(f, n, f->staging); if (rc) return rc; }
  else {
    vxk::launch_scatter_clusters(f->staging, view(f), f->V, n, f->stream);
    vxk::launch_build_clb(view(f), f->V, n, f->stream);
    clusters_written(f, f->V);
  }
  VX_HIP(f, hipStreamSynchronize(f->stream));
  rc = append_meta(f, f->V, n, fix, coe, eig_val, eig_vec, merged);
  if (rc) { f->wstore.nnz = nnz0; return rc; }
  f->V += n;
  f->wide_dirty = true;
  release_push_buffers(f);
  return VXBA_OK;
}

namespace {
// the factor's grow-only device scratch (staging of host arrays on their way into the planes): no hipMalloc / hipFree per call
int ensure_scratch(vxba_factor* f, size_t need) {
  if (need > f->scratch_cap) {
    VX_HIP(f, hipStreamSynchronize(f->stream));
    if (f->d_scratch) VX_HIP(f, hipFree(f->d_scratch));
    f->d_scratch = nullptr; f->scratch_cap = 0;
    VX_HIP(f, hipMalloc((void**)&f->d_scratch, need + need / 4));
    f->scratch_cap = need + need / 4;
  }
  return VXBA_OK;
}
}  // namespace

// LidarFactor::push_voxel for sparse incidence (the top level of the hierarchical BA pushes voxels seen from a handful of ~100
// submap poses, loop_refine.hpp:358-405): the caller lists only the observed (voxel, frame) entries.  The planes are filled on the
// device, so neither a dense n x W x 10 host array (0.8 GB at n = 100k, W = 99) nor its PCIe transfer exists -- nnz x 88 bytes go up.
int vxba_push_voxels_csr(vxba_factor* f, int n, const int64_t* row_ptr, const int32_t* frame_idx, const double* clusters, const double* fix, const double* coe,
                         const double* eig_val, const double* eig_vec, const double* merged) {
  VX_LOCK(f);
  if (!f || n < 0 || (n > 0 && (!row_ptr || !fix || !coe))) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: null input");
  if (n == 0) return VXBA_OK;
  if (row_ptr[0] != 0) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: row_ptr[0] must be 0");
  const int64_t nnz = row_ptr[n];
  for (int a = 0; a < n; a++) {
    if (row_ptr[a + 1] < row_ptr[a] || row_ptr[a + 1] - row_ptr[a] > f->W) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: row_ptr must be non-decreasing with at most win_size entries per voxel");
    if (!(coe[a] >= 0.0)) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: coe must be >= 0");
  }
  if (nnz > 0 && (!frame_idx || !clusters)) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: null entries");
  hipSetDevice(f->device);
  int rc = ensure_capacity(f, f->V + n);
  if (rc) return rc;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t b_ptr = up((size_t)(n + 1) * sizeof(int64_t)), b_fr = up((size_t)std::max<int64_t>(1, nnz) * sizeof(int32_t)), b_cl = up((size_t)std::max<int64_t>(1, nnz) * 10 * sizeof(double));
  rc = ensure_scratch(f, b_ptr + b_fr + b_cl + 256);
  if (rc) return rc;
  char* q = f->d_scratch;
  long long* d_ptr = (long long*)q; q += b_ptr;
  int* d_fr = (int*)q; q += b_fr;
  double* d_cl = (double*)q; q += b_cl;
  int* d_bad = (int*)q;
  VX_HIP(f, hipMemsetAsync(d_bad, 0, sizeof(int), f->stream));
  VX_HIP(f, hipMemcpyAsync(d_ptr, row_ptr, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream));
  if (nnz > 0) {
    VX_HIP(f, hipMemcpyAsync(d_fr, frame_idx, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, f->stream));
    VX_HIP(f, hipMemcpyAsync(d_cl, clusters, (size_t)nnz * 10 * sizeof(double), hipMemcpyHostToDevice, f->stream));
  }
  if (is_wide(f)) {   // straight into the compressed-row store: nothing dense exists anywhere
    const char* emsg = nullptr;
    if (vxw::store_reserve(f->wstore, f->V + n, f->wstore.nnz + nnz, f->V, f->stream, &emsg) != 0) return fail(f, VXBA_ERR_HIP, emsg ? emsg : "wide store: allocation failed");
    vxw::store_append_csr(f->wstore, f->V, n, d_ptr, d_fr, d_cl, nnz, f->W, d_bad, f->stream);
  } else {
    vxk::launch_scatter_clusters_csr(d_ptr, d_fr, d_cl, view(f), f->V, n, d_bad, f->stream);
    vxk::launch_build_clb(view(f), f->V, n, f->stream);
    clusters_written(f, f->V);
  }
  int bad = 0;
  VX_HIP(f, hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  if (bad) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: frame indices must be strictly increasing inside a voxel and below win_size (nothing was appended)");
  rc = append_meta(f, f->V, n, fix, coe, eig_val, eig_vec, merged);
  if (rc) return rc;
  if (is_wide(f)) f->wstore.nnz += nnz;
  f->V += n;
  f->wide_dirty = true;
  release_push_buffers(f);
  return VXBA_OK;
}

int vxba_push_points(vxba_factor* f, int n_voxels, int64_t n_points, const double* xyz_body, const int64_t* cell_ptr, const double* fix,
                     const double* coe) {
  VX_LOCK(f);
  if (!f || n_voxels < 0 || n_points < 0 || !cell_ptr || (n_points > 0 && !xyz_body)) return fail(f, VXBA_ERR_ARG, "push_points: null input");
  if (n_voxels == 0) return VXBA_OK;
  const int64_t ncells = (int64_t)n_voxels * f->W;
  if (cell_ptr[0] != 0 || cell_ptr[ncells] != n_points) return fail(f, VXBA_ERR_ARG, "push_points: cell_ptr must span [0, n_points]");
  if (coe)
    for (int a = 0; a < n_voxels; a++)
      if (!(coe[a] >= 0.0)) return fail(f, VXBA_ERR_ARG, "push_points: coe must be >= 0");
  hipSetDevice(f->device);
  int rc = ensure_capacity(f, f->V + n_voxels);
  if (rc) return rc;
  const size_t b_xyz = ((std::max<size_t>(1, (size_t)n_points * 3) * sizeof(double)) + 255) / 256 * 256;
  rc = ensure_scratch(f, b_xyz + (size_t)(ncells + 1) * sizeof(int64_t));
  if (rc) return rc;
  double* d_xyz = (double*)f->d_scratch;
  int64_t* d_ptr = (int64_t*)(f->d_scratch + b_xyz);
  hipError_t e = hipSuccess;
  auto cleanup = [&]() {};
  if (n_points) e = hipMemcpyAsync(d_xyz, xyz_body, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice, f->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_ptr, cell_ptr, (size_t)(ncells + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream);
  if (e != hipSuccess) { cleanup(); f->err = std::string("push_points H2D: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  const long long nnz0 = f->wstore.nnz;
  if (is_wide(f)) {
    // cell sums (the caller's cell table is dense, frame-major: cell = frame * n_voxels + voxel) into a staging block of the same shape,
    // from there into the compressed-row store
    rc = ensure_staging(f, (size_t)n_voxels * f->W * 10);
    if (rc) return rc;
    {
      ScopedKernelTimer t(f, 3);
      vxk::launch_k1_build_aos(d_xyz, d_ptr, ncells, f->staging, f->stream);
    }
    rc = wide_append_dense(f, n_voxels, f->staging, 1);
    if (rc) { f->wstore.nnz = nnz0; return rc; }
  } else {
    {
      ScopedKernelTimer t(f, 3);
      vxk::launch_k1_build(d_xyz, d_ptr, n_voxels, f->W, view(f), f->V, f->stream);
    }
    vxk::launch_build_clb(view(f), f->V, n_voxels, f->stream);
    clusters_written(f, f->V);
  }
  e = hipStreamSynchronize(f->stream);
  if (e == hipSuccess) e = hipGetLastError();
  cleanup();
  if (e != hipSuccess) { f->wstore.nnz = nnz0; f->err = std::string("K1: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  rc = append_meta(f, f->V, n_voxels, fix, coe, nullptr, nullptr, nullptr);
  if (rc) { f->wstore.nnz = nnz0; return rc; }
  f->V += n_voxels;
  f->wide_dirty = true;
  release_push_buffers(f);
  return VXBA_OK;
}

int vxba_read_clusters(vxba_factor* f, int head, int end, double* clusters) {
  VX_LOCK(f);
  if (!f || !clusters) return VXBA_ERR_ARG;
  int rc = check_range(f, head, end);
  if (rc) return rc;
  const int n = end - head;
  if (n == 0) return VXBA_OK;
  hipSetDevice(f->device);
  const size_t len = (size_t)n * f->W * 10;
  rc = ensure_staging(f, len);
  if (rc) return rc;
  if (is_wide(f)) vxw::store_expand(f->wstore, head, n, f->W, f->staging, f->stream);
  else vxk::launch_gather_clusters(view(f), head, n, f->staging, f->stream);
  VX_HIP(f, hipMemcpyAsync(clusters, f->staging, len * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  return VXBA_OK;
}

int vxba_acc_evaluate2(vxba_factor* f, const double* Rp, int head, int end, double* Hess, double* JacT, double* residual) {
  VX_LOCK(f);
  if (!f || !Rp || !Hess || !JacT || !residual) return fail(f, VXBA_ERR_ARG, "acc_evaluate2: null argument");
  int rc = check_range(f, head, end);
  if (rc) return rc;
  hipSetDevice(f->device);
  rc = sweep_hess_host(f, Rp, head, end);
  if (rc) return rc;
  const int n = 6 * f->W;
  std::memcpy(Hess, f->h_packed, sizeof(double) * n * n);
  std::memcpy(JacT, f->h_packed + (size_t)n * n, sizeof(double) * n);
  *residual = f->h_packed[(size_t)n * n + n];
  return VXBA_OK;
}

int vxba_evaluate_only_residual(vxba_factor* f, const double* Rp, int head, int end, double* residual) {
  VX_LOCK(f);
  if (!f || !Rp || !residual) return fail(f, VXBA_ERR_ARG, "evaluate_only_residual: null argument");
  int rc = check_range(f, head, end);
  if (rc) return rc;
  hipSetDevice(f->device);
  return sweep_residual_host(f, Rp, head, end, residual);
}

int vxba_acc_evaluate2_device(vxba_factor* f, const double* Rp, int head, int end, double* d_out) {
  VX_LOCK(f);
  if (!f || !Rp || !d_out) return fail(f, VXBA_ERR_ARG, "acc_evaluate2_device: null argument");
  int rc = check_range(f, head, end);
  if (rc) return rc;
  hipSetDevice(f->device);
  return sweep_hess_device(f, Rp, nullptr, nullptr, nullptr, head, end, d_out);
}

int vxba_evaluate_only_residual_device(vxba_factor* f, const double* Rp, int head, int end, double* d_out) {
  VX_LOCK(f);
  if (!f || !Rp || !d_out) return fail(f, VXBA_ERR_ARG, "evaluate_only_residual_device: null argument");
  int rc = check_range(f, head, end);
  if (rc) return rc;
  hipSetDevice(f->device);
  return sweep_residual_device(f, Rp, nullptr, 0, head, end, d_out);
}

int vxba_read_cache(vxba_factor* f, int head, int end, double* eig_val, double* eig_vec, double* merged) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  int rc = check_range(f, head, end);
  if (rc) return rc;
  const int n = end - head;
  if (n == 0) return VXBA_OK;
  hipSetDevice(f->device);
  rc = ensure_staging(f, (size_t)n * 22);
  if (rc) return rc;
  const FactorView fv = view(f);
  double* s = f->staging;
  if (eig_val) {
    vxk::launch_gather_rows(fv.eigval, f->VS, head, n, 3, s, f->stream);
    VX_HIP(f, hipMemcpyAsync(eig_val, s, sizeof(double) * n * 3, hipMemcpyDeviceToHost, f->stream));
  }
  s += (size_t)n * 3;
  if (eig_vec) {
    vxk::launch_gather_rows(fv.eigvec, f->VS, head, n, 9, s, f->stream);
    VX_HIP(f, hipMemcpyAsync(eig_vec, s, sizeof(double) * n * 9, hipMemcpyDeviceToHost, f->stream));
  }
  s += (size_t)n * 9;
  if (merged) {
    vxk::launch_gather_rows(fv.merged, f->VS, head, n, 10, s, f->stream);
    VX_HIP(f, hipMemcpyAsync(merged, s, sizeof(double) * n * 10, hipMemcpyDeviceToHost, f->stream));
  }
  VX_HIP(f, hipStreamSynchronize(f->stream));
  return VXBA_OK;
}

int vxba_snapshot_cache(vxba_factor* f) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "snapshot_cache on an empty factor");
  hipSetDevice(f->device);
  if (f->snapshot_vs != f->VS) {
    if (f->snapshot) { VX_HIP(f, hipStreamSynchronize(f->stream)); VX_HIP(f, hipFree(f->snapshot)); f->snapshot = nullptr; }
    VX_HIP(f, hipMalloc((void**)&f->snapshot, (size_t)N_CACHE_PLANES * f->VS * sizeof(double)));
    f->snapshot_vs = f->VS;
  }
  VX_HIP(f, hipMemcpyAsync(f->snapshot, view(f).eigval, (size_t)N_CACHE_PLANES * f->VS * sizeof(double), hipMemcpyDeviceToDevice, f->stream));
  f->snapshot_v = f->V;
  return VXBA_OK;
}

int vxba_restore_cache(vxba_factor* f) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  if (!f->snapshot || f->snapshot_v != f->V || f->snapshot_vs != f->VS) return fail(f, VXBA_ERR_STATE, "no matching cache snapshot");
  hipSetDevice(f->device);
  VX_HIP(f, hipMemcpyAsync(view(f).eigval, f->snapshot, (size_t)N_CACHE_PLANES * f->VS * sizeof(double), hipMemcpyDeviceToDevice, f->stream));
  return VXBA_OK;
}

int vxba_plane_fit_judge(int device, int64_t n, const double* clusters, int min_point, double min_eigen_value, double eigen_ratio_thre,
                         double factor_ratio_max, double* eig_val, double* eig_vec, uint8_t* flags) {
  if (n < 0 || (n > 0 && (!clusters || !eig_val || !eig_vec))) return VXBA_ERR_ARG;
  if (n == 0) return VXBA_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  using vxs::Lease;
  Lease lease(device, Lease::padded((size_t)n * 10 * 8) + Lease::padded((size_t)n * 3 * 8) + Lease::padded((size_t)n * 9 * 8) + Lease::padded((size_t)n));
  if (!lease.ok()) return VXBA_ERR_HIP;
  double* d_c = lease.take<double>((size_t)n * 10 * 8);
  double* d_l = lease.take<double>((size_t)n * 3 * 8);
  double* d_u = lease.take<double>((size_t)n * 9 * 8);
  unsigned char* d_f = flags ? lease.take<unsigned char>((size_t)n) : nullptr;
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipMemcpy(d_c, clusters, (size_t)n * 10 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vxk::PlaneCriteria pc{min_point, min_eigen_value, eigen_ratio_thre, factor_ratio_max};
    vxk::launch_k4_plane_fit(d_c, n, d_l, d_u, flags ? &pc : nullptr, d_f, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(eig_val, d_l, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(eig_vec, d_u, (size_t)n * 9 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && flags) e = hipMemcpy(flags, d_f, (size_t)n, hipMemcpyDeviceToHost);
  return e == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}

int vxba_plane_fit(int device, int64_t n, const double* clusters, double* eig_val, double* eig_vec) {
  return vxba_plane_fit_judge(device, n, clusters, 0, 0.0, 0.0, 0.0, eig_val, eig_vec, nullptr);
}

int vxba_build_clusters(int device, int64_t n_cells, int64_t n_points, const double* xyz, const int64_t* cell_ptr, double* clusters) {
  if (n_cells < 0 || n_points < 0 || !cell_ptr || !clusters || (n_points > 0 && !xyz)) return VXBA_ERR_ARG;
  if (n_cells == 0) return VXBA_OK;
  if (cell_ptr[0] != 0 || cell_ptr[n_cells] != n_points) return VXBA_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  using vxs::Lease;
  Lease lease(device, Lease::padded((size_t)n_points * 3 * 8) + Lease::padded((size_t)(n_cells + 1) * 8) + Lease::padded((size_t)n_cells * 10 * 8));
  if (!lease.ok()) return VXBA_ERR_HIP;
  double* d_xyz = lease.take<double>((size_t)n_points * 3 * 8);
  int64_t* d_ptr = lease.take<int64_t>((size_t)(n_cells + 1) * 8);
  double* d_cl = lease.take<double>((size_t)n_cells * 10 * 8);
  hipError_t e = hipSuccess;
  if (e == hipSuccess && n_points) e = hipMemcpy(d_xyz, xyz, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ptr, cell_ptr, (size_t)(n_cells + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vxk::launch_k1_build_aos(d_xyz, d_ptr, n_cells, d_cl, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(clusters, d_cl, (size_t)n_cells * 10 * sizeof(double), hipMemcpyDeviceToHost);
  return e == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}

// Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:367-442).  The whole loop is enqueued on the stream without a host
// round trip: the LM state (poses, damping, accept/reject flags) lives in device memory (vxk::LMState), the solve and
// the accept/reject step are single-workgroup kernels, and the sweeps gate themselves on the state's flags exactly
// where the reference branches (is_calc_hess, the early break).  One D2H copy + one sync at the end.
static int damping_iter_impl(vxba_factor* f, double* Rp, int max_iter, double* hess_out, double* resis_out, double* trace_out, int* n_trace,
                             int* is_converge) {
  VX_LOCK(f);
  if (!f || !Rp || max_iter < 0 || max_iter > vxk::LM_MAX_ITER) return fail(f, VXBA_ERR_ARG, "damping_iter: bad argument (max_iter <= 64)");
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "damping_iter on an empty factor");
  hipSetDevice(f->device);
  if (is_wide(f)) {
    // wide window (voxel_map.hpp:367-442 unchanged in structure): sweeps on the GPU; the damped (6W)-dimensional step by a dense
    // device Cholesky (only dxi, q1 and residual1 come back: ~5 KB) or, if that is unavailable / the system is not positive
    // definite, by the host's pivoted LDL^T on the downloaded system; accept/reject on the host.
    const int W = f->W, n = 6 * W;
    if (!f->wide_solver && !f->wide_solver_tried && f->opt[VXBA_OPT_WIDE_DEVICE_SOLVE] != 0) {
      // the library's own blocked Cholesky (vxba_wide.hip): a few device buffers, no third-party library to load
      f->wide_solver = vxw::wide_solver_create(n, f->stream);
      f->wide_solver_tried = true;
    }
    const bool use_device_solver = f->wide_solver && f->opt[VXBA_OPT_WIDE_DEVICE_SOLVE] != 0;
    double u = 0.01, v = 2;
    std::vector<double> x(Rp, Rp + 12 * W), x_temp(x), dxi(n), Hh, Jh;
    vxh::LMWorkspace ws;
    double residual1 = 0, residual2 = 0, q1 = 0;
    bool is_calc_hess = true, converge = true, host_copy_valid = false;
    int nt = 0;
    for (int i = 0; i < max_iter; i++) {
      const bool recomputed = is_calc_hess;
      if (is_calc_hess) {
        int rc = sweep_hess_device(f, x.data(), nullptr, nullptr, nullptr, 0, f->V, f->d_packed);
        if (rc) return rc;
        host_copy_valid = false;
      }
      bool on_device = false;
      if (use_device_solver) {
        double r1 = 0;
        on_device = vxw::wide_solver_step(f->wide_solver, f->d_packed, u, f->stream, dxi.data(), &q1, &r1, f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] != 0) == 0;
        if (!on_device) f->fused_fallbacks++;     // counted like the narrow loop's fallback (VXBA_STAT_FUSED_FALLBACKS)
        if (on_device) {
          if (is_calc_hess) residual1 = r1;
          for (int j = 0; j < W; j++) {
            vxh::right_multiply_exp(&x[12 * j], &dxi[6 * j], &x_temp[12 * j]);
            for (int k = 0; k < 3; k++) x_temp[12 * j + 9 + k] = x[12 * j + 9 + k] + dxi[6 * j + 3 + k];
          }
        }
      }
      if (!on_device) {
        if (!host_copy_valid) {
          VX_HIP(f, hipMemcpyAsync(f->h_packed, f->d_packed, vxba_packed_len(f) * sizeof(double), hipMemcpyDeviceToHost, f->stream));
          VX_HIP(f, hipStreamSynchronize(f->stream));
          Hh.assign(f->h_packed, f->h_packed + (size_t)n * n);
          Jh.assign(f->h_packed + (size_t)n * n, f->h_packed + (size_t)n * n + n);
          if (is_calc_hess) residual1 = f->h_packed[(size_t)n * n + n];
          host_copy_valid = true;
        }
        q1 = vxh::lm_damped_step(W, Hh.data(), Jh.data(), u, x.data(), x_temp.data(), ws);
      }
      if (i == 0 && resis_out) resis_out[0] = residual1;
      int rc = sweep_residual_host(f, x_temp.data(), 0, f->V, &residual2);
      if (rc) return rc;
      const double q = residual1 - residual2;
      const double u_used = u, v_used = v;
      const bool accepted = vxh::lm_update_damping(residual1, residual2, q1, u, v);
      if (accepted) { x = x_temp; is_calc_hess = true; }
      else { is_calc_hess = false; converge = false; }
      if (trace_out) {
        double* o = trace_out + (size_t)VXBA_TRACE_COLS * nt;
        o[0] = residual1; o[1] = residual2; o[2] = u_used; o[3] = v_used; o[4] = q; o[5] = q1; o[6] = accepted; o[7] = recomputed;
      }
      nt++;
      if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
    }
    if (hess_out) {   // *hess = the last Hessian that was computed, before the gauge fix (voxel_map.hpp:391)
      VX_HIP(f, hipMemcpyAsync(f->h_packed, f->d_packed, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
      VX_HIP(f, hipStreamSynchronize(f->stream));
      std::memcpy(hess_out, f->h_packed, sizeof(double) * n * n);
    }
    if (resis_out) resis_out[1] = residual2;
    if (n_trace) *n_trace = nt;
    if (is_converge) *is_converge = converge ? 1 : 0;
    std::memcpy(Rp, x.data(), sizeof(double) * 12 * W);
    return VXBA_OK;
  }
  const int W = f->W, n = 6 * W;
  PoseArg x0;
  fill_poses(f, Rp, x0);
  vxk::launch_lm_init(f->d_lm, x0, W, 0, f->stream);
  // The accept/reject step of iteration i is taken in the prologue of iteration i+1's Hessian sweep (every workgroup
  // recomputes it from ctl[c]; workgroup 0 persists it into ctl[c^1]); a stand-alone decision kernel closes the loop.
  int c = 0;
  vxk::LMPending pend;
  std::memset(&pend, 0, sizeof pend);
  const bool spec = spec_collective(f);
  int spec_nparts = 0;
  for (int i = 0; spec && i < max_iter; i++) {
    int rc = spec_hess_phase(f, Rp, &c, i == 0, i > 0, false, nullptr, spec_nparts);
    if (rc) return rc;
    const unsigned seq = fused_solve(f) ? ++f->lm_seq : 0u;
    if (!seq) vxk::launch_lm_solve(f->d_lm, c, W, f->stream);
    rc = sweep_residual_device(f, nullptr, f->d_lm, c, 0, f->V, nullptr, &spec_nparts, seq);
    if (rc) return rc;
  }
  if (spec && max_iter > 0) { int rc = spec_final_decision(f, Rp, &c, spec_nparts); if (rc) return rc; }
  for (int i = 0; !spec && i < max_iter; i++) {
    int rc = sweep_hess_device(f, Rp, f->d_lm, &c, &pend, 0, f->V, f->d_packed);
    if (rc) return rc;
    // damped solve + residual sweep at the trial state: one launch (the solve is workgroup 0 of the sweep) unless
    // VXBA_FUSED_SOLVE=0; without a collective the sweep's wave partials are summed by whoever takes the decision
    const unsigned seq = fused_solve(f) ? ++f->lm_seq : 0u;
    if (!seq) vxk::launch_lm_solve(f->d_lm, c, W, f->stream);
    int nparts = 0;
    rc = sweep_residual_device(f, nullptr, f->d_lm, c, 0, f->V, has_collective(f) ? f->d_scalar : nullptr, &nparts, seq);
    if (rc) return rc;
    pend.pending = 1; pend.restart = 0;
    pend.d_scalar = has_collective(f) ? f->d_scalar : nullptr;
    pend.partial = f->d_partial2; pend.nparts = nparts;
  }
  if (pend.pending) { vxk::launch_lm_update(f->d_lm, c, pend, x0, W, f->stream); c ^= 1; }
  VX_HIP(f, hipGetLastError());
  VX_HIP(f, hipMemcpyAsync(f->h_lm, f->d_lm, sizeof(vxk::LMState), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  if (f->h_lm->error) {
    f->solve_timed_out = true;
    return fail(f, VXBA_ERR_STATE, "damping_iter: a residual-sweep workgroup timed out waiting for the in-launch solve");
  }
  const vxk::LMCtl& st = f->h_lm->ctl[c];
  std::memcpy(Rp, st.x, sizeof(double) * 12 * W);
  if (hess_out) std::memcpy(hess_out, f->h_lm->hess_out, sizeof(double) * n * n);
  if (resis_out) { resis_out[0] = st.resis[0]; resis_out[1] = st.resis[1]; }
  const int nt = std::min(st.iter, vxk::LM_MAX_ITER);
  if (trace_out) std::memcpy(trace_out, f->h_lm->trace, sizeof(double) * VXBA_TRACE_COLS * nt);
  if (n_trace) *n_trace = nt;
  if (is_converge) *is_converge = st.converge;
  return VXBA_OK;
}

// Host-only LM shell over caller-supplied sweeps (same control flow as voxel_map.hpp:367-442).
int vxba_damping_iter_generic(int W, double* Rp, int max_iter, vxba_hess_fn hess_fn, vxba_resid_fn resid_fn, void* ctx, double* hess_out,
                              double* resis_out, double* trace_out, int* n_trace, int* is_converge) {
  if (W < 1 || !Rp || max_iter < 0 || !hess_fn || !resid_fn) return VXBA_ERR_ARG;
  const int n = 6 * W;
  double u = 0.01, v = 2;
  std::vector<double> packed((size_t)n * n + n + 1), Hess((size_t)n * n), JacT(n), x(Rp, Rp + 12 * W), x_temp(x);
  vxh::LMWorkspace ws;
  double residual1 = 0, residual2 = 0;
  bool is_calc_hess = true, converge = true;
  int nt = 0;
  for (int i = 0; i < max_iter; i++) {
    const bool recomputed = is_calc_hess;
    if (is_calc_hess) {
      if (hess_fn(ctx, x.data(), packed.data()) != 0) return VXBA_ERR_STATE;
      std::memcpy(Hess.data(), packed.data(), sizeof(double) * n * n);
      std::memcpy(JacT.data(), packed.data() + (size_t)n * n, sizeof(double) * n);
      residual1 = packed[(size_t)n * n + n];
      if (hess_out) std::memcpy(hess_out, Hess.data(), sizeof(double) * n * n);  // *hess = Hess, before the gauge fix
    }
    if (i == 0 && resis_out) resis_out[0] = residual1;
    const double q1 = vxh::lm_damped_step(W, Hess.data(), JacT.data(), u, x.data(), x_temp.data(), ws);
    if (resid_fn(ctx, x_temp.data(), &residual2) != 0) return VXBA_ERR_STATE;
    const double q = residual1 - residual2;
    const double u_used = u, v_used = v;
    const bool accepted = vxh::lm_update_damping(residual1, residual2, q1, u, v);
    if (accepted) { x = x_temp; is_calc_hess = true; }
    else { is_calc_hess = false; converge = false; }
    if (trace_out) {
      double* o = trace_out + (size_t)VXBA_TRACE_COLS * nt;
      o[0] = residual1; o[1] = residual2; o[2] = u_used; o[3] = v_used; o[4] = q; o[5] = q1; o[6] = accepted; o[7] = recomputed;
    }
    nt++;
    if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
  }
  if (resis_out) resis_out[1] = residual2;
  if (n_trace) *n_trace = nt;
  if (is_converge) *is_converge = converge ? 1 : 0;
  std::memcpy(Rp, x.data(), sizeof(double) * 12 * W);
  return VXBA_OK;
}

static int lm_steps_impl(vxba_factor* f, const double* Rp_init, int n_steps, int steps_per_solve, double* Rp_out, double* last_resis,
                  int64_t* stats_out) {
  VX_LOCK(f);
  if (!f || !Rp_init || n_steps < 0 || steps_per_solve < 1) return fail(f, VXBA_ERR_ARG, "lm_steps: bad argument");
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "lm_steps on an empty factor");
  VX_NARROW_ONLY(f, "lm_steps");
  hipSetDevice(f->device);
  const int W = f->W;
  PoseArg x0;
  fill_poses(f, Rp_init, x0);
  if (!f->snapshot || f->snapshot_v != f->V || f->snapshot_vs != f->VS) return fail(f, VXBA_ERR_STATE, "lm_steps needs vxba_snapshot_cache first");
  vxk::launch_lm_init(f->d_lm, x0, W, 1, f->stream);
  int c = 0;
  vxk::LMPending pend;
  std::memset(&pend, 0, sizeof pend);
  const bool spec = spec_collective(f);
  int spec_nparts = 0;
  bool prev_last = false;
  for (int s = 0; spec && s < n_steps; s++) {
    const bool first = (s % steps_per_solve) == 0;
    const bool last = ((s + 1) % steps_per_solve) == 0 && s + 1 < n_steps;
    int rc = spec_hess_phase(f, Rp_init, &c, first, s > 0, prev_last, first ? f->snapshot : nullptr, spec_nparts);
    if (rc) return rc;
    const unsigned seq = fused_solve(f) ? ++f->lm_seq : 0u;
    if (!seq) vxk::launch_lm_solve(f->d_lm, c, W, f->stream);
    rc = sweep_residual_device(f, nullptr, f->d_lm, c, 0, f->V, nullptr, &spec_nparts, seq);
    if (rc) return rc;
    prev_last = last;
  }
  if (spec && n_steps > 0) { int rc = spec_final_decision(f, Rp_init, &c, spec_nparts); if (rc) return rc; }
  for (int s = 0; !spec && s < n_steps; s++) {
    // a new window every steps_per_solve steps: its first Hessian sweep reads the SNAPSHOT cache directly (the re-seeded
    // cache of a new window -- no copy) and its prologue resets poses and damping (pend.restart of the previous step);
    // the residual sweeps keep writing the live cache
    const bool first = (s % steps_per_solve) == 0;
    const bool last = ((s + 1) % steps_per_solve) == 0 && s + 1 < n_steps;
    int rc = sweep_hess_device(f, Rp_init, f->d_lm, &c, &pend, 0, f->V, f->d_packed, first ? f->snapshot : nullptr);
    if (rc) return rc;
    const unsigned seq = fused_solve(f) ? ++f->lm_seq : 0u;
    if (!seq) vxk::launch_lm_solve(f->d_lm, c, W, f->stream);
    int nparts = 0;
    rc = sweep_residual_device(f, nullptr, f->d_lm, c, 0, f->V, has_collective(f) ? f->d_scalar : nullptr, &nparts, seq);
    if (rc) return rc;
    pend.pending = 1; pend.restart = last ? 1 : 0;
    pend.d_scalar = has_collective(f) ? f->d_scalar : nullptr;
    pend.partial = f->d_partial2; pend.nparts = nparts;
  }
  if (pend.pending) { vxk::launch_lm_update(f->d_lm, c, pend, x0, W, f->stream); c ^= 1; }
  VX_HIP(f, hipGetLastError());
  VX_HIP(f, hipMemcpyAsync(f->h_lm, f->d_lm, sizeof(vxk::LMState), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  if (f->h_lm->error) return fail(f, VXBA_ERR_STATE, "lm_steps: a residual-sweep workgroup timed out waiting for the in-launch solve");
  const vxk::LMCtl& st = f->h_lm->ctl[c];
  if (Rp_out) std::memcpy(Rp_out, st.x, sizeof(double) * 12 * W);
  if (last_resis) { last_resis[0] = st.residual1; last_resis[1] = st.residual2; }
  if (stats_out) { stats_out[0] = st.iter; stats_out[1] = st.n_accept; stats_out[2] = st.n_reject; }
  return VXBA_OK;
}

// OctreeGBA::cut_voxel + recut on the GPU (vxba_voxelize.hip); the accepted voxels go straight from the staging arrays into
// the factor's planes -- nothing returns to the host except their count (and the ids, if asked for).
static int voxelize_push_impl(vxba_factor* f, int64_t n_points, const double* xyz_local, bool xyz_on_device, const int64_t* frame_ptr, const double* Rp,
                              const vxba_voxelize_params* params, int64_t* n_pushed, uint64_t* node_ids, int64_t ids_capacity) {
  VX_LOCK(f);
  if (!f || n_points < 0 || !frame_ptr || !Rp || !params || !n_pushed || (n_points > 0 && !xyz_local))
    return fail(f, VXBA_ERR_ARG, "voxelize_push: null argument");
  if (params->max_layer < 0 || params->max_layer > 3 || !(params->voxel_size > 0)) return fail(f, VXBA_ERR_ARG, "voxelize_push: max_layer in 0..3, voxel_size > 0");
  if (frame_ptr[0] != 0 || frame_ptr[f->W] != n_points) return fail(f, VXBA_ERR_ARG, "voxelize_push: frame_ptr must span [0, n_points]");
  if (n_points > 0xffffffffll) return fail(f, VXBA_ERR_UNSUPPORTED, "voxelize_push: more than 2^32 points");
  *n_pushed = 0;
  if (n_points == 0) return VXBA_OK;
  hipSetDevice(f->device);
  const int W = f->W;
  int floor_pts = params->min_points;
  for (int k = 0; k <= params->max_layer; k++)
    if (params->min_points_layer[k] > 0) floor_pts = std::min(floor_pts, params->min_points_layer[k]);
  const int64_t cap = n_points / (std::max(floor_pts, 0) + 1) + 1;   // a factor owns > min_points points, and no point twice
  // one grow-only scratch allocation per factor, carved up here (a hipMalloc / hipFree pair per buffer and call cost more than the sorts)
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  // wide windows take the accepted voxels as compressed rows (a point belongs to at most one factor voxel: n_points entries suffice)
  const bool csr = is_wide(f);
  const size_t b_xyz = xyz_on_device ? 0 : up((size_t)n_points * 3 * sizeof(double)), b_fp = up((size_t)(W + 1) * sizeof(long long)),
               b_cl = csr ? up((size_t)n_points * 10 * sizeof(double)) : up((size_t)cap * W * 10 * sizeof(double)),
               b_rp = csr ? up((size_t)(cap + 1) * sizeof(long long)) : 0, b_ef = csr ? up((size_t)n_points * sizeof(int)) : 0, b_ev = up((size_t)cap * 3 * sizeof(double)), b_evec = up((size_t)cap * 9 * sizeof(double)),
               b_m = up((size_t)cap * 10 * sizeof(double)), b_id = up((size_t)cap * sizeof(unsigned long long)), b_fix = up((size_t)cap * 10 * sizeof(double)),
               b_coe = up((size_t)cap * sizeof(double));
  const size_t need = b_xyz + b_fp + b_cl + b_rp + b_ef + b_ev + b_evec + b_m + b_id + b_fix + b_coe + 256;
  {
    int rcs = ensure_scratch(f, need);
    if (rcs) return rcs;
  }
  char* q = f->d_scratch;
  auto carve = [&](size_t bytes) { char* r = q; q += bytes; return r; };
  double* d_xyz_own = (double*)carve(b_xyz);
  long long* d_fp = (long long*)carve(b_fp);
  double* d_cl = (double*)carve(b_cl);
  long long* d_rp = (long long*)carve(b_rp); int* d_ef = (int*)carve(b_ef);
  double* d_ev = (double*)carve(b_ev); double* d_evec = (double*)carve(b_evec); double* d_m = (double*)carve(b_m);
  unsigned long long* d_id = (unsigned long long*)carve(b_id);
  double* d_fix = (double*)carve(b_fix); double* d_coe = (double*)carve(b_coe);
  int* d_bad = (int*)carve(256);
  const double* d_xyz = xyz_on_device ? xyz_local : d_xyz_own;
  if (!xyz_on_device) VX_HIP(f, hipMemcpyAsync(d_xyz_own, xyz_local, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice, f->stream));
  VX_HIP(f, hipMemcpyAsync(d_fp, frame_ptr, (size_t)(W + 1) * sizeof(long long), hipMemcpyHostToDevice, f->stream));
  int rcp = upload_poses(f, Rp);
  if (rcp) return rcp;
  vxv::VoxelizeParams vp;
  vp.voxel_size = params->voxel_size; vp.max_layer = params->max_layer; vp.min_points = params->min_points;
  vp.min_eigen_value = params->min_eigen_value; vp.factor_ratio_max = params->factor_ratio_max;
  for (int k = 0; k < 4; k++) { vp.eigen_ratio[k] = params->eigen_ratio[k]; vp.min_points_layer[k] = params->min_points_layer[k]; }
  vp.min_frames = params->min_frames;
  vxv::VoxelizeOutput out{cap, d_cl, d_ev, d_evec, d_m, d_id};
  if (csr) { out.d_row_ptr = d_rp; out.d_eframe = d_ef; out.ecap = n_points; }
  const char* emsg = nullptr;
  const long long n = vxv::voxelize(W, n_points, d_xyz, d_fp, f->d_poses, vp, f->stream, &out, &emsg);
  if (n < 0) return fail(f, VXBA_ERR_STATE, emsg ? emsg : "voxelize failed");
  if (n > 0) {
    int rc = ensure_capacity(f, f->V + (int)n);
    if (rc) return rc;
    const FactorView fv = view(f);
    const int v0 = f->V;
    vxv::fill(d_fix, n * 10, 0.0, f->stream);
    vxv::fill(d_coe, n, 1.0, f->stream);
    if (csr) {
      const char* em2 = nullptr;
      if (vxw::store_reserve(f->wstore, f->V + (int)n, f->wstore.nnz + out.n_entries, f->V, f->stream, &em2) != 0) return fail(f, VXBA_ERR_HIP, em2 ? em2 : "wide store: allocation failed");
      VX_HIP(f, hipMemsetAsync(d_bad, 0, sizeof(int), f->stream));   // rows the voxelisation wrote itself: never set
      vxw::store_append_csr(f->wstore, f->V, (int)n, d_rp, d_ef, d_cl, out.n_entries, W, d_bad, f->stream);
      f->wstore.nnz += out.n_entries;
    } else {
      vxk::launch_scatter_clusters(d_cl, fv, v0, (int)n, f->stream);
      vxk::launch_build_clb(fv, v0, (int)n, f->stream);
      clusters_written(f, v0);
    }
    vxk::launch_scatter_rows(d_fix, fv.fix, f->VS, v0, (int)n, 10, f->stream);
    vxk::launch_scatter_rows(d_coe, fv.coe, f->VS, v0, (int)n, 1, f->stream);
    vxk::launch_scatter_rows(d_ev, fv.eigval, f->VS, v0, (int)n, 3, f->stream);
    vxk::launch_scatter_rows(d_evec, fv.eigvec, f->VS, v0, (int)n, 9, f->stream);
    vxk::launch_scatter_rows(d_m, fv.merged, f->VS, v0, (int)n, 10, f->stream);
    vxk::launch_seed_aux(fv, v0, v0 + (int)n, f->stream);
    if (node_ids && ids_capacity > 0)
      VX_HIP(f, hipMemcpyAsync(node_ids, d_id, (size_t)std::min<int64_t>(n, ids_capacity) * sizeof(uint64_t), hipMemcpyDeviceToHost, f->stream));
    VX_HIP(f, hipStreamSynchronize(f->stream));
    VX_HIP(f, hipGetLastError());
    f->V += (int)n;
    f->wide_dirty = true;
  }
  *n_pushed = n;
  release_push_buffers(f);
  return VXBA_OK;
}

int vxba_voxelize_push(vxba_factor* f, int64_t n_points, const double* xyz_local, const int64_t* frame_ptr, const double* Rp,
                       const vxba_voxelize_params* params, int64_t* n_pushed, uint64_t* node_ids, int64_t ids_capacity) {
  return voxelize_push_impl(f, n_points, xyz_local, false, frame_ptr, Rp, params, n_pushed, node_ids, ids_capacity);
}
int vxba_voxelize_push_device(vxba_factor* f, int64_t n_points, const double* d_xyz_local, const int64_t* frame_ptr, const double* Rp,
                              const vxba_voxelize_params* params, int64_t* n_pushed, uint64_t* node_ids, int64_t ids_capacity) {
  return voxelize_push_impl(f, n_points, d_xyz_local, true, frame_ptr, Rp, params, n_pushed, node_ids, ids_capacity);
}

int vxba_debug_mfma_probe(int device, const double* A16x4, const double* B4x16, double* D16x16) {
  if (!A16x4 || !B4x16 || !D16x16) return VXBA_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_NODEV;
  double* d = nullptr;
  if (hipMalloc((void**)&d, (64 + 64 + 256) * sizeof(double)) != hipSuccess) return VXBA_ERR_HIP;
  hipError_t e = hipMemcpy(d, A16x4, 64 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + 64, B4x16, 64 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) { vxk::launch_mfma_probe(d, d + 64, d + 128, nullptr); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpy(D16x16, d + 128, 256 * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(d);
  return e == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}

int vxba_debug_stamps(int clear, unsigned long long* out, size_t n) {
  if (clear) vxk::debug_clear_stamps();
  if (out && n) { (void)hipDeviceSynchronize(); vxk::debug_read_stamps(out, n); }
  return VXBA_OK;
}

// ---- links for vxba_map.hip (vxba_internal.h) ----
int vxba_internal_push_voxels_device(vxba_factor* f, int n, const double* d_clusters, const double* d_fix, const double* d_coe, const double* d_eigval,
                                     const double* d_eigvec, const double* d_merged) {
  VX_LOCK(f);
  if (!f || n < 0 || (n > 0 && (!d_clusters || !d_fix || !d_coe || !d_eigval || !d_eigvec || !d_merged))) return fail(f, VXBA_ERR_ARG, "push_voxels_device: null argument");
  if (n == 0) return VXBA_OK;
  hipSetDevice(f->device);
  int rc = ensure_capacity(f, f->V + n);
  if (rc) return rc;
  const FactorView fv = view(f);
  const int v0 = f->V;
  if (is_wide(f)) { rc = wide_append_dense(f, n, d_clusters); if (rc) return rc; }
  else {
    vxk::launch_scatter_clusters(d_clusters, fv, v0, n, f->stream);
    vxk::launch_build_clb(fv, v0, n, f->stream);
    clusters_written(f, v0);
  }
  vxk::launch_scatter_rows(d_fix, fv.fix, f->VS, v0, n, 10, f->stream);
  vxk::launch_scatter_rows(d_coe, fv.coe, f->VS, v0, n, 1, f->stream);
  vxk::launch_scatter_rows(d_eigval, fv.eigval, f->VS, v0, n, 3, f->stream);
  vxk::launch_scatter_rows(d_eigvec, fv.eigvec, f->VS, v0, n, 9, f->stream);
  vxk::launch_scatter_rows(d_merged, fv.merged, f->VS, v0, n, 10, f->stream);
  vxk::launch_seed_aux(fv, v0, v0 + n, f->stream);
  VX_HIP(f, hipStreamSynchronize(f->stream));
  VX_HIP(f, hipGetLastError());
  f->V += n;
  f->wide_dirty = true;
  return VXBA_OK;
}
int vxba_internal_factor_device(const vxba_factor* f) { return f ? f->device : -1; }
int vxba_internal_cache_view(vxba_factor* f, const double** eigval, const double** eigvec, const double** merged, int* VS, int* V) {
  VX_LOCK(f);
  if (!f || !eigval || !eigvec || !merged || !VS || !V) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  VX_HIP(f, hipStreamSynchronize(f->stream));
  const FactorView fv = view(f);
  *eigval = fv.eigval; *eigvec = fv.eigvec; *merged = fv.merged; *VS = f->VS; *V = f->V;
  return VXBA_OK;
}

// Test access to the structured LiDAR-inertial solve (host code, needs no GPU): A (m x m, full symmetric), b -> x.
int vxba_debug_band_schur(int m, const double* A, const double* b, int nframes, int lead_y, int tail_x, double* x) {
  if (!A || !b || !x || m != lead_y + 15 * nframes + tail_x) return VXBA_ERR_ARG;
  const vxh::LiIndexSets s = vxh::li_index_sets(nframes, lead_y, tail_x);
  vxh::BandSchurWork w;
  return vxh::band_schur_solve(m, A, m, nullptr, b, s.Y.data(), (int)s.Y.size(), s.bw, s.X.data(), (int)s.X.size(), s.xlo.data(), x, w) ? VXBA_OK : VXBA_ERR_STATE;
}

int vxba_set_option(vxba_factor* f, int option, int value) {
  if (!f) return VXBA_ERR_ARG;
  VX_LOCK(f);
  switch (option) {
    case VXBA_OPT_FUSED_SOLVE: case VXBA_OPT_SPEC_COLLECTIVE: case VXBA_OPT_WIDE_DEVICE_SOLVE: case VXBA_OPT_LI_DEVICE_LOOP:
    case VXBA_OPT_DEBUG_SOLVE_TIMEOUT: case VXBA_OPT_LI_STRUCTURED_SOLVE: case VXBA_OPT_LI_QUEUED_SWEEPS:
      if (value != 0 && value != 1) return fail(f, VXBA_ERR_ARG, "vxba_set_option: this option takes 0 or 1");
      break;
    case VXBA_OPT_K2_VOXELS_PER_BLOCK:
      if (value < 32 || value > 64) return fail(f, VXBA_ERR_ARG, "vxba_set_option: voxels per block must be in [32, 64]");
      break;
    default: return fail(f, VXBA_ERR_ARG, "vxba_set_option: unknown option");
  }
  f->opt[option] = value;
  return VXBA_OK;
}
int vxba_get_option(const vxba_factor* f, int option, int* value) {
  if (f && value && option == VXBA_STAT_FUSED_FALLBACKS) { *value = f->fused_fallbacks; return VXBA_OK; }
  if (f && value && option == VXBA_STAT_LI_LAST_CALL_US) { *value = (int)(f->li_last_call_us + 0.5); return VXBA_OK; }
  if (!f || !value || option < 0 || option >= VXBA_OPT_COUNT) return VXBA_ERR_ARG;
  *value = f->opt[option];
  return VXBA_OK;
}

int vxba_set_precision(vxba_factor* f, int mode) {
  VX_LOCK(f);
  if (!f || (mode != VXBA_PRECISION_F64 && mode != VXBA_PRECISION_MIXED && mode != VXBA_PRECISION_MIXED_F32_CLUSTERS))
    return fail(f, VXBA_ERR_ARG, "set_precision: mode must be VXBA_PRECISION_F64, VXBA_PRECISION_MIXED or VXBA_PRECISION_MIXED_F32_CLUSTERS");
  if (mode != VXBA_PRECISION_F64) VX_NARROW_ONLY(f, "mixed precision");
  f->precision = mode;
  return VXBA_OK;
}

int vxba_set_profiling(vxba_factor* f, int on) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  f->profiling = on;
  return VXBA_OK;
}

int vxba_get_kernel_times(vxba_factor* f, double ms_sum[4], int64_t calls[4], int reset) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  int rc = drain_events(f);
  if (rc) return rc;
  for (int k = 0; k < 4; k++) {
    if (ms_sum) ms_sum[k] = f->ms_sum[k];
    if (calls) calls[k] = f->calls[k];
    if (reset) { f->ms_sum[k] = 0; f->calls[k] = 0; }
  }
  return VXBA_OK;
}

int vxba_get_collective_time(vxba_factor* f, double* ms_sum, int64_t* calls, int reset) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  int rc = drain_events(f);
  if (rc) return rc;
  if (ms_sum) *ms_sum = f->ms_sum[4];
  if (calls) *calls = f->calls[4];
  if (reset) { f->ms_sum[4] = 0; f->calls[4] = 0; }
  return VXBA_OK;
}

int vxba_nnz(vxba_factor* f, int64_t* nnz) {
  VX_LOCK(f);
  if (!f || !nnz) return VXBA_ERR_ARG;
  *nnz = 0;
  if (f->V == 0) return VXBA_OK;
  hipSetDevice(f->device);
  if (is_wide(f)) {
    const char* emsg = nullptr;
    const long long c = vxw::store_count_observed(f->wstore, f->V, f->stream, &emsg);
    if (c < 0) return fail(f, VXBA_ERR_HIP, emsg ? emsg : "nnz: count failed");
    *nnz = c;
    return VXBA_OK;
  }
  VX_HIP(f, hipMemsetAsync(f->d_count, 0, sizeof(unsigned long long), f->stream));
  vxk::launch_count_nnz(view(f), f->V, f->d_count, f->stream);
  unsigned long long h = 0;
  VX_HIP(f, hipMemcpyAsync(&h, f->d_count, sizeof h, hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  *nnz = (int64_t)h;
  return VXBA_OK;
}

int vxba_device_bytes(const vxba_factor* f, int64_t bytes[4]) {
  VX_LOCK(const_cast<vxba_factor*>(f));
  if (!f || !bytes) return VXBA_ERR_ARG;
  const size_t d = sizeof(double);
  size_t store = (size_t)n_planes(f) * f->VS * d + (f->snapshot ? (size_t)N_CACHE_PLANES * f->snapshot_vs * d : 0);
  if (is_wide(f)) store += vxw::store_bytes(f->wstore);
  else if (f->clb) store += vxk::k3_clb_len(f->W, f->VS) * d;
  if (f->cl32) store += (size_t)10 * f->W * f->cl32_vs * sizeof(float);
  size_t work = f->partial2_len * d + f->partial3_len * d + f->xlen * d + sizeof(vxk::LMState);
  if (is_wide(f)) work += vxw::index_bytes(f->wide, f->W) + vxw::wide_solver_bytes(f->wide_solver) + sizeof(double) * 12 * VXBA_MAX_WIN_WIDE;
  const size_t scratch = f->staging_len * d + f->scratch_cap;
  bytes[0] = (int64_t)store; bytes[1] = (int64_t)work; bytes[2] = (int64_t)scratch; bytes[3] = (int64_t)(store + work + scratch);
  return VXBA_OK;
}

int vxba_algorithmic_bytes(const vxba_factor* f, double bytes[2]) {
  VX_LOCK(const_cast<vxba_factor*>(f));
  if (!f || !bytes) return VXBA_ERR_ARG;
  int64_t nnz = 0;
  int rc = vxba_nnz(const_cast<vxba_factor*>(f), &nnz);
  if (rc) return rc;
  // K3 = 80 nnz + 136 V read;  K2 = 80 nnz + 88 V read + 176 V written
  bytes[0] = 80.0 * (double)nnz + 136.0 * f->V;
  bytes[1] = (f->precision == VXBA_PRECISION_MIXED_F32_CLUSTERS ? 40.0 : 80.0) * (double)nnz + 264.0 * f->V;   // f32 cluster rows: 10 floats per entry
  return VXBA_OK;
}

}  // extern "C"

// ---- entry points that may have summed through the peers' mailboxes: a peer that never arrived must not pass silently ----
static int peer_check(vxba_factor* f, int rc) {
  if (rc != VXBA_OK || !f || !has_peer(f)) return rc;
  int st = 0;
  const int r2 = vxba_peer_status(f, &st);
  if (r2 != VXBA_OK) return r2;
  return st ? fail(f, VXBA_ERR_STATE, "peer all-reduce: a peer did not arrive within the wait bound (results are not a sum)") : VXBA_OK;
}
int vxba_damping_iter(vxba_factor* f, double* Rp, int max_iter, double* hess_out, double* resis_out, double* trace_out, int* n_trace, int* is_converge) {
  int rc = damping_iter_impl(f, Rp, max_iter, hess_out, resis_out, trace_out, n_trace, is_converge);
  if (rc == VXBA_ERR_STATE && f && f->solve_timed_out && has_collective(f)) {
    // Sharded: a timeout is a per-GPU event, and this rank has already issued the call's all-reduces -- a rank-local retry would issue
    // more of them which no other rank matches (RCCL hangs, the mailbox sequence numbers drift apart).  The error goes to the caller,
    // who switches VXBA_OPT_FUSED_SOLVE off on ALL ranks and calls again.
    f->solve_timed_out = false;
    return fail(f, VXBA_ERR_STATE, "in-launch solve timed out on a sharded factor: set VXBA_OPT_FUSED_SOLVE = 0 on every rank and retry");
  }
  if (rc == VXBA_ERR_STATE && f && f->solve_timed_out) {
    // The in-launch solve relies on workgroup 0 of the residual sweep making progress while the others poll (bounded): true for
    // in-order dispatch on an otherwise idle device, not guaranteed under CU masking / a serialising profiler / a co-resident
    // kernel.  A timeout is therefore not an error of the caller's: run the same call again with the solve as its own launch.
    // Rp is untouched on the failure path; the (lambda, U, merged) cache the first Hessian sweep needs is the one of the entry
    // poses, which the failed attempt has overwritten -- rebuild it first.
    f->solve_timed_out = false;
    const int saved = f->opt[VXBA_OPT_FUSED_SOLVE];
    f->opt[VXBA_OPT_FUSED_SOLVE] = 0;
    double r = 0;
    rc = vxba_evaluate_only_residual(f, Rp, 0, f->V, &r);
    if (rc == VXBA_OK) rc = damping_iter_impl(f, Rp, max_iter, hess_out, resis_out, trace_out, n_trace, is_converge);
    f->opt[VXBA_OPT_FUSED_SOLVE] = saved;
    f->fused_fallbacks++;
  }
  return peer_check(f, rc);
}
int vxba_lm_steps(vxba_factor* f, const double* Rp_init, int n_steps, int steps_per_solve, double* Rp_out, double* last_resis, int64_t* stats_out) {
  return peer_check(f, lm_steps_impl(f, Rp_init, n_steps, steps_per_solve, Rp_out, last_resis, stats_out));
}
