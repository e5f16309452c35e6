// C-ABI implementation, shared machinery (namespace vxc): device buffers and views of a factor, the asynchronous sweeps every driver is
// built from, the profiling brackets, and the all-reduce of a voxel-sharded factor (one-shot kernel over the peers' mailboxes, RCCL,
// or the caller's hook).  Entry points live in vxba_capi.hip (factor, sweeps, options), vxba_capi_lm.hip (LM drivers),
// vxba_capi_dist.hip (attach / detach of the collectives) and vxba_capi_li.hip (inertial half).
#include "vxba_capi_internal.hpp"

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
  const size_t p2 = std::max((size_t)want / 32 + 2, (size_t)512);   // one partial per workgroup of 32..64 voxels (vxk::k2_voxels_per_block), or one per sweep workgroup of a fused launch
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
  f->opt[VXBA_OPT_LI_DEVICE_LOOP] = 0;   // (removed in round 4; the slot stays so that the option numbers do not move)
  f->opt[VXBA_OPT_FUSED_SWEEPS] = flag("VXBA_FUSED_SWEEPS", 1);
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
  if (end == head) {
    VX_HIP(f, hipMemsetAsync(d_out, 0, plen * sizeof(double), f->stream));
    // a rank whose shard of a voxel-sharded wide window came out empty still takes part in the sum (the other ranks are waiting in it)
    return (is_wide(f) && !lm && has_collective(f)) ? shard_allreduce(f, d_out, plen) : VXBA_OK;
  }
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
    const bool reset = lm && fused_sweeps(f);   // a fused residual + Hessian launch may follow: its residual slots start as NaN
    vxk::launch_k3_finalize(f->d_partial3, nblocks, f->W, lm, c_now, has_collective(f) ? 0 : 1, d_out, f->stream, 0, nullptr, 0, reset ? f->d_partial2 : nullptr,
                            reset ? vxk::K23_MAX_SWEEP_BLOCKS : 0);
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
  if (end == head) {
    if (d_out) VX_HIP(f, hipMemsetAsync(d_out, 0, sizeof(double), f->stream));
    return (is_wide(f) && !lm && d_out && has_collective(f)) ? shard_allreduce(f, d_out, 1) : VXBA_OK;
  }
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
    nparts = vxk::launch_k2_residual(fv, pa, lm, c, fused_seq, head, end, part, f->opt[VXBA_OPT_K2_VOXELS_PER_BLOCK] | (f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] == 1 ? 0x10000 : 0), f->stream, a, b);
    if (a && b) f->pending.push_back({a, b, 1});
  } else {
    nparts = vxk::launch_k2_residual(fv, pa, lm, c, fused_seq, head, end, part, f->opt[VXBA_OPT_K2_VOXELS_PER_BLOCK] | (f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] == 1 ? 0x10000 : 0), f->stream);
  }
  if (nparts_out) *nparts_out = nparts;
  if (d_out) {
    vxk::launch_sum_partials(f->d_partial2, nparts, d_out, f->stream);
    VX_HIP(f, hipGetLastError());
    return shard_allreduce(f, d_out, 1);
  }
  return VXBA_OK;
}

// The device-resident LM loop's fused launch (vxba_k23.hpp): damped solve of ctl[*c] | residual sweep at its trial poses | Hessian sweep at the
// same poses, then the reduction that takes the step's accept / reject decision into ctl[*c ^ 1] and adopts the system of an accepted step.
// Replaces [solve + residual sweep] | Hessian sweep | reduction of two consecutive iterations wherever a Hessian sweep follows a residual sweep
// inside one solve; needs the in-launch solve, no collective, a narrow window and f64 cluster rows.
bool fused_sweeps(const vxba_factor* f) {
  // 1 (default): fused unless the factor's last call was reject-heavy; 2: always; 0: never
  if (f->opt[VXBA_OPT_FUSED_SWEEPS] == 1 && f->reject_heavy) return false;
  return f->opt[VXBA_OPT_FUSED_SWEEPS] != 0 && fused_solve(f) && !has_collective(f) && !is_wide(f) && f->V > 0 && f->precision != VXBA_PRECISION_MIXED_F32_CLUSTERS &&
         vxk::k23_supported(view(f));
}
int sweep_fused_device(vxba_factor* f, vxk::LMState* lm, int* c, unsigned seq) {
  int rc = ensure_partials3(f);
  if (rc) return rc;
  const FactorView fv = view(f);
  const int nv = vxk::k3_nv(f->W);
  const int nbatches = (f->V - 1) / nv + 1;
  const int nwg = vxk::k23_sweep_blocks(nbatches, f->cus);
  const int flags = f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] == 1 ? 1 : 0;
  int got;
  if (f->profiling & 32) {   // events bound to the dispatch itself
    hipEvent_t a = get_event(f), b = get_event(f);
    got = vxk::launch_k23_fused(fv, lm, *c, seq, 0, f->V, f->d_partial2, f->d_partial3, nwg, f->precision == VXBA_PRECISION_MIXED ? 1 : 0, flags, f->stream, a, b);
    if (a && b) f->pending.push_back({a, b, 5});
  } else {
    got = vxk::launch_k23_fused(fv, lm, *c, seq, 0, f->V, f->d_partial2, f->d_partial3, nwg, f->precision == VXBA_PRECISION_MIXED ? 1 : 0, flags, f->stream);
  }
  if (got != nwg) return fail(f, VXBA_ERR_STATE, "fused residual + Hessian launch refused this factor");
  *c ^= 1;   // the launch's solve workgroup has decided the step into the other control block; the reduction is gated by it
  {
    ScopedKernelTimer t(f, 2);
    // (no reset of the residual slots here: the launch's solve workgroup has put them back to NaN itself once it had read them all)
    vxk::launch_k3_finalize(f->d_partial3, nwg, f->W, lm, *c, 1, f->d_packed, f->stream, 0, nullptr, 0, nullptr, 0);
  }
  VX_HIP(f, hipGetLastError());
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

// The speculative loop with the fused launch (round 6): [solve of ctl[*c] | residual sweep at its trial poses | Hessian sweep at the same poses]
// in one launch -- flags bit 1: the solve workgroup takes NO decision, the shard's residual is not the window's -- then what spec_hess_phase
// runs behind its own Hessian sweep: the reduction with the residual sums riding behind the packed system, ONE all-reduce, the decision + adoption
// from the reduced buffer.  Same collectives per iteration, one launch and a cold sweep start fewer.  Every rank takes the same path (the
// predicate depends on options and on the LM history, which is identical on all ranks).
// Opt-in (VXBA_OPT_FUSED_SWEEPS = 2) on a factor with a collective: the fused launch takes whole CUs (one 8-wave workgroup with 147 KB of LDS
// each) and its sweep workgroups wait for workgroup 0 inside the launch -- fine with one process per GPU, but two process ranks SHARING a
// device (the plumbing runs of tests/test_gpu_two_rank.py, bench.py with VXBA_BENCH_DEVICE) time-slice each other for longer than the wait
// is bounded to, and a sharded factor cannot fall back rank-locally.  Measured with one rank through RCCL at cfg2: no gain at a 50k-voxel shard.
bool fused_sweeps_spec(const vxba_factor* f) {
  return f->opt[VXBA_OPT_FUSED_SWEEPS] == 2 && fused_solve(f) && spec_collective(f) && f->V > 0 && f->precision != VXBA_PRECISION_MIXED_F32_CLUSTERS &&
         vxk::k23_supported(view(f));
}
int spec_fused_phase(vxba_factor* f, const double* Rp0, int* c, unsigned seq, int* k2_nparts) {
  int rc = ensure_partials3(f);
  if (rc) return rc;
  PoseArg pa;
  fill_poses(f, Rp0, pa);
  const FactorView fv = view(f);
  const int nv = vxk::k3_nv(f->W);
  const int nbatches = (f->V - 1) / nv + 1;
  const int nwg = vxk::k23_sweep_blocks(nbatches, f->cus);
  const int flags = 2 | (f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] == 1 ? 1 : 0);
  int got;
  if (f->profiling & 32) {
    hipEvent_t a = get_event(f), b = get_event(f);
    got = vxk::launch_k23_fused(fv, f->d_lm, *c, seq, 0, f->V, f->d_partial2, f->d_partial3, nwg, f->precision == VXBA_PRECISION_MIXED ? 1 : 0, flags, f->stream, a, b);
    if (a && b) f->pending.push_back({a, b, 5});
  } else {
    got = vxk::launch_k23_fused(fv, f->d_lm, *c, seq, 0, f->V, f->d_partial2, f->d_partial3, nwg, f->precision == VXBA_PRECISION_MIXED ? 1 : 0, flags, f->stream);
  }
  if (got != nwg) return fail(f, VXBA_ERR_STATE, "fused residual + Hessian launch refused this factor");
  *k2_nparts = nwg;   // one residual sum per sweep workgroup
  {
    ScopedKernelTimer t(f, 2);
    vxk::launch_k3_finalize(f->d_partial3, nwg, f->W, f->d_lm, *c, 0, f->d_packed, f->stream, 1, f->d_partial2, nwg);
  }
  VX_HIP(f, hipGetLastError());
  rc = shard_allreduce(f, f->d_packed, vxba_packed_len(f) + 1);
  if (rc) return rc;
  vxk::launch_lm_spec_unpack(f->d_lm, *c, f->d_packed, f->W, 1, 0, pa, f->stream);
  *c ^= 1;
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
  VX_HIP(f, stream_wait_spin(f->stream));
  return VXBA_OK;
}
int sweep_residual_host(vxba_factor* f, const double* Rp, int head, int end, double* residual) {
  int rc = sweep_residual_device(f, Rp, nullptr, 0, head, end, f->d_scalar);
  if (rc) return rc;
  VX_HIP(f, hipMemcpyAsync(f->h_scalar, f->d_scalar, sizeof(double), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, stream_wait_spin(f->stream));
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
