// wbx_dist.hip — the one exchange step of the path on several GPUs, behind the C ABI.
//
// One process per GPU; tracks are sharded in contiguous ranges (wbx_shard_tracks), every rank mixes its own tracks
// into an UN-clamped partial master [K][C][F], and once per render the partials are summed onto rank 0 over RCCL
// (xGMI inside a node) and clamped there (the clamp of engine.cpp:1627-1636 must follow the sum, a clamped partial
// would be wrong).  Per-track peaks never leave the GPU that owns the track.  The message is K*C*F floats — 1 MiB at
// K = 256: latency-bound, far from a link's bandwidth — so the point is to keep it OFF the mix's critical path:
//
//   * the collective runs on a highest-priority stream of its own and only waits for the render's sum kernel;
//   * the partial masters rotate through a ring of three device buffers, so render i+1 and i+2 proceed while the
//     exchange of render i is in flight; a buffer is reused when the work that read it has finished (device-side
//     event, the host never blocks);
//   * the root's clamp writes the final master straight into the caller's destination (device memory or pinned,
//     device-mapped host memory — plain kernel stores, no copy engine).
//
// Two modes: WBX_DIST_REDUCE — one ncclReduce(sum) (RCCL's summation order is implementation-defined, inside the
// 1e-6 RMS budget); WBX_DIST_ORDERED — ncclGather of the N partials to the root and a fixed-order add
// ((((0 + p0) + p1) + ...) + pN-1, rank order = track order) in a kernel: bit-reproducible whatever the topology.
//
// RCCL is loaded on first use (dlopen of librccl.so.1): a single-GPU host never maps it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "wbx_ctx.h"

namespace wbx {
void launch_ordered_add(const float* gathered, float* dst, size_t n, uint32_t world, int clamp, hipStream_t s);
}

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl* rccl() {
  static Rccl r;
  if (r.lib || !r.err.empty()) return &r;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
  }
  if (!r.lib) {
    r.err = std::string("cannot load RCCL: ") + dlerror();
    return &r;
  }
  auto sym = [&](const char* name) {
    void* p = dlsym(r.lib, name);
    if (!p && r.err.empty()) r.err = std::string("RCCL lacks ") + name;
    return p;
  };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.Gather = reinterpret_cast<decltype(r.Gather)>(sym("ncclGather"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  return &r;
}

}  // namespace

namespace wbx {

constexpr int kDistRing = 3;

struct DistState {
  ncclComm_t comm = nullptr;
  uint32_t rank = 0, world = 1;
  int mode = WBX_DIST_REDUCE;
  hipStream_t comm_stream = nullptr;       // highest priority: the small exchange must not queue behind a mix
  float* master[kDistRing] = {};           // un-clamped partial masters, [max_blocks][C][F] each
  float* gathered = nullptr;               // root, ordered mode: [world][max_blocks*C*F]
  hipEvent_t slot_free[kDistRing] = {};    // the exchange (non-root: the send, root: the clamp) that last read the slot
  bool slot_used[kDistRing] = {};
  hipEvent_t sum_ev = nullptr;             // the render's sum, for in-stream sums
  uint64_t seq = 0;                        // renders so far
  int slot = -1;                           // ring slot of the last render
  bool exchanged = true;                   // the last render's exchange has been issued
  double* scalar = nullptr;                // device scratch of wbx_dist_max / _barrier
};

// called by launch_mix_sum: where this render's partial master goes; `sum_stream` (the stream its sum kernel runs on)
// is made to wait until the slot's previous exchange has read it
float* dist_begin_render(wbx_ctx* c, hipStream_t sum_stream, hipError_t* err) {
  DistState* d = c->dist;
  d->slot = (int)(d->seq++ % kDistRing);
  d->exchanged = false;
  *err = hipSuccess;
  if (d->slot_used[d->slot]) *err = hipStreamWaitEvent(sum_stream, d->slot_free[d->slot], 0);
  return d->master[d->slot];
}

void dist_destroy(wbx_ctx* c) {
  DistState* d = c->dist;
  if (!d) return;
  if (d->comm_stream) (void)hipStreamSynchronize(d->comm_stream);
  if (d->comm) (void)rccl()->CommDestroy(d->comm);
  for (int i = 0; i < kDistRing; i++) {
    if (d->master[i]) (void)hipFree(d->master[i]);
    if (d->slot_free[i]) (void)hipEventDestroy(d->slot_free[i]);
  }
  if (d->gathered) (void)hipFree(d->gathered);
  if (d->scalar) (void)hipFree(d->scalar);
  if (d->sum_ev) (void)hipEventDestroy(d->sum_ev);
  if (d->comm_stream) (void)hipStreamDestroy(d->comm_stream);
  delete d;
  c->dist = nullptr;
}

}  // namespace wbx

#define WBX_NCCL(ctx, call)                                                                             \
  do {                                                                                                  \
    ncclResult_t _r = (call);                                                                           \
    if (_r != ncclSuccess) return fail((ctx), WBX_ERR_DEVICE, (std::string(#call ": ") + rccl()->GetErrorString(_r)).c_str()); \
  } while (0)

// contiguous track ranges in rank order, so that the in-GPU summation order equals the reference's track order
// within a shard and the rank order of the ordered exchange equals it across shards
extern "C" void wbx_shard_tracks(uint32_t n_tracks, uint32_t world, uint32_t rank, uint32_t* first, uint32_t* count) {
  if (world == 0) world = 1;
  const uint32_t base = n_tracks / world, rem = n_tracks % world;
  if (first) *first = rank * base + (rank < rem ? rank : rem);
  if (count) *count = base + (rank < rem ? 1u : 0u);
}

extern "C" wbx_status wbx_dist_new_id(wbx_dist_id* out) {
  if (!out) return WBX_ERR_INVALID;
  static_assert(sizeof(wbx_dist_id) == sizeof(ncclUniqueId), "wbx_dist_id must hold an ncclUniqueId");
  Rccl* r = rccl();
  if (!r->err.empty()) return WBX_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return WBX_ERR_DEVICE;
  std::memcpy(out->bytes, &id, sizeof(id));
  return WBX_OK;
}

extern "C" wbx_status wbx_dist_init(wbx_ctx* c, const wbx_dist_id* id, uint32_t rank, uint32_t world, int mode) {
  if (!c || !id || world == 0 || rank >= world || (mode != WBX_DIST_REDUCE && mode != WBX_DIST_ORDERED)) return WBX_ERR_INVALID;
  if (c->dist) return fail(c, WBX_ERR_INVALID, "wbx_dist_init: already initialised");
  Rccl* r = rccl();
  if (!r->err.empty()) return fail(c, WBX_ERR_UNSUPPORTED, r->err.c_str());
  (void)hipSetDevice(c->cfg.device);
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, sync_main(c));
  DistState* d = new (std::nothrow) DistState();
  if (!d) return WBX_ERR_OOM;
  c->dist = d;
  d->rank = rank;
  d->world = world;
  d->mode = mode;
  const size_t n = (size_t)c->cfg.max_blocks * c->cfg.channels * c->cfg.block_frames;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipError_t e = hipStreamCreateWithPriority(&d->comm_stream, hipStreamNonBlocking, hi);
  for (int i = 0; i < kDistRing && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->master[i], n * sizeof(float));
    if (e == hipSuccess) e = hipMemset(d->master[i], 0, n * sizeof(float));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->slot_free[i], hipEventDisableTiming);
  }
  if (e == hipSuccess && mode == WBX_DIST_ORDERED && rank == 0) e = hipMalloc((void**)&d->gathered, n * world * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&d->scalar, 2 * sizeof(double));
  if (e == hipSuccess) e = hipEventCreateWithFlags(&d->sum_ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    dist_destroy(c);
    return fail(c, WBX_ERR_DEVICE, "wbx_dist_init", e);
  }
  ncclUniqueId nid;
  std::memcpy(&nid, id->bytes, sizeof(nid));
  const ncclResult_t nr = r->CommInitRank(&d->comm, (int)world, nid, (int)rank);   // collective: blocks until every rank has called
  if (nr != ncclSuccess) {
    d->comm = nullptr;
    dist_destroy(c);
    return fail(c, WBX_ERR_DEVICE, (std::string("ncclCommInitRank: ") + r->GetErrorString(nr)).c_str());
  }
  c->clamp = false;   // partials are clamped on the root, after the sum
  return WBX_OK;
}

extern "C" wbx_status wbx_dist_shutdown(wbx_ctx* c) {
  if (!c) return WBX_ERR_INVALID;
  if (!c->dist) return WBX_OK;
  (void)hipSetDevice(c->cfg.device);
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, sync_main(c));
  dist_destroy(c);
  c->clamp = true;
  return WBX_OK;
}

extern "C" wbx_status wbx_dist_info(wbx_ctx* c, uint32_t* rank, uint32_t* world, int* mode) {
  if (!c) return WBX_ERR_INVALID;
  if (rank) *rank = c->dist ? c->dist->rank : 0u;
  if (world) *world = c->dist ? c->dist->world : 1u;
  if (mode) *mode = c->dist ? c->dist->mode : WBX_DIST_REDUCE;
  return WBX_OK;
}

// After a submit / render: sum the ranks' partial masters onto rank 0 and, there, clamp the sum into `dst`
// ([K][C][F] floats: device memory or pinned, device-mapped host memory; ignored on the other ranks).  Asynchronous:
// everything is enqueued on the exchange stream behind the render's sum kernel.
extern "C" wbx_status wbx_dist_exchange(wbx_ctx* c, void* dst) {
  if (!c) return WBX_ERR_INVALID;
  DistState* d = c->dist;
  if (!d) return fail(c, WBX_ERR_INVALID, "wbx_dist_exchange: wbx_dist_init has not been called");
  if (d->slot < 0 || d->exchanged) return fail(c, WBX_ERR_FAILED, "wbx_dist_exchange: no render since the last exchange");
  if (d->rank == 0 && (!dst || ((uintptr_t)dst & 15u))) return fail(c, WBX_ERR_INVALID, "wbx_dist_exchange: the root needs a 16-byte aligned destination");
  (void)hipSetDevice(c->cfg.device);
  Rccl* r = rccl();
  const size_t n = (size_t)c->last_K * c->cfg.channels * c->cfg.block_frames;
  float* buf = d->master[d->slot];
  // the exchange stream waits for the kernel that wrote the partial: the render's sum (on its own stream for long
  // renders, on the main stream for short ones)
  if (c->sum_pending >= 0) {
    WBX_HIP(c, hipStreamWaitEvent(d->comm_stream, c->sum_done[c->sum_pending], 0));
  } else {
    WBX_HIP(c, hipEventRecord(d->sum_ev, c->stream));
    WBX_HIP(c, hipStreamWaitEvent(d->comm_stream, d->sum_ev, 0));
  }
  if (d->mode == WBX_DIST_ORDERED) {
    WBX_NCCL(c, r->Gather(buf, d->gathered, n, ncclFloat, 0, d->comm, d->comm_stream));
    if (d->rank == 0) launch_ordered_add(d->gathered, (float*)dst, n, d->world, 1, d->comm_stream);
  } else {
    WBX_NCCL(c, r->Reduce(buf, buf, n, ncclFloat, ncclSum, 0, d->comm, d->comm_stream));   // in place on the root
    if (d->rank == 0) launch_clamp_into(buf, (float*)dst, n, 1, d->comm_stream);
  }
  WBX_HIP(c, hipGetLastError());
  WBX_HIP(c, hipEventRecord(d->slot_free[d->slot], d->comm_stream));
  d->slot_used[d->slot] = true;
  d->exchanged = true;
  return WBX_OK;
}

extern "C" wbx_status wbx_dist_sync(wbx_ctx* c) {
  if (!c) return WBX_ERR_INVALID;
  if (!c->dist) return wbx_sync(c);
  (void)hipSetDevice(c->cfg.device);
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, sync_main(c));
  WBX_HIP(c, hipStreamSynchronize(c->dist->comm_stream));
  drain_events(c);
  return WBX_OK;
}

// max of `value` over all ranks, in place (the bench's max-over-ranks timing); also a barrier
extern "C" wbx_status wbx_dist_max(wbx_ctx* c, double* value) {
  if (!c || !value) return WBX_ERR_INVALID;
  DistState* d = c->dist;
  if (!d || d->world == 1) return WBX_OK;
  (void)hipSetDevice(c->cfg.device);
  WBX_HIP(c, hipMemcpyAsync(d->scalar, value, sizeof(double), hipMemcpyHostToDevice, d->comm_stream));
  WBX_NCCL(c, rccl()->AllReduce(d->scalar, d->scalar + 1, 1, ncclDouble, ncclMax, d->comm, d->comm_stream));
  WBX_HIP(c, hipMemcpyAsync(value, d->scalar + 1, sizeof(double), hipMemcpyDeviceToHost, d->comm_stream));
  WBX_HIP(c, hipStreamSynchronize(d->comm_stream));
  return WBX_OK;
}

extern "C" wbx_status wbx_dist_barrier(wbx_ctx* c) {
  double v = 0.0;
  return wbx_dist_max(c, &v);
}
