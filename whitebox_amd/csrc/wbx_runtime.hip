// wbx_runtime.hip — host side of libwbx.so: the C ABI of include/wbx.h over the gfx950 kernels.
//
//   layer 1 (wbx_ctx)     clip pool in HBM, routing, plan upload, launches, result fetch
//   layer 2 (wbx_engine)  the reference's Engine/Track surface: host keeps what the UI thread edits
//                         (clip lists, parameters, transport), the device keeps what the audio thread
//                         mutates per block (sequencer + sampler state) and does all per-block work
//
// There is no CPU implementation of the mix in this library: without a gfx950 device the create calls
// fail with WBX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numbers>
#include <string>
#include <vector>

#include "../../include/wbx.h"
#include "wbx_clip_edit.h"
#include "wbx_dev.h"
#include "wbx_seq.h"

namespace wbx {
void launch_plan(const PlanArgs& a, hipStream_t s);
void launch_gen(const GenArgs& a, uint32_t max_grid, hipStream_t s);
void launch_mix(const MixArgs& a, uint32_t n_blocks, int variant, bool stride_rows, hipStream_t s);
void launch_sum(const SumArgs& a, uint32_t n_blocks, hipStream_t s);
void launch_clamp(float* buf, size_t n, hipStream_t s);
void launch_clamp_into(const float* src, float* dst, size_t n, int clamp, hipStream_t s);
void launch_convert(const float* master, void* dst, uint32_t n_blocks, uint32_t F, uint32_t C, int fmt, hipStream_t s);
void launch_synth(void* dst, uint64_t frames, uint64_t key, float amp, int fmt, hipStream_t s);
void launch_deinterleave(const void* src, void* dst0, void* dst1, uint64_t frames, uint32_t channels, uint32_t elem,
                         hipStream_t s);
void launch_mip(const MipArgs& a, int format, int bits, hipStream_t s);
}  // namespace wbx

using namespace wbx;

namespace {

constexpr int kEventRing = 64;
// Plan buffers, partial-sum buffers and their events form a ring of three: the plan of render i may start as soon as
// the mix of render i-3 and the sum of render i-3 are over, i.e. a full render before its own mix — the one-wave-per-
// track plan kernel is starved for CU slots while a mix runs, so it needs that much slack to stay off the critical path.
constexpr int kRing = 3;
constexpr uint32_t kOverlapMinBlocks = 8;   // renders shorter than this run plan, mix and sum on the main stream

struct ClipSlot {
  void* base = nullptr;     // one allocation holding all channels
  size_t stride = 0;        // bytes between channel rows
  DSample d{};
  bool used = false;
  // waveform mip-maps (built on request): one allocation, level l at mip_off[l], [channels][mip_count[l]] elements
  void* mip = nullptr;
  int mip_bits = 0;
  std::vector<size_t> mip_off;
  std::vector<uint64_t> mip_count;
};

template <class T>
struct DevBuf {             // grow-only device array
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct wbx_ctx {
  wbx_config cfg{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  std::vector<ClipSlot> clips;
  DevBuf<DSample> d_samples;
  bool samples_dirty = true;

  // routing
  uint32_t routing_tracks = 0, n_buses = 0;
  std::vector<int32_t> track_bus;
  std::vector<uint32_t> order;
  std::vector<DGroup> groups;
  DevBuf<uint32_t> d_order;
  DevBuf<DGroup> d_groups;
  bool routing_dirty = true;

  // The plan of a render (track-block records, overflow pool, pre-render queue + rows) is double-buffered:
  // the sequencer of step i+1 runs on `plan_stream` while the mix of step i runs on `stream`.
  struct PlanBuf {
    DevBuf<DRow> prows;               // [K][N] 16-B plan rows
    DevBuf<DTrackBlock> tmpl;         // templates the rows point at (one per steady run / per block with events)
    uint32_t tmpl_cap = 0;
    DevBuf<DSeg> pool;
    uint32_t pool_chunks = 0;
    uint32_t* counters = nullptr;     // [0] pool chunks allocated, [1] status bits, [2] generic records queued,
                                      // [3] templates allocated
    DevBuf<uint32_t> gen_list;        // pre-render queue of KIND_GENERIC records
    DevBuf<float> rows;               // [gen_cap][C][F+8] pre-rendered mixing buffers
    DevBuf<DTrackBlock> saved;        // original records of the queue (plan read-back)
    uint32_t gen_cap = 0;
    hipEvent_t planned = nullptr;     // recorded on plan_stream when plan + pre-render are done
    hipEvent_t consumed = nullptr;    // (not owned) ctx->mix_done[] of the render whose mix read this buffer
    bool consumed_valid = false;
  } pb[kRing];
  int cur = 0;
  hipStream_t plan_stream = nullptr;
  bool overlap = true;
  DevBuf<float> d_zero;               // zero page (F+8 floats)
  uint32_t* levels_target = nullptr;  // [N][C] running per-track maxima (VUMeter::level), or null
  DevBuf<float> d_partial2[kRing];    // group partials, one per render in flight (a sum may still read an older one)
  DevBuf<float> d_master, d_buses, d_peaks, d_gains;
  // The sum of render i runs on its own stream beside the mix of render i+1 (it is PCIe-bound when the master goes to
  // host memory and needs few CUs).  sum_pending: a sum has been issued that the main stream has not waited for yet.
  hipStream_t sum_stream = nullptr;
  hipEvent_t mix_done[kRing] = {}, sum_done[kRing] = {};
  bool sum_valid[kRing] = {};
  int sum_pending = -1;
  uint32_t render_seq = 0;
  bool partial_wait_done = false;     // the caller already ordered this render after the sum of two renders ago
  bool sum_overlap = true;            // WBX_SUM_OVERLAP=0: sum on the main stream
  DevBuf<uint8_t> d_conv;
  std::vector<DTrackBlock> h_tb;      // layer-1 staging
  std::vector<DRow> h_rows;
  std::vector<DSeg> h_pool;

  uint32_t last_K = 0, last_N = 0;
  uint32_t* status_dst = nullptr;     // set by wbx_engine_process around its render: where sum_kernel drops the plan status
  bool buses_alias_partials = false;  // see build_routing
  const float* last_buses = nullptr;  // where the last render's bus sums are: d_buses or the partial buffer
  bool buses_clean = false;           // d_buses zeroed since the last routing change / reallocation
  float* last_master = nullptr;       // where the last render / submit put its master (d_master, the caller's target, or
  bool last_master_on_host = false;   // the engine's pinned staging block, which is host memory)
  bool clamp = true;
  float* master_target = nullptr;     // caller-owned device buffer, or null: d_master

  // kernel timing (mix kernel)
  hipEvent_t ev[kEventRing][3]{};       // before the mix, after the mix, after the sum
  int ev_pending = 0;
  double mix_ms_total = 0.0;
  double tail_ms_total = 0.0;          // mix end -> sum end (launch gap + sum kernel incl. its PCIe stores)
  uint64_t mix_launches = 0;
  bool profiling = true;
  int mix_unroll = 0;                 // WBX_MIX_VARIANT=10*U+W forces a kernel variant (results are identical);
                                      // 0 = chosen per render: 24 when resampled or integer-PCM clips are present, else 43
  bool has_window_clips = true;
  bool has_integer_clips = false;
  bool force_g = false;
  bool has_stride_clips = true;       // fp32 clips played at speed > 0.999, != 1 may occur (layer 1: unknown, assume so)
};

namespace {

inline wbx_ctx::PlanBuf& PB(wbx_ctx* c) { return c->pb[c->cur]; }

wbx_status fail(wbx_ctx* c, wbx_status s, const char* what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return s;
}

#define WBX_HIP(ctx, call)                                                  \
  do {                                                                      \
    hipError_t _e = (call);                                                 \
    if (_e != hipSuccess) return fail((ctx), WBX_ERR_DEVICE, #call, _e);    \
  } while (0)

size_t fmt_bytes(int fmt) {
  switch (fmt) {
    case WBX_FMT_I16: return 2;
    case WBX_FMT_I24:
    case WBX_FMT_I32:
    case WBX_FMT_F32: return 4;
    default: return 0;
  }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// make the main stream wait for the sum that is still running beside it (device-side; a following
// hipStreamSynchronize(c->stream) then covers it)
hipError_t join_sum(wbx_ctx* c) {
  if (c->sum_pending < 0) return hipSuccess;
  const hipError_t e = hipStreamWaitEvent(c->stream, c->sum_done[c->sum_pending], 0);
  c->sum_pending = -1;
  return e;
}

void drain_events(wbx_ctx* c) {
  for (int i = 0; i < c->ev_pending; i++) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[i][0], c->ev[i][1]) == hipSuccess) {
      c->mix_ms_total += ms;
      c->mix_launches++;
      if (hipEventElapsedTime(&ms, c->ev[i][1], c->ev[i][2]) == hipSuccess) c->tail_ms_total += ms;
    }
  }
  c->ev_pending = 0;
}

// default routing: identity order, groups of group_size, everything straight into the master
void build_routing(wbx_ctx* c, uint32_t n_tracks) {
  const uint32_t G = c->cfg.group_size;
  c->order.clear();
  c->groups.clear();
  auto emit = [&](const std::vector<uint32_t>& members, int32_t bus) {
    for (size_t i = 0; i < members.size(); i += G) {
      DGroup g{};
      g.first = (uint32_t)c->order.size();
      g.count = (uint32_t)std::min<size_t>(G, members.size() - i);
      g.bus = bus;
      for (uint32_t k = 0; k < g.count; k++) c->order.push_back(members[i + k]);
      c->groups.push_back(g);
    }
  };
  std::vector<uint32_t> direct;
  std::vector<std::vector<uint32_t>> per_bus(c->n_buses);
  for (uint32_t t = 0; t < n_tracks; t++) {
    int32_t bus = (c->n_buses && t < c->track_bus.size()) ? c->track_bus[t] : -1;
    if (bus >= 0 && (uint32_t)bus < c->n_buses)
      per_bus[bus].push_back(t);
    else
      direct.push_back(t);
  }
  emit(direct, -1);
  for (uint32_t u = 0; u < c->n_buses; u++) emit(per_bus[u], (int32_t)u);
  c->routing_tracks = n_tracks;
  c->routing_dirty = true;
  // every bus exactly one group and nothing routed straight to the master: group g's partial sum IS bus g's sum, so
  // the bus output can alias the partial buffer instead of being written a second time by the sum kernel
  c->buses_alias_partials = c->n_buses > 0 && c->groups.size() == c->n_buses;
  for (size_t g = 0; g < c->groups.size() && c->buses_alias_partials; g++)
    if (c->groups[g].bus != (int32_t)g) c->buses_alias_partials = false;
}

wbx_status upload_tables(wbx_ctx* c, uint32_t n_tracks) {
  if (c->routing_tracks != n_tracks) {
    build_routing(c, n_tracks);
    c->buses_clean = false;
  }
  if (c->routing_dirty) {
    WBX_HIP(c, join_sum(c));                          // a sum beside the main stream may still read d_groups
    WBX_HIP(c, hipStreamSynchronize(c->stream));
    WBX_HIP(c, c->d_order.ensure(std::max<size_t>(1, c->order.size())));
    WBX_HIP(c, c->d_groups.ensure(std::max<size_t>(1, c->groups.size())));
    if (!c->order.empty())
      WBX_HIP(c, hipMemcpyAsync(c->d_order.p, c->order.data(), c->order.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                                c->stream));
    if (!c->groups.empty())
      WBX_HIP(c, hipMemcpyAsync(c->d_groups.p, c->groups.data(), c->groups.size() * sizeof(DGroup),
                                hipMemcpyHostToDevice, c->stream));
    WBX_HIP(c, hipStreamSynchronize(c->stream));   // host vectors may change right after
    c->routing_dirty = false;
  }
  if (c->samples_dirty) {
    std::vector<DSample> tab(c->clips.size());
    c->has_integer_clips = false;
    for (size_t i = 0; i < c->clips.size(); i++) {
      tab[i] = c->clips[i].d;
      if (c->clips[i].used && c->clips[i].d.format != FMT_F32) c->has_integer_clips = true;
    }
    WBX_HIP(c, c->d_samples.ensure(std::max<size_t>(1, tab.size())));
    if (!tab.empty()) WBX_HIP(c, hipMemcpy(c->d_samples.p, tab.data(), tab.size() * sizeof(DSample), hipMemcpyHostToDevice));
    c->samples_dirty = false;
  }
  return WBX_OK;
}

wbx_status ensure_result_buffers(wbx_ctx* c, uint32_t K, uint32_t N) {
  const size_t CF = (size_t)c->cfg.channels * c->cfg.block_frames;
  for (auto& B : c->pb)
    if (B.prows.cap < (size_t)K * N) {
      WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
      WBX_HIP(c, hipStreamSynchronize(c->stream));
      WBX_HIP(c, B.prows.ensure((size_t)K * N));
    }
  const size_t need_partial = (size_t)K * std::max<size_t>(1, c->groups.size()) * CF;
  if (c->d_partial2[0].cap < need_partial || c->d_master.cap < (size_t)K * CF ||
      c->d_peaks.cap < (size_t)K * N * c->cfg.channels || (c->n_buses && c->d_buses.cap < (size_t)K * c->n_buses * CF)) {
    WBX_HIP(c, join_sum(c));                          // a sum may still be using the buffers about to be replaced
    WBX_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& v : c->sum_valid) v = false;
  }
  for (auto& P : c->d_partial2) WBX_HIP(c, P.ensure(need_partial));
  WBX_HIP(c, c->d_master.ensure((size_t)K * CF));
  WBX_HIP(c, c->d_peaks.ensure((size_t)K * N * c->cfg.channels));
  if (c->n_buses && c->d_buses.cap < (size_t)K * c->n_buses * CF) {
    WBX_HIP(c, hipStreamSynchronize(c->stream));
    WBX_HIP(c, c->d_buses.ensure((size_t)K * c->n_buses * CF));
    c->buses_clean = false;
  }
  return WBX_OK;
}

// room for `n` templates in both plan buffers (grow-only)
wbx_status ensure_template_capacity(wbx_ctx* c, size_t n) {
  n = std::max<size_t>(n, 64);
  for (auto& B : c->pb) {
    if (n <= B.tmpl_cap) continue;
    WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
    WBX_HIP(c, hipStreamSynchronize(c->stream));
    WBX_HIP(c, B.tmpl.ensure(n));
    B.tmpl_cap = (uint32_t)n;
  }
  return WBX_OK;
}

// room for `rows` pre-rendered generic track-blocks (grow-only)
wbx_status ensure_gen_capacity(wbx_ctx* c, size_t rows) {
  rows = std::max<size_t>(rows, 64);
  const size_t row_floats = (size_t)c->cfg.channels * (c->cfg.block_frames + 8);
  for (auto& B : c->pb) {
    if (rows <= B.gen_cap) continue;
    WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
    WBX_HIP(c, hipStreamSynchronize(c->stream));
    WBX_HIP(c, B.gen_list.ensure(rows));
    WBX_HIP(c, B.rows.ensure(rows * row_floats));
    WBX_HIP(c, B.saved.ensure(rows));
    B.gen_cap = (uint32_t)rows;
  }
  return WBX_OK;
}

// pre-render of the queued generic records of the current plan buffer
wbx_status launch_pre_render(wbx_ctx* c, uint32_t K, hipStream_t on) {
  const uint32_t C = c->cfg.channels, F = c->cfg.block_frames;
  GenArgs ga{};
  ga.tmpl = PB(c).tmpl.p;
  ga.pool = PB(c).pool.p;
  ga.gen_list = PB(c).gen_list.p;
  ga.gen_count = PB(c).counters + 2;
  ga.rows = PB(c).rows.p;
  ga.saved = PB(c).saved.p;
  ga.gen_cap = PB(c).gen_cap;
  ga.block_frames = F;
  ga.channels = C;
  // one wave per queued row, grid-stride: no more workgroups than the device holds at once (256 CUs x 6 workgroups at
  // the kernel's register budget), or the surplus would start when the first ones have finished their whole share
  launch_gen(ga, K < kOverlapMinBlocks ? 64u * K : 1536u, on);
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

// mix + sum over the current plan buffer, on the main stream
wbx_status launch_mix_sum(wbx_ctx* c, uint32_t K, uint32_t N) {
  const uint32_t C = c->cfg.channels, F = c->cfg.block_frames;
  MixArgs m{};
  m.rows = PB(c).prows.p;
  m.tmpl = PB(c).tmpl.p;
  m.zero_page = c->d_zero.p;
  m.pool = PB(c).pool.p;
  m.order = c->d_order.p;
  m.groups = c->d_groups.p;
  const int pp = (int)(c->render_seq % kRing);
  // this partial buffer was last read by the sum of kRing renders ago; the engine path has already made the PLAN
  // stream wait for that sum (the mix waits for the plan), which keeps the barrier off the main stream
  if (c->sum_valid[pp] && !c->partial_wait_done) WBX_HIP(c, hipStreamWaitEvent(c->stream, c->sum_done[pp], 0));
  c->partial_wait_done = false;
  m.partial = c->d_partial2[pp].p;
  m.peaks = c->d_peaks.p;
  m.levels = c->levels_target;
  m.n_tracks = N;
  m.n_groups = (uint32_t)c->groups.size();
  m.block_frames = F;
  m.channels = C;
  m.tiles = ((C * F / 4) + 255u) / 256u;
  m.n_blocks = K;
  if (m.tiles > 1) WBX_HIP(c, hipMemsetAsync(c->d_peaks.p, 0, (size_t)K * N * C * sizeof(float), c->stream));
  // the kernel timer is for batch renders; the one-block callback path skips its three event records
  const bool timed = c->profiling && K > 1;
  if (m.n_groups) {
    if (timed) {
      if (c->ev_pending == kEventRing) {
        WBX_HIP(c, hipEventSynchronize(c->ev[kEventRing - 1][1]));
        drain_events(c);
      }
      WBX_HIP(c, hipEventRecord(c->ev[c->ev_pending][0], c->stream));
    }
    launch_mix(m, K, c->mix_unroll ? c->mix_unroll : ((c->has_window_clips || c->has_integer_clips) ? 24 : 43),
               // the G instances also carry the pipelined modes for chunks that mix storage formats with resampled rows
               c->force_g || c->has_stride_clips || (c->has_window_clips && c->has_integer_clips), c->stream);
    if (timed) {
      WBX_HIP(c, hipEventRecord(c->ev[c->ev_pending][1], c->stream));
    }
  }
  // the plan buffer is free as soon as the MIX has read it: releasing it before the sum lets the next plan run
  // beside sum_kernel (the GPU is nearly idle there) instead of competing with the next mix for CU slots — started
  // together with a mix, the one-wave-per-track plan kernel is starved until that mix drains
  WBX_HIP(c, hipEventRecord(c->mix_done[pp], c->stream));
  SumArgs s{};
  s.partial = c->d_partial2[pp].p;
  s.groups = c->d_groups.p;
  s.master = c->master_target ? c->master_target : c->d_master.p;
  c->last_master = s.master;
  c->last_master_on_host = false;
  s.buses = (c->n_buses && !c->buses_alias_partials) ? c->d_buses.p : nullptr;
  c->last_buses = c->n_buses ? (c->buses_alias_partials ? c->d_partial2[pp].p : c->d_buses.p) : nullptr;
  s.n_groups = m.n_groups;
  s.n_buses = c->n_buses;
  s.block_frames = F;
  s.channels = C;
  s.clamp = c->clamp ? 1u : 0u;
  s.status_src = c->status_dst ? PB(c).counters : nullptr;
  s.status_dst = c->status_dst;
  if (c->n_buses && !c->buses_alias_partials && !c->buses_clean) {
    // buses without member groups must read as zero; every bus that has members is rewritten by each render, so the
    // buffer only needs clearing when the routing or the allocation changed (64 MB per render saved on config 4)
    WBX_HIP(c, hipMemsetAsync(c->d_buses.p, 0, c->d_buses.cap * sizeof(float), c->stream));
    c->buses_clean = true;
  }
  // short renders (the one-block callback path above all) keep everything on the main stream: the cross-stream
  // hand-overs cost more than the few microseconds of overlap they could buy
  const bool sum_beside = c->sum_overlap && K >= kOverlapMinBlocks;
  hipStream_t ss = sum_beside ? c->sum_stream : c->stream;
  if (sum_beside) WBX_HIP(c, hipStreamWaitEvent(ss, c->mix_done[pp], 0));   // (an earlier pending sum is ordered before this one by ss)
  launch_sum(s, K, ss);
  if (m.n_groups && timed) {
    WBX_HIP(c, hipEventRecord(c->ev[c->ev_pending][2], ss));
    c->ev_pending++;
  }
  if (sum_beside) {
    WBX_HIP(c, hipEventRecord(c->sum_done[pp], ss));
    c->sum_valid[pp] = true;
    c->sum_pending = pp;
  }
  c->render_seq++;
  WBX_HIP(c, hipGetLastError());
  c->last_K = K;
  c->last_N = N;
  return WBX_OK;
}

}  // namespace

// =================================================================================================
// library
// =================================================================================================
extern "C" const char* wbx_version(void) { return "wbx 0.1 (gfx950)"; }

extern "C" int wbx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

extern "C" const char* wbx_status_string(wbx_status s) {
  switch (s) {
    case WBX_OK: return "ok";
    case WBX_ERR_FAILED: return "failed";
    case WBX_ERR_UNIMPLEMENTED: return "unimplemented";
    case WBX_ERR_UNSUPPORTED: return "unsupported";
    case WBX_ERR_INVALID: return "invalid argument";
    case WBX_ERR_NO_DEVICE: return "no gfx950 device";
    case WBX_ERR_DEVICE: return "HIP error";
    case WBX_ERR_OOM: return "out of memory";
    case WBX_ERR_OVERFLOW: return "segment plan overflow";
    default: return "unknown";
  }
}

// =================================================================================================
// layer 1
// =================================================================================================
extern "C" wbx_status wbx_create(const wbx_config* cfg, wbx_ctx** out) {
  if (!cfg || !out) return WBX_ERR_INVALID;
  *out = nullptr;
  if (cfg->channels < 1 || cfg->channels > 2 || cfg->block_frames < 4 || (cfg->block_frames & 3u) ||
      cfg->block_frames > 32768 || cfg->max_tracks == 0 || cfg->max_blocks == 0 || cfg->max_blocks > 2048 ||
      cfg->sample_rate == 0)
    return WBX_ERR_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || cfg->device < 0 || cfg->device >= n) return WBX_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return WBX_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return WBX_ERR_NO_DEVICE;   // kernels exist for gfx950 only
  if (hipSetDevice(cfg->device) != hipSuccess) return WBX_ERR_NO_DEVICE;

  wbx_ctx* c = new (std::nothrow) wbx_ctx();
  if (!c) return WBX_ERR_OOM;
  c->cfg = *cfg;
  // default 128: one staging round per workgroup; a context that can only render one block per call (the audio
  // callback) takes 64 — twice the workgroups for the one block, ≈15 % less latency at 4096 tracks
  if (c->cfg.group_size == 0) c->cfg.group_size = c->cfg.max_blocks == 1 ? kStage / 2 : kStage;
  if (const char* u = std::getenv("WBX_MIX_VARIANT")) c->mix_unroll = std::atoi(u);
  if (const char* u = std::getenv("WBX_FORCE_G")) c->force_g = std::atoi(u) != 0;   // A/B aid: always the G instances
  if (cfg->stream) {
    c->stream = (hipStream_t)cfg->stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return WBX_ERR_DEVICE;
    }
    c->own_stream = true;
  }
  for (int i = 0; i < kEventRing; i++) {
    if (hipEventCreate(&c->ev[i][0]) != hipSuccess || hipEventCreate(&c->ev[i][1]) != hipSuccess ||
        hipEventCreate(&c->ev[i][2]) != hipSuccess) {
      wbx_destroy(c);
      return WBX_ERR_DEVICE;
    }
  }
  {
    // The sequencer runs beside the mix of the previous render.  It is a few dozen latency-bound waves (one
    // lane per track): at low or equal priority they starve behind the 16 mix waves of their CU and the
    // plan becomes the critical path, so the plan stream gets the HIGHEST priority — the issue slots it
    // takes from the bandwidth-bound mix are negligible.  WBX_OVERLAP=0 runs everything on the main stream;
    // WBX_PLAN_PRIO=lo|hi picks the priority (tuning knobs).
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const char* ov = std::getenv("WBX_OVERLAP");
    c->overlap = !(ov && ov[0] == '0');
    const char* pp = std::getenv("WBX_PLAN_PRIO");
    const int prio = (pp && pp[0] == 'l') ? lo : hi;
    if (hipStreamCreateWithPriority(&c->plan_stream, hipStreamNonBlocking, prio) != hipSuccess) {
      wbx_destroy(c);
      return WBX_ERR_DEVICE;
    }
    // the sum stream: highest priority as well — its few hundred small workgroups start while the next mix floods
    // the device.  WBX_SUM_OVERLAP=0 keeps the sum on the main stream.
    const char* so = std::getenv("WBX_SUM_OVERLAP");
    c->sum_overlap = !(so && so[0] == '0');
    const char* sp = std::getenv("WBX_SUM_PRIO");
    bool ok = hipStreamCreateWithPriority(&c->sum_stream, hipStreamNonBlocking, (sp && sp[0] == 'l') ? lo : hi) == hipSuccess;
    for (int i = 0; i < kRing && ok; i++)
      ok = hipEventCreateWithFlags(&c->mix_done[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&c->sum_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      wbx_destroy(c);
      return WBX_ERR_DEVICE;
    }
  }
  size_t chunks = cfg->max_segments ? (cfg->max_segments + kChunk - 1) / kChunk
                                    : std::max<size_t>(1024, (size_t)cfg->max_blocks * cfg->max_tracks / 8);
  for (auto& B : c->pb) {
    B.pool_chunks = (uint32_t)chunks;
    if (B.pool.ensure(chunks * kChunk) != hipSuccess || hipMalloc((void**)&B.counters, 4 * sizeof(uint32_t)) != hipSuccess ||
        hipEventCreateWithFlags(&B.planned, hipEventDisableTiming) != hipSuccess) {
      wbx_destroy(c);
      return WBX_ERR_OOM;
    }
    (void)hipMemset(B.counters, 0, 4 * sizeof(uint32_t));
  }
  if (c->d_zero.ensure(cfg->block_frames + 8) != hipSuccess) {
    wbx_destroy(c);
    return WBX_ERR_OOM;
  }
  (void)hipMemset(c->d_zero.p, 0, (cfg->block_frames + 8) * sizeof(float));
  *out = c;
  return WBX_OK;
}

extern "C" void wbx_destroy(wbx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  if (c->plan_stream) (void)hipStreamSynchronize(c->plan_stream);
  if (c->sum_stream) (void)hipStreamSynchronize(c->sum_stream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& s : c->clips) {
    if (s.base) (void)hipFree(s.base);
    if (s.mip) (void)hipFree(s.mip);
  }
  c->d_samples.release();
  c->d_order.release();
  c->d_groups.release();
  for (auto& B : c->pb) {
    B.prows.release();
    B.tmpl.release();
    B.pool.release();
    B.gen_list.release();
    B.rows.release();
    B.saved.release();
    if (B.counters) (void)hipFree(B.counters);
    if (B.planned) (void)hipEventDestroy(B.planned);
  }
  if (c->plan_stream) (void)hipStreamDestroy(c->plan_stream);
  if (c->sum_stream) (void)hipStreamDestroy(c->sum_stream);
  for (int i = 0; i < kRing; i++) {
    if (c->mix_done[i]) (void)hipEventDestroy(c->mix_done[i]);
    if (c->sum_done[i]) (void)hipEventDestroy(c->sum_done[i]);
  }
  for (auto& P : c->d_partial2) P.release();
  c->d_master.release();
  c->d_buses.release();
  c->d_peaks.release();
  c->d_gains.release();
  c->d_conv.release();
  c->d_zero.release();
  for (int i = 0; i < kEventRing; i++) {
    if (c->ev[i][0]) (void)hipEventDestroy(c->ev[i][0]);
    if (c->ev[i][1]) (void)hipEventDestroy(c->ev[i][1]);
    if (c->ev[i][2]) (void)hipEventDestroy(c->ev[i][2]);
  }
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char* wbx_last_error(const wbx_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

static wbx_status clip_alloc(wbx_ctx* c, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                             uint64_t frames, size_t* stride_out) {
  if (!c) return WBX_ERR_INVALID;
  const size_t eb = fmt_bytes(format);
  if (!eb) return fail(c, WBX_ERR_UNSUPPORTED, "clip format");
  if (channels < 1 || channels > 2) return fail(c, WBX_ERR_UNSUPPORTED, "clip channel count (1 or 2)");
  if (frames >= 2147483632ull) return fail(c, WBX_ERR_UNSUPPORTED, "clip longer than 2^31-16 frames");
  if (clip >= (1u << 24)) return fail(c, WBX_ERR_INVALID, "clip id");
  (void)hipSetDevice(c->cfg.device);
  if (clip >= c->clips.size()) c->clips.resize(clip + 1);
  ClipSlot& s = c->clips[clip];
  if (s.base) {
    WBX_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(s.base);
    if (s.mip) (void)hipFree(s.mip);
    s = ClipSlot{};
  }
  const size_t stride = align_up((frames + kPad) * eb, 256);
  WBX_HIP(c, hipMalloc(&s.base, stride * channels));
  s.d.ch[0] = s.base;
  s.d.ch[1] = channels > 1 ? (const void*)((const char*)s.base + stride) : s.base;   // mono wraps (i % channels)
  s.d.count = frames;
  s.d.format = (uint32_t)format;
  s.d.channels = channels;
  s.d.sample_rate = sample_rate;
  s.used = true;
  s.stride = stride;
  c->samples_dirty = true;
  *stride_out = stride;
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_upload(wbx_ctx* c, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                                      uint64_t frames, const void* const* planar) {
  if (!c || !planar) return WBX_ERR_INVALID;
  size_t stride = 0;
  wbx_status st = clip_alloc(c, clip, format, channels, sample_rate, frames, &stride);
  if (st != WBX_OK) return st;
  const size_t eb = fmt_bytes(format);
  ClipSlot& s = c->clips[clip];
  WBX_HIP(c, hipMemsetAsync(s.base, 0, stride * channels, c->stream));   // the 16 padding frames read as zero
  for (uint32_t ch = 0; ch < channels; ch++)
    WBX_HIP(c, hipMemcpyAsync((char*)s.base + stride * ch, planar[ch], frames * eb, hipMemcpyHostToDevice, c->stream));
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_synth(wbx_ctx* c, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                                     uint64_t frames, uint64_t seed, uint32_t key_track, float amp) {
  if (!c) return WBX_ERR_INVALID;
  size_t stride = 0;
  wbx_status st = clip_alloc(c, clip, format, channels, sample_rate, frames, &stride);
  if (st != WBX_OK) return st;
  ClipSlot& s = c->clips[clip];
  for (uint32_t ch = 0; ch < channels; ch++) {
    const uint64_t key = seed ^ ((uint64_t)key_track << 40) ^ ((uint64_t)ch << 32);
    launch_synth((char*)s.base + stride * ch, frames, key, amp, format, c->stream);
  }
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}


// ---- clip ingest (dsp/sample.cpp:29-43, :112-197) ------------------------------------------------
extern "C" wbx_status wbx_clip_ingest_device(wbx_ctx* c, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                                             uint64_t frames, const void* device_interleaved) {
  if (!c || (!device_interleaved && frames)) return WBX_ERR_INVALID;
  if ((uintptr_t)device_interleaved & 15u) return fail(c, WBX_ERR_INVALID, "interleaved device buffer must be 16-byte aligned");
  size_t stride = 0;
  wbx_status st = clip_alloc(c, clip, format, channels, sample_rate, frames, &stride);
  if (st != WBX_OK) return st;
  const size_t eb = fmt_bytes(format);
  ClipSlot& s = c->clips[clip];
  // the 16 padding frames (and the alignment slack) read as zero: sample.cpp:127,140
  for (uint32_t ch = 0; ch < channels; ch++)
    WBX_HIP(c, hipMemsetAsync((char*)s.base + stride * ch + frames * eb, 0, stride - frames * eb, c->stream));
  launch_deinterleave(device_interleaved, s.base, (char*)s.base + (channels > 1 ? stride : 0), frames, channels, (uint32_t)eb,
                      c->stream);
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_upload_interleaved(wbx_ctx* c, uint32_t clip, int format, uint32_t channels,
                                                  uint32_t sample_rate, uint64_t frames, const void* interleaved) {
  if (!c || (!interleaved && frames)) return WBX_ERR_INVALID;
  size_t stride = 0;
  wbx_status st = clip_alloc(c, clip, format, channels, sample_rate, frames, &stride);
  if (st != WBX_OK) return st;
  const size_t eb = fmt_bytes(format);
  ClipSlot& s = c->clips[clip];
  for (uint32_t ch = 0; ch < channels; ch++)
    WBX_HIP(c, hipMemsetAsync((char*)s.base + stride * ch + frames * eb, 0, stride - frames * eb, c->stream));
  // chunks of the decoder's interleaved output go host -> device staging -> transposed into the channel rows;
  // two staging buffers so the copy of chunk i+1 overlaps the transposition of chunk i
  const uint64_t chunk_frames = (uint64_t)4 << 20;   // a multiple of 4 frames: lane groups never straddle chunks
  const size_t chunk_bytes = (size_t)chunk_frames * channels * eb;
  void* stage[2] = {nullptr, nullptr};
  hipEvent_t freed[2] = {nullptr, nullptr};
  const int nstage = frames > chunk_frames ? 2 : 1;
  for (int i = 0; i < nstage; i++) {
    WBX_HIP(c, hipMalloc(&stage[i], std::min<size_t>(chunk_bytes, std::max<size_t>(16, (size_t)frames * channels * eb))));
    WBX_HIP(c, hipEventCreateWithFlags(&freed[i], hipEventDisableTiming));
  }
  hipError_t err = hipSuccess;
  uint64_t done = 0;
  for (int i = 0; done < frames && err == hipSuccess; i++) {
    const uint64_t n = std::min<uint64_t>(chunk_frames, frames - done);
    const int b = i % nstage;
    if (i >= nstage) err = hipEventSynchronize(freed[b]);
    if (err == hipSuccess)
      err = hipMemcpyAsync(stage[b], (const char*)interleaved + (size_t)done * channels * eb, (size_t)n * channels * eb,
                           hipMemcpyHostToDevice, c->stream);
    if (err == hipSuccess) {
      launch_deinterleave(stage[b], (char*)s.base + (size_t)done * eb, (char*)s.base + (channels > 1 ? stride : 0) + (size_t)done * eb,
                          n, channels, (uint32_t)eb, c->stream);
      err = hipEventRecord(freed[b], c->stream);
    }
    done += n;
  }
  if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
  for (int i = 0; i < nstage; i++) {
    (void)hipFree(stage[i]);
    (void)hipEventDestroy(freed[i]);
  }
  if (err != hipSuccess) return fail(c, WBX_ERR_DEVICE, "clip ingest", err);
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_download(wbx_ctx* c, uint32_t clip, uint32_t channel, void* dst) {
  if (!c || !dst || clip >= c->clips.size() || !c->clips[clip].used) return WBX_ERR_INVALID;
  const ClipSlot& s = c->clips[clip];
  if (channel >= s.d.channels) return fail(c, WBX_ERR_INVALID, "channel out of range");
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  WBX_HIP(c, hipMemcpy(dst, (const char*)s.base + s.stride * channel, (size_t)s.d.count * fmt_bytes((int)s.d.format), hipMemcpyDeviceToHost));
  return WBX_OK;
}

// ---- waveform mip-maps (gfx/waveform_visual.cpp:9-246) -------------------------------------------
extern "C" uint32_t wbx_mip_levels(uint64_t frames) {
  uint32_t n = 0;
  for (uint64_t sample_count = frames; sample_count > 64; sample_count /= 4) n++;   // :195, :236
  return n;
}

extern "C" uint64_t wbx_mip_data_count(uint64_t frames, uint32_t level) {
  const uint64_t block_count = (uint64_t)1 << (2u * level);   // 2^(current_mip - 1), current_mip = 1 + 2*level
  uint64_t n = frames / block_count;                           // :198
  n += n % 2;                                                  // :199
  return n;
}

extern "C" wbx_status wbx_clip_build_mipmaps(wbx_ctx* c, uint32_t clip, int quality) {
  if (!c || clip >= c->clips.size() || !c->clips[clip].used) return WBX_ERR_INVALID;
  if (quality != 0 && quality != 1) return fail(c, WBX_ERR_INVALID, "quality: 0 (Low, int8) or 1 (High, int16)");
  ClipSlot& s = c->clips[clip];
  const int fmt = (int)s.d.format;
  if (fmt != FMT_I16 && fmt != FMT_I32 && fmt != FMT_F32) return fail(c, WBX_ERR_UNSUPPORTED, "mip-maps: clip format");
  (void)hipSetDevice(c->cfg.device);
  const uint32_t levels = wbx_mip_levels(s.d.count);
  if (levels > 24) return fail(c, WBX_ERR_UNSUPPORTED, "mip-maps: too many levels");
  const int bits = quality ? 16 : 8;
  const size_t esz = bits / 8;
  if (s.mip) {
    WBX_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(s.mip);
    s.mip = nullptr;
  }
  s.mip_off.assign(levels, 0);
  s.mip_count.assign(levels, 0);
  s.mip_bits = bits;
  size_t total = 0;
  for (uint32_t l = 0; l < levels; l++) {
    s.mip_count[l] = wbx_mip_data_count(s.d.count, l);
    s.mip_off[l] = total;
    total += align_up((size_t)s.mip_count[l] * s.d.channels * esz, 256);
  }
  if (!levels) return WBX_OK;
  const uint32_t tiles = (uint32_t)((s.d.count + kMipTile - 1) / kMipTile);
  const size_t nodes_off = total;
  const size_t nodes_per_ch = (size_t)tiles + tiles / 4 + 1;   // tile nodes + ping-pong scratch of the upper levels
  total += nodes_per_ch * s.d.channels * sizeof(MipNode);
  WBX_HIP(c, hipMalloc(&s.mip, total));
  for (uint32_t ch = 0; ch < s.d.channels; ch++) {
    MipArgs a{};
    a.src = (const char*)s.base + s.stride * ch;
    a.count = s.d.count;
    a.n_levels = levels;
    a.n_tiles = tiles;
    a.tile_nodes = (MipNode*)((char*)s.mip + nodes_off) + nodes_per_ch * ch;
    for (uint32_t l = 0; l < levels; l++) {
      a.level_out[l] = (char*)s.mip + s.mip_off[l] + (size_t)s.mip_count[l] * ch * esz;
      a.data_count[l] = s.mip_count[l];
    }
    launch_mip(a, fmt, bits, c->stream);
  }
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_mipmap_device(wbx_ctx* c, uint32_t clip, uint32_t level, const void** data, uint64_t* count) {
  if (!c || clip >= c->clips.size() || !c->clips[clip].used) return WBX_ERR_INVALID;
  const ClipSlot& s = c->clips[clip];
  if (!s.mip_bits || level >= s.mip_off.size()) return fail(c, WBX_ERR_INVALID, "no such mip level (call wbx_clip_build_mipmaps)");
  if (data) *data = (const char*)s.mip + s.mip_off[level];
  if (count) *count = s.mip_count[level];
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_fetch_mipmap(wbx_ctx* c, uint32_t clip, uint32_t level, void* dst) {
  const void* p = nullptr;
  uint64_t n = 0;
  wbx_status st = wbx_clip_mipmap_device(c, clip, level, &p, &n);
  if (st != WBX_OK) return st;
  if (!dst) return WBX_ERR_INVALID;
  const ClipSlot& s = c->clips[clip];
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  WBX_HIP(c, hipMemcpy(dst, p, (size_t)n * s.d.channels * (s.mip_bits / 8), hipMemcpyDeviceToHost));
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_free(wbx_ctx* c, uint32_t clip) {
  if (!c || clip >= c->clips.size() || !c->clips[clip].base) return WBX_ERR_INVALID;
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  (void)hipFree(c->clips[clip].base);
  if (c->clips[clip].mip) (void)hipFree(c->clips[clip].mip);
  c->clips[clip] = ClipSlot{};
  c->samples_dirty = true;
  return WBX_OK;
}

extern "C" wbx_status wbx_set_routing(wbx_ctx* c, uint32_t n_tracks, const int32_t* track_bus, uint32_t n_buses) {
  if (!c || n_tracks > c->cfg.max_tracks) return WBX_ERR_INVALID;
  if (track_bus && n_buses) {
    c->track_bus.assign(track_bus, track_bus + n_tracks);
    c->n_buses = n_buses;
  } else {
    c->track_bus.clear();
    c->n_buses = 0;
  }
  build_routing(c, n_tracks);
  c->buses_clean = false;
  return WBX_OK;
}

extern "C" wbx_status wbx_set_clamp(wbx_ctx* c, int on) {
  if (!c) return WBX_ERR_INVALID;
  c->clamp = on != 0;
  return WBX_OK;
}

extern "C" wbx_status wbx_set_master_target(wbx_ctx* c, void* device_buffer) {
  if (!c) return WBX_ERR_INVALID;
  c->master_target = (float*)device_buffer;
  return WBX_OK;
}

extern "C" wbx_status wbx_submit(wbx_ctx* c, uint32_t K, uint32_t N, const wbx_segment* segs, const uint32_t* seg_offsets,
                                 const float* gains) {
  if (!c || !seg_offsets || !gains || K == 0 || N == 0) return WBX_ERR_INVALID;
  if (K > c->cfg.max_blocks || N > c->cfg.max_tracks) return fail(c, WBX_ERR_INVALID, "K or N above the configured maximum");
  (void)hipSetDevice(c->cfg.device);
  WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
  wbx_status st = upload_tables(c, N);
  if (st != WBX_OK) return st;
  st = ensure_result_buffers(c, K, N);
  if (st != WBX_OK) return st;
  WBX_HIP(c, hipStreamSynchronize(c->stream));   // staging vectors are reused
  const uint32_t F = c->cfg.block_frames, C = c->cfg.channels;
  c->h_tb.assign((size_t)K * N, DTrackBlock{});
  c->h_rows.assign((size_t)K * N, DRow{});
  c->h_pool.clear();
  std::vector<uint32_t> gen_idx;
  uint32_t chunks = 0;
  for (size_t bt = 0; bt < (size_t)K * N; bt++) {
    DTrackBlock& tb = c->h_tb[bt];
    tb.g[0] = gains[bt * 2 + 0];
    tb.g[1] = gains[bt * 2 + (C > 1 ? 1 : 0)];
    const uint32_t s0 = seg_offsets[bt], s1 = seg_offsets[bt + 1];
    if (s1 < s0) return fail(c, WBX_ERR_INVALID, "seg_offsets not monotone");
    if (s1 - s0 > kMaxSegs) return fail(c, WBX_ERR_OVERFLOW, "more than 16 segments in one track-block");
    if (s1 != s0 && !segs) return WBX_ERR_INVALID;
    for (uint32_t i = s0; i < s1; i++) {
      const wbx_segment& sg = segs[i];
      if (sg.clip >= c->clips.size() || !c->clips[sg.clip].used) return fail(c, WBX_ERR_INVALID, "segment names an unknown clip");
      // the prologue of Sampler::stream (sampler.cpp:99-104) through the shared walker
      DTrackState ts{};
      ts.cur_type = EV_PLAY;
      ts.cur_sample = 0;
      ts.cur_gain = sg.gain;
      ts.playback_speed = sg.playback_speed;
      ts.sample_offset = sg.sample_offset;
      DSeg d{};
      BlockWalker w{};
      DTrackBlock scratch{};
      TrackCache tc{};
      tc.clip_idx = tc.smp_idx = 0xFFFFFFFFu;
      uint32_t pc = 0, stbits = 0;
      w.st = &ts;
      w.cache = &tc;
      w.samples = &c->clips[sg.clip].d;
      w.tb = &scratch;
      w.pool = &d;
      w.pool_count = &pc;
      w.pool_chunks = 0;
      w.status = &stbits;
      w.n_samples = F;
      w.n_channels = C;
      w.dst_rate = (double)c->cfg.sample_rate;
      w.start_sample = 0;
      w.nseg = 0;
      w.chunk = 0xFFFFFFFFu;
      w.stream(sg.num_samples, sg.buffer_offset);
      d = get_seg0(scratch);
      d.sample = sg.clip;
      const uint32_t k = i - s0;
      if (k == 0) {
        set_seg0(&tb, d);
      } else {
        if (k == 1) {
          tb.extra = chunks++;
          c->h_pool.resize((size_t)chunks * kChunk);
        }
        c->h_pool[(size_t)tb.extra * kChunk + (k - 1)] = d;
      }
    }
    tb.nseg = (uint8_t)(s1 - s0);
    tb.kind = classify(tb, F);
    if (tb.kind == KIND_GENERIC) gen_idx.push_back((uint32_t)bt);
    // host-sequenced plans use one template per track-block: row bt -> template bt, position inside the template
    DRow& row = c->h_rows[bt];
    row.pos = tb.pos;
    row.tmpl = tb.nseg ? (uint32_t)bt : 0xFFFFFFFFu;
    row.flags = tb.kind == KIND_SILENT ? ROW_SILENT : 0u;
  }
  st = ensure_gen_capacity(c, gen_idx.size());
  if (st != WBX_OK) return st;
  st = ensure_template_capacity(c, (size_t)K * N);
  if (st != WBX_OK) return st;
  {
    uint32_t counters[4] = {0u, 0u, (uint32_t)gen_idx.size(), (uint32_t)((size_t)K * N)};
    WBX_HIP(c, hipMemcpyAsync(PB(c).counters, counters, sizeof(counters), hipMemcpyHostToDevice, c->stream));
    if (!gen_idx.empty())
      WBX_HIP(c, hipMemcpyAsync(PB(c).gen_list.p, gen_idx.data(), gen_idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    WBX_HIP(c, hipStreamSynchronize(c->stream));   // counters / gen_idx are stack / local storage
  }
  if (chunks > PB(c).pool_chunks) {
    WBX_HIP(c, PB(c).pool.ensure((size_t)chunks * kChunk));
    PB(c).pool_chunks = chunks;
  }
  WBX_HIP(c, hipMemcpyAsync(PB(c).tmpl.p, c->h_tb.data(), c->h_tb.size() * sizeof(DTrackBlock), hipMemcpyHostToDevice, c->stream));
  WBX_HIP(c, hipMemcpyAsync(PB(c).prows.p, c->h_rows.data(), c->h_rows.size() * sizeof(DRow), hipMemcpyHostToDevice, c->stream));
  if (!c->h_pool.empty())
    WBX_HIP(c, hipMemcpyAsync(PB(c).pool.p, c->h_pool.data(), c->h_pool.size() * sizeof(DSeg), hipMemcpyHostToDevice, c->stream));
  st = launch_pre_render(c, K, c->stream);
  if (st != WBX_OK) return st;
  return launch_mix_sum(c, K, N);
}

extern "C" wbx_status wbx_sync(wbx_ctx* c) {
  if (!c) return WBX_ERR_INVALID;
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  return WBX_OK;
}

extern "C" wbx_status wbx_fetch(wbx_ctx* c, float* const* master_planar, float* peaks, float* buses) {
  if (!c) return WBX_ERR_INVALID;
  if (c->last_K == 0) return fail(c, WBX_ERR_FAILED, "nothing submitted");
  WBX_HIP(c, join_sum(c));
  const uint32_t K = c->last_K, N = c->last_N, C = c->cfg.channels, F = c->cfg.block_frames;
  if (master_planar) {
    // device [K][C][F] -> host planar[c][b*F + j]
    for (uint32_t ch = 0; ch < C; ch++)
      if (c->last_master_on_host) {   // Engine::process: the block is already on the host
        WBX_HIP(c, hipStreamSynchronize(c->stream));
        for (uint32_t b = 0; b < K; b++)
          std::memcpy(master_planar[ch] + (size_t)b * F, c->last_master + ((size_t)b * C + ch) * F, F * sizeof(float));
      } else {
        WBX_HIP(c, hipMemcpy2DAsync(master_planar[ch], F * sizeof(float), c->last_master + (size_t)ch * F,
                                    (size_t)C * F * sizeof(float), F * sizeof(float), K, hipMemcpyDeviceToHost, c->stream));
      }
  }
  if (peaks) WBX_HIP(c, hipMemcpyAsync(peaks, c->d_peaks.p, (size_t)K * N * C * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (buses) {
    if (!c->n_buses) return fail(c, WBX_ERR_INVALID, "no buses configured");
    WBX_HIP(c, hipMemcpyAsync(buses, c->last_buses, (size_t)K * c->n_buses * C * F * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  uint32_t pc[4] = {0, 0, 0, 0};
  WBX_HIP(c, hipMemcpy(pc, PB(c).counters, sizeof(pc), hipMemcpyDeviceToHost));
  if (pc[1] & 3u) return fail(c, WBX_ERR_OVERFLOW, "segment plan overflow (raise wbx_config.max_segments)");
  if (pc[1] & 8u) return fail(c, WBX_ERR_OVERFLOW, "more boundary / non-fp32 track-blocks than pre-render rows");
  if (pc[1] & 16u) return fail(c, WBX_ERR_OVERFLOW, "plan template array full");
  return WBX_OK;
}

extern "C" wbx_status wbx_fetch_interleaved(wbx_ctx* c, int out_format, void* dst) {
  if (!c || !dst) return WBX_ERR_INVALID;
  if (c->last_K == 0) return fail(c, WBX_ERR_FAILED, "nothing submitted");
  WBX_HIP(c, join_sum(c));
  size_t eb;
  switch (out_format) {
    case WBX_OUT_I16: eb = 2; break;
    case WBX_OUT_I24_X8:
    case WBX_OUT_I32:
    case WBX_OUT_F32: eb = 4; break;
    default: return fail(c, WBX_ERR_UNSUPPORTED, "interleaved output format");
  }
  const uint32_t K = c->last_K, C = c->cfg.channels, F = c->cfg.block_frames;
  const size_t bytes = (size_t)K * F * C * eb;
  WBX_HIP(c, c->d_conv.ensure(bytes));
  launch_convert(c->last_master, c->d_conv.p, K, F, C, out_format, c->stream);   // pinned staging is device-readable too
  WBX_HIP(c, hipMemcpyAsync(dst, c->d_conv.p, bytes, hipMemcpyDeviceToHost, c->stream));
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  return WBX_OK;
}

extern "C" wbx_status wbx_partial_master(wbx_ctx* c, void** device_ptr, size_t* n_floats) {
  if (!c || !device_ptr) return WBX_ERR_INVALID;
  if (c->last_K == 0) return fail(c, WBX_ERR_FAILED, "nothing submitted");
  WBX_HIP(c, join_sum(c));
  *device_ptr = c->last_master;
  if (n_floats) *n_floats = (size_t)c->last_K * c->cfg.channels * c->cfg.block_frames;
  return WBX_OK;
}

extern "C" wbx_status wbx_finalize_master(wbx_ctx* c, void* device_partial, uint32_t n_blocks, int clamp, void* stream) {
  if (!c || !device_partial || n_blocks == 0) return WBX_ERR_INVALID;
  hipStream_t on = stream ? (hipStream_t)stream : c->stream;
  if (c->sum_pending >= 0) WBX_HIP(c, hipStreamWaitEvent(on, c->sum_done[c->sum_pending], 0));
  if (clamp) launch_clamp((float*)device_partial, (size_t)n_blocks * c->cfg.channels * c->cfg.block_frames, on);
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

extern "C" wbx_status wbx_finalize_master_into(wbx_ctx* c, const void* device_partial, void* dst, uint32_t n_blocks, int clamp,
                                               void* stream) {
  if (!c || !device_partial || !dst || n_blocks == 0) return WBX_ERR_INVALID;
  if (((uintptr_t)device_partial | (uintptr_t)dst) & 15u) return fail(c, WBX_ERR_INVALID, "finalize: buffers must be 16-byte aligned");
  hipStream_t on = stream ? (hipStream_t)stream : c->stream;
  if (c->sum_pending >= 0) WBX_HIP(c, hipStreamWaitEvent(on, c->sum_done[c->sum_pending], 0));
  launch_clamp_into((const float*)device_partial, (float*)dst, (size_t)n_blocks * c->cfg.channels * c->cfg.block_frames, clamp, on);
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

extern "C" wbx_status wbx_kernel_time(wbx_ctx* c, int reset, double* mix_ms_avg, uint64_t* mix_launches) {
  if (!c) return WBX_ERR_INVALID;
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  if (mix_ms_avg) *mix_ms_avg = c->mix_launches ? c->mix_ms_total / (double)c->mix_launches : 0.0;
  if (mix_launches) *mix_launches = c->mix_launches;
  if (reset) {
    c->mix_ms_total = 0.0;
    c->tail_ms_total = 0.0;
    c->mix_launches = 0;
  }
  return WBX_OK;
}

extern "C" wbx_status wbx_tail_time(wbx_ctx* c, double* tail_ms_avg) {
  if (!c || !tail_ms_avg) return WBX_ERR_INVALID;
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  *tail_ms_avg = c->mix_launches ? c->tail_ms_total / (double)c->mix_launches : 0.0;
  return WBX_OK;
}

// =================================================================================================
// layer 2: the engine surface
// =================================================================================================
namespace {

enum : uint32_t { PARAM_VOLUME = 0, PARAM_PAN = 1, PARAM_MUTE = 2 };   // reference TrackParameter, track.h:29-34

struct ParamMsg {
  uint32_t id;
  double value;
};

struct HostTrack {
  std::vector<HostClip> clips;         // sorted by min_time (Track::update_clip_ordering, track.cpp:159-180)
  float volume = 0.0f, pan = 0.0f, pan_coeffs[2] = {0.0f, 0.0f};
  bool mute = false;
  bool ui_solo = false;                 // ui_parameter_state.solo (track.h:52)
  std::vector<ParamMsg> msgs;          // TrackMessage::ParamChange ring (track.h:131), drained at the next block
  int32_t bus = -1;
  DPatch patch{};
};

// math::db_to_linear<float>, reference core/core_math.h:83-89
float db_to_linear(float x) {
  if (x <= -72.0f) return 0.0f;
  return std::pow(10.0f, (float)((double)x * 0.05));
}

// calculate_panning_coefs(p, ConstantPower_3db), reference core/panning_law.cpp:9-32
void pan_constant_power_3db(float p, float* l, float* r) {
  double x = 0.5 * ((double)p + 1.0);
  double left = std::sin(0.5 * std::numbers::pi * (1.0 - x));
  double right = std::sin(0.5 * std::numbers::pi * x);
  double boost = std::sqrt(2.0);
  *l = (float)(left * boost);
  *r = (float)(right * boost);
}

}  // namespace

struct wbx_engine {
  wbx_ctx* ctx = nullptr;
  std::string err;
  std::vector<HostTrack> tracks;
  uint32_t n_buses = 0;
  double ppq = 96.0;                    // engine.h:43
  double playhead = 0.0, playhead_start = 0.0, sample_position = 0.0, beat_duration = 0.5;
  bool playing = false;
  bool clips_dirty = true, gains_dirty = true, routing_dirty = true, patches_pending = false;
  bool any_slow_clip = false;           // a clip the mix kernel cannot stream directly (speed > 4096)
  bool any_window_clip = false;         // a clip that is linearly resampled (playback speed != 1)
  bool any_stride_clip = false;         // a clip read with per-frame taps: fp32 played faster than recorded (speed > 0.999, != 1), resampled integer PCM
  size_t total_clips = 0;
  uint32_t next_clip_uid = 0;
  // Engine::process (one block per call): pinned, device-mapped host staging the sum kernel writes the block into
  // and the plan status lands in — the callback path then needs no copy-engine transfer at all
  DPatch* h_patch[kRing] = {};          // pinned patch buffers the plan kernel reads in place
  uint32_t patch_cap[kRing] = {};
  hipEvent_t patch_done[kRing] = {};    // the plan kernel that read the buffer
  bool patch_valid[kRing] = {};
  uint32_t patch_seq = 0;
  float* h_block = nullptr;             // [C][F]
  uint32_t* h_status = nullptr;         // plan counters [4]
  size_t d_clips_count = 0;
  bool clips_uploaded = false;          // the device holds a clip table (its internal_state_changed flags are live)
  bool clips_edited = false;            // a clip list changed since the previous plan (PlanArgs::clips_changed)
  uint32_t state_tracks = 0;            // tracks that have device state

  DevBuf<DClip> d_clips;
  DevBuf<uint32_t> d_clip_first;
  DevBuf<DTrackState> d_state;
  DevBuf<DPatch> d_patch;
  DevBuf<float> d_gains, d_levels;
};

namespace {

wbx_status efail(wbx_engine* e, wbx_status s, const char* what) {
  if (e) e->err = what;
  return s;
}

#define WBX_EHIP(e, call)                                      \
  do {                                                         \
    hipError_t _e = (call);                                    \
    if (_e != hipSuccess) {                                    \
      (e)->err = std::string(#call) + ": " + hipGetErrorString(_e); \
      return WBX_ERR_DEVICE;                                   \
    }                                                          \
  } while (0)

// Track::find_next_clip over the host copy (track.cpp:182-213)
bool host_find_next_clip(const HostTrack& t, double time_pos, uint32_t* idx) {
  if (t.clips.empty()) return false;
  if (t.clips.back().d.max_time < time_pos) return false;
  *idx = edit::lower_bound_max(t.clips, time_pos);
  return true;
}

// Track::reset_playback_state, track.cpp:220-232
void reset_playback_state(wbx_engine* e, HostTrack& t, double time_pos, bool refresh_voices) {
  if (!refresh_voices) {
    uint32_t idx = 0;
    bool has = host_find_next_clip(t, time_pos, &idx);
    t.patch.flags |= PATCH_CLIPIDX;
    t.patch.has_clip_idx = has ? 1u : 0u;
    t.patch.clip_idx = idx;
  }
  t.patch.flags |= PATCH_REFRESH;
  t.patch.refresh_voice = refresh_voices ? 1u : 0u;
  e->patches_pending = true;
}

// bookkeeping shared by every clip-list edit: which clips the hot loop can stream directly
void note_clip(wbx_engine* e, const DClip& c);

// after a clip-list edit: Track::update_clip_ordering + reset_playback_state(playhead, true)
void finish_edit(wbx_engine* e, HostTrack& t);

}  // namespace

extern "C" wbx_status wbx_engine_create(const wbx_config* cfg, wbx_engine** out) {
  if (!cfg || !out) return WBX_ERR_INVALID;
  *out = nullptr;
  wbx_ctx* c = nullptr;
  wbx_status st = wbx_create(cfg, &c);
  if (st != WBX_OK) return st;
  wbx_engine* e = new (std::nothrow) wbx_engine();
  if (!e) {
    wbx_destroy(c);
    return WBX_ERR_OOM;
  }
  e->ctx = c;
  *out = e;
  return WBX_OK;
}

extern "C" void wbx_engine_destroy(wbx_engine* e) {
  if (!e) return;
  if (e->ctx) {
    (void)hipSetDevice(e->ctx->cfg.device);
    (void)hipStreamSynchronize(e->ctx->stream);
  }
  e->d_clips.release();
  e->d_clip_first.release();
  e->d_state.release();
  e->d_patch.release();
  e->d_gains.release();
  e->d_levels.release();
  for (int i = 0; i < kRing; i++) {
    if (e->h_patch[i]) (void)hipHostFree(e->h_patch[i]);
    if (e->patch_done[i]) (void)hipEventDestroy(e->patch_done[i]);
  }
  if (e->h_block) (void)hipHostFree(e->h_block);
  if (e->h_status) (void)hipHostFree(e->h_status);
  wbx_destroy(e->ctx);
  delete e;
}

extern "C" const char* wbx_engine_last_error(const wbx_engine* e) {
  if (!e) return "null engine";
  if (!e->err.empty()) return e->err.c_str();
  return wbx_last_error(e->ctx);
}

extern "C" wbx_ctx* wbx_engine_ctx(wbx_engine* e) { return e ? e->ctx : nullptr; }

extern "C" wbx_status wbx_engine_set_bpm(wbx_engine* e, double bpm) {   // engine.cpp:24-30
  if (!e || !(bpm > 0.0)) return WBX_ERR_INVALID;
  e->beat_duration = 60.0 / bpm;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_set_playhead_position(wbx_engine* e, double beat) {   // engine.cpp:32-41
  if (!e) return WBX_ERR_INVALID;
  e->playhead_start = beat;
  e->playhead = beat;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_add_track(wbx_engine* e, uint32_t* track_out) {   // engine.cpp:200-208, Track::Track track.cpp:22-27
  if (!e) return WBX_ERR_INVALID;
  if (e->tracks.size() >= e->ctx->cfg.max_tracks) return efail(e, WBX_ERR_INVALID, "max_tracks reached");
  e->tracks.emplace_back();
  const uint32_t t = (uint32_t)e->tracks.size() - 1;
  wbx_track_set_volume(e, t, 0.0f);
  wbx_track_set_pan(e, t, 0.0f);
  wbx_track_set_mute(e, t, 0);
  e->clips_dirty = e->routing_dirty = e->gains_dirty = true;
  if (track_out) *track_out = t;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_set_buses(wbx_engine* e, uint32_t n_buses) {
  if (!e) return WBX_ERR_INVALID;
  e->n_buses = n_buses;
  e->routing_dirty = true;
  return WBX_OK;
}

extern "C" wbx_status wbx_track_set_volume(wbx_engine* e, uint32_t t, float db) {   // track.cpp:47-57
  if (!e || t >= e->tracks.size()) return WBX_ERR_INVALID;
  e->tracks[t].msgs.push_back({PARAM_VOLUME, (double)db_to_linear(db)});
  return WBX_OK;
}

extern "C" wbx_status wbx_track_set_pan(wbx_engine* e, uint32_t t, float pan) {   // track.cpp:59-68
  if (!e || t >= e->tracks.size()) return WBX_ERR_INVALID;
  e->tracks[t].msgs.push_back({PARAM_PAN, (double)pan});
  return WBX_OK;
}

extern "C" wbx_status wbx_track_set_mute(wbx_engine* e, uint32_t t, int mute) {   // track.cpp:70-79
  if (!e || t >= e->tracks.size()) return WBX_ERR_INVALID;
  e->tracks[t].msgs.push_back({PARAM_MUTE, (double)(mute ? 1 : 0)});
  return WBX_OK;
}

namespace {

// new track i = old track order[i] (order.size() = new track count): the per-track device state (sequencer,
// sampler, running levels) follows its Track object, as the pointers in the reference's vector do
wbx_status permute_tracks(wbx_engine* e, const std::vector<uint32_t>& order) {
  wbx_ctx* c = e->ctx;
  (void)hipSetDevice(c->cfg.device);
  WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
  WBX_EHIP(e, join_sum(c));
  WBX_EHIP(e, hipStreamSynchronize(c->stream));
  const uint32_t old_n = (uint32_t)e->tracks.size(), new_n = (uint32_t)order.size();
  if (e->state_tracks) {
    const uint32_t C = c->cfg.channels;
    std::vector<DTrackState> st(e->state_tracks), st2(std::max<size_t>(new_n, 1));
    std::vector<float> lv((size_t)e->state_tracks * C), lv2((size_t)std::max<uint32_t>(new_n, 1) * C, 0.0f);
    WBX_EHIP(e, hipMemcpy(st.data(), e->d_state.p, st.size() * sizeof(DTrackState), hipMemcpyDeviceToHost));
    WBX_EHIP(e, hipMemcpy(lv.data(), e->d_levels.p, lv.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < new_n; i++) {
      if (order[i] < e->state_tracks) {
        st2[i] = st[order[i]];
        for (uint32_t ch = 0; ch < C; ch++) lv2[(size_t)i * C + ch] = lv[(size_t)order[i] * C + ch];
      } else {
        st2[i] = DTrackState{};   // a track added since the last render
      }
    }
    WBX_EHIP(e, hipMemset(e->d_state.p, 0, e->d_state.cap * sizeof(DTrackState)));
    WBX_EHIP(e, hipMemset(e->d_levels.p, 0, e->d_levels.cap * sizeof(float)));
    if (new_n) {
      WBX_EHIP(e, hipMemcpy(e->d_state.p, st2.data(), (size_t)new_n * sizeof(DTrackState), hipMemcpyHostToDevice));
      WBX_EHIP(e, hipMemcpy(e->d_levels.p, lv2.data(), (size_t)new_n * C * sizeof(float), hipMemcpyHostToDevice));
    }
    e->state_tracks = new_n;
  }
  std::vector<HostTrack> moved(new_n);
  for (uint32_t i = 0; i < new_n; i++) moved[i] = std::move(e->tracks[order[i]]);
  e->tracks = std::move(moved);
  (void)old_n;
  e->clips_dirty = e->gains_dirty = e->routing_dirty = true;
  e->total_clips = 0;
  for (auto& tr : e->tracks) e->total_clips += tr.clips.size();
  return WBX_OK;
}

}  // namespace

extern "C" wbx_status wbx_engine_delete_track(wbx_engine* e, uint32_t slot) {   // engine.cpp:210-218
  if (!e || slot >= e->tracks.size()) return WBX_ERR_INVALID;
  std::vector<uint32_t> order;
  for (uint32_t i = 0; i < e->tracks.size(); i++)
    if (i != slot) order.push_back(i);
  return permute_tracks(e, order);
}

extern "C" wbx_status wbx_engine_clear_all(wbx_engine* e) {   // engine.cpp:59-66: every track goes
  if (!e) return WBX_ERR_INVALID;
  return permute_tracks(e, std::vector<uint32_t>{});
}

extern "C" wbx_status wbx_engine_move_track(wbx_engine* e, uint32_t from_slot, uint32_t to_slot) {   // engine.cpp:228-243
  if (!e || from_slot >= e->tracks.size() || to_slot >= e->tracks.size()) return WBX_ERR_INVALID;
  if (from_slot == to_slot) return WBX_OK;
  std::vector<uint32_t> order(e->tracks.size());
  for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
  order.erase(order.begin() + from_slot);
  order.insert(order.begin() + to_slot, from_slot);
  return permute_tracks(e, order);
}

extern "C" wbx_status wbx_engine_solo_track(wbx_engine* e, uint32_t slot) {   // engine.cpp:245-262
  if (!e || slot >= e->tracks.size()) return WBX_ERR_INVALID;
  bool mute = false;
  if (e->tracks[slot].ui_solo) {
    e->tracks[slot].ui_solo = false;
  } else {
    e->tracks[slot].ui_solo = true;
    wbx_track_set_mute(e, slot, 0);
    mute = true;
  }
  for (uint32_t i = 0; i < e->tracks.size(); i++) {
    if (i == slot) continue;
    e->tracks[i].ui_solo = false;
    wbx_track_set_mute(e, i, mute ? 1 : 0);
  }
  return WBX_OK;
}

extern "C" wbx_status wbx_track_set_bus(wbx_engine* e, uint32_t t, int32_t bus) {
  if (!e || t >= e->tracks.size()) return WBX_ERR_INVALID;
  e->tracks[t].bus = bus;
  e->routing_dirty = true;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_add_sample(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate,
                                            uint64_t frames, const void* const* planar, uint32_t* sample_out) {
  if (!e || !sample_out) return WBX_ERR_INVALID;
  const uint32_t id = (uint32_t)e->ctx->clips.size();
  wbx_status st = wbx_clip_upload(e->ctx, id, format, channels, sample_rate, frames, planar);
  if (st == WBX_OK) *sample_out = id;
  return st;
}

extern "C" wbx_status wbx_engine_add_sample_interleaved(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate,
                                                        uint64_t frames, const void* interleaved, uint32_t* sample_out) {
  if (!e || !sample_out) return WBX_ERR_INVALID;
  const uint32_t id = (uint32_t)e->ctx->clips.size();
  wbx_status st = wbx_clip_upload_interleaved(e->ctx, id, format, channels, sample_rate, frames, interleaved);
  if (st == WBX_OK) *sample_out = id;
  return st;
}

extern "C" wbx_status wbx_engine_add_sample_synth(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate,
                                                  uint64_t frames, uint64_t seed, uint32_t key_track, float amp,
                                                  uint32_t* sample_out) {
  if (!e || !sample_out) return WBX_ERR_INVALID;
  const uint32_t id = (uint32_t)e->ctx->clips.size();
  wbx_status st = wbx_clip_synth(e->ctx, id, format, channels, sample_rate, frames, seed, key_track, amp);
  if (st == WBX_OK) *sample_out = id;
  return st;
}

namespace {

void note_clip(wbx_engine* e, const DClip& c) {
  const DSample& smp = e->ctx->clips[c.sample].d;
  const double ps = ((double)smp.sample_rate / (double)e->ctx->cfg.sample_rate) * c.speed;   // sampler.h:24
  // every block of such a clip goes through the pre-render pass (the hot loop takes playback speeds up to 4096)
  if (!(ps > 0.0 && ps <= 4096.0)) e->any_slow_clip = true;
  if (ps != 1.0) e->any_window_clip = true;
  if ((ps > 0.999 || smp.format != FMT_F32) && ps != 1.0) e->any_stride_clip = true;
}

void finish_edit(wbx_engine* e, HostTrack& t) {
  edit::update_clip_ordering(t.clips);
  reset_playback_state(e, t, e->playhead, true);   // engine.cpp:360,395,405,416,426,437,449,459,473
  e->clips_dirty = true;
  e->clips_edited = true;
  e->total_clips = 0;
  for (auto& tr : e->tracks) e->total_clips += tr.clips.size();
}

double rate_of_sample(const wbx_engine* e, uint32_t sample) { return (double)e->ctx->clips[sample].d.sample_rate; }

}  // namespace

// Engine::add_audio_clip -> add_to_cliplist, engine.cpp:293-309, :409-461.  A clip that overlaps existing ones
// trims, splits or deletes them through reserve_track_region (engine.cpp:478-569), as the reference does.
extern "C" wbx_status wbx_engine_add_audio_clip(wbx_engine* e, uint32_t track, double min_time, double max_time,
                                                double start_offset, uint32_t sample, double speed, float gain) {
  if (!e || track >= e->tracks.size()) return WBX_ERR_INVALID;
  if (sample >= e->ctx->clips.size() || !e->ctx->clips[sample].used) return efail(e, WBX_ERR_INVALID, "unknown sample");
  if (!(min_time <= max_time)) return efail(e, WBX_ERR_INVALID, "min_time > max_time");
  HostTrack& t = e->tracks[track];
  const bool empty = t.clips.empty();
  const bool back = !empty && t.clips.back().d.max_time < min_time;
  const bool front = !empty && !back && t.clips.front().d.min_time > max_time;
  ClipQuery q{};
  if (!empty && !back && !front && edit::query_clip_by_range(t.clips, min_time, max_time, &q))
    edit::reserve_track_region(t.clips, q.first, q.last, min_time, max_time, 0u, e->beat_duration,
                               [&](uint32_t smp) { return rate_of_sample(e, smp); }, &e->next_clip_uid);
  HostClip c{};
  c.d.min_time = min_time;
  c.d.max_time = max_time;
  c.d.start_offset = start_offset;
  c.d.speed = speed;
  c.d.gain = gain;
  c.d.sample = sample;
  c.d.internal_state_changed = 0;
  c.d.uid = ++e->next_clip_uid;
  t.clips.push_back(c);
  note_clip(e, c.d);
  finish_edit(e, t);
  return WBX_OK;
}

// Engine::move_clip, engine.cpp:346-363
extern "C" wbx_status wbx_engine_move_clip(wbx_engine* e, uint32_t track, uint32_t clip, double relative_pos) {
  if (!e || track >= e->tracks.size() || clip >= e->tracks[track].clips.size()) return WBX_ERR_INVALID;
  if (relative_pos == 0.0) return WBX_OK;
  HostTrack& t = e->tracks[track];
  const uint32_t uid = t.clips[clip].d.uid;
  double mn, mx;
  edit::calc_move_clip(t.clips[clip].d.min_time, t.clips[clip].d.max_time, relative_pos, 0.0, &mn, &mx);
  ClipQuery q{};
  if (edit::query_clip_by_range(t.clips, mn, mx, &q))
    edit::reserve_track_region(t.clips, q.first, q.last, mn, mx, uid, e->beat_duration,
                               [&](uint32_t smp) { return rate_of_sample(e, smp); }, &e->next_clip_uid);
  for (auto& c : t.clips)
    if (c.d.uid == uid) {
      c.d.min_time = mn;
      c.d.max_time = mx;
      c.d.internal_state_changed = 1;
      c.flag_dirty = true;
    }
  finish_edit(e, t);
  return WBX_OK;
}

// Engine::resize_clip, engine.cpp:365-398
extern "C" wbx_status wbx_engine_resize_clip(wbx_engine* e, uint32_t track, uint32_t clip, double relative_pos,
                                             double resize_limit, double min_length, int left_side, int shift,
                                             int stretch) {
  if (!e || track >= e->tracks.size() || clip >= e->tracks[track].clips.size()) return WBX_ERR_INVALID;
  if (relative_pos == 0.0) return WBX_OK;
  HostTrack& t = e->tracks[track];
  const DClip c0 = t.clips[clip].d;
  const DSample& smp = e->ctx->clips[c0.sample].d;
  const edit::ResizeResult r =
      edit::calc_resize_clip(c0.min_time, c0.max_time, c0.start_offset, c0.speed, (double)smp.sample_rate, (double)smp.count,
                             relative_pos, resize_limit, min_length, c0.min_time, e->beat_duration, left_side != 0,
                             shift != 0, stretch != 0, false);
  ClipQuery q{};
  if (edit::query_clip_by_range(t.clips, r.min, r.max, &q))
    edit::reserve_track_region(t.clips, q.first, q.last, r.min, r.max, c0.uid, e->beat_duration,
                               [&](uint32_t s2) { return rate_of_sample(e, s2); }, &e->next_clip_uid);
  for (auto& c : t.clips)
    if (c.d.uid == c0.uid) {
      if (left_side)
        c.d.min_time = r.min;
      else
        c.d.max_time = r.max;
      c.d.start_offset = r.start_offset;
      if (stretch) c.d.speed = r.speed;
      c.d.internal_state_changed = (shift || stretch) ? 1u : 0u;
      c.flag_dirty = true;
      note_clip(e, c.d);
    }
  finish_edit(e, t);
  return WBX_OK;
}

// Engine::delete_clip, engine.cpp:400-407
extern "C" wbx_status wbx_engine_delete_clip(wbx_engine* e, uint32_t track, uint32_t clip) {
  if (!e || track >= e->tracks.size() || clip >= e->tracks[track].clips.size()) return WBX_ERR_INVALID;
  HostTrack& t = e->tracks[track];
  t.clips[clip].deleted = true;
  finish_edit(e, t);
  return WBX_OK;
}

// Engine::delete_region, engine.cpp:463-475
extern "C" wbx_status wbx_engine_delete_region(wbx_engine* e, uint32_t track, double min, double max) {
  if (!e || track >= e->tracks.size() || !(min <= max)) return WBX_ERR_INVALID;
  HostTrack& t = e->tracks[track];
  ClipQuery q{};
  if (!edit::query_clip_by_range(t.clips, min, max, &q)) return WBX_OK;
  edit::reserve_track_region(t.clips, q.first, q.last, min, max, 0u, e->beat_duration,
                             [&](uint32_t smp) { return rate_of_sample(e, smp); }, &e->next_clip_uid);
  finish_edit(e, t);
  return WBX_OK;
}

// Engine::set_clip_gain, engine.cpp:1460-1464
extern "C" wbx_status wbx_engine_set_clip_gain(wbx_engine* e, uint32_t track, uint32_t clip, float gain) {
  if (!e || track >= e->tracks.size() || clip >= e->tracks[track].clips.size()) return WBX_ERR_INVALID;
  e->tracks[track].clips[clip].d.gain = gain;
  e->clips_dirty = true;
  e->clips_edited = true;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_clip_count(wbx_engine* e, uint32_t track, uint32_t* count) {
  if (!e || track >= e->tracks.size() || !count) return WBX_ERR_INVALID;
  *count = (uint32_t)e->tracks[track].clips.size();
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_get_clip(wbx_engine* e, uint32_t track, uint32_t clip, wbx_clip_info* out) {
  if (!e || track >= e->tracks.size() || clip >= e->tracks[track].clips.size() || !out) return WBX_ERR_INVALID;
  const DClip& d = e->tracks[track].clips[clip].d;
  out->min_time = d.min_time;
  out->max_time = d.max_time;
  out->start_offset = d.start_offset;
  out->speed = d.speed;
  out->gain = d.gain;
  out->sample = d.sample;
  return WBX_OK;
}

// the clip placement arithmetic on its own (engine/clip_edit.h:10-150), for hosts that preview an edit
extern "C" void wbx_calc_move_clip(double clip_min, double clip_max, double relative_pos, double min_move, double* new_min,
                                   double* new_max) {
  edit::calc_move_clip(clip_min, clip_max, relative_pos, min_move, new_min, new_max);
}

extern "C" void wbx_calc_resize_clip(double clip_min, double clip_max, double clip_start_offset, double clip_speed,
                                     double sample_rate, double sample_count, double relative_pos, double resize_limit,
                                     double min_length, double min_resize_pos, double beat_duration, int is_min, int shift,
                                     int stretch, int clamp_at_resize_pos, double* out_min, double* out_max,
                                     double* out_start_offset, double* out_speed) {
  const edit::ResizeResult r = edit::calc_resize_clip(clip_min, clip_max, clip_start_offset, clip_speed, sample_rate,
                                                      sample_count, relative_pos, resize_limit, min_length, min_resize_pos,
                                                      beat_duration, is_min != 0, shift != 0, stretch != 0,
                                                      clamp_at_resize_pos != 0);
  *out_min = r.min;
  *out_max = r.max;
  *out_start_offset = r.start_offset;
  *out_speed = r.speed;
}

extern "C" double wbx_calc_clip_shift(double start_offset, double relative_pos, double beat_duration, double sample_rate) {
  return edit::calc_clip_shift(start_offset, relative_pos, beat_duration, sample_rate);
}

extern "C" double wbx_shift_clip_content(double start_offset, double speed, double sample_rate, double relative_pos,
                                         double beat_duration) {
  return edit::shift_clip_content(start_offset, speed, sample_rate, relative_pos, beat_duration);
}

extern "C" wbx_status wbx_engine_play(wbx_engine* e) {   // engine.cpp:68-80
  if (!e) return WBX_ERR_INVALID;
  for (auto& t : e->tracks) reset_playback_state(e, t, e->playhead_start, false);
  e->sample_position = 0;
  e->playing = true;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_stop(wbx_engine* e) {   // engine.cpp:82-93, Track::stop track.cpp:249-256
  if (!e) return WBX_ERR_INVALID;
  e->playing = false;
  e->playhead = e->playhead_start;
  for (auto& t : e->tracks) t.patch.flags |= PATCH_STOP;
  e->patches_pending = true;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_render(wbx_engine* e, uint32_t K) {
  if (!e || K == 0) return WBX_ERR_INVALID;
  wbx_ctx* c = e->ctx;
  e->err.clear();
  if (K > c->cfg.max_blocks) return efail(e, WBX_ERR_INVALID, "n_blocks above wbx_config.max_blocks");
  const uint32_t N = (uint32_t)e->tracks.size();
  (void)hipSetDevice(c->cfg.device);
  const uint32_t C = c->cfg.channels, F = c->cfg.block_frames;
  hipStream_t s = c->stream;
  if (N == 0) {
    // Engine::process with an empty track list: output_buffer.clear() and the transport advance (engine.cpp:1598,
    // :1619-1623) — silence
    WBX_EHIP(e, join_sum(c));
    WBX_EHIP(e, c->d_master.ensure((size_t)K * C * F));
    float* master = c->master_target ? c->master_target : c->d_master.p;
    WBX_EHIP(e, hipMemsetAsync(master, 0, (size_t)K * C * F * sizeof(float), s));
    c->last_master = master;
    c->last_master_on_host = false;
    c->last_K = K;
    c->last_N = 0;
    const double sample_rate = (double)c->cfg.sample_rate;
    for (uint32_t b = 0; b < K; b++) {
      const double buffer_duration_in_beats = ((double)F / sample_rate) / e->beat_duration;
      if (e->playing) {
        e->sample_position += beat_to_samples(buffer_duration_in_beats, sample_rate, e->beat_duration);
        e->playhead = e->playhead + buffer_duration_in_beats;
      }
    }
    return WBX_OK;
  }

  // -- parameters: drain the message rings (process_track_messages track.cpp:773-779) and apply them
  //    (track.cpp:618-643); the factor used per sample is fl(volume * pan_coeffs[c]) (track.cpp:728-731)
  for (auto& t : e->tracks) {
    if (t.msgs.empty()) continue;
    for (const ParamMsg& m : t.msgs) {
      switch (m.id) {
        case PARAM_VOLUME: t.volume = (float)m.value; break;
        case PARAM_PAN:
          t.pan = (float)m.value;
          pan_constant_power_3db(t.pan, &t.pan_coeffs[0], &t.pan_coeffs[1]);
          break;
        case PARAM_MUTE: t.mute = m.value > 0.0; break;
        default: break;
      }
    }
    t.msgs.clear();
    e->gains_dirty = true;
  }
  if (e->gains_dirty) {
    std::vector<float> g((size_t)N * 2);
    for (uint32_t t = 0; t < N; t++) {
      const HostTrack& tr = e->tracks[t];
      float volume = tr.mute ? 0.0f : tr.volume;
      g[2 * t + 0] = volume * tr.pan_coeffs[0];
      g[2 * t + 1] = volume * tr.pan_coeffs[1];
    }
    WBX_EHIP(e, e->d_gains.ensure(g.size()));
    WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
    WBX_EHIP(e, hipStreamSynchronize(s));
    WBX_EHIP(e, hipMemcpy(e->d_gains.p, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice));
    e->gains_dirty = false;
  }

  // -- clip lists
  if (e->clips_dirty) {
    WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
    WBX_EHIP(e, hipStreamSynchronize(s));
    // Clip::internal_state_changed is cleared by the sequencer on the device (track.cpp:373,392,418): before
    // the table is replaced, take the live flags back for every clip no edit has touched since the last upload
    if (e->clips_uploaded && e->d_clips_count) {
      std::vector<DClip> live(e->d_clips_count);
      WBX_EHIP(e, hipMemcpy(live.data(), e->d_clips.p, live.size() * sizeof(DClip), hipMemcpyDeviceToHost));
      std::vector<uint32_t> flag(e->next_clip_uid + 1, 2u);
      for (const DClip& d : live)
        if (d.uid < flag.size()) flag[d.uid] = d.internal_state_changed;
      for (auto& tr : e->tracks)
        for (auto& hc : tr.clips)
          if (!hc.flag_dirty && hc.d.uid < flag.size() && flag[hc.d.uid] != 2u) hc.d.internal_state_changed = flag[hc.d.uid];
    }
    std::vector<uint32_t> first(N + 1, 0);
    std::vector<DClip> flat;
    for (uint32_t t = 0; t < N; t++) {
      first[t] = (uint32_t)flat.size();
      for (auto& hc : e->tracks[t].clips) {
        flat.push_back(hc.d);
        hc.flag_dirty = false;
      }
    }
    first[N] = (uint32_t)flat.size();
    WBX_EHIP(e, e->d_clips.ensure(std::max<size_t>(1, flat.size())));
    WBX_EHIP(e, e->d_clip_first.ensure(N + 1));
    e->d_clips_count = flat.size();
    e->clips_uploaded = true;
    if (!flat.empty()) WBX_EHIP(e, hipMemcpy(e->d_clips.p, flat.data(), flat.size() * sizeof(DClip), hipMemcpyHostToDevice));
    WBX_EHIP(e, hipMemcpy(e->d_clip_first.p, first.data(), first.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    e->clips_dirty = false;
  }

  // -- per-track device state for tracks added since the last render
  if (e->state_tracks < N) {
    DevBuf<DTrackState> grown;
    WBX_EHIP(e, grown.ensure(std::max<size_t>(N, c->cfg.max_tracks)));
    WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
    WBX_EHIP(e, hipStreamSynchronize(s));
    WBX_EHIP(e, hipMemset(grown.p, 0, grown.cap * sizeof(DTrackState)));
    if (e->state_tracks) WBX_EHIP(e, hipMemcpy(grown.p, e->d_state.p, e->state_tracks * sizeof(DTrackState), hipMemcpyDeviceToDevice));
    e->d_state.release();
    e->d_state = grown;
    WBX_EHIP(e, e->d_levels.ensure((size_t)c->cfg.max_tracks * 2));
    if (e->state_tracks == 0) WBX_EHIP(e, hipMemset(e->d_levels.p, 0, e->d_levels.cap * sizeof(float)));
    e->state_tracks = N;
  }

  // -- pending state edits (play / stop / clip-list changes)
  // The patches sit in pinned host memory that the plan kernel reads directly (one 16-B read per lane): no copy, and
  // above all no stream synchronisation — a drain here would empty the queue of renders the host has run ahead by.
  // Three buffers in rotation; a buffer is refilled only after the plan kernel that last read it has finished.
  const DPatch* d_patch = nullptr;
  int patch_slot = -1;
  if (e->patches_pending) {
    patch_slot = (int)(e->patch_seq++ % kRing);
    if (e->patch_cap[patch_slot] < N) {
      // (re)allocate all three at once — pinning memory synchronises the device, so it must not happen in mid-run
      WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
      WBX_EHIP(e, hipStreamSynchronize(s));
      const uint32_t cap = std::max<uint32_t>(N, c->cfg.max_tracks);
      for (int i = 0; i < kRing; i++) {
        if (e->h_patch[i]) WBX_EHIP(e, hipHostFree(e->h_patch[i]));
        e->h_patch[i] = nullptr;
        WBX_EHIP(e, hipHostMalloc((void**)&e->h_patch[i], (size_t)cap * sizeof(DPatch), hipHostMallocDefault));
        e->patch_cap[i] = cap;
        e->patch_valid[i] = false;
        if (!e->patch_done[i]) WBX_EHIP(e, hipEventCreateWithFlags(&e->patch_done[i], hipEventDisableTiming));
      }
    }
    if (e->patch_valid[patch_slot]) WBX_EHIP(e, hipEventSynchronize(e->patch_done[patch_slot]));
    for (uint32_t t = 0; t < N; t++) {
      e->h_patch[patch_slot][t] = e->tracks[t].patch;
      e->tracks[t].patch = DPatch{};
    }
    d_patch = e->h_patch[patch_slot];
    e->patches_pending = false;
  }

  // -- routing
  if (e->routing_dirty || c->routing_tracks != N) {
    std::vector<int32_t> tb(N);
    for (uint32_t t = 0; t < N; t++) tb[t] = e->tracks[t].bus;
    wbx_status st = wbx_set_routing(c, N, e->n_buses ? tb.data() : nullptr, e->n_buses);
    if (st != WBX_OK) return st;
    e->routing_dirty = false;
  }
  wbx_status st = upload_tables(c, N);
  if (st != WBX_OK) return st;
  st = ensure_result_buffers(c, K, N);
  if (st != WBX_OK) return st;

  // -- rows for the track-blocks the hot loop cannot stream directly: every clip start / end inside a block,
  //    and all blocks of integer-PCM or fast-forward clips
  {
    const size_t all = (size_t)K * N;
    const size_t rows = e->any_slow_clip ? all : std::min(all, 4 * e->total_clips + 2 * (size_t)N + 64);
    st = ensure_gen_capacity(c, rows);
    if (st != WBX_OK) return st;
    // templates: one per block with events (same bound as above) + one per steady run (a run ends at every event)
    st = ensure_template_capacity(c, std::min(all, rows + 2 * (size_t)N + 64) + (size_t)N);
    if (st != WBX_OK) return st;
  }

  // -- plan (sequencer on the device) + pre-render on the plan stream, into the other plan buffer; it may run
  //    while the mix of the previous render is still busy on the main stream
  c->cur = (c->cur + 1) % kRing;
  wbx_ctx::PlanBuf& B = PB(c);
  const bool plan_beside = c->overlap && K >= kOverlapMinBlocks;
  hipStream_t ps = plan_beside ? c->plan_stream : s;
  if (B.consumed_valid) WBX_EHIP(e, hipStreamWaitEvent(ps, B.consumed, 0));   // the mix that read this buffer two renders ago
  {
    const int pp = (int)(c->render_seq % kRing);
    if (plan_beside && c->sum_valid[pp]) {   // ... and the sum that read the partial buffer this render's mix will write
      WBX_EHIP(e, hipStreamWaitEvent(ps, c->sum_done[pp], 0));
      c->partial_wait_done = true;
    }
  }
  WBX_EHIP(e, hipMemsetAsync(B.counters, 0, 4 * sizeof(uint32_t), ps));
  const double sample_rate = (double)c->cfg.sample_rate;
  PlanArgs a{};
  a.clips = e->d_clips.p;
  a.clip_first = e->d_clip_first.p;
  a.samples = c->d_samples.p;
  a.state = e->d_state.p;
  a.patch = d_patch;
  a.gains = e->d_gains.p;
  a.rows = B.prows.p;
  a.tmpl = B.tmpl.p;
  a.tmpl_count = B.counters + 3;
  a.tmpl_cap = B.tmpl_cap;
  a.pool = B.pool.p;
  a.pool_count = B.counters;
  a.status = B.counters + 1;
  a.gen_list = B.gen_list.p;
  a.gen_count = B.counters + 2;
  a.gen_cap = B.gen_cap;
  a.pool_chunks = B.pool_chunks;
  a.n_tracks = N;
  a.n_blocks = K;
  a.block_frames = F;
  a.channels = C;
  a.sample_rate = sample_rate;
  a.playing = e->playing ? 1u : 0u;
  a.clips_changed = e->clips_edited ? 1u : 0u;
  e->clips_edited = false;
  a.playhead = e->playhead;
  a.sample_position = e->sample_position;
  a.beat_duration = e->beat_duration;
  launch_plan(a, ps);
  if (patch_slot >= 0) {
    WBX_EHIP(e, hipEventRecord(e->patch_done[patch_slot], ps));
    e->patch_valid[patch_slot] = true;
  }
  st = launch_pre_render(c, K, ps);
  if (st != WBX_OK) return st;
  if (plan_beside) WBX_EHIP(e, hipEventRecord(B.planned, ps));   // (in-stream: the mix simply follows)

  // -- mix + sum on the main stream, after the plan
  if (plan_beside) WBX_EHIP(e, hipStreamWaitEvent(s, B.planned, 0));
  c->levels_target = reinterpret_cast<uint32_t*>(e->d_levels.p);
  c->has_window_clips = e->any_window_clip;
  c->has_stride_clips = e->any_stride_clip;
  const int mix_parity = (int)(c->render_seq % kRing);
  st = launch_mix_sum(c, K, N);
  if (st != WBX_OK) return st;
  B.consumed = c->mix_done[mix_parity];   // recorded right after the mix: the plan buffer is free before the sum runs
  B.consumed_valid = true;

  // -- transport: the host repeats the arithmetic of Engine::process (engine.cpp:1578-1585, :1619-1623) that
  //    the plan kernel performs for its K blocks, so both sides hold the same playhead / sample_position bits
  double playhead = e->playhead, sample_position = e->sample_position;
  for (uint32_t b = 0; b < K; b++) {
    double buffer_duration = (double)F / sample_rate;
    double current_beat_duration = e->beat_duration;
    double buffer_duration_in_beats = buffer_duration / current_beat_duration;
    double next_playhead_pos = playhead + buffer_duration_in_beats;
    if (e->playing) {
      sample_position += beat_to_samples(buffer_duration_in_beats, sample_rate, current_beat_duration);
      playhead = next_playhead_pos;
    }
  }
  e->playhead = playhead;
  e->sample_position = sample_position;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_process(wbx_engine* e, float* const* out_planar) {   // engine.cpp:1576-1654
  if (!e || !out_planar) return WBX_ERR_INVALID;
  wbx_ctx* c = e->ctx;
  if (c->master_target) {   // the caller redirected the master: leave it there and fetch the ordinary way
    wbx_status st = wbx_engine_render(e, 1);
    if (st != WBX_OK) return st;
    return wbx_fetch(c, out_planar, nullptr, nullptr);
  }
  const uint32_t C = c->cfg.channels, F = c->cfg.block_frames;
  if (!e->h_block) {
    WBX_EHIP(e, hipHostMalloc((void**)&e->h_block, (size_t)C * F * sizeof(float), hipHostMallocDefault));
    WBX_EHIP(e, hipHostMalloc((void**)&e->h_status, 4 * sizeof(uint32_t), hipHostMallocDefault));
  }
  c->master_target = e->h_block;          // sum_kernel's stores go over PCIe into the staging block,
  c->status_dst = e->h_status;            // and it drops the plan status next to it
  wbx_status st = wbx_engine_render(e, 1);
  c->master_target = nullptr;
  c->status_dst = nullptr;
  if (st != WBX_OK) return st;
  WBX_EHIP(e, join_sum(c));
  WBX_EHIP(e, hipStreamSynchronize(c->stream));
  drain_events(c);
  for (uint32_t ch = 0; ch < C; ch++) std::memcpy(out_planar[ch], e->h_block + (size_t)ch * F, F * sizeof(float));
  c->last_master_on_host = true;   // set after launch_mix_sum cleared it: the master of this block is e->h_block
  const uint32_t flags = e->h_status[1];
  if (flags & 3u) return efail(e, WBX_ERR_OVERFLOW, "segment plan overflow (raise wbx_config.max_segments)");
  if (flags & 8u) return efail(e, WBX_ERR_OVERFLOW, "more boundary / non-fp32 track-blocks than pre-render rows");
  if (flags & 16u) return efail(e, WBX_ERR_OVERFLOW, "plan template array full");
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_transport(wbx_engine* e, double* playhead, double* sample_position, int* playing) {
  if (!e) return WBX_ERR_INVALID;
  if (playhead) *playhead = e->playhead;
  if (sample_position) *sample_position = e->sample_position;
  if (playing) *playing = e->playing ? 1 : 0;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_levels(wbx_engine* e, float* levels, uint32_t n_tracks) {
  if (!e || !levels || n_tracks > e->state_tracks) return WBX_ERR_INVALID;
  wbx_ctx* c = e->ctx;
  const size_t n = (size_t)n_tracks * c->cfg.channels;
  WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
  WBX_EHIP(e, hipMemcpyAsync(levels, e->d_levels.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  WBX_EHIP(e, hipMemsetAsync(e->d_levels.p, 0, n * sizeof(float), c->stream));   // VUMeter::update exchanges with 0 (vu_meter.h:33)
  WBX_EHIP(e, hipStreamSynchronize(c->stream));
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_fetch_plan(wbx_engine* e, wbx_plan_record* out, size_t cap, size_t* n_out) {
  if (!e || !n_out) return WBX_ERR_INVALID;
  wbx_ctx* c = e->ctx;
  if (c->last_K == 0) return efail(e, WBX_ERR_FAILED, "nothing rendered");
  const uint32_t K = c->last_K, N = c->last_N;
  std::vector<DTrackBlock> tb((size_t)K * N);
  uint32_t pc[4] = {0, 0, 0, 0};
  WBX_EHIP(e, hipStreamSynchronize(c->stream));
  WBX_EHIP(e, hipMemcpy(pc, PB(c).counters, sizeof(pc), hipMemcpyDeviceToHost));
  {
    // rebuild the per-(block, track) records from the 16-B rows and the templates they point at
    std::vector<DRow> rows((size_t)K * N);
    const uint32_t nt = std::min(pc[3], PB(c).tmpl_cap);
    std::vector<DTrackBlock> tmpl(nt);
    WBX_EHIP(e, hipMemcpy(rows.data(), PB(c).prows.p, rows.size() * sizeof(DRow), hipMemcpyDeviceToHost));
    if (nt) WBX_EHIP(e, hipMemcpy(tmpl.data(), PB(c).tmpl.p, nt * sizeof(DTrackBlock), hipMemcpyDeviceToHost));
    // templates the pre-render pass rewrote: put the sequencer's originals back
    const uint32_t ng = std::min(pc[2], PB(c).gen_cap);
    if (ng) {
      std::vector<uint32_t> idx(ng);
      std::vector<DTrackBlock> saved(ng);
      WBX_EHIP(e, hipMemcpy(idx.data(), PB(c).gen_list.p, ng * sizeof(uint32_t), hipMemcpyDeviceToHost));
      WBX_EHIP(e, hipMemcpy(saved.data(), PB(c).saved.p, ng * sizeof(DTrackBlock), hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < ng; i++)
        if (idx[i] < tmpl.size()) tmpl[idx[i]] = saved[i];
    }
    for (size_t i = 0; i < rows.size(); i++) {
      tb[i] = DTrackBlock{};
      if (rows[i].tmpl >= tmpl.size()) continue;   // no stream call at all in this track-block
      tb[i] = tmpl[rows[i].tmpl];
      if (rows[i].flags & ROW_POS) tb[i].pos = rows[i].pos;
    }
  }
  const uint32_t used = std::min(pc[0], PB(c).pool_chunks);
  std::vector<DSeg> pool((size_t)used * kChunk);
  if (used) WBX_EHIP(e, hipMemcpy(pool.data(), PB(c).pool.p, pool.size() * sizeof(DSeg), hipMemcpyDeviceToHost));
  size_t n = 0;
  for (uint32_t b = 0; b < K; b++)
    for (uint32_t t = 0; t < N; t++) {
      const DTrackBlock& r = tb[(size_t)b * N + t];
      for (uint32_t i = 0; i < r.nseg; i++) {
        const DSeg s0 = get_seg0(r);
        const DSeg* sg = (i == 0) ? &s0 : (r.extra < used ? &pool[(size_t)r.extra * kChunk + (i - 1)] : nullptr);
        if (!sg) continue;
        if (out && n < cap) {
          wbx_plan_record& o = out[n];
          o.block = b;
          o.track = t;
          o.buffer_offset = sg->dst_start;
          o.num_samples = sg->req_len;
          o.num_actual = sg->len;
          o.sample = sg->sample;
          o.sample_offset = sg->pos;
          o.playback_speed = sg->speed;
          o.gain = sg->gain;
          o._pad = sg->flags;
        }
        n++;
      }
    }
  *n_out = n;
  if (pc[1] & 3u) return efail(e, WBX_ERR_OVERFLOW, "segment plan overflow");
  return WBX_OK;
}
