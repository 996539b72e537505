// wbx_ctx.h — internals shared by the translation units of libwbx.so's host side (wbx_runtime.hip: layer 1,
// wbx_engine.hip: layer 2, wbx_dist.hip: multi-GPU).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/wbx.h"
#include "wbx_dev.h"

namespace wbx {
void launch_plan(const PlanArgs& a, hipStream_t s);
void launch_times_copy(const DBlockTime* host_pinned, DBlockTime* dev, uint32_t n_blocks, uint32_t* zero_counters, hipStream_t s);
void launch_gen(const GenArgs& a, uint32_t max_grid, hipStream_t s);
void launch_plan_segments(const PlanArgs& a, const SegArgs& g, bool beside, hipStream_t s);
const char* launch_mix(const MixArgs& a, uint32_t n_blocks, int variant, int family, hipStream_t s, hipEvent_t t0 = nullptr,
                       hipEvent_t t1 = nullptr);   // -> the instance's name
void launch_sum(const SumArgs& a, uint32_t n_blocks, hipStream_t s);
// the one-block callback as one launch (wbx_callback.h): sequencer + mix + sum + completion flag -> the instance's name
const char* launch_callback(const MixArgs& m, const PlanArgs& p, const SumArgs& s, uint32_t* done, uint32_t done_base, uint32_t done_base2, bool spread,
                            uint32_t* gave_up, uint32_t spin_bound, uint32_t* flag, uint32_t seq, int family, bool window_rows, unsigned long long* dbg, hipStream_t st);
uint32_t callback_spread_limit();      // grids of at most this many workgroups are resident at once (the device's CU count)
void launch_clamp(float* buf, size_t n, hipStream_t s);
void launch_clamp_into(const float* src, float* dst, size_t n, int clamp, hipStream_t s);
void launch_convert(const float* master, void* dst, uint32_t n_blocks, uint32_t F, uint32_t C, int fmt, hipStream_t s);
void launch_synth(void* dst, uint64_t frames, uint64_t key, float amp, int fmt, hipStream_t s);
void launch_levels_take(uint32_t* levels, uint32_t* dst, uint32_t n, hipStream_t s);
bool probe_xcd_layout(hipStream_t s, uint32_t* n_xcds);   // workgroup ids round-robin over 8 / 4 / 2 / 1 XCDs? (wbx_kernels.hip)
void launch_deinterleave(const void* src, void* dst0, void* dst1, uint64_t frames, uint32_t channels, uint32_t elem,
                         hipStream_t s);
void launch_mip(const MipArgs& a, int format, int bits, hipStream_t s);
}  // namespace wbx

namespace wbx {

constexpr int kEventRing = 64;
// Plan buffers, partial-sum buffers and their events form a ring of three: the plan of render i may start as soon as
// the mix of render i-3 and the sum of render i-3 are over, i.e. a full render before its own mix — the one-wave-per-
// track plan kernel is starved for CU slots while a mix runs, so it needs that much slack to stay off the critical path.
constexpr int kRing = 3;
constexpr uint32_t kPaceRing = 64;
constexpr uint32_t kCbDoneWords = 2 * 16 * 64;   // wbx_ctx::d_cb_done: two counters of kCbLanes words, kCbStride apart (wbx_callback.h)
constexpr uint32_t kOverlapMinBlocks = 8;   // renders shorter than this run plan, mix and sum on the main stream

// Clip audio lives in slabs of 1 GiB carved up in order (64-KiB granules): a session of thousands of clips is a few
// dozen large allocations, which the driver backs with large contiguous fragments (measured: the mix kernel's launch time
// is bimodal from process to process with one allocation per clip, 3-5 % apart, and stays at the fast end with slabs,
// tools/ab_arena.sh).  A slab's space is reused when the last clip in it has been freed; slabs go back to the
// driver with the context.  Clips above 256 MiB get an allocation of their own.
struct ClipSlab {
  char* mem = nullptr;
  size_t size = 0, used = 0;   // [0, used): handed out in order (bump); [used, size): untouched
  uint32_t live = 0;           // clips inside
  size_t live_bytes = 0;
  // extents below `used` that released clips gave back, sorted by offset, neighbours merged: first-fit for the next clip
  // that fits (replacing a clip again and again, or add / delete cycles beside a long-lived clip, stay inside the slab)
  std::vector<std::pair<size_t, size_t>> holes;   // (offset, bytes)
};

struct ClipSlot {
  void* alloc = nullptr;    // an allocation of its own (hipFree), or
  ClipSlab* slab = nullptr; // the slab it lives in
  size_t slab_off = 0, slab_len = 0;   // ... and its extent there (the gap in front of the clip included)
  void* base = nullptr;     // first channel row; all channels in one piece
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

struct DistState;   // wbx_dist.hip

}  // namespace wbx

using namespace wbx;   // (internal header: only the host-side translation units of the library include it)

struct wbx_ctx {
  wbx_config cfg{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  std::vector<ClipSlot> clips;
  std::vector<std::unique_ptr<ClipSlab>> slabs;   // clip storage (slab_mu: clips are built outside the editor lock)
  std::mutex slab_mu;
  std::atomic<uint32_t> slab_seq{0};  // clips placed so far (seeds the gap in front of the next one)
  DevBuf<DSample> d_samples;
  bool samples_dirty = true;

  // routing
  uint32_t routing_tracks = 0, n_buses = 0;
  std::vector<int32_t> track_bus;
  std::vector<uint32_t> order;
  std::vector<DGroup> groups;         // member lists cut into pieces of group_size tracks (one workgroup per piece and block)
  std::vector<DGroup> groups_exact;   // the member lists whole: the reference's summation order (build_routing)
  bool buses_alias_exact = false;     // buses_alias_partials of groups_exact
  bool whole_lists_now = false;       // the render being issued takes groups_exact (render_walks_whole_lists)
  uint32_t longest_list = 0;          // tracks in the longest member list
  mutable bool chain_broken = false;  // a chained render reported a failed hand-over (plan_status_to_error)
  bool chain_now = false;             // ... as chained workgroup-sized pieces (render_chains_groups)
  uint32_t chain_epoch = 0;
  DevBuf<uint32_t> d_chain;           // chained renders: the "running sum is out" words
  uint32_t* d_sticky_status = nullptr;   // failure bits (32 | 64) of every chained render since the last report (render_status)
  uint32_t exact_min_blocks = 1024;   // renders of at least this many blocks do, when the library picks the grouping
                                      // (WBX_EXACT_MIN_BLOCKS; 0 = never)
  DevBuf<uint32_t> d_order;
  DevBuf<DGroup> d_groups;
  bool routing_dirty = true;

  // The plan of a render (track-block records, overflow pool, pre-render queue + rows) is double-buffered:
  // the sequencer of step i+1 runs on `plan_stream` while the mix of step i runs on `stream`.
  struct PlanBuf {
    DevBuf<DRow> prows;               // [K][N] 16-B plan rows
    DevBuf<DTrackBlock> tmpl;         // templates the rows point at (one per steady run / per block with events)
    uint32_t tmpl_cap = 0;
    bool static_tmpl = false;         // the last plan into this buffer gave track t the templates 2t, 2t + 1 (no allocation count)
    DevBuf<DSeg> pool;
    uint32_t pool_chunks = 0;
    DevBuf<DBlockTime> times;         // [K] per-block transport records of a batch render (PlanArgs::times)
    uint32_t* counters = nullptr;     // [0] pool chunks allocated, [1] status bits, [2] generic records queued,
                                      // [3] templates allocated
    bool counters_zero = true;        // cleared already (at creation, or by the sum kernel of a callback block)
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
  DevBuf<float> d_master, d_buses, d_gains;
  DevBuf<float> d_peaks[2];           // per-track-block peaks: one buffer per mix stream (two mixes may be in flight)
  float* last_peaks = nullptr;        // where the last render / submit put its peaks
  // WBX_MIX_ALT=1 (experiment, off by default): consecutive batch renders of layer 2 alternate between the main stream
  // and `alt_stream`, so that nothing orders mix i+1 after mix i and the head of one can fill the CUs the tail of the
  // other leaves idle.  Everything else stays on the main stream, which joins the alternate one (join_alt) wherever it
  // joins the sum stream.  Measured: no gain in step time, each kernel's own interval grows by ~45 %.
  hipStream_t alt_stream = nullptr;
  hipStream_t cur_mix_stream = nullptr;   // the stream of the mix about to be / last launched
  int alt_pending = -1;                   // partial-buffer index of a mix on alt_stream the main stream has not joined
  bool mix_alternate = false;             // WBX_MIX_ALT=1
  // The sum of render i runs on its own stream beside the mix of render i+1 (it is PCIe-bound when the master goes to
  // host memory and needs few CUs).  sum_pending: a sum has been issued that the main stream has not waited for yet.
  hipStream_t sum_stream = nullptr;
  hipEvent_t mix_done[kRing] = {}, sum_done[kRing] = {};
  // recorded right behind the sum KERNEL, in front of the copy that takes a staged master to host memory: from here on the
  // partial buffer may be written again.  What the next user of the buffer waits for (round 6) — sum_done lies behind the
  // copy, 8 MB over PCIe per 2048-block render: a 256-track session's plans waited 0.3 ms for it and its mixes 0.07 ms for them
  hipEvent_t partial_free[kRing] = {};
  bool sum_valid[kRing] = {};
  int sum_pending = -1;
  uint32_t render_seq = 0;
  bool partial_wait_done = false;     // the caller already ordered this render after the sum of two renders ago
  bool sum_overlap = true;            // WBX_SUM_OVERLAP=0: sum on the main stream
  DevBuf<uint8_t> d_conv;
  DevBuf<unsigned long long> d_dbg;   // diagnostic (WBX_DBG_CLOCK=1): per-workgroup start / end times of the last mix
  size_t dbg_wgs = 0;
  std::vector<DTrackBlock> h_tb;      // layer-1 staging
  std::vector<DRow> h_rows;
  std::vector<DSeg> h_pool;

  uint32_t last_K = 0, last_N = 0;
  uint32_t* status_dst = nullptr;     // set by wbx_engine_process around its render: where sum_kernel drops the plan status
  // the one-launch callback (wbx_callback.h): set by wbx_engine_process around its render — the sequencer's arguments (the
  // launch then runs plan + mix + sum as one kernel), the pinned word the kernel writes `cb_seq` into when master and status
  // are out, and whether launch_mix_sum took that path
  const PlanArgs* cb_plan = nullptr;
  uint32_t* cb_flag = nullptr;
  uint32_t* cb_gave_up = nullptr;     // pinned word: the launch's number when one of its workgroups gave up at the spread barrier
  uint32_t cb_seq = 0;
  bool cb_launched = false;
  uint32_t* d_cb_done = nullptr;      // device: "workgroups done" ticket counter, never reset: launches count from cb_base
  uint32_t cb_base = 0;               // ... the first counter (every multi-group launch arrives there)
  uint32_t cb_base2 = 0;              // ... the second one (only launches whose workgroups each add a share of the master)
  uint64_t cb_launches = 0, cb_spread_launches = 0;   // one-launch callbacks issued / ... with the spread sum
  bool seg_broken = false;            // plan_seg_kernel found its XCD layout broken (status bit 7): one lane per track from then on
  uint32_t cb_spin_bound = 40000;     // polls of the spread barrier before a workgroup gives up (~50 ms; WBX_CB_SPIN_BOUND, read at wbx_create)
  // A/B switches read ONCE, at wbx_create (a test sets the variable and creates a new context): the audio callback never calls
  // getenv, and what a context decides at plan time (masked rows, lane space) cannot disagree with what it launches
  uint32_t n_xcds = 0;                // the XCD layout probe of wbx_create: 8 / 4 / 2 / 1, or 0 — not round-robin: no chained pieces
                                      // (chain_broken), no segmented sequencer (seg_broken) from the start
  bool knob_ragged_off = false;       // WBX_RAGGED=0: blocks between the instances' shapes take the general instance
  bool knob_cb_any_off = false;       // WBX_CB_ANY=0: the one-launch callback only for blocks that are exactly one 256-lane workgroup
  bool knob_no_uniform = false;       // WBX_NO_UNIFORM=1: MixArgs::uniform_speed withheld (the one-ratio modes off)
  bool knob_masked_rows_off = false;  // WBX_MASKED_ROWS=0: every clip boundary through the pre-render pass
  bool knob_chain_off = false;        // WBX_CHAIN=0: long renders walk whole member lists instead of chaining 128-track pieces
  bool knob_no_lean16 = false;        // WBX_NO_LEAN16: sessions of 16-bit PCM only through family 1
  bool knob_no_fam3 = false;          // WBX_NO_FAM3: resampled-integer sessions through family 1
  bool knob_no_cl2 = false;           // WBX_NO_CL2: never both channels of a frame in one lane
  bool knob_cb_fenced = false;        // WBX_CB_FENCED=1: release / acquire fences in the one-launch callback
  bool knob_partial_free_off = false; // WBX_PARTIAL_FREE=0: a partial buffer's next user waits for sum_done (behind the master's copy-out), as until round 5
  unsigned dev_event_flags = 0x2;     // hipEventDisableTiming [| hipEventReleaseToDevice]: events only other streams of this device wait for
  bool knob_mix_marker = false;       // WBX_MIX_MARKER=1: mix_done and the pace event as markers on the mix stream (as until round 5)
  bool knob_fast_partial_off = false; // WBX_FAST_PARTIAL=0: every partial stream call through the clamped masked arithmetic (as until round 5)
  int knob_packed_x = -1;             // WBX_PACKED_X=0|1: the packed masked-row instances off / on for every shape (-1: the library's choice)
  bool cb_no_spread = false;          // a spread launch gave up waiting for the whole grid (not resident at once: a CU mask, a
                                      // device shared with another process): the context keeps to "the last workgroup adds"
  uint32_t cb_flags = 1;              // completion words the launch writes (one, or one per workgroup: cb_flag[0 .. cb_flags))
  uint32_t cb_flag_cap = 1;           // ... and how many the engine's pinned block holds
  bool zero_status = false;           // ... and whether it clears the counters for the buffer's next plan
  bool buses_alias_partials = false;  // see build_routing
  const float* last_buses = nullptr;  // where the last render's bus sums are: d_buses or the partial buffer
  bool buses_clean = false;           // d_buses zeroed since the last routing change / reallocation
  float* last_master = nullptr;       // where the last render / submit put its master (d_master, the caller's target, or
  bool last_master_on_host = false;   // the engine's pinned staging block, which is host memory)
  bool clamp = true;
  float* master_target = nullptr;     // caller-owned device buffer, or null: d_master
  bool master_target_on_host = false; // ... it is pinned host memory: batch renders stage the master in d_stage and copy it out
  DevBuf<float> d_stage[kRing];       // (see launch_mix_sum: a sum that stores 8 MB over PCIe itself holds up the next mix)
  int master_format = 0;              // wbx_set_master_format: 0 planar fp32, else WBX_OUT_* interleaved (sum kernel epilogue)
  int last_master_format = 0;         // ... of the last render
  const float* master_init = nullptr; // wbx_set_master_init: the running sum the first group starts from, or null: zero

  // kernel timing (mix kernel)
  hipEvent_t ev[kEventRing][3]{};       // before the mix, after the mix, after the sum
  int ev_pending = 0;
  double mix_ms_total = 0.0;
  double gap_ms_total = 0.0;           // end of one mix -> start of the next, consecutive launches of one drain (wbx_gap_time)
  uint64_t gap_count = 0;
  double tail_ms_total = 0.0;          // mix end -> sum end (launch gap + sum kernel incl. its PCIe stores)
  uint64_t mix_launches = 0;
  bool profiling = true;
  const char* mix_kernel_name = "";   // the instance launch_mix chose last (wbx_kernel_name)
  double last_uniform_speed = 0.0;    // MixArgs::uniform_speed of the last launch (wbx_render_uniform_speed)
  int mix_unroll = 0;                 // WBX_MIX_VARIANT=10*U+W forces a kernel variant (results are identical);
                                      // 0 = chosen per render: 24 when resampled or integer-PCM clips are present, else 43
  bool has_window_clips = true;
  bool has_integer_clips = false;
  bool has_non16_clips = false;       // a clip asset that is not 16-bit PCM
  bool has_lean16_clips = false;      // resampled clips exist and all of them are 16-bit PCM at speeds up to 0.999 (layer 2)
  bool has_cut_tracks = true;         // some track holds more than one clip (layer 1: unknown, assume so)
  bool force_g = false;
  bool short_render_now = false;      // the render being issued is shorter than kOverlapMinBlocks (the callback path)
  uint32_t render_blocks_now = 0;     // ... its length in blocks
  bool has_taps_clips = true;         // ... rows read with per-frame taps (KIND_STRIDE) may occur: the instance must carry MODE_G
  bool has_stride_clips = true;       // fp32 clips played at speed > 0.999, != 1 may occur (layer 1: unknown, assume so)
  bool auto_group = false;            // wbx_config.group_size was 0: the library picks the track-group size
  uint32_t masked_rows = 0;           // the current plan holds partial-coverage rows / ROW_PAIRs for the hot loop (layer 2; PlanArgs level)
  double uniform_speed = 0.0;         // MixArgs::uniform_speed of the next launch (layer 2; 0 for host-sequenced plans)

  hipStream_t upload_stream = nullptr; // clip uploads of layer 2 run here, outside the engine's editor lock
  hipEvent_t ready_ev = nullptr;       // wbx_master_ready: results of an in-stream sum, for a foreign stream
  hipEvent_t pace_ev[kPaceRing] = {};  // wbx_pace
  uint64_t pace_seq = 0;
  // layer 2 back pointer: asks whether a clip list still names a sample (wbx_clip_free), null for a bare ctx
  bool (*sample_in_use)(void* owner, uint32_t sample) = nullptr;
  void* owner = nullptr;
  wbx::DistState* dist = nullptr;      // multi-GPU exchange (wbx_dist.hip), null on a single GPU
};

namespace wbx {

inline wbx_ctx::PlanBuf& PB(wbx_ctx* c) { return c->pb[c->cur]; }

inline wbx_status fail(wbx_ctx* c, wbx_status s, const char* what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return s;
}

#define WBX_HIP(ctx, call)                                                       \
  do {                                                                           \
    hipError_t _e = (call);                                                      \
    if (_e != hipSuccess) return ::wbx::fail((ctx), WBX_ERR_DEVICE, #call, _e);  \
  } while (0)

inline size_t fmt_bytes(int fmt) {
  switch (fmt) {
    case WBX_FMT_I16: return 2;
    case WBX_FMT_I24:
    case WBX_FMT_I32:
    case WBX_FMT_F32: return 4;
    default: return 0;
  }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

enum : int { CLIP_SRC_PLANAR = 0, CLIP_SRC_INTERLEAVED_HOST = 1, CLIP_SRC_INTERLEAVED_DEVICE = 2, CLIP_SRC_SYNTH = 3 };
struct ClipFill {           // where a new clip's audio comes from
  int kind;
  const void* const* planar;   // CLIP_SRC_PLANAR: host channel arrays
  const void* interleaved;     // CLIP_SRC_INTERLEAVED_*: [frames][channels]
  uint64_t seed;               // CLIP_SRC_SYNTH
  uint32_t key_track;
  float amp;
};

// wbx_runtime.hip
wbx_status clip_build(wbx_ctx* c, ClipSlot& s, int format, uint32_t channels, uint32_t sample_rate, uint64_t frames,
                      const ClipFill& f, hipStream_t on);
wbx_status clip_publish(wbx_ctx* c, uint32_t clip, ClipSlot& s);
void clip_release(wbx_ctx* c, ClipSlot& s);
hipError_t join_sum(wbx_ctx* c);
hipError_t join_alt(wbx_ctx* c);
hipError_t sync_main(wbx_ctx* c);          // the host waits for the main stream and every mix / sum beside it
hipStream_t pick_mix_stream(wbx_ctx* c, uint32_t K, bool alternate);
void drain_events(wbx_ctx* c, int upto = 0);
wbx_status upload_tables(wbx_ctx* c, uint32_t n_tracks);
wbx_status ensure_result_buffers(wbx_ctx* c, uint32_t K, uint32_t N);
wbx_status ensure_template_capacity(wbx_ctx* c, size_t n);
wbx_status ensure_gen_capacity(wbx_ctx* c, size_t rows);
wbx_status ensure_pool_slack(wbx_ctx* c);   // twice the default overflow pool (renders planned by segments; not when the host fixed max_segments)
wbx_status launch_pre_render(wbx_ctx* c, uint32_t K, hipStream_t on);
wbx_status launch_mix_sum(wbx_ctx* c, uint32_t K, uint32_t N);
wbx_status plan_status_to_error(wbx_ctx* c, uint32_t bits);
wbx_status render_status(wbx_ctx* c);      // the latched hand-over failures of chained renders (the streams must be idle)
float* begin_master(wbx_ctx* c, hipStream_t writer, hipError_t* err);
bool render_walks_whole_lists(const wbx_ctx* c, uint32_t K);
bool render_chains_groups(const wbx_ctx* c, uint32_t K);
int mix_family(const wbx_ctx* c);
bool mix_two_channels_per_lane(const wbx_ctx* c);
uint32_t mix_takes_masked_rows(const wbx_ctx* c, bool window_clips, bool stride_clips);
bool callback_is_one_launch(const wbx_ctx* c);
inline uint32_t lane_span_of(const wbx_ctx* c) { return native_lane_span(c->cfg.channels, c->cfg.block_frames >> 2, c->knob_ragged_off); }
uint32_t callback_lane_span(const wbx_ctx* c);   // lanes per channel of the one-launch callback's 256-lane workgroup, 0: three launches

// wbx_dist.hip
float* dist_begin_render(wbx_ctx* c, hipStream_t sum_stream, hipError_t* err);
const float* dist_mix_init(wbx_ctx* c, uint32_t K, hipStream_t mix_stream, wbx_status* st);
bool dist_receives_running_sum(const wbx_ctx* c);   // chain mode, rank > 0: every render continues the previous rank's sum
hipError_t dist_mix_issued(wbx_ctx* c, hipStream_t mix_stream);
void dist_destroy(wbx_ctx* c);

}  // namespace wbx
