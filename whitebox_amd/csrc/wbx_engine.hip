// wbx_engine.hip — layer 2 of libwbx.so (wbx_engine): the reference's Engine / Track surface.
//
// The host keeps what the UI thread edits (clip lists, parameters, transport: wbx_host.h, plain C++ shared with the
// CPU-only test harness), the device keeps what the audio thread mutates per block (sequencer + sampler state) and does
// all per-block work.  Threading is the reference's (wbx_host.h): one audio thread in wbx_engine_process / _render
// holding the editor lock for its host side, one UI thread whose edits take the same lock and whose
// wbx_track_set_volume / _pan / _mute go through the per-track SPSC ring without it.
#include <chrono>

#include "wbx_ctx.h"
#include "wbx_host.h"
#include "wbx_seq.h"

using namespace wbx;

struct wbx_engine {
  wbx_ctx* ctx = nullptr;
  HostSession hs;
  int knob_plan_seg = -1;               // WBX_PLAN_SEG, read once at wbx_engine_create: 0 off, n > 0 segments of n blocks, -1 unset
  // pinned host buffers the plan kernel reads in place (one 16-B / 8-B read per lane): state patches and per-track
  // gains reach the device without a copy and, above all, without a stream synchronisation that would drain the
  // renders the audio thread has run ahead by.  Rings of three; a buffer is refilled only after the plan kernel that
  // last read it has finished.
  DPatch* h_patch[kRing] = {};
  uint32_t patch_cap[kRing] = {};
  hipEvent_t patch_done[kRing] = {};
  bool patch_valid[kRing] = {};
  uint32_t patch_seq = 0;
  float* h_gains[kRing] = {};           // [N][2] fl(volume * pan_coeffs[c])
  uint32_t gains_cap = 0;
  hipEvent_t gains_done[kRing] = {};
  bool gains_valid[kRing] = {};
  int gains_slot = -1;                  // the buffer plans currently read
  std::vector<float> gains_tmp;
  // the one-block callback reads the gains from a device copy (refreshed, in stream order, only when a parameter changed):
  // a read of pinned host memory is the longest round trip in its sequencer's latency chain
  DevBuf<float> d_gains_cb;
  uint64_t gains_gen = 0, d_gains_cb_gen = ~0ull;   // how often the pinned gains were rebuilt / which build the copy mirrors
  // the per-block transport records of a batch render (PlanArgs::times): K dependent additions, done here on the host — it
  // repeats that arithmetic anyway to keep its own transport — and moved in front of the plan by a kernel that reads this
  // pinned table (one lane of the GPU beside a running mix took 0.15-0.2 ms for 2048 blocks).  A ring of eight: the host
  // runs at most that many renders ahead of the sequencer before it waits for a table's copy.
  static constexpr int kTimesRing = 8;
  DBlockTime* h_times[kTimesRing] = {};
  uint32_t times_cap[kTimesRing] = {};
  hipEvent_t times_done[kTimesRing] = {};
  bool times_valid[kTimesRing] = {};
  uint32_t times_seq = 0;
  // Engine::process (one block per call): pinned, device-mapped host staging the sum kernel writes the block into
  // and the plan status lands in — the callback path then needs no copy-engine transfer at all
  float* h_block = nullptr;             // [C][F]
  uint32_t* h_status = nullptr;         // plan counters [4]; from [8]: the one-launch callback's completion words (its sequence
                                        // number, one word or one per workgroup), kCbFlags of them
  static constexpr uint32_t kCbFlags = 256;
  uint32_t cb_seq = 0;                  // ... of the last launch
  uint32_t cb_give_ups = 0;             // one-launch callbacks mixed again because a workgroup gave up at the spread barrier
  bool in_process = false;              // render_locked runs inside wbx_engine_process, which waits for the block: the
                                        // pinned tables need no completion events
  bool gen_skipped = false;             // ... and left the pre-render launch out (expecting an empty queue)
  bool plan_status_on_host = false;     // the counters of the last plan are in h_status (the device copy was cleared)
  size_t d_clips_count = 0;
  bool clips_uploaded = false;          // the device holds a clip table (its internal_state_changed flags are live)
  // the sequencer cut along the time axis (wbx_seq.h plan_segment): seam states of the render being planned — one buffer, the
  // two plan kernels that use it run back to back on one stream — and its running statistics
  DevBuf<DTrackState> d_seam;           // [2][N][segments]: guesses, end states
  DevBuf<uint32_t> d_seg_stats;         // [2] tracks with a seam that did not hold, segments planned again
  DevBuf<uint32_t> d_seg_ticket;        // [max_tracks] SegArgs::ticket (all zero between launches)
  hipStream_t seam_stream = nullptr;    // the stream that used d_seam last
  hipEvent_t seam_done = nullptr;
  bool table_flags = false;             // the uploaded clip table holds a set internal_state_changed flag (segmented plans off)
  uint32_t* h_flags_left = nullptr;     // pinned: how many of them are left — the sequencer counts down as it clears them
  uint64_t seg_renders = 0;             // renders planned by segments so far
  uint32_t seg_last_segs = 0;           // ... segments per track of the last one
  uint32_t state_tracks = 0;            // tracks that have device state
  std::vector<DClip> flat;              // staging of the clip table upload
  std::vector<uint32_t> first;

  DevBuf<DClip> d_clips;
  DevBuf<uint32_t> d_clip_first;
  // [max_tracks], allocated once by the first render.  Invariant: the slots from state_tracks on are all-zero (a track
  // added later starts from cleared state without any device work: no allocation, no copy, no synchronisation on the
  // audio thread).  permute_tracks_locked keeps it.
  DevBuf<DTrackState> d_state;
  DevBuf<float> d_levels;               // [max_tracks][2] running maxima, same invariant
  // wbx_engine_levels (UI thread): the take kernel runs on a stream of its own into this pinned block; the editor lock
  // is held only while the take is ENQUEUED behind the renders in flight, never across the wait for it
  hipStream_t levels_stream = nullptr;
  hipEvent_t levels_ev = nullptr;
  // the sequencer of render n+1 continues from the per-track state render n's left: on one stream that is the stream's
  // order; when the stream changes (short renders plan on the main stream, batch renders on the plan stream) the new one
  // waits for this event, recorded at the old one's tail
  hipStream_t last_plan_stream = nullptr;
  hipEvent_t plan_handover = nullptr;
  uint32_t* h_levels = nullptr;         // [max_tracks][2]
};

namespace {

// The message of the last failed engine call made by the CALLING thread: the UI thread and the audio thread fail
// independently, one shared string would be a data race.
thread_local std::string tls_err;

wbx_status efail(wbx_engine*, wbx_status s, const char* what) {
  tls_err = what;
  return s;
}

// a failed layer-1 call made under the lock: take its message along
wbx_status cfail(wbx_engine* e, wbx_status s) {
  if (s != WBX_OK) tls_err = e->ctx->err;
  return s;
}

#define WBX_EHIP(e, call)                                           \
  do {                                                              \
    hipError_t _e = (call);                                         \
    if (_e != hipSuccess) {                                         \
      tls_err = std::string(#call) + ": " + hipGetErrorString(_e);  \
      return WBX_ERR_DEVICE;                                        \
    }                                                               \
  } while (0)

// Blocks per segment when the sequencer of this render is cut along the time axis (wbx_seq.h: plan_segment), 0 = one lane
// per track walks all K blocks.  Taken by batch renders of sessions with tracks cut into clips — a chain of dependent look-ups
// per clip boundary and lane: c3 cut into 5.3-block clips planned 2048 blocks in 5.4 ms (4.4 ms for 128-frame blocks, in front
// of a 2.6 ms mix) — while the transport runs and no clip carries a set internal_state_changed flag (the sequencer clears
// those in passing, track.cpp:373,392,418: the run-up of one lane and the real walk of another would race for them).
// About 16 k lanes: segments of K * N / 24576 blocks rounded up to a power of two, at least 32 (measured, profiles/
// r04_ab_seglen.txt: 256 tracks best at 32-64 blocks — 16 and 128 are 9-15 % slower —, 4096 tracks x 2048 short blocks at 512).
// WBX_PLAN_SEG=0: off; =<n>: segments of n blocks (A/B aid, tests).
uint32_t plan_segment_length(wbx_engine* e, uint32_t K, uint32_t N, bool playing) {
  if (e->knob_plan_seg == 0) return 0u;   // (WBX_PLAN_SEG as wbx_engine_create read it: the audio thread never calls getenv)
  const uint32_t forced = e->knob_plan_seg > 0 ? (uint32_t)e->knob_plan_seg : 0u;
  if (e->table_flags && e->h_flags_left && *reinterpret_cast<volatile uint32_t*>(e->h_flags_left) == 0u)
    e->table_flags = false;   // every flag of the table has been cleared by a plan that is over
  if (!playing || e->table_flags || e->in_process || N == 0u || e->ctx->seg_broken) return 0u;
  if (forced) return forced < K ? forced : 0u;
  if (!e->hs.cut_tracks || K < 128u) return 0u;
  // (also where the one-lane walk would hide behind the previous mix — c3 / 16-bit sessions cut into 5.3- or 20-block clips at
  //  512 frames: the segment lanes are done early, the next mix starts 0.06 instead of 0.15 ms behind its predecessor, steps
  //  0-3 % shorter, never longer: profiles/r04_ab_planseg.txt)
  uint32_t len = 32u;
  while ((uint64_t)(K / len) * N > 24576u && len < K) len *= 2u;
  return len < K ? len : 0u;
}

// A/B aid (WBX_FORCE_CUT=1): an uncut session through the instances a session cut into clips takes
bool force_cut_instances() {
  static const bool on = [] { const char* v = std::getenv("WBX_FORCE_CUT"); return v && v[0] == '1'; }();
  return on;
}

bool sample_in_use_cb(void* owner, uint32_t sample) { return static_cast<wbx_engine*>(owner)->hs.sample_referenced(sample); }

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
  e->hs.max_tracks = cfg->max_tracks;
  e->hs.dst_rate = cfg->sample_rate;
  if (const char* v = std::getenv("WBX_PLAN_SEG")) e->knob_plan_seg = std::max(0, std::atoi(v));
  if (const char* v = std::getenv("WBX_PLAN_LANES")) {   // tuning knob; measured on c3 cut into clips of 5.3 / 20 blocks:
    const int n = std::atoi(v);                          // 64, 32, 16 and 8 tracks per wave within 2 % of each other
    if (n == 1 || n == 2 || n == 4 || n == 8 || n == 16 || n == 32 || n == 64) e->hs.plan_lanes_knob = (uint32_t)n;
  }
  c->owner = e;
  c->sample_in_use = sample_in_use_cb;
  if (hipStreamCreateWithFlags(&e->levels_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&e->levels_ev, e->ctx->dev_event_flags) != hipSuccess ||
      hipHostMalloc((void**)&e->h_levels, (size_t)cfg->max_tracks * 2 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
    wbx_engine_destroy(e);
    return WBX_ERR_OOM;
  }
  *out = e;
  return WBX_OK;
}

extern "C" void wbx_engine_destroy(wbx_engine* e) {
  if (!e) return;
  if (e->ctx) {
    (void)hipSetDevice(e->ctx->cfg.device);
    (void)hipStreamSynchronize(e->ctx->plan_stream);
    (void)sync_main(e->ctx);
  }
  if (e->levels_stream) {
    (void)hipStreamSynchronize(e->levels_stream);
    (void)hipStreamDestroy(e->levels_stream);
  }
  if (e->levels_ev) (void)hipEventDestroy(e->levels_ev);
  if (e->plan_handover) (void)hipEventDestroy(e->plan_handover);
  if (e->h_levels) (void)hipHostFree(e->h_levels);
  e->d_clips.release();
  e->d_clip_first.release();
  e->d_state.release();
  e->d_levels.release();
  e->d_gains_cb.release();
  for (int i = 0; i < kRing; i++) {
    if (e->h_patch[i]) (void)hipHostFree(e->h_patch[i]);
    if (e->patch_done[i]) (void)hipEventDestroy(e->patch_done[i]);
    if (e->h_gains[i]) (void)hipHostFree(e->h_gains[i]);
    if (e->gains_done[i]) (void)hipEventDestroy(e->gains_done[i]);
  }
  if (e->seam_done) (void)hipEventDestroy(e->seam_done);
  if (e->h_flags_left) (void)hipHostFree(e->h_flags_left);
  for (int i = 0; i < wbx_engine::kTimesRing; i++) {
    if (e->h_times[i]) (void)hipHostFree(e->h_times[i]);
    if (e->times_done[i]) (void)hipEventDestroy(e->times_done[i]);
  }
  if (e->h_block) (void)hipHostFree(e->h_block);
  if (e->h_status) (void)hipHostFree(e->h_status);
  wbx_destroy(e->ctx);
  delete e;
}

// Engine::set_audio_channel_config (engine.cpp:43-57) on a live engine: new block size / channel count / device rate,
// tracks, clips, samples and the transport stay — like the reference, which only resizes its mixing buffers.  Sampler
// state of clips that are playing keeps the playback speed it was started with (Sampler::reset_state ran with the
// old rate), as in the reference.
extern "C" wbx_status wbx_engine_set_audio_channel_config(wbx_engine* e, uint32_t output_channels, uint32_t buffer_size,
                                                          uint32_t sample_rate) {
  if (!e) return WBX_ERR_INVALID;
  if (output_channels < 1 || output_channels > 2 || buffer_size < 4 || (buffer_size & 3u) || buffer_size > 32768 || sample_rate == 0)
    return efail(e, WBX_ERR_INVALID, "set_audio_channel_config: channels 1-2, buffer size a multiple of 4 up to 32768");
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  wbx_ctx* c = e->ctx;
  if (c->cfg.channels == output_channels && c->cfg.block_frames == buffer_size && c->cfg.sample_rate == sample_rate) return WBX_OK;
  if (c->dist) return efail(e, WBX_ERR_UNSUPPORTED, "set_audio_channel_config: shut the multi-GPU exchange down first (its buffers have the old shape)");
  (void)hipSetDevice(c->cfg.device);
  WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
  WBX_EHIP(e, join_sum(c));
  WBX_EHIP(e, hipStreamSynchronize(c->sum_stream));
  WBX_EHIP(e, sync_main(c));
  WBX_EHIP(e, hipStreamSynchronize(e->levels_stream));   // (a meter read in flight: d_levels is cleared below)
  drain_events(c);
  c->cfg.channels = output_channels;
  c->cfg.block_frames = buffer_size;
  c->cfg.sample_rate = sample_rate;
  // everything whose size or row pitch depends on C or F is dropped and re-grown by the next render
  for (auto& B : c->pb) {
    B.rows.release();
    B.gen_list.release();
    B.saved.release();
    B.gen_cap = 0;
    B.consumed_valid = false;
  }
  for (auto& P : c->d_partial2) P.release();
  for (auto& v : c->sum_valid) v = false;
  c->sum_pending = -1;
  c->d_master.release();
  c->d_buses.release();
  for (auto& P : c->d_peaks) P.release();
  c->buses_clean = false;
  c->d_zero.release();
  WBX_EHIP(e, c->d_zero.ensure(buffer_size + 8));
  WBX_EHIP(e, hipMemset(c->d_zero.p, 0, (buffer_size + 8) * sizeof(float)));
  c->last_K = 0;
  if (e->h_block) {
    WBX_EHIP(e, hipHostFree(e->h_block));
    e->h_block = nullptr;
  }
  if (e->d_levels.p) WBX_EHIP(e, hipMemset(e->d_levels.p, 0, e->d_levels.cap * sizeof(float)));
  // the destination rate enters every clip's playback speed: which clips the hot loop streams directly is re-derived
  e->hs.set_dst_rate_locked(sample_rate);
  return WBX_OK;
}

extern "C" const char* wbx_engine_last_error(const wbx_engine* e) {
  if (!e) return "null engine";
  return tls_err.c_str();
}

extern "C" wbx_ctx* wbx_engine_ctx(wbx_engine* e) { return e ? e->ctx : nullptr; }

extern "C" wbx_status wbx_engine_set_bpm(wbx_engine* e, double bpm) {   // engine.cpp:24-30: an atomic store, no lock
  if (!e || !(bpm > 0.0)) return WBX_ERR_INVALID;
  e->hs.set_bpm(bpm);
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_set_playhead_position(wbx_engine* e, double beat) {   // engine.cpp:32-41
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.set_playhead_position_locked(beat);
  e->hs.note_edit_locked();
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_add_track(wbx_engine* e, uint32_t* track_out) {   // engine.cpp:200-208
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (e->hs.n_tracks() >= e->ctx->cfg.max_tracks) return efail(e, WBX_ERR_INVALID, "max_tracks reached");
  const uint32_t t = e->hs.add_track_locked();
  if (track_out) *track_out = t;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_set_buses(wbx_engine* e, uint32_t n_buses) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.n_buses = n_buses;
  e->hs.routing_dirty = true;
  e->hs.note_edit_locked();
  return WBX_OK;
}

// Track::set_volume / set_pan / set_mute (track.cpp:47-79): UI thread, no lock — a message into the track's ring,
// applied by the audio thread at the start of its next block.  Like the reference's, the producer yields while the
// ring (63 usable entries) is full: it needs a running audio thread to make progress then.
extern "C" wbx_status wbx_track_set_volume(wbx_engine* e, uint32_t t, float db) {
  if (!e || !e->hs.valid_track(t)) return WBX_ERR_INVALID;
  e->hs.set_volume(t, db);
  return WBX_OK;
}

extern "C" wbx_status wbx_track_set_pan(wbx_engine* e, uint32_t t, float pan) {
  if (!e || !e->hs.valid_track(t)) return WBX_ERR_INVALID;
  e->hs.set_pan(t, pan);
  return WBX_OK;
}

extern "C" wbx_status wbx_track_set_mute(wbx_engine* e, uint32_t t, int mute) {
  if (!e || !e->hs.valid_track(t)) return WBX_ERR_INVALID;
  e->hs.set_mute(t, mute != 0);
  return WBX_OK;
}

namespace {

// new track i = old track order[i] (order.size() = new track count): the per-track device state (sequencer,
// sampler, running levels) follows its Track object, as the pointers in the reference's vector do
wbx_status permute_tracks_locked(wbx_engine* e, const std::vector<uint32_t>& order) {
  wbx_ctx* c = e->ctx;
  (void)hipSetDevice(c->cfg.device);
  WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
  WBX_EHIP(e, join_sum(c));
  WBX_EHIP(e, sync_main(c));
  // (a meter read of the UI thread may still be in flight on its own stream — wbx_engine_levels drops the editor lock
  //  before it waits: its exchange-with-zero must not land in the slots rewritten below)
  WBX_EHIP(e, hipStreamSynchronize(e->levels_stream));
  const uint32_t new_n = (uint32_t)order.size();
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
  e->hs.permute_tracks_locked(order);
  return WBX_OK;
}

}  // namespace

extern "C" wbx_status wbx_engine_delete_track(wbx_engine* e, uint32_t slot) {   // engine.cpp:210-218
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(slot)) return WBX_ERR_INVALID;
  std::vector<uint32_t> order;
  for (uint32_t i = 0; i < e->hs.n_tracks(); i++)
    if (i != slot) order.push_back(i);
  return permute_tracks_locked(e, order);
}

extern "C" wbx_status wbx_engine_clear_all(wbx_engine* e) {   // engine.cpp:59-66: every track goes
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  return permute_tracks_locked(e, std::vector<uint32_t>{});
}

extern "C" wbx_status wbx_engine_move_track(wbx_engine* e, uint32_t from_slot, uint32_t to_slot) {   // engine.cpp:228-243
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(from_slot) || !e->hs.valid_track(to_slot)) return WBX_ERR_INVALID;
  if (from_slot == to_slot) return WBX_OK;
  std::vector<uint32_t> order(e->hs.n_tracks());
  for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
  order.erase(order.begin() + from_slot);
  order.insert(order.begin() + to_slot, from_slot);
  return permute_tracks_locked(e, order);
}

extern "C" wbx_status wbx_engine_solo_track(wbx_engine* e, uint32_t slot) {   // engine.cpp:245-262 (no lock: messages only)
  if (!e || !e->hs.valid_track(slot)) return WBX_ERR_INVALID;
  e->hs.solo_track(slot);
  return WBX_OK;
}

extern "C" wbx_status wbx_track_set_bus(wbx_engine* e, uint32_t t, int32_t bus) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(t)) return WBX_ERR_INVALID;
  e->hs.tracks[t]->bus = bus;
  e->hs.routing_dirty = true;
  return WBX_OK;
}

// ---- effect slot: kept in the boundary, nothing processed through it ----
extern "C" wbx_status wbx_engine_add_plugin_to_track(wbx_engine* e, uint32_t track, const wbx_plugin* plugin) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(track)) return WBX_ERR_INVALID;
  if (plugin)
    return efail(e, WBX_ERR_UNIMPLEMENTED, "effect processing is not part of this path (third-party VST3 arithmetic): the slot stays empty");
  e->hs.tracks[track]->plugin = nullptr;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_delete_plugin_from_track(wbx_engine* e, uint32_t track) {   // engine.h:229
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(track)) return WBX_ERR_INVALID;
  e->hs.tracks[track]->plugin = nullptr;
  return WBX_OK;
}

extern "C" wbx_status wbx_track_get_plugin(wbx_engine* e, uint32_t track, const wbx_plugin** plugin_out) {
  if (!e || !plugin_out) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  if (!e->hs.valid_track(track)) return WBX_ERR_INVALID;
  *plugin_out = static_cast<const wbx_plugin*>(e->hs.tracks[track]->plugin);
  return WBX_OK;
}

namespace {

// Sample assets (SampleAsset, engine/assets_table.h:22-35).  The reference decodes a file outside the editor lock and
// only links the finished asset in; here the id is reserved under the lock, the audio goes to HBM on the upload
// stream without it, and the finished slot is published under the lock again — the audio thread never waits for a
// transfer.
wbx_status add_sample_common(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate, uint64_t frames,
                             const ClipFill& f, uint32_t* sample_out) {
  if (!e || !sample_out) return WBX_ERR_INVALID;
  wbx_ctx* c = e->ctx;
  uint32_t id;
  {
    LockGuard g(e->hs.editor_lock);
    id = (uint32_t)c->clips.size();
    c->clips.emplace_back();            // reserved, unused until published
    e->hs.samples.resize(c->clips.size());
  }
  ClipSlot s;
  wbx_status st = clip_build(c, s, format, channels, sample_rate, frames, f, c->upload_stream);
  if (st == WBX_OK && hipStreamSynchronize(c->upload_stream) != hipSuccess) {
    clip_release(c, s);
    st = WBX_ERR_DEVICE;
  }
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (st != WBX_OK) return cfail(e, st);
  st = clip_publish(c, id, s);
  if (st != WBX_OK) return cfail(e, st);
  SampleMeta& m = e->hs.samples[id];
  m.format = (uint32_t)format;
  m.channels = channels;
  m.sample_rate = sample_rate;
  m.count = frames;
  m.used = true;
  *sample_out = id;
  return WBX_OK;
}

}  // namespace

extern "C" wbx_status wbx_engine_add_sample(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate,
                                            uint64_t frames, const void* const* planar, uint32_t* sample_out) {
  ClipFill f{};
  f.kind = CLIP_SRC_PLANAR;
  f.planar = planar;
  return add_sample_common(e, format, channels, sample_rate, frames, f, sample_out);
}

extern "C" wbx_status wbx_engine_add_sample_interleaved(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate,
                                                        uint64_t frames, const void* interleaved, uint32_t* sample_out) {
  ClipFill f{};
  f.kind = CLIP_SRC_INTERLEAVED_HOST;
  f.interleaved = interleaved;
  return add_sample_common(e, format, channels, sample_rate, frames, f, sample_out);
}

extern "C" wbx_status wbx_engine_add_sample_synth(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate,
                                                  uint64_t frames, uint64_t seed, uint32_t key_track, float amp,
                                                  uint32_t* sample_out) {
  ClipFill f{};
  f.kind = CLIP_SRC_SYNTH;
  f.seed = seed;
  f.key_track = key_track;
  f.amp = amp;
  return add_sample_common(e, format, channels, sample_rate, frames, f, sample_out);
}

// Drop a sample asset (SampleAsset::release when its last reference goes, assets_table.h:22-35): refused while a clip
// list still names it.  The engine-side form of wbx_clip_free: it takes the editor lock.
extern "C" wbx_status wbx_engine_delete_sample(wbx_engine* e, uint32_t sample) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_sample(sample)) return efail(e, WBX_ERR_INVALID, "unknown sample");
  if (e->hs.sample_referenced(sample)) return efail(e, WBX_ERR_INVALID, "sample is still referenced by a clip (delete the clips first)");
  const wbx_status st = wbx_clip_free(e->ctx, sample);
  if (st != WBX_OK) return cfail(e, st);
  e->hs.samples[sample] = SampleMeta{};
  return WBX_OK;
}

// Engine::add_audio_clip -> add_to_cliplist, engine.cpp:293-309, :409-461
extern "C" wbx_status wbx_engine_add_audio_clip(wbx_engine* e, uint32_t track, double min_time, double max_time,
                                                double start_offset, uint32_t sample, double speed, float gain) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(track)) return WBX_ERR_INVALID;
  if (!e->hs.valid_sample(sample) && sample < e->ctx->clips.size() && e->ctx->clips[sample].used) {
    // a clip the host put into the engine's pool through layer 1 (wbx_clip_upload on wbx_engine_ctx): adopt it
    const DSample& d = e->ctx->clips[sample].d;
    if (e->hs.samples.size() <= sample) e->hs.samples.resize(sample + 1);
    e->hs.samples[sample] = SampleMeta{d.format, d.channels, d.sample_rate, d.count, true};
  }
  if (!e->hs.valid_sample(sample)) return efail(e, WBX_ERR_INVALID, "unknown sample");
  if (!(min_time <= max_time)) return efail(e, WBX_ERR_INVALID, "min_time > max_time");
  e->hs.add_audio_clip_locked(track, min_time, max_time, start_offset, sample, speed, gain);
  return WBX_OK;
}

// Engine::move_clip, engine.cpp:346-363
extern "C" wbx_status wbx_engine_move_clip(wbx_engine* e, uint32_t track, uint32_t clip, double relative_pos) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(track) || clip >= e->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  e->hs.move_clip_locked(track, clip, relative_pos);
  return WBX_OK;
}

// Engine::resize_clip, engine.cpp:365-398
extern "C" wbx_status wbx_engine_resize_clip(wbx_engine* e, uint32_t track, uint32_t clip, double relative_pos,
                                             double resize_limit, double min_length, int left_side, int shift,
                                             int stretch) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(track) || clip >= e->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  e->hs.resize_clip_locked(track, clip, relative_pos, resize_limit, min_length, left_side != 0, shift != 0, stretch != 0);
  return WBX_OK;
}

// Engine::delete_clip, engine.cpp:400-407
extern "C" wbx_status wbx_engine_delete_clip(wbx_engine* e, uint32_t track, uint32_t clip) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(track) || clip >= e->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  e->hs.delete_clip_locked(track, clip);
  return WBX_OK;
}

// Engine::delete_region, engine.cpp:463-475
extern "C" wbx_status wbx_engine_delete_region(wbx_engine* e, uint32_t track, double min, double max) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(track) || !(min <= max)) return WBX_ERR_INVALID;
  e->hs.delete_region_locked(track, min, max);
  return WBX_OK;
}

// Engine::set_clip_gain, engine.cpp:1460-1464
extern "C" wbx_status wbx_engine_set_clip_gain(wbx_engine* e, uint32_t track, uint32_t clip, float gain) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.note_edit_locked();
  if (!e->hs.valid_track(track) || clip >= e->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  e->hs.set_clip_gain_locked(track, clip, gain);
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_clip_count(wbx_engine* e, uint32_t track, uint32_t* count) {
  if (!e || !count) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  if (!e->hs.valid_track(track)) return WBX_ERR_INVALID;
  *count = (uint32_t)e->hs.tracks[track]->clips.size();
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_get_clip(wbx_engine* e, uint32_t track, uint32_t clip, wbx_clip_info* out) {
  if (!e || !out) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  if (!e->hs.valid_track(track) || clip >= e->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  const DClip& d = e->hs.tracks[track]->clips[clip].d;
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
  LockGuard g(e->hs.editor_lock);
  e->hs.play_locked();
  e->hs.note_edit_locked();
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_stop(wbx_engine* e) {   // engine.cpp:82-93, Track::stop track.cpp:249-256
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  e->hs.stop_locked();
  e->hs.note_edit_locked();
  return WBX_OK;
}

namespace {

// (re)allocate the three pinned patch and gain buffers at once — pinning memory synchronises the device, so it must
// not happen in mid-run
wbx_status ensure_pinned_tables(wbx_engine* e, uint32_t N) {
  wbx_ctx* c = e->ctx;
  if (e->gains_cap >= N && e->patch_cap[0] >= N) return WBX_OK;
  WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
  WBX_EHIP(e, sync_main(c));
  const uint32_t cap = std::max<uint32_t>(N, c->cfg.max_tracks);
  for (int i = 0; i < kRing; i++) {
    if (e->h_patch[i]) WBX_EHIP(e, hipHostFree(e->h_patch[i]));
    if (e->h_gains[i]) WBX_EHIP(e, hipHostFree(e->h_gains[i]));
    e->h_patch[i] = nullptr;
    e->h_gains[i] = nullptr;
    WBX_EHIP(e, hipHostMalloc((void**)&e->h_patch[i], (size_t)cap * sizeof(DPatch), hipHostMallocDefault));
    WBX_EHIP(e, hipHostMalloc((void**)&e->h_gains[i], (size_t)cap * 2 * sizeof(float), hipHostMallocDefault));
    e->patch_cap[i] = cap;
    e->patch_valid[i] = false;
    e->gains_valid[i] = false;
    if (!e->patch_done[i]) WBX_EHIP(e, hipEventCreateWithFlags(&e->patch_done[i], e->ctx->dev_event_flags));
    if (!e->gains_done[i]) WBX_EHIP(e, hipEventCreateWithFlags(&e->gains_done[i], e->ctx->dev_event_flags));
  }
  e->gains_cap = cap;
  e->gains_slot = -1;
  e->hs.gains_dirty = true;   // the buffers are new: fill one
  return WBX_OK;
}

// the audio thread's render, editor lock held by the caller
wbx_status render_locked(wbx_engine* e, uint32_t K) {
  wbx_ctx* c = e->ctx;
  HostSession& hs = e->hs;
  tls_err.clear();
  if (K > c->cfg.max_blocks) return efail(e, WBX_ERR_INVALID, "n_blocks above wbx_config.max_blocks");
  const uint32_t N = hs.n_tracks();
  (void)hipSetDevice(c->cfg.device);
  const uint32_t C = c->cfg.channels, F = c->cfg.block_frames;
  hipStream_t s = c->stream;
  hs.render_edit_seq = hs.edit_seq;
  // engine.cpp:1579,1585: the tempo and the play state are read once per block
  const double beat_duration = hs.beat_duration.load(std::memory_order_relaxed);
  const bool playing = hs.playing.load(std::memory_order_relaxed);
  if (N == 0) {
    // Engine::process with an empty track list: output_buffer.clear() and the transport advance (engine.cpp:1598,
    // :1619-1623) — silence
    // — or, with a running master to continue (wbx_set_master_init; WBX_DIST_CHAIN on a rank whose shard is empty: more ranks
    // than tracks, or all of its tracks deleted), that sum handed on unchanged: the receive must be posted like any other
    // render's, or the previous rank's send is never matched and the sum of all earlier ranks is lost
    WBX_EHIP(e, join_sum(c));
    WBX_EHIP(e, c->d_master.ensure((size_t)K * C * F));
    const bool continues = c->master_init || dist_receives_running_sum(c);
    if (c->master_format && (c->dist || continues))
      return efail(e, WBX_ERR_UNSUPPORTED, "an empty session continues a running master / feeds a multi-GPU exchange in planar fp32 only");
    hipError_t me = hipSuccess;
    float* master = begin_master(c, s, &me);
    WBX_EHIP(e, me);
    if (continues) {
      wbx_status ist = WBX_OK;
      const float* init = c->dist ? dist_mix_init(c, K, s, &ist) : c->master_init;
      if (ist != WBX_OK) return cfail(e, ist);
      launch_clamp_into(init, master, (size_t)K * C * F, c->clamp ? 1 : 0, s);   // (the clamp follows the last addition: none here)
      if (c->dist) WBX_EHIP(e, dist_mix_issued(c, s));
      WBX_EHIP(e, hipGetLastError());
    } else {   // (all-zero bytes are silence in every output format; packed 24-bit counts its 3 bytes per sample)
      const size_t eb = c->master_format == WBX_OUT_I16 ? 2 : c->master_format == WBX_OUT_I24 ? 3 : 4;
      WBX_EHIP(e, hipMemsetAsync(master, 0, (size_t)K * C * F * eb, s));
    }
    c->last_master = master;
    c->last_master_format = c->master_format;
    c->last_master_on_host = false;
    c->last_K = K;
    c->last_N = 0;
    hs.advance_transport_locked(K, F, beat_duration);
    return WBX_OK;
  }

  wbx_status st = ensure_pinned_tables(e, N);
  if (st != WBX_OK) return st;

  // -- parameters: drain the message rings (process_track_messages track.cpp:773-779) and apply them
  //    (track.cpp:618-643); the factor used per sample is fl(volume * pan_coeffs[c]) (track.cpp:728-731)
  if (hs.drain_params_locked()) {
    const int slot = (e->gains_slot + 1) % kRing;
    if (e->gains_valid[slot]) WBX_EHIP(e, hipEventSynchronize(e->gains_done[slot]));   // its last reader, >= 2 changes ago
    hs.build_gains_locked(e->gains_tmp);
    std::memcpy(e->h_gains[slot], e->gains_tmp.data(), e->gains_tmp.size() * sizeof(float));
    e->gains_slot = slot;
    e->gains_gen++;
  }

  // -- clip lists
  if (hs.clips_dirty) {
    WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
    WBX_EHIP(e, sync_main(c));
    // Clip::internal_state_changed is cleared by the sequencer on the device (track.cpp:373,392,418): before
    // the table is replaced, take the live flags back for every clip no edit has touched since the last upload
    if (e->clips_uploaded && e->d_clips_count) {
      std::vector<DClip> live(e->d_clips_count);
      WBX_EHIP(e, hipMemcpy(live.data(), e->d_clips.p, live.size() * sizeof(DClip), hipMemcpyDeviceToHost));
      hs.merge_live_flags_locked(live.data(), live.size());
    }
    hs.flatten_clips_locked(e->flat, e->first);
    WBX_EHIP(e, e->d_clips.ensure(std::max<size_t>(1, e->flat.size())));
    WBX_EHIP(e, e->d_clip_first.ensure(N + 1));
    e->d_clips_count = e->flat.size();
    e->clips_uploaded = true;
    if (!e->h_flags_left) WBX_EHIP(e, hipHostMalloc((void**)&e->h_flags_left, 64, hipHostMallocDefault));
    uint32_t set_flags = 0;   // (both streams were drained above: nothing counts down right now)
    for (const DClip& dc : e->flat) set_flags += dc.internal_state_changed != 0 ? 1u : 0u;
    *e->h_flags_left = set_flags;
    e->table_flags = set_flags != 0u;
    if (!e->flat.empty()) WBX_EHIP(e, hipMemcpy(e->d_clips.p, e->flat.data(), e->flat.size() * sizeof(DClip), hipMemcpyHostToDevice));
    WBX_EHIP(e, hipMemcpy(e->d_clip_first.p, e->first.data(), e->first.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }

  // -- per-track device state: allocated once at max_tracks capacity; tracks added since the last render find their
  //    slots cleared already (wbx_engine::d_state)
  if (e->state_tracks < N) {
    if (e->d_state.cap < N) {
      WBX_EHIP(e, hipStreamSynchronize(c->plan_stream));
      WBX_EHIP(e, sync_main(c));
      WBX_EHIP(e, e->d_state.ensure(std::max<size_t>(N, c->cfg.max_tracks)));
      WBX_EHIP(e, e->d_levels.ensure((size_t)c->cfg.max_tracks * 2));
      WBX_EHIP(e, hipMemset(e->d_state.p, 0, e->d_state.cap * sizeof(DTrackState)));
      WBX_EHIP(e, hipMemset(e->d_levels.p, 0, e->d_levels.cap * sizeof(float)));
    }
    e->state_tracks = N;
  }

  // -- pending state edits (play / stop / clip-list changes)
  const DPatch* d_patch = nullptr;
  int patch_slot = -1;
  if (hs.patches_pending) {
    patch_slot = (int)(e->patch_seq++ % kRing);
    if (e->patch_valid[patch_slot]) WBX_EHIP(e, hipEventSynchronize(e->patch_done[patch_slot]));
    hs.take_patches_locked(e->h_patch[patch_slot]);
    d_patch = e->h_patch[patch_slot];
  }

  // -- routing
  if (hs.routing_dirty || c->routing_tracks != N) {
    std::vector<int32_t> tb(N);
    for (uint32_t t = 0; t < N; t++) tb[t] = hs.tracks[t]->bus;
    st = wbx_set_routing(c, N, hs.n_buses ? tb.data() : nullptr, hs.n_buses);
    if (st != WBX_OK) return cfail(e, st);
    hs.routing_dirty = false;
  }
  st = upload_tables(c, N);
  if (st != WBX_OK) return cfail(e, st);
  st = ensure_result_buffers(c, K, N);
  if (st != WBX_OK) return cfail(e, st);

  // -- rows for the track-blocks the hot loop cannot stream directly, and the plan's templates
  c->has_window_clips = hs.any_window_clip;
  c->has_stride_clips = hs.any_stride_clip;
  c->has_taps_clips = hs.any_taps_clip;
  c->has_lean16_clips = hs.any_win16_clip && !hs.any_other_window_clip;
  c->has_cut_tracks = hs.cut_tracks != 0 || force_cut_instances();
  c->short_render_now = K < kOverlapMinBlocks;
  c->render_blocks_now = K;
  c->whole_lists_now = render_walks_whole_lists(c, K);   // (enters the choice of the mix instance; reads the flags above)
  c->chain_now = render_chains_groups(c, K);
  c->masked_rows = mix_takes_masked_rows(c, hs.any_window_clip, hs.any_stride_clip);
  // the sequencer of this render: one lane per track, or — long renders of sessions cut into clips — per (track, segment)
  const uint32_t seg_len = plan_segment_length(e, K, N, playing);
  const uint32_t n_segs = seg_len ? (K + seg_len - 1u) / seg_len : 1u;
  // (segments: a track whose seam missed is planned again from there, and what its replaced segments queued / took from the
  //  pool / reserved stays allocated and unused — room for both versions of every row, so that a miss on a session that leans
  //  on the pre-render pass cannot turn into a capacity error)
  st = ensure_gen_capacity(c, hs.gen_rows_hint(K, c->masked_rows, ((double)F / (double)c->cfg.sample_rate) / beat_duration) * (seg_len ? 2u : 1u));
  if (st != WBX_OK) return cfail(e, st);
  if (seg_len) {
    st = ensure_pool_slack(c);
    if (st != WBX_OK) return cfail(e, st);
  }
  st = ensure_template_capacity(c, std::max(hs.template_hint(K, n_segs), (size_t)2 * N));   // (2 N: the one-launch callback's static pairs)
  if (st != WBX_OK) return cfail(e, st);

  // -- plan (sequencer on the device) + pre-render on the plan stream, into the other plan buffer; it may run
  //    while the mix of the previous render is still busy on the main stream
  c->cur = (c->cur + 1) % kRing;
  wbx_ctx::PlanBuf& B = PB(c);
  const bool plan_beside = c->overlap && K >= kOverlapMinBlocks;
  hipStream_t ps = plan_beside ? c->plan_stream : s;
  hipStream_t ms = pick_mix_stream(c, K, true);   // main stream, or the alternate one for every other batch render
  const bool plan_event = ps != ms;               // the mix runs on another stream than its plan
  if (B.consumed_valid) WBX_EHIP(e, hipStreamWaitEvent(ps, B.consumed, 0));   // the mix that read this buffer two renders ago
  {
    const int pp = (int)(c->render_seq % kRing);
    if (plan_beside && c->sum_valid[pp]) {   // ... and the sum that read the partial buffer this render's mix will write
      WBX_EHIP(e, hipStreamWaitEvent(ps, c->knob_partial_free_off ? c->sum_done[pp] : c->partial_free[pp], 0));   // (the sum KERNEL: not the copy-out of its master behind it)
      c->partial_wait_done = true;
    }
  }
  // the plan's four counters start from zero: cleared by the kernel that brings the transport table in front of the plan when
  // there is one (below), else by a memset — a 16-byte fill is a launch of its own, ~50 us beside a running mix, at the head of
  // the chain fill -> table -> plan -> pre-render that a short render's mix waits for
  bool need_zero = !B.counters_zero;
  B.counters_zero = false;
  e->plan_status_on_host = false;
  const double sample_rate = (double)c->cfg.sample_rate;
  PlanArgs a{};
  a.clips = e->d_clips.p;
  a.clip_first = e->d_clip_first.p;
  a.samples = c->d_samples.p;
  a.state = e->d_state.p;
  a.patch = d_patch;
  a.gains = e->h_gains[e->gains_slot];
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
  a.playing = playing ? 1u : 0u;
  a.clips_changed = hs.clips_edited ? 1u : 0u;
  a.flags_left = e->h_flags_left;
  hs.clips_edited = false;
  c->short_render_now = K < kOverlapMinBlocks;
  c->render_blocks_now = K;
  c->whole_lists_now = render_walks_whole_lists(c, K);   // (enters the choice of the mix instance below)
  c->chain_now = render_chains_groups(c, K);
  // clip boundaries inside a block stay in the hot loop when the mix instance of this render can take them
  c->has_window_clips = hs.any_window_clip;
  c->has_stride_clips = hs.any_stride_clip;
  c->has_taps_clips = hs.any_taps_clip;
  c->has_lean16_clips = hs.any_win16_clip && !hs.any_other_window_clip;
  c->has_cut_tracks = hs.cut_tracks != 0 || force_cut_instances();
  c->masked_rows = mix_takes_masked_rows(c, hs.any_window_clip, hs.any_stride_clip);
  a.masked_rows = c->masked_rows;
  a.tmpl_reserve = HostSession::template_reserve(K);
  a.lanes = hs.plan_lanes(K);
  a.playhead = hs.playhead;
  a.sample_position = hs.sample_position;
  a.beat_duration = beat_duration;
  a.times = nullptr;
  // (WBX_PLAN_LDS_TABLE=0: A/B aid — the device-memory table for cut sessions too.  Measured again in round 4, with the run
  //  end estimated instead of searched: plan of c3 cut into 5.3-block clips 5.4 ms from LDS, 10.9 ms from device memory, 4 /
  //  2 / 1 tracks per wave 12-25 ms: every record look-up at a clip boundary is a memory round trip for its lane)
  static const bool lds_table_for_cut = [] { const char* v = std::getenv("WBX_PLAN_LDS_TABLE"); return !(v && v[0] == '0'); }();
  if (seg_len) a.tmpl_reserve = 8u;   // (a lane plans seg_len blocks, not K: a smaller reservation strands less)
  if (seg_len || (plan_beside && (!c->has_cut_tracks || !lds_table_for_cut))) {
    // Batch render of a session whose tracks are single clips (a steady run per track, a handful of look-ups): the transport
    // records live in device memory and the sequencer takes the register-capped instance — nothing in LDS, a wave no larger
    // than a mix wave, so it runs BESIDE the previous mix instead of in the drain at its end.  Sessions cut into clips
    // search the records all the time: theirs stay in LDS (device-memory latency tripled their plan) with the roomy instance.
    WBX_EHIP(e, B.times.ensure(K));
    a.times = B.times.p;
    const int ts = (int)(e->times_seq++ % wbx_engine::kTimesRing);
    if (e->times_valid[ts]) WBX_EHIP(e, hipEventSynchronize(e->times_done[ts]));   // its copy of 8 renders ago (long over)
    if (e->times_cap[ts] < K) {
      if (e->h_times[ts]) (void)hipHostFree(e->h_times[ts]);
      e->h_times[ts] = nullptr;
      e->times_cap[ts] = 0;
      WBX_EHIP(e, hipHostMalloc((void**)&e->h_times[ts], (size_t)K * sizeof(DBlockTime), hipHostMallocDefault));
      e->times_cap[ts] = K;
    }
    if (!e->times_done[ts]) WBX_EHIP(e, hipEventCreateWithFlags(&e->times_done[ts], e->ctx->dev_event_flags));
    block_times(a, e->h_times[ts]);   // wbx_seq.h: the source the device compiles
    static const bool by_memset = [] { const char* v = std::getenv("WBX_COUNTERS_MEMSET"); return v && v[0] == '1'; }();   // A/B aid
    launch_times_copy(e->h_times[ts], B.times.p, K, (need_zero && !by_memset) ? B.counters : nullptr, ps);
    if (!by_memset) need_zero = false;
    WBX_EHIP(e, hipEventRecord(e->times_done[ts], ps));
    e->times_valid[ts] = true;
  }
  if (need_zero) WBX_EHIP(e, hipMemsetAsync(B.counters, 0, 4 * sizeof(uint32_t), ps));
  if (e->last_plan_stream && e->last_plan_stream != ps) {
    // (found by the random-pieces test once it left renders unfetched: a batch render's plan, on the idle plan stream,
    //  overtook the plan of a short render still queued on the main stream behind earlier mixes and read the state before
    //  that one had written it)
    if (!e->plan_handover) WBX_EHIP(e, hipEventCreateWithFlags(&e->plan_handover, e->ctx->dev_event_flags));
    WBX_EHIP(e, hipEventRecord(e->plan_handover, e->last_plan_stream));
    WBX_EHIP(e, hipStreamWaitEvent(ps, e->plan_handover, 0));
  }
  e->last_plan_stream = ps;
  // The one-block callback of a session whose clip boundaries stay in the hot loop: the pre-render queue is empty
  // unless a block holds three or more stream calls or overlapping ones.  Leave the launch out; wbx_engine_process
  // looks at the queue counter afterwards and repeats pre-render + mix for the (rare) block that needed it.
  // (sessions whose boundary blocks do go through the pre-render pass take the same bet when the block can be one launch:
  //  a steady block — nearly all — wins two launches, a boundary block pays the repeat)
  e->gen_skipped = e->in_process && !hs.any_slow_clip && (c->masked_rows || (K == 1u && callback_is_one_launch(c)));
  // ... and then sequencer, mix and sum are ONE launch (wbx_callback.h): every mix workgroup plans its own tracks first, the
  // last one to finish sums the block and tells the host
  const bool one_launch = e->gen_skipped && K == 1u && callback_is_one_launch(c);
  B.static_tmpl = one_launch;
  if (one_launch) a.tmpl_reserve = 0u;   // track t owns templates 2t, 2t + 1: no allocation round trip in the latency chain
  if (one_launch) {
    if (e->d_gains_cb_gen != e->gains_gen || e->d_gains_cb.cap < (size_t)N * 2) {
      WBX_EHIP(e, e->d_gains_cb.ensure(std::max<size_t>((size_t)c->cfg.max_tracks, N) * 2));
      WBX_EHIP(e, hipMemcpyAsync(e->d_gains_cb.p, e->h_gains[e->gains_slot], (size_t)N * 2 * sizeof(float), hipMemcpyHostToDevice, ps));
      e->d_gains_cb_gen = e->gains_gen;
    }
    a.gains = e->d_gains_cb.p;
  }
  if (seg_len) {
    // one lane per (track, segment); the lane that completes a track checks its seams; the seam states live in one buffer
    const size_t per = (size_t)N * n_segs;
    if (e->seam_stream && e->seam_stream != ps) {   // (its last user ran on the other stream)
      if (!e->seam_done) WBX_EHIP(e, hipEventCreateWithFlags(&e->seam_done, e->ctx->dev_event_flags));
      WBX_EHIP(e, hipEventRecord(e->seam_done, e->seam_stream));
      WBX_EHIP(e, hipStreamWaitEvent(ps, e->seam_done, 0));
    }
    if (e->d_seam.cap < 2 * per) {
      if (e->seam_stream) WBX_EHIP(e, hipStreamSynchronize(e->seam_stream));
      WBX_EHIP(e, e->d_seam.ensure(2 * per));
    }
    if (!e->d_seg_stats.p) {
      WBX_EHIP(e, e->d_seg_stats.ensure(2));
      WBX_EHIP(e, hipMemsetAsync(e->d_seg_stats.p, 0, 2 * sizeof(uint32_t), ps));
    }
    if (e->d_seg_ticket.cap < N) {
      if (e->seam_stream) WBX_EHIP(e, hipStreamSynchronize(e->seam_stream));
      const size_t cap = std::max<size_t>(N, c->cfg.max_tracks);
      WBX_EHIP(e, e->d_seg_ticket.ensure(cap));
      WBX_EHIP(e, hipMemsetAsync(e->d_seg_ticket.p, 0, e->d_seg_ticket.cap * sizeof(uint32_t), ps));
    }
    e->seam_stream = ps;
    SegArgs g{e->d_seam.p, e->d_seam.p + per, e->d_seg_stats.p, e->d_seg_ticket.p, seg_len, n_segs};
    launch_plan_segments(a, g, plan_beside, ps);
    e->seg_renders++;
    e->seg_last_segs = n_segs;
  } else if (!one_launch) {
    launch_plan(a, ps);
  }
  if (!e->in_process) {
    if (patch_slot >= 0) {
      WBX_EHIP(e, hipEventRecord(e->patch_done[patch_slot], ps));
      e->patch_valid[patch_slot] = true;
    }
    WBX_EHIP(e, hipEventRecord(e->gains_done[e->gains_slot], ps));   // (re-recorded by every plan that reads the buffer)
    e->gains_valid[e->gains_slot] = true;
  }
  if (!e->gen_skipped) {
    st = launch_pre_render(c, K, ps);
    if (st != WBX_OK) return cfail(e, st);
  }
  if (plan_event) WBX_EHIP(e, hipEventRecord(B.planned, ps));   // (plan and mix on one stream: the mix simply follows)

  // -- mix (main stream, or the alternate one for every other batch render) + sum, after the plan
  if (plan_event) WBX_EHIP(e, hipStreamWaitEvent(ms, B.planned, 0));
  c->levels_target = reinterpret_cast<uint32_t*>(e->d_levels.p);
  c->has_window_clips = hs.any_window_clip;
  c->has_stride_clips = hs.any_stride_clip;
  c->has_taps_clips = hs.any_taps_clip;
  c->has_lean16_clips = hs.any_win16_clip && !hs.any_other_window_clip;
  c->uniform_speed = hs.uniform_window_speed();
  const int mix_parity = (int)(c->render_seq % kRing);
  c->cb_plan = one_launch ? &a : nullptr;
  if (one_launch) {
    c->cb_flag = e->h_status + 8;
    c->cb_gave_up = e->h_status + 7;
    c->cb_flag_cap = wbx_engine::kCbFlags;
    if (++e->cb_seq == 0u) ++e->cb_seq;   // (0 is what the completion and election words hold before any launch)
    c->cb_seq = e->cb_seq;
  }
  st = launch_mix_sum(c, K, N);
  c->cb_plan = nullptr;
  if (st != WBX_OK) return cfail(e, st);
  B.consumed = c->mix_done[mix_parity];   // recorded right after the mix: the plan buffer is free before the sum runs
  B.consumed_valid = !c->cb_launched;   // (the one-launch callback records no event: wbx_engine_process waits for the launch itself)

  // -- transport: the host repeats the arithmetic of Engine::process (engine.cpp:1578-1585, :1619-1623) that
  //    the plan kernel performs for its K blocks, so both sides hold the same playhead / sample_position bits
  hs.advance_transport_locked(K, F, beat_duration);
  return WBX_OK;
}

}  // namespace

extern "C" wbx_status wbx_engine_render(wbx_engine* e, uint32_t K) {
  if (!e || K == 0) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  return render_locked(e, K);
}

namespace {
wbx_status process_block(wbx_engine* e, float* const* out_planar, int out_format, void* out_il);
}

extern "C" wbx_status wbx_engine_process(wbx_engine* e, float* const* out_planar) {   // engine.cpp:1576-1654
  if (!e || !out_planar) return WBX_ERR_INVALID;
  return process_block(e, out_planar, 0, nullptr);
}

// Engine::process + the back end's conversion to the device format (audio_io_pulseaudio.cpp:419-461:
// output_buffer.interleave_samples_to(buffer, 0, n, output_sample_format) -> core/audio_format_conv.cpp:5-91) in one
// call: the conversion is the epilogue of the sum kernel, the block arrives interleaved — no planar round trip, no extra
// launch.  dst: F * C samples of the format (packed 24-bit: the reference's bytes, see WBX_OUT_I24).
extern "C" wbx_status wbx_engine_process_interleaved(wbx_engine* e, int out_format, void* dst) {
  if (!e || !dst) return WBX_ERR_INVALID;
  if (out_format != WBX_OUT_I16 && out_format != WBX_OUT_I24 && out_format != WBX_OUT_I24_X8 && out_format != WBX_OUT_I32 &&
      out_format != WBX_OUT_F32)
    return efail(e, WBX_ERR_UNSUPPORTED, "interleaved output format");
  return process_block(e, nullptr, out_format, dst);
}

namespace {

wbx_status process_block_locked(wbx_engine* e, float* const* out_planar, int out_format, void* out_il);

wbx_status process_block(wbx_engine* e, float* const* out_planar, int out_format, void* out_il) {
  const auto t_in = std::chrono::steady_clock::now();   // ScopedPerformanceCounter, engine.cpp:1577
  LockGuard g(e->hs.editor_lock);   // held for the whole block, like editor_lock in Engine::process (engine.cpp:1587-1651)
  const wbx_status st = process_block_locked(e, out_planar, out_format, out_il);
  // perf_measurer.update(duration, audio_buffer_duration_ms), engine.cpp:1653 (there behind the unlock; one writer either way)
  e->hs.perf_update(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count(), e->ctx->cfg.block_frames);
  return st;
}

wbx_status process_block_locked(wbx_engine* e, float* const* out_planar, int out_format, void* out_il) {
  wbx_ctx* c = e->ctx;
  if (c->master_target || c->dist) {   // the caller redirected the master: leave it there and fetch the ordinary way
    if (out_format) return efail(e, WBX_ERR_UNSUPPORTED, "wbx_engine_process_interleaved: not with a redirected master / a multi-GPU exchange");
    wbx_status st = render_locked(e, 1);
    if (st != WBX_OK) return st;
    return cfail(e, wbx_fetch(c, out_planar, nullptr, nullptr));
  }
  const int saved_format = c->master_format;
  c->master_format = out_format;
  struct Restore {
    wbx_ctx* c;
    int f;
    ~Restore() { c->master_format = f; }
  } restore{c, saved_format};
  const uint32_t C = c->cfg.channels, F = c->cfg.block_frames;
  if (!e->h_block) WBX_EHIP(e, hipHostMalloc((void**)&e->h_block, (size_t)C * F * sizeof(float), hipHostMallocDefault));
  if (!e->h_status) {
    WBX_EHIP(e, hipHostMalloc((void**)&e->h_status, (8 + wbx_engine::kCbFlags) * sizeof(uint32_t), hipHostMallocDefault));
    std::memset(e->h_status, 0, (8 + wbx_engine::kCbFlags) * sizeof(uint32_t));
  }
  c->master_target = e->h_block;          // sum_kernel's stores go over PCIe into the staging block,
  c->status_dst = e->h_status;            // and it drops the plan status next to it
  e->in_process = true;
  c->zero_status = true;
  c->cb_launched = false;
  wbx_status st = render_locked(e, 1);
  e->in_process = false;
  const bool one_launch = st == WBX_OK && c->cb_launched;
  if (st == WBX_OK && e->hs.n_tracks() != 0) {
    if (one_launch) {
      // the kernel's own word: master and status are in host memory when it reads this launch's number.  Polled — no
      // completion signal between the device and the return of the callback; a launch that never reports (a device fault)
      // falls through to the stream's own error after two seconds
      // (one word, or one per workgroup when every workgroup stores a share of the master: all of them)
      volatile uint32_t* flag = e->h_status + 8;
      const uint32_t n_flags = c->cb_flags, seq = e->cb_seq;
      const auto t0 = std::chrono::steady_clock::now();
      uint32_t spins = 0, have = 0;
      while (have < n_flags) {
        if (flag[have] == seq) {
          have++;
          continue;
        }
        __builtin_ia32_pause();
        if ((++spins & 0xFFFFu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
          (void)sync_main(c);
          for (have = 0; have < n_flags && flag[have] == seq; have++) {
          }
          if (have < n_flags) {   // the launch never reported: start the ticket count afresh
            st = efail(e, WBX_ERR_DEVICE, "the one-launch callback did not report its block");
            if (c->d_cb_done) (void)hipMemset(c->d_cb_done, 0, kCbDoneWords * sizeof(uint32_t));
            c->cb_base = c->cb_base2 = 0;
          }
          break;
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    } else if (hipError_t he = sync_main(c); he != hipSuccess) {
      st = WBX_ERR_DEVICE;
    }
    if (st == WBX_OK) {
      PB(c).counters_zero = e->h_status[2] == 0u;   // (sum_kernel cleared them unless something was queued)
      e->plan_status_on_host = true;
    }
    if (st == WBX_OK && one_launch && e->h_status[7] == e->cb_seq) {
      // A workgroup of the spread sum gave up waiting for the rest of its grid (the grid was not resident at once: a CU mask
      // the attribute does not show, a device shared with another process): shares of the master were added from incomplete
      // group sums.  The launch itself is over (its flag is the LAST workgroup's) and the plan is intact — the counters were
      // copied, and cleared, only by a workgroup that had seen every sequencer lane finished (workgroup 0 with the full count,
      // else the reporter, which does not clear: wbx_callback.h) —: mix and sum the block again through three launches, and
      // keep to "the last workgroup adds everything" from here on.  h_status holds the plan's counters already; this sum must
      // not drop the device's copy over them (workgroup 0 may have cleared it: an overflow bit of this block would be lost).
      c->cb_no_spread = true;
      e->cb_give_ups++;
      c->zero_status = false;
      c->status_dst = nullptr;
      st = launch_mix_sum(c, 1, e->hs.n_tracks());
      c->status_dst = e->h_status;
      if (st == WBX_OK && sync_main(c) != hipSuccess) st = WBX_ERR_DEVICE;
      PB(c).counters_zero = false;
      if (st != WBX_OK) tls_err = c->err;
    }
    if (st == WBX_OK && e->gen_skipped && e->h_status[2] != 0u) {
      // the block did queue records for the pre-render pass: run it now and mix again (the plan is untouched; the
      // first pass counted those records as silence, so the running levels hold nothing wrong)
      c->zero_status = false;
      st = launch_pre_render(c, 1, c->stream);
      if (st == WBX_OK) st = launch_mix_sum(c, 1, e->hs.n_tracks());
      if (st == WBX_OK && sync_main(c) != hipSuccess) st = WBX_ERR_DEVICE;
      PB(c).counters_zero = false;
      if (st != WBX_OK) tls_err = c->err;
    }
  }
  c->zero_status = false;
  c->master_target = nullptr;
  c->status_dst = nullptr;
  if (st != WBX_OK) return st;
  if (!one_launch) WBX_EHIP(e, sync_main(c));
  for (int i = 0; i < kRing; i++) e->patch_valid[i] = e->gains_valid[i] = false;   // every plan that read them is over
  drain_events(c);
  if (out_format == 0) {
    for (uint32_t ch = 0; ch < C; ch++) std::memcpy(out_planar[ch], e->h_block + (size_t)ch * F, F * sizeof(float));
  } else if (out_format == WBX_OUT_I24) {   // (the reference's writer leaves the other 3*F*(C-1) bytes of the block untouched)
    std::memcpy(out_il, e->h_block, (size_t)F * 3);
  } else {
    std::memcpy(out_il, e->h_block, (size_t)F * C * (out_format == WBX_OUT_I16 ? 2 : 4));
  }
  c->last_master_on_host = true;   // set after launch_mix_sum cleared it: the master of this block is e->h_block
  if (e->hs.n_tracks() == 0) return WBX_OK;
  return cfail(e, plan_status_to_error(c, e->h_status[1]));
}

}  // namespace

extern "C" wbx_status wbx_engine_transport(wbx_engine* e, double* playhead, double* sample_position, int* playing) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  if (playhead) *playhead = e->hs.playhead;
  if (sample_position) *sample_position = e->hs.sample_position;
  if (playing) *playing = e->hs.playing.load(std::memory_order_relaxed) ? 1 : 0;
  return WBX_OK;
}

// VUMeter::level of every track: the maximum since the last call, reset by the call (VUMeter::update's
// `level.exchange(0.0f)`, vu_meter.h:33).  UI thread.  Tracks that have not been through a render yet (just added, or
// the audio callback is not running) read 0.  The reference's meters are lock-free atomics; here the editor lock is held
// only while the take is enqueued behind the renders issued so far (two stream calls), never across the device wait —
// the audio thread's next block is not held up by a meter read.
extern "C" wbx_status wbx_engine_levels(wbx_engine* e, float* levels, uint32_t n_tracks) {
  if (!e || !levels) return WBX_ERR_INVALID;
  wbx_ctx* c = e->ctx;
  (void)hipSetDevice(c->cfg.device);
  uint32_t have = 0, C = 0;
  {
    LockGuard g(e->hs.editor_lock);
    C = c->cfg.channels;
    have = std::min(n_tracks, e->state_tracks);
    if (have) {
      // every render issued before this call is covered: the take is ordered after the mixes in flight
      WBX_EHIP(e, join_alt(c));
      WBX_EHIP(e, hipEventRecord(e->levels_ev, c->stream));
      WBX_EHIP(e, hipStreamWaitEvent(e->levels_stream, e->levels_ev, 0));
      launch_levels_take(reinterpret_cast<uint32_t*>(e->d_levels.p), e->h_levels, have * C, e->levels_stream);
    }
  }
  if (have) {
    WBX_EHIP(e, hipStreamSynchronize(e->levels_stream));
    std::memcpy(levels, e->h_levels, (size_t)have * C * sizeof(float));
  }
  if (n_tracks > have) std::memset(levels + (size_t)have * C, 0, (size_t)(n_tracks - have) * C * sizeof(float));
  return WBX_OK;
}

// What the last process / render took from the shared state: the number of locked edits that had completed when it
// took the editor lock, and per track the cumulative count of parameter messages it (and its predecessors) drained.
// With the UI thread's own log of what it issued this reconstructs, block by block, the exact state the audio thread
// rendered from (tests/test_gpu_threads.py).
extern "C" wbx_status wbx_engine_thread_stats(wbx_engine* e, uint64_t* edits_seen, uint64_t* drained, uint32_t n_tracks) {
  if (!e) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  if (edits_seen) *edits_seen = e->hs.render_edit_seq;
  if (drained) {
    if (n_tracks > e->hs.n_tracks()) return WBX_ERR_INVALID;
    for (uint32_t t = 0; t < n_tracks; t++) drained[t] = e->hs.tracks[t]->drained;
  }
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_sequencer_stats(wbx_engine* e, uint64_t out[4]) {
  if (!e || !out) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  uint32_t st[2] = {0u, 0u};
  if (e->d_seg_stats.p && e->seam_stream) {
    WBX_EHIP(e, hipStreamSynchronize(e->seam_stream));
    WBX_EHIP(e, hipMemcpy(st, e->d_seg_stats.p, sizeof(st), hipMemcpyDeviceToHost));
  }
  out[0] = e->seg_renders;
  out[1] = st[0];
  out[2] = st[1];
  out[3] = e->seg_last_segs;
  return WBX_OK;
}

// Engine::perf_measurer.get_usage() (core/timing.h:64-66; read by ui/control_bar.cpp:54): the share of its period the audio
// callback has been taking, an exponential average over wbx_engine_process / _process_interleaved calls, clamped to [0, 1].
// Any thread.  last_block_ms (optional): the wall time of the last call, what the last update was fed.
extern "C" wbx_status wbx_engine_perf_usage(wbx_engine* e, double* usage, double* last_block_ms) {
  if (!e || !usage) return WBX_ERR_INVALID;
  *usage = e->hs.perf_get_usage();
  if (last_block_ms) {
    LockGuard g(e->hs.editor_lock);
    *last_block_ms = e->hs.last_block_ms;
  }
  return WBX_OK;
}
// the arithmetic behind it, host-only (tests hold it to the reference's PerformanceMeasurer and period helpers bit for bit)
extern "C" double wbx_calc_perf_update(double usage, double duration_ms, double period_ms) { return HostSession::perf_step(usage, duration_ms, period_ms); }
extern "C" double wbx_calc_perf_usage(double usage) { return HostSession::perf_clamped(usage); }
extern "C" double wbx_calc_buffer_period_ms(uint32_t buffer_size, uint32_t sample_rate) { return HostSession::buffer_period_ms(buffer_size, sample_rate); }

extern "C" wbx_status wbx_engine_callback_stats(wbx_engine* e, uint64_t out[4]) {
  if (!e || !out) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  out[0] = e->ctx->cb_launches;
  out[1] = e->ctx->cb_spread_launches;
  out[2] = e->cb_give_ups;
  out[3] = e->ctx->cb_no_spread ? 1u : 0u;
  return WBX_OK;
}

extern "C" wbx_status wbx_engine_fetch_plan(wbx_engine* e, wbx_plan_record* out, size_t cap, size_t* n_out) {
  if (!e || !n_out) return WBX_ERR_INVALID;
  LockGuard g(e->hs.editor_lock);
  wbx_ctx* c = e->ctx;
  if (c->last_K == 0) return efail(e, WBX_ERR_FAILED, "nothing rendered");
  const uint32_t K = c->last_K, N = c->last_N;
  uint32_t pc[4] = {0, 0, 0, 0};
  WBX_EHIP(e, sync_main(c));
  if (e->plan_status_on_host)
    std::memcpy(pc, e->h_status, sizeof(pc));
  else
    WBX_EHIP(e, hipMemcpy(pc, PB(c).counters, sizeof(pc), hipMemcpyDeviceToHost));
  std::vector<DRow> rows((size_t)K * N);
  const uint32_t nt = std::min(PB(c).static_tmpl ? 2u * N : pc[3], PB(c).tmpl_cap);
  std::vector<DTrackBlock> tmpl(nt);
  if (!rows.empty()) WBX_EHIP(e, hipMemcpy(rows.data(), PB(c).prows.p, rows.size() * sizeof(DRow), hipMemcpyDeviceToHost));
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
  const uint32_t used = std::min(pc[0], PB(c).pool_chunks);
  std::vector<DSeg> pool((size_t)used * kChunk);
  if (used) WBX_EHIP(e, hipMemcpy(pool.data(), PB(c).pool.p, pool.size() * sizeof(DSeg), hipMemcpyDeviceToHost));
  const size_t n = plan_records(K, N, rows.data(), tmpl.data(), tmpl.size(), pool.data(), used, out, cap);
  *n_out = n;
  // (every status bit the way wbx_fetch reports it — bit 7, the segmented sequencer's XCD check, included; bit 3, "more
  //  boundary rows than pre-render rows", concerns the audio, not the records handed out here)
  return cfail(e, plan_status_to_error(c, pc[1] & ~8u));
}
