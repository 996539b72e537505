// wbx_host.h — host side of layer 2 (wbx_engine): everything the reference's UI thread and audio thread share.
//
// Plain C++20, no HIP: the same code is compiled into libwbx.so (wbx_runtime.hip drives the device with what this
// file prepares) and into the CPU-only host harness of the tests (tests/cpp/host_sim.cpp: sequencer parity against
// the oracle without a GPU, and a ThreadSanitizer build of the two-thread contract).
//
// Threading contract — the reference's (SURVEY §8(b) "Threading"):
//   * exactly ONE audio thread calls process / render; it holds `editor_lock` for the whole host side of the block
//     (Engine::process, engine.cpp:1587-1651);
//   * ONE UI thread edits: every clip / track / transport edit takes the same lock (engine.cpp:35,70,85,204,211,231,
//     301,349,376,...); Track::set_volume / set_pan / set_mute do NOT lock — they push a TrackMessage::ParamChange into
//     the track's single-producer / single-consumer ring of 64 entries whose producer yields while it is full
//     (track.cpp:47-79, core/queue.h:142-196, track.h:131) and the audio thread drains it at the start of the track's
//     next block (process_track_messages, track.cpp:773-779);
//   * Engine::set_bpm stores an atomic double (engine.cpp:24-30), meters are atomics (vu_meter.h:17).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <numbers>
#include <thread>
#include <vector>

#include "wbx_clip_edit.h"
#include "wbx_dev.h"

namespace wbx {

// The editor lock: what the reference's Spinlock (core/thread.h:11-35) is to Engine::process and the edit calls — a
// test-and-test-and-set flag; a waiter watches the flag with plain loads and gives its time slice away between looks.
struct SpinLock {
  std::atomic_flag held = ATOMIC_FLAG_INIT;
  bool try_lock() noexcept { return !held.test(std::memory_order_relaxed) && !held.test_and_set(std::memory_order_acquire); }
  void lock() noexcept {
    while (held.test_and_set(std::memory_order_acquire))
      do std::this_thread::yield();
      while (held.test(std::memory_order_relaxed));
  }
  void unlock() noexcept { held.clear(std::memory_order_release); }
};

struct LockGuard {
  SpinLock& l;
  explicit LockGuard(SpinLock& s) : l(s) { l.lock(); }
  ~LockGuard() { l.unlock(); }
  LockGuard(const LockGuard&) = delete;
  LockGuard& operator=(const LockGuard&) = delete;
};

enum : uint32_t { PARAM_VOLUME = 0, PARAM_PAN = 1, PARAM_MUTE = 2 };   // reference TrackParameter, track.h:29-34

struct ParamMsg {       // TrackMessage::ParamChange (track.h:75-90): parameter id + value as double
  uint32_t id;
  double value;
};

// The track's parameter-message queue — the contract of the reference's ConcurrentRingBuffer<TrackMessage> (core/queue.h:
// 142-196) as Track uses it (track.cpp:22-27 set_capacity(64)): one producer (UI thread), one consumer (audio thread), 63
// messages in flight at most, the producer yields while it is full.  Two free-running message counts over a 64-slot array;
// each side keeps its last view of the other's count and looks again only when that view says "full" / "empty", so an
// uncontended push or pop touches one shared cache line, not two.
struct ParamRing {
  static constexpr uint32_t kSlots = 64, kInFlight = kSlots - 1;
  struct alignas(64) Side {
    std::atomic<uint32_t> count{0};   // messages pushed (producer side) / popped (consumer side) so far
    uint32_t other_seen = 0;          // the opposite count as last read by this side's thread
  };
  Side pushed, popped;
  ParamMsg slot[kSlots];

  bool try_push(const ParamMsg& m) {
    const uint32_t n = pushed.count.load(std::memory_order_relaxed);
    if (n - pushed.other_seen >= kInFlight) {
      pushed.other_seen = popped.count.load(std::memory_order_acquire);
      if (n - pushed.other_seen >= kInFlight) return false;
    }
    slot[n & (kSlots - 1)] = m;
    pushed.count.store(n + 1, std::memory_order_release);
    return true;
  }
  void push(const ParamMsg& m) {
    while (!try_push(m)) std::this_thread::yield();
  }
  bool pop(ParamMsg& m) {
    const uint32_t n = popped.count.load(std::memory_order_relaxed);
    if (popped.other_seen == n) {
      popped.other_seen = pushed.count.load(std::memory_order_acquire);
      if (popped.other_seen == n) return false;
    }
    m = slot[n & (kSlots - 1)];
    popped.count.store(n + 1, std::memory_order_release);
    return true;
  }
};

// math::db_to_linear<float>, reference core/core_math.h:83-89
inline float db_to_linear(float x) {
  if (x <= -72.0f) return 0.0f;
  return std::pow(10.0f, (float)((double)x * 0.05));
}

// calculate_panning_coefs(p, ConstantPower_3db), reference core/panning_law.cpp:9-32
inline void pan_constant_power_3db(float p, float* l, float* r) {
  const double x = 0.5 * ((double)p + 1.0);
  const double left = std::sin(0.5 * std::numbers::pi * (1.0 - x));
  const double right = std::sin(0.5 * std::numbers::pi * x);
  const double boost = std::sqrt(2.0);
  *l = (float)(left * boost);
  *r = (float)(right * boost);
}

struct SampleMeta {        // what the clip edits and the kernel-instance choice need to know of a SampleAsset
  uint32_t format = 0, channels = 0, sample_rate = 0;
  uint64_t count = 0;
  bool used = false;
};

struct HostTrack {
  std::vector<HostClip> clips;          // sorted by min_time (Track::update_clip_ordering, track.cpp:159-180)
  edit::ClipIds clip_ids;                   // Track::clip_allocator (track.h:105): which identity a new clip takes over
  // audio-side parameter_state (track.h:129), touched by the audio thread only
  float volume = 0.0f, pan = 0.0f, pan_coeffs[2] = {0.0f, 0.0f};
  bool mute = false;
  // ui_parameter_state (track.h:128), touched by the UI thread only
  bool ui_solo = false;
  ParamRing msgs;                       // track_msg_queue (track.h:131)
  uint64_t drained = 0;                 // messages the audio thread has taken out of the ring so far
  int32_t bus = -1;
  DPatch patch{};
  const void* plugin = nullptr;         // effect slot (Track::plugin_instance, track.h:124): always empty today
};

// Everything of a wbx_engine that is not device memory.  Methods named *_locked expect the caller to hold
// editor_lock; set_volume / set_pan / set_mute / set_bpm are the lock-free UI-thread entry points.
struct HostSession {
  SpinLock editor_lock;                 // Engine::editor_lock, engine.h:41
  std::vector<std::unique_ptr<HostTrack>> tracks;
  std::vector<SampleMeta> samples;
  uint32_t max_tracks = 0, dst_rate = 48000;
  uint32_t n_buses = 0;
  double ppq = 96.0;                    // engine.h:43
  double playhead = 0.0, playhead_start = 0.0, sample_position = 0.0;
  std::atomic<double> beat_duration{0.5};   // engine.h:45: written by set_bpm without the lock
  std::atomic<bool> playing{false};
  std::atomic<uint32_t> msgs_pending{0};   // parameter messages posted since the audio thread last went through the rings
  bool clips_dirty = true, gains_dirty = true, routing_dirty = true, patches_pending = false;
  bool clips_edited = false;            // a clip list changed since the previous plan (PlanArgs::clips_changed)
  bool any_slow_clip = false;           // a clip the mix kernel cannot stream directly (playback speed > 4096)
  bool any_window_clip = false;         // a clip that is linearly resampled (playback speed != 1)
  bool any_stride_clip = false;         // needs the everything family: fp32 played faster than recorded, resampled integer PCM
  bool any_taps_clip = false;           // ... of those, the ones the hot loop reads with per-frame taps (KIND_STRIDE: speed > 0.999)
  bool any_win16_clip = false;          // 16-bit PCM resampled at a speed up to 0.999 (the 5-sample-window path)
  bool any_other_window_clip = false;   // any other clip at a speed != 1
  bool any_crawl_clip = false;          // (count - offset) / speed may exceed 2^32: every block owns a plan template
  // the one playback speed of all resampled clips seen so far (44.1 kHz clips in a 48 kHz session ...), for the mix
  // kernel's hoisted position products: 0 = none yet, < 0 = several / outside the narrow-window range
  double window_speed = 0.0;
  bool rate_flags_sticky = false;       // the flags above still cover the row kinds of an earlier destination rate
  size_t total_clips = 0;
  size_t short_clips = 0;               // clips shorter than one block (gen_rows_hint), valid for block length short_clips_key
  double short_clips_key = -1.0;
  size_t cut_tracks = 0;                // tracks that hold more than one clip (clip boundaries inside blocks are the rule there)
  uint32_t next_clip_uid = 0;
  uint64_t edit_seq = 0;                // locked edits completed so far (UI thread, under the lock)
  uint64_t render_edit_seq = 0;         // edit_seq as the last process / render saw it

  // ---- the load figure of the audio thread: Engine::perf_measurer (engine.h:64; core/timing.h:54-67) ----
  // Engine::process starts a counter (engine.cpp:1577) and ends with perf_measurer.update(its duration in ms,
  // audio_buffer_duration_ms) (:1653); the UI reads get_usage() (ui/control_bar.cpp:54).  Here: the wall time of one
  // wbx_engine_process call — launch, device pass, copy-out — against the block's period.  One writer (the audio thread),
  // any reader; the arithmetic lives in static functions so that tests hold it to the reference's bit for bit.
  std::atomic<double> perf_usage{0.0};
  double last_block_ms = 0.0;               // the duration the last update was fed (under the editor lock)
  static double buffer_period_ms(uint32_t buffer_size, uint32_t sample_rate) {   // engine.cpp:52 through audio_io.h:187-195
    constexpr double kHundredNsPerSecond = 1e7;
    const double units = kHundredNsPerSecond * (buffer_size / (double)sample_rate);
    const int64_t period = (int64_t)(units + (units < 0.0 ? -0.5 : 0.5));         // math::round: truncation of x +- 0.5
    return 1000.0 * (double)period / kHundredNsPerSecond;
  }
  static double perf_step(double usage, double duration_ms, double period_ms) {  // timing.h:57-61
    const double share = duration_ms / period_ms;
    return usage + 0.25 * (share - usage);
  }
  static double perf_clamped(double usage) {                                       // timing.h:64-66 (math::clamp to [0, 1])
    const double capped = usage < 1.0 ? usage : 1.0;
    return capped > 0.0 ? capped : 0.0;
  }
  void perf_update(double duration_ms, uint32_t buffer_size) {
    last_block_ms = duration_ms;
    perf_usage.store(perf_step(perf_usage.load(std::memory_order_relaxed), duration_ms, buffer_period_ms(buffer_size, dst_rate)),
                     std::memory_order_release);
  }
  double perf_get_usage() const { return perf_clamped(perf_usage.load(std::memory_order_acquire)); }

  double uniform_window_speed() const { return window_speed > 0.0 ? window_speed : 0.0; }
  uint32_t n_tracks() const { return (uint32_t)tracks.size(); }
  bool valid_track(uint32_t t) const { return t < tracks.size(); }
  bool valid_sample(uint32_t s) const { return s < samples.size() && samples[s].used; }
  double rate_of_sample(uint32_t s) const { return (double)samples[s].sample_rate; }

  // ---- UI thread, no lock (track.cpp:47-79) ----
  // (msgs_pending: "some ring may hold a message" — the audio thread looks through the rings of all tracks only then; at 4096
  //  tracks that scan, two cache lines per ring, was ~10 us of every callback before the launch)
  void post(uint32_t t, const ParamMsg& m) {
    tracks[t]->msgs.push(m);
    msgs_pending.fetch_add(1u, std::memory_order_release);
  }
  void set_volume(uint32_t t, float db) { post(t, {PARAM_VOLUME, (double)db_to_linear(db)}); }
  void set_pan(uint32_t t, float pan) { post(t, {PARAM_PAN, (double)pan}); }
  void set_mute(uint32_t t, bool mute) { post(t, {PARAM_MUTE, (double)(mute ? 1 : 0)}); }
  void set_bpm(double bpm) { beat_duration.store(60.0 / bpm, std::memory_order_release); }   // engine.cpp:24-30

  // ---- edits, under the lock ----
  void note_edit_locked() { edit_seq++; }

  // engine.cpp:200-208; Track::Track (track.cpp:22-27) sends its defaults — 0 dB, centre, unmuted — through the
  // message ring like any later change: the audio-side parameter_state stays zero until the first block drains them
  uint32_t add_track_locked() {
    tracks.emplace_back(new HostTrack());
    tracks.back()->clip_ids.fresh = &next_clip_uid;
    const uint32_t t = (uint32_t)tracks.size() - 1;
    set_volume(t, 0.0f);
    set_pan(t, 0.0f);
    set_mute(t, false);
    clips_dirty = routing_dirty = gains_dirty = true;
    return t;
  }

  // new track i = old track order[i]; the Track objects keep their identity (rings included), as the reference's
  // vector of Track* does (engine.cpp:210-243)
  void permute_tracks_locked(const std::vector<uint32_t>& order) {
    std::vector<std::unique_ptr<HostTrack>> moved(order.size());
    for (size_t i = 0; i < order.size(); i++) moved[i] = std::move(tracks[order[i]]);
    tracks = std::move(moved);
    clips_dirty = gains_dirty = routing_dirty = true;
    recount_clips();
  }

  // Engine::solo_track, engine.cpp:245-262 (UI thread; it only pushes mute messages)
  void solo_track(uint32_t slot) {
    bool mute = false;
    if (tracks[slot]->ui_solo) {
      tracks[slot]->ui_solo = false;
    } else {
      tracks[slot]->ui_solo = true;
      set_mute(slot, false);
      mute = true;
    }
    for (uint32_t i = 0; i < tracks.size(); i++) {
      if (i == slot) continue;
      tracks[i]->ui_solo = false;
      set_mute(i, mute);
    }
  }

  void recount_clips() {
    short_clips_key = -1.0;   // (the count of clips shorter than a block is taken again by the next render)
    total_clips = 0;
    cut_tracks = 0;
    for (auto& tr : tracks) {
      total_clips += tr->clips.size();
      if (tr->clips.size() > 1) cut_tracks++;
    }
  }

  // which clips the hot loop can stream directly, and how much plan storage a render may need
  void note_clip(const DClip& c) {
    const SampleMeta& smp = samples[c.sample];
    const double ps = ((double)smp.sample_rate / (double)dst_rate) * c.speed;   // sampler.h:24
    if (!(ps > 0.0 && ps <= 4096.0)) any_slow_clip = true;
    if (ps != 1.0) {
      any_window_clip = true;
      if (smp.format == FMT_I16 && ps > 0.0 && ps <= 0.999)
        any_win16_clip = true;
      else
        any_other_window_clip = true;
      if (!(ps >= 0.67 && ps <= 0.999) || (window_speed != 0.0 && window_speed != ps))
        window_speed = -1.0;
      else
        window_speed = ps;
    }
    if ((ps > 0.999 || smp.format != FMT_F32) && ps != 1.0) any_stride_clip = true;
    if (ps > 0.999 && ps != 1.0) any_taps_clip = true;
    // BlockWalker::stream / plan_steady_run leave the shared-template path when (count - offset) >= speed * 2^32
    if (!(ps * 4294967040.0 > (double)smp.count)) any_crawl_clip = true;
  }

  void rederive_clip_flags() {
    any_slow_clip = any_window_clip = any_stride_clip = any_crawl_clip = any_taps_clip = false;
    any_win16_clip = any_other_window_clip = false;
    window_speed = 0.0;
    for (auto& t : tracks)
      for (auto& hc : t->clips) note_clip(hc.d);
  }

  // A new destination rate (Engine::set_audio_channel_config on a live engine).  The flags above are what the mix
  // kernel's instance and chunk modes are chosen from — a promise about every row the sequencer can emit.  Clips that are
  // PLAYING keep the playback speed their sampler was reset with (Sampler::reset_state ran with the old rate, sampler.h:
  // 18-27; DTrackState holds it), so while the transport runs the promise must cover the old rate's row kinds as well:
  // the old flags stay OR-ed in and no single resampling ratio is assumed, until a stop re-triggers every track.
  void set_dst_rate_locked(uint32_t rate) {
    const bool changed = rate != dst_rate;
    const bool o_slow = any_slow_clip, o_win = any_window_clip, o_stride = any_stride_clip, o_crawl = any_crawl_clip, o_taps = any_taps_clip;
    const bool o_w16 = any_win16_clip, o_other = any_other_window_clip;
    dst_rate = rate;
    rederive_clip_flags();
    if (changed && (playing.load(std::memory_order_relaxed) || rate_flags_sticky)) {
      any_slow_clip |= o_slow;
      any_window_clip |= o_win;
      any_stride_clip |= o_stride;
      any_taps_clip |= o_taps;
      any_crawl_clip |= o_crawl;
      any_win16_clip |= o_w16;
      any_other_window_clip |= o_other;
      if (any_window_clip) window_speed = -1.0;
      rate_flags_sticky = true;
    }
  }

  // Track::find_next_clip over the host copy (track.cpp:182-213)
  static bool find_next_clip(const HostTrack& t, double time_pos, uint32_t* idx) {
    if (t.clips.empty()) return false;
    if (t.clips.back().d.max_time < time_pos) return false;
    *idx = edit::lower_bound_max(t.clips, time_pos);
    return true;
  }

  // Track::reset_playback_state, track.cpp:220-232
  void reset_playback_state(HostTrack& t, double time_pos, bool refresh_voices) {
    if (!refresh_voices) {
      uint32_t idx = 0;
      const bool has = find_next_clip(t, time_pos, &idx);
      t.patch.flags |= PATCH_CLIPIDX;
      t.patch.has_clip_idx = has ? 1u : 0u;
      t.patch.clip_idx = idx;
    }
    t.patch.flags |= PATCH_REFRESH;
    t.patch.refresh_voice = refresh_voices ? 1u : 0u;
    patches_pending = true;
  }

  // after a clip-list edit: Track::update_clip_ordering + reset_playback_state(playhead, true)
  // (engine.cpp:360,395,405,416,426,437,449,459,473)
  void finish_edit(HostTrack& t) {
    edit::update_clip_ordering(t.clips, t.clip_ids);
    reset_playback_state(t, playhead, true);
    clips_dirty = true;
    clips_edited = true;
    recount_clips();
  }

  // Engine::add_audio_clip -> add_to_cliplist, engine.cpp:293-309, :409-461.  A clip that overlaps existing ones trims,
  // splits or deletes them through reserve_track_region (engine.cpp:478-569), as the reference does.
  void add_audio_clip_locked(uint32_t track, double min_time, double max_time, double start_offset, uint32_t sample,
                             double speed, float gain) {
    HostTrack& t = *tracks[track];
    const double bd = beat_duration.load(std::memory_order_relaxed);
    const bool empty = t.clips.empty();
    const bool back = !empty && t.clips.back().d.max_time < min_time;
    const bool front = !empty && !back && t.clips.front().d.min_time > max_time;
    ClipQuery q{};
    const uint32_t uid = t.clip_ids.take();   // engine.cpp:302: the Clip object exists before the list is touched
    if (!empty && !back && !front && edit::query_clip_by_range(t.clips, min_time, max_time, &q))
      edit::reserve_track_region(t.clips, q.first, q.last, min_time, max_time, 0u, bd,
                                 [&](uint32_t smp) { return rate_of_sample(smp); }, t.clip_ids);
    HostClip c{};
    c.d.min_time = min_time;
    c.d.max_time = max_time;
    c.d.start_offset = start_offset;
    c.d.speed = speed;
    c.d.gain = gain;
    c.d.sample = sample;
    c.d.internal_state_changed = 0;
    c.d.uid = uid;
    c.flag_dirty = true;   // (a new object: no live device flag belongs to it, whatever its id named before)
    t.clips.push_back(c);
    note_clip(c.d);
    finish_edit(t);
  }

  // Engine::move_clip, engine.cpp:346-363
  void move_clip_locked(uint32_t track, uint32_t clip, double relative_pos) {
    if (relative_pos == 0.0) return;
    HostTrack& t = *tracks[track];
    const double bd = beat_duration.load(std::memory_order_relaxed);
    const uint32_t uid = t.clips[clip].d.uid;
    double mn, mx;
    edit::calc_move_clip(t.clips[clip].d.min_time, t.clips[clip].d.max_time, relative_pos, 0.0, &mn, &mx);
    ClipQuery q{};
    if (edit::query_clip_by_range(t.clips, mn, mx, &q))
      edit::reserve_track_region(t.clips, q.first, q.last, mn, mx, uid, bd, [&](uint32_t smp) { return rate_of_sample(smp); },
                                 t.clip_ids);
    for (auto& c : t.clips)
      if (c.d.uid == uid) {
        c.d.min_time = mn;
        c.d.max_time = mx;
        c.d.internal_state_changed = 1;
        c.flag_dirty = true;
      }
    finish_edit(t);
  }

  // Engine::resize_clip, engine.cpp:365-398
  void resize_clip_locked(uint32_t track, uint32_t clip, double relative_pos, double resize_limit, double min_length,
                          bool left_side, bool shift, bool stretch) {
    if (relative_pos == 0.0) return;
    HostTrack& t = *tracks[track];
    const double bd = beat_duration.load(std::memory_order_relaxed);
    const DClip c0 = t.clips[clip].d;
    const SampleMeta& smp = samples[c0.sample];
    const edit::ResizeResult r =
        edit::calc_resize_clip(c0.min_time, c0.max_time, c0.start_offset, c0.speed, (double)smp.sample_rate, (double)smp.count,
                               relative_pos, resize_limit, min_length, c0.min_time, bd, left_side, shift, stretch, false);
    ClipQuery q{};
    if (edit::query_clip_by_range(t.clips, r.min, r.max, &q))
      edit::reserve_track_region(t.clips, q.first, q.last, r.min, r.max, c0.uid, bd,
                                 [&](uint32_t s2) { return rate_of_sample(s2); }, t.clip_ids);
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
        note_clip(c.d);
      }
    finish_edit(t);
  }

  // Engine::delete_clip, engine.cpp:400-407
  void delete_clip_locked(uint32_t track, uint32_t clip) {
    HostTrack& t = *tracks[track];
    t.clips[clip].deleted = true;
    finish_edit(t);
  }

  // Engine::delete_region, engine.cpp:463-475
  void delete_region_locked(uint32_t track, double min, double max) {
    HostTrack& t = *tracks[track];
    ClipQuery q{};
    if (!edit::query_clip_by_range(t.clips, min, max, &q)) return;
    edit::reserve_track_region(t.clips, q.first, q.last, min, max, 0u, beat_duration.load(std::memory_order_relaxed),
                               [&](uint32_t smp) { return rate_of_sample(smp); }, t.clip_ids);
    finish_edit(t);
  }

  // Engine::set_clip_gain, engine.cpp:1460-1464
  void set_clip_gain_locked(uint32_t track, uint32_t clip, float gain) {
    tracks[track]->clips[clip].d.gain = gain;
    clips_dirty = true;
    clips_edited = true;
  }

  void set_playhead_position_locked(double beat) {   // engine.cpp:32-41
    playhead_start = beat;
    playhead = beat;
  }

  void play_locked() {   // engine.cpp:68-80
    for (auto& t : tracks) reset_playback_state(*t, playhead_start, false);
    sample_position = 0;
    playing.store(true, std::memory_order_relaxed);
  }

  void stop_locked() {   // engine.cpp:82-93, Track::stop track.cpp:249-256
    playing.store(false, std::memory_order_relaxed);
    playhead = playhead_start;
    for (auto& t : tracks) t->patch.flags |= PATCH_STOP;
    patches_pending = true;
    if (rate_flags_sticky) {   // every sampler restarts at the current rate from here on (set_dst_rate_locked)
      rate_flags_sticky = false;
      rederive_clip_flags();
    }
  }

  bool sample_referenced(uint32_t sample) const {
    for (auto& t : tracks)
      for (auto& c : t->clips)
        if (c.d.sample == sample) return true;
    return false;
  }

  // ---- audio thread, under the lock: what a process / render takes from the shared state ----

  // process_track_messages (track.cpp:773-779) + the parameter application of Track::process (track.cpp:618-643).
  // Returns true when a per-track factor fl(volume * pan_coeffs[c]) (track.cpp:728-731) may have changed.
  bool drain_params_locked() {
    // nothing posted since the last look (a message whose count arrives after this exchange is taken by the next block —
    // as if it had been posted a moment later; one that is drained before its count arrives costs the next block a scan)
    if (msgs_pending.exchange(0u, std::memory_order_acq_rel) == 0u) return gains_dirty;
    for (auto& tp : tracks) {
      HostTrack& t = *tp;
      ParamMsg m;
      bool any = false;
      while (t.msgs.pop(m)) {
        any = true;
        t.drained++;
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
      if (any) gains_dirty = true;
    }
    return gains_dirty;
  }

  void build_gains_locked(std::vector<float>& g) {
    const uint32_t N = n_tracks();
    g.resize((size_t)N * 2);
    for (uint32_t t = 0; t < N; t++) {
      const HostTrack& tr = *tracks[t];
      const float volume = tr.mute ? 0.0f : tr.volume;
      g[2 * t + 0] = volume * tr.pan_coeffs[0];
      g[2 * t + 1] = volume * tr.pan_coeffs[1];
    }
    gains_dirty = false;
  }

  // Clip::internal_state_changed is cleared by the sequencer where the plan runs (track.cpp:373,392,418): before the
  // clip table is replaced, take the live flags back for every clip no edit has touched since the last upload
  void merge_live_flags_locked(const DClip* live, size_t n) {
    std::vector<uint32_t> flag(next_clip_uid + 1, 2u);
    for (size_t i = 0; i < n; i++)
      if (live[i].uid < flag.size()) flag[live[i].uid] = live[i].internal_state_changed;
    for (auto& tr : tracks)
      for (auto& hc : tr->clips)
        if (!hc.flag_dirty && hc.d.uid < flag.size() && flag[hc.d.uid] != 2u) hc.d.internal_state_changed = flag[hc.d.uid];
  }

  void flatten_clips_locked(std::vector<DClip>& flat, std::vector<uint32_t>& first) {
    const uint32_t N = n_tracks();
    first.assign(N + 1, 0);
    flat.clear();
    for (uint32_t t = 0; t < N; t++) {
      first[t] = (uint32_t)flat.size();
      for (auto& hc : tracks[t]->clips) {
        flat.push_back(hc.d);
        hc.flag_dirty = false;
      }
    }
    first[N] = (uint32_t)flat.size();
    clips_dirty = false;
  }

  void take_patches_locked(DPatch* dst) {
    const uint32_t N = n_tracks();
    for (uint32_t t = 0; t < N; t++) {
      dst[t] = tracks[t]->patch;
      tracks[t]->patch = DPatch{};
    }
    patches_pending = false;
  }

  // plan storage a render of K blocks may need: pre-render rows (track-blocks with a clip start / end inside them,
  // all blocks of fast-forward clips) and templates (one per block with events + one per steady run + one per run of
  // a finished clip; every block when a crawling clip is present)
  // track-blocks of a render that can hold a clip start / end (every block for fast-forward clips)
  size_t boundary_blocks_hint(uint32_t K) const {
    const size_t all = (size_t)K * n_tracks();
    return any_slow_clip ? all : std::min(all, 4 * total_clips + 2 * (size_t)n_tracks() + 64);
  }
  // pre-render rows a render may queue.  Without masked rows: every boundary block.  With them (the hot loop takes blocks
  // of one or two stream calls itself) only a block with THREE or more calls is queued, and that takes a clip shorter than
  // a block: two rows per such clip (it can straddle a block seam) + slack for the wrap-around quirks of track.cpp:359-361.
  // — A session of millions of clips used to reserve a scratch row for every boundary block (4 KiB each: 35 GB per plan
  // buffer for c3 cut into 5.3-block clips at 2048 blocks) that the masked-row path never touches.
  size_t gen_rows_hint(uint32_t K, uint32_t masked_level = 0, double block_beats = 0.0) {
    const size_t all = (size_t)K * n_tracks();
    if (!masked_level || any_slow_clip || !(block_beats > 0.0)) return boundary_blocks_hint(K);
    if (short_clips_key != block_beats) {
      short_clips = 0;
      for (auto& tr : tracks)
        for (auto& c : tr->clips)
          if (c.d.max_time - c.d.min_time < block_beats) short_clips++;
      short_clips_key = block_beats;
    }
    return std::min(all, 2 * short_clips + (size_t)n_tracks() + 64);
  }
  // (a ROW_PAIR block takes two templates; every track may strand part of a reservation of `reserve` templates)
  // tracks per wave of the plan kernel: a session cut into many clips meets a clip boundary every few blocks on every
  // track, and a wave runs the (long) boundary path whenever ANY of its tracks does
  uint32_t plan_lanes_knob = 0;   // WBX_PLAN_LANES as the engine read it when it was created (0: unset)
  uint32_t plan_lanes(uint32_t K) const {
    (void)K;
    return plan_lanes_knob ? plan_lanes_knob : 64u;
  }
  static uint32_t template_reserve(uint32_t K) { return K >= 64u ? 32u : K >= 8u ? 8u : 1u; }
  // (lanes: lanes per track of the sequencer — more than one when it is cut along the time axis; every lane may strand a
  //  reservation and splits a steady run at its seam)
  size_t template_hint(uint32_t K, uint32_t lanes = 1u) const {
    const size_t all = (size_t)K * n_tracks();
    const size_t stranded = (size_t)((lanes > 1u ? 8u : template_reserve(K)) + (lanes > 1u ? 3u : 1u)) * n_tracks() * lanes;
    if (any_crawl_clip) return 2 * all + stranded;
    return std::min(2 * all, 2 * boundary_blocks_hint(K) + 3 * (size_t)n_tracks() + 64) + stranded;
  }

  // the transport advance of Engine::process for K blocks (engine.cpp:1578-1585, :1619-1623), the arithmetic the plan
  // kernel performs as well, so both sides hold the same playhead / sample_position bits
  void advance_transport_locked(uint32_t K, uint32_t block_frames, double bd) {
    const double sample_rate = (double)dst_rate;
    const bool pl = playing.load(std::memory_order_relaxed);
    double ph = playhead, sp = sample_position;
    for (uint32_t b = 0; b < K; b++) {
      const double buffer_duration = (double)block_frames / sample_rate;
      const double buffer_duration_in_beats = buffer_duration / bd;
      const double next_playhead_pos = ph + buffer_duration_in_beats;
      if (pl) {
        const double sec = buffer_duration_in_beats * bd;       // beat_to_samples, core_math.h:209-212: two
        sp += sec * sample_rate;                                // separately rounded multiplies
        ph = next_playhead_pos;
      }
    }
    playhead = ph;
    sample_position = sp;
  }
};

}  // namespace wbx
