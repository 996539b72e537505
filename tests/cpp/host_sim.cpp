// host_sim.cpp — TEST HARNESS: the host side of layer 2 without a device.
//
// Compiles the product's own host headers — wbx_host.h (session state, SPSC parameter rings, editor lock, clip edits)
// and wbx_seq.h (the clip sequencer: the very source plan_kernel runs one lane per track) — with plain g++ and runs the
// sequencer track by track on the CPU.  No per-sample work exists here and nothing of this is linked into libwbx.so;
// it gives the seek math and the clip-edit logic CPU-side coverage against the oracle (tests/test_host_sim.py) and,
// built with -fsanitize=thread -DHOST_SIM_MAIN, a ThreadSanitizer run of the UI-thread / audio-thread contract.
//
//   g++ -std=c++20 -O2 -ffp-contract=off -shared -fPIC host_sim.cpp -o libwbxhostsim.so
//   g++ -std=c++20 -O1 -g -ffp-contract=off -fsanitize=thread -DHOST_SIM_MAIN host_sim.cpp -o host_tsan -lpthread
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/wbx.h"
#include "../../whitebox_amd/csrc/wbx_host.h"
#include "../../whitebox_amd/csrc/wbx_seq.h"

using namespace wbx;

struct HostSim {
  HostSession hs;
  uint32_t block_frames = 512, channels = 2, max_blocks = 1;
  // what lives in HBM in the product
  std::vector<DSample> samples;
  std::vector<DClip> clips;
  std::vector<uint32_t> clip_first;
  std::vector<DTrackState> state;
  std::vector<float> gains;
  std::vector<DPatch> patch;
  // the plan of the last render
  std::vector<DRow> rows;
  std::vector<DTrackBlock> tmpl;
  std::vector<DSeg> pool;
  std::vector<uint32_t> gen_list;
  uint32_t counters[4] = {0, 0, 0, 0};
  uint32_t last_K = 0, last_N = 0;
  bool clips_uploaded = false;
  uint32_t masked_rows = 0;   // plan as for a mix instance that takes partial rows / ROW_PAIRs in its hot loop
  // the sequencer cut along the time axis (wbx_seq.h plan_segment / plan_check_seams): blocks per segment (0: one walk per
  // track), and what the seam check found — renders planned that way, tracks with a seam that did not hold, segments redone
  uint32_t seg_len = 0;
  bool table_flags = false;
  uint32_t flags_left = 0;    // set internal_state_changed flags of the table (the sequencer counts down as it clears them)
  uint64_t seg_renders = 0, seg_tracks_redone = 0, seg_segments_redone = 0, seg_lanes = 0;
};

extern "C" {

HostSim* hsim_create(uint32_t max_tracks, uint32_t max_blocks, uint32_t block_frames, uint32_t channels, uint32_t sample_rate) {
  HostSim* s = new HostSim();
  s->hs.max_tracks = max_tracks;
  s->hs.dst_rate = sample_rate;
  s->block_frames = block_frames;
  s->channels = channels;
  s->max_blocks = max_blocks;
  return s;
}
void hsim_destroy(HostSim* s) { delete s; }
void hsim_set_masked_rows(HostSim* s, uint32_t on) { s->masked_rows = on; }
void hsim_set_segments(HostSim* s, uint32_t seg_len) { s->seg_len = seg_len; }
void hsim_segment_stats(HostSim* s, uint64_t* out4) {
  out4[0] = s->seg_renders;
  out4[1] = s->seg_tracks_redone;
  out4[2] = s->seg_segments_redone;
  out4[3] = s->seg_lanes;
}

// (row flags, template kind) of every (block, track) of the last render — how the sequencer classified each track-block
int hsim_row_kinds(HostSim* s, uint32_t* flags, uint32_t* kinds, size_t cap) {
  const size_t n = (size_t)s->last_K * s->last_N;
  if (cap < n) return WBX_ERR_INVALID;
  for (size_t i = 0; i < n; i++) {
    flags[i] = s->rows[i].flags;
    kinds[i] = s->rows[i].tmpl < s->tmpl.size() ? s->tmpl[s->rows[i].tmpl].kind : 0xFFu;
  }
  return WBX_OK;
}

int hsim_set_bpm(HostSim* s, double bpm) {
  if (!(bpm > 0.0)) return WBX_ERR_INVALID;
  s->hs.set_bpm(bpm);
  return WBX_OK;
}
int hsim_set_playhead_position(HostSim* s, double beat) {
  LockGuard g(s->hs.editor_lock);
  s->hs.set_playhead_position_locked(beat);
  s->hs.note_edit_locked();
  return WBX_OK;
}
int hsim_add_track(HostSim* s, uint32_t* out) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (s->hs.n_tracks() >= s->hs.max_tracks) return WBX_ERR_INVALID;
  const uint32_t t = s->hs.add_track_locked();
  if (out) *out = t;
  return WBX_OK;
}
int hsim_track_set_volume(HostSim* s, uint32_t t, float db) {
  if (!s->hs.valid_track(t)) return WBX_ERR_INVALID;
  s->hs.set_volume(t, db);
  return WBX_OK;
}
int hsim_track_set_pan(HostSim* s, uint32_t t, float pan) {
  if (!s->hs.valid_track(t)) return WBX_ERR_INVALID;
  s->hs.set_pan(t, pan);
  return WBX_OK;
}
int hsim_track_set_mute(HostSim* s, uint32_t t, int m) {
  if (!s->hs.valid_track(t)) return WBX_ERR_INVALID;
  s->hs.set_mute(t, m != 0);
  return WBX_OK;
}
int hsim_solo_track(HostSim* s, uint32_t t) {
  if (!s->hs.valid_track(t)) return WBX_ERR_INVALID;
  s->hs.solo_track(t);
  return WBX_OK;
}

static int permute(HostSim* s, const std::vector<uint32_t>& order) {
  std::vector<DTrackState> st2(order.size());
  for (size_t i = 0; i < order.size(); i++) st2[i] = order[i] < s->state.size() ? s->state[order[i]] : DTrackState{};
  s->state = st2;
  s->hs.permute_tracks_locked(order);
  return WBX_OK;
}
int hsim_delete_track(HostSim* s, uint32_t slot) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (!s->hs.valid_track(slot)) return WBX_ERR_INVALID;
  std::vector<uint32_t> order;
  for (uint32_t i = 0; i < s->hs.n_tracks(); i++)
    if (i != slot) order.push_back(i);
  return permute(s, order);
}
int hsim_move_track(HostSim* s, uint32_t from, uint32_t to) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (!s->hs.valid_track(from) || !s->hs.valid_track(to)) return WBX_ERR_INVALID;
  if (from == to) return WBX_OK;
  std::vector<uint32_t> order(s->hs.n_tracks());
  for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
  order.erase(order.begin() + from);
  order.insert(order.begin() + to, from);
  return permute(s, order);
}

// a sample asset: only what the sequencer reads of it (rate, length, format); the channel pointers are fake addresses
int hsim_add_sample(HostSim* s, int format, uint32_t channels, uint32_t sample_rate, uint64_t frames, uint32_t* out) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  const uint32_t id = (uint32_t)s->samples.size();
  DSample d{};
  d.ch[0] = (const void*)(((uintptr_t)id + 1u) << 32);
  d.ch[1] = channels > 1 ? (const void*)((((uintptr_t)id + 1u) << 32) + (1u << 31)) : d.ch[0];
  d.count = frames;
  d.format = (uint32_t)format;
  d.channels = channels;
  d.sample_rate = sample_rate;
  s->samples.push_back(d);
  s->hs.samples.push_back(SampleMeta{(uint32_t)format, channels, sample_rate, frames, true});
  *out = id;
  return WBX_OK;
}

int hsim_add_audio_clip(HostSim* s, uint32_t track, double mn, double mx, double so, uint32_t sample, double speed, float gain) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (!s->hs.valid_track(track) || !s->hs.valid_sample(sample) || !(mn <= mx)) return WBX_ERR_INVALID;
  s->hs.add_audio_clip_locked(track, mn, mx, so, sample, speed, gain);
  return WBX_OK;
}
int hsim_move_clip(HostSim* s, uint32_t track, uint32_t clip, double rel) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (!s->hs.valid_track(track) || clip >= s->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  s->hs.move_clip_locked(track, clip, rel);
  return WBX_OK;
}
int hsim_resize_clip(HostSim* s, uint32_t track, uint32_t clip, double rel, double limit, double min_length, int left, int shift,
                     int stretch) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (!s->hs.valid_track(track) || clip >= s->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  s->hs.resize_clip_locked(track, clip, rel, limit, min_length, left != 0, shift != 0, stretch != 0);
  return WBX_OK;
}
int hsim_delete_clip(HostSim* s, uint32_t track, uint32_t clip) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (!s->hs.valid_track(track) || clip >= s->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  s->hs.delete_clip_locked(track, clip);
  return WBX_OK;
}
int hsim_delete_region(HostSim* s, uint32_t track, double mn, double mx) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (!s->hs.valid_track(track) || !(mn <= mx)) return WBX_ERR_INVALID;
  s->hs.delete_region_locked(track, mn, mx);
  return WBX_OK;
}
int hsim_set_clip_gain(HostSim* s, uint32_t track, uint32_t clip, float gain) {
  LockGuard g(s->hs.editor_lock);
  s->hs.note_edit_locked();
  if (!s->hs.valid_track(track) || clip >= s->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  s->hs.set_clip_gain_locked(track, clip, gain);
  return WBX_OK;
}
int hsim_clip_count(HostSim* s, uint32_t track, uint32_t* n) {
  LockGuard g(s->hs.editor_lock);
  if (!s->hs.valid_track(track)) return WBX_ERR_INVALID;
  *n = (uint32_t)s->hs.tracks[track]->clips.size();
  return WBX_OK;
}
int hsim_get_clip(HostSim* s, uint32_t track, uint32_t clip, wbx_clip_info* out) {
  LockGuard g(s->hs.editor_lock);
  if (!s->hs.valid_track(track) || clip >= s->hs.tracks[track]->clips.size()) return WBX_ERR_INVALID;
  const DClip& d = s->hs.tracks[track]->clips[clip].d;
  *out = wbx_clip_info{d.min_time, d.max_time, d.start_offset, d.speed, d.gain, d.sample};
  return WBX_OK;
}
int hsim_play(HostSim* s) {
  LockGuard g(s->hs.editor_lock);
  s->hs.play_locked();
  s->hs.note_edit_locked();
  return WBX_OK;
}
int hsim_stop(HostSim* s) {
  LockGuard g(s->hs.editor_lock);
  s->hs.stop_locked();
  s->hs.note_edit_locked();
  return WBX_OK;
}

// the audio thread's render: the host side of wbx_engine_render (wbx_engine.hip render_locked) with the plan kernel's
// work done here, track by track
int hsim_render(HostSim* s, uint32_t K) {
  if (K == 0 || K > s->max_blocks) return WBX_ERR_INVALID;
  LockGuard g(s->hs.editor_lock);
  HostSession& hs = s->hs;
  const uint32_t N = hs.n_tracks(), F = s->block_frames;
  hs.render_edit_seq = hs.edit_seq;
  const double beat_duration = hs.beat_duration.load(std::memory_order_relaxed);
  const bool playing = hs.playing.load(std::memory_order_relaxed);
  s->last_K = K;
  s->last_N = N;
  if (N == 0) {
    hs.advance_transport_locked(K, F, beat_duration);
    return WBX_OK;
  }
  if (hs.drain_params_locked()) hs.build_gains_locked(s->gains);
  if (hs.clips_dirty) {
    if (s->clips_uploaded && !s->clips.empty()) hs.merge_live_flags_locked(s->clips.data(), s->clips.size());
    hs.flatten_clips_locked(s->clips, s->clip_first);
    s->clips_uploaded = true;
    s->flags_left = 0;
    for (const DClip& dc : s->clips) s->flags_left += dc.internal_state_changed != 0 ? 1u : 0u;
    s->table_flags = s->flags_left != 0u;
  }
  if (s->state.size() < N) s->state.resize(N, DTrackState{});
  const DPatch* patch = nullptr;
  if (hs.patches_pending) {
    s->patch.resize(N);
    hs.take_patches_locked(s->patch.data());
    patch = s->patch.data();
  }
  hs.routing_dirty = false;
  s->rows.assign((size_t)K * N, DRow{});
  // (the conditions of wbx_engine.hip plan_segment_length with a forced segment length)
  if (s->table_flags && s->flags_left == 0u) s->table_flags = false;
  const uint32_t L = (s->seg_len && s->seg_len < K && playing && !s->table_flags) ? s->seg_len : 0u;
  const uint32_t S = L ? (K + L - 1u) / L : 1u;
  s->tmpl.assign(hs.template_hint(K, S) + 64, DTrackBlock{});
  s->gen_list.assign(hs.gen_rows_hint(K) + 64, 0u);
  const uint32_t pool_chunks = (uint32_t)std::max<size_t>(1024, (size_t)s->max_blocks * hs.max_tracks / 8);
  s->pool.assign((size_t)pool_chunks * kChunk, DSeg{});
  std::memset(s->counters, 0, sizeof(s->counters));
  PlanArgs a{};
  a.clips = s->clips.data();
  a.clip_first = s->clip_first.data();
  a.samples = s->samples.data();
  a.state = s->state.data();
  a.patch = patch;
  a.gains = s->gains.data();
  a.rows = s->rows.data();
  a.tmpl = s->tmpl.data();
  a.tmpl_count = &s->counters[3];
  a.tmpl_cap = (uint32_t)s->tmpl.size();
  a.pool = s->pool.data();
  a.pool_count = &s->counters[0];
  a.status = &s->counters[1];
  a.gen_list = s->gen_list.data();
  a.gen_count = &s->counters[2];
  a.gen_cap = (uint32_t)s->gen_list.size();
  a.pool_chunks = pool_chunks;
  a.n_tracks = N;
  a.n_blocks = K;
  a.block_frames = F;
  a.channels = s->channels;
  a.sample_rate = (double)hs.dst_rate;
  a.playing = playing ? 1u : 0u;
  a.clips_changed = hs.clips_edited ? 1u : 0u;
  a.flags_left = &s->flags_left;
  hs.clips_edited = false;
  a.masked_rows = s->masked_rows;
  a.tmpl_reserve = HostSession::template_reserve(K);
  a.playhead = hs.playhead;
  a.sample_position = hs.sample_position;
  a.beat_duration = beat_duration;
  std::vector<DBlockTime> times(K);
  block_times(a, times.data());
  if (L) {
    // every lane of plan_seg_kernel, in an order no lane may rely on (segments backwards, tracks forwards), then the seam pass
    a.tmpl_reserve = 8u;
    std::vector<DTrackState> guess((size_t)N * S), ends((size_t)N * S);
    const DBlockTime* tv = times.data();
    for (uint32_t sg = S; sg-- > 0u;)
      for (uint32_t t = 0; t < N; t++) plan_segment(a, t, sg, L, S, tv, &guess[(size_t)t * S + sg], &ends[(size_t)t * S + sg]);
    for (uint32_t t = 0; t < N; t++) {
      const uint32_t bad = plan_check_seams(a, t, S, guess.data(), ends.data(), [](const DTrackState* p) { return *p; });
      if (bad < S) {
        plan_redo_track(a, t, bad, L, tv, ends[(size_t)t * S + bad - 1u]);
        s->seg_tracks_redone += 1u;
        s->seg_segments_redone += S - bad;
      }
    }
    s->seg_renders++;
    s->seg_lanes += (uint64_t)N * (S - 1u);
  } else {
    for (uint32_t t = 0; t < N; t++) plan_track(a, t, times.data());
  }
  hs.advance_transport_locked(K, F, beat_duration);
  return WBX_OK;
}

// plan status bits of the last render (PlanArgs::status) and the number of templates / pre-render rows it used
int hsim_plan_counters(HostSim* s, uint32_t* out4) {
  std::memcpy(out4, s->counters, sizeof(s->counters));
  return WBX_OK;
}
uint32_t hsim_template_capacity(HostSim* s) { return (uint32_t)s->tmpl.size(); }

int hsim_fetch_plan(HostSim* s, wbx_plan_record* out, size_t cap, size_t* n_out) {
  LockGuard g(s->hs.editor_lock);
  const uint32_t used = std::min<uint32_t>(s->counters[0], (uint32_t)(s->pool.size() / kChunk));
  const size_t nt = std::min<size_t>(s->counters[3], s->tmpl.size());
  *n_out = plan_records(s->last_K, s->last_N, s->rows.data(), s->tmpl.data(), nt, s->pool.data(), used, out, cap);
  if (s->counters[1] & 3u) return WBX_ERR_OVERFLOW;
  if (s->counters[1] & 16u) return WBX_ERR_OVERFLOW;
  return WBX_OK;
}

// the gains the mix would use: [N][2] fl(volume * pan_coeffs[c]) after the last render's drain
int hsim_gains(HostSim* s, float* out, uint32_t n_tracks) {
  LockGuard g(s->hs.editor_lock);
  if (n_tracks * 2 > s->gains.size()) return WBX_ERR_INVALID;
  std::memcpy(out, s->gains.data(), (size_t)n_tracks * 2 * sizeof(float));
  return WBX_OK;
}

int hsim_transport(HostSim* s, double* playhead, double* sample_position, int* playing) {
  LockGuard g(s->hs.editor_lock);
  *playhead = s->hs.playhead;
  *sample_position = s->hs.sample_position;
  *playing = s->hs.playing.load() ? 1 : 0;
  return WBX_OK;
}

int hsim_thread_stats(HostSim* s, uint64_t* edits_seen, uint64_t* drained, uint32_t n_tracks) {
  LockGuard g(s->hs.editor_lock);
  if (edits_seen) *edits_seen = s->hs.render_edit_seq;
  if (drained) {
    if (n_tracks > s->hs.n_tracks()) return WBX_ERR_INVALID;
    for (uint32_t t = 0; t < n_tracks; t++) drained[t] = s->hs.tracks[t]->drained;
  }
  return WBX_OK;
}

}  // extern "C"

#ifdef HOST_SIM_MAIN
// ThreadSanitizer driver: one UI thread hammering Track::set_volume / set_pan / set_mute (lock-free rings) and clip
// edits (editor lock) while the audio thread renders block after block — the reference's two-thread contract.  Checks
// on top of TSan's race detection: every message pushed is eventually drained, in order per track (the last value of
// each parameter wins), and the edit counter the audio thread saw never runs backwards.
#include <atomic>
#include <cmath>
int main() {
  const uint32_t N = 24, BLOCKS = 3000;
  HostSim* s = hsim_create(N, 1, 128, 2, 48000);
  hsim_set_bpm(s, 120.0);
  uint32_t smp = 0;
  hsim_add_sample(s, WBX_FMT_F32, 2, 44100, 200000, &smp);
  for (uint32_t t = 0; t < N; t++) {
    uint32_t id;
    hsim_add_track(s, &id);
    for (int k = 0; k < 4; k++) hsim_add_audio_clip(s, t, 0.05 * k + 0.001 * t, 0.05 * k + 0.04, 10.0 * k, smp, k & 1 ? 0.5 : 1.0, 1.0f);
  }
  hsim_play(s);
  std::atomic<bool> stop{false};
  std::atomic<uint64_t> pushed{0};
  float last_db[N];
  for (uint32_t t = 0; t < N; t++) last_db[t] = 0.0f;
  std::thread ui([&] {
    uint64_t i = 0;
    while (!stop.load(std::memory_order_relaxed)) {
      const uint32_t t = (uint32_t)(i * 7u % N);
      const float db = -0.25f * (float)(i % 97u);
      hsim_track_set_volume(s, t, db);
      last_db[t] = db;
      hsim_track_set_pan(s, t, (float)((int)(i % 21u) - 10) * 0.1f);
      if (i % 5u == 0) hsim_track_set_mute(s, t, (int)((i / 5u) & 1u));
      pushed.fetch_add(i % 5u == 0 ? 3 : 2, std::memory_order_relaxed);
      if (i % 3u == 0) {
        uint32_t n = 0;
        hsim_clip_count(s, t, &n);
        if (n) {
          if (i % 9u == 0)
            hsim_move_clip(s, t, (uint32_t)(i % n), ((i & 1u) ? 1.0 : -1.0) * 0.003);
          else
            hsim_set_clip_gain(s, t, (uint32_t)(i % n), 0.5f + 0.01f * (float)(i % 50u));
        }
      }
      if (i % 997u == 0) hsim_set_bpm(s, 100.0 + (double)(i % 40u));
      i++;
    }
  });
  uint64_t prev_seen = 0;
  bool ok = true;
  for (uint32_t b = 0; b < BLOCKS; b++) {
    if (hsim_render(s, 1) != WBX_OK) ok = false;
    uint64_t seen = 0;
    hsim_thread_stats(s, &seen, nullptr, 0);
    if (seen < prev_seen) ok = false;
    prev_seen = seen;
    if (b == BLOCKS / 2) {
      hsim_stop(s);
      hsim_play(s);
    }
  }
  stop.store(true);
  ui.join();
  hsim_render(s, 1);   // drains what the UI thread pushed last
  uint64_t drained[N], total = 0;
  hsim_thread_stats(s, nullptr, drained, N);
  for (uint32_t t = 0; t < N; t++) total += drained[t];
  // Track::Track pushes three messages per track (track.cpp:22-27)
  if (total != pushed.load() + 3ull * N) {
    std::printf("drained %llu of %llu messages\n", (unsigned long long)total, (unsigned long long)(pushed.load() + 3ull * N));
    ok = false;
  }
  for (uint32_t t = 0; t < N; t++)
    if (s->hs.tracks[t]->volume != db_to_linear(last_db[t])) ok = false;
  hsim_destroy(s);
  std::printf(ok ? "host_tsan ok\n" : "host_tsan FAILED\n");
  return ok ? 0 : 1;
}
#endif
