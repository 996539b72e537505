// ref_engine_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// A script-driven door to the REFERENCE'S OWN clip sequencer and block driver: Track::process_event (engine/track.cpp:258-451,
// the seek math), find_next_clip / reset_playback_state / update_clip_ordering, Track::process (:587-736), Engine::process
// (engine/engine.cpp:1576-1654), set_bpm / set_playhead_position / set_audio_channel_config, add_track / delete_track /
// move_track / solo_track, add_audio_clip / add_to_cliplist / move_clip / resize_clip / delete_clip / set_clip_gain, with the
// reference's Pool<Clip>, Vector, Sampler (dsp/sampler.cpp), panning law and VUMeter underneath.
//
// How it is built without third-party stand-ins (oracle/Makefile, target _ref/wbref_engine):
//   * engine/track.cpp, engine/engine.cpp, engine/assets_table.cpp, engine/vu_meter.h, engine/audio_record.h include
//     core/debug.h = third-party spdlog, absent from the image.  Nothing is written in spdlog's place.  The recipe cuts, at
//     FUNCTION BOUNDARIES and out of the files where they lie, the regions of those sources that hold no `Log::` line (build
//     outputs under _ref/eng/, git-ignored, never committed), and this file compiles those texts UNMODIFIED, in the reference's
//     own order, against the reference's own headers.  track.cpp's per-function logging is behind its WB_DBG_LOG_* macros,
//     which the file defines only `#ifndef _NDEBUG`: the build takes that configuration (-D_NDEBUG), so Track::process and
//     process_event are whole.
//   * What the cuts leave out, because an unconditional Log line sits inside the function: Engine::play / stop / record /
//     stop_record, Engine::reserve_track_region (engine.cpp:478-569), add_plugin_to_track, Track::process_track_messages
//     (track.cpp:773-813) and the plugin edit callbacks, SampleTable/MidiTable::shutdown.  Of these the path needs:
//       - play / stop: `harness_play` / `harness_stop` below restate their statements (engine.cpp:70-79, 83-91);
//       - Track::process_track_messages: defined below with its ParamChange case only (track.cpp:777-779, one statement);
//       - the plugin callbacks: their addresses are taken by track.h; defined below as traps (never reached: no plugin);
//       - reserve_track_region: NOT provided.  An edit that would reach it (a clip landing on other clips) is REFUSED by the
//         driver before the reference's code is entered (it asks the reference's own Track::query_clip_by_range), and the
//         differential test skips that edit on the oracle side too.  Overlap trimming therefore stays KAT-pinned
//         (tests/test_oracle_kat.py) with its arithmetic pinned through clip_edit.h (libwbref.so).
//   * Functions of the cut regions that call code from absent libraries (Sample::load_file -> libsndfile,
//     WaveformVisual::create / ~WaveformVisual -> renderer, pm_close_plugin -> VST3 host, load_notes_from_file -> midi-parser,
//     AudioRecordQueue's writer) are never reached by a script; their symbols stay UNRESOLVED in the executable
//     (-Wl,--unresolved-symbols=ignore-all, lazy binding) rather than being given made-up bodies.
//
// Protocol: argv[1] = script (text, one operation per line), argv[2] = sample data (raw planar channels), argv[3] = result
// file (binary records, layout in tests/ref_engine.py).  Every line is answered by one record, so the test knows which edits the
// reference took and which the driver refused.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <ctime>
#include <functional>
#include <mutex>
#include <numbers>
#include <optional>
#include <random>
#include <string>
#include <thread>
#include <vector>

// ---- what engine/track.h includes (track.h:3-23), vu_meter.h's body cut in place of the header ------------------------------
#include "engine/audio_param.h"
#include "engine/clip.h"
#include "core/audio_buffer.h"
#include "core/bit_manipulation.h"
#include "core/memory.h"
#include "core/vector.h"
#include "dsp/param_queue.h"
#include "dsp/sampler.h"
#include "engine/etypes.h"
#include "engine/event.h"
#include "engine/event_list.h"
#include "engine/midi_voice.h"
#include "plughost/plugin_interface.h"
#include "engine/test_synth.h"
#include "engine/track_input.h"
#include "core/core_math.h"
#include "_ref/eng/vu_meter_h_body.inc"      // namespace wb { enum LevelMeterColorMode; struct VUMeter } (vu_meter.h:9-)
#include "_ref/eng/track_h_body.inc"         // namespace wb { ... struct Track ... } (track.h:25-)
// ---- what engine/engine.h includes (engine.h:3-13), audio_record.h's body cut in place of the header ------------------------
#include "_ref/eng/audio_record_h_body.inc"  // (audio_record.h includes core/debug.h for a commented-out line)
#include "engine/clip_edit.h"
#include "core/common.h"
#include "core/thread.h"
#include "core/timing.h"
#include "plughost/plugin_manager.h"
#include "_ref/eng/engine_h_body.inc"        // namespace wb { ... struct Engine ...; extern Engine g_engine; }
// ---- what the three .cpp files include besides ------------------------------------------------------------------------------
#include "engine/assets_table.h"
#include "core/algorithm.h"
#include "core/midi_file.h"
#include "extern/xxhash.h"
#include "core/panning_law.h"
#include "core/queue.h"
#include "dsp/dsp_ops.h"
#include "engine/audio_io.h"

// engine/assets_table.cpp: everything but SampleTable::shutdown / MidiTable::shutdown
#include "_ref/eng/assets_r1.inc"   // namespace wb { ... SampleAsset::release ... SampleTable::destroy_unused
#include "_ref/eng/assets_r2.inc"   // MidiAsset::MidiAsset ... MidiTable::destroy
#include "_ref/eng/assets_r3.inc"   // SampleTable g_sample_table; MidiTable g_midi_table; }
// dsp/sample.cpp: Sample's constructors, destructor and resize (:45-110; the rest is libsndfile / vorbis / dr_mp3)
namespace wb {
#include "_ref/eng/sample_r1.inc"
}
// engine/track.cpp: from `namespace wb {` to the line in front of Track::process_track_messages
#include "_ref/eng/track_cpp_body.inc"
}  // namespace wb  (the cut ends inside the reference's namespace)
// engine/engine.cpp
using namespace std::chrono_literals;
#include "_ref/eng/engine_r1.inc"   // namespace wb { round_ppq, ~Engine, set_bpm, set_playhead_position, set_audio_channel_config, clear_all
#include "_ref/eng/engine_r2.inc"   // arm_track_recording ... add_track, delete_track, move_track, solo_track
#include "_ref/eng/engine_r3.inc"   // add_clip_from_file, add_audio_clip ... move_clip, resize_clip, delete_clip, add_to_cliplist, delete_region, query_clip_by_range
#include "_ref/eng/engine_r4.inc"   // set_clip_gain
#include "_ref/eng/engine_r5.inc"   // delete_plugin_from_track, get_song_length, update_audio_visualization, process
#include "_ref/eng/engine_r6.inc"   // Engine g_engine; }

// ---- the harness's own definitions (see the header of this file) ------------------------------------------------------------
namespace wb {
void Track::process_track_messages(double) {
  TrackMessage msg;
  while (track_msg_queue.pop(msg)) {
    if (msg.type != TrackMessage::ParamChange) std::abort();   // MIDI notes / plugin parameters: not on the path
    param_queue.push_back_value(0, msg.plugin_param_change.id, msg.plugin_param_change.value);   // track.cpp:777-779
  }
}
PluginResult Track::plugin_begin_edit(void*, PluginInterface*, uint32_t) { std::abort(); }
PluginResult Track::plugin_perform_edit(void*, PluginInterface*, uint32_t, double) { std::abort(); }
PluginResult Track::plugin_end_edit(void*, PluginInterface*, uint32_t) { std::abort(); }
}  // namespace wb

using namespace wb;

static void harness_play(Engine& e) {   // engine.cpp:70-79 (not recording)
  e.editor_lock.lock();
  for (auto track : e.tracks) track->reset_playback_state(e.playhead_start, false);
  e.playhead_updated.store(false, std::memory_order_release);
  e.sample_position = 0;
  e.playing = true;
  e.editor_lock.unlock();
}
static void harness_stop(Engine& e) {   // engine.cpp:85-91 (not recording)
  e.editor_lock.lock();
  e.playing = false;
  e.playhead = e.playhead_start;
  e.playhead_ui = e.playhead_start;
  for (auto track : e.tracks) track->stop();
  e.editor_lock.unlock();
}

namespace {
FILE* g_out;
template <class T>
void put(const T& v) { std::fwrite(&v, sizeof(T), 1, g_out); }
void put_u32(uint32_t v) { put(v); }
void put_f64(double v) { put(v); }

std::vector<SampleAsset*> g_assets;

// synthetic fp32 clip audio: the keyed integer-hash generator of whitebox_amd/synth.py (input generation only — the same bits
// as the device's synth_kernel and the oracle's wbo_synth_f32), so that bench.py's workloads need no gigabyte data file
inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void synth_f32(float* dst, size_t frames, uint64_t key, float amp) {
  for (size_t i = 0; i < frames; i++) {
    const uint64_t u = splitmix64(key ^ (uint64_t)i);
    dst[i] = (float)((int64_t)(u >> 40) - (1 << 23)) * 1.1920928955078125e-07f * amp;
  }
}
double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// the clip list of a track (min, max, start_offset, speed, gain, asset index): what an edit left behind
void dump_clips(Track* t) {
  put_u32((uint32_t)t->clips.size());
  for (auto c : t->clips) {
    put_f64(c->min_time); put_f64(c->max_time); put_f64(c->start_offset); put_f64(c->audio.speed);
    put(c->audio.gain);
    uint32_t ai = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < g_assets.size(); i++) if (g_assets[i] == c->audio.asset) ai = i;
    put_u32(ai);
  }
}
// would this [min, max) land on other clips (i.e. would the reference call reserve_track_region)?
// add_to_cliplist (engine.cpp:409-461) reaches it only past its three early exits and a non-empty range query.
bool add_needs_trim(Track* t, double min_time, double max_time) {
  auto& clips = t->clips;
  if (clips.size() == 0) return false;
  if (clips.back()->max_time < min_time) return false;
  if (clips.front()->min_time > max_time) return false;
  return t->query_clip_by_range(min_time, max_time).has_value();
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* script = std::fopen(argv[1], "r");
  FILE* dataf = std::fopen(argv[2], "rb");
  g_out = std::fopen(argv[3], "wb");
  if (!script || !dataf || !g_out) return 2;
  std::fseek(dataf, 0, SEEK_END);
  const long data_size = std::ftell(dataf);
  std::fseek(dataf, 0, SEEK_SET);
  std::vector<unsigned char> blob((size_t)data_size);
  if (data_size && std::fread(blob.data(), 1, (size_t)data_size, dataf) != (size_t)data_size) return 2;

  // engines[0] = the reference's g_engine.  More engines exist only for the sub-bus composition (`bus` / `runbus` below): SURVEY
  // A13's oracle — "the same composition executed with the reference's AudioBuffer::mix on per-track buffers produced by the
  // reference Track::process" — is formed from whole Engine::process calls, one engine per bus.
  std::vector<Engine*> engines{ &g_engine };
  Engine* cur = &g_engine;
#define E (*cur)
  uint32_t channels = 2, frames = 512;
  double rate = 48000.0, last_bpm = 0.0;
  char line[512], op[32];
  uint32_t block_no = 0;
  while (std::fgets(line, sizeof line, script)) {
    if (std::sscanf(line, "%31s", op) != 1 || op[0] == '#') continue;
    const char* a = line + std::strlen(op);
    uint32_t status = 1;   // 1 = taken by the reference's code, 0 = refused (would need reserve_track_region), 2 = bad argument
    if (!std::strcmp(op, "cfg")) {
      unsigned c, f, r;
      std::sscanf(a, "%u %u %u", &c, &f, &r);
      channels = c; frames = f; rate = (double)r;
      for (auto e : engines) e->set_audio_channel_config(0, c, f, r);
    } else if (!std::strcmp(op, "rate")) {     // rate <r>: the back end reconfigured to another device rate, block shape kept
      unsigned r; std::sscanf(a, "%u", &r);
      rate = (double)r;
      E.set_audio_channel_config(0, channels, frames, r);
    } else if (!std::strcmp(op, "bpm")) {
      double v; std::sscanf(a, "%lf", &v);
      last_bpm = v;
      for (auto e : engines) e->set_bpm(v);
    } else if (!std::strcmp(op, "seek")) {
      double v; std::sscanf(a, "%lf", &v);
      E.set_playhead_position(v);
    } else if (!std::strcmp(op, "play")) {
      for (auto e : engines) harness_play(*e);
    } else if (!std::strcmp(op, "stop")) {
      for (auto e : engines) harness_stop(*e);
    } else if (!std::strcmp(op, "bus")) {      // bus <b>: the operations that follow build the engine that renders sub-bus b
      unsigned b; std::sscanf(a, "%u", &b);
      while (engines.size() <= b) {
        Engine* e = new Engine();
        e->set_audio_channel_config(0, channels, frames, (uint32_t)rate);
        if (last_bpm != 0.0) e->set_bpm(last_bpm);
        engines.push_back(e);
      }
      cur = engines[b];
    } else if (!std::strcmp(op, "sample")) {   // sample <AudioFormat> <channels> <rate> <count> <byte offset into the data file>
      unsigned fmt, ch, r; unsigned long long count, off;
      std::sscanf(a, "%u %u %u %llu %llu", &fmt, &ch, &r, &count, &off);
      Sample s((AudioFormat)fmt, r);
      s.channels = ch;
      s.count = (size_t)count;
      // 16-bit PCM in int16, everything else in 4-byte containers (24-bit files are stored as int32: sample.cpp:20; the sampler
      // reads AudioFormat::I24 through get_sample_data<int32_t>, sampler.cpp:121-132,172-181)
      const size_t es = (AudioFormat)fmt == AudioFormat::I16 ? 2 : 4;
      for (unsigned c = 0; c < ch; c++) {   // load_file's buffers: count + sample_padding elements, zeroed (sample.cpp:127-142)
        std::byte* p = (std::byte*)std::calloc((size_t)count + Sample::sample_padding, es);
        std::memcpy(p, blob.data() + off + (size_t)c * count * es, (size_t)count * es);
        s.sample_data.push_back(p);
      }
      g_assets.push_back(new SampleAsset{ &g_sample_table, (uint64_t)g_assets.size() + 1, 1u, std::move(s), nullptr, true });
    } else if (!std::strcmp(op, "synth")) {    // synth <channels> <rate> <count> <seed> <key track> <amp>: an fp32 sample from the keyed generator
      unsigned ch, r; unsigned long long count, seed, kt; float amp;
      std::sscanf(a, "%u %u %llu %llu %llu %a", &ch, &r, &count, &seed, &kt, &amp);
      Sample s(AudioFormat::F32, r);
      s.channels = ch;
      s.count = (size_t)count;
      for (unsigned c = 0; c < ch; c++) {
        float* p = (float*)std::calloc((size_t)count + Sample::sample_padding, sizeof(float));
        synth_f32(p, (size_t)count, seed ^ (kt << 40) ^ ((unsigned long long)c << 32), amp);
        s.sample_data.push_back((std::byte*)p);
      }
      g_assets.push_back(new SampleAsset{ &g_sample_table, (uint64_t)g_assets.size() + 1, 1u, std::move(s), nullptr, true });
    } else if (!std::strcmp(op, "track")) {
      E.add_track("t");
    } else if (!std::strcmp(op, "vol")) {
      unsigned t; float v; std::sscanf(a, "%u %f", &t, &v);
      E.tracks[t]->set_volume(v);
    } else if (!std::strcmp(op, "pan")) {
      unsigned t; float v; std::sscanf(a, "%u %f", &t, &v);
      E.tracks[t]->set_pan(v);
    } else if (!std::strcmp(op, "mute")) {
      unsigned t, v; std::sscanf(a, "%u %u", &t, &v);
      E.tracks[t]->set_mute(v != 0);
    } else if (!std::strcmp(op, "clip")) {     // clip <track> <min> <max> <start_offset> <sample> <speed> <gain>: Engine::add_audio_clip
      unsigned t, si; double mn, mx, so, sp; float g;
      std::sscanf(a, "%u %la %la %la %u %la %a", &t, &mn, &mx, &so, &si, &sp, &g);
      Track* tr = E.tracks[t];
      if (add_needs_trim(tr, mn, mx)) status = 0;
      else E.add_audio_clip(tr, "c", mn, mx, so, AudioClip{ .asset = g_assets[si], .speed = sp, .gain = g });
    } else if (!std::strcmp(op, "delclip")) {  // delclip <track> <index>: Engine::delete_clip
      unsigned t, i; std::sscanf(a, "%u %u", &t, &i);
      Track* tr = E.tracks[t];
      if (i >= tr->clips.size()) status = 2; else E.delete_clip(tr, tr->clips[i]);
    } else if (!std::strcmp(op, "gain")) {     // gain <track> <index> <gain>: Engine::set_clip_gain
      unsigned t, i; float g; std::sscanf(a, "%u %u %a", &t, &i, &g);
      if (i >= E.tracks[t]->clips.size()) status = 2; else E.set_clip_gain(E.tracks[t], i, g);
    } else if (!std::strcmp(op, "move")) {     // move <track> <index> <relative_pos>: Engine::move_clip
      unsigned t, i; double rel; std::sscanf(a, "%u %u %la", &t, &i, &rel);
      Track* tr = E.tracks[t];
      if (i >= tr->clips.size()) status = 2;
      else {
        Clip* c = tr->clips[i];
        auto [mn, mx] = calc_move_clip(c, rel);
        // move_clip asks the track for [mn, mx) with the clip still in the list: any hit (the clip itself included) goes to
        // reserve_track_region
        if (rel != 0.0 && tr->query_clip_by_range(mn, mx).has_value()) status = 0; else E.move_clip(tr, c, rel);
      }
    } else if (!std::strcmp(op, "query")) {    // query <track> <min> <max>: Track::query_clip_by_range (track.cpp:112-157) — what add / move /
      unsigned t; double mn, mx;               // resize / delete_region hand to reserve_track_region; answered with its own record
      std::sscanf(a, "%u %la %la", &t, &mn, &mx);
      auto q = E.tracks[t]->query_clip_by_range(mn, mx);
      put_u32(0x51525900u); put_u32(q ? 1u : 0u); put_u32(q ? q->first : 0u); put_u32(q ? q->last : 0u);
      put_f64(q ? q->first_offset : 0.0); put_f64(q ? q->last_offset : 0.0);
      continue;
    } else if (!std::strcmp(op, "deltrack")) {
      unsigned s; std::sscanf(a, "%u", &s);
      E.delete_track(s);
    } else if (!std::strcmp(op, "movetrack")) {
      unsigned f, t; std::sscanf(a, "%u %u", &f, &t);
      E.move_track(f, t);
    } else if (!std::strcmp(op, "solo")) {
      unsigned s; std::sscanf(a, "%u", &s);
      E.solo_track(s);
    } else if (!std::strcmp(op, "run")) {      // run <n>: n calls of Engine::process, one record per block
      unsigned n; std::sscanf(a, "%u", &n);
      AudioBuffer<float> in(frames, channels), out(frames, channels);
      put_u32(0x52554E00u); put_u32(n);
      for (unsigned b = 0; b < n; b++, block_no++) {
        E.process(in, out, rate);
        put_u32(0x424C4B00u); put_u32(block_no);
        for (uint32_t c = 0; c < channels; c++) std::fwrite(out.channel_buffers[c], sizeof(float), frames, g_out);
        put_f64(E.playhead); put_f64(E.sample_position);
        put_u32((uint32_t)E.tracks.size());
        for (auto t : E.tracks) {
          put_u32((uint32_t)t->audio_event_buffer.size());
          for (auto& ev : t->audio_event_buffer) {
            put_u32((uint32_t)ev.type); put_u32(ev.buffer_offset); put_f64(ev.time);
            put_f64(ev.type == EventType::PlaySample ? ev.speed : 0.0);
            put((uint64_t)(ev.type == EventType::PlaySample ? ev.sample_offset : 0));
          }
          put_u32((uint32_t)t->current_audio_event.type);
          put_f64(t->sampler.playback_speed_); put_f64(t->sampler.sample_offset_);
          for (int c = 0; c < 2; c++) put(t->level_meter[c].level.load());
        }
      }
      continue;
    } else if (!std::strcmp(op, "runbus")) {   // runbus <n>: n blocks of every engine; master = the buses added in order (AudioBuffer::mix)
      unsigned n; std::sscanf(a, "%u", &n);
      AudioBuffer<float> in(frames, channels), master(frames, channels);
      std::vector<AudioBuffer<float>*> outs;
      for (size_t b = 0; b < engines.size(); b++) outs.push_back(new AudioBuffer<float>(frames, channels));
      put_u32(0x52554200u); put_u32(n); put_u32((uint32_t)engines.size());
      for (unsigned blk = 0; blk < n; blk++) {
        // every Engine::process ends in the hard clamp (engine.cpp:1627-1636): an engine's output is its bus's un-clamped sum only
        // while no sample of it reaches +-1 — bus_peak is reported and the test asserts it.  The master written here is the
        // UN-clamped sum of the buses (the product is asked for the same: wbx_set_clamp(0)); the clamp itself is A10's row.
        float bus_peak = 0.0f;
        master.clear();
        for (size_t b = 0; b < engines.size(); b++) {
          engines[b]->process(in, *outs[b], rate);
          master.mix(*outs[b]);                                          // audio_buffer.h:73-82
        }
        for (uint32_t c = 0; c < channels; c++) std::fwrite(master.channel_buffers[c], sizeof(float), frames, g_out);
        for (size_t b = 0; b < engines.size(); b++)
          for (uint32_t c = 0; c < channels; c++) {
            std::fwrite(outs[b]->channel_buffers[c], sizeof(float), frames, g_out);
            for (uint32_t j = 0; j < frames; j++) bus_peak = std::max(bus_peak, std::fabs(outs[b]->channel_buffers[c][j]));
          }
        put(bus_peak);
        put_f64(engines[0]->playhead); put_f64(engines[0]->sample_position);
      }
      continue;
    } else if (!std::strcmp(op, "bench")) {    // bench <blocks> <budget seconds> <max passes>: play, <blocks> x Engine::process, stop — timed
      unsigned n, max_passes; double budget; std::sscanf(a, "%u %lf %u", &n, &budget, &max_passes);
      AudioBuffer<float> in(frames, channels), out(frames, channels);
      double elapsed = 0.0;
      unsigned passes = 0;
      std::vector<float> head((size_t)channels * frames * std::min(n, 4u));
      while (elapsed < budget && passes < max_passes) {
        harness_play(E);
        for (unsigned b = 0; b < n; b++) {
          const double t0 = now_s();
          E.process(in, out, rate);
          elapsed += now_s() - t0;
          if (passes == 0 && b < 4)
            for (uint32_t c = 0; c < channels; c++)
              std::memcpy(&head[((size_t)b * channels + c) * frames], out.channel_buffers[c], sizeof(float) * frames);
        }
        harness_stop(E);
        passes++;
      }
      put_u32(0x42454E00u); put_u32(n); put_u32(passes); put_f64(elapsed);
      put_u32((uint32_t)head.size());
      std::fwrite(head.data(), sizeof(float), head.size(), g_out);   // the first blocks of the first pass: checked against the oracle
      continue;
    } else if (!std::strcmp(op, "clips")) {    // clips: the clip lists of all tracks
      put_u32(0x434C5000u); put_u32((uint32_t)E.tracks.size());
      for (auto t : E.tracks) dump_clips(t);
      continue;
    } else {
      status = 2;
    }
    put_u32(0x4F500000u); put_u32(status);
  }
  std::fclose(g_out);
  std::_Exit(0);   // no teardown: destroying the tables would walk into the renderer's ~WaveformVisual
}
