// wbx_adapter.hpp — header-only C++ adapter over the C ABI (wbx.h) with the reference's own shapes.
//
// A reference maintainer who wants the MI355X path behind the existing C++ host replaces `wb::Engine
// g_engine` (src/engine/engine.cpp:1715, engine.h:273) by `wbx::Engine`: same method names, argument
// meaning and buffer type as
//   wb::AudioBuffer<float>                         src/core/audio_buffer.h:14-175
//   wb::Engine::set_audio_channel_config/set_bpm/add_track/add_audio_clip/play/stop/process
//                                                   src/engine/engine.h:68-113,235-239
//   wb::Track::set_volume/set_pan/set_mute          src/engine/track.h:137-139
// Compiles with any C++17 compiler (no HIP headers needed); link against libwbx.so.
#pragma once
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "wbx.h"

namespace wbx {

enum class AudioFormat : int {   // values of the reference's enum, src/core/audio_format.h:7-20 (= WBX_FMT_* / WBX_OUT_*)
  I16 = WBX_OUT_I16, I24 = WBX_OUT_I24, I24_X8 = WBX_OUT_I24_X8, I32 = WBX_OUT_I32, F32 = WBX_OUT_F32
};

// Planar sample buffer with the reference's fields, layout and semantics (audio_buffer.h:14-175): the first 16 channel
// pointers live inside the object (`internal_channel_buffers`), `channel_buffers` points at them or, beyond 16
// channels, at a heap array; every channel is a separate 32-byte aligned, zero-filled allocation.
template <typename T>
struct AudioBuffer {
  static_assert(sizeof(T) == 4, "the mix path is fp32");
  static constexpr uint32_t internal_buffer_capacity = 16;
  static constexpr uint32_t alignment = 32;

  uint32_t n_samples{};
  uint32_t n_channels{};
  uint32_t channel_capacity = internal_buffer_capacity;
  T* internal_channel_buffers[internal_buffer_capacity]{};
  T** channel_buffers{};

  AudioBuffer() : channel_buffers(internal_channel_buffers) {}
  AudioBuffer(uint32_t sample_count, uint32_t channel_count) : n_samples(sample_count), channel_buffers(internal_channel_buffers) {
    resize_channel_array_(channel_count);
    for (uint32_t i = 0; i < n_channels; i++) channel_buffers[i] = alloc_channel(sample_count);
  }
  AudioBuffer(const AudioBuffer&) = delete;
  AudioBuffer& operator=(const AudioBuffer&) = delete;
  ~AudioBuffer() {
    for (uint32_t i = 0; i < n_channels; i++) std::free(channel_buffers[i]);
    if (channel_buffers != internal_channel_buffers) std::free(channel_buffers);
  }
  T* get_write_pointer(uint32_t channel, uint32_t sample_offset = 0) {
    assert(channel < n_channels && "Channel out of range");
    return channel_buffers[channel] + sample_offset;
  }
  const T* get_read_pointer(uint32_t channel, uint32_t sample_offset = 0) const {
    assert(channel < n_channels && "Channel out of range");
    return channel_buffers[channel] + sample_offset;
  }
  void set_sample(uint32_t channel, uint32_t sample_offset, T sample) const { channel_buffers[channel][sample_offset] = sample; }
  void mix_sample(uint32_t channel, uint32_t sample_offset, T sample) const { channel_buffers[channel][sample_offset] += sample; }
  void clear() {   // audio_buffer.h:67-71
    for (uint32_t i = 0; i < n_channels; i++) std::memset(channel_buffers[i], 0, n_samples * sizeof(T));
  }
  void mix(const AudioBuffer<T>& other) {   // audio_buffer.h:73-82: iterates THIS buffer's channels
    assert(n_samples == other.n_samples);
    for (uint32_t i = 0; i < n_channels; i++)
      for (uint32_t j = 0; j < n_samples; j++) channel_buffers[i][j] += other.channel_buffers[i][j];
  }
  // resize(samples, clear): keeps the old contents unless `clear`, zero-fills the growth (audio_buffer.h:84-110)
  void resize(uint32_t samples, bool clear = false) {
    if (samples == n_samples) return;
    for (uint32_t i = 0; i < n_channels; i++) {
      T* fresh = alloc_channel(samples);
      if (!clear) std::memcpy(fresh, channel_buffers[i], (samples < n_samples ? samples : n_samples) * sizeof(T));
      std::free(channel_buffers[i]);
      channel_buffers[i] = fresh;
    }
    n_samples = samples;
  }
  // resize_channel(count): new channels are zeroed, dropped ones freed (audio_buffer.h:112-132)
  void resize_channel(uint32_t channel_count) {
    assert(n_samples != 0);
    if (channel_count == n_channels) return;
    const uint32_t old = n_channels;
    for (uint32_t i = channel_count; i < old; i++) {
      std::free(channel_buffers[i]);
      channel_buffers[i] = nullptr;
    }
    resize_channel_array_(channel_count);
    for (uint32_t i = old; i < channel_count; i++) channel_buffers[i] = alloc_channel(n_samples);
  }
  // planar -> interleaved (audio_buffer.h:143-160).  The fp32 case is a copy and stays on the host
  // (convert_to_interleaved_f32, audio_format_conv.cpp:79-91); the integer device formats are produced on the GPU from
  // the master it already holds: wbx_fetch_interleaved(ctx, WBX_OUT_*, dst) — nothing of the path computes on the CPU.
  void interleave_samples_to(void* dst, uint32_t offset, uint32_t count, AudioFormat format = AudioFormat::F32) const {
    assert(format == AudioFormat::F32 && "integer formats: wbx_fetch_interleaved");
    (void)format;
    float* out = static_cast<float*>(dst);
    for (uint32_t c = 0; c < n_channels; c++)
      for (uint32_t i = 0; i < count; i++) out[(size_t)i * n_channels + c] = channel_buffers[c][offset + i];
  }
  // interleaved f32 -> planar (convert_to_deinterleaved_f32, audio_format_conv.cpp:93-105; the reference has no other case)
  void deinterleave_samples_from(const void* src, uint32_t dst_offset, uint32_t count, AudioFormat format = AudioFormat::F32) {
    assert(format == AudioFormat::F32);
    (void)format;
    const float* in = static_cast<const float*>(src);
    for (uint32_t c = 0; c < n_channels; c++)
      for (uint32_t i = 0; i < count; i++) channel_buffers[c][dst_offset + i] = in[(size_t)i * n_channels + c];
  }
  // audio_buffer.h:162-174 (the pointer array moves to the heap beyond `channel_capacity` channels)
  void resize_channel_array_(uint32_t channel_count) {
    if (channel_count > channel_capacity) {
      T** grown = static_cast<T**>(std::calloc(channel_count, sizeof(T*)));
      assert(grown && "Cannot allocate memory for audio channel array");
      std::memcpy(grown, channel_buffers, (n_channels < channel_count ? n_channels : channel_count) * sizeof(T*));
      if (channel_buffers != internal_channel_buffers) std::free(channel_buffers);
      channel_buffers = grown;
      channel_capacity = channel_count;
    }
    n_channels = channel_count;
  }

 private:
  static T* alloc_channel(uint32_t samples) {
    T* p = static_cast<T*>(std::aligned_alloc(alignment, ((samples * sizeof(T) + alignment - 1) / alignment) * alignment + alignment));
    assert(p && "Cannot allocate memory for audio buffer");
    std::memset(p, 0, samples * sizeof(T));
    return p;
  }
};

struct Error : std::runtime_error {
  wbx_status status;
  Error(wbx_status s, const std::string& what) : std::runtime_error(what), status(s) {}
};

struct Engine;

// The audio-side half of engine/vu_meter.h (push_samples, :20-30) runs on the GPU as running per-track maxima;
// Engine::fetch_levels() moves them in here with the reference's CAS-max.  update() / get_value() keep the UI code that
// is written against Track::level_meter compiling unchanged (ui/controls.cpp: level_meter[c].update(...), .get_value()).
struct VUMeter {
  std::atomic<float> level{0.0f};
  float current_level = 0.0f;   // vu_meter.h:18
  void push_level(float peak) {   // the CAS-max tail of VUMeter::push_samples, vu_meter.h:26-29
    float seen = level.load(std::memory_order_relaxed);
    while (seen < peak && !level.compare_exchange_weak(seen, peak, std::memory_order_release, std::memory_order_relaxed)) {
    }
  }
  // the maximum since the last call; resets it (what the host's own VUMeter::update starts from)
  float take_level() { return level.exchange(0.0f, std::memory_order_acq_rel); }
  // UI rate, vu_meter.h:32-44: the meter jumps up to a new maximum at once and falls back towards the newest one through a
  // one-pole release whose time constant is 1 / (frame_rate * speed) frames
  void update(float frame_rate, float speed) {
    const float peak = take_level();
    if (peak > current_level) {
      current_level = peak;
      return;
    }
    const float release = 1.0f - std::exp(-1.0f / (frame_rate * speed));
    current_level += (peak - current_level) * release;
  }
  float get_value() const { return current_level; }
};

struct Track {   // track.h:110-139
  Engine* engine{};
  uint32_t index{};
  std::string name;
  VUMeter level_meter[2]{};             // track.h:121
  const wbx_plugin* plugin_instance{};  // track.h:124: the effect slot, always empty in this path
  Track(Engine* e, uint32_t i, std::string n) : engine(e), index(i), name(std::move(n)) {}
  void set_volume(float db);
  void set_pan(float pan);
  void set_mute(bool mute);
};

struct AudioClip {   // clip.h:39-45: the asset is a sample id returned by Engine::add_sample
  uint32_t asset{};
  double speed = 1.0;
  float gain = 1.0f;
};

struct Engine {
  wbx_engine* h{};
  uint32_t num_output_channels = 0, audio_buffer_size = 0, audio_sample_rate = 0;
  std::vector<std::unique_ptr<Track>> tracks;
  // device-side limits, fixed when the engine is first configured (not part of the reference's surface)
  uint32_t max_tracks = 4096, max_blocks = 1;
  int device = 0;
  // The audio path has no exceptions in the reference (Engine::process returns void, failures are asserts): process()
  // never throws — a failed block leaves silence in the output buffer and latches its status here.
  std::atomic<wbx_status> process_status{WBX_OK};
  std::string process_error;   // written by the audio thread only
  // engine.h:33 / :64 — the block's period in ms (engine.cpp:52) and the load figure process() keeps (engine.cpp:1653):
  // `g_engine.perf_measurer.get_usage()` of ui/control_bar.cpp:54 reads the same here
  double audio_buffer_duration_ms = 0;
  struct PerformanceMeasurer {
    Engine* owner;
    double get_usage() const {
      double u = 0.0;
      return (owner->h && wbx_engine_perf_usage(owner->h, &u, nullptr) == WBX_OK) ? u : 0.0;
    }
  } perf_measurer{this};

  // set_audio_channel_config(in, out, buffer_size, sample_rate), engine.cpp:43-57.  The first call creates the device
  // context; later calls (the audio backend was reconfigured) resize it in place — tracks and clips stay, as in the
  // reference.
  void set_audio_channel_config(uint32_t /*input_channels*/, uint32_t output_channels, uint32_t buffer_size, uint32_t sample_rate) {
    if (!h) {
      wbx_config cfg{};
      cfg.device = device;
      cfg.max_tracks = max_tracks;
      cfg.max_blocks = max_blocks;
      cfg.block_frames = buffer_size;
      cfg.channels = output_channels;
      cfg.sample_rate = sample_rate;
      check(wbx_engine_create(&cfg, &h), "wbx_engine_create");
    } else {
      check(wbx_engine_set_audio_channel_config(h, output_channels, buffer_size, sample_rate), "set_audio_channel_config");
    }
    num_output_channels = output_channels;
    audio_buffer_size = buffer_size;
    audio_sample_rate = sample_rate;
    audio_buffer_duration_ms = wbx_calc_buffer_period_ms(buffer_size, sample_rate);
  }
  ~Engine() {
    if (h) wbx_engine_destroy(h);
  }
  void set_bpm(double bpm) { check(wbx_engine_set_bpm(h, bpm), "set_bpm"); }
  void set_playhead_position(double beat) { check(wbx_engine_set_playhead_position(h, beat), "set_playhead_position"); }
  Track* add_track(const std::string& name) {
    uint32_t idx = 0;
    check(wbx_engine_add_track(h, &idx), "add_track");
    tracks.emplace_back(new Track(this, idx, name));
    return tracks.back().get();
  }
  // engine.cpp:210-262
  void delete_track(uint32_t slot) {
    check(wbx_engine_delete_track(h, slot), "delete_track");
    tracks.erase(tracks.begin() + slot);
    for (uint32_t i = 0; i < tracks.size(); i++) tracks[i]->index = i;
  }
  void move_track(uint32_t from_slot, uint32_t to_slot) {
    check(wbx_engine_move_track(h, from_slot, to_slot), "move_track");
    auto t = std::move(tracks[from_slot]);
    tracks.erase(tracks.begin() + from_slot);
    tracks.insert(tracks.begin() + to_slot, std::move(t));
    for (uint32_t i = 0; i < tracks.size(); i++) tracks[i]->index = i;
  }
  void solo_track(uint32_t slot) { check(wbx_engine_solo_track(h, slot), "solo_track"); }
  void clear_all() {   // engine.cpp:59-66
    check(wbx_engine_clear_all(h), "clear_all");
    tracks.clear();
  }
  // decoded clip audio -> HBM (what SampleAsset / Sample hold in the reference: assets_table.h:22-35, sample.h:18-28)
  uint32_t add_sample(int format, uint32_t channels, uint32_t sample_rate, uint64_t frames, const void* const* planar) {
    uint32_t id = 0;
    check(wbx_engine_add_sample(h, format, channels, sample_rate, frames, planar, &id), "add_sample");
    return id;
  }
  // the same from a decoder's interleaved frames (what Sample::load_file reads: sample.cpp:154-183); the
  // deinterleave of sample.cpp:29-43 runs on the GPU
  uint32_t add_sample_interleaved(int format, uint32_t channels, uint32_t sample_rate, uint64_t frames, const void* interleaved) {
    uint32_t id = 0;
    check(wbx_engine_add_sample_interleaved(h, format, channels, sample_rate, frames, interleaved, &id), "add_sample_interleaved");
    return id;
  }
  // Engine::add_audio_clip(track, name, min_time, max_time, start_offset, clip_info), engine.h:106-113
  void add_audio_clip(Track* track, const std::string& /*name*/, double min_time, double max_time, double start_offset,
                      const AudioClip& clip_info) {
    check(wbx_engine_add_audio_clip(h, track->index, min_time, max_time, start_offset, clip_info.asset, clip_info.speed,
                                    clip_info.gain),
          "add_audio_clip");
  }
  void play() { check(wbx_engine_play(h), "play"); }
  void stop() { check(wbx_engine_stop(h), "stop"); }
  // void Engine::process(const AudioBuffer<float>&, AudioBuffer<float>&, double), engine.h:235-239
  // Audio thread.  Never throws: on a failure the block is silence and process_status / process_error say why.
  void process(const AudioBuffer<float>& /*input_buffer*/, AudioBuffer<float>& output_buffer, double sample_rate) noexcept {
    assert(output_buffer.n_samples == audio_buffer_size && output_buffer.n_channels == num_output_channels);
    assert(sample_rate == (double)audio_sample_rate);
    (void)sample_rate;
    const wbx_status st = wbx_engine_process(h, output_buffer.channel_buffers);
    if (st != WBX_OK) {
      output_buffer.clear();
      process_error = wbx_engine_last_error(h);
      process_status.store(st, std::memory_order_release);
    }
  }
  // process() and the back end's output_buffer.interleave_samples_to(dst, 0, audio_buffer_size, format)
  // (audio_io_pulseaudio.cpp:419-461, audio_buffer.h:143-160) in one call: the device-format conversion of
  // core/audio_format_conv.cpp runs on the GPU as the last step of the block.  Never throws (see process()).
  void process_interleaved(void* dst, AudioFormat format) noexcept {
    const wbx_status st = wbx_engine_process_interleaved(h, static_cast<int>(format), dst);
    if (st != WBX_OK) {
      const size_t eb = format == AudioFormat::I16 ? 2 : format == AudioFormat::I24 ? 3 : 4;
      std::memset(dst, 0, (size_t)audio_buffer_size * num_output_channels * eb);
      process_error = wbx_engine_last_error(h);
      process_status.store(st, std::memory_order_release);
    }
  }
  // UI thread, once per frame before the host's meters decay (Track::level_meter[c].take_level()): the running per-track maxima the GPU kept since
  // the last call (VUMeter::push_samples, vu_meter.h:20-30) go into the tracks' meters
  void fetch_levels() {
    if (tracks.empty()) return;
    std::vector<float> lv(tracks.size() * num_output_channels);
    check(wbx_engine_levels(h, lv.data(), (uint32_t)tracks.size()), "fetch_levels");
    for (size_t t = 0; t < tracks.size(); t++)
      for (uint32_t c = 0; c < num_output_channels && c < 2; c++) tracks[t]->level_meter[c].push_level(lv[t * num_output_channels + c]);
  }
  // Engine::add_plugin_to_track / delete_plugin_from_track (engine.h:227-229): the slot exists, processing through it
  // is PluginResult::Unimplemented — returns nullptr like the reference does when a plugin cannot be opened
  const wbx_plugin* add_plugin_to_track(Track* track, const wbx_plugin* plugin) {
    if (wbx_engine_add_plugin_to_track(h, track->index, plugin) != WBX_OK) return nullptr;
    track->plugin_instance = nullptr;
    return nullptr;
  }
  void delete_plugin_from_track(Track* track) {
    check(wbx_engine_delete_plugin_from_track(h, track->index), "delete_plugin_from_track");
    track->plugin_instance = nullptr;
  }
  void check(wbx_status s, const char* where) const {
    if (s != WBX_OK) throw Error(s, std::string(where) + ": " + (h ? wbx_engine_last_error(h) : wbx_status_string(s)));
  }
};

inline void Track::set_volume(float db) { engine->check(wbx_track_set_volume(engine->h, index, db), "Track::set_volume"); }
inline void Track::set_pan(float pan) { engine->check(wbx_track_set_pan(engine->h, index, pan), "Track::set_pan"); }
inline void Track::set_mute(bool mute) { engine->check(wbx_track_set_mute(engine->h, index, mute ? 1 : 0), "Track::set_mute"); }

}  // namespace wbx
