// ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver over the REFERENCE'S OWN code, compiled from where it lies under
// /root/reference/src by oracle/Makefile into oracle/_ref/libwbref.so (git-ignored, never committed,
// no reference source is copied into this repo).  It is used to (1) validate the C restatement in
// wb_oracle.c bit-for-bit and (2) generate the golden vectors under tests/golden/.
//
// Linked reference translation units (compile unmodified, no third-party headers needed):
//   dsp/sampler.cpp  core/panning_law.cpp  core/audio_format_conv.cpp
// Header-only reference code used here: core/audio_buffer.h, dsp/dsp_ops.h, core/core_math.h,
//   dsp/sampler.h, dsp/sample.h, engine/clip_edit.h (+ engine/clip.h, engine/assets_table.h it includes).
//
// NOT buildable here (and therefore not in this library): engine/engine.cpp, engine/track.cpp,
// engine/vu_meter.h — they include core/debug.h which needs third-party spdlog (absent from the
// image; writing a stand-in is not allowed).  (Two pieces of such files need nothing of what their file
// lacks and are compiled on their own, cut out of the file where it lies by oracle/Makefile: struct VUMeter
// -> ref_vu_driver.cpp, summarize_for_mipmaps_impl of gfx/waveform_visual.cpp -> ref_mip_driver.cpp.)  dsp/sample.cpp needs libsndfile/dr_mp3/vorbis, so
// wb::Sample's out-of-line constructor/destructor are not linked either: a wb::Sample is
// materialised below by assigning its public fields inside zeroed storage, and is never destroyed.
#include <cstdint>
#include <cstring>
#include <new>

#include "core/algorithm.h"
#include "core/audio_buffer.h"
#include "core/audio_format_conv.h"
#include "core/core_math.h"
#include "core/memory.h"
#include "core/panning_law.h"
#include "core/timing.h"
#include "dsp/dsp_ops.h"
#include "dsp/sampler.h"
#include "engine/audio_io.h"
#include "engine/clip_edit.h"

namespace {

struct SampleBox {
  alignas(wb::Sample) unsigned char storage[sizeof(wb::Sample)];
  wb::Sample* get() { return reinterpret_cast<wb::Sample*>(storage); }
};

// Fill the public fields the sampler reads (sample.h:21-28).  No constructor / destructor runs.
wb::Sample* make_sample(SampleBox& box, int format, uint32_t channels, uint32_t sample_rate, size_t count,
                        const void* const* planar) {
  std::memset(box.storage, 0, sizeof(box.storage));
  wb::Sample* s = box.get();
  s->format = (wb::AudioFormat)format;
  s->channels = channels;
  s->sample_rate = sample_rate;
  s->count = count;
  s->capacity = count;
  s->sample_data.data_ = (std::byte**)planar;
  s->sample_data.size_ = channels;
  s->sample_data.capacity_ = channels;
  return s;
}

// A wb::Clip / wb::SampleAsset pair materialised by field assignment in zeroed storage (their out-of-line
// members live in assets_table.cpp / sample.cpp, which need third-party libraries): only the fields
// clip_edit.h reads are set, and neither object is ever destroyed.
struct ClipBox {
  alignas(wb::Clip) unsigned char clip[sizeof(wb::Clip)];
  alignas(wb::SampleAsset) unsigned char asset[sizeof(wb::SampleAsset)];
  wb::Clip* make(double min_time, double max_time, double start_offset, double speed, double sample_rate,
                 double sample_count) {
    std::memset(clip, 0, sizeof(clip));
    std::memset(asset, 0, sizeof(asset));
    wb::SampleAsset* a = reinterpret_cast<wb::SampleAsset*>(asset);
    a->sample_instance.sample_rate = (uint32_t)sample_rate;
    a->sample_instance.count = (size_t)sample_count;
    wb::Clip* c = reinterpret_cast<wb::Clip*>(clip);
    c->type = wb::ClipType::Audio;
    c->min_time = min_time;
    c->max_time = max_time;
    c->start_offset = start_offset;
    c->audio.asset = a;
    c->audio.speed = speed;
    c->audio.gain = 1.0f;
    return c;
  }
};

}  // namespace

extern "C" {

// engine/clip_edit.h:10-16
void ref_calc_move_clip(double clip_min, double clip_max, double relative_pos, double min_move, double* new_min,
                        double* new_max) {
  ClipBox box;
  wb::Clip* c = box.make(clip_min, clip_max, 0.0, 1.0, 48000.0, 0.0);
  wb::ClipMoveResult r = wb::calc_move_clip(c, relative_pos, min_move);
  *new_min = r.min;
  *new_max = r.max;
}

// engine/clip_edit.h:18-126
void ref_calc_resize_clip(double clip_min, double clip_max, double clip_start_offset, double clip_speed,
                          double sample_rate, double sample_count, double relative_pos, double resize_limit,
                          double min_length, double min_resize_pos, double beat_duration, int is_min, int shift,
                          int stretch, int clamp_at_resize_pos, double* out_min, double* out_max,
                          double* out_start_offset, double* out_speed) {
  ClipBox box;
  wb::Clip* c = box.make(clip_min, clip_max, clip_start_offset, clip_speed, sample_rate, sample_count);
  wb::ClipResizeResult r = wb::calc_resize_clip(c, relative_pos, resize_limit, min_length, min_resize_pos, beat_duration,
                                                is_min != 0, shift != 0, stretch != 0, clamp_at_resize_pos != 0);
  *out_min = r.min;
  *out_max = r.max;
  *out_start_offset = r.start_offset;
  *out_speed = r.speed;
}

// engine/clip_edit.h:128-137
double ref_calc_clip_shift(double start_offset, double relative_pos, double beat_duration, double sample_rate) {
  return wb::calc_clip_shift(true, start_offset, relative_pos, beat_duration, sample_rate);
}

// engine/clip_edit.h:139-150
double ref_shift_clip_content(double start_offset, double speed, double sample_rate, double relative_pos,
                              double beat_duration) {
  ClipBox box;
  wb::Clip* c = box.make(0.0, 1.0, start_offset, speed, sample_rate, 1e9);
  return wb::shift_clip_content(c, relative_pos, beat_duration);
}


// Engine::perf_measurer — what Engine::process ends with (engine.cpp:1577,1653): PerformanceMeasurer::update / get_usage,
// core/timing.h:54-67 (header-only), on a measurer that starts from `usage`
double ref_perf_update(double usage, double duration_ms, double target_ms) {
  wb::PerformanceMeasurer m;
  m.usage.store(usage);
  m.update(duration_ms, target_ms);
  return m.usage.load();
}
double ref_perf_get_usage(double usage) {
  wb::PerformanceMeasurer m;
  m.usage.store(usage);
  return m.get_usage();
}
// Engine::audio_buffer_duration_ms as set_audio_channel_config computes it (engine.cpp:52): period_to_ms(buffer_size_to_period(..)),
// engine/audio_io.h:187-195 (header-only)
double ref_buffer_duration_ms(uint32_t buffer_size, uint32_t sample_rate) {
  return wb::period_to_ms(wb::buffer_size_to_period(buffer_size, sample_rate));
}

// math::db_to_linear<float>, core/core_math.h:83-89
float ref_db_to_linear(float db) { return wb::math::db_to_linear<float>(db); }

// calculate_panning_coefs, core/panning_law.cpp:9-32
void ref_pan_coefs(float p, int law, float* l, float* r) {
  wb::PanningCoefficient c = wb::calculate_panning_coefs(p, (wb::PanningLaw)law);
  *l = c.left;
  *r = c.right;
}

double ref_beat_to_samples(double beat, double sr, double bd) { return wb::beat_to_samples(beat, sr, bd); }
double ref_samples_to_beat(double smp, double sr, double bd) { return wb::samples_to_beat(smp, sr, bd); }

// dsp::apply_gain / find_abs_maximum, dsp/dsp_ops.h:10-31
void ref_apply_gain(float* buf, uint32_t n, float g) { wb::dsp::apply_gain<float>(buf, n, g); }
float ref_find_abs_maximum(const float* buf, uint32_t n) { return wb::dsp::find_abs_maximum<float>(buf, n); }

// Sampler::reset_state + Sampler::stream on a caller-described sample; state is passed in/out.
// dsp/sampler.h:18-27, dsp/sampler.cpp:88-210
void ref_sampler_reset(double* playback_speed, double* sample_offset, double off, double speed, double src_rate,
                       double dst_rate) {
  wb::dsp::Sampler s{};
  s.reset_state(wb::dsp::ResamplerType::Linear, off, speed, src_rate, dst_rate);
  *playback_speed = s.playback_speed_;
  *sample_offset = s.sample_offset_;
}

void ref_sampler_stream(double* playback_speed, double* sample_offset, int format, uint32_t channels,
                        uint32_t sample_rate, size_t count, const void* const* planar, uint32_t num_channels,
                        uint32_t num_samples, uint32_t buffer_offset, float gain, float** dst) {
  SampleBox box;
  wb::Sample* smp = make_sample(box, format, channels, sample_rate, count, planar);
  wb::dsp::Sampler s{};
  s.playback_speed_ = *playback_speed;
  s.sample_offset_ = *sample_offset;
  s.resampler_type_ = wb::dsp::ResamplerType::Linear;
  s.stream(smp, num_channels, num_samples, buffer_offset, gain, dst);
  *playback_speed = s.playback_speed_;
  *sample_offset = s.sample_offset_;
}

// One block of the reference's per-sample arithmetic with the SEQUENCING SUPPLIED AS DATA:
// for each track (in order): mixing_buffer.clear() (audio_buffer.h:67-71); Sampler::stream per
// segment (sampler.cpp:88-210); apply_gain(volume*pan) (dsp_ops.h:27-31, track.cpp:728-731);
// abs-max (dsp_ops.h:10-19 == vu_meter.h:20-25); then AudioBuffer::mix into the bus / output
// (audio_buffer.h:73-82); finally the master clamp loop of engine.cpp:1627-1636 (restated, 6 lines,
// engine.cpp itself is not buildable here).
struct ref_segment {
  double playback_speed;   // Sampler::playback_speed_
  double sample_offset;    // Sampler::sample_offset_ at the start of the segment
  uint32_t track;          // segments must be grouped by track, ascending
  uint32_t dst_start;      // buffer_offset
  uint32_t len;            // num_samples handed to Sampler::stream
  float gain;              // clip gain
  int sample;              // index into the sample arrays
};

void ref_mix_block(uint32_t n_tracks, uint32_t n_channels, uint32_t n_frames, const ref_segment* segs, uint32_t n_segs,
                   const int* smp_format, const uint32_t* smp_channels, const uint32_t* smp_rate,
                   const size_t* smp_count, const void* const* const* smp_planar, const float* track_gains /*[T][C]*/,
                   const int* track_bus /*[T] or null*/, uint32_t n_buses, float* const* out /*[C][F]*/,
                   float* bus_out /*[n_buses][C][F] or null*/, float* peaks /*[T][C] or null*/,
                   double* seg_end_offset /*[n_segs] or null*/, int clamp) {
  wb::AudioBuffer<float> output(n_frames, n_channels);
  wb::AudioBuffer<float> mixing(n_frames, n_channels);
  wb::AudioBuffer<float>* buses = nullptr;
  if (n_buses) {
    buses = (wb::AudioBuffer<float>*)::operator new(sizeof(wb::AudioBuffer<float>) * n_buses);
    for (uint32_t u = 0; u < n_buses; u++) new (&buses[u]) wb::AudioBuffer<float>(n_frames, n_channels);
  }
  output.clear();
  uint32_t si = 0;
  for (uint32_t t = 0; t < n_tracks; t++) {
    mixing.clear();
    while (si < n_segs && segs[si].track == t) {
      const ref_segment& sg = segs[si];
      SampleBox box;
      wb::Sample* smp = make_sample(box, smp_format[sg.sample], smp_channels[sg.sample], smp_rate[sg.sample],
                                    smp_count[sg.sample], smp_planar[sg.sample]);
      wb::dsp::Sampler s{};
      s.playback_speed_ = sg.playback_speed;
      s.sample_offset_ = sg.sample_offset;
      s.resampler_type_ = wb::dsp::ResamplerType::Linear;
      s.stream(smp, n_channels, sg.len, sg.dst_start, sg.gain, mixing.channel_buffers);
      if (seg_end_offset) seg_end_offset[si] = s.sample_offset_;
      si++;
    }
    for (uint32_t c = 0; c < n_channels; c++) {
      wb::dsp::apply_gain<float>(mixing.channel_buffers[c], n_frames, track_gains[t * n_channels + c]);
      if (peaks) peaks[t * n_channels + c] = wb::dsp::find_abs_maximum<float>(mixing.channel_buffers[c], n_frames);
    }
    if (buses && track_bus && track_bus[t] >= 0)
      buses[track_bus[t]].mix(mixing);
    else
      output.mix(mixing);
  }
  if (buses) {
    for (uint32_t u = 0; u < n_buses; u++) {
      output.mix(buses[u]);
      if (bus_out)
        for (uint32_t c = 0; c < n_channels; c++)
          std::memcpy(bus_out + ((size_t)u * n_channels + c) * n_frames, buses[u].channel_buffers[c],
                      n_frames * sizeof(float));
      buses[u].~AudioBuffer<float>();
    }
    ::operator delete(buses);
  }
  if (clamp) {  // engine.cpp:1627-1636
    for (uint32_t i = 0; i < output.n_channels; i++) {
      float* channel = output.get_write_pointer(i);
      for (uint32_t j = 0; j < output.n_samples; j++) {
        if (channel[j] > 1.0) {
          channel[j] = 1.0;
        } else if (channel[j] < -1.0) {
          channel[j] = -1.0;
        }
      }
    }
  }
  for (uint32_t c = 0; c < n_channels; c++) std::memcpy(out[c], output.channel_buffers[c], n_frames * sizeof(float));
}

// core/algorithm.h:24-40 with the comparator of its call sites in the clip sequencer (track.cpp:126-127, :206):
// index of the first clip whose max_time is not <= value, clamped to the last clip (right starts at n-1)
uint32_t ref_find_lower_bound_max_time(const double* max_times, uint32_t n, double value) {
  ClipBox* boxes = new ClipBox[n];
  wb::Clip** clips = new wb::Clip*[n];
  for (uint32_t i = 0; i < n; i++) clips[i] = boxes[i].make(0.0, max_times[i], 0.0, 1.0, 48000.0, 0.0);
  wb::Clip** begin = clips;
  wb::Clip** end = clips + n;
  auto it = wb::find_lower_bound(begin, end, value, [](wb::Clip* clip, double time_pos) { return clip->max_time <= time_pos; });
  const uint32_t idx = (uint32_t)(it - begin);
  delete[] clips;
  delete[] boxes;
  return idx;
}

// core/audio_format_conv.cpp:5-106
void ref_f32_to_i16(int16_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  wb::convert_f32_to_interleaved_i16(dst, src, off, n, nch);
}
void ref_f32_to_i24(unsigned char* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  wb::convert_f32_to_interleaved_i24((std::byte*)dst, src, off, n, nch);
}
void ref_f32_to_i24_x8(int32_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  wb::convert_f32_to_interleaved_i24_x8(dst, src, off, n, nch);
}
void ref_f32_to_i32(int32_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  wb::convert_f32_to_interleaved_i32(dst, src, off, n, nch);
}
void ref_f32_to_f32(float* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  wb::convert_to_interleaved_f32(dst, src, off, n, nch);
}


// Pool<Clip> (core/memory.h:41-110), the allocator behind Track::allocate_clip / destroy_clip (track.h:156-168): which chunk
// an allocation gets (the one freed last, else a fresh one) and what a freed chunk holds.  ops[i] > 0: allocate (the i-th
// live object gets handle = number of allocations so far, 1-based; the chunk is filled with 0xFF like a live object);
// ops[i] < 0: free the object with handle -ops[i].  out[i]: for an allocation the chunk's identity (1-based order in which
// chunks were first handed out); for a free 1 when every byte of the chunk behind the free-list link reads zero afterwards
// (the `audio.gain` a dangling Clip* then reads), else 0.
int ref_pool_script(const int* ops, int n, int* out) {
  wb::Pool<wb::Clip> pool;                   // one per script, like a Track's clip_allocator (track.h:117)
  void* seen[4096];
  int n_seen = 0;
  void* by_handle[4096] = {};
  int n_alloc = 0;
  for (int i = 0; i < n; i++) {
    if (ops[i] > 0) {
      void* p = pool.allocate();
      if (!p || n_alloc >= 4095) return -1;
      std::memset(p, 0xFF, sizeof(wb::Clip));
      by_handle[++n_alloc] = p;
      int id = 0;
      for (int k = 0; k < n_seen; k++)
        if (seen[k] == p) id = k + 1;
      if (!id) {
        if (n_seen >= 4096) return -1;
        seen[n_seen++] = p;
        id = n_seen;
      }
      out[i] = id;
    } else {
      void* p = by_handle[-ops[i]];
      if (!p) return -2;
      by_handle[-ops[i]] = nullptr;
      pool.free(p);
      const unsigned char* b = (const unsigned char*)p;
      int zero = 1;
      for (size_t k = sizeof(wb::PoolChunk); k < sizeof(wb::Clip); k++) zero &= b[k] == 0;
      out[i] = zero;
    }
  }
  return 0;
}

}  // extern "C"
