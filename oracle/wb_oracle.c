/*
 * wb_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY).  See wb_oracle.h for scope and pinning status.
 * Plain-C restatement of the reference algorithm; each function cites the reference file:line
 * (relative to /root/reference/src).  Never linked into the product.
 */
#include "wb_oracle.h"

#include <limits.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * scalar helpers
 * ---------------------------------------------------------------------------------------------- */

/* core/core_math.h:83-89 — db_to_linear<float>: x <= -72 -> 0, else powf(10, (float)((double)x*0.05)) */
float wbo_db_to_linear(float db) {
  if (db <= -72.0f)
    return 0.0f;
  return powf(10.0f, (float)((double)db * 0.05));
}

/* core/panning_law.cpp:9-32 — only Linear and ConstantPower_3db compute anything (Q7). */
void wbo_pan_coefs(float p, int law, float* left_out, float* right_out) {
  double boost = 0.0, left = 0.0, right = 0.0;
  double x = 0.5 * ((double)p + 1.0);
  const double pi = 3.141592653589793238462643383279502884; /* std::numbers::pi */
  switch (law) {
    case WBO_PAN_LINEAR:
      left = (1.0 - x) * 0.5;
      right = x * 0.5;
      boost = 2.0;
      break;
    case WBO_PAN_CP_3DB:
      left = sin(0.5 * pi * (1.0 - x));
      right = sin(0.5 * pi * x);
      boost = sqrt(2.0);
      break;
    default: break;
  }
  *left_out = (float)(left * boost);
  *right_out = (float)(right * boost);
}

/* core/core_math.h:209-212 */
double wbo_beat_to_samples(double beat, double sample_rate, double beat_duration) {
  double sec = beat * beat_duration;
  return sec * sample_rate;
}

/* core/core_math.h:204-207 */
double wbo_samples_to_beat(double samples, double sample_rate, double beat_duration) {
  double sec = samples / sample_rate;
  return sec / beat_duration;
}

/* The load figure Engine::process ends with (engine.cpp:1577 ScopedPerformanceCounter, :1653 perf_measurer.update(duration in
 * ms, audio_buffer_duration_ms)): PerformanceMeasurer::update, core/timing.h:57-62 — a quarter of the way from the old figure to
 * this block's duration / period — and get_usage, :64-66 (math::clamp: core_math.h, max(min(x, hi), lo) by compares). */
double wbo_perf_update(double usage, double duration_ms, double target_ms) {
  double percentage = duration_ms / target_ms;
  return usage + 0.25 * (percentage - usage);
}
double wbo_perf_get_usage(double usage) {
  double hi = usage < 1.0 ? usage : 1.0;
  return hi > 0.0 ? hi : 0.0;
}
/* Engine::audio_buffer_duration_ms (engine.cpp:52): period_to_ms(buffer_size_to_period(buffer_size, sample_rate)),
 * engine/audio_io.h:187-195 — the period in 100-ns units, rounded as math::round does (core_math.h:61-63: truncation of
 * x +- 0.5 through int64), and back to milliseconds */
double wbo_buffer_duration_ms(uint32_t buffer_size, uint32_t sample_rate) {
  const double unit_100_ns = 10000000.0;
  double x = unit_100_ns * (buffer_size / (double)sample_rate);
  int64_t period = (int64_t)(double)(int64_t)(x + (x < 0.0 ? -0.5 : 0.5));
  return 1000.0 * period / unit_100_ns;
}

/* ------------------------------------------------------------------------------------------------
 * buffers
 * ---------------------------------------------------------------------------------------------- */

/* core/audio_buffer.h:67-71 */
void wbo_clear(float* const* ch, uint32_t n_channels, uint32_t n_samples) {
  for (uint32_t i = 0; i < n_channels; i++)
    memset(ch[i], 0, (size_t)n_samples * sizeof(float));
}

/* core/audio_buffer.h:73-82 */
void wbo_mix(float* const* dst, const float* const* src, uint32_t n_channels, uint32_t n) {
  for (uint32_t i = 0; i < n_channels; i++) {
    const float* o = src[i];
    float* b = dst[i];
    for (uint32_t j = 0; j < n; j++)
      b[j] += o[j];
  }
}

/* dsp/dsp_ops.h:27-31 */
void wbo_apply_gain(float* buf, uint32_t count, float gain) {
  for (uint32_t i = 0; i < count; i++)
    buf[i] *= gain;
}

/* engine/vu_meter.h:20-25: new_level = max(new_level, abs(x)) with math::max(a,b)= b<a?a:b and
 * math::abs(x)= x<0?-x:x (core_math.h:19-31). */
float wbo_abs_max(const float* buf, uint32_t count) {
  float lvl = 0.0f;
  for (uint32_t i = 0; i < count; i++) {
    float v = buf[i];
    float a = v < 0 ? -v : v;
    lvl = a < lvl ? lvl : a;
  }
  return lvl;
}

/* engine/engine.cpp:1627-1636 (comparison against double literals 1.0 / -1.0) */
void wbo_master_clamp(float* const* ch, uint32_t n_channels, uint32_t n_samples) {
  for (uint32_t i = 0; i < n_channels; i++) {
    float* c = ch[i];
    for (uint32_t j = 0; j < n_samples; j++) {
      if (c[j] > 1.0)
        c[j] = 1.0;
      else if (c[j] < -1.0)
        c[j] = -1.0;
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * output format conversion, core/audio_format_conv.cpp
 * ---------------------------------------------------------------------------------------------- */

/* audio_format_conv.cpp:5-20 */
void wbo_f32_to_interleaved_i16(int16_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  const float min_val = -(float)INT16_MIN; /* 32768 */
  const float max_val = (float)INT16_MAX;  /* 32767 */
  for (uint32_t c = 0; c < nch; c++) {
    const float* s = src[c] + off;
    for (size_t i = 0; i < n; i++) {
      float v = s[i];
      dst[i * nch + c] = (int16_t)(v > 0.0f ? v * max_val : v * min_val);
    }
  }
}

/* audio_format_conv.cpp:22-43 — NOTE the reference's destination index ignores the channel and the
 * channel count (every channel overwrites bytes [0, 3n)); restated as written. */
void wbo_f32_to_interleaved_i24(uint8_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  const float min_val = 8388608.0f, max_val = 8388607.0f;
  for (uint32_t c = 0; c < nch; c++) {
    const float* s = src[c] + off;
    size_t num_bytes = n * 3, j = 0;
    for (size_t i = 0; i < num_bytes; i += 3) {
      float v = s[j];
      int32_t q = v > 0.0f ? (int32_t)(v * max_val) : (int32_t)(v * min_val);
      dst[i + 0] = (uint8_t)(q);
      dst[i + 1] = (uint8_t)(q >> 8);
      dst[i + 2] = (uint8_t)(q >> 16);
      j++;
    }
  }
}

/* audio_format_conv.cpp:45-60 */
void wbo_f32_to_interleaved_i24_x8(int32_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  const float min_val = 8388608.0f, max_val = 8388607.0f;
  for (uint32_t c = 0; c < nch; c++) {
    const float* s = src[c] + off;
    for (size_t i = 0; i < n; i++) {
      float v = s[i];
      int32_t q = v > 0.0f ? (int32_t)(v * max_val) : (int32_t)(v * min_val);
      dst[i * nch + c] = (int32_t)(q & 0xFFFFFF);
    }
  }
}

/* audio_format_conv.cpp:62-77 */
void wbo_f32_to_interleaved_i32(int32_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  const double min_val = -(double)INT32_MIN, max_val = (double)INT32_MAX;
  for (uint32_t c = 0; c < nch; c++) {
    const float* s = src[c] + off;
    for (size_t i = 0; i < n; i++) {
      float v = s[i];
      dst[i * nch + c] = (int32_t)(v > 0.0f ? (double)v * max_val : (double)v * min_val);
    }
  }
}

/* audio_format_conv.cpp:79-91 */
void wbo_f32_to_interleaved_f32(float* dst, const float* const* src, size_t off, size_t n, uint32_t nch) {
  for (uint32_t c = 0; c < nch; c++) {
    const float* s = src[c] + off;
    for (size_t i = 0; i < n; i++)
      dst[i * nch + c] = s[i];
  }
}

/* ------------------------------------------------------------------------------------------------
 * Sampler
 * ---------------------------------------------------------------------------------------------- */

/* dsp/sampler.h:18-27 */
void wbo_sampler_reset(wbo_sampler* s, double sample_offset, double speed, double src_rate, double dst_rate) {
  s->playback_speed = (src_rate / dst_rate) * speed;
  s->sample_offset = sample_offset;
}

static inline float clampf(float x, float lo, float hi) { /* core_math.h:33-37 */
  float m = x < hi ? x : hi;
  return m > lo ? m : lo;
}
static inline double clampd(double x, double lo, double hi) {
  double m = x < hi ? x : hi;
  return m > lo ? m : lo;
}

/* dsp/sampler.cpp:34-59 sample_linear<T,Fmt>; normalisers from :7-18.
 * Q1: the reference indexes src_channels[i] with NO "% channels" wrap here (reads out of bounds for
 * a mono clip into a stereo bus); the build DEFINES the wrap (i % channels) for both paths.
 * Q12: a NEGATIVE playback speed (calc_resize_clip with stretch, clip_edit.h:59-67,110-118: sample_count /
 * (old_length + num_samples) goes negative when a clip is shrunk by more than its sample's stretched length) makes the
 * position run backwards and below zero; `src_sample[ix]` with ix < 0 (sampler.cpp:53-54) then reads the heap in front
 * of the channel array — undefined behaviour, in the compiled reference whatever the allocator left there (its value
 * changes from run to run).  The build DEFINES a tap at a negative index as 0 (the mirror image of the 16 zero frames
 * behind a clip, Q2); positions, ix = trunc(x) and the NEGATIVE fraction fx = x - ix are the reference's arithmetic. */
#define TAP(src, k) ((k) < 0 ? 0 : (src)[(k)])
static void linear_f32(const wbo_sample* smp, uint32_t nch, uint32_t n, uint32_t boff, float gain, double speed,
                       double pos, float* const* out) {
  for (int32_t i = 0; i < (int32_t)nch; i++) {
    const float* src = (const float*)smp->data[(uint32_t)i % smp->channels];
    float* dst = out[i] + boff;
    for (int32_t j = 0; j < (int32_t)n; j++) {
      const double x = pos + ((double)j * speed);
      const int64_t ix = (int64_t)x;
      const float fx = (float)(x - (double)ix);
      const float a = (float)(1.0f * (float)TAP(src, ix));
      const float b = (float)(1.0f * (float)TAP(src, ix + 1));
      const float s = a + fx * (b - a);
      dst[j] += s * gain;
    }
  }
}

static void linear_i16(const wbo_sample* smp, uint32_t nch, uint32_t n, uint32_t boff, float gain, double speed,
                       double pos, float* const* out) {
  const float norm = (float)(1.0 / (double)INT16_MAX); /* sampler.cpp:9-10 */
  for (int32_t i = 0; i < (int32_t)nch; i++) {
    const int16_t* src = (const int16_t*)smp->data[(uint32_t)i % smp->channels];
    float* dst = out[i] + boff;
    for (int32_t j = 0; j < (int32_t)n; j++) {
      const double x = pos + ((double)j * speed);
      const int64_t ix = (int64_t)x;
      const float fx = (float)(x - (double)ix);
      const float a = (float)(norm * (float)TAP(src, ix));
      const float b = (float)(norm * (float)TAP(src, ix + 1));
      const float s = a + fx * (b - a);
      dst[j] += s * gain;
    }
  }
}

static void linear_i32c(const wbo_sample* smp, double norm, uint32_t nch, uint32_t n, uint32_t boff, float gain,
                        double speed, double pos, float* const* out) {
  for (int32_t i = 0; i < (int32_t)nch; i++) {
    const int32_t* src = (const int32_t*)smp->data[(uint32_t)i % smp->channels];
    float* dst = out[i] + boff;
    for (int32_t j = 0; j < (int32_t)n; j++) {
      const double x = pos + ((double)j * speed);
      const int64_t ix = (int64_t)x;
      const float fx = (float)(x - (double)ix);
      const float a = (float)(norm * (double)TAP(src, ix));
      const float b = (float)(norm * (double)TAP(src, ix + 1));
      const float s = a + fx * (b - a);
      dst[j] += s * gain;
    }
  }
}

static void sampler_stream_impl(wbo_sampler* s, const wbo_sample* smp, uint32_t num_channels, uint32_t num_samples,
                                uint32_t buffer_offset, float gain, float* const* dst, uint32_t dst_frames);

/* dsp/sampler.cpp:88-210 */
void wbo_sampler_stream(wbo_sampler* s, const wbo_sample* smp, uint32_t num_channels, uint32_t num_samples,
                        uint32_t buffer_offset, float gain, float* const* dst) {
  sampler_stream_impl(s, smp, num_channels, num_samples, buffer_offset, gain, dst, UINT32_MAX);
}

/* dst_frames: size of the destination buffers.  When Track::process hands Sampler::stream an event_length that
 * wrapped around (uint32 arithmetic at track.cpp:669 with events out of buffer order — after edits while playing,
 * and WITHOUT any edit when a clip edge falls exactly on a block edge: `% buffer_size` at track.cpp:359-361,423-425
 * then yields buffer_offset 0 for an event that belongs to the END of the block, behind events with larger
 * offsets), the reference writes past the end of the block buffer: undefined behaviour.  Oracle and
 * product both DEFINE that case: the write is clipped to the block, the sampler offset still advances by
 * the full (wrapped) length exactly as sampler.cpp:103,209 compute it. */
static void sampler_stream_impl(wbo_sampler* s, const wbo_sample* smp, uint32_t num_channels, uint32_t num_samples,
                                uint32_t buffer_offset, float gain, float* const* dst, uint32_t dst_frames) {
  const float i16_norm = 1.0f / (float)INT16_MAX;                 /* :95 */
  const double i24_norm = 1.0 / (double)((1 << 23) - 1);          /* :96 */
  const double i32_norm = 1.0 / (double)INT32_MAX;                /* :97 */

  if (s->sample_offset >= (double)smp->count)                     /* :99-100 (size_t -> double compare) */
    return;

  double stream_max_length = ((double)smp->count - s->sample_offset) / s->playback_speed; /* :102 */
  double next_sample_offset = s->sample_offset + ((double)num_samples * s->playback_speed); /* :103 */
  double cl = ceil(stream_max_length);
  uint32_t lim = (uint32_t)cl;                                     /* :104 */
  uint32_t n = num_samples < lim ? num_samples : lim;
  if (dst_frames != UINT32_MAX && (uint64_t)buffer_offset + n > dst_frames)
    n = buffer_offset < dst_frames ? dst_frames - buffer_offset : 0;

  if (s->playback_speed == 1.0) {                                  /* :106 (Q3) */
    uint32_t off = (uint32_t)s->sample_offset;                     /* :107 */
    switch (smp->format) {
      case WBO_FMT_I16:                                            /* :109-120 */
        for (uint32_t i = 0; i < num_channels; i++) {
          const int16_t* d = (const int16_t*)smp->data[i % smp->channels];
          float* o = dst[i] + buffer_offset;
          for (uint32_t j = 0; j < n; j++) {
            float v = (float)d[off + j] * i16_norm;
            o[j] += clampf(v, -1.0f, 1.0f) * gain;
          }
        }
        break;
      case WBO_FMT_I24:                                            /* :121-132 */
        for (uint32_t i = 0; i < num_channels; i++) {
          const int32_t* d = (const int32_t*)smp->data[i % smp->channels];
          float* o = dst[i] + buffer_offset;
          for (uint32_t j = 0; j < n; j++) {
            double v = (double)d[off + j] * i24_norm;
            o[j] += (float)clampd(v, -1.0, 1.0) * gain;
          }
        }
        break;
      case WBO_FMT_I32:                                            /* :133-144 */
        for (uint32_t i = 0; i < num_channels; i++) {
          const int32_t* d = (const int32_t*)smp->data[i % smp->channels];
          float* o = dst[i] + buffer_offset;
          for (uint32_t j = 0; j < n; j++) {
            double v = (double)d[off + j] * i32_norm;
            o[j] += (float)clampd(v, -1.0, 1.0) * gain;
          }
        }
        break;
      case WBO_FMT_F32:                                            /* :145-156 */
        for (uint32_t i = 0; i < num_channels; i++) {
          const float* d = (const float*)smp->data[i % smp->channels];
          float* o = dst[i] + buffer_offset;
          for (uint32_t j = 0; j < n; j++) {
            float v = d[off + j];
            o[j] += v * gain;
          }
        }
        break;
      default: break;
    }
  } else {                                                         /* :159-207 */
    switch (smp->format) {
      case WBO_FMT_I16: linear_i16(smp, num_channels, n, buffer_offset, gain, s->playback_speed, s->sample_offset, dst); break;
      case WBO_FMT_I24:
        linear_i32c(smp, (double)(1.0 / (double)((1 << 23) - 1)), num_channels, n, buffer_offset, gain,
                    s->playback_speed, s->sample_offset, dst);
        break;
      case WBO_FMT_I32:
        linear_i32c(smp, (double)(1.0 / (double)INT32_MAX), num_channels, n, buffer_offset, gain, s->playback_speed,
                    s->sample_offset, dst);
        break;
      case WBO_FMT_F32: linear_f32(smp, num_channels, n, buffer_offset, gain, s->playback_speed, s->sample_offset, dst); break;
      default: break;
    }
  }

  s->sample_offset = next_sample_offset;                           /* :209 */
}

/* ------------------------------------------------------------------------------------------------
 * engine / tracks
 * ---------------------------------------------------------------------------------------------- */

enum { PARAM_VOLUME = 0, PARAM_PAN = 1, PARAM_MUTE = 2 }; /* track.h:29-34 */

wbo_engine* wbo_engine_create(uint32_t out_channels, uint32_t buffer_size, uint32_t sample_rate) {
  wbo_engine* e = (wbo_engine*)calloc(1, sizeof(wbo_engine));
  e->out_channels = out_channels;
  e->buffer_size = buffer_size;
  e->sample_rate = sample_rate;
  e->ppq = 96.0; /* engine.h:43 */
  for (uint32_t c = 0; c < out_channels && c < 16; c++)
    e->mixbuf[c] = (float*)calloc(buffer_size, sizeof(float));
  return e;
}

void wbo_engine_destroy(wbo_engine* e) {
  if (!e) return;
  for (uint32_t i = 0; i < e->n_tracks; i++) {
    free(e->tracks[i].clips);
    free(e->tracks[i].free_uids);
  }
  free(e->tracks);
  free(e->samples);
  for (int c = 0; c < 16; c++) free(e->mixbuf[c]);
  free(e->busbuf);
  free(e->seglog);
  free(e);
}

/* engine.cpp:24-30 */
void wbo_engine_set_bpm(wbo_engine* e, double bpm) { e->beat_duration = 60.0 / bpm; }

/* engine.cpp:32-41 */
void wbo_engine_set_playhead(wbo_engine* e, double beat) {
  e->playhead_start = beat;
  e->playhead = beat;
}

void wbo_engine_set_buses(wbo_engine* e, uint32_t n_buses) {
  e->n_buses = n_buses;
  free(e->busbuf);
  e->busbuf = n_buses ? (float*)calloc((size_t)n_buses * e->out_channels * e->buffer_size, sizeof(float)) : NULL;
}

int wbo_engine_add_sample(wbo_engine* e, int format, uint32_t channels, uint32_t sample_rate, size_t count,
                          const void* const* planar) {
  if (e->n_samples_tab == e->cap_samples) {
    e->cap_samples = e->cap_samples ? e->cap_samples * 2 : 64;
    e->samples = (wbo_sample*)realloc(e->samples, e->cap_samples * sizeof(wbo_sample));
  }
  wbo_sample* s = &e->samples[e->n_samples_tab];
  s->format = format;
  s->channels = channels;
  s->sample_rate = sample_rate;
  s->count = count;
  s->data = planar;
  return (int)e->n_samples_tab++;
}

static void push_msg(wbo_track* t, uint32_t id, double value) {
  if (t->n_msgs < WBO_MAX_MSGS) { /* reference ring has capacity 64 and the producer spins when full */
    t->msgs[t->n_msgs].id = id;
    t->msgs[t->n_msgs].value = value;
    t->n_msgs++;
  }
}

/* track.cpp:47-57 */
void wbo_track_set_volume(wbo_engine* e, int track, float db) {
  float v = wbo_db_to_linear(db);
  push_msg(&e->tracks[track], PARAM_VOLUME, (double)v);
}
/* track.cpp:59-68 */
void wbo_track_set_pan(wbo_engine* e, int track, float pan) { push_msg(&e->tracks[track], PARAM_PAN, (double)pan); }
/* track.cpp:70-79 */
void wbo_track_set_mute(wbo_engine* e, int track, int mute) {
  push_msg(&e->tracks[track], PARAM_MUTE, (double)(mute ? 1 : 0));
}
void wbo_track_set_bus(wbo_engine* e, int track, int bus) { e->tracks[track].bus = bus; }

/* engine.cpp:200-208 + Track::Track() track.cpp:22-27 */
int wbo_engine_add_track(wbo_engine* e) {
  if (e->n_tracks == e->cap_tracks) {
    e->cap_tracks = e->cap_tracks ? e->cap_tracks * 2 : 64;
    e->tracks = (wbo_track*)realloc(e->tracks, e->cap_tracks * sizeof(wbo_track));
  }
  wbo_track* t = &e->tracks[e->n_tracks];
  memset(t, 0, sizeof(*t));
  t->current_event.clip = -1;
  t->bus = -1;
  int idx = (int)e->n_tracks++;
  wbo_track_set_volume(e, idx, 0.0f);
  wbo_track_set_pan(e, idx, 0.0f);
  wbo_track_set_mute(e, idx, 0);
  return idx;
}

/* engine.cpp:210-218 (the track object goes away with everything it owns) */
void wbo_engine_delete_track(wbo_engine* e, uint32_t slot) {
  free(e->tracks[slot].clips);
  free(e->tracks[slot].free_uids);
  memmove(&e->tracks[slot], &e->tracks[slot + 1], (e->n_tracks - slot - 1) * sizeof(wbo_track));
  e->n_tracks--;
}

/* engine.cpp:228-243: the Track objects keep their state, only the order (= summation order) changes */
void wbo_engine_move_track(wbo_engine* e, uint32_t from_slot, uint32_t to_slot) {
  if (from_slot == to_slot) return;
  wbo_track tmp = e->tracks[from_slot];
  if (from_slot < to_slot) {
    for (uint32_t i = from_slot; i < to_slot; i++) e->tracks[i] = e->tracks[i + 1];
  } else {
    for (uint32_t i = from_slot; i > to_slot; i--) e->tracks[i] = e->tracks[i - 1];
  }
  e->tracks[to_slot] = tmp;
}

/* engine.cpp:245-262 */
void wbo_engine_solo_track(wbo_engine* e, uint32_t slot) {
  int mute = 0;
  if (e->tracks[slot].ui_solo) {
    e->tracks[slot].ui_solo = 0;
  } else {
    e->tracks[slot].ui_solo = 1;
    wbo_track_set_mute(e, (int)slot, 0);
    mute = 1;
  }
  for (uint32_t i = 0; i < e->n_tracks; i++) {
    if (i == slot) continue;
    if (e->tracks[i].ui_solo) e->tracks[i].ui_solo = 0;
    wbo_track_set_mute(e, (int)i, mute);
  }
}

/* core/algorithm.h:24-40 with the predicate clip->max_time <= value */
static uint32_t lower_bound_max_time(const wbo_clip* clips, uint32_t n, double value) {
  int64_t left = 0, right = (int64_t)n - 1;
  while (left < right) {
    int64_t middle = (left + right) >> 1;
    if (clips[middle].max_time <= value)
      left = middle + 1;
    else
      right = middle;
  }
  return (uint32_t)right;
}

/* exposed so that tests can pin it against the reference's own template (oracle/_ref) */
uint32_t wbo_lower_bound_max_time(const double* max_times, uint32_t n, double value) {
  wbo_clip* tmp = (wbo_clip*)calloc(n ? n : 1, sizeof(wbo_clip));
  for (uint32_t i = 0; i < n; i++) tmp[i].max_time = max_times[i];
  uint32_t r = lower_bound_max_time(tmp, n, value);
  free(tmp);
  return r;
}

/* track.cpp:182-213 — returns 1 and *idx when a next clip exists */
static int find_next_clip(const wbo_track* t, double time_pos, uint32_t* idx) {
  if (t->n_clips == 0) return 0;
  if (t->clips[t->n_clips - 1].max_time < time_pos) return 0;
  *idx = lower_bound_max_time(t->clips, t->n_clips, time_pos); /* clips[i].id == i after ordering */
  return 1;
}

/* track.cpp:220-232 */
static void reset_playback_state(wbo_track* t, double time_pos, int refresh_voices) {
  if (!refresh_voices) {
    uint32_t idx = 0;
    int has = find_next_clip(t, time_pos, &idx);
    t->has_clip_idx = has;
    t->clip_idx = idx;
    t->partially_ended = 0;
  }
  t->refresh_voice = refresh_voices;
}

/* ---- clip placement arithmetic, engine/clip_edit.h ------------------------------------------------ */

static inline double dmax(double a, double b) { return b < a ? a : b; }   /* math::max, core_math.h:28-31 */
static inline double dmin(double a, double b) { return a < b ? a : b; }   /* math::min, core_math.h:23-26 */

/* clip_edit.h:10-16 */
void wbo_calc_move_clip(double clip_min, double clip_max, double relative_pos, double min_move, double* new_min,
                        double* new_max) {
  const double new_pos = dmax(clip_min + relative_pos, min_move);
  *new_min = new_pos;
  *new_max = new_pos + (clip_max - clip_min);
}

/* clip_edit.h:18-126 (audio clip with an asset) */
void wbo_calc_resize_clip(double clip_min, double clip_max, double clip_start_offset, double clip_speed,
                          double sample_rate, double sample_count, double relative_pos, double resize_limit,
                          double min_length, double min_resize_pos, double beat_duration, int is_min, int shift,
                          int stretch, int clamp_at_resize_pos, double* out_min, double* out_max,
                          double* out_start_offset, double* out_speed) {
  if (!is_min) {                                                   /* :29-75 */
    const double old_max = clip_max;
    const double actual_min_length = resize_limit + min_length - clip_min;
    double new_max = dmax(clip_max + relative_pos, 0.0);
    double length = new_max - clip_min;
    if (length < actual_min_length)
      new_max = clip_min + actual_min_length;
    double start_offset = clip_start_offset;
    double new_speed = 1.0;
    if (shift) {
      double mult = clip_speed;
      start_offset = wbo_samples_to_beat(start_offset, sample_rate, beat_duration);
      if (old_max < new_max)
        start_offset -= (new_max - old_max) * mult;
      else
        start_offset += (old_max - new_max) * mult;
      start_offset = dmax(start_offset, 0.0);
      start_offset = dmin(start_offset, sample_count);
      start_offset = wbo_beat_to_samples(start_offset, sample_rate, beat_duration);
    }
    if (stretch) {
      double old_length = sample_count / clip_speed;
      double num_samples = wbo_beat_to_samples(relative_pos, sample_rate, beat_duration);
      new_speed = sample_count / (old_length + num_samples);
    }
    *out_min = clip_min;
    *out_max = new_max;
    *out_start_offset = start_offset;
    *out_speed = new_speed;
    return;
  }
  const double old_min = clip_min;                                 /* :77-125 */
  const double actual_min_length = clip_max - resize_limit + min_length;
  double new_min = dmax(clip_min + relative_pos, 0.0);
  double length = clip_max - new_min;
  if (length < actual_min_length)
    new_min = clip_max - actual_min_length;
  if (clamp_at_resize_pos && new_min < min_resize_pos)
    new_min = min_resize_pos;
  double start_offset = clip_start_offset;
  double new_speed = 1.0;
  if (!shift) {
    start_offset = wbo_samples_to_beat(start_offset, sample_rate, beat_duration);
    if (old_min < new_min)
      start_offset -= old_min - new_min;
    else
      start_offset += new_min - old_min;
    if (start_offset < 0.0)
      new_min = new_min - start_offset;
    start_offset = dmax(start_offset, 0.0);
    start_offset = wbo_beat_to_samples(start_offset, sample_rate, beat_duration);
  }
  if (stretch) {
    double old_length = sample_count / clip_speed;
    double num_samples = wbo_beat_to_samples(old_min - new_min, sample_rate, beat_duration);
    new_speed = sample_count / (old_length + num_samples);
  }
  *out_min = new_min;
  *out_max = clip_max;
  *out_start_offset = start_offset;
  *out_speed = new_speed;
}

/* clip_edit.h:128-137, audio branch */
double wbo_calc_clip_shift(double start_offset, double relative_pos, double beat_duration, double sample_rate) {
  const double offset_in_beat = wbo_samples_to_beat(start_offset, sample_rate, beat_duration);
  return wbo_beat_to_samples(dmax(offset_in_beat - relative_pos, 0.0), sample_rate, beat_duration);
}

/* clip_edit.h:139-150, audio branch */
double wbo_shift_clip_content(double start_offset, double speed, double sample_rate, double relative_pos,
                              double beat_duration) {
  relative_pos *= speed;
  return wbo_calc_clip_shift(start_offset, relative_pos, beat_duration, sample_rate);
}

/* ---- clip list edits ---------------------------------------------------------------------------- */

/* Track::query_clip_by_range, track.cpp:112-157 */
static int query_clip_by_range(const wbo_track* t, double min, double max, uint32_t* first_out, uint32_t* last_out) {
  if (t->n_clips == 0) return 0;
  if (max <= t->clips[0].min_time) return 0;
  if (min >= t->clips[t->n_clips - 1].max_time) return 0;
  uint32_t first = lower_bound_max_time(t->clips, t->n_clips, min);
  uint32_t last = lower_bound_max_time(t->clips, t->n_clips, max);
  uint32_t first_clip = first, last_clip = last;
  if (first == last && (max <= t->clips[first].min_time || min >= t->clips[last].max_time)) return 0;
  if (min > t->clips[first].max_time) first_clip++;
  if (!(max > t->clips[last].min_time)) last_clip--;
  *first_out = first_clip;
  *last_out = last_clip;
  return 1;
}

int wbo_track_query_clip_by_range(const wbo_engine* e, int track, double min, double max, uint32_t* first, uint32_t* last) {
  return query_clip_by_range(&e->tracks[track], min, max, first, last);
}

static int cmp_clip(const void* a, const void* b) {
  double x = ((const wbo_clip*)a)->min_time, y = ((const wbo_clip*)b)->min_time;
  return (x > y) - (x < y);
}

/* Pool<Clip>::free (core/memory.h:80-86) as Track::destroy_clip calls it (track.h:165-168): the chunk is zeroed and pushed
 * on the head of the pool's free list */
static void free_clip_uid(wbo_track* t, uint32_t uid) {
  if (t->n_free_uids == t->cap_free_uids) {
    t->cap_free_uids = t->cap_free_uids ? t->cap_free_uids * 2 : 8;
    t->free_uids = (uint32_t*)realloc(t->free_uids, t->cap_free_uids * sizeof(uint32_t));
  }
  t->free_uids[t->n_free_uids++] = uid;
}

/* Pool<Clip>::allocate (core/memory.h:65-78) as Track::allocate_clip calls it (track.h:156-158): the head of the free
 * list — the chunk freed LAST — else a chunk that has never been used */
static uint32_t alloc_clip_uid(wbo_engine* e, wbo_track* t) {
  if (t->n_free_uids) return t->free_uids[--t->n_free_uids];
  return ++e->next_clip_uid;
}

/* Track::update_clip_ordering, track.cpp:159-180: drop deleted clips (destroyed at once, in list order: :170-172), sort
 * by min_time */
static void update_clip_ordering(wbo_track* t) {
  uint32_t n = 0;
  for (uint32_t i = 0; i < t->n_clips; i++) {
    if (t->clips[i].deleted)
      free_clip_uid(t, t->clips[i].uid);
    else
      t->clips[n++] = t->clips[i];
  }
  t->n_clips = n;
  qsort(t->clips, t->n_clips, sizeof(wbo_clip), cmp_clip);
}

static wbo_clip* push_clip_uid(wbo_track* t, uint32_t uid) {
  if (t->n_clips == t->cap_clips) {
    t->cap_clips = t->cap_clips ? t->cap_clips * 2 : 4;
    t->clips = (wbo_clip*)realloc(t->clips, t->cap_clips * sizeof(wbo_clip));
  }
  wbo_clip* c = &t->clips[t->n_clips++];
  memset(c, 0, sizeof(*c));
  c->uid = uid;
  return c;
}

static wbo_clip* push_clip(wbo_engine* e, wbo_track* t) { return push_clip_uid(t, alloc_clip_uid(e, t)); }

static double clip_shift(const wbo_engine* e, const wbo_clip* c, double relative_pos) {
  return wbo_shift_clip_content(c->start_offset, c->speed, (double)e->samples[c->sample].sample_rate, relative_pos,
                                e->beat_duration);
}

/* Engine::reserve_track_region, engine.cpp:478-569.  ignore_uid = 0: no clip is ignored. */
static void reserve_track_region(wbo_engine* e, wbo_track* t, uint32_t first_clip, uint32_t last_clip, double min,
                                 double max, uint32_t ignore_uid) {
  if (t->n_clips == 0) return;
  if (first_clip == last_clip) {                                   /* :493-533 */
    wbo_clip* clip = &t->clips[first_clip];
    if (clip->uid == ignore_uid) return;
    if (min > clip->min_time && max < clip->max_time) {            /* split into two parts */
      wbo_clip copy = *clip;
      wbo_clip* nc = push_clip(e, t);                              /* may move t->clips */
      clip = &t->clips[first_clip];
      uint32_t uid = nc->uid;
      *nc = copy;                                                  /* Clip(const Clip&), clip.h:91-111: `deleted` and */
      nc->internal_state_changed = 0;                              /* `internal_state_changed` keep their defaults   */
      nc->deleted = 0;
      nc->uid = uid;
      nc->min_time = max;
      nc->start_offset = clip_shift(e, nc, clip->min_time - max);
      clip->max_time = min;
    } else if (min > clip->min_time) {
      clip->max_time = min;
    } else if (max < clip->max_time) {
      clip->start_offset = clip_shift(e, clip, clip->min_time - max);
      clip->min_time = max;
    } else {
      clip->deleted = 1;
    }
    return;
  }
  wbo_clip* first = &t->clips[first_clip];                         /* :535-568 */
  wbo_clip* last = &t->clips[last_clip];
  if (first->uid != ignore_uid && min > first->min_time) {
    first->max_time = min;
    first_clip++;
  }
  if (last->uid != ignore_uid && max < last->max_time) {
    last->start_offset = clip_shift(e, last, last->min_time - max);
    last->min_time = max;
    last_clip--;
  }
  if (first_clip <= last_clip && last_clip < t->n_clips)
    for (uint32_t i = first_clip; i <= last_clip; i++)
      if (t->clips[i].uid != ignore_uid) t->clips[i].deleted = 1;
}

/* engine.cpp:293-309 add_audio_clip -> :409-461 add_to_cliplist */
int wbo_engine_add_audio_clip(wbo_engine* e, int track, double min_time, double max_time, double start_offset,
                              int sample, double speed, float gain) {
  wbo_track* t = &e->tracks[track];
  const int empty = t->n_clips == 0;
  const int back = !empty && t->clips[t->n_clips - 1].max_time < min_time;
  const int front = !empty && !back && t->clips[0].min_time > max_time;
  uint32_t qf = 0, ql = 0;
  const int hit = (!empty && !back && !front) ? query_clip_by_range(t, min_time, max_time, &qf, &ql) : 0;
  const uint32_t uid = alloc_clip_uid(e, t);                       /* :302 the Clip object exists before the list is touched */
  if (hit) reserve_track_region(e, t, qf, ql, min_time, max_time, 0);
  wbo_clip* c = push_clip_uid(t, uid);
  c->min_time = min_time;
  c->max_time = max_time;
  c->start_offset = start_offset;
  c->speed = speed;
  c->gain = gain;
  c->sample = sample;
  update_clip_ordering(t);
  reset_playback_state(t, e->playhead, 1);                         /* engine.cpp:416,426,437,449,459 */
  return 0;
}

/* engine.cpp:346-363 */
int wbo_engine_move_clip(wbo_engine* e, int track, uint32_t clip, double relative_pos) {
  wbo_track* t = &e->tracks[track];
  if (clip >= t->n_clips) return -4;
  if (relative_pos == 0.0) return 0;
  const uint32_t uid = t->clips[clip].uid;
  double mn, mx;
  wbo_calc_move_clip(t->clips[clip].min_time, t->clips[clip].max_time, relative_pos, 0.0, &mn, &mx);
  uint32_t qf, ql;
  if (query_clip_by_range(t, mn, mx, &qf, &ql)) reserve_track_region(e, t, qf, ql, mn, mx, uid);
  for (uint32_t i = 0; i < t->n_clips; i++)
    if (t->clips[i].uid == uid) {
      t->clips[i].min_time = mn;
      t->clips[i].max_time = mx;
      t->clips[i].internal_state_changed = 1;
    }
  update_clip_ordering(t);
  reset_playback_state(t, e->playhead, 1);
  return 0;
}

/* engine.cpp:365-398 */
int wbo_engine_resize_clip(wbo_engine* e, int track, uint32_t clip, double relative_pos, double resize_limit,
                           double min_length, int left_side, int shift, int stretch) {
  wbo_track* t = &e->tracks[track];
  if (clip >= t->n_clips) return -4;
  if (relative_pos == 0.0) return 0;
  const wbo_clip c0 = t->clips[clip];
  const wbo_sample* smp = &e->samples[c0.sample];
  double mn, mx, so, sp;
  wbo_calc_resize_clip(c0.min_time, c0.max_time, c0.start_offset, c0.speed, (double)smp->sample_rate, (double)smp->count,
                       relative_pos, resize_limit, min_length, c0.min_time, e->beat_duration, left_side, shift, stretch,
                       0, &mn, &mx, &so, &sp);
  uint32_t qf, ql;
  if (query_clip_by_range(t, mn, mx, &qf, &ql)) reserve_track_region(e, t, qf, ql, mn, mx, c0.uid);
  for (uint32_t i = 0; i < t->n_clips; i++)
    if (t->clips[i].uid == c0.uid) {
      if (left_side)
        t->clips[i].min_time = mn;
      else
        t->clips[i].max_time = mx;
      t->clips[i].start_offset = so;
      if (stretch) t->clips[i].speed = sp;
      t->clips[i].internal_state_changed = (shift || stretch) ? 1 : 0;
    }
  update_clip_ordering(t);
  reset_playback_state(t, e->playhead, 1);
  return 0;
}

/* engine.cpp:400-407 */
int wbo_engine_delete_clip(wbo_engine* e, int track, uint32_t clip) {
  wbo_track* t = &e->tracks[track];
  if (clip >= t->n_clips) return -4;
  t->clips[clip].deleted = 1;
  update_clip_ordering(t);
  reset_playback_state(t, e->playhead, 1);
  return 0;
}

/* engine.cpp:1460-1464 */
int wbo_engine_set_clip_gain(wbo_engine* e, int track, uint32_t clip, float gain) {
  wbo_track* t = &e->tracks[track];
  if (clip >= t->n_clips) return -4;
  t->clips[clip].gain = gain;
  return 0;
}

/* engine.cpp:463-475 */
int wbo_engine_delete_region(wbo_engine* e, int track, double min, double max) {
  wbo_track* t = &e->tracks[track];
  uint32_t qf, ql;
  if (!query_clip_by_range(t, min, max, &qf, &ql)) return 0;
  reserve_track_region(e, t, qf, ql, min, max, 0);
  update_clip_ordering(t);
  reset_playback_state(t, e->playhead, 1);
  return 0;
}

uint32_t wbo_track_clip_count(const wbo_engine* e, int track) { return e->tracks[track].n_clips; }
const wbo_clip* wbo_track_clip(const wbo_engine* e, int track, uint32_t i) { return &e->tracks[track].clips[i]; }

/* engine.cpp:68-80 */
void wbo_engine_play(wbo_engine* e) {
  for (uint32_t i = 0; i < e->n_tracks; i++)
    reset_playback_state(&e->tracks[i], e->playhead_start, 0);
  e->sample_position = 0;
  e->playing = 1;
}

/* engine.cpp:82-93 + Track::stop track.cpp:249-256 */
void wbo_engine_stop(wbo_engine* e) {
  e->playing = 0;
  e->playhead = e->playhead_start;
  for (uint32_t i = 0; i < e->n_tracks; i++) {
    wbo_track* t = &e->tracks[i];
    memset(&t->current_event, 0, sizeof(t->current_event));
    t->current_event.clip = -1;
    t->n_events = 0;
  }
}

static void push_event(wbo_track* t, int type, uint32_t buffer_offset, double time, double speed,
                       uint64_t sample_offset, int clip) {
  if (t->n_events < WBO_MAX_EVENTS) {
    wbo_event* ev = &t->events[t->n_events++];
    ev->type = type;
    ev->buffer_offset = buffer_offset;
    ev->time = time;
    ev->speed = speed;
    ev->sample_offset = sample_offset;
    ev->clip = clip;
  }
}

/* track.cpp:258-451, audio branch (MIDI clips / recording are out of scope) */
void wbo_track_process_event(wbo_engine* e, wbo_track* t, double start_time, double end_time, double sample_position,
                             double beat_duration, double buffer_duration, double sample_rate, uint32_t buffer_size) {
  (void)e;
  (void)buffer_duration;
  if (t->n_clips == 0) {                                           /* :268-284 */
    if (t->refresh_voice) {
      push_event(t, WBO_EV_STOP, 0, start_time, 0.0, 0, -1);
      t->has_clip_idx = 0;
      t->refresh_voice = 0;
    }
    return;
  }

  uint32_t num_clips = t->n_clips;
  if (t->refresh_voice) {                                          /* :287-340 */
    uint32_t at = 0;
    int has_at = find_next_clip(t, start_time, &at);
    if (has_at) {
      if (t->has_clip_idx) {
        uint32_t idx = t->clip_idx;
        if (idx < num_clips) {
          const wbo_clip* clip = &t->clips[at];
          if (at != idx && start_time >= clip->min_time && start_time <= clip->max_time) {
            push_event(t, WBO_EV_STOP, 0, start_time, 0.0, 0, -1);
            t->clip_idx = at;
            t->partially_ended = 0;
          } else if (at == idx && (start_time < clip->min_time || start_time > clip->max_time)) {
            push_event(t, WBO_EV_STOP, 0, start_time, 0.0, 0, -1);
            t->clip_idx = at;
            t->partially_ended = 0;
          }
        }
      } else {
        t->has_clip_idx = 1;
        t->clip_idx = at;
      }
    } else {
      push_event(t, WBO_EV_STOP, 0, start_time, 0.0, 0, -1);
      t->has_clip_idx = 0;
    }
    t->refresh_voice = 0;
  }

  if (!t->has_clip_idx)                                            /* :342-346 */
    return;

  uint32_t next_clip = t->clip_idx;
  while (next_clip < num_clips) {                                  /* :349-446 */
    wbo_clip* clip = &t->clips[next_clip];
    double min_time = clip->min_time;
    double max_time = clip->max_time;

    if (min_time > end_time)
      break;

    if (min_time >= start_time) {                                  /* :357-374 started from beginning */
      double offset_from_start = wbo_beat_to_samples(min_time - start_time, sample_rate, beat_duration);
      double sample_offset = sample_position + offset_from_start;
      uint32_t buffer_offset = (uint32_t)((uint64_t)sample_offset % (uint64_t)buffer_size);
      push_event(t, WBO_EV_PLAY, buffer_offset, min_time, clip->speed, (uint64_t)clip->start_offset, (int)next_clip);
      clip->internal_state_changed = 0;
    } else if (start_time > min_time && !t->partially_ended) {     /* :375-393 started in the middle (Q5) */
      double relative_start_time = start_time - min_time;
      double sample_pos = wbo_beat_to_samples(relative_start_time, sample_rate, beat_duration);
      uint64_t sample_offset = (uint64_t)(clip->start_offset + (sample_pos * clip->speed));
      push_event(t, WBO_EV_PLAY, 0, start_time, clip->speed, sample_offset, (int)next_clip);
      clip->internal_state_changed = 0;
    } else if (clip->internal_state_changed && t->partially_ended) { /* :394-419 */
      double relative_start_time = start_time - min_time;
      double sample_pos = wbo_beat_to_samples(relative_start_time, sample_rate, beat_duration);
      uint64_t sample_offset = (uint64_t)(clip->start_offset + (sample_pos * clip->speed));
      push_event(t, WBO_EV_STOP, 0, start_time, 0.0, 0, -1);
      push_event(t, WBO_EV_PLAY, 0, start_time, clip->speed, sample_offset, (int)next_clip);
      clip->internal_state_changed = 0;
    }

    if (max_time <= end_time) {                                    /* :421-434 */
      double offset_from_start = wbo_beat_to_samples(max_time - start_time, sample_rate, beat_duration);
      double sample_offset = sample_position + offset_from_start;
      uint32_t buffer_offset = (uint32_t)((uint64_t)sample_offset % (uint64_t)buffer_size);
      push_event(t, WBO_EV_STOP, buffer_offset, max_time, 0.0, 0, -1);
      t->partially_ended = 0;
    } else {                                                       /* :435-442 */
      t->partially_ended = 1;
      break;
    }
    next_clip++;
  }
  t->clip_idx = next_clip;                                         /* :450 */
}

void wbo_engine_enable_seglog(wbo_engine* e, int on) { e->seglog_enabled = on; }

static void log_stream(wbo_engine* e, const wbo_track* t, uint32_t dst_start, uint32_t len) {
  if (!e->seglog_enabled) return;
  if (e->n_seglog == e->cap_seglog) {
    e->cap_seglog = e->cap_seglog ? e->cap_seglog * 2 : 256;
    e->seglog = (wbo_seglog*)realloc(e->seglog, e->cap_seglog * sizeof(wbo_seglog));
  }
  wbo_seglog* s = &e->seglog[e->n_seglog++];
  s->playback_speed = t->sampler.playback_speed;
  s->sample_offset = t->sampler.sample_offset;
  s->track = (uint32_t)(t - e->tracks);
  s->dst_start = dst_start;
  s->len = len;
  s->gain = t->cur_gain;
  s->sample = t->cur_sample;
}

/* current_audio_event.clip->audio.gain is read at every stream call (track.cpp:676,716): follow the clip by
 * identity so that set_clip_gain takes effect on the clip that is already playing.
 *
 * Quirk Q10 — the clip that is sounding was destroyed by an edit (delete_clip, delete_region, or an add / move / resize
 * whose reserve_track_region covered it): Track::update_clip_ordering destroys it at once (track.cpp:159-175),
 * Pool::free zeroes its chunk (core/memory.h:80-86), no event follows for the track, and Track::process keeps streaming
 * current_audio_event through the dangling pointer: `gain` reads 0.0f — the sampler keeps advancing, the track is
 * silent until its next event.  (Use after free in the reference, deterministic in the compiled engine: the chunk stays
 * mapped.)  When a later add / split on the same track takes the chunk over, the read returns the NEW clip's gain: the
 * uid stands for the chunk (alloc_clip_uid / free_clip_uid), so the search below finds exactly that clip. */
static void refresh_current_gain(wbo_track* t) {
  if (t->current_event.type != WBO_EV_PLAY) return;
  for (uint32_t i = 0; i < t->n_clips; i++)
    if (t->clips[i].uid == t->cur_clip_uid) {
      t->cur_gain = t->clips[i].gain;
      return;
    }
  t->cur_gain = 0.0f;
}

/* track.cpp:587-736 (no plugin: write_buffer == output_buffer; Q6 fenced off) */
static void track_process(wbo_engine* e, wbo_track* t, float* const* out, double sample_rate, double beat_duration,
                          double buffer_duration_in_beats, double sample_position, double start_time, double end_time,
                          int playing) {
  const uint32_t n_samples = e->buffer_size, n_channels = e->out_channels;

  /* :602 process_track_messages (:773-779) then :618-643 apply param_queue */
  if (playing)
    wbo_track_process_event(e, t, start_time, end_time, sample_position, beat_duration, buffer_duration_in_beats,
                            sample_rate, n_samples);
  for (uint32_t i = 0; i < t->n_msgs; i++) {
    double value = t->msgs[i].value;
    switch (t->msgs[i].id) {
      case PARAM_VOLUME: t->volume = (float)value; break;
      case PARAM_PAN:
        t->pan = (float)value;
        wbo_pan_coefs(t->pan, WBO_PAN_CP_3DB, &t->pan_coeffs[0], &t->pan_coeffs[1]);
        break;
      case PARAM_MUTE: t->mute = value > 0.0 ? 1 : 0; break;
      default: break;
    }
  }
  t->n_msgs = 0; /* :735 */

  if (playing) {                                                   /* :664-724 */
    refresh_current_gain(t);
    uint32_t next = 0, end = t->n_events;
    uint32_t start_sample = 0;
    while (start_sample < n_samples) {
      if (next != end) {
        const wbo_event* ne = &t->events[next];
        uint32_t event_length = ne->buffer_offset - start_sample;  /* uint32 arithmetic, as in the reference */
        if (t->current_event.type == WBO_EV_PLAY) {
          log_stream(e, t, start_sample, event_length);
          sampler_stream_impl(&t->sampler, &e->samples[t->cur_sample], n_channels, event_length, start_sample,
                              t->cur_gain, out, n_samples);
        }
        if (ne->type == WBO_EV_PLAY) {                             /* :687-697 */
          const wbo_clip* clip = &t->clips[ne->clip];
          const wbo_sample* smp = &e->samples[clip->sample];
          wbo_sampler_reset(&t->sampler, (double)ne->sample_offset, ne->speed, (double)smp->sample_rate, sample_rate);
          t->cur_gain = clip->gain;
          t->cur_sample = clip->sample;
          t->cur_clip_uid = clip->uid;
        }
        t->current_event = *ne;
        start_sample += event_length;
        next++;
      } else {
        uint32_t event_length = n_samples - start_sample;
        if (t->current_event.type == WBO_EV_PLAY) {
          log_stream(e, t, start_sample, event_length);
          sampler_stream_impl(&t->sampler, &e->samples[t->cur_sample], n_channels, event_length, start_sample,
                              t->cur_gain, out, n_samples);
        }
        start_sample = n_samples;
      }
    }
  }

  float volume = t->mute ? 0.0f : t->volume;                       /* :728-733 */
  for (uint32_t i = 0; i < n_channels; i++) {
    float g = volume * t->pan_coeffs[i < 2 ? i : 1];
    wbo_apply_gain(out[i], n_samples, g);
    float p = wbo_abs_max(out[i], n_samples);
    if (i < 2) {
      t->block_peak[i] = p;
      if (t->level[i] < p) t->level[i] = p;                        /* vu_meter.h:26-29 */
    }
  }
}

/* engine.cpp:1576-1654.  With n_buses > 0 (extension A13, not in the reference): each track's block is
 * mixed into its bus in track order (AudioBuffer::mix), then buses 0..n-1 are mixed into the output in
 * bus order, then the clamp. */
void wbo_engine_process(wbo_engine* e, float* const* out, float* bus_out) { wbo_engine_process_ex(e, out, bus_out, 1); }

static void engine_process_impl(wbo_engine* e, float* const* out, float* bus_out, int clamp, int keep_out);

void wbo_engine_process_ex(wbo_engine* e, float* const* out, float* bus_out, int clamp) {
  engine_process_impl(e, out, bus_out, clamp, 0);
}

/* NOT in the reference: Engine::process WITHOUT the output_buffer.clear() of engine.cpp:1598 — `out` holds the running,
 * un-clamped sum of the tracks that precede this engine's in a session split over several engines, and the track loop
 * (engine.cpp:1600-1617) continues it.  The checker for the product's wbx_set_master_init / WBX_DIST_CHAIN: a chain of
 * such calls over the shards is, addition for addition, one Engine::process over all tracks. */
void wbo_engine_process_from(wbo_engine* e, float* const* out, int clamp) { engine_process_impl(e, out, NULL, clamp, 1); }

static void engine_process_impl(wbo_engine* e, float* const* out, float* bus_out, int clamp, int keep_out) {
  const uint32_t F = e->buffer_size, C = e->out_channels;
  double sample_rate = (double)e->sample_rate;
  double buffer_duration = (double)F / sample_rate;                /* :1578 */
  double current_beat_duration = e->beat_duration;
  double current_playhead_position = e->playhead;
  double buffer_duration_in_beats = buffer_duration / current_beat_duration; /* :1581 */
  double next_playhead_pos = e->playhead + buffer_duration_in_beats;         /* :1582 */
  int currently_playing = e->playing;

  for (uint32_t i = 0; i < e->n_tracks; i++)                       /* :1589-1596 */
    e->tracks[i].n_events = 0;
  e->n_seglog = 0;

  if (!keep_out) wbo_clear(out, C, F);                             /* :1598 */
  if (e->n_buses)
    memset(e->busbuf, 0, (size_t)e->n_buses * C * F * sizeof(float));

  for (uint32_t i = 0; i < e->n_tracks; i++) {                     /* :1600-1617 */
    wbo_track* t = &e->tracks[i];
    wbo_clear(e->mixbuf, C, F);
    track_process(e, t, e->mixbuf, sample_rate, current_beat_duration, buffer_duration_in_beats, e->sample_position,
                  current_playhead_position, next_playhead_pos, currently_playing);
    if (e->n_buses && t->bus >= 0) {
      float* b[16];
      for (uint32_t c = 0; c < C; c++) b[c] = e->busbuf + ((size_t)t->bus * C + c) * F;
      wbo_mix(b, (const float* const*)e->mixbuf, C, F);
    } else {
      wbo_mix(out, (const float* const*)e->mixbuf, C, F);
    }
  }
  if (e->n_buses) {
    for (uint32_t u = 0; u < e->n_buses; u++) {
      const float* b[16];
      for (uint32_t c = 0; c < C; c++) b[c] = e->busbuf + ((size_t)u * C + c) * F;
      wbo_mix(out, b, C, F);
    }
    if (bus_out) memcpy(bus_out, e->busbuf, (size_t)e->n_buses * C * F * sizeof(float));
  }

  if (currently_playing) {                                         /* :1619-1623 */
    e->sample_position += wbo_beat_to_samples(buffer_duration_in_beats, sample_rate, current_beat_duration);
    e->playhead = next_playhead_pos;
  }

  if (clamp) wbo_master_clamp(out, C, F);                          /* :1627-1636 */
}

/* synthetic input (not part of the reference): u = splitmix64(key ^ i); v = ((u>>40) - 2^23) * 2^-23; v*amp */
static uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void wbo_synth_f32(float* dst, size_t frames, uint64_t key, float amp, size_t pad) {
  for (size_t i = 0; i < frames; i++) {
    uint64_t u = splitmix64(key ^ (uint64_t)i);
    float v = (float)((int64_t)(u >> 40) - (1 << 23)) * 1.1920928955078125e-07f;
    dst[i] = v * amp;
  }
  for (size_t i = 0; i < pad; i++) dst[frames + i] = 0.0f;
}


/* ================================================================================================
 * next rows: clip ingest + waveform mip-maps (both pinned to the reference's own functions, see wb_oracle.h)
 * ================================================================================================ */

/* dsp/sample.cpp:29-43 */
size_t wbo_deinterleave(void* const* dst, const void* src, size_t num_read, size_t written, int channels, size_t elem) {
  for (int i = 0; i < channels; i++) {
    unsigned char* channel_data = (unsigned char*)dst[i];
    const unsigned char* s = (const unsigned char*)src;
    for (size_t j = 0; j < num_read; j++)
      memcpy(channel_data + (written + j) * elem, s + ((size_t)channels * j + (size_t)i) * elem, elem);
  }
  return written + num_read;
}

/* gfx/waveform_visual.cpp:195-239: while (sample_count > 64) { ...; sample_count /= 4; current_mip += 2; } */
uint32_t wbo_mip_levels(size_t count) {
  uint32_t n = 0;
  size_t sample_count = count;
  while (sample_count > 64) {
    n++;
    sample_count /= 4;
  }
  return n;
}

/* :197-199 */
size_t wbo_mip_data_count(size_t count, uint32_t level) {
  const uint32_t current_mip = 1u + 2u * level;
  const size_t block_count = (size_t)1 << (current_mip - 1);
  size_t mip_data_count = count / block_count;
  mip_data_count += mip_data_count % 2;
  return mip_data_count;
}

/* (T)conv as gcc/x86-64 evaluates it: float/double -> int32 with truncation (cvttss2si / cvttsd2si return
 * 0x80000000 for NaN and out-of-range), then the low bits of that int */
static int32_t x86_cvtt_f32(float v) {
  if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
  return (int32_t)v;
}
static int32_t x86_cvtt_f64(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
  return (int32_t)v;
}

#define WBO_MIP_BODY(T, TMIN, TMAX, CONVERT)                                                          \
  for (size_t i = 0; i < output_count; i += 2) {                                                      \
    size_t idx = i * block_count;                                                                     \
    size_t chunk_length = chunk_count < sample_count - idx ? chunk_count : sample_count - idx;        \
    T min_val = TMAX;                                                                                 \
    T max_val = TMIN;                                                                                 \
    size_t min_idx = 0, max_idx = 0;                                                                  \
    for (size_t j = 0; j < chunk_length; j++) {                                                       \
      T value = (T)(CONVERT);                                                                         \
      if (value < min_val) {                                                                          \
        min_val = value;                                                                              \
        min_idx = j;                                                                                  \
      }                                                                                               \
      if (value > max_val) {                                                                          \
        max_val = value;                                                                              \
        max_idx = j;                                                                                  \
      }                                                                                               \
    }                                                                                                 \
    if (max_idx < min_idx) {                                                                          \
      o[i] = max_val;                                                                                 \
      o[i + 1] = min_val;                                                                             \
    } else {                                                                                          \
      o[i] = min_val;                                                                                 \
      o[i + 1] = max_val;                                                                             \
    }                                                                                                 \
  }

/* gfx/waveform_visual.cpp:9-173 */
void wbo_mip_summarize(int format, size_t count, const void* data, uint32_t level, int out_bits, void* out) {
  const uint32_t current_mip = 1u + 2u * level;
  const size_t chunk_count = (size_t)1 << current_mip;
  const size_t block_count = (size_t)1 << (current_mip - 1);
  const size_t output_count = wbo_mip_data_count(count, level);
  const size_t sample_count = count;
  const double tmin = out_bits == 8 ? -128.0 : -32768.0, tmax = out_bits == 8 ? 127.0 : 32767.0;
  if (format == 3) {   /* I16, :58-97 */
    const int16_t* sample = (const int16_t*)data;
    const float conv_div_min = (float)tmin / (float)-32768.0f;
    const float conv_div_max = (float)tmax / (float)32767.0f;
#define CV x86_cvtt_f32((float)sample[idx + j] * (sample[idx + j] >= 0 ? conv_div_max : conv_div_min))
    if (out_bits == 8) {
      int8_t* o = (int8_t*)out;
      WBO_MIP_BODY(int8_t, INT8_MIN, INT8_MAX, CV)
    } else {
      int16_t* o = (int16_t*)out;
      WBO_MIP_BODY(int16_t, INT16_MIN, INT16_MAX, CV)
    }
#undef CV
  } else if (format == 7 || format == 5) {   /* I32, :98-137 (double arithmetic) */
    const int32_t* sample = (const int32_t*)data;
    const double conv_div_min = tmin / (double)INT32_MIN;
    const double conv_div_max = tmax / (double)INT32_MAX;
#define CV x86_cvtt_f64((double)sample[idx + j] * (sample[idx + j] >= 0 ? conv_div_max : conv_div_min))
    if (out_bits == 8) {
      int8_t* o = (int8_t*)out;
      WBO_MIP_BODY(int8_t, INT8_MIN, INT8_MAX, CV)
    } else {
      int16_t* o = (int16_t*)out;
      WBO_MIP_BODY(int16_t, INT16_MIN, INT16_MAX, CV)
    }
#undef CV
  } else if (format == 9) {   /* F32, :138-170: x * (x >= 0 ? T max : -T min) */
    const float* sample = (const float*)data;
    const float kpos = (float)tmax, kneg = (float)(-tmin);
#define CV x86_cvtt_f32((float)sample[idx + j] * (sample[idx + j] >= 0.0f ? kpos : kneg))
    if (out_bits == 8) {
      int8_t* o = (int8_t*)out;
      WBO_MIP_BODY(int8_t, INT8_MIN, INT8_MAX, CV)
    } else {
      int16_t* o = (int16_t*)out;
      WBO_MIP_BODY(int16_t, INT16_MIN, INT16_MAX, CV)
    }
#undef CV
  }
}
