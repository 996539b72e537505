// wbx_seq.h — the clip sequencer and sampler-state arithmetic of one track for one block.
//
// Follows, operation for operation and in fp64 / int64, the reference's
//   Track::process_event (audio branch)       src/engine/track.cpp:258-451
//   the segment loop of Track::process         src/engine/track.cpp:664-724
//   Sampler::reset_state / stream prologue     src/dsp/sampler.h:18-27, src/dsp/sampler.cpp:99-104,209
//   beat_to_samples                            src/core/core_math.h:209-212
// so that every buffer offset, sample offset and segment length is bit-identical to the reference's.
// It is compiled for the device (plan kernel: one lane per track) and for the host (layer-1 segment
// conversion and the CPU-side unit tests of the seek math).  No per-sample work happens here.
//
// Build with -ffp-contract=off: a fused multiply-add anywhere in here changes results.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else   // host-only build of the sequencer (tests/cpp/host_sim.cpp, plain g++): the qualifiers mean nothing there
#define __host__
#define __device__
struct uint4 {   // (no over-alignment: the records it views live in ordinary host containers)
  unsigned int x, y, z, w;
};
#endif
#include <math.h>

#include "wbx_dev.h"

namespace wbx {

enum : uint32_t { EV_NONE = 0, EV_STOP = 1, EV_PLAY = 2 };  // reference EventType, src/engine/event.h:11-15

// core_math.h:209-212 — two separately rounded multiplies
__host__ __device__ inline double beat_to_samples(double beat, double sample_rate, double beat_duration) {
  double sec = beat * beat_duration;
  return sec * sample_rate;
}

// Float -> integer conversions the reference leaves to the compiler where the value may be out of range (undefined in C++,
// deterministic in the compiled engine): stated here as x86-64 performs them, because the device's own conversions SATURATE
// and the host's do not.  cvttsd2si: truncation towards zero; NaN and anything outside [-2^63, 2^63) give 0x8000000000000000.
__host__ __device__ inline long long i64_of_double_x86(double x) {
  if (!(x >= -9223372036854775808.0 && x < 9223372036854775808.0)) return (long long)0x8000000000000000ull;
  return (long long)x;
}
// (uint64_t)x / (size_t)x — track.cpp:378-379,399-400 `(size_t)(start_offset + sample_pos * speed)`, negative for a clip that
// plays backwards (quirk Q12): below 2^63 the signed conversion reinterpreted (a negative x wraps to 2^64 + trunc(x)), from
// 2^63 on the conversion of x - 2^63 with the top bit flipped — the two branches gcc emits
__host__ __device__ inline uint64_t u64_of_double_x86(double x) {
  if (x >= 9223372036854775808.0) return (uint64_t)i64_of_double_x86(x - 9223372036854775808.0) ^ 0x8000000000000000ull;
  return (uint64_t)i64_of_double_x86(x);
}
// (uint32_t)x — sampler.cpp:104,107: converted through a signed 64-bit integer, the low word kept
__host__ __device__ inline uint32_t u32_of_double_x86(double x) { return (uint32_t)(uint64_t)i64_of_double_x86(x); }
__host__ __device__ inline uint32_t u32_of_ceil(double x) { return u32_of_double_x86(ceil(x)); }

// the plan's status word (PlanArgs::status): tracks raise their bits side by side — on the device an atomic OR (the one-launch
// callback reads the word in the same launch, from behind another L2: a plain read-modify-write could stay in a cache)
__host__ __device__ inline void raise_status(uint32_t* status, uint32_t bits) {
  if (!status) return;
#if defined(__HIP_DEVICE_COMPILE__)
  atomicOr(status, bits);
#else
  status[0] |= bits;
#endif
}

// segment 0 of a track-block lives inline in the record
__host__ __device__ inline void set_seg0(DTrackBlock* tb, const DSeg& s) {
  tb->src[0] = s.src[0];
  tb->src[1] = s.src[1];
  tb->pos = s.pos;
  tb->speed = s.speed;
  tb->gain = s.gain;
  tb->dst_start = s.dst_start;
  tb->len = s.len;
  tb->req_len = s.req_len;
  tb->format = s.format;
  tb->flags = s.flags;
  tb->sample = s.sample;
}

__host__ __device__ inline DSeg get_seg0(const DTrackBlock& tb) {
  DSeg s;
  s.src[0] = tb.src[0];
  s.src[1] = tb.src[1];
  s.pos = tb.pos;
  s.speed = tb.speed;
  s.gain = tb.gain;
  s.dst_start = tb.dst_start;
  s.len = tb.len;
  s.req_len = tb.req_len;
  s.format = tb.format;
  s.flags = tb.flags;
  s.sample = tb.sample;
  return s;
}

// find_lower_bound with the predicate clip->max_time <= value (core/algorithm.h:24-40, track.cpp:206)
__host__ __device__ inline uint32_t lower_bound_max_time(const DClip* clips, uint32_t n, double value) {
  long long left = 0, right = (long long)n - 1;
  while (left < right) {
    long long middle = (left + right) >> 1;
    if (clips[middle].max_time <= value)
      left = middle + 1;
    else
      right = middle;
  }
  return (uint32_t)right;
}

// Track::find_next_clip, track.cpp:182-213
__host__ __device__ inline bool find_next_clip(const DClip* clips, uint32_t n, double time_pos, uint32_t* idx) {
  if (n == 0) return false;
  if (clips[n - 1].max_time < time_pos) return false;
  *idx = lower_bound_max_time(clips, n, time_pos);
  return true;
}

// The walker below receives events as process_event produces them and performs the matching iteration
// of Track::process's segment loop immediately.  That is equivalent to building the whole event list
// first (track.cpp:604-615) and walking it afterwards (:664-724): the loop never feeds back into the
// sequencer, and because start_sample always becomes the event's buffer_offset (< buffer_size) under
// the reference's uint32 arithmetic, every event of the list is consumed.
// Register-resident copies of the clip and sample records a track is currently using: in the steady
// state (a clip playing through the block) a planned block then touches no global memory except its
// own 64-B output record.
struct TrackCache {
  DClip clip;
  DSample smp;
  uint32_t clip_idx;   // index within the track's clip list, 0xFFFFFFFF = empty
  uint32_t smp_idx;    // sample id, 0xFFFFFFFF = empty
  // The clip BEHIND the current one is fetched when the current one is: a track is one lane, so a clip boundary would
  // otherwise wait a full global-memory round trip for the next clip's record, with nothing else to run meanwhile.
  DClip next;
  uint32_t next_idx;   // 0xFFFFFFFF = empty
  // A clip whose timeline region outlasts its audio makes one zero-length "finished" stream call per block
  // (sampler.cpp:99-100) for the rest of the region, every one with the same record: they share one template.
  // (what identifies the record of the last such block — kept field by field, in registers: a 64-B struct compared
  //  through a pointer would put the sequencer's locals into scratch memory)
  double fin_pos, fin_speed;
  float fin_gain;
  uint32_t fin_sample, fin_shape;   // shape = dst_start << 16 | req_len
  uint32_t fin_tmpl;   // its template index, 0xFFFFFFFF = none yet
  // templates are reserved kTmplReserve at a time: one atomic round trip per reservation instead of one per block
  // with events (a session cut into short clips has an event in every few blocks of every track)
  uint32_t tmpl_next, tmpl_end;
};
constexpr uint32_t kTmplReserve = 8;   // PlanArgs::tmpl_reserve of batch renders (1 for the one-block callback)

struct BlockWalker {
  DTrackState* st;
  const DSample* samples;
  TrackCache* cache;
  DTrackBlock* tb;          // record being filled
  DSeg* pool;
  uint32_t* pool_count;
  uint32_t pool_chunks;
  uint32_t* status;
  uint32_t n_samples;       // block frames
  uint32_t n_channels;
  double dst_rate;
  uint32_t start_sample;
  uint32_t nseg;
  uint32_t chunk;
  bool dry;                 // advance the state only: no record, no pool chunk, no status bit (plan_segment's run-up)

  // The second stream call of a block is kept in `seg1` (the usual clip boundary: one clip ends, the next starts —
  // it becomes template tmpl + 1 of a ROW_PAIR and never touches the pool); the overflow pool is only allocated
  // when a third call arrives (or, for a pair the hot loop cannot take, when the block is finished).
  DSeg seg1;
  __host__ __device__ bool alloc_chunk() {
    if (dry) return false;
    uint32_t c;
#if defined(__HIP_DEVICE_COMPILE__)
    c = atomicAdd(pool_count, 1u);
#else
    c = (*pool_count)++;
#endif
    if (c >= pool_chunks) {
      raise_status(status, 1u);
      return false;
    }
    chunk = c;
    tb->extra = c;
    pool[(size_t)chunk * kChunk] = seg1;
    return true;
  }
  // where call number `nseg` (>= 2) of this block goes in the overflow pool
  __host__ __device__ DSeg* slot() {
    if (dry) return nullptr;
    if (nseg >= kMaxSegs) {
      raise_status(status, 2u);
      return nullptr;
    }
    if (nseg == 2 && !alloc_chunk()) return nullptr;
    if (chunk == 0xFFFFFFFFu) return nullptr;
    return &pool[(size_t)chunk * kChunk + (nseg - 1)];
  }

  // Sampler::stream up to (not including) the per-sample loops: sampler.cpp:99-104 and :209
  __host__ __device__ const DSample& sample_of(uint32_t id) {
    if (cache->smp_idx != id) {
      cache->smp = samples[id];
      cache->smp_idx = id;
    }
    return cache->smp;
  }

  __host__ __device__ void stream(uint32_t num_samples, uint32_t buffer_offset) {
    const DSample& smp = sample_of(st->cur_sample);
    DSeg* s = nseg >= 2 ? slot() : nullptr;
    DSeg seg;
    seg.src[0] = smp.ch[0];
    seg.src[1] = smp.ch[n_channels > 1 ? 1 : 0];
    seg.pos = st->sample_offset;
    seg.speed = st->playback_speed;
    seg.gain = st->cur_gain;
    seg.dst_start = (uint16_t)buffer_offset;
    seg.req_len = (uint16_t)(num_samples > 0xFFFFu ? 0xFFFFu : num_samples);
    seg.format = (uint8_t)smp.format;
    seg.flags = 0;
    seg.sample = st->cur_sample;
    if (st->sample_offset >= (double)smp.count) {          // :99-100 finished streaming
      seg.len = 0;
      seg.flags = SEG_FINISHED;
    } else {
      const double cnt = (double)smp.count, off = st->sample_offset, sp = st->playback_speed;
      double next_sample_offset = off + ((double)num_samples * sp);                              // :103
      uint32_t n;
      // :102,:104  n = min(num_samples, (uint32)ceil((count - offset) / speed)).  The quotient is only needed
      // near the clip tail: when count - offset exceeds (num_samples + 1) * speed by a whole frame, the
      // rounded quotient is >= num_samples + 1 and (being below 2^32) survives the uint32 conversion, so
      // the minimum is num_samples and the fp64 division can be skipped without changing any result.
      if (sp > 0.0 && off + (double)(num_samples + 1u) * sp + 1.0 <= cnt && (cnt - off) < sp * 4294967040.0) {
        n = num_samples;
      } else {
        double stream_max_length = (cnt - off) / sp;                                             // :102
        uint32_t lim = u32_of_ceil(stream_max_length);                                           // :104
        n = num_samples < lim ? num_samples : lim;
      }
      if (buffer_offset + (uint64_t)n > n_samples) {       // the reference would write out of bounds here
        n = buffer_offset < n_samples ? n_samples - buffer_offset : 0;
        seg.flags |= SEG_CLIPPED;
        raise_status(status, 4u);
      }
      seg.len = (uint16_t)n;
      st->sample_offset = next_sample_offset;                                                    // :209
    }
    if (nseg == 0) {
      set_seg0(tb, seg);
      nseg = 1;
    } else if (nseg == 1) {
      seg1 = seg;
      nseg = 2;
    } else if (s) {
      *s = seg;
      nseg++;
    }
  }

  // one `next_event != end` iteration of the loop at track.cpp:668-709
  __host__ __device__ void on_event(uint32_t type, uint32_t buffer_offset, double speed, uint64_t sample_offset,
                                    const DClip* clip) {
    uint32_t event_length = buffer_offset - start_sample;   // uint32 arithmetic as in the reference
    if (st->cur_type == EV_PLAY) stream(event_length, start_sample);
    if (type == EV_PLAY) {                                  // :687-697 Sampler::reset_state (sampler.h:18-27)
      const DSample& smp = sample_of(clip->sample);
      st->playback_speed = ((double)smp.sample_rate / dst_rate) * speed;
      st->sample_offset = (double)sample_offset;
      st->cur_gain = clip->gain;
      st->cur_sample = clip->sample;
      st->cur_clip_uid = clip->uid;
    }
    st->cur_type = type;
    start_sample += event_length;
  }

  // the `else` arm, track.cpp:710-722
  __host__ __device__ void finish() {
    uint32_t event_length = n_samples - start_sample;
    if (st->cur_type == EV_PLAY) stream(event_length, start_sample);
    start_sample = n_samples;
  }
};

// clip->internal_state_changed = false (track.cpp:373,392,418): the cached copy always, the clip list in
// HBM only when the flag was actually set.
// (flags_left, optional: a count of the set flags in the table, in host memory — the host plans by segments only while it
//  reads zero there, plan_segment below)
__host__ __device__ inline void clear_state_changed(DClip* cached, DClip* global, uint32_t* flags_left) {
  if (cached->internal_state_changed) {
    cached->internal_state_changed = 0;
    global->internal_state_changed = 0;
    if (flags_left) {
#if defined(__HIP_DEVICE_COMPILE__)
      __hip_atomic_fetch_sub(flags_left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
      flags_left[0] -= 1u;
#endif
    }
  }
}

// (uint32)((uint64)sample_offset % buffer_size), track.cpp:359-361,423-425.  A 64-bit remainder by a run-time value is
// a long software routine on the GPU; buffer sizes are powers of two in practice.
__host__ __device__ inline uint32_t mod_buffer(uint64_t x, uint32_t buffer_size) {
  if ((buffer_size & (buffer_size - 1u)) == 0u) return (uint32_t)x & (buffer_size - 1u);
  return (uint32_t)(x % (uint64_t)buffer_size);
}

// Track::process_event, audio branch — track.cpp:258-451.  MIDI clips and recording are out of scope.
__host__ __device__ inline void process_event(BlockWalker& w, DClip* clips, uint32_t num_clips, double start_time,
                                              double end_time, double sample_position, double beat_duration,
                                              double sample_rate, uint32_t buffer_size, uint32_t* flags_left = nullptr) {
  DTrackState* st = w.st;
  if (num_clips == 0) {                                          // :268-284
    if (st->refresh_voice) {
      w.on_event(EV_STOP, 0, 0.0, 0, nullptr);
      st->has_clip_idx = 0;
      st->refresh_voice = 0;
    }
    return;
  }

  if (st->refresh_voice) {                                       // :287-340
    uint32_t at = 0;
    if (find_next_clip(clips, num_clips, start_time, &at)) {
      if (st->has_clip_idx) {
        uint32_t idx = st->clip_idx;
        if (idx < num_clips) {
          const DClip* clip = &clips[at];
          if (at != idx && start_time >= clip->min_time && start_time <= clip->max_time) {
            w.on_event(EV_STOP, 0, 0.0, 0, nullptr);
            st->clip_idx = at;
            st->partially_ended = 0;
          } else if (at == idx && (start_time < clip->min_time || start_time > clip->max_time)) {
            w.on_event(EV_STOP, 0, 0.0, 0, nullptr);
            st->clip_idx = at;
            st->partially_ended = 0;
          }
        }
      } else {
        st->has_clip_idx = 1;
        st->clip_idx = at;
      }
    } else {
      w.on_event(EV_STOP, 0, 0.0, 0, nullptr);
      st->has_clip_idx = 0;
    }
    st->refresh_voice = 0;
  }

  if (!st->has_clip_idx) return;                                 // :342-346

  uint32_t next_clip = st->clip_idx;
  TrackCache* cc = w.cache;
  while (next_clip < num_clips) {                                // :349-446
    if (cc->clip_idx != next_clip) {
      // (always through `next`: choosing between a copy from the cache and a load from HBM would make the compiler
      //  select between ADDRESSES and keep the whole cache in scratch memory instead of registers)
      if (cc->next_idx != next_clip) cc->next = clips[next_clip];
      cc->clip = cc->next;
      cc->clip_idx = next_clip;
      cc->next_idx = 0xFFFFFFFFu;
      if (next_clip + 1u < num_clips) {   // (not needed before the next boundary: the load's latency is hidden)
        cc->next = clips[next_clip + 1u];
        cc->next_idx = next_clip + 1u;
      }
    }
    DClip* clip = &cc->clip;
    double min_time = clip->min_time;
    double max_time = clip->max_time;

    if (min_time > end_time) break;

    if (min_time >= start_time) {                                // :357-374 started from the beginning
      double offset_from_start = beat_to_samples(min_time - start_time, sample_rate, beat_duration);
      double sample_offset = sample_position + offset_from_start;
      uint32_t buffer_offset = mod_buffer(u64_of_double_x86(sample_offset), buffer_size);
      w.on_event(EV_PLAY, buffer_offset, clip->speed, u64_of_double_x86(clip->start_offset), clip);
      clear_state_changed(clip, &clips[next_clip], flags_left);
    } else if (start_time > min_time && !st->partially_ended) {  // :375-393 started in the middle
      double relative_start_time = start_time - min_time;
      double sample_pos = beat_to_samples(relative_start_time, sample_rate, beat_duration);
      uint64_t sample_offset = u64_of_double_x86(clip->start_offset + (sample_pos * clip->speed));
      w.on_event(EV_PLAY, 0, clip->speed, sample_offset, clip);
      clear_state_changed(clip, &clips[next_clip], flags_left);
    } else if (clip->internal_state_changed && st->partially_ended) {  // :394-419
      double relative_start_time = start_time - min_time;
      double sample_pos = beat_to_samples(relative_start_time, sample_rate, beat_duration);
      uint64_t sample_offset = u64_of_double_x86(clip->start_offset + (sample_pos * clip->speed));
      w.on_event(EV_STOP, 0, 0.0, 0, nullptr);
      w.on_event(EV_PLAY, 0, clip->speed, sample_offset, clip);
      clear_state_changed(clip, &clips[next_clip], flags_left);
    }

    if (max_time <= end_time) {                                  // :421-434 reaching the end of the clip
      double offset_from_start = beat_to_samples(max_time - start_time, sample_rate, beat_duration);
      double sample_offset = sample_position + offset_from_start;
      uint32_t buffer_offset = mod_buffer(u64_of_double_x86(sample_offset), buffer_size);
      w.on_event(EV_STOP, buffer_offset, 0.0, 0, nullptr);
      st->partially_ended = 0;
    } else {                                                     // :435-442
      st->partially_ended = 1;
      break;
    }
    next_clip++;
  }
  st->clip_idx = next_clip;                                      // :450
}

// classify a finished track-block record for the mix kernel's wave-uniform dispatch
__host__ __device__ inline uint8_t classify(const DTrackBlock& tb, uint32_t block_frames) {
  if (tb.nseg == 0) return KIND_SILENT;
  if (tb.nseg == 1) {
    const DTrackBlock& s = tb;
    if (s.len == 0) return KIND_SILENT;
    if (s.format != FMT_F32 && s.dst_start == 0 && s.len == block_frames && s.pos >= 0.0 && s.pos < 2147483000.0 &&
        s.speed == 1.0)                                          // integer PCM streamed at unity speed, sampler.cpp:109-144
      return s.format == FMT_I16 ? KIND_UNITY_I16 : KIND_UNITY_I32;
    if (s.format == FMT_F32 && s.dst_start == 0 && s.len == block_frames && s.pos >= 0.0 && s.pos < 2147483000.0) {
      if (s.speed == 1.0) return KIND_UNITY;                     // sampler.cpp:106
      // taps of 4 consecutive frames fit a 5-sample window only while floor(x_e) - floor(x_0) <= e: keep a
      // margin below 1.0 so fp64 rounding of j*speed can never push it over
      if (s.speed > 0.0 && s.speed <= 0.999) return KIND_WINDOW;
      if (s.speed > 0.999 && s.speed <= 4096.0) return KIND_STRIDE;   // downsampling / fast-forward: per-frame taps
    }
    // resampled integer PCM (sampler.cpp:159-207): per-frame taps at any speed
    if (s.format != FMT_F32 && s.dst_start == 0 && s.len == block_frames && s.pos >= 0.0 && s.pos < 2147483000.0 &&
        s.speed > 0.0 && s.speed <= 4096.0)
      return s.speed > 0.999 ? KIND_STRIDE : s.format == FMT_I16 ? KIND_WINDOW_I16 : KIND_WINDOW;   // 24/32-bit: 4-byte containers, the fp32 window loads
  }
  return KIND_GENERIC;
}

// `n` adjacent slots in the render's template array (0xFFFFFFFF + status bit 4 when the array is full), taken from
// the track's current reservation
__host__ __device__ inline uint32_t alloc_template(const PlanArgs& a, TrackCache* cache, uint32_t n = 1u) {
  if (cache->tmpl_next + n > cache->tmpl_end) {
    uint32_t base;
    if (a.tmpl_reserve == 0u) {   // (a track's static pair is used up: cannot happen in a one-block render)
      raise_status(a.status, 16u);
      return 0xFFFFFFFFu;
    }
    const uint32_t take = a.tmpl_reserve > n ? a.tmpl_reserve : n;
#if defined(__HIP_DEVICE_COMPILE__)
    base = atomicAdd(a.tmpl_count, take);
#else
    base = *a.tmpl_count;
    *a.tmpl_count += take;
#endif
    if (base + take > a.tmpl_cap) {
      raise_status(a.status, 16u);
      return 0xFFFFFFFFu;
    }
    cache->tmpl_next = base;
    cache->tmpl_end = base + take;
  }
  const uint32_t i = cache->tmpl_next;
  cache->tmpl_next += n;
  return i;
}

// What the hot loop can render of ONE stream call that covers only part of the block (PlanArgs::masked_rows = level):
// source positions below 2^31; fp32 at unity speed or at a speed the 5-sample window holds (level >= 1), integer PCM at
// unity speed (level 2: sessions whose integer clips all play at the session rate), 16-bit PCM also at a speed the window
// holds (level 3: sessions of 16-bit clips only, the lean 16-bit family of mix_kernel), every format at every streamable
// speed (level 4: the everything family).  KIND_GENERIC: it cannot.
__host__ __device__ inline uint8_t masked_kind(const DSeg& s, uint32_t block_frames, uint32_t level) {
  if (s.len == 0) return KIND_SILENT;
  if (!(s.pos >= 0.0 && s.pos < 2147483000.0)) return KIND_GENERIC;
  if ((uint32_t)s.dst_start + s.len > block_frames) return KIND_GENERIC;
  if (level >= 4u) {   // the everything family: any storage format at any speed the hot loop streams, as classify() names it
    if (!(s.speed > 0.0 && s.speed <= 4096.0) || !(s.pos + (double)s.len * s.speed < 2147483000.0)) return KIND_GENERIC;
    if (s.speed == 1.0) return s.format == FMT_F32 ? KIND_UNITY : s.format == FMT_I16 ? KIND_UNITY_I16 : KIND_UNITY_I32;
    if (s.speed > 0.999) return KIND_STRIDE;
    return s.format == FMT_I16 ? KIND_WINDOW_I16 : KIND_WINDOW;
  }
  if (s.format != FMT_F32) {
    if (level >= 3u && s.format == FMT_I16 && s.speed > 0.0 && s.speed <= 0.999) return KIND_WINDOW_I16;
    if (level < 2u || s.speed != 1.0) return KIND_GENERIC;
    return s.format == FMT_I16 ? KIND_UNITY_I16 : KIND_UNITY_I32;
  }
  if (s.speed == 1.0) return KIND_UNITY;
  if (s.speed > 0.0 && s.speed <= 0.999) return KIND_WINDOW;
  return KIND_GENERIC;
}

// One track, one block: Track::process minus the per-sample work (track.cpp:587-736).
// `dry`: only the track's state moves on — nothing is allocated or stored (plan_segment's run-up to its first block)
__host__ __device__ inline void plan_track_block(const PlanArgs& a, uint32_t t, uint32_t b, DTrackState* st,
                                                 DClip* clips, uint32_t num_clips, TrackCache* cache,
                                                 const DBlockTime& bt, float gl, float gr, bool dry = false) {
  // the record is assembled in registers and written to HBM once, as four 16-B stores: building it in place
  // would turn every field update into a global store followed (in classify) by a dependent global load
  DTrackBlock rec;
  DTrackBlock* tb = &rec;
  rec.src[0] = nullptr;
  rec.src[1] = nullptr;
  rec.pos = 0.0;
  rec.speed = 0.0;
  rec.gain = 0.0f;
  rec.dst_start = 0;
  rec.req_len = 0;
  rec.format = 0;
  rec.flags = 0;
  rec.sample = 0;
  BlockWalker w;
  w.st = st;
  w.samples = a.samples;
  w.cache = cache;
  w.tb = tb;
  w.pool = a.pool;
  w.pool_count = a.pool_count;
  w.pool_chunks = a.pool_chunks;
  w.status = dry ? nullptr : a.status;
  w.dry = dry;
  w.n_samples = a.block_frames;
  w.n_channels = a.channels;
  w.dst_rate = a.sample_rate;
  w.start_sample = 0;
  w.nseg = 0;
  w.chunk = 0xFFFFFFFFu;
  tb->extra = 0;
  tb->len = 0;
  if (a.playing) {
    // Steady state — a clip that started in an earlier block plays through this whole block: process_event
    // (track.cpp:258-451) would walk to `max_time <= end_time` being false, emit no event and leave every
    // field as it is, so only the tail of the segment loop (track.cpp:710-722) remains.
    const bool steady = st->cur_type == EV_PLAY && st->has_clip_idx && !st->refresh_voice && st->partially_ended &&
                        st->clip_idx < num_clips && cache->clip_idx == st->clip_idx &&
                        !cache->clip.internal_state_changed && cache->clip.min_time < bt.start_time &&
                        cache->clip.max_time > bt.end_time;
    if (!steady)
      process_event(w, clips, num_clips, bt.start_time, bt.end_time, bt.sample_position, bt.beat_duration,
                    a.sample_rate, a.block_frames, a.flags_left);
    w.finish();
  }
  if (dry) return;
  tb->g[0] = gl;
  tb->g[1] = gr;
  tb->nseg = (uint8_t)w.nseg;
  tb->_pad = 0;
  tb->kind = classify(*tb, a.block_frames);
  // A clip start / end inside the block as the hot loop takes it (PlanArgs::masked_rows): one partial fp32 stream
  // call becomes a masked KIND_UNITY / KIND_WINDOW record, two calls that do not overlap become a ROW_PAIR.
  bool pair = false;
  uint8_t kind1 = KIND_SILENT;
  if (a.masked_rows && tb->kind == KIND_GENERIC && w.nseg <= 2u) {
    const uint8_t kind0 = masked_kind(get_seg0(rec), a.block_frames, a.masked_rows);
    if (w.nseg == 1u) {
      if (kind0 != KIND_GENERIC) tb->kind = kind0 == KIND_SILENT ? kind0 : (uint8_t)(kind0 | KIND_PARTIAL);
    } else {
      kind1 = masked_kind(w.seg1, a.block_frames, a.masked_rows);
      if (kind0 != KIND_GENERIC && kind1 != KIND_GENERIC && (uint32_t)rec.dst_start + rec.len <= w.seg1.dst_start) {
        pair = true;
        tb->kind = kind0 == KIND_SILENT ? kind0 : (uint8_t)(kind0 | KIND_PARTIAL);
        if (kind1 != KIND_SILENT) kind1 |= KIND_PARTIAL;
      }
    }
  }
  if (w.nseg == 2u && !pair) (void)w.alloc_chunk();   // the pre-render pass and the read-back find call 1 in the pool
  uint32_t tmpl_index = 0xFFFFFFFFu;
  {
    const uint4* srcq = reinterpret_cast<const uint4*>(&rec);
    DRow row;
    row.pos = rec.pos;
    row.tmpl = 0xFFFFFFFFu;
    row.flags = ROW_SILENT;
    if (pair) {
      const uint32_t ti = alloc_template(a, cache, 2u);
      if (ti != 0xFFFFFFFFu) {
        DTrackBlock rec1;
        set_seg0(&rec1, w.seg1);
        rec1.g[0] = gl;
        rec1.g[1] = gr;
        rec1.nseg = 1;
        rec1.kind = kind1;
        rec1._pad = 0;
        rec1.extra = 0;
        const uint4* srcq1 = reinterpret_cast<const uint4*>(&rec1);
        uint4* dstq = reinterpret_cast<uint4*>(&a.tmpl[ti]);
        dstq[0] = srcq[0];
        dstq[1] = srcq[1];
        dstq[2] = srcq[2];
        dstq[3] = srcq[3];
        dstq[4] = srcq1[0];
        dstq[5] = srcq1[1];
        dstq[6] = srcq1[2];
        dstq[7] = srcq1[3];
        row.tmpl = ti;
        row.flags = ROW_PAIR | ((rec.kind == KIND_SILENT && kind1 == KIND_SILENT) ? ROW_SILENT : 0u);
      }
    } else if (rec.nseg != 0) {   // also for calls that render nothing (finished clip): the plan keeps every stream call
      // (gains, format, source pointers and flags follow from the sample and from the render: the same for the run)
      const bool finished = rec.nseg == 1 && rec.len == 0 && rec.flags == SEG_FINISHED;
      const uint32_t shape = ((uint32_t)rec.dst_start << 16) | rec.req_len;
      const bool same = finished && cache->fin_tmpl != 0xFFFFFFFFu && cache->fin_pos == rec.pos && cache->fin_speed == rec.speed &&
                        cache->fin_gain == rec.gain && cache->fin_sample == rec.sample && cache->fin_shape == shape;
      if (same) {          // the same finished call as in the block before: no new template
        row.tmpl = cache->fin_tmpl;
      } else {
        const uint32_t ti = alloc_template(a, cache);
        if (ti != 0xFFFFFFFFu) {
          uint4* dstq = reinterpret_cast<uint4*>(&a.tmpl[ti]);
          dstq[0] = srcq[0];
          dstq[1] = srcq[1];
          dstq[2] = srcq[2];
          dstq[3] = srcq[3];
          row.tmpl = ti;
          row.flags = rec.kind == KIND_SILENT ? ROW_SILENT : 0u;
          if (finished) {
            cache->fin_pos = rec.pos;
            cache->fin_speed = rec.speed;
            cache->fin_gain = rec.gain;
            cache->fin_sample = rec.sample;
            cache->fin_shape = shape;
            cache->fin_tmpl = ti;
          }
        }
      }
    }
    *reinterpret_cast<uint4*>(&a.rows[(size_t)b * a.n_tracks + t]) = *reinterpret_cast<const uint4*>(&row);
    tmpl_index = row.tmpl;
  }
  if (tb->kind == KIND_GENERIC && tmpl_index != 0xFFFFFFFFu) {   // queue it for the pre-render pass
    uint32_t row;
#if defined(__HIP_DEVICE_COMPILE__)
    row = atomicAdd(a.gen_count, 1u);
#else
    row = (*a.gen_count)++;
#endif
    if (row < a.gen_cap)
      a.gen_list[row] = tmpl_index;
    else
      raise_status(a.status, 8u);
  }
}

// A run of steady-state blocks: the clip that is playing covers each of them completely and is far from
// its tail.  For such a block Track::process_event emits nothing and changes nothing, and Sampler::stream
// (sampler.cpp:99-104,209) reduces to  n = num_samples;  sample_offset_ += (double)num_samples * speed.
// Everything but the position is loop-invariant, so the run is planned by a short loop: two beat
// comparisons, the tail guard, one fp64 add and the 64-B store per block.  The records are field-for-field
// what plan_track_block produces (the parity tests compare them with the oracle's call log).  Returns the
// number of blocks planned (0: the general path must handle block b).
// `end`: plan no further than block end - 1 (the render's length, or a segment's seam); `dry`: as in plan_track_block.
// TV: the per-block transport records — a plain array, or TimesWindow (a window of them in LDS, the rest in device memory).
template <class TV>
__host__ __device__ inline uint32_t plan_steady_run(const PlanArgs& a, uint32_t t, uint32_t b, DTrackState* st,
                                                    uint32_t num_clips, TrackCache* cache,
                                                    const TV& times, float gl, float gr, uint32_t end, bool dry = false) {
  if (!(a.playing && st->cur_type == EV_PLAY && st->has_clip_idx && !st->refresh_voice && st->partially_ended &&
        st->clip_idx < num_clips && cache->clip_idx == st->clip_idx && !cache->clip.internal_state_changed &&
        cache->smp_idx == st->cur_sample))
    return 0;
  const DSample& smp = cache->smp;
  const double cnt = (double)smp.count, sp = st->playback_speed;
  if (!(sp > 0.0)) return 0;
  const uint32_t F = a.block_frames;
  const double step = (double)F * sp;                      // (double)num_samples * playback_speed_, sampler.cpp:103
  const double guard = (double)(F + 1u) * sp + 1.0;        // see BlockWalker::stream
  const double qmax = sp * 4294967040.0;
  const double min_time = cache->clip.min_time, max_time = cache->clip.max_time;
  DTrackBlock rec;
  rec.src[0] = smp.ch[0];
  rec.src[1] = smp.ch[a.channels > 1 ? 1 : 0];
  rec.pos = 0.0;
  rec.speed = sp;
  rec.gain = st->cur_gain;
  rec.g[0] = gl;
  rec.g[1] = gr;
  rec.nseg = 1;
  rec.kind = KIND_SILENT;
  rec.dst_start = 0;
  rec.len = (uint16_t)F;
  rec.req_len = (uint16_t)F;
  rec.format = (uint8_t)smp.format;
  rec.flags = 0;
  rec._pad = 0;
  rec.sample = st->cur_sample;
  rec.extra = 0;
  // How many consecutive blocks lie strictly inside the clip?  The block windows are computed by repeated
  // fp64 addition of a positive step (engine.cpp:1582,1621), so start_time and end_time are non-decreasing
  // in the block index: min_time < start_time only needs checking at the first block, and the last block
  // with max_time > end_time is found by bisection — no per-block LDS read on the critical path below.
  const DBlockTime t_b = times[b];
  if (!(min_time < t_b.start_time)) return 0;
  // n_time = the first i with !(max_time > times[b + i].end_time).  The windows advance by one (rounded) block length per
  // block, so the answer is known to within a block or two from ONE record: estimate, then let the table itself decide —
  // the same predicate on the same records as a bisection, a couple of look-ups instead of eleven (a session cut into
  // clips searches once per clip, and each look-up is a memory round trip for the lane).
  uint32_t lo = 0, hi = end - b;
  {
    const double dt = t_b.end_time - t_b.start_time;
    const double est = (max_time - t_b.end_time) / dt;
    if (dt > 0.0 && est >= 0.0 && est < (double)hi) {
      uint32_t i = (uint32_t)est;
      uint32_t steps = 0;
      while (i < hi && max_time > times[b + i].end_time && steps < 8u) {
        i++;
        steps++;
      }
      while (i > 0u && !(max_time > times[b + i - 1u].end_time) && steps < 8u) {
        i--;
        steps++;
      }
      if (steps < 8u) {   // (i is the answer: every block below it passes, block i does not — or i == hi)
        lo = hi = i;
      }
    } else if (dt > 0.0 && est >= (double)hi && max_time > times[b + hi - 1u].end_time) {
      lo = hi;            // the clip outlasts the render
    } else if (dt > 0.0 && est < 0.0 && !(max_time > t_b.end_time)) {
      hi = 0u;            // it ends inside this very block
    }
  }
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (max_time > times[b + mid].end_time)
      lo = mid + 1;
    else
      hi = mid;
  }
  const uint32_t n_time = lo;
  double off = st->sample_offset;
  // the shape of every record of the run (classify only looks at the position through its range check)
  if (n_time == 0 || !(off + guard <= cnt && (cnt - off) < qmax)) return 0;
  rec.pos = off;
  rec.kind = classify(rec, F);
  if (rec.kind == KIND_GENERIC || rec.kind == KIND_SILENT) return 0;   // general path (pre-render queue)
  if (dry) {   // the positions of the run, nothing else: the same additions in the same order
    uint32_t n = 0;
    while (n < n_time) {
      if (!(off + guard <= cnt && (cnt - off) < qmax && off < 2147483000.0)) break;
      off = off + step;                                    // sampler.cpp:209
      n++;
    }
    st->sample_offset = off;
    return n;
  }
  const uint32_t ti = alloc_template(a, cache);
  if (ti == 0xFFFFFFFFu) return 0;
  {
    const uint4* srcq = reinterpret_cast<const uint4*>(&rec);
    uint4* dstq = reinterpret_cast<uint4*>(&a.tmpl[ti]);
    dstq[0] = srcq[0];
    dstq[1] = srcq[1];
    dstq[2] = srcq[2];
    dstq[3] = srcq[3];
  }
  DRow row;
  row.tmpl = ti;
  row.flags = ROW_POS;
  uint32_t n = 0;
  {
    // Most of the run is far from the clip tail.  The tail / position guards below hold for block i as long as
    // off_i <= lim; off_i grows by one rounded addition per block, so off_i <= (off_0 + i*step) + i*ulp and a
    // whole frame of slack covers the accumulated rounding of any batch (i <= 2048, values < 2^31).  Those
    // blocks need no per-block test: the loop carries nothing but the fp64 addition.
    const double lim = (cnt - guard < 2147482999.0 ? cnt - guard : 2147482999.0) - 1.0;
    uint32_t n_safe = 0;
    if (off <= lim) {
      const double q = (lim - off) / step;
      n_safe = q >= 4096.0 ? 4096u : (uint32_t)q;
    }
    if (n_safe > n_time) n_safe = n_time;
    DRow* dst = &a.rows[(size_t)b * a.n_tracks + t];
    for (; n < n_safe; n++) {
      row.pos = off;
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&row);
      dst += a.n_tracks;
      off = off + step;                                    // sampler.cpp:209
    }
  }
  while (n < n_time) {
    // near the clip tail the general path takes over (exact division); 2147483000 is classify's position bound
    if (!(off + guard <= cnt && (cnt - off) < qmax && off < 2147483000.0)) break;
    row.pos = off;
    *reinterpret_cast<uint4*>(&a.rows[(size_t)(b + n) * a.n_tracks + t]) = *reinterpret_cast<const uint4*>(&row);
    off = off + step;                                      // sampler.cpp:209
    n++;
  }
  st->sample_offset = off;
  return n;
}

// The transport of K consecutive blocks, exactly the arithmetic of Engine::process (engine.cpp:1578-1585 per block,
// :1619-1623 between blocks).
__host__ __device__ inline void block_times(const PlanArgs& a, DBlockTime* times) {
  double playhead = a.playhead, sample_position = a.sample_position;
  // the reference evaluates these three per block from the same operands: same bits every time, so once (a division per
  // block made this loop — one lane, K dependent steps — the longest part of a batch render's plan)
  const double buffer_duration = (double)a.block_frames / a.sample_rate;                                   // :1578
  const double buffer_duration_in_beats = buffer_duration / a.beat_duration;                              // :1581
  const double block_samples = beat_to_samples(buffer_duration_in_beats, a.sample_rate, a.beat_duration);   // :1620
  const bool playing = a.playing != 0u;
  for (uint32_t i = 0; i < a.n_blocks; i++) {
    const double next_playhead_pos = playhead + buffer_duration_in_beats;            // :1582
    times[i] = DBlockTime{playhead, next_playhead_pos, sample_position, a.beat_duration};
    if (playing) {
      sample_position += block_samples;                                              // :1620
      playhead = next_playhead_pos;                                                  // :1621
    }
  }
}

// The state a track enters a render with: what the previous render left, the pending patch applied, the playing clip's gain
// re-read after an edit.
__host__ __device__ inline DTrackState plan_initial_state(const PlanArgs& a, uint32_t t, const DClip* clips, uint32_t nc) {
  DTrackState st = a.state[t];
  if (a.patch) {
    const DPatch p = a.patch[t];
    if (p.flags & PATCH_CLIPIDX) {   // Track::reset_playback_state(time, false), track.cpp:220-232
      st.has_clip_idx = p.has_clip_idx;
      st.clip_idx = p.clip_idx;
      st.partially_ended = 0;
    }
    if (p.flags & PATCH_REFRESH) st.refresh_voice = p.refresh_voice;
    if (p.flags & PATCH_STOP) st.cur_type = EV_NONE;   // Track::stop, track.cpp:249-256
  }
  if (a.clips_changed && st.cur_type == EV_PLAY) {
    // the reference reads current_audio_event.clip->audio.gain at every stream call (track.cpp:676,716):
    // after an edit (set_clip_gain, re-sorted list) find the playing clip again by identity.  A clip an edit DESTROYED
    // while it was sounding (Track::update_clip_ordering, track.cpp:159-175) reads 0.0f from then on — Pool::free has
    // zeroed its chunk (core/memory.h:80-86): the sampler keeps advancing, the track is silent until its next event —
    // unless a newer clip of the track has taken the chunk over (the uid stands for the chunk: wbx_clip_edit.h ClipIds).
    float gain = 0.0f;
    for (uint32_t i = 0; i < nc; i++)
      if (clips[i].uid == st.cur_clip_uid) gain = clips[i].gain;
    st.cur_gain = gain;
  }
  return st;
}

// The records the track is about to use — the clip it stands on, the clip behind it, the sample that is playing — are
// fetched TOGETHER, as soon as the state names them: the sequencer is one lane per track, and finding them one after the
// other (state -> clip -> sample) is three dependent memory round trips in front of a one-block render.  (Filling the
// cache changes no result: every use of a cached record compares its index first.)
__host__ __device__ inline void plan_cache_init(const PlanArgs& a, uint32_t t, const DTrackState& st, const DClip* clips, uint32_t nc,
                                                TrackCache* cache) {
  cache->clip_idx = 0xFFFFFFFFu;
  cache->next_idx = 0xFFFFFFFFu;
  cache->smp_idx = 0xFFFFFFFFu;
  cache->fin_tmpl = 0xFFFFFFFFu;
  cache->tmpl_next = cache->tmpl_end = 0u;
  if (a.tmpl_reserve == 0u) {    // one-block renders: the track owns templates 2t and 2t + 1 (a block takes one, or a ROW_PAIR's two)
    cache->tmpl_next = 2u * t;   // — no atomic round trip in the callback's latency chain
    cache->tmpl_end = 2u * t + 2u;
  }
  if (st.has_clip_idx && st.clip_idx < nc) {
    cache->clip = clips[st.clip_idx];
    cache->clip_idx = st.clip_idx;
    if (st.clip_idx + 1u < nc) {
      cache->next = clips[st.clip_idx + 1u];
      cache->next_idx = st.clip_idx + 1u;
    }
  }
  if (st.cur_type == EV_PLAY) {
    cache->smp = a.samples[st.cur_sample];
    cache->smp_idx = st.cur_sample;
  }
}

// blocks [b, end) of one track: steady runs and general blocks in turn
template <class TV>
__host__ __device__ inline void plan_blocks(const PlanArgs& a, uint32_t t, DTrackState* st, DClip* clips, uint32_t nc, TrackCache* cache,
                                            const TV& times, float gl, float gr, uint32_t b, uint32_t end, bool dry) {
  while (b < end) {
    b += plan_steady_run(a, t, b, st, nc, cache, times, gl, gr, end, dry);   // tight loop over the common case
    if (b < end) {
      plan_track_block(a, t, b, st, clips, nc, cache, times[b], gl, gr, dry);
      b++;
    }
  }
}

// One track through the K blocks of a render: what a lane of plan_kernel does (and the host harness of the tests,
// track by track).  Applies the pending state patch, then alternates steady runs and general blocks.
template <class TV>
__host__ __device__ inline void plan_track(const PlanArgs& a, uint32_t t, const TV& times) {
  const uint32_t c0 = a.clip_first[t];
  const uint32_t nc = a.clip_first[t + 1] - c0;
  DClip* clips = const_cast<DClip*>(a.clips) + c0;
  DTrackState st = plan_initial_state(a, t, clips, nc);
  TrackCache cache;
  plan_cache_init(a, t, st, clips, nc, &cache);
  const float gl = a.gains[2 * t + 0], gr = a.gains[2 * t + 1];
  plan_blocks(a, t, &st, clips, nc, &cache, times, gl, gr, 0u, a.n_blocks, false);
  a.state[t] = st;
}

// ------------------------------------------------------------------------------------------------------------------------
// The sequencer of a long render, cut ALONG THE TIME AXIS (round 4).  One lane per track walks the K blocks of a render one
// after the other; a session cut into clips is a chain of dependent record look-ups per clip boundary, 4096 tracks are 64
// waves, and the plan of a 2048-block render of such a session took 5.4 ms beside a 6.5 ms mix (and 4.4 ms in front of a
// 2.6 ms mix of 128-frame blocks).  The state a track has at block b is a function of what happened before b — but nearly
// always of very little of it: a clip that STARTS FROM ITS BEGINNING (track.cpp:357-374) resets the sampler and every field
// of the event state that a later block reads.  So the render is cut into segments of L blocks, one lane per (track,
// segment); the lane of segment s > 0
//   1. finds the last clip start in front of its first block b0 (binary search over the track's clips, one look at the
//      transport records) — or, when no clip of the track started inside this render before b0, takes the track's real
//      entry state and block 0,
//   2. runs the very same sequencer code DRY from there to b0 (state only: no template, no row, no queue entry) — a few blocks
//      for a session cut into clips, a loop of fp64 additions for one long clip,
//   3. notes the state it arrived with (its GUESS for block b0), plans its own blocks for real, notes the state it ends with.
// plan_check_seams then walks the seams of a track in order: segment s stands if its guess equals, bit for bit, the state
// segment s - 1 ended with (segment 0 starts from the truth, so by induction every standing segment was planned from the
// state the one-lane walk would have had); at the first seam that differs — a clip the sequencer skipped, overlapping or
// out-of-order events, anything the shortcut does not foresee — the rest of the track is planned again in one walk from
// the true state (plan_redo_track), over the speculative rows.  Results are therefore the serial walk's whatever the guess was; only the time
// depends on it.  (The templates, pool chunks and queue entries of a replaced segment stay allocated and unused.)
// Not taken — the host decides, wbx_engine.hip — when a clip flag (internal_state_changed, cleared by the sequencer as it
// passes: track.cpp:373,392,418) may be set: a dry lane and a real one would race for it.
// ------------------------------------------------------------------------------------------------------------------------

// the transport records of a segment's neighbourhood in LDS (host: a plain array), all others in device memory
struct TimesWindow {
  const DBlockTime* all;   // [K]
  const DBlockTime* win;   // records w0 .. w1 - 1
  uint32_t w0, w1;
  __host__ __device__ DBlockTime operator[](uint32_t i) const {
    const DBlockTime* p = (i >= w0 && i < w1) ? win + (i - w0) : all + i;
    return *p;
  }
};

// the clip start the lane of a segment that begins at block b0 (> 0) runs up from: -> the block to start at; *st = the state
// to start with (the guess: "stopped in front of clip j", or the track's entry state at block 0)
template <class TV>
__host__ __device__ inline uint32_t plan_anchor(const PlanArgs& a, const DClip* clips, uint32_t nc, const TV& times, uint32_t b0,
                                                const DTrackState& entry, DTrackState* st) {
  *st = entry;
  if (nc == 0u || !a.playing) return 0u;
  const double T = times[b0].start_time;   // == end_time of block b0 - 1: a clip with min_time <= T has been started by then
  // j = the last clip with min_time <= T (the list is sorted by min_time; any list gives SOME j, and a wrong one is caught at the seam)
  uint32_t lo = 0u, hi = nc;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (clips[mid].min_time <= T)
      lo = mid + 1u;
    else
      hi = mid;
  }
  if (lo == 0u) return 0u;
  const uint32_t j = lo - 1u;
  const double mt = clips[j].min_time;
  const DBlockTime t0 = times[0];
  if (!(mt >= t0.start_time)) return 0u;   // it started before this render (or in the middle): the entry state decides
  // the first block whose window reaches min_time (min_time <= end_time): estimate from the first record, settle on the table
  const double dt = t0.end_time - t0.start_time;
  uint32_t i = 0u;
  if (dt > 0.0) {
    const double est = (mt - t0.start_time) / dt;
    i = est >= (double)b0 ? b0 - 1u : (uint32_t)est;
  }
  if (i >= b0) i = b0 - 1u;
  uint32_t steps = 0u;
  while (i + 1u < b0 && !(mt <= times[i].end_time) && steps < 64u) {
    i++;
    steps++;
  }
  while (i > 0u && mt <= times[i - 1u].end_time && steps < 64u) {
    i--;
    steps++;
  }
  if (steps >= 64u) return 0u;             // (a transport the estimate does not fit: walk from the entry state)
  DTrackState g{};
  g.has_clip_idx = 1u;
  g.clip_idx = j;
  g.cur_type = EV_STOP;
  *st = g;
  return i;
}

// lane (t, s) of the segmented plan; L = blocks per segment.  -> *guess (s > 0): the state it arrived at its first block
// with, *end: the state it left its last block with (the caller stores them where the seam check finds them: [N][S] each)
template <class TV>
__host__ __device__ inline void plan_segment(const PlanArgs& a, uint32_t t, uint32_t s, uint32_t L, uint32_t S, const TV& times,
                                             DTrackState* guess, DTrackState* end) {
  const uint32_t c0 = a.clip_first[t];
  const uint32_t nc = a.clip_first[t + 1] - c0;
  DClip* clips = const_cast<DClip*>(a.clips) + c0;
  const DTrackState entry = plan_initial_state(a, t, clips, nc);
  const uint32_t b0 = s * L, b1 = (b0 + L < a.n_blocks) ? b0 + L : a.n_blocks;
  DTrackState st = entry;
  uint32_t from = 0u;
  if (s > 0u) from = plan_anchor(a, clips, nc, times, b0, entry, &st);
  TrackCache cache;
  plan_cache_init(a, t, st, clips, nc, &cache);
  const float gl = a.gains[2 * t + 0], gr = a.gains[2 * t + 1];
  // (one loop for the run-up and the segment itself — one copy of the sequencer in the kernel — with a stop at the seam)
  uint32_t b = s > 0u ? from : b0;
  if (s > 0u && b >= b0) *guess = st;   // (no run-up at all: cannot happen — plan_anchor returns a block in front of b0)
  while (b < b1) {
    const bool dry = b < b0;
    const uint32_t stop = dry ? b0 : b1;
    b += plan_steady_run(a, t, b, &st, nc, &cache, times, gl, gr, stop, dry);
    if (b < stop) {
      plan_track_block(a, t, b, &st, clips, nc, &cache, times[b], gl, gr, dry);
      b++;
    }
    if (dry && b == b0) *guess = st;
  }
  *end = st;
}

__host__ __device__ inline bool same_state(const DTrackState& x, const DTrackState& y) {
  // (bit patterns: a NaN position must compare equal to itself, -0.0 differs from +0.0)
  const uint32_t* p = reinterpret_cast<const uint32_t*>(&x);
  const uint32_t* q = reinterpret_cast<const uint32_t*>(&y);
  bool same = true;
  for (uint32_t i = 0; i < sizeof(DTrackState) / 4u; i++) same = same && p[i] == q[i];
  return same;
}

// The seams of track t in order, once all its segments are planned (LD: how a stored state is read — on the device the
// states were written by other workgroups of the same launch).  -> S when every guess stood (the track's state for the next
// render is then in place), else the first segment whose guess did not: plan_redo_track plans the track again from there.
template <class LD>
__host__ __device__ inline uint32_t plan_check_seams(const PlanArgs& a, uint32_t t, uint32_t S, const DTrackState* guess,
                                                     const DTrackState* ends, const LD& ld) {
  uint32_t s = 1u;
  DTrackState prev = ld(&ends[(size_t)t * S]);
  while (s < S) {
    if (!same_state(ld(&guess[(size_t)t * S + s]), prev)) return s;
    prev = ld(&ends[(size_t)t * S + s]);
    s++;
  }
  a.state[t] = prev;
  return S;
}

// segments s .. S - 1 of track t again, in one walk from `from`, the state segment s - 1 really ended with
template <class TV>
__host__ __device__ inline void plan_redo_track(const PlanArgs& a, uint32_t t, uint32_t s, uint32_t L, const TV& times,
                                                const DTrackState& from) {
  const uint32_t c0 = a.clip_first[t];
  const uint32_t nc = a.clip_first[t + 1] - c0;
  DClip* clips = const_cast<DClip*>(a.clips) + c0;
  DTrackState st = from;
  TrackCache cache;
  plan_cache_init(a, t, st, clips, nc, &cache);
  const float gl = a.gains[2 * t + 0], gr = a.gains[2 * t + 1];
  plan_blocks(a, t, &st, clips, nc, &cache, times, gl, gr, s * L, a.n_blocks, false);
  a.state[t] = st;
}

// Host side: the per-(block, track) stream-call records of a finished plan, rebuilt from the 16-B rows, the templates
// they point at and the overflow pool, ordered by (block, track, call).  Rec has the fields of wbx_plan_record.
template <class Rec>
inline size_t plan_records(uint32_t K, uint32_t N, const DRow* rows, const DTrackBlock* tmpl, size_t n_tmpl, const DSeg* pool,
                           uint32_t pool_used, Rec* out, size_t cap) {
  size_t n = 0;
  for (uint32_t b = 0; b < K; b++)
    for (uint32_t t = 0; t < N; t++) {
      const DRow& row = rows[(size_t)b * N + t];
      if (row.tmpl >= n_tmpl) continue;   // no stream call at all in this track-block
      DTrackBlock r = tmpl[row.tmpl];
      if (row.flags & ROW_POS) r.pos = row.pos;
      const bool pair = (row.flags & ROW_PAIR) && (size_t)row.tmpl + 1 < n_tmpl;   // call 1 is the template behind
      const DSeg s1 = pair ? get_seg0(tmpl[row.tmpl + 1]) : DSeg{};
      for (uint32_t i = 0; i < r.nseg; i++) {
        const DSeg s0 = get_seg0(r);
        const DSeg* sg = (i == 0) ? &s0 : (pair && i == 1) ? &s1
                         : (r.extra < pool_used ? &pool[(size_t)r.extra * kChunk + (i - 1)] : nullptr);
        if (!sg) continue;
        if (out && n < cap) {
          Rec& o = out[n];
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
  return n;
}

}  // namespace wbx
