// wbx_clip_edit.h — host-side clip placement arithmetic and clip-list edits of the engine adapter.
//
// Follows the reference's
//   calc_move_clip / calc_resize_clip / calc_clip_shift / shift_clip_content     src/engine/clip_edit.h:10-150
//   Track::query_clip_by_range                                                     src/engine/track.cpp:112-157
//   Track::update_clip_ordering                                                    src/engine/track.cpp:159-180
//   Engine::reserve_track_region                                                   src/engine/engine.cpp:478-569
// in fp64, operation for operation (this is the arithmetic that produces the
// min_time / max_time / start_offset / speed values the sequencer consumes).  UI-rate code: plain C++.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "wbx_dev.h"

namespace wbx {

struct HostClip {
  DClip d;
  bool deleted = false;      // Clip::deleted, swept by update_clip_ordering
  bool flag_dirty = false;   // internal_state_changed was set by an edit since the last upload
};

struct ClipQuery {
  uint32_t first, last;
};

namespace edit {

inline double samples_to_beat(double samples, double sample_rate, double beat_duration) {   // core_math.h:204-207
  double sec = samples / sample_rate;
  return sec / beat_duration;
}
inline double beat_to_samples_h(double beat, double sample_rate, double beat_duration) {     // core_math.h:209-212
  double sec = beat * beat_duration;
  return sec * sample_rate;
}
inline double maxd(double a, double b) { return b < a ? a : b; }   // math::max
inline double mind(double a, double b) { return a < b ? a : b; }   // math::min

// clip_edit.h:10-16
inline void calc_move_clip(double clip_min, double clip_max, double relative_pos, double min_move, double* new_min,
                           double* new_max) {
  const double new_pos = maxd(clip_min + relative_pos, min_move);
  *new_min = new_pos;
  *new_max = new_pos + (clip_max - clip_min);
}

struct ResizeResult {
  double min, max, start_offset, speed;
};

// clip_edit.h:18-126 for an audio clip whose asset has `sample_rate` / `sample_count`
inline ResizeResult calc_resize_clip(double clip_min, double clip_max, double clip_start_offset, double clip_speed,
                                     double sample_rate, double sample_count, double relative_pos, double resize_limit,
                                     double min_length, double min_resize_pos, double beat_duration, bool is_min,
                                     bool shift, bool stretch, bool clamp_at_resize_pos) {
  ResizeResult r{};
  r.speed = 1.0;
  if (!is_min) {   // right edge, :29-75
    const double right0 = clip_max;
    const double shortest = resize_limit + min_length - clip_min;
    double right = maxd(clip_max + relative_pos, 0.0);
    if (right - clip_min < shortest) right = clip_min + shortest;
    double so = clip_start_offset;
    if (shift) {
      so = samples_to_beat(so, sample_rate, beat_duration);
      if (right0 < right)
        so -= (right - right0) * clip_speed;
      else
        so += (right0 - right) * clip_speed;
      so = maxd(so, 0.0);
      so = mind(so, sample_count);
      so = beat_to_samples_h(so, sample_rate, beat_duration);
    }
    if (stretch) {
      const double span_samples = sample_count / clip_speed;
      const double delta_samples = beat_to_samples_h(relative_pos, sample_rate, beat_duration);
      r.speed = sample_count / (span_samples + delta_samples);
    }
    r.min = clip_min;
    r.max = right;
    r.start_offset = so;
    return r;
  }
  const double left0 = clip_min;   // left edge, :77-125
  const double shortest = clip_max - resize_limit + min_length;
  double left = maxd(clip_min + relative_pos, 0.0);
  if (clip_max - left < shortest) left = clip_max - shortest;
  if (clamp_at_resize_pos && left < min_resize_pos) left = min_resize_pos;
  double so = clip_start_offset;
  if (!shift) {
    so = samples_to_beat(so, sample_rate, beat_duration);
    if (left0 < left)
      so -= left0 - left;
    else
      so += left - left0;
    if (so < 0.0) left = left - so;
    so = maxd(so, 0.0);
    so = beat_to_samples_h(so, sample_rate, beat_duration);
  }
  if (stretch) {
    const double span_samples = sample_count / clip_speed;
    const double delta_samples = beat_to_samples_h(left0 - left, sample_rate, beat_duration);
    r.speed = sample_count / (span_samples + delta_samples);
  }
  r.min = left;
  r.max = clip_max;
  r.start_offset = so;
  return r;
}

// clip_edit.h:128-137 (audio)
inline double calc_clip_shift(double start_offset, double relative_pos, double beat_duration, double sample_rate) {
  const double offset_in_beat = samples_to_beat(start_offset, sample_rate, beat_duration);
  return beat_to_samples_h(maxd(offset_in_beat - relative_pos, 0.0), sample_rate, beat_duration);
}

// clip_edit.h:139-150 (audio)
inline double shift_clip_content(double start_offset, double speed, double sample_rate, double relative_pos,
                                 double beat_duration) {
  relative_pos *= speed;
  return calc_clip_shift(start_offset, relative_pos, beat_duration, sample_rate);
}

// find_lower_bound with `clip->max_time <= value` (core/algorithm.h:24-40)
inline uint32_t lower_bound_max(const std::vector<HostClip>& c, double value) {
  long long left = 0, right = (long long)c.size() - 1;
  while (left < right) {
    long long middle = (left + right) >> 1;
    if (c[(size_t)middle].d.max_time <= value)
      left = middle + 1;
    else
      right = middle;
  }
  return (uint32_t)right;
}

// Track::query_clip_by_range, track.cpp:112-157
inline bool query_clip_by_range(const std::vector<HostClip>& c, double min, double max, ClipQuery* q) {
  if (c.empty()) return false;
  if (max <= c.front().d.min_time) return false;
  if (min >= c.back().d.max_time) return false;
  const uint32_t first = lower_bound_max(c, min), last = lower_bound_max(c, max);
  if (first == last && (max <= c[first].d.min_time || min >= c[last].d.max_time)) return false;
  q->first = first;
  q->last = last;
  if (min > c[first].d.max_time) q->first++;
  if (!(max > c[last].d.min_time)) q->last--;
  return true;
}

// A clip's uid stands for the pool chunk its Clip object occupies in the reference (Track::clip_allocator, track.h:105):
// Pool::free pushes a destroyed clip's chunk on a LIFO free list, Pool::allocate pops it (core/memory.h:65-86), so the
// next clip created on the track takes over the chunk — and with it the identity of whatever still points there
// (current_audio_event.clip of a clip that was sounding when an edit destroyed it, track.cpp:676,716).
struct ClipIds {
  std::vector<uint32_t> free_list;   // chunks of destroyed clips, most recently freed last
  uint32_t* fresh;                   // the session's counter of never-used ids
  uint32_t take() {
    if (!free_list.empty()) {
      const uint32_t id = free_list.back();
      free_list.pop_back();
      return id;
    }
    return ++*fresh;
  }
};

// Track::update_clip_ordering, track.cpp:159-180: deleted clips are destroyed at once, in list order (:170-172)
inline void update_clip_ordering(std::vector<HostClip>& c, ClipIds& ids) {
  for (const HostClip& x : c)
    if (x.deleted) ids.free_list.push_back(x.d.uid);
  c.erase(std::remove_if(c.begin(), c.end(), [](const HostClip& x) { return x.deleted; }), c.end());
  const auto by_start = [](const HostClip& a, const HostClip& b) { return a.d.min_time < b.d.min_time; };
  // appending in timeline order is the common edit: a linear check instead of a sort of thousands of clips
  if (!std::is_sorted(c.begin(), c.end(), by_start)) std::sort(c.begin(), c.end(), by_start);
}

// Engine::reserve_track_region, engine.cpp:478-569.  `rate_of(sample)` gives the asset's sample rate;
// ignore_uid = 0 ignores nothing; `ids` names a clip created by a split (track->clip_allocator.allocate(), :503).
template <class RateOf>
inline void reserve_track_region(std::vector<HostClip>& c, uint32_t first_clip, uint32_t last_clip, double min, double max,
                                 uint32_t ignore_uid, double beat_duration, RateOf rate_of, ClipIds& ids) {
  if (c.empty()) return;
  auto shifted = [&](const HostClip& k, double rel) {
    return shift_clip_content(k.d.start_offset, k.d.speed, rate_of(k.d.sample), rel, beat_duration);
  };
  if (first_clip == last_clip) {   // :493-533
    if (c[first_clip].d.uid == ignore_uid) return;
    if (min > c[first_clip].d.min_time && max < c[first_clip].d.max_time) {   // the region splits the clip in two
      HostClip right = c[first_clip];   // Clip(const Clip&), clip.h:91-111: internal_state_changed keeps its default
      right.d.uid = ids.take();
      right.d.internal_state_changed = 0;
      right.flag_dirty = true;          // (a new object: no live device flag belongs to it, whatever its id named before)
      right.d.min_time = max;
      right.d.start_offset = shifted(right, c[first_clip].d.min_time - max);
      c[first_clip].d.max_time = min;
      c.push_back(right);
    } else if (min > c[first_clip].d.min_time) {
      c[first_clip].d.max_time = min;
    } else if (max < c[first_clip].d.max_time) {
      c[first_clip].d.start_offset = shifted(c[first_clip], c[first_clip].d.min_time - max);
      c[first_clip].d.min_time = max;
    } else {
      c[first_clip].deleted = true;
    }
    return;
  }
  if (c[first_clip].d.uid != ignore_uid && min > c[first_clip].d.min_time) {   // :535-568
    c[first_clip].d.max_time = min;
    first_clip++;
  }
  if (c[last_clip].d.uid != ignore_uid && max < c[last_clip].d.max_time) {
    c[last_clip].d.start_offset = shifted(c[last_clip], c[last_clip].d.min_time - max);
    c[last_clip].d.min_time = max;
    last_clip--;
  }
  if (first_clip <= last_clip && last_clip < c.size())
    for (uint32_t i = first_clip; i <= last_clip; i++)
      if (c[i].d.uid != ignore_uid) c[i].deleted = true;
}

}  // namespace edit
}  // namespace wbx
