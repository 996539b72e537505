/*
 * wb_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the whitebox per-block mix hot path.
 *
 * This is a plain-C restatement of the reference's algorithm (native-m/whitebox @ 2025-07-25),
 * every function citing the reference file:line it follows.  It exists to CHECK the HIP product
 * path; it is never linked into, imported by, or called from the product (libwbx.so /
 * whitebox_amd).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - sample-rate arithmetic (Sampler::stream, panning law, dB->linear, beat<->sample, apply_gain,
 *     abs-max, AudioBuffer::mix/clear, f32->int conversion) is pinned bit-for-bit against the
 *     reference's OWN translation units compiled into oracle/_ref/libwbref.so (sampler.cpp,
 *     panning_law.cpp, audio_format_conv.cpp + header-only audio_buffer.h, dsp_ops.h, core_math.h)
 *     and against golden vectors generated from that build (tests/golden/).
 *   - the clip sequencer / block driver (Track::process_event, Track::process, Engine::process, find_next_clip,
 *     reset_playback_state, update_clip_ordering, add_audio_clip / add_to_cliplist / delete_clip / move_clip / set_clip_gain,
 *     delete_track / move_track / solo_track, set_bpm / set_playhead_position) is pinned bit-for-bit against the reference's
 *     OWN code since round 5: engine/track.cpp and engine/engine.cpp include core/debug.h (third-party spdlog, absent, and no
 *     stand-in is written), so oracle/Makefile cuts the regions of those files that hold no Log:: line out of them where they
 *     lie — at function boundaries, into build outputs — and oracle/ref_engine_driver.cpp compiles those texts unmodified
 *     into oracle/_ref/wbref_engine.  tests/test_ref_engine.py compares per block the master, transport, every track's
 *     AudioEvent list, sampler state and VU level, and the clip lists after edits (soak: 296 330 sessions / 4 570 895 blocks, 0
 *     divergences, profiles/r05_refseq_soak.txt); tests/golden/sequencer.npz carries its answers to 40 scripts everywhere.
 *     NOT in the cut (an unconditional Log:: line inside the function): Engine::reserve_track_region (overlap trimming:
 *     KAT-pinned, its arithmetic pinned through clip_edit.h), Engine::play / stop and Track::process_track_messages (a few
 *     statements each, restated by the driver and said so there).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math  (no FMA contraction: the reference build has
 * none, CMakeLists.txt has no -march/-ffast-math).
 */
#ifndef WB_ORACLE_H
#define WB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AudioFormat values, src/core/audio_format.h:7-20 */
enum {
  WBO_FMT_UNKNOWN = 0, WBO_FMT_I8 = 1, WBO_FMT_U8 = 2, WBO_FMT_I16 = 3, WBO_FMT_U16 = 4,
  WBO_FMT_I24 = 5, WBO_FMT_I24_X8 = 6, WBO_FMT_I32 = 7, WBO_FMT_U32 = 8, WBO_FMT_F32 = 9, WBO_FMT_F64 = 10
};

/* PanningLaw values, src/core/panning_law.h:5-11 */
enum { WBO_PAN_LINEAR = 0, WBO_PAN_BALANCED = 1, WBO_PAN_CP_3DB = 2, WBO_PAN_CP_4_5DB = 3, WBO_PAN_CP_6DB = 4 };

/* EventType, src/engine/event.h:11-15 */
enum { WBO_EV_NONE = 0, WBO_EV_STOP = 1, WBO_EV_PLAY = 2 };

/* ---- scalar helpers ------------------------------------------------------------------------ */
float wbo_db_to_linear(float db);                                  /* core_math.h:83-89 */
void wbo_pan_coefs(float p, int law, float* left, float* right);   /* panning_law.cpp:9-32 */
double wbo_beat_to_samples(double beat, double sample_rate, double beat_duration);   /* core_math.h:209-212 */
double wbo_samples_to_beat(double samples, double sample_rate, double beat_duration); /* core_math.h:204-207 */
double wbo_perf_update(double usage, double duration_ms, double target_ms);          /* core/timing.h:57-62 (engine.cpp:1653) */
double wbo_perf_get_usage(double usage);                                              /* core/timing.h:64-66 */
double wbo_buffer_duration_ms(uint32_t buffer_size, uint32_t sample_rate);            /* engine/audio_io.h:187-195 (engine.cpp:52) */

/* ---- buffers (planar fp32, AudioBuffer<float> semantics) ------------------------------------ */
void wbo_clear(float* const* ch, uint32_t n_channels, uint32_t n_samples);                  /* audio_buffer.h:67-71 */
void wbo_mix(float* const* dst, const float* const* src, uint32_t n_channels, uint32_t n);  /* audio_buffer.h:73-82 */
void wbo_apply_gain(float* buf, uint32_t count, float gain);                                /* dsp_ops.h:27-31 */
float wbo_abs_max(const float* buf, uint32_t count);        /* vu_meter.h:20-25 (== dsp_ops.h:10-19) */
void wbo_master_clamp(float* const* ch, uint32_t n_channels, uint32_t n_samples);           /* engine.cpp:1627-1636 */

/* ---- output format conversion, audio_format_conv.cpp:5-106 ------------- */
void wbo_f32_to_interleaved_i16(int16_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch);
void wbo_f32_to_interleaved_i24(uint8_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch);
void wbo_f32_to_interleaved_i24_x8(int32_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch);
void wbo_f32_to_interleaved_i32(int32_t* dst, const float* const* src, size_t off, size_t n, uint32_t nch);
void wbo_f32_to_interleaved_f32(float* dst, const float* const* src, size_t off, size_t n, uint32_t nch);

/* ---- Sample / Sampler, src/dsp/sample.h:18-28, src/dsp/sampler.h:13-36 ----------------------- */
typedef struct wbo_sample {
  int format;              /* WBO_FMT_* ; I24 is stored in int32 containers (sample.cpp:20) */
  uint32_t channels;
  uint32_t sample_rate;
  size_t count;            /* frames; each channel array must hold count + 16 readable frames (sample.h:19) */
  const void* const* data; /* data[channel] planar */
} wbo_sample;

typedef struct wbo_sampler {
  double playback_speed;
  double sample_offset;
} wbo_sampler;

void wbo_sampler_reset(wbo_sampler* s, double sample_offset, double speed, double src_rate, double dst_rate); /* sampler.h:18-27 */
void wbo_sampler_stream(wbo_sampler* s, const wbo_sample* smp, uint32_t num_channels, uint32_t num_samples,
                        uint32_t buffer_offset, float gain, float* const* dst);                              /* sampler.cpp:88-210 */

/* ---- clips / events / tracks / engine -------------------------------------------------------- */
typedef struct wbo_clip {     /* src/engine/clip.h:39-45,55-75 (audio fields only) */
  double min_time, max_time;  /* beats */
  double start_offset;        /* samples */
  double speed;               /* AudioClip::speed */
  float gain;                 /* AudioClip::gain */
  int sample;                 /* index into engine sample table */
  int internal_state_changed;
  int deleted;                /* Clip::deleted (clip.h:63), swept by update_clip_ordering */
  uint32_t uid;               /* identity = the pool chunk of the Clip object (the reference holds Clip* pointers across
                               * re-sorts); unique among live clips, re-used LIFO per track like Pool<Clip>'s chunks */
} wbo_clip;

typedef struct wbo_event {    /* src/engine/event.h:66-74 */
  int type;
  uint32_t buffer_offset;
  double time;
  double speed;
  uint64_t sample_offset;
  int clip;                   /* index into the track's clip list, -1 if none */
} wbo_event;

#define WBO_MAX_EVENTS 64
#define WBO_MAX_MSGS 64

typedef struct wbo_track {
  wbo_clip* clips; uint32_t n_clips, cap_clips;     /* sorted by min_time (track.cpp:176) */
  /* TrackEventState, track.h:36-44 */
  int has_clip_idx; uint32_t clip_idx; int refresh_voice; int partially_ended;
  wbo_event events[WBO_MAX_EVENTS]; uint32_t n_events;
  wbo_event current_event;                            /* track.h:112 */
  float cur_gain; int cur_sample;                     /* resolved from current_event.clip at event time */
  uint32_t cur_clip_uid;                              /* current_audio_event.clip (gain is read through it, track.cpp:676,716) */
  wbo_sampler sampler;
  /* TrackParameterState (audio side), track.h:46-53 */
  float volume, pan, pan_coeffs[2]; int mute;
  /* pending TrackMessage::ParamChange, track.cpp:47-79,773-779 */
  struct { uint32_t id; double value; } msgs[WBO_MAX_MSGS]; uint32_t n_msgs;
  float level[2];                                     /* VUMeter::level (max since last read) */
  float block_peak[2];                                /* max|m| of the last processed block */
  int bus;                                            /* extension A13: sub-bus id, -1 = none */
  int ui_solo;                                        /* ui_parameter_state.solo, track.h:52 (UI flag read by solo_track) */
  /* Track::clip_allocator (track.h:105, core/memory.h:40-86): a clip's uid stands for the pool chunk its Clip object lives
   * in.  Pool::free pushes the chunk on a LIFO free list and Pool::allocate pops it, so a clip created after another one
   * was destroyed takes over that one's chunk — and its identity for anything still pointing at the chunk. */
  uint32_t* free_uids; uint32_t n_free_uids, cap_free_uids;
} wbo_track;

typedef struct wbo_seglog {  /* one Sampler::stream call issued by Track::process (track.cpp:678,718) */
  double playback_speed;      /* Sampler::playback_speed_ */
  double sample_offset;       /* Sampler::sample_offset_ before the call */
  uint32_t track, dst_start, len;
  float gain;
  int sample;
} wbo_seglog;

typedef struct wbo_engine {
  wbo_track* tracks; uint32_t n_tracks, cap_tracks;
  wbo_sample* samples; uint32_t n_samples_tab, cap_samples;
  uint32_t out_channels, buffer_size, sample_rate;    /* set_audio_channel_config, engine.cpp:43-57 */
  double ppq;                                         /* engine.h:43 */
  double playhead, playhead_start, sample_position, beat_duration;
  int playing;
  uint32_t n_buses;                                   /* 0 = reference behaviour (no buses) */
  float* mixbuf[16];                                  /* Engine::mixing_buffer */
  float* busbuf;                                      /* [n_buses][C][F] scratch (extension) */
  wbo_seglog* seglog; uint32_t n_seglog, cap_seglog;  /* stream calls of the LAST processed block (if enabled) */
  int seglog_enabled;
  uint32_t next_clip_uid;
} wbo_engine;

wbo_engine* wbo_engine_create(uint32_t out_channels, uint32_t buffer_size, uint32_t sample_rate);
void wbo_engine_destroy(wbo_engine* e);
void wbo_engine_set_bpm(wbo_engine* e, double bpm);                /* engine.cpp:24-30 */
void wbo_engine_set_playhead(wbo_engine* e, double beat);          /* engine.cpp:32-41 */
void wbo_engine_set_buses(wbo_engine* e, uint32_t n_buses);        /* extension A13 */
int wbo_engine_add_sample(wbo_engine* e, int format, uint32_t channels, uint32_t sample_rate, size_t count,
                          const void* const* planar);             /* borrows the pointers */
int wbo_engine_add_track(wbo_engine* e);                          /* engine add_track + Track::Track() track.cpp:22-27 */
void wbo_track_set_volume(wbo_engine* e, int track, float db);    /* track.cpp:47-57 */
void wbo_track_set_pan(wbo_engine* e, int track, float pan);      /* track.cpp:59-68 */
void wbo_track_set_mute(wbo_engine* e, int track, int mute);      /* track.cpp:70-79 */
void wbo_track_set_bus(wbo_engine* e, int track, int bus);        /* extension A13 */
/* returns 0 ok; overlapping clips are trimmed / split / deleted by reserve_track_region (engine.cpp:478-569) */
int wbo_engine_add_audio_clip(wbo_engine* e, int track, double min_time, double max_time, double start_offset,
                              int sample, double speed, float gain); /* engine.cpp:293-309,409-461 */
/* ---- clip placement arithmetic and list edits -------------------------------- */
/* engine/clip_edit.h:10-16 */
void wbo_calc_move_clip(double clip_min, double clip_max, double relative_pos, double min_move, double* new_min, double* new_max);
/* engine/clip_edit.h:18-126.  sample_rate / sample_count describe the clip's asset. */
void wbo_calc_resize_clip(double clip_min, double clip_max, double clip_start_offset, double clip_speed,
                          double sample_rate, double sample_count, double relative_pos, double resize_limit,
                          double min_length, double min_resize_pos, double beat_duration, int is_min, int shift,
                          int stretch, int clamp_at_resize_pos, double* out_min, double* out_max,
                          double* out_start_offset, double* out_speed);
/* engine/clip_edit.h:128-137 (audio) */
double wbo_calc_clip_shift(double start_offset, double relative_pos, double beat_duration, double sample_rate);
/* engine/clip_edit.h:139-150 (audio) */
double wbo_shift_clip_content(double start_offset, double speed, double sample_rate, double relative_pos, double beat_duration);
/* wb::find_lower_bound with the sequencer's predicate (core/algorithm.h:24-40; track.cpp:126-127,206); n >= 1 */
uint32_t wbo_lower_bound_max_time(const double* max_times, uint32_t n, double value);
/* Track::query_clip_by_range, track.cpp:112-157: returns 1 and fills first/last when the range touches clips */
int wbo_track_query_clip_by_range(const wbo_engine* e, int track, double min, double max, uint32_t* first, uint32_t* last);
/* Engine::move_clip engine.cpp:346-363, resize_clip :365-398, delete_clip :400-407, set_clip_gain :1460-1464,
 * delete_region :463-475.  `clip` = index in the track's current (sorted) clip list. */
int wbo_engine_move_clip(wbo_engine* e, int track, uint32_t clip, double relative_pos);
int wbo_engine_resize_clip(wbo_engine* e, int track, uint32_t clip, double relative_pos, double resize_limit,
                           double min_length, int left_side, int shift, int stretch);
int wbo_engine_delete_clip(wbo_engine* e, int track, uint32_t clip);
int wbo_engine_set_clip_gain(wbo_engine* e, int track, uint32_t clip, float gain);
int wbo_engine_delete_region(wbo_engine* e, int track, double min, double max);
uint32_t wbo_track_clip_count(const wbo_engine* e, int track);
const wbo_clip* wbo_track_clip(const wbo_engine* e, int track, uint32_t i);

/* Engine::delete_track engine.cpp:210-218, move_track :228-243, solo_track :245-262 */
void wbo_engine_delete_track(wbo_engine* e, uint32_t slot);
void wbo_engine_move_track(wbo_engine* e, uint32_t from_slot, uint32_t to_slot);
void wbo_engine_solo_track(wbo_engine* e, uint32_t slot);

void wbo_engine_play(wbo_engine* e);                               /* engine.cpp:68-80 */
void wbo_engine_stop(wbo_engine* e);                               /* engine.cpp:82-93 */
void wbo_engine_enable_seglog(wbo_engine* e, int on);
/* one block; out[c] planar, out_channels x buffer_size.  engine.cpp:1576-1654
 * bus_out (optional, may be NULL): [n_buses][C][F] planar bus sums (extension A13). */
void wbo_engine_process(wbo_engine* e, float* const* out, float* bus_out);
/* same, with the final clamp optional (clamp = 0: the un-clamped sum, what one shard of a multi-GPU
 * session contributes before the cross-GPU reduce) */
void wbo_engine_process_ex(wbo_engine* e, float* const* out, float* bus_out, int clamp);
/* NOT in the reference: the same without clearing `out` first (engine.cpp:1598) — `out` holds the running un-clamped
 * sum of the tracks before this engine's; checker for wbx_set_master_init / WBX_DIST_CHAIN.  No sub-buses. */
void wbo_engine_process_from(wbo_engine* e, float* const* out, int clamp);

/* synthetic input generator (integer hash; input generation only, same bits as whitebox_amd/synth.py) */
void wbo_synth_f32(float* dst, size_t frames, uint64_t key, float amp, size_t pad);

/* exposed for KAT tests of the seek math */
void wbo_track_process_event(wbo_engine* e, wbo_track* t, double start_time, double end_time, double sample_position,
                             double beat_duration, double buffer_duration, double sample_rate, uint32_t buffer_size);
                                                                   /* track.cpp:258-451 (audio branch) */

/* ---- next rows: clip ingest and waveform mip-maps ---------------------------------
 * Clip ingest: dsp/sample.cpp as a whole needs libsndfile/vorbis/dr_mp3, absent from the image; its transposition,
 * deinterleave_samples<T> (:29-43), uses nothing of them but the NAME of libsndfile's count type.  oracle/Makefile cuts the
 * function out of the file where it lies and ref_deint_driver.cpp compiles the text unmodified inside a class template whose
 * parameter is called sf_count_t (instantiated at int64_t and int32_t) into _ref/libwbref_deint.so;
 * tests/test_oracle_vs_ref.py holds wbo_deinterleave to it bit for bit and tests/golden/ingest.npz carries its outputs.
 * load_file's own lines around it (sf_open / sf_readf_*, :112-197) cannot be built: the 1024-frame loop and the zeroed
 * 16-frame padding are restated from the text (KAT-pinned).
 * Mip-maps: gfx/waveform_visual.cpp as a whole needs spdlog + the renderer, but its summariser (:9-173) needs only the
 * reference's core/ headers: oracle/Makefile compiles that function from the file where it lies into
 * _ref/libwbref_mip.so (ref_mip_driver.cpp), tests/test_oracle_vs_ref.py holds wbo_mip_summarize to it bit for bit and
 * tests/golden/mip.npz carries its outputs.  The restatements below follow the cited lines. */

/* deinterleave_samples<T>, dsp/sample.cpp:29-43: dst[c][written + j] = src[channels*j + c]; returns written + n.
 * elem = bytes per sample (2: I16, 4: I32/F32). */
size_t wbo_deinterleave(void* const* dst, const void* src, size_t num_read, size_t written, int channels, size_t elem);

/* WaveformVisual::create, gfx/waveform_visual.cpp:181-246: the mip chain of one sample.
 * level l = 0,1,..: current_mip = 1 + 2l, chunk_count = 2^mip, block_count = 2^(mip-1),
 * mip_data_count = count/block_count rounded up to even; levels exist while count / 4^l > 64. */
uint32_t wbo_mip_levels(size_t count);
size_t wbo_mip_data_count(size_t count, uint32_t level);
/* summarize_for_mipmaps_impl<T>, gfx/waveform_visual.cpp:9-173, one channel, one level.
 * format: 3 = I16, 7 = I32 (also 24-bit in 32-bit containers), 9 = F32; out_bits: 8 (Low quality) or 16 (High).
 * out receives mip_data_count values (int8_t or int16_t). Out-of-range float->int conversions follow x86-64
 * (cvttss2si / cvttsd2si "integer indefinite", then truncation to T). */
void wbo_mip_summarize(int format, size_t count, const void* data, uint32_t level, int out_bits, void* out);

#ifdef __cplusplus
}
#endif
#endif
