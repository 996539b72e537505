// wbx_dev.h — data laid out in HBM, shared by the host runtime and the gfx950 kernels.
//
// Layout summary (DESIGN.md "Data layout in HBM"):
//   clip audio     per clip and channel one 256-B aligned planar array of `frames + 16` elements
//                  (16 zero frames of tail padding = Sample::sample_padding, reference src/dsp/sample.h:19)
//   DSample[]      sample/clip table: channel base pointers (mono wraps: both point at channel 0)
//   DClip[]        per-track clip lists, sorted by min_time (CSR: clip_first[t] .. clip_first[t+1])
//   DTrackState[]  the sequencer + sampler state a Track carries across blocks
//   DTrackBlock[]  [K][N] one 64-B record per (block, track): what to render — the "plan"
//   partial        [K][NG][C][F] fp32 group sums,  master [K][C][F],  bus [K][NB][C][F],  peaks [K][N][C]
#pragma once
#include <cstdlib>
#include <stdint.h>

namespace wbx {

constexpr uint32_t kPad = 16;          // Sample::sample_padding
constexpr uint32_t kMaxSegs = 16;      // Sampler::stream calls per (block, track)
constexpr uint32_t kChunk = kMaxSegs - 1;  // overflow-pool chunk: segments 1..15 of one track-block
#ifndef WBX_KSTAGE
#define WBX_KSTAGE 128
#endif
constexpr uint32_t kStage = WBX_KSTAGE;        // track-block records staged in LDS at a time

enum : uint32_t { FMT_I16 = 3, FMT_I24 = 5, FMT_I32 = 7, FMT_F32 = 9 };  // reference AudioFormat values

// kind of a track-block, decided when the plan is made (wave-uniform dispatch in the mix kernel)
enum : uint8_t {
  KIND_SILENT = 0,   // nothing to render
  KIND_UNITY = 1,    // one fp32 segment covering the whole block at playback_speed == 1.0 (sampler.cpp:145-156)
  KIND_WINDOW = 2,   // one fp32 (or 24/32-bit PCM: `format`) segment covering the whole block, 0 < playback_speed <= 0.999
                     // (linear, sampler.cpp:34-59)
                     // (with PlanArgs::masked_rows, fp32 KIND_UNITY / KIND_WINDOW records may cover only frames
                     //  [dst_start, dst_start + len) of the block: "masked rows")
  KIND_GENERIC = 3,  // anything else: several segments, partial coverage, playback speed above 4096
  KIND_UNITY_I16 = 4,  // one 16-bit PCM segment covering the whole block at playback_speed == 1.0 (sampler.cpp:109-120)
  KIND_UNITY_I32 = 5,  // the same for 24-bit (in 32-bit containers) and 32-bit PCM (sampler.cpp:121-144)
  KIND_STRIDE = 6,     // one fp32 segment covering the whole block, playback_speed > 0.999 and != 1 (linear, sampler.cpp:34-59):
                       // the taps of a lane's 4 frames no longer fit one 5-sample window, each frame loads its own pair;
                       // also resampled integer PCM above 0.999
  KIND_WINDOW_I16 = 7  // one 16-bit PCM segment covering the whole block, 0 < playback_speed <= 0.999 (linear): the 5-sample
                       // window of a lane's 4 frames is one 8-B and one 4-B load
};

// flag in DTrackBlock::kind: the record covers only frames [dst_start, dst_start + len) of the block (a masked row of the
// mix kernel's hot loop, PlanArgs::masked_rows); whole-block records — nearly all — are recognised without their bounds
constexpr uint8_t KIND_PARTIAL = 0x80;
constexpr uint8_t KIND_MASK = 0x7F;

enum : uint8_t {
  SEG_FINISHED = 1,  // Sampler::stream returned early: sample_offset_ >= count (sampler.cpp:99-100)
  SEG_CLIPPED = 2    // reference would have written past the block (uint32 wrap of event_length, track.cpp:669)
};

struct DSample {
  const void* ch[2];
  uint64_t count;
  uint32_t format, channels, sample_rate, _pad;
};

struct DClip {            // reference src/engine/clip.h:39-45,55-75 (audio fields)
  double min_time, max_time, start_offset, speed;
  float gain;
  uint32_t sample;
  uint32_t internal_state_changed;
  uint32_t uid;           // stable identity across re-sorts (the reference holds Clip* pointers)
};

struct DTrackState {      // reference TrackEventState track.h:36-44, current_audio_event track.h:112, Sampler sampler.h:13-16
  uint32_t has_clip_idx, clip_idx, refresh_voice, partially_ended;
  uint32_t cur_type;      // EventType of current_audio_event
  uint32_t cur_sample;
  float cur_gain;
  uint32_t cur_clip_uid;  // current_audio_event.clip: its gain is re-read every block (track.cpp:676,716)
  double playback_speed, sample_offset;
};

struct DPatch {           // host-side edits applied to DTrackState before the next plan
  uint32_t flags;         // PATCH_*
  uint32_t has_clip_idx, clip_idx;
  uint32_t refresh_voice;
};
enum : uint32_t { PATCH_CLIPIDX = 1, PATCH_REFRESH = 2, PATCH_STOP = 4 };

struct DSeg {             // one Sampler::stream call (48 B) — overflow-pool entry
  const void* src[2];
  double pos;             // Sampler::sample_offset_ before the call
  double speed;           // Sampler::playback_speed_
  float gain;             // AudioClip::gain
  uint16_t dst_start;
  uint16_t len;           // num_actual_samples (sampler.cpp:104)
  uint16_t req_len;       // num_samples as requested
  uint8_t format;
  uint8_t flags;
  uint32_t sample;
};

// One (block, track) record, 64 B = four 16-B quads laid out for the mix kernel's LDS reads:
//   Q0 src[0..1]   Q1 pos, speed   Q2 gain, g[0], g[1], nseg|kind|dst_start   Q3 the rest
// Segment 0 lives inline; segments 1..nseg-1 (rare: a clip boundary inside the block) in the pool.
struct DTrackBlock {
  const void* src[2];     // Q0
  double pos;             // Q1
  double speed;
  float gain;             // Q2
  float g[2];             // fl(volume * pan_coeffs[c]), 0 when muted (track.cpp:728-731)
  uint8_t nseg;           // stream calls (including zero-length / finished ones)
  uint8_t kind;
  uint16_t dst_start;
  uint16_t len;           // Q3
  uint16_t req_len;
  uint8_t format;
  uint8_t flags;
  uint16_t _pad;
  uint32_t sample;
  uint32_t extra;         // overflow chunk index (segments 1..nseg-1), valid when nseg > 1
};
static_assert(sizeof(DSeg) == 48, "DSeg must be 48 bytes");
static_assert(sizeof(DTrackBlock) == 64, "DTrackBlock must be 64 bytes");

// The plan a render hands from the sequencer to the mix kernel is two-level.  Per (block, track) there is one
// 16-B DRow; the 64-B DTrackBlock records it refers to ("templates") are shared: a clip playing through many
// blocks produces ONE template (everything but the position) and a row per block that carries the sampler
// position, while a block with events owns a complete template.  Rows of consecutive tracks are adjacent, so
// the sequencer's stores are coalesced and a steady track-block costs 16 B of plan traffic, not 64.
struct DRow {
  double pos;        // Sampler::sample_offset_ at the start of the block (valid with ROW_POS)
  uint32_t tmpl;     // index of the DTrackBlock template; unused for ROW_SILENT
  uint32_t flags;    // ROW_*
};
enum : uint32_t {
  ROW_SILENT = 1,    // nothing to render for this track in this block
  ROW_POS = 2,       // the template is shared by a run of blocks: take the position from the row
  ROW_PAIR = 4       // a clip boundary inside the block: TWO stream calls that do not overlap (one clip ends, the next
                     // starts), as two single-segment templates at tmpl and tmpl + 1 — the mix kernel renders both as
                     // masked rows in its hot loop (no pre-render pass, no overflow-pool entry)
};
static_assert(sizeof(DRow) == 16, "DRow must be 16 bytes");

struct DBlockTime {       // per-block transport scalars computed by the host exactly as engine.cpp:1578-1585
  double start_time, end_time, sample_position, beat_duration;
};

struct DGroup {           // tracks order[first .. first+count) are summed in order by one workgroup
  uint32_t first, count;
  int32_t bus;            // -1: straight into the master
  uint32_t flags;         // GROUP_*: its place in the member list (all direct tracks / the tracks of one bus) it is a piece of
};
// Chained renders (MixArgs::chain): the pieces of a member list are ONE sum — piece i starts from the running sum piece
// i-1 left (GROUP_CHAIN_IN) and hands its own on (GROUP_CHAIN_OUT); only a list's last piece holds a sum the sum kernel adds
enum : uint32_t { GROUP_CHAIN_IN = 1, GROUP_CHAIN_OUT = 2 };

struct PlanArgs {
  const DClip* clips;
  const uint32_t* clip_first;   // [N+1]
  const DSample* samples;
  DTrackState* state;           // [N]
  const DPatch* patch;          // [N] or null
  const float* gains;           // [N][2]
  DRow* rows;                   // [K][N]
  DTrackBlock* tmpl;            // [tmpl_cap] templates, allocated with tmpl_count
  uint32_t* tmpl_count;
  uint32_t tmpl_cap;
  DSeg* pool;                   // [pool_chunks][kChunk]
  uint32_t* pool_count;         // allocated chunks
  uint32_t* status;             // bit0: pool overflow, bit1: > kMaxSegs calls, bit2: SEG_CLIPPED happened,
                                // bit3: more generic track-blocks than pre-render rows, bit4: out of templates
  uint32_t* gen_list;           // [gen_cap] template index of every KIND_GENERIC record
  uint32_t* gen_count;
  uint32_t gen_cap;
  uint32_t pool_chunks;
  uint32_t n_tracks, n_blocks, block_frames, channels;
  double sample_rate;
  double playhead, sample_position, beat_duration;   // transport at the first block (engine.h:44-46)
  DBlockTime* times;            // batch renders: the K per-block transport records in device memory (the host computes them,
                                // a copy in front of the plan); null: every workgroup computes them into its LDS (short renders: one launch less)
  uint32_t playing;
  uint32_t clips_changed;       // the clip lists were edited since the previous plan: re-read the current clip's gain
  uint32_t masked_rows;         // the mix instance of this render takes partial-coverage rows (one segment, or a
                                // ROW_PAIR) in its hot loop: do not queue them for the pre-render pass.  1: fp32 rows
                                // (unity / window); 2: also integer PCM at unity speed (masked_kind, wbx_seq.h)
  uint32_t tmpl_reserve;        // templates a track reserves per atomic (8 for batch renders, 1 for one-block renders); 0: none —
                                // track t owns templates 2t, 2t + 1 (the one-launch callback; tmpl_cap >= 2 * n_tracks)
  uint32_t* flags_left;         // (optional, host memory) how many clips of the table still carry internal_state_changed: the
                                // sequencer counts down as it clears them (track.cpp:373,392,418)
  uint32_t lanes;               // tracks per wave of plan_kernel (64, or fewer for sessions cut into many clips: a wave
                                // executes every branch any of its tracks takes, so its time is set by the number of
                                // clip boundaries in the wave — fewer tracks per wave, more waves side by side)
};

struct SegArgs {                // the sequencer cut along the time axis (wbx_seq.h plan_segment / plan_check_seams)
  DTrackState* guess;           // [N][n_segs] the state a segment's lane arrived at its first block with
  DTrackState* ends;            // [N][n_segs] ... and left its last block with
  uint32_t* stats;              // [2] tracks with a seam that did not hold, segments planned again (running totals; may be null)
  uint32_t* ticket;             // [N] segments of the track that are planned (the lane that completes it checks the seams and
                                //     resets it)
  uint32_t seg_len, n_segs;     // blocks per segment, segments per render (n_segs * seg_len >= n_blocks)
};

struct GenArgs {                // pre-render of KIND_GENERIC track-blocks into scratch rows
  DTrackBlock* tmpl;            // templates; the queued ones are rewritten in place to KIND_UNITY reads of their row
  const DSeg* pool;
  const uint32_t* gen_list;
  const uint32_t* gen_count;
  float* rows;                  // [gen_cap][C][F + 8]
  DTrackBlock* saved;           // [gen_cap] the original records (wbx_engine_fetch_plan)
  uint32_t gen_cap, block_frames, channels;
};

struct MixArgs {
  const DRow* rows;             // [K][N]
  const DTrackBlock* tmpl;      // templates
  const float* zero_page;       // >= F + 8 zero floats: what padding / silent records read
  const DSeg* pool;
  const uint32_t* order;        // [N] track permutation (routing order)
  const DGroup* groups;         // [NG]
  float* partial;               // [K][NG][C][F]
  const float* init;            // [K][C][F] or null: what the first group's sum starts from (wbx_set_master_init)
  float* peaks;                 // [K][N][C]
  uint32_t* levels;             // [N][C] running maxima (VUMeter::level) as uint images, or null
  uint32_t n_tracks, n_groups, block_frames, channels;
  uint32_t tiles;               // ceil(C*lane_span / 256)
  uint32_t lane_span;           // lanes the instance gives one channel of one block: F/4 for a block of the instance's own
                                // size, the next such size above it otherwise (the lanes beyond F/4 clone the last four frames)
  uint32_t n_blocks;            // K (the sub-block instances of the mix kernel cover ceil(K/SB) workgroups per group)
  uint32_t masked_rows;         // rows may be ROW_PAIR / partial-coverage (PlanArgs::masked_rows of the same render)
  uint32_t* chain;              // chained render: [workgroup columns][n_groups] "this piece's running sum is out" words
                                // ((chain_epoch << 4) | XCC id + 1 once out: never cleared between renders); null: every
                                // group starts from zero (or MixArgs::init) and the sum kernel adds the group sums
  // One-block callbacks of sessions that are ONE group (up to 64 tracks, no sub-buses): the workgroup is the block's whole sum,
  // so it clamps and stores the master itself (and hands the plan status to the host) — no sum kernel, one launch fewer
  // in the latency path.  Null: the group sums go to `partial` for sum_kernel.
  float* fused_master;          // [K][C][F], usually pinned host memory
  uint32_t fused_clamp;
  uint32_t* fused_status_src;   // as SumArgs::status_src / status_dst / zero_status
  uint32_t* fused_status_dst;
  uint32_t fused_zero_status;
  uint32_t chain_epoch;         // this render's tag, 1 .. 2^28-1 (words are zeroed on allocation and when the tag wraps)
  uint32_t* chain_status;       // ... bit 5 of this word is set when a wait for a predecessor gave up (never, unless the
                                // device's in-order workgroup dispatch is not what it is documented to be)
  uint32_t partial_through;     // the one-launch callback: the group sum is written THROUGH to memory (agent-scope stores) — a
                                // workgroup behind another XCD's L2 adds the group sums in this same launch
  uint32_t* chain_sticky;       // ... and of this one, which belongs to the context and is cleared only when a host call reports it
  unsigned long long* dbg_clock;   // diagnostic (WBX_DBG_CLOCK=1): [workgroups][4] start / end wall-clock ticks, HW_ID, XCC_ID, or null
  double uniform_speed;         // > 0: every linearly resampled row of this render plays at exactly this speed, which lies
                                // in [0.67, 0.999] (one resampling ratio in the whole session); 0: no such promise
  int packed_x;                 // WBX_PACKED_X as the context read it at creation (-1: unset; packed_masked_variant)
  uint32_t fast_partial;        // partial stream calls that start >= 4 samples into their clip: the unmasked arithmetic + frame
                                // masks instead of the clamped per-frame form (wbx_mix.h fast_part; WBX_FAST_PARTIAL=0: off)
};

constexpr uint32_t kSumGridBlocks = 512;   // blocks in flight of one sum launch (x its tiles = waves)

struct SumArgs {
  const float* partial;         // [K][NG][C][F]
  const DGroup* groups;
  float* master;                // [K][C][F]
  void* out_il;                 // or: interleaved device-format samples [K*F][C] (packed 24-bit: [K][F*C*3] bytes), then
  uint32_t out_format;          // master is unused; WBX_OUT_* (3 i16, 5 packed i24, 6 i24 in 32, 7 i32, 9 f32)
  float* buses;                 // [K][NB][C][F] or null
  uint32_t n_groups, n_buses, block_frames, channels;
  uint32_t n_blocks;            // K (filled in by launch_sum; the grid's x covers min(K, kSumGridBlocks) and walks the rest)
  uint32_t clamp;
  uint32_t chain;               // chained render: groups flagged GROUP_CHAIN_OUT hold intermediate running sums — skip them
  uint32_t* status_src;         // optional: the plan's 4 counters, copied to status_dst (pinned host memory) so that
  uint32_t* status_dst;         // the one-block callback path learns the plan status without another launch
  uint32_t zero_status;         // ... and cleared for the next plan that uses this buffer (no memset launch per block),
                                // unless the pre-render queue is not empty (the host then repeats pre-render + mix)
};

// ---- waveform mip-maps (wbx_media.hip) ----
struct MipNode {
  int mn, mx;        // T-valued
  int ord;           // 1: maximum first
  int empty;
};

constexpr uint32_t kMipTile = 2048;       // samples per workgroup = the chunk of level 5
constexpr uint32_t kMipTileLevels = 6;    // levels 0..5 are complete inside a tile

struct MipArgs {
  const void* src;            // one channel of a clip
  uint64_t count;
  void* level_out[24];        // this channel's output row of every level
  uint64_t data_count[24];    // mip_data_count per level
  uint32_t n_levels;
  MipNode* tile_nodes;        // [tiles + tiles/4 + 1] the level-5 node of every tile, then scratch of the upper levels
  uint32_t n_tiles;
};

// The lane space of the instance a block of F = 4 * S4 frames and C channels takes (MixArgs::lane_span): lanes per channel and
// block.  The instances are cut for blocks of 128 frames (stereo), 256, 512, 1024 ... — what the reference's settings dialog
// offers (ui/settings.cpp:22-24) — but the block a device back end really opens is its period, realigned to 32 frames
// (config.cpp:146-149,217-222): 480 frames for WASAPI's 10 ms at 48 kHz, 416 at 44.1 kHz, 960 for 20 ms.  Such a block takes
// the next shape above it; its surplus lanes clone the block's last four frames (wbx_mix.h).  WBX_RAGGED=0: the general
// instance of earlier rounds instead (A/B aid).
// (`ragged_off`: WBX_RAGGED=0 as the context read it when it was created — the audio callback never calls getenv, and the
//  plan-time and launch-time choices of one context cannot disagree)
inline uint32_t native_lane_span(uint32_t C, uint32_t S4, bool ragged_off) {
  const uint32_t lanes = C * S4;
  const bool exact = ((lanes % 256u == 0u) && (S4 % 64u == 0u)) || (C == 2u && S4 == 32u) ||
                     (S4 % 64u == 0u && (lanes == 128u || lanes == 64u));
  if (exact || ragged_off) return S4;
  if (C == 2u) return S4 <= 32u ? 32u : S4 <= 64u ? 64u : S4 <= 128u ? 128u : S4 <= 256u ? 256u : (S4 + 127u) / 128u * 128u;
  return S4 <= 64u ? 64u : S4 <= 128u ? 128u : S4 <= 256u ? 256u : (S4 + 255u) / 256u * 256u;
}

// Does a render of n_blocks short blocks (shorter than a 256-lane workgroup) of a session cut into clips take the PACKED
// masked-row instance (mix_kernel_x) instead of one block per workgroup?  Measured (tools/ab.py packed, profiles/r04_ab_packed.txt):
// 128-frame stereo (four blocks per workgroup) +5-9 % over the one-wave instance; 256-frame stereo / 512-frame mono (two blocks)
// 6-13 % BEHIND theirs — those keep one block per workgroup.  Renders of a few blocks: one workgroup per block is the shorter
// chain.  WBX_PACKED_X=0|1: A/B aid, tests (1 = every shape that has a packed instance) — `forced` is what the context read
// when it was created (-1: unset), handed to the launchers as MixArgs::packed_x.
inline int packed_masked_variant(uint32_t n_blocks, bool stereo128, int forced) {
  if (forced >= 0) return forced;
  return (stereo128 && n_blocks >= 8u) ? 1 : 0;
}

}  // namespace wbx
