/*
 * wbx.h — C ABI of the MI355X-native whitebox mix path (libwbx.so).
 *
 * Drop-in boundary for the reference's per-block multitrack mix
 *   Engine::process -> Track::process -> dsp::Sampler::stream / dsp::apply_gain /
 *   VUMeter::push_samples -> AudioBuffer::mix -> master clamp
 * (reference: src/engine/engine.cpp:1576-1654, src/engine/track.cpp:587-736,
 *  src/dsp/sampler.cpp:34-59,88-210, src/dsp/dsp_ops.h:27-31, src/engine/vu_meter.h:20-30,
 *  src/core/audio_buffer.h:73-82).
 *
 * Two layers, both plain C (pointers + sizes, no C++/torch types):
 *
 *   Layer 1  wbx_ctx      the device mix runtime.  The HOST keeps the reference's own clip
 *                         sequencer (Track::process_event) and hands the device, per block and
 *                         track, the Sampler::stream calls it would have made ("segments") plus
 *                         the block-rate gains; the device does every per-sample operation.
 *   Layer 2  wbx_engine   the whole path, sequencer included, resident on the device behind the
 *                         reference's Engine/Track surface (set_bpm, add_track, add_audio_clip,
 *                         Track::set_volume/pan/mute, play/stop, process).
 *
 * Conventions: every call returns wbx_status (0 = ok; negatives mirror the reference's
 * PluginResult, src/plughost/plugin_interface.h:24-29, plus device errors); host pointers are
 * caller-owned, device memory is ctx-owned; no callbacks into the host; nothing here falls back to the CPU
 * — without a gfx950 device wbx_create/wbx_engine_create return WBX_ERR_NO_DEVICE.
 * Threads: layer 1 (wbx_ctx) has one submitting thread per ctx.  Layer 2 (wbx_engine) has the reference's
 * contract: ONE audio thread in wbx_engine_process / wbx_engine_render, which hold the engine's editor lock for
 * the host side of the block (Engine::process, engine.cpp:1587-1651), and ONE UI thread for everything else —
 * its edits take the same lock, wbx_track_set_volume / _pan / _mute and wbx_engine_solo_track go through a
 * per-track single-producer ring of 64 messages without it (track.cpp:47-79, core/queue.h:142-196),
 * wbx_engine_set_bpm is an atomic store (engine.cpp:24-30).
 */
#ifndef WBX_H
#define WBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int wbx_status;
enum {
  WBX_OK = 0,
  WBX_ERR_FAILED = -1,        /* PluginResult::Failed        */
  WBX_ERR_UNIMPLEMENTED = -2, /* PluginResult::Unimplemented */
  WBX_ERR_UNSUPPORTED = -3,   /* PluginResult::Unsupported   */
  WBX_ERR_INVALID = -4,       /* bad argument                */
  WBX_ERR_NO_DEVICE = -5,     /* no HIP device / not gfx950  */
  WBX_ERR_DEVICE = -6,        /* HIP runtime error (see wbx_last_error) */
  WBX_ERR_OOM = -7,
  WBX_ERR_OVERFLOW = -8       /* more segments in one block than the plan can hold */
};

/* AudioFormat of clip storage — values of the reference enum, src/core/audio_format.h:7-20.
 * I24 clips are stored in 32-bit containers, as the reference does (src/dsp/sample.cpp:20). */
enum { WBX_FMT_I16 = 3, WBX_FMT_I24 = 5, WBX_FMT_I32 = 7, WBX_FMT_F32 = 9 };

/* Interleaved device-output formats for wbx_*_fetch_interleaved — reference converters
 * src/core/audio_format_conv.cpp:5-91 (values of the reference's AudioFormat enum).
 * WBX_OUT_I24 (packed 3-byte samples, audio_format_conv.cpp:22-43) mirrors the reference's bytes, quirk included:
 * its writer's destination index ignores the channel and the channel count, so every channel overwrites bytes
 * [0, 3*F) of a block's output and the LAST channel is what remains; the other 3*F*(C-1) bytes of the block's
 * 3*F*C-byte region are never written (wbx leaves them as the caller passed them).
 * Float -> integer conversions give the x86 results the reference build produces, also out of range (master
 * left un-clamped, NaN): cvttss2si / cvttsd2si return 0x80000000, the i16 / i24 paths then truncate. */
enum { WBX_OUT_I16 = 3, WBX_OUT_I24 = 5, WBX_OUT_I24_X8 = 6, WBX_OUT_I32 = 7, WBX_OUT_F32 = 9 };

typedef struct wbx_config {
  int32_t device;          /* HIP device ordinal */
  uint32_t max_tracks;     /* N upper bound */
  uint32_t max_blocks;     /* K upper bound per submit/render (1..4096) */
  uint32_t block_frames;   /* F, Engine::audio_buffer_size (reference default 512, src/config.cpp:146); multiple of 4 */
  uint32_t channels;       /* C, output channels: 1 or 2 (reference: 2, src/config.cpp:224) */
  uint32_t sample_rate;    /* destination rate, Engine::audio_sample_rate */
  uint32_t group_size;     /* tracks summed in index order by one workgroup; the master is the in-order sum of the group
                              sums, so group_size >= N reproduces the reference's strictly sequential order (engine.cpp:
                              1600-1617) bit for bit.  0 = the library picks: renders of >= 1024 blocks add ALL tracks of a
                              block (of a bus) in track order — the reference's order, bit-exact master, at the full rate:
                              128-track workgroups that continue each other's running sum, the blocks of the render supply
                              the parallelism (blocks shorter than 256 lanes count by the workgroup: 2048 blocks of 256
                              frames, 4096 of 128 — wbx_render_order tells); shorter renders take groups of 128 (when
                              max_blocks == 1, the audio-callback configuration: one group up to 16 tracks and one TRACK per group up to
                              64 — both the reference's order bit for bit — then 4 / 8 / 16 tracks above 64 / 256 / 512
                              tracks), within 1e-6 RMS of the reference's order at mix-bus levels (DESIGN.md "Summation
                              order" states the levels). */
  uint32_t max_segments;   /* extra (beyond one per track-block) segment slots per launch; 0 = default */
  void* stream;            /* hipStream_t to launch on, or NULL: the ctx creates its own */
} wbx_config;

typedef struct wbx_ctx wbx_ctx;
typedef struct wbx_engine wbx_engine;

/* One Sampler::stream call of one track in one block (dsp/sampler.h:29-35, sampler.cpp:88-210),
 * with the sampler state the host's sequencer holds at that point. */
typedef struct wbx_segment {
  double sample_offset;    /* Sampler::sample_offset_ before the call */
  double playback_speed;   /* Sampler::playback_speed_ ((src_rate/dst_rate)*clip speed, sampler.h:24) */
  uint32_t clip;           /* clip id given to wbx_clip_upload */
  uint32_t buffer_offset;  /* first destination frame in the block */
  uint32_t num_samples;    /* frames requested (the device applies the clip-tail limit of sampler.cpp:102-104) */
  float gain;              /* AudioClip::gain */
} wbx_segment;

/* ---- library ------------------------------------------------------------------------------- */
const char* wbx_version(void);
int wbx_device_count(void);                 /* gfx950 devices visible; 0 without a GPU (never an error) */
const char* wbx_status_string(wbx_status);

/* ---- layer 1: device mix runtime ------------------------------------------------------------
 * replaces the body of the per-track loop of Engine::process (engine.cpp:1600-1617) and the master
 * clamp (engine.cpp:1627-1636). */
wbx_status wbx_create(const wbx_config* cfg, wbx_ctx** out);
void wbx_destroy(wbx_ctx* ctx);
const char* wbx_last_error(const wbx_ctx* ctx);

/* Clip audio -> HBM (planar, +16 zero frames of tail padding like Sample::sample_padding,
 * src/dsp/sample.h:19, sample.cpp:127,140).  Replaces the storage half of `struct Sample`
 * (sample.h:18-28).  frames < 2^31-16. */
wbx_status wbx_clip_upload(wbx_ctx* ctx, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                           uint64_t frames, const void* const* planar);
/* Device-side synthetic clip (bench/test helper; integer-hash generator, DESIGN.md "Synthetic input"). */
wbx_status wbx_clip_synth(wbx_ctx* ctx, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                          uint64_t frames, uint64_t seed, uint32_t key_track, float amp);
wbx_status wbx_clip_free(wbx_ctx* ctx, uint32_t clip);
/* Clip storage in numbers.  Clip audio lives in slabs (64 MiB, 256 MiB, then 1 GiB each) carved up in order; the extent of
 * a freed or replaced clip is reused by the next clip that fits (first fit), a slab whose last clip went is empty again;
 * slabs go back to the driver with the context.  bytes_reserved: what the pool holds from the driver; bytes_live: what
 * its clips occupy.  Any pointer may be NULL. */
wbx_status wbx_clip_pool_stats(wbx_ctx* ctx, uint32_t* n_slabs, uint64_t* bytes_reserved, uint64_t* bytes_live);

/* Clip ingest: interleaved frames as a decoder delivers them (sf_readf_short/int/float, drmp3_read_pcm_frames_f32)
 * -> the same planar storage.  Replaces deinterleave_samples<T> + the allocation/padding of Sample::load_file
 * (src/dsp/sample.cpp:29-43, :127-142, :154-183); the transposition runs on the GPU.
 *   wbx_clip_upload_interleaved  `interleaved` is host memory ([frames][channels]); staged through pinned chunks
 *   wbx_clip_ingest_device       `interleaved` is device memory, 16-byte aligned (a GPU decoder / own staging) */
wbx_status wbx_clip_upload_interleaved(wbx_ctx* ctx, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                                       uint64_t frames, const void* interleaved);
wbx_status wbx_clip_ingest_device(wbx_ctx* ctx, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                                  uint64_t frames, const void* device_interleaved);
/* planar clip audio back to the host (tests; also Sample::get_read_pointer's role for host-side tools) */
wbx_status wbx_clip_download(wbx_ctx* ctx, uint32_t clip, uint32_t channel, void* dst);

/* Waveform peak mip-maps of a resident clip: WaveformVisual::create + summarize_for_mipmaps_impl<T>
 * (src/gfx/waveform_visual.cpp:9-246).  quality: 0 = Low (int8_t), 1 = High (int16_t) (waveform_visual.h:11-14).
 * Level l holds, per channel, mip_data_count(l) values = ordered (first, second) min/max pairs of chunks of
 * 2^(2l+1) samples; layout of a level [channels][mip_data_count] like the reference's upload buffer (:206-226).
 * Formats: I16, I32, F32 (the reference's switch has no other case). */
uint32_t wbx_mip_levels(uint64_t frames);                          /* number of levels (:195 while count/4^l > 64) */
uint64_t wbx_mip_data_count(uint64_t frames, uint32_t level);      /* :197-199 */
wbx_status wbx_clip_build_mipmaps(wbx_ctx* ctx, uint32_t clip, int quality);
wbx_status wbx_clip_fetch_mipmap(wbx_ctx* ctx, uint32_t clip, uint32_t level, void* dst);
/* device pointer + element count of a level (what WaveformMipmap{data, count} holds for the renderer) */
wbx_status wbx_clip_mipmap_device(wbx_ctx* ctx, uint32_t clip, uint32_t level, const void** data, uint64_t* count);

/* Track -> sub-bus routing (extension of the reference, which has no buses: SURVEY §8(a) A13).
 * track_bus[t] in [0, n_buses) or -1 (straight to master).  NULL / n_buses 0 = reference behaviour. */
wbx_status wbx_set_routing(wbx_ctx* ctx, uint32_t n_tracks, const int32_t* track_bus, uint32_t n_buses);

/* How a render of n_blocks blocks is summed, with the routing as it stands (after the last render / wbx_set_routing):
 * workgroup-level groups per block, the longest of them, and whether that IS the reference's order (every member list —
 * all tracks, or the tracks of each bus — added sequentially by one workgroup: engine.cpp:1600-1617, bit-exact master). */
wbx_status wbx_render_order(wbx_ctx* ctx, uint32_t n_blocks, uint32_t* n_groups, uint32_t* longest_group, int* reference_order);
/* PCI bus id ("0000:05:00.0") and marketing / arch name of the ctx's device (multi-GPU hosts log which rank got which). */
wbx_status wbx_device_info(wbx_ctx* ctx, char* pci_bus_id, size_t n_pci, char* name, size_t n_name);

/* Mix K blocks of n_tracks tracks.  segs[] holds the Sampler::stream calls grouped by (block, track):
 * those of (b, t) are segs[seg_offsets[b*n_tracks+t] .. seg_offsets[b*n_tracks+t+1]).  gains[(b*n_tracks+t)*2+c]
 * = fl(volume*pan_coeffs[c]) (0 when muted), the factor of track.cpp:728-731.  Asynchronous. */
wbx_status wbx_submit(wbx_ctx* ctx, uint32_t n_blocks, uint32_t n_tracks, const wbx_segment* segs,
                      const uint32_t* seg_offsets, const float* gains);

/* Results of the last submit (blocks until done).  Any pointer may be NULL.
 *  master_planar[c] : K*F floats, block after block (the AudioBuffer the audio thread hands to process)
 *  peaks            : [K][N][C]  max|x| per track/channel/block (VUMeter::push_samples, vu_meter.h:20-25); NaN samples
 *                     are ignored (the reference's running maximum restarts after a NaN, an order-dependent result)
 *  buses            : [K][n_buses][C][F] bus sums */
wbx_status wbx_fetch(wbx_ctx* ctx, float* const* master_planar, float* peaks, float* buses);
wbx_status wbx_fetch_interleaved(wbx_ctx* ctx, int out_format, void* dst);  /* K*F*C interleaved samples */
/* Let later submits / renders leave their master as interleaved samples of a device format (WBX_OUT_*; 0 = planar fp32
 * again): the conversion of core/audio_format_conv.cpp:5-91 becomes the epilogue of the sum kernel — no separate
 * conversion launch, no planar master in between.  The master target (the ctx's own buffer or wbx_set_master_target's,
 * then at least K*F*C samples of the format) holds [K*F][C] samples, packed 24-bit as the reference writes it (see
 * WBX_OUT_I24).  wbx_fetch_interleaved of the same format is then a plain copy; wbx_fetch's planar master is not
 * available.  Not with a multi-GPU exchange (partial masters stay planar fp32). */
wbx_status wbx_set_master_format(wbx_ctx* ctx, int out_format);
wbx_status wbx_sync(wbx_ctx* ctx);
/* Waits like wbx_sync and reports what no other call may have had the chance to: a render of >= 1024 blocks that adds in
 * the reference's order as chained 128-track pieces checks every hand-over between its workgroups, and a failed one makes
 * that render's master INVALID (WBX_ERR_DEVICE; later renders walk whole member lists instead).  wbx_fetch,
 * wbx_fetch_interleaved, wbx_engine_process, wbx_engine_fetch_plan and wbx_dist_sync return that status themselves; a
 * host that reads a caller-owned master target (wbx_set_master_target) directly — it never fetches — must call this
 * (or one of those) before it trusts the buffer.  The failure is latched: every render since the last report counts.
 * Replaces nothing in the reference (engine.cpp:1600-1617 is one thread, one order). */
wbx_status wbx_render_status(wbx_ctx* ctx);
/* Order `stream` (a hipStream_t; NULL = the ctx stream) after every kernel that writes the results of the last
 * submit / render (master, bus sums, peaks).  The sum of a render runs on a stream of its own beside the next mix:
 * work the CALLER enqueues that reads a caller-owned master target (wbx_set_master_target) — a collective, a copy,
 * an own kernel — must be ordered with this call first; wbx_fetch*, wbx_sync, wbx_partial_master, wbx_finalize_master*
 * and the wbx_dist_* calls do it themselves.  Device-side ordering only: the host does not block. */
wbx_status wbx_master_ready(wbx_ctx* ctx, void* stream);

/* Multi-GPU: the un-clamped partial master of the last submit, on the device, [K][C][F] fp32 ... */
wbx_status wbx_partial_master(wbx_ctx* ctx, void** device_ptr, size_t* n_floats);
/* ... and, on the root after the RCCL reduce, the clamp of engine.cpp:1627-1636 over a device buffer.
 * `stream`: hipStream_t to launch on (e.g. the stream that waited for the collective), NULL = the ctx stream. */
wbx_status wbx_finalize_master(wbx_ctx* ctx, void* device_partial, uint32_t n_blocks, int clamp, void* stream);
/* Out of place: clamp (or just move) the reduced master into `dst`, which may be device memory or pinned,
 * device-mapped host memory — the final master then leaves the GPU as the kernel's own stores. */
wbx_status wbx_finalize_master_into(wbx_ctx* ctx, const void* device_partial, void* dst, uint32_t n_blocks, int clamp,
                                    void* stream);
wbx_status wbx_set_clamp(wbx_ctx* ctx, int clamp_on_submit); /* 0: leave the master un-clamped (shard mode) */
/* Write the master of later submits/renders into a caller-owned DEVICE buffer of at least
 * max_blocks*C*F floats (e.g. the send buffer of the RCCL reduce); NULL restores the ctx-owned one. */
wbx_status wbx_set_master_target(wbx_ctx* ctx, void* device_buffer);
/* Start the master of later submits / renders from a running sum instead of the cleared buffer: `device_buffer`
 * ([max_blocks][C][F] floats, device memory or pinned device-mapped host memory, 16-byte aligned) holds the UN-clamped
 * master of the engine that owns the tracks BEFORE this one's in the session's track order; the first group in summation
 * order continues from it, so that a session split over several engines (several GPUs: WBX_DIST_CHAIN does exactly this
 * over RCCL; or one GPU, to go beyond max_tracks) is added in the reference's strictly sequential order
 * (engine.cpp:1600-1617) across the split.  NULL restores the cleared start.  Not with sub-buses. */
wbx_status wbx_set_master_init(wbx_ctx* ctx, const void* device_buffer);

/* ---- multi-GPU: tracks sharded over one process per GPU, ONE exchange per render ---------------
 * The reference has no distributed code; this is the path's only exchange step: every rank mixes its own contiguous
 * track range into an un-clamped partial master, the partials are summed onto rank 0 over RCCL (xGMI), and the clamp
 * of engine.cpp:1627-1636 runs there, after the sum.  Protocol per rank:
 *     wbx_shard_tracks(N, world, rank, &first, &count);          // which tracks this rank builds its session from
 *     rank 0: wbx_dist_new_id(&id) and hand the 128 bytes to the other ranks (file, socket, MPI, ...)
 *     wbx_dist_init(ctx, &id, rank, world, WBX_DIST_REDUCE);      // collective: returns when every rank has called
 *     loop: wbx_engine_render / wbx_submit;  wbx_dist_exchange(ctx, dst);      // dst: rank 0 only
 *     wbx_dist_sync(ctx);  ...  wbx_dist_shutdown(ctx);
 * After wbx_dist_init the ctx writes its (un-clamped) masters into an internal ring of three device buffers, so up
 * to two further renders may be issued while an exchange is in flight; nothing in the loop blocks the host.
 * WBX_DIST_REDUCE: one ncclReduce(sum) — RCCL's summation order is implementation-defined (inside the 1e-6 RMS budget).
 * WBX_DIST_ORDERED: ncclGather + a fixed-order add on the root, (((0 + p0) + p1) + ...) in rank = track order:
 * bit-reproducible on any topology (SURVEY §7 hard part 8).  Both add SHARD sums, which is not the reference's
 * track-after-track association: at 32768 tracks the difference leaves the 1e-6 RMS budget once the master runs at
 * mix-bus level (profiles/r03_level_probe.txt).
 * WBX_DIST_CHAIN: the reference's order across GPUs — rank g receives the running un-clamped master of rank g-1, its mix
 * continues that sum with its own tracks, rank g+1 gets the result; the LAST rank clamps and holds the result
 * (wbx_dist_result_rank; its wbx_dist_exchange takes `dst`).  With renders of >= 1024 blocks (whole-list walks) the
 * master is bit-identical to a single engine over all tracks.  The ranks form a pipeline over consecutive renders: same
 * throughput, a render's latency grows with the world size.  Not with sub-buses.
 * wbx_dist_init fails with WBX_ERR_FAILED when the collective start-up does not complete within
 * WBX_DIST_INIT_TIMEOUT_S seconds (environment, default 60, 0 = wait for ever). */
#define WBX_DIST_ID_BYTES 128
typedef struct wbx_dist_id {
  char bytes[WBX_DIST_ID_BYTES];
} wbx_dist_id;
enum { WBX_DIST_REDUCE = 0, WBX_DIST_ORDERED = 1, WBX_DIST_CHAIN = 2 };
void wbx_shard_tracks(uint32_t n_tracks, uint32_t world, uint32_t rank, uint32_t* first, uint32_t* count);
wbx_status wbx_dist_new_id(wbx_dist_id* out);
wbx_status wbx_dist_init(wbx_ctx* ctx, const wbx_dist_id* id, uint32_t rank, uint32_t world, int mode);
wbx_status wbx_dist_info(wbx_ctx* ctx, uint32_t* rank, uint32_t* world, int* mode);
wbx_status wbx_dist_result_rank(wbx_ctx* ctx, uint32_t* rank);   /* 0 (reduce / ordered) or world - 1 (chain) */
/* dst (the result rank only): [K][C][F] floats, device memory or pinned device-mapped host memory, 16-byte aligned */
wbx_status wbx_dist_exchange(wbx_ctx* ctx, void* dst);
/* average ms one exchange took on its own stream, over the exchanges completed so far (after wbx_dist_sync) */
wbx_status wbx_dist_exchange_time(wbx_ctx* ctx, double* ms_avg, uint64_t* n_exchanges);
/* every rank hands in `bytes` (<= 4096) bytes of host memory and gets all ranks' bytes in rank order; also a barrier */
wbx_status wbx_dist_allgather(wbx_ctx* ctx, const void* send, void* recv, size_t bytes);
wbx_status wbx_dist_sync(wbx_ctx* ctx);                  /* host waits for the renders and exchanges issued so far */
wbx_status wbx_dist_barrier(wbx_ctx* ctx);               /* all ranks */
wbx_status wbx_dist_max(wbx_ctx* ctx, double* value);    /* max over all ranks, in place (also a barrier) */
wbx_status wbx_dist_shutdown(wbx_ctx* ctx);

/* Render-ahead bound for asynchronous hosts: call after each submit / render; the host waits until the render issued
 * `max_ahead` (1..63) calls earlier has left the main stream.  Far deeper queues stall inside the HIP runtime. */
wbx_status wbx_pace(wbx_ctx* ctx, uint32_t max_ahead);

/* Pinned, device-mapped host memory (hipHostMalloc): a destination for wbx_set_master_target / wbx_dist_exchange /
 * wbx_finalize_master_into that the GPU writes with plain stores. */
wbx_status wbx_host_alloc(size_t bytes, void** out);
wbx_status wbx_host_free(void* p);

/* Timing of the dominant kernel (HIP events on the ctx stream): average ms per launch since reset. */
wbx_status wbx_kernel_time(wbx_ctx* ctx, int reset, double* mix_ms_avg, uint64_t* mix_launches);
/* Average ms from the end of the mix kernel to the end of the sum kernel over the same launches (launch gap + the
 * group/bus/master sum including its stores to a host-resident master target); read before resetting. */
wbx_status wbx_tail_time(wbx_ctx* ctx, double* tail_ms_avg);
/* ... and the mean idle time between two consecutive mix launches (end time stamp of one to start time stamp of the next, over
 * the pairs timed since the last reset of wbx_kernel_time; *gaps: how many) — what a step spends outside its dominant kernel
 * while renders run back to back. */
wbx_status wbx_gap_time(wbx_ctx* ctx, double* gap_ms_avg, uint64_t* gaps);
/* The template instance of the dominant kernel that the last render launched, as rocprofv3 prints it
 * ("wbx::mix_kernel<2, true, 3, 0, 1, 1, 2, 128>"); "" before the first render. */
const char* wbx_kernel_name(wbx_ctx* ctx);
/* The one resampling ratio the last render's mix was told about (MixArgs::uniform_speed: every linearly resampled row of the
 * render plays at exactly this Sampler::playback_speed_, sampler.h:24 — the hoisted-product chunk modes), 0.0 when the
 * session holds more than one ratio, none, or the plan came through layer 1.  Introspection, like wbx_kernel_name. */
double wbx_render_uniform_speed(wbx_ctx* ctx);
/* What wbx_create's probe found: the number of XCDs a launch's workgroup ids are dealt to round-robin (8 on MI355X; 4 / 2 / 1 in
 * other partition modes) — the layout the chained 128-track pieces of long renders and the segmented sequencer rest on — or 0:
 * not such a layout; the context then walks whole member lists and plans by one lane per track (same results).  Introspection. */
uint32_t wbx_xcd_count(wbx_ctx* ctx);

/* ---- layer 2: the engine surface -----------------------------------------------------------
 * Mirrors wb::Engine / wb::Track (src/engine/engine.h, track.h).  Beats are doubles as in the
 * reference; all seek / sample-index math is done in the reference's order in fp64/int64. */
wbx_status wbx_engine_create(const wbx_config* cfg, wbx_engine** out);  /* + set_audio_channel_config, engine.cpp:43-57 */
void wbx_engine_destroy(wbx_engine* e);
/* Engine::set_audio_channel_config (engine.cpp:43-57) on a live engine — a new block size, channel count or device
 * rate (the audio backend was reconfigured).  Tracks, clips, samples and the transport stay. */
wbx_status wbx_engine_set_audio_channel_config(wbx_engine* e, uint32_t output_channels, uint32_t buffer_size,
                                               uint32_t sample_rate);
/* message of the last failed wbx_engine_* / wbx_track_* call made by the CALLING thread */
const char* wbx_engine_last_error(const wbx_engine* e);
wbx_ctx* wbx_engine_ctx(wbx_engine* e);

wbx_status wbx_engine_set_bpm(wbx_engine* e, double bpm);                       /* engine.cpp:24-30 */
wbx_status wbx_engine_set_playhead_position(wbx_engine* e, double beat);       /* engine.cpp:32-41 */
wbx_status wbx_engine_add_track(wbx_engine* e, uint32_t* track_out);           /* engine.cpp:200-208 */
wbx_status wbx_engine_set_buses(wbx_engine* e, uint32_t n_buses);              /* extension A13 */
wbx_status wbx_track_set_volume(wbx_engine* e, uint32_t track, float db);      /* track.cpp:47-57 */
wbx_status wbx_track_set_pan(wbx_engine* e, uint32_t track, float pan);        /* track.cpp:59-68 */
wbx_status wbx_track_set_mute(wbx_engine* e, uint32_t track, int mute);        /* track.cpp:70-79 */
/* Engine::delete_track (engine.cpp:210-218), move_track (:228-243; the order is the summation order),
 * solo_track (:245-262; toggles the UI solo flag and mutes / unmutes the other tracks through set_mute).
 * Track indices above a deleted / between moved slots shift exactly as the reference's vector does. */
wbx_status wbx_engine_delete_track(wbx_engine* e, uint32_t slot);
wbx_status wbx_engine_clear_all(wbx_engine* e);                       /* Engine::clear_all, engine.cpp:59-66 */
wbx_status wbx_engine_move_track(wbx_engine* e, uint32_t from_slot, uint32_t to_slot);
wbx_status wbx_engine_solo_track(wbx_engine* e, uint32_t slot);
/* extension A13: bus < 0 or >= the configured number of buses routes the track straight into the master */
wbx_status wbx_track_set_bus(wbx_engine* e, uint32_t track, int32_t bus);

/* Effect slot of a track (Track::plugin_instance, track.h:124; call site track.cpp:645-662; attach / detach
 * Engine::add_plugin_to_track / delete_plugin_from_track, engine.h:227-229).  The reference's effects are third-party
 * VST3 binaries whose arithmetic is not part of this path: the slot is kept in the boundary so that a host can
 * describe such a session, but nothing is processed through it — attaching a plugin returns WBX_ERR_UNIMPLEMENTED
 * (PluginResult::Unimplemented) and leaves the slot empty; detaching and querying always succeed.
 * wbx_plugin_process_info has the fields of PluginProcessInfo (plughost/plugin_interface.h:77-90) with the
 * AudioBuffer<float>* members flattened to planar channel arrays. */
typedef struct wbx_plugin_process_info {
  uint32_t sample_count;
  uint32_t input_buffer_count;
  uint32_t output_buffer_count;
  uint32_t n_channels;
  float* const* input_buffer;       /* [n_channels][sample_count] */
  float* const* output_buffer;
  void* input_event_list;           /* MidiEventList*, unused (MIDI is out of scope) */
  double sample_rate;
  double tempo;
  double project_time_in_ppq;
  int64_t project_time_in_samples;
  int32_t playing;
} wbx_plugin_process_info;
typedef struct wbx_plugin {
  void* userdata;
  int (*process)(void* userdata, wbx_plugin_process_info* info);   /* PluginInterface::process, :146; returns a status */
} wbx_plugin;
wbx_status wbx_engine_add_plugin_to_track(wbx_engine* e, uint32_t track, const wbx_plugin* plugin);
wbx_status wbx_engine_delete_plugin_from_track(wbx_engine* e, uint32_t track);
wbx_status wbx_track_get_plugin(wbx_engine* e, uint32_t track, const wbx_plugin** plugin_out);   /* always NULL today */
/* Sample assets (SampleAsset, engine/assets_table.h:22-35): upload once, reference by id from clips. */
wbx_status wbx_engine_add_sample(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate, uint64_t frames,
                                 const void* const* planar, uint32_t* sample_out);
/* the same from interleaved decoder output (wbx_clip_upload_interleaved) */
wbx_status wbx_engine_add_sample_interleaved(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate,
                                             uint64_t frames, const void* interleaved, uint32_t* sample_id);
wbx_status wbx_engine_add_sample_synth(wbx_engine* e, int format, uint32_t channels, uint32_t sample_rate,
                                       uint64_t frames, uint64_t seed, uint32_t key_track, float amp,
                                       uint32_t* sample_out);
/* drop a sample no clip names any more (under the editor lock: use this, not wbx_clip_free, on an engine's pool) */
wbx_status wbx_engine_delete_sample(wbx_engine* e, uint32_t sample);
/* Engine::add_audio_clip (engine.cpp:293-309 -> add_to_cliplist :409-461).  A clip that overlaps existing ones
 * trims / splits / deletes them through reserve_track_region (engine.cpp:478-569), like the reference. */
wbx_status wbx_engine_add_audio_clip(wbx_engine* e, uint32_t track, double min_time, double max_time,
                                     double start_offset, uint32_t sample, double speed, float gain);

/* Clip edits (UI thread in the reference).  `clip` is the index in the track's clip list, which is kept sorted
 * by min_time (Track::update_clip_ordering, track.cpp:159-180): indices change after every edit. */
typedef struct wbx_clip_info {
  double min_time, max_time;   /* beats */
  double start_offset;         /* samples */
  double speed;                /* AudioClip::speed */
  float gain;                  /* AudioClip::gain */
  uint32_t sample;
} wbx_clip_info;
wbx_status wbx_engine_move_clip(wbx_engine* e, uint32_t track, uint32_t clip, double relative_pos);   /* engine.cpp:346-363 */
wbx_status wbx_engine_resize_clip(wbx_engine* e, uint32_t track, uint32_t clip, double relative_pos, double resize_limit,
                                  double min_length, int left_side, int shift, int stretch);          /* engine.cpp:365-398 */
wbx_status wbx_engine_delete_clip(wbx_engine* e, uint32_t track, uint32_t clip);                       /* engine.cpp:400-407 */
wbx_status wbx_engine_delete_region(wbx_engine* e, uint32_t track, double min_time, double max_time);  /* engine.cpp:463-475 */
wbx_status wbx_engine_set_clip_gain(wbx_engine* e, uint32_t track, uint32_t clip, float gain);         /* engine.cpp:1460-1464 */
wbx_status wbx_engine_clip_count(wbx_engine* e, uint32_t track, uint32_t* count);
wbx_status wbx_engine_get_clip(wbx_engine* e, uint32_t track, uint32_t clip, wbx_clip_info* out);
/* The clip placement arithmetic by itself (src/engine/clip_edit.h:10-150; audio clips), fp64, host-only. */
void wbx_calc_move_clip(double clip_min, double clip_max, double relative_pos, double min_move, double* new_min,
                        double* new_max);
void wbx_calc_resize_clip(double clip_min, double clip_max, double clip_start_offset, double clip_speed, double sample_rate,
                          double sample_count, double relative_pos, double resize_limit, double min_length,
                          double min_resize_pos, double beat_duration, int is_min, int shift, int stretch,
                          int clamp_at_resize_pos, double* out_min, double* out_max, double* out_start_offset,
                          double* out_speed);
double wbx_calc_clip_shift(double start_offset, double relative_pos, double beat_duration, double sample_rate);
double wbx_shift_clip_content(double start_offset, double speed, double sample_rate, double relative_pos,
                              double beat_duration);
wbx_status wbx_engine_play(wbx_engine* e);                                      /* engine.cpp:68-80 */
wbx_status wbx_engine_stop(wbx_engine* e);                                      /* engine.cpp:82-93 */

/* Engine::process (engine.cpp:1576-1654): one block into out_planar[c][0..F). */
wbx_status wbx_engine_process(wbx_engine* e, float* const* out_planar);
/* Engine::process and the audio back end's conversion to the device's sample format in one call (the call site
 * audio_io_pulseaudio.cpp:419-461: process(), then output_buffer.interleave_samples_to(buffer, 0, n, format) ->
 * core/audio_format_conv.cpp:5-91): dst receives the block as F*C interleaved samples of `out_format` (WBX_OUT_*), the
 * conversion running as the epilogue of the sum kernel — one launch fewer than process + wbx_fetch_interleaved. */
wbx_status wbx_engine_process_interleaved(wbx_engine* e, int out_format, void* dst);
/* K consecutive blocks in one device pass (render-ahead / offline): sequencing, mixing, summing and
 * clamping all on the device.  Asynchronous; results via wbx_fetch(wbx_engine_ctx(e), ...). */
wbx_status wbx_engine_render(wbx_engine* e, uint32_t n_blocks);
/* Transport after the last process/render (engine.h:44-46), for bit-exact checks. */
wbx_status wbx_engine_transport(wbx_engine* e, double* playhead, double* sample_position, int* playing);
/* VUMeter::level per track/channel: max since the last call (vu_meter.h:20-40). levels: [n_tracks][C]. */
wbx_status wbx_engine_levels(wbx_engine* e, float* levels, uint32_t n_tracks);

/* What the last process / render took from the state it shares with the UI thread (the threading contract is the
 * reference's: one audio thread in process / render holding the editor lock — Engine::editor_lock, engine.cpp:1587-1651
 * — one UI thread whose edits take the same lock while Track::set_volume / set_pan / set_mute go through a per-track
 * single-producer ring of 64 messages, track.cpp:47-79, core/queue.h:142-196):
 *   edits_seen  number of locked edit calls (clip / track / transport edits, play, stop) completed before it
 *   drained     [n_tracks] cumulative parameter messages the audio thread has taken from each track's ring
 * Together with the UI thread's own log this reconstructs the exact state every block was rendered from. */
wbx_status wbx_engine_thread_stats(wbx_engine* e, uint64_t* edits_seen, uint64_t* drained, uint32_t n_tracks);

/* How the device sequencer planned the renders so far (diagnostic; no reference counterpart — Track::process_event,
 * track.cpp:258-451, is one thread).  Long renders of sessions cut into clips are planned by one lane per (track, SEGMENT of
 * the render) instead of one lane per track: a segment's lane works out the state its first block starts from by itself, and
 * the lane that completes a track checks every seam against the state the segment before really ended with, planning again —
 * in one walk, from the true state — whatever follows a seam that did not hold (results are the one-walk plan's either way).
 *   out[0]  renders planned by segments     out[1]  tracks that had a seam that did not hold
 *   out[2]  segments planned again          out[3]  segments per track of the last such render */
wbx_status wbx_engine_sequencer_stats(wbx_engine* e, uint64_t out[4]);

/* How wbx_engine_process — the audio callback, audio_io_pulseaudio.cpp:396-466 calling Engine::process once per device
 * period — has run so far (diagnostic; no reference counterpart).  A block is ONE launch where an instance exists for the
 * block shape; its sum is spread over all workgroups behind a barrier that needs the whole grid resident at once.  A workgroup
 * that waits there too long (a CU mask, a device shared with another process) gives up; the block is then mixed and summed
 * again through three launches — its result is the same — and the engine stops spreading.
 *   out[0]  blocks that ran as one launch       out[1]  ... of those, with the sum spread over the workgroups
 *   out[2]  blocks mixed again after a give-up  out[3]  1 when the engine has stopped spreading */
wbx_status wbx_engine_callback_stats(wbx_engine* e, uint64_t out[4]);

/* Engine::perf_measurer (engine.h:64, core/timing.h:54-67): Engine::process times itself (ScopedPerformanceCounter,
 * engine.cpp:1577) and ends with perf_measurer.update(duration_ms, audio_buffer_duration_ms) (engine.cpp:1653) — usage moves a
 * quarter of the way towards duration / period; the UI shows get_usage(), clamped to [0, 1] (ui/control_bar.cpp:54).  Here every
 * wbx_engine_process / wbx_engine_process_interleaved call feeds its own wall time (launch, device pass, copy-out) and the
 * period of its block, period_to_ms(buffer_size_to_period(block_frames, sample_rate)) (engine.cpp:52, engine/audio_io.h:187-195).
 * Any thread.  last_block_ms (may be NULL): what the last update was fed.  Batch renders (wbx_engine_render) do not feed it:
 * the reference has no such call. */
wbx_status wbx_engine_perf_usage(wbx_engine* e, double* usage, double* last_block_ms);
/* the arithmetic behind it, host-only: PerformanceMeasurer::update (timing.h:57-62), ::get_usage (timing.h:64-66),
 * Engine::audio_buffer_duration_ms (engine.cpp:52) */
double wbx_calc_perf_update(double usage, double duration_ms, double period_ms);
double wbx_calc_perf_usage(double usage);
double wbx_calc_buffer_period_ms(uint32_t buffer_size, uint32_t sample_rate);

/* The plan the device sequencer produced for the last process/render: one record per Sampler::stream
 * call, ordered by (block, track, call).  For seek-math parity checks (bit patterns, not tolerances). */
typedef struct wbx_plan_record {
  uint32_t block, track;
  uint32_t buffer_offset, num_samples;   /* as handed to Sampler::stream */
  uint32_t num_actual;                   /* after the clip-tail limit, sampler.cpp:102-104 */
  uint32_t sample;
  double sample_offset;                  /* Sampler::sample_offset_ before the call */
  double playback_speed;
  float gain;
  uint32_t _pad;
} wbx_plan_record;
wbx_status wbx_engine_fetch_plan(wbx_engine* e, wbx_plan_record* out, size_t cap, size_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* WBX_H */
