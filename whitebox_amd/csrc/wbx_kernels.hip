// wbx_kernels.hip — gfx950 (CDNA4, wave64) kernels of the whitebox mix path.  HIP only, no other target.
//
//   plan_kernel      one lane per track: the reference's clip sequencer + sampler-state update for K
//                    consecutive blocks (wbx_seq.h), emitting a 16-B DRow per (block, track) and one 64-B
//                    DTrackBlock template per run of steady blocks / per block with events
//   gen_kernel       pre-render of the rare track-blocks the hot loop cannot stream (clip boundaries, ...)
//   mix_kernel       the hot kernel: workgroup = (track group, block[, frame tile]); the group's
//                    records (gain / pan / resample parameters) are staged in LDS, each lane owns 4
//                    consecutive output frames of one channel, clip audio is streamed with 16-B loads
//                    through a software pipeline, rendered, scaled and accumulated in registers IN TRACK
//                    ORDER; per-track peaks via permlane-swap / DPP wave maxima, four tracks at a time.
//                    Resampled rows: 5-sample window (fp32, 16-bit PCM) or per-frame tap pairs (speed > 1,
//                    24/32-bit PCM); blocks shorter than 512 frames: several blocks per workgroup
//   sum_kernel       group sums -> bus sums -> master (fixed order), master clamp
//   clamp / clamp_into / convert / synth: small helpers
//
// HBM-bound integer/fp32 streaming: no MFMA (≈3 flop per 4 B).  Parity-critical arithmetic uses the
// explicitly rounded intrinsics (__fmul_rn, __dadd_rn, ...) and the file is built with
// -ffp-contract=off so that nothing is fused: the reference build has no FMA.
#include <cstdlib>

#include "wbx_mix.h"
#include "wbx_seq.h"
#include "wbx_sum.h"
#include "wbx_callback.h"

namespace wbx {

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void plan_body(const PlanArgs& a) {
  // The K per-block transport records are the same for every track: one lane computes them into LDS with
  // exactly the arithmetic of Engine::process (engine.cpp:1578-1585 per block, :1619-1623 between blocks).
  // Batch renders read them from device memory instead (PlanArgs::times, computed by the host and copied in): K records in the LDS of
  // every workgroup — 64 KiB for 2048 blocks — keep the sequencer's workgroups off any CU that holds four mix workgroups.
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const DBlockTime* s_times = a.times;
  if (!s_times) {
    if (threadIdx.x == 0) block_times(a, reinterpret_cast<DBlockTime*>(s_raw));
    __syncthreads();
    s_times = reinterpret_cast<const DBlockTime*>(s_raw);
  }
  if (threadIdx.x >= a.lanes) return;
  const uint32_t t = blockIdx.x * a.lanes + threadIdx.x;
  if (t >= a.n_tracks) return;
  plan_track(a, t, s_times);   // wbx_seq.h: the same source runs on the host in the CPU-side tests
}

__global__ __launch_bounds__(64) void plan_kernel(PlanArgs a) { plan_body(a); }

// The sequencer of a batch render runs BESIDE the previous render's mix.  At most 128 VGPRs (four waves per SIMD): a wave
// then fits the hole ONE retiring mix wave leaves.  With the 243 the unbounded instance takes it needs two holes on one
// SIMD at once, which a running mix never offers — the high-priority plan sat out the whole mix, ran in the drain at its
// end, and the next mix waited for it.  The spills (scratch) make it slower alone; beside a mix it is hidden.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void plan_kernel_beside(PlanArgs a) { plan_body(a); }

// The sequencer cut along the time axis (wbx_seq.h, plan_segment): workgroup = 64 tracks x ONE segment of seg_len blocks, so
// that the transport records its lanes look at all the time — the segment's own and a margin either side — are one window in
// LDS (a few KiB instead of the 64 KiB of all K records: the workgroups fit beside a running mix); the occasional look further
// back (the run-up of a long clip) goes to the device table.
constexpr uint32_t kSegMargin = 48;   // records in front of the segment (the run-up of a clip of a session cut into clips) and behind it
__device__ __forceinline__ void plan_seg_body(const PlanArgs& a, const SegArgs& g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  DBlockTime* win = reinterpret_cast<DBlockTime*>(s_raw);
  // Workgroup -> (64-track group, segment).  Linear workgroup ids are dealt to the 8 XCDs round-robin; the id is laid out so
  // that id mod 8 = group mod 8: every segment of a track runs on ONE XCD, behind one L2 — the rows of a segment whose seam
  // guess missed are written a second time by the lane that completes the track, and two versions of a row in two L2s would
  // reach memory in no defined order.  Segments are dispatched in order (segment 0 of every group first).
  const uint32_t n_groups = (a.n_tracks + 63u) / 64u, g8 = (n_groups + 7u) / 8u;
  const uint32_t q = blockIdx.x >> 3, s = q / g8, grp = (q - s * g8) * 8u + (blockIdx.x & 7u);
  if (grp >= n_groups) return;   // (group counts that are no multiple of 8: the padding of the last octet)
  const uint32_t b0 = s * g.seg_len;
  const uint32_t w0 = b0 > kSegMargin ? b0 - kSegMargin : 0u;
  const uint32_t w1 = b0 + g.seg_len + kSegMargin < a.n_blocks ? b0 + g.seg_len + kSegMargin : a.n_blocks;
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.times + w0);
    uint4* dst = reinterpret_cast<uint4*>(win);
    const uint32_t n16 = (w1 - w0) * (uint32_t)(sizeof(DBlockTime) / 16u);
    for (uint32_t i = threadIdx.x; i < n16; i += 64u) dst[i] = src[i];
  }
  __syncthreads();
  const uint32_t t = grp * 64u + threadIdx.x;
  if (t >= a.n_tracks) return;
  const TimesWindow tv{a.times, win, w0, w1};
  DTrackState gs{}, en{};
  plan_segment(a, t, s, g.seg_len, g.n_segs, tv, &gs, &en);
  // The seam states go where the lane that completes the track finds them — a lane of another workgroup, maybe behind another
  // L2: write-through stores, acknowledged before the ticket is taken, read back past the caches (the pattern of the
  // one-launch callback, wbx_callback.h: no fence — a fence is an L2 write-back for everybody).
  const size_t at = (size_t)t * g.n_segs + s;
  auto put = [](DTrackState* dst, const DTrackState& v) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(DTrackState) / 4u; i++) __hip_atomic_store(d + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (s > 0u) put(&g.guess[at], gs);
  put(&g.ends[at], en);
  __builtin_amdgcn_s_waitcnt(0);
  // The ticket's low half counts the segments of the track that are done; its high half collects the XCDs they ran on (one
  // bit each, or-ed in before the count: the lane that completes the track has everybody's).  What the layout above rests on —
  // linear workgroup ids dealt round-robin to 8 / 4 / 2 / 1 XCDs — is thereby CHECKED where it matters: a track that is planned
  // again over rows another L2 may still hold a dirty copy of raises status bit 7, the render reports WBX_ERR_DEVICE and the
  // engine plans by one lane per track from then on (wbx_engine.hip).
  const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(63508) & 15u;   // HW_REG_XCC_ID (hwreg 20, offset 0, 4 bits)
  __hip_atomic_fetch_or(g.ticket + t, 0x10000u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t had = __hip_atomic_fetch_add(g.ticket + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((had & 0xFFFFu) + 1u != g.n_segs) return;
  // this lane completed the track: the seams in order.  All stood (nearly always): the track's state for the next render is the
  // last segment's.  One did not: this lane plans the rest of the track again, from the state the segment before really ended
  // with (every other lane of the track is done: nobody writes those rows any more).
  __hip_atomic_store(g.ticket + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  auto get = [](const DTrackState* src) {
    DTrackState v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(DTrackState) / 4u; i++) w[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  const uint32_t bad = plan_check_seams(a, t, g.n_segs, g.guess, g.ends, get);
  if (bad < g.n_segs) {   // (rare: the lanes of this wave that are done wait for it)
    const uint32_t xccs = had >> 16;
    if (xccs & (xccs - 1u)) raise_status(a.status, 128u);   // the track's segments sat behind more than one L2
    plan_redo_track(a, t, bad, g.seg_len, tv, get(&g.ends[(size_t)t * g.n_segs + bad - 1u]));
    if (g.stats) {
      atomicAdd(g.stats + 0, 1u);
      atomicAdd(g.stats + 1, g.n_segs - bad);
    }
  }
}
__global__ __launch_bounds__(64) void plan_seg_kernel(PlanArgs a, SegArgs g) { plan_seg_body(a, g); }
// (the register-capped form, as plan_kernel_beside: a wave that fits the hole one retiring mix wave leaves)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void plan_seg_kernel_beside(PlanArgs a, SegArgs g) { plan_seg_body(a, g); }

// the host's table of per-block transport records (pinned memory) -> device memory, in front of a batch render's plan.  A
// kernel of our own, not hipMemcpyAsync: the runtime's host-to-device path made the submitting thread wait for the stream
// (measured: the renders of a 256-track session then ran one after the other instead of overlapped)
// (`zero`: the plan buffer's four counters, cleared here instead of by a memset launch of their own — or null)
__global__ __launch_bounds__(256) void times_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16, uint32_t* zero) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n16) dst[i] = src[i];
  if (zero && i < 4u) zero[i] = 0u;
}

// ------------------------------------------------------------------------------------------------
// per-sample rendering helpers
// ------------------------------------------------------------------------------------------------

// One source sample of Sampler::stream for destination frame jj of segment sg, channel c:
// unity path sampler.cpp:106-158, linear path sampler.cpp:34-59 (normalisers :7-18 and :95-97).
__device__ __forceinline__ float sample_at(const DSeg& sg, uint32_t c, uint32_t jj) {
  const void* base = c ? sg.src[1] : sg.src[0];   // (no dynamic index: keeps the descriptor in registers)
  const float WBX_GLOBAL* bf = as_global<float>(base);
  const int16_t WBX_GLOBAL* b16 = as_global<int16_t>(base);
  const int32_t WBX_GLOBAL* b32 = as_global<int32_t>(base);
  if (sg.speed == 1.0) {
    const uint32_t idx = u32_of_double_x86(sg.pos) + jj;                 // :107 (the low word, as x86-64 converts)
    switch (sg.format) {
      case FMT_F32: return bf[idx];
      case FMT_I16: {
        const float norm = 1.0f / 32767.0f;                              // :95
        return clampf(__fmul_rn((float)b16[idx], norm), -1.0f, 1.0f);
      }
      case FMT_I24: {
        const double norm = 1.0 / 8388607.0;                             // :96
        return (float)clampd(__dmul_rn((double)b32[idx], norm), -1.0, 1.0);
      }
      default: {
        const double norm = 1.0 / 2147483647.0;                          // :97
        return (float)clampd(__dmul_rn((double)b32[idx], norm), -1.0, 1.0);
      }
    }
  }
  const double x = __dadd_rn(sg.pos, __dmul_rn((double)(int32_t)jj, sg.speed));   // :50
  const long long ix = (long long)x;                                               // :51
  const float fx = (float)__dsub_rn(x, (double)ix);                                // :52 (negative for x < 0: ix truncates)
  // Q12: a NEGATIVE playback speed (calc_resize_clip's stretch, clip_edit.h:59-67,110-118) runs the position
  // below zero; the reference then reads the heap in front of the channel array (sampler.cpp:53-54, undefined).  A tap at
  // a negative index reads 0 — never memory in front of the clip.
  const bool ta = ix >= 0, tb = ix >= -1;
  float a, b;
  switch (sg.format) {
    case FMT_F32:
      a = ta ? bf[ix] : 0.0f;
      b = tb ? bf[ix + 1] : 0.0f;
      break;
    case FMT_I16: {
      const float norm = (float)(1.0 / 32767.0);                                   // :9-10
      a = __fmul_rn(norm, (float)(ta ? b16[ix] : (int16_t)0));
      b = __fmul_rn(norm, (float)(tb ? b16[ix + 1] : (int16_t)0));
      break;
    }
    case FMT_I24: {
      const double norm = 1.0 / 8388607.0;                                         // :11-12
      a = (float)__dmul_rn(norm, (double)(ta ? b32[ix] : 0));
      b = (float)__dmul_rn(norm, (double)(tb ? b32[ix + 1] : 0));
      break;
    }
    default: {
      const double norm = 1.0 / 2147483647.0;                                      // :13-14
      a = (float)__dmul_rn(norm, (double)(ta ? b32[ix] : 0));
      b = (float)__dmul_rn(norm, (double)(tb ? b32[ix + 1] : 0));
      break;
    }
  }
  return __fadd_rn(a, __fmul_rn(fx, __fsub_rn(b, a)));                             // :55
}

// Generic track-block: any number of segments, any coverage, any format.  Returns the track's
// mixing-buffer value for frame j of channel c BEFORE the track gain (the buffer the reference clears
// at engine.cpp:1602 and Sampler::stream accumulates into, sampler.cpp:56,152).
// One lane's 4 frames of a generic track-block: the mixing-buffer values BEFORE the track gain.
__device__ __forceinline__ f4 render_generic(const DTrackBlock& tb, const DSeg* pool, uint32_t c, uint32_t j0) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const uint32_t nseg = tb.nseg;
  for (uint32_t s = 0; s < nseg; s++) {
    const DSeg sg = (s == 0) ? get_seg0(tb) : pool[(size_t)tb.extra * kChunk + (s - 1)];
    const uint32_t d0 = sg.dst_start, n = sg.len;
    if (j0 + 4u <= d0 || j0 >= d0 + n) continue;   // none of the lane's frames lies in this segment
    // fp32 segments whose source positions stay below 2^31 (all but multi-hour clips): 32-bit index math, and one
    // 16-B load for a lane whose four frames all lie inside a unity-speed segment
    // (and a speed above zero: a negative one runs the position below zero, where `fract` is not x - trunc(x) and the
    //  taps in front of the clip read 0 — sample_at, Q12)
    const bool small = sg.pos >= 0.0 && sg.speed > 0.0 && sg.pos + (double)n * (sg.speed > 1.0 ? sg.speed : 1.0) < 2147483000.0;
    if (sg.format == FMT_F32 && small) {
      const float WBX_GLOBAL* bf = as_global<float>(c ? sg.src[1] : sg.src[0]);
      if (sg.speed == 1.0) {
        const uint32_t base = (uint32_t)sg.pos;                                            // sampler.cpp:107
        if (j0 >= d0 && j0 + 4u <= d0 + n) {
          const f4u v = *reinterpret_cast<const f4u WBX_GLOBAL*>(bf + base + (j0 - d0));
          acc[0] = __fadd_rn(acc[0], __fmul_rn(v.x, sg.gain));                             // :151-152
          acc[1] = __fadd_rn(acc[1], __fmul_rn(v.y, sg.gain));
          acc[2] = __fadd_rn(acc[2], __fmul_rn(v.z, sg.gain));
          acc[3] = __fadd_rn(acc[3], __fmul_rn(v.w, sg.gain));
        } else {
#pragma unroll
          for (uint32_t e = 0; e < 4; e++) {
            const uint32_t j = j0 + e;
            if (j >= d0 && j < d0 + n) acc[e] = __fadd_rn(acc[e], __fmul_rn(bf[base + (j - d0)], sg.gain));
          }
        }
      } else {
#pragma unroll
        for (uint32_t e = 0; e < 4; e++) {
          const uint32_t j = j0 + e;
          if (j >= d0 && j < d0 + n) {
            const double x = __dadd_rn(sg.pos, __dmul_rn((double)(int32_t)(j - d0), sg.speed));   // :50
            const int ix = (int)x;                                                                // :51
            const float fx = (float)__builtin_amdgcn_fract(x);                                    // :52 (x >= 0: exact)
            const float a = bf[ix], b = bf[ix + 1];
            acc[e] = __fadd_rn(acc[e], __fmul_rn(__fadd_rn(a, __fmul_rn(fx, __fsub_rn(b, a))), sg.gain));   // :55-56
          }
        }
      }
      continue;
    }
#pragma unroll
    for (uint32_t e = 0; e < 4; e++) {
      const uint32_t j = j0 + e;
      if (j >= d0 && j < d0 + n) acc[e] = __fadd_rn(acc[e], __fmul_rn(sample_at(sg, c, j - d0), sg.gain));
    }
  }
  return f4{acc[0], acc[1], acc[2], acc[3]};
}

// ------------------------------------------------------------------------------------------------
// gen: pre-render of the KIND_GENERIC track-blocks (clip boundaries inside a block, several segments ...).
// One WAVE per queued record, no LDS and no barriers: the 64-B record is read one dword per lane and broadcast
// into scalar registers, so the segment loop and its format / speed branches are wave-uniform; the wave renders
// the track's mixing buffer (all segments, sampler.cpp:88-210) into a scratch row and rewrites the record as a
// KIND_UNITY read of that row with clip gain 1, so the mix kernel's hot loop never sees partial rows.  The rows
// are independent latency chains (queue entry -> record -> segments -> samples): a full complement of waves
// per CU, each on its own row, hides them.  Grid-stride over the queue.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 6) void gen_kernel(GenArgs a) {
  const uint32_t count = min(*a.gen_count, a.gen_cap);
  const uint32_t F = a.block_frames, C = a.channels, S4 = F >> 2;
  const size_t row_floats = (size_t)C * (F + 8);
  const uint32_t lane = threadIdx.x & 63u;
  // (waves are independent — no LDS, no barrier: launched as one-wave workgroups, which find a slot beside a running mix
  //  where a four-wave workgroup with scratch waits for the mix to drain)
  const uint32_t wpb = blockDim.x >> 6;
  const uint32_t wave = blockIdx.x * wpb + (threadIdx.x >> 6), n_waves = gridDim.x * wpb;
  // the queue entry and the record of the wave's NEXT row are fetched while the current one is rendered (queued
  // templates are distinct, so the rewrite at the end of a row cannot touch the one in flight)
  uint32_t idx_n = 0u, w_n = 0u;
  if (wave < count) {
    idx_n = a.gen_list[wave];
    w_n = reinterpret_cast<const uint32_t*>(a.tmpl + idx_n)[lane & 15u];
  }
  for (uint32_t i = wave; i < count; i += n_waves) {
    const uint32_t idx = idx_n, w = w_n;
    if (i + n_waves < count) {
      idx_n = a.gen_list[i + n_waves];
      w_n = reinterpret_cast<const uint32_t*>(a.tmpl + idx_n)[lane & 15u];
    }
    DTrackBlock rec;   // assembled field by field from the broadcast dwords (wave-uniform, scalar registers)
    {
      auto rl = [&](int k) { return (uint32_t)__builtin_amdgcn_readlane((int)w, k); };
      rec.src[0] = (const void*)(((uint64_t)rl(1) << 32) | rl(0));
      rec.src[1] = (const void*)(((uint64_t)rl(3) << 32) | rl(2));
      rec.pos = __longlong_as_double((long long)(((uint64_t)rl(5) << 32) | rl(4)));
      rec.speed = __longlong_as_double((long long)(((uint64_t)rl(7) << 32) | rl(6)));
      rec.gain = __uint_as_float(rl(8));
      rec.g[0] = __uint_as_float(rl(9));
      rec.g[1] = __uint_as_float(rl(10));
      const uint32_t q = rl(11), h = rl(12), m = rl(13);
      rec.nseg = (uint8_t)(q & 0xFFu);
      rec.kind = (uint8_t)((q >> 8) & 0xFFu);   // (queued records are KIND_GENERIC: never flagged KIND_PARTIAL)
      rec.dst_start = (uint16_t)(q >> 16);
      rec.len = (uint16_t)(h & 0xFFFFu);
      rec.req_len = (uint16_t)(h >> 16);
      rec.format = (uint8_t)(m & 0xFFu);
      rec.flags = (uint8_t)((m >> 8) & 0xFFu);
      rec._pad = (uint16_t)(m >> 16);
      rec.sample = rl(14);
      rec.extra = rl(15);
    }
    float* row = a.rows + (size_t)i * row_floats;
    // The usual boundary row: one or two fp32 segments that do not overlap (a clip ends and/or the next one starts
    // inside the block), source positions below 2^31.  Every frame then belongs to at most one segment: pick the
    // segment per frame and read its tap pair with ONE unconditional 8-B load (clamped index), eight frames of a
    // lane in flight together — straight-line code, so the row costs a few memory round trips instead of one per
    // segment and slot.  Anything else takes the general renderer.
    const DSeg s0 = get_seg0(rec);
    bool fast = (rec.nseg == 1u || rec.nseg == 2u) && s0.format == FMT_F32 && s0.len != 0u && s0.pos >= 0.0 &&
                s0.speed > 0.0 && s0.pos + (double)s0.len * (s0.speed > 1.0 ? s0.speed : 1.0) < 2147483000.0;
    DSeg s1 = s0;
    s1.dst_start = 0xFFFFu;   // (no second segment: never selected)
    s1.len = 0u;
    if (fast && rec.nseg == 2u) {
      s1 = a.pool[(size_t)rec.extra * kChunk];
      fast = s1.format == FMT_F32 && s1.len != 0u && s1.pos >= 0.0 && s1.speed > 0.0 &&
             s1.pos + (double)s1.len * (s1.speed > 1.0 ? s1.speed : 1.0) < 2147483000.0 &&
             (uint32_t)s0.dst_start + s0.len <= s1.dst_start;
    }
    if (fast) {
      typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
      const uint32_t d0 = s0.dst_start, n0 = s0.len, d1 = s1.dst_start, n1 = s1.len;
      const bool unity0 = s0.speed == 1.0, unity1 = s1.speed == 1.0;
      struct Taps {
        f2u t0, t1, t2, t3;
        float f0, f1, f2, f3;
        uint32_t meta;   // per frame e, bits 4e..4e+2: inside a segment / that segment plays at unity speed / second segment
      };
      auto load_slot = [&](uint32_t slot) {
        Taps t;
        const bool valid = slot < C * S4;
        const uint32_t c = valid ? slot / S4 : 0u, j0 = valid ? (slot - c * S4) * 4u : 0u;
        const float WBX_GLOBAL* p0 = as_global<float>(c ? s0.src[1] : s0.src[0]);
        const float WBX_GLOBAL* p1 = as_global<float>(c ? s1.src[1] : s1.src[0]);
        t.meta = 0u;
        auto one = [&](uint32_t e, f2u& tap, float& fx) {
          const uint32_t j = j0 + e;
          const bool second = j >= d1;
          const uint32_t d = second ? d1 : d0, n = second ? n1 : n0;
          const bool in = valid && j >= d && j < d + n;
          const uint32_t jj = in ? j - d : 0u;
          const double pos = second ? s1.pos : s0.pos, sp = second ? s1.speed : s0.speed;
          const bool unity = second ? unity1 : unity0;
          const double x = __dadd_rn(pos, __dmul_rn((double)(int32_t)jj, sp));                  // sampler.cpp:50
          const int ix = unity ? (int)((uint32_t)pos + jj) : (int)x;                           // :107 / :51
          fx = (float)__builtin_amdgcn_fract(x);                                                // :52 (x >= 0: exact)
          tap = *reinterpret_cast<const f2u WBX_GLOBAL*>((second ? p1 : p0) + ix);
          t.meta |= ((in ? 1u : 0u) | (unity ? 2u : 0u) | (second ? 4u : 0u)) << (4u * e);
        };
        one(0u, t.t0, t.f0);
        one(1u, t.t1, t.f1);
        one(2u, t.t2, t.f2);
        one(3u, t.t3, t.f3);
        return t;
      };
      auto store_slot = [&](uint32_t slot, const Taps& t) {
        if (slot >= C * S4) return;
        const uint32_t c = slot / S4, j0 = (slot - c * S4) * 4u;
        auto one = [&](uint32_t e, const f2u& tap, float fx) {
          const uint32_t m = t.meta >> (4u * e);
          const float lin = __fadd_rn(tap.x, __fmul_rn(fx, __fsub_rn(tap.y, tap.x)));           // :55
          const float smp = (m & 2u) ? tap.x : lin;                                             // :151
          const float g = (m & 4u) ? s1.gain : s0.gain;
          return (m & 1u) ? __fadd_rn(0.0f, __fmul_rn(smp, g)) : 0.0f;                          // :56 / :152 into the cleared buffer
        };
        const f4 r = {one(0u, t.t0, t.f0), one(1u, t.t1, t.f1), one(2u, t.t2, t.f2), one(3u, t.t3, t.f3)};
        *reinterpret_cast<f4*>(row + (size_t)c * (F + 8) + j0) = r;
      };
      for (uint32_t slot0 = lane; slot0 < C * S4; slot0 += 128u) {
        const Taps ta = load_slot(slot0), tb = load_slot(slot0 + 64u);
        store_slot(slot0, ta);
        store_slot(slot0 + 64u, tb);
      }
    } else {
      for (uint32_t slot = lane; slot < C * S4; slot += 64u) {
        const uint32_t c = slot / S4, j0 = (slot - c * S4) * 4u;
        const f4 r = render_generic(rec, a.pool, c, j0);
        *reinterpret_cast<f4*>(row + (size_t)c * (F + 8) + j0) = r;
      }
    }
    if (lane < 8u * C) {   // the 8 floats behind each channel row that a 5-sample window load may touch
      const uint32_t c = lane >> 3;
      row[(size_t)c * (F + 8) + F + (lane & 7u)] = 0.0f;
    }
    if (lane < 16u) reinterpret_cast<uint32_t*>(a.saved + i)[lane] = w;
    if (lane == 0u) {
      DTrackBlock u = rec;
      u.src[0] = row;
      u.src[1] = row + (C > 1 ? (F + 8) : 0);
      u.pos = 0.0;
      u.speed = 1.0;
      u.gain = 1.0f;
      u.dst_start = 0;             // the row holds the whole block (the record's own segment bounds live on in `saved`)
      u.len = (uint16_t)F;
      u.kind = KIND_UNITY;
      u.format = FMT_F32;   // the row is fp32 whatever the clip's storage format (MODE_G reads by format)
      a.tmpl[idx] = u;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// sum: grid = (n_blocks, C*F/4/64), block = 64 (one wave).  master = (((direct groups in order) + bus 0) + bus 1) + ...
// with bus u = in-order sum of its groups (AudioBuffer::mix order, audio_buffer.h:73-82), then the
// clamp of engine.cpp:1627-1636.  Groups arrive sorted: direct ones first, then by bus.
// ------------------------------------------------------------------------------------------------
template <int PF, bool BUSES, bool IL = false>
__global__ __launch_bounds__(64) void sum_kernel(SumArgs a) {
  if (a.status_dst && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) {
    const uint32_t queued = a.status_src[2];
    a.status_dst[threadIdx.x] = a.status_src[threadIdx.x];
    if (a.zero_status && queued == 0u) a.status_src[threadIdx.x] = 0u;
  }
  const uint32_t F = a.block_frames, C = a.channels;
  const uint32_t slot = blockIdx.y * 64u + threadIdx.x;
  if (slot >= ((IL ? F : C * F) >> 2)) return;
  // a wave walks blocks blockIdx.x, + gridDim.x, ...: the launch stays a few waves per CU however long the render is, so the
  // NEXT render's mix (other stream) finds free wave slots at once — a sum that fills every slot of the chip with waves
  // waiting on their PCIe stores holds that mix back for its whole duration
  for (uint32_t b = blockIdx.x; b < a.n_blocks; b += gridDim.x) sum_block<PF, BUSES, IL>(a, b, slot);   // wbx_sum.h
}

// buses with no member groups stay zero: cleared before the launch by the runtime.

// the clamp alone, over a device buffer (root rank after the RCCL reduce)
__global__ __launch_bounds__(256) void clamp_kernel(float* buf, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float v = buf[i];
  buf[i] = v > 1.0f ? 1.0f : (v < -1.0f ? -1.0f : v);
}

// the same out of place: dst may be pinned host memory (the root's final master leaves the GPU as plain stores of
// this kernel — no copy-engine transfer, whose completion latency is erratic next to a bandwidth-bound kernel)
__global__ __launch_bounds__(256) void clamp_into_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n, int clamp) {
  const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 4u;
  if (i >= n) return;
  if (i + 4u <= n) {
    f4 v = *reinterpret_cast<const f4*>(src + i);
    if (clamp) {
      v.x = v.x > 1.0f ? 1.0f : (v.x < -1.0f ? -1.0f : v.x);
      v.y = v.y > 1.0f ? 1.0f : (v.y < -1.0f ? -1.0f : v.y);
      v.z = v.z > 1.0f ? 1.0f : (v.z < -1.0f ? -1.0f : v.z);
      v.w = v.w > 1.0f ? 1.0f : (v.w < -1.0f ? -1.0f : v.w);
    }
    *reinterpret_cast<f4*>(dst + i) = v;
  } else {
    for (size_t k = i; k < n; k++) {
      const float v = src[k];
      dst[k] = clamp ? (v > 1.0f ? 1.0f : (v < -1.0f ? -1.0f : v)) : v;
    }
  }
}

// multi-GPU, ordered mode: the partial masters of all ranks lie behind one another in `g` ([world][n]); the master is
// their sum in RANK order starting from the cleared output buffer — (((0 + p0) + p1) + ...) — like the reference adds
// track after track (audio_buffer.h:73-82), then the clamp of engine.cpp:1627-1636 (compare-based: NaN passes)
__global__ __launch_bounds__(256) void ordered_add_kernel(const float* __restrict__ g, float* __restrict__ dst, size_t n, uint32_t world, int clamp) {
  const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 4u;
  if (i >= n) return;   // (n is a multiple of 4: C*F/4 lanes per block)
  f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (uint32_t r = 0; r < world; r++) {
    const f4 v = *reinterpret_cast<const f4*>(g + (size_t)r * n + i);
    acc.x = __fadd_rn(acc.x, v.x);
    acc.y = __fadd_rn(acc.y, v.y);
    acc.z = __fadd_rn(acc.z, v.z);
    acc.w = __fadd_rn(acc.w, v.w);
  }
  if (clamp) {
    acc.x = acc.x > 1.0f ? 1.0f : (acc.x < -1.0f ? -1.0f : acc.x);
    acc.y = acc.y > 1.0f ? 1.0f : (acc.y < -1.0f ? -1.0f : acc.y);
    acc.z = acc.z > 1.0f ? 1.0f : (acc.z < -1.0f ? -1.0f : acc.z);
    acc.w = acc.w > 1.0f ? 1.0f : (acc.w < -1.0f ? -1.0f : acc.w);
  }
  *reinterpret_cast<f4*>(dst + i) = acc;
}

// planar fp32 master [K][C][F] -> interleaved device-format samples [K*F][C]; reference
// core/audio_format_conv.cpp:5-20 (i16), :45-60 (i24 in 32-bit containers), :62-77 (i32), :79-91 (f32).
// Packed 24-bit (:22-43): the reference's writer has no channel term in its destination index, so what a block's
// conversion leaves is the LAST channel, 3 bytes per frame at byte 3*frame — dst is [K][F][3] here.
__global__ __launch_bounds__(256) void convert_kernel(const float* master, void* dst, uint32_t n_blocks, uint32_t F,
                                                      uint32_t C, int fmt) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;   // index into [K*F][C]  (packed 24-bit: [K*F])
  if (fmt == 5) {
    if (i >= (size_t)n_blocks * F) return;
    const uint32_t b = (uint32_t)(i / F), j = (uint32_t)(i % F);
    const float v = master[((size_t)b * C + (C - 1u)) * F + j];
    const int q = to_i24(v);
    uint8_t* o = (uint8_t*)dst + i * 3u;
    o[0] = (uint8_t)q;
    o[1] = (uint8_t)(q >> 8);
    o[2] = (uint8_t)(q >> 16);
    return;
  }
  const size_t total = (size_t)n_blocks * F * C;
  if (i >= total) return;
  const uint32_t c = (uint32_t)(i % C);
  const size_t frame = i / C;
  const uint32_t b = (uint32_t)(frame / F), j = (uint32_t)(frame % F);
  const float v = master[((size_t)b * C + c) * F + j];
  switch (fmt) {
    case 3: ((int16_t*)dst)[i] = (int16_t)x86_cvtt_f32(v > 0.0f ? __fmul_rn(v, 32767.0f) : __fmul_rn(v, 32768.0f)); break;
    case 6: {
      const int q = v > 0.0f ? x86_cvtt_f32(__fmul_rn(v, 8388607.0f)) : x86_cvtt_f32(__fmul_rn(v, 8388608.0f));
      ((int32_t*)dst)[i] = q & 0xFFFFFF;
      break;
    }
    case 7:
      ((int32_t*)dst)[i] = x86_cvtt_f64(v > 0.0f ? __dmul_rn((double)v, 2147483647.0) : __dmul_rn((double)v, 2147483648.0));
      break;
    default: ((float*)dst)[i] = v; break;
  }
}

// VUMeter::update's read of the running maxima (vu_meter.h:33: `level.exchange(0.0f)`): every level is exchanged with
// 0 and handed to the host (dst: pinned, device-mapped memory).  Runs on a stream of its own beside whatever mix is in
// flight — the mix kernel raises the same words with atomicMax, so a maximum that arrives after the exchange is seen by
// the next read, exactly like the reference's std::atomic<float>.
__global__ __launch_bounds__(256) void levels_take_kernel(uint32_t* levels, uint32_t* dst, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) dst[i] = atomicExch(levels + i, 0u);
}

// synthetic clip generator — same integer hash as whitebox_amd/synth.py (bench/test input only)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_kernel(void* dst, uint64_t frames, uint64_t key, float amp, int fmt) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= frames + kPad) return;
  const uint64_t u = splitmix64(key ^ i);
  const bool pad = i >= frames;
  switch (fmt) {
    case FMT_F32: {
      const float v = (float)((int64_t)(u >> 40) - (1 << 23)) * 1.1920928955078125e-07f;   // * 2^-23, exact
      ((float*)dst)[i] = pad ? 0.0f : __fmul_rn(v, amp);
      break;
    }
    case FMT_I16: ((int16_t*)dst)[i] = pad ? (int16_t)0 : (int16_t)((int64_t)(u >> 48) - 32768); break;
    case FMT_I24: ((int32_t*)dst)[i] = pad ? 0 : (int32_t)((int64_t)(u >> 40) - (1 << 23)); break;
    default: ((int32_t*)dst)[i] = pad ? 0 : (int32_t)((int64_t)(u >> 32) - (1ll << 31)); break;
  }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from wbx_runtime.hip)
// ------------------------------------------------------------------------------------------------
void launch_times_copy(const DBlockTime* host_pinned, DBlockTime* dev, uint32_t n_blocks, uint32_t* zero_counters, hipStream_t s) {
  const uint32_t n16 = n_blocks * (uint32_t)(sizeof(DBlockTime) / 16u);
  hipLaunchKernelGGL(times_copy_kernel, dim3((n16 + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<const uint4*>(host_pinned),
                     reinterpret_cast<uint4*>(dev), n16, zero_counters);
}

void launch_plan(const PlanArgs& a, hipStream_t s) {
  const uint32_t nb = (a.n_tracks + a.lanes - 1u) / a.lanes;
  static const bool roomy = [] { const char* v = std::getenv("WBX_PLAN_BESIDE"); return v && v[0] == '0'; }();   // A/B aid
  if (a.times && !roomy)
    hipLaunchKernelGGL(plan_kernel_beside, dim3(nb), dim3(64), 0, s, a);
  else
    hipLaunchKernelGGL(plan_kernel, dim3(nb), dim3(64), a.times ? 0 : a.n_blocks * sizeof(DBlockTime), s, a);
}

void launch_plan_segments(const PlanArgs& a, const SegArgs& g, bool beside, hipStream_t s) {
  const uint32_t n_groups = (a.n_tracks + 63u) / 64u;
  const dim3 grid(8u * ((n_groups + 7u) / 8u) * g.n_segs);   // (the kernel decodes: id mod 8 = track group mod 8)
  const size_t lds = (size_t)(g.seg_len + 2u * kSegMargin) * sizeof(DBlockTime);
  static const bool roomy = [] { const char* v = std::getenv("WBX_PLAN_BESIDE"); return v && v[0] == '0'; }();   // A/B aid
  if (beside && !roomy)
    hipLaunchKernelGGL(plan_seg_kernel_beside, grid, dim3(64), lds, s, a, g);
  else
    hipLaunchKernelGGL(plan_seg_kernel, grid, dim3(64), lds, s, a, g);
}

void launch_gen(const GenArgs& a, uint32_t max_grid, hipStream_t s) {
  // grid-stride over the queue: the grid only bounds the parallelism (a short render cannot queue many rows)
  const uint32_t wgs = (a.gen_cap + 3u) / 4u;   // one wave per queued record
  const uint32_t grid = wgs < max_grid ? (wgs ? wgs : 1u) : max_grid;
  hipLaunchKernelGGL(gen_kernel, dim3(grid * 4u), dim3(64), 0, s, a);   // (max_grid counts four-wave units)
}

// which instance of the hot kernel a render takes: the family's translation unit holds the instances
// (family 0: fp32 + integer PCM at unity speed; 1: everything; 2: sessions of 16-bit PCM only)
const char* launch_mix(const MixArgs& a, uint32_t n_blocks, int variant, int family, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  const uint32_t S4 = a.lane_span;          // (lanes per channel and block of the instance's lane space: F/4, or the next shape above it)
  const uint32_t lanes = a.channels * S4;   // lanes one block needs
  const bool full = (lanes % 256u == 0u) && (S4 % 64u == 0u);
  if (family == 3) return launch_mix_fam3(a, n_blocks, variant, s, t0, t1);   // (falls back to family 1 for shapes it has no instance for)
  // stereo 256-frame blocks with both channels per lane (families 0 and 2): one wave = one block
  if (variant >= 1000 && family != 1 && a.channels == 2u && S4 == 64u)
    return family == 2 ? launch_mix_fam2(a, n_blocks, variant, s, t0, t1) : launch_mix_fam0(a, n_blocks, variant, s, t0, t1);
  // blocks shorter than a workgroup: the short-block instances exist for the everything family and the lean fp32 one
  if (!full) return family != 0 ? launch_mix_fam1(a, n_blocks, s, t0, t1) : launch_mix_fam0(a, n_blocks, variant, s, t0, t1);
  if (family == 1) return launch_mix_fam1(a, n_blocks, s, t0, t1);
  if (family == 2) return launch_mix_fam2(a, n_blocks, variant, s, t0, t1);
  return launch_mix_fam0(a, n_blocks, variant, s, t0, t1);
}

// The spread sum's ticket barrier needs every workgroup of the grid resident at once: at most one per CU the process may use.
// A CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK) takes CUs away without the attribute saying so: no spreading then.  (A device
// shared with another process can still hold workgroups back; the barrier's wait is bounded, a give-up is reported and the
// block is mixed again through three launches — wbx_engine_process — and the context stops spreading.)
uint32_t callback_spread_limit() {
  static const uint32_t n_cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return (uint32_t)(n > 0 ? n : 0);
  }();
  static const bool no_spread = [] {
    const char* v = std::getenv("WBX_CB_SPREAD");   // A/B aid
    if (v && v[0] == '0') return true;
    for (const char* name : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK"}) {
      const char* m = std::getenv(name);
      if (m && m[0]) return true;
    }
    return false;
  }();
  return no_spread ? 0u : (n_cus < 256u ? n_cus : 256u);
}

const char* launch_callback(const MixArgs& m, const PlanArgs& p, const SumArgs& s0, uint32_t* done, uint32_t done_base, uint32_t done_base2, bool spread,
                            uint32_t* gave_up, uint32_t spin_bound, uint32_t* flag, uint32_t seq, int family, bool window_rows, unsigned long long* dbg, hipStream_t st) {
  SumArgs s = s0;
  s.n_blocks = 1u;
  const bool fenced = m.partial_through == 0u;   // WBX_CB_FENCED=1 (A/B aid), as the context read it
  // (the election word: word 1 of the counter block — the counters' own words are multiples of kCbStride)
  CallbackArgs cb{done, spread ? 1u : 0u, done_base, done_base2, done + 1, gave_up, flag, seq, m.n_groups, fenced ? 1u : 0u, spin_bound, dbg};
  if (family == 1 || family == 3) return launch_callback_fam1(m, p, s, cb, st);   // (3 = 1 without the per-frame taps: one instance serves both)
  if (family == 2) return launch_callback_fam2(m, p, s, cb, st);
  return launch_callback_fam0(m, p, s, cb, window_rows, st);
}

void launch_sum(const SumArgs& a0, uint32_t n_blocks, hipStream_t s) {
  SumArgs a = a0;
  a.n_blocks = n_blocks;
  // at most ~2048 waves (8 per CU) whatever the render length: see the kernel's block loop
  const uint32_t gx = n_blocks < kSumGridBlocks ? n_blocks : kSumGridBlocks;
  if (a.out_il) {   // interleaved device-format output: a lane owns 4 frames of every channel
    const uint32_t tiles = ((a.block_frames >> 2) + 63u) / 64u;
    if (a.n_buses != 0u)
      hipLaunchKernelGGL((sum_kernel<16, true, true>), dim3(gx, tiles), dim3(64), 0, s, a);
    else
      hipLaunchKernelGGL((sum_kernel<16, false, true>), dim3(gx, tiles), dim3(64), 0, s, a);
    return;
  }
  const uint32_t tiles = (((a.channels * a.block_frames) >> 2) + 63u) / 64u;
  if (a.n_buses != 0u)
    hipLaunchKernelGGL((sum_kernel<16, true>), dim3(gx, tiles), dim3(64), 0, s, a);
  else if (n_blocks < 8u && a.n_groups > 16u)
    hipLaunchKernelGGL((sum_kernel<32, false>), dim3(gx, tiles), dim3(64), 0, s, a);
  else
    hipLaunchKernelGGL((sum_kernel<16, false>), dim3(gx, tiles), dim3(64), 0, s, a);
}

void launch_clamp(float* buf, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(clamp_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, buf, n);
}

void launch_clamp_into(const float* src, float* dst, size_t n, int clamp, hipStream_t s) {
  hipLaunchKernelGGL(clamp_into_kernel, dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, s, src, dst, n, clamp);
}

void launch_ordered_add(const float* gathered, float* dst, size_t n, uint32_t world, int clamp, hipStream_t s) {
  hipLaunchKernelGGL(ordered_add_kernel, dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, s, gathered, dst, n, world, clamp);
}

void launch_convert(const float* master, void* dst, uint32_t n_blocks, uint32_t F, uint32_t C, int fmt, hipStream_t s) {
  const size_t total = (size_t)n_blocks * F * C;
  hipLaunchKernelGGL(convert_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, master, dst, n_blocks, F, C,
                     fmt);
}

void launch_levels_take(uint32_t* levels, uint32_t* dst, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(levels_take_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, levels, dst, n);
}

// The layout two hand-overs rest on — a launch's linear workgroup ids dealt round-robin to 8 / 4 / 2 / 1 XCDs, so that the
// 128-track pieces of a block (chained renders) and the segments of a track (plan_seg_kernel) meet behind ONE L2 — probed
// once per context: workgroup i of a small 1-D grid notes the XCD it runs on.  -> true when ids repeat with a period of
// 8, 4, 2 or 1 (MI355X: 8); anything else (a 6-XCD part, a partition mode that deals differently) and the context walks
// whole member lists and plans by one lane per track from the start, instead of finding out in its first long render.
__global__ __launch_bounds__(64) void xcc_probe_kernel(uint32_t* out) {
  if (threadIdx.x == 0u) out[blockIdx.x] = (uint32_t)__builtin_amdgcn_s_getreg(63508) & 15u;   // HW_REG_XCC_ID
}
bool probe_xcd_layout(hipStream_t s, uint32_t* n_xcds) {
  constexpr uint32_t kProbe = 64;
  uint32_t* d = nullptr;
  uint32_t h[kProbe];
  *n_xcds = 0;
  if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return false;
  hipLaunchKernelGGL(xcc_probe_kernel, dim3(kProbe), dim3(64), 0, s, d);
  const bool ok = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  (void)hipFree(d);
  if (!ok) return false;
  for (uint32_t period : {1u, 2u, 4u, 8u}) {
    bool fits = true, distinct = true;
    for (uint32_t i = period; i < kProbe && fits; i++) fits = h[i] == h[i - period];
    for (uint32_t i = 0; i < period && distinct; i++)
      for (uint32_t j = 0; j < i; j++) distinct = distinct && h[i] != h[j];
    if (fits && distinct) {
      *n_xcds = period;
      return true;
    }
  }
  return false;
}

void launch_synth(void* dst, uint64_t frames, uint64_t key, float amp, int fmt, hipStream_t s) {
  const uint64_t n = frames + kPad;
  hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, dst, frames, key, amp, fmt);
}

}  // namespace wbx
