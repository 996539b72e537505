// wbx_kernels.hip — gfx950 (CDNA4, wave64) kernels of the whitebox mix path.  HIP only, no other target.
//
//   plan_kernel      one lane per track: the reference's clip sequencer + sampler-state update for K
//                    consecutive blocks (wbx_seq.h), emitting one 64-B DTrackBlock per (block, track)
//   mix_kernel       the hot kernel: workgroup = (track group, block[, frame tile]); the group's
//                    DTrackBlock records (gain / pan / resample parameters) are staged in LDS, each
//                    lane owns 4 consecutive output frames of one channel, clip audio is streamed
//                    with 16-B loads, rendered, scaled and accumulated in registers IN TRACK ORDER;
//                    per-track peaks via wavefront shuffle-max + LDS
//   sum_kernel       group sums -> bus sums -> master (fixed order), master clamp
//   finalize/convert/levels/synth: small helpers
//
// HBM-bound integer/fp32 streaming: no MFMA (≈3 flop per 4 B).  Parity-critical arithmetic uses the
// explicitly rounded intrinsics (__fmul_rn, __dadd_rn, ...) and the file is built with
// -ffp-contract=off so that nothing is fused: the reference build has no FMA.
#include <hip/hip_runtime.h>

#include "wbx_dev.h"
#include "wbx_seq.h"

namespace wbx {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-B load at 4-B alignment

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void plan_kernel(PlanArgs a) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n_tracks) return;
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
  const uint32_t c0 = a.clip_first[t];
  const uint32_t nc = a.clip_first[t + 1] - c0;
  DClip* clips = const_cast<DClip*>(a.clips) + c0;
  for (uint32_t b = 0; b < a.n_blocks; b++) plan_track_block(a, t, b, &st, clips, nc);
  a.state[t] = st;
}

// ------------------------------------------------------------------------------------------------
// per-sample rendering helpers
// ------------------------------------------------------------------------------------------------

// reference math::clamp (core_math.h:33-37)
__device__ __forceinline__ float clampf(float x, float lo, float hi) {
  float m = x < hi ? x : hi;
  return m > lo ? m : lo;
}
__device__ __forceinline__ double clampd(double x, double lo, double hi) {
  double m = x < hi ? x : hi;
  return m > lo ? m : lo;
}

// One source sample of Sampler::stream for destination frame jj of segment sg, channel c:
// unity path sampler.cpp:106-158, linear path sampler.cpp:34-59 (normalisers :7-18 and :95-97).
__device__ __forceinline__ float sample_at(const DSeg& sg, uint32_t c, uint32_t jj) {
  const void* base = sg.src[c];
  if (sg.speed == 1.0) {
    const uint32_t idx = (uint32_t)sg.pos + jj;                          // :107
    switch (sg.format) {
      case FMT_F32: return ((const float*)base)[idx];
      case FMT_I16: {
        const float norm = 1.0f / 32767.0f;                              // :95
        return clampf(__fmul_rn((float)((const int16_t*)base)[idx], norm), -1.0f, 1.0f);
      }
      case FMT_I24: {
        const double norm = 1.0 / 8388607.0;                             // :96
        return (float)clampd(__dmul_rn((double)((const int32_t*)base)[idx], norm), -1.0, 1.0);
      }
      default: {
        const double norm = 1.0 / 2147483647.0;                          // :97
        return (float)clampd(__dmul_rn((double)((const int32_t*)base)[idx], norm), -1.0, 1.0);
      }
    }
  }
  const double x = __dadd_rn(sg.pos, __dmul_rn((double)(int32_t)jj, sg.speed));   // :50
  const long long ix = (long long)x;                                               // :51
  const float fx = (float)__dsub_rn(x, (double)ix);                                // :52
  float a, b;
  switch (sg.format) {
    case FMT_F32:
      a = ((const float*)base)[ix];
      b = ((const float*)base)[ix + 1];
      break;
    case FMT_I16: {
      const float norm = (float)(1.0 / 32767.0);                                   // :9-10
      a = __fmul_rn(norm, (float)((const int16_t*)base)[ix]);
      b = __fmul_rn(norm, (float)((const int16_t*)base)[ix + 1]);
      break;
    }
    case FMT_I24: {
      const double norm = 1.0 / 8388607.0;                                         // :11-12
      a = (float)__dmul_rn(norm, (double)((const int32_t*)base)[ix]);
      b = (float)__dmul_rn(norm, (double)((const int32_t*)base)[ix + 1]);
      break;
    }
    default: {
      const double norm = 1.0 / 2147483647.0;                                      // :13-14
      a = (float)__dmul_rn(norm, (double)((const int32_t*)base)[ix]);
      b = (float)__dmul_rn(norm, (double)((const int32_t*)base)[ix + 1]);
      break;
    }
  }
  return __fadd_rn(a, __fmul_rn(fx, __fsub_rn(b, a)));                             // :55
}

// Generic track-block: any number of segments, any coverage, any format.  Returns the track's
// mixing-buffer value for frame j of channel c BEFORE the track gain (the buffer the reference clears
// at engine.cpp:1602 and Sampler::stream accumulates into, sampler.cpp:56,152).
__device__ __forceinline__ float render_generic(const DTrackBlock& tb, const DSeg* pool, uint32_t c, uint32_t j) {
  float acc = 0.0f;
  const uint32_t nseg = tb.nseg;
  for (uint32_t s = 0; s < nseg; s++) {
    const DSeg& sg = (s == 0) ? tb.s0 : pool[(size_t)tb.extra * kChunk + (s - 1)];
    const uint32_t d0 = sg.dst_start, n = sg.len;
    if (j >= d0 && j < d0 + n) acc = __fadd_rn(acc, __fmul_rn(sample_at(sg, c, j - d0), sg.gain));
  }
  return acc;
}

__device__ __forceinline__ float pick(float w0, float w1, float w2, float w3, int k) {
  return k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3));
}

__device__ __forceinline__ float absmax4(f4 m) {
  return fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w)));
}

// ------------------------------------------------------------------------------------------------
// mix: grid = (n_groups, n_blocks, tiles), block = 256 lanes (4 waves).
// Lane -> (channel c, frames j0..j0+3).  With F = 512, C = 2: waves 0-1 own the left channel, waves
// 2-3 the right one, every wave-level load is one contiguous 1 KiB row of a clip.
// ------------------------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void mix_kernel(MixArgs a) {
  __shared__ __attribute__((aligned(16))) DTrackBlock s_tb[kStage];
  __shared__ uint32_t s_pk[kStage * 2];

  const uint32_t g = blockIdx.x, b = blockIdx.y, tile = blockIdx.z;
  const uint32_t tid = threadIdx.x;
  const DGroup grp = a.groups[g];
  const uint32_t F = a.block_frames, C = a.channels, N = a.n_tracks;
  const uint32_t S4 = F >> 2;
  const uint32_t slot = tile * 256u + tid;
  const bool active = slot < C * S4;
  const uint32_t c = active ? slot / S4 : 0u;
  const uint32_t j0 = active ? (slot - c * S4) * 4u : 0u;
  // lanes of an aligned `span`-lane group share a channel (span = largest power of two dividing F/4, <= 64)
  uint32_t span = S4 & (~S4 + 1u);
  span = span > 64u ? 64u : span;
  const uint32_t lane = tid & 63u;

  f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};

  for (uint32_t chunk0 = 0; chunk0 < grp.count; chunk0 += kStage) {
    const uint32_t cn = (grp.count - chunk0) < kStage ? (grp.count - chunk0) : kStage;
    __syncthreads();
    // stage the group's records (per-track gain / pan / resample parameters) in LDS: 4 x 16 B per record
    for (uint32_t i = tid; i < cn * 4u; i += 256u) {
      const uint32_t rec = i >> 2, q = i & 3u;
      const uint32_t track = a.order[grp.first + chunk0 + rec];
      reinterpret_cast<uint4*>(s_tb)[i] = reinterpret_cast<const uint4*>(a.tb + (size_t)b * N + track)[q];
    }
    if (tid < kStage * 2u) s_pk[tid] = 0u;
    __syncthreads();

    for (uint32_t u0 = 0; u0 < cn; u0 += U) {
      f4 v[U];
      float w4[U];
      int kind[U];
      // phase A: issue the clip loads of U tracks back to back
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t tl = u0 + u;
        int k = (tl < cn) ? (int)s_tb[tl].kind : (int)KIND_SILENT;
        k = __builtin_amdgcn_readfirstlane(k);
        kind[u] = k;
        v[u] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        w4[u] = 0.0f;
        if (active) {
          if (k == KIND_UNITY) {
            const DSeg& sg = s_tb[tl].s0;
            const float* p = (const float*)sg.src[c] + ((uint32_t)sg.pos + j0);      // sampler.cpp:107,151
            v[u] = *reinterpret_cast<const f4u*>(p);
          } else if (k == KIND_WINDOW) {
            const DSeg& sg = s_tb[tl].s0;
            const double x0 = __dadd_rn(sg.pos, __dmul_rn((double)(int32_t)j0, sg.speed));
            const float* p = (const float*)sg.src[c] + (int)trunc(x0);
            v[u] = *reinterpret_cast<const f4u*>(p);   // taps of 4 consecutive frames lie in p[0..4]
            w4[u] = p[4];
          }
        }
      }
      // phase B: render, scale, accumulate — strictly in track order
      float pk[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t tl = u0 + u;
        f4 m = {0.0f, 0.0f, 0.0f, 0.0f};
        const int k = kind[u];
        if (k != KIND_SILENT && active) {
          const DTrackBlock& tb = s_tb[tl];
          const float gc = tb.g[c];
          if (k == KIND_UNITY) {
            const float cg = tb.s0.gain;
            m.x = __fmul_rn(__fmul_rn(v[u].x, cg), gc);                             // sampler.cpp:152, track.cpp:731
            m.y = __fmul_rn(__fmul_rn(v[u].y, cg), gc);
            m.z = __fmul_rn(__fmul_rn(v[u].z, cg), gc);
            m.w = __fmul_rn(__fmul_rn(v[u].w, cg), gc);
          } else if (k == KIND_WINDOW) {
            const double pos = tb.s0.pos, speed = tb.s0.speed;
            const float cg = tb.s0.gain;
            const double x0 = __dadd_rn(pos, __dmul_rn((double)(int32_t)j0, speed));
            const int ix0 = (int)trunc(x0);
            float r[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const double x = __dadd_rn(pos, __dmul_rn((double)(int32_t)(j0 + e), speed));   // sampler.cpp:50
              const double tx = trunc(x);                                                    // :51 (x >= 0)
              const float fx = (float)__dsub_rn(x, tx);                                      // :52
              const int kk = (int)tx - ix0;                                                  // 0..e
              const float sa = pick(v[u].x, v[u].y, v[u].z, v[u].w, kk);
              const float sb = pick(v[u].y, v[u].z, v[u].w, w4[u], kk);
              const float s = __fadd_rn(sa, __fmul_rn(fx, __fsub_rn(sb, sa)));               // :55
              r[e] = __fmul_rn(__fmul_rn(s, cg), gc);                                        // :56, track.cpp:731
            }
            m = f4{r[0], r[1], r[2], r[3]};
          } else {
            m.x = __fmul_rn(render_generic(tb, a.pool, c, j0 + 0), gc);
            m.y = __fmul_rn(render_generic(tb, a.pool, c, j0 + 1), gc);
            m.z = __fmul_rn(render_generic(tb, a.pool, c, j0 + 2), gc);
            m.w = __fmul_rn(render_generic(tb, a.pool, c, j0 + 3), gc);
          }
          acc.x = __fadd_rn(acc.x, m.x);                                                      // audio_buffer.h:73-82
          acc.y = __fadd_rn(acc.y, m.y);
          acc.z = __fadd_rn(acc.z, m.z);
          acc.w = __fadd_rn(acc.w, m.w);
        }
        pk[u] = absmax4(m);                                                                   // vu_meter.h:20-25
      }
      // per-track peak: shuffle-max across the lanes that share a channel, then one LDS atomic per group
      for (uint32_t off = 1; off < span; off <<= 1) {
#pragma unroll
        for (int u = 0; u < U; u++) pk[u] = fmaxf(pk[u], __shfl_xor(pk[u], (int)off, 64));
      }
      if ((lane & (span - 1u)) == 0u && active) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint32_t tl = u0 + u;
          if (tl < cn && kind[u] != KIND_SILENT) atomicMax(&s_pk[tl * 2u + c], __float_as_uint(pk[u]));
        }
      }
    }
    __syncthreads();
    if (tid < cn * C) {
      const uint32_t rec = tid / C, ch = tid - rec * C;
      const uint32_t track = a.order[grp.first + chunk0 + rec];
      uint32_t* dst = reinterpret_cast<uint32_t*>(a.peaks) + ((size_t)b * N + track) * C + ch;
      if (a.tiles == 1u)
        *dst = s_pk[rec * 2u + ch];
      else
        atomicMax(dst, s_pk[rec * 2u + ch]);   // peaks are non-negative floats: uint order == float order
    }
  }

  if (active) {
    float* out = a.partial + (((size_t)b * a.n_groups + g) * C + c) * F + j0;
    *reinterpret_cast<f4*>(out) = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// sum: grid = (n_blocks, tiles), block = 256.  master = (((direct groups in order) + bus 0) + bus 1) + ...
// with bus u = in-order sum of its groups (AudioBuffer::mix order, audio_buffer.h:73-82), then the
// clamp of engine.cpp:1627-1636.  Groups arrive sorted: direct ones first, then by bus.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_kernel(SumArgs a) {
  const uint32_t b = blockIdx.x;
  const uint32_t F = a.block_frames, C = a.channels;
  const uint32_t slot = blockIdx.y * 256u + threadIdx.x;
  if (slot >= (C * F) >> 2) return;
  const size_t e0 = (size_t)slot * 4u;
  const size_t stride = (size_t)C * F;
  const float* p = a.partial + (size_t)b * a.n_groups * stride + e0;

  f4 master = {0.0f, 0.0f, 0.0f, 0.0f};
  f4 busacc = {0.0f, 0.0f, 0.0f, 0.0f};
  int cur = -1;
  constexpr int PF = 8;
  for (uint32_t g0 = 0; g0 < a.n_groups; g0 += PF) {
    f4 v[PF];
#pragma unroll
    for (int i = 0; i < PF; i++)
      if (g0 + i < a.n_groups) v[i] = *reinterpret_cast<const f4*>(p + (size_t)(g0 + i) * stride);
#pragma unroll
    for (int i = 0; i < PF; i++) {
      if (g0 + i >= a.n_groups) break;
      const int bus = a.groups[g0 + i].bus;
      if (bus != cur) {
        if (cur >= 0) {
          if (a.buses) *reinterpret_cast<f4*>(a.buses + ((size_t)b * a.n_buses + cur) * stride + e0) = busacc;
          master.x = __fadd_rn(master.x, busacc.x);
          master.y = __fadd_rn(master.y, busacc.y);
          master.z = __fadd_rn(master.z, busacc.z);
          master.w = __fadd_rn(master.w, busacc.w);
        }
        busacc = f4{0.0f, 0.0f, 0.0f, 0.0f};
        cur = bus;
      }
      if (bus < 0) {
        master.x = __fadd_rn(master.x, v[i].x);
        master.y = __fadd_rn(master.y, v[i].y);
        master.z = __fadd_rn(master.z, v[i].z);
        master.w = __fadd_rn(master.w, v[i].w);
      } else {
        busacc.x = __fadd_rn(busacc.x, v[i].x);
        busacc.y = __fadd_rn(busacc.y, v[i].y);
        busacc.z = __fadd_rn(busacc.z, v[i].z);
        busacc.w = __fadd_rn(busacc.w, v[i].w);
      }
    }
  }
  if (cur >= 0) {
    if (a.buses) *reinterpret_cast<f4*>(a.buses + ((size_t)b * a.n_buses + cur) * stride + e0) = busacc;
    master.x = __fadd_rn(master.x, busacc.x);
    master.y = __fadd_rn(master.y, busacc.y);
    master.z = __fadd_rn(master.z, busacc.z);
    master.w = __fadd_rn(master.w, busacc.w);
  }
  if (a.clamp) {   // engine.cpp:1627-1636: compare, don't min/max (NaN passes through unchanged)
    master.x = master.x > 1.0f ? 1.0f : (master.x < -1.0f ? -1.0f : master.x);
    master.y = master.y > 1.0f ? 1.0f : (master.y < -1.0f ? -1.0f : master.y);
    master.z = master.z > 1.0f ? 1.0f : (master.z < -1.0f ? -1.0f : master.z);
    master.w = master.w > 1.0f ? 1.0f : (master.w < -1.0f ? -1.0f : master.w);
  }
  *reinterpret_cast<f4*>(a.master + (size_t)b * stride + e0) = master;
}

// buses with no member groups stay zero: cleared before the launch by the runtime.

// the clamp alone, over a device buffer (root rank after the RCCL reduce)
__global__ __launch_bounds__(256) void clamp_kernel(float* buf, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float v = buf[i];
  buf[i] = v > 1.0f ? 1.0f : (v < -1.0f ? -1.0f : v);
}

// VUMeter::level semantics: running maximum (vu_meter.h:26-29); levels[t][c] = max(levels, max_b peaks[b][t][c])
__global__ __launch_bounds__(256) void levels_kernel(const float* peaks, float* levels, uint32_t n_blocks, uint32_t nc) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= nc) return;
  float m = levels[i];
  for (uint32_t b = 0; b < n_blocks; b++) {
    const float p = peaks[(size_t)b * nc + i];
    m = m < p ? p : m;
  }
  levels[i] = m;
}

// planar fp32 master [K][C][F] -> interleaved device-format samples [K*F][C]; reference
// core/audio_format_conv.cpp:5-20 (i16), :45-60 (i24 in 32-bit containers), :62-77 (i32), :79-91 (f32)
__global__ __launch_bounds__(256) void convert_kernel(const float* master, void* dst, uint32_t n_blocks, uint32_t F,
                                                      uint32_t C, int fmt) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;   // index into [K*F][C]
  const size_t total = (size_t)n_blocks * F * C;
  if (i >= total) return;
  const uint32_t c = (uint32_t)(i % C);
  const size_t frame = i / C;
  const uint32_t b = (uint32_t)(frame / F), j = (uint32_t)(frame % F);
  const float v = master[((size_t)b * C + c) * F + j];
  switch (fmt) {
    case 3: ((int16_t*)dst)[i] = (int16_t)(int)(v > 0.0f ? __fmul_rn(v, 32767.0f) : __fmul_rn(v, 32768.0f)); break;
    case 6: {
      const int q = v > 0.0f ? (int)__fmul_rn(v, 8388607.0f) : (int)__fmul_rn(v, 8388608.0f);
      ((int32_t*)dst)[i] = q & 0xFFFFFF;
      break;
    }
    case 7:
      ((int32_t*)dst)[i] = (int32_t)(v > 0.0f ? __dmul_rn((double)v, 2147483647.0) : __dmul_rn((double)v, 2147483648.0));
      break;
    default: ((float*)dst)[i] = v; break;
  }
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
void launch_plan(const PlanArgs& a, hipStream_t s) {
  const uint32_t nb = (a.n_tracks + 63u) / 64u;
  hipLaunchKernelGGL(plan_kernel, dim3(nb), dim3(64), 0, s, a);
}

void launch_mix(const MixArgs& a, uint32_t n_blocks, hipStream_t s) {
  hipLaunchKernelGGL(mix_kernel<8>, dim3(a.n_groups, n_blocks, a.tiles), dim3(256), 0, s, a);
}

void launch_sum(const SumArgs& a, uint32_t n_blocks, hipStream_t s) {
  const uint32_t tiles = (((a.channels * a.block_frames) >> 2) + 255u) / 256u;
  hipLaunchKernelGGL(sum_kernel, dim3(n_blocks, tiles), dim3(256), 0, s, a);
}

void launch_clamp(float* buf, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(clamp_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, buf, n);
}

void launch_levels(const float* peaks, float* levels, uint32_t n_blocks, uint32_t nc, hipStream_t s) {
  hipLaunchKernelGGL(levels_kernel, dim3((nc + 255u) / 256u), dim3(256), 0, s, peaks, levels, n_blocks, nc);
}

void launch_convert(const float* master, void* dst, uint32_t n_blocks, uint32_t F, uint32_t C, int fmt, hipStream_t s) {
  const size_t total = (size_t)n_blocks * F * C;
  hipLaunchKernelGGL(convert_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, master, dst, n_blocks, F, C,
                     fmt);
}

void launch_synth(void* dst, uint64_t frames, uint64_t key, float amp, int fmt, hipStream_t s) {
  const uint64_t n = frames + kPad;
  hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, dst, frames, key, amp, fmt);
}

}  // namespace wbx
