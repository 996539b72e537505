// wbx_sum.h — group sums -> bus sums -> master for ONE block: the body of sum_kernel (wbx_kernels.hip) and of the one-launch
// callback's last workgroup (wbx_callback.h).  master = (((direct groups in order) + bus 0) + bus 1) + ... with bus u = the
// in-order sum of its groups (AudioBuffer::mix order, audio_buffer.h:73-82), then the clamp of engine.cpp:1627-1636.
// Groups arrive sorted: direct ones first, then by bus.
#pragma once
#include <type_traits>

#include "wbx_mix.h"

namespace wbx {

// float -> int32 as the reference's x86 build converts (cvttss2si / cvttsd2si): truncation toward zero, and the
// "integer indefinite" 0x80000000 for NaN and for anything outside [-2^31, 2^31) — the GPU's own conversion saturates
// and maps NaN to 0, which differs whenever the master is left un-clamped or holds NaN.
__device__ __forceinline__ int x86_cvtt_f32(float t) { return (t >= -2147483648.0f && t < 2147483648.0f) ? (int)t : (int)0x80000000; }
__device__ __forceinline__ int x86_cvtt_f64(double t) { return (t >= -2147483648.0 && t < 2147483648.0) ? (int)t : (int)0x80000000; }

// One sample of the master in an interleaved device format (core/audio_format_conv.cpp:5-20 i16, :45-60 i24 in 32-bit
// containers, :62-77 i32): asymmetric scales, truncation toward zero, the x86 conversion results.
__device__ __forceinline__ int to_i16(float v) { return x86_cvtt_f32(v > 0.0f ? __fmul_rn(v, 32767.0f) : __fmul_rn(v, 32768.0f)); }
__device__ __forceinline__ int to_i24(float v) { return v > 0.0f ? x86_cvtt_f32(__fmul_rn(v, 8388607.0f)) : x86_cvtt_f32(__fmul_rn(v, 8388608.0f)); }
__device__ __forceinline__ int to_i32(float v) { return x86_cvtt_f64(v > 0.0f ? __dmul_rn((double)v, 2147483647.0) : __dmul_rn((double)v, 2147483648.0)); }

// PF = group partials in flight per lane: 16 for batch renders (the kernel runs beside the next mix and must stay small),
// 32 for the one-block callback, whose sum is a chain of dependent HBM round trips — 128 groups are four of them, not eight
// IL: the master leaves as INTERLEAVED device-format samples (SumArgs::out_format: what the audio back end hands the
// device, audio_io_pulseaudio.cpp:419-461 -> AudioBuffer::interleave_samples_to -> core/audio_format_conv.cpp) instead
// of planar fp32 — the conversion is the epilogue of the sum, no separate launch and no planar round trip.  A lane then
// owns 4 frames of EVERY channel (grid.y covers F/4 slots).
// sum_block: the lane's slot (4 frames of one channel; IL: of every channel) of block b
// SYS: the master goes to pinned host memory and the host learns of it from a flag written inside the same launch (the
// one-launch callback): system-scope stores (store_f4_system, wbx_mix.h) — acknowledged when really on their way
template <bool SYS>
__device__ __forceinline__ void store16(void* p, const uint4& v) {
  if constexpr (SYS)
    store_u4_system(p, v);
  else
    *reinterpret_cast<uint4*>(p) = v;
}

// where sum_block finds group g's sum of elements e0 .. e0+3 of block b: in the partial buffer ...
struct PartialFromMemory {
  const float* base;   // the block's [n_groups][C][F]
  size_t stride;
  __device__ __forceinline__ f4 operator()(size_t e0, uint32_t g) const { return *reinterpret_cast<const f4*>(base + e0 + (size_t)g * stride); }
};
// ... or in LDS, where the workgroup's lanes have put them side by side with ONE load each (the one-launch callback: the 256
// group sums of a 4096-track block are one memory round trip instead of one per PF of them).  row[c]: the sums of the
// lane's slot for elements below / from F on (interleaved output: the two channels of the slot's frames), [n_groups] each
struct PartialFromLds {
  const f4* row[2];
  uint32_t split;      // elements from here on come from row[1]
  __device__ __forceinline__ f4 operator()(size_t e0, uint32_t g) const { return row[e0 >= split ? 1 : 0][g]; }
};

template <int PF, bool BUSES, bool IL, bool SYS = false, class LOAD = PartialFromMemory>
__device__ __forceinline__ void sum_block(const SumArgs& a, uint32_t b, uint32_t slot, const LOAD* from = nullptr) {
  const uint32_t F = a.block_frames, C = a.channels;
  const size_t stride = (size_t)C * F;
  const PartialFromMemory mem{a.partial + (size_t)b * a.n_groups * stride, stride};
  auto load = [&](size_t e0, uint32_t g) {
    if constexpr (std::is_same<LOAD, PartialFromMemory>::value)
      return mem(e0, g);
    else
      return (*from)(e0, g);
  };

  // the master of elements e0 .. e0+3 of the block ([C][F] order): groups in order, buses in order, clamp
  auto sum_at = [&](size_t e0) {
  f4 master = {0.0f, 0.0f, 0.0f, 0.0f};
  f4 busacc = {0.0f, 0.0f, 0.0f, 0.0f};
  int cur = -1;
  if constexpr (!BUSES) {
    // no sub-buses (the reference's own topology): every group goes straight into the master, in order — nothing but
    // the loads, PF of them in flight, and the adds.  (Chained render: the pieces before the last hold intermediate
    // running sums; the last one holds THE sum.)
    for (uint32_t g0 = a.chain ? a.n_groups - 1u : 0u; g0 < a.n_groups; g0 += PF) {
      f4 v[PF];
#pragma unroll
      for (int i = 0; i < PF; i++) {
        const uint32_t g = g0 + i < a.n_groups ? g0 + i : a.n_groups - 1u;   // (clamped: straight-line loads)
        v[i] = load(e0, g);
      }
      __builtin_amdgcn_sched_barrier(0);   // all PF loads are issued before the first add waits for one
#pragma unroll
      for (int i = 0; i < PF; i++) {
        if (g0 + i < a.n_groups) {   // (a predicate, not a break: the unrolled array must stay in registers)
          master.x = __fadd_rn(master.x, v[i].x);
          master.y = __fadd_rn(master.y, v[i].y);
          master.z = __fadd_rn(master.z, v[i].z);
          master.w = __fadd_rn(master.w, v[i].w);
        }
      }
    }
  } else
  for (uint32_t g0 = 0; g0 < a.n_groups; g0 += PF) {
    f4 v[PF];
#pragma unroll
    for (int i = 0; i < PF; i++)
      if (g0 + i < a.n_groups) v[i] = load(e0, g0 + i);
#pragma unroll
    for (int i = 0; i < PF; i++) {
      if (g0 + i >= a.n_groups) break;
      if (a.chain && (a.groups[g0 + i].flags & GROUP_CHAIN_OUT)) continue;   // an intermediate running sum of a chained list
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
  return master;
  };

  if constexpr (!IL) {
    const size_t e0 = (size_t)slot * 4u;
    const f4 mv = sum_at(e0);
    store16<SYS>(a.master + (size_t)b * stride + e0, uint4{__float_as_uint(mv.x), __float_as_uint(mv.y), __float_as_uint(mv.z), __float_as_uint(mv.w)});
  } else {
    const uint32_t j0 = slot * 4u;
    f4 m[2];
    m[0] = sum_at(j0);
    m[1] = C > 1u ? sum_at((size_t)F + j0) : m[0];
    const float s[2][4] = {{m[0].x, m[0].y, m[0].z, m[0].w}, {m[1].x, m[1].y, m[1].z, m[1].w}};
    const size_t f0 = (size_t)b * F + j0;   // first frame of the lane in the whole render
    // (stereo: the lane's 4 frames x 2 channels leave as one or two 16-byte stores — the destination is usually pinned host
    //  memory, where narrow stores waste the PCIe write path)
    uint32_t w[8];   // the 8 interleaved samples of a stereo lane as 32-bit words (i16: packed in pairs into w[0..3])
    const uint32_t fmt = a.out_format;
    if (fmt == 5u) {   // packed 24-bit: the reference's writer has no channel term in its destination index (audio_format_conv.cpp:
      // 22-43), so a block's region of F*C*3 bytes holds the LAST channel's samples in its first 3*F bytes; the rest is
      // never written.  12 bytes per lane: three dword stores.
      const int q0 = to_i24(s[1][0]), q1 = to_i24(s[1][1]), q2 = to_i24(s[1][2]), q3 = to_i24(s[1][3]);   // (s[1] = the last channel, also for mono)
      uint32_t* o = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(a.out_il) + (size_t)b * F * C * 3u + (size_t)j0 * 3u);
      const uint32_t o0 = ((uint32_t)q0 & 0xFFFFFFu) | ((uint32_t)q1 << 24), o1 = (((uint32_t)q1 >> 8) & 0xFFFFu) | ((uint32_t)q2 << 16),
                     o2 = (((uint32_t)q2 >> 16) & 0xFFu) | ((uint32_t)q3 << 8);
      if constexpr (SYS) {
        __hip_atomic_store(o + 0, o0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(o + 1, o1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(o + 2, o2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        o[0] = o0;
        o[1] = o1;
        o[2] = o2;
      }
      return;
    }
    auto conv = [&](float v) -> uint32_t {
      return fmt == 3u ? (uint32_t)(uint16_t)(int16_t)to_i16(v) : fmt == 6u ? (uint32_t)(to_i24(v) & 0xFFFFFF)
             : fmt == 7u ? (uint32_t)to_i32(v) : __float_as_uint(v);
    };
    if (C == 2u) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        w[2 * k] = conv(s[0][k]);
        w[2 * k + 1] = conv(s[1][k]);
      }
      if (fmt == 3u) {
        uint4 o = {w[0] | (w[1] << 16), w[2] | (w[3] << 16), w[4] | (w[5] << 16), w[6] | (w[7] << 16)};
        store16<SYS>(reinterpret_cast<int16_t*>(a.out_il) + f0 * 2u, o);
      } else {
        uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(a.out_il) + f0 * 2u);
        store16<SYS>(o, uint4{w[0], w[1], w[2], w[3]});
        store16<SYS>(o + 1, uint4{w[4], w[5], w[6], w[7]});
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) w[k] = conv(s[0][k]);
      if (fmt == 3u) {
        uint32_t* o = reinterpret_cast<uint32_t*>(reinterpret_cast<int16_t*>(a.out_il) + f0);
        if constexpr (SYS) {
          __hip_atomic_store(o + 0, w[0] | (w[1] << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(o + 1, w[2] | (w[3] << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
          *reinterpret_cast<uint2*>(o) = uint2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
        }
      } else {
        store16<SYS>(reinterpret_cast<uint32_t*>(a.out_il) + f0, uint4{w[0], w[1], w[2], w[3]});
      }
    }
  }
}

}  // namespace wbx
