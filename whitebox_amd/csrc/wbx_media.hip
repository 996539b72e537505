// wbx_media.hip — the two steps either side of the mix path that touch the same resident clip audio
// (SURVEY 8(f) rows 3 and 4), gfx950 only:
//
//   deinterleave_kernel   clip ingest: interleaved decoder frames -> channel-planar clip storage in HBM
//                         (deinterleave_samples<T>, dsp/sample.cpp:29-43, as used by Sample::load_file :112-197)
//   mip_tile_kernel       waveform peak mip-maps, levels 0..5 (chunk 2 .. 2048 samples) in ONE pass over the clip
//   mip_upper_kernel      levels 6.. from the per-tile summaries
//                         (summarize_for_mipmaps_impl<T> / WaveformVisual::create, gfx/waveform_visual.cpp:9-246)
//
// Both are HBM-bound byte/integer work: coalesced 16-B accesses, no MFMA.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "wbx_dev.h"

namespace wbx {

// ------------------------------------------------------------------------------------------------
// ingest.  One lane = 4 consecutive frames: a 4*C-element contiguous read, one 4-element store per channel.
// ------------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ __launch_bounds__(256) void deinterleave_kernel(const T* __restrict__ src, T* __restrict__ dst0,
                                                           T* __restrict__ dst1, uint64_t frames) {
  const uint64_t f0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;
  if (f0 >= frames) return;
  T* const dst[2] = {dst0, dst1};
  if (f0 + 4u <= frames) {
    T v[4 * C];
    typedef T vec_in __attribute__((ext_vector_type(4 * C)));
    typedef T vec_out __attribute__((ext_vector_type(4)));
    // the interleaved source is 16-byte aligned (checked by the caller); a 32-byte lane read is two 16-B loads
    typedef vec_in vec_in_u __attribute__((aligned(sizeof(T) * 4 * C < 16 ? sizeof(T) * 4 * C : 16)));
    const vec_in_u in = *reinterpret_cast<const vec_in_u*>(src + f0 * C);
#pragma unroll
    for (int i = 0; i < 4 * C; i++) v[i] = in[i];
#pragma unroll
    for (int c = 0; c < C; c++) {
      vec_out o;
#pragma unroll
      for (int j = 0; j < 4; j++) o[j] = v[j * C + c];                                     // sample.cpp:39
      *reinterpret_cast<vec_out*>(dst[c] + f0) = o;   // channel rows are 256-B aligned and f0 is a multiple of 4
    }
  } else {
    for (uint64_t f = f0; f < frames; f++)
      for (int c = 0; c < C; c++) dst[c][f] = src[f * C + c];
  }
}

void launch_deinterleave(const void* src, void* dst0, void* dst1, uint64_t frames, uint32_t channels, uint32_t elem,
                         hipStream_t s) {
  if (!frames) return;
  const dim3 grid((uint32_t)((frames + 1023u) / 1024u)), block(256);
#define WBX_DI(T, C) hipLaunchKernelGGL((deinterleave_kernel<T, C>), grid, block, 0, s, (const T*)src, (T*)dst0, (T*)dst1, frames)
  if (elem == 2) {
    if (channels == 1) WBX_DI(uint16_t, 1); else WBX_DI(uint16_t, 2);
  } else {
    if (channels == 1) WBX_DI(uint32_t, 1); else WBX_DI(uint32_t, 2);
  }
#undef WBX_DI
}

// ------------------------------------------------------------------------------------------------
// waveform mip-maps.
//
// Level l (l = 0, 1, ...) summarises chunks of 2^(2l+1) samples as an ordered (first, second) pair of the chunk's
// minimum and maximum, "first" being the one that occurs earlier (waveform_visual.cpp:46-52).  A chunk's summary
// is the ordered merge of its four quarter-chunks' summaries, so all levels come out of one read of the clip:
//   node = (mn, mx, ord) with ord = 1 when the first occurrence of mx precedes the first occurrence of mn.
// Merging children left to right keeps exactly the reference's first-occurrence rule (strict < and > in its
// scan).  Nodes are built from the samples that exist (chunks at the clip's end are shorter), and a level's pair is
// STORED only when the reference's mip_data_count for that level includes it (:197-199) — the reference drops a
// trailing partial chunk at some levels and keeps it at others.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ MipNode mip_leaf(int v) { return MipNode{v, v, 0, 0}; }
__device__ __forceinline__ MipNode mip_empty() { return MipNode{0, 0, 0, 1}; }

// a ++ b (b follows a in time)
__device__ __forceinline__ MipNode mip_merge(const MipNode& a, const MipNode& b) {
  if (b.empty) return a;
  if (a.empty) return b;
  MipNode r;
  r.empty = 0;
  const bool min_in_b = b.mn < a.mn;    // first occurrence of the minimum lies in b only if b is strictly lower
  const bool max_in_b = b.mx > a.mx;
  r.mn = min_in_b ? b.mn : a.mn;
  r.mx = max_in_b ? b.mx : a.mx;
  if (min_in_b == max_in_b)
    r.ord = min_in_b ? b.ord : a.ord;   // both in the same child: that child's order
  else
    r.ord = min_in_b ? 1 : 0;           // max in a, min in b: maximum first; min in a, max in b: minimum first
  return r;
}

// (T)conv as the reference's x86-64 build evaluates it: cvttss2si/cvttsd2si (integer indefinite 0x80000000 for NaN
// and out-of-range), then the low bits as T
template <int BITS>
__device__ __forceinline__ int mip_narrow(int v) {
  return BITS == 8 ? (int)(int8_t)v : (int)(int16_t)v;
}
template <int BITS>
__device__ __forceinline__ int mip_from_f32(float conv) {
  const int i = (conv > -2147483904.0f && conv < 2147483648.0f) ? (int)conv : (int)0x80000000;
  return mip_narrow<BITS>(i);
}
template <int BITS>
__device__ __forceinline__ int mip_from_f64(double conv) {
  const int i = (conv > -2147483649.0 && conv < 2147483648.0) ? (int)conv : (int)0x80000000;
  return mip_narrow<BITS>(i);
}

// one sample -> T (waveform_visual.cpp:69-76 I16, :109-116 I32, :147-150 F32)
template <int FMT, int BITS>
__device__ __forceinline__ int mip_convert(uint32_t raw) {
  constexpr double tmin = BITS == 8 ? -128.0 : -32768.0, tmax = BITS == 8 ? 127.0 : 32767.0;
  if (FMT == FMT_I16) {
    const int s = (int)(int16_t)raw;
    constexpr float kmin = (float)tmin / -32768.0f, kmax = (float)tmax / 32767.0f;
    return mip_from_f32<BITS>(__fmul_rn((float)s, s >= 0 ? kmax : kmin));
  } else if (FMT == FMT_F32) {
    const float x = __uint_as_float(raw);
    return mip_from_f32<BITS>(__fmul_rn(x, x >= 0.0f ? (float)tmax : (float)-tmin));
  } else {
    const int s = (int)raw;
    constexpr double kmin = tmin / -2147483648.0, kmax = tmax / 2147483647.0;
    return mip_from_f64<BITS>(__dmul_rn((double)s, s >= 0 ? kmax : kmin));
  }
}

template <typename OT>
__device__ __forceinline__ void mip_store(OT* out, uint64_t pair, uint64_t data_count, const MipNode& n) {
  if (n.empty || 2u * pair + 1u >= data_count) return;
  typedef OT pair_t __attribute__((ext_vector_type(2)));
  pair_t p;
  p.x = (OT)(n.ord ? n.mx : n.mn);
  p.y = (OT)(n.ord ? n.mn : n.mx);
  *reinterpret_cast<pair_t*>(out + 2u * pair) = p;
}

template <int FMT, int BITS>
__global__ __launch_bounds__(256) void mip_tile_kernel(MipArgs a) {
  typedef typename std::conditional<BITS == 8, int8_t, int16_t>::type OT;
  __shared__ MipNode s_nodes[256];
  const uint32_t tid = threadIdx.x;
  const uint64_t tile = blockIdx.x;
  const uint64_t s0 = tile * kMipTile + (uint64_t)tid * 8u;

  // 8 consecutive samples per lane (16 B of 16-bit PCM, 32 B otherwise)
  uint32_t raw[8];
  const bool whole = s0 + 8u <= a.count;
  if (FMT == FMT_I16) {
    const uint16_t* p = (const uint16_t*)a.src + s0;
    if (whole) {
      const uint4 w = *reinterpret_cast<const uint4*>(p);
      raw[0] = w.x & 0xFFFFu; raw[1] = w.x >> 16; raw[2] = w.y & 0xFFFFu; raw[3] = w.y >> 16;
      raw[4] = w.z & 0xFFFFu; raw[5] = w.z >> 16; raw[6] = w.w & 0xFFFFu; raw[7] = w.w >> 16;
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) raw[i] = s0 + i < a.count ? p[i] : 0u;
    }
  } else {
    const uint32_t* p = (const uint32_t*)a.src + s0;
    if (whole) {
      const uint4 w0 = *reinterpret_cast<const uint4*>(p), w1 = *reinterpret_cast<const uint4*>(p + 4);
      raw[0] = w0.x; raw[1] = w0.y; raw[2] = w0.z; raw[3] = w0.w;
      raw[4] = w1.x; raw[5] = w1.y; raw[6] = w1.z; raw[7] = w1.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) raw[i] = s0 + i < a.count ? p[i] : 0u;
    }
  }

  // level 0: chunks of 2 samples, 4 per lane
  MipNode l0[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint64_t s = s0 + 2u * q;
    MipNode n = mip_empty();
    if (s < a.count) n = mip_leaf(mip_convert<FMT, BITS>(raw[2 * q]));
    if (s + 1u < a.count) n = mip_merge(n, mip_leaf(mip_convert<FMT, BITS>(raw[2 * q + 1])));
    l0[q] = n;
  }
  if (whole && s0 + 8u <= a.data_count[0]) {   // the lane's 4 pairs as one 8-B / 16-B store
    typedef OT oct_t __attribute__((ext_vector_type(8)));
    oct_t o;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      o[2 * q] = (OT)(l0[q].ord ? l0[q].mx : l0[q].mn);
      o[2 * q + 1] = (OT)(l0[q].ord ? l0[q].mn : l0[q].mx);
    }
    *reinterpret_cast<oct_t*>((OT*)a.level_out[0] + s0) = o;
  } else {
#pragma unroll
    for (int q = 0; q < 4; q++) mip_store<OT>((OT*)a.level_out[0], (s0 >> 1) + q, a.data_count[0], l0[q]);
  }
  // level 1: chunk of 8 = this lane's 4 level-0 nodes
  MipNode n = mip_merge(mip_merge(l0[0], l0[1]), mip_merge(l0[2], l0[3]));
  if (a.n_levels > 1) mip_store<OT>((OT*)a.level_out[1], s0 >> 3, a.data_count[1], n);

  // levels 2..5: 4-to-1 merges through LDS (64, 16, 4, 1 nodes per tile)
  s_nodes[tid] = n;
  uint32_t width = 256;
#pragma unroll
  for (uint32_t lvl = 2; lvl < kMipTileLevels; lvl++) {
    __syncthreads();
    width >>= 2;
    MipNode m = mip_empty();
    if (tid < width) {
      m = mip_merge(mip_merge(s_nodes[4 * tid], s_nodes[4 * tid + 1]), mip_merge(s_nodes[4 * tid + 2], s_nodes[4 * tid + 3]));
      if (lvl < a.n_levels) mip_store<OT>((OT*)a.level_out[lvl], tile * width + tid, a.data_count[lvl], m);
    }
    __syncthreads();
    if (tid < width) s_nodes[tid] = m;
  }
  __syncthreads();
  if (tid == 0) a.tile_nodes[tile] = s_nodes[0];
}

// levels >= 6 from the per-tile nodes: ONE workgroup of 1024 lanes walks up the tree, four nodes into one per
// level, ping-ponging between the two halves of the node scratch (a few thousand nodes per clip — latency, not
// bandwidth; a chain of 4^(l-5) dependent loads per output, the obvious formulation, costs 10x more).
template <int BITS>
__global__ __launch_bounds__(1024) void mip_upper_kernel(MipArgs a) {
  typedef typename std::conditional<BITS == 8, int8_t, int16_t>::type OT;
  MipNode* cur = a.tile_nodes;
  MipNode* nxt = a.tile_nodes + a.n_tiles;
  uint64_t n = a.n_tiles;
  for (uint32_t lvl = kMipTileLevels; lvl < a.n_levels; lvl++) {
    const uint64_t m = (n + 3u) >> 2;
    for (uint64_t g = threadIdx.x; g < m; g += 1024u) {
      MipNode r = cur[4 * g];
      if (4 * g + 1 < n) r = mip_merge(r, cur[4 * g + 1]);
      if (4 * g + 2 < n) r = mip_merge(r, cur[4 * g + 2]);
      if (4 * g + 3 < n) r = mip_merge(r, cur[4 * g + 3]);
      nxt[g] = r;
      mip_store<OT>((OT*)a.level_out[lvl], g, a.data_count[lvl], r);
    }
    __threadfence_block();
    __syncthreads();
    MipNode* t = cur;
    cur = nxt;
    nxt = t;
    n = m;
  }
}

void launch_mip(const MipArgs& a, int format, int bits, hipStream_t s) {
  if (!a.n_levels || !a.n_tiles) return;
  const dim3 grid(a.n_tiles), block(256);
#define WBX_MIP(F, B) hipLaunchKernelGGL((mip_tile_kernel<F, B>), grid, block, 0, s, a)
  if (format == FMT_I16) {
    if (bits == 8) WBX_MIP(FMT_I16, 8); else WBX_MIP(FMT_I16, 16);
  } else if (format == FMT_F32) {
    if (bits == 8) WBX_MIP(FMT_F32, 8); else WBX_MIP(FMT_F32, 16);
  } else {
    if (bits == 8) WBX_MIP(FMT_I32, 8); else WBX_MIP(FMT_I32, 16);
  }
#undef WBX_MIP
  if (a.n_levels > kMipTileLevels) {
    if (bits == 8)
      hipLaunchKernelGGL((mip_upper_kernel<8>), dim3(1), dim3(1024), 0, s, a);
    else
      hipLaunchKernelGGL((mip_upper_kernel<16>), dim3(1), dim3(1024), 0, s, a);
  }
}

}  // namespace wbx
