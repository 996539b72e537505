// wbx_mix.h — the hot kernel of the mix path, mix_kernel<U, FULL, W, FAM, SB, CW, CL, T>, and what it is made of.  A header:
// the instances are compiled by family (FAM) in translation units of their own — wbx_mix_fam0.hip (fp32 + integer PCM at
// unity speed), wbx_mix_fam1.hip (everything), wbx_mix_fam2.hip (the 16-bit family) — so that a family can grow modes
// without touching the others' register budgets or build time; wbx_kernels.hip keeps the small kernels and the
// dispatcher (launch_mix).  gfx950 (CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>

#include "wbx_dev.h"

namespace wbx {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-B load at 4-B alignment

// Clip pointers travel through LDS / records as plain 64-bit values; telling the compiler they are
// GLOBAL (address space 1) makes it emit global_load_* instead of flat_load_* (flat loads also occupy
// the LDS path and its counter).
#define WBX_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const T WBX_GLOBAL* as_global(const void* p) {
  return (const T WBX_GLOBAL*)(uintptr_t)p;
}

// reference math::clamp (core_math.h:33-37)
// 16 bytes to (pinned) HOST memory, acknowledged only when the write is on its way to the host: a system-scope store.  The
// one-launch callback (wbx_callback.h) tells the audio thread "master and status are there" with a flag it writes from inside
// the kernel; an ordinary store is acknowledged by the L2 at once, and the flag — another wave's store, another L2 channel —
// overtook the data (measured: the left channel of the first block read as zeros).  s_waitcnt vmcnt(0) behind THIS store
// waits for the real acknowledgement.
__device__ __forceinline__ void store_f4_system(float* p, f4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_u4_system(void* p, uint4 u) {
  const f4 v = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) {
  float m = x < hi ? x : hi;
  return m > lo ? m : lo;
}
__device__ __forceinline__ double clampd(double x, double lo, double hi) {
  double m = x < hi ? x : hi;
  return m > lo ? m : lo;
}

// max |m| over the 4 frames of a lane (vu_meter.h:20-25).  The operands are results of fp32 multiplies (never
// signalling NaNs), so the source modifiers can be used directly: two instructions instead of the four that
// fmaxf's canonicalisation rules cost.
__device__ __forceinline__ float absmax4(f4 m) {
  float t;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t) : "v"(m.x), "v"(m.y), "v"(m.z));
  asm("v_max_f32_e64 %0, |%1|, %2" : "=v"(t) : "v"(m.w), "v"(t));
  return t;
}

// Branch-free tap selection for the 5-sample window: returns w[k] / w[k+1] for k in [0, E].
// Selection is done with sign masks and v_bfi_b32 ((m & a) | (~m & b)) rather than compare + v_cndmask_b32: on
// gfx950 a VCC-conditioned v_cndmask_b32 issues at about a fifth of the rate of a plain VALU op
// (tools/ubench/valu_rate.hip), and an if/switch would become exec-mask control flow.
__device__ __forceinline__ float sel_neg(int t, float a, float b) {   // t < 0 ? a : b
  const int m = t >> 31;
  float r;   // inline asm: the optimiser would fold the mask form back into compare + v_cndmask_b32
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}
template <int E>
__device__ __forceinline__ void taps(const f4& v, float w4, int k, float& sa, float& sb) {
  sa = v.x;
  sb = v.y;
  if (E >= 1) {
    sa = sel_neg(k - 1, sa, v.y);
    sb = sel_neg(k - 1, sb, v.z);
  }
  if (E >= 2) {
    sa = sel_neg(k - 2, sa, v.z);
    sb = sel_neg(k - 2, sb, v.w);
  }
  if (E >= 3) {
    sa = sel_neg(k - 3, sa, v.w);
    sb = sel_neg(k - 3, sb, w4);
  }
}

// The same selection when the playback speed lies in [kNarrowSpeed, 0.999]: frame j0+E then starts at window
// sample E-1 or E (E*speed + frac(x0) lies in (E-1, E+1) with margin far above the fp64 rounding of the
// positions), so one mask and two selects per frame are enough.
constexpr double kNarrowSpeed = 0.67;
template <int E>
__device__ __forceinline__ void taps_narrow(const f4& v, float w4, int k, float& sa, float& sb) {
  const int t = k - E;   // negative: the frame starts at window sample E-1
  if (E == 1) {
    sa = sel_neg(t, v.x, v.y);
    sb = sel_neg(t, v.y, v.z);
  } else if (E == 2) {
    sa = sel_neg(t, v.y, v.z);
    sb = sel_neg(t, v.z, v.w);
  } else {
    sa = sel_neg(t, v.z, v.w);
    sb = sel_neg(t, v.w, w4);
  }
}

// max over the 64 lanes of a wave, delivered in lane 63, with DPP row operations only (no LDS crossbar):
// quad butterflies, row_half_mirror, row_mirror, then row_bcast:15 / row_bcast:31 into the upper rows.
// The values are non-negative floats, so the comparison is done on their bit patterns as unsigned integers:
// 0 is the identity (what a masked-out row keeps), no NaN canonicalisation is needed, and each step folds into
// one v_max_u32 with a DPP source operand.
__device__ __forceinline__ float wave_max_lane63(float x) {
  uint32_t v = __float_as_uint(x);
#define WBX_DPP_MAX(ctrl, rmask)                                                                      \
  {                                                                                                  \
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, (ctrl), (rmask), 0xF, true);  \
    v = v > o ? v : o;                                                                               \
  }
  WBX_DPP_MAX(0xB1, 0xF)    // quad_perm [1,0,3,2]
  WBX_DPP_MAX(0x4E, 0xF)    // quad_perm [2,3,0,1]
  WBX_DPP_MAX(0x141, 0xF)   // row_half_mirror
  WBX_DPP_MAX(0x140, 0xF)   // row_mirror: every lane of a 16-lane row now holds the row maximum
  WBX_DPP_MAX(0x142, 0xA)   // row_bcast:15 -> rows 1 and 3
  WBX_DPP_MAX(0x143, 0xC)   // row_bcast:31 -> rows 2 and 3: lane 63 holds the wave maximum
#undef WBX_DPP_MAX
  return __uint_as_float(v);
}

// The wave maxima of FOUR tracks at once (values non-negative floats compared as unsigned integers).  gfx950's
// v_permlane32_swap / v_permlane16_swap transpose while reducing: after two levels the four 16-lane rows of one
// register hold the partial maxima of tracks 0, 2, 1, 3, and the four DPP row steps finish all of them together
// — 10 VALU instructions per four tracks instead of 24.  Every lane of row r returns the maximum of track
// kQuadRowTrack[r].
__device__ __forceinline__ uint32_t wave_max_quad(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3) {
  typedef unsigned int u2v __attribute__((ext_vector_type(2)));
  const u2v a = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);   // {p0.lo | p1.lo}, {p0.hi | p1.hi}
  const uint32_t m01 = a.x > a.y ? a.x : a.y;                             // lanes 0-31: track 0, lanes 32-63: track 1
  const u2v b = __builtin_amdgcn_permlane32_swap(p2, p3, false, false);
  const uint32_t m23 = b.x > b.y ? b.x : b.y;
  const u2v c = __builtin_amdgcn_permlane16_swap(m01, m23, false, false); // rows {0:t0 2:t2 1:t1 3:t3} x two halves
  uint32_t v = c.x > c.y ? c.x : c.y;
#define WBX_DPP_MAX(ctrl)                                                                         \
  {                                                                                               \
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, (ctrl), 0xF, 0xF, true);  \
    v = v > o ? v : o;                                                                            \
  }
  WBX_DPP_MAX(0xB1) WBX_DPP_MAX(0x4E) WBX_DPP_MAX(0x141) WBX_DPP_MAX(0x140)
#undef WBX_DPP_MAX
  return v;
}

// The same when the two 32-lane halves of the wave hold different channels (128-frame stereo blocks): the maxima
// of four tracks per half.  The first swap level then separates the channels instead of folding them: x keeps
// channel 0 of tracks {0 | 1}, y channel 1; rows end up holding tracks 0, 2, 1, 3 as above, once per channel.
__device__ __forceinline__ void wave_max_quad_halves(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t& ch0,
                                                     uint32_t& ch1) {
  typedef unsigned int u2v __attribute__((ext_vector_type(2)));
  const u2v a = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);   // x = {p0.lo | p1.lo}, y = {p0.hi | p1.hi}
  const u2v b = __builtin_amdgcn_permlane32_swap(p2, p3, false, false);
  const u2v c0 = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
  const u2v c1 = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
  uint32_t v = c0.x > c0.y ? c0.x : c0.y, w = c1.x > c1.y ? c1.x : c1.y;
#define WBX_DPP_MAX(x, ctrl)                                                                      \
  {                                                                                               \
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xF, 0xF, true); \
    (x) = (x) > o ? (x) : o;                                                                      \
  }
  WBX_DPP_MAX(v, 0xB1) WBX_DPP_MAX(v, 0x4E) WBX_DPP_MAX(v, 0x141) WBX_DPP_MAX(v, 0x140)
  WBX_DPP_MAX(w, 0xB1) WBX_DPP_MAX(w, 0x4E) WBX_DPP_MAX(w, 0x141) WBX_DPP_MAX(w, 0x140)
#undef WBX_DPP_MAX
  ch0 = v;
  ch1 = w;
}

enum : int { MODE_U = 0, MODE_W = 1, MODE_I16 = 2, MODE_I32 = 3, MODE_MIXED = 4, MODE_WN = 5, MODE_G = 6, MODE_WI = 7, MODE_WIN = 8, MODE_MU = 9, MODE_MW = 10, MODE_MWN = 11,
             MODE_WNU = 12, MODE_WINU = 13 };   // row shapes of a staged chunk (…U: every resampled row at MixArgs::uniform_speed)

// the loads of one track that are in flight while other tracks are being rendered (CL = channels per lane)
struct Win {
  f4 v;        // UNITY: the 4 source frames; WINDOW: window samples 0..3
  float w4;    // WINDOW: window sample 4
};
template <int CL>
struct PreT {
  Win w[CL];   // one window per channel of the lane
  int ix0;     // WINDOW: integer source position of v.x
  float fx0;   // WINDOW: interpolation fraction of frame j0
};
template <int CL>
struct PreGT {  // MODE_G (per-frame taps): v = first tap, b = second tap, fx = fraction of each of the lane's 4 frames
  f4 v[CL], b[CL], fx;
};
// the rendered frames of one track: a lane's four frames of each of its channels
template <int CL>
struct RowT {
  f4 c[CL];
};

// ------------------------------------------------------------------------------------------------
// mix: grid = (n_blocks, n_groups, tiles) — the workgroups in flight together work on neighbouring BLOCKS of the
// same track group (XCD-aware order, see below), so they read long contiguous runs of the same clips.
// block = 256 lanes (4 waves).  Lane -> (channel c, frames j0..j0+3).  With F = 512, C = 2: waves 0-1
// own the left channel, waves 2-3 the right one; every wave-level load is one contiguous ~1 KiB row.
//
// By the time this kernel runs every record is a whole-block unity or window row (gen_kernel rewrote the generic
// ones; silent ones and the padding of the last batch read the zero page with zero gain), and every staged chunk
// runs in ONE mode chosen from the row shapes it holds (MODE_*), so the load phase is straight-line code — e.g.
// exactly one 16-B + one 4-B load per track for resampled fp32 rows — with no branches, which lets the compiler
// count outstanding loads exactly and keep two batches in flight.
//
//   U     tracks per batch; two batches are in flight (software pipeline: the loads of batch i+1 are
//         issued before batch i is rendered), so 2*U clip rows per wave are outstanding
//   FULL  every lane owns a slot and every wave stays inside one block, so that a staged record is wave-uniform:
//         no lane predicate, record fields in scalar registers
//   W     waves per SIMD the register budget is capped for
//   G     the instance carries the per-frame-tap and 16-bit window modes (MODE_G, MODE_WI, MODE_WIN); sessions
//         without such clips run the instance without them
//   SB    consecutive blocks per workgroup (1: C*F/4 is a multiple of 256; 2 / 4: blocks of 128 / 64 lanes)
//   CW    channels per wave (1: the channel is a scalar; 2: 128-frame stereo, a channel per 32-lane half)
//   CL    channels per lane (1: a wave renders one channel of its frames; 2: stereo, both channels of a lane's four
//         frames — the position / fraction arithmetic of sampler.cpp:50-52 is per frame, not per sample, so a resampled
//         row costs it once instead of once per channel, and one set of record scalars serves both.  Workgroups of
//         128 lanes = one 512-frame block)
// ------------------------------------------------------------------------------------------------
// (the kernel's body as a function: mix_kernel below is nothing else; the one-launch callback — wbx_callback.h — runs the
//  sequencer of the workgroup's tracks in front of it and the block's sum behind it)
template <int U, bool FULL, int FAM, int SB, int CW, int CL, int T, int X = 0>
__device__ __forceinline__ void mix_body(const MixArgs& a) {
  // FAM: which chunk modes the instance carries — every mode it carries costs registers in all the others.
  //   0  fp32 (unity / window) and integer PCM at unity speed: U, W, WN, WNU, I16, I32, MU, MIXED
  //   1  everything: also per-frame taps, 16-bit / 24-bit / 32-bit window rows, windows of several formats in one chunk
  //   2  sessions whose clips are all 16-bit PCM at speeds up to 0.999 or exactly 1 (CD-rate files in a 48 kHz project): U,
  //      I16, MU, WI, WIN, WINU; a chunk that holds fp32 rows next to 16-bit window rows (a pre-rendered block) -> MIXED
  //   3  family 1 without the per-frame taps (MODE_G): sessions with resampled integer PCM but no clip that needs them — what
  //      that mode costs in registers is the room for both channels per lane
  constexpr bool G = FAM == 1 || FAM == 3;
  constexpr bool STRIDE = FAM == 1;   // per-frame taps (MODE_G)
  constexpr bool W16 = FAM >= 1;      // the 16-bit window modes
  constexpr bool LEAN16 = FAM == 2;
  static_assert(SB == 1 || FULL, "sub-blocks need waves that stay inside one block");
  static_assert(CW == 1 || (CW == 2 && (SB == 4 || (SB == 1 && T == 64))), "two channels per wave: 128-frame stereo blocks, one block per wave");
  static_assert(CL == 1 || (CL == 2 && FULL && SB == 1 && CW == 1), "two channels per lane: stereo blocks of 4 * T frames");
  static_assert(T == 256 / CL || (CL == 2 && (T == 64 || T == 256)) || (CL == 1 && SB == 1 && (T == 64 || T == 128)), "lanes per workgroup");
  constexpr uint32_t kT = (uint32_t)T;   // lanes per workgroup (CL = 2: one block of 4 * T frames — 256, 512 or 1024)
  // tracks staged at a time (four sub-blocks: half, to keep 4 workgroups per CU in LDS; one-wave workgroups: half, so
  // that twelve of them fit — 12.6 KiB each)
  // X (round 4): a PACKED instance (SB = 2 / 4 blocks per workgroup) that takes masked rows too — short blocks of a session cut
  // into clips used to leave the packed instances for one-wave workgroups.  1: half the tracks per chunk, so that the doubled
  // row space costs no LDS (occupancy as before); [2: the packed instances' own chunk length at twice the LDS — three
  // workgroups per CU; measured 3-10 % behind 1 and not instantiated]
  constexpr bool XP = X != 0;
  static_assert(!XP || (SB > 1 && FULL), "masked rows in a packed instance");
  constexpr uint32_t kSt = XP ? (kStage >> (X == 1 ? 1 : 0)) / (SB == 4 ? 2u : 1u) : SB == 4 ? kStage / 2 : kT == 64u ? kStage / 2 : kStage;
  // EXP: the instance takes the sequencer's masked rows (MixArgs::masked_rows): a track-block with a clip boundary in
  // it is a ROW_PAIR of two single-segment records, so a chunk of kSt tracks stages up to 2 * kSt rows
  constexpr bool EXP = XP || (SB == 1 && FULL && (kSt == 128 || kT == 64u));
  constexpr uint32_t kMaxRows = EXP ? 2 * kSt : kSt;
  constexpr uint32_t kRecs = kMaxRows + 2 * U + 4;      // staged records + null padding for the last batches
  __shared__ __attribute__((aligned(16))) DTrackBlock s_tb[SB * kRecs];   // [sub-block][record]
  constexpr uint32_t kWaves = kT / 64u;
  constexpr uint32_t kPS = (CL == 2 && kWaves == 4u) ? 8u : 4u;   // peak slots per record: one per (wave, channel of the wave)
  __shared__ uint32_t s_pk[SB * kRecs * kPS];   // FULL: one slot per (record, wave[, channel]), plain stores; else (record, channel), atomics
  __shared__ uint32_t s_wc[kPS];         // FULL: sub-block * C + channel a slot holds (unused slots: none)
  // (XP: everything but the routing entries per sub-block)
  __shared__ __attribute__((aligned(16))) DRow s_rows[EXP ? 2 * SB * kSt : 1];   // EXP: the plan rows of this chunk and (prefetched) of the next
  __shared__ uint32_t s_ord[EXP ? kSt : 1];       // EXP: the routing-order entries of the next chunk (prefetched)
  __shared__ uint16_t s_map[EXP ? SB * 2 * kSt : 1];   // EXP: staged row -> local track (bit 15: the second record of its pair)
  __shared__ uint16_t s_off[EXP ? SB * (kSt + 1) : 1];   // EXP: local track -> its first staged row
  __shared__ uint32_t s_wpairs[3];                // EXP: pairs in waves 0 and 1, staged rows of the chunk
  __shared__ uint32_t s_hp[XP ? 8 : 1];           // XP: pairs per 32-lane half of the row-fetching lanes
  __shared__ uint32_t s_tot[XP ? SB : 1];         // XP: staged rows of each sub-block
  __shared__ int s_shape;                         // the row shapes the chunk holds (OR over its records)

  // Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  Consecutive blocks
  // of a group read adjacent pieces of the same clip rows (they share the cache line at the seam), so give every
  // XCD a contiguous run of blocks: id x -> block (x % 8) * K/8 + x / 8.
  const unsigned long long dbg_t0 = a.dbg_clock ? wall_clock64() : 0ull;
  uint32_t bx = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) bx = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t g = blockIdx.y, tile = blockIdx.z;
  const uint32_t tid = threadIdx.x;
  const DGroup grp = a.groups[g];
  const uint32_t F = a.block_frames, C = a.channels, N = a.n_tracks;
  const uint32_t S4 = F >> 2;
  // Block sizes between the shapes the instances are cut for (round 5).  A device period is whatever the back end grants — 10 ms
  // of WASAPI shared mode are 480 frames at 48 kHz, 441 -> 416 at 44.1 kHz once config.cpp:217-222 has realigned them to 32 —
  // and such a block used to fall to the general instance (lane predicates, records per lane from LDS: 0.29-0.33 of the
  // roofline against 0.67).  Now the FULL instances serve it: the instance's lane space gives every channel of a block
  // MixArgs::lane_span lanes (>= F/4: the next shape an instance exists for), and the lanes beyond F/4 CLONE the block's last
  // four frames — same addresses (cache hits), same values (a maximum does not change), nothing stored.  Every wave still
  // stays inside one channel of one block, records stay wave-uniform, no lane predicate anywhere in the loops; a block of the
  // instance's own size has lane_span = F/4 and no clones.
  const uint32_t Lc = FULL ? a.lane_span : S4;
  // SB > 1 (blocks shorter than a workgroup: C*Lc * SB == 256): the workgroup renders SB consecutive blocks, every
  // wave stays inside one (sub-block, channel) and reads its own sub-block's records
  const uint32_t sub = SB > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / (C * Lc))) : 0u;
  const uint32_t rb = sub * kRecs;                 // this wave's records in s_tb
  const uint32_t b = bx * SB + sub;
  const bool bvalid = SB == 1 || b < a.n_blocks;   // the last workgroup of an odd render has an empty sub-block
  const uint32_t slot = SB > 1 ? tid - sub * (C * Lc) : tile * kT + tid;
  const bool active = FULL ? true : (slot < C * S4);
  uint32_t c = active ? slot / Lc : 0u;
  if (FULL && CL == 1 && c >= C) c = C - 1u;                    // (a spare wave behind the last channel's: clones, too)
  if (FULL && CW == 1) c = __builtin_amdgcn_readfirstlane(c);   // CW == 2: lanes 0-31 channel 0, lanes 32-63 channel 1
  const uint32_t jn = active ? slot - (slot / Lc) * Lc : 0u;    // the lane's place among its channel's lanes
  if (CL == 2) c = 0u;                                          // both channels in every lane: element ch of the arrays below
  // `owns`: the lane's four frames are the block's (not a clone of its last four) and this lane stores them
  const bool owns = FULL ? (jn < S4 && slot < C * Lc) : active;
  const uint32_t j0 = active ? (jn < S4 ? jn : S4 - 1u) * 4u : 0u;
  // lanes of an aligned `span`-lane group share a channel (span = largest power of two dividing F/4, <= 64)
  uint32_t span = 64u;
  if (!FULL) {
    span = S4 & (~S4 + 1u);
    span = span > 64u ? 64u : span;
  }
  const uint32_t lane = tid & 63u;
  const double j0d = (double)(int32_t)j0;

  using Pre = PreT<CL>;
  using PreG = PreGT<CL>;
  using Row = RowT<CL>;
  Row acc;
#pragma unroll
  for (int ch = 0; ch < CL; ch++) acc.c[ch] = f4{0.0f, 0.0f, 0.0f, 0.0f};
  // MixArgs::init: the running sum of the tracks BEFORE this engine's in the session's track order (another engine's
  // un-clamped master: the previous shard of a multi-GPU chain) — the first group in summation order starts from it instead
  // of the cleared buffer, so the additions continue exactly where that engine stopped
  if (a.init && g == 0u && owns && bvalid) {
#pragma unroll
    for (int ch = 0; ch < CL; ch++)
      acc.c[ch] = *reinterpret_cast<const f4*>(a.init + ((size_t)b * C + (CL == 2 ? (uint32_t)ch : c)) * F + j0);
  }

  // Chained render (MixArgs::chain): this group is a piece of a longer member list and CONTINUES the running sum of the
  // piece before it — the reference's strictly sequential order (engine.cpp:1600-1617) at the parallelism and the dynamic
  // workgroup scheduling of the grouped order.  Workgroup (x, g) is dispatched after (x, g-1) (lower linear id, in-order
  // dispatch), so the predecessor is running or done; its "sum is out" word is read here, early — the staging of this
  // group's records hides the round trip — and waited for right before the first row is added.
  // Coherence: workgroups are handed to the 8 XCDs round-robin by linear id and the grid's x extent is a multiple of 8
  // (the host chains only then), so (x, g-1) and (x, g) sit behind the SAME L2: the running sum is written with plain
  // stores (the vector L1 writes through), the word after they are acknowledged, and both are read past the L1 (agent-scope
  // loads, sc1: the L1 is bypassed, the XCD's own L2 answers).  Workgroup scope would poll a stale L1 line for ever; going
  // through memory (system scope) cost 40 % of the kernel time.  The word carries the writer's XCC id: a
  // successor on another XCD — the one thing this rests on — is reported (status bit 6), never silently wrong.
  const bool chain_in = a.chain != nullptr && (grp.flags & GROUP_CHAIN_IN) != 0u;
  const bool chain_out = a.chain != nullptr && (grp.flags & GROUP_CHAIN_OUT) != 0u;
  uint32_t* chain_word = a.chain ? a.chain + ((size_t)(blockIdx.x * a.tiles + tile) * a.n_groups + g) : nullptr;
  const uint32_t my_xcc = a.chain ? ((uint32_t)__builtin_amdgcn_s_getreg(63508) & 0xFu) + 1u : 0u;   // HW_REG_XCC_ID + 1
  // a word reads (epoch << 4) | XCC id + 1 once its piece is out; the epoch is the render's, so the words are never cleared
  // between renders (a memset in front of every mix sat behind the previous render's sum and cost its whole duration)
  const uint32_t chain_tag = a.chain_epoch << 4;
  uint32_t chain_seen = 0u;
  if (chain_in && tid == 0u) chain_seen = __hip_atomic_load(chain_word - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // frame positions j0+e as doubles, once per lane (the fp64 operand of sampler.cpp:50)
  const double jd1 = j0d + 1.0, jd2 = j0d + 2.0, jd3 = j0d + 3.0;
  // The usual session has ONE resampling ratio (44.1 kHz clips in a 48 kHz session): the products fl(j * speed) of
  // sampler.cpp:50 then depend on the lane only, not on the track — once per lane instead of four fp64 multiplies per
  // track (the loop is VALU-bound and fp64 runs at half rate).  MixArgs::uniform_speed (> 0) is the host's word that
  // every KIND_WINDOW / KIND_WINDOW_I16 row of this render plays at exactly that speed (bit for bit).
  const double us = a.uniform_speed;
  const double up0 = __dmul_rn(j0d, us), up1 = __dmul_rn(jd1, us), up2 = __dmul_rn(jd2, us), up3 = __dmul_rn(jd3, us);

  // A staged record is wave-uniform.  Instead of broadcast-reading its fields from LDS one by one (every such read
  // returns 64 x 8..16 B through the LDS data path), each lane reads ONE dword of the 64-B record and the fields
  // are pulled into scalar registers with v_readlane: one 4-B LDS read per (track, phase), and the values feed the
  // vector ALU as scalar operands.
  struct URec {
    const void* src[CL];   // src[c] (CL == 2: both channels)
    double pos, speed;
    float gain, gc[CL];    // clip gain, fl(volume * pan_c)
    uint32_t kind, format;
    uint32_t d, n;     // EXP: the stream call covers frames [d, d + n) of the block (whole-block records: 0, F)
    bool partial;      // EXP: KIND_PARTIAL
  };
  auto load_urec = [&](uint32_t rec) {
    const int w = (int)reinterpret_cast<const uint32_t*>(&s_tb[rb + rec])[lane & 15u];
    auto rl = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readlane(w, (int)i); };
    URec r;
    if (CW == 2) {   // both channels' pointer and gain as scalars, a per-lane select between them
      const uint64_t s0 = ((uint64_t)rl(1) << 32) | rl(0), s1 = ((uint64_t)rl(3) << 32) | rl(2);
      r.src[0] = (const void*)(c ? s1 : s0);
      r.pos = __longlong_as_double((long long)(((uint64_t)rl(5) << 32) | rl(4)));
      r.speed = __longlong_as_double((long long)(((uint64_t)rl(7) << 32) | rl(6)));
      r.gain = __uint_as_float(rl(8));
      r.gc[0] = __uint_as_float(c ? rl(10) : rl(9));
      const uint32_t q11 = rl(11);
      r.kind = (q11 >> 8) & KIND_MASK;
      r.format = rl(13) & 0xFFu;
      if (EXP) {   // (as below)
        r.partial = ((q11 >> 15) & 1u) != 0u;
        r.d = 0u;
        r.n = F;
        if (r.partial) {
          r.d = q11 >> 16;
          r.n = rl(12) & 0xFFFFu;
        }
      }
      return r;
    }
    const uint32_t cs = FULL ? c : 0u;   // (only used when FULL: the channel is wave-uniform)
#pragma unroll
    for (int ch = 0; ch < CL; ch++) {
      const uint32_t cc = CL == 2 ? (uint32_t)ch : cs;
      r.src[ch] = (const void*)(((uint64_t)rl(2u * cc + 1u) << 32) | rl(2u * cc));
      r.gc[ch] = __uint_as_float(rl(9u + cc));
    }
    r.pos = __longlong_as_double((long long)(((uint64_t)rl(5) << 32) | rl(4)));
    r.speed = __longlong_as_double((long long)(((uint64_t)rl(7) << 32) | rl(6)));
    r.gain = __uint_as_float(rl(8));
    const uint32_t q11 = rl(11);
    r.kind = (q11 >> 8) & KIND_MASK;
    r.format = rl(13) & 0xFFu;
    if (EXP) {
      // whole-block records (no KIND_PARTIAL flag) need neither their bounds nor the arithmetic that goes with them
      r.partial = ((q11 >> 15) & 1u) != 0u;
      r.d = 0u;
      r.n = F;
      if (r.partial) {
        r.d = q11 >> 16;
        r.n = rl(12) & 0xFFFFu;
      }
    }
    return r;
  };
  const uint32_t wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)j0);   // FULL: the wave's first frame
  constexpr uint32_t kWF = CW == 2 ? 128u : 256u;   // ... and how many it covers (CW == 2: the 128 frames of the block, once per half-wave)
  const uint32_t wave_end = wave_base + kWF < F ? wave_base + kWF : F;   // (a block shorter than the instance's lane space ends earlier)
  // Masked rows (EXP).  Frame j0+e of the block is frame (j0+e-d) of the stream call; clamped into the call, so that
  // what the masked-out frames of a lane load stays inside the clip — they are zeroed afterwards.  For a whole-block
  // record (d = 0, n = F) this is j0+e itself.
  auto call_frame = [&](uint32_t e, uint32_t d, uint32_t n) {
    const int x = (int)(j0 + e) - (int)d, hi = (int)n - 1;
    const int lo = x > 0 ? x : 0;
    return lo < hi ? lo : hi;
  };
  // all-ones when frame j0+e lies inside [d, d+n), zero otherwise
  auto frame_mask = [&](uint32_t e, uint32_t d, uint32_t n) { return (uint32_t)0 - (uint32_t)((j0 + e - d) < n); };
  auto and_mask = [&](float v, uint32_t m) { return __uint_as_float(__float_as_uint(v) & m); };
  // (round 6) The cheap form of a masked row.  A partial stream call that starts at least four samples into its clip — every clip
  // of a track cut into back-to-back clips but the one that opens its sample — can be rendered by the UNMASKED arithmetic of a
  // call that covers the whole wave (frame j of the block is frame j - d of the call: one subtraction in the position) followed by
  // the frame masks: a lane that straddles the call's start keeps its own, negative, call frame, so its frames inside the call
  // sit where that arithmetic expects them, and what its frames in front of the call read lies inside the clip (>= pos - 3
  // samples >= 1); frames behind the call's end read at most five samples past it: the clip's 16 frames of padding.  Lanes
  // wholly outside the call load at its first / last frame and are masked to +0.0 whatever they compute.  The values of the
  // frames inside the call are those of the clamped form bit for bit (same position expression: pos + fl((j - d) * speed)).
  // Calls that start within the first four samples keep the clamped form (row_*_masked below).  wave-uniform.
  auto fast_part = [&](const URec& r) {
    return EXP && a.fast_partial != 0u && r.partial && __double2hiint(r.pos) >= 0x40100000;   // pos >= 4.0 (pos is never negative here)
  };
  auto part_cf0 = [&](const URec& r) {   // the call frame a partial row's loads start from
    if (fast_part(r)) {
      const int x = (int)j0 - (int)r.d, hi = (int)r.n - 1;
      return ((int)j0 + 3 < (int)r.d) ? 0 : (x < hi ? x : hi);
    }
    return call_frame(0u, r.d, r.n);
  };

  // ---- per-row arithmetic (each returns the 4 frames of one track AFTER clip gain and track gain) ----
  // fp32 row at unity speed: sampler.cpp:151-152, track.cpp:731
  // (clip gain exactly 1.0 — the usual case — is wave-uniform: s * 1.0f is s, bit for bit, so that multiply is left out)
  auto row_f32 = [&](const f4& v, float cg, float gc) {
    f4 m;
    if (__float_as_uint(cg) == 0x3F800000u) {
      m.x = __fmul_rn(v.x, gc);
      m.y = __fmul_rn(v.y, gc);
      m.z = __fmul_rn(v.z, gc);
      m.w = __fmul_rn(v.w, gc);
    } else {
      m.x = __fmul_rn(__fmul_rn(v.x, cg), gc);
      m.y = __fmul_rn(__fmul_rn(v.y, cg), gc);
      m.z = __fmul_rn(__fmul_rn(v.z, cg), gc);
      m.w = __fmul_rn(__fmul_rn(v.w, cg), gc);
    }
    return m;
  };
  // 16-bit PCM at unity speed, sampler.cpp:109-120: clamp((float)d * (1.0f/32767), -1, 1) * gain — the four samples
  // before the gains ...
  auto norm_i16 = [&](int lo, int hi) {
    const float norm = 1.0f / 32767.0f;                                                   // :95
    const float d0 = (float)(short)(lo & 0xFFFF), d1 = (float)(short)((unsigned)lo >> 16);
    const float d2 = (float)(short)(hi & 0xFFFF), d3 = (float)(short)((unsigned)hi >> 16);
    return f4{clampf(__fmul_rn(d0, norm), -1.0f, 1.0f), clampf(__fmul_rn(d1, norm), -1.0f, 1.0f),
              clampf(__fmul_rn(d2, norm), -1.0f, 1.0f), clampf(__fmul_rn(d3, norm), -1.0f, 1.0f)};
  };
  // ... and the row (clip gain, then track gain: two roundings also when the clip gain is 1.0 — s * 1.0f is s)
  auto row_i16 = [&](int lo, int hi, float cg, float gc) {
    const f4 x = norm_i16(lo, hi);
    return f4{__fmul_rn(__fmul_rn(x.x, cg), gc), __fmul_rn(__fmul_rn(x.y, cg), gc), __fmul_rn(__fmul_rn(x.z, cg), gc),
              __fmul_rn(__fmul_rn(x.w, cg), gc)};
  };
  // 24-bit (32-bit containers) / 32-bit PCM at unity speed, sampler.cpp:121-144: (float)clamp((double)d * norm, -1, 1) * gain
  auto norm_i32 = [&](const f4& bits, uint32_t format) {
    const double norm = format == FMT_I24 ? 1.0 / 8388607.0 : 1.0 / 2147483647.0;         // :96-97
    return f4{(float)clampd(__dmul_rn((double)__float_as_int(bits.x), norm), -1.0, 1.0),
              (float)clampd(__dmul_rn((double)__float_as_int(bits.y), norm), -1.0, 1.0),
              (float)clampd(__dmul_rn((double)__float_as_int(bits.z), norm), -1.0, 1.0),
              (float)clampd(__dmul_rn((double)__float_as_int(bits.w), norm), -1.0, 1.0)};
  };
  auto row_i32 = [&](const f4& bits, uint32_t format, float cg, float gc) {
    const f4 x = norm_i32(bits, format);
    return f4{__fmul_rn(__fmul_rn(x.x, cg), gc), __fmul_rn(__fmul_rn(x.y, cg), gc), __fmul_rn(__fmul_rn(x.z, cg), gc),
              __fmul_rn(__fmul_rn(x.w, cg), gc)};
  };
  // Linear resample (sampler.cpp:34-59) of a lane's four frames from a 5-sample window per channel.  Position, integer
  // part and fraction of a frame (:50-52) belong to the frame, not to the sample: win_pos works them out once, win_row
  // applies them to one channel's window.
  struct WPos {
    float fx[4];   // interpolation fraction of frames j0..j0+3
    int k[4];      // window sample the frame starts at: (int)x - ix0 (k[0] = 0)
  };
  auto win_pos = [&](auto shifted, auto uni, int ix0, float fx0, double pos, double speed, double d0) {
    constexpr bool UNI = decltype(uni)::value;           // the row plays at MixArgs::uniform_speed: products hoisted (not with SHIFTED)
    constexpr bool SHIFTED = decltype(shifted)::value;   // the stream call starts at block frame d0: call frame = j - d0
    WPos wp;
    wp.fx[0] = fx0;   // frame j0: position and fraction already known from the load phase; its taps are window samples 0 and 1
    wp.k[0] = 0;
#define WBX_POS(E, JD)                                                                                   \
  {                                                                                                      \
    const double x = __dadd_rn(pos, (UNI && !SHIFTED) ? up##E : __dmul_rn(SHIFTED ? (JD) - d0 : (JD), speed));   /* sampler.cpp:50 */ \
    wp.fx[E] = (float)__builtin_amdgcn_fract(x);                          /* :52 (x >= 0, exact) */      \
    wp.k[E] = (int)x - ix0;                                               /* :51 */                      \
  }
    WBX_POS(1, jd1) WBX_POS(2, jd2) WBX_POS(3, jd3)
#undef WBX_POS
    return wp;
  };
  auto win_row = [&](auto narrow, auto unitg, const Win& w, const WPos& wp, float cg, float gc) {
    constexpr bool UNITG = decltype(unitg)::value;   // clip gain == 1.0f: fl(s * 1) = s, the multiply is left out
    constexpr bool NARROW = decltype(narrow)::value;
    float q[4];
    {
      const float s = __fadd_rn(w.v.x, __fmul_rn(wp.fx[0], __fsub_rn(w.v.y, w.v.x)));     // :55
      q[0] = __fmul_rn(UNITG ? s : __fmul_rn(s, cg), gc);                                 // :56, track.cpp:731
    }
#define WBX_TAP(E)                                                                                      \
  {                                                                                                     \
    float sa, sb;                                                                                       \
    if (NARROW)                                                                                         \
      taps_narrow<E>(w.v, w.w4, wp.k[E], sa, sb);                                                       \
    else                                                                                                \
      taps<E>(w.v, w.w4, wp.k[E], sa, sb);                                                              \
    const float s = __fadd_rn(sa, __fmul_rn(wp.fx[E], __fsub_rn(sb, sa)));  /* :55 */                   \
    q[E] = __fmul_rn(UNITG ? s : __fmul_rn(s, cg), gc);                     /* :56, track.cpp:731 */    \
  }
    WBX_TAP(1) WBX_TAP(2) WBX_TAP(3)
#undef WBX_TAP
    return f4{q[0], q[1], q[2], q[3]};
  };
  auto row_window_at = [&](auto narrow, auto shifted, auto uni, const Pre& p, double pos, double speed, double d0, float cg,
                           const float (&gc)[CL]) {
    const WPos wp = win_pos(shifted, uni, p.ix0, p.fx0, pos, speed, d0);
    Row m;
    if (__float_as_uint(cg) == 0x3F800000u) {   // wave-uniform
#pragma unroll
      for (int ch = 0; ch < CL; ch++) m.c[ch] = win_row(narrow, std::true_type{}, p.w[ch], wp, cg, gc[ch]);
    } else {
#pragma unroll
      for (int ch = 0; ch < CL; ch++) m.c[ch] = win_row(narrow, std::false_type{}, p.w[ch], wp, cg, gc[ch]);
    }
    return m;
  };
  auto row_window = [&](auto narrow, const Pre& p, double pos, double speed, float cg, const float (&gc)[CL]) {
    return row_window_at(narrow, std::false_type{}, std::false_type{}, p, pos, speed, 0.0, cg, gc);
  };
  // the window (5-sample) loads of one fp32 row; also valid for unity rows (pos integral, speed 1.0)
  auto load_window = [&](const void* const (&src_c)[CL], double pos, double prod0, Pre& p) {
    const double x0 = __dadd_rn(pos, prod0);                                              // sampler.cpp:50, frame j0 (of the call): prod0 = fl(j * speed)
    const int ix0 = (int)x0;                                                              // :51 (x >= 0: truncation)
    if (active) {   // both loads unconditional: straight-line code lets the compiler count outstanding loads exactly
#pragma unroll
      for (int ch = 0; ch < CL; ch++) {
        const float WBX_GLOBAL* src = as_global<float>(src_c[ch]) + ix0;
        p.w[ch].v = __builtin_nontemporal_load(reinterpret_cast<const f4u WBX_GLOBAL*>(src));   // the taps of frames j0..j0+3 lie in src[0..4]
        p.w[ch].w4 = src[4];
      }
    }
    p.ix0 = ix0;
    // :52 fx = (float)(x - (double)ix): for x >= 0 that difference is x - floor(x), which v_fract_f64
    // delivers exactly (the subtraction is exact in fp64), one instruction instead of trunc + sub
    p.fx0 = (float)__builtin_amdgcn_fract(x0);
  };
  // the same window for a 16-bit PCM row: samples ix0..ix0+3 in one 8-B load (2-byte aligned), ix0+4 in the low half
  // of a 4-B load; the halves stay packed until the render phase
  auto load_window16 = [&](const void* const (&src_c)[CL], double pos, double prod0, Pre& p) {
    typedef int i2u __attribute__((ext_vector_type(2), aligned(2)));
    typedef int i1w __attribute__((aligned(2)));
    const double x0 = __dadd_rn(pos, prod0);                                              // sampler.cpp:50, frame j0
    const int ix0 = (int)x0;                                                              // :51
    if (active) {
#pragma unroll
      for (int ch = 0; ch < CL; ch++) {
        const short WBX_GLOBAL* src = as_global<short>(src_c[ch]) + ix0;
        const i2u w = __builtin_nontemporal_load(reinterpret_cast<const i2u WBX_GLOBAL*>(src));
        p.w[ch].v.x = __int_as_float(w.x);
        p.w[ch].v.y = __int_as_float(w.y);
        p.w[ch].w4 = __int_as_float(*reinterpret_cast<const i1w WBX_GLOBAL*>(src + 4));
      }
    }
    p.ix0 = ix0;
    p.fx0 = (float)__builtin_amdgcn_fract(x0);                                            // :52
  };
  // 16-bit PCM row, linear resample: taps a = norm * (float)src[ix] (sampler.cpp:9-10,53-54), then as fp32
  // (the packed window of load_window16 as floats.  `unity`: the samples of a unity-speed row read through the window
  //  loads — sampler.cpp:109-120 normalises those with 1.0f / 32767 and clamps, the linear path does neither)
  auto unpack16 = [&](const Pre& p, bool unity) {
    const float norm = unity ? 1.0f / 32767.0f : (float)(1.0 / 32767.0);
    Pre f;
#pragma unroll
    for (int ch = 0; ch < CL; ch++) {
      const int lo = __float_as_int(p.w[ch].v.x), hi = __float_as_int(p.w[ch].v.y), tl = __float_as_int(p.w[ch].w4);
      float s5[5] = {(float)(short)(lo & 0xFFFF), (float)(short)((unsigned)lo >> 16), (float)(short)(hi & 0xFFFF),
                     (float)(short)((unsigned)hi >> 16), (float)(short)(tl & 0xFFFF)};
#pragma unroll
      for (int i = 0; i < 5; i++) {
        s5[i] = unity ? __fmul_rn(s5[i], norm) : __fmul_rn(norm, s5[i]);
        if (unity) s5[i] = clampf(s5[i], -1.0f, 1.0f);
      }
      f.w[ch].v = f4{s5[0], s5[1], s5[2], s5[3]};
      f.w[ch].w4 = s5[4];
    }
    f.ix0 = p.ix0;
    f.fx0 = p.fx0;
    return f;
  };
  auto row_window16 = [&](auto narrow, auto uni, const Pre& p, double pos, double speed, float cg, const float (&gc)[CL]) {
    return row_window_at(narrow, std::false_type{}, uni, unpack16(p, false), pos, speed, 0.0, cg, gc);
  };
  // ... of a stream call that starts at block frame d0 and covers all of this wave's frames
  auto row_window16_shifted = [&](auto narrow, const Pre& p, double pos, double speed, double d0, float cg, const float (&gc)[CL]) {
    return row_window_at(narrow, std::true_type{}, std::false_type{}, unpack16(p, false), pos, speed, d0, cg, gc);
  };
  // 24/32-bit PCM row (4-byte containers: the fp32 window loads), linear resample: taps a = (float)(norm * (double)src[ix])
  // (sampler.cpp:11-14,53-54), then as fp32
  auto row_window32 = [&](auto narrow, auto uni, const Pre& p, uint32_t fmt, double pos, double speed, float cg, const float (&gc)[CL]) {
    const double norm = fmt == FMT_I24 ? 1.0 / 8388607.0 : 1.0 / 2147483647.0;
    Pre f;
#pragma unroll
    for (int ch = 0; ch < CL; ch++) {
      f.w[ch].v.x = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].v.x));
      f.w[ch].v.y = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].v.y));
      f.w[ch].v.z = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].v.z));
      f.w[ch].v.w = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].v.w));
      f.w[ch].w4 = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].w4));
    }
    f.ix0 = p.ix0;
    f.fx0 = p.fx0;
    return row_window_at(narrow, std::false_type{}, uni, f, pos, speed, 0.0, cg, gc);
  };
  // per-frame taps for any playback speed and storage format (sampler.cpp:50-52 for each of the lane's 4 frames):
  // four unaligned loads of the pair {src[ix], src[ix+1]} (8 B; 4 B for 16-bit PCM, kept packed in v) per channel; also
  // valid for unity rows (fx = 0, first tap = the sample itself).  fmt is wave-uniform.
  typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
  typedef int i1u __attribute__((aligned(2)));
  auto load_stride = [&](const void* const (&src_c)[CL], double pos, double speed, uint32_t fmt, bool part, uint32_t d, uint32_t n,
                         PreG& p) {
    double jd[4] = {j0d, jd1, jd2, jd3};
    if (part) {   // a stream call that covers [d, d + n) of the block: the frame inside the call, clamped into it
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) jd[k] = (double)call_frame(k, d, n);
    }
    float f4x[4];
    int ix[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double x = __dadd_rn(pos, __dmul_rn(jd[k], speed));                           // :50
      ix[k] = (int)x;                                                                     // :51
      f4x[k] = (float)__builtin_amdgcn_fract(x);                                          // :52
    }
#pragma unroll
    for (int ch = 0; ch < CL; ch++) {
      float a4[4] = {0.0f, 0.0f, 0.0f, 0.0f}, b4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (fmt == FMT_I16) {
        const short WBX_GLOBAL* base = as_global<short>(src_c[ch]);
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (active) a4[k] = __int_as_float(*reinterpret_cast<const i1u WBX_GLOBAL*>(base + ix[k]));
      } else {   // fp32 and 24/32-bit PCM: 4-byte containers
        const float WBX_GLOBAL* base = as_global<float>(src_c[ch]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (active) {
            const f2u t = *reinterpret_cast<const f2u WBX_GLOBAL*>(base + ix[k]);
            a4[k] = t.x;
            b4[k] = t.y;
          }
        }
      }
      p.v[ch] = f4{a4[0], a4[1], a4[2], a4[3]};
      p.b[ch] = f4{b4[0], b4[1], b4[2], b4[3]};
    }
    p.fx = f4{f4x[0], f4x[1], f4x[2], f4x[3]};
  };
  // one channel of the row of a per-frame-tap record; kind and fmt are wave-uniform
  auto row_stride = [&](const f4& pv, const f4& pb, const f4& pfx, uint32_t kind, uint32_t fmt, float cg, float gc) {
    const float va[4] = {pv.x, pv.y, pv.z, pv.w}, vb[4] = {pb.x, pb.y, pb.z, pb.w};
    const float fx[4] = {pfx.x, pfx.y, pfx.z, pfx.w};
    float m[4];
    if (kind == KIND_UNITY) {                     // fp32 at unity speed, pre-rendered rows, silent and padding records
#pragma unroll
      for (int k = 0; k < 4; k++) m[k] = __fmul_rn(__fmul_rn(va[k], cg), gc);                // sampler.cpp:151-152
    } else if (kind == KIND_UNITY_I16) {          // sampler.cpp:109-120
      const float norm = 1.0f / 32767.0f;                                                    // :95
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float d = (float)(short)(__float_as_int(va[k]) & 0xFFFF);
        m[k] = __fmul_rn(__fmul_rn(clampf(__fmul_rn(d, norm), -1.0f, 1.0f), cg), gc);
      }
    } else if (kind == KIND_UNITY_I32) {          // sampler.cpp:121-144
      const double norm = fmt == FMT_I24 ? 1.0 / 8388607.0 : 1.0 / 2147483647.0;             // :96-97
#pragma unroll
      for (int k = 0; k < 4; k++)
        m[k] = __fmul_rn(__fmul_rn((float)clampd(__dmul_rn((double)__float_as_int(va[k]), norm), -1.0, 1.0), cg), gc);
    } else {                                      // linear interpolation, sampler.cpp:34-59
      float ta[4], tb[4];
      if (fmt == FMT_F32) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          ta[k] = va[k];
          tb[k] = vb[k];
        }
      } else if (fmt == FMT_I16) {
        const float norm = (float)(1.0 / 32767.0);                                           // :9-10
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int w = __float_as_int(va[k]);
          ta[k] = __fmul_rn(norm, (float)(short)(w & 0xFFFF));
          tb[k] = __fmul_rn(norm, (float)(short)((unsigned)w >> 16));
        }
      } else {
        const double norm = fmt == FMT_I24 ? 1.0 / 8388607.0 : 1.0 / 2147483647.0;           // :11-14
#pragma unroll
        for (int k = 0; k < 4; k++) {
          ta[k] = (float)__dmul_rn(norm, (double)__float_as_int(va[k]));
          tb[k] = (float)__dmul_rn(norm, (double)__float_as_int(vb[k]));
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++)                                                            // :55-56, track.cpp:731
        m[k] = __fmul_rn(__fmul_rn(__fadd_rn(ta[k], __fmul_rn(fx[k], __fsub_rn(tb[k], ta[k]))), cg), gc);
    }
    return f4{m[0], m[1], m[2], m[3]};
  };
  // fp32 unity row covering only [d, d+n): the lane's load starts at call frame jb = call_frame(0), frame e reads
  // sample call_frame(e) - jb of it (0..e; the same sample again for frames clamped at the call's ends)
  auto row_f32_masked = [&](const f4& v, uint32_t d, uint32_t n, float cg, float gc) {
    const int jb = call_frame(0u, d, n);
    float q[4];
    q[0] = and_mask(__fmul_rn(__fmul_rn(v.x, cg), gc), frame_mask(0u, d, n));
#pragma unroll
    for (uint32_t e = 1; e < 4; e++) {
      const int k = call_frame(e, d, n) - jb;
      float sv = sel_neg(k - 1, v.x, v.y);
      if (e >= 2) sv = sel_neg(k - 2, sv, v.z);
      if (e >= 3) sv = sel_neg(k - 3, sv, v.w);
      q[e] = and_mask(__fmul_rn(__fmul_rn(sv, cg), gc), frame_mask(e, d, n));                    // sampler.cpp:151-152, track.cpp:731
    }
    return f4{q[0], q[1], q[2], q[3]};
  };
  // the same for a row read through the 5-sample window (the window loads started at call frame call_frame(0)): a
  // resampled row (sampler.cpp:34-59) or, `unity`, a unity-speed row inside a chunk of window rows
  auto row_window_masked = [&](const Pre& p, double pos, double speed, bool unity, uint32_t d, uint32_t n, float cg,
                               const float (&gc)[CL]) {
    WPos wp;
    wp.fx[0] = p.fx0;
    wp.k[0] = 0;
#define WBX_MPOS(E)                                                                                       \
  {                                                                                                       \
    const double x = __dadd_rn(pos, __dmul_rn((double)call_frame(E, d, n), speed));   /* sampler.cpp:50 */ \
    wp.fx[E] = (float)__builtin_amdgcn_fract(x);                                      /* :52 */           \
    wp.k[E] = (int)x - p.ix0;                                                         /* :51 */           \
  }
    WBX_MPOS(1) WBX_MPOS(2) WBX_MPOS(3)
#undef WBX_MPOS
    Row m;
#pragma unroll
    for (int ch = 0; ch < CL; ch++) {
      const Win& w = p.w[ch];
      float q[4];
      {
        const float lin = __fadd_rn(w.v.x, __fmul_rn(wp.fx[0], __fsub_rn(w.v.y, w.v.x)));         // :55
        q[0] = and_mask(__fmul_rn(__fmul_rn(unity ? w.v.x : lin, cg), gc[ch]), frame_mask(0u, d, n));
      }
#define WBX_MTAP(E)                                                                                       \
  {                                                                                                       \
    float sa, sb;                                                                                         \
    taps<E>(w.v, w.w4, wp.k[E], sa, sb);                                                                  \
    const float lin = __fadd_rn(sa, __fmul_rn(wp.fx[E], __fsub_rn(sb, sa)));          /* :55 */           \
    q[E] = and_mask(__fmul_rn(__fmul_rn(unity ? sa : lin, cg), gc[ch]), frame_mask(E, d, n));             \
  }
      WBX_MTAP(1) WBX_MTAP(2) WBX_MTAP(3)
#undef WBX_MTAP
      m.c[ch] = f4{q[0], q[1], q[2], q[3]};
    }
    return m;
  };
  auto row_window16_masked = [&](const Pre& p, double pos, double speed, bool unity, uint32_t d, uint32_t n, float cg,
                                 const float (&gc)[CL]) {
    return row_window_masked(unpack16(p, unity), pos, speed, unity, d, n, cg, gc);
  };
  // frames outside the stream call [d, d + n) contribute an exact +0.0
  auto mask_row = [&](Row& m, uint32_t d, uint32_t n) {
    const uint32_t k0 = frame_mask(0u, d, n), k1 = frame_mask(1u, d, n), k2 = frame_mask(2u, d, n), k3 = frame_mask(3u, d, n);
#pragma unroll
    for (int ch = 0; ch < CL; ch++) {
      m.c[ch].x = and_mask(m.c[ch].x, k0);
      m.c[ch].y = and_mask(m.c[ch].y, k1);
      m.c[ch].z = and_mask(m.c[ch].z, k2);
      m.c[ch].w = and_mask(m.c[ch].w, k3);
    }
  };
  // accumulate a row; pk[ch] = the lane's max |m| of channel ch
  auto add_row = [&](const Row& m0, float (&pk)[CL]) {
#pragma unroll
    for (int ch = 0; ch < CL; ch++) {
      f4 m = m0.c[ch];
      if (!FULL && !active) m = f4{0.0f, 0.0f, 0.0f, 0.0f};
      acc.c[ch].x = __fadd_rn(acc.c[ch].x, m.x);                                          // audio_buffer.h:73-82
      acc.c[ch].y = __fadd_rn(acc.c[ch].y, m.y);
      acc.c[ch].z = __fadd_rn(acc.c[ch].z, m.z);
      acc.c[ch].w = __fadd_rn(acc.c[ch].w, m.w);
      pk[ch] = absmax4(m);                                                                // vu_meter.h:20-25
    }
  };
  // FULL: the slot of s_pk[record][kPS] that takes the wave's peak of channel element ch — the wave itself (CL == 1:
  // four waves, one channel each) or channel * waves + wave (CL == 2: every wave holds both channels); s_wc[slot] names
  // the channel a slot holds
  const uint32_t wave = tid >> 6;
  auto pk_slot = [&](int ch) { return CL == 2 ? kWaves * (uint32_t)ch + wave : wave; };
  // per-track peak: wavefront max (DPP) when the wave is channel-uniform, shuffle-max across the lanes that
  // share a channel otherwise; then one LDS atomic per wave / lane group
  auto post_peak = [&](float pk, uint32_t tl, int ch) {
    if (FULL && CW == 2) {
      for (uint32_t off = 1; off < 32u; off <<= 1) pk = fmaxf(pk, __shfl_xor(pk, (int)off, 64));
      if ((lane & 31u) == 0u) s_pk[(rb + tl) * kPS + c] = __float_as_uint(pk);      // slot = channel: one wave per sub-block
    } else if (FULL) {
      pk = wave_max_lane63(pk);
      if (lane == 63u) s_pk[(rb + tl) * kPS + pk_slot(ch)] = __float_as_uint(pk);   // this wave's own slot: no atomic
    } else {
      for (uint32_t off = 1; off < span; off <<= 1) pk = fmaxf(pk, __shfl_xor(pk, (int)off, 64));
      if ((lane & (span - 1u)) == 0u && active) atomicMax(&s_pk[tl * 2u + c], __float_as_uint(pk));
    }
  };

  // the same for four consecutive tracks tl..tl+3 at once (FULL only): lane 0 of row r stores the wave maximum
  // of track tl + kQuadRowTrack[r] into this wave's slot
  const uint32_t quad_row = (((lane >> 4) & 1u) * 2u + (lane >> 5)) * kPS;   // rows hold tracks 0,2,1,3
  const bool quad_writer = (lane & 15u) == 0u;
  auto post_peak4 = [&](const float (&pk)[4], uint32_t tl, int ch) {
    if (CW == 2) {
      uint32_t v0, v1;
      wave_max_quad_halves(__float_as_uint(pk[0]), __float_as_uint(pk[1]), __float_as_uint(pk[2]), __float_as_uint(pk[3]), v0, v1);
      if (quad_writer) {
        uint32_t* slots = &s_pk[(rb + tl + (((lane >> 4) & 1u) * 2u + (lane >> 5))) * kPS];
        slots[0] = v0;
        slots[1] = v1;
      }
      return;
    }
    const uint32_t v = wave_max_quad(__float_as_uint(pk[0]), __float_as_uint(pk[1]), __float_as_uint(pk[2]), __float_as_uint(pk[3]));
    if (quad_writer) s_pk[(rb + tl) * kPS + quad_row + pk_slot(ch)] = v;
  };

  // ---- phase A: the clip loads of the U tracks starting at local index u0 (straight-line per mode) ----
  //  MODE_U    every row is an fp32 unity row: one 16-B load per track, no fp64
  //  MODE_W    fp32 rows, some linearly resampled: 16-B + 4-B load per track (unity rows use the same formula)
  //  MODE_I16  every row is 16-bit PCM at unity speed: one 8-B load per track (half the bytes of fp32)
  //  MODE_I32  every row is 24/32-bit PCM at unity speed: one 16-B load per track
  //  (CL == 2: per track and channel)
  auto issue = [&](auto mode, uint32_t u0, auto& pre) {
    constexpr int MODE = decltype(mode)::value;
    constexpr int D = (int)std::extent_v<std::remove_reference_t<decltype(pre)>>;
#pragma unroll
    for (int u = 0; u < D; u++) {
      URec r;
      if (FULL) {
        r = load_urec(u0 + u);
      } else {
        const DTrackBlock& t = s_tb[u0 + u];
        r.src[0] = t.src[c];
        r.pos = t.pos;
        r.speed = t.speed;
        r.format = t.format;
        r.kind = t.kind & KIND_MASK;
      }
      if constexpr (MODE == MODE_G) {
        load_stride(r.src, r.pos, r.speed, (uint32_t)__builtin_amdgcn_readfirstlane((int)r.format), EXP && r.partial, r.d, r.n, pre[u]);
      } else if constexpr (MODE == MODE_W || MODE == MODE_WN || MODE == MODE_WNU) {
        double prod0;
        const bool part = EXP && r.partial;
        if (part)
          prod0 = __dmul_rn((double)part_cf0(r), r.speed);
        else if (MODE == MODE_WNU)   // whole-block rows of a one-ratio chunk: the hoisted product for the resampled ones, j * 1.0
          prod0 = __builtin_amdgcn_readfirstlane((int)r.kind) == KIND_WINDOW ? up0 : j0d;   // for the unity ones — no multiply
        else
          prod0 = __dmul_rn(j0d, r.speed);
        load_window(r.src, r.pos, prod0, pre[u]);
      } else if constexpr (MODE == MODE_WI || MODE == MODE_WIN || MODE == MODE_WINU) {
        double prod0;
        if (EXP && (LEAN16 || G) && r.partial)    // the lane's first frame inside the stream call (partial 16-bit window rows)
          prod0 = __dmul_rn((double)part_cf0(r), r.speed);
        else if (MODE == MODE_WINU)   // one-ratio chunk: the hoisted product for the resampled rows, j * 1.0 for the unity ones
          prod0 = __builtin_amdgcn_readfirstlane((int)r.kind) == KIND_WINDOW_I16 ? up0 : j0d;
        else
          prod0 = __dmul_rn(j0d, r.speed);
        load_window16(r.src, r.pos, prod0, pre[u]);
      } else if constexpr (MODE == MODE_MW || MODE == MODE_MWN) {
        // unity and window rows of several storage formats (16-bit loops at another rate next to 24-bit stems ...):
        // every row reads 16 B + 4 B at its first sample — the five window samples of a 4-byte format, the low
        // 10 B for a 16-bit window, the four (8 B or 16 B) samples of a unity row — so the loads stay straight-line
        typedef float f4a2 __attribute__((ext_vector_type(4), aligned(2)));
        typedef float f1a2 __attribute__((aligned(2)));
        const int k = __builtin_amdgcn_readfirstlane((int)r.kind);
        const uint32_t sh = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.format) == FMT_I16 ? 1u : 2u;
        // (EXP: a partial stream call starts at the lane's first frame INSIDE the call, clamped into it)
        const uint32_t cf0 = (EXP && r.partial) ? (uint32_t)part_cf0(r) : j0;   // (may be a negative frame: fast_part)
        const double x0 = __dadd_rn(r.pos, __dmul_rn((EXP && r.partial) ? (double)(int32_t)cf0 : j0d, r.speed));   // sampler.cpp:50
        const bool win = k == KIND_WINDOW || k == KIND_WINDOW_I16;
        const int ix0 = win ? (int)x0 : (int)((uint32_t)r.pos + cf0);                     // :51 / :107
        if (active) {
#pragma unroll
          for (int ch = 0; ch < CL; ch++) {
            const char WBX_GLOBAL* p = as_global<char>(r.src[ch]) + ((size_t)(uint32_t)ix0 << sh);
            pre[u].w[ch].v = __builtin_nontemporal_load(reinterpret_cast<const f4a2 WBX_GLOBAL*>(p));
            pre[u].w[ch].w4 = *reinterpret_cast<const f1a2 WBX_GLOBAL*>(p + 16);
          }
        }
        pre[u].ix0 = ix0;
        pre[u].fx0 = (float)__builtin_amdgcn_fract(x0);                                   // :52
      } else if constexpr (MODE == MODE_MU) {
        // unity rows of several storage formats: one 16-B load per row whatever the format (a 16-bit row uses its
        // low half; the rest is its neighbour's samples or the clip's padding), so the loads stay straight-line
        typedef float f4a2 __attribute__((ext_vector_type(4), aligned(2)));
        // sampler.cpp:107 (EXP: the lane's frame inside the stream call — j0 itself for a whole-block record)
        const uint32_t off = (uint32_t)r.pos + ((EXP && r.partial) ? (uint32_t)part_cf0(r) : j0);   // (a negative call frame wraps back: pos >= 4 there)
        const uint32_t sh = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.format) == FMT_I16 ? 1u : 2u;
        if (active) {
#pragma unroll
          for (int ch = 0; ch < CL; ch++) {
            const char WBX_GLOBAL* p = as_global<char>(r.src[ch]) + ((size_t)off << sh);
            pre[u].w[ch].v = __builtin_nontemporal_load(reinterpret_cast<const f4a2 WBX_GLOBAL*>(p));
          }
        }
      } else {
        // sampler.cpp:107 (EXP: the lane's frame inside the stream call — j0 itself for a whole-block record)
        const uint32_t off = (uint32_t)r.pos + ((EXP && r.partial) ? (uint32_t)part_cf0(r) : j0);   // (a negative call frame wraps back: pos >= 4 there)
#pragma unroll
        for (int ch = 0; ch < CL; ch++) {
          if (MODE == MODE_I16) {
            typedef int i2u __attribute__((ext_vector_type(2), aligned(2)));
            const short WBX_GLOBAL* p = as_global<short>(r.src[ch]) + off;
            if (active) {
              const i2u w = __builtin_nontemporal_load(reinterpret_cast<const i2u WBX_GLOBAL*>(p));
              pre[u].w[ch].v.x = __int_as_float(w.x);
              pre[u].w[ch].v.y = __int_as_float(w.y);
            }
          } else {   // MODE_U, MODE_I32: 4 x 32-bit
            const float WBX_GLOBAL* p = as_global<float>(r.src[ch]) + off;
            if (active) pre[u].w[ch].v = __builtin_nontemporal_load(reinterpret_cast<const f4u WBX_GLOBAL*>(p));
          }
        }
      }
    }
  };

  // ---- phase B: render, scale, accumulate — strictly in track order; pk[u][ch] = per-lane max |m| of track u0+u
  auto render = [&](auto mode, uint32_t u0, auto& pre, float (*pk)[CL]) {
    constexpr int MODE = decltype(mode)::value;
    constexpr int D = (int)std::extent_v<std::remove_reference_t<decltype(pre)>>;
#pragma unroll
    for (int u = 0; u < D; u++) {
      URec r;
      if (FULL) {
        r = load_urec(u0 + u);
      } else {
        const DTrackBlock& t = s_tb[u0 + u];
        r.pos = t.pos;
        r.speed = t.speed;
        r.gain = t.gain;
        r.gc[0] = t.g[c];
        r.kind = t.kind & KIND_MASK;
        r.format = t.format;
      }
      const float cg = r.gain;
      const float (&gc)[CL] = r.gc;
      Row m;
      if constexpr (MODE == MODE_G) {
        const uint32_t kk = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.kind);
        const uint32_t ff = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.format);
#pragma unroll
        for (int ch = 0; ch < CL; ch++) m.c[ch] = row_stride(pre[u].v[ch], pre[u].b[ch], pre[u].fx, kk, ff, cg, gc[ch]);
        if (EXP && r.partial) {   // frames outside the stream call contribute an exact +0.0
#pragma unroll
          for (int ch = 0; ch < CL; ch++) {
            m.c[ch].x = and_mask(m.c[ch].x, frame_mask(0u, r.d, r.n));
            m.c[ch].y = and_mask(m.c[ch].y, frame_mask(1u, r.d, r.n));
            m.c[ch].z = and_mask(m.c[ch].z, frame_mask(2u, r.d, r.n));
            m.c[ch].w = and_mask(m.c[ch].w, frame_mask(3u, r.d, r.n));
          }
        }
      } else {
      // G: a partial row read through window-shaped loads, whatever its kind and storage format — its loaded samples
      // normalised to fp32 (the linear path's normalisers sampler.cpp:9-14 for window rows, the unity path's with their
      // clamp :109-144 for unity rows), then the general masked arithmetic.  `packed16`: the 16-bit samples arrived packed
      // (v.x, v.y = samples 0..3, w4 = sample 4 in its low half)
      auto partial_any = [&](auto narrow, const Pre& p, int k, uint32_t fmt, bool packed16) {
        const bool unity = k == KIND_UNITY || k == KIND_UNITY_I16 || k == KIND_UNITY_I32;
        Pre f = p;
        if (packed16) {
          f = unpack16(p, unity);
        } else if (fmt != FMT_F32) {
#pragma unroll
          for (int ch = 0; ch < CL; ch++) {
            if (unity) {
              f.w[ch].v = norm_i32(p.w[ch].v, fmt);
            } else {
              const double norm = fmt == FMT_I24 ? 1.0 / 8388607.0 : 1.0 / 2147483647.0;
              f.w[ch].v.x = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].v.x));
              f.w[ch].v.y = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].v.y));
              f.w[ch].v.z = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].v.z));
              f.w[ch].v.w = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].v.w));
              f.w[ch].w4 = (float)__dmul_rn(norm, (double)__float_as_int(p.w[ch].w4));
            }
          }
        }
        if (fast_part(r)) {   // the cheap form (fast_part above): the unmasked arithmetic of a whole-wave call + frame masks
          Row mm;
          if (unity) {
#pragma unroll
            for (int ch = 0; ch < CL; ch++) mm.c[ch] = row_f32(f.w[ch].v, cg, gc[ch]);
          } else {
            mm = row_window_at(narrow, std::true_type{}, std::false_type{}, f, r.pos, r.speed, (double)r.d, cg, gc);
          }
          if (!(r.d <= wave_base && r.d + r.n >= wave_end)) mask_row(mm, r.d, r.n);
          return mm;
        }
        return row_window_masked(f, r.pos, r.speed, unity, r.d, r.n, cg, gc);
      };
      // per channel: the rows whose arithmetic has nothing to share between the channels of a frame
      auto each_of = [&](const auto& pu, auto f) {
#pragma unroll
        for (int ch = 0; ch < CL; ch++) m.c[ch] = f(pu.w[ch], gc[ch]);
      };
      auto each = [&](auto f) { each_of(pre[u], f); };
      auto each_f32 = [&]() { each([&](const Win& w, float g) { return row_f32(w.v, cg, g); }); };
      auto each_i16 = [&]() { each([&](const Win& w, float g) { return row_i16(__float_as_int(w.v.x), __float_as_int(w.v.y), cg, g); }); };
      auto each_i32 = [&](uint32_t fmt) { each([&](const Win& w, float g) { return row_i32(w.v, fmt, cg, g); }); };
      if constexpr (MODE == MODE_W || MODE == MODE_WN || MODE == MODE_WNU) {
        const int k = __builtin_amdgcn_readfirstlane((int)r.kind);
        const uint32_t fmt = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.format);
        constexpr std::integral_constant<bool, MODE != MODE_W> narrow{};
        constexpr std::integral_constant<bool, MODE == MODE_WNU> uni{};
        if (EXP && r.partial) {   // a stream call that covers part of the block (wave-uniform)
          if (r.d >= wave_end || r.d + r.n <= wave_base) {   // ... none of this wave's 256 frames: an exact +0.0
#pragma unroll
            for (int ch = 0; ch < CL; ch++) m.c[ch] = f4{0.0f, 0.0f, 0.0f, 0.0f};
          } else if (G && (fmt != FMT_F32 || k == KIND_UNITY_I32)) {   // 24 / 32-bit PCM (G instances only)
            if constexpr (G) m = partial_any(narrow, pre[u], k, fmt, false);
          } else if (const bool all = r.d <= wave_base && r.d + r.n >= wave_end; all || fast_part(r)) {
            // ... all of this wave's frames — or some, of a call that starts deep enough in its clip: the same arithmetic + masks
            if (k == KIND_WINDOW)
              m = row_window_at(narrow, std::true_type{}, std::false_type{}, pre[u], r.pos, r.speed, (double)r.d, cg, gc);
            else
              each_f32();
            if (!all) mask_row(m, r.d, r.n);
          } else {
            m = row_window_masked(pre[u], r.pos, r.speed, k != KIND_WINDOW, r.d, r.n, cg, gc);
          }
        } else if (k == KIND_WINDOW) {
          if (G && fmt != FMT_F32) {   // 24/32-bit PCM through the same window loads (G instances only)
            if constexpr (G) m = row_window32(narrow, uni, pre[u], fmt, r.pos, r.speed, cg, gc);
          } else {
            m = row_window_at(narrow, std::false_type{}, uni, pre[u], r.pos, r.speed, 0.0, cg, gc);
          }
        } else if (G && k == KIND_UNITY_I32) {
          if constexpr (G) each_i32(fmt);
        } else {
          each_f32();   // KIND_UNITY (also pre-rendered rows, silent and padding records)
        }
      } else if constexpr (MODE == MODE_WI || MODE == MODE_WIN || MODE == MODE_WINU) {
        const bool win = __builtin_amdgcn_readfirstlane((int)r.kind) == KIND_WINDOW_I16;
        constexpr std::integral_constant<bool, MODE != MODE_WI> narrow{};
        if (EXP && (LEAN16 || G) && r.partial) {   // a stream call that covers part of the block (wave-uniform)
          if (r.d >= wave_end || r.d + r.n <= wave_base) {   // ... none of this wave's frames: an exact +0.0
#pragma unroll
            for (int ch = 0; ch < CL; ch++) m.c[ch] = f4{0.0f, 0.0f, 0.0f, 0.0f};
          } else if (const bool all = r.d <= wave_base && r.d + r.n >= wave_end; all || fast_part(r)) {   // ... all of them (or: see fast_part)
            if (win)
              m = row_window16_shifted(narrow, pre[u], r.pos, r.speed, (double)r.d, cg, gc);
            else
              each_i16();
            if (!all) mask_row(m, r.d, r.n);
          } else {
            m = row_window16_masked(pre[u], r.pos, r.speed, !win, r.d, r.n, cg, gc);
          }
        } else if (win)
          m = row_window16(narrow, std::integral_constant<bool, MODE == MODE_WINU>{}, pre[u], r.pos, r.speed, cg, gc);
        else
          each_i16();   // unity, silent, padding
      } else if constexpr (MODE == MODE_MW || MODE == MODE_MWN) {
        const int k = __builtin_amdgcn_readfirstlane((int)r.kind);
        const uint32_t fmt = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.format);
        constexpr std::integral_constant<bool, MODE == MODE_MWN> narrow{};
        if (EXP && r.partial) {   // a stream call that covers part of the block
          if (r.d >= wave_end || r.d + r.n <= wave_base) {   // ... none of this wave's frames: an exact +0.0
#pragma unroll
            for (int ch = 0; ch < CL; ch++) m.c[ch] = f4{0.0f, 0.0f, 0.0f, 0.0f};
          } else {
            Pre q = pre[u];
            const bool p16 = k == KIND_WINDOW_I16 || k == KIND_UNITY_I16;
            if (p16) {
#pragma unroll
              for (int ch = 0; ch < CL; ch++) q.w[ch].w4 = pre[u].w[ch].v.z;   // the fifth sample of a 16-bit window: low half of the third dword
            }
            m = partial_any(narrow, q, k, fmt, p16);
          }
        } else if (k == KIND_WINDOW_I16) {
          Pre q = pre[u];
#pragma unroll
          for (int ch = 0; ch < CL; ch++) q.w[ch].w4 = pre[u].w[ch].v.z;   // the fifth sample of a 16-bit window is the low half of the load's third dword
          m = row_window16(narrow, std::false_type{}, q, r.pos, r.speed, cg, gc);
        } else if (k == KIND_WINDOW) {
          if (fmt == FMT_F32)
            m = row_window(narrow, pre[u], r.pos, r.speed, cg, gc);
          else
            m = row_window32(narrow, std::false_type{}, pre[u], fmt, r.pos, r.speed, cg, gc);
        } else if (k == KIND_UNITY_I16) {
          each_i16();
        } else if (k == KIND_UNITY_I32) {
          each_i32(fmt);
        } else {
          each_f32();
        }
      } else {
        // unity rows: MODE_U fp32, MODE_I16 / MODE_I32 integer PCM, MODE_MU any of them by the record's kind
        const int k = MODE == MODE_U ? KIND_UNITY : MODE == MODE_I16 ? KIND_UNITY_I16 : MODE == MODE_I32 ? KIND_UNITY_I32
                                                  : __builtin_amdgcn_readfirstlane((int)r.kind);
        const uint32_t fmt = (MODE == MODE_I32 || MODE == MODE_MU) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)r.format) : 0u;
        if (EXP && r.partial && !(r.d <= wave_base && r.d + r.n >= wave_end)) {   // a stream call that covers part of the block
          if (r.d >= wave_end || r.d + r.n <= wave_base) {   // ... none of this wave's frames: an exact +0.0
#pragma unroll
            for (int ch = 0; ch < CL; ch++) m.c[ch] = f4{0.0f, 0.0f, 0.0f, 0.0f};
          } else if (fast_part(r)) {   // ... some, of a call that starts deep enough in its clip: the plain row + masks
            if (k == KIND_UNITY_I16)
              each_i16();
            else if (k == KIND_UNITY_I32)
              each_i32(fmt);
            else
              each_f32();
            mask_row(m, r.d, r.n);
          } else if (k == KIND_UNITY_I16) {   // ... some: normalise the four loaded samples, then select and mask as for fp32
            each([&](const Win& w, float g) {
              return row_f32_masked(norm_i16(__float_as_int(w.v.x), __float_as_int(w.v.y)), r.d, r.n, cg, g);
            });
          } else if (k == KIND_UNITY_I32) {
            each([&](const Win& w, float g) { return row_f32_masked(norm_i32(w.v, fmt), r.d, r.n, cg, g); });
          } else {
            each([&](const Win& w, float g) { return row_f32_masked(w.v, r.d, r.n, cg, g); });
          }
        } else if (k == KIND_UNITY_I16) {
          each_i16();
        } else if (k == KIND_UNITY_I32) {
          each_i32(fmt);
        } else {
          each_f32();
        }
      }
      }
      add_row(m, pk[u]);
    }
  };

  // two-stage software pipeline over batches of D tracks (trip count is uniform: padded with null records);
  // one iteration covers kIter = lcm(2D, 4) tracks so that the peaks can be reduced four tracks at a time.
  // D = U except for the per-frame-tap mode, whose rows hold 12 registers each: depth 1 keeps it out of scratch
  auto pipeline = [&](auto mode, uint32_t cn) {
    constexpr int D = decltype(mode)::value == MODE_G ? 1 : U;
    constexpr int kIter = (2 * D) % 4 == 0 ? 2 * D : 4;
    std::conditional_t<decltype(mode)::value == MODE_G, PreG, Pre> pa[D], pb[D];
    issue(mode, 0, pa);
    for (uint32_t u0 = 0; u0 < cn; u0 += kIter) {
      float pk[kIter][CL];
#pragma unroll
      for (int h = 0; h < kIter; h += 2 * D) {
        issue(mode, u0 + h + D, pb);
        render(mode, u0 + h, pa, pk + h);
        issue(mode, u0 + h + 2 * D, pa);
        render(mode, u0 + h + D, pb, pk + h + D);
      }
#pragma unroll
      for (int ch = 0; ch < CL; ch++) {
        if (FULL) {
#pragma unroll
          for (int q = 0; q < kIter; q += 4) {
            const float quad[4] = {pk[q][ch], pk[q + 1][ch], pk[q + 2][ch], pk[q + 3][ch]};
            post_peak4(quad, u0 + q, ch);
          }
        } else {
#pragma unroll
          for (int q = 0; q < kIter; q++) post_peak(pk[q][ch], u0 + q, ch);
        }
      }
    }
  };

  // a chunk that mixes storage formats (e.g. a pre-rendered fp32 boundary row among 16-bit tracks): one row at
  // a time with a wave-uniform dispatch on the row kind — correct for any combination, not software-pipelined
  auto mixed = [&](uint32_t cn) {
    for (uint32_t tl = 0; tl < cn; tl++) {
      const DTrackBlock& r = s_tb[rb + tl];
      const int k = __builtin_amdgcn_readfirstlane((int)(r.kind & KIND_MASK));
      const float cg = r.gain;
      const uint32_t off = (uint32_t)r.pos + j0;
      Row m;
      float gcs[CL];
      const void* srcs[CL];
#pragma unroll
      for (int ch = 0; ch < CL; ch++) {
        const uint32_t cc = CL == 2 ? (uint32_t)ch : c;
        gcs[ch] = r.g[cc];
        srcs[ch] = r.src[cc];
      }
      if (k == KIND_WINDOW) {
        Pre p;
        load_window(srcs, r.pos, __dmul_rn(j0d, r.speed), p);
        m = row_window(std::false_type{}, p, r.pos, r.speed, cg, gcs);
      } else if (W16 && (k == KIND_WINDOW_I16 || (EXP && LEAN16 && k == KIND_UNITY_I16 && (r.kind & KIND_PARTIAL)))) {
        // 16-bit rows through the window loads; family 2 also meets partial ones here (a chunk that holds a pre-rendered
        // fp32 row next to the masked rows of its other tracks): every wave takes the masked arithmetic
        Pre p;
        if (EXP && LEAN16 && (r.kind & KIND_PARTIAL)) {
          const uint32_t d = r.dst_start, n = r.len;
          load_window16(srcs, r.pos, __dmul_rn((double)call_frame(0u, d, n), r.speed), p);
          m = row_window16_masked(p, r.pos, r.speed, k != KIND_WINDOW_I16, d, n, cg, gcs);
        } else {
          load_window16(srcs, r.pos, __dmul_rn(j0d, r.speed), p);
          m = row_window16(std::false_type{}, std::false_type{}, p, r.pos, r.speed, cg, gcs);
        }
      } else {
#pragma unroll
        for (int ch = 0; ch < CL; ch++) {
          if (k == KIND_UNITY_I16) {
            typedef int i2u __attribute__((ext_vector_type(2), aligned(2)));
            i2u w = {0, 0};
            if (active) w = *reinterpret_cast<const i2u WBX_GLOBAL*>(as_global<short>(srcs[ch]) + off);
            m.c[ch] = row_i16(w.x, w.y, cg, gcs[ch]);
          } else {
            f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (active) v = *reinterpret_cast<const f4u WBX_GLOBAL*>(as_global<float>(srcs[ch]) + off);
            m.c[ch] = (k == KIND_UNITY_I32) ? row_i32(v, r.format, cg, gcs[ch]) : row_f32(v, cg, gcs[ch]);
          }
        }
      }
      float pk[CL];
      add_row(m, pk);
#pragma unroll
      for (int ch = 0; ch < CL; ch++) post_peak(pk[ch], tl, ch);
    }
  };

  // A group is walked in chunks of up to kSt tracks.  A LONG walk — one workgroup adding ALL tracks of its block in track
  // order, the reference's own summation order (engine.cpp:1600-1617) — passes dozens of chunk seams, each a phase
  // without clip loads in flight (rows -> templates -> LDS, then the pipeline fills again).  Two things keep the seams
  // short: the rows and routing entries of the NEXT chunk are fetched while this one is staged (a seam then costs one
  // round trip for the templates, not three dependent ones), and the template loads of a chunk are all issued before the
  // first is waited for.  (Measured, tools/ab.py seams: a walk with NO staging at its seams at all would be 3-4 % faster
  // on fp32 sessions, 10 % on 16-bit resampled ones; shortening the first chunk by a per-workgroup phase so that the
  // workgroups of a CU meet their seams apart: no effect.)
  uint32_t chunk0 = 0u, cn2 = 0u;
  int mode = MODE_U;
  for (uint32_t chunk_i = 0u; chunk0 < grp.count; chunk_i++) {
    const uint32_t left = grp.count - chunk0;
    const uint32_t cn = left < kSt ? left : kSt;     // tracks of this chunk
    cn2 = cn;                                         // staged rows of this chunk
    __syncthreads();
    if (tid == 0u) s_shape = 0;
    // stage the group's records (per-track gain / pan / resample parameters) in LDS: 4 x 16 B per record.
    // A record = the template its 16-B plan row points at, with the row's position patched in when the template
    // is shared by a run of blocks; silent rows become all-zero records (kind 0).
    if constexpr (XP) {
      // The same two passes per SUB-BLOCK (every sub-block is a block of its own: other plan rows, other pairs, another
      // number of staged rows).  Pass 1, one lane per (sub-block, track): routing entry -> plan row (kept in LDS for pass 2),
      // a pair counts twice; the prefix sum runs over the 32-lane halves of the fetching waves (a sub-block's tracks are one,
      // two or four halves).  Pass 2 copies the template quads of all sub-blocks through the maps, all loads in flight at once.
      constexpr uint32_t kHalves = kSt / 32u;   // 32-lane halves per sub-block
      // (as below: the rows of the NEXT chunk and the routing entries of the one after it are fetched while this chunk is
      //  staged — a packed workgroup walks four chunks of 32 tracks, each seam three dependent round trips otherwise)
      DRow* rows_cur = s_rows + (chunk_i & 1u) * (SB * kSt);
      DRow* rows_nxt = s_rows + ((chunk_i & 1u) ^ 1u) * (SB * kSt);
      const uint32_t n0 = chunk0 + cn, nleft = grp.count - n0, ncn = nleft < kSt ? nleft : kSt;       // the next chunk
      const uint32_t m0 = n0 + ncn, mleft = grp.count - m0, mcn = mleft < kSt ? mleft : kSt;         // the one after it
      uint32_t before = 0u;
      bool is_pair = false;
      const uint32_t psb = tid / kSt, pt = tid - psb * kSt;   // (lanes below SB * kSt)
      DRow nrow;
      nrow.pos = 0.0;
      nrow.tmpl = 0xFFFFFFFFu;
      nrow.flags = ROW_SILENT;
      uint32_t nord = 0u;
      if (tid < SB * kSt) {
        DRow row = nrow;
        const uint32_t bb = bx * SB + psb;
        const bool bval = bb < a.n_blocks;
        uint32_t o1 = 0u;
        if (chunk_i == 0u) {
          uint32_t o0 = 0u;
          if (pt < cn) o0 = a.order[grp.first + pt];
          if (pt < ncn) o1 = a.order[grp.first + n0 + pt];
          if (pt < cn && bval) row = a.rows[(size_t)bb * N + o0];
          *reinterpret_cast<uint4*>(&rows_cur[tid]) = *reinterpret_cast<const uint4*>(&row);
        } else {
          row = rows_cur[tid];
          o1 = s_ord[pt];
        }
        if (pt < ncn && bval) nrow = a.rows[(size_t)bb * N + o1];
        if (psb == 0u && pt < mcn) nord = a.order[grp.first + m0 + pt];
        is_pair = (row.flags & (ROW_PAIR | ROW_SILENT)) == ROW_PAIR;
        const unsigned long long half = (lane & 32u) ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
        const unsigned long long bal = __ballot(is_pair) & half;
        before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if ((lane & 31u) == 0u) s_hp[tid >> 5] = (uint32_t)__popcll(bal);
      }
      __syncthreads();
      if (tid < SB * kSt) {
        uint32_t off = pt + before, total = cn;
#pragma unroll
        for (uint32_t h = 0; h < kHalves; h++) {
          const uint32_t n = s_hp[psb * kHalves + h];
          if (psb * kHalves + h < (tid >> 5)) off += n;
          total += n;
        }
        if (pt < cn) {
          s_off[psb * (kSt + 1u) + pt] = (uint16_t)off;
          s_map[psb * 2u * kSt + off] = (uint16_t)pt;
          if (is_pair) s_map[psb * 2u * kSt + off + 1u] = (uint16_t)(pt | 0x8000u);
        }
        if (pt == 0u) {
          s_off[psb * (kSt + 1u) + cn] = (uint16_t)total;
          s_tot[psb] = total;
        }
      }
      __syncthreads();
      cn2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tot[sub]);   // this wave's sub-block
      constexpr uint32_t kQ = (SB * kRecs * 4u + kT - 1u) / kT;
      uint4 wq[kQ];
      uint32_t tq = tid;
      asm volatile("" : "+v"(tq));
#pragma unroll
      for (uint32_t it = 0; it < kQ; it++) {
        const uint32_t i = tq + it * kT, sbq = i / (kRecs * 4u), r4 = i - sbq * (kRecs * 4u), rec = r4 >> 2, q = r4 & 3u;
        const bool in = sbq < (uint32_t)SB && rec < s_tot[sbq < (uint32_t)SB ? sbq : 0u];
        const uint32_t mp = s_map[in ? sbq * 2u * kSt + rec : 0u];
        const DRow row = rows_cur[(in ? sbq * kSt : 0u) + (mp & 0x7FFFu)];
        const bool live = in && !(row.flags & ROW_SILENT);
        // (a lane without a live record reads template 0: the load stays unconditional, its result is dropped)
        wq[it] = reinterpret_cast<const uint4*>(a.tmpl + (live ? row.tmpl + (mp >> 15) : 0u))[q];
      }
#pragma unroll
      for (uint32_t it = 0; it < kQ; it++) {
        const uint32_t i = tq + it * kT, sbq = i / (kRecs * 4u), r4 = i - sbq * (kRecs * 4u), rec = r4 >> 2, q = r4 & 3u;
        if (i < SB * kRecs * 4u) {
          uint4 w = {0u, 0u, 0u, 0u};
          if (rec < s_tot[sbq]) {
            const uint32_t mp = s_map[sbq * 2u * kSt + rec];
            const DRow row = rows_cur[sbq * kSt + (mp & 0x7FFFu)];
            if (!(row.flags & ROW_SILENT)) {
              w = wq[it];
              if (q == 1u && (row.flags & ROW_POS)) {          // quad 1 = {pos, speed}
                const uint2 pb = *reinterpret_cast<const uint2*>(&row.pos);
                w.x = pb.x;
                w.y = pb.y;
              }
            }
          }
          reinterpret_cast<uint4*>(s_tb)[i] = w;
        }
      }
      if (tid < SB * kSt) {   // (arrived with the templates: loads return in order)
        *reinterpret_cast<uint4*>(&rows_nxt[tid]) = *reinterpret_cast<const uint4*>(&nrow);
        if (psb == 0u) s_ord[pt] = nord;
      }
    } else if constexpr (EXP) {
      // Rows may be ROW_PAIRs (a clip boundary inside the block: two single-segment templates; only with
      // MixArgs::masked_rows).  Pass 1, one lane per track: the 16-B plan row (fetched by the previous chunk's staging,
      // except for the first chunk), a pair counts as two staged rows, wave-ballot prefix sum -> the track's first
      // staged row; the loads for the next chunk's rows and the routing entries of the one after it are issued here and
      // land in LDS at the end of the staging.  Pass 2 copies the records through the staged-row -> (track, record) map.
      DRow* rows_cur = s_rows + (chunk_i & 1u) * kSt;
      DRow* rows_nxt = s_rows + ((chunk_i & 1u) ^ 1u) * kSt;
      const uint32_t n0 = chunk0 + cn, nleft = grp.count - n0, ncn = nleft < kSt ? nleft : kSt;       // the next chunk
      const uint32_t m0 = n0 + ncn, mleft = grp.count - m0, mcn = mleft < kSt ? mleft : kSt;         // the one after it
      uint32_t before = 0u;
      bool is_pair = false;
      DRow nrow;
      nrow.pos = 0.0;
      nrow.tmpl = 0xFFFFFFFFu;
      nrow.flags = ROW_SILENT;
      uint32_t nord = 0u;
      if (tid < kSt) {
        DRow row = nrow;
        uint32_t o1 = 0u;
        if (chunk_i == 0u) {
          uint32_t o0 = 0u;
          if (tid < cn) o0 = a.order[grp.first + tid];
          if (tid < ncn) o1 = a.order[grp.first + n0 + tid];
          if (tid < cn) row = a.rows[(size_t)bx * N + o0];
        } else {
          row = rows_cur[tid];
          o1 = s_ord[tid];
        }
        if (tid < ncn) nrow = a.rows[(size_t)bx * N + o1];
        if (tid < mcn) nord = a.order[grp.first + m0 + tid];
        if (chunk_i == 0u) *reinterpret_cast<uint4*>(&rows_cur[tid]) = *reinterpret_cast<const uint4*>(&row);
        is_pair = (row.flags & (ROW_PAIR | ROW_SILENT)) == ROW_PAIR;
        const unsigned long long bal = __ballot(is_pair);
        before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0u) s_wpairs[tid >> 6] = (uint32_t)__popcll(bal);
        if (kSt <= 64u && tid == 0u) s_wpairs[1] = 0u;   // (one wave fetches all rows)
      }
      __syncthreads();
      if (tid < cn) {
        const uint32_t off = tid + before + (tid >= 64u ? s_wpairs[0] : 0u);
        s_off[tid] = (uint16_t)off;
        s_map[off] = (uint16_t)tid;
        if (is_pair) s_map[off + 1u] = (uint16_t)(tid | 0x8000u);
      }
      if (tid == 0u) {
        const uint32_t total = cn + s_wpairs[0] + s_wpairs[1];
        s_off[cn] = (uint16_t)total;
        s_wpairs[2] = total;
      }
      __syncthreads();
      cn2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_wpairs[2]);
      // the template quads of all staged rows: every load is issued before the first one is waited for (one round trip
      // per chunk; a lane holds up to kQ quads for a moment)
      constexpr uint32_t kQ = (kRecs * 4u + kT - 1u) / kT;
      const uint32_t nq = (cn2 * 4u + kT - 1u) / kT;   // iterations that reach a staged row (wave-uniform)
      uint4 wq[kQ];
      uint32_t tq = tid;   // (opaque: the per-iteration indices below are cheap to recompute, hoisted out of the chunk loop
      asm volatile("" : "+v"(tq));   //  they would sit in registers — spilled ones — through the whole pipeline)
#pragma unroll
      for (uint32_t it = 0; it < kQ; it++) {
        wq[it] = uint4{0u, 0u, 0u, 0u};
        if (it < nq) {
          const uint32_t i = tq + it * kT, rec = i >> 2, q = i & 3u;
          const bool in = rec < cn2;
          const uint32_t mp = s_map[in ? rec : 0u];
          const DRow row = rows_cur[mp & 0x7FFFu];
          const bool live = in && !(row.flags & ROW_SILENT);
          // (a lane without a live record reads template 0: the load stays unconditional, its result is dropped)
          wq[it] = reinterpret_cast<const uint4*>(a.tmpl + (live ? row.tmpl + (mp >> 15) : 0u))[q];
        }
      }
#pragma unroll
      for (uint32_t it = 0; it < kQ; it++) {
        const uint32_t i = tq + it * kT, rec = i >> 2, q = i & 3u;
        if (i < kRecs * 4u) {
          uint4 w = {0u, 0u, 0u, 0u};
          if (it < nq && rec < cn2) {
            const uint32_t mp = s_map[rec];
            const DRow row = rows_cur[mp & 0x7FFFu];
            if (!(row.flags & ROW_SILENT)) {
              w = wq[it];
              if (q == 1u && (row.flags & ROW_POS)) {          // quad 1 = {pos, speed}
                const uint2 pb = *reinterpret_cast<const uint2*>(&row.pos);
                w.x = pb.x;
                w.y = pb.y;
              }
            }
          }
          reinterpret_cast<uint4*>(s_tb)[i] = w;
        }
      }
      if (tid < kSt) {   // (arrived with the templates: loads return in order)
        *reinterpret_cast<uint4*>(&rows_nxt[tid]) = *reinterpret_cast<const uint4*>(&nrow);
        s_ord[tid] = nord;
      }
    } else {
    for (uint32_t i = tid; i < SB * kRecs * 4u; i += kT) {
      const uint32_t sb = SB > 1 ? i / (kRecs * 4u) : 0u;
      const uint32_t rec = (i - sb * kRecs * 4u) >> 2, q = i & 3u;
      const uint32_t bb = bx * SB + sb;
      uint4 w = {0u, 0u, 0u, 0u};
      if (rec < cn && (SB == 1 || bb < a.n_blocks)) {
        const uint32_t track = a.order[grp.first + chunk0 + rec];
        const DRow row = a.rows[(size_t)bb * N + track];
        if (!(row.flags & ROW_SILENT)) {
          w = reinterpret_cast<const uint4*>(a.tmpl + row.tmpl)[q];
          if (q == 1u && (row.flags & ROW_POS)) {          // quad 1 = {pos, speed}
            const uint2 pb = *reinterpret_cast<const uint2*>(&row.pos);
            w.x = pb.x;
            w.y = pb.y;
          }
        }
      }
      reinterpret_cast<uint4*>(s_tb)[i] = w;
    }
    }
    for (uint32_t i = tid; i < SB * kRecs * kPS; i += kT) s_pk[i] = 0u;
    if (FULL && lane == 0u) {
      if (CL == 2) {   // slot channel * waves + wave
        s_wc[wave] = 0u;
        s_wc[kWaves + wave] = 1u;
        if (2u * kWaves < kPS && wave == 0u) {   // (one-wave workgroups: slots 2 and 3 stay empty)
          s_wc[2] = 0xFFFFFFFFu;
          s_wc[3] = 0xFFFFFFFFu;
        }
      } else {
        s_wc[wave] = sub * C + c;
      }
    }
    __syncthreads();
    // which row shapes does this chunk hold?  (bit 0 fp32 unity, 1 fp32 window, 2 16-bit, 3 24/32-bit)
    int shape = 0;
    for (uint32_t i0 = tid; i0 < SB * (XP ? kMaxRows : cn2); i0 += kT) {
      if (XP && i0 % kMaxRows >= s_tot[XP ? i0 / kMaxRows : 0u]) continue;   // (XP: every sub-block has its own number of staged rows)
      const uint32_t i = XP ? (i0 / kMaxRows) * kRecs + i0 % kMaxRows : SB > 1 ? (i0 / cn2) * kRecs + i0 % cn2 : i0;
      const int k = s_tb[i].kind & KIND_MASK;
      shape |= k == KIND_UNITY ? 1 : k == KIND_WINDOW ? (s_tb[i].format == FMT_F32 ? 2 : 128) : k == KIND_UNITY_I16 ? 4 : k == KIND_UNITY_I32 ? 8
               : k == KIND_STRIDE ? 64 : k == KIND_WINDOW_I16 ? 32 : 0;
      if ((k == KIND_WINDOW || k == KIND_WINDOW_I16) && !(s_tb[i].speed >= kNarrowSpeed)) shape |= 16;   // needs the general tap selection
    }
    if (shape) atomicOr(&s_shape, shape);
    __syncthreads();
    const int shapes = __builtin_amdgcn_readfirstlane(s_shape);
    const int has_f32 = shapes & 3, has_win = shapes & 2, has_i16 = shapes & 4, has_i32 = shapes & 8, has_wide = shapes & 16;
    // (G instances only: sessions without such clips run the instance that does not carry these modes)
    const int has_stride = STRIDE ? (shapes & 64) : 0;    // per-frame taps
    const int has_win16 = W16 ? (shapes & 32) : 0;   // 16-bit PCM window rows
    const int has_win32 = G ? (shapes & 128) : 0;    // 24/32-bit PCM window rows
    if (LEAN16) {
      if (has_win16)
        mode = (!has_f32 && !has_i32) ? (has_wide ? MODE_WI : us > 0.0 ? MODE_WINU : MODE_WIN) : MODE_MIXED;
      else if (has_i16)
        mode = (!has_f32 && !has_i32) ? MODE_I16 : (has_win || has_i32) ? MODE_MIXED : MODE_MU;
      else
        mode = (has_win || has_i32) ? MODE_MIXED : MODE_U;
    } else if (has_stride) {
      mode = MODE_G;             // reads every kind and format, whatever else the chunk holds
    } else if (has_win16) {
      mode = (!has_f32 && !has_i32 && !has_win32) ? (has_wide ? MODE_WI : us > 0.0 ? MODE_WINU : MODE_WIN) : (has_wide ? MODE_MW : MODE_MWN);
    } else if (has_i16) {
      mode = (has_win32 || (G && has_win)) ? (has_wide ? MODE_MW : MODE_MWN)
             : (!has_i32 && !has_f32) ? MODE_I16 : has_win ? MODE_MIXED : MODE_MU;
    } else if (has_win || has_win32) {
      // fp32 unity + window rows; in G instances also 24/32-bit PCM unity + window rows (all 4-byte containers)
      mode = (G || !has_i32) ? (has_wide ? MODE_W : us > 0.0 ? MODE_WNU : MODE_WN) : MODE_MIXED;   // (the position arithmetic does not look at the storage format)
    } else {
      mode = !has_i32 ? MODE_U : !has_f32 ? MODE_I32 : MODE_MU;   // unity rows of several formats
    }
    // silent records and the padding up to a whole number of batches become "read the zero page, gain 0" rows
    // of the chunk's own shape, so that the load phase stays straight-line
    for (uint32_t i = tid; i < SB * kRecs; i += kT) {
      DTrackBlock& r = s_tb[i];
      // (KIND_GENERIC: only when the one-block callback skipped the pre-render pass on the expectation of an empty queue;
      //  the host then repeats pre-render + mix for that block — here the record counts as silence)
      if ((SB > 1 ? i % kRecs : i) >= (XP ? s_tot[XP ? i / kRecs : 0u] : cn2) || (r.kind & KIND_MASK) == KIND_SILENT || (r.kind & KIND_MASK) == KIND_GENERIC) {
        r.src[0] = a.zero_page;
        r.src[1] = a.zero_page;
        r.pos = 0.0;
        r.speed = 1.0;
        r.gain = 0.0f;
        r.g[0] = 0.0f;
        r.g[1] = 0.0f;
        r.dst_start = 0;                 // (a whole-block row for the masked-row arithmetic of the EXP instances)
        r.len = (uint16_t)F;
        const bool m16 = mode == MODE_I16 || mode == MODE_WI || mode == MODE_WIN || mode == MODE_WINU;
        r.format = m16 ? FMT_I16 : mode == MODE_I32 ? FMT_I32 : FMT_F32;
        r.kind = m16 ? KIND_UNITY_I16 : mode == MODE_I32 ? KIND_UNITY_I32 : KIND_UNITY;
      }
    }
    __syncthreads();

    if (chain_in && chunk_i == 0u) {
      if (tid == 0u) {
        uint32_t spins = 0u;
        while ((chain_seen & ~0xFu) != chain_tag && spins < 400000u) {   // (bounded: ~0.2 s; a give-up is reported, never a hang)
          __builtin_amdgcn_s_sleep(8);
          chain_seen = __hip_atomic_load(chain_word - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          spins++;
        }
        const bool chain_ok = (chain_seen & ~0xFu) == chain_tag;
        // (the per-plan word goes with its plan buffer three renders later; the context's own word keeps the failure until a
        //  host call has reported it: wbx_render_status and every fetch / sync that returns a status)
        if (!chain_ok && a.chain_status) {
          atomicOr(a.chain_status, 32u);
          atomicOr(a.chain_sticky, 32u);
        }
        if (chain_ok && (chain_seen & 0xFu) != my_xcc && a.chain_status) {
          atomicOr(a.chain_status, 64u);
          atomicOr(a.chain_sticky, 64u);
        }
      }
      __syncthreads();
      if (owns && bvalid) {
#pragma unroll
        for (int ch = 0; ch < CL; ch++) {
          const uint32_t* src = reinterpret_cast<const uint32_t*>(
              a.partial + (((size_t)b * a.n_groups + (g - 1u)) * C + (CL == 2 ? (uint32_t)ch : c)) * F + j0);
          acc.c[ch].x = __uint_as_float(__hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // (sc1: past the L1)
          acc.c[ch].y = __uint_as_float(__hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          acc.c[ch].z = __uint_as_float(__hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          acc.c[ch].w = __uint_as_float(__hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
      }
    }

    switch (mode) {
      case MODE_U: pipeline(std::integral_constant<int, MODE_U>{}, cn2); break;
      case MODE_W:
        if constexpr (!LEAN16) pipeline(std::integral_constant<int, MODE_W>{}, cn2);
        break;
      case MODE_WN:
        if constexpr (!LEAN16) pipeline(std::integral_constant<int, MODE_WN>{}, cn2);
        break;
      case MODE_WNU:
        if constexpr (!LEAN16) pipeline(std::integral_constant<int, MODE_WNU>{}, cn2);
        break;
      case MODE_G:
        if constexpr (STRIDE) pipeline(std::integral_constant<int, MODE_G>{}, cn2);
        break;
      case MODE_WI:
        if constexpr (W16) pipeline(std::integral_constant<int, MODE_WI>{}, cn2);
        break;
      case MODE_WIN:
        if constexpr (W16) pipeline(std::integral_constant<int, MODE_WIN>{}, cn2);
        break;
      case MODE_WINU:
        if constexpr (W16) pipeline(std::integral_constant<int, MODE_WINU>{}, cn2);
        break;
      case MODE_MU: pipeline(std::integral_constant<int, MODE_MU>{}, cn2); break;
      case MODE_MW:
        if constexpr (G) pipeline(std::integral_constant<int, MODE_MW>{}, cn2);
        break;
      case MODE_MWN:
        if constexpr (G) pipeline(std::integral_constant<int, MODE_MWN>{}, cn2);
        break;
      case MODE_I16: pipeline(std::integral_constant<int, MODE_I16>{}, cn2); break;
      case MODE_I32:
        if constexpr (!LEAN16) pipeline(std::integral_constant<int, MODE_I32>{}, cn2);
        break;
      default: mixed(cn2); break;
    }

    __syncthreads();
    for (uint32_t i0 = tid; i0 < SB * cn * C; i0 += kT) {
      const uint32_t sb = SB > 1 ? i0 / (cn * C) : 0u;
      const uint32_t i = i0 - sb * cn * C;
      const uint32_t rec = i / C, ch = i - rec * C;
      const uint32_t bb = bx * SB + sb;
      if (SB > 1 && bb >= a.n_blocks) continue;
      const uint32_t track = a.order[grp.first + chunk0 + rec];
      uint32_t* dst = reinterpret_cast<uint32_t*>(a.peaks) + ((size_t)bb * N + track) * C + ch;
      uint32_t pk = 0u;   // peaks are non-negative floats: uint order == float order
      if (FULL && CW == 2) {
        const uint32_t r0 = EXP ? s_off[sb * (kSt + 1u) + rec] : rec;   // (EXP: the track's staged row, or the two of its pair)
        const uint32_t r1 = EXP ? s_off[sb * (kSt + 1u) + rec + 1u] : rec + 1u;
        for (uint32_t rr = r0; rr < r1; rr++) {
          const uint32_t v = s_pk[(sb * kRecs + rr) * kPS + ch];
          pk = pk > v ? pk : v;
        }
      } else if (FULL) {
        // (EXP: the track's staged row, or the two of its pair — one peak over both stream calls)
        const uint32_t r0 = EXP ? s_off[sb * (kSt + 1u) + rec] : rec;
        const uint32_t r1 = EXP ? s_off[sb * (kSt + 1u) + rec + 1u] : rec + 1u;
        for (uint32_t rr = r0; rr < r1; rr++) {
          const uint32_t* slots = &s_pk[(sb * kRecs + rr) * kPS];
#pragma unroll
          for (uint32_t w = 0; w < kPS; w++)
            if (s_wc[w] == sb * C + ch) pk = pk > slots[w] ? pk : slots[w];
        }
      } else {
        pk = s_pk[rec * 2u + ch];
      }
      if (a.tiles == 1u)
        *dst = pk;
      else
        atomicMax(dst, pk);
      // VUMeter::level keeps the maximum until the UI reads it (vu_meter.h:26-29)
      if (a.levels && pk != 0u) atomicMax(a.levels + (size_t)track * C + ch, pk);
    }
    chunk0 += cn;
  }

  if (a.dbg_clock && tid == 0u) {   // (diagnostic, WBX_DBG_CLOCK=1: when did this workgroup start and end — 100 MHz wall clock)
    const uint32_t wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    a.dbg_clock[4u * wg] = dbg_t0;
    a.dbg_clock[4u * wg + 1u] = wall_clock64();
    a.dbg_clock[4u * wg + 2u] = (unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID: cu / sh / se of this wave
    a.dbg_clock[4u * wg + 3u] = (unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
  }
  if (a.fused_master) {   // (wave-uniform) this workgroup's sum IS the block's: what sum_kernel would do with one group
    if (a.fused_status_dst && blockIdx.x == 0u && tid < 4u) {
      const uint32_t queued = a.fused_status_src[2];
      if (a.partial_through)
        __hip_atomic_store(a.fused_status_dst + tid, a.fused_status_src[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else
        a.fused_status_dst[tid] = a.fused_status_src[tid];
      if (a.fused_zero_status && queued == 0u) a.fused_status_src[tid] = 0u;
    }
    if (owns && bvalid) {
#pragma unroll
      for (int ch = 0; ch < CL; ch++) {
        f4 m = acc.c[ch];
        m.x = __fadd_rn(0.0f, m.x);   // the master starts from the cleared buffer (engine.cpp:1598): -0.0 becomes +0.0
        m.y = __fadd_rn(0.0f, m.y);
        m.z = __fadd_rn(0.0f, m.z);
        m.w = __fadd_rn(0.0f, m.w);
        if (a.fused_clamp) {          // engine.cpp:1627-1636: compare, don't min/max (NaN passes through unchanged)
          m.x = m.x > 1.0f ? 1.0f : (m.x < -1.0f ? -1.0f : m.x);
          m.y = m.y > 1.0f ? 1.0f : (m.y < -1.0f ? -1.0f : m.y);
          m.z = m.z > 1.0f ? 1.0f : (m.z < -1.0f ? -1.0f : m.z);
          m.w = m.w > 1.0f ? 1.0f : (m.w < -1.0f ? -1.0f : m.w);
        }
        float* mo = a.fused_master + ((size_t)b * C + (CL == 2 ? (uint32_t)ch : c)) * F + j0;
        if (a.partial_through)   // (wave-uniform) the one-launch callback: the host is told by a flag from inside this launch
          store_f4_system(mo, m);
        else
          *reinterpret_cast<f4*>(mo) = m;
      }
    }
  } else if (owns && bvalid) {
#pragma unroll
    for (int ch = 0; ch < CL; ch++) {
      float* out = a.partial + (((size_t)b * a.n_groups + g) * C + (CL == 2 ? (uint32_t)ch : c)) * F + j0;
      if (a.partial_through) {   // (wave-uniform) the one-launch callback: read by a workgroup behind another L2 in this same launch
        uint32_t* o = reinterpret_cast<uint32_t*>(out);
        __hip_atomic_store(o + 0, __float_as_uint(acc.c[ch].x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, __float_as_uint(acc.c[ch].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 2, __float_as_uint(acc.c[ch].z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 3, __float_as_uint(acc.c[ch].w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *reinterpret_cast<f4*>(out) = acc.c[ch];   // (chained: the running sum for the next piece — the L1 writes through to the XCD's L2)
      }
    }
  }
  if (chain_out) {
    __builtin_amdgcn_s_waitcnt(0);   // every store of this wave has been acknowledged ...
    __syncthreads();                 // ... of every wave of the workgroup
    if (tid == 0u) __hip_atomic_store(chain_word, chain_tag | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int U, bool FULL, int W, int FAM, int SB, int CW = 1, int CL = 1, int T = 256 / CL>
__global__ __launch_bounds__(T, W) void mix_kernel(MixArgs a) {
  mix_body<U, FULL, FAM, SB, CW, CL, T>(a);
}
// the packed instances (SB blocks per 256-lane workgroup) that take masked rows: a kernel of its own, so that the names of
// the others stay what every profile of earlier rounds calls them
template <int U, int W, int FAM, int SB, int CW, int X>
__global__ __launch_bounds__(256, W) void mix_kernel_x(MixArgs a) {
  mix_body<U, true, FAM, SB, CW, 1, 256, X>(a);
}

// one instance, launched; t0 / t1 (optional): events that take the kernel's own start and end times (hipExtLaunchKernelGGL:
// the time stamps of its dispatch packet — no event packets of their own in the stream).  -> the instance's name as
// rocprofv3 prints it
#define WBX_MIX(U, FULL, W, FAM, SB, CW, CL, T, GRID, BLOCK)                                                   \
  {                                                                                                            \
    name = "wbx::mix_kernel<" #U ", " #FULL ", " #W ", " #FAM ", " #SB ", " #CW ", " #CL ", " #T ">";          \
    hipExtLaunchKernelGGL((mix_kernel<U, FULL, W, FAM, SB, CW, CL, T>), GRID, BLOCK, 0, s, t0, t1, 0, a);      \
  }

// a packed instance that takes masked rows (X = 1: half-length chunks)
#define WBX_MIX_X(U, W, FAM, SB, CW, X, GRID)                                                                  \
  {                                                                                                            \
    name = "wbx::mix_kernel_x<" #U ", " #W ", " #FAM ", " #SB ", " #CW ", " #X ">";                            \
    hipExtLaunchKernelGGL((mix_kernel_x<U, W, FAM, SB, CW, X>), GRID, dim3(256), 0, s, t0, t1, 0, a);          \
  }

// the instances of one family (wbx_mix_fam<N>.hip); `variant`: 10 * U + W, or >= 1000 for both channels of a frame per lane
const char* launch_mix_fam0(const MixArgs& a, uint32_t n_blocks, int variant, hipStream_t s, hipEvent_t t0, hipEvent_t t1);
const char* launch_mix_fam1(const MixArgs& a, uint32_t n_blocks, hipStream_t s, hipEvent_t t0, hipEvent_t t1);
const char* launch_mix_fam2(const MixArgs& a, uint32_t n_blocks, int variant, hipStream_t s, hipEvent_t t0, hipEvent_t t1);
const char* launch_mix_fam3(const MixArgs& a, uint32_t n_blocks, int variant, hipStream_t s, hipEvent_t t0, hipEvent_t t1);

}  // namespace wbx
