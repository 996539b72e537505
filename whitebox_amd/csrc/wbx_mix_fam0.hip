// wbx_mix_fam0.hip — mix_kernel instances of family 0: fp32 rows (unity / 5-sample window) and integer PCM at unity speed
// (chunk modes U, W, WN, WNU, I16, I32, MU, MIXED).  What the BASELINE configurations 2-5 take.
#include <cstdlib>

#include "wbx_mix.h"
#include "wbx_callback.h"

namespace wbx {

const char* launch_mix_fam0(const MixArgs& a, uint32_t n_blocks, int variant, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  const char* name = "";
  const dim3 grid(n_blocks, a.n_groups, a.tiles), block(256);
  const uint32_t S4 = a.lane_span;   // (the instance's lane space: F/4, or the next shape above it)
  const uint32_t lanes = a.channels * S4;   // lanes one block needs
  const bool full = (lanes % 256u == 0u) && (S4 % 64u == 0u);
  if (!full && a.masked_rows) {
    // short blocks of a session cut into clips, renders of 8 blocks and more: the packed instances that take masked rows
    // (2 or 4 blocks per workgroup like the ones below, the staging of the one-block instances per sub-block)
    const bool st128 = S4 == 32u && a.channels == 2u, two = S4 % 64u == 0u && lanes == 128u, four = S4 == 64u && lanes == 64u;
    const int x = packed_masked_variant(n_blocks, st128, a.packed_x);
    if (x && (st128 || two || four)) {
      const uint32_t sb = st128 ? 4u : 256u / lanes;
      const dim3 gx((n_blocks + sb - 1u) / sb, a.n_groups, 1);
      if (st128)
        WBX_MIX_X(2, 4, 0, 4, 2, 1, gx)
      else if (two)
        WBX_MIX_X(2, 4, 0, 2, 1, 1, gx)
      else
        WBX_MIX_X(2, 4, 0, 4, 1, 1, gx)
      return name;
    }
  }
  // variant >= 1000, stereo 256-frame blocks: one wave = one block with both channels of a frame in a lane
  if (variant >= 1000 && a.channels == 2u && S4 == 64u) {
    WBX_MIX(2, true, 3, 0, 1, 1, 2, 64, grid, dim3(64))
    return name;
  }
  if (!full && a.masked_rows) {
    // short blocks of a session cut into clips: one block per workgroup (a wave, or two), the instances that take the
    // sequencer's masked rows — clip boundaries stay in the hot loop instead of going through the pre-render pass
    if (S4 == 32u && a.channels == 2u) {          // 128-frame stereo: one wave, a channel per half-wave
      WBX_MIX(2, true, 3, 0, 1, 2, 1, 64, grid, dim3(64))
      return name;
    }
    if (S4 % 64u == 0u && lanes == 128u) {        // 256-frame stereo (a wave per channel), 512-frame mono
      WBX_MIX(2, true, 3, 0, 1, 1, 1, 128, grid, dim3(128))
      return name;
    }
    if (S4 == 64u && lanes == 64u) {              // 256-frame mono
      WBX_MIX(2, true, 3, 0, 1, 1, 1, 64, grid, dim3(64))
      return name;
    }
  }
  if (!full) {
    // blocks shorter than a workgroup whose waves are still channel-uniform (256 frames; 512 mono): 2 or 4
    // consecutive blocks per workgroup, same code as the full instances
    if (S4 == 32u && a.channels == 2u) {   // 128-frame stereo blocks: one block per wave, a channel per half-wave
      const dim3 g4((n_blocks + 3u) / 4u, a.n_groups, 1);
      WBX_MIX(2, true, 4, 0, 4, 2, 1, 256, g4, block)
      return name;
    }
    if (S4 % 64u == 0u && (lanes == 128u || lanes == 64u)) {
      const uint32_t sb = 256u / lanes;
      const dim3 g2((n_blocks + sb - 1u) / sb, a.n_groups, 1);
      if (sb == 2u)
        WBX_MIX(2, true, 4, 0, 2, 1, 1, 256, g2, block)
      else
        WBX_MIX(2, true, 4, 0, 4, 1, 1, 256, g2, block)
      return name;
    }
    return launch_mix_fam1(a, n_blocks, s, t0, t1);   // any other block shape: the general instance
  }
  // variant >= 1000: stereo 512-frame blocks with both channels of a frame in one lane (workgroups of 128 lanes = one
  // block; 26 KiB of LDS each: three waves per SIMD)
  if (variant >= 1000 && a.channels == 2u && S4 == 128u && a.tiles == 1u) {
    if (variant == 1013)
      WBX_MIX(1, true, 3, 0, 1, 1, 2, 128, grid, dim3(128))
    else if (variant == 1042)   // twice the rows in flight per wave at two waves per SIMD (whole-list walks of 1024 blocks: four workgroups per CU)
      WBX_MIX(4, true, 2, 0, 1, 1, 2, 128, grid, dim3(128))
    else
      WBX_MIX(2, true, 3, 0, 1, 1, 2, 128, grid, dim3(128))
    return name;
  }
  // ... and 1024-frame ones: workgroups of 256 lanes = one block
  if (variant >= 1000 && a.channels == 2u && S4 == 256u) {
    const dim3 g1(n_blocks, a.n_groups, 1);
    WBX_MIX(2, true, 3, 0, 1, 1, 2, 256, g1, dim3(256))
    return name;
  }
  // variant = 10*U + W: U tracks per pipeline stage, W = waves per SIMD the register budget is capped for
  // (tuning knob WBX_MIX_VARIANT; every variant computes identical results)
  switch (variant) {
#define WBX_V(U, W) case 10 * U + W: WBX_MIX(U, true, W, 0, 1, 1, 1, 256, grid, block) break;
    WBX_V(2, 4) WBX_V(4, 3) WBX_V(8, 2)   // (1/6, 2/5, 2/6, 4/4, 4/5 spill and were 10-60 % slower)
#undef WBX_V
    default: WBX_MIX(2, true, 4, 0, 1, 1, 1, 256, grid, block) break;
  }
  return name;
}

// the one-launch callback (wbx_callback.h), 512-frame stereo / 1024-frame mono blocks: sessions with resampled or integer-PCM
// clips stage two rows per pipeline batch, fp32 sessions at the session rate four
const char* launch_callback_fam0(const MixArgs& a, const PlanArgs& p, const SumArgs& s, const CallbackArgs& cb, bool window, hipStream_t st) {
  const char* name = "";
  // (a callback workgroup is a latency chain — one pipeline batch per memory round trip — and has a CU to itself: many rows
  //  per batch.  WBX_CB_U=2|4|8: A/B aid)
  static const int u = [] { const char* v = std::getenv("WBX_CB_U"); return v ? std::atoi(v) : 0; }();
  (void)window;
  // (measured, 16 tracks per workgroup: the rows that pad the last pipeline batch are rendered like real ones, and a
  //  callback workgroup is bound by instruction issue — U = 2: 11.3 us, 4: 11.6, 8: 13.4)
  if (u == 8)
    WBX_CALLBACK(8, 0)
  else if (u == 4)
    WBX_CALLBACK(4, 0)
  else
    WBX_CALLBACK(2, 0)
  return name;
}

}  // namespace wbx
