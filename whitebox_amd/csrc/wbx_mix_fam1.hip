// wbx_mix_fam1.hip — mix_kernel instances of family 1, "everything": also per-frame taps (MODE_G: fp32 played faster than
// recorded, resampled integer PCM above 0.999), 16 / 24 / 32-bit window rows, windows of several storage formats in one
// chunk (MW, MWN).  Sessions with such clips take it; so does every block shape the lean families have no instance for.
#include "wbx_mix.h"
#include "wbx_callback.h"

namespace wbx {

const char* launch_mix_fam1(const MixArgs& a, uint32_t n_blocks, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  const char* name = "";
  const dim3 grid(n_blocks, a.n_groups, a.tiles), block(256);
  const uint32_t S4 = a.lane_span;   // (the instance's lane space: F/4, or the next shape above it)
  const uint32_t lanes = a.channels * S4;   // lanes one block needs
  const bool full = (lanes % 256u == 0u) && (S4 % 64u == 0u);
  if (!full && a.masked_rows) {
    // short blocks of a session cut into clips, renders of 8 blocks and more: the packed instances that take masked rows
    const bool st128 = S4 == 32u && a.channels == 2u, two = S4 % 64u == 0u && lanes == 128u, four = S4 == 64u && lanes == 64u;
    const int x = packed_masked_variant(n_blocks, st128, a.packed_x);
    if (x && (st128 || two || four)) {
      const uint32_t sb = st128 ? 4u : 256u / lanes;
      const dim3 gx((n_blocks + sb - 1u) / sb, a.n_groups, 1);
      if (st128)
        WBX_MIX_X(2, 4, 1, 4, 2, 1, gx)
      else if (two)
        WBX_MIX_X(2, 4, 1, 2, 1, 1, gx)
      else
        WBX_MIX_X(2, 4, 1, 4, 1, 1, gx)
      return name;
    }
  }
  if (!full && a.masked_rows) {
    // short blocks of a session cut into clips: one block per workgroup (a wave, or two), the instances that take the
    // sequencer's masked rows — clip boundaries stay in the hot loop instead of going through the pre-render pass
    if (S4 == 32u && a.channels == 2u) {          // 128-frame stereo: one wave, a channel per half-wave
      WBX_MIX(2, true, 3, 1, 1, 2, 1, 64, grid, dim3(64))
      return name;
    }
    if (S4 % 64u == 0u && lanes == 128u) {        // 256-frame stereo (a wave per channel), 512-frame mono
      WBX_MIX(2, true, 3, 1, 1, 1, 1, 128, grid, dim3(128))
      return name;
    }
    if (S4 == 64u && lanes == 64u) {              // 256-frame mono
      WBX_MIX(2, true, 3, 1, 1, 1, 1, 64, grid, dim3(64))
      return name;
    }
  }
  if (!full) {
    if (S4 == 32u && a.channels == 2u) {   // 128-frame stereo blocks: one block per wave, a channel per half-wave
      const dim3 g4((n_blocks + 3u) / 4u, a.n_groups, 1);
      WBX_MIX(2, true, 4, 1, 4, 2, 1, 256, g4, block)
      return name;
    }
    if (S4 % 64u == 0u && (lanes == 128u || lanes == 64u)) {   // 256 frames stereo, 256 / 512 frames mono: 2 or 4 blocks per workgroup
      const uint32_t sb = 256u / lanes;
      const dim3 g2((n_blocks + sb - 1u) / sb, a.n_groups, 1);
      if (sb == 2u)
        WBX_MIX(2, true, 4, 1, 2, 1, 1, 256, g2, block)
      else
        WBX_MIX(2, true, 4, 1, 4, 1, 1, 256, g2, block)
      return name;
    }
    WBX_MIX(2, false, 1, 1, 1, 1, 1, 256, grid, block)   // any block shape: lane predicates, records read from LDS per lane
    return name;
  }
  // (W = 4 although this instance spills a few registers there: at W = 3 it is 5-10 % slower)
  WBX_MIX(2, true, 4, 1, 1, 1, 1, 256, grid, block)
  return name;
}

const char* launch_callback_fam1(const MixArgs& a, const PlanArgs& p, const SumArgs& s, const CallbackArgs& cb, hipStream_t st) {
  const char* name = "";
  WBX_CALLBACK(2, 1)
  return name;
}

}  // namespace wbx
