// wbx_mix_fam3.hip — mix_kernel instances of family 3: family 1's chunk modes without the per-frame taps (MODE_G).  Sessions
// with resampled integer PCM (24-bit stems at another rate, 16-bit loops next to them) but no clip played faster than
// recorded take it: without that mode the instance with both channels of a frame per lane fits its register budget.
#include "wbx_mix.h"

namespace wbx {

const char* launch_mix_fam3(const MixArgs& a, uint32_t n_blocks, int variant, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  const char* name = "";
  const dim3 grid(n_blocks, a.n_groups, a.tiles);
  const uint32_t S4 = a.lane_span;   // (the instance's lane space: F/4, or the next shape above it)
  if (variant >= 1000 && a.channels == 2u && S4 == 128u && a.tiles == 1u) {
    // one row per pipeline batch: with two, this family's widest modes spill 84 B per lane at three waves per SIMD
    // (measured, one box: i24r 0.650 -> 0.690 of the roofline, mixr 0.501 -> 0.530, cut into clips +2-3 %; two waves per SIMD
    // without spills — <2,true,2,...> — 0.62 / 0.51, <4,true,2,...> 0.61 / 0.48).  WBX_MIX_VARIANT=1022: two rows per batch.
    if (variant == 1022)
      WBX_MIX(2, true, 3, 3, 1, 1, 2, 128, grid, dim3(128))
    else
      WBX_MIX(1, true, 3, 3, 1, 1, 2, 128, grid, dim3(128))
    return name;
  }
  return launch_mix_fam1(a, n_blocks, s, t0, t1);   // every other block shape: the everything family holds all of this one's modes
}

}  // namespace wbx
