// wbx_mix_fam3.hip — mix_kernel instances of family 3: family 1's chunk modes without the per-frame taps (MODE_G).  Sessions
// with resampled integer PCM (24-bit stems at another rate, 16-bit loops next to them) but no clip played faster than
// recorded take it: without that mode the instance with both channels of a frame per lane fits its register budget.
#include "wbx_mix.h"

namespace wbx {

const char* launch_mix_fam3(const MixArgs& a, uint32_t n_blocks, int variant, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  const char* name = "";
  const dim3 grid(n_blocks, a.n_groups, a.tiles);
  const uint32_t S4 = a.block_frames >> 2;
  if (variant >= 1000 && a.channels == 2u && S4 == 128u && a.tiles == 1u) {
    if (variant == 1013)        // (tuning variants, WBX_MIX_VARIANT: one row per batch; twice the rows at two waves per SIMD)
      WBX_MIX(1, true, 3, 3, 1, 1, 2, 128, grid, dim3(128))
    else if (variant == 1042)
      WBX_MIX(4, true, 2, 3, 1, 1, 2, 128, grid, dim3(128))
    else if (variant == 1022)
      WBX_MIX(2, true, 2, 3, 1, 1, 2, 128, grid, dim3(128))
    else
      WBX_MIX(2, true, 3, 3, 1, 1, 2, 128, grid, dim3(128))
    return name;
  }
  return launch_mix_fam1(a, n_blocks, s, t0, t1);   // every other block shape: the everything family holds all of this one's modes
}

}  // namespace wbx
