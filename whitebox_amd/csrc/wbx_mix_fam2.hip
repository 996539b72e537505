// wbx_mix_fam2.hip — mix_kernel instances of family 2: sessions whose clips are all 16-bit PCM at speeds up to 0.999 or
// exactly 1 (CD-rate files in a 48 kHz project): chunk modes U, I16, MU, WI, WIN, WINU; a chunk that holds a pre-rendered
// fp32 row next to 16-bit window rows goes one row at a time.
#include "wbx_mix.h"
#include "wbx_callback.h"

namespace wbx {

const char* launch_mix_fam2(const MixArgs& a, uint32_t n_blocks, int variant, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  const char* name = "";
  const dim3 grid(n_blocks, a.n_groups, a.tiles), block(256);
  const uint32_t S4 = a.lane_span;   // (the instance's lane space: F/4, or the next shape above it)
  if (variant >= 1000 && a.channels == 2u && S4 == 64u)          // 256-frame stereo blocks: one wave = one block
    WBX_MIX(2, true, 3, 2, 1, 1, 2, 64, grid, dim3(64))
  else if (variant == 1042 && a.channels == 2u && S4 == 128u && a.tiles == 1u)
    WBX_MIX(4, true, 2, 2, 1, 1, 2, 128, grid, dim3(128))
  else if (variant == 1013 && a.channels == 2u && S4 == 128u && a.tiles == 1u)
    WBX_MIX(1, true, 3, 2, 1, 1, 2, 128, grid, dim3(128))
  else if (variant >= 1000 && a.channels == 2u && S4 == 128u && a.tiles == 1u)
    WBX_MIX(2, true, 3, 2, 1, 1, 2, 128, grid, dim3(128))
  else if (variant >= 1000 && a.channels == 2u && S4 == 256u)
    WBX_MIX(2, true, 3, 2, 1, 1, 2, 256, dim3(n_blocks, a.n_groups, 1), dim3(256))
  else
    WBX_MIX(2, true, 4, 2, 1, 1, 1, 256, grid, block)
  return name;
}

const char* launch_callback_fam2(const MixArgs& a, const PlanArgs& p, const SumArgs& s, const CallbackArgs& cb, hipStream_t st) {
  const char* name = "";
  WBX_CALLBACK(2, 2)
  return name;
}

}  // namespace wbx
