// ref_mip_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// extern "C" door to the REFERENCE'S OWN summarize_for_mipmaps_impl<T> (gfx/waveform_visual.cpp:9-173), for validating
// wbo_mip_summarize (wb_oracle.c) and generating tests/golden/mip.npz.  The function is a file-local template in a
// translation unit whose other functions need spdlog and the renderer; oracle/Makefile cuts the function's text out of the
// file where it lies into _ref/mip_impl.inc (a build output, git-ignored) and this driver compiles it unmodified, with the
// reference's own headers and nothing else — no stand-in header, no copied source in the repo.
// The per-level arguments follow WaveformVisual::create (waveform_visual.cpp:194-221), which itself cannot be built
// (g_renderer): level l has current_mip = 1 + 2l, chunk_count = 2^mip, block_count = 2^(mip-1),
// mip_data_count = count / block_count rounded up to even.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>

#include "core/audio_format.h"
#include "core/common.h"
#include "core/core_math.h"

#include "_ref/mip_impl.inc"   // namespace wb { template<typename T> static void summarize_for_mipmaps_impl(...) { ... }
}  // namespace wb — the cut ends in front of the namespace's next function

extern "C" void ref_mip_summarize(int format, size_t count, const void* data, uint32_t level, int out_bits, void* out) {
  const uint32_t current_mip = 1u + 2u * level;
  const size_t chunk_count = size_t(1) << current_mip;
  const size_t block_count = size_t(1) << (current_mip - 1);
  size_t mip_data_count = count / block_count;
  mip_data_count += mip_data_count % 2;
  if (out_bits == 8)
    wb::summarize_for_mipmaps_impl((wb::AudioFormat)format, count, (const std::byte*)data, chunk_count, block_count,
                                   mip_data_count, (int8_t*)out);
  else
    wb::summarize_for_mipmaps_impl((wb::AudioFormat)format, count, (const std::byte*)data, chunk_count, block_count,
                                   mip_data_count, (int16_t*)out);
}
