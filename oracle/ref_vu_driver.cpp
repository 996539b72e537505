// ref_vu_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// extern "C" door to the REFERENCE'S OWN VUMeter (engine/vu_meter.h:15-45: push_samples' abs-max loop and CAS-max into
// `level`, update()'s exchange).  The header also includes core/debug.h (third-party spdlog, absent from the image) without
// the struct using anything of it; oracle/Makefile cuts the struct's text out of the header where it lies into
// _ref/vu_meter_struct.inc (a build output, git-ignored) and this driver compiles it unmodified with the headers the struct
// does use.  No stand-in header, no copied source in the repo.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "core/audio_buffer.h"
#include "core/core_math.h"

namespace wb {
#include "_ref/vu_meter_struct.inc"
}  // namespace wb

extern "C" {
// one meter, a sequence of `n_blocks` blocks of `n` samples pushed into it (Track::process calls push_samples once per block
// and channel, track.cpp:732); levels[b] = VUMeter::level after block b.  reset_every > 0: `level.exchange(0.0f)` (what
// update() does at UI rate, vu_meter.h:33) before every reset_every-th block.
void ref_vu_push_blocks(const float* samples, uint32_t n, uint32_t n_blocks, uint32_t reset_every, float* levels) {
  wb::VUMeter m{};
  m.level.store(0.0f);
  wb::AudioBuffer<float> buf(n, 1);
  for (uint32_t b = 0; b < n_blocks; b++) {
    if (reset_every && b && b % reset_every == 0) m.level.exchange(0.0f, std::memory_order_release);
    std::memcpy(buf.get_write_pointer(0), samples + (size_t)b * n, (size_t)n * sizeof(float));
    m.push_samples(buf, 0);
    levels[b] = m.level.load();
  }
}
}
