// ref_deint_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// extern "C" door to the REFERENCE'S OWN deinterleave_samples<T> (dsp/sample.cpp:29-43), the transposition Sample::load_file
// (sample.cpp:112-197) runs on every 1024-frame chunk a decoder hands it.  sample.cpp as a whole includes <sndfile.h>,
// <vorbis/vorbisfile.h> and dr_mp3 — absent from the image, and stand-ins are not written.  The function itself uses none of
// their code: the only thing of libsndfile's in it is the NAME of its count type, `sf_count_t`.  oracle/Makefile cuts the
// function's text (its `template<typename T>` line to its closing brace) out of the file where it lies into
// _ref/deinterleave_impl.inc (a build output, git-ignored, never committed), and this driver compiles that text UNMODIFIED as a
// member of a class template whose parameter carries that name — so the text is compiled for ANY integer count type rather than
// against a made-up sndfile.h.  It is instantiated at int64_t (what libsndfile 1.2.2, the reference's pinned version —
// Dependencies.cmake:131-139 —, publishes sf_count_t as) and, to show that nothing rests on that choice, at int32_t.
//
// What the driver adds of its own is load_file's call site (sample.cpp:127-142,160-185): per-channel buffers of
// frames + sample_padding (16) zeroed elements, and the `while (read 1024 frames) written = deinterleave_samples(...)` loop.
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "core/vector.h"

namespace wb {
template<typename sf_count_t>
struct SampleCppCut {
#include "_ref/deinterleave_impl.inc"   // template<typename T> static sf_count_t deinterleave_samples(Vector<std::byte*>& dst, ...)
};
}  // namespace wb

namespace {
template<typename CountT, typename T>
int64_t run(void* const* dst, const void* src, int64_t frames, int channels, int64_t chunk) {
  wb::Vector<std::byte*> data;
  data.reserve((uint32_t)channels);
  for (int c = 0; c < channels; c++) data.push_back((std::byte*)dst[c]);
  const T* p = (const T*)src;
  CountT written = 0;
  for (int64_t at = 0; at < frames; at += chunk) {
    CountT n = (CountT)((frames - at) < chunk ? (frames - at) : chunk);   // what sf_readf_* returns for the chunk
    written = wb::SampleCppCut<CountT>::template deinterleave_samples<T>(data, p + (size_t)at * channels, n, (CountT)frames,
                                                                         written, channels);
  }
  return (int64_t)written;
}
}  // namespace

extern "C" {
// dst[c]: caller-allocated (frames + 16) * elem bytes, zeroed (load_file's malloc + memset, sample.cpp:127-142).
// elem = 2 (I16) | 4 (I32 / F32: the same 4-byte moves); count_bits = 64 | 32: the integer type standing behind `sf_count_t`.
// Returns num_frames_written, -1 for an argument the reference has no instance for.
int64_t ref_deinterleave(void* const* dst, const void* src, int64_t frames, int channels, int elem, int64_t chunk,
                         int count_bits) {
  if (chunk <= 0 || channels <= 0 || frames < 0) return -1;
  if (count_bits == 64) {
    if (elem == 2) return run<int64_t, int16_t>(dst, src, frames, channels, chunk);
    if (elem == 4) return run<int64_t, int32_t>(dst, src, frames, channels, chunk);
  } else if (count_bits == 32) {
    if (elem == 2) return run<int32_t, int16_t>(dst, src, frames, channels, chunk);
    if (elem == 4) return run<int32_t, int32_t>(dst, src, frames, channels, chunk);
  }
  return -1;
}
// the float instance moves floats (sample.cpp:176-181); kept apart so that NaN payloads are seen to survive it
int64_t ref_deinterleave_f32(void* const* dst, const float* src, int64_t frames, int channels, int64_t chunk) {
  if (chunk <= 0 || channels <= 0 || frames < 0) return -1;
  return run<int64_t, float>(dst, src, frames, channels, chunk);
}
}
