// Host-only checks of wbx::AudioBuffer (include/wbx_adapter.hpp) for the behaviours the reference's own
// test/test_audio_buffer.cpp exercises on wb::AudioBuffer: construct, resize with/without clearing,
// resize_channel grow/shrink — plus clear/mix/interleave round trip.  No GPU, no libwbx calls.
#include <cstdio>
#include <cstring>
#include <vector>

#include "wbx_adapter.hpp"

#define REQUIRE(x)                                              \
  do {                                                          \
    if (!(x)) {                                                 \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); \
      return 1;                                                 \
    }                                                           \
  } while (0)

int main() {
  {   // construct
    wbx::AudioBuffer<float> b(128, 2);
    REQUIRE(b.n_samples == 128 && b.n_channels == 2);
    for (uint32_t c = 0; c < b.n_channels; c++) {
      REQUIRE(b.get_read_pointer(c) != nullptr);
      for (uint32_t i = 0; i < 128; i++) REQUIRE(b.get_read_pointer(c)[i] == 0.0f);
    }
  }
  {   // resize with clearing
    wbx::AudioBuffer<float> b(128, 2);
    b.get_write_pointer(0)[5] = 1.0f;
    b.resize(256, true);
    REQUIRE(b.n_samples == 256);
    for (uint32_t i = 0; i < 256; i++) REQUIRE(b.get_read_pointer(0)[i] == 0.0f);
  }
  {   // expand without clearing keeps the data, zero-fills the growth; shrinking keeps the prefix
    std::vector<float> ref(256);
    for (size_t i = 0; i < ref.size(); i++) ref[i] = (float)((i * 37) % 101) / 50.0f - 1.0f;
    wbx::AudioBuffer<float> b(256, 2);
    for (uint32_t c = 0; c < 2; c++) std::memcpy(b.get_write_pointer(c), ref.data(), 256 * sizeof(float));
    b.resize(512);
    REQUIRE(b.n_samples == 512);
    for (uint32_t c = 0; c < 2; c++) {
      REQUIRE(std::memcmp(b.get_read_pointer(c), ref.data(), 256 * sizeof(float)) == 0);
      for (uint32_t i = 256; i < 512; i++) REQUIRE(b.get_read_pointer(c)[i] == 0.0f);
    }
    b.resize(100);
    REQUIRE(b.n_samples == 100 && std::memcmp(b.get_read_pointer(1), ref.data(), 100 * sizeof(float)) == 0);
    b.resize(100);   // same size: no-op
    REQUIRE(b.n_samples == 100);
  }
  {   // resize_channel
    wbx::AudioBuffer<float> b(256, 2);
    b.resize_channel(4);
    REQUIRE(b.n_channels == 4);
    for (uint32_t c = 0; c < 4; c++) REQUIRE(b.get_read_pointer(c) != nullptr && b.get_read_pointer(c)[255] == 0.0f);
    wbx::AudioBuffer<float> d(256, 4);
    d.resize_channel(2);
    REQUIRE(d.n_channels == 2);
    for (uint32_t c = 0; c < 2; c++) REQUIRE(d.get_read_pointer(c) != nullptr);
  }
  {   // clear, mix (audio_buffer.h:67-82), interleave round trip
    wbx::AudioBuffer<float> a(64, 2), b(64, 2), r(64, 2);
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t i = 0; i < 64; i++) {
        a.set_sample(c, i, 0.25f * (float)i);
        b.set_sample(c, i, (float)c + 0.5f);
      }
    a.mix(b);
    REQUIRE(a.get_read_pointer(1)[10] == 0.25f * 10 + 1.5f);
    a.mix_sample(0, 3, 1.0f);
    REQUIRE(a.get_read_pointer(0)[3] == 0.75f + 0.5f + 1.0f);
    std::vector<float> inter(128);
    a.interleave_samples_to(inter.data(), 0, 64);
    REQUIRE(inter[2 * 10 + 1] == a.get_read_pointer(1)[10]);
    r.deinterleave_samples_from(inter.data(), 0, 64);
    for (uint32_t c = 0; c < 2; c++) REQUIRE(std::memcmp(r.get_read_pointer(c), a.get_read_pointer(c), 64 * sizeof(float)) == 0);
    a.clear();
    REQUIRE(a.get_read_pointer(0)[3] == 0.0f);
  }
  {   // the reference's layout (audio_buffer.h:19-23): 16 channel pointers inside the object, a heap array beyond
    wbx::AudioBuffer<float> e;
    REQUIRE(e.channel_buffers == e.internal_channel_buffers && e.channel_capacity == 16 && e.n_channels == 0);
    wbx::AudioBuffer<float> b(32, 2);
    REQUIRE(b.channel_buffers == b.internal_channel_buffers);
    REQUIRE(((uintptr_t)b.channel_buffers[0] % wbx::AudioBuffer<float>::alignment) == 0);
    b.get_write_pointer(1)[7] = 3.0f;
    b.resize_channel(20);   // past the internal capacity: the pointers move, the channels stay
    REQUIRE(b.n_channels == 20 && b.channel_buffers != b.internal_channel_buffers && b.channel_capacity >= 20);
    REQUIRE(b.get_read_pointer(1)[7] == 3.0f && b.get_read_pointer(19)[31] == 0.0f);
    b.resize_channel(3);
    REQUIRE(b.n_channels == 3 && b.get_read_pointer(1)[7] == 3.0f);
    std::vector<float> inter(3 * 32);
    b.interleave_samples_to(inter.data(), 0, 32, wbx::AudioFormat::F32);
    REQUIRE(inter[7 * 3 + 1] == 3.0f);
  }
  std::printf("audio_buffer ok\n");
  return 0;
}
