// C++ host using the reference-shaped adapter (include/wbx_adapter.hpp) over libwbx.so, checked block by
// block against the CPU oracle (oracle/wb_oracle.h).  Session = the survey's seek-math KAT
// plus a resampled track: written the way a reference host would write it (engine.h / track.h names).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "wb_oracle.h"
#include "wbx_adapter.hpp"

int main() {
  const uint32_t F = 512, C = 2, SR = 48000, NB = 6;
  const size_t cnt = 6000;
  std::vector<float> ramp(cnt + 16, 0.0f), ramp2(cnt + 16, 0.0f);
  for (size_t i = 0; i < cnt; i++) {
    ramp[i] = 0.001f * (float)(i + 1);
    ramp2[i] = 0.0005f * (float)((i * 7) % 1000);
  }
  const void* planar[2] = {ramp.data(), ramp2.data()};

  // ---- product, through the adapter -------------------------------------------------------------
  wbx::Engine g_engine;
  g_engine.max_tracks = 8;
  g_engine.set_audio_channel_config(0, C, 128, 44100);   // as a host does at start-up, before the device is known ...
  g_engine.set_audio_channel_config(0, C, F, SR);          // ... and again when the backend settles: same engine
  g_engine.set_bpm(120.0);
  const uint32_t s48 = g_engine.add_sample(WBX_FMT_F32, 2, 48000, cnt, planar);
  const uint32_t s44 = g_engine.add_sample(WBX_FMT_F32, 2, 44100, cnt, planar);
  wbx::Track* t0 = g_engine.add_track("kat");
  wbx::Track* t1 = g_engine.add_track("resampled");
  t0->set_volume(0.0f);
  t1->set_volume(-6.0f);
  t1->set_pan(0.3f);
  g_engine.add_audio_clip(t0, "a", 100.0 / 24000, 700.0 / 24000, 10.0, wbx::AudioClip{s48, 1.0, 0.5f});
  g_engine.add_audio_clip(t0, "b", 812.0 / 24000, 2000.0 / 24000, 0.0, wbx::AudioClip{s48, 1.0, 1.0f});
  g_engine.add_audio_clip(t1, "c", 40.0 / 24000, 2500.0 / 24000, 3.0, wbx::AudioClip{s44, 1.0, 0.8f});
  g_engine.play();

  // ---- oracle ------------------------------------------------------------------------------------
  wbo_engine* o = wbo_engine_create(C, F, SR);
  wbo_engine_set_bpm(o, 120.0);
  const int o48 = wbo_engine_add_sample(o, WBO_FMT_F32, 2, 48000, cnt, planar);
  const int o44 = wbo_engine_add_sample(o, WBO_FMT_F32, 2, 44100, cnt, planar);
  wbo_engine_add_track(o);
  wbo_engine_add_track(o);
  wbo_track_set_volume(o, 0, 0.0f);
  wbo_track_set_volume(o, 1, -6.0f);
  wbo_track_set_pan(o, 1, 0.3f);
  wbo_engine_add_audio_clip(o, 0, 100.0 / 24000, 700.0 / 24000, 10.0, o48, 1.0, 0.5f);
  wbo_engine_add_audio_clip(o, 0, 812.0 / 24000, 2000.0 / 24000, 0.0, o48, 1.0, 1.0f);
  wbo_engine_add_audio_clip(o, 1, 40.0 / 24000, 2500.0 / 24000, 3.0, o44, 1.0, 0.8f);
  wbo_engine_play(o);

  wbx::AudioBuffer<float> in(F, C), out(F, C);
  std::vector<float> ol(F), orr(F);
  float* optr[2] = {ol.data(), orr.data()};
  for (uint32_t b = 0; b < NB; b++) {
    g_engine.process(in, out, (double)SR);
    wbo_engine_process(o, optr, nullptr);
    for (uint32_t c = 0; c < C; c++)
      if (std::memcmp(out.channel_buffers[c], optr[c], F * sizeof(float)) != 0) {
        std::printf("MISMATCH block %u channel %u\n", b, c);
        return 1;
      }
  }
  if (out.n_samples != F || out.n_channels != C) return 2;
  if (g_engine.process_status.load() != WBX_OK) return 3;
  wbo_engine_destroy(o);
  {   // the load figure Engine::process keeps (engine.cpp:1653; ui/control_bar.cpp:54 reads g_engine.perf_measurer.get_usage())
    const double u = g_engine.perf_measurer.get_usage();
    if (!(u > 0.0 && u <= 1.0)) return 9;
    if (g_engine.audio_buffer_duration_ms != wbo_buffer_duration_ms(F, SR)) return 9;   // engine.cpp:52
  }

  // meters: the maxima since the last read, as the host's own VUMeter::update takes them (vu_meter.h:32-33)
  g_engine.fetch_levels();
  if (!(g_engine.tracks[0]->level_meter[0].take_level() > 0.0f) || !(g_engine.tracks[1]->level_meter[1].take_level() > 0.0f)) return 4;
  if (g_engine.tracks[0]->level_meter[0].take_level() != 0.0f) return 4;   // (read and reset)
  {   // the UI-rate half the reference's controls call on Track::level_meter (vu_meter.h:32-44): attack at once, release by a one-pole
    wbx::VUMeter& vm = g_engine.tracks[1]->level_meter[0];
    (void)vm.take_level();              // (what fetch_levels left there)
    vm.push_level(0.5f);
    vm.update(60.0f, 0.25f);
    if (vm.get_value() != 0.5f) return 4;
    vm.update(60.0f, 0.25f);            // nothing new: falls by (1 - exp(-1/15)) of the way to 0
    const float expect = 0.5f + (0.0f - 0.5f) * (1.0f - std::exp(-1.0f / (60.0f * 0.25f)));
    // (within an ulp or two: the compiler folds this expression, the meter calls expf at run time)
    if (std::fabs(vm.get_value() - expect) > 1e-6f || !(vm.get_value() < 0.5f)) return 4;
  }
  // the effect slot exists and stays empty
  wbx_plugin fx{nullptr, nullptr};
  if (g_engine.add_plugin_to_track(t0, &fx) != nullptr || t0->plugin_instance != nullptr) return 5;

  // the audio backend is reconfigured (engine.cpp:43-57): half the buffer size — tracks, clips and samples stay.
  // Checked against a fresh oracle built at the new size from the same session.
  const uint32_t F2 = 256;
  g_engine.stop();
  g_engine.set_audio_channel_config(0, C, F2, SR);
  g_engine.play();
  wbo_engine* o2 = wbo_engine_create(C, F2, SR);
  wbo_engine_set_bpm(o2, 120.0);
  const int p48 = wbo_engine_add_sample(o2, WBO_FMT_F32, 2, 48000, cnt, planar);
  const int p44 = wbo_engine_add_sample(o2, WBO_FMT_F32, 2, 44100, cnt, planar);
  wbo_engine_add_track(o2);
  wbo_engine_add_track(o2);
  wbo_track_set_volume(o2, 0, 0.0f);
  wbo_track_set_volume(o2, 1, -6.0f);
  wbo_track_set_pan(o2, 1, 0.3f);
  wbo_engine_add_audio_clip(o2, 0, 100.0 / 24000, 700.0 / 24000, 10.0, p48, 1.0, 0.5f);
  wbo_engine_add_audio_clip(o2, 0, 812.0 / 24000, 2000.0 / 24000, 0.0, p48, 1.0, 1.0f);
  wbo_engine_add_audio_clip(o2, 1, 40.0 / 24000, 2500.0 / 24000, 3.0, p44, 1.0, 0.8f);
  wbo_engine_play(o2);
  wbx::AudioBuffer<float> in2(F2, C), out2(F2, C);
  for (uint32_t b = 0; b < 2 * NB; b++) {
    g_engine.process(in2, out2, (double)SR);
    wbo_engine_process(o2, optr, nullptr);
    for (uint32_t c = 0; c < C; c++)
      if (std::memcmp(out2.channel_buffers[c], optr[c], F2 * sizeof(float)) != 0) {
        std::printf("MISMATCH after set_audio_channel_config: block %u channel %u\n", b, c);
        return 6;
      }
  }
  wbo_engine_destroy(o2);
  if (g_engine.process_status.load() != WBX_OK) return 7;
  std::printf("adapter ok: %u + %u blocks bit-identical to the oracle\n", NB, 2 * NB);
  return 0;
}
