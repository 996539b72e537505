// wbx_runtime.hip — layer 1 of libwbx.so (wbx_ctx): clip pool in HBM, routing, plan upload, launches, result fetch.
// Layer 2 (the engine surface) lives in wbx_engine.hip, the multi-GPU exchange in wbx_dist.hip.
//
// There is no CPU implementation of the mix in this library: without a gfx950 device the create calls
// fail with WBX_ERR_NO_DEVICE.
#include "wbx_ctx.h"
#include "wbx_seq.h"

using namespace wbx;

namespace wbx {

// make the main stream wait for the sum that is still running beside it (device-side; a following
// sync_main(c) then covers it)
hipError_t join_sum(wbx_ctx* c) {
  if (c->sum_pending < 0) return hipSuccess;
  const hipError_t e = hipStreamWaitEvent(c->stream, c->sum_done[c->sum_pending], 0);
  c->sum_pending = -1;
  return e;
}

// the same for a mix that runs on the alternate stream
hipError_t join_alt(wbx_ctx* c) {
  if (c->alt_pending < 0) return hipSuccess;
  const hipError_t e = hipStreamWaitEvent(c->stream, c->mix_done[c->alt_pending], 0);
  c->alt_pending = -1;
  return e;
}

hipError_t sync_main(wbx_ctx* c) {
  hipError_t e = join_sum(c);
  if (e == hipSuccess) e = join_alt(c);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  return e;
}

// which stream the next mix runs on: batch renders of layer 2 alternate (see wbx_ctx::alt_stream)
hipStream_t pick_mix_stream(wbx_ctx* c, uint32_t K, bool alternate) {
  const bool alt = alternate && c->mix_alternate && c->alt_stream && K >= kOverlapMinBlocks && c->cur_mix_stream == c->stream;
  c->cur_mix_stream = alt ? c->alt_stream : c->stream;
  return c->cur_mix_stream;
}

// the kernel timer's pending launches -> the totals; `upto` > 0: only the oldest `upto` of them (a full ring: the newer half
// stays pending, so the submitting thread never waits for the launch it has just issued — waiting for THAT drained the whole
// run-ahead every 64 renders: ~9 us per step of a 0.4 ms render)
void drain_events(wbx_ctx* c, int upto) {
  const int n = (upto > 0 && upto < c->ev_pending) ? upto : c->ev_pending;
  for (int i = 0; i < n; i++) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[i][0], c->ev[i][1]) == hipSuccess) {
      c->mix_ms_total += ms;
      c->mix_launches++;
      if (hipEventElapsedTime(&ms, c->ev[i][1], c->ev[i][2]) == hipSuccess) c->tail_ms_total += ms;
      // the idle time between two consecutive mixes (end of one to start of the next, the kernels' own time stamps): what a
      // step spends outside its dominant kernel while the device runs back to back
      if (i > 0 && hipEventElapsedTime(&ms, c->ev[i - 1][1], c->ev[i][0]) == hipSuccess && ms >= 0.0f && ms < 50.0f) {
        c->gap_ms_total += ms;
        c->gap_count++;
      }
    }
  }
  for (int i = n; i < c->ev_pending; i++)   // (the handles change places: every slot keeps three valid events)
    for (int k = 0; k < 3; k++) std::swap(c->ev[i - n][k], c->ev[i][k]);
  c->ev_pending -= n;
}

// Routing tables.  `order`: the direct tracks in index order, then the members of bus 0, bus 1, ... — the order the sums
// run in.  Two group sets over it: `groups` cuts every member list into workgroup-sized pieces of group_size tracks
// (the master is the in-order sum of the piece sums), `groups_exact` keeps every member list whole — one workgroup walks
// all direct tracks of its block in track order, which IS the reference's summation (engine.cpp:1600-1617,
// audio_buffer.h:73-82: bit-exact master).  Which set a render takes: render_walks_whole_lists().
void build_routing(wbx_ctx* c, uint32_t n_tracks) {
  uint32_t G = c->cfg.group_size;
  // the callback configuration: the one block's latency is a chain of dependent work per workgroup (≈0.4 us a track row), so
  // the library cuts a session into MANY small groups and lets the spread sum of callback_kernel add them (tools/cb_groups.py,
  // profiles/r04_callback_groups.txt — us per 512-frame block, one group / the choice below: 24 tracks 27.6 / 25.9, 64 tracks
  // 42.9 / 25.6, 256 tracks 48.2 (64-track groups) / 26.1, 1024 tracks 47.0 / 30.5).  Up to 16 tracks: one group, whose
  // workgroup stores the master itself.  Up to 64 tracks: one track per group — the master is then the in-order sum of the
  // track buffers, which IS the reference's summation (engine.cpp:1600-1617), bit for bit as with one group.  Above that
  // groups of 4 / 8 / 16 (at most ≈64 workgroups below 1024 tracks: past that the ticket barrier's cost grows faster than
  // the rows per workgroup shrink; 4096 tracks: 256 workgroups, one per CU).
  if (c->auto_group && c->cfg.max_blocks == 1)
    G = n_tracks <= 16u ? kStage / 2 : n_tracks <= 64u ? 1u : n_tracks <= 256u ? 4u : n_tracks <= 512u ? 8u : kStage / 8;
  c->order.clear();
  c->groups.clear();
  c->groups_exact.clear();
  auto emit = [&](const std::vector<uint32_t>& members, int32_t bus) {
    const uint32_t base = (uint32_t)c->order.size();
    for (uint32_t t : members) c->order.push_back(t);
    for (size_t i = 0; i < members.size(); i += G) {
      DGroup g{};
      g.first = base + (uint32_t)i;
      g.count = (uint32_t)std::min<size_t>(G, members.size() - i);
      g.bus = bus;
      g.flags = (i ? GROUP_CHAIN_IN : 0u) | (i + G < members.size() ? GROUP_CHAIN_OUT : 0u);   // its place in the list (chained renders)
      c->groups.push_back(g);
    }
    if (!members.empty()) {
      DGroup g{};
      g.first = base;
      g.count = (uint32_t)members.size();
      g.bus = bus;
      c->groups_exact.push_back(g);
    }
  };
  std::vector<uint32_t> direct;
  std::vector<std::vector<uint32_t>> per_bus(c->n_buses);
  for (uint32_t t = 0; t < n_tracks; t++) {
    int32_t bus = (c->n_buses && t < c->track_bus.size()) ? c->track_bus[t] : -1;
    if (bus >= 0 && (uint32_t)bus < c->n_buses)
      per_bus[bus].push_back(t);
    else
      direct.push_back(t);
  }
  emit(direct, -1);
  for (uint32_t u = 0; u < c->n_buses; u++) emit(per_bus[u], (int32_t)u);
  c->routing_tracks = n_tracks;
  c->routing_dirty = true;
  // every bus exactly one group and nothing routed straight to the master: group g's partial sum IS bus g's sum, so
  // the bus output can alias the partial buffer instead of being written a second time by the sum kernel
  auto aliases = [&](const std::vector<DGroup>& gs) {
    if (c->n_buses == 0 || gs.size() != c->n_buses) return false;
    for (size_t g = 0; g < gs.size(); g++)
      if (gs[g].bus != (int32_t)g) return false;
    return true;
  };
  c->buses_alias_partials = aliases(c->groups);
  c->buses_alias_exact = aliases(c->groups_exact);
  c->longest_list = 0;
  for (auto& g : c->groups_exact) c->longest_list = std::max(c->longest_list, g.count);
}

// Does a render of K blocks take `groups_exact` (one workgroup per member list and block)?  Only when the library
// picks the grouping (wbx_config.group_size == 0) and the render is long enough to fill the device with one workgroup
// per block: the parallelism that track groups give a short render comes from the K blocks of a long one.  Measured on
// c3 (profiles/): from about a thousand blocks per render on, whole-list walks run at the grouped order's rate.
// What counts is the number of workgroup COLUMNS, not of blocks: the instances for blocks shorter than a workgroup put 2 or 4
// consecutive blocks into one, and a 1024-block render of 128-frame blocks through them is 256 columns — 256 chains, or 256
// walks, on a device that holds a thousand workgroups (measured: 0.30 of the roofline instead of 0.60).
// (-> blocks per workgroup; *resident: how many workgroups of that instance the device holds at once, in units of the 1024
//  that the four-wave instances come to.  A chained piece waits for its predecessor while it occupies a slot: with fewer
//  columns than resident workgroups several pieces of a block are resident TOGETHER and all but one of them wait — measured
//  on the one-wave instances, 3072 resident: 0.25 of the roofline at 1024 columns, 0.45 at 2048, against 0.6 unchained)
static uint32_t blocks_per_workgroup(const wbx_ctx* c, uint32_t K, uint32_t* resident) {
  const uint32_t C = c->cfg.channels, S4 = lane_span_of(c), lanes = C * S4;   // (the instance's lane space)
  *resident = 1u;
  if ((lanes % 256u == 0u) && (S4 % 64u == 0u)) return 1u;
  const bool short_ok = (C == 2u && S4 == 32u) || (S4 % 64u == 0u && lanes == 128u) || (S4 == 64u && lanes == 64u);
  const uint32_t packed = (C == 2u && S4 == 32u) ? 4u : (S4 % 64u == 0u && (lanes == 128u || lanes == 64u)) ? 256u / lanes : 1u;
  const int fam = mix_family(c);
  if (short_ok && c->has_cut_tracks && !c->knob_masked_rows_off) {   // masked rows ...
    if (!(fam == 2 && C == 2u && S4 == 64u) && packed_masked_variant(K, C == 2u && S4 == 32u, c->knob_packed_x)) return packed;   // ... in the packed instances
    *resident = lanes <= 64u ? 3u : 2u;
    return 1u;                                                   // ... in the one-block-per-workgroup instances
  }
  if (C == 2u && S4 == 64u && (fam == 0 || fam == 2) && (c->has_integer_clips || c->has_cut_tracks)) {   // one wave = one block
    *resident = 3u;
    return 1u;
  }
  return packed;
}

bool render_walks_whole_lists(const wbx_ctx* c, uint32_t K) {
  uint32_t resident = 1u;
  const uint32_t bpw = blocks_per_workgroup(c, K, &resident);
  return c->auto_group && c->exact_min_blocks != 0u && K / bpw >= c->exact_min_blocks * resident;
}

// ... and of those, which chain the workgroup-sized pieces instead of walking a list in one workgroup: the same order of
// additions, but scheduled like the grouped order (many short workgroups, dispatched dynamically) — a static assignment of
// one long walk per workgroup ends when its slowest shader engine does (profiles/r03_wg_clocks.txt: 25-40 % behind the mean).
// WBX_CHAIN=0: walk the lists whole.
bool render_chains_groups(const wbx_ctx* c, uint32_t K) {
  // (a reported hand-over failure: whole-list walks from then on.  WBX_MIX_ALT=1 runs two renders' mixes side by side, and
  //  the words of both would share d_chain with only the epoch to tell them apart: render i+1's pieces overwrite words
  //  render i's successors still poll — no chaining there)
  const bool off = c->knob_chain_off || c->chain_broken || c->mix_alternate;
  // (K a multiple of 32: every instance's grid then has an x extent that is a multiple of 8, which keeps the pieces of a
  //  block on one XCD — what the chain's L2-level hand-over rests on; other lengths walk the lists whole)
  return render_walks_whole_lists(c, K) && !off && c->longest_list > c->cfg.group_size && (K % 32u) == 0u;
}

wbx_status upload_tables(wbx_ctx* c, uint32_t n_tracks) {
  if (c->routing_tracks != n_tracks) {
    build_routing(c, n_tracks);
    c->buses_clean = false;
  }
  if (c->routing_dirty) {
    WBX_HIP(c, join_sum(c));                          // a sum beside the main stream may still read d_groups
    WBX_HIP(c, sync_main(c));
    WBX_HIP(c, c->d_order.ensure(std::max<size_t>(1, c->order.size())));
    WBX_HIP(c, c->d_groups.ensure(std::max<size_t>(1, c->groups.size() + c->groups_exact.size())));   // [groups | groups_exact]
    if (!c->order.empty())
      WBX_HIP(c, hipMemcpyAsync(c->d_order.p, c->order.data(), c->order.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                                c->stream));
    if (!c->groups.empty())
      WBX_HIP(c, hipMemcpyAsync(c->d_groups.p, c->groups.data(), c->groups.size() * sizeof(DGroup),
                                hipMemcpyHostToDevice, c->stream));
    if (!c->groups_exact.empty())
      WBX_HIP(c, hipMemcpyAsync(c->d_groups.p + c->groups.size(), c->groups_exact.data(),
                                c->groups_exact.size() * sizeof(DGroup), hipMemcpyHostToDevice, c->stream));
    WBX_HIP(c, sync_main(c));   // host vectors may change right after
    c->routing_dirty = false;
  }
  if (c->samples_dirty) {
    std::vector<DSample> tab(c->clips.size());
    c->has_integer_clips = false;
    c->has_non16_clips = false;
    for (size_t i = 0; i < c->clips.size(); i++) {
      tab[i] = c->clips[i].d;
      if (c->clips[i].used && c->clips[i].d.format != FMT_F32) c->has_integer_clips = true;
      if (c->clips[i].used && c->clips[i].d.format != FMT_I16) c->has_non16_clips = true;
    }
    WBX_HIP(c, c->d_samples.ensure(std::max<size_t>(1, tab.size())));
    if (!tab.empty()) WBX_HIP(c, hipMemcpy(c->d_samples.p, tab.data(), tab.size() * sizeof(DSample), hipMemcpyHostToDevice));
    c->samples_dirty = false;
  }
  return WBX_OK;
}

wbx_status ensure_result_buffers(wbx_ctx* c, uint32_t K, uint32_t N) {
  const size_t CF = (size_t)c->cfg.channels * c->cfg.block_frames;
  for (auto& B : c->pb)
    if (B.prows.cap < (size_t)K * N) {
      WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
      WBX_HIP(c, sync_main(c));
      WBX_HIP(c, B.prows.ensure((size_t)K * N));
    }
  const size_t need_partial = (size_t)K * std::max<size_t>(1, c->groups.size()) * CF;
  if (c->d_partial2[0].cap < need_partial || c->d_master.cap < (size_t)K * CF ||
      c->d_peaks[0].cap < (size_t)K * N * c->cfg.channels || (c->n_buses && c->d_buses.cap < (size_t)K * c->n_buses * CF)) {
    WBX_HIP(c, join_sum(c));                          // a sum may still be using the buffers about to be replaced
    WBX_HIP(c, sync_main(c));
    for (auto& v : c->sum_valid) v = false;
  }
  for (auto& P : c->d_partial2) WBX_HIP(c, P.ensure(need_partial));
  WBX_HIP(c, c->d_master.ensure((size_t)K * CF));
  for (auto& P : c->d_peaks) WBX_HIP(c, P.ensure((size_t)K * N * c->cfg.channels));
  if (c->n_buses && c->d_buses.cap < (size_t)K * c->n_buses * CF) {
    WBX_HIP(c, sync_main(c));
    WBX_HIP(c, c->d_buses.ensure((size_t)K * c->n_buses * CF));
    c->buses_clean = false;
  }
  return WBX_OK;
}

// room for `n` templates in both plan buffers (grow-only)
wbx_status ensure_template_capacity(wbx_ctx* c, size_t n) {
  n = std::max<size_t>(n, 64);
  for (auto& B : c->pb) {
    if (n <= B.tmpl_cap) continue;
    WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
    WBX_HIP(c, sync_main(c));
    WBX_HIP(c, B.tmpl.ensure(n));
    B.tmpl_cap = (uint32_t)n;
  }
  return WBX_OK;
}

// room for `rows` pre-rendered generic track-blocks (grow-only)
wbx_status ensure_gen_capacity(wbx_ctx* c, size_t rows) {
  rows = std::max<size_t>(rows, 64);
  const size_t row_floats = (size_t)c->cfg.channels * (c->cfg.block_frames + 8);
  for (auto& B : c->pb) {
    if (rows <= B.gen_cap) continue;
    WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
    WBX_HIP(c, sync_main(c));
    WBX_HIP(c, B.gen_list.ensure(rows));
    WBX_HIP(c, B.rows.ensure(rows * row_floats));
    WBX_HIP(c, B.saved.ensure(rows));
    B.gen_cap = (uint32_t)rows;
  }
  return WBX_OK;
}

// The overflow pool (stream calls 3.. of a block) at twice its default size, once: a render planned by segments leaves the
// chunks of replaced segments allocated (grow-only; a host that set wbx_config.max_segments keeps exactly that budget).
wbx_status ensure_pool_slack(wbx_ctx* c) {
  if (c->cfg.max_segments) return WBX_OK;
  const size_t chunks = 2 * std::max<size_t>(1024, (size_t)c->cfg.max_blocks * c->cfg.max_tracks / 8);
  for (auto& B : c->pb) {
    if (chunks <= B.pool_chunks) continue;
    WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
    WBX_HIP(c, sync_main(c));
    WBX_HIP(c, B.pool.ensure(chunks * kChunk));
    B.pool_chunks = (uint32_t)chunks;
  }
  return WBX_OK;
}

// pre-render of the queued generic records of the current plan buffer
wbx_status launch_pre_render(wbx_ctx* c, uint32_t K, hipStream_t on) {
  const uint32_t C = c->cfg.channels, F = c->cfg.block_frames;
  GenArgs ga{};
  ga.tmpl = PB(c).tmpl.p;
  ga.pool = PB(c).pool.p;
  ga.gen_list = PB(c).gen_list.p;
  ga.gen_count = PB(c).counters + 2;
  ga.rows = PB(c).rows.p;
  ga.saved = PB(c).saved.p;
  ga.gen_cap = PB(c).gen_cap;
  ga.block_frames = F;
  ga.channels = C;
  // one wave per queued row, grid-stride: no more workgroups than the device holds at once (256 CUs x 6 workgroups at
  // the kernel's register budget), or the surplus would start when the first ones have finished their whole share
  // (a plan made for a masked-row mix instance queues only what is left over: blocks with three or more stream calls,
  //  overlapping calls — a handful per render at most)
  launch_gen(ga, K < kOverlapMinBlocks ? 64u * K : c->masked_rows ? 128u : 1536u, on);
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

// which chunk modes the mix instance of the next launch carries (mix_kernel<.., FAM, ..>): 1 (everything) also holds the
// pipelined modes for chunks that mix storage formats with resampled rows; 2: sessions of 16-bit PCM only, resampled at
// speeds up to 0.999 or not at all
int mix_family(const wbx_ctx* c) {
  if (c->force_g) return 1;
  if (c->has_lean16_clips && !c->has_non16_clips && !c->knob_no_lean16) return 2;
  if (c->has_stride_clips || (c->has_window_clips && c->has_integer_clips))
    return (c->has_taps_clips || c->knob_no_fam3) ? 1 : 3;   // (3 = 1 without the per-frame taps)
  return 0;
}

// stereo sessions with integer-PCM clips or with tracks cut into several clips, blocks of 256 / 512 / 1024 frames: the
// instances with both channels of a frame in one lane (position and masked-row arithmetic once per frame, one set of record
// scalars for both channels).  Measured (tools/ab_cl2.sh, tools/ab_masked.sh, tools/ab_blocks.sh; slab-allocated sessions):
// integer PCM +2-10 %, sessions cut into clips +5-11 % (2 x at 256 frames, where the other instances have no masked
// rows), fp32 sessions of one clip per track 3-8 % slower (they fetch 1.06 x their bytes instead of 1.02 x) — those keep
// one channel per wave.
// fp32 sessions of one clip per track with resampled clips (c3), chained renders of 2048 blocks and more: the two-channels-per-
// lane instance with ONE row per pipeline batch, <1,true,3,0,1,1,2,128>.  Round 2 measured these sessions 3 % slower through
// the CL = 2 instances — at 256-block renders in the grouped order; at 2048 chained blocks, with the one-ratio modes taken
// again, five alternating runs on one box (profiles/r05_ab_c3_instances.txt) read 0.711 of the roofline for it, 0.701 for
// <2,true,3,..,2,128>, 0.694 for the one-channel-per-wave <2,true,4,..,1,256>; at 1024 and 256 blocks nothing to choose.
bool mix_long_chained_window_render(const wbx_ctx* c) {
  static const bool off = [] { const char* v = std::getenv("WBX_NO_LONG_CL2"); return v && v[0] == '1'; }();   // A/B aid
  const uint32_t F = 4u * lane_span_of(c);
  return !off && c->cfg.channels == 2u && F == 512u && c->chain_now && c->render_blocks_now >= 2048u && c->has_window_clips &&
         !c->has_integer_clips && !c->has_cut_tracks && !c->has_stride_clips && !c->n_buses;
}

bool mix_two_channels_per_lane(const wbx_ctx* c) {
  const uint32_t F = 4u * lane_span_of(c);   // (the block size of the instance's lane space)
  if (c->mix_unroll) return c->mix_unroll >= 1000;   // WBX_MIX_VARIANT
  if (c->cfg.channels != 2u || c->knob_no_cl2) return false;
  if (!(F == 512u || F == 1024u || F == 256u)) return false;
  // the callback path (a handful of workgroups, each a chain of dependent rows): a wave per channel half — four waves share
  // the chain instead of two (measured, 4096 / 64 tracks: 16-bit resampled 53 -> 51 / 55 -> 49 us, cut into clips 63 -> 58 /
  // 66 -> 58, 24-bit 55 -> 53 / 58 -> 53).  256-frame blocks keep the one-wave instances: only those take their masked rows.
  if (c->short_render_now && F != 256u) return false;
  // a render whose workgroups walk whole member lists of many staged chunks: the half-size workgroups of these instances
  // put six of them on a CU, and the walk runs 15 % faster than through the four-wave ones (c3, 1024 blocks: 2.94 vs 3.43 ms)
  if (c->whole_lists_now && !c->chain_now && c->longest_list > 2u * kStage) return true;
  if (mix_long_chained_window_render(c)) return true;
  return c->has_integer_clips || c->has_cut_tracks;
}

// Can the mix instance a render of this shape will launch take masked rows (partial-coverage records, ROW_PAIRs) in its
// hot loop, and which (PlanArgs::masked_rows)?  Only the lean whole-workgroup-per-block instances do
// (mix_kernel<U, true, W, false, 1, ...>): blocks of C*F/4 lanes a multiple of 256, sessions without per-frame-tap clips.
// 1: fp32 rows, unity or resampled; 2: also integer PCM at unity speed — sessions whose integer clips all play at the
// session rate and that hold no resampled clip (those take the instances with the mixed-format window modes).
uint32_t mix_takes_masked_rows(const wbx_ctx* c, bool window_clips, bool stride_clips) {
  const uint32_t S4 = lane_span_of(c), lanes = c->cfg.channels * S4;
  bool full = (lanes % 256u == 0u) && (S4 % 64u == 0u);
  // (256-frame stereo blocks: the one-wave instances with both channels per lane, the lean families only)
  if (!full && c->cfg.channels == 2u && S4 == 64u && (mix_family(c) == 0 || mix_family(c) == 2) && mix_two_channels_per_lane(c)) full = true;
  // short blocks — 128-frame stereo, 256-frame stereo in the families without that one-wave instance, 256 / 512-frame mono:
  // one-block-per-workgroup instances (a wave or two) exist for families 0 and 1; a session with tracks cut into clips takes
  // them, the others keep the instances that put 2 or 4 blocks into a workgroup
  if (!full && c->has_cut_tracks &&
      ((c->cfg.channels == 2u && S4 == 32u) || (S4 % 64u == 0u && lanes == 128u) || (S4 == 64u && lanes == 64u)))
    full = true;
  if (c->knob_masked_rows_off) return 0u;   // WBX_MASKED_ROWS=0, A/B aid: send every boundary row through the pre-render pass
  if (!full) return 0u;
  if (mix_family(c) == 1 || mix_family(c) == 3) return 4u;   // the everything family: every row kind it streams, also as a masked row
  if (mix_family(c) == 2) return 3u;   // sessions of 16-bit PCM only: also their resampled rows
  if (stride_clips) return 0u;
  if (!c->has_integer_clips) return 1u;
  return window_clips ? 0u : 2u;
}

// Does a one-block render of wbx_engine_process run as ONE launch (wbx_callback.h)?  Blocks that are exactly one 256-lane
// workgroup (512-frame stereo, 1024-frame mono: the instances that exist), no multi-GPU exchange.  WBX_CALLBACK_FUSED=0: the
// three launches of earlier rounds (A/B aid; results are identical).
// (round 5) ... and every block that FITS one: the callback is a latency path — what counts is one dispatch instead of three, not
// how many of the workgroup's lanes own frames — so a 128- or 256-frame stereo block (the low-latency settings of
// ui/settings.cpp:22-24) runs through the same 256-lane instance with lane_span = 256 / C, its surplus lanes cloning the block's
// last four frames (wbx_mix.h).  WBX_CB_ANY=0: only the shapes whose batch renders take a 256-lane workgroup per block.
uint32_t callback_lane_span(const wbx_ctx* c) {
  const uint32_t C = c->cfg.channels, S4 = c->cfg.block_frames >> 2;
  const uint32_t nat = lane_span_of(c);
  if (C * nat == 256u && (nat % 64u) == 0u) return nat;
  if (c->knob_cb_any_off || c->knob_ragged_off) return 0u;
  return C * S4 <= 256u ? 256u / C : 0u;
}
bool callback_is_one_launch(const wbx_ctx* c) {
  static const bool off = [] { const char* v = std::getenv("WBX_CALLBACK_FUSED"); return v && v[0] == '0'; }();
  return !off && !c->dist && callback_lane_span(c) != 0u && !c->mix_unroll;
}

// where the master of the render about to be issued goes; `writer` is the stream its last writer runs on
float* begin_master(wbx_ctx* c, hipStream_t writer, hipError_t* err) {
  *err = hipSuccess;
  if (c->dist) return dist_begin_render(c, writer, err);
  return c->master_target ? c->master_target : c->d_master.p;
}

// mix + sum over the current plan buffer, on the main stream
wbx_status launch_mix_sum(wbx_ctx* c, uint32_t K, uint32_t N) {
  const uint32_t C = c->cfg.channels, F = c->cfg.block_frames;
  MixArgs m{};
  m.rows = PB(c).prows.p;
  m.tmpl = PB(c).tmpl.p;
  m.zero_page = c->d_zero.p;
  m.pool = PB(c).pool.p;
  m.order = c->d_order.p;
  // the group set of this render: workgroup-sized pieces, or the whole member lists (the reference's summation order)
  const bool chained = c->whole_lists_now && render_chains_groups(c, K);
  const bool whole = c->whole_lists_now && !chained;
  const DGroup* d_groups = c->d_groups.p + (whole ? c->groups.size() : 0);
  const uint32_t n_groups = (uint32_t)(whole ? c->groups_exact.size() : c->groups.size());
  const bool buses_alias = whole ? c->buses_alias_exact : c->buses_alias_partials;
  m.groups = d_groups;
  const int pp = (int)(c->render_seq % kRing);
  // this partial buffer was last read by the sum of kRing renders ago; the engine path has already made the PLAN
  // stream wait for that sum (the mix waits for the plan), which keeps the barrier off the main stream
  if (!c->cur_mix_stream) c->cur_mix_stream = c->stream;
  hipStream_t ms = c->cur_mix_stream;             // the main stream, or the alternate one (pick_mix_stream)
  const int pk = ms == c->stream ? 0 : 1;         // its peaks buffer
  if (c->sum_valid[pp] && !c->partial_wait_done) WBX_HIP(c, hipStreamWaitEvent(ms, c->knob_partial_free_off ? c->sum_done[pp] : c->partial_free[pp], 0));
  c->partial_wait_done = false;
  m.partial = c->d_partial2[pp].p;
  m.peaks = c->d_peaks[pk].p;
  c->last_peaks = m.peaks;
  m.levels = c->levels_target;
  // short renders (the one-block callback path above all) keep everything on the main stream: the cross-stream
  // hand-overs cost more than the few microseconds of overlap they could buy
  const bool sum_beside = c->sum_overlap && K >= kOverlapMinBlocks;
  hipStream_t ss = sum_beside ? c->sum_stream : c->stream;
  // (argument checks first: once dist_mix_init has posted this render's receive, a failure would leave it unmatched)
  if (c->master_format && c->dist) return fail(c, WBX_ERR_UNSUPPORTED, "wbx_set_master_format: a multi-GPU partial master stays planar fp32");
  if (c->n_buses && (c->master_init || dist_receives_running_sum(c)))
    return fail(c, WBX_ERR_UNSUPPORTED, "a running master (wbx_set_master_init / WBX_DIST_CHAIN) cannot be continued through sub-buses");
  // where this render's master goes (the ctx's own buffer, the caller's target, or the multi-GPU ring slot) and what its
  // sum starts from (zero; wbx_set_master_init's running sum; chain mode: the previous rank's, received for this render)
  float* master_dst = nullptr;
  {
    hipError_t me = hipSuccess;
    master_dst = begin_master(c, ss, &me);
    WBX_HIP(c, me);
    wbx_status ist = WBX_OK;
    m.init = c->dist ? dist_mix_init(c, K, ms, &ist) : c->master_init;
    if (ist != WBX_OK) return ist;
  }
  m.n_tracks = N;
  m.n_groups = n_groups;
  m.block_frames = F;
  m.channels = C;
  m.lane_span = lane_span_of(c);
  m.packed_x = c->knob_packed_x;
  m.fast_partial = c->knob_fast_partial_off ? 0u : 1u;
  m.tiles = (C * m.lane_span + 255u) / 256u;
  m.n_blocks = K;
  m.masked_rows = c->masked_rows ? 1u : 0u;
  {   // one resampling ratio for every window row of this render (layer 2's word; MODE_WNU / WINU: the products fl(j * speed)
      // hoisted out of the track loop).  WBX_NO_UNIFORM=1: A/B aid.  [Round 3's split of this function lost this line: the
      // modes were carried but never taken until round 5 — SQ_INSTS_VALU_MUL_F64 of the r03-r05 PMC passes shows it.]
    m.uniform_speed = c->knob_no_uniform ? 0.0 : c->uniform_speed;
    c->last_uniform_speed = m.uniform_speed;
  }
  // A short render of a session that is one group (the callback configuration up to 64 tracks; no sub-buses, planar fp32
  // master, nothing to continue): the mix workgroup clamps and stores the master itself — the sum kernel is not launched
  // A short render's sum runs on the main stream (and a one-group session's mix stores the master itself): neither may
  // overtake the sum of a batch render that is still pending on the sum stream — they write the same master (and bus)
  // buffers, and the late one would overwrite the head of the short render's blocks.  (Found by the edit scripts once they
  // stopped fetching every render: nobody had made the main stream wait, because a fetch in between always had.)
  if (ss == c->stream) WBX_HIP(c, join_sum(c));
  bool fused = false;
  {
    static const bool off = [] { const char* v = std::getenv("WBX_FUSE_SUM"); return v && v[0] == '0'; }();   // A/B aid
    fused = !off && K < kOverlapMinBlocks && n_groups == 1u && c->n_buses == 0u && m.tiles == 1u && !c->master_format && !c->dist &&
            !m.init && !chained;
    m.fused_master = nullptr;
    if (fused) {
      m.fused_master = master_dst;
      m.fused_clamp = c->clamp ? 1u : 0u;
      m.fused_status_src = c->status_dst ? PB(c).counters : nullptr;
      m.fused_status_dst = c->status_dst;
      m.fused_zero_status = (c->status_dst && c->zero_status) ? 1u : 0u;
    }
  }
  m.chain = nullptr;
  m.chain_status = nullptr;
  m.chain_sticky = nullptr;
  if (chained) {   // one "sum is out" word per (workgroup column, group), tagged with this render's epoch: no clearing
    const size_t words = (size_t)K * m.tiles * n_groups;   // between renders (a memset here waits out the previous sum)
    const size_t had = c->d_chain.cap;
    WBX_HIP(c, c->d_chain.ensure(words));
    c->chain_epoch = (c->chain_epoch + 1u) & 0x0FFFFFFFu;
    if (c->d_chain.cap != had || c->chain_epoch == 0u) {   // fresh storage, or the tag wrapped: start over from zeroed words
      WBX_HIP(c, hipMemsetAsync(c->d_chain.p, 0, c->d_chain.cap * sizeof(uint32_t), ms));
      if (c->chain_epoch == 0u) c->chain_epoch = 1u;
    }
    m.chain_epoch = c->chain_epoch;
    m.chain = c->d_chain.p;
    m.chain_status = PB(c).counters + 1;
    m.chain_sticky = c->d_sticky_status;
  }
  {
    static const bool dbg = std::getenv("WBX_DBG_CLOCK") != nullptr;   // diagnostic: per-workgroup start / end times of the mix
    m.dbg_clock = nullptr;
    if (dbg) {
      c->dbg_wgs = (size_t)K * n_groups * m.tiles;
      WBX_HIP(c, c->d_dbg.ensure(4 * c->dbg_wgs));
      m.dbg_clock = c->d_dbg.p;
    }
  }
  if (m.tiles > 1) WBX_HIP(c, hipMemsetAsync(m.peaks, 0, (size_t)K * N * C * sizeof(float), ms));
  // the kernel timer is for batch renders; the one-block callback path skips its three event records
  const bool timed = c->profiling && K > 1;
  const bool one_launch = c->cb_plan != nullptr && K == 1u && m.n_groups != 0u && callback_is_one_launch(c);
  if (one_launch) {   // (the callback instance's own lane space: one 256-lane workgroup per block)
    m.lane_span = callback_lane_span(c);
    m.tiles = 1u;
  }
  c->cb_launched = false;
  bool by_kernel_event = false;
  if (m.n_groups && !one_launch) {
    if (timed) {
      if (c->ev_pending == kEventRing) {   // the older half: 32 launches behind the newest, over long ago
        WBX_HIP(c, hipEventSynchronize(c->ev[kEventRing / 2 - 1][2]));
        drain_events(c, kEventRing / 2);
      }
    }
    // the timer's two events ride on the kernel's own dispatch packet (start / end time stamps of the kernel itself): no
    // event packets between two mixes.  WBX_TIMER_PACKETS=1: the old way, an event record either side (A/B aid)
    static const bool packets = std::getenv("WBX_TIMER_PACKETS") != nullptr;
    if (timed && packets) WBX_HIP(c, hipEventRecord(c->ev[c->ev_pending][0], ms));
    // (unity-speed fp32 sessions: four rows per pipeline batch at three waves per SIMD, <4,true,3,..> — round 2's choice at
    //  256-block renders; in renders of >= 2048 blocks of large sessions two rows at four waves, <2,true,4,..>, is ahead:
    //  c4 0.72-0.75 of the roofline against 0.67-0.69, u4096 0.755-0.777 against 0.746-0.763, one box, alternating
    //  (profiles/r05_ab_c3_instances.txt); 256-track sessions: nothing to choose.  WBX_NO_LONG_24=1: the old choice)
    static const bool no_long_24 = [] { const char* v = std::getenv("WBX_NO_LONG_24"); return v && v[0] == '1'; }();
    const bool long_large = !no_long_24 && K >= 2048u && N >= 1024u;
    c->mix_kernel_name = launch_mix(m, K, c->mix_unroll ? c->mix_unroll : mix_two_channels_per_lane(c) ? (mix_long_chained_window_render(c) ? 1013 : 1023)
                                          : ((c->has_window_clips || c->has_integer_clips || long_large) ? 24 : 43),
               mix_family(c), ms, (timed && !packets) ? c->ev[c->ev_pending][0] : nullptr,
               (timed && !packets) ? c->ev[c->ev_pending][1] : nullptr);
    if (timed && packets) WBX_HIP(c, hipEventRecord(c->ev[c->ev_pending][1], ms));
    if (c->dist) WBX_HIP(c, dist_mix_issued(c, ms));
    // (round 6) When the kernel carries the timer's stop event and its sum runs on another stream, THAT event is what the sum
    // stream waits for, and mix_done — what later plans wait for — is recorded over there: no marker packet behind the mix on
    // its own stream, where the next mix queues (every packet between two mixes is a round trip of the command processor to
    // the queue in host memory, behind whatever the copy engine is posting).  WBX_MIX_MARKER=1: the marker, as until round 5.
    by_kernel_event = timed && !packets && !c->knob_mix_marker && !c->dist && !c->mix_alternate && ms == c->stream && sum_beside;
  }
  // the plan buffer is free as soon as the MIX has read it: releasing it before the sum lets the next plan run
  // beside sum_kernel (the GPU is nearly idle there) instead of competing with the next mix for CU slots — started
  // together with a mix, the one-wave-per-track plan kernel is starved until that mix drains
  if (!one_launch && !by_kernel_event) WBX_HIP(c, hipEventRecord(c->mix_done[pp], ms));   // (one launch: recorded behind it, below)
  if (ms != c->stream) c->alt_pending = pp;
  // A master bound for pinned host memory leaves a batch render through a device staging buffer and the copy engine.  Stored
  // by the sum kernel itself, its megabytes of posted writes fill the GPU's upstream queue in a few microseconds and drain at
  // the PCIe rate; the command processor's own reads of host memory (the next dispatch packet, its signals) wait behind
  // them, and the next mix started only when the sum had ended — 0.15 ms per 2048-block render.  (One-block callbacks keep
  // the direct stores: 4 KB, and no second operation in the latency path.  Packed 24-bit too: its writer leaves part of
  // the target untouched, which a whole-buffer copy would not.)
  float* const master_home = master_dst;
  size_t stage_bytes = 0;
  if (sum_beside && !c->dist && c->master_target && c->master_target_on_host && c->master_format != 5) {
    static const bool direct = [] { const char* v = std::getenv("WBX_HOST_MASTER_DIRECT"); return v && v[0] == '1'; }();   // A/B aid
    // (from 4 MB: below, the copy's fixed cost on the sum stream outweighs the hold-up — a 256-track session rendering 256
    //  blocks at a time lost a quarter of its rate to it, and gains a quarter at 2048)
    const size_t bytes = (size_t)K * F * C * (c->master_format == 3 ? 2u : 4u);
    if (!direct && bytes >= (4u << 20)) {
      stage_bytes = bytes;
      WBX_HIP(c, c->d_stage[pp].ensure((stage_bytes + 3u) / 4u));
      master_dst = c->d_stage[pp].p;
    }
  }
  SumArgs s{};
  s.partial = c->d_partial2[pp].p;
  s.groups = d_groups;
  s.master = master_dst;
  if (c->master_format) {   // the device format is the sum's epilogue: interleaved samples instead of planar fp32
    s.out_il = master_dst;
    s.out_format = (uint32_t)c->master_format;
  }
  c->last_master_format = c->master_format;
  c->last_master = master_home;
  c->last_master_on_host = false;
  s.buses = (c->n_buses && !buses_alias) ? c->d_buses.p : nullptr;
  c->last_buses = c->n_buses ? (buses_alias ? c->d_partial2[pp].p : c->d_buses.p) : nullptr;
  s.n_groups = m.n_groups;
  s.n_buses = c->n_buses;
  s.block_frames = F;
  s.channels = C;
  s.clamp = c->clamp ? 1u : 0u;
  s.chain = chained ? 1u : 0u;
  s.status_src = c->status_dst ? PB(c).counters : nullptr;
  s.status_dst = c->status_dst;
  s.zero_status = (c->status_dst && c->zero_status) ? 1u : 0u;
  if (by_kernel_event && ss != ms) {
    WBX_HIP(c, hipStreamWaitEvent(ss, c->ev[c->ev_pending][1], 0));   // the kernel's own completion ...
    WBX_HIP(c, hipEventRecord(c->mix_done[pp], ss));                   // ... and, over here, "the mix has read its plan buffer"
  } else if (by_kernel_event) {
    WBX_HIP(c, hipEventRecord(c->mix_done[pp], ms));
  } else if (ss != ms) {
    WBX_HIP(c, hipStreamWaitEvent(ss, c->mix_done[pp], 0));   // (an earlier pending sum is ordered before this one by ss)
  }

  if (c->n_buses && !buses_alias && !c->buses_clean) {
    // buses without member groups must read as zero; every bus that has members is rewritten by each render, so the
    // buffer only needs clearing when the routing or the allocation changed (64 MB per render saved on config 4).
    // On the SUM's stream, in front of the sum: cleared on the mix stream behind the mix, it raced with a sum that runs
    // beside (renders of 8 blocks and more) and could wipe the bus sums of the first render after a routing change —
    // found by the randomised sessions once they drew renders that long.
    WBX_HIP(c, hipMemsetAsync(c->d_buses.p, 0, c->d_buses.cap * sizeof(float), ss));
    c->buses_clean = true;
  }
  if (ss == c->stream && ms != c->stream) c->alt_pending = -1;            // (the main stream has just joined that mix)
  if (one_launch) {
    // sequencer + mix + sum in one dispatch; the kernel itself tells the host when master and status are out (cb_flag)
    if (!c->d_cb_done) {
      WBX_HIP(c, hipMalloc((void**)&c->d_cb_done, kCbDoneWords * sizeof(uint32_t)));   // (two spread counters: wbx_callback.h)
      WBX_HIP(c, hipMemsetAsync(c->d_cb_done, 0, kCbDoneWords * sizeof(uint32_t), ms));
      c->cb_base = c->cb_base2 = 0;
    }
    unsigned long long* cb_dbg = nullptr;
    {
      static const bool dbg = std::getenv("WBX_CB_DBG") != nullptr;   // diagnostic: the phases of every workgroup (tools/cb_clocks.py)
      if (dbg) {
        c->dbg_wgs = ((size_t)6 * m.n_groups + 3) / 4;
        WBX_HIP(c, c->d_dbg.ensure(4 * c->dbg_wgs));
        cb_dbg = c->d_dbg.p;
      }
    }
    m.partial_through = c->knob_cb_fenced ? 0u : 1u;
    // every workgroup adds a share of the master when the whole grid is resident at once (at most one workgroup per CU: two
    // fit) and the engine's pinned block has a completion word for each of them
    const bool spread = !fused && !c->cb_no_spread && m.n_groups <= callback_spread_limit();
    c->cb_flags = 1u;
    c->mix_kernel_name = launch_callback(m, *c->cb_plan, s, c->d_cb_done, c->cb_base, c->cb_base2, spread, c->cb_gave_up, c->cb_spin_bound, c->cb_flag, c->cb_seq, mix_family(c),
                                         c->has_window_clips || c->has_integer_clips, cb_dbg, ms);
    // (a one-group block takes no ticket; a grid larger than the device — "the last workgroup adds everything" — only the
    //  first one: the second counter has a base of its own, or the first spread launch after such a block would wait for a
    //  count that wrapped)
    if (!fused) c->cb_base += m.n_groups;
    if (spread) c->cb_base2 += m.n_groups;
    c->cb_launches++;
    if (spread) c->cb_spread_launches++;
    c->cb_launched = true;
    // (no event behind it: wbx_engine_process waits for the launch's own word before it returns, nothing can overlap it)
  } else if (!fused) {
    launch_sum(s, K, ss);
  }
  if (sum_beside) WBX_HIP(c, hipEventRecord(c->partial_free[pp], ss));   // (in front of the staged master's copy: wbx_ctx.h)
  if (stage_bytes) WBX_HIP(c, hipMemcpyAsync(master_home, master_dst, stage_bytes, hipMemcpyDeviceToHost, ss));
  if (m.n_groups && timed) {
    WBX_HIP(c, hipEventRecord(c->ev[c->ev_pending][2], ss));
    c->ev_pending++;
  }
  if (sum_beside) {
    WBX_HIP(c, hipEventRecord(c->sum_done[pp], ss));
    c->sum_valid[pp] = true;
    c->sum_pending = pp;
  }
  c->render_seq++;
  WBX_HIP(c, hipGetLastError());
  c->last_K = K;
  c->last_N = N;
  return WBX_OK;
}

}  // namespace wbx

// =================================================================================================
// library
// =================================================================================================
extern "C" const char* wbx_version(void) { return "wbx 0.1 (gfx950)"; }

extern "C" int wbx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

extern "C" const char* wbx_status_string(wbx_status s) {
  switch (s) {
    case WBX_OK: return "ok";
    case WBX_ERR_FAILED: return "failed";
    case WBX_ERR_UNIMPLEMENTED: return "unimplemented";
    case WBX_ERR_UNSUPPORTED: return "unsupported";
    case WBX_ERR_INVALID: return "invalid argument";
    case WBX_ERR_NO_DEVICE: return "no gfx950 device";
    case WBX_ERR_DEVICE: return "HIP error";
    case WBX_ERR_OOM: return "out of memory";
    case WBX_ERR_OVERFLOW: return "segment plan overflow";
    default: return "unknown";
  }
}

// =================================================================================================
// layer 1
// =================================================================================================
extern "C" wbx_status wbx_create(const wbx_config* cfg, wbx_ctx** out) {
  if (!cfg || !out) return WBX_ERR_INVALID;
  *out = nullptr;
  if (cfg->channels < 1 || cfg->channels > 2 || cfg->block_frames < 4 || (cfg->block_frames & 3u) ||
      cfg->block_frames > 32768 || cfg->max_tracks == 0 || cfg->max_blocks == 0 || cfg->max_blocks > 4096 ||
      cfg->sample_rate == 0)
    return WBX_ERR_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || cfg->device < 0 || cfg->device >= n) return WBX_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return WBX_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return WBX_ERR_NO_DEVICE;   // kernels exist for gfx950 only
  if (hipSetDevice(cfg->device) != hipSuccess) return WBX_ERR_NO_DEVICE;

  wbx_ctx* c = new (std::nothrow) wbx_ctx();
  if (!c) return WBX_ERR_OOM;
  c->cfg = *cfg;
  // default 128: one staging round per workgroup; a context that can only render one block per call (the audio
  // callback) takes many small groups instead — one group up to 16 tracks, ONE TRACK per group up to 64 (both: the
  // reference's summation order, bit for bit), 4 / 8 / 16 tracks above 64 / 256 / 512 (build_routing) — more workgroups
  // for the one block, whose mix is a latency chain per workgroup; the block's group sums are added by the callback
  // kernel's spread sum (by sum_kernel when the block shape takes the three-launch path).
  c->auto_group = c->cfg.group_size == 0;
  if (c->cfg.group_size == 0) c->cfg.group_size = c->cfg.max_blocks == 1 ? kStage / 2 : kStage;
  if (const char* u = std::getenv("WBX_MIX_VARIANT")) c->mix_unroll = std::atoi(u);
  if (const char* u = std::getenv("WBX_EXACT_MIN_BLOCKS")) c->exact_min_blocks = (uint32_t)std::atoi(u);   // 0: never
  if (const char* u = std::getenv("WBX_FORCE_G")) c->force_g = std::atoi(u) != 0;   // A/B aid: always the G instances
  if (const char* u = std::getenv("WBX_KERNEL_TIMER")) c->profiling = std::atoi(u) != 0;   // 0: no HIP-event kernel timer
  // Events whose only waiters are other streams of this device (or a host that waits for "done" and reads nothing the device
  // wrote): released to the DEVICE — a marker's default release is to the system.  (No measurable effect by itself; the A/Bs
  // that seemed to show one were reading the copy engine's two speeds: EXPERIMENTS.md.)  Results leave through sum_done and the
  // callback's own system-scope stores, which keep the system scope.  WBX_EVENT_SCOPE=system: as until round 5 (A/B aid).
  const unsigned scope = [] { const char* es = std::getenv("WBX_EVENT_SCOPE"); return (es && es[0] == 's') ? 0u : (unsigned)hipEventReleaseToDevice; }();
  c->dev_event_flags = hipEventDisableTiming | scope;
  if (cfg->stream) {
    c->stream = (hipStream_t)cfg->stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return WBX_ERR_DEVICE;
    }
    c->own_stream = true;
  }
  for (int i = 0; i < kEventRing; i++) {
    // (timing events: only their time stamps are read — no release to the system behind the kernel that carries them)
    if (hipEventCreateWithFlags(&c->ev[i][0], scope) != hipSuccess || hipEventCreateWithFlags(&c->ev[i][1], scope) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev[i][2], scope) != hipSuccess) {
      wbx_destroy(c);
      return WBX_ERR_DEVICE;
    }
  }
  {
    // The sequencer runs beside the mix of the previous render.  It is a few dozen latency-bound waves (one
    // lane per track): at low or equal priority they starve behind the 16 mix waves of their CU and the
    // plan becomes the critical path, so the plan stream gets the HIGHEST priority — the issue slots it
    // takes from the bandwidth-bound mix are negligible.  WBX_OVERLAP=0 runs everything on the main stream;
    // WBX_PLAN_PRIO=lo|hi picks the priority (tuning knobs).
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const char* ov = std::getenv("WBX_OVERLAP");
    c->overlap = !(ov && ov[0] == '0');
    const char* pp = std::getenv("WBX_PLAN_PRIO");
    const int prio = (pp && pp[0] == 'l') ? lo : hi;
    if (hipStreamCreateWithPriority(&c->plan_stream, hipStreamNonBlocking, prio) != hipSuccess) {
      wbx_destroy(c);
      return WBX_ERR_DEVICE;
    }
    // the sum stream: highest priority as well — its few hundred small workgroups start while the next mix floods
    // the device.  WBX_SUM_OVERLAP=0 keeps the sum on the main stream.
    const char* so = std::getenv("WBX_SUM_OVERLAP");
    c->sum_overlap = !(so && so[0] == '0');
    const char* sp = std::getenv("WBX_SUM_PRIO");
    const unsigned dev_ev = c->dev_event_flags;
    bool ok = hipStreamCreateWithPriority(&c->sum_stream, hipStreamNonBlocking, (sp && sp[0] == 'l') ? lo : hi) == hipSuccess;
    for (int i = 0; i < kRing && ok; i++)
      ok = hipEventCreateWithFlags(&c->mix_done[i], dev_ev) == hipSuccess &&
           hipEventCreateWithFlags(&c->sum_done[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&c->partial_free[i], dev_ev) == hipSuccess;
    if (ok) ok = hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) == hipSuccess;
    if (ok) ok = hipStreamCreateWithFlags(&c->alt_stream, hipStreamNonBlocking) == hipSuccess;
    // measured (tools/ab_alt.sh): with consecutive mixes on alternating streams the two kernels share the device for
    // their whole length (each takes 1.05-1.2 ms instead of 0.74) and the step time does not move — off by default
    if (const char* sb = std::getenv("WBX_CB_SPIN_BOUND")) c->cb_spin_bound = (uint32_t)std::atoi(sb);   // (tests: 0 forces the give-up path)
    {   // the A/B switches the render path consults (wbx_ctx.h: read once, here)
      auto is = [](const char* name, char what) { const char* v = std::getenv(name); return v && v[0] == what; };
      c->knob_ragged_off = is("WBX_RAGGED", '0');
      c->knob_cb_any_off = is("WBX_CB_ANY", '0');
      c->knob_no_uniform = is("WBX_NO_UNIFORM", '1');
      c->knob_masked_rows_off = is("WBX_MASKED_ROWS", '0');
      c->knob_chain_off = is("WBX_CHAIN", '0');
      c->knob_no_lean16 = std::getenv("WBX_NO_LEAN16") != nullptr;
      c->knob_no_fam3 = std::getenv("WBX_NO_FAM3") != nullptr;
      c->knob_no_cl2 = std::getenv("WBX_NO_CL2") != nullptr;
      c->knob_cb_fenced = is("WBX_CB_FENCED", '1');
      c->knob_partial_free_off = is("WBX_PARTIAL_FREE", '0');
      c->knob_mix_marker = is("WBX_MIX_MARKER", '1');
      c->knob_fast_partial_off = is("WBX_FAST_PARTIAL", '0');
      if (const char* v = std::getenv("WBX_PACKED_X")) c->knob_packed_x = std::atoi(v) != 0 ? 1 : 0;
    }
    // the workgroup-id -> XCD layout the chained pieces and the segmented sequencer rest on, probed before anything relies on it
    // (WBX_XCD_PROBE_FAIL=1: tests take the fallback path)
    if (ok) {
      const char* pf = std::getenv("WBX_XCD_PROBE_FAIL");
      if (!probe_xcd_layout(c->stream, &c->n_xcds) || (pf && pf[0] == '1')) {
        c->n_xcds = 0;
        c->chain_broken = true;
        c->seg_broken = true;
      }
    }
    const char* ma = std::getenv("WBX_MIX_ALT");
    c->mix_alternate = ma && ma[0] == '1';
    if (!ok) {
      wbx_destroy(c);
      return WBX_ERR_DEVICE;
    }
  }
  size_t chunks = cfg->max_segments ? (cfg->max_segments + kChunk - 1) / kChunk
                                    : std::max<size_t>(1024, (size_t)cfg->max_blocks * cfg->max_tracks / 8);
  for (auto& B : c->pb) {
    B.pool_chunks = (uint32_t)chunks;
    if (B.pool.ensure(chunks * kChunk) != hipSuccess || hipMalloc((void**)&B.counters, 4 * sizeof(uint32_t)) != hipSuccess ||
        hipEventCreateWithFlags(&B.planned, c->dev_event_flags) != hipSuccess) {
      wbx_destroy(c);
      return WBX_ERR_OOM;
    }
    (void)hipMemset(B.counters, 0, 4 * sizeof(uint32_t));
  }
  if (c->d_zero.ensure(cfg->block_frames + 8) != hipSuccess || hipMalloc((void**)&c->d_sticky_status, sizeof(uint32_t)) != hipSuccess) {
    wbx_destroy(c);
    return WBX_ERR_OOM;
  }
  (void)hipMemset(c->d_sticky_status, 0, sizeof(uint32_t));
  (void)hipMemset(c->d_zero.p, 0, (cfg->block_frames + 8) * sizeof(float));
  *out = c;
  return WBX_OK;
}

extern "C" void wbx_destroy(wbx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  dist_destroy(c);
  if (c->plan_stream) (void)hipStreamSynchronize(c->plan_stream);
  if (c->sum_stream) (void)hipStreamSynchronize(c->sum_stream);
  if (c->alt_stream) (void)hipStreamSynchronize(c->alt_stream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& s : c->clips) {
    if (s.alloc) (void)hipFree(s.alloc);
    if (s.mip) (void)hipFree(s.mip);
  }
  for (auto& sl : c->slabs)
    if (sl->mem) (void)hipFree(sl->mem);
  c->d_samples.release();
  c->d_order.release();
  c->d_groups.release();
  for (auto& B : c->pb) {
    B.prows.release();
    B.tmpl.release();
    B.pool.release();
    B.times.release();
    B.gen_list.release();
    B.rows.release();
    B.saved.release();
    if (B.counters) (void)hipFree(B.counters);
    if (B.planned) (void)hipEventDestroy(B.planned);
  }
  if (c->plan_stream) (void)hipStreamDestroy(c->plan_stream);
  if (c->sum_stream) (void)hipStreamDestroy(c->sum_stream);
  if (c->upload_stream) (void)hipStreamDestroy(c->upload_stream);
  if (c->alt_stream) (void)hipStreamDestroy(c->alt_stream);
  if (c->ready_ev) (void)hipEventDestroy(c->ready_ev);
  for (auto& ev : c->pace_ev)
    if (ev) (void)hipEventDestroy(ev);
  for (int i = 0; i < kRing; i++) {
    if (c->mix_done[i]) (void)hipEventDestroy(c->mix_done[i]);
    if (c->sum_done[i]) (void)hipEventDestroy(c->sum_done[i]);
    if (c->partial_free[i]) (void)hipEventDestroy(c->partial_free[i]);
  }
  for (auto& P : c->d_partial2) P.release();
  c->d_master.release();
  for (auto& st : c->d_stage) st.release();
  c->d_buses.release();
  for (auto& P : c->d_peaks) P.release();
  c->d_gains.release();
  c->d_conv.release();
  c->d_chain.release();
  if (c->d_sticky_status) (void)hipFree(c->d_sticky_status);
  if (c->d_cb_done) (void)hipFree(c->d_cb_done);
  c->d_dbg.release();
  c->d_zero.release();
  for (int i = 0; i < kEventRing; i++) {
    if (c->ev[i][0]) (void)hipEventDestroy(c->ev[i][0]);
    if (c->ev[i][1]) (void)hipEventDestroy(c->ev[i][1]);
    if (c->ev[i][2]) (void)hipEventDestroy(c->ev[i][2]);
  }
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char* wbx_last_error(const wbx_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

namespace wbx {

void clip_release(wbx_ctx* c, ClipSlot& s) {
  if (s.alloc) (void)hipFree(s.alloc);
  if (s.slab) {
    std::lock_guard<std::mutex> g(c->slab_mu);
    ClipSlab& sl = *s.slab;
    sl.live_bytes -= std::min(sl.live_bytes, s.slab_len);
    if (sl.live && --sl.live == 0) {   // the whole slab is free again
      sl.used = 0;
      sl.holes.clear();
    } else if (s.slab_off + s.slab_len == sl.used) {   // the newest extent: the bump pointer steps back (over a hole that ends there, too)
      sl.used = s.slab_off;
      if (!sl.holes.empty() && sl.holes.back().first + sl.holes.back().second == sl.used) {
        sl.used = sl.holes.back().first;
        sl.holes.pop_back();
      }
    } else {                           // a hole, merged with its neighbours
      auto it = std::lower_bound(sl.holes.begin(), sl.holes.end(), std::make_pair(s.slab_off, (size_t)0));
      it = sl.holes.insert(it, std::make_pair(s.slab_off, s.slab_len));
      if (it + 1 != sl.holes.end() && it->first + it->second == (it + 1)->first) {
        it->second += (it + 1)->second;
        it = sl.holes.erase(it + 1) - 1;
      }
      if (it != sl.holes.begin() && (it - 1)->first + (it - 1)->second == it->first) {
        (it - 1)->second += it->second;
        sl.holes.erase(it);
      }
    }
  }
  if (s.mip) (void)hipFree(s.mip);
  s = ClipSlot{};
}

// Clip audio -> a fresh slot `s` (not yet visible to any render): allocation, the 16 zero frames of tail padding
// (Sample::sample_padding, sample.cpp:127,140) and the fill, all enqueued on `on`.  Touches nothing of the ctx but
// its error string, so layer 2 can run it outside the editor lock on the upload stream.
wbx_status clip_build(wbx_ctx* c, ClipSlot& s, int format, uint32_t channels, uint32_t sample_rate, uint64_t frames,
                      const ClipFill& f, hipStream_t on) {
  const size_t eb = fmt_bytes(format);
  if (!eb) return fail(c, WBX_ERR_UNSUPPORTED, "clip format");
  if (channels < 1 || channels > 2) return fail(c, WBX_ERR_UNSUPPORTED, "clip channel count (1 or 2)");
  if (frames >= 2147483632ull) return fail(c, WBX_ERR_UNSUPPORTED, "clip longer than 2^31-16 frames");
  if (f.kind == CLIP_SRC_PLANAR && !f.planar) return WBX_ERR_INVALID;
  if ((f.kind == CLIP_SRC_INTERLEAVED_HOST || f.kind == CLIP_SRC_INTERLEAVED_DEVICE) && !f.interleaved && frames) return WBX_ERR_INVALID;
  if (f.kind == CLIP_SRC_INTERLEAVED_DEVICE && ((uintptr_t)f.interleaved & 15u))
    return fail(c, WBX_ERR_INVALID, "interleaved device buffer must be 16-byte aligned");
  (void)hipSetDevice(c->cfg.device);
  s = ClipSlot{};
  const size_t stride = align_up((frames + kPad) * eb, 256);   // (varying the distance between a clip's channel rows: no effect)
  {
    static const bool use_slabs = !(std::getenv("WBX_CLIP_ARENA") && std::getenv("WBX_CLIP_ARENA")[0] == '0');   // A/B aid
    constexpr size_t kSlab = (size_t)1 << 30, kGranule = (size_t)64 << 10;   // (8-GiB slabs, 2-MiB granules: no difference)
    // A pseudo-random gap of 0..15 granules in front of every clip (at most an eighth of the clip): a session of equally
    // long clips has one clip-to-clip stride, and some strides alias in the HBM address hash — the workgroups in flight
    // read the same offset of many clips at once (c4 with 9.06-MiB clips: 0.82 instead of 0.73 ms per launch;
    // tools/ab_arena.sh).  WBX_SLAB_JITTER=0: A/B aid.
    static const bool jitter = !(std::getenv("WBX_SLAB_JITTER") && std::getenv("WBX_SLAB_JITTER")[0] == '0');
    const size_t body = align_up(stride * channels, kGranule);
    const uint32_t span = (uint32_t)std::min<size_t>(16, body / kGranule / 8 + 1);
    const size_t gap = jitter ? (size_t)((((c->slab_seq.fetch_add(1u, std::memory_order_relaxed) + 1u) * 2654435761u) >> 8) % span) * kGranule : 0;
    const size_t need = body + gap;
    if (use_slabs && need <= kSlab / 4) {   // (slab sizes grow 64 MiB, 256 MiB, 1 GiB, 1 GiB ...: small sessions stay small)
      std::lock_guard<std::mutex> g(c->slab_mu);
      ClipSlab* sl = nullptr;
      size_t at = 0;
      bool in_hole = false;
      for (auto it = c->slabs.rbegin(); it != c->slabs.rend() && !sl; ++it) {   // the newest slab first: a hole that fits, else its tail
        for (auto h = (*it)->holes.begin(); h != (*it)->holes.end() && !sl; ++h)
          if (h->second >= need) {
            sl = it->get();
            at = h->first;
            in_hole = true;
            if (h->second == need) {
              sl->holes.erase(h);
            } else {
              h->first += need;
              h->second -= need;
            }
          }
        if (!sl && (*it)->size - (*it)->used >= need) sl = it->get();
      }
      if (!sl) {
        std::unique_ptr<ClipSlab> fresh(new (std::nothrow) ClipSlab());
        if (!fresh) return WBX_ERR_OOM;
        const size_t grown = c->slabs.size() >= 2 ? kSlab : ((size_t)64 << 20) << (2 * c->slabs.size());
        const size_t sz = std::max(grown, need);
        if (hipMalloc((void**)&fresh->mem, sz) == hipSuccess) {
          fresh->size = sz;
          c->slabs.push_back(std::move(fresh));
          sl = c->slabs.back().get();
        } else {
          (void)hipGetLastError();   // a device too full for another slab: this clip gets an allocation of its own below
        }
      }
      if (sl) {
        if (!in_hole) {
          at = sl->used;
          sl->used += need;
        }
        s.slab = sl;
        s.slab_off = at;
        s.slab_len = need;
        s.base = sl->mem + at + gap;
        sl->live++;
        sl->live_bytes += need;
      }
    }
    if (!s.slab) {
      WBX_HIP(c, hipMalloc(&s.alloc, stride * channels));
      s.base = s.alloc;
    }
  }
  s.d.ch[0] = s.base;
  s.d.ch[1] = channels > 1 ? (const void*)((const char*)s.base + stride) : s.base;   // mono wraps (i % channels)
  s.d.count = frames;
  s.d.format = (uint32_t)format;
  s.d.channels = channels;
  s.d.sample_rate = sample_rate;
  s.used = true;
  s.stride = stride;
  hipError_t err = hipSuccess;
  void* stage[2] = {nullptr, nullptr};
  hipEvent_t freed[2] = {nullptr, nullptr};
  auto pad_tails = [&]() {   // the padding frames (and the alignment slack) read as zero
    for (uint32_t ch = 0; ch < channels && err == hipSuccess; ch++)
      err = hipMemsetAsync((char*)s.base + stride * ch + frames * eb, 0, stride - frames * eb, on);
  };
  switch (f.kind) {
    case CLIP_SRC_PLANAR:
      pad_tails();
      for (uint32_t ch = 0; ch < channels && err == hipSuccess; ch++)
        err = hipMemcpyAsync((char*)s.base + stride * ch, f.planar[ch], frames * eb, hipMemcpyHostToDevice, on);
      if (err == hipSuccess) err = hipStreamSynchronize(on);   // the caller's arrays may go away
      break;
    case CLIP_SRC_SYNTH:
      for (uint32_t ch = 0; ch < channels; ch++) {
        const uint64_t key = f.seed ^ ((uint64_t)f.key_track << 40) ^ ((uint64_t)ch << 32);
        launch_synth((char*)s.base + stride * ch, frames, key, f.amp, format, on);   // writes the padding as well
      }
      err = hipGetLastError();
      break;
    case CLIP_SRC_INTERLEAVED_DEVICE:
      pad_tails();
      if (err == hipSuccess) {
        launch_deinterleave(f.interleaved, s.base, (char*)s.base + (channels > 1 ? stride : 0), frames, channels, (uint32_t)eb, on);
        err = hipGetLastError();
      }
      break;
    default: {   // CLIP_SRC_INTERLEAVED_HOST
      pad_tails();
      // chunks of the decoder's interleaved output go host -> device staging -> transposed into the channel rows;
      // two staging buffers so the copy of chunk i+1 overlaps the transposition of chunk i
      const uint64_t chunk_frames = (uint64_t)4 << 20;   // a multiple of 4 frames: lane groups never straddle chunks
      const size_t chunk_bytes = (size_t)chunk_frames * channels * eb;
      const int nstage = frames > chunk_frames ? 2 : 1;
      for (int i = 0; i < nstage && err == hipSuccess; i++) {
        err = hipMalloc(&stage[i], std::min<size_t>(chunk_bytes, std::max<size_t>(16, (size_t)frames * channels * eb)));
        if (err == hipSuccess) err = hipEventCreateWithFlags(&freed[i], hipEventDisableTiming);
      }
      uint64_t done = 0;
      for (int i = 0; done < frames && err == hipSuccess; i++) {
        const uint64_t n = std::min<uint64_t>(chunk_frames, frames - done);
        const int b = i % nstage;
        if (i >= nstage) err = hipEventSynchronize(freed[b]);
        if (err == hipSuccess)
          err = hipMemcpyAsync(stage[b], (const char*)f.interleaved + (size_t)done * channels * eb, (size_t)n * channels * eb,
                               hipMemcpyHostToDevice, on);
        if (err == hipSuccess) {
          launch_deinterleave(stage[b], (char*)s.base + (size_t)done * eb,
                              (char*)s.base + (channels > 1 ? stride : 0) + (size_t)done * eb, n, channels, (uint32_t)eb, on);
          err = hipEventRecord(freed[b], on);
        }
        done += n;
      }
      if (err == hipSuccess) err = hipStreamSynchronize(on);
      else (void)hipStreamSynchronize(on);   // nothing may still read the staging buffers freed below
      for (int i = 0; i < 2; i++) {
        if (stage[i]) (void)hipFree(stage[i]);
        if (freed[i]) (void)hipEventDestroy(freed[i]);
      }
      break;
    }
  }
  if (err != hipSuccess) {
    (void)hipStreamSynchronize(on);
    clip_release(c, s);
    return fail(c, WBX_ERR_DEVICE, "clip upload", err);
  }
  return WBX_OK;
}

// make slot `clip` of the pool hold `s` (an earlier clip of that id is freed once the device is done with it)
wbx_status clip_publish(wbx_ctx* c, uint32_t clip, ClipSlot& s) {
  if (clip >= (1u << 24)) {
    clip_release(c, s);
    return fail(c, WBX_ERR_INVALID, "clip id");
  }
  if (clip >= c->clips.size()) c->clips.resize(clip + 1);
  ClipSlot& dst = c->clips[clip];
  if (dst.base) {
    (void)hipStreamSynchronize(c->plan_stream);
    (void)join_sum(c);
    (void)sync_main(c);
    clip_release(c, dst);
  }
  dst = std::move(s);
  s = ClipSlot{};
  c->samples_dirty = true;
  return WBX_OK;
}

}  // namespace wbx

static wbx_status clip_create(wbx_ctx* c, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate, uint64_t frames,
                              const ClipFill& f) {
  if (!c) return WBX_ERR_INVALID;
  if (clip >= (1u << 24)) return fail(c, WBX_ERR_INVALID, "clip id");
  ClipSlot s;
  wbx_status st = clip_build(c, s, format, channels, sample_rate, frames, f, c->stream);
  if (st != WBX_OK) return st;
  return clip_publish(c, clip, s);
}

extern "C" wbx_status wbx_clip_upload(wbx_ctx* c, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                                      uint64_t frames, const void* const* planar) {
  if (!c || !planar) return WBX_ERR_INVALID;
  ClipFill f{};
  f.kind = CLIP_SRC_PLANAR;
  f.planar = planar;
  return clip_create(c, clip, format, channels, sample_rate, frames, f);
}

extern "C" wbx_status wbx_clip_synth(wbx_ctx* c, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                                     uint64_t frames, uint64_t seed, uint32_t key_track, float amp) {
  ClipFill f{};
  f.kind = CLIP_SRC_SYNTH;
  f.seed = seed;
  f.key_track = key_track;
  f.amp = amp;
  return clip_create(c, clip, format, channels, sample_rate, frames, f);
}

// ---- clip ingest (dsp/sample.cpp:29-43, :112-197) ------------------------------------------------
extern "C" wbx_status wbx_clip_ingest_device(wbx_ctx* c, uint32_t clip, int format, uint32_t channels, uint32_t sample_rate,
                                             uint64_t frames, const void* device_interleaved) {
  ClipFill f{};
  f.kind = CLIP_SRC_INTERLEAVED_DEVICE;
  f.interleaved = device_interleaved;
  return clip_create(c, clip, format, channels, sample_rate, frames, f);
}

extern "C" wbx_status wbx_clip_upload_interleaved(wbx_ctx* c, uint32_t clip, int format, uint32_t channels,
                                                  uint32_t sample_rate, uint64_t frames, const void* interleaved) {
  ClipFill f{};
  f.kind = CLIP_SRC_INTERLEAVED_HOST;
  f.interleaved = interleaved;
  return clip_create(c, clip, format, channels, sample_rate, frames, f);
}

extern "C" wbx_status wbx_clip_download(wbx_ctx* c, uint32_t clip, uint32_t channel, void* dst) {
  if (!c || !dst || clip >= c->clips.size() || !c->clips[clip].used) return WBX_ERR_INVALID;
  const ClipSlot& s = c->clips[clip];
  if (channel >= s.d.channels) return fail(c, WBX_ERR_INVALID, "channel out of range");
  WBX_HIP(c, sync_main(c));
  WBX_HIP(c, hipMemcpy(dst, (const char*)s.base + s.stride * channel, (size_t)s.d.count * fmt_bytes((int)s.d.format), hipMemcpyDeviceToHost));
  return WBX_OK;
}

// ---- waveform mip-maps (gfx/waveform_visual.cpp:9-246) -------------------------------------------
extern "C" uint32_t wbx_mip_levels(uint64_t frames) {
  uint32_t n = 0;
  for (uint64_t sample_count = frames; sample_count > 64; sample_count /= 4) n++;   // :195, :236
  return n;
}

extern "C" uint64_t wbx_mip_data_count(uint64_t frames, uint32_t level) {
  const uint64_t block_count = (uint64_t)1 << (2u * level);   // 2^(current_mip - 1), current_mip = 1 + 2*level
  uint64_t n = frames / block_count;                           // :198
  n += n % 2;                                                  // :199
  return n;
}

extern "C" wbx_status wbx_clip_build_mipmaps(wbx_ctx* c, uint32_t clip, int quality) {
  if (!c || clip >= c->clips.size() || !c->clips[clip].used) return WBX_ERR_INVALID;
  if (quality != 0 && quality != 1) return fail(c, WBX_ERR_INVALID, "quality: 0 (Low, int8) or 1 (High, int16)");
  ClipSlot& s = c->clips[clip];
  const int fmt = (int)s.d.format;
  if (fmt != FMT_I16 && fmt != FMT_I32 && fmt != FMT_F32) return fail(c, WBX_ERR_UNSUPPORTED, "mip-maps: clip format");
  (void)hipSetDevice(c->cfg.device);
  const uint32_t levels = wbx_mip_levels(s.d.count);
  if (levels > 24) return fail(c, WBX_ERR_UNSUPPORTED, "mip-maps: too many levels");
  const int bits = quality ? 16 : 8;
  const size_t esz = bits / 8;
  if (s.mip) {
    WBX_HIP(c, sync_main(c));
    (void)hipFree(s.mip);
    s.mip = nullptr;
  }
  s.mip_off.assign(levels, 0);
  s.mip_count.assign(levels, 0);
  s.mip_bits = bits;
  size_t total = 0;
  for (uint32_t l = 0; l < levels; l++) {
    s.mip_count[l] = wbx_mip_data_count(s.d.count, l);
    s.mip_off[l] = total;
    total += align_up((size_t)s.mip_count[l] * s.d.channels * esz, 256);
  }
  if (!levels) return WBX_OK;
  const uint32_t tiles = (uint32_t)((s.d.count + kMipTile - 1) / kMipTile);
  const size_t nodes_off = total;
  const size_t nodes_per_ch = (size_t)tiles + tiles / 4 + 1;   // tile nodes + ping-pong scratch of the upper levels
  total += nodes_per_ch * s.d.channels * sizeof(MipNode);
  WBX_HIP(c, hipMalloc(&s.mip, total));
  for (uint32_t ch = 0; ch < s.d.channels; ch++) {
    MipArgs a{};
    a.src = (const char*)s.base + s.stride * ch;
    a.count = s.d.count;
    a.n_levels = levels;
    a.n_tiles = tiles;
    a.tile_nodes = (MipNode*)((char*)s.mip + nodes_off) + nodes_per_ch * ch;
    for (uint32_t l = 0; l < levels; l++) {
      a.level_out[l] = (char*)s.mip + s.mip_off[l] + (size_t)s.mip_count[l] * ch * esz;
      a.data_count[l] = s.mip_count[l];
    }
    launch_mip(a, fmt, bits, c->stream);
  }
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_mipmap_device(wbx_ctx* c, uint32_t clip, uint32_t level, const void** data, uint64_t* count) {
  if (!c || clip >= c->clips.size() || !c->clips[clip].used) return WBX_ERR_INVALID;
  const ClipSlot& s = c->clips[clip];
  if (!s.mip_bits || level >= s.mip_off.size()) return fail(c, WBX_ERR_INVALID, "no such mip level (call wbx_clip_build_mipmaps)");
  if (data) *data = (const char*)s.mip + s.mip_off[level];
  if (count) *count = s.mip_count[level];
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_fetch_mipmap(wbx_ctx* c, uint32_t clip, uint32_t level, void* dst) {
  const void* p = nullptr;
  uint64_t n = 0;
  wbx_status st = wbx_clip_mipmap_device(c, clip, level, &p, &n);
  if (st != WBX_OK) return st;
  if (!dst) return WBX_ERR_INVALID;
  const ClipSlot& s = c->clips[clip];
  WBX_HIP(c, sync_main(c));
  WBX_HIP(c, hipMemcpy(dst, p, (size_t)n * s.d.channels * (s.mip_bits / 8), hipMemcpyDeviceToHost));
  return WBX_OK;
}

extern "C" wbx_status wbx_clip_free(wbx_ctx* c, uint32_t clip) {
  if (!c || clip >= c->clips.size() || !c->clips[clip].base) return WBX_ERR_INVALID;
  // layer 2: a clip list that still names the sample would make the sequencer hand the mix a dangling pointer
  if (c->sample_in_use && c->sample_in_use(c->owner, clip))
    return fail(c, WBX_ERR_INVALID, "sample is still referenced by a clip (delete the clips first)");
  (void)hipSetDevice(c->cfg.device);
  WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, sync_main(c));
  clip_release(c, c->clips[clip]);
  c->samples_dirty = true;
  return WBX_OK;
}

// clip storage in numbers: slabs allocated, bytes reserved from the driver (slabs + clips with an allocation of their
// own), bytes held by live clips
extern "C" wbx_status wbx_clip_pool_stats(wbx_ctx* c, uint32_t* n_slabs, uint64_t* bytes_reserved, uint64_t* bytes_live) {
  if (!c) return WBX_ERR_INVALID;
  std::lock_guard<std::mutex> g(c->slab_mu);
  uint64_t reserved = 0, live = 0;
  for (auto& sl : c->slabs) {
    reserved += sl->size;
    live += sl->live_bytes;
  }
  for (auto& s : c->clips)
    if (s.alloc) {
      reserved += s.stride * s.d.channels;
      live += s.stride * s.d.channels;
    }
  if (n_slabs) *n_slabs = (uint32_t)c->slabs.size();
  if (bytes_reserved) *bytes_reserved = reserved;
  if (bytes_live) *bytes_live = live;
  return WBX_OK;
}

extern "C" wbx_status wbx_set_routing(wbx_ctx* c, uint32_t n_tracks, const int32_t* track_bus, uint32_t n_buses) {
  if (!c || n_tracks > c->cfg.max_tracks) return WBX_ERR_INVALID;
  if (track_bus && n_buses) {
    c->track_bus.assign(track_bus, track_bus + n_tracks);
    c->n_buses = n_buses;
  } else {
    c->track_bus.clear();
    c->n_buses = 0;
  }
  build_routing(c, n_tracks);
  c->buses_clean = false;
  return WBX_OK;
}

// how a render of n_blocks blocks is summed, with the routing as it stands (the last render's, or wbx_set_routing's)
extern "C" wbx_status wbx_render_order(wbx_ctx* c, uint32_t n_blocks, uint32_t* n_groups, uint32_t* longest_group,
                                       int* reference_order) {
  if (!c || n_blocks == 0) return WBX_ERR_INVALID;
  const bool chained = render_chains_groups(c, n_blocks);
  const bool whole = render_walks_whole_lists(c, n_blocks) && !chained;
  const std::vector<DGroup>& gs = whole ? c->groups_exact : c->groups;
  uint32_t longest = 0;
  for (auto& g : gs) longest = std::max(longest, g.count);
  if (n_groups) *n_groups = (uint32_t)gs.size();
  if (longest_group) *longest_group = longest;
  // (chained: the pieces are workgroup-sized, the additions are one sequence per member list all the same)
  if (reference_order) *reference_order = (whole || chained || gs.size() == c->groups_exact.size()) ? 1 : 0;
  return WBX_OK;
}

extern "C" wbx_status wbx_device_info(wbx_ctx* c, char* pci_bus_id, size_t n_pci, char* name, size_t n_name) {
  if (!c) return WBX_ERR_INVALID;
  if (pci_bus_id && n_pci) {
    pci_bus_id[0] = 0;
    WBX_HIP(c, hipDeviceGetPCIBusId(pci_bus_id, (int)n_pci, c->cfg.device));
  }
  if (name && n_name) {
    hipDeviceProp_t p;
    WBX_HIP(c, hipGetDeviceProperties(&p, c->cfg.device));
    std::snprintf(name, n_name, "%s (%s)", p.name, p.gcnArchName);
  }
  return WBX_OK;
}

// diagnostic (not in wbx.h; WBX_DBG_CLOCK=1): start / end wall-clock ticks (100 MHz) of every workgroup of the last mix
extern "C" wbx_status wbx_debug_wg_clocks(wbx_ctx* c, unsigned long long* out, size_t cap, size_t* n_wgs) {
  if (!c || !n_wgs) return WBX_ERR_INVALID;
  *n_wgs = c->dbg_wgs;
  if (!out || !c->d_dbg.p) return WBX_OK;
  WBX_HIP(c, sync_main(c));
  WBX_HIP(c, hipMemcpy(out, c->d_dbg.p, std::min(cap, 4 * c->dbg_wgs) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return WBX_OK;
}

extern "C" wbx_status wbx_set_clamp(wbx_ctx* c, int on) {
  if (!c) return WBX_ERR_INVALID;
  c->clamp = on != 0;
  return WBX_OK;
}

static size_t out_format_bytes(int fmt) {
  switch (fmt) {
    case WBX_OUT_I16: return 2;
    case WBX_OUT_I24: return 3;
    case WBX_OUT_I24_X8:
    case WBX_OUT_I32:
    case WBX_OUT_F32: return 4;
    default: return 0;
  }
}

extern "C" wbx_status wbx_set_master_format(wbx_ctx* c, int out_format) {
  if (!c) return WBX_ERR_INVALID;
  if (out_format != 0 && out_format_bytes(out_format) == 0) return fail(c, WBX_ERR_UNSUPPORTED, "interleaved output format");
  c->master_format = out_format;
  return WBX_OK;
}

extern "C" wbx_status wbx_set_master_init(wbx_ctx* c, const void* device_buffer) {
  if (!c) return WBX_ERR_INVALID;
  if ((uintptr_t)device_buffer & 15u) return fail(c, WBX_ERR_INVALID, "wbx_set_master_init: the buffer must be 16-byte aligned");
  c->master_init = (const float*)device_buffer;
  return WBX_OK;
}

extern "C" wbx_status wbx_set_master_target(wbx_ctx* c, void* device_buffer) {
  if (!c) return WBX_ERR_INVALID;
  c->master_target = (float*)device_buffer;
  c->master_target_on_host = false;
  if (device_buffer) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, device_buffer) == hipSuccess) c->master_target_on_host = at.type == hipMemoryTypeHost;
    else (void)hipGetLastError();
  }
  return WBX_OK;
}

extern "C" wbx_status wbx_submit(wbx_ctx* c, uint32_t K, uint32_t N, const wbx_segment* segs, const uint32_t* seg_offsets,
                                 const float* gains) {
  if (!c || !seg_offsets || !gains || K == 0 || N == 0) return WBX_ERR_INVALID;
  if (K > c->cfg.max_blocks || N > c->cfg.max_tracks) return fail(c, WBX_ERR_INVALID, "K or N above the configured maximum");
  (void)hipSetDevice(c->cfg.device);
  WBX_HIP(c, hipStreamSynchronize(c->plan_stream));
  wbx_status st = upload_tables(c, N);
  if (st != WBX_OK) return st;
  st = ensure_result_buffers(c, K, N);
  if (st != WBX_OK) return st;
  WBX_HIP(c, sync_main(c));   // staging vectors are reused
  const uint32_t F = c->cfg.block_frames, C = c->cfg.channels;
  c->h_tb.assign((size_t)K * N, DTrackBlock{});
  c->h_rows.assign((size_t)K * N, DRow{});
  c->h_pool.clear();
  std::vector<uint32_t> gen_idx;
  uint32_t chunks = 0;
  for (size_t bt = 0; bt < (size_t)K * N; bt++) {
    DTrackBlock& tb = c->h_tb[bt];
    tb.g[0] = gains[bt * 2 + 0];
    tb.g[1] = gains[bt * 2 + (C > 1 ? 1 : 0)];
    const uint32_t s0 = seg_offsets[bt], s1 = seg_offsets[bt + 1];
    if (s1 < s0) return fail(c, WBX_ERR_INVALID, "seg_offsets not monotone");
    if (s1 - s0 > kMaxSegs) return fail(c, WBX_ERR_OVERFLOW, "more than 16 segments in one track-block");
    if (s1 != s0 && !segs) return WBX_ERR_INVALID;
    for (uint32_t i = s0; i < s1; i++) {
      const wbx_segment& sg = segs[i];
      if (sg.clip >= c->clips.size() || !c->clips[sg.clip].used) return fail(c, WBX_ERR_INVALID, "segment names an unknown clip");
      // the ABI is the trust boundary: a negative / NaN position or a non-positive / NaN speed would index outside the clip
      if (!(sg.sample_offset >= 0.0) || !(sg.playback_speed > 0.0) || !(sg.playback_speed < 1e12))
        return fail(c, WBX_ERR_INVALID, "segment needs sample_offset >= 0 and 0 < playback_speed < 1e12");
      // the prologue of Sampler::stream (sampler.cpp:99-104) through the shared walker
      DTrackState ts{};
      ts.cur_type = EV_PLAY;
      ts.cur_sample = 0;
      ts.cur_gain = sg.gain;
      ts.playback_speed = sg.playback_speed;
      ts.sample_offset = sg.sample_offset;
      DSeg d{};
      BlockWalker w{};
      DTrackBlock scratch{};
      TrackCache tc{};
      tc.clip_idx = tc.next_idx = tc.smp_idx = tc.fin_tmpl = 0xFFFFFFFFu;
      uint32_t pc = 0, stbits = 0;
      w.st = &ts;
      w.cache = &tc;
      w.samples = &c->clips[sg.clip].d;
      w.tb = &scratch;
      w.pool = &d;
      w.pool_count = &pc;
      w.pool_chunks = 0;
      w.status = &stbits;
      w.n_samples = F;
      w.n_channels = C;
      w.dst_rate = (double)c->cfg.sample_rate;
      w.start_sample = 0;
      w.nseg = 0;
      w.chunk = 0xFFFFFFFFu;
      w.stream(sg.num_samples, sg.buffer_offset);
      d = get_seg0(scratch);
      d.sample = sg.clip;
      const uint32_t k = i - s0;
      if (k == 0) {
        set_seg0(&tb, d);
      } else {
        if (k == 1) {
          tb.extra = chunks++;
          c->h_pool.resize((size_t)chunks * kChunk);
        }
        c->h_pool[(size_t)tb.extra * kChunk + (k - 1)] = d;
      }
    }
    tb.nseg = (uint8_t)(s1 - s0);
    tb.kind = classify(tb, F);
    if (tb.kind == KIND_GENERIC) gen_idx.push_back((uint32_t)bt);
    // host-sequenced plans use one template per track-block: row bt -> template bt, position inside the template
    DRow& row = c->h_rows[bt];
    row.pos = tb.pos;
    row.tmpl = tb.nseg ? (uint32_t)bt : 0xFFFFFFFFu;
    row.flags = tb.kind == KIND_SILENT ? ROW_SILENT : 0u;
  }
  st = ensure_gen_capacity(c, gen_idx.size());
  if (st != WBX_OK) return st;
  st = ensure_template_capacity(c, (size_t)K * N);
  if (st != WBX_OK) return st;
  {
    uint32_t counters[4] = {0u, 0u, (uint32_t)gen_idx.size(), (uint32_t)((size_t)K * N)};
    WBX_HIP(c, hipMemcpyAsync(PB(c).counters, counters, sizeof(counters), hipMemcpyHostToDevice, c->stream));
    if (!gen_idx.empty())
      WBX_HIP(c, hipMemcpyAsync(PB(c).gen_list.p, gen_idx.data(), gen_idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    WBX_HIP(c, sync_main(c));   // counters / gen_idx are stack / local storage
  }
  if (chunks > PB(c).pool_chunks) {
    WBX_HIP(c, PB(c).pool.ensure((size_t)chunks * kChunk));
    PB(c).pool_chunks = chunks;
  }
  WBX_HIP(c, hipMemcpyAsync(PB(c).tmpl.p, c->h_tb.data(), c->h_tb.size() * sizeof(DTrackBlock), hipMemcpyHostToDevice, c->stream));
  WBX_HIP(c, hipMemcpyAsync(PB(c).prows.p, c->h_rows.data(), c->h_rows.size() * sizeof(DRow), hipMemcpyHostToDevice, c->stream));
  if (!c->h_pool.empty())
    WBX_HIP(c, hipMemcpyAsync(PB(c).pool.p, c->h_pool.data(), c->h_pool.size() * sizeof(DSeg), hipMemcpyHostToDevice, c->stream));
  c->short_render_now = K < kOverlapMinBlocks;
  c->render_blocks_now = K;
  c->whole_lists_now = render_walks_whole_lists(c, K);
  c->chain_now = render_chains_groups(c, K);
  (void)pick_mix_stream(c, K, false);   // host-sequenced plans are uploaded on the main stream: their mix follows there
  c->masked_rows = 0u;   // host-sequenced plans send every partial row through the pre-render pass
  c->uniform_speed = 0.0;   // ... and make no promise about their playback speeds
  st = launch_pre_render(c, K, c->stream);
  if (st != WBX_OK) return st;
  return launch_mix_sum(c, K, N);
}

extern "C" wbx_status wbx_sync(wbx_ctx* c) {
  if (!c) return WBX_ERR_INVALID;
  WBX_HIP(c, sync_main(c));
  drain_events(c);
  return WBX_OK;
}

extern "C" wbx_status wbx_master_ready(wbx_ctx* c, void* stream) {
  if (!c) return WBX_ERR_INVALID;
  hipStream_t on = stream ? (hipStream_t)stream : c->stream;
  if (on == c->stream) {
    WBX_HIP(c, join_sum(c));
    return WBX_OK;
  }
  if (c->sum_pending >= 0) {   // the sum runs on its own stream: its event orders any stream
    WBX_HIP(c, hipStreamWaitEvent(on, c->sum_done[c->sum_pending], 0));
  } else {                     // everything is on the main stream: mark where it stands now
    if (!c->ready_ev) WBX_HIP(c, hipEventCreateWithFlags(&c->ready_ev, hipEventDisableTiming));
    WBX_HIP(c, hipEventRecord(c->ready_ev, c->stream));
    WBX_HIP(c, hipStreamWaitEvent(on, c->ready_ev, 0));
  }
  return WBX_OK;
}

wbx_status wbx::plan_status_to_error(wbx_ctx* c, uint32_t bits) {
  if (bits & 3u) return fail(c, WBX_ERR_OVERFLOW, "segment plan overflow (raise wbx_config.max_segments)");
  if (bits & 8u) return fail(c, WBX_ERR_OVERFLOW, "more boundary / non-fp32 track-blocks than pre-render rows");
  if (bits & 16u) return fail(c, WBX_ERR_OVERFLOW, "plan template array full");
  if (bits & 128u) {
    // the segmented sequencer planned a track again over rows its first versions of which were written behind another L2
    // (the workgroup-id -> XCD layout plan_seg_kernel rests on did not hold on this device): the rows of that render are in
    // doubt, it says so, and the engine plans by one lane per track from here on
    c->seg_broken = true;
    return fail(c, WBX_ERR_DEVICE, "the segmented sequencer's lanes of one track ran on more than one XCD and a seam was planned again: "
                                   "this render is invalid, later renders are planned by one lane per track");
  }
  if (bits & 96u) {
    // the chained hand-over rests on in-order workgroup dispatch and on a block's pieces sharing an XCD; a render that saw
    // either fail says so — its results are invalid — and the context walks whole lists from here on (same order of
    // additions, no hand-over between workgroups)
    c->chain_broken = true;
    return fail(c, WBX_ERR_DEVICE, (bits & 32u) ? "a chained workgroup gave up waiting for its predecessor's running sum: this render is invalid, "
                                                    "later renders walk whole lists"
                                                  : "a chained workgroup ran on another XCD than its predecessor: this render is invalid, later "
                                                    "renders walk whole lists");
  }
  return WBX_OK;
}

// Chained renders report a failed hand-over through a word of the context that outlives their plan buffers: read it (the
// streams are idle), clear it, turn it into the error — and into whole-list walks from then on.
wbx_status wbx::render_status(wbx_ctx* c) {
  if (!c->d_sticky_status || c->chain_epoch == 0u) return WBX_OK;   // (no chained render so far)
  uint32_t bits = 0;
  WBX_HIP(c, hipMemcpy(&bits, c->d_sticky_status, sizeof(bits), hipMemcpyDeviceToHost));
  if (!bits) return WBX_OK;
  WBX_HIP(c, hipMemset(c->d_sticky_status, 0, sizeof(uint32_t)));
  return plan_status_to_error(c, bits & 96u);
}

extern "C" wbx_status wbx_render_status(wbx_ctx* c) {
  if (!c) return WBX_ERR_INVALID;
  (void)hipSetDevice(c->cfg.device);
  WBX_HIP(c, sync_main(c));
  drain_events(c);
  return render_status(c);
}

extern "C" wbx_status wbx_fetch(wbx_ctx* c, float* const* master_planar, float* peaks, float* buses) {
  if (!c) return WBX_ERR_INVALID;
  if (c->last_K == 0) return fail(c, WBX_ERR_FAILED, "nothing submitted");
  WBX_HIP(c, join_sum(c));
  const uint32_t K = c->last_K, N = c->last_N, C = c->cfg.channels, F = c->cfg.block_frames;
  if (master_planar && c->last_master_format)
    return fail(c, WBX_ERR_INVALID, "the last render left its master in a device format (wbx_set_master_format): wbx_fetch_interleaved");
  if (master_planar) {
    // device [K][C][F] -> host planar[c][b*F + j]
    for (uint32_t ch = 0; ch < C; ch++)
      if (c->last_master_on_host) {   // Engine::process: the block is already on the host
        WBX_HIP(c, sync_main(c));
        for (uint32_t b = 0; b < K; b++)
          std::memcpy(master_planar[ch] + (size_t)b * F, c->last_master + ((size_t)b * C + ch) * F, F * sizeof(float));
      } else {
        WBX_HIP(c, hipMemcpy2DAsync(master_planar[ch], F * sizeof(float), c->last_master + (size_t)ch * F,
                                    (size_t)C * F * sizeof(float), F * sizeof(float), K, hipMemcpyDeviceToHost, c->stream));
      }
  }
  if (peaks) WBX_HIP(c, hipMemcpyAsync(peaks, c->last_peaks, (size_t)K * N * C * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (buses) {
    if (!c->n_buses) return fail(c, WBX_ERR_INVALID, "no buses configured");
    WBX_HIP(c, hipMemcpyAsync(buses, c->last_buses, (size_t)K * c->n_buses * C * F * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  WBX_HIP(c, sync_main(c));
  drain_events(c);
  uint32_t pc[4] = {0, 0, 0, 0};
  WBX_HIP(c, hipMemcpy(pc, PB(c).counters, sizeof(pc), hipMemcpyDeviceToHost));
  const wbx_status rs = render_status(c);   // (an earlier, unfetched render's failure counts too)
  const wbx_status ps = plan_status_to_error(c, pc[1]);
  return ps != WBX_OK ? ps : rs;
}

extern "C" wbx_status wbx_fetch_interleaved(wbx_ctx* c, int out_format, void* dst) {
  if (!c || !dst) return WBX_ERR_INVALID;
  if (c->last_K == 0) return fail(c, WBX_ERR_FAILED, "nothing submitted");
  WBX_HIP(c, join_sum(c));
  size_t eb;
  switch (out_format) {
    case WBX_OUT_I16: eb = 2; break;
    case WBX_OUT_I24: eb = 3; break;
    case WBX_OUT_I24_X8:
    case WBX_OUT_I32:
    case WBX_OUT_F32: eb = 4; break;
    default: return fail(c, WBX_ERR_UNSUPPORTED, "interleaved output format");
  }
  const uint32_t K = c->last_K, C = c->cfg.channels, F = c->cfg.block_frames;
  if (c->last_master_format) {   // the sum kernel already wrote this format (wbx_set_master_format): a copy
    if (c->last_master_format != out_format) return fail(c, WBX_ERR_INVALID, "the last render's master is in another device format");
    if (c->last_master_on_host) {
      WBX_HIP(c, sync_main(c));
      if (out_format == WBX_OUT_I24)
        for (uint32_t b = 0; b < K; b++) std::memcpy((char*)dst + (size_t)b * F * C * 3, (const char*)c->last_master + (size_t)b * F * C * 3, (size_t)F * 3);
      else
        std::memcpy(dst, c->last_master, (size_t)K * F * C * eb);
      return render_status(c);
    }
    if (out_format == WBX_OUT_I24)
      WBX_HIP(c, hipMemcpy2DAsync(dst, (size_t)F * C * 3, c->last_master, (size_t)F * C * 3, (size_t)F * 3, K, hipMemcpyDeviceToHost, c->stream));
    else
      WBX_HIP(c, hipMemcpyAsync(dst, c->last_master, (size_t)K * F * C * eb, hipMemcpyDeviceToHost, c->stream));
    WBX_HIP(c, sync_main(c));
    return render_status(c);
  }
  if (out_format == WBX_OUT_I24) {
    // convert_f32_to_interleaved_i24 (audio_format_conv.cpp:22-43) writes byte 3*i.. of EVERY channel's sample i — the
    // destination index has no channel term — so per converted block the last channel's packed samples fill bytes
    // [0, 3F) and the remaining 3F(C-1) bytes of the block's region are never touched: reproduced as written
    WBX_HIP(c, c->d_conv.ensure((size_t)K * F * 3));
    launch_convert(c->last_master, c->d_conv.p, K, F, C, out_format, c->stream);
    WBX_HIP(c, hipMemcpy2DAsync(dst, (size_t)F * C * 3, c->d_conv.p, (size_t)F * 3, (size_t)F * 3, K, hipMemcpyDeviceToHost, c->stream));
    WBX_HIP(c, sync_main(c));
    return render_status(c);
  }
  const size_t bytes = (size_t)K * F * C * eb;
  WBX_HIP(c, c->d_conv.ensure(bytes));
  launch_convert(c->last_master, c->d_conv.p, K, F, C, out_format, c->stream);   // pinned staging is device-readable too
  WBX_HIP(c, hipMemcpyAsync(dst, c->d_conv.p, bytes, hipMemcpyDeviceToHost, c->stream));
  WBX_HIP(c, sync_main(c));
  return render_status(c);
}

extern "C" wbx_status wbx_partial_master(wbx_ctx* c, void** device_ptr, size_t* n_floats) {
  if (!c || !device_ptr) return WBX_ERR_INVALID;
  if (c->last_K == 0) return fail(c, WBX_ERR_FAILED, "nothing submitted");
  WBX_HIP(c, join_sum(c));
  *device_ptr = c->last_master;
  if (n_floats) *n_floats = (size_t)c->last_K * c->cfg.channels * c->cfg.block_frames;
  return WBX_OK;
}

extern "C" wbx_status wbx_finalize_master(wbx_ctx* c, void* device_partial, uint32_t n_blocks, int clamp, void* stream) {
  if (!c || !device_partial || n_blocks == 0) return WBX_ERR_INVALID;
  hipStream_t on = stream ? (hipStream_t)stream : c->stream;
  if (c->sum_pending >= 0) WBX_HIP(c, hipStreamWaitEvent(on, c->sum_done[c->sum_pending], 0));
  if (clamp) launch_clamp((float*)device_partial, (size_t)n_blocks * c->cfg.channels * c->cfg.block_frames, on);
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

extern "C" wbx_status wbx_finalize_master_into(wbx_ctx* c, const void* device_partial, void* dst, uint32_t n_blocks, int clamp,
                                               void* stream) {
  if (!c || !device_partial || !dst || n_blocks == 0) return WBX_ERR_INVALID;
  if (((uintptr_t)device_partial | (uintptr_t)dst) & 15u) return fail(c, WBX_ERR_INVALID, "finalize: buffers must be 16-byte aligned");
  hipStream_t on = stream ? (hipStream_t)stream : c->stream;
  if (c->sum_pending >= 0) WBX_HIP(c, hipStreamWaitEvent(on, c->sum_done[c->sum_pending], 0));
  launch_clamp_into((const float*)device_partial, (float*)dst, (size_t)n_blocks * c->cfg.channels * c->cfg.block_frames, clamp, on);
  WBX_HIP(c, hipGetLastError());
  return WBX_OK;
}

// Bound how far the submitting thread runs ahead of the device: beyond a few dozen queued renders the HIP runtime
// blocks a launch until its queue has emptied and the device then idles.  Call after each submit / render: the host
// waits (only) until the render issued `max_ahead` calls ago has left the main stream.
extern "C" wbx_status wbx_pace(wbx_ctx* c, uint32_t max_ahead) {
  if (!c || max_ahead == 0 || max_ahead >= kPaceRing) return WBX_ERR_INVALID;
  (void)hipSetDevice(c->cfg.device);
  const uint32_t slot = (uint32_t)(c->pace_seq % kPaceRing);
  if (!c->pace_ev[slot]) WBX_HIP(c, hipEventCreateWithFlags(&c->pace_ev[slot], c->dev_event_flags));   // (the host waits for "done", reads nothing)
  // (behind the last render's sum when that runs on its own stream: no marker between two mixes; WBX_MIX_MARKER=1: on the mix stream)
  hipStream_t where = c->cur_mix_stream ? c->cur_mix_stream : c->stream;
  if (!c->knob_mix_marker && c->sum_pending >= 0 && c->sum_stream && !c->dist) where = c->sum_stream;
  WBX_HIP(c, hipEventRecord(c->pace_ev[slot], where));
  if (c->pace_seq >= max_ahead) WBX_HIP(c, hipEventSynchronize(c->pace_ev[(c->pace_seq - max_ahead) % kPaceRing]));
  c->pace_seq++;
  return WBX_OK;
}

extern "C" wbx_status wbx_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return WBX_ERR_INVALID;
  *out = nullptr;
  if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) return WBX_ERR_OOM;
  std::memset(*out, 0, bytes);
  return WBX_OK;
}

extern "C" wbx_status wbx_host_free(void* p) {
  if (!p) return WBX_OK;
  return hipHostFree(p) == hipSuccess ? WBX_OK : WBX_ERR_DEVICE;
}

extern "C" const char* wbx_kernel_name(wbx_ctx* c) { return c ? c->mix_kernel_name : ""; }
extern "C" double wbx_render_uniform_speed(wbx_ctx* c) { return c ? c->last_uniform_speed : 0.0; }
extern "C" uint32_t wbx_xcd_count(wbx_ctx* c) { return c ? c->n_xcds : 0u; }

extern "C" wbx_status wbx_kernel_time(wbx_ctx* c, int reset, double* mix_ms_avg, uint64_t* mix_launches) {
  if (!c) return WBX_ERR_INVALID;
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, sync_main(c));
  drain_events(c);
  if (mix_ms_avg) *mix_ms_avg = c->mix_launches ? c->mix_ms_total / (double)c->mix_launches : 0.0;
  if (mix_launches) *mix_launches = c->mix_launches;
  if (reset) {
    c->mix_ms_total = 0.0;
    c->tail_ms_total = 0.0;
    c->mix_launches = 0;
    c->gap_ms_total = 0.0;
    c->gap_count = 0;
  }
  return WBX_OK;
}

extern "C" wbx_status wbx_gap_time(wbx_ctx* c, double* gap_ms_avg, uint64_t* gaps) {
  if (!c || !gap_ms_avg) return WBX_ERR_INVALID;
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, sync_main(c));
  drain_events(c);
  *gap_ms_avg = c->gap_count ? c->gap_ms_total / (double)c->gap_count : 0.0;
  if (gaps) *gaps = c->gap_count;
  return WBX_OK;
}

extern "C" wbx_status wbx_tail_time(wbx_ctx* c, double* tail_ms_avg) {
  if (!c || !tail_ms_avg) return WBX_ERR_INVALID;
  WBX_HIP(c, join_sum(c));
  WBX_HIP(c, sync_main(c));
  drain_events(c);
  *tail_ms_avg = c->mix_launches ? c->tail_ms_total / (double)c->mix_launches : 0.0;
  return WBX_OK;
}
