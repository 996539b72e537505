// wbx_callback.h — the real-time callback as ONE launch.
//
// The reference's only caller is the audio thread: one Engine::process per device period (audio_io_pulseaudio.cpp:396-466,
// audio_io_wasapi.cpp:708).  As three dependent launches — sequencer, mix, sum — a block cost ~35 us of launch-to-launch
// latency whatever it computed.  callback_kernel is those three in one dispatch, without a device-wide barrier:
//
//   1. prologue  every workgroup is a track group of the block (the grid of mix_kernel for K = 1); its first lanes run the
//                sequencer (wbx_seq.h plan_track: Track::process_event + the segment loop, one lane per track) for the
//                group's OWN tracks — every track belongs to exactly one group — and write rows and templates where the
//                mix expects them;
//   2. mix       mix_body (wbx_mix.h), unchanged: the group's sum goes to `partial` (a one-group session: straight to the
//                master);
//   3. epilogue  the group sums become the master without a fourth party.  ONE workgroup adding the 256 group sums of a
//                4096-track block moves 1 MiB through one CU's L1 (64 B per clock: 8 us before any latency — measured 15-19);
//                so when the grid fits the device at once (at most one workgroup per CU: they are all resident), the
//                workgroups meet at a ticket counter and EACH adds a share of the block's frames (sum_block, wbx_sum.h — the
//                same code and the same order of additions as sum_kernel: a lane owns 4 frames and walks the groups in
//                order), clamps, converts and stores them into pinned host memory; the one that finishes last copies the
//                plan status and then writes the launch's sequence number, which the audio thread polls: no completion
//                signal, no interrupt between the device and the callback's return.  A grid larger than the device keeps
//                "the last workgroup adds everything" (a spin barrier would wait for workgroups that cannot start).
//
// Results are those of the three-launch path bit for bit (same functions, same order).  Occupancy does not matter here, so
// the instances are built for two waves per SIMD (256 registers): the sequencer's locals stay in registers.
#pragma once
#include "wbx_mix.h"
#include "wbx_seq.h"
#include "wbx_sum.h"

namespace wbx {

struct CallbackArgs {
  uint32_t* done;      // device, two counters of kCbLanes words kCbStride apart each: workgroups that have stored their group
                       // sum / their share of the master — counted from `base`, never reset
  uint32_t spread;     // every workgroup adds a share of the master (the grid is resident at once); 0: the last one adds it all
  uint32_t base;       // what the first counter reads when this launch starts (every multi-group launch arrives there)
  uint32_t base2;      // ... the second one (only spread launches arrive there: it has a base of its own)
  uint32_t* elect;     // device word: the launch number of the last launch whose reporter has been chosen (not spread: of the
                       // workgroups that see the full count, the first to swap its launch's number in adds and reports);
                       // (give-ups at the spread barrier travel with the second tickets: kCbGaveUp / kCbGaveUp0)
  uint32_t* gave_up;   // pinned host: `seq` — written in front of `flag` — when a workgroup of this launch gave up at the spread barrier
  uint32_t* flag;      // pinned host: `seq` once master and status are out.  (One word per workgroup, the host waiting for all of
                       // them, was tried instead of the second ticket: 3 us less on the device, 6 us more until the audio thread
                       // had seen all 256 — the words share cache lines the polling core keeps losing to the next write.)
  uint32_t seq;
  uint32_t n_wgs;
  uint32_t fenced;     // A/B aid (WBX_CB_FENCED=1): release / acquire fences instead of write-through stores + s_waitcnt
  uint32_t spin_bound; // polls of the spread barrier before a workgroup gives up (WBX_CB_SPIN_BOUND: tests force the give-up path)
  unsigned long long* dbg;   // diagnostic (WBX_CB_DBG=1): [n_wgs][6] wall-clock ticks at start / plan done / mix done / ticket / end, XCC id
};

// A ticket counter 256 workgroups arrive at within a microsecond is ONE address at the memory-side atomic unit: the arrivals
// serialise (≈20 ns each: 5-6 us before the last one is through).  The counter is therefore kCbLanes words, kCbStride words
// apart (different memory channels): workgroup w adds to word w % kCbLanes, and whoever needs the total reads all of them —
// lanes 0..15 of a wave, one load each, one round trip — and adds them up.  tid < 64 must call these together.
constexpr uint32_t kCbLanes = 16, kCbStride = 64;
// the second ticket's total: arrivals in its low ten bits (at most 256 workgroups spread), workgroups that had given up at the
// barrier in the next ten, "workgroup 0 was one of them" above
constexpr uint32_t kCbGaveUp = 1u << 10, kCbGaveUp0 = 1u << 20;
__device__ __forceinline__ uint32_t cb_total(const uint32_t* cnt, uint32_t lane) {
  uint32_t v = lane < kCbLanes ? __hip_atomic_load(cnt + lane * kCbStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  v += (uint32_t)__shfl_xor((int)v, 8, 64);
  v += (uint32_t)__shfl_xor((int)v, 4, 64);
  v += (uint32_t)__shfl_xor((int)v, 2, 64);
  v += (uint32_t)__shfl_xor((int)v, 1, 64);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// arrive, then the total as it stands once this workgroup's own arrival is through (the workgroup whose arrival is the
// last one to complete sees them all)
__device__ __forceinline__ uint32_t cb_arrive(uint32_t* cnt, uint32_t wg, uint32_t lane, uint32_t inc = 1u) {
  uint32_t old = 0u;
  if (lane == 0u) old = __hip_atomic_fetch_add(cnt + (wg % kCbLanes) * kCbStride, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);   // (the returned value orders the reads below behind the arrival)
  asm volatile("" ::"s"(old) : "memory");
  return cb_total(cnt, lane);
}

template <int U, int FAM>
__global__ __launch_bounds__(256, 2) void callback_kernel(MixArgs a, PlanArgs p, SumArgs s, CallbackArgs cb) {
  const uint32_t tid = threadIdx.x;
  const uint32_t wg = blockIdx.y;
  if (cb.dbg && tid == 0u) {
    cb.dbg[6u * wg] = wall_clock64();
    cb.dbg[6u * wg + 5u] = (unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
  }
  {
    // -- 1. the sequencer of this group's tracks, for the one block (Engine::process's transport: engine.cpp:1578-1585)
    __shared__ DBlockTime s_time;
    if (tid == 0u) block_times(p, &s_time);
    __syncthreads();
    const DGroup grp = a.groups[blockIdx.y];
    for (uint32_t i = tid; i < grp.count; i += 256u) plan_track(p, a.order[grp.first + i], &s_time);
    // rows, templates and per-track state are out (acknowledged by the L2 this CU sits behind: the vector L1 writes through)
    // before any lane of the workgroup stages them.  No cache maintenance: nothing of this was in this CU's L1 before.
    // (Handing rows and templates to the mix through LDS as well — its staging then reads no memory — was measured: the mix
    //  phase stayed at 11.5 us, the sequencer phase grew by 0.5 us, an 8-track block by 2 us.  Not kept.)
    if (cb.fenced) __threadfence();
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  if (cb.dbg && tid == 0u) cb.dbg[6u * wg + 1u] = wall_clock64();
  // -- 2. the mix of this group (512-frame stereo / 1024-frame mono: one workgroup per block and group)
  mix_body<U, true, FAM, 1, 1, 1, 256>(a);
  if (cb.dbg && tid == 0u) cb.dbg[6u * wg + 2u] = wall_clock64();

  // -- 3. the block's sum.  The group sums cross XCDs (each has its own L2): mix_body stored this group's with agent-scope
  // stores (MixArgs::partial_through: written through to memory, acknowledged before the ticket is taken).  They are read
  // with ordinary 16-B loads: an XCD's L2 cannot hold an older copy of another group's row — the launch began with clean
  // caches and nothing on that XCD has touched those lines since — so every one of them misses and is answered by memory.
  // No L2 write-back / invalidate anywhere: a release / acquire fence pair cost every one of the 256 workgroups of a
  // 4096-track block its share of an L2 flush (measured: 77 -> 120 us per block), and reading the sums past the L2 dword by
  // dword (agent-scope loads) cost 56 us.
  __shared__ uint32_t s_ticket;
  bool report = true;   // this workgroup writes the flag
  uint32_t second_total = 0u;   // spread: the second counter as this workgroup's arrival found it (count + give-up markers)
  if (!a.fused_master) {
    if (cb.fenced) __threadfence();
    __builtin_amdgcn_s_waitcnt(0);   // this wave's stores are acknowledged ...
    __syncthreads();                 // ... every wave's
    if (tid < 64u) {
      // (the counters are never reset — nobody knows when the last poller has left; the host hands every launch the count
      //  all earlier launches have left behind, cb.base)
      uint32_t n = cb_arrive(cb.done, wg, tid) - cb.base;
      if (cb.spread) {   // wait for the others (all resident: the host spreads only grids of at most one workgroup per CU)
        uint32_t spins = 0u;
        // (bounded: ~50 ms, a thousand times what the slowest workgroup of a block takes; a give-up is reported — with the second
        //  ticket and the pinned gave_up word, below — never a hang: the host mixes the block again through three launches and
        //  the context stops spreading)
        while (n < cb.n_wgs && spins < cb.spin_bound) {
          __builtin_amdgcn_s_sleep(1);
          n = cb_total(cb.done, tid) - cb.base;
          spins++;
        }
        // (a give-up — n < n_wgs in s_ticket — travels with this workgroup's SECOND ticket: kCbGaveUp on top of its arrival, and
        //  kCbGaveUp0 when it is workgroup 0, whose plan counters then have not gone out; the workgroup that reports the launch
        //  reads both out of the total it takes anyway — no word of its own, no extra round trip behind the second ticket)
      }
      if (tid == 0u) s_ticket = n;
    }
    __syncthreads();
    if (cb.dbg && tid == 0u) cb.dbg[6u * wg + 3u] = wall_clock64();
    // (not spread: whoever sees the full count adds everything — two workgroups whose arrivals complete together may both
    //  see it; ONE of them is chosen, or the second would copy the plan's counters after the first has cleared them)
    if (!cb.spread) {
      if (s_ticket != cb.n_wgs) return;
      __shared__ uint32_t s_chosen;
      if (tid == 0u) s_chosen = __hip_atomic_exchange(cb.elect, cb.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != cb.seq ? 1u : 0u;
      __syncthreads();
      if (!s_chosen) return;
    }
    if (cb.fenced) __threadfence();
    const uint32_t F = s.block_frames, C = s.channels;
    // a lane owns a slot (4 frames of one channel; interleaved output: of every channel) and walks the groups in order
    const uint32_t n_slots = (s.out_il ? F : C * F) >> 2;
    auto add_slot = [&](uint32_t slot, const PartialFromLds* lds) {
      if (lds) {
        if (s.out_il) {
          if (s.n_buses) sum_block<16, true, true, true, PartialFromLds>(s, 0u, slot, lds);
          else sum_block<16, false, true, true, PartialFromLds>(s, 0u, slot, lds);
        } else {
          if (s.n_buses) sum_block<16, true, false, true, PartialFromLds>(s, 0u, slot, lds);
          else sum_block<16, false, false, true, PartialFromLds>(s, 0u, slot, lds);
        }
      } else if (s.out_il) {
        if (s.n_buses) sum_block<16, true, true, true>(s, 0u, slot);
        else sum_block<16, false, true, true>(s, 0u, slot);
      } else {
        if (s.n_buses) sum_block<16, true, false, true>(s, 0u, slot);
        else sum_block<32, false, false, true>(s, 0u, slot);
      }
    };
    if (cb.spread && cb.n_wgs <= 256u) {
      // Spread: workgroup w owns slots w, w + n_wgs, ...  A lane walking the 256 groups of a 4096-track block by itself is
      // eight dependent memory round trips (32 loads in flight); instead the workgroup's 256 lanes fetch (slot, group) pairs —
      // one 16-B load each, ONE round trip — into LDS, and one lane per slot adds them from there in group order.
      __shared__ f4 s_part[2][256];
      const uint32_t ng = s.n_groups;
      const uint32_t spr = 256u / ng;                 // slots a round covers (n_wgs == n_groups <= 256: at least one)
      const size_t stride = (size_t)C * F;
      const uint32_t ls = tid / ng, g = tid - ls * ng;
      for (uint32_t k0 = 0u; wg + k0 * cb.n_wgs < n_slots; k0 += spr) {
        const uint32_t slot = wg + (k0 + ls) * cb.n_wgs;
        if (ls < spr && slot < n_slots) {
          const float* src = s.partial + (size_t)g * stride + (size_t)slot * 4u;
          s_part[0][ls * ng + g] = *reinterpret_cast<const f4*>(src);
          if (s.out_il && C > 1u) s_part[1][ls * ng + g] = *reinterpret_cast<const f4*>(src + F);
        }
        __syncthreads();
        if (!s.out_il && !s.n_buses && !s.chain && ng >= 32u) {
          // (round 6) the usual block — planar fp32 master, no sub-buses — of a session of many groups: a slot's FOUR frames go
          // to four lanes, one dependent chain of ng additions each, instead of one lane issuing all 4 * ng of them (a 4096-track
          // block: 256 group sums, ~3 us of one lane's instruction issue behind the barrier, a quarter of that this way).  The
          // same additions in the same order per sample as sum_block; the quad's first lane stores the 16 bytes.
          if (tid < 64u) {
            const uint32_t q = tid >> 2, k = tid & 3u;
            const uint32_t my = wg + (k0 + q) * cb.n_wgs;
            const bool own = q < spr && my < n_slots;
            const float* col = reinterpret_cast<const float*>(&s_part[0][(own ? q : 0u) * ng]) + k;   // group g's sum of this frame: col[4 * g]
            float acc = 0.0f;
            constexpr int kAhead = 32;
            for (uint32_t g0 = 0u; g0 < ng; g0 += kAhead) {
              float v[kAhead];
#pragma unroll
              for (int i = 0; i < kAhead; i++) v[i] = col[4u * (g0 + i < ng ? g0 + i : ng - 1u)];   // (clamped: straight-line reads)
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int i = 0; i < kAhead; i++)
                if (g0 + i < ng) acc = __fadd_rn(acc, v[i]);                                         // audio_buffer.h:73-82, group order
            }
            if (s.clamp) acc = acc > 1.0f ? 1.0f : (acc < -1.0f ? -1.0f : acc);                      // engine.cpp:1627-1636 (compares: NaN passes)
            const uint32_t base = tid & ~3u;
            const float y = __shfl(acc, (int)base + 1, 64), z = __shfl(acc, (int)base + 2, 64), w = __shfl(acc, (int)base + 3, 64);
            if (own && k == 0u)
              store_u4_system(s.master + (size_t)my * 4u, uint4{__float_as_uint(acc), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)});
          }
        } else {
          const uint32_t my = wg + (k0 + tid) * cb.n_wgs;
          if (tid < spr && my < n_slots) {
            const PartialFromLds lds{{&s_part[0][tid * ng], &s_part[(s.out_il && C > 1u) ? 1 : 0][tid * ng]}, s.out_il ? F : 0xFFFFFFFFu};
            add_slot(my, &lds);
          }
        }
        __syncthreads();
      }
    } else {
      const uint32_t first = cb.spread ? wg + cb.n_wgs * tid : tid, step = cb.spread ? cb.n_wgs * 256u : 256u;
      for (uint32_t slot = first; slot < n_slots; slot += step) add_slot(slot, nullptr);
    }
    // the plan's counters go to the host with workgroup 0's share (spread: beside the others' sums, not behind the second
    // ticket) / with the last workgroup's master.  Only a workgroup that has SEEN the full count may touch them: every
    // sequencer lane of the launch is then over, the counters are final and clearing them takes nothing from a plan_track
    // still running.  Workgroup 0 of a spread launch that gave up at the barrier leaves them alone and says so (its second ticket);
    // the launch's reporter — behind the second ticket, where every workgroup is through — copies them instead, uncleared.
    if (!cb.spread || wg == 0u) {
      if (cb.spread && s_ticket != cb.n_wgs) {
        // (nothing: its second ticket says so)
      } else if (s.status_dst && tid < 4u) {   // (as sum_kernel does: the plan's counters for the host, cleared for the buffer's next plan)
        const uint32_t queued = __hip_atomic_load(s.status_src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(s.status_dst + tid, __hip_atomic_load(s.status_src + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        if (s.zero_status && queued == 0u) __hip_atomic_store(s.status_src + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (cb.spread) {   // whoever stores its share last reports (its own stores and, through the ticket, everybody's are acknowledged)
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      const bool gave_up_here = s_ticket != cb.n_wgs;   // (the first ticket's count as this workgroup left the barrier)
      __syncthreads();                                 // (every wave has read s_ticket before it is written again)
      if (tid < 64u) {
        const uint32_t inc = 1u + (gave_up_here ? kCbGaveUp + (wg == 0u ? kCbGaveUp0 : 0u) : 0u);
        const uint32_t n = cb_arrive(cb.done + kCbLanes * kCbStride, wg, tid, inc) - cb.base2;
        if (tid == 0u) s_ticket = n;
      }
      __syncthreads();
      // (two may see the full count: both write the same words, the status went out with workgroup 0's share.  A launch with a
      //  give-up leaves the marker bits in the counter for good — the context never spreads again: this counter is not read any more)
      report = (s_ticket & (kCbGaveUp - 1u)) == cb.n_wgs;
      second_total = s_ticket;
    }
  }
  if (!report) return;
  // did a workgroup of this launch give up at the barrier — did workgroup 0, whose plan counters then have not gone out?  Both
  // came with the second tickets (a spread launch; 0 otherwise)
  const uint32_t gave = (second_total / kCbGaveUp) & (kCbGaveUp - 1u), wg0_gave = second_total / kCbGaveUp0;
  // (a spread launch whose workgroup 0 gave up: the counters have not gone out — every workgroup is past its second ticket
  //  now, so they are final; copied, not cleared: the host mixes this block again and clears them itself)
  if (wg0_gave && s.status_dst && tid < 4u)
    __hip_atomic_store(s.status_dst + tid, __hip_atomic_load(s.status_src + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  // master and status (pinned host memory) went out as system-scope stores: once this workgroup's are acknowledged — every
  // wave's — they are on their way to the host in front of its flag
  if (cb.fenced) __threadfence_system();
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0u) {
    if (gave && cb.gave_up) {
      __hip_atomic_store(cb.gave_up, cb.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_s_waitcnt(0);   // (on its way to the host in front of the flag)
    }
    __hip_atomic_store(cb.flag, cb.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (cb.dbg) cb.dbg[6u * wg + 4u] = wall_clock64();
  }
}

#define WBX_CALLBACK(U, FAM)                                                                             \
  {                                                                                                      \
    name = "wbx::callback_kernel<" #U ", " #FAM ">";                                                     \
    hipLaunchKernelGGL((callback_kernel<U, FAM>), dim3(1, a.n_groups, 1), dim3(256), 0, st, a, p, s, cb); \
  }

// one per family (wbx_mix_fam<N>.hip); -> the instance's name
const char* launch_callback_fam0(const MixArgs& a, const PlanArgs& p, const SumArgs& s, const CallbackArgs& cb, bool window, hipStream_t st);
const char* launch_callback_fam1(const MixArgs& a, const PlanArgs& p, const SumArgs& s, const CallbackArgs& cb, hipStream_t st);
const char* launch_callback_fam2(const MixArgs& a, const PlanArgs& p, const SumArgs& s, const CallbackArgs& cb, hipStream_t st);

}  // namespace wbx
