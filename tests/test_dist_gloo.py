"""The N>1 path on CPU: world_size-2 `gloo` processes, tracks sharded in contiguous ranges, un-clamped
partial masters summed onto rank 0 through whitebox_amd.dist.MasterReducer, clamp after the reduce.
The per-rank partials come from the oracle (no GPU here); what is under test is the sharding, the
collective plumbing and the clamp-after-reduce order that bench.py --gpus N uses with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_TRACKS, N_BLOCKS = 22, 4          # odd split: 11 + 11; use 23 for ragged below


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _session(n_tracks, amp):
    from whitebox_amd import synth
    return synth.make_session("dist", n_tracks, n_blocks=N_BLOCKS, src_rate=44100, seed=0xD157, amp=amp)


def _shard_spec(spec, first, count):
    """The sub-session a rank owns: same global track keys / parameters, local track indices."""
    import copy
    sub = copy.copy(spec)
    sub.n_tracks = count
    sub.samples = spec.samples[first:first + count]
    sub.volumes_db = spec.volumes_db[first:first + count]
    sub.pans = spec.pans[first:first + count]
    sub.mutes = spec.mutes[first:first + count]
    sub.clips = [copy.copy(c) for c in spec.clips if first <= c.track < first + count]
    for c in sub.clips:
        c.track -= first
    return sub


def _worker(rank, world, port, n_tracks, amp, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_ffi as O
    from whitebox_amd.dist import MasterReducer, shard_tracks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = _session(n_tracks, amp)
    first, count = shard_tracks(n_tracks, world, rank)
    e = O.build_oracle_engine(_shard_spec(spec, first, count))
    e.play()
    partial = np.stack([e.process(clamp=False)[0] for _ in range(N_BLOCKS)])       # [K][C][F], un-clamped
    e.close()
    t = torch.from_numpy(partial.copy())

    def finalize(buf):      # the root's clamp (engine.cpp:1627-1636); on the GPU this is wbx_finalize_master
        a = buf.numpy()
        for b in range(a.shape[0]):
            chans = [a[b, c] for c in range(a.shape[1])]
            O.lib().wbo_master_clamp(O.planar_ptrs(chans), a.shape[1], a.shape[2])

    red = MasterReducer(finalize, root=0)
    red.reduce(t, slot=0)
    red.finish(t, slot=0)
    if rank == 0:
        q.put(t.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_tracks,amp", [(22, None), (23, 0.6)])
def test_two_rank_shard_reduce_clamp(n_tracks, amp):
    import oracle_ffi as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_tracks, amp, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    e = O.build_oracle_engine(_session(n_tracks, amp))
    e.play()
    want = np.stack([e.process()[0] for _ in range(N_BLOCKS)])
    e.close()
    d = got.astype(np.float64) - want.astype(np.float64)
    assert np.sqrt(np.mean(d * d)) <= 1e-6            # shard sums are added in a different order than the track loop
    if amp:                                           # the hot session really clamps, and only after the reduce
        assert (np.abs(want) == 1.0).any() and np.abs(got).max() <= 1.0


def test_shard_ranges_cover_all_tracks():
    from whitebox_amd.dist import shard_tracks
    for n in (1, 7, 8, 4096, 32768, 1000):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                f, c = shard_tracks(n, w, r)
                seen += list(range(f, f + c))
            assert seen == list(range(n))


def _worker_pipelined(rank, world, port, n_tracks, q):
    """bench.py's N>1 loop shape: one render per step into a ring of three master buffers, the reduce of step i
    enqueued asynchronously, the finalize (root clamp) of step i-1 issued afterwards, the last one drained."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_ffi as O
    from whitebox_amd.dist import MasterReducer, shard_tracks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = _session(n_tracks, 0.6)
    for s in spec.samples:
        s.frames *= 4                      # 10 blocks are rendered here
    first, count = shard_tracks(n_tracks, world, rank)
    e = O.build_oracle_engine(_shard_spec(spec, first, count))
    e.play()
    NS, K, steps = 3, 2, 5
    masters = [torch.zeros(K, 2, 512) for _ in range(NS)]
    results = []

    def finalize(buf):
        a = buf.numpy()
        for b in range(a.shape[0]):
            chans = [a[b, c] for c in range(a.shape[1])]
            O.lib().wbo_master_clamp(O.planar_ptrs(chans), a.shape[1], a.shape[2])
        results.append(a.copy())

    red = MasterReducer(finalize, root=0)
    for i in range(steps):
        slot = i % NS
        masters[slot].copy_(torch.from_numpy(np.stack([e.process(clamp=False)[0] for _ in range(K)])))   # "render"
        red.reduce(masters[slot], slot=slot)
        if i >= 1:
            red.finish(masters[(i - 1) % NS], slot=(i - 1) % NS)
    red.finish(masters[(steps - 1) % NS], slot=(steps - 1) % NS)
    e.close()
    if rank == 0:
        q.put(np.concatenate(results))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_tracks", [(2, 23), (3, 20)])
def test_pipelined_reduce_ring_of_three(world, n_tracks):
    import oracle_ffi as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, world, port, n_tracks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from whitebox_amd import synth
    spec = synth.make_session("dist", n_tracks, n_blocks=N_BLOCKS, src_rate=44100, seed=0xD157, amp=0.6)
    for s in spec.samples:
        s.frames *= 4                      # 10 blocks are rendered here
    e = O.build_oracle_engine(spec)
    e.play()
    want = np.stack([e.process()[0] for _ in range(10)])
    e.close()
    assert got.shape == want.shape
    d = got.astype(np.float64) - want.astype(np.float64)
    assert np.sqrt(np.mean(d * d)) <= 1e-6
    assert np.abs(got).max() <= 1.0
