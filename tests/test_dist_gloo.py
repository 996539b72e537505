"""The N>1 path on CPU: world_size-2/3 `gloo` processes, tracks sharded in contiguous ranges by the library's own
wbx_shard_tracks, un-clamped partial masters in libwbx's [K][C][F] layout summed onto rank 0, clamp after the sum.
There is no GPU here, so the per-rank partials come from the oracle and the exchange itself is `GlooExchange` below —
a CPU stand-in with the contract of wbx_dist_exchange (whitebox_amd/csrc/wbx_dist.hip): a ring of three partial
buffers, the exchange of render i issued asynchronously and completed after render i+1 has been issued, REDUCE mode
(collective sum, order implementation-defined) or ORDERED mode (gather + (((0 + p0) + p1) + ...) in rank order, the
arithmetic of ordered_add_kernel).  Under test: the sharding, the buffer protocol, the clamp-after-sum order and the
ordered mode's bit-reproducibility.  Real RCCL is exercised on the GPU box (tests/test_gpu_dist.py, world 1) and by
`bench.py --gpus N`."""
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


class GlooExchange:
    """CPU stand-in for wbx_dist_exchange over torch.distributed/gloo (see the module docstring)."""
    RING = 3

    def __init__(self, rank, world, mode, clamp_fn):
        self.rank, self.world, self.mode, self.clamp_fn = rank, world, mode, clamp_fn
        self.pending = {}

    def exchange(self, partial: torch.Tensor, slot: int):
        """asynchronous: returns at once; finish(slot) completes it"""
        if self.mode == "reduce":
            w = dist.reduce(partial, dst=0, op=dist.ReduceOp.SUM, async_op=True) if self.world > 1 else None
            self.pending[slot] = (w, partial, None)
        else:
            parts = [torch.empty_like(partial) for _ in range(self.world)] if self.rank == 0 else None
            w = dist.gather(partial, parts, dst=0, async_op=True) if self.world > 1 else None
            if self.world == 1:
                parts = [partial.clone()]
            self.pending[slot] = (w, partial, parts)

    def finish(self, slot: int):
        w, partial, parts = self.pending.pop(slot)
        if w is not None:
            w.wait()
        if self.rank != 0:
            return None
        if parts is not None:                      # ordered_add_kernel: start from the cleared buffer, rank order
            acc = np.zeros(partial.shape, np.float32)
            for p in parts:
                acc = (acc + p.numpy()).astype(np.float32)
            out = acc
        else:
            out = partial.numpy().copy()
        self.clamp_fn(out)
        return out


def _clamp_with_oracle(O):
    def clamp(a):      # the root's clamp (engine.cpp:1627-1636); on the GPU: clamp_into_kernel / ordered_add_kernel
        for b in range(a.shape[0]):
            chans = [a[b, c] for c in range(a.shape[1])]
            O.lib().wbo_master_clamp(O.planar_ptrs(chans), a.shape[1], a.shape[2])
    return clamp


def _worker(rank, world, port, n_tracks, amp, mode, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_ffi as O
    from whitebox_amd.dist import shard_tracks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = _session(n_tracks, amp)
    first, count = shard_tracks(n_tracks, world, rank)
    e = O.build_oracle_engine(_shard_spec(spec, first, count))
    e.play()
    partial = np.stack([e.process(clamp=False)[0] for _ in range(N_BLOCKS)])       # [K][C][F], un-clamped
    e.close()
    ex = GlooExchange(rank, world, mode, _clamp_with_oracle(O))
    ex.exchange(torch.from_numpy(partial.copy()), 0)
    out = ex.finish(0)
    if rank == 0:
        q.put((out, partial))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, target, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("mode", ["reduce", "ordered"])
@pytest.mark.parametrize("n_tracks,amp", [(22, None), (23, 0.6)])
def test_two_rank_shard_exchange_clamp(n_tracks, amp, mode):
    import oracle_ffi as O
    from whitebox_amd.dist import shard_tracks
    got, part0 = _run(2, _worker, (n_tracks, amp, mode))
    e = O.build_oracle_engine(_session(n_tracks, amp))
    e.play()
    want = np.stack([e.process()[0] for _ in range(N_BLOCKS)])
    e.close()
    d = got.astype(np.float64) - want.astype(np.float64)
    assert np.sqrt(np.mean(d * d)) <= 1e-6            # shard sums are added in a different order than the track loop
    if amp:                                           # the hot session really clamps, and only after the sum
        assert (np.abs(want) == 1.0).any() and np.abs(got).max() <= 1.0
    if mode == "ordered":
        # bit-reproducible: (0 + p0) + p1 with the shards' own partials, whatever order they arrived in
        f1, c1 = shard_tracks(n_tracks, 2, 1)
        e1 = O.build_oracle_engine(_shard_spec(_session(n_tracks, amp), f1, c1))
        e1.play()
        part1 = np.stack([e1.process(clamp=False)[0] for _ in range(N_BLOCKS)])
        e1.close()
        exp = ((np.zeros_like(part0) + part0).astype(np.float32) + part1).astype(np.float32)
        _clamp_with_oracle(O)(exp)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_shard_ranges_cover_all_tracks():
    from whitebox_amd.dist import shard_tracks
    for n in (1, 7, 8, 4096, 32768, 1000):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                f, c = shard_tracks(n, w, r)
                seen += list(range(f, f + c))
            assert seen == list(range(n))
    assert shard_tracks(32768, 8, 3) == (12288, 4096)        # BASELINE configs[4]: tracks [4096 g, 4096 (g+1)) on GPU g


def test_rendezvous_file_hands_the_id_to_the_other_ranks(tmp_path, monkeypatch):
    """rank 0 publishes the 128-byte communicator id atomically; the other ranks poll for it (no RCCL needed here:
    the id's origin is patched)"""
    import ctypes as C
    import threading
    from whitebox_amd import _ffi, dist as wd
    monkeypatch.setenv("WBX_RDZV", str(tmp_path / "rdzv"))

    class FakeLib:
        def wbx_dist_new_id(self, buf):
            C.memmove(buf, bytes(range(128)), 128)
            return 0
    monkeypatch.setattr(_ffi, "lib", lambda: FakeLib())
    got = {}
    th = threading.Thread(target=lambda: got.setdefault("id", bytes(wd.exchange_id(1, 2, timeout_s=20))))
    th.start()
    root = bytes(wd.exchange_id(0, 2))
    th.join(timeout=30)
    assert root == bytes(range(128)) and got["id"] == root
    with pytest.raises(TimeoutError):
        monkeypatch.setenv("WBX_RDZV", str(tmp_path / "nobody"))
        wd.exchange_id(1, 2, timeout_s=0.2)


def _worker_pipelined(rank, world, port, n_tracks, mode, q):
    """bench.py's N>1 loop shape: one render per step into the ring of three partial buffers, the exchange of step i
    issued asynchronously, completed after the render of step i+1 has been issued, the last one drained."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_ffi as O
    from whitebox_amd.dist import shard_tracks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = _session(n_tracks, 0.6)
    for s in spec.samples:
        s.frames *= 4                      # 10 blocks are rendered here
    first, count = shard_tracks(n_tracks, world, rank)
    e = O.build_oracle_engine(_shard_spec(spec, first, count))
    e.play()
    NS, K, steps = GlooExchange.RING, 2, 5
    masters = [torch.zeros(K, 2, 512) for _ in range(NS)]
    results = []
    ex = GlooExchange(rank, world, mode, _clamp_with_oracle(O))
    for i in range(steps):
        slot = i % NS
        masters[slot].copy_(torch.from_numpy(np.stack([e.process(clamp=False)[0] for _ in range(K)])))   # "render"
        ex.exchange(masters[slot], slot)
        if i >= 1:
            results.append(ex.finish((i - 1) % NS))
    results.append(ex.finish((steps - 1) % NS))
    e.close()
    if rank == 0:
        q.put(np.concatenate(results))
    dist.barrier()
    dist.destroy_process_group()


def _worker_chain(rank, world, port, n_tracks, amp, q):
    """WBX_DIST_CHAIN's protocol with gloo send / recv: five renders of two blocks; rank g receives rank g-1's running,
    un-clamped master for render i, continues it with its own tracks (the oracle's Engine::process without the output
    clear = what the mix kernel does with MixArgs::init), hands it to rank g+1; the LAST rank clamps and holds the
    result.  Receives land in a ring of three incoming buffers and sends leave from a ring of three partial buffers, as
    in wbx_dist.hip; every rank issues recv(i) / render(i) / send(i) in that order, so the ranks form a pipeline."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_ffi as O
    from whitebox_amd.dist import shard_tracks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = _session(n_tracks, amp)
    for s in spec.samples:
        s.frames *= 4
    first, count = shard_tracks(n_tracks, world, rank)
    e = O.build_oracle_engine(_shard_spec(spec, first, count))
    e.play()
    NS, K, steps = 3, 2, 5
    incoming = [torch.zeros(K, 2, 512) for _ in range(NS)]
    partial = [torch.zeros(K, 2, 512) for _ in range(NS)]
    sends, results = {}, []
    clamp = _clamp_with_oracle(O)
    for i in range(steps):
        slot = i % NS
        if rank > 0:
            dist.recv(incoming[slot], src=rank - 1)
        else:
            incoming[slot].zero_()
        run = incoming[slot].numpy()
        out = np.stack([e.process_from(run[b]) for b in range(K)])
        if slot in sends:                                      # the slot's previous send must be out before it is refilled
            sends.pop(slot).wait()
        partial[slot].copy_(torch.from_numpy(out))
        if rank + 1 < world:
            sends[slot] = dist.isend(partial[slot], dst=rank + 1)
        else:
            final = partial[slot].numpy().copy()
            clamp(final)
            results.append(final)
    for w in sends.values():
        w.wait()
    e.close()
    if rank == world - 1:
        q.put(np.concatenate(results))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_tracks,amp", [(2, 23, 0.6), (3, 20, None), (3, 22, 0.6)])
def test_chain_mode_is_the_reference_order_across_ranks(world, n_tracks, amp):
    """Bit-identical to ONE engine over all tracks — hot sessions included, where the sum of shard sums is not."""
    import oracle_ffi as O
    got = _run(world, _worker_chain, (n_tracks, amp))
    spec = _session(n_tracks, amp)
    for s in spec.samples:
        s.frames *= 4
    e = O.build_oracle_engine(spec)
    e.play()
    want = np.stack([e.process()[0] for _ in range(10)])
    e.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if amp:
        assert (np.abs(want) == 1.0).any()


@pytest.mark.parametrize("world,n_tracks,mode", [(2, 23, "reduce"), (3, 20, "reduce"), (3, 20, "ordered")])
def test_pipelined_exchange_ring_of_three(world, n_tracks, mode):
    import oracle_ffi as O
    got = _run(world, _worker_pipelined, (n_tracks, mode))
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
