"""The multi-GPU exchange of libwbx (wbx_dist_*, whitebox_amd/csrc/wbx_dist.hip) on the GPU box: the same code
path N ranks run — ring of three partial-master buffers, RCCL collective on its own stream, clamp on the root into
pinned host memory — with a REAL RCCL communicator of world size 1, against the oracle and against the single-GPU
default path, and world 2 / 4 between processes that SHARE the device (RCCL's socket transport: test_ranks_that_share_the_device_…).
(N ranks on N devices: the driver's scaling run; the CPU side covers the protocol with gloo.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_ffi as O
import whitebox_amd as W
from whitebox_amd import synth
from whitebox_amd.dist import CHAIN, ORDERED, REDUCE, Dist, PinnedBuffer
from whitebox_amd.engine import build_engine

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def oracle_master(spec, n_blocks, clamp=True):
    e = O.build_oracle_engine(spec)
    e.play()
    m = np.stack([e.process(clamp=clamp)[0] for _ in range(n_blocks)])
    e.close()
    return m


@pytest.mark.parametrize("mode", [REDUCE, ORDERED, CHAIN])
@pytest.mark.parametrize("K", [4, 16])          # short renders sum in-stream, long ones on the sum stream
def test_world1_exchange_through_rccl_equals_the_oracle(mode, K):
    steps = 5
    spec = synth.make_session("dist1", 40, src_rate=44100, n_blocks=K * steps, seed=0xD151, amp=0.5)   # hot: clamps
    want = oracle_master(spec, K * steps)
    assert (np.abs(want) == 1.0).any()
    eng = build_engine(spec, max_blocks=K)
    d = Dist(eng.ctx, 0, 1, mode)
    outs = [PinnedBuffer(K * 2 * 512) for _ in range(steps)]
    eng.play()
    for i in range(steps):                       # five renders through the ring of three, no host sync in between
        eng.render(K)
        d.exchange(outs[i].ptr)
    d.sync()
    for i in range(steps):
        got = outs[i].array.reshape(K, 2, 512)
        assert np.array_equal(bits(got), bits(want[i * K:(i + 1) * K])), (mode, i)
    # the rank's own (un-clamped) partial stays readable the ordinary way
    m, pk, _ = eng.ctx.fetch(peaks=True)
    un = oracle_master(spec, K * steps, clamp=False)
    assert np.array_equal(bits(m), bits(un[(steps - 1) * K:]))
    d.shutdown()
    # ... and after shutdown the context is a single-GPU context again (clamps itself)
    eng.stop()
    eng.play()
    eng.render(K)
    m2, _, _ = eng.ctx.fetch()
    assert np.array_equal(bits(m2), bits(want[:K]))
    for o in outs:
        o.close()
    eng.close()


def test_exchange_protocol_errors():
    spec = synth.make_session("dist2", 4, n_blocks=4, seed=0xD152)
    eng = build_engine(spec, max_blocks=2)
    L = W.lib()
    assert L.wbx_dist_exchange(eng.ctx.h, None) == -4                 # not initialised
    d = Dist(eng.ctx, 0, 1, REDUCE)
    assert L.wbx_dist_exchange(eng.ctx.h, None) == -1                 # nothing rendered yet
    eng.play()
    eng.render(2)
    assert L.wbx_dist_exchange(eng.ctx.h, None) == -4                 # the root needs a destination
    out = PinnedBuffer(2 * 2 * 512)
    d.exchange(out.ptr)
    assert L.wbx_dist_exchange(eng.ctx.h, out.ptr) == -1              # one exchange per render
    import ctypes as C
    assert L.wbx_dist_init(eng.ctx.h, (C.c_char * 128)(), 0, 1, 0) == -4   # already initialised
    d.sync()
    d.shutdown()
    out.close()
    eng.close()


def test_bench_multi_gpu_launch_paths():
    """`python bench.py --gpus 2` run plainly on a one-GPU box fails with a clear message (not a launcher error), and
    --force-dist-path runs the whole multi-GPU loop through RCCL on one rank."""
    have = W.lib().wbx_device_count()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "2"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and f"needs {have + 1} gfx950 devices" in r.stderr, r.stderr[-500:]
    for mode in ("reduce", "ordered", "chain"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--blocks", "64", "--steps", "6",
                            "--warmup", "1", "--ramp-steps", "4", "--force-dist-path", "--dist-mode", mode,
                            "--no-cpu-baseline", "--no-configs"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]                  # ONE JSON line on stdout, nothing of RCCL's banner
        assert r.stdout.strip().splitlines()[-1] == lines[0]
        line = json.loads(lines[0])
        assert line["n_gpus"] == 1 and line["value"] > 0 and "wbx_dist_exchange" in line["config"]["exchange"]
        assert 0.0 < line["master_peak"] <= 1.0
        # what the exchange saw, and the head of the run against the oracle
        assert line["rccl_world"] == 1 and len(line["devices"]) == 1 and line["devices"][0].count(":") == 2
        assert line["exchange_ms_avg"] > 0.0 and line["tracks_per_gpu"] == 256
        assert line["distinct_devices"] is True and len(line["ranks"]) == 1 and line["ranks"][0]["mix_ms_avg"] > 0
        assert line["verify"]["ok"] and line["verify"]["peaks_equal"] and line["verify"]["plan_rows_equal"]


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_that_share_the_device_exchange_over_rccl(world):
    """wbx_dist.hip with world > 1 between processes, on a box with ONE GPU: every rank claims a host of its own
    (NCCL_HOSTID — bench.py sets it under WBX_SHARE_DEVICE=1), so RCCL accepts ranks that share the device and carries the
    exchange over its socket transport on the loopback interface.  Slow — but it is the launcher, the rendezvous,
    ncclCommInitRank across processes, ncclReduce / the gather + fixed-order add / the send-receive chain, and the
    all-gather of the ranks' facts, for real; the bench's own check compares the head of what the ranks rendered together
    (world x 1024 tracks) with the oracle: chain mode bit-exact — the reference's order across ranks."""
    env = dict(os.environ, WBX_SHARE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    for mode in ("reduce", "ordered", "chain"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dist-mode", mode, "--tracks", "1024",
                            "--blocks", "64", "--session-blocks", "256", "--steps", "4", "--warmup", "1", "--ramp-steps", "2",
                            "--no-cpu-baseline", "--no-configs", "--latency-blocks", "0"]
                           # (a render of 64 blocks adds 128-track groups; one group per rank = the reference's order inside a rank)
                           + (["--group-size", "1024"] if mode == "chain" else []),
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (mode, r.stderr[-2500:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        line = json.loads(lines[0])
        assert line["n_gpus"] == world and line["rccl_world"] == world and len(line["devices"]) == world
        assert line["config"]["total_tracks"] == 1024 * world and line["exchange_ms_avg"] > 0.0
        v = line["verify"]
        assert v["ok"] and v["tracks"] == 1024 * world and v["peaks_equal"] and v["plan_rows_equal"], v
        if mode == "chain":
            assert v["master_bit_exact"], v
        ranks = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith('{"rank"')]
        assert sorted(x["rank"] for x in ranks) == list(range(world))       # every rank said what it ran on
        # the record alone says what this run was: world processes on ONE device, the exchange over RCCL's socket transport
        # (N ranks on N devices would read distinct_devices true and transport "p2p")
        assert line["distinct_devices"] is False and line["transport"] == "socket", (line["transport"], line["ranks"])
        assert [x["rank"] for x in line["ranks"]] == list(range(world))
        assert all(x["mix_ms_avg"] > 0 and x["exchange_ms_avg"] > 0 and x["transport"]["kinds"] == ["socket"] for x in line["ranks"])
        assert all(x["exchange"]["world"] == world and x["exchange"]["mode"] == mode for x in ranks)


def test_dist_info_and_allgather_at_world_1():
    spec = synth.make_session("dist3", 4, n_blocks=2, seed=0xD153)
    eng = build_engine(spec, max_blocks=2)
    d = Dist(eng.ctx, 0, 1, CHAIN)
    assert d.result_rank == 0 and d.allgather(b"hello", 16) == [b"hello"]
    out = PinnedBuffer(2 * 2 * 512)
    eng.play()
    eng.render(2)
    d.exchange(out.ptr)
    d.sync()
    info = d.info()
    assert info["world"] == 1 and info["mode"] == "chain" and info["exchanges"] == 1 and info["exchange_ms_avg"] > 0.0
    assert info["devices"] == [eng.ctx.device_info()["pci"]]
    assert np.array_equal(bits(out.array.reshape(2, 2, 512)), bits(oracle_master(spec, 2)))
    d.shutdown()
    out.close()
    eng.close()
