"""bench.py's own N-rank launcher (`python bench.py --gpus N` run plainly) without a GPU: the ranks are stub commands.
Covered: the environment every rank gets (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / the rendezvous path and this
launch's nonce), failure propagation (a rank that dies ends the others instead of leaving them in the RCCL rendezvous),
ranks that ignore SIGTERM, rendezvous-file cleanup, and the reader side of the rendezvous ignoring a stale file."""
import importlib.util
import json
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stub(tmp_path, body):
    path = tmp_path / "rank_stub.py"
    path.write_text("import json, os, signal, sys, time\n" + body)
    return [sys.executable, str(path)]


def test_every_rank_gets_its_environment_and_the_file_is_cleaned_up(tmp_path):
    out = tmp_path / "env"
    out.mkdir()
    cmd = _stub(tmp_path, f"""
keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "WBX_RDZV", "WBX_RDZV_NONCE", "HSA_ENABLE_IPC_MODE_LEGACY"]
env = {{k: os.environ.get(k) for k in keys}}
if env["RANK"] == "0":
    open(env["WBX_RDZV"], "wb").write(b"x" * 144)      # what rank 0 of a real launch leaves behind
json.dump(env, open(os.path.join({str(out)!r}, env["RANK"]), "w"))
""")
    rc = _bench().self_launch(3, rank_cmd=cmd)
    assert rc == 0
    envs = [json.load(open(out / str(r))) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and len({e["WBX_RDZV"] for e in envs}) == 1
    assert len({e["WBX_RDZV_NONCE"] for e in envs}) == 1 and envs[0]["WBX_RDZV_NONCE"]
    assert not os.path.exists(envs[0]["WBX_RDZV"])          # removed by the launcher


def test_a_failing_rank_ends_the_launch(tmp_path):
    """rank 1 dies at once with exit code 7; ranks 0 and 2 would wait for a minute (the RCCL rendezvous): they are
    terminated, and the launcher returns 7 within seconds"""
    cmd = _stub(tmp_path, """
if os.environ["RANK"] == "1":
    sys.exit(7)
time.sleep(60)
""")
    t0 = time.time()
    rc = _bench().self_launch(3, rank_cmd=cmd)
    assert rc == 7 and time.time() - t0 < 20


def test_a_rank_that_ignores_sigterm_is_killed(tmp_path):
    marker = tmp_path / "armed"
    cmd = _stub(tmp_path, f"""
if os.environ["RANK"] == "0":
    signal.signal(signal.SIGTERM, signal.SIG_IGN)
    open({str(marker)!r}, "w").write("1")
    time.sleep(60)
while not os.path.exists({str(marker)!r}):
    time.sleep(0.01)
sys.exit(5)
""")
    t0 = time.time()
    rc = _bench().self_launch(2, rank_cmd=cmd, grace_s=1.0)
    assert rc == 5 and time.time() - t0 < 20


def test_a_stale_rendezvous_file_is_not_taken_for_this_launch(tmp_path, monkeypatch):
    """A crashed launch left a 144-byte file at the path: a reader of THIS launch (another nonce) keeps waiting for its
    own rank 0, and takes the id as soon as a file with its nonce is there."""
    import threading
    from whitebox_amd import dist as wd
    path = tmp_path / "rdzv"
    monkeypatch.setenv("WBX_RDZV", str(path))
    monkeypatch.setenv("WBX_RDZV_NONCE", "launch-A")
    stale = wd.launch_nonce()
    path.write_bytes(stale + bytes([1]) * 128)
    monkeypatch.setenv("WBX_RDZV_NONCE", "launch-B")
    assert wd.launch_nonce() != stale
    with pytest.raises(TimeoutError):
        wd.exchange_id(1, 2, timeout_s=0.3)
    got = {}
    th = threading.Thread(target=lambda: got.setdefault("id", bytes(wd.exchange_id(1, 2, timeout_s=20))))
    th.start()
    time.sleep(0.2)
    tmp = tmp_path / "rdzv.tmp"
    tmp.write_bytes(wd.launch_nonce() + bytes([2]) * 128)
    os.replace(tmp, path)
    th.join(timeout=30)
    assert got["id"] == bytes([2]) * 128


def test_rccl_transport_lines_are_recognised(tmp_path):
    """bench.py reads what carried a rank's exchange from RCCL's own log: peer-to-peer between devices (xGMI / PCIe), host
    shared memory, or a network transport (what ranks that share one device fall back to)"""
    import bench
    log = tmp_path / "rccl.log"
    log.write_text("\n".join([
        "runc:71:102 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC",
        "runc:71:102 [0] NCCL INFO Channel 01/0 : 0[1c000] -> 2[1d000] via SHM/direct/direct",
        "runc:71:102 [0] NCCL INFO Channel 00/0 : 0[0] -> 3[0] [send] via NET/Socket/0",
        "runc:71:102 [0] NCCL INFO Channel 00/0 : 3[0] -> 0[0] [receive] via NET/Socket/0",
        "runc:71:102 [0] NCCL INFO Channel 00/0 : 0[0] -> 4[0] [send] via NET/IB/1/GDRDMA",
        "runc:71:102 [0] NCCL INFO Connected all rings", ""]))
    got = bench.parse_rccl_transports(str(log), 0)
    assert got["kinds"] == ["net:IB", "p2p", "shm", "socket"] and got["lines"] == 5
    assert got["peers"] == {"1": ["p2p"], "2": ["shm"], "3": ["socket"], "4": ["net:IB"]}
    assert bench.parse_rccl_transports(str(tmp_path / "absent.log"), 0)["kinds"] == []


def test_a_stale_reference_executable_is_refused(tmp_path):
    """bench.py's CPU baseline times oracle/_ref/wbref_engine, a prebuilt, untracked file that travels with the tree: it is taken
    only when the stamp oracle/Makefile left beside it is the hash of THIS tree's driver source + recipe (advisor, round 5) — a
    build of an older driver, an unstamped copy or a missing file fall back to the port."""
    import hashlib
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_stamp", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    od = tmp_path / "oracle"
    (od / "_ref").mkdir(parents=True)
    (od / "ref_engine_driver.cpp").write_text("// driver v2\n")
    (od / "Makefile").write_text("all:\n")
    exe = od / "_ref" / "wbref_engine"
    want = hashlib.sha256(b"// driver v2\nall:\n").hexdigest()[:16]
    assert b.ref_engine_stamp(str(exe), str(od)) == (False, want, None, None)                 # nothing there
    exe.write_bytes(b"\x7fELF...")
    assert b.ref_engine_stamp(str(exe), str(od))[:3] == (False, want, None)                   # unstamped
    (od / "_ref" / "wbref_engine.stamp").write_text("0123456789abcdef\n")
    assert b.ref_engine_stamp(str(exe), str(od))[:3] == (False, want, "0123456789abcdef")     # built from another driver / recipe
    (od / "_ref" / "wbref_engine.stamp").write_text(want + "\n")
    ok, w, st, sha = b.ref_engine_stamp(str(exe), str(od))
    assert ok and st == want and sha == hashlib.sha256(b"\x7fELF...").hexdigest()[:16]
    # ... and the tree's own executable, where it has been built, carries the tree's stamp
    real = os.path.join(root, "oracle", "_ref", "wbref_engine")
    if os.path.exists(real):
        assert b.ref_engine_stamp(real, os.path.join(root, "oracle"))[0]
