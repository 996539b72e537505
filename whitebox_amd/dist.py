"""Multi-GPU driver side of the mix path: one process per GPU, tracks sharded in contiguous ranges, one exchange
per render.  The exchange itself — partial-master ring, RCCL reduce (or gather + fixed-order add) on its own
high-priority stream, clamp on the root — lives in libwbx.so (whitebox_amd/csrc/wbx_dist.hip, wbx_dist_* in
include/wbx.h); this module only does what a host process has to: pick the rank's track range, get rank 0's
128-byte communicator id to the other ranks, and pin host memory for the root's output.

The reference has no distributed code: this is the one exchange step the path has.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional, Tuple

import numpy as np

from . import _ffi

REDUCE, ORDERED, CHAIN = 0, 1, 2


def shard_tracks(n_tracks: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous range [first, first+count) of global track indices owned by `rank` (wbx_shard_tracks): ranges are
    in rank order, so in-GPU summation order equals the reference's track order within a shard."""
    first, count = C.c_uint32(), C.c_uint32()
    _ffi.lib().wbx_shard_tracks(n_tracks, world, rank, C.byref(first), C.byref(count))
    return first.value, count.value


def rendezvous_path() -> str:
    """A file every rank of THIS launch agrees on: the launcher's port and process id (torchrun: the agent is the
    parent of every worker; bench.py's own launcher passes WBX_RDZV)."""
    if os.environ.get("WBX_RDZV"):
        return os.environ["WBX_RDZV"]
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"wbx_rdzv_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")


def launch_nonce() -> bytes:
    """16 bytes that name THIS launch, the same on all of its ranks and different from any earlier launch's: what keeps a
    reader from taking a stale rendezvous file (a crashed launch that used the same port and parent pid) for rank 0's.
    bench.py's launcher passes WBX_RDZV_NONCE; under torchrun the run id plus the agent's pid and start time serve."""
    import hashlib
    tag = os.environ.get("WBX_RDZV_NONCE")
    if not tag:
        ppid = os.getppid()
        try:
            started = open(f"/proc/{ppid}/stat").read().rsplit(")", 1)[1].split()[19]   # the parent's start time (clock ticks)
        except Exception:
            started = "?"
        tag = f"{os.environ.get('TORCHELASTIC_RUN_ID', '')}:{os.environ.get('MASTER_PORT', '0')}:{ppid}:{started}"
    return hashlib.sha256(tag.encode()).digest()[:16]


def exchange_id(rank: int, world: int, timeout_s: float = 120.0) -> C.Array:
    """Rank 0 makes the communicator id (wbx_dist_new_id) and publishes it through the rendezvous file; the other
    ranks wait for it.  The file holds the launch nonce followed by the 128 id bytes; rank 0 removes whatever lies at
    the path first, creates its file exclusively (O_EXCL, mode 0600) under a private name and renames it into place;
    readers ignore a file whose nonce is not this launch's."""
    buf = (C.c_char * 128)()
    path = rendezvous_path()
    nonce = launch_nonce()
    if rank == 0:
        st = _ffi.lib().wbx_dist_new_id(buf)
        if st != 0:
            raise _ffi.WbxError(st, "wbx_dist_new_id", "RCCL could not be loaded" if st == -3 else "")
        if world > 1:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            tmp = f"{path}.{os.getpid()}.tmp"
            fd = os.open(tmp, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600)
            with os.fdopen(fd, "wb") as f:
                f.write(nonce + bytes(buf))
            os.replace(tmp, path)
        return buf
    t0 = time.time()
    while True:
        try:
            data = open(path, "rb").read()
            if len(data) == 144 and data[:16] == nonce:
                C.memmove(buf, data[16:], 128)
                return buf
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no communicator id of this launch at {path} after {timeout_s:.0f} s")
        time.sleep(0.02)


class PinnedBuffer:
    """hipHostMalloc'ed fp32 array (wbx_host_alloc): the GPU writes it with plain stores, numpy reads it in place."""

    def __init__(self, n_floats: int):
        self.L = _ffi.lib()
        p = C.c_void_p()
        st = self.L.wbx_host_alloc(n_floats * 4, C.byref(p))
        if st != 0:
            raise _ffi.WbxError(st, "wbx_host_alloc")
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_float * n_floats).from_address(self.ptr))

    def close(self):
        if self.ptr:
            self.array = None
            self.L.wbx_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dist:
    """wbx_dist_* of one rank's context."""

    def __init__(self, ctx, rank: int, world: int, mode: int = REDUCE, id_buf: Optional[C.Array] = None):
        self.L = _ffi.lib()
        self.ctx, self.rank, self.world, self.mode = ctx, rank, world, mode
        if id_buf is None:
            id_buf = exchange_id(rank, world)
        self._check(self.L.wbx_dist_init(ctx.h, id_buf, rank, world, mode), "wbx_dist_init")
        self.active = True

    def _check(self, st, where):
        if st != 0:
            raise _ffi.WbxError(st, where, self.L.wbx_last_error(self.ctx.h).decode())

    @property
    def result_rank(self) -> int:
        """the rank whose wbx_dist_exchange takes the destination: 0, or the last one in chain mode"""
        r = C.c_uint32()
        self._check(self.L.wbx_dist_result_rank(self.ctx.h, C.byref(r)), "wbx_dist_result_rank")
        return r.value

    def allgather(self, payload: bytes, width: int = 64) -> list:
        """every rank's `payload` (padded to `width` bytes), in rank order"""
        send = (C.c_char * width)(*payload[:width].ljust(width, b"\0"))
        recv = (C.c_char * (width * self.world))()
        self._check(self.L.wbx_dist_allgather(self.ctx.h, send, recv, width), "wbx_dist_allgather")
        raw = bytes(recv)
        return [raw[i * width:(i + 1) * width].rstrip(b"\0") for i in range(self.world)]

    def info(self) -> dict:
        """What the exchange saw: RCCL's world size, the device of every rank (PCI bus ids, rank order), the mode, the
        average time of one exchange on its stream (call after sync)."""
        rank, world, mode = C.c_uint32(), C.c_uint32(), C.c_int()
        self._check(self.L.wbx_dist_info(self.ctx.h, C.byref(rank), C.byref(world), C.byref(mode)), "wbx_dist_info")
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.L.wbx_dist_exchange_time(self.ctx.h, C.byref(ms), C.byref(n)), "wbx_dist_exchange_time")
        pci = self.ctx.device_info()["pci"].encode()
        devices = [d.decode() for d in self.allgather(pci, 32)]
        return {"rank": rank.value, "world": world.value, "mode": ("reduce", "ordered", "chain")[mode.value],
                "devices": devices, "exchange_ms_avg": ms.value, "exchanges": n.value, "result_rank": self.result_rank}

    def exchange(self, dst_ptr: Optional[int]):
        self._check(self.L.wbx_dist_exchange(self.ctx.h, dst_ptr), "wbx_dist_exchange")

    def sync(self):
        self._check(self.L.wbx_dist_sync(self.ctx.h), "wbx_dist_sync")

    def barrier(self):
        self._check(self.L.wbx_dist_barrier(self.ctx.h), "wbx_dist_barrier")

    def max(self, value: float) -> float:
        v = C.c_double(value)
        self._check(self.L.wbx_dist_max(self.ctx.h, C.byref(v)), "wbx_dist_max")
        return v.value

    def shutdown(self):
        if self.active:
            self.active = False
            self._check(self.L.wbx_dist_shutdown(self.ctx.h), "wbx_dist_shutdown")
            if self.rank == 0 and self.world > 1:
                try:
                    os.remove(rendezvous_path())
                except OSError:
                    pass
