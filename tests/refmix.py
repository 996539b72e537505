"""Drive the reference's own per-sample code (oracle/_ref: Sampler::stream, apply_gain, abs-max,
AudioBuffer::mix, clamp) over a SessionSpec, with the block SEQUENCING supplied by the oracle's
restated sequencer (segment log).  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

import oracle_ffi as O


class RefMixer:
    def __init__(self, spec):
        self.R = O.ref()
        assert self.R is not None
        self.spec = spec
        n = len(spec.samples)
        self.data = [spec.sample_data(i) for i in range(n)]
        self.fmt = (C.c_int * n)(*[O.FMT[s.fmt] for s in spec.samples])
        self.chs = (C.c_uint32 * n)(*[s.channels for s in spec.samples])
        self.rate = (C.c_uint32 * n)(*[s.rate for s in spec.samples])
        self.cnt = (C.c_size_t * n)(*[s.frames for s in spec.samples])
        self._pp = [O.void_ptrs(d) for d in self.data]
        self.planar = (O.c_voidpp * n)(*[C.cast(p, O.c_voidpp) for p in self._pp])
        self.bus = (C.c_int * spec.n_tracks)(*(spec.track_bus if spec.track_bus is not None else [-1] * spec.n_tracks))

    def block(self, seglog, gains: np.ndarray, clamp=True):
        """seglog: OracleEngine.seglog() of this block; gains [T][C] fp32.  Returns master, buses, peaks, ends."""
        sp = self.spec
        T, Cc, F = sp.n_tracks, sp.channels, sp.block
        segs = (O.RefSegment * max(1, len(seglog)))()
        for i, (t, ds, ln, off, spd, g, smp) in enumerate(seglog):
            segs[i] = O.RefSegment(spd, off, t, ds, ln, g, smp)
        out = [np.zeros(F, np.float32) for _ in range(Cc)]
        bus = np.zeros((sp.n_buses, Cc, F), np.float32) if sp.n_buses else None
        peaks = np.zeros((T, Cc), np.float32)
        ends = np.zeros(max(1, len(seglog)), np.float64)
        gains = np.ascontiguousarray(gains, np.float32)
        self.R.ref_mix_block(T, Cc, F, segs, len(seglog), self.fmt, self.chs, self.rate, self.cnt, self.planar,
                             gains.ctypes.data_as(O.c_f32p), self.bus if sp.n_buses else None, sp.n_buses,
                             O.planar_ptrs(out), bus.ctypes.data_as(O.c_f32p) if bus is not None else None,
                             peaks.ctypes.data_as(O.c_f32p), ends.ctypes.data_as(C.POINTER(C.c_double)), int(clamp))
        return np.stack(out), bus, peaks, ends[:len(seglog)]
