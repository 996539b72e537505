"""Load the committed golden vectors (tests/golden/, produced by oracle/gen_golden.py from the
reference's own translation units).  TEST INFRASTRUCTURE."""
import json
import os

import numpy as np

from whitebox_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLDEN, "sessions.json")) as f:
        return json.load(f)


def session_names():
    return sorted(manifest()["sessions"].keys())


def load_session(name):
    """-> (SessionSpec, n_blocks, dict of golden arrays)"""
    m = manifest()
    kw = dict(m["sessions"][name])
    spec = synth.make_session(name, kw.pop("n_tracks"), n_blocks=m["n_blocks"], **kw)
    g = dict(np.load(os.path.join(GOLDEN, f"session_{name}.npz")))
    return spec, m["n_blocks"], g


def golden_segments(g, block):
    """rows of the fixture's segment table for one block:
    (track, dst_start, len, sample_offset_bits, speed_bits, gain_bits, sample, end_offset_bits)"""
    s = g["segs"]
    return [tuple(int(v) for v in r[1:]) for r in s if int(r[0]) == block]
