import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from whitebox_amd import synth
from whitebox_amd.engine import build_engine
spec = synth.make_session("lv", 6, n_blocks=2, seed=0x1E7)
for variant in ("plain", "add", "add_nolevels"):
    eng = build_engine(spec, max_blocks=2, spare_tracks=2)
    eng.play()
    eng.render(2)
    m, pk, _ = eng.ctx.fetch(peaks=True)
    print(variant, "render1 master max", np.abs(m).max(axis=(1, 2)), "peaks", pk.max(axis=(1, 2)))
    if variant != "plain":
        eng.add_track("late")
    if variant != "add_nolevels":
        print(" levels", eng.levels().max(axis=1))
    eng.render(2)
    m, pk, _ = eng.ctx.fetch(peaks=True)
    print(variant, "render2 master max", np.abs(m).max(axis=(1, 2)), "peaks", pk.max(axis=(1, 2)), "transport", eng.transport())
    print(" levels", eng.levels().max(axis=1))
    eng.close()
