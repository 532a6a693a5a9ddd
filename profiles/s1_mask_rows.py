#!/usr/bin/env python3
"""Stage-1 iterations (examples/stage1_step.py) with the loop-mask backward in 64 x 16 (variant 0 / 3) or flat 64 x 8 regions (variant 2):
python profiles/s1_mask_rows.py  -> it/s per shape and variant, alternating."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import __graft_entry__ as g
g.build()
import stage1_step as S
from videoloop3d_amd import MPI
_init = MPI.MPMesh.__init__
VAR = [0]
def init(self, *a, **k):
    _init(self, *a, **k)
    self.spec = dataclasses.replace(self.spec, variant=VAR[0])
MPI.MPMesh.__init__ = init
for name, kw in (("native", {}), ("720p_1p1", dict(frame=(720, 1280), crop=(720, 1280), scale=1.1)), ("720p_1p6", dict(frame=(720, 1280), crop=(720, 1280), scale=1.6, crop_aware_adam=False))):
    for r in range(2):
        for v in (0, 2):
            VAR[0] = v
            print(name, "variant", v, "%.1f it/s" % S.run(iters=40, **kw)["iters_per_s"], flush=True)
