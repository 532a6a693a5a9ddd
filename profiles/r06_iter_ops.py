#!/usr/bin/env python3
"""round 6: which Python line launches each small device kernel of a tile-culled stage-2 iteration (torch profiler with stacks, one iteration)."""
import os, sys, types, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import stage2_schedule as S
from torch.profiler import profile, ProfilerActivity
import videoloop3d_amd.train_3dvid as T
orig = T.run_iter
calls = []
def spy(model, opt, item, args, dev):
    calls.append((model, opt, item, args, dev))
    return orig(model, opt, item, args, dev)
S_run_iter = T.run_iter
T.run_iter = spy
warnings.simplefilter("ignore")
S.run(epochs=1, levels=1, sparsify=True)
T.run_iter = orig
model, opt, item, args, dev = calls[-1]
for _ in range(3): orig(model, opt, item, args, dev)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    orig(model, opt, item, args, dev)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
for e in sorted(evs, key=lambda e: e.time_range.start):
    ks = ", ".join(f"{k.name[:48]} {k.duration:.1f}us" for k in e.kernels)
    st = [f for f in (e.stack or []) if "videoloop3d_amd" in f or "examples" in f][:2]
    print(f"{e.name[:40]:40s} | {ks[:110]:110s} | {' <- '.join(s.split('/')[-1][:60] for s in st)}")
