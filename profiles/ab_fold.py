#!/usr/bin/env python3
"""In-process A/B of the LDS-staged vote-fold tile shapes (variant bits 12-15 = shape index + 1; 0 = the library's choice) on
resident 720p clips, both loss configurations: kernel time per shape, outputs compared with shape 1's bit for bit.
  python profiles/ab_fold.py 0,1,2,3,4,5,6 [rounds]        (shape + 100: the same with the dynamic covering loops, variant bit 9)"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from videoloop3d_amd import synth, _lib as L
from videoloop3d_amd.utils_vid import find_nn_indices
shapes = [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4,5").split(",")]   # shape[.rows per batch]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
x = synth.make_video(52, 719, 1279, seed=3, device=dev)
y = synth.make_video(75, 719, 1279, seed=4, device=dev)
for name, (ps, s, al) in {"ref": (11, 4, 0.5), "other": (3, 2, None)}.items():
    nn, desc, xv, yv = find_nn_indices(x, y, ps, 3, s, 1, al)
    base = desc.variant & 0xfff
    outs = [torch.empty((3, desc.Tx, desc.H, desc.W), device=dev) for _ in range(2)]
    w = torch.empty((desc.Tx, desc.H, desc.W), device=dev)
    acc = torch.empty((), dtype=torch.float64, device=dev)
    res, ref = {v: [] for v in shapes}, None
    for r in range(rounds + 1):
        for v in shapes:
            desc.variant = (base & ~0x200) | (int(v) % 100 << 12) | (0x200 if v >= 100 else 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(L.lib().vl3d_vote_fold_robust(desc, L.ptr(yv), L.ptr(nn), L.ptr(xv), L.RHO["barron"], -2.0, 0.1, L.ptr(outs[0]), L.ptr(w),
                                                  L.ptr(outs[1]), L.ptr(acc), L.stream_ptr(dev)), "fold")
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[v].append(e0.elapsed_time(e1))
            else:
                cur = (outs[0].clone(), outs[1].clone(), w.clone(), float(acc))
                if ref is None:
                    ref = cur
                else:
                    same = all(torch.equal(a, b) for a, b in zip(cur[:3], ref[:3]))
                    print(f"{name} shape {v}: outputs {'bit-equal' if same else 'DIFFER'} loss rel diff {abs(cur[3]-ref[3])/abs(ref[3]):.1e}")
    for v in shapes:
        print(f"{name:6s} shape {v}  fold median {statistics.median(res[v]):.3f} min {min(res[v]):.3f} ms")
