#!/usr/bin/env python3
"""In-process A/B of render kernel variants on ONE resident cfg3 stack (same buffers, same clocks, interleaved rounds): separate
processes differ by up to 10 % on the same box (allocation, clock state), which is more than most variants are worth.
  python profiles/ab_inproc.py --variants 0,3 [--rounds 6] [--reps 4] [--stack-scale 1.0] [--reg] [--dtype f32|f16] [--T 50]
Prints per variant the median / min of the forward and backward times (HIP events on the launch stream)."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,3")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--stack-scale", type=float, default=1.0)
ap.add_argument("--reg", action="store_true", help="smoothness regularisers on (the REG backward)")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--D", type=int, default=32)
ap.add_argument("--T", type=int, default=50)
ap.add_argument("--H", type=int, default=720)
ap.add_argument("--W", type=int, default=1280)
a = ap.parse_args()

import __graft_entry__ as ge  # noqa: E402
ge.build()
from videoloop3d_amd import synth  # noqa: E402
from videoloop3d_amd.render import RenderSpec, render_planes, render_planes_with_smoothness  # noqa: E402
from videoloop3d_amd.utils_mpi import compute_homography, make_depths  # noqa: E402

dev = torch.device("cuda:0")
D, T, H, W = a.D, a.T, a.H, a.W
Hs, Ws = int(H * a.stack_scale), int(W * a.stack_scale)
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                           make_depths(D, 1.0, 100.0).flip(0)[None])[0]
shift = torch.tensor([[1.0, 0, (Ws - W) // 2], [0, 1.0, (Hs - H) // 2], [0, 0, 1.0]])     # frame centred in the larger plane (MPV.py:55-56)
homos_d = (shift @ homos).to(dev)
stack = synth.make_plane_stack(D, T, Hs, Ws, seed=2, device=dev, dtype=torch.float16 if a.dtype == "f16" else torch.float32).requires_grad_(True)
g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
variants = [int(v, 0) for v in a.variants.split(",")]
ev = lambda: torch.cuda.Event(enable_timing=True)
res = {v: ([], []) for v in variants}


def once(v):
    spec = RenderSpec.mpv(variant=v)
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    if a.reg:
        rgb, alpha, sums = render_planes_with_smoothness(stack, homos_d, H, W, spec)
        obj = (rgb * g).sum() + 1e-6 * sums.sum()
        e1.record()
        (gs,) = torch.autograd.grad(obj, stack)
    else:
        rgb, alpha = render_planes(stack, homos_d, H, W, spec)
        e1.record()
        (gs,) = torch.autograd.grad(rgb, stack, g)
    e2.record()
    torch.cuda.synchronize()
    once.last = gs
    return e0.elapsed_time(e1), e1.elapsed_time(e2)


g_first = None
for v in variants:
    once(v)
    if g_first is None:
        g_first = once.last.clone() if T <= 4 else None       # (bit comparison of the gradients: small shapes only -- a second 23.6 GB buffer otherwise)
    elif T <= 4:
        print(f"variant {v:#x}: gradient bits equal to variant {variants[0]:#x}: {torch.equal(g_first, once.last)}")
for r in range(a.rounds):
    for v in variants:
        for _ in range(a.reps):
            f, b = once(v)
            res[v][0].append(f)
            res[v][1].append(b)
print(f"D={D} T={T} {H}x{W} stack {Hs}x{Ws} {a.dtype} reg={a.reg}: {a.rounds} rounds x {a.reps} reps, interleaved")
for v in variants:
    f, b = res[v]
    print(f"variant {v:#6x}  fwd median {statistics.median(f):7.3f} min {min(f):7.3f}   bwd median {statistics.median(b):7.3f} min {min(b):7.3f} ms")
