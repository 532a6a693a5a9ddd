#!/usr/bin/env python3
"""Round-2 golden vectors, generated like G1..G10 by IMPORTING the reference (tests/golden/make_golden.py holds the stand-ins
for the three absent third-party modules and the recipe).  Runs only in the build container.

G12  helpers next to the path: utils_mpi.gen_mpi_vertices (utils_mpi.py:80-89), utils_vid.Patch3DMSE / Patch3DAvg (utils_vid.py:437-445).
G11  the reference's DIRECT loss path on inputs that do NOT fit the patch grid (utils_vid.py:206-229, 265-286): UnfoldNd floors the
     grid, FoldNd writes into the full x.shape -- `Patch3DGPNNDirectLoss` ('gpnn', the parser default) and `FindNNpatchAndMerge`.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    warnings.simplefilter("ignore")
    torch.set_num_threads(8)
    MG._install_standins()
    sys.path.insert(0, MG.REF)
    import utils_vid as V  # noqa  (reference, with the stand-ins)
    from videoloop3d_amd import synth

    x = synth.make_video(9, 18, 21, seed=31)
    y = synth.make_video(12, 18, 21, seed=32)
    out = {"x": x.numpy(), "y": y.numpy()}
    for ps, pt, s, st, al in ((5, 3, 2, 1, 1e10), (3, 2, 2, 2, 0.5), (4, 3, 3, 1, 1e10)):
        key = f"ps{ps}_pt{pt}_s{s}_st{st}_a{al:g}"
        sm, w = V.FindNNpatchAndMerge(x, y, patch_size=ps, patcht_size=pt, stride=s, stridet=st, alpha=al)
        out[key + "_sum"], out[key + "_weight"] = sm.numpy(), w.numpy()
        xr = x.clone().requires_grad_(True)
        L = V.Patch3DGPNNDirectLoss()
        loss = L(xr, y, rou="-2", scaling=0.1, patch_size=ps, patcht_size=pt, stride=s, stridet=st, alpha=al)
        (g,) = torch.autograd.grad(loss, xr)
        out[key + "_loss"], out[key + "_grad"], out[key + "_y2x"] = np.float32(loss.item()), g.numpy(), L.last_y2x.numpy()
    np.savez_compressed(os.path.join(HERE, "g11_direct_anysize.npz"), **out)
    # ---- G12 -----------------------------------------------------------------------------------------------------------
    import utils_mpi as R  # noqa  (reference, as-is)
    K = torch.tensor([[50., 0, 30.5], [0, 52., 19.25], [0, 0, 1]])
    pd = R.make_depths(5, 1.0, 100.0).flip(0)
    g12 = {"K": K.numpy(), "planedepth": pd.numpy(), "verts": R.gen_mpi_vertices(40, 60, K, 5, 7, pd).numpy()}
    xa, ya = synth.make_video(7, 9, 11, seed=41), synth.make_video(10, 9, 11, seed=42)
    g12.update(xa=xa.numpy(), ya=ya.numpy(), mse=np.float32(V.Patch3DMSE(xa, ya).item()), avg=np.float32(V.Patch3DAvg(xa, ya).item()),
               mse_rev=np.float32(V.Patch3DMSE(ya, xa).item()))
    np.savez_compressed(os.path.join(HERE, "g12_helpers.npz"), **g12)
    print("wrote g11_direct_anysize.npz", {k: getattr(v, "shape", None) for k, v in out.items() if k.endswith("_sum")})


if __name__ == "__main__":
    main()
