import sys, os, time, torch
sys.path.insert(0, ".")
from videoloop3d_amd import synth
from videoloop3d_amd import utils_vid as UV
from videoloop3d_amd.utils_vid import Patch3DGPNNLowMemLoss
dev = torch.device("cuda:0")
H, W = 719, 1279
clips = [tuple(int(v) for v in c.split("x")) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [(82, 75), (82, 150)]
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "2"]
for Tx, Ty in clips:
    x = synth.make_video(Tx, H, W, seed=3, device=dev).requires_grad_(True)
    y = synth.make_video(Ty, H, W, seed=4, device=dev)
    for variant in variants:
        UV.KERNEL_VARIANT = int(variant, 0)
        L = Patch3DGPNNLowMemLoss()
        def step():
            loss = L(x, y, macro_block=65, patch_size=11, stride=4, patcht_size=3, stridet=1, rou="-2", scaling=0.1, alpha=0.5)
            loss.backward()
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize()
        print(f"Tx={Tx} Ty={Ty} ref cfg 720p  NN variant {variant}: {5 / (time.perf_counter() - t0):6.1f} it/s")
