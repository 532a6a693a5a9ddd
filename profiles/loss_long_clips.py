import sys, os, time, torch
sys.path.insert(0, ".")
from videoloop3d_amd import synth
from videoloop3d_amd.utils_vid import Patch3DGPNNLowMemLoss
dev = torch.device("cuda:0")
H, W = 719, 1279
for Tx, Ty in ((82, 75), (82, 150)):
    x = synth.make_video(Tx, H, W, seed=3, device=dev).requires_grad_(True)
    y = synth.make_video(Ty, H, W, seed=4, device=dev)
    for variant in ("0", "2"):
        os.environ["VL3D_LOSS_VARIANT"] = variant
        L = Patch3DGPNNLowMemLoss()
        def step():
            loss = L(x, y, macro_block=65, patch_size=11, stride=4, patcht_size=3, stridet=1, rou="-2", scaling=0.1, alpha=0.5)
            loss.backward()
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize()
        print(f"Tx={Tx} Ty={Ty} ref cfg 720p  NN variant {variant}: {5 / (time.perf_counter() - t0):6.1f} it/s")
