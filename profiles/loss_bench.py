"""Per-kernel timing of the looping loss (HIP events, warm, averaged) at 720p and the native crop; run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoloop3d_amd import synth
from videoloop3d_amd.utils_vid import _nn_and_fold, find_nn_indices
import __graft_entry__ as g
g.build()
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (H, W) in ((720, 1280), (180, 320)):
    x = synth.make_video(52, H, W, seed=3, device=dev)
    y = synth.make_video(75, H, W, seed=4, device=dev)
    for name, (ps, s, alpha) in {"ref": (11, 4, 0.0), "other": (3, 2, None)}.items():
        h = (H - ps) // s * s + ps; w = (W - ps) // s * s + ps
        xs, ys = x[..., :h, :w], y[..., :h, :w]
        t_nn = timeit(lambda: find_nn_indices(xs, ys, ps, 3, s, 1, alpha))
        t_all = timeit(lambda: _nn_and_fold(xs, ys, ps, 3, s, 1, alpha, True))
        print(f"{H}x{W} {name}: patchnn(+transposes) {t_nn:.3f} ms, fold {t_all - t_nn:.3f} ms, total {t_all:.3f} ms")
