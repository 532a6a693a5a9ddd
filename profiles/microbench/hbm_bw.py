"""Achievable HBM bandwidth on this box with plain torch kernels (read-only, write-only, copy), 24 GB working set
(the size of the cfg3 plane stack).  python profiles/microbench/hbm_bw.py"""
import torch
n = 6 * 1024**3 // 1  # 6 Gi floats = 24 GiB
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = n * 4 / 1e9
ms = t(lambda: x.sum());            print(f"read  (sum)    {ms:7.3f} ms  {gb/ms:6.2f} TB/s")
ms = t(lambda: y.fill_(1.0));       print(f"write (fill)   {ms:7.3f} ms  {gb/ms:6.2f} TB/s")
ms = t(lambda: y.copy_(x));         print(f"copy  (r+w)    {ms:7.3f} ms  {2*gb/ms:6.2f} TB/s (sum of both directions)")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add   (r+w)    {ms:7.3f} ms  {2*gb/ms:6.2f} TB/s")
