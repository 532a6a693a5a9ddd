import sys, torch, time
sys.path.insert(0, ".")
from videoloop3d_amd import tiles
dev = torch.device("cuda:0")
p0 = torch.randn(32, 50, 396, 704, 4, device=dev)
def timeit(opt, p):
    p.grad = torch.randn_like(p)
    for _ in range(2): opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): opt.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
pa = p0.clone().requires_grad_(True)
print("torch fused Adam  %.2f ms" % timeit(torch.optim.Adam([pa], lr=1e-3, eps=6e-8, fused=True), pa))
del pa; torch.cuda.empty_cache()
pb = p0.clone().requires_grad_(True)
print("torch Adam (foreach) %.2f ms" % timeit(torch.optim.Adam([pb], lr=1e-3, eps=6e-8), pb))
del pb; torch.cuda.empty_cache()
pc = p0.clone().requires_grad_(True)
print("TileAdam dense    %.2f ms" % timeit(tiles.TileAdam([pc], lr=1e-3, eps=6e-8), pc))
