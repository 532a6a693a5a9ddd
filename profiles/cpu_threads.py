import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
for n in (16, 32, 64, 128):
    os.cpu_count_orig = os.cpu_count
    torch.set_num_threads(n)
    import types
    # monkeypatch cores
    bench.os = types.SimpleNamespace(cpu_count=lambda n=n: n, path=os.path, environ=os.environ)
    r = bench.cpu_baseline(32, 720, 1280, 1, "mpv")
    print(n, r["value"], r["sample"], flush=True)
