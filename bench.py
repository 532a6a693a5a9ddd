#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X (contract: see the task prompt / DESIGN.md §Measurement).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

step       = fused render forward + backward of the cfg3 workload (D=32 planes, T=50 frames, 720p; BASELINE.json
             configs[2], the configuration the metric is quoted on; 47 GB of stack+grad, fits one GPU).
metric     = rendered Mpix/s (fwd+bwd): T*H*W output pixels per step / step time, whole job over all N GPUs.
N > 1      = the SAME frame is split into N row bands (strong scaling, north star "tiles shard across the 8 GPUs");
             each rank owns the stack rows its band touches, ONE RCCL all-gather of the composited band per step
             (overlapped with the backward on a side stream), no collective on the gradient path.
Also reported on the same JSON line: roofline (HIP-event kernel times vs algorithmic bytes), cpu_baseline (the CPU oracle
timed on this box's host cores on a bounded sample), and the stage-2 looping-loss iters/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (measured here: 6.4 TB/s read-only and one-shot copy streams, profiles/microbench)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--D", type=int, default=32)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--H", type=int, default=720)
    ap.add_argument("--W", type=int, default=1280)
    ap.add_argument("--spec", default="mpv", choices=["mpv", "utils_mpi"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--stack-dtype", default="f32", choices=["f32", "f16"],
                    help="storage type of the plane stack (f16 = cfg5 of BASELINE.json; arithmetic is fp32 either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-loss", action="store_true")
    ap.add_argument("--no-stage2", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--loss-steps", type=int, default=5)
    ap.add_argument("--gather-algo", default="auto", choices=["auto", "ring", "direct"],
                    help="N > 1: RCCL all_gather (ring) or grouped all-peers send/recv (direct), videoloop3d_amd.dist.all_gather_frame")
    ap.add_argument("--config", default=None, choices=["cfg3", "cfg4", "cfg5"],
                    help="BASELINE.json shorthand: cfg3 = D32 T50 720p fp32 (default shapes), cfg4 = D64 T80 1080p fp32 (8 GPUs), "
                         "cfg5 = D96 T120 2160p fp16 stack (8 GPUs); overrides --D/--T/--H/--W/--stack-dtype")
    ap.add_argument("--no-exchange-halo-grads", dest="exchange_halo_grads", action="store_false",
                    help="N > 1: leave out the sum of the replicated halo rows' gradient with the neighbours (on by default: the step is "
                         "training-complete, every replica of a stack row ends the step with the single-GPU gradient)")
    ap.add_argument("--no-loss-band", dest="loss_band", action="store_false",
                    help="N > 1: leave out the extra (untimed) leg, the looping loss on this rank's rows of the gathered frame")
    a = ap.parse_args()
    if a.config:
        a.D, a.T, a.H, a.W, a.stack_dtype = {"cfg3": (32, 50, 720, 1280, "f32"), "cfg4": (64, 80, 1080, 1920, "f32"),
                                              "cfg5": (96, 120, 2160, 3840, "f16")}[a.config]
    return a


def cpu_baseline(D, H, W, frames, spec_name, passes=3):
    """The CPU oracle (port of the reference's PyTorch CPU path, pinned to the reference goldens) on a bounded sample:
    `frames` frames of the same D/720p workload, fwd+bwd, all host cores."""
    from oracle import mpi_oracle as MO
    from videoloop3d_amd import synth
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths
    # threads: the oracle's index/gather-heavy torch ops peak at 32 threads on the 2x64-core EPYC host of the GPU box
    # (profiles/cpu_threads.py: 16 -> 0.18, 32 -> 0.20, 64 -> 0.18, 128 -> 0.12, 256 -> 0.035 Mpix/s)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    stack = synth.make_plane_stack(D, frames, H, W, seed=2).requires_grad_(True)
    ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
    homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None],
                               torch.tensor([0., 0., 1.]).expand(1, D, 3), make_depths(D, 1.0, 100.0).flip(0)[None])[0]
    ospec = MO.RenderSpec() if spec_name == "utils_mpi" else MO.RenderSpec(pixel_center=0.5, coord_mode="affine",
                                                                            border="hardcut", act_order="post")
    g = synth.hash_uniform((frames, H, W, 3), seed=5) - 0.5

    def one_pass(st, gg):
        rgb, _, _ = MO.render_planes(st, homos, H, W, ospec)
        (gs,) = torch.autograd.grad(rgb, st, gg)
        return gs

    one_pass(stack[:, :1].detach().requires_grad_(True), g[:1])            # warm-up (thread pool, allocator), one frame, untimed
    times = []
    for _ in range(passes):
        t0 = time.perf_counter()
        one_pass(stack, g)
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {"value": frames * H * W / dt / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"{frames} frame(s) of the D={D} {H}x{W} workload, fwd+bwd, torch CPU fp32 oracle; mean of {passes} timed passes "
                      f"after a warm-up ({', '.join(f'{t:.1f}' for t in times)} s; {cores} threads of {os.cpu_count()} logical cores)"}


def loss_bench(dev, H, W, T, Ty, steps):
    """stage-2 looping-loss iters/s: one iter = NN search + vote-fold + robust mean + backward to x, both shipped cfgs."""
    from videoloop3d_amd import synth
    from videoloop3d_amd.utils_vid import Patch3DGPNNLowMemLoss, PreparedClip
    import warnings
    x = synth.make_video(T + 2, H, W, seed=3, device=dev).requires_grad_(True)
    y = synth.make_video(Ty, H, W, seed=4, device=dev)
    # the captured clip is constant training data: its layout change for the NN search is done once, outside the iterations, as the
    # stage-2 dataset does per pyramid level (videoloop3d_amd/train_3dvid.py MVVidPatchDataset); x, the render, is rewritten every time
    yp = PreparedClip(y).crop(0, 0)
    cfgs = {"ref": dict(macro_block=65, patch_size=11, stride=4, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=0),     # swd_alpha_ref = 0, as shipped (configs/mpv_base.txt:52)
            "other": dict(macro_block=65, patch_size=3, stride=2, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=10000)}
    out = {}
    for name, cfg in cfgs.items():
        lm = Patch3DGPNNLowMemLoss()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for it in range(steps + 1):
                if it == 1:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                loss = lm(x, y, y_prepared=yp, **cfg)
                (gx,) = torch.autograd.grad(loss, x)
            torch.cuda.synchronize()
        it_s = (time.perf_counter() - t0) / steps
        out[name] = {"iters_per_s": 1.0 / it_s, "loss": float(loss.detach())}
        # ... and as MPMeshVid.forward runs it (MPV.py:484-507 -> videoloop3d_amd/MPV.py _LoopPrologue): the render's NHWC output goes through the
        # loop padding ONCE, which writes the loss's video and the search's gram16 form of x in the same pass -- no video_to_gram16_k of x.
        # One iteration = prologue + search + fold / loss + their backward down to the NHWC frames.
        from videoloop3d_amd.MPV import _LoopPrologue
        from videoloop3d_amd.utils_vid import PreparedX
        frames = x.detach()[0, :, :T].permute(1, 2, 3, 0).contiguous().requires_grad_(True)          # [T,H,W,3], what the render returns
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for it in range(steps + 1):
                if it == 1:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                xx, xg = _LoopPrologue.apply(frames, None, 2, True)
                loss_p = lm(xx, y, y_prepared=yp, x_prepared=PreparedX(xg, T + 2, H, W), **cfg)
                (gf,) = torch.autograd.grad(loss_p, frames)
            torch.cuda.synchronize()
        out[name]["prepared"] = {"iters_per_s": steps / (time.perf_counter() - t0), "loss": float(loss_p.detach()),
                                 "what": "loop padding (writes x and its gram16 form) + search + fold / loss + backward to the NHWC frames: the loss side of MPMeshVid.forward"}
        # roofline of the NN search (K3, the dominant kernel of the loss): HIP events around the search alone
        from videoloop3d_amd.utils_vid import find_nn_indices, fit_patch
        ps, st_, pt = cfg["patch_size"], cfg["stride"], cfg["patcht_size"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            h_, w_ = fit_patch(H, "h", ps, st_), fit_patch(W, "w", ps, st_)
        xs, ys = x.detach()[..., :h_, :w_], y[..., :h_, :w_]
        al = None if cfg["alpha"] > 100 else cfg["alpha"]
        find_nn_indices(xs, ys, ps, pt, st_, 1, al, y_prepared=yp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            find_nn_indices(xs, ys, ps, pt, st_, 1, al, y_prepared=yp)
        e1.record()
        torch.cuda.synchronize()
        nn_ms = e0.elapsed_time(e1) / 3
        B = ((h_ - ps) // st_ + 1) * ((w_ - ps) // st_ + 1)
        n1, n2, d = T + 2 - pt + 1, Ty - pt + 1, 3 * pt * ps * ps
        ref_flops = 2.0 * B * n1 * n2 * d                                    # as the reference performs them (utils_vid.py:82)
        # what the kernel performs: per 4 locations the frame-pair energies of a ps x (ps + 3 stride) region; the temporal diagonal sum
        # makes the patch distance out of them
        TxP, TyP = -(-(T + 2) // 4) * 4, -(-Ty // 4) * 4
        cells = (B / 4) * ps * (ps + 3 * st_)                                  # (row, column) cells of the regions
        if TxP <= 64 and TyP <= 80:
            # v6, half-precision matrix cores at fp32-class accuracy (csrc/vl3d_loss.hip patchnn6_k): one v_mfma_f32_16x16x32_f16 per TWO cells
            # and 16x16 frame-pair tile, 4 x 5 tiles; 9 of its 16 k slots per cell carry the three split products of the three channels.
            # `achieved` counts the useful multiply-adds (3 channels x the real frame pairs, as one fp32 product each), `issued` what the tiles cost.
            own_flops = 2.0 * cells * 3 * TxP * TyP
            issued = 2.0 * (cells / 2) * 32 * 64 * 80
            PEAK = 256 * 4 * 1024 * 2.4e9 / 1e12                               # f16 MFMA: 1024 flop per clock and SIMD = 2516 TFLOP/s dense
            kern, bound = "patchnn6_k (+ video_to_gram16_k of x; y prepared once per clip)", "mfma-f16 (split hi/lo, three products)"
        else:
            # v4, vector ALUs: (sub, fma) = 3 flop per (cell, channel, frame pair)
            own_flops = issued = 3.0 * cells * 3 * TxP * TyP
            PEAK = 256 * 4 * 16 * 2 * 2.4e9 / 1e12                             # fp32 FMA, one per lane and clock: 78.6 TFLOP/s (157.3 packed)
            kern, bound = "patchnn4_k (+ video_to_pixel_major_k x2)", "valu-fp32"
        out[name]["roofline_nn"] = {"kernel": kern, "bound": bound, "avg_ms": nn_ms,
                                    "flops_as_reference": ref_flops, "achieved_as_reference": ref_flops / (nn_ms * 1e-3) / 1e12,
                                    "flops_performed": own_flops, "flops_issued": issued, "achieved": own_flops / (nn_ms * 1e-3) / 1e12,
                                    "peak": PEAK, "unit": "TFLOP/s", "frac": own_flops / (nn_ms * 1e-3) / 1e12 / PEAK,
                                    "frac_issued": issued / (nn_ms * 1e-3) / 1e12 / PEAK,
                                    "separable_lower_bound_flops": 2.0 * 3 * (T + 2) * Ty * H * W}
        # the whole iteration against HBM on its compulsory bytes: x read by the search and again by the residual, y read
        # by the search and by the vote-fold, the gradient written (y2x / weight stay in registers unless a caller asks for them)
        comp = 4.0 * H * W * (3 * (T + 2) * 3 + 3 * Ty * 2)
        out[name]["roofline_loss"] = {"bound": "hbm", "compulsory_bytes": comp, "ms_per_iter": it_s * 1e3, "achieved": comp / it_s / 1e9,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": comp / it_s / 1e9 / HBM_PEAK_GBS}
    out["compulsory_bytes"] = 4.0 * H * W * (3 * (T + 2) * 3 + 3 * Ty * 2)
    out["shape"] = f"x[1,3,{T + 2},{H},{W}] y[1,3,{Ty},{H},{W}]"
    return out


def main():
    a = parse()
    # ONE JSON line on stdout: whatever the legs' own code prints (pyramid / dataset messages of the drivers) goes to stderr
    out, sys.stdout = sys.stdout, sys.stderr
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    # VL3D_BENCH_BACKEND=gloo is a debugging aid: lets N ranks share one GPU to exercise the N>1 code path on a 1-GPU box
    backend = os.environ.get("VL3D_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from videoloop3d_amd import synth
    from videoloop3d_amd.dist import all_gather_frame, exchange_halo_grads, halo_overlaps, plan_bands, render_band
    from videoloop3d_amd.render import RenderSpec, render_planes
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths

    D, T, H, W = a.D, a.T, a.H, a.W
    Hs, Ws = H, W
    spec = RenderSpec.mpv(variant=a.variant) if a.spec == "mpv" else RenderSpec(variant=a.variant)
    ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
    homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None],
                               torch.tensor([0., 0., 1.]).expand(1, D, 3), make_depths(D, 1.0, 100.0).flip(0)[None])[0]
    homos_d = homos.to(dev)

    # ---- inputs resident in HBM before the timed region ------------------------------------------------------------------
    if world == 1:
        stack = synth.make_plane_stack(D, T, Hs, Ws, seed=2, device=dev,
                                       dtype=torch.float16 if a.stack_dtype == "f16" else torch.float32).requires_grad_(True)
        g_rgb = (synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5)
        band = None
    else:
        bands = plan_bands(homos, H, W, Hs, world, spec)
        band = bands[rank]
        full_rows = band.src1 - band.src0
        sdt = torch.float16 if a.stack_dtype == "f16" else torch.float32
        stack = torch.empty((D, T, full_rows, Ws, 4), dtype=sdt, device=dev)
        per_plane = T * Hs * Ws * 4
        for d in range(D):      # same values as the single-GPU stack (fp32 hash, then the storage type), rows [src0,src1) only
            for t in range(T):
                off = d * per_plane + (t * Hs + band.src0) * Ws * 4
                sl = synth.hash_uniform((full_rows, Ws, 4), 2, device=dev, offset=off) * 4.0 - 2.0
                sl[..., 3] -= 2.0
                stack[d, t] = sl.to(sdt)
        stack.requires_grad_(True)
        g_full_off = lambda t: (t * H + band.row0) * W * 3
        g_rgb = torch.stack([synth.hash_uniform((band.rows, W, 3), 5, device=dev, offset=g_full_off(t)) for t in range(T)]) - 0.5
        comm_stream = torch.cuda.Stream(device=dev)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    fwd_ms, bwd_ms = [], []

    last = {}
    gather_algo = [a.gather_algo]

    def step(timed):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        if band is None:
            rgb, alpha = render_planes(stack, homos_d, H, W, spec)
        else:
            rgb, alpha = render_band(stack, homos_d, band, W, Hs, spec)
        e1.record()
        frame = rgb.detach()
        if band is not None:
            comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm_stream):
                if backend == "nccl":
                    try:
                        frame = all_gather_frame(rgb.detach(), bands, algo=gather_algo[0])
                    except RuntimeError as ex:      # an all-peers gather this RCCL build refuses: every rank sees the same error -> ring
                        if gather_algo[0] == "ring":
                            raise
                        print(f"[bench] rank {rank}: gather algo {gather_algo[0]!r} failed ({ex}); falling back to RCCL's all_gather", file=sys.stderr)
                        gather_algo[0] = "ring"
                        frame = all_gather_frame(rgb.detach(), bands, algo="ring")
                else:   # debugging path only (gloo moves host tensors)
                    frame = all_gather_frame(rgb.detach().cpu(), bands, algo=a.gather_algo)
        (gs,) = torch.autograd.grad(rgb, stack, g_rgb)
        if band is not None and a.exchange_halo_grads:
            gs = exchange_halo_grads(gs, bands) if backend == "nccl" else exchange_halo_grads(gs.cpu(), bands).to(dev)
        e2.record()
        if band is not None:
            torch.cuda.current_stream().wait_stream(comm_stream)
        if timed:
            fwd_ms.append((e0, e1))
            bwd_ms.append((e1, e2))
        last["rgb"], last["frame"], last["gs"] = rgb.detach(), frame, gs
        return gs

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # every rank is really there before anything is timed: a collective over the backend that will carry the step's traffic
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        assert dist.get_world_size() == world and int(ones.item()) == world, f"backend {backend} reports {dist.get_world_size()} ranks / sum {ones.item()}, expected {world}"
    for _ in range(a.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- what was gathered is what was rendered: per-band checksums travel beside the frame, and the frame's own checksum is the
    #      same number at every N (band renders are bit-identical to the full render: tests/test_gpu_render.py row-band tests)
    def bits_sum(t):
        # (slice by slice along the first axis: the int64 copy of a 24 GB gradient would not be a small temporary)
        tot = 0
        for part in (t if t.dim() > 1 and t.numel() > (1 << 28) else [t]):
            tot += int(part.contiguous().view(torch.int32).to(torch.int64).sum().item())
        return tot
    frame = last["frame"]
    frame_checksum = bits_sum(frame)
    gather_check = None
    if world > 1:
        mine = bits_sum(last["rgb"])
        sums = [None] * world
        dist.all_gather_object(sums, mine)
        ok = all(bits_sum(frame[:, b.row0:b.row0 + b.rows]) == sums[b.rank] for b in bands)
        okt = torch.tensor([1 if ok else 0], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        gather_check = bool(int(okt.item()))
        assert gather_check, "the gathered frame differs from the rendered bands"

    # the stack gradient over a PARTITION of the stack rows -- rank r's rows from its first row up to the next rank's first row: with the
    # halo exchange every replica of a row holds the sum of all bands' contributions, so the sums over ranks are the N = 1 numbers up to
    # the rounding of that sum's order (a halo texel adds two per-band partial sums where one GPU adds pixel by pixel: a few ulps on
    # those texels -- the fp64 sum and absolute sum agree to ~1e-7 relative; rows no band touches have gradient 0 at N = 1)
    def f64_sums(t):
        tot, tota = 0.0, 0.0
        for part in (t if t.dim() > 1 and t.numel() > (1 << 28) else [t]):
            pd = part.double()
            tot += float(pd.sum().item())
            tota += float(pd.abs().sum().item())
        return tot, tota
    gs_last = last.pop("gs")
    if world == 1:
        grad_sums = f64_sums(gs_last)
    elif a.exchange_halo_grads:
        hi = bands[rank + 1].src0 if rank + 1 < world else band.src1
        tg = torch.tensor(f64_sums(gs_last[:, :, :max(0, hi - band.src0)]), dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tg)
        grad_sums = (float(tg[0].item()), float(tg[1].item()))
    else:
        grad_sums = None
    del gs_last
    ms_per_step = dt / a.steps * 1e3
    pix_per_step = T * H * W
    value = pix_per_step / (dt / a.steps) / 1e6

    f_ms = sum(s.elapsed_time(e) for s, e in fwd_ms) / len(fwd_ms)
    b_ms = sum(s.elapsed_time(e) for s, e in bwd_ms) / len(bwd_ms)
    my_pix = T * (H if band is None else band.rows) * W
    # ALGORITHMIC bytes: fwd 16*D+12 B/pixel-frame, bwd 12 + 16*D (re-read) + 16*D (grad write) B/pixel-frame
    tex = 8 if a.stack_dtype == "f16" else 16          # bytes per stack texel; the gradient has the stack's dtype (include/vl3d.h)
    fwd_bytes = my_pix * (tex * D + 12)
    bwd_bytes = my_pix * (2 * tex * D + 12)

    def roof(name, nbytes, ms):
        ach = nbytes / (ms * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "traffic_source": None, "avg_ms": ms, "algorithmic_bytes": nbytes}
    r_f = roof("render_fwd2x_k", fwd_bytes, f_ms)
    r_b = roof("render_bwd_pair_k (+bwd_plan_k, bwd_owner_table_k, bwd_windows_k)", bwd_bytes, b_ms)
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            tr = json.load(open(pmc))
            key = f"D{D}_T{T}_{H}x{W}_{a.spec}_v{a.variant}"
            if key in tr and world == 1:
                r_f["traffic"], r_b["traffic"] = tr[key].get("fwd"), tr[key].get("bwd")
                # NOT a measurement of this run: the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs) of the same command
                r_f["traffic_source"] = r_b["traffic_source"] = "profiles/pmc_traffic.json (" + str(tr.get("_source", "rocprofv3 --pmc passes, profiles/run_profiles.sh")) + ")"
        except Exception:
            pass
    dominant = r_b if b_ms >= f_ms else r_f

    res = {
        "metric": f"rendered Mpix/s (fwd+bwd) D={D} planes {H}p", "value": value, "unit": "Mpix/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "stack_storage": a.stack_dtype,
        "config": {"workload": f"{a.config or 'cfg3'} stage-2 MPV render fwd+bwd: D={D} planes, T={T} frames, {H}x{W}, "
                               f"{a.spec} convention, plane stack (D,T,H,W,4) {a.stack_dtype} resident in HBM",
                   "parallelism": "single GPU" if world == 1 else f"{world} row bands + 1 all-gather of the composited frame",
                   "variant": a.variant},
        "roofline": dominant, "roofline_fwd": r_f, "roofline_bwd": r_b,
        "fwd_bwd_algorithmic_frac": (fwd_bytes + bwd_bytes) / ((f_ms + b_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "frame_checksum": frame_checksum,       # sum of the frame's bit patterns: identical at N = 1, 2, 4, 8
        "grad_sums": grad_sums,                 # (sum, sum of absolute values) of the stack gradient in fp64, after the halo exchange: the N = 1 numbers to ~1e-7
        # what this rank keeps in HBM for the step: its rows of the stack, the gradient of the same size and type, frame-sized buffers
        "resident_bytes_per_rank": {"stack": stack.numel() * stack.element_size(), "grad_stack": stack.numel() * stack.element_size(),
                                    "frames": int(last["frame"].numel() * 4 + g_rgb.numel() * 4 + last["rgb"].numel() * 4),
                                    "torch_peak_allocated": int(torch.cuda.max_memory_allocated(dev))},
    }
    if world > 1:
        band_bytes = T * max(b.rows for b in bands) * W * 3 * 4
        algo = gather_algo[0] if gather_algo[0] != "auto" else ("direct" if (world > 2 and backend == "nccl") else "ring")
        ov = halo_overlaps(bands, rank)
        res["collective"] = {
            "op": "all-gather of the composited bands", "algo": algo, "verified": gather_check, "bytes_per_rank_sent": band_bytes * (world - 1) if algo == "direct" else band_bytes,
            "frame_bytes": T * H * W * 3 * 4,
            # xGMI: point to point, ~153 GB/s per link and direction, 7 links per GPU (MI355X_MICROARCH.md)
            "expected_ms_ring": (world - 1) * band_bytes / 153e9 * 1e3, "expected_ms_direct": band_bytes / 153e9 * 1e3,
            "halo_grad_exchange": {"in_step": bool(a.exchange_halo_grads), "peers": len(ov),
                                   "bytes_sent": sum((hi - lo) for _, lo, hi in ov) * D * T * Ws * 16},
        }
    if world > 1 and a.loss_band:
        # the looping loss on this rank's rows of the gathered frame (dist.looping_loss_band): halo patch rows recomputed, one scalar
        # all-reduce; x = the gathered frame (T + 2 loop-padded frames), y = a synthetic captured clip
        from videoloop3d_amd.dist import looping_loss_band
        import warnings
        fr = last["frame"].to(dev).permute(3, 0, 1, 2)[None]                      # [1,3,T,H,W]
        x = torch.cat([fr, fr[:, :, :2]], 2).contiguous().requires_grad_(True)
        y = synth.make_video(75, H, W, seed=4, device=dev)
        cfg = dict(patch_size=3, stride=2, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=10000)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for it in range(4):
                if it == 1:
                    fence()
                    tl = time.perf_counter()
                ls, n = looping_loss_band(x, y, band.row0, band.rows, **cfg)
                (gx,) = torch.autograd.grad(ls, x)
                tot = torch.tensor([float(ls.detach()), float(n)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(tot)
            fence()
        res["loss_band"] = {"iters_per_s": 3 / (time.perf_counter() - tl), "loss": float(tot[0] / tot[1]), "cfg": "other (ps 3, stride 2), Ty = 75"}
    if rank == 0:
        # the extra legs (loss, stage-1 shape, storage / culling variants, end-to-end iteration, CPU baseline) run at N = 1 only
        if world > 1:
            a.no_loss = a.no_stage2 = a.no_cpu_baseline = True
        if not a.no_loss:
            try:
                res["loss"] = loss_bench(dev, H, W, T, 75, a.loss_steps)
                res["loss_native_crop"] = loss_bench(dev, 180, 320, T, 75, a.loss_steps)
            except Exception as e:   # the headline render number must still be reported
                res["loss"] = {"error": repr(e)}
        if not a.no_stage2:
            try:    # cfg2 of BASELINE.json (stage-1 shape): ONE 720p frame, D=32 -- launch-latency regime, reported beside the headline
                st1 = synth.make_plane_stack(D, 1, Hs, Ws, seed=2, device=dev).requires_grad_(True)
                g1 = synth.hash_uniform((1, H, W, 3), seed=5, device=dev) - 0.5
                for it in range(25):
                    if it == 5:
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                    r1, _ = render_planes(st1, homos_d, H, W, spec)
                    (gs1,) = torch.autograd.grad(r1, st1, g1)
                torch.cuda.synchronize()
                dt1 = (time.perf_counter() - t1) / 20
                res["cfg2_single_frame"] = {"value": H * W / dt1 / 1e6, "unit": "Mpix/s", "ms_per_step": dt1 * 1e3,
                                            "frac": H * W * (48 * D + 24) / dt1 / 1e9 / HBM_PEAK_GBS,      # fwd+bwd algorithmic bytes / wall time / HBM peak
                                            "workload": f"D={D}, T=1, {H}x{W} render fwd+bwd (stage-1 shape)"}
                del st1, gs1
            except Exception as e:
                res["cfg2_single_frame"] = {"error": repr(e)}
            try:    # cfg5's storage format on the cfg3 geometry: fp16 stack AND fp16 gradient (8-byte texels), fp32 arithmetic
                st16 = stack.detach().half().requires_grad_(True)
                for it in range(6):
                    if it == 2:
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                    r1, _ = render_planes(st16, homos_d, H, W, spec)
                    (gs1,) = torch.autograd.grad(r1, st16, g_rgb)
                torch.cuda.synchronize()
                dt16 = (time.perf_counter() - t1) / 4
                res["fp16_stack_storage"] = {"value": T * H * W / dt16 / 1e6, "unit": "Mpix/s", "ms_per_step": dt16 * 1e3,
                                             "frac": T * H * W * (24 * D + 24) / dt16 / 1e9 / HBM_PEAK_GBS,      # 8-byte texels: fwd 8 D + 12, bwd 12 + 16 D
                                             "workload": "cfg3 geometry with the plane stack and its gradient stored as fp16 (cfg5's format); fp32 arithmetic"}
                del st16, r1, gs1
            except Exception as e:
                res["fp16_stack_storage"] = {"error": repr(e)}
            try:    # tile culling (MPI.py:288-442): the same cfg3 render on a stack with ~20 % of its quads kept, with and without
                    # the per-workgroup plane skipping (bit-identical results); in place on the resident stack, last use of it
                from videoloop3d_amd import tiles
                QH, QW = 35, 63                                     # configs/mpi_base.txt:14-15 (36 x 64 vertices)
                qy, qx = torch.meshgrid(torch.arange(QH, device=dev), torch.arange(QW, device=dev), indexing="ij")
                keep = torch.zeros((D, QH, QW), dtype=torch.bool, device=dev)
                for d in range(D):                                  # one coherent blob per plane, ~20 % of its area
                    cy, cx = (7 * d + 3) % QH, (11 * d + 5) % QW
                    keep[d] = ((qy - cy).abs() <= QH // 5) & ((qx - cx).abs() <= QW // 4)
                with torch.no_grad():
                    tiles.cull_stack_(stack, keep)
                times = {}
                # culled: the autograd contract (a dense gradient: the texels no kept quad can read are written as zeros -- 23.6 GB of stores whatever
                # is kept); lean: VL3D_GRAD_CULLED_UNWRITTEN, the contract of the training path (WindowAdam / TileAdam never read those texels)
                for name, qk, lean in (("plain", None, False), ("culled", keep, False), ("lean", keep, True)):
                    for it in range(6):
                        if it == 2:
                            torch.cuda.synchronize()
                            t1 = time.perf_counter()
                        r1, _ = render_planes(stack, homos_d, H, W, spec, quad_keep=qk, grad_culled_unwritten=lean)
                        (gs1,) = torch.autograd.grad(r1, stack, g_rgb)
                    torch.cuda.synchronize()
                    times[name] = (time.perf_counter() - t1) / 4
                    del r1, gs1
                kept = float(keep.float().mean())
                res["tile_culling"] = {"kept_quads": kept, "ms_plain": times["plain"] * 1e3,
                                       "ms_culled": times["culled"] * 1e3, "ms_culled_lean": times["lean"] * 1e3,
                                       "value": T * H * W / times["lean"] / 1e6, "unit": "Mpix/s",
                                       # the kept quads' share of the plain step's algorithmic bytes / time / peak
                                       "frac_kept_bytes": kept * T * H * W * (48 * D + 24) / times["lean"] / 1e9 / HBM_PEAK_GBS,
                                       "frac_kept_bytes_dense_grad": kept * T * H * W * (48 * D + 24) / times["culled"] / 1e9 / HBM_PEAK_GBS,
                                       "workload": "cfg3 render fwd+bwd on a tile-culled stack (one blob of kept quads per plane): plain = without the quad "
                                                   "map, culled = with it and a dense gradient (zeros written for culled texels: the autograd contract), "
                                                   "lean = with it and VL3D_GRAD_CULLED_UNWRITTEN (the training path's contract); value = lean"}
            except Exception as e:
                res["tile_culling"] = {"error": repr(e)}
            try:    # the geometry a shipped stage-2 iteration renders (configs/mpv_base.txt:10-11,33-34): stack stored at 1.1x the frame,
                    # rgb_smooth / a_smooth on -> forward = render + regulariser sums, backward = frame-pair kernel WITH the regularisers
                stack = None
                torch.cuda.empty_cache()
                from videoloop3d_amd.render import render_planes_with_smoothness
                Hs2, Ws2 = int(H * 1.1), int(W * 1.1)
                shift = torch.tensor([[1.0, 0, (Ws2 - W) // 2], [0, 1.0, (Hs2 - H) // 2], [0, 0, 1.0]])          # MPV.py:55-56
                h2 = (shift @ homos).to(dev)
                st2 = synth.make_plane_stack(D, T, Hs2, Ws2, seed=2, device=dev).requires_grad_(True)
                tf, tb = [], []
                for it in range(7):
                    q0, q1, q2 = ev(), ev(), ev()
                    q0.record()
                    r2, _, sums2 = render_planes_with_smoothness(st2, h2, H, W, spec)
                    obj = (r2 * g_rgb).sum() + 1e-6 * sums2.sum()
                    q1.record()
                    (gs2,) = torch.autograd.grad(obj, st2)
                    q2.record()
                    torch.cuda.synchronize()
                    if it >= 2:
                        tf.append(q0.elapsed_time(q1)); tb.append(q1.elapsed_time(q2))
                    del gs2, r2, sums2, obj
                f2, b2 = sum(tf) / len(tf), sum(tb) / len(tb)
                px = T * H * W
                tex = T * Hs2 * Ws2
                res["reference_geometry"] = {
                    "workload": f"cfg3 frames from a {Hs2}x{Ws2} stack (mpi_h/w_scale 1.1) with the smoothness regularisers on: D={D}, T={T}",
                    "value": px / ((f2 + b2) * 1e-3) / 1e6, "unit": "Mpix/s", "fwd_ms": f2, "bwd_ms": b2,
                    "roofline_bwd": {"kernel": "render_bwd_pair_k<REG> (sign words of the forward; + pre-pass)", "bound": "hbm", "avg_ms": b2,
                                     "achieved": px * (32 * D + 12) / (b2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": px * (32 * D + 12) / (b2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes": px * (32 * D + 12),
                                     "texel_footprint_bytes": tex * 32 * D + px * 12,      # every texel of the 1.1x stack is read and its gradient written
                                     "frac_texel_footprint": (tex * 32 * D + px * 12) / (b2 * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    "roofline_fwd": {"kernel": "render_fwd_reg_k (render + regulariser sums + sign words in one pass; + reg_masks_k, reg_flags_k, reg_slot_fwd_k)", "bound": "hbm", "avg_ms": f2,
                                     "achieved": px * (16 * D + 12) / (f2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": px * (16 * D + 12) / (f2 * 1e-3) / 1e9 / HBM_PEAK_GBS}}
                del st2
            except Exception as e:
                res["reference_geometry"] = {"error": repr(e)}
            try:    # end-to-end stage-2 iterations on the drop-in module (render crop + looping loss + fused regularisers + Adam)
                stack = None
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "examples"))
                import stage2_step
                res["stage2_step"] = stage2_step.run(iters=10, dev=str(dev))
                res["stage2_step"]["reference_estimate"] = "0.4-0.9 it/s on the authors' GPU (BASELINE.md, derived from README wall times)"
            except Exception as e:
                res["stage2_step"] = {"error": repr(e)}
            try:    # texture memory of tile-culled models in the packed form (videoloop3d_amd/packed.py), from the block tables alone
                from videoloop3d_amd.packed import PackedLayout
                fp = {}
                for name, (D_, T_, Hs_, Ws_) in {"cfg3_720p": (32, 50, 720, 1280), "cfg3_training_stack_396x704": (32, 50, 396, 704),
                                                 "cfg5_band_of_8_fp32": (96, 120, 311, 3840)}.items():
                    QH, QW = 35, 63
                    qy, qx = torch.meshgrid(torch.arange(QH, device=dev), torch.arange(QW, device=dev), indexing="ij")
                    keep = torch.zeros((D_, QH, QW), dtype=torch.bool, device=dev)
                    for d in range(D_):
                        cy, cx = (7 * d + 3) % QH, (11 * d + 5) % QW
                        keep[d] = ((qy - cy).abs() <= QH // 5) & ((qx - cx).abs() <= QW // 4)
                    blob = ((qy - QH // 2).abs() <= QH // 4) & ((qx - QW // 2).abs() <= QW // 4)      # the moving part of the scene: one region
                    lay = PackedLayout(keep, keep & blob[None], T_, Hs_, Ws_)
                    fp[name] = {"kept_quads": float(keep.float().mean()), "dynamic_of_kept": float((keep & blob[None]).float().sum() / keep.float().sum()),
                                "dense_bytes": lay.dense_bytes, "packed_bytes": lay.pool_bytes, "fraction": lay.pool_bytes / lay.dense_bytes,
                                "blocks_static": lay.n_static, "blocks_dynamic": lay.n_dynamic}
                    del lay, keep
                res["packed_footprint"] = fp
            except Exception as e:
                res["packed_footprint"] = {"error": repr(e)}
            try:    # BASELINE.json configs[1]: a stage-1 iteration (train_3d.py:189-250: MPMesh.forward + MSE + loop-mask entropy + the four
                    # regularisers of configs/mpi_base.txt:37-40 + Adam), reference-native crop and the full 720p frame
                torch.cuda.empty_cache()
                import stage1_step
                res["stage1_step"] = {"native_crop": stage1_step.run(dev=str(dev)),
                                      "cfg2_720p_frame": stage1_step.run(frame=(720, 1280), crop=(720, 1280), scale=1.1, dev=str(dev)),
                                      # ... and at the SHIPPED plane size (configs/mpi_base.txt:11-12: mpi_h/w_scale = 1.6)
                                      "cfg2_720p_frame_scale1p6": stage1_step.run(frame=(720, 1280), crop=(720, 1280), scale=1.6, dev=str(dev))}
            except Exception as e:
                res["stage1_step"] = {"error": repr(e)}
            try:    # the offline renderer (scripts/script_render_video.py as videoloop3d_amd/render_video.py): eval frames per second at 720p
                torch.cuda.empty_cache()
                import render_video as render_video_example
                res["offline_render"] = render_video_example.run(dev=str(dev))
            except Exception as e:
                res["offline_render"] = {"error": repr(e)}
            try:    # stage 1 END TO END (videoloop3d_amd/train_3d.py on the reference's schedule: 8 views x 9 crops per epoch, sparsify switch-over)
                torch.cuda.empty_cache()
                import stage1_train
                res["stage1_train"] = stage1_train.run(dev=str(dev))
            except Exception as e:
                res["stage1_train"] = {"error": repr(e)}
            try:    # the same iteration on the REFERENCE'S SCHEDULE (train_3dvid.py:22-66, 263-290): 8 views with their own poses, the
                    # {4, 4, 9} crops per view of the last three pyramid levels, shuffled -- a crop's texel window comes back after 32-72
                    # other crops, which is what the crop-aware optimiser's deferral has to live with (examples/stage2_schedule.py)
                torch.cuda.empty_cache()
                import stage2_schedule
                # dense: the optimiser step inside the render backward (vl3d_render_bwd_adam, the default); dense_two_kernels: vl3d_render_bwd + step kernel
                res["stage2_schedule"] = {"dense": stage2_schedule.run(dev=str(dev)), "dense_two_kernels": stage2_schedule.run(dev=str(dev), fused=False),
                                          "tile_culled": stage2_schedule.run(dev=str(dev), sparsify=True),
                                          "tile_exact": stage2_schedule.run(dev=str(dev), sparsify=True, tile_exact=True),
                                          "tile_culled_two_kernels": stage2_schedule.run(dev=str(dev), sparsify=True, fused=False)}
            except Exception as e:
                res["stage2_schedule"] = {"error": repr(e)}
        if not a.no_cpu_baseline:
            stack = None
            torch.cuda.empty_cache()
            res["cpu_baseline"] = cpu_baseline(D, H, W, a.cpu_frames, a.spec)
        else:
            res["cpu_baseline"] = None
        res["build"] = dict(ge.BUILD_INFO)

        def pick(d, *path):
            for k in path:
                if not isinstance(d, dict) or k not in d:
                    return None
                d = d[k]
            return d

        def r4(v):
            return float(f"{v:.4g}") if isinstance(v, (int, float)) and not isinstance(v, bool) else v
        # Every leg in full goes to a side file (per-level arrays, histograms, footprints, workloads); the stdout line carries the headline,
        # `roofline`, `cpu_baseline`, `build` and -- LAST, where a consumer that keeps only the tail of the line still finds it -- a flat
        # `summary` of every leg.  The line stays under 6 KB.
        detail_path = os.environ.get("VL3D_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as fh:
                json.dump(res, fh)
        except OSError as e:
            detail_path = f"(not written: {e})"
        summ = {}
        for cfg_ in ("ref", "other"):
            summ[f"loss720_{cfg_}_it_s"] = pick(res, "loss", cfg_, "iters_per_s")
            summ[f"loss720_{cfg_}_prepared_it_s"] = pick(res, "loss", cfg_, "prepared", "iters_per_s")
            summ[f"loss720_{cfg_}_nn_ms"] = pick(res, "loss", cfg_, "roofline_nn", "avg_ms")
            summ[f"loss720_{cfg_}_hbm_frac"] = pick(res, "loss", cfg_, "roofline_loss", "frac")
            summ[f"loss720_{cfg_}_nn_mfma_frac_issued"] = pick(res, "loss", cfg_, "roofline_nn", "frac_issued")
            summ[f"loss_crop_{cfg_}_it_s"] = pick(res, "loss_native_crop", cfg_, "iters_per_s")
        summ["fwd_ms"], summ["fwd_frac"], summ["fwd_traffic_GB"] = r_f["avg_ms"], r_f["frac"], (r_f["traffic"] or 0) / 1e9 or None
        summ["bwd_ms"], summ["bwd_frac"], summ["bwd_traffic_GB"] = r_b["avg_ms"], r_b["frac"], (r_b["traffic"] or 0) / 1e9 or None
        summ["cfg2_ms"], summ["cfg2_mpix_s"] = pick(res, "cfg2_single_frame", "ms_per_step"), pick(res, "cfg2_single_frame", "value")
        summ["cfg2_frac"] = pick(res, "cfg2_single_frame", "frac")
        summ["fp16_ms"], summ["fp16_mpix_s"], summ["fp16_frac"] = pick(res, "fp16_stack_storage", "ms_per_step"), pick(res, "fp16_stack_storage", "value"), pick(res, "fp16_stack_storage", "frac")
        summ["cull_ms_plain"], summ["cull_ms_culled"], summ["cull_kept"] = pick(res, "tile_culling", "ms_plain"), pick(res, "tile_culling", "ms_culled"), pick(res, "tile_culling", "kept_quads")
        summ["cull_ms_lean"], summ["cull_frac_kept"] = pick(res, "tile_culling", "ms_culled_lean"), pick(res, "tile_culling", "frac_kept_bytes")
        summ["refgeo_mpix_s"], summ["refgeo_fwd_ms"], summ["refgeo_bwd_ms"] = pick(res, "reference_geometry", "value"), pick(res, "reference_geometry", "fwd_ms"), pick(res, "reference_geometry", "bwd_ms")
        summ["refgeo_fwd_frac"], summ["refgeo_bwd_frac"] = pick(res, "reference_geometry", "roofline_fwd", "frac"), pick(res, "reference_geometry", "roofline_bwd", "frac")
        for k_, n_ in (("native_crop", "s1_crop"), ("cfg2_720p_frame", "s1_720p_1p1"), ("cfg2_720p_frame_scale1p6", "s1_720p_1p6")):
            summ[f"{n_}_it_s"] = pick(res, "stage1_step", k_, "iters_per_s")
        summ["render_spiral_fps"], summ["render_fixed_view_fps"] = pick(res, "offline_render", "spiral", "frames_per_s"), pick(res, "offline_render", "fixed_view", "frames_per_s")
        summ["s1_train_epochs_per_min"], summ["s1_train_it_s"] = pick(res, "stage1_train", "epochs_per_min"), pick(res, "stage1_train", "iters_per_s")
        summ["s1_train_140_epochs_s"] = pick(res, "stage1_train", "projected_140_epochs_s")
        for k_ in ("ref", "other", "other_tile_culled", "other_tile_culled_packed"):
            summ[f"s2step_{k_}_it_s"] = pick(res, "stage2_step", k_, "iters_per_s")
        for k_ in ("dense", "dense_two_kernels", "tile_culled", "tile_exact", "tile_culled_two_kernels"):
            summ[f"s2sched_{k_}_it_s"] = pick(res, "stage2_schedule", k_, "iters_per_s")
            summ[f"s2sched_{k_}_iter_frac"] = pick(res, "stage2_schedule", k_, "roofline_iter", "frac")
        for k_ in list(summ):
            if summ[k_] is None:
                del summ[k_]
        errors = {k: v["error"][:120] for k, v in res.items() if isinstance(v, dict) and "error" in v}
        if errors:
            summ["errors"] = errors
        head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "stack_storage", "config", "frame_checksum", "grad_sums", "collective", "loss_band", "roofline", "cpu_baseline", "build"]
        line = {k: res[k] for k in head if k in res}
        line["detail_file"] = os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) else detail_path
        line["summary"] = {k: r4(v) for k, v in summ.items()}
        text = json.dumps(line)
        for k_ in ("build", "collective"):      # never silently exceed what the record keeps: the longest free-text fields go first
            if len(text) > 6000 and k_ in line:
                line[k_] = {"see": "detail_file"}
                text = json.dumps(line)
        print(text, file=out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
