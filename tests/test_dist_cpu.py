"""Multi-GPU path on CPU: row-band plan + one all-gather (gloo, world_size 2 and 3) reproduces the full frame.

The band renderer here is the CPU oracle (tests may use it); what is under test is the product's sharding logic
(videoloop3d_amd/dist.py: split_rows, source_row_range / halo, band_spec, all_gather_frame)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mpi_oracle as MO
from videoloop3d_amd import synth
from videoloop3d_amd.dist import all_gather_frame, band_spec, exchange_halo_grads, halo_overlaps, plan_bands, source_row_range, split_rows
from videoloop3d_amd.render import RenderSpec


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(D=5, T=2, Hs=66, Ws=80, H=60, W=72):
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=3)
    ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
    tar_e = tar_e.clone()
    tar_e[:3, 3] *= 3.0
    homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                               make_depths(D, 1.0, 100.0).flip(0)[None])[0]
    homos = torch.tensor([[1.0, 0, 4.0], [0, 1.0, 3.0], [0, 0, 1.0]]) @ homos
    return stack, homos, (D, T, Hs, Ws, H, W)


def _oracle_spec(spec):
    return MO.RenderSpec(pixel_center=spec.pixel_center, coord_mode=spec.coord_mode, scale=spec.scale, offset=spec.offset,
                         border=spec.border, act_order=spec.act_order, rgb_act=spec.rgb_act, alpha_act=spec.alpha_act)


def _render_band_oracle(stack, homos, band, W, Hs, spec):
    local = stack[:, :, band.src0:band.src1]
    bs = band_spec(spec, band, Hs)
    shift = torch.tensor([[1.0, 0, 0], [0, 1.0, float(band.row0)], [0, 0, 1.0]])     # window=(row0, 0)
    rgb, alpha, _ = MO.render_planes(local, homos @ shift, band.rows, W, _oracle_spec(bs))
    return rgb


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stack, homos, (D, T, Hs, Ws, H, W) = _scene()
        spec = RenderSpec.mpv()
        bands = plan_bands(homos, H, W, Hs, world, spec)
        mine = _render_band_oracle(stack, homos, bands[rank], W, Hs, spec)
        frame = all_gather_frame(mine, bands)
        direct = all_gather_frame(mine, bands, algo="direct")          # all-peers send/recv: the same bytes as the ring
        # the band-major form (no assembly copy) and a row range out of it -- what the band loss reads: own rows + a few of the neighbours'
        from videoloop3d_amd.dist import frame_rows
        parts = all_gather_frame(mine, bands, algo="direct", layout="bands")
        parts_ring = all_gather_frame(mine, bands, algo="ring", layout="bands")
        lo, hi = max(bands[rank].row0 - 3, 0), min(bands[rank].row0 + bands[rank].rows + 4, H)
        chunked = frame if (len(parts) == world and parts[rank].data_ptr() == mine.contiguous().data_ptr()
                            and torch.equal(frame_rows(parts, lo, hi), frame[:, lo:hi]) and torch.equal(frame_rows(parts_ring, 0, H), frame)) else frame + 1
        full, _, _ = MO.render_planes(stack, homos, H, W, _oracle_spec(spec))
        err = float((frame - full).abs().max())
        # (each comparison on its own: a failure says WHICH one)
        flags = torch.tensor([float(frame.shape == full.shape and err <= 2e-5), float(torch.equal(direct, frame)), float(torch.equal(chunked, frame))])
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        if rank == 0:
            out.put((flags.tolist(), err, [b.__dict__ for b in bands]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])      # (8: the node size north_star names -- bands of a few rows, halos spanning several ranks)
def test_row_bands_allgather_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, err, bands = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok[0] == 1.0, ("gathered frame differs from the single-rank render", err, bands)
    assert ok[1] == 1.0, "the all-peers gather differs from the ring"
    assert ok[2] == 1.0, "the all-peers gather in several grouped launches differs from the ring"
    assert sum(b["rows"] for b in bands) == 60 and bands[0]["row0"] == 0


def _grad_worker(rank, world, port, out):
    """every rank: gradient of ITS band w.r.t. ITS stack rows, then dist.exchange_halo_grads -> the single-rank gradient of those rows."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stack, homos, (D, T, Hs, Ws, H, W) = _scene()
        spec = RenderSpec.mpv()
        bands = plan_bands(homos, H, W, Hs, world, spec)
        b = bands[rank]
        g = synth.hash_uniform((T, H, W, 3), seed=9) - 0.5
        full_in = stack.clone().requires_grad_(True)
        full, _, _ = MO.render_planes(full_in, homos, H, W, _oracle_spec(spec))
        (g_full,) = torch.autograd.grad(full, full_in, g)
        local = stack[:, :, b.src0:b.src1].clone().requires_grad_(True)
        shift = torch.tensor([[1.0, 0, 0], [0, 1.0, float(b.row0)], [0, 0, 1.0]])
        rgb, _, _ = MO.render_planes(local, homos @ shift, b.rows, W, _oracle_spec(band_spec(spec, b, Hs)))
        (g_local,) = torch.autograd.grad(rgb, local, g[:, b.row0:b.row0 + b.rows])
        partial_err = float((g_local - g_full[:, :, b.src0:b.src1]).abs().max())      # before the exchange the halo rows are partial
        g_done = exchange_halo_grads(g_local.clone(), bands)
        err = float((g_done - g_full[:, :, b.src0:b.src1]).abs().max())
        # replicas of a shared row must agree bit for bit on all their holders: gather row checksums and compare on the overlaps
        sums = [None] * world
        dist.all_gather_object(sums, (b.src0, g_done.double().sum(dim=(0, 1, 3, 4)).tolist()))
        same = True
        for p, lo, hi in halo_overlaps(bands, rank):
            s0, v = sums[p]
            mine_v = sums[rank][1]
            same &= all(mine_v[r - b.src0] == v[r - s0] for r in range(lo, hi))
        stats = torch.tensor([err, -partial_err, 0.0 if same else 1.0, float(max(len(halo_overlaps(bands, r)) for r in range(world)))])
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        if rank == 0:
            out.put(stats.tolist())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_halo_gradient_exchange_gives_the_single_rank_gradient(world):
    """stack rows inside a parallax halo are replicated on neighbouring ranks, each with a PARTIAL gradient; after ONE
    neighbour exchange every holder has the single-rank gradient of every row it holds, identical bits on all holders -- the
    sharded model can take optimiser steps without the replicas drifting (world 4 on this small frame: halos taller than a band,
    i.e. rows shared by three ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err, neg_partial, differs, max_peers = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err <= 1e-5, err
    assert -neg_partial > 1e-3            # the exchange was needed: some rank's halo rows were incomplete before it
    assert differs == 0.0
    if world == 4:
        assert max_peers >= 2             # some rank exchanges with more than one peer


def test_split_rows_ragged():
    assert split_rows(720, 8) == [(90 * i, 90) for i in range(8)]
    parts = split_rows(10, 4)
    assert [n for _, n in parts] == [3, 3, 2, 2] and parts[-1][0] + parts[-1][1] == 10
    assert all(n > 0 for _, n in split_rows(8, 8))


def test_halo_covers_every_tap():
    """the stack rows a band keeps contain every tap row of every band pixel on every plane (brute force)."""
    stack, homos, (D, T, Hs, Ws, H, W) = _scene()
    spec = RenderSpec.mpv()
    for world in (2, 4, 7):
        for b in plan_bands(homos, H, W, Hs, world, spec):
            xs, ys = MO._homography_source_coords(H, W, homos, spec.pixel_center)
            ty = ys[:, b.row0:b.row0 + b.rows]
            cov = (ty >= 0) & (ty <= Hs - 1)
            if cov.any():
                assert int(torch.floor(ty[cov]).min()) >= b.src0
                assert int(torch.floor(ty[cov]).max()) + 1 <= b.src1 - 1 or b.src1 == Hs
            lo, hi = source_row_range(homos, b.row0, b.rows, W, Hs, spec)
            assert (lo, hi) == (b.src0, b.src1)


def test_band_spec_requires_affine():
    from videoloop3d_amd.dist import Band
    with pytest.raises(RuntimeError, match="affine"):
        band_spec(RenderSpec(), Band(0, 0, 4, 0, 8), 16)


def test_loss_band_rows_cover_every_patch_of_the_owned_rows():
    """videoloop3d_amd/dist.py loss_band_rows: the sub-image a rank needs starts on the global patch grid, satisfies the trimming rule,
    contains every patch that covers an owned row, and the owned rows of all ranks tile the frame."""
    from videoloop3d_amd.dist import loss_band_rows
    for H, ps, s in ((179, 11, 4), (719, 11, 4), (65, 3, 2), (40, 5, 5), (23, 7, 1)):
        assert (H - ps) % s == 0
        h_o = (H - ps) // s + 1
        for world in (1, 2, 3, 8):
            rows = split_rows(H, world)
            assert rows[0][0] == 0 and sum(r for _, r in rows) == H
            for row0, n in rows:
                a, b = loss_band_rows(row0, n, H, ps, s)
                assert a % s == 0 and (b - a - ps) % s == 0 and 0 <= a <= row0 and row0 + n <= b <= H
                for eta in (row0, row0 + n - 1):
                    covering = [by for by in range(h_o) if by * s <= eta < by * s + ps]
                    assert covering and all(a <= by * s and by * s + ps <= b for by in covering)
