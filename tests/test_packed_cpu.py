"""Host logic of the packed storage of tile-culled models (videoloop3d_amd/packed.py): block table, pack / unpack round trip, sizes."""
import torch

from videoloop3d_amd import tiles
from videoloop3d_amd.packed import TS, PackedLayout


def _maps(D, QH, QW, seed=3):
    g = torch.Generator().manual_seed(seed)
    keep = torch.rand((D, QH, QW), generator=g) < 0.3
    dyn = keep & (torch.rand((D, QH, QW), generator=g) < 0.5)
    return keep, dyn


def test_block_table_covers_exactly_the_texels_kept_quads_can_read():
    D, T, Hs, Ws = 4, 3, 45, 70
    keep, dyn = _maps(D, 5, 7)
    lay = PackedLayout(keep, dyn, T, Hs, Ws)
    keep_t, dyn_t = tiles.quad_to_texel_mask(keep, Hs, Ws), tiles.quad_to_texel_mask(dyn, Hs, Ws)
    assert lay.blocks.shape == (D, -(-Hs // TS), -(-Ws // TS))
    present = (lay.blocks >= 0)
    up = lambda b: b.repeat_interleave(TS, 1).repeat_interleave(TS, 2)[:, :Hs, :Ws]
    assert bool((up(present) | ~keep_t).all())                                  # every texel a kept quad can read has storage
    assert bool((up(present & ((lay.blocks & 1) == 1)) | ~dyn_t).all())         # ... per frame where a dynamic quad can read it
    # no block is stored in vain: a present block holds a kept texel, a dynamic block a dynamic one
    pad = lambda m: torch.nn.functional.pad(m.float(), (0, lay.blocks.shape[2] * TS - Ws, 0, lay.blocks.shape[1] * TS - Hs))
    anyb = lambda m: pad(m).reshape(D, lay.blocks.shape[1], TS, lay.blocks.shape[2], TS).amax(dim=(2, 4)) > 0
    assert torch.equal(present, anyb(keep_t)) and torch.equal(present & ((lay.blocks & 1) == 1), anyb(dyn_t))
    # slots: static blocks one, dynamic blocks T, consecutive and disjoint
    slots = []
    for e in lay.blocks.flatten().tolist():
        if e >= 0:
            slots += list(range(e >> 1, (e >> 1) + (T if e & 1 else 1)))
    assert sorted(slots) == list(range(lay.n_slots)) and lay.n_slots == lay.n_static + T * lay.n_dynamic
    assert lay.pool_bytes == lay.n_slots * 1024 and lay.dense_bytes == D * T * Hs * Ws * 16


def test_pack_unpack_round_trip_and_frame_subsets():
    D, T, Hs, Ws = 3, 4, 37, 52
    keep, dyn = _maps(D, 4, 6, seed=5)
    stack = torch.randn(D, T, Hs, Ws, 4)
    keep_t, dyn_t = tiles.quad_to_texel_mask(keep, Hs, Ws), tiles.quad_to_texel_mask(dyn, Hs, Ws)
    # the dense model's invariant: static texels identical in every frame (the dense path keeps them so)
    st = (keep_t & ~dyn_t)[:, None, :, :, None].expand_as(stack)
    stack = torch.where(st, stack[:, :1].expand_as(stack), stack)
    tiles.cull_stack_(stack, keep)
    lay, pool = PackedLayout.from_dense(stack, keep, dyn)
    assert pool.shape == (lay.n_slots * 64, 4)
    back = torch.stack([lay.unpack_plane(pool, d) for d in range(D)])
    # every texel a kept quad can read comes back bit for bit; the others read as culled
    m = keep_t[:, None, :, :, None].expand_as(stack)
    assert torch.equal(back[m], stack[m])
    dead = ~keep_t[:, None].expand(D, T, Hs, Ws)
    # (a culled texel inside a stored block keeps whatever the dense stack held there: here the culled logit too)
    assert bool((back[..., 3][dead] == tiles.CULLED_ALPHA).all())
    sub = lay.unpack_frames(pool, [2, 0])
    assert torch.equal(sub[:, 0], back[:, 2]) and torch.equal(sub[:, 1], back[:, 0])


def test_packed_footprint_of_the_bench_models():
    """the 16 %-kept model of the bench (one blob of kept quads per plane, half of them dynamic): cfg3 dims and a cfg5-shaped band --
    sizes from the block table alone (nothing of that size is allocated)."""
    for (D, T, Hs, Ws, QH, QW) in ((32, 50, 396, 704, 35, 63), (32, 50, 720, 1280, 35, 63), (96, 120, 311, 3840, 35, 63)):
        qy, qx = torch.meshgrid(torch.arange(QH), torch.arange(QW), indexing="ij")
        keep = torch.zeros((D, QH, QW), dtype=torch.bool)
        for d in range(D):
            cy, cx = (7 * d + 3) % QH, (11 * d + 5) % QW
            keep[d] = ((qy - cy).abs() <= QH // 5) & ((qx - cx).abs() <= QW // 4)
        dyn = keep & ((qy + qx) % 2 == 0)[None]
        lay = PackedLayout(keep, dyn, T, Hs, Ws)
        frac = lay.pool_bytes / lay.dense_bytes
        assert 0.05 < frac < 0.30, frac
