"""Camera side and selection logic of the offline renderer (videoloop3d_amd/render_video.py; scripts/script_render_video.py:33-87,
dataloader.py:60-134, 205-260) against golden G18 = the reference's own dataloader.load_llff_data on synthetic poses_bounds.npy
(tests/golden/make_golden_r05.py).  CPU only."""
import numpy as np
import pytest

from videoloop3d_amd import render_video as RV


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("tag", ["s0", "s1"])
def test_llff_poses_and_spiral_match_the_reference(golden, name, tag):
    g = golden("g18_render_poses.npz")
    factor, b0, b1, frm, scal = g[f"{name}_{tag}_args"]
    poses, intrins, bds, rposes, rintr = RV.load_llff_poses(g[f"{name}_poses_bounds"], factor=int(factor), recenter=True, bd_factor=(b0, b1),
                                                            render_frm=int(frm), render_scaling=float(scal))
    k = f"{name}_{tag}_"
    for got, key in ((poses, "poses"), (intrins, "intrins"), (bds, "bds"), (rposes, "render_poses"), (rintr, "render_intrins")):
        want = g[k + key]
        assert got.shape == want.shape and got.dtype == want.dtype == np.float32, key
        assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), (key, np.abs(got - want).max())
    if tag == "s0":
        avg = RV.poses_avg(g[k + "poses"])
        assert np.abs(avg - g[f"{name}_poses_avg"]).max() <= 1e-6
        ext, K, near, far = RV.reference_camera(g[k + "poses"], g[k + "intrins"], g[k + "bds"])
        assert ext.shape == (4, 4) and np.abs(ext @ np.concatenate([avg[:, :4], [[0, 0, 0, 1]]]) - np.eye(4)).max() <= 1e-5
        assert near == float(g[k + "bds"].min()) and far == float(g[k + "bds"].max()) and np.array_equal(K, g[k + "intrins"][0])


def test_view_and_time_selection():
    # script_render_video.py:47-85 on 12 spiral poses, 4 training views, a 5-frame loop
    rp = np.arange(12 * 12, dtype=np.float32).reshape(12, 3, 4)
    ri = np.tile(np.eye(3, dtype=np.float32), (12, 1, 1))
    tp = -np.arange(4 * 12, dtype=np.float32).reshape(4, 3, 4)
    ti = 2 * np.tile(np.eye(3, dtype=np.float32), (4, 1, 1))
    vp, vi, rt = RV.select_views_times(rp, ri, tp, ti, 5)
    assert np.array_equal(vp, rp) and rt.tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1]                  # the spiral, frame i % T
    vp, vi, rt = RV.select_views_times(rp, ri, tp, ti, 5, v="r3")
    assert len(vp) == 5 and all(np.array_equal(p, rp[3]) for p in vp) and rt.tolist() == [0, 1, 2, 3, 4]  # one spiral pose, one loop
    vp, vi, rt = RV.select_views_times(rp, ri, tp, ti, 5, v="2")
    assert all(np.array_equal(p, tp[2]) for p in vp) and all(np.array_equal(k, ti[2]) for k in vi)
    vp, vi, rt = RV.select_views_times(rp, ri, tp, ti, 5, v="test", test_view_idx="1,3")
    assert all(np.array_equal(p, tp[1]) for p in vp)
    assert RV.select_views_times(rp, ri, tp, ti, 5, t="0,2,7")[2].tolist() == [0, 2, 2] and len(RV.select_views_times(rp, ri, tp, ti, 5, t="0,2,7")[0]) == 3
    assert RV.select_views_times(rp, ri, tp, ti, 5, t="1:4,4:1")[2].tolist() == [1, 2, 3, 4, 3, 2]       # ranges, end excluded, descending allowed
    assert RV.select_views_times(rp, ri, tp, ti, 5, t="6")[2].tolist() == [1]
    assert RV.default_render_frames(50) == 150 and RV.default_render_frames(50, 80) == 80 and RV.default_render_frames(7) == 126
