"""The hand-written stable radix sort and scans (csrc/sf_sort.hpp) through the entry points that use them: the voxel grids
(order and centroids bit-exact against the oracle at sizes that need several tiles and every pass), the grid index
(idempotent NN on every map point), and the crop's (d^2, index) order.  No reference counterpart of its own: the
reference sorts inside pcl::VoxelGrid (global_map_frames_manager.cpp:142-146)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,extent", [(1, 1.0), (63, 0.5), (4097, 3.0), (300_000, 30.0), (1_000_003, 8.0)])
def test_voxel_grid_matches_oracle_across_sizes(api, ctx, orc, n, extent):
    rng = np.random.default_rng(n)
    pts = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    if n > 100:
        pts[rng.choice(n, 7, replace=False)] = np.nan                     # non-finite points sort last and are dropped
        pts[5] = pts[6]                                                   # duplicates: stable order decides the float32 sum
    for flavour, fn in (("pcl", orc.voxel_pcl), ("o3d", orc.voxel_o3d)):
        if flavour == "o3d" and n > 100:
            q = pts[np.isfinite(pts).all(1)]
        else:
            q = pts
        c = api.Cloud(ctx, q)
        assert c.voxel_downsample(0.1, flavour) == 0
        want = fn(q, 0.1)
        got = c.download()
        if flavour == "pcl":
            assert np.array_equal(got, want[0]) and np.array_equal(c.voxel_out_ids(), want[2])
        else:
            o = np.lexsort((want[2][:, 2], want[2][:, 1], want[2][:, 0]))   # the oracle's voxel order is a hash map's: compare as sets keyed by ijk
            g_ijk = c.voxel_out_ids().reshape(-1, 3)
            go = np.lexsort((g_ijk[:, 2], g_ijk[:, 1], g_ijk[:, 0]))
            assert np.array_equal(g_ijk[go], want[2][o]) and np.array_equal(c.voxel_out_means_f64()[go], want[0][o])


def test_index_and_crop_order(api, ctx, orc):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-20, 20, (700_001, 3)).astype(np.float32)
    mp = api.Map(ctx, api.Cloud(ctx, pts), 0.3)
    sel = rng.choice(len(pts), 50_000, replace=False)
    idx, d2 = mp.nn(pts[sel])
    assert (d2 == 0).all() and (idx == sel).mean() > 0.999                # (exact duplicates may answer with their twin)
    c = api.Cloud(ctx, pts)
    c.crop_radius([1.0, -2.0, 0.5], 9.0, sorted=True)
    want_pts, want_idx = orc.crop_radius(pts, [1.0, -2.0, 0.5], 9.0)
    assert np.array_equal(c.last_indices(), want_idx) and np.array_equal(c.download(), want_pts)
