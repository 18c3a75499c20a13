"""Pins the oracle's search / crop / voxel restatements with independent tools (the reference
holds no fixtures for them: SURVEY.md §4, §8c): scipy.spatial.cKDTree, brute force and
closed-form lattice answers."""
import numpy as np
import pytest
from scipy.spatial import cKDTree


def test_kdtree_matches_ckdtree_and_bruteforce(orc, small_world):
    m, scan = small_world["map"], small_world["scan"]
    tree = orc.KdTreeF(m)
    q = np.concatenate([scan, scan + np.float32(0.9)])
    idx, d2 = tree.nn(q)
    dd, ii = cKDTree(m.astype(np.float64)).query(q.astype(np.float64))
    same = idx == ii
    # exact search: identical neighbours except exact distance ties
    assert same.mean() > 0.9995
    assert np.allclose(d2[~same].astype(np.float64), dd[~same] ** 2, rtol=1e-5)
    assert np.allclose(d2.astype(np.float64), dd ** 2, rtol=2e-6, atol=1e-12)
    bi, bd = orc.bruteforce_nn(m, q[:400])
    assert np.array_equal(bd, d2[:400])            # same float32 expression -> bit-equal
    assert (bi == idx[:400]).mean() > 0.99


def test_kdtree_f64_and_edge_cases(orc):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-5, 5, (5000, 3))
    q = rng.uniform(-6, 6, (700, 3))
    idx, d2 = orc.KdTreeD(pts).nn(q)
    dd, ii = cKDTree(pts).query(q)
    assert np.array_equal(idx, ii)
    assert np.allclose(d2, dd ** 2, rtol=1e-12)
    # empty tree, non-finite query, non-finite map points are not indexed (PCL behaviour)
    e_idx, e_d2 = orc.KdTreeF(np.zeros((0, 3), np.float32)).nn(np.zeros((2, 3), np.float32))
    assert (e_idx == -1).all() and np.isinf(e_d2).all()
    m = np.array([[0, 0, 0], [np.nan, 0, 0], [1, 0, 0], [np.inf, 1, 1]], np.float32)
    t = orc.KdTreeF(m)
    i2, _ = t.nn(np.array([[0.9, 0, 0], [np.nan, 0, 0], [100, 1, 1]], np.float32))
    assert list(i2) == [2, -1, 2]
    # duplicates: distance zero, any of the duplicates
    dup = np.array([[1, 2, 3]] * 40 + [[4, 5, 6]], np.float32)
    i3, d3 = orc.KdTreeF(dup).nn(np.array([[1, 2, 3]], np.float32))
    assert d3[0] == 0 and 0 <= i3[0] < 40


def test_subsample_and_floor(orc):
    pts = np.arange(30, dtype=np.float32).reshape(10, 3)
    assert np.array_equal(orc.uniform_subsample(pts, 3), pts[::3])
    assert np.array_equal(orc.uniform_subsample(pts, 2), pts[::2])
    assert np.array_equal(orc.uniform_subsample(pts[:2], 3), pts[:2])      # size < step: untouched (hpp:58-61)
    assert len(orc.uniform_subsample(pts[:3], 3)) == 1
    z = np.array([[0, 0, -1], [0, 0, 0], [0, 0, 1e-6], [1, 1, 2]], np.float32)
    assert np.array_equal(orc.remove_floor(z), z[2:])                       # strictly z > 0


def test_crop_radius_order_and_threshold(orc, small_world):
    m = small_world["map"]
    c = np.array([1.0, -0.5, 0.25], np.float32)
    out, idx = orc.crop_radius(m, c, 2.0)
    d2 = ((m.astype(np.float64) - c) ** 2).sum(1)
    expect = np.nonzero(d2 < 4.0)[0]
    assert abs(len(idx) - len(expect)) <= 2                                 # float32 vs float64 at the rim
    assert set(idx) - set(np.nonzero(d2 < 4.0 + 1e-4)[0]) == set()
    dsel = d2[idx]
    assert (np.diff(dsel) >= -1e-5).all()                                   # ascending distance (sorted FLANN radius search)
    assert np.array_equal(out, m[idx])
    out0, idx0 = orc.crop_radius(m, c, 0.0)
    assert len(idx0) == 0


def test_crop_boxes(orc):
    rng = np.random.default_rng(5)
    p = rng.uniform(-10, 20, (20000, 3)).astype(np.float32)
    p[5] = [np.nan, 1, 1]
    lo, hi = [0, -7.5, 0], [15, 7.5, 7.5]                                   # localization_node.py:53-56
    p[6] = [15.0, 7.5, 7.5]                                                 # inclusive upper corner
    p[7] = [0.0, -7.5, 0.0]                                                 # inclusive lower corner
    out, idx = orc.crop_aabb(p, lo, hi)
    pd = p.astype(np.float64)
    ok = (pd[:, 0] >= 0) & (pd[:, 0] <= 15) & (pd[:, 1] >= -7.5) & (pd[:, 1] <= 7.5) & (pd[:, 2] >= 0) & (pd[:, 2] <= 7.5)
    assert np.array_equal(idx, np.nonzero(ok)[0])
    assert 6 in idx and 7 in idx and 5 not in idx
    th = 0.4
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    center, ext = np.array([3.0, 1.0, 2.0]), np.array([30.0, 15.0, 15.0])
    out2, idx2 = orc.crop_obb(p, center, R, ext)
    loc = (pd - center) @ R                                                 # d . R[:, k]
    ok2 = (np.abs(loc) <= ext / 2).all(1)
    assert np.array_equal(idx2, np.nonzero(ok2)[0])


def test_voxel_pcl_lattice_closed_form(orc):
    # points at voxel centres of a 7 x 5 x 3 lattice (leaf 0.1), two points per voxel
    i, j, k = np.meshgrid(np.arange(7), np.arange(5), np.arange(3), indexing="ij")
    centres = np.stack([i, j, k], -1).reshape(-1, 3).astype(np.float64) * 0.1 + 0.05 + np.array([-0.3, 0.2, 1.0])
    pts = np.concatenate([centres - 0.01, centres + 0.01]).astype(np.float32)
    out, vidx, ovox, st = orc.voxel_pcl(pts, 0.1)
    assert st == 0 and len(out) == 105
    inv = np.float32(1.0) / np.float32(0.1)
    ijk = np.floor(pts * inv).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    expect = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    assert np.array_equal(vidx, expect)                                     # i + j*dx + k*dx*dy, bit-exact
    assert np.array_equal(ovox, np.unique(expect))                          # ascending index output
    assert np.allclose(out[np.searchsorted(ovox, expect[:105])], centres, atol=2e-6)


def test_voxel_pcl_edge_cases(orc):
    assert len(orc.voxel_pcl(np.zeros((0, 3), np.float32), 0.1)[0]) == 0
    p = np.array([[0, 0, 0], [np.nan, 1, 1], [0.05, 0.05, 0.05], [3, 3, 3]], np.float32)
    out, vidx, ovox, st = orc.voxel_pcl(p, 0.1)
    assert vidx[1] == -1 and vidx[0] == vidx[2] and len(out) == 2
    assert np.array_equal(out[0], (p[0] + p[2]) / np.float32(2))
    # int32 overflow: PCL warns and returns the input unchanged
    big = np.array([[0, 0, 0], [2000, 2000, 2000]], np.float32)
    out, vidx, ovox, st = orc.voxel_pcl(big, 0.1)
    assert st == -1 and np.array_equal(out, big)


def test_voxel_o3d_origin_and_means(orc):
    rng = np.random.default_rng(9)
    p = rng.uniform(-3, 3, (5000, 3)).astype(np.float32).astype(np.float64)
    out, ijk, oijk, st = orc.voxel_o3d(p, 0.1)
    vmin = p.min(0) - 0.05                                                  # min_bound - voxel/2
    expect = np.floor((p - vmin) / 0.1).astype(np.int32)
    assert np.array_equal(ijk, expect)
    keys, inv = np.unique(expect, axis=0, return_inverse=True)
    assert np.array_equal(oijk, keys)                                       # lexicographic (i, j, k)
    means = np.zeros_like(out)
    np.add.at(means, inv.ravel(), p)
    means /= np.bincount(inv.ravel())[:, None]
    assert np.allclose(out, means, rtol=0, atol=1e-12)


def test_normals_against_numpy_eigh(orc):
    rng = np.random.default_rng(2)
    xy = rng.uniform(-1, 1, (3000, 2))
    plane = np.c_[xy, 0.3 * xy[:, 0] - 0.2 * xy[:, 1] + rng.normal(0, 1e-3, 3000)].astype(np.float32)
    nrm, cnt = orc.normals_radius(plane, 0.2)
    true_n = np.array([-0.3, 0.2, 1.0])
    true_n /= np.linalg.norm(true_n)
    inner = cnt >= 8
    assert (np.abs(nrm[inner] @ true_n) > 0.999).all()
    tree = cKDTree(plane.astype(np.float64))
    for i in rng.choice(3000, 40, replace=False):
        nb = tree.query_ball_point(plane[i].astype(np.float64), 0.2)
        assert len(nb) == cnt[i]
        if len(nb) >= 3:
            P = plane[nb].astype(np.float64)
            w, v = np.linalg.eigh(np.cov(P.T, bias=True))
            assert abs(abs(v[:, 0] @ nrm[i]) - 1) < 1e-5


def test_voxel_pcl64_equals_pcl_and_survives_int32_overflow(orc, synth):
    """The int64-index voxel grid (extension for maps past 2^31 voxels) is the SAME arithmetic:
    identical ids / order / centroids wherever pcl::VoxelGrid does not overflow; where PCL gives
    up ("Leaf size is too small") it still filters, and every centroid is the float32 mean of the
    points that share its cell."""
    raw = synth.make_map(60_000)
    raw[5] = [np.nan, 0, 0]
    a, vidx, ovox, st = orc.voxel_pcl(raw, 0.1)
    b, vidx64, ovox64 = orc.voxel_pcl64(raw, 0.1)
    assert st == 0 and np.array_equal(a, b) and np.array_equal(vidx.astype(np.int64), vidx64) and np.array_equal(ovox.astype(np.int64), ovox64)
    rng = np.random.default_rng(0)
    big = np.concatenate([rng.uniform(0, 1, (2000, 3)), rng.uniform(0, 1, (2000, 3)) + [2500.0, 1800.0, 900.0]]).astype(np.float32)
    _, _, _, st = orc.voxel_pcl(big, 0.1)
    assert st == -1                                                        # PCL: output = input
    c, pid, oid = orc.voxel_pcl64(big, 0.1)
    assert oid.max() > 2**31 and np.all(np.diff(oid) > 0) and len(c) < len(big)
    inv = np.float32(1.0) / np.float32(0.1)
    ijk = np.floor(big * inv).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    lin = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    assert np.array_equal(lin, pid)
    for k in (0, len(oid) // 2, len(oid) - 1):
        members = big[pid == oid[k]]
        s = np.zeros(3, np.float32)
        for p in members:
            s = s + p
        assert np.array_equal(c[k], s / np.float32(len(members)))


def test_normals_covariance_against_numpy(orc, synth):
    raw = synth.make_map(20_000)
    m = orc.voxel_pcl(raw, 0.1)[0]
    nrm, cnt, cov = orc.normals_radius_cov(m, 0.3)
    nrm2, cnt2 = orc.normals_radius(m, 0.3)
    assert np.array_equal(nrm, nrm2) and np.array_equal(cnt, cnt2)
    from scipy.spatial import cKDTree
    tree = cKDTree(m.astype(np.float64))
    for i in (0, 17, 500, len(m) - 1):
        nb = m[tree.query_ball_point(m[i].astype(np.float64), 0.3 + 1e-12)].astype(np.float64)
        d2 = ((nb - m[i].astype(np.float64)) ** 2).sum(1)
        nb = nb[d2 <= 0.3 * 0.3]
        assert len(nb) == cnt[i]
        if len(nb) >= 3:
            C = np.cov(nb.T, bias=True)
            assert np.allclose(cov[i], [C[0, 0], C[0, 1], C[0, 2], C[1, 1], C[1, 2], C[2, 2]], rtol=1e-9, atol=1e-15)
            w, v = np.linalg.eigh(C)
            assert abs(abs(v[:, 0] @ nrm[i]) - 1) < 1e-5
