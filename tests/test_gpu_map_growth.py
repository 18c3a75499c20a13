"""Incremental map growth (BASELINE config 4; the reference merges clouds with `*map_cloud += *cloud`,
global_map_frames_manager.cpp:131, and voxel-filters the sum, :142-146): a registered scan is
transformed into the map frame, appended to the map cloud on the device, the voxel grid is applied
again and the index rebuilt.  Bit-exact against the oracle's voxel grid of the same concatenation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_map_grows_by_registered_scans(api, ctx, orc, synth):
    rng = np.random.default_rng(11)
    world = synth.make_map(400_000)                                  # 20 m x 20 m x 10 m
    west = world[world[:, 0] < 2.0]                                  # the map knows the western part only
    map_cloud = api.Cloud(ctx, west)
    assert map_cloud.voxel_downsample(0.1, "pcl") == 0
    ds0 = map_cloud.download()
    ref0 = orc.voxel_pcl(west, 0.1)[0]
    assert np.array_equal(ds0, ref0)

    # a scan that sees x in [-2, 6]: half known, half new, taken from pose T
    T = synth.make_T((0.05, -0.03, 0.02), (0.1, -0.2, 0.5))
    seen = world[(world[:, 0] > -2.0) & (world[:, 0] < 6.0)]
    seen = seen[rng.choice(len(seen), 60_000, replace=False)]
    Ti = np.linalg.inv(T)
    scan = (seen.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)

    # register it against the old map (only the overlapping part finds neighbours)
    mp = api.Map(ctx, map_cloud, 0.25)
    icp = api.Icp(ctx, 0.5, 30, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source(scan)
    r = icp.align("o3d_p2p")
    dt, dr = synth.pose_error(r["T64"], T)
    assert 0.3 < r["fitness"] < 0.8 and dt < 0.05 and dr < 5e-3           # the unknown half pulls on the boundary points

    # grow: transform by the estimated pose, append, voxel grid, rebuild
    reg = api.Cloud(ctx, scan).transform(r["T"])
    reg_host = reg.download()
    Tf = np.asarray(r["T"], dtype=np.float32)
    x, y, z = scan[:, 0], scan[:, 1], scan[:, 2]
    assert np.array_equal(reg_host, np.stack([Tf[k, 0] * x + Tf[k, 1] * y + Tf[k, 2] * z + Tf[k, 3] for k in range(3)], 1))   # icp_point_to_point.cpp:103-105
    n_before = len(map_cloud)
    map_cloud.append(reg)
    assert len(map_cloud) == n_before + len(scan) and len(reg) == len(scan)
    assert np.array_equal(map_cloud.download(), np.concatenate([ds0, reg_host]))
    assert map_cloud.voxel_downsample(0.1, "pcl") == 0
    grown = map_cloud.download()
    assert np.array_equal(grown, orc.voxel_pcl(np.concatenate([ds0, reg_host]), 0.1)[0])
    assert grown[:, 0].max() > 5.5 and len(grown) > len(ds0)

    # a second scan, further east, now locks on the grown map with nearly all of its points
    mp2 = api.Map(ctx, map_cloud, 0.25)
    seen2 = world[(world[:, 0] > 1.0) & (world[:, 0] < 5.5)]
    seen2 = seen2[rng.choice(len(seen2), 40_000, replace=False)]
    T2 = synth.make_T((-0.04, 0.06, 0.01), (0.0, 0.1, -0.4))
    T2i = np.linalg.inv(T2)
    scan2 = (seen2.astype(np.float64) @ T2i[:3, :3].T + T2i[:3, 3]).astype(np.float32)
    icp.set_target(mp2)
    icp.set_source(scan2)
    r2 = icp.align("o3d_p2p")
    icp.set_target(mp)
    r_old = icp.align("o3d_p2p")
    assert r2["fitness"] > 0.9 and r2["fitness"] > r_old["fitness"] + 0.3
    dt, dr = synth.pose_error(r2["T64"], T2)
    assert dt < 0.05 and dr < 5e-3
    # appending an empty cloud and appending across buffers that must grow several times
    empty = api.Cloud(ctx, np.zeros((0, 3), np.float32))
    n = len(map_cloud)
    map_cloud.append(empty)
    assert len(map_cloud) == n
    acc = api.Cloud(ctx, west[:10])
    for k in range(1, 6):
        acc.append(api.Cloud(ctx, west[10 * k:10 * (k + 1) * (k + 1)]))
    assert np.array_equal(acc.download(), np.concatenate([west[:10]] + [west[10 * k:10 * (k + 1) * (k + 1)] for k in range(1, 6)]))
    with pytest.raises(api.SlamFusionError):
        acc.append(acc)


def test_voxel_merge_equals_append_plus_filter(api, ctx, orc, synth):
    """sf_cloud_voxel_merge == `*map_cloud += *cloud` + VoxelGrid (global_map_frames_manager.cpp:131,142-146), bit for bit,
    through several growth steps: pending points that hit existing voxels (centroids re-summed: old point first), that open
    new voxels inside, before and after the map's extent (the bounding box -- hence every linear index -- changes), duplicates
    and non-finite points; and the fall-back when the map is not a filtered cloud."""
    rng = np.random.default_rng(5)
    prev = api.voxel_merge_min_points(0)                             # (small maps take the full path by default: merge them here)
    base = synth.make_map(300_000)                                   # 17 m x 17 m x 10 m
    host = orc.voxel_pcl(base[base[:, 0] < 0.0], 0.1)[0]
    dev = api.Cloud(ctx, host.copy())
    assert dev.voxel_downsample(0.1, "pcl") == 0                      # (already filtered: stays itself up to merges)
    host = orc.voxel_pcl(host, 0.1)[0]
    assert np.array_equal(dev.download(), host)
    steps = [base[(base[:, 0] >= -1.0) & (base[:, 0] < 2.0)][:40_000],                      # overlaps the edge: touched and new voxels
             (base[:30_000] * [1.0, 1.0, 1.0] + [0.0, 0.0, 0.004]).astype(np.float32),      # nearly the same points again: mostly touched voxels
             (rng.uniform(-1, 1, (5_000, 3)) * [12.0, 12.0, 8.0]).astype(np.float32),      # beyond the extent on every side: min_b and the strides change
             np.repeat(base[:50], 40, axis=0)]                                              # 40 copies of 50 points
    steps[0][7] = [np.nan, 0.0, 0.0]
    steps[3][11] = [0.0, np.inf, 0.0]
    for k, add in enumerate(steps):
        st, merged = dev.voxel_merge(api.Cloud(ctx, add), 0.1)
        assert st == 0 and merged, k
        host = orc.voxel_pcl(np.concatenate([host, add]), 0.1)[0]
        got = dev.download()
        assert got.shape == host.shape and np.array_equal(got, host), k
    # a cloud that is not voxel-filtered (two points per voxel): keys not strictly ascending -> the full path, same result
    raw = api.Cloud(ctx, np.concatenate([host[:1000], host[:1000] + np.float32(0.001)]))
    st, merged = raw.voxel_merge(api.Cloud(ctx, steps[2]), 0.1)
    assert st == 0 and not merged
    assert np.array_equal(raw.download(), orc.voxel_pcl(np.concatenate([host[:1000], host[:1000] + np.float32(0.001), steps[2]]), 0.1)[0])
    # empty pending: the full path (== re-filtering the map)
    st, merged = dev.voxel_merge(api.Cloud(ctx, np.zeros((0, 3), np.float32)), 0.1)
    assert not merged and np.array_equal(dev.download(), orc.voxel_pcl(host, 0.1)[0])
    api.voxel_merge_min_points(prev)
    st, merged = dev.voxel_merge(api.Cloud(ctx, steps[2]), 0.1)       # below the default threshold: the full path, same semantics
    assert st == 0 and not merged


def _same_index(a, b):
    for k in ("pts4", "cell_start", "org"):
        if a[k].shape != b[k].shape or not np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)):
            return k
    if a["inv_h"] != b["inv_h"] or a["gap_eps"] != b["gap_eps"]:
        return "scalars"
    return ""


def test_index_patch_equals_rebuild(api, ctx, synth):
    """sf_map_patch == sf_map_build of the merged cloud with the same cell (points in cell order with their ids, cell table,
    geometry: bit for bit) over growth steps that touch voxels, open new ones inside and beyond the map's
    upper faces (the grid gets more cells per row: every cell id changes, the order does not), that replace a point which
    held one of the map's bounds, that reach below its smallest coordinates (with an origin lattice the origin stays), and the
    cases that must take the build: the origin moves, the cloud was changed between merge and patch, the merge took its full
    path, the index is of another cloud.  setTargetPointCloud after `*map_cloud += *cloud` + VoxelGrid:
    icp_point_to_point.cpp:49-55, global_map_frames_manager.cpp:131,142-146."""
    rng = np.random.default_rng(9)
    prev = api.voxel_merge_min_points(0)
    try:
        base = synth.make_map(400_000)                                # 20 m x 20 m x 10 m
        inner = base[(np.abs(base[:, 0]) < 6.0) & (np.abs(base[:, 1]) < 6.0)]
        dev = api.Cloud(ctx, inner)
        dev.voxel_downsample(0.1, "pcl")
        lo, hi = dev.download().min(0), dev.download().max(0)
        for cell, lattice in ((0.25, 0), (0.0, 0), (0.25, 64)):     # an explicit cell, the one the build chooses itself, an origin lattice (sf_map_set_origin_lattice)
            dev = api.Cloud(ctx, inner)
            dev.voxel_downsample(0.1, "pcl")
            new_map = lambda cloud, c: api.Map(ctx).set_origin_lattice(lattice).build(cloud, c)
            mp = new_map(dev, cell)
            h, dims0 = mp.cell_size()
            core = inner[(np.abs(inner[:, 0]) < 5.5) & (np.abs(inner[:, 1]) < 5.5) & (np.abs(inner[:, 2]) < 4.5)]    # (away from the points that hold the map's extremes)
            near = lambda n: (core[rng.choice(len(core), n, replace=False)] + rng.normal(0, 0.004, (n, 3))).astype(np.float32)

            def top():                                                # the point that holds the largest y right now
                pts = dev.download()
                return pts[np.argmax(pts[:, 1])]

            def bottom():                                             # ... the smallest x
                pts = dev.download()
                return pts[np.argmin(pts[:, 0])]
            steps = [("touch + fill", near(30_000), True),
                     ("beyond +x / +y / +z", np.concatenate([near(5_000), (rng.uniform(0, 1, (20_000, 3)) * (hi - lo + [3.0, 2.0, 1.0]) + lo + 0.01).astype(np.float32)]), True),
                     ("duplicates", np.repeat(near(200), 30, axis=0), True),
                     ("the point that holds the largest y is replaced", lambda: np.concatenate([near(1_000), (top() - np.float32(0.001))[None]]), True),   # (one reduction for the new bounds, then patched)
                     ("the point that holds the smallest x is replaced", lambda: np.concatenate([near(1_000), (bottom() + np.float32(0.001))[None]]), lattice > 0),  # (the smallest x moves with its centroid: with it the origin, unless that is a lattice point)
                     ("below the smallest coordinates", np.concatenate([near(1_000), (lo - [0.5, 0.3, 0.2]).astype(np.float32)[None]]), lattice > 0),
                     ("far below the origin", np.concatenate([near(1_000), (lo - [40.0, 0.0, 0.0]).astype(np.float32)[None]]), False)]
            for name, add, expect in steps:
                add = add() if callable(add) else add
                st, merged = dev.voxel_merge(api.Cloud(ctx, add), 0.1)
                assert st == 0 and merged, name
                patched = mp.patch(dev)
                assert patched == expect, (name, cell, lattice, mp.last_patch)
                ref = new_map(dev, h)
                assert _same_index(mp.index(), ref.index()) == "", (name, cell, _same_index(mp.index(), ref.index()))
                assert mp.cell_size() == ref.cell_size() and len(mp) == len(ref) == len(dev)
                pts = dev.download()                                  # (the bounds the merge carries over are the bounds of the points)
                org = pts.min(0) if lattice == 0 else (np.floor(pts.min(0).astype(np.float64) / (lattice * float(h))) * (lattice * float(h))).astype(np.float32)
                assert np.array_equal(ref.index()["org"], org)
                assert ref.cell_size()[1] == tuple(int(np.floor((float(pts[:, d].max()) - float(org[d])) / float(h))) + 1 for d in range(3))
                if name.startswith("beyond"):
                    assert mp.cell_size()[1] != dims0                 # more cells per row: the patch renumbered them
            # the patched index answers like the rebuilt one
            q = (dev.download()[rng.choice(len(dev), 20_000)] + rng.normal(0, 0.05, (20_000, 3))).astype(np.float32)
            ia, da = mp.nn(q, 0.25)
            ib, db = ref.nn(q, 0.25)
            assert np.array_equal(ia, ib) and np.array_equal(da, db)
            # the cloud changes between merge and patch: nothing of the merge may be trusted
            st, merged = dev.voxel_merge(api.Cloud(ctx, near(5_000)), 0.1)
            assert merged
            dev.transform(np.eye(4, dtype=np.float32))
            assert not mp.patch(dev)
            assert _same_index(mp.index(), new_map(dev, h).index()) == ""
            # a merge that took the full path (pending empty), then a good one again
            dev.voxel_merge(api.Cloud(ctx, np.zeros((0, 3), np.float32)), 0.1)
            assert not mp.patch(dev)
            st, merged = dev.voxel_merge(api.Cloud(ctx, near(5_000)), 0.1)
            assert merged and mp.patch(dev)
            assert _same_index(mp.index(), new_map(dev, h).index()) == ""
            # an index of another cloud, and a merge of another cloud in between (the merge tables are the context's)
            other = api.Cloud(ctx, dev.download()[::2].copy())
            other.voxel_downsample(0.1, "pcl")
            mo = new_map(other, h)
            st, merged = dev.voxel_merge(api.Cloud(ctx, near(3_000)), 0.1)
            assert merged and not mo.patch(dev)
            assert _same_index(mo.index(), new_map(dev, h).index()) == ""
            mp.build(dev, h)
            st, merged = dev.voxel_merge(api.Cloud(ctx, near(3_000)), 0.1)
            st2, merged2 = other.voxel_merge(api.Cloud(ctx, near(2_000)), 0.1)
            assert merged and merged2 and not mp.patch(dev)
            assert _same_index(mp.index(), new_map(dev, h).index()) == ""
    finally:
        api.voxel_merge_min_points(prev)


def test_index_patch_registration_identical(api, ctx, synth):
    """A registration against the patched index equals one against the rebuilt index, bit for bit, normals re-estimated on both."""
    rng = np.random.default_rng(10)
    prev = api.voxel_merge_min_points(0)
    try:
        base = synth.make_map(300_000)
        dev = api.Cloud(ctx, base[base[:, 0] < 2.0])
        dev.voxel_downsample(0.1, "pcl")
        mp = api.Map(ctx, dev, 0.25)
        add = base[(base[:, 0] >= 1.0) & ((base[:, 0] < 1.8) | (base[:, 0] >= 2.0)) & (base[:, 0] < 6.0) & (np.abs(base[:, 1]) < 8.0) & (np.abs(base[:, 2]) < 4.5)]
        add = (add + np.float32(0.003)).astype(np.float32)               # (towards +x only, the points that hold the map's extremes left alone: the origin stays)
        st, merged = dev.voxel_merge(api.Cloud(ctx, add), 0.1)
        assert merged and mp.patch(dev)
        ref = api.Map(ctx, dev, 0.25)
        ds = dev.download()
        scan, _ = synth.make_scan(ds[(ds[:, 0] > 0.0) & (ds[:, 0] < 5.0)], 20_000)
        out = []
        for m in (mp, ref):
            m.estimate_normals(0.25)
            icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
            icp.set_target(m)
            icp.set_source(scan)
            out.append([icp.align(mode) for mode in ("ref_cpp", "o3d_p2p", "p2plane")])
            icp.close()
        for a, b in zip(*out):
            assert np.array_equal(a["T64"], b["T64"]) and a["iterations"] == b["iterations"] and a["fitness"] == b["fitness"]
        na, nb = mp.download_normals(), ref.download_normals()
        assert np.array_equal(na[0], nb[0]) and np.array_equal(na[1], nb[1])
    finally:
        api.voxel_merge_min_points(prev)


def test_index_patch_edge_cases(api, ctx, synth):
    """sf_map_patch where one side of the merge is empty or odd: every pending point re-observes an existing voxel (no new
    voxel: ids do not shift), none does (only new voxels), non-finite pending points, one pending point; and a sparse map whose
    cell table is built by scan (more than 2^28 cells).  Always bit-identical to sf_map_build of the merged cloud."""
    rng = np.random.default_rng(21)
    prev = api.voxel_merge_min_points(0)
    try:
        base = synth.make_map(200_000)
        inner = base[(np.abs(base[:, 0]) < 5.0) & (np.abs(base[:, 1]) < 5.0) & (np.abs(base[:, 2]) < 4.0)]
        dev = api.Cloud(ctx, inner)
        dev.voxel_downsample(0.1, "pcl")
        mp = api.Map(ctx, dev, 0.25)
        pts = dev.download()
        core = pts[(np.abs(pts[:, 0]) < 4.5) & (np.abs(pts[:, 1]) < 4.5) & (np.abs(pts[:, 2]) < 3.5)]
        free = (np.floor(rng.uniform(-4, 4, (3000, 3)) * 10) / 10 + 0.05).astype(np.float32)          # voxel centres ...
        occupied = {tuple(v) for v in np.floor(pts / 0.1).astype(np.int64)}
        free = np.array([p for p in free if tuple(np.floor(p / 0.1).astype(np.int64)) not in occupied], np.float32)   # ... of voxels the map does not have
        nan_mix = np.concatenate([core[:500] + np.float32(0.002), np.full((3, 3), np.nan, np.float32), free[:50] + np.float32(0.01)])
        steps = [("only re-observed voxels", core[rng.choice(len(core), 5_000, replace=False)].copy(), 0),
                 ("only new voxels", free[: len(free) // 2], None),
                 ("non-finite pending points", nan_mix, None),
                 ("one pending point", core[7:8] + np.float32(0.001), 0)]
        for name, add, grew in steps:
            n0 = len(dev)
            st, merged = dev.voxel_merge(api.Cloud(ctx, add), 0.1)
            assert st == 0 and merged, name
            if grew is not None:
                assert len(dev) - n0 == grew, name
            assert mp.patch(dev), (name, mp.last_patch)
            assert _same_index(mp.index(), api.Map(ctx, dev, 0.25).index()) == "", name
        with pytest.raises(api.SlamFusionError):                      # a map without an index has nothing to carry over
            api.Map(ctx).patch(dev)
        # sparse map, 300 m x 300 m x 20 m (PCL's int32 voxel index still holds) at a 0.15 m cell: 5.4e8 cells, the table built by scan
        corners = np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5]]) * np.array([285.0, 285.0, 8.0])
        sparse = np.concatenate([c + synth.make_map(40_000, seed=60 + k)[:20_000] % 10.0 for k, c in enumerate(corners)]).astype(np.float32)
        sparse = np.concatenate([sparse, np.array([[-1, -1, -1], [299, 299, 19]], np.float32)])   # pin the bounds away from the clusters
        big = api.Cloud(ctx, sparse)
        assert big.voxel_downsample(0.1, "pcl") == 0
        mb = api.Map(ctx, big, 0.15)
        assert np.prod(np.array(mb.cell_size()[1], dtype=np.float64)) > 2.0 ** 28
        add = (corners[2] + rng.uniform(0, 12, (8_000, 3))).astype(np.float32)
        st, merged = big.voxel_merge(api.Cloud(ctx, add), 0.1)
        assert st == 0 and merged
        assert mb.patch(big), mb.last_patch
        ref = api.Map(ctx, big, 0.15)
        a, b = mb.index(), ref.index()
        assert _same_index(a, b) == ""
    finally:
        api.voxel_merge_min_points(prev)


@pytest.mark.parametrize("kind", ["dense", "sparse", "lattice"])
def test_cell_table_is_the_lower_bound_of_the_sorted_keys(api, ctx, synth, kind):
    """cell_start[c] = first sorted position whose cell id is >= c, for every c in [0, cells] -- checked against numpy on the index as
    it lies in HBM, for the table built point by point (dense maps), by tails + max-scan (fewer than one point per 16 cells) and
    behind a snapped origin (sf_map_set_origin_lattice: whole empty layers in front of the data); the points are in (cell, id)
    order and every point is there once."""
    rng = np.random.default_rng(31)
    if kind == "sparse":
        pts = np.concatenate([c + rng.uniform(0, 3, (5_000, 3)) for c in ([0, 0, 0], [60, 40, 5], [20, 70, 9])]).astype(np.float32)
    else:
        pts = synth.make_map(120_000)[:100_000]
    mp = api.Map(ctx).set_origin_lattice(64 if kind == "lattice" else 0).build(api.Cloud(ctx, pts), 0.25)
    ix, (h, dims) = mp.index(), mp.cell_size()
    cells = int(dims[0]) * int(dims[1]) * int(dims[2])
    if kind != "lattice":
        assert (cells > 16 * len(pts)) == (kind == "sparse")     # (which builder ran; a lattice always takes the scan)
    p4 = ix["pts4"]
    ids = p4[:, 3].copy().view(np.uint32)
    assert np.array_equal(np.sort(ids), np.arange(len(pts), dtype=np.uint32)) and np.array_equal(p4[:, :3], pts[ids])
    inv_h = np.float32(ix["inv_h"])
    c = [np.clip(np.floor((p4[:, d] - ix["org"][d]) * inv_h), 0, dims[d] - 1).astype(np.int64) for d in range(3)]   # float32 arithmetic, as k_cell_keys
    keys = (c[2] * dims[1] + c[1]) * dims[0] + c[0]
    assert np.all(np.diff(keys) >= 0) and np.all((np.diff(keys) > 0) | (np.diff(ids.astype(np.int64)) > 0))              # (cell, id) order
    assert np.array_equal(ix["cell_start"].astype(np.int64), np.searchsorted(keys, np.arange(cells + 1), side="left"))
    if kind == "lattice":
        step = 64 * float(h)
        assert np.array_equal(ix["org"], (np.floor(pts.min(0).astype(np.float64) / step) * step).astype(np.float32))
    with pytest.raises(api.SlamFusionError):
        mp.set_origin_lattice(-1)
