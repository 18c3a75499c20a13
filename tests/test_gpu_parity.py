"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the
oracle on the same seeded inputs and against the committed golden vectors.
Bars (BASELINE.json north_star): integer voxel indices / NN indices / crop index sets
bit-exact; poses within 1e-4 m and 1e-5 rad of the float64 oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-5


def assert_pose_close(synth, T, Tref, tm=POSE_TOL_M, tr=POSE_TOL_RAD):
    dt, dr = synth.pose_error(T, Tref)
    assert dt < tm and dr < tr, (dt, dr)


# ------------------------------------------------------------------ voxel grids (a4, a5)
def test_voxel_pcl_bit_exact(api, ctx, orc, small_world):
    raw = small_world["raw"].copy()
    raw[11] = [np.nan, 0, 0]
    raw[12] = [0, np.inf, 0]
    c = api.Cloud(ctx, raw)
    flags = c.voxel_downsample(0.1, "pcl")
    ds, vidx, ovox, st = orc.voxel_pcl(raw, 0.1)
    assert flags == 0 and st == 0
    assert np.array_equal(c.voxel_point_ids(), vidx)          # int32 linear indices, bit-exact
    assert np.array_equal(c.voxel_out_ids(), ovox)            # ascending index output order
    assert np.array_equal(c.download(), ds)                   # float32 centroids, same summation order


def test_voxel_o3d_bit_exact(api, ctx, orc, small_world):
    raw = small_world["raw"]
    c = api.Cloud(ctx, raw)
    c.voxel_downsample(0.1, "o3d")
    means, ijk, oijk, st = orc.voxel_o3d(raw.astype(np.float64), 0.1)
    assert np.array_equal(c.voxel_point_ids().reshape(-1, 3), ijk)
    assert np.array_equal(c.voxel_out_ids().reshape(-1, 3), oijk)
    assert np.array_equal(c.voxel_out_means_f64(), means)     # float64 means, bit-exact
    assert np.array_equal(c.download(), means.astype(np.float32))


def test_voxel_edge_cases(api, ctx, orc):
    c = api.Cloud(ctx, np.zeros((0, 3), np.float32))
    assert c.voxel_downsample(0.1, "pcl") == 0 and len(c) == 0
    big = np.array([[0, 0, 0], [2000, 2000, 2000], [1, 1, 1]], np.float32)
    c = api.Cloud(ctx, big)
    assert c.voxel_downsample(0.1, "pcl") == api.SF_FLAG_VOXEL_OVERFLOW      # PCL: warn + output = input
    assert np.array_equal(c.download(), big)
    one = api.Cloud(ctx, np.array([[1.5, 2.5, 3.5]], np.float32))
    one.voxel_downsample(0.1, "pcl")
    assert np.array_equal(one.download(), [[1.5, 2.5, 3.5]])
    nan = api.Cloud(ctx, np.array([[np.nan, 0, 0]], np.float32))
    nan.voxel_downsample(0.1, "pcl")
    assert len(nan) == 0
    with pytest.raises(api.SlamFusionError):
        api.Cloud(ctx, np.array([[np.nan, 0, 0], [1, 1, 1]], np.float32)).voxel_downsample(0.1, "o3d")
    # many points in one voxel: sequential float32 sum order must match the oracle
    rng = np.random.default_rng(0)
    dense = (rng.uniform(0, 0.1, (5000, 3)) + [3.0, 4.0, 5.0]).astype(np.float32)
    c = api.Cloud(ctx, dense)
    c.voxel_downsample(0.1, "pcl")
    assert np.array_equal(c.download(), orc.voxel_pcl(dense, 0.1)[0])


# ------------------------------------------------------------------ crops / subsample / transform (a1-a3, a8, a19, a20)
def test_crops_match_oracle(api, ctx, orc, small_world):
    m = small_world["map"].copy()
    m[3] = [np.nan, 0, 0]
    center = np.array([1.0, -0.5, 0.25], np.float32)
    o_pts, o_idx = orc.crop_radius(m, center, 2.0)
    c = api.Cloud(ctx, m).crop_radius(center, 2.0, sorted=True)
    assert np.array_equal(c.last_indices(), o_idx)            # PCL order: ascending (distance, index)
    assert np.array_equal(c.download(), o_pts)
    c = api.Cloud(ctx, m).crop_radius(center, 2.0, sorted=False)
    assert np.array_equal(c.last_indices(), np.sort(o_idx))   # same set, index order
    o_pts, o_idx = orc.crop_aabb(m, [0, -7.5, 0], [15, 7.5, 7.5])
    c = api.Cloud(ctx, m).crop_aabb([0, -7.5, 0], [15, 7.5, 7.5])
    assert np.array_equal(c.last_indices(), o_idx) and np.array_equal(c.download(), o_pts)
    th = 0.4
    R = 0.8 * np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]) + 0.2 * np.eye(3)
    o_pts, o_idx = orc.crop_obb(m, [0.5, 0.2, 0.0], R, [3.0, 1.5, 1.5])
    c = api.Cloud(ctx, m).crop_obb([0.5, 0.2, 0.0], R, [3.0, 1.5, 1.5])
    assert np.array_equal(c.last_indices(), o_idx) and np.array_equal(c.download(), o_pts)
    assert np.array_equal(api.Cloud(ctx, m).remove_floor().download(), orc.remove_floor(m))
    for step in (2, 3, 15):
        assert np.array_equal(api.Cloud(ctx, m).subsample(step).download(), orc.uniform_subsample(m, step), equal_nan=True)
    assert np.array_equal(api.Cloud(ctx, m[:2]).subsample(3).download(), m[:2], equal_nan=True)
    assert len(api.Cloud(ctx, np.zeros((0, 3), np.float32)).crop_radius(center, 1.0)) == 0
    assert len(api.Cloud(ctx, m).crop_radius(center + 1000, 1.0)) == 0


def test_transform_bit_identical_to_unfused_float32(api, ctx, synth, small_world):
    scan = small_world["scan"]
    T = (0.8 * synth.make_T((12.5, -3.25, 0.75), (1.0, -2.0, 33.0)) + 0.2 * synth.make_T((12.0, -3.0, 1.0), (0, 0, 30.0))).astype(np.float32)
    got = api.Cloud(ctx, scan).transform(T).download()
    x, y, z = scan[:, 0], scan[:, 1], scan[:, 2]
    exp = np.stack([T[r, 0] * x + T[r, 1] * y + T[r, 2] * z + T[r, 3] for r in range(3)], 1)   # icp_point_to_point.cpp:103-105
    assert np.array_equal(got, exp)


# ------------------------------------------------------------------ NN (a6, a9)
def test_nn_exact_against_kdtree(api, ctx, orc, small_world):
    m, scan = small_world["map"], small_world["scan"]
    q = np.concatenate([scan, scan + np.float32(0.4), scan[:500] * np.float32(1.7), np.array([[1e6, 0, 0], [np.nan, 0, 0], [-40, -40, 9]], np.float32)])
    oi, od = orc.KdTreeF(m).nn(q)
    for cell in (0.0, 0.15, 0.25, 0.6):
        mp = api.Map(ctx, api.Cloud(ctx, m), cell)
        gi, gd = mp.nn(q)
        fin = np.isfinite(od)
        assert np.array_equal(gd[fin], od[fin])               # squared distances bit-equal (L2_Simple order)
        diff = gi != oi
        assert diff.mean() < 1e-3 and np.array_equal(gd[diff], od[diff])   # only exact distance ties may differ
        assert gi[-2] == -1
        for thr in (0.5, 0.01):
            ti, td = mp.nn(q, thr)
            assert np.array_equal(ti >= 0, od < thr)          # strict "<" like icp_point_to_point.cpp:70
            assert np.array_equal(td[ti >= 0], od[ti >= 0])


def test_nn_windows_equal_searching_the_cropped_cloud(api, ctx, orc, small_world):
    m, scan = small_world["map"], small_world["scan"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    center = np.array([0.5, 0.5, 0.0], np.float32)
    crop, cidx = orc.crop_radius(m, center, 3.0)
    oi, od = orc.KdTreeF(crop).nn(scan)
    mp.window_sphere(center, 3.0)
    gi, gd = mp.nn(scan)
    assert np.array_equal(gd, od)
    assert (cidx[oi] == gi).mean() > 0.999
    R = np.eye(3)
    crop, cidx = orc.crop_obb(m, [0, 0, 0], R, [4.0, 3.0, 2.0])
    oi, od = orc.KdTreeF(crop).nn(scan)
    mp.window_obb([0, 0, 0], R, [4.0, 3.0, 2.0])
    gi, gd = mp.nn(scan)
    assert np.array_equal(gd, od)
    mp.window_none()
    assert np.array_equal(mp.nn(scan)[1], orc.KdTreeF(m).nn(scan)[1])


def test_nn_degenerate_maps(api, ctx, orc):
    empty = api.Map(ctx, api.Cloud(ctx, np.zeros((0, 3), np.float32)))
    i, d = empty.nn(np.zeros((3, 3), np.float32))
    assert (i == -1).all()
    single = api.Map(ctx, api.Cloud(ctx, np.array([[1, 2, 3]], np.float32)))
    i, d = single.nn(np.array([[1, 2, 4], [50, 50, 50]], np.float32))
    assert list(i) == [0, 0] and d[0] == 1.0
    dup = np.array([[1, 2, 3]] * 100 + [[np.nan, 0, 0]] + [[1.5, 2, 3]], np.float32)
    mp = api.Map(ctx, api.Cloud(ctx, dup))
    i, d = mp.nn(np.array([[1, 2, 3], [1.4, 2, 3]], np.float32))
    assert d[0] == 0 and i[0] < 100 and i[1] == 101
    line = np.c_[np.linspace(0, 1000, 5000), np.zeros(5000), np.zeros(5000)].astype(np.float32)   # extreme aspect ratio
    mp = api.Map(ctx, api.Cloud(ctx, line))
    q = np.array([[500.3, 7.0, -2.0], [-30, 0, 0]], np.float32)
    gi, gd = mp.nn(q)
    oi, od = orc.KdTreeF(line).nn(q)
    assert np.array_equal(gd, od)


# ------------------------------------------------------------------ normals (x2)
def test_normals_match_oracle(api, ctx, orc, small_world):
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    for radius in (0.25, 0.4):
        mp.estimate_normals(radius)
        gn, gc = mp.download_normals()
        on, oc = orc.normals_radius(m, radius)
        assert np.array_equal(gc, oc)
        dots = np.abs((gn.astype(np.float64) * on).sum(1))
        assert dots.min() > 1 - 1e-5                          # same eigenvector up to summation-order rounding
        assert np.array_equal(gn[gc < 3], on[gc < 3])         # (0,0,1) fallback


# ------------------------------------------------------------------ ICP (a10-a13, a21, x1)
REF_CASES = [dict(max_corr=0.5, iters=10, accept=0.05, eps=1e-5),      # localization_node.cpp:24-28
             dict(max_corr=0.5, iters=10, accept=0.001, eps=1e-5),
             dict(max_corr=0.5, iters=15, accept=0.001, eps=5e-3),     # lazy re-search path
             dict(max_corr=5.0, iters=80, accept=0.4, eps=1e-2)]       # strong fallback, :226-229


@pytest.mark.parametrize("case", REF_CASES)
def test_icp_ref_cpp_parity(api, ctx, orc, synth, small_world, case):
    m, scan = small_world["map"], small_world["scan"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    init = (0.95 * synth.make_T((0.01, 0.0, 0.0), (0, 0, 0.05)) + 0.05 * np.eye(4)).astype(np.float32)   # blended (non-rigid) prior
    icp = api.Icp(ctx, case["max_corr"], case["iters"], case["accept"], case["eps"])
    icp.set_target(mp)
    icp.set_source(scan)
    icp.set_initial_transformation(init)
    r = icp.align("ref_cpp")
    o64 = orc.icp_ref_cpp(scan, m, init, case["max_corr"], case["iters"], case["accept"], case["eps"], precise=True)
    o32 = orc.icp_ref_cpp(scan, m, init, case["max_corr"], case["iters"], case["accept"], case["eps"], precise=False)
    assert r["iterations"] == o64["iterations"] == o32["iterations"]
    assert r["converged"] == o64["converged"] and r["n_corr"] == o64["n_corr"]
    assert r["n_research"] - 1 == o64["n_research"]
    assert_pose_close(synth, r["T64"], o64["T"])
    assert abs(r["error"] - o64["error"]) < 1e-5
    # the float32-sequential mirror of the reference sits within its own rounding spread
    dt, dr = synth.pose_error(r["T64"], o32["T"])
    assert dt < 5e-4 and dr < 5e-5


def test_icp_ref_cpp_few_correspondences(api, ctx, small_world):
    mp = api.Map(ctx, api.Cloud(ctx, small_world["map"]), 0.25)
    icp = api.Icp(ctx, 0.5, 10, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source(small_world["scan"][:60] + np.float32(500))
    init = np.eye(4, dtype=np.float32)
    init[1, 3] = 0.5
    icp.set_initial_transformation(init)
    r = icp.align("ref_cpp")
    # ICPResult(initial_transform_) defaults: icp_point_to_point.cpp:196-200, .h:28-39
    assert r["iterations"] == 0 and not r["converged"] and r["error"] == np.float32(1e6)
    assert np.array_equal(r["T"], init) and r["flags"] & api.SF_ICP_FLAG_FEW_CORR


def test_icp_o3d_and_p2plane_parity(api, ctx, orc, synth, small_world):
    m, scan = small_world["map"], small_world["scan"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    normals, _ = mp.download_normals()
    for init in (np.eye(4), synth.make_T((0.05, 0.02, -0.01), (0.1, 0.0, 0.3))):
        icp = api.Icp(ctx, 0.5, 30, 0.05, 1e-5)
        icp.set_target(mp)
        icp.set_source(scan)
        icp.set_initial_transformation(init)
        r = icp.align("o3d_p2p")
        o = orc.icp_o3d_p2p(scan, m, init, 0.5, 30)
        assert r["iterations"] == o["iterations"] and r["converged"] == o["converged"] and r["n_corr"] == o["n_corr"]
        assert_pose_close(synth, r["T64"], o["T"], 1e-9, 1e-9)
        assert abs(r["rmse"] - o["error"]) < 1e-9 and abs(r["fitness"] - o["fitness"]) < 1e-12
        icp.set_num_iterations(20)
        r = icp.align("p2plane")
        o = orc.icp_p2plane(scan, m, normals, init, 0.5, 20)
        assert r["iterations"] == o["iterations"] == 20
        assert_pose_close(synth, r["T64"], o["T"], 1e-9, 1e-9)
    assert_pose_close(synth, r["T64"], synth.t_true(), 2e-3, 2e-4)   # and it is the right answer


def test_icp_max_iteration_cap_and_zero_iterations(api, ctx, orc, synth, small_world):
    m, scan = small_world["map"], small_world["scan"][:3000]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    for iters in (0, 1, 2):
        icp = api.Icp(ctx, 0.5, iters, 0.05, 1e-5)
        icp.set_target(mp)
        icp.set_source(scan)
        r = icp.align("o3d_p2p")
        o = orc.icp_o3d_p2p(scan, m, None, 0.5, iters)
        assert r["iterations"] == o["iterations"] == iters
        assert_pose_close(synth, r["T64"], o["T"], 1e-9, 1e-9)
        r = icp.align("ref_cpp")
        o = orc.icp_ref_cpp(scan, m, None, 0.5, iters, 0.05, 1e-5, precise=True)
        assert r["iterations"] == o["iterations"]
        assert_pose_close(synth, r["T64"], o["T"])


def test_icp_window_equals_cropped_target(api, ctx, orc, synth, small_world):
    """The reference crops the map to 10 m around the pose and indexes the crop
    (localization_node.cpp:302-303); here the crop is a window on the whole-map index."""
    m, scan = small_world["map"], small_world["scan"]
    center = np.array([0.5, -0.5, 0.0], np.float32)
    near = scan[((scan - center) ** 2).sum(1) < 9.0]
    crop, _ = orc.crop_radius(m, center, 3.5)
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.window_sphere(center, 3.5)
    icp = api.Icp(ctx, 0.5, 10, 0.01, 1e-5)
    icp.set_target(mp)
    icp.set_source(near)
    r = icp.align("ref_cpp")
    o = orc.icp_ref_cpp(near, crop, None, 0.5, 10, 0.01, 1e-5, precise=True)
    assert r["iterations"] == o["iterations"] and r["n_corr"] == o["n_corr"]
    assert_pose_close(synth, r["T64"], o["T"])
    # private target built from the cropped cloud (setTargetPointCloud semantics) gives the same
    icp2 = api.Icp(ctx, 0.5, 10, 0.01, 1e-5)
    icp2.set_target(crop)
    icp2.set_source(near)
    r2 = icp2.align("ref_cpp")
    assert r2["iterations"] == r["iterations"]
    assert_pose_close(synth, r2["T64"], r["T64"], 1e-6, 1e-7)


def test_icp_batch_graph_and_determinism(api, ctx, orc, synth, small_world):
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    normals, _ = mp.download_normals()
    scans = np.stack([synth.make_scan(m, 4000, scan_id=k)[0] for k in range(3)])
    inits = np.stack([np.eye(4), synth.make_T((0.02, 0, 0), (0, 0, 0.1)), synth.make_T((0, -0.03, 0.01), (0.1, 0, 0))])
    icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source_batch(scans)
    icp.set_initial_batch(inits)
    res = icp.align_batch("p2plane")
    for k in range(3):
        o = orc.icp_p2plane(scans[k], m, normals, inits[k], 0.5, 20)
        assert_pose_close(synth, res[k]["T64"], o["T"], 1e-9, 1e-9)
    icp.use_graph(True)
    res_g = icp.align_batch("p2plane")
    res_g2 = icp.align_batch("p2plane")
    for k in range(3):
        assert np.array_equal(res[k]["T64"], res_g[k]["T64"])     # graph replay == plain launches, bitwise
        assert np.array_equal(res_g[k]["T64"], res_g2[k]["T64"])  # run-to-run bitwise reproducible (no float atomics)
    for mode, iters in (("o3d_p2p", 30), ("ref_cpp", 10)):
        icp.set_num_iterations(iters)
        a = icp.align_batch(mode)
        b = icp.align_batch(mode)
        for k in range(3):
            assert np.array_equal(a[k]["T64"], b[k]["T64"]) and a[k]["iterations"] == b[k]["iterations"]


@pytest.mark.parametrize("offset,margin,resumes", [(0.0, 1.0, 0), (0.45, 1.0, 0), (0.45, 0.2, 1)])
def test_sharded_steps_equal_single_shot(api, ctx, orc, synth, small_world, offset, margin, resumes):
    """Two x-slabs on one GPU: summing the two exchange records by hand must reproduce the
    unsharded registration (the multi-GPU path minus RCCL).  Three scans in flight.  With a 0.2 m
    margin and a 0.45 m start offset the scans move out of the margin of their owned-query arrays:
    they stop with SF_ICP_FLAG_SHARD_STALE (on both slabs alike) and the driver resumes them."""
    import torch
    from slam_sensor_fusion_amd import sharded
    m = small_world["map"]
    scans = np.stack([synth.make_scan(m, 5000, scan_id=s)[0] for s in range(3)])
    inits = np.stack([synth.make_T((offset, 0.0, 0.0), (0, 0, 0)), synth.make_T((0.0, 0.0, 0.0), (0, 0, 0)), synth.make_T((0.0, offset / 2, 0.0), (0, 0, 0))])
    edges = sharded.slab_edges(m[:, 0], 2)
    full = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    full.estimate_normals(0.25)
    icp0 = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
    icp0.set_target(full)
    icp0.set_source_batch(scans)
    icp0.set_initial_batch(inits)
    ref = icp0.align_batch("p2plane")
    maps, icps = [], []
    xb = [torch.zeros(3 * 32, dtype=torch.float64, device="cuda") for _ in range(2)]
    for r in range(2):
        keep = sharded.slab_select(m, edges, r, halo=0.5 + 0.25 + 0.25)
        mp = api.Map(ctx, api.Cloud(ctx, m[keep]), 0.25)
        mp.estimate_normals(0.25)
        icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
        icp.set_target(mp)
        icp.set_source_batch(scans)
        icp.set_initial_batch(inits)
        icp.set_shard(float(edges[r]), float(edges[r + 1]))
        icp.set_shard_margin(margin)
        icp.set_exchange_buffer(xb[r].data_ptr(), 3 * 256)
        maps.append(mp)
        icps.append(icp)

    class BothSlabs:
        """The two ranks in lock step; the 'all-reduce' is the sum of the two exchange buffers."""

        def step_begin(self, mode, first):
            for icp in icps:
                icp.step_begin(mode, first)

        def step_end(self, mode, last):
            for icp in icps:
                icp.step_end(mode, last)

        def fetch_results(self):
            a, b = icps[0].fetch_results(), icps[1].fetch_results()
            for x, y in zip(a, b):
                assert np.array_equal(x["T64"], y["T64"]) and x["flags"] == y["flags"] and x["iterations"] == y["iterations"]
            return a

    owned = []

    def allreduce():
        ctx.synchronize()
        total = xb[0] + xb[1]
        owned.append(total[0::32].cpu().numpy().copy())
        xb[0].copy_(total)
        xb[1].copy_(total)
        torch.cuda.synchronize()

    drv = sharded.ShardedIcp(BothSlabs(), "p2plane", 20, allreduce)
    res = drv.align()
    assert drv.resumes >= resumes and (resumes > 0 or drv.resumes == 0)
    for k in range(3):
        assert res[k]["iterations"] == 20 and res[k]["n_corr"] == ref[k]["n_corr"] and res[k]["flags"] == 0
        assert_pose_close(synth, res[k]["T64"], ref[k]["T64"], 1e-9, 1e-10)
    final = [o for o in owned if o.any()][-1]
    assert all(final[k] == ref[k]["n_corr"] or final[k] == 0 for k in range(3))   # every query owned by exactly one slab


# ------------------------------------------------------------------ committed golden vectors
def test_gpu_against_golden(api, ctx, synth):
    g = load_golden("registration_small.npz")
    c = api.Cloud(ctx, g["raw"])
    c.voxel_downsample(0.1, "pcl")
    assert np.array_equal(c.voxel_point_ids(), g["vox_point_ids"]) and np.array_equal(c.voxel_out_ids(), g["vox_out_ids"])
    assert np.array_equal(c.download(), g["map"])
    fin = g["raw"][np.isfinite(g["raw"]).all(1)]
    c2 = api.Cloud(ctx, fin)
    c2.voxel_downsample(0.1, "o3d")
    assert np.array_equal(c2.voxel_out_ids().reshape(-1, 3), g["o3d_out_ijk"]) and np.array_equal(c2.voxel_out_means_f64(), g["o3d_means"])
    mp = api.Map(ctx, c, 0.25)
    gi, gd = mp.nn(g["nn_queries"])
    assert np.array_equal(gd, g["nn_d2"]) and (gi == g["nn_idx"]).mean() > 0.999
    mp.estimate_normals(0.3)
    gn, gc = mp.download_normals()
    assert np.array_equal(gc, g["normal_counts"]) and np.abs((gn * g["normals"]).sum(1)).min() > 1 - 1e-5
    assert np.array_equal(np.sort(api.Cloud(ctx, g["map"]).crop_radius([0.3, -0.2, 0.1], 0.8, sorted=True).last_indices()), np.sort(g["crop_radius_idx"]))
    assert np.array_equal(api.Cloud(ctx, g["map"]).crop_radius([0.3, -0.2, 0.1], 0.8, sorted=True).last_indices(), g["crop_radius_idx"])
    assert np.array_equal(api.Cloud(ctx, g["map"]).crop_aabb([0, -0.5, 0], [1.0, 0.5, 0.5]).last_indices(), g["crop_aabb_idx"])
    assert np.array_equal(api.Cloud(ctx, g["map"]).crop_obb([0.1, 0.2, 0.0], g["obb_R"], [1.5, 0.8, 0.8]).last_indices(), g["crop_obb_idx"])
    mp.set_normals(g["normals"])
    icp = api.Icp(ctx, 0.5, 10, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source(g["scan"])
    r = icp.align("ref_cpp")
    assert [r["iterations"], int(r["converged"]), r["n_corr"]] == list(g["ref64_meta"][:3])
    assert_pose_close(synth, r["T64"], g["ref64_T"])
    icp.set_num_iterations(30)
    r = icp.align("o3d_p2p")
    assert r["iterations"] == g["o3d_meta"][0]
    assert_pose_close(synth, r["T64"], g["o3d_T"], 1e-9, 1e-9)
    icp.set_num_iterations(20)
    r = icp.align("p2plane")
    assert_pose_close(synth, r["T64"], g["pl_T"], 1e-9, 1e-9)


@pytest.mark.parametrize("mode", ["o3d_p2p", "p2plane"])
def test_query_order_is_free(api, ctx, synth, small_world, mode):
    """SF_ORDER_CELL walks every scan in map-cell order (sorted on the device at the start of the
    alignment).  Correspondences and per-point terms do not depend on the order, so counts are
    equal and poses agree to the rounding of the float64 record sums; a given order is bitwise
    reproducible, with and without hipGraph replay; a shuffled copy of a scan gives the same answer."""
    mp = api.Map(ctx, api.Cloud(ctx, small_world["map"]), 0.25)
    mp.estimate_normals(0.25)
    rng = np.random.default_rng(5)
    scans = np.stack([synth.make_scan(small_world["map"], 6000, scan_id=s)[0] for s in range(3)])
    scans[1, ::97] = np.nan
    out = {}
    for order in ("as_given", "cell"):
        for graph in (False, True):
            icp = api.Icp(ctx, 0.5, 12, 0.05, 1e-5)
            icp.set_target(mp)
            icp.set_query_order(order)
            icp.use_graph(graph)
            icp.set_source_batch(scans)
            out[order, graph] = [icp.align_batch(mode) for _ in range(2)]
    for order in ("as_given", "cell"):
        ref = out[order, False][0]
        for rs in (out[order, False][1], out[order, True][0], out[order, True][1]):
            for a, b in zip(ref, rs):
                assert np.array_equal(a["T64"], b["T64"]) and a["rmse"] == b["rmse"] and a["n_corr"] == b["n_corr"]
    for a, b in zip(out["as_given", False][0], out["cell", False][0]):
        assert a["n_corr"] == b["n_corr"] and a["iterations"] == b["iterations"] and a["flags"] == b["flags"]
        assert np.abs(a["T64"] - b["T64"]).max() < 1e-12 and abs(a["rmse"] - b["rmse"]) < 1e-13
    icp = api.Icp(ctx, 0.5, 12, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_query_order("cell")
    icp.set_source_batch(np.stack([s[rng.permutation(len(s))] for s in scans]))
    for a, b in zip(out["cell", False][0], icp.align_batch(mode)):
        assert a["n_corr"] == b["n_corr"] and np.abs(a["T64"] - b["T64"]).max() < 1e-12


@pytest.mark.parametrize("mode", ["o3d_p2p", "p2plane"])
def test_nn_reuse_is_exact(api, ctx, synth, small_world, mode):
    """Neighbour reuse (sf_icp_set_nn_reuse) skips the search of a query whose neighbour provably
    cannot have changed.  The registration must be bit-identical with it on and off — with and
    without a map window, for cell-ordered and as-given queries, for scans that start far away,
    contain NaNs or partly miss the map."""
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    scans = np.stack([synth.make_scan(m, 6000, scan_id=s)[0] for s in range(4)])
    scans[1, ::53] = np.nan
    scans[2, :2000] += np.float32(30.0)                              # a third of the scan is nowhere near the map
    inits = np.stack([np.eye(4), np.eye(4), np.eye(4), synth.make_T((0.4, -0.2, 0.1), (0.0, 0.0, 1.5))])
    for window in (False, True):
        if window:
            mp.window_sphere([0.0, 0.0, 0.0], 4.0)
        else:
            mp.window_none()
        for order in ("as_given", "cell"):
            out = []
            for reuse in (False, True):
                icp = api.Icp(ctx, 0.5, 25, 0.05, 1e-5)
                icp.set_target(mp)
                icp.set_query_order(order)
                icp.set_nn_reuse(reuse)
                icp.set_source_batch(scans)
                icp.set_initial_batch(inits)
                out.append(icp.align_batch(mode))
            for a, b in zip(*out):
                assert a["n_corr"] == b["n_corr"] and a["iterations"] == b["iterations"] and a["flags"] == b["flags"]
                assert np.array_equal(a["T64"], b["T64"]) and a["rmse"] == b["rmse"] and a["fitness"] == b["fitness"]
    mp.window_none()


def test_nn_reuse_fuzz(api, ctx, synth):
    """Randomised scenes for the neighbour-reuse certificate: clustered and duplicated map points,
    points on cell faces, sparse maps, several cell sizes, large and tiny start offsets — the
    registration must not depend on the switch in a single bit."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SF_FUZZ_SEED", "77")))
    for trial in range(int(os.environ.get("SF_FUZZ_TRIALS", "24"))):
        kind = trial % 4
        n_map = int(rng.integers(2_000, 60_000))
        if kind == 0:                                   # uniform volume
            m = rng.uniform(-6, 6, (n_map, 3))
        elif kind == 1:                                 # clusters + exact duplicates
            c = rng.uniform(-6, 6, (40, 3))
            m = c[rng.integers(0, 40, n_map)] + rng.normal(0, 0.15, (n_map, 3))
            k7 = len(m[1::7])
            m[::7][:k7] = m[1::7]
        elif kind == 2:                                 # points snapped to a lattice: ties and cell faces
            m = np.round(rng.uniform(-6, 6, (n_map, 3)) * 8) / 8
        else:                                           # planes
            m = rng.uniform(-6, 6, (n_map, 3))
            m[: n_map // 2, 2] = 0.0
            m[n_map // 2:, 0] = 2.0
        m = m.astype(np.float32)
        cell = float(rng.choice([0.0, 0.15, 0.25, 0.5]))
        mp = api.Map(ctx, api.Cloud(ctx, m), cell)
        mp.estimate_normals(0.4)
        n_scan = int(rng.integers(300, 5000))
        scans, inits = [], []
        for s in range(3):
            T = synth.make_T(rng.normal(0, 0.1 if s else 0.01, 3), rng.normal(0, 1.0 if s else 0.05, 3))
            idx = rng.integers(0, len(m), n_scan)
            p = m[idx].astype(np.float64) + rng.normal(0, 0.02, (n_scan, 3))
            Ti = np.linalg.inv(T)
            scans.append((p @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32))
            inits.append(np.eye(4))
        scans = np.stack(scans)
        for mode in ("p2plane", "o3d_p2p"):
            thr = float(rng.choice([0.3, 0.5, 1.0]))
            res = []
            for reuse in (False, True):
                icp = api.Icp(ctx, thr, 15, 0.05, 1e-5)
                icp.set_target(mp)
                icp.set_nn_reuse(reuse)
                icp.set_query_order("cell" if trial % 2 else "as_given")
                icp.set_source_batch(scans)
                icp.set_initial_batch(np.stack(inits))
                res.append(icp.align_batch(mode))
            for a, b in zip(*res):
                assert a["n_corr"] == b["n_corr"] and a["iterations"] == b["iterations"] and a["flags"] == b["flags"], (trial, mode)
                assert np.array_equal(a["T64"], b["T64"], equal_nan=True) and (a["rmse"] == b["rmse"] or (np.isnan(a["rmse"]) and np.isnan(b["rmse"]))), (trial, mode)


def test_nn_reuse_fuzz_wide_scans(api, ctx, synth):
    """The same property for scans above 131 072 points, which take the other summation order (two queries per lane, one
    wave reduction) and, from the fifth launch on, the wave-level verify path with all loads in flight: clustered,
    duplicated, lattice and planar maps with extents up to 40 m (the motion bound's lever arm), rotations of several
    degrees in the start poses; reuse on == reuse off in every bit, graph replay == plain launches."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SF_FUZZ_SEED", "78")))
    for trial in range(max(4, int(os.environ.get("SF_FUZZ_TRIALS", "24")) // 6)):
        kind = trial % 4
        n_map = int(rng.integers(100_000, 400_000))
        ext = float(rng.choice([6.0, 12.0, 20.0]))
        if kind == 0:
            m = rng.uniform(-ext, ext, (n_map, 3)) * [1.0, 1.0, 0.2]
        elif kind == 1:
            c = rng.uniform(-ext, ext, (400, 3))
            m = c[rng.integers(0, 400, n_map)] + rng.normal(0, 0.2, (n_map, 3))
            k7 = len(m[1::7])
            m[::7][:k7] = m[1::7]
        elif kind == 2:
            m = np.round(rng.uniform(-ext, ext, (n_map, 3)) * [1.0, 1.0, 0.2] * 8) / 8
        else:
            m = rng.uniform(-ext, ext, (n_map, 3))
            m[: n_map // 2, 2] = 0.0
            m[n_map // 2:, 0] = 2.0
        m = m.astype(np.float32)
        mp = api.Map(ctx, api.Cloud(ctx, m), float(rng.choice([0.0, 0.25, 0.5])))
        mp.estimate_normals(0.4)
        n_scan = int(rng.integers(131_073, 160_000))
        scans, inits = [], []
        for s in range(3):
            T = synth.make_T(rng.normal(0, 0.08 if s else 0.01, 3), rng.normal(0, 0.6 if s else 0.05, 3))
            idx = rng.integers(0, len(m), n_scan)
            p = m[idx].astype(np.float64) + rng.normal(0, 0.02, (n_scan, 3))
            Ti = np.linalg.inv(T)
            scans.append((p @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32))
            inits.append(np.eye(4))
        scans = np.stack(scans)
        for mode in ("p2plane", "o3d_p2p"):
            thr = float(rng.choice([0.3, 0.5, 1.0]))
            res = []
            for reuse, graph in ((False, False), (True, False), (True, True)):
                icp = api.Icp(ctx, thr, 12, 0.05, 1e-5)
                icp.set_target(mp)
                icp.set_nn_reuse(reuse)
                icp.set_freeze(False)       # bit for bit: the frozen-pair evaluation sums in another order (tests/test_gpu_freeze.py and below)
                icp.use_graph(graph)
                icp.set_query_order("cell" if trial % 2 else "as_given")
                icp.set_source_batch(scans)
                icp.set_initial_batch(np.stack(inits))
                res.append(icp.align_batch(mode))
                if mode == "p2plane" and reuse and not graph:   # frozen pairs on the same adversarial maps: same pairs, sums equal to rounding
                    icp.set_freeze(True)
                    fz = icp.align_batch(mode)
                    for a, b in zip(res[-1], fz):
                        assert a["n_corr"] == b["n_corr"] and a["iterations"] == b["iterations"] and a["flags"] == b["flags"], (trial, mode)
                        assert np.allclose(a["T64"], b["T64"], rtol=0, atol=1e-9, equal_nan=True), (trial, np.abs(a["T64"] - b["T64"]).max())
                icp.close()
            for other in res[1:]:
                for a, b in zip(res[0], other):
                    assert a["n_corr"] == b["n_corr"] and a["iterations"] == b["iterations"] and a["flags"] == b["flags"], (trial, mode)
                    assert np.array_equal(a["T64"], b["T64"], equal_nan=True) and (a["rmse"] == b["rmse"] or (np.isnan(a["rmse"]) and np.isnan(b["rmse"]))), (trial, mode)


def test_switches_between_alignments_with_graph_replay(api, ctx, synth, small_world):
    """One sf_icp object, hipGraph replay on, switches flipped between alignments: the captured
    launch list carries the neighbour-cache and query-array pointers, so a flip must re-capture."""
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    scans = np.stack([synth.make_scan(m, 5000, scan_id=s)[0] for s in range(2)])
    icp = api.Icp(ctx, 0.5, 15, 0.05, 1e-5)
    icp.set_target(mp)
    icp.use_graph(True)
    icp.set_source_batch(scans)
    ref = None
    for reuse, order in [(True, "cell"), (False, "cell"), (True, "cell"), (True, "as_given"), (False, "as_given"), (True, "cell")]:
        icp.set_nn_reuse(reuse)
        icp.set_query_order(order)
        for _ in range(2):
            r = icp.align_batch("p2plane")
            key = order
            if ref is None:
                ref = {}
            if key not in ref:
                ref[key] = r
            for a, b in zip(ref[key], r):
                assert np.array_equal(a["T64"], b["T64"]) and a["n_corr"] == b["n_corr"] and a["rmse"] == b["rmse"]
    # stepping API without sharding: same result as the one-shot alignment
    icp.set_query_order("cell")
    icp.set_nn_reuse(True)
    for k in range(15):
        icp.step_begin("p2plane", first=1 if k == 0 else 0)
        icp.step_end("p2plane", last=(k == 14))
    stepped = icp.fetch_results()
    for a, b in zip(ref["cell"], stepped):
        assert np.array_equal(a["T64"], b["T64"]) and a["iterations"] == b["iterations"]


def test_nn_exact_fuzz(api, ctx, orc):
    """sf_map_nn (the ICP kernel's wave-cooperative search) against the oracle's exact kd-tree on
    uniform, clustered, lattice and planar maps, several cell sizes, thresholds and query noises, with
    coordinates up to 3 km from the origin (stresses the float32 slack of the pruning bounds): the hit
    set and every squared distance must be bit-identical."""
    rng = np.random.default_rng(2024)
    for trial in range(16):
        kind = trial % 4
        n = int(rng.integers(5_000, 120_000))
        if kind == 0:
            m = rng.uniform(-20, 20, (n, 3))
        elif kind == 1:
            c = rng.uniform(-20, 20, (60, 3))
            m = c[rng.integers(0, 60, n)] + rng.normal(0, 0.3, (n, 3))
        elif kind == 2:
            m = np.round(rng.uniform(-20, 20, (n, 3)) * 8) / 8
        else:
            m = rng.uniform(-20, 20, (n, 3))
            m[: n // 2, 2] = 0.0
        m = m.astype(np.float32) + np.float32(rng.choice([0.0, 500.0, 3000.0]))
        mp = api.Map(ctx, api.Cloud(ctx, m), float(rng.choice([0.0, 0.1, 0.25, 0.5, 1.0])))
        q = (m[rng.integers(0, n, 10_000)] + rng.normal(0, rng.choice([0.001, 0.05, 0.5]), (10_000, 3))).astype(np.float32)
        thr = float(rng.choice([0.25, 1.0, 1e9]))
        gi, gd = mp.nn(q, thr)
        oi, od = orc.KdTreeF(m).nn(q)
        ok = od < thr
        assert np.array_equal(gi >= 0, ok), trial
        assert np.array_equal(gd[ok], od[ok]), trial
        uniq = ok & (m[np.maximum(gi, 0)] == m[np.maximum(oi, 0)]).all(1)
        assert uniq.sum() == ok.sum(), trial                 # same point up to exact duplicates
