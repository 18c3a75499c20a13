"""GPU parity tests added in round 2: the hipGraph cache key, the int64-index voxel
grid (SF_VOXEL_PCL64), the covariance export of the normal estimation (SURVEY x2 "+6 cov"), the
checked PointCloud2 entry point.  All through the C ABI, against the oracle."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_graph_replay_follows_parameters_and_map_rebuilds(api, ctx, synth, small_world):
    """A captured launch list bakes thresholds, IcpParams and the SfGrid (by value) into its kernel
    arguments.  With hipGraph replay on, changing a parameter through the setters or rebuilding the
    SAME sf_map handle in place (map growth: the allocation usually survives, its content does not)
    must re-capture: every alignment equals the graph-off run bit for bit."""
    m = small_world["map"]
    scans = np.stack([synth.make_scan(m, 4000, scan_id=s)[0] for s in range(2)])
    cloud = api.Cloud(ctx, m[: len(m) - 3000])
    mp = api.Map(ctx, cloud, 0.25)
    mp.estimate_normals(0.25)
    ref_map = api.Map(ctx, api.Cloud(ctx, m[: len(m) - 3000]), 0.25)
    ref_map.estimate_normals(0.25)

    def pair():
        out = []
        for graph, target in ((True, mp), (False, ref_map)):
            icp = api.Icp(ctx, 5.0, 12, 0.4, 1e-2)
            icp.set_target(target)
            icp.use_graph(graph)
            icp.set_source_batch(scans)
            out.append(icp)
        return out

    g, p = pair()

    def same(mode):
        a, b = g.align_batch(mode), p.align_batch(mode)
        for x, y in zip(a, b):
            assert np.array_equal(x["T64"], y["T64"]) and x["iterations"] == y["iterations"] and x["n_corr"] == y["n_corr"] and x["error"] == y["error"], mode

    for mode in ("ref_cpp", "o3d_p2p", "p2plane"):
        same(mode)                                       # capture at the coarse parameters (localization_node.cpp:226-229)
        for icp in (g, p):                               # coarse -> fine switch, :24-28
            icp.set_max_correspondence_dist(0.5)
            icp.set_transformation_epsilon(1e-5)
            icp.set_acceptable_mean_error(0.05)
        same(mode)
        same(mode)                                       # and a plain replay
        for icp in (g, p):
            icp.set_max_correspondence_dist(5.0)
            icp.set_transformation_epsilon(1e-2)
            icp.set_acceptable_mean_error(0.4)
    # rebuild the same Map handle with a slightly larger cloud (sf_cloud_append use case): pointer may survive
    bigger = api.Cloud(ctx, m)
    mp.build(bigger, 0.25)
    mp.estimate_normals(0.25)
    ref_map = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    ref_map.estimate_normals(0.25)
    p.set_target(ref_map)
    for mode in ("p2plane", "o3d_p2p", "ref_cpp"):
        same(mode)
    mp.estimate_normals(0.4)                             # new normals in the same buffer
    ref_map.estimate_normals(0.4)
    same("p2plane")


def test_voxel_pcl64_bit_exact_and_past_int32(api, ctx, orc, small_world):
    raw = small_world["raw"].copy()
    raw[11] = [np.nan, 0, 0]
    c = api.Cloud(ctx, raw)
    assert c.voxel_downsample(0.1, "pcl64") == 0
    ds, vidx, ovox = orc.voxel_pcl64(raw, 0.1)
    assert np.array_equal(c.voxel_point_ids64(), vidx) and np.array_equal(c.voxel_out_ids64(), ovox) and np.array_equal(c.download(), ds)
    c32 = api.Cloud(ctx, raw)
    c32.voxel_downsample(0.1, "pcl")
    assert np.array_equal(c32.download(), ds) and np.array_equal(c32.voxel_point_ids().astype(np.int64), vidx)   # same arithmetic where int32 suffices
    with pytest.raises(api.SlamFusionError):
        c.voxel_point_ids()                                   # the 32-bit getter refuses 64-bit ids
    rng = np.random.default_rng(0)
    big = np.concatenate([rng.uniform(0, 2, (20000, 3)), rng.uniform(0, 2, (20000, 3)) + [2500.0, 1800.0, 900.0]]).astype(np.float32)
    c = api.Cloud(ctx, big)
    assert c.voxel_downsample(0.1, "pcl") == api.SF_FLAG_VOXEL_OVERFLOW and len(c) == len(big)     # PCL gives up
    assert c.voxel_downsample(0.1, "pcl64") == 0
    ds, vidx, ovox = orc.voxel_pcl64(big, 0.1)
    assert ovox.max() > 2**31
    assert np.array_equal(c.voxel_point_ids64(), vidx) and np.array_equal(c.voxel_out_ids64(), ovox) and np.array_equal(c.download(), ds)


def test_normals_covariance_matches_oracle(api, ctx, orc, small_world):
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    with pytest.raises(api.SlamFusionError):
        mp.download_covariances()
    for radius in (0.25, 0.4):
        mp.estimate_normals(radius, covariance=True)
        gn, gc = mp.download_normals()
        cov = mp.download_covariances()
        on, oc, ocov = orc.normals_radius_cov(m, radius)
        assert np.array_equal(gc, oc)
        scale = np.abs(ocov).max()
        assert np.abs(cov - ocov).max() <= 1e-12 * scale            # same float64 sums up to the order of the neighbours
        assert np.array_equal(cov[gc < 3], np.zeros_like(cov[gc < 3]))
        # the normal is the eigenvector of the exported covariance's smallest eigenvalue
        k = np.nonzero(gc >= 6)[0][::97]
        C = np.zeros((len(k), 3, 3))
        C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2] = cov[k].T
        C = C + np.transpose(C, (0, 2, 1)) - np.einsum("nij,ij->nij", C, np.eye(3))
        w, v = np.linalg.eigh(C)
        well = (w[:, 1] - w[:, 0]) > 1e-6 * w[:, 2]
        assert (np.abs(np.einsum("ni,ni->n", v[:, :, 0], gn[k].astype(np.float64)))[well] > 1 - 1e-5).all()
    mp.estimate_normals(0.25)                                       # without the flag the export is gone again
    with pytest.raises(api.SlamFusionError):
        mp.download_covariances()


def test_pointcloud2_message_contract(api, ctx):
    rng = np.random.default_rng(4)
    xyz = rng.normal(size=(3 * 700, 3)).astype(np.float32)
    fields = [SimpleNamespace(name="x", offset=0, datatype=7), SimpleNamespace(name="y", offset=4, datatype=7), SimpleNamespace(name="z", offset=8, datatype=7)]
    # organised cloud with padded rows: height 3, width 700, row_step > width * point_step
    step, width, height = 16, 700, 3
    row_step = width * step + 40
    buf = np.zeros((height, row_step), np.uint8)
    rows = xyz.reshape(height, width, 3)
    for r in range(height):
        pts = np.zeros((width, step), np.uint8)
        pts[:, :12] = rows[r].copy().view(np.uint8).reshape(-1, 12)
        buf[r, : width * step] = pts.reshape(-1)
    msg = SimpleNamespace(width=width, height=height, point_step=step, row_step=row_step, is_bigendian=False, data=buf.tobytes(), fields=fields)
    assert np.array_equal(api.Cloud(ctx).from_pointcloud2(msg).download(), xyz)
    # FLOAT64 fields are rounded to float32
    f64 = rng.normal(size=(500, 3))
    f8 = [SimpleNamespace(name=n, offset=8 * k, datatype=8) for k, n in enumerate("xyz")]
    msg64 = SimpleNamespace(width=500, height=1, point_step=24, row_step=0, is_bigendian=False, data=f64.tobytes(), fields=f8)
    assert np.array_equal(api.Cloud(ctx).from_pointcloud2(msg64).download(), f64.astype(np.float32))
    # short buffer, big-endian payload, mixed or integer datatypes: refused, nothing is read out of bounds
    short = SimpleNamespace(width=width, height=height, point_step=step, row_step=row_step, is_bigendian=False, data=buf.tobytes()[:-100], fields=fields)
    with pytest.raises(api.SlamFusionError):
        api.Cloud(ctx).from_pointcloud2(short)
    with pytest.raises(api.SlamFusionError):
        api.Cloud(ctx).from_pointcloud2(SimpleNamespace(width=500, height=1, point_step=24, row_step=0, is_bigendian=True, data=f64.tobytes(), fields=f8))
    mixed = [SimpleNamespace(name="x", offset=0, datatype=7), SimpleNamespace(name="y", offset=4, datatype=8), SimpleNamespace(name="z", offset=12, datatype=7)]
    with pytest.raises(api.SlamFusionError):
        api.Cloud(ctx).from_pointcloud2(SimpleNamespace(width=10, height=1, point_step=24, row_step=0, is_bigendian=False, data=bytes(240), fields=mixed))
    ints = [SimpleNamespace(name=n, offset=4 * k, datatype=5) for k, n in enumerate("xyz")]
    with pytest.raises(api.SlamFusionError):
        api.Cloud(ctx).from_pointcloud2(SimpleNamespace(width=10, height=1, point_step=12, row_step=0, is_bigendian=False, data=bytes(120), fields=ints))
    empty = SimpleNamespace(width=0, height=0, point_step=12, row_step=0, is_bigendian=False, data=b"", fields=fields)
    assert len(api.Cloud(ctx).from_pointcloud2(empty)) == 0


def test_per_scan_preprocessing_reuses_buffers(api, ctx, orc, small_world):
    """The crops ping-pong between two persistent buffers (no hipMalloc per scan): a sequence of
    crops on one cloud object, repeated over several 'scans', must keep matching the oracle."""
    m = small_world["map"]
    c = api.Cloud(ctx)
    rng = np.random.default_rng(1)
    for k in range(4):
        pts = (m[rng.permutation(len(m))[: 20000 + 1000 * k]] + np.float32(0.01 * k)).astype(np.float32)
        c.upload(pts)
        exp = orc.uniform_subsample(pts, 2)
        c.subsample(2)
        assert np.array_equal(c.download(), exp)
        exp, idx = orc.crop_radius(exp, np.zeros(3, np.float32), 4.0)
        c.crop_radius(np.zeros(3, np.float32), 4.0, sorted=False)
        assert np.array_equal(c.last_indices(), np.sort(idx))
        exp = exp[np.argsort(idx, kind="stable")]
        assert np.array_equal(c.download(), exp)
        exp2 = orc.remove_floor(exp)
        c.remove_floor()
        assert np.array_equal(c.download(), exp2)


def test_single_scan_ref_cpp_graph_survives_changing_counts_and_windows(api, ctx, orc, synth, small_world):
    """The per-scan path: REF_CPP with ONE scan reads the scan's point count and the map window from device memory, so
    one captured launch list serves a stream of scans whose sizes differ (same count rounded up to 4096) while the
    window moves -- and every result equals the plain-launch alignment bit for bit, and the oracle's."""
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    g = api.Icp(ctx, 0.5, 10, 0.001, 1e-5)
    p = api.Icp(ctx, 0.5, 10, 0.001, 1e-5)
    for icp in (g, p):
        icp.set_target(mp)
        icp.set_fused(False)                                     # the launch list is what a graph captures (the single-launch form: next test)
    g.use_graph(True)
    rng = np.random.default_rng(8)
    for k, n in enumerate((9000, 8700, 9500, 8200, 9900)):
        scan = synth.make_scan(m, n, scan_id=30 + k)[0]
        center = np.array([0.3 * k - 0.5, 0.2 * k, 0.0], np.float32)
        mp.window_sphere(center, 4.0 + 0.1 * k)
        init = synth.make_T(rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3)).astype(np.float32)
        res = []
        for icp in (g, p):
            icp.set_source(scan)
            icp.set_initial_transformation(init)
            res.append(icp.align("ref_cpp"))
        a, b = res
        assert np.array_equal(a["T64"], b["T64"]) and a["iterations"] == b["iterations"] and a["n_corr"] == b["n_corr"] and a["error"] == b["error"]
        crop, _ = orc.crop_radius(m, center, 4.0 + 0.1 * k)
        o = orc.icp_ref_cpp(scan, crop, init, 0.5, 10, 0.001, 1e-5, precise=True)
        assert a["iterations"] == o["iterations"] and a["n_corr"] == o["n_corr"]
        dt, dr = synth.pose_error(a["T64"], o["T"])
        assert dt < 1e-4 and dr < 1e-5
    captures, launches = g.graph_counts()
    assert captures == 1 and launches == 5                     # 8 200 .. 9 900 points all round up to 12 288
    g.set_source(synth.make_scan(m, 3000, scan_id=99)[0])      # another capacity class: one more capture
    g.align("ref_cpp")
    assert g.graph_counts() == (2, 6)


def test_ref_cpp_single_launch_equals_launch_list(api, ctx, orc, synth, small_world):
    """REF_CPP with every workgroup resident at once runs as ONE launch (k_ref_fused: grid barriers instead of kernel
    boundaries, the controller evaluated by every workgroup): bit-identical to the launch list in every parameter case of
    the parity test (early accept, iteration cap, lazy re-search, strong fallback), with a sphere / box window, for a
    batch of scans of which one dies at once, and for a stream of scans through one object."""
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    cases = [dict(max_corr=0.5, iters=10, accept=0.05, eps=1e-5), dict(max_corr=0.5, iters=10, accept=0.001, eps=1e-5),
             dict(max_corr=0.5, iters=15, accept=0.001, eps=5e-3), dict(max_corr=5.0, iters=80, accept=0.4, eps=1e-2), dict(max_corr=0.5, iters=0, accept=0.001, eps=1e-5)]
    keys = ("iterations", "converged", "n_corr", "n_research", "flags", "error")
    scans = np.stack([synth.make_scan(m, 7000, scan_id=50 + k)[0] for k in range(3)])
    scans[2] += np.float32(400.0)                                   # nothing within reach: < 10 correspondences, initial T back
    inits = np.stack([synth.make_T((0.02 * k, -0.01, 0.0), (0, 0.02, 0.1 * k)) for k in range(3)])
    for case in cases:
        for window in (None, "sphere", "box"):
            if window == "sphere":
                mp.window_sphere(np.array([0.2, -0.1, 0.0], np.float32), 4.5)
            elif window == "box":
                mp.window_obb(np.array([0.1, 0.2, 0.0]), synth.make_T((0, 0, 0), (0, 0, 20.0))[:3, :3], np.array([4.0, 3.5, 2.0]))
            else:
                mp.window_none()
            out = []
            for fused in (True, False):
                icp = api.Icp(ctx, case["max_corr"], case["iters"], case["accept"], case["eps"])
                icp.set_fused(fused)
                icp.set_target(mp)
                icp.set_source_batch(scans)
                icp.set_initial_batch(inits)
                r = icp.align_batch("ref_cpp")
                assert icp.fused_count() == (1 if fused else 0)
                icp.set_source(scans[1][:4321])                    # one scan, a count that is no multiple of anything
                icp.set_initial_transformation(inits[1].astype(np.float32))
                r.append(icp.align("ref_cpp"))
                assert icp.fused_count() == (2 if fused else 0)
                out.append(r)
            for a, b in zip(*out):
                assert np.array_equal(a["T64"], b["T64"]) and all(a[k] == b[k] for k in keys), (case, window)
            assert out[0][2]["flags"] & api.SF_ICP_FLAG_FEW_CORR and out[0][2]["iterations"] == 0
    mp.window_none()
    icp = api.Icp(ctx, 0.5, 10, 0.001, 1e-5)                        # a stream through one object: the barrier counters return to zero
    icp.set_target(mp)
    for k in range(6):
        scan = synth.make_scan(m, 5000 + 700 * k, scan_id=70 + k)[0]
        init = synth.make_T((0.01 * k, 0.0, 0.0), (0, 0, 0.05 * k)).astype(np.float32)
        icp.set_source(scan)
        icp.set_initial_transformation(init)
        r = icp.align("ref_cpp")
        o = orc.icp_ref_cpp(scan, m, init, 0.5, 10, 0.001, 1e-5, precise=True)
        assert r["iterations"] == o["iterations"] and r["n_corr"] == o["n_corr"]
        dt, dr = synth.pose_error(r["T64"], o["T"])
        assert dt < 1e-4 and dr < 1e-5
    assert icp.fused_count() == 6


def test_o3d_and_p2plane_single_launch_equals_launch_list(api, ctx, orc, synth, small_world):
    """The float64 modes as ONE launch (k_icp_fused: the neighbour cache in registers, grid barriers, the solve evaluated
    by every workgroup): bit-identical to the launch list -- with and without neighbour reuse, with a sphere / box window,
    for a batch in which one scan has nothing within reach, for a single scan of odd size -- and equal to the oracle."""
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    normals, _ = mp.download_normals()
    keys = ("iterations", "converged", "n_corr", "n_research", "flags", "rmse", "fitness")
    scans = np.stack([synth.make_scan(m, 7000, scan_id=60 + k)[0] for k in range(3)])
    scans[2] += np.float32(400.0)
    inits = np.stack([synth.make_T((0.02 * k, -0.01, 0.0), (0, 0.02, 0.1 * k)) for k in range(3)])
    for mode, iters in (("o3d_p2p", 30), ("p2plane", 12), ("o3d_p2p", 0)):
        for window in (None, "sphere", "box"):
            if window == "sphere":
                mp.window_sphere(np.array([0.2, -0.1, 0.0], np.float32), 4.5)
            elif window == "box":
                mp.window_obb(np.array([0.1, 0.2, 0.0]), synth.make_T((0, 0, 0), (0, 0, 20.0))[:3, :3], np.array([4.0, 3.5, 2.0]))
            else:
                mp.window_none()
            out = []
            for fused, reuse in ((True, True), (False, True), (True, False), (False, False)):
                icp = api.Icp(ctx, 0.5, iters, 0.05, 1e-5)
                icp.set_fused(fused)
                icp.set_nn_reuse(reuse)
                icp.set_target(mp)
                icp.set_source_batch(scans)
                icp.set_initial_batch(inits)
                r = icp.align_batch(mode)
                assert icp.fused_count() == (1 if fused else 0)
                icp.set_source(scans[1][:4321])
                icp.set_initial_transformation(inits[1])
                r.append(icp.align(mode))
                assert icp.fused_count() == (2 if fused else 0)
                out.append(r)
            for other in out[1:]:
                for a, b in zip(out[0], other):
                    assert np.array_equal(a["T64"], b["T64"]) and all(a[k] == b[k] for k in keys), (mode, window)
            assert out[0][2]["n_corr"] == 0
            if window is None and iters > 0:
                src = scans[1][:4321]
                o = orc.icp_o3d_p2p(src, m, inits[1], 0.5, iters) if mode == "o3d_p2p" else orc.icp_p2plane(src, m, normals, inits[1], 0.5, iters)
                dt, dr = synth.pose_error(out[0][3]["T64"], o["T"])
                assert out[0][3]["iterations"] == o["iterations"] and dt < 1e-9 and dr < 1e-9
    mp.window_none()


def test_single_launch_fuzz_against_launch_list(api, ctx, synth, small_world):
    """Seeded fuzz of the single-launch alignments (all three modes) against their launch lists: random scan sizes
    (1 .. 40 k points, batches of 1-3), iteration counts, thresholds, windows, start poses near and far -- every result
    must be bit-identical.  Exercises the grid barriers and the cross-workgroup visibility of the slab rows under varying
    grid sizes; SF_FUZZ_TRIALS / SF_FUZZ_SEED override the defaults."""
    import os
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    rng = np.random.default_rng(int(os.environ.get("SF_FUZZ_SEED", "2024")))
    keys = ("iterations", "converged", "n_corr", "n_research", "flags", "error", "rmse", "fitness")
    trials = int(os.environ.get("SF_FUZZ_TRIALS", "40"))
    pool = np.concatenate([synth.make_scan(m, 30000, scan_id=500 + k)[0] for k in range(2)])
    for t in range(trials):
        mode = ("ref_cpp", "o3d_p2p", "p2plane")[t % 3]
        B = int(rng.integers(1, 4))
        n = int(rng.choice([1, 9, 63, 64, 65, 255, 256, 257, 1000, 4097, int(rng.integers(2, 40000 // B))]))
        scans = np.stack([pool[rng.choice(len(pool), n, replace=False)] for _ in range(B)])
        if rng.random() < 0.3:
            scans[rng.integers(0, B)] += np.float32(rng.choice([0.3, 3.0, 300.0]))
        inits = np.stack([synth.make_T(rng.normal(0, 0.03, 3), rng.normal(0, 0.3, 3)) for _ in range(B)])
        kind = rng.integers(0, 3)
        if kind == 1:
            mp.window_sphere(rng.normal(0, 0.5, 3).astype(np.float32), float(rng.uniform(1.0, 6.0)))
        elif kind == 2:
            mp.window_obb(rng.normal(0, 0.5, 3), synth.make_T((0, 0, 0), rng.normal(0, 20, 3))[:3, :3], rng.uniform(1.0, 5.0, 3))
        else:
            mp.window_none()
        prm = (float(rng.choice([0.1, 0.5, 5.0])), int(rng.integers(0, 25)), float(rng.choice([0.001, 0.05, 0.4])), float(rng.choice([1e-5, 5e-3])))
        out = []
        for fused in (True, False):
            icp = api.Icp(ctx, *prm)
            icp.set_fused(fused)
            icp.set_target(mp)
            icp.set_source_batch(scans)
            icp.set_initial_batch(inits)
            out.append(icp.align_batch(mode))
            assert icp.fused_count() == int(fused)
        for a, b in zip(*out):
            assert np.array_equal(a["T64"], b["T64"]) and all(a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]) for k in keys), (t, mode, B, n, prm, kind)
    mp.window_none()


def test_single_launch_under_concurrent_load(api, ctx, synth, small_world):
    """The grid barriers' release / acquire hand-off is easiest to get wrong where an idle chip hides it: the
    single-launch alignments are repeated while a second stream keeps the device busy with large batched alignments
    (uneven load, caches warm with other data) -- every repetition must reproduce the launch list's result bit for bit."""
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    busy_ctx = api.Context(0)
    busy_map = api.Map(busy_ctx, api.Cloud(busy_ctx, m), 0.25)
    busy_map.estimate_normals(0.25)
    big = np.stack([synth.make_scan(m, 60000, scan_id=900 + k)[0] for k in range(8)])
    busy = api.Icp(busy_ctx, 0.5, 20, 0.05, 1e-5)
    busy.set_fused(False)
    busy.set_target(busy_map)
    busy.set_source_batch(big)
    busy.set_initial_batch(None)
    keys = ("iterations", "converged", "n_corr", "n_research", "flags")
    cases = []
    for k, (mode, n) in enumerate([("ref_cpp", 9000), ("o3d_p2p", 5000), ("p2plane", 13000), ("ref_cpp", 700)]):
        scan = synth.make_scan(m, n, scan_id=950 + k)[0]
        init = synth.make_T((0.03, -0.02, 0.01), (0, 0.1, 0.4)).astype(np.float32)
        ref = api.Icp(ctx, 0.5, 12, 0.001, 1e-5)
        ref.set_fused(False)
        ref.set_target(mp)
        ref.set_source(scan)
        ref.set_initial_transformation(init)
        one = api.Icp(ctx, 0.5, 12, 0.001, 1e-5)
        one.set_target(mp)
        one.set_source(scan)
        one.set_initial_transformation(init)
        cases.append((mode, one, ref.align(mode)))
    reps = 120
    for rep in range(reps):
        for _ in range(3):
            busy.align_batch_async("p2plane")                            # ~1 ms of device work each, queued ahead on the other stream
        for mode, one, expect in cases:
            r = one.align(mode)
            assert np.array_equal(r["T64"], expect["T64"]) and all(r[k] == expect[k] for k in keys), (rep, mode)
    busy_ctx.synchronize()
    assert all(one.fused_count() == reps for _, one, _ in cases)


def test_large_sparse_extents_index_at_a_fine_cell(api, ctx, orc, synth):
    """The dense cell table may take a quarter of the device's free memory: an explicit 0.25 m cell over a
    1 km x 1 km x 100 m extent (6.4e9 cells, 64-bit cell ids, 25.6 GB of the 288 GB) and over 500 m x 500 m x 50 m
    (8e8 cells: 32-bit ids, bounds by scan) index and answer exactly like the oracle kd-tree; a cell no device could
    hold is still refused."""
    rng = np.random.default_rng(12)
    for extent, expect_wide in (((500.0, 500.0, 50.0), False), ((1000.0, 1000.0, 100.0), True)):
        ext = np.array(extent)
        corners = np.array([[0, 0, 0], [1, 1, 1], [1, 0, 1], [0.5, 0.5, 0.5], [0.03, 0.97, 0.2]]) * (ext - 20.0)
        pts = np.concatenate([c + synth.make_map(60_000, seed=40 + k)[:40_000] % 20.0 for k, c in enumerate(corners)]).astype(np.float32)
        pts = np.concatenate([pts, np.array([[0, 0, 0], ext], np.float32)])       # pin the bounding box
        mp = api.Map(ctx, api.Cloud(ctx, pts), 0.25)
        cell, dims = mp.cell_size()
        ncell = int(dims[0]) * int(dims[1]) * int(dims[2])
        assert abs(cell - 0.25) < 1e-7 and (ncell >= 2**32) == expect_wide and ncell > 2**28
        q = np.concatenate([pts[rng.choice(len(pts), 4000)] + rng.normal(0, 0.05, (4000, 3)).astype(np.float32),       # near the clusters
                            (rng.uniform(0, 1, (300, 3)) * ext).astype(np.float32)])                                   # in the void: many rings
        oi, od = orc.KdTreeF(pts).nn(q)
        gi, gd = mp.nn(q, 4.0)
        hit = od < 4.0
        assert np.array_equal(gi >= 0, hit) and np.array_equal(gd[hit], od[hit])
        assert (gi[hit] != oi[hit]).mean() < 1e-3
        scan = (pts[rng.choice(200_000, 3000, replace=False)].astype(np.float64) + rng.normal(0, 0.01, (3000, 3)))
        T = synth.make_T((0.05, -0.03, 0.02), (0, 0, 0.5))
        src = ((scan - T[:3, 3]) @ T[:3, :3]).astype(np.float32)
        icp = api.Icp(ctx, 0.5, 30, 0.05, 1e-5)
        icp.set_target(mp)
        icp.set_source(src)
        r = icp.align("o3d_p2p")
        o = orc.icp_o3d_p2p(src, pts, np.eye(4), 0.5, 30)
        assert r["iterations"] == o["iterations"] and r["n_corr"] == o["n_corr"]
        dt, dr = synth.pose_error(r["T64"], o["T"])
        assert dt < 1e-9 and dr < 1e-9
        del mp, icp
    with pytest.raises(api.SlamFusionError):
        api.Map(ctx, api.Cloud(ctx, pts), 0.01)                                   # 1e15 cells


def test_gpu_against_golden_extensions(api, ctx):
    from conftest import load_golden
    g, e = load_golden("registration_small.npz"), load_golden("extensions_small.npz")
    c = api.Cloud(ctx, e["far"])
    assert c.voxel_downsample(0.1, "pcl") == api.SF_FLAG_VOXEL_OVERFLOW
    assert c.voxel_downsample(0.1, "pcl64") == 0
    assert np.array_equal(c.download(), e["far_ds"]) and np.array_equal(c.voxel_point_ids64(), e["far_point_ids"]) and np.array_equal(c.voxel_out_ids64(), e["far_out_ids"])
    mp = api.Map(ctx, api.Cloud(ctx, g["map"]), 0.25)
    mp.estimate_normals(0.3, covariance=True)
    assert np.abs(mp.download_covariances() - e["cov6"]).max() <= 1e-12 * np.abs(e["cov6"]).max()
    bf = api.BruteForceAlignment(ctx)
    bf.setXYZStep(0.1, 0.1, 0.05)
    bf.setXYZRange(0.3, 0.3, 0.1)
    bf.setRotationStep(np.pi / 18.0)
    bf.setRotationRange(np.pi / 6.0)
    bf.setMeanErrorThreshold(1e-9)
    bf.setInitialGuess(e["bf_prev"])
    bf.setSourceCloud(e["bf_scan"])
    bf.setTargetCloud(mp)
    assert not bf.alignClouds()
    r = bf.last_result()
    assert np.array_equal(r["scores"], e["bf_scores"]) and [r["index"], r["n_candidates"]] == list(e["bf_index"])   # serial float32 sums, bit for bit
    assert np.array_equal(bf.getBestTransformation(), e["bf_best_T"])
