"""BASELINE.json full sizes on the GPU (200k-pt scans vs 2M / 10M-pt maps).  The oracle
needs minutes at these sizes, so these tests use size-independent properties of the domain:
voxel ids strictly ascending and consistent with their centroids, NN idempotence on map
points, NN vs brute force on a sample, registration recovering the generating transform,
batched == single, and a bounded oracle spot check."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world_2m(api, ctx, synth):
    raw = synth.make_map(2_000_000)
    c = api.Cloud(ctx, raw)
    c.voxel_downsample(0.1, "pcl")
    ids = c.voxel_out_ids()
    pid = c.voxel_point_ids()
    ds = c.download()
    mp = api.Map(ctx, c, 0.25)
    mp.estimate_normals(0.25)
    return dict(raw=raw, map=ds, ids=ids, pid=pid, mp=mp)


def test_voxel_properties_2m(world_2m):
    raw, ds, ids, pid = world_2m["raw"], world_2m["map"], world_2m["ids"], world_2m["pid"]
    assert (np.diff(ids.astype(np.int64)) > 0).all()                    # ascending, unique
    assert np.array_equal(np.unique(pid), ids)                          # one output per occupied voxel
    assert 0.94 < len(ds) / len(raw) < 0.96                             # ~95 % survive at 100 pts/m^3
    # PCL's index formula recomputed on the host in float32 (closed form, no oracle)
    inv = np.float32(1.0) / np.float32(0.1)
    ijk = np.floor(raw * inv)
    mn = np.floor(raw.min(0) * inv)
    d = (np.floor(raw.max(0) * inv) - mn + 1).astype(np.int64)
    lin = ((ijk - mn).astype(np.int64) * [1, d[0], d[0] * d[1]]).sum(1)
    assert np.array_equal(lin, pid.astype(np.int64))
    # every centroid lies in its own voxel (up to float32 rounding at the faces)
    cijk = np.floor(ds * inv) - mn
    clin = (cijk.astype(np.int64) * [1, d[0], d[0] * d[1]]).sum(1)
    assert (clin == ids).mean() > 0.9999
    # float64 mean per voxel agrees with the float32 sequential centroid
    sums = np.zeros((len(ids), 3))
    pos = np.searchsorted(ids, pid)
    np.add.at(sums, pos, raw.astype(np.float64))
    cnt = np.bincount(pos, minlength=len(ids))
    assert np.abs(sums / cnt[:, None] - ds).max() < 1e-4


def test_nn_properties_2m(api, ctx, orc, synth, world_2m):
    ds, mp = world_2m["map"], world_2m["mp"]
    rng = np.random.default_rng(0)
    pick = rng.choice(len(ds), 200_000, replace=False)
    idx, d2 = mp.nn(ds[pick])
    assert (d2 == 0).all()                                              # idempotence: a map point is its own NN
    assert (idx == pick).mean() > 0.9999
    scan, src = synth.make_scan(ds, 200_000)
    q = (scan.astype(np.float64) @ synth.t_true()[:3, :3].T + synth.t_true()[:3, 3]).astype(np.float32)
    gi, gd = mp.nn(q)
    assert (gi == src).mean() > 0.97                                    # 1 cm noise: almost always the generating point
    sub = rng.choice(len(q), 3000, replace=False)
    oi, od = orc.KdTreeF(ds).nn(q[sub])                                 # bounded oracle spot check
    assert np.array_equal(gd[sub], od) and (gi[sub] == oi).mean() > 0.999


def test_registration_200k_vs_2m(api, ctx, synth, world_2m):
    ds, mp = world_2m["map"], world_2m["mp"]
    scans = np.stack([synth.make_scan(ds, 200_000, scan_id=k)[0] for k in range(2)])
    icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source_batch(scans)
    icp.set_initial_batch(None)
    res = icp.align_batch("p2plane")
    for r in res:
        dt, dr = synth.pose_error(r["T64"], synth.t_true())
        assert r["iterations"] == 20 and r["fitness"] > 0.999
        assert dt < 2e-4 and dr < 1e-5                                  # 200k points: noise floor ~ sigma / sqrt(N)
    icp.set_source(scans[1])
    single = icp.align("p2plane")                                       # AUTO: the batch was cell-ordered, one scan is not
    assert np.abs(single["T64"] - res[1]["T64"]).max() < 1e-12 and single["n_corr"] == res[1]["n_corr"]
    for order in ("cell", "as_given"):                                  # same order: batched == single, bitwise
        icp.set_query_order(order)
        icp.set_source_batch(scans)
        both = icp.align_batch("p2plane")
        icp.set_source(scans[1])
        assert np.array_equal(icp.align("p2plane")["T64"], both[1]["T64"])
    icp.set_query_order("auto")
    icp.set_num_iterations(30)
    r = icp.align("o3d_p2p")
    dt, dr = synth.pose_error(r["T64"], synth.t_true())
    assert r["converged"] and dt < 2e-4 and dr < 1e-5
    icp.set_num_iterations(10)
    icp.set_acceptable_mean_error(0.001)
    r = icp.align("ref_cpp")
    dt, dr = synth.pose_error(r["T64"], synth.t_true())
    assert r["iterations"] == 10 and dt < 2e-3 and dr < 1e-4


def test_registration_200k_vs_10m(api, ctx, synth):
    raw = synth.make_map(10_000_000)
    c = api.Cloud(ctx, raw)
    del raw
    c.voxel_downsample(0.1, "pcl")
    ds = c.download()
    assert 9.4e6 < len(ds) < 9.6e6
    mp = api.Map(ctx, c, 0.25)
    mp.estimate_normals(0.25)
    scan, _ = synth.make_scan(ds, 200_000)
    icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source(scan)
    r = icp.align("p2plane")
    dt, dr = synth.pose_error(r["T64"], synth.t_true())
    assert r["iterations"] == 20 and r["fitness"] > 0.999 and dt < 2e-4 and dr < 1e-5


def test_registration_130k_vs_50m(api, ctx, synth):
    """Config-5-sized case on one GPU: a 50 M-point map (voxel grid -> 47.6 M points, 223 x 223 x 10 m,
    ~2 x 10^8 grid cells), normals for every map point, a 130 k-point scan taken from a 60 m
    neighbourhood (a spinning LiDAR's footprint).  PCL's int32 voxel index does not overflow here
    (2236 x 2236 x 100 < 2^31); the grid-index cell ids are 32-bit as well."""
    raw = synth.make_map(50_000_000)
    c = api.Cloud(ctx, raw)
    del raw
    flags = c.voxel_downsample(0.1, "pcl")
    assert flags == 0
    ids = c.voxel_out_ids()
    assert (np.diff(ids.astype(np.int64)) > 0).all()
    ds = c.download()
    assert 47.0e6 < len(ds) < 48.0e6
    mp = api.Map(ctx, c, 0.25)
    mp.estimate_normals(0.25)
    near = ds[(np.abs(ds[:, 0] - 40.0) < 30.0) & (np.abs(ds[:, 1] + 25.0) < 30.0)]
    T = synth.make_T((0.12, 0.06, -0.03), (0.05, -0.02, 0.2))
    scan, _ = synth.make_scan(near, 130_000, scan_id=4, T=T)
    icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source(scan)
    r = icp.align("p2plane")
    dt, dr = synth.pose_error(r["T64"], T)
    assert r["iterations"] == 20 and r["fitness"] > 0.999 and dt < 3e-4 and dr < 1e-5
    # window = the reference's 10 m crop: points outside it must not be matched
    mp.window_sphere([40.0, -25.0, 0.0], 10.0)
    assert 0 < mp.window_count() < 0.01 * len(ds)
    r = icp.align("p2plane")
    inside = ((scan.astype(np.float64) @ T[:3, :3].T + T[:3, 3] - [40.0, -25.0, 0.0]) ** 2).sum(1) < 9.5 ** 2
    assert inside.sum() * 0.9 < r["n_corr"] < len(scan) * 0.25
