"""sf_node_* (csrc/sf_node.cpp): LocalizationNode::localizationCallback (localization_node.cpp:263-344) as one
library call.  The Python mirror of the same orchestration (localization_flow.LocalizationFlow) is checked against the
oracle flow in tests/test_adapters.py; here the library's own orchestration must give the same poses BIT FOR BIT -- over a
stream with re-crops, from a PointCloud2 message, through the gates, and through the start-up lock (brute force, then the
"strong" ICP)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from test_adapters import make_sensor_scan

pytestmark = pytest.mark.gpu


def messages_for(truth, k, rng, gps_cov, odom_cov):
    q = Rotation.from_matrix(truth[:3, :3]).as_quat()
    odom = dict(q_wxyz=[q[3], q[0], q[1], q[2]], t=truth[:3, 3] + rng.normal(0, 0.002, 3), covariance=odom_cov)
    gps = dict(latitude=-22.9068 + 1e-7 * k, longitude=-43.1729, altitude=12.0, position_covariance=gps_cov)
    return gps, odom


def test_native_node_equals_python_flow_on_a_stream(api, ctx, orc, synth):
    from slam_sensor_fusion_amd.localization_flow import LocalizationFlow, NativeLocalizationFlow
    from slam_sensor_fusion_amd.localization_python import messages
    raw = synth.make_map(400_000, seed=31)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    lla0 = np.array([[-22.9068, -43.1729, 12.0], [-22.90681, -43.17291, 12.1], [-22.90679, -43.17289, 11.9]])
    mtg = orc.map_T_global(lla0, np.zeros(3, np.float32))
    flows = [LocalizationFlow(ctx, ds, mtg, altitude_table=lla0), NativeLocalizationFlow(ctx, ds, mtg, altitude_table=lla0),
             NativeLocalizationFlow(ctx, ds, mtg, altitude_table=lla0)]
    for f in flows:
        f.coarse_alignment_complete_ = True
    gps_cov, odom_cov = np.diag([0.25, 0.25, 0.25]).ravel(), np.diag([1e-4] * 6).ravel()
    rng = np.random.default_rng(5)
    recrops = 0
    for k in range(14):
        truth = synth.make_T((0.45 * k - 3.0, 0.05 * k, 0.0), (0, 0, 1.0 * k))
        scan = make_sensor_scan(synth, ds, truth, 8000, 100 + k)
        gps, odom = messages_for(truth, k, rng, gps_cov, odom_cov)
        if k == 5:                                                       # a dropped message (:269-276) changes nothing
            bad = dict(gps, altitude=-1.0)
            before = [f.map_T_sensor_.copy() for f in flows]
            assert all(f.localizationCallback(scan, bad, odom) is None for f in flows)
            assert flows[1].out_.status == api.SF_NODE_GATED_ALTITUDE
            assert all(np.array_equal(f.map_T_sensor_, b) for f, b in zip(flows, before))
        outs = []
        for i, f in enumerate(flows):
            f.compassCallback(90.0 - k)
            outs.append(f.localizationCallback(messages.PointCloud2(scan) if i == 2 else scan, gps, odom))
        if k == 0:
            assert all(o is None for o in outs) and flows[1].out_.status == api.SF_NODE_FIRST_MESSAGE
            assert np.array_equal(flows[0].map_T_sensor_, flows[1].map_T_sensor_) and np.array_equal(flows[0].odom_T_sensor_previous_, flows[1].odom_T_sensor_previous_)
            for f in flows:                                              # the float32 UTM pose is metre-level: start from the truth
                f.map_T_sensor_ = truth.astype(np.float32)
                f.map_T_ref_ = truth.astype(np.float32)
            continue
        py, nat, pc2 = flows
        for other, o in ((nat, outs[1]), (pc2, outs[2])):
            assert np.array_equal(outs[0], o), k
            assert np.array_equal(py.last["prior"], other.last["prior"]) and py.last["n_scan"] == other.last["n_scan"]
            assert np.array_equal(py.last["odom"], other.last["odom"]) and np.array_equal(py.last["gps"], other.last["gps"]) and py.last["gains"] == other.last["gains"]
            a, b = py.last["icp"], other.last["icp"]
            assert np.array_equal(a["T64"], b["T64"]) and all(a[x] == b[x] for x in ("iterations", "converged", "n_corr", "n_research", "error", "flags"))
            assert np.array_equal(py.map_T_ref_, other.map_T_ref_)
        recrops += nat.out_.recropped
        assert synth.pose_error(outs[1], truth)[0] < 0.1
    assert recrops >= 2                                                  # the 3 m re-crop rule fired
    assert nat.icp_.fused_count() == 13                                   # every per-scan alignment was one launch


@pytest.mark.parametrize("bf_hits", [False, True])
def test_native_node_start_up_lock_equals_python_flow(api, ctx, orc, synth, bf_hits):
    """Both orchestrations from scratch: first message, then the coarse phase over the node's full 7 776-candidate pose
    grid.  On this sparse floor-free map (stride 3 x 15) the node's 0.1 threshold is missed, so the "strong" ICP runs on
    every scan (bf_hits = False); with the threshold raised above the best score the brute force locks at once."""
    from slam_sensor_fusion_amd.localization_flow import LocalizationFlow, NativeLocalizationFlow
    raw = synth.make_map(400_000, seed=41)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    lla0 = np.array([[-22.9068, -43.1729, 12.0]])
    mtg = orc.map_T_global(lla0, np.zeros(1, np.float32))
    py, nat = LocalizationFlow(ctx, ds, mtg, altitude_table=lla0), NativeLocalizationFlow(ctx, ds, mtg, altitude_table=lla0)
    if bf_hits:
        for f in (py, nat):
            f.brute_force_alignment_.setMeanErrorThreshold(0.6)
    gps_cov, odom_cov = np.diag([0.25, 0.25, 0.25]).ravel(), np.diag([1e-4] * 6).ravel()
    rng = np.random.default_rng(9)
    ran = []
    for k in range(5):
        truth = synth.make_T((0.27 + 0.1 * k, -0.14, 0.03), (0, 0, 6.0))
        scan = make_sensor_scan(synth, ds, truth, 6000, 77 + k)
        gps, odom = messages_for(truth, k, rng, gps_cov, odom_cov)
        a, b = py.localizationCallback(scan, gps, odom), nat.localizationCallback(scan, gps, odom)
        if k == 0:
            assert a is None and b is None
            start = np.eye(4, dtype=np.float32)                          # a start pose 0.3 m / 6 degrees off, inside the pose grid
            for f in (py, nat):
                f.map_T_sensor_ = start
                f.map_T_ref_ = start
            continue
        assert (a is None) == (b is None), k
        assert py.coarse_alignment_complete_ == nat.coarse_alignment_complete_
        assert np.array_equal(py.map_T_sensor_, nat.map_T_sensor_), k
        ran.append(int(nat.out_.coarse_ran))
        if nat.out_.coarse_ran == 2:
            r, o = py.last_coarse["icp"], nat.out_.coarse_icp.as_dict()
            assert np.array_equal(r["T64"], o["T64"]) and r["iterations"] == o["iterations"] and r["converged"] == o["converged"]
        if a is not None:
            assert np.array_equal(a, b)
    if bf_hits:
        # locked by the brute force on the first scan (the first candidate under the loose threshold, not the best one:
        # brute_force_alignment.cpp:107-117 returns at once), fine alignments after
        assert ran == [1, 0, 0, 0] and nat.coarse_alignment_complete_ and a is not None
    else:
        assert ran[0] == 2                                               # brute force missed, the "strong" ICP ran


def test_source_scan_one_pass_equals_separate_operations(api, ctx, orc, synth, small_world):
    """sf_icp_set_source_scan = sf_cloud_subsample + sf_cloud_crop_radius + sf_icp_set_source_cloud (a1, a2, a7) in one
    pass with the count left on the device: same points in the same order, hence the same alignment bit for bit -- as one
    launch and as the launch list; the count comes back with the result; non-finite points, clouds shorter than the
    stride and empty crops behave like the separate operations."""
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    rng = np.random.default_rng(3)
    keys = ("iterations", "converged", "n_corr", "n_research", "flags", "error")
    for case, (n_raw, stride, radius, center) in enumerate([(20001, 2, 3.0, (0.0, 0.0, 0.0)), (9000, 3, 2.2, (0.4, -0.3, 0.1)), (700, 1, 50.0, (0, 0, 0)), (5, 7, 50.0, (0, 0, 0)),
                                                             (4000, 2, 0.01, (90.0, 0, 0))]):
        raw = synth.make_scan(m, n_raw, scan_id=200 + case)[0]
        if case == 0:
            raw[rng.choice(n_raw, 50, replace=False), rng.integers(0, 3, 50)] = np.nan
            raw[17] = np.inf
        ref = api.Cloud(ctx, raw).subsample(stride).crop_radius(center, radius)
        sub = orc.uniform_subsample(raw, stride)
        o_pts = sub[np.sort(orc.crop_radius(sub, center, radius)[1])]   # the oracle returns PCL's distance order; index order here
        assert np.array_equal(ref.download(), o_pts)
        init = synth.make_T((0.02, -0.01, 0.0), (0, 0.01, 0.2)).astype(np.float32)
        res = []
        for one_pass in (False, True):
            for fused in (True, False):
                icp = api.Icp(ctx, 0.5, 10, 0.001, 1e-5)
                icp.set_fused(fused)
                icp.set_target(mp)
                cloud = api.Cloud(ctx, raw)
                if one_pass:
                    icp.set_source_scan(cloud, stride, center, radius)
                    assert len(cloud) == n_raw and np.array_equal(cloud.download(), raw, equal_nan=True)   # the raw cloud is untouched
                else:
                    icp.set_source(ref)
                icp.set_initial_transformation(init)
                r = icp.align("ref_cpp")
                assert icp.source_count() == len(o_pts)
                res.append(r)
                if one_pass and len(o_pts) >= 10:
                    with pytest.raises(api.SlamFusionError):
                        icp.align("o3d_p2p")                             # the count is not on the host: REF_CPP only
                    icp.set_source(ref)                                  # an ordinary source afterwards works as before
                    assert np.array_equal(icp.align("o3d_p2p")["T64"], api_o3d(api, ctx, mp, ref, init)["T64"])
        for r in res[1:]:
            assert np.array_equal(r["T64"], res[0]["T64"]) and all(r[k] == res[0][k] for k in keys), case
        if len(o_pts) >= 10:
            o = orc.icp_ref_cpp(o_pts, m, init, 0.5, 10, 0.001, 1e-5, precise=True)
            assert res[0]["iterations"] == o["iterations"] and res[0]["n_corr"] == o["n_corr"]
        else:
            assert res[0]["flags"] & api.SF_ICP_FLAG_FEW_CORR and res[0]["iterations"] == 0


def test_cpp_node_class_equals_native_flow(tmp_path, api, ctx, orc, synth):
    """include/localization/localization_node.h (LocalizationCore over sf_node_*) compiled with g++ against the C ABI
    and driven from files: the poses it prints equal the ones the Python wrapper of the same entry points gets."""
    import os
    import struct
    import subprocess
    from conftest import ROOT
    from slam_sensor_fusion_amd.localization_flow import NativeLocalizationFlow
    exe = str(tmp_path / "test_node_class")
    libdir = os.path.join(ROOT, "slam_sensor_fusion_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_node_class.cpp"), "-o", exe,
                           "-L" + libdir, "-lslamfusion", "-Wl,-rpath," + libdir])
    raw = synth.make_map(400_000, seed=31)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    ds.tofile(tmp_path / "map.bin")
    lla0 = np.array([[-22.9068, -43.1729, 12.0]])
    mtg = api.map_T_global(lla0, np.zeros(1, np.float32))
    flow = NativeLocalizationFlow(ctx, ds, mtg, altitude_table=lla0)
    flow.coarse_alignment_complete_ = True
    gps_cov, odom_cov = np.diag([0.25, 0.25, 0.25]).ravel(), np.diag([1e-4] * 6).ravel()
    rng = np.random.default_rng(5)
    expect, n_msg = [], 8
    with open(tmp_path / "messages.bin", "wb") as f:
        prev_truth = None
        for k in range(n_msg):
            truth = synth.make_T((0.45 * k - 1.5, 0.05 * k, 0.0), (0, 0, 1.0 * k))
            scan = make_sensor_scan(synth, ds, truth, 8000, 300 + k)
            gps, odom = messages_for(truth, k, rng, gps_cov, odom_cov)
            start = np.zeros(16, np.float32)
            if k == 1:
                start = prev_truth.astype(np.float32).ravel()
                flow.map_T_sensor_ = prev_truth.astype(np.float32)
                flow.map_T_ref_ = prev_truth.astype(np.float32)
            f.write(struct.pack("<qd", len(scan), 90.0 - k))
            f.write(np.concatenate([[gps["latitude"], gps["longitude"], gps["altitude"]], gps["position_covariance"]]).astype(np.float64).tobytes())
            f.write(np.concatenate([odom["q_wxyz"], odom["t"], odom["covariance"]]).astype(np.float64).tobytes())
            f.write(start.tobytes())
            f.write(scan.tobytes())
            flow.compassCallback(90.0 - k)
            out = flow.localizationCallback(scan, gps, odom)
            expect.append((out is not None, flow.out_.icp.iterations if out is not None else 0, int(flow.out_.n_scan), np.array(flow.out_.map_T_sensor, np.float32)))
            prev_truth = truth
    lines = subprocess.check_output([exe, str(tmp_path / "map.bin"), str(tmp_path / "messages.bin"), str(n_msg)]).decode().strip().splitlines()
    assert len(lines) == n_msg
    for (ok, iters, n_scan, T), line in zip(expect, lines):
        v = line.split()
        assert int(v[0]) == int(ok) and int(v[1]) == iters and int(v[2]) == n_scan
        assert np.array_equal(np.array([float(x) for x in v[3:19]], np.float32), T)
    assert sum(e[0] for e in expect) == n_msg - 1


def test_native_node_under_concurrent_load(api, ctx, orc, synth):
    """The node's whole fast path (pinned staging, one-pass source, single-launch alignment, result in pinned host
    memory) while a second stream keeps the device busy: every pose equals the one the Python mirror computes with the
    launch list on an otherwise idle device."""
    from slam_sensor_fusion_amd.localization_flow import LocalizationFlow, NativeLocalizationFlow
    raw = synth.make_map(400_000, seed=31)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    lla0 = np.array([[-22.9068, -43.1729, 12.0]])
    mtg = orc.map_T_global(lla0, np.zeros(1, np.float32))
    ref, nat = LocalizationFlow(ctx, ds, mtg, altitude_table=lla0), NativeLocalizationFlow(ctx, ds, mtg, altitude_table=lla0)
    ref.icp_.set_fused(False)
    ref.icp_.use_graph(False)
    for f in (ref, nat):
        f.coarse_alignment_complete_ = True
    gps_cov, odom_cov = np.diag([0.25, 0.25, 0.25]).ravel(), np.diag([1e-4] * 6).ravel()
    rng = np.random.default_rng(5)
    msgs = []
    for k in range(30):
        truth = synth.make_T((0.15 * k - 2.0, 0.02 * k, 0.0), (0, 0, 0.5 * k))
        msgs.append((truth, make_sensor_scan(synth, ds, truth, 9000, 700 + k)) + messages_for(truth, k, rng, gps_cov, odom_cov))
    expect = []
    for k, (truth, scan, gps, odom) in enumerate(msgs):                  # the reference run, nothing else on the device
        ref.compassCallback(90.0 - k)
        expect.append(ref.localizationCallback(scan, gps, odom))
        if k == 0:
            ref.map_T_sensor_ = truth.astype(np.float32)
            ref.map_T_ref_ = truth.astype(np.float32)
    busy_ctx = api.Context(0)
    busy_map = api.Map(busy_ctx, api.Cloud(busy_ctx, ds), 0.25)
    busy_map.estimate_normals(0.25)
    busy = api.Icp(busy_ctx, 0.5, 20, 0.05, 1e-5)
    busy.set_fused(False)
    busy.set_target(busy_map)
    busy.set_source_batch(np.stack([synth.make_scan(ds, 60000, scan_id=800 + k)[0] for k in range(8)]))
    busy.set_initial_batch(None)
    for k, (truth, scan, gps, odom) in enumerate(msgs):
        for _ in range(3):
            busy.align_batch_async("p2plane")
        nat.compassCallback(90.0 - k)
        out = nat.localizationCallback(scan, gps, odom)
        if k == 0:
            assert out is None and expect[0] is None
            nat.map_T_sensor_ = truth.astype(np.float32)
            nat.map_T_ref_ = truth.astype(np.float32)
            continue
        assert np.array_equal(out, expect[k]), k
    busy_ctx.synchronize()
    assert nat.icp_.fused_count() == len(msgs) - 1 and ref.icp_.fused_count() == 0


def api_o3d(api, ctx, mp, cloud, init):
    icp = api.Icp(ctx, 0.5, 10, 0.001, 1e-5)
    icp.set_target(mp)
    icp.set_source(cloud)
    icp.set_initial_transformation(init)
    return icp.align("o3d_p2p")


def test_two_contexts_both_on_the_single_launch_path(api, ctx, orc, synth):
    """ADVICE r2 / VERDICT r2 item 6: two contexts (two streams, two host threads) run single-launch alignments at the same
    time.  Their grids are admitted by the library's per-device ledger only while they fit on the device together, so the
    grid barriers cannot starve each other: every result equals the sequential one bit for bit, nothing is redone, nothing
    times out.  A grid that would not fit beside the ones in flight takes the launch list (same bits)."""
    import threading
    raw = synth.make_map(200_000, seed=77)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    scans = [synth.make_scan(ds, 13_000, scan_id=900 + k)[0] for k in range(2)]

    def make(context, scan, iters=10):
        mp = api.Map(context, api.Cloud(context, ds), 0.5)
        mp.estimate_normals(0.25)
        icp = api.Icp(context, 0.5, iters, 0.001, 1e-5)
        icp.set_target(mp)
        icp.set_source(scan)
        return icp
    ctx2 = api.Context(0)
    a, b = make(ctx, scans[0]), make(ctx2, scans[1])
    want = [a.align("ref_cpp"), b.align("ref_cpp")]                       # one after the other
    assert a.fused_count() == 1 and b.fused_count() == 1
    got, errs = [[], []], []

    def run(icp, out):
        try:
            for _ in range(200):
                out.append(icp.align("ref_cpp"))
        except Exception as e:                                            # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=run, args=(a, got[0])), threading.Thread(target=run, args=(b, got[1]))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert len(got[k]) == 200
        for r in got[k]:
            assert np.array_equal(r["T64"], want[k]["T64"]) and r["iterations"] == want[k]["iterations"] and r["flags"] == want[k]["flags"]
    assert a.fused_count() == 201 and b.fused_count() == 201              # both stayed on the single launch ...
    assert a.fused_redone() == 0 and b.fused_redone() == 0                # ... and no barrier ever gave up
    # a grid that does not fit beside one in flight: enqueue a large single-launch alignment on one context and, before
    # fetching it, ask the other context for one that needs more than what is left -> launch list, same result
    big = np.stack([synth.make_scan(ds, 16_000, scan_id=950 + k)[0] for k in range(5)])
    c, d = make(ctx, big[0], 20), make(ctx2, big[0], 20)
    c.set_source_batch(big)
    d.set_source_batch(big)
    ref = c.align_batch("p2plane")
    n_fused = c.fused_count()
    assert n_fused == 1                                                   # 5 x 63 rows of 256 points (of the 512 the device holds) fit alone
    c.align_batch_async("p2plane")                                        # in flight: holds its share of the device
    res_d = d.align_batch("p2plane")                                      # 315 + 315 rows > 512: does not fit beside it
    res_c = c.fetch_results()
    assert c.fused_count() == 2 and d.fused_count() == 0
    for r, w in zip(res_c, ref):
        assert np.array_equal(r["T64"], w["T64"])
    for r, w in zip(res_d, ref):
        assert np.array_equal(r["T64"], w["T64"]) and r["iterations"] == w["iterations"]
    assert d.align_batch("p2plane")[0]["iterations"] == ref[0]["iterations"] and d.fused_count() == 1   # alone again: single launch
    # a barrier that gives up (another process holding the compute units; injected here): the alignment is redone through
    # the launch list inside the fetch -- same result, no error -- and the object stays on the launch list
    a.test_inject_barrier_timeout()
    r = a.align("ref_cpp")
    assert np.array_equal(r["T64"], want[0]["T64"]) and r["iterations"] == want[0]["iterations"] and r["flags"] == want[0]["flags"]
    assert a.fused_redone() == 1 and a.fused_count() == 202
    r = a.align("ref_cpp")
    assert np.array_equal(r["T64"], want[0]["T64"]) and a.fused_count() == 202
    ctx2.synchronize()


def test_start_outside_the_map_recrops_until_the_crop_holds_points(api, ctx, orc, synth):
    """ADVICE r2 (sf_node.cpp:134): the reference re-crops while ref_cropped_map_cloud_ is EMPTY (localization_node.cpp:299),
    e.g. after a first fix more than the 10 m crop radius away from every map point -- every callback, not only after 3 m of
    travel.  Native node and Python mirror agree bit for bit on the way in; flow.last of an earlier scan keeps its values."""
    from slam_sensor_fusion_amd.localization_flow import LocalizationFlow, NativeLocalizationFlow
    raw = synth.make_map(400_000, seed=31)                                  # x, y in [-10, 10]
    ds = orc.voxel_pcl(raw, 0.1)[0]
    lla0 = np.array([[-22.9068, -43.1729, 12.0]])
    mtg = orc.map_T_global(lla0, np.zeros(1, np.float32))
    flows = [LocalizationFlow(ctx, ds, mtg, altitude_table=lla0), NativeLocalizationFlow(ctx, ds, mtg, altitude_table=lla0)]
    for f in flows:
        f.coarse_alignment_complete_ = True
    gps_cov, odom_cov = np.diag([0.25, 0.25, 0.25]).ravel(), np.diag([1e-4] * 6).ravel()
    rng = np.random.default_rng(9)
    recropped, kept = [], None
    for k in range(8):
        truth = synth.make_T((-21.5 + 1.0 * k, 0.0, 0.0), (0, 0, 0))       # starts 11.5 m west of the map's edge, 1 m per scan
        scan = make_sensor_scan(synth, ds, synth.make_T((-8.0, 0.0, 0.0), (0, 0, 0)), 6000, 300 + k)   # (what it sees does not matter out there)
        gps, odom = messages_for(truth, k, rng, gps_cov, odom_cov)
        outs = []
        for f in flows:
            f.compassCallback(90.0)
            outs.append(f.localizationCallback(scan, gps, odom))
        if k == 0:
            for f in flows:
                f.map_T_sensor_ = truth.astype(np.float32)
                f.map_T_ref_ = truth.astype(np.float32)
            continue
        assert np.array_equal(outs[0], outs[1]), k
        recropped.append(int(flows[1].out_.recropped))
        for f in flows:                                                  # hold the pose on the drive (nothing to register against out there)
            f.map_T_sensor_ = truth.astype(np.float32)
        if k == 1:
            kept = (flows[1].last, np.array(flows[1].last["prior"]), np.array(flows[0].last["prior"]))
    # crop centre at x = -20.5, -19.5: empty (map starts at -10, radius 10) -> re-cropped on every callback although < 3 m were
    # travelled; from -18.5 on it holds points and the 3 m rule takes over
    assert recropped[:3] == [1, 1, 1] and sum(recropped[3:]) <= 2, recropped
    assert np.array_equal(kept[0]["prior"], kept[1]) and np.array_equal(kept[1], kept[2])   # flow.last of scan 1, read again six scans later
