"""Host-side mirrors of the reference's own interfaces (the rows either side of the kernels):
the C++ `ICPPointToPoint` / point_cloud_processing / StochasticFilter headers, the
`localization_python`-shaped LocalizationNode and the C++ node's per-scan orchestration.
Each GPU test replays the same sequence with oracle functions and compares."""
import os
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import ROOT

CPP_SRC = os.path.join(ROOT, "tests", "cpp", "test_icp_class.cpp")


def build_cpp(tmp_path):
    exe = str(tmp_path / "test_icp_class")
    libdir = os.path.join(ROOT, "slam_sensor_fusion_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), CPP_SRC, "-o", exe,
                           "-L" + libdir, "-lslamfusion", "-Wl,-rpath," + libdir])
    return exe


def test_cpp_mirror_compiles_against_the_c_abi(tmp_path, api):
    api.load_library()
    assert os.path.exists(build_cpp(tmp_path))


def test_python_adapter_imports_without_ros_and_geo_matches_oracle(orc):
    from slam_sensor_fusion_amd.localization_python import geo, messages
    import slam_sensor_fusion_amd.localization_python as lp
    assert hasattr(lp, "LocalizationNode") and hasattr(lp, "main")
    for name in ("readFilterPtcRegionPoints", "buildNavOdomMsg", "computeGpsCoarsePoseInMapFrame",
                 "computeModelPosePredictionFromOdometry", "compassCallback", "syncCallback", "publishMapCallback"):
        assert callable(getattr(lp.LocalizationNode, name))              # localization_node.py:105-269
    rng = np.random.default_rng(0)
    for lat, lon in np.c_[rng.uniform(-79, 83, 200), rng.uniform(-179, 179, 200)]:
        e, n, _, _ = geo.from_latlon(lat, lon)
        oe, on = orc.utm_from_latlon(lat, lon)
        assert abs(e - oe) < 1e-6 and abs(n - on) < 1e-6
    pts = rng.normal(size=(50, 3)).astype(np.float32)
    assert np.array_equal(messages.read_points_xyz(messages.PointCloud2(pts)), pts)


@pytest.mark.gpu
def test_cpp_class_matches_oracle(tmp_path, orc, synth, small_world):
    exe = build_cpp(tmp_path)
    m, scan = small_world["map"], small_world["scan"]
    m.tofile(tmp_path / "map.bin")
    scan.tofile(tmp_path / "scan.bin")
    out = subprocess.check_output([exe, str(tmp_path / "map.bin"), str(tmp_path / "scan.bin"), "0.5", "10", "0.001", "1e-5"]).decode().split()
    n_crop, iters, conv, err = int(out[0]), int(out[1]), int(out[2]), float(out[3])
    T = np.array([float(v) for v in out[4:20]]).reshape(4, 4)
    s = orc.uniform_subsample(scan, 2)                                   # localization_node.cpp:292
    crop, _ = orc.crop_radius(s, [0, 0, 0], 10.0)                        # :296
    o = orc.icp_ref_cpp(crop, m, None, 0.5, 10, 0.001, 1e-5, precise=True)
    assert n_crop == len(crop) and iters == o["iterations"] and conv == int(o["converged"])
    dt, dr = synth.pose_error(T, o["T"])
    assert dt < 1e-4 and dr < 1e-5 and abs(err - o["error"]) < 1e-5


def make_sensor_scan(synth, map_pts, T_sensor, n, seed):
    rng = np.random.default_rng(seed)
    idx = rng.choice(len(map_pts), n, replace=False)
    p = map_pts[idx].astype(np.float64) + rng.normal(0, 0.01, (n, 3))
    Ti = np.linalg.inv(T_sensor)
    return (p @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)


@pytest.mark.gpu
def test_python_node_matches_oracle_flow(api, ctx, orc, synth):
    from slam_sensor_fusion_amd.localization_python import LocalizationNode, geo, messages
    raw = synth.make_map(60_000, seed=21)
    raw[:, 0] += 8.0                                                     # scene in front of the sensor (AABB x in [0, 15])
    raw[:, 2] += 4.0
    node = LocalizationNode(map_points=raw, map_T_global=np.eye(4), ctx=ctx)
    means, _, _, _ = orc.voxel_o3d(raw.astype(np.float64), 0.1)
    ds = means.astype(np.float32)
    assert np.array_equal(node.map_original.download(), ds)
    # before any compass message the callback is gated (:197)
    node.syncCallback(messages.PointCloud2(ds[:10]), messages.Odometry(), messages.NavSatFix())
    assert "/localization/map_T_sensor" not in node.published
    node.compassCallback(messages.Float64(88.0))
    assert abs(node.current_compass - np.radians(2.0)) < 1e-12
    lat, lon = -22.9068, -43.1729
    e0, n0, _, _ = geo.from_latlon(lat, lon)
    mtg = np.eye(4)
    mtg[:3, 3] = [-e0, -n0, 0.0]                                         # map origin at the first fix
    node.map_T_global = mtg
    o_map_T_sensor, o_prev = np.eye(4), np.eye(4)
    for k in range(1, 4):
        truth = synth.make_T((0.05 * k, 0.01 * k, 0.0), (0, 0, 0.2 * k))
        scan = make_sensor_scan(synth, ds, truth, 6000, k)
        q = Rotation.from_matrix(truth[:3, :3]).as_quat()
        odom = messages.Odometry(position=truth[:3, 3], orientation_xyzw=q, stamp=float(k))
        gps = messages.NavSatFix(lat, lon, 1.0)
        node.syncCallback(messages.PointCloud2(scan), odom, gps)
        # ---- the same callback with oracle functions (localization_node.py:193-243)
        o_cur = np.eye(4)
        o_cur[:3, :3] = Rotation.from_quat(q).as_matrix()
        o_cur[:3, 3] = truth[:3, 3]
        o_odom = (o_cur @ np.linalg.inv(o_prev)) @ o_map_T_sensor
        G = np.eye(4)
        G[:3, :3] = Rotation.from_euler("xyz", [0, 0, node.current_compass]).as_matrix()
        oe, on = orc.utm_from_latlon(lat, lon)
        G[:3, 3] = [oe, on, 1.0]
        coarse = 0.2 * (mtg @ G) + 0.8 * o_odom
        cs, _ = orc.crop_aabb(scan, [0, -7.5, 0], [15, 7.5, 7.5])
        cm, _ = orc.crop_obb(ds, coarse[:3, 3], coarse[:3, :3], [30.0, 15.0, 15.0])
        o = orc.icp_o3d_p2p(cs, cm, coarse, 0.5, 30)
        o_map_T_sensor, o_prev = o["T"], o_cur
        r = node.last_icp_result
        assert r["iterations"] == o["iterations"] and r["n_corr"] == o["n_corr"]
        dt, dr = synth.pose_error(node.map_T_sensor, o["T"])
        assert dt < 1e-7 and dr < 1e-8
        dt, dr = synth.pose_error(node.map_T_sensor, truth)
        assert dt < 5e-3 and dr < 5e-4
    msg = node.published["/localization/map_T_sensor"][-1]
    assert msg.header.frame_id == "map" and msg.child_frame_id == "sensor"
    assert node.published["/localization/map_T_sensor_gps"][-1].child_frame_id == "sensor_gps"
    assert len(node.published["/localization/cropped_scan_map_frame"]) == 3
    # empty map crop: warn and leave the state untouched (:226-228)
    before = node.map_T_sensor.copy()
    far = messages.Odometry(position=(500.0, 0, 0), stamp=9.0)
    node.odom_previous_T_sensor = np.eye(4)
    node.map_T_sensor = np.eye(4)
    node.syncCallback(messages.PointCloud2(scan), far, messages.NavSatFix(lat, lon, 1.0))
    assert node.get_logger().lines[-1] == ("warn", "Cropped map has no points, not localizing ...")
    assert np.array_equal(node.map_T_sensor, np.eye(4)) and not np.array_equal(before, np.eye(4))


@pytest.mark.gpu
def test_cpp_node_orchestration_matches_oracle_flow(api, ctx, orc, synth):
    """localization_node.cpp:263-344 over 14 scans: prediction, GPS pose, gains, blend,
    StochasticFilter, window re-crop after 3 m, ref_cpp ICP — GPU flow vs oracle flow."""
    from slam_sensor_fusion_amd.localization_flow import LocalizationFlow
    raw = synth.make_map(400_000, seed=31)                               # 20 x 20 x 10 m
    ds = orc.voxel_pcl(raw, 0.1)[0]
    lla0 = np.array([[-22.9068, -43.1729, 12.0], [-22.90681, -43.17291, 12.1], [-22.90679, -43.17289, 11.9]])
    mtg = orc.map_T_global(lla0, np.zeros(3, np.float32))
    flow = LocalizationFlow(ctx, ds, mtg, altitude_table=lla0)
    flow.coarse_alignment_complete_ = True                               # the lock is given; the coarse phase has its own test
    sub = orc.uniform_subsample(ds, 3)                                   # :20
    assert np.array_equal(flow.map_cloud_.download(), sub)
    ofilter = orc.StochasticFilter(4, 3.0)
    gps_cov, odom_cov = np.diag([0.25, 0.25, 0.25]).ravel(), np.diag([1e-4] * 6).ravel()
    rng = np.random.default_rng(5)
    o_mts = o_prev = o_ref = o_crop = None
    recrops = 0
    for k in range(14):
        truth = synth.make_T((0.45 * k - 3.0, 0.05 * k, 0.0), (0, 0, 1.0 * k))
        scan = make_sensor_scan(synth, ds, truth, 8000, 100 + k)
        q = Rotation.from_matrix(truth[:3, :3]).as_quat()
        odom = dict(q_wxyz=[q[3], q[0], q[1], q[2]], t=truth[:3, 3] + rng.normal(0, 0.002, 3), covariance=odom_cov)
        gps = dict(latitude=-22.9068 + 1e-7 * k, longitude=-43.1729, altitude=12.0, position_covariance=gps_cov)
        flow.compassCallback(90.0 - k)
        yaw = orc.compass_to_yaw(90.0 - k)
        out = flow.localizationCallback(scan, gps, odom)
        o_cur = orc.quat_to_pose(odom["q_wxyz"], odom["t"])
        o_gps = orc.gps_pose(mtg, yaw, gps["latitude"], gps["longitude"], orc.closest_altitude(lla0, gps["latitude"], gps["longitude"]))
        if k == 0:
            assert out is None
            o_mts, o_ref, o_prev = o_gps, o_gps.copy(), o_cur
            # the float32 UTM pose is only metre-accurate: start both flows from the truth instead
            flow.map_T_sensor_ = truth.astype(np.float32)
            flow.map_T_ref_ = truth.astype(np.float32)
            o_mts, o_ref = truth.astype(np.float32), truth.astype(np.float32)
            continue
        s = orc.crop_radius(orc.uniform_subsample(scan, 2), [0, 0, 0], 10.0)[0]
        rel = orc.mat4f_mul(orc.mat4f_inverse(o_mts), o_ref)
        if np.linalg.norm(rel[:3, 3]) > 3.0 or o_crop is None:
            o_crop = orc.crop_radius(sub, o_mts[:3, 3], 10.0)[0]
            o_ref = o_mts.copy()
            recrops += 1
        o_odom = orc.odom_prediction(o_mts, o_prev, o_cur)
        g_o, g_g = orc.pose_gains(gps_cov, odom_cov)
        prior = orc.blend(g_o, o_odom, g_g, o_gps)
        ofilter.add_pose(prior)
        prior = ofilter.apply(o_mts, prior)
        o = orc.icp_ref_cpp(s, o_crop, prior, 0.5, 10, 0.05, 1e-5, precise=True)
        o_mts, o_prev = o["T"].astype(np.float32), o_cur
        assert flow.last["n_scan"] == len(s)
        assert np.allclose(flow.last["prior"], prior, atol=2e-5)
        assert flow.last["icp"]["iterations"] == o["iterations"] and flow.last["icp"]["n_corr"] == o["n_corr"]
        dt, dr = synth.pose_error(out, o["T"])
        assert dt < 1e-4 and dr < 1e-5, (k, dt, dr)
        dt, dr = synth.pose_error(out, truth)
        assert dt < 0.1 and dr < 1e-2                                 # accept = 0.05 m stops ref_cpp early (cpp:215-219)
    assert recrops >= 2                                                  # the 3 m re-crop rule fired


@pytest.mark.gpu
def test_cpp_node_coarse_alignment_matches_oracle(api, ctx, orc, synth):
    """performCoarseAlignment (localization_node.cpp:200-261): map crop in PCL order -> stride 15 ->
    removeFloor on both clouds -> brute force; when that misses, the "strong" ICP
    (max dist 5.0, eps 1e-2, accept 0.4, 80 iterations) from the brute-force best pose."""
    from slam_sensor_fusion_amd.localization_flow import LocalizationFlow
    raw = synth.make_map(400_000, seed=41)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    sub = orc.uniform_subsample(ds, 3)
    truth = synth.make_T((0.27, -0.14, 0.03), (0, 0, 6.0))
    scan = make_sensor_scan(synth, ds, truth, 6000, 77)
    # the sparse, floor-free coarse map (stride 3 * 15) leaves a mean squared NN distance of ~0.1-0.2 m^2
    # even at the right pose: take the hit threshold just above the best score of the exhaustive run
    o_scan0 = orc.remove_floor(orc.crop_radius(orc.uniform_subsample(scan, 2), [0, 0, 0], 10.0)[0])
    o_map0 = orc.remove_floor(orc.uniform_subsample(orc.crop_radius(sub, [0, 0, 0], 10.0)[0], 15))
    exhaustive = orc.bf_align(o_scan0, o_map0, np.eye(4), x_range=0.5, y_range=0.5, threshold=1e-6)
    thr_hit = float(np.float32(exhaustive["scores"].min() * 1.02))
    for thr, expect_bf_hit in ((thr_hit, True), (1e-6, False)):
        flow = LocalizationFlow(ctx, ds, np.eye(4))
        bf = flow.brute_force_alignment_
        bf.setXYZRange(0.5, 0.5, 0.1)                                    # smaller grid than the node's so the oracle stays fast
        bf.setMeanErrorThreshold(thr)
        flow.map_T_sensor_ = np.eye(4, dtype=np.float32)
        flow.map_T_ref_ = np.eye(4, dtype=np.float32)
        s = api.Cloud(ctx, scan).subsample(2).crop_radius([0, 0, 0], 10.0, sorted=True)
        ok = flow.performCoarseAlignment(s)
        # ---- oracle replay
        o_scan = orc.remove_floor(orc.crop_radius(orc.uniform_subsample(scan, 2), [0, 0, 0], 10.0)[0])
        o_map = orc.remove_floor(orc.uniform_subsample(orc.crop_radius(sub, [0, 0, 0], 10.0)[0], 15))
        assert flow.last_coarse["n_scan"] == len(o_scan) and flow.last_coarse["n_map"] == len(o_map)
        o = orc.bf_align(o_scan, o_map, np.eye(4), x_range=0.5, y_range=0.5, threshold=thr)
        assert o["found"] == expect_bf_hit
        if expect_bf_hit:
            assert ok and flow.coarse_alignment_complete_
            assert np.array_equal(flow.map_T_sensor_, o["best_T"])
        else:
            oi = orc.icp_ref_cpp(o_scan, o_map, o["best_T"], 5.0, 80, 0.4, 1e-2, precise=True)
            r = flow.last_coarse["icp"]
            assert r["iterations"] == oi["iterations"] and r["converged"] == oi["converged"]
            dt, dr = synth.pose_error(r["T64"], oi["T"])
            assert dt < 1e-4 and dr < 1e-5
            assert ok == oi["converged"]
            if ok:
                assert bf.firstAlignmentCompleted()                      # resetFirstAlignment(true), :237
