"""Pins the oracle's pose-fusion restatement: UTM against oracle/_ref (the reference's own
geo_lib.hpp compiled from /root/reference), quaternions against scipy Rotation (which the
reference's Python twin itself uses), StochasticFilter against its documented weights and
a scripted gate trip, BruteForceAlignment candidate order."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import load_golden


def test_utm_matches_reference_build(orc):
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference absent on this machine); golden test covers it")
    rng = np.random.default_rng(0)
    for lat, lon in np.c_[rng.uniform(-80, 84, 300), rng.uniform(-180, 180, 300)]:
        assert orc.ll_to_utm(lat, lon) == orc.ref_ll_to_utm(lat, lon)          # bit-identical doubles
    for lat, lon in [(60.0, 5.0), (0.0, 0.0), (-22.9068, -43.1729), (45.0, 179.99), (10.0, -180.0)]:
        assert orc.ll_to_utm(lat, lon) == orc.ref_ll_to_utm(lat, lon)


def test_utm_golden_from_reference(orc):
    g = load_golden("fusion.npz")
    for (lat, lon), (n, e) in zip(g["latlon"], g["utm_ref"]):
        assert orc.ll_to_utm(lat, lon) == (n, e)
    # northern-hemisphere points carry the +10 000 000 m offset too (geo_lib.hpp:82)
    n, e = orc.ll_to_utm(48.8566, 2.3522)
    assert 1.5e7 < n < 1.6e7 and 4e5 < e < 5e5
    # the Python twin's utm.from_latlon does not: they differ by exactly the offset (to series accuracy)
    e2, n2 = orc.utm_from_latlon(48.8566, 2.3522)
    assert abs((n - 1e7) - n2) < 0.01 and abs(e - e2) < 0.01
    e3, n3 = orc.utm_from_latlon(-22.9068, -43.1729)
    n4, e4 = orc.ll_to_utm(-22.9068, -43.1729)
    assert abs(n3 - n4) < 0.01 and abs(e3 - e4) < 0.01


def test_quaternion_and_yaw_against_scipy(orc):
    rng = np.random.default_rng(1)
    for _ in range(20):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        T = orc.quat_to_pose(q, [1, 2, 3])
        R = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        assert np.allclose(T[:3, :3], R, atol=2e-6) and np.allclose(T[:3, 3], [1, 2, 3])
    assert abs(orc.compass_to_yaw(90.0)) < 1e-7
    assert abs(orc.compass_to_yaw(0.0) - np.pi / 2) < 1e-6
    assert abs(orc.compass_to_yaw(350.0) - (np.radians(-260) + 2 * np.pi)) < 1e-6   # wrapped into [-pi, pi]
    assert abs(orc.compass_to_yaw(-100.0) - (np.radians(190) - 2 * np.pi)) < 1e-6


def test_inverse_mul_prediction(orc, synth):
    A = synth.make_T((1, 2, 3), (10, 20, 30)).astype(np.float32)
    assert np.allclose(orc.mat4f_inverse(A), np.linalg.inv(A.astype(np.float64)), atol=1e-5)
    B = synth.make_T((-1, 0.5, 0), (0, 5, -40)).astype(np.float32)
    assert np.allclose(orc.mat4f_mul(A, B), A.astype(np.float64) @ B, atol=1e-5)
    mts = synth.make_T((5, 5, 1), (0, 0, 45)).astype(np.float32)
    pred = orc.odom_prediction(mts, A, B)
    assert np.allclose(pred, mts.astype(np.float64) @ np.linalg.inv(A.astype(np.float64)) @ B, atol=2e-5)


def test_gains_and_blend(orc):
    go, gg = orc.pose_gains(np.diag([0.25, 0.25, 0.25]), np.diag([1e-4] * 6))
    assert abs(go - 0.75 / 0.7503) < 1e-6 and abs(gg - 0.0003 / 0.7503) < 1e-6   # odom gain = tr(gps)/sum
    assert orc.pose_gains(np.eye(3), np.eye(6), fixed=True) == (np.float32(0.95), np.float32(0.05))
    a, b = orc.pose_gains(np.zeros((3, 3)), np.zeros((6, 6)))
    assert np.isnan(a) and np.isnan(b)                                             # 0/0, as the reference
    T1, T2 = np.eye(4, dtype=np.float32), 2 * np.eye(4, dtype=np.float32)
    assert np.allclose(orc.blend(0.8, T1, 0.2, T2), 0.8 * T1 + 0.2 * T2)


def test_gps_pose_float32_quantisation(orc):
    g = load_golden("fusion.npz")
    mtg = g["map_T_global"]
    pose = orc.gps_pose(mtg, orc.compass_to_yaw(75.0), -22.90685, -43.17295, 12.2)
    assert np.array_equal(pose, g["gps_pose"])
    # float64 evaluation of the same chain: the float32 path is only metre-accurate at
    # UTM magnitudes (ulp(1e7) = 1 m) — the reference's own behaviour, reproduced not fixed
    n, e = orc.ll_to_utm(-22.90685, -43.17295)
    yaw = orc.compass_to_yaw(75.0)
    G = np.eye(4)
    G[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
    G[:3, 3] = [e, n, 12.2]
    exact = mtg @ G
    assert np.abs(pose[:3, 3] - exact[:3, 3]).max() < 3.0
    assert orc.closest_altitude(np.array([[1.0, 1.0, 5.0], [2.0, 2.0, 9.0]]), 1.9, 2.2) == 9.0
    assert orc.closest_altitude(np.zeros((0, 3)), 1.0, 1.0) == 0.0


def test_stochastic_filter_weights_and_gate(orc, synth):
    f = orc.StochasticFilter(4, 3.0)
    assert np.allclose(f.weights(), [0.03206, 0.08714, 0.23688, 0.64391], atol=1e-5)   # e^(i-Q)/sum
    g = load_golden("fusion.npz")
    prev = np.eye(4, dtype=np.float32)
    zs = []
    w = f.weights().astype(np.float64)
    poses = [np.eye(4)] + [p.astype(np.float64) for p in g["filter_poses"]]
    for k, T in enumerate(g["filter_poses"]):
        f.add_pose(T)
        z = f.zscore(prev, T)
        out = f.apply(prev, T)
        zs.append(z)
        if k < 3:
            assert z == 0.0                              # queue not full yet (cpp:60-63)
        if k == 7:
            assert z > 3.0                               # scripted 1.5 m jump trips the 3-sigma gate
            assert not np.array_equal(out, T)
            # replaced by sum_i w_i * (queue_i * previous) (cpp:103-109).  The queue already holds
            # the outlier transition itself (addPoseToQueue runs first, localization_node.cpp:331-332)
            # with the largest weight, so the "filtered" pose is still pulled 64 % of the way: a
            # reference quirk that is reproduced, not fixed.
            trans = [np.linalg.inv(poses[j]) @ poses[j + 1] for j in range(k - 3, k + 1)]
            expect = sum(wi * (Q @ prev.astype(np.float64)) for wi, Q in zip(w, trans))
            assert np.allclose(out, expect, atol=1e-5)
            assert 0.6 < out[0, 3] < T[0, 3]
        elif z <= 3.0:
            assert np.array_equal(out, T)
        assert np.array_equal(out, g["filter_out"][k])
        prev = T
    assert np.array_equal(np.array(zs, np.float32), g["filter_z"])


def test_brute_force_candidate_order(orc):
    # localization_node.cpp:39-43 -> 18 x 18 x 4 x 6 = 7776 candidates, each axis -0,+0,-s,+s,...
    x = orc.bf_sequence(1.5, 0.1)
    z = orc.bf_sequence(0.1, 0.05)
    yaw = orc.bf_sequence(np.pi / 6.0, np.pi / 18.0)
    assert (len(x), len(z), len(yaw)) == (18, 4, 6)
    assert len(x) * len(x) * len(z) * len(yaw) == 7776
    assert list(x[:6]) == [np.float32(0.0), np.float32(0.0), np.float32(-0.1), np.float32(0.1), np.float32(-0.2), np.float32(0.2)]
    assert not np.signbit(x[0])      # -i*step with int i = 0 is +0.0: the zero offset is simply tried twice
