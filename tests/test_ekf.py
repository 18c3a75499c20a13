"""Extension f-4: error-state EKF pose prior with IMU pre-integration — host code of
libslamfusion (sf_ekf_*), compared with this build's own numpy restatement (oracle/ekf_np.py;
PARITY UNPINNED: the reference has no EKF and no IMU consumer) and exercised on a simulated drive.
No GPU needed: the library loads, the EKF is pure host arithmetic."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401


@pytest.fixture(scope="module")
def api_host():
    from slam_sensor_fusion_amd import api
    api.load_library()
    return api


def rot(rpy):
    from slam_sensor_fusion_amd import synth
    return synth.rpy_to_R(*rpy)


def test_ekf_matches_numpy_restatement(api_host):
    from oracle import ekf_np
    rng = np.random.default_rng(3)
    e, o = api_host.Ekf(), ekf_np.Ekf()
    T0 = np.eye(4)
    T0[:3, :3], T0[:3, 3] = rot((0.02, -0.03, 0.7)), (3.0, -2.0, 0.5)
    P0 = rng.uniform(0.01, 0.5, 9)
    for f in (e, o):
        f.reset(T0, [1.0, 0.2, 0.0], P0)
        f.set_noise(2e-3, 5e-2, None)
        f.set_bias([0.001, -0.002, 0.0005], [0.02, -0.01, 0.03], [1e-4] * 3, [1e-2] * 3)
        f.set_bias_noise(1e-4, 1e-3)
    odo_prev = np.eye(4)
    for step in range(40):
        n = int(rng.integers(1, 12))
        gyro = rng.normal(0, 0.2, (n, 3)) + [0, 0, 0.3]
        accel = rng.normal(0, 0.5, (n, 3)) + [0.2, 0.0, 9.80665]
        e.predict_imu(gyro, accel, 0.005)
        o.predict_imu(gyro, accel, 0.005)
        kind = step % 4
        if kind == 0:
            z, cov = o.p + rng.normal(0, 0.3, 3), np.diag(rng.uniform(0.05, 0.4, 3)) + 0.01
            e.update_position(z, cov)
            o.update_position(z, cov)
        elif kind == 1:
            yaw, var = np.arctan2(o.R[1, 0], o.R[0, 0]) + rng.normal(0, 0.05) + (2 * np.pi if step % 8 == 1 else 0.0), 0.01
            e.update_yaw(yaw, var)
            o.update_yaw(yaw, var)
        elif kind == 2:
            Tm = np.eye(4)
            Tm[:3, :3] = o.R @ ekf_np.so3_exp(rng.normal(0, 0.01, 3))
            Tm[:3, 3] = o.p + rng.normal(0, 0.02, 3)
            e.update_pose(Tm, [1e-4] * 3, [1e-5] * 3)
            o.update_pose(Tm, [1e-4] * 3, [1e-5] * 3)
        else:
            odo_cur = odo_prev.copy()
            odo_cur[:3, :3] = odo_prev[:3, :3] @ rot(rng.normal(0, 0.01, 3))
            odo_cur[:3, 3] = odo_prev[:3, 3] + rng.normal(0, 0.05, 3)
            e.predict_odometry(odo_prev, odo_cur, [1e-4] * 3, [1e-6] * 3)
            o.predict_odometry(odo_prev, odo_cur, [1e-4] * 3, [1e-6] * 3)
            odo_prev = odo_cur
        T, v, P = e.state()
        assert np.abs(T - o.pose()).max() < 1e-10 and np.abs(v - o.v).max() < 1e-10
        assert np.abs(P - o.P[:9, :9]).max() < 1e-10 * max(1.0, np.abs(o.P).max())
        bg, ba, P15 = e.full_state()
        assert np.abs(bg - o.bg).max() < 1e-12 and np.abs(ba - o.ba).max() < 1e-12
        assert np.abs(P15 - o.P).max() < 1e-10 * max(1.0, np.abs(o.P).max())
        assert np.allclose(P15, P15.T) and np.linalg.eigvalsh(P15).min() > -1e-12
        assert np.abs(T[:3, :3] @ T[:3, :3].T - np.eye(3)).max() < 1e-9


def test_ekf_tracks_a_simulated_drive(api_host):
    """Circle at 5 m/s, 200 Hz IMU with noise, GPS (0.5 m) + compass (2 deg) at 10 Hz: the filter stays
    within a fraction of the GPS noise; adding ICP poses (1 cm) at 10 Hz brings it to centimetres; IMU
    alone drifts."""
    rng = np.random.default_rng(9)
    dt, rate, w, speed = 0.005, 20, 0.2, 5.0
    ekf, ekf_icp, dead = api_host.Ekf(), api_host.Ekf(), api_host.Ekf()
    T0 = np.eye(4)
    for f in (ekf, ekf_icp, dead):
        f.reset(T0, [speed, 0.0, 0.0], [0.25] * 3 + [0.1] * 3 + [1e-3] * 3)
        f.set_noise(2e-3, 5e-2, None)
    t, err, err_icp, err_dead = 0.0, [], [], []
    for k in range(600):                                   # 60 s
        gyro, accel = [], []
        for _ in range(rate):
            yaw = w * t
            R = rot((0.0, 0.0, yaw))
            a_world = np.array([-speed * w * np.sin(yaw), speed * w * np.cos(yaw), 0.0])      # centripetal
            accel.append(R.T @ (a_world - np.array([0.0, 0.0, -9.80665])) + rng.normal(0, 5e-2, 3))
            gyro.append(np.array([0.0, 0.0, w]) + rng.normal(0, 2e-3, 3))
            t += dt
        for f in (ekf, ekf_icp, dead):
            f.predict_imu(np.array(gyro), np.array(accel), dt)
        yaw = w * t
        p_true = np.array([speed / w * np.sin(yaw), speed / w * (1.0 - np.cos(yaw)), 0.0])
        for f in (ekf, ekf_icp):
            f.update_position(p_true + rng.normal(0, 0.5, 3), np.diag([0.25] * 3))
            f.update_yaw(yaw + rng.normal(0, np.radians(2.0)), np.radians(2.0) ** 2)
        Tm = np.eye(4)
        Tm[:3, :3], Tm[:3, 3] = rot((0.0, 0.0, yaw)), p_true + rng.normal(0, 0.01, 3)
        ekf_icp.update_pose(Tm, [1e-4] * 3, [1e-6] * 3)
        if k >= 100:
            err.append(np.linalg.norm(ekf.state()[0][:3, 3] - p_true))
            err_icp.append(np.linalg.norm(ekf_icp.state()[0][:3, 3] - p_true))
            err_dead.append(np.linalg.norm(dead.state()[0][:3, 3] - p_true))
    assert np.median(err) < 0.35 and np.median(err_icp) < 0.02
    assert np.median(err_dead) > 3 * np.median(err)
    T, v, P = ekf.state()
    assert abs(np.linalg.norm(v) - speed) < 0.5 and np.sqrt(P[0, 0]) < 0.5


def test_ekf_jacobians_against_numerical_differentiation():
    """The transition Jacobians are checked against the NOMINAL propagation itself (the numpy
    restatement shares the equations with the C++ and cannot catch a wrong Jacobian): perturb the state by
    an error vector, propagate both, and read the propagated error back -- it must equal F delta to first order."""
    from oracle import ekf_np
    rng = np.random.default_rng(0)

    def boxplus(f, d):
        g = ekf_np.Ekf()
        g.p, g.v, g.R = f.p + d[0:3], f.v + d[3:6], f.R @ ekf_np.so3_exp(d[6:9])
        g.bg, g.ba, g.g = f.bg + d[9:12], f.ba + d[12:15], f.g
        return g

    def boxminus(a, b):
        return np.concatenate([a.p - b.p, a.v - b.v, ekf_np.so3_log(b.R.T @ a.R), a.bg - b.bg, a.ba - b.ba])

    base = ekf_np.Ekf()
    base.p, base.v, base.R = rng.normal(size=3), rng.normal(size=3), ekf_np.so3_exp(rng.normal(0, 0.5, 3))
    base.bg, base.ba = rng.normal(0, 0.01, 3), rng.normal(0, 0.1, 3)
    wm, am, dt = rng.normal(0, 0.5, 3), rng.normal(0, 2.0, 3) + [0, 0, 9.8], 0.01
    dR, dtr = ekf_np.so3_exp(rng.normal(0, 0.05, 3)), rng.normal(0, 0.3, 3)
    Tp = np.eye(4)
    Tc = np.eye(4)
    Tc[:3, :3], Tc[:3, 3] = dR, dtr
    F_imu = base.imu_jacobian(wm - base.bg, am - base.ba, dt)
    F_odo = base.odometry_jacobian(dR, dtr)
    eps = 1e-6
    for name, F, step in (("imu", F_imu, lambda f: f.predict_imu([wm], [am], dt)), ("odometry", F_odo, lambda f: f.predict_odometry(Tp, Tc))):
        num = np.zeros((15, 15))
        for k in range(15):
            d = np.zeros(15)
            d[k] = eps
            a, b = boxplus(base, d), boxplus(base, -d)
            step(a)
            step(b)
            num[:, k] = boxminus(a, b) / (2 * eps)
        # F = I + dt A is first order in dt: what is left is O(dt^2 |a|) (e.g. dp <- -R [a]x dtheta dt^2 / 2), two
        # orders below the dt |a| ~ 0.1 entries being checked; the odometry step's F is exact
        tol = 10.0 * dt * dt if name == "imu" else 1e-7
        assert np.abs(num - F).max() < max(tol, 1e-7), (name, np.abs(num - F).max())


def test_ekf_estimates_imu_biases(api_host):
    """Straight drive at 1 m/s with 1 deg/s of yaw (BASELINE config 4's stream), 100 Hz IMU with constant gyro and
    accelerometer biases, ICP-grade pose fixes at 10 Hz: the 15-state filter recovers the biases; the same filter
    with the bias states frozen (variance 0 = the round-1 9-state filter) keeps a velocity / tilt error."""
    rng = np.random.default_rng(4)
    dt, per, w = 0.01, 10, np.radians(1.0)
    bg_true, ba_true = np.array([0.004, -0.003, 0.002]), np.array([0.08, -0.05, 0.06])
    full, frozen = api_host.Ekf(), api_host.Ekf()
    for f in (full, frozen):
        f.reset(np.eye(4), [1.0, 0.0, 0.0], [1e-4] * 3 + [1e-2] * 3 + [1e-4] * 3)
        f.set_noise(1e-3, 2e-2, None)
    full.set_bias(None, None, [1e-4] * 3, [1e-1] * 3)
    full.set_bias_noise(1e-5, 1e-4)
    t, ev_full, ev_frozen = 0.0, [], []
    for k in range(600):
        gyro, accel = [], []
        for _ in range(per):
            yaw = w * t
            R = rot((0.0, 0.0, yaw))
            a_world = np.array([-w * np.sin(yaw), w * np.cos(yaw), 0.0])
            accel.append(R.T @ (a_world + np.array([0.0, 0.0, 9.80665])) + ba_true + rng.normal(0, 2e-2, 3))
            gyro.append(np.array([0.0, 0.0, w]) + bg_true + rng.normal(0, 1e-3, 3))
            t += dt
        yaw = w * t
        Tm = np.eye(4)
        Tm[:3, :3] = rot((0.0, 0.0, yaw))
        Tm[:3, 3] = [np.sin(yaw) / w, (1.0 - np.cos(yaw)) / w, 0.0]
        Tm[:3, 3] += rng.normal(0, 0.01, 3)
        v_true = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        for f, ev in ((full, ev_full), (frozen, ev_frozen)):
            f.predict_imu(np.array(gyro), np.array(accel), dt)
            if k >= 300:
                ev.append(np.linalg.norm(f.state()[1] - v_true))       # velocity error just before the fix
            f.update_pose(Tm, [1e-4] * 3, [1e-6] * 3)
    bg, ba, P = full.full_state()
    assert np.abs(bg - bg_true).max() < 5e-4 and np.abs(ba - ba_true).max() < 2e-2
    assert np.sqrt(np.diag(P)[9:12]).max() < 1e-3 and np.sqrt(np.diag(P)[12:15]).max() < 3e-2
    bgf, baf, _ = frozen.full_state()
    assert np.array_equal(bgf, np.zeros(3)) and np.array_equal(baf, np.zeros(3))
    assert np.median(ev_full) < 0.5 * np.median(ev_frozen)


def test_ekf_argument_errors(api_host):
    e = api_host.Ekf()
    with pytest.raises(api_host.SlamFusionError):
        e.predict_imu(np.zeros((2, 3)), np.zeros((2, 3)), 0.0)
    with pytest.raises(api_host.SlamFusionError):
        e.update_yaw(0.1, 0.0)
    with pytest.raises(api_host.SlamFusionError):
        e.set_noise(-1.0, 0.1)
    e.predict_imu(np.zeros((0, 3)), np.zeros((0, 3)), 0.01)       # nothing to integrate is fine
    with pytest.raises(api_host.SlamFusionError):
        e.set_bias(None, None, [-1.0, 0, 0], None)
    with pytest.raises(api_host.SlamFusionError):
        e.set_bias_noise(-1.0, 0.0)
    T, v, P = e.state()
    assert np.array_equal(T, np.eye(4)) and np.array_equal(P, np.eye(9))
