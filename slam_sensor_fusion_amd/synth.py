"""Seeded synthetic maps, scans and sensor streams.

The reference ships no recorded data (its map folder ~/Desktop/map_data is external,
localization/src/localization_node.cpp:7), so parity tests and the bench use these
generators.  numpy only; everything is float32 after generation.
"""
import numpy as np

MAP_SEED = 1000
SCAN_SEED = 2000
STREAM_SEED = 3000
DENSITY = 1000.0          # raw points per m^2 of footprint (= 100 pts/m^3 over 10 m height)
T_TRUE_XYZ = (0.10, -0.05, 0.02)
T_TRUE_RPY_DEG = (0.02, -0.03, 0.10)
NOISE_SIGMA = 0.01


def rpy_to_R(roll, pitch, yaw):
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_T(xyz, rpy_deg):
    T = np.eye(4)
    T[:3, :3] = rpy_to_R(*np.radians(rpy_deg))
    T[:3, 3] = xyz
    return T


def t_true():
    return make_T(T_TRUE_XYZ, T_TRUE_RPY_DEG)


def make_map(m_points, seed=MAP_SEED):
    """M raw points uniform in [-L/2, L/2]^2 x [-5, 5] m with L = sqrt(M / 1000)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    L = float(np.sqrt(m_points / DENSITY))
    pts = np.empty((m_points, 3), dtype=np.float32)
    pts[:, 0] = rng.uniform(-L / 2, L / 2, m_points)
    pts[:, 1] = rng.uniform(-L / 2, L / 2, m_points)
    pts[:, 2] = rng.uniform(-5.0, 5.0, m_points)
    return pts


def make_scan(map_ds, n_points, scan_id=0, T=None, sigma=NOISE_SIGMA):
    """N noisy samples of the (downsampled) map seen from the frame T (default T_true):
    p_scan = T^-1 (p_map + n).  Registration with the identity prior should recover T."""
    rng = np.random.Generator(np.random.PCG64(SCAN_SEED + scan_id))
    T = t_true() if T is None else np.asarray(T, dtype=np.float64)
    n_points = min(n_points, len(map_ds))
    idx = rng.choice(len(map_ds), size=n_points, replace=False)
    p = map_ds[idx].astype(np.float64) + rng.normal(0.0, sigma, (n_points, 3))
    Tinv = np.linalg.inv(T)
    s = p @ Tinv[:3, :3].T + Tinv[:3, 3]
    return s.astype(np.float32), idx


def pose_error(T_est, T_ref):
    """(translation error [m], rotation error [rad]) between two 4x4 poses."""
    T_est = np.asarray(T_est, dtype=np.float64)
    T_ref = np.asarray(T_ref, dtype=np.float64)
    dt = float(np.linalg.norm(T_est[:3, 3] - T_ref[:3, 3]))
    dR = T_est[:3, :3] @ T_ref[:3, :3].T
    c = (np.trace(dR) - 1.0) / 2.0
    # robust small-angle: use the skew part
    skew = 0.5 * np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
    ang = float(np.arctan2(np.linalg.norm(skew), c))
    return dt, ang


def make_stream(n_scans, seed=STREAM_SEED):
    """Config-4 stream: truth advancing 0.1 m/scan along +x with 0.1 deg/scan yaw; odometry =
    truth + 2 mm drift per step; GPS = truth + N(0, 0.5 m); compass = yaw + N(0, 2 deg)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    truth, odom, gps_xyz, compass = [], [], [], []
    drift = np.zeros(3)
    for k in range(n_scans):
        T = make_T((0.1 * k, 0.0, 0.0), (0.0, 0.0, 0.1 * k))
        truth.append(T)
        drift = drift + rng.normal(0.0, 0.002, 3)
        To = T.copy()
        To[:3, 3] += drift
        odom.append(To)
        gps_xyz.append(T[:3, 3] + rng.normal(0.0, 0.5, 3))
        compass.append(np.radians(0.1 * k) + rng.normal(0.0, np.radians(2.0)))
    return dict(truth=np.array(truth), odom=np.array(odom), gps_xyz=np.array(gps_xyz),
                compass=np.array(compass), gps_cov=np.diag([0.25, 0.25, 0.25]),
                odom_cov=np.diag([1e-4] * 6))
