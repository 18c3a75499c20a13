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
    R = Tinv[:3, :3]
    # (a [n, 3] x [3, 3] matmul takes numpy's slow small-dimension path: 0.12 s per 200 k points; this is the same sum)
    s = p[:, 0:1] * R[:, 0] + p[:, 1:2] * R[:, 1] + p[:, 2:3] * R[:, 2] + Tinv[:3, 3]
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


def make_imu(n_scans, scan_period=0.1, rate_hz=100, gyro_bias=(0.004, -0.003, 0.002), accel_bias=(0.08, -0.05, 0.06),
             gyro_sigma=2e-3, accel_sigma=5e-2, seed=STREAM_SEED + 1):
    """IMU samples for make_stream's truth (0.1 m and 0.1 deg per scan = 1 m/s along +x of the MAP and 1 deg/s of yaw at
    10 Hz scans; the path is straight while the sensor turns, so the world acceleration is zero): per scan interval
    k -> k+1 `rate_hz * scan_period` samples of gyro (rad/s) and specific force (m/s^2) in the sensor frame, with
    constant biases and white noise.  Returns (gyro[n_scans-1, m, 3], accel[n_scans-1, m, 3], dt)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    m = int(round(rate_hz * scan_period))
    dt = scan_period / m
    w = np.radians(0.1) / scan_period
    gyro = np.zeros((n_scans - 1, m, 3))
    accel = np.zeros((n_scans - 1, m, 3))
    gyro[..., 2] = w
    accel[..., 2] = 9.80665                                  # yaw-only attitude: gravity stays on the sensor's z axis
    gyro += np.asarray(gyro_bias) + rng.normal(0.0, gyro_sigma, gyro.shape)
    accel += np.asarray(accel_bias) + rng.normal(0.0, accel_sigma, accel.shape)
    return gyro, accel, dt


def make_corridor(length_m, width_m, x0=-12.0, seed=MAP_SEED + 7):
    """Raw points uniform in [x0, x0 + length] x [-width/2, width/2] x [-5, 5] m at the bench density (100 pts/m^3):
    the world a config-4 vehicle drives through (make_stream advances along +x)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(length_m * width_m * DENSITY)
    pts = np.empty((n, 3), dtype=np.float32)
    pts[:, 0] = rng.uniform(x0, x0 + length_m, n)
    pts[:, 1] = rng.uniform(-width_m / 2, width_m / 2, n)
    pts[:, 2] = rng.uniform(-5.0, 5.0, n)
    return pts


# ------------------------------------------------------------------ config 5: ring-structured scan vs a city map
CITY_SEED = 4000


def make_city(extent_m, n_boxes, seed=CITY_SEED):
    """Ground plane z = 0 over [-extent/2, extent/2]^2 plus seeded axis-aligned boxes (buildings).
    Returns boxes[n, 6] = (x0, y0, z0, x1, y1, z1) with z0 = 0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.uniform(-extent_m / 2, extent_m / 2, (n_boxes, 2))
    half = rng.uniform(3.0, 12.0, (n_boxes, 2))
    h = rng.uniform(3.0, 30.0, n_boxes)
    keep = np.linalg.norm(c, axis=1) > 15.0          # keep the sensor's start area free
    c, half, h = c[keep], half[keep], h[keep]
    return np.c_[c - half, np.zeros(len(c)), c + half, h]


def sample_city(boxes, extent_m, m_points, sigma=0.005, seed=CITY_SEED + 1):
    """M points on the city's surfaces (ground, walls, roofs), uniform by area, with sigma noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    dx, dy, dz = boxes[:, 3] - boxes[:, 0], boxes[:, 4] - boxes[:, 1], boxes[:, 5]
    # faces: ground, then per box: roof, 2 walls normal to x, 2 walls normal to y
    areas = np.concatenate([[extent_m * extent_m], dx * dy, dy * dz, dy * dz, dx * dz, dx * dz])
    nb = len(boxes)
    face = rng.choice(len(areas), size=m_points, p=areas / areas.sum())
    u, v = rng.uniform(0, 1, m_points), rng.uniform(0, 1, m_points)
    p = np.empty((m_points, 3))
    g = face == 0
    p[g] = np.c_[(u[g] - 0.5) * extent_m, (v[g] - 0.5) * extent_m, np.zeros(g.sum())]
    for kind in range(5):
        sel = (face >= 1 + kind * nb) & (face < 1 + (kind + 1) * nb)
        b = boxes[face[sel] - 1 - kind * nb]
        uu, vv = u[sel], v[sel]
        x = b[:, 0] + uu * (b[:, 3] - b[:, 0])
        y = b[:, 1] + (vv if kind == 0 else uu) * (b[:, 4] - b[:, 1])
        z = vv * b[:, 5]
        if kind == 0:
            p[sel] = np.c_[x, y, b[:, 5]]
        elif kind == 1:
            p[sel] = np.c_[b[:, 0], y, z]
        elif kind == 2:
            p[sel] = np.c_[b[:, 3], y, z]
        elif kind == 3:
            p[sel] = np.c_[x, b[:, 1], z]
        else:
            p[sel] = np.c_[x, b[:, 4], z]
    p += rng.normal(0.0, sigma, p.shape)
    return p.astype(np.float32)


def raycast_scan(boxes, T_sensor, rings=64, azimuths=2032, elev_deg=(-24.8, 2.0), max_range=80.0, sigma=NOISE_SIGMA, seed=CITY_SEED + 2):
    """Spinning-LiDAR scan (rings x azimuths rays, ring-major order like a driver's packet order)
    of the city from the pose T_sensor (map <- sensor).  Returns the hits in the SENSOR frame."""
    rng = np.random.Generator(np.random.PCG64(seed))
    el = np.radians(np.linspace(elev_deg[0], elev_deg[1], rings))[:, None]
    az = np.linspace(0.0, 2 * np.pi, azimuths, endpoint=False)[None, :]
    d_s = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el) * np.ones_like(az)], -1).reshape(-1, 3)
    T = np.asarray(T_sensor, dtype=np.float64)
    o, d = T[:3, 3], d_s @ T[:3, :3].T
    t_hit = np.full(len(d), np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = -o[2] / d[:, 2]                                        # ground z = 0
        t_hit = np.where((d[:, 2] < 0) & (tg > 0), tg, t_hit)
        inv = 1.0 / d
        for b in boxes:                                             # slab test per box, all rays at once
            t0, t1 = (b[:3] - o) * inv, (b[3:] - o) * inv
            tn, tf = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
            hit = (tn <= tf) & (tn > 0)
            t_hit = np.where(hit & (tn < t_hit), tn, t_hit)
    ok = t_hit < max_range
    p = d_s[ok] * t_hit[ok, None] + rng.normal(0.0, sigma, (ok.sum(), 3))
    return p.astype(np.float32)
