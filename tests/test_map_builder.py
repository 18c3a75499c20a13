"""The Python twin's offline map tool (`optimize_global_map_pose.{MapBuilder, make_map_data}`,
/root/reference localization_python/localization_python/optimize_global_map_pose.py:8-121) through the top-level
`localization_python` alias, against the oracle's restatement (scipy Rotation + the C restatement of utm.from_latlon).
Host-only: PCD tiles go through libslamfusion's sf_pcd_* (no device call).  Parity unpinned upstream (no fixture)."""
import os

import numpy as np
import pytest


def write_folder(api, d, rows_cols, n_odom_near=7):
    rng = np.random.default_rng(3)
    tiles = []
    for k in range(3):
        pts = rng.uniform(-5, 5, (400 + 50 * k, 3)).astype(np.float32)
        api.pcd_write_binary(os.path.join(d, "cloud_%d.pcd" % (k + 1)), pts)
        tiles.append(pts)
    odom = np.r_[rng.uniform(-0.2, 0.2, (n_odom_near, 3)), rng.uniform(2, 3, (4, 3)), rng.uniform(-0.1, 0.1, (3, 3))]   # near, far, near again (not counted)
    with open(os.path.join(d, "odometry_positions.txt"), "w") as f:
        f.write("tx ty tz\n")
        np.savetxt(f, odom, fmt="%.8f")
    n = len(odom)
    lat = -22.9 + rng.normal(0, 1e-5, n)
    lon = -43.2 + rng.normal(0, 1e-5, n)
    alt = 12.0 + rng.normal(0, 0.1, n)
    cols = [lat, lon, alt] + [rng.normal(m, 0.01, n) for m in (0.02, -0.01, 1.2)][:rows_cols - 3]
    if rows_cols == 4:
        cols = [lat, lon, alt, rng.normal(1.2, 0.01, n)]
    with open(os.path.join(d, "gps_imu_poses.txt"), "w") as f:
        f.write("lat lon alt r p y\n" if rows_cols == 6 else "lat lon alt y\n")
        np.savetxt(f, np.c_[tuple(cols)], fmt="%.8f")
    return tiles, odom, np.c_[tuple(cols)]


def test_make_map_data_matches_the_oracle(api, orc, tmp_path):
    from localization_python.optimize_global_map_pose import MapBuilder, make_map_data
    d = str(tmp_path)
    tiles, odom, rows = write_folder(api, d, 6)
    order = [f for f in os.listdir(d) if f.endswith(".pcd")]                     # the reference merges in os.listdir order
    cloud, T = make_map_data(d, "map.pcd")
    # file round trip of what was written (8 decimals): the oracle gets the same parsed numbers
    odom_p = np.loadtxt(os.path.join(d, "odometry_positions.txt"), skiprows=1)
    rows_p = np.loadtxt(os.path.join(d, "gps_imu_poses.txt"), skiprows=1)
    want, n = orc.map_builder_py(odom_p, rows_p)
    assert n == 7                                                                # the leading poses under 0.5 m; the later near ones do not count
    assert np.abs(T - want).max() < 1e-9 * max(1.0, np.abs(want).max())
    assert np.array_equal(np.load(os.path.join(d, "map_T_global.npy")), T)
    merged = np.concatenate([tiles[int(f.split("_")[1].split(".")[0]) - 1] for f in order])
    assert np.array_equal(cloud.points.astype(np.float32), merged)
    assert np.array_equal(api.pcd_read(os.path.join(d, "map.pcd")), merged)      # binary PCD v0.7 through sf_pcd_*
    mb = MapBuilder(d)
    _, count = mb.load_odom_positions()
    rpy, t = mb.load_global_poses()
    assert count == 7 and len(rpy) == len(rows) and rpy[0].shape == (3,)
    e, nn_ = orc.utm_from_latlon(rows_p[0, 0], rows_p[0, 1])
    assert abs(t[0][0] - e) < 1e-6 and abs(t[0][1] - nn_) < 1e-6 and t[0][2] == rows_p[0, 2]
    assert np.allclose(mb.get_map_T_global(), np.eye(4)) and len(mb.get_map()) == 0   # nothing computed yet


def test_the_recorders_four_column_file_raises_like_the_reference(api, orc, tmp_path):
    """gps_imu_poses.txt as mapping/src/map_data_save_node.cpp:29,93-97 writes it has 4 columns: pose[3:7] is ONE value and
    from_euler('xyz', ...) raises ValueError -- in the reference (scipy), in the oracle (scipy) and in the drop-in."""
    from localization_python.optimize_global_map_pose import MapBuilder
    d = str(tmp_path)
    write_folder(api, d, 4)
    mb = MapBuilder(d)
    with pytest.raises(ValueError):
        mb.optimize_map_T_global()
    with pytest.raises(ValueError):
        orc.map_builder_py(np.loadtxt(os.path.join(d, "odometry_positions.txt"), skiprows=1), np.loadtxt(os.path.join(d, "gps_imu_poses.txt"), skiprows=1))


def test_no_tiles_is_the_reference_failure_path(api, tmp_path):
    from localization_python.optimize_global_map_pose import make_map_data
    cloud, T = make_map_data(str(tmp_path), "map.pcd")
    assert len(cloud) == 0 and np.array_equal(T, np.eye(4))


def test_euler_xyz_equals_scipy():
    from scipy.spatial.transform import Rotation
    from localization_python.optimize_global_map_pose import euler_xyz_to_matrix
    rng = np.random.default_rng(1)
    for _ in range(20):
        a = rng.uniform(-3, 3, 3)
        assert np.abs(euler_xyz_to_matrix(a) - Rotation.from_euler('xyz', a).as_matrix()).max() < 1e-14
