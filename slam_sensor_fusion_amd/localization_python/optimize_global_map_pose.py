"""`localization_python.optimize_global_map_pose` of the reference (MapBuilder, make_map_data:
localization_python/localization_python/optimize_global_map_pose.py:8-121) as a facade over this build's file layer.
The interface is the reference's (class, method and attribute names, return types); the bodies are this build's own:
array expressions over the parsed files, tiles through libslamfusion's PCD v0.7 reader / writer (sf_pcd_read,
sf_pcd_write_binary: csrc/sf_io.cpp) instead of Open3D, UTM through geo.from_latlon (the series of the `utm` package),
the pose composed and inverted in closed form.  Offline, host-side: no device work.

Behaviour that callers of the reference may rely on, each covered by tests/test_map_builder.py:
  * only the LEADING run of odometry positions closer than 0.5 m to the origin counts (:21-32): a later return to the
    origin does not;
  * columns 3..6 of every gps_imu_poses.txt row are the Euler angles (:42): the 6-column layout of the comment at :38
    gives three, the recorder's 4-column layout (mapping/src/map_data_save_node.cpp:29,93-97) gives one -- and a
    ValueError, as scipy's Rotation.from_euler('xyz', ...) raises in the reference;
  * tiles are merged in directory-listing order, an existing map.pcd included on a second run (:52-62);
  * make_map_data hands create_save_map a name already joined with the folder (:113 with :62).
"""
import os

import numpy as np

from . import geo

POSE_LIMIT_M = 0.50       # odometry positions this close to the origin describe the start pose (:17)
POSES_USED_AT_MOST = 50   # (:14)


class PointCloud:
    """What the callers use of o3d.geometry.PointCloud: `points`, `+=`, len()."""

    def __init__(self, points=None):
        self.points = np.zeros((0, 3), np.float64) if points is None else np.asarray(points, dtype=np.float64).reshape(-1, 3)

    def __iadd__(self, other):
        self.points = np.concatenate([self.points, np.asarray(other.points, dtype=np.float64).reshape(-1, 3)])
        return self

    def __len__(self):
        return len(self.points)


def euler_xyz_to_matrix(angles):
    """Rotation.from_euler('xyz', angles).as_matrix(): extrinsic rotations about x, then y, then z, i.e. Rz(c) Ry(b) Rx(a),
    written out; anything but three angles is the ValueError scipy raises."""
    a = np.asarray(angles, dtype=np.float64)
    if a.ndim != 1 or a.shape[0] != 3:
        raise ValueError("Expected `angles` to be at most 2-dimensional with width equal to number of axes specified, got %r for 3 axes" % (a.shape,))
    (sa, sb, sc), (ca, cb, cc) = np.sin(a), np.cos(a)
    return np.array([[cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa],
                     [sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa],
                     [-sb, cb * sa, cb * ca]])


def _table(path):
    """A whitespace table with one header line, always two-dimensional."""
    return np.atleast_2d(np.loadtxt(path, skiprows=1))


def _leading_run(flags):
    """Length of the run of True values at the start of a boolean vector."""
    stops = np.flatnonzero(~np.asarray(flags, dtype=bool))
    return int(stops[0]) if stops.size else int(len(flags))


def _rigid_inverse(rotation, translation):
    """[R t; 0 1]^-1 = [R^T  -R^T t; 0 1]."""
    out = np.eye(4)
    out[:3, :3] = rotation.T
    out[:3, 3] = -rotation.T @ translation
    return out


class MapBuilder:
    def __init__(self, map_folder: str) -> None:
        self.map_folder = map_folder
        self.odom_poses_file_path = os.path.join(map_folder, "odometry_positions.txt")
        self.gps_imu_data_file_path = os.path.join(map_folder, "gps_imu_poses.txt")
        self.max_num_poses_to_optimize = POSES_USED_AT_MOST
        self.max_translation_pose_transform = POSE_LIMIT_M
        self.map_pcd = PointCloud()
        self.map_T_global = np.eye(4)

    def load_odom_positions(self):
        """-> (positions [n, 3], how many of the leading ones lie within max_translation_pose_transform of the origin)"""
        xyz = _table(self.odom_poses_file_path)
        return xyz, _leading_run(np.linalg.norm(xyz, axis=1) < self.max_translation_pose_transform)

    def load_global_poses(self):
        """-> (list of Euler-angle vectors, list of [easting, northing, altitude]) for every row of gps_imu_poses.txt"""
        rows = _table(self.gps_imu_data_file_path)
        fixes = [geo.from_latlon(lat, lon)[:2] for lat, lon in rows[:, :2]]
        return list(rows[:, 3:7]), [np.array([e, n, alt]) for (e, n), alt in zip(fixes, rows[:, 2])]

    def create_save_map(self, map_pcd_name: str) -> bool:
        from slam_sensor_fusion_amd import api
        names = [name for name in os.listdir(self.map_folder) if name.endswith(".pcd")]
        if not names:
            print("No pcd files found in the map folder!")
            return False
        tiles = [api.pcd_read(os.path.join(self.map_folder, name)) for name in names]
        self.map_pcd += PointCloud(np.concatenate(tiles))
        api.pcd_write_binary(os.path.join(self.map_folder, map_pcd_name), self.map_pcd.points.astype(np.float32))
        return True

    def optimize_map_T_global(self) -> np.ndarray:
        """map_T_global from the mean of the first poses (those recorded before the platform left its start pose, at most
        max_num_poses_to_optimize, no more than there are global poses): the inverse of [Rxyz(mean angles) | mean position]."""
        angles, positions = self.load_global_poses()
        used = min(self.load_odom_positions()[1], len(angles), self.max_num_poses_to_optimize)
        print("Optimizing the global to map transformation using {} poses".format(used))
        self.map_T_global = _rigid_inverse(euler_xyz_to_matrix(np.mean(angles[:used], axis=0)), np.mean(positions[:used], axis=0))
        return self.map_T_global

    def get_map(self) -> PointCloud:
        return self.map_pcd

    def get_map_T_global(self) -> np.ndarray:
        return self.map_T_global


def make_map_data(map_folder: str, map_name: str):
    builder = MapBuilder(map_folder=map_folder)
    if not builder.create_save_map(map_pcd_name=os.path.join(map_folder, map_name)):
        print("Failed to save map!")
        return PointCloud(), np.eye(4)
    np.save(os.path.join(map_folder, "map_T_global.npy"), builder.optimize_map_T_global())
    return builder.get_map(), builder.get_map_T_global()


if __name__ == "__main__":  # pragma: no cover
    make_map_data(map_folder=os.path.join(os.getenv("HOME"), "Desktop/map_data"), map_name="map.pcd")
