"""Drop-in for the reference's `localization_python.optimize_global_map_pose` (MapBuilder, make_map_data;
/root/reference: localization_python/localization_python/optimize_global_map_pose.py:8-121) over this build's own file
layer: the tiles are read and the merged map is written by libslamfusion's PCD v0.7 code (sf_pcd_read /
sf_pcd_write_binary, csrc/sf_io.cpp) instead of Open3D, the UTM series is geo.from_latlon (= utm.from_latlon), the
Euler composition is the intrinsic-free 'xyz' of scipy's Rotation.from_euler.  Offline, host-side code: no device work.

Kept as the reference has it, quirks included:
  * load_odom_positions counts the LEADING poses whose norm is below 0.5 m and stops at the first one beyond (:21-32);
  * load_global_poses slices pose[3:7] from every row of gps_imu_poses.txt (:42) -- a 6-column file (lat lon alt r p y,
    the comment at :38) yields the three angles; the recorder's own 4-column file (lat lon alt y,
    mapping/src/map_data_save_node.cpp:29,93-97) yields ONE value and Rotation.from_euler('xyz', ...) then raises
    ValueError, exactly as the reference does with scipy;
  * tiles are merged in os.listdir order and map.pcd itself is merged again on a second run (:52-62);
  * make_map_data joins map_folder twice when map_name is relative (:113,62) -- os.path.join keeps that harmless only for
    absolute folders; reproduced by doing the same joins.
"""
import os

import numpy as np

from . import geo


class PointCloud:
    """The two things the callers use of o3d.geometry.PointCloud: `points` and `+=`."""

    def __init__(self, points=None):
        self.points = np.zeros((0, 3), np.float64) if points is None else np.asarray(points, dtype=np.float64).reshape(-1, 3)

    def __iadd__(self, other):
        self.points = np.concatenate([self.points, np.asarray(other.points, dtype=np.float64).reshape(-1, 3)])
        return self

    def __len__(self):
        return len(self.points)


def euler_xyz_to_matrix(angles):
    """scipy.spatial.transform.Rotation.from_euler('xyz', angles).as_matrix(): extrinsic rotations about x, then y, then z
    (R = Rz(c) Ry(b) Rx(a)); anything but three angles is the ValueError scipy raises."""
    a = np.asarray(angles, dtype=np.float64)
    if a.ndim != 1 or a.shape[0] != 3:
        raise ValueError("Expected `angles` to be at most 2-dimensional with width equal to number of axes specified, got %r for 3 axes" % (a.shape,))
    (sa, sb, sc), (ca, cb, cc) = np.sin(a), np.cos(a)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


class MapBuilder:
    def __init__(self, map_folder: str) -> None:
        self.map_folder = map_folder
        self.odom_poses_file_path = os.path.join(map_folder, "odometry_positions.txt")
        self.gps_imu_data_file_path = os.path.join(map_folder, "gps_imu_poses.txt")
        self.max_num_poses_to_optimize = 50
        self.map_pcd = PointCloud()
        self.map_T_global = np.eye(4)
        self.max_translation_pose_transform = 0.50  # [m]

    def load_odom_positions(self):
        odom_positions = np.loadtxt(fname=self.odom_poses_file_path, skiprows=1)
        count = 0
        for i in range(len(odom_positions)):
            if np.linalg.norm(odom_positions[i]) < self.max_translation_pose_transform:
                count += 1
            else:
                break
        return odom_positions, count

    def load_global_poses(self):
        gps_imu_poses = np.loadtxt(fname=self.gps_imu_data_file_path, skiprows=1)
        global_t_map_list, global_rpy_map_list = [], []
        for pose in gps_imu_poses:
            global_rpy_map_list.append(pose[3:7])
            utm_x, utm_y, _, _ = geo.from_latlon(pose[0], pose[1])
            global_t_map_list.append(np.asarray([utm_x, utm_y, pose[2]]))
        return global_rpy_map_list, global_t_map_list

    def create_save_map(self, map_pcd_name: str) -> bool:
        from slam_sensor_fusion_amd import api
        pcd_files = [f for f in os.listdir(self.map_folder) if f.endswith(".pcd")]
        if len(pcd_files) == 0:
            print("No pcd files found in the map folder!")
            return False
        for pcd_file in pcd_files:
            self.map_pcd += PointCloud(api.pcd_read(os.path.join(self.map_folder, pcd_file)))
        api.pcd_write_binary(os.path.join(self.map_folder, map_pcd_name), self.map_pcd.points.astype(np.float32))
        return True

    def optimize_map_T_global(self) -> np.ndarray:
        _, n_valid_poses = self.load_odom_positions()
        global_rpy_map, global_t_map = self.load_global_poses()
        n_poses = min(n_valid_poses, len(global_rpy_map), self.max_num_poses_to_optimize)
        print("Optimizing the global to map transformation using {} poses".format(n_poses))
        mean_rpy = np.mean(global_rpy_map[:n_poses], axis=0)
        mean_t = np.mean(global_t_map[:n_poses], axis=0)
        global_T_map = np.eye(4)
        global_T_map[:3, :3] = euler_xyz_to_matrix(mean_rpy)
        global_T_map[:3, 3] = mean_t
        self.map_T_global = np.linalg.inv(global_T_map)
        return self.map_T_global

    def get_map(self) -> PointCloud:
        return self.map_pcd

    def get_map_T_global(self) -> np.ndarray:
        return self.map_T_global


def make_map_data(map_folder: str, map_name: str):
    map_builder = MapBuilder(map_folder=map_folder)
    map_pcd_name = os.path.join(map_folder, map_name)
    if not map_builder.create_save_map(map_pcd_name=map_pcd_name):
        print("Failed to save map!")
        return PointCloud(), np.eye(4)
    map_T_global = map_builder.optimize_map_T_global()
    np.save(os.path.join(map_folder, "map_T_global.npy"), map_T_global)
    return map_builder.get_map(), map_builder.get_map_T_global()


if __name__ == "__main__":  # pragma: no cover
    _, _ = make_map_data(map_folder=os.path.join(os.getenv("HOME"), "Desktop/map_data"), map_name="map.pcd")
