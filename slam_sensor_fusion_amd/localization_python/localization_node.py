"""LocalizationNode with the reference's method names, topic names, frame ids and constants
(localization_python/localization_python/localization_node.py:19-283), running the hot path
on the MI355X through libslamfusion.so.  rclpy is optional: without ROS 2 the node is a plain
object — feed it duck-typed messages (messages.py) and read what it published from
`node.published[topic]`.

What moved to the device (reference line -> here):
  :47      map_original.voxel_down_sample(0.1)      -> Cloud.voxel_downsample(0.1, "o3d") + Map (grid index, built once)
  :105-115 Python list-comprehension AABB crop      -> Cloud.crop_aabb (predicate + ballot compaction)
  :222-225 OrientedBoundingBox crop of the WHOLE map -> Map.window_obb (a predicate inside the search, no copy)
  :232-237 transform + registration_icp p2p         -> Icp.align("o3d_p2p") with the coarse pose as initial transform
"""
from time import time

import numpy as np
from scipy.spatial.transform import Rotation as R

from .. import api
from . import geo, messages

try:  # pragma: no cover - ROS 2 is not present in this image
    import rclpy
    from rclpy.node import Node as _RosNode
except ImportError:
    rclpy = None
    _RosNode = object


class _Logger:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(("info", msg))

    def warn(self, msg):
        self.lines.append(("warn", msg))


class LocalizationNode(_RosNode):
    #####################################################################
    # region Initialization
    #####################################################################
    def __init__(self, map_points=None, map_T_global=None, device=0, ctx=None):
        """map_points: N x 3 array standing for map.pcd (localization_node.py:34-45 loads it from
        ~/Desktop/map_data — PCD I/O is a §8(f-3) "next" row); map_T_global: 4x4 (map_T_global.npy)."""
        if rclpy is not None:  # pragma: no cover
            super().__init__('localization_node')
        self._logger = _Logger()
        self.published = {}

        # Frame transforms
        self.map_T_global = np.eye(4) if map_T_global is None else np.asarray(map_T_global, dtype=np.float64)

        # Map: voxel grid 0.1 (:47) and the device-resident NN index, once
        self.ctx = ctx if ctx is not None else api.Context(device)
        self.map_loaded = False
        self.map_original = None
        if map_points is not None:
            cloud = api.Cloud(self.ctx, np.asarray(map_points, dtype=np.float32))
            cloud.voxel_downsample(0.1, "o3d")
            self.map_original = cloud
            self.map_index = api.Map(self.ctx, cloud, 0.0)
            self.map_loaded = True
        self.map_t_global = self.map_T_global[:3, 3]
        self.map_R_global = self.map_T_global[:3, :3]

        # Cloud, GPS and Compass parameters (:53-58)
        self.icp_conversion_threshold = 0.5  # [m]
        bbox_side = 15.0  # [m]
        self.min_boundaries = [0, -bbox_side / 2, 0]
        self.max_boundaries = [bbox_side, bbox_side / 2, bbox_side / 2]
        self.extent = np.array([bbox_side * 2, bbox_side, bbox_side])
        self.current_compass = None

        # Odometry parameters (:61-62)
        self.map_T_sensor = np.eye(4)
        self.odom_previous_T_sensor = np.eye(4)

        self.icp = api.Icp(self.ctx, self.icp_conversion_threshold, 30, 0.05, 1e-5)
        if self.map_loaded:
            self.icp.set_target(self.map_index)
        self.get_logger().info('Localization node started!')

    def get_logger(self):
        return self._logger

    def _publish(self, topic, msg):
        self.published.setdefault(topic, []).append(msg)

    #####################################################################
    # region Conversions
    #####################################################################
    def readFilterPtcRegionPoints(self, ptc_msg) -> np.ndarray:
        # :105-115 — skip_nans + inclusive AABB, on the device
        pts = messages.read_points_xyz(ptc_msg)
        cloud = api.Cloud(self.ctx, pts)
        cloud.crop_aabb(self.min_boundaries, self.max_boundaries)
        self._cropped_scan_cloud = cloud
        return cloud.download().astype(np.float64)

    def buildNavOdomMsg(self, T: np.ndarray, frame_id: str, child_frame_id: str, stamp: float):
        # :117-131
        q = R.from_matrix(T[:3, :3]).as_quat()
        return messages.Odometry(position=(T[0, 3], T[1, 3], T[2, 3]), orientation_xyzw=q, stamp=stamp,
                                 frame_id=frame_id, child_frame_id=child_frame_id)

    def computeGpsCoarsePoseInMapFrame(self, gps_msg) -> np.ndarray:
        # :133-147
        global_R_sensor = R.from_euler('xyz', [0, 0, self.current_compass]).as_matrix()
        utm_e, utm_n, _, _ = geo.from_latlon(gps_msg.latitude, gps_msg.longitude)
        global_t_sensor = np.array([utm_e, utm_n, gps_msg.altitude])
        global_T_sensor = np.eye(4)
        global_T_sensor[:3, :3] = global_R_sensor
        global_T_sensor[:3, 3] = global_t_sensor
        return self.map_T_global @ global_T_sensor

    def computeModelPosePredictionFromOdometry(self, odometry_msg):
        # :149-167 — note the LEFT multiplication (differs from the C++ node)
        o = odometry_msg.pose.pose.orientation
        p = odometry_msg.pose.pose.position
        odom_current_T_sensor = np.eye(4)
        odom_current_T_sensor[0:3, 0:3] = R.from_quat(np.array([o.x, o.y, o.z, o.w])).as_matrix()
        odom_current_T_sensor[:3, 3] = np.array([p.x, p.y, p.z])
        odom_current_T_odom_previous = odom_current_T_sensor @ np.linalg.inv(self.odom_previous_T_sensor)
        return odom_current_T_sensor, odom_current_T_odom_previous @ self.map_T_sensor

    #####################################################################
    # region Callbacks
    #####################################################################
    def publishMapCallback(self):
        # :174-183
        if not self.map_loaded:
            self.get_logger().warn('Map not loaded yet, not publishing ...')
            return
        self._publish('/localization/map', messages.PointCloud2(self.map_original.download(), frame_id='map'))

    def compassCallback(self, compass_msg) -> None:
        # :185-191
        self.current_compass = np.radians(90 - compass_msg.data)
        if self.current_compass > np.pi:
            self.current_compass -= 2 * np.pi
        elif self.current_compass < -np.pi:
            self.current_compass += 2 * np.pi

    def syncCallback(self, pointcloud_msg, odometry_msg, gps_msg) -> None:
        # :193-269
        if not self.map_loaded:
            self.get_logger().warn('Map not loaded yet, not localizing ...')
            return
        if not self.current_compass:  # yaw == 0.0 counts as "not received", like the reference (:197)
            self.get_logger().warn('Compass not received yet, not localizing ...')
            return
        self.get_logger().info('Localization callback called!')
        start = time()

        odom_current_T_sensor, map_current_T_sensor_odom = self.computeModelPosePredictionFromOdometry(odometry_msg)
        map_current_T_sensor_gps = self.computeGpsCoarsePoseInMapFrame(gps_msg)

        gps_compass_weight = 0.2
        model_weight = 1 - gps_compass_weight
        map_T_sensor_coarse = gps_compass_weight * map_current_T_sensor_gps + model_weight * map_current_T_sensor_odom

        self.readFilterPtcRegionPoints(ptc_msg=pointcloud_msg)
        cropped_scan = self._cropped_scan_cloud
        # OBB crop of the map (:222-225) -> window on the resident index
        self.map_index.window_obb(map_T_sensor_coarse[:3, 3], map_T_sensor_coarse[:3, :3], self.extent)
        if self.map_index.window_count() == 0:
            self.get_logger().warn('Cropped map has no points, not localizing ...')
            return

        start_icp = time()
        self.icp.set_source(cropped_scan)
        self.icp.set_initial_transformation(np.asarray(map_T_sensor_coarse, dtype=np.float64))
        result = self.icp.align("o3d_p2p")
        end_icp = time()
        self.get_logger().info('ICP time: {}'.format(end_icp - start_icp))
        self.last_icp_result = result

        # lidar_pose_adjustment.transformation @ map_T_sensor_coarse (:243): the device composes
        # every update onto the initial transform, so T64 already is that product
        self.map_T_sensor = result["T64"]
        stamp = odometry_msg.header.stamp
        self._publish('/localization/map_T_sensor', self.buildNavOdomMsg(self.map_T_sensor, 'map', 'sensor', stamp))
        self.odom_previous_T_sensor = odom_current_T_sensor
        end = time()
        self.get_logger().info('Callback time: {}'.format(end - start))

        self._publish('/localization/map_T_sensor_coarse', self.buildNavOdomMsg(map_T_sensor_coarse, 'map', 'sensor', stamp))
        self._publish('/localization/odom_T_sensor', self.buildNavOdomMsg(odom_current_T_sensor, 'map', 'sensor', stamp))
        self._publish('/localization/map_T_sensor_gps', self.buildNavOdomMsg(map_current_T_sensor_gps, 'map', 'sensor_gps', stamp))
        moved = cropped_scan.copy().transform(self.map_T_sensor.astype(np.float32))
        self._publish('/localization/cropped_scan_map_frame', messages.PointCloud2(moved.download(), stamp=stamp, frame_id='map'))


def main(args=None):  # pragma: no cover - needs ROS 2
    if rclpy is None:
        raise SystemExit("rclpy is not available: construct LocalizationNode(map_points=...) directly and feed it messages")
    rclpy.init(args=args)
    node = LocalizationNode()
    rclpy.spin(node)
    node.destroy_node()
    rclpy.shutdown()
