"""Duck-typed stand-ins for the ROS 2 messages the node consumes / produces (rclpy and the
message packages are absent in this image; with ROS 2 present the real messages work as well,
only attribute access is used).  Field names follow sensor_msgs/PointCloud2, NavSatFix,
nav_msgs/Odometry and std_msgs/Float64 / Header."""
from types import SimpleNamespace

import numpy as np


def Header(stamp=0.0, frame_id=""):
    return SimpleNamespace(stamp=stamp, frame_id=frame_id)


def Float64(data=0.0):
    return SimpleNamespace(data=float(data))


def NavSatFix(latitude=0.0, longitude=0.0, altitude=0.0, position_covariance=None, stamp=0.0):
    cov = np.zeros(9) if position_covariance is None else np.asarray(position_covariance, dtype=np.float64).reshape(9)
    return SimpleNamespace(header=Header(stamp, "gps"), latitude=float(latitude), longitude=float(longitude),
                           altitude=float(altitude), position_covariance=cov)


def Odometry(position=(0.0, 0.0, 0.0), orientation_xyzw=(0.0, 0.0, 0.0, 1.0), covariance=None, stamp=0.0,
             frame_id="", child_frame_id=""):
    p = SimpleNamespace(x=float(position[0]), y=float(position[1]), z=float(position[2]))
    q = SimpleNamespace(x=float(orientation_xyzw[0]), y=float(orientation_xyzw[1]), z=float(orientation_xyzw[2]),
                        w=float(orientation_xyzw[3]))
    cov = np.zeros(36) if covariance is None else np.asarray(covariance, dtype=np.float64).reshape(36)
    return SimpleNamespace(header=Header(stamp, frame_id), child_frame_id=child_frame_id,
                           pose=SimpleNamespace(pose=SimpleNamespace(position=p, orientation=q), covariance=cov))


def PointCloud2(xyz, stamp=0.0, frame_id="sensor"):
    """xyz32 cloud: `data` is the packed little-endian float32 x,y,z buffer (point_step 12),
    like sensor_msgs_py.point_cloud2.create_cloud_xyz32 produces."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    return SimpleNamespace(header=Header(stamp, frame_id), height=1, width=len(xyz), point_step=12,
                           row_step=12 * len(xyz), is_dense=False, data=xyz.tobytes())


def read_points_xyz(msg):
    """xyz float32 array of a PointCloud2-like message (packed xyz32 or any point_step >= 12
    with x,y,z at offsets 0,4,8, which is what /cloud_registered_body carries)."""
    buf = np.frombuffer(msg.data, dtype=np.uint8)
    n = msg.width * msg.height
    step = msg.point_step
    if step == 12:
        return buf.view(np.float32).reshape(n, 3)
    return np.ascontiguousarray(buf.reshape(n, step)[:, :12]).view(np.float32).reshape(n, 3)
