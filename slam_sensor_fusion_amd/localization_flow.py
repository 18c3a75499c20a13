"""The C++ node's per-scan orchestration (localization/src/localization_node.cpp:263-344)
over the C ABI: same order, same constants, same state variables — minus the ROS 2 shell
(rclcpp, message_filters, publishers: out of scope).  The start-up lock
(performCoarseAlignment: BruteForceAlignment, then the "strong" ICP) is included; callers that
already hold a lock set `coarse_alignment_complete_ = True`.

Host work (a14-a18, microseconds) runs in libslamfusion's C++ fusion functions; device
work is subsample -> radius crop -> (window change) -> ICP.  The reference re-crops the map and
rebuilds a FLANN tree every 3 m of travel (:299-305); here that is only a new window on the
resident whole-map index.
"""
import numpy as np

from . import api


class LocalizationFlow:
    # localization_node.h:142,145 and localization_node.cpp:19-35
    ref_frame_distance_ = 3.0
    cloud_crop_radius_ = 10.0
    icp_mode_ = "ref_cpp"              # ICPPointToPoint::calculateAlignment (icp_point_to_point.cpp:185-254)
    # grid cell of the whole-map index: the reference's "d2 < 0.5" rule (icp_point_to_point.cpp:70) is a 0.707 m search
    # radius on a stride-3 map (0.3 m point spacing).  Measured per alignment of ~13 k points (outer rings shared by the
    # wave, sf_nn.hpp): 0.72 m (every search inside the 27-cell block) 409 us, 0.5 m 349 us, 0.36 m 355 us, 0.25 m 381 us
    index_cell_ = 0.5
    pcl_crop_order_ = False            # True: the scan crop keeps PCL's ascending-distance output order (point_cloud_processing.hpp:40-52)

    def __init__(self, ctx, map_points, map_T_global, altitude_table=None, map_is_downsampled=True):
        self.ctx = ctx
        cloud = api.Cloud(ctx, np.asarray(map_points, dtype=np.float32))
        if not map_is_downsampled:
            cloud.voxel_downsample(0.1, "pcl")              # getMapCloud(0.1f), :19
        cloud.subsample(3)                                   # applyUniformSubsample(map_cloud_, 3), :20
        self.map_cloud_ = cloud
        self.map_index_ = api.Map(ctx, cloud, self.index_cell_)
        self.map_T_global_ = np.asarray(map_T_global, dtype=np.float64)
        self.altitude_table_ = np.zeros((0, 3)) if altitude_table is None else np.asarray(altitude_table, dtype=np.float64)
        self.icp_ = api.Icp(ctx, 0.5, 10, 0.05, 1e-5)        # :24-28
        self.icp_.set_target(self.map_index_)
        # a per-scan REF_CPP alignment runs as one launch (sf_icp_set_fused, on by default); where that does not apply (scans
        # too large to be resident at once) the ~50 small launches are replayed as one hipGraph: the scan's point count and the
        # map crop are read from device memory by those kernels, so the captured list survives both changing from scan to scan
        self.icp_.use_graph(True)
        self.coarse_pose_filter_ = api.StochasticFilter(4, 3.0)   # :32-34
        self.brute_force_alignment_ = api.BruteForceAlignment(ctx)   # :38-43
        self.brute_force_alignment_.setMeanErrorThreshold(0.1)
        self.brute_force_alignment_.setXYZStep(0.1, 0.1, 0.05)
        self.brute_force_alignment_.setXYZRange(1.5, 1.5, 0.1)
        self.brute_force_alignment_.setRotationStep(np.pi / 18.0)
        self.brute_force_alignment_.setRotationRange(np.pi / 6.0)
        self.coarse_alignment_complete_ = False
        self.ref_cropped_map_cloud_ = None
        self.map_T_sensor_ = np.eye(4, dtype=np.float32)
        self.odom_T_sensor_previous_ = np.eye(4, dtype=np.float32)
        self.map_T_ref_ = np.eye(4, dtype=np.float32)
        self.have_window_ = False
        self.window_empty_ = False
        self.current_compass_yaw_ = 0.0
        self.first_time_ = True
        self.scan_cloud_ = None
        self.imu_ = None
        self.last = {}

    def compassCallback(self, compass_deg):                  # :62-77
        self.current_compass_yaw_ = api.compass_to_yaw(compass_deg)

    def computeGpsCoarsePoseInMapFrame(self, lat, lon):      # :112-128
        alt = api.closest_altitude(self.altitude_table_, lat, lon)
        return api.gps_pose(self.map_T_global_, self.current_compass_yaw_, lat, lon, alt)

    def performCoarseAlignment(self, scan_cloud):
        """localization_node.cpp:200-261: brute force over the pose grid, else the "strong" ICP."""
        bf = self.brute_force_alignment_
        if bf.firstAlignmentCompleted():
            return True
        if self.ref_cropped_map_cloud_ is None:              # cropPointCloudThroughRadius output, PCL order (:302)
            self.ref_cropped_map_cloud_ = self.map_cloud_.copy().crop_radius(self.map_T_ref_[:3, 3], self.cloud_crop_radius_, sorted=True)
        map_cloud_temp = self.ref_cropped_map_cloud_.copy()
        scan_cloud_temp = scan_cloud.copy()
        map_cloud_temp.subsample(15)                          # :211
        map_cloud_temp.remove_floor()                         # :212
        scan_cloud_temp.remove_floor()                        # :213
        bf.setInitialGuess(self.map_T_sensor_)
        bf.setSourceCloud(scan_cloud_temp)
        self._coarse_map_ = api.Map(self.ctx, map_cloud_temp, 0.0)
        bf.setTargetCloud(self._coarse_map_)
        self.last_coarse = dict(n_map=len(map_cloud_temp), n_scan=len(scan_cloud_temp))
        if not bf.alignClouds():
            # :221-247 — the ICP keeps this sparse target until the next re-crop, like the reference
            self.icp_.set_target(self._coarse_map_)
            self.icp_.set_source(scan_cloud_temp)
            self.icp_.set_initial_transformation(bf.getBestTransformation())
            self.icp_.set_max_correspondence_dist(5.0)
            self.icp_.set_transformation_epsilon(1e-2)
            self.icp_.set_acceptable_mean_error(0.4)
            self.icp_.set_num_iterations(80)
            icp_result = self.icp_.align("ref_cpp")
            self.last_coarse["icp"] = icp_result
            if icp_result["converged"]:
                self.icp_.set_max_correspondence_dist(0.5)
                self.icp_.set_transformation_epsilon(1e-5)
                self.icp_.set_acceptable_mean_error(0.05)
                self.icp_.set_num_iterations(10)
                bf.resetFirstAlignment(True)
                self.coarse_alignment_complete_ = True
                self.map_T_sensor_ = icp_result["T"]
                return True
            bf.resetFirstAlignment(False)
            return False
        self.coarse_alignment_complete_ = True
        self.map_T_sensor_ = bf.getBestTransformation()
        return True

    def localizationCallback(self, scan_xyz, gps, odom, imu=None):
        """gps = dict(latitude, longitude, altitude, position_covariance[9]);
        odom = dict(q_wxyz, t, covariance[36]); imu (extension flows only) = dict(gyro[n,3], accel[n,3], dt): the
        samples since the previous scan.  Returns map_T_sensor or None when gated."""
        self.imu_ = imu
        if gps["altitude"] < 0:                              # :269-276
            return None
        odom_T_sensor_current = api.quat_to_pose(odom["q_wxyz"], odom["t"])
        if self.first_time_:                                 # :278-283 -> :181-198
            self.map_T_sensor_ = self.computeGpsCoarsePoseInMapFrame(gps["latitude"], gps["longitude"])
            self.map_T_ref_ = self.map_T_sensor_.copy()
            self.odom_T_sensor_previous_ = odom_T_sensor_current
            self.first_time_ = False
            return None

        # PREPROCESSING :290-305
        if self.scan_cloud_ is None:
            self.scan_cloud_ = api.Cloud(self.ctx)           # one device cloud for every scan: its buffers persist
        scan = self.scan_cloud_
        if hasattr(scan_xyz, "point_step"):                  # a PointCloud2-like message: unpacked on the device (f-2)
            scan.from_pointcloud2(scan_xyz)
        else:
            scan.upload(scan_xyz)
        scan.subsample(2)
        # index order, not PCL's distance order: the order of the source points only moves the rounding of the float64
        # record sums (the reference's own float32 sums depend on it far more), and the sort is 50 us + two allocations per scan
        scan.crop_radius([0.0, 0.0, 0.0], self.cloud_crop_radius_, sorted=self.pcl_crop_order_)
        sensor_T_ref = api.mat4f_mul(api.mat4f_inverse(self.map_T_sensor_), self.map_T_ref_)
        # re-crop when the sensor has moved on, or while the crop holds no point at all (ref_cropped_map_cloud_->empty(), :299)
        if np.linalg.norm(sensor_T_ref[:3, 3].astype(np.float32)) > self.ref_frame_distance_ or not self.have_window_ or self.window_empty_:
            self.map_index_.window_sphere(self.map_T_sensor_[:3, 3], self.cloud_crop_radius_)
            self.window_empty_ = bool(self.map_index_.nn(np.asarray(self.map_T_sensor_[:3, 3], np.float32)[None])[0][0] < 0)
            self.icp_.set_target(self.map_index_)             # icp_->setTargetPointCloud(ref_cropped_map_cloud_), :303
            self.ref_cropped_map_cloud_ = None               # materialised only if the coarse phase needs it
            self.map_T_ref_ = self.map_T_sensor_.copy()
            self.have_window_ = True

        # COARSE ALIGNMENT :307-315
        if not self.coarse_alignment_complete_:
            if not self.performCoarseAlignment(scan):
                return None

        # FINE ALIGNMENT :318-338
        prior = self.pose_prior(gps, odom, odom_T_sensor_current)
        self.icp_.set_source(scan)
        self.icp_.set_initial_transformation(prior.astype(np.float32))
        result = self.icp_.align(self.icp_mode_)
        self.map_T_sensor_ = result["T"]                     # no has_converged check, :338
        self.after_alignment(result, scan)
        self.odom_T_sensor_previous_ = odom_T_sensor_current  # :341
        self.last.update(prior=prior, icp=result, n_scan=len(scan))
        return self.map_T_sensor_

    def pose_prior(self, gps, odom, odom_T_sensor_current):
        """The reference's prior (:318-332): odometry prediction, GPS/compass pose, covariance-weighted blend, StochasticFilter."""
        map_T_sensor_odom = api.odom_prediction(self.map_T_sensor_, self.odom_T_sensor_previous_, odom_T_sensor_current)
        map_T_sensor_gps = self.computeGpsCoarsePoseInMapFrame(gps["latitude"], gps["longitude"])
        odometry_gain, gps_compass_gain = api.pose_gains(gps["position_covariance"], odom["covariance"], fixed=False)
        prior = api.blend(odometry_gain, map_T_sensor_odom, gps_compass_gain, map_T_sensor_gps)
        self.coarse_pose_filter_.addPoseToQueue(prior)
        prior = self.coarse_pose_filter_.applyGaussianFilterToCurrentPose(self.map_T_sensor_, prior)
        self.last = dict(odom=map_T_sensor_odom, gps=map_T_sensor_gps, gains=(odometry_gain, gps_compass_gain))
        return prior

    def after_alignment(self, result, scan):
        pass


class NativeLocalizationFlow:
    """The same orchestration run by the library itself (sf_node_*, csrc/sf_node.cpp): one C call per scan instead of
    ~25 ctypes calls and their numpy conversions.  Same interface and the same poses, bit for bit, as LocalizationFlow
    (tests/test_gpu_node.py); what a C++ node would link against."""

    def __init__(self, ctx, map_points, map_T_global, altitude_table=None, map_is_downsampled=True):
        self.ctx = ctx
        self.node_ = api.Node(ctx, map_points, map_T_global, altitude_table, index_cell=LocalizationFlow.index_cell_,
                              pcl_crop_order=int(LocalizationFlow.pcl_crop_order_), icp_mode=LocalizationFlow.icp_mode_,
                              ref_frame_distance=LocalizationFlow.ref_frame_distance_, cloud_crop_radius=LocalizationFlow.cloud_crop_radius_,
                              map_is_downsampled=int(map_is_downsampled))
        self.icp_ = self.node_.icp
        self.brute_force_alignment_ = self.node_.bf
        self.last = {}

    map_T_sensor_ = property(lambda self: self.node_.get_pose(api.SF_NODE_POSE_MAP_T_SENSOR), lambda self, T: self.node_.set_pose(api.SF_NODE_POSE_MAP_T_SENSOR, T))
    map_T_ref_ = property(lambda self: self.node_.get_pose(api.SF_NODE_POSE_MAP_T_REF), lambda self, T: self.node_.set_pose(api.SF_NODE_POSE_MAP_T_REF, T))
    odom_T_sensor_previous_ = property(lambda self: self.node_.get_pose(api.SF_NODE_POSE_ODOM_PREVIOUS), lambda self, T: self.node_.set_pose(api.SF_NODE_POSE_ODOM_PREVIOUS, T))
    coarse_alignment_complete_ = property(lambda self: self.node_.coarse_alignment_complete(), lambda self, v: self.node_.set_coarse_alignment_complete(v))

    def compassCallback(self, compass_deg):
        self.node_.compass(compass_deg)

    def localizationCallback(self, scan_xyz, gps, odom, imu=None):
        out = self.node_.callback(scan_xyz, gps, odom)
        self.out_ = out
        if out.status != api.SF_NODE_OK:
            return None
        # a copy per scan: api.Node hands out ONE structure that every callback overwrites, and flow.last of an earlier
        # scan must keep saying what that scan saw (as LocalizationFlow.last does)
        self.last = _LazyLast(type(out).from_buffer_copy(out))
        return np.array(out.map_T_sensor, dtype=np.float32).reshape(4, 4)


class _LazyLast(dict):
    """flow.last of the native flow: converted from the output structure only when somebody reads it"""

    def __init__(self, out):
        super().__init__()
        self._out = out

    def __missing__(self, key):
        o = self._out
        m = lambda a: np.array(a, dtype=np.float32).reshape(4, 4)
        v = {"prior": lambda: m(o.prior), "icp": lambda: o.icp.as_dict(), "n_scan": lambda: int(o.n_scan), "odom": lambda: m(o.odom_pose), "gps": lambda: m(o.gps_pose),
             "gains": lambda: (float(o.odometry_gain), float(o.gps_compass_gain))}[key]()
        self[key] = v
        return v


class EkfLocalizationFlow(LocalizationFlow):
    """The same per-scan orchestration with the pose prior from the error-state EKF (extension f-4, sf_ekf_*)
    instead of blend + StochasticFilter: odometry-delta prediction, GPS position and compass yaw updates before
    the alignment, the ICP pose as a measurement after it.  Not reference behaviour."""

    icp_pos_var_ = 0.05 ** 2            # the alignment stops at a 5 cm mean error (acceptable_mean_error)
    icp_rot_var_ = np.radians(0.5) ** 2
    compass_var_ = np.radians(2.0) ** 2
    start_sigma_m_ = 0.05
    start_sigma_rad_ = np.radians(0.5)

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.ekf_ = api.Ekf()
        self.ekf_started_ = False

    def pose_prior(self, gps, odom, odom_T_sensor_current):
        if not self.ekf_started_:
            # the start pose is a lock (coarse alignment or the caller's), not a GPS fix: a GPS-sized initial
            # covariance would let the first 0.5 m GPS sample drag the prior out of the ICP's reach
            self.ekf_.reset(self.map_T_sensor_.astype(np.float64), None, [self.start_sigma_m_ ** 2] * 3 + [1.0] * 3 + [self.start_sigma_rad_ ** 2] * 3)
            self.ekf_started_ = True
        cov = np.asarray(odom["covariance"], dtype=np.float64).reshape(6, 6)
        self.ekf_.predict_odometry(self.odom_T_sensor_previous_.astype(np.float64), odom_T_sensor_current.astype(np.float64),
                                   np.diag(cov)[:3], np.diag(cov)[3:])
        if "map_xyz" in gps:                                 # GPS already expressed in the map frame
            p_gps = np.asarray(gps["map_xyz"], dtype=np.float64)
        else:
            p_gps = self.computeGpsCoarsePoseInMapFrame(gps["latitude"], gps["longitude"])[:3, 3].astype(np.float64)
        self.ekf_.update_position(p_gps, np.asarray(gps["position_covariance"], dtype=np.float64).reshape(3, 3))
        self.ekf_.update_yaw(float(self.current_compass_yaw_), self.compass_var_)
        prior = self.ekf_.state()[0]
        self.last = dict(gps=p_gps)
        return prior.astype(np.float32)

    def after_alignment(self, result, scan):
        self.ekf_.update_pose(np.asarray(result["T"], dtype=np.float64), [self.icp_pos_var_] * 3, [self.icp_rot_var_] * 3)
        self.map_T_sensor_ = self.ekf_.state()[0].astype(np.float32)


class ImuEkfMappingFlow(EkfLocalizationFlow):
    """BASELINE config 4 as it is worded: the per-scan orchestration with (i) the 15-state EKF driven by the IMU
    samples that arrived since the previous scan (sf_ekf_predict_imu: pre-integration with gyro / accelerometer bias
    states) instead of the odometry delta, GPS + compass + ICP pose as measurements, and (ii) incremental map
    growth: every registered scan is transformed into the map frame (applyTransformation, float32) and collected;
    every `grow_every` scans (the recorder's tile cadence, mapping/include/mapping/map_data_save_node.h:72) the
    collected points are appended to the map cloud (`*map_cloud += *cloud`, global_map_frames_manager.cpp:131), the
    voxel grid is applied again (:142-146) and the NN index rebuilt -- all on the device.  Not reference behaviour
    (the reference localises against a fixed map and never reads the IMU); every piece it is built from is."""

    # A map that grows by the vehicle's own registrations inherits every registration error, so the alignment runs to
    # convergence (the Python twin's registration_icp semantics: NN in every iteration, float64, relative criteria
    # 1e-6) instead of stopping at the C++ node's 5 cm mean-error rule, which is sized for a fixed surveyed map.
    # Two more settings follow from the same concern.  Scan points that fall just beyond the frontier of the known map
    # find the frontier's points as "nearest neighbours" and pull the pose back towards the known side: measured on
    # the oracle, -4.2 mm per registration at the reference's 0.5 m correspondence distance, -0.2 mm at 0.2 m (the
    # IMU-driven prior is good to a few cm, so 0.2 m loses nothing); against the stride-3 index of the C++ node
    # (localization_node.cpp:20) the same pull is -12 mm, so the grown map is indexed at full resolution.
    icp_mode_ = "o3d_p2p"
    mapping_icp_iterations_ = 30       # localization_node.py:236
    mapping_max_corr_ = 0.2
    icp_pos_var_ = 0.01 ** 2
    icp_rot_var_ = np.radians(0.1) ** 2
    grow_every_ = 10
    voxel_ = 0.1
    index_stride_ = 1
    origin_lattice_cells_ = 64
    gyro_sigma_, accel_sigma_ = 2e-3, 5e-2
    gyro_bias_var_, accel_bias_var_ = 1e-4, 1e-2
    gyro_bias_walk_, accel_bias_walk_ = 1e-5, 1e-4
    start_velocity_var_ = 1.0

    def __init__(self, ctx, map_points, map_T_global, altitude_table=None, grow_every=None, voxel_flavour="pcl"):
        super().__init__(ctx, map_points, map_T_global, altitude_table)
        if grow_every is not None:
            self.grow_every_ = int(grow_every)
        self.icp_.set_num_iterations(self.mapping_icp_iterations_)
        self.icp_.set_max_correspondence_dist(self.mapping_max_corr_)
        self.map_index_.set_origin_lattice(self.origin_lattice_cells_)   # a map that grows: the grid origin survives growth in any direction (sf_map_patch)
        self.index_cloud_ = self.map_cloud_
        if self.index_stride_ != 3:                           # the parent indexed the stride-3 copy
            self.index_cloud_ = api.Cloud(ctx, np.asarray(map_points, dtype=np.float32))
            self.index_cloud_.subsample(self.index_stride_)
            self.map_index_.build(self.index_cloud_, 0.0)
            self.map_cloud_ = self.index_cloud_
            self.icp_.set_target(self.map_index_)
        self.voxel_flavour_ = voxel_flavour
        self.map_full_ = api.Cloud(ctx, np.asarray(map_points, dtype=np.float32))   # the voxel-filtered map at full resolution
        self.pending_ = api.Cloud(ctx, np.zeros((0, 3), np.float32))
        self.registered_ = api.Cloud(ctx)
        self.scans_since_growth_ = 0
        self.growths_ = 0
        self.merges_ = 0                                      # growth steps that took the merge path of sf_cloud_voxel_merge
        self.patches_ = 0                                     # ... and whose index was merged from the old one (sf_map_patch)
        self.patch_codes_ = {}                                # sf_map_patch's answer (1 / SF_PATCH_*) -> growth steps
        self.on_grow = None                                   # test hook: on_grow(flow) just before a growth step

    def pose_prior(self, gps, odom, odom_T_sensor_current):
        if not self.ekf_started_:
            self.ekf_.reset(self.map_T_sensor_.astype(np.float64), None,
                            [self.start_sigma_m_ ** 2] * 3 + [self.start_velocity_var_] * 3 + [self.start_sigma_rad_ ** 2] * 3)
            self.ekf_.set_noise(self.gyro_sigma_, self.accel_sigma_, None)
            self.ekf_.set_bias(None, None, [self.gyro_bias_var_] * 3, [self.accel_bias_var_] * 3)
            self.ekf_.set_bias_noise(self.gyro_bias_walk_, self.accel_bias_walk_)
            self.ekf_started_ = True
        imu = self.imu_
        if imu is not None and len(imu["gyro"]) > 0:
            self.ekf_.predict_imu(imu["gyro"], imu["accel"], float(imu["dt"]))
        else:                                                # no IMU samples for this interval: the odometry delta
            cov = np.asarray(odom["covariance"], dtype=np.float64).reshape(6, 6)
            self.ekf_.predict_odometry(self.odom_T_sensor_previous_.astype(np.float64), odom_T_sensor_current.astype(np.float64),
                                       np.diag(cov)[:3], np.diag(cov)[3:])
        if "map_xyz" in gps:
            p_gps = np.asarray(gps["map_xyz"], dtype=np.float64)
        else:
            p_gps = self.computeGpsCoarsePoseInMapFrame(gps["latitude"], gps["longitude"])[:3, 3].astype(np.float64)
        self.ekf_.update_position(p_gps, np.asarray(gps["position_covariance"], dtype=np.float64).reshape(3, 3))
        self.ekf_.update_yaw(float(self.current_compass_yaw_), self.compass_var_)
        prior = self.ekf_.state()[0]
        self.last = dict(gps=p_gps)
        return prior.astype(np.float32)

    def after_alignment(self, result, scan):
        super().after_alignment(result, scan)
        # the registered scan in the map frame, collected on the device
        self.registered_.copy_from(scan)
        self.registered_.transform(self.map_T_sensor_)
        self.pending_.append(self.registered_)
        self.scans_since_growth_ += 1
        if self.scans_since_growth_ >= self.grow_every_:
            self.grow_map()

    def grow_map(self):
        if self.on_grow is not None:
            self.on_grow(self)
        if self.voxel_flavour_ == "pcl":                      # the voxel filter of the concatenation as a merge into the filtered map (bit-identical)
            _, merged = self.map_full_.voxel_merge(self.pending_, self.voxel_)
            self.merges_ += int(merged)
        else:
            self.map_full_.append(self.pending_)
            self.map_full_.voxel_downsample(self.voxel_, self.voxel_flavour_)
        if self.index_stride_ == 1:                           # the index is built straight from the voxel-filtered map (it copies what it needs)
            if self.index_cloud_ is self.map_full_:           # ... and carried over the growth step when it indexed this very cloud before the merge
                self.patches_ += int(self.map_index_.patch(self.map_full_))
                self.patch_codes_[self.map_index_.last_patch] = self.patch_codes_.get(self.map_index_.last_patch, 0) + 1
            else:
                self.index_cloud_ = self.map_full_
                self.map_index_.build(self.index_cloud_, 0.0)
        else:
            self.index_cloud_.copy_from(self.map_full_)
            self.index_cloud_.subsample(self.index_stride_)
            self.map_index_.build(self.index_cloud_, 0.0)
        self.map_cloud_ = self.index_cloud_
        self.icp_.set_target(self.map_index_)
        self.have_window_ = False                             # the rebuilt index has no window yet: set at the next scan
        self.ref_cropped_map_cloud_ = None
        self.pending_.upload(np.zeros((0, 3), np.float32))
        self.scans_since_growth_ = 0
        self.growths_ += 1
