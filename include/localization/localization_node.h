// include/localization/localization_node.h — the arithmetic of the reference's LocalizationNode
// (/root/reference/localization/include/localization/localization_node.h:31-168, localization/src/localization_node.cpp)
// without its ROS 2 shell: the constructor's set-up (:10-59), compassCallback (:62-77) and localizationCallback
// (:263-344, coarse alignment :200-261 included) over the C ABI (sf_node_*), one call per message.  The rclcpp node keeps
// its subscriptions, the synchroniser and its publishers and owns one of these.
#ifndef SLAMFUSION_LOCALIZATION_LOCALIZATION_NODE_H
#define SLAMFUSION_LOCALIZATION_LOCALIZATION_NODE_H

#include "localization/global_map_frames_manager.h"

class LocalizationCore {
public:
    // map_cloud: getMapCloud(0.1f) of the frames manager (:19); the stride-3 subsample (:20), the ICP (:24-28), the
    // StochasticFilter (:32-34) and the brute-force pose grid (:38-43) are set up with the node's constants
    LocalizationCore(const slamfusion::PointCloud &map_cloud, const slamfusion::Matrix4d &map_T_global, const std::vector<double> &altitude_table_lat_lon_alt)
    {
        sf_node_params p;
        sf_node_default_params(&p);
        double rm[16];
        map_T_global.toRowMajor(rm);
        if (sf_node_create(slamfusion::default_context(), map_cloud.xyz.data(), (int64_t)map_cloud.size(), rm, altitude_table_lat_lon_alt.data(),
                           (int)(altitude_table_lat_lon_alt.size() / 3), &p, &node_) != SF_OK)
            throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
    }
    explicit LocalizationCore(GlobalMapFramesManager &frames) : LocalizationCore(*frames.getMapCloud(0.1f), frames.getMapTGlobal(), frames.altitudeTable()) {}
    ~LocalizationCore() { sf_node_destroy(node_); }
    LocalizationCore(const LocalizationCore &) = delete;
    LocalizationCore &operator=(const LocalizationCore &) = delete;

    // :62-77
    void compassCallback(const double compass_deg) { sf_node_compass(node_, compass_deg); }

    // :263-344.  Returns true when map_T_sensor holds a new pose (false: message gated, first message, or no lock yet).
    bool localizationCallback(const slamfusion::PointCloud &scan, const sf_gps_fix &gps, const sf_odom &odom, slamfusion::Matrix4f &map_T_sensor)
    {
        if (sf_node_callback_xyz(node_, scan.xyz.data(), (int64_t)scan.size(), &gps, &odom, &last_) != SF_OK)
            throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
        map_T_sensor = slamfusion::Matrix4f::fromRowMajor(last_.map_T_sensor);
        return last_.status == SF_NODE_OK;
    }
    // the same from the raw sensor_msgs/PointCloud2 fields
    bool localizationCallback(const void *data, int64_t data_bytes, int64_t width, int64_t height, int point_step, int64_t row_step, int off_x, int off_y, int off_z, int datatype,
                              bool is_bigendian, const sf_gps_fix &gps, const sf_odom &odom, slamfusion::Matrix4f &map_T_sensor)
    {
        if (sf_node_callback_pointcloud2(node_, data, data_bytes, width, height, point_step, row_step, off_x, off_y, off_z, datatype, is_bigendian ? 1 : 0, &gps, &odom, &last_) !=
            SF_OK)
            throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
        map_T_sensor = slamfusion::Matrix4f::fromRowMajor(last_.map_T_sensor);
        return last_.status == SF_NODE_OK;
    }

    const sf_node_output &last() const { return last_; } // prior, ICP result, gains, crop size of the last message
    void setPose(const slamfusion::Matrix4f &map_T_sensor)
    {
        float rm[16];
        map_T_sensor.toRowMajor(rm);
        sf_node_set_pose(node_, SF_NODE_POSE_MAP_T_SENSOR, rm);
        sf_node_set_pose(node_, SF_NODE_POSE_MAP_T_REF, rm);
    }
    void setCoarseAlignmentComplete(bool v) { sf_node_set_coarse_alignment_complete(node_, v ? 1 : 0); }
    bool coarseAlignmentComplete() const { return sf_node_coarse_alignment_complete(node_) != 0; }
    sf_node *handle() { return node_; }

private:
    sf_node *node_ = nullptr;
    sf_node_output last_{};
};

#endif
