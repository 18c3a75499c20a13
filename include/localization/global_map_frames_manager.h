// include/localization/global_map_frames_manager.h — host-side mirror of the reference's GlobalMapFramesManager
// (/root/reference/localization/include/localization/global_map_frames_manager.h:34-100,
// localization/src/global_map_frames_manager.cpp) over the C ABI (sf_frames_*): same constructor, same three public
// methods.  The map cloud comes back as a device cloud (the voxel grid of :142-146 runs on the device) or, like the
// reference, as a host point set.
#ifndef SLAMFUSION_LOCALIZATION_GLOBAL_MAP_FRAMES_MANAGER_H
#define SLAMFUSION_LOCALIZATION_GLOBAL_MAP_FRAMES_MANAGER_H

#include "localization/icp_point_to_point.h"

namespace slamfusion {
// 4x4 double matrix, COLUMN-major like Eigen::Matrix4d
struct Matrix4d {
    double data[16];
    double &operator()(int r, int c) { return data[4 * c + r]; }
    double operator()(int r, int c) const { return data[4 * c + r]; }
    void toRowMajor(double out[16]) const
    {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) out[4 * r + c] = (*this)(r, c);
    }
    static Matrix4d fromRowMajor(const double in[16])
    {
        Matrix4d m;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) m(r, c) = in[4 * r + c];
        return m;
    }
};
} // namespace slamfusion

class GlobalMapFramesManager {
public:
    // global_map_frames_manager.cpp:3-66
    GlobalMapFramesManager(const std::string data_folder, const std::string map_name, const std::size_t num_poses_max)
        : fr_(sf_frames_create(data_folder.c_str(), map_name.c_str(), (int64_t)num_poses_max))
    {
        if (!fr_) throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
    }
    ~GlobalMapFramesManager() { sf_frames_destroy(fr_); }
    GlobalMapFramesManager(const GlobalMapFramesManager &) = delete;
    GlobalMapFramesManager &operator=(const GlobalMapFramesManager &) = delete;

    // :93-151 — map.pcd cache, or merge of the recorded tiles + voxel grid + save; the cloud stays on the device
    void getMapCloud(const float voxel_size, sf_cloud *out) const
    {
        int cached = 0;
        if (sf_frames_get_map_cloud(fr_, out, voxel_size, &cached) != SF_OK) throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
    }
    // the reference's signature: a host point set
    slamfusion::PointCloud::Ptr getMapCloud(const float voxel_size) const
    {
        sf_cloud *c = nullptr;
        if (sf_cloud_create(slamfusion::default_context(), &c) != SF_OK) throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
        auto out = std::make_shared<slamfusion::PointCloud>();
        try {
            getMapCloud(voxel_size, c);
            int64_t n = 0;
            sf_cloud_size(c, &n);
            out->xyz.resize((std::size_t)n * 3);
            if (sf_cloud_download(c, out->xyz.data(), n, &n) != SF_OK) throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
        } catch (...) {
            sf_cloud_destroy(c);
            throw;
        }
        sf_cloud_destroy(c);
        return out;
    }
    // :153-248
    slamfusion::Matrix4d getMapTGlobal()
    {
        double rm[16];
        if (sf_frames_get_map_T_global(fr_, rm) != SF_OK) throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
        return slamfusion::Matrix4d::fromRowMajor(rm);
    }
    // :69-91
    float getClosestAltitude(const double lat, const double lon) const { return sf_frames_get_closest_altitude(fr_, lat, lon); }

    // for sf_node_create: rows of (lat, lon, alt)
    std::vector<double> altitudeTable() const
    {
        int64_t rows = 0;
        sf_frames_altitude_table(fr_, nullptr, 0, &rows);
        std::vector<double> t((std::size_t)rows * 3);
        if (rows > 0) sf_frames_altitude_table(fr_, t.data(), rows, &rows);
        return t;
    }

private:
    sf_frames *fr_;
};

#endif
