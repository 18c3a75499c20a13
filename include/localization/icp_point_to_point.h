// include/localization/icp_point_to_point.h — host-side mirror (C++17, header only) of the
// reference's ICPPointToPoint class, implemented over the C ABI of libslamfusion.so.
//
// Same class name, same constructor, same setters, same calculateAlignment() and the same
// ICPResult fields as /root/reference/localization/include/localization/icp_point_to_point.h:
// 28-39 (ICPResult) and :41-136 (class), so the reference node's call sites
// (localization/src/localization_node.cpp:28-29,223-236,303,335-338) compile unchanged
// against this header.  PCL and Eigen are not present in this image, so the point-cloud and
// matrix types are the minimal stand-ins below; with PCL/Eigen available define
// SLAMFUSION_WITH_PCL_EIGEN before including and the overloads taking
// pcl::PointCloud<pcl::PointXYZ>::Ptr / Eigen::Matrix4f are enabled.
#ifndef SLAMFUSION_LOCALIZATION_ICP_POINT_TO_POINT_H
#define SLAMFUSION_LOCALIZATION_ICP_POINT_TO_POINT_H

#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "slamfusion.h"

#ifdef SLAMFUSION_WITH_PCL_EIGEN
#include <Eigen/Dense>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#endif

namespace slamfusion {

// 4x4 float matrix, COLUMN-major like Eigen::Matrix4f (so Eigen::Map<Eigen::Matrix4f>(m.data)
// is the same matrix); operator()(r, c) reads like Eigen.
struct Matrix4f {
    float data[16];
    Matrix4f() { *this = Identity(); }
    static Matrix4f Identity()
    {
        Matrix4f m(0);
        for (int d = 0; d < 4; ++d) m(d, d) = 1.0f;
        return m;
    }
    float &operator()(int r, int c) { return data[4 * c + r]; }
    float operator()(int r, int c) const { return data[4 * c + r]; }
    void toRowMajor(float out[16]) const
    {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) out[4 * r + c] = (*this)(r, c);
    }
    static Matrix4f fromRowMajor(const float in[16])
    {
        Matrix4f m(0);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) m(r, c) = in[4 * r + c];
        return m;
    }

private:
    explicit Matrix4f(int) { std::memset(data, 0, sizeof(data)); }
};

// xyz float32 triplets (pcl::PointXYZ without its padding)
struct PointCloud {
    std::vector<float> xyz;
    std::size_t size() const { return xyz.size() / 3; }
    void push_back(float x, float y, float z) { xyz.push_back(x); xyz.push_back(y); xyz.push_back(z); }
    using Ptr = std::shared_ptr<PointCloud>;
};

// one device context per process/thread (the reference node is single-threaded,
// localization/src/main.cpp:18)
inline sf_ctx *default_context()
{
    static sf_ctx *ctx = nullptr;
    if (!ctx && sf_ctx_create(0, nullptr, &ctx) != SF_OK)
        throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
    return ctx;
}

} // namespace slamfusion

// icp_point_to_point.h:28-39
struct ICPResult {
    ICPResult(slamfusion::Matrix4f T, float error, int iterations, bool has_converged)
        : transformation(T), error(error), iterations(iterations), has_converged(has_converged) {}
    ICPResult(slamfusion::Matrix4f T) : transformation(T) {}
    ICPResult() {}

    slamfusion::Matrix4f transformation = slamfusion::Matrix4f::Identity();
    float error{1e6};
    int iterations{0};
    bool has_converged{false};
};

class ICPPointToPoint {
public:
    // icp_point_to_point.h:49 / icp_point_to_point.cpp:3-12
    ICPPointToPoint(const float max_correspondence_dist, const int num_iterations, const float acceptable_mean_error, const float transformation_epsilon)
    {
        check(sf_icp_create(slamfusion::default_context(), max_correspondence_dist, num_iterations, acceptable_mean_error, transformation_epsilon, &icp_));
    }
    ~ICPPointToPoint() { sf_icp_destroy(icp_); }
    ICPPointToPoint(const ICPPointToPoint &) = delete;
    ICPPointToPoint &operator=(const ICPPointToPoint &) = delete;

    void setMaxCorrespondenceDist(const float v) { check(sf_icp_set_max_correspondence_dist(icp_, v)); }
    void setNumIterations(const int v) { check(sf_icp_set_num_iterations(icp_, v)); }
    void setTransformationEpsilon(const float v) { check(sf_icp_set_transformation_epsilon(icp_, v)); }
    void setAcceptableMeanError(const float v) { check(sf_icp_set_acceptable_mean_error(icp_, v)); }
    void setDebugMode(bool debug_mode) { check(sf_icp_set_debug_mode(icp_, debug_mode ? 1 : 0)); }

    void setInitialTransformation(const slamfusion::Matrix4f &initial_transformation)
    {
        float rm[16];
        initial_transformation.toRowMajor(rm);
        check(sf_icp_set_initial_transformation(icp_, rm));
    }
    void setSourcePointCloud(const slamfusion::PointCloud::Ptr &source_cloud)
    {
        check(sf_icp_set_source(icp_, source_cloud->xyz.data(), (int64_t)source_cloud->size()));
    }
    void setTargetPointCloud(const slamfusion::PointCloud::Ptr &target_cloud)
    {
        check(sf_icp_set_target(icp_, target_cloud->xyz.data(), (int64_t)target_cloud->size()));
    }
    // MI355X-native addition: share a prebuilt whole-map index instead of re-indexing a crop
    void setTargetMap(sf_map *map) { check(sf_icp_set_target_map(icp_, map)); }

#ifdef SLAMFUSION_WITH_PCL_EIGEN
    void setInitialTransformation(const Eigen::Matrix4f &T)
    {
        const Eigen::Matrix<float, 4, 4, Eigen::RowMajor> rm = T;
        check(sf_icp_set_initial_transformation(icp_, rm.data()));
    }
    void setSourcePointCloud(const pcl::PointCloud<pcl::PointXYZ>::Ptr &cloud) { check(sf_icp_set_source(icp_, pack(*cloud).data(), (int64_t)cloud->size())); }
    void setTargetPointCloud(const pcl::PointCloud<pcl::PointXYZ>::Ptr &cloud) { check(sf_icp_set_target(icp_, pack(*cloud).data(), (int64_t)cloud->size())); }
#endif

    // icp_point_to_point.cpp:185-254
    ICPResult calculateAlignment()
    {
        sf_icp_result r;
        check(sf_icp_align(icp_, SF_ICP_REF_CPP, &r));
        if (r.flags & SF_ICP_FLAG_FEW_CORR)
            std::cerr << "[ICP ERROR] Not enough valid correspondences found. Aborting." << std::endl;
        return ICPResult(slamfusion::Matrix4f::fromRowMajor(r.T), r.error, r.iterations, r.converged != 0);
    }

private:
    static void check(int rc)
    {
        if (rc != SF_OK) throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
    }
#ifdef SLAMFUSION_WITH_PCL_EIGEN
    static std::vector<float> pack(const pcl::PointCloud<pcl::PointXYZ> &c)
    {
        std::vector<float> v(3 * c.size());
        for (std::size_t i = 0; i < c.size(); ++i) { v[3 * i] = c[i].x; v[3 * i + 1] = c[i].y; v[3 * i + 2] = c[i].z; }
        return v;
    }
#endif
    sf_icp *icp_ = nullptr;
};

#endif // SLAMFUSION_LOCALIZATION_ICP_POINT_TO_POINT_H
