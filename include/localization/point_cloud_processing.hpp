// include/localization/point_cloud_processing.hpp — host-side mirror of the reference's free
// functions (/root/reference/localization/include/localization/point_cloud_processing.hpp:
// 31-92) over the C ABI: same names, same argument meaning, same in/out behaviour.
#ifndef SLAMFUSION_LOCALIZATION_POINT_CLOUD_PROCESSING_H
#define SLAMFUSION_LOCALIZATION_POINT_CLOUD_PROCESSING_H

#include "localization/icp_point_to_point.h"

namespace slamfusion {
namespace detail {
struct CloudHandle {
    sf_cloud *h = nullptr;
    CloudHandle() { if (sf_cloud_create(default_context(), &h) != SF_OK) throw std::runtime_error(sf_last_error()); }
    ~CloudHandle() { sf_cloud_destroy(h); }
};
inline void check(int rc) { if (rc != SF_OK) throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error()); }
inline void upload(CloudHandle &c, const PointCloud &p) { check(sf_cloud_upload(c.h, p.xyz.data(), (int64_t)p.size())); }
inline void download(CloudHandle &c, PointCloud &p)
{
    int64_t n = 0;
    check(sf_cloud_size(c.h, &n));
    p.xyz.resize((std::size_t)n * 3);
    check(sf_cloud_download(c.h, p.xyz.data(), n, nullptr));
}
} // namespace detail
} // namespace slamfusion

// point_cloud_processing.hpp:31-53 — points within `radius` of T's translation, PCL order
static inline void cropPointCloudThroughRadius(const slamfusion::Matrix4f &T, const double radius, slamfusion::PointCloud::Ptr &cloud,
                                               slamfusion::PointCloud::Ptr &cropped_cloud)
{
    slamfusion::detail::CloudHandle c;
    slamfusion::detail::upload(c, *cloud);
    const float center[3] = {T(0, 3), T(1, 3), T(2, 3)};
    slamfusion::detail::check(sf_cloud_crop_radius(c.h, center, radius, /*sorted=*/1));
    slamfusion::detail::download(c, *cropped_cloud);
}

// point_cloud_processing.hpp:55-74
static inline void applyUniformSubsample(slamfusion::PointCloud::Ptr &cloud, const std::size_t point_step)
{
    slamfusion::detail::CloudHandle c;
    slamfusion::detail::upload(c, *cloud);
    slamfusion::detail::check(sf_cloud_subsample(c.h, (int)point_step));
    slamfusion::detail::download(c, *cloud);
}

// point_cloud_processing.hpp:76-92
static inline void removeFloor(slamfusion::PointCloud::Ptr &cloud)
{
    slamfusion::detail::CloudHandle c;
    slamfusion::detail::upload(c, *cloud);
    slamfusion::detail::check(sf_cloud_remove_floor(c.h));
    slamfusion::detail::download(c, *cloud);
}

#endif
