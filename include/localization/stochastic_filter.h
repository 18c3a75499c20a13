// include/localization/stochastic_filter.h — host-side mirror of the reference's
// StochasticFilter (/root/reference/localization/include/localization/stochastic_filter.h:12-76,
// localization/src/stochastic_filter.cpp) over the C ABI (sf_sfilter_*).
#ifndef SLAMFUSION_LOCALIZATION_STOCHASTIC_FILTER_H
#define SLAMFUSION_LOCALIZATION_STOCHASTIC_FILTER_H

#include "localization/icp_point_to_point.h"

class StochasticFilter {
public:
    StochasticFilter(const std::size_t queue_size = 10, const float n_std_dev_threshold = 1.0f)
        : f_(sf_sfilter_create((int)queue_size, n_std_dev_threshold))
    {
        if (!f_) throw std::runtime_error("sf_sfilter_create failed");
    }
    ~StochasticFilter() { sf_sfilter_destroy(f_); }
    StochasticFilter(const StochasticFilter &) = delete;
    StochasticFilter &operator=(const StochasticFilter &) = delete;

    void addPoseToQueue(const slamfusion::Matrix4f &origin_pose_current)
    {
        float rm[16];
        origin_pose_current.toRowMajor(rm);
        sf_sfilter_add_pose_to_queue(f_, rm);
    }
    void setMaximumLinearVelocity(const float max_linear_velocity) { sf_sfilter_set_maximum_linear_velocity(f_, max_linear_velocity); }
    slamfusion::Matrix4f applyGaussianFilterToCurrentPose(const slamfusion::Matrix4f &origin_pose_previous, const slamfusion::Matrix4f &origin_pose_current) const
    {
        float a[16], b[16], o[16];
        origin_pose_previous.toRowMajor(a);
        origin_pose_current.toRowMajor(b);
        sf_sfilter_apply_gaussian_filter(f_, a, b, o);
        return slamfusion::Matrix4f::fromRowMajor(o);
    }

private:
    sf_sfilter *f_;
};

#endif
