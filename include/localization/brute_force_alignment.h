// include/localization/brute_force_alignment.h — host-side mirror of the reference's
// BruteForceAlignment (/root/reference/localization/include/localization/
// brute_force_alignment.h:22-112, localization/src/brute_force_alignment.cpp) over the C ABI
// (sf_bf_*): same method names and state semantics.
#ifndef SLAMFUSION_LOCALIZATION_BRUTE_FORCE_ALIGNMENT_H
#define SLAMFUSION_LOCALIZATION_BRUTE_FORCE_ALIGNMENT_H

#include "localization/icp_point_to_point.h"

class BruteForceAlignment {
public:
    BruteForceAlignment() { check(sf_bf_create(slamfusion::default_context(), &bf_)); }
    ~BruteForceAlignment() { sf_bf_destroy(bf_); }
    BruteForceAlignment(const BruteForceAlignment &) = delete;
    BruteForceAlignment &operator=(const BruteForceAlignment &) = delete;

    void setXYZStep(const float x_step, const float y_step, const float z_step) { check(sf_bf_set_xyz_step(bf_, x_step, y_step, z_step)); }
    void setXYZRange(const float x, const float y, const float z) { check(sf_bf_set_xyz_range(bf_, x, y, z)); }
    void setRotationStep(const float yaw_step) { check(sf_bf_set_rotation_step(bf_, yaw_step)); }
    void setRotationRange(const float yaw) { check(sf_bf_set_rotation_range(bf_, yaw)); }
    void setMeanErrorThreshold(const float error_threshold) { check(sf_bf_set_mean_error_threshold(bf_, error_threshold)); }
    void setSourceCloud(const slamfusion::PointCloud::Ptr &cloud) { check(sf_bf_set_source(bf_, cloud->xyz.data(), (int64_t)cloud->size())); }
    void setTargetCloud(const slamfusion::PointCloud::Ptr &cloud) { check(sf_bf_set_target(bf_, cloud->xyz.data(), (int64_t)cloud->size())); }
    void setTargetMap(sf_map *map) { check(sf_bf_set_target_map(bf_, map)); }
    void setInitialGuess(const slamfusion::Matrix4f &initial_guess)
    {
        float rm[16];
        initial_guess.toRowMajor(rm);
        check(sf_bf_set_initial_guess(bf_, rm));
    }
    void resetFirstAlignment(const bool value) { check(sf_bf_reset_first_alignment(bf_, value ? 1 : 0)); }
    bool alignClouds()
    {
        int found = 0;
        check(sf_bf_align_clouds(bf_, &found));
        return found != 0;
    }
    bool firstAlignmentCompleted() const { return sf_bf_first_alignment_completed(bf_) != 0; }
    slamfusion::Matrix4f getBestTransformation() const
    {
        float rm[16];
        check(sf_bf_get_best_transformation(bf_, rm));
        return slamfusion::Matrix4f::fromRowMajor(rm);
    }

private:
    static void check(int rc)
    {
        if (rc != SF_OK) throw std::runtime_error(std::string("libslamfusion: ") + sf_last_error());
    }
    sf_bf *bf_ = nullptr;
};

#endif
