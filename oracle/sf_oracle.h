/*
 * sf_oracle.h — CPU restatement ("oracle") of the scan-to-map registration path of
 * viniciusvidal2/slam-sensor-fusion.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (libslamfusion.so) never
 * links, loads or falls back to anything under oracle/.
 *
 * PARITY UNPINNED upstream: the reference ships no tests, fixtures or golden vectors for
 * this path, and its arithmetic lives in un-vendored PCL / FLANN /
 * Eigen / Open3D (no versions pinned; ROS 2 Humble era inferred: PCL 1.12, FLANN 1.9.1,
 * Eigen 3.4, Open3D >= 0.13).  The restatement therefore follows the reference's own call
 * sites plus the published algorithm of each third-party call, and is pinned by
 *   (1) oracle/_ref/libsfref.so — the one reference file that compiles stand-alone
 *       (localization/include/localization/geo_lib.hpp), built from where it lies;
 *   (2) independent in-container tools: scipy.spatial.cKDTree (exact NN),
 *       numpy.linalg.svd (Kabsch), scipy Rotation (quaternion / euler);
 *   (3) analytic known answers (tests/test_oracle_*.py).
 *
 * Conventions: all 4x4 matrices are ROW-MAJOR arrays of 16 (Eigen in the reference is
 * column-major; values are layout independent).  Point clouds are AoS xyz triplets.
 * Built with -ffp-contract=off so that f32 expressions round exactly like the
 * reference's x86-64 build (no FMA: localization/CMakeLists.txt sets no -march).
 */
#ifndef SF_ORACLE_H
#define SF_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ kd-tree (exact) */
/* Restates pcl::KdTreeFLANN<PointXYZ> == flann::KDTreeSingleIndex<L2_Simple<float>>,
 * leaf_max_size 15, eps 0, reorder on.  Call sites: icp_point_to_point.cpp:54,68;
 * point_cloud_processing.hpp:37-45; brute_force_alignment.cpp:72-73,102.            */
typedef struct orc_kdtree_f orc_kdtree_f;
typedef struct orc_kdtree_d orc_kdtree_d;

orc_kdtree_f *orc_kdtree_f_build(const float *xyz, int n, int leaf_max);
void orc_kdtree_f_free(orc_kdtree_f *t);
int orc_kdtree_f_size(const orc_kdtree_f *t);
/* 1-NN for m queries; idx[i] = -1 and d2[i] = +inf when the tree is empty or the query
 * is not finite.  d2 is the SQUARED distance, summed x,y,z in that order (L2_Simple). */
void orc_kdtree_f_nn(const orc_kdtree_f *t, const float *q, int m, int *idx, float *d2);

orc_kdtree_d *orc_kdtree_d_build(const double *xyz, int n, int leaf_max);
void orc_kdtree_d_free(orc_kdtree_d *t);
void orc_kdtree_d_nn(const orc_kdtree_d *t, const double *q, int m, int *idx, double *d2);

/* O(n*m) brute force with the same distance expression (checker for the checker). */
void orc_bruteforce_nn_f(const float *tgt, int n, const float *q, int m, int *idx, float *d2);

/* ------------------------------------------------------------------ crops / subsample */
/* applyUniformSubsample — point_cloud_processing.hpp:55-74. Returns kept count. */
int orc_uniform_subsample(const float *xyz, int n, int step, float *out);
/* cropPointCloudThroughRadius — point_cloud_processing.hpp:31-53.  center = T[:3,3];
 * keep d2 < (float)(radius*radius); output ascending (d2, index) like a sorted FLANN
 * radius search.  out_idx may be NULL. */
int orc_crop_radius(const float *xyz, int n, const float center[3], double radius,
                    float *out, int *out_idx);
/* removeFloor — point_cloud_processing.hpp:76-92 (keep z > 0). */
int orc_remove_floor(const float *xyz, int n, float *out);
/* readFilterPtcRegionPoints — localization_node.py:105-115 (inclusive AABB, NaNs skipped). */
int orc_crop_aabb(const float *xyz, int n, const double lo[3], const double hi[3], float *out,
                  int *out_idx);
/* Open3D OrientedBoundingBox crop — localization_node.py:222-225.  R row-major 3x3 (used
 * as given, not re-orthonormalised), extent full lengths; |d.R[:,k]| <= extent[k]/2. */
int orc_crop_obb(const float *xyz, int n, const double center[3], const double R[9],
                 const double extent[3], float *out, int *out_idx);

/* ------------------------------------------------------------------ voxel grids */
/* PCL VoxelGrid (float32) — global_map_frames_manager.cpp:142-146.  Returns the number
 * of voxels written, or -1 on the int32 index overflow PCL reports (output = input then,
 * count n copied).  vox_idx (optional, size n) receives each input point's linear voxel
 * index (-1 for non-finite points); out_vox (optional) the index of each output voxel,
 * ascending.  Centroids: f32 sums in ascending point index order within the voxel. */
int orc_voxel_pcl(const float *xyz, int n, float leaf, float *out, int32_t *vox_idx,
                  int32_t *out_vox);
/* The same arithmetic with an int64 linear index: never reports overflow (extension for maps past
 * 2^31 voxels; equals orc_voxel_pcl wherever that does not overflow). */
int orc_voxel_pcl64(const float *xyz, int n, float leaf, float *out, int64_t *vox_idx, int64_t *out_vox);
/* Open3D voxel_down_sample (float64) — localization_node.py:47.  ijk (optional, n*3)
 * receives int32 voxel coordinates; output sorted by (i,j,k) lexicographic (Open3D's own
 * order is unordered_map order — compare as a set); out_ijk optional (count*3). */
int orc_voxel_o3d(const double *xyz, int n, double voxel, double *out, int32_t *ijk,
                  int32_t *out_ijk);

/* ------------------------------------------------------------------ small linear algebra */
/* SVD of a 3x3 (row-major): A = U diag(S) V^T, S descending.  One-sided Jacobi. */
void orc_svd3_d(const double A[9], double U[9], double S[3], double V[9]);
void orc_svd3_f(const float A[9], float U[9], float S[3], float V[9]);
/* Kabsch step of calculateStepBestTransformation (icp_point_to_point.cpp:112-159).
 * precise=0: float32 sequential sums exactly as written there; precise=1: float64. */
void orc_kabsch(const float *src, const float *tgt, int n, int precise, double T[16]);

/* ------------------------------------------------------------------ ICP drivers */
typedef struct {
    double T[16];      /* final transformation, row-major                          */
    double error;      /* ref_cpp: last_error_; o3d/p2plane: inlier_rmse            */
    double fitness;    /* o3d/p2plane: #corr / #source                              */
    int iterations;    /* steps applied                                             */
    int converged;     /* ref_cpp: last_error_ < acceptable; o3d: criteria met      */
    int n_corr;        /* correspondences in the last search                        */
    int n_research;    /* ref_cpp: lazy re-searches taken                           */
} orc_icp_result;

/* ICPPointToPoint::calculateAlignment — icp_point_to_point.cpp:185-254, with the
 * squared-vs-unsquared threshold of :70 reproduced.  precise=0 mirrors the reference's
 * float32 arithmetic statement by statement; precise=1 runs the same control flow in
 * float64 (ground truth for the 1e-4 m / 1e-5 rad criterion). */
int orc_icp_ref_cpp(const float *src, int n, const float *tgt, int m, const float init[16],
                    float max_corr_dist, int num_iters, float accept_err, float eps,
                    int precise, orc_icp_result *out);

/* Open3D registration_icp, TransformationEstimationPointToPoint, default relative
 * criteria 1e-6 — localization_node.py:233-237.  float64 throughout.  */
int orc_icp_o3d_p2p(const float *src, int n, const float *tgt, int m, const double init[16],
                    double max_dist, int max_iter, orc_icp_result *out);

/* Extension (no reference code): point-to-plane Gauss-Newton, NN every
 * iteration, exactly num_iters iterations, true-distance threshold. */
int orc_icp_p2plane(const float *src, int n, const float *tgt, const float *tgt_normals,
                    int m, const double init[16], double max_dist, int num_iters,
                    orc_icp_result *out);

/* Extension: PCA normals from all neighbours within `radius`
 * (self included); < 3 neighbours => (0,0,1).  Sign: n.z >= 0 (then n.y, n.x). */
void orc_normals_radius(const float *xyz, int n, double radius, float *normals,
                        int *n_neighbors);
/* the same plus each point's neighbourhood covariance (xx xy xz yy yz zz, centred, divided by the
 * neighbour count; zeros below 3 neighbours) -- SURVEY x2 "+6 cov", BASELINE config 5 */
void orc_normals_radius_cov(const float *xyz, int n, double radius, float *normals,
                            int *n_neighbors, double *cov6);

/* ------------------------------------------------------------------ pose fusion (host) */
/* UTM::LLtoUTM — geo_lib.hpp:38-83 (always +10 000 000 N). */
void orc_ll_to_utm(double lat, double lon, double *northing, double *easting);
/* utm.from_latlon as called at localization_node.py:138 (third-party `utm`, unpinned). */
void orc_utm_from_latlon(double lat, double lon, double *easting, double *northing);
/* Eigen::Quaternionf::toRotationMatrix as used at localization_node.cpp:94-103. */
void orc_quat_to_pose(const double q_wxyz[4], const double t[3], float T[16]);
void orc_mat4f_inverse(const float A[16], float out[16]);
void orc_mat4f_mul(const float A[16], const float B[16], float out[16]);
/* computePosePredictionFromOdometry — localization_node.cpp:89-110. */
void orc_odom_prediction(const float map_T_sensor[16], const float odom_T_prev[16],
                         const float odom_T_cur[16], float out[16]);
/* compass callback — localization_node.cpp:64-76. */
float orc_compass_to_yaw(double compass_deg);
/* getClosestAltitude — global_map_frames_manager.cpp:69-91; table = rows of lat,lon,alt */
float orc_closest_altitude(const double *table, int rows, double lat, double lon);
/* computeGpsCoarsePoseInMapFrame — localization_node.cpp:112-128. */
void orc_gps_pose(const double map_T_global[16], float yaw, double lat, double lon,
                  float table_alt, float out[16]);
/* computePoseGainsFromCovarianceMatrices — localization_node.cpp:151-179. */
void orc_pose_gains(const double gps_cov[9], const double odom_cov[36], int fixed,
                    float *odom_gain, float *gps_gain);
/* prior blend — localization_node.cpp:329. */
void orc_blend(float g_odom, const float T_odom[16], float g_gps, const float T_gps[16],
               float out[16]);
/* computeMapTGlobal — global_map_frames_manager.cpp:209-248. */
void orc_map_T_global(const double *latlonalt, const float *yaw, int n, double out[16]);

/* StochasticFilter — stochastic_filter.cpp (all). */
typedef struct orc_sfilter orc_sfilter;
orc_sfilter *orc_sfilter_new(int queue_size, float z_threshold);
void orc_sfilter_free(orc_sfilter *f);
void orc_sfilter_weights(const orc_sfilter *f, float *w);
void orc_sfilter_add_pose(orc_sfilter *f, const float pose[16]);
float orc_sfilter_zscore(const orc_sfilter *f, const float prev[16], const float cur[16]);
void orc_sfilter_apply(const orc_sfilter *f, const float prev[16], const float cur[16],
                       float out[16]);

/* BruteForceAlignment candidate order — brute_force_alignment.cpp:148-180.  Writes up to
 * cap values, returns count. */
int orc_bf_sequence(float range, float step, float *seq, int cap);

/* BruteForceAlignment::alignClouds — brute_force_alignment.cpp:65-136.  prev_T (row-major)
 * is map_T_sensor_previous_ on entry and is updated like the reference on a miss (:123).
 * Returns 1 when a candidate's mean squared NN distance fell below `threshold`.
 * best_T: the transformation getBestTransformation() would return afterwards;
 * index: position of the chosen candidate in x,y,z,yaw nesting order; scores (optional,
 * n_candidates floats): the float32 score of every candidate that was evaluated (early
 * exit leaves the rest untouched). */
typedef struct {
    float x_step, y_step, z_step, yaw_step, x_range, y_range, z_range, yaw_range, threshold;
} orc_bf_params;
int orc_bf_align(const float *src, int n, const float *tgt, int m, float prev_T[16],
                 const orc_bf_params *prm, float best_T[16], float *best_score, int *index,
                 int *n_candidates, float *scores);

#ifdef __cplusplus
}
#endif
#endif
