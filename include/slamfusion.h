/*
 * slamfusion.h — C ABI of libslamfusion.so: MI355X-native (gfx950, hand-written HIP)
 * scan-to-map registration hot path, drop-in for that path of
 * viniciusvidal2/slam-sensor-fusion.  Plain C types only; no torch, PCL or Eigen types.
 *
 * The reference has no FFI/plugin interface: the seam is the C++ class
 * ICPPointToPoint (localization/include/localization/icp_point_to_point.h:41-136), the
 * free functions of localization/include/localization/point_cloud_processing.hpp:31-92,
 * the PCL VoxelGrid call (localization/src/global_map_frames_manager.cpp:142-146) and,
 * in the Python twin, three Open3D calls (localization_python/localization_python/
 * localization_node.py:47,222-225,233-237).  Each entry point below cites what it
 * replaces.  INTEGRATION.md shows the reference-side bindings.
 *
 * Conventions
 *   - 4x4 matrices: ROW-MAJOR arrays of 16 (Eigen::Matrix4f in the reference is
 *     column-major: pass M.transpose().data() or use include/localization adapters).
 *   - Point clouds: AoS xyz float32 triplets (pcl::PointXYZ minus its padding).
 *   - Every call returns sf_status (0 = ok, < 0 = infrastructure error; text from
 *     sf_last_error()).  Algorithmic "no lock" is NOT an error: it is reported exactly
 *     like the reference, in sf_icp_result (initial transform, error 1e6, iterations 0,
 *     converged 0 — icp_point_to_point.cpp:196-200, icp_point_to_point.h:28-39).
 *   - Not thread-safe: one sf_ctx per host thread (the reference is single-threaded,
 *     localization/src/main.cpp:18).  One HIP stream per context; calls that return
 *     data to the host synchronise that stream, all others only enqueue.
 *   - There is NO CPU fallback: without a HIP device sf_ctx_create fails.
 */
#ifndef SLAMFUSION_H
#define SLAMFUSION_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_VERSION 210

typedef enum {
    SF_OK = 0,
    SF_ERR_INVALID = -1,  /* bad argument                                    */
    SF_ERR_HIP = -2,      /* HIP runtime failure (message has the hipError)   */
    SF_ERR_NOMEM = -3,
    SF_ERR_STATE = -4,    /* call order (e.g. align before set_source)        */
    SF_ERR_OVERFLOW = -5, /* index space too large for the requested grid     */
    SF_ERR_COMM = -6      /* multi-GPU: a collective timed out or was aborted by a peer; every rank of the communicator gets it */
} sf_status;

typedef struct sf_ctx sf_ctx;
typedef struct sf_cloud sf_cloud;
typedef struct sf_map sf_map;
typedef struct sf_icp sf_icp;

int sf_version(void);
const char *sf_last_error(void);

/* ------------------------------------------------------------------ context */
/* One context = one device + one stream.  stream == NULL creates a private stream;
 * otherwise the caller's hipStream_t is used (e.g. a torch ExternalStream) and not owned. */
int sf_ctx_create(int device_id, void *hip_stream, sf_ctx **out);
void sf_ctx_destroy(sf_ctx *ctx);
void *sf_ctx_stream(sf_ctx *ctx);
int sf_ctx_synchronize(sf_ctx *ctx);
int sf_ctx_device_name(sf_ctx *ctx, char *buf, int cap);

/* ------------------------------------------------------------------ clouds (device point sets) */
/* a7: setSourcePointCloud / convertPclToEigen upload — icp_point_to_point.cpp:44-47,86-97 */
int sf_cloud_create(sf_ctx *ctx, sf_cloud **out);
void sf_cloud_destroy(sf_cloud *c);
int sf_cloud_upload(sf_cloud *c, const float *xyz, int64_t n);
int sf_cloud_upload_f64(sf_cloud *c, const double *xyz, int64_t n); /* rounds to f32 */
/* enqueue only (no host synchronisation): with PINNED host memory the copy is asynchronous and stream-ordered -- e.g. on a
 * copy stream's context into a staging cloud that a compute stream picks up with sf_icp_set_source_batch_device */
int sf_cloud_upload_async(sf_cloud *c, const float *xyz, int64_t n);
void *sf_cloud_device_ptr(sf_cloud *c); /* float[n][3] on the device (valid until the cloud is modified) */
int sf_cloud_from_device(sf_cloud *c, const void *d_xyz, int64_t n); /* device-to-device copy */
int sf_cloud_size(sf_cloud *c, int64_t *n);
int sf_cloud_download(sf_cloud *c, float *xyz, int64_t cap, int64_t *n);
int sf_cloud_copy(sf_cloud *dst, sf_cloud *src);
/* a1: applyUniformSubsample — point_cloud_processing.hpp:55-74 (no-op when n < step) */
int sf_cloud_subsample(sf_cloud *c, int step);
/* a2: cropPointCloudThroughRadius — point_cloud_processing.hpp:31-53.  center = T[:3,3];
 * keeps d2 < (float)(radius*radius).  sorted != 0 reproduces PCL's ascending
 * (distance, index) output order; sorted == 0 keeps index order (faster). */
int sf_cloud_crop_radius(sf_cloud *c, const float center[3], double radius, int sorted);
/* a3: removeFloor — point_cloud_processing.hpp:76-92 (keep z > 0) */
int sf_cloud_remove_floor(sf_cloud *c);
/* a19: readFilterPtcRegionPoints — localization_node.py:105-115 (inclusive AABB, NaN skipped) */
int sf_cloud_crop_aabb(sf_cloud *c, const double lo[3], const double hi[3]);
/* a20: OrientedBoundingBox crop — localization_node.py:222-225; R row-major 3x3 */
int sf_cloud_crop_obb(sf_cloud *c, const double center[3], const double R[9], const double extent[3]);
/* a8: applyTransformation — icp_point_to_point.cpp:99-110 (general affine 3x4, float32,
 * unfused multiply/add so results are bit-identical to the reference's x86-64 build) */
int sf_cloud_transform(sf_cloud *c, const float T[16]);
/* `*map_cloud += *cloud` — global_map_frames_manager.cpp:131: dst <- [dst; src] on the device (src unchanged).
 * Map growth = transform the registered scan into the map frame, append, voxel-downsample, sf_map_build. */
int sf_cloud_append(sf_cloud *dst, const sf_cloud *src);
/* indices (into the cloud before the call) kept by the LAST crop/subsample on this cloud */
int sf_cloud_last_indices(sf_cloud *c, int32_t *idx, int64_t cap, int64_t *n);

/* f-2: PointCloud2 -> xyz on the device (pcl::fromROSMsg, localization_node.cpp:290-291;
 * pc2.read_points, localization_node.py:106-111): raw little-endian message buffer,
 * n_points = width*height, float32 fields at byte offsets off_x/y/z inside point_step */
int sf_cloud_from_pointcloud2(sf_cloud *c, const void *data, int64_t n_points, int point_step, int off_x, int off_y, int off_z);
/* the whole sensor_msgs/PointCloud2 contract, checked: data_bytes must cover height rows of row_step bytes
 * (row_step = 0: width * point_step), x/y/z datatype 7 (FLOAT32) or 8 (FLOAT64, rounded to float32 like
 * pcl::fromROSMsg's field mapping), big-endian payloads are refused (SF_ERR_INVALID).  No allocation and no host
 * synchronisation per call once the staging buffer has grown to the message size. */
#define SF_PC2_FLOAT32 7
#define SF_PC2_FLOAT64 8
int sf_cloud_from_pointcloud2_msg(sf_cloud *c, const void *data, int64_t data_bytes, int64_t width, int64_t height, int point_step, int64_t row_step,
                                  int off_x, int off_y, int off_z, int datatype, int is_bigendian);
/* f-3: PCD v0.7 files (ascii / binary / binary_compressed read; "DATA binary" write exactly
 * as pcl::io::savePCDFileBinary does for PointXYZ — mapping/src/map_data_save_node.cpp:74) */
int sf_cloud_load_pcd(sf_cloud *c, const char *path);
int sf_cloud_save_pcd(sf_cloud *c, const char *path);
int sf_pcd_read(const char *path, float **xyz, int64_t *n); /* caller frees with sf_free */
int sf_pcd_write_binary(const char *path, const float *xyz, int64_t n);
void sf_free(void *p);

/* a4 / a5: voxel grids.  flavour SF_VOXEL_PCL = pcl::VoxelGrid float32
 * (global_map_frames_manager.cpp:142-146): int32 linear index, ascending-index output,
 * on int32 overflow the cloud is left unchanged and *status_flags gets
 * SF_FLAG_VOXEL_OVERFLOW (PCL warns and returns its input).  SF_VOXEL_O3D = Open3D
 * voxel_down_sample float64 (localization_node.py:47): origin min_bound - v/2, output
 * ordered by (i,j,k). */
#define SF_VOXEL_PCL 0
#define SF_VOXEL_O3D 1
/* EXTENSION (no reference counterpart): pcl::VoxelGrid's arithmetic with the linear index kept in 64 bits, for maps
 * past 2^31 voxels (BASELINE config 5: a 50 M-point city at leaf 0.1 m) where PCL -- and SF_VOXEL_PCL -- return the
 * cloud unfiltered.  Wherever SF_VOXEL_PCL does not overflow the two give identical ids, order and centroids. */
#define SF_VOXEL_PCL64 2
#define SF_FLAG_VOXEL_OVERFLOW 1
int sf_cloud_voxel_downsample(sf_cloud *c, double leaf, int flavour, int *status_flags);
/* Incremental map growth: exactly sf_cloud_append(map, pending) + sf_cloud_voxel_downsample(map, leaf, SF_VOXEL_PCL, ...)
 * (`*map_cloud += *cloud` + the voxel filter, global_map_frames_manager.cpp:131,142-146) -- bit for bit -- computed as a MERGE
 * when `map` is itself voxel-filtered at that leaf: only the pending points are sorted, their voxels found among the map's by
 * binary search, touched centroids re-summed in point order, one copy pass opens the gaps for the new voxels.  Falls back to
 * the two calls whenever the preconditions do not hold (map keys not strictly ascending under the union's geometry, index
 * overflow, an empty side); *merged (may be NULL) says which way it went.  Both clouds on the same context. */
int sf_cloud_voxel_merge(sf_cloud *map, sf_cloud *pending, double leaf, int *status_flags, int *merged);
/* maps smaller than this take the full path inside sf_cloud_voxel_merge (default 4 000 000 points: below, the full filter is as fast);
 * returns the previous value, n < 0 only reads it */
int sf_cloud_voxel_merge_min_points(int64_t n);
/* introspection for parity tests: per-input-point voxel ids of the LAST downsample
 * (PCL: int32 linear index, -1 for non-finite; O3D: int32 i,j,k triplets) and the ids /
 * float64 means of the output voxels (O3D flavour keeps float64 means). */
int sf_cloud_voxel_point_ids(sf_cloud *c, int32_t *ids, int64_t cap_values, int64_t *n_values);
int sf_cloud_voxel_out_ids(sf_cloud *c, int32_t *ids, int64_t cap_values, int64_t *n_values);
int sf_cloud_voxel_out_means_f64(sf_cloud *c, double *xyz, int64_t cap_points, int64_t *n_points);
int sf_cloud_voxel_point_ids64(sf_cloud *c, int64_t *ids, int64_t cap_values, int64_t *n_values); /* after SF_VOXEL_PCL64 */
int sf_cloud_voxel_out_ids64(sf_cloud *c, int64_t *ids, int64_t cap_values, int64_t *n_values);

/* ------------------------------------------------------------------ map (NN index) */
/* a6: setTargetPointCloud — icp_point_to_point.cpp:49-55.  Instead of a FLANN kd-tree
 * over a 10 m crop rebuilt every 3 m, the WHOLE map gets one device-resident uniform-grid
 * index (cell size `cell`, 0 = automatic); the reference's crop becomes a window. */
int sf_map_create(sf_ctx *ctx, sf_map **out);
void sf_map_destroy(sf_map *m);
int sf_map_build(sf_map *m, sf_cloud *cloud, float cell);
int sf_map_size(sf_map *m, int64_t *n);
int sf_map_cell_size(sf_map *m, float *cell, int32_t dims[3]);
/* The index carried over a growth step (`*map_cloud += *cloud` + voxel filter + setTargetPointCloud:
 * global_map_frames_manager.cpp:131,142-146, icp_point_to_point.cpp:49-55): exactly sf_map_build(m, cloud, <the cell m has>)
 * -- points in cell order with their ids, and the cell table, bit for bit -- computed as a MERGE of the old index with the centroids the last
 * sf_cloud_voxel_merge(cloud, ...) wrote, when `m` indexes `cloud` as it was before that merge and nothing has touched
 * the cloud since: the entries that stay keep their order (one streaming pass), only the new centroids are sorted.
 * Takes the build itself whenever that does not hold (the merge took its full path, the cloud changed in between, the
 * smallest coordinate of the map moved, 64-bit cell ids); *patched (may be NULL) says which way it went: 1 = merged,
 * SF_PATCH_* (<= 0) = built, and why.  Normals and the window are dropped as sf_map_build drops them; re-attach the map
 * to its sf_icp (sf_icp_set_target). */
#define SF_PATCH_NO_MERGE        0   /* no usable record: the merge took its full path, the cloud changed since, another cloud's index */
#define SF_PATCH_BOUND_REPLACED (-1) /* (not returned any more: a replaced bound costs one reduction over the cloud, then the patch goes on) */
#define SF_PATCH_ORIGIN_MOVED   (-2) /* a new centroid lies below the grid origin: every cell changes */
#define SF_PATCH_LIMITS         (-3) /* 2^28 points / 2^32 cells / table memory */
#define SF_PATCH_CLAMPED_POINT  (-4) /* a point the old grid had clamped to its upper face now lies in a cell of its own */
int sf_map_patch(sf_map *m, sf_cloud *cloud, int *patched);
/* For a map that is going to grow (sf_map_patch): the grid starts at the multiple of `cells` cells (0 = off, the default: at the
 * smallest coordinates themselves) below the smallest coordinates, so growth by a few metres in any direction -- below the
 * origin too -- leaves every point's cell coordinates, and a patch possible; the table gets at most `cells` empty cells per
 * axis in front of the data.  Takes effect with the next sf_map_build.  Maps that register large BATCHES are better left
 * without: the scans' ordering keys span the empty margin as well (measured: -4 % at the metric configuration with 64). */
int sf_map_set_origin_lattice(sf_map *m, int cells);
/* the index as it lies in HBM, for parity tests: points indexed, cells, grid origin, 1 / cell, the pruning slack;
 * pts4 = float[n][4] (x, y, z, bitcast point id) in cell order, cell_start = uint32[n_cells + 1] (either may be NULL) */
int sf_map_index_info(sf_map *m, int64_t *n_indexed, int64_t *n_cells, float org[3], float *inv_h, float *gap_eps);
int sf_map_download_index(sf_map *m, float *pts4, int64_t cap_points, uint32_t *cell_start, int64_t cap_cells);
/* window = the reference's map crop, applied as a predicate inside the search:
 * sphere: d2(center,p) < (float)(radius*radius) like cropPointCloudThroughRadius
 * (localization_node.cpp:302); obb: like localization_node.py:222-225; none: whole map */
int sf_map_window_none(sf_map *m);
int sf_map_window_sphere(sf_map *m, const float center[3], double radius);
int sf_map_window_obb(sf_map *m, const double center[3], const double R[9], const double extent[3]);
/* number of indexed map points inside the current window (the reference skips the scan when
 * its cropped map is empty, localization_node.py:226-228) */
int sf_map_window_count(sf_map *m, int64_t *n);
/* extension x2 (no reference code): PCA normals from neighbours within `radius` */
int sf_map_estimate_normals(sf_map *m, float radius);
/* the same pass also keeps each point's 3x3 neighbourhood covariance (6 unique float64 entries xx xy xz yy yz zz,
 * centred on the neighbourhood mean, divided by the neighbour count; zeros below 3 neighbours) -- SURVEY x2
 * "normals + covariance", BASELINE config 5 */
int sf_map_estimate_normals_cov(sf_map *m, float radius, int with_covariance);
int sf_map_download_covariances(sf_map *m, double *cov6, int64_t cap_points, int64_t *n); /* original point order */
int sf_map_set_normals(sf_map *m, const float *normals, int64_t n); /* original point order */
int sf_map_download_normals(sf_map *m, float *normals, int32_t *n_neighbors, int64_t cap, int64_t *n);
/* raw exact 1-NN (a9 without the threshold): idx in ORIGINAL point order, d2 squared
 * float32 summed x,y,z like FLANN L2_Simple; idx = -1 if nothing within max_d2. */
int sf_map_nn(sf_map *m, const float *queries, int64_t n, float max_d2, int32_t *idx, float *d2);

/* ------------------------------------------------------------------ ICP */
/* a13: ICPResult — icp_point_to_point.h:28-39 (+ float64 and diagnostics) */
typedef struct {
    float T[16];        /* transformation (row-major)                               */
    float error;        /* ref_cpp: last_error_ ; o3d/p2plane: inlier_rmse           */
    int32_t iterations; /* steps applied                                             */
    int32_t converged;  /* has_converged                                             */
    int32_t n_corr;     /* correspondences of the last search                        */
    int32_t n_research; /* correspondence searches performed                         */
    int32_t flags;      /* SF_ICP_FLAG_*                                             */
    double fitness;     /* o3d/p2plane: n_corr / n_source                            */
    double rmse;        /* o3d/p2plane                                               */
    double T64[16];     /* same transformation before rounding to float             */
} sf_icp_result;
#define SF_ICP_FLAG_FEW_CORR 1   /* < 10 correspondences: reference returns the initial T */
#define SF_ICP_FLAG_SINGULAR 2   /* p2plane normal equations not positive definite       */
#define SF_ICP_FLAG_BARRIER_TIMEOUT 8 /* internal: a grid barrier of the single-launch REF_CPP form gave up; sf_icp_fetch_results turns it into SF_ERR_HIP */
#define SF_ICP_FLAG_SHARD_STALE 4 /* sharded stepping: the scan moved out of the margin of this rank's owned-query arrays and stopped; resume with sf_icp_step_begin(first = 2) */

/* modes: REF_CPP = ICPPointToPoint::calculateAlignment (icp_point_to_point.cpp:185-254,
 * lazy re-search, d2 < max_correspondence_dist quirk of :70);  O3D_P2P = Open3D
 * registration_icp point-to-point as called at localization_node.py:233-237;
 * P2PLANE = extension x1, Gauss-Newton, exactly num_iterations iterations. */
#define SF_ICP_REF_CPP 0
#define SF_ICP_O3D_P2P 1
#define SF_ICP_P2PLANE 2

/* ctor — icp_point_to_point.h:49 / icp_point_to_point.cpp:3-12 */
int sf_icp_create(sf_ctx *ctx, float max_correspondence_dist, int num_iterations,
                  float acceptable_mean_error, float transformation_epsilon, sf_icp **out);
void sf_icp_destroy(sf_icp *icp);
/* setters — icp_point_to_point.cpp:14-42 */
int sf_icp_set_max_correspondence_dist(sf_icp *icp, float v);
int sf_icp_set_num_iterations(sf_icp *icp, int v);
int sf_icp_set_transformation_epsilon(sf_icp *icp, float v);
int sf_icp_set_acceptable_mean_error(sf_icp *icp, float v);
int sf_icp_set_initial_transformation(sf_icp *icp, const float T[16]);
int sf_icp_set_initial_transformation_f64(sf_icp *icp, const double T[16]);
int sf_icp_set_debug_mode(sf_icp *icp, int on);
/* setSourcePointCloud — icp_point_to_point.cpp:44-47 (copies; caller keeps its cloud) */
int sf_icp_set_source(sf_icp *icp, const float *xyz, int64_t n);
int sf_icp_set_source_cloud(sf_icp *icp, sf_cloud *cloud);
/* The node's scan preprocessing and setSourcePointCloud in one pass, without a host synchronisation
 * (localization_node.cpp:290-297,333): the source becomes every stride-th point of `raw` that is finite and within `radius`
 * of `center`, in index order -- exactly sf_cloud_subsample(stride) + sf_cloud_crop_radius(center, radius, 0) +
 * sf_icp_set_source_cloud, same predicates, same points.  `raw` is left untouched; the count stays on the device, so the
 * source serves unsharded single-scan REF_CPP alignments only.  sf_icp_source_count: the number of points the last fetched
 * single-scan REF_CPP alignment ran on. */
int sf_icp_set_source_scan(sf_icp *icp, sf_cloud *raw, int stride, const float center[3], double radius);
int sf_icp_source_count(sf_icp *icp, int64_t *n);
/* setTargetPointCloud — icp_point_to_point.cpp:49-55: either index a cloud privately
 * (reference semantics) or share a prebuilt whole-map index (no copy, no rebuild). */
int sf_icp_set_target(sf_icp *icp, const float *xyz, int64_t n);
int sf_icp_set_target_map(sf_icp *icp, sf_map *map);
/* calculateAlignment — icp_point_to_point.cpp:185-254 */
int sf_icp_align(sf_icp *icp, int mode, sf_icp_result *out);

/* batched throughput form: `batch` scans of n_per_scan points each (xyz contiguous,
 * scan-major), one initial transform per scan (inits = batch*16 doubles, NULL = identity),
 * all registered concurrently against the same target map in ONE kernel sequence. */
int sf_icp_set_source_batch(sf_icp *icp, const float *xyz, int64_t n_per_scan, int batch);
int sf_icp_set_source_batch_device(sf_icp *icp, const void *d_xyz, int64_t n_per_scan, int batch);
int sf_icp_set_initial_batch_f64(sf_icp *icp, const double *inits);
int sf_icp_align_batch(sf_icp *icp, int mode, sf_icp_result *out /* batch entries */);
/* enqueue only (no host sync); results fetched later.  sf_icp_fetch_results returns the LATEST enqueued alignment as it
 * was enqueued (`out`: as many entries as ITS batch) -- a source or priors set since belong to the next alignment. */
int sf_icp_align_batch_async(sf_icp *icp, int mode);
int sf_icp_fetch_results(sf_icp *icp, sf_icp_result *out);
/* The results of the alignment enqueued BEFORE the latest one, when the two ran side by side (two
 * sf_icp_align_batch_async calls with no fetch between them, sf_icp_set_pipeline on, launch list): waits for that
 * alignment only, the latest one goes on running.  The streaming loop is "set the next batch, enqueue its alignment,
 * fetch the previous one's result": every result reaches the host and the device never drains.  `out`: as many
 * entries as THAT alignment's batch.  SF_ERR_STATE when there is nothing to fetch: the latest alignment did not run
 * beside its predecessor (a fetch in between, pipeline off, a single-launch or profiled alignment) -- then the
 * predecessor's states have been overwritten -- or the call was made already.  No reference counterpart (one scan per
 * callback, localization_node.cpp:337). */
int sf_icp_fetch_previous(sf_icp *icp, sf_icp_result *out);
int sf_icp_use_graph(sf_icp *icp, int on); /* replay the launch sequence as a hipGraph */
/* how often the launch sequence was captured / launched as a graph since creation (a capture per alignment means the
 * cache key keeps changing).  REF_CPP with ONE scan reads the scan's point count and the map window from device memory,
 * so scans of similar size (same count rounded up to 4096) and a moving window replay the same graph. */
int sf_icp_graph_counts(sf_icp *icp, int64_t *captures, int64_t *launches);
/* Alignments small enough for every workgroup to be resident at once (the nodes' per-scan paths: ~13 k points; on MI355X
 * up to 65 k points in REF_CPP and O3D_P2P, 131 k in P2PLANE, summed over the batch -- from the occupancy query, minus one
 * workgroup per CU) run as ONE launch -- search, record,
 * controller / solve and step behind in-kernel grid barriers (REF_CPP: icp_point_to_point.cpp:185-254), bit-identical to
 * the launch list; on by default, larger alignments, sharded and profiled ones take the launch list.
 * sf_icp_fused_count: how many alignments have taken the single-launch form since creation. */
int sf_icp_set_fused(sf_icp *icp, int on);
int sf_icp_fused_count(sf_icp *icp, int64_t *launches);
/* Residency of a single-launch grid is kept by a per-device ledger of the grids in flight (all contexts of the process);
 * an alignment that does not fit takes the launch list.  Should a grid barrier still give up (another PROCESS holding the
 * compute units), the alignment is redone through the launch list inside sf_icp_fetch_results and the object stays on
 * the launch list: sf_icp_fused_redone counts how often that happened. */
int sf_icp_fused_redone(sf_icp *icp, int64_t *redone);
int sf_icp_test_inject_barrier_timeout(sf_icp *icp); /* test hook: treat the next single-launch alignment as timed out (exercises the redo) */
/* Order in which the points of a scan are walked by O3D_P2P / P2PLANE.  The correspondences and
 * every per-point term are independent of it; only the rounding of the record sums changes
 * (deterministically for a given order).  CELL sorts each scan by the map-grid cell of its points
 * under the initial pose at the start of every alignment (on the device, inside sf_icp_align*), so
 * that scans in flight share map lines in L2; AUTO = CELL when scans x points >= 300 000. */
#define SF_ORDER_AUTO 0
#define SF_ORDER_AS_GIVEN 1
#define SF_ORDER_CELL 2
int sf_icp_set_query_order(sf_icp *icp, int order);
/* Neighbour reuse (default on, O3D_P2P / P2PLANE).  Every full search also proves a lower bound on
 * the distance from the query to every map point other than its neighbour; in a later iteration a
 * query that has moved by less than the slack of that bound provably keeps its neighbour, and its
 * search is skipped.  Results are bit-identical with the switch on or off (tested); it only
 * changes how much of the "nearest neighbour every iteration" work has to be redone. */
int sf_icp_set_nn_reuse(sf_icp *icp, int on);
/* Scans above `points` points (at most 131 072: what the single-launch kernels can take) are registered through the launch list with
 * two queries per lane -- the form the frozen pairs need.  Without a call the rule is: above 131 072 points always; above 65 536 when
 * the batch has more rows than any single-launch kernel keeps resident (so no alignment of that shape could run as one launch:
 * nothing to stay bit-compatible with) -- a full 64-ring scan (<= 130 048 returns) registered in a batch is wide and may freeze,
 * registered alone it is not.  A call makes `points` the whole rule.  The limit is part of the summation order: results on either
 * side of it agree to float64 rounding, not bitwise. */
int sf_icp_set_wide_scan_points(sf_icp *icp, int64_t points);
/* Consecutive sf_icp_align_batch_async calls on unchanged inputs (source, initial poses, target) overlap: a launch list
 * enqueued while an earlier alignment of this object has not been fetched yet runs on an internal stream with its own copy of
 * everything an alignment writes, so an alignment's last launches (with frozen pairs: a chain of 16 us kernels on an idle
 * device) run under the next one's searching launches.  The context's stream waits for every such alignment as soon as it is
 * enqueued: a fetch, an upload or a map rebuild issued afterwards is ordered behind it exactly as before, and a changed input
 * makes the next alignment wait for everything before it.  A caller that fetches each result before the next alignment never
 * leaves the context's stream.  Results are those of the same alignment run alone (same kernels, same data; tested bitwise).
 * A source set FROM HOST MEMORY (sf_icp_set_source_batch / sf_icp_set_source) while an alignment is unfetched does not wait
 * for it either: the object keeps two source sets, the upload and its conversion take the one the alignment in flight does
 * not read, on the stream the next alignment will run on -- the streaming loop "set the next batch, enqueue its alignment,
 * sf_icp_fetch_previous" uploads batch k+1 beside the alignment of batch k (measured: +41 % over the same loop with the
 * upload behind the alignment; profiles/LADDER.md).  Sources handed over as DEVICE pointers or clouds stay on the context's
 * stream, behind whatever produced them.
 * on: 1 (default) / 0 = every alignment on the context's stream.  No reference counterpart (one scan at a time,
 * localization_node.cpp:337). */
int sf_icp_set_pipeline(sf_icp *icp, int on);
/* Tile search (O3D_P2P / P2PLANE launch list, scans above 131 072 points, whole map, unsharded, queries ordered): the
 * launches of an alignment in which nearly every query searches -- the first four with the neighbour reuse on, all of them
 * with it off -- run tile by tile: the scans' queries are sorted by map tile (a box of grid cells), one workgroup stages a
 * tile's points and cell table in LDS and serves every query of every scan that falls into it from there; the pairs go
 * through the neighbour cache and are summed in the usual order.  The exact search of icp_point_to_point.cpp:64-69 /
 * localization_node.py:233-237, the same pairs and bit-identical sums as without (tested); what changes is that a
 * candidate is an LDS read instead of a 16-byte access through the L1.  mode: 0 never, 1 (default) when the batch holds at
 * least 2 M queries, 2 whenever the conditions above hold. */
int sf_icp_set_tile_search(sf_icp *icp, int mode);
/* out[0] = the last alignment used tile search; [1..3] cells per tile (x, y, z); [4..6] tiles per axis; with
 * sf_icp_profile_enable: [7] queries that searched in the tile launches, [8] settled out of LDS, [9] walked the global
 * index because they had left the staged region, [10] went on to ring 2 and beyond, [11] tiles too dense to stage */
int sf_icp_tile_info(sf_icp *icp, int64_t out[12]);
/* Frozen pairs (P2PLANE launch list, scans above 131 072 points, neighbour reuse on, whole map, unsharded): once a scan's
 * pairs are certified to stay as they are while it moves by a guard distance more, one launch forms the 96 moments of
 * those pairs and the iterations after it evaluate the normal equations from the moments (a polynomial in the pose)
 * instead of streaming the scan; queries too close to a change stay "active" and are evaluated launch by launch.
 * Same pairs, same float64 sums up to summation rounding (~1e-13 of a pose).  on: 0 never, 1 (default) when the batch holds
 * at least 0.7 M queries -- a frozen launch costs a fixed latency whatever the batch, a small batch is faster
 * without --, 2 always.  No reference counterpart (the reference searches every point in every iteration,
 * icp_point_to_point.cpp:64-69). */
int sf_icp_set_freeze(sf_icp *icp, int on);
/* guard = max(guard_scale x motion of the last pose update, guard_min) [m]; a freeze launch is asked for once that is at most
 * guard_max (a larger guard means long active lists), at most max_tries times per alignment, from launch index
 * from_launch (>= 4) on.  Defaults 8, 2e-5, 3e-4, 3, 5. */
int sf_icp_set_freeze_params(sf_icp *icp, float guard_scale, float guard_min, float guard_max, int max_tries, int from_launch);
/* of the last batched alignment, summed over its scans: {freeze launches that held, thaws (moved beyond the guard),
 * freeze launches that did not hold, active queries of the last freeze launch, scans frozen at the end} */
int sf_icp_freeze_stats(sf_icp *icp, int64_t out[5]);

/* multi-GPU (map tile-sharded along x with halo): this rank only
 * accumulates queries whose TRANSFORMED x lies in [x_lo, x_hi); per iteration
 *   sf_icp_step_begin  -> NN + partial normal equations into the exchange buffer
 *   (caller all-reduces sf_icp_exchange_ptr over RCCL on sf_ctx_stream)
 *   sf_icp_step_end    -> identical solve on every rank
 * first = 1 starts an alignment (state <- initial transforms; sharded: this rank's owned-query
 * candidates are compacted, cell-ordered and gathered -- one host synchronisation), first = 0 is
 * the next iteration.  Sharded: a scan whose points move close to the margin (1 m) of those arrays
 * stops with SF_ICP_FLAG_SHARD_STALE (same decision on every rank); after the loop the caller
 * fetches the results and, if any scan carries the flag, calls sf_icp_step_begin(first = 2) --
 * rebuild at the current poses, flag cleared -- and runs the remaining iterations. */
int sf_icp_set_shard(sf_icp *icp, float x_lo, float x_hi);
int sf_icp_set_shard_margin(sf_icp *icp, float margin_m); /* default 1 m: how far a scan may move before its owned-query arrays are rebuilt */
int sf_icp_set_exchange_buffer(sf_icp *icp, void *d_buf, int64_t nbytes); /* optional: caller-owned (torch tensor) */
void *sf_icp_exchange_ptr(sf_icp *icp, int64_t *nbytes);
int sf_icp_step_begin(sf_icp *icp, int mode, int first);
int sf_icp_step_end(sf_icp *icp, int mode, int last);

/* The same loop driven from the C side (no host work between iterations).  sf_comm is an RCCL communicator bound to a
 * context's stream; RCCL is resolved at run time (sf_comm_load_rccl: path of the library the process must share, NULL =
 * the already loaded one or librccl.so.1).  Rank 0 draws sf_comm_unique_id (128 bytes), the launcher (torch.distributed,
 * MPI, a file) hands it to every rank, each rank calls sf_comm_create. */
typedef struct sf_comm sf_comm;
int sf_comm_load_rccl(const char *library_path);
int sf_comm_unique_id(void *id128);
int sf_comm_create(sf_ctx *ctx, int nranks, int rank, const void *id128, sf_comm **out);
void sf_comm_destroy(sf_comm *c);
int sf_comm_size(const sf_comm *c, int *nranks, int *rank);
int sf_comm_allreduce_f64(sf_comm *c, void *d_buf, int64_t count); /* in place, on the context's stream (barriers, timing) */
/* The second transport (SURVEY.md §8e option ii): a hand-written P2P all-gather + fixed-rank-order sum.  Every rank keeps an
 * exchange region in its device memory, exports it (hipIpc) and maps its peers'; an all-reduce is ONE single-workgroup
 * kernel on the context's stream that stores the rank's doubles into every region, raises a flag there, waits for the
 * peers' flags (bounded) and adds the nranks slots in rank order -- bitwise identical on every rank and run to run, a few
 * microseconds over xGMI, and the only transport that runs 2-4 ranks as separate processes on ONE device.
 *   sf_comm_p2p_create     this rank's region; max_count = largest all-reduce in doubles (32 x scans in flight)
 *   sf_comm_p2p_handle     -> SF_COMM_P2P_HANDLE_BYTES the launcher hands to every rank (torch.distributed, MPI, a file)
 *   sf_comm_p2p_connect    <- the nranks handles in rank order
 *   sf_comm_p2p_rendezvous handle + connect through a POSIX shared-memory object `name` (ranks of one node, no launcher)
 * A rank whose wait runs out (sf_comm_set_timeout, default 20 s) or that meets an abort raises the abort word of every
 * region: all ranks leave their collectives and sf_comm_status / sf_icp_align_sharded return SF_ERR_COMM -- nobody is
 * left waiting in a collective.  sf_comm_abort does that from the host (RCCL: ncclCommAbort). */
#define SF_COMM_RCCL 0
#define SF_COMM_P2P 1
#define SF_COMM_P2P_HANDLE_BYTES 128
int sf_comm_p2p_create(sf_ctx *ctx, int nranks, int rank, int64_t max_count, sf_comm **out);
int sf_comm_p2p_handle(sf_comm *c, void *handle);
int sf_comm_p2p_connect(sf_comm *c, const void *handles);
int sf_comm_p2p_rendezvous(sf_comm *c, const char *name, double timeout_s);
int sf_comm_set_timeout(sf_comm *c, double seconds);
int sf_comm_abort(sf_comm *c);
int sf_comm_status(sf_comm *c);
int sf_comm_kind(const sf_comm *c);
/* one whole pass (every iteration) enqueued; first = 1 start, 2 resume the scans that stopped stale */
int sf_icp_align_sharded_async(sf_icp *icp, int mode, sf_comm *comm, int first);
/* blocking: passes until no scan is stale (identical decisions on every rank), results like sf_icp_align_batch */
int sf_icp_align_sharded(sf_icp *icp, int mode, sf_comm *comm, sf_icp_result *out, int *resumes /* or NULL */);
/* several slabs held by ONE process on one device and context (members[m]: slab m + halo indexed, sf_icp_set_shard given,
 * same source batch and initial transforms): lockstep on the stream, the all-reduce is a device sum in member order */
int sf_icp_align_group(sf_icp **members, int n_members, int mode, sf_icp_result *out, int *resumes /* or NULL */);
/* owned-query candidates of this rank per scan in the last sharded alignment */
int sf_icp_owned_counts(sf_icp *icp, int64_t *counts, int cap);
/* Routing ("all-reduce only when the submap spans tiles"): the slabs [lo[b], hi[b]] scan b can reach -- the x-extent of its
 * bounding box under its initial pose (inits: batch x 16 doubles or NULL) widened by margin, against the slab edges
 * (n_slabs + 1 ascending values, first -inf, last +inf).  lo == hi: one rank registers the scan alone, unsharded, no
 * collective; otherwise the ranks lo..hi do, on a communicator of just those ranks.  hi < lo: no finite point. */
int sf_shard_route(const float *xyz, int64_t n_per_scan, int batch, const double *inits, const double *edges, int n_slabs, double margin,
                   int32_t *lo, int32_t *hi);

/* profiling: wraps every NN kernel of the following aligns in hipEvents on the context
 * stream; sf_icp_profile_read returns launches and summed milliseconds since enabling. */
int sf_icp_profile_enable(sf_icp *icp, int on);
int sf_icp_profile_read(sf_icp *icp, int64_t *nn_launches, double *nn_ms_total);
/* every profiled NN launch in launch order: duration [ms] and, for the fused O3D_P2P / P2PLANE kernel, how many queries and
 * waves ran the search in it (the rest kept their certified neighbour); any output may be NULL, *n = launches recorded */
int sf_icp_profile_read_launches(sf_icp *icp, float *ms, uint32_t *searched_queries, uint32_t *searched_waves, int64_t cap, int64_t *n);
/* the sharded path's other phases while profiling is on: per-launch durations [ms] of one kind, in order (ms NULL: count only) */
#define SF_PROF_NN 0
#define SF_PROF_REDUCE 1      /* slab rows -> one exchange record per scan */
#define SF_PROF_COLLECTIVE 2  /* the all-reduce of the records (RCCL or P2P), as the stream saw it: waiting for peers included */
#define SF_PROF_SOLVE 3       /* the identical solve on every rank */
#define SF_PROF_SHARD_BUILD 4 /* owned-query compaction + ordering at the start / resume of an alignment (one host sync inside) */
#define SF_PROF_KINDS 5
int sf_icp_profile_read_phases(sf_icp *icp, int kind, float *ms, int64_t cap, int64_t *n);

/* ------------------------------------------------------------------ BruteForceAlignment (start-up coarse lock) */
/* localization/include/localization/brute_force_alignment.h:22-112 and
 * localization/src/brute_force_alignment.cpp: every candidate pose of the x,y,z,yaw grid is
 * scored by the mean SQUARED NN distance of the source points (all candidates of one x slice
 * in one launch); the first candidate under the threshold, in the reference's nesting
 * order, wins.  Same setters, same state (previous / best transformation, first-alignment
 * flag, the trace == 4 rule of setInitialGuess). */
typedef struct sf_bf sf_bf;
int sf_bf_create(sf_ctx *ctx, sf_bf **out);
void sf_bf_destroy(sf_bf *bf);
int sf_bf_set_xyz_step(sf_bf *bf, float x_step, float y_step, float z_step);
int sf_bf_set_xyz_range(sf_bf *bf, float x, float y, float z);
int sf_bf_set_rotation_step(sf_bf *bf, float yaw_step);
int sf_bf_set_rotation_range(sf_bf *bf, float yaw);
int sf_bf_set_mean_error_threshold(sf_bf *bf, float error_threshold);
int sf_bf_set_initial_guess(sf_bf *bf, const float T[16]);
int sf_bf_set_source(sf_bf *bf, const float *xyz, int64_t n);
int sf_bf_set_source_cloud(sf_bf *bf, sf_cloud *cloud);
int sf_bf_set_target(sf_bf *bf, const float *xyz, int64_t n);
int sf_bf_set_target_map(sf_bf *bf, sf_map *map);
int sf_bf_reset_first_alignment(sf_bf *bf, int value);
int sf_bf_align_clouds(sf_bf *bf, int *found);
int sf_bf_first_alignment_completed(sf_bf *bf);
int sf_bf_get_best_transformation(sf_bf *bf, float T[16]);
/* diagnostics of the last alignClouds: chosen candidate (nesting-order index), its score,
 * and the float32 score of every candidate evaluated before the early exit (NaN after it) */
int sf_bf_last_result(sf_bf *bf, int32_t *index, float *score, int32_t *n_candidates, float *scores, int64_t cap);

/* ------------------------------------------------------------------ GlobalMapFramesManager (start-up I/O) */
/* localization/src/global_map_frames_manager.cpp: map.pcd cache or merge of the recorded
 * cloud_<n>.pcd tiles + PCL voxel grid (on the device) + save (:93-151); map_T_global from
 * odometry_positions.txt / gps_imu_poses.txt with the bad-reading filter (:153-248);
 * altitude look-up table (:60-63, :69-91). */
typedef struct sf_frames sf_frames;
sf_frames *sf_frames_create(const char *data_folder, const char *map_name, int64_t num_poses_max);
void sf_frames_destroy(sf_frames *fr);
int sf_frames_get_map_cloud(sf_frames *fr, sf_cloud *out, float voxel_size, int *loaded_cached);
int sf_frames_get_map_T_global(sf_frames *fr, double T[16]);
float sf_frames_get_closest_altitude(sf_frames *fr, double lat, double lon);
int sf_frames_altitude_table(sf_frames *fr, double *table_lat_lon_alt, int64_t cap_rows, int64_t *rows);

/* ------------------------------------------------------------------ MapDataSaver file writers (recorder side of f-3) */
/* mapping/src/map_data_save_node.cpp:12-29 (folder wiped and re-created, "tx ty tz" / "lat lon alt y" headers),
 * :61-98 (per synchronized cloud/GPS/odometry triple: cloud appended to the open tile, cloud_<counter>.pcd written
 * with savePCDFileBinary every 10 clouds -- map_data_save_node.h:72 --, one line per log: odometry with default
 * ostream formatting, GPS with std::fixed << std::setprecision(8)), :100-112 (open tile flushed on shutdown).
 * compass_yaw is the recorder's own double conversion of the heading (:38-49): sf_recorder_compass_yaw. */
typedef struct sf_recorder sf_recorder;
int sf_recorder_create(const char *map_data_path, sf_recorder **out);
int sf_recorder_add(sf_recorder *r, const float *xyz, int64_t n, const double odom_xyz[3], double lat, double lon, double alt, double compass_yaw);
int sf_recorder_shutdown(sf_recorder *r);
void sf_recorder_destroy(sf_recorder *r);
double sf_recorder_compass_yaw(double compass_hdg_deg);

/* ------------------------------------------------------------------ pose fusion (host, float32 like the reference) */
/* a14: computePosePredictionFromOdometry — localization_node.cpp:89-110 */
void sf_fusion_quat_to_pose(const double q_wxyz[4], const double t[3], float T[16]);
void sf_fusion_odom_prediction(const float map_T_sensor[16], const float odom_T_prev[16],
                               const float odom_T_cur[16], float out[16]);
/* a15: compass callback :64-76, UTM::LLtoUTM geo_lib.hpp:38-83, getClosestAltitude
 * global_map_frames_manager.cpp:69-91, computeGpsCoarsePoseInMapFrame :112-128 */
float sf_fusion_compass_to_yaw(double compass_deg);
void sf_fusion_ll_to_utm(double lat, double lon, double *northing, double *easting);
float sf_fusion_closest_altitude(const double *table_lat_lon_alt, int rows, double lat, double lon);
void sf_fusion_gps_pose(const double map_T_global[16], float yaw, double lat, double lon,
                        float table_alt, float out[16]);
/* a16: computePoseGainsFromCovarianceMatrices — localization_node.cpp:151-179 */
void sf_fusion_pose_gains(const double gps_cov[9], const double odom_cov[36], int fixed,
                          float *odom_gain, float *gps_gain);
/* a17: prior blend — localization_node.cpp:329 */
void sf_fusion_blend(float g_odom, const float T_odom[16], float g_gps, const float T_gps[16],
                     float out[16]);
/* computeMapTGlobal — global_map_frames_manager.cpp:209-248 */
void sf_fusion_map_T_global(const double *latlonalt, const float *yaw, int n, double out[16]);
void sf_fusion_mat4f_inverse(const float A[16], float out[16]);
void sf_fusion_mat4f_mul(const float A[16], const float B[16], float out[16]);
/* a18: StochasticFilter — stochastic_filter.cpp:3-27,44-113 */
typedef struct sf_sfilter sf_sfilter;
sf_sfilter *sf_sfilter_create(int queue_size, float n_std_dev_threshold);
void sf_sfilter_destroy(sf_sfilter *f);
void sf_sfilter_set_maximum_linear_velocity(sf_sfilter *f, float v);
void sf_sfilter_weights(const sf_sfilter *f, float *w);
void sf_sfilter_add_pose_to_queue(sf_sfilter *f, const float pose[16]);
float sf_sfilter_pose_zscore(const sf_sfilter *f, const float prev[16], const float cur[16]);
void sf_sfilter_apply_gaussian_filter(const sf_sfilter *f, const float prev[16],
                                      const float cur[16], float out[16]);

/* ------------------------------------------------------------------ the node's per-scan callback as one call
 * LocalizationNode::localizationCallback and its helpers -- localization/src/localization_node.cpp:263-344 (callback),
 * :62-77 (compass), :112-128 (GPS pose), :181-261 (coarse alignment), constructor constants :19-43 -- over the entry
 * points above: same order, same constants, same state, everything device-side on the context's stream.  The ROS 2 shell
 * (subscriptions, synchroniser, publishers) stays with the caller. */
typedef struct sf_node sf_node;
typedef struct {
    float ref_frame_distance; /* 3 m: re-crop distance, localization_node.h:142                                     */
    float cloud_crop_radius;  /* 10 m: scan and map crop radius, localization_node.h:145                            */
    float index_cell;         /* grid cell of the whole-map index (0.5 m; 0 = automatic)                              */
    int32_t pcl_crop_order;   /* 1: the scan crop keeps PCL's ascending-distance order (point_cloud_processing.hpp:40-52) */
    int32_t icp_mode;         /* SF_ICP_REF_CPP (the node's ICPPointToPoint)                                          */
    int32_t map_is_downsampled; /* 0: apply getMapCloud's 0.1 m voxel grid (:19) before the stride-3 subsample (:20) */
} sf_node_params;
typedef struct { /* the sensor_msgs/NavSatFix fields the node reads */
    double latitude, longitude, altitude;
    double position_covariance[9];
} sf_gps_fix;
typedef struct { /* the nav_msgs/Odometry fields the node reads */
    double q_wxyz[4];
    double t[3];
    double covariance[36];
} sf_odom;
#define SF_NODE_OK 0              /* map_T_sensor is the new pose                                           */
#define SF_NODE_GATED_ALTITUDE 1  /* negative GPS altitude: message dropped (:269-276)                       */
#define SF_NODE_FIRST_MESSAGE 2   /* first message: pose initialised from GPS + compass, no alignment (:278-283) */
#define SF_NODE_COARSE_FAILED 3   /* the start-up lock was not found on this scan (:307-315)                 */
typedef struct {
    int32_t status;          /* SF_NODE_*                                                                     */
    int32_t recropped;       /* the map window moved on this scan (:299-305)                                  */
    int32_t coarse_ran;      /* 0 no, 1 brute force only, 2 brute force + the "strong" ICP                    */
    int32_t pad_;
    int64_t n_scan;          /* scan points after the stride-2 subsample and the radius crop                  */
    float map_T_sensor[16];  /* the node's pose after this message                                            */
    float prior[16];         /* the filtered prior handed to the ICP (:318-332)                               */
    float odom_pose[16], gps_pose[16];
    float odometry_gain, gps_compass_gain;
    sf_icp_result icp;       /* the fine alignment                                                            */
    sf_icp_result coarse_icp; /* the "strong" ICP of the coarse phase when coarse_ran == 2                    */
} sf_node_output;
void sf_node_default_params(sf_node_params *p);
/* map_xyz: the map cloud (getMapCloud's output when map_is_downsampled); altitude table rows = (lat, lon, alt) */
int sf_node_create(sf_ctx *ctx, const float *map_xyz, int64_t n_map, const double map_T_global[16], const double *altitude_table_lat_lon_alt, int rows,
                   const sf_node_params *params /* or NULL */, sf_node **out);
void sf_node_destroy(sf_node *n);
int sf_node_compass(sf_node *n, double compass_deg); /* compassCallback :62-77 */
int sf_node_callback_xyz(sf_node *n, const float *xyz, int64_t n_points, const sf_gps_fix *gps, const sf_odom *odom, sf_node_output *out);
int sf_node_callback_pointcloud2(sf_node *n, const void *data, int64_t data_bytes, int64_t width, int64_t height, int point_step, int64_t row_step, int off_x, int off_y,
                                 int off_z, int datatype, int is_bigendian, const sf_gps_fix *gps, const sf_odom *odom, sf_node_output *out);
#define SF_NODE_POSE_MAP_T_SENSOR 0
#define SF_NODE_POSE_MAP_T_REF 1
#define SF_NODE_POSE_ODOM_PREVIOUS 2
int sf_node_set_pose(sf_node *n, int which, const float T[16]); /* callers that already hold a lock */
int sf_node_get_pose(sf_node *n, int which, float T[16]);
int sf_node_set_coarse_alignment_complete(sf_node *n, int complete);
int sf_node_coarse_alignment_complete(sf_node *n);
sf_icp *sf_node_icp(sf_node *n); /* the node's icp_ (parameters, counters) */
struct sf_bf;
struct sf_bf *sf_node_bf(sf_node *n); /* the node's brute_force_alignment_ (pose grid, threshold) */

/* f-4 (EXTENSION, no reference counterpart): error-state EKF pose prior with IMU pre-integration.
 * The reference has no EKF and never reads the IMU (SURVEY.md "Read this first"); its prior is the
 * blend + StochasticFilter above.  Nominal state: position, velocity (map frame), attitude R (map <- sensor),
 * gyro bias, accelerometer bias; 15x15 covariance over (dp, dv, dtheta, dbg, dba), attitude error on the right.
 * Host code, float64, row-major.  After sf_ekf_reset the bias states carry zero variance and zero random walk, i.e.
 * the filter is the 9-state (p, v, theta) one until sf_ekf_set_bias / sf_ekf_set_bias_noise give them room. */
typedef struct sf_ekf sf_ekf;
int sf_ekf_create(sf_ekf **out);
void sf_ekf_destroy(sf_ekf *e);
int sf_ekf_reset(sf_ekf *e, const double T[16], const double v[3] /* or NULL */, const double P_diag[9] /* or NULL: identity */);
int sf_ekf_set_noise(sf_ekf *e, double gyro_sigma, double accel_sigma, const double gravity[3] /* or NULL: (0, 0, -9.80665) */);
/* bias estimates and their variances (any argument may be NULL = unchanged); bias random walks per sqrt(second) */
int sf_ekf_set_bias(sf_ekf *e, const double gyro_bias[3], const double accel_bias[3], const double gyro_bias_var[3], const double accel_bias_var[3]);
int sf_ekf_set_bias_noise(sf_ekf *e, double gyro_bias_walk, double accel_bias_walk);
/* n IMU samples (rad/s, specific force m/s^2 in the sensor frame) of period dt, integrated one by one */
int sf_ekf_predict_imu(sf_ekf *e, const double *gyro, const double *accel, int64_t n, double dt);
/* the reference's prediction (localization_node.cpp:89-110) as an EKF step: pose <- pose * (prev^-1 cur), covariance
 * through the step's Jacobian (dp <- -R [d_t]x dtheta, dtheta <- d_R^T dtheta) plus the odometry noise */
int sf_ekf_predict_odometry(sf_ekf *e, const double odom_T_prev[16], const double odom_T_cur[16], const double cov_pos[3], const double cov_rot[3]);
int sf_ekf_update_position(sf_ekf *e, const double p_map[3], const double cov[9]);   /* GPS, already in the map frame (sf_fusion_gps_pose) */
int sf_ekf_update_yaw(sf_ekf *e, double yaw, double var);                            /* compass (sf_fusion_compass_to_yaw) */
int sf_ekf_update_pose(sf_ekf *e, const double T[16], const double cov_pos[3], const double cov_rot[3]);  /* ICP result */
int sf_ekf_get(const sf_ekf *e, double T[16], double v[3], double P[81]);            /* any output may be NULL; P = the (dp, dv, dtheta) block */
int sf_ekf_get_full(const sf_ekf *e, double gyro_bias[3], double accel_bias[3], double P[225]); /* biases and the whole 15x15 covariance */

#ifdef __cplusplus
}
#endif
#endif
