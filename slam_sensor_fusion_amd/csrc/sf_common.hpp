// sf_common.hpp — internals shared by the libslamfusion translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "slamfusion.h"

namespace sf {

void set_error(const char *fmt, ...);

#define SF_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            sf::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                          __LINE__);                                                          \
            return SF_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define SF_CHECK(cond, code, ...)                                                             \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            sf::set_error(__VA_ARGS__);                                                       \
            return code;                                                                      \
        }                                                                                     \
    } while (0)

#define SF_TRY(expr)                                                                          \
    do {                                                                                      \
        int rc_ = (expr);                                                                     \
        if (rc_ != SF_OK) return rc_;                                                         \
    } while (0)

// growable device buffer (never shrinks: keeps graph-captured pointers stable)
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    uint32_t epoch = 0; // bumped whenever the allocation (hence the pointer) changes: captured hipGraphs key on it
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); } // error paths between reserve() and release() do not leak
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SF_OK;
        ++epoch;
        if (p) {
            hipError_t e = hipFree(p);
            (void)e;
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return SF_ERR_NOMEM;
        }
        cap = want;
        return SF_OK;
    }
    void release()
    {
        if (p) {
            hipError_t e = hipFree(p);
            (void)e;
            ++epoch;
        }
        p = nullptr;
        cap = 0;
    }
    // exchange the allocations of two buffers (ping-pong outputs without a hipMalloc / hipFree per call)
    void swap(DevBuf &o)
    {
        void *tp = p; p = o.p; o.p = tp;
        size_t tc = cap; cap = o.cap; o.cap = tc;
        ++epoch; ++o.epoch;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

} // namespace sf

struct sf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipDeviceProp_t prop;
    sf::DevBuf scratch;   // rocPRIM temporary storage
    sf::DevBuf scratch2;  // block counts / small reductions
    sf::DevBuf sort_hist, scan_tiles; // sf_sort.hpp: digit histograms of the radix sort, tile sums of the scans
    sf::DevBuf merge_tmp;  // sf_cloud_voxel_merge: per-voxel tables of the pending points
    uint64_t merge_epoch = 0; // bumped by every merge that rewrites merge_tmp (sf_cloud::MergeRecord::epoch)
    sf::DevBuf vox_tmp[6]; // voxel grids: keys, keys', point ids, point ids', head flags, positions -- kept between calls (a growing map re-voxelises every few scans)
    void *h_pinned = nullptr; // small pinned staging (4 KiB)
    // pageable -> device uploads go through two pinned buffers in turn (sf::upload_staged): the runtime's own staging of
    // a pageable hipMemcpyAsync moves ~5 GB/s and holds the host until the stream has reached the copy
    struct Stage { void *p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool pending = false; } stage[2];
    int stage_next = 0;
    int refs = 0;          // live clouds / maps / icps created on this context
    bool zombie = false;   // sf_ctx_destroy was called while children were alive
};

// AoS xyz float32 point set on the device
struct sf_cloud {
    sf_ctx *ctx = nullptr;
    sf::DevBuf xyz;       // float[n][3]
    int64_t n = 0;
    sf::DevBuf spare;     // output of the next crop / subsample / sort (swapped with xyz: no allocation per scan)
    sf::DevBuf flags;     // predicate flags of the crops
    sf::DevBuf raw;       // persistent staging of sf_cloud_from_pointcloud2 / sf_cloud_upload_f64 (no allocation per scan)
    sf::DevBuf last_idx;  // int32 indices kept by the last crop/subsample
    int64_t n_last_idx = -1;
    // voxel introspection (parity tests)
    bool vox_wide = false; // the last downsample kept int64 ids (SF_VOXEL_PCL64)
    sf::DevBuf vox_point_ids; int64_t n_vox_point_vals = 0;
    sf::DevBuf vox_out_ids;   int64_t n_vox_out_vals = 0;
    sf::DevBuf vox_out_means; int64_t n_vox_out_pts = 0; // float64 means (O3D flavour)
    // process-unique stamp of the contents: every call that may change the points takes a new one (sf::cloud_touch).
    // sf_map_build remembers the stamp it indexed; sf_map_patch (sf_map.hip) only trusts an index whose stamp is the one
    // the last merge started from.
    uint64_t stamp = 0;
    // bounds of the points when all are finite and nothing has changed since (stamp == bounds_stamp): a merge knows the bounds it
    // leaves behind unless it replaced a point that held one, so the next merge need not reduce over the map again
    uint64_t bounds_stamp = 0;
    float bounds_mn[3] = {0, 0, 0}, bounds_mx[3] = {0, 0, 0};
    // what the last sf_cloud_voxel_merge that took the merge path did, for sf_map_patch: the tables live in ctx->merge_tmp
    // (valid while ctx->merge_epoch == epoch and stamp == stamp_after)
    struct MergeRecord {
        bool valid = false;
        uint64_t epoch = 0, stamp_before = 0, stamp_after = 0;
        int64_t n_old = 0, n_groups = 0, n_fresh = 0;
        float old_mn[3] = {0, 0, 0}, old_mx[3] = {0, 0, 0}; // bounds of the map before the merge
        float cen_mn[3] = {0, 0, 0}, cen_mx[3] = {0, 0, 0}; // bounds of the centroids it wrote
        bool touched_extreme = false;                       // an old point holding one of the old bounds was replaced: the new bounds are not known without a pass
        const uint32_t *coarse = nullptr;                                                                   // [n_old / 64 + 2]: fresh voxels whose rank is below 64 b
        const uint32_t *g_rank = nullptr, *g_fresh = nullptr, *fresh_pos = nullptr, *fresh_rank = nullptr; // per voxel of the pending points: rank among the old points, 1 = not in the old map, fresh voxels before it; the ranks of the fresh ones
        const float *g_centroid = nullptr, *g_old = nullptr;                                               // its centroid (old point first, then the pending ones); the old point it replaces
    } merge;
};

// window predicate applied inside the NN search (the reference's map crop)
struct SfWindow {
    int kind;          // 0 none, 1 sphere, 2 obb
    float c[3];        // sphere centre
    float r2;          // sphere radius^2 (float, like FLANN)
    double oc[3];      // obb centre
    double oR[9];      // obb axes (columns), row-major
    double ohalf[3];   // obb half extents
};

struct SfGrid {
    float org[3];      // grid origin (min corner)
    float inv_h;       // 1 / cell
    float h;           // cell size
    float gap_eps;     // what a computed query-to-cell-face distance may exceed the true one by (float32 rounding of grid coordinates)
    int dim[3];        // cells per axis
    const uint32_t *cell_start; // [ncell + 1]
    const float4 *pts;          // sorted by cell: x, y, z, bitcast(original index)
    const float4 *nrm;          // sorted normals (w = neighbour count) or nullptr
    int64_t n;
};

struct sf_map {
    sf_ctx *ctx = nullptr;
    sf::DevBuf pts4, nrm4, cell_start, keys, vals, keys2, vals2;
    sf::DevBuf pts4_alt, patch_tmp; // sf_map_patch writes the patched index beside the old one and swaps
    double h_exact = 0;       // the cell as sf_map_build chose it
    float src_mn[3] = {0, 0, 0}, src_mx[3] = {0, 0, 0}; // bounds of the indexed cloud (grid.org: src_mn, snapped down to the origin lattice if there is one)
    int origin_lattice = 0;   // sf_map_set_origin_lattice: 0 = the grid starts at the smallest coordinates, n = at the multiple of n cells below them
    uint64_t src_stamp = 0;   // sf_cloud::stamp of the cloud the index describes
    sf::DevBuf d_window; // the window in device memory (REF_CPP kernels read it there: a captured launch list survives a moving crop)
    sf::DevBuf cov6;   // optional: the 6 unique entries of each point's neighbourhood covariance (sorted order), sf_map_estimate_normals
    int64_t n = 0;
    bool built = false, has_normals = false, has_cov = false;
    uint64_t generation = 0; // process-unique stamp of the index contents (build / normals): captured hipGraphs key on it
    SfGrid grid{};
    SfWindow window{};
};

struct sf_comm;

namespace sf {
// sf_shard.cpp: in-place float64 sum over the communicator, enqueued on its context's stream
int comm_allreduce_f64(sf_comm *c, void *d_buf, int64_t count);
sf_ctx *comm_ctx(const sf_comm *c);
// device-side helpers implemented in sf_cloud.hip, used across TUs
int compact_cloud(sf_cloud *c, const uint8_t *d_flags);
int ensure_scratch(sf_ctx *ctx, size_t bytes);
// dst (device) <- src (any host memory), stream-ordered on the context's stream; src may be freed on return
int upload_staged(sf_ctx *ctx, void *dst, const void *src, size_t bytes);
uint64_t next_generation();
inline void cloud_touch(sf_cloud *c) { c->stamp = next_generation(); c->merge.valid = false; }
// children keep their context alive: any destruction order is safe
void ctx_retain(sf_ctx *ctx);
void ctx_release(sf_ctx *ctx);
struct MinMaxHost { float mn[3], mx[3]; int64_t n_finite; };
struct MinMaxDev { float mn[3], mx[3]; unsigned long long cnt; }; // zeros when no point is finite
// the same reduction left on the device (no host synchronisation): *d_out is written on the stream
int cloud_minmax_enqueue(sf_ctx *ctx, const float *d_xyz, int64_t n, MinMaxDev *d_out);
// min/max over the finite points of a device AoS cloud (synchronises the stream)
int cloud_minmax(sf_ctx *ctx, const float *d_xyz, int64_t n, MinMaxHost *out);
} // namespace sf
