// sf_voxel.hip — voxel-grid downsampling of the map on the device (gfx950).
//
// SF_VOXEL_PCL restates pcl::VoxelGrid<PointXYZ> (float32) as the reference calls it at
// localization/src/global_map_frames_manager.cpp:142-146 (leaf 0.1f from
// localization/src/localization_node.cpp:19): int32 linear voxel index
// i + j*dx + k*dx*dy of floor(p * (1/leaf)) - min_b, centroid per voxel, output in
// ascending index order; int32 overflow -> input returned unchanged (PCL warns).
// SF_VOXEL_O3D restates Open3D voxel_down_sample (float64) as called at
// localization_python/localization_python/localization_node.py:47: index
// floor((p - (min_bound - v/2)) / v), float64 mean per voxel.
// Pipeline: key kernel -> stable radix sort of (key, point id) (sf_sort.hpp, hand-written) -> head flags ->
// exclusive scan (sf_sort.hpp) -> one lane per voxel sums its points sequentially in ascending point
// id (so the float32 / float64 sums are bit-identical to the oracle's) -> scatter.
#include "sf_common.hpp"

#include "sf_sort.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

namespace {

inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)sf::div_up(n > 0 ? n : 1, b); }

struct PclGeom { float inv; int min_b[3]; int64_t mul[3]; uint64_t invalid_key; /* one past the largest voxel index: non-finite points sort last */ };

// K = uint32_t / ID = int32_t: pcl::VoxelGrid's int32 linear index (SF_VOXEL_PCL);
// K = uint64_t / ID = int64_t: the same arithmetic with the index kept in 64 bits (SF_VOXEL_PCL64)
template <class K, class ID>
__global__ void k_pcl_keys(const float *__restrict__ xyz, int64_t n, PclGeom g, K *__restrict__ keys, uint32_t *__restrict__ vals, ID *__restrict__ point_ids)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    K key = (K)g.invalid_key;
    ID id = -1;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
        // voxel_grid.hpp: static_cast<int>(std::floor(p.x * inverse_leaf_size_[0]) - static_cast<float>(min_b_[0]))
        const int i0 = (int)(floorf(__fmul_rn(x, g.inv)) - (float)g.min_b[0]);
        const int i1 = (int)(floorf(__fmul_rn(y, g.inv)) - (float)g.min_b[1]);
        const int i2 = (int)(floorf(__fmul_rn(z, g.inv)) - (float)g.min_b[2]);
        id = (ID)((ID)i0 * (ID)g.mul[0] + (ID)i1 * (ID)g.mul[1] + (ID)i2 * (ID)g.mul[2]);
        key = (K)id;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
    point_ids[i] = id;
}

struct O3dGeom { double vmin[3]; double voxel; };

__global__ void k_o3d_keys(const float *__restrict__ xyz, int64_t n, O3dGeom g, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, int32_t *__restrict__ point_ijk)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // ref_coord = (p - voxel_min_bound) / voxel_size, float64 (PointCloud.cpp VoxelDownSample)
    const int a = (int)floor(__ddiv_rn((double)xyz[3 * i] - g.vmin[0], g.voxel));
    const int b = (int)floor(__ddiv_rn((double)xyz[3 * i + 1] - g.vmin[1], g.voxel));
    const int c = (int)floor(__ddiv_rn((double)xyz[3 * i + 2] - g.vmin[2], g.voxel));
    keys[i] = ((uint64_t)(uint32_t)a << 42) | ((uint64_t)(uint32_t)b << 21) | (uint64_t)(uint32_t)c;
    vals[i] = (uint32_t)i;
    point_ijk[3 * i] = a; point_ijk[3 * i + 1] = b; point_ijk[3 * i + 2] = c;
}

template <class K>
__global__ void k_heads(const K *__restrict__ keys, int64_t n_valid, uint32_t *__restrict__ flags)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_valid) return;
    flags[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
}

// one lane per voxel: float32 sums in ascending point id (pcl::CentroidPoint / AccumulatorXYZ)
template <class K, class ID>
__global__ void k_pcl_centroids(const float *__restrict__ xyz, const K *__restrict__ keys, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ flags,
                                const uint32_t *__restrict__ pos, int64_t n_valid, float *__restrict__ out, ID *__restrict__ out_ids)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_valid || !flags[j]) return;
    const K key = keys[j];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int cnt = 0;
    for (int64_t k = j; k < n_valid && keys[k] == key; ++k) {
        const size_t p = vals[k];
        sx = __fadd_rn(sx, xyz[3 * p]);
        sy = __fadd_rn(sy, xyz[3 * p + 1]);
        sz = __fadd_rn(sz, xyz[3 * p + 2]);
        ++cnt;
    }
    const float c = (float)cnt;
    const size_t o = pos[j];
    out[3 * o] = __fdiv_rn(sx, c); out[3 * o + 1] = __fdiv_rn(sy, c); out[3 * o + 2] = __fdiv_rn(sz, c);
    out_ids[o] = (ID)key;
}

// float64 means in ascending point id (open3d AccumulatedPoint)
__global__ void k_o3d_means(const float *__restrict__ xyz, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ flags,
                            const uint32_t *__restrict__ pos, int64_t n, float *__restrict__ out, double *__restrict__ out64, int32_t *__restrict__ out_ijk)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !flags[j]) return;
    const uint64_t key = keys[j];
    double sx = 0, sy = 0, sz = 0;
    int cnt = 0;
    for (int64_t k = j; k < n && keys[k] == key; ++k) {
        const size_t p = vals[k];
        sx = __dadd_rn(sx, (double)xyz[3 * p]);
        sy = __dadd_rn(sy, (double)xyz[3 * p + 1]);
        sz = __dadd_rn(sz, (double)xyz[3 * p + 2]);
        ++cnt;
    }
    const double c = (double)cnt;
    const size_t o = pos[j];
    const double mx = __ddiv_rn(sx, c), my = __ddiv_rn(sy, c), mz = __ddiv_rn(sz, c);
    out64[3 * o] = mx; out64[3 * o + 1] = my; out64[3 * o + 2] = mz;
    out[3 * o] = (float)mx; out[3 * o + 1] = (float)my; out[3 * o + 2] = (float)mz;
    out_ijk[3 * o] = (int32_t)(key >> 42); out_ijk[3 * o + 1] = (int32_t)((key >> 21) & 0x1fffff); out_ijk[3 * o + 2] = (int32_t)(key & 0x1fffff);
}

// heads -> exclusive scan -> number of voxels (synchronises)
template <class K>
int scan_heads(sf_ctx *ctx, const K *keys, int64_t n_valid, uint32_t *flags, uint32_t *pos, int64_t *n_vox)
{
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_heads<K>, dim3(nblk(n_valid)), dim3(256), 0, st, keys, n_valid, flags);
    SF_TRY(sf::scan_u32<0>(ctx, flags, pos, n_valid));
    uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned);
    SF_HIP(hipMemcpyAsync(h, pos + (n_valid - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SF_HIP(hipMemcpyAsync(h + 1, flags + (n_valid - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    *n_vox = (int64_t)h[0] + (int64_t)h[1];
    return SF_OK;
}

template <class K, class ID>
int voxel_pcl(sf_cloud *c, float leaf, int *status_flags)
{
    constexpr bool WIDE = sizeof(K) == 8;
    sf_ctx *ctx = c->ctx;
    hipStream_t st = ctx->stream;
    const int64_t n = c->n;
    sf::MinMaxHost mm;
    SF_TRY(sf::cloud_minmax(ctx, c->xyz.as<float>(), n, &mm));
    if (mm.n_finite == 0) { c->n = 0; return SF_OK; }
    const float inv = 1.0f / leaf;
    // voxel_grid.hpp overflow test, float32 arithmetic then int64
    const int64_t dx = (int64_t)((mm.mx[0] - mm.mn[0]) * inv) + 1;
    const int64_t dy = (int64_t)((mm.mx[1] - mm.mn[1]) * inv) + 1;
    const int64_t dz = (int64_t)((mm.mx[2] - mm.mn[2]) * inv) + 1;
    if (!WIDE && dx * dy * dz > (int64_t)INT32_MAX) {
        if (status_flags) *status_flags |= SF_FLAG_VOXEL_OVERFLOW;
        return SF_OK; // "Leaf size is too small ... Integer indices would overflow": output = input
    }
    PclGeom g;
    g.inv = inv;
    int64_t div_b[3];
    for (int d = 0; d < 3; ++d) {
        g.min_b[d] = (int)std::floor(mm.mn[d] * inv);
        const int max_b = (int)std::floor(mm.mx[d] * inv);
        div_b[d] = (int64_t)max_b - (int64_t)g.min_b[d] + 1;
    }
    g.mul[0] = 1; g.mul[1] = div_b[0]; g.mul[2] = div_b[0] * div_b[1];
    if (WIDE) SF_CHECK((double)div_b[0] * (double)div_b[1] * (double)div_b[2] < 9.0e18, SF_ERR_OVERFLOW, "voxel index does not fit 63 bits");

    // temporaries live in the context (a growing map re-voxelises every few scans: no hipMalloc / hipFree per call)
    sf::DevBuf &keys = ctx->vox_tmp[0], &keys2 = ctx->vox_tmp[1], &vals = ctx->vox_tmp[2], &vals2 = ctx->vox_tmp[3], &flags = ctx->vox_tmp[4], &pos = ctx->vox_tmp[5];
    sf::DevBuf &out = c->spare;
    SF_TRY(keys.reserve(sizeof(K) * (size_t)n));
    SF_TRY(keys2.reserve(sizeof(K) * (size_t)n));
    SF_TRY(vals.reserve(sizeof(uint32_t) * (size_t)n));
    SF_TRY(vals2.reserve(sizeof(uint32_t) * (size_t)n));
    SF_TRY(flags.reserve(sizeof(uint32_t) * (size_t)n));
    SF_TRY(pos.reserve(sizeof(uint32_t) * (size_t)n));
    SF_TRY(c->vox_point_ids.reserve(sizeof(ID) * (size_t)n));
    g.invalid_key = (uint64_t)div_b[0] * (uint64_t)div_b[1] * (uint64_t)div_b[2];
    hipLaunchKernelGGL((k_pcl_keys<K, ID>), dim3(nblk(n)), dim3(256), 0, st, c->xyz.as<float>(), n, g, keys.as<K>(), vals.as<uint32_t>(), c->vox_point_ids.as<ID>());
    unsigned end_bit = 1; // the bits of the largest key in use (invalid_key itself when a point is not finite)
    while (end_bit < 8 * sizeof(K) && (g.invalid_key >> end_bit) != 0) ++end_bit;
    K *skeys = nullptr;
    uint32_t *svals = nullptr;
    SF_TRY(sf::radix_sort_pairs<K>(ctx, keys.as<K>(), keys2.as<K>(), vals.as<uint32_t>(), vals2.as<uint32_t>(), n, end_bit, &skeys, &svals));
    int64_t n_vox = 0;
    SF_TRY(scan_heads<K>(ctx, skeys, mm.n_finite, flags.as<uint32_t>(), pos.as<uint32_t>(), &n_vox));
    SF_TRY(out.reserve(sizeof(float) * 3 * (size_t)n_vox));
    SF_TRY(c->vox_out_ids.reserve(sizeof(ID) * (size_t)n_vox));
    hipLaunchKernelGGL((k_pcl_centroids<K, ID>), dim3(nblk(mm.n_finite)), dim3(256), 0, st, c->xyz.as<float>(), skeys, svals, flags.as<uint32_t>(),
                       pos.as<uint32_t>(), mm.n_finite, out.as<float>(), c->vox_out_ids.as<ID>());
    SF_HIP(hipGetLastError());
    c->xyz.swap(out); // stream-ordered: nothing here is freed
    c->n_vox_point_vals = n;
    c->n_vox_out_vals = n_vox;
    c->n_vox_out_pts = 0;
    c->vox_wide = WIDE;
    c->n = n_vox;
    c->n_last_idx = -1;
    return SF_OK;
}

int voxel_o3d(sf_cloud *c, double voxel)
{
    sf_ctx *ctx = c->ctx;
    hipStream_t st = ctx->stream;
    const int64_t n = c->n;
    sf::MinMaxHost mm;
    SF_TRY(sf::cloud_minmax(ctx, c->xyz.as<float>(), n, &mm));
    SF_CHECK(mm.n_finite == n, SF_ERR_INVALID, "Open3D-flavour voxel grid needs finite points (%lld of %lld are)", (long long)mm.n_finite, (long long)n);
    O3dGeom g;
    g.voxel = voxel;
    double span = 0;
    for (int d = 0; d < 3; ++d) {
        g.vmin[d] = (double)mm.mn[d] - voxel * 0.5;
        const double vmax = (double)mm.mx[d] + voxel * 0.5;
        span = std::max(span, vmax - g.vmin[d]);
    }
    SF_CHECK(!(voxel * (double)INT_MAX < span), SF_ERR_OVERFLOW, "voxel_size is too small.");
    SF_CHECK(span / voxel + 2 < 2097152.0, SF_ERR_OVERFLOW, "more than 2^21 voxels along one axis");

    sf::DevBuf &keys = ctx->vox_tmp[0], &keys2 = ctx->vox_tmp[1], &vals = ctx->vox_tmp[2], &vals2 = ctx->vox_tmp[3], &flags = ctx->vox_tmp[4], &pos = ctx->vox_tmp[5];
    sf::DevBuf &out = c->spare;
#define VX_TRY(e) SF_TRY(e)
    VX_TRY(keys.reserve(sizeof(uint64_t) * (size_t)n));
    VX_TRY(keys2.reserve(sizeof(uint64_t) * (size_t)n));
    VX_TRY(vals.reserve(sizeof(uint32_t) * (size_t)n));
    VX_TRY(vals2.reserve(sizeof(uint32_t) * (size_t)n));
    VX_TRY(flags.reserve(sizeof(uint32_t) * (size_t)n));
    VX_TRY(pos.reserve(sizeof(uint32_t) * (size_t)n));
    VX_TRY(c->vox_point_ids.reserve(sizeof(int32_t) * 3 * (size_t)n));
    hipLaunchKernelGGL(k_o3d_keys, dim3(nblk(n)), dim3(256), 0, st, c->xyz.as<float>(), n, g, keys.as<uint64_t>(), vals.as<uint32_t>(), c->vox_point_ids.as<int32_t>());
    uint64_t *skeys = nullptr;
    uint32_t *svals = nullptr;
    VX_TRY(sf::radix_sort_pairs<uint64_t>(ctx, keys.as<uint64_t>(), keys2.as<uint64_t>(), vals.as<uint32_t>(), vals2.as<uint32_t>(), n, 63, &skeys, &svals));
    int64_t n_vox = 0;
    VX_TRY(scan_heads<uint64_t>(ctx, skeys, n, flags.as<uint32_t>(), pos.as<uint32_t>(), &n_vox));
    VX_TRY(out.reserve(sizeof(float) * 3 * (size_t)n_vox));
    VX_TRY(c->vox_out_ids.reserve(sizeof(int32_t) * 3 * (size_t)n_vox));
    VX_TRY(c->vox_out_means.reserve(sizeof(double) * 3 * (size_t)n_vox));
    hipLaunchKernelGGL(k_o3d_means, dim3(nblk(n)), dim3(256), 0, st, c->xyz.as<float>(), skeys, svals, flags.as<uint32_t>(), pos.as<uint32_t>(), n,
                       out.as<float>(), c->vox_out_means.as<double>(), c->vox_out_ids.as<int32_t>());
    SF_HIP(hipGetLastError());
#undef VX_TRY
    c->xyz.swap(out);
    c->vox_wide = false;
    c->n_vox_point_vals = 3 * n;
    c->n_vox_out_vals = 3 * n_vox;
    c->n_vox_out_pts = n_vox;
    c->n = n_vox;
    c->n_last_idx = -1;
    return SF_OK;
}

int download_i32(sf_cloud *c, const sf::DevBuf &buf, int64_t have, int32_t *dst, int64_t cap, int64_t *n)
{
    if (n) *n = have;
    SF_CHECK(cap >= have && (dst || have == 0), SF_ERR_INVALID, "buffer too small: %lld < %lld", (long long)cap, (long long)have);
    if (have > 0) {
        SF_HIP(hipMemcpyAsync(dst, buf.p, sizeof(int32_t) * (size_t)have, hipMemcpyDeviceToHost, c->ctx->stream));
        SF_HIP(hipStreamSynchronize(c->ctx->stream));
    }
    return SF_OK;
}

} // namespace

// ------------------------------------------------------------------ incremental map growth: the voxel grid as a merge
// `*map_cloud += *cloud` followed by the voxel filter (global_map_frames_manager.cpp:131,142-146) when map_cloud already
// IS a voxel-filtered cloud (config 4: the map grows by a few registered scans at a time).  The filtered map is sorted by
// voxel index and holds one point per voxel, so re-filtering the concatenation needs no sort of the map: key the old points
// (they re-enter the filter as points: a centroid is keyed by where it lies, exactly as the full filter would), check that
// the keys are strictly ascending (a centroid that rounding has pushed across a voxel face breaks that: rare -> full path),
// sort only the pending points, find each of their voxels among the old keys by binary search, sum old point first and
// then the pending ones in ascending id (the oracle's float32 order), and copy the map once, opening the gaps for the
// new voxels.  The result is bit-identical to the full path (tests/test_gpu_map_growth.py, test_gpu_config4_stream.py).
namespace {

__device__ __forceinline__ uint32_t pcl_key_u32(const PclGeom &g, float x, float y, float z)
{
    // (k_pcl_keys for a finite point)
    const int i0 = (int)(floorf(__fmul_rn(x, g.inv)) - (float)g.min_b[0]);
    const int i1 = (int)(floorf(__fmul_rn(y, g.inv)) - (float)g.min_b[1]);
    const int i2 = (int)(floorf(__fmul_rn(z, g.inv)) - (float)g.min_b[2]);
    return (uint32_t)(i0 * (int)g.mul[0] + i1 * (int)g.mul[1] + i2 * (int)g.mul[2]);
}

// the old (filtered, finite) points: keys under the union's geometry, which must be strictly ascending -- one pass, keys only
__global__ __launch_bounds__(256) void k_merge_old_keys(const float *__restrict__ xyz, int64_t n, PclGeom g, uint32_t *__restrict__ keys, uint32_t *__restrict__ bad)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t k = pcl_key_u32(g, xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]);
    keys[j] = k;
    uint32_t prev = __shfl_up(k, 1, 64);
    if ((threadIdx.x & 63) == 0 && j > 0) prev = pcl_key_u32(g, xyz[3 * j - 3], xyz[3 * j - 2], xyz[3 * j - 1]);
    if (j > 0 && k <= prev) *bad = 1u;
}

// coarse[b] = number of fresh voxels whose rank is below 64 b: old point j moves up by coarse[j / 64] plus the few in its own 64
__global__ void k_merge_coarse(const uint32_t *__restrict__ fresh_rank, int64_t n_fresh, int64_t n_coarse, uint32_t *__restrict__ coarse)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_coarse) return;
    const uint64_t v = (uint64_t)b * 64u;
    int64_t lo = 0, hi = n_fresh;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)fresh_rank[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    coarse[b] = (uint32_t)lo;
}

// one lane per voxel of the pending points: where it goes in the old map and its centroid
__global__ void k_merge_groups(const float *__restrict__ old_xyz, const uint32_t *__restrict__ old_keys, int64_t n_old, const float *__restrict__ new_xyz,
                               const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos,
                               int64_t m_valid, uint32_t *__restrict__ g_rank, uint32_t *__restrict__ g_fresh, float *__restrict__ g_centroid, float *__restrict__ g_old)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m_valid || !flags[j]) return;
    const uint32_t key = keys[j];
    int64_t lo = 0, hi = n_old; // first old key >= key
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (old_keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    const bool found = lo < n_old && old_keys[lo] == key;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int cnt = 0;
    if (found) { // the old point has the smallest id of its voxel: first in the sum
        sx = __fadd_rn(sx, old_xyz[3 * lo]); sy = __fadd_rn(sy, old_xyz[3 * lo + 1]); sz = __fadd_rn(sz, old_xyz[3 * lo + 2]);
        cnt = 1;
    }
    const float ox = sx, oy = sy, oz = sz; // (0 + x = x: the old point itself, for sf_map_patch)
    for (int64_t k = j; k < m_valid && keys[k] == key; ++k) {
        const size_t p = vals[k];
        sx = __fadd_rn(sx, new_xyz[3 * p]); sy = __fadd_rn(sy, new_xyz[3 * p + 1]); sz = __fadd_rn(sz, new_xyz[3 * p + 2]);
        ++cnt;
    }
    const float c = (float)cnt;
    const size_t g = pos[j];
    g_rank[g] = (uint32_t)lo;
    g_fresh[g] = found ? 0u : 1u;
    g_centroid[3 * g] = __fdiv_rn(sx, c); g_centroid[3 * g + 1] = __fdiv_rn(sy, c); g_centroid[3 * g + 2] = __fdiv_rn(sz, c);
    g_old[3 * g] = ox; g_old[3 * g + 1] = oy; g_old[3 * g + 2] = oz;
}

__device__ __forceinline__ int float_ordered(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
inline float float_from_ordered(int o)
{
    const int i = o >= 0 ? o : o ^ 0x7fffffff;
    float f;
    memcpy(&f, &i, sizeof(f));
    return f;
}

struct MergeExtremes { int mn[3], mx[3]; uint32_t touched_extreme, pad; };

// bounds of the centroids; does a replaced old point hold one of the map's bounds?  (a few workgroups, grid-stride: the six
// atomics per workgroup all land on one cache line)
__global__ __launch_bounds__(256) void k_merge_extremes(const uint32_t *__restrict__ g_fresh, const float *__restrict__ g_centroid, const float *__restrict__ g_old, int64_t n_groups,
                                                         float mnx, float mny, float mnz, float mxx, float mxy, float mxz, MergeExtremes *__restrict__ out)
{
    __shared__ int s_lo[4][3], s_hi[4][3];
    int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    bool touched = false;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int o = float_ordered(g_centroid[3 * g + d]);
            lo[d] = min(lo[d], o);
            hi[d] = max(hi[d], o);
        }
        if (!g_fresh[g]) {
            const float px = g_old[3 * g], py = g_old[3 * g + 1], pz = g_old[3 * g + 2];
            touched |= px <= mnx || py <= mny || pz <= mnz || px >= mxx || py >= mxy || pz >= mxz;
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { lo[d] = min(lo[d], __shfl_xor(lo[d], s, 64)); hi[d] = max(hi[d], __shfl_xor(hi[d], s, 64)); }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][d] = lo[d]; s_hi[threadIdx.x >> 6][d] = hi[d]; }
    }
    if (touched) out->touched_extreme = 1u;
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        atomicMin(&out->mn[d], min(min(s_lo[0][d], s_lo[1][d]), min(s_lo[2][d], s_lo[3][d])));
        atomicMax(&out->mx[d], max(max(s_hi[0][d], s_hi[1][d]), max(s_hi[2][d], s_hi[3][d])));
    }
}

// the ranks of the fresh voxels, compacted (ascending, since the groups are)
__global__ void k_merge_fresh_ranks(const uint32_t *__restrict__ g_rank, const uint32_t *__restrict__ g_fresh, const uint32_t *__restrict__ fresh_pos, int64_t n_groups,
                                    uint32_t *__restrict__ fresh_rank)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_groups && g_fresh[g]) fresh_rank[fresh_pos[g]] = g_rank[g];
}

// old point j moves up by the number of fresh voxels that sort before it (fresh_rank <= j)
__global__ __launch_bounds__(256) void k_merge_copy(const float *__restrict__ old_xyz, int64_t n_old, const uint32_t *__restrict__ fresh_rank, const uint32_t *__restrict__ coarse,
                                                     float *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_old) return;
    uint32_t f = coarse[j >> 6];
    const uint32_t f_end = coarse[(j >> 6) + 1];
    while (f < f_end && (int64_t)fresh_rank[f] <= j) ++f;
    const size_t d = (size_t)j + (size_t)f;
    out[3 * d] = old_xyz[3 * j]; out[3 * d + 1] = old_xyz[3 * j + 1]; out[3 * d + 2] = old_xyz[3 * j + 2];
}

// the centroids of the touched and the fresh voxels into their places
__global__ void k_merge_place(const uint32_t *__restrict__ g_rank, const uint32_t *__restrict__ g_fresh, const uint32_t *__restrict__ fresh_pos, const float *__restrict__ g_centroid,
                              int64_t n_groups, float *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    // fresh_pos[g] = fresh voxels before group g: an old point of rank r has exactly that many fresh voxels sorting before it
    const size_t d = (size_t)g_rank[g] + (size_t)fresh_pos[g];
    out[3 * d] = g_centroid[3 * g]; out[3 * d + 1] = g_centroid[3 * g + 1]; out[3 * d + 2] = g_centroid[3 * g + 2];
}

} // namespace

// below this map size the full filter is as fast (measured: 0.24 against 0.52 ms at 1 M points -- the merge is bound by its ~25 launches
// and three host round trips --, 0.29 against 0.33 at 3 M, 1.17 against 0.65 at 19 M)
static int64_t g_merge_min_points = 4000000;
extern "C" int sf_cloud_voxel_merge_min_points(int64_t n)
{
    const int64_t old = g_merge_min_points;
    if (n >= 0) g_merge_min_points = n;
    return (int)std::min<int64_t>(old, INT32_MAX);
}

extern "C" int sf_cloud_voxel_merge(sf_cloud *map, sf_cloud *pending, double leaf_d, int *status_flags, int *merged)
{
    SF_CHECK(map && pending && map != pending && leaf_d > 0, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(map->ctx == pending->ctx, SF_ERR_INVALID, "both clouds must live on the same context");
    if (status_flags) *status_flags = 0;
    if (merged) *merged = 0;
    sf_ctx *ctx = map->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = map->n, m = pending->n;
    const float leaf = (float)leaf_d, inv = 1.0f / leaf;
    const uint64_t stamp_before = map->stamp;
    if (m > 0) sf::cloud_touch(map); // (an empty pending cloud leaves the filtered map as it is)
    auto full_path = [&]() -> int {
        SF_TRY(sf_cloud_append(map, pending));
        return sf_cloud_voxel_downsample(map, leaf_d, SF_VOXEL_PCL, status_flags);
    };
    if (n == 0 || m == 0 || n + m >= (int64_t)0x7fffffff || n < g_merge_min_points) return full_path();
    sf::MinMaxDev mm_a, mm_b;
    // geometry of the union, as pcl::VoxelGrid computes it over the concatenated cloud
    // (both reductions enqueued, ONE synchronisation: on a map of a few million points the merge is bound by its host round trips)
    sf::DevBuf &mt = ctx->merge_tmp;
    const int64_t n_coarse = n / 64 + 2;
    SF_TRY(mt.reserve(sizeof(uint32_t) * 4 * (size_t)m + sizeof(float) * 6 * (size_t)m + 64 + 2 * sizeof(sf::MinMaxDev) + sizeof(uint32_t) * (size_t)n_coarse));
    sf::MinMaxDev *d_mm = reinterpret_cast<sf::MinMaxDev *>(mt.as<unsigned char>() + sizeof(uint32_t) * 4 * (size_t)m + sizeof(float) * 6 * (size_t)m + 64);
    uint32_t *coarse = reinterpret_cast<uint32_t *>(d_mm + 2);
    const bool know_bounds = stamp_before != 0 && map->bounds_stamp == stamp_before; // (left behind by the previous merge: all points finite)
    if (!know_bounds) SF_TRY(sf::cloud_minmax_enqueue(ctx, map->xyz.as<float>(), n, d_mm));
    {
        sf::MinMaxDev h_a;
        if (!know_bounds) SF_HIP(hipMemcpyAsync(&h_a, d_mm, sizeof(h_a), hipMemcpyDeviceToHost, st)); // (the partial buffer of the reduction is reused by the next one: stream order)
        SF_TRY(sf::cloud_minmax_enqueue(ctx, pending->xyz.as<float>(), m, d_mm + 1));
        sf::MinMaxDev h_b;
        SF_HIP(hipMemcpyAsync(&h_b, d_mm + 1, sizeof(h_b), hipMemcpyDeviceToHost, st));
        SF_HIP(hipStreamSynchronize(st));
        if (know_bounds) {
            for (int d = 0; d < 3; ++d) { h_a.mn[d] = map->bounds_mn[d]; h_a.mx[d] = map->bounds_mx[d]; }
            h_a.cnt = (unsigned long long)n;
        }
        mm_a = h_a;
        mm_b = h_b;
    }
    sf::MinMaxHost a, b;
    for (int d = 0; d < 3; ++d) { a.mn[d] = mm_a.mn[d]; a.mx[d] = mm_a.mx[d]; b.mn[d] = mm_b.mn[d]; b.mx[d] = mm_b.mx[d]; }
    a.n_finite = (int64_t)mm_a.cnt;
    b.n_finite = (int64_t)mm_b.cnt;
    if (a.n_finite != n || b.n_finite == 0) return full_path();
    float mn[3], mx[3];
    for (int d = 0; d < 3; ++d) { mn[d] = std::min(a.mn[d], b.mn[d]); mx[d] = std::max(a.mx[d], b.mx[d]); }
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) return full_path(); // (PCL's overflow behaviour lives there)
    PclGeom g;
    g.inv = inv;
    int64_t div_b[3];
    for (int d = 0; d < 3; ++d) {
        g.min_b[d] = (int)std::floor(mn[d] * inv);
        div_b[d] = (int64_t)(int)std::floor(mx[d] * inv) - (int64_t)g.min_b[d] + 1;
    }
    g.mul[0] = 1; g.mul[1] = div_b[0]; g.mul[2] = div_b[0] * div_b[1];
    g.invalid_key = (uint64_t)div_b[0] * (uint64_t)div_b[1] * (uint64_t)div_b[2];
    unsigned end_bit = 1;
    while (end_bit < 32 && (g.invalid_key >> end_bit) != 0) ++end_bit;

    sf::DevBuf &old_keys = ctx->vox_tmp[0], &keys = ctx->vox_tmp[1], &vals = ctx->vox_tmp[2], &vals2 = ctx->vox_tmp[3], &flags = ctx->vox_tmp[4], &pos = ctx->vox_tmp[5];
    SF_TRY(old_keys.reserve(sizeof(uint32_t) * (size_t)(n + 2 * m)));
    SF_TRY(keys.reserve(sizeof(uint32_t) * (size_t)(n + 2 * m)));
    SF_TRY(vals.reserve(sizeof(uint32_t) * (size_t)(n + 2 * m)));
    SF_TRY(vals2.reserve(sizeof(uint32_t) * (size_t)(n + 2 * m)));
    SF_TRY(flags.reserve(sizeof(uint32_t) * (size_t)(n + 2 * m)));
    SF_TRY(pos.reserve(sizeof(uint32_t) * (size_t)(n + 2 * m)));
    // per group: rank, fresh flag, fresh prefix, fresh ranks, centroid; + the "keys not ascending" flag (mt reserved above)
    uint32_t *g_rank = mt.as<uint32_t>(), *g_fresh = g_rank + m, *fresh_pos = g_fresh + m, *fresh_rank = fresh_pos + m;
    ++ctx->merge_epoch;
    float *g_centroid = reinterpret_cast<float *>(fresh_rank + m), *g_old = g_centroid + 3 * m;
    uint32_t *bad = reinterpret_cast<uint32_t *>(g_old + 3 * m);
    MergeExtremes *d_ext = reinterpret_cast<MergeExtremes *>(bad + 4); // (inside the 64 spare bytes)
    SF_TRY(map->vox_point_ids.reserve(sizeof(int32_t) * (size_t)m)); // (the key kernel writes per-point ids: scratch here)
    SF_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), st));
    // old points (all finite, checked above): keys under the union's geometry, must be strictly ascending
    hipLaunchKernelGGL(k_merge_old_keys, dim3(nblk(n)), dim3(256), 0, st, map->xyz.as<float>(), n, g, old_keys.as<uint32_t>(), bad);
    // pending points: keys, stable sort, voxel heads
    uint32_t *nk = keys.as<uint32_t>(), *nk2 = nk + m, *nv = vals.as<uint32_t>(), *nv2 = nv + m;
    hipLaunchKernelGGL((k_pcl_keys<uint32_t, int32_t>), dim3(nblk(m)), dim3(256), 0, st, pending->xyz.as<float>(), m, g, nk, nv, map->vox_point_ids.as<int32_t>());
    uint32_t *sk = nullptr, *sv = nullptr;
    SF_TRY(sf::radix_sort_pairs<uint32_t>(ctx, nk, nk2, nv, nv2, m, end_bit, &sk, &sv));
    int64_t n_groups = 0;
    uint32_t *h_flag = reinterpret_cast<uint32_t *>(ctx->h_pinned) + 8;
    SF_HIP(hipMemcpyAsync(h_flag, bad, sizeof(uint32_t), hipMemcpyDeviceToHost, st)); // (read back by the synchronisation inside scan_heads)
    SF_TRY(scan_heads<uint32_t>(ctx, sk, b.n_finite, flags.as<uint32_t>(), pos.as<uint32_t>(), &n_groups));
    if (*h_flag) return full_path();
    hipLaunchKernelGGL(k_merge_groups, dim3(nblk(b.n_finite)), dim3(256), 0, st, map->xyz.as<float>(), old_keys.as<uint32_t>(), n, pending->xyz.as<float>(), sk, sv, flags.as<uint32_t>(),
                       pos.as<uint32_t>(), b.n_finite, g_rank, g_fresh, g_centroid, g_old);
    SF_TRY(sf::scan_u32<0>(ctx, g_fresh, fresh_pos, n_groups));
    uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned);
    MergeExtremes *h_ext = reinterpret_cast<MergeExtremes *>(static_cast<unsigned char *>(ctx->h_pinned) + 128);
    for (int d = 0; d < 3; ++d) { h_ext->mn[d] = INT32_MAX; h_ext->mx[d] = INT32_MIN; }
    h_ext->touched_extreme = h_ext->pad = 0u;
    SF_HIP(hipMemcpyAsync(d_ext, h_ext, sizeof(MergeExtremes), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_merge_extremes, dim3((unsigned)std::min<int64_t>(64, nblk(n_groups))), dim3(256), 0, st, g_fresh, g_centroid, g_old, n_groups, a.mn[0], a.mn[1], a.mn[2],
                       a.mx[0], a.mx[1], a.mx[2], d_ext);
    SF_HIP(hipMemcpyAsync(h, fresh_pos + (n_groups - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SF_HIP(hipMemcpyAsync(h + 1, g_fresh + (n_groups - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SF_HIP(hipMemcpyAsync(h_ext, d_ext, sizeof(MergeExtremes), hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    const int64_t n_fresh = (int64_t)h[0] + (int64_t)h[1], n_out = n + n_fresh;
    sf::DevBuf &out = map->spare;
    SF_TRY(out.reserve(sizeof(float) * 3 * (size_t)n_out));
    hipLaunchKernelGGL(k_merge_fresh_ranks, dim3(nblk(n_groups)), dim3(256), 0, st, g_rank, g_fresh, fresh_pos, n_groups, fresh_rank);
    hipLaunchKernelGGL(k_merge_coarse, dim3(nblk(n_coarse)), dim3(256), 0, st, fresh_rank, n_fresh, n_coarse, coarse);
    hipLaunchKernelGGL(k_merge_copy, dim3(nblk(n)), dim3(256), 0, st, map->xyz.as<float>(), n, fresh_rank, coarse, out.as<float>());
    hipLaunchKernelGGL(k_merge_place, dim3(nblk(n_groups)), dim3(256), 0, st, g_rank, g_fresh, fresh_pos, g_centroid, n_groups, out.as<float>());
    SF_HIP(hipGetLastError());
    map->xyz.swap(out);
    map->n = n_out;
    map->n_last_idx = -1;
    map->n_vox_point_vals = map->n_vox_out_vals = map->n_vox_out_pts = 0; // (per-point / per-voxel ids are those of a full pass only)
    // what sf_map_patch needs to carry an index of the old map over to the new one
    sf_cloud::MergeRecord &rec = map->merge;
    rec.valid = true;
    rec.epoch = ctx->merge_epoch;
    rec.stamp_before = stamp_before;
    rec.stamp_after = map->stamp;
    rec.n_old = n; rec.n_groups = n_groups; rec.n_fresh = n_fresh;
    rec.g_rank = g_rank; rec.g_fresh = g_fresh; rec.fresh_pos = fresh_pos; rec.fresh_rank = fresh_rank; rec.g_centroid = g_centroid; rec.g_old = g_old; rec.coarse = coarse;
    rec.touched_extreme = h_ext->touched_extreme != 0;
    for (int d = 0; d < 3; ++d) {
        rec.old_mn[d] = a.mn[d]; rec.old_mx[d] = a.mx[d];
        rec.cen_mn[d] = float_from_ordered(h_ext->mn[d]); rec.cen_mx[d] = float_from_ordered(h_ext->mx[d]);
    }
    if (!rec.touched_extreme) { // every old point that held a bound is still there: the bounds of the merged map are known
        for (int d = 0; d < 3; ++d) { map->bounds_mn[d] = std::min(a.mn[d], rec.cen_mn[d]); map->bounds_mx[d] = std::max(a.mx[d], rec.cen_mx[d]); }
        map->bounds_stamp = map->stamp;
    }
    if (merged) *merged = 1;
    return SF_OK;
}

extern "C" int sf_cloud_voxel_downsample(sf_cloud *c, double leaf, int flavour, int *status_flags)
{
    SF_CHECK(c && leaf > 0, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(flavour == SF_VOXEL_PCL || flavour == SF_VOXEL_O3D || flavour == SF_VOXEL_PCL64, SF_ERR_INVALID, "unknown voxel flavour %d", flavour);
    SF_HIP(hipSetDevice(c->ctx->device));
    if (status_flags) *status_flags = 0;
    sf::cloud_touch(c);
    c->n_vox_point_vals = c->n_vox_out_vals = c->n_vox_out_pts = 0;
    if (c->n == 0) return SF_OK;
    if (flavour == SF_VOXEL_PCL) return voxel_pcl<uint32_t, int32_t>(c, (float)leaf, status_flags);
    if (flavour == SF_VOXEL_PCL64) return voxel_pcl<uint64_t, int64_t>(c, (float)leaf, status_flags);
    return voxel_o3d(c, leaf);
}

extern "C" int sf_cloud_voxel_point_ids(sf_cloud *c, int32_t *ids, int64_t cap_values, int64_t *n_values)
{
    SF_CHECK(c, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(!c->vox_wide, SF_ERR_STATE, "the last downsample used 64-bit indices: sf_cloud_voxel_point_ids64");
    return download_i32(c, c->vox_point_ids, c->n_vox_point_vals, ids, cap_values, n_values);
}

extern "C" int sf_cloud_voxel_out_ids(sf_cloud *c, int32_t *ids, int64_t cap_values, int64_t *n_values)
{
    SF_CHECK(c, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(!c->vox_wide, SF_ERR_STATE, "the last downsample used 64-bit indices: sf_cloud_voxel_out_ids64");
    return download_i32(c, c->vox_out_ids, c->n_vox_out_vals, ids, cap_values, n_values);
}

namespace {
int download_i64(sf_cloud *c, const sf::DevBuf &buf, int64_t have, int64_t *dst, int64_t cap, int64_t *n)
{
    if (n) *n = have;
    SF_CHECK(cap >= have && (dst || have == 0), SF_ERR_INVALID, "buffer too small: %lld < %lld", (long long)cap, (long long)have);
    if (have > 0) {
        SF_HIP(hipMemcpyAsync(dst, buf.p, sizeof(int64_t) * (size_t)have, hipMemcpyDeviceToHost, c->ctx->stream));
        SF_HIP(hipStreamSynchronize(c->ctx->stream));
    }
    return SF_OK;
}
} // namespace

extern "C" int sf_cloud_voxel_point_ids64(sf_cloud *c, int64_t *ids, int64_t cap_values, int64_t *n_values)
{
    SF_CHECK(c, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(c->vox_wide || c->n_vox_point_vals == 0, SF_ERR_STATE, "the last downsample used 32-bit indices");
    return download_i64(c, c->vox_point_ids, c->n_vox_point_vals, ids, cap_values, n_values);
}

extern "C" int sf_cloud_voxel_out_ids64(sf_cloud *c, int64_t *ids, int64_t cap_values, int64_t *n_values)
{
    SF_CHECK(c, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(c->vox_wide || c->n_vox_out_vals == 0, SF_ERR_STATE, "the last downsample used 32-bit indices");
    return download_i64(c, c->vox_out_ids, c->n_vox_out_vals, ids, cap_values, n_values);
}

extern "C" int sf_cloud_voxel_out_means_f64(sf_cloud *c, double *xyz, int64_t cap_points, int64_t *n_points)
{
    SF_CHECK(c, SF_ERR_INVALID, "bad arguments");
    if (n_points) *n_points = c->n_vox_out_pts;
    SF_CHECK(cap_points >= c->n_vox_out_pts && (xyz || c->n_vox_out_pts == 0), SF_ERR_INVALID, "buffer too small");
    if (c->n_vox_out_pts > 0) {
        SF_HIP(hipMemcpyAsync(xyz, c->vox_out_means.p, sizeof(double) * 3 * (size_t)c->n_vox_out_pts, hipMemcpyDeviceToHost, c->ctx->stream));
        SF_HIP(hipStreamSynchronize(c->ctx->stream));
    }
    return SF_OK;
}
