/* voxel.c — TEST INFRASTRUCTURE (oracle); see sf_oracle.h.
 *
 * orc_voxel_pcl restates pcl::VoxelGrid<PointXYZ>::applyFilter (PCL 1.12,
 * filters/impl/voxel_grid.hpp; un-vendored, version unpinned) as the reference calls it at
 * localization/src/global_map_frames_manager.cpp:142-146 with leaf 0.1f
 * (localization/src/localization_node.cpp:19).
 * orc_voxel_o3d restates open3d::geometry::PointCloud::VoxelDownSample (Open3D >= 0.13,
 * cpp/open3d/geometry/PointCloud.cpp; un-vendored, unpinned) as called at
 * localization_python/localization_python/localization_node.py:47.
 */
#include "sf_oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t vox; int pt; } vp_t;
static int vp_cmp(const void *a, const void *b)
{
    const vp_t *x = (const vp_t *)a, *y = (const vp_t *)b;
    if (x->vox != y->vox) return x->vox < y->vox ? -1 : 1;
    return (x->pt > y->pt) - (x->pt < y->pt); /* stable: ascending point index */
}

int orc_voxel_pcl(const float *xyz, int n, float leaf, float *out, int32_t *vox_idx,
                  int32_t *out_vox)
{
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf; /* inverse_leaf_size_ = Ones / leaf_size_ */
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int finite_cnt = 0;
    for (int i = 0; i < n; ++i) { /* getMinMax3D, non-finite skipped */
        const float *p = xyz + 3 * (size_t)i;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        ++finite_cnt;
        for (int d = 0; d < 3; ++d) {
            if (p[d] < mn[d]) mn[d] = p[d];
            if (p[d] > mx[d]) mx[d] = p[d];
        }
    }
    if (finite_cnt == 0) return 0;
    int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
    int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
    int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) { /* "Leaf size is too small" -> output = input */
        memcpy(out, xyz, sizeof(float) * 3 * (size_t)n);
        if (vox_idx) for (int i = 0; i < n; ++i) vox_idx[i] = -1;
        return -1;
    }
    int min_b[3], max_b[3], div_b[3], mul[3];
    for (int d = 0; d < 3; ++d) {
        min_b[d] = (int)floorf(mn[d] * inv);
        max_b[d] = (int)floorf(mx[d] * inv);
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];

    vp_t *v = (vp_t *)malloc(sizeof(vp_t) * (size_t)finite_cnt);
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) {
            if (vox_idx) vox_idx[i] = -1;
            continue;
        }
        int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        int idx = i0 * mul[0] + i1 * mul[1] + i2 * mul[2];
        if (vox_idx) vox_idx[i] = idx;
        v[k].vox = (uint32_t)idx;
        v[k].pt = i;
        ++k;
    }
    qsort(v, (size_t)k, sizeof(vp_t), vp_cmp);
    int nout = 0;
    for (int first = 0; first < k;) {
        int last = first + 1;
        while (last < k && v[last].vox == v[first].vox) ++last;
        float sx = 0.f, sy = 0.f, sz = 0.f; /* AccumulatorXYZ: Vector3f += */
        for (int j = first; j < last; ++j) {
            const float *p = xyz + 3 * (size_t)v[j].pt;
            sx += p[0]; sy += p[1]; sz += p[2];
        }
        float cnt = (float)(last - first);
        out[3 * (size_t)nout + 0] = sx / cnt;
        out[3 * (size_t)nout + 1] = sy / cnt;
        out[3 * (size_t)nout + 2] = sz / cnt;
        if (out_vox) out_vox[nout] = (int32_t)v[first].vox;
        ++nout;
        first = last;
    }
    free(v);
    return nout;
}

typedef struct { int32_t i, j, k; int pt; } ijk_t;
static int ijk_cmp(const void *a, const void *b)
{
    const ijk_t *x = (const ijk_t *)a, *y = (const ijk_t *)b;
    if (x->i != y->i) return x->i < y->i ? -1 : 1;
    if (x->j != y->j) return x->j < y->j ? -1 : 1;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    return (x->pt > y->pt) - (x->pt < y->pt);
}

int orc_voxel_o3d(const double *xyz, int n, double voxel, double *out, int32_t *ijk,
                  int32_t *out_ijk)
{
    if (n <= 0 || !(voxel > 0.0)) return n <= 0 ? 0 : -2;
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            double v = xyz[3 * (size_t)i + d];
            if (!isfinite(v)) return -2;
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    double vmin[3], vmax[3], span = 0;
    for (int d = 0; d < 3; ++d) {
        vmin[d] = mn[d] - voxel * 0.5;
        vmax[d] = mx[d] + voxel * 0.5;
        if (vmax[d] - vmin[d] > span) span = vmax[d] - vmin[d];
    }
    if (voxel * (double)INT_MAX < span) return -1; /* "voxel_size is too small." */
    ijk_t *v = (ijk_t *)malloc(sizeof(ijk_t) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const double *p = xyz + 3 * (size_t)i;
        v[i].i = (int32_t)floor((p[0] - vmin[0]) / voxel);
        v[i].j = (int32_t)floor((p[1] - vmin[1]) / voxel);
        v[i].k = (int32_t)floor((p[2] - vmin[2]) / voxel);
        v[i].pt = i;
        if (ijk) { ijk[3 * (size_t)i] = v[i].i; ijk[3 * (size_t)i + 1] = v[i].j; ijk[3 * (size_t)i + 2] = v[i].k; }
    }
    qsort(v, (size_t)n, sizeof(ijk_t), ijk_cmp);
    int nout = 0;
    for (int first = 0; first < n;) {
        int last = first + 1;
        while (last < n && v[last].i == v[first].i && v[last].j == v[first].j && v[last].k == v[first].k) ++last;
        double sx = 0, sy = 0, sz = 0; /* AccumulatedPoint: point_ += in index order */
        for (int j = first; j < last; ++j) {
            const double *p = xyz + 3 * (size_t)v[j].pt;
            sx += p[0]; sy += p[1]; sz += p[2];
        }
        double cnt = (double)(last - first);
        out[3 * (size_t)nout + 0] = sx / cnt;
        out[3 * (size_t)nout + 1] = sy / cnt;
        out[3 * (size_t)nout + 2] = sz / cnt;
        if (out_ijk) { out_ijk[3 * (size_t)nout] = v[first].i; out_ijk[3 * (size_t)nout + 1] = v[first].j; out_ijk[3 * (size_t)nout + 2] = v[first].k; }
        ++nout;
        first = last;
    }
    free(v);
    return nout;
}

/* orc_voxel_pcl64 — the SAME arithmetic as orc_voxel_pcl (float32 inverse leaf, floorf, min_b
 * offsets, float32 centroid sums in ascending point index) with the linear voxel index kept in
 * int64, so the "Leaf size is too small ... Integer indices would overflow" early return of
 * pcl::VoxelGrid never triggers.  No reference counterpart (the reference's PCL returns its input
 * unchanged there, global_map_frames_manager.cpp:142-146): a documented extension for maps past
 * 2^31 voxels (BASELINE config 5); on inputs that do not overflow it equals orc_voxel_pcl. */
typedef struct { int64_t vox; int pt; } vp64_t;
static int vp64_cmp(const void *a, const void *b)
{
    const vp64_t *x = (const vp64_t *)a, *y = (const vp64_t *)b;
    if (x->vox != y->vox) return x->vox < y->vox ? -1 : 1;
    return (x->pt > y->pt) - (x->pt < y->pt);
}

int orc_voxel_pcl64(const float *xyz, int n, float leaf, float *out, int64_t *vox_idx, int64_t *out_vox)
{
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int finite_cnt = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        ++finite_cnt;
        for (int d = 0; d < 3; ++d) {
            if (p[d] < mn[d]) mn[d] = p[d];
            if (p[d] > mx[d]) mx[d] = p[d];
        }
    }
    if (finite_cnt == 0) return 0;
    int min_b[3], max_b[3];
    int64_t div_b[3], mul[3];
    for (int d = 0; d < 3; ++d) {
        min_b[d] = (int)floorf(mn[d] * inv);
        max_b[d] = (int)floorf(mx[d] * inv);
        div_b[d] = (int64_t)max_b[d] - (int64_t)min_b[d] + 1;
    }
    mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];
    vp64_t *v = (vp64_t *)malloc(sizeof(vp64_t) * (size_t)finite_cnt);
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) {
            if (vox_idx) vox_idx[i] = -1;
            continue;
        }
        int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        int64_t idx = (int64_t)i0 * mul[0] + (int64_t)i1 * mul[1] + (int64_t)i2 * mul[2];
        if (vox_idx) vox_idx[i] = idx;
        v[k].vox = idx;
        v[k].pt = i;
        ++k;
    }
    qsort(v, (size_t)k, sizeof(vp64_t), vp64_cmp);
    int nout = 0;
    for (int first = 0; first < k;) {
        int last = first + 1;
        while (last < k && v[last].vox == v[first].vox) ++last;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int j = first; j < last; ++j) {
            const float *p = xyz + 3 * (size_t)v[j].pt;
            sx += p[0]; sy += p[1]; sz += p[2];
        }
        float cnt = (float)(last - first);
        out[3 * (size_t)nout + 0] = sx / cnt;
        out[3 * (size_t)nout + 1] = sy / cnt;
        out[3 * (size_t)nout + 2] = sz / cnt;
        if (out_vox) out_vox[nout] = v[first].vox;
        ++nout;
        first = last;
    }
    free(v);
    return nout;
}
