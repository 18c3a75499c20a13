/* crops.c — TEST INFRASTRUCTURE (oracle); see sf_oracle.h.
 * Restates localization/include/localization/point_cloud_processing.hpp:31-92 and the two
 * crops of localization_python/localization_python/localization_node.py:105-115,222-225. */
#include "sf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* point_cloud_processing.hpp:55-74 — no-op when size < step, else indices 0,step,2step.. */
int orc_uniform_subsample(const float *xyz, int n, int step, float *out)
{
    if (step <= 0 || n < step) {
        memcpy(out, xyz, sizeof(float) * 3 * (size_t)n);
        return n;
    }
    int k = 0;
    for (size_t i = 0; i < (size_t)n; i += (size_t)step) {
        out[3 * (size_t)k + 0] = xyz[3 * i + 0];
        out[3 * (size_t)k + 1] = xyz[3 * i + 1];
        out[3 * (size_t)k + 2] = xyz[3 * i + 2];
        ++k;
    }
    return k;
}

typedef struct { float d2; int idx; } di_t;
static int di_cmp(const void *a, const void *b)
{
    const di_t *x = (const di_t *)a, *y = (const di_t *)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* point_cloud_processing.hpp:31-53.  pcl::search::KdTree::radiusSearch -> FLANN radius
 * search with radius^2 as float, strict "<", results sorted by (distance, index). */
int orc_crop_radius(const float *xyz, int n, const float center[3], double radius, float *out,
                    int *out_idx)
{
    const float r2 = (float)(radius * radius);
    di_t *hits = (di_t *)malloc(sizeof(di_t) * (size_t)(n > 0 ? n : 1));
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        float acc = 0.f, diff;
        diff = center[0] - p[0]; acc += diff * diff;
        diff = center[1] - p[1]; acc += diff * diff;
        diff = center[2] - p[2]; acc += diff * diff;
        if (acc < r2) { hits[k].d2 = acc; hits[k].idx = i; ++k; }
    }
    qsort(hits, (size_t)k, sizeof(di_t), di_cmp);
    for (int j = 0; j < k; ++j) {
        memcpy(out + 3 * (size_t)j, xyz + 3 * (size_t)hits[j].idx, sizeof(float) * 3);
        if (out_idx) out_idx[j] = hits[j].idx;
    }
    free(hits);
    return k;
}

/* point_cloud_processing.hpp:76-92 */
int orc_remove_floor(const float *xyz, int n, float *out)
{
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (p[2] > 0) { memcpy(out + 3 * (size_t)k, p, sizeof(float) * 3); ++k; }
    }
    return k;
}

/* localization_node.py:105-115 — skip_nans=True then inclusive bounds in float64 */
int orc_crop_aabb(const float *xyz, int n, const double lo[3], const double hi[3], float *out,
                  int *out_idx)
{
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (isnan(p[0]) || isnan(p[1]) || isnan(p[2])) continue;
        double x = p[0], y = p[1], z = p[2];
        if (x >= lo[0] && x <= hi[0] && y >= lo[1] && y <= hi[1] && z >= lo[2] && z <= hi[2]) {
            memcpy(out + 3 * (size_t)k, p, sizeof(float) * 3);
            if (out_idx) out_idx[k] = i;
            ++k;
        }
    }
    return k;
}

/* localization_node.py:222-225 — Open3D OrientedBoundingBox::GetPointIndicesWithinBoundingBox:
 * d = p - center; |d . R.col(k)| <= extent[k] / 2 for k = 0,1,2 (float64, inclusive). */
int orc_crop_obb(const float *xyz, int n, const double center[3], const double R[9],
                 const double extent[3], float *out, int *out_idx)
{
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        double d0 = (double)p[0] - center[0], d1 = (double)p[1] - center[1],
               d2 = (double)p[2] - center[2];
        int inside = 1;
        for (int c = 0; c < 3; ++c) {
            double proj = d0 * R[0 * 3 + c] + d1 * R[1 * 3 + c] + d2 * R[2 * 3 + c];
            if (!(fabs(proj) <= extent[c] / 2)) inside = 0;
        }
        if (inside) {
            memcpy(out + 3 * (size_t)k, p, sizeof(float) * 3);
            if (out_idx) out_idx[k] = i;
            ++k;
        }
    }
    return k;
}
