/* icp.c — TEST INFRASTRUCTURE (oracle); see sf_oracle.h.
 * ICP drivers: ref_cpp (reference C++), o3d_p2p (reference Python, via Open3D's published
 * registration_icp), p2plane (extension), and radius PCA normals (extension). */
#include "sf_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_kabsch_f_(const float *src, const float *tgt, int n, float T[16]);
void orc_kabsch_d_(const double *src, const double *tgt, int n, double T[16]);

/* ---------------------------------------------------------------- ref_cpp, two arithmetics */
#define REAL float
#define SUF f
#define KDTREE orc_kdtree_f
#define KDBUILD orc_kdtree_f_build
#define KDFREE orc_kdtree_f_free
#define KDNN orc_kdtree_f_nn
#define KABSCH orc_kabsch_f_
#define SQRT sqrtf
#define FABS fabsf
#define REAL_MAX_ FLT_MAX
#include "icp_ref_impl.inc"
#undef REAL
#undef SUF
#undef KDTREE
#undef KDBUILD
#undef KDFREE
#undef KDNN
#undef KABSCH
#undef SQRT
#undef FABS
#undef REAL_MAX_

#define REAL double
#define SUF d
#define KDTREE orc_kdtree_d
#define KDBUILD orc_kdtree_d_build
#define KDFREE orc_kdtree_d_free
#define KDNN orc_kdtree_d_nn
#define KABSCH orc_kabsch_d_
#define SQRT sqrt
#define FABS fabs
#define REAL_MAX_ DBL_MAX
#include "icp_ref_impl.inc"
#undef REAL
#undef SUF
#undef KDTREE
#undef KDBUILD
#undef KDFREE
#undef KDNN
#undef KABSCH
#undef SQRT
#undef FABS
#undef REAL_MAX_

static double *to_double(const float *a, size_t cnt)
{
    double *d = (double *)malloc(sizeof(double) * (cnt ? cnt : 1));
    for (size_t i = 0; i < cnt; ++i) d[i] = a[i];
    return d;
}

int orc_icp_ref_cpp(const float *src, int n, const float *tgt, int m, const float init[16],
                    float max_corr_dist, int num_iters, float accept_err, float eps,
                    int precise, orc_icp_result *out)
{
    if (!precise) return icp_ref_f(src, n, tgt, m, init, max_corr_dist, num_iters, accept_err, eps, out);
    double *s = to_double(src, 3 * (size_t)n), *t = to_double(tgt, 3 * (size_t)m), i0[16];
    for (int i = 0; i < 16; ++i) i0[i] = init[i];
    /* the reference stores last_error_ as float: FLT_MAX start is irrelevant for doubles */
    int rc = icp_ref_d(s, n, t, m, i0, max_corr_dist, num_iters, accept_err, eps, out);
    free(s);
    free(t);
    return rc;
}

/* ---------------------------------------------------------------- helpers (float64) */
static void mat4_mul_d(const double A[16], const double B[16], double C[16])
{
    double R[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            R[4 * r + c] = A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c] + A[4 * r + 2] * B[8 + c] + A[4 * r + 3] * B[12 + c];
    memcpy(C, R, sizeof(R));
}

static void transform_d(const double T[16], const double *in, double *outp, int n)
{
    for (int i = 0; i < n; ++i) {
        const double *p = in + 3 * (size_t)i;
        double x = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
        double y = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
        double z = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
        outp[3 * (size_t)i] = x; outp[3 * (size_t)i + 1] = y; outp[3 * (size_t)i + 2] = z;
    }
}

/* open3d GetRegistrationResultAndCorrespondences: hybrid search, knn 1, d2 < r^2 */
static void o3d_eval(const orc_kdtree_d *tree, const double *pcd, int n, double max_dist,
                     int *corr, int *n_corr, double *fitness, double *rmse)
{
    double err2 = 0, r2 = max_dist * max_dist;
    int k = 0;
    for (int i = 0; i < n; ++i) {
        int idx;
        double d2;
        orc_kdtree_d_nn(tree, pcd + 3 * (size_t)i, 1, &idx, &d2);
        if (idx >= 0 && d2 < r2) { corr[i] = idx; err2 += d2; ++k; }
        else corr[i] = -1;
    }
    *n_corr = k;
    if (k == 0) { *fitness = 0; *rmse = 0; }
    else { *fitness = (double)k / (double)n; *rmse = sqrt(err2 / (double)k); }
}

/* localization_node.py:232-237 -> open3d::pipelines::registration::RegistrationICP with
 * TransformationEstimationPointToPoint (Eigen::umeyama, no scaling == Kabsch) and
 * ICPConvergenceCriteria(relative_fitness 1e-6, relative_rmse 1e-6, max_iteration).   */
int orc_icp_o3d_p2p(const float *src, int n, const float *tgt, int m, const double init[16],
                    double max_dist, int max_iter, orc_icp_result *out)
{
    double *t = to_double(tgt, 3 * (size_t)m), *s0 = to_double(src, 3 * (size_t)n);
    double *pcd = (double *)malloc(sizeof(double) * 3 * (size_t)(n ? n : 1));
    double *a = (double *)malloc(sizeof(double) * 3 * (size_t)(n ? n : 1));
    double *b = (double *)malloc(sizeof(double) * 3 * (size_t)(n ? n : 1));
    int *corr = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    orc_kdtree_d *tree = orc_kdtree_d_build(t, m, 15);
    double T[16];
    memcpy(T, init, sizeof(T));
    transform_d(T, s0, pcd, n);
    int ncorr;
    double fit, rmse;
    o3d_eval(tree, pcd, n, max_dist, corr, &ncorr, &fit, &rmse);
    int it = 0, conv = 0;
    for (int i = 0; i < max_iter; ++i) {
        double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (ncorr > 0) {
            int k = 0;
            for (int j = 0; j < n; ++j)
                if (corr[j] >= 0) {
                    memcpy(a + 3 * (size_t)k, pcd + 3 * (size_t)j, sizeof(double) * 3);
                    memcpy(b + 3 * (size_t)k, t + 3 * (size_t)corr[j], sizeof(double) * 3);
                    ++k;
                }
            orc_kabsch_d_(a, b, k, upd);
        }
        mat4_mul_d(upd, T, T);
        transform_d(upd, pcd, pcd, n);
        ++it;
        double pf = fit, pr = rmse;
        o3d_eval(tree, pcd, n, max_dist, corr, &ncorr, &fit, &rmse);
        if (fabs(pf - fit) < 1e-6 && fabs(pr - rmse) < 1e-6) { conv = 1; break; }
    }
    memcpy(out->T, T, sizeof(T));
    out->error = rmse; out->fitness = fit; out->iterations = it; out->converged = conv;
    out->n_corr = ncorr; out->n_research = it + 1;
    orc_kdtree_d_free(tree);
    free(t); free(s0); free(pcd); free(a); free(b); free(corr);
    return 0;
}

/* ---------------------------------------------------------------- point-to-plane GN */
/* in-place LDL^T solve of the symmetric 6x6 system A x = b; returns 0 on success */
static int ldlt6(double A[36], double b[6], double x[6])
{
    double L[36] = {0}, D[6];
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        if (!(fabs(d) > 0) || !isfinite(d)) return -1;
        D[j] = d;
        L[6 * j + j] = 1;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[6 * i + j];
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = v / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k]; y[i] = v; }
    for (int i = 0; i < 6; ++i) y[i] /= D[i];
    for (int i = 5; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * x[k]; x[i] = v; }
    return 0;
}

/* Open3D TransformVector6dToMatrix4d: R = Rz(v[2]) * Ry(v[1]) * Rx(v[0]), t = v[3:6] */
static void vec6_to_mat4(const double v[6], double T[16])
{
    double ca = cos(v[0]), sa = sin(v[0]), cb = cos(v[1]), sb = sin(v[1]), cg = cos(v[2]), sg = sin(v[2]);
    double R[9] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa,
                   sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa,
                   -sb, cb * sa, cb * ca};
    for (int i = 0; i < 16; ++i) T[i] = 0;
    T[15] = 1;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c]; T[4 * r + 3] = v[3 + r]; }
}

int orc_icp_p2plane(const float *src, int n, const float *tgt, const float *tgt_normals, int m,
                    const double init[16], double max_dist, int num_iters, orc_icp_result *out)
{
    double *t = to_double(tgt, 3 * (size_t)m), *s0 = to_double(src, 3 * (size_t)n);
    double *pcd = (double *)malloc(sizeof(double) * 3 * (size_t)(n ? n : 1));
    int *corr = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    orc_kdtree_d *tree = orc_kdtree_d_build(t, m, 15);
    double T[16];
    memcpy(T, init, sizeof(T));
    int it = 0, ncorr = 0;
    double fit = 0, rmse = 0;
    for (int i = 0; i < num_iters; ++i) {
        transform_d(T, s0, pcd, n);
        o3d_eval(tree, pcd, n, max_dist, corr, &ncorr, &fit, &rmse);
        double A[36] = {0}, g[6] = {0};
        for (int j = 0; j < n; ++j) {
            if (corr[j] < 0) continue;
            const double *s = pcd + 3 * (size_t)j, *q = t + 3 * (size_t)corr[j];
            const float *nf = tgt_normals + 3 * (size_t)corr[j];
            double nx = nf[0], ny = nf[1], nz = nf[2];
            double r = (s[0] - q[0]) * nx + (s[1] - q[1]) * ny + (s[2] - q[2]) * nz;
            double J[6] = {s[1] * nz - s[2] * ny, s[2] * nx - s[0] * nz, s[0] * ny - s[1] * nx, nx, ny, nz};
            for (int a = 0; a < 6; ++a) { g[a] += J[a] * r; for (int b = 0; b < 6; ++b) A[6 * a + b] += J[a] * J[b]; }
        }
        double rhs[6], x[6];
        for (int a = 0; a < 6; ++a) rhs[a] = -g[a];
        if (ncorr < 6 || ldlt6(A, rhs, x) != 0) break;
        double upd[16];
        vec6_to_mat4(x, upd);
        mat4_mul_d(upd, T, T);
        ++it;
    }
    memcpy(out->T, T, sizeof(T));
    out->error = rmse; out->fitness = fit; out->iterations = it; out->converged = (it == num_iters);
    out->n_corr = ncorr; out->n_research = it;
    orc_kdtree_d_free(tree);
    free(t); free(s0); free(pcd); free(corr);
    return 0;
}

/* ---------------------------------------------------------------- radius PCA normals */
/* symmetric 3x3 eigen decomposition (cyclic Jacobi); returns eigenvector of the
 * smallest eigenvalue */
static void smallest_eigvec(const double C[9], double nrm[3])
{
    double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(a, C, sizeof(a));
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
        double dia = fabs(a[0]) + fabs(a[4]) + fabs(a[8]);
        if (off <= 1e-300 || off <= DBL_EPSILON * dia * 1e-3) break;
        static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
        for (int k = 0; k < 3; ++k) {
            int p = P[k], q = Q[k];
            double apq = a[3 * p + q];
            if (fabs(apq) < 1e-300) continue;
            double theta = (a[3 * q + q] - a[3 * p + p]) / (2 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
            double c = 1 / sqrt(t * t + 1), s = t * c;
            for (int i = 0; i < 3; ++i) { /* A <- A J */
                double aip = a[3 * i + p], aiq = a[3 * i + q];
                a[3 * i + p] = c * aip - s * aiq;
                a[3 * i + q] = s * aip + c * aiq;
            }
            for (int i = 0; i < 3; ++i) { /* A <- J^T A */
                double api = a[3 * p + i], aqi = a[3 * q + i];
                a[3 * p + i] = c * api - s * aqi;
                a[3 * q + i] = s * api + c * aqi;
            }
            for (int i = 0; i < 3; ++i) {
                double vip = v[3 * i + p], viq = v[3 * i + q];
                v[3 * i + p] = c * vip - s * viq;
                v[3 * i + q] = s * vip + c * viq;
            }
        }
    }
    int best = 0;
    if (a[4] < a[3 * best + best]) best = 1;
    if (a[8] < a[3 * best + best]) best = 2;
    double x = v[best], y = v[3 + best], z = v[6 + best];
    double nn = sqrt(x * x + y * y + z * z);
    if (!(nn > 0)) { nrm[0] = 0; nrm[1] = 0; nrm[2] = 1; return; }
    x /= nn; y /= nn; z /= nn;
    if (z < 0 || (z == 0 && (y < 0 || (y == 0 && x < 0)))) { x = -x; y = -y; z = -z; }
    nrm[0] = x; nrm[1] = y; nrm[2] = z;
}

typedef struct { int cell; int pt; } cp_t;
static int cp_cmp(const void *a, const void *b)
{
    const cp_t *x = (const cp_t *)a, *y = (const cp_t *)b;
    if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
    return (x->pt > y->pt) - (x->pt < y->pt);
}

static void normals_impl(const float *xyz, int n, double radius, float *normals, int *n_neighbors, double *cov6)
{
    if (n <= 0) return;
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            double v = xyz[3 * (size_t)i + d];
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    double h = radius;
    int dim[3];
    for (;;) {
        double cells = 1;
        for (int d = 0; d < 3; ++d) { dim[d] = (int)floor((mx[d] - mn[d]) / h) + 1; cells *= dim[d]; }
        if (cells <= 6.4e7) break;
        h *= 2;
    }
    int ncell = dim[0] * dim[1] * dim[2];
    cp_t *cp = (cp_t *)malloc(sizeof(cp_t) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        int c[3];
        for (int d = 0; d < 3; ++d) {
            c[d] = (int)floor((xyz[3 * (size_t)i + d] - mn[d]) / h);
            if (c[d] >= dim[d]) c[d] = dim[d] - 1;
        }
        cp[i].cell = (c[2] * dim[1] + c[1]) * dim[0] + c[0];
        cp[i].pt = i;
    }
    qsort(cp, (size_t)n, sizeof(cp_t), cp_cmp);
    int *start = (int *)calloc((size_t)ncell + 1, sizeof(int));
    for (int i = 0; i < n; ++i) start[cp[i].cell + 1]++;
    for (int c = 0; c < ncell; ++c) start[c + 1] += start[c];
    const double r2 = radius * radius;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        int c[3];
        for (int d = 0; d < 3; ++d) { c[d] = (int)floor((p[d] - mn[d]) / h); if (c[d] >= dim[d]) c[d] = dim[d] - 1; }
        double sum[3] = {0, 0, 0};
        int cnt = 0;
        for (int pass = 0; pass < 2; ++pass) {
            double mean[3] = {0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0};
            if (pass == 1) { if (cnt < 3) break; for (int d = 0; d < 3; ++d) mean[d] = sum[d] / cnt; }
            for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                int cx = c[0] + dx, cy = c[1] + dy, cz = c[2] + dz;
                if (cx < 0 || cy < 0 || cz < 0 || cx >= dim[0] || cy >= dim[1] || cz >= dim[2]) continue;
                int cell = (cz * dim[1] + cy) * dim[0] + cx;
                for (int j = start[cell]; j < start[cell + 1]; ++j) {
                    const float *q = xyz + 3 * (size_t)cp[j].pt;
                    double ex = (double)q[0] - p[0], ey = (double)q[1] - p[1], ez = (double)q[2] - p[2];
                    if (!(ex * ex + ey * ey + ez * ez <= r2)) continue;
                    if (pass == 0) { sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2]; ++cnt; }
                    else {
                        double ax = q[0] - mean[0], ay = q[1] - mean[1], az = q[2] - mean[2];
                        C[0] += ax * ax; C[1] += ax * ay; C[2] += ax * az; C[3] += ay * ay; C[4] += ay * az; C[5] += az * az;
                    }
                }
            }
            if (pass == 1) {
                double M[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]}, nv[3];
                smallest_eigvec(M, nv);
                normals[3 * (size_t)i] = (float)nv[0]; normals[3 * (size_t)i + 1] = (float)nv[1]; normals[3 * (size_t)i + 2] = (float)nv[2];
                if (cov6) for (int d = 0; d < 6; ++d) cov6[6 * (size_t)i + d] = C[d] / cnt; /* xx xy xz yy yz zz, normalised by the neighbour count */
            }
        }
        if (cnt < 3) {
            normals[3 * (size_t)i] = 0; normals[3 * (size_t)i + 1] = 0; normals[3 * (size_t)i + 2] = 1;
            if (cov6) for (int d = 0; d < 6; ++d) cov6[6 * (size_t)i + d] = 0.0;
        }
        if (n_neighbors) n_neighbors[i] = cnt;
    }
    free(cp);
    free(start);
}

void orc_normals_radius(const float *xyz, int n, double radius, float *normals, int *n_neighbors)
{
    normals_impl(xyz, n, radius, normals, n_neighbors, NULL);
}

void orc_normals_radius_cov(const float *xyz, int n, double radius, float *normals, int *n_neighbors, double *cov6)
{
    normals_impl(xyz, n, radius, normals, n_neighbors, cov6);
}
