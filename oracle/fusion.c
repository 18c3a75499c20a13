/* fusion.c — TEST INFRASTRUCTURE (oracle); see sf_oracle.h.
 * Host-side pose prior: odometry prediction, GPS/compass pose, gains, blend and the
 * StochasticFilter, restated from localization/src/localization_node.cpp:62-77,89-128,
 * 151-179,329, localization/src/stochastic_filter.cpp, geo_lib.hpp:38-83 and
 * global_map_frames_manager.cpp:69-91,209-248.  float32 where the reference is float32. */
#include "sf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ geodesy */
/* geo_lib.hpp:38-83 (transverse Mercator series, WGS84; the reference adds the southern
 * false northing unconditionally, :82).  Checked against oracle/_ref (the real header). */
void orc_ll_to_utm(double lat, double lon, double *northing, double *easting)
{
    const double a = 6378137.0;
    const double e = 0.0818191908;
    const double e2 = e * e;
    const double k0 = 0.9996;
    const double deg = 0.017453292519943295769236907684886;

    const double lon_w = (lon + 180.0) - (int)((lon + 180.0) / 360.0) * 360.0 - 180.0;
    const double phi = lat * deg;
    const double lam = lon_w * deg;
    int zone = (int)((lon_w + 180.0) / 6.0) + 1;
    if (lat >= 56.0 && lat < 64.0 && lon_w >= 3.0 && lon_w < 12.0) zone = 32;
    const double lam0 = (((double)zone - 1.0) * 6.0 - 180.0 + 3.0) * deg;
    const double ep2 = (e2) / (1.0 - e2);

    const double N = a / sqrt(1.0 - e2 * sin(phi) * sin(phi));
    const double T = tan(phi) * tan(phi);
    const double C = ep2 * cos(phi) * cos(phi);
    const double A = cos(phi) * (lam - lam0);

    const double M = a * ((1 - e2 / 4.0 - 3.0 * e2 * e2 / 64.0 - 5.0 * e2 * e2 * e2 / 256.0) * phi
                          - (3.0 * e2 / 8.0 + 3.0 * e2 * e2 / 32.0 + 45.0 * e2 * e2 * e2 / 1024.0) * sin(2.0 * phi)
                          + (15.0 * e2 * e2 / 256.0 + 45.0 * e2 * e2 * e2 / 1024.0) * sin(4.0 * phi)
                          - (35.0 * e2 * e2 * e2 / 3072.0) * sin(6.0 * phi));

    *easting = (k0 * N * (A + (1 - T + C) * A * A * A / 6.0
                          + (5.0 - 18.0 * T + T * T + 72.0 * C - 58.0 * ep2) * A * A * A * A * A / 120.0)
                + 500000.0);
    *northing = k0 * (M + N * tan(phi) * (A * A / 2 + (5.0 - T + 9.0 * C + 4.0 * C * C) * A * A * A * A / 24.0
                                          + (61.0 - 58.0 * T + T * T + 600.0 * C - 330.0 * ep2) * A * A * A * A * A * A / 720.0))
                + 10000000.0;
}

/* python `utm` package (0.7.x, un-vendored, unpinned) from_latlon as called at
 * localization_node.py:138: Krueger/USGS series with E = 0.00669438, hemisphere-aware
 * false northing.  Parity unpinned (package absent here). */
void orc_utm_from_latlon(double lat, double lon, double *easting, double *northing)
{
    const double K0 = 0.9996, E = 0.00669438, R = 6378137.0;
    const double E2 = E * E, E3 = E2 * E, E_P2 = E / (1 - E);
    const double M1 = 1 - E / 4 - 3 * E2 / 64 - 5 * E3 / 256;
    const double M2 = 3 * E / 8 + 3 * E2 / 32 + 45 * E3 / 1024;
    const double M3 = 15 * E2 / 256 + 45 * E3 / 1024;
    const double M4 = 35 * E3 / 3072;
    const double lat_rad = lat * M_PI / 180.0;
    const double ls = sin(lat_rad), lc = cos(lat_rad), lt = ls / lc;
    const double lt2 = lt * lt, lt4 = lt2 * lt2;
    int zone;
    if (lat >= 56 && lat < 64 && lon >= 3 && lon < 12) zone = 32;
    else if (lat >= 72 && lat <= 84 && lon >= 0) {
        if (lon < 9) zone = 31; else if (lon < 21) zone = 33; else if (lon < 33) zone = 35;
        else if (lon < 42) zone = 37; else zone = (int)((lon + 180) / 6) % 60 + 1;
    } else zone = (int)((lon + 180) / 6) % 60 + 1;
    const double lon_rad = lon * M_PI / 180.0;
    const double central = ((zone - 1) * 6 - 180 + 3) * M_PI / 180.0;
    const double n = R / sqrt(1 - E * ls * ls);
    const double c = E_P2 * lc * lc;
    double dl = lon_rad - central;
    dl = fmod(dl + M_PI, 2 * M_PI);
    if (dl < 0) dl += 2 * M_PI;
    dl -= M_PI;
    const double a = lc * dl, a2 = a * a, a3 = a2 * a, a4 = a3 * a, a5 = a4 * a, a6 = a5 * a;
    const double m = R * (M1 * lat_rad - M2 * sin(2 * lat_rad) + M3 * sin(4 * lat_rad) - M4 * sin(6 * lat_rad));
    *easting = K0 * n * (a + a3 / 6 * (1 - lt2 + c) + a5 / 120 * (5 - 18 * lt2 + lt4 + 72 * c - 58 * E_P2)) + 500000;
    *northing = K0 * (m + n * lt * (a2 / 2 + a4 / 24 * (5 - lt2 + 9 * c + 4 * c * c) + a6 / 720 * (61 - 58 * lt2 + lt4 + 600 * c - 330 * E_P2)));
    if (lat < 0) *northing += 10000000;
}

/* ------------------------------------------------------------------ 4x4 float helpers */
void orc_mat4f_mul(const float A[16], const float B[16], float out[16])
{
    float R[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            R[4 * r + c] = ((A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c]) + A[4 * r + 2] * B[8 + c]) + A[4 * r + 3] * B[12 + c];
    memcpy(out, R, sizeof(R));
}

/* general inverse by cofactors (Eigen's fixed 4x4 inverse is cofactor based as well;
 * rounding differs in the last bits, tolerance documented in the tests) */
void orc_mat4f_inverse(const float m[16], float out[16])
{
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    float id = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * id;
}

/* Eigen::Quaternionf(w,x,y,z).toRotationMatrix() + translation, localization_node.cpp:94-103 */
void orc_quat_to_pose(const double q_wxyz[4], const double t[3], float T[16])
{
    const float w = (float)q_wxyz[0], x = (float)q_wxyz[1], y = (float)q_wxyz[2], z = (float)q_wxyz[3];
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    float R[16] = {1.0f - (tyy + tzz), txy - twz, txz + twy, (float)t[0],
                   txy + twz, 1.0f - (txx + tzz), tyz - twx, (float)t[1],
                   txz - twy, tyz + twx, 1.0f - (txx + tyy), (float)t[2],
                   0, 0, 0, 1};
    memcpy(T, R, sizeof(R));
}

/* localization_node.cpp:105-109 */
void orc_odom_prediction(const float map_T_sensor[16], const float odom_T_prev[16],
                         const float odom_T_cur[16], float out[16])
{
    float inv[16], rel[16];
    orc_mat4f_inverse(odom_T_prev, inv);
    orc_mat4f_mul(inv, odom_T_cur, rel);
    orc_mat4f_mul(map_T_sensor, rel, out);
}

/* localization_node.cpp:64-76 (float member, double comparisons against M_PI) */
float orc_compass_to_yaw(double compass_deg)
{
    float yaw = (float)((90.0 - compass_deg) * M_PI / 180.0);
    if (yaw > M_PI) yaw -= 2 * M_PI;
    else if (yaw < -M_PI) yaw += 2 * M_PI;
    return yaw;
}

/* global_map_frames_manager.cpp:69-91 */
float orc_closest_altitude(const double *table, int rows, double lat, double lon)
{
    if (rows <= 0) return 0.0f;
    double best = 1.7976931348623157e308;
    float alt = 0.0f;
    for (int i = 0; i < rows; ++i) {
        double d = sqrt(pow(lat - table[3 * i], 2) + pow(lon - table[3 * i + 1], 2));
        if (d < best) { best = d; alt = (float)table[3 * i + 2]; }
    }
    return alt;
}

/* localization_node.cpp:112-128.  UTM magnitudes are stored into a float32 matrix. */
void orc_gps_pose(const double map_T_global[16], float yaw, double lat, double lon,
                  float table_alt, float out[16])
{
    double n, e;
    orc_ll_to_utm(lat, lon, &n, &e);
    const float s = sinf(yaw), c = cosf(yaw);
    const float one_c = 1.0f - c;
    float G[16] = {0.0f + c, 0.0f - s, 0.0f, (float)e,
                   0.0f + s, 0.0f + c, 0.0f, (float)n,
                   0.0f, 0.0f, one_c * 1.0f + c, table_alt,
                   0, 0, 0, 1};
    float M[16];
    for (int i = 0; i < 16; ++i) M[i] = (float)map_T_global[i];
    orc_mat4f_mul(M, G, out);
}

/* localization_node.cpp:151-179 */
void orc_pose_gains(const double gps_cov[9], const double odom_cov[36], int fixed,
                    float *odom_gain, float *gps_gain)
{
    if (fixed) { *odom_gain = 0.95f; *gps_gain = 0.05f; return; }
    const float odom_w = ((float)odom_cov[0] + (float)odom_cov[7]) + (float)odom_cov[14];
    const float gps_w = ((float)gps_cov[0] + (float)gps_cov[4]) + (float)gps_cov[8];
    const float total = odom_w + gps_w;
    *odom_gain = gps_w / total;
    *gps_gain = odom_w / total;
}

/* localization_node.cpp:329 */
void orc_blend(float g_odom, const float T_odom[16], float g_gps, const float T_gps[16], float out[16])
{
    for (int i = 0; i < 16; ++i) out[i] = g_odom * T_odom[i] + g_gps * T_gps[i];
}

/* global_map_frames_manager.cpp:209-248 */
void orc_map_T_global(const double *latlonalt, const float *yaw, int n, double out[16])
{
    double t[3] = {0, 0, 0}, yavg = 0;
    for (int i = 0; i < n; ++i) {
        double no, ea;
        orc_ll_to_utm(latlonalt[3 * i], latlonalt[3 * i + 1], &no, &ea);
        t[0] += ea; t[1] += no; t[2] += latlonalt[3 * i + 2];
        yavg += (double)yaw[i];
    }
    for (int d = 0; d < 3; ++d) t[d] /= (double)n;
    yavg /= (double)n;
    const double ang = -yavg, s = sin(ang), c = cos(ang);
    const double R[9] = {c, -s, 0, s, c, 0, 0, 0, (1 - c) + c};
    for (int i = 0; i < 16; ++i) out[i] = 0;
    out[15] = 1;
    for (int r = 0; r < 3; ++r) {
        for (int cc = 0; cc < 3; ++cc) out[4 * r + cc] = R[3 * r + cc];
        out[4 * r + 3] = (-R[3 * r]) * t[0] + (-R[3 * r + 1]) * t[1] + (-R[3 * r + 2]) * t[2];
    }
}

/* ------------------------------------------------------------------ StochasticFilter */
struct orc_sfilter {
    int q;
    float thr;
    float min_d, max_d;
    float *w;
    float prev[16];
    float *queue; /* up to q matrices */
    int count;
};

/* stochastic_filter.cpp:3-27 */
orc_sfilter *orc_sfilter_new(int queue_size, float z_threshold)
{
    orc_sfilter *f = (orc_sfilter *)calloc(1, sizeof(*f));
    f->q = queue_size;
    f->thr = z_threshold;
    f->min_d = 0.05f;
    f->max_d = 0.20f;
    f->w = (float *)malloc(sizeof(float) * (size_t)queue_size);
    f->queue = (float *)malloc(sizeof(float) * 16 * (size_t)queue_size);
    for (int i = 0; i < 16; ++i) f->prev[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    float sum = 0.0f;
    for (int i = 0; i < queue_size; ++i) { f->w[i] = expf((float)(i - queue_size)); }
    for (int i = 0; i < queue_size; ++i) sum += f->w[i];
    for (int i = 0; i < queue_size; ++i) f->w[i] /= sum;
    return f;
}

void orc_sfilter_free(orc_sfilter *f)
{
    if (!f) return;
    free(f->w); free(f->queue); free(f);
}

void orc_sfilter_weights(const orc_sfilter *f, float *w) { memcpy(w, f->w, sizeof(float) * (size_t)f->q); }

/* stochastic_filter.cpp:44-55 */
void orc_sfilter_add_pose(orc_sfilter *f, const float pose[16])
{
    if (f->count >= f->q) {
        memmove(f->queue, f->queue + 16, sizeof(float) * 16 * (size_t)(f->q - 1));
        f->count = f->q - 1;
    }
    float inv[16];
    orc_mat4f_inverse(f->prev, inv);
    orc_mat4f_mul(inv, pose, f->queue + 16 * (size_t)f->count);
    f->count++;
    memcpy(f->prev, pose, sizeof(float) * 16);
}

/* stochastic_filter.cpp:57-92 */
float orc_sfilter_zscore(const orc_sfilter *f, const float prev[16], const float cur[16])
{
    if (f->count < f->q) return 0.0f;
    float mean[16] = {0}, xyz[64][3];
    for (int i = 0; i < f->q && i < 64; ++i) {
        float tmp[16];
        orc_mat4f_mul(prev, f->queue + 16 * (size_t)i, tmp);
        for (int k = 0; k < 16; ++k) mean[k] += f->w[i] * tmp[k];
        xyz[i][0] = tmp[3]; xyz[i][1] = tmp[7]; xyz[i][2] = tmp[11];
    }
    const float m[3] = {mean[3], mean[7], mean[11]};
    float sd[3] = {0, 0, 0};
    for (int i = 0; i < f->q && i < 64; ++i)
        for (int d = 0; d < 3; ++d) sd[d] += f->w[i] * fabsf(xyz[i][d] - m[d]);
    float z = -INFINITY;
    const float cur_t[3] = {cur[3], cur[7], cur[11]};
    for (int d = 0; d < 3; ++d) {
        if (sd[d] < f->min_d) sd[d] = f->min_d;            /* cwiseMax */
        if (sd[d] > f->max_d / 3.0f) sd[d] = f->max_d / 3.0f; /* cwiseMin */
        float zd = fabsf(cur_t[d] - m[d]) / sd[d];
        if (zd > z) z = zd;
    }
    return z;
}

/* stochastic_filter.cpp:94-113 — note queue_i * prev here vs prev * queue_i above */
void orc_sfilter_apply(const orc_sfilter *f, const float prev[16], const float cur[16], float out[16])
{
    const float z = orc_sfilter_zscore(f, prev, cur);
    if (z > f->thr) {
        float mean[16] = {0};
        for (int i = 0; i < f->q; ++i) {
            float wq[16], tmp[16];
            for (int k = 0; k < 16; ++k) wq[k] = f->w[i] * f->queue[16 * (size_t)i + k];
            orc_mat4f_mul(wq, prev, tmp);
            for (int k = 0; k < 16; ++k) mean[k] += tmp[k];
        }
        memcpy(out, mean, sizeof(mean));
        return;
    }
    memcpy(out, cur, sizeof(float) * 16);
}

/* brute_force_alignment.cpp:160-179: -0, +0, -s, +s, ... for i < range/(2 step) + 1 */
int orc_bf_sequence(float range, float step, float *seq, int cap)
{
    int k = 0;
    for (int i = 0; i < range / (2 * step) + 1; ++i) {
        if (k < cap) seq[k] = -i * step;
        ++k;
        if (k < cap) seq[k] = i * step;
        ++k;
    }
    return k;
}

/* ------------------------------------------------------------------ BruteForceAlignment */
/* brute_force_alignment.cpp:65-136: every candidate (x, y, z, yaw) in nesting order,
 * T = previous * [Rz(yaw) | x y z] in float32, score = mean of the SQUARED NN distances
 * (float32 sequential sum), first candidate under the threshold wins at once. */
int orc_bf_align(const float *src, int n, const float *tgt, int m, float prev_T[16],
                 const orc_bf_params *prm, float best_T[16], float *best_score, int *index,
                 int *n_candidates, float *scores)
{
    float xs[512], ys[512], zs[512], ws[512];
    const int nx = orc_bf_sequence(prm->x_range, prm->x_step, xs, 512);
    const int ny = orc_bf_sequence(prm->y_range, prm->y_step, ys, 512);
    const int nz = orc_bf_sequence(prm->z_range, prm->z_step, zs, 512);
    const int nw = orc_bf_sequence(prm->yaw_range, prm->yaw_step, ws, 512);
    if (n_candidates) *n_candidates = nx * ny * nz * nw;
    orc_kdtree_f *tree = orc_kdtree_f_build(tgt, m, 15);
    float bestT[16], bscore = 3.402823466e+38f;
    for (int i = 0; i < 16; ++i) bestT[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    int bidx = -1, cand = 0, found = 0;
    float *q = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    int *nn_i = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    float *nn_d = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int a = 0; a < nx && !found; ++a)
        for (int b = 0; b < ny && !found; ++b)
            for (int c = 0; c < nz && !found; ++c)
                for (int d = 0; d < nw && !found; ++d, ++cand) {
                    const float yaw = ws[d];
                    const float sn = sinf(yaw), cs = cosf(yaw), one_c = 1.0f - cs;
                    const float L[16] = {0.0f + cs, 0.0f - sn, 0.0f, xs[a],
                                         0.0f + sn, 0.0f + cs, 0.0f, ys[b],
                                         0.0f, 0.0f, one_c * 1.0f + cs, zs[c],
                                         0, 0, 0, 1};
                    float T[16];
                    orc_mat4f_mul(prev_T, L, T);
                    for (int i = 0; i < n; ++i) { /* T * Vector4f(p, 1): column combination, k ascending */
                        const float *p = src + 3 * (size_t)i;
                        q[3 * (size_t)i + 0] = ((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3] * 1.0f;
                        q[3 * (size_t)i + 1] = ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7] * 1.0f;
                        q[3 * (size_t)i + 2] = ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11] * 1.0f;
                    }
                    orc_kdtree_f_nn(tree, q, n, nn_i, nn_d);
                    float score = 0.0f;
                    for (int i = 0; i < n; ++i) score += nn_d[i];
                    score /= (float)n;
                    if (scores) scores[cand] = score;
                    if (score < bscore) { bscore = score; memcpy(bestT, T, sizeof(bestT)); bidx = cand; }
                    if (score < prm->threshold) {
                        memcpy(best_T, T, sizeof(float) * 16);
                        *best_score = score;
                        *index = cand;
                        found = 1;
                    }
                }
    if (!found) {
        memcpy(prev_T, bestT, sizeof(bestT)); /* :123 */
        memcpy(best_T, bestT, sizeof(bestT));
        *best_score = bscore;
        *index = bidx;
        found = bscore < prm->threshold; /* :126-131 (cannot be true here, kept for fidelity) */
    }
    orc_kdtree_f_free(tree);
    free(q); free(nn_i); free(nn_d);
    return found;
}
