// sf_fusion.cpp — host-side pose prior of the registration path (float32 like the
// reference; microseconds of scalar math, it stays on the CPU and supplies the ICP's
// initial transformation without any extra device synchronisation).
//
// Restates localization/src/localization_node.cpp:62-77 (compass), :89-110 (odometry
// prediction), :112-128 (GPS/compass pose), :151-179 (gains), :329 (blend);
// localization/include/localization/geo_lib.hpp:38-83 (UTM);
// localization/src/global_map_frames_manager.cpp:69-91,209-248 and
// localization/src/stochastic_filter.cpp (StochasticFilter).  Built with
// -ffp-contract=off: the reference's x86-64 build has no FMA.
#include "slamfusion.h"

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct M4 {
    float v[16];
    float &operator()(int r, int c) { return v[4 * r + c]; }
    float operator()(int r, int c) const { return v[4 * r + c]; }
};

M4 load(const float *p) { M4 m; std::memcpy(m.v, p, sizeof(m.v)); return m; }
void store(const M4 &m, float *p) { std::memcpy(p, m.v, sizeof(m.v)); }

M4 zero() { M4 m; for (float &x : m.v) x = 0.0f; return m; }
M4 identity() { M4 m = zero(); for (int d = 0; d < 4; ++d) m(d, d) = 1.0f; return m; }

// coefficient-wise product, k ascending (Eigen's lazy fixed-size product)
M4 mul(const M4 &a, const M4 &b)
{
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = a(i, 0) * b(0, j);
            acc = acc + a(i, 1) * b(1, j);
            acc = acc + a(i, 2) * b(2, j);
            acc = acc + a(i, 3) * b(3, j);
            r(i, j) = acc;
        }
    return r;
}

float det3(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}

// adjugate / determinant (Eigen's 4x4 inverse is cofactor based too; last-bit rounding
// may differ from its SSE kernel)
M4 inverse(const M4 &m)
{
    M4 adj;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s[9];
            int k = 0;
            for (int i = 0; i < 4; ++i) {
                if (i == r) continue;
                for (int j = 0; j < 4; ++j) {
                    if (j == c) continue;
                    s[k++] = m(i, j);
                }
            }
            const float minor = det3(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
            adj(c, r) = ((r + c) & 1) ? -minor : minor;
        }
    const float det = m(0, 0) * adj(0, 0) + m(0, 1) * adj(1, 0) + m(0, 2) * adj(2, 0) + m(0, 3) * adj(3, 0);
    const float inv_det = 1.0f / det;
    M4 out;
    for (int i = 0; i < 16; ++i) out.v[i] = adj.v[i] * inv_det;
    return out;
}

// Eigen::AngleAxisf(angle, UnitZ).toRotationMatrix()
void rot_z(float angle, M4 &m)
{
    const float s = std::sin(angle), c = std::cos(angle);
    const float one_c = 1.0f - c;
    m(0, 0) = 0.0f + c; m(0, 1) = 0.0f - s; m(0, 2) = 0.0f;
    m(1, 0) = 0.0f + s; m(1, 1) = 0.0f + c; m(1, 2) = 0.0f;
    m(2, 0) = 0.0f;     m(2, 1) = 0.0f;     m(2, 2) = one_c * 1.0f + c;
}

} // namespace

extern "C" void sf_fusion_mat4f_inverse(const float A[16], float out[16]) { store(inverse(load(A)), out); }
extern "C" void sf_fusion_mat4f_mul(const float A[16], const float B[16], float out[16]) { store(mul(load(A), load(B)), out); }

// Eigen::Quaternionf::toRotationMatrix — localization_node.cpp:94-103
extern "C" void sf_fusion_quat_to_pose(const double q_wxyz[4], const double t[3], float T[16])
{
    const float w = (float)q_wxyz[0], x = (float)q_wxyz[1], y = (float)q_wxyz[2], z = (float)q_wxyz[3];
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M4 m = identity();
    m(0, 0) = 1.0f - (tyy + tzz); m(0, 1) = txy - twz; m(0, 2) = txz + twy;
    m(1, 0) = txy + twz; m(1, 1) = 1.0f - (txx + tzz); m(1, 2) = tyz - twx;
    m(2, 0) = txz - twy; m(2, 1) = tyz + twx; m(2, 2) = 1.0f - (txx + tyy);
    m(0, 3) = (float)t[0]; m(1, 3) = (float)t[1]; m(2, 3) = (float)t[2];
    store(m, T);
}

// localization_node.cpp:105-109
extern "C" void sf_fusion_odom_prediction(const float map_T_sensor[16], const float odom_T_prev[16], const float odom_T_cur[16], float out[16])
{
    const M4 previous_T_current = mul(inverse(load(odom_T_prev)), load(odom_T_cur));
    store(mul(load(map_T_sensor), previous_T_current), out);
}

// localization_node.cpp:64-76
extern "C" float sf_fusion_compass_to_yaw(double compass_deg)
{
    float yaw = static_cast<float>((90.0 - compass_deg) * M_PI / 180.0);
    if (yaw > M_PI) yaw -= 2 * M_PI;
    else if (yaw < -M_PI) yaw += 2 * M_PI;
    return yaw;
}

// geo_lib.hpp:38-83 — WGS84 transverse Mercator series; the southern false northing is
// added unconditionally (:82), reproduced
extern "C" void sf_fusion_ll_to_utm(double lat, double lon, double *northing, double *easting)
{
    constexpr double kA = 6378137.0, kE = 0.0818191908, kK0 = 0.9996;
    constexpr double kDeg = 0.017453292519943295769236907684886;
    const double e2 = kE * kE;
    const double lon_n = (lon + 180.0) - int((lon + 180.0) / 360.0) * 360.0 - 180.0;
    const double phi = lat * kDeg, lam = lon_n * kDeg;
    int zone = static_cast<int>((lon_n + 180.0) / 6.0) + 1;
    if (lat >= 56.0 && lat < 64.0 && lon_n >= 3.0 && lon_n < 12.0) zone = 32;
    const double lam0 = ((static_cast<double>(zone) - 1.0) * 6.0 - 180.0 + 3.0) * kDeg;
    const double ep2 = (e2) / (1.0 - e2);
    const double N = kA / std::sqrt(1.0 - e2 * std::sin(phi) * std::sin(phi));
    const double T = std::tan(phi) * std::tan(phi);
    const double C = ep2 * std::cos(phi) * std::cos(phi);
    const double A = std::cos(phi) * (lam - lam0);
    const double M = kA * ((1 - e2 / 4.0 - 3.0 * e2 * e2 / 64.0 - 5.0 * e2 * e2 * e2 / 256.0) * phi
                           - (3.0 * e2 / 8.0 + 3.0 * e2 * e2 / 32.0 + 45.0 * e2 * e2 * e2 / 1024.0) * std::sin(2.0 * phi)
                           + (15.0 * e2 * e2 / 256.0 + 45.0 * e2 * e2 * e2 / 1024.0) * std::sin(4.0 * phi)
                           - (35.0 * e2 * e2 * e2 / 3072.0) * std::sin(6.0 * phi));
    *easting = (kK0 * N * (A + (1 - T + C) * A * A * A / 6.0 + (5.0 - 18.0 * T + T * T + 72.0 * C - 58.0 * ep2) * A * A * A * A * A / 120.0) + 500000.0);
    *northing = kK0 * (M + N * std::tan(phi) * (A * A / 2 + (5.0 - T + 9.0 * C + 4.0 * C * C) * A * A * A * A / 24.0
                                                + (61.0 - 58.0 * T + T * T + 600.0 * C - 330.0 * ep2) * A * A * A * A * A * A / 720.0))
                + 10000000.0;
}

// global_map_frames_manager.cpp:69-91
extern "C" float sf_fusion_closest_altitude(const double *tab, int rows, double lat, double lon)
{
    if (rows <= 0 || !tab) return 0.0f;
    double min_dist = std::numeric_limits<double>::max();
    float alt = 0.0f;
    for (int i = 0; i < rows; ++i) {
        const double dist = std::sqrt(std::pow(lat - tab[3 * i], 2) + std::pow(lon - tab[3 * i + 1], 2));
        if (dist < min_dist) { min_dist = dist; alt = static_cast<float>(tab[3 * i + 2]); }
    }
    return alt;
}

// localization_node.cpp:112-128 — UTM magnitudes go through float32 storage
extern "C" void sf_fusion_gps_pose(const double map_T_global[16], float yaw, double lat, double lon, float table_alt, float out[16])
{
    double utm_n, utm_e;
    sf_fusion_ll_to_utm(lat, lon, &utm_n, &utm_e);
    M4 global_T_sensor = identity();
    rot_z(yaw, global_T_sensor);
    global_T_sensor(0, 3) = static_cast<float>(utm_e);
    global_T_sensor(1, 3) = static_cast<float>(utm_n);
    global_T_sensor(2, 3) = table_alt;
    M4 mtg;
    for (int i = 0; i < 16; ++i) mtg.v[i] = static_cast<float>(map_T_global[i]);
    store(mul(mtg, global_T_sensor), out);
}

// localization_node.cpp:151-179
extern "C" void sf_fusion_pose_gains(const double gps_cov[9], const double odom_cov[36], int fixed, float *odom_gain, float *gps_gain)
{
    if (fixed) { *odom_gain = 0.95f; *gps_gain = 0.05f; return; }
    const float odom_weight = (static_cast<float>(odom_cov[0]) + static_cast<float>(odom_cov[7])) + static_cast<float>(odom_cov[14]);
    const float gps_weight = (static_cast<float>(gps_cov[0]) + static_cast<float>(gps_cov[4])) + static_cast<float>(gps_cov[8]);
    const float total = odom_weight + gps_weight;
    *odom_gain = gps_weight / total;
    *gps_gain = odom_weight / total;
}

// localization_node.cpp:329 — element-wise blend, not re-orthonormalised
extern "C" void sf_fusion_blend(float g_odom, const float T_odom[16], float g_gps, const float T_gps[16], float out[16])
{
    for (int i = 0; i < 16; ++i) out[i] = g_odom * T_odom[i] + g_gps * T_gps[i];
}

// global_map_frames_manager.cpp:209-248
extern "C" void sf_fusion_map_T_global(const double *latlonalt, const float *yaw, int n, double out[16])
{
    double t[3] = {0, 0, 0}, yaw_avg = 0;
    for (int i = 0; i < n; ++i) {
        double no, ea;
        sf_fusion_ll_to_utm(latlonalt[3 * i], latlonalt[3 * i + 1], &no, &ea);
        t[0] += ea; t[1] += no; t[2] += latlonalt[3 * i + 2];
        yaw_avg += static_cast<double>(yaw[i]);
    }
    for (double &x : t) x /= static_cast<double>(n);
    yaw_avg /= static_cast<double>(n);
    const double s = std::sin(-yaw_avg), c = std::cos(-yaw_avg);
    const double R[9] = {c, -s, 0, s, c, 0, 0, 0, (1 - c) + c};
    for (int i = 0; i < 16; ++i) out[i] = 0;
    out[15] = 1;
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) out[4 * r + k] = R[3 * r + k];
        out[4 * r + 3] = (-R[3 * r]) * t[0] + (-R[3 * r + 1]) * t[1] + (-R[3 * r + 2]) * t[2];
    }
}

// ------------------------------------------------------------------ StochasticFilter
struct sf_sfilter {
    std::size_t queue_size;
    float n_std_dev_threshold;
    float min_distance_per_scan, max_distance_per_scan;
    std::vector<float> weights;
    std::vector<M4> queue;
    M4 previous;
};

// stochastic_filter.cpp:3-27
extern "C" sf_sfilter *sf_sfilter_create(int queue_size, float n_std_dev_threshold)
{
    if (queue_size <= 0) return nullptr;
    sf_sfilter *f = new sf_sfilter();
    f->queue_size = static_cast<std::size_t>(queue_size);
    f->n_std_dev_threshold = n_std_dev_threshold;
    f->queue.reserve(f->queue_size);
    f->previous = identity();
    f->min_distance_per_scan = 0.05f;
    f->max_distance_per_scan = 0.20f;
    f->weights.resize(f->queue_size);
    for (int i = 0; i < queue_size; ++i) f->weights[i] = std::exp(static_cast<float>(i - queue_size));
    float sum = 0.0f;
    for (float w : f->weights) sum += w;
    for (float &w : f->weights) w /= sum;
    return f;
}

extern "C" void sf_sfilter_destroy(sf_sfilter *f) { delete f; }

// stochastic_filter.cpp:39-42 (scan rate 10 Hz)
extern "C" void sf_sfilter_set_maximum_linear_velocity(sf_sfilter *f, float v) { f->max_distance_per_scan = v / 10.0f; }

extern "C" void sf_sfilter_weights(const sf_sfilter *f, float *w) { std::memcpy(w, f->weights.data(), sizeof(float) * f->queue_size); }

// stochastic_filter.cpp:44-55
extern "C" void sf_sfilter_add_pose_to_queue(sf_sfilter *f, const float pose[16])
{
    if (f->queue.size() >= f->queue_size) f->queue.erase(f->queue.begin());
    const M4 cur = load(pose);
    f->queue.push_back(mul(inverse(f->previous), cur));
    f->previous = cur;
}

// stochastic_filter.cpp:57-92
extern "C" float sf_sfilter_pose_zscore(const sf_sfilter *f, const float prev[16], const float cur[16])
{
    if (f->queue.size() < f->queue_size) return 0.0f;
    const M4 origin_prev = load(prev);
    M4 mean = zero();
    std::vector<float> xyz(3 * f->queue_size);
    for (std::size_t i = 0; i < f->queue_size; ++i) {
        const M4 tmp = mul(origin_prev, f->queue[i]);
        for (int k = 0; k < 16; ++k) mean.v[k] += f->weights[i] * tmp.v[k];
        xyz[3 * i] = tmp(0, 3); xyz[3 * i + 1] = tmp(1, 3); xyz[3 * i + 2] = tmp(2, 3);
    }
    const float mean_t[3] = {mean(0, 3), mean(1, 3), mean(2, 3)};
    float sd[3] = {0.0f, 0.0f, 0.0f};
    for (std::size_t i = 0; i < f->queue_size; ++i)
        for (int d = 0; d < 3; ++d) sd[d] += f->weights[i] * std::fabs(xyz[3 * i + d] - mean_t[d]);
    float z = -std::numeric_limits<float>::infinity();
    for (int d = 0; d < 3; ++d) {
        sd[d] = std::fmax(sd[d], f->min_distance_per_scan);
        sd[d] = std::fmin(sd[d], f->max_distance_per_scan / 3.0f);
        const float zd = std::fabs(cur[4 * d + 3] - mean_t[d]) / sd[d];
        if (zd > z) z = zd;
    }
    return z;
}

// stochastic_filter.cpp:94-113 (queue_i * previous here, previous * queue_i in the z-score)
extern "C" void sf_sfilter_apply_gaussian_filter(const sf_sfilter *f, const float prev[16], const float cur[16], float out[16])
{
    if (sf_sfilter_pose_zscore(f, prev, cur) > f->n_std_dev_threshold) {
        const M4 origin_prev = load(prev);
        M4 mean = zero();
        for (std::size_t i = 0; i < f->queue_size; ++i) {
            M4 wq;
            for (int k = 0; k < 16; ++k) wq.v[k] = f->weights[i] * f->queue[i].v[k];
            const M4 term = mul(wq, origin_prev);
            for (int k = 0; k < 16; ++k) mean.v[k] += term.v[k];
        }
        store(mean, out);
        return;
    }
    std::memcpy(out, cur, sizeof(float) * 16);
}
