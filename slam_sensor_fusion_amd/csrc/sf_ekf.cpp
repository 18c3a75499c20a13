// sf_ekf.cpp — error-state EKF pose prior with IMU pre-integration (BASELINE config 4).
//
// EXTENSION: the reference has no EKF and no IMU consumer (its prior is the covariance-weighted
// blend + StochasticFilter of localization_node.cpp:89-179,329-332, built in sf_fusion.cpp); there
// is nothing to be in parity with, and oracle/ekf_np.py is this build's own numpy restatement.
// Host code (a handful of 9x9 products per scan — microseconds), float64.
//
// Nominal state: position p, velocity v (map frame), attitude R (map <- sensor).  Error state
// dx = (dp, dv, dtheta) with the attitude error on the right: R_true = R * Exp(dtheta).
//   predict (IMU sample, period dt):  a_w = R a + g;  p += v dt + a_w dt^2 / 2;  v += a_w dt;  R = R Exp(w dt)
//       F = I + dt [[0 I 0], [0 0 -R [a]x], [0 0 -[w]x]],   Q = diag(0, (sigma_a dt)^2, (sigma_g dt)^2)
//   predict (odometry delta, the reference's a14 prediction): p += R dt_odom;  R = R dR_odom;  P += blockdiag(R C_p R^T, 0, C_r)
//   update: y = z - h(x), K = P H^T (H P H^T + R_m)^-1, inject K y, Joseph form for P.
//       GPS position: H = [I 0 0];  compass yaw: h = atan2(R10, R00), H = [0 0 e_z^T R] (near-level);
//       ICP pose: position as above plus r = Log(R^T R_meas) with H = [0 0 I].
#include "slamfusion.h"

#include <cmath>
#include <cstring>
#include <new>

struct sf_ekf {
    double p[3], v[3], R[9];
    double P[81];
    double sigma_g = 1e-3, sigma_a = 1e-2;
    double g[3] = {0.0, 0.0, -9.80665};
};

namespace {

void skew(const double w[3], double S[9])
{
    S[0] = 0; S[1] = -w[2]; S[2] = w[1];
    S[3] = w[2]; S[4] = 0; S[5] = -w[0];
    S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}

void mul33(const double A[9], const double B[9], double C[9])
{
    double T[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
    std::memcpy(C, T, sizeof(T));
}

void mulv3(const double A[9], const double x[3], double y[3])
{
    double t[3];
    for (int r = 0; r < 3; ++r) t[r] = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2];
    std::memcpy(y, t, sizeof(t));
}

// Rodrigues: Exp(w)
void so3_exp(const double w[3], double E[9])
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    double a, b; // sin(th)/th, (1 - cos(th))/th^2
    if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
    else { a = std::sin(th) / th; b = (1.0 - std::cos(th)) / th2; }
    double S[9], S2[9];
    skew(w, S);
    mul33(S, S, S2);
    for (int i = 0; i < 9; ++i) E[i] = a * S[i] + b * S2[i];
    E[0] += 1.0; E[4] += 1.0; E[8] += 1.0;
}

// Log(R) for rotations away from pi (the residual of a measurement update)
void so3_log(const double R[9], double w[3])
{
    double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double th = std::acos(c);
    const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double k = th < 1e-8 ? 0.5 + th * th / 12.0 : th / (2.0 * std::sin(th));
    for (int i = 0; i < 3; ++i) w[i] = k * v[i];
}

// C(n x m) = A(n x k) B(k x m)
void matmul(const double *A, const double *B, double *C, int n, int k, int m)
{
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < m; ++c) {
            double s = 0.0;
            for (int i = 0; i < k; ++i) s += A[r * k + i] * B[i * m + c];
            C[r * m + c] = s;
        }
}

void transpose(const double *A, double *At, int n, int m)
{
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < m; ++c) At[c * n + r] = A[r * m + c];
}

// in-place inverse of an n x n matrix (n <= 6), partial pivoting; false if singular
bool invert(double *A, int n)
{
    double M[6 * 12];
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < 2 * n; ++c) M[r * 2 * n + c] = c < n ? A[r * n + c] : (c - n == r ? 1.0 : 0.0);
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r)
            if (std::fabs(M[r * 2 * n + col]) > std::fabs(M[piv * 2 * n + col])) piv = r;
        if (!(std::fabs(M[piv * 2 * n + col]) > 1e-300)) return false;
        if (piv != col)
            for (int c = 0; c < 2 * n; ++c) { const double t = M[col * 2 * n + c]; M[col * 2 * n + c] = M[piv * 2 * n + c]; M[piv * 2 * n + c] = t; }
        const double d = M[col * 2 * n + col];
        for (int c = 0; c < 2 * n; ++c) M[col * 2 * n + c] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            const double f = M[r * 2 * n + col];
            if (f != 0.0)
                for (int c = 0; c < 2 * n; ++c) M[r * 2 * n + c] -= f * M[col * 2 * n + c];
        }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) A[r * n + c] = M[r * 2 * n + n + c];
    return true;
}

// measurement update with residual y (m), Jacobian H (m x 9), noise Rm (m x m); m <= 6
int update(sf_ekf *e, const double *y, const double *H, const double *Rm, int m)
{
    double Ht[9 * 6], PHt[9 * 6], S[36], K[9 * 6];
    transpose(H, Ht, m, 9);
    matmul(e->P, Ht, PHt, 9, 9, m);
    matmul(H, PHt, S, m, 9, m);
    for (int i = 0; i < m * m; ++i) S[i] += Rm[i];
    if (!invert(S, m)) return SF_ERR_INVALID;
    matmul(PHt, S, K, 9, m, m);
    double dx[9];
    matmul(K, y, dx, 9, m, 1);
    for (int i = 0; i < 3; ++i) { e->p[i] += dx[i]; e->v[i] += dx[3 + i]; }
    double E[9];
    so3_exp(dx + 6, E);
    mul33(e->R, E, e->R);
    // Joseph form: P = (I - K H) P (I - K H)^T + K Rm K^T
    double A[81], At[81], T1[81], T2[81], KR[9 * 6], Kt[6 * 9], KRK[81];
    matmul(K, H, A, 9, m, 9);
    for (int i = 0; i < 81; ++i) A[i] = -A[i];
    for (int i = 0; i < 9; ++i) A[10 * i] += 1.0;
    transpose(A, At, 9, 9);
    matmul(A, e->P, T1, 9, 9, 9);
    matmul(T1, At, T2, 9, 9, 9);
    matmul(K, Rm, KR, 9, m, m);
    transpose(K, Kt, 9, m);
    matmul(KR, Kt, KRK, 9, m, 9);
    for (int i = 0; i < 81; ++i) e->P[i] = T2[i] + KRK[i];
    for (int r = 0; r < 9; ++r) // keep it symmetric against rounding
        for (int c = r + 1; c < 9; ++c) e->P[9 * r + c] = e->P[9 * c + r] = 0.5 * (e->P[9 * r + c] + e->P[9 * c + r]);
    return SF_OK;
}

} // namespace

extern "C" int sf_ekf_create(sf_ekf **out)
{
    if (!out) return SF_ERR_INVALID;
    sf_ekf *e = new (std::nothrow) sf_ekf();
    if (!e) return SF_ERR_NOMEM;
    std::memset(e->p, 0, sizeof(e->p));
    std::memset(e->v, 0, sizeof(e->v));
    std::memset(e->R, 0, sizeof(e->R));
    e->R[0] = e->R[4] = e->R[8] = 1.0;
    std::memset(e->P, 0, sizeof(e->P));
    for (int i = 0; i < 9; ++i) e->P[10 * i] = 1.0;
    *out = e;
    return SF_OK;
}

extern "C" void sf_ekf_destroy(sf_ekf *e) { delete e; }

extern "C" int sf_ekf_reset(sf_ekf *e, const double T[16], const double v[3], const double P_diag[9])
{
    if (!e || !T) return SF_ERR_INVALID;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) e->R[3 * r + c] = T[4 * r + c];
        e->p[r] = T[4 * r + 3];
        e->v[r] = v ? v[r] : 0.0;
    }
    std::memset(e->P, 0, sizeof(e->P));
    for (int i = 0; i < 9; ++i) e->P[10 * i] = P_diag ? P_diag[i] : 1.0;
    return SF_OK;
}

extern "C" int sf_ekf_set_noise(sf_ekf *e, double gyro_sigma, double accel_sigma, const double gravity[3])
{
    if (!e || !(gyro_sigma >= 0.0) || !(accel_sigma >= 0.0)) return SF_ERR_INVALID;
    e->sigma_g = gyro_sigma;
    e->sigma_a = accel_sigma;
    if (gravity)
        for (int i = 0; i < 3; ++i) e->g[i] = gravity[i];
    return SF_OK;
}

extern "C" int sf_ekf_predict_imu(sf_ekf *e, const double *gyro, const double *accel, int64_t n, double dt)
{
    if (!e || n < 0 || (n > 0 && (!gyro || !accel)) || !(dt > 0.0)) return SF_ERR_INVALID;
    for (int64_t k = 0; k < n; ++k) {
        const double *w = gyro + 3 * k, *a = accel + 3 * k;
        double aw[3], Ra[9], Sa[9], Sw[9];
        mulv3(e->R, a, aw);
        for (int i = 0; i < 3; ++i) aw[i] += e->g[i];
        // F = I + dt * A, A = [[0 I 0], [0 0 -R [a]x], [0 0 -[w]x]]  (evaluated at the state before the step)
        skew(a, Sa);
        skew(w, Sw);
        mul33(e->R, Sa, Ra);
        double F[81];
        std::memset(F, 0, sizeof(F));
        for (int i = 0; i < 9; ++i) F[10 * i] = 1.0;
        for (int i = 0; i < 3; ++i) F[9 * i + 3 + i] = dt;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                F[9 * (3 + r) + 6 + c] = -dt * Ra[3 * r + c];
                F[9 * (6 + r) + 6 + c] += -dt * Sw[3 * r + c];
            }
        double Ft[81], T1[81], T2[81];
        transpose(F, Ft, 9, 9);
        matmul(F, e->P, T1, 9, 9, 9);
        matmul(T1, Ft, T2, 9, 9, 9);
        std::memcpy(e->P, T2, sizeof(T2));
        const double qv = e->sigma_a * dt * e->sigma_a * dt, qt = e->sigma_g * dt * e->sigma_g * dt;
        for (int i = 0; i < 3; ++i) { e->P[10 * (3 + i)] += qv; e->P[10 * (6 + i)] += qt; }
        // nominal state
        for (int i = 0; i < 3; ++i) {
            e->p[i] += e->v[i] * dt + 0.5 * aw[i] * dt * dt;
            e->v[i] += aw[i] * dt;
        }
        const double wd[3] = {w[0] * dt, w[1] * dt, w[2] * dt};
        double E[9];
        so3_exp(wd, E);
        mul33(e->R, E, e->R);
    }
    return SF_OK;
}

extern "C" int sf_ekf_predict_odometry(sf_ekf *e, const double odom_T_prev[16], const double odom_T_cur[16], const double cov_pos[3], const double cov_rot[3])
{
    if (!e || !odom_T_prev || !odom_T_cur) return SF_ERR_INVALID;
    // delta = prev^-1 cur (rigid)
    double Rp[9], Rc[9], Rpt[9], dR[9], dt3[3], tmp[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { Rp[3 * r + c] = odom_T_prev[4 * r + c]; Rc[3 * r + c] = odom_T_cur[4 * r + c]; }
    transpose(Rp, Rpt, 3, 3);
    mul33(Rpt, Rc, dR);
    for (int i = 0; i < 3; ++i) tmp[i] = odom_T_cur[4 * i + 3] - odom_T_prev[4 * i + 3];
    mulv3(Rpt, tmp, dt3);
    double step[3];
    mulv3(e->R, dt3, step);
    for (int i = 0; i < 3; ++i) e->p[i] += step[i];
    // P_pp += R C_p R^T, P_tt += C_r (evaluated with the attitude before the step)
    if (cov_pos) {
        double C[9] = {cov_pos[0], 0, 0, 0, cov_pos[1], 0, 0, 0, cov_pos[2]}, Rt[9], T1[9], T2[9];
        transpose(e->R, Rt, 3, 3);
        mul33(e->R, C, T1);
        mul33(T1, Rt, T2);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) e->P[9 * r + c] += T2[3 * r + c];
    }
    if (cov_rot)
        for (int i = 0; i < 3; ++i) e->P[10 * (6 + i)] += cov_rot[i];
    mul33(e->R, dR, e->R);
    return SF_OK;
}

extern "C" int sf_ekf_update_position(sf_ekf *e, const double z[3], const double cov[9])
{
    if (!e || !z || !cov) return SF_ERR_INVALID;
    double H[27] = {0}, y[3];
    for (int i = 0; i < 3; ++i) { H[9 * i + i] = 1.0; y[i] = z[i] - e->p[i]; }
    return update(e, y, H, cov, 3);
}

extern "C" int sf_ekf_update_yaw(sf_ekf *e, double yaw, double var)
{
    if (!e || !(var > 0.0)) return SF_ERR_INVALID;
    const double PI = 3.14159265358979323846;
    double y = yaw - std::atan2(e->R[3], e->R[0]);
    y = std::fmod(y + PI, 2.0 * PI);
    if (y < 0) y += 2.0 * PI;
    y -= PI;
    double H[9] = {0, 0, 0, 0, 0, 0, e->R[6], e->R[7], e->R[8]}; // world-z component of R dtheta
    return update(e, &y, H, &var, 1);
}

extern "C" int sf_ekf_update_pose(sf_ekf *e, const double T[16], const double cov_pos[3], const double cov_rot[3])
{
    if (!e || !T || !cov_pos || !cov_rot) return SF_ERR_INVALID;
    double H[54] = {0}, y[6], Rm[36] = {0}, Rmeas[9], Rt[9], dR[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rmeas[3 * r + c] = T[4 * r + c];
    transpose(e->R, Rt, 3, 3);
    mul33(Rt, Rmeas, dR);
    so3_log(dR, y + 3);
    for (int i = 0; i < 3; ++i) {
        y[i] = T[4 * i + 3] - e->p[i];
        H[9 * i + i] = 1.0;
        H[9 * (3 + i) + 6 + i] = 1.0;
        Rm[7 * i] = cov_pos[i];
        Rm[7 * (3 + i)] = cov_rot[i];
    }
    return update(e, y, H, Rm, 6);
}

extern "C" int sf_ekf_get(const sf_ekf *e, double T[16], double v[3], double P[81])
{
    if (!e) return SF_ERR_INVALID;
    if (T) {
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T[4 * r + c] = e->R[3 * r + c];
            T[4 * r + 3] = e->p[r];
        }
        T[12] = T[13] = T[14] = 0.0;
        T[15] = 1.0;
    }
    if (v) std::memcpy(v, e->v, sizeof(e->v));
    if (P) std::memcpy(P, e->P, sizeof(e->P));
    return SF_OK;
}
