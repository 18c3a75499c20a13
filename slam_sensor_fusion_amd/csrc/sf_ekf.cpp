// sf_ekf.cpp — error-state EKF pose prior with IMU pre-integration (BASELINE config 4).
//
// EXTENSION: the reference has no EKF and no IMU consumer (its prior is the covariance-weighted
// blend + StochasticFilter of localization_node.cpp:89-179,329-332, built in sf_fusion.cpp); there
// is nothing to be in parity with, and oracle/ekf_np.py is this build's own numpy restatement.
// Host code (a handful of 15x15 products per IMU sample — microseconds), float64.
//
// Nominal state: position p, velocity v (map frame), attitude R (map <- sensor), gyro bias bg, accelerometer
// bias ba.  Error state dx = (dp, dv, dtheta, dbg, dba), attitude error on the right: R_true = R * Exp(dtheta).
//   predict (IMU sample w_m, a_m, period dt):  w = w_m - bg, a = a_m - ba, a_w = R a + g;
//       p += v dt + a_w dt^2 / 2;  v += a_w dt;  R = R Exp(w dt)
//       F = I + dt [[0 I 0 0 0], [0 0 -R[a]x 0 -R], [0 0 -[w]x -I 0], [0 ...], [0 ...]]
//       Q = diag(0, (sigma_a dt)^2, (sigma_g dt)^2, sigma_bg^2 dt, sigma_ba^2 dt)        (biases: random walks)
//   predict (odometry delta d = prev^-1 cur, the reference's a14 prediction): p += R d_t;  R = R d_R
//       F: dp += -R [d_t]x dtheta;  dtheta <- d_R^T dtheta;   Q = blockdiag(R C_p R^T, 0, C_r, 0, 0)
//   update: y = z - h(x), K = P H^T (H P H^T + R_m)^-1, inject K y, Joseph form for P.
//       GPS position: H = [I 0 0 0 0];  compass yaw: h = atan2(R10, R00), H = [0 0 e_z^T R 0 0] (near-level);
//       ICP pose: position as above plus r = Log(R^T R_meas) with H = [0 0 I 0 0].
// With zero bias variance and zero bias walk (the defaults after sf_ekf_reset) the bias states never move and the
// filter is the 9-state (p, v, theta) one.
#include "slamfusion.h"

#include <cmath>
#include <cstring>
#include <new>

constexpr int NS = 15; // error-state size

struct sf_ekf {
    double p[3], v[3], R[9], bg[3], ba[3];
    double P[NS * NS];
    double sigma_g = 1e-3, sigma_a = 1e-2;
    double sigma_bg = 0.0, sigma_ba = 0.0; // bias random walks (rad/s/sqrt(s), m/s^2/sqrt(s))
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

// first-order polar step back onto SO(3): R <- R (3 I - R^T R) / 2 (products of many small rotations drift)
void reorthonormalise(double R[9])
{
    double Rt[9], G[9];
    transpose(R, Rt, 3, 3);
    mul33(Rt, R, G);
    for (int i = 0; i < 9; ++i) G[i] = -0.5 * G[i];
    G[0] += 1.5; G[4] += 1.5; G[8] += 1.5;
    mul33(R, G, R);
}

// P <- F P F^T
void propagate(sf_ekf *e, const double *F)
{
    double Ft[NS * NS], T1[NS * NS], T2[NS * NS];
    transpose(F, Ft, NS, NS);
    matmul(F, e->P, T1, NS, NS, NS);
    matmul(T1, Ft, T2, NS, NS, NS);
    std::memcpy(e->P, T2, sizeof(T2));
}

// measurement update with residual y (m), Jacobian H (m x NS), noise Rm (m x m); m <= 6
int update(sf_ekf *e, const double *y, const double *H, const double *Rm, int m)
{
    double Ht[NS * 6], PHt[NS * 6], S[36], K[NS * 6];
    transpose(H, Ht, m, NS);
    matmul(e->P, Ht, PHt, NS, NS, m);
    matmul(H, PHt, S, m, NS, m);
    for (int i = 0; i < m * m; ++i) S[i] += Rm[i];
    if (!invert(S, m)) return SF_ERR_INVALID;
    matmul(PHt, S, K, NS, m, m);
    double dx[NS];
    matmul(K, y, dx, NS, m, 1);
    for (int i = 0; i < 3; ++i) { e->p[i] += dx[i]; e->v[i] += dx[3 + i]; e->bg[i] += dx[9 + i]; e->ba[i] += dx[12 + i]; }
    double E[9];
    so3_exp(dx + 6, E);
    mul33(e->R, E, e->R);
    // Joseph form: P = (I - K H) P (I - K H)^T + K Rm K^T
    double A[NS * NS], At[NS * NS], T1[NS * NS], T2[NS * NS], KR[NS * 6], Kt[6 * NS], KRK[NS * NS];
    matmul(K, H, A, NS, m, NS);
    for (int i = 0; i < NS * NS; ++i) A[i] = -A[i];
    for (int i = 0; i < NS; ++i) A[(NS + 1) * i] += 1.0;
    transpose(A, At, NS, NS);
    matmul(A, e->P, T1, NS, NS, NS);
    matmul(T1, At, T2, NS, NS, NS);
    matmul(K, Rm, KR, NS, m, m);
    transpose(K, Kt, NS, m);
    matmul(KR, Kt, KRK, NS, m, NS);
    for (int i = 0; i < NS * NS; ++i) e->P[i] = T2[i] + KRK[i];
    for (int r = 0; r < NS; ++r) // keep it symmetric against rounding
        for (int c = r + 1; c < NS; ++c) e->P[NS * r + c] = e->P[NS * c + r] = 0.5 * (e->P[NS * r + c] + e->P[NS * c + r]);
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
    std::memset(e->bg, 0, sizeof(e->bg));
    std::memset(e->ba, 0, sizeof(e->ba));
    std::memset(e->R, 0, sizeof(e->R));
    e->R[0] = e->R[4] = e->R[8] = 1.0;
    std::memset(e->P, 0, sizeof(e->P));
    for (int i = 0; i < 9; ++i) e->P[(NS + 1) * i] = 1.0;
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
        e->bg[r] = e->ba[r] = 0.0;
    }
    std::memset(e->P, 0, sizeof(e->P));
    for (int i = 0; i < 9; ++i) e->P[(NS + 1) * i] = P_diag ? P_diag[i] : 1.0;
    return SF_OK;
}

extern "C" int sf_ekf_set_bias(sf_ekf *e, const double gyro_bias[3], const double accel_bias[3], const double gyro_bias_var[3], const double accel_bias_var[3])
{
    if (!e) return SF_ERR_INVALID;
    for (int i = 0; i < 3; ++i) {
        if (gyro_bias) e->bg[i] = gyro_bias[i];
        if (accel_bias) e->ba[i] = accel_bias[i];
    }
    for (int i = 0; i < 3; ++i) {
        if (gyro_bias_var) {
            if (!(gyro_bias_var[i] >= 0.0)) return SF_ERR_INVALID;
            for (int c = 0; c < NS; ++c) e->P[NS * (9 + i) + c] = e->P[NS * c + 9 + i] = 0.0;
            e->P[(NS + 1) * (9 + i)] = gyro_bias_var[i];
        }
        if (accel_bias_var) {
            if (!(accel_bias_var[i] >= 0.0)) return SF_ERR_INVALID;
            for (int c = 0; c < NS; ++c) e->P[NS * (12 + i) + c] = e->P[NS * c + 12 + i] = 0.0;
            e->P[(NS + 1) * (12 + i)] = accel_bias_var[i];
        }
    }
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

extern "C" int sf_ekf_set_bias_noise(sf_ekf *e, double gyro_bias_walk, double accel_bias_walk)
{
    if (!e || !(gyro_bias_walk >= 0.0) || !(accel_bias_walk >= 0.0)) return SF_ERR_INVALID;
    e->sigma_bg = gyro_bias_walk;
    e->sigma_ba = accel_bias_walk;
    return SF_OK;
}

extern "C" int sf_ekf_predict_imu(sf_ekf *e, const double *gyro, const double *accel, int64_t n, double dt)
{
    if (!e || n < 0 || (n > 0 && (!gyro || !accel)) || !(dt > 0.0)) return SF_ERR_INVALID;
    for (int64_t k = 0; k < n; ++k) {
        double w[3], a[3];
        for (int i = 0; i < 3; ++i) { w[i] = gyro[3 * k + i] - e->bg[i]; a[i] = accel[3 * k + i] - e->ba[i]; }
        double aw[3], Ra[9], Sa[9], Sw[9];
        mulv3(e->R, a, aw);
        for (int i = 0; i < 3; ++i) aw[i] += e->g[i];
        // F = I + dt * A (evaluated at the state before the step)
        skew(a, Sa);
        skew(w, Sw);
        mul33(e->R, Sa, Ra);
        double F[NS * NS];
        std::memset(F, 0, sizeof(F));
        for (int i = 0; i < NS; ++i) F[(NS + 1) * i] = 1.0;
        for (int i = 0; i < 3; ++i) {
            F[NS * i + 3 + i] = dt;            // dp <- dv
            F[NS * (6 + i) + 9 + i] = -dt;     // dtheta <- dbg
        }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                F[NS * (3 + r) + 6 + c] = -dt * Ra[3 * r + c];      // dv <- dtheta
                F[NS * (3 + r) + 12 + c] = -dt * e->R[3 * r + c];   // dv <- dba
                F[NS * (6 + r) + 6 + c] += -dt * Sw[3 * r + c];     // dtheta <- dtheta
            }
        propagate(e, F);
        const double qv = e->sigma_a * dt * e->sigma_a * dt, qt = e->sigma_g * dt * e->sigma_g * dt;
        const double qbg = e->sigma_bg * e->sigma_bg * dt, qba = e->sigma_ba * e->sigma_ba * dt;
        for (int i = 0; i < 3; ++i) {
            e->P[(NS + 1) * (3 + i)] += qv;
            e->P[(NS + 1) * (6 + i)] += qt;
            e->P[(NS + 1) * (9 + i)] += qbg;
            e->P[(NS + 1) * (12 + i)] += qba;
        }
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
    if (n > 0) reorthonormalise(e->R);
    return SF_OK;
}

extern "C" int sf_ekf_predict_odometry(sf_ekf *e, const double odom_T_prev[16], const double odom_T_cur[16], const double cov_pos[3], const double cov_rot[3])
{
    if (!e || !odom_T_prev || !odom_T_cur) return SF_ERR_INVALID;
    // delta = prev^-1 cur (rigid)
    double Rp[9], Rc[9], Rpt[9], dR[9], dRt[9], dt3[3], tmp[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { Rp[3 * r + c] = odom_T_prev[4 * r + c]; Rc[3 * r + c] = odom_T_cur[4 * r + c]; }
    transpose(Rp, Rpt, 3, 3);
    mul33(Rpt, Rc, dR);
    for (int i = 0; i < 3; ++i) tmp[i] = odom_T_cur[4 * i + 3] - odom_T_prev[4 * i + 3];
    mulv3(Rpt, tmp, dt3);
    // covariance first, with the attitude before the step: p' = p + R Exp(dtheta) d  =>  dp' = dp - R [d]x dtheta;
    // R' = R Exp(dtheta) dR = R dR Exp(dR^T dtheta)  =>  dtheta' = dR^T dtheta
    double F[NS * NS], Sd[9], RSd[9];
    std::memset(F, 0, sizeof(F));
    for (int i = 0; i < NS; ++i) F[(NS + 1) * i] = 1.0;
    skew(dt3, Sd);
    mul33(e->R, Sd, RSd);
    transpose(dR, dRt, 3, 3);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            F[NS * r + 6 + c] = -RSd[3 * r + c];
            F[NS * (6 + r) + 6 + c] = dRt[3 * r + c];
        }
    propagate(e, F);
    if (cov_pos) {
        double C[9] = {cov_pos[0], 0, 0, 0, cov_pos[1], 0, 0, 0, cov_pos[2]}, Rt[9], T1[9], T2[9];
        transpose(e->R, Rt, 3, 3);
        mul33(e->R, C, T1);
        mul33(T1, Rt, T2);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) e->P[NS * r + c] += T2[3 * r + c];
    }
    if (cov_rot)
        for (int i = 0; i < 3; ++i) e->P[(NS + 1) * (6 + i)] += cov_rot[i];
    double step[3];
    mulv3(e->R, dt3, step);
    for (int i = 0; i < 3; ++i) e->p[i] += step[i];
    mul33(e->R, dR, e->R);
    reorthonormalise(e->R);
    return SF_OK;
}

extern "C" int sf_ekf_update_position(sf_ekf *e, const double z[3], const double cov[9])
{
    if (!e || !z || !cov) return SF_ERR_INVALID;
    double H[3 * NS] = {0}, y[3];
    for (int i = 0; i < 3; ++i) { H[NS * i + i] = 1.0; y[i] = z[i] - e->p[i]; }
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
    double H[NS] = {0};
    H[6] = e->R[6]; H[7] = e->R[7]; H[8] = e->R[8]; // world-z component of R dtheta
    return update(e, &y, H, &var, 1);
}

extern "C" int sf_ekf_update_pose(sf_ekf *e, const double T[16], const double cov_pos[3], const double cov_rot[3])
{
    if (!e || !T || !cov_pos || !cov_rot) return SF_ERR_INVALID;
    double H[6 * NS] = {0}, y[6], Rm[36] = {0}, Rmeas[9], Rt[9], dR[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rmeas[3 * r + c] = T[4 * r + c];
    transpose(e->R, Rt, 3, 3);
    mul33(Rt, Rmeas, dR);
    so3_log(dR, y + 3);
    for (int i = 0; i < 3; ++i) {
        y[i] = T[4 * i + 3] - e->p[i];
        H[NS * i + i] = 1.0;
        H[NS * (3 + i) + 6 + i] = 1.0;
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
    if (P) // the (dp, dv, dtheta) block
        for (int r = 0; r < 9; ++r)
            for (int c = 0; c < 9; ++c) P[9 * r + c] = e->P[NS * r + c];
    return SF_OK;
}

extern "C" int sf_ekf_get_full(const sf_ekf *e, double gyro_bias[3], double accel_bias[3], double P[225])
{
    if (!e) return SF_ERR_INVALID;
    if (gyro_bias) std::memcpy(gyro_bias, e->bg, sizeof(e->bg));
    if (accel_bias) std::memcpy(accel_bias, e->ba, sizeof(e->ba));
    if (P) std::memcpy(P, e->P, sizeof(e->P));
    return SF_OK;
}
