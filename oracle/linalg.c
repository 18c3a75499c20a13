/* linalg.c — TEST INFRASTRUCTURE (oracle); see sf_oracle.h.
 * 3x3 SVD (one-sided Jacobi) standing in for Eigen::JacobiSVD<Matrix3f>
 * (localization/src/icp_point_to_point.cpp:137) and the Kabsch step of :112-159. */
#include "sf_oracle.h"
#include <float.h>
#include <math.h>
#include <string.h>

#define DEFINE_SVD3(REAL, NAME, EPS, TINY, SQRT, FABS)                                        \
    void NAME(const REAL A[9], REAL U[9], REAL S[3], REAL V[9])                               \
    {                                                                                         \
        REAL u[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};                                        \
        memcpy(u, A, sizeof(u));                                                              \
        static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};                                  \
        for (int sweep = 0; sweep < 60; ++sweep) {                                            \
            int rotated = 0;                                                                  \
            for (int k = 0; k < 3; ++k) {                                                     \
                int p = P[k], q = Q[k];                                                       \
                REAL al = 0, be = 0, ga = 0;                                                  \
                for (int i = 0; i < 3; ++i) {                                                 \
                    al += u[3 * i + p] * u[3 * i + p];                                        \
                    be += u[3 * i + q] * u[3 * i + q];                                        \
                    ga += u[3 * i + p] * u[3 * i + q];                                        \
                }                                                                             \
                if (FABS(ga) <= (EPS)*SQRT(al * be) || FABS(ga) < (TINY)) continue;           \
                REAL zeta = (be - al) / (2 * ga);                                             \
                REAL t = (zeta >= 0 ? (REAL)1 : (REAL)-1) / (FABS(zeta) + SQRT(1 + zeta * zeta)); \
                REAL c = 1 / SQRT(1 + t * t), s = c * t;                                      \
                for (int i = 0; i < 3; ++i) {                                                 \
                    REAL a = u[3 * i + p], b = u[3 * i + q];                                  \
                    u[3 * i + p] = c * a - s * b;                                             \
                    u[3 * i + q] = s * a + c * b;                                             \
                    a = v[3 * i + p]; b = v[3 * i + q];                                       \
                    v[3 * i + p] = c * a - s * b;                                             \
                    v[3 * i + q] = s * a + c * b;                                             \
                }                                                                             \
                rotated = 1;                                                                  \
            }                                                                                 \
            if (!rotated) break;                                                              \
        }                                                                                     \
        REAL s[3];                                                                            \
        int ord[3] = {0, 1, 2};                                                               \
        for (int j = 0; j < 3; ++j)                                                           \
            s[j] = SQRT(u[j] * u[j] + u[3 + j] * u[3 + j] + u[6 + j] * u[6 + j]);             \
        for (int a = 0; a < 2; ++a)                                                           \
            for (int b = a + 1; b < 3; ++b)                                                   \
                if (s[ord[b]] > s[ord[a]]) { int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; } \
        for (int j = 0; j < 3; ++j) {                                                         \
            int o = ord[j];                                                                   \
            S[j] = s[o];                                                                      \
            for (int i = 0; i < 3; ++i) { U[3 * i + j] = u[3 * i + o]; V[3 * i + j] = v[3 * i + o]; } \
        }                                                                                     \
        REAL thr = S[0] * (EPS)*8;                                                            \
        int rank = 0;                                                                         \
        for (int j = 0; j < 3; ++j)                                                           \
            if (S[j] > thr && S[j] > 0) {                                                     \
                for (int i = 0; i < 3; ++i) U[3 * i + j] /= S[j];                             \
                ++rank;                                                                       \
            }                                                                                 \
        if (rank == 0) { REAL I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; memcpy(U, I, sizeof(I)); }  \
        if (rank == 1) { /* any unit vector orthogonal to column 0 */                         \
            REAL a0 = U[0], a1 = U[3], a2 = U[6], b0, b1, b2;                                 \
            if (FABS(a0) <= FABS(a1) && FABS(a0) <= FABS(a2)) { b0 = 0; b1 = -a2; b2 = a1; }  \
            else if (FABS(a1) <= FABS(a2)) { b0 = -a2; b1 = 0; b2 = a0; }                     \
            else { b0 = -a1; b1 = a0; b2 = 0; }                                               \
            REAL nb = SQRT(b0 * b0 + b1 * b1 + b2 * b2);                                      \
            U[1] = b0 / nb; U[4] = b1 / nb; U[7] = b2 / nb;                                   \
            rank = 2;                                                                         \
        }                                                                                     \
        if (rank == 2) { /* column 2 = column 0 x column 1 */                                 \
            U[2] = U[3] * U[7] - U[6] * U[4];                                                 \
            U[5] = U[6] * U[1] - U[0] * U[7];                                                 \
            U[8] = U[0] * U[4] - U[3] * U[1];                                                 \
        }                                                                                     \
    }

DEFINE_SVD3(double, orc_svd3_d, DBL_EPSILON, 1e-300, sqrt, fabs)
DEFINE_SVD3(float, orc_svd3_f, FLT_EPSILON, 1e-37f, sqrtf, fabsf)

#define DEFINE_KABSCH(REAL, NAME, SVD)                                                        \
    static void NAME(const REAL *src, const REAL *tgt, int n, REAL T[16])                     \
    {                                                                                         \
        REAL cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};                                            \
        for (int i = 0; i < n; ++i)                                                           \
            for (int d = 0; d < 3; ++d) { cs[d] += src[3 * (size_t)i + d]; ct[d] += tgt[3 * (size_t)i + d]; } \
        for (int d = 0; d < 3; ++d) { cs[d] /= (REAL)n; ct[d] /= (REAL)n; }                   \
        REAL H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                                              \
        for (int i = 0; i < n; ++i) {                                                         \
            REAL a[3], b[3];                                                                  \
            for (int d = 0; d < 3; ++d) { a[d] = src[3 * (size_t)i + d] - cs[d]; b[d] = tgt[3 * (size_t)i + d] - ct[d]; } \
            for (int r = 0; r < 3; ++r)                                                       \
                for (int c = 0; c < 3; ++c) H[3 * r + c] += a[r] * b[c];                      \
        }                                                                                     \
        REAL U[9], S[3], V[9], R[9];                                                          \
        SVD(H, U, S, V);                                                                      \
        for (int pass = 0; pass < 2; ++pass) {                                                \
            for (int r = 0; r < 3; ++r)                                                       \
                for (int c = 0; c < 3; ++c)                                                   \
                    R[3 * r + c] = V[3 * r + 0] * U[3 * c + 0] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2]; \
            REAL det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]); \
            if (pass == 1 || !(det < 0)) break;                                               \
            V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8];                                         \
        }                                                                                     \
        for (int i = 0; i < 16; ++i) T[i] = 0;                                                \
        T[15] = 1;                                                                            \
        for (int r = 0; r < 3; ++r) {                                                         \
            for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];                          \
            T[4 * r + 3] = ct[r] - (R[3 * r + 0] * cs[0] + R[3 * r + 1] * cs[1] + R[3 * r + 2] * cs[2]); \
        }                                                                                     \
    }

DEFINE_KABSCH(float, kabsch_f, orc_svd3_f)
DEFINE_KABSCH(double, kabsch_d, orc_svd3_d)

/* exported for icp.c */
void orc_kabsch_f_(const float *src, const float *tgt, int n, float T[16]) { kabsch_f(src, tgt, n, T); }
void orc_kabsch_d_(const double *src, const double *tgt, int n, double T[16]) { kabsch_d(src, tgt, n, T); }

#include <stdlib.h>
void orc_kabsch(const float *src, const float *tgt, int n, int precise, double T[16])
{
    if (!precise) {
        float Tf[16];
        kabsch_f(src, tgt, n, Tf);
        for (int i = 0; i < 16; ++i) T[i] = Tf[i];
        return;
    }
    double *s = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    double *t = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    for (size_t i = 0; i < 3 * (size_t)n; ++i) { s[i] = src[i]; t[i] = tgt[i]; }
    kabsch_d(s, t, n, T);
    free(s);
    free(t);
}
