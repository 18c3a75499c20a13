// sf_icp.hip — ICP on the device (gfx950): fused transform + exact 1-NN + normal-equation
// accumulation, fixed-order reduction, on-device solve and loop control.
//
// Replaces ICPPointToPoint::calculateAlignment and its helpers
// (localization/src/icp_point_to_point.cpp:57-84,99-170,185-254) and the Open3D
// registration_icp call of localization_python/localization_python/localization_node.py:
// 233-237; adds the point-to-plane Gauss-Newton extension.
//
// Once per alignment (O3D_P2P / P2PLANE, enough work to pay for it): every scan's points are
// ordered by the map cell they fall in under the initial pose (k_query_keys, rocPRIM radix sort,
// k_gather_queries), so that the scans in flight walk the map together.
// Per iteration:
//   k_nn_red        one lane per source point: s = T*x0 (float64); the neighbour found by the
//                   point's last search is kept when a bound proved by that search shows it
//                   cannot have changed (bit-identical result), otherwise exact grid 1-NN of
//                   float32(s) (wave-cooperative, sf_nn.hpp); the pair's contribution in
//                   float64, wave64 transposing-butterfly reduce (permlane swaps + DPP) -> LDS
//                   across the 4 waves -> one partial record per workgroup in a slab (no float
//                   atomics: bitwise reproducible).  Workgroups are placed XCD-aware: every XCD
//                   sweeps its own contiguous part of the (cell-ordered) chunks for all scans in
//                   flight, so they share map lines in that XCD's L2.
//   k_reduce_solve  one workgroup per scan: fixed-order sum of the slab, then one lane
//                   solves (3x3 Jacobi SVD Kabsch / 6x6 LDL^T), composes T and updates the
//                   convergence flags in device memory — no host round trip per iteration.
// REF_CPP keeps the reference's data-dependent control flow (lazy re-search, shrinking
// source set, float32 point updates) with device-side flags; kernels whose phase is not
// active return immediately, so the launch list is static and graph-capturable.
#include "sf_common.hpp"
#include "sf_nn.hpp"
#include "sf_tile.hpp"
#include "sf_order.hpp"
#include "sf_p2p.hpp"

#include <cfloat>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <mutex>
#include <string>

namespace {

constexpr int BLK = 256;
constexpr int NREC_P2P = 17;
constexpr int NREC_PLANE = 30;
constexpr int REC_STRIDE = 32; // doubles per scan in the exchange buffer

struct IcpState {
    double T[16];
    double rec[REC_STRIDE];
    double fitness, rmse, prev_fitness, prev_rmse;
    float last_error, err_pending;
    float step[12];
    int step_pending;
    int iterations, done, research, n_corr, n_research, flags, converged;
    int n_points; // single-scan REF_CPP with the count in device memory: the count the alignment ran on (sf_icp_source_count)
    int cache_live; // launch list, O3D_P2P / P2PLANE: the scan's neighbour-cache entries have been written since the last (re)start
    int froze_launch; // frozen pairs: index of the launch in which the scan FIRST froze, -1 if it never did (the host's schedule learns from it)
    double T_list[12]; // sharded path: the pose this rank's owned-query arrays were built at
    double motion;     // launch list: upper bound on how far any point of the source batch has moved since the alignment began (sum over the pose updates)
};

struct IcpParams {
    float max_corr;   // reference: compared against d2 directly (icp_point_to_point.cpp:70)
    float accept;
    float eps;
    int num_iters;
};

inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)sf::div_up(n > 0 ? n : 1, b); }

// ------------------------------------------------------------------ small device linear algebra (one lane)
// Every loop below has compile-time bounds and is fully unrolled so the little matrices
// live in registers (runtime-indexed arrays would go to scratch memory).
__device__ __forceinline__ void mat4_mul(const double A[16], const double B[16], double C[16])
{
    double R[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            R[4 * r + c] = A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c] + A[4 * r + 2] * B[8 + c] + A[4 * r + 3] * B[12 + c];
#pragma unroll
    for (int i = 0; i < 16; ++i) C[i] = R[i];
}

// 1 / sqrt(x) for x > 0, finite: the hardware estimate and two Newton steps (full float64 accuracy to an ulp or two).
// The solve runs on ONE lane while the rest of its workgroup -- on the per-scan path the whole grid -- waits, so the
// float64 divide / sqrt expansions (a few dozen dependent instructions each) are what an iteration's controller costs:
// measured 6.2 us per controller step with divides and square roots, 3.2-4.0 us with this.
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}

// One Hestenes rotation of columns P, Q (the smaller angle): with al = |u_P|^2, be = |u_Q|^2, ga = u_P . u_Q,
// cos 2t = |be - al| / w, sin 2t = 2 ga sgn(be - al) / w, w = sqrt((be - al)^2 + 4 ga^2); c = sqrt((1 + cos 2t) / 2),
// s = sin 2t / (2 c) -- two reciprocal square roots, no divide.
template <int P, int Q>
__device__ __forceinline__ bool jacobi_pair(double (&u)[9], double (&v)[9])
{
    double al = 0, be = 0, ga = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        al += u[3 * i + P] * u[3 * i + P];
        be += u[3 * i + Q] * u[3 * i + Q];
        ga += u[3 * i + P] * u[3 * i + Q];
    }
    if (ga * ga <= (DBL_EPSILON * DBL_EPSILON) * (al * be) || fabs(ga) < 1e-150) return false;
    const double tau = be - al;
    const double r = rsqrt_nr(tau * tau + 4.0 * ga * ga);
    const double c2 = 0.5 + 0.5 * fabs(tau) * r;
    const double rc = rsqrt_nr(c2);
    const double c = c2 * rc, s = (tau >= 0 ? ga : -ga) * r * rc;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a = u[3 * i + P], b = u[3 * i + Q];
        u[3 * i + P] = c * a - s * b;
        u[3 * i + Q] = s * a + c * b;
        a = v[3 * i + P]; b = v[3 * i + Q];
        v[3 * i + P] = c * a - s * b;
        v[3 * i + Q] = s * a + c * b;
    }
    return true;
}

template <int A, int B>
__device__ __forceinline__ void swap_cols_if_less(double (&s)[3], double (&u)[9], double (&v)[9], double (&inv)[3])
{
    if (s[B] > s[A]) {
        double t = s[A]; s[A] = s[B]; s[B] = t;
        t = inv[A]; inv[A] = inv[B]; inv[B] = t;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            t = u[3 * i + A]; u[3 * i + A] = u[3 * i + B]; u[3 * i + B] = t;
            t = v[3 * i + A]; v[3 * i + A] = v[3 * i + B]; v[3 * i + B] = t;
        }
    }
}

// one-sided Jacobi SVD of a 3x3 (row-major), S descending — stands in for
// Eigen::JacobiSVD<Matrix3f> at icp_point_to_point.cpp:137 (float64 here)
__device__ __forceinline__ void svd3(const double (&A)[9], double (&U)[9], double (&S)[3], double (&V)[9])
{
#pragma unroll
    for (int i = 0; i < 9; ++i) { U[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = jacobi_pair<0, 1>(U, V);
        rotated = jacobi_pair<0, 2>(U, V) || rotated;
        rotated = jacobi_pair<1, 2>(U, V) || rotated;
        if (!rotated) break;
    }
    double invS[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double n2 = U[j] * U[j] + U[3 + j] * U[3 + j] + U[6 + j] * U[6 + j];
        invS[j] = n2 > 1e-300 ? rsqrt_nr(n2) : 0.0;
        S[j] = n2 * invS[j];
    }
    swap_cols_if_less<0, 1>(S, U, V, invS);
    swap_cols_if_less<0, 2>(S, U, V, invS);
    swap_cols_if_less<1, 2>(S, U, V, invS);
    const double thr = S[0] * DBL_EPSILON * 8;
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (S[j] > thr && S[j] > 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) U[3 * i + j] *= invS[j];
            ++rank;
        }
    if (rank == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    if (rank == 1) {
        const double a0 = U[0], a1 = U[3], a2 = U[6];
        double b0, b1, b2;
        if (fabs(a0) <= fabs(a1) && fabs(a0) <= fabs(a2)) { b0 = 0; b1 = -a2; b2 = a1; }
        else if (fabs(a1) <= fabs(a2)) { b0 = -a2; b1 = 0; b2 = a0; }
        else { b0 = -a1; b1 = a0; b2 = 0; }
        const double nb = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
        U[1] = b0 / nb; U[4] = b1 / nb; U[7] = b2 / nb;
        rank = 2;
    }
    if (rank == 2) {
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

// Kabsch step (icp_point_to_point.cpp:112-159) from the uncentred sums of the record:
// rec[0]=n, [1..3]=sum s, [4..6]=sum t, [7..15]=sum s t^T.  H = sum s t^T - n cs ct^T.
__device__ __forceinline__ void kabsch_from_record(const double *rec, double (&T)[16])
{
    const double n = rec[0];
    double cs[3], ct[3], H[9];
    const double inv_n = 1.0 / n;
#pragma unroll
    for (int d = 0; d < 3; ++d) { cs[d] = rec[1 + d] * inv_n; ct[d] = rec[4 + d] * inv_n; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) H[3 * r + c] = rec[7 + 3 * r + c] - n * cs[r] * ct[c];
    double U[9], S[3], V[9], R[9];
    svd3(H, U, S, V);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
        const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (pass == 1 || !(det < 0)) break;
        V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = 0;
    T[15] = 1;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
        T[4 * r + 3] = ct[r] - (R[3 * r] * cs[0] + R[3 * r + 1] * cs[1] + R[3 * r + 2] * cs[2]);
    }
}

__device__ __forceinline__ int ldlt6(const double (&A)[36], const double (&b)[6], double (&x)[6])
{
    double L[36], D[6], y[6];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        if (!(fabs(d) > 0) || !isfinite(d)) bad = true;
        D[j] = d;
        L[6 * j + j] = 1;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = A[6 * i + j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = v / d;
        }
    }
    if (bad) return -1;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] /= D[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * x[k];
        x[i] = v;
    }
    return 0;
}

// R = Rz(v2) Ry(v1) Rx(v0), t = v[3:6] (Open3D TransformVector6dToMatrix4d)
__device__ __forceinline__ void vec6_to_mat4(const double (&v)[6], double (&T)[16])
{
    double ca, sa, cb, sb, cg, sg; // one range reduction per angle
    sincos(v[0], &sa, &ca);
    sincos(v[1], &sb, &cb);
    sincos(v[2], &sg, &cg);
    const double R[9] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa,
                         sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa,
                         -sb, cb * sa, cb * ca};
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = 0;
    T[15] = 1;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
        T[4 * r + 3] = v[3 + r];
    }
}

// ------------------------------------------------------------------ reductions
// Wave-level sum of 32 doubles per lane in ~125 VALU instructions and no LDS traffic
// (a plain xor-shuffle reduction of 30 doubles costs 360 ds_bpermute + 180 adds per wave and
// dominated the kernel's instruction issue — profiles/).  Transposing butterfly: at every
// step each lane keeps HALF of its values and adds its partner's copy of that half, so the
// number of live values halves while the number of lanes summed doubles:
//   32 -> 16  partner l ^ 32   v_permlane32_swap   (gfx950)
//   16 ->  8  partner l ^ 16   v_permlane16_swap   (gfx950)
//    8 ->  4  partner l ^ 8    DPP row_ror:8
//    4 ->  2  partner 7-(l&7)  DPP row_half_mirror
//    2 ->  1  partner l ^ 2    DPP quad_perm [2,3,0,1]
//    final    partner l ^ 1    DPP quad_perm [1,0,3,2]
// Afterwards lane l holds the 64-lane total of component (l >> 1).  Fixed order => bitwise
// reproducible.
__device__ __forceinline__ double swap_add_32(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

__device__ __forceinline__ double swap_add_16(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

template <int DPP_CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2loint(v), DPP_CTRL, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2hiint(v), DPP_CTRL, 0xf, 0xf, false);
    return __hiloint2double((int)hi, (int)lo);
}

template <int DPP_CTRL>
__device__ __forceinline__ double dpp_keep_add(double a, double b, bool upper)
{
    const double keep = upper ? b : a, give = upper ? a : b;
    return keep + dpp_f64<DPP_CTRL>(give);
}

__device__ __forceinline__ double wave_reduce_32(double (&v)[32])
{
    const int lane = threadIdx.x & 63;
    double w16[16], w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) w16[i] = swap_add_32(v[i], v[i + 16]);
#pragma unroll
    for (int i = 0; i < 8; ++i) w8[i] = swap_add_16(w16[i], w16[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = dpp_keep_add<0x128>(w8[i], w8[i + 4], (lane & 8) != 0);   // row_ror:8
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = dpp_keep_add<0x141>(w4[i], w4[i + 2], (lane & 4) != 0);   // row_half_mirror
    const double w1 = dpp_keep_add<0x4e>(w2[0], w2[1], (lane & 2) != 0);                          // quad_perm [2,3,0,1]
    return w1 + dpp_f64<0xb1>(w1);                                                                // quad_perm [1,0,3,2]
}

// 16-value form of the same butterfly (half the live registers): afterwards lane l holds the
// 64-lane total of component (l >> 2) & 15.
__device__ __forceinline__ double wave_reduce_16(double (&v)[16])
{
    const int lane = threadIdx.x & 63;
    double w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) w8[i] = swap_add_32(v[i], v[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = swap_add_16(w8[i], w8[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = dpp_keep_add<0x128>(w4[i], w4[i + 2], (lane & 8) != 0);   // row_ror:8
    double z = dpp_keep_add<0x141>(w2[0], w2[1], (lane & 4) != 0);                                // row_half_mirror
    z = z + dpp_f64<0x4e>(z);                                                                     // quad_perm [2,3,0,1]
    return z + dpp_f64<0xb1>(z);                                                                  // quad_perm [1,0,3,2]
}

// every lane gets the 64-lane total of one value
__device__ __forceinline__ double wave_reduce_1(double v)
{
    v = swap_add_32(v, v);
    v = swap_add_16(v, v);
    v = v + dpp_f64<0x128>(v);
    v = v + dpp_f64<0x141>(v);
    v = v + dpp_f64<0x4e>(v);
    return v + dpp_f64<0xb1>(v);
}

// Slab layout: partials[(scan * nblocks + block) * 32 + component] — a workgroup's record
// is one contiguous 256-byte row, so both the store here and the column sums below are
// coalesced.
template <int NREC>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NREC], double *__restrict__ dst)
{
    double v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = c < NREC ? acc[c] : 0.0;
    const double total = wave_reduce_32(v);
    __shared__ double s[BLK / 64][32];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((lane & 1) == 0) s[w][lane >> 1] = total;
    __syncthreads();
    if (threadIdx.x < NREC) {
        const int c = threadIdx.x;
        dst[c] = ((s[0][c] + s[1][c]) + s[2][c]) + s[3][c];
    }
}

// Fixed-order sum of one scan's slab by RBLK = 1024 threads: thread (slice s, component c)
// adds rows s, s+32, s+64, ... then the 32 slices are added in order.  Result in rec[]
// (shared).  Deterministic: no atomics, the order depends only on nblocks.
constexpr int RBLK = 1024;
// NT = threads of the calling workgroup (a multiple of 32, at most 1024): with fewer than 1024 a thread takes several
// slices one after the other -- every (slice, component) sum is the same expression, so the result is bit-identical
// for every NT.
template <int NREC, int NT = RBLK>
__device__ __forceinline__ void reduce_partials(const double *__restrict__ part, int nblocks, double *rec)
{
    __shared__ double sl[32][REC_STRIDE + 1];
    const int c = threadIdx.x & 31;
    for (int sidx = threadIdx.x >> 5; sidx < 32; sidx += NT / 32) {
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (c < NREC) {
            int b = sidx;
            for (; b + 96 < nblocks; b += 128) {
                const double a0 = part[(size_t)b * REC_STRIDE + c], a1 = part[(size_t)(b + 32) * REC_STRIDE + c];
                const double a2 = part[(size_t)(b + 64) * REC_STRIDE + c], a3 = part[(size_t)(b + 96) * REC_STRIDE + c];
                v0 += a0; v1 += a1; v2 += a2; v3 += a3;
            }
            for (; b < nblocks; b += 32) v0 += part[(size_t)b * REC_STRIDE + c];
        }
        sl[sidx][c] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    if (threadIdx.x < REC_STRIDE) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) v += sl[k][threadIdx.x];
        rec[threadIdx.x] = threadIdx.x < NREC ? v : 0.0;
    }
    __syncthreads();
}

// ------------------------------------------------------------------ helper kernels
// source points as SoA (coalesced lane loads of the kernels) and as float4 records (ONE 16-byte access
// per point for the gather of the query ordering, instead of three scattered 4-byte ones)
__global__ void k_soa_from_aos(const float *__restrict__ aos, int64_t n, float *__restrict__ x, float *__restrict__ y, float *__restrict__ z, float4 *__restrict__ rec)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = aos[3 * i], b = aos[3 * i + 1], c = aos[3 * i + 2];
    x[i] = a; y[i] = b; z[i] = c;
    rec[i] = make_float4(a, b, c, 0.0f);
}

// small batches carry their initial transforms in the kernel arguments (no copy, nothing for the host to wait on)
constexpr int INIT_ARGS_MAX = 8;
struct InitArgs { double T[INIT_ARGS_MAX][16]; };

__global__ void k_state_init(IcpState *__restrict__ st, const double *__restrict__ inits, InitArgs args, int batch)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    IcpState s;
    for (int i = 0; i < 16; ++i) s.T[i] = inits ? inits[16 * b + i] : args.T[b][i];
    for (int i = 0; i < REC_STRIDE; ++i) s.rec[i] = 0;
    s.fitness = s.rmse = s.prev_fitness = s.prev_rmse = 0;
    s.last_error = FLT_MAX; // icp_point_to_point.cpp:205
    s.err_pending = 0;
    for (int i = 0; i < 12; ++i) s.step[i] = 0;
    s.step_pending = 0;
    s.iterations = s.done = s.research = s.n_corr = s.n_research = s.flags = s.converged = 0;
    s.n_points = -1;
    s.cache_live = 0;
    s.froze_launch = -1;
    s.motion = 0.0;
    for (int i = 0; i < 12; ++i) s.T_list[i] = s.T[i];
    st[b] = s;
}

// ------------------------------------------------------------------ fused transform + NN + accumulate
// MODE 1: point-to-point record (17):  n, sum s[3], sum t[3], sum s t^T[9], sum d2
// MODE 2: point-to-plane record (30):  n, sum r^2, JtJ upper[21], Jtr[6], sum d2
// Q queries per lane (grid.x = ceil(n / (256 * Q)), grid.y = scans in the batch), one after the
// other: neighbour (reuse certificate or search), then the contributions of the lane's pairs are
// added and go through ONE wave reduction.  Q is part of the summation order, so it is a function of
// the scan size alone (sf_icp::qpl): scans up to 131 072 points -- everything the single-launch
// kernels can take, whose rows are 256 points -- use Q = 1 and match them bit for bit; larger scans
// use Q = SF_WIDE_QPL.  Round 1 measured Q = 2 within 1 % of Q = 1 when every query searched; since
// the neighbour reuse most launches of an alignment only verify, and a verifying wave is
// vector-issue bound with the record reduction (~160 of its ~390 instructions) the largest item:
// Q = 2 halves that per query.
#ifndef SF_WIDE_QPL
#define SF_WIDE_QPL 2
#endif
constexpr int64_t WIDE_SCAN_POINTS = 131072;
constexpr int64_t WIDE_AUTO_POINTS = 65536; // see sf_icp::wide_auto
constexpr int VERIFY_FROM_SEARCH = 4; // wide scans: from the fifth launch of an alignment on a wave first tries to verify all its queries at once
#ifndef NN_RED_WAVES
#define NN_RED_WAVES 4
#endif
constexpr int NN_STATS_SHARDS = 256; // counters of one profiled launch

struct LanePair {
    double sx, sy, sz; // the transformed scan point (float64)
    float px, py, pz;  // its neighbour
    float4 tn;         // the neighbour's normal (MODE 2)
    bool ok;
};
struct QueryIn;

// Neighbour reuse with an exactness certificate.  The last full search of this query found neighbour c1 = (point,
// index) and proved every OTHER map point at least D away from where the query was then.  Since then the query has
// moved by at most M_now - M_then, where M (IcpState::motion) adds up, over the pose updates of the alignment, the
// largest displacement any point of the source batch's bounding box can have had (an affine map moves a box's corners
// the most).  If |q - p| < D - (M_now - M_then) (triangle inequality, with rounding margins: both positions are
// float32 roundings of float64 products) no other point can be nearer, so the search would return the same
// neighbour -- it is skipped, the result is bit-identical.  What is kept per query is E = D + M_then, ONE float
// (in the w of the cached normal; its own array in MODE 1), not the position of the last search: a verifying wave
// streams 12 B of scan point + 16 B neighbour + 16 B normal-and-E = 44 B per query (round 2: 60).  ICP steps shrink
// geometrically, so after the first few iterations whole waves certify; a wave with any lane left runs the search
// for just those lanes.  E <= 0 marks an entry without a search behind it.
// Returns whether the query must search; hit / tn = the certified pair (or "none yet"), seed = where a search may start.
__device__ __forceinline__ bool reuse_certificate(bool valid, float qx, float qy, float qz, float thr, float m_now, float e, const float4 &c1, const float4 &c2, sf::NNHit &hit,
                                                  float4 &tn, sf::NNHit &seed)
{
    hit.d2 = thr;
    hit.j = -1;
    hit.px = hit.py = hit.pz = 0.0f;
    hit.lb2 = 0.0f;
    tn = make_float4(0.f, 0.f, 0.f, 0.f); // the neighbour's normal (MODE 2)
    valid = valid && isfinite(qx) && isfinite(qy) && isfinite(qz); // a non-finite query has no neighbour and needs no search to know it
    bool need = valid;
    seed = sf::NNHit{0.0f, -1, 0.0f, 0.0f, 0.0f, 0.0f};
    if (valid) {
        if (e > 0.0f) {
            const int32_t jc = __float_as_int(c1.w);
            // e - m_now bounds D - |q - q_then| from below; the float32 roundings of q and q_then are charged here
            const float reach = e - m_now * 1.000002f - (fabsf(qx) + fabsf(qy) + fabsf(qz) + 1.0f) * 1.3e-7f - (e + m_now) * 5.0e-7f - 1.0e-6f; // (the last but one: float32 arithmetic on E and M themselves, whatever their size)
            if (jc >= 0) {
                const float d2n = sf::l2_simple(qx, qy, qz, c1.x, c1.y, c1.z);
                if (sqrtf(d2n) * 1.0001f + 1.0e-6f < reach) {
                    need = false;
                    if (d2n < thr) {
                        hit.d2 = d2n; hit.j = jc; hit.px = c1.x; hit.py = c1.y; hit.pz = c1.z;
                        tn = c2;
                    }
                } else {
                    // not certified: the search still starts from the old neighbour's current distance instead of the
                    // acceptance threshold (exact all the same -- sf_nn.hpp -- and ranges beyond it are pruned unvisited;
                    // measured: 2 % on the second to fourth launch of an alignment)
                    seed.d2 = d2n; seed.j = jc; seed.px = c1.x; seed.py = c1.y; seed.pz = c1.z;
                }
            } else if (sqrtf(thr) * 1.0001f + 1.0e-6f < reach) {
                need = false; // still nothing within the acceptance radius
            }
        }
    }
    return need;
}

// The same certificate with the position of the last search kept per query (c0 = (position, D)): the single-launch
// kernels hold their cache in registers, where the position costs no traffic and the bound is the query's own motion.
__device__ __forceinline__ bool reuse_certificate_pos(bool valid, float qx, float qy, float qz, float thr, const float4 &c0, const float4 &c1, const float4 &c2, sf::NNHit &hit,
                                                  float4 &tn, sf::NNHit &seed)
{
    hit.d2 = thr;
    hit.j = -1;
    hit.px = hit.py = hit.pz = 0.0f;
    hit.lb2 = 0.0f;
    tn = make_float4(0.f, 0.f, 0.f, 0.f); // the neighbour's normal (MODE 2)
    bool need = valid;
    seed = sf::NNHit{0.0f, -1, 0.0f, 0.0f, 0.0f, 0.0f};
    if (valid) {
        if (c0.w > 0.0f) {
            const int32_t jc = __float_as_int(c1.w);
            const float dx = qx - c0.x, dy = qy - c0.y, dz = qz - c0.z;
            const float reach = c0.w * 0.9999f - sqrtf(dx * dx + dy * dy + dz * dz) * 1.0001f - 1.0e-6f;
            if (jc >= 0) {
                const float d2n = sf::l2_simple(qx, qy, qz, c1.x, c1.y, c1.z);
                if (sqrtf(d2n) * 1.0001f + 1.0e-6f < reach) {
                    need = false;
                    if (d2n < thr) {
                        hit.d2 = d2n; hit.j = jc; hit.px = c1.x; hit.py = c1.y; hit.pz = c1.z;
                        tn = c2;
                    }
                } else {
                    // not certified: the search still starts from the old neighbour's current distance instead of the
                    // acceptance threshold (exact all the same -- sf_nn.hpp -- and ranges beyond it are pruned unvisited;
                    // measured: 2 % on the second to fourth launch of an alignment)
                    seed.d2 = d2n; seed.j = jc; seed.px = c1.x; seed.py = c1.y; seed.pz = c1.z;
                }
            } else if (sqrtf(thr) * 1.0001f + 1.0e-6f < reach) {
                need = false; // still nothing within the acceptance radius
            }
        }
    }
    return need;
}

// what a lane reads of its query: the scan point under the current pose (float64 products, rounded to the float32 the
// search works in), whether this rank owns it, and -- once the scan's cache entries are written -- its cached pair
struct QueryIn {
    double sx, sy, sz;
    float qx, qy, qz;
    float4 c1, c2; // (neighbour, index), (neighbour's normal, E)   [MODE 1: E alone, from its own array]
    float e;
    size_t o;
    bool valid;
};

// no side effects: the loads of several queries of a lane can all be in flight together
template <int MODE, bool SHARD>
__device__ __forceinline__ QueryIn query_in(const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z, int n, int b, const IcpState *S, float xlo,
                                            float xhi, const uint32_t *__restrict__ own_off, const float4 *__restrict__ qcache, int64_t cache_n, bool cache_live, int slot,
                                            int n_live)
{
    QueryIn q;
    q.sx = q.sy = q.sz = 0.0;
    q.qx = q.qy = q.qz = 0.0f;
    q.c1 = q.c2 = make_float4(0.f, 0.f, 0.f, 0.f);
    q.e = 0.0f;
    q.o = 0;
    q.valid = false;
    if (slot < n_live) {
        const size_t o = SHARD ? (size_t)own_off[b] + (size_t)slot : (size_t)b * n + (size_t)slot;
        q.o = o;
        if (cache_live) {
            q.c1 = qcache[o];
            if (MODE == 2) { q.c2 = qcache[(size_t)cache_n + o]; q.e = q.c2.w; }
            else q.e = reinterpret_cast<const float *>(qcache + (size_t)cache_n)[o]; // MODE 1: E has an array of its own
        }
        const double x0 = X0x[o], y0 = X0y[o], z0 = X0z[o];
        q.sx = S->T[0] * x0 + S->T[1] * y0 + S->T[2] * z0 + S->T[3];
        q.sy = S->T[4] * x0 + S->T[5] * y0 + S->T[6] * z0 + S->T[7];
        q.sz = S->T[8] * x0 + S->T[9] * y0 + S->T[10] * z0 + S->T[11];
        q.qx = (float)q.sx; q.qy = (float)q.sy; q.qz = (float)q.sz;
        q.valid = !SHARD || (q.qx >= xlo && q.qx < xhi);
    }
    return q;
}

__device__ __forceinline__ LanePair make_pair(const QueryIn &q, const sf::NNHit &hit, const float4 &tn)
{
    LanePair P;
    P.sx = q.sx; P.sy = q.sy; P.sz = q.sz;
    P.px = hit.px; P.py = hit.py; P.pz = hit.pz;
    P.tn = tn;
    P.ok = hit.j >= 0;
    return P;
}

template <int MODE, bool WINDOW, bool SHARD>
__device__ __forceinline__ LanePair nn_pair(const SfGrid &g, const SfWindow &w, const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z,
                                            int n, int b, const IcpState *S, float thr, float xlo, float xhi, const uint32_t *__restrict__ own_off,
                                            float4 *__restrict__ qcache, int64_t cache_n, int slot, int n_live, sf::WaveNN *ws, uint32_t *__restrict__ stats,
                                            sf::PhaseClock *pc = nullptr)
{
    // once the scan's entries have been written (every lane writes its entry in the first launch after a (re)start,
    // searched or not, so the cache is never reset) the cache streams are requested together with the scan point: one
    // round trip for a verifying wave instead of dependent ones
    const bool cache_live = qcache != nullptr && S->cache_live != 0;
    const QueryIn q = query_in<MODE, SHARD>(X0x, X0y, X0z, n, b, S, xlo, xhi, own_off, qcache, cache_n, cache_live, slot, n_live);
    const float qx = q.qx, qy = q.qy, qz = q.qz;
    const size_t o = q.o;
    float *ecache = reinterpret_cast<float *>(qcache + (size_t)cache_n);
    const float m_now = (float)S->motion;
    sf::NNHit hit;
    float4 tn;
    sf::NNHit seed;
    const bool need = reuse_certificate(q.valid, qx, qy, qz, thr, m_now, q.e, q.c1, q.c2, hit, tn, seed);
    // every lane takes part in the search (lanes without a query still execute other lanes' tasks)
    const unsigned long long need_mask = __ballot(need);
    if (stats && (threadIdx.x & 63) == 0 && need_mask) { // profiling only (integer counters: order independent); sharded: one address would serialise 100 k waves
        uint32_t *sh = stats + 2 * ((blockIdx.x + blockIdx.y * gridDim.x) & (NN_STATS_SHARDS - 1));
        atomicAdd(&sh[0], (uint32_t)__popcll(need_mask));
        atomicAdd(&sh[1], 1u);
    }
    SF_PH(pc, 0);
    if (need_mask != 0ull) {
        const sf::NNHit h = sf::nn_search_wave<WINDOW>(g, w, need, qx, qy, qz, thr, ws, seed, pc);
        if (need) {
            hit = h;
            // only the winner's normal is fetched after the search; its coordinates come in registers
            if (MODE == 2 && h.j >= 0) tn = g.nrm[h.j];
            if (qcache) {
                // E = D + M, rounded down: D = sqrt(lb2) is the proven distance of every other point
                const float en = fmaxf(sqrtf(h.lb2) * 0.9999f + m_now * 0.999998f - 1.0e-6f, 1.0e-30f);
                qcache[o] = make_float4(h.px, h.py, h.pz, __int_as_float(h.j));
                if (MODE == 2) qcache[(size_t)cache_n + o] = make_float4(tn.x, tn.y, tn.z, en);
                else ecache[o] = en;
            }
        }
    }
    if (qcache && !cache_live && !need && slot < n_live) { // first launch after a (re)start: a lane that did not search (outside the slab) leaves "no search behind it"
        if (MODE == 2) qcache[(size_t)cache_n + o] = make_float4(0.f, 0.f, 0.f, 0.f);
        else ecache[o] = 0.0f;
    }
    SF_PH(pc, 6);
    return make_pair(q, hit, tn);
}

// lanes without a correspondence contribute exact zeros: their point, neighbour and normal are zeroed once (a non-finite
// dead query must not turn 0 * s into NaN), after which every term is plain arithmetic on zeros
struct PairTerms { double wgt, ux, uy, uz, tx, ty, tz, ex, ey, ez, d2, r; double J[6]; };
template <int MODE>
__device__ __forceinline__ PairTerms pair_terms(const LanePair &P)
{
    PairTerms t;
    const bool ok = P.ok;
    t.wgt = ok ? 1.0 : 0.0;
    t.ux = ok ? P.sx : 0.0; t.uy = ok ? P.sy : 0.0; t.uz = ok ? P.sz : 0.0;
    t.tx = (double)(ok ? P.px : 0.0f); t.ty = (double)(ok ? P.py : 0.0f); t.tz = (double)(ok ? P.pz : 0.0f);
    t.ex = t.ux - t.tx; t.ey = t.uy - t.ty; t.ez = t.uz - t.tz;
    t.d2 = t.ex * t.ex + t.ey * t.ey + t.ez * t.ez;
    if (MODE == 2) {
        const double nx = (double)(ok ? P.tn.x : 0.0f), ny = (double)(ok ? P.tn.y : 0.0f), nz = (double)(ok ? P.tn.z : 0.0f);
        t.r = t.ex * nx + t.ey * ny + t.ez * nz;
        t.J[0] = t.uy * nz - t.uz * ny; t.J[1] = t.uz * nx - t.ux * nz; t.J[2] = t.ux * ny - t.uy * nx;
        t.J[3] = nx; t.J[4] = ny; t.J[5] = nz;
    } else {
        t.r = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) t.J[a] = 0.0;
    }
    return t;
}

// the 32 record slots of one pair, half h (0: slots 0..15, 1: slots 16..31), added to v.  The products go in as fused
// multiply-adds (float64 accumulation of this library's own sums -- no reference arithmetic to mirror here, and the vector
// issue slots are what the kernel runs out of: one instruction per term instead of two)
template <int MODE>
__device__ __forceinline__ void add_half(const PairTerms &t, int h, double (&v)[16])
{
    if (MODE == 1) {
        if (h == 0) {
            v[0] += t.wgt;
            v[1] += t.ux; v[2] += t.uy; v[3] += t.uz;
            v[4] += t.tx; v[5] += t.ty; v[6] += t.tz;
            v[7] = fma(t.ux, t.tx, v[7]); v[8] = fma(t.ux, t.ty, v[8]); v[9] = fma(t.ux, t.tz, v[9]);
            v[10] = fma(t.uy, t.tx, v[10]); v[11] = fma(t.uy, t.ty, v[11]); v[12] = fma(t.uy, t.tz, v[12]);
            v[13] = fma(t.uz, t.tx, v[13]); v[14] = fma(t.uz, t.ty, v[14]); v[15] = fma(t.uz, t.tz, v[15]);
        } else {
            v[0] += t.d2;
        }
    } else {
        // record[0..15] = n, sum r^2, JtJ (0,0) (0,1) .. (0,5) (1,1) .. (1,5) (2,2) (2,3) (2,4)
        // record[16..31] = JtJ (2,5) (3,3) .. (5,5), Jtr[6], sum d2, 0, 0
        if (h == 0) { v[0] += t.wgt; v[1] = fma(t.r, t.r, v[1]); }
        int k = 2;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c = a; c < 6; ++c) {
                if (h == 0 && k < 16) v[k] = fma(t.J[a], t.J[c], v[k]);
                if (h == 1 && k >= 16) v[k - 16] = fma(t.J[a], t.J[c], v[k - 16]);
                ++k;
            }
        }
        if (h == 1) {
#pragma unroll
            for (int a = 0; a < 6; ++a) v[7 + a] = fma(t.J[a], t.r, v[7 + a]);
            v[13] += t.d2;
        }
    }
}

#ifdef SF_PHASE_TRACE
__device__ unsigned long long g_phase_trace[sf::PH_SHARDS * sf::PH_SLOTS];
#endif
template <int MODE, bool WINDOW, bool SHARD, int Q>
__global__ __launch_bounds__(BLK, NN_RED_WAVES) void k_nn_red(SfGrid g, SfWindow w, const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z,
                                                int n, const IcpState *__restrict__ st, float thr, float xlo, float xhi, double *__restrict__ partials, int nblocks,
                                                const uint32_t *__restrict__ own_off, float4 *__restrict__ qcache, int64_t cache_n, uint32_t *__restrict__ stats)
{
    constexpr int NREC = MODE == 2 ? NREC_PLANE : NREC_P2P;
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs in launch order, so
    // linear id L runs on XCD L % 8.  Each XCD sweeps its own CONTIGUOUS eighth of the chunks
    // (chunk = 256 * Q consecutive queries of a scan), all scans of the batch adjacent in time:
    // with cell-ordered queries, chunk c of every scan covers about the same stretch of the map (to
    // within a chunk or so), so neighbouring chunks must meet in the same L2.  grid.x is padded
    // to a multiple of 8.
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int kk = L >> 3;
    const int b = kk % (int)gridDim.y;
    const int bx = (L & 7) * ((int)gridDim.x >> 3) + kk / (int)gridDim.y;
    if (bx >= nblocks) return;
    const IcpState *S = st + b;
    if (S->done) return;
    // sharded: X0x/y/z are this rank's compact arrays of owned-query candidates (slab widened by
    // the margin at the pose the arrays were built at, cell-ordered, scan b at [own_off[b], own_off[b+1]));
    // the exact slab predicate is still applied per lane
    const int n_live = SHARD ? (int)(own_off[b + 1] - own_off[b]) : n;
    if (SHARD && bx * (BLK * Q) >= n_live) return; // k_reduce_only reads only the rows that exist
    __shared__ sf::WaveNN nn_ws[BLK / 64];
    __shared__ double stage[BLK / 64][32];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the pairs are kept, not their terms (13 against 36 registers per query across the next search); the terms are
    // formed once per half below
#ifdef SF_PHASE_TRACE
    sf::PhaseClock pclk, *pc = &pclk;
    pclk.start();
#else
    sf::PhaseClock *pc = nullptr;
#endif
    LanePair P[Q];
    // Q > 1, from the launch on in which most waves only verify: the loads of ALL the lane's queries go out together (one
    // round trip per wave, Q times the bytes in flight) and, if every lane of the wave certifies every one of its
    // queries, the pairs come straight from them.  A wave with anything left to search drops what it loaded and takes the
    // queries one after the other as always -- the same pairs either way, so which path a wave takes changes no bit.
    bool fast = false;
    if (Q > 1 && qcache != nullptr && S->cache_live != 0 && S->n_research >= VERIFY_FROM_SEARCH) {
        const float m_now = (float)S->motion;
        bool any_need = false;
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const int slot = bx * (BLK * Q) + u * BLK + (int)threadIdx.x;
            const QueryIn q = query_in<MODE, SHARD>(X0x, X0y, X0z, n, b, S, xlo, xhi, own_off, qcache, cache_n, true, slot, n_live);
            sf::NNHit hit, seed;
            float4 tn;
            any_need = reuse_certificate(q.valid, q.qx, q.qy, q.qz, thr, m_now, q.e, q.c1, q.c2, hit, tn, seed) || any_need;
            P[u] = make_pair(q, hit, tn);
        }
        fast = __ballot(any_need) == 0ull;
    }
    if (!fast) {
        asm volatile("" ::: "memory"); // nothing loaded above stays live across the searches below
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const int slot = bx * (BLK * Q) + u * BLK + (int)threadIdx.x;
            P[u] = nn_pair<MODE, WINDOW, SHARD>(g, w, X0x, X0y, X0z, n, b, S, thr, xlo, xhi, own_off, qcache, cache_n, slot, n_live, &nn_ws[wv], stats, pc);
        }
    }
    // the lane's pairs added, reduced over the wave in two halves of 16 values (keeps the live
    // registers low enough for 4+ waves per SIMD), staged per wave in LDS
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.0;
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const PairTerms t = pair_terms<MODE>(P[u]);
            add_half<MODE>(t, h, v);
        }
        if (MODE == 1 && h == 1) {
            const double t1 = wave_reduce_1(v[0]);
            if (lane == 0) stage[wv][16] = t1;
        } else {
            const double t0 = wave_reduce_16(v);
            if ((lane & 3) == 0) stage[wv][16 * h + (lane >> 2)] = t0;
        }
    }
    SF_PH(pc, 7);
    __syncthreads();
    if (threadIdx.x < NREC) {
        const int c = threadIdx.x;
        double *dst = partials + ((size_t)b * nblocks + bx) * REC_STRIDE;
        dst[c] = ((stage[0][c] + stage[1][c]) + stage[2][c]) + stage[3][c];
    }
#ifdef SF_PHASE_TRACE
    SF_PH(pc, 8);
    pclk.count(14, 1u);
    if (lane == 0) {
        unsigned long long *dstp = g_phase_trace + (size_t)(L & (sf::PH_SHARDS - 1)) * sf::PH_SLOTS;
        for (int i = 0; i < sf::PH_SLOTS; ++i) atomicAdd(&dstp[i], (unsigned long long)pclk.acc[i]);
    }
#endif
}

// ------------------------------------------------------------------ query order
// Queries are independent and every sum has a fixed order, so the order of a scan's points is a
// free choice (it changes the record sums by rounding only, deterministically).  Ordering each
// scan by the grid cell of its point under the INITIAL pose makes chunk c of every scan in the
// batch cover the same stretch of the cell-sorted map; k_nn_red then places all the workgroups of
// one chunk on one XCD, so a map line is fetched from HBM once per batch instead of once per scan.
// position of a cell in the order the queries are walked in.  The map itself is stored z-major
// (rows along x, then y, then z), so a query's z +- 1 neighbour rows are far away in memory; walking
// the queries in blocks of ORDER_YBLK rows of y through ALL z layers brings the queries that touch
// those rows close together in time (L2), at no cost for the y +- 1 rows.
constexpr int ORDER_YBLK = 16; // measured: 8 and 16 equal (-5.6 % kernel time vs plain z-major order), 32 worse; blocking x as well gains nothing
__device__ __forceinline__ uint64_t order_cell(const SfGrid &g, int cx, int cy, int cz)
{
    const uint64_t yb = (uint64_t)(cy / ORDER_YBLK), yi = (uint64_t)(cy % ORDER_YBLK);
    return ((yb * (uint64_t)g.dim[2] + (uint64_t)cz) * ORDER_YBLK + yi) * (uint64_t)g.dim[0] + (uint64_t)cx;
}

// 10-bit bucket key of global query o (scan o / n): its cell under the scan's current pose, in walk order
struct CellKeyFn {
    SfGrid g;
    const float *x, *y, *z;
    const IcpState *st;
    int n, shift;
    const uint16_t *lut; // density-equalised buckets (order_lut_build): key = lut[walk position >> lut_shift], or nullptr
    int lut_shift;
    // The key is a locality hint, not a result: float32 arithmetic (the pose rounded once per workgroup) puts a
    // query that sits on a bucket border into one of the two buckets, deterministically.
    struct Point { float x, y, z; };
    struct Pose { float T[12]; };
    __device__ __forceinline__ Pose prepare(int b) const
    {
        Pose P;
#pragma unroll
        for (int k = 0; k < 12; ++k) P.T[k] = (float)st[b].T[k];
        return P;
    }
    __device__ __forceinline__ Point load(uint32_t o) const { return Point{x[o], y[o], z[o]}; }
    __device__ __forceinline__ uint32_t key(const Pose &P, const Point &p) const
    {
        const float qx = fmaf(P.T[0], p.x, fmaf(P.T[1], p.y, fmaf(P.T[2], p.z, P.T[3])));
        const float qy = fmaf(P.T[4], p.x, fmaf(P.T[5], p.y, fmaf(P.T[6], p.z, P.T[7])));
        const float qz = fmaf(P.T[8], p.x, fmaf(P.T[9], p.y, fmaf(P.T[10], p.z, P.T[11])));
        if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return sf::ORD_KEY_NONE; // non-finite queries go to the end of their scan
        const int cx = (int)fminf(fmaxf(floorf((qx - g.org[0]) * g.inv_h), 0.0f), (float)(g.dim[0] - 1));
        const int cy = (int)fminf(fmaxf(floorf((qy - g.org[1]) * g.inv_h), 0.0f), (float)(g.dim[1] - 1));
        const int cz = (int)fminf(fmaxf(floorf((qz - g.org[2]) * g.inv_h), 0.0f), (float)(g.dim[2] - 1));
        if (lut) return (uint32_t)lut[order_cell(g, cx, cy, cz) >> lut_shift];
        const uint64_t key = order_cell(g, cx, cy, cz) >> shift;
        return (uint32_t)(key < (uint64_t)(sf::ORD_KEY_NONE - 1u) ? key : (uint64_t)(sf::ORD_KEY_NONE - 1u));
    }
};

__global__ __launch_bounds__(sf::ORD_BLK) void k_order_hist(sf::OrderSrc s, CellKeyFn kf, uint16_t *__restrict__ keys, uint32_t *__restrict__ counts)
{
    sf::order_hist_body(s, kf, keys, counts);
}

// Buckets of equal MAP-POINT count instead of equal cell count.  The plain key cuts the walk order into 1 024 stretches of as
// many cells each: on a map of surfaces most of those are empty and the scan's queries crowd into the few that are not -- the
// order then resolves little where the points are.  The table maps a fine stretch of the walk order (up to 65 536 of them) to
// the share of the map's points that lie before it, scaled to the key range: built once per index (sf_icp_set_target*).
constexpr int ORDER_LUT_LOG2 = 16;
__global__ __launch_bounds__(256) void k_order_lut_hist(SfGrid g, int lut_shift, uint32_t *__restrict__ hist)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= g.n) return;
    const float4 p = g.pts[j];
    const int cx = (int)fminf(fmaxf(floorf((p.x - g.org[0]) * g.inv_h), 0.0f), (float)(g.dim[0] - 1));
    const int cy = (int)fminf(fmaxf(floorf((p.y - g.org[1]) * g.inv_h), 0.0f), (float)(g.dim[1] - 1));
    const int cz = (int)fminf(fmaxf(floorf((p.z - g.org[2]) * g.inv_h), 0.0f), (float)(g.dim[2] - 1));
    atomicAdd(&hist[order_cell(g, cx, cy, cz) >> lut_shift], 1u);
}
// one workgroup: exclusive prefix of the histogram -> key of every stretch
__global__ __launch_bounds__(1024) void k_order_lut_make(const uint32_t *__restrict__ hist, int nbins, uint32_t n_points, uint16_t *__restrict__ lut)
{
    __shared__ uint32_t part[1024];
    const int per = (nbins + 1023) / 1024, a = (int)threadIdx.x * per, b = min(a + per, nbins);
    uint32_t s = 0;
    for (int k = a; k < b; ++k) s += hist[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { // Hillis-Steele, inclusive
        const uint32_t v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = part[threadIdx.x] - s;
    const uint64_t top = (uint64_t)(sf::ORD_KEY_NONE - 1u), n = n_points > 0u ? n_points : 1u;
    for (int k = a; k < b; ++k) {
        const uint64_t key = run * top / n;
        lut[k] = (uint16_t)(key < top ? key : top - 1u);
        run += hist[k];
    }
}

__global__ __launch_bounds__(sf::ORD_BLK) void k_order_scatter(sf::OrderSrc s, const uint16_t *__restrict__ keys, const uint32_t *__restrict__ starts, uint32_t *__restrict__ out)
{
    sf::order_scatter_body(s, keys, starts, out);
}

// the ordered ids read coalesced, the queries' float4 records gathered (four in flight per lane), the cell-ordered
// SoA arrays written coalesced
constexpr int GATHER_PER_LANE = 4;
constexpr int GATHER_TILE = 256 * GATHER_PER_LANE;
// seg_n > 0: uniform segments of seg_n queries, placed like the ordering kernels (sf::order_block: a scan's tiles on one
// XCD, whose L2 then holds the scan's 16-byte records while they are gathered); seg_n = 0: one flat range (ragged segments)
__global__ __launch_bounds__(256) void k_order_gather(const float4 *__restrict__ rec, const uint32_t *__restrict__ idx, int64_t total, int seg_n, int seg_tiles, int nseg,
                                                      float *__restrict__ Xx, float *__restrict__ Xy, float *__restrict__ Xz)
{
    int64_t base, end = total;
    if (seg_n > 0) {
        int b, tile;
        if (!sf::order_block(seg_tiles, nseg, &b, &tile)) return;
        base = (int64_t)b * seg_n + (int64_t)tile * GATHER_TILE + threadIdx.x;
        end = (int64_t)(b + 1) * seg_n;
    } else {
        base = (int64_t)blockIdx.x * GATHER_TILE + threadIdx.x;
    }
    uint32_t o[GATHER_PER_LANE];
    float4 v[GATHER_PER_LANE];
#pragma unroll
    for (int k = 0; k < GATHER_PER_LANE; ++k) o[k] = base + 256 * k < end ? idx[base + 256 * k] : 0u;
#pragma unroll
    for (int k = 0; k < GATHER_PER_LANE; ++k) v[k] = rec[o[k]];
#pragma unroll
    for (int k = 0; k < GATHER_PER_LANE; ++k)
        if (base + 256 * k < end) {
            Xx[base + 256 * k] = v[k].x;
            Xy[base + 256 * k] = v[k].y;
            Xz[base + 256 * k] = v[k].z;
        }
}

// ------------------------------------------------------------------ tile search (sf_tile.hpp): queries sorted by map tile, searched out of LDS
// 20-bit tile key of global query o under the scan's current pose (sorted in two 10-bit digits: sf_order.hpp is a stable
// counting pass, least significant digit first); non-finite queries carry the largest key and end up last in their scan
constexpr uint32_t TILE_KEY_NONE = (1u << 20) - 1u;
struct TileKeyFn {
    SfGrid g;
    sf::SfTiles tl;
    const float *x, *y, *z;
    const IcpState *st;
    int n, digit;
    struct Point { float x, y, z; };
    struct Pose { float T[12]; };
    __device__ __forceinline__ Pose prepare(int b) const
    {
        Pose P;
#pragma unroll
        for (int k = 0; k < 12; ++k) P.T[k] = (float)st[b].T[k];
        return P;
    }
    __device__ __forceinline__ Point load(uint32_t o) const { return Point{x[o], y[o], z[o]}; }
    // (a locality key like CellKeyFn's: float32 arithmetic decides which of two tiles a query on their border is binned
    // to -- the search itself takes the query's exact cell and only asks whether the staged region covers it)
    __device__ __forceinline__ uint32_t full_key(const Pose &P, const Point &p) const
    {
        const float qx = fmaf(P.T[0], p.x, fmaf(P.T[1], p.y, fmaf(P.T[2], p.z, P.T[3])));
        const float qy = fmaf(P.T[4], p.x, fmaf(P.T[5], p.y, fmaf(P.T[6], p.z, P.T[7])));
        const float qz = fmaf(P.T[8], p.x, fmaf(P.T[9], p.y, fmaf(P.T[10], p.z, P.T[11])));
        if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return TILE_KEY_NONE;
        const int cx = (int)fminf(fmaxf(floorf((qx - g.org[0]) * g.inv_h), 0.0f), (float)(g.dim[0] - 1));
        const int cy = (int)fminf(fmaxf(floorf((qy - g.org[1]) * g.inv_h), 0.0f), (float)(g.dim[1] - 1));
        const int cz = (int)fminf(fmaxf(floorf((qz - g.org[2]) * g.inv_h), 0.0f), (float)(g.dim[2] - 1));
        return sf::tile_of_cell(tl, cx, cy, cz);
    }
    __device__ __forceinline__ uint32_t key(const Pose &P, const Point &p) const { return (full_key(P, p) >> (sf::ORD_KEY_BITS * digit)) & (uint32_t)(sf::ORD_BINS - 1); }
};

__global__ __launch_bounds__(sf::ORD_BLK) void k_order_hist_tile(sf::OrderSrc s, TileKeyFn kf, uint16_t *__restrict__ keys, uint32_t *__restrict__ counts)
{
    sf::order_hist_body(s, kf, keys, counts);
}

// tile key of every query of the ORDERED arrays (kf.x / y / z = the ordered copy), scan by scan
__global__ __launch_bounds__(256) void k_tile_keys(TileKeyFn kf, int64_t total, uint32_t *__restrict__ tkey)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int b = (int)(o / kf.n);
    tkey[o] = kf.full_key(kf.prepare(b), kf.load((uint32_t)o));
}

// seg[t * batch + b] = first position in scan b's ordered queries whose tile key is >= t, for t in [0, ntiles] (tile-major:
// the workgroup of tile t reads its `batch` starts and ends as two contiguous rows)
__global__ __launch_bounds__(256) void k_tile_starts(const uint32_t *__restrict__ tkey, int n, int batch, int ntiles, uint32_t *__restrict__ seg)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)(ntiles + 1) * batch) return;
    const uint32_t t = (uint32_t)(i / batch);
    const int b = (int)(i % batch);
    const uint32_t *k = tkey + (size_t)b * n;
    int lo = 0, hi = n; // first position with key >= t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (k[mid] < t) lo = mid + 1; else hi = mid;
    }
    seg[i] = (uint32_t)lo;
}

// statistics of the tile launches (integers: order independent): [0] queries that searched, [1] of them settled out of LDS,
// [2] walked the global index because the staged region did not cover them, [3] went on to ring 2 and beyond after the LDS
// search, [4] tiles whose region did not fit
constexpr int TILE_STATS = 8;
#ifdef SF_PHASE_TRACE
__device__ unsigned long long g_tile_trace[sf::PH_SHARDS * sf::PH_SLOTS];
#define TT_MARK(i) do { if (wv == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tt_acc[i] += (unsigned)(n_ - tt_t); tt_t = n_; } } while (0)
#else
#define TT_MARK(i) do { } while (0)
#endif

// One workgroup per map tile: stage the tile, then the tile's queries of every scan in flight (certificate, search out of LDS,
// cache entry).  No sums are formed here -- k_red_cached adds the pairs up from the cache in k_nn_red's order.
template <int MODE>
__global__ __launch_bounds__(sf::TILE_BLK) void k_tile_search(SfGrid g, sf::SfTiles tl, const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z,
                                                              int n, int batch, const IcpState *__restrict__ st, float thr, const uint32_t *__restrict__ seg,
                                                              float4 *__restrict__ qcache, int64_t cache_n, int reuse, unsigned long long *__restrict__ stats)
{
    // XCD-aware placement (as k_nn_red): linear id L runs on XCD L % 8; each XCD takes a contiguous eighth of the tiles,
    // so neighbouring tiles -- which share their halos -- meet in one L2
    const int per = ((int)gridDim.x) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (tile >= tl.ntiles) return;
    __shared__ sf::TileLds lds;
    __shared__ double Tm[sf::TILE_SCANS][12];
    __shared__ float mot[sf::TILE_SCANS];
    __shared__ int live[sf::TILE_SCANS];
    __shared__ uint32_t pre[sf::TILE_SCANS + 1], gofs[sf::TILE_SCANS];
    __shared__ unsigned int cnt[4];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const sf::TileRegion R = sf::tile_region(g, tl, (uint32_t)tile);
#ifdef SF_PHASE_TRACE
    unsigned tt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tt_t = __builtin_amdgcn_s_memtime();
#endif
    bool staged = false, fits = false;
    if (tid < 4) cnt[tid] = 0;
    float *ecache = reinterpret_cast<float *>(qcache + (size_t)cache_n);
    SfWindow nowin;
    nowin.kind = 0;
    for (int b0 = 0; b0 < batch; b0 += sf::TILE_SCANS) {
        __syncthreads(); // the previous group's tables are no longer read
        if (wv == 0) {
            const int b = b0 + lane;
            uint32_t s0 = 0, len = 0;
            if (b < batch && !st[b].done) {
                s0 = seg[(size_t)tile * batch + b];
                len = seg[(size_t)(tile + 1) * batch + b] - s0;
            }
            uint32_t v = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)v, o);
                if (lane >= o) v += t;
            }
            pre[lane + 1] = v;
            if (lane == 0) pre[0] = 0;
            gofs[lane] = (uint32_t)((size_t)b * n + s0);
        }
        for (int i = tid; i < sf::TILE_SCANS * 12; i += sf::TILE_BLK) {
            const int b = b0 + i / 12;
            Tm[i / 12][i % 12] = b < batch ? st[b].T[i % 12] : 0.0;
        }
        if (tid < sf::TILE_SCANS) {
            const int b = b0 + tid;
            mot[tid] = b < batch ? (float)st[b].motion : 0.0f;
            live[tid] = (b < batch && reuse && st[b].cache_live != 0) ? 1 : 0;
        }
        __syncthreads();
        TT_MARK(0);
        const uint32_t total = pre[sf::TILE_SCANS];
        if (total == 0) continue; // (uniform)
        if (!staged) { // only tiles that hold queries are staged
            fits = sf::tile_stage(g, R, &lds);
            staged = true;
            if (!fits && tid == 0 && stats) atomicAdd(&stats[4], 1ull);
        }
        TT_MARK(1);
        for (uint32_t i = (uint32_t)tid; i < total; i += sf::TILE_BLK) {
            // the scan this query belongs to: the last prefix not above i
            int bl = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) bl += (pre[bl + step] <= i) ? step : 0;
            const size_t o = (size_t)gofs[bl] + (size_t)(i - pre[bl]);
            const bool cache_live = live[bl] != 0;
            float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f), c2 = c1;
            float e = 0.0f;
            if (cache_live) {
                c1 = qcache[o];
                if (MODE == 2) { c2 = qcache[(size_t)cache_n + o]; e = c2.w; }
                else e = ecache[o];
            }
            const double x0 = X0x[o], y0 = X0y[o], z0 = X0z[o];
            const double *T = Tm[bl];
            const double sx = T[0] * x0 + T[1] * y0 + T[2] * z0 + T[3];
            const double sy = T[4] * x0 + T[5] * y0 + T[6] * z0 + T[7];
            const double sz = T[8] * x0 + T[9] * y0 + T[10] * z0 + T[11];
            const float qx = (float)sx, qy = (float)sy, qz = (float)sz;
            const float m_now = mot[bl];
            sf::NNHit chit, seed;
            float4 tn;
            bool need = reuse_certificate(true, qx, qy, qz, thr, m_now, e, c1, c2, chit, tn, seed);
            if (!need) {
                if (!cache_live) { // first launch after a (re)start: a lane that does not search (non-finite query) leaves "no search behind it"
                    if (MODE == 2) qcache[(size_t)cache_n + o] = make_float4(0.f, 0.f, 0.f, 0.f);
                    else ecache[o] = 0.0f;
                }
                continue;
            }
            // what nn_search_wave does before it walks: a query farther from the map's box than the acceptance radius has no neighbour
            sf::NNHit hit;
            hit.d2 = sf::search_start(thr);
            hit.j = -1;
            hit.px = hit.py = hit.pz = 0.0f;
            hit.lb2 = 3.0e38f;
            bool walk = g.n > 0;
            {
                const float ox = fmaxf(fmaxf(g.org[0] - qx, qx - (g.org[0] + (float)g.dim[0] * g.h)), 0.0f);
                const float oy = fmaxf(fmaxf(g.org[1] - qy, qy - (g.org[1] + (float)g.dim[1] * g.h)), 0.0f);
                const float oz = fmaxf(fmaxf(g.org[2] - qz, qz - (g.org[2] + (float)g.dim[2] * g.h)), 0.0f);
                const float gap = fmaxf(sqrtf(ox * ox + oy * oy + oz * oz) * 0.9995f - 1.0e-3f, 0.0f);
                if (gap * gap > thr) { hit.lb2 = gap * gap; walk = false; }
            }
            if (walk) {
                const sf::QueryGeo G = sf::query_geo(g, qx, qy, qz);
                int ring_from = 1; // 0: settled out of LDS
                if (fits && sf::tile_serves(g, R, G)) {
                    bool more;
                    const sf::TileHit th = sf::tile_search(g, R, &lds, G, qx, qy, qz, thr, seed.d2, seed.j, &more);
                    hit.d2 = th.d2; hit.j = th.j; hit.lb2 = th.lb2;
                    if (th.loc >= 0) { const float4 p = lds.pts[th.loc]; hit.px = p.x; hit.py = p.y; hit.pz = p.z; }
                    else if (th.j >= 0) { hit.px = seed.px; hit.py = seed.py; hit.pz = seed.pz; }
                    ring_from = more ? 2 : 0; // more: no map point within the boundary of the 27 cells (rare)
                    if (stats) atomicAdd(&cnt[more ? 3 : 1], 1u);
                } else { // the query has left what this tile staged (or the tile did not fit): the global index, lane by lane
                    if (seed.j >= 0 && seed.d2 < thr) { hit.d2 = seed.d2; hit.j = seed.j; hit.px = seed.px; hit.py = seed.py; hit.pz = seed.pz; }
                    if (stats) atomicAdd(&cnt[2], 1u);
                }
                if (ring_from) {
                    sf::nn_rings<false>(g, nowin, qx, qy, qz, ring_from, hit);
                    hit.lb2 = 0.0f; // no bound kept for the per-lane rings
                }
            }
            if (stats) atomicAdd(&cnt[0], 1u);
            float4 nrm = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 2 && hit.j >= 0) nrm = g.nrm[hit.j];
            const float en = fmaxf(sqrtf(hit.lb2) * 0.9999f + m_now * 0.999998f - 1.0e-6f, 1.0e-30f);
            qcache[o] = make_float4(hit.px, hit.py, hit.pz, __int_as_float(hit.j));
            if (MODE == 2) qcache[(size_t)cache_n + o] = make_float4(nrm.x, nrm.y, nrm.z, en);
            else ecache[o] = en;
        }
    }
    TT_MARK(2);
    __syncthreads();
    TT_MARK(3);
    if (stats && tid < 4 && cnt[tid]) atomicAdd(&stats[tid], (unsigned long long)cnt[tid]);
#ifdef SF_PHASE_TRACE
    if (tid == 0) {
        tt_acc[7] = 1;
        for (int i = 0; i < 8; ++i) atomicAdd(&g_tile_trace[(size_t)(blockIdx.x & (sf::PH_SHARDS - 1)) * sf::PH_SLOTS + i], (unsigned long long)tt_acc[i]);
    }
#endif
}

// The records of one launch from the neighbour cache alone (every entry is a pair that holds at the current pose: k_tile_search
// has just certified or searched it): rows, summation order and arithmetic of k_nn_red<MODE, false, false, Q> -- the sums are
// bit-identical to what k_nn_red forms from the same pairs.
template <int MODE, int Q>
__global__ __launch_bounds__(BLK) void k_red_cached(const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z, int n,
                                                    const IcpState *__restrict__ st, float thr, double *__restrict__ partials, int nblocks, const float4 *__restrict__ qcache,
                                                    int64_t cache_n)
{
    constexpr int NREC = MODE == 2 ? NREC_PLANE : NREC_P2P;
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int kk = L >> 3;
    const int b = kk % (int)gridDim.y;
    const int bx = (L & 7) * ((int)gridDim.x >> 3) + kk / (int)gridDim.y;
    if (bx >= nblocks) return;
    const IcpState *S = st + b;
    if (S->done) return;
    __shared__ double stage[BLK / 64][32];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    LanePair P[Q];
#pragma unroll
    for (int u = 0; u < Q; ++u) {
        const int slot = bx * (BLK * Q) + u * BLK + (int)threadIdx.x;
        const QueryIn q = query_in<MODE, false>(X0x, X0y, X0z, n, b, S, 0.0f, 0.0f, nullptr, qcache, cache_n, true, slot, n);
        sf::NNHit hit;
        hit.d2 = thr;
        hit.j = -1;
        hit.px = hit.py = hit.pz = 0.0f;
        hit.lb2 = 0.0f;
        float4 tn = make_float4(0.f, 0.f, 0.f, 0.f);
        const int32_t jc = __float_as_int(q.c1.w);
        if (slot < n && jc >= 0 && isfinite(q.qx) && isfinite(q.qy) && isfinite(q.qz)) {
            const float d2n = sf::l2_simple(q.qx, q.qy, q.qz, q.c1.x, q.c1.y, q.c1.z);
            if (d2n < thr) { hit.d2 = d2n; hit.j = jc; hit.px = q.c1.x; hit.py = q.c1.y; hit.pz = q.c1.z; tn = q.c2; }
        }
        P[u] = make_pair(q, hit, tn);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.0;
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const PairTerms t = pair_terms<MODE>(P[u]);
            add_half<MODE>(t, h, v);
        }
        if (MODE == 1 && h == 1) {
            const double t1 = wave_reduce_1(v[0]);
            if (lane == 0) stage[wv][16] = t1;
        } else {
            const double t0 = wave_reduce_16(v);
            if ((lane & 3) == 0) stage[wv][16 * h + (lane >> 2)] = t0;
        }
    }
    __syncthreads();
    if (threadIdx.x < NREC) {
        const int c = threadIdx.x;
        double *dst = partials + ((size_t)b * nblocks + bx) * REC_STRIDE;
        dst[c] = ((stage[0][c] + stage[1][c]) + stage[2][c]) + stage[3][c];
    }
}

// ------------------------------------------------------------------ sharded path: owned queries
// A rank owns the queries whose TRANSFORMED x lies in its slab [xlo, xhi).  Evaluating that per
// lane over the whole batch would leave 1/N of the lanes of every wave busy, so at the start of an
// alignment each rank compacts, per scan, the queries within the slab widened by a margin (1 m)
// (order-preserving: per-workgroup counts by ballot + popcount, one-workgroup scan per scan,
// scatter with the mbcnt lane rank), orders them by map cell like the unsharded path and gathers
// them into compact arrays; k_nn_red walks only those.  The arrays stay valid while no point of
// the scan has moved by that margin since: that is checked on the device after every solve
// (identically on every rank, every rank holds the same T); a scan that moves further stops with
// SF_ICP_FLAG_SHARD_STALE and the host rebuilds and resumes it (sf_icp_step_begin(first = 2)).

struct ScanBox { float lo[3], hi[3]; };

// bounding box of the finite points of every scan of the batch (SoA, scan b at [b * n, b * n + n)): a grid of (BOX_PARTS, batch)
// workgroups leaves partial boxes, one more launch folds them.  A scan without a finite point gets the empty box at the origin.
constexpr int BOX_PARTS = 32;
__global__ __launch_bounds__(256) void k_scan_boxes_partial(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, int n, ScanBox *__restrict__ part)
{
    const int b = blockIdx.y, p = blockIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const size_t base = (size_t)b * n;
    for (int i = p * 256 + (int)threadIdx.x; i < n; i += BOX_PARTS * 256) {
        const float v[3] = {x[base + i], y[base + i], z[base + i]};
        if (isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2])) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], v[d]); hi[d] = fmaxf(hi[d], v[d]); }
        }
    }
    __shared__ float s[BLK / 64][6];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { s[threadIdx.x >> 6][d] = lo[d]; s[threadIdx.x >> 6][3 + d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        part[(size_t)b * BOX_PARTS + p].lo[d] = fminf(fminf(s[0][d], s[1][d]), fminf(s[2][d], s[3][d]));
        part[(size_t)b * BOX_PARTS + p].hi[d] = fmaxf(fmaxf(s[0][3 + d], s[1][3 + d]), fmaxf(s[2][3 + d], s[3][3 + d]));
    }
}

__global__ void k_scan_boxes_final(const ScanBox *__restrict__ part, int batch, ScanBox *__restrict__ boxes)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    ScanBox r;
    for (int d = 0; d < 3; ++d) { r.lo[d] = INFINITY; r.hi[d] = -INFINITY; }
    for (int p = 0; p < BOX_PARTS; ++p)
        for (int d = 0; d < 3; ++d) { r.lo[d] = fminf(r.lo[d], part[(size_t)b * BOX_PARTS + p].lo[d]); r.hi[d] = fmaxf(r.hi[d], part[(size_t)b * BOX_PARTS + p].hi[d]); }
    if (!(r.lo[0] <= r.hi[0])) { for (int d = 0; d < 3; ++d) r.lo[d] = r.hi[d] = 0.0f; }
    boxes[b] = r;
}

__device__ __forceinline__ unsigned own_lane_rank(unsigned long long ballot)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0u));
}

__device__ __forceinline__ bool own_candidate(const IcpState *S, float x0, float y0, float z0, float xlo, float xhi, float margin)
{
    const float qx = (float)(S->T[0] * (double)x0 + S->T[1] * (double)y0 + S->T[2] * (double)z0 + S->T[3]);
    return qx >= xlo - margin && qx < xhi + margin;
}

// resume: scans that stopped because their arrays went stale run again
__global__ void k_own_resume(IcpState *__restrict__ st, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    if (st[b].flags & SF_ICP_FLAG_SHARD_STALE) {
        st[b].flags &= ~SF_ICP_FLAG_SHARD_STALE;
        st[b].done = 0;
    }
}

__global__ __launch_bounds__(BLK) void k_own_count(const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z, int n,
                                                   const IcpState *__restrict__ st, float xlo, float xhi, float margin, uint32_t *__restrict__ blk_counts, int nblocks,
                                                   unsigned long long *__restrict__ wave_keep)
{
    const int b = blockIdx.y;
    const IcpState *S = st + b;
    if (S->done) return;
    const int i = blockIdx.x * BLK + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const size_t o = (size_t)b * n + i;
        keep = own_candidate(S, X0x[o], X0y[o], X0z[o], xlo, xhi, margin);
    }
    __shared__ uint32_t wcnt[BLK / 64];
    const unsigned long long bal = __ballot(keep);
    if ((threadIdx.x & 63) == 0) {
        wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
        // the screening result of this wave, kept for the scatter pass: every rank screens ALL points of every scan in
        // flight (the O(world) part of a rank's work), the second pass then reads 1 bit per point instead of the point
        wave_keep[((size_t)b * nblocks + blockIdx.x) * (BLK / 64) + (threadIdx.x >> 6)] = bal;
    }
    __syncthreads();
    if (threadIdx.x == 0) blk_counts[(size_t)b * nblocks + blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// one workgroup per scan: exclusive scan of its block counts in place, total -> own_count[b]
__global__ __launch_bounds__(1024) void k_own_scan(const IcpState *__restrict__ st, uint32_t *__restrict__ blk_counts, int nblocks, uint32_t *__restrict__ own_count)
{
    const int b = blockIdx.x;
    if (st[b].done) {
        if (threadIdx.x == 0) own_count[b] = 0;
        return;
    }
    uint32_t *v = blk_counts + (size_t)b * nblocks;
    __shared__ uint32_t s[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t x = i < nblocks ? v[i] : 0u;
        s[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t t = threadIdx.x >= (unsigned)off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t incl = s[threadIdx.x], c0 = carry;
        if (i < nblocks) v[i] = c0 + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c0 + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) own_count[b] = carry;
}

// own_idx[own_off[b] + rank within the scan] = global query index b * n + i
__global__ __launch_bounds__(BLK) void k_own_scatter(int n, const IcpState *__restrict__ st, const unsigned long long *__restrict__ wave_keep, const uint32_t *__restrict__ blk_off,
                                                     int nblocks, const uint32_t *__restrict__ own_off, uint32_t *__restrict__ own_idx)
{
    const int b = blockIdx.y;
    if (st[b].done) return;
    const int i = blockIdx.x * BLK + threadIdx.x;
    const int wv = threadIdx.x >> 6;
    const unsigned long long *wk = wave_keep + ((size_t)b * nblocks + blockIdx.x) * (BLK / 64); // k_own_count's ballots of this workgroup
    const unsigned long long bal = wk[wv];
    if (!((bal >> (threadIdx.x & 63)) & 1ull)) return;
    uint32_t off = own_off[b] + blk_off[(size_t)b * nblocks + blockIdx.x];
    for (int k = 0; k < wv; ++k) off += (uint32_t)__popcll(wk[k]);
    own_idx[off + own_lane_rank(bal)] = (uint32_t)((size_t)b * n + i);
}

// the pose the owned arrays were built at
__global__ void k_own_mark(IcpState *__restrict__ st, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch || st[b].done) return;
    for (int i = 0; i < 12; ++i) st[b].T_list[i] = st[b].T[i];
    st[b].cache_live = 0; // the compact indices of this scan have changed: its cache entries mean nothing until rewritten
}

// after a pose update: has any point of the scan's bounding box come close to the margin away from
// where it was when the owned arrays were built?  (an affine map of a box moves its corners the most)
__device__ __forceinline__ void own_check_motion(IcpState *S, const ScanBox &box, float margin)
{
    double worst = 0.0;
    for (int c = 0; c < 8; ++c) {
        const double x = (c & 1) ? box.hi[0] : box.lo[0], y = (c & 2) ? box.hi[1] : box.lo[1], z = (c & 4) ? box.hi[2] : box.lo[2];
        double d2 = 0.0;
        for (int r = 0; r < 3; ++r) {
            const double a = S->T[4 * r] * x + S->T[4 * r + 1] * y + S->T[4 * r + 2] * z + S->T[4 * r + 3];
            const double o = S->T_list[4 * r] * x + S->T_list[4 * r + 1] * y + S->T_list[4 * r + 2] * z + S->T_list[4 * r + 3];
            d2 += (a - o) * (a - o);
        }
        worst = d2 > worst ? d2 : worst;
    }
    if (!(worst < 0.81 * (double)margin * (double)margin)) {
        S->flags |= SF_ICP_FLAG_SHARD_STALE;
        S->done = 1;
    }
}

// after a pose update of the launch list: IcpState::motion grows by the largest displacement the update gave any point of
// the source batch's bounding box (rounded up), and the scan's cache entries count as written (reuse_certificate)
__device__ __forceinline__ void track_motion(IcpState *S, const double *To, const ScanBox &box)
{
    double worst = 0.0;
    for (int c = 0; c < 8; ++c) {
        const double x = (c & 1) ? box.hi[0] : box.lo[0], y = (c & 2) ? box.hi[1] : box.lo[1], z = (c & 4) ? box.hi[2] : box.lo[2];
        double d2 = 0.0;
        for (int r = 0; r < 3; ++r) {
            const double d = (S->T[4 * r] - To[4 * r]) * x + (S->T[4 * r + 1] - To[4 * r + 1]) * y + (S->T[4 * r + 2] - To[4 * r + 2]) * z + (S->T[4 * r + 3] - To[4 * r + 3]);
            d2 += d * d;
        }
        worst = d2 > worst ? d2 : worst;
    }
    const double step = sqrt(worst) * 1.000001 + 1.0e-9;
    S->motion = (step == step) ? S->motion + step : 1.0e30; // a non-finite pose: nothing certifies any more
    S->cache_live = 1;
}

// ------------------------------------------------------------------ solves (thread 0 of the scan's workgroup)
// Open3D RegistrationICP loop body after a correspondence search (k-th search, K = max_iteration)
__device__ __forceinline__ void solve_o3d(IcpState *S, const double *rec, int n_src, int k, int K)
{
    k = S->n_research; // == the host's loop index on the unsharded path; the state's own count survives a resume
    const double n = rec[0];
    const double fitness = n_src > 0 ? n / (double)n_src : 0.0;
    const double rmse = n > 0 ? sqrt(rec[16] / n) : 0.0;
    S->fitness = fitness;
    S->rmse = rmse;
    S->n_corr = (int)n;
    S->n_research += 1;
    if (k > 0 && fabs(S->prev_fitness - fitness) < 1e-6 && fabs(S->prev_rmse - rmse) < 1e-6) {
        S->converged = 1;
        S->done = 1;
        return;
    }
    if (k >= K) { S->done = 1; return; }
    if (n > 0) {
        double upd[16], Tc[16];
        kabsch_from_record(rec, upd);
#pragma unroll
        for (int i = 0; i < 16; ++i) Tc[i] = S->T[i];
        mat4_mul(upd, Tc, Tc);
#pragma unroll
        for (int i = 0; i < 16; ++i) S->T[i] = Tc[i];
    }
    S->iterations += 1;
    S->prev_fitness = fitness;
    S->prev_rmse = rmse;
}

// upd_out (optional, 12 doubles): the pose update this solve applied, T <- upd T (the identity when it applied none)
__device__ __forceinline__ void solve_plane(IcpState *S, const double *rec, int n_src, int K, double *upd_out = nullptr)
{
    if (upd_out)
        for (int i = 0; i < 12; ++i) upd_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
    const double n = rec[0];
    S->fitness = n_src > 0 ? n / (double)n_src : 0.0;
    S->rmse = n > 0 ? sqrt(rec[29] / n) : 0.0;
    S->n_corr = (int)n;
    S->n_research += 1;
    double A[36], rhs[6], x[6];
    {
        int k = 2;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = a; c < 6; ++c) { A[6 * a + c] = rec[k]; A[6 * c + a] = rec[k]; ++k; }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) rhs[a] = -rec[23 + a];
    if (n < 6 || ldlt6(A, rhs, x) != 0) {
        S->flags |= SF_ICP_FLAG_SINGULAR;
        S->done = 1;
        return;
    }
    double upd[16], Tc[16];
    vec6_to_mat4(x, upd);
    if (upd_out)
        for (int i = 0; i < 12; ++i) upd_out[i] = upd[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) Tc[i] = S->T[i];
    mat4_mul(upd, Tc, Tc);
#pragma unroll
    for (int i = 0; i < 16; ++i) S->T[i] = Tc[i];
    S->iterations += 1;
    if (S->iterations >= K) { S->converged = 1; S->done = 1; }
}

template <int MODE>
__global__ __launch_bounds__(RBLK) void k_reduce_solve(IcpState *__restrict__ st, const double *__restrict__ partials, int nblocks, int n_src, int k, int K,
                                                       const ScanBox *__restrict__ boxp)
{
    constexpr int NREC = MODE == 2 ? NREC_PLANE : NREC_P2P;
    const int b = blockIdx.x;
    IcpState *S = st + b;
    if (S->done) return;
    __shared__ double rec[REC_STRIDE];
    reduce_partials<NREC>(partials + (size_t)b * nblocks * REC_STRIDE, nblocks, rec);
    if (threadIdx.x == 0) {
        for (int c = 0; c < NREC; ++c) S->rec[c] = rec[c];
        double To[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) To[i] = S->T[i];
        if (MODE == 1) solve_o3d(S, rec, n_src, k, K);
        else solve_plane(S, rec, n_src, K);
        track_motion(S, To, boxp[b]);
    }
}

// ------------------------------------------------------------------ frozen pairs (P2PLANE, wide scans, neighbour reuse on)
// Once the certificate holds for (nearly) every query, the launches of an alignment re-stream 44 bytes per query only
// to find that no pair changed -- yet with the pairs fixed the normal equations are a POLYNOMIAL in the pose:
// with y = R x + t,  r = n.(y - p) = n.y - c  (c = n.p),  J = [y x n; n], every entry of sum J J^T, sum J r, sum r^2
// and sum |y - p|^2 is a combination of the 96 sums (moments) below with coefficients of degree <= 2 in (R, t).
// So one launch (the "freeze" launch) forms the moments of all pairs that are certain to stay as they are, and the
// iterations after it evaluate the sums from the moments in O(1) -- no pass over the scan -- and go on solving and
// updating the pose exactly as before (float64 throughout: the per-pair terms are float64 products of the same
// quantities, so the two forms differ by summation rounding only, ~1e-13 of a pose; tested against the launch-by-launch
// evaluation and against the oracle at its usual 1e-9).
// "Certain to stay": at the freeze launch a query is FROZEN if its certificate (reuse_certificate) would still hold
// after the scan has moved by `guard` more than it has so far AND its accepted / rejected status (d2 < thr) cannot
// change within that motion; everything else is ACTIVE: listed per slab row, taken out of the moments and evaluated
// launch by launch as before (certificate, search if it fails) by one wave per row.  After every pose update the scan's
// motion since the freeze (IcpState::motion, the bound the certificate itself uses) is compared with the guard: beyond
// it the scan thaws -- the next launch treats every query the ordinary way (the cache entries were never touched) --
// and may freeze again.  The guard is a multiple of the last update's motion: ICP steps shrink geometrically.
// The pairs and therefore the result do not depend on the guard, only how many queries are active does.
// Moments are taken in the coordinates of the freeze pose, x = T_f x0 (float64), so (R, t) = T_now T_f^-1 stays near
// the identity.  Layout (k6(a,b): 00 01 02 11 12 22):
//   [0,36)  n_c n_f x_d x_e   (6 k6(c,f) + k6(d,e))      [36,54) n_c n_f x_d (36 + 3 k6 + d)     [54,60) n_c n_f
//   [60,69) c n_a x_d (60 + 3 a + d)   [69,72) c n_a   72 c^2   73 count
//   [74,80) x_d x_e   [80,83) x_d   [83,92) p_a x_d (83 + 3 a + d)   [92,95) p_a   95 |p|^2
constexpr int FZ_NMOM = 96;
constexpr int FZ_CAP = 128;  // active queries one slab row can list (of its 256 * Q)
constexpr int64_t FREEZE_AUTO_MIN_QUERIES = 700000; // sf_icp_set_freeze(1): batches from this many queries on (measured, 200 k-point scans, two lanes: 2 in flight -3 %, 4: +6 %, 8: +1 %; on one lane the break-even was 0.8 M)
struct FreezeState {
    int mode;         // 0: every query evaluated launch by launch, 1: the next launch is a freeze launch, 2: frozen
    int tries;        // freeze launches that did not hold (a row's list overflowed) + thaws
    int froze, thawed;
    int64_t n_active; // active queries of the last freeze launch
    float guard;      // motion allowed after the freeze [m]
    int launch_mode;  // `mode` as the launch in flight found it (the solve's bookkeeping needs it when reduce and solve are two kernels)
    double motion0;   // IcpState::motion the freeze launch classified with
    double Tf[12];    // the pose of the freeze launch
    double D[12];     // the pose relative to it: the product of the updates solved since (exact for ANY initial pose -- a float32 or
                      // blended, not quite rigid prior (localization_node.cpp:329) included; T_now Tf^-1 through a rigid inverse was not)
    double mom[FZ_NMOM];
};
struct FreezeParams { float guard_scale, guard_min, guard_max; int max_tries; };

__device__ __forceinline__ constexpr int k6a(int k) { return k < 3 ? 0 : (k < 5 ? 1 : 2); }
__device__ __forceinline__ constexpr int k6b(int k) { return k < 3 ? k : (k < 5 ? k - 2 : 2); }
__device__ __forceinline__ constexpr int k6(int a, int b) { return a <= b ? (a == 0 ? b : (a == 1 ? 2 + b : 5)) : (b == 0 ? a : (b == 1 ? 2 + a : 5)); }

struct MomIn { double x[3], n[3], p[3], c, w; };
__device__ __forceinline__ MomIn mom_in(const LanePair &P, bool take)
{
    MomIn t;
    const bool ok = P.ok && take;
    t.w = ok ? 1.0 : 0.0;
    t.x[0] = ok ? P.sx : 0.0; t.x[1] = ok ? P.sy : 0.0; t.x[2] = ok ? P.sz : 0.0;
    t.p[0] = (double)(ok ? P.px : 0.0f); t.p[1] = (double)(ok ? P.py : 0.0f); t.p[2] = (double)(ok ? P.pz : 0.0f);
    t.n[0] = (double)(ok ? P.tn.x : 0.0f); t.n[1] = (double)(ok ? P.tn.y : 0.0f); t.n[2] = (double)(ok ? P.tn.z : 0.0f);
    t.c = t.n[0] * t.p[0] + t.n[1] * t.p[1] + t.n[2] * t.p[2];
    return t;
}
// acc + moment I of the pair, the last product fused into the addition
template <int I>
__device__ __forceinline__ double mom_fma(const MomIn &t, double acc)
{
    if constexpr (I < 36) return fma(t.n[k6a(I / 6)] * t.n[k6b(I / 6)], t.x[k6a(I % 6)] * t.x[k6b(I % 6)], acc);
    else if constexpr (I < 54) return fma(t.n[k6a((I - 36) / 3)] * t.n[k6b((I - 36) / 3)], t.x[(I - 36) % 3], acc);
    else if constexpr (I < 60) return fma(t.n[k6a(I - 54)], t.n[k6b(I - 54)], acc);
    else if constexpr (I < 69) return fma(t.c * t.n[(I - 60) / 3], t.x[(I - 60) % 3], acc);
    else if constexpr (I < 72) return fma(t.c, t.n[I - 69], acc);
    else if constexpr (I == 72) return fma(t.c, t.c, acc);
    else if constexpr (I == 73) return acc + t.w;
    else if constexpr (I < 80) return fma(t.x[k6a(I - 74)], t.x[k6b(I - 74)], acc);
    else if constexpr (I < 83) return acc + t.x[I - 80];
    else if constexpr (I < 92) return fma(t.p[(I - 83) / 3], t.x[(I - 83) % 3], acc);
    else if constexpr (I < 95) return acc + t.p[I - 92];
    else return fma(t.p[0], t.p[0], fma(t.p[1], t.p[1], fma(t.p[2], t.p[2], acc)));
}
template <int H, int K = 0>
__device__ __forceinline__ void mom_add16(const MomIn &t, double (&v)[16])
{
    if constexpr (K < 16) {
        v[K] = mom_fma<16 * H + K>(t, v[K]);
        mom_add16<H, K + 1>(t, v);
    }
}

// whether the query is certain to keep its pair (and its accepted / rejected status) while the scan moves by at most
// `guard` more: the certificate of reuse_certificate evaluated at motion m_now + guard with the distance grown by guard
// (sharded: `in_range` = the slot holds a candidate of this rank; whether the rank OWNS it -- x inside its slab -- must not
// change within the guard either)
template <bool SHARD>
__device__ __forceinline__ bool stays_frozen(const QueryIn &q, bool in_range, float thr, float m_now, float guard, float xlo, float xhi)
{
    if (!in_range) return true; // no query here: nothing to evaluate, ever
    if (!(isfinite(q.qx) && isfinite(q.qy) && isfinite(q.qz))) return true; // never has a pair (the search takes no non-finite query; no slab holds it)
    if (SHARD) {
        const float gx = guard * 1.000002f + (fabsf(q.qx) + 1.0f) * 2.6e-7f + 1.0e-6f; // the motion + the float32 roundings of x then and now
        if (q.valid) {
            if (!(q.qx - gx >= xlo && q.qx + gx < xhi)) return false; // may leave the slab
        } else {
            return q.qx + gx < xlo || q.qx - gx >= xhi; // stays some other rank's (else: may enter the slab)
        }
    }
    if (!(q.e > 0.0f)) return false;
    const float m = m_now + guard * 1.000002f;
    const float reach = q.e - m * 1.000002f - (fabsf(q.qx) + fabsf(q.qy) + fabsf(q.qz) + 3.0f * guard + 1.0f) * 2.6e-7f - (q.e + m) * 5.0e-7f - 1.0e-6f;
    const float rt = sqrtf(thr);
    const int32_t jc = __float_as_int(q.c1.w);
    if (jc < 0) return rt * 1.0001f + 1.0e-6f < reach;
    const float d = sqrtf(sf::l2_simple(q.qx, q.qy, q.qz, q.c1.x, q.c1.y, q.c1.z));
    if (!((d + guard) * 1.0001f + 1.0e-6f < reach)) return false;
    const bool stays_in = (d + guard) * 1.0001f + 1.0e-6f < rt * 0.9999f;
    const bool stays_out = (d - guard) * 0.9999f - 1.0e-6f > rt * 1.0001f;
    return stays_in || stays_out;
}

// k_nn_red for the launches that may freeze: P2PLANE, no window, unsharded, Q queries per lane.  Per scan (FreezeState::mode):
//   0  the ordinary launch (same pairs, same sums, same row as k_nn_red)
//   1  freeze launch: ordinary pairs; frozen ones into the moment row, active ones into the ordinary row and the row's list
//   2  frozen: the scan's active queries, FZ_CAP per workgroup (wave 0: certificate, search if it fails), into the first rows; the other workgroups leave
struct FzArgs { // what k_nn_red_fz and k_nn_red_fz_few pass on to the row
    const float *X0x, *X0y, *X0z;
    int n;
    const IcpState *st;
    float thr, xlo, xhi;
    double *partials;
    int nblocks;
    const uint32_t *own_off;
    float4 *qcache;
    int64_t cache_n;
    uint32_t *stats;
    const FreezeState *fz;
    double *mom_part;
    uint32_t *act_cnt;
    uint16_t *act_ids;
    const uint32_t *act_all;
};

// slab row bx of scan b (not done) in mode fmode; every `return` is taken by whole waves, and outside the frozen mode by the
// whole workgroup
template <int Q, bool SHARD>
__device__ __forceinline__ void nn_red_fz_row(const SfGrid &g, const SfWindow &w, const FzArgs &A, int b, int bx, int fmode)
{
    constexpr int MODE = 2;
    constexpr int NREC = NREC_PLANE;
    static_assert(FZ_CAP == 64 * Q, "wave 0 takes FZ_CAP active queries in Q rounds of 64");
    const float *__restrict__ X0x = A.X0x, *__restrict__ X0y = A.X0y, *__restrict__ X0z = A.X0z;
    const int n = A.n, nblocks = A.nblocks;
    const float thr = A.thr, xlo = A.xlo, xhi = A.xhi;
    double *__restrict__ partials = A.partials, *__restrict__ mom_part = A.mom_part;
    const uint32_t *__restrict__ own_off = A.own_off, *__restrict__ act_all = A.act_all;
    float4 *__restrict__ qcache = A.qcache;
    const int64_t cache_n = A.cache_n;
    uint32_t *__restrict__ stats = A.stats, *__restrict__ act_cnt = A.act_cnt;
    uint16_t *__restrict__ act_ids = A.act_ids;
    const FreezeState *__restrict__ fz = A.fz;
    const IcpState *S = A.st + b;
    // sharded: this rank's compact arrays of owned-query candidates, as in k_nn_red
    const int n_live = SHARD ? (int)(own_off[b + 1] - own_off[b]) : n;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t row = (size_t)b * nblocks + bx;
    double *dst = partials + row * REC_STRIDE;
    __shared__ sf::WaveNN nn_ws[BLK / 64];
    LanePair P[Q];
    if (fmode == 2) {
        // the scan's active queries, FZ_CAP per workgroup from the scan's list (the reduce kernel wrote it, in row and slot
        // order); workgroups beyond the list leave at once, the reduce kernel adds only the rows
        // that were written.  (Measured as a full grid, 31.3 us per launch: four waves side by side with 64 each 33.8, a
        // fixed grid of resident workgroups walking the rows with the per-scan state in LDS 32.1 -- and 13 us with this path
        // returning at once: the dispatch of 25 000 workgroups that find their scan frozen.  Hence k_nn_red_fz_few, 19 us.)
        // Waves 0 .. Q-1 take 64 of the piece's queries each, side by side (one round trip chain instead of Q in a row); the
        // whole workgroup meets at the barrier and the waves' records are added in wave order.
        const int64_t total = fz[b].n_active, first = (int64_t)bx * FZ_CAP;
        if (first >= total) return;
        __shared__ double fstage[Q][32];
        if (wv < Q) {
            const uint32_t cnt = (uint32_t)min<int64_t>(total - first, FZ_CAP);
            const uint32_t *list = act_all + (size_t)b * nblocks * FZ_CAP + (size_t)first;
            const uint32_t i = (uint32_t)(wv * 64 + lane);
            const int slot = i < cnt ? (int)list[i] : n_live; // n_live: no query
            const LanePair A1 = nn_pair<MODE, false, SHARD>(g, w, X0x, X0y, X0z, n, b, S, thr, xlo, xhi, own_off, qcache, cache_n, slot, n_live, &nn_ws[wv], stats);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = 0.0;
                const PairTerms t = pair_terms<MODE>(A1);
                add_half<MODE>(t, h, v);
                const double t0 = wave_reduce_16(v);
                if ((lane & 3) == 0) fstage[wv][16 * h + (lane >> 2)] = t0;
            }
        }
        __syncthreads();
        if (threadIdx.x < NREC) {
            double v = fstage[0][threadIdx.x];
#pragma unroll
            for (int u = 1; u < Q; ++u) v += fstage[u][threadIdx.x];
            dst[threadIdx.x] = v;
        }
        return;
    }
    if (SHARD && bx * (BLK * Q) >= n_live) return; // the reduce kernels read only the rows that exist
    __shared__ double stage[BLK / 64][32];
    __shared__ uint32_t act_w[BLK / 64][Q];
    bool active[Q];
#pragma unroll
    for (int u = 0; u < Q; ++u) active[u] = false;
    bool fast = false;
    const float m_now = (float)S->motion;
    const float guard = fz[b].guard;
    const bool attempted = qcache != nullptr && S->cache_live != 0;
    if (attempted) {
        bool any_need = false;
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const int slot = bx * (BLK * Q) + u * BLK + (int)threadIdx.x;
            const QueryIn q = query_in<MODE, SHARD>(X0x, X0y, X0z, n, b, S, xlo, xhi, own_off, qcache, cache_n, true, slot, n_live);
            sf::NNHit hit, seed;
            float4 tn;
            any_need = reuse_certificate(q.valid, q.qx, q.qy, q.qz, thr, m_now, q.e, q.c1, q.c2, hit, tn, seed) || any_need;
            P[u] = make_pair(q, hit, tn);
            if (fmode == 1) active[u] = !stays_frozen<SHARD>(q, slot < n_live, thr, m_now, guard, xlo, xhi);
        }
        fast = __ballot(any_need) == 0ull;
    }
    if (!fast) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const int slot = bx * (BLK * Q) + u * BLK + (int)threadIdx.x;
            P[u] = nn_pair<MODE, false, SHARD>(g, w, X0x, X0y, X0z, n, b, S, thr, xlo, xhi, own_off, qcache, cache_n, slot, n_live, &nn_ws[wv], stats);
            // (freeze launch: the classes stand as taken above -- a lane that searches here failed the plain certificate, so it
            // failed the guarded one and is active; the others' pairs are the ones the attempt found)
            if (fmode == 1 && !attempted) active[u] = slot < n_live;
        }
    }
    if (fmode == 1) {
        // the row's active list, in slot order (fixed: it is the summation order of the frozen launches)
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const unsigned long long bal = __ballot(active[u]);
            if (lane == 0) act_w[wv][u] = (uint32_t)__popcll(bal);
        }
        __syncthreads();
        uint32_t total = 0;
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            uint32_t base = 0;
#pragma unroll
            for (int ww = 0; ww < BLK / 64; ++ww) {
                if (ww < wv) base += act_w[ww][u];
                total += act_w[ww][u];
            }
            uint32_t before = 0;
#pragma unroll
            for (int uu = 0; uu < Q; ++uu)
                if (uu < u)
#pragma unroll
                    for (int ww = 0; ww < BLK / 64; ++ww) before += act_w[ww][uu];
            const unsigned long long bal = __ballot(active[u]);
            const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            const uint32_t pos = before + base + rank;
            if (active[u] && pos < (uint32_t)FZ_CAP) act_ids[row * FZ_CAP + pos] = (uint16_t)(u * BLK + (int)threadIdx.x);
        }
        if (threadIdx.x == 0) act_cnt[row] = total; // beyond FZ_CAP: the freeze does not hold (k_reduce_solve_fz)
    }
    // the ordinary record: every pair (mode 0) / the active pairs (freeze launch: three waves in four have none -- their
    // record is zero without forming and reducing it)
    bool wave_has_pairs = true;
    if (fmode == 1) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < Q; ++u) any = any || (P[u].ok && active[u]);
        wave_has_pairs = __ballot(any) != 0ull;
        if (!wave_has_pairs && lane < 32) stage[wv][lane] = 0.0;
    }
    if (wave_has_pairs)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.0;
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            LanePair A = P[u];
            if (fmode == 1) A.ok = A.ok && active[u];
            const PairTerms t = pair_terms<MODE>(A);
            add_half<MODE>(t, h, v);
        }
        const double t0 = wave_reduce_16(v);
        if ((lane & 3) == 0) stage[wv][16 * h + (lane >> 2)] = t0;
    }
    __syncthreads();
    if (threadIdx.x < NREC) {
        const int c = threadIdx.x;
        dst[c] = ((stage[0][c] + stage[1][c]) + stage[2][c]) + stage[3][c];
    }
    if (fmode != 1) return;
    // the moments of the frozen pairs, 16 at a time through the same wave reduction
    __shared__ double mstage[BLK / 64][FZ_NMOM];
#define SF_FZ_CHUNK(H)                                                     \
    {                                                                      \
        double v[16];                                                      \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) v[k] = 0.0;         \
        _Pragma("unroll") for (int u = 0; u < Q; ++u)                      \
        {                                                                  \
            const MomIn t = mom_in(P[u], !active[u]);                      \
            mom_add16<H>(t, v);                                            \
        }                                                                  \
        const double t0 = wave_reduce_16(v);                               \
        if ((lane & 3) == 0) mstage[wv][16 * H + (lane >> 2)] = t0;        \
    }
    SF_FZ_CHUNK(0) SF_FZ_CHUNK(1) SF_FZ_CHUNK(2) SF_FZ_CHUNK(3) SF_FZ_CHUNK(4) SF_FZ_CHUNK(5)
#undef SF_FZ_CHUNK
    __syncthreads();
    if (threadIdx.x < FZ_NMOM) {
        const int c = threadIdx.x;
        mom_part[row * FZ_NMOM + c] = ((mstage[0][c] + mstage[1][c]) + mstage[2][c]) + mstage[3][c];
    }
}

// one workgroup per slab row, placed as k_nn_red's: the launch in which scans freeze (and any launch that may have to treat
// every query the ordinary way at full speed)
template <int Q, bool SHARD>
__global__ __launch_bounds__(BLK, NN_RED_WAVES) void k_nn_red_fz(SfGrid g, SfWindow w, FzArgs A)
{
    const int L = blockIdx.y * gridDim.x + blockIdx.x; // placement as k_nn_red
    const int kk = L >> 3;
    const int b = kk % (int)gridDim.y;
    const int bx = (L & 7) * ((int)gridDim.x >> 3) + kk / (int)gridDim.y;
    if (bx >= A.nblocks) return;
    if (A.st[b].done) return;
    nn_red_fz_row<Q, SHARD>(g, w, A, b, bx, A.fz[b].mode);
}

// The launches AFTER the first chance to freeze: FZ_FEW workgroups per scan.  A frozen scan needs one workgroup per 128
// active queries (0.2 % of the queries here: 4 of its 391 rows) -- as a full grid the other 25 000 workgroups of a 64-scan
// launch cost 13 us just to be dispatched, find the scan frozen and leave (measured: 31 us per launch, 13 of them with the
// row function returning at once).  A scan that is NOT frozen (its freeze launch was voided, it thawed, or the last update
// was still too large to ask) is walked by its FZ_FEW workgroups row by row, stride FZ_FEW: correct, at roughly 3/4 of the
// full grid's speed.  grid (FZ_FEW, batch): workgroup r of every scan runs on XCD r % 8 and takes the rows = r (mod FZ_FEW).
constexpr int FZ_FEW = 16;
template <int Q, bool SHARD>
__global__ __launch_bounds__(BLK, NN_RED_WAVES) void k_nn_red_fz_few(SfGrid g, SfWindow w, FzArgs A)
{
    const int r = (int)blockIdx.x, b = (int)blockIdx.y;
    if (A.st[b].done) return;
    const int fmode = A.fz[b].mode;
    if (fmode == 2) {
        const int64_t total = A.fz[b].n_active;
        for (int bx = r; (int64_t)bx * FZ_CAP < total; bx += FZ_FEW) {
            nn_red_fz_row<Q, SHARD>(g, w, A, b, bx, 2);
            __syncthreads();
        }
        return;
    }
    for (int bx = r; bx < A.nblocks; bx += FZ_FEW) {
        nn_red_fz_row<Q, SHARD>(g, w, A, b, bx, fmode);
        __syncthreads(); // the row's LDS (search state, reduction stages, list counters) is free again
    }
}

// fixed-order column sums of a slab with rows of STRIDE doubles, NCOL columns (NCOL <= 128), by NT threads (a multiple of
// 128): thread (slice s of NT / 128, column c of 128) adds rows s, s + NT / 128, ...; the slices are then added in order
// (NS = 8 slices whatever NT: a workgroup of fewer than 1024 threads takes several slices per thread, one after the other --
// every (slice, column) sum is the same expression, so the result is bit-identical for every NT, as reduce_partials')
template <int STRIDE, int NCOL, int NT>
__device__ __forceinline__ void reduce_columns(const double *__restrict__ part, int nrows, double *out)
{
    constexpr int NS = 8;
    static_assert(NT % 128 == 0 && NT / 128 >= 1 && NT / 128 <= NS && NCOL <= 128, "slices of 128 columns");
    __shared__ double sl[NS][128];
    const int c = threadIdx.x & 127;
    for (int sidx = threadIdx.x >> 7; sidx < NS; sidx += NT / 128) {
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (c < NCOL) {
            int r = sidx;
            for (; r + 3 * NS < nrows; r += 4 * NS) {
                const double a0 = part[(size_t)r * STRIDE + c], a1 = part[(size_t)(r + NS) * STRIDE + c];
                const double a2 = part[(size_t)(r + 2 * NS) * STRIDE + c], a3 = part[(size_t)(r + 3 * NS) * STRIDE + c];
                v0 += a0; v1 += a1; v2 += a2; v3 += a3;
            }
            for (; r < nrows; r += NS) v0 += part[(size_t)r * STRIDE + c];
        }
        sl[sidx][c] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    if (threadIdx.x < NCOL) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) v += sl[k][threadIdx.x];
        out[threadIdx.x] = v;
    }
    __syncthreads();
}

// the P2PLANE record (30 sums) of the frozen pairs at the pose D = (R, t) relative to the freeze pose, from their moments;
// threads 0..63 of the workgroup work, the result is ADDED to rec[] by thread 0 (after a barrier)
__device__ __forceinline__ void frozen_record(const double *__restrict__ mom, const double *D, double *rec)
{
    __shared__ double YY[6][6]; // [k6(c,f)][k6(b,e)] = sum n_c n_f y_b y_e
    __shared__ double YN[6][3]; // sum n_c n_f y_b
    __shared__ double CY[3][3]; // sum c n_a y_b
    __shared__ double SD;       // sum |y - p|^2
    const int tid = threadIdx.x;
    auto R = [&](int r, int c) { return D[4 * r + c]; };
    auto T = [&](int r) { return D[4 * r + 3]; };
    if (tid < 36) {
        const int cf = tid / 6, be = tid % 6, bb = k6a(be), ee = k6b(be);
        double v = 0.0;
        for (int d = 0; d < 3; ++d)
            for (int e = 0; e < 3; ++e) v += R(bb, d) * R(ee, e) * mom[6 * cf + k6(d, e)];
        for (int d = 0; d < 3; ++d) v += (R(bb, d) * T(ee) + T(bb) * R(ee, d)) * mom[36 + 3 * cf + d];
        v += T(bb) * T(ee) * mom[54 + cf];
        YY[cf][be] = v;
    } else if (tid < 54) {
        const int cf = (tid - 36) / 3, bb = (tid - 36) % 3;
        double v = T(bb) * mom[54 + cf];
        for (int d = 0; d < 3; ++d) v += R(bb, d) * mom[36 + 3 * cf + d];
        YN[cf][bb] = v;
    } else if (tid < 63) {
        const int a = (tid - 54) / 3, bb = (tid - 54) % 3;
        double v = T(bb) * mom[69 + a];
        for (int d = 0; d < 3; ++d) v += R(bb, d) * mom[60 + 3 * a + d];
        CY[a][bb] = v;
    } else if (tid == 63) {
        double yy = 0.0, yp = 0.0;
        for (int bb = 0; bb < 3; ++bb) {
            double q2 = 0.0, l = 0.0, pp = T(bb) * mom[92 + bb];
            for (int d = 0; d < 3; ++d) {
                for (int e = 0; e < 3; ++e) q2 += R(bb, d) * R(bb, e) * mom[74 + k6(d, e)];
                l += R(bb, d) * mom[80 + d];
                pp += R(bb, d) * mom[83 + 3 * bb + d];
            }
            yy += q2 + 2.0 * T(bb) * l + mom[73] * T(bb) * T(bb);
            yp += pp;
        }
        SD = yy - 2.0 * yp + mom[95];
    }
    __syncthreads();
    if (tid == 0) {
        // (y x n)_a = y_b n_c - y_c n_b with (a, b, c) cyclic
        auto yy = [&](int c, int f, int bb, int ee) { return YY[k6(c, f)][k6(bb, ee)]; };
        double A[6][6], rhs[6];
        for (int a = 0; a < 3; ++a) {
            const int b1 = (a + 1) % 3, c1 = (a + 2) % 3;
            for (int a2 = a; a2 < 3; ++a2) {
                const int b2 = (a2 + 1) % 3, c2 = (a2 + 2) % 3;
                // (y_b1 n_c1 - y_c1 n_b1)(y_b2 n_c2 - y_c2 n_b2)
                A[a][a2] = yy(c1, c2, b1, b2) - yy(c1, b2, b1, c2) - yy(b1, c2, c1, b2) + yy(b1, b2, c1, c2);
            }
            for (int f = 0; f < 3; ++f) A[a][3 + f] = YN[k6(c1, f)][b1] - YN[k6(b1, f)][c1];
            double s1 = 0.0, s2 = 0.0; // sum_e n_c1 n_e y_b1 y_e, sum_e n_b1 n_e y_c1 y_e
            for (int e = 0; e < 3; ++e) { s1 += yy(c1, e, b1, e); s2 += yy(b1, e, c1, e); }
            rhs[a] = (s1 - CY[c1][b1]) - (s2 - CY[b1][c1]);
        }
        for (int c = 0; c < 3; ++c) {
            for (int f = c; f < 3; ++f) A[3 + c][3 + f] = mom[54 + k6(c, f)];
            double s1 = 0.0;
            for (int e = 0; e < 3; ++e) s1 += YN[k6(c, e)][e];
            rhs[3 + c] = s1 - mom[69 + c];
        }
        double r2 = mom[72];
        for (int c = 0; c < 3; ++c) {
            for (int f = 0; f < 3; ++f) r2 += yy(c, f, c, f);
            r2 -= 2.0 * CY[c][c];
        }
        rec[0] += mom[73];
        rec[1] += r2;
        int k = 2;
        for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c) rec[k++] += A[a][c];
        for (int a = 0; a < 6; ++a) rec[23 + a] += rhs[a];
        rec[29] += SD;
    }
    __syncthreads();
}

struct FreezeBufs {
    FreezeState *fz;          // nullptr: no frozen pairs in this launch
    const double *mom_part;   // [scan][row][FZ_NMOM]
    const uint32_t *act_cnt;  // [scan][row]
    const uint16_t *act_ids;  // [scan][row][FZ_CAP]
    uint32_t *act_all;        // [scan][rows x FZ_CAP]: the scan's list
};

// The reduce half of a launch that may freeze, by the scan's workgroup of RBLK threads: the record of the scan's pairs at
// the launch's pose in rec[] (shared, REC_STRIDE) -- slab rows (every pair / the active ones) + the frozen pairs from
// their moments; a freeze launch's moments and lists are folded and the freeze held or voided here.  `stride`: slab rows
// per scan, `rows`: the rows an ordinary launch of this scan writes.  What a rank freezes is its own business (sharded: its
// owned queries): the record it contributes is the same sum either way.
template <int NT>
__device__ __forceinline__ void freeze_fold(IcpState *S, FreezeState *F, int b, const double *__restrict__ partials, int stride, int rows, const FreezeBufs &fb, double *rec)
{
    static_assert(NT >= FZ_NMOM && NT >= 64, "the moments are copied and contracted by the first threads");
    const int fmode = F->mode;
    __shared__ double mom[FZ_NMOM];
    __shared__ double D[12];
    __shared__ uint32_t act_worst;
    __shared__ uint32_t act_total;
    // frozen: only the first rows were written, by the workgroups that had a piece of the scan's active list
    const int rows_live = fmode == 2 ? (int)((F->n_active + FZ_CAP - 1) / FZ_CAP) : rows;
    reduce_partials<NREC_PLANE, NT>(partials + (size_t)b * stride * REC_STRIDE, rows_live, rec);
    if (fmode == 1) {
        reduce_columns<FZ_NMOM, FZ_NMOM, NT>(fb.mom_part + (size_t)b * stride * FZ_NMOM, rows, mom);
        // the rows' active lists -> one list per scan (row order, slot order inside a row).  First whether every row could list
        // its active queries at all (a row beyond FZ_CAP: the freeze does not hold, NO list is built -- the prefix below would run
        // past the scan's region of act_all into the next scan's list) ...
        __shared__ uint32_t pre[NT];
        if (threadIdx.x == 0) { act_worst = 0u; act_total = 0u; }
        __syncthreads();
        {
            uint32_t worst = 0u, sum = 0u;
            for (int r = (int)threadIdx.x; r < rows; r += NT) {
                const uint32_t c = fb.act_cnt[(size_t)b * stride + r];
                worst = max(worst, c);
                sum += c;
            }
            if (worst > (uint32_t)FZ_CAP) atomicMax(&act_worst, worst);
            if (sum) atomicAdd(&act_total, sum); // (integers: order independent)
        }
        __syncthreads();
        if (act_worst <= (uint32_t)FZ_CAP) { // ... then where each row's piece starts, and the copy (rows * FZ_CAP entries at most: inside the scan's region)
            uint32_t run = 0;
            for (int r0 = 0; r0 < rows; r0 += NT) {
                const int r = r0 + (int)threadIdx.x;
                const uint32_t c = r < rows ? fb.act_cnt[(size_t)b * stride + r] : 0u;
                pre[threadIdx.x] = c;
                __syncthreads();
                for (int off = 1; off < NT; off <<= 1) { // inclusive scan of the chunk
                    const uint32_t t = (int)threadIdx.x >= off ? pre[threadIdx.x - off] : 0u;
                    __syncthreads();
                    pre[threadIdx.x] += t;
                    __syncthreads();
                }
                const uint32_t start = run + pre[threadIdx.x] - c;
                if (r < rows && (size_t)start + c <= (size_t)stride * FZ_CAP)
                    for (uint32_t i = 0; i < c; ++i) fb.act_all[(size_t)b * stride * FZ_CAP + start + i] = (uint32_t)r * (uint32_t)(BLK * SF_WIDE_QPL) + fb.act_ids[((size_t)b * stride + r) * FZ_CAP + i];
                run += pre[NT - 1];
                __syncthreads();
            }
        }
    } else if (fmode == 2) {
        if (threadIdx.x < FZ_NMOM) mom[threadIdx.x] = F->mom[threadIdx.x];
    }
    if (fmode != 0) {
        if (threadIdx.x == 0) {
            if (fmode == 1) { // the freeze pose is this launch's pose: D = identity
                for (int i = 0; i < 12; ++i) D[i] = (i % 5 == 0) ? 1.0 : 0.0;
            } else { // the updates solved since the freeze launch (freeze_after_update): T_now = D T_f whatever T_f is
                for (int i = 0; i < 12; ++i) D[i] = F->D[i];
            }
        }
        __syncthreads();
        frozen_record(mom, D, rec);
    }
    if (threadIdx.x == 0) {
        F->launch_mode = fmode;
        if (fmode == 1) {
            F->n_active = (int64_t)act_total;
            if (act_worst > (uint32_t)FZ_CAP) { // a row could not list its active queries: this launch was an ordinary one in two parts, nothing is frozen
                F->mode = 0;
                F->tries += 1;
            } else {
                F->mode = 2;
                F->froze += 1;
                if (S->froze_launch < 0) S->froze_launch = S->n_research; // (this launch's index: the solve that follows counts it; a later re-freeze after a thaw does not move it)
                F->motion0 = S->motion;
                for (int i = 0; i < 12; ++i) { F->Tf[i] = S->T[i]; F->D[i] = (i % 5 == 0) ? 1.0 : 0.0; }
                for (int i = 0; i < FZ_NMOM; ++i) F->mom[i] = mom[i];
            }
        }
    }
    __syncthreads();
}

// the solve half's bookkeeping, by the lane that solved (m0 / m1: IcpState::motion before / after the pose update)
// upd: the pose update the solve has just applied (solve_plane)
__device__ __forceinline__ void freeze_after_update(const IcpState *S, FreezeState *F, double m0, double m1, const FreezeParams &fp, int request, const double *upd)
{
    if (F->mode == 2) { // D <- upd D (3 x 4 affine composition, float64)
        double Dn[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 4; ++c) Dn[4 * r + c] = upd[4 * r] * F->D[c] + upd[4 * r + 1] * F->D[4 + c] + upd[4 * r + 2] * F->D[8 + c] + (c == 3 ? upd[4 * r + 3] : 0.0);
        }
        for (int i = 0; i < 12; ++i) F->D[i] = Dn[i];
    }
    if (F->mode == 2 && !(m1 - F->motion0 <= (double)F->guard * 0.98)) { // the next launch's pose is beyond what the frozen queries were cleared for
        F->mode = 0;
        F->tries += 1;
        F->thawed += 1;
    }
    if (F->mode == 0 && F->launch_mode == 0 && request && !S->done && F->tries < fp.max_tries) {
        // a guard of a few times the last update's motion (ICP steps shrink geometrically); while that is still large the
        // active lists would be long (or overflow: a freeze launch for nothing) -- wait for a later launch
        const float gd = fmaxf((float)(m1 - m0) * fp.guard_scale, fp.guard_min);
        if (gd <= fp.guard_max) { F->guard = gd; F->mode = 1; }
    }
}

// k_reduce_solve<2> for the launches that may freeze (see k_nn_red_fz); `request`: this solve may ask for a freeze launch
__global__ __launch_bounds__(RBLK) void k_reduce_solve_fz(IcpState *__restrict__ st, const double *__restrict__ partials, int nblocks, int n_src, int K,
                                                          const ScanBox *__restrict__ boxp, FreezeBufs fb, FreezeParams fp, int request)
{
    const int b = blockIdx.x;
    IcpState *S = st + b;
    if (S->done) return;
    FreezeState *F = fb.fz + b;
    __shared__ double rec[REC_STRIDE];
    freeze_fold<RBLK>(S, F, b, partials, nblocks, nblocks, fb, rec);
    if (threadIdx.x == 0) {
        for (int c = 0; c < NREC_PLANE; ++c) S->rec[c] = rec[c];
        double To[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) To[i] = S->T[i];
        const double m0 = S->motion;
        double upd[12];
        solve_plane(S, rec, n_src, K, upd);
        track_motion(S, To, boxp[b]);
        freeze_after_update(S, F, m0, S->motion, fp, request, upd);
    }
}

__global__ void k_fz_init(FreezeState *__restrict__ fz, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    FreezeState *F = fz + b;
    F->mode = 0; F->tries = 0; F->froze = 0; F->thawed = 0; F->n_active = 0; F->guard = 0.0f; F->launch_mode = 0; F->motion0 = 0.0;
}

// The reduce kernels of the stepping / sharded paths run on SBLK = 256 threads, one wave per SIMD: a 1024-thread workgroup at
// 122 VGPRs (the frozen-pair fold) needs the WHOLE register file of a compute unit, and when several ranks share one device a
// peer's k_gather_solve waves spinning for this very record sit on every unit -- measured: 4 ranks x 64 scans on one GPU never
// got their publish kernels placed (every rank timed out in the collective).  A slab of a shard has few rows anyway.
constexpr int SBLK = 256;
// multi-GPU split: reduce into the exchange buffer, all-reduce outside, then solve
template <int MODE>
__global__ __launch_bounds__(SBLK) void k_reduce_only(IcpState *__restrict__ st, const double *__restrict__ partials, int nblocks, double *__restrict__ xchg,
                                                      const uint32_t *__restrict__ own_off, int qpl, FreezeBufs fb)
{
    constexpr int NREC = MODE == 2 ? NREC_PLANE : NREC_P2P;
    const int b = blockIdx.x;
    __shared__ double rec[REC_STRIDE];
    if (st[b].done) { // keep the all-reduce shape: contribute zeros
        if (threadIdx.x < REC_STRIDE) xchg[(size_t)b * REC_STRIDE + threadIdx.x] = 0.0;
        return;
    }
    const int rows = own_off ? (int)((own_off[b + 1] - own_off[b] + BLK * qpl - 1) / (BLK * qpl)) : nblocks; // sharded: workgroups beyond the owned queries wrote nothing
    if (MODE == 2 && fb.fz) freeze_fold<SBLK>(st + b, fb.fz + b, b, partials, nblocks, rows, fb, rec);
    else reduce_partials<NREC, SBLK>(partials + (size_t)b * nblocks * REC_STRIDE, rows, rec);
    if (threadIdx.x < REC_STRIDE) xchg[(size_t)b * REC_STRIDE + threadIdx.x] = rec[threadIdx.x];
}

template <int MODE>
__global__ void k_solve_only(IcpState *__restrict__ st, const double *__restrict__ xchg, int n_src, int K, int batch, const ScanBox *__restrict__ boxp, float margin,
                             FreezeState *__restrict__ fz, FreezeParams fp, int request)
{
    constexpr int NREC = MODE == 2 ? NREC_PLANE : NREC_P2P;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    IcpState *S = st + b;
    if (S->done) return;
    double rec[REC_STRIDE];
#pragma unroll
    for (int c = 0; c < REC_STRIDE; ++c) rec[c] = c < NREC ? xchg[(size_t)b * REC_STRIDE + c] : 0.0;
#pragma unroll
    for (int c = 0; c < NREC; ++c) S->rec[c] = rec[c];
    double To[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) To[i] = S->T[i];
    const double m0 = S->motion;
    double upd[12];
    if (MODE == 1) solve_o3d(S, rec, n_src, 0, K);
    else solve_plane(S, rec, n_src, K, upd);
    track_motion(S, To, boxp[b]);
    if (margin > 0.0f && !S->done) own_check_motion(S, boxp[b], margin); // sharded path only (the scan's own box, computed on the device)
    if (MODE == 2 && fz) freeze_after_update(S, fz + b, m0, S->motion, fp, request, upd);
}

// ------------------------------------------------------------------ sharded step over the P2P transport, two kernels
// With a P2P communicator the all-reduce needs no kernel of its own: the workgroup that reduces a scan's slab stores the
// record straight into every rank's exchange region and raises that scan's flag there (k_reduce_publish); the wave that
// solves a scan waits for the scan's flags, adds the ranks' records in rank order and solves from LDS (k_gather_solve).
// Per iteration two small kernels instead of three (reduce / all-reduce / solve), and no 148 bytes of scratch per lane for the
// record.  Protocol, fences and failure behaviour as k_p2p_allreduce (sf_shard.cpp); flags are per scan (sf_p2p.hpp).
template <int MODE>
__global__ __launch_bounds__(SBLK) void k_reduce_publish(IcpState *__restrict__ st, const double *__restrict__ partials, int nblocks, const uint32_t *__restrict__ own_off, int qpl,
                                                         sf::P2pView v, FreezeBufs fb)
{
    constexpr int NREC = MODE == 2 ? NREC_PLANE : NREC_P2P;
    const int b = blockIdx.x, tid = (int)threadIdx.x, R = v.peers.nranks, me = v.peers.rank, par = (int)(v.seq & 1ull);
    __shared__ double rec[REC_STRIDE];
    __shared__ int dead;
    unsigned char *mine = v.peers.region[me];
    if (tid == 0) dead = __hip_atomic_load(sf::p2p_abort(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 1 : 0;
    if (st[b].done) { // keep the exchange's shape: contribute zeros
        if (tid < REC_STRIDE) rec[tid] = 0.0;
        __syncthreads();
    } else {
        const int rows = own_off ? (int)((own_off[b + 1] - own_off[b] + BLK * qpl - 1) / (BLK * qpl)) : nblocks;
        if (MODE == 2 && fb.fz) freeze_fold<SBLK>(st + b, fb.fz + b, b, partials, nblocks, rows, fb, rec);
        else reduce_partials<NREC, SBLK>(partials + (size_t)b * nblocks * REC_STRIDE, rows, rec); // (both end with a barrier)
    }
    if (dead) return; // the communicator is poisoned: k_gather_solve reports it
    for (int i = tid; i < REC_STRIDE * R; i += SBLK) {
        const int c = i & (REC_STRIDE - 1), r = i / REC_STRIDE;
        __hip_atomic_store(sf::p2p_slot(v.peers.region[r], par, me, R, v.max_count) + (size_t)b * REC_STRIDE + c, rec[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid < R) __hip_atomic_store(sf::p2p_sflag(v.peers.region[tid], b, me), v.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int MODE>
__global__ __launch_bounds__(64) void k_gather_solve(IcpState *__restrict__ st, int n_src, int K, const ScanBox *__restrict__ boxp, float margin, sf::P2pView v,
                                                     FreezeState *__restrict__ fz, FreezeParams fp, int request)
{
    constexpr int NREC = MODE == 2 ? NREC_PLANE : NREC_P2P;
    const int b = blockIdx.x, lane = (int)threadIdx.x, R = v.peers.nranks, me = v.peers.rank, par = (int)(v.seq & 1ull);
    __shared__ double rec[REC_STRIDE];
    __shared__ int verdict;
    unsigned char *mine = v.peers.region[me];
    if (lane == 0) verdict = __hip_atomic_load(sf::p2p_abort(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 2 : 0;
    __syncthreads();
    if (verdict == 0 && lane < R) {
        const long long t0 = wall_clock64();
        int bad = 0;
        while (__hip_atomic_load(sf::p2p_sflag(mine, b, lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < v.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (__hip_atomic_load(sf::p2p_abort(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { bad = 2; break; }
            if (wall_clock64() - t0 > v.spin_ticks) { bad = 1; break; }
        }
        if (bad) atomicMax(&verdict, bad);
    }
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    IcpState *S = st + b;
    if (verdict != 0) { // poison every region this rank reaches, report, stop the scan
        if (lane < R) __hip_atomic_store(sf::p2p_abort(v.peers.region[lane]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (lane == 0) {
            if (__hip_atomic_load(v.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) __hip_atomic_store(v.status, (uint32_t)verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            S->done = 1;
        }
        return;
    }
    if (lane < REC_STRIDE) { // the ranks' records in rank order: the same bits on every rank
        double s = 0.0;
        for (int r = 0; r < R; ++r) s += __hip_atomic_load(sf::p2p_slot(mine, par, r, R, v.max_count) + (size_t)b * REC_STRIDE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        rec[lane] = lane < NREC ? s : 0.0;
    }
    __syncthreads();
    if (lane == 0 && !S->done) {
#pragma unroll
        for (int c = 0; c < NREC; ++c) S->rec[c] = rec[c];
        double To[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) To[i] = S->T[i];
        const double m0 = S->motion;
        double upd[12];
        if (MODE == 1) solve_o3d(S, rec, n_src, 0, K);
        else solve_plane(S, rec, n_src, K, upd);
        track_motion(S, To, boxp[b]);
        if (margin > 0.0f && !S->done) own_check_motion(S, boxp[b], margin);
        if (MODE == 2 && fz) freeze_after_update(S, fz + b, m0, S->motion, fp, request, upd);
    }
}

// ------------------------------------------------------------------ the per-scan source in one pass
// The node's preprocessing of a scan (localization_node.cpp:290-297: applyUniformSubsample(2), cropPointCloudThroughRadius
// around the sensor) and setSourcePointCloud as two launches with no host synchronisation between the upload and the
// alignment: the same predicates as sf_cloud_subsample + sf_cloud_crop_radius (every stride-th point; finite and FLANN
// L2_Simple d2 < r2, unfused float32), the survivors in index order written straight into the alignment's source arrays,
// their count left in device memory (the REF_CPP kernels bound themselves by it).  Measured on the per-scan path: the
// separate operations are 9 launches and a host synchronisation (the count sizes the next launch) = ~100 us between
// the upload and the alignment for ~25 us of device work.
__device__ __forceinline__ bool prep_keep(const float *__restrict__ raw, int64_t n_raw, int stride, int64_t c, int64_t n_cand, float cx, float cy, float cz, float r2, float &x,
                                          float &y, float &z)
{
    if (c >= n_cand) return false;
    const int64_t i = c * stride;
    (void)n_raw;
    x = raw[3 * i]; y = raw[3 * i + 1]; z = raw[3 * i + 2];
    const bool fin = isfinite(x) && isfinite(y) && isfinite(z);
    const float dx = cx - x, dy = cy - y, dz = cz - z; // k_flag_radius of sf_cloud.hip, to the letter
    float d2 = dx * dx;
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    return fin && d2 < r2;
}

__global__ __launch_bounds__(BLK) void k_prep_count(const float *__restrict__ raw, int64_t n_raw, int stride, int64_t n_cand, float cx, float cy, float cz, float r2,
                                                    uint32_t *__restrict__ blk_count)
{
    float x, y, z;
    const bool keep = prep_keep(raw, n_raw, stride, (int64_t)blockIdx.x * BLK + threadIdx.x, n_cand, cx, cy, cz, r2, x, y, z);
    __shared__ uint32_t wsum[BLK / 64];
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(BLK) void k_prep_scatter(const float *__restrict__ raw, int64_t n_raw, int stride, int64_t n_cand, float cx, float cy, float cz, float r2,
                                                      const uint32_t *__restrict__ blk_count, float *__restrict__ X0x, float *__restrict__ X0y, float *__restrict__ X0z,
                                                      float4 *__restrict__ rec, int *__restrict__ n_out)
{
    // workgroups before this one (a few hundred at most: every workgroup adds them up itself, in integers)
    __shared__ uint32_t red[BLK / 64];
    __shared__ uint32_t wsum[BLK / 64];
    uint32_t before = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += BLK) before += blk_count[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = before;
    float x = 0.0f, y = 0.0f, z = 0.0f;
    const bool keep = prep_keep(raw, n_raw, stride, (int64_t)blockIdx.x * BLK + threadIdx.x, n_cand, cx, cy, cz, r2, x, y, z);
    const unsigned long long m = __ballot(keep);
    const uint32_t lane_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    uint32_t base = red[0] + red[1] + red[2] + red[3];
    for (int k = 0; k < wv; ++k) base += wsum[k];
    if (keep) {
        const uint32_t o = base + lane_rank;
        X0x[o] = x; X0y[o] = y; X0z[o] = z;
        rec[o] = make_float4(x, y, z, 0.0f);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = (int)(red[0] + red[1] + red[2] + red[3] + wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// ------------------------------------------------------------------ REF_CPP mode
// X <- init * X0 in float32, unfused, exactly icp_point_to_point.cpp:99-110,191-192
// n_live (single-scan alignments): the scan's point count read from device memory, so that a captured launch list
// stays valid while the count changes from scan to scan (the grid is sized for a rounded-up capacity); nullptr: n itself
__global__ void k_set_int(int *__restrict__ p, int v) { *p = v; }
__global__ void k_set_window(SfWindow *__restrict__ p, SfWindow w) { *p = w; }

__global__ void k_ref_init(const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z, int n, const int *__restrict__ n_live,
                           IcpState *__restrict__ st, float *__restrict__ Xx, float *__restrict__ Xy, float *__restrict__ Xz, float4 *__restrict__ corr)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_live) n = *n_live;
    if (i == 0) st[b].n_points = n; // nobody else touches this field
    if (i >= n) return;
    const IcpState *S = st + b;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = (float)S->T[k];
    const size_t o = (size_t)b * n + i;
    const float x = X0x[o], y = X0y[o], z = X0z[o];
    Xx[o] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], x), __fmul_rn(T[1], y)), __fmul_rn(T[2], z)), T[3]);
    Xy[o] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], x), __fmul_rn(T[5], y)), __fmul_rn(T[6], z)), T[7]);
    Xz[o] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], x), __fmul_rn(T[9], y)), __fmul_rn(T[10], z)), T[11]);
    corr[o] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(0)); // alive
}

// sourceTargetCorrespondences (icp_point_to_point.cpp:57-84): points without a match die
// for good (corr.w = -1), the survivors remember their target: its coordinates and its sorted position in one float4
// record, so that the per-iteration record kernel streams 16 bytes per point instead of gathering them from the map
// (measured at 32 scans in flight: k_ref_red 140-170 us as a gather).
// The wave-cooperative search of sf_nn.hpp (every lane of a wave takes part, dead points and the tail included);
// workgroups are placed like k_nn_red's: each XCD sweeps a contiguous eighth of the (cell-ordered) chunks for all scans
// in flight.  grid.x is padded to a multiple of 8.
template <bool WINDOW>
__global__ __launch_bounds__(BLK) void k_ref_nn(SfGrid g, const SfWindow *__restrict__ wdev, const float *__restrict__ Xx, const float *__restrict__ Xy,
                                                const float *__restrict__ Xz, int n, const int *__restrict__ n_live, const IcpState *__restrict__ st, float thr, int force,
                                                float4 *__restrict__ corr, int nblocks)
{
    SfWindow w;
    if (WINDOW) w = *wdev; // the map crop lives in device memory: it moves with the pose, a captured launch list does not
    else w.kind = 0;
    if (n_live) n = *n_live;
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int kk = L >> 3;
    const int b = kk % (int)gridDim.y;
    const int bx = (L & 7) * ((int)gridDim.x >> 3) + kk / (int)gridDim.y;
    if (bx >= nblocks) return;
    const IcpState *S = st + b;
    if (S->done || !(force || S->research)) return;
    __shared__ sf::WaveNN nn_ws[BLK / 64];
    const int i = bx * BLK + threadIdx.x;
    const size_t o = (size_t)b * n + (size_t)(i < n ? i : 0);
    const float4 old = i < n ? corr[o] : make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    const bool live = __float_as_int(old.w) >= 0;
    const float qx = live ? Xx[o] : 0.0f, qy = live ? Xy[o] : 0.0f, qz = live ? Xz[o] : 0.0f;
    // a re-search (cpp:221-224: the error has stalled, the points have hardly moved) starts from the target the point
    // already has: exact all the same (sf_nn.hpp, seed), and almost every neighbour range is pruned unvisited
    sf::NNHit seed{0.0f, -1, 0.0f, 0.0f, 0.0f, 0.0f};
    if (live && !force) { seed.d2 = sf::l2_simple(qx, qy, qz, old.x, old.y, old.z); seed.j = __float_as_int(old.w); seed.px = old.x; seed.py = old.y; seed.pz = old.z; }
    const sf::NNHit hit = sf::nn_search_wave<WINDOW, true>(g, w, live, qx, qy, qz, thr, &nn_ws[threadIdx.x >> 6], seed);
    if (live) corr[o] = make_float4(hit.px, hit.py, hit.pz, __int_as_float(hit.j));
}

// optional in-place X <- step * X (float32, unfused), then the 17-scalar record over the
// live pairs: n, sum s, sum t, sum s t^T, sum ||s - t|| (float32 norm, as
// calculateErrorMetric icp_point_to_point.cpp:161-170 computes it per pair)
__global__ __launch_bounds__(BLK) void k_ref_red(float *__restrict__ Xx, float *__restrict__ Xy, float *__restrict__ Xz, int n,
                                                 const int *__restrict__ n_live, const IcpState *__restrict__ st, const float4 *__restrict__ corr, int apply_step,
                                                 int only_if_research, double *__restrict__ partials, int nblocks)
{
    if (n_live) n = *n_live;
    const int b = blockIdx.y;
    const IcpState *S = st + b;
    if (S->done) return;
    if (only_if_research && !S->research) return;
    const bool do_step = apply_step && S->step_pending;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = S->step[k];
    double acc[NREC_P2P];
#pragma unroll
    for (int c = 0; c < NREC_P2P; ++c) acc[c] = 0.0;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += nblocks * BLK) {
        const size_t o = (size_t)b * n + i;
        const float4 p = corr[o];
        if (__float_as_int(p.w) < 0) continue;
        float x = Xx[o], y = Xy[o], z = Xz[o];
        if (do_step) {
            const float nx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], x), __fmul_rn(T[1], y)), __fmul_rn(T[2], z)), T[3]);
            const float ny = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], x), __fmul_rn(T[5], y)), __fmul_rn(T[6], z)), T[7]);
            const float nz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], x), __fmul_rn(T[9], y)), __fmul_rn(T[10], z)), T[11]);
            x = nx; y = ny; z = nz;
            Xx[o] = x; Xy[o] = y; Xz[o] = z;
        }
        const float dx = x - p.x, dy = y - p.y, dz = z - p.z;
        const float nrm = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dz, dz))));
        const double sx = x, sy = y, sz = z, tx = p.x, ty = p.y, tz = p.z;
        acc[0] += 1.0;
        acc[1] += sx; acc[2] += sy; acc[3] += sz;
        acc[4] += tx; acc[5] += ty; acc[6] += tz;
        acc[7] += sx * tx; acc[8] += sx * ty; acc[9] += sx * tz;
        acc[10] += sy * tx; acc[11] += sy * ty; acc[12] += sy * tz;
        acc[13] += sz * tx; acc[14] += sz * ty; acc[15] += sz * tz;
        acc[16] += (double)nrm;
    }
    block_reduce_store<NREC_P2P>(acc, partials + ((size_t)b * nblocks + blockIdx.x) * REC_STRIDE);
}

// step = Kabsch(record); T <- step * T in float32 (icp_point_to_point.cpp:226-228)
__device__ __forceinline__ void ref_take_step(IcpState *S, const double *rec, float error)
{
    double upd[16];
    kabsch_from_record(rec, upd);
    float sf[16], Tf[16], Tn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sf[i] = (float)upd[i]; Tf[i] = (float)S->T[i]; }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            Tn[4 * r + c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(sf[4 * r], Tf[c]), __fmul_rn(sf[4 * r + 1], Tf[4 + c])), __fmul_rn(sf[4 * r + 2], Tf[8 + c])),
                                      __fmul_rn(sf[4 * r + 3], Tf[12 + c]));
#pragma unroll
    for (int i = 0; i < 16; ++i) S->T[i] = Tn[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) S->step[i] = sf[i];
    S->step_pending = 1;
    S->last_error = error;
    S->iterations += 1;
}

// phase 0: after the initial search (cpp:195-200); phase 1: top of loop iteration
// (cpp:209-224); phase 2: after a lazy re-search (cpp:223-226).  One lane; rec = the record of the live pairs.
__device__ __forceinline__ void ref_decide(IcpState *S, const double *rec, const IcpParams prm, int phase)
{
    for (int c = 0; c < NREC_P2P; ++c) S->rec[c] = rec[c];
    const double n = rec[0];
    if (phase == 0) {
        S->n_corr = (int)n;
        S->n_research = 1;
        if (n < 10) { S->flags |= SF_ICP_FLAG_FEW_CORR; S->done = 1; }
        return;
    }
    if (phase == 1) {
        S->step_pending = 0;
        if (!(n >= 1)) { S->flags |= SF_ICP_FLAG_FEW_CORR; S->done = 1; return; }
        const float error = (float)(rec[16] / n);
        if (error < prm.accept) { S->last_error = error; S->done = 1; return; }
        if (fabsf(S->last_error - error) < prm.eps) { S->research = 1; S->err_pending = error; return; }
        ref_take_step(S, rec, error);
        return;
    }
    // phase 2: new correspondences are in rec
    S->research = 0;
    S->n_corr = (int)n;
    S->n_research += 1;
    if (!(n >= 1)) { S->flags |= SF_ICP_FLAG_FEW_CORR; S->done = 1; return; }
    ref_take_step(S, rec, S->err_pending);
}

__global__ __launch_bounds__(RBLK) void k_ref_decide(IcpState *__restrict__ st, const double *__restrict__ partials, int nblocks, IcpParams prm, int phase)
{
    const int b = blockIdx.x;
    IcpState *S = st + b;
    if (S->done) return;
    if (phase == 2 && !S->research) return;
    __shared__ double rec[REC_STRIDE];
    reduce_partials<NREC_P2P>(partials + (size_t)b * nblocks * REC_STRIDE, nblocks, rec);
    if (threadIdx.x != 0) return;
    ref_decide(S, rec, prm, phase);
}

// ------------------------------------------------------------------ REF_CPP, the whole alignment in ONE launch
// For scans small enough that every workgroup is resident at once (the per-scan path of the node: ~13 k points after the
// stride-2 subsample and the 10 m crop = 64 workgroups) the 4 + 5 K - 1 launches above -- most of them returning at once
// because their phase is not active, each still a launch and a dependent kernel boundary -- become one: a workgroup keeps
// its 256 points, their correspondences and neighbours in registers from the first search to the last step, the record of
// the live pairs goes through the same per-workgroup rows (block_reduce_store) and the same fixed-order sum
// (reduce_partials) as above, and EVERY workgroup evaluates the controller (ref_decide) on its own copy of the state in
// LDS: same inputs, same code, same decision everywhere, so all workgroups of a scan walk the same sequence of phases and
// meet at the same grid barriers -- one per record, none for the decisions.  Results are bit-identical to the launch
// list (tests/test_gpu_round2.py).
// Grid barrier (per scan, workgroups of one scan only): a monotonic arrival counter.  Producer side: every wave drains
// its stores, workgroup barrier, lane 0 agent-scope release fence + drain, agent-scope add.  Consumer side: lane 0 polls
// with relaxed agent-scope loads (served by L2, not the CU's L1) and s_sleep, ONE agent-scope acquire fence + drain,
// workgroup barrier, plain loads (MI355X: per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed
// by another CU's stores -- workgroup scope is not enough).  Every spin is bounded (FUSED_SPIN_TICKS of the 100 MHz
// clock): a workgroup that gives up raises SF_ICP_FLAG_BARRIER_TIMEOUT in the state and leaves, so the grid always
// drains; the host then redoes that alignment through the launch list (sf_icp_fetch_results) -- the caller never sees it.
// Residency is what the barrier rests on: the library keeps a per-device ledger of the single-launch grids in flight
// (all contexts of the process) and admits a new one only while the sum of their shares of the device stays below one;
// what does not fit takes the launch list.  Another PROCESS on the device is beyond the ledger: that is what the
// bounded spin and the redo are for.
constexpr long long FUSED_SPIN_TICKS = 50000000; // 0.5 s (a barrier normally completes in microseconds; a timed-out alignment is redone through the launch list)
#ifdef SF_FUSED_TRACE
__device__ unsigned long long g_fused_trace[512];
#define FTRACE(code) do { if (bx == 0 && b == 0 && threadIdx.x == 0 && tr_n < 511) g_fused_trace[1 + tr_n++] = ((unsigned long long)wall_clock64() << 8) | (unsigned)(code); } while (0)
#else
#define FTRACE(code) do { } while (0)
#endif

__device__ __forceinline__ bool ref_grid_barrier(uint32_t *ctr, uint32_t target, int *ok_lds)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > FUSED_SPIN_TICKS) { ok = 0; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}

template <bool WINDOW>
__global__ __launch_bounds__(BLK) void k_ref_fused(SfGrid g, SfWindow w, const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z, int n,
                                                   const int *__restrict__ n_live, IcpState *__restrict__ st, IcpParams prm, float thr, double *__restrict__ partials, int nblocks,
                                                   uint32_t *__restrict__ bar, IcpState *__restrict__ host_out)
{
    const int b = blockIdx.y, bx = blockIdx.x;
    if (n_live) n = *n_live; // single scan whose count was left on the device (sf_icp_set_source_scan)
    __shared__ IcpState S;
    __shared__ double rec[REC_STRIDE];
    __shared__ sf::WaveNN nn_ws[BLK / 64];
    __shared__ int bar_ok;
    static_assert(sizeof(IcpState) % 4 == 0, "copied word by word");
    for (int k = threadIdx.x; k < (int)(sizeof(IcpState) / 4); k += BLK) reinterpret_cast<uint32_t *>(&S)[k] = reinterpret_cast<const uint32_t *>(st + b)[k];
    __syncthreads();
    if (threadIdx.x == 0) S.n_points = n;
    uint32_t *ctr = bar + 2 * b, *fin = bar + 2 * b + 1;
    // TWO slabs per scan, used in turn: a workgroup that is through a barrier may write its next row while a slower one is
    // still summing the rows of the record before (there is one barrier per record, between writing and reading; found by
    // the soak tests as a rare last-bits difference under load).  Nobody can be more than one record ahead -- the next
    // barrier needs everybody's arrival -- so two buffers are enough.
    double *const slab_even = partials + (size_t)b * nblocks * REC_STRIDE, *const slab_odd = partials + ((size_t)gridDim.y + b) * nblocks * REC_STRIDE;
    const int i = bx * BLK + (int)threadIdx.x;
    float x = 0.0f, y = 0.0f, z = 0.0f, tx = 0.0f, ty = 0.0f, tz = 0.0f;
    int corr = -1;
    if (i < n) { // X <- init * X0 (k_ref_init)
        const size_t o = (size_t)b * n + i;
        const float x0 = X0x[o], y0 = X0y[o], z0 = X0z[o];
        float T[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = (float)S.T[k];
        x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], x0), __fmul_rn(T[1], y0)), __fmul_rn(T[2], z0)), T[3]);
        y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], x0), __fmul_rn(T[5], y0)), __fmul_rn(T[6], z0)), T[7]);
        z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], x0), __fmul_rn(T[9], y0)), __fmul_rn(T[10], z0)), T[11]);
        corr = 0;
    }
    uint32_t passed = 0; // grid barriers behind this workgroup
    bool alive = true;   // false: a barrier timed out
#ifdef SF_FUSED_TRACE
    int tr_n = 0;
#endif
    FTRACE(0);
    // sourceTargetCorrespondences (k_ref_nn): points without a match die for good
    auto search = [&](bool first) {
        const bool live = corr >= 0;
        sf::NNHit seed{0.0f, -1, 0.0f, 0.0f, 0.0f, 0.0f}; // a re-search starts from the target the point already has (k_ref_nn)
        if (live && !first) { seed.d2 = sf::l2_simple(x, y, z, tx, ty, tz); seed.j = corr; seed.px = tx; seed.py = ty; seed.pz = tz; }
        const sf::NNHit hit = sf::nn_search_wave<WINDOW, true>(g, w, live, x, y, z, thr, &nn_ws[threadIdx.x >> 6], seed);
        if (live) { corr = hit.j; tx = hit.px; ty = hit.py; tz = hit.pz; }
        __syncthreads();
        FTRACE(1);
    };
    // k_ref_red + the fixed-order sum of k_ref_decide: rec <- record of the live pairs (after the pending step, if asked)
    auto record = [&](bool apply_step) {
        double acc[NREC_P2P];
#pragma unroll
        for (int c = 0; c < NREC_P2P; ++c) acc[c] = 0.0;
        if (corr >= 0) {
            if (apply_step && S.step_pending) {
                const float nx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(S.step[0], x), __fmul_rn(S.step[1], y)), __fmul_rn(S.step[2], z)), S.step[3]);
                const float ny = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(S.step[4], x), __fmul_rn(S.step[5], y)), __fmul_rn(S.step[6], z)), S.step[7]);
                const float nz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(S.step[8], x), __fmul_rn(S.step[9], y)), __fmul_rn(S.step[10], z)), S.step[11]);
                x = nx; y = ny; z = nz;
            }
            const float dx = x - tx, dy = y - ty, dz = z - tz;
            const float nrm = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dz, dz))));
            const double sx = x, sy = y, sz = z, px = tx, py = ty, pz = tz;
            acc[0] += 1.0;
            acc[1] += sx; acc[2] += sy; acc[3] += sz;
            acc[4] += px; acc[5] += py; acc[6] += pz;
            acc[7] += sx * px; acc[8] += sx * py; acc[9] += sx * pz;
            acc[10] += sy * px; acc[11] += sy * py; acc[12] += sy * pz;
            acc[13] += sz * px; acc[14] += sz * py; acc[15] += sz * pz;
            acc[16] += (double)nrm;
        }
        double *slab = (passed & 1u) ? slab_odd : slab_even;
        block_reduce_store<NREC_P2P>(acc, slab + (size_t)bx * REC_STRIDE);
        FTRACE(2);
        ++passed;
        alive = ref_grid_barrier(ctr, passed * (uint32_t)nblocks, &bar_ok);
        FTRACE(3);
        if (alive) reduce_partials<NREC_P2P, BLK>(slab, nblocks, rec);
        FTRACE(4);
    };
    auto decide = [&, prm](int phase) {
        if (threadIdx.x == 0) ref_decide(&S, rec, prm, phase);
        __syncthreads();
        FTRACE(5);
    };
    const int K = prm.num_iters;
    if (!S.done) {
        search(true);
        record(false);
        if (alive) decide(0);
        for (int it = 0; alive && it < K && !S.done; ++it) {
            decide(1);
            if (S.done) break;
            if (S.research) {
                search(false);
                record(false);
                if (!alive) break;
                decide(2);
                if (S.done) break;
            }
            if (it + 1 < K) record(true);
        }
    }
    __syncthreads();
#ifdef SF_FUSED_TRACE
    if (bx == 0 && b == 0 && threadIdx.x == 0) g_fused_trace[0] = (unsigned long long)tr_n;
#endif
    if (!alive) {
        if (threadIdx.x == 0) {
            atomicOr(&st[b].flags, SF_ICP_FLAG_BARRIER_TIMEOUT);
            if (host_out) host_out[b].flags = SF_ICP_FLAG_BARRIER_TIMEOUT;
        }
        return; // the counters stay as they are: the host resets them when it sees the flag
    }
    if (bx == 0)
        for (int k = threadIdx.x; k < (int)(sizeof(IcpState) / 4); k += BLK) {
            const uint32_t v = reinterpret_cast<const uint32_t *>(&S)[k];
            reinterpret_cast<uint32_t *>(st + b)[k] = v;
            if (host_out) reinterpret_cast<uint32_t *>(host_out + b)[k] = v; // pinned host memory: the result needs no copy back
        }
    // the workgroup that leaves last puts the counters back to zero for the next launch (everybody is past its last barrier)
    if (threadIdx.x == 0) {
        const uint32_t left = __hip_atomic_fetch_add(fin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (uint32_t)nblocks - 1u) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------ O3D_P2P / P2PLANE, the whole alignment in ONE launch
// The same single-launch form for the float64 modes (the Python node's registration_icp call, the mapping flow, small
// batches): per iteration the body of k_nn_red -- with the lane's neighbour cache (position of its last search, runner-up
// bound, neighbour, normal) kept in REGISTERS instead of three float4 streams -- the slab row, one grid barrier, the
// fixed-order slab sum and the solve of k_reduce_solve evaluated by every workgroup on its LDS copy of the state.
// Bit-identical to the launch list (same rows, same sums, same certificate; tests/test_gpu_round2.py).
template <int MODE, bool WINDOW, bool REUSE>
__global__ __launch_bounds__(BLK) void k_icp_fused(SfGrid g, SfWindow w, const float *__restrict__ X0x, const float *__restrict__ X0y, const float *__restrict__ X0z, int n,
                                                   IcpState *__restrict__ st, float thr, int K, double *__restrict__ partials, int nblocks, uint32_t *__restrict__ bar,
                                                   IcpState *__restrict__ host_out)
{
    constexpr int NREC = MODE == 2 ? NREC_PLANE : NREC_P2P;
    const int b = blockIdx.y, bx = blockIdx.x;
    __shared__ IcpState S;
    __shared__ double rec[REC_STRIDE];
    __shared__ sf::WaveNN nn_ws[BLK / 64];
    __shared__ double stage[BLK / 64][32];
    __shared__ int bar_ok;
    for (int k = threadIdx.x; k < (int)(sizeof(IcpState) / 4); k += BLK) reinterpret_cast<uint32_t *>(&S)[k] = reinterpret_cast<const uint32_t *>(st + b)[k];
    __syncthreads();
    uint32_t *ctr = bar + 2 * b, *fin = bar + 2 * b + 1;
    double *const slab_even = partials + (size_t)b * nblocks * REC_STRIDE, *const slab_odd = partials + ((size_t)gridDim.y + b) * nblocks * REC_STRIDE; // used in turn: see k_ref_fused
    const int slot = bx * BLK + (int)threadIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool have = slot < n;
    double x0 = 0.0, y0 = 0.0, z0 = 0.0;
    if (have) {
        const size_t o = (size_t)b * n + (size_t)slot;
        x0 = X0x[o]; y0 = X0y[o]; z0 = X0z[o];
    }
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0; // the lane's neighbour cache (k_nn_red keeps it in memory)
    uint32_t passed = 0;
    bool alive = true;
    const int launches = MODE == 1 ? K + 1 : K; // what the launch list enqueues
    for (int it = 0; it < launches && !S.done; ++it) {
        double sx = 0, sy = 0, sz = 0;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (have) {
            sx = S.T[0] * x0 + S.T[1] * y0 + S.T[2] * z0 + S.T[3];
            sy = S.T[4] * x0 + S.T[5] * y0 + S.T[6] * z0 + S.T[7];
            sz = S.T[8] * x0 + S.T[9] * y0 + S.T[10] * z0 + S.T[11];
            qx = (float)sx; qy = (float)sy; qz = (float)sz;
        }
        sf::NNHit hit;
        float4 tn;
        sf::NNHit seed;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool live = REUSE && S.n_research > 0;
        const bool need = reuse_certificate_pos(have, qx, qy, qz, thr, live ? c0 : z4, live ? c1 : z4, live ? c2 : z4, hit, tn, seed);
        if (__ballot(need) != 0ull) {
            const sf::NNHit h = sf::nn_search_wave<WINDOW>(g, w, need, qx, qy, qz, thr, &nn_ws[wv], seed);
            if (need) {
                hit = h;
                if (MODE == 2 && h.j >= 0) tn = g.nrm[h.j];
                if (REUSE) {
                    c0 = make_float4(qx, qy, qz, sqrtf(h.lb2));
                    c1 = make_float4(h.px, h.py, h.pz, __int_as_float(h.j));
                    c2 = tn;
                }
            }
        }
        LanePair P;
        P.sx = sx; P.sy = sy; P.sz = sz;
        P.px = hit.px; P.py = hit.py; P.pz = hit.pz;
        P.tn = tn;
        P.ok = hit.j >= 0;
        const PairTerms T = pair_terms<MODE>(P);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = 0.0;
            add_half<MODE>(T, h, v);
            if (MODE == 1 && h == 1) {
                const double t1 = wave_reduce_1(v[0]);
                if (lane == 0) stage[wv][16] = t1;
            } else {
                const double t0 = wave_reduce_16(v);
                if ((lane & 3) == 0) stage[wv][16 * h + (lane >> 2)] = t0;
            }
        }
        __syncthreads();
        double *slab = (passed & 1u) ? slab_odd : slab_even;
        if (threadIdx.x < NREC) {
            const int c = threadIdx.x;
            slab[(size_t)bx * REC_STRIDE + c] = ((stage[0][c] + stage[1][c]) + stage[2][c]) + stage[3][c];
        }
        ++passed;
        alive = ref_grid_barrier(ctr, passed * (uint32_t)nblocks, &bar_ok);
        if (!alive) break;
        reduce_partials<NREC, BLK>(slab, nblocks, rec);
        if (threadIdx.x == 0) {
            for (int c = 0; c < NREC; ++c) S.rec[c] = rec[c];
            if (MODE == 1) solve_o3d(&S, rec, n, 0, K);
            else solve_plane(&S, rec, n, K);
        }
        __syncthreads();
    }
    __syncthreads();
    if (!alive) {
        if (threadIdx.x == 0) {
            atomicOr(&st[b].flags, SF_ICP_FLAG_BARRIER_TIMEOUT);
            if (host_out) host_out[b].flags = SF_ICP_FLAG_BARRIER_TIMEOUT;
        }
        return;
    }
    if (bx == 0)
        for (int k = threadIdx.x; k < (int)(sizeof(IcpState) / 4); k += BLK) {
            const uint32_t v = reinterpret_cast<const uint32_t *>(&S)[k];
            reinterpret_cast<uint32_t *>(st + b)[k] = v;
            if (host_out) reinterpret_cast<uint32_t *>(host_out + b)[k] = v;
        }
    if (threadIdx.x == 0) {
        const uint32_t left = __hip_atomic_fetch_add(fin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (uint32_t)nblocks - 1u) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

} // namespace

// ==================================================================== host side
struct sf_icp {
    sf_ctx *ctx = nullptr;
    IcpParams prm{};
    int debug = 0;
    // source
    sf::DevBuf X0, X;        // SoA: x[B*n], y[B*n], z[B*n]
    sf::DevBuf X0r;          // the same points as float4 records (gather source of the query ordering)
    sf::DevBuf qcache;       // neighbour reuse, two float4 arrays of cache_n entries: (neighbour, its index), (neighbour's normal, E) -- O3D_P2P: the second array holds E alone, as floats
    int64_t cache_n = 0;
    bool reuse = true;       // sf_icp_set_nn_reuse
    sf::DevBuf Xq, qkeys, qkeys2, qidx, qidx2; // cell-ordered copy of X0 and the sort's buffers
    int order = SF_ORDER_AUTO;
    bool ordered = false;    // this alignment reads Xq
    sf::DevBuf corr;         // float4 [B*n] (REF_CPP): a point's target (x, y, z, sorted position or -1 = dead)
    int64_t n = 0;           // points per scan
    bool n_on_device = false; // sf_icp_set_source_scan: n is an upper bound, the count itself is in n_dev (single scan, REF_CPP)
    int64_t n_cap = 0;       // single scan: n rounded up (launch geometry of the REF_CPP kernels)
    int64_t plane = 0;       // component stride of the SoA arrays X0 / X / Xq
    sf::DevBuf n_dev;        // the point count in device memory (single-scan REF_CPP kernels read it)
    int batch = 0;
    std::vector<double> inits; // batch * 16
    std::vector<double> inits_uploaded; // what d_inits holds (the copy is skipped while nothing changed)
    double *h_inits = nullptr;          // pinned staging of the batch's initial transforms
    size_t h_inits_cap = 0;
    hipEvent_t inits_ev = nullptr;      // recorded after the staging buffer's last copy
    bool inits_ev_pending = false;
    uint32_t d_inits_epoch = 0xffffffffu;
    bool have_source = false;
    // target
    sf_map *map = nullptr;
    sf_map *own_map = nullptr;
    sf_cloud *own_cloud = nullptr;
    // device state
    sf::DevBuf state, d_inits, partials, xchg_own;
    void *xchg = nullptr;
    int64_t xchg_bytes = 0;
    int nblocks = 0;         // workgroups of 256 points covering a scan (REF_CPP kernels, owned-query compaction)
    int nblocks_nn = 0;      // k_nn_red workgroups per scan = slab rows (256 * qpl queries each)
    int qpl = 1;             // queries per lane of k_nn_red: 1 up to wide_from points per scan, SF_WIDE_QPL beyond (part of the summation order)
    int64_t wide_from = WIDE_SCAN_POINTS; // sf_icp_set_wide_scan_points: scans above this many points take two queries per lane (and may freeze)
    bool wide_auto = true;   // no explicit limit: scans above WIDE_AUTO_POINTS are wide too when the batch is beyond every single-launch kernel (set_source_common)
    std::vector<IcpState> h_state;
    // sharding
    bool shard = false;
    float xlo = 0, xhi = 0;
    sf::DevBuf own_idx, own_blk, own_count, own_off, own_keep; // sharded path: owned-query compaction (own_keep: the screening ballots, 1 bit per point)
    std::vector<uint32_t> h_own;                      // host copy of counts / offsets
    int64_t own_total = 0;                            // owned-query candidates of this rank (all scans)
    float own_margin = 1.0f;                          // sf_icp_set_shard_margin
    int own_nblocks = 1;                              // workgroups per scan on the sharded path (largest scan)
    sf::DevBuf d_box;                        // sf::MinMaxDev: bounding box of the source batch (finite points), left on the device
    sf::DevBuf d_boxes, d_box_parts;         // ScanBox per scan (motion bound of the reuse certificate, owned-array margin), and the partial boxes of their reduction
    sf::DevBuf stage;                        // persistent upload staging of sf_icp_set_source* (AoS)
    int last_mode = 0;
    // REF_CPP in one launch (k_ref_fused)
    bool fused = true;       // sf_icp_set_fused
    bool last_fused = false; // the last alignment ran as the single launch
    sf::DevBuf bar;          // per scan: {arrival counter, departure counter} of the grid barrier
    int fused_limit[3] = {-1, -1, -1}; // per mode: workgroups that are certainly resident together (-1: not asked yet)
    double fused_share = 0.0; // share of the device the single-launch grid in flight holds in the ledger (0: none)
    int64_t fused_redone = 0; // alignments redone through the launch list after a barrier gave up
    bool inject_timeout = false; // test hook (sf_icp_test_inject_barrier_timeout): treat the next single-launch alignment as timed out
    int64_t fused_launches = 0;
    IcpState *h_pin = nullptr; // pinned, device-visible: the single-launch forms write the final states here themselves (no copy back)
    size_t h_pin_cap = 0;
    // graph
    bool use_graph = false;
    hipGraphExec_t graph_exec = nullptr;
    int64_t graph_captures = 0, graph_replays = 0; // sf_icp_graph_counts
    // everything a captured launch list bakes in: kernel arguments passed by value (thresholds, IcpParams, the
    // SfGrid geometry and pointers) and the addresses of this object's buffers.  A replay is only valid while all of
    // it is unchanged; sf_map stamps every build / normals pass with a process-unique generation, DevBuf counts its
    // reallocations (epoch).
    struct GraphKey {
        int mode = -1, iters = -1, batch = -1, window = -1, ordered = -1, reuse = -1;
        int64_t n = -1;
        const void *map = nullptr;
        uint64_t map_generation = 0, epochs = 0;
        float max_corr = 0, accept = 0, eps = 0;
        bool operator==(const GraphKey &o) const
        {
            return mode == o.mode && iters == o.iters && batch == o.batch && window == o.window && ordered == o.ordered && reuse == o.reuse && n == o.n && map == o.map &&
                   map_generation == o.map_generation && epochs == o.epochs && max_corr == o.max_corr && accept == o.accept && eps == o.eps;
        }
    } graph_key;
    // Consecutive asynchronous alignments of the launch list overlap: an alignment's last launches (frozen pairs: fourteen 16 us
    // kernels on an otherwise idle device) run under the next one's searching launches.  Two LANES take turns; each has its own
    // stream and its own copy of everything an alignment writes (`other` holds the lane that is not in the members of this struct).
    // Ordering (LaneScope): a lane starts after the context's stream as it stood when the inputs -- source, initial poses,
    // target -- last changed, and the context's stream waits for every alignment right after it is enqueued, so whatever the caller
    // enqueues next (a fetch, an upload, a map rebuild) is ordered behind it as before; only back-to-back alignments of unchanged
    // inputs run side by side.  Same kernels, same data, same results (tests/test_gpu_pipeline.py).
    struct Lane {
        sf::DevBuf X, qcache, Xq, qkeys, qkeys2, qidx, qidx2, corr, state, d_inits, partials, fz_state, fz_part, fz_cnt, fz_ids, fz_all, qtkey, tseg, tile_stats;
        hipGraphExec_t graph_exec = nullptr;
        GraphKey graph_key;
        std::vector<double> inits_uploaded;
        uint32_t d_inits_epoch = 0xffffffffu;
    } other;
    // Two SOURCE SETS (the members hold one, `other_src` the other): a source set while an alignment of this object is still
    // unfetched goes into the set that alignment does not read, on the stream of the lane the next alignment will take, so the
    // next batch's upload / conversion runs beside the alignment in flight instead of behind it (SrcScope).  Per set an event
    // marks the end of the last alignment that read it; `src_ready` the end of the last upload that ran on a lane's stream.
    struct SrcSet {
        sf::DevBuf X0, X0r, stage, d_boxes, d_box_parts, n_dev;
        int64_t n = 0, n_cap = 0, plane = 0;
        int batch = 0, nblocks = 0, nblocks_nn = 0, qpl = 1;
        bool have_source = false, n_on_device = false;
    } other_src;
    int src_set = 0;              // which set the members hold
    hipEvent_t src_used[2] = {nullptr, nullptr}, src_ready = nullptr;
    bool src_used_rec[2] = {false, false};
    bool src_ahead = false;       // the members' source was written on a lane's stream (src_ready) and no alignment has been ordered behind it yet
    // started[l]: recorded on lane l's stream when its latest alignment begins (behind everything it waits for).  A source written
    // ahead starts its upload behind the START of the alignment on the other lane: two uploads issued back to back -- the host
    // fetched two results that became ready together -- would otherwise share the link, end together and let both alignments
    // start together, a lockstep in which nothing overlaps (measured: upload + upload, then alignment + alignment, for good).
    hipEvent_t started[2] = {nullptr, nullptr};
    bool started_rec[2] = {false, false};
    hipEvent_t unmarked_ev = nullptr;
    bool src_unmarked_use = false; // an alignment read a source set before the lanes' events existed (the per-scan path never creates them): the first source written ahead waits for the context's stream instead
    // what the lane's last alignment was (sf_icp_fetch_previous reads the OTHER lane's states with the other lane's description)
    struct LaneMeta {
        bool valid = false;
        int batch = 0, mode = 0;
        std::vector<double> inits;
    } meta, other_meta;
    bool prev_ok = false;         // the alignment before the latest one ran on the other lane and has not been overwritten or fetched
    std::vector<IcpState> h_prev;
    sf::DevBuf order_lut;         // k_order_lut_*: histogram (uint32) and key table (uint16) over the walk order of the attached index
    uint64_t order_lut_gen = 0;   // sf_map::generation the table was built for
    int order_lut_shift = 0, order_lut_on = 1; // order_lut_on: SF_ORDER_LUT=0 in the environment switches the table off (A/B runs)
    int lane = 0;                 // the lane the members hold
    int pipeline = 1;             // sf_icp_set_pipeline: 0 = every alignment on the context's stream
    hipStream_t lane_stream[2] = {nullptr, nullptr};
    hipEvent_t lane_done[2] = {nullptr, nullptr}, main_mark = nullptr;
    bool lane_used[2] = {false, false}; // lane_done[l] has been recorded
    bool unfetched = false;       // an alignment has been enqueued since the last fetch (the next one may run beside it)
    bool mark_valid = false;      // main_mark stands for the inputs as described by the fields below
    uint64_t src_version = 0, mark_src_version = 0, mark_map_generation = 0;
    const void *mark_map = nullptr;
    SfWindow mark_window{};
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    int64_t prof_launches = 0;
    double prof_ms = 0;
    std::vector<float> prof_each;   // duration of every profiled launch, in launch order
    std::vector<int> ev_kind;       // SF_PROF_* of every event pair
    std::vector<float> prof_phase[SF_PROF_KINDS]; // sharded path: durations of the other phases, in order
    sf::DevBuf nn_stats;            // per profiled k_nn_red launch: {queries that searched, waves that searched}
    // frozen pairs (k_nn_red_fz)
    int freeze = 1;                 // sf_icp_set_freeze: 0 off, 1 when the batch is large enough to gain (FREEZE_AUTO_MIN_QUERIES), 2 always
    FreezeParams fz_prm{8.0f, 2.0e-5f, 3.0e-4f, 3};
    int fz_from = 5;                // launch index of the first launch that may be a freeze launch
    bool fz_from_auto = true;       // learnt from the last fetched alignment (sf_icp_fetch_results); sf_icp_set_freeze_params fixes it
    int fz_fetches = 0;             // fetched alignments since fz_from was last reset to its default (every FZ_PROBE_EVERY-th starts over)
    int fz_step = 0;                // stepping paths: launches since the pass began
    sf::DevBuf fz_state, fz_part, fz_cnt, fz_ids, fz_all;
    int64_t nn_stats_used = 0;
    static constexpr int64_t NN_STATS_CAP = 1024;
    // tile search (sf_tile.hpp): the searching launches of large batches
    int tile_mode = 0;              // sf_icp_set_tile_search: 0 off, 1 when the batch is large enough to gain (TILE_AUTO_MIN_QUERIES), 2 whenever possible
    bool tile_on = false;           // this alignment's queries are sorted by tile and its searching launches run k_tile_search
    sf::SfTiles tiles{};
    sf::DevBuf qtkey, tseg, tile_stats; // tile key of every ordered query; per (tile, scan) segment starts; counters
};

namespace {

float *soa(sf::DevBuf &b, int64_t total, int axis) { return b.as<float>() + (size_t)axis * (size_t)total; }
// the query arrays this alignment walks: the cell-ordered copy or the scans as given
const float *src(sf_icp *icp, int axis)
{
    if (icp->shard) return soa(icp->Xq, icp->own_total, axis); // compact, cell-ordered owned queries (shard_build)
    return soa(icp->ordered ? icp->Xq : icp->X0, icp->plane, axis);
}

int fused_capacity(sf_icp *icp, int mode); // (workgroups the single-launch kernels keep resident; defined with them)

// AUTO orders when there is enough work for the order to pay for the sort.  Measured, 200 k-point
// scans, 20 iterations: 1 scan in flight +1.5 % (break-even), 2: +8 %, 4: +32 %, 32: +65 %
constexpr int64_t ORDER_AUTO_MIN_QUERIES = 300000;

// key = position of the query's cell in the walk order (order_cell), shifted down to the 20 bits the sort takes
int order_key_shift(const SfGrid &g)
{
    const uint64_t ny_pad = ((uint64_t)g.dim[1] + ORDER_YBLK - 1) / ORDER_YBLK * ORDER_YBLK; // order_cell pads y to whole blocks
    const uint64_t ncell = (uint64_t)g.dim[0] * ny_pad * (uint64_t)g.dim[2];
    int cbits = 0;
    while (cbits < 63 && (1ull << cbits) < ncell) ++cbits;
    return std::max(0, cbits - sf::ORD_KEY_BITS);
}

// neighbour reuse starts empty at every alignment (a zero bound certifies nothing)
int reuse_reset(sf_icp *icp, int64_t count)
{
    if (!icp->reuse && !icp->tile_on) return SF_OK; // (tile launches hand their pairs to k_red_cached through the cache arrays)
    const size_t c = (size_t)std::max<int64_t>(count, 1);
    // nothing is cleared: a scan's entries count only once IcpState::cache_live says they have been written (every lane
    // writes its entry in the first launch after a start or a rebuild of the owned arrays)
    SF_TRY(icp->qcache.reserve(sizeof(float4) * 2 * c));
    icp->cache_n = (int64_t)c;
    return SF_OK;
}

// ---- tile search: which alignments, which tiles
// The searching launches of a batch go tile by tile when there is enough work to amortise staging every tile that holds a
// query and the second sorting pass (measured break-even: TILE_AUTO_MIN_QUERIES), on the whole map (no window), unsharded,
// with the queries ordered.  Tile shape: cells per axis such that the staged region (core + 2 cells all round) holds about
// 3/4 of TILE_PCAP points at the map's mean density -- 8 x 8 x 4 cells at the metric configuration (0.25 m cells, 1.5 points
// per cell: 12 x 12 x 8 = 1 152 staged cells, ~1 700 points) -- within the limits of the LDS tables.
constexpr int64_t TILE_AUTO_MIN_QUERIES = 2000000;
bool tile_plan(const SfGrid &g, sf::SfTiles *out)
{
    const double ncell = (double)g.dim[0] * (double)g.dim[1] * (double)g.dim[2];
    if (g.n <= 0 || ncell <= 0) return false;
    const double density = (double)g.n / ncell; // points per cell (empty space included: an upper-bound-free mean; denser tiles fall back one by one)
    int tc[3] = {8, 8, 4};
    auto region_cells = [&](const int *c) { return (double)(c[0] + 2 * sf::TILE_HALO) * (c[1] + 2 * sf::TILE_HALO) * (c[2] + 2 * sf::TILE_HALO); };
    auto fits_tables = [&](const int *c) { return c[0] + 2 * sf::TILE_HALO <= sf::TILE_RX_MAX && (c[1] + 2 * sf::TILE_HALO) * (c[2] + 2 * sf::TILE_HALO) <= sf::TILE_ROWS_MAX; };
    const double budget = 0.75 * sf::TILE_PCAP;
    while (region_cells(tc) * density > budget) { // shrink the longest axis
        int a = tc[0] >= tc[1] && tc[0] >= tc[2] ? 0 : (tc[1] >= tc[2] ? 1 : 2);
        if (tc[a] <= 2) return false; // too dense for any tile: the global index
        tc[a] /= 2;
    }
    for (;;) { // grow the shortest axis while there is room (sparse maps)
        int a = tc[2] <= tc[1] && tc[2] <= tc[0] ? 2 : (tc[1] <= tc[0] ? 1 : 0);
        int t2[3] = {tc[0], tc[1], tc[2]};
        t2[a] *= 2;
        if (!fits_tables(t2) || region_cells(t2) * density > budget) break;
        tc[0] = t2[0]; tc[1] = t2[1]; tc[2] = t2[2];
    }
    if (!fits_tables(tc)) return false;
    sf::SfTiles tl;
    int64_t nt = 1;
    for (int a = 0; a < 3; ++a) {
        tl.tc[a] = tc[a];
        tl.nt[a] = (g.dim[a] + tc[a] - 1) / tc[a];
        nt *= tl.nt[a];
    }
    if (nt >= (int64_t)TILE_KEY_NONE) return false;
    tl.ntiles = (int)nt;
    *out = tl;
    return true;
}

bool tile_wanted(const sf_icp *icp, int mode)
{
    if (icp->tile_mode == 0 || icp->shard || icp->map->window.kind != 0 || (mode != SF_ICP_P2PLANE && mode != SF_ICP_O3D_P2P)) return false;
    if (icp->qpl == 1) return false; // scans the single-launch kernels can take keep their summation order (and their launch list)
    return icp->tile_mode == 2 || icp->n * icp->batch >= TILE_AUTO_MIN_QUERIES;
}

// launches of an alignment that run tile by tile: those in which nearly every query searches -- all of them with the neighbour
// reuse off, the first VERIFY_FROM_SEARCH with it on (afterwards whole waves certify and k_nn_red streams the cache)
bool tile_launch(const sf_icp *icp, int k) { return icp->tile_on && (!icp->reuse || k < VERIFY_FROM_SEARCH); }

// the segmented stable bucket sort of sf_order.hpp: nseg segments (uniform: nseg scans of icp->n queries; sharded: the
// owned-query candidates of each scan, seg_off / src_idx on the device), `longest` = the longest segment
// The key table of the attached index (k_order_lut_*), (re)built when the index has changed.  Enqueued on the context's stream
// BEFORE an alignment takes a lane: the lanes order themselves behind the context's stream whenever the target changes
// (LaneScope), so both see the finished table.
constexpr int ORDER_LUT_BINS = (1 << ORDER_LUT_LOG2) + 1;
int order_lut_prepare(sf_icp *icp)
{
    if (!icp->order_lut_on || !icp->map || !icp->map->built) return SF_OK;
    const SfGrid &g = icp->map->grid;
    if (g.n <= 0 || (icp->order_lut.p && icp->order_lut_gen == icp->map->generation)) return SF_OK;
    const uint64_t ny_pad = ((uint64_t)g.dim[1] + ORDER_YBLK - 1) / ORDER_YBLK * ORDER_YBLK;
    const uint64_t span = (uint64_t)g.dim[0] * ny_pad * (uint64_t)g.dim[2];
    int cbits = 0;
    while (cbits < 63 && (1ull << cbits) < span) ++cbits;
    icp->order_lut_shift = std::max(0, cbits - ORDER_LUT_LOG2);
    const int nbins = (int)(span >> icp->order_lut_shift) + 1;
    SF_TRY(icp->order_lut.reserve(sizeof(uint32_t) * (size_t)ORDER_LUT_BINS + sizeof(uint16_t) * (size_t)ORDER_LUT_BINS));
    uint32_t *hist = icp->order_lut.as<uint32_t>();
    uint16_t *lut = reinterpret_cast<uint16_t *>(hist + ORDER_LUT_BINS);
    hipStream_t st = icp->ctx->stream;
    SF_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * (size_t)ORDER_LUT_BINS, st));
    hipLaunchKernelGGL(k_order_lut_hist, dim3(nblk(g.n)), dim3(256), 0, st, g, icp->order_lut_shift, hist);
    hipLaunchKernelGGL(k_order_lut_make, dim3(1), dim3(1024), 0, st, hist, nbins, (uint32_t)g.n, lut);
    SF_HIP(hipGetLastError());
    icp->order_lut_gen = icp->map->generation;
    return SF_OK;
}

int run_order_sort(sf_icp *icp, int nseg, int64_t longest, int64_t total, const uint32_t *seg_off, const uint32_t *src_idx)
{
    const SfGrid &g = icp->map->grid;
    const int64_t all = icp->n * icp->batch;
    if (seg_off) icp->tile_on = false; // (the sharded path's owned-query arrays keep the cell order)
    sf::OrderSrc src;
    src.src_idx = src_idx;
    src.seg_off = seg_off;
    src.n = (int)icp->n;
    src.tiles = (int)std::max<int64_t>(1, sf::div_up(longest, sf::ORD_TILE));
    CellKeyFn kf;
    kf.g = g;
    kf.x = soa(icp->X0, icp->plane, 0); kf.y = soa(icp->X0, icp->plane, 1); kf.z = soa(icp->X0, icp->plane, 2);
    (void)all;
    kf.st = icp->state.as<IcpState>();
    kf.shift = order_key_shift(g);
    kf.lut = nullptr;
    kf.lut_shift = 0;
    if (icp->order_lut_on && icp->order_lut.p && icp->order_lut_gen == icp->map->generation) { // (prepared by order_lut_prepare; otherwise the plain key)
        kf.lut = reinterpret_cast<const uint16_t *>(icp->order_lut.as<uint32_t>() + ORDER_LUT_BINS);
        kf.lut_shift = icp->order_lut_shift;
    }
    const size_t cap = (size_t)std::max<int64_t>(total, 1);
    SF_TRY(icp->Xq.reserve(sizeof(float) * 3 * std::max(cap, (size_t)(seg_off ? 0 : icp->plane))));
    SF_TRY(icp->qkeys.reserve(sizeof(uint16_t) * cap));                                                      // bucket key of every element
    SF_TRY(icp->qkeys2.reserve(sizeof(uint32_t) * (size_t)nseg * (size_t)(src.tiles + 1) * sf::ORD_BINS));  // per-tile bucket counts / starts
    SF_TRY(icp->qidx.reserve(sizeof(uint32_t) * cap));                                                       // ordered query ids
    uint16_t *keys = icp->qkeys.as<uint16_t>();
    uint32_t *counts = icp->qkeys2.as<uint32_t>(), *ordered = icp->qidx.as<uint32_t>();
    hipStream_t s = icp->ctx->stream;
    src.nseg = nseg;
    const dim3 grid(sf::order_grid(src.tiles, nseg)), blk(sf::ORD_BLK);
    if (icp->tile_on) { // by map tile: a 20-bit key, least significant 10-bit digit first (stable passes)
        TileKeyFn tk;
        tk.g = g; tk.tl = icp->tiles;
        tk.x = kf.x; tk.y = kf.y; tk.z = kf.z;
        tk.st = kf.st; tk.n = (int)icp->n;
        const bool two = icp->tiles.ntiles > sf::ORD_BINS;
        uint32_t *first = ordered;
        if (two) {
            SF_TRY(icp->qidx2.reserve(sizeof(uint32_t) * cap));
            first = icp->qidx2.as<uint32_t>();
        }
        tk.digit = 0;
        hipLaunchKernelGGL(k_order_hist_tile, grid, blk, 0, s, src, tk, keys, counts);
        hipLaunchKernelGGL(sf::k_order_scan, dim3((unsigned)nseg), dim3(sf::ORD_BINS), 0, s, counts, src.tiles);
        hipLaunchKernelGGL(k_order_scatter, grid, blk, 0, s, src, keys, counts, first);
        if (two) {
            sf::OrderSrc src2 = src;
            src2.src_idx = first;
            tk.digit = 1;
            hipLaunchKernelGGL(k_order_hist_tile, grid, blk, 0, s, src2, tk, keys, counts);
            hipLaunchKernelGGL(sf::k_order_scan, dim3((unsigned)nseg), dim3(sf::ORD_BINS), 0, s, counts, src2.tiles);
            hipLaunchKernelGGL(k_order_scatter, grid, blk, 0, s, src2, keys, counts, ordered);
        }
    } else {
        hipLaunchKernelGGL(k_order_hist, grid, blk, 0, s, src, kf, keys, counts);
        hipLaunchKernelGGL(sf::k_order_scan, dim3((unsigned)nseg), dim3(sf::ORD_BINS), 0, s, counts, src.tiles);
        hipLaunchKernelGGL(k_order_scatter, grid, blk, 0, s, src, keys, counts, ordered);
    }
    const int64_t qplane = seg_off ? total : icp->plane; // sharded: the compact arrays have their own length
    if (seg_off) {
        hipLaunchKernelGGL(k_order_gather, dim3(nblk(total, GATHER_TILE)), dim3(256), 0, s, icp->X0r.as<float4>(), ordered, total, 0, 0, 0, soa(icp->Xq, qplane, 0),
                           soa(icp->Xq, qplane, 1), soa(icp->Xq, qplane, 2));
    } else {
        const int gt = (int)std::max<int64_t>(1, sf::div_up(icp->n, GATHER_TILE));
        hipLaunchKernelGGL(k_order_gather, dim3(sf::order_grid(gt, nseg)), dim3(256), 0, s, icp->X0r.as<float4>(), ordered, total, (int)icp->n, gt, nseg,
                           soa(icp->Xq, qplane, 0), soa(icp->Xq, qplane, 1), soa(icp->Xq, qplane, 2));
    }
    if (icp->tile_on) { // where each tile's queries start in every scan's ordered array
        TileKeyFn tk;
        tk.g = g; tk.tl = icp->tiles;
        tk.x = soa(icp->Xq, qplane, 0); tk.y = soa(icp->Xq, qplane, 1); tk.z = soa(icp->Xq, qplane, 2);
        tk.st = kf.st; tk.n = (int)icp->n; tk.digit = 0;
        const int64_t rows = (int64_t)icp->tiles.ntiles + 1;
        SF_TRY(icp->qtkey.reserve(sizeof(uint32_t) * cap));
        SF_TRY(icp->tseg.reserve(sizeof(uint32_t) * (size_t)rows * (size_t)nseg));
        SF_TRY(icp->tile_stats.reserve(sizeof(unsigned long long) * TILE_STATS));
        SF_HIP(hipMemsetAsync(icp->tile_stats.p, 0, sizeof(unsigned long long) * TILE_STATS, s));
        hipLaunchKernelGGL(k_tile_keys, dim3(nblk(total)), dim3(256), 0, s, tk, total, icp->qtkey.as<uint32_t>());
        hipLaunchKernelGGL(k_tile_starts, dim3(nblk(rows * nseg)), dim3(256), 0, s, icp->qtkey.as<uint32_t>(), (int)icp->n, nseg, icp->tiles.ntiles, icp->tseg.as<uint32_t>());
    }
    SF_HIP(hipGetLastError());
    return SF_OK;
}

int order_queries(sf_icp *icp, int mode)
{
    const int64_t total = icp->n * icp->batch;
    const bool want = icp->order == SF_ORDER_CELL || (icp->order == SF_ORDER_AUTO && total >= ORDER_AUTO_MIN_QUERIES);
    icp->ordered = false;
    icp->tile_on = false;
    if (!want || total == 0 || icp->map->grid.n == 0 || icp->n_on_device) return SF_OK; // (a count left on the device: the tail of the arrays is not data)
    icp->tile_on = tile_wanted(icp, mode) && tile_plan(icp->map->grid, &icp->tiles);
    SF_TRY(run_order_sort(icp, icp->batch, icp->n, total, nullptr, nullptr));
    icp->ordered = true;
    return SF_OK;
}

void raw_swap(sf::DevBuf &a, sf::DevBuf &b) // the allocations change places, epochs included (a captured graph keeps pointing at ITS lane's buffers)
{
    std::swap(a.p, b.p);
    std::swap(a.cap, b.cap);
    std::swap(a.epoch, b.epoch);
}

void src_flip(sf_icp *icp)
{
    sf_icp::SrcSet &o = icp->other_src;
    raw_swap(icp->X0, o.X0); raw_swap(icp->X0r, o.X0r); raw_swap(icp->stage, o.stage); raw_swap(icp->d_boxes, o.d_boxes); raw_swap(icp->d_box_parts, o.d_box_parts);
    raw_swap(icp->n_dev, o.n_dev);
    std::swap(icp->n, o.n); std::swap(icp->n_cap, o.n_cap); std::swap(icp->plane, o.plane);
    std::swap(icp->batch, o.batch); std::swap(icp->nblocks, o.nblocks); std::swap(icp->nblocks_nn, o.nblocks_nn); std::swap(icp->qpl, o.qpl);
    std::swap(icp->have_source, o.have_source); std::swap(icp->n_on_device, o.n_on_device);
    icp->src_set ^= 1;
}

// the lanes' streams and every event of the pipeline, once
int ensure_lanes(sf_icp *icp)
{
    for (int l = 0; l < 2; ++l) {
        if (!icp->lane_stream[l]) SF_HIP(hipStreamCreateWithFlags(&icp->lane_stream[l], hipStreamNonBlocking));
        if (!icp->lane_done[l]) SF_HIP(hipEventCreateWithFlags(&icp->lane_done[l], hipEventDisableTiming));
        if (!icp->src_used[l]) SF_HIP(hipEventCreateWithFlags(&icp->src_used[l], hipEventDisableTiming));
        if (!icp->started[l]) SF_HIP(hipEventCreateWithFlags(&icp->started[l], hipEventDisableTiming));
    }
    if (!icp->main_mark) SF_HIP(hipEventCreateWithFlags(&icp->main_mark, hipEventDisableTiming));
    if (!icp->src_ready) SF_HIP(hipEventCreateWithFlags(&icp->src_ready, hipEventDisableTiming));
    if (!icp->unmarked_ev) SF_HIP(hipEventCreateWithFlags(&icp->unmarked_ev, hipEventDisableTiming));
    return SF_OK;
}

// RAII around a call that writes the source.  While an alignment of this object is unfetched (and the lanes are on) the new
// source goes into the OTHER source set, on the stream of the lane the next alignment will take: it waits for the last
// alignment that read that set, not for the alignment in flight.  Otherwise: the set at hand on the context's stream, as before.
struct SrcScope {
    sf_icp *icp;
    hipStream_t main = nullptr;
    bool ahead = false;
    uint64_t version0 = 0;
    int rc = SF_OK;
    SrcScope(sf_icp *i, bool allowed) : icp(i)
    {
        main = icp->ctx->stream;
        version0 = icp->src_version;
        static const bool ahead_on = []() { const char *e = std::getenv("SF_SRC_AHEAD"); return !e || std::atoi(e) != 0; }(); // (A/B switch)
        const bool go = ahead_on && allowed && icp->pipeline != 0 && icp->unfetched && !icp->shard && !icp->profiling;
        if (!go) {
            // the set at hand on the context's stream: behind an upload that went to a lane's stream, if there was one
            if (icp->src_ahead) {
                if (hipStreamWaitEvent(main, icp->src_ready, 0) != hipSuccess) { rc = SF_ERR_HIP; return; }
                icp->src_ahead = false;
            }
            return;
        }
        rc = ensure_lanes(icp);
        if (rc != SF_OK) return;
        if (!icp->src_ahead) src_flip(icp); // (a second source before any alignment: the same set, the same stream, again)
        hipStream_t ls = icp->lane_stream[icp->lane ^ 1];
        if (icp->src_unmarked_use) { // readers that left no event: everything the context's stream holds
            if (hipEventRecord(icp->unmarked_ev, main) != hipSuccess || hipStreamWaitEvent(ls, icp->unmarked_ev, 0) != hipSuccess) { rc = SF_ERR_HIP; return; }
            icp->src_unmarked_use = false;
        }
        if (icp->src_used_rec[icp->src_set] && hipStreamWaitEvent(ls, icp->src_used[icp->src_set], 0) != hipSuccess) { rc = SF_ERR_HIP; return; }
        static const bool stagger_on = []() { const char *e = std::getenv("SF_UPLOAD_STAGGER"); return !e || std::atoi(e) != 0; }(); // (A/B switch)
        if (stagger_on && icp->started_rec[icp->lane] && hipStreamWaitEvent(ls, icp->started[icp->lane], 0) != hipSuccess) { rc = SF_ERR_HIP; return; } // see sf_icp::started
        icp->ctx->stream = ls;
        ahead = true;
    }
    ~SrcScope()
    {
        hipStream_t ran = icp->ctx->stream;
        icp->ctx->stream = main;
        if (!ahead) return;
        if (hipEventRecord(icp->src_ready, ran) == hipSuccess) icp->src_ahead = true;
        // (this source did not pass the context's stream: the mark of the inputs that did stands as it stood)
        if (icp->mark_valid && icp->mark_src_version == version0) icp->mark_src_version = icp->src_version;
    }
};

// outputs = false (a source written ahead of an alignment in flight): the lane's output buffers are left to lane_reserve at the next enqueue
int icp_alloc(sf_icp *icp, int64_t n, int batch, bool outputs = true)
{
    const int64_t total = n * batch;
    // plane = distance between the x, y and z components of the SoA arrays: rounded up and never shrinking, so that the
    // component pointers a captured launch list holds survive scans of slightly different sizes
    icp->plane = std::max<int64_t>(icp->plane, sf::div_up(std::max<int64_t>(total, 1), 4096) * 4096);
    SF_TRY(icp->X0.reserve(sizeof(float) * 3 * (size_t)icp->plane));
    SF_TRY(icp->X0r.reserve(sizeof(float4) * (size_t)icp->plane));
    if (outputs) {
        const auto e0 = icp->state.epoch;
        SF_TRY(icp->state.reserve(sizeof(IcpState) * (size_t)batch));
        if (icp->state.epoch != e0) icp->meta.valid = false; // (the states of this lane's last alignment went with the old allocation: sf_icp_fetch_previous has nothing to read)
        SF_TRY(icp->d_inits.reserve(sizeof(double) * 16 * (size_t)batch));
    }
    // single scan: workgroups for the point count rounded up to 4096 (the kernels bound themselves by the count in
    // device memory), so the launch geometry -- and with it a captured graph -- is shared by scans of similar size
    icp->n_cap = batch == 1 ? sf::div_up(std::max<int64_t>(n, 1), 4096) * 4096 : n;
    icp->nblocks = (int)std::max<int64_t>(1, sf::div_up(icp->n_cap, BLK));
    // Two queries per lane are part of the summation order, so the choice must not depend on which path runs an alignment: above
    // wide_from always; between WIDE_AUTO_POINTS and wide_from when no single-launch kernel could take the batch anyway (more
    // rows than any of them keeps resident) -- a 64-ring sensor's 130 k points in a batch then run wide and may freeze.
    bool wide = n > icp->wide_from;
    if (!wide && icp->wide_auto && n > WIDE_AUTO_POINTS) {
        const int64_t cap = std::max(fused_capacity(icp, SF_ICP_O3D_P2P), fused_capacity(icp, SF_ICP_P2PLANE));
        wide = sf::div_up(n, BLK) * (int64_t)batch > cap;
    }
    icp->qpl = wide ? SF_WIDE_QPL : 1;
    icp->nblocks_nn = (int)std::max<int64_t>(1, sf::div_up(n, BLK * icp->qpl));
    if (outputs) {
        SF_TRY(icp->partials.reserve(sizeof(double) * (size_t)REC_STRIDE * (size_t)icp->nblocks * (size_t)batch));
        SF_TRY(icp->xchg_own.reserve(sizeof(double) * REC_STRIDE * (size_t)batch));
    }
    if (icp->inits.size() != (size_t)batch * 16) { // (by the priors' own size, not by the batch of whichever source set was at hand)
        icp->inits.assign((size_t)batch * 16, 0.0);
        for (int b = 0; b < batch; ++b)
            for (int d = 0; d < 4; ++d) icp->inits[(size_t)b * 16 + 5 * d] = 1.0;
    }
    icp->n = n;
    icp->batch = batch;
    icp->h_state.resize((size_t)batch);
    return SF_OK;
}

int icp_set_source_device_aos(sf_icp *icp, const float *d_aos, int64_t n, int batch, bool ahead = false)
{
    SF_TRY(icp_alloc(icp, n, batch, !ahead));
    icp->n_on_device = false;
    const int64_t total = n * batch;
    if (total > 0)
        hipLaunchKernelGGL(k_soa_from_aos, dim3(nblk(total)), dim3(256), 0, icp->ctx->stream, d_aos, total, soa(icp->X0, icp->plane, 0), soa(icp->X0, icp->plane, 1),
                           soa(icp->X0, icp->plane, 2), icp->X0r.as<float4>());
    SF_TRY(icp->n_dev.reserve(sizeof(int)));
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, icp->ctx->stream, icp->n_dev.as<int>(), (int)n);
    SF_HIP(hipGetLastError());
    // bounding box of every scan on its own, reduced on the device and left there (no host synchronisation per scan): what moves
    // a scan's points is bounded by ITS box (a batch of 10 m scans spread over a 100 m map has a 100 m box and every rotation a
    // 50 m lever arm).  (The box of the whole batch, which rounds 1-3 also formed here, had no reader left.)
    SF_TRY(icp->d_boxes.reserve(sizeof(ScanBox) * (size_t)std::max(batch, 1)));
    SF_TRY(icp->d_box_parts.reserve(sizeof(ScanBox) * BOX_PARTS * (size_t)std::max(batch, 1)));
    hipLaunchKernelGGL(k_scan_boxes_partial, dim3(BOX_PARTS, (unsigned)std::max(batch, 1)), dim3(256), 0, icp->ctx->stream, soa(icp->X0, icp->plane, 0), soa(icp->X0, icp->plane, 1),
                       soa(icp->X0, icp->plane, 2), (int)n, icp->d_box_parts.as<ScanBox>());
    hipLaunchKernelGGL(k_scan_boxes_final, dim3(nblk(std::max(batch, 1), 64)), dim3(64), 0, icp->ctx->stream, icp->d_box_parts.as<ScanBox>(), std::max(batch, 1), icp->d_boxes.as<ScanBox>());
    icp->have_source = true;
    icp->src_version += 1;
    return SF_OK;
}

// HIP events around a phase of the alignment on the context's stream (profiling only).  kind 0 = the NN kernel (the
// per-launch list of sf_icp_profile_read_launches); the sharded path also times its other phases (sf_icp_profile_read_phases)
struct ProfScope {
    sf_icp *icp;
    explicit ProfScope(sf_icp *i, int kind = SF_PROF_NN) : icp(i)
    {
        if (!icp->profiling) return;
        while (icp->ev.size() < icp->ev_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { icp->profiling = false; return; }
            icp->ev.push_back(e);
        }
        if (icp->ev_kind.size() < icp->ev_used / 2 + 1) icp->ev_kind.resize(icp->ev_used / 2 + 1);
        icp->ev_kind[icp->ev_used / 2] = kind;
        hipError_t e = hipEventRecord(icp->ev[icp->ev_used], icp->ctx->stream);
        (void)e;
    }
    ~ProfScope()
    {
        if (!icp->profiling) return;
        hipError_t e = hipEventRecord(icp->ev[icp->ev_used + 1], icp->ctx->stream);
        (void)e;
        icp->ev_used += 2;
    }
};

void prof_collect(sf_icp *icp)
{
    for (size_t k = 0; k + 1 < icp->ev_used; k += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, icp->ev[k], icp->ev[k + 1]) != hipSuccess) continue;
        const int kind = k / 2 < icp->ev_kind.size() ? icp->ev_kind[k / 2] : SF_PROF_NN;
        if (kind == SF_PROF_NN) { icp->prof_ms += ms; icp->prof_launches += 1; icp->prof_each.push_back(ms); }
        else if (kind > 0 && kind < SF_PROF_KINDS) icp->prof_phase[kind].push_back(ms);
    }
    icp->ev_used = 0;
}

sf_icp::GraphKey graph_key_now(const sf_icp *icp, int mode)
{
    sf_icp::GraphKey k;
    k.mode = mode; k.iters = icp->prm.num_iters; k.batch = icp->batch; k.window = icp->map->window.kind; k.ordered = (int)icp->ordered | ((int)icp->tile_on << 1); k.reuse = (int)icp->reuse | (icp->freeze << 1) | (icp->fz_from << 3);
    k.n = (mode == SF_ICP_REF_CPP && icp->batch == 1) ? -icp->n_cap : icp->n; // REF_CPP, one scan: any count of the same capacity replays
    k.map = (const void *)icp->map;
    k.map_generation = icp->map->generation;
    k.max_corr = icp->prm.max_corr; k.accept = icp->prm.accept; k.eps = icp->prm.eps + icp->fz_prm.guard_scale * 1.0e-3f + icp->fz_prm.guard_min + icp->fz_prm.guard_max + (float)icp->fz_prm.max_tries; // (the freeze parameters travel by value too)
    const sf::DevBuf *bufs[] = {&icp->X0, &icp->X0r, &icp->X, &icp->Xq, &icp->qcache, &icp->corr, &icp->state, &icp->partials, &icp->d_box, &icp->d_boxes, &icp->n_dev,
                                &icp->map->pts4, &icp->map->nrm4, &icp->map->cell_start, &icp->map->d_window, &icp->fz_state, &icp->fz_part, &icp->fz_cnt, &icp->fz_ids, &icp->fz_all, &icp->tseg, &icp->tile_stats};
    k.epochs = (uint64_t)icp->plane;
    for (const sf::DevBuf *b : bufs) k.epochs = (k.epochs * 1000003ull + b->epoch) * 1000003ull + (uint64_t)(uintptr_t)b->p; // (the address too: the two source sets take turns under one lane's graph)
    return k;
}

float o3d_thr(const sf_icp *icp) { return (float)((double)icp->prm.max_corr * (double)icp->prm.max_corr); }

// one_per_lane: a wide scan's launch with ONE query per lane (rows of 256 queries: icp->nblocks of them) -- the launches in
// which nearly every query searches, where the second query per lane only costs registers (enqueue_align)
template <int MODE>
void launch_nn_red(sf_icp *icp, bool sharded = false, bool one_per_lane = false)
{
    sf_map *m = icp->map;
    const int nb = sharded ? (one_per_lane ? icp->own_nblocks * icp->qpl : icp->own_nblocks) : (one_per_lane ? icp->nblocks : icp->nblocks_nn);
    const dim3 grid((unsigned)((nb + 7) & ~7), (unsigned)icp->batch), blk(BLK); // see the XCD mapping in k_nn_red
    const float *x = src(icp, 0), *y = src(icp, 1), *z = src(icp, 2);
    const IcpState *st = icp->state.as<IcpState>();
    double *part = icp->partials.as<double>();
    const float thr = o3d_thr(icp);
    hipStream_t s = icp->ctx->stream;
    ProfScope ps(icp);
    uint32_t *stats = nullptr;
    if (icp->profiling && icp->nn_stats.p && icp->nn_stats_used < sf_icp::NN_STATS_CAP) stats = icp->nn_stats.as<uint32_t>() + 2 * NN_STATS_SHARDS * icp->nn_stats_used++;
    const bool win = m->window.kind != 0;
#define SF_LAUNCH_NNRED_Q(W, S, QQ)                                                                                                                              \
    hipLaunchKernelGGL((k_nn_red<MODE, W, S, QQ>), grid, blk, 0, s, m->grid, m->window, x, y, z, (int)icp->n, st, thr, icp->xlo, icp->xhi, part, nb, \
                       icp->own_off.as<uint32_t>(), icp->reuse ? icp->qcache.as<float4>() : nullptr, icp->cache_n, stats)
#define SF_LAUNCH_NNRED(W, S)                                  \
    do {                                                       \
        if (icp->qpl == 1 || one_per_lane) SF_LAUNCH_NNRED_Q(W, S, 1); \
        else SF_LAUNCH_NNRED_Q(W, S, SF_WIDE_QPL);             \
    } while (0)
    if (win && sharded) SF_LAUNCH_NNRED(true, true);
    else if (win) SF_LAUNCH_NNRED(true, false);
    else if (sharded) SF_LAUNCH_NNRED(false, true);
    else SF_LAUNCH_NNRED(false, false);
#undef SF_LAUNCH_NNRED_Q
#undef SF_LAUNCH_NNRED
}

// a searching launch tile by tile: k_tile_search leaves every query's pair in the neighbour cache, k_red_cached sums them in
// k_nn_red's order (the same records, bit for bit)
template <int MODE>
void launch_tile_search(sf_icp *icp, bool one_per_lane)
{
    sf_map *m = icp->map;
    const int nb = one_per_lane ? icp->nblocks : icp->nblocks_nn;
    const float *x = src(icp, 0), *y = src(icp, 1), *z = src(icp, 2);
    const IcpState *st = icp->state.as<IcpState>();
    const float thr = o3d_thr(icp);
    hipStream_t s = icp->ctx->stream;
    ProfScope ps(icp);
    unsigned long long *stats = icp->profiling ? icp->tile_stats.as<unsigned long long>() : nullptr;
    if (icp->profiling && icp->nn_stats.p && icp->nn_stats_used < sf_icp::NN_STATS_CAP) icp->nn_stats_used++; // (keeps the per-launch statistics slots of k_nn_red aligned with the launch list)
    const unsigned tgrid = (unsigned)((icp->tiles.ntiles + 7) & ~7);
    hipLaunchKernelGGL((k_tile_search<MODE>), dim3(tgrid), dim3(sf::TILE_BLK), 0, s, m->grid, icp->tiles, x, y, z, (int)icp->n, icp->batch, st, thr, icp->tseg.as<uint32_t>(),
                       icp->qcache.as<float4>(), icp->cache_n, (int)icp->reuse, stats);
    const dim3 grid((unsigned)((nb + 7) & ~7), (unsigned)icp->batch), blk(BLK);
    if (icp->qpl == 1 || one_per_lane)
        hipLaunchKernelGGL((k_red_cached<MODE, 1>), grid, blk, 0, s, x, y, z, (int)icp->n, st, thr, icp->partials.as<double>(), nb, icp->qcache.as<float4>(), icp->cache_n);
    else
        hipLaunchKernelGGL((k_red_cached<MODE, SF_WIDE_QPL>), grid, blk, 0, s, x, y, z, (int)icp->n, st, thr, icp->partials.as<double>(), nb, icp->qcache.as<float4>(), icp->cache_n);
}

// frozen pairs: P2PLANE launch list of wide scans with the neighbour reuse on, whole map; sharded: each rank freezes its own
// owned queries (what it contributes to the all-reduced record is the same sum either way)
bool freeze_on(const sf_icp *icp, int mode)
{
    // a frozen launch costs one wave's chain of round trips (16 us) + the solve, whatever the batch: for a small batch a verifying
    // launch is cheaper than that (200 k-point scans, ms per alignment without / with: one scan 0.39 / 0.45, four 0.77 / 0.75,
    // eight 1.07 / 0.97, sixteen 1.68 / 1.42)
    const int64_t queries = icp->shard ? icp->own_total : icp->n * icp->batch;
    const bool wanted = icp->freeze == 2 || (icp->freeze == 1 && queries >= FREEZE_AUTO_MIN_QUERIES);
    return mode == SF_ICP_P2PLANE && wanted && icp->reuse && icp->qpl == SF_WIDE_QPL && icp->map->window.kind == 0 && icp->prm.num_iters > icp->fz_from + 1 &&
           icp->fz_from >= VERIFY_FROM_SEARCH;
}

int freeze_alloc(sf_icp *icp)
{
    const size_t rows = (size_t)icp->batch * (size_t)std::max(icp->nblocks_nn, icp->shard ? icp->own_nblocks : 0);
    SF_TRY(icp->fz_state.reserve(sizeof(FreezeState) * (size_t)icp->batch));
    SF_TRY(icp->fz_part.reserve(sizeof(double) * FZ_NMOM * rows));
    SF_TRY(icp->fz_cnt.reserve(sizeof(uint32_t) * rows));
    SF_TRY(icp->fz_ids.reserve(sizeof(uint16_t) * FZ_CAP * rows));
    SF_TRY(icp->fz_all.reserve(sizeof(uint32_t) * FZ_CAP * rows));
    return SF_OK;
}

FreezeBufs freeze_bufs(sf_icp *icp, bool on)
{
    FreezeBufs fb;
    fb.fz = on ? icp->fz_state.as<FreezeState>() : nullptr;
    fb.mom_part = icp->fz_part.as<double>();
    fb.act_cnt = icp->fz_cnt.as<uint32_t>();
    fb.act_ids = icp->fz_ids.as<uint16_t>();
    fb.act_all = icp->fz_all.as<uint32_t>();
    return fb;
}

// few: FZ_FEW workgroups per scan (k_nn_red_fz_few: the launches after the first chance to freeze)
void launch_nn_red_fz(sf_icp *icp, bool sharded, bool few)
{
    sf_map *m = icp->map;
    const int nb = sharded ? icp->own_nblocks : icp->nblocks_nn;
    const dim3 grid_full((unsigned)((nb + 7) & ~7), (unsigned)icp->batch), grid_few((unsigned)FZ_FEW, (unsigned)icp->batch), blk(BLK);
    ProfScope ps(icp);
    FzArgs A;
    A.X0x = src(icp, 0); A.X0y = src(icp, 1); A.X0z = src(icp, 2);
    A.n = (int)icp->n;
    A.st = icp->state.as<IcpState>();
    A.thr = o3d_thr(icp); A.xlo = icp->xlo; A.xhi = icp->xhi;
    A.partials = icp->partials.as<double>();
    A.nblocks = nb;
    A.own_off = icp->own_off.as<uint32_t>();
    A.qcache = icp->qcache.as<float4>();
    A.cache_n = icp->cache_n;
    A.stats = nullptr;
    if (icp->profiling && icp->nn_stats.p && icp->nn_stats_used < sf_icp::NN_STATS_CAP) A.stats = icp->nn_stats.as<uint32_t>() + 2 * NN_STATS_SHARDS * icp->nn_stats_used++;
    A.fz = icp->fz_state.as<FreezeState>();
    A.mom_part = icp->fz_part.as<double>();
    A.act_cnt = icp->fz_cnt.as<uint32_t>();
    A.act_ids = icp->fz_ids.as<uint16_t>();
    A.act_all = icp->fz_all.as<uint32_t>();
    hipStream_t s = icp->ctx->stream;
    if (few) {
        if (sharded) hipLaunchKernelGGL((k_nn_red_fz_few<SF_WIDE_QPL, true>), grid_few, blk, 0, s, m->grid, m->window, A);
        else hipLaunchKernelGGL((k_nn_red_fz_few<SF_WIDE_QPL, false>), grid_few, blk, 0, s, m->grid, m->window, A);
    } else {
        if (sharded) hipLaunchKernelGGL((k_nn_red_fz<SF_WIDE_QPL, true>), grid_full, blk, 0, s, m->grid, m->window, A);
        else hipLaunchKernelGGL((k_nn_red_fz<SF_WIDE_QPL, false>), grid_full, blk, 0, s, m->grid, m->window, A);
    }
}

// the stepping paths (sharded loop, sf_icp_step_begin / _end): which launch of the pass this is
bool freeze_nn_now(const sf_icp *icp, int mode) { return freeze_on(icp, mode) && icp->fz_step >= icp->fz_from; }
bool freeze_solve_now(const sf_icp *icp, int mode) { return freeze_on(icp, mode) && icp->fz_step + 1 >= icp->fz_from; }
int freeze_start_pass(sf_icp *icp, int mode)
{
    icp->fz_step = 0;
    if (!freeze_on(icp, mode)) return SF_OK;
    SF_TRY(freeze_alloc(icp));
    hipLaunchKernelGGL(k_fz_init, dim3(nblk(icp->batch, 64)), dim3(64), 0, icp->ctx->stream, icp->fz_state.as<FreezeState>(), icp->batch);
    return SF_OK;
}

// A hipMemcpyAsync from pageable memory makes the host wait until the stream has reached it -- that would turn every
// "async" alignment into a host synchronisation (measured: it kept a second stream's upload from overlapping).  Small
// batches travel in the kernel arguments; larger ones through a pinned staging buffer, and only when they changed.
int launch_state_init(sf_icp *icp)
{
    hipStream_t s = icp->ctx->stream;
    const int B = icp->batch;
    InitArgs args;
    if (B <= INIT_ARGS_MAX) {
        std::memcpy(args.T, icp->inits.data(), sizeof(double) * 16 * (size_t)B);
        hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, s, icp->state.as<IcpState>(), (const double *)nullptr, args, B);
        return SF_OK;
    }
    const size_t bytes = sizeof(double) * 16 * (size_t)B;
    if (icp->inits_uploaded != icp->inits || icp->d_inits_epoch != icp->d_inits.epoch) {
        if (icp->h_inits_cap < bytes) {
            if (icp->inits_ev_pending) { SF_HIP(hipEventSynchronize(icp->inits_ev)); icp->inits_ev_pending = false; }
            if (icp->h_inits) { hipError_t e = hipHostFree(icp->h_inits); (void)e; icp->h_inits = nullptr; }
            SF_HIP(hipHostMalloc((void **)&icp->h_inits, bytes * 2, hipHostMallocDefault));
            icp->h_inits_cap = bytes * 2;
        }
        if (!icp->inits_ev) SF_HIP(hipEventCreateWithFlags(&icp->inits_ev, hipEventDisableTiming));
        if (icp->inits_ev_pending) SF_HIP(hipEventSynchronize(icp->inits_ev)); // the previous copy has left the staging buffer
        std::memcpy(icp->h_inits, icp->inits.data(), bytes);
        SF_HIP(hipMemcpyAsync(icp->d_inits.p, icp->h_inits, bytes, hipMemcpyHostToDevice, s));
        SF_HIP(hipEventRecord(icp->inits_ev, s));
        icp->inits_ev_pending = true;
        icp->inits_uploaded = icp->inits;
        icp->d_inits_epoch = icp->d_inits.epoch;
    }
    hipLaunchKernelGGL(k_state_init, dim3(nblk(B, 64)), dim3(64), 0, s, icp->state.as<IcpState>(), icp->d_inits.as<double>(), args, B);
    return SF_OK;
}

// enqueue the whole alignment (no host synchronisation)
int enqueue_align(sf_icp *icp, int mode)
{
    sf_map *m = icp->map;
    hipStream_t s = icp->ctx->stream;
    IcpState *st = icp->state.as<IcpState>();
    double *part = icp->partials.as<double>();
    const int K = icp->prm.num_iters;
    const int B = icp->batch;
    const int n = (int)icp->n;
    if (mode == SF_ICP_O3D_P2P) {
        for (int k = 0; k <= K; ++k) {
            if (tile_launch(icp, k)) launch_tile_search<1>(icp, false);
            else launch_nn_red<1>(icp);
            hipLaunchKernelGGL(k_reduce_solve<1>, dim3(B), dim3(RBLK), 0, s, st, part, icp->nblocks_nn, n, k, K, icp->d_boxes.as<ScanBox>());
        }
    } else if (mode == SF_ICP_P2PLANE) {
        const bool fz = freeze_on(icp, mode);
        if (fz) { // (buffers: freeze_alloc, before any capture)
            hipLaunchKernelGGL(k_fz_init, dim3(nblk(B, 64)), dim3(64), 0, s, icp->fz_state.as<FreezeState>(), B);
        }
        // wide scans: the first launches -- nearly every query searches -- run one query per lane (fewer registers: measured
        // 929, 804, 508, 414 against 954, 836, 539, 439 us); from the launch in which waves start to certify whole, two
        const bool wide = icp->qpl > 1 && !icp->shard;
        for (int k = 0; k < K; ++k) {
            const bool q1 = wide && k < VERIFY_FROM_SEARCH; // (the same schedule with the reuse off: it is part of the summation order, and reuse on == off bit for bit)
            if (fz && k >= icp->fz_from) launch_nn_red_fz(icp, false, k > icp->fz_from);
            else if (tile_launch(icp, k)) launch_tile_search<2>(icp, q1);
            else launch_nn_red<2>(icp, false, q1);
            if (fz && k + 1 >= icp->fz_from) // (after a one-query-per-lane launch too: an ordinary record of twice as many rows, and the solve may ask for the freeze launch)
                hipLaunchKernelGGL(k_reduce_solve_fz, dim3(B), dim3(RBLK), 0, s, st, part, q1 ? icp->nblocks : icp->nblocks_nn, n, K, icp->d_boxes.as<ScanBox>(), freeze_bufs(icp, true),
                                   icp->fz_prm, (int)(k + 2 < K));
            else if (q1)
                hipLaunchKernelGGL(k_reduce_solve<2>, dim3(B), dim3(RBLK), 0, s, st, part, icp->nblocks, n, k, K, icp->d_boxes.as<ScanBox>());
            else
                hipLaunchKernelGGL(k_reduce_solve<2>, dim3(B), dim3(RBLK), 0, s, st, part, icp->nblocks_nn, n, k, K, icp->d_boxes.as<ScanBox>());
        }
    } else {
        SF_TRY(icp->X.reserve(sizeof(float) * 3 * (size_t)std::max<int64_t>(icp->plane, 1)));
        SF_TRY(icp->corr.reserve(sizeof(float4) * (size_t)std::max<int64_t>(icp->plane, 1)));
        float *Xx = soa(icp->X, icp->plane, 0), *Xy = soa(icp->X, icp->plane, 1), *Xz = soa(icp->X, icp->plane, 2);
        float4 *corr = icp->corr.as<float4>();
        const dim3 gpts((unsigned)icp->nblocks, (unsigned)B), gred((unsigned)icp->nblocks, (unsigned)B);
        const int *nl = B == 1 ? icp->n_dev.as<int>() : nullptr; // single scan: the count comes from the device (see k_ref_init)
        const bool win = m->window.kind != 0;
        const float thr = icp->prm.max_corr; // squared-vs-unsquared quirk, icp_point_to_point.cpp:70
        const dim3 gnn((unsigned)((icp->nblocks + 7) & ~7), (unsigned)B); // see the XCD mapping in k_ref_nn
        auto nn = [&](int force) {
            ProfScope ps(icp);
            if (win) hipLaunchKernelGGL(k_ref_nn<true>, gnn, dim3(BLK), 0, s, m->grid, m->d_window.as<SfWindow>(), Xx, Xy, Xz, n, nl, st, thr, force, corr, icp->nblocks);
            else hipLaunchKernelGGL(k_ref_nn<false>, gnn, dim3(BLK), 0, s, m->grid, (const SfWindow *)nullptr, Xx, Xy, Xz, n, nl, st, thr, force, corr, icp->nblocks);
        };
        auto red = [&](int apply, int only_research) {
            hipLaunchKernelGGL(k_ref_red, gred, dim3(BLK), 0, s, Xx, Xy, Xz, n, nl, st, corr, apply, only_research, part, icp->nblocks);
        };
        auto decide = [&](int phase) { hipLaunchKernelGGL(k_ref_decide, dim3(B), dim3(RBLK), 0, s, st, part, icp->nblocks, icp->prm, phase); };
        hipLaunchKernelGGL(k_ref_init, gpts, dim3(BLK), 0, s, src(icp, 0), src(icp, 1), src(icp, 2), n, nl, st, Xx, Xy, Xz, corr); // the cell-ordered copy when ordering is on
        nn(1);
        red(0, 0);
        decide(0);
        for (int i = 0; i < K; ++i) {
            decide(1);
            nn(0);
            red(0, 1);
            decide(2);
            if (i + 1 < K) red(1, 0);
        }
    }
    SF_HIP(hipGetLastError());
    return SF_OK;
}

// The single-launch form needs every workgroup of the grid resident at once (its grid barrier waits for all of them).
// The occupancy query can read one workgroup per CU high on this part (MI355X_MICROARCH.md, residency), so one is
// taken off and the rest capped at 4 per CU; larger alignments take the launch list.
template <class... KERNELS>
int resident_workgroups(sf_icp *icp, KERNELS... kernels)
{
    int per_cu = 1 << 20;
    bool ok = true;
    auto ask = [&](auto kernel) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, BLK, 0) != hipSuccess) ok = false;
        per_cu = std::min(per_cu, v);
    };
    (ask(kernels), ...);
    hipDeviceProp_t prop;
    if (!ok || hipGetDeviceProperties(&prop, icp->ctx->device) != hipSuccess) return 0;
    return std::max(std::min(per_cu - 1, 4), 0) * prop.multiProcessorCount;
}

int fused_capacity(sf_icp *icp, int mode)
{
    int &slot = icp->fused_limit[mode];
    if (slot >= 0) return slot;
    if (mode == SF_ICP_REF_CPP) slot = resident_workgroups(icp, k_ref_fused<true>, k_ref_fused<false>);
    else if (mode == SF_ICP_O3D_P2P)
        slot = resident_workgroups(icp, k_icp_fused<1, true, true>, k_icp_fused<1, false, true>, k_icp_fused<1, true, false>, k_icp_fused<1, false, false>);
    else
        slot = resident_workgroups(icp, k_icp_fused<2, true, true>, k_icp_fused<2, false, true>, k_icp_fused<2, true, false>, k_icp_fused<2, false, false>);
    return slot;
}

// single-launch grids in flight per device, as shares of what the device can hold resident (all contexts of this process)
struct FusedLedger {
    std::mutex mu;
    double used[64] = {};
} g_fused_ledger;

void fused_release(sf_icp *icp)
{
    if (icp->fused_share <= 0.0) return;
    std::lock_guard<std::mutex> lk(g_fused_ledger.mu);
    double &u = g_fused_ledger.used[icp->ctx->device & 63];
    u = std::max(0.0, u - icp->fused_share);
    icp->fused_share = 0.0;
}

bool fused_reserve(sf_icp *icp, double share)
{
    std::lock_guard<std::mutex> lk(g_fused_ledger.mu);
    double &u = g_fused_ledger.used[icp->ctx->device & 63];
    if (u + share > 1.0 + 1e-9) return false;
    u += share;
    icp->fused_share = share;
    return true;
}

bool fused_eligible(sf_icp *icp, int mode)
{
    const int64_t rows = mode == SF_ICP_REF_CPP ? icp->nblocks : icp->nblocks_nn;
    // the single-launch kernels keep one point per lane (rows of 256): wide scans (qpl > 1) are not theirs
    const int64_t cap = fused_capacity(icp, mode);
    if (!(icp->fused && !icp->profiling && !icp->shard && (mode == SF_ICP_REF_CPP || icp->qpl == 1) && cap > 0 && rows * icp->batch <= cap)) return false;
    fused_release(icp); // the previous alignment of this object has been fetched or superseded
    return fused_reserve(icp, (double)(rows * icp->batch) / (double)cap);
}

template <int MODE>
void launch_icp_fused(sf_icp *icp, dim3 grid)
{
    sf_map *m = icp->map;
    hipStream_t s = icp->ctx->stream;
    const float thr = o3d_thr(icp);
    const bool win = m->window.kind != 0;
#define SF_LAUNCH_ICPF(W, R)                                                                                                                                               \
    hipLaunchKernelGGL((k_icp_fused<MODE, W, R>), grid, dim3(BLK), 0, s, m->grid, m->window, src(icp, 0), src(icp, 1), src(icp, 2), (int)icp->n, icp->state.as<IcpState>(), thr, \
                       icp->prm.num_iters, icp->partials.as<double>(), icp->nblocks_nn, icp->bar.as<uint32_t>(), icp->h_pin)
    if (win && icp->reuse) SF_LAUNCH_ICPF(true, true);
    else if (win) SF_LAUNCH_ICPF(true, false);
    else if (icp->reuse) SF_LAUNCH_ICPF(false, true);
    else SF_LAUNCH_ICPF(false, false);
#undef SF_LAUNCH_ICPF
}

int launch_fused(sf_icp *icp, int mode)
{
    sf_map *m = icp->map;
    hipStream_t s = icp->ctx->stream;
    const int B = icp->batch;
    const size_t need = sizeof(uint32_t) * 2 * (size_t)B;
    if (icp->bar.cap < need) {
        SF_TRY(icp->bar.reserve(need));
        SF_HIP(hipMemsetAsync(icp->bar.p, 0, icp->bar.cap, s)); // afterwards the kernel leaves the counters at zero itself
    }
    if (icp->h_pin_cap < (size_t)B) {
        if (icp->h_pin) { hipError_t e = hipHostFree(icp->h_pin); (void)e; icp->h_pin = nullptr; icp->h_pin_cap = 0; }
        SF_HIP(hipHostMalloc((void **)&icp->h_pin, sizeof(IcpState) * (size_t)B, hipHostMallocDefault));
        icp->h_pin_cap = (size_t)B;
    }
    // two slabs per scan (see k_ref_fused)
    SF_TRY(icp->partials.reserve(sizeof(double) * (size_t)REC_STRIDE * (size_t)std::max(icp->nblocks, icp->nblocks_nn) * (size_t)B * 2));
    if (mode != SF_ICP_REF_CPP) {
        const dim3 grid_nn((unsigned)icp->nblocks_nn, (unsigned)B);
        if (mode == SF_ICP_O3D_P2P) launch_icp_fused<1>(icp, grid_nn);
        else launch_icp_fused<2>(icp, grid_nn);
        SF_HIP(hipGetLastError());
        icp->fused_launches += 1;
        return SF_OK;
    }
    const dim3 grid((unsigned)icp->nblocks, (unsigned)B);
    const float thr = icp->prm.max_corr; // squared-vs-unsquared quirk, icp_point_to_point.cpp:70
    IcpParams prm = icp->prm;
    const int *nl = icp->n_on_device ? icp->n_dev.as<int>() : nullptr;
    if (m->window.kind != 0)
        hipLaunchKernelGGL(k_ref_fused<true>, grid, dim3(BLK), 0, s, m->grid, m->window, src(icp, 0), src(icp, 1), src(icp, 2), (int)icp->n, nl, icp->state.as<IcpState>(), prm, thr,
                           icp->partials.as<double>(), icp->nblocks, icp->bar.as<uint32_t>(), icp->h_pin);
    else
        hipLaunchKernelGGL(k_ref_fused<false>, grid, dim3(BLK), 0, s, m->grid, m->window, src(icp, 0), src(icp, 1), src(icp, 2), (int)icp->n, nl, icp->state.as<IcpState>(), prm, thr,
                           icp->partials.as<double>(), icp->nblocks, icp->bar.as<uint32_t>(), icp->h_pin);
    SF_HIP(hipGetLastError());
    icp->fused_launches += 1;
    return SF_OK;
}

// a grid barrier of the single-launch form gave up (another process holding the CUs its workgroups needed): the counters
// are put back, this object takes the launch list from now on, and the alignment is redone that way -- the initial
// transforms and the source are untouched, so the caller gets the result it asked for, late
int redo_after_barrier_timeout(sf_icp *icp);
int check_barrier_flags(sf_icp *icp)
{
    if (!icp->last_fused) return SF_OK;
    bool bad = false;
    for (int b = 0; b < icp->batch; ++b) bad = bad || (icp->h_state[(size_t)b].flags & SF_ICP_FLAG_BARRIER_TIMEOUT);
    bad = bad || icp->inject_timeout;
    icp->inject_timeout = false;
    if (!bad) return SF_OK;
    hipError_t e = hipMemsetAsync(icp->bar.p, 0, icp->bar.cap, icp->ctx->stream);
    (void)e;
    for (int &v : icp->fused_limit) v = 0; // no second attempt on this object
    icp->fused_redone += 1;
    return redo_after_barrier_timeout(icp);
}

int check_ready(sf_icp *icp, int mode)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    SF_CHECK(mode >= 0 && mode <= 2, SF_ERR_INVALID, "unknown mode %d", mode);
    SF_CHECK(icp->have_source, SF_ERR_STATE, "no source cloud set");
    SF_CHECK(icp->map && icp->map->built, SF_ERR_STATE, "no target set");
    SF_CHECK(mode != SF_ICP_P2PLANE || icp->map->has_normals, SF_ERR_STATE, "point-to-plane needs map normals (sf_map_estimate_normals)");
    SF_CHECK(!icp->n_on_device || (mode == SF_ICP_REF_CPP && !icp->shard), SF_ERR_STATE, "a source set by sf_icp_set_source_scan serves unsharded REF_CPP alignments only");
    SF_CHECK(icp->prm.num_iters >= 0, SF_ERR_INVALID, "negative iteration count");
    return SF_OK;
}

void fill_result(const sf_icp *icp, int mode, const IcpState &S, const double *init, sf_icp_result *r)
{
    const bool failed = (mode == SF_ICP_REF_CPP) && (S.flags & SF_ICP_FLAG_FEW_CORR) && S.iterations == 0;
    for (int i = 0; i < 16; ++i) {
        r->T64[i] = failed ? init[i] : S.T[i];
        r->T[i] = (float)r->T64[i];
    }
    r->iterations = S.iterations;
    r->n_corr = S.n_corr;
    r->n_research = S.n_research;
    r->flags = S.flags;
    r->fitness = S.fitness;
    r->rmse = S.rmse;
    if (mode == SF_ICP_REF_CPP) {
        // ICPResult defaults (icp_point_to_point.h:28-39) on the < 10 correspondences path
        r->error = failed ? 1e6f : S.last_error;
        r->converged = failed ? 0 : (S.last_error < icp->prm.accept);
    } else {
        r->error = (float)S.rmse;
        r->converged = S.converged;
    }
}

// the states of the last alignment on the host: copied back, or -- single-launch forms -- already written to pinned host
// memory by the kernel itself (measured on the per-scan path: enqueueing the 608-byte copy costs more host time than the
// kernel's stores)
// count: the scans of the alignment whose states are fetched (-1: the current source's -- the stepping / sharded paths)
int states_to_host(sf_icp *icp, int count = -1)
{
    hipStream_t s = icp->ctx->stream;
    const size_t nb = (size_t)(count >= 0 ? count : icp->batch);
    if (icp->h_state.size() < nb) icp->h_state.resize(nb);
    if (icp->last_fused && icp->h_pin) {
        SF_HIP(hipStreamSynchronize(s));
        std::memcpy(icp->h_state.data(), icp->h_pin, sizeof(IcpState) * nb);
        return SF_OK;
    }
    SF_HIP(hipMemcpyAsync(icp->h_state.data(), icp->state.p, sizeof(IcpState) * nb, hipMemcpyDeviceToHost, s));
    SF_HIP(hipStreamSynchronize(s));
    return SF_OK;
}

int fetch_states(sf_icp *icp)
{
    SF_TRY(states_to_host(icp));
    if (icp->profiling) prof_collect(icp);
    return SF_OK;
}

} // namespace

extern "C" int sf_icp_create(sf_ctx *ctx, float max_correspondence_dist, int num_iterations, float acceptable_mean_error, float transformation_epsilon, sf_icp **out)
{
    SF_CHECK(ctx && out, SF_ERR_INVALID, "bad arguments");
    sf_icp *icp = new (std::nothrow) sf_icp();
    SF_CHECK(icp, SF_ERR_NOMEM, "out of host memory");
    icp->ctx = ctx;
    sf::ctx_retain(ctx);
    icp->prm.max_corr = max_correspondence_dist;
    icp->prm.num_iters = num_iterations;
    icp->prm.accept = acceptable_mean_error;
    icp->prm.eps = transformation_epsilon;
    if (const char *e = std::getenv("SF_ORDER_LUT")) icp->order_lut_on = std::atoi(e) != 0;
    *out = icp;
    return SF_OK;
}

extern "C" void sf_icp_destroy(sf_icp *icp)
{
    if (!icp) return;
    hipError_t e = hipStreamSynchronize(icp->ctx->stream);
    (void)e;
    fused_release(icp);
    if (icp->graph_exec) { e = hipGraphExecDestroy(icp->graph_exec); (void)e; }
    if (icp->other.graph_exec) { e = hipGraphExecDestroy(icp->other.graph_exec); (void)e; }
    for (int l = 0; l < 2; ++l) {
        if (icp->lane_stream[l]) { e = hipStreamSynchronize(icp->lane_stream[l]); (void)e; e = hipStreamDestroy(icp->lane_stream[l]); (void)e; }
        if (icp->lane_done[l]) { e = hipEventDestroy(icp->lane_done[l]); (void)e; }
        if (icp->src_used[l]) { e = hipEventDestroy(icp->src_used[l]); (void)e; }
        if (icp->started[l]) { e = hipEventDestroy(icp->started[l]); (void)e; }
    }
    if (icp->src_ready) { e = hipEventDestroy(icp->src_ready); (void)e; }
    if (icp->unmarked_ev) { e = hipEventDestroy(icp->unmarked_ev); (void)e; }
    if (icp->main_mark) { e = hipEventDestroy(icp->main_mark); (void)e; }
    for (hipEvent_t ev : icp->ev) { e = hipEventDestroy(ev); (void)e; }
    if (icp->inits_ev) { e = hipEventDestroy(icp->inits_ev); (void)e; }
    if (icp->h_inits) { e = hipHostFree(icp->h_inits); (void)e; }
    icp->X0.release(); icp->X0r.release(); icp->qcache.release(); icp->X.release(); icp->Xq.release(); icp->qkeys.release(); icp->qkeys2.release(); icp->qidx.release(); icp->qidx2.release(); icp->corr.release(); icp->bar.release(); if (icp->h_pin) { hipError_t eh = hipHostFree(icp->h_pin); (void)eh; icp->h_pin = nullptr; } icp->state.release(); icp->d_inits.release();
    icp->n_dev.release(); icp->nn_stats.release(); icp->d_box.release(); icp->d_boxes.release(); icp->d_box_parts.release(); icp->stage.release(); icp->partials.release(); icp->xchg_own.release(); icp->own_idx.release(); icp->own_blk.release(); icp->own_count.release(); icp->own_off.release(); icp->own_keep.release();
    if (icp->own_map) sf_map_destroy(icp->own_map);
    if (icp->own_cloud) sf_cloud_destroy(icp->own_cloud);
    sf_ctx *ctx = icp->ctx;
    delete icp;
    sf::ctx_release(ctx);
}

extern "C" int sf_icp_set_max_correspondence_dist(sf_icp *icp, float v) { SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL"); icp->prm.max_corr = v; return SF_OK; }
extern "C" int sf_icp_set_num_iterations(sf_icp *icp, int v) { SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL"); icp->prm.num_iters = v; return SF_OK; }
extern "C" int sf_icp_set_transformation_epsilon(sf_icp *icp, float v) { SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL"); icp->prm.eps = v; return SF_OK; }
extern "C" int sf_icp_set_acceptable_mean_error(sf_icp *icp, float v) { SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL"); icp->prm.accept = v; return SF_OK; }
extern "C" int sf_icp_set_debug_mode(sf_icp *icp, int on) { SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL"); icp->debug = on; return SF_OK; }

extern "C" int sf_icp_set_initial_transformation(sf_icp *icp, const float T[16])
{
    SF_CHECK(icp && T, SF_ERR_INVALID, "bad arguments");
    if (icp->inits.size() < 16) icp->inits.assign(16, 0.0);
    for (int i = 0; i < 16; ++i) icp->inits[i] = T[i];
    return SF_OK;
}

extern "C" int sf_icp_set_initial_transformation_f64(sf_icp *icp, const double T[16])
{
    SF_CHECK(icp && T, SF_ERR_INVALID, "bad arguments");
    if (icp->inits.size() < 16) icp->inits.assign(16, 0.0);
    for (int i = 0; i < 16; ++i) icp->inits[i] = T[i];
    return SF_OK;
}

extern "C" int sf_icp_set_initial_batch_f64(sf_icp *icp, const double *inits)
{
    SF_CHECK(icp && icp->batch > 0, SF_ERR_STATE, "set the source batch first");
    icp->inits.assign((size_t)icp->batch * 16, 0.0);
    for (int b = 0; b < icp->batch; ++b)
        for (int i = 0; i < 16; ++i) icp->inits[(size_t)b * 16 + i] = inits ? inits[(size_t)b * 16 + i] : (i % 5 == 0 ? 1.0 : 0.0);
    return SF_OK;
}

extern "C" int sf_icp_set_source_batch(sf_icp *icp, const float *xyz, int64_t n_per_scan, int batch)
{
    SF_CHECK(icp && n_per_scan >= 0 && batch >= 1 && (xyz || n_per_scan == 0), SF_ERR_INVALID, "bad arguments");
    SF_CHECK(n_per_scan * batch < (int64_t)0x7fffffff, SF_ERR_OVERFLOW, "too many source points");
    SF_HIP(hipSetDevice(icp->ctx->device));
    std::vector<double> keep = icp->inits;
    const int64_t total = n_per_scan * batch;
    // host memory has no producer on the device to wait for: with an alignment in flight the upload takes the other source set
    // and the next lane's stream (SrcScope) -- the next batch is uploaded and converted beside the alignment, not behind it
    SrcScope src(icp, true);
    SF_TRY(src.rc);
    // persistent staging buffer, no allocation and no host synchronisation per scan: a copy from pageable host
    // memory is staged by the runtime before hipMemcpyAsync returns (the caller may reuse xyz at once); pinned
    // host memory is read asynchronously, stream-ordered -- then the caller owns the usual lifetime rule
    SF_TRY(icp->stage.reserve(sizeof(float) * 3 * (size_t)std::max<int64_t>(total, 1)));
    if (total > 0) SF_HIP(hipMemcpyAsync(icp->stage.p, xyz, sizeof(float) * 3 * (size_t)total, hipMemcpyHostToDevice, icp->ctx->stream));
    int rc = icp_set_source_device_aos(icp, icp->stage.as<float>(), n_per_scan, batch, src.ahead);
    if (batch == 1 && keep.size() >= 16) icp->inits.assign(keep.begin(), keep.begin() + 16); // setters are order independent
    return rc;
}

extern "C" int sf_icp_set_source_batch_device(sf_icp *icp, const void *d_xyz, int64_t n_per_scan, int batch)
{
    SF_CHECK(icp && n_per_scan >= 0 && batch >= 1 && (d_xyz || n_per_scan == 0), SF_ERR_INVALID, "bad arguments");
    SF_CHECK(n_per_scan * batch < (int64_t)0x7fffffff, SF_ERR_OVERFLOW, "too many source points");
    SF_HIP(hipSetDevice(icp->ctx->device));
    std::vector<double> keep = icp->inits;
    SrcScope src(icp, false); // whatever produced d_xyz is ordered on the context's stream: the conversion stays there, behind it
    SF_TRY(src.rc);
    int rc = icp_set_source_device_aos(icp, reinterpret_cast<const float *>(d_xyz), n_per_scan, batch);
    if (batch == 1 && keep.size() >= 16) icp->inits.assign(keep.begin(), keep.begin() + 16);
    return rc;
}

extern "C" int sf_icp_set_source(sf_icp *icp, const float *xyz, int64_t n) { return sf_icp_set_source_batch(icp, xyz, n, 1); }

// the node's scan preprocessing + setSourcePointCloud in one pass (see k_prep_count): `raw` is left as it is
extern "C" int sf_icp_set_source_scan(sf_icp *icp, sf_cloud *raw, int stride, const float center[3], double radius)
{
    SF_CHECK(icp && raw && center && stride > 0 && radius >= 0, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(raw->ctx == icp->ctx, SF_ERR_INVALID, "cloud and icp live on different contexts");
    SF_CHECK(!icp->shard, SF_ERR_STATE, "not for the sharded path");
    SF_HIP(hipSetDevice(icp->ctx->device));
    const int64_t n_raw = raw->n;
    if (n_raw < stride) stride = 1; // point_cloud_processing.hpp:58-61: a cloud shorter than the step is left untouched
    const int64_t n_cand = sf::div_up(n_raw, stride);
    SF_CHECK(n_cand < ((int64_t)1 << 31) - 4096, SF_ERR_OVERFLOW, "too many points");
    SrcScope src(icp, false); // (the raw cloud is a product of the context's stream)
    SF_TRY(src.rc);
    SF_TRY(icp_alloc(icp, n_cand, 1)); // capacity: every candidate survives
    SF_TRY(icp->n_dev.reserve(sizeof(int)));
    hipStream_t s = icp->ctx->stream;
    if (n_cand == 0) {
        hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, s, icp->n_dev.as<int>(), 0);
    } else {
        const unsigned nb = nblk(n_cand, BLK);
        SF_TRY(icp->own_blk.reserve(sizeof(uint32_t) * (size_t)nb));
        const float r2 = (float)(radius * radius);
        hipLaunchKernelGGL(k_prep_count, dim3(nb), dim3(BLK), 0, s, raw->xyz.as<float>(), n_raw, stride, n_cand, center[0], center[1], center[2], r2, icp->own_blk.as<uint32_t>());
        hipLaunchKernelGGL(k_prep_scatter, dim3(nb), dim3(BLK), 0, s, raw->xyz.as<float>(), n_raw, stride, n_cand, center[0], center[1], center[2], r2, icp->own_blk.as<uint32_t>(),
                           soa(icp->X0, icp->plane, 0), soa(icp->X0, icp->plane, 1), soa(icp->X0, icp->plane, 2), icp->X0r.as<float4>(), icp->n_dev.as<int>());
    }
    SF_HIP(hipGetLastError());
    icp->n_on_device = true;
    icp->have_source = true;
    icp->src_version += 1;
    return SF_OK;
}

// points the last single-scan REF_CPP alignment ran on (after sf_icp_align / sf_icp_fetch_results): what sf_icp_set_source_scan kept
extern "C" int sf_icp_source_count(sf_icp *icp, int64_t *n)
{
    SF_CHECK(icp && n, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(icp->batch == 1 && !icp->h_state.empty() && icp->h_state[0].n_points >= 0, SF_ERR_STATE, "no single-scan REF_CPP alignment has been fetched");
    *n = icp->h_state[0].n_points;
    return SF_OK;
}

extern "C" int sf_icp_set_source_cloud(sf_icp *icp, sf_cloud *cloud)
{
    SF_CHECK(icp && cloud, SF_ERR_INVALID, "bad arguments");
    return sf_icp_set_source_batch_device(icp, cloud->xyz.p, cloud->n, 1);
}

extern "C" int sf_icp_set_target_map(sf_icp *icp, sf_map *map)
{
    SF_CHECK(icp && map, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(map->built, SF_ERR_STATE, "map not built");
    icp->map = map;
    return SF_OK;
}

extern "C" int sf_icp_set_target(sf_icp *icp, const float *xyz, int64_t n)
{
    SF_CHECK(icp && n >= 0 && (xyz || n == 0), SF_ERR_INVALID, "bad arguments");
    if (!icp->own_cloud) SF_TRY(sf_cloud_create(icp->ctx, &icp->own_cloud));
    if (!icp->own_map) SF_TRY(sf_map_create(icp->ctx, &icp->own_map));
    SF_TRY(sf_cloud_upload(icp->own_cloud, xyz, n));
    SF_TRY(sf_map_build(icp->own_map, icp->own_cloud, 0.0f));
    icp->map = icp->own_map;
    return SF_OK;
}

extern "C" int sf_icp_set_query_order(sf_icp *icp, int order)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    SF_CHECK(order >= SF_ORDER_AUTO && order <= SF_ORDER_CELL, SF_ERR_INVALID, "unknown query order %d", order);
    icp->order = order;
    return SF_OK;
}

extern "C" int sf_icp_set_nn_reuse(sf_icp *icp, int on)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    icp->reuse = on != 0;
    if (icp->graph_exec) { hipError_t e = hipGraphExecDestroy(icp->graph_exec); (void)e; icp->graph_exec = nullptr; } // the captured launches carry the cache pointers
    return SF_OK;
}

extern "C" int sf_icp_set_wide_scan_points(sf_icp *icp, int64_t points)
{
    SF_CHECK(icp && points >= 1024 && points <= WIDE_SCAN_POINTS, SF_ERR_INVALID, "the limit must lie in [1 024, 131 072] (what the single-launch kernels can take)");
    icp->wide_from = points; // takes effect with the next source
    icp->wide_auto = false;  // an explicit limit is the whole rule
    return SF_OK;
}

extern "C" int sf_icp_set_pipeline(sf_icp *icp, int on)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    icp->pipeline = on != 0;
    return SF_OK;
}

extern "C" int sf_icp_set_tile_search(sf_icp *icp, int mode)
{
    SF_CHECK(icp && mode >= 0 && mode <= 2, SF_ERR_INVALID, "bad arguments");
    icp->tile_mode = mode;
    return SF_OK;
}

extern "C" int sf_icp_tile_info(sf_icp *icp, int64_t out[12])
{
    SF_CHECK(icp && out, SF_ERR_INVALID, "bad arguments");
    for (int i = 0; i < 12; ++i) out[i] = 0;
    out[0] = icp->tile_on ? 1 : 0;
    if (!icp->tile_on) return SF_OK;
    for (int a = 0; a < 3; ++a) { out[1 + a] = icp->tiles.tc[a]; out[4 + a] = icp->tiles.nt[a]; }
    if (icp->tile_stats.p) {
        SF_HIP(hipSetDevice(icp->ctx->device));
        unsigned long long h[TILE_STATS];
        SF_HIP(hipMemcpyAsync(h, icp->tile_stats.p, sizeof(h), hipMemcpyDeviceToHost, icp->ctx->stream));
        SF_HIP(hipStreamSynchronize(icp->ctx->stream));
        for (int i = 0; i < 5; ++i) out[7 + i] = (int64_t)h[i];
    }
    return SF_OK;
}

extern "C" int sf_icp_set_freeze(sf_icp *icp, int on)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    SF_CHECK(on >= 0 && on <= 2, SF_ERR_INVALID, "sf_icp_set_freeze: 0 (off), 1 (auto) or 2 (always)");
    icp->freeze = on;
    return SF_OK;
}

namespace {
// The launches between fz_from and the one in which the scans do freeze walk every query through the few-workgroup kernels
// (k_nn_red_fz_few: 16 workgroups per scan, ~60 % of the full grid's speed) -- nothing on the metric configuration, where
// every scan freezes at the first chance, but measured on ring scans with a 0.3 m / 1.5 degree prior (tools/city_bench.py):
// scans froze at launch 12 and the six launches before cost more than the frozen ones saved (-8 %).  Scans that follow each
// other converge alike, so the schedule of the NEXT alignment starts asking where this one froze: fz_from = the launch by
// which every scan that froze had frozen, never earlier than the default; an alignment that froze nothing pushes it out of
// reach (freeze_on), and every FZ_PROBE_EVERY-th fetched alignment starts again from the default.
constexpr int FZ_FROM_DEFAULT = 5, FZ_PROBE_EVERY = 32;
void freeze_learn_schedule(sf_icp *icp)
{
    if (!icp->fz_from_auto || icp->last_fused || icp->shard || icp->last_mode != SF_ICP_P2PLANE) return;
    const int K = icp->prm.num_iters;
    if (++icp->fz_fetches >= FZ_PROBE_EVERY) { icp->fz_fetches = 0; icp->fz_from = FZ_FROM_DEFAULT; return; }
    if (!freeze_on(icp, icp->last_mode)) return; // (nothing was tried: nothing learnt)
    int latest = -1, never = 0;
    for (int b = 0; b < icp->batch; ++b) {
        const IcpState &S = icp->h_state[(size_t)b];
        if (S.froze_launch >= 0) latest = std::max(latest, S.froze_launch);
        else if (S.iterations >= K) never += 1;
    }
    if (latest < 0) icp->fz_from = std::max(FZ_FROM_DEFAULT, K); // nothing froze: not worth the slower launches next time
    else if (never * 4 <= icp->batch) icp->fz_from = std::min(std::max(FZ_FROM_DEFAULT, latest), std::max(FZ_FROM_DEFAULT, K - 2));
}
} // namespace

extern "C" int sf_icp_set_freeze_params(sf_icp *icp, float guard_scale, float guard_min, float guard_max, int max_tries, int from_launch)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    SF_CHECK(guard_scale >= 0.0f && guard_min >= 0.0f && guard_max >= guard_min && max_tries >= 0 && from_launch >= VERIFY_FROM_SEARCH && from_launch < 1024, SF_ERR_INVALID,
             "bad freeze parameters");
    icp->fz_prm = FreezeParams{guard_scale, guard_min, guard_max, max_tries};
    icp->fz_from = from_launch;
    icp->fz_from_auto = false;
    return SF_OK;
}

extern "C" int sf_icp_freeze_stats(sf_icp *icp, int64_t out[5])
{
    SF_CHECK(icp && out, SF_ERR_INVALID, "bad arguments");
    for (int i = 0; i < 5; ++i) out[i] = 0;
    if (!freeze_on(icp, icp->last_mode) || icp->last_fused || !icp->fz_state.p || icp->batch <= 0) return SF_OK;
    std::vector<FreezeState> h((size_t)icp->batch);
    SF_HIP(hipMemcpyAsync(h.data(), icp->fz_state.p, sizeof(FreezeState) * h.size(), hipMemcpyDeviceToHost, icp->ctx->stream));
    SF_HIP(hipStreamSynchronize(icp->ctx->stream));
    for (const FreezeState &f : h) {
        out[0] += f.froze;             // freeze launches that held
        out[1] += f.thawed;            // scans that moved beyond their guard afterwards
        out[2] += f.tries - f.thawed;  // freeze launches that did not hold (a row's active list overflowed)
        out[3] += f.n_active;          // active queries of the last freeze launch
        out[4] += f.mode == 2 ? 1 : 0; // scans frozen when the alignment ended
    }
    return SF_OK;
}

extern "C" int sf_icp_graph_counts(sf_icp *icp, int64_t *captures, int64_t *launches)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    if (captures) *captures = icp->graph_captures;
    if (launches) *launches = icp->graph_replays;
    return SF_OK;
}

// REF_CPP alignments whose workgroups are all resident at once run as ONE launch (k_ref_fused) unless this is switched
// off; launches (optional) <- how many alignments have taken that form
extern "C" int sf_icp_set_fused(sf_icp *icp, int on)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    icp->fused = on != 0;
    return SF_OK;
}

#ifdef SF_PHASE_TRACE
extern "C" int sf_icp_tile_trace(unsigned long long *out)
{
    unsigned long long h[sf::PH_SHARDS * sf::PH_SLOTS];
    if (hipDeviceSynchronize() != hipSuccess) return SF_ERR_HIP;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_trace), sizeof(h)) != hipSuccess) return SF_ERR_HIP;
    for (int i = 0; i < sf::PH_SLOTS; ++i) { out[i] = 0; for (int s = 0; s < sf::PH_SHARDS; ++s) out[i] += h[s * sf::PH_SLOTS + i]; }
    memset(h, 0, sizeof(h));
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace), h, sizeof(h)) == hipSuccess ? SF_OK : SF_ERR_HIP;
}
// reads and clears the per-phase tick sums of the k_nn_red launches since the last call (diagnostic build only)
extern "C" int sf_icp_phase_trace(unsigned long long *out)
{
    unsigned long long h[sf::PH_SHARDS * sf::PH_SLOTS];
    if (hipDeviceSynchronize() != hipSuccess) return SF_ERR_HIP;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_trace), sizeof(h)) != hipSuccess) return SF_ERR_HIP;
    for (int i = 0; i < sf::PH_SLOTS; ++i) { out[i] = 0; for (int s = 0; s < sf::PH_SHARDS; ++s) out[i] += h[s * sf::PH_SLOTS + i]; }
    memset(h, 0, sizeof(h));
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_trace), h, sizeof(h)) == hipSuccess ? SF_OK : SF_ERR_HIP;
}
#endif
#ifdef SF_FUSED_TRACE
extern "C" int sf_icp_fused_trace(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_trace), sizeof(unsigned long long) * 512) == hipSuccess ? SF_OK : SF_ERR_HIP; }
#endif
extern "C" int sf_icp_fused_count(sf_icp *icp, int64_t *launches)
{
    SF_CHECK(icp && launches, SF_ERR_INVALID, "bad arguments");
    *launches = icp->fused_launches;
    return SF_OK;
}

extern "C" int sf_icp_fused_redone(sf_icp *icp, int64_t *redone)
{
    SF_CHECK(icp && redone, SF_ERR_INVALID, "bad arguments");
    *redone = icp->fused_redone;
    return SF_OK;
}

// test hook: the next single-launch alignment of this object is handled as if one of its grid barriers had given up
extern "C" int sf_icp_test_inject_barrier_timeout(sf_icp *icp)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    icp->inject_timeout = true;
    return SF_OK;
}

extern "C" int sf_icp_use_graph(sf_icp *icp, int on)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    icp->use_graph = on != 0;
    return SF_OK;
}

namespace {
void lane_flip(sf_icp *icp)
{
    sf_icp::Lane &o = icp->other;
    raw_swap(icp->X, o.X); raw_swap(icp->qcache, o.qcache); raw_swap(icp->Xq, o.Xq); raw_swap(icp->qkeys, o.qkeys); raw_swap(icp->qkeys2, o.qkeys2);
    raw_swap(icp->qidx, o.qidx); raw_swap(icp->qidx2, o.qidx2); raw_swap(icp->corr, o.corr); raw_swap(icp->state, o.state); raw_swap(icp->d_inits, o.d_inits);
    raw_swap(icp->partials, o.partials); raw_swap(icp->fz_state, o.fz_state); raw_swap(icp->fz_part, o.fz_part); raw_swap(icp->fz_cnt, o.fz_cnt);
    raw_swap(icp->fz_ids, o.fz_ids); raw_swap(icp->fz_all, o.fz_all); raw_swap(icp->qtkey, o.qtkey); raw_swap(icp->tseg, o.tseg); raw_swap(icp->tile_stats, o.tile_stats);
    std::swap(icp->graph_exec, o.graph_exec);
    std::swap(icp->graph_key, o.graph_key);
    std::swap(icp->inits_uploaded, o.inits_uploaded);
    std::swap(icp->d_inits_epoch, o.d_inits_epoch);
    std::swap(icp->meta, icp->other_meta);
    icp->lane ^= 1;
}

// what icp_alloc sizes for the lane in the members, for the lane that has just been flipped in
int lane_reserve(sf_icp *icp)
{
    const int batch = std::max(icp->batch, 1);
    SF_TRY(icp->state.reserve(sizeof(IcpState) * (size_t)batch));
    SF_TRY(icp->d_inits.reserve(sizeof(double) * 16 * (size_t)batch));
    SF_TRY(icp->partials.reserve(sizeof(double) * (size_t)REC_STRIDE * (size_t)std::max(icp->nblocks, 1) * (size_t)batch));
    return SF_OK;
}

// RAII around the enqueue of one alignment.  An alignment enqueued while an earlier one of this object has not been fetched yet
// (back-to-back sf_icp_align_batch_async: the throughput pattern) takes the OTHER lane's buffers and that lane's stream; the
// first one after a fetch -- and every alignment of a caller that fetches each result before asking for the next -- runs on the
// context's stream with the buffers it finds, as before.  Per lane an event marks its last alignment, whatever stream it ran on.
struct LaneScope {
    sf_icp *icp;
    hipStream_t main = nullptr;
    bool piped = false;
    int rc = SF_OK;
    LaneScope(sf_icp *i, bool allowed) : icp(i)
    {
        main = icp->ctx->stream;
        if (!allowed && !icp->lane_done[0]) { // the per-scan path (single launch) of an object that has never piped: no streams, no events, nothing to record
            icp->src_unmarked_use = true;
            return;
        }
        rc = ensure_lanes(icp); // (every alignment marks the end of its reading of the source set: the events must exist)
        if (rc != SF_OK) return;
        if (!allowed) {
            // the context's stream, the buffers at hand -- behind a source that was written on a lane's stream, if one was
            if (icp->src_ahead) {
                if (hipStreamWaitEvent(main, icp->src_ready, 0) != hipSuccess) { rc = SF_ERR_HIP; return; }
                icp->src_ahead = false;
            }
            return;
        }
        // the mark: where the context's stream stood when the inputs last changed (taken before anything of this alignment is enqueued)
        const sf_map *m = icp->map;
        // (the initial poses are not part of it: they travel in the kernel arguments or through the lane's own copy, on the lane's stream)
        const bool same = icp->mark_valid && icp->mark_src_version == icp->src_version && icp->mark_map == (const void *)m && icp->mark_map_generation == m->generation &&
                          std::memcmp(&icp->mark_window, &m->window, sizeof(SfWindow)) == 0;
        if (!same) {
            if (hipEventRecord(icp->main_mark, main) != hipSuccess) { rc = SF_ERR_HIP; return; }
            icp->mark_valid = true;
            icp->mark_src_version = icp->src_version;
            icp->mark_map = (const void *)m;
            icp->mark_map_generation = m->generation;
            icp->mark_window = m->window;
        }
        if (!icp->unfetched && !icp->src_ahead) { // nothing of this object in flight: the context's stream, the buffers at hand
            if (hipEventRecord(icp->started[icp->lane], main) == hipSuccess) icp->started_rec[icp->lane] = true;
            return;
        }
        lane_flip(icp); // take turns (a source written ahead went to this lane's stream)
        rc = lane_reserve(icp);
        if (rc != SF_OK) return;
        hipStream_t ls = icp->lane_stream[icp->lane];
        // after the inputs, and after this lane's previous alignment (which may have run on the context's stream)
        if (hipStreamWaitEvent(ls, icp->main_mark, 0) != hipSuccess) { rc = SF_ERR_HIP; return; }
        if (icp->lane_used[icp->lane] && hipStreamWaitEvent(ls, icp->lane_done[icp->lane], 0) != hipSuccess) { rc = SF_ERR_HIP; return; }
        if (icp->src_ahead) {
            if (hipStreamWaitEvent(ls, icp->src_ready, 0) != hipSuccess) { rc = SF_ERR_HIP; return; }
            icp->src_ahead = false;
        }
        icp->ctx->stream = ls;
        piped = true;
        if (hipEventRecord(icp->started[icp->lane], ls) == hipSuccess) icp->started_rec[icp->lane] = true;
    }
    ~LaneScope()
    {
        hipStream_t ran = icp->ctx->stream;
        icp->ctx->stream = main;
        if (!icp->lane_done[icp->lane]) return; // (the lanes have never been set up: nothing to mark)
        if (icp->src_used[icp->src_set] && hipEventRecord(icp->src_used[icp->src_set], ran) == hipSuccess) icp->src_used_rec[icp->src_set] = true; // the last reader of this source set
        if (hipEventRecord(icp->lane_done[icp->lane], ran) != hipSuccess) return;
        icp->lane_used[icp->lane] = true;
        // whatever the caller enqueues next on the context's stream is ordered behind this alignment
        if (piped) { hipError_t e = hipStreamWaitEvent(main, icp->lane_done[icp->lane], 0); (void)e; }
    }
};
} // namespace

extern "C" int sf_icp_align_batch_async(sf_icp *icp, int mode)
{
    SF_TRY(check_ready(icp, mode));
    SF_CHECK(!icp->shard, SF_ERR_STATE, "a sharded alignment needs the exchange between its halves: use sf_icp_step_begin / sf_icp_step_end");
    SF_HIP(hipSetDevice(icp->ctx->device));
    icp->last_mode = mode;
    icp->last_fused = fused_eligible(icp, mode);
    SF_TRY(order_lut_prepare(icp));
    // the launch list takes a lane (see sf_icp::Lane); the single launch, profiled runs and a count left on the device stay on the context's stream
    const bool beside = icp->unfetched; // an alignment of this object is still unfetched: this one may run beside it
    LaneScope lanes(icp, icp->pipeline != 0 && !icp->last_fused && !icp->profiling && !icp->n_on_device);
    SF_TRY(lanes.rc);
    if (!lanes.piped) SF_TRY(lane_reserve(icp)); // (a source written ahead of an alignment in flight left the output buffers alone: icp_alloc)
    icp->unfetched = true;
    icp->prev_ok = lanes.piped && beside && icp->other_meta.valid; // (not piped: this alignment runs in the buffers of the one before it)
    icp->meta.valid = true;
    icp->meta.batch = icp->batch;
    icp->meta.mode = mode;
    icp->meta.inits = icp->inits;
    hipStream_t s = icp->ctx->stream;
    SF_TRY(launch_state_init(icp));
    if (icp->last_fused) { // everything resident at once: the whole alignment is one launch (window and count by value)
        SF_TRY(order_queries(icp, mode));
        return launch_fused(icp, mode);
    }
    if (mode == SF_ICP_REF_CPP && icp->map->window.kind != 0) { // the map crop as it stands now, for the kernels that read it from the device
        SF_TRY(icp->map->d_window.reserve(sizeof(SfWindow)));
        hipLaunchKernelGGL(k_set_window, dim3(1), dim3(1), 0, s, icp->map->d_window.as<SfWindow>(), icp->map->window);
    }
    SF_TRY(order_queries(icp, mode)); // plain launches ahead of the (replayed) iteration graph
    if (mode != SF_ICP_REF_CPP) SF_TRY(reuse_reset(icp, icp->n * icp->batch));
    if (freeze_on(icp, mode)) SF_TRY(freeze_alloc(icp));
    // O3D_P2P / P2PLANE take the window by value (it moves with the pose: plain launches then); REF_CPP reads it from device memory
    if (icp->use_graph && !icp->profiling && (icp->map->window.kind == 0 || mode == SF_ICP_REF_CPP)) {
        if (mode == SF_ICP_REF_CPP) { // buffers must exist before capture (and before the key is formed)
            SF_TRY(icp->X.reserve(sizeof(float) * 3 * (size_t)std::max<int64_t>(icp->plane, 1)));
            SF_TRY(icp->corr.reserve(sizeof(float4) * (size_t)std::max<int64_t>(icp->plane, 1)));
        }
        const sf_icp::GraphKey key = graph_key_now(icp, mode);
        const bool hit = icp->graph_exec && key == icp->graph_key;
        if (!hit) {
            if (icp->graph_exec) { hipError_t e = hipGraphExecDestroy(icp->graph_exec); (void)e; icp->graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            SF_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            int rc = enqueue_align(icp, mode);
            hipError_t e = hipStreamEndCapture(s, &graph);
            if (rc != SF_OK) return rc;
            SF_CHECK(e == hipSuccess, SF_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            e = hipGraphInstantiate(&icp->graph_exec, graph, nullptr, nullptr, 0);
            hipError_t e2 = hipGraphDestroy(graph);
            (void)e2;
            SF_CHECK(e == hipSuccess, SF_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
            icp->graph_key = key;
            icp->graph_captures += 1;
        }
        icp->graph_replays += 1;
        SF_HIP(hipGraphLaunch(icp->graph_exec, s));
        return SF_OK;
    }
    return enqueue_align(icp, mode);
}

namespace {
int redo_after_barrier_timeout(sf_icp *icp)
{
    SF_TRY(sf_icp_align_batch_async(icp, icp->last_mode)); // fused_limit is zero now: the launch list
    SF_TRY(states_to_host(icp));
    for (int b = 0; b < icp->batch; ++b)
        SF_CHECK(!(icp->h_state[(size_t)b].flags & SF_ICP_FLAG_BARRIER_TIMEOUT), SF_ERR_HIP, "alignment redone through the launch list still carries a barrier flag");
    return SF_OK;
}
} // namespace

extern "C" int sf_icp_fetch_results(sf_icp *icp, sf_icp_result *out)
{
    SF_CHECK(icp && out, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(icp->batch > 0, SF_ERR_STATE, "nothing to fetch");
    // the latest alignment as it was enqueued: a source or priors set since (for the NEXT alignment) do not describe it
    const bool described = icp->meta.valid && icp->meta.batch > 0 && !icp->shard;
    const int nb = described ? icp->meta.batch : icp->batch;
    SF_TRY(states_to_host(icp, nb));
    icp->unfetched = false;
    fused_release(icp); // the grid has drained
    if (icp->profiling) prof_collect(icp);
    if (nb == icp->batch) SF_TRY(check_barrier_flags(icp));
    for (int b = 0; b < nb; ++b)
        fill_result(icp, described ? icp->meta.mode : icp->last_mode, icp->h_state[(size_t)b], described ? &icp->meta.inits[(size_t)b * 16] : &icp->inits[(size_t)b * 16], out + b);
    if (nb == icp->batch) freeze_learn_schedule(icp);
    return SF_OK;
}

// The alignment enqueued BEFORE the latest one, when the two ran side by side on the lanes (two sf_icp_align_batch_async
// calls with no fetch between them): its states lie in the other lane's buffers until the next piped alignment takes
// them.  Waits for that alignment only -- the latest one goes on running -- so "enqueue the next batch's alignment, fetch
// the previous one's result" delivers every result without draining the device.
extern "C" int sf_icp_fetch_previous(sf_icp *icp, sf_icp_result *out)
{
    SF_CHECK(icp && out, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(icp->prev_ok && icp->other_meta.valid && icp->other_meta.batch > 0, SF_ERR_STATE,
             "no earlier alignment to fetch: the latest alignment did not run beside the one before it (fetched in between, pipeline off, single launch) or it was fetched already");
    const int ol = icp->lane ^ 1;
    const sf_icp::LaneMeta &M = icp->other_meta;
    SF_CHECK(icp->lane_used[ol] && icp->lane_done[ol] && icp->lane_stream[ol] && icp->other.state.p, SF_ERR_STATE, "the other lane holds no alignment");
    SF_HIP(hipSetDevice(icp->ctx->device));
    SF_HIP(hipEventSynchronize(icp->lane_done[ol]));
    icp->h_prev.resize((size_t)M.batch);
    // on the other lane's own stream: neither the context's stream (it waits for the latest alignment) nor the null stream is touched
    SF_HIP(hipMemcpyAsync(icp->h_prev.data(), icp->other.state.p, sizeof(IcpState) * (size_t)M.batch, hipMemcpyDeviceToHost, icp->lane_stream[ol]));
    SF_HIP(hipStreamSynchronize(icp->lane_stream[ol]));
    for (int b = 0; b < M.batch; ++b) fill_result(icp, M.mode, icp->h_prev[(size_t)b], &M.inits[(size_t)b * 16], out + b);
    icp->prev_ok = false;
    return SF_OK;
}

extern "C" int sf_icp_align_batch(sf_icp *icp, int mode, sf_icp_result *out)
{
    SF_CHECK(out, SF_ERR_INVALID, "out is NULL");
    SF_TRY(sf_icp_align_batch_async(icp, mode));
    int rc = sf_icp_fetch_results(icp, out);
    if (rc == SF_OK && icp->debug)
        for (int b = 0; b < icp->batch; ++b)
            fprintf(stdout, "[ICP INFO] scan %d: iterations %d, error %g, correspondences %d, converged %d\n", b, out[b].iterations, (double)out[b].error,
                    out[b].n_corr, out[b].converged);
    return rc;
}

extern "C" int sf_icp_align(sf_icp *icp, int mode, sf_icp_result *out)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    SF_CHECK(icp->have_source, SF_ERR_STATE, "no source cloud set");
    SF_CHECK(icp->batch == 1, SF_ERR_STATE, "sf_icp_align needs a single source scan (use sf_icp_align_batch)");
    return sf_icp_align_batch(icp, mode, out);
}

// ------------------------------------------------------------------ multi-GPU stepping
extern "C" int sf_icp_set_shard(sf_icp *icp, float x_lo, float x_hi)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    icp->shard = x_lo > -INFINITY || x_hi < INFINITY;
    icp->xlo = x_lo;
    icp->xhi = x_hi;
    return SF_OK;
}

extern "C" int sf_icp_set_shard_margin(sf_icp *icp, float margin_m)
{
    SF_CHECK(icp && margin_m > 0.0f && std::isfinite(margin_m), SF_ERR_INVALID, "the margin must be positive");
    icp->own_margin = margin_m;
    return SF_OK;
}

extern "C" int sf_icp_set_exchange_buffer(sf_icp *icp, void *d_buf, int64_t nbytes)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    icp->xchg = d_buf;
    icp->xchg_bytes = d_buf ? nbytes : 0;
    return SF_OK;
}

extern "C" void *sf_icp_exchange_ptr(sf_icp *icp, int64_t *nbytes)
{
    if (!icp) return nullptr;
    const int64_t need = (int64_t)sizeof(double) * REC_STRIDE * std::max(icp->batch, 1);
    if (nbytes) *nbytes = need;
    if (icp->xchg && icp->xchg_bytes >= need) return icp->xchg;
    return icp->xchg_own.p;
}

namespace {

// sharded path, start (or resume) of an alignment: compact this rank's owned-query candidates of
// every running scan, order them by map cell, gather them into the arrays k_nn_red<SHARD> walks.
// One host synchronisation (the counts size the sort and the launch grid).
int shard_build(sf_icp *icp, bool resume)
{
    ProfScope ps(icp, SF_PROF_SHARD_BUILD);
    const int B = icp->batch;
    const int n = (int)icp->n;
    const int nbf = icp->nblocks; // workgroups covering a whole scan
    hipStream_t s = icp->ctx->stream;
    IcpState *st = icp->state.as<IcpState>();
    SF_TRY(icp->own_blk.reserve(sizeof(uint32_t) * (size_t)nbf * (size_t)B));
    SF_TRY(icp->own_count.reserve(sizeof(uint32_t) * (size_t)B));
    SF_TRY(icp->own_off.reserve(sizeof(uint32_t) * (size_t)(B + 1)));
    const float *X = soa(icp->X0, icp->plane, 0), *Y = soa(icp->X0, icp->plane, 1), *Z = soa(icp->X0, icp->plane, 2);
    const dim3 grid((unsigned)nbf, (unsigned)B);
    if (resume) hipLaunchKernelGGL(k_own_resume, dim3(nblk(B, 64)), dim3(64), 0, s, st, B);
    SF_TRY(icp->own_keep.reserve(sizeof(unsigned long long) * (size_t)nbf * (size_t)B * (BLK / 64)));
    hipLaunchKernelGGL(k_own_count, grid, dim3(BLK), 0, s, X, Y, Z, n, st, icp->xlo, icp->xhi, icp->own_margin, icp->own_blk.as<uint32_t>(), nbf, icp->own_keep.as<unsigned long long>());
    hipLaunchKernelGGL(k_own_scan, dim3((unsigned)B), dim3(1024), 0, s, st, icp->own_blk.as<uint32_t>(), nbf, icp->own_count.as<uint32_t>());
    icp->h_own.assign((size_t)B + 1, 0u);
    SF_HIP(hipMemcpyAsync(icp->h_own.data() + 1, icp->own_count.p, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    SF_HIP(hipStreamSynchronize(s));
    uint32_t maxc = 0;
    for (int b = 0; b < B; ++b) {
        maxc = std::max(maxc, icp->h_own[(size_t)b + 1]);
        icp->h_own[(size_t)b + 1] += icp->h_own[(size_t)b]; // counts -> offsets
    }
    const int64_t own = (int64_t)icp->h_own[(size_t)B];
    icp->own_total = own;
    icp->own_nblocks = (int)std::max<int64_t>(1, sf::div_up((int64_t)maxc, BLK * icp->qpl));
    SF_HIP(hipMemcpyAsync(icp->own_off.p, icp->h_own.data(), sizeof(uint32_t) * (size_t)(B + 1), hipMemcpyHostToDevice, s));
    const size_t cap = (size_t)std::max<int64_t>(own, 1);
    SF_TRY(icp->own_idx.reserve(sizeof(uint32_t) * cap));
    SF_TRY(icp->Xq.reserve(sizeof(float) * 3 * cap));
    if (own > 0) {
        hipLaunchKernelGGL(k_own_scatter, grid, dim3(BLK), 0, s, n, st, icp->own_keep.as<unsigned long long>(), icp->own_blk.as<uint32_t>(), nbf, icp->own_off.as<uint32_t>(),
                           icp->own_idx.as<uint32_t>());
        // cell-order every scan's candidates and gather them into the compact arrays (segments = own_off, elements ->
        // global query ids = own_idx); with an empty map every key is equal and the stable sort keeps the compacted order
        SF_TRY(run_order_sort(icp, B, (int64_t)maxc, own, icp->own_off.as<uint32_t>(), icp->own_idx.as<uint32_t>()));
    }
    hipLaunchKernelGGL(k_own_mark, dim3(nblk(B, 64)), dim3(64), 0, s, st, B);
    SF_HIP(hipGetLastError());
    return reuse_reset(icp, own); // the compact indices have changed
}

} // namespace

// first: 1 = start an alignment, 0 = next iteration, 2 = resume the scans that stopped with
// SF_ICP_FLAG_SHARD_STALE (sharded path: their owned-query arrays are rebuilt at the current pose)
extern "C" int sf_icp_step_begin(sf_icp *icp, int mode, int first)
{
    SF_TRY(check_ready(icp, mode));
    SF_CHECK(mode != SF_ICP_REF_CPP, SF_ERR_INVALID, "stepping supports O3D_P2P and P2PLANE");
    SF_CHECK(first >= 0 && first <= 2, SF_ERR_INVALID, "first must be 0, 1 or 2");
    SF_CHECK(first != 2 || icp->shard, SF_ERR_STATE, "resume (first = 2) is a sharded-path operation");
    SF_HIP(hipSetDevice(icp->ctx->device));
    icp->last_mode = mode;
    icp->last_fused = false;
    if (first) SF_TRY(order_lut_prepare(icp));
    if (first == 1) SF_TRY(launch_state_init(icp));
    if (icp->shard) {
        if (first) SF_TRY(shard_build(icp, first == 2));
    } else if (first) {
        SF_TRY(order_queries(icp, mode));
        SF_TRY(reuse_reset(icp, icp->n * icp->batch));
    }
    if (first) SF_TRY(freeze_start_pass(icp, mode));
    double *x = reinterpret_cast<double *>(sf_icp_exchange_ptr(icp, nullptr));
    hipStream_t s = icp->ctx->stream;
    // wide scans: the first launches of a pass with one query per lane, as enqueue_align (rows of 256: twice as many)
    const bool q1 = mode == SF_ICP_P2PLANE && icp->qpl > 1 && icp->fz_step < VERIFY_FROM_SEARCH;
    const int nb = icp->shard ? (q1 ? icp->own_nblocks * icp->qpl : icp->own_nblocks) : (q1 ? icp->nblocks : icp->nblocks_nn);
    const int qpl_now = q1 ? 1 : icp->qpl;
    const uint32_t *off = icp->shard ? icp->own_off.as<uint32_t>() : nullptr;
    if (mode == SF_ICP_O3D_P2P) {
        launch_nn_red<1>(icp, icp->shard);
        ProfScope ps(icp, SF_PROF_REDUCE);
        hipLaunchKernelGGL(k_reduce_only<1>, dim3(icp->batch), dim3(SBLK), 0, s, icp->state.as<IcpState>(), icp->partials.as<double>(), nb, x, off, icp->qpl, freeze_bufs(icp, false));
    } else {
        if (freeze_nn_now(icp, mode)) launch_nn_red_fz(icp, icp->shard, icp->fz_step > icp->fz_from);
        else launch_nn_red<2>(icp, icp->shard, q1);
        ProfScope ps(icp, SF_PROF_REDUCE);
        hipLaunchKernelGGL(k_reduce_only<2>, dim3(icp->batch), dim3(SBLK), 0, s, icp->state.as<IcpState>(), icp->partials.as<double>(), nb, x, off, qpl_now,
                           freeze_bufs(icp, freeze_nn_now(icp, mode)));
    }
    SF_HIP(hipGetLastError());
    return SF_OK;
}

extern "C" int sf_icp_step_end(sf_icp *icp, int mode, int last)
{
    SF_TRY(check_ready(icp, mode));
    (void)last;
    const double *x = reinterpret_cast<const double *>(sf_icp_exchange_ptr(icp, nullptr));
    hipStream_t s = icp->ctx->stream;
    const int K = icp->prm.num_iters;
    ProfScope ps(icp, SF_PROF_SOLVE);
    if (mode == SF_ICP_O3D_P2P)
        hipLaunchKernelGGL(k_solve_only<1>, dim3(nblk(icp->batch, 64)), dim3(64), 0, s, icp->state.as<IcpState>(), x, (int)icp->n, K, icp->batch, icp->d_boxes.as<ScanBox>(),
                           icp->shard ? icp->own_margin : 0.0f, (FreezeState *)nullptr, icp->fz_prm, 0);
    else
        hipLaunchKernelGGL(k_solve_only<2>, dim3(nblk(icp->batch, 64)), dim3(64), 0, s, icp->state.as<IcpState>(), x, (int)icp->n, K, icp->batch, icp->d_boxes.as<ScanBox>(),
                           icp->shard ? icp->own_margin : 0.0f, freeze_solve_now(icp, mode) ? icp->fz_state.as<FreezeState>() : (FreezeState *)nullptr, icp->fz_prm,
                           (int)(icp->fz_step + 2 < K));
    icp->fz_step += 1;
    SF_HIP(hipGetLastError());
    return SF_OK;
}

// every profiled NN launch since sf_icp_profile_enable, in launch order: its duration and -- for the fused
// O3D_P2P / P2PLANE kernel -- how many queries / waves ran the search (the others kept their certified neighbour)
extern "C" int sf_icp_profile_read_launches(sf_icp *icp, float *ms, uint32_t *searched_queries, uint32_t *searched_waves, int64_t cap, int64_t *n)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    SF_HIP(hipStreamSynchronize(icp->ctx->stream));
    prof_collect(icp);
    const int64_t have = (int64_t)icp->prof_each.size();
    if (n) *n = have;
    if (!ms && !searched_queries && !searched_waves) return SF_OK;
    SF_CHECK(cap >= have, SF_ERR_INVALID, "buffer too small: %lld < %lld", (long long)cap, (long long)have);
    const size_t per = (size_t)2 * NN_STATS_SHARDS;
    std::vector<uint32_t> st(per * (size_t)std::max<int64_t>(icp->nn_stats_used, 1), 0u);
    if (icp->nn_stats_used > 0) SF_HIP(hipMemcpy(st.data(), icp->nn_stats.p, sizeof(uint32_t) * per * (size_t)icp->nn_stats_used, hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < have; ++k) {
        if (ms) ms[k] = icp->prof_each[(size_t)k];
        uint64_t q = 0, w = 0;
        if (k < icp->nn_stats_used && icp->last_mode != SF_ICP_REF_CPP)
            for (int sh = 0; sh < NN_STATS_SHARDS; ++sh) { q += st[per * (size_t)k + 2 * sh]; w += st[per * (size_t)k + 2 * sh + 1]; }
        if (searched_queries) searched_queries[k] = (uint32_t)q;
        if (searched_waves) searched_waves[k] = (uint32_t)w;
    }
    return SF_OK;
}

// ------------------------------------------------------------------ sharded alignment driven from the C side
// One rank's whole alignment: per iteration NN + slab reduce (sf_icp_step_begin), the all-reduce of the batch's
// exchange records on the context's stream, the identical solve on every rank (sf_icp_step_end) -- enqueued
// back to back, no host involvement between iterations.  The `sum` callback is the collective: RCCL
// (sf_icp_align_sharded) or, for several slabs held by ONE process on one device, a fixed-order device sum
// (sf_icp_align_group).
namespace {

constexpr int GROUP_MAX = 16;
struct GroupBufs { double *p[GROUP_MAX]; int n; };

// every member's buffer <- sum over the members, in member order (bitwise deterministic)
__global__ void k_group_sum(GroupBufs g, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int m = 0; m < g.n; ++m) s += g.p[m][i];
    for (int m = 0; m < g.n; ++m) g.p[m][i] = s;
}

int sharded_steps(const sf_icp *icp, int mode) { return mode == SF_ICP_O3D_P2P ? icp->prm.num_iters + 1 : icp->prm.num_iters; }

bool any_stale(const sf_icp *icp)
{
    for (int b = 0; b < icp->batch; ++b)
        if (icp->h_state[(size_t)b].flags & SF_ICP_FLAG_SHARD_STALE) return true;
    return false;
}

} // namespace

namespace {
// one iteration of the sharded loop over a P2P communicator: NN + slab rows, reduce + publish, gather + solve
int shard_step_p2p(sf_icp *icp, int mode, int first, const sf::P2pView &view)
{
    SF_HIP(hipSetDevice(icp->ctx->device));
    icp->last_mode = mode;
    icp->last_fused = false;
    if (first) SF_TRY(order_lut_prepare(icp));
    if (first == 1) SF_TRY(launch_state_init(icp));
    if (first) SF_TRY(shard_build(icp, first == 2));
    if (first) SF_TRY(freeze_start_pass(icp, mode));
    hipStream_t s = icp->ctx->stream;
    IcpState *st = icp->state.as<IcpState>();
    const bool q1 = mode == SF_ICP_P2PLANE && icp->qpl > 1 && icp->fz_step < VERIFY_FROM_SEARCH; // as enqueue_align: one query per lane while nearly every query searches
    const int nb = q1 ? icp->own_nblocks * icp->qpl : icp->own_nblocks, B = icp->batch, K = icp->prm.num_iters, qpl_now = q1 ? 1 : icp->qpl;
    const bool fz_nn = freeze_nn_now(icp, mode), fz_solve = freeze_solve_now(icp, mode);
    if (mode == SF_ICP_O3D_P2P) launch_nn_red<1>(icp, true);
    else if (fz_nn) launch_nn_red_fz(icp, true, icp->fz_step > icp->fz_from);
    else launch_nn_red<2>(icp, true, q1);
    {
        ProfScope ps(icp, SF_PROF_REDUCE);
        if (mode == SF_ICP_O3D_P2P) hipLaunchKernelGGL(k_reduce_publish<1>, dim3(B), dim3(SBLK), 0, s, st, icp->partials.as<double>(), nb, icp->own_off.as<uint32_t>(), icp->qpl, view, freeze_bufs(icp, false));
        else hipLaunchKernelGGL(k_reduce_publish<2>, dim3(B), dim3(SBLK), 0, s, st, icp->partials.as<double>(), nb, icp->own_off.as<uint32_t>(), qpl_now, view, freeze_bufs(icp, fz_nn));
    }
    {
        ProfScope ps(icp, SF_PROF_COLLECTIVE); // the wait for the peers' records AND the solve
        if (mode == SF_ICP_O3D_P2P) hipLaunchKernelGGL(k_gather_solve<1>, dim3(B), dim3(64), 0, s, st, (int)icp->n, K, icp->d_boxes.as<ScanBox>(), icp->own_margin, view, (FreezeState *)nullptr, icp->fz_prm, 0);
        else hipLaunchKernelGGL(k_gather_solve<2>, dim3(B), dim3(64), 0, s, st, (int)icp->n, K, icp->d_boxes.as<ScanBox>(), icp->own_margin, view,
                                fz_solve ? icp->fz_state.as<FreezeState>() : (FreezeState *)nullptr, icp->fz_prm, (int)(icp->fz_step + 2 < K));
    }
    icp->fz_step += 1;
    SF_HIP(hipGetLastError());
    return SF_OK;
}
} // namespace

extern "C" int sf_icp_align_sharded_async(sf_icp *icp, int mode, sf_comm *comm, int first)
{
    SF_TRY(check_ready(icp, mode));
    SF_CHECK(comm && sf::comm_ctx(comm) == icp->ctx, SF_ERR_INVALID, "the communicator must live on the icp's context (same stream)");
    SF_CHECK(icp->shard, SF_ERR_STATE, "sf_icp_set_shard first");
    SF_CHECK(first == 1 || first == 2, SF_ERR_INVALID, "first must be 1 (start) or 2 (resume)");
    const int steps = sharded_steps(icp, mode);
    int rc = SF_OK;
    for (int k = 0; k < steps && rc == SF_OK; ++k) {
        sf::P2pView view;
        const int p2p = sf::comm_p2p_begin(comm, (int64_t)REC_STRIDE * icp->batch, &view);
        if (p2p < 0) { rc = p2p; break; }
        if (p2p == 1) { // P2P: the reduce kernel publishes, the solve kernel gathers (two kernels per iteration, no all-reduce of its own)
            rc = shard_step_p2p(icp, mode, k == 0 ? first : 0, view);
            continue;
        }
        rc = sf_icp_step_begin(icp, mode, k == 0 ? first : 0);
        if (rc == SF_OK) {
            ProfScope ps(icp, SF_PROF_COLLECTIVE);
            rc = sf::comm_allreduce_f64(comm, sf_icp_exchange_ptr(icp, nullptr), (int64_t)REC_STRIDE * icp->batch);
        }
        if (rc == SF_OK) rc = sf_icp_step_end(icp, mode, k == steps - 1);
    }
    // this rank stops mid-loop: its peers must not be left waiting in the collectives it will never join
    if (rc != SF_OK && rc != SF_ERR_COMM) {
        const std::string why = sf_last_error();
        sf_comm_abort(comm);
        sf::set_error("%s", why.c_str());
    }
    return rc;
}

// blocking form: passes until no scan is left stale (every rank takes the same decisions: the states are identical)
extern "C" int sf_icp_align_sharded(sf_icp *icp, int mode, sf_comm *comm, sf_icp_result *out, int *resumes)
{
    SF_CHECK(out, SF_ERR_INVALID, "out is NULL");
    int first = 1, n_resume = 0;
    if (icp) {
        const int steps = sharded_steps(icp, mode);
        for (int attempt = 0; attempt <= steps + 1; ++attempt) { // every resume completes at least one iteration
            SF_TRY(sf_icp_align_sharded_async(icp, mode, comm, first));
            SF_TRY(fetch_states(icp));
            SF_TRY(sf_comm_status(comm)); // a collective that timed out or was aborted left the records unsummed: no result
            if (!any_stale(icp)) {
                for (int b = 0; b < icp->batch; ++b) fill_result(icp, icp->last_mode, icp->h_state[(size_t)b], &icp->inits[(size_t)b * 16], out + b);
                if (resumes) *resumes = n_resume;
                return SF_OK;
            }
            first = 2;
            ++n_resume;
        }
    }
    sf::set_error("sharded alignment did not finish");
    return SF_ERR_STATE;
}

// n slabs of one map held by ONE process on one device (members[m] indexes slab m + halo, set_shard(x_lo, x_hi)
// given, the same source batch and initial transforms on every member, all on the same context): the members step
// in lockstep on the context's stream and the "all-reduce" is a device sum of their exchange buffers in member
// order.  Results (identical on every member) are returned from member 0.
extern "C" int sf_icp_align_group(sf_icp **members, int n, int mode, sf_icp_result *out, int *resumes)
{
    SF_CHECK(members && n >= 1 && n <= GROUP_MAX && out, SF_ERR_INVALID, "bad arguments (1..%d members)", GROUP_MAX);
    GroupBufs gb;
    gb.n = n;
    for (int m = 0; m < n; ++m) {
        SF_TRY(check_ready(members[m], mode));
        SF_CHECK(members[m]->shard, SF_ERR_STATE, "member %d: sf_icp_set_shard first", m);
        SF_CHECK(members[m]->ctx == members[0]->ctx && members[m]->batch == members[0]->batch && members[m]->n == members[0]->n &&
                     members[m]->prm.num_iters == members[0]->prm.num_iters,
                 SF_ERR_INVALID, "member %d does not match member 0 (context, batch, points per scan, iterations)", m);
    }
    sf_icp *lead = members[0];
    const int steps = sharded_steps(lead, mode), count = REC_STRIDE * lead->batch;
    hipStream_t s = lead->ctx->stream;
    int first = 1, n_resume = 0;
    for (int attempt = 0; attempt <= steps + 1; ++attempt) {
        for (int k = 0; k < steps; ++k) {
            for (int m = 0; m < n; ++m) {
                SF_TRY(sf_icp_step_begin(members[m], mode, k == 0 ? first : 0));
                gb.p[m] = reinterpret_cast<double *>(sf_icp_exchange_ptr(members[m], nullptr));
            }
            hipLaunchKernelGGL(k_group_sum, dim3(nblk(count, 256)), dim3(256), 0, s, gb, count);
            for (int m = 0; m < n; ++m) SF_TRY(sf_icp_step_end(members[m], mode, k == steps - 1));
        }
        SF_HIP(hipGetLastError());
        for (int m = 0; m < n; ++m) SF_TRY(fetch_states(members[m]));
        for (int m = 1; m < n; ++m) // the slabs must agree bit for bit: same records in, same solve
            for (int b = 0; b < lead->batch; ++b)
                SF_CHECK(std::memcmp(members[m]->h_state[(size_t)b].T, lead->h_state[(size_t)b].T, sizeof(double) * 16) == 0 &&
                             members[m]->h_state[(size_t)b].flags == lead->h_state[(size_t)b].flags,
                         SF_ERR_STATE, "slab %d diverged from slab 0 on scan %d", m, b);
        if (!any_stale(lead)) {
            for (int b = 0; b < lead->batch; ++b) fill_result(lead, lead->last_mode, lead->h_state[(size_t)b], &lead->inits[(size_t)b * 16], out + b);
            if (resumes) *resumes = n_resume;
            return SF_OK;
        }
        first = 2;
        ++n_resume;
    }
    sf::set_error("group alignment did not finish");
    return SF_ERR_STATE;
}

// queries this rank owned in the LAST iteration of the last sharded alignment, per scan (n_corr of the exchange
// record before the collective is not kept; this is the owned-candidate count of the compact arrays)
extern "C" int sf_icp_owned_counts(sf_icp *icp, int64_t *counts, int cap)
{
    SF_CHECK(icp && counts && cap >= icp->batch, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(icp->shard && icp->h_own.size() == (size_t)icp->batch + 1, SF_ERR_STATE, "no sharded alignment has run");
    for (int b = 0; b < icp->batch; ++b) counts[b] = (int64_t)icp->h_own[(size_t)b + 1] - (int64_t)icp->h_own[(size_t)b];
    return SF_OK;
}

// ------------------------------------------------------------------ profiling
extern "C" int sf_icp_profile_enable(sf_icp *icp, int on)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    icp->profiling = on != 0;
    icp->ev_used = 0;
    icp->prof_launches = 0;
    icp->prof_ms = 0;
    icp->prof_each.clear();
    for (auto &v : icp->prof_phase) v.clear();
    icp->nn_stats_used = 0;
    if (on) {
        SF_TRY(icp->nn_stats.reserve(sizeof(uint32_t) * 2 * NN_STATS_SHARDS * sf_icp::NN_STATS_CAP));
        SF_HIP(hipMemsetAsync(icp->nn_stats.p, 0, sizeof(uint32_t) * 2 * NN_STATS_SHARDS * sf_icp::NN_STATS_CAP, icp->ctx->stream));
    }
    return SF_OK;
}

extern "C" int sf_icp_profile_read(sf_icp *icp, int64_t *nn_launches, double *nn_ms_total)
{
    SF_CHECK(icp, SF_ERR_INVALID, "icp is NULL");
    SF_HIP(hipStreamSynchronize(icp->ctx->stream));
    prof_collect(icp);
    if (nn_launches) *nn_launches = icp->prof_launches;
    if (nn_ms_total) *nn_ms_total = icp->prof_ms;
    return SF_OK;
}

// durations [ms] of one phase kind (SF_PROF_REDUCE, SF_PROF_COLLECTIVE, SF_PROF_SOLVE, SF_PROF_SHARD_BUILD) of the sharded
// path since sf_icp_profile_enable, in order; ms may be NULL to ask for the count
extern "C" int sf_icp_profile_read_phases(sf_icp *icp, int kind, float *ms, int64_t cap, int64_t *n)
{
    SF_CHECK(icp && kind > 0 && kind < SF_PROF_KINDS, SF_ERR_INVALID, "bad arguments");
    SF_HIP(hipStreamSynchronize(icp->ctx->stream));
    prof_collect(icp);
    const std::vector<float> &v = icp->prof_phase[kind];
    if (n) *n = (int64_t)v.size();
    if (!ms) return SF_OK;
    SF_CHECK(cap >= (int64_t)v.size(), SF_ERR_INVALID, "buffer too small");
    std::memcpy(ms, v.data(), sizeof(float) * v.size());
    return SF_OK;
}
