// sf_bf.hip — BruteForceAlignment on the device.
//
// Replaces BruteForceAlignment::alignClouds (localization/src/brute_force_alignment.cpp:
// 65-136): the start-up coarse search over 18 x 18 x 4 x 6 = 7 776 candidate poses
// (localization/src/localization_node.cpp:39-43), each scored by the mean SQUARED
// nearest-neighbour distance of every source point (N kd-tree descents per candidate, the most
// expensive thing the reference ever does).  Here one kernel scores a whole slice of
// candidates at once — grid.y = candidate, one lane per source point, the exact grid NN of
// sf_nn.hpp with an unbounded threshold — and the host walks the scores in the reference's
// nesting order so "first candidate under the threshold wins" is preserved exactly.
// Candidate matrices are built on the host in float32 with the reference's operation order,
// so the returned transformation is bit-identical to the reference's T.
#include "sf_common.hpp"
#include "sf_nn.hpp"

#include <cmath>
#include <vector>

struct sf_bf {
    sf_ctx *ctx = nullptr;
    // brute_force_alignment.h:87-112 defaults
    float x_step = 0.1f, y_step = 0.1f, z_step = 0.1f, yaw_step = (float)(M_PI / 90.0f);
    float x_range = 0.5f, y_range = 0.5f, z_range = 0.5f, yaw_range = (float)(M_PI / 6.0f);
    float threshold = 0.1f;
    bool first_alignment_completed = false;
    float previous[16], best[16];
    sf_map *target = nullptr;
    sf_map *own_map = nullptr;
    sf_cloud *own_cloud = nullptr;
    sf::DevBuf src;        // SoA x[n] y[n] z[n]
    int64_t n = 0;
    sf::DevBuf poses, partials, scores;
    std::vector<float> last_scores;
    int last_index = -1;
    float last_score = 0.0f;
};

namespace {

constexpr int BLK = 256;
inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)sf::div_up(n > 0 ? n : 1, b); }

__global__ void k_bf_soa(const float *__restrict__ aos, int64_t n, float *__restrict__ x, float *__restrict__ y, float *__restrict__ z)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] = aos[3 * i]; y[i] = aos[3 * i + 1]; z[i] = aos[3 * i + 2];
}

// every (candidate, source point) squared NN distance: grid (point blocks, candidates); T * Vector4f(p, 1) in float32,
// unfused, column combination with k ascending (brute_force_alignment.cpp:98); NN d2 unbounded (:102).  The search is
// the wave-cooperative one of the ICP kernels (every lane of a wave takes part, the tail included).
template <bool WINDOW>
__global__ __launch_bounds__(BLK, 4) void k_bf_score(SfGrid g, SfWindow w, const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z, int n,
                                                     const float *__restrict__ poses, float *__restrict__ d2_out)
{
    __shared__ sf::WaveNN nn_ws[BLK / 64];
    const int c = blockIdx.y;
    const float *T = poses + (size_t)c * 12;
    const int i = blockIdx.x * BLK + threadIdx.x;
    const bool live = i < n;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (live) {
        const float x = X[i], y = Y[i], z = Z[i];
        qx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], x), __fmul_rn(T[1], y)), __fmul_rn(T[2], z)), T[3]);
        qy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], x), __fmul_rn(T[5], y)), __fmul_rn(T[6], z)), T[7]);
        qz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], x), __fmul_rn(T[9], y)), __fmul_rn(T[10], z)), T[11]);
    }
    const sf::NNHit hit = sf::nn_search_wave<WINDOW, true>(g, w, live, qx, qy, qz, 3.0e38f, &nn_ws[threadIdx.x >> 6]);
    if (live) d2_out[(size_t)c * n + i] = hit.j >= 0 ? hit.d2 : 0.0f;
}

// The reference adds the distances of one candidate SERIALLY in float32, in source order, and divides by N in float32
// (brute_force_alignment.cpp:95-105): a score within float32 summation error of the 0.1 threshold decides the early exit,
// so the sum is reproduced bit for bit -- one wave per candidate loads 64 distances at a time (coalesced) and lane 0's
// accumulator takes them in order through v_readlane.
__global__ __launch_bounds__(64) void k_bf_finish(const float *__restrict__ d2, int n, float *__restrict__ scores)
{
    const int c = blockIdx.x, lane = threadIdx.x;
    const float *row = d2 + (size_t)c * n;
    float acc = 0.0f;
    for (int base = 0; base < n; base += 64) {
        const float v = base + lane < n ? row[base + lane] : 0.0f;
        const int cnt = min(64, n - base);
        if (cnt == 64) {
#pragma unroll
            for (int l = 0; l < 64; ++l) acc = __fadd_rn(acc, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)));
        } else {
            for (int l = 0; l < cnt; ++l) acc = __fadd_rn(acc, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)));
        }
    }
    if (lane == 0) scores[c] = __fdiv_rn(acc, (float)n);
}

// brute_force_alignment.cpp:148-180
std::vector<float> test_sequence(float range, float step)
{
    std::vector<float> seq;
    for (int i = 0; i < range / (2 * step) + 1; ++i) {
        seq.push_back(-i * step);
        seq.push_back(i * step);
    }
    return seq;
}

// float32 coefficient-wise 4x4 product, k ascending
void mul4(const float *A, const float *B, float *C)
{
    float R[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float acc = A[4 * r] * B[c];
            acc = acc + A[4 * r + 1] * B[4 + c];
            acc = acc + A[4 * r + 2] * B[8 + c];
            acc = acc + A[4 * r + 3] * B[12 + c];
            R[4 * r + c] = acc;
        }
    for (int i = 0; i < 16; ++i) C[i] = R[i];
}

} // namespace

extern "C" int sf_bf_create(sf_ctx *ctx, sf_bf **out)
{
    SF_CHECK(ctx && out, SF_ERR_INVALID, "bad arguments");
    sf_bf *bf = new (std::nothrow) sf_bf();
    SF_CHECK(bf, SF_ERR_NOMEM, "out of host memory");
    bf->ctx = ctx;
    sf::ctx_retain(ctx);
    for (int i = 0; i < 16; ++i) bf->previous[i] = bf->best[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    *out = bf;
    return SF_OK;
}

extern "C" void sf_bf_destroy(sf_bf *bf)
{
    if (!bf) return;
    hipError_t e = hipStreamSynchronize(bf->ctx->stream);
    (void)e;
    bf->src.release(); bf->poses.release(); bf->partials.release(); bf->scores.release();
    if (bf->own_map) sf_map_destroy(bf->own_map);
    if (bf->own_cloud) sf_cloud_destroy(bf->own_cloud);
    sf_ctx *ctx = bf->ctx;
    delete bf;
    sf::ctx_release(ctx);
}

extern "C" int sf_bf_set_xyz_step(sf_bf *bf, float x, float y, float z) { SF_CHECK(bf, SF_ERR_INVALID, "bf is NULL"); bf->x_step = x; bf->y_step = y; bf->z_step = z; return SF_OK; }
extern "C" int sf_bf_set_xyz_range(sf_bf *bf, float x, float y, float z) { SF_CHECK(bf, SF_ERR_INVALID, "bf is NULL"); bf->x_range = x; bf->y_range = y; bf->z_range = z; return SF_OK; }
extern "C" int sf_bf_set_rotation_step(sf_bf *bf, float v) { SF_CHECK(bf, SF_ERR_INVALID, "bf is NULL"); bf->yaw_step = v; return SF_OK; }
extern "C" int sf_bf_set_rotation_range(sf_bf *bf, float v) { SF_CHECK(bf, SF_ERR_INVALID, "bf is NULL"); bf->yaw_range = v; return SF_OK; }
extern "C" int sf_bf_set_mean_error_threshold(sf_bf *bf, float v) { SF_CHECK(bf, SF_ERR_INVALID, "bf is NULL"); bf->threshold = v; return SF_OK; }
extern "C" int sf_bf_reset_first_alignment(sf_bf *bf, int value) { SF_CHECK(bf, SF_ERR_INVALID, "bf is NULL"); bf->first_alignment_completed = value != 0; return SF_OK; }
extern "C" int sf_bf_first_alignment_completed(sf_bf *bf) { return bf && bf->first_alignment_completed ? 1 : 0; }

// brute_force_alignment.cpp:44-51: only while no guess has been received (trace == 4)
extern "C" int sf_bf_set_initial_guess(sf_bf *bf, const float T[16])
{
    SF_CHECK(bf && T, SF_ERR_INVALID, "bad arguments");
    const float trace = ((bf->previous[0] + bf->previous[5]) + bf->previous[10]) + bf->previous[15];
    if (trace == 4.0f)
        for (int i = 0; i < 16; ++i) bf->previous[i] = T[i];
    return SF_OK;
}

extern "C" int sf_bf_set_source_cloud(sf_bf *bf, sf_cloud *cloud)
{
    SF_CHECK(bf && cloud, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(cloud->n < (int64_t)0x7fffffff, SF_ERR_OVERFLOW, "too many source points");
    SF_HIP(hipSetDevice(bf->ctx->device));
    const int64_t n = cloud->n;
    SF_TRY(bf->src.reserve(sizeof(float) * 3 * (size_t)std::max<int64_t>(n, 1)));
    if (n > 0)
        hipLaunchKernelGGL(k_bf_soa, dim3(nblk(n)), dim3(256), 0, bf->ctx->stream, cloud->xyz.as<float>(), n, bf->src.as<float>(), bf->src.as<float>() + n,
                           bf->src.as<float>() + 2 * n);
    SF_HIP(hipGetLastError());
    bf->n = n;
    return SF_OK;
}

extern "C" int sf_bf_set_source(sf_bf *bf, const float *xyz, int64_t n)
{
    SF_CHECK(bf && n >= 0 && (xyz || n == 0), SF_ERR_INVALID, "bad arguments");
    sf_cloud *tmp = nullptr;
    SF_TRY(sf_cloud_create(bf->ctx, &tmp));
    int rc = sf_cloud_upload(tmp, xyz, n);
    if (rc == SF_OK) rc = sf_bf_set_source_cloud(bf, tmp);
    sf_cloud_destroy(tmp);
    return rc;
}

extern "C" int sf_bf_set_target_map(sf_bf *bf, sf_map *map)
{
    SF_CHECK(bf && map && map->built, SF_ERR_INVALID, "bad arguments / map not built");
    bf->target = map;
    return SF_OK;
}

extern "C" int sf_bf_set_target(sf_bf *bf, const float *xyz, int64_t n)
{
    SF_CHECK(bf && n >= 0 && (xyz || n == 0), SF_ERR_INVALID, "bad arguments");
    if (!bf->own_cloud) SF_TRY(sf_cloud_create(bf->ctx, &bf->own_cloud));
    if (!bf->own_map) SF_TRY(sf_map_create(bf->ctx, &bf->own_map));
    SF_TRY(sf_cloud_upload(bf->own_cloud, xyz, n));
    SF_TRY(sf_map_build(bf->own_map, bf->own_cloud, 0.0f));
    bf->target = bf->own_map;
    return SF_OK;
}

// brute_force_alignment.cpp:65-136.  *found = 1 when a candidate scored under the threshold.
extern "C" int sf_bf_align_clouds(sf_bf *bf, int *found)
{
    SF_CHECK(bf && found, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(bf->target && bf->target->built, SF_ERR_STATE, "no target set");
    SF_CHECK(bf->n > 0, SF_ERR_STATE, "no source cloud set");
    SF_HIP(hipSetDevice(bf->ctx->device));
    hipStream_t st = bf->ctx->stream;
    const std::vector<float> xs = test_sequence(bf->x_range, bf->x_step), ys = test_sequence(bf->y_range, bf->y_step);
    const std::vector<float> zs = test_sequence(bf->z_range, bf->z_step), ws = test_sequence(bf->yaw_range, bf->yaw_step);
    const size_t per_x = ys.size() * zs.size() * ws.size();
    const size_t total = xs.size() * per_x;
    SF_CHECK(per_x > 0 && per_x <= 65535, SF_ERR_INVALID, "candidate slice of %zu poses does not fit one launch", per_x);
    const int n = (int)bf->n;
    const int nblocks = (int)sf::div_up(n, BLK);
    SF_TRY(bf->poses.reserve(sizeof(float) * 12 * per_x));
    SF_CHECK((double)per_x * (double)n < 2.0e9, SF_ERR_OVERFLOW, "%zu candidates x %d points per slice is too large", per_x, n);
    SF_TRY(bf->partials.reserve(sizeof(float) * per_x * (size_t)n)); // one squared distance per (candidate of the slice, point)
    SF_TRY(bf->scores.reserve(sizeof(float) * per_x));
    std::vector<float> T(16 * total), hpose(12 * per_x), hscore(per_x);
    bf->last_scores.assign(total, NAN);
    const float *X = bf->src.as<float>(), *Y = X + n, *Z = X + 2 * (size_t)n;
    float best_T[16], best_score = 3.402823466e+38f;
    for (int i = 0; i < 16; ++i) best_T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    int best_idx = -1;
    *found = 0;
    size_t cand = 0;
    for (size_t a = 0; a < xs.size(); ++a) {
        // one slice = every (y, z, yaw) for this x, in the reference's nesting order
        size_t k = 0;
        for (size_t b = 0; b < ys.size(); ++b)
            for (size_t c = 0; c < zs.size(); ++c)
                for (size_t d = 0; d < ws.size(); ++d, ++k) {
                    const float yaw = ws[d];
                    const float sn = std::sin(yaw), cs = std::cos(yaw), one_c = 1.0f - cs; // AngleAxisf(yaw, UnitZ)
                    const float L[16] = {0.0f + cs, 0.0f - sn, 0.0f, xs[a], 0.0f + sn, 0.0f + cs, 0.0f, ys[b], 0.0f, 0.0f, one_c * 1.0f + cs, zs[c], 0, 0, 0, 1};
                    float *Tk = &T[16 * (cand + k)];
                    mul4(bf->previous, L, Tk);                                             // T = map_T_sensor_previous_ * T
                    for (int i = 0; i < 12; ++i) hpose[12 * k + i] = Tk[i];
                }
        SF_HIP(hipMemcpyAsync(bf->poses.p, hpose.data(), sizeof(float) * 12 * per_x, hipMemcpyHostToDevice, st));
        const dim3 grid((unsigned)nblocks, (unsigned)per_x);
        if (bf->target->window.kind)
            hipLaunchKernelGGL(k_bf_score<true>, grid, dim3(BLK), 0, st, bf->target->grid, bf->target->window, X, Y, Z, n, bf->poses.as<float>(), bf->partials.as<float>());
        else
            hipLaunchKernelGGL(k_bf_score<false>, grid, dim3(BLK), 0, st, bf->target->grid, bf->target->window, X, Y, Z, n, bf->poses.as<float>(), bf->partials.as<float>());
        hipLaunchKernelGGL(k_bf_finish, dim3((unsigned)per_x), dim3(64), 0, st, bf->partials.as<float>(), n, bf->scores.as<float>());
        SF_HIP(hipGetLastError());
        SF_HIP(hipMemcpyAsync(hscore.data(), bf->scores.p, sizeof(float) * per_x, hipMemcpyDeviceToHost, st));
        SF_HIP(hipStreamSynchronize(st));
        for (k = 0; k < per_x; ++k) { // the reference's sequential decision, :107-119
            const float score = hscore[k];
            bf->last_scores[cand + k] = score;
            if (score < best_score) { best_score = score; std::memcpy(best_T, &T[16 * (cand + k)], sizeof(best_T)); best_idx = (int)(cand + k); }
            if (score < bf->threshold) {
                std::memcpy(bf->best, &T[16 * (cand + k)], sizeof(bf->best));
                bf->first_alignment_completed = true;
                bf->last_index = (int)(cand + k);
                bf->last_score = score;
                *found = 1;
                return SF_OK;
            }
        }
        cand += per_x;
    }
    std::memcpy(bf->previous, best_T, sizeof(best_T)); // :123
    bf->last_index = best_idx;
    bf->last_score = best_score;
    if (best_score < bf->threshold) { // :126-131
        std::memcpy(bf->best, best_T, sizeof(best_T));
        bf->first_alignment_completed = true;
        *found = 1;
    }
    return SF_OK;
}

// brute_force_alignment.cpp:143-146
extern "C" int sf_bf_get_best_transformation(sf_bf *bf, float T[16])
{
    SF_CHECK(bf && T, SF_ERR_INVALID, "bad arguments");
    std::memcpy(T, bf->first_alignment_completed ? bf->best : bf->previous, sizeof(float) * 16);
    return SF_OK;
}

extern "C" int sf_bf_last_result(sf_bf *bf, int32_t *index, float *score, int32_t *n_candidates, float *scores, int64_t cap)
{
    SF_CHECK(bf, SF_ERR_INVALID, "bf is NULL");
    if (index) *index = bf->last_index;
    if (score) *score = bf->last_score;
    if (n_candidates) *n_candidates = (int32_t)bf->last_scores.size();
    if (scores) {
        SF_CHECK(cap >= (int64_t)bf->last_scores.size(), SF_ERR_INVALID, "buffer too small");
        std::memcpy(scores, bf->last_scores.data(), sizeof(float) * bf->last_scores.size());
    }
    return SF_OK;
}
