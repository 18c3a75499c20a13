// sf_cloud.hip — context, error channel and device point-set operations (gfx950).
//
// Replaces, on the device, the reference's per-scan preprocessing
// (localization/include/localization/point_cloud_processing.hpp:31-92 and
// localization_python/localization_python/localization_node.py:105-115,222-225):
// every crop is a predicate kernel + an order-preserving wave-ballot stream compaction
// instead of "build a kd-tree over the whole cloud for one radius query" (hpp:37-45).
#include "sf_common.hpp"

#include "sf_sort.hpp"

#include <atomic>
#include <cstring>

namespace sf {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int ensure_scratch(sf_ctx *ctx, size_t bytes) { return ctx->scratch.reserve(bytes); }
uint64_t next_generation()
{
    static std::atomic<uint64_t> g{0};
    return ++g;
}
} // namespace sf

extern "C" int sf_version(void) { return SF_VERSION; }
extern "C" const char *sf_last_error(void) { return sf::g_err; }

// ------------------------------------------------------------------ context
extern "C" int sf_ctx_create(int device_id, void *hip_stream, sf_ctx **out)
{
    SF_CHECK(out, SF_ERR_INVALID, "sf_ctx_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        sf::set_error("no HIP device available (%s): libslamfusion has no CPU fallback",
                      e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return SF_ERR_HIP;
    }
    SF_CHECK(device_id >= 0 && device_id < ndev, SF_ERR_INVALID, "device %d out of range (%d)", device_id, ndev);
    SF_HIP(hipSetDevice(device_id));
    sf_ctx *ctx = new (std::nothrow) sf_ctx();
    SF_CHECK(ctx, SF_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
    SF_HIP(hipGetDeviceProperties(&ctx->prop, device_id));
    if (hip_stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
        ctx->own_stream = false;
    } else {
        SF_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    SF_HIP(hipHostMalloc(&ctx->h_pinned, 4096, hipHostMallocDefault));
    *out = ctx;
    return SF_OK;
}

static void ctx_free(sf_ctx *ctx)
{
    hipError_t e = hipStreamSynchronize(ctx->stream);
    (void)e;
    ctx->scratch.release();
    ctx->scratch2.release();
    if (ctx->h_pinned) { e = hipHostFree(ctx->h_pinned); (void)e; }
    for (auto &st : ctx->stage) {
        if (st.p) { e = hipHostFree(st.p); (void)e; }
        if (st.done) { e = hipEventDestroy(st.done); (void)e; }
    }
    if (ctx->own_stream) { e = hipStreamDestroy(ctx->stream); (void)e; }
    delete ctx;
}

namespace sf {
// Measured on the per-scan path (720 KB scan, pageable numpy buffer): hipMemcpyAsync + synchronise 143 us; a memcpy into
// pinned memory + an asynchronous DMA returns after the memcpy.  Uploads above STAGE_MAX go to the runtime directly
// (one-off map loads: pinning gigabytes is not worth it).
constexpr size_t STAGE_MAX = (size_t)64 << 20;
int upload_staged(sf_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return SF_OK;
    hipStream_t s = ctx->stream;
    if (bytes > STAGE_MAX) {
        SF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        SF_HIP(hipStreamSynchronize(s));
        return SF_OK;
    }
    sf_ctx::Stage &st = ctx->stage[ctx->stage_next];
    ctx->stage_next ^= 1;
    if (st.pending) { SF_HIP(hipEventSynchronize(st.done)); st.pending = false; } // its previous copy has left the buffer
    if (st.cap < bytes) {
        if (st.p) { hipError_t e = hipHostFree(st.p); (void)e; st.p = nullptr; st.cap = 0; }
        const size_t want = bytes + (bytes >> 2) + 4096;
        SF_HIP(hipHostMalloc(&st.p, want, hipHostMallocDefault));
        st.cap = want;
    }
    if (!st.done) SF_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    std::memcpy(st.p, src, bytes);
    SF_HIP(hipMemcpyAsync(dst, st.p, bytes, hipMemcpyHostToDevice, s));
    SF_HIP(hipEventRecord(st.done, s));
    st.pending = true;
    return SF_OK;
}

void ctx_retain(sf_ctx *ctx) { ctx->refs += 1; }
void ctx_release(sf_ctx *ctx)
{
    ctx->refs -= 1;
    if (ctx->zombie && ctx->refs <= 0) ctx_free(ctx);
}
} // namespace sf

extern "C" void sf_ctx_destroy(sf_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->refs > 0) { // children still alive: the last one to go frees the context
        hipError_t e = hipStreamSynchronize(ctx->stream);
        (void)e;
        ctx->zombie = true;
        return;
    }
    ctx_free(ctx);
}

extern "C" void *sf_ctx_stream(sf_ctx *ctx) { return ctx ? ctx->stream : nullptr; }

extern "C" int sf_ctx_synchronize(sf_ctx *ctx)
{
    SF_CHECK(ctx, SF_ERR_INVALID, "ctx is NULL");
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_ctx_device_name(sf_ctx *ctx, char *buf, int cap)
{
    SF_CHECK(ctx && buf && cap > 0, SF_ERR_INVALID, "bad arguments");
    snprintf(buf, (size_t)cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    return SF_OK;
}

// ------------------------------------------------------------------ cloud basics
extern "C" int sf_cloud_create(sf_ctx *ctx, sf_cloud **out)
{
    SF_CHECK(ctx && out, SF_ERR_INVALID, "bad arguments");
    sf_cloud *c = new (std::nothrow) sf_cloud();
    SF_CHECK(c, SF_ERR_NOMEM, "out of host memory");
    c->ctx = ctx;
    sf::ctx_retain(ctx);
    *out = c;
    return SF_OK;
}

extern "C" void sf_cloud_destroy(sf_cloud *c)
{
    if (!c) return;
    hipError_t e = hipStreamSynchronize(c->ctx->stream);
    (void)e;
    c->xyz.release(); c->spare.release(); c->flags.release(); c->raw.release(); c->last_idx.release(); c->vox_point_ids.release();
    c->vox_out_ids.release(); c->vox_out_means.release();
    sf_ctx *ctx = c->ctx;
    delete c;
    sf::ctx_release(ctx);
}

static void cloud_reset_meta(sf_cloud *c)
{
    c->n_last_idx = -1;
    c->n_vox_point_vals = c->n_vox_out_vals = c->n_vox_out_pts = 0;
}

extern "C" int sf_cloud_upload(sf_cloud *c, const float *xyz, int64_t n)
{
    SF_CHECK(c && n >= 0 && (xyz || n == 0), SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    SF_TRY(c->xyz.reserve(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1)));
    if (n > 0) SF_TRY(sf::upload_staged(c->ctx, c->xyz.p, xyz, sizeof(float) * 3 * (size_t)n)); // stream-ordered; the caller may free xyz on return
    c->n = n;
    cloud_reset_meta(c);
    return SF_OK;
}

// enqueue only: for PINNED host memory the copy is asynchronous and stream-ordered (the caller keeps the buffer
// untouched until the stream has passed it); pageable memory is staged by the runtime before the call returns
extern "C" int sf_cloud_upload_async(sf_cloud *c, const float *xyz, int64_t n)
{
    SF_CHECK(c && n >= 0 && (xyz || n == 0), SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    SF_TRY(c->xyz.reserve(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1)));
    if (n > 0) SF_HIP(hipMemcpyAsync(c->xyz.p, xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->ctx->stream));
    c->n = n;
    cloud_reset_meta(c);
    return SF_OK;
}

extern "C" void *sf_cloud_device_ptr(sf_cloud *c)
{
    if (c) sf::cloud_touch(c); // whoever holds the pointer may write through it
    return c ? c->xyz.p : nullptr;
}

extern "C" int sf_cloud_upload_f64(sf_cloud *c, const double *xyz, int64_t n)
{
    SF_CHECK(c && n >= 0 && (xyz || n == 0), SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    std::vector<float> tmp((size_t)n * 3);
    for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = (float)xyz[i];
    return sf_cloud_upload(c, tmp.data(), n);
}

extern "C" int sf_cloud_from_device(sf_cloud *c, const void *d_xyz, int64_t n)
{
    SF_CHECK(c && n >= 0 && (d_xyz || n == 0), SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    SF_TRY(c->xyz.reserve(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1)));
    if (n > 0)
        SF_HIP(hipMemcpyAsync(c->xyz.p, d_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToDevice, c->ctx->stream));
    c->n = n;
    cloud_reset_meta(c);
    return SF_OK;
}

extern "C" int sf_cloud_copy(sf_cloud *dst, sf_cloud *src)
{
    SF_CHECK(dst && src, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(dst);
    return sf_cloud_from_device(dst, src->xyz.p, src->n);
}

extern "C" int sf_cloud_size(sf_cloud *c, int64_t *n)
{
    SF_CHECK(c && n, SF_ERR_INVALID, "bad arguments");
    *n = c->n;
    return SF_OK;
}

extern "C" int sf_cloud_download(sf_cloud *c, float *xyz, int64_t cap, int64_t *n)
{
    SF_CHECK(c && (xyz || cap == 0), SF_ERR_INVALID, "bad arguments");
    if (n) *n = c->n;
    SF_CHECK(cap >= c->n, SF_ERR_INVALID, "buffer too small: %lld < %lld", (long long)cap, (long long)c->n);
    if (c->n > 0) {
        SF_HIP(hipMemcpyAsync(xyz, c->xyz.p, sizeof(float) * 3 * (size_t)c->n, hipMemcpyDeviceToHost, c->ctx->stream));
        SF_HIP(hipStreamSynchronize(c->ctx->stream));
    }
    return SF_OK;
}

extern "C" int sf_cloud_last_indices(sf_cloud *c, int32_t *idx, int64_t cap, int64_t *n)
{
    SF_CHECK(c, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(c->n_last_idx >= 0, SF_ERR_STATE, "no crop/subsample has run on this cloud");
    if (n) *n = c->n_last_idx;
    SF_CHECK(cap >= c->n_last_idx && (idx || c->n_last_idx == 0), SF_ERR_INVALID, "buffer too small");
    if (c->n_last_idx > 0) {
        SF_HIP(hipMemcpyAsync(idx, c->last_idx.p, sizeof(int32_t) * (size_t)c->n_last_idx, hipMemcpyDeviceToHost, c->ctx->stream));
        SF_HIP(hipStreamSynchronize(c->ctx->stream));
    }
    return SF_OK;
}

// ------------------------------------------------------------------ stream compaction
// Order-preserving, deterministic, three launches: per-block counts (wave ballot +
// popcount), single-block exclusive scan of the block counts, scatter with the in-wave
// rank from mbcnt.  256 threads = 4 waves of 64.
namespace {

constexpr int CB = 256;

__device__ inline unsigned lane_rank(unsigned long long ballot)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0u));
}

__global__ __launch_bounds__(CB) void k_count_flags(const uint8_t *__restrict__ flags, int64_t n, uint32_t *__restrict__ block_counts)
{
    __shared__ uint32_t wcnt[CB / 64];
    int64_t i = (int64_t)blockIdx.x * CB + threadIdx.x;
    bool keep = i < n && flags[i] != 0;
    unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// exclusive scan of nb values by one block of 1024 threads; writes offsets in place and
// the grand total to total[0]
__global__ __launch_bounds__(1024) void k_scan_blocks(uint32_t *__restrict__ v, int64_t nb, uint32_t *__restrict__ total)
{
    __shared__ uint32_t s[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint32_t x = i < nb ? v[i] : 0u;
        s[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint32_t t = threadIdx.x >= (unsigned)off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t incl = s[threadIdx.x];
        uint32_t c0 = carry;
        if (i < nb) v[i] = c0 + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c0 + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}

__global__ __launch_bounds__(CB) void k_scatter_flags(const uint8_t *__restrict__ flags, int64_t n, const uint32_t *__restrict__ block_off,
                                                      const float *__restrict__ in, float *__restrict__ out, int32_t *__restrict__ out_idx)
{
    __shared__ uint32_t wcnt[CB / 64];
    int64_t i = (int64_t)blockIdx.x * CB + threadIdx.x;
    bool keep = i < n && flags[i] != 0;
    unsigned long long b = __ballot(keep);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) wcnt[w] = (uint32_t)__popcll(b);
    __syncthreads();
    if (!keep) return;
    uint32_t off = block_off[blockIdx.x];
    for (int k = 0; k < w; ++k) off += wcnt[k];
    off += lane_rank(b);
    out[3 * (size_t)off + 0] = in[3 * (size_t)i + 0];
    out[3 * (size_t)off + 1] = in[3 * (size_t)i + 1];
    out[3 * (size_t)off + 2] = in[3 * (size_t)i + 2];
    out_idx[off] = (int32_t)i;
}

} // namespace

namespace sf {
// keeps points whose flag != 0, preserving order; records kept indices in c->last_idx
int compact_cloud(sf_cloud *c, const uint8_t *d_flags)
{
    sf_ctx *ctx = c->ctx;
    int64_t n = c->n;
    if (n == 0) { c->n_last_idx = 0; return SF_OK; }
    int64_t nb = div_up(n, CB);
    SF_TRY(ctx->scratch2.reserve(sizeof(uint32_t) * (size_t)(nb + 1)));
    uint32_t *bc = ctx->scratch2.as<uint32_t>();
    DevBuf &out = c->spare; // ping-pong with xyz: no hipMalloc / hipFree per crop
    SF_TRY(out.reserve(sizeof(float) * 3 * (size_t)n));
    SF_TRY(c->last_idx.reserve(sizeof(int32_t) * (size_t)n));
    hipLaunchKernelGGL(k_count_flags, dim3((unsigned)nb), dim3(CB), 0, ctx->stream, d_flags, n, bc);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, bc, nb, bc + nb);
    hipLaunchKernelGGL(k_scatter_flags, dim3((unsigned)nb), dim3(CB), 0, ctx->stream, d_flags, n, bc, c->xyz.as<float>(), out.as<float>(), c->last_idx.as<int32_t>());
    uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned);
    SF_HIP(hipMemcpyAsync(h, bc + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream)); // the kept count is a host value (sf_cloud_size)
    SF_HIP(hipGetLastError());
    c->xyz.swap(out);
    c->n = h[0];
    c->n_last_idx = c->n;
    return SF_OK;
}
} // namespace sf

// ------------------------------------------------------------------ predicates
namespace {

__global__ void k_flag_radius(const float *__restrict__ xyz, int64_t n, float cx, float cy, float cz, float r2, uint8_t *__restrict__ flags,
                              float *__restrict__ d2out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    bool fin = isfinite(x) && isfinite(y) && isfinite(z);
    // FLANN L2_Simple: diff = a - b with a = query (centre); result += diff*diff, x,y,z
    float dx = cx - x, dy = cy - y, dz = cz - z;
    float d2 = dx * dx;
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    flags[i] = (fin && d2 < r2) ? 1 : 0;
    if (d2out) d2out[i] = d2;
}

__global__ void k_flag_floor(const float *__restrict__ xyz, int64_t n, uint8_t *__restrict__ flags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = xyz[3 * i + 2] > 0.0f ? 1 : 0;
}

struct Box { double lo[3], hi[3]; };
__global__ void k_flag_aabb(const float *__restrict__ xyz, int64_t n, Box b, uint8_t *__restrict__ flags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float xf = xyz[3 * i], yf = xyz[3 * i + 1], zf = xyz[3 * i + 2];
    bool ok = !(isnan(xf) || isnan(yf) || isnan(zf));
    double x = xf, y = yf, z = zf;
    ok = ok && x >= b.lo[0] && x <= b.hi[0] && y >= b.lo[1] && y <= b.hi[1] && z >= b.lo[2] && z <= b.hi[2];
    flags[i] = ok ? 1 : 0;
}

struct Obb { double c[3], R[9], half[3]; };
__global__ void k_flag_obb(const float *__restrict__ xyz, int64_t n, Obb o, uint8_t *__restrict__ flags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d0 = (double)xyz[3 * i] - o.c[0], d1 = (double)xyz[3 * i + 1] - o.c[1], d2 = (double)xyz[3 * i + 2] - o.c[2];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double proj = d0 * o.R[k] + d1 * o.R[3 + k];
        proj = proj + d2 * o.R[6 + k];
        ok = ok && (fabs(proj) <= o.half[k]);
    }
    flags[i] = ok ? 1 : 0;
}

__global__ void k_subsample(const float *__restrict__ in, int64_t n_out, int step, float *__restrict__ out, int32_t *__restrict__ idx)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    int64_t i = k * step;
    out[3 * k] = in[3 * i]; out[3 * k + 1] = in[3 * i + 1]; out[3 * k + 2] = in[3 * i + 2];
    idx[k] = (int32_t)i;
}

struct Aff { float m[12]; };
// icp_point_to_point.cpp:99-110 — separate multiplies and adds, left to right
// (this TU is built with -ffp-contract=off so nothing is fused)
__global__ void k_transform(float *__restrict__ xyz, int64_t n, Aff T)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float tx = T.m[0] * x + T.m[1] * y + T.m[2] * z + T.m[3];
    float ty = T.m[4] * x + T.m[5] * y + T.m[6] * z + T.m[7];
    float tz = T.m[8] * x + T.m[9] * y + T.m[10] * z + T.m[11];
    xyz[3 * i] = tx; xyz[3 * i + 1] = ty; xyz[3 * i + 2] = tz;
}

__global__ void k_pack_d2_idx(const float *__restrict__ d2, const int32_t *__restrict__ src_idx, int64_t n, uint64_t *__restrict__ keys)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // d2 >= 0 and finite: its bit pattern is monotone as an unsigned integer
    keys[i] = ((uint64_t)__float_as_uint(d2[src_idx[i]]) << 32) | (uint32_t)src_idx[i];
}

__global__ void k_gather_by_key(const float *__restrict__ in, const uint64_t *__restrict__ keys, int64_t n, float *__restrict__ out, int32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = (uint32_t)(keys[i] & 0xffffffffu);
    out[3 * i] = in[3 * (size_t)s]; out[3 * i + 1] = in[3 * (size_t)s + 1]; out[3 * i + 2] = in[3 * (size_t)s + 2];
    idx[i] = (int32_t)s;
}

inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)sf::div_up(n > 0 ? n : 1, b); }

} // namespace

extern "C" int sf_cloud_subsample(sf_cloud *c, int step)
{
    SF_CHECK(c && step > 0, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    if (c->n < step) { // point_cloud_processing.hpp:58-61: untouched
        c->n_last_idx = -1;
        return SF_OK;
    }
    int64_t n_out = sf::div_up(c->n, step);
    sf::DevBuf &out = c->spare;
    SF_TRY(out.reserve(sizeof(float) * 3 * (size_t)n_out));
    SF_TRY(c->last_idx.reserve(sizeof(int32_t) * (size_t)n_out));
    hipLaunchKernelGGL(k_subsample, dim3(nblk(n_out)), dim3(256), 0, c->ctx->stream, c->xyz.as<float>(), n_out, step, out.as<float>(), c->last_idx.as<int32_t>());
    SF_HIP(hipGetLastError());
    c->xyz.swap(out); // stream-ordered: the old buffer is only reused by later work on the same stream
    c->n = n_out;
    c->n_last_idx = n_out;
    return SF_OK;
}

extern "C" int sf_cloud_crop_radius(sf_cloud *c, const float center[3], double radius, int sorted)
{
    SF_CHECK(c && center, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    sf_ctx *ctx = c->ctx;
    int64_t n = c->n;
    if (n == 0) { c->n_last_idx = 0; return SF_OK; }
    const float r2 = (float)(radius * radius);
    sf::DevBuf &flags = c->flags;
    sf::DevBuf d2;
    SF_TRY(flags.reserve((size_t)n));
    if (sorted) SF_TRY(d2.reserve(sizeof(float) * (size_t)n));
    hipLaunchKernelGGL(k_flag_radius, dim3(nblk(n)), dim3(256), 0, ctx->stream, c->xyz.as<float>(), n, center[0], center[1], center[2], r2,
                       flags.as<uint8_t>(), sorted ? d2.as<float>() : nullptr);
    sf::DevBuf orig; // compaction replaces c->xyz; the sorted path gathers from the original
    if (sorted) {
        SF_TRY(orig.reserve(sizeof(float) * 3 * (size_t)n));
        SF_HIP(hipMemcpyAsync(orig.p, c->xyz.p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    int rc = sf::compact_cloud(c, flags.as<uint8_t>());
    if (rc == SF_OK && sorted && c->n > 1) {
        int64_t k = c->n;
        sf::DevBuf keys, keys2, out;
        const uint64_t *sorted_keys = nullptr;
        rc = keys.reserve(sizeof(uint64_t) * (size_t)k);
        if (rc == SF_OK) rc = keys2.reserve(sizeof(uint64_t) * (size_t)k);
        if (rc == SF_OK) rc = out.reserve(sizeof(float) * 3 * (size_t)k);
        if (rc == SF_OK) {
            hipLaunchKernelGGL(k_pack_d2_idx, dim3(nblk(k)), dim3(256), 0, ctx->stream, d2.as<float>(), c->last_idx.as<int32_t>(), k, keys.as<uint64_t>());
            // (d2 bits, index) packed: d2 >= 0, so the unsigned order of the float bits is the numeric order; all 64 bits
            uint64_t *sk = nullptr;
            uint32_t *sv = nullptr;
            rc = sf::radix_sort_pairs<uint64_t>(ctx, keys.as<uint64_t>(), keys2.as<uint64_t>(), nullptr, nullptr, k, 64, &sk, &sv);
            sorted_keys = sk;
            if (rc == SF_OK) {
                hipLaunchKernelGGL(k_gather_by_key, dim3(nblk(k)), dim3(256), 0, ctx->stream, orig.as<float>(), sorted_keys, k, out.as<float>(), c->last_idx.as<int32_t>());
                hipError_t e2 = hipStreamSynchronize(ctx->stream);
                if (e2 != hipSuccess) { sf::set_error("sync: %s", hipGetErrorString(e2)); rc = SF_ERR_HIP; }
                else c->xyz.swap(out);
            }
        }
        keys.release(); keys2.release(); out.release();
    }
    if (sorted) { // d2 / orig are freed on return
        hipError_t es = hipStreamSynchronize(ctx->stream);
        (void)es;
    }
    return rc;
}

extern "C" int sf_cloud_remove_floor(sf_cloud *c)
{
    SF_CHECK(c, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    if (c->n == 0) { c->n_last_idx = 0; return SF_OK; }
    sf::DevBuf &flags = c->flags;
    SF_TRY(flags.reserve((size_t)c->n));
    hipLaunchKernelGGL(k_flag_floor, dim3(nblk(c->n)), dim3(256), 0, c->ctx->stream, c->xyz.as<float>(), c->n, flags.as<uint8_t>());
    return sf::compact_cloud(c, flags.as<uint8_t>());
}

extern "C" int sf_cloud_crop_aabb(sf_cloud *c, const double lo[3], const double hi[3])
{
    SF_CHECK(c && lo && hi, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    if (c->n == 0) { c->n_last_idx = 0; return SF_OK; }
    Box b;
    for (int d = 0; d < 3; ++d) { b.lo[d] = lo[d]; b.hi[d] = hi[d]; }
    sf::DevBuf &flags = c->flags;
    SF_TRY(flags.reserve((size_t)c->n));
    hipLaunchKernelGGL(k_flag_aabb, dim3(nblk(c->n)), dim3(256), 0, c->ctx->stream, c->xyz.as<float>(), c->n, b, flags.as<uint8_t>());
    return sf::compact_cloud(c, flags.as<uint8_t>());
}

extern "C" int sf_cloud_crop_obb(sf_cloud *c, const double center[3], const double R[9], const double extent[3])
{
    SF_CHECK(c && center && R && extent, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    if (c->n == 0) { c->n_last_idx = 0; return SF_OK; }
    Obb o;
    for (int d = 0; d < 3; ++d) { o.c[d] = center[d]; o.half[d] = extent[d] / 2; }
    for (int k = 0; k < 9; ++k) o.R[k] = R[k];
    sf::DevBuf &flags = c->flags;
    SF_TRY(flags.reserve((size_t)c->n));
    hipLaunchKernelGGL(k_flag_obb, dim3(nblk(c->n)), dim3(256), 0, c->ctx->stream, c->xyz.as<float>(), c->n, o, flags.as<uint8_t>());
    return sf::compact_cloud(c, flags.as<uint8_t>());
}

extern "C" int sf_cloud_transform(sf_cloud *c, const float T[16])
{
    SF_CHECK(c && T, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_HIP(hipSetDevice(c->ctx->device));
    if (c->n == 0) return SF_OK;
    Aff a;
    for (int k = 0; k < 12; ++k) a.m[k] = T[k];
    hipLaunchKernelGGL(k_transform, dim3(nblk(c->n)), dim3(256), 0, c->ctx->stream, c->xyz.as<float>(), c->n, a);
    SF_HIP(hipGetLastError());
    return SF_OK;
}

// *map_cloud += *cloud (global_map_frames_manager.cpp:131): dst <- [dst; src], on the device.  Together with
// sf_cloud_transform, sf_cloud_voxel_downsample and sf_map_build this is incremental map growth.
extern "C" int sf_cloud_append(sf_cloud *dst, const sf_cloud *src)
{
    SF_CHECK(dst && src && dst != src, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(dst);
    SF_CHECK(dst->ctx->device == src->ctx->device, SF_ERR_INVALID, "clouds live on different devices");
    SF_CHECK(dst->n + src->n < (int64_t)0x7fffffff, SF_ERR_OVERFLOW, "cloud too large for 32-bit point ids");
    SF_HIP(hipSetDevice(dst->ctx->device));
    if (src->n == 0) return SF_OK;
    hipStream_t s = dst->ctx->stream;
    if (src->ctx->stream != s) SF_HIP(hipStreamSynchronize(src->ctx->stream)); // src must be complete before dst's stream reads it
    const size_t b0 = sizeof(float) * 3 * (size_t)dst->n, b1 = sizeof(float) * 3 * (size_t)src->n;
    if (b0 + b1 > dst->xyz.cap) { // DevBuf::reserve does not keep the content: grow by hand (1.5 x so repeated appends stay linear)
        sf::DevBuf nb;
        SF_TRY(nb.reserve(b0 + b1 + ((b0 + b1) >> 1)));
        if (b0) SF_HIP(hipMemcpyAsync(nb.p, dst->xyz.p, b0, hipMemcpyDeviceToDevice, s));
        SF_HIP(hipStreamSynchronize(s));
        dst->xyz.swap(nb); // nb (the old allocation) is freed on scope exit
    }
    SF_HIP(hipMemcpyAsync(static_cast<char *>(dst->xyz.p) + b0, src->xyz.p, b1, hipMemcpyDeviceToDevice, s));
    dst->n += src->n;
    dst->n_last_idx = -1;
    dst->n_vox_point_vals = dst->n_vox_out_vals = dst->n_vox_out_pts = 0;
    return SF_OK;
}

// ------------------------------------------------------------------ bounds of the finite points
namespace {
using sf::MinMaxDev;

__global__ __launch_bounds__(256) void k_minmax_partial(const float *__restrict__ xyz, int64_t n, MinMaxDev *__restrict__ part)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned long long cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
            mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
            ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], off));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off));
        }
        cnt += __shfl_xor(cnt, off);
    }
    __shared__ MinMaxDev s[4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int d = 0; d < 3; ++d) { s[w].mn[d] = mn[d]; s[w].mx[d] = mx[d]; }
        s[w].cnt = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxDev r = s[0];
        for (int k = 1; k < 4; ++k) {
            for (int d = 0; d < 3; ++d) { r.mn[d] = fminf(r.mn[d], s[k].mn[d]); r.mx[d] = fmaxf(r.mx[d], s[k].mx[d]); }
            r.cnt += s[k].cnt;
        }
        part[blockIdx.x] = r;
    }
}

// one wave: lane l folds partials l, l + 64, ... and the 64 lanes fold by shuffles (a single thread walking 1024
// partials with dependent loads took 160 us -- measured -- on the per-scan path)
__global__ __launch_bounds__(64) void k_minmax_final(const MinMaxDev *__restrict__ part, int nb, MinMaxDev *__restrict__ out)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned long long cnt = 0;
    for (int k = threadIdx.x; k < nb; k += 64) {
        const MinMaxDev p = part[k];
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], p.mn[d]); mx[d] = fmaxf(mx[d], p.mx[d]); }
        cnt += p.cnt;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], off));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off));
        }
        cnt += __shfl_xor(cnt, off);
    }
    if (threadIdx.x != 0) return;
    MinMaxDev r;
    for (int d = 0; d < 3; ++d) { r.mn[d] = cnt ? mn[d] : 0.0f; r.mx[d] = cnt ? mx[d] : 0.0f; }
    r.cnt = cnt;
    *out = r;
}
} // namespace

namespace sf {
int cloud_minmax_enqueue(sf_ctx *ctx, const float *d_xyz, int64_t n, MinMaxDev *d_out)
{
    const int nb = n > 0 ? (int)std::min<int64_t>(1024, div_up(n, 256)) : 0;
    SF_TRY(ctx->scratch2.reserve(sizeof(MinMaxDev) * (size_t)(nb + 1)));
    MinMaxDev *part = ctx->scratch2.as<MinMaxDev>();
    if (nb > 0) hipLaunchKernelGGL(k_minmax_partial, dim3(nb), dim3(256), 0, ctx->stream, d_xyz, n, part);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(64), 0, ctx->stream, part, nb, d_out);
    SF_HIP(hipGetLastError());
    return SF_OK;
}

int cloud_minmax(sf_ctx *ctx, const float *d_xyz, int64_t n, MinMaxHost *out)
{
    for (int d = 0; d < 3; ++d) { out->mn[d] = 0; out->mx[d] = 0; }
    out->n_finite = 0;
    if (n <= 0) return SF_OK;
    const int nb = (int)std::min<int64_t>(1024, div_up(n, 256));
    SF_TRY(ctx->scratch2.reserve(sizeof(MinMaxDev) * (size_t)(nb + 2)));
    MinMaxDev *part = ctx->scratch2.as<MinMaxDev>();
    SF_TRY(cloud_minmax_enqueue(ctx, d_xyz, n, part + nb + 1));
    part = ctx->scratch2.as<MinMaxDev>();
    MinMaxDev h;
    SF_HIP(hipMemcpyAsync(&h, part + nb + 1, sizeof(MinMaxDev), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    for (int d = 0; d < 3; ++d) { out->mn[d] = h.mn[d]; out->mx[d] = h.mx[d]; }
    out->n_finite = (int64_t)h.cnt;
    return SF_OK;
}
} // namespace sf

// ------------------------------------------------------------------ PointCloud2 unpack
// pcl::fromROSMsg (localization/src/localization_node.cpp:290-291) / pc2.read_points
// (localization_python/.../localization_node.py:106-111) on the device: the raw message buffer
// is uploaded once and the x, y, z float32 fields are gathered from their byte offsets.
namespace {
// point i lives at row (i / width) * row_step + (i % width) * point_step; fields are read byte-wise (they need
// not be aligned in the message); F64: rounded to float32 like the field mapping of pcl::fromROSMsg
template <bool F64>
__global__ void k_unpack_pc2(const uint8_t *__restrict__ raw, int64_t n, int64_t width, int point_step, int64_t row_step, int ox, int oy, int oz, float *__restrict__ xyz)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = raw + (size_t)(i / width) * (size_t)row_step + (size_t)(i % width) * (size_t)point_step;
    auto rd = [&](int off) -> float {
        uint32_t lo = (uint32_t)p[off] | ((uint32_t)p[off + 1] << 8) | ((uint32_t)p[off + 2] << 16) | ((uint32_t)p[off + 3] << 24);
        if (!F64) return __uint_as_float(lo);
        uint32_t hi = (uint32_t)p[off + 4] | ((uint32_t)p[off + 5] << 8) | ((uint32_t)p[off + 6] << 16) | ((uint32_t)p[off + 7] << 24);
        return (float)__hiloint2double((int)hi, (int)lo);
    };
    xyz[3 * i] = rd(ox); xyz[3 * i + 1] = rd(oy); xyz[3 * i + 2] = rd(oz);
}
} // namespace

extern "C" int sf_cloud_from_pointcloud2_msg(sf_cloud *c, const void *data, int64_t data_bytes, int64_t width, int64_t height, int point_step, int64_t row_step,
                                             int off_x, int off_y, int off_z, int datatype, int is_bigendian)
{
    SF_CHECK(c && width >= 0 && height >= 0 && data_bytes >= 0, SF_ERR_INVALID, "bad arguments");
    sf::cloud_touch(c);
    SF_CHECK(!is_bigendian, SF_ERR_INVALID, "big-endian PointCloud2 payloads are not supported");
    SF_CHECK(datatype == SF_PC2_FLOAT32 || datatype == SF_PC2_FLOAT64, SF_ERR_INVALID, "x/y/z datatype %d: FLOAT32 (7) or FLOAT64 (8) expected", datatype);
    const int fsz = datatype == SF_PC2_FLOAT64 ? 8 : 4;
    SF_CHECK(point_step >= 3 * fsz && off_x >= 0 && off_y >= 0 && off_z >= 0 && off_x + fsz <= point_step && off_y + fsz <= point_step && off_z + fsz <= point_step,
             SF_ERR_INVALID, "field offsets do not fit point_step %d", point_step);
    SF_CHECK(height == 0 || width < ((int64_t)1 << 31) / height, SF_ERR_OVERFLOW, "too many points");
    const int64_t n_points = width * height;
    if (row_step == 0) row_step = width * (int64_t)point_step;
    SF_CHECK(row_step >= width * (int64_t)point_step, SF_ERR_INVALID, "row_step %lld is shorter than width x point_step", (long long)row_step);
    const int64_t need = n_points == 0 ? 0 : (height - 1) * row_step + width * (int64_t)point_step;
    SF_CHECK(data_bytes >= need && (data || need == 0), SF_ERR_INVALID, "PointCloud2 data holds %lld bytes, %lld x %lld points need %lld", (long long)data_bytes,
             (long long)width, (long long)height, (long long)need);
    SF_HIP(hipSetDevice(c->ctx->device));
    SF_TRY(c->xyz.reserve(sizeof(float) * 3 * (size_t)(n_points > 0 ? n_points : 1)));
    if (n_points > 0) {
        SF_TRY(c->raw.reserve((size_t)need)); // persistent staging: no hipMalloc / hipFree per scan
        SF_TRY(sf::upload_staged(c->ctx, c->raw.p, data, (size_t)need)); // through pinned staging; the host buffer is free on return
        if (datatype == SF_PC2_FLOAT64)
            hipLaunchKernelGGL(k_unpack_pc2<true>, dim3(nblk(n_points)), dim3(256), 0, c->ctx->stream, c->raw.as<uint8_t>(), n_points, width, point_step, row_step, off_x, off_y, off_z, c->xyz.as<float>());
        else
            hipLaunchKernelGGL(k_unpack_pc2<false>, dim3(nblk(n_points)), dim3(256), 0, c->ctx->stream, c->raw.as<uint8_t>(), n_points, width, point_step, row_step, off_x, off_y, off_z, c->xyz.as<float>());
        SF_HIP(hipGetLastError());
    }
    c->n = n_points;
    cloud_reset_meta(c);
    return SF_OK;
}

// the unchecked legacy form: n_points tightly packed points of point_step bytes (the caller vouches for the buffer length)
extern "C" int sf_cloud_from_pointcloud2(sf_cloud *c, const void *data, int64_t n_points, int point_step, int off_x, int off_y, int off_z)
{
    SF_CHECK(n_points >= 0 && point_step > 0, SF_ERR_INVALID, "bad arguments");
    return sf_cloud_from_pointcloud2_msg(c, data, n_points * (int64_t)point_step, n_points, n_points > 0 ? 1 : 0, point_step, 0, off_x, off_y, off_z, SF_PC2_FLOAT32, 0);
}
