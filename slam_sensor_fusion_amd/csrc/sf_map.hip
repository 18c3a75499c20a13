// sf_map.hip — device-resident uniform-grid index over the WHOLE map (gfx950).
//
// Replaces ICPPointToPoint::setTargetPointCloud (localization/src/icp_point_to_point.cpp:
// 49-55: deep copy + FLANN kd-tree over a 10 m crop, rebuilt every 3 m of travel,
// localization/src/localization_node.cpp:299-305).  The map is indexed once; the crop
// becomes a window predicate (sf_map_window_*).  Layout in HBM:
//   pts4[n]       float4  x, y, z, bitcast(original index), sorted by cell id (x fastest)
//   cell_start[ncell+1] u32  first sorted position of each cell
//   nrm4[n]       float4  normal xyz + neighbour count (sorted order), optional
#include "sf_common.hpp"
#include "sf_nn.hpp"

#include "sf_sort.hpp"
#include <cmath>

namespace {

inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)sf::div_up(n > 0 ? n : 1, b); }

struct GridGeom { float org[3]; float inv_h; int dim[3]; uint64_t ncell; };

// K = uint32_t while the grid has fewer than 2^32 cells, uint64_t beyond (large sparse extents: a dense table over
// 1 km x 1 km x 100 m at 0.25 m is 6.4e9 cells = 25.6 GB of the 288 GB)
template <class K>
__global__ void k_cell_keys(const float *__restrict__ xyz, int64_t n, GridGeom g, K *__restrict__ keys, uint32_t *__restrict__ vals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    K key = (K)g.ncell; // non-finite points sort last and are not indexed (PCL drops them too)
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
        int cx = (int)fminf(fmaxf(floorf((x - g.org[0]) * g.inv_h), 0.0f), (float)(g.dim[0] - 1));
        int cy = (int)fminf(fmaxf(floorf((y - g.org[1]) * g.inv_h), 0.0f), (float)(g.dim[1] - 1));
        int cz = (int)fminf(fmaxf(floorf((z - g.org[2]) * g.inv_h), 0.0f), (float)(g.dim[2] - 1));
        key = ((K)cz * (K)g.dim[1] + (K)cy) * (K)g.dim[0] + (K)cx;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

// Large tables: a point must not fill the empty cells in front of it one by one (a gap can be billions of cells).  The
// END of every non-empty cell is written to the entry of the cell after it (t[key + 1] = position after the run), the
// rest stays 0, and an inclusive MAX scan turns that into cell_start: the largest end among the cells before c is the
// first sorted position whose key is >= c.
template <class K>
__global__ void k_cell_tails(const K *__restrict__ keys, int64_t n_valid, uint32_t *__restrict__ cell_start)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_valid) return;
    const K k = keys[j];
    if (j == n_valid - 1 || keys[j + 1] != k) cell_start[(size_t)k + 1] = (uint32_t)(j + 1);
}

__global__ void k_carry_max(uint32_t *__restrict__ p) { p[0] = max(p[0], p[-1]); }

__global__ void k_gather_sorted(const float *__restrict__ xyz, const uint32_t *__restrict__ vals, int64_t n, float4 *__restrict__ pts4, uint32_t *__restrict__ inv_perm)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t i = vals[j];
    pts4[j] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __uint_as_float(i));
    inv_perm[i] = (uint32_t)j;
}

// cell_start[c] = first sorted position whose key >= c, for c in [0, ncell] (tables below 2^28 cells: the gaps are short)
__global__ void k_cell_bounds(const uint32_t *__restrict__ keys, int64_t n_valid, uint32_t ncell, uint32_t *__restrict__ cell_start)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_valid) return;
    uint32_t k = keys[j];
    uint32_t prev_next = j == 0 ? 0u : keys[j - 1] + 1u;
    for (uint32_t c = prev_next; c <= k; ++c) cell_start[c] = (uint32_t)j;
    if (j == n_valid - 1)
        for (uint32_t c = k + 1; c <= ncell; ++c) cell_start[c] = (uint32_t)n_valid;
}

__global__ void k_fill_u32(uint32_t *p, int64_t n, uint32_t v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

} // namespace

// SfGrid / SfWindow are passed to kernels by value
namespace {
template <bool WINDOW>
__global__ __launch_bounds__(256) void k_map_nn_t(SfGrid g, SfWindow w, const float *__restrict__ q, int64_t n, float thr, int32_t *__restrict__ idx,
                                                 float *__restrict__ d2)
{
    // the wave-cooperative search of the ICP kernel (every lane of the wave takes part, with or without a query)
    __shared__ sf::WaveNN ws[256 / 64];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const float qx = valid ? q[3 * i] : 0.0f, qy = valid ? q[3 * i + 1] : 0.0f, qz = valid ? q[3 * i + 2] : 0.0f;
    const sf::NNHit hit = sf::nn_search_wave<WINDOW, true>(g, w, valid, qx, qy, qz, thr, &ws[threadIdx.x >> 6]);
    if (!valid) return;
    idx[i] = hit.j >= 0 ? (int32_t)__float_as_uint(g.pts[hit.j].w) : -1;
    d2[i] = hit.j >= 0 ? hit.d2 : INFINITY;
}
} // namespace

extern "C" int sf_map_create(sf_ctx *ctx, sf_map **out)
{
    SF_CHECK(ctx && out, SF_ERR_INVALID, "bad arguments");
    sf_map *m = new (std::nothrow) sf_map();
    SF_CHECK(m, SF_ERR_NOMEM, "out of host memory");
    m->ctx = ctx;
    sf::ctx_retain(ctx);
    *out = m;
    return SF_OK;
}

extern "C" void sf_map_destroy(sf_map *m)
{
    if (!m) return;
    hipError_t e = hipStreamSynchronize(m->ctx->stream);
    (void)e;
    m->pts4.release(); m->nrm4.release(); m->cov6.release(); m->d_window.release(); m->cell_start.release(); m->keys.release(); m->vals.release();
    m->keys2.release(); m->vals2.release(); m->inv_perm.release();
    sf_ctx *ctx = m->ctx;
    delete m;
    sf::ctx_release(ctx);
}

extern "C" int sf_map_build(sf_map *m, sf_cloud *cloud, float cell)
{
    SF_CHECK(m && cloud, SF_ERR_INVALID, "bad arguments");
    // the search addresses candidates through a buffer descriptor (byte offsets and size in 32 bits, sf_nn.hpp)
    SF_CHECK(cloud->n < (int64_t)(1 << 28), SF_ERR_OVERFLOW, "map too large for one GPU index (2^28 points): shard it");
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = cloud->n;
    m->built = false;
    m->has_normals = false;
    m->n = 0;
    m->window.kind = 0;
    const float *xyz = cloud->xyz.as<float>();

    // 1. bounds of the finite points
    sf::MinMaxHost mm;
    SF_TRY(sf::cloud_minmax(ctx, xyz, n, &mm));
    const int64_t n_valid = mm.n_finite;

    // 2. geometry: automatic cell ~ 1.5 points per cell, clamped; grow until it fits u32
    double ext[3];
    for (int d = 0; d < 3; ++d) ext[d] = n_valid > 0 ? (double)mm.mx[d] - (double)mm.mn[d] : 0.0;
    double h = cell;
    if (!(h > 0)) {
        double vol = std::max(ext[0], 1e-3) * std::max(ext[1], 1e-3) * std::max(ext[2], 1e-3);
        h = std::cbrt(1.5 * vol / (double)std::max<int64_t>(n_valid, 1));
        h = std::min(std::max(h, 0.05), 1.0e6);
    }
    // the dense cell table may take a quarter of what the device has free (288 GB HBM: a 1 km x 1 km x 100 m extent at
    // 0.25 m is 6.4e9 cells = 25.6 GB), never more than 2^35 cells; automatic sizing grows the cell until it fits, an
    // explicit cell that does not fit is refused
    size_t free_b = 0, total_b = 0;
    SF_HIP(hipMemGetInfo(&free_b, &total_b));
    const double max_cells = std::min(34359738368.0, std::max(1.0e9, (double)free_b / 4.0 / sizeof(uint32_t)));
    int dim[3];
    for (;;) {
        double cells = 1;
        bool ok = true;
        for (int d = 0; d < 3; ++d) {
            double c = std::floor(ext[d] / h) + 1;
            if (c > 2.0e9) ok = false;
            dim[d] = ok ? (int)c : 1;
            cells *= c;
        }
        if (ok && cells <= max_cells) break;
        SF_CHECK(cell <= 0, SF_ERR_OVERFLOW, "cell %.4g gives too many cells for this extent (%.3g; this device holds a table of %.3g)", (double)cell, cells, max_cells);
        h *= 1.5;
    }
    GridGeom g;
    for (int d = 0; d < 3; ++d) { g.org[d] = n_valid > 0 ? mm.mn[d] : 0.0f; g.dim[d] = dim[d]; }
    g.inv_h = (float)(1.0 / h);
    g.ncell = (uint64_t)dim[0] * (uint64_t)dim[1] * (uint64_t)dim[2];
    const bool wide = g.ncell >= 0xffffffffull;     // keys (cell ids, ncell itself for non-finite points) no longer fit 32 bits
    const bool by_scan = g.ncell > (1ull << 28);    // table large enough for long empty stretches: bounds by scan

    // 3. keys -> stable radix sort -> gather
    const size_t np = (size_t)std::max<int64_t>(n, 1), ksz = wide ? sizeof(uint64_t) : sizeof(uint32_t);
    SF_TRY(m->keys.reserve(ksz * np));
    SF_TRY(m->vals.reserve(sizeof(uint32_t) * np));
    SF_TRY(m->keys2.reserve(ksz * np));
    SF_TRY(m->vals2.reserve(sizeof(uint32_t) * np));
    SF_TRY(m->pts4.reserve(sizeof(float4) * np));
    SF_TRY(m->inv_perm.reserve(sizeof(uint32_t) * np));
    SF_TRY(m->cell_start.reserve(sizeof(uint32_t) * ((size_t)g.ncell + 8))); // [pad | start[0..ncell] | pad..]: sf_nn.hpp reads start[c-1..c+2] in one load
    const void *sorted_keys = m->keys2.p; // where the sort leaves its result (either ping-pong buffer)
    uint32_t *sorted_vals = m->vals2.as<uint32_t>();
    if (n > 0) {
        unsigned bits = 1;
        while (bits < 64 && (1ull << bits) <= (unsigned long long)g.ncell) ++bits;
        // stable: points of a cell stay in ascending point id (deterministic in-cell order => deterministic tie-breaks)
        if (wide) {
            hipLaunchKernelGGL(k_cell_keys<uint64_t>, dim3(nblk(n)), dim3(256), 0, st, xyz, n, g, m->keys.as<uint64_t>(), m->vals.as<uint32_t>());
            uint64_t *sk = nullptr;
            SF_TRY(sf::radix_sort_pairs<uint64_t>(ctx, m->keys.as<uint64_t>(), m->keys2.as<uint64_t>(), m->vals.as<uint32_t>(), m->vals2.as<uint32_t>(), n, bits, &sk, &sorted_vals));
            sorted_keys = sk;
        } else {
            hipLaunchKernelGGL(k_cell_keys<uint32_t>, dim3(nblk(n)), dim3(256), 0, st, xyz, n, g, m->keys.as<uint32_t>(), m->vals.as<uint32_t>());
            uint32_t *sk = nullptr;
            SF_TRY(sf::radix_sort_pairs<uint32_t>(ctx, m->keys.as<uint32_t>(), m->keys2.as<uint32_t>(), m->vals.as<uint32_t>(), m->vals2.as<uint32_t>(), n, bits, &sk, &sorted_vals));
            sorted_keys = sk;
        }
        hipLaunchKernelGGL(k_gather_sorted, dim3(nblk(n)), dim3(256), 0, st, xyz, sorted_vals, n, m->pts4.as<float4>(), m->inv_perm.as<uint32_t>());
    }
    uint32_t *cs = m->cell_start.as<uint32_t>() + 1;
    if (by_scan && n_valid > 0) {
        // [pad = 0 | t[0..ncell] | pads]: ends of the non-empty cells, then an inclusive max scan in pieces of 2^30 entries
        // (the carry is the entry before the piece); the trailing pads come out as n_valid like the rest of the tail
        const size_t entries = (size_t)g.ncell + 8;
        SF_HIP(hipMemsetAsync(m->cell_start.p, 0, sizeof(uint32_t) * entries, st));
        if (wide) hipLaunchKernelGGL(k_cell_tails<uint64_t>, dim3(nblk(n_valid)), dim3(256), 0, st, static_cast<const uint64_t *>(sorted_keys), n_valid, cs);
        else hipLaunchKernelGGL(k_cell_tails<uint32_t>, dim3(nblk(n_valid)), dim3(256), 0, st, static_cast<const uint32_t *>(sorted_keys), n_valid, cs);
        const size_t piece = (size_t)1 << 30;
        uint32_t *t = m->cell_start.as<uint32_t>();
        for (size_t off = 0; off < entries; off += piece) {
            const size_t len = std::min(piece, entries - off);
            if (off > 0) hipLaunchKernelGGL(k_carry_max, dim3(1), dim3(1), 0, st, t + off);
            SF_TRY(sf::scan_u32<1>(ctx, t + off, t + off, (int64_t)len)); // (the carry of a later piece sits in its first entry: k_carry_max)
        }
    } else {
        hipLaunchKernelGGL(k_fill_u32, dim3(nblk((int64_t)g.ncell + 8)), dim3(256), 0, st, m->cell_start.as<uint32_t>(), (int64_t)g.ncell + 8, (uint32_t)n_valid);
        SF_HIP(hipMemsetAsync(m->cell_start.p, 0, sizeof(uint32_t), st));
        if (n_valid > 0)
            hipLaunchKernelGGL(k_cell_bounds, dim3(nblk(n_valid)), dim3(256), 0, st, static_cast<const uint32_t *>(sorted_keys), n_valid, (uint32_t)g.ncell, cs);
        else
            hipLaunchKernelGGL(k_fill_u32, dim3(nblk((int64_t)g.ncell + 2)), dim3(256), 0, st, cs, (int64_t)g.ncell + 2, 0u);
    }
    SF_HIP(hipGetLastError());
    SF_HIP(hipStreamSynchronize(st));

    m->n = n;
    SfGrid &G = m->grid;
    for (int d = 0; d < 3; ++d) { G.org[d] = g.org[d]; G.dim[d] = dim[d]; }
    G.inv_h = g.inv_h;
    G.h = (float)h;
    // |computed grid coordinate - true one| <= 2^-23 * coordinate for the point and for the query, so a
    // face distance derived from them can be too long by 2^-22 * (largest coordinate) cells; 1.5 x for the
    // roundings of the gap arithmetic itself.  Every pruning decision of the search subtracts it (sf_nn.hpp).
    G.gap_eps = 1.5f * 2.384186e-7f * (float)std::max(dim[0], std::max(dim[1], dim[2])) * (float)h;
    G.cell_start = m->cell_start.as<uint32_t>() + 1;
    G.pts = m->pts4.as<float4>();
    G.nrm = nullptr;
    G.n = n_valid;
    m->built = true;
    m->has_cov = false;
    m->generation = sf::next_generation();
    return SF_OK;
}

extern "C" int sf_map_size(sf_map *m, int64_t *n)
{
    SF_CHECK(m && n, SF_ERR_INVALID, "bad arguments");
    *n = m->n;
    return SF_OK;
}

extern "C" int sf_map_cell_size(sf_map *m, float *cell, int32_t dims[3])
{
    SF_CHECK(m && m->built, SF_ERR_STATE, "map not built");
    if (cell) *cell = m->grid.h;
    if (dims) for (int d = 0; d < 3; ++d) dims[d] = m->grid.dim[d];
    return SF_OK;
}

extern "C" int sf_map_window_none(sf_map *m)
{
    SF_CHECK(m, SF_ERR_INVALID, "bad arguments");
    m->window.kind = 0;
    return SF_OK;
}

extern "C" int sf_map_window_sphere(sf_map *m, const float center[3], double radius)
{
    SF_CHECK(m && center, SF_ERR_INVALID, "bad arguments");
    m->window.kind = 1;
    for (int d = 0; d < 3; ++d) m->window.c[d] = center[d];
    m->window.r2 = (float)(radius * radius);
    return SF_OK;
}

extern "C" int sf_map_window_obb(sf_map *m, const double center[3], const double R[9], const double extent[3])
{
    SF_CHECK(m && center && R && extent, SF_ERR_INVALID, "bad arguments");
    m->window.kind = 2;
    for (int d = 0; d < 3; ++d) { m->window.oc[d] = center[d]; m->window.ohalf[d] = extent[d] / 2; }
    for (int k = 0; k < 9; ++k) m->window.oR[k] = R[k];
    return SF_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_window_count(SfGrid g, SfWindow w, unsigned long long *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool in = false;
    if (j < g.n) {
        const float4 p = g.pts[j];
        in = sf::window_accepts(w, p.x, p.y, p.z);
    }
    const unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned long long)__popcll(b)); // integer: order independent
}
} // namespace

extern "C" int sf_map_window_count(sf_map *m, int64_t *n)
{
    SF_CHECK(m && m->built && n, SF_ERR_STATE, "map not built");
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    if (m->window.kind == 0 || m->grid.n == 0) { *n = m->grid.n; return SF_OK; }
    SF_TRY(ctx->scratch2.reserve(sizeof(unsigned long long)));
    unsigned long long *d = ctx->scratch2.as<unsigned long long>();
    SF_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_window_count, dim3(nblk(m->grid.n)), dim3(256), 0, ctx->stream, m->grid, m->window, d);
    unsigned long long *h = reinterpret_cast<unsigned long long *>(ctx->h_pinned);
    SF_HIP(hipMemcpyAsync(h, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    *n = (int64_t)h[0];
    return SF_OK;
}

extern "C" int sf_map_nn(sf_map *m, const float *queries, int64_t n, float max_d2, int32_t *idx, float *d2)
{
    SF_CHECK(m && m->built, SF_ERR_STATE, "map not built");
    SF_CHECK(n >= 0 && (n == 0 || (queries && idx && d2)), SF_ERR_INVALID, "bad arguments");
    if (n == 0) return SF_OK;
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    sf::DevBuf dq, di, dd;
    SF_TRY(dq.reserve(sizeof(float) * 3 * (size_t)n));
    SF_TRY(di.reserve(sizeof(int32_t) * (size_t)n));
    SF_TRY(dd.reserve(sizeof(float) * (size_t)n));
    SF_HIP(hipMemcpyAsync(dq.p, queries, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (m->window.kind)
        hipLaunchKernelGGL(k_map_nn_t<true>, dim3(nblk(n)), dim3(256), 0, ctx->stream, m->grid, m->window, dq.as<float>(), n, max_d2, di.as<int32_t>(), dd.as<float>());
    else
        hipLaunchKernelGGL(k_map_nn_t<false>, dim3(nblk(n)), dim3(256), 0, ctx->stream, m->grid, m->window, dq.as<float>(), n, max_d2, di.as<int32_t>(), dd.as<float>());
    SF_HIP(hipGetLastError());
    SF_HIP(hipMemcpyAsync(idx, di.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipMemcpyAsync(d2, dd.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

// ------------------------------------------------------------------ normals (extension x2, no reference code)
// PCA of all neighbours within `radius` (self included), float64, two passes (mean, then
// centred covariance); smallest-eigenvalue vector by cyclic Jacobi.  The batch is
// 3 x k . k x 3 with k ~ 10-30: VALU work, not a dense contraction worth MFMA.
namespace {

__device__ void smallest_eigvec(const double C[9], double nrm[3])
{
    double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) a[i] = C[i];
    for (int sweep = 0; sweep < 50; ++sweep) {
        const double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
        const double dia = fabs(a[0]) + fabs(a[4]) + fabs(a[8]);
        if (off <= 1e-300 || off <= 2.220446049250313e-16 * dia * 1e-3) break;
        for (int k = 0; k < 3; ++k) {
            const int p = k == 2 ? 1 : 0, q = k == 0 ? 1 : 2;
            const double apq = a[3 * p + q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = (a[3 * q + q] - a[3 * p + p]) / (2 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
            const double c = 1 / sqrt(t * t + 1), s = t * c;
            for (int i = 0; i < 3; ++i) {
                const double aip = a[3 * i + p], aiq = a[3 * i + q];
                a[3 * i + p] = c * aip - s * aiq;
                a[3 * i + q] = s * aip + c * aiq;
            }
            for (int i = 0; i < 3; ++i) {
                const double api = a[3 * p + i], aqi = a[3 * q + i];
                a[3 * p + i] = c * api - s * aqi;
                a[3 * q + i] = s * api + c * aqi;
            }
            for (int i = 0; i < 3; ++i) {
                const double vip = v[3 * i + p], viq = v[3 * i + q];
                v[3 * i + p] = c * vip - s * viq;
                v[3 * i + q] = s * vip + c * viq;
            }
        }
    }
    int best = 0;
    if (a[4] < a[3 * best + best]) best = 1;
    if (a[8] < a[3 * best + best]) best = 2;
    double x = v[best], y = v[3 + best], z = v[6 + best];
    const double nn = sqrt(x * x + y * y + z * z);
    if (!(nn > 0)) { nrm[0] = 0; nrm[1] = 0; nrm[2] = 1; return; }
    x /= nn; y /= nn; z /= nn;
    if (z < 0 || (z == 0 && (y < 0 || (y == 0 && x < 0)))) { x = -x; y = -y; z = -z; }
    nrm[0] = x; nrm[1] = y; nrm[2] = z;
}

__global__ __launch_bounds__(256) void k_normals(SfGrid g, double r2, int R, float4 *__restrict__ nrm4, double *__restrict__ cov6)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= g.n) return;
    const float4 p = g.pts[j];
    const int nx = g.dim[0], ny = g.dim[1], nz = g.dim[2];
    const int cx = (int)fminf(fmaxf(floorf((p.x - g.org[0]) * g.inv_h), 0.0f), (float)(nx - 1));
    const int cy = (int)fminf(fmaxf(floorf((p.y - g.org[1]) * g.inv_h), 0.0f), (float)(ny - 1));
    const int cz = (int)fminf(fmaxf(floorf((p.z - g.org[2]) * g.inv_h), 0.0f), (float)(nz - 1));
    const int x0 = max(cx - R, 0), x1 = min(cx + R, nx - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, ny - 1);
    const int z0 = max(cz - R, 0), z1 = min(cz + R, nz - 1);
    double sum[3] = {0, 0, 0}, mean[3] = {0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            if (cnt < 3) break;
            for (int d = 0; d < 3; ++d) mean[d] = sum[d] / cnt;
        }
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const size_t row = ((size_t)z * ny + y) * nx;
                const uint32_t a = g.cell_start[row + x0], b = g.cell_start[row + x1 + 1];
                for (uint32_t k = a; k < b; ++k) {
                    const float4 q = g.pts[k];
                    const double ex = (double)q.x - (double)p.x, ey = (double)q.y - (double)p.y, ez = (double)q.z - (double)p.z;
                    if (!(ex * ex + ey * ey + ez * ez <= r2)) continue;
                    if (pass == 0) { sum[0] += q.x; sum[1] += q.y; sum[2] += q.z; ++cnt; }
                    else {
                        const double ax = q.x - mean[0], ay = q.y - mean[1], az = q.z - mean[2];
                        C[0] += ax * ax; C[1] += ax * ay; C[2] += ax * az; C[3] += ay * ay; C[4] += ay * az; C[5] += az * az;
                    }
                }
            }
    }
    double nv[3] = {0, 0, 1};
    if (cnt >= 3) {
        const double M[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
        smallest_eigvec(M, nv);
    }
    nrm4[j] = make_float4((float)nv[0], (float)nv[1], (float)nv[2], __int_as_float(cnt));
    if (cov6) { // xx xy xz yy yz zz of the centred neighbourhood, divided by the neighbour count (zeros below 3 neighbours)
#pragma unroll
        for (int d = 0; d < 6; ++d) cov6[6 * (size_t)j + d] = cnt >= 3 ? C[d] / (double)cnt : 0.0;
    }
}

__global__ void k_normals_from_host_order(SfGrid g, const float *__restrict__ nrm_orig, float4 *__restrict__ nrm4)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= g.n) return;
    const uint32_t i = __float_as_uint(g.pts[j].w);
    nrm4[j] = make_float4(nrm_orig[3 * (size_t)i], nrm_orig[3 * (size_t)i + 1], nrm_orig[3 * (size_t)i + 2], __int_as_float(0));
}

__global__ void k_cov_to_host_order(SfGrid g, const double *__restrict__ cov6, double *__restrict__ cov_orig)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= g.n) return;
    const uint32_t i = __float_as_uint(g.pts[j].w);
#pragma unroll
    for (int d = 0; d < 6; ++d) cov_orig[6 * (size_t)i + d] = cov6[6 * (size_t)j + d];
}

__global__ void k_normals_to_host_order(SfGrid g, const float4 *__restrict__ nrm4, float *__restrict__ nrm_orig, int32_t *__restrict__ cnt_orig)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= g.n) return;
    const uint32_t i = __float_as_uint(g.pts[j].w);
    const float4 v = nrm4[j];
    nrm_orig[3 * (size_t)i] = v.x; nrm_orig[3 * (size_t)i + 1] = v.y; nrm_orig[3 * (size_t)i + 2] = v.z;
    cnt_orig[i] = __float_as_int(v.w);
}

} // namespace

extern "C" int sf_map_estimate_normals_cov(sf_map *m, float radius, int with_covariance)
{
    SF_CHECK(m && m->built, SF_ERR_STATE, "map not built");
    SF_CHECK(radius > 0, SF_ERR_INVALID, "radius must be positive");
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    SF_TRY(m->nrm4.reserve(sizeof(float4) * (size_t)std::max<int64_t>(m->n, 1)));
    m->has_cov = false;
    if (with_covariance) SF_TRY(m->cov6.reserve(sizeof(double) * 6 * (size_t)std::max<int64_t>(m->n, 1)));
    const int R = std::max(1, (int)std::ceil((double)radius / (double)m->grid.h - 1e-9));
    if (m->grid.n > 0)
        hipLaunchKernelGGL(k_normals, dim3(nblk(m->grid.n)), dim3(256), 0, ctx->stream, m->grid, (double)radius * (double)radius, R, m->nrm4.as<float4>(),
                           with_covariance ? m->cov6.as<double>() : nullptr);
    SF_HIP(hipGetLastError());
    SF_HIP(hipStreamSynchronize(ctx->stream));
    m->grid.nrm = m->nrm4.as<float4>();
    m->has_normals = true;
    m->has_cov = with_covariance != 0;
    m->generation = sf::next_generation();
    return SF_OK;
}

extern "C" int sf_map_estimate_normals(sf_map *m, float radius) { return sf_map_estimate_normals_cov(m, radius, 0); }

extern "C" int sf_map_download_covariances(sf_map *m, double *cov6, int64_t cap, int64_t *n)
{
    SF_CHECK(m && m->built && m->has_cov, SF_ERR_STATE, "no covariances (sf_map_estimate_normals_cov with with_covariance = 1)");
    if (n) *n = m->n;
    SF_CHECK(cap >= m->n && cov6, SF_ERR_INVALID, "buffer too small");
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    sf::DevBuf dc;
    const size_t bytes = sizeof(double) * 6 * (size_t)std::max<int64_t>(m->n, 1);
    SF_TRY(dc.reserve(bytes));
    SF_HIP(hipMemsetAsync(dc.p, 0, bytes, ctx->stream)); // points that are not indexed (non-finite) keep zeros
    if (m->grid.n > 0)
        hipLaunchKernelGGL(k_cov_to_host_order, dim3(nblk(m->grid.n)), dim3(256), 0, ctx->stream, m->grid, m->cov6.as<double>(), dc.as<double>());
    SF_HIP(hipMemcpyAsync(cov6, dc.p, sizeof(double) * 6 * (size_t)m->n, hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_map_set_normals(sf_map *m, const float *normals, int64_t n)
{
    SF_CHECK(m && m->built, SF_ERR_STATE, "map not built");
    SF_CHECK(normals && n == m->n, SF_ERR_INVALID, "normals must match the map size (%lld)", (long long)m->n);
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    SF_TRY(m->nrm4.reserve(sizeof(float4) * (size_t)std::max<int64_t>(m->n, 1)));
    sf::DevBuf tmp;
    SF_TRY(tmp.reserve(sizeof(float) * 3 * (size_t)std::max<int64_t>(n, 1)));
    SF_HIP(hipMemcpyAsync(tmp.p, normals, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (m->grid.n > 0)
        hipLaunchKernelGGL(k_normals_from_host_order, dim3(nblk(m->grid.n)), dim3(256), 0, ctx->stream, m->grid, tmp.as<float>(), m->nrm4.as<float4>());
    SF_HIP(hipStreamSynchronize(ctx->stream));
    m->grid.nrm = m->nrm4.as<float4>();
    m->has_normals = true;
    m->has_cov = false;
    m->generation = sf::next_generation();
    return SF_OK;
}

extern "C" int sf_map_download_normals(sf_map *m, float *normals, int32_t *n_neighbors, int64_t cap, int64_t *n)
{
    SF_CHECK(m && m->built && m->has_normals, SF_ERR_STATE, "no normals");
    if (n) *n = m->n;
    SF_CHECK(cap >= m->n && normals, SF_ERR_INVALID, "buffer too small");
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    sf::DevBuf dn, dc;
    SF_TRY(dn.reserve(sizeof(float) * 3 * (size_t)std::max<int64_t>(m->n, 1)));
    SF_TRY(dc.reserve(sizeof(int32_t) * (size_t)std::max<int64_t>(m->n, 1)));
    SF_HIP(hipMemsetAsync(dn.p, 0, sizeof(float) * 3 * (size_t)std::max<int64_t>(m->n, 1), ctx->stream));
    SF_HIP(hipMemsetAsync(dc.p, 0, sizeof(int32_t) * (size_t)std::max<int64_t>(m->n, 1), ctx->stream));
    if (m->grid.n > 0)
        hipLaunchKernelGGL(k_normals_to_host_order, dim3(nblk(m->grid.n)), dim3(256), 0, ctx->stream, m->grid, m->nrm4.as<float4>(), dn.as<float>(), dc.as<int32_t>());
    SF_HIP(hipMemcpyAsync(normals, dn.p, sizeof(float) * 3 * (size_t)m->n, hipMemcpyDeviceToHost, ctx->stream));
    if (n_neighbors) SF_HIP(hipMemcpyAsync(n_neighbors, dc.p, sizeof(int32_t) * (size_t)m->n, hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}
