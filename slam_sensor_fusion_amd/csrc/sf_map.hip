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
#include <cstring>

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

__global__ void k_gather_sorted(const float *__restrict__ xyz, const uint32_t *__restrict__ vals, int64_t n, float4 *__restrict__ pts4)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t i = vals[j];
    pts4[j] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __uint_as_float(i));
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

namespace {
// The grid's origin: the smallest coordinate, or -- sf_map_set_origin_lattice -- that coordinate snapped DOWN to a multiple of
// `lattice` cells, so that a map which grows by a few metres in ANY direction keeps its origin, every point its cell
// coordinates and the index its order (sf_map_patch).  The result is a float <= mn.
inline float grid_origin(float mn, double h, int lattice)
{
    if (lattice <= 0) return mn;
    const double step = (double)lattice * h;
    return (float)(std::floor((double)mn / step) * step);
}

inline bool table_by_scan(uint64_t ncell, int64_t n_points, int lattice) { return ncell > (1ull << 28) || lattice > 0 || ncell > 16ull * (uint64_t)std::max<int64_t>(n_points, 1); }

// cell_start[0 .. ncell] (+ pads) from the sorted keys of the n_valid indexed points; the buffer is reserved by the caller
int build_cell_table(sf_map *m, const GridGeom &g, const void *sorted_keys, int64_t n_valid, bool wide, bool by_scan)
{
    sf_ctx *ctx = m->ctx;
    hipStream_t st = ctx->stream;
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
    return SF_OK;
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
    m->keys2.release(); m->vals2.release(); m->pts4_alt.release(); m->patch_tmp.release();
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
    if (n > 0 && cloud->stamp != 0 && cloud->bounds_stamp == cloud->stamp) { // left behind by a voxel merge: all points finite, nothing has changed since
        for (int d = 0; d < 3; ++d) { mm.mn[d] = cloud->bounds_mn[d]; mm.mx[d] = cloud->bounds_mx[d]; }
        mm.n_finite = n;
    } else {
        SF_TRY(sf::cloud_minmax(ctx, xyz, n, &mm));
    }
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
    float org[3] = {0, 0, 0};
    for (;;) {
        h = (double)(float)h; // the cell is a float32 everywhere else (sf_map_cell_size, SfGrid::h): a rebuild with the reported cell is the same index
        double cells = 1;
        bool ok = true;
        for (int d = 0; d < 3; ++d) {
            org[d] = n_valid > 0 ? grid_origin(mm.mn[d], h, m->origin_lattice) : 0.0f;
            double c = std::floor((n_valid > 0 ? (double)mm.mx[d] - (double)org[d] : 0.0) / h) + 1;
            if (c > 2.0e9) ok = false;
            dim[d] = ok ? (int)c : 1;
            cells *= c;
        }
        if (ok && cells <= max_cells) break;
        SF_CHECK(cell <= 0, SF_ERR_OVERFLOW, "cell %.4g gives too many cells for this extent (%.3g; this device holds a table of %.3g)", (double)cell, cells, max_cells);
        h *= 1.5;
    }
    GridGeom g;
    for (int d = 0; d < 3; ++d) { g.org[d] = org[d]; g.dim[d] = dim[d]; }
    g.inv_h = (float)(1.0 / h);
    g.ncell = (uint64_t)dim[0] * (uint64_t)dim[1] * (uint64_t)dim[2];
    const bool wide = g.ncell >= 0xffffffffull;     // keys (cell ids, ncell itself for non-finite points) no longer fit 32 bits
    // long empty stretches -- a large or sparsely filled table, the margin in front of a snapped origin (whole empty layers) --:
    // bounds by scan (k_cell_bounds has ONE lane walk the gap in front of its point: 0.2 s for 16 M empty cells)
    const bool by_scan = table_by_scan(g.ncell, n_valid, m->origin_lattice);

    // 3. keys -> stable radix sort -> gather
    const size_t np = (size_t)std::max<int64_t>(n, 1), ksz = wide ? sizeof(uint64_t) : sizeof(uint32_t);
    SF_TRY(m->keys.reserve(ksz * np));
    SF_TRY(m->vals.reserve(sizeof(uint32_t) * np));
    SF_TRY(m->keys2.reserve(ksz * np));
    SF_TRY(m->vals2.reserve(sizeof(uint32_t) * np));
    SF_TRY(m->pts4.reserve(sizeof(float4) * np));
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
        hipLaunchKernelGGL(k_gather_sorted, dim3(nblk(n)), dim3(256), 0, st, xyz, sorted_vals, n, m->pts4.as<float4>());
    }
    SF_TRY(build_cell_table(m, g, sorted_keys, n_valid, wide, by_scan));
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
    m->h_exact = h;
    for (int d = 0; d < 3; ++d) { m->src_mn[d] = n_valid > 0 ? mm.mn[d] : 0.0f; m->src_mx[d] = n_valid > 0 ? mm.mx[d] : 0.0f; }
    m->src_stamp = cloud->stamp;
    return SF_OK;
}

// ------------------------------------------------------------------ the index carried over a map growth step
// `*map_cloud += *cloud` + voxel filter + setTargetPointCloud (global_map_frames_manager.cpp:131,142-146,
// icp_point_to_point.cpp:49-55) when the filter ran as a merge (sf_cloud_voxel_merge): the merge knows which old points
// stay (their ids move up by the fresh voxels in front of them), which are replaced by a new centroid and which voxels are
// new.  With the grid origin unchanged, a point's cell coordinates are what they were, the cell-sorted order of the points that
// stay is what it was (ascending id inside a cell, and ids keep their order), and the new index is a MERGE of two sorted
// sequences: the old entries minus the replaced ones, and the centroids sorted by (cell, id).  One streaming pass over the
// old entries instead of four sort passes and a random gather; the result is bit-identical to sf_map_build of the merged
// cloud with the same cell (tests/test_gpu_map_growth.py).  Anything this cannot express -- the smallest coordinate of
// the map changed, a point that sat clamped at the old upper face, 64-bit cell ids -- takes the build.
namespace {

struct PatchGeom { float org[3]; float inv_h; int dim[3]; int old_dim[3]; };

__device__ __forceinline__ uint32_t patch_key(const PatchGeom &g, float x, float y, float z, bool *moved)
{
    // the coordinates before the clamp to the upper face, then under the new and under the old grid
    const int rx = (int)fminf(fmaxf(floorf((x - g.org[0]) * g.inv_h), 0.0f), 2.0e9f), ry = (int)fminf(fmaxf(floorf((y - g.org[1]) * g.inv_h), 0.0f), 2.0e9f),
              rz = (int)fminf(fmaxf(floorf((z - g.org[2]) * g.inv_h), 0.0f), 2.0e9f);
    const int cx = min(rx, g.dim[0] - 1), cy = min(ry, g.dim[1] - 1), cz = min(rz, g.dim[2] - 1);
    if (moved) *moved = cx != min(rx, g.old_dim[0] - 1) || cy != min(ry, g.old_dim[1] - 1) || cz != min(rz, g.old_dim[2] - 1); // one of the grids clamps it into another cell
    return ((uint32_t)cz * (uint32_t)g.dim[1] + (uint32_t)cy) * (uint32_t)g.dim[0] + (uint32_t)cx;
}

struct PatchFlags { uint32_t moved, pad; };

// thread t in two roles.  As voxel g = t: key of its centroid, and the old entry it replaces marked in the bitmap over the old
// sorted positions (+ the count of its block of 256 positions)
__global__ void k_patch_groups(PatchGeom geo, const uint32_t *__restrict__ g_rank, const uint32_t *__restrict__ g_fresh, const float *__restrict__ g_centroid,
                               const float *__restrict__ g_old, int64_t n_groups, const float4 *__restrict__ pts, int64_t n_old, uint32_t *__restrict__ ins_key,
                               uint32_t *__restrict__ ins_val, uint32_t *__restrict__ bitmap, uint32_t *__restrict__ blk_cnt, PatchFlags *__restrict__ ext)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    ins_key[g] = patch_key(geo, g_centroid[3 * g], g_centroid[3 * g + 1], g_centroid[3 * g + 2], nullptr);
    ins_val[g] = (uint32_t)g;
    if (!g_fresh[g]) { // the old point of this voxel: where it lies in the index (entries ascend in (cell, id))
        const uint32_t r = g_rank[g], key = patch_key(geo, g_old[3 * g], g_old[3 * g + 1], g_old[3 * g + 2], nullptr);
        int64_t lo = 0, hi = n_old;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            const float4 a = pts[mid];
            const uint32_t k = patch_key(geo, a.x, a.y, a.z, nullptr);
            if (k < key || (k == key && __float_as_uint(a.w) < r)) lo = mid + 1;
            else hi = mid;
        }
        if (lo >= n_old || __float_as_uint(pts[lo].w) != r) { ext->moved = 1u; return; } // (an order the search cannot follow: the build decides)
        const uint32_t p = (uint32_t)lo;
        atomicOr(&bitmap[p >> 5], 1u << (p & 31u));
        atomicAdd(&blk_cnt[p >> 8], 1u);
    }
}

// sorted centroid e -> (cell, rank among the old ids) in one word: old entry (cell', id) comes after it iff packed(e) <= (cell' << 32 | id)
__global__ void k_patch_pack(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval, const uint32_t *__restrict__ g_rank, int64_t n_groups, uint64_t *__restrict__ packed)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_groups) packed[e] = ((uint64_t)skey[e] << 32) | (uint64_t)g_rank[sval[e]];
}

// per block of 256 old entries: its candidates among the sorted centroids, [first with cell >= the block's first cell, first with
// cell > its last cell) -- searched by one thread per bound, all blocks at once (inside k_patch_old the same two searches
// were a serial chain of ~18 dependent loads in front of every workgroup)
__global__ void k_patch_ranges(PatchGeom geo, const float4 *__restrict__ pts, int64_t n_old, const uint64_t *__restrict__ packed, int64_t n_groups, int64_t n_blocks,
                               uint32_t *__restrict__ range)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n_blocks) return;
    const int64_t blk = t >> 1, p0 = blk * 256;
    const bool upper = (t & 1) != 0;
    const float4 a = pts[upper ? min(p0 + 255, n_old - 1) : p0];
    const uint64_t k = (uint64_t)patch_key(geo, a.x, a.y, a.z, nullptr) << 32;
    const uint64_t v = upper ? (k | 0xffffffffull) : k;
    int64_t lo = 0, hi = n_groups;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (upper ? packed[mid] <= v : packed[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    range[t] = (uint32_t)lo;
}

__device__ __forceinline__ uint32_t patch_del_before(const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ blk_pre, uint32_t p)
{
    const uint32_t blk = p >> 8, w = (p & 255u) >> 5;
    uint32_t c = blk_pre[blk];
    for (uint32_t k = 0; k < w; ++k) c += (uint32_t)__popc(bitmap[blk * 8u + k]);
    return c + (uint32_t)__popc(bitmap[blk * 8u + w] & ((1u << (p & 31u)) - 1u));
}

// one lane per OLD entry, in sorted order: where it goes, with its new id
__global__ __launch_bounds__(256) void k_patch_old(PatchGeom geo, const float4 *__restrict__ pts, int64_t n_old, const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ blk_pre,
                                                    const uint64_t *__restrict__ packed, const uint32_t *__restrict__ range, const uint32_t *__restrict__ coarse,
                                                    const uint32_t *__restrict__ fresh_rank,
                                                    float4 *__restrict__ pts_out, uint32_t *__restrict__ keys_out, PatchFlags *__restrict__ ext)
{
    __shared__ uint32_t words[8], wpre[8];
    const int64_t p0 = (int64_t)blockIdx.x * 256, p = p0 + threadIdx.x;
    if (threadIdx.x >= 64 && threadIdx.x < 72) words[threadIdx.x - 64] = bitmap[(size_t)blockIdx.x * 8 + (threadIdx.x - 64)];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = blk_pre[blockIdx.x];
        for (int k = 0; k < 8; ++k) { wpre[k] = c; c += (uint32_t)__popc(words[k]); }
    }
    __syncthreads();
    if (p >= n_old) return;
    const uint32_t w = threadIdx.x >> 5, bit = threadIdx.x & 31u;
    if ((words[w] >> bit) & 1u) return; // replaced by a centroid
    const float4 a = pts[p];
    bool moved;
    const uint32_t key = patch_key(geo, a.x, a.y, a.z, &moved);
    if (moved) ext->moved = 1u;
    const uint32_t j = __float_as_uint(a.w);
    const uint64_t mine = ((uint64_t)key << 32) | (uint64_t)j;
    int64_t lo = range[2 * (size_t)blockIdx.x], hi = range[2 * (size_t)blockIdx.x + 1]; // centroids that sort before this entry: packed <= mine
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (packed[mid] <= mine) lo = mid + 1;
        else hi = mid;
    }
    uint32_t f = coarse[j >> 6];
    const uint32_t f_end = coarse[(j >> 6) + 1];
    while (f < f_end && fresh_rank[f] <= j) ++f;
    const uint32_t del = wpre[w] + (uint32_t)__popc(words[w] & ((1u << bit) - 1u));
    const size_t o = (size_t)p - (size_t)del + (size_t)lo;
    const uint32_t id = j + f;
    pts_out[o] = make_float4(a.x, a.y, a.z, __uint_as_float(id));
    keys_out[o] = key;
}

// one lane per centroid, in (cell, id) order: the old entries in front of it, less the replaced ones, plus its own rank
__global__ __launch_bounds__(256) void k_patch_new(PatchGeom geo, const float4 *__restrict__ pts, int64_t n_old, const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ blk_pre,
                                                    const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval, int64_t n_groups, const uint32_t *__restrict__ g_rank,
                                                    const uint32_t *__restrict__ fresh_pos, const float *__restrict__ g_centroid, float4 *__restrict__ pts_out,
                                                    uint32_t *__restrict__ keys_out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_groups) return;
    const uint32_t key = skey[e], g = sval[e], r = g_rank[g];
    int64_t lo = 0, hi = n_old; // old entries with (cell, id) < (key, r)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const float4 a = pts[mid];
        const uint32_t k = patch_key(geo, a.x, a.y, a.z, nullptr);
        if (k < key || (k == key && __float_as_uint(a.w) < r)) lo = mid + 1;
        else hi = mid;
    }
    const size_t o = (size_t)e + (size_t)lo - (size_t)patch_del_before(bitmap, blk_pre, (uint32_t)lo);
    const uint32_t id = r + fresh_pos[g];
    pts_out[o] = make_float4(g_centroid[3 * (size_t)g], g_centroid[3 * (size_t)g + 1], g_centroid[3 * (size_t)g + 2], __uint_as_float(id));
    keys_out[o] = key;
}

} // namespace

extern "C" int sf_map_patch(sf_map *m, sf_cloud *cloud, int *patched)
{
    SF_CHECK(m && cloud, SF_ERR_INVALID, "bad arguments");
    SF_CHECK(m->built, SF_ERR_STATE, "sf_map_patch: the map has no index yet (sf_map_build)");
    SF_CHECK(m->ctx == cloud->ctx, SF_ERR_INVALID, "map and cloud must live on the same context");
    if (patched) *patched = 0;
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const float cell = (float)m->h_exact;
    const sf_cloud::MergeRecord &rec = cloud->merge;
    const SfGrid old = m->grid;
    const uint64_t old_cells = (uint64_t)old.dim[0] * (uint64_t)old.dim[1] * (uint64_t)old.dim[2];
    const int64_t n_old = rec.n_old, ng = rec.n_groups, n_out = cloud->n;
    auto rebuild = [&](int why) -> int { // *patched: 0 / negative = the build ran, and why
        if (patched) *patched = why;
        return sf_map_build(m, cloud, cell);
    };
    if (!rec.valid || rec.epoch != ctx->merge_epoch || rec.stamp_after != cloud->stamp || rec.stamp_before != m->src_stamp || n_old != m->n || old.n != m->n || ng <= 0 ||
        n_out != n_old + rec.n_fresh)
        return rebuild(SF_PATCH_NO_MERGE);
    if (n_out >= (int64_t)(1 << 28) || old_cells >= 0xffffffffull) return rebuild(SF_PATCH_LIMITS);

    // 1. do the bounds of the map survive?  (the merge looked: sf_voxel.hip, k_merge_extremes)
    float new_mn[3], new_mx[3];
    for (int d = 0; d < 3; ++d)
        if (rec.old_mn[d] != m->src_mn[d] || rec.old_mx[d] != m->src_mx[d]) return rebuild(SF_PATCH_NO_MERGE); // (not the bounds this index was built on)
    if (rec.touched_extreme) {
        // a point that held a bound was replaced: one reduction over the merged cloud says what the bounds are now (a tenth
        // of a build)
        sf::MinMaxHost mm;
        SF_TRY(sf::cloud_minmax(ctx, cloud->xyz.as<float>(), n_out, &mm));
        if (mm.n_finite != n_out) return rebuild(SF_PATCH_BOUND_REPLACED);
        for (int d = 0; d < 3; ++d) { cloud->bounds_mn[d] = new_mn[d] = mm.mn[d]; cloud->bounds_mx[d] = new_mx[d] = mm.mx[d]; }
        cloud->bounds_stamp = cloud->stamp; // (the next merge need not look again)
    } else {
        for (int d = 0; d < 3; ++d) { new_mn[d] = std::min(m->src_mn[d], rec.cen_mn[d]); new_mx[d] = std::max(m->src_mx[d], rec.cen_mx[d]); }
    }
    for (int d = 0; d < 3; ++d) // the origin a build would choose must be the one the index has: every cell changes otherwise
        if (grid_origin(new_mn[d], m->h_exact, m->origin_lattice) != old.org[d]) return rebuild(SF_PATCH_ORIGIN_MOVED);

    // 2. the geometry a build of the merged cloud would choose (sf_map_build, step 2, explicit cell)
    const double h = m->h_exact;
    size_t free_b = 0, total_b = 0;
    SF_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += m->cell_start.cap; // (the table being replaced counts as free: the build would reuse it)
    const double max_cells = std::min(34359738368.0, std::max(1.0e9, (double)free_b / 4.0 / sizeof(uint32_t)));
    GridGeom g;
    PatchGeom pg;
    double cells = 1;
    for (int d = 0; d < 3; ++d) {
        const double c = std::floor(((double)new_mx[d] - (double)old.org[d]) / h) + 1;
        if (c > 2.0e9) return rebuild(SF_PATCH_LIMITS);
        g.org[d] = pg.org[d] = old.org[d];
        g.dim[d] = pg.dim[d] = (int)c;
        pg.old_dim[d] = old.dim[d];
        cells *= c;
    }
    if (cells > max_cells || cells >= 4294967295.0) return rebuild(SF_PATCH_LIMITS);
    g.inv_h = pg.inv_h = old.inv_h;
    g.ncell = (uint64_t)g.dim[0] * (uint64_t)g.dim[1] * (uint64_t)g.dim[2];
    const bool by_scan = table_by_scan(g.ncell, n_out, m->origin_lattice); // (as sf_map_build)

    // 3. the centroids in (cell, id) order; the replaced entries as a bitmap over the old sorted positions
    const int64_t nb256 = sf::div_up(n_old, 256) + 1;
    const size_t off_key = 0, off_key2 = off_key + 4 * (size_t)ng, off_val = off_key2 + 4 * (size_t)ng, off_val2 = off_val + 4 * (size_t)ng, off_packed = (off_val2 + 4 * (size_t)ng + 7) & ~(size_t)7,
                 off_bitmap = off_packed + 8 * (size_t)ng, off_cnt = off_bitmap + 4 * 8 * (size_t)nb256, off_pre = off_cnt + 4 * (size_t)nb256, off_range = off_pre + 4 * (size_t)nb256,
                 off_ext = (off_range + 8 * (size_t)nb256 + 15) & ~(size_t)15, total = off_ext + sizeof(PatchFlags);
    SF_TRY(m->patch_tmp.reserve(total));
    unsigned char *base = m->patch_tmp.as<unsigned char>();
    uint32_t *ins_key = reinterpret_cast<uint32_t *>(base + off_key), *ins_key2 = reinterpret_cast<uint32_t *>(base + off_key2), *ins_val = reinterpret_cast<uint32_t *>(base + off_val),
             *ins_val2 = reinterpret_cast<uint32_t *>(base + off_val2), *bitmap = reinterpret_cast<uint32_t *>(base + off_bitmap), *blk_cnt = reinterpret_cast<uint32_t *>(base + off_cnt),
             *blk_pre = reinterpret_cast<uint32_t *>(base + off_pre), *range = reinterpret_cast<uint32_t *>(base + off_range);
    uint64_t *packed = reinterpret_cast<uint64_t *>(base + off_packed);
    PatchFlags *d_ext = reinterpret_cast<PatchFlags *>(base + off_ext);
    PatchFlags *h_ext = reinterpret_cast<PatchFlags *>(static_cast<unsigned char *>(ctx->h_pinned) + 256);
    SF_TRY(m->pts4_alt.reserve(sizeof(float4) * (size_t)n_out));
    SF_TRY(m->keys.reserve(sizeof(uint32_t) * (size_t)n_out));
    SF_HIP(hipMemsetAsync(bitmap, 0, off_range - off_bitmap, st)); // bitmap, block counts, their prefix
    SF_HIP(hipMemsetAsync(d_ext, 0, sizeof(PatchFlags), st));
    hipLaunchKernelGGL(k_patch_groups, dim3(nblk(ng)), dim3(256), 0, st, pg, rec.g_rank, rec.g_fresh, rec.g_centroid, rec.g_old, ng, old.pts, n_old, ins_key, ins_val, bitmap, blk_cnt,
                       d_ext);
    unsigned bits = 1;
    while (bits < 32 && (1ull << bits) <= (unsigned long long)g.ncell) ++bits;
    uint32_t *skey = nullptr, *sval = nullptr;
    SF_TRY(sf::radix_sort_pairs<uint32_t>(ctx, ins_key, ins_key2, ins_val, ins_val2, ng, bits, &skey, &sval)); // stable: equal cells stay in ascending id
    hipLaunchKernelGGL(k_patch_pack, dim3(nblk(ng)), dim3(256), 0, st, skey, sval, rec.g_rank, ng, packed);
    SF_TRY(sf::scan_u32<0>(ctx, blk_cnt, blk_pre, nb256));
    hipLaunchKernelGGL(k_patch_ranges, dim3(nblk(2 * (nb256 - 1))), dim3(256), 0, st, pg, old.pts, n_old, packed, ng, nb256 - 1, range);

    // 4. the merge
    float4 *pts_out = m->pts4_alt.as<float4>();
    uint32_t *keys_out = m->keys.as<uint32_t>();
    hipLaunchKernelGGL(k_patch_old, dim3(nblk(n_old)), dim3(256), 0, st, pg, old.pts, n_old, bitmap, blk_pre, packed, range, rec.coarse, rec.fresh_rank, pts_out, keys_out, d_ext);
    hipLaunchKernelGGL(k_patch_new, dim3(nblk(ng)), dim3(256), 0, st, pg, old.pts, n_old, bitmap, blk_pre, skey, sval, ng, rec.g_rank, rec.fresh_pos, rec.g_centroid, pts_out, keys_out);
    SF_HIP(hipMemcpyAsync(h_ext, d_ext, sizeof(PatchFlags), hipMemcpyDeviceToHost, st));

    // 5. the cell table from the merged keys (from here on the old index is gone: an error leaves the map without one)
    m->built = false;
    SF_TRY(m->cell_start.reserve(sizeof(uint32_t) * ((size_t)g.ncell + 8)));
    SF_TRY(build_cell_table(m, g, keys_out, n_out, false, by_scan));
    SF_HIP(hipGetLastError());
    SF_HIP(hipStreamSynchronize(st));
    m->pts4.swap(m->pts4_alt);
    if (h_ext->moved) return rebuild(SF_PATCH_CLAMPED_POINT); // a point that the old grid had clamped to its upper face: its cell, and the order, changed

    m->n = n_out;
    m->built = true;
    m->has_normals = false;
    m->window.kind = 0;
    SfGrid &G = m->grid;
    for (int d = 0; d < 3; ++d) G.dim[d] = g.dim[d];
    G.gap_eps = 1.5f * 2.384186e-7f * (float)std::max(g.dim[0], std::max(g.dim[1], g.dim[2])) * (float)h;
    G.cell_start = m->cell_start.as<uint32_t>() + 1;
    G.pts = m->pts4.as<float4>();
    G.nrm = nullptr;
    G.n = n_out;
    m->has_cov = false;
    m->generation = sf::next_generation();
    for (int d = 0; d < 3; ++d) { m->src_mn[d] = new_mn[d]; m->src_mx[d] = new_mx[d]; }
    m->src_stamp = cloud->stamp;
    if (patched) *patched = 1;
    return SF_OK;
}

// the index as it lies in HBM, for the parity tests: pts4 (x, y, z, bitcast id) in cell order, cell_start[0 .. n_cells]
extern "C" int sf_map_index_info(sf_map *m, int64_t *n_indexed, int64_t *n_cells, float org[3], float *inv_h, float *gap_eps)
{
    SF_CHECK(m && m->built, SF_ERR_STATE, "map not built");
    if (n_indexed) *n_indexed = m->grid.n;
    if (n_cells) *n_cells = (int64_t)m->grid.dim[0] * (int64_t)m->grid.dim[1] * (int64_t)m->grid.dim[2];
    if (org) for (int d = 0; d < 3; ++d) org[d] = m->grid.org[d];
    if (inv_h) *inv_h = m->grid.inv_h;
    if (gap_eps) *gap_eps = m->grid.gap_eps;
    return SF_OK;
}

extern "C" int sf_map_download_index(sf_map *m, float *pts4, int64_t cap_points, uint32_t *cell_start, int64_t cap_cells)
{
    SF_CHECK(m && m->built, SF_ERR_STATE, "map not built");
    sf_ctx *ctx = m->ctx;
    SF_HIP(hipSetDevice(ctx->device));
    const int64_t n = m->grid.n, nc = (int64_t)m->grid.dim[0] * (int64_t)m->grid.dim[1] * (int64_t)m->grid.dim[2] + 1;
    SF_CHECK((!pts4 || cap_points >= n) && (!cell_start || cap_cells >= nc), SF_ERR_INVALID, "buffer too small");
    if (pts4 && n > 0) SF_HIP(hipMemcpyAsync(pts4, m->pts4.p, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (cell_start) SF_HIP(hipMemcpyAsync(cell_start, m->grid.cell_start, sizeof(uint32_t) * (size_t)nc, hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_map_set_origin_lattice(sf_map *m, int cells)
{
    SF_CHECK(m && cells >= 0 && cells <= 4096, SF_ERR_INVALID, "the lattice is 0 (off) to 4 096 cells");
    m->origin_lattice = cells; // takes effect with the next sf_map_build
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
