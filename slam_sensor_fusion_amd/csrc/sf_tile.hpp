// sf_tile.hpp — the neighbour search out of LDS: map tiles staged once per workgroup, every query of every scan in
// flight that falls into the tile served from there (device side, gfx950).
//
// Why (measured, round 4: s_memtime stamps per phase of k_nn_red, tools/phase_trace.py; profiles/LADDER.md): the
// search that walks the grid index in global memory is a chain of ~10 dependent round trips per wave, each 4-5
// thousand cycles long on a loaded chip because every candidate is a 16-byte per-lane access the L1 has to look up
// (~670 line accesses per 64 queries, 62 % of them L1 misses): the kernel waits on the texture path, neither on HBM nor
// on the ALUs.  Cutting instructions (a chunk queue with a selection network: same accesses) bought nothing.  What
// removes the accesses is locality the batch already has: with 64 scans in flight every map cell is visited by two
// queries per launch.  So the map is cut into TILES (a box of cells plus a halo of HALO cells), the queries of all
// scans are sorted by tile once per alignment (sf_order.hpp, two 10-bit passes), and one workgroup per tile
//   1. copies the tile's points (contiguous row runs of the cell-sorted map: coalesced) and a 16-bit local cell table
//      into LDS,
//   2. takes the tile's queries of every scan: transform, reuse certificate, and for those that must search the exact
//      1-NN over the 27 cells around the query's cell -- own cell, then the neighbouring cells and rows while their gap
//      is below the best so far, four candidates per trip through a min3 / med3 selection network -- all of it LDS reads,
//   3. writes the pair (neighbour, index, normal, runner-up bound) into the neighbour cache; the records are summed
//      from there by k_red_cached in the order k_nn_red would have used (bit-identical sums).
// A query is served from LDS while the 27 cells around its CURRENT cell lie inside the staged region: with HALO = 2 that
// holds until it has moved a cell (0.25 m) away from where it was binned; beyond, or when ring 1 does not settle it,
// the lane walks the global index (nn_rings) -- same exact result, counted in the statistics.
// Exactness and tie rule are those of sf_nn.hpp: lexicographic (d2, sorted position) minimum, strict "<" against the
// threshold, distances in FLANN's L2_Simple order, pruning only by gaps that cannot hold a better candidate.
#pragma once
#include "sf_nn.hpp"

namespace sf {

constexpr int TILE_HALO = 2;
constexpr int TILE_PCAP = 2304;    // points a staged region may hold (a tile beyond it is searched through the global index)
constexpr int TILE_RX_MAX = 16;    // region cells along x
constexpr int TILE_ROWS_MAX = 160; // region rows (y, z)
constexpr int TILE_BLK = 256;
constexpr int TILE_SCANS = 64;     // scans whose poses are staged together (larger batches: in groups)

struct SfTiles {
    int tc[3];   // core cells per tile along x, y, z
    int nt[3];   // tiles per axis
    int ntiles;
};

// tile of a cell, x fastest
__device__ __forceinline__ uint32_t tile_of_cell(const SfTiles &tl, int cx, int cy, int cz)
{
    return ((uint32_t)(cz / tl.tc[2]) * (uint32_t)tl.nt[1] + (uint32_t)(cy / tl.tc[1])) * (uint32_t)tl.nt[0] + (uint32_t)(cx / tl.tc[0]);
}

struct TileRegion { // wave-uniform
    int x0, y0, z0; // first cell of the staged region
    int rx, ry, rz; // its extent in cells
};

__device__ __forceinline__ TileRegion tile_region(const SfGrid &g, const SfTiles &tl, uint32_t tile)
{
    const int tx = (int)(tile % (uint32_t)tl.nt[0]), ty = (int)((tile / (uint32_t)tl.nt[0]) % (uint32_t)tl.nt[1]), tz = (int)(tile / ((uint32_t)tl.nt[0] * (uint32_t)tl.nt[1]));
    TileRegion R;
    R.x0 = max(tx * tl.tc[0] - TILE_HALO, 0);
    R.y0 = max(ty * tl.tc[1] - TILE_HALO, 0);
    R.z0 = max(tz * tl.tc[2] - TILE_HALO, 0);
    R.rx = min((tx + 1) * tl.tc[0] + TILE_HALO, g.dim[0]) - R.x0;
    R.ry = min((ty + 1) * tl.tc[1] + TILE_HALO, g.dim[1]) - R.y0;
    R.rz = min((tz + 1) * tl.tc[2] + TILE_HALO, g.dim[2]) - R.z0;
    return R;
}

struct TileLds {
    float4 pts[TILE_PCAP + 4];                          // x, y, z, bitcast(sorted position in the map)
    uint16_t lcs[TILE_ROWS_MAX * (TILE_RX_MAX + 1)];    // row r, cell x: first local point of the cell; entry rx: the row's end
    uint32_t row_a[TILE_ROWS_MAX];                      // staging: the row's first sorted position
    uint32_t row_base[TILE_ROWS_MAX + 1];               // staging: the row's first local point
    int overflow;
};

// every thread of the workgroup; ends with a barrier.  -> false: the region does not fit (the tile's queries walk the global index)
__device__ __forceinline__ bool tile_stage(const SfGrid &g, const TileRegion &R, TileLds *L)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nrows = R.ry * R.rz, rxp = R.rx + 1;
    for (int r = tid; r < nrows; r += TILE_BLK) {
        const int y = R.y0 + r % R.ry, z = R.z0 + r / R.ry;
        const size_t c0 = ((size_t)z * g.dim[1] + y) * g.dim[0] + R.x0;
        const uint32_t a = g.cell_start[c0], e = g.cell_start[c0 + R.rx];
        L->row_a[r] = a;
        L->row_base[r + 1] = e - a; // lengths first, the prefix below
    }
    __syncthreads();
    if (wv == 0) { // exclusive prefix over the rows: 64 at a time
        uint32_t carry = 0;
        for (int r0 = 0; r0 < nrows; r0 += 64) {
            const int r = r0 + lane;
            uint32_t v = r < nrows ? L->row_base[r + 1] : 0u;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)v, o);
                if (lane >= o) v += t;
            }
            if (r < nrows) L->row_base[r + 1] = carry + v;
            carry += (uint32_t)__shfl((int)v, 63);
        }
        if (lane == 0) {
            L->row_base[0] = 0;
            L->overflow = carry > (uint32_t)TILE_PCAP ? 1 : 0;
        }
    }
    __syncthreads();
    const bool fits = L->overflow == 0;
    if (fits) {
        // points: four rows per wave step (16 lanes each: a row of the region holds ~18 points at the metric density)
        const int sub = lane >> 4, sl = lane & 15;
        for (int r = wv * 4 + sub; r < nrows; r += 4 * (TILE_BLK / 64)) {
            const uint32_t a = L->row_a[r], base = L->row_base[r], len = L->row_base[r + 1] - base;
            for (uint32_t k = (uint32_t)sl; k < len; k += 16u) {
                const float4 p = g.pts[a + k];
                L->pts[base + k] = make_float4(p.x, p.y, p.z, __uint_as_float(a + k));
            }
        }
        // local cell table
        for (int i = tid; i < nrows * rxp; i += TILE_BLK) {
            const int r = i / rxp, x = i - r * rxp;
            const int y = R.y0 + r % R.ry, z = R.z0 + r / R.ry;
            const size_t c0 = ((size_t)z * g.dim[1] + y) * g.dim[0] + R.x0;
            L->lcs[r * (TILE_RX_MAX + 1) + x] = (uint16_t)(L->row_base[r] + (g.cell_start[c0 + x] - L->row_a[r]));
        }
    }
    __syncthreads();
    return fits;
}

struct TileHit {
    float d2;   // squared distance of the best candidate (the largest float below the threshold if none)
    int j;      // its sorted position in the map, -1 if none
    int loc;    // its position in the staged region (-1: none found there -- the seed, if any, stands)
    float lb2;  // every OTHER map point is at least sqrt(lb2) away
};

// local candidates [a, min(a + 4, b)) through the selection network of sf_nn.hpp's chunk4: nearest (lowest sorted position
// among equals) and runner-up of the four in four instructions, one 64-bit compare against the best so far
__device__ __forceinline__ void tile_chunk4(const TileLds *L, uint32_t a, uint32_t b, float qx, float qy, float qz, TileHit &hit)
{
    constexpr float BIG = 3.0e38f;
    const bool v1 = a + 1 < b, v2 = a + 2 < b, v3 = a + 3 < b;
    const float4 p0 = L->pts[a], p1 = L->pts[a + 1], p2 = L->pts[a + 2], p3 = L->pts[a + 3]; // (the array carries four entries of slack)
    const float d0 = l2_simple(qx, qy, qz, p0.x, p0.y, p0.z);
    const float d1 = v1 ? l2_simple(qx, qy, qz, p1.x, p1.y, p1.z) : BIG;
    const float d2 = v2 ? l2_simple(qx, qy, qz, p2.x, p2.y, p2.z) : BIG;
    const float d3 = v3 ? l2_simple(qx, qy, qz, p3.x, p3.y, p3.z) : BIG;
    const float m3 = min3f(d0, d1, d2);
    const float m1 = fminf(m3, d3);
    const float m2 = __builtin_amdgcn_fmed3f(m3, __builtin_amdgcn_fmed3f(d0, d1, d2), d3); // second smallest of the four
    const int k1 = d0 == m1 ? 0 : (d1 == m1 ? 1 : (d2 == m1 ? 2 : 3));
    const int j1 = (int)a + k1 + (__float_as_int(p0.w) - (int)a); // candidates of one range are consecutive in the map: sorted position = local + the range's offset
    const unsigned long long mine = hit_key(m1, j1), cur = hit_key(hit.d2, hit.j);
    const bool better = mine < cur, same = mine == cur;
    const float r = better ? hit.d2 : (same ? BIG : m1); // a displaced best or a losing nearest is a runner-up; the best itself met again is not
    hit.lb2 = min3f(hit.lb2, m2, r);
    if (better) { hit.d2 = m1; hit.j = j1; hit.loc = (int)a + k1; }
}

__device__ __forceinline__ void tile_scan(const TileLds *L, uint32_t a, uint32_t b, float qx, float qy, float qz, TileHit &hit)
{
    for (uint32_t j = a; j < b; j += 4u) tile_chunk4(L, j, b, qx, qy, qz, hit);
}

// is the query's ring-1 neighbourhood inside the staged region?
__device__ __forceinline__ bool tile_serves(const SfGrid &g, const TileRegion &R, const QueryGeo &G)
{
    return max(G.cx - 1, 0) >= R.x0 && min(G.cx + 1, g.dim[0] - 1) < R.x0 + R.rx && max(G.cy - 1, 0) >= R.y0 && min(G.cy + 1, g.dim[1] - 1) < R.y0 + R.ry &&
           max(G.cz - 1, 0) >= R.z0 && min(G.cz + 1, g.dim[2] - 1) < R.z0 + R.rz;
}

// exact ring-1 search of one query over the staged region (tile_serves holds).  seed: as nn_search_wave.
// -> hit (lb2 includes the boundary of the 27 cells); *more: ring 1 does not settle the query (go on with nn_rings from ring 2)
__device__ __forceinline__ TileHit tile_search(const SfGrid &g, const TileRegion &R, const TileLds *L, const QueryGeo &G, float qx, float qy, float qz, float thr, float seed_d2,
                                               int seed_j, bool *more)
{
    TileHit hit;
    hit.d2 = search_start(thr);
    hit.j = -1;
    hit.loc = -1;
    hit.lb2 = 3.0e38f;
    if (seed_j >= 0 && seed_d2 < thr) { hit.d2 = seed_d2; hit.j = seed_j; }
    const int lx = G.cx - R.x0, ly = G.cy - R.y0, lz = G.cz - R.z0;
    const int xa = max(lx - 1, 0), xd = min(lx + 2, R.rx); // clipped at the grid border: the missing neighbour cell reads as an empty range
    const int nyg = g.dim[1], nzg = g.dim[2];
    {
        const uint16_t *row = L->lcs + (lz * R.ry + ly) * (TILE_RX_MAX + 1);
        const uint32_t s0 = row[xa], s1 = row[lx], s2 = row[lx + 1], s3 = row[xd];
        tile_scan(L, s1, s2, qx, qy, qz, hit);
        if (s0 < s1) {
            if (G.gxm2 * 0.998f < hit.d2) tile_scan(L, s0, s1, qx, qy, qz, hit);
            else hit.lb2 = fminf(hit.lb2, G.gxm2);
        }
        if (s2 < s3) {
            if (G.gxp2 * 0.998f < hit.d2) tile_scan(L, s2, s3, qx, qy, qz, hit);
            else hit.lb2 = fminf(hit.lb2, G.gxp2);
        }
    }
    for (int k = 0; k < 8; ++k) {
        const int dy = row_dy(k), dz = row_dz(k);
        if (!((unsigned)(G.cy + dy) < (unsigned)nyg && (unsigned)(G.cz + dz) < (unsigned)nzg)) continue;
        const float g2 = row_gap2(G, k);
        if (!(g2 * 0.998f < hit.d2)) { hit.lb2 = fminf(hit.lb2, g2); continue; }
        const uint16_t *row = L->lcs + ((lz + dz) * R.ry + (ly + dy)) * (TILE_RX_MAX + 1);
        const uint32_t s0 = row[xa], s1 = row[lx], s2 = row[lx + 1], s3 = row[xd];
        const float gm = g2 + G.gxm2, gp = g2 + G.gxp2;
        const bool xm = gm * 0.998f < hit.d2, xp = gp * 0.998f < hit.d2;
        if (!xm && s0 < s1) hit.lb2 = fminf(hit.lb2, gm);
        if (!xp && s2 < s3) hit.lb2 = fminf(hit.lb2, gp);
        tile_scan(L, xm ? s0 : s1, xp ? s3 : s2, qx, qy, qz, hit);
    }
    // exactness of ring 1 (the test of nn_search_wave): the nearest face of the 27 cells that still has grid cells behind it
    const int nx = g.dim[0];
    const float gx = (qx - g.org[0]) * g.inv_h, gy = (qy - g.org[1]) * g.inv_h, gz = (qz - g.org[2]) * g.inv_h;
    const float h = g.h;
    float mface = 3.0e38f;
    if (G.cx - 1 > 0) mface = fminf(mface, (gx - (float)(G.cx - 1)) * h);
    if (G.cx + 1 < nx - 1) mface = fminf(mface, ((float)(G.cx + 2) - gx) * h);
    if (G.cy - 1 > 0) mface = fminf(mface, (gy - (float)(G.cy - 1)) * h);
    if (G.cy + 1 < nyg - 1) mface = fminf(mface, ((float)(G.cy + 2) - gy) * h);
    if (G.cz - 1 > 0) mface = fminf(mface, (gz - (float)(G.cz - 1)) * h);
    if (G.cz + 1 < nzg - 1) mface = fminf(mface, ((float)(G.cz + 2) - gz) * h);
    const float mm = safe_gap(mface, g.gap_eps) * 0.999f;
    *more = false;
    if (mface < 3.0e38f) {
        hit.lb2 = fminf(hit.lb2, mm * mm);
        *more = !(hit.d2 <= mm * mm);
    }
    return hit;
}

} // namespace sf
