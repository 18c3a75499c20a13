// sf_order.hpp — query ordering: each scan's points bucketed by the map cell they fall in under the initial pose
// (gfx950, hand-written; replaces the rocPRIM radix_sort_pairs + key kernels of round 1 in the timed path).
//
// What is ordered: per scan (segment) a 10-bit key = position of the query's cell in the walk order of
// order_cell() shifted down to 10 bits, i.e. 1024 buckets per scan; inside a bucket the queries keep their
// original order (stable).  Measured on MI355X (200 k-point scans vs the 10 M-point map, 32 in flight, k_nn_red per
// launch): no order 194 us, 4 key bits 180, 6 bits 159, 8 bits 156, 10 bits 146, 12 bits 148, 16 bits 143, 20 bits
// 143 -- what the search needs is that chunk c of every scan in flight covers the same stretch of the map, not a
// total order, so ONE stable counting pass on 10 bits replaces round 1's four rocPRIM radix passes over a 24-bit
// key (and a two-pass 20-bit version of this file, which cost 133 us more per step than the 2 % it bought).
// Scans are independent segments: the order of a scan does not depend on what else is in the batch.
//
//   k_order_hist     per-tile bucket counts (LDS histogram, keys computed from the coordinates, the key of every
//                    query kept as a u16 so the scatter does not recompute it)
//   k_order_scan     per segment: exclusive prefix over tiles and over buckets, in place
//   k_order_scatter  rank inside the tile + prefix -> the query's id at its ordered position
//   k_order_gather   (sf_icp.hip) ids read coalesced, float4 records gathered, cell-ordered SoA written coalesced
//
// Stability inside a tile without a per-element sequential loop: a tile is split into contiguous quarters, one per
// wave; each wave keeps its own running counter per bucket in LDS (started at the tile's start for that bucket +
// what the waves before it hold) and walks its quarter 64 elements at a time: the lanes whose bucket equals mine are
// found with one ballot per key bit (10 ballots), my rank among them is the popcount below my lane, and the last of
// them advances the counter.  All counts are integers: the result is the unique stable order, bitwise reproducible.
#pragma once
#include "sf_common.hpp"

namespace sf {

constexpr int ORD_KEY_BITS = 10;
constexpr int ORD_BINS = 1 << ORD_KEY_BITS;
constexpr uint32_t ORD_KEY_NONE = ORD_BINS - 1u;  // non-finite queries: last bucket of their scan
constexpr int ORD_BLK = 256;
constexpr int ORD_WAVES = ORD_BLK / 64;
constexpr int ORD_PER_LANE = 16;                  // elements per lane
constexpr int ORD_TILE = ORD_BLK * ORD_PER_LANE;  // 4096 elements per workgroup
constexpr int ORD_WAVE_SPAN = 64 * ORD_PER_LANE;  // contiguous elements per wave

struct OrderSrc {
    const uint32_t *src_idx;     // element e of the segment space -> global query id, or nullptr (identity)
    const uint32_t *seg_off;     // segment offsets [nseg + 1], or nullptr (uniform: segment b = [b * n, (b + 1) * n))
    int n;                       // uniform segment length
    int tiles;                   // tiles per segment
    int nseg;                    // segments (grid: order_grid(tiles, nseg), see order_block)
};

// XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs in launch order (linear id L runs on XCD L % 8).
// A one-dimensional grid of tiles x (segments rounded up to 8): all tiles of segment b run on XCD b % 8, eight segments
// at a time, so what a segment's workgroups write at random inside the segment (the ordered ids: 4-byte stores into
// 0.8 MB per 200 k-point scan) is merged into whole lines in ONE L2 and what they read at random (k_order_gather: 16-byte
// records, 3.2 MB per scan) is fetched into one L2 once -- a (tiles, segments) grid spreads every segment over the
// eight L2s, which do not share lines.  -> false: no such segment (padding)
__device__ __forceinline__ bool order_block(int tiles, int nseg, int *b, int *tile)
{
    const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
    *tile = k % tiles;
    *b = (k / tiles) * 8 + xcd;
    return *b < nseg;
}
inline unsigned order_grid(int tiles, int nseg) { return (unsigned)tiles * (unsigned)((nseg + 7) & ~7); }

__device__ __forceinline__ void order_segment(const OrderSrc &s, int b, uint32_t *start, uint32_t *len)
{
    if (s.seg_off) { *start = s.seg_off[b]; *len = s.seg_off[b + 1] - s.seg_off[b]; }
    else { *start = (uint32_t)b * (uint32_t)s.n; *len = (uint32_t)s.n; }
}

// the lanes of the wave whose key equals mine (only lanes with valid = true take part)
__device__ __forceinline__ unsigned long long order_peers(uint32_t key, bool valid)
{
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int k = 0; k < ORD_KEY_BITS; ++k) {
        const unsigned long long bk = __ballot((key >> k) & 1u);
        m &= ((key >> k) & 1u) ? bk : ~bk;
    }
    return m;
}

// element index (inside the segment) of chunk c of this lane: a wave owns ORD_WAVE_SPAN consecutive elements and
// walks them 64 at a time
__device__ __forceinline__ uint32_t order_element(int tile, int c)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    return (uint32_t)tile * ORD_TILE + (uint32_t)wv * ORD_WAVE_SPAN + 64u * (uint32_t)c + (uint32_t)lane;
}

// counts[(seg * (tiles + 1) + tile) * ORD_BINS + key]; keys[segment space] <- KEYFN(global query id)
template <class KEYFN>
__device__ __forceinline__ void order_hist_body(const OrderSrc &s, KEYFN keyfn, uint16_t *__restrict__ keys, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t h[ORD_BINS];
    int b, tile;
    if (!order_block(s.tiles, s.nseg, &b, &tile)) return;
    uint32_t seg_start, seg_len;
    order_segment(s, b, &seg_start, &seg_len);
    uint32_t *dst = counts + ((size_t)b * (s.tiles + 1) + tile) * ORD_BINS; // row `tiles` of a segment's table: the bucket starts (k_order_scan)
    if ((uint32_t)tile * ORD_TILE >= seg_len) { // nothing of this segment here (shorter segments of a ragged batch)
        for (int d = threadIdx.x; d < ORD_BINS; d += ORD_BLK) dst[d] = 0;
        return;
    }
    for (int d = threadIdx.x; d < ORD_BINS; d += ORD_BLK) h[d] = 0;
    __syncthreads();
    // every load of the lane first (KEYFN::load: 16 x 3 coordinate loads in flight), then the arithmetic and the stores
    const typename KEYFN::Pose pose = keyfn.prepare(b); // segment b is scan b
    typename KEYFN::Point pt[ORD_PER_LANE];
#pragma unroll
    for (int c = 0; c < ORD_PER_LANE; ++c) {
        const uint32_t e = order_element(tile, c);
        const uint32_t o = e < seg_len ? (s.src_idx ? s.src_idx[seg_start + e] : seg_start + e) : seg_start;
        pt[c] = keyfn.load(o);
    }
#pragma unroll
    for (int c = 0; c < ORD_PER_LANE; ++c) {
        const uint32_t e = order_element(tile, c);
        if (e < seg_len) {
            const uint32_t key = keyfn.key(pose, pt[c]);
            keys[seg_start + e] = (uint16_t)key;
            atomicAdd(&h[key], 1u); // integer: order independent
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < ORD_BINS; d += ORD_BLK) dst[d] = h[d];
}

// one workgroup of ORD_BINS threads per segment: counts -> exclusive start of (bucket, tile) within the segment
__global__ __launch_bounds__(ORD_BINS) void k_order_scan(uint32_t *__restrict__ counts, int tiles)
{
    __shared__ uint32_t sc[ORD_BINS];
    const int d = threadIdx.x;
    uint32_t *c = counts + (size_t)blockIdx.x * (tiles + 1) * ORD_BINS;
    uint32_t run = 0;
    for (int t0 = 0; t0 < tiles; t0 += 8) { // exclusive prefix over the tiles for this bucket (coalesced across buckets), 8 loads in flight
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t0 + k < tiles ? c[(size_t)(t0 + k) * ORD_BINS + d] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (t0 + k < tiles) c[(size_t)(t0 + k) * ORD_BINS + d] = run;
            run += v[k];
        }
    }
    sc[d] = run; // total of bucket d
    __syncthreads();
    for (int off = 1; off < ORD_BINS; off <<= 1) { // inclusive scan over the buckets
        const uint32_t t = d >= off ? sc[d - off] : 0u;
        __syncthreads();
        sc[d] += t;
        __syncthreads();
    }
    sc[d] -= run; // exclusive: where bucket d starts in the segment
    __syncthreads();
    // kept apart from the per-tile prefixes (the scatter adds the two): row `tiles` of the table
    c[(size_t)tiles * ORD_BINS + d] = sc[d];
}

// out[seg_start + rank] <- global query id
__device__ __forceinline__ void order_scatter_body(const OrderSrc &s, const uint16_t *__restrict__ keys, const uint32_t *__restrict__ starts, uint32_t *__restrict__ out)
{
    __shared__ uint32_t ctr[ORD_WAVES][ORD_BINS];
    int b, tile;
    if (!order_block(s.tiles, s.nseg, &b, &tile)) return;
    uint32_t seg_start, seg_len;
    order_segment(s, b, &seg_start, &seg_len);
    if ((uint32_t)tile * ORD_TILE >= seg_len) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int d = threadIdx.x; d < ORD_BINS * ORD_WAVES; d += ORD_BLK) (&ctr[0][0])[d] = 0;
    __syncthreads();
    uint32_t key[ORD_PER_LANE];
    bool valid[ORD_PER_LANE];
#pragma unroll
    for (int c = 0; c < ORD_PER_LANE; ++c) {
        const uint32_t e = order_element(tile, c);
        valid[c] = e < seg_len;
        key[c] = valid[c] ? (uint32_t)keys[seg_start + e] : 0u;
    }
#pragma unroll
    for (int c = 0; c < ORD_PER_LANE; ++c)
        if (valid[c]) atomicAdd(&ctr[wv][key[c]], 1u);
    __syncthreads();
    // per-wave counts -> per-wave running start: bucket start + tile prefix + what the waves before hold
    const uint32_t *seg_tab = starts + (size_t)b * (s.tiles + 1) * ORD_BINS;
    for (int d = threadIdx.x; d < ORD_BINS; d += ORD_BLK) {
        uint32_t run = seg_tab[(size_t)tile * ORD_BINS + d] + seg_tab[(size_t)s.tiles * ORD_BINS + d];
#pragma unroll
        for (int w = 0; w < ORD_WAVES; ++w) {
            const uint32_t v = ctr[w][d];
            ctr[w][d] = run;
            run += v;
        }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t rank[ORD_PER_LANE];
#pragma unroll
    for (int c = 0; c < ORD_PER_LANE; ++c) { // the wave's chunks in order: stability.  LDS only in this loop: the global stores follow
        const unsigned long long peers = order_peers(key[c], valid[c]);
        rank[c] = valid[c] ? ctr[wv][key[c]] + (uint32_t)__popcll(peers & below) : 0u;
        __builtin_amdgcn_wave_barrier(); // every lane has read the counter before the group's last lane advances it
        if (valid[c] && (peers >> lane) <= 1ull) ctr[wv][key[c]] += (uint32_t)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t id[ORD_PER_LANE];
#pragma unroll
    for (int c = 0; c < ORD_PER_LANE; ++c) {
        const uint32_t e = order_element(tile, c);
        id[c] = valid[c] ? (s.src_idx ? s.src_idx[seg_start + e] : seg_start + e) : 0u;
    }
#pragma unroll
    for (int c = 0; c < ORD_PER_LANE; ++c)
        if (valid[c]) out[seg_start + rank[c]] = id[c];
}

} // namespace sf
