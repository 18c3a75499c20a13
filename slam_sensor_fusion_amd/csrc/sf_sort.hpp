// sf_sort.hpp — stable LSD radix sort of (key, value) pairs, hand-written for gfx950 (wave64).
//
// Used by the voxel grids (sf_voxel.hip: key = voxel index, value = point id), the grid index of the map (sf_map.hip:
// key = cell id) and PCL's radius-search order of the crops (sf_cloud.hip).  No reference counterpart: the reference
// sorts with std::sort inside pcl::VoxelGrid (global_map_frames_manager.cpp:142-146) and builds a kd-tree
// (icp_point_to_point.cpp:49-55); the oracle restates those.  Stability is part of the contract: points of one voxel /
// one cell stay in ascending point id, which fixes the order of the float32 centroid sums and the tie-breaks of the search.
//
// One pass per digit (at most 8 bits; the width is chosen so that all passes are equally wide), three launches per pass:
//   k_sort_hist     a workgroup counts the digits of its tile of 4096 keys (LDS histogram) -> hist[digit][tile]
//   k_sort_scan     one workgroup per digit: exclusive prefix of its row over the tiles, the row total aside;
//                   the 256 totals are prefixed by the scatter itself (one wave)
//   k_sort_scatter  a workgroup reads its tile once (keys and values stay in registers), ranks it stably and moves it
//                   through LDS into digit order -- the keys, then the values through the same storage --, then writes
//                   runs of equal digits to consecutive addresses.  Tiles are dealt to the XCDs in contiguous eighths.
// Ranking: wave w owns the contiguous quarter w of the tile; in each of its 16 rounds the 64 lanes find the lanes
// with the same digit by one ballot per digit bit (rank = popcount of the lower ones), add the wave's running count of
// that digit (LDS, wave-private: no workgroup barrier inside the loop) and the first lane of each group advances the
// count.  Tile order, quarter order, round order and lane order are all ascending, so equal keys keep their input order.
#pragma once
#include "sf_common.hpp"

namespace sf {

constexpr int SORT_BLK = 256, SORT_ITEMS = 16, SORT_TILE = SORT_BLK * SORT_ITEMS, SORT_WAVES = SORT_BLK / 64;
constexpr int SORT_ITEMS_SMALL = 4;
constexpr int64_t SORT_SMALL_N = 1 << 20;

// ITEMS = keys per thread: SORT_ITEMS for large inputs, SORT_ITEMS_SMALL below SORT_SMALL_N pairs (four times the workgroups
// and a quarter of the ranking rounds each: a sort of 200 k pairs is bound by its few, long workgroups otherwise)
template <class K, int ITEMS>
__global__ __launch_bounds__(SORT_BLK) void k_sort_hist(const K *__restrict__ keys, int64_t n, int shift, uint32_t mask, uint32_t *__restrict__ hist, int ntiles)
{
    constexpr int TILE = SORT_BLK * ITEMS, PER = 16 / (int)sizeof(K); // PER keys per 16-byte load
    __shared__ uint32_t h[SORT_WAVES][256];                          // wave-private counts: a quarter of the collisions
#pragma unroll
    for (int k = 0; k < SORT_WAVES; ++k) h[k][threadIdx.x] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * TILE;
    uint32_t *mine = h[threadIdx.x >> 6];
    if (base + TILE <= n && (reinterpret_cast<uintptr_t>(keys + base) & 15u) == 0u) { // a full, aligned tile: 16 bytes per lane and load
        const uint4 *src = reinterpret_cast<const uint4 *>(keys + base);
        uint4 v[ITEMS / PER];
#pragma unroll
        for (int it = 0; it < ITEMS / PER; ++it) v[it] = src[it * SORT_BLK + (int)threadIdx.x];
#pragma unroll
        for (int it = 0; it < ITEMS / PER; ++it) {
            if (sizeof(K) == 4) {
                atomicAdd(&mine[(v[it].x >> shift) & mask], 1u); atomicAdd(&mine[(v[it].y >> shift) & mask], 1u);
                atomicAdd(&mine[(v[it].z >> shift) & mask], 1u); atomicAdd(&mine[(v[it].w >> shift) & mask], 1u);
            } else {
                const uint64_t k0 = ((uint64_t)v[it].y << 32) | v[it].x, k1 = ((uint64_t)v[it].w << 32) | v[it].z;
                atomicAdd(&mine[(uint32_t)(k0 >> shift) & mask], 1u); atomicAdd(&mine[(uint32_t)(k1 >> shift) & mask], 1u);
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int64_t i = base + it * SORT_BLK + (int)threadIdx.x;
            if (i < n) atomicAdd(&mine[(uint32_t)(keys[i] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x <= mask) hist[(size_t)threadIdx.x * (size_t)ntiles + blockIdx.x] = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
}

// exclusive scan of 256 values held one per thread (the calling workgroup has 256 threads); returns the exclusive prefix,
// *total = the sum
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *sh /* [SORT_WAVES] */, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    uint32_t off = 0, sum = 0;
#pragma unroll
    for (int k = 0; k < SORT_WAVES; ++k) {
        if (k < w) off += sh[k];
        sum += sh[k];
    }
    __syncthreads();
    if (total) *total = sum;
    return off + inc - v;
}

// one workgroup per digit: hist[d][0..ntiles) -> exclusive prefix over the tiles, totals[d] = the row's sum
static __global__ __launch_bounds__(SORT_BLK) void k_sort_scan(uint32_t *__restrict__ hist, int ntiles, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t sh[SORT_WAVES];
    uint32_t *row = hist + (size_t)blockIdx.x * (size_t)ntiles;
    const int per = (ntiles + SORT_BLK - 1) / SORT_BLK;
    const int a = (int)threadIdx.x * per, b = min(a + per, ntiles);
    uint32_t s = 0;
    for (int t = a; t < b; ++t) s += row[t];
    uint32_t tot;
    uint32_t run = block_excl_scan_256(s, sh, &tot);
    for (int t = a; t < b; ++t) {
        const uint32_t v = row[t];
        row[t] = run;
        run += v;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

template <class K, int ITEMS>
__global__ __launch_bounds__(SORT_BLK) void k_sort_scatter(const K *__restrict__ keys, const uint32_t *__restrict__ vals, int64_t n, int shift, uint32_t mask, int bits,
                                                          const uint32_t *__restrict__ hist, const uint32_t *__restrict__ totals, int ntiles, K *__restrict__ keys_out,
                                                          uint32_t *__restrict__ vals_out)
{
    constexpr int TILE = SORT_BLK * ITEMS;
    __shared__ K skeys[TILE];                  // the keys in digit order; then, in the same storage, the values (more workgroups per CU than with both at once)
    __shared__ uint32_t wrun[SORT_WAVES][256]; // first the wave's digit counts, then its running positions
    __shared__ uint32_t goff[256];             // global position of local position 0 of each digit's run (may wrap: uint arithmetic)
    __shared__ uint32_t sh[SORT_WAVES];
    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    // XCD-aware placement: workgroups are dealt to the 8 XCDs in turn, so the ones with equal blockIdx % 8 share an L2.  Each
    // of those groups takes a CONTIGUOUS eighth of the tiles: the runs of one digit written by neighbouring tiles are
    // neighbours in memory, and the partial cache lines at their seams then meet in one L2 before they go out.
    const int q8 = ntiles >> 3, r8 = ntiles & 7, x8 = (int)blockIdx.x & 7;
    const int tile = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + ((int)blockIdx.x >> 3);
    const int64_t base = (int64_t)tile * TILE + (int64_t)w * (TILE / SORT_WAVES);
    // (asked for now, used after the counting: global start of digit d = totals of the smaller digits + this digit's prefix over the earlier tiles)
    const uint32_t tot = ((uint32_t)tid <= mask) ? totals[tid] : 0u;
    const uint32_t before = ((uint32_t)tid <= mask) ? hist[(size_t)tid * (size_t)ntiles + tile] : 0u;
#pragma unroll
    for (int k = 0; k < SORT_WAVES; ++k) wrun[k][tid] = 0u;
    __syncthreads();
    K key[ITEMS];
    uint32_t val[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t i = base + r * 64 + lane;
        const bool in = i < n;
        key[r] = in ? keys[i] : (K)0;
        val[r] = (in && vals) ? vals[i] : 0u;
        if (in) atomicAdd(&wrun[w][(uint32_t)(key[r] >> shift) & mask], 1u);
    }
    __syncthreads();
    // thread d: the tile's count of digit d, where its run starts inside the tile and in the output
    {
        uint32_t c[SORT_WAVES], tile_count = 0;
#pragma unroll
        for (int k = 0; k < SORT_WAVES; ++k) { c[k] = wrun[k][tid]; tile_count += c[k]; }
        const uint32_t dstart = block_excl_scan_256(tile_count, sh, nullptr);
        const uint32_t gbase = block_excl_scan_256(tot, sh, nullptr);
        uint32_t run = dstart;
#pragma unroll
        for (int k = 0; k < SORT_WAVES; ++k) { wrun[k][tid] = run; run += c[k]; }
        goff[tid] = ((uint32_t)tid <= mask) ? gbase + before - dstart : 0u;
    }
    __syncthreads();
    // stable ranking, wave-private
    uint32_t pos[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t i = base + r * 64 + lane;
        const bool in = i < n;
        const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
        unsigned long long peers = __ballot(in);
        for (int b = 0; b < bits; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = peers & ((1ull << lane) - 1ull);
        pos[r] = 0u;
        if (in) pos[r] = wrun[w][d] + (uint32_t)__popcll(below);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (in && below == 0ull) wrun[w][d] += (uint32_t)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (in) skeys[pos[r]] = key[r];
    }
    __syncthreads();
    const int64_t tile_n = min((int64_t)TILE, n - (int64_t)tile * TILE);
    uint32_t o[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int j = it * SORT_BLK + tid;
        o[it] = 0u;
        if (j < tile_n) {
            const K k = skeys[j];
            o[it] = goff[(uint32_t)(k >> shift) & mask] + (uint32_t)j;
            keys_out[o[it]] = k;
        }
    }
    if (!vals_out) return;
    __syncthreads();
    uint32_t *svals = reinterpret_cast<uint32_t *>(skeys);
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
        if (base + r * 64 + lane < n) svals[pos[r]] = val[r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int j = it * SORT_BLK + tid;
        if (j < tile_n) vals_out[o[it]] = svals[j];
    }
}

// Device-wide scans of uint32 arrays (head flags -> output positions; cell ends -> cell starts), three launches:
// per-tile reduction, one workgroup over the tile sums, per-tile scan with the carry.  OP: 0 = sum (exclusive), 1 = max (inclusive).
template <int OP>
__device__ __forceinline__ uint32_t scan_op(uint32_t a, uint32_t b) { return OP == 0 ? a + b : (a > b ? a : b); }

template <int OP>
__global__ __launch_bounds__(SORT_BLK) void k_scan_reduce(const uint32_t *__restrict__ in, int64_t n, uint32_t *__restrict__ tile_sum)
{
    __shared__ uint32_t sh[SORT_WAVES];
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE + (int)threadIdx.x;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k)
        if (base + (int64_t)k * SORT_BLK < n) s = scan_op<OP>(s, in[base + (int64_t)k * SORT_BLK]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s = scan_op<OP>(s, __shfl_xor(s, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = scan_op<OP>(scan_op<OP>(sh[0], sh[1]), scan_op<OP>(sh[2], sh[3]));
}

// one workgroup: tile_sum[t] <- combination of the tiles before t (exclusive), sequential over chunks of 256
template <int OP>
__global__ __launch_bounds__(SORT_BLK) void k_scan_tiles(uint32_t *__restrict__ tile_sum, int ntiles, uint32_t carry0)
{
    __shared__ uint32_t sh[SORT_BLK];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = carry0;
    __syncthreads();
    for (int b0 = 0; b0 < ntiles; b0 += SORT_BLK) {
        const int t = b0 + (int)threadIdx.x;
        const uint32_t v = t < ntiles ? tile_sum[t] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < SORT_BLK; o <<= 1) { // Hillis-Steele, inclusive
            const uint32_t a = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] = scan_op<OP>(sh[threadIdx.x], a);
            __syncthreads();
        }
        const uint32_t c = carry;
        const uint32_t excl = threadIdx.x > 0 ? scan_op<OP>(c, sh[threadIdx.x - 1]) : c;
        if (t < ntiles) tile_sum[t] = excl;
        __syncthreads();
        if (threadIdx.x == SORT_BLK - 1) carry = scan_op<OP>(c, sh[SORT_BLK - 1]);
        __syncthreads();
    }
}

// wave w owns the contiguous quarter w of the tile and walks it in 16 rounds of 64 consecutive elements (coalesced both
// ways): a wave-level inclusive scan per round, the carry handed from round to round; the quarters' totals meet in LDS
template <int OP>
__global__ __launch_bounds__(SORT_BLK) void k_scan_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int64_t n, const uint32_t *__restrict__ tile_sum)
{
    __shared__ uint32_t sh[SORT_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * (SORT_TILE / SORT_WAVES) + lane;
    uint32_t v[SORT_ITEMS], inc[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) v[r] = base + r * 64 < n ? in[base + r * 64] : 0u;
    uint32_t carry = 0; // combination of the wave's earlier rounds
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        uint32_t x = v[r];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(x, o, 64);
            if (lane >= o) x = scan_op<OP>(x, t);
        }
        inc[r] = scan_op<OP>(carry, x);
        carry = __shfl(inc[r], 63, 64);
    }
    if (lane == 0) sh[w] = carry;
    __syncthreads();
    uint32_t off = tile_sum[blockIdx.x];
    for (int k = 0; k < w; ++k) off = scan_op<OP>(off, sh[k]);
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        if (base + r * 64 < n) {
            if (OP == 0) out[base + r * 64] = off + inc[r] - v[r]; // exclusive
            else out[base + r * 64] = scan_op<OP>(off, inc[r]);
        }
    }
}

// out[i] = sum of in[0..i) (OP 0) or max(carry0, in[0..i]) (OP 1); in == out allowed.  Enqueue only.
template <int OP>
int scan_u32(sf_ctx *ctx, const uint32_t *in, uint32_t *out, int64_t n, uint32_t carry0 = 0u)
{
    if (n <= 0) return SF_OK;
    const int ntiles = (int)div_up(n, SORT_TILE);
    SF_TRY(ctx->scan_tiles.reserve(sizeof(uint32_t) * (size_t)ntiles));
    uint32_t *ts = ctx->scan_tiles.as<uint32_t>();
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL((k_scan_reduce<OP>), dim3((unsigned)ntiles), dim3(SORT_BLK), 0, st, in, n, ts);
    hipLaunchKernelGGL((k_scan_tiles<OP>), dim3(1), dim3(SORT_BLK), 0, st, ts, ntiles, carry0);
    hipLaunchKernelGGL((k_scan_apply<OP>), dim3((unsigned)ntiles), dim3(SORT_BLK), 0, st, in, out, n, ts);
    SF_HIP(hipGetLastError());
    return SF_OK;
}

// Sorts n pairs by the low `end_bit` bits of the key, stable.  vals / vals_alt may both be NULL (keys only).  keys / vals and keys_alt / vals_alt are ping-pong buffers of n
// elements; *keys_sorted / *vals_sorted point at the buffers that hold the result.  Enqueue only (context stream).
template <class K>
int radix_sort_pairs(sf_ctx *ctx, K *keys, K *keys_alt, uint32_t *vals, uint32_t *vals_alt, int64_t n, unsigned end_bit, K **keys_sorted, uint32_t **vals_sorted)
{
    *keys_sorted = keys;
    *vals_sorted = vals;
    if (n <= 1 || end_bit == 0) return SF_OK;
    SF_CHECK(n < (int64_t)0xffffffffll, SF_ERR_OVERFLOW, "radix sort: more than 2^32 - 1 elements");
    const int passes = (int)((end_bit + 7) / 8), bits = (int)((end_bit + passes - 1) / passes);
    const bool small = n < SORT_SMALL_N;
    const int ntiles = (int)div_up(n, small ? SORT_BLK * SORT_ITEMS_SMALL : SORT_TILE);
    const uint32_t mask = (1u << bits) - 1u;
    SF_TRY(ctx->sort_hist.reserve(sizeof(uint32_t) * ((size_t)(mask + 1) * (size_t)ntiles + 256)));
    uint32_t *hist = ctx->sort_hist.as<uint32_t>(), *totals = hist + (size_t)(mask + 1) * (size_t)ntiles;
    hipStream_t st = ctx->stream;
    K *ka = keys, *kb = keys_alt;
    uint32_t *va = vals, *vb = vals_alt;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * bits;
        if (small) hipLaunchKernelGGL((k_sort_hist<K, SORT_ITEMS_SMALL>), dim3((unsigned)ntiles), dim3(SORT_BLK), 0, st, ka, n, shift, mask, hist, ntiles);
        else hipLaunchKernelGGL((k_sort_hist<K, SORT_ITEMS>), dim3((unsigned)ntiles), dim3(SORT_BLK), 0, st, ka, n, shift, mask, hist, ntiles);
        hipLaunchKernelGGL(k_sort_scan, dim3(mask + 1), dim3(SORT_BLK), 0, st, hist, ntiles, totals);
        if (small) hipLaunchKernelGGL((k_sort_scatter<K, SORT_ITEMS_SMALL>), dim3((unsigned)ntiles), dim3(SORT_BLK), 0, st, ka, va, n, shift, mask, bits, hist, totals, ntiles, kb, vb);
        else hipLaunchKernelGGL((k_sort_scatter<K, SORT_ITEMS>), dim3((unsigned)ntiles), dim3(SORT_BLK), 0, st, ka, va, n, shift, mask, bits, hist, totals, ntiles, kb, vb);
        K *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    SF_HIP(hipGetLastError());
    *keys_sorted = ka;
    *vals_sorted = va;
    return SF_OK;
}

} // namespace sf
