// sf_nn.hpp — exact 1-NN over the uniform-grid map index (device side, gfx950).
//
// Replaces the N serial FLANN kd-tree descents of
// localization/src/icp_point_to_point.cpp:64-69 (sourceTargetCorrespondences) and the
// Open3D hybrid search behind localization_python/.../localization_node.py:233-237.
// One lane per query.  Two forms: nn_search (each lane on its own; map queries, REF_CPP mode,
// the brute-force scorer) and nn_search_wave (the 64 lanes of a wave share the work; the ICP hot
// kernel k_nn_red).
//
// Layout: the map is sorted by cell (x fastest), so a row of cells is one contiguous candidate
// range, and ONE 16-byte look-up returns the bounds of the three cells of a row.  The query's own
// cell is scanned first; its x neighbours and the 8 neighbouring rows only while their gap to the
// query is smaller than the best distance so far.
// What was measured on MI355X and shaped this file (profiles/): a branch per
// candidate load serialises the loads into one round trip each; clamped duplicate loads keep them
// in flight together but every duplicate is an L1 access, and the L1 / texture path is what the
// kernel runs out of -- buffer addressing with an out-of-range offset gives predication without
// either cost; the winner's coordinates stay in registers (re-reading them costs a round trip);
// a per-cell 128-byte bucket layout and 32-byte point+normal records were both slower; after
// the own-cell step only ~6 of 64 lanes have neighbour ranges left to scan, which is why
// nn_search_wave deals that work out across the wave.
//
// Exactness: after scanning the block of radius R around the query's cell, every
// unscanned point is at least m = distance(query, block boundary) away; the search stops
// when best <= m^2 (best starts at the acceptance threshold, so "nothing acceptable
// outside" ends it too), otherwise the block grows by one ring.  Cells, rows and rings are only
// skipped when their distance from the query -- computed in float32 grid coordinates, minus
// SfGrid::gap_eps for the rounding of those coordinates (a point and a query are binned by the
// same monotone float expression, so the true face distance can be shorter than the computed
// one by at most 2^-22 x the largest cell coordinate), times 0.998 -- is not below the best.
// Distance = FLANN L2_Simple in float32: ((dx*dx) + dy*dy) + dz*dz, unfused; strict "<".
#pragma once
#include "sf_common.hpp"

namespace sf {

// SF_PHASE_TRACE (diagnostic build only, tools/phase_trace.py): s_memtime stamps at the phase boundaries of a searching
// wave, summed per phase over the launch.  No stamp executes in the product build.
#ifdef SF_PHASE_TRACE
constexpr int PH_SLOTS = 16, PH_SHARDS = 64;
struct PhaseClock {
    unsigned long long t;
    unsigned acc[PH_SLOTS];
    __device__ __forceinline__ void start() { for (int i = 0; i < PH_SLOTS; ++i) acc[i] = 0; t = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int i) { const unsigned long long n = __builtin_amdgcn_s_memtime(); acc[i] += (unsigned)(n - t); t = n; }
    __device__ __forceinline__ void count(int i, unsigned v) { acc[i] += v; }
};
#define SF_PH(pc, i) do { if (pc) (pc)->mark(i); } while (0)
#define SF_PHC(pc, i, v) do { if (pc) (pc)->count(i, v); } while (0)
#else
struct PhaseClock {};
#define SF_PH(pc, i) do { } while (0)
#define SF_PHC(pc, i, v) do { } while (0)
#endif

struct NNHit {
    float d2;         // squared distance of the best candidate (the largest float below the threshold if none)
    int j;            // sorted position of the best candidate, -1 if none
    float px, py, pz; // its coordinates (kept in registers: re-reading the winner costs a round trip)
    float lb2;        // nn_search_wave only: every OTHER map point is at least sqrt(lb2) away from the query
};

__device__ __forceinline__ float l2_simple(float qx, float qy, float qz, float px, float py, float pz)
{
    float dx = qx - px, dy = qy - py, dz = qz - pz;
    float r = __fmul_rn(dx, dx);
    r = __fadd_rn(r, __fmul_rn(dy, dy));
    r = __fadd_rn(r, __fmul_rn(dz, dz));
    return r;
}

__device__ __forceinline__ bool window_accepts(const SfWindow &w, float px, float py, float pz)
{
    if (w.kind == 1) return l2_simple(w.c[0], w.c[1], w.c[2], px, py, pz) < w.r2;
    if (w.kind == 2) {
        double d0 = (double)px - w.oc[0], d1 = (double)py - w.oc[1], d2 = (double)pz - w.oc[2];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double proj = __dadd_rn(__dadd_rn(__dmul_rn(d0, w.oR[k]), __dmul_rn(d1, w.oR[3 + k])), __dmul_rn(d2, w.oR[6 + k]));
            ok = ok && (fabs(proj) <= w.ohalf[k]);
        }
        return ok;
    }
    return true;
}

// A lower bound of the distance from the query to every point the window accepts (0 when the query is inside the
// window or the bound would be in doubt).  A point p passes a sphere window iff |p - c|^2 < r2, so every accepted point
// is at least |q - c| - r away (triangle inequality); for a box window the largest single-axis excess does the same.
// When that bound exceeds sqrt(thr) no candidate can be accepted AND close enough: the search of such a query is
// skipped -- no result changes, only the ring walk that would end empty is spared (on the per-scan path the scan points
// beyond the 10 m map crop of localization_node.cpp:302 are a large share of every scan) -- and the bound itself is the
// query's runner-up bound for the neighbour-reuse certificate.
__device__ __forceinline__ float window_gap(const SfWindow &w, float qx, float qy, float qz)
{
    if (w.kind == 1) {
        const float dx = qx - w.c[0], dy = qy - w.c[1], dz = qz - w.c[2];
        return fmaxf(sqrtf(dx * dx + dy * dy + dz * dz) * 0.9995f - sqrtf(w.r2) * 1.0005f - 1.0e-4f, 0.0f);
    }
    if (w.kind == 2) {
        const double d0 = (double)qx - w.oc[0], d1 = (double)qy - w.oc[1], d2 = (double)qz - w.oc[2];
        double out = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { // the axes need not be orthonormal (the prior is a blend): an excess along axis k is a distance only after division by its norm
            const double proj = d0 * w.oR[k] + d1 * w.oR[3 + k] + d2 * w.oR[6 + k];
            const double n2 = w.oR[k] * w.oR[k] + w.oR[3 + k] * w.oR[3 + k] + w.oR[6 + k] * w.oR[6 + k];
            const double over = fmax(fabs(proj) - w.ohalf[k], 0.0);
            out = fmax(out, over / sqrt(fmax(n2, 1.0e-30)));
        }
        return fmaxf((float)out * 0.9995f - 1.0e-4f, 0.0f);
    }
    return 0.0f;
}

// The best candidate is the LEXICOGRAPHIC minimum of (d2, j) over everything visited -- the same rule in every
// search form, whatever the order ranges are walked in.  One 64-bit unsigned compare of (float bits of d2, index): d2
// is a non-negative finite float, so its bit pattern orders like its value.  "No candidate yet" is the pair
// (largest float below the acceptance threshold, index -1 = 0xffffffff): a candidate is then taken iff d2 < threshold,
// strictly (search_start below).
// TRACK: hit.lb2 follows the smallest squared distance among the examined candidates that did not end up as the
// best (a displaced best included); the best itself, met again, is not "another point".
__device__ __forceinline__ unsigned long long hit_key(float d2, int j) { return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(uint32_t)j; }

__device__ __forceinline__ float search_start(float thr) { return thr > 0.0f ? __uint_as_float(__float_as_uint(thr) - 1u) : -1.0f; }

template <bool WINDOW, bool TRACK = false>
__device__ __forceinline__ void consider(const SfWindow &w, float px, float py, float pz, int j, bool valid, float qx, float qy, float qz, NNHit &hit)
{
    const float d2 = l2_simple(qx, qy, qz, px, py, pz);
    const bool take = valid && hit_key(d2, j) < hit_key(hit.d2, hit.j) && (!WINDOW || window_accepts(w, px, py, pz));
    // runner-up bound: hit.d2 <= hit.lb2 holds while ranges are scanned, so the median of (this candidate, the best, the
    // bound) IS min(bound, the loser of this comparison) -- one v_med3_f32; the best itself met again and slots without a
    // candidate enter as +big and leave the bound alone
    if (TRACK) hit.lb2 = __builtin_amdgcn_fmed3f((valid && j != hit.j) ? d2 : 3.0e38f, hit.d2, hit.lb2);
    if (take) {
        hit.d2 = d2;
        hit.j = j;
        hit.px = px; hit.py = py; hit.pz = pz;
    }
}

// candidates [j, min(j + 4, b)) of a non-empty CSR range: four independent 16-byte loads, issued
// together.  Measured (TA / TCP counters): the kernel is bound by the number of per-lane accesses
// the L1 (TCP) has to look up, ~1 per clock per CU -- so a lane must not touch memory for a
// candidate that does not exist.  A branch per candidate serialises the loads (a round trip
// each, measured), clamping the index into the range keeps them in flight together but every
// duplicate is still an access; buffer addressing does both: an out-of-range offset makes the TA
// drop the lane's access and return zeros, without control flow.
#ifndef SF_NN_CLAMPED_LOADS
__device__ __forceinline__ float4 load_point(const SfGrid &g, uint32_t j, bool valid)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g.pts, 0, (int)((uint32_t)g.n * 16u), 0x00020000);
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, valid ? (int)(j * 16u) : -1, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#endif

template <bool WINDOW, bool TRACK = false>
__device__ __forceinline__ void scan4(const SfGrid &g, const SfWindow &w, uint32_t j, uint32_t b, float qx, float qy, float qz, NNHit &hit)
{
#ifndef SF_NN_CLAMPED_LOADS
    const float4 p0 = load_point(g, j, true), p1 = load_point(g, j + 1, j + 1 < b), p2 = load_point(g, j + 2, j + 2 < b), p3 = load_point(g, j + 3, j + 3 < b);
    consider<WINDOW, TRACK>(w, p0.x, p0.y, p0.z, (int)j, true, qx, qy, qz, hit);
    consider<WINDOW, TRACK>(w, p1.x, p1.y, p1.z, (int)(j + 1), j + 1 < b, qx, qy, qz, hit);
    consider<WINDOW, TRACK>(w, p2.x, p2.y, p2.z, (int)(j + 2), j + 2 < b, qx, qy, qz, hit);
    consider<WINDOW, TRACK>(w, p3.x, p3.y, p3.z, (int)(j + 3), j + 3 < b, qx, qy, qz, hit);
#else
    const uint32_t last = b - 1;
    const uint32_t j1 = min(j + 1, last), j2 = min(j + 2, last), j3 = min(j + 3, last);
    const float4 p0 = g.pts[j], p1 = g.pts[j1], p2 = g.pts[j2], p3 = g.pts[j3];
    consider<WINDOW, TRACK>(w, p0.x, p0.y, p0.z, (int)j, true, qx, qy, qz, hit);
    consider<WINDOW, TRACK>(w, p1.x, p1.y, p1.z, (int)j1, j + 1 < b, qx, qy, qz, hit);
    consider<WINDOW, TRACK>(w, p2.x, p2.y, p2.z, (int)j2, j + 2 < b, qx, qy, qz, hit);
    consider<WINDOW, TRACK>(w, p3.x, p3.y, p3.z, (int)j3, j + 3 < b, qx, qy, qz, hit);
#endif
}

// CSR candidates [a, b).  (Eight loads in flight per trip instead of four: measured slower at 32 scans in flight --
// 545 -> 609 us for a searching launch, 89 -> 94 us for a verifying one -- and slower on the per-scan path too, even
// when only the REF_CPP search was built that way: 404 -> 434 us per search at 32 in flight, callback 0.72 -> 0.76 ms.)
template <bool WINDOW, bool TRACK = false>
__device__ __forceinline__ void scan_range(const SfGrid &g, const SfWindow &w, uint32_t a, uint32_t b, float qx, float qy, float qz, NNHit &hit)
{
    for (uint32_t j = a; j < b; j += 4) scan4<WINDOW, TRACK>(g, w, j, b, qx, qy, qz, hit);
}

__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); } // one v_min3_f32

// cell_start[c-1 .. c+2] in ONE 16-byte load (the table carries one pad entry in front, so
// c - 1 >= -1 is addressable; dword alignment is enough for global_load_dwordx4)
struct __attribute__((packed, aligned(4))) RowBounds { uint32_t s0, s1, s2, s3; };
__device__ __forceinline__ RowBounds load_row_bounds(const SfGrid &g, size_t cell)
{
    return *reinterpret_cast<const RowBounds *>(g.cell_start + cell - 1);
}

// one (y,z) row around column cx: the query's own x cell first, then the left / right cell
// only while their gap (row gap + x gap) is still smaller than the best distance
template <bool WINDOW>
__device__ __forceinline__ void scan_row(const SfGrid &g, const SfWindow &w, const RowBounds &rb, float gap2yz, float gxm2, float gxp2, bool xm, bool xp, float qx,
                                         float qy, float qz, NNHit &hit)
{
    scan_range<WINDOW>(g, w, rb.s1, rb.s2, qx, qy, qz, hit);
    if (xm && (gap2yz + gxm2) * 0.998f < hit.d2) scan_range<WINDOW>(g, w, rb.s0, rb.s1, qx, qy, qz, hit);
    if (xp && (gap2yz + gxp2) * 0.998f < hit.d2) scan_range<WINDOW>(g, w, rb.s2, rb.s3, qx, qy, qz, hit);
}

// a face distance with the rounding slack of the grid coordinates taken off (never negative): the
// true distance is at least this
__device__ __forceinline__ float safe_gap(float gap, float eps) { return fmaxf(gap - eps, 0.0f); }

// distance (in cells, >= 0) from grid coordinate gc to the cell interval [c, c+1]
__device__ __forceinline__ float cell_gap(float gc, int c, int cq)
{
    if (c == cq) return 0.0f;
    float d = c < cq ? gc - (float)(c + 1) : (float)c - gc;
    return d > 0.0f ? d : 0.0f;
}

// thr: candidates are accepted iff d2 < thr (reference: max_correspondence_dist_ itself,
// icp_point_to_point.cpp:70; Open3D: radius^2)
// rings R_first, R_first + 1, ... around the query's cell until the exactness test holds
template <bool WINDOW>
__device__ __forceinline__ void nn_rings(const SfGrid &g, const SfWindow &w, float qx, float qy, float qz, int R_first, NNHit &hit)
{
    const float gx = (qx - g.org[0]) * g.inv_h, gy = (qy - g.org[1]) * g.inv_h, gz = (qz - g.org[2]) * g.inv_h;
    const int nx = g.dim[0], ny = g.dim[1], nz = g.dim[2];
    // clamp in float first: a far-away query must not overflow the int conversion
    const int cx = (int)fminf(fmaxf(floorf(gx), 0.0f), (float)(nx - 1));
    const int cy = (int)fminf(fmaxf(floorf(gy), 0.0f), (float)(ny - 1));
    const int cz = (int)fminf(fmaxf(floorf(gz), 0.0f), (float)(nz - 1));
    const float h = g.h;
    const int rcap = max(nx, max(ny, nz));

    for (int R = R_first; R <= rcap; ++R) {
        {
            // ---- CSR path: rows of 2R+1 cells
            const int x0 = max(cx - R, 0), x1 = min(cx + R, nx - 1);
            const int y0 = max(cy - R, 0), y1 = min(cy + R, ny - 1);
            const int z0 = max(cz - R, 0), z1 = min(cz + R, nz - 1);
            if (R == 1) {
                // Ring 1 on the CSR rows.  Per query: one 16-byte look-up gives the bounds of the
                // three cells of a row; the own cell is scanned first, the x neighbours and the
                // 8 neighbouring rows only while their gap is smaller than the best distance.
                // Which rows a query needs depends on where it sits in its cell, so the row list is
                // per-LANE data (bit mask walked with ffs) and the next row's bounds are requested
                // before the current row is scanned.
                //   k      0   1   2   3   4   5   6   7
                //   dy    -1  +1   0   0  -1  +1  -1  +1
                //   dz     0   0  -1  +1  -1  -1  +1  +1
                constexpr uint32_t OYP = 0u | (2u << 2) | (1u << 4) | (1u << 6) | (0u << 8) | (2u << 10) | (0u << 12) | (2u << 14);
                constexpr uint32_t OZP = 1u | (1u << 2) | (0u << 4) | (2u << 6) | (0u << 8) | (0u << 10) | (2u << 12) | (2u << 14);
                const float fx = gx - (float)cx, fy = gy - (float)cy, fz = gz - (float)cz;
                const float ge = g.gap_eps;
                const float gxm = safe_gap(fx * h, ge), gxp = safe_gap((1.0f - fx) * h, ge);
                const float gym = safe_gap(fy * h, ge), gyp = safe_gap((1.0f - fy) * h, ge);
                const float gzm = safe_gap(fz * h, ge), gzp = safe_gap((1.0f - fz) * h, ge);
                const float gxm2 = gxm * gxm, gxp2 = gxp * gxp;
                const bool xm = cx > 0, xp = cx < nx - 1;
                auto row_gap2 = [&](int k) -> float {
                    const int dy = (int)((OYP >> (2 * k)) & 3u) - 1, dz = (int)((OZP >> (2 * k)) & 3u) - 1;
                    const float ry = dy < 0 ? gym : (dy > 0 ? gyp : 0.0f);
                    const float rz = dz < 0 ? gzm : (dz > 0 ? gzp : 0.0f);
                    return ry * ry + rz * rz;
                };
                auto row_cell = [&](int k) -> size_t {
                    const int dy = (int)((OYP >> (2 * k)) & 3u) - 1, dz = (int)((OZP >> (2 * k)) & 3u) - 1;
                    return ((size_t)(cz + dz) * ny + (cy + dy)) * nx + cx;
                };
                {
                    const RowBounds rb = load_row_bounds(g, ((size_t)cz * ny + cy) * nx + cx);
                    scan_row<WINDOW>(g, w, rb, 0.0f, gxm2, gxp2, xm, xp, qx, qy, qz, hit);
                }
                uint32_t mask = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int dy = (int)((OYP >> (2 * k)) & 3u) - 1, dz = (int)((OZP >> (2 * k)) & 3u) - 1;
                    const bool inside = (unsigned)(cy + dy) < (unsigned)ny && (unsigned)(cz + dz) < (unsigned)nz;
                    if (inside && row_gap2(k) * 0.998f < hit.d2) mask |= 1u << k;
                }
                int k = -1;
                RowBounds cur = {0, 0, 0, 0};
                if (mask) {
                    k = __ffs((int)mask) - 1;
                    mask &= mask - 1;
                    cur = load_row_bounds(g, row_cell(k));
                }
                while (k >= 0) {
                    int k1 = -1;
                    RowBounds nxt = cur;
                    while (mask) {
                        const int kk = __ffs((int)mask) - 1;
                        mask &= mask - 1;
                        if (row_gap2(kk) * 0.998f < hit.d2) {
                            k1 = kk;
                            nxt = load_row_bounds(g, row_cell(kk));
                            break;
                        }
                    }
                    const float g2 = row_gap2(k);
                    if (g2 * 0.998f < hit.d2) scan_row<WINDOW>(g, w, cur, g2, gxm2, gxp2, xm, xp, qx, qy, qz, hit);
                    k = k1;
                    cur = nxt;
                }
            } else {
                for (int z = z0; z <= z1; ++z) {
                    const float rz = safe_gap(cell_gap(gz, z, cz) * h, g.gap_eps);
                    for (int y = y0; y <= y1; ++y) {
                        const float ry = safe_gap(cell_gap(gy, y, cy) * h, g.gap_eps);
                        if ((ry * ry + rz * rz) * 0.998f >= hit.d2) continue;
                        const size_t row = ((size_t)z * ny + y) * nx;
                        scan_range<WINDOW>(g, w, g.cell_start[row + x0], g.cell_start[row + x1 + 1], qx, qy, qz, hit);
                    }
                }
            }
        }
        // distance from the query to the nearest face of the scanned block that still has
        // grid cells behind it
        float m = 3.0e38f;
        if (cx - R > 0) m = fminf(m, (gx - (float)(cx - R)) * h);
        if (cx + R < nx - 1) m = fminf(m, ((float)(cx + R + 1) - gx) * h);
        if (cy - R > 0) m = fminf(m, (gy - (float)(cy - R)) * h);
        if (cy + R < ny - 1) m = fminf(m, ((float)(cy + R + 1) - gy) * h);
        if (cz - R > 0) m = fminf(m, (gz - (float)(cz - R)) * h);
        if (cz + R < nz - 1) m = fminf(m, ((float)(cz + R + 1) - gz) * h);
        if (m >= 3.0e38f) break; // whole grid scanned
        const float mm = safe_gap(m, g.gap_eps) * 0.999f;
        if (hit.d2 <= mm * mm) break;
    }
}

template <bool WINDOW>
__device__ __forceinline__ NNHit nn_search(const SfGrid &g, const SfWindow &w, float qx, float qy, float qz, float thr)
{
    NNHit hit;
    hit.d2 = search_start(thr);
    hit.j = -1;
    hit.px = hit.py = hit.pz = 0.0f;
    hit.lb2 = 0.0f;
    if (!(isfinite(qx) && isfinite(qy) && isfinite(qz)) || g.n == 0) return hit;
    if (WINDOW) {
        const float gap = window_gap(w, qx, qy, qz);
        if (gap * gap > thr) return hit;
    }
    nn_rings<WINDOW>(g, w, qx, qy, qz, 1, hit);
    return hit;
}

// ------------------------------------------------------------------ wave-cooperative search
// Measured (pass counters, 200 k-point scans vs the 10 M-point map): after every lane has scanned
// its query's own cell (one step with all 64 lanes busy), the per-lane walk above spends ~10 more
// wave-level steps on the x neighbours / neighbouring rows that still survive pruning, with ~6 of
// 64 lanes active in each -- in total fewer than 64 lane-steps of real work.  Here the surviving
// (query, range) TASKS of the whole wave go into an LDS queue and are dealt out one per lane, so
// that work takes one or two full steps.  A task scans one contiguous candidate range:
//   t = 0, 1   the left / right x neighbour of the own cell
//   t = 2 + k  neighbouring row k (own-x cell plus the x neighbours whose gap is still below the best)
//   t = 10     what is left of an own cell that holds more than four points
// and lowers its owner's packed (d2, j) in LDS with ds_min_u64.  The result is the lexicographic
// minimum of (d2, j) over everything visited; pruning only ever skips ranges that cannot hold a
// candidate as good as the current best (0.2 % margin), and a task reports ties with the best it
// started from, so the result does not depend on the order in which lanes finish.
struct WaveNN {
    unsigned long long best[64]; // (float bits of d2) << 32 | j; j = 0xffffffff: none
    uint32_t lb2[64];            // float bits: lower bound of the squared distance to every point but the best
    float4 q[64];
    RowBounds rb0[64];           // bounds of the own cell and its x neighbours (tasks 0, 1, 10)
    float gap[6][64];            // the owner's gaps (gxm2, gxp2, gym, gyp, gzm, gzp) and ...
    unsigned long long cell0[64]; // ... the linear index of its cell: a task gets its geometry from here instead of recomputing it (~45 instructions per task)
    uint16_t task[64 * 11];      // owner lane << 4 | t, grouped by t
};

__device__ __forceinline__ unsigned long long pack_hit(float d2, int j) { return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(uint32_t)j; }

struct QueryGeo {
    int cx, cy, cz;
    float gxm2, gxp2;          // (gap to the left / right x neighbour)^2, 3e38 at the grid border
    float gym, gyp, gzm, gzp;  // gaps to the neighbouring rows
};

__device__ __forceinline__ QueryGeo query_geo(const SfGrid &g, float qx, float qy, float qz)
{
    QueryGeo G;
    const float gx = (qx - g.org[0]) * g.inv_h, gy = (qy - g.org[1]) * g.inv_h, gz = (qz - g.org[2]) * g.inv_h;
    G.cx = (int)fminf(fmaxf(floorf(gx), 0.0f), (float)(g.dim[0] - 1));
    G.cy = (int)fminf(fmaxf(floorf(gy), 0.0f), (float)(g.dim[1] - 1));
    G.cz = (int)fminf(fmaxf(floorf(gz), 0.0f), (float)(g.dim[2] - 1));
    const float h = g.h;
    const float fx = gx - (float)G.cx, fy = gy - (float)G.cy, fz = gz - (float)G.cz;
    const float ge = g.gap_eps;
    const float gxm = safe_gap(fx * h, ge), gxp = safe_gap((1.0f - fx) * h, ge);
    G.gxm2 = G.cx > 0 ? gxm * gxm : 3.0e38f;
    G.gxp2 = G.cx < g.dim[0] - 1 ? gxp * gxp : 3.0e38f;
    G.gym = safe_gap(fy * h, ge);
    G.gyp = safe_gap((1.0f - fy) * h, ge);
    G.gzm = safe_gap(fz * h, ge);
    G.gzp = safe_gap((1.0f - fz) * h, ge);
    return G;
}

//   k      0   1   2   3   4   5   6   7
//   dy    -1  +1   0   0  -1  +1  -1  +1
//   dz     0   0  -1  +1  -1  -1  +1  +1
__device__ __forceinline__ int row_dy(int k) { return (int)((0x8858u >> (2 * k)) & 3u) - 1; } // dy + 1, two bits per k
__device__ __forceinline__ int row_dz(int k) { return (int)((0xa085u >> (2 * k)) & 3u) - 1; }

__device__ __forceinline__ float row_gap2(const QueryGeo &G, int k)
{
    const int dy = row_dy(k), dz = row_dz(k);
    const float ry = dy < 0 ? G.gym : (dy > 0 ? G.gyp : 0.0f);
    const float rz = dz < 0 ? G.gzm : (dz > 0 ? G.gzp : 0.0f);
    return ry * ry + rz * rz;
}

// every lane of the wave must call this (lanes without a query pass valid = false: they still work).
// Besides the nearest neighbour the result carries lb2: a lower bound of the squared distance from
// the query to every OTHER map point (the runner-up among the examined candidates, the gaps of
// everything that was pruned, the boundary of the 27-cell block) -- what k_nn_red needs to prove,
// one iteration later, that the neighbour cannot have changed.
// seed (optional, seed.j >= 0): a map point already known for this query -- its neighbour of an earlier search -- with its
// CURRENT squared distance seed.d2 (l2_simple of the query and the point).  The search then starts from that bound
// instead of the acceptance threshold: the same exact result (the seed is a candidate like any other, met again when
// its range is scanned), most neighbour ranges pruned before they are visited.
// COOP: how the rare queries that need more than ring 1 go on (see the end of the function)
template <bool WINDOW, bool COOP = false>
__device__ __forceinline__ NNHit nn_search_wave(const SfGrid &g, const SfWindow &w, bool valid, float qx, float qy, float qz, float thr, WaveNN *ws, NNHit seed = NNHit{0.0f, -1, 0.0f, 0.0f, 0.0f, 0.0f},
                                               PhaseClock *pc = nullptr)
{
    const int lane = (int)__lane_id();
    const int nx = g.dim[0], ny = g.dim[1], nz = g.dim[2];
    NNHit hit;
    hit.d2 = search_start(thr);
    hit.j = -1;
    hit.px = hit.py = hit.pz = 0.0f;
    hit.lb2 = 3.0e38f;
    valid = valid && isfinite(qx) && isfinite(qy) && isfinite(qz) && g.n > 0;
    if (WINDOW && valid) {
        const float gap = window_gap(w, qx, qy, qz);
        if (gap * gap > thr) { // nothing the window accepts is within the acceptance radius
            hit.lb2 = gap * gap;
            valid = false;
        }
    }
    if (valid) {
        // a query farther from the map's bounding box (the grid's: every point is inside it) than the acceptance radius has
        // no neighbour -- known without a walk through empty rings, and the distance to the box is its runner-up bound: scan
        // points beyond the rim of the map certify as "still nothing" from then on instead of searching in every launch
        const float ox = fmaxf(fmaxf(g.org[0] - qx, qx - (g.org[0] + (float)nx * g.h)), 0.0f);
        const float oy = fmaxf(fmaxf(g.org[1] - qy, qy - (g.org[1] + (float)ny * g.h)), 0.0f);
        const float oz = fmaxf(fmaxf(g.org[2] - qz, qz - (g.org[2] + (float)nz * g.h)), 0.0f);
        const float gap = fmaxf(sqrtf(ox * ox + oy * oy + oz * oz) * 0.9995f - 1.0e-3f, 0.0f); // (1 mm: the float32 corner of the box)
        if (gap * gap > thr) {
            hit.lb2 = gap * gap;
            valid = false;
        }
    }
    if (valid && seed.j >= 0 && seed.d2 < thr) { hit.d2 = seed.d2; hit.j = seed.j; hit.px = seed.px; hit.py = seed.py; hit.pz = seed.pz; }
    uint32_t mask = 0;
#ifdef SF_PHASE_TRACE
    QueryGeo G{};
    RowBounds rb0{0, 0, 0, 0};
    if (valid) {
        G = query_geo(g, qx, qy, qz);
        rb0 = load_row_bounds(g, ((size_t)G.cz * ny + G.cy) * nx + G.cx);
        ws->rb0[lane] = rb0;
    }
    SF_PH(pc, 1);
    if (valid && rb0.s1 < rb0.s2) scan4<WINDOW, true>(g, w, rb0.s1, rb0.s2, qx, qy, qz, hit);
    SF_PH(pc, 2);
    if (valid) {
#else
    if (valid) {
        const QueryGeo G = query_geo(g, qx, qy, qz);
        const RowBounds rb0 = load_row_bounds(g, ((size_t)G.cz * ny + G.cy) * nx + G.cx);
        ws->rb0[lane] = rb0;
        if (rb0.s1 < rb0.s2) scan4<WINDOW, true>(g, w, rb0.s1, rb0.s2, qx, qy, qz, hit);
#endif
        if (rb0.s1 + 4 < rb0.s2) {
            if (hit.d2 > 0.0f) mask |= 1u << 10;
            else hit.lb2 = 0.0f; // the rest of the cell is not examined
        }
        // a range that is not queued because its gap is too large bounds the runner-up by that gap
        if (rb0.s0 < rb0.s1) {
            if (G.gxm2 * 0.998f < hit.d2) mask |= 1u;
            else hit.lb2 = fminf(hit.lb2, G.gxm2);
        }
        if (rb0.s2 < rb0.s3) {
            if (G.gxp2 * 0.998f < hit.d2) mask |= 2u;
            else hit.lb2 = fminf(hit.lb2, G.gxp2);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool inside = (unsigned)(G.cy + row_dy(k)) < (unsigned)ny && (unsigned)(G.cz + row_dz(k)) < (unsigned)nz;
            const float gap2 = row_gap2(G, k);
            if (inside) {
                if (gap2 * 0.998f < hit.d2) mask |= 4u << k;
                else hit.lb2 = fminf(hit.lb2, gap2);
            }
        }
        if (mask) { // what the tasks of this query need of its geometry
            ws->gap[0][lane] = G.gxm2; ws->gap[1][lane] = G.gxp2;
            ws->gap[2][lane] = G.gym; ws->gap[3][lane] = G.gyp; ws->gap[4][lane] = G.gzm; ws->gap[5][lane] = G.gzp;
            ws->cell0[lane] = ((unsigned long long)G.cz * (unsigned long long)ny + (unsigned long long)G.cy) * (unsigned long long)nx + (unsigned long long)G.cx;
        }
    }
    ws->best[lane] = pack_hit(hit.d2, hit.j);
    ws->lb2[lane] = __float_as_uint(hit.lb2);
    ws->q[lane] = make_float4(qx, qy, qz, 0.0f);
    int total = 0;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
        const bool has = (mask >> t) & 1u;
        const unsigned long long bal = __ballot(has);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (has) ws->task[total + rank] = (uint16_t)((lane << 4) | t);
        total += __popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    SF_PH(pc, 3);
    SF_PHC(pc, 11, (unsigned)total);
#ifdef SF_PHASE_TRACE
    unsigned lane_trips = 0, lane_cands = 0;
#endif
    for (int i0 = 0; i0 < total; i0 += 64) {
        SF_PHC(pc, 9, 1u);
        const int idx = i0 + lane;
#ifdef SF_PHASE_TRACE
        unsigned my_trips = 0;
#endif
        if (idx < total) {
            const uint32_t e = ws->task[idx];
            const int owner = (int)(e >> 4), t = (int)(e & 15u);
            const float4 Q = ws->q[owner];
            QueryGeo G; // from the owner's entry (cx, cy, cz are not needed: the row is addressed relative to the owner's cell)
            G.cx = G.cy = G.cz = 0;
            G.gxm2 = ws->gap[0][owner]; G.gxp2 = ws->gap[1][owner];
            G.gym = ws->gap[2][owner]; G.gyp = ws->gap[3][owner]; G.gzm = ws->gap[4][owner]; G.gzp = ws->gap[5][owner];
            const unsigned long long start = __atomic_load_n(&ws->best[owner], __ATOMIC_RELAXED);
            const float cur = __uint_as_float((uint32_t)(start >> 32));
            const bool own_row = t < 2 || t == 10;
            const float g2 = own_row ? (t == 0 ? G.gxm2 : (t == 1 ? G.gxp2 : 0.0f)) : row_gap2(G, t - 2);
            float bound = g2; // what this task contributes to the owner's runner-up bound
            if (g2 * 0.998f < cur) {
                uint32_t a, b;
                bound = 3.0e38f;
                if (own_row) {
                    const RowBounds rb = ws->rb0[owner];
                    a = t == 0 ? rb.s0 : (t == 1 ? rb.s2 : rb.s1 + 4);
                    b = t == 0 ? rb.s1 : (t == 1 ? rb.s3 : rb.s2);
                } else {
                    const long long step = ((long long)row_dz(t - 2) * ny + row_dy(t - 2)) * nx; // the row's cell relative to the owner's (inside the grid: checked when queued)
                    const RowBounds rb = load_row_bounds(g, (size_t)((long long)ws->cell0[owner] + step));
                    const bool xm = (g2 + G.gxm2) * 0.998f < cur, xp = (g2 + G.gxp2) * 0.998f < cur;
                    a = xm ? rb.s0 : rb.s1;
                    b = xp ? rb.s3 : rb.s2;
                    if (!xm && rb.s0 < rb.s1) bound = g2 + G.gxm2;
                    if (!xp && rb.s2 < rb.s3) bound = fminf(bound, g2 + G.gxp2);
                }
                // the task starts from the owner's best as it stands: ties are settled by the index rule of consider(),
                // the best itself -- met again in its own range -- is neither taken nor counted as a runner-up
                NNHit h;
                h.d2 = cur;
                h.j = (int)(uint32_t)start;
                h.px = h.py = h.pz = 0.0f;
                h.lb2 = 3.0e38f;
                scan_range<WINDOW, true>(g, w, a, b, Q.x, Q.y, Q.z, h);
#ifdef SF_PHASE_TRACE
                my_trips = (b - a + 3u) / 4u; lane_cands += b - a; lane_trips += my_trips;
#endif
                bound = fminf(bound, h.lb2);
                const unsigned long long mine = pack_hit(h.d2, h.j);
                if (mine != start) { // something lexicographically smaller: whichever of (the best by now, this candidate) loses is a runner-up
                    const unsigned long long old = atomicMin(&ws->best[owner], mine);
                    bound = fminf(bound, __uint_as_float((uint32_t)((old > mine ? old : mine) >> 32)));
                }
            }
            atomicMin(&ws->lb2[owner], __float_as_uint(bound));
        }
#ifdef SF_PHASE_TRACE
        { unsigned m = my_trips; for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o)); SF_PHC(pc, 10, m); }
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    SF_PH(pc, 4);
#ifdef SF_PHASE_TRACE
    { unsigned m = lane_trips, c = lane_cands; for (int o = 32; o > 0; o >>= 1) { m += (unsigned)__shfl_xor((int)m, o); c += (unsigned)__shfl_xor((int)c, o); } SF_PHC(pc, 12, m); SF_PHC(pc, 13, c); }
#endif
    bool more = false; // ring 1 did not settle this query
    if (valid) {
        const unsigned long long m = ws->best[lane];
        const int j = (int)(uint32_t)m;
        hit.lb2 = __uint_as_float(ws->lb2[lane]);
        if (j != hit.j) { // a task found something better: fetch the winner's coordinates
            const float4 p = g.pts[j];
            hit.d2 = __uint_as_float((uint32_t)(m >> 32));
            hit.j = j;
            hit.px = p.x; hit.py = p.y; hit.pz = p.z;
        }
        // exactness of ring 1 (same test as nn_rings); otherwise on to ring 2
        const float gx = (qx - g.org[0]) * g.inv_h, gy = (qy - g.org[1]) * g.inv_h, gz = (qz - g.org[2]) * g.inv_h;
        const int cx = (int)fminf(fmaxf(floorf(gx), 0.0f), (float)(nx - 1));
        const int cy = (int)fminf(fmaxf(floorf(gy), 0.0f), (float)(ny - 1));
        const int cz = (int)fminf(fmaxf(floorf(gz), 0.0f), (float)(nz - 1));
        const float h = g.h;
        float mface = 3.0e38f;
        if (cx - 1 > 0) mface = fminf(mface, (gx - (float)(cx - 1)) * h);
        if (cx + 1 < nx - 1) mface = fminf(mface, ((float)(cx + 2) - gx) * h);
        if (cy - 1 > 0) mface = fminf(mface, (gy - (float)(cy - 1)) * h);
        if (cy + 1 < ny - 1) mface = fminf(mface, ((float)(cy + 2) - gy) * h);
        if (cz - 1 > 0) mface = fminf(mface, (gz - (float)(cz - 1)) * h);
        if (cz + 1 < nz - 1) mface = fminf(mface, ((float)(cz + 2) - gz) * h);
        const float mm = safe_gap(mface, g.gap_eps) * 0.999f;
        if (mface < 3.0e38f) {
            hit.lb2 = fminf(hit.lb2, mm * mm); // nothing outside the 27 cells is closer than their boundary
            more = !(hit.d2 <= mm * mm);
        }
        if (!COOP && more) { // lane by lane (the dense batch kernels: rare there, and the cooperative form below costs k_nn_red 8 VGPRs = one wave per SIMD)
            nn_rings<WINDOW>(g, w, qx, qy, qz, 2, hit);
            hit.lb2 = 0.0f; // no bound kept for the per-lane rings
        }
    }
    SF_PH(pc, 5);
    if (!COOP) return hit;
    // COOP: queries whose best is still farther than the boundary of their 27 cells (scan points with no map point nearby:
    // new ground, moving objects, the rim of the map crop) go on ring by ring TOGETHER.  Lane by lane that is a serial walk
    // of (2R+1)^2 rows by a handful of lanes while the rest of the wave idles -- measured on the per-scan path: it made
    // every index cell below the acceptance radius slower than a 0.72 m cell (alignment 409 us at 0.72 m; 557 / 498 us at
    // 0.5 / 0.36 m lane by lane, 349 / 355 us with this).  Here the (query, row) pairs of ring R are dealt out over all 64
    // lanes like the ring-1 tasks: rows outside the previous block are scanned whole, rows inside it only in their two new
    // end cells, each pruned by its gap against the owner's best as it stands.  Same result as nn_rings: the lexicographic
    // minimum of (d2, j) over everything visited, and only ranges that cannot hold a better candidate are skipped.
    unsigned long long unres = __ballot(more);
    const int rcap = max(nx, max(ny, nz));
    for (int R = 2; unres != 0ull && R <= rcap; ++R) {
        {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(unres >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)unres, 0u));
            if (more) {
                ws->task[rank] = (uint16_t)lane;
                ws->best[lane] = pack_hit(hit.d2, hit.j);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int side = 2 * R + 1, nrows = side * side, total = __popcll(unres) * nrows;
        const float h = g.h, ge = g.gap_eps;
        for (int idx = lane; idx < total; idx += 64) {
            const int u = idx / nrows, r = idx - u * nrows;
            const int owner = (int)ws->task[u];
            const int dyi = r / side - R, dzi = r - (r / side) * side - R;
            const float4 Q = ws->q[owner];
            const float gx = (Q.x - g.org[0]) * g.inv_h, gy = (Q.y - g.org[1]) * g.inv_h, gz = (Q.z - g.org[2]) * g.inv_h;
            const int cx = (int)fminf(fmaxf(floorf(gx), 0.0f), (float)(nx - 1));
            const int cy = (int)fminf(fmaxf(floorf(gy), 0.0f), (float)(ny - 1));
            const int cz = (int)fminf(fmaxf(floorf(gz), 0.0f), (float)(nz - 1));
            const int y = cy + dyi, z = cz + dzi;
            if ((unsigned)y >= (unsigned)ny || (unsigned)z >= (unsigned)nz) continue;
            const float ry = safe_gap(cell_gap(gy, y, cy) * h, ge), rz = safe_gap(cell_gap(gz, z, cz) * h, ge);
            const float g2 = ry * ry + rz * rz;
            const unsigned long long start = __atomic_load_n(&ws->best[owner], __ATOMIC_RELAXED);
            const float cur = __uint_as_float((uint32_t)(start >> 32));
            if (g2 * 0.998f >= cur) continue;
            NNHit t;
            t.d2 = cur;
            t.j = (int)(uint32_t)start;
            t.px = t.py = t.pz = 0.0f;
            t.lb2 = 0.0f;
            const size_t row = ((size_t)z * ny + y) * nx;
            const bool inner = abs(dyi) < R && abs(dzi) < R; // scanned up to x +- (R - 1) by the rings before
            if (!inner) {
                const int x0 = max(cx - R, 0), x1 = min(cx + R, nx - 1);
                scan_range<WINDOW>(g, w, g.cell_start[row + x0], g.cell_start[row + x1 + 1], Q.x, Q.y, Q.z, t);
            } else {
                if (cx - R >= 0) {
                    const float gl = safe_gap((gx - (float)(cx - R + 1)) * h, ge);
                    if ((g2 + gl * gl) * 0.998f < t.d2) scan_range<WINDOW>(g, w, g.cell_start[row + cx - R], g.cell_start[row + cx - R + 1], Q.x, Q.y, Q.z, t);
                }
                if (cx + R <= nx - 1) {
                    const float gr = safe_gap(((float)(cx + R) - gx) * h, ge);
                    if ((g2 + gr * gr) * 0.998f < t.d2) scan_range<WINDOW>(g, w, g.cell_start[row + cx + R], g.cell_start[row + cx + R + 1], Q.x, Q.y, Q.z, t);
                }
            }
            const unsigned long long mine = pack_hit(t.d2, t.j);
            if (mine != start) atomicMin(&ws->best[owner], mine);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (more) {
            const unsigned long long m = ws->best[lane];
            const int j = (int)(uint32_t)m;
            if (j != hit.j) {
                const float4 p = g.pts[j];
                hit.d2 = __uint_as_float((uint32_t)(m >> 32));
                hit.j = j;
                hit.px = p.x; hit.py = p.y; hit.pz = p.z;
            }
            hit.lb2 = 0.0f; // no runner-up bound is kept beyond ring 1
            const float gx = (qx - g.org[0]) * g.inv_h, gy = (qy - g.org[1]) * g.inv_h, gz = (qz - g.org[2]) * g.inv_h;
            const int cx = (int)fminf(fmaxf(floorf(gx), 0.0f), (float)(nx - 1));
            const int cy = (int)fminf(fmaxf(floorf(gy), 0.0f), (float)(ny - 1));
            const int cz = (int)fminf(fmaxf(floorf(gz), 0.0f), (float)(nz - 1));
            float mface = 3.0e38f; // the nearest face of the block of radius R that still has grid cells behind it
            if (cx - R > 0) mface = fminf(mface, (gx - (float)(cx - R)) * h);
            if (cx + R < nx - 1) mface = fminf(mface, ((float)(cx + R + 1) - gx) * h);
            if (cy - R > 0) mface = fminf(mface, (gy - (float)(cy - R)) * h);
            if (cy + R < ny - 1) mface = fminf(mface, ((float)(cy + R + 1) - gy) * h);
            if (cz - R > 0) mface = fminf(mface, (gz - (float)(cz - R)) * h);
            if (cz + R < nz - 1) mface = fminf(mface, ((float)(cz + R + 1) - gz) * h);
            const float mm = safe_gap(mface, ge) * 0.999f;
            more = mface < 3.0e38f && !(hit.d2 <= mm * mm);
        }
        unres = __ballot(more);
    }
    return hit;
}

} // namespace sf
