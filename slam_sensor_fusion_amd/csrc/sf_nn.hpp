// sf_nn.hpp — exact 1-NN over the uniform-grid map index (device side, gfx950).
//
// Replaces the N serial FLANN kd-tree descents of
// localization/src/icp_point_to_point.cpp:64-69 (sourceTargetCorrespondences) and the
// Open3D hybrid search behind localization_python/.../localization_node.py:233-237.
// One lane per query.  The map is sorted by cell (x fastest), so the 2R+1 cells of one
// (y,z) row are ONE contiguous candidate range: a (2R+1)^3 block costs (2R+1)^2 range
// look-ups, each followed by 16-byte candidate loads (x, y, z, original index).
// The path is latency-bound (random 16-byte gathers served by L2 / Infinity Cache), so
// the code is written for memory-level parallelism: the centre row first (it usually
// holds the answer and prunes the rest), then the range bounds of all surviving rows in
// one batch of independent loads, candidates fetched four at a time.
// Exactness: after scanning the block of radius R around the query's cell, every
// unscanned point is at least m = distance(query, block boundary) away; the search stops
// when best <= m^2 (best starts at the acceptance threshold, so "nothing acceptable
// outside" ends it too), otherwise the block grows by one ring.
// Distance = FLANN L2_Simple in float32: ((dx*dx) + dy*dy) + dz*dz, unfused; strict "<".
#pragma once
#include "sf_common.hpp"

namespace sf {

struct NNHit {
    float d2;      // squared distance of the best candidate (== threshold if none)
    int j;         // sorted position of the best candidate, -1 if none
    float4 p;      // its x, y, z, bitcast(original index)
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

template <bool WINDOW>
__device__ __forceinline__ void consider(const SfWindow &w, const float4 &p, uint32_t j, bool valid, float qx, float qy, float qz, NNHit &hit)
{
    const float d2 = l2_simple(qx, qy, qz, p.x, p.y, p.z);
    if (valid && d2 < hit.d2) {
        if (!WINDOW || window_accepts(w, p.x, p.y, p.z)) {
            hit.d2 = d2;
            hit.j = (int)j;
            hit.p = p;
        }
    }
}

// candidates [a, b), four independent 16-byte loads in flight per step (indices clamped
// into the range, the tail is masked)
template <bool WINDOW>
__device__ __forceinline__ void scan_range(const SfGrid &g, const SfWindow &w, uint32_t a, uint32_t b, float qx, float qy, float qz, NNHit &hit)
{
    for (uint32_t j = a; j < b; j += 4) {
        const uint32_t last = b - 1;
        const uint32_t j1 = min(j + 1, last), j2 = min(j + 2, last), j3 = min(j + 3, last);
        const float4 p0 = g.pts[j], p1 = g.pts[j1], p2 = g.pts[j2], p3 = g.pts[j3];
        consider<WINDOW>(w, p0, j, true, qx, qy, qz, hit);
        consider<WINDOW>(w, p1, j1, j + 1 < b, qx, qy, qz, hit);
        consider<WINDOW>(w, p2, j2, j + 2 < b, qx, qy, qz, hit);
        consider<WINDOW>(w, p3, j3, j + 3 < b, qx, qy, qz, hit);
    }
}

// distance (in cells, >= 0) from grid coordinate gc to the cell interval [c, c+1]
__device__ __forceinline__ float cell_gap(float gc, int c, int cq)
{
    if (c == cq) return 0.0f;
    float d = c < cq ? gc - (float)(c + 1) : (float)c - gc;
    return d > 0.0f ? d : 0.0f;
}

// thr: candidates are accepted iff d2 < thr (reference: max_correspondence_dist_ itself,
// icp_point_to_point.cpp:70; Open3D: radius^2)
template <bool WINDOW>
__device__ __forceinline__ NNHit nn_search(const SfGrid &g, const SfWindow &w, float qx, float qy, float qz, float thr)
{
    NNHit hit;
    hit.d2 = thr;
    hit.j = -1;
    hit.p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(isfinite(qx) && isfinite(qy) && isfinite(qz)) || g.n == 0) return hit;
    const float gx = (qx - g.org[0]) * g.inv_h, gy = (qy - g.org[1]) * g.inv_h, gz = (qz - g.org[2]) * g.inv_h;
    const int nx = g.dim[0], ny = g.dim[1], nz = g.dim[2];
    // clamp in float first: a far-away query must not overflow the int conversion
    const int cx = (int)fminf(fmaxf(floorf(gx), 0.0f), (float)(nx - 1));
    const int cy = (int)fminf(fmaxf(floorf(gy), 0.0f), (float)(ny - 1));
    const int cz = (int)fminf(fmaxf(floorf(gz), 0.0f), (float)(nz - 1));
    const float h = g.h;
    const int rcap = max(nx, max(ny, nz));

    for (int R = 1; R <= rcap; ++R) {
        const int x0 = max(cx - R, 0), x1 = min(cx + R, nx - 1);
        if (R == 1) {
            // centre row first: it usually holds the answer and prunes most other rows
            {
                const size_t row = ((size_t)cz * ny + cy) * nx;
                const uint32_t a = g.cell_start[row + x0], b = g.cell_start[row + x1 + 1];
                scan_range<WINDOW>(g, w, a, b, qx, qy, qz, hit);
            }
            // The 8 neighbouring rows (faces first, then corners).  Which of them a query still
            // needs depends on where it sits inside its cell, so the row is per-LANE data: every
            // lane walks only its own needed rows (a bit mask), the wave iterates
            // max-over-lanes(popcount) times instead of 8, and the range bounds of a lane's
            // next row are requested before its current row is scanned.
            // row k: dy = OY(k) - 1, dz = OZ(k) - 1 with 2-bit fields packed in constants
            //   k      0   1   2   3   4   5   6   7
            //   dy    -1  +1   0   0  -1  +1  -1  +1
            //   dz     0   0  -1  +1  -1  -1  +1  +1
            constexpr uint32_t OYP = 0u | (2u << 2) | (1u << 4) | (1u << 6) | (0u << 8) | (2u << 10) | (0u << 12) | (2u << 14);
            constexpr uint32_t OZP = 1u | (1u << 2) | (0u << 4) | (2u << 6) | (0u << 8) | (0u << 10) | (2u << 12) | (2u << 14);
            // gaps to the four neighbouring slabs (in metres, >= 0); corner gap = sum of squares
            const float fy = gy - (float)cy, fz = gz - (float)cz;
            const float gym = fmaxf(fy, 0.0f) * h, gyp = fmaxf(1.0f - fy, 0.0f) * h;
            const float gzm = fmaxf(fz, 0.0f) * h, gzp = fmaxf(1.0f - fz, 0.0f) * h;
            const bool ym = cy > 0, yp = cy < ny - 1, zm = cz > 0, zp = cz < nz - 1;
            auto row_gap2 = [&](int k) -> float {
                const int dy = (int)((OYP >> (2 * k)) & 3u) - 1, dz = (int)((OZP >> (2 * k)) & 3u) - 1;
                const float ry = dy < 0 ? gym : (dy > 0 ? gyp : 0.0f);
                const float rz = dz < 0 ? gzm : (dz > 0 ? gzp : 0.0f);
                return (ry * ry + rz * rz) * 0.998f;
            };
            uint32_t mask = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dy = (int)((OYP >> (2 * k)) & 3u) - 1, dz = (int)((OZP >> (2 * k)) & 3u) - 1;
                const bool inside = (dy < 0 ? ym : (dy > 0 ? yp : true)) && (dz < 0 ? zm : (dz > 0 ? zp : true));
                if (inside && row_gap2(k) < hit.d2) mask |= 1u << k;
            }
            uint32_t a = 0, b = 0;
            int k = -1;
            if (mask) {
                k = __ffs((int)mask) - 1;
                mask &= mask - 1;
                const int dy = (int)((OYP >> (2 * k)) & 3u) - 1, dz = (int)((OZP >> (2 * k)) & 3u) - 1;
                const size_t row = ((size_t)(cz + dz) * ny + (cy + dy)) * nx;
                a = g.cell_start[row + x0];
                b = g.cell_start[row + x1 + 1];
            }
            while (k >= 0) {
                uint32_t a1 = 0, b1 = 0;
                int k1 = -1;
                if (mask) { // request the next row's bounds before scanning this one
                    k1 = __ffs((int)mask) - 1;
                    mask &= mask - 1;
                    const int dy = (int)((OYP >> (2 * k1)) & 3u) - 1, dz = (int)((OZP >> (2 * k1)) & 3u) - 1;
                    const size_t row = ((size_t)(cz + dz) * ny + (cy + dy)) * nx;
                    a1 = g.cell_start[row + x0];
                    b1 = g.cell_start[row + x1 + 1];
                }
                if (row_gap2(k) < hit.d2) scan_range<WINDOW>(g, w, a, b, qx, qy, qz, hit);
                k = k1;
                a = a1;
                b = b1;
            }
        } else {
            const int y0 = max(cy - R, 0), y1 = min(cy + R, ny - 1);
            const int z0 = max(cz - R, 0), z1 = min(cz + R, nz - 1);
            for (int z = z0; z <= z1; ++z) {
                const float rz = cell_gap(gz, z, cz) * h;
                for (int y = y0; y <= y1; ++y) {
                    const float ry = cell_gap(gy, y, cy) * h;
                    if ((ry * ry + rz * rz) * 0.998f >= hit.d2) continue;
                    const size_t row = ((size_t)z * ny + y) * nx;
                    const uint32_t a = g.cell_start[row + x0], b = g.cell_start[row + x1 + 1];
                    scan_range<WINDOW>(g, w, a, b, qx, qy, qz, hit);
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
        const float mm = fmaxf(m, 0.0f) * 0.999f;
        if (hit.d2 <= mm * mm) break;
    }
    return hit;
}

} // namespace sf
