// feasibility probe: random table-entry reads over a 13 / 26 GB table + dependent point loads from a 160 MB array
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int EB, int NP>
__global__ __launch_bounds__(256) void k_probe(const uint4 *__restrict__ tbl, const float4 *__restrict__ pts, const uint32_t *__restrict__ vox, const float4 *__restrict__ q,
                                               int64_t n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = vox[i];
    const float4 qq = q[i];
    const uint4 A = tbl[(size_t)v * (EB / 16)];
    uint4 B = make_uint4(0, 0, 0, 0);
    if (EB >= 32) B = tbl[(size_t)v * (EB / 16) + 1];
    uint32_t j[8] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
    float best = 1e30f;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const float4 p = pts[j[k] % 10000000u];
        const float dx = p.x - qq.x, dy = p.y - qq.y, dz = p.z - qq.z;
        best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
    out[i] = best;
}

__global__ void k_fill(uint32_t *t, size_t words, int epw)
{
    for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (size_t)gridDim.x * 256) { const size_t v = w / epw; t[w] = (uint32_t)((v / 64) * 3 / 2 + (w % 8)); }
}

int main()
{
    const size_t ncell = 6400000, nvox = ncell * 64, npts = 10000000;
    const int B = 64, NQ = 200000;
    const int64_t n = (int64_t)B * NQ;
    std::vector<uint32_t> vox(n);
    std::mt19937_64 rng(1);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < NQ; ++i) vox[(size_t)b * NQ + i] = (uint32_t)(rng() % nvox);
        std::sort(vox.begin() + (size_t)b * NQ, vox.begin() + (size_t)(b + 1) * NQ);
    }
    uint32_t *d_vox; float4 *d_q, *d_pts; float *d_out; uint4 *d_tbl;
    CK(hipMalloc(&d_vox, n * 4)); CK(hipMalloc(&d_q, n * 16)); CK(hipMalloc(&d_pts, npts * 16)); CK(hipMalloc(&d_out, n * 4));
    CK(hipMemcpy(d_vox, vox.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_q, 0, n * 16)); CK(hipMemset(d_pts, 0, npts * 16));
    for (int eb : {32, 64}) {
        const size_t bytes = nvox * (size_t)eb;
        CK(hipMalloc(&d_tbl, bytes));
        hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint32_t *)d_tbl, bytes / 4, eb / 4);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int np : {0, 4, 7}) {
            float best_ms = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                const dim3 g((unsigned)((n + 255) / 256));
                if (eb == 32) {
                    if (np == 0) hipLaunchKernelGGL((k_probe<32, 0>), g, dim3(256), 0, 0, d_tbl, d_pts, d_vox, d_q, n, d_out);
                    if (np == 4) hipLaunchKernelGGL((k_probe<32, 4>), g, dim3(256), 0, 0, d_tbl, d_pts, d_vox, d_q, n, d_out);
                    if (np == 7) hipLaunchKernelGGL((k_probe<32, 7>), g, dim3(256), 0, 0, d_tbl, d_pts, d_vox, d_q, n, d_out);
                } else {
                    if (np == 0) hipLaunchKernelGGL((k_probe<64, 0>), g, dim3(256), 0, 0, d_tbl, d_pts, d_vox, d_q, n, d_out);
                    if (np == 4) hipLaunchKernelGGL((k_probe<64, 4>), g, dim3(256), 0, 0, d_tbl, d_pts, d_vox, d_q, n, d_out);
                    if (np == 7) hipLaunchKernelGGL((k_probe<64, 7>), g, dim3(256), 0, 0, d_tbl, d_pts, d_vox, d_q, n, d_out);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0) best_ms = std::min(best_ms, ms);
            }
            printf("entry %d B (table %.1f GB), %d point loads: %.1f us for %lld queries\n", eb, bytes / 1e9, np, best_ms * 1e3, (long long)n);
            fflush(stdout);
        }
        CK(hipFree(d_tbl));
    }
    return 0;
}
