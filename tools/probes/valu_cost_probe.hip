// Issue cost of the vector instructions the search kernel is made of, with 1 / 2 / 4 / 8 waves per SIMD on every CU:
// cycles per wave-instruction per SIMD = (s_memtime ticks of a wave's loop) / (instructions of the loop x waves on the SIMD).
// 8 independent destination registers per instruction kind (no dependent chains shorter than 8 instructions).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/valu_cost_probe.hip -o /tmp/valu_cost_probe && /tmp/valu_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// F(n): the instruction with destination / accumulator operand %n (n = "0" .. "7"); %8 = a second source of the same kind, %9 = a float source
#define ALL8(F) F("0") F("1") F("2") F("3") F("4") F("5") F("6") F("7")
#define LOOP32(F) ALL8(F) ALL8(F) ALL8(F) ALL8(F)

#define K_F32(NAME, F)                                                                                                                  \
    __global__ __launch_bounds__(512) void NAME(unsigned long long *out, float seed, int iters)                                         \
    {                                                                                                                                   \
        float r[8];                                                                                                                     \
        for (int i = 0; i < 8; ++i) r[i] = seed + (float)(threadIdx.x * 8 + i);                                                         \
        float y = seed * 1.0001f, z = seed;                                                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                     \
        for (int it = 0; it < iters; ++it)                                                                                              \
            asm volatile(LOOP32(F) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(y), "v"(z) : "vcc", "s20", "s21"); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                     \
        float s = 0.f;                                                                                                                  \
        for (int i = 0; i < 8; ++i) s += r[i];                                                                                          \
        if (s == 12345.678f) out[1] = 1;                                                                                                \
        if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                            \
    }

#define K_W64(NAME, TYPE, INIT, F)                                                                                                      \
    __global__ __launch_bounds__(512) void NAME(unsigned long long *out, float seed, int iters)                                         \
    {                                                                                                                                   \
        TYPE r[8];                                                                                                                      \
        for (int i = 0; i < 8; ++i) { const float v = seed + (float)(threadIdx.x * 8 + i); r[i] = INIT; }                               \
        const float v = seed * 1.0001f;                                                                                                 \
        TYPE y = INIT;                                                                                                                  \
        float z = seed;                                                                                                                 \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                     \
        for (int it = 0; it < iters; ++it)                                                                                              \
            asm volatile(LOOP32(F) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(y), "v"(z) : "vcc", "s20", "s21"); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                     \
        double s = 0.;                                                                                                                  \
        for (int i = 0; i < 8; ++i) s += *(double *)&r[i];                                                                              \
        if (s == 12345.678) out[1] = 1;                                                                                                 \
        if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                            \
    }

#define F_MUL(n) "v_mul_f32 %" n ", %" n ", %8\n"
#define F_ADD(n) "v_add_f32 %" n ", %" n ", %8\n"
#define F_FMA(n) "v_fma_f32 %" n ", %" n ", %8, %9\n"
#define F_CND(n) "v_cndmask_b32 %" n ", %" n ", %8, vcc\n"
#define F_CMPV(n) "v_cmp_lt_f32 vcc, %" n ", %8\n"
#define F_CMPS(n) "v_cmp_lt_f32 s[20:21], %" n ", %8\n"
#define F_CMPEQ(n) "v_cmp_eq_u32 vcc, %" n ", %8\n"
#define F_MIN3(n) "v_min3_f32 %" n ", %" n ", %8, %9\n"
#define F_MED3(n) "v_med3_f32 %" n ", %" n ", %8, %9\n"
#define F_MAX(n) "v_max_f32 %" n ", %" n ", %8\n"
#define F_MULLO(n) "v_mul_lo_u32 %" n ", %" n ", %8\n"
#define F_MOV(n) "v_mov_b32 %" n ", %8\n"
#define F_SQRT(n) "v_sqrt_f32 %" n ", %" n "\n"
#define F_FLOOR(n) "v_floor_f32 %" n ", %" n "\n"
#define F_FRACT(n) "v_fract_f32 %" n ", %" n "\n"
#define F_CVTI(n) "v_cvt_i32_f32 %" n ", %" n "\n"
#define F_CVTF(n) "v_cvt_f32_i32 %" n ", %" n "\n"
#define F_DPP(n) "v_mov_b32_dpp %" n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_ADDU(n) "v_add_u32 %" n ", %" n ", %8\n"
#define F_ADD3(n) "v_add3_u32 %" n ", %" n ", %8, %9\n"
#define F_AND(n) "v_and_b32 %" n ", %" n ", %8\n"
#define F_BFE(n) "v_bfe_u32 %" n ", %" n ", 1, 5\n"
#define F_MINU(n) "v_min_u32 %" n ", %" n ", %8\n"
#define F_SWAP(n) "v_permlane32_swap_b32 %" n ", %" n "\n"
#define F_MBCNT(n) "v_mbcnt_lo_u32_b32 %" n ", %8, %" n "\n"
#define F_LSHLADD(n) "v_lshl_add_u32 %" n ", %" n ", 2, %8\n"
#define F_NOP(n) "s_nop 0\n"
#define F_CND64(n) "v_cndmask_b32_e64 %" n ", %" n ", %8, s[20:21]\n"
#define F_CNDOUT(n) "v_cndmask_b32 %" n ", %8, %9, vcc\n"
#define F_CMPCND(n) "v_cmp_lt_f32 vcc, %" n ", %8\nv_cndmask_b32 %" n ", %" n ", %8, vcc\n"
#define F_CMPCND64(n) "v_cmp_lt_f32 s[20:21], %" n ", %8\nv_cndmask_b32_e64 %" n ", %" n ", %8, s[20:21]\n"
#define F_MINMAX(n) "v_min_f32 %" n ", %" n ", %8\nv_max_f32 %" n ", %" n ", %9\n"
#define F_SUBMUL(n) "v_sub_f32 %" n ", %" n ", %8\nv_mul_f32 %" n ", %" n ", %" n "\n"

#define P_MUL(n) "v_pk_mul_f32 %" n ", %" n ", %8\n"
#define P_ADD(n) "v_pk_add_f32 %" n ", %" n ", %8\n"
#define P_FMA(n) "v_pk_fma_f32 %" n ", %" n ", %8, %8\n"
#define D_ADD(n) "v_add_f64 %" n ", %" n ", %8\n"
#define D_MUL(n) "v_mul_f64 %" n ", %" n ", %8\n"
#define D_FMA(n) "v_fma_f64 %" n ", %" n ", %8, %8\n"
#define D_CMPU64(n) "v_cmp_lt_u64 vcc, %" n ", %8\n"
#define D_CMPF64(n) "v_cmp_lt_f64 vcc, %" n ", %8\n"
#define D_MOV64(n) "v_mov_b64 %" n ", %8\n"
#define D_LSHLADD64(n) "v_lshl_add_u64 %" n ", %" n ", 1, %8\n"
#define D_MAD64(n) "v_mad_u64_u32 %" n ", vcc, %9, %9, %" n "\n"
#define D_CVT_F64_F32(n) "v_cvt_f64_f32 %" n ", %9\n"
#define D_MIN64(n) "v_min_f64 %" n ", %" n ", %8\n"
#define D_DPP64(n) "v_mov_b64_dpp %" n ", %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"

K_F32(k_mul, F_MUL) K_F32(k_add, F_ADD) K_F32(k_fma, F_FMA) K_F32(k_cnd, F_CND) K_F32(k_cmpv, F_CMPV) K_F32(k_cmps, F_CMPS) K_F32(k_cmpeq, F_CMPEQ)
K_F32(k_min3, F_MIN3) K_F32(k_med3, F_MED3) K_F32(k_max, F_MAX) K_F32(k_mullo, F_MULLO) K_F32(k_mov, F_MOV) K_F32(k_sqrt, F_SQRT)
K_F32(k_floor, F_FLOOR) K_F32(k_fract, F_FRACT) K_F32(k_cvti, F_CVTI) K_F32(k_cvtf, F_CVTF) K_F32(k_dpp, F_DPP) K_F32(k_addu, F_ADDU) K_F32(k_add3, F_ADD3)
K_F32(k_and, F_AND) K_F32(k_bfe, F_BFE) K_F32(k_minu, F_MINU) K_F32(k_swap, F_SWAP) K_F32(k_mbcnt, F_MBCNT) K_F32(k_lshladd, F_LSHLADD) K_F32(k_nop, F_NOP) K_F32(k_cnd64, F_CND64) K_F32(k_cndout, F_CNDOUT) K_F32(k_cmpcnd, F_CMPCND) K_F32(k_cmpcnd64, F_CMPCND64) K_F32(k_minmax, F_MINMAX) K_F32(k_submul, F_SUBMUL)
K_W64(k_pkmul, f2, (f2{v, v * 2.f}), P_MUL) K_W64(k_pkadd, f2, (f2{v, v * 2.f}), P_ADD) K_W64(k_pkfma, f2, (f2{v, v * 2.f}), P_FMA)
K_W64(k_dadd, double, (double)v, D_ADD) K_W64(k_dmul, double, (double)v, D_MUL) K_W64(k_dfma, double, (double)v, D_FMA)
K_W64(k_cmpu64, double, (double)v, D_CMPU64) K_W64(k_cmpf64, double, (double)v, D_CMPF64) K_W64(k_mov64, double, (double)v, D_MOV64)
K_W64(k_lshladd64, double, (double)v, D_LSHLADD64) K_W64(k_mad64, double, (double)v, D_MAD64) K_W64(k_cvtdf, double, (double)v, D_CVT_F64_F32)
K_W64(k_dmin, double, (double)v, D_MIN64)

// LDS: 64-bit atomic min without return and 16-byte reads at per-lane addresses (the task loop's traffic)
__global__ __launch_bounds__(512) void k_lds(unsigned long long *out, float seed, int iters)
{
    __shared__ unsigned long long best[512];
    __shared__ float4 q[512];
    best[threadIdx.x] = ~0ull;
    q[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    unsigned long long v = ((unsigned long long)__float_as_uint(seed) << 32) | threadIdx.x;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = (threadIdx.x * 7 + u * 13 + it) & 511;
            atomicMin(&best[o], v - u);
            acc += q[o].x;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345.678f) out[1] = best[3];
    if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long *, float, int);
struct Entry { const char *name; kern_t k; int per_iter; };

int main()
{
    const Entry ents[] = {
        {"s_nop 0", k_nop, 32}, {"v_mov_b32", k_mov, 32}, {"v_mul_f32", k_mul, 32}, {"v_add_f32", k_add, 32}, {"v_fma_f32", k_fma, 32}, {"v_max_f32", k_max, 32},
        {"v_min3_f32", k_min3, 32}, {"v_med3_f32", k_med3, 32}, {"v_cndmask_b32(vcc)", k_cnd, 32}, {"v_cndmask_b32_e64(sgpr pair)", k_cnd64, 32}, {"v_cndmask_b32 d!=src (vcc)", k_cndout, 32}, {"v_cmp->vcc + v_cndmask (pair)", k_cmpcnd, 32}, {"v_cmp->sgpr + v_cndmask_e64 (pair)", k_cmpcnd64, 32}, {"v_min_f32 + v_max_f32 dependent (pair)", k_minmax, 32}, {"v_sub_f32 + v_mul_f32 dependent (pair)", k_submul, 32}, {"v_cmp_lt_f32->vcc", k_cmpv, 32}, {"v_cmp_lt_f32->sgpr", k_cmps, 32},
        {"v_cmp_eq_u32->vcc", k_cmpeq, 32}, {"v_add_u32", k_addu, 32}, {"v_add3_u32", k_add3, 32}, {"v_and_b32", k_and, 32}, {"v_bfe_u32", k_bfe, 32}, {"v_min_u32", k_minu, 32},
        {"v_lshl_add_u32", k_lshladd, 32}, {"v_mul_lo_u32", k_mullo, 32}, {"v_floor_f32", k_floor, 32}, {"v_fract_f32", k_fract, 32}, {"v_cvt_i32_f32", k_cvti, 32},
        {"v_cvt_f32_i32", k_cvtf, 32}, {"v_sqrt_f32", k_sqrt, 32}, {"v_mov_b32_dpp", k_dpp, 32}, {"v_permlane32_swap", k_swap, 32}, {"v_mbcnt_lo", k_mbcnt, 32},
        {"v_pk_mul_f32", k_pkmul, 32}, {"v_pk_add_f32", k_pkadd, 32}, {"v_pk_fma_f32", k_pkfma, 32},
        {"v_add_f64", k_dadd, 32}, {"v_mul_f64", k_dmul, 32}, {"v_fma_f64", k_dfma, 32}, {"v_min_f64", k_dmin, 32}, {"v_cmp_lt_u64->vcc", k_cmpu64, 32}, {"v_cmp_lt_f64->vcc", k_cmpf64, 32},
        {"v_mov_b64", k_mov64, 32}, {"v_lshl_add_u64", k_lshladd64, 32}, {"v_mad_u64_u32", k_mad64, 32}, {"v_cvt_f64_f32", k_cvtdf, 32},
        {"ds_min_u64 + ds_read_b128 pair", k_lds, 8},
    };
    unsigned long long *d_out;
    const int max_waves = 256 * 4 * 8;
    CK(hipMalloc(&d_out, (2 + max_waves) * 8));
    std::vector<unsigned long long> h(2 + max_waves);
    const int iters = 2000;
    printf("%-32s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD)\n", "instruction", "1 w/SIMD", "2", "4", "8");
    for (const Entry &e : ents) {
        printf("%-32s", e.name);
        for (int wps : {1, 2, 4, 8}) {
            // one workgroup per CU: 4 * wps waves, so every SIMD of the CU holds wps of them; 256 CUs
            const int threads = 64 * 4 * wps;
            const int blocks = threads <= 512 ? 256 : 256; // 1024-thread blocks are not used: two 512-thread blocks per CU at wps = 4 / 8
            const int bt = threads <= 512 ? threads : 512;
            const int nb = blocks * (threads / bt);
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(e.k, dim3(nb), dim3(bt), 0, 0, d_out, 1.5f, iters);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), d_out, (2 + nb * (bt / 64)) * 8, hipMemcpyDeviceToHost));
                std::vector<unsigned long long> t(h.begin() + 2, h.begin() + 2 + nb * (bt / 64));
                std::sort(t.begin(), t.end());
                const double med = (double)t[t.size() / 2];
                best = std::min(best, med / ((double)iters * e.per_iter * wps));
            }
            printf(" %8.2f", best);
        }
        printf("\n");
    }
    return 0;
}
