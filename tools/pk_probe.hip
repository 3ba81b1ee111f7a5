// pk_probe.hip -- issue cost of the instructions k_fp_q (16-bit codes, dcmt_kernels_fp_q16.h) is made of, measured like
// tools/issue_probe.hip: cycles per wave64 instruction per SIMD at 8 / 4 / 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pk_probe tools/pk_probe.hip && ./pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters)
{
    unsigned a = threadIdx.x * 3u, b = threadIdx.x * 5u + 1u, c = 3u + threadIdx.x, d = 4u;
    unsigned e = 5u + threadIdx.x, f = 6u, g = 7u, h = 8u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#define OPS8 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
            if (MODE == 0) asm volatile("v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %1, %1, %2\n v_pk_max_u16 %2, %2, %3\n v_pk_max_u16 %3, %3, %0" OPS8);
            if (MODE == 1) asm volatile("v_pk_min_u16 %4, %0, %1\n v_pk_max_u16 %5, %0, %1\n v_pk_min_u16 %6, %2, %3\n v_pk_max_u16 %7, %2, %3\n"
                                        "v_pk_min_u16 %0, %4, %6\n v_pk_max_u16 %2, %4, %6\n v_pk_min_u16 %1, %5, %7\n v_pk_max_u16 %3, %5, %7" OPS8);   // network of 4 exchanges
            if (MODE == 2) asm volatile("v_min_f32 %4, %0, %1\n v_max_f32 %5, %0, %1\n v_min_f32 %6, %2, %3\n v_max_f32 %7, %2, %3\n"
                                        "v_min_f32 %0, %4, %6\n v_max_f32 %2, %4, %6\n v_min_f32 %1, %5, %7\n v_max_f32 %3, %5, %7" OPS8);               // the same in f32
            if (MODE == 3) asm volatile("v_max_u32 %0, %0, %1\n v_max_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_max_u32 %3, %3, %0" OPS8);
            if (MODE == 4) asm volatile("v_max3_u32 %0, %0, %1, %2\n v_max3_u32 %1, %1, %2, %3\n v_max3_u32 %2, %2, %3, %0\n v_max3_u32 %3, %3, %0, %1" OPS8);
            if (MODE == 5) asm volatile("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %1, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n"
                                        "v_max_u32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %3, %3, %3 row_shl:2 row_mask:0xf bank_mask:0xf" OPS8);
            if (MODE == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 16\n v_alignbit_b32 %1, %1, %2, 16\n v_alignbit_b32 %2, %2, %3, 16\n v_alignbit_b32 %3, %3, %0, 16" OPS8);
            if (MODE == 7) asm volatile("v_lshl_or_b32 %0, %0, 16, %1\n v_lshl_or_b32 %1, %1, 16, %2\n v_lshl_or_b32 %2, %2, 16, %3\n v_lshl_or_b32 %3, %3, 16, %0" OPS8);
            if (MODE == 8) asm volatile("v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_bfi_b32 %2, %2, %3, %0\n v_bfi_b32 %3, %3, %0, %1" OPS8);
            if (MODE == 9) asm volatile("v_cvt_f32_u32 %0, %1\n v_cvt_f32_u32 %1, %2\n v_cvt_f32_u32 %2, %3\n v_cvt_f32_u32 %3, %0" OPS8);
            if (MODE == 10) asm volatile("v_pk_max_u16 %0, %0, %1\n v_add_f32 %4, %4, %5\n v_pk_max_u16 %1, %1, %2\n v_add_f32 %5, %5, %6" OPS8);          // packed beside f32 adds
            if (MODE == 11) asm volatile("v_pk_max_u16 %0, %0, %1\n v_max_f32 %4, %4, %5\n v_pk_min_u16 %1, %1, %2\n v_max_f32 %5, %5, %6" OPS8);          // packed beside f32 max
            if (MODE == 12) asm volatile("v_cmp_le_u32 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_le_u32 vcc, %1, %2\n v_cndmask_b32 %5, %5, %6, vcc" OPS8 :: "vcc");
            if (MODE == 13) asm volatile("v_and_b32 %0, 0xffff, %1\n v_lshrrev_b32 %1, 16, %2\n v_and_b32 %2, 0xffff, %3\n v_lshrrev_b32 %3, 16, %0" OPS8);
            if (MODE == 14) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_pk_max_u16 %4, %4, %5\n v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_pk_min_u16 %5, %5, %6" OPS8);
            if (MODE == 15) asm volatile("v_pk_min_u16 %4, %0, %1\n v_pk_max_u16 %0, %0, %1\n v_pk_min_u16 %1, %4, %2\n v_pk_max_u16 %2, %4, %2\n"
                                         "v_pk_min_u16 %4, %0, %3\n v_pk_max_u16 %3, %0, %3\n v_pk_min_u16 %0, %1, %4\n v_pk_max_u16 %1, %1, %4" OPS8);  // a dependent chain of exchanges
            if (MODE == 16) asm volatile("v_min_f32 %4, %0, %1\n v_max_f32 %0, %0, %1\n v_min_f32 %1, %4, %2\n v_max_f32 %2, %4, %2\n"
                                         "v_min_f32 %4, %0, %3\n v_max_f32 %3, %0, %3\n v_min_f32 %0, %1, %4\n v_max_f32 %1, %1, %4" OPS8);              // the same chain in f32
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

struct Mode { const char* name; int n; };
template <int MODE> void run(const Mode& m, unsigned* d, int wg_per_cu)
{
    const int blocks = 256 * wg_per_cu, iters = 1000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = 2.4e9 * ms * 1e-3 / ((double)wg_per_cu * iters * 8);
    printf("%-46s %d waves/SIMD  %6.2f cycles per block  (%5.2f per instruction, %d instr)\n", m.name, wg_per_cu, cyc, cyc / m.n, m.n);
}

int main()
{
    unsigned* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    static const Mode M[] = {{"4 v_pk_max_u16", 4}, {"network: 4 x (v_pk_min_u16 + v_pk_max_u16)", 8}, {"network: 4 x (v_min_f32 + v_max_f32)", 8}, {"4 v_max_u32", 4},
        {"4 v_max3_u32", 4}, {"4 v_max_u32_dpp", 4}, {"4 v_alignbit_b32", 4}, {"4 v_lshl_or_b32", 4}, {"4 v_bfi_b32", 4}, {"4 v_cvt_f32_u32", 4},
        {"2 v_pk_max_u16 + 2 v_add_f32", 4}, {"2 v_pk + 2 v_max_f32", 4}, {"2 x (v_cmp_le_u32, v_cndmask)", 4}, {"v_and / v_lshrrev", 4},
        {"2 v_mov_dpp + 2 v_pk", 4}, {"chain: 4 dependent packed exchanges", 8}, {"chain: 4 dependent f32 exchanges", 8}};
#define RUN(I, W) run<I>(M[I], d, W)
#define ALLW(I) RUN(I, 8); RUN(I, 4); RUN(I, 3)
    ALLW(0); ALLW(1); ALLW(2); ALLW(3); ALLW(4); ALLW(5); ALLW(6); ALLW(7); ALLW(8); ALLW(9); ALLW(10); ALLW(11); ALLW(12); ALLW(13); ALLW(14); ALLW(15); ALLW(16);
    RUN(15, 1); RUN(16, 1);
    return 0;
}
