// issue_probe.hip -- which VALU instructions share an issue slot on gfx950?  tools/dpp_probe.hip found two classes:
// "slow" (v_max/min/med3/cmp/cndmask/DPP/packed: ~4.4 cycles per wave64 instruction per SIMD) and "fast" (f32 add/mul/fma,
// mov, and/or/xor, bitop3, add_u32: ~2.4), and that a 1:1 mix of the two costs barely more than the slow half alone.
// This probe measures mixes at other ratios, at 8 / 2 / 1 waves per SIMD, and classifies more opcodes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define R4(s) s s s s
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f + 1.f, c = 3.f, d = 4.f;
    float e = 5.f + threadIdx.x, f = 6.f, g = 7.f, h = 8.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#define OPS8 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
            // counts: S slow, F fast per asm block (see table in main)
            if (MODE == 0) asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0" OPS8);                                     // 4S
            if (MODE == 1) asm volatile("v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %4" OPS8);                                     // 4F
            if (MODE == 2) asm volatile("v_max_f32 %0, %0, %1\n v_add_f32 %4, %4, %5\n v_max_f32 %1, %1, %2\n v_add_f32 %5, %5, %6" OPS8);                                     // 2S 2F
            if (MODE == 3) asm volatile("v_max_f32 %0, %0, %1\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_max_f32 %1, %1, %2\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %4" OPS8);   // 2S 4F
            if (MODE == 4) asm volatile("v_max_f32 %0, %0, %1\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_max_f32 %1, %1, %2\n v_add_f32 %7, %7, %4\n v_add_f32 %4, %4, %6\n v_add_f32 %5, %5, %7" OPS8);   // 2S 6F
            if (MODE == 5) asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_add_f32 %4, %4, %5\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n v_add_f32 %5, %5, %6" OPS8);   // 4S 2F
            if (MODE == 6) asm volatile("v_min_f32 %4, %0, %1\n v_bitop3_b32 %5, %0, %1, %4 bitop3:0x96\n v_min_f32 %6, %2, %3\n v_bitop3_b32 %7, %2, %3, %6 bitop3:0x96\n"
                                        "v_min_f32 %0, %4, %6\n v_bitop3_b32 %2, %4, %6, %0 bitop3:0x96\n v_min_f32 %1, %5, %7\n v_bitop3_b32 %3, %5, %7, %1 bitop3:0x96" OPS8);   // 4S 4F, dependent like a real network
            if (MODE == 7) asm volatile("v_min_f32 %4, %0, %1\n v_max_f32 %5, %0, %1\n v_min_f32 %6, %2, %3\n v_max_f32 %7, %2, %3\n"
                                        "v_min_f32 %0, %4, %6\n v_max_f32 %2, %4, %6\n v_min_f32 %1, %5, %7\n v_max_f32 %3, %5, %7" OPS8);                                      // 8S: the same network with min+max
            if (MODE == 8) asm volatile("v_ashrrev_i32 %0, 1, %1\n v_ashrrev_i32 %1, 1, %2\n v_ashrrev_i32 %2, 1, %3\n v_ashrrev_i32 %3, 1, %0" OPS8);
            if (MODE == 9) asm volatile("v_lshrrev_b32 %0, 1, %1\n v_lshrrev_b32 %1, 1, %2\n v_lshrrev_b32 %2, 1, %3\n v_lshrrev_b32 %3, 1, %0" OPS8);
            if (MODE == 10) asm volatile("v_lshlrev_b32 %0, 1, %1\n v_lshlrev_b32 %1, 1, %2\n v_lshlrev_b32 %2, 1, %3\n v_lshlrev_b32 %3, 1, %0" OPS8);
            if (MODE == 11) asm volatile("v_max_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %4, %4, %5\n v_max_f32_dpp %1, %2, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %5, %5, %6" OPS8);   // 2S(dpp) 2F
            if (MODE == 12) asm volatile("v_med3_f32 %0, %0, %1, %2\n v_xor_b32 %4, %4, %5\n v_med3_f32 %1, %1, %2, %3\n v_xor_b32 %5, %5, %6" OPS8);                             // 2S 2F
            if (MODE == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %8\n v_add_f32 %4, %4, %5\n v_cndmask_b32_e64 %1, %1, %2, %8\n v_add_f32 %5, %5, %6" OPS8 : "s"(0x5555555555555555ull));
            if (MODE == 14) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %4, %4, %5\n v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %5, %5, %6" OPS8);
            if (MODE == 15) asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1" OPS8);
            if (MODE == 16) asm volatile("v_alignbit_b32 %0, %0, %1, 8\n v_alignbit_b32 %1, %1, %2, 8\n v_alignbit_b32 %2, %2, %3, 8\n v_alignbit_b32 %3, %3, %0, 8" OPS8);
            if (MODE == 17) asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1" OPS8);
            if (MODE == 18) asm volatile("v_add3_u32 %0, %0, %1, %2\n v_lshl_add_u32 %1, %1, 2, %3\n v_add3_u32 %2, %2, %3, %0\n v_lshl_add_u32 %3, %3, 2, %1" OPS8);
            if (MODE == 19) asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0" OPS8);
            if (MODE == 20) asm volatile("v_sub_u32 %0, %0, %1\n v_subrev_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_not_b32 %3, %0" OPS8);
            if (MODE == 21) asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" OPS8);
            if (MODE == 22) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_add_f32 %4, %4, %5\n v_cmp_lt_f32 vcc, %1, %2\n v_add_f32 %5, %5, %6" OPS8 :: "vcc");                         // 2S(cmp) 2F
            if (MODE == 23) asm volatile("v_max3_f32 %0, %0, %1, %2\n v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_max3_f32 %1, %1, %2, %3\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %4" OPS8);   // 2S 4F
            if (MODE == 24) asm volatile("v_max_f32 %0, %0, %1\n v_fma_f32 %4, %4, %5, %6\n v_max_f32 %1, %1, %2\n v_fma_f32 %5, %5, %6, %7" OPS8);                                 // 2S 2F(fma)
            if (MODE == 25) asm volatile("v_max_f32 %0, %0, %1\n v_mov_b32 %4, %5\n v_max_f32 %1, %1, %2\n v_mov_b32 %5, %6" OPS8);                                               // 2S 2F(mov)
            if (MODE == 26) asm volatile("v_sub_f32 %4, %0, %1\n v_ashrrev_i32 %4, 31, %4\n v_bitop3_b32 %0, %4, %1, %0 bitop3:0xca\n v_sub_f32 %5, %2, %3\n v_ashrrev_i32 %5, 31, %5\n v_bitop3_b32 %2, %5, %3, %2 bitop3:0xca" OPS8);   // select by sign: 2 x (F, ?, F)

            if (MODE == 30) asm volatile("v_max_f32 %0, %0, %1\n v_xor_b32 %4, %4, %5\n v_max_f32 %1, %1, %2\n v_xor_b32 %5, %5, %6" OPS8);
            if (MODE == 31) asm volatile("v_med3_f32 %0, %0, %1, %2\n v_add_f32 %4, %4, %5\n v_med3_f32 %1, %1, %2, %3\n v_add_f32 %5, %5, %6" OPS8);
            if (MODE == 32) asm volatile("v_max_f32 %0, %0, %1\n v_bitop3_b32 %4, %4, %5, %6 bitop3:0x96\n v_max_f32 %1, %1, %2\n v_bitop3_b32 %5, %5, %6, %7 bitop3:0x96" OPS8);
            if (MODE == 33) asm volatile("v_min_f32 %4, %0, %1\n v_xor_b32 %5, %0, %1\n v_xor_b32 %5, %5, %4\n v_min_f32 %6, %2, %3\n v_xor_b32 %7, %2, %3\n v_xor_b32 %7, %7, %6\n"
                                         "v_min_f32 %0, %4, %6\n v_xor_b32 %2, %4, %6\n v_xor_b32 %2, %2, %0\n v_min_f32 %1, %5, %7\n v_xor_b32 %3, %5, %7\n v_xor_b32 %3, %3, %1" OPS8);   // 4 CE = 4S + 8F (VOP2 only)
            if (MODE == 34) asm volatile("v_max3_f32 %0, %0, %1, %2\n v_add_f32 %4, %4, %5\n v_max3_f32 %1, %1, %2, %3\n v_add_f32 %5, %5, %6" OPS8);
            if (MODE == 35) asm volatile("v_max_f32 %0, %0, %1\n v_and_b32 %4, %4, %5\n v_max_f32 %1, %1, %2\n v_or_b32 %5, %5, %6" OPS8);
            if (MODE == 36) asm volatile("v_max_f32 %0, %0, %1\n v_add_u32 %4, %4, %5\n v_max_f32 %1, %1, %2\n v_add_u32 %5, %5, %6" OPS8);
            if (MODE == 37) asm volatile("v_max_f32 %0, %0, %1\n v_mul_f32 %4, %4, %5\n v_max_f32 %1, %1, %2\n v_sub_f32 %5, %5, %6" OPS8);
            if (MODE == 38) asm volatile("v_min_f32 %4, %0, %1\n v_add_f32 %5, %4, %1\n v_min_f32 %6, %2, %3\n v_add_f32 %7, %6, %3\n v_min_f32 %0, %5, %7\n v_add_f32 %1, %0, %4\n v_min_f32 %2, %1, %6\n v_add_f32 %3, %2, %5" OPS8);   // 4S 4F, each add depends on the min before it
            if (MODE == 39) asm volatile("v_max_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32 %4, %5\n v_max_f32_dpp %1, %2, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32 %5, %6" OPS8);
            if (MODE == 40) asm volatile("v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %4" OPS8);
            if (MODE == 41) asm volatile("v_bitop3_b32 %4, %4, %5, %6 bitop3:0x96\n v_bitop3_b32 %5, %5, %6, %7 bitop3:0x96\n v_bitop3_b32 %6, %6, %7, %4 bitop3:0x96\n v_bitop3_b32 %7, %7, %4, %5 bitop3:0x96" OPS8);
            if (MODE == 42) asm volatile("v_med3_f32 %0, %0, %1, %2\n v_mov_b32 %4, %5\n v_med3_f32 %1, %1, %2, %3\n v_mov_b32 %5, %6" OPS8);
            if (MODE == 43) asm volatile("v_max_f32_e64 %0, %0, %1\n v_add_f32 %4, %4, %5\n v_max_f32_e64 %1, %1, %2\n v_add_f32 %5, %5, %6" OPS8);
            if (MODE == 44) asm volatile("v_med3_f32 %0, %0, %1, %2\n v_fma_f32 %4, %4, %5, %6\n v_med3_f32 %1, %1, %2, %3\n v_fma_f32 %5, %5, %6, %7" OPS8);
            if (MODE == 45) asm volatile("v_min_f32 %4, %0, %1\n v_max_f32 %5, %0, %1\n v_add_f32 %6, %6, %7\n v_min_f32 %0, %4, %2\n v_max_f32 %2, %4, %2\n v_add_f32 %7, %7, %6" OPS8);   // 4S 2F
            if (MODE == 46) asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6" OPS8);   // 2S then 2F (not alternating)
            if (MODE == 47) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_f32 %4, %4, %5\n v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_f32 %5, %5, %6" OPS8);   // dpp mov + plain max
            if (MODE == 48) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %5, %5, %6, vcc" OPS8 :: "vcc");
            if (MODE == 49) asm volatile("v_max_f32 %0, %0, %1\n v_lshrrev_b32 %4, 1, %5\n v_max_f32 %1, %1, %2\n v_ashrrev_i32 %5, 1, %6" OPS8);
            if (MODE == 27) asm volatile("v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %1, %2\n v_cvt_f32_ubyte2 %2, %3\n v_cvt_f32_ubyte3 %3, %0" OPS8);
            if (MODE == 28) asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n s_nop 3" OPS8);
            if (MODE == 29) asm volatile("v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %4\n s_nop 3" OPS8);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

struct Mode { const char* name; int n_instr; };
template <int MODE> void run(const Mode& m, float* d, int wg_per_cu)
{
    const int blocks = 256 * wg_per_cu, iters = 1000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles per asm block per SIMD: every SIMD runs wg_per_cu waves, each iters * 8 blocks
    const double cyc_block = 2.4e9 * ms * 1e-3 / ((double)wg_per_cu * iters * 8);
    printf("%-44s %d waves/SIMD  %7.3f ms  %6.2f cycles per block  (%5.2f per instruction, %d instr)\n", m.name, wg_per_cu, ms, cyc_block,
           cyc_block / m.n_instr, m.n_instr);
}

// v_min_f32 / v_max_f32 must pass a denormal's bits through untouched for the xor trick (hi = a ^ b ^ lo)
__global__ void k_denorm(unsigned* out)
{
    const float a = __builtin_bit_cast(float, 0x00000005u), b = __builtin_bit_cast(float, 0x00400000u), z = -0.0f, p = 0.0f;
    float lo, hi, lz, m3;
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(lz) : "v"(z), "v"(p));
    asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(a), "v"(b), "v"(p));
    out[0] = __builtin_bit_cast(unsigned, lo); out[1] = __builtin_bit_cast(unsigned, hi);
    out[2] = __builtin_bit_cast(unsigned, lz); out[3] = __builtin_bit_cast(unsigned, m3);
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    static const Mode M[] = {
        {"4 v_max_f32", 4}, {"4 v_add_f32", 4}, {"2 v_max + 2 v_add", 4}, {"2 v_max + 4 v_add", 6}, {"2 v_max + 6 v_add", 8}, {"4 v_max + 2 v_add", 6},
        {"network: 4 (v_min + bitop3 xor)", 8}, {"network: 4 (v_min + v_max)", 8}, {"4 v_ashrrev_i32", 4}, {"4 v_lshrrev_b32", 4}, {"4 v_lshlrev_b32", 4},
        {"2 v_max_dpp + 2 v_add", 4}, {"2 v_med3 + 2 v_xor", 4}, {"2 v_cndmask(sgpr) + 2 v_add", 4}, {"2 v_mov_dpp + 2 v_add", 4}, {"4 v_perm_b32", 4},
        {"4 v_alignbit_b32", 4}, {"v_and_or / v_or3", 4}, {"v_add3_u32 / v_lshl_add_u32", 4}, {"v_mad_u32_u24 / v_mul_u32_u24", 4}, {"v_sub_u32 / v_not", 4},
        {"4 v_mul_lo_u32", 4}, {"2 v_cmp + 2 v_add", 4}, {"2 v_max3 + 4 v_xor", 6}, {"2 v_max + 2 v_fma", 4}, {"2 v_max + 2 v_mov", 4},
        {"2 x (v_sub, v_ashrrev 31, bitop3 select)", 6}, {"4 v_cvt_f32_ubyteN", 4}, {"4 v_max + s_nop 3", 4}, {"4 v_add + s_nop 3", 4},
        {"2 v_max + 2 v_xor", 4}, {"2 v_med3 + 2 v_add", 4}, {"2 v_max + 2 v_bitop3", 4}, {"network: 4 (v_min + 2 dependent v_xor)", 12}, {"2 v_max3 + 2 v_add", 4},
        {"2 v_max + v_and + v_or", 4}, {"2 v_max + 2 v_add_u32", 4}, {"2 v_max + v_mul + v_sub", 4}, {"4 x (v_min, dependent v_add)", 8}, {"2 v_max_dpp + 2 v_mov", 4},
        {"4 v_xor", 4}, {"4 v_bitop3", 4}, {"2 v_med3 + 2 v_mov", 4}, {"2 v_max_e64 + 2 v_add", 4}, {"2 v_med3 + 2 v_fma", 4}, {"4 S (min/max pairs) + 2 v_add", 6},
        {"2 v_max, then 2 v_add", 4}, {"2 v_mov_dpp + 2 v_max", 4}, {"2 x (v_cmp, v_cndmask vcc)", 4}, {"2 v_max + v_lshr + v_ashr", 4}};
#define RUN(I, W) run<I>(M[I], d, W)
#define ALLW(I) RUN(I, 8); RUN(I, 2); RUN(I, 1)
    ALLW(0); ALLW(1); ALLW(2); ALLW(3); ALLW(4); ALLW(5); ALLW(6); ALLW(7);
    RUN(8, 8); RUN(9, 8); RUN(10, 8); RUN(11, 8); RUN(12, 8); RUN(13, 8); RUN(14, 8); RUN(15, 8); RUN(16, 8); RUN(17, 8); RUN(18, 8); RUN(19, 8);
    RUN(20, 8); RUN(21, 8); RUN(22, 8); RUN(23, 8); RUN(24, 8); RUN(25, 8); RUN(26, 8); RUN(27, 8); RUN(28, 8); RUN(29, 8);
    RUN(30, 8); RUN(31, 8); RUN(32, 8); RUN(33, 8); RUN(34, 8); RUN(35, 8); RUN(36, 8); RUN(37, 8); RUN(38, 8); RUN(39, 8); RUN(40, 8); RUN(41, 8);
    RUN(42, 8); RUN(43, 8); RUN(44, 8); RUN(45, 8); RUN(46, 8); RUN(47, 8); RUN(48, 8); RUN(49, 8); RUN(33, 4); RUN(38, 4); RUN(45, 4); RUN(7, 4);
    unsigned* u; (void)hipMalloc(&u, 16);
    hipLaunchKernelGGL(k_denorm, dim3(1), dim3(1), 0, 0, u);
    unsigned hu[4]; (void)hipMemcpy(hu, u, 16, hipMemcpyDeviceToHost);
    printf("denormals through v_min/v_max/v_med3: min %08x (want 00000005) max %08x (want 00400000) min(-0,+0) %08x med3 %08x (want 00000005)\n", hu[0], hu[1], hu[2], hu[3]);
    return 0;
}
