// dpp_probe.hip -- issue rate of the DPP forms the streaming kernels lean on (gfx950).
// Each kernel runs a long chain of one instruction form; all 256 CUs x 8 waves per SIMD busy.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f + 1.f, c = 3.f, d = 4.f;
    const unsigned long long msk = 0x5555555555555555ull ^ (unsigned long long)blockIdx.x;
    const int mv = (threadIdx.x * 4) & 255;
    unsigned sc = blockIdx.x;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa = {a, b}, pb = {c, d}, pc = {b, a}, pd = {d, c};
    float e = 0.f, g = 0.f; int mvv = mv;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) { asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 1) { asm volatile("v_max_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_f32_dpp %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_f32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_f32_dpp %3, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 2) { asm volatile("v_max_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_f32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 3) { asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 4) { asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 5) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc"); }
            if (MODE == 6) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %4\n v_cndmask_b32_e64 %1, %1, %2, %4\n v_cndmask_b32_e64 %2, %2, %3, %4\n v_cndmask_b32_e64 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(msk)); }
            if (MODE == 7) { asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(mv)); }
            if (MODE == 8) { asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_cmp_ge_f32 vcc, %1, %2\n v_cmp_ge_f32 vcc, %2, %3\n v_cmp_ge_f32 vcc, %3, %0" :: "v"(a), "v"(b), "v"(c), "v"(d) : "vcc"); }
            if (MODE == 9) { asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 10) { asm volatile("v_add_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 11) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 12) { asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_ge_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc"); }
            if (MODE == 13) { asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(mv)); }
            if (MODE == 14) { asm volatile("v_pk_max_f16 %0, %0, %1\n v_pk_max_f16 %1, %1, %2\n v_pk_max_f16 %2, %2, %3\n v_pk_max_f16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 16) { asm volatile("v_max_f32 %0, %0, %1\n s_nop 0\n v_max_f32 %1, %1, %2\n s_nop 0\n v_max_f32 %2, %2, %3\n s_nop 0\n v_max_f32 %3, %3, %0\n s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 17) { asm volatile("v_max_f32 %0, %0, %1\n s_add_u32 %4, %4, 1\n v_max_f32 %1, %1, %2\n s_add_u32 %4, %4, 3\n v_max_f32 %2, %2, %3\n s_mul_i32 %4, %4, 5\n v_max_f32 %3, %3, %0\n s_add_u32 %4, %4, 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(sc)); }
            if (MODE == 18) { asm volatile("v_max_f32 %0, %0, %1\n s_nop 1\n v_max_f32_dpp %1, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_max_f32_dpp %2, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_max_f32_dpp %3, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 19) { asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %0, %0, %2\n v_max_f32 %0, %0, %3\n v_max_f32 %0, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

            if (MODE == 20) { asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0" : "+v"(pa), "+v"(pb)); asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0" : "+v"(pc), "+v"(pd)); }
            if (MODE == 21) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0" : "+v"(pa), "+v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0" : "+v"(pc), "+v"(pd)); }
            if (MODE == 22) { asm volatile("v_pk_mov_b32 %0, %1, %1\n v_pk_mov_b32 %1, %0, %0" : "+v"(pa), "+v"(pb)); asm volatile("v_pk_mov_b32 %0, %1, %1\n v_pk_mov_b32 %1, %0, %0" : "+v"(pc), "+v"(pd)); }
            if (MODE == 23) { asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 24) { asm volatile("v_and_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_and_b32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 25) { asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %1, %1, %2\n v_max_i32 %2, %2, %3\n v_max_i32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 26) { asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 27) { asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 28) { asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 29) { asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 30) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 31) { asm volatile("v_max_f32_e64 %0, %0, %1\n v_max_f32_e64 %1, %1, %2\n v_max_f32_e64 %2, %2, %3\n v_max_f32_e64 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 32) { asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 33) { asm volatile("v_maximum3_f32 %0, %0, %1, %2\n v_maximum3_f32 %1, %1, %2, %3\n v_minimum3_f32 %2, %2, %3, %0\n v_minimum3_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 34) { asm volatile("v_max3_i32 %0, %0, %1, %2\n v_med3_i32 %1, %1, %2, %3\n v_min3_u32 %2, %2, %3, %0\n v_med3_u32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 35) { asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_min_i16 %1, %1, %2\n v_pk_max_u16 %2, %2, %3\n v_pk_min_u16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 36) { asm volatile("v_lshlrev_b32 %0, 1, %1\n v_lshrrev_b32 %1, 1, %2\n v_ashrrev_i32 %2, 1, %3\n v_lshlrev_b32 %3, 1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 37) { asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_subrev_f32 %2, %2, %3\n v_sub_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 38) { asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 39) { asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 40) { asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_i32 vcc, %1, %2\n v_cmp_eq_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0" :: "v"(a), "v"(b), "v"(c), "v"(d) : "vcc"); }
            if (MODE == 41) { asm volatile("v_mul_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %2, %3, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %3, %0, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 42) { asm volatile("v_max_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 43) { asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); asm volatile("ds_read_b32 %0, %1\n ds_read_b32 %2, %1 offset:256" : "=v"(e), "+v"(mvv), "=v"(g)); }
            if (MODE == 44) { asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(pa), "+v"(pb)); asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(pc), "+v"(pd)); }
            if (MODE == 45) { asm volatile("v_cvt_f32_u32 %0, %1\n v_cvt_u32_f32 %1, %2\n v_cvt_f32_i32 %2, %3\n v_cvt_i32_f32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 46) { asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n s_setprio 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (MODE == 15) { asm volatile("v_max_u32 %0, %0, %1\n v_max_i32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_i32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (float)sc + pa.x + pa.y + pb.x + pb.y + pc.x + pd.y + e + g;
}

template <int MODE> void run(const char* name, float* d)
{
    const int blocks = 256 * 8, iters = 2000;       // 8 workgroups of 4 waves per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 64;     // wave-instructions
    printf("%-28s %.3f ms  %.2f T wave-instr/s  (%.2f cycles per instr per SIMD at 2.4 GHz)\n", name, ms, insts / ms / 1e9,
           1024.0 * 2.4e9 * ms * 1e-3 / insts);
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_max_f32", d);
    run<1>("v_max_f32_dpp wave_shr:1", d);
    run<2>("v_max_f32_dpp row_shr:1", d);
    run<3>("v_mov_b32_dpp wave_shl:1", d);
    run<4>("v_max3_f32", d);
    run<5>("v_cndmask_b32 vcc", d);
    run<6>("v_cndmask_b32_e64 sgpr", d);
    run<7>("v_bfi_b32", d);
    run<8>("v_cmp_ge_f32 -> vcc", d);
    run<9>("v_med3_f32", d);
    run<10>("v_add_f32 / v_mul_f32", d);
    run<11>("v_mov_b32", d);
    run<12>("v_cmp + v_cndmask pairs", d);
    run<13>("ds_bpermute_b32", d);
    run<14>("v_pk_max_f16", d);
    run<15>("v_max/min_u32/i32", d);
    run<16>("v_max + s_nop 0 alternating", d);
    run<17>("v_max + SALU alternating", d);
    run<18>("dependent DPP chain + s_nop 1", d);
    run<19>("dependent v_max chain", d);

    run<20>("v_pk_add_f32 (2 flop/lane)", d);
    run<21>("v_pk_fma_f32", d);
    run<44>("v_pk_mul_f32", d);
    run<22>("v_pk_mov_b32", d);
    run<23>("v_add_u32", d);
    run<24>("v_and/or/xor_b32", d);
    run<25>("v_max_i32", d);
    run<26>("v_min_u32", d);
    run<27>("v_fma_f32", d);
    run<28>("v_fmac_f32", d);
    run<29>("v_add_f32_dpp wave_shr:1", d);
    run<30>("v_mov_b32_dpp row_shr:1", d);
    run<31>("v_max_f32_e64", d);
    run<32>("v_min_f32", d);
    run<33>("v_maximum3/minimum3_f32", d);
    run<34>("v_max3/med3/min3 i32/u32", d);
    run<35>("v_pk_max/min_i16/u16", d);
    run<36>("shifts", d);
    run<37>("v_sub_f32", d);
    run<38>("v_bitop3_b32", d);
    run<39>("v_permlane32/16_swap", d);
    run<40>("v_cmp int -> vcc", d);
    run<41>("v_mul_f32_dpp mixed", d);
    run<42>("v_max_f32 / v_add_f32 alternating", d);
    run<43>("2 v_max + 2 ds_read_b32", d);
    run<45>("v_cvt", d);
    run<46>("v_max x4 + s_setprio", d);
    return 0;
}
