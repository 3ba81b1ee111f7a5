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
            if (MODE == 15) { asm volatile("v_max_u32 %0, %0, %1\n v_max_i32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_i32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (float)sc;
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
    return 0;
}
