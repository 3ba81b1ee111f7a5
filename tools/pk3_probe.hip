// pk3_probe.hip -- round 3: do the packed THREE-input f16 instructions gfx950 has (v_pk_maximum3_f16 / v_pk_minimum3_f16, VOP3P)
// (a) order 16-bit codes like unsigned integers -- in the positive normal range 0x0400..0x7bff, and in the subnormal range below it --
// and (b) issue at the price of one v_pk_max_u16?  Beside them: v_bitop3_b32 (for med3 = a ^ b ^ c ^ min3 ^ max3), the op_sel forms
// (a half of a register chosen per operand: the neighbour shifts of a packed pair), the packed f32 arithmetic, v_permlane*_swap.
//   hipcc --offload-arch=gfx950 -O3 -o tools/pk3_probe tools/pk3_probe.hip && tools/pk3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// ---------------- (a) semantics
__global__ void k_sem(const unsigned* a, const unsigned* b, const unsigned* c, unsigned* mx, unsigned* mn, unsigned* sel, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = a[i], y = b[i], z = c[i], r1, r2, r3;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r1) : "v"(x), "v"(y), "v"(z));
    asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r2) : "v"(x), "v"(y), "v"(z));
    // op_sel: low result = max3(x.hi, y.lo, z.lo), high result = max3(x.lo, y.hi, z.hi)
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r3) : "v"(x), "v"(y), "v"(z));
    mx[i] = r1; mn[i] = r2; sel[i] = r3;
}

static unsigned short max3h(unsigned short a, unsigned short b, unsigned short c) { unsigned short m = a > b ? a : b; return m > c ? m : c; }
static unsigned short min3h(unsigned short a, unsigned short b, unsigned short c) { unsigned short m = a < b ? a : b; return m < c ? m : c; }

static void semantics(unsigned lo, unsigned hi, const char* name)
{
    const int n = 1 << 20;
    unsigned *ha = (unsigned*)malloc(n * 4), *hb = (unsigned*)malloc(n * 4), *hc = (unsigned*)malloc(n * 4);
    unsigned *h1 = (unsigned*)malloc(n * 4), *h2 = (unsigned*)malloc(n * 4), *h3 = (unsigned*)malloc(n * 4);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 20); };
    auto code = [&]() { return lo + rnd() % (hi - lo + 1); };
    for (int i = 0; i < n; ++i) { ha[i] = code() | (code() << 16); hb[i] = code() | (code() << 16); hc[i] = code() | (code() << 16); }
    // near ties and the range ends
    for (int i = 0; i < 4096; ++i) { unsigned v = lo + (i % (hi - lo + 1)); ha[i] = v | (hi << 16); hb[i] = (v + (v < hi)) | (lo << 16); hc[i] = lo | ((hi - (hi > lo)) << 16); }
    unsigned *a, *b, *c, *d1, *d2, *d3;
    (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4); (void)hipMalloc(&c, n * 4); (void)hipMalloc(&d1, n * 4); (void)hipMalloc(&d2, n * 4); (void)hipMalloc(&d3, n * 4);
    (void)hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(n / 256), dim3(256), 0, 0, a, b, c, d1, d2, d3, n);
    (void)hipMemcpy(h1, d1, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h2, d2, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h3, d3, n * 4, hipMemcpyDeviceToHost);
    long bad1 = 0, bad2 = 0, bad3 = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned short al = ha[i], ah = ha[i] >> 16, bl = hb[i], bh = hb[i] >> 16, cl = hc[i], ch = hc[i] >> 16;
        const unsigned w1 = max3h(al, bl, cl) | ((unsigned)max3h(ah, bh, ch) << 16), w2 = min3h(al, bl, cl) | ((unsigned)min3h(ah, bh, ch) << 16);
        const unsigned w3 = max3h(ah, bl, cl) | ((unsigned)max3h(al, bh, ch) << 16);
        bad1 += h1[i] != w1; bad2 += h2[i] != w2; bad3 += h3[i] != w3;
        if ((h1[i] != w1 || h2[i] != w2 || h3[i] != w3) && bad1 + bad2 + bad3 <= 3)
            printf("   e.g. a=%08x b=%08x c=%08x: max3 %08x (want %08x) min3 %08x (want %08x) sel %08x (want %08x)\n", ha[i], hb[i], hc[i], h1[i], w1, h2[i], w2, h3[i], w3);
    }
    printf("semantics %-34s codes 0x%04x..0x%04x: pk_maximum3 wrong %ld, pk_minimum3 wrong %ld, op_sel form wrong %ld of %d\n", name, lo, hi, bad1, bad2, bad3, n);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(d3);
    free(ha); free(hb); free(hc); free(h1); free(h2); free(h3);
}

// ---------------- (b) issue cost
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters)
{
    unsigned a = 0x0400u + threadIdx.x * 3u, b = 0x0400u + threadIdx.x * 5u + 1u, c = 0x0403u + threadIdx.x, d = 0x0404u;
    unsigned e = 0x0405u + threadIdx.x, f = 0x0406u, g = 0x0407u, h = 0x0408u;
    a |= a << 16; b |= b << 16; c |= c << 16; d |= d << 16; e |= e << 16; f |= f << 16; g |= g << 16; h |= h << 16;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p = {1.0f + threadIdx.x, 2.0f}, q = {0.5f, 0.25f}, r = {3.0f, 1.0f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#define OPS8 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
            if (MODE == 0) asm volatile("v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %1, %1, %2\n v_pk_max_u16 %2, %2, %3\n v_pk_max_u16 %3, %3, %0" OPS8);
            if (MODE == 1) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2\n v_pk_maximum3_f16 %1, %1, %2, %3\n v_pk_maximum3_f16 %2, %2, %3, %0\n v_pk_maximum3_f16 %3, %3, %0, %1" OPS8);
            if (MODE == 2) asm volatile("v_pk_minimum3_f16 %4, %0, %1, %2\n v_pk_maximum3_f16 %5, %0, %1, %2\n v_pk_minimum3_f16 %6, %1, %2, %3\n v_pk_maximum3_f16 %7, %1, %2, %3\n"
                                        "v_pk_minimum3_f16 %0, %4, %6, %5\n v_pk_maximum3_f16 %2, %4, %6, %7\n v_pk_minimum3_f16 %1, %5, %7, %4\n v_pk_maximum3_f16 %3, %5, %7, %6" OPS8);
            if (MODE == 3) asm volatile("v_pk_max_f16 %0, %0, %1\n v_pk_min_f16 %1, %1, %2\n v_pk_max_f16 %2, %2, %3\n v_pk_min_f16 %3, %3, %0" OPS8);
            if (MODE == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96" OPS8);
            // sort3 with the xor median: min3, max3, two bitop3
            if (MODE == 5) asm volatile("v_pk_minimum3_f16 %4, %0, %1, %2\n v_pk_maximum3_f16 %5, %0, %1, %2\n v_bitop3_b32 %6, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %6, %6, %4, %5 bitop3:0x96\n"
                                        "v_pk_minimum3_f16 %0, %4, %6, %3\n v_pk_maximum3_f16 %1, %5, %6, %3\n v_bitop3_b32 %7, %4, %6, %3 bitop3:0x96\n v_bitop3_b32 %2, %7, %0, %1 bitop3:0x96" OPS8);
            if (MODE == 6) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n v_pk_maximum3_f16 %1, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n"
                                        "v_pk_minimum3_f16 %2, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n v_pk_minimum3_f16 %3, %3, %0, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" OPS8);
            if (MODE == 7) asm volatile("v_pk_max_u16 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_max_u16 %1, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]\n"
                                        "v_pk_min_u16 %2, %2, %3 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_min_u16 %3, %3, %0 op_sel:[1,0] op_sel_hi:[0,1]" OPS8);
            // three-input packed beside cheap-class logic and f32 adds
            if (MODE == 8) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2\n v_xor_b32 %4, %4, %5\n v_pk_minimum3_f16 %1, %1, %2, %3\n v_xor_b32 %5, %5, %6" OPS8);
            if (MODE == 9) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2\n v_add_f32 %4, %4, %5\n v_pk_minimum3_f16 %1, %1, %2, %3\n v_add_f32 %5, %5, %6" OPS8);
            if (MODE == 10) asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_fma_f32 %1, %1, %2, %0\n v_pk_mul_f32 %2, %2, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p), "+v"(q), "+v"(r));
            if (MODE == 11) asm volatile("v_pk_maximum3_f16 %3, %3, %4, %5\n v_pk_fma_f32 %1, %1, %2, %0\n v_pk_minimum3_f16 %4, %4, %5, %3\n v_pk_add_f32 %0, %0, %2" : "+v"(p), "+v"(q), "+v"(r), "+v"(a), "+v"(b), "+v"(c));
            if (MODE == 12) asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7" OPS8);
            if (MODE == 13) asm volatile("v_max3_u32 %0, %0, %1, %2\n v_min3_u32 %1, %1, %2, %3\n v_med3_u32 %2, %2, %3, %0\n v_max3_u32 %3, %3, %0, %1" OPS8);
            if (MODE == 14) asm volatile("v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n v_max_u16_sdwa %1, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n"
                                         "v_min_u16_sdwa %2, %2, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n v_min_u16_sdwa %3, %3, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1" OPS8);
            if (MODE == 15) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                         "v_cvt_f32_f16_sdwa %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" OPS8);
            if (MODE == 16) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                                         "v_mov_b32_dpp %2, %3 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_bcast:15 row_mask:0xf bank_mask:0xf" OPS8);
            if (MODE == 17) asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1" OPS8);
            if (MODE == 18) asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_sub_u16 %1, %1, %2 clamp\n v_pk_add_u16 %2, %2, %3\n v_pk_sub_u16 %3, %3, %0 clamp" OPS8);
            if (MODE == 19) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %0" OPS8);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (unsigned)(p.x + p.y + q.x + q.y + r.x + r.y);
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
    printf("%-58s %d waves/SIMD  %6.2f cycles per block  (%5.2f per instruction, %d instr)\n", m.name, wg_per_cu, cyc, cyc / m.n, m.n);
}

int main()
{
    semantics(0x0400, 0x7bff, "(positive normal f16)");
    semantics(0x0000, 0x03ff, "(zero and subnormal f16)");
    semantics(0x0000, 0x7c00, "(0 .. +inf)");
    semantics(0x0000, 0xffff, "(all 16-bit patterns: expected wrong)");
    unsigned* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    static const Mode M[] = {{"4 v_pk_max_u16", 4}, {"4 v_pk_maximum3_f16", 4}, {"network: 4 x (v_pk_minimum3_f16 + v_pk_maximum3_f16)", 8}, {"2 v_pk_max_f16 + 2 v_pk_min_f16", 4},
        {"4 v_bitop3_b32", 4}, {"2 x sort3 = (min3, max3, bitop3, bitop3)", 8}, {"4 v_pk_{max,min}imum3_f16 with op_sel", 4}, {"4 v_pk_{max,min}_u16 with op_sel", 4},
        {"2 v_pk_*3_f16 + 2 v_xor", 4}, {"2 v_pk_*3_f16 + 2 v_add_f32", 4}, {"4 v_pk_{add,fma,mul}_f32", 4}, {"2 v_pk_*3_f16 + 2 v_pk_*_f32", 4}, {"4 v_permlane{16,32}_swap", 4},
        {"4 v_{max3,min3,med3}_u32", 4}, {"4 v_{max,min}_u16_sdwa (half selects)", 4}, {"4 v_cvt_f32_f16_sdwa", 4}, {"4 v_mov_dpp (quad_perm, row_ror, mirror, bcast15)", 4},
        {"4 v_perm_b32", 4}, {"4 v_pk_{add,sub clamp}_u16", 4}, {"4 v_cvt_pk_f16_f32", 4}};
#define RUN(I, W) run<I>(M[I], d, W)
#define ALLW(I) RUN(I, 8); RUN(I, 4); RUN(I, 3)
    ALLW(0); ALLW(1); ALLW(2); ALLW(3); ALLW(4); ALLW(5); ALLW(6); ALLW(7); ALLW(8); ALLW(9); ALLW(10); ALLW(11); ALLW(12); ALLW(13); ALLW(14); ALLW(15); ALLW(16); ALLW(17); ALLW(18); ALLW(19);
    return 0;
}
