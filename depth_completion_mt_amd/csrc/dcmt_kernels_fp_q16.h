// dcmt_kernels_fp_q16.h -- k_fp_q: H7, H9..H11 for frames whose depths are multiples of 1/256 m (the KITTI depth format: every
// frame the reference's lidar-only and lidar-camera callers feed the path is a uint16 PNG payload / 256, LO/main.cpp:75-82).
//
// H2..H9 only ever SELECT values (max, min, median), so on such a frame every value of X2..X9 is one of
//     j / 256,   j in [-5119, 25600]         (an empty pixel k / 256 with k < 26, an inverted depth (25600 - k) / 256, or the 100
//                                             of a column without valid pixels, LO :110)
// -- with max_depth = 100 and thr = 0.1 (the reference's constants; other values run the f32 kernels).  code = j + 6143 (Q16,
// dcmt_kernels_fused.h) is 0x0400 .. 0x7bff: the map is strictly increasing, so maxima / minima / medians of codes are the codes of the
// f32 results, and a hole (x < 0.1f) is code <= 6168.  Two adjacent columns then fit ONE register (low half = column 2l, high half =
// column 2l + 1) and packed instructions work on both at the price of one v_max_f32: what k_fp_p (dcmt_kernels_fp_pair.h) gains from two
// columns per lane, without its doubled register state -- this kernel keeps 3 waves per SIMD.
//
// Which packed instructions: the code range is the bit patterns of the positive NORMAL half-floats, which order as f16 exactly as they
// do as u16.  gfx950 has packed two-input minima / maxima for u16 AND packed THREE-input ones for f16 (v_pk_maximum3_f16 /
// v_pk_minimum3_f16, VOP3P, new on gfx950; tools/pk3_probe.hip: exact on these patterns, one v_pk_max_u16's issue cost) -- but no packed
// med3 of any type.  (Round 2 wrote "no packed three-input min / max" here; that was wrong.)  So: the vertical 31-row maximum is 4
// three-input instructions per register instead of 6; the median keeps its two-input MERGE55 (26), takes MID20 with two exchanges folded
// (34, median_pk3_nets.h), sort5 from min3 / max3 and XORs (10 + 5), the closing selection with its minimum in two min3 (8): 53 + 4
// neighbour moves per row for two columns (k_fp_s: 38.5 per column with v_med3_f32).  The three-input networks of
// median_shared_nets3.h are half med3 and gain nothing here: a med3 emulated from min3 + max3 + two XORs (or two v_med3_f16 on the
// halves) costs what the exchange it replaces costs.
// Behind the median: where the redo chain follows (FILLED), Gaussian and final invert run on the codes as integer-valued floats --
// exact arithmetic, nothing rounds in the oracle's sequence either (PostPipeP::after_median_codes has the argument); otherwise the
// median is converted back (code -> f32 is exact: one v_cvt and one fused multiply-add) and the Gaussian, the masked select and the
// final invert are the f32 code of k_fp_p (PostPipeP::after_median).  Either way the output is bit-identical to k_fp_s's.
//
// Fill: the vertical 31-maximum runs packed; the horizontal one, on rows that have a hole, unpacks the two halves and is k_fp_p's
// scheme on unsigned integers (0 is the neutral element; register B holds the 30 halo columns unpacked, one per lane), with the two
// values a lane fetches from lane l - 8 in one word and the two from lane l + 8 in another: two ds_bpermutes.
//
// X6U16 = true: X6 arrives as the codes themselves (k_pre_p<Q16OUT>, which has checked the frame's values and raised a flag
// otherwise -- dcmt.hip reruns the f32 kernels behind that flag, so what this kernel makes of a frame that is no grid is never
// looked at).  X6U16 = false (X6 as f32, converted while loading) was the first stage of this work and is kept for experiments.
#pragma once

#include "dcmt_kernels_fp_pair.h"
#include "median_pk3_nets.h"

namespace dcmt {

typedef unsigned short us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned qmax(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2v, a), __builtin_bit_cast(us2v, b)));
}
__device__ __forceinline__ unsigned qmin(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2v, a), __builtin_bit_cast(us2v, b)));
}
__device__ __forceinline__ unsigned umax2(unsigned a, unsigned b) { return a > b ? a : b; }

// The codes are ordered as f16 too (Q16, dcmt_kernels_fused.h), and for f16 -- for nothing else 16 bits wide -- gfx950 has packed THREE-input
// minima / maxima: v_pk_maximum3_f16 / v_pk_minimum3_f16 (tools/pk3_probe.hip: exact on these bit patterns, the price of one v_pk_max_u16).
typedef _Float16 hf2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hmax2(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_maximum(__builtin_bit_cast(hf2v, a), __builtin_bit_cast(hf2v, b)));
}
__device__ __forceinline__ unsigned hmin2(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_minimum(__builtin_bit_cast(hf2v, a), __builtin_bit_cast(hf2v, b)));
}
__device__ __forceinline__ unsigned hmax3(unsigned a, unsigned b, unsigned c)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(hf2v, a), __builtin_bit_cast(hf2v, b)), __builtin_bit_cast(hf2v, c)));
}
__device__ __forceinline__ unsigned hmin3(unsigned a, unsigned b, unsigned c)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_bit_cast(hf2v, a), __builtin_bit_cast(hf2v, b)), __builtin_bit_cast(hf2v, c)));
}
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) { return umax2(umax2(a, b), c); }



// shifts with 0 in the lane without a source (unsigned codes: 0 is the neutral element of max)
__device__ __forceinline__ unsigned u_left(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned u_right(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true); }
// (bound_ctrl: a lane without a source reads 0 -- what `old = 0` gave, without the v_mov that sets it up)
__device__ __forceinline__ unsigned u_row_shr1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned u_row_shl1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned u_row_ror8(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true); }
__device__ __forceinline__ void u_row_scans4(unsigned a, unsigned b, unsigned& pa, unsigned& sa, unsigned& pb, unsigned& sb)
{
    // the first step writes the four scans from a and b themselves (bound_ctrl: max(0, x) = x in a lane without a source): no copies
#define DCMT_U4(N) "v_max_u32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                   "v_max_u32_dpp %1, %1, %1 row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                   "v_max_u32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                   "v_max_u32_dpp %3, %3, %3 row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t"
    // (b's scans are only looked at in lanes 0..7 (prefix) and 56..63 (suffix): eight lanes each, three steps)
    asm("s_nop 1\n\t"
        "v_max_u32_dpp %0, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_max_u32_dpp %1, %4, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_max_u32_dpp %2, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_max_u32_dpp %3, %5, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        DCMT_U4(2) DCMT_U4(4)
        "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_u32_dpp %1, %1, %1 row_shl:8 row_mask:0xf bank_mask:0xf\n\t" : "=&v"(pa), "=&v"(sa), "=&v"(pb), "=&v"(sb) : "v"(a), "v"(b));
#undef DCMT_U4
}

__device__ __forceinline__ void u_row_scans2(unsigned a, unsigned& pa, unsigned& sa)
{
#define DCMT_U2(N) "v_max_u32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                   "v_max_u32_dpp %1, %1, %1 row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t"
        "v_max_u32_dpp %0, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_max_u32_dpp %1, %2, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t" DCMT_U2(2) "s_nop 0\n\t" DCMT_U2(4) "s_nop 0\n\t" DCMT_U2(8) : "=&v"(pa), "=&v"(sa) : "v"(a));
#undef DCMT_U2
}

// The exact 5x5 median of dcmt_median.h on packed pairs: the two-input MERGE55 of median_shared_nets.h (the three-input forms
// need a packed med3, which does not exist), MID20 from median_pk3_nets.h, sort5 and the closing selection below.
#define DCMT_QCX(a, b)   { const unsigned lo_ = qmin(v[a], v[b]); v[b] = qmax(v[a], v[b]); v[a] = lo_; }
#define DCMT_QCMIN(a, b) { v[a] = qmin(v[a], v[b]); }
#define DCMT_QCMAX(a, b) { v[b] = qmax(v[a], v[b]); }
// sort5 with the three-input instructions: (a, b, c) = sort3(v0, v1, v2) as min3 / max3 and the middle one as the XOR of the five (a
// multiset identity, exact on bit patterns, ties included: v_bitop3_b32 0x96), (d, e) = sort2(v3, v4), then the merge of 3 + 2 by rank:
//     s0 = min(a, d)   s1 = min3(max(a, d), b, e)   s3 = max3(min(c, e), b, d)   s4 = max(c, e)   s2 = XOR of the five inputs and the other four
// 10 min / max + 5 XORs (3.9 cycles each, tools/pk3_probe.hip) instead of 18 min / max.  Checked exhaustively against sorted() on
// five values of five levels (tests/test_lane_schemes.py restates it).
__device__ __forceinline__ unsigned xor3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ void q_sort5(unsigned (&v)[5])
{
    const unsigned a = hmin3(v[0], v[1], v[2]), c = hmax3(v[0], v[1], v[2]), t = xor3(v[0], v[1], v[2]), b = xor3(t, a, c);
    const unsigned d = hmin2(v[3], v[4]), e = hmax2(v[3], v[4]);
    const unsigned s0 = hmin2(a, d), s4 = hmax2(c, e);
    const unsigned s1 = hmin3(hmax2(a, d), b, e), s3 = hmax3(hmin2(c, e), b, d);
    const unsigned s2 = xor3(xor3(t, v[3], v[4]), xor3(s0, s1, s3), s4);
    v[0] = s0; v[1] = s1; v[2] = s2; v[3] = s3; v[4] = s4;
}
__device__ __forceinline__ void q_merge55(const unsigned (&a)[5], const unsigned (&b)[5], unsigned (&P)[10])
{
    constexpr int out[10] = DCMT_MERGE55_OUT;
    unsigned v[10];
#pragma unroll
    for (int k = 0; k < 5; ++k) { v[k] = a[k]; v[5 + k] = b[k]; }
    DCMT_MERGE55_NET(DCMT_QCX, DCMT_QCMIN, DCMT_QCMAX)
#pragma unroll
    for (int k = 0; k < 10; ++k) P[k] = v[out[k]];
}
__device__ __forceinline__ void q_mid20(const unsigned (&pa)[10], const unsigned (&pb)[10], unsigned (&C)[6])
{
    // (median_pk3_nets.h: the two-input network with two of its exchanges folded into min3 / max3)
#define DCMT_IN_(k) ((k) < 10 ? pa[(k) < 10 ? (k) : 0] : pb[(k) >= 10 ? (k) - 10 : 0])
#define DCMT_OUT_(k) C[k]
    DCMT_MID20_PK3(DCMT_IN_, DCMT_OUT_)
#undef DCMT_IN_
#undef DCMT_OUT_
}
#undef DCMT_QCX
#undef DCMT_QCMIN
#undef DCMT_QCMAX
// 6th smallest of sorted C (6) u sorted a (5): min(C5, max(a0,C4), max(a1,C3), max(a2,C2), max(a3,C1), max(a4,C0)); the six-way minimum
// in three-input instructions: 5 + 3 instead of 5 + 5
__device__ __forceinline__ unsigned q_final6(const unsigned (&C)[6], const unsigned (&a)[5])
{
    const unsigned m0 = hmax2(a[0], C[4]), m1 = hmax2(a[1], C[3]), m2 = hmax2(a[2], C[2]), m3 = hmax2(a[3], C[1]), m4 = hmax2(a[4], C[0]);
    return hmin2(hmin3(C[5], m0, m1), hmin3(m2, m3, m4));
}
struct MedianColumnQ {       // MedianColumn (dcmt_median.h) on packed pairs
    unsigned SE[4][5], SO[5], P[2][10], C[6];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 5; ++k) SE[q][k] = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) SO[k] = 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) { P[0][k] = 0; P[1][k] = 0; }
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = 0;
    }
    template <int PP>
    __device__ __forceinline__ unsigned step(const unsigned (&s)[5])
    {
        unsigned m;
        if constexpr ((PP & 1) == 0) {
            constexpr int qs = (PP >> 1) & 3;
            q_merge55(SO, s, P[(PP >> 1) & 1]);
            q_mid20(P[((PP >> 1) + 1) & 1], P[(PP >> 1) & 1], C);
            m = q_final6(C, SE[(qs + 2) & 3]);
#pragma unroll
            for (int k = 0; k < 5; ++k) SE[qs][k] = s[k];
        } else {
            m = q_final6(C, s);
#pragma unroll
            for (int k = 0; k < 5; ++k) SO[k] = s[k];
        }
        return m;
    }
};

#ifndef DCMT_FPQ_WAVES
#define DCMT_FPQ_WAVES 0
#endif
#ifndef DCMT_FPQ_PFD
#define DCMT_FPQ_PFD 6
#endif
// BREG = true: the 30 halo columns of the 31-wide maximum ride in a second register (k_fp_p's layout): 120 output columns per wave,
//   144 VGPRs, 3 waves per SIMD.  BREG = false: no second register -- the strips overlap by the halo instead (X7 is exact for the
//   lanes 8..55 = 96 columns, 88 of them output): a fifth less work per wave, a quarter more waves, and few enough VGPRs for 4
//   waves per SIMD.
struct FpQ {
    template <bool BREG> static constexpr int halo() { return BREG ? FpP::H : 16 + FpP::H; }
    template <bool BREG> static constexpr int vw() { return 128 - 2 * halo<BREG>(); }
};
template <bool BLUR, bool X6U16, bool BREG = true, bool FILLED = false>
__global__ __launch_bounds__(256)
#if DCMT_FPQ_WAVES
__attribute__((amdgpu_waves_per_eu(DCMT_FPQ_WAVES, DCMT_FPQ_WAVES)))
#endif
void k_fp_q(const void* __restrict__ x6_, float* __restrict__ dst, int* __restrict__ counters,
            int rows_all, int cols, int strips, int batch, int xcd_map, float max_depth, float thr, const int* __restrict__ tb,
            int tbands)
{
    // per wave: centre values and A's 18-row maxima (packed pairs, one word per lane), B's 18-row maxima
    __shared__ unsigned s_delay[4][16 * (64 + 64 + (BREG ? 32 : 0))];   // 10 (8) KiB per wave, 40 (32) KiB per workgroup
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int f, strip;
    if (!wave_strip(blockIdx.x, wave, strips, batch, xcd_map, f, strip)) return;
    int* cnt = frame_counters(counters, f);
    const size_t fo = (size_t)f * rows_all * cols;
    constexpr int HALO = FpQ::halo<BREG>(), VW = FpQ::vw<BREG>();
    const int gx0 = strip * VW - HALO;
    const int gxe = gx0 + 2 * lane;
    // B, one column per lane, unpacked (k_fp_p's layout).  The dead lanes 16..47 (two whole DPP rows) shadow lane 0: same column,
    // same values, same delay-line word -- their stores write what lane 0 writes
    const int gxb = lane < 8 ? gx0 + 128 + 2 * lane : (lane < 16 ? gx0 + 128 + 2 * (lane - 8) + 1 :
                    (lane >= 56 ? gx0 - 16 + 2 * (lane - 56) : (lane >= 48 ? gx0 - 16 + 2 * (lane - 48) + 1 : gx0 + 128)));
    const int gxec = min(max(gxe, 0), cols - 2), gxoc = gxec + 1, gxbc = min(max(gxb, 0), cols - 1);
    int tie = 0, tio = 0, tib = 0, bie = rows_all - 1, bio = rows_all - 1, bib = rows_all - 1, V = 0;
    if (tb) {
        table_rows(tb, f, cols, tbands, rows_all, gxec, tie, bie);
        table_rows(tb, f, cols, tbands, rows_all, gxoc, tio, bio);
        if constexpr (BREG) table_rows(tb, f, cols, tbands, rows_all, gxbc, tib, bib);
        else tib = 0x7fffffff;
        V = __builtin_amdgcn_readfirstlane(max(wave_min_i(min(min(tie, tio), tib)) - 8, 0));
    }
    const int rows = rows_all - V;
    constexpr unsigned EB = X6U16 ? 2u : 4u;                         // bytes per X6 element
    FrameBuf sf;
    sf.init(reinterpret_cast<const float*>(static_cast<const char*>(x6_) + fo * EB), (size_t)rows_all * cols * EB / 4);
    const unsigned rowb = EB * (unsigned)cols;
    const unsigned sbe = EB * (unsigned)gxec + (unsigned)V * rowb, sbo = sbe + EB, sbb = EB * (unsigned)gxbc + (unsigned)V * rowb;
    const unsigned fle = EB * (unsigned)gxec + (unsigned)max(tie, V) * rowb, cee = EB * (unsigned)gxec + (unsigned)max(bie, V) * rowb;
    const unsigned flo = EB * (unsigned)gxoc + (unsigned)max(tio, V) * rowb, ceo = EB * (unsigned)gxoc + (unsigned)max(bio, V) * rowb;
    const unsigned flb = EB * (unsigned)gxbc + (unsigned)max(tib, V) * rowb, ceb = EB * (unsigned)gxbc + (unsigned)max(bib, V) * rowb;
    auto clamp3 = [](unsigned a, unsigned lo, unsigned hi) -> unsigned { unsigned r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi)); return r; };
    // one column's code at a (clamped) byte offset
    auto ld_code = [&](unsigned off) -> unsigned {
        if constexpr (X6U16) return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(sf.rs, off, 0, 0);
        else return Q16::code(sf.ld_at(off));
    };
    struct Raw { unsigned e, o, b; };
    auto ld_row = [&](int row) -> Raw {                              // row relative to V, already clamped to [0, rows)
        Raw r = {ld_code(clamp3(sbe + (unsigned)row * rowb, fle, cee)), ld_code(clamp3(sbo + (unsigned)row * rowb, flo, ceo)), 0u};
        if constexpr (BREG) r.b = ld_code(clamp3(sbb + (unsigned)row * rowb, flb, ceb));
        return r;
    };
    auto pack = [](unsigned e, unsigned o) -> unsigned { return e | (o << 16); };
    const bool outside = gxe < 0 || gxe >= cols;
    const bool own = !outside && 2 * lane >= HALO && 2 * lane < 128 - HALO;
    const unsigned long long own_mask = __ballot(own);
    const bool edge_strip = gx0 < 0 || gx0 + 127 >= cols;
    const int rep_l = min(max((0 - gx0) >> 1, 0), 63), rep_r = min(max((cols - 2 - gx0) >> 1, 0), 63);
    const int a_m8 = ((lane - 8) & 63) * 4, a_p8 = ((lane + 8) & 63) * 4;
    const bool b_lo = lane < 8, b_hi = lane >= 56;
    unsigned* sd = s_delay[wave];
    unsigned (*dl_c)[64] = reinterpret_cast<unsigned (*)[64]>(sd);
    unsigned (*dl_a)[64] = reinterpret_cast<unsigned (*)[64]>(sd + 16 * 64);
    unsigned (*dl_b)[32] = reinterpret_cast<unsigned (*)[32]>(sd + 16 * 128);
    const int lb = lane < 16 ? lane : (lane >= 48 ? lane - 32 : 0);

    PostPipeP<BLUR, HALO, FILLED, FILLED> pipe;                      // codes are grid values                                      // only its after_median() half is used
    pipe.init(dst + fo + (size_t)V * cols, rows, cols, gx0, lane, max_depth, thr);
    MedianColumnQ mc;
    mc.init();

    const bool warm = V > 0;
    unsigned xa0 = 0, xb0 = 0;                                       // cold start: 0 is the neutral element
    if (warm) { const Raw r = ld_row(0); xa0 = pack(r.e, r.o); xb0 = r.b; }
    unsigned PFA[16], PFB[16], W2A[16], W6A[16], W2B[16], W6B[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { PFA[q] = PFB[q] = 0; W2A[q] = W6A[q] = xa0; W2B[q] = W6B[q] = xb0; }
#pragma unroll
    for (int q = 0; q < 16; ++q) { dl_c[q][lane] = xa0; dl_a[q][lane] = xa0; if constexpr (BREG) dl_b[q][lb] = xb0; }
    constexpr int PFD = DCMT_FPQ_PFD;        // rows of load lookahead
#pragma unroll
    for (int q = 0; q < PFD; ++q) {
        const Raw r = ld_row(min(max(q + (warm ? 16 : 0) - 15, 0), rows - 1));
        PFA[q] = pack(r.e, r.o); PFB[q] = r.b;
    }
    unsigned vpa = xa0, vpb = xb0, x7_prev = xa0;
    int before = 0, after = 0;
    unsigned pend_v = xa0, pend_f1 = 0, pend_f2 = 0;
    unsigned long long pend_hme = 0, pend_hmo = 0;
    unsigned nxt_c = xa0, nxt_a = xa0, nxt_b = xb0;

    // the fill front end of step t: returns X7 (packed codes) of image row t - 31
    auto fill_step = [&](auto P_, int t) -> unsigned {
        constexpr int p = decltype(P_)::value;
        const unsigned xa = PFA[p], xb = PFB[p];
        {
            const Raw r = ld_row(min(max(t + PFD - 15, 0), rows - 1));
            PFA[(p + PFD) & 15] = pack(r.e, r.o); PFB[(p + PFD) & 15] = r.b;
        }
        const int o = t - 31;
        unsigned x7 = pend_v;
        if ((pend_hme | pend_hmo) != 0ull) {
            const unsigned d = qmax(pend_f1, pend_f2);
            const bool he = __builtin_amdgcn_inverse_ballot_w64(pend_hme), ho = __builtin_amdgcn_inverse_ballot_w64(pend_hmo);
            const unsigned m = (he ? 0xffffu : 0u) | (ho ? 0xffff0000u : 0u);
            x7 = (d & m) | (pend_v & ~m);
            if ((unsigned)o < (unsigned)rows) {
                before += __builtin_popcountll(pend_hme & own_mask) + __builtin_popcountll(pend_hmo & own_mask);
                after += __builtin_popcountll(__builtin_amdgcn_ballot_w64((x7 << 16) <= Q16::HOLE_MAX_HI) & own_mask) +
                         __builtin_popcountll(__builtin_amdgcn_ballot_w64(x7 <= Q16::HOLE_MAX_HI) & own_mask);
            }
        }
        if (edge_strip) {                                           // out-of-image columns replicate the edge column
            const unsigned l0 = (unsigned)__shfl((int)x7, rep_l, 64), r0 = (unsigned)__shfl((int)x7, rep_r, 64);
            if (gxe < 0) x7 = (l0 & 0xffffu) | (l0 << 16);
            if (gxe >= cols) x7 = (r0 >> 16) | (r0 & 0xffff0000u);
        }
        if (o >= rows) { asm volatile("" ::); x7 = x7_prev; }
        x7_prev = x7;
        // vertical 31-max: A packed (two-input instructions), B unpacked
        const unsigned w2a = hmax2(xa, vpa);
        vpa = xa;
        W2A[p] = w2a;
        const unsigned w6a = hmax3(w2a, W2A[(p + 14) & 15], W2A[(p + 12) & 15]);
        W6A[p] = w6a;
        const unsigned w18a = hmax3(w6a, W6A[(p + 10) & 15], W6A[(p + 4) & 15]);
        const unsigned v = nxt_c, w18a_old = nxt_a;
        nxt_c = dl_c[(p + 2) & 15][lane];
        nxt_a = dl_a[(p + 4) & 15][lane];
        dl_c[p][lane] = xa;
        dl_a[p][lane] = w18a;
        const unsigned w31a = hmax2(w18a, w18a_old);
        unsigned w31b = 0;
        if constexpr (BREG) {
            // (one code per lane in the low half, 0 above it: the packed f16 forms order these words too, and their three-input one is
            // formed reliably -- of two chained v_max_u32 the compiler fuses only one)
            const unsigned w2b = hmax2(xb, vpb);
            vpb = xb;
            W2B[p] = w2b;
            const unsigned w6b = hmax3(w2b, W2B[(p + 14) & 15], W2B[(p + 12) & 15]);
            W6B[p] = w6b;
            const unsigned w18b = hmax3(w6b, W6B[(p + 10) & 15], W6B[(p + 4) & 15]);
            const unsigned w18b_old = nxt_b;
            nxt_b = dl_b[(p + 4) & 15][lb];
            dl_b[p][lb] = w18b;
            w31b = hmax2(w18b, w18b_old);
        }
        const unsigned long long vme = __builtin_amdgcn_ballot_w64((v << 16) <= Q16::HOLE_MAX_HI), vmo = __builtin_amdgcn_ballot_w64(v <= Q16::HOLE_MAX_HI);
        if ((vme | vmo) != 0ull) {
            // horizontal 31-max: k_fp_p's scheme on the unpacked halves
            const unsigned e = w31a & 0xffffu, od = w31a >> 16;
            unsigned sx, so, px, pe;
            if constexpr (BREG) {
                const unsigned bo = u_row_ror8(w31b);
                unsigned pa, sa, pb, sb;
                u_row_scans4(umax2(e, od), umax2(w31b, bo), pa, sa, pb, sb);
                // exclusive scans = the inclusive ones shifted by a lane; which register a lane hands on is chosen BEFORE the shift, at
                // the source lane (lanes 0..6 feed the halo lanes 1..7, lane 7 feeds lane 8 of A; lanes 57..63 likewise): two shifts, not four
                px = u_row_shr1(lane < 7 ? pb : pa); sx = u_row_shl1(lane > 56 ? sb : sa);
                so = umax2(b_hi ? bo : od, sx);
                pe = umax2(b_lo ? w31b : e, px);
            } else {
                // (the fetches of lanes 0..7 and 56..63 wrap around the wave: their X7 is not exact, and nothing reads it)
                unsigned pa, sa;
                u_row_scans2(umax2(e, od), pa, sa);
                px = u_row_shr1(pa); sx = u_row_shl1(sa);
                so = umax2(od, sx); pe = umax2(e, px);
            }
            // out_E(l) = max(SO(l-8), PX(l+8)), out_O(l) = max(SX(l-8), PE(l+8)): the two values a lane fetches from lane l-8 ride in one
            // word (SO low, SX high), so do the two from lane l+8 (PX low, PE high) -- two ds_bpermutes, not four, and their packed maximum
            // is the pair (out_E, out_O) as it is needed
            pend_f1 = (unsigned)__builtin_amdgcn_ds_bpermute(a_m8, (int)(so | (sx << 16)));
            pend_f2 = (unsigned)__builtin_amdgcn_ds_bpermute(a_p8, (int)(px | (pe << 16)));
        }
        pend_v = v;
        pend_hme = vme; pend_hmo = vmo;
        return x7;
    };
    // post step u: the median of image row u - 4 on packed pairs, then PostPipeP's f32 tail
    auto post_step = [&](auto PP_, unsigned x, int u) {
        constexpr int PP = decltype(PP_)::value;
        const unsigned rl = u_left(x), rr = u_right(x);              // columns 2l-2, 2l-1 | 2l+2, 2l+3
        unsigned s[5] = {rl, __builtin_amdgcn_alignbit(x, rl, 16), x, __builtin_amdgcn_alignbit(rr, x, 16), rr};
        q_sort5(s);
        const unsigned m = mc.template step<PP>(s);
        if constexpr (BLUR && FILLED) pipe.template after_median_codes<PP>(m, u);       // codes all the way: exact (PostPipeP)
        else pipe.template after_median<PP>(Q16::value(m & 0xffffu), Q16::value(m >> 16), u);
    };

    // steps 0..31 (16..31 after a warm start): fill only.  The last of them returns X7 row 0, which the post pipeline takes three times
    // (its replicated rows -2 and -1, and row 0: post steps 0, 1, 2)
    for (int t0 = warm ? 16 : 0; t0 < 32; t0 += 16) {
        static_for<0, 16>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            const unsigned x7 = fill_step(P_, t0 + p);
            if constexpr (p == 15) {
                if (t0 == 16) {
                    post_step(std::integral_constant<int, 0>{}, x7, 0);
                    post_step(std::integral_constant<int, 1>{}, x7, 1);
                    post_step(std::integral_constant<int, 2>{}, x7, 2);
                }
            }
        });
    }
    // steps 32..rows+34: post step u = t - 29 takes X7 row u - 2 = t - 31, the row this step's fill front end returns
    const int nsteps = rows + 35;
    int t0 = 32;
    auto main_step = [&](auto P_) {
        constexpr int p = decltype(P_)::value;
        const int t = t0 + p, u = t - 29;
        const unsigned x7 = fill_step(P_, t);
        post_step(std::integral_constant<int, ((p + 3) & 7)>{}, x7, u);
        if constexpr (p == 3) {
            // u == 6: output row 0 of the shifted frame (image row V) has just been stored; the V rows above it are equal
            if (t0 == 32 && V > 0) {
                FrameBuf top;
                top.init(dst + fo, (size_t)V * cols);
                const unsigned tbo = pipe.outlane ? pipe.ob : kDropOffset;
                for (int r = 0; r < V; ++r) st2(top, tbo, r, cols, pipe.last_out);
            }
        }
    };
    // (the sixteen steps of a block in four quarters with a way out behind each: the last block of a wave is 7.5 steps too long on
    // average otherwise -- 3 % of its row steps)
    for (; t0 < nsteps; t0 += 16) {
        static_for<0, 4>(main_step);
        if (t0 + 4 >= nsteps) break;
        static_for<4, 8>(main_step);
        if (t0 + 8 >= nsteps) break;
        static_for<8, 12>(main_step);
        if (t0 + 12 >= nsteps) break;
        static_for<12, 16>(main_step);
    }
    if (lane == 0) {
        if (before) atomicAdd(&cnt[0], before);
        if (after) atomicAdd(&cnt[1], after);
    }
}

}  // namespace dcmt
