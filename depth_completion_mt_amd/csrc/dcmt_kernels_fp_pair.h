// dcmt_kernels_fp_pair.h -- k_fp_p: H7, H9..H11 (k_fp_s of dcmt_kernels_fused.h) with TWO adjacent columns per lane.
//
// k_fp_s is bound by the issue of its max / min / med3 / DPP instructions ("What binds"): 62 of them per row
// step for 56 output columns.  A wave64 of this kernel owns 128 columns (lane l = columns c0 + 2l "E" and c0 + 2l + 1 "O",
// 120 of them output) and needs fewer of those instructions per column:
//   * the median's horizontal sort: the 5-windows of a lane's two columns share four values {O[l-1], E, O, E[l+1]}.  Those are
//     sorted once (9 three-input instructions, the neighbour shifts folded into the first exchanges) and each column inserts
//     its fifth value (min, med3, med3, med3, max): 2 + 9 + 2 x 5 = 21 per column pair instead of 2 x (4 + 12);
//   * the Gaussian's four neighbour shifts serve both columns (4 per pair instead of 8);
//   * the 15 + 15 halo columns of the 31-wide maximum (register B, one column per lane as in k_fp_s) are paid once per 128
//     columns instead of once per 64.
//
// Horizontal 31-wide maximum in the pair layout (tests/test_lane_schemes.py restates this index arithmetic in numpy and
// checks it lane by lane against the definition).  m = max(E, O) per lane; PX / SX = EXCLUSIVE prefix / suffix maxima of m inside each
// 16-lane DPP row (the inclusive row scans of k_fp_s, then one row_shr:1 / row_shl:1 that leaves -FLT_MAX in the lane
// without a source).  The window of column 2l is O[l-8], lanes l-7 .. l+7; that of column 2l+1 is lanes l-7 .. l+7, E[l+8]:
//     out_E(l) = max( SO(l-8), PX(l+8) )        SO = max(O, SX)   (suffix starting at a lane's odd column)
//     out_O(l) = max( SX(l-8), PE(l+8) )        PE = max(E, PX)   (prefix ending at a lane's even column)
// -- lanes l-8 and l+8 lie in neighbouring DPP rows, each term covers its row's part of the window, and where the window
// lies inside ONE row (l = 8 mod 16 for E, 7 mod 16 for O) the other term is the -FLT_MAX of an exclusive scan's first lane.
// Four ds_bpermutes, two addresses.  The halo: virtual lanes 64..71 (the 16 columns right of the strip) and -8..-1 (left)
// wrap onto physical lanes 0..7 and 56..63 of B's scans; B holds one column per lane -- even columns of the right halo
// in lanes 0..7, odd ones in 8..15, even columns of the left halo in lanes 56..63, odd ones in 48..55 -- so one row_ror:8 puts
// a virtual lane's odd column beside its even one, and the values handed to the bpermutes are selected per SOURCE lane.
//
// Everything else is k_fp_s: vertical 31-max in four three-input instructions per register, the three 15 / 13-step delays in
// wave-private LDS, LDS round trips skewed by one step, rows without holes skip the horizontal part, the top extension zone
// is not streamed (table mode), the post pipeline runs 32 steps behind in the same step.  Loads stay one dword per column:
// a lane's two columns have different (first, last) valid rows, so their row clamps differ.
// 2 waves per workgroup (37 KB of LDS), ~200 VGPRs: 2 waves per SIMD, each with two independent columns' worth of work.
// Requires an even number of columns and 8-byte aligned frames; other shapes run k_fp_s.  Bit-identical to k_fp_s.
#pragma once

#include "dcmt_kernels_pair.h"

namespace dcmt {

struct FpP {
    static constexpr int H = 4;                  // columns lost per side: 2 (median) + 2 (Gaussian)
    static constexpr int VW = 128 - 2 * H;       // output columns per wave
    static constexpr int LAG = 32;               // post step u = t - LAG
    static constexpr int WPB = 2;                // waves per workgroup
    static constexpr int BW = 33;                // words per slot of B's delay line: 32 live lanes + one word the dead lanes share
};

// lanes without a source keep -FLT_MAX
__device__ __forceinline__ float row_shr1_neg(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -FLT_MAX), __builtin_bit_cast(int, v), 0x111 /*row_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_shl1_neg(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -FLT_MAX), __builtin_bit_cast(int, v), 0x101 /*row_shl:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_ror8(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /*row_ror:8*/, 0xf, 0xf, false));
}
// inclusive prefix and suffix maxima inside each 16-lane row of a and of b, the four scans' steps interleaved (every DPP read
// is three instructions behind the write of its source: no hazard padding inside the block)
__device__ __forceinline__ void row_scans4(float a, float b, float& pa, float& sa, float& pb, float& sb)
{
    pa = a; sa = a; pb = b; sb = b;
#define DCMT_S4(N) "v_max_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                   "v_max_f32_dpp %1, %1, %1 row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                   "v_max_f32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                   "v_max_f32_dpp %3, %3, %3 row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t" DCMT_S4(1) DCMT_S4(2) DCMT_S4(4) DCMT_S4(8) : "+v"(pa), "+v"(sa), "+v"(pb), "+v"(sb));
#undef DCMT_S4
}

// sorted 5-windows of a lane's two columns: se = sort{E[l-1], O[l-1], E, O, E[l+1]}, so = sort{O[l-1], E, O, E[l+1], O[l+1]}
__device__ __forceinline__ void sort5_pair(F2 x, float (&se)[5], float (&so)[5])
{
    const float a = from_left(x.o), d = from_right(x.e);                 // O[l-1], E[l+1]: shared
    const float el = from_left(x.e), orr = from_right(x.o);              // E[l-1] (E's window only), O[l+1] (O's only)
    const float lo1 = fmin2(a, x.e), hi1 = fmax2(a, x.e), lo2 = fmin2(d, x.o), hi2 = fmax2(d, x.o);
    const float y = fmin2(hi1, hi2);
    const float s0 = fmin2(lo1, lo2), s1 = __builtin_amdgcn_fmed3f(lo1, lo2, y), s2 = fmax3(lo1, lo2, y), s3 = fmax2(hi1, hi2);
    se[0] = fmin2(el, s0); se[1] = __builtin_amdgcn_fmed3f(el, s0, s1); se[2] = __builtin_amdgcn_fmed3f(el, s1, s2);
    se[3] = __builtin_amdgcn_fmed3f(el, s2, s3); se[4] = fmax2(el, s3);
    so[0] = fmin2(orr, s0); so[1] = __builtin_amdgcn_fmed3f(orr, s0, s1); so[2] = __builtin_amdgcn_fmed3f(orr, s1, s2);
    so[3] = __builtin_amdgcn_fmed3f(orr, s2, s3); so[4] = fmax2(orr, s3);
}

// PostPipe (dcmt_kernels_fused.h) for two columns per lane, MODE 11 only.  Step u takes X7 row clamp(u - 2), finishes the
// median of image row u - 4 and the output of image row u - 6.
// FILLED: see PostPipe.  GRID (k_fp_q, with FILLED): every median is a multiple of 1/256 and >= thr = 0.1, i.e. >= 26/256 = 0.1015625;
// the blurred value is a convex combination of such medians evaluated with a dozen roundings of 2^-24 each, so it stays above
// 0.10156 > thr and the final invert (LO :191-202) is `max_depth - value` without its compare and select.
template <bool BLUR, int HALO = FpP::H, bool FILLED = false, bool GRID = false>
struct PostPipeP {
    MedianColumn mce, mco;
    F2 G1[8], MR[8];
    F2 last_out;
    FrameBuf of;
    unsigned ob;             // byte offset of this lane's (clamped) column pair
    int rows, cols, gx, rle, rlo;
    bool outlane, edge_strip, outside;
    float max_depth, thr;

    __device__ __forceinline__ void init(float* out_frame, int rows_, int cols_, int gx0, int lane, float max_depth_, float thr_)
    {
        of.init(out_frame, (size_t)rows_ * cols_); rows = rows_; cols = cols_; max_depth = max_depth_; thr = thr_;
        gx = gx0 + 2 * lane;
        ob = 4u * (unsigned)min(max(gx, 0), cols - 2);
        outside = gx < 0 || gx >= cols;                            // cols and gx are even: both columns inside or both outside
        outlane = !outside && 2 * lane >= HALO && 2 * lane < 128 - HALO;
        // reflect-101 sources of the Gaussian's out-of-image columns: parity is preserved (cols is even), so E comes from an E slot, O from an O slot
        rle = (reflect101(gx, cols) - gx0) >> 1;
        rlo = (reflect101(gx + 1, cols) - 1 - gx0) >> 1;
        edge_strip = gx0 < 0 || gx0 + 127 >= cols;
        mce.init(); mco.init();
        last_out = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) { G1[q] = {0.f, 0.f}; MR[q] = {0.f, 0.f}; }
    }

    template <int PP>
    __device__ __forceinline__ void step(F2 x, int u)
    {
        // ---- H9 (LO :170)
        float se[5], so[5];
        sort5_pair(x, se, so);
        after_median<PP>(mce.template step<PP>(se), mco.template step<PP>(so), u);
    }

    // everything behind the median of image row u - 4 (me, mo = its two columns): H10, H11, the store of output row u - 6
    template <int PP>
    __device__ __forceinline__ void after_median(float me, float mo, int u)
    {
        MR[(PP + 4) & 7] = {me, mo};
        // ---- H10 (LO :179): horizontal [1 4 6 4 1]/16 with reflect-101 columns
        if constexpr (BLUR) {
            F2 mf = {me, mo};
            if (edge_strip) {
                const float re = __shfl(me, rle, 64), ro = __shfl(mo, rlo, 64);
                if (outside) mf = {re, ro};
            }
            const float el = from_left(mf.e), ol = from_left(mf.o), er = from_right(mf.e), orr = from_right(mf.o);
            // column 2l: neighbours O[l-1], O | E[l-1], E[l+1];  column 2l+1: E, E[l+1] | O[l-1], O[l+1]
            G1[(PP + 4) & 7] = {gauss_taps(mf.e, __fadd_rn(ol, mf.o), __fadd_rn(el, er)), gauss_taps(mf.o, __fadd_rn(mf.e, er), __fadd_rn(ol, orr))};
        }
        // ---- vertical pass + select + invert for output row o = u - 6
        const int o = u - 6;
        if ((unsigned)o < (unsigned)rows) {
            const F2 mo_ = MR[(PP + 2) & 7];
            auto finish = [&](F2 u1, F2 u2, F2 d1, F2 d2) {
                F2 val = mo_;
                if constexpr (BLUR) {
                    const F2 g0 = G1[(PP + 2) & 7];
                    const float ae = gauss_taps(g0.e, __fadd_rn(u1.e, d1.e), __fadd_rn(u2.e, d2.e));
                    const float ao = gauss_taps(g0.o, __fadd_rn(u1.o, d1.o), __fadd_rn(u2.o, d2.o));
                    if (FILLED || mo_.e >= thr) val.e = ae;              // LO :184
                    if (FILLED || mo_.o >= thr) val.o = ao;
                }
                if constexpr (BLUR && FILLED && GRID) val = {__fsub_rn(max_depth, val.e), __fsub_rn(max_depth, val.o)};
                else val = {invert_valid(val.e, max_depth, thr), invert_valid(val.o, max_depth, thr)};   // LO :191-202
                st2(of, outlane ? ob : kDropOffset, o, cols, val);
                last_out = val;
            };
            const F2 g_p2 = G1[(PP + 4) & 7], g_p1 = G1[(PP + 3) & 7], g_0 = G1[(PP + 2) & 7], g_m1 = G1[(PP + 1) & 7], g_m2 = G1[PP];
            if (BLUR && (o < 2 || o + 2 >= rows)) {                      // reflect-101 rows (rows >= 8 guaranteed)
                finish(o >= 1 ? g_m1 : g_p1,
                       o >= 2 ? g_m2 : (o == 1 ? g_0 : g_p2),
                       o + 1 < rows ? g_p1 : g_m1,
                       o + 2 < rows ? g_p2 : (o + 2 == rows ? g_0 : g_m2));
            } else {
                finish(g_m1, g_m2, g_p1, g_p2);
            }
        }
    }

    // The same for k_fp_q's packed medians m = (code_E, code_O) where BLUR && FILLED && GRID: IN EXACT ARITHMETIC.  A code is an integer
    // below 2^15, value = (code - 6143) / 256.  The horizontal pass of the reference order  c*k0 + s1*k1 + s2*k2  on such values only
    // ever forms multiples of 2^-12 below 128 (19 bits), the vertical pass multiples of 2^-16 below 128 (23 bits), and the final
    // 100 - value is a multiple of 2^-16 below 128 too: no operation of the oracle's sequence rounds, so any other sequence
    // without a rounding gives the same bits.  This one works on the codes as integer-valued floats with the weights 1 4 6 4 1
    // (horizontal sums <= 16 * 31743 < 2^19, vertical <= 256 * 31743 < 2^23) and scales once at the end:
    //     out = 100 - (N / 65536 - 6143 / 256) = fma(N, -2^-16, 123.99609375).
    // Horizontal, two columns per lane, four lane shifts (each folded into an add):  S1 = O[l-1] + O,  S2 = E + E[l+1],
    //     G_E = 4 (E + S1) + (S2[l-1] + S2) = 6 E + 4 (O[l-1] + O) + E[l-1] + E[l+1],   G_O = 4 (O + S2) + (S1 + S1[l+1]):
    // 2 conversions + 8 + 2 * 4 + 2 instructions per row step instead of 4 + 14 + 10 + 2.
    template <int PP>
    __device__ __forceinline__ void after_median_codes(unsigned m, int u)
    {
        static_assert(BLUR && FILLED && GRID, "exact only on grid values; the select of LO :184 is not in here");
        float ce = (float)(m & 0xffffu), co = (float)(m >> 16);
        if (edge_strip) {
            const float re = __shfl(ce, rle, 64), ro = __shfl(co, rlo, 64);
            if (outside) { ce = re; co = ro; }
        }
        const float s1 = __fadd_rn(from_left(co), co), s2 = __fadd_rn(ce, from_right(ce));
        const float ue = __fadd_rn(from_left(s2), s2), wo = __fadd_rn(from_right(s1), s1);
        G1[(PP + 4) & 7] = {__builtin_fmaf(__fadd_rn(ce, s1), 4.0f, ue), __builtin_fmaf(__fadd_rn(co, s2), 4.0f, wo)};
        const int o = u - 6;
        if ((unsigned)o < (unsigned)rows) {
            const float top = __fadd_rn(max_depth, (float)Q16::OFFSET * 0.00390625f);     // 100 + 6143 / 256: exact
            auto finish = [&](F2 u1, F2 u2, F2 d1, F2 d2) {
                const F2 g0 = G1[(PP + 2) & 7];
                const float ne = __builtin_fmaf(g0.e, 6.0f, __builtin_fmaf(__fadd_rn(u1.e, d1.e), 4.0f, __fadd_rn(u2.e, d2.e)));
                const float no = __builtin_fmaf(g0.o, 6.0f, __builtin_fmaf(__fadd_rn(u1.o, d1.o), 4.0f, __fadd_rn(u2.o, d2.o)));
                const F2 val = {__builtin_fmaf(ne, -0x1p-16f, top), __builtin_fmaf(no, -0x1p-16f, top)};
                st2(of, outlane ? ob : kDropOffset, o, cols, val);
                last_out = val;
            };
            const F2 g_p2 = G1[(PP + 4) & 7], g_p1 = G1[(PP + 3) & 7], g_0 = G1[(PP + 2) & 7], g_m1 = G1[(PP + 1) & 7], g_m2 = G1[PP];
            if (o < 2 || o + 2 >= rows) {                                // reflect-101 rows, as above
                finish(o >= 1 ? g_m1 : g_p1,
                       o >= 2 ? g_m2 : (o == 1 ? g_0 : g_p2),
                       o + 1 < rows ? g_p1 : g_m1,
                       o + 2 < rows ? g_p2 : (o + 2 == rows ? g_0 : g_m2));
            } else {
                finish(g_m1, g_m2, g_p1, g_p2);
            }
        }
    }
};

// wave -> (frame, strip) for WPB waves per workgroup (wave_strip of dcmt_kernels_fused.h has 4)
template <int WPB>
__device__ __forceinline__ bool wave_strip_n(int b, int wave, int strips, int batch, int xcd_map, int& f, int& strip)
{
    if (xcd_map) {
        const int g = (b >> 3) * WPB + wave;
        if (g >= (batch >> 3) * strips) return false;
        f = (g / strips) * 8 + (b & 7);
        strip = g % strips;
    } else {
        const int g = b * WPB + wave;
        if (g >= batch * strips) return false;
        f = g / strips;
        strip = g % strips;
    }
    return true;
}

template <bool BLUR>
__global__ __launch_bounds__(64 * FpP::WPB)
void k_fp_p(const float* __restrict__ x6, float* __restrict__ dst, int* __restrict__ counters,
            int rows_all, int cols, int strips, int batch, int xcd_map, float max_depth, float thr, const int* __restrict__ tb,
            int tbands)
{
    // per wave: centre values and A's 18-row maxima (two columns per lane, 8 bytes), B's 18-row maxima
    __shared__ __attribute__((aligned(16))) float s_delay[FpP::WPB][16 * (128 + 128 + FpP::BW)];   // 18496 bytes per wave: the 8-byte slots stay aligned
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int f, strip;
    if (!wave_strip_n<FpP::WPB>(blockIdx.x, wave, strips, batch, xcd_map, f, strip)) return;
    int* cnt = frame_counters(counters, f);
    const size_t fo = (size_t)f * rows_all * cols;
    const int gx0 = strip * FpP::VW - FpP::H;
    const int gxe = gx0 + 2 * lane;                                     // column of E; O = gxe + 1
    // B, one column per lane: right halo (columns gx0+128 ..) even columns in lanes 0..7, odd ones in lanes 8..15; left halo
    // (columns gx0-16 .. gx0-1) even columns in lanes 56..63, odd ones in lanes 48..55; the other lanes are dead (their own E column)
    const int gxb = lane < 8 ? gx0 + 128 + 2 * lane : (lane < 16 ? gx0 + 128 + 2 * (lane - 8) + 1 :
                    (lane >= 56 ? gx0 - 16 + 2 * (lane - 56) : (lane >= 48 ? gx0 - 16 + 2 * (lane - 48) + 1 : gxe)));
    const int gxec = min(max(gxe, 0), cols - 2), gxoc = gxec + 1, gxbc = min(max(gxb, 0), cols - 1);   // clamped: replicate == constant border for a max filter
    // ---- the extension zones (k_fp_s has the argument): row indices clamped per column into [ti, bi]; the wave treats
    // row V = (smallest ti of the columns it reads) - 8 as the top of its frame
    int tie = 0, tio = 0, tib = 0, bie = rows_all - 1, bio = rows_all - 1, bib = rows_all - 1, V = 0;
    if (tb) {
        table_rows(tb, f, cols, tbands, rows_all, gxec, tie, bie);
        table_rows(tb, f, cols, tbands, rows_all, gxoc, tio, bio);
        table_rows(tb, f, cols, tbands, rows_all, gxbc, tib, bib);
        V = __builtin_amdgcn_readfirstlane(max(wave_min_i(min(min(tie, tio), tib)) - 8, 0));
    }
    const int rows = rows_all - V;                                   // rows of the frame as this wave sees it (>= 9)
    FrameBuf sf;
    sf.init(x6 + fo, (size_t)rows_all * cols);
    const unsigned rowb = 4u * (unsigned)cols;
    const unsigned sbe = 4u * (unsigned)gxec + (unsigned)V * rowb, sbo = sbe + 4u, sbb = 4u * (unsigned)gxbc + (unsigned)V * rowb;
    const unsigned fle = 4u * (unsigned)gxec + (unsigned)max(tie, V) * rowb, cee = 4u * (unsigned)gxec + (unsigned)max(bie, V) * rowb;
    const unsigned flo = 4u * (unsigned)gxoc + (unsigned)max(tio, V) * rowb, ceo = 4u * (unsigned)gxoc + (unsigned)max(bio, V) * rowb;
    const unsigned flb = 4u * (unsigned)gxbc + (unsigned)max(tib, V) * rowb, ceb = 4u * (unsigned)gxbc + (unsigned)max(bib, V) * rowb;
    auto clamp3 = [](unsigned a, unsigned lo, unsigned hi) -> unsigned { unsigned r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi)); return r; };
    auto ld_e = [&](int row) -> float { return sf.ld_at(clamp3(sbe + (unsigned)row * rowb, fle, cee)); };   // row relative to V, already clamped to [0, rows)
    auto ld_o = [&](int row) -> float { return sf.ld_at(clamp3(sbo + (unsigned)row * rowb, flo, ceo)); };
    auto ld_b = [&](int row) -> float { return sf.ld_at(clamp3(sbb + (unsigned)row * rowb, flb, ceb)); };
    const bool outside = gxe < 0 || gxe >= cols;
    const bool own = !outside && 2 * lane >= FpP::H && 2 * lane < 128 - FpP::H;     // column pairs this wave accounts for
    const unsigned long long own_mask = __ballot(own);
    const bool edge_strip = gx0 < 0 || gx0 + 127 >= cols;
    // BORDER_REPLICATE columns for the median: left of the image = column 0 (an E slot), right of it = column cols-1 (an O slot)
    const int rep_l = min(max((0 - gx0) >> 1, 0), 63), rep_r = min(max((cols - 2 - gx0) >> 1, 0), 63);
    const int a_m8 = ((lane - 8) & 63) * 4, a_p8 = ((lane + 8) & 63) * 4;            // ds_bpermute byte addresses
    const bool b_lo = lane < 8, b_hi = lane >= 56;                                   // this lane supplies halo values to the (l+8) / (l-8) fetches
    float* sd = s_delay[wave];
    F2 (*dl_c)[64] = reinterpret_cast<F2 (*)[64]>(sd);
    F2 (*dl_a)[64] = reinterpret_cast<F2 (*)[64]>(sd + 16 * 128);
    float (*dl_b)[FpP::BW] = reinterpret_cast<float (*)[FpP::BW]>(sd + 16 * 256);
    const int lb = lane < 16 ? lane : (lane >= 48 ? lane - 32 : 32);

    PostPipeP<BLUR> pipe;
    pipe.init(dst + fo + (size_t)V * cols, rows, cols, gx0, lane, max_depth, thr);

    constexpr float NEG = -FLT_MAX;
    // warm start (k_fp_s): with V > 0 the first 16 steps would feed row V sixteen times; they are replaced by the state they leave
    const bool warm = V > 0;
    const float xe0 = warm ? ld_e(0) : NEG, xo0 = warm ? ld_o(0) : NEG, xb0 = warm ? ld_b(0) : NEG;
    float PFE[16], PFO[16], PFB[16], W2E[16], W2O[16], W6E[16], W6O[16], W2B[16], W6B[16];
    F2 DL[8];
#pragma unroll
    for (int q = 0; q < 16; ++q) { PFE[q] = PFO[q] = PFB[q] = 0.f; W2E[q] = W6E[q] = xe0; W2O[q] = W6O[q] = xo0; W2B[q] = W6B[q] = xb0; }
#pragma unroll
    for (int q = 0; q < 8; ++q) DL[q] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q) { dl_c[q][lane] = {xe0, xo0}; dl_a[q][lane] = {xe0, xo0}; dl_b[q][lb] = xb0; }
#ifndef DCMT_FPP_PFD
#define DCMT_FPP_PFD 6
#endif
    constexpr int PFD = DCMT_FPP_PFD;        // rows of load lookahead
#pragma unroll
    for (int q = 0; q < PFD; ++q) {
        const int row = min(max(q + (warm ? 16 : 0) - 15, 0), rows - 1);
        PFE[q] = ld_e(row); PFO[q] = ld_o(row); PFB[q] = ld_b(row);
    }
    float vpe = xe0, vpo = xo0, vpb = xb0;
    F2 x7_prev = warm ? F2{xe0, xo0} : F2{0.f, 0.f};
    int before = 0, after = 0;
    // software skew (k_fp_s): what step t issues to the LDS crossbar / delay lines is consumed in step t + 1
    F2 pend_v = {xe0, xo0};
    float pend_f1 = NEG, pend_f2 = NEG, pend_f3 = NEG, pend_f4 = NEG;
    unsigned long long pend_hme = 0, pend_hmo = 0;                       // hole masks of stream row t - 1 (wave-uniform: SGPRs)
    F2 nxt_c = {xe0, xo0}, nxt_a = {xe0, xo0};
    float nxt_b = xb0;

    // the fill front end of step t: returns X7 of image row t - 31 for both columns of every lane
    auto fill_step = [&](auto P_, int t) -> F2 {
        constexpr int p = decltype(P_)::value;
        const float xe = PFE[p], xo = PFO[p], xb = PFB[p];
        {
            const int row = min(max(t + PFD - 15, 0), rows - 1);
            PFE[(p + PFD) & 15] = ld_e(row); PFO[(p + PFD) & 15] = ld_o(row); PFB[(p + PFD) & 15] = ld_b(row);
        }
        // finish row t - 1 - 30 from what step t - 1 left pending
        const int o = t - 31;
        F2 x7 = pend_v;
        if ((pend_hme | pend_hmo) != 0ull) {                        // a row without holes passes through untouched
            const float de = fmax2(pend_f1, pend_f2), dd = fmax2(pend_f3, pend_f4);
            const bool he = __builtin_amdgcn_inverse_ballot_w64(pend_hme), ho = __builtin_amdgcn_inverse_ballot_w64(pend_hmo);   // x < thr, LO :140
            x7 = {he ? de : pend_v.e, ho ? dd : pend_v.o};
            if ((unsigned)o < (unsigned)rows) {                     // hole counts on the scalar unit
                before += __builtin_popcountll(pend_hme & own_mask) + __builtin_popcountll(pend_hmo & own_mask);
                after += __builtin_popcountll(__builtin_amdgcn_ballot_w64(x7.e < thr) & own_mask) +
                         __builtin_popcountll(__builtin_amdgcn_ballot_w64(x7.o < thr) & own_mask);
            }
        }
        if (edge_strip) {                                           // out-of-image columns replicate the edge column
            const float l0 = __shfl(x7.e, rep_l, 64), r0 = __shfl(x7.o, rep_r, 64);
            if (gxe < 0) x7 = {l0, l0};
            if (gxe >= cols) x7 = {r0, r0};
        }
        if (o >= rows) { asm volatile("" ::); x7 = x7_prev; }       // rows below the image replicate the last row (scalar branch, last steps only)
        x7_prev = x7;
        // vertical 31-max of the three registers, four instructions each (k_fp_s)
        const float w2e = fmax2(xe, vpe), w2o = fmax2(xo, vpo), w2b = fmax2(xb, vpb);
        vpe = xe; vpo = xo; vpb = xb;
        W2E[p] = w2e; W2O[p] = w2o; W2B[p] = w2b;
        const float w6e = fmax3(w2e, W2E[(p + 14) & 15], W2E[(p + 12) & 15]), w6o = fmax3(w2o, W2O[(p + 14) & 15], W2O[(p + 12) & 15]);
        const float w6b = fmax3(w2b, W2B[(p + 14) & 15], W2B[(p + 12) & 15]);
        W6E[p] = w6e; W6O[p] = w6o; W6B[p] = w6b;
        const float w18e = fmax3(w6e, W6E[(p + 10) & 15], W6E[(p + 4) & 15]), w18o = fmax3(w6o, W6O[(p + 10) & 15], W6O[(p + 4) & 15]);
        const float w18b = fmax3(w6b, W6B[(p + 10) & 15], W6B[(p + 4) & 15]);
        // LDS delay lines (slots as in k_fp_s: centre values delayed by 15 steps, 18-row maxima by 13)
        const F2 v = nxt_c, w18a_old = nxt_a;
        const float w18b_old = nxt_b;
        nxt_c = dl_c[(p + 2) & 15][lane];
        nxt_a = dl_a[(p + 4) & 15][lane];
        nxt_b = dl_b[(p + 4) & 15][lb];
        dl_c[p][lane] = {xe, xo};
        dl_a[p][lane] = {w18e, w18o};
        dl_b[p][lb] = w18b;
        const float w31e = fmax2(w18e, w18a_old.e), w31o = fmax2(w18o, w18a_old.o), w31b = fmax2(w18b, w18b_old);
        // horizontal 31-max (header comment); skipped when none of the wave's 128 columns is a hole in this row
        const unsigned long long vme = __builtin_amdgcn_ballot_w64(v.e < thr), vmo = __builtin_amdgcn_ballot_w64(v.o < thr);
        if ((vme | vmo) != 0ull) {
            const float bo = row_ror8(w31b);                        // the odd column beside the even one in lanes 0..7 and 56..63
            float pa, sa, pb, sb;
            row_scans4(fmax2(w31e, w31o), fmax2(w31b, bo), pa, sa, pb, sb);
            const float pxa = row_shr1_neg(pa), sxa = row_shl1_neg(sa), pxb = row_shr1_neg(pb), sxb = row_shl1_neg(sb);
            const float sx = b_hi ? sxb : sxa, so = fmax2(b_hi ? bo : w31o, sx);
            const float px = b_lo ? pxb : pxa, pe = fmax2(b_lo ? w31b : w31e, px);
            pend_f1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a_m8, __builtin_bit_cast(int, so)));
            pend_f2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a_p8, __builtin_bit_cast(int, px)));
            pend_f3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a_m8, __builtin_bit_cast(int, sx)));
            pend_f4 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a_p8, __builtin_bit_cast(int, pe)));
        }
        pend_v = v;
        pend_hme = vme; pend_hmo = vmo;
        return x7;
    };

    // steps 0..31 (16..31 after a warm start): fill only (X7 row 0 appears at t = 31)
    for (int t0 = warm ? 16 : 0; t0 < FpP::LAG; t0 += 16) {
        static_for<0, 16>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            DL[p & 7] = fill_step(P_, t0 + p);
        });
    }
    DL[5] = DL[6] = DL[7];                       // rows -2, -1 replicate row 0
    const int nsteps = rows + 38;
    for (int t0 = FpP::LAG; t0 < nsteps; t0 += 16) {
        static_for<0, 16>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            const int t = t0 + p, u = t - FpP::LAG;
            DL[p & 7] = fill_step(P_, t);
            pipe.template step<(p & 7)>(DL[(p + 5) & 7], u);
            if constexpr (p == 6) {
                // u == 6: output row 0 of the shifted frame (image row V) has just been stored; the V rows above it are equal
                if (t0 == FpP::LAG && V > 0) {
                    FrameBuf top;
                    top.init(dst + fo, (size_t)V * cols);
                    const unsigned tbo = pipe.outlane ? pipe.ob : kDropOffset;
                    for (int r = 0; r < V; ++r) st2(top, tbo, r, cols, pipe.last_out);
                }
            }
        });
    }
    if (lane == 0) {
        if (before) atomicAdd(&cnt[0], before);
        if (after) atomicAdd(&cnt[1], after);
    }
}

}  // namespace dcmt
