// dcmt_kernels_fp_h16.h -- k_fp_h: k_fp_q (H7, H9..H11 on 16-bit codes, two columns per register half) with the horizontal
// 31-wide maximum as a ROW PIPELINE THROUGH LDS instead of DPP scans + ds_bpermute.
//
// What k_fp_q spends on a row that has a hole (about half of the row steps of a KITTI-like frame): 14 DPP scan steps, 10 maxima on
// the unpacked halves, 5 lane shifts, 8 selects that route the halo register into the scans, 4 ds_bpermutes -- 46 VALU
// instructions of the slow class for 128 columns.  Here the wave keeps the row of 31-ROW maxima as a plain array of 16-bit codes in
// its own LDS (160 columns: the strip's 128 and 16 of halo either side) and widens the window stage by stage, every stage one
// packed maximum per register on values read back at a column shift -- an LDS read at a byte offset is a shift by ANY number of
// columns, odd ones included, and costs no VALU instruction at all:
//     V0 = the 31-row maxima                                 (written by the step that finishes them)
//     W3 (c) = max3(V0(c),  V0(c + 1),  V0(c + 2))            columns c .. c + 2
//     W7 (c) = max3(W3(c),  W3(c + 2),  W3(c + 4))            c .. c + 6
//     W19(c) = max3(W7(c),  W7(c + 6),  W7(c + 12))           c .. c + 18
//     out(c) = max (W19(c - 15), W19(c - 3))                  c - 15 .. c + 15
// The stages of ONE row run in consecutive row steps (stage k of row r in step r + 30 + k), so every step runs all four stages, each
// on a different row, on what the step before wrote (LDS executes a wave's instructions in order): the reads are issued first
// thing in the step and arrive while the loads and the vertical maxima are being issued; nothing of it is live across the median.
// A lane's two columns (one register, half-word pair) do the stages as packed f16 maxima; the 32 halo columns ride one per
// lane in register B as in k_fp_q and run the same stages on 32-bit values.  4 + 3 maxima per row step instead of 46
// instructions; X7 appears three steps later than in k_fp_q (the centre values take the extra steps in a four-register ring
// behind their LDS delay line).
// Rows without a hole (and none among the four rows behind them in the pipeline) skip the stages altogether, as before.
//
// The codes are ORDERED AS f16 here (Q16: code = 256 x + 6143 lies in 0x0400 .. 0x7bff, the positive normal half-floats, whose bit
// patterns order like the integers they are), which opens the packed THREE-input instructions gfx950 has for f16 and for nothing
// else 16 bits wide -- v_pk_maximum3_f16 / v_pk_minimum3_f16, priced in tools/pk3_probe.hip at the cost of one v_pk_max_u16: the
// vertical 31-row maximum is 4 instead of 6 instructions per register, the stages above are three-input, and the median's
// closing selection folds its chain of minima (median_pk3_nets.h).
#pragma once

#include "dcmt_kernels_fp_q16.h"

namespace dcmt {

struct FpH {
    static constexpr int ARRB = 384;                                   // bytes per stage array: 160 columns + 32 of reach past the right halo
    static constexpr int WORDS = 16 * 64 + 16 * 64 + 16 * 32 + 4 * ARRB / 4;   // per wave: centre delay line, A's and B's 18-row maxima, four stage arrays
    static constexpr int LAG = 34;                                     // fill_step(t) returns X7 row t - LAG
};

#ifndef DCMT_FPH_PFD
#define DCMT_FPH_PFD 4
#endif
template <bool BLUR, bool FILLED = false>
__global__ __launch_bounds__(256)
void k_fp_h(const void* __restrict__ x6_, float* __restrict__ dst, int* __restrict__ counters,
            int rows_all, int cols, int strips, int batch, int xcd_map, float max_depth, float thr, const int* __restrict__ tb,
            int tbands)
{
    __shared__ __attribute__((aligned(16))) unsigned s_mem[4][FpH::WORDS];      // 11.5 KiB per wave, 46 KiB per workgroup
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int f, strip;
    if (!wave_strip(blockIdx.x, wave, strips, batch, xcd_map, f, strip)) return;
    int* cnt = frame_counters(counters, f);
    const size_t fo = (size_t)f * rows_all * cols;
    constexpr int HALO = FpP::H, VW = FpP::VW;
    const int gx0 = strip * VW - HALO;
    const int gxe = gx0 + 2 * lane;
    // Stage arrays: slot j holds column gx0 - 16 + j.  A's pair sits at j = 16 + 2 lane; B, one column per lane: the right halo
    // (columns gx0 + 128 ..) in lanes 0..15 at j = 144 + lane, the left halo (gx0 - 16 ..) in lanes 48..63 at j = lane - 48.  The dead
    // lanes 16..47 shadow lane 0: same column, same values, same words -- their stores write what lane 0 writes.
    const int kb = lane < 16 ? 144 + lane : (lane >= 48 ? lane - 48 : 144);
    const int gxb = gx0 - 16 + kb;
    const int gxec = min(max(gxe, 0), cols - 2), gxoc = gxec + 1, gxbc = min(max(gxb, 0), cols - 1);
    int tie = 0, tio = 0, tib = 0, bie = rows_all - 1, bio = rows_all - 1, bib = rows_all - 1, V = 0;
    if (tb) {
        table_rows(tb, f, cols, tbands, rows_all, gxec, tie, bie);
        table_rows(tb, f, cols, tbands, rows_all, gxoc, tio, bio);
        table_rows(tb, f, cols, tbands, rows_all, gxbc, tib, bib);
        V = __builtin_amdgcn_readfirstlane(max(wave_min_i(min(min(tie, tio), tib)) - 8, 0));
    }
    const int rows = rows_all - V;
    FrameBuf sf;
    sf.init(reinterpret_cast<const float*>(static_cast<const char*>(x6_) + fo * 2u), (size_t)rows_all * cols / 2);
    const unsigned rowb = 2u * (unsigned)cols;
    const unsigned sbe = 2u * (unsigned)gxec + (unsigned)V * rowb, sbo = sbe + 2u, sbb = 2u * (unsigned)gxbc + (unsigned)V * rowb;
    const unsigned fle = 2u * (unsigned)gxec + (unsigned)max(tie, V) * rowb, cee = 2u * (unsigned)gxec + (unsigned)max(bie, V) * rowb;
    const unsigned flo = 2u * (unsigned)gxoc + (unsigned)max(tio, V) * rowb, ceo = 2u * (unsigned)gxoc + (unsigned)max(bio, V) * rowb;
    const unsigned flb = 2u * (unsigned)gxbc + (unsigned)max(tib, V) * rowb, ceb = 2u * (unsigned)gxbc + (unsigned)max(bib, V) * rowb;
    auto clamp3 = [](unsigned a, unsigned lo, unsigned hi) -> unsigned { unsigned r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi)); return r; };
    auto ld_code = [&](unsigned off) -> unsigned { return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(sf.rs, off, 0, 0); };
    struct Raw { unsigned e, o, b; };
    auto ld_row = [&](int row) -> Raw {                              // row relative to V, already clamped to [0, rows)
        return {ld_code(clamp3(sbe + (unsigned)row * rowb, fle, cee)), ld_code(clamp3(sbo + (unsigned)row * rowb, flo, ceo)),
                ld_code(clamp3(sbb + (unsigned)row * rowb, flb, ceb))};
    };
    auto pack = [](unsigned e, unsigned o) -> unsigned { return e | (o << 16); };
    const bool outside = gxe < 0 || gxe >= cols;
    const bool own = !outside && 2 * lane >= HALO && 2 * lane < 128 - HALO;
    const unsigned long long own_mask = __ballot(own);
    const bool edge_strip = gx0 < 0 || gx0 + 127 >= cols;
    const int rep_l = min(max((0 - gx0) >> 1, 0), 63), rep_r = min(max((cols - 2 - gx0) >> 1, 0), 63);
    unsigned* sd = s_mem[wave];
    unsigned (*dl_c)[64] = reinterpret_cast<unsigned (*)[64]>(sd);
    unsigned (*dl_a)[64] = reinterpret_cast<unsigned (*)[64]>(sd + 16 * 64);
    unsigned (*dl_b)[32] = reinterpret_cast<unsigned (*)[32]>(sd + 16 * 128);
    const int lb = lane < 16 ? lane : (lane >= 48 ? lane - 32 : 0);
    char* const arr = reinterpret_cast<char*>(sd + 16 * 160);
    char* const pa = arr + 32 + 4 * lane;                              // this lane's pair in stage array 0
    char* const pb = arr + 2 * kb;                                     // B's column in stage array 0
    constexpr int AB = FpH::ARRB;
    auto rd32 = [](const char* p) -> unsigned { unsigned w; __builtin_memcpy(&w, p, 4); return w; };       // any 2-byte aligned address: ds_read_b32
    auto rd16 = [](const char* p) -> unsigned { return *reinterpret_cast<const unsigned short*>(p); };
    auto wr32 = [](char* p, unsigned w) { *reinterpret_cast<unsigned*>(p) = w; };
    auto wr16 = [](char* p, unsigned w) { *reinterpret_cast<unsigned short*>(p) = (unsigned short)w; };

    PostPipeP<BLUR, HALO, FILLED, FILLED> pipe;                      // only its after_median() half is used
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
    for (int q = 0; q < 16; ++q) { dl_c[q][lane] = xa0; dl_a[q][lane] = xa0; dl_b[q][lb] = xb0; }
    constexpr int PFD = DCMT_FPH_PFD;        // rows of load lookahead
#pragma unroll
    for (int q = 0; q < PFD; ++q) {
        const Raw r = ld_row(min(max(q + (warm ? 16 : 0) - 15, 0), rows - 1));
        PFA[q] = pack(r.e, r.o); PFB[q] = r.b;
    }
    unsigned vpa = xa0, vpb = xb0, x7_prev = xa0;
    int before = 0, after = 0;
    unsigned nxt_c = xa0, nxt_a = xa0, nxt_b = xb0;
    unsigned VC[4] = {xa0, xa0, xa0, xa0};                           // centre values of rows t - 31 .. t - 34
    int live = 0;                                                    // row steps the stage pipeline still has a row with holes in it

    // the fill front end of step t: returns X7 (packed codes) of image row t - 34
    auto fill_step = [&](auto P_, int t) -> unsigned {
        constexpr int p = decltype(P_)::value;
        const unsigned v = nxt_c;                                    // the centre value of row t - 30
        if ((__builtin_amdgcn_ballot_w64((v << 16) <= Q16::HOLE_MAX_HI) | __builtin_amdgcn_ballot_w64(v <= Q16::HOLE_MAX_HI)) != 0ull) live = 5;
        // what the step before left in the stage arrays (rows t - 31 .. t - 34), issued first: the loads and the vertical maxima below
        // run while these are on their way.  (Values read outside a run are never looked at.)
        unsigned a10, a11, a12, a20, a21, a22, a30, a31, a32, a40, a41, b10, b11, b12, b20, b21, b22, b30, b31, b32;
        if (live > 0) {                                             // a row with holes is somewhere in the four stages (or enters them now)
            a10 = rd32(pa); a11 = rd32(pa + 2); a12 = rd32(pa + 4);
            a20 = rd32(pa + AB); a21 = rd32(pa + AB + 4); a22 = rd32(pa + AB + 8);
            a30 = rd32(pa + 2 * AB); a31 = rd32(pa + 2 * AB + 12); a32 = rd32(pa + 2 * AB + 24);
            a40 = rd32(pa + 3 * AB - 30); a41 = rd32(pa + 3 * AB - 6);
            b10 = rd16(pb); b11 = rd16(pb + 2); b12 = rd16(pb + 4);
            b20 = rd16(pb + AB); b21 = rd16(pb + AB + 4); b22 = rd16(pb + AB + 8);
            b30 = rd16(pb + 2 * AB); b31 = rd16(pb + 2 * AB + 12); b32 = rd16(pb + 2 * AB + 24);
        }
        const unsigned xa = PFA[p], xb = PFB[p];
        {
            const Raw r = ld_row(min(max(t + PFD - 15, 0), rows - 1));
            PFA[(p + PFD) & 15] = pack(r.e, r.o); PFB[(p + PFD) & 15] = r.b;
        }
        // vertical 31-max: A packed (f16 order: three-input), B unpacked
        const unsigned w2a = hmax2(xa, vpa);
        vpa = xa;
        W2A[p] = w2a;
        const unsigned w6a = hmax3(w2a, W2A[(p + 14) & 15], W2A[(p + 12) & 15]);
        W6A[p] = w6a;
        const unsigned w18a = hmax3(w6a, W6A[(p + 10) & 15], W6A[(p + 4) & 15]);
        const unsigned w18a_old = nxt_a;
        nxt_c = dl_c[(p + 2) & 15][lane];
        nxt_a = dl_a[(p + 4) & 15][lane];
        dl_c[p][lane] = xa;
        dl_a[p][lane] = w18a;
        const unsigned w31a = hmax2(w18a, w18a_old);
        const unsigned w2b = umax2(xb, vpb);
        vpb = xb;
        W2B[p] = w2b;
        const unsigned w6b = umax3(w2b, W2B[(p + 14) & 15], W2B[(p + 12) & 15]);
        W6B[p] = w6b;
        const unsigned w18b = umax3(w6b, W6B[(p + 10) & 15], W6B[(p + 4) & 15]);
        const unsigned w18b_old = nxt_b;
        nxt_b = dl_b[(p + 4) & 15][lb];
        dl_b[p][lb] = w18b;
        const unsigned w31b = umax2(w18b, w18b_old);
        // the row whose X7 this step returns: t - 34
        const int o = t - FpH::LAG;
        const unsigned v4 = VC[p & 3];
        VC[p & 3] = v;
        unsigned x7 = v4;
        asm volatile("" : "+s"(live));                              // (the same test again, on the scalar unit: no mask kept in a VGPR across the block above)
        if (live > 0) {
            // the stage values are looked at HERE, behind the vertical maxima (the empty statements tie them to w31a / w31b: without them the
            // compiler computes the stages right behind the reads and waits for those first thing in the step; B's also hide that they are
            // 16-bit values, which the compiler would otherwise pack into pairs at three instructions per pair)
            asm volatile("" : "+v"(a10), "+v"(a20), "+v"(a30), "+v"(a40) : "v"(w31a));
            asm volatile("" : "+v"(b10), "+v"(b11), "+v"(b12), "+v"(b20), "+v"(b21), "+v"(b22), "+v"(b30), "+v"(b31), "+v"(b32) : "v"(w31b));
            __builtin_amdgcn_wave_barrier();
            // stage 0 = this step's maxima (row t - 30), stages 1..3 on rows t - 31 .. t - 33, the closing maximum on row t - 34
            wr32(pa, w31a); wr16(pb, w31b);
            wr32(pa + AB, hmax3(a10, a11, a12)); wr16(pb + AB, umax3(b10, b11, b12));
            wr32(pa + 2 * AB, hmax3(a20, a21, a22)); wr16(pb + 2 * AB, umax3(b20, b21, b22));
            wr32(pa + 3 * AB, hmax3(a30, a31, a32)); wr16(pb + 3 * AB, umax3(b30, b31, b32));
            __builtin_amdgcn_wave_barrier();
            const unsigned long long hme = __builtin_amdgcn_ballot_w64((v4 << 16) <= Q16::HOLE_MAX_HI), hmo = __builtin_amdgcn_ballot_w64(v4 <= Q16::HOLE_MAX_HI);
            if ((hme | hmo) != 0ull) {
                const unsigned out = hmax2(a40, a41);
                const bool he = __builtin_amdgcn_inverse_ballot_w64(hme), ho = __builtin_amdgcn_inverse_ballot_w64(hmo);
                const unsigned m = (he ? 0xffffu : 0u) | (ho ? 0xffff0000u : 0u);
                x7 = (out & m) | (v4 & ~m);
                if ((unsigned)o < (unsigned)rows) {
                    before += __builtin_popcountll(hme & own_mask) + __builtin_popcountll(hmo & own_mask);
                    after += __builtin_popcountll(__builtin_amdgcn_ballot_w64((x7 << 16) <= Q16::HOLE_MAX_HI) & own_mask) +
                             __builtin_popcountll(__builtin_amdgcn_ballot_w64(x7 <= Q16::HOLE_MAX_HI) & own_mask);
                }
            }
            --live;
        }
        if (edge_strip) {                                           // out-of-image columns replicate the edge column
            const unsigned l0 = (unsigned)__shfl((int)x7, rep_l, 64), r0 = (unsigned)__shfl((int)x7, rep_r, 64);
            if (gxe < 0) x7 = (l0 & 0xffffu) | (l0 << 16);
            if (gxe >= cols) x7 = (r0 >> 16) | (r0 & 0xffff0000u);
        }
        if (o >= rows) { asm volatile("" ::); x7 = x7_prev; }
        x7_prev = x7;
        return x7;
    };
    // post step u: the median of image row u - 4 on packed pairs, then PostPipeP's f32 tail
    auto post_step = [&](auto PP_, unsigned x, int u) {
        constexpr int PP = decltype(PP_)::value;
        const unsigned rl = u_left(x), rr = u_right(x);              // columns 2l-2, 2l-1 | 2l+2, 2l+3
        unsigned s[5] = {rl, __builtin_amdgcn_alignbit(x, rl, 16), x, __builtin_amdgcn_alignbit(rr, x, 16), rr};
        q_sort5(s);
        const unsigned m = mc.template step<PP>(s);
        pipe.template after_median<PP>(Q16::value(m & 0xffffu), Q16::value(m >> 16), u);
    };

    // steps 0..31 (16..31 after a warm start): fill only
    for (int t0 = warm ? 16 : 0; t0 < 32; t0 += 16) {
        static_for<0, 16>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            (void)fill_step(P_, t0 + p);
        });
    }
    // steps 32..rows+37: post step u = t - 32 takes X7 row u - 2 = t - 34, the row this step's fill front end returns.  X7 row 0 appears
    // in step 34, and the post pipeline takes it three times (its replicated rows -2 and -1, and row 0: post steps 0, 1, 2)
    const int nsteps = rows + 38;
    for (int t0 = 32; t0 < nsteps; t0 += 16) {
        static_for<0, 16>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            const int t = t0 + p, u = t - 32;
            const unsigned x7 = fill_step(P_, t);
            if constexpr (p < 2) {
                if (t0 != 32) post_step(std::integral_constant<int, (p & 7)>{}, x7, u);
            } else if constexpr (p == 2) {
                if (t0 == 32) {
                    post_step(std::integral_constant<int, 0>{}, x7, 0);
                    post_step(std::integral_constant<int, 1>{}, x7, 1);
                }
                post_step(std::integral_constant<int, 2>{}, x7, u);
            } else {
                post_step(std::integral_constant<int, (p & 7)>{}, x7, u);
            }
            if constexpr (p == 6) {
                // u == 6: output row 0 of the shifted frame (image row V) has just been stored; the V rows above it are equal
                if (t0 == 32 && V > 0) {
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
