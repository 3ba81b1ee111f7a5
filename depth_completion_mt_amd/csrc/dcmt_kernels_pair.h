// dcmt_kernels_pair.h -- k_pre_p: H2..H6 (k_pre_s of dcmt_kernels_fused.h) with TWO adjacent columns per lane.
//
// A wave64 owns 128 columns over the full image height: lane l holds columns c0 + 2l ("E") and c0 + 2l + 1 ("O") in two
// registers and streams down the rows like k_pre_s.  What that buys (every max / min / select / DPP instruction costs the
// same ~4.4 issue cycles whatever it does; with 313 M of them per 1024 frames k_pre_s sat at the edge between issue- and
// memory-bound, this kernel needs 218 M and is bound by memory alone, which is what small batches and the label-masked
// variant's H5 + H6 feel):
//   * half of every horizontal window is already in the lane.  With m = max(E, O):
//         3-window   E: max(O[l-1], m)            O: max(E[l+1], m)                      3 instructions per column pair (4)
//         r -> r+1   E: max(W_O[l-1], W_O[l])     O: max(W_E[l+1], W_E[l])   (r >= 1)    2 instructions per column pair (4)
//     so the 5-wide windows of the close cost 5 instead of 8 and the 7-wide window of the small fill 7 instead of 12,
//     each [l-1] / [l+1] a DPP wave shift folded into the v_max / v_min;
//   * the chain's reach (7 columns left, 9 right for the as-compiled element) costs 18 of 128 lanes-columns instead of
//     20 of 64: 108 output columns per wave instead of 44;
//   * rows arrive as 8 bytes per lane (buffer_load_dwordx2, 6 rows ahead) and leave as one buffer_store_dwordx2: 512-byte
//     row segments.  (An LDS-DMA ring like k_pre_s's was measured too: 6 % slower here -- the plain loads already move whole
//     lines, and the ring ties the loads' waits to the stores.)
// Per row step: 64 instructions (56 of the 4.4-cycle class) for 108 columns (k_pre_s: 41.5 for 44).
//
// Requires an even number of columns (a lane's two columns are both inside the image or both outside, its 8-byte accesses
// are aligned).  Other shapes run k_pre_s.
//
// H6 (column extension): the zones above a column's first valid row ti and below its last valid row bi are constant down
// the column.  In table mode (tb != nullptr) they are not written at all: the kernel records (ti, bi) per column and the
// reader clamps its row index into [ti, bi] (k_fp_s, one v_med3 per load) -- a third of X6 never crosses HBM on
// velodyne-like frames, and row bands of one strip (small batches) need no pass over each other's rows.  Without a
// table the zones are written as the reference does (probes, stop_after, the unfused kernels).
#pragma once

#include "dcmt_kernels_fused.h"

namespace dcmt {

struct F2 { float e, o; };

__device__ __forceinline__ F2 p_max(F2 a, F2 b) { return {fmax2(a.e, b.e), fmax2(a.o, b.o)}; }
__device__ __forceinline__ F2 p_max3(F2 a, F2 b, F2 c) { return {fmax3(a.e, b.e, c.e), fmax3(a.o, b.o, c.o)}; }
__device__ __forceinline__ F2 p_min3(F2 a, F2 b, F2 c) { return {fmin3(a.e, b.e, c.e), fmin3(a.o, b.o, c.o)}; }
__device__ __forceinline__ F2 p_sel(bool c, F2 a, F2 b) { return {c ? a.e : b.e, c ? a.o : b.o}; }
// windows over columns, for both columns of every lane
__device__ __forceinline__ F2 p_hmax3(F2 v) { const float m = fmax2(v.e, v.o); return {fmax2(from_left(v.o), m), fmax2(from_right(v.e), m)}; }
__device__ __forceinline__ F2 p_hmin3(F2 v) { const float m = fmin2(v.e, v.o); return {fmin2(from_left(v.o), m), fmin2(from_right(v.e), m)}; }
__device__ __forceinline__ F2 p_grow_max(F2 w) { return {fmax2(from_left(w.o), w.o), fmax2(from_right(w.e), w.e)}; }   // radius r >= 1 -> r + 1
__device__ __forceinline__ F2 p_grow_min(F2 w) { return {fmin2(from_left(w.o), w.o), fmin2(from_right(w.e), w.e)}; }

typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
// (cast the loaded vector as a whole: with hipcc 7.2, bit-casting its elements one by one -- bit_cast<float>(v.y) on the builtin's
// result -- compiles to a ONE-dword load whose value serves as both elements)
__device__ __forceinline__ F2 ld2(const FrameBuf& b, unsigned lane_bytes, int row, int cols)
{
    const f2v v = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(b.rs, lane_bytes, row * cols * 4, 0));
    return {v.x, v.y};
}
__device__ __forceinline__ void st2(const FrameBuf& b, unsigned lane_bytes, int row, int cols, F2 v)
{
    const f2v f = {v.e, v.o};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, f), b.rs, lane_bytes, row * cols * 4, 0);
}

template <int K0KIND, bool START4>
struct PreP {
    // columns lost to the left / right of a strip: the chain's horizontal reach (k_pre_s has the derivation); START4 runs H5 only
    static constexpr int HL0 = START4 ? 3 : (K0KIND == K0_AS_COMPILED ? 0 : 2) + 2 + 2 + 3;
    static constexpr int HR0 = START4 ? 3 : 2 + 2 + 2 + 3;
    static constexpr int HL = (HL0 + 3) / 4 * 4;                    // strip origins are multiples of 4 columns: whole 16-byte pieces
    static constexpr int VW = (128 - HL - HR0) / 4 * 4;             // output columns per wave, a multiple of 4 like HL: 108 (as compiled), 104 (diamond), 120 (START4)
    static constexpr int LAT = 9;
};

constexpr int kMaxBands = 8;                 // row bands per strip (table slots per frame)

// Row bands: with few frames a grid of full-height strips cannot fill the GPU (128 frames x 12 strips = 1536 waves for 4096
// slots), so a strip may be cut into `bands` row ranges, one wave each.  A band starts its stream 18 rows above its first row
// with cold rings (exactly like the start below the leading empty rows) and accounts for the x5 rows [r0, r1) only; the
// per-column (ti, bi) of the bands go to one table slot per band, which the reader combines (table_rows), which is why
// bands need table mode.
// Q16OUT: X6 leaves as 16-bit codes (Q16 in dcmt_kernels_fused.h: code = 256 x + 6143, two columns per dword) for k_fp_h / k_fp_q -- half the X6 traffic.
// Exact only if every value stored is a multiple of 1/256 in the code range, which holds whenever the frame's depths are
// (the KITTI format); the kernel checks it on every value it really stores and raises *q16_bad otherwise (the caller then
// reruns the f32 kernels, gated on that flag).  Table mode only.
// gate: a launch that only runs if *gate != 0 (the f32 rerun behind a Q16 attempt); nullptr = always.
// q16_seen: a word of mapped host memory set together with the flag, so that the host can stop attempting on data that is no grid.
// q16_clear: nullptr, or the flag the next attempt will use (cleared by this launch's first thread).

#ifndef DCMT_PRE_WAVES
#define DCMT_PRE_WAVES 0
#endif
template <int K0KIND, bool START4 = false, bool U16 = false, bool NORM = false, bool Q16OUT = false>
__global__ __launch_bounds__(256)
#if DCMT_PRE_WAVES
__attribute__((amdgpu_waves_per_eu(DCMT_PRE_WAVES, DCMT_PRE_WAVES)))
#endif
void k_pre_p(const void* __restrict__ src_, float* __restrict__ x6, int rows, int cols, int strips, int bands,
             int batch, int xcd_map, float max_depth, float thr, float in_scale, const float* __restrict__ coef,
             int* __restrict__ tb, int* __restrict__ counters, int* __restrict__ q16_bad, const int* __restrict__ gate, int* __restrict__ q16_seen,
             int* __restrict__ q16_clear)
{
    static_assert(!(U16 && START4), "the uint16 ingest is the first kernel of the path");
    static_assert(!(NORM && (U16 || START4)), "normalisation applies to raw f32 frames");
    static_assert(!(NORM && Q16OUT), "normalised frames are not multiples of 1/256");
    if (gate && *gate == 0) return;
    // the flag of the NEXT 16-bit attempt (a ring of flags, dcmt.hip): cleared here, a whole call ahead of its use, instead of by a
    // memset in the stream in front of every call (a fill is a dependent operation with a 30-70 us bubble behind the previous kernel)
    if (q16_clear && blockIdx.x == 0 && threadIdx.x == 0) *q16_clear = 0;
    const float* src = static_cast<const float*>(src_);
    using G = PreP<K0KIND, START4>;
    constexpr int ROFF = START4 ? 6 : 0;         // image row of stream row s is s - ROFF
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int f, unit;
    if (!wave_strip(blockIdx.x, wave, strips * bands, batch, xcd_map, f, unit)) return;   // whole waves leave; no barrier is used below
    const int strip = unit / bands, band = unit - strip * bands;
    clear_frame_counters(counters, f, unit == 0, lane);
    const int gx0 = strip * G::VW - G::HL;
    const int gx = gx0 + 2 * lane;                                   // column of E; O = gx + 1
    const bool incol = gx >= 0 && gx < cols;                         // cols is even and gx is even: E and O are inside or outside together
    const bool outlane = incol && 2 * lane >= G::HL && 2 * lane < G::HL + G::VW;
    const size_t fo = (size_t)f * rows * cols;
    const int gxc = min(max(gx, 0), cols - 2);
    float na = 1.0f, nb = 0.0f;
    if constexpr (NORM) { na = coef[2 * f]; nb = coef[2 * f + 1]; }
    FrameBuf ob, ib;
    if constexpr (Q16OUT) ob.init(reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(x6) + fo), (size_t)rows * cols / 2);
    else ob.init(x6 + fo, (size_t)rows * cols);
    if constexpr (U16) ib.init(reinterpret_cast<const float*>(static_cast<const uint16_t*>(src_) + fo), (size_t)rows * cols / 2);
    else ib.init(src + fo, (size_t)rows * cols);
    const unsigned oc = 4u * (unsigned)gxc;                          // byte offset of the column pair in an f32 row
    const unsigned qc = 2u * (unsigned)gxc;                          // ... in a row of 16-bit codes
    unsigned bad = 0;                                                // Q16OUT: a stored value was not a code
    auto load_row = [&](int r) -> F2 {            // image row r (already clamped) of this lane's two columns, in metres
        if constexpr (U16) {
            const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(ib.rs, 2u * (unsigned)gxc, r * cols * 2, 0);
            return {__fmul_rn((float)(w & 0xffffu), in_scale), __fmul_rn((float)(w >> 16), in_scale)};
        } else {
            return ld2(ib, oc, r, cols);
        }
    };

    // rows this wave accounts for
    const int r0 = (int)((long long)rows * band / bands), r1 = (int)((long long)rows * (band + 1) / bands);

    constexpr float NEG = -FLT_MAX, POS = FLT_MAX;
    const F2 NEG2 = {NEG, NEG}, POS2 = {POS, POS};
    F2 PF[8];                                    // prefetched input rows (plain loads)
    float OX[8], S1E[8];                         // as compiled: x2 of the odd column, x2 of the even column of the next lane
    F2 XR[8], A3[8];                             // diamond: x2 rows, horizontal 3-max rows
    F2 H4[8], HE[8], H7[8], E4[8], T7[8];
#ifndef DCMT_PAIR_PFD
#define DCMT_PAIR_PFD 6
#endif
    constexpr int PFD = DCMT_PAIR_PFD;           // rows of load lookahead (plain loads; the ring has 8 slots)

    // ---- where the stream starts: below the leading empty rows (k_pre_s has the argument), or 18 rows above the band
    // START4 (the input is X4, only H5 runs): x5(m) reaches X4 rows m-3 .. m+3 and step i feeds X4 row i - 6 while finishing x5 row
    // i - 9, so the stream may start at step zv itself and its x5 rows are exact from S - 3 on.
    // Warm start: if the rows above zv are not just empty but exactly 0.0 in the wave's columns (a KITTI frame's empty pixels are),
    // every stage's rows above zv - 6 are 0.0 too (each depends on input rows at most 6 further down), so the stream starts AT zv
    // with its rings holding those zeros -- exact from x5 row zv - 9 on -- instead of 24 rows earlier with cold rings.
    constexpr int DZ = START4 ? 0 : 18, DM = START4 ? -3 : 9;
    int S = 0;
    bool warm0 = false;
    if (band > 0) S = max(r0 - 18, 0) & ~7;
    else {
        // CH-row chunks, two in flight: the scan is a chain of dependent round trips to memory, one per chunk; the two chunks in flight when
        // the first valid row turns up are read again by the stream (2 * CH rows per wave).  Measured, CH = 16 / 8 / 4 / 2: k_pre_p 0.597 / 0.573-0.583 /
        // 0.585-0.61 / 0.585-0.62 ms per 1024 frames, 2.08 / 1.92 / 1.88 GB read
#ifndef DCMT_SCAN_CH
#define DCMT_SCAN_CH 8
#endif
        constexpr int CH = DCMT_SCAN_CH;
        bool zeros = true;                                           // every chunk above zv held nothing but 0.0
        auto valid16 = [&](const F2 (&v)[CH]) -> bool {
            // the bit patterns are OR-ed (v_or3_b32): "exactly +0.0 everywhere" is a statement about bits -- rows holding a -0.0 take the
            // cold start (the warm start would seed the rings with +0.0)
            float m = fmax2(v[0].e, v[0].o);
            unsigned ob = __builtin_bit_cast(unsigned, v[0].e) | __builtin_bit_cast(unsigned, v[0].o);
#pragma unroll
            for (int q = 1; q < CH; q += 1) { m = fmax3(m, v[q].e, v[q].o); ob = ob | __builtin_bit_cast(unsigned, v[q].e) | __builtin_bit_cast(unsigned, v[q].o); }
            const bool valid = __builtin_amdgcn_ballot_w64(m >= thr) != 0ull;
            if (!valid) zeros = zeros && __builtin_amdgcn_ballot_w64(ob != 0u) == 0ull;
            return valid;
        };
        auto load16 = [&](F2 (&v)[CH], int z) {
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                F2 x = load_row(min(z + q, rows - 1));
                if constexpr (NORM) x = {norm_apply(x.e, na, nb), norm_apply(x.o, na, nb)};
                v[q] = x;
            }
        };
        F2 va[CH], vb[CH];
        int zv = r1;
        load16(va, 0);
        for (int z = 0; z < r1; z += 2 * CH) {
            load16(vb, z + CH);
            if (valid16(va)) { zv = z; break; }
            if (z + CH >= r1) break;
            load16(va, z + 2 * CH);
            if (valid16(vb)) { zv = z + CH; break; }
        }
        warm0 = !START4 && zeros && zv >= 32 && zv < r1;
        S = warm0 ? zv : max(zv - DZ, 0) & ~7;
    }
    const int m0 = warm0 ? S - 9 : max(S > 0 ? S + (band > 0 ? 9 : DM) : 0, r0);   // first x5 row this wave accounts for
    {
        const float ng = warm0 ? 0.0f : NEG, ps = warm0 ? 0.0f : POS;
        const F2 ng2 = {ng, ng}, ps2 = {ps, ps};
#pragma unroll
        for (int q = 0; q < 8; ++q) { OX[q] = ng; S1E[q] = ng; XR[q] = ng2; A3[q] = ng2; H4[q] = ng2; HE[q] = ps2; H7[q] = ng2; E4[q] = ng2; T7[q] = ng2; PF[q] = {0.f, 0.f}; }
    }

#pragma unroll
    for (int q = 0; q < PFD; ++q) PF[q] = load_row(min(max(S + q - ROFF, 0), rows - 1));

    int tie = 0x7fffffff, tio = 0x7fffffff, bie = -1, bio = -1;   // first / last valid row of X5 in this lane's two columns (within the band)
    unsigned long long seen_e = 0, seen_o = 0;                   // columns that have had their first valid row (wave masks)
    // Q16OUT, raw f32 input: the grid check is made on every INPUT value the stream consumes (below); the running maxima of the codes
    // and the OR of the residues are looked at once, behind the loop
    unsigned chk_max = 0, chk_or = 0;

    const int nsteps = r1 + G::LAT;
    // One row step.  INNER: every row the step touches (i - 6 .. i) lies inside the image, every lane's columns do, and the x5 row
    // it finishes belongs to this wave -- the border selects (one per stage and column) and the row tests are compiled out.  The
    // host of the loop below decides per block of eight steps; strips at the image's left / right edge never qualify.
    auto row_step = [&](auto INNER_, auto P_, int i) __attribute__((always_inline)) {
        constexpr bool INNER = decltype(INNER_)::value;
        constexpr int p = decltype(P_)::value;
        F2 raw = PF[p];
        PF[(p + PFD) & 7] = load_row(INNER ? i + PFD - ROFF : min(max(i + PFD - ROFF, 0), rows - 1));
        F2 e4;
        const int l = i - 6;
        if constexpr (START4) {
            e4 = raw;                                               // X4 row l
        } else {
            if constexpr (NORM) raw = {norm_apply(raw.e, na, nb), norm_apply(raw.o, na, nb)};
            if constexpr (Q16OUT && !U16) {
                // is every depth a multiple of 1/256 m with a code (Q16: 256 x in [-5119, 30719])?  t = fma(x, 256, 1.5 * 2^23) lies in
                // [2^23, 2^24) for every such x, where the float IS the integer 256 x + 1.5 * 2^23 (low mantissa bits): its distance from
                // the smallest admissible value, as an unsigned number, is at most 35838 -- anything out of range wraps to more --, and
                // 256 x - (t - 1.5 * 2^23), one fused operation, is zero iff nothing was rounded away.  Adds, not compares: these issue
                // beside the chain's max / min instructions.  Every value loaded is a pixel of the frame (clamped rows and columns repeat
                // pixels), so no lane needs masking; the values the kernel stores are selections of the inputs and of 100 - input.
                constexpr float kMg = 12582912.0f;                   // 1.5 * 2^23
                constexpr unsigned kBase = 0x4B000000u + 4194304u - 5119u;
                const float te = __builtin_fmaf(raw.e, 256.0f, kMg), to = __builtin_fmaf(raw.o, 256.0f, kMg);
                const float re = __builtin_fmaf(raw.e, 256.0f, -__fsub_rn(te, kMg)), ro = __builtin_fmaf(raw.o, 256.0f, -__fsub_rn(to, kMg));
                chk_max = max(max(chk_max, __builtin_bit_cast(unsigned, te) - kBase), __builtin_bit_cast(unsigned, to) - kBase);
                chk_or |= __builtin_bit_cast(unsigned, re) | __builtin_bit_cast(unsigned, ro);
                asm volatile("" : "+v"(chk_max), "+v"(chk_or));          // here, not "some time in this block of eight": left to itself the scheduler
                                                                         // defers the check and keeps the raw rows of all eight steps alive (+22 VGPRs)
            }
            // ---- H2 on load (LO :55-67); outside the image: the dilate border value
            F2 x2 = {invert_valid(raw.e, max_depth, thr), invert_valid(raw.o, max_depth, thr)};
            if constexpr (!INNER) x2 = p_sel(incol && i < rows, x2, NEG2);
            // ---- H3 (LO :71-80), row j = i - 2
            const int j = i - 2;
            F2 y3;
            if constexpr (K0KIND == K0_AS_COMPILED) {
                // dst(r,c) = max(src(r-1,c+1), src(r+2,c+2)).  c = 2l: O[l] of row j-1, E[l+1] of row j+2; c = 2l+1: E[l+1] of row j-1, O[l+1] of row j+2
                const float s1e = from_right(x2.e);
                y3.e = fmax2(OX[(p + 5) & 7], s1e);
                y3.o = fmax2(from_right(x2.o), S1E[(p + 5) & 7]);
                OX[p] = x2.o;
                S1E[p] = s1e;
            } else {
                // 13-tap diamond: rows j-2 and j+2 centre only, j-1 and j+1 three wide, j five wide
                const F2 a3 = p_hmax3(x2);
                XR[p] = x2;
                A3[p] = a3;
                const F2 a5j = p_grow_max(A3[(p + 6) & 7]);          // row j
                y3 = p_max(p_max3(XR[(p + 4) & 7] /* j-2 */, A3[(p + 5) & 7] /* j-1 */, a5j), p_max(A3[(p + 7) & 7] /* j+1 */, x2 /* j+2 */));
            }
            if constexpr (!INNER) y3 = p_sel(incol && (unsigned)j < (unsigned)rows, y3, NEG2);
            // ---- H4 dilate 5x5 (LO :85): horizontal on row j, vertical gives row k = j - 2
            H4[(p + 6) & 7] = p_grow_max(p_hmax3(y3));
            const int k = i - 4;
            F2 d4 = p_max3(p_max3(H4[(p + 2) & 7], H4[(p + 3) & 7], H4[(p + 4) & 7]), H4[(p + 5) & 7], H4[(p + 6) & 7]);
            if constexpr (!INNER) d4 = p_sel(incol && (unsigned)k < (unsigned)rows, d4, POS2);     // border value of the erode
            // ---- H4 erode 5x5: row l = k - 2
            HE[(p + 4) & 7] = p_grow_min(p_hmin3(d4));
            e4 = p_min3(p_min3(HE[(p + 0) & 7], HE[(p + 1) & 7], HE[(p + 2) & 7]), HE[(p + 3) & 7], HE[(p + 4) & 7]);
        }
        if constexpr (!INNER) e4 = p_sel(incol && (unsigned)l < (unsigned)rows, e4, NEG2);         // border value of the 7x7 dilate
        E4[(p + 2) & 7] = e4;                                                // slot of row l = i-6
        // ---- H5 (LO :88-100): dilate 7x7, row m = l - 3, then fill where x < 0.1
        H7[(p + 2) & 7] = p_grow_max(p_grow_max(p_hmax3(e4)));
        const int m = i - 9;
        T7[p] = p_max3(H7[(p + 0) & 7], H7[(p + 1) & 7], H7[(p + 2) & 7]);   // rows i-8 .. i-6
        const F2 d7 = p_max3(T7[p], T7[(p + 5) & 7] /* rows i-11 .. i-9 */, H7[(p + 4) & 7] /* row i-12 */);
        const F2 e = E4[(p + 7) & 7];                                        // row m = i-9
        const F2 x5 = {e.e < thr ? d7.e : e.e, e.o < thr ? d7.o : e.o};
        const bool inrows = INNER || (m >= m0 && m < r1);
        if (inrows) {
            // ---- H6 bookkeeping (LO :112-121): first / last row with x >= 0.1.  The compare's wave mask is all that is needed: the last
            // valid row is "this row, where valid", the first one "this row, where valid for the first time" -- one select each
            const unsigned long long ve = __builtin_amdgcn_ballot_w64(x5.e >= thr), vo = __builtin_amdgcn_ballot_w64(x5.o >= thr);
            tie = __builtin_amdgcn_inverse_ballot_w64(ve & ~seen_e) ? m : tie; tio = __builtin_amdgcn_inverse_ballot_w64(vo & ~seen_o) ? m : tio;
            bie = __builtin_amdgcn_inverse_ballot_w64(ve) ? m : bie; bio = __builtin_amdgcn_inverse_ballot_w64(vo) ? m : bio;
            seen_e |= ve; seen_o |= vo;
        }
        // both columns leave in one 8-byte store once either has had its first valid row (what lands above a column's own first
        // valid row is never read in table mode and rewritten by the epilogue otherwise); the store itself is issued on every
        // step by every lane, aimed past the buffer when there is nothing to write (no branch in the row step)
        // (a band below the first one stores all its rows: the rows between a column's first valid row in an upper band and its
        // first one here are real holes that the reader looks at)
        const bool real = inrows && outlane && (band > 0 || __builtin_amdgcn_inverse_ballot_w64(seen_e | seen_o));
        if constexpr (Q16OUT) {
            // code = 256 x + OFFSET through the float adder: y = fma(x, 256, OFFSET + 2^23) lies in [2^23, 2^24) for every code, where the
            // low mantissa bits ARE the integer -- no v_cvt; the two low halves leave as one dword (v_perm_b32)
            constexpr float kMagic = (float)(Q16::OFFSET + 8388608);
            const float ye = __builtin_fmaf(x5.e, 256.0f, kMagic), yo = __builtin_fmaf(x5.o, 256.0f, kMagic);
            if constexpr (START4 || U16) {
                // not fed by raw f32 frames (X4 of the label-masked stage; the uint16 payload, whose scale may be any): the values stored are
                // checked themselves.  Exact iff 256 x is an integer: then y - (OFFSET + 2^23) gives 256 x back bit for bit; the range: u -
                // CODE_MIN and CODE_MAX - u both wrap to something huge outside [CODE_MIN, CODE_MAX].
                const unsigned ue = __builtin_bit_cast(unsigned, ye) - 0x4B000000u, uo = __builtin_bit_cast(unsigned, yo) - 0x4B000000u;
                const unsigned diff = (__builtin_bit_cast(unsigned, __fsub_rn(ye, kMagic)) ^ __builtin_bit_cast(unsigned, __fmul_rn(x5.e, 256.0f))) |
                                      (__builtin_bit_cast(unsigned, __fsub_rn(yo, kMagic)) ^ __builtin_bit_cast(unsigned, __fmul_rn(x5.o, 256.0f))) |
                                      (((ue - Q16::CODE_MIN) | (uo - Q16::CODE_MIN) | (Q16::CODE_MAX - ue) | (Q16::CODE_MAX - uo)) >> 15);
                bad |= real ? diff : 0u;
            }
            const unsigned packed = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, yo), __builtin_bit_cast(unsigned, ye), 0x05040100u);
            __builtin_amdgcn_raw_buffer_store_b32(packed, ob.rs, real ? qc : kDropOffset, (inrows ? m : 0) * cols * 2, 0);
        } else {
            st2(ob, real ? oc : kDropOffset, inrows ? m : 0, cols, x5);
        }
    };
    // three loops, not one loop with a choice inside (which costs 40 registers: the allocator then has to agree on every ring slot's
    // register across two unrolled bodies): the blocks of eight steps in front of the first inner one, the inner ones, the rest
    const bool strip_inner = gx0 >= 0 && gx0 + 127 < cols;
    const int inner_lo = max(START4 ? 6 + ROFF : 6, m0 + 9), inner_hi = min(rows + ROFF - PFD, r1 + 9);   // inner blocks: i0 >= inner_lo, i0 + 7 < inner_hi (the loads reach PFD rows ahead)
    int i0 = S;
    for (; i0 < nsteps && !(strip_inner && i0 >= inner_lo && i0 + 7 < inner_hi); i0 += 8)
        static_for<0, 8>([&](auto P_) { row_step(std::false_type{}, P_, i0 + decltype(P_)::value); });
    for (; i0 < nsteps && strip_inner && i0 + 7 < inner_hi; i0 += 8)
        static_for<0, 8>([&](auto P_) { row_step(std::true_type{}, P_, i0 + decltype(P_)::value); });
    for (; i0 < nsteps; i0 += 8)
        static_for<0, 8>([&](auto P_) { row_step(std::false_type{}, P_, i0 + decltype(P_)::value); });
    if constexpr (Q16OUT && !START4 && !U16) bad |= (chk_max > 35838u ? 1u : 0u) | chk_or;
    if (tb) {
        // table mode: [f][0][col] = first valid row (rows - 1 for an empty column, whose last row gets the 100 of LO :110, :125-127),
        // [f][1][col] = last valid row (same).  With bands every band leaves its own slot [f][band][.][col] as found (INT_MAX / -1
        // for none) and the reader combines them (table_rows); the last band puts the 100 into the last row of every column IT found
        // nothing in -- a row the reader only looks at when no band found anything (otherwise it lies below the column's last valid row).
        int* tt = tb + ((size_t)f * bands + band) * 2 * cols, *bt = tt + cols;
        if (outlane) {
            const bool ee = bie < 0, eo = bio < 0;
            if ((ee | eo) && band == bands - 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (Q16OUT) {                                   // 100 = the largest code
                    if (ee) __builtin_amdgcn_raw_buffer_store_b16((short)Q16::CODE_MAX, ob.rs, qc + 2u * (unsigned)((rows - 1) * cols), 0, 0);
                    if (eo) __builtin_amdgcn_raw_buffer_store_b16((short)Q16::CODE_MAX, ob.rs, qc + 2u + 2u * (unsigned)((rows - 1) * cols), 0, 0);
                } else {
                    if (ee) ob.st_at(oc + 4u * (unsigned)((rows - 1) * cols), 100.0f);
                    if (eo) ob.st_at(oc + 4u + 4u * (unsigned)((rows - 1) * cols), 100.0f);
                }
            }
            const bool tr = bands == 1;                                  // unbanded: the translation of an empty column is done here
            *reinterpret_cast<int2*>(tt + gx) = make_int2(ee && tr ? rows - 1 : tie, eo && tr ? rows - 1 : tio);
            *reinterpret_cast<int2*>(bt + gx) = make_int2(ee && tr ? rows - 1 : bie, eo && tr ? rows - 1 : bio);
        }
        if constexpr (Q16OUT) {
            if (__builtin_amdgcn_ballot_w64(bad != 0u) != 0ull && lane == 0) { atomicOr(q16_bad, 1); *q16_seen = 1; }   // q16_seen: host memory the next call looks at
        }
        return;
    }
    if constexpr (Q16OUT) return;                                // (codes need the table: the host never launches this)
    // ---- H6 as the reference writes it (LO :122-127): rows >= last valid take its value, rows <= first valid take its
    // value; a column without valid pixels ends as 100 everywhere (:110, :125-127).  One column at a time (4-byte stores).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this lane's own stores, before it reads some of them back
    auto extend = [&](unsigned cb, int ti, int bi) {
        float tv = ob.ld_at(cb + 4u * (unsigned)(min(ti, rows - 1) * cols)), bv = ob.ld_at(cb + 4u * (unsigned)(max(bi, 0) * cols));
        if (bi < 0) { ti = rows - 1; tv = 100.0f; bi = rows; }
        if (!outlane) { ti = -1; bi = rows; }
        const int tmax = wave_max_i(ti);
        for (int r = 0; r <= tmax; ++r)
            if (r <= ti) ob.st(cb, r, cols, tv);
        const int bmin = wave_min_i(bi);
        for (int r = bmin; r < rows; ++r)
            if (r >= bi) ob.st(cb, r, cols, bv);
    };
    extend(oc, tie, bie);
    extend(oc + 4u, tio, bio);
}

}  // namespace dcmt

namespace dcmt {

// ---------------------------------------------------------------------------------
// LC variant, label-masked stage with two columns per lane (k_label_stage_s of dcmt_kernels_fused.h has the scheme:
// LC/img_completion_lc.cpp:78-102 -- for every label c: region = x * [label == c]; region = erode5(dilate5(dilate_k0(region)));
// x[label == c] = region[label == c]).  A wave64 runs the masked H2..H4 pipeline for up to four labels side by side: every
// label's bounding box, grown by the chain's reach (4 / 6 columns left / right for the as-compiled element, 6 / 6 for the
// diamond) and aligned to an even column, is a segment of lane PAIRS, so a 22-column superpixel takes 16-17 lanes instead
// of 32 and the horizontal windows cost what they cost in k_pre_p.  Lanes carry their own label, image rows and columns;
// the DPP shifts that cross a segment boundary only ever reach halo columns (outside the label's box, so never written:
// the write-back is masked by label == c, and a label's pixels all lie inside its box).
// Rows: a label's pixels lie in rows y0 .. y1, so the masked image is all zero above y0 and below y1.  If every label of the
// pass starts at least 10 rows below the image top, the stream starts AT y0 with the rings holding what six all-zero rows
// leave behind (zeros: max / min of zeros; no border value is within reach) instead of six rows above it: h + 6 row steps
// per label instead of h + 12.
// G labels per wave (host: by mean label area); a group that does not fit in 64 lanes runs in several passes, a box wider
// than one pass (116 columns) in chunks.  Any G is correct for any label plane.  Measured (1216x352, 1360 labels of ~22 columns,
// 256 frames): G = 1 / 2 / 3 / 4 -> 1.43 / 0.87 / 0.61 / 0.54 ms; giving a wave MORE labels than fit side by side (6, 8: two or
// three passes per wave, fewer waves) is slower again (0.69 / 0.62 ms): the passes are short (h + 6 row steps) and bound by
// the latency of their first rows, so many short waves beat fewer long ones.
// ---------------------------------------------------------------------------------
// INTERIOR: every row and column the pass touches lies inside the image (and the stream starts warm): no border value can
// appear, so the border selects go away and the row addresses advance by one fast add per step instead of clamp + multiply.
template <int K0KIND, bool NORM, bool INTERIOR>
__device__ __forceinline__ void label_pipeline_p(const FrameBuf& sb, const FrameBuf& lb, const FrameBuf& ob, int L, int y0, int gx, int ox0, int ox1,
                                                 bool warm, int nsteps, int rows, int cols, float max_depth, float thr, float na, float nb)
{
    constexpr float NEG = -FLT_MAX, POS = FLT_MAX;
    const F2 NEG2 = {NEG, NEG}, POS2 = {POS, POS}, Z2 = {0.f, 0.f};
    const bool incol = gx >= 0 && gx < cols;                     // gx and cols are even: both columns inside or both outside
    const bool oute = incol && gx >= ox0 && gx <= ox1, outo = incol && gx + 1 >= ox0 && gx + 1 <= ox1;   // columns this pass may write
    const unsigned gb = 4u * (unsigned)min(max(gx, 0), cols - 2);
    const unsigned rowb = 4u * (unsigned)cols;
    auto row_off = [&](int r) -> unsigned { return gb + rowb * (unsigned)min(max(r, 0), rows - 1); };
    F2 PF[8], H4[8], HE[8], XR[8], A3[8];
    float OX[8], S1E[8];
    int PLe[8], PLo[8], LBe[8], LBo[8];
    const bool w = INTERIOR || warm;
    const F2 h4i = w ? Z2 : NEG2, hei = w ? Z2 : POS2;
    const float s0i = w ? 0.f : NEG;
#pragma unroll
    for (int q = 0; q < 8; ++q) { PF[q] = Z2; H4[q] = h4i; HE[q] = hei; XR[q] = {s0i, s0i}; A3[q] = {s0i, s0i}; OX[q] = s0i; S1E[q] = s0i; PLe[q] = PLo[q] = LBe[q] = LBo[q] = -1; }
    const int rs = w ? y0 : y0 - 6;                // image row of step 0
#ifndef DCMT_LABEL_PFD
#define DCMT_LABEL_PFD 4
#endif
    constexpr int PFD = DCMT_LABEL_PFD;
    // INTERIOR: wa = byte offset of (row i - 6, this lane's columns), the row written in step i; the loads reach ahead of it
    // through the scalar offset.  (rs >= 10, so wa never goes below the frame.)
    unsigned wa = gb + rowb * (unsigned)(rs - 6);
    auto fetch = [&](int slot, int r, int ahead) {
        unsigned vo, so;
        if constexpr (INTERIOR) { vo = wa; so = rowb * (unsigned)(6 + ahead); } else { vo = row_off(r); so = 0; }
        const f2v v = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(sb.rs, vo, so, 0));
        const int2 li = __builtin_bit_cast(int2, __builtin_amdgcn_raw_buffer_load_b64(lb.rs, vo, so, 0));
        PF[slot] = {v.x, v.y}; PLe[slot] = li.x; PLo[slot] = li.y;
    };
#pragma unroll
    for (int q = 0; q < PFD; ++q) fetch(q, rs + q, q);
    int s0 = 0;
    auto step = [&](auto P_) __attribute__((always_inline)) {
            constexpr int p = decltype(P_)::value;
            const int i = rs + (s0 + p);             // image row fed by this step (per lane)
            F2 raw = PF[p];
            const int le = PLe[p], lo = PLo[p];
            fetch((p + PFD) & 7, i + PFD, PFD);
            LBe[p] = le; LBo[p] = lo;
            // masked copy (LC :94-95): the label's pixels keep their H2 value, other image pixels are 0; outside the image the dilate border value
            if constexpr (NORM) raw = {norm_apply(raw.e, na, nb), norm_apply(raw.o, na, nb)};
            F2 x2 = {le == L ? invert_valid(raw.e, max_depth, thr) : 0.0f, lo == L ? invert_valid(raw.o, max_depth, thr) : 0.0f};
            if constexpr (!INTERIOR) x2 = p_sel(incol && (unsigned)i < (unsigned)rows, x2, NEG2);
            const int j = i - 2;
            F2 y3;
            if constexpr (K0KIND == K0_AS_COMPILED) {
                const float s1e = from_right(x2.e);
                y3.e = fmax2(OX[(p + 5) & 7], s1e);
                y3.o = fmax2(from_right(x2.o), S1E[(p + 5) & 7]);
                OX[p] = x2.o;
                S1E[p] = s1e;
            } else {
                const F2 a3 = p_hmax3(x2);
                XR[p] = x2;
                A3[p] = a3;
                const F2 a5j = p_grow_max(A3[(p + 6) & 7]);
                y3 = p_max(p_max3(XR[(p + 4) & 7], A3[(p + 5) & 7], a5j), p_max(A3[(p + 7) & 7], x2));
            }
            if constexpr (!INTERIOR) y3 = p_sel(incol && (unsigned)j < (unsigned)rows, y3, NEG2);
            H4[(p + 6) & 7] = p_grow_max(p_hmax3(y3));
            const int k = i - 4;
            F2 d4 = p_max3(p_max3(H4[(p + 2) & 7], H4[(p + 3) & 7], H4[(p + 4) & 7]), H4[(p + 5) & 7], H4[(p + 6) & 7]);
            if constexpr (!INTERIOR) d4 = p_sel(incol && (unsigned)k < (unsigned)rows, d4, POS2);
            HE[(p + 4) & 7] = p_grow_min(p_hmin3(d4));
            const int l = i - 6;
            const F2 e4 = p_min3(p_min3(HE[(p + 0) & 7], HE[(p + 1) & 7], HE[(p + 2) & 7]), HE[(p + 3) & 7], HE[(p + 4) & 7]);
            // write-back only where the label is this one (LC :101): a 4-byte store per column, aimed past the buffer where it is not
            // (a row l outside the image is never written: the clamped loads' labels are not consulted for it)
            if constexpr (INTERIOR) {
                ob.st_at((oute && LBe[(p + 2) & 7] == L) ? wa : kDropOffset, e4.e);
                ob.st_at((outo && LBo[(p + 2) & 7] == L) ? wa + 4u : kDropOffset, e4.o);
                wa += rowb;
            } else {
                const bool inl = (unsigned)l < (unsigned)rows;
                const unsigned wo = gb + rowb * (unsigned)l;
                ob.st_at((inl && oute && LBe[(p + 2) & 7] == L) ? wo : kDropOffset, e4.e);
                ob.st_at((inl && outo && LBo[(p + 2) & 7] == L) ? wo + 4u : kDropOffset, e4.o);
            }
    };
    // (a pass is h + 6 steps, ~28: a way out in the middle of the eight-step block saves two of the ~3.5 steps a pass runs past its end)
    for (; s0 < nsteps; s0 += 8) {
        static_for<0, 2>(step);
        if (s0 + 2 >= nsteps) break;
        static_for<2, 4>(step);
        if (s0 + 4 >= nsteps) break;
        static_for<4, 6>(step);
        if (s0 + 6 >= nsteps) break;
        static_for<6, 8>(step);
    }
}

constexpr int kLabelGroupMax = 4;     // labels one wave may be given (it runs them in passes of as many as fit side by side)
#ifndef DCMT_LABEL_WAVES
#define DCMT_LABEL_WAVES 0
#endif
template <int K0KIND, bool NORM>
__global__ __launch_bounds__(256)
#if DCMT_LABEL_WAVES
__attribute__((amdgpu_waves_per_eu(DCMT_LABEL_WAVES, DCMT_LABEL_WAVES)))
#endif
void k_label_stage_p(const float* __restrict__ src, const int32_t* __restrict__ labels, int n_labels, int G,
                     int* __restrict__ bb_min, int* __restrict__ bb_max, float* __restrict__ x4,
                     int rows, int cols, float max_depth, float thr, const float* __restrict__ coef)
{
    constexpr int HL = K0KIND == K0_AS_COMPILED ? 4 : 6, HR = 6, H = 6;
    constexpr int VW = (128 - HL - HR - 1) & ~1;         // output columns of one chunk of a box wider than a wave
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int first = wv * G, last = min(first + G, n_labels);      // this wave's labels
    if (first >= n_labels) return;
    const int f = blockIdx.y;
    const size_t bo = (size_t)f * n_labels;
    const size_t fo = (size_t)f * rows * cols, fe = (size_t)rows * cols;
    float na = 1.0f, nb = 0.0f;
    if constexpr (NORM) { na = coef[2 * f]; nb = coef[2 * f + 1]; }
    FrameBuf sb, lb, ob;
    sb.init(src + fo, fe);
    lb.init(reinterpret_cast<const float*>(labels + fo), fe);
    ob.init(x4 + fo, fe);
    // the boxes of this wave's (at most kLabelGroupMax) labels, fetched together: one round trip to memory, not one per label
    int by0[kLabelGroupMax], bx0[kLabelGroupMax], by1[kLabelGroupMax], bx1[kLabelGroupMax];
#pragma unroll
    for (int q = 0; q < kLabelGroupMax; ++q) {
        const int c = min(first + q, n_labels - 1);
        const int2 mn = *reinterpret_cast<const int2*>(bb_min + (bo + c) * 2), mx = *reinterpret_cast<const int2*>(bb_max + (bo + c) * 2);
        by0[q] = mn.x; bx0[q] = mn.y; by1[q] = first + q < last ? mx.x : -1; bx1[q] = mx.y;
    }
    int k = 0;                                       // next label of the wave (index into the arrays above); G <= kLabelGroupMax
    int cx = 0, cj = -1;                             // a box being walked in chunks: next chunk's first column, its label index
    const int nl = last - first;
    while (k < nl || cj >= 0) {
        int L = -2, y0 = 0, gx = -2, ox0 = 0, ox1 = -1;  // lanes without a segment: no label, outside the image
        int hmax = 0, ymin = 0x7fffffff, ymax = -1, xlo = 0x7fffffff, xhi = -1;   // of the pass: tallest box, first / last starting row, first / last column touched
        if (cj >= 0) {
            const int sx = (cx - HL) & ~1;
            L = first + cj; y0 = by0[cj]; gx = sx + 2 * lane; ox0 = cx; ox1 = min(cx + VW - 1, bx1[cj]);
            hmax = by1[cj] - by0[cj] + 1; ymin = ymax = by0[cj]; xlo = sx; xhi = sx + 127;
            cx += VW;
            if (cx > bx1[cj]) cj = -1;
        } else {
            // pack labels k, k+1, ... side by side, in order, while their segments fit in 64 lanes
            int used = 0, nseg = 0;
            bool full = false;
#pragma unroll
            for (int q = 0; q < kLabelGroupMax; ++q) {
                if (full || q < k || q >= nl) continue;
                if (by1[q] >= 0) {                                   // (a label may own no pixel: skipped)
                    const int sx = (bx0[q] - HL) & ~1;
                    const int need = ((bx1[q] + HR - sx) >> 1) + 1;
                    if (used + need > 64) { full = true; continue; }
                    if (lane >= used && lane < used + need) { L = first + q; y0 = by0[q]; gx = sx + 2 * (lane - used); ox0 = bx0[q]; ox1 = bx1[q]; }
                    used += need; ++nseg;
                    hmax = max(hmax, by1[q] - by0[q] + 1); ymin = min(ymin, by0[q]); ymax = max(ymax, by0[q]);
                    xlo = min(xlo, sx); xhi = max(xhi, sx + 2 * need - 1);
                }
                k = q + 1;
            }
            if (nseg == 0) {
                if (k >= nl) break;                                   // only empty labels were left
                cj = k; cx = bx0[k]; ++k;                             // label k alone is wider than a wave: walk its box in chunks
                continue;
            }
        }
        const bool warm = ymin >= 10;
        const int nsteps = hmax + (warm ? H : 2 * H);
        // interior pass: warm, every column touched inside the image, every row fed or prefetched inside it
        const bool interior = warm && xlo >= 0 && xhi < cols && ymax + nsteps + 7 + DCMT_LABEL_PFD < rows;   // (the step loop runs in eights, the loads DCMT_LABEL_PFD rows ahead)
        if (interior) label_pipeline_p<K0KIND, NORM, true>(sb, lb, ob, L, y0, gx, ox0, ox1, true, nsteps, rows, cols, max_depth, thr, na, nb);
        else label_pipeline_p<K0KIND, NORM, false>(sb, lb, ob, L, y0, gx, ox0, ox1, warm, nsteps, rows, cols, max_depth, thr, na, nb);
    }
    // every label's box is read by exactly this one wave: it leaves the tables as k_label_bbox expects to find them (no box), so that no
    // memset has to run in the stream in front of the next call
    if (lane < nl) {
        int2* mn = reinterpret_cast<int2*>(bb_min + (bo + first + lane) * 2);
        int2* mx = reinterpret_cast<int2*>(bb_max + (bo + first + lane) * 2);
        *mn = make_int2(kBboxNone, kBboxNone); *mx = make_int2(-1, -1);
    }
}

}  // namespace dcmt
