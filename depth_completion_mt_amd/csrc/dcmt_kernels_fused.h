// dcmt_kernels_fused.h -- the fast path for the default configuration (a preset first
// element, images of at least 8x8): two launches per batch (k_pre_s, k_fp_s), the hot
// stencils held entirely in registers, no barriers, LDS only as wave-private delay lines and
// row rings; rows addressed through buffer resources (FrameBuf).
//
//   k_pre_s   H2..H6   one wave64 owns a strip of columns over the FULL image height and
//                      streams down the rows: lane = column, vertical windows live in
//                      rolling registers, horizontal windows come from DPP wave shifts
//                      (v_max_f32_dpp ... wave_shr:1 / wave_shl:1 -- no LDS, no barriers).
//                      Because the wave sees whole columns, the column extension (H6) needs
//                      no second kernel and no atomics: first/last valid row are tracked in
//                      registers and the extension zones are written in the epilogue.
//   k_fp_s    H7..H11  the same streaming layout: 31x31 dilate-fill (vertical doubling in
//                      registers, horizontal 16-lane DPP prefix/suffix scans + two ds_bpermutes),
//                      feeding the 5x5 median (time-shared sorting networks), the separable
//                      Gaussian, the masked select and the final invert in the same step.
//   k_fill_s / k_post_s  the two halves of k_fp_s as separate kernels: the hole-closure loop
//                      (H8) for the rare frames that need it, and the stop_after probes.
//
// Row indices in comments: i = input row of the current step; stage outputs lag behind.
#pragma once

#include <type_traits>

#include "dcmt_kernels_v1.h"

namespace dcmt {

// value held by the lane to the left / right (column c-1 / c+1); lanes without a source
// get 0 -- those are halo lanes whose results are never used.
__device__ __forceinline__ float from_left(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}

// horizontal windows over lanes: 3 = [c-1,c+1], 5 = [c-2,c+2], 7 = [c-3,c+3]
__device__ __forceinline__ float hmax3(float v) { const float a = fmax2(from_left(v), v); return fmax2(from_right(a), a); }
__device__ __forceinline__ float hmin3(float v) { const float a = fmin2(from_left(v), v); return fmin2(from_right(a), a); }
__device__ __forceinline__ float hgrow_max(float w) { const float l = from_left(w); return fmax2(from_right(w), l); }
__device__ __forceinline__ float hgrow_min(float w) { const float l = from_left(w); return fmin2(from_right(w), l); }

// One frame as a buffer resource: rows are addressed as (per-lane 32-bit byte offset in a VGPR) + (wave-uniform row
// byte offset in an SGPR) -- buffer_load_dword v, v_off, s[rsrc], s_row offen -- so a row access costs no VALU
// address arithmetic at all (a flat global access spends one 64-bit per-lane add per row) and the lane offsets need
// one VGPR each instead of a 64-bit pointer pair.  Raw buffer, stride 0, num_records = the frame's bytes.
// A store every lane issues on every step but that must write nothing (halo lanes, steps without an output row) is aimed
// at kDropOffset.  Rule relied on (CDNA3/4 ISA, buffer instructions, "range checking"): for a raw buffer (stride 0, no
// swizzle) an access is out of range when voffset + inst_offset + access size > num_records -- the SGPR offset is NOT part
// of the check -- and an out-of-range store is dropped (a load returns 0).  dcmt_create admits frames of at most
// 0x1ffffff0 pixels, so num_records = frame bytes <= 0x7fffffc0 < kDropOffset for every frame this library accepts.
constexpr unsigned kDropOffset = 0x7ffffff0u;
struct FrameBuf {
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ void init(const float* frame, size_t elems)
    {
        const size_t bytes = elems * sizeof(float);
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(frame), 0, bytes > 0x7fffffffu ? 0x7fffffff : (int)bytes, 0x00020000);
    }
    __device__ __forceinline__ float ld(unsigned lane_bytes, int row, int cols) const
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_bytes, row * cols * 4, 0));
    }
    __device__ __forceinline__ void st(unsigned lane_bytes, int row, int cols, float v) const
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, lane_bytes, row * cols * 4, 0);
    }
    // per-lane rows: the whole byte offset in the VGPR
    __device__ __forceinline__ float ld_at(unsigned bytes) const
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, bytes, 0, 0));
    }
    __device__ __forceinline__ void st_at(unsigned bytes, float v) const
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, bytes, 0, 0);
    }
};


// 16-bit codes of depths that are multiples of 1/256 m (k_pre_p<Q16OUT> -> k_fp_h / k_fp_q, dcmt_kernels_fp_q16.h has the argument).
// code = 256 x + 6143 for x = j / 256, j in [-5119, 25600] (an empty pixel, an inverted depth down to 100 - 119.996 m, the 100 of an
// empty column): 0x0400 .. 0x7bff -- the bit patterns of the positive NORMAL half-floats, which order exactly like the integers they
// are.  So the codes can be compared as unsigned integers (v_pk_max_u16 ...) AND as f16 (v_pk_maximum3_f16 / v_pk_minimum3_f16:
// gfx950's only packed three-input min / max; tools/pk3_probe.hip checks the ordering on the device, subnormals included).
// A depth beyond 119.996 m has no code: the check in k_pre_p raises the flag and the f32 kernels rerun.
struct Q16 {
    static constexpr int OFFSET = 6143;
    static constexpr unsigned CODE_MIN = 0x0400u, CODE_MAX = 0x7bffu;    // = 25600 + OFFSET: the 100 of LO :110
    static constexpr unsigned HOLE_MAX = 25 + OFFSET;            // x < 0.1f  <=>  256 x <= 25  <=>  code <= HOLE_MAX
    static constexpr unsigned HOLE_MAX_HI = (HOLE_MAX << 16) | 0xffffu;   // the same test on the high half of a packed pair
    __device__ static __forceinline__ unsigned code(float x) { return (unsigned)((int)__fmul_rn(x, 256.0f) + OFFSET); }
    __device__ static __forceinline__ float value(unsigned c) { return __builtin_fmaf((float)c, 0.00390625f, -23.99609375f); }   // (c - 6143) / 256, exact
    __host__ __device__ static bool params_ok(float max_depth, float thr) { return max_depth == 100.0f && thr == 0.1f; }
};

__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}


// Workgroup -> (frame, unit) mapping.  Workgroups are dealt round-robin over the 8 XCDs
// (blocks b and b+8 share an XCD and its 4 MiB L2), so with xcd_map every XCD works through
// whole frames: the units (strips / tiles) of one 1.7 MB frame run side by side on one XCD
// and their overlapping halo reads hit that XCD's L2 instead of going out to HBM once per
// XCD.  Placement only affects speed, never results.  units = units per frame.
__device__ __forceinline__ void frame_unit(int b, int units, int batch, int xcd_map, int& f, int& u)
{
    if (xcd_map) {
        const int xcd = b & 7, slot = b >> 3;
        f = (slot / units) * 8 + xcd;
        u = slot % units;
    } else {
        f = b / units;
        u = b - f * units;
    }
    (void)batch;
}

// Wave -> (frame, strip) for the kernels that give every wave one strip: the pairs are dealt to waves in one flat
// sequence, so a strip count that is not a multiple of 4 (22 post strips at 1216 columns) leaves no workgroup with idle
// waves holding its LDS; with xcd_map the sequence runs per XCD over that XCD's frames.  Returns false for the waves past
// the end.  Grid: wave_grid().
__device__ __forceinline__ bool wave_strip(int b, int wave, int strips, int batch, int xcd_map, int& f, int& strip)
{
    if (xcd_map) {
        const int g = (b >> 3) * 4 + wave;
        if (g >= (batch >> 3) * strips) return false;
        f = (g / strips) * 8 + (b & 7);
        strip = g % strips;
    } else {
        const int g = b * 4 + wave;
        if (g >= batch * strips) return false;
        f = g / strips;
        strip = g % strips;
    }
    return true;
}

// The hole counters of a frame (k_fp_s / k_fill_s add to them later in the stream) start every call at zero: the first
// kernel of the chain clears them, one wave per frame -- a memset in front would be one more dependent operation in the
// stream (2 us + a 6 us gap, measured).  counters == nullptr: the caller has cleared them.
__device__ __forceinline__ void clear_frame_counters(int* counters, int f, bool first_wave_of_frame, int lane)
{
    if (counters && first_wave_of_frame)
        for (int i = lane; i < kCntStride; i += 64) counters[(size_t)f * kCntStride + i] = 0;
}

// (first, last) valid row of column c of frame f from the table k_pre_s / k_pre_p leave in table mode.  A launch with row
// bands leaves one slot per band -- [f][band][0][col] = first valid row of the band's rows or INT_MAX, [f][band][1][col] =
// last valid row or -1 -- and the reader combines them; a column no band found a valid row in is 100 everywhere
// (LO :110, :125-127): both rows point at the last row, where the last band has put the 100.  Unbanded launches have
// done that translation themselves.
__device__ __forceinline__ void table_rows(const int* tb, int f, int cols, int bands, int rows, int c, int& first, int& last)
{
    const int* tt = tb + (size_t)f * bands * 2 * cols + c;
    first = tt[0]; last = tt[cols];
    if (bands > 1) {
        for (int b = 1; b < bands; ++b) { first = min(first, tt[(size_t)b * 2 * cols]); last = max(last, tt[(size_t)b * 2 * cols + cols]); }
        if (last < 0) first = last = rows - 1;
    }
}

// ---------------------------------------------------------------------------------
// RowRing: wave-private LDS ring filled by LDS-DMA.  One global_load_lds_dwordx4 moves a 4-row
// x 64-column block (64 lanes x 16 B, the widest access) for the wave's strip straight into
// LDS -- no VGPRs, asynchronous -- and every row is then one conflict-free ds_read_b32 per
// lane.  Measured on MI355X (tools/bw_probe.hip): the one-dword-per-lane row loads of a strip
// kernel stream at 4.0 TB/s, this form at 4.7 TB/s = the plain float4-copy rate of the box.
// Requirements (checked by the dispatcher): cols % 4 == 0 and the strip origin gx0 % 4 == 0,
// so every lane's 16-byte source is aligned and never straddles the image edge.
// Stream row s (s = 0, 1, ...) is image row clamp(s - ROW_OFF): replicated rows for free.
// Column groups outside the image are clamped as a group: they deliver in-image substitutes
// from the 4 edge columns (fine for masked or max-filtered halos; k_post_s patches its lanes).
// The ring belongs to ONE wave: its own counted vmcnt is all the ordering the reads need.
// ---------------------------------------------------------------------------------
template <int SLOTS, int ROW_OFF>
struct RowRing {
    float* ring;            // SLOTS x 256 floats, this wave's
    const float* src;       // frame base + clamped group column of this lane
    int rows, cols, lrow;

    __device__ __forceinline__ void init(float* wave_ring, const float* frame, int rows_, int cols_, int gx0, int lane)
    {
        ring = wave_ring; rows = rows_; cols = cols_;
        lrow = lane >> 4;
        src = frame + min(max(gx0 + 4 * (lane & 15), 0), cols_ - 4);
    }
    // stream rows 4*chunk .. 4*chunk+3  ->  ring slot chunk % SLOTS
    __device__ __forceinline__ void issue(int chunk) const
    {
        const int r = min(max(4 * chunk + lrow - ROW_OFF, 0), rows - 1);
        const float* g = src + (size_t)r * cols;
        float* l = ring + (chunk % SLOTS) * 256;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
    // all but the `keep` youngest DMAs of this wave have landed (stores issued in between only make
    // the wait stricter: vmcnt counts loads and stores together, in order)
    template <int KEEP>
    __device__ __forceinline__ void wait() const { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory"); }
    __device__ __forceinline__ float read(int s, int lane) const { return ring[((s >> 2) % SLOTS) * 256 + (s & 3) * 64 + lane]; }
};

// uint16 payload -> metres, for the paths that do not fuse the ingest (staged kernels)
__global__ void k_u16_to_f32(const uint16_t* __restrict__ in, float* __restrict__ out, size_t n, float scale)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = __fmul_rn((float)in[i], scale);
}

// ---------------------------------------------------------------------------------
// k_pre_s
// ---------------------------------------------------------------------------------
enum { K0_AS_COMPILED = 0, K0_DIAMOND = 1 };

template <int K0KIND, bool WIDE = false>
struct PreS {
    // lanes lost to the left / right of a strip: the chain's horizontal reach.
    // as-compiled element: taps (dy,dx) = (-1,+1),(+2,+2): reach 0 left, 2 right; diamond: 2 / 2.
    // WIDE (LDS-DMA rows): the strip origin must be a multiple of 4 columns: HL rounded up to 8/12, VW to 48/44.
    static constexpr int HL0 = (K0KIND == K0_AS_COMPILED ? 0 : 2) + 2 + 2 + 3;
    static constexpr int HL = WIDE ? (HL0 + 3) / 4 * 4 : HL0;
    static constexpr int HR = 2 + 2 + 2 + 3;
    static constexpr int VW = WIDE ? (64 - HL - HR) / 4 * 4 : 64 - HL - HR;   // output columns per wave
    static constexpr int LAT = 9;                // rows between the input row and the finished X5 row
};

// START4: the input is already X4 (the label-masked stage of the LC variant produced it): only H5 and H6
// run; the row fed at step i is image row i - 6, so that all the ring slots below keep their meaning.
// U16: the input is the KITTI uint16 depth PNG payload; metres = value * in_scale (the reference's ingest,
// LO/main.cpp:75-82: imread(IMREAD_ANYDEPTH) + convertTo(CV_32F, 1./256)) is fused into the load.
// NORM: N1's per-frame min-max normalisation (coef[2f], coef[2f+1] from k_norm_coef) applied to every
// raw value before H2 -- cv::normalize in front of the path, SL/main_sl.cpp:370 / :523.
template <int K0KIND, bool WIDE, bool START4 = false, bool U16 = false, bool NORM = false>
__global__ __launch_bounds__(256)
void k_pre_s(const void* __restrict__ src_, float* __restrict__ x6, int rows, int cols, int strips,
             int batch, int xcd_map, float max_depth, float thr, float in_scale, const float* __restrict__ coef,
             int* __restrict__ tb, int* __restrict__ counters)
{
    static_assert(!(U16 && (WIDE || START4)), "the uint16 ingest uses the plain row loads");
    static_assert(!(NORM && (U16 || START4)), "normalisation applies to raw f32 frames");
    const float* src = static_cast<const float*>(src_);
    using G = PreS<K0KIND, WIDE>;
    constexpr int ROFF = START4 ? 6 : 0;         // image row of stream row s is s - ROFF
#ifndef DCMT_PRE_SLOTS
#define DCMT_PRE_SLOTS 8
#endif
    constexpr int SLOTS = DCMT_PRE_SLOTS, AHEAD = SLOTS - 1;      // ring slots per wave; 4-row blocks in flight
    __shared__ __attribute__((aligned(16))) float s_ring[WIDE ? 4 * SLOTS * 256 : 4];   // 4 waves x SLOTS x (4 rows x 64 columns)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);             // wave-uniform: keep it scalar
    int f, strip;
    if (!wave_strip(blockIdx.x, wave, strips, batch, xcd_map, f, strip)) return;   // whole waves leave; no barrier is used below
    clear_frame_counters(counters, f, strip == 0, lane);
    const int gx = strip * G::VW - G::HL + lane;
    const bool incol = gx >= 0 && gx < cols;
    const bool outlane = incol && lane >= G::HL && lane < G::HL + G::VW;
    const size_t fo = (size_t)f * rows * cols;
    const int gxc = min(max(gx, 0), cols - 1);       // loads are unconditional, from clamped addresses
    float na = 1.0f, nb = 0.0f;
    if constexpr (NORM) { na = coef[2 * f]; nb = coef[2 * f + 1]; }
    const uint16_t* sp16 = static_cast<const uint16_t*>(src_) + fo + gxc;
    FrameBuf ob, ib;                              // the output / input frame; this lane's column at byte offset oc
    ob.init(x6 + fo, (size_t)rows * cols);
    ib.init(src + fo, (size_t)rows * cols);
    const unsigned oc = 4u * (unsigned)gxc;
    auto load_row = [&](int r) -> float {          // image row r (already clamped) of this lane's column
        if constexpr (U16) return __fmul_rn((float)sp16[(size_t)r * cols], in_scale);
        else return ib.ld(oc, r, cols);
    };
    RowRing<SLOTS, ROFF> rr;
    if constexpr (WIDE) rr.init(s_ring + wave * SLOTS * 256, src + fo, rows, cols, strip * G::VW - G::HL, lane);

    constexpr float NEG = -FLT_MAX, POS = FLT_MAX;
    // rolling rows, indexed by (row & 7); fully unrolled below so every index is static
    float PF[8];                                 // prefetched input rows
    float XR[8];                                 // diamond only: x2 rows
    float A3[8];                                 // diamond only: horizontal 3-max rows
    float S1[8];                                 // as-compiled only: x2 shifted by one column
    float H4[8], HE[8], H7[8], E4[8], T7[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { XR[q] = NEG; A3[q] = NEG; S1[q] = NEG; H4[q] = NEG; HE[q] = POS; H7[q] = NEG; E4[q] = NEG; T7[q] = NEG; PF[q] = 0.f; }

#ifndef DCMT_PRE_PFD
#define DCMT_PRE_PFD 4
#endif
    constexpr int PFD = DCMT_PRE_PFD;            // rows of load lookahead (dword path)

    // ---- leading empty rows (the upper third of a velodyne frame): nothing to compute there.  x5(m, c) can only be valid
    // (>= thr) if some input pixel within the chain's reach -- rows m-9 .. m+9 -- is valid: maxima and minima of values
    // below thr (the -FLT_MAX dilate border included; the erode's +FLT_MAX border never wins against an in-image row) stay
    // below thr.  So with zv = the first row in which any of this wave's columns holds a valid value, every x5 row above
    // zv - 9 is invalid: it is neither stored (rows above the column's first valid row belong to H6) nor counted.  The stream
    // starts at row S <= zv - 18 with cold rings; x5 rows from S + 9 on see only rings filled from real rows (reach 9), the
    // rows before are discarded (m0).  S is a multiple of 8 so that the ring phases stay compile-time.
    int S = 0;
    if constexpr (!START4) {
        auto valid16 = [&](const float (&v)[16]) -> bool {
            float m = fmax3(fmax3(fmax3(v[0], v[1], v[2]), v[3], v[4]), v[5], v[6]);
            m = fmax3(fmax3(fmax3(m, v[7], v[8]), v[9], v[10]), v[11], v[12]);
            m = fmax2(fmax3(m, v[13], v[14]), v[15]);
            return __builtin_amdgcn_ballot_w64(m >= thr) != 0ull;      // norm_apply and the uint16 scale are applied by load16
        };
        auto load16 = [&](float (&v)[16], int z) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float x = load_row(min(z + q, rows - 1));
                if constexpr (NORM) x = norm_apply(x, na, nb);
                v[q] = x;
            }
        };
        float va[16], vb[16];
        int zv = rows;
        load16(va, 0);
        for (int z = 0; z < rows; z += 32) {       // two 16-row chunks per trip, the next one in flight while this one is looked at
            load16(vb, z + 16);
            if (valid16(va)) { zv = z; break; }
            if (z + 16 >= rows) break;
            load16(va, z + 32);
            if (valid16(vb)) { zv = z + 16; break; }
        }
        S = max(zv - 18, 0) & ~7;
    }
    const int m0 = S > 0 ? S + 9 : 0;            // first x5 row this wave accounts for

    if constexpr (WIDE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the scan's loads
#pragma unroll
        for (int q = 0; q < AHEAD; ++q) rr.issue((S >> 2) + q);   // AHEAD 4-row blocks ahead
        rr.template wait<0>();                   // from here on the counted waits below see a fixed pattern of younger operations
    } else {
#pragma unroll
        for (int q = 0; q < PFD; ++q) PF[q] = load_row(min(max(S + q - ROFF, 0), rows - 1));
    }

    int ti = 0x7fffffff, bi = -1;                // first / last valid row of X5 in this lane's column

    const int nsteps = rows + G::LAT;
    for (int i0 = S; i0 < nsteps; i0 += 8) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int i = i0 + p;
            // ---- H2 on load (LO :55-67); outside the image: the dilate border value
            float raw;
            if constexpr (WIDE) {
                // gfx9 counts loads and stores in ONE counter, in issue order.  Every step issues exactly one store (below), so
                // behind the block needed now there are always AHEAD younger DMAs and 4 * AHEAD stores: that count waits for exactly that block
                // and leaves the prefetch and the stores in flight (vmcnt(3) would also wait for all but two of the stores).
                if ((p & 3) == 0) { rr.issue((i >> 2) + AHEAD); rr.template wait<5 * AHEAD>(); }
                raw = rr.read(i, lane);
            } else {
                raw = PF[p];
                PF[(p + PFD) & 7] = load_row(min(max(i + PFD - ROFF, 0), rows - 1));
            }
            float e4;
            const int l = i - 6;
            if constexpr (START4) {
                e4 = raw;                                               // X4 row l
            } else {
            if constexpr (NORM) raw = norm_apply(raw, na, nb);
            const float x2 = (incol && i < rows) ? invert_valid(raw, max_depth, thr) : NEG;
            // ---- H3 (LO :71-80), row j = i - 2
            const int j = i - 2;
            float y3;
            if constexpr (K0KIND == K0_AS_COMPILED) {
                // dst(r,c) = max(src(r-1,c+1), src(r+2,c+2))
                const float s1 = from_right(x2);
                const float s2 = from_right(s1);
                S1[p] = s1;
                y3 = fmax2(S1[(p + 5) & 7] /* row i-3 = j-1 */, s2 /* row i = j+2 */);
            } else {
                // 13-tap diamond: rows j-2 and j+2 centre only, j-1 and j+1 three wide, j five wide
                const float a3 = hmax3(x2);
                XR[p] = x2;
                A3[p] = a3;
                const float a3j = A3[(p + 6) & 7];                       // row j
                const float a5j = hgrow_max(a3j);
                y3 = fmax2(fmax3(XR[(p + 4) & 7] /* j-2 */, A3[(p + 5) & 7] /* j-1 */, a5j),
                           fmax2(A3[(p + 7) & 7] /* j+1 */, x2 /* j+2 */));
            }
            y3 = (incol && (unsigned)j < (unsigned)rows) ? y3 : NEG;
            // ---- H4 dilate 5x5 (LO :85): horizontal on row j, vertical gives row k = j - 2
            H4[(p + 6) & 7] = hgrow_max(hmax3(y3));                      // slot of row j = i-2
            const int k = i - 4;
            float d4 = fmax3(fmax3(H4[(p + 2) & 7], H4[(p + 3) & 7], H4[(p + 4) & 7]), H4[(p + 5) & 7], H4[(p + 6) & 7]);
            d4 = (incol && (unsigned)k < (unsigned)rows) ? d4 : POS;     // border value of the erode
            // ---- H4 erode 5x5: row l = k - 2
            HE[(p + 4) & 7] = hgrow_min(hmin3(d4));                      // slot of row k = i-4
            e4 = fmin3(fmin3(HE[(p + 0) & 7], HE[(p + 1) & 7], HE[(p + 2) & 7]), HE[(p + 3) & 7], HE[(p + 4) & 7]);
            }
            e4 = (incol && (unsigned)l < (unsigned)rows) ? e4 : NEG;     // border value of the 7x7 dilate
            E4[(p + 2) & 7] = e4;                                        // slot of row l = i-6
            // ---- H5 (LO :88-100): dilate 7x7, row m = l - 3, then fill where x < 0.1
            H7[(p + 2) & 7] = hgrow_max(hgrow_max(hmax3(e4)));
            const int m = i - 9;
            // vertical 7-max over rows m-3 .. m+3 = i-12 .. i-6 in two three-input instructions: the maximum of the three
            // newest rows is kept per step, so the window is (this step's triple, the triple of three steps ago, row i-12)
            T7[p] = fmax3(H7[(p + 0) & 7], H7[(p + 1) & 7], H7[(p + 2) & 7]);      // rows i-8 .. i-6
            const float d7 = fmax3(T7[p], T7[(p + 5) & 7] /* rows i-11 .. i-9 */, H7[(p + 4) & 7] /* row i-12 */);
            const float e = E4[(p + 7) & 7];                             // row m = i-9
            const float x5 = e < thr ? d7 : e;
            const bool inrows = m >= m0 && m < rows;
            if (inrows) {
                // ---- H6 bookkeeping (LO :112-121): first / last row with x > 0.1 (their values are re-read
                // from the rows stored below, in the epilogue)
                const bool valid = x5 >= thr;
                ti = min(ti, valid ? m : 0x7fffffff);
                bi = valid ? m : bi;
            }
            // rows above the first valid one are written by the epilogue.  The store instruction itself is issued on every step
            // by every lane (the DMA waits above count on it): lanes and steps with nothing to write aim past the end of the
            // buffer resource, where the hardware drops the write
            ob.st((inrows && outlane && m >= ti) ? oc : kDropOffset, inrows ? m : 0, cols, x5);
        }
    }
    // ---- H6 (LO :122-127): rows >= last valid take its value, rows <= first valid take its
    // value; a column without valid pixels ends as 100 everywhere (:110, :125-127)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this lane's own stores, before it reads two of them back
    if (tb) {
        // table mode (k_fp_s reads X6): the rows above a column's first valid row ti all equal row ti and the rows below its last
        // valid row bi all equal row bi, so neither zone is written -- the reader clamps its row index into [ti, bi] instead
        // (a third of X6 on velodyne-like frames stays out of HBM, both ways).  Table: [f][0][col] = ti, [f][1][col] = bi.  An
        // empty column is 100 everywhere (LO :110, :125-127): both entries point at the last row, which gets the 100.
        if (bi < 0) { ti = bi = rows - 1; ob.st(outlane ? oc : kDropOffset, rows - 1, cols, 100.0f); }
        if (outlane) { tb[(size_t)f * 2 * cols + gx] = ti; tb[(size_t)f * 2 * cols + cols + gx] = bi; }
        return;
    }
    const float bv = ob.ld_at(oc + 4u * (unsigned)(max(bi, 0) * cols));
    {
        float tv = ob.ld_at(oc + 4u * (unsigned)(min(ti, rows - 1) * cols));
        if (bi < 0) { ti = rows - 1; tv = 100.0f; bi = rows; }
        if (!outlane) { ti = -1; bi = rows; }
        const int tmax = wave_max_i(ti);
        for (int r = 0; r <= tmax; ++r)
            if (r <= ti) ob.st(oc, r, cols, tv);
    }
    const int bmin = wave_min_i(bi);
    for (int r = bmin; r < rows; ++r)
        if (r >= bi) ob.st(oc, r, cols, bv);
}

// ---------------------------------------------------------------------------------
// LC variant, fast path (LC/img_completion_lc.cpp:78-102): for every label c
//     region = x * [label == c];  region = erode5(dilate5(dilate_k0(region)));  x[label == c] = region[label == c]
// Labels are disjoint and each write-back depends only on the pre-loop values of its own label.
//   k_label_bbox     one pass over the label plane: bounding box of every label (atomics only where a
//                    run of equal labels starts / ends), and X4 = H2 for the pixels no label claims;
//   k_label_stage_s  one wave64 per label pair (or per label): the k_pre_s pipeline (H2, H3, H4) on the masked
//                    image over the label's bounding box grown by the chain's reach, lane = column, rows
//                    streamed; writes X4 where label == c.  Two boxes that fit side by side share the wave;
//                    boxes wider than 64 - reach columns are walked in chunks.
//   k_pre_s<START4>  H5 + H6 on X4, then k_fp_s as for img_completion.
// ---------------------------------------------------------------------------------
// LDS_TABLE: the workgroup (64 columns x kBboxRows rows) first reduces into a per-label table in LDS (ds_min /
// ds_max, no global traffic) and then issues global atomics only for the handful of labels it touched;
// without it (label tables too big for LDS) every run start / end goes to global memory directly.
constexpr int kBboxNone = 0x7f7f7f7f;        // bb_min entry of a label without pixels (what a memset with 0x7f leaves; bb_max: -1)
constexpr int kBboxRows = 128;               // rows per workgroup (32 per wave, in groups of 8): the table init / flush is paid once per 128 rows
template <bool LDS_TABLE>
__global__ __launch_bounds__(256)
void k_label_bbox(const float* __restrict__ src, const int32_t* __restrict__ labels, int n_labels,
                  int* __restrict__ bb_min, int* __restrict__ bb_max, float* __restrict__ x4,
                  int rows, int cols, float max_depth, float thr, const float* __restrict__ coef)
{
    extern __shared__ __attribute__((aligned(16))) int s_bb[];     // LDS_TABLE: [n_labels][4] = ymin, xmin, ymax, xmax
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.z;
    const int gx = blockIdx.x * 64 + lane;
    const size_t fo = (size_t)f * rows * cols;
    int* mn = bb_min + (size_t)f * n_labels * 2;       // [label][0] = ymin, [1] = xmin   (init 0x7f7f7f7f)
    int* mx = bb_max + (size_t)f * n_labels * 2;       // [label][0] = ymax, [1] = xmax   (init -1)
    if constexpr (LDS_TABLE) {
        for (int i = threadIdx.x; i < n_labels; i += 256) {
            s_bb[4 * i] = 0x7fffffff; s_bb[4 * i + 1] = 0x7fffffff; s_bb[4 * i + 2] = -1; s_bb[4 * i + 3] = -1;
        }
        __syncthreads();
    }
    const bool in = gx < cols;
    for (int g = 0; g < kBboxRows / 32; ++g) {
    const int gy0 = blockIdx.y * kBboxRows + (g * 4 + wave) * 8;
    if (gy0 >= rows) break;
    int lrow[8];                                   // all eight label rows in flight before the first is looked at
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int gy = gy0 + r;
        lrow[r] = (in && gy < rows) ? labels[fo + (size_t)gy * cols + gx] : -1;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int gy = gy0 + r;
        if (gy >= rows) break;
        int l = lrow[r];
        const bool lv = in && l >= 0 && l < n_labels;
        if (!lv) l = -1;
        const int left = __builtin_amdgcn_update_dpp(-2, l, 0x138, 0xf, 0xf, false);   // lane 0 keeps -2: always a run start
        const int right = __builtin_amdgcn_update_dpp(-2, l, 0x130, 0xf, 0xf, false);
        if constexpr (LDS_TABLE) {
            // a run that continues the same label straight above / below cannot move that label's top / bottom row
            const bool top = r == 0 || lrow[r > 0 ? r - 1 : 0] != l, bot = r == 7 || gy + 1 >= rows || lrow[r < 7 ? r + 1 : 7] != l;
            if (lv && l != left) {
                if (top) atomicMin(&s_bb[4 * l], gy);
                if (bot) atomicMax(&s_bb[4 * l + 2], gy);
                atomicMin(&s_bb[4 * l + 1], gx);
            }
            if (lv && l != right) atomicMax(&s_bb[4 * l + 3], gx);
        } else {
            if (lv && l != left) { atomicMin(&mn[2 * l], gy); atomicMax(&mx[2 * l], gy); atomicMin(&mn[2 * l + 1], gx); }
            if (lv && l != right) atomicMax(&mx[2 * l + 1], gx);
        }
        if (in && !lv) {
            float v = src[fo + (size_t)gy * cols + gx];
            if (coef) v = norm_apply(v, coef[2 * f], coef[2 * f + 1]);    // N1 normalisation, if any
            x4[fo + (size_t)gy * cols + gx] = invert_valid(v, max_depth, thr);
        }
    }
    }
    if constexpr (LDS_TABLE) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_labels; i += 256) {
            if (s_bb[4 * i + 2] >= 0) {                // touched by this workgroup
                atomicMin(&mn[2 * i], s_bb[4 * i]); atomicMin(&mn[2 * i + 1], s_bb[4 * i + 1]);
                atomicMax(&mx[2 * i], s_bb[4 * i + 2]); atomicMax(&mx[2 * i + 1], s_bb[4 * i + 3]);
            }
        }
    }
}

// The masked pipeline of one label (or of two labels side by side) down the rows of its box.  L, y0, y1, gx and
// outlane may be wave-uniform (one label per wave: the compiler keeps them on the scalar unit) or differ between
// the two lane segments of a packed wave; nsteps is always uniform.
template <int K0KIND, bool NORM, typename IntT>
__device__ __forceinline__ void label_pipeline(const FrameBuf& sb, const FrameBuf& lb, const FrameBuf& ob, IntT L, IntT y0, IntT y1, int gx,
                                               bool outlane, int nsteps, int rows, int cols, float max_depth, float thr, float na, float nb)
{
    constexpr int H = 6;                           // rows of reach above and below (H3 + H4)
    constexpr float NEG = -FLT_MAX, POS = FLT_MAX;
    const bool incol = gx >= 0 && gx < cols;
    const unsigned gb = 4u * (unsigned)min(max(gx, 0), cols - 1);
    auto row_off = [&](IntT r) -> unsigned { return gb + 4u * (unsigned)(min(max(r, (IntT)0), (IntT)(rows - 1)) * cols); };
    float PF[8], XR[8], A3[8], S1[8], H4[8], HE[8];
    int PL[8], LB[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { PF[q] = 0.f; XR[q] = NEG; A3[q] = NEG; S1[q] = NEG; H4[q] = NEG; HE[q] = POS; PL[q] = -1; LB[q] = -1; }
    const IntT rs = y0 - H;                        // image row of step 0
    constexpr int PFD = 4;
#pragma unroll
    for (int q = 0; q < PFD; ++q) {
        const unsigned ro = row_off(rs + q);
        PF[q] = sb.ld_at(ro); PL[q] = __builtin_bit_cast(int, lb.ld_at(ro));
    }
    for (int s0 = 0; s0 < nsteps; s0 += 8) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const IntT i = rs + (s0 + p);            // image row fed by this step
            const float raw = PF[p];
            const int lab = PL[p];
            {
                const unsigned ro = row_off(i + PFD);
                PF[(p + PFD) & 7] = sb.ld_at(ro); PL[(p + PFD) & 7] = __builtin_bit_cast(int, lb.ld_at(ro));
            }
            LB[p] = lab;
            // masked copy (LC :94-95): the label's pixels keep their H2 value, other image pixels are 0;
            // outside the image the dilate border value
            const bool inimg = incol && (unsigned)i < (unsigned)rows;
            const float x2 = inimg ? (lab == L ? invert_valid(NORM ? norm_apply(raw, na, nb) : raw, max_depth, thr) : 0.0f) : NEG;
            const IntT j = i - 2;
            float y3;
            if constexpr (K0KIND == K0_AS_COMPILED) {
                const float s1 = from_right(x2);
                const float s2 = from_right(s1);
                S1[p] = s1;
                y3 = fmax2(S1[(p + 5) & 7], s2);
            } else {
                const float a3 = hmax3(x2);
                XR[p] = x2;
                A3[p] = a3;
                const float a5j = hgrow_max(A3[(p + 6) & 7]);
                y3 = fmax2(fmax3(XR[(p + 4) & 7], A3[(p + 5) & 7], a5j), fmax2(A3[(p + 7) & 7], x2));
            }
            y3 = (incol && (unsigned)j < (unsigned)rows) ? y3 : NEG;
            H4[(p + 6) & 7] = hgrow_max(hmax3(y3));
            const IntT k = i - 4;
            float d4 = fmax3(fmax3(H4[(p + 2) & 7], H4[(p + 3) & 7], H4[(p + 4) & 7]), H4[(p + 5) & 7], H4[(p + 6) & 7]);
            d4 = (incol && (unsigned)k < (unsigned)rows) ? d4 : POS;
            HE[(p + 4) & 7] = hgrow_min(hmin3(d4));
            const IntT l = i - 6;
            const float e4 = fmin3(fmin3(HE[(p + 0) & 7], HE[(p + 1) & 7], HE[(p + 2) & 7]), HE[(p + 3) & 7], HE[(p + 4) & 7]);
            // write-back only where the label is this one (LC :101)
            // every lane stores on every step (exact load waits); the ones with nothing to write aim past the buffer (dropped)
            ob.st_at((l >= y0 && l <= y1 && outlane && LB[(p + 2) & 7] == L) ? gb + 4u * (unsigned)(l * cols) : kDropOffset, e4);
        }
    }
}

// PAIRS: one wave64 per pair of labels (2w, 2w+1).  The chain's horizontal reach is HL columns to the left and HR
// to the right (as-compiled element: 4 / 6, diamond: 6 / 6).  If both boxes, each grown by that halo, fit side by
// side in the wave's 64 lanes, they run as two lane segments of ONE pass -- lanes carry their own label, box and
// image row, and the DPP shifts that cross the segment boundary only ever reach halo lanes; superpixel boxes are
// ~20 columns wide, so a lone box leaves two thirds of the wave idle.  A pair that does not fit runs one label
// after the other.  !PAIRS: one wave per label (the host picks this when the labels are few and large: packing
// cannot apply and twice as many, shorter waves fill the GPU better).  Boxes wider than VW = 64 - HL - HR
// columns are walked in chunks.  Either mode is correct for any label plane; the choice only affects speed.
template <int K0KIND, bool NORM, bool PAIRS>
__global__ __launch_bounds__(256)
void k_label_stage_s(const float* __restrict__ src, const int32_t* __restrict__ labels, int n_labels,
                     int* __restrict__ bb_min, int* __restrict__ bb_max, float* __restrict__ x4,
                     int rows, int cols, float max_depth, float thr, const float* __restrict__ coef)
{
    constexpr int HL = K0KIND == K0_AS_COMPILED ? 4 : 6, HR = 6, VW = 64 - HL - HR, H = 6;
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int La = PAIRS ? 2 * wv : wv, Lb = La + 1;
    if (La >= n_labels) return;
    const int f = blockIdx.y;
    const size_t bo = (size_t)f * n_labels;
    const int y0a = bb_min[(bo + La) * 2], x0a = bb_min[(bo + La) * 2 + 1], y1a = bb_max[(bo + La) * 2], x1a = bb_max[(bo + La) * 2 + 1];
    int y0b = 0, x0b = 0, y1b = -1, x1b = -1;
    if (PAIRS && Lb < n_labels) { y0b = bb_min[(bo + Lb) * 2]; x0b = bb_min[(bo + Lb) * 2 + 1]; y1b = bb_max[(bo + Lb) * 2]; x1b = bb_max[(bo + Lb) * 2 + 1]; }
    const bool ea = y1a >= 0, eb = y1b >= 0;     // a label may own no pixel
    if (!ea && !eb) return;
    // every label's box is read by exactly this one wave: at its end it leaves the tables as k_label_bbox expects to find them (no box: what an
    // empty label's entries still hold), so that no memset has to run in the stream in front of the next call
    auto reset_boxes = [&]() {
        if (lane == 0) {
            bb_min[(bo + La) * 2] = kBboxNone; bb_min[(bo + La) * 2 + 1] = kBboxNone; bb_max[(bo + La) * 2] = -1; bb_max[(bo + La) * 2 + 1] = -1;
            if (PAIRS && Lb < n_labels) { bb_min[(bo + Lb) * 2] = kBboxNone; bb_min[(bo + Lb) * 2 + 1] = kBboxNone; bb_max[(bo + Lb) * 2] = -1; bb_max[(bo + Lb) * 2 + 1] = -1; }
        }
    };
    const size_t fo = (size_t)f * rows * cols, fe = (size_t)rows * cols;
    float na = 1.0f, nb = 0.0f;                    // N1 normalisation
    if constexpr (NORM) { na = coef[2 * f]; nb = coef[2 * f + 1]; }
    FrameBuf sb, lb, ob;
    sb.init(src + fo, fe);
    lb.init(reinterpret_cast<const float*>(labels + fo), fe);
    ob.init(x4 + fo, fe);
    if constexpr (PAIRS) {
        const int wa = x1a - x0a + 1, wb = x1b - x0b + 1;
        if (ea && eb && wa + wb + 2 * (HL + HR) <= 64) {
            const int split = HL + wa + HR;        // first lane of the second segment
            const bool second = lane >= split;
            const int li = second ? lane - split : lane;
            const int gx = (second ? x0b : x0a) - HL + li;
            const bool outlane = gx >= 0 && gx < cols && li >= HL && li < HL + (second ? wb : wa);
            const int ha = y1a - y0a, hb = y1b - y0b;
            label_pipeline<K0KIND, NORM, int>(sb, lb, ob, second ? Lb : La, second ? y0b : y0a, second ? y1b : y1a, gx, outlane,
                                              (ha > hb ? ha : hb) + 1 + 2 * H, rows, cols, max_depth, thr, na, nb);
            reset_boxes();
            return;
        }
    }
    for (int w = 0; w < (PAIRS ? 2 : 1); ++w) {
        const int L = w ? Lb : La, y0 = w ? y0b : y0a, y1 = w ? y1b : y1a, x0 = w ? x0b : x0a, x1 = w ? x1b : x1a;
        if (y1 < 0) continue;
        for (int cx = x0; cx <= x1; cx += VW) {
            const int gx = cx - HL + lane;
            const bool outlane = gx >= 0 && gx < cols && lane >= HL && lane < HL + VW && gx <= x1;
            label_pipeline<K0KIND, NORM, int>(sb, lb, ob, L, y0, y1, gx, outlane, y1 - y0 + 1 + 2 * H, rows, cols, max_depth, thr, na, nb);
        }
    }
    reset_boxes();
}

// ---------------------------------------------------------------------------------
// k_post_s : H9 median 5x5 (replicate), H10 Gaussian (reflect-101) + select, H11 invert
// mode: 9 = stop after the median, 10 = after the blur, 11 = everything
//
// The exact median is time-shared down the column (tools/gen_median_shared.py has the scheme
// and its verification): every row's 5 horizontal neighbours are sorted once (12 three-input
// instructions), every second row a pair of sorted rows is merged (19) and the six middle order
// statistics of the 4-row core are extracted (24); each window's median is then the 6th smallest of
// those six and the sorted fifth row (5).  38.5 instructions per pixel instead of ~200
// (tools/gen_median_3in.py has the three-input rewriting of the networks and its proof).
// ---------------------------------------------------------------------------------
struct PostS {
    static constexpr int H = 4;                  // 2 (median) + 2 (Gaussian) lanes lost per side
    static constexpr int VW = 64 - 2 * H;
};

// One pass of the [1 4 6 4 1]/16 filter in the reference order  c*k0 + s1*k1 + s2*k2  (s1, s2 = the already rounded
// sums of the two neighbour pairs; every product rounded, the sum taken left to right).  k1 = 1/4 and k2 = 1/16 are
// powers of two, so those two products are exact and folding them into fused multiply-adds changes no rounding:
// round(a + exact(s*k)) is what the unfused sequence computes too.  (The one exception is a product that underflows
// into a subnormal and loses bits there, |s| < 2^-122 -- forty orders of magnitude below a depth in metres.)
__device__ __forceinline__ float gauss_taps(float c, float s1, float s2)
{
    return __builtin_fmaf(s2, 0.0625f, __builtin_fmaf(s1, 0.25f, __fmul_rn(c, 0.375f)));
}

// The streaming post pipeline (H9..H11) of one wave: state + one step.  Shared by k_post_s
// (input rows from global memory) and k_fp_s (input rows straight from the fill stage).
// MODE (9/10/11) and BLUR are compile-time: a run-time branch around the ring updates would
// make every join copy the rings (whole-array phis).  PP = u & 7 is the static ring phase.
//
// Step u takes X7 row clamp(u - 2) (replicate rows), finishes the median of image row u - 4
// and the output of image row u - 6.
// FILLED: the caller guarantees that every frame whose X7 still holds a hole is recomputed afterwards (k_fp_s followed by the
// redo chain).  A frame without holes has every median >= thr, so the masked select of LO :184 always takes the blurred value
// and its compare + select (two of the slow instructions per pixel) are left out; what a frame WITH holes gets is overwritten.
template <int MODE, bool BLUR, bool FILLED = false>
struct PostPipe {
    static constexpr bool do_blur = BLUR && MODE >= 10;
    MedianColumn mc;         // the vertical half of the median (dcmt_median.h)
    float G1[8], MR[8];      // horizontal Gaussian / median rows, slot (image row) & 7
    float last_out;          // the value this lane stored last (k_fp_s repeats output row 0 above a shifted origin)
    // per-lane constants
    FrameBuf of;             // output frame
    unsigned ob;             // byte offset of this lane's (clamped) column
    int rows, cols, gx, rl;
    int so0, so1;            // output rows [so0, so1) are stored (all of them, unless the wave is one row band of its strip: k_fp_s)
    bool outlane, edge_strip;
    float max_depth, thr;

    __device__ __forceinline__ void init(float* out_frame, int rows_, int cols_, int gx0, int lane, float max_depth_, float thr_)
    {
        of.init(out_frame, (size_t)rows_ * cols_); rows = rows_; cols = cols_; max_depth = max_depth_; thr = thr_;
        so0 = 0; so1 = rows_;
        gx = gx0 + lane;
        ob = 4u * (unsigned)min(max(gx, 0), cols - 1);
        outlane = gx >= 0 && gx < cols && lane >= PostS::H && lane < 64 - PostS::H;
        rl = reflect101(gx, cols) - gx0;      // reflect-101 source lane for the Gaussian's out-of-image columns
        edge_strip = gx0 < 0 || gx0 + 63 >= cols;
        mc.init();
        last_out = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { G1[q] = 0.f; MR[q] = 0.f; }
    }

    template <int PP>
    __device__ __forceinline__ void step(float x, int u)
    {
        // ---- H9 (LO :170)
        float s[5];
        {
            const float l1 = from_left(x), r1 = from_right(x);
            s[0] = from_left(l1); s[1] = l1; s[2] = x; s[3] = r1; s[4] = from_right(r1);
        }
        sort5(s);
        const float m = mc.template step<PP>(s);                       // median of the window whose bottom row is u
        const int j = u - 4;                                           // image row of this median
        if constexpr (MODE == 9) {
            if ((unsigned)j < (unsigned)rows && outlane) of.st(ob, j, cols, m);
            return;
        }
        MR[(PP + 4) & 7] = m;
        // ---- H10 (LO :179): horizontal [1 4 6 4 1]/16 with reflect-101 columns
        if constexpr (do_blur) {
            float mf = m;
            if (edge_strip) { const float mr = __shfl(m, rl, 64); mf = (gx < 0 || gx >= cols) ? mr : m; }
            const float ml1 = from_left(mf), mr1 = from_right(mf);
            const float ml2 = from_left(ml1), mr2 = from_right(mr1);
            G1[(PP + 4) & 7] = gauss_taps(mf, __fadd_rn(ml1, mr1), __fadd_rn(ml2, mr2));
        }
        // ---- vertical pass + select + invert for output row o = j - 2 = u - 6
        const int o = u - 6;
        if ((unsigned)(o - so0) < (unsigned)(so1 - so0)) {
            // slots: row o -> (PP+2)&7, o+1 -> PP+3, o+2 -> PP+4, o-1 -> PP+1, o-2 -> PP
            const float mo = MR[(PP + 2) & 7];
            // the rest of the row, given the four vertical neighbours of the horizontal pass
            auto finish = [&](float u1, float u2, float d1, float d2) {
                float val = mo;
                if constexpr (do_blur) {
                    const float acc = gauss_taps(G1[(PP + 2) & 7], __fadd_rn(u1, d1), __fadd_rn(u2, d2));
                    if (FILLED || mo >= thr) val = acc;                 // LO :184
                }
                if constexpr (MODE >= 11) val = invert_valid(val, max_depth, thr);  // LO :191-202
                of.st(outlane ? ob : kDropOffset, o, cols, val);     // every lane stores; halo lanes aim past the buffer (dropped)
                last_out = val;
            };
            const float g_p2 = G1[(PP + 4) & 7], g_p1 = G1[(PP + 3) & 7], g_0 = G1[(PP + 2) & 7];
            const float g_m1 = G1[(PP + 1) & 7], g_m2 = G1[PP];
            if (do_blur && (o < 2 || o + 2 >= rows)) {                    // reflect-101 rows (rows >= 8 guaranteed): the first and
                finish(o >= 1 ? g_m1 : g_p1,                             // last two rows get their own copy of the tail, so the
                       o >= 2 ? g_m2 : (o == 1 ? g_0 : g_p2),            // common path reads the ring registers in place
                       o + 1 < rows ? g_p1 : g_m1,
                       o + 2 < rows ? g_p2 : (o + 2 == rows ? g_0 : g_m2));
            } else {
                finish(g_m1, g_m2, g_p1, g_p2);
            }
        }
    }
};

// only_if_holes: recompute pass behind k_fp_s -- frames that needed no loop application are
// already final and are skipped.
template <int MODE, bool BLUR>
__global__ __launch_bounds__(256)
void k_post_s(const float* __restrict__ pp0, const float* __restrict__ pp1, float* __restrict__ dst,
              const int* __restrict__ counters, int n_apps_launched, int rows, int cols, int strips,
              int batch, int xcd_map, float max_depth, float thr, int only_if_holes)
{
    const int lane = threadIdx.x & 63;
    int f, sg;
    frame_unit(blockIdx.x, (strips + 3) / 4, batch, xcd_map, f, sg);
    const int strip = sg * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keep it scalar
    if (strip >= strips) return;
    const int* cnt = counters + (size_t)f * kCntStride;
    if (only_if_holes && cnt[1] == 0) return;
    const int a = apps_done(cnt, n_apps_launched);
    const size_t fo = (size_t)f * rows * cols;
    const int gx0 = strip * PostS::VW - PostS::H;
    const int gxc = min(max(gx0 + lane, 0), cols - 1);               // BORDER_REPLICATE for the median
    FrameBuf sf;
    sf.init(((a & 1) ? pp1 : pp0) + fo, (size_t)rows * cols);
    const unsigned sb = 4u * (unsigned)gxc;
    PostPipe<MODE, BLUR> pipe;
    pipe.init(dst + fo, rows, cols, gx0, lane, max_depth, thr);

    float PF[8];
    constexpr int PFD = 4;
#pragma unroll
    for (int q = 0; q < 8; ++q) PF[q] = 0.f;
#pragma unroll
    for (int q = 0; q < PFD; ++q) PF[q] = sf.ld(sb, min(max(q - 2, 0), rows - 1), cols);

    const int nsteps = rows + 6;                 // the last output row o = t - 6 = rows - 1
    for (int t0 = 0; t0 < nsteps; t0 += 8) {
        static_for<0, 8>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            const int t = t0 + p;
            const float x = PF[p];
            PF[(p + PFD) & 7] = sf.ld(sb, min(max(t + PFD - 2, 0), rows - 1), cols);
            pipe.template step<p>(x, t);
        });
    }
}

// ---------------------------------------------------------------------------------
// k_fill_s : x = x < 0.1 ? dilate31(x) : x, streaming like k_pre_s (LO :131-144 and one
// iteration of :146-166).  One wave64 owns 64 columns over the full height.
//   vertical 31-max:   doubling in rolling registers (w2, w4, w8, w16, then w16 + w16 shifted
//                      by 15): 5 v_max per row, no redundancy (the strip is full height);
//   horizontal 31-max: the wave's 64 lanes are four DPP rows of 16.  P / S = inclusive prefix /
//                      suffix maxima inside each 16-lane row (4 + 4 v_max_f32_dpp row_shr/row_shl),
//                      M = max(P, S) = the row's maximum.  A window [c-15, c+15] always covers
//                      the whole 16-block of c, the tail of the block before it and the head
//                      of the block after it:   out(c) = max(S(c-15), M(c), P(c+15)),
//                      with the two neighbours fetched by ds_bpermute (no LDS storage).
//                      34 of the 64 lanes (15..48) produce output.
// A max filter centred inside the image gives the same result with replicated borders as with
// the -FLT_MAX constant border, so all loads are simply clamped into the image: no masks.
// app == 0: H7, counts the holes it sees (cnt[0]) and leaves (cnt[1]).
// app >= 1: loop iteration; frames whose previous application left no holes return at once.
// redo:     application 0 again, without counting, for the frames k_fp_s could not finish.
// ---------------------------------------------------------------------------------
struct FillS {
    static constexpr int R = 15;
    static constexpr int VW = 64 - 2 * R;        // 34 output columns per wave
};

// x = max(x, x shifted inside its 16-lane DPP row); lanes without a source keep x (bound_ctrl:0
// disables them).  Written as in-place inline asm: hipcc cannot fold a "keep" DPP move into
// v_max_f32 (it emits v_mov + v_mov_dpp + v_max), and it pads no hazards for asm, so the two
// wait states between a VALU write of x and its DPP read are inside the string.
#define DCMT_MAX_DPP(NAME, CTRL)                                                              \
    __device__ __forceinline__ void NAME(float& x)                                            \
    {                                                                                         \
        asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(x)); \
    }
DCMT_MAX_DPP(max_shr1, "row_shr:1") DCMT_MAX_DPP(max_shr2, "row_shr:2") DCMT_MAX_DPP(max_shr4, "row_shr:4") DCMT_MAX_DPP(max_shr8, "row_shr:8")
DCMT_MAX_DPP(max_shl1, "row_shl:1") DCMT_MAX_DPP(max_shl2, "row_shl:2") DCMT_MAX_DPP(max_shl4, "row_shl:4") DCMT_MAX_DPP(max_shl8, "row_shl:8")
#undef DCMT_MAX_DPP
__device__ __forceinline__ float row_prefix_max(float x) { max_shr1(x); max_shr2(x); max_shr4(x); max_shr8(x); return x; }
__device__ __forceinline__ float row_suffix_max(float x) { max_shl1(x); max_shl2(x); max_shl4(x); max_shl8(x); return x; }
// Three scans at once (prefix of a, suffix of a, prefix of b), their steps interleaved: every DPP read is two
// instructions behind the write of its source, which is exactly the two wait states the hazard needs, so the whole
// block carries one s_nop instead of twelve.
__device__ __forceinline__ void row_scans3(float a, float b, float& pa, float& sa, float& pb)
{
    pa = a; sa = a; pb = b;
#define DCMT_S3(SH, N) "v_max_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                       "v_max_f32_dpp %1, %1, %1 row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                       "v_max_f32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t" DCMT_S3(, 1) DCMT_S3(, 2) DCMT_S3(, 4) DCMT_S3(, 8) : "+v"(pa), "+v"(sa), "+v"(pb));
#undef DCMT_S3
}

__global__ __launch_bounds__(256)
void k_fill_s(const float* __restrict__ in, float* __restrict__ out, int* __restrict__ counters,
              int rows, int cols, int strips, int batch, int xcd_map, float thr, int app, int redo, const int* __restrict__ tb, int tbands,
              const unsigned short* __restrict__ in16, const int* __restrict__ q16_bad)
{
    const int lane = threadIdx.x & 63;
    int f, sg;
    frame_unit(blockIdx.x, (strips + 3) / 4, batch, xcd_map, f, sg);
    const int strip = sg * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (strip >= strips) return;
    int* cnt = frame_counters(counters, f);
    if (app >= 1 && cnt[app] == 0) return;       // holes left by application app-1: none
    if (redo && cnt[1] == 0) return;             // redo of application 0 behind k_fp_s: only frames that need the loop
    const int gx = strip * FillS::VW - FillS::R + lane;
    const bool outlane = gx >= 0 && gx < cols && lane >= FillS::R && lane < 64 - FillS::R;
    const size_t fo = (size_t)f * rows * cols;
    const int gxc = min(max(gx, 0), cols - 1);
    const float* sp = in + fo + gxc;
    float* op = out + fo + gxc;
    // in16: X6 was left as 16-bit codes (k_pre_p<Q16OUT>) unless that attempt raised *q16_bad and the f32 kernels reran into `in`
    const bool q16 = in16 != nullptr && *q16_bad == 0;
    const unsigned short* sq = in16 ? in16 + fo + gxc : nullptr;
    auto ld_in = [&](size_t off) -> float { return q16 ? Q16::value(sq[off]) : sp[off]; };
    const int a_lo = ((lane - FillS::R) & 63) * 4, a_hi = ((lane + FillS::R) & 63) * 4;   // bpermute byte addresses
    // tb: `in` is an X6 whose extension zones were never written (k_pre_s / k_pre_p, table mode): the rows above a column's first
    // valid row equal that row, the rows below its last valid row equal that one, so the row index is clamped per lane
    int tl = 0, bl = rows - 1;
    if (tb) table_rows(tb, f, cols, tbands, rows, gxc, tl, bl);

    // rolling rows, slot = row & 15 (16-step unroll keeps every index static)
    float PF[16], XC[16], W2[16], W4[16], W8[16], W16[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { PF[q] = 0.f; XC[q] = W2[q] = W4[q] = W8[q] = W16[q] = -FLT_MAX; }
    // step t handles the (row-clamped) input row v = t - 15 and emits output row o = t - 30
    constexpr int PFD = 8;
#pragma unroll
    for (int q = 0; q < PFD; ++q) PF[q] = ld_in((size_t)min(max(min(max(q - FillS::R, 0), rows - 1), tl), bl) * cols);
    float vprev = -FLT_MAX;
    int before = 0, after = 0;
    const int nsteps = rows + 2 * FillS::R;
    for (int t0 = 0; t0 < nsteps; t0 += 16) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int t = t0 + p;
            const float x = PF[p];
            PF[(p + PFD) & 15] = ld_in((size_t)min(max(min(max(t + PFD - FillS::R, 0), rows - 1), tl), bl) * cols);
            XC[p] = x;
            // vertical: windows ending at row t of 2, 4, 8, 16, 31 rows
            const float w2 = fmax2(x, vprev);
            vprev = x;
            W2[p] = w2;
            const float w4 = fmax2(w2, W2[(p + 14) & 15]);          // t-2
            W4[p] = w4;
            const float w8 = fmax2(w4, W4[(p + 12) & 15]);          // t-4
            W8[p] = w8;
            const float w16 = fmax2(w8, W8[(p + 8) & 15]);          // t-8
            const float w31 = fmax2(w16, W16[(p + 1) & 15]);        // t-15: rows t-30 .. t
            W16[p] = w16;
            // horizontal: columns c-15 .. c+15 of the vertical maxima
            const float P = row_prefix_max(w31), S = row_suffix_max(w31);
            const float s_lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a_lo, __builtin_bit_cast(int, S)));
            const float p_hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a_hi, __builtin_bit_cast(int, P)));
            const float d = fmax3(fmax2(P, S), s_lo, p_hi);
            // output row o = t - 30; its own value sits 15 steps back
            const int o = t - 2 * FillS::R;
            if ((unsigned)o < (unsigned)rows) {
                const float v = XC[(p + 1) & 15];                    // virtual row (t-15) - 15 ... see below
                const bool hole = v < thr;                           // LO :140 / :154
                const float r = hole ? d : v;
                if (outlane) op[(size_t)o * cols] = r;
                before += __builtin_popcountll(__ballot(hole && outlane));
                after += __builtin_popcountll(__ballot(r < thr && outlane));
            }
        }
    }
    // hole counts of this strip (accumulated on the scalar unit)
    if (lane == 0 && !redo) {                     // a redo recomputes what k_fp_s already counted
        if (app == 0 && before) atomicAdd(&cnt[0], before);
        if (after) atomicAdd(&cnt[1 + app], after);
    }
}

// ---------------------------------------------------------------------------------
// k_fp_s : k_fill_s (application 0) and k_post_s in ONE streaming kernel -- X7 never goes to
// memory.  One wave64 owns the 64 columns of a post strip (56 of them produce output).
//
// The 31-wide horizontal maximum needs 15 more columns on each side.  They ride in a second
// register per lane ("B"): lanes 0..14 hold the 15 columns right of the strip, lanes 49..63
// the 15 columns left of it, so that both are whole 16-lane DPP rows and the same
// prefix / suffix row scans apply.  With M = max(P, S) of the lane's own 16-block:
//     out(c) = max( S'(c >= 15 ? c - 15 : 63 - c), M(c), P'((c + 15) mod 64) ),
//     S' = lane >= 49 ? P_B : S_A,   P' = lane <= 14 ? P_B : P_A
// -- the left halo sits in B in descending column order, so both halos need the same prefix scan.  Every one of the 64
// lanes therefore gets its exact X7 value, and the post pipeline (PostPipe) runs on it in the
// same step, 29 steps behind the fill front end (30 rows of fill latency + one step of skew; its own 2 rows of replicate
// padding are the three post steps run at t = 31): step t feeds X7 row t - 31 to post step u = t - 29.
//
// The vertical 31-max of both registers is the doubling of k_fill_s; the two 15-step delays that
// are only read once (the centre values and the B register's 16-row maxima) live in a
// wave-private LDS delay line instead of 32 VGPRs.
//
// Correct only for frames that need no hole-closure loop application (the common case: every
// hole is filled by H7).  The kernel counts the holes it leaves per frame (cnt[1]); frames with
// cnt[1] > 0 are recomputed afterwards by k_fill_s / k_post_s (their only_if_holes modes).
// ---------------------------------------------------------------------------------
constexpr int kBandHalo = 19;        // rows of X6 an output row needs above and below it: 15 (H7) + 2 (median) + 2 (Gaussian)
struct FpS {
    static constexpr int LAG = 29;               // post step u = t - LAG: it takes X7 row u - 2 = t - 31, which the fill front end returns in the same step
};

template <bool BLUR, bool FILLED = false>
__global__ __launch_bounds__(256)
void k_fp_s(const float* __restrict__ x6, float* __restrict__ dst, int* __restrict__ counters,
            int rows_all, int cols, int strips, int batch, int xcd_map, float max_depth, float thr, const int* __restrict__ tb,
            int tbands, const int* __restrict__ gate, int fbands)
{
    if (gate && *gate == 0) return;          // the f32 rerun behind a 16-bit attempt (k_fp_q): only if that attempt raised its flag
    // per wave, three 15-step delay lines: centre values, A's 16-row maxima (64 lanes each), B's 16-row
    // maxima (only its 30 halo lanes: packed to 32) -- 10 KiB per wave, 40 KiB per workgroup: 4 fit a CU
    __shared__ float s_delay[4][16 * (64 + 64 + 32)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int f, unit;
    if (!wave_strip(blockIdx.x, wave, strips * fbands, batch, xcd_map, f, unit)) return;
    const int strip = unit % strips, band = unit / strips;           // fbands > 1: the strip in row bands, one wave each (small batches)
    int* cnt = frame_counters(counters, f);
    const size_t fo = (size_t)f * rows_all * cols;
    const int gx0 = strip * PostS::VW - PostS::H;
    const int gxa = gx0 + lane;
    // B: lanes 0..14 = the 15 columns right of the strip (ascending); lanes 49..63 = the 15 columns left of it
    // in DESCENDING order (lane 49 = gx0-1 ... lane 63 = gx0-15; lane 48 duplicates gx0-1), so that the
    // suffix the left side needs is a PREFIX over lanes too and one row scan serves both halos
    const int gxb = lane <= 14 ? gx0 + 64 + lane : (lane >= 49 ? gx0 + 48 - lane : (lane == 48 ? gx0 - 1 : gxa));
    const int gxac = min(max(gxa, 0), cols - 1), gxbc = min(max(gxb, 0), cols - 1);   // clamped: replicate == constant border for a max filter
    // ---- the extension zones.  In table mode k_pre_s / k_pre_p write neither the rows above a column's first valid row ti (they
    // all equal row ti) nor the rows below its last valid row bi (they equal row bi): a row index is clamped per lane into
    // [ti, bi].  With T = the smallest ti of the 94 columns this wave reads, rows 0 .. T of X6 are constant down every one of
    // those columns and hole-free, hence so are X7 (rows <= T), the median (rows <= T - 2) and the output (rows <= T - 4,
    // whichever border rule the top rows use).  The wave therefore treats row V = T - 8 as the top of its frame: the clamped
    // row loads reproduce the real rows above V (they equal row V), the median's replicated and the Gaussian's reflected
    // rows above V are rows of the constant zone like the real ones, and the first output row (V) is stored to rows
    // 0 .. V-1 as well -- on velodyne-like frames (upper third empty) that is a quarter of the row steps of this kernel.
    int tia = 0, tib = 0, bia = rows_all - 1, bib = rows_all - 1, V = 0;
    if (tb) {
        table_rows(tb, f, cols, tbands, rows_all, gxac, tia, bia);
        table_rows(tb, f, cols, tbands, rows_all, gxbc, tib, bib);
        V = __builtin_amdgcn_readfirstlane(max(wave_min_i(min(tia, tib)) - 8, 0));   // wave-uniform: keep the row arithmetic scalar
    }
    // ---- row bands (fbands > 1: a batch too small to fill the GPU with one wave per strip).  The rows V .. rows_all-1 in fbands
    // equal parts; a band [b0, b1) streams the rows b0 - 19 .. b1 + 18 (an output row needs X6 within 15 + 2 + 2 rows), treats that
    // range as its frame -- what the border rules make of its first and last 19 rows is never stored or counted -- and stores
    // only its own rows.  Band 0 keeps the frame's real top (and the rows above V), the last band its real bottom.
    int so0 = 0, E = rows_all;
    const int V_top = V;
    if (fbands > 1) {
        const int span = rows_all - V_top;
        const int b0 = V_top + (int)((long long)span * band / fbands), b1 = V_top + (int)((long long)span * (band + 1) / fbands);
        if (band > 0) V = max(V_top, b0 - kBandHalo);
        if (band < fbands - 1) E = min(rows_all, b1 + kBandHalo);
        so0 = b0 - V;
        E = max(E, V + 9);
        // (so1 below; a band of a very short range may be empty: nothing stored, nothing counted)
    }
    const int rows = E - V;                                          // rows of the frame as this wave sees it (>= 9)
    int so1 = rows;
    if (fbands > 1) {
        const int span = rows_all - V_top;
        so1 = V_top + (int)((long long)span * (band + 1) / fbands) - V;
        if (band == fbands - 1) so1 = rows;
    }
    FrameBuf sf;
    sf.init(x6 + fo, (size_t)rows_all * cols);
    const unsigned rowb = 4u * (unsigned)cols;
    const unsigned sba = 4u * (unsigned)gxac + (unsigned)V * rowb;   // byte offset of (row V, this lane's column)
    const unsigned sbb = 4u * (unsigned)gxbc + (unsigned)V * rowb;
    // (a row band may start below a column's last valid row: every row it sees of that column is row bi)
    const unsigned fla = 4u * (unsigned)gxac + (unsigned)min(max(tia, V), bia) * rowb, cea = 4u * (unsigned)gxac + (unsigned)bia * rowb;
    const unsigned flb = 4u * (unsigned)gxbc + (unsigned)min(max(tib, V), bib) * rowb, ceb = 4u * (unsigned)gxbc + (unsigned)bib * rowb;
    // (lo <= hi: the median of the three is the clamp.  hipcc has no builtin for the integer med3 and does not form it from min(max()).)
    auto clamp3 = [](unsigned a, unsigned lo, unsigned hi) -> unsigned { unsigned r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi)); return r; };
    auto ld_a = [&](int row) -> float { return sf.ld_at(clamp3(sba + (unsigned)row * rowb, fla, cea)); };   // row relative to V, already clamped to [0, rows)
    auto ld_b = [&](int row) -> float { return sf.ld_at(clamp3(sbb + (unsigned)row * rowb, flb, ceb)); };
    const bool own = gxa >= 0 && gxa < cols && lane >= PostS::H && lane < 64 - PostS::H;   // columns this wave accounts for
    const unsigned long long own_mask = __ballot(own);          // wave-uniform: the hole counts stay on the scalar unit
    const bool edge_strip = gx0 < 0 || gx0 + 63 >= cols;
    const int src_lane = min(max(gxa, 0), cols - 1) - gx0;          // BORDER_REPLICATE columns for the median
    const int a_lo = (lane < 15 ? 63 - lane : lane - 15) * 4, a_hi = ((lane + 15) & 63) * 4;   // ds_bpermute byte addresses
    float (*dl_c)[64] = reinterpret_cast<float (*)[64]>(s_delay[wave]);
    float (*dl_a)[64] = reinterpret_cast<float (*)[64]>(s_delay[wave] + 16 * 64);
    float (*dl_b)[32] = reinterpret_cast<float (*)[32]>(s_delay[wave] + 16 * 128);
    const int lb = lane <= 14 ? lane : (lane >= 48 ? lane - 32 : 15);   // B's live lanes 0..14, 48..63 -> words 0..14, 16..31; the dead lanes share word 15

    PostPipe<11, BLUR, FILLED> pipe;
    pipe.init(dst + fo + (size_t)V * cols, rows, cols, gx0, lane, max_depth, thr);
    pipe.so0 = so0; pipe.so1 = so1;

    constexpr float NEG = -FLT_MAX;
    // Warm start: with V > 0 the rows above V equal row V (a hole-free row of the constant zone), so the first 16 steps -- which
    // would feed row V sixteen times (stream rows 0..15 = image rows V-15..V, clamped) -- are replaced by the state they leave
    // behind: every window maximum and every delay-line slot holds row V.
    const bool warm = V > 0;
    const float xa0 = warm ? ld_a(0) : NEG, xb0 = warm ? ld_b(0) : NEG;
    float PFA[16], PFB[16], W2A[16], W6A[16], W2B[16], W6B[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { PFA[q] = PFB[q] = 0.f; W2A[q] = W6A[q] = xa0; W2B[q] = W6B[q] = xb0; }
#pragma unroll
    for (int q = 0; q < 16; ++q) { dl_c[q][lane] = xa0; dl_a[q][lane] = xa0; dl_b[q][lb] = xb0; }
#ifndef DCMT_FP_PFD
#define DCMT_FP_PFD 6
#endif
    constexpr int PFD = DCMT_FP_PFD;         // rows of load lookahead
#pragma unroll
    for (int q = 0; q < PFD; ++q) {
        const int row = min(max(q + (warm ? 16 : 0) - 15, 0), rows - 1);       // step 0's (or step 16's) first rows
        PFA[q] = ld_a(row); PFB[q] = ld_b(row);
    }
    float vpa = xa0, vpb = xb0, x7_prev = warm ? xa0 : 0.f;
    int before = 0, after = 0;
    // Software skew: the two ds_bpermutes and the two delay-line reads issued in step t are consumed
    // in step t + 1, so their LDS round trips overlap the next step's arithmetic instead of
    // stalling the wave (s_waitcnt) three times per step.
    float pend_m = NEG, pend_v = xa0, pend_slo = NEG, pend_phi = NEG;   // of stream row t - 1
    unsigned long long pend_hm = 0;                                      // its hole mask (wave-uniform, lives in SGPRs)
    float nxt_c = xa0, nxt_a = xa0, nxt_b = xb0;                         // delay-line values for the next step

    // the fill front end of step t: returns X7 of image row t - 31 for every lane
    auto fill_step = [&](auto P_, int t) -> float {
        constexpr int p = decltype(P_)::value;
        const float xa = PFA[p], xb = PFB[p];
        {
            const int row = min(max(t + PFD - 15, 0), rows - 1);
            PFA[(p + PFD) & 15] = ld_a(row); PFB[(p + PFD) & 15] = ld_b(row);
        }
        // finish row t - 1 - 30 from what step t - 1 left pending
        const int o = t - 31;
        float x7 = pend_v;
        if (pend_hm != 0ull) {                                      // a row without holes passes through untouched
            const float d = fmax3(pend_m, pend_slo, pend_phi);
            const bool hole = __builtin_amdgcn_inverse_ballot_w64(pend_hm);   // pend_v < thr, LO :140
            x7 = hole ? d : pend_v;
            if ((unsigned)(o - so0) < (unsigned)(so1 - so0)) {      // hole counts on the scalar unit: ballot + s_bcnt1 (a band: its own rows)
                before += __builtin_popcountll(pend_hm & own_mask);
                after += __builtin_popcountll(__builtin_amdgcn_ballot_w64(x7 < thr) & own_mask);
            }
        }
        if (edge_strip) x7 = __shfl(x7, src_lane, 64);             // out-of-image columns replicate the edge column
        if (o >= rows) { asm volatile("" ::); x7 = x7_prev; }       // rows below the image replicate the last row (median border); the
                                                                    // empty asm keeps this a scalar branch (last 7 steps only) instead of a select per step
        x7_prev = x7;
        // vertical 31-max, both registers, in FOUR instructions per register (three-input maxima cost what two-input ones do):
        //   w2 = rows t-1..t;  w6 = max3(w2, w2[t-2], w2[t-4]) = rows t-5..t;  w18 = max3(w6, w6[t-6], w6[t-12]) = rows t-17..t;
        //   w31 = max(w18, w18[t-13]) = rows t-30..t          (k_fill_s: doubling 2, 4, 8, 16, +15 in five)
        const float w2a = fmax2(xa, vpa), w2b = fmax2(xb, vpb);
        vpa = xa; vpb = xb;
        W2A[p] = w2a; W2B[p] = w2b;
        const float w6a = fmax3(w2a, W2A[(p + 14) & 15], W2A[(p + 12) & 15]), w6b = fmax3(w2b, W2B[(p + 14) & 15], W2B[(p + 12) & 15]);
        W6A[p] = w6a; W6B[p] = w6b;
        const float w18a = fmax3(w6a, W6A[(p + 10) & 15], W6A[(p + 4) & 15]), w18b = fmax3(w6b, W6B[(p + 10) & 15], W6B[(p + 4) & 15]);
        // LDS delay lines: slot (t & 15) is written now.  Centre values are delayed by 15 steps: slot ((t + 1) & 15), written 15
        // steps ago, was fetched during the previous step, and slot ((t + 2) & 15) is fetched now for the next one.  The 18-row
        // maxima are delayed by 13: the slot fetched now for the next step is ((t + 1) - 13) & 15 = (t + 4) & 15.
        const float v = nxt_c, w18a_old = nxt_a, w18b_old = nxt_b;  // stream row t - 15; maxima of step t - 13
        nxt_c = dl_c[(p + 2) & 15][lane];
        nxt_a = dl_a[(p + 4) & 15][lane];
        nxt_b = dl_b[(p + 4) & 15][lb];
        dl_c[p][lane] = xa;
        dl_a[p][lane] = w18a;
        dl_b[p][lb] = w18b;
        const float w31a = fmax2(w18a, w18a_old), w31b = fmax2(w18b, w18b_old);
        // horizontal 31-max: scans now, the cross-row fetches land during the next step.  The fill only
        // replaces holes (x < thr), so a row in which none of this wave's 64 columns is a hole needs no
        // horizontal maximum at all (wave-uniform skip; the vertical state above is always kept current).
        const unsigned long long vm = __builtin_amdgcn_ballot_w64(v < thr);
        if (vm != 0ull) {
            float PA, SA, PB;
            row_scans3(w31a, w31b, PA, SA, PB);
            const float Sm = lane >= 49 ? PB : SA, Pm = lane <= 14 ? PB : PA;   // lane 48 of S' is still S_A(48), needed by c = 63
            pend_slo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a_lo, __builtin_bit_cast(int, Sm)));
            pend_phi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a_hi, __builtin_bit_cast(int, Pm)));
            pend_m = fmax2(PA, SA);
        }
        pend_v = v;
        pend_hm = vm;
        return x7;
    };

    // steps 0..31 (16..31 after a warm start): fill only.  The last of them returns X7 row 0, which the post pipeline takes three times
    // (its replicated rows -2 and -1, and row 0: post steps 0, 1, 2)
    for (int t0 = warm ? 16 : 0; t0 < 32; t0 += 16) {
        static_for<0, 16>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            const float x7 = fill_step(P_, t0 + p);
            if constexpr (p == 15) {
                if (t0 == 16) {
                    pipe.template step<0>(x7, 0);
                    pipe.template step<1>(x7, 1);
                    pipe.template step<2>(x7, 2);
                }
            }
        });
    }
    // steps 32..rows+34: fill + post; post step u = t - FpS::LAG takes X7 row u - 2 = t - 31, the row this step's fill front end returns
    const int nsteps = rows + 35;
    int t0 = 32;
    auto main_step = [&](auto P_) {
        constexpr int p = decltype(P_)::value;
        const int t = t0 + p, u = t - FpS::LAG;
        const float x7 = fill_step(P_, t);
        pipe.template step<((p + 3) & 7)>(x7, u);
        if constexpr (p == 3) {
            // u == 6: output row 0 of the shifted frame (image row V) has just been stored; the V rows above it are equal
            if (t0 == 32 && V > 0 && band == 0) {
                FrameBuf top;
                top.init(dst + fo, (size_t)V * cols);
                const unsigned tb = pipe.outlane ? pipe.ob : kDropOffset;
                for (int r = 0; r < V; ++r) top.st(tb, r, cols, pipe.last_out);
            }
        }
    };
    for (; t0 < nsteps; t0 += 16) {                                  // (four quarters with a way out behind each: see k_fp_q)
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
