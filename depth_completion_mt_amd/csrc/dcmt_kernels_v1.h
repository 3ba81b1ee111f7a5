// dcmt_kernels_v1.h -- "staged" gfx950 kernels: one launch per stage group, every stage
// staging its tile + halo in LDS.  These are the GENERAL path: any image size, any 5x5
// first element, the label-masked variant, per-stage dumps for parity probes and the
// hole-closure loop.  The fused fast path for the default configuration lives in
// dcmt_kernels_fused.h and falls back to these kernels frame by frame.
//
// Stage names H2..H11 follow SURVEY.md section 8a; LO = reference
// src/DC_lidar_only/img_completion.cpp, LC = src/DC_lidar_camera/img_completion_lc.cpp.
//
// Conventions shared by all kernels here:
//  * grid = (tiles_x, tiles_y, batch); 256 threads (4 wave64) per workgroup;
//  * an LDS "region" is the output tile grown by the stage group's total radius; region
//    coordinates (y,x) map to image coordinates (ty0+y, tx0+x);
//  * positions of a region that fall outside the image hold the border value of the
//    stage that will read them next (-FLT_MAX before a dilate, +FLT_MAX before an erode:
//    cv::dilate / cv::erode with BORDER_CONSTANT and the default border value), so the
//    window loops need no bounds tests;
//  * LDS pitches are odd so that row-wise and column-wise walks are both conflict-free
//    for ds_read_b32 (bank = dword address mod 32).
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "dcmt_median.h"

namespace dcmt {

constexpr int kThreads = 256;
constexpr int kMaxIters = 64;          // hard cap of dcmt_params.max_fill_iters
constexpr int kCntStride = kMaxIters + 4;  // ints per frame in the counter block

// counter block layout per frame (ints):
//  [0]            holes seen by H7 (= holes after the column extension)
//  [1 + k]        holes left after fill application k (k = 0 is H7 itself, k >= 1 the loop)
__device__ __forceinline__ int* frame_counters(int* counters, int f) { return counters + (size_t)f * kCntStride; }


// for every (y,x) in [y0,y1) x [x0,x1), consecutive threads on consecutive x
template <typename F>
__device__ __forceinline__ void for_rect(int y0, int y1, int x0, int x1, F f)
{
    const int w = x1 - x0, n = (y1 - y0) * w;
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const int y = i / w, x = i - y * w;
        f(y0 + y, x0 + x);
    }
}

__device__ __forceinline__ float invert_valid(float v, float max_depth, float thr)
{
    return v >= thr ? max_depth - v : v;     // LO :59-63 / :194-198
}

// ---------------------------------------------------------------------------------
// N1: per-frame min-max normalisation in front of the cascade -- what the stereo-lidar callers do with
// cv::normalize(projected, normalized, 0, 80 | 100, NORM_MINMAX) before they call the path
// (SL/main_sl.cpp:370, :523).  OpenCV semantics restated: smin/smax = the frame's extrema;
//   scale = (dmax - dmin) * (smax - smin > DBL_EPSILON ? 1 / (smax - smin) : 0)   in double,
//   for a CV_32F destination  scale = (float)scale,  shift = (float)dmin - (float)(smin * scale),
//   dst = src * (float)scale + (float)shift        (convertTo, f32 arithmetic, one rounding per op).
// With empty pixels (0) in the frame smin = 0, so shift = dmin and for dmin = 0 the result is the
// single rounding of src * scale: bit-exact whatever the host library fuses.
// k_minmax leaves per frame {max of key(x), max of ~key(x)} (key = order-preserving u32 image of
// the float), k_norm_coef turns them into the f32 pair (a, b) the first kernel of the chain applies.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_key(float v)
{
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unkey(uint32_t k)
{
    return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float norm_apply(float v, float a, float b) { return __fadd_rn(__fmul_rn(v, a), b); }

constexpr int kMinmaxUnits = 8;              // workgroups per frame

__global__ __launch_bounds__(256)
void k_minmax(const float* __restrict__ src, uint32_t* __restrict__ stats, size_t frame_elems, int batch, int xcd_map)
{
    int f, u;
    if (xcd_map) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; f = (slot / kMinmaxUnits) * 8 + xcd; u = slot % kMinmaxUnits; }
    else { f = blockIdx.x / kMinmaxUnits; u = blockIdx.x - f * kMinmaxUnits; }
    (void)batch;
    const float* s = src + (size_t)f * frame_elems;
    const size_t per = (frame_elems + kMinmaxUnits - 1) / kMinmaxUnits;
    const size_t b0 = (size_t)u * per, b1 = b0 + per < frame_elems ? b0 + per : frame_elems;
    float lo0 = FLT_MAX, lo1 = FLT_MAX, lo2 = FLT_MAX, lo3 = FLT_MAX, hi0 = -FLT_MAX, hi1 = -FLT_MAX, hi2 = -FLT_MAX, hi3 = -FLT_MAX;
    size_t i = b0 + threadIdx.x;
    for (; i + 768 < b1; i += 1024) {        // four independent coalesced dword loads in flight per thread
        const float v0 = s[i], v1 = s[i + 256], v2 = s[i + 512], v3 = s[i + 768];
        lo0 = fmin2(lo0, v0); hi0 = fmax2(hi0, v0); lo1 = fmin2(lo1, v1); hi1 = fmax2(hi1, v1);
        lo2 = fmin2(lo2, v2); hi2 = fmax2(hi2, v2); lo3 = fmin2(lo3, v3); hi3 = fmax2(hi3, v3);
    }
    for (; i < b1; i += 256) { const float v = s[i]; lo0 = fmin2(lo0, v); hi0 = fmax2(hi0, v); }
    float lo = fmin2(fmin2(lo0, lo1), fmin2(lo2, lo3)), hi = fmax2(fmax2(hi0, hi1), fmax2(hi2, hi3));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fmin2(lo, __shfl_xor(lo, o, 64)); hi = fmax2(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0 && b0 < b1) {
        atomicMax(&stats[2 * f], f32_key(hi));
        atomicMax(&stats[2 * f + 1], ~f32_key(lo));
    }
}

// (leaves the frame's two keys zeroed for the next call: no memset in the stream in front of k_minmax)
__global__ void k_norm_coef(uint32_t* __restrict__ stats, float* __restrict__ coef, int batch, float lo, float hi)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= batch) return;
    const double smax = (double)f32_unkey(stats[2 * f]), smin = (double)f32_unkey(~stats[2 * f + 1]);
    const double dmin = lo < hi ? (double)lo : (double)hi, dmax = lo < hi ? (double)hi : (double)lo;
    double scale = (dmax - dmin) * (smax - smin > 2.220446049250313e-16 ? 1.0 / (smax - smin) : 0.0);
    scale = (double)(float)scale;
    const double shift = (double)(float)dmin - (double)(float)(smin * scale);
    coef[2 * f] = (float)scale;
    coef[2 * f + 1] = (float)shift;
    stats[2 * f] = 0u; stats[2 * f + 1] = 0u;
}

// the normalised frames themselves (stop_after = DCMT_STAGE_NORMALIZE)
__global__ __launch_bounds__(256)
void k_norm_write(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ coef, size_t frame_elems, int batch)
{
    const size_t n = frame_elems * (size_t)batch;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / frame_elems;
        dst[i] = norm_apply(src[i], coef[2 * f], coef[2 * f + 1]);
    }
}

// ---------------------------------------------------------------------------------
// N2: LiDAR points -> sparse depth image (SL/main_sl.cpp:478-520).  Pass 1: one thread per point; the pixel's winner
// is the LAST point in file order = the largest point index (atomicMax on an int plane, -1 = empty).  Pass 2: one
// thread per pixel recomputes the winning point's depth (the same deterministic arithmetic) or writes 0.
// Every product and sum is rounded separately (__fmul_rn / __fadd_rn: no FMA contraction), sums left to right.
// ---------------------------------------------------------------------------------
struct ProjMats { float T[12]; float P[12]; };   // the three rows of T that are used; P

__device__ __forceinline__ float dot4_rn(const float* m, float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z)), m[3]);
}

// returns false if the point is dropped; otherwise pixel (u, v) and its depth
__device__ __forceinline__ bool project_point(const ProjMats& M, float x, float y, float z, int rows, int cols, int& u, int& v, float& depth)
{
    const float tx = dot4_rn(M.T, x, y, z), ty = dot4_rn(M.T + 4, x, y, z), tz = dot4_rn(M.T + 8, x, y, z);
    if (!(tz > 0.0f)) return false;                                   // SL :487
    const float px = dot4_rn(M.P, tx, ty, tz), py = dot4_rn(M.P + 4, tx, ty, tz), pz = dot4_rn(M.P + 8, tx, ty, tz);
    if (pz == 0.0f) return false;                                     // x/0 is +-inf or NaN: fails every bound below
    const float uf = __fdiv_rn(px, pz), vf = __fdiv_rn(py, pz);       // SL :502-503
    if (!(uf >= 0.0f && uf < (float)cols && vf >= 0.0f && vf < (float)rows)) return false;   // SL :506-507
    u = (int)uf; v = (int)vf; depth = pz;
    return true;
}

// depth of a point the first pass has already accepted: p.z of project_point, the same operations in the same order
__device__ __forceinline__ float point_depth(const ProjMats& M, float x, float y, float z)
{
    const float tx = dot4_rn(M.T, x, y, z), ty = dot4_rn(M.T + 4, x, y, z), tz = dot4_rn(M.T + 8, x, y, z);
    return dot4_rn(M.P + 8, tx, ty, tz);
}

// The winner plane holds TAGS: (generation << idx_bits) | global point index.  A call only looks at tags of its own generation, so
// the plane is not cleared between calls (one 0.48 GB memset per 256 sweeps less); the host clears it when the generations run out
// or a call needs more index bits than the plane's layout has.  Within a generation the larger tag is the larger point index =
// the later point in file order (a frame's points are contiguous), the reference's "last writer wins".
__global__ __launch_bounds__(256)
void k_project_scatter(const float* __restrict__ pts, const int* __restrict__ offsets, int n_points, int batch,
                       ProjMats M, unsigned* __restrict__ winner, int rows, int cols, unsigned gen_tag)
{
    // frame of the workgroup's first point: one search per workgroup (offsets[f] <= i < offsets[f+1]); a thread's own frame is
    // that one or, where sweeps end inside the workgroup's 256 points, a later one
    __shared__ int s_lo;
    const int i0 = blockIdx.x * 256;
    if (threadIdx.x == 0) {
        int lo = 0, hi = batch;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offsets[mid] <= i0) lo = mid; else hi = mid; }
        s_lo = lo;
    }
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i >= n_points) return;
    int lo = s_lo;
    while (lo + 1 < batch && offsets[lo + 1] <= i) ++lo;
    const float4 p = *reinterpret_cast<const float4*>(pts + 4 * (size_t)i);       // x, y, z, reflectance (16-byte records)
    int u, v; float d;
    if (!project_point(M, p.x, p.y, p.z, rows, cols, u, v, d)) return;
    atomicMax(&winner[((size_t)lo * rows + v) * cols + u], gen_tag | (unsigned)i);
}

// One thread per PW neighbouring pixels of the whole batch (the tags hold global point indices, so a thread's pixels may lie in two
// frames): 8- / 16-byte accesses where the batch size and the buffers allow.  Most pixels have no point of this generation and are a
// zero; the winners' depths are recomputed from their points (the depth alone: the pixel is known, so the two divisions and the bounds
// test of the first pass are not repeated).
template <int PW>
__global__ __launch_bounds__(256)
void k_project_resolve(const float* __restrict__ pts, ProjMats M, const unsigned* __restrict__ winner, float* __restrict__ sparse,
                       size_t n_px, unsigned gen_tag, int idx_bits)
{
    const size_t i = (blockIdx.x * (size_t)256 + threadIdx.x) * PW;
    if (i >= n_px) return;
    unsigned w[PW];
    if constexpr (PW == 4) { const uint4 ww = *reinterpret_cast<const uint4*>(winner + i); w[0] = ww.x; w[1] = ww.y; w[2] = ww.z; w[3] = ww.w; }
    else if constexpr (PW == 2) { const uint2 ww = *reinterpret_cast<const uint2*>(winner + i); w[0] = ww.x; w[1] = ww.y; }
    else w[0] = winner[i];
    const unsigned mask = (1u << idx_bits) - 1u;
    float o[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        o[k] = 0.0f;
        if ((w[k] & ~mask) == gen_tag) {
            const float4 p = *reinterpret_cast<const float4*>(pts + 4 * (size_t)(w[k] & mask));
            o[k] = point_depth(M, p.x, p.y, p.z);
        }
    }
    if constexpr (PW == 4) *reinterpret_cast<float4*>(sparse + i) = make_float4(o[0], o[1], o[2], o[3]);
    else if constexpr (PW == 2) *reinterpret_cast<float2*>(sparse + i) = make_float2(o[0], o[1]);
    else sparse[i] = o[0];
}

// ---------------------------------------------------------------------------------
// N4: the stereo photometric refinement behind the path (SL/main_sl.cpp:715-885, driven from :1165-1246).  One thread
// per pixel runs all sweeps for its own disparity; the images are read as bytes and the EntryType fields (value,
// d/dx) are formed on the fly: value = the grey byte, derivative.x = .5 * right - .5 * left neighbour for interior
// pixels (rows 1..R-2, cols 1..C-2), 0 on the border (:719-745; byte-valued, so exact in f32).
// calculateObservationDerivatives (:749-801) as called from optimize_IG: r = i exactly, so r0 = i, dr = 0, dr1 = 1 and
// the lower row of the bilinear patch carries weight 0 -- only row i matters; c0 = (int)(c + 0.5) (a double sum,
// truncated), accepted when 0 <= c0 and c0 + 1 <= cols, and `at<>(r0, cols)` for c0 + 1 == cols is the next element in
// memory, i.e. the first pixel of the next row (past the buffer for the last row: that one pixel is left as it is).
// f32 arithmetic, one rounding per operation, in the reference's order.
// ---------------------------------------------------------------------------------
struct StereoP { float baseline, focal, damp, max_depth; int iterations; };

__device__ __forceinline__ float grey_dx(const uint8_t* g, int r, int c, int rows, int cols)
{
    if (r < 1 || r >= rows - 1 || c < 1 || c >= cols - 1) return 0.0f;
    return __fsub_rn(__fmul_rn(0.5f, (float)g[(size_t)r * cols + c + 1]), __fmul_rn(0.5f, (float)g[(size_t)r * cols + c - 1]));
}

// Grid: (1, rows, batch) with the row in LDS -- one workgroup stages the right image's row once and its threads walk the whole
// row, 256 columns at a time (five workgroups per 1242-pixel row, each staging the row, took 1.12 ms instead of 0.91) --
// or (ceil(cols / 256), rows, batch) without; no index divisions.  Per Gauss-Newton sweep the reference
// touches six right-image bytes (p00, p01 and the central differences at both); p01 is the neighbour of p00 in memory
// (e1 = e0 + 1, also across the end of a row, which the reference's at<>() reads too), so the six are four distinct bytes:
// g[e0-1], g[e0], g[e1], g[e1+1].  All arithmetic as in the oracle: one rounding per operation, IEEE divisions where the
// reference divides (initial disparity, every sweep's step, final depth).
// LDS_ROW: the workgroup first stages the whole right-image row (and the first pixel of the next one) in LDS: a sweep's
// bytes then come from ds_read_u8 instead of four byte gathers through the texture path, which is what bound the first
// version (1.42 ms per 256 pairs of 1242x375 with ~250 VALU instructions per pixel: the CU's address unit takes a 64-lane byte
// load at a few lanes per cycle).  Dynamic LDS: cols + 4 bytes.
template <bool LDS_ROW>
__global__ __launch_bounds__(256)
void k_stereo_refine(const float* __restrict__ depth, const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                     float* __restrict__ out, int rows, int cols, int batch, StereoP P)
{
    extern __shared__ uint8_t s_row[];             // LDS_ROW: [0] pad, [1 .. n] the row (and the next row's first pixel), two pads
    const int j0 = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if constexpr (LDS_ROW) {
        const uint8_t* g = right + (size_t)blockIdx.z * rows * cols + (size_t)i * cols;
        const int n = i + 1 < rows ? cols + 1 : cols;
        for (int k = threadIdx.x; k < cols + 4; k += 256) s_row[k] = (k >= 1 && k <= n) ? g[k - 1] : (uint8_t)0;
        __syncthreads();
    }
    const size_t fe = (size_t)rows * cols;
    const int fei = (int)fe;                                                    // a frame has < 2^29 pixels (dcmt_create)
    const float bf = __fmul_rn(P.baseline, P.focal);
    const uint8_t* gr = right + (size_t)blockIdx.z * fe;
    // the grid's x extent times 256 threads may cover only part of the row: a thread takes every (gridDim.x * 256)-th column
    for (int j = j0; j < cols; j += gridDim.x * 256) {
    const size_t idx = (size_t)blockIdx.z * fe + (size_t)i * cols + j;
    const float d0 = depth[idx];
    float disp = d0 > 0.0f ? __fdiv_rn(bf, d0) : 0.0f;                          // get_initial_disparity :852-856
    const float lv = (float)left[idx];
    const bool row_in = i >= 1 && i < rows - 1;
    for (int k = 0; k < P.iterations; ++k) {                                    // optimize_IG :809-841
        const float c = __fsub_rn((float)j, disp);
        const int c0 = (int)((double)c + 0.5);
        const int e0 = i * cols + c0, e1 = e0 + 1;                              // p00, p01 (p01 may be the next row's first pixel)
        // the sweep leaves the disparity alone where the reference `continue`s; everything below is computed regardless, from a
        // clamped column (no branch, one LDS round trip per sweep instead of three)
        const bool live = !(c0 < 0 || c0 + 1 > cols || disp == 0.0f) && e1 < fei;
        const int cc = min(max(c0, 0), cols - 1);
        const bool wrap = cc + 1 == cols;                                       // p01 = (i + 1, 0)
        float gm, g0, g1, gp;                                                   // g[e0-1], g[e0], g[e1], g[e1+1]
        if constexpr (LDS_ROW) {
            unsigned w;
            __builtin_memcpy(&w, s_row + cc, 4);                                // bytes cc-1 .. cc+2 of the row (pad in front)
            gm = (float)(w & 0xffu); g0 = (float)((w >> 8) & 0xffu); g1 = (float)((w >> 16) & 0xffu); gp = (float)(w >> 24);
        } else {
            const int b0 = i * cols + cc;
            g0 = (float)gr[b0];
            gm = (float)gr[max(b0 - 1, 0)];
            g1 = (float)gr[min(b0 + 1, fei - 1)];
            gp = (float)gr[min(b0 + 2, fei - 1)];
        }
        // central differences (calculateMeasuementDerivatives :715-745): 0 on the image border
        const float dx0 = (row_in && cc >= 1 && cc < cols - 1) ? __fsub_rn(__fmul_rn(0.5f, g1), __fmul_rn(0.5f, gm)) : 0.0f;
        const float dx1 = (!wrap && row_in && cc + 1 < cols - 1) ? __fsub_rn(__fmul_rn(0.5f, gp), __fmul_rn(0.5f, g0)) : 0.0f;   // (a wrapped p01 sits in column 0: border, 0)
        const float dc = __fsub_rn(c, (float)c0), dc1 = __fsub_rn(1.0f, dc);
        const float value = __fadd_rn(__fmul_rn(g0, dc1), __fmul_rn(g1, dc));
        const float dx = __fadd_rn(__fmul_rn(dx0, dc1), __fmul_rn(dx1, dc));
        float error = __fsub_rn(value, lv);                                     // :819
        error = error > 255.0f ? 255.0f : error;
        error = error < -255.0f ? -255.0f : error;
        const float jcr = __fmul_rn(-1.0f, dx);                                 // J = -1 (:830-832)
        const float H = __fadd_rn(__fmul_rn(jcr, jcr), P.damp);
        const float b = __fmul_rn(jcr, error);
        const float next = __fadd_rn(disp, __fdiv_rn(-b, H));                   // :836-837
        disp = live ? next : disp;
    }
    float o = 0.0f;                                                             // retrieve_optimized_depth :868-880
    if (disp > 0.0f) { o = __fdiv_rn(bf, disp); if (o > P.max_depth) o = P.max_depth; }
    out[idx] = o;
    }
}

// ---------------------------------------------------------------------------------
// Per-call state of the staged path, without an init launch (a dependent launch costs ~5 us, as much as a quarter of a
// stage on a single frame): the column statistics are kept per TILE ROW -- colstat[frame][tile row][2][cols], every entry
// written exactly once by the workgroup that owns that tile, no atomics across workgroups -- and reduced over the tile rows
// by the reader (k_fill31_v1); the hole counters of a frame are zeroed by the first workgroup of the first kernel.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void zero_frame_counters(int* __restrict__ counters, int f)
{
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < kCntStride; i += kThreads) counters[(size_t)f * kCntStride + i] = 0;
}

// ---------------------------------------------------------------------------------
// k_pre_v1: H2 invert, H3 dilate(k0), H4 close 5x5, H5 7x7 small fill  ->  X5,
//           plus per-column first/last valid row of X5 (input of H6).
// Total radius 2 + 2 + 2 + 3 = 9.
// ---------------------------------------------------------------------------------
template <int TH, int TW>
struct PreGeom {
    static constexpr int R = 9, RH = TH + 2 * R, RW = TW + 2 * R, P = RW | 1;
};

// Shared tail of k_pre_v1 / k_pre_labeled_v1: B holds X4 on [6,RH-6)x[6,RW-6) with
// -FLT_MAX outside the image; A is scratch.  Computes X5 on the tile and the column stats.
template <int TH, int TW>
__device__ __forceinline__ void small_fill_and_stats(float* A, float* B, int* smin, int* smax,
                                                     float* __restrict__ x5f, int* __restrict__ colstat_f,
                                                     int rows, int cols, int ty0, int tx0, float thr)
{
    using G = PreGeom<TH, TW>;
    constexpr int R = G::R, RH = G::RH, RW = G::RW, P = G::P;
    // H5 (LO :88-100): s = dilate7(x); x = x < 0.1 ? s : x
    for_rect(6, RH - 6, R, RW - R, [&](int y, int x) {
        const float* b = B + y * P + x;
        float m = fmax2(fmax2(b[-3], b[-2]), fmax2(b[-1], b[0]));
        m = fmax2(m, fmax2(fmax2(b[1], b[2]), b[3]));
        A[y * P + x] = m;
    });
    for (int i = threadIdx.x; i < TW; i += kThreads) { smin[i] = 0x7fffffff; smax[i] = -1; }
    __syncthreads();
    for_rect(R, RH - R, R, RW - R, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        if (gy >= rows || gx >= cols) return;     // tile origin is inside the image: gy,gx >= 0 here
        const float* a = A + y * P + x;
        float s = fmax2(fmax2(a[-3 * P], a[-2 * P]), fmax2(a[-P], a[0]));
        s = fmax2(s, fmax2(fmax2(a[P], a[2 * P]), a[3 * P]));
        const float x4 = B[y * P + x];
        const float v = x4 < thr ? s : x4;
        x5f[(size_t)gy * cols + gx] = v;
        if (v >= thr) {                           // LO :113,:117 `> 0.1`
            atomicMin(&smin[x - R], gy);
            atomicMax(&smax[x - R], gy);
        }
    });
    __syncthreads();
    for (int i = threadIdx.x; i < TW; i += kThreads) {              // this tile row's slot: first / last valid row, or none
        const int gx = tx0 + R + i;
        if (gx < cols) { colstat_f[gx] = smin[i]; colstat_f[cols + gx] = smax[i]; }
    }
}

template <int TH, int TW>
__device__ __forceinline__ void dump_plane(const float* plane, float* __restrict__ dstf,
                                           int rows, int cols, int ty0, int tx0)
{
    using G = PreGeom<TH, TW>;
    for_rect(G::R, G::RH - G::R, G::R, G::RW - G::R, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        if (gy < rows && gx < cols) dstf[(size_t)gy * cols + gx] = plane[y * G::P + x];
    });
}

template <int TH, int TW>
__global__ __launch_bounds__(kThreads)
void k_pre_v1(const float* __restrict__ src, float* __restrict__ x5, int* __restrict__ colstat, int* __restrict__ counters,
              float* __restrict__ dump, int rows, int cols, float max_depth, float thr,
              uint32_t k0bits, int dump_stage, const float* __restrict__ coef)
{
    zero_frame_counters(counters, blockIdx.z);
    using G = PreGeom<TH, TW>;
    constexpr int R = G::R, RH = G::RH, RW = G::RW, P = G::P;
    __shared__ float A[RH * P];
    __shared__ float B[RH * P];
    __shared__ int smin[TW], smax[TW];

    const int f = blockIdx.z;
    const int ty0 = blockIdx.y * TH - R, tx0 = blockIdx.x * TW - R;
    const size_t fo = (size_t)f * rows * cols;
    const float* s = src + fo;
    const float na = coef ? coef[2 * f] : 1.0f, nb = coef ? coef[2 * f + 1] : 0.0f;   // N1 normalisation, if any

    // H2 (LO :55-67) on load; outside the image: border value of the dilate that follows
    for_rect(0, RH, 0, RW, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        float v = -FLT_MAX;
        if (gy >= 0 && gy < rows && gx >= 0 && gx < cols) {
            v = s[(size_t)gy * cols + gx];
            if (coef) v = norm_apply(v, na, nb);
            v = invert_valid(v, max_depth, thr);
        }
        A[y * P + x] = v;
    });
    __syncthreads();
    if (dump_stage == 2) { dump_plane<TH, TW>(A, dump + fo, rows, cols, ty0, tx0); return; }

    // H3 (LO :71-80): dst(p) = max over non-zero taps k of src(p + k - anchor)
    for_rect(2, RH - 2, 2, RW - 2, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        float m = -FLT_MAX;
#pragma unroll
        for (int t = 0; t < 25; ++t)
            if ((k0bits >> t) & 1u) m = fmax2(m, A[(y + t / 5 - 2) * P + x + t % 5 - 2]);
        const bool in = gy >= 0 && gy < rows && gx >= 0 && gx < cols;
        B[y * P + x] = in ? m : -FLT_MAX;
    });
    __syncthreads();
    if (dump_stage == 3) { dump_plane<TH, TW>(B, dump + fo, rows, cols, ty0, tx0); return; }

    // H4 (LO :84-85) MORPH_CLOSE = dilate 5x5 then erode 5x5, each as row pass + column pass
    for_rect(2, RH - 2, 4, RW - 4, [&](int y, int x) {
        const float* b = B + y * P + x;
        A[y * P + x] = fmax2(fmax2(fmax2(b[-2], b[-1]), fmax2(b[0], b[1])), b[2]);
    });
    __syncthreads();
    for_rect(4, RH - 4, 4, RW - 4, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        const float* a = A + y * P + x;
        const float m = fmax2(fmax2(fmax2(a[-2 * P], a[-P]), fmax2(a[0], a[P])), a[2 * P]);
        const bool in = gy >= 0 && gy < rows && gx >= 0 && gx < cols;
        B[y * P + x] = in ? m : FLT_MAX;          // border value of the erode
    });
    __syncthreads();
    for_rect(4, RH - 4, 6, RW - 6, [&](int y, int x) {
        const float* b = B + y * P + x;
        A[y * P + x] = fmin2(fmin2(fmin2(b[-2], b[-1]), fmin2(b[0], b[1])), b[2]);
    });
    __syncthreads();
    for_rect(6, RH - 6, 6, RW - 6, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        const float* a = A + y * P + x;
        const float m = fmin2(fmin2(fmin2(a[-2 * P], a[-P]), fmin2(a[0], a[P])), a[2 * P]);
        const bool in = gy >= 0 && gy < rows && gx >= 0 && gx < cols;
        B[y * P + x] = in ? m : -FLT_MAX;         // border value of the 7x7 dilate
    });
    __syncthreads();
    if (dump_stage == 4) { dump_plane<TH, TW>(B, dump + fo, rows, cols, ty0, tx0); return; }

    small_fill_and_stats<TH, TW>(A, B, smin, smax, x5 + fo, colstat + ((size_t)f * gridDim.y + blockIdx.y) * 2 * cols, rows, cols, ty0, tx0, thr);
}

// ---------------------------------------------------------------------------------
// k_pre_labeled_v1: LC variant.  H2, then for every label c present near the tile
//   region = x * [label == c]  (other in-image pixels 0, LC :94-95)
//   region = erode5(dilate5(dilate_k0(region)))   (LC :97-100)
//   x[label == c] = region[label == c]            (LC :101)
// then H5 and the column statistics exactly as k_pre_v1.  Labels are disjoint and each
// write-back only depends on the pre-loop values of its own label, so the label order is
// irrelevant and every tile can process just the labels it contains.
// ---------------------------------------------------------------------------------
template <int TH, int TW>
__global__ __launch_bounds__(kThreads)
void k_pre_labeled_v1(const float* __restrict__ src, const int32_t* __restrict__ labels, int n_labels,
                      float* __restrict__ x5, int* __restrict__ colstat, int* __restrict__ counters, float* __restrict__ dump,
                      int rows, int cols, float max_depth, float thr, uint32_t k0bits, int dump_stage,
                      const float* __restrict__ coef)
{
    zero_frame_counters(counters, blockIdx.z);
    using G = PreGeom<TH, TW>;
    constexpr int R = G::R, RH = G::RH, RW = G::RW, P = G::P;
    __shared__ float X0[RH * P];     // H2 output (never modified)
    __shared__ int   L[RH * P];      // labels; -2 outside the image
    __shared__ float A[RH * P];
    __shared__ float B[RH * P];
    __shared__ float X4[RH * P];     // result of the masked stage, needed on [6,RH-6)x[6,RW-6)
    __shared__ int smin[TW], smax[TW];
    __shared__ int s_next;
    __shared__ int s_box[4];

    const int f = blockIdx.z;
    const int ty0 = blockIdx.y * TH - R, tx0 = blockIdx.x * TW - R;
    const size_t fo = (size_t)f * rows * cols;
    const float* s = src + fo;
    const int32_t* lab = labels + fo;
    const float na = coef ? coef[2 * f] : 1.0f, nb = coef ? coef[2 * f + 1] : 0.0f;   // N1 normalisation, if any

    for_rect(0, RH, 0, RW, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        float v = -FLT_MAX;
        int l = -2;
        if (gy >= 0 && gy < rows && gx >= 0 && gx < cols) {
            v = s[(size_t)gy * cols + gx];
            if (coef) v = norm_apply(v, na, nb);
            v = invert_valid(v, max_depth, thr);
            l = lab[(size_t)gy * cols + gx];
            if (l < 0 || l >= n_labels) l = -1;     // never matched by the loop `c in [0,n)` (LC :78)
        }
        X0[y * P + x] = v;
        L[y * P + x] = l;
        X4[y * P + x] = v;                          // unlabeled pixels keep the H2 value
    });
    __syncthreads();
    if (dump_stage == 2) { dump_plane<TH, TW>(X0, dump + fo, rows, cols, ty0, tx0); return; }

    int cur = -1;
    for (;;) {
        // next label > cur present on [6,RH-6)x[6,RW-6)
        if (threadIdx.x == 0) s_next = 0x7fffffff;
        __syncthreads();
        int mine = 0x7fffffff;
        for_rect(6, RH - 6, 6, RW - 6, [&](int y, int x) {
            const int l = L[y * P + x];
            if (l > cur && l < mine) mine = l;
        });
        if (mine != 0x7fffffff) atomicMin(&s_next, mine);
        __syncthreads();
        cur = s_next;
        if (cur == 0x7fffffff) break;

        // bounding box of this label inside the needed area: every pass below only has to cover the
        // box grown by what the later passes still need (superpixels are compact: ~4x less work)
        if (threadIdx.x == 0) { s_box[0] = RH; s_box[1] = -1; s_box[2] = RW; s_box[3] = -1; }
        __syncthreads();
        {
            int y0 = RH, y1 = -1, x0 = RW, x1 = -1;
            for_rect(6, RH - 6, 6, RW - 6, [&](int y, int x) {
                if (L[y * P + x] != cur) return;
                y0 = min(y0, y); y1 = max(y1, y); x0 = min(x0, x); x1 = max(x1, x);
            });
            if (y1 >= 0) { atomicMin(&s_box[0], y0); atomicMax(&s_box[1], y1); atomicMin(&s_box[2], x0); atomicMax(&s_box[3], x1); }
        }
        __syncthreads();
        const int by0 = s_box[0], by1 = s_box[1] + 1, bx0 = s_box[2], bx1 = s_box[3] + 1;   // [by0,by1) x [bx0,bx1), inside [6,RH-6)x[6,RW-6)

        // masked copy: label pixels keep their value, other in-image pixels are 0,
        // outside the image the dilate border value
        for_rect(by0 - 6, by1 + 6, bx0 - 6, bx1 + 6, [&](int y, int x) {
            const int l = L[y * P + x];
            A[y * P + x] = l == -2 ? -FLT_MAX : (l == cur ? X0[y * P + x] : 0.0f);
        });
        __syncthreads();
        for_rect(by0 - 4, by1 + 4, bx0 - 4, bx1 + 4, [&](int y, int x) {
            float m = -FLT_MAX;
#pragma unroll
            for (int t = 0; t < 25; ++t)
                if ((k0bits >> t) & 1u) m = fmax2(m, A[(y + t / 5 - 2) * P + x + t % 5 - 2]);
            B[y * P + x] = L[y * P + x] == -2 ? -FLT_MAX : m;
        });
        __syncthreads();
        for_rect(by0 - 4, by1 + 4, bx0 - 2, bx1 + 2, [&](int y, int x) {
            const float* b = B + y * P + x;
            A[y * P + x] = fmax2(fmax2(fmax2(b[-2], b[-1]), fmax2(b[0], b[1])), b[2]);
        });
        __syncthreads();
        for_rect(by0 - 2, by1 + 2, bx0 - 2, bx1 + 2, [&](int y, int x) {
            const float* a = A + y * P + x;
            const float m = fmax2(fmax2(fmax2(a[-2 * P], a[-P]), fmax2(a[0], a[P])), a[2 * P]);
            B[y * P + x] = L[y * P + x] == -2 ? FLT_MAX : m;
        });
        __syncthreads();
        for_rect(by0 - 2, by1 + 2, bx0, bx1, [&](int y, int x) {
            const float* b = B + y * P + x;
            A[y * P + x] = fmin2(fmin2(fmin2(b[-2], b[-1]), fmin2(b[0], b[1])), b[2]);
        });
        __syncthreads();
        for_rect(by0, by1, bx0, bx1, [&](int y, int x) {
            if (L[y * P + x] != cur) return;
            const float* a = A + y * P + x;
            X4[y * P + x] = fmin2(fmin2(fmin2(a[-2 * P], a[-P]), fmin2(a[0], a[P])), a[2 * P]);
        });
        __syncthreads();
    }
    // X4 outside the image must be the border value of the 7x7 dilate; it is (-FLT_MAX from the load)
    if (dump_stage == 3 || dump_stage == 4) { dump_plane<TH, TW>(X4, dump + fo, rows, cols, ty0, tx0); return; }

    small_fill_and_stats<TH, TW>(A, X4, smin, smax, x5 + fo, colstat + ((size_t)f * gridDim.y + blockIdx.y) * 2 * cols, rows, cols, ty0, tx0, thr);
}

// ---------------------------------------------------------------------------------
// k_fill31_v1: one application of "s = dilate31(x); x = x < 0.1 ? s : x".
//   app == 0 : H6 + H7  -- the input is X5 with the column extension (LO :103-129) applied
//              on load from the per-column first/last valid rows; counts the holes it sees.
//   app >= 1 : one iteration of the H8 loop (LO :146-166); skipped for frames whose
//              previous application left no holes (an application without holes changes
//              nothing, so skipping it is exact).
// The 31-wide max is built by doubling: windows of 2, 4, 8, 16, then 16 + 16 overlapping.
// ---------------------------------------------------------------------------------
template <int TH, int TW>
struct FillGeom {
    static constexpr int R = 15, RH = TH + 2 * R, RW = TW + 2 * R, P = RW | 1;
};

template <int TH, int TW>
__global__ __launch_bounds__(kThreads)
void k_fill31_v1(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ colstat,
                 int* __restrict__ counters, int rows, int cols, float thr, int app, int dump_extend, int stat_rows)
{
    // stat_rows: tile rows of the kernel that wrote colstat (its tiles may be taller than this kernel's)
    using G = FillGeom<TH, TW>;
    constexpr int R = G::R, RH = G::RH, RW = G::RW, P = G::P;
    __shared__ float A[RH * P];
    __shared__ float B[RH * P];
    __shared__ float C[TH * TW];                 // the tile's own input values
    __shared__ int   cti[RW], cbi[RW];
    __shared__ float ctv[RW], cbv[RW];
    __shared__ int   s_before, s_after;

    const int f = blockIdx.z;
    int* cnt = frame_counters(counters, f);
    if (app >= 1 && cnt[1 + app - 1] == 0) return;      // uniform per workgroup

    const int ty0 = blockIdx.y * TH - R, tx0 = blockIdx.x * TW - R;
    const size_t fo = (size_t)f * rows * cols;
    const float* xin = in + fo;

    if (threadIdx.x == 0) { s_before = 0; s_after = 0; }
    if (app == 0) {
        const int* cs = colstat + (size_t)f * stat_rows * 2 * cols;      // [tile row][2][cols]
        for (int x = threadIdx.x; x < RW; x += kThreads) {
            const int gx = tx0 + x;
            int ti = 0, bi = -1; float tv = 0.f, bv = 0.f;
            if (gx >= 0 && gx < cols) {
                ti = 0x7fffffff;
                for (int tr = 0; tr < stat_rows; ++tr) {
                    ti = min(ti, cs[(size_t)tr * 2 * cols + gx]);
                    bi = max(bi, cs[(size_t)tr * 2 * cols + cols + gx]);
                }
                if (bi >= 0) { tv = xin[(size_t)ti * cols + gx]; bv = xin[(size_t)bi * cols + gx]; }
            }
            cti[x] = ti; cbi[x] = bi; ctv[x] = tv; cbv[x] = bv;
        }
    }
    __syncthreads();
    for_rect(0, RH, 0, RW, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        float v = -FLT_MAX;
        if (gy >= 0 && gy < rows && gx >= 0 && gx < cols) {
            v = xin[(size_t)gy * cols + gx];
            if (app == 0) {
                // LO :122-127: rows >= last valid take its value, then rows <= first valid take
                // its value; a column without valid pixels ends as 100 everywhere (:110,:125-127)
                const int bi = cbi[x];
                if (bi < 0) v = 100.0f;
                else if (gy <= cti[x]) v = ctv[x];
                else if (gy >= bi) v = cbv[x];
            }
        }
        A[y * P + x] = v;
        if (y >= R && y < RH - R && x >= R && x < RW - R) C[(y - R) * TW + (x - R)] = v;
    });
    __syncthreads();
    if (dump_extend) {
        for_rect(0, TH, 0, TW, [&](int y, int x) {
            const int gy = ty0 + R + y, gx = tx0 + R + x;
            if (gy < rows && gx < cols) out[fo + (size_t)gy * cols + gx] = C[y * TW + x];
        });
        return;
    }

    // row phase: A -> B -> A -> B -> A -> B ; B[y][x] = max A0[y][x .. x+30]
    float* s0 = A; float* s1 = B;
    {
        const int steps[5] = {1, 2, 4, 8, 15};
        int win = 1;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int st = steps[k];
            win += st;                            // 2,4,8,16,31
            for_rect(0, RH, 0, RW - win + 1, [&](int y, int x) {
                s1[y * P + x] = fmax2(s0[y * P + x], s0[y * P + x + st]);
            });
            __syncthreads();
            float* t = s0; s0 = s1; s1 = t;
        }
        // column phase on the TW tile columns (x in [0,TW) after the row phase)
        win = 1;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int st = steps[k];
            win += st;
            for_rect(0, RH - win + 1, 0, TW, [&](int y, int x) {
                s1[y * P + x] = fmax2(s0[y * P + x], s0[(y + st) * P + x]);
            });
            __syncthreads();
            float* t = s0; s0 = s1; s1 = t;
        }
    }
    // s0[y][x], y < TH, x < TW = dilate31 at tile pixel (y,x)
    int before = 0, after = 0;
    for_rect(0, TH, 0, TW, [&](int y, int x) {
        const int gy = ty0 + R + y, gx = tx0 + R + x;
        if (gy >= rows || gx >= cols) return;
        const float v = C[y * TW + x];
        const bool hole = v < thr;                // LO :140 / :154
        const float o = hole ? s0[y * P + x] : v;
        out[fo + (size_t)gy * cols + gx] = o;
        before += hole;
        after += o < thr;
    });
    if (before) atomicAdd(&s_before, before);
    if (after) atomicAdd(&s_after, after);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (app == 0 && s_before) atomicAdd(&cnt[0], s_before);
        if (s_after) atomicAdd(&cnt[1 + app], s_after);
    }
}

// ---------------------------------------------------------------------------------
// k_post_v1: H9 median 5x5 (BORDER_REPLICATE), H10 Gaussian [1 4 6 4 1]/16 separable
//            (BORDER_REFLECT_101) + masked select, H11 invert.  Reads the buffer that holds
//            the frame after its last fill application.
// mode: 8 = copy only (probe of H8), 9 = stop after the median, 10 = after the blur, 11 = all
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// which ping-pong buffer holds frame f after the loop applications that actually ran
__device__ __forceinline__ int apps_done(const int* cnt, int n_apps_launched)
{
    int a = 0;
    while (a < n_apps_launched && cnt[1 + a] > 0) ++a;   // application a+1 ran iff application a left holes
    return a;
}

template <int TH, int TW>
__global__ __launch_bounds__(kThreads)
void k_post_v1(const float* __restrict__ pp0, const float* __restrict__ pp1, float* __restrict__ dst,
               const int* __restrict__ counters, int n_apps_launched, int rows, int cols,
               float max_depth, float thr, int blur, int mode)
{
    constexpr int R = 4, RH = TH + 2 * R, RW = TW + 2 * R, P = RW | 1;
    __shared__ float A[RH * P];          // x, replicate-padded
    __shared__ float B[RH * P];          // median
    __shared__ float C[(TH + 4) * TW];   // horizontal Gaussian on rows tile-2 .. tile+2

    const int f = blockIdx.z;
    const int a = apps_done(counters + (size_t)f * kCntStride, n_apps_launched);
    const size_t fo = (size_t)f * rows * cols;
    const float* xin = ((a & 1) ? pp1 : pp0) + fo;
    float* o = dst + fo;
    const int ty0 = blockIdx.y * TH - R, tx0 = blockIdx.x * TW - R;

    if (mode == 8) {
        for_rect(0, TH, 0, TW, [&](int y, int x) {
            const int gy = ty0 + R + y, gx = tx0 + R + x;
            if (gy < rows && gx < cols) o[(size_t)gy * cols + gx] = xin[(size_t)gy * cols + gx];
        });
        return;
    }
    for_rect(0, RH, 0, RW, [&](int y, int x) {
        const int gy = min(max(ty0 + y, 0), rows - 1), gx = min(max(tx0 + x, 0), cols - 1);
        A[y * P + x] = xin[(size_t)gy * cols + gx];
    });
    __syncthreads();
    // H9 on tile +- 2: one thread per (column, segment of kSeg rows) streams down its rows with the time-shared
    // networks of dcmt_median.h -- 16 row sorts and 8 pair merges for 12 medians instead of 12 full 25-input networks
    // (~50 instead of ~230 min/max per pixel, 80 instead of 300 LDS reads).  A is replicate-padded, so every position of
    // the region has its full window; positions outside the image are computed and never read.
    {
        constexpr int kSeg = 12, kCols = RW - 4, kSegs = (RH - 4 + kSeg - 1) / kSeg;
        for (int t = threadIdx.x; t < kCols * kSegs; t += kThreads) {
            const int x = 2 + t % kCols, y0 = 2 + (t / kCols) * kSeg;           // medians of rows y0 .. y0 + kSeg - 1
            MedianColumn mc;
            mc.init();
            static_for<0, kSeg + 4>([&](auto U_) {
                constexpr int u = decltype(U_)::value;
                const int yi = min(y0 - 2 + u, RH - 1);                          // input row (clamped only in a short last segment)
                const float* a = A + yi * P + x;
                float s5[5] = {a[-2], a[-1], a[0], a[1], a[2]};
                sort5(s5);
                const float m = mc.template step<(u & 7)>(s5);
                if constexpr (u >= 4) {
                    const int yo = y0 + u - 4;
                    if (yo < RH - 2) B[yo * P + x] = m;
                }
            });
        }
    }
    __syncthreads();
    const bool do_blur = blur == 1 && mode >= 10;
    if (do_blur) {
        // horizontal pass for the in-image rows of tile +- 2 (LO :179); reflect-101 in image coordinates
        for_rect(2, RH - 2, R, RW - R, [&](int y, int x) {
            const int gy = ty0 + y, gx = tx0 + x;
            if (gy < 0 || gy >= rows || gx >= cols) return;
            const float* b = B + y * P;
            const float c0 = b[x];
            const float l1 = b[reflect101(gx - 1, cols) - tx0], r1 = b[reflect101(gx + 1, cols) - tx0];
            const float l2 = b[reflect101(gx - 2, cols) - tx0], r2 = b[reflect101(gx + 2, cols) - tx0];
            float acc = __fmul_rn(c0, 0.375f);
            acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(l1, r1), 0.25f));
            acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(l2, r2), 0.0625f));
            C[(y - 2) * TW + (x - R)] = acc;
        });
        __syncthreads();
    }
    for_rect(R, RH - R, R, RW - R, [&](int y, int x) {
        const int gy = ty0 + y, gx = tx0 + x;
        if (gy >= rows || gx >= cols) return;
        const float m = B[y * P + x];
        float v = m;
        if (do_blur) {
            const int cx = x - R;
            const int yu1 = reflect101(gy - 1, rows) - ty0 - 2, yd1 = reflect101(gy + 1, rows) - ty0 - 2;
            const int yu2 = reflect101(gy - 2, rows) - ty0 - 2, yd2 = reflect101(gy + 2, rows) - ty0 - 2;
            float acc = __fmul_rn(C[(y - 2) * TW + cx], 0.375f);
            acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(C[yu1 * TW + cx], C[yd1 * TW + cx]), 0.25f));
            acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(C[yu2 * TW + cx], C[yd2 * TW + cx]), 0.0625f));
            if (m >= thr) v = acc;               // LO :184 `> 0.1`
        }
        if (mode >= 11) v = invert_valid(v, max_depth, thr);   // H11 (LO :191-202)
        o[(size_t)gy * cols + gx] = v;
    });
}

}  // namespace dcmt
