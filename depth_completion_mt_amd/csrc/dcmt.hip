// dcmt.hip -- implementation of the C ABI in include/dcmt.h on gfx950 (MI355X).
//
// Host side of the hot path: context (device scratch owned per GPU), launch sequencing of
// the kernels in dcmt_kernels_v1.h / dcmt_kernels_fused.h, and the host<->device copies of
// the cv::Mat entry point.  No PyTorch, no OpenCV, no CPU fallback: if there is no gfx950
// device every entry point fails with DCMT_E_NO_DEVICE / DCMT_E_HIP.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <new>
#include <vector>

#include "dcmt.h"
#include "dcmt_kernels_v1.h"
#include "dcmt_kernels_fused.h"
#include "dcmt_kernels_pair.h"
#include "dcmt_kernels_fp_pair.h"
#include "dcmt_kernels_fp_q16.h"
#include "dcmt_kernels_fp_h16.h"
#include "dcmt_kernels_slic.h"

using namespace dcmt;

struct dcmt_ctx {
    int device = 0;
    int max_rows = 0, max_cols = 0, max_batch = 0;
    size_t frame_elems = 0;           // max_rows * max_cols
    // device scratch
    float* x5 = nullptr;              // [max_batch][rows][cols] : cascade after the small fill (= pp[1], see dcmt_create)
    float* pp[2] = {nullptr, nullptr};// ping-pong of the large-fill applications
    int* colstat = nullptr;           // [max_batch][tile rows][2][cols]  (staged path)
    int* counters = nullptr;          // [max_batch][kCntStride]
    int* tb = nullptr;                // [max_batch][2][max_cols]: first / last valid row of every X6 column (k_pre table mode -> k_fp_s)
    uint32_t* norm_stats = nullptr;   // [max_batch][2]  N1: order-preserving keys of each frame's max and (inverted) min
    float* norm_coef = nullptr;       // [max_batch][2]  N1: dst = src * a + b
    // host-entry staging (allocated on first use)
    float* d_in = nullptr;
    float* d_out = nullptr;
    int32_t* d_lab = nullptr;
    int* h_counters = nullptr;        // pinned
    hipStream_t own_stream = nullptr;
    // state of the last call
    hipStream_t last_stream = nullptr;
    int last_batch = 0;
    int last_apps_launched = 0;       // loop applications (app >= 1) enqueued
    int last_has_loop = 0;            // the call went at least through H8
    int last_hip_error = 0;
    char last_path[160] = "";         // dcmt_last_path: the kernels the last call dispatched
    int timing = 0;                   // dcmt_set_kernel_timing: events around the kernel groups of the streaming path
    hipEvent_t tev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int tev_valid = 0;                // the last call recorded all five
    int poison = 0;                   // env DCMT_POISON=1: fill the staging output with NaN before every host call
    int chunk = 0;                    // frames per chunk of the fused path (0 = whole batch); env DCMT_CHUNK
    int xcd_map = 1;                  // XCD-aware workgroup->frame mapping; env DCMT_XCD_MAP=0 disables
    int wide = 1;                     // LDS-DMA row loads where alignment allows; env DCMT_WIDE=0 disables
    int fuse_fp = 1;                  // H7..H11 in one kernel (k_fp_s); env DCMT_FUSE_FP=0 keeps k_fill_s + k_post_s
    int top_table = 1;                // k_pre leaves the extension zones of X6 unwritten, k_fp_s clamps its rows and starts below the top one; env DCMT_TOP_TABLE=0 disables
    int pair = 1;                     // two columns per lane in H2..H6 (k_pre_p) where the width is even; env DCMT_PAIR=0 keeps k_pre_s
    int bands = 0;                    // row bands per strip in k_pre_p (0 = by batch size); env DCMT_BANDS
    int fbands = 0;                   // row bands per strip in k_fp_s (0 = by batch size); env DCMT_FBANDS
    int fp_pair = 0;                  // env DCMT_FP_PAIR=1: two columns per lane in H7..H11 (k_fp_p: 8 % fewer VALU instructions, but its ~200 VGPRs leave 2 waves per SIMD and it is slower)
    int fp_q16 = 1;                   // X6 as 16-bit codes + k_fp_q wherever the frames allow it (multiples of 1/256 m: checked on the device, the f32
                                      // kernels rerun behind a raised flag); env DCMT_FP_Q16=0 disables
    int assume_filled = 1;            // k_fp_s / k_fp_q without the median >= thr select where the redo chain follows; env DCMT_ASSUME_FILLED=0 keeps it
    int fp_h = 0;                     // env DCMT_FP_H=1: k_fp_h (the horizontal 31-maximum as a row pipeline through LDS: fewer VALU instructions, measured slower) instead of
                                      // k_fp_q
    int q16_breg = 1;                 // k_fp_q with the halo columns in a second register (120 output columns per wave, 3 waves per SIMD); 0 = wider strip
                                      // overlap instead (88 output columns, 4 waves per SIMD: measured 2.5 % slower -- the kernel is bound by issue, not by
                                      // occupancy); env DCMT_Q16_BREG
    int q16_min_waves = 2600;         // ... and the batch is large enough: k_fp_q has half as many, longer waves than k_fp_s (3 per SIMD instead of 4), so it
                                      // only pays from about one round of them on (measured, 352x1216, frames per call, whole step against the f32 kernels:
                                      // 128 -12 %, 256 +3 %, 512 +4 %, 1024 +5 %; threshold = 236 frames); env DCMT_Q16_MIN_WAVES
    unsigned short* x6q = nullptr;    // [max_batch][rows][cols] X6 as 16-bit codes (k_pre_p<Q16OUT> -> k_fp_q)
    int* q16_bad = nullptr;           // a ring of kQ16Flags flags; attempt n uses flag n % kQ16Flags: raised by k_pre_p<Q16OUT> when a value it stored was
                                      // not a code, and cleared one attempt ahead by that kernel too (no memset in the stream)
    unsigned q16_attempts = 0;
    int* q16_seen = nullptr;          // pinned host word (and its device address) the same kernel sets: the NEXT calls skip the 16-bit attempt
    int* q16_seen_dev = nullptr;
    int q16_skip = 0;                 // calls left without an attempt (after a raised flag: 63, then one more try)
    unsigned* winner = nullptr;       // N2: the winner plane of dcmt_project_points_dev (tags: generation | point index), allocated by its first call
    size_t winner_elems = 0;
    int winner_bits = 0;              // index bits of the plane's tag layout
    unsigned winner_gen = 0;          // generation of the last call (0: the plane is all zeros and nothing has been written)
    int* bb_min = nullptr;            // LC fast path: per (frame, label) bounding boxes, grown on demand
    int* bb_max = nullptr;
    size_t bb_ints = 0;
    // N3 (SLIC) scratch, allocated by the first dcmt_slic_labels_dev call
    int* slic_cells = nullptr;                  // two cell sets: counts [batch][cells] + overflow flags [batch] each, then the index lists [batch][cells][kSlicCellCap] each
    size_t slic_cell_cap = 0;                   // cells per frame that buffer holds
    double* slic_centers[2] = {nullptr, nullptr};
    unsigned long long* slic_sums = nullptr;
    size_t slic_center_cap = 0;                 // centres per frame the two buffers above hold
    int label_group = 0;              // LC fast path, two columns per lane: labels side by side per wave (0 = by label size); env DCMT_LABEL_GROUP
    int label_pairs = -1;             // LC fast path: one wave per label pair (1), per label (0), by label size (-1); env DCMT_LABEL_PAIRS
    int min_fused_batch = 3;          // smaller batches use the staged kernels (measured crossover with both streaming kernels in row bands,
                                      // tools/batch_sweep.py: 1 frame 20.6 k staged / 17.1 k streaming, 2 frames 34.7 k / 34.0 k, 3 frames 39.5 k / 50.4 k,
                                      // 4 frames 42.6 k / 65.7 k, 8 frames 52 k / 124 k frames/s); env DCMT_MIN_FUSED_BATCH
};

namespace {

constexpr int TH = 32, TW = 64;
constexpr int FTH_FEW = 16;          // tile height of the staged kernels for a handful of frames: twice the workgroups, a shorter
                                     // critical path (a single frame's 209 tiles of 32 rows leave a fifth of the CUs idle)

#define DCMT_HIP(ctx, call)                                        \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) {                                    \
            if (ctx) (ctx)->last_hip_error = (int)e_;              \
            return e_ == hipErrorOutOfMemory ? DCMT_E_NOMEM : DCMT_E_HIP; \
        }                                                          \
    } while (0)


// Every entry point that takes a context runs with that context's device current and puts the caller's current device back
// before it returns: one process may drive several GPUs, one host thread + one dcmt_ctx + one stream per GPU (HIP's current
// device is per thread), and a library that is shared with a framework (torch) must not move that framework's device.
struct DeviceGuard {
    int prev = -1, rc = DCMT_OK;
    explicit DeviceGuard(dcmt_ctx* ctx)
    {
        if (!ctx) return;                                   // the entry point reports DCMT_E_INVALID itself
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; rc = DCMT_E_HIP; return; }
        if (prev != ctx->device) {
            const hipError_t e = hipSetDevice(ctx->device);
            if (e != hipSuccess) { ctx->last_hip_error = (int)e; rc = DCMT_E_HIP; prev = -1; }
        } else prev = -1;                                   // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DCMT_ON_DEVICE(ctx) DeviceGuard dev_guard_(ctx); if (dev_guard_.rc != DCMT_OK) return dev_guard_.rc

uint32_t k0_bits(const uint8_t k0[25])
{
    uint32_t b = 0;
    for (int i = 0; i < 25; ++i) if (k0[i]) b |= 1u << i;
    return b;
}

// the library is built with -ffinite-math-only: test the exponent bits, not the value
bool finite_bits(float v)
{
    uint32_t b;
    std::memcpy(&b, &v, sizeof b);
    return (b & 0x7f800000u) != 0x7f800000u;
}

int check_params(const dcmt_ctx* ctx, const void* a, const void* b, int rows, int cols, int batch, const dcmt_params* p)
{
    if (!ctx || !a || !b || !p) return DCMT_E_INVALID;
    if (rows < 1 || cols < 1 || batch < 1) return DCMT_E_INVALID;
    if (batch > ctx->max_batch || rows > ctx->max_rows || cols > ctx->max_cols) return DCMT_E_INVALID;
    if (p->blur == DCMT_BLUR_BILATERAL) return DCMT_E_UNSUPPORTED;
    if (p->blur != DCMT_BLUR_NONE && p->blur != DCMT_BLUR_GAUSSIAN) return DCMT_E_INVALID;
    if (p->max_fill_iters < 1 || p->max_fill_iters > kMaxIters) return DCMT_E_INVALID;
    if (p->spec_fill_iters < 0 || p->spec_fill_iters > kMaxIters) return DCMT_E_INVALID;
    const bool norm = (p->flags & DCMT_FLAG_NORMALIZE) != 0;
    if (p->stop_after < (norm ? DCMT_STAGE_NORMALIZE : DCMT_STAGE_INVERT) || p->stop_after > DCMT_STAGE_FINAL) return DCMT_E_INVALID;
    if (norm && !(finite_bits(p->norm_lo) && finite_bits(p->norm_hi))) return DCMT_E_INVALID;
    if (k0_bits(p->k0) == 0) return DCMT_E_INVALID;
    return DCMT_OK;
}

// Scratch only one of the paths uses is allocated by the first call that takes that path (never again afterwards): the column
// statistics of the staged tile kernels, the 16-bit plane of k_pre_p<Q16OUT> -> k_fp_q.
int ensure_colstat(dcmt_ctx* ctx)
{
    if (!ctx->colstat)
        DCMT_HIP(ctx, hipMalloc((void**)&ctx->colstat, sizeof(int) * 2 * (size_t)ctx->max_cols * ((ctx->max_rows + FTH_FEW - 1) / FTH_FEW) * ctx->max_batch));
    return DCMT_OK;
}
int ensure_x6q(dcmt_ctx* ctx)
{
    if (!ctx->x6q) DCMT_HIP(ctx, hipMalloc((void**)&ctx->x6q, sizeof(unsigned short) * ctx->frame_elems * (size_t)ctx->max_batch + 16));
    return DCMT_OK;
}

// grid of the kernels that deal (frame, strip) pairs to waves in one flat sequence (wave_strip in dcmt_kernels_fused.h)
constexpr unsigned kQ16Flags = 64;

dim3 wave_grid(int strips, int batch, int xcd_map)
{
    return dim3(xcd_map ? 8 * (((batch / 8) * strips + 3) / 4) : (batch * strips + 3) / 4);
}

dim3 wave_grid_n(int strips, int batch, int xcd_map, int wpb)
{
    return dim3(xcd_map ? 8 * (((batch / 8) * strips + wpb - 1) / wpb) : (batch * strips + wpb - 1) / wpb);
}

dim3 tile_grid(int rows, int cols, int batch) { return dim3((cols + TW - 1) / TW, (rows + TH - 1) / TH, batch); }

int k0_preset(uint32_t kb)
{
    uint8_t k[25];
    dcmt_k0_as_compiled(k);
    if (kb == k0_bits(k)) return K0_AS_COMPILED;
    dcmt_k0_diamond(k);
    if (kb == k0_bits(k)) return K0_DIAMOND;
    return -1;
}

// The hole-closure loop shared by both paths.  `launch_app(i)` enqueues application i
// (reads pp[(i-1)&1], writes pp[i&1], skips frames without holes).  Returns the number of
// applications enqueued through *apps.
template <typename LaunchApp>
int fill_loop(dcmt_ctx* ctx, int batch, const dcmt_params* p, hipStream_t st, bool sync_loop, LaunchApp launch_app, int* apps_out)
{
    int rc = DCMT_OK, apps = 0;
    if (sync_loop) {
        // Iteration i of the reference's loop (LO :146-166) counts the holes left by application
        // i-1, fills them (application i: a no-op when there are none) and stops when it saw
        // none; the cap bounds i.
        for (int i = 1;; ++i) {
            DCMT_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->counters, sizeof(int) * (size_t)batch * kCntStride,
                                         hipMemcpyDeviceToHost, st));
            DCMT_HIP(ctx, hipStreamSynchronize(st));
            bool any = false;
            for (int f = 0; f < batch; ++f) {
                const int n_i = ctx->h_counters[(size_t)f * kCntStride + i];   // [1 + (i-1)]
                any |= n_i > 0;
                if (p->verbose) std::printf("%d\n", n_i);                       // LO :161
            }
            if (!any) break;
            launch_app(i);
            DCMT_HIP(ctx, hipGetLastError());
            apps = i;
            if (i >= p->max_fill_iters) { rc = DCMT_E_NOT_CONVERGED; break; }
        }
    } else {
        int n = p->spec_fill_iters;
        if (n > p->max_fill_iters) n = p->max_fill_iters;
        for (int i = 1; i <= n; ++i) launch_app(i);
        apps = n;
        DCMT_HIP(ctx, hipGetLastError());
    }
    *apps_out = apps;
    return rc;
}

// Fast path.  Whole chain: k_pre_s -> k_fp_s, then three launches that return at once for every frame
// k_fp_s finished (k_fill_s redo, k_fill_s loop applications, k_post_s only_if_holes).  stop_after probes:
// k_pre_s -> k_fill_s (-> loop) -> k_post_s / copy.  Preconditions are checked by the caller.
// ctx->chunk (env DCMT_CHUNK, default 0 = off) walks the batch in chunks; measured slower, kept as a knob.
// d_x4 != nullptr: X4 is already there (LC fast path): k_pre_s only runs H5 + H6 on it.
// d_src16 != nullptr: uint16 ingest fused into k_pre_s.
int run_chain_fused(dcmt_ctx* ctx, int k0kind, const float* d_src, float* d_dst, int rows, int cols, int batch,
                    const dcmt_params* p, hipStream_t st, bool sync_loop, const float* d_x4 = nullptr,
                    const uint16_t* d_src16 = nullptr, float in_scale = 1.0f, const float* coef = nullptr)
{
    const int stop = p->stop_after;
    ctx->last_stream = st;
    ctx->last_batch = batch;
    ctx->last_apps_launched = 0;
    ctx->last_has_loop = 0;
    // (the hole counters are cleared by the first kernel of the chain: clear_frame_counters)
    auto stamp = [&](int i) { if (ctx->timing && ctx->tev[i]) (void)hipEventRecord(ctx->tev[i], st); };
    stamp(1);
    const size_t fe = (size_t)rows * cols;
    const int chunk = (ctx->chunk > 0 && !sync_loop) ? ctx->chunk : batch;   // the host-synchronised loop works on the whole batch
    const hipStream_t ps = st;
    const bool bl = p->blur == DCMT_BLUR_GAUSSIAN;
    int rc = DCMT_OK, apps_all = 0;
    // 16-bit X6: a frame that is no multiple of 1/256 m costs the attempt AND the f32 rerun; once a call has raised the flag (seen
    // here at the start of a later call, without synchronising) the next 63 calls go straight to the f32 kernels.  (The uint16 entry
    // point's depths are multiples of 1/256 m by construction, but a payload beyond 30719 -- 119.996 m -- has no code either.)
    bool q16_try = ctx->fp_q16 != 0;
    if (q16_try) {
        if (*(volatile int*)ctx->q16_seen) { *(volatile int*)ctx->q16_seen = 0; ctx->q16_skip = 63; }
        if (ctx->q16_skip > 0) { --ctx->q16_skip; q16_try = false; }
    }
    for (int f0 = 0; f0 < batch; f0 += chunk) {
        const int nb = batch - f0 < chunk ? batch - f0 : chunk;
        const int xm = (ctx->xcd_map && nb % 8 == 0) ? 1 : 0;
        const float* src = (d_x4 ? d_x4 : d_src) + f0 * fe;
        float* dst = d_dst + f0 * fe;
        float* x6 = ctx->x5 + f0 * fe;
        float* pp0 = ctx->pp[0] + f0 * fe;
        float* pp1 = ctx->pp[1] + f0 * fe;
        int* cnt = ctx->counters + (size_t)f0 * kCntStride;
        const float* cf = coef ? coef + 2 * (size_t)f0 : nullptr;
        // table mode: only the k_fp_s path reads X6 through the per-column table (the probes and the unfused kernels get a fully written X6)
        int bands = 1;                      // row bands of k_pre_p = table slots per frame
        bool q16 = false;                   // this chunk's X6 is 16-bit codes (in ctx->x6q)
        int* qbad = ctx->q16_bad;           // ... and the flag its attempt raises on a frame that has none
        int* tc = (stop == DCMT_STAGE_FINAL && ctx->fuse_fp && ctx->top_table) ? ctx->tb : nullptr;   // chunks follow each other in the stream: each may use the whole table
        {
            float* o6 = stop == DCMT_STAGE_EXTEND ? dst : x6;
            // LDS-DMA rows need 16-byte aligned sources: cols % 4 == 0 and a 16-byte aligned base
            const bool wide = ctx->wide && cols % 4 == 0 && ((uintptr_t)src % 16 == 0) && !d_src16;
            const uint16_t* src16 = d_src16 ? d_src16 + f0 * fe : nullptr;
#define DCMT_PRE(KIND, WIDE) { using G = PreS<KIND, WIDE>; const int strips = (cols + G::VW - 1) / G::VW; \
                if (d_x4) hipLaunchKernelGGL((k_pre_s<KIND, WIDE, true, false>), wave_grid(strips, nb, xm), dim3(256), 0, ps, (const void*)src, o6, \
                                             rows, cols, strips, nb, xm, p->max_depth, p->valid_thresh, 1.0f, (const float*)nullptr, tc, cnt); \
                else if (src16) hipLaunchKernelGGL((k_pre_s<KIND, false, false, true>), wave_grid(strips, nb, xm), dim3(256), 0, ps, (const void*)src16, o6, \
                                             rows, cols, strips, nb, xm, p->max_depth, p->valid_thresh, in_scale, (const float*)nullptr, tc, cnt); \
                else if (cf) hipLaunchKernelGGL((k_pre_s<KIND, WIDE, false, false, true>), wave_grid(strips, nb, xm), dim3(256), 0, ps, (const void*)src, o6, \
                                        rows, cols, strips, nb, xm, p->max_depth, p->valid_thresh, 1.0f, cf, tc, cnt); \
                else hipLaunchKernelGGL((k_pre_s<KIND, WIDE, false, false>), wave_grid(strips, nb, xm), dim3(256), 0, ps, (const void*)src, o6, \
                                        rows, cols, strips, nb, xm, p->max_depth, p->valid_thresh, 1.0f, (const float*)nullptr, tc, cnt); }
            // two columns per lane (k_pre_p) wherever a lane's 8-byte accesses are aligned: even width, 8-byte aligned frames
            const bool pair = ctx->pair && cols % 2 == 0 && cols >= 8 && ((uintptr_t)(src16 ? (const void*)src16 : (const void*)src) % (src16 ? 4 : 8) == 0) &&
                              ((uintptr_t)o6 % 8 == 0);
            // row bands: full-height strips of a small batch leave most wave slots empty; bands need the (ti, bi) table (one slot per band)
            if (pair && tc) {
                const int pstr = (cols + PreP<K0_AS_COMPILED, false>::VW - 1) / PreP<K0_AS_COMPILED, false>::VW;
                bands = ctx->bands > 0 ? ctx->bands : ((long long)nb * pstr >= 2560 ? 1 : (int)((2560 + (long long)nb * pstr - 1) / ((long long)nb * pstr)));
                if (bands > rows / 32) bands = rows / 32 > 0 ? rows / 32 : 1;
                if (bands > kMaxBands) bands = kMaxBands;
            }
#define DCMT_PREP(KIND, QOUT, O6, QBAD, GATE, QCLR) { using G4 = PreP<KIND, true>; using G0 = PreP<KIND, false>; \
                if (d_x4) { const int strips = (cols + G4::VW - 1) / G4::VW; \
                    hipLaunchKernelGGL((k_pre_p<KIND, true, false, false, QOUT>), wave_grid(strips * bands, nb, xm), dim3(256), 0, ps, (const void*)src, O6, \
                                       rows, cols, strips, bands, nb, xm, p->max_depth, p->valid_thresh, 1.0f, (const float*)nullptr, tc, cnt, QBAD, GATE, ctx->q16_seen_dev, QCLR); } \
                else { const int strips = (cols + G0::VW - 1) / G0::VW; \
                    if (src16) hipLaunchKernelGGL((k_pre_p<KIND, false, true, false, QOUT>), wave_grid(strips * bands, nb, xm), dim3(256), 0, ps, (const void*)src16, O6, \
                                       rows, cols, strips, bands, nb, xm, p->max_depth, p->valid_thresh, in_scale, (const float*)nullptr, tc, cnt, QBAD, GATE, ctx->q16_seen_dev, QCLR); \
                    else if (cf) { if constexpr (!QOUT) hipLaunchKernelGGL((k_pre_p<KIND, false, false, true>), wave_grid(strips * bands, nb, xm), dim3(256), 0, ps, (const void*)src, O6, \
                                       rows, cols, strips, bands, nb, xm, p->max_depth, p->valid_thresh, 1.0f, cf, tc, cnt, QBAD, GATE, ctx->q16_seen_dev, QCLR); } \
                    else hipLaunchKernelGGL((k_pre_p<KIND, false, false, false, QOUT>), wave_grid(strips * bands, nb, xm), dim3(256), 0, ps, (const void*)src, O6, \
                                       rows, cols, strips, bands, nb, xm, p->max_depth, p->valid_thresh, 1.0f, (const float*)nullptr, tc, cnt, QBAD, GATE, ctx->q16_seen_dev, QCLR); } }
            // 16-bit X6 (k_pre_p<Q16OUT> -> k_fp_q): the whole chain in table mode, two columns per lane, the reference's constants, no
            // normalisation in front (normalised depths are no multiples of 1/256)
            // in place (or overlapping) f32 calls never take the 16-bit attempt: k_fp_q writes dst BEFORE the gated f32 rerun would read src again
            // (the f32 kernels alone are alias-safe: src is only read by k_pre into ctx scratch, dst is written last)
            const bool src_dst_overlap = !src16 && (uintptr_t)src < (uintptr_t)(dst + (size_t)nb * fe) && (uintptr_t)dst < (uintptr_t)(src + (size_t)nb * fe);
            q16 = q16_try && (long long)nb * ((cols + FpP::VW - 1) / FpP::VW) >= ctx->q16_min_waves && pair && tc && !cf && Q16::params_ok(p->max_depth, p->valid_thresh) && (uintptr_t)dst % 8 == 0 &&
                  (!src16 || in_scale == 0.00390625f) && !src_dst_overlap;
            if (q16) { const int erc = ensure_x6q(ctx); if (erc != DCMT_OK) return erc; }
            float* x6q = reinterpret_cast<float*>(ctx->x6q + f0 * fe);
            if (q16) {
                // this attempt's flag (cleared by the previous attempt's kernel, or by dcmt_create) and the next one's, which this attempt's kernel clears
                qbad = ctx->q16_bad + ctx->q16_attempts % kQ16Flags;
                int* qnext = ctx->q16_bad + (ctx->q16_attempts + 1) % kQ16Flags;
                ++ctx->q16_attempts;
                if (k0kind == K0_AS_COMPILED) DCMT_PREP(K0_AS_COMPILED, true, x6q, qbad, (const int*)nullptr, qnext) else DCMT_PREP(K0_DIAMOND, true, x6q, qbad, (const int*)nullptr, qnext)
            } else if (pair) {
                if (k0kind == K0_AS_COMPILED) DCMT_PREP(K0_AS_COMPILED, false, o6, (int*)nullptr, (const int*)nullptr, (int*)nullptr) else DCMT_PREP(K0_DIAMOND, false, o6, (int*)nullptr, (const int*)nullptr, (int*)nullptr)
            } else {
                if (k0kind == K0_AS_COMPILED) { if (wide) DCMT_PRE(K0_AS_COMPILED, true) else DCMT_PRE(K0_AS_COMPILED, false) }
                else { if (wide) DCMT_PRE(K0_DIAMOND, true) else DCMT_PRE(K0_DIAMOND, false) }
            }
#undef DCMT_PRE
            DCMT_HIP(ctx, hipGetLastError());
            stamp(2);
            std::snprintf(ctx->last_path, sizeof ctx->last_path, "%s%s%s", d_x4 ? "k_label_bbox + k_label_stage + " : "",
                          q16 ? (src16 ? "k_pre_p<U16,Q16OUT>" : "k_pre_p<Q16OUT>") : pair ? (d_x4 ? "k_pre_p<START4>" : src16 ? "k_pre_p<U16>" : cf ? "k_pre_p<NORM>" : "k_pre_p") : "k_pre_s",
                          bands > 1 ? " (row bands)" : "");
            if (stop == DCMT_STAGE_EXTEND) continue;
        }
        const int fstrips = (cols + FillS::VW - 1) / FillS::VW;
        const dim3 fgrid(((fstrips + 3) / 4) * nb);
        if (stop == DCMT_STAGE_FINAL && ctx->fuse_fp) {
            // one kernel for H7..H11; frames it leaves with holes are redone by the unfused kernels below
            const int pstrips = (cols + PostS::VW - 1) / PostS::VW;
            const dim3 pg(((pstrips + 3) / 4) * nb), b256(256);
            // k_fp_s deals (frame, strip) pairs to waves in one flat sequence (per XCD with the XCD map): no half-empty workgroups
            // row bands for k_fp_s: a batch whose strips are fewer than two waves per SIMD runs every strip as fb_s bands, about one
            // round of three waves per SIMD in all (a band pays 19 + 19 rows of halo and 19 steps of pipeline: only worth it while the
            // GPU is not full -- from ~100 frames of 1216 columns on there is one band)
            int fb_s = 1;
            if (tc) {
                const long long w1 = (long long)nb * pstrips;
                fb_s = ctx->fbands > 0 ? ctx->fbands : (w1 >= 2048 ? 1 : (int)((3072 + w1 / 2) / w1));
                if (fb_s > rows / 32) fb_s = rows / 32 > 0 ? rows / 32 : 1;
                if (fb_s < 1) fb_s = 1;
            }
            const dim3 fpg = wave_grid(pstrips * fb_s, nb, xm);
            // two columns per lane (k_fp_p) wherever a lane's 8-byte stores are aligned: even width, 8-byte aligned frames
            const bool fpp = ctx->fp_pair && cols % 2 == 0 && cols >= 8 && ((uintptr_t)dst % 8 == 0);
            // frames this kernel leaves with holes are recomputed by the redo chain below whenever that chain is enqueued (always on the host
            // entry points, with spec_fill_iters >= 1 on the device ones): then the kernel may leave out the select that only such frames need
            const bool filled = bl && ctx->assume_filled && (sync_loop || (p->spec_fill_iters >= 1 && p->max_fill_iters >= 1));
            if (q16) {
                const void* xq = ctx->x6q + f0 * fe;
#define DCMT_FPQ(BL, BREG, FILLED) { const int qstrips = (cols + FpQ::vw<BREG>() - 1) / FpQ::vw<BREG>(); \
                    hipLaunchKernelGGL((k_fp_q<BL, true, BREG, FILLED>), wave_grid(qstrips, nb, xm), b256, 0, st, xq, dst, cnt, rows, cols, qstrips, nb, xm, \
                                       p->max_depth, p->valid_thresh, (const int*)tc, bands); }
#define DCMT_FPH(BL, FILLED) { const int qstrips = (cols + FpP::VW - 1) / FpP::VW; \
                    hipLaunchKernelGGL((k_fp_h<BL, FILLED>), wave_grid(qstrips, nb, xm), b256, 0, st, xq, dst, cnt, rows, cols, qstrips, nb, xm, \
                                       p->max_depth, p->valid_thresh, (const int*)tc, bands); }
                // (k_fp_h<BLUR, !FILLED> -- the select of LO :184 kept, only where a device caller asks for no speculative loop applications -- needs more
                //  than the 168 registers three waves per SIMD leave: that combination stays with k_fp_q)
                const bool fph = ctx->fp_h && ctx->q16_breg && (filled || !bl);
                if (fph) { if (filled) DCMT_FPH(true, true) else DCMT_FPH(false, false) }
                else if (ctx->q16_breg) { if (filled) DCMT_FPQ(true, true, true) else if (bl) DCMT_FPQ(true, true, false) else DCMT_FPQ(false, true, false) }
                else               { if (bl) DCMT_FPQ(true, false, false) else DCMT_FPQ(false, false, false) }
#undef DCMT_FPQ
#undef DCMT_FPH
                {
                    // frames that are no multiples of 1/256 m: both f32 kernels again, gated on the flag the attempt raised (they return at once otherwise)
                    // (the uint16 entry point too: a payload beyond 30719 = 119.996 m has no code)
                    float* o6 = x6;
                    const uint16_t* src16 = d_src16 ? d_src16 + f0 * fe : nullptr; const float* cf = nullptr; const hipStream_t ps = st;
                    const float* src = (d_x4 ? d_x4 : d_src) + f0 * fe;
                    if (k0kind == K0_AS_COMPILED) DCMT_PREP(K0_AS_COMPILED, false, o6, (int*)nullptr, (const int*)qbad, (int*)nullptr) else DCMT_PREP(K0_DIAMOND, false, o6, (int*)nullptr, (const int*)qbad, (int*)nullptr)
                    if (filled) hipLaunchKernelGGL((k_fp_s<true, true>), fpg, b256, 0, st, x6, dst, cnt, rows, cols, pstrips, nb, xm, p->max_depth, p->valid_thresh, (const int*)tc, bands, (const int*)qbad, fb_s);
                    else if (bl) hipLaunchKernelGGL((k_fp_s<true>), fpg, b256, 0, st, x6, dst, cnt, rows, cols, pstrips, nb, xm, p->max_depth, p->valid_thresh, (const int*)tc, bands, (const int*)qbad, fb_s);
                    else    hipLaunchKernelGGL((k_fp_s<false>), fpg, b256, 0, st, x6, dst, cnt, rows, cols, pstrips, nb, xm, p->max_depth, p->valid_thresh, (const int*)tc, bands, (const int*)qbad, fb_s);
                }
            }
            else if (fpp) {
                const int qstrips = (cols + FpP::VW - 1) / FpP::VW;
                const dim3 qg = wave_grid_n(qstrips, nb, xm, FpP::WPB), qb(64 * FpP::WPB);
                if (bl) hipLaunchKernelGGL((k_fp_p<true>), qg, qb, 0, st, x6, dst, cnt, rows, cols, qstrips, nb, xm, p->max_depth, p->valid_thresh, (const int*)tc, bands);
                else    hipLaunchKernelGGL((k_fp_p<false>), qg, qb, 0, st, x6, dst, cnt, rows, cols, qstrips, nb, xm, p->max_depth, p->valid_thresh, (const int*)tc, bands);
            }
            else if (filled) hipLaunchKernelGGL((k_fp_s<true, true>), fpg, b256, 0, st, x6, dst, cnt, rows, cols, pstrips, nb, xm, p->max_depth, p->valid_thresh, (const int*)tc, bands, (const int*)nullptr, fb_s);
            else if (bl) hipLaunchKernelGGL((k_fp_s<true>), fpg, b256, 0, st, x6, dst, cnt, rows, cols, pstrips, nb, xm, p->max_depth, p->valid_thresh, (const int*)tc, bands, (const int*)nullptr, fb_s);
            else    hipLaunchKernelGGL((k_fp_s<false>), fpg, b256, 0, st, x6, dst, cnt, rows, cols, pstrips, nb, xm, p->max_depth, p->valid_thresh, (const int*)tc, bands, (const int*)nullptr, fb_s);
#undef DCMT_PREP
            DCMT_HIP(ctx, hipGetLastError());
            stamp(3);
            { const size_t n_ = std::strlen(ctx->last_path);
              std::snprintf(ctx->last_path + n_, sizeof ctx->last_path - n_, " + %s", q16 ? (ctx->fp_h && ctx->q16_breg && (filled || !bl) ? "k_fp_h" : "k_fp_q") : fpp ? "k_fp_p" : (fb_s > 1 ? "k_fp_s (row bands)" : "k_fp_s")); }
            ctx->last_has_loop = 1;
            const int n_redo = sync_loop ? p->max_fill_iters : (p->spec_fill_iters < p->max_fill_iters ? p->spec_fill_iters : p->max_fill_iters);
            if (n_redo > 0) {
                if (sync_loop) {      // host entry points: look before launching anything else
                    DCMT_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->counters, sizeof(int) * (size_t)batch * kCntStride, hipMemcpyDeviceToHost, st));
                    DCMT_HIP(ctx, hipStreamSynchronize(st));
                    bool any = false;
                    for (int f = 0; f < batch; ++f) any |= ctx->h_counters[(size_t)f * kCntStride + 1] > 0;
                    if (!any) { if (p->verbose) for (int f = 0; f < batch; ++f) std::printf("0\n"); continue; }
                }
                hipLaunchKernelGGL(k_fill_s, fgrid, b256, 0, st, x6, pp0, cnt, rows, cols, fstrips, nb, xm, p->valid_thresh, 0, 1, (const int*)tc, bands,
                                   q16 ? (const unsigned short*)(ctx->x6q + f0 * fe) : (const unsigned short*)nullptr, (const int*)qbad);
                int apps = 0;
                const int lrc = fill_loop(ctx, batch, p, st, sync_loop, [&](int i) {
                    hipLaunchKernelGGL(k_fill_s, fgrid, b256, 0, st, (i & 1) ? pp0 : pp1, (i & 1) ? pp1 : pp0, cnt, rows, cols,
                                       fstrips, nb, xm, p->valid_thresh, i, 0, (const int*)nullptr, 1, (const unsigned short*)nullptr, (const int*)nullptr);
                }, &apps);
                if (lrc != DCMT_OK && lrc != DCMT_E_NOT_CONVERGED) return lrc;
                if (lrc != DCMT_OK) rc = lrc;
                apps_all = apps;
                if (bl) hipLaunchKernelGGL((k_post_s<11, true>), pg, b256, 0, st, pp0, pp1, dst, cnt, apps, rows, cols, pstrips, nb, xm, p->max_depth, p->valid_thresh, 1);
                else    hipLaunchKernelGGL((k_post_s<11, false>), pg, b256, 0, st, pp0, pp1, dst, cnt, apps, rows, cols, pstrips, nb, xm, p->max_depth, p->valid_thresh, 1);
                DCMT_HIP(ctx, hipGetLastError());
            }
            stamp(4);
            ctx->tev_valid = ctx->timing && chunk == batch;
            continue;
        }
        hipLaunchKernelGGL(k_fill_s, fgrid, dim3(256), 0, st, x6, stop == DCMT_STAGE_FILL31 ? dst : pp0, cnt, rows, cols,
                           fstrips, nb, xm, p->valid_thresh, 0, 0, (const int*)nullptr, 1, (const unsigned short*)nullptr, (const int*)nullptr);
        DCMT_HIP(ctx, hipGetLastError());
        if (stop == DCMT_STAGE_FILL31) continue;

        ctx->last_has_loop = 1;
        int apps = 0;
        const int lrc = fill_loop(ctx, batch, p, st, sync_loop, [&](int i) {
            hipLaunchKernelGGL(k_fill_s, fgrid, dim3(256), 0, st, (i & 1) ? pp0 : pp1, (i & 1) ? pp1 : pp0, cnt, rows, cols,
                               fstrips, nb, xm, p->valid_thresh, i, 0, (const int*)nullptr, 1, (const unsigned short*)nullptr, (const int*)nullptr);
        }, &apps);
        if (lrc != DCMT_OK && lrc != DCMT_E_NOT_CONVERGED) return lrc;
        if (lrc != DCMT_OK) rc = lrc;
        apps_all = apps;

        if (stop <= DCMT_STAGE_FILLLOOP) {
            hipLaunchKernelGGL((k_post_v1<TH, TW>), tile_grid(rows, cols, nb), dim3(kThreads), 0, st, pp0, pp1, dst, cnt, apps,
                               rows, cols, p->max_depth, p->valid_thresh, p->blur, 8);
        } else {
            const int strips = (cols + PostS::VW - 1) / PostS::VW;
            const dim3 g(((strips + 3) / 4) * nb), b(256);
#define DCMT_POST(MODE, BLUR) hipLaunchKernelGGL((k_post_s<MODE, BLUR>), g, b, 0, st, pp0, pp1, dst, cnt, apps, rows, cols, strips, \
                                                 nb, xm, p->max_depth, p->valid_thresh, 0)
            if (stop == DCMT_STAGE_MEDIAN5) DCMT_POST(9, false);
            else if (stop == DCMT_STAGE_BLUR) { if (bl) DCMT_POST(10, true); else DCMT_POST(10, false); }
            else { if (bl) DCMT_POST(11, true); else DCMT_POST(11, false); }
#undef DCMT_POST
        }
        DCMT_HIP(ctx, hipGetLastError());
    }
    ctx->last_apps_launched = apps_all;
    return rc;
}

// Enqueues the cascade on `st`.  sync_loop: run the hole-closure loop exactly as the
// reference would, reading the hole counters back between applications (host entry
// points); otherwise enqueue p->spec_fill_iters applications speculatively.
int run_chain(dcmt_ctx* ctx, const float* d_src, const int32_t* d_labels, int n_labels, int use_superpixel,
              float* d_dst, int rows, int cols, int batch, const dcmt_params* p, bool force_gaussian,
              hipStream_t st, bool sync_loop, const uint16_t* d_src16 = nullptr, float in_scale = 1.0f)
{
    const dim3 grid = tile_grid(rows, cols, batch), block(kThreads);
    const bool few = batch < 12;                    // measured up to 8 frames (tools/batch_sweep.py): +15 % at 3 and 6, +7 % at 8
    const dim3 fgrid((cols + TW - 1) / TW, few ? (rows + FTH_FEW - 1) / FTH_FEW : (rows + TH - 1) / TH, batch);
#define DCMT_FILL31(...) { if (few) hipLaunchKernelGGL((k_fill31_v1<FTH_FEW, TW>), fgrid, block, 0, st, __VA_ARGS__); \
                           else hipLaunchKernelGGL((k_fill31_v1<TH, TW>), fgrid, block, 0, st, __VA_ARGS__); }
    const uint32_t kb = k0_bits(p->k0);
    const int stop = p->stop_after;
    const int blur = force_gaussian ? (int)DCMT_BLUR_GAUSSIAN : p->blur;
    ctx->tev_valid = 0;
    if (ctx->timing && ctx->tev[0]) (void)hipEventRecord(ctx->tev[0], st);
    const float* coef = nullptr;
    if (p->flags & DCMT_FLAG_NORMALIZE) {
        // N1: one read-only pass for the per-frame extrema, then (a, b) per frame; the first kernel of whichever
        // path runs below applies them while it loads
        if (d_src16) return DCMT_E_UNSUPPORTED;
        const size_t fe = (size_t)rows * cols;
        // (norm_stats is all zero here: dcmt_create cleared it, k_norm_coef clears what it has read)
        hipLaunchKernelGGL(k_minmax, dim3(kMinmaxUnits * batch), dim3(256), 0, st, d_src, ctx->norm_stats, fe, batch,
                           (ctx->xcd_map && batch % 8 == 0) ? 1 : 0);
        hipLaunchKernelGGL(k_norm_coef, dim3((batch + 63) / 64), dim3(64), 0, st, ctx->norm_stats, ctx->norm_coef, batch, p->norm_lo, p->norm_hi);
        DCMT_HIP(ctx, hipGetLastError());
        coef = ctx->norm_coef;
        if (stop == DCMT_STAGE_NORMALIZE) {
            hipLaunchKernelGGL(k_norm_write, dim3(2048), dim3(256), 0, st, d_src, d_dst, coef, fe, batch);
            DCMT_HIP(ctx, hipGetLastError());
            ctx->last_stream = st; ctx->last_batch = batch; ctx->last_apps_launched = 0; ctx->last_has_loop = 0;
            return DCMT_OK;
        }
    }
    {
        const int kind = k0_preset(kb);
        const bool labeled = d_labels && use_superpixel;
        // the streaming kernels give one wave a whole column strip: a handful of frames cannot fill the
        // GPU with them, there the staged tile kernels (hundreds of small workgroups per frame) win
        const bool big_enough = batch >= ctx->min_fused_batch || (p->flags & DCMT_FLAG_FORCE_FUSED);
        if (!(p->flags & DCMT_FLAG_FORCE_STAGED) && big_enough && labeled && kind >= 0 && rows >= 8 && cols >= 8 && n_labels > 0 &&
            (stop == DCMT_STAGE_FINAL || stop == DCMT_STAGE_CLOSE5)) {
            // LC fast path: bounding boxes -> one wave per label (masked H2..H4) -> X4 -> the img_completion kernels
            const size_t need = (size_t)batch * n_labels * 2;
            if (need > ctx->bb_ints) {
                (void)hipFree(ctx->bb_min); (void)hipFree(ctx->bb_max);
                ctx->bb_min = ctx->bb_max = nullptr; ctx->bb_ints = 0;
                DCMT_HIP(ctx, hipMalloc((void**)&ctx->bb_min, sizeof(int) * need));
                DCMT_HIP(ctx, hipMalloc((void**)&ctx->bb_max, sizeof(int) * need));
                ctx->bb_ints = need;
                // "no box" everywhere, once: the label stage's waves put every entry they have read back into this state (a fill in front
                // of every call is two dependent operations with a bubble behind the previous call's last kernel each)
                DCMT_HIP(ctx, hipMemsetAsync(ctx->bb_min, 0x7f, sizeof(int) * need, st));
                DCMT_HIP(ctx, hipMemsetAsync(ctx->bb_max, 0xff, sizeof(int) * need, st));
            }
            float* x4 = stop == DCMT_STAGE_CLOSE5 ? d_dst : ctx->pp[0];       // (dead before the redo chain writes pp[0]; x5 shares pp[1])
            const dim3 bg((cols + 63) / 64, (rows + kBboxRows - 1) / kBboxRows, batch);
            const size_t table = sizeof(int) * 4 * (size_t)n_labels;
            if (table <= 48 * 1024 && !std::getenv("DCMT_BBOX_GLOBAL"))
                hipLaunchKernelGGL(k_label_bbox<true>, bg, dim3(256), table, st, d_src, d_labels, n_labels, ctx->bb_min, ctx->bb_max, x4,
                                   rows, cols, p->max_depth, p->valid_thresh, coef);
            else
                hipLaunchKernelGGL(k_label_bbox<false>, bg, dim3(256), 0, st, d_src, d_labels, n_labels, ctx->bb_min, ctx->bb_max, x4,
                                   rows, cols, p->max_depth, p->valid_thresh, coef);
            // two columns per lane (k_label_stage_p) where a lane's 8-byte accesses are aligned; G labels side by side per wave, from the
            // mean label area (a grown box of w + 10 columns takes (w + 10) / 2 + 1 lanes; SLIC-like labels are a few columns wider
            // than the square root of their area)
            const bool lpair = ctx->pair && cols % 2 == 0 && cols >= 8 && (uintptr_t)d_src % 8 == 0 && (uintptr_t)d_labels % 8 == 0 && (uintptr_t)x4 % 8 == 0;
            int G = 1;
            {
                const double w = std::sqrt((double)rows * cols / n_labels) + 4.0;
                const int lanes = (int)((w + 10.0) / 2.0) + 1;
                G = (64 + lanes / 4) / (lanes > 0 ? lanes : 1);        // as many as fit side by side, rounded up when they nearly do (the rest gets a pass of its own)
                if (G < 1) G = 1;
                if (G > kLabelGroupMax) G = kLabelGroupMax;
                if (ctx->label_group >= 1 && ctx->label_group <= kLabelGroupMax) G = ctx->label_group;
            }
            const int lwaves = (n_labels + G - 1) / G;
            const dim3 lgp((lwaves + 3) / 4, batch);
#define DCMT_LSTAGEP(KIND, NORM) hipLaunchKernelGGL((k_label_stage_p<KIND, NORM>), lgp, dim3(256), 0, st, d_src, d_labels, n_labels, G, \
                                                   ctx->bb_min, ctx->bb_max, x4, rows, cols, p->max_depth, p->valid_thresh, coef)
            // labels of about 22 columns or less (the mean box of an even partition, with SLIC-like slack) can share a
            // wave: one wave per label PAIR; few large labels: one wave per label (see k_label_stage_s)
            const bool pairs = ctx->label_pairs >= 0 ? ctx->label_pairs != 0 : (double)rows * cols / n_labels <= 22.0 * 22.0;
            const dim3 lg(pairs ? (n_labels + 7) / 8 : (n_labels + 3) / 4, batch);
#define DCMT_LSTAGE(KIND, NORM) { if (pairs) hipLaunchKernelGGL((k_label_stage_s<KIND, NORM, true>), lg, dim3(256), 0, st, d_src, d_labels, n_labels, \
                                                   ctx->bb_min, ctx->bb_max, x4, rows, cols, p->max_depth, p->valid_thresh, coef); \
                                  else hipLaunchKernelGGL((k_label_stage_s<KIND, NORM, false>), lg, dim3(256), 0, st, d_src, d_labels, n_labels, \
                                                   ctx->bb_min, ctx->bb_max, x4, rows, cols, p->max_depth, p->valid_thresh, coef); }
            if (lpair) {
                if (kind == K0_AS_COMPILED) { if (coef) DCMT_LSTAGEP(K0_AS_COMPILED, true); else DCMT_LSTAGEP(K0_AS_COMPILED, false); }
                else { if (coef) DCMT_LSTAGEP(K0_DIAMOND, true); else DCMT_LSTAGEP(K0_DIAMOND, false); }
            } else if (kind == K0_AS_COMPILED) { if (coef) DCMT_LSTAGE(K0_AS_COMPILED, true) else DCMT_LSTAGE(K0_AS_COMPILED, false) }
            else { if (coef) DCMT_LSTAGE(K0_DIAMOND, true) else DCMT_LSTAGE(K0_DIAMOND, false) }
#undef DCMT_LSTAGE
#undef DCMT_LSTAGEP
            DCMT_HIP(ctx, hipGetLastError());
            if (stop == DCMT_STAGE_CLOSE5) {
                ctx->last_stream = st; ctx->last_batch = batch; ctx->last_apps_launched = 0; ctx->last_has_loop = 0;
                return DCMT_OK;
            }
            dcmt_params q = *p;
            q.blur = blur;
            return run_chain_fused(ctx, kind, d_src, d_dst, rows, cols, batch, &q, st, sync_loop, x4);
        }
        if (!(p->flags & DCMT_FLAG_FORCE_STAGED) && big_enough && !labeled && kind >= 0 && rows >= 8 && cols >= 8 &&
            stop >= DCMT_STAGE_EXTEND) {
            dcmt_params q = *p;
            q.blur = blur;
            return run_chain_fused(ctx, kind, d_src, d_dst, rows, cols, batch, &q, st, sync_loop, nullptr, d_src16, in_scale, coef);
        }
        if (d_src16) {   // the staged kernels take f32: convert into scratch that nothing writes before they have read it
            const size_t n = (size_t)batch * rows * cols;
            hipLaunchKernelGGL(k_u16_to_f32, dim3(1024), dim3(256), 0, st, d_src16, ctx->pp[0], n, in_scale);
            d_src = ctx->pp[0];
        }
    }
    ctx->last_stream = st;
    ctx->last_batch = batch;
    ctx->last_apps_launched = 0;
    ctx->last_has_loop = 0;

    { const int erc = ensure_colstat(ctx); if (erc != DCMT_OK) return erc; }
    std::snprintf(ctx->last_path, sizeof ctx->last_path, "%s + k_fill31_v1 + k_post_v1 (staged tile kernels)", (d_labels && use_superpixel) ? "k_pre_labeled_v1" : "k_pre_v1");
    const int dump = stop <= DCMT_STAGE_CLOSE5 ? stop : 0;
    int stat_rows = (int)grid.y;                   // tile rows of the kernel that writes the column statistics
    if (d_labels && use_superpixel && few) {
        hipLaunchKernelGGL((k_pre_labeled_v1<FTH_FEW, TW>), fgrid, block, 0, st, d_src, d_labels, n_labels,
                           stop == DCMT_STAGE_FILL7 ? d_dst : ctx->x5, ctx->colstat, ctx->counters, d_dst, rows, cols,
                           p->max_depth, p->valid_thresh, kb, dump, coef);
        stat_rows = (int)fgrid.y;
    } else if (d_labels && use_superpixel) {
        hipLaunchKernelGGL((k_pre_labeled_v1<TH, TW>), grid, block, 0, st, d_src, d_labels, n_labels,
                           stop == DCMT_STAGE_FILL7 ? d_dst : ctx->x5, ctx->colstat, ctx->counters, d_dst, rows, cols,
                           p->max_depth, p->valid_thresh, kb, dump, coef);
    } else if (few) {
        hipLaunchKernelGGL((k_pre_v1<FTH_FEW, TW>), fgrid, block, 0, st, d_src,
                           stop == DCMT_STAGE_FILL7 ? d_dst : ctx->x5, ctx->colstat, ctx->counters, d_dst, rows, cols,
                           p->max_depth, p->valid_thresh, kb, dump, coef);
        stat_rows = (int)fgrid.y;
    } else {
        hipLaunchKernelGGL((k_pre_v1<TH, TW>), grid, block, 0, st, d_src,
                           stop == DCMT_STAGE_FILL7 ? d_dst : ctx->x5, ctx->colstat, ctx->counters, d_dst, rows, cols,
                           p->max_depth, p->valid_thresh, kb, dump, coef);
    }
    DCMT_HIP(ctx, hipGetLastError());
    if (stop <= DCMT_STAGE_FILL7) return DCMT_OK;

    // H6 + H7
    if (stop == DCMT_STAGE_EXTEND) {
        DCMT_FILL31(ctx->x5, d_dst, ctx->colstat, ctx->counters, rows, cols, p->valid_thresh, 0, 1, stat_rows)
        DCMT_HIP(ctx, hipGetLastError());
        return DCMT_OK;
    }
    DCMT_FILL31(ctx->x5, stop == DCMT_STAGE_FILL31 ? d_dst : ctx->pp[0], ctx->colstat, ctx->counters, rows, cols, p->valid_thresh, 0, 0, stat_rows)
    DCMT_HIP(ctx, hipGetLastError());
    if (stop == DCMT_STAGE_FILL31) return DCMT_OK;

    // H8
    ctx->last_has_loop = 1;
    int apps = 0;
    const int rc = fill_loop(ctx, batch, p, st, sync_loop, [&](int i) {
        DCMT_FILL31(ctx->pp[(i - 1) & 1], ctx->pp[i & 1], ctx->colstat, ctx->counters, rows, cols, p->valid_thresh, i, 0, stat_rows)
    }, &apps);
    if (rc != DCMT_OK && rc != DCMT_E_NOT_CONVERGED) return rc;
    ctx->last_apps_launched = apps;

    const int mode = stop <= DCMT_STAGE_FILLLOOP ? 8 : stop;
    if (few) hipLaunchKernelGGL((k_post_v1<FTH_FEW, TW>), fgrid, block, 0, st, ctx->pp[0], ctx->pp[1], d_dst, ctx->counters, apps,
                                rows, cols, p->max_depth, p->valid_thresh, blur, mode);
    else hipLaunchKernelGGL((k_post_v1<TH, TW>), grid, block, 0, st, ctx->pp[0], ctx->pp[1], d_dst, ctx->counters, apps,
                            rows, cols, p->max_depth, p->valid_thresh, blur, mode);
    DCMT_HIP(ctx, hipGetLastError());
    return rc;
#undef DCMT_FILL31
}

int ensure_host_staging(dcmt_ctx* ctx, bool labels)
{
    const size_t bytes = sizeof(float) * ctx->frame_elems * (size_t)ctx->max_batch;
    if (!ctx->d_in) DCMT_HIP(ctx, hipMalloc((void**)&ctx->d_in, bytes));
    if (!ctx->d_out) DCMT_HIP(ctx, hipMalloc((void**)&ctx->d_out, bytes));
    if (labels && !ctx->d_lab) DCMT_HIP(ctx, hipMalloc((void**)&ctx->d_lab, bytes));
    if (!ctx->own_stream) DCMT_HIP(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    return DCMT_OK;
}

int host_call(dcmt_ctx* ctx, const float* src, size_t srs, size_t sfs, const int32_t* labels, size_t lrs, size_t lfs,
              int n_labels, int use_superpixel, float* dst, size_t drs, size_t dfs, int rows, int cols, int batch,
              const dcmt_params* p, bool force_gaussian)
{
    int rc = check_params(ctx, src, dst, rows, cols, batch, p);
    if (rc != DCMT_OK) return rc;
    if (srs < sizeof(float) * (size_t)cols || drs < sizeof(float) * (size_t)cols) return DCMT_E_INVALID;
    if (labels && lrs < sizeof(int32_t) * (size_t)cols) return DCMT_E_INVALID;
    rc = ensure_host_staging(ctx, labels != nullptr);
    if (rc != DCMT_OK) return rc;
    hipStream_t st = ctx->own_stream;
    const size_t row_b = sizeof(float) * (size_t)cols, frame_b = row_b * rows;
    if (p->verbose == 1) {
        // what img_completion prints before it starts (LO :29, :41-50): the dimensions and the largest input value (start value 0.0, :22)
        std::printf("NUMERO ROWS, COLS: %d %d\n", rows, cols);
        for (int f = 0; f < batch; ++f) {
            float mx = 0.0f;
            for (int r = 0; r < rows; ++r) {
                const float* row = reinterpret_cast<const float*>(reinterpret_cast<const char*>(src) + f * sfs + r * srs);
                for (int c = 0; c < cols; ++c) mx = row[c] > mx ? row[c] : mx;
            }
            std::printf("max range is%g\n", (double)mx);               // operator<<(float): six significant digits, as %g
        }
    }
    for (int f = 0; f < batch; ++f) {
        if (srs == row_b)      // contiguous rows (the usual cv::Mat): one linear copy instead of a pitched one
            DCMT_HIP(ctx, hipMemcpyAsync((char*)ctx->d_in + f * frame_b, (const char*)src + f * sfs, frame_b, hipMemcpyHostToDevice, st));
        else
        DCMT_HIP(ctx, hipMemcpy2DAsync((char*)ctx->d_in + f * frame_b, row_b, (const char*)src + f * sfs, srs, row_b, rows,
                                       hipMemcpyHostToDevice, st));
        if (labels)
            DCMT_HIP(ctx, hipMemcpy2DAsync((char*)ctx->d_lab + f * frame_b, row_b, (const char*)labels + f * lfs, lrs, row_b,
                                           rows, hipMemcpyHostToDevice, st));
    }
    if (ctx->poison)   // DCMT_POISON=1: stale output can never pass for fresh output (tests)
        DCMT_HIP(ctx, hipMemsetAsync(ctx->d_out, 0xFF, frame_b * (size_t)batch, st));
    const int chain_rc = run_chain(ctx, ctx->d_in, labels ? ctx->d_lab : nullptr, n_labels, use_superpixel, ctx->d_out,
                                   rows, cols, batch, p, force_gaussian, st, true);
    if (chain_rc != DCMT_OK && chain_rc != DCMT_E_NOT_CONVERGED) return chain_rc;
    for (int f = 0; f < batch; ++f) {
        if (drs == row_b)
            DCMT_HIP(ctx, hipMemcpyAsync((char*)dst + f * dfs, (const char*)ctx->d_out + f * frame_b, frame_b, hipMemcpyDeviceToHost, st));
        else
            DCMT_HIP(ctx, hipMemcpy2DAsync((char*)dst + f * dfs, drs, (const char*)ctx->d_out + f * frame_b, row_b, row_b, rows,
                                           hipMemcpyDeviceToHost, st));
    }
    DCMT_HIP(ctx, hipStreamSynchronize(st));
    return chain_rc;
}

}  // namespace

extern "C" {

int dcmt_version(void) { return DCMT_VERSION; }

int dcmt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* dcmt_strerror(int s)
{
    switch (s) {
        case DCMT_OK: return "ok";
        case DCMT_E_INVALID: return "invalid argument";
        case DCMT_E_UNSUPPORTED: return "unsupported (bilateral blur: the reference's in-place cv::bilateralFilter call throws)";
        case DCMT_E_NOMEM: return "out of memory";
        case DCMT_E_HIP: return "HIP runtime error";
        case DCMT_E_NOT_CONVERGED: return "hole-closure loop hit max_fill_iters with holes left";
        case DCMT_E_NO_DEVICE: return "no gfx950 device";
        default: return "unknown status";
    }
}

void dcmt_k0_as_compiled(uint8_t k0[25])
{
    // reference img_completion.cpp:71-77: the first 25 bytes of `int d[5][5]` = {0,0,1,0,0, 0,1,...}
    // on a little-endian host: only byte 8 (row 1, col 3) and byte 24 (row 4, col 4) are non-zero
    std::memset(k0, 0, 25);
    k0[1 * 5 + 3] = 1;
    k0[4 * 5 + 4] = 1;
}

void dcmt_k0_diamond(uint8_t k0[25])
{
    static const uint8_t d[25] = {0, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 0};
    std::memcpy(k0, d, 25);
}

void dcmt_default_params(dcmt_params* p)
{
    std::memset(p, 0, sizeof(*p));
    p->max_depth = 100.0f;
    p->valid_thresh = 0.1f;
    dcmt_k0_as_compiled(p->k0);
    p->blur = DCMT_BLUR_GAUSSIAN;
    p->max_fill_iters = kMaxIters;
    p->spec_fill_iters = 1;
    p->stop_after = DCMT_STAGE_FINAL;
    p->verbose = 0;
    p->norm_lo = 0.0f;
    p->norm_hi = 100.0f;      // SL/main_sl.cpp:370 (the labeled call at :523 uses 80)
}

int dcmt_create(int device, int max_rows, int max_cols, int max_batch, dcmt_ctx** out)
{
    if (!out || max_rows < 1 || max_cols < 1 || max_batch < 1 || max_batch > 65535) return DCMT_E_INVALID;
    // a frame is one raw buffer resource addressed with 32-bit byte offsets, and kDropOffset (dcmt_kernels_fused.h) must lie
    // beyond its last byte for the "store that writes nothing" idiom: frame bytes <= 0x7fffffc0
    if ((size_t)max_rows * (size_t)max_cols > (size_t)0x1ffffff0) return DCMT_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return DCMT_E_NO_DEVICE;
    if (device < 0 || device >= n) return DCMT_E_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DCMT_E_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DCMT_E_NO_DEVICE;   // kernels are built for gfx950 only
    dcmt_ctx* ctx = new (std::nothrow) dcmt_ctx();
    if (!ctx) return DCMT_E_NOMEM;
    ctx->device = device;
    ctx->max_rows = max_rows; ctx->max_cols = max_cols; ctx->max_batch = max_batch;
    ctx->frame_elems = (size_t)max_rows * max_cols;
    { const char* e = std::getenv("DCMT_POISON"); ctx->poison = e && e[0] == '1'; }
    { const char* e = std::getenv("DCMT_CHUNK"); if (e) ctx->chunk = std::atoi(e); }
    { const char* e = std::getenv("DCMT_XCD_MAP"); if (e) ctx->xcd_map = std::atoi(e); }
    { const char* e = std::getenv("DCMT_WIDE"); if (e) ctx->wide = std::atoi(e); }
    { const char* e = std::getenv("DCMT_FUSE_FP"); if (e) ctx->fuse_fp = std::atoi(e); }
    { const char* e = std::getenv("DCMT_FP_PAIR"); if (e) ctx->fp_pair = std::atoi(e); }
    { const char* e = std::getenv("DCMT_FP_Q16"); if (e) ctx->fp_q16 = std::atoi(e); }
    { const char* e = std::getenv("DCMT_Q16_MIN_WAVES"); if (e) ctx->q16_min_waves = std::atoi(e); }
    { const char* e = std::getenv("DCMT_Q16_BREG"); if (e) ctx->q16_breg = std::atoi(e); }
    { const char* e = std::getenv("DCMT_FP_H"); if (e) ctx->fp_h = std::atoi(e); }
    { const char* e = std::getenv("DCMT_ASSUME_FILLED"); if (e) ctx->assume_filled = std::atoi(e); }
    { const char* e = std::getenv("DCMT_TOP_TABLE"); if (e) ctx->top_table = std::atoi(e); }
    { const char* e = std::getenv("DCMT_PAIR"); if (e) ctx->pair = std::atoi(e); }
    { const char* e = std::getenv("DCMT_BANDS"); if (e) ctx->bands = std::atoi(e); }
    { const char* e = std::getenv("DCMT_FBANDS"); if (e) ctx->fbands = std::atoi(e); }
    { const char* e = std::getenv("DCMT_MIN_FUSED_BATCH"); if (e) ctx->min_fused_batch = std::atoi(e); }
    { const char* e = std::getenv("DCMT_LABEL_PAIRS"); if (e) ctx->label_pairs = std::atoi(e); }
    { const char* e = std::getenv("DCMT_LABEL_GROUP"); if (e) ctx->label_group = std::atoi(e); }
    DeviceGuard dev_guard_(ctx);                    // allocate on the context's device, leave the caller's current device as it was
    auto fail = [&](int rc) { dcmt_destroy(ctx); return rc; };
    if (dev_guard_.rc != DCMT_OK) return fail(dev_guard_.rc);
    const size_t plane = sizeof(float) * ctx->frame_elems * (size_t)max_batch;
    // two planes: X6 (x5) is dead once the first fill application of the hole-closure loop has read it, and that application writes pp[0], so
    // x5 shares pp[1] (the second application's output); X4 of the label-masked stage and the staged kernels' uint16 conversion live in
    // pp[0], which nothing writes before they have been read
    if (hipMalloc((void**)&ctx->pp[0], plane) != hipSuccess) return fail(DCMT_E_NOMEM);
    if (hipMalloc((void**)&ctx->pp[1], plane) != hipSuccess) return fail(DCMT_E_NOMEM);
    ctx->x5 = ctx->pp[1];
    if (hipMalloc((void**)&ctx->counters, sizeof(int) * (size_t)kCntStride * max_batch) != hipSuccess) return fail(DCMT_E_NOMEM);
    if (hipMalloc((void**)&ctx->q16_bad, sizeof(int) * kQ16Flags) != hipSuccess) return fail(DCMT_E_NOMEM);
    if (hipMemset(ctx->q16_bad, 0, sizeof(int) * kQ16Flags) != hipSuccess) return fail(DCMT_E_HIP);
    if (hipHostMalloc((void**)&ctx->q16_seen, 64, hipHostMallocMapped) != hipSuccess) return fail(DCMT_E_NOMEM);
    *ctx->q16_seen = 0;
    if (hipHostGetDevicePointer((void**)&ctx->q16_seen_dev, ctx->q16_seen, 0) != hipSuccess) return fail(DCMT_E_HIP);
    // (first, last) table: one slot per frame and row band.  Bands are only chosen while frames x strips x bands stays near one
    // round of waves (run_chain_fused), so frames x bands <= max_batch + 2560; an explicit DCMT_BANDS may go up to kMaxBands each.
    const size_t tb_slots = ctx->bands > 0 ? (size_t)max_batch * kMaxBands : std::min<size_t>((size_t)max_batch * kMaxBands, (size_t)max_batch + 2560);
    if (hipMalloc((void**)&ctx->tb, sizeof(int) * 2 * (size_t)max_cols * tb_slots) != hipSuccess) return fail(DCMT_E_NOMEM);
    if (hipMalloc((void**)&ctx->norm_stats, sizeof(uint32_t) * 2 * (size_t)max_batch) != hipSuccess) return fail(DCMT_E_NOMEM);
    if (hipMemset(ctx->norm_stats, 0, sizeof(uint32_t) * 2 * (size_t)max_batch) != hipSuccess) return fail(DCMT_E_HIP);
    if (hipMalloc((void**)&ctx->norm_coef, sizeof(float) * 2 * (size_t)max_batch) != hipSuccess) return fail(DCMT_E_NOMEM);
    if (hipHostMalloc((void**)&ctx->h_counters, sizeof(int) * (size_t)kCntStride * max_batch, hipHostMallocDefault) != hipSuccess)
        return fail(DCMT_E_NOMEM);
    *out = ctx;
    return DCMT_OK;
}

void dcmt_destroy(dcmt_ctx* ctx)
{
    if (!ctx) return;
    DeviceGuard dev_guard_(ctx);
    if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); (void)hipStreamDestroy(ctx->own_stream); }
    (void)hipFree(ctx->pp[0]); (void)hipFree(ctx->pp[1]);
    (void)hipFree(ctx->colstat); (void)hipFree(ctx->counters); (void)hipFree(ctx->tb);
    (void)hipFree(ctx->x6q); (void)hipFree(ctx->q16_bad); if (ctx->q16_seen) (void)hipHostFree(ctx->q16_seen);
    (void)hipFree(ctx->norm_stats); (void)hipFree(ctx->norm_coef);
    (void)hipFree(ctx->d_in); (void)hipFree(ctx->d_out); (void)hipFree(ctx->d_lab);
    (void)hipFree(ctx->bb_min); (void)hipFree(ctx->bb_max); (void)hipFree(ctx->winner);
    (void)hipFree(ctx->slic_cells); (void)hipFree(ctx->slic_centers[0]); (void)hipFree(ctx->slic_centers[1]);
    (void)hipFree(ctx->slic_sums);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    for (auto e : ctx->tev) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

int dcmt_complete_f32(dcmt_ctx* ctx, const float* src, size_t srs, size_t sfs, float* dst, size_t drs, size_t dfs,
                      int rows, int cols, int batch, const dcmt_params* params)
{
    DCMT_ON_DEVICE(ctx);
    return host_call(ctx, src, srs, sfs, nullptr, 0, 0, 0, 0, dst, drs, dfs, rows, cols, batch, params, false);
}

int dcmt_complete_labeled_f32(dcmt_ctx* ctx, const float* src, size_t srs, size_t sfs, const int32_t* labels, size_t lrs,
                              size_t lfs, int n_labels, float* dst, size_t drs, size_t dfs, int rows, int cols, int batch,
                              const dcmt_params* params, int use_superpixel)
{
    DCMT_ON_DEVICE(ctx);
    if (!labels) return DCMT_E_INVALID;
    return host_call(ctx, src, srs, sfs, labels, lrs, lfs, n_labels, use_superpixel, dst, drs, dfs, rows, cols, batch,
                     params, true);
}

int dcmt_complete_f32_dev(dcmt_ctx* ctx, const float* d_src, float* d_dst, int rows, int cols, int batch,
                          const dcmt_params* params, void* stream)
{
    DCMT_ON_DEVICE(ctx);
    int rc = check_params(ctx, d_src, d_dst, rows, cols, batch, params);
    if (rc != DCMT_OK) return rc;
    return run_chain(ctx, d_src, nullptr, 0, 0, d_dst, rows, cols, batch, params, false, (hipStream_t)stream, false);
}

int dcmt_complete_u16_dev(dcmt_ctx* ctx, const uint16_t* d_src, float scale, float* d_dst, int rows, int cols, int batch,
                          const dcmt_params* params, void* stream)
{
    DCMT_ON_DEVICE(ctx);
    int rc = check_params(ctx, d_src, d_dst, rows, cols, batch, params);
    if (rc != DCMT_OK) return rc;
    return run_chain(ctx, nullptr, nullptr, 0, 0, d_dst, rows, cols, batch, params, false, (hipStream_t)stream, false, d_src, scale);
}

int dcmt_complete_labeled_f32_dev(dcmt_ctx* ctx, const float* d_src, const int32_t* d_labels, int n_labels, float* d_dst,
                                  int rows, int cols, int batch, const dcmt_params* params, int use_superpixel, void* stream)
{
    DCMT_ON_DEVICE(ctx);
    int rc = check_params(ctx, d_src, d_dst, rows, cols, batch, params);
    if (rc != DCMT_OK) return rc;
    if (!d_labels) return DCMT_E_INVALID;
    return run_chain(ctx, d_src, d_labels, n_labels, use_superpixel, d_dst, rows, cols, batch, params, true,
                     (hipStream_t)stream, false);
}

int dcmt_project_points_dev(dcmt_ctx* ctx, const float* d_points, const int32_t* d_offsets, int n_points, int batch,
                            const float T[16], const float P[12], float* d_sparse, int rows, int cols, void* stream)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !d_offsets || !T || !P || !d_sparse || n_points < 0 || (n_points > 0 && !d_points)) return DCMT_E_INVALID;
    if ((uintptr_t)d_points % 16 != 0) return DCMT_E_INVALID;           // the 16-byte point records are read whole
    if (rows < 1 || cols < 1 || batch < 1 || batch > ctx->max_batch || rows > ctx->max_rows || cols > ctx->max_cols) return DCMT_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    ProjMats M;
    std::memcpy(M.T, T, sizeof(float) * 12);       // the bottom row of T is never used (SL :483-485)
    std::memcpy(M.P, P, sizeof(float) * 12);
    // the winner plane: tags of generation g = (g << idx_bits) | point index, g >= 1 (0 = the cleared plane).  It is cleared when it
    // is (re)allocated, when a call needs more index bits than its layout has, and when the generations run out.
    const size_t n_px = (size_t)batch * rows * cols;
    int need_bits = 1;
    while (need_bits < 31 && ((size_t)1 << need_bits) <= (size_t)n_points) ++need_bits;
    if (need_bits > 30) return DCMT_E_INVALID;                             // (2^30 points per call: 16 GiB of records)
    if (n_px > ctx->winner_elems) {
        (void)hipFree(ctx->winner); ctx->winner = nullptr; ctx->winner_elems = 0;
        DCMT_HIP(ctx, hipMalloc((void**)&ctx->winner, sizeof(unsigned) * n_px));
        ctx->winner_elems = n_px; ctx->winner_bits = 0; ctx->winner_gen = 0;
    }
    const bool relayout = need_bits > ctx->winner_bits;
    if (relayout) ctx->winner_bits = need_bits < 24 ? 24 : need_bits;       // (room for 16 M points per call before the next re-layout)
    const unsigned gen_max = (1u << (32 - ctx->winner_bits)) - 1u;
    if (relayout || ctx->winner_gen == 0 || ctx->winner_gen >= gen_max) {
        DCMT_HIP(ctx, hipMemsetAsync(ctx->winner, 0, sizeof(unsigned) * ctx->winner_elems, st));
        ctx->winner_gen = 0;
    }
    const unsigned gen_tag = ++ctx->winner_gen << ctx->winner_bits;
    unsigned* winner = ctx->winner;
    if (n_points > 0)
        hipLaunchKernelGGL(k_project_scatter, dim3((n_points + 255) / 256), dim3(256), 0, st, d_points, d_offsets, n_points, batch, M,
                           winner, rows, cols, gen_tag);
    if (n_px % 4 == 0 && (uintptr_t)d_sparse % 16 == 0)
        hipLaunchKernelGGL(k_project_resolve<4>, dim3((unsigned)((n_px / 4 + 255) / 256)), dim3(256), 0, st, d_points, M, winner, d_sparse, n_px, gen_tag, ctx->winner_bits);
    else if (n_px % 2 == 0 && (uintptr_t)d_sparse % 8 == 0)
        hipLaunchKernelGGL(k_project_resolve<2>, dim3((unsigned)((n_px / 2 + 255) / 256)), dim3(256), 0, st, d_points, M, winner, d_sparse, n_px, gen_tag, ctx->winner_bits);
    else
        hipLaunchKernelGGL(k_project_resolve<1>, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, st, d_points, M, winner, d_sparse, n_px, gen_tag, ctx->winner_bits);
    DCMT_HIP(ctx, hipGetLastError());
    return DCMT_OK;
}

void dcmt_default_stereo_params(dcmt_stereo_params* p)
{
    p->baseline = 0.54f;          // SL/main_sl.cpp:847
    p->focal = 9.597910e+02f;     // :848
    p->damp = 500.0f;             // :808
    p->max_depth = 100.0f;        // :876
    p->iterations = 4;            // :805
}

int dcmt_stereo_refine_dev(dcmt_ctx* ctx, const float* d_depth, const uint8_t* d_left, const uint8_t* d_right, float* d_refined,
                           int rows, int cols, int batch, const dcmt_stereo_params* params, void* stream)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !d_depth || !d_left || !d_right || !d_refined || !params) return DCMT_E_INVALID;
    if (rows < 1 || cols < 1 || batch < 1 || batch > ctx->max_batch || rows > ctx->max_rows || cols > ctx->max_cols) return DCMT_E_INVALID;
    if (params->iterations > 1000) return DCMT_E_INVALID;
    StereoP P{params->baseline, params->focal, params->damp, params->max_depth, params->iterations < 0 ? 4 : params->iterations};
    if (rows > 65535 || batch > 65535) return DCMT_E_INVALID;                  // grid dimensions y, z
    const dim3 sg((cols + 255) / 256, rows, batch), sg1(1, rows, batch);
    if (cols + 4 <= 48 * 1024)      // the right-image row fits the workgroup's LDS: one workgroup stages it and walks the whole row
        hipLaunchKernelGGL(k_stereo_refine<true>, sg1, dim3(256), (size_t)cols + 4, (hipStream_t)stream, d_depth, d_left, d_right, d_refined, rows, cols, batch, P);
    else
        hipLaunchKernelGGL(k_stereo_refine<false>, sg, dim3(256), 0, (hipStream_t)stream, d_depth, d_left, d_right, d_refined, rows, cols, batch, P);
    DCMT_HIP(ctx, hipGetLastError());
    return DCMT_OK;
}

int dcmt_slic_num_centers(int rows, int cols, int step)
{
    if (rows < 1 || cols < 1 || step < 1) return 0;
    int nx = 0, ny = 0;
    for (int i = step; i < cols - step / 2; i += step) ++nx;     // slic.cpp:33-34
    for (int j = step; j < rows - step / 2; j += step) ++ny;
    return nx * ny;
}

int dcmt_slic_labels_dev(dcmt_ctx* ctx, const uint8_t* d_lab, int rows, int cols, int batch, int step, int nc,
                         int32_t* d_labels, double* d_centers, void* stream)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !d_lab || !d_labels) return DCMT_E_INVALID;
    if (rows < 1 || cols < 1 || batch < 1 || batch > ctx->max_batch || rows > ctx->max_rows || cols > ctx->max_cols) return DCMT_E_INVALID;
    if (step < 6 || nc < 1) return DCMT_E_INVALID;
    const int n = dcmt_slic_num_centers(rows, cols, step);
    hipStream_t st = (hipStream_t)stream;
    const size_t px = (size_t)batch * rows * cols;
    if (n == 0) { DCMT_HIP(ctx, hipMemsetAsync(d_labels, 0xFF, sizeof(int32_t) * px, st)); return DCMT_OK; }
    int cell_px = step;                                                    // cells of step x step pixels
    { const char* e = std::getenv("DCMT_SLIC_CELL_SCALE"); if (e && std::atoi(e) > 1) cell_px = step * std::atoi(e); }   // tests: crowded cells
    const int gx = (cols - 1) / cell_px + 1, gy = (rows - 1) / cell_px + 1;
    const size_t cells = (size_t)gx * gy;
    // two cell sets (the assignment reads one while the next centres are binned into the other); per set [batch][cells] counts
    // with the [batch] overflow flags right behind, and [batch][cells][kSlicCellCap] centre indices
    if (cells > ctx->slic_cell_cap) {
        (void)hipFree(ctx->slic_cells); ctx->slic_cells = nullptr; ctx->slic_cell_cap = 0;
        DCMT_HIP(ctx, hipMalloc((void**)&ctx->slic_cells, 2 * sizeof(int) * ((size_t)ctx->max_batch * cells * (1 + kSlicCellCap) + ctx->max_batch)));
        ctx->slic_cell_cap = cells;
    }
    const size_t n_cnt = (size_t)batch * cells + batch;
    int* set_cnt[2] = {ctx->slic_cells, ctx->slic_cells + n_cnt};
    int* set_ovf[2] = {set_cnt[0] + (size_t)batch * cells, set_cnt[1] + (size_t)batch * cells};
    int* set_list[2] = {ctx->slic_cells + 2 * n_cnt, ctx->slic_cells + 2 * n_cnt + (size_t)batch * cells * kSlicCellCap};
    if ((size_t)n > ctx->slic_center_cap) {
        (void)hipFree(ctx->slic_centers[0]); (void)hipFree(ctx->slic_centers[1]); (void)hipFree(ctx->slic_sums);
        ctx->slic_centers[0] = ctx->slic_centers[1] = nullptr; ctx->slic_sums = nullptr; ctx->slic_center_cap = 0;
        const size_t cn = (size_t)n * ctx->max_batch;
        DCMT_HIP(ctx, hipMalloc((void**)&ctx->slic_centers[0], sizeof(double) * 5 * cn));
        DCMT_HIP(ctx, hipMalloc((void**)&ctx->slic_centers[1], sizeof(double) * 5 * cn));
        DCMT_HIP(ctx, hipMalloc((void**)&ctx->slic_sums, sizeof(unsigned long long) * 6 * cn));
        ctx->slic_center_cap = (size_t)n;
    }
    DCMT_HIP(ctx, hipMemsetAsync(d_labels, 0xFF, sizeof(int32_t) * px, st));                    // clusters = -1 (slic.cpp:24)
    DCMT_HIP(ctx, hipMemsetAsync(set_cnt[0], 0, sizeof(int) * 2 * n_cnt, st));                  // both cell sets' counts and flags
    DCMT_HIP(ctx, hipMemsetAsync(ctx->slic_sums, 0, sizeof(unsigned long long) * 6 * (size_t)n * batch, st));   // every iteration leaves them zeroed
    hipLaunchKernelGGL(k_slic_init, dim3((n + 63) / 64, batch), dim3(64), 0, st, d_lab, ctx->slic_centers[0], rows, cols, step, n,
                       set_cnt[0], set_list[0], set_ovf[0], cell_px, gx, gy);
    const size_t nb_threads = std::max((size_t)n * batch, n_cnt);
    // tile height: the tallest the step allows, unless that leaves the GPU short of workgroups (a caller streaming single frames:
    // 114 tiles of 64 x 64 per 1216 x 352 image for 256 CUs) -- then shorter tiles, more of them, shorter columns per thread
    int th = slic_tile_rows(cell_px);
    { const char* e = std::getenv("DCMT_SLIC_TH"); if (e && (std::atoi(e) == 16 || std::atoi(e) == 32 || std::atoi(e) == 64) && std::atoi(e) <= th) th = std::atoi(e);
      else while (th > 16 && (size_t)((cols + kSlicTW - 1) / kSlicTW) * ((rows + th - 1) / th) * batch < 1024) th /= 2; }
    for (int it = 0; it < 10; ++it) {                                                           // NR_ITERATIONS (slic.h:20)
        double* cur = ctx->slic_centers[it & 1];
        double* nxt = ctx->slic_centers[(it + 1) & 1];
        const int a = it & 1, b = a ^ 1;
        if (th == 64)
            hipLaunchKernelGGL(k_slic_assign<64>, dim3((cols + kSlicTW - 1) / kSlicTW, (rows + 63) / 64, batch), dim3(256), 0, st, d_lab, cur,
                               set_cnt[a], set_list[a], set_ovf[a], d_labels, ctx->slic_sums, rows, cols, step, nc, n, gx, gy, cell_px);
        else if (th == 32)
            hipLaunchKernelGGL(k_slic_assign<32>, dim3((cols + kSlicTW - 1) / kSlicTW, (rows + 31) / 32, batch), dim3(256), 0, st, d_lab, cur,
                               set_cnt[a], set_list[a], set_ovf[a], d_labels, ctx->slic_sums, rows, cols, step, nc, n, gx, gy, cell_px);
        else
            hipLaunchKernelGGL(k_slic_assign<16>, dim3((cols + kSlicTW - 1) / kSlicTW, (rows + 15) / 16, batch), dim3(256), 0, st, d_lab, cur,
                               set_cnt[a], set_list[a], set_ovf[a], d_labels, ctx->slic_sums, rows, cols, step, nc, n, gx, gy, cell_px);
        hipLaunchKernelGGL(k_slic_norm_bin, dim3((unsigned)((nb_threads + 255) / 256)), dim3(256), 0, st, ctx->slic_sums, nxt, n, batch,
                           set_cnt[b], set_list[b], set_ovf[b], set_cnt[a], (int)n_cnt, cell_px, gx, gy);
        DCMT_HIP(ctx, hipGetLastError());
    }
    if (d_centers)       // ten iterations: the final centres are back in buffer 0
        DCMT_HIP(ctx, hipMemcpyAsync(d_centers, ctx->slic_centers[0], sizeof(double) * 5 * (size_t)n * batch, hipMemcpyDeviceToDevice, st));
    return DCMT_OK;
}

// ---- host variants of N2 / N3 / N4: temporary device buffers, synchronous ----------------------------------------
namespace {
struct DevBuf {                                    // freed when the call returns, whatever path it takes
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(dcmt_ctx* ctx, size_t bytes) { DCMT_HIP(ctx, hipMalloc(&p, bytes ? bytes : 1)); return DCMT_OK; }
};
int host_stream(dcmt_ctx* ctx, hipStream_t* st)
{
    if (!ctx->own_stream) DCMT_HIP(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    *st = ctx->own_stream;
    return DCMT_OK;
}
}  // namespace

int dcmt_project_points(dcmt_ctx* ctx, const float* points, int n_points, const float T[16], const float P[12],
                        float* sparse, size_t srs, int rows, int cols)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !sparse || !T || !P || n_points < 0 || (n_points > 0 && !points) || rows < 1 || cols < 1) return DCMT_E_INVALID;
    if (srs < sizeof(float) * (size_t)cols) return DCMT_E_INVALID;
    hipStream_t st;
    int rc = host_stream(ctx, &st);
    if (rc != DCMT_OK) return rc;
    DevBuf dp, doff, dout;
    const size_t row_b = sizeof(float) * (size_t)cols;
    if ((rc = dp.alloc(ctx, sizeof(float) * 4 * (size_t)n_points)) != DCMT_OK || (rc = doff.alloc(ctx, sizeof(int32_t) * 2)) != DCMT_OK ||
        (rc = dout.alloc(ctx, row_b * rows)) != DCMT_OK) return rc;
    const int32_t off[2] = {0, n_points};
    if (n_points) DCMT_HIP(ctx, hipMemcpyAsync(dp.p, points, sizeof(float) * 4 * (size_t)n_points, hipMemcpyHostToDevice, st));
    DCMT_HIP(ctx, hipMemcpyAsync(doff.p, off, sizeof(off), hipMemcpyHostToDevice, st));
    rc = dcmt_project_points_dev(ctx, (const float*)dp.p, (const int32_t*)doff.p, n_points, 1, T, P, (float*)dout.p, rows, cols, st);
    if (rc != DCMT_OK) return rc;
    DCMT_HIP(ctx, hipMemcpy2DAsync(sparse, srs, dout.p, row_b, row_b, rows, hipMemcpyDeviceToHost, st));
    DCMT_HIP(ctx, hipStreamSynchronize(st));
    return DCMT_OK;
}

int dcmt_slic_labels(dcmt_ctx* ctx, const uint8_t* lab, size_t lrs, int rows, int cols, int step, int nc, int32_t* labels, double* centers)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !lab || !labels || rows < 1 || cols < 1 || lrs < 3 * (size_t)cols) return DCMT_E_INVALID;
    hipStream_t st;
    int rc = host_stream(ctx, &st);
    if (rc != DCMT_OK) return rc;
    const int n = dcmt_slic_num_centers(rows, cols, step);
    DevBuf dl, dlab, dc;
    const size_t row_b = 3 * (size_t)cols, px = (size_t)rows * cols;
    if ((rc = dl.alloc(ctx, row_b * rows)) != DCMT_OK || (rc = dlab.alloc(ctx, sizeof(int32_t) * px)) != DCMT_OK ||
        (rc = dc.alloc(ctx, sizeof(double) * 5 * (size_t)(n > 0 ? n : 1))) != DCMT_OK) return rc;
    DCMT_HIP(ctx, hipMemcpy2DAsync(dl.p, row_b, lab, lrs, row_b, rows, hipMemcpyHostToDevice, st));
    rc = dcmt_slic_labels_dev(ctx, (const uint8_t*)dl.p, rows, cols, 1, step, nc, (int32_t*)dlab.p, centers ? (double*)dc.p : nullptr, st);
    if (rc != DCMT_OK) return rc;
    DCMT_HIP(ctx, hipMemcpyAsync(labels, dlab.p, sizeof(int32_t) * px, hipMemcpyDeviceToHost, st));
    if (centers && n > 0) DCMT_HIP(ctx, hipMemcpyAsync(centers, dc.p, sizeof(double) * 5 * (size_t)n, hipMemcpyDeviceToHost, st));
    DCMT_HIP(ctx, hipStreamSynchronize(st));
    return DCMT_OK;
}

int dcmt_stereo_refine(dcmt_ctx* ctx, const float* depth, size_t drs, const uint8_t* left, size_t lrs, const uint8_t* right, size_t rrs,
                       float* refined, size_t ors, int rows, int cols, const dcmt_stereo_params* params)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !depth || !left || !right || !refined || !params || rows < 1 || cols < 1) return DCMT_E_INVALID;
    if (drs < sizeof(float) * (size_t)cols || ors < sizeof(float) * (size_t)cols || lrs < (size_t)cols || rrs < (size_t)cols) return DCMT_E_INVALID;
    hipStream_t st;
    int rc = host_stream(ctx, &st);
    if (rc != DCMT_OK) return rc;
    DevBuf dd, dl, dr, dout;
    const size_t frow = sizeof(float) * (size_t)cols, brow = (size_t)cols;
    if ((rc = dd.alloc(ctx, frow * rows)) != DCMT_OK || (rc = dl.alloc(ctx, brow * rows)) != DCMT_OK || (rc = dr.alloc(ctx, brow * rows)) != DCMT_OK ||
        (rc = dout.alloc(ctx, frow * rows)) != DCMT_OK) return rc;
    DCMT_HIP(ctx, hipMemcpy2DAsync(dd.p, frow, depth, drs, frow, rows, hipMemcpyHostToDevice, st));
    DCMT_HIP(ctx, hipMemcpy2DAsync(dl.p, brow, left, lrs, brow, rows, hipMemcpyHostToDevice, st));
    DCMT_HIP(ctx, hipMemcpy2DAsync(dr.p, brow, right, rrs, brow, rows, hipMemcpyHostToDevice, st));
    rc = dcmt_stereo_refine_dev(ctx, (const float*)dd.p, (const uint8_t*)dl.p, (const uint8_t*)dr.p, (float*)dout.p, rows, cols, 1, params, st);
    if (rc != DCMT_OK) return rc;
    DCMT_HIP(ctx, hipMemcpy2DAsync(refined, ors, dout.p, frow, frow, rows, hipMemcpyDeviceToHost, st));
    DCMT_HIP(ctx, hipStreamSynchronize(st));
    return DCMT_OK;
}

static int read_counters(dcmt_ctx* ctx)
{
    DCMT_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->counters, sizeof(int) * (size_t)ctx->last_batch * kCntStride,
                                 hipMemcpyDeviceToHost, ctx->last_stream));
    DCMT_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    return DCMT_OK;
}

int dcmt_last_fill_iters(dcmt_ctx* ctx, int* out, int n)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !out || n < 0 || n > ctx->last_batch) return DCMT_E_INVALID;
    if (!ctx->last_has_loop) return DCMT_E_INVALID;
    int rc = read_counters(ctx);
    if (rc != DCMT_OK) return rc;
    for (int f = 0; f < n; ++f) {
        const int* c = ctx->h_counters + (size_t)f * kCntStride;
        int a = 0;
        while (a < ctx->last_apps_launched && c[1 + a] > 0) ++a;
        if (c[1 + a] > 0) { out[f] = -1; rc = DCMT_E_NOT_CONVERGED; }
        else out[f] = a + 1;
    }
    return rc;
}

int dcmt_last_holes_after_extend(dcmt_ctx* ctx, int* out, int n)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !out || n < 0 || n > ctx->last_batch) return DCMT_E_INVALID;
    int rc = read_counters(ctx);
    if (rc != DCMT_OK) return rc;
    for (int f = 0; f < n; ++f) out[f] = ctx->h_counters[(size_t)f * kCntStride];
    return DCMT_OK;
}

int dcmt_last_hip_error(const dcmt_ctx* ctx) { return ctx ? ctx->last_hip_error : 0; }

const char* dcmt_last_path(const dcmt_ctx* ctx) { return ctx ? ctx->last_path : ""; }

int dcmt_set_kernel_timing(dcmt_ctx* ctx, int on)
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx) return DCMT_E_INVALID;
    if (on)
        for (auto& e : ctx->tev)
            if (!e) DCMT_HIP(ctx, hipEventCreate(&e));
    ctx->timing = on != 0;
    ctx->tev_valid = 0;
    return DCMT_OK;
}

int dcmt_last_kernel_times(dcmt_ctx* ctx, float ms[DCMT_N_KERNEL_TIMES])
{
    DCMT_ON_DEVICE(ctx);
    if (!ctx || !ms || !ctx->tev_valid) return DCMT_E_INVALID;
    DCMT_HIP(ctx, hipEventSynchronize(ctx->tev[4]));
    for (int i = 0; i < DCMT_N_KERNEL_TIMES; ++i) DCMT_HIP(ctx, hipEventElapsedTime(&ms[i], ctx->tev[i], ctx->tev[i + 1]));
    return DCMT_OK;
}

}  // extern "C"
