/*
 * dcmt.h -- C ABI of the MI355X-native (gfx950 / CDNA4) morphological depth-completion
 * cascade.  This is the drop-in boundary for the ONE hot path of
 * PatrizioPerugini/depth_completion_MT:
 *
 *   void img_completion(const cv::Mat&, cv::Mat&, const bool&, const std::string&)
 *        reference: src/DC_lidar_only/img_completion.cpp:17-20
 *   void interpolate_with_superpixels(Slic&, const cv::Mat&, cv::Mat&, const std::string&, int)
 *        reference: src/DC_lidar_camera/img_completion_lc.cpp:34-38
 *
 * The reference has no FFI or plugin registry: those two free-function signatures ARE the
 * interface, and the header that declares them (img_completion.h) is missing from the
 * reference repository.  include/img_completion.h in this repo is that header; it is a
 * thin C++ shim over the entry points below (INTEGRATION.md shows the binding).
 *
 * Plain C: no C++ types, no exceptions, no torch types cross this boundary.  Every
 * function returns DCMT_OK (0) or a negative dcmt_status.  A dcmt_ctx is bound to one
 * GPU and owns all scratch memory; it must not be used from two threads at once (one ctx
 * per GPU per host thread -- frames are independent, so multi-GPU is one ctx per device).
 * Successive *_dev calls on one ctx must be ordered on the device too: the same stream, or streams the caller has ordered with
 * events (a call's kernels use the scratch memory -- and clear flags -- the next call's kernels rely on).  Work that should overlap
 * goes to two contexts.
 *
 * Devices and threads: every entry point that takes a ctx makes ctx's device current for the
 * duration of the call and restores the calling thread's current device before it returns, so
 * one process may drive all GPUs of a node from one host thread per GPU (or from one thread,
 * ctx after ctx) without ever calling hipSetDevice itself.  Device pointers and the stream
 * passed to a *_dev entry point must belong to ctx's device; stream == NULL is that device's
 * default stream.
 *
 * Values: frames must be finite (no NaN, no +-Inf).  The library is built with
 * -ffinite-math-only (its max/min networks drop the NaN-quieting pass), so a NaN or Inf in an
 * input frame gives an unspecified (but memory-safe) result in the pixels its windows reach.
 * The reference's own result for such pixels depends on the min/max flavour of the OpenCV
 * build (SIMD vs scalar), so there is nothing to be bit-exact with.  -0.0 is treated as 0.0.
 *
 * Depth grids: frames whose depths are all multiples of 1/256 m below 120 m -- what a KITTI depth PNG / 256 holds,
 * DC_lidar_only/main.cpp:75-82 -- let large device batches keep their intermediate image as 16-bit integers.  The library
 * finds that out itself on the device (and repeats the step with its f32 kernels when a frame turns out otherwise); results
 * are the same bits either way, and no promise about the values is asked of the caller.  What it costs when the depths are NOT
 * such multiples: a call that makes the attempt runs both of its large kernels twice (the 16-bit attempt, then the f32 kernels
 * behind the flag the attempt raised), and the context then skips the attempt for its next 63 calls before trying once more --
 * arbitrary f32 depths pay one double run in 64 calls.  Depths beyond 119.996 m (inverted values below -20 m: outside the 15-bit code
 * range) count as "not on the grid" as well, also on the uint16 entry point (payloads above 30719).
 *
 * In place: d_dst may be d_src (the reference's function is in place by signature: dense = sparse.clone(), :27), or overlap
 * it; the library then never takes the 16-bit attempt (its f32 kernels read src completely before dst is first written).
 * Empty pixels: the sign of a zero in an output pixel that stays empty (an all-empty frame) is +0.0 where the input held +0.0; an
 * input -0.0 counts as 0.0 (empty) and may come out as either zero.
 */
#ifndef DCMT_H
#define DCMT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCMT_VERSION 120 /* 0.1.2: dcmt_last_path; verbose == 2 (hole counts only: interpolate_with_superpixels) */

typedef struct dcmt_ctx dcmt_ctx;

typedef enum {
    DCMT_OK              = 0,
    DCMT_E_INVALID       = -1, /* bad argument (null pointer, size beyond the ctx limits, empty k0) */
    DCMT_E_UNSUPPORTED   = -2, /* blur == DCMT_BLUR_BILATERAL: the reference's call throws too (img_completion.cpp:174) */
    DCMT_E_NOMEM         = -3, /* device or host allocation failed */
    DCMT_E_HIP           = -4, /* a HIP runtime call failed; dcmt_last_hip_error() has the code */
    DCMT_E_NOT_CONVERGED = -5, /* the hole-closure loop (img_completion.cpp:146-166) hit max_fill_iters with holes left */
    DCMT_E_NO_DEVICE     = -6  /* no gfx950 device visible / wrong architecture */
} dcmt_status;

typedef enum {
    DCMT_BLUR_NONE      = 0, /* any blur_type string other than the two below */
    DCMT_BLUR_GAUSSIAN  = 1, /* "gaussian": GaussianBlur 5x5 sigma 0 + masked select (img_completion.cpp:176-189) */
    DCMT_BLUR_BILATERAL = 2  /* "bilateral": rejected with DCMT_E_UNSUPPORTED */
} dcmt_blur;

/* Stages of the cascade, for `stop_after` (parity probes; 11 = the whole chain). */
typedef enum {
    DCMT_STAGE_NORMALIZE = 1, /* only with DCMT_FLAG_NORMALIZE: the min-max normalised frames themselves
                                 (cv::normalize in front of the path, DC_stereo_lidar/main_sl.cpp:370, :523) */
    DCMT_STAGE_INVERT   = 2,  /* img_completion.cpp:55-67   */
    DCMT_STAGE_DILATE_K = 3,  /* :71-80   first dilate, element k0 */
    DCMT_STAGE_CLOSE5   = 4,  /* :84-85   5x5 close (LC: the label-masked stage, img_completion_lc.cpp:78-102) */
    DCMT_STAGE_FILL7    = 5,  /* :88-100  7x7 small-hole fill */
    DCMT_STAGE_EXTEND   = 6,  /* :103-129 column extension */
    DCMT_STAGE_FILL31   = 7,  /* :131-144 31x31 large-hole fill */
    DCMT_STAGE_FILLLOOP = 8,  /* :146-166 repeat until no holes */
    DCMT_STAGE_MEDIAN5  = 9,  /* :170     5x5 median */
    DCMT_STAGE_BLUR     = 10, /* :172-189 blur + masked select */
    DCMT_STAGE_FINAL    = 11  /* :191-202 invert back to metres */
} dcmt_stage;

/* All literals of the reference in one POD (SURVEY.md section 5 "Config / flags"). */
typedef struct {
    float   max_depth;       /* 100.0f                    img_completion.cpp:23 */
    float   valid_thresh;    /* 0.1f: valid <=> x >= valid_thresh, hole <=> x < valid_thresh.
                                The reference writes `depth > 0.1` / `depth < 0.1` against the DOUBLE
                                literal; for f32 inputs that is exactly x >= 0.1f / x < 0.1f. */
    uint8_t k0[25];          /* first structuring element, row-major 5x5, anchor centre, non-zero = tap.
                                Default: what the reference COMPILES to (2 taps), see dcmt_k0_as_compiled. */
    uint8_t _pad[3];
    int32_t blur;            /* dcmt_blur */
    int32_t max_fill_iters;  /* cap on the hole-closure loop (reference: unbounded); default 64 */
    int32_t spec_fill_iters; /* device-pointer entry points only: how many loop applications are enqueued
                                speculatively without a host round trip (each is skipped on the device for
                                frames that have no holes left).  Default 1.  If a frame still has holes
                                after them, dcmt_last_fill_iters() reports DCMT_E_NOT_CONVERGED for it. */
    int32_t stop_after;      /* dcmt_stage; DCMT_STAGE_FINAL for the whole chain */
    int32_t verbose;         /* host entry points: 1 = print what img_completion prints to stdout (dimensions :29, "max range is" :50, one
                                hole count per loop iteration :161); 2 = the hole counts only (all that interpolate_with_superpixels
                                prints, img_completion_lc.cpp:173); 0 = nothing */
    int32_t flags;           /* DCMT_FLAG_* */
    float   norm_lo;         /* DCMT_FLAG_NORMALIZE: the two range arguments of cv::normalize (alpha, beta); */
    float   norm_hi;         /* the stereo-lidar callers pass (0, 100) and (0, 80).  Defaults 0, 100. */
} dcmt_params;

/* Use the general staged kernels even where the fused fast path applies (A/B tests, debugging).
 * Both paths produce identical bits. */
#define DCMT_FLAG_FORCE_STAGED 1
/* Use the fused streaming kernels even for a batch too small to fill the GPU with them (by default
 * batches of fewer than 3 frames take the staged tile kernels, whose many small workgroups have the
 * lower latency for a single frame).  Both paths produce identical bits. */
#define DCMT_FLAG_FORCE_FUSED 2

/* The caller-side pre-step of the stereo-lidar executables fused in front of the cascade: every frame is first
 * min-max normalised, exactly as `cv::normalize(src, dst, norm_lo, norm_hi, cv::NORM_MINMAX)` does for a
 * CV_32F destination (DC_stereo_lidar/main_sl.cpp:370 before img_completion, :523 before
 * interpolate_with_superpixels): one extra read-only pass finds each frame's extrema, the first kernel of
 * the chain applies dst = src * a + b (f32, unfused) while it loads.  f32 entry points only
 * (dcmt_complete_u16_dev returns DCMT_E_UNSUPPORTED: the reference never combines the two ingests).
 * A frame whose values are all equal normalises to min(norm_lo, norm_hi) everywhere, as in OpenCV. */
#define DCMT_FLAG_NORMALIZE 4

/* ---- lifetime --------------------------------------------------------------------- */

/* Number of usable devices (0 if none); never fails. */
int dcmt_device_count(void);

/* Creates a context on `device` able to process up to max_batch frames of up to
 * max_rows x max_cols per call.  Allocates the device scratch every path needs up front (8 B per
 * pixel per frame of max_batch); two buffers only one path uses are allocated by the first call that takes it and kept (the
 * 16-bit plane of large on-grid batches, 2 B per pixel; the column statistics of the small-batch tile kernels), so no call
 * after the first of its kind allocates.  A frame may hold at most 2^29 - 16 pixels (it is addressed with 32-bit byte offsets
 * and one offset just below 2^31 is kept free as "nowhere"); max_batch at most 65535. */
int dcmt_create(int device, int max_rows, int max_cols, int max_batch, dcmt_ctx **out);
void dcmt_destroy(dcmt_ctx *ctx);

/* ---- parameters ------------------------------------------------------------------- */

void dcmt_default_params(dcmt_params *p);
/* The element the reference's `cv::Mat(5,5,CV_8UC1,d)` over `int d[5][5]` really is
 * (img_completion.cpp:71-77): taps at (row 1,col 3) and (row 4,col 4) only. */
void dcmt_k0_as_compiled(uint8_t k0[25]);
/* The 13-tap radius-2 diamond the reference's comment intends. */
void dcmt_k0_diamond(uint8_t k0[25]);

/* ---- img_completion --------------------------------------------------------------- */

/* HOST pointers, synchronous: copies in, runs the cascade, copies out.  Backs the cv::Mat
 * shim (replaces the body of img_completion, img_completion.cpp:17-204).
 * src/dst: f32, `batch` frames of rows x cols; row and frame strides in BYTES (cv::Mat::step
 * for a single Mat; frame strides are ignored when batch == 1).  src is never written.
 * The hole-closure loop runs exactly as many times as the reference would (up to
 * max_fill_iters); returns DCMT_E_NOT_CONVERGED if the cap was hit (dst is still written). */
int dcmt_complete_f32(dcmt_ctx *ctx,
                      const float *src, size_t src_row_stride, size_t src_frame_stride,
                      float *dst, size_t dst_row_stride, size_t dst_frame_stride,
                      int rows, int cols, int batch, const dcmt_params *params);

/* DEVICE pointers, stream-ordered, asynchronous: the measured path.  d_src/d_dst are
 * contiguous [batch][rows][cols] f32 in device memory of ctx's GPU; `stream` is a
 * hipStream_t (NULL = the default stream).  Never synchronises; never allocates (the labeled variant
 * grows its per-label bounding-box table the first time it sees a larger batch x n_labels). */
int dcmt_complete_f32_dev(dcmt_ctx *ctx, const float *d_src, float *d_dst,
                          int rows, int cols, int batch, const dcmt_params *params, void *stream);

/* The reference's ingest fused into the first kernel: d_src holds the KITTI uint16 depth PNG payload
 * ([batch][rows][cols], device memory), metres = value * scale -- what src/DC_lidar_only/main.cpp:75-82
 * does with imread(IMREAD_ANYDEPTH) + convertTo(CV_32F, 1./256) before calling img_completion.  Reads
 * 2 instead of 4 bytes per pixel; otherwise identical to dcmt_complete_f32_dev on the converted frame. */
int dcmt_complete_u16_dev(dcmt_ctx *ctx, const uint16_t *d_src, float scale, float *d_dst,
                          int rows, int cols, int batch, const dcmt_params *params, void *stream);

/* ---- interpolate_with_superpixels -------------------------------------------------- */

/* As above with a label plane: int32 [rows][cols] ROW-MAJOR per frame (the reference keeps
 * Slic::clusters[col][row]; the shim transposes), labels outside [0,n_labels) (e.g. -1)
 * are not touched by the masked stage.  use_superpixel == 0 runs the unmasked first stage
 * (img_completion_lc.cpp:59-64).  The Gaussian is applied whatever params->blur says, as in
 * the reference (img_completion_lc.cpp:183 ignores blur_type). */
int dcmt_complete_labeled_f32(dcmt_ctx *ctx,
                              const float *src, size_t src_row_stride, size_t src_frame_stride,
                              const int32_t *labels, size_t lab_row_stride, size_t lab_frame_stride,
                              int n_labels,
                              float *dst, size_t dst_row_stride, size_t dst_frame_stride,
                              int rows, int cols, int batch, const dcmt_params *params,
                              int use_superpixel);
int dcmt_complete_labeled_f32_dev(dcmt_ctx *ctx, const float *d_src, const int32_t *d_labels,
                                  int n_labels, float *d_dst, int rows, int cols, int batch,
                                  const dcmt_params *params, int use_superpixel, void *stream);

/* ---- producer of the path's input: LiDAR points -> sparse depth image (N2) ----------- */

/* What the stereo-lidar executables do between reading the velodyne .bin and calling the path
 * (DC_stereo_lidar/main_sl.cpp:478-520 in withSuperPixels, the same loop at :320-366 in vedi_pc):
 *     t = T * (x, y, z, 1)                 keep the point if t.z > 0                    (:480-490)
 *     p = P * (t.x, t.y, t.z, 1);  uf = p.x / p.z,  vf = p.y / p.z                      (:499-503)
 *     if 0 <= uf < cols and 0 <= vf < rows:  image(int(vf), int(uf)) = p.z              (:506-518)
 * in file order, so a later point overwrites an earlier one that fell into the same pixel.
 * d_points: device, [n_points][4] f32 = x, y, z, reflectance -- the KITTI .bin layout the reference reads (:468-472),
 * 16-byte aligned (every device allocation is; the records are read whole): DCMT_E_INVALID otherwise;
 * frame f owns points [d_offsets[f], d_offsets[f+1]) (device array of batch+1 ints, d_offsets[batch] == n_points).
 * T: 4x4, P: 3x4, both ROW-major host arrays (Eigen's default storage is column-major: pass the transposes' data()).
 * d_sparse: [batch][rows][cols] f32, written completely (0 = no point).  All arithmetic is f32, one rounding per
 * operation, sums left to right -- the order of the reference's hand-written transform; Eigen's evaluation order for
 * `P * p.homogeneous()` is not pinned, so against a real build a point whose uf or vf sits
 * within an ulp of an integer may land in the neighbouring pixel.  Stream-ordered, never synchronises; uses ctx
 * scratch, so do not overlap it with another call on the same ctx. */
int dcmt_project_points_dev(dcmt_ctx *ctx, const float *d_points, const int32_t *d_offsets, int n_points,
                            int batch, const float T[16], const float P[12], float *d_sparse,
                            int rows, int cols, void *stream);

/* ---- producer of the label plane: SLIC superpixels (N3) -------------------------------- */

/* Slic::generate_superpixels (DC_lidar_camera/slic.cpp:101-182; called at main_lc.cpp:200 with 1200 superpixels and
 * DC_stereo_lidar/main_sl.cpp:450 with 100) on the device: the grid of centres moved to their 3x3 gradient minimum
 * (init_data :19-57), then NR_ITERATIONS = 10 rounds of "every pixel of a centre's [c - step, c + step) window takes the
 * centre with the smallest distance (compute_dist :59-68, f64), the lowest index on ties" and "every centre becomes the
 * mean of its pixels".  d_lab: [batch][rows][cols][3] uint8 -- the image the reference passes (its cv::cvtColor(BGR2Lab)
 * output); step, nc: the reference's int arguments (step = (int)sqrt(w*h / n_superpixels), nc = 50 / 40).
 * d_labels: [batch][rows][cols] int32, row-major (the reference's clusters[col][row]), -1 = never reached: exactly what
 * dcmt_complete_labeled_f32_dev takes, with n_labels = dcmt_slic_num_centers(rows, cols, step).
 * d_centers (may be NULL): [batch][n][5] f64 = L, a, b, x, y after the last iteration.
 * The first call allocates SLIC scratch (per-centre and per-cell tables, a few hundred KiB per frame of max_batch);
 * later calls with the same or a larger step never allocate.
 * Requires step >= 6 (below that the reference's gradient probe reads outside the image) and nc >= 1. */
int dcmt_slic_num_centers(int rows, int cols, int step);
int dcmt_slic_labels_dev(dcmt_ctx *ctx, const uint8_t *d_lab, int rows, int cols, int batch, int step, int nc,
                         int32_t *d_labels, double *d_centers, void *stream);

/* ---- consumer of the path's output: stereo photometric refinement (N4) ----------------- */

/* What DC_stereo_lidar/main_sl.cpp does with the dense depth (:1165-1246): depth -> disparity (get_initial_disparity
 * :846-861), four damped Gauss-Newton sweeps that move every pixel's disparity along its epipolar line so that the right
 * image matches the left one (optimize_IG :804-843 with calculateObservationDerivatives :749-801 on images whose
 * derivatives calculateMeasuementDerivatives :715-747 made), disparity -> depth clamped to max_depth
 * (retrieve_optimized_depth :863-885).  Every pixel only ever touches its own disparity, so the sweeps are independent
 * per pixel.  d_left / d_right: [batch][rows][cols] uint8 grey images (the reference's cv::cvtColor(BGR2GRAY) outputs,
 * :1167-1171); d_depth: the path's output; d_refined: [batch][rows][cols] f32, 0 where the disparity ends up <= 0.
 * iterations < 0 selects the reference's 4; iterations == 0 is the pure depth -> disparity -> depth round trip
 * (its `depth_pre_optim`, :1225-1226). */
typedef struct {
    float   baseline;      /* 0.54f          :847 */
    float   focal;         /* 9.597910e+02f  :848 */
    float   damp;          /* 500            :808 */
    float   max_depth;     /* 100            :876 */
    int32_t iterations;    /* 4              :805 */
} dcmt_stereo_params;
void dcmt_default_stereo_params(dcmt_stereo_params *p);
int dcmt_stereo_refine_dev(dcmt_ctx *ctx, const float *d_depth, const uint8_t *d_left, const uint8_t *d_right,
                           float *d_refined, int rows, int cols, int batch, const dcmt_stereo_params *params,
                           void *stream);

/* ---- the same three on HOST memory (one frame, synchronous): what the cv::Mat shim calls ---------------- */

/* Each copies its inputs to the device, runs the device entry point above and copies the result back; device buffers
 * are allocated for the call and freed again (these calls are bound by the PCIe copies, not by that).
 * Row strides in BYTES (cv::Mat::step); labels / centres / points are contiguous. */
int dcmt_project_points(dcmt_ctx *ctx, const float *points, int n_points, const float T[16], const float P[12],
                        float *sparse, size_t sparse_row_stride, int rows, int cols);
int dcmt_slic_labels(dcmt_ctx *ctx, const uint8_t *lab, size_t lab_row_stride, int rows, int cols, int step, int nc,
                     int32_t *labels /* [rows][cols] */, double *centers /* [n][5] or NULL */);
int dcmt_stereo_refine(dcmt_ctx *ctx, const float *depth, size_t depth_row_stride,
                       const uint8_t *left, size_t left_row_stride, const uint8_t *right, size_t right_row_stride,
                       float *refined, size_t refined_row_stride, int rows, int cols,
                       const dcmt_stereo_params *params);

/* ---- probes ------------------------------------------------------------------------ */

/* Per frame of the last call on ctx: the number of iterations the reference's while-loop
 * (img_completion.cpp:146-166) ran (>= 1), i.e. the length of the hole-count sequence it
 * prints.  Synchronises with the last call's stream.  out[i] = -1 for a frame that still
 * had holes when the launches ran out (then the return value is DCMT_E_NOT_CONVERGED). */
int dcmt_last_fill_iters(dcmt_ctx *ctx, int *out, int n);
/* Per frame of the last call: holes seen by the first 31x31 fill (img_completion.cpp:131-144). */
int dcmt_last_holes_after_extend(dcmt_ctx *ctx, int *out, int n);

const char *dcmt_strerror(int status);
int dcmt_last_hip_error(const dcmt_ctx *ctx);
/* The kernels the last cascade call on ctx dispatched, e.g. "k_pre_p<Q16OUT> + k_fp_q" (the dispatch depends on batch size,
 * frame shape, alignment and -- for the 16-bit form -- on what earlier calls found in their frames).  Owned by ctx. */
const char *dcmt_last_path(const dcmt_ctx *ctx);

/* Measurement aid (bench.py's live per-kernel split).  With timing on, the streaming path of every following *_dev call
 * records HIP events on the caller's stream around its kernel groups (a few microseconds per call; off by default).
 * dcmt_last_kernel_times synchronises with the last call's stream and returns milliseconds:
 *   ms[0] the kernels in front of H5 that the call added (label-masked stage, normalisation scan; 0 for img_completion),
 *   ms[1] k_pre_s (H2..H6, or H5..H6 behind the label stage),  ms[2] k_fp_s (H7..H11),
 *   ms[3] the launches behind it (hole-closure redo / loop / recompute: near zero unless a frame needed the loop).
 * DCMT_E_INVALID if timing was off for the last call or the call did not take the streaming path. */
#define DCMT_N_KERNEL_TIMES 4
int dcmt_set_kernel_timing(dcmt_ctx *ctx, int on);
int dcmt_last_kernel_times(dcmt_ctx *ctx, float ms[DCMT_N_KERNEL_TIMES]);
int dcmt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DCMT_H */
