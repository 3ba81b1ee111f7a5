/*
 * dcmt_oracle.h -- CPU restatement of the reference's `img_completion` cascade.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call anything declared here.  The shipped
 * path (depth_completion_mt_amd/, include/dcmt.h) never falls back to it.
 *
 * PARITY UNPINNED: the arithmetic of the reference lives in OpenCV (un-vendored,
 * version un-pinned; cv::dilate / morphologyEx / medianBlur / GaussianBlur), which is
 * not installed in the build image, and the reference ships no tests, fixtures or
 * golden vectors for this path.  This file restates the reference's stage order
 * (/root/reference/src/DC_lidar_only/img_completion.cpp:17-204 and
 * src/DC_lidar_camera/img_completion_lc.cpp:34-203) plus OpenCV's documented
 * semantics; it has never been diffed against an executing OpenCV.
 */
#ifndef DCMT_ORACLE_H
#define DCMT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Stage ids: "run the chain up to and including this stage" (for per-stage parity). */
enum {
    DCMT_O_STAGE_INVERT   = 2,   /* H2  img_completion.cpp:55-67   */
    DCMT_O_STAGE_DILATE_K = 3,   /* H3  :71-80                      */
    DCMT_O_STAGE_CLOSE5   = 4,   /* H4  :84-85                      */
    DCMT_O_STAGE_FILL7    = 5,   /* H5  :88-100                     */
    DCMT_O_STAGE_EXTEND   = 6,   /* H6  :103-129                    */
    DCMT_O_STAGE_FILL31   = 7,   /* H7  :131-144                    */
    DCMT_O_STAGE_FILLLOOP = 8,   /* H8  :146-166                    */
    DCMT_O_STAGE_MEDIAN5  = 9,   /* H9  :170                        */
    DCMT_O_STAGE_BLUR     = 10,  /* H10 :172-189                    */
    DCMT_O_STAGE_FINAL    = 11   /* H11 :191-202                    */
};

enum { DCMT_O_BLUR_NONE = 0, DCMT_O_BLUR_GAUSSIAN = 1 };

typedef struct {
    float   max_depth;        /* 100.0f, img_completion.cpp:23 */
    uint8_t k0[25];           /* first structuring element, row-major 5x5, anchor centre */
    int     blur;             /* DCMT_O_BLUR_* ("gaussian" is what every caller passes) */
    int     max_fill_iters;   /* cap on the H8 while-loop (reference: unbounded) */
    int     stop_after;       /* DCMT_O_STAGE_*; DCMT_O_STAGE_FINAL for the whole chain */
} dcmt_oracle_params;

/* Fills defaults: as-compiled 2-tap k0, gaussian, 64 iterations, whole chain. */
void dcmt_oracle_default_params(dcmt_oracle_params *p);
/* The two k0 presets: what the reference compiles to (int[5][5] read as bytes) and the
 * 13-tap diamond its comment intends. */
void dcmt_oracle_k0_as_compiled(uint8_t k0[25]);
void dcmt_oracle_k0_diamond(uint8_t k0[25]);

/* img_completion (LO/img_completion.cpp:17-204).  src/dst: contiguous rows*cols f32.
 * fill_iters (may be NULL) receives the number of H8 loop iterations the reference
 * would have run (>=1); holes_after_extend (may be NULL) the hole count seen by H7.
 * Returns 0, or -1 if max_fill_iters was hit with holes left. */
int dcmt_oracle_img_completion(const float *src, float *dst, int rows, int cols,
                               const dcmt_oracle_params *p,
                               int *fill_iters, int *holes_after_extend);

/* interpolate_with_superpixels (LC/img_completion_lc.cpp:34-203).  labels is
 * int32[rows][cols] row-major (the reference stores clusters[col][row]); labels outside
 * [0,n_labels) are left untouched by the masked stage.  use_superpixel==0 is the plain
 * chain with the Gaussian applied unconditionally. */
int dcmt_oracle_interpolate_with_superpixels(const float *src, const int32_t *labels,
                                             int n_labels, float *dst, int rows, int cols,
                                             const dcmt_oracle_params *p, int use_superpixel,
                                             int *fill_iters);
/* Same, but literally one whole-image pass per label as the reference does
 * (LC :78-102).  O(n_labels*rows*cols): small cases only; checks the ROI version. */
int dcmt_oracle_interpolate_with_superpixels_bruteforce(const float *src, const int32_t *labels,
                                             int n_labels, float *dst, int rows, int cols,
                                             const dcmt_oracle_params *p, int use_superpixel,
                                             int *fill_iters);

/* Batch helper for the CPU baseline: frames are independent; `threads` OpenMP threads
 * split the batch (threads<=1: plain loop). */
int dcmt_oracle_img_completion_batch(const float *src, float *dst, int rows, int cols,
                                     int batch, const dcmt_oracle_params *p, int threads);

/* Individual primitives (exposed so tests can cross-check separable vs brute force). */
void dcmt_oracle_dilate_mask5(const float *src, float *dst, int rows, int cols, const uint8_t k[25]);
void dcmt_oracle_dilate_rect(const float *src, float *dst, int rows, int cols, int ksize);
void dcmt_oracle_erode_rect(const float *src, float *dst, int rows, int cols, int ksize);
void dcmt_oracle_dilate_rect_bruteforce(const float *src, float *dst, int rows, int cols, int ksize);
void dcmt_oracle_erode_rect_bruteforce(const float *src, float *dst, int rows, int cols, int ksize);
void dcmt_oracle_median5(const float *src, float *dst, int rows, int cols);          /* row-vectorised networks */
void dcmt_oracle_median5_simple(const float *src, float *dst, int rows, int cols);   /* the definition */
/* the chain entry points use the definition instead of the networks (process-wide switch; tests) */
void dcmt_oracle_use_definitional_median(int on);
void dcmt_oracle_gaussian5(const float *src, float *dst, int rows, int cols);
void dcmt_oracle_extend_columns(float *x, int rows, int cols);

/* N1: cv::normalize(src, dst, a, b, NORM_MINMAX) for CV_32F, the pre-step of the stereo-lidar callers
 * (DC_stereo_lidar/main_sl.cpp:370, :523). */
void dcmt_oracle_normalize_minmax(const float *src, float *dst, int rows, int cols, float a, float b);

/* N2: LiDAR points -> sparse depth image (DC_stereo_lidar/main_sl.cpp:478-520): points [n][4] f32 (x,y,z,reflectance),
 * T 4x4 and P 3x4 row-major; dst is overwritten completely (0 = no point); later points overwrite earlier ones. */
void dcmt_oracle_project_points(const float *points, int n, const float T[16], const float P[12],
                                float *dst, int rows, int cols);

/* N3: Slic::generate_superpixels (DC_lidar_camera/slic.cpp:101-182 with init_data :19-57, compute_dist :59-68,
 * find_local_minimum :71-98) -- the producer of the label plane of interpolate_with_superpixels.
 * lab: the 8-bit 3-channel image the caller hands over ([rows][cols][3]; the reference passes cv::cvtColor(BGR2Lab)).
 * step, nc as the reference's int parameters (the callers' double step is truncated at the call, main_lc.cpp:200).
 * labels: int32 [rows][cols] row-major (the reference's clusters[col][row]); -1 = never assigned.
 * centers (may be NULL): [n][5] doubles = L, a, b, x, y after the last iteration.  Returns the number of centres
 * (= slic.centers.size(), the n_labels of the label-masked stage), or -1 if max_centers is too small / step < 6. */
int dcmt_oracle_slic(const uint8_t *lab, int rows, int cols, int step, int nc, int32_t *labels,
                     double *centers, int max_centers);

/* N4: the stereo refinement behind the path (DC_stereo_lidar/main_sl.cpp:715-885 as driven from :1165-1246):
 * depth -> disparity, `iterations` damped Gauss-Newton sweeps per pixel against the right grey image, disparity -> depth
 * clamped to max_depth.  left / right: uint8 [rows][cols]; dst f32 [rows][cols]. */
void dcmt_oracle_stereo_refine(const float *depth, const uint8_t *left, const uint8_t *right, float *dst,
                               int rows, int cols, float baseline, float focal, float damp, float max_depth,
                               int iterations);

/* Deterministic KITTI-like synthetic sparse frame. */
void dcmt_oracle_synth_frame(float *dst, int rows, int cols, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
