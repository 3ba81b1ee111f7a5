/*
 * dcmt_oracle.c -- CPU restatement of the reference's `img_completion` cascade.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see dcmt_oracle.h).  PARITY UNPINNED: OpenCV,
 * which holds the reference's arithmetic, is not available here and the reference has
 * no tests/fixtures for this path; OpenCV's documented semantics are restated below.
 *
 * Every function cites the reference lines it follows.  LO = src/DC_lidar_only,
 * LC = src/DC_lidar_camera (paths relative to /root/reference).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: the Gaussian and the synthetic generator must not be
 * FMA-contracted, so that the numpy restatement (oracle/np_restatement.py) and the HIP
 * kernels (explicit __fmul_rn/__fadd_rn) can agree with this file bit for bit.
 */
#include "dcmt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- thresholds -----------------------------------------------------------------
 * The reference compares a float against the *double* literal 0.1
 * (LO/img_completion.cpp:59,96,113,117,140,154,184,194): the float is promoted.
 * (double)0.1f > 0.1, so "valid" is x >= 0.1f and "hole" is x < 0.1f; written here
 * exactly as the reference writes it. */
static inline int gt_thresh(float x) { return (double)x > 0.1; }
static inline int lt_thresh(float x) { return (double)x < 0.1; }

/* OpenCV std::max/std::min on floats: max(a,b) = (a < b) ? b : a. */
static inline float fmax_cv(float a, float b) { return a < b ? b : a; }
static inline float fmin_cv(float a, float b) { return b < a ? b : a; }

void dcmt_oracle_k0_as_compiled(uint8_t k0[25])
{
    /* LO/img_completion.cpp:71-77: `int d[5][5]` handed to cv::Mat(5,5,CV_8UC1,d):
     * OpenCV reads the first 25 BYTES of the 100-byte int array (row step 5). */
    static const int d[25] = { 0,0,1,0,0, 0,1,1,1,0, 1,1,1,1,1, 0,1,1,1,0, 0,0,1,0,0 };
    memcpy(k0, d, 25);
}

void dcmt_oracle_k0_diamond(uint8_t k0[25])
{
    static const uint8_t d[25] = { 0,0,1,0,0, 0,1,1,1,0, 1,1,1,1,1, 0,1,1,1,0, 0,0,1,0,0 };
    memcpy(k0, d, 25);
}

void dcmt_oracle_default_params(dcmt_oracle_params *p)
{
    p->max_depth = 100.0f;                 /* LO/img_completion.cpp:23 */
    dcmt_oracle_k0_as_compiled(p->k0);
    p->blur = DCMT_O_BLUR_GAUSSIAN;        /* every caller passes "gaussian" (LO/main.cpp:93) */
    p->max_fill_iters = 64;
    p->stop_after = DCMT_O_STAGE_FINAL;
}

/* ---- H2 / H11: invert (LO/img_completion.cpp:55-67, 191-202) --------------------- */
static void invert_valid(float *x, long n, float max_depth)
{
    for (long i = 0; i < n; ++i)
        if (gt_thresh(x[i])) x[i] = max_depth - x[i];
}

/* ---- cv::dilate with an arbitrary 5x5 element (LO/img_completion.cpp:80) ---------
 * dst(r,c) = max over non-zero k(kr,kc) of src(r+kr-2, c+kc-2); the element is NOT
 * reflected; BORDER_CONSTANT with the default border value = -FLT_MAX for dilation. */
void dcmt_oracle_dilate_mask5(const float *src, float *dst, int rows, int cols, const uint8_t k[25])
{
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float m = -FLT_MAX;
            for (int kr = 0; kr < 5; ++kr)
                for (int kc = 0; kc < 5; ++kc) {
                    if (!k[kr * 5 + kc]) continue;
                    int rr = r + kr - 2, cc = c + kc - 2;
                    float v = (rr < 0 || rr >= rows || cc < 0 || cc >= cols)
                                  ? -FLT_MAX : src[(long)rr * cols + cc];
                    m = fmax_cv(m, v);
                }
            dst[(long)r * cols + c] = m;
        }
}

/* ---- rectangular dilate / erode ---------------------------------------------------
 * cv::dilate / cv::erode with Mat::ones(k,k): max / min over the k x k window, anchor
 * centre, out-of-image taps = -FLT_MAX / +FLT_MAX.  Brute-force form first (obviously
 * the definition), then the separable form used by the chain (max/min are exact,
 * associative and commutative, so both give identical bits; tests check it). */
static void morph_rect_bruteforce(const float *src, float *dst, int rows, int cols, int ksize, int is_dilate)
{
    const int h = ksize / 2;
    const float border = is_dilate ? -FLT_MAX : FLT_MAX;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float m = border;
            for (int dr = -h; dr <= h; ++dr)
                for (int dc = -h; dc <= h; ++dc) {
                    int rr = r + dr, cc = c + dc;
                    float v = (rr < 0 || rr >= rows || cc < 0 || cc >= cols)
                                  ? border : src[(long)rr * cols + cc];
                    m = is_dilate ? fmax_cv(m, v) : fmin_cv(m, v);
                }
            dst[(long)r * cols + c] = m;
        }
}

void dcmt_oracle_dilate_rect_bruteforce(const float *s, float *d, int rows, int cols, int k)
{ morph_rect_bruteforce(s, d, rows, cols, k, 1); }
void dcmt_oracle_erode_rect_bruteforce(const float *s, float *d, int rows, int cols, int k)
{ morph_rect_bruteforce(s, d, rows, cols, k, 0); }

static void morph_rect_separable(const float *src, float *dst, int rows, int cols, int ksize, int is_dilate)
{
    const int h = ksize / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)rows * cols);
    /* row pass: in-image taps only (an out-of-image tap is the identity of max/min
     * unless the whole window is outside, which cannot happen: the centre is inside) */
    for (int r = 0; r < rows; ++r) {
        const float *s = src + (long)r * cols;
        float *t = tmp + (long)r * cols;
        for (int c = 0; c < cols; ++c) {
            int lo = c - h < 0 ? 0 : c - h, hi = c + h >= cols ? cols - 1 : c + h;
            float m = s[lo];
            if (is_dilate) { for (int cc = lo + 1; cc <= hi; ++cc) m = fmax_cv(m, s[cc]); }
            else           { for (int cc = lo + 1; cc <= hi; ++cc) m = fmin_cv(m, s[cc]); }
            t[c] = m;
        }
    }
    /* column pass */
    for (int r = 0; r < rows; ++r) {
        int lo = r - h < 0 ? 0 : r - h, hi = r + h >= rows ? rows - 1 : r + h;
        float *d = dst + (long)r * cols;
        memcpy(d, tmp + (long)lo * cols, sizeof(float) * cols);
        for (int rr = lo + 1; rr <= hi; ++rr) {
            const float *t = tmp + (long)rr * cols;
            if (is_dilate) { for (int c = 0; c < cols; ++c) d[c] = fmax_cv(d[c], t[c]); }
            else           { for (int c = 0; c < cols; ++c) d[c] = fmin_cv(d[c], t[c]); }
        }
    }
    free(tmp);
}

void dcmt_oracle_dilate_rect(const float *s, float *d, int rows, int cols, int k)
{ morph_rect_separable(s, d, rows, cols, k, 1); }
void dcmt_oracle_erode_rect(const float *s, float *d, int rows, int cols, int k)
{ morph_rect_separable(s, d, rows, cols, k, 0); }

/* ---- fill holes from a dilated copy (LO/img_completion.cpp:88-100, 131-144, 147-159)
 * s = dilate(clone(x), ones(k,k)); x = (x < 0.1) ? s : x.  Returns #holes seen. */
static int fill_from_dilate(float *x, float *scratch, int rows, int cols, int ksize)
{
    int holes = 0;
    dcmt_oracle_dilate_rect(x, scratch, rows, cols, ksize);
    for (long i = 0, n = (long)rows * cols; i < n; ++i)
        if (lt_thresh(x[i])) { x[i] = scratch[i]; ++holes; }
    return holes;
}

/* ---- H6: column extension (LO/img_completion.cpp:103-129) -------------------------- */
void dcmt_oracle_extend_columns(float *x, int rows, int cols)
{
    for (int j = 0; j < cols; ++j) {
        int max_index = 0;         float max_val = -1;     /* :108-109 */
        float min_val = 100;       int min_index = rows - 1; /* :110-111 */
        for (int i = 0; i < rows; ++i) {
            if (gt_thresh(x[(long)i * cols + j])) { max_index = i; max_val = x[(long)i * cols + j]; }
            if (gt_thresh(x[(long)(rows - 1 - i) * cols + j])) {
                min_index = rows - 1 - i; min_val = x[(long)(rows - 1 - i) * cols + j];
            }
        }
        for (int i = max_index; i < rows; ++i) x[(long)i * cols + j] = max_val;  /* :122-124 */
        for (int i = min_index; i >= 0; --i)   x[(long)i * cols + j] = min_val;  /* :125-127 */
    }
}

/* ---- H9: cv::medianBlur(x, x, 5) on CV_32F (LO/img_completion.cpp:170) -------------
 * exact median of the 5x5 window, BORDER_REPLICATE, behaves out of place.
 * dcmt_oracle_median5_simple is the definition (gather 25, select the 13th); dcmt_oracle_median5
 * is what the chain and the CPU baseline use: the same exact median through min/max networks
 * applied to whole rows (gcc vectorises the loops), so that the baseline is not dominated by a
 * scalar selection sort.  The two are compared bit for bit in tests/test_oracle.py. */
void dcmt_oracle_median5_simple(const float *src, float *dst, int rows, int cols)
{
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float w[25];
            int n = 0;
            for (int dr = -2; dr <= 2; ++dr) {
                int rr = r + dr; rr = rr < 0 ? 0 : (rr >= rows ? rows - 1 : rr);
                for (int dc = -2; dc <= 2; ++dc) {
                    int cc = c + dc; cc = cc < 0 ? 0 : (cc >= cols ? cols - 1 : cc);
                    w[n++] = src[(long)rr * cols + cc];
                }
            }
            /* selection sort of the 13 smallest: w[12] is the median */
            for (int i = 0; i <= 12; ++i) {
                int mi = i;
                for (int j = i + 1; j < 25; ++j) if (w[j] < w[mi]) mi = j;
                float t = w[i]; w[i] = w[mi]; w[mi] = t;
            }
            dst[(long)r * cols + c] = w[12];
        }
}

/* compare-exchange of two rows, element by element */
static void cx_rows(float *lo, float *hi, int n)
{
    for (int i = 0; i < n; ++i) {
        const float a = lo[i], b = hi[i];
        lo[i] = fmin_cv(a, b);
        hi[i] = fmax_cv(a, b);
    }
}

#include "median_nets.h"

void dcmt_oracle_median5(const float *src, float *dst, int rows, int cols)
{
    /* S[k][r] = k-th smallest of the 5 horizontal (column-replicated) neighbours of every pixel of row r */
    const size_t plane = (size_t)rows * cols;
    float *S = (float *)malloc(sizeof(float) * plane * 5);
    float *w = (float *)malloc(sizeof(float) * (size_t)cols * 20);   /* 20 work rows */
    float *v[20];
    for (int k = 0; k < 20; ++k) v[k] = w + (size_t)k * cols;
#define ROWCX(a, b) cx_rows(v[a], v[b], cols);
    for (int r = 0; r < rows; ++r) {
        const float *s = src + (size_t)r * cols;
        for (int k = 0; k < 5; ++k)
            for (int c = 0; c < cols; ++c) {
                int cc = c + k - 2; cc = cc < 0 ? 0 : (cc >= cols ? cols - 1 : cc);
                v[k][c] = s[cc];
            }
        DCMT_SORT5_NET(ROWCX, ROWCX, ROWCX)
        for (int k = 0; k < 5; ++k) memcpy(S + plane * k + (size_t)r * cols, v[k], sizeof(float) * cols);
    }
    /* window of output row j = rows j-2..j+2 (row-replicated): merge two pairs of sorted rows, take the six
     * middle order statistics of those 20 values, then the 6th smallest of them and the fifth row
     * (tools/gen_median_shared.py explains and verifies the scheme) */
    static const int m55[10] = DCMT_MERGE55_OUT;
    static const int mid[6] = DCMT_MID20_OUT;
    float *P = (float *)malloc(sizeof(float) * (size_t)cols * 20);
    for (int j = 0; j < rows; ++j) {
        int rr[5];
        for (int k = 0; k < 5; ++k) { int r = j + k - 2; rr[k] = r < 0 ? 0 : (r >= rows ? rows - 1 : r); }
        for (int half = 0; half < 2; ++half) {                 /* pairs (rr[0],rr[1]) and (rr[2],rr[3]) */
            for (int k = 0; k < 5; ++k) {
                memcpy(v[k], S + plane * k + (size_t)rr[2 * half] * cols, sizeof(float) * cols);
                memcpy(v[5 + k], S + plane * k + (size_t)rr[2 * half + 1] * cols, sizeof(float) * cols);
            }
            DCMT_MERGE55_NET(ROWCX, ROWCX, ROWCX)
            for (int k = 0; k < 10; ++k) memcpy(P + (size_t)(10 * half + k) * cols, v[m55[k]], sizeof(float) * cols);
        }
        memcpy(w, P, sizeof(float) * (size_t)cols * 20);
        DCMT_MID20_NET(ROWCX, ROWCX, ROWCX)
        const float *a[5], *C[6];
        for (int k = 0; k < 5; ++k) a[k] = S + plane * k + (size_t)rr[4] * cols;
        for (int k = 0; k < 6; ++k) C[k] = v[mid[k]];
        float *d = dst + (size_t)j * cols;
        for (int c = 0; c < cols; ++c) {
            float m = C[5][c];
            m = fmin_cv(m, fmax_cv(a[0][c], C[4][c]));
            m = fmin_cv(m, fmax_cv(a[1][c], C[3][c]));
            m = fmin_cv(m, fmax_cv(a[2][c], C[2][c]));
            m = fmin_cv(m, fmax_cv(a[3][c], C[1][c]));
            m = fmin_cv(m, fmax_cv(a[4][c], C[0][c]));
            d[c] = m;
        }
    }
#undef ROWCX
    free(P); free(w); free(S);
}

/* ---- H10: cv::GaussianBlur(s, s, Size(5,5), 0) (LO/img_completion.cpp:179) ---------
 * sigma<=0 and ksize<=7 selects OpenCV's fixed table [1,4,6,4,1]/16; separable, f32
 * accumulation, BORDER_REFLECT_101.  Canonical scalar order (OpenCV's symmetric row /
 * column filters): k0*c + k1*(l1+r1) + k2*(l2+r2), left to right, no FMA. */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

void dcmt_oracle_gaussian5(const float *src, float *dst, int rows, int cols)
{
    const float k0 = 0.375f, k1 = 0.25f, k2 = 0.0625f;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)rows * cols);
    for (int r = 0; r < rows; ++r) {
        const float *s = src + (long)r * cols;
        for (int c = 0; c < cols; ++c) {
            float l1 = s[reflect101(c - 1, cols)], r1 = s[reflect101(c + 1, cols)];
            float l2 = s[reflect101(c - 2, cols)], r2 = s[reflect101(c + 2, cols)];
            float acc = s[c] * k0;
            acc = acc + (l1 + r1) * k1;
            acc = acc + (l2 + r2) * k2;
            tmp[(long)r * cols + c] = acc;
        }
    }
    for (int r = 0; r < rows; ++r) {
        const float *u1 = tmp + (long)reflect101(r - 1, rows) * cols;
        const float *d1 = tmp + (long)reflect101(r + 1, rows) * cols;
        const float *u2 = tmp + (long)reflect101(r - 2, rows) * cols;
        const float *d2 = tmp + (long)reflect101(r + 2, rows) * cols;
        const float *m = tmp + (long)r * cols;
        for (int c = 0; c < cols; ++c) {
            float acc = m[c] * k0;
            acc = acc + (u1[c] + d1[c]) * k1;
            acc = acc + (u2[c] + d2[c]) * k2;
            dst[(long)r * cols + c] = acc;
        }
    }
    free(tmp);
}

/* Which median the chain runs: 0 = the row-vectorised networks (default; the CPU baseline), 1 = the definition
 * (dcmt_oracle_median5_simple: gather 25, select the 13th), which shares nothing with the HIP kernels' networks.
 * Process-wide; set it before calling the chain (tests only). */
static int g_median_definitional = 0;
void dcmt_oracle_use_definitional_median(int on) { g_median_definitional = on != 0; }

/* ---- tail shared by both entry points: H5 .. H11 ---------------------------------- */
static int chain_tail(float *x, float *scratch, int rows, int cols, const dcmt_oracle_params *p,
                      int blur, int *fill_iters, int *holes_after_extend)
{
    const long n = (long)rows * cols;
    int rc = 0, iters = 0;
    if (fill_iters) *fill_iters = 0;
    if (holes_after_extend) *holes_after_extend = 0;

    /* H5 small fill, 7x7 (LO :88-100) */
    fill_from_dilate(x, scratch, rows, cols, 7);
    if (p->stop_after <= DCMT_O_STAGE_FILL7) return 0;
    /* H6 column extension (LO :103-129), `densify` is the constant true */
    dcmt_oracle_extend_columns(x, rows, cols);
    if (p->stop_after <= DCMT_O_STAGE_EXTEND) return 0;
    /* H7 large fill, 31x31 (LO :131-144) */
    {
        int h = fill_from_dilate(x, scratch, rows, cols, 31);
        if (holes_after_extend) *holes_after_extend = h;
    }
    if (p->stop_after <= DCMT_O_STAGE_FILL31) return 0;
    /* H8 while(true){ dilate31; count; fill; if(count==0) break; } (LO :146-166) */
    for (;;) {
        int holes = fill_from_dilate(x, scratch, rows, cols, 31);
        ++iters;
        if (holes == 0) break;
        if (iters >= p->max_fill_iters) { rc = -1; break; }
    }
    if (fill_iters) *fill_iters = iters;
    if (p->stop_after <= DCMT_O_STAGE_FILLLOOP) return rc;
    /* H9 median 5x5 (LO :170) */
    if (g_median_definitional) dcmt_oracle_median5_simple(x, scratch, rows, cols);
    else dcmt_oracle_median5(x, scratch, rows, cols);
    memcpy(x, scratch, sizeof(float) * n);
    if (p->stop_after <= DCMT_O_STAGE_MEDIAN5) return rc;
    /* H10 Gaussian + masked select (LO :176-189); "bilateral" throws in OpenCV
     * (in-place bilateralFilter) and no caller uses it: not restated. */
    if (blur == DCMT_O_BLUR_GAUSSIAN) {
        dcmt_oracle_gaussian5(x, scratch, rows, cols);
        for (long i = 0; i < n; ++i) if (gt_thresh(x[i])) x[i] = scratch[i];
    }
    if (p->stop_after <= DCMT_O_STAGE_BLUR) return rc;
    /* H11 invert back (LO :191-202) */
    invert_valid(x, n, p->max_depth);
    return rc;
}

int dcmt_oracle_img_completion(const float *src, float *dst, int rows, int cols,
                               const dcmt_oracle_params *p, int *fill_iters, int *holes_after_extend)
{
    const long n = (long)rows * cols;
    float *scratch = (float *)malloc(sizeof(float) * (size_t)n);
    int rc = 0;
    if (fill_iters) *fill_iters = 0;
    if (holes_after_extend) *holes_after_extend = 0;
    memcpy(dst, src, sizeof(float) * n);                       /* H0 clone (LO :27) */
    /* H1 max scan (LO :41-50) only feeds a print: not restated. */
    invert_valid(dst, n, p->max_depth);                        /* H2 */
    if (p->stop_after <= DCMT_O_STAGE_INVERT) goto done;
    dcmt_oracle_dilate_mask5(dst, scratch, rows, cols, p->k0); /* H3 (in-place call == out of place) */
    memcpy(dst, scratch, sizeof(float) * n);
    if (p->stop_after <= DCMT_O_STAGE_DILATE_K) goto done;
    dcmt_oracle_dilate_rect(dst, scratch, rows, cols, 5);      /* H4 MORPH_CLOSE = dilate then erode */
    dcmt_oracle_erode_rect(scratch, dst, rows, cols, 5);
    if (p->stop_after <= DCMT_O_STAGE_CLOSE5) goto done;
    rc = chain_tail(dst, scratch, rows, cols, p, p->blur, fill_iters, holes_after_extend);
done:
    free(scratch);
    return rc;
}

/* ---- LC: label-masked first stage (LC/img_completion_lc.cpp:78-102) ---------------
 * for c in [0,n_labels): region = zeros; region[label==c] = x[label==c];
 * region = erode5(dilate5(dilateK(region))) with the image-border sentinels;
 * x[label==c] = region[label==c].  Labels are disjoint and every write-back reads only
 * the pre-loop x restricted to its own label... except that it does not: the loop body
 * reads `dense_r_img` (LC :95), which earlier iterations have already modified, but only
 * at pixels of OTHER labels, which the mask zeroes.  So the order is irrelevant. */
static void masked_stage_bruteforce(float *x, const int32_t *labels, int n_labels,
                                    int rows, int cols, const uint8_t k0[25])
{
    const long n = (long)rows * cols;
    float *a = (float *)malloc(sizeof(float) * (size_t)n);
    float *b = (float *)malloc(sizeof(float) * (size_t)n);
    for (int c = 0; c < n_labels; ++c) {
        for (long i = 0; i < n; ++i) a[i] = labels[i] == c ? x[i] : 0.0f;
        dcmt_oracle_dilate_mask5(a, b, rows, cols, k0);
        dcmt_oracle_dilate_rect_bruteforce(b, a, rows, cols, 5);
        dcmt_oracle_erode_rect_bruteforce(a, b, rows, cols, 5);
        for (long i = 0; i < n; ++i) if (labels[i] == c) x[i] = b[i];
    }
    free(a); free(b);
}

/* Same result, one bounding box per label.  A pixel's result depends on inputs within
 * Chebyshev distance 6 (2 for k0 + 2 + 2), so each label is processed on its bounding
 * box grown by 8 and clipped to the image; window taps are clipped to the IMAGE (not to
 * the box), and whatever lies in the box but is not of this label is 0, as in the
 * reference's zero-initialised `superpixelRegion` (LC :94-95). */
static void masked_stage_roi(float *x, const int32_t *labels, int n_labels,
                             int rows, int cols, const uint8_t k0[25])
{
    const long n = (long)rows * cols;
    int *r0 = (int *)malloc(sizeof(int) * 4 * (size_t)(n_labels > 0 ? n_labels : 1));
    int *r1 = r0 + n_labels, *c0 = r1 + n_labels, *c1 = c0 + n_labels;
    float *out = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(out, x, sizeof(float) * n);
    for (int l = 0; l < n_labels; ++l) { r0[l] = rows; r1[l] = -1; c0[l] = cols; c1[l] = -1; }
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            int l = labels[(long)r * cols + c];
            if (l < 0 || l >= n_labels) continue;
            if (r < r0[l]) r0[l] = r;
            if (r > r1[l]) r1[l] = r;
            if (c < c0[l]) c0[l] = c;
            if (c > c1[l]) c1[l] = c;
        }
    for (int l = 0; l < n_labels; ++l) {
        if (r1[l] < 0) continue;
        const int G = 8;
        int br0 = r0[l] - G < 0 ? 0 : r0[l] - G, br1 = r1[l] + G >= rows ? rows - 1 : r1[l] + G;
        int bc0 = c0[l] - G < 0 ? 0 : c0[l] - G, bc1 = c1[l] + G >= cols ? cols - 1 : c1[l] + G;
        int bh = br1 - br0 + 1, bw = bc1 - bc0 + 1;
        float *m  = (float *)malloc(sizeof(float) * (size_t)bh * bw * 3);
        float *d1 = m + (long)bh * bw, *d2 = d1 + (long)bh * bw;
        for (int r = 0; r < bh; ++r)
            for (int c = 0; c < bw; ++c) {
                long gi = (long)(br0 + r) * cols + bc0 + c;
                m[(long)r * bw + c] = labels[gi] == l ? x[gi] : 0.0f;
            }
        /* helper: value of plane `pl` at image coords, `outside` when off the IMAGE;
         * taps inside the image but outside the box are never needed by box-interior
         * pixels that matter (distance argument above); they read as `inside_far`. */
#define AT(pl, rr, cc, outside, inside_far)                                              \
        (((rr) < 0 || (rr) >= rows || (cc) < 0 || (cc) >= cols) ? (outside)              \
         : (((rr) < br0 || (rr) > br1 || (cc) < bc0 || (cc) > bc1) ? (inside_far)        \
            : (pl)[(long)((rr) - br0) * bw + ((cc) - bc0)]))
        for (int r = br0; r <= br1; ++r)
            for (int c = bc0; c <= bc1; ++c) {
                float mx = -FLT_MAX;
                for (int kr = 0; kr < 5; ++kr)
                    for (int kc = 0; kc < 5; ++kc)
                        if (k0[kr * 5 + kc]) {
                            float v = AT(m, r + kr - 2, c + kc - 2, -FLT_MAX, 0.0f);
                            mx = fmax_cv(mx, v);
                        }
                d1[(long)(r - br0) * bw + (c - bc0)] = mx;
            }
        for (int r = br0; r <= br1; ++r)
            for (int c = bc0; c <= bc1; ++c) {
                float mx = -FLT_MAX;
                for (int dr = -2; dr <= 2; ++dr)
                    for (int dc = -2; dc <= 2; ++dc) {
                        float v = AT(d1, r + dr, c + dc, -FLT_MAX, -FLT_MAX);
                        mx = fmax_cv(mx, v);
                    }
                d2[(long)(r - br0) * bw + (c - bc0)] = mx;
            }
        for (int r = r0[l]; r <= r1[l]; ++r)
            for (int c = c0[l]; c <= c1[l]; ++c) {
                if (labels[(long)r * cols + c] != l) continue;
                float mn = FLT_MAX;
                for (int dr = -2; dr <= 2; ++dr)
                    for (int dc = -2; dc <= 2; ++dc) {
                        float v = AT(d2, r + dr, c + dc, FLT_MAX, FLT_MAX);
                        mn = fmin_cv(mn, v);
                    }
                out[(long)r * cols + c] = mn;
            }
#undef AT
        free(m);
    }
    memcpy(x, out, sizeof(float) * n);
    free(out); free(r0);
}

static int lc_chain(const float *src, const int32_t *labels, int n_labels, float *dst,
                    int rows, int cols, const dcmt_oracle_params *p, int use_superpixel,
                    int *fill_iters, int bruteforce)
{
    const long n = (long)rows * cols;
    float *scratch = (float *)malloc(sizeof(float) * (size_t)n);
    int rc = 0;
    if (fill_iters) *fill_iters = 0;
    memcpy(dst, src, sizeof(float) * n);                         /* LC :43 */
    invert_valid(dst, n, p->max_depth);                          /* LC :45-52 */
    if (p->stop_after <= DCMT_O_STAGE_INVERT) goto done;
    if (use_superpixel == 0) {                                   /* LC :59-64 */
        dcmt_oracle_dilate_mask5(dst, scratch, rows, cols, p->k0);
        memcpy(dst, scratch, sizeof(float) * n);
        if (p->stop_after <= DCMT_O_STAGE_DILATE_K) goto done;
        dcmt_oracle_dilate_rect(dst, scratch, rows, cols, 5);
        dcmt_oracle_erode_rect(scratch, dst, rows, cols, 5);
    } else {                                                     /* LC :78-102 */
        if (bruteforce) masked_stage_bruteforce(dst, labels, n_labels, rows, cols, p->k0);
        else            masked_stage_roi(dst, labels, n_labels, rows, cols, p->k0);
    }
    if (p->stop_after <= DCMT_O_STAGE_CLOSE5) goto done;
    /* LC :105-202: the tail; the Gaussian is unconditional (blur_type is unused, LC :183) */
    rc = chain_tail(dst, scratch, rows, cols, p, DCMT_O_BLUR_GAUSSIAN, fill_iters, NULL);
done:
    free(scratch);
    return rc;
}

int dcmt_oracle_interpolate_with_superpixels(const float *src, const int32_t *labels, int n_labels,
                                             float *dst, int rows, int cols,
                                             const dcmt_oracle_params *p, int use_superpixel,
                                             int *fill_iters)
{ return lc_chain(src, labels, n_labels, dst, rows, cols, p, use_superpixel, fill_iters, 0); }

int dcmt_oracle_interpolate_with_superpixels_bruteforce(const float *src, const int32_t *labels,
                                             int n_labels, float *dst, int rows, int cols,
                                             const dcmt_oracle_params *p, int use_superpixel,
                                             int *fill_iters)
{ return lc_chain(src, labels, n_labels, dst, rows, cols, p, use_superpixel, fill_iters, 1); }

int dcmt_oracle_img_completion_batch(const float *src, float *dst, int rows, int cols,
                                     int batch, const dcmt_oracle_params *p, int threads)
{
    const long n = (long)rows * cols;
    int rc = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(min : rc)
    for (int f = 0; f < batch; ++f) {
        int r = dcmt_oracle_img_completion(src + f * n, dst + f * n, rows, cols, p, NULL, NULL);
        if (r < rc) rc = r;
    }
    return rc;
}

/* ---- N1: cv::normalize(src, dst, a, b, NORM_MINMAX), CV_32F -> CV_32F ------------------
 * The step the stereo-lidar executables run in front of the path
 * (/root/reference/src/DC_stereo_lidar/main_sl.cpp:370 with (0,100) before img_completion,
 * :523 with (0,80) before interpolate_with_superpixels).  OpenCV's published algorithm restated
 * (modules/core/src/norm.cpp, cv::normalize, NORM_MINMAX branch; convertTo with scale):
 *   minMaxIdx -> smin, smax (double);  dmin = min(a,b), dmax = max(a,b);
 *   scale = (dmax - dmin) * (smax - smin > DBL_EPSILON ? 1/(smax - smin) : 0);
 *   for a CV_32F result: scale = (float)scale; shift = (float)dmin - (float)(smin*scale);
 *   dst = src * (float)scale + (float)shift in f32 (scalar path: one rounding per operation; a
 *   SIMD/FMA build rounds once -- identical whenever shift == 0, i.e. whenever the frame has an
 *   empty (0) pixel and dmin == 0, which is every frame the reference feeds it).
 * PARITY UNPINNED like the rest of this file. */
void dcmt_oracle_normalize_minmax(const float *src, float *dst, int rows, int cols, float a, float b)
{
    const size_t n = (size_t)rows * cols;
    double smin = src[0], smax = src[0];
    for (size_t i = 1; i < n; ++i) {
        if (src[i] < smin) smin = src[i];
        if (src[i] > smax) smax = src[i];
    }
    const double dmin = a < b ? a : b, dmax = a < b ? b : a;
    double scale = (dmax - dmin) * (smax - smin > 2.220446049250313e-16 ? 1.0 / (smax - smin) : 0.0);
    scale = (float)scale;
    const double shift = (double)(float)dmin - (double)(float)(smin * scale);
    const float fa = (float)scale, fb = (float)shift;
    for (size_t i = 0; i < n; ++i) {
        const float m = src[i] * fa;       /* built with -ffp-contract=off: two roundings */
        dst[i] = m + fb;
    }
}

/* ---- N2: LiDAR points -> sparse depth image ------------------------------------------
 * /root/reference/src/DC_stereo_lidar/main_sl.cpp:478-520 (withSuperPixels; vedi_pc :320-366 is the same loop):
 * the rigid transform is written out by hand there (:483-485, f32, left to right), points with z <= 0 are dropped
 * (:487), `P * p.homogeneous()` is an Eigen 3x4 * 4 product (:499-500), then the two divisions (:502-503), the
 * float bounds test (:506-507), truncation to int (:511-512) and the store of the projected z (:518).  Restated with
 * every product and sum rounded to f32 and sums taken left to right (this file is built with -ffp-contract=off).
 * PARITY UNPINNED: Eigen is absent, its reduction order for the 4-term rows (and whether the build fuses
 * multiply-adds) is not pinned by the reference, which has no build system. */
static inline float dot4_rn(const float *m, float x, float y, float z)
{
    float a = m[0] * x;
    float b = m[1] * y;
    a = a + b;
    b = m[2] * z;
    a = a + b;
    return a + m[3];
}

void dcmt_oracle_project_points(const float *points, int n, const float T[16], const float P[12],
                                float *dst, int rows, int cols)
{
    for (size_t i = 0; i < (size_t)rows * cols; ++i) dst[i] = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float x = points[4 * (size_t)i], y = points[4 * (size_t)i + 1], z = points[4 * (size_t)i + 2];
        const float tx = dot4_rn(T, x, y, z), ty = dot4_rn(T + 4, x, y, z), tz = dot4_rn(T + 8, x, y, z);
        if (!(tz > 0.0f)) continue;
        const float px = dot4_rn(P, tx, ty, tz), py = dot4_rn(P + 4, tx, ty, tz), pz = dot4_rn(P + 8, tx, ty, tz);
        const float uf = px / pz, vf = py / pz;
        if (uf >= 0.0f && uf < (float)cols && vf >= 0.0f && vf < (float)rows)
            dst[(size_t)(int)vf * cols + (int)uf] = pz;
    }
}

/* ---- N3: SLIC superpixels ---------------------------------------------------------------
 * /root/reference/src/DC_lidar_camera/slic.cpp, restated line by line in the same loop orders:
 *   init_data :19-57          labels -1, centres on a grid (x outer, y inner: `i = step; i < cols - step/2; i += step`),
 *                             each moved to the lowest-gradient pixel of its 3x3 neighbourhood (find_local_minimum
 *                             :71-98, first minimum in x-outer / y-inner order, the comparison on
 *                             sqrt(pow(d1,2)) + sqrt(pow(d2,2)) = |d1| + |d2|);
 *   generate_superpixels :101-182, NR_ITERATIONS = 10 (slic.h:20):
 *     distances = FLT_MAX; for every centre in index order, every pixel of its [c - step, c + step) window
 *     (`int k = centers[j][3] - step; k < centers[j][3] + step`: truncating conversion, double comparison) takes the
 *     centre if compute_dist is strictly smaller -- ties therefore stay with the lower centre index;
 *     compute_dist :59-68 in double: dc, ds Euclidean, sqrt((dc/nc)^2 + (ds/ns)^2), ns = step;
 *     then every centre becomes the mean of its pixels (:150-172; sums of integers, exact in double).
 *   A pixel no window reaches keeps the label it had (the reference resets the distances, not the clusters).
 * pow(x, 2) is restated as x * x (what it evaluates to); everything else is +, -, /, sqrt on doubles, each correctly
 * rounded, no contraction (-ffp-contract=off).  A centre that loses all its pixels becomes NaN in the reference and
 * its window loop never runs again (`k < NaN` is false); here it is flagged dead to the same effect.
 * create_connectivity (:186-259) only fills a local array nothing reads, so it is not part of the path.
 * PARITY UNPINNED, as the rest of this file (never diffed against an executing build of the reference). */
static double slic_dist(const double *c, int x, int y, const uint8_t *px, int nc, int ns)
{
    const double d0 = c[0] - (double)px[0], d1 = c[1] - (double)px[1], d2 = c[2] - (double)px[2];
    const double dc = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    const double e0 = c[3] - (double)x, e1 = c[4] - (double)y;
    const double ds = sqrt(e0 * e0 + e1 * e1);
    const double a = dc / (double)nc, b = ds / (double)ns;
    return sqrt(a * a + b * b);
}

int dcmt_oracle_slic(const uint8_t *lab, int rows, int cols, int step, int nc, int32_t *labels,
                     double *centers_out, int max_centers)
{
    if (step < 6 || nc < 1) return -1;          /* the 3x3 gradient probe of init_data would read outside the image */
    const int ns = step;
    int n = 0;
    for (int i = step; i < cols - step / 2; i += step)
        for (int j = step; j < rows - step / 2; j += step) ++n;
    if (n > max_centers) return -1;
    double *C = (double *)malloc(sizeof(double) * 5 * (size_t)(n > 0 ? n : 1));
    double *dist = (double *)malloc(sizeof(double) * (size_t)rows * cols);
    long long *sum = (long long *)malloc(sizeof(long long) * 6 * (size_t)(n > 0 ? n : 1));
    char *dead = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    if (!C || !dist || !sum || !dead) { free(C); free(dist); free(sum); free(dead); return -1; }
    for (size_t p = 0; p < (size_t)rows * cols; ++p) labels[p] = -1;
#define LAB(y, x) (lab + 3 * ((size_t)(y) * cols + (x)))
    int c = 0;
    for (int i = step; i < cols - step / 2; i += step)
        for (int j = step; j < rows - step / 2; j += step) {
            double min_grad = FLT_MAX;
            int mx = i, my = j;
            for (int ii = i - 1; ii < i + 2; ++ii)
                for (int jj = j - 1; jj < j + 2; ++jj) {
                    const double i1 = LAB(jj + 1, ii)[0], i2 = LAB(jj, ii + 1)[0], i3 = LAB(jj, ii)[0];
                    if (fabs(i1 - i3) + fabs(i2 - i3) < min_grad) { min_grad = fabs(i1 - i3) + fabs(i2 - i3); mx = ii; my = jj; }
                }
            C[5 * c] = LAB(my, mx)[0]; C[5 * c + 1] = LAB(my, mx)[1]; C[5 * c + 2] = LAB(my, mx)[2];
            C[5 * c + 3] = mx; C[5 * c + 4] = my;
            ++c;
        }
    for (int it = 0; it < 10; ++it) {
        for (size_t p = 0; p < (size_t)rows * cols; ++p) dist[p] = FLT_MAX;
        for (int j = 0; j < n; ++j) {
            if (dead[j]) continue;
            const double cx = C[5 * j + 3], cy = C[5 * j + 4];
            for (int k = (int)(cx - step); k < cx + step; ++k)
                for (int l = (int)(cy - step); l < cy + step; ++l)
                    if (k >= 0 && k < cols && l >= 0 && l < rows) {
                        const double d = slic_dist(C + 5 * j, k, l, LAB(l, k), nc, ns);
                        if (d < dist[(size_t)l * cols + k]) { dist[(size_t)l * cols + k] = d; labels[(size_t)l * cols + k] = j; }
                    }
        }
        for (int j = 0; j < 6 * n; ++j) sum[j] = 0;
        for (int x = 0; x < cols; ++x)
            for (int y = 0; y < rows; ++y) {
                const int id = labels[(size_t)y * cols + x];
                if (id != -1) {
                    const uint8_t *px = LAB(y, x);
                    sum[6 * id] += px[0]; sum[6 * id + 1] += px[1]; sum[6 * id + 2] += px[2];
                    sum[6 * id + 3] += x; sum[6 * id + 4] += y; sum[6 * id + 5] += 1;
                }
            }
        for (int j = 0; j < n; ++j) {
            dead[j] = sum[6 * j + 5] == 0;      /* 0/0: NaN in the reference until pixels carry this label again */
            if (dead[j]) { for (int q = 0; q < 5; ++q) C[5 * j + q] = NAN; continue; }
            for (int q = 0; q < 5; ++q) C[5 * j + q] = (double)sum[6 * j + q] / (double)sum[6 * j + 5];
        }
    }
#undef LAB
    if (centers_out) memcpy(centers_out, C, sizeof(double) * 5 * (size_t)n);
    free(C); free(dist); free(sum); free(dead);
    return n;
}

/* ---- N4: stereo photometric refinement ------------------------------------------------
 * /root/reference/src/DC_stereo_lidar/main_sl.cpp, the sequence main runs at :1165-1246:
 *   EntryType images (:1173-1189): value = the grey byte as float, derivative 0;
 *   calculateMeasuementDerivatives :715-747: derivative.x = .5*v(c+1) - .5*v(c-1) for rows 1..R-2, cols 1..C-2;
 *   get_initial_disparity :846-861: disparity = (baseline*focal)/depth where depth > 0, else 0;
 *   optimize_IG :804-843, 4 sweeps; per pixel: pixel_right = j - disparity; calculateObservationDerivatives
 *     :749-801 with img_point = (i, pixel_right): r = i, c = pixel_right, r0 = (int)(r+0.5) = i, c0 = (int)(c+0.5)
 *     (double sum, truncated); rejected if c0 < 0 or c0 + 1 > cols (:762-772, with its `>` comparisons);
 *     dr = r - r0 = 0, dr1 = 1: the lower patch row has weight 0 and drops out; value and derivative are the linear
 *     interpolation between entries (i, c0) and (i, c0 + 1) -- for c0 + 1 == cols that `at<>` is the next element in
 *     memory, the first pixel of row i + 1 (past the buffer on the last row: the reference's behaviour is undefined
 *     there, this restatement leaves that pixel unchanged);
 *     error = value - left value, clamped to +-255; J = -1; H = (J*dx)^2 + 500; disparity += -(J*dx*error)/H;
 *   retrieve_optimized_depth :863-885: depth = (baseline*focal)/disparity where disparity > 0, capped at 100.
 * f32, one rounding per operation, the reference's order (this file: -ffp-contract=off).  PARITY UNPINNED. */
static float grey_dx(const uint8_t *g, int r, int c, int rows, int cols)
{
    if (r < 1 || r >= rows - 1 || c < 1 || c >= cols - 1) return 0.0f;
    const float a = 0.5f * (float)g[(size_t)r * cols + c + 1];
    const float b = 0.5f * (float)g[(size_t)r * cols + c - 1];
    return a - b;
}

void dcmt_oracle_stereo_refine(const float *depth, const uint8_t *left, const uint8_t *right, float *dst,
                               int rows, int cols, float baseline, float focal, float damp, float max_depth,
                               int iterations)
{
    const size_t fe = (size_t)rows * cols;
    const float bf = baseline * focal;
    float *disp = (float *)malloc(sizeof(float) * fe);
    for (size_t p = 0; p < fe; ++p) disp[p] = depth[p] > 0.0f ? bf / depth[p] : 0.0f;
    for (int k = 0; k < iterations; ++k)
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < cols; ++j) {
                float *dp = disp + (size_t)i * cols + j;
                const float c = (float)j - *dp;
                const int c0 = (int)((double)c + 0.5);
                if (c0 < 0 || c0 + 1 > cols || *dp == 0.0f) continue;
                const size_t e0 = (size_t)i * cols + c0, e1 = e0 + 1;
                if (e1 >= fe) continue;
                const int r1 = (int)(e1 / cols), c1 = (int)(e1 % cols);
                const float dc = c - (float)c0, dc1 = 1.0f - dc;
                float t0 = (float)right[e0] * dc1, t1 = (float)right[e1] * dc;
                const float value = t0 + t1;
                t0 = grey_dx(right, i, c0, rows, cols) * dc1; t1 = grey_dx(right, r1, c1, rows, cols) * dc;
                const float dx = t0 + t1;
                float error = value - (float)left[(size_t)i * cols + j];
                if (error > 255.0f) error = 255.0f;
                if (error < -255.0f) error = -255.0f;
                const float jcr = -1.0f * dx;
                float H = jcr * jcr;
                H = H + damp;
                const float b = jcr * error;
                *dp = *dp + (-b / H);
            }
    for (size_t p = 0; p < fe; ++p) {
        float o = 0.0f;
        if (disp[p] > 0.0f) { o = bf / disp[p]; if (o > max_depth) o = max_depth; }
        dst[p] = o;
    }
    free(disp);
}

/* ---- synthetic KITTI-like sparse frame ---------------------
 * Counter-based: every pixel is a pure function of (seed,row,col), so numpy
 * (depth_completion_mt_amd/synth.py) reproduces it bit for bit. */
static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void dcmt_oracle_synth_frame(float *dst, int rows, int cols, uint64_t seed)
{
    const uint64_t key = splitmix64(seed);
    for (int r = 0; r < rows; ++r) {
        const int re = (int)(((long)r * 352) / rows);      /* KITTI-crop row this row mimics */
        double t = (re - 110) / 60.0;
        t = t < 0.0 ? 0.0 : (t > 1.5 ? 1.5 : t);
        const uint32_t thr = (uint32_t)floor(t * 0.05 * 16777216.0);
        const int den = re - 172 < 2 ? 2 : re - 172;
        double base = (1.65 * 721.5) / den;
        base = base < 2.0 ? 2.0 : (base > 85.0 ? 85.0 : base);
        for (int c = 0; c < cols; ++c) {
            const uint64_t h = splitmix64(key ^ (((uint64_t)(uint32_t)r << 32) | (uint32_t)c));
            const uint32_t u1 = (uint32_t)(h >> 40) & 0xFFFFFFu;
            const uint32_t u2 = (uint32_t)(h >> 16) & 0xFFFFFFu;
            float v = 0.0f;
            if (u1 < thr) {
                double f = 0.6 + 0.8 * (u2 / 16777216.0);
                double d = base * f;
                d = d < 0.5 ? 0.5 : (d > 85.0 ? 85.0 : d);
                double q = floor(d * 256.0 + 0.5);          /* KITTI uint16 PNG units */
                v = (float)(q / 256.0);
            }
            dst[(long)r * cols + c] = v;
        }
    }
}
