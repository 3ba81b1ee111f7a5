// dcmt_kernels_slic.h -- N3: SLIC superpixel labels on the device, the producer of the label plane of
// interpolate_with_superpixels (reference: src/DC_lidar_camera/slic.cpp:101-182, Slic::generate_superpixels, with
// init_data :19-57, compute_dist :59-68, find_local_minimum :71-98).
//
// The reference walks the centres in index order and lets a pixel take a centre whose distance is STRICTLY smaller,
// so the result per pixel is  argmin over (distance, centre index)  of the centres whose [c - step, c + step) window
// contains it -- an order-free definition, evaluated here centre-major with atomics:
//   k_slic_dist<false>   one workgroup per (centre, frame): atomicMin of the f64 distance bits into a u64 plane
//                        (distances are non-negative, so the bit patterns order like the values);
//   k_slic_dist<true>    the same walk again: where this centre's distance IS the plane's minimum, atomicMin of the
//                        centre index into the new-label plane -> the lowest index among exact ties;
//   k_slic_accum         per (centre, frame): sums L, a, b, x, y and the count over the pixels of its window that
//                        now carry its label (integers: exact, order-free), one atomicAdd per quantity;
//   k_slic_merge         per pixel: a pixel no window reached keeps its old label (the reference resets the
//                        distances each iteration, not the clusters) and still counts for its old centre;
//   k_slic_norm          centre = sums / count in f64; a centre without pixels is dead from then on (in the
//                        reference it turns NaN and its window loop `k < NaN` never runs again).
// Distances are computed exactly as the reference's compute_dist: f64, one rounding per operation, no contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dcmt {

constexpr unsigned long long kSlicDead = 0x7ff8000000000000ull;     // bit pattern of a dead centre's x (a NaN)
constexpr int kSlicNoLabel = 0x7f7f7f7f;                             // new-label plane: not reached this iteration

__device__ __forceinline__ double slic_dist(const double* c, int x, int y, const uint8_t* px, double nc, double ns)
{
    const double d0 = c[0] - (double)px[0], d1 = c[1] - (double)px[1], d2 = c[2] - (double)px[2];
    const double dc = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2)));
    const double e0 = c[3] - (double)x, e1 = c[4] - (double)y;
    const double ds = __dsqrt_rn(__dadd_rn(__dmul_rn(e0, e0), __dmul_rn(e1, e1)));
    const double a = __ddiv_rn(dc, nc), b = __ddiv_rn(ds, ns);
    return __dsqrt_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(b, b)));
}

// centre index -> grid position: x outer, y inner (slic.cpp:33-34)
__device__ __forceinline__ void slic_grid(int c, int rows, int step, int& i, int& j)
{
    int ny = 0;
    for (int y = step; y < rows - step / 2; y += step) ++ny;
    i = step * (c / ny + 1);
    j = step * (c % ny + 1);
}

__global__ void k_slic_init(const uint8_t* __restrict__ lab, double* __restrict__ centers, int rows, int cols, int step, int n)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (c >= n) return;
    const uint8_t* img = lab + (size_t)f * rows * cols * 3;
    int i, j;
    slic_grid(c, rows, step, i, j);
    double min_grad = 3.4028234663852886e38;       // FLT_MAX as the reference's double (slic.cpp:72)
    int mx = i, my = j;
    for (int ii = i - 1; ii < i + 2; ++ii)
        for (int jj = j - 1; jj < j + 2; ++jj) {
            const double i1 = img[3 * ((size_t)(jj + 1) * cols + ii)], i2 = img[3 * ((size_t)jj * cols + ii + 1)],
                         i3 = img[3 * ((size_t)jj * cols + ii)];
            const double g = __dadd_rn(fabs(i1 - i3), fabs(i2 - i3));
            if (g < min_grad) { min_grad = g; mx = ii; my = jj; }
        }
    double* C = centers + ((size_t)f * n + c) * 5;
    const uint8_t* px = img + 3 * ((size_t)my * cols + mx);
    C[0] = px[0]; C[1] = px[1]; C[2] = px[2]; C[3] = mx; C[4] = my;
}

// the window of a centre: [k0, k1) x [l0, l1), clipped to the image; false if dead or empty
__device__ __forceinline__ bool slic_window(const double* C, int step, int rows, int cols, int& k0, int& k1, int& l0, int& l1)
{
    if (__builtin_bit_cast(unsigned long long, C[3]) == kSlicDead) return false;
    const double cx = C[3], cy = C[4];
    k0 = (int)(cx - (double)step); l0 = (int)(cy - (double)step);          // `int k = centers[j][3] - step`: truncation
    k1 = (int)__builtin_ceil(cx + (double)step); l1 = (int)__builtin_ceil(cy + (double)step);   // `k < centers[j][3] + step`
    k0 = max(k0, 0); l0 = max(l0, 0); k1 = min(k1, cols); l1 = min(l1, rows);
    return k0 < k1 && l0 < l1;
}

template <bool PICK>
__global__ __launch_bounds__(256)
void k_slic_dist(const uint8_t* __restrict__ lab, const double* __restrict__ centers, unsigned long long* __restrict__ dist,
                 int* __restrict__ label_new, int rows, int cols, int step, int nc, int n)
{
    const int j = blockIdx.x, f = blockIdx.y;
    const double* C = centers + ((size_t)f * n + j) * 5;
    int k0, k1, l0, l1;
    if (!slic_window(C, step, rows, cols, k0, k1, l0, l1)) return;
    const double c[5] = {C[0], C[1], C[2], C[3], C[4]};
    const size_t fo = (size_t)f * rows * cols;
    const int kw = k1 - k0, total = kw * (l1 - l0);
    for (int t = threadIdx.x; t < total; t += 256) {
        const int l = l0 + t / kw, k = k0 + t % kw;
        const size_t p = fo + (size_t)l * cols + k;
        const unsigned long long d = __builtin_bit_cast(unsigned long long, slic_dist(c, k, l, lab + 3 * p, (double)nc, (double)step));
        if constexpr (PICK) { if (d == dist[p]) atomicMin(&label_new[p], j); }
        else atomicMin(&dist[p], d);
    }
}

__global__ __launch_bounds__(256)
void k_slic_accum(const uint8_t* __restrict__ lab, const double* __restrict__ centers, const int* __restrict__ label_new,
                  unsigned long long* __restrict__ sums, int rows, int cols, int step, int n)
{
    __shared__ unsigned long long s_part[4][6];
    const int j = blockIdx.x, f = blockIdx.y;
    const double* C = centers + ((size_t)f * n + j) * 5;
    int k0, k1, l0, l1;
    if (!slic_window(C, step, rows, cols, k0, k1, l0, l1)) return;
    const size_t fo = (size_t)f * rows * cols;
    const int kw = k1 - k0, total = kw * (l1 - l0);
    unsigned long long a[6] = {0, 0, 0, 0, 0, 0};
    for (int t = threadIdx.x; t < total; t += 256) {
        const int l = l0 + t / kw, k = k0 + t % kw;
        const size_t p = fo + (size_t)l * cols + k;
        if (label_new[p] == j) {
            const uint8_t* px = lab + 3 * p;
            a[0] += px[0]; a[1] += px[1]; a[2] += px[2]; a[3] += (unsigned)k; a[4] += (unsigned)l; a[5] += 1;
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[q] += __shfl_down(a[q], o, 64);
    }
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < 6; ++q) s_part[threadIdx.x >> 6][q] = a[q];
    __syncthreads();
    if (threadIdx.x < 6) {
        const unsigned long long v = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        if (v) atomicAdd(&sums[((size_t)f * n + j) * 6 + threadIdx.x], v);
    }
}

__global__ __launch_bounds__(256)
void k_slic_merge(const uint8_t* __restrict__ lab, const int* __restrict__ label_new, int* __restrict__ labels,
                  unsigned long long* __restrict__ sums, int rows, int cols, int n, int batch)
{
    const size_t fe = (size_t)rows * cols, total = fe * batch;
    for (size_t p = blockIdx.x * (size_t)256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const int ln = label_new[p];
        if (ln != kSlicNoLabel) { labels[p] = ln; continue; }
        const int old = labels[p];
        if (old < 0) continue;
        const size_t f = p / fe, r = p - f * fe;
        unsigned long long* s = sums + (f * n + old) * 6;
        const uint8_t* px = lab + 3 * p;
        atomicAdd(&s[0], (unsigned long long)px[0]); atomicAdd(&s[1], (unsigned long long)px[1]); atomicAdd(&s[2], (unsigned long long)px[2]);
        atomicAdd(&s[3], (unsigned long long)(r % cols)); atomicAdd(&s[4], (unsigned long long)(r / cols)); atomicAdd(&s[5], 1ull);
    }
}

__global__ void k_slic_norm(const unsigned long long* __restrict__ sums, double* __restrict__ centers, int total)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total) return;
    const unsigned long long* s = sums + (size_t)c * 6;
    double* C = centers + (size_t)c * 5;
    if (s[5] == 0) {
        for (int q = 0; q < 5; ++q) C[q] = __builtin_bit_cast(double, kSlicDead);
        return;
    }
    const double cnt = (double)s[5];
    for (int q = 0; q < 5; ++q) C[q] = __ddiv_rn((double)s[q], cnt);
}

}  // namespace dcmt
