// dcmt_kernels_slic.h -- N3: SLIC superpixel labels on the device, the producer of the label plane of
// interpolate_with_superpixels (reference: src/DC_lidar_camera/slic.cpp:101-182, Slic::generate_superpixels, with
// init_data :19-57, compute_dist :59-68, find_local_minimum :71-98).
//
// The reference walks the centres in index order and lets a pixel take a centre whose distance is STRICTLY smaller,
// so the result per pixel is  argmin over (distance, centre index)  of the centres whose [c - step, c + step) window
// contains it -- an order-free definition.  Per iteration:
//   k_slic_bin      centres -> cells of step x step pixels (count + short index list per cell);
//   k_slic_assign   one thread per pixel: the candidates are the centres of the 3 x 3 cells around it whose window
//                   holds the pixel; the winner is found on the un-rooted distance, and only candidates within 4e-12
//                   of the minimum (exact ties included) are decided with the reference's own f64 arithmetic
//                   (three sqrt, two divisions) as (distance, index).  A pixel no window reaches keeps its old label
//                   (the reference resets the distances each iteration, not the clusters) and still counts for its
//                   old centre;
//                   the winners' L, a, b, x, y and a count are summed per centre as integers (exact, order-free), in
//                   LDS per tile and flushed with global atomics;
//   k_slic_norm     centre = sums / count in f64; a centre without pixels is dead from then on (in the
//                   reference it turns NaN and its window loop `k < NaN` never runs again).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dcmt {

constexpr unsigned long long kSlicDead = 0x7ff8000000000000ull;     // bit pattern of a dead centre's x (a NaN)

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

// ---- pixel-major assignment -------------------------------------------------------------------------------------
// Centres are binned into cells of cell_px x cell_px pixels, cell_px = step (any cell_px >= step is correct: tests use
// larger cells to force list overflows) (k_slic_bin: per cell a count and up to kSlicCellCap indices);
// a centre whose window [c - step, c + step) contains pixel x has floor(c / step) within one cell of floor(x / step),
// so every pixel looks at the 3 x 3 cells around its own.  If any cell of a frame overflows its list, the frame's
// pixels walk all centres instead (correct, slow, never seen on SLIC-like data).
//
// The reference's compute_dist costs three square roots and two divisions in f64.  Which candidate is smallest is
// almost always decided by the un-rooted sum  q = Sc / nc^2 + Ss / ns^2  (Sc, Ss the two sums of squares): the value
// the reference computes differs from sqrt(q) by at most a few units of 1e-16 relative (eight roundings), so if the
// smallest q is below every other by more than a factor 1 - 1e-12 the reference's comparison has the same winner.
// Only candidates inside that band (exact ties included) are evaluated with slic_dist itself and compared as
// (distance, centre index), which is the reference's rule.
constexpr int kSlicCellCap = 4;

__global__ void k_slic_bin(const double* __restrict__ centers, int* __restrict__ cell_cnt, int* __restrict__ cell_list,
                           int* __restrict__ overflow, int cell_px, int n, int gx, int gy)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (j >= n) return;
    const double* C = centers + ((size_t)f * n + j) * 5;
    if (__builtin_bit_cast(unsigned long long, C[3]) == kSlicDead) return;
    const int cx = min(max((int)(C[3] / (double)cell_px), 0), gx - 1), cy = min(max((int)(C[4] / (double)cell_px), 0), gy - 1);
    const size_t cell = ((size_t)f * gy + cy) * gx + cx;
    const int slot = atomicAdd(&cell_cnt[cell], 1);
    if (slot < kSlicCellCap) cell_list[cell * kSlicCellCap + slot] = j;
    else overflow[f] = 1;
}

// One workgroup per tile of kSlicTW x kSlicTH pixels.  The cells that can hold a candidate of any pixel of the tile
// (the tile's cell range grown by one) are staged in LDS once -- counts, centre indices and the centres themselves --
// and every thread then works through its pixels (one column, kSlicTH / 4 rows) against the 3 x 3 cells around each.
// One sweep keeps the smallest and second smallest un-rooted distance; only if the second is inside the band around
// the first (ties included) does the pixel take the exact second sweep.  The winners' L, a, b, x, y, 1 are summed
// per staged centre in LDS (32-bit: a tile holds at most 1024 pixels) and flushed with one global atomicAdd per
// non-zero entry, so no separate accumulation pass over the image is needed.
constexpr int kSlicTW = 64, kSlicTH = 16, kSlicMaxCells = 64;   // 35 KB of LDS; steps below 8 can exceed 64 cells and take the global walk

__global__ __launch_bounds__(256)
void k_slic_assign(const uint8_t* __restrict__ lab, const double* __restrict__ centers, const int* __restrict__ cell_cnt,
                   const int* __restrict__ cell_list, const int* __restrict__ overflow, int* __restrict__ labels,
                   unsigned long long* __restrict__ sums, int rows, int cols, int step, int nc, int n, int gx, int gy, int cell_px)
{
    __shared__ double s_c[kSlicMaxCells * kSlicCellCap][5];
    __shared__ int s_idx[kSlicMaxCells * kSlicCellCap];
    __shared__ unsigned s_acc[kSlicMaxCells * kSlicCellCap][6];
    __shared__ int s_cnt[kSlicMaxCells];
    const int f = blockIdx.z, tx0 = blockIdx.x * kSlicTW, ty0 = blockIdx.y * kSlicTH;
    const int cxa = max(tx0 / cell_px - 1, 0), cxb = min((min(tx0 + kSlicTW, cols) - 1) / cell_px + 1, gx - 1);
    const int cya = max(ty0 / cell_px - 1, 0), cyb = min((min(ty0 + kSlicTH, rows) - 1) / cell_px + 1, gy - 1);
    const int ncx = cxb - cxa + 1, ncells = ncx * (cyb - cya + 1);
    const bool all = overflow[f] != 0 || ncells > kSlicMaxCells;       // walk every centre from global memory instead
    const double* CF = centers + (size_t)f * n * 5;
    if (!all) {
        for (int c = threadIdx.x; c < ncells; c += 256) {
            const size_t cell = ((size_t)f * gy + cya + c / ncx) * gx + cxa + c % ncx;
            s_cnt[c] = min(cell_cnt[cell], kSlicCellCap);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < ncells * kSlicCellCap; e += 256) {
            const int c = e / kSlicCellCap, k = e % kSlicCellCap;
            if (k < s_cnt[c]) {
                const size_t cell = ((size_t)f * gy + cya + c / ncx) * gx + cxa + c % ncx;
                const int j = cell_list[cell * kSlicCellCap + k];
                s_idx[e] = j;
#pragma unroll
                for (int q = 0; q < 5; ++q) s_c[e][q] = CF[(size_t)j * 5 + q];
#pragma unroll
                for (int q = 0; q < 6; ++q) s_acc[e][q] = 0u;
            }
        }
        __syncthreads();
    }
    const int x = tx0 + (threadIdx.x & 63);
    const double inc2 = 1.0 / ((double)nc * (double)nc), ins2 = 1.0 / ((double)step * (double)step), fx = x;
    for (int r = 0; r < kSlicTH / 4; ++r) {
        const int y = ty0 + (threadIdx.x >> 6) * (kSlicTH / 4) + r;
        if (x >= cols || y >= rows) continue;
        const size_t p = ((size_t)f * rows + y) * cols + x;
        const uint8_t* px = lab + 3 * p;
        const double p0 = px[0], p1 = px[1], p2 = px[2], fy = y;
        // does the centre C cover the pixel (exactly the reference's loop bounds)?  if so, its un-rooted distance
        auto probe = [&](const double* C, bool checked, double& q) -> bool {    // checked: the integer window already passed
            const double cx = C[3], cy = C[4];
            if (!checked) {
                if (__builtin_bit_cast(unsigned long long, cx) == kSlicDead) return false;
                if (x < (int)(cx - (double)step) || !(fx < cx + (double)step) || y < (int)(cy - (double)step) || !(fy < cy + (double)step)) return false;
            }
            const double d0 = C[0] - p0, d1 = C[1] - p1, d2 = C[2] - p2, e0 = cx - fx, e1 = cy - fy;
            q = (d0 * d0 + d1 * d1 + d2 * d2) * inc2 + (e0 * e0 + e1 * e1) * ins2;
            return true;
        };
        double q1 = 1.0e300, q2 = 1.0e300;         // smallest and second smallest un-rooted distance
        int e1 = -1;                                // entry (LDS slot, or centre index in the `all` walk) of the smallest
        auto sweep = [&](auto&& visit) {
            if (all) { for (int j = 0; j < n; ++j) visit(CF + (size_t)j * 5, j, false); return; }
            const int lcx = x / cell_px - cxa, lcy = y / cell_px - cya;
            for (int cy = max(lcy - 1, 0); cy <= min(lcy + 1, cyb - cya); ++cy)
                for (int cx = max(lcx - 1, 0); cx <= min(lcx + 1, ncx - 1); ++cx) {
                    const int c = cy * ncx + cx;
                    for (int k = 0; k < s_cnt[c]; ++k) visit(s_c[c * kSlicCellCap + k], c * kSlicCellCap + k, false);
                }
        };
        sweep([&](const double* C, int e, bool checked) {
            double q;
            if (!probe(C, checked, q)) return;
            if (q < q1) { q2 = q1; q1 = q; e1 = e; } else if (q < q2) q2 = q;
        });
        int best;                                   // winning centre index
        if (e1 < 0) {                               // no window reaches this pixel: it keeps its label and still counts for it
            best = labels[p];
            if (best < 0) continue;
            e1 = -1;
        } else if (q2 > q1 * (1.0 + 4.0e-12) + 1.0e-300) {
            best = all ? e1 : s_idx[e1];
        } else {                                    // inside the band: the reference's own arithmetic decides, (distance, index)
            const double band = q1 * (1.0 + 4.0e-12) + 1.0e-300;
            double dbest = 0.0;
            best = 0x7fffffff;
            int ebest = -1;
            sweep([&](const double* C, int e, bool checked) {
                double q;
                if (!probe(C, checked, q) || q > band) return;
                const int j = all ? e : s_idx[e];
                const double d = slic_dist(C, x, y, px, (double)nc, (double)step);
                if (ebest < 0 || d < dbest || (d == dbest && j < best)) { dbest = d; best = j; ebest = e; }
            });
            e1 = ebest;
        }
        if (e1 >= 0) labels[p] = best;
        if (!all && e1 >= 0) {
            atomicAdd(&s_acc[e1][0], (unsigned)px[0]); atomicAdd(&s_acc[e1][1], (unsigned)px[1]); atomicAdd(&s_acc[e1][2], (unsigned)px[2]);
            atomicAdd(&s_acc[e1][3], (unsigned)x); atomicAdd(&s_acc[e1][4], (unsigned)y); atomicAdd(&s_acc[e1][5], 1u);
        } else {
            unsigned long long* sj = sums + ((size_t)f * n + best) * 6;
            atomicAdd(&sj[0], (unsigned long long)px[0]); atomicAdd(&sj[1], (unsigned long long)px[1]); atomicAdd(&sj[2], (unsigned long long)px[2]);
            atomicAdd(&sj[3], (unsigned long long)x); atomicAdd(&sj[4], (unsigned long long)y); atomicAdd(&sj[5], 1ull);
        }
    }
    if (!all) {
        __syncthreads();
        for (int e = threadIdx.x; e < ncells * kSlicCellCap; e += 256) {
            if (e % kSlicCellCap < s_cnt[e / kSlicCellCap] && s_acc[e][5] != 0u) {
                unsigned long long* sj = sums + ((size_t)f * n + s_idx[e]) * 6;
#pragma unroll
                for (int q = 0; q < 6; ++q) atomicAdd(&sj[q], (unsigned long long)s_acc[e][q]);
            }
        }
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
