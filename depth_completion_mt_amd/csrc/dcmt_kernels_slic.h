// dcmt_kernels_slic.h -- N3: SLIC superpixel labels on the device, the producer of the label plane of
// interpolate_with_superpixels (reference: src/DC_lidar_camera/slic.cpp:101-182, Slic::generate_superpixels, with
// init_data :19-57, compute_dist :59-68, find_local_minimum :71-98).
//
// The reference walks the centres in index order and lets a pixel take a centre whose distance is STRICTLY smaller,
// so the result per pixel is  argmin over (distance, centre index)  of the centres whose [c - step, c + step) window
// contains it -- an order-free definition.  Per iteration:
//   k_slic_assign   one thread per pixel: the candidates are the centres of the 3 x 3 cells around it whose window
//                   holds the pixel; the winner is found on the un-rooted distance, and only candidates within 4e-12
//                   of the minimum (exact ties included) are decided with the reference's own f64 arithmetic
//                   (three sqrt, two divisions) as (distance, index).  A pixel no window reaches keeps its old label
//                   (the reference resets the distances each iteration, not the clusters) and still counts for its
//                   old centre;
//                   the winners' L, a, b, x, y and a count are summed per centre as integers (exact, order-free), in
//                   LDS per tile and flushed with global atomics;
//   k_slic_norm_bin centre = sums / count in f64; a centre without pixels is dead from then on (in the
//                   reference it turns NaN and its window loop `k < NaN` never runs again); the new centres go into
//                   cells of step x step pixels (count + short index list per cell) for the next assignment.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dcmt {

constexpr unsigned long long kSlicDead = 0x7ff8000000000000ull;     // bit pattern of a dead centre's x (a NaN)
constexpr int kSlicCellCap = 4;                                     // centre indices a cell's list holds

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

// a centre goes into the list of its cell (at most kSlicCellCap per cell; more: the frame's overflow flag)
__device__ __forceinline__ void slic_bin_one(const double* C, int j, int f, int* cell_cnt, int* cell_list, int* overflow,
                                             int cell_px, int gx, int gy)
{
    if (__builtin_bit_cast(unsigned long long, C[3]) == kSlicDead) return;
    const int cx = min(max((int)(C[3] / (double)cell_px), 0), gx - 1), cy = min(max((int)(C[4] / (double)cell_px), 0), gy - 1);
    const size_t cell = ((size_t)f * gy + cy) * gx + cx;
    const int slot = atomicAdd(&cell_cnt[cell], 1);
    if (slot < kSlicCellCap) cell_list[cell * kSlicCellCap + slot] = j;
    else overflow[f] = 1;
}

// initial centres (slic.cpp:19-57), binned into the (zeroed) cell set the first assignment reads
__global__ void k_slic_init(const uint8_t* __restrict__ lab, double* __restrict__ centers, int rows, int cols, int step, int n,
                            int* __restrict__ cell_cnt, int* __restrict__ cell_list, int* __restrict__ overflow, int cell_px, int gx, int gy)
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
    slic_bin_one(C, c, f, cell_cnt, cell_list, overflow, cell_px, gx, gy);
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
// larger cells to force list overflows) (slic_bin_one: per cell a count and up to kSlicCellCap indices);
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

// One workgroup per tile of kSlicTW x TH pixels.  Staged in LDS once per tile: the entries (centre index, centre,
// integer window [k0, k1) x [l0, l1) = the reference's loop bounds, slic_window) of the cells that can hold a candidate
// of any pixel of the tile (the tile's cell range grown by one); per cell the flat list of the entries of the 3 x 3 cells
// around it (at most 32 -- a cell whose list is longer sends the tile to the slow walk); and, as bit masks over those
// lists, which entries' windows hold a given column (per cell row and tile column) and a given row (per cell column and
// tile row).  A pixel's candidates are then the AND of two masks -- about 4 of the 9 entries around it -- and a thread
// walks only over the set bits: every trip of the loop computes a distance that is needed (the loop over all nine ran
// the arithmetic for every lane whenever any lane's candidate passed: 322 -> 180 VALU instructions per 64 pixels).
// One sweep keeps the smallest and second smallest un-rooted distance; only if the second is inside the band around
// the first (ties included) does the pixel take the exact second sweep.
//
// The winners' L, a, b, x, y, 1 are summed per staged centre in LDS (32-bit: a tile holds at most 4096 pixels) and
// flushed with one global atomicAdd per non-zero entry, so no separate accumulation pass over the image is needed.
// The lanes of a wave are 64 neighbouring pixels of one row: most of them add to the SAME centre, and same-address LDS
// atomics of one instruction are served one lane after the other (measured: a quarter of the kernel at 18-pixel
// superpixels, half of it at 68).  So a thread works down a column (TH / 4 rows), sums the run of equal winners in
// registers and adds a run to the table when it ends.  (Several copies of the table, lane & 3: slower -- LDS, occupancy.)
// Tile height (template parameter TH): as tall as the tile's cells fit -- the staging is paid once per tile and a thread's
// runs get longer -- 64 rows = 16 per thread for steps from 16 up (at most 7 x 7 staged cells), 32 for steps 11 to 15
// (9 x 6), 16 for the steps below (6: at most 14 x 6 = 84 of the 128 cells a tile may stage).
constexpr int kSlicTW = 64, kSlicMaxCells = 128;
__host__ __device__ constexpr int slic_tile_rows(int step) { return step >= 16 ? 64 : (step >= 11 ? 32 : 16); }
constexpr int kSlicEntries = 128;                                // staged centres per tile, numbered densely (more: slow walk)
constexpr int kSlicListCap = 9 * kSlicCellCap;
constexpr int kSlicMaskBits = 32;
constexpr int kSlicInnerX = 12, kSlicInnerY = 8;                // cells a tile's own pixels may span (more: slow walk)
static_assert(kSlicMaxCells == 128, "the prefix over the staged cells takes two cells per lane");
static_assert(kSlicEntries <= 256 && (kSlicEntries & (kSlicEntries - 1)) == 0, "entry numbers are stored as bytes; stale list bytes are masked into range");

template <int TH>
__global__ __launch_bounds__(256)
void k_slic_assign(const uint8_t* __restrict__ lab, const double* __restrict__ centers, const int* __restrict__ cell_cnt,
                   const int* __restrict__ cell_list, const int* __restrict__ overflow, int* __restrict__ labels,
                   unsigned long long* __restrict__ sums, int rows, int cols, int step, int nc, int n, int gx, int gy, int cell_px)
{
    __shared__ __attribute__((aligned(16))) double s_c[kSlicEntries][6];      // per staged centre: the affine form of its screening distance (below)
    __shared__ int4 s_win[kSlicEntries];
    __shared__ int s_idx[kSlicEntries];
    __shared__ unsigned s_acc[kSlicEntries][6];
    __shared__ int s_cnt[kSlicMaxCells], s_base[kSlicMaxCells + 1];       // entries of cell c: s_base[c] .. s_base[c] + s_cnt[c] - 1
    __shared__ __attribute__((aligned(4))) uint8_t s_list[kSlicMaxCells][kSlicListCap];
    __shared__ int s_nlist[kSlicMaxCells];
    __shared__ unsigned s_xmask[kSlicInnerY][kSlicTW];
    __shared__ unsigned s_ymask[kSlicInnerX][TH];
    __shared__ int s_slow, s_slow2;                  // too many entries / too long a list (separate flags: each is read behind its own barrier)
    const int f = blockIdx.z, tx0 = blockIdx.x * kSlicTW, ty0 = blockIdx.y * TH;
    const int tx1 = min(tx0 + kSlicTW, cols), ty1 = min(ty0 + TH, rows);          // the tile's pixels: [tx0, tx1) x [ty0, ty1)
    const int icx0 = tx0 / cell_px, icx1 = (tx1 - 1) / cell_px, icy0 = ty0 / cell_px, icy1 = (ty1 - 1) / cell_px;   // their cells
    const int cxa = max(icx0 - 1, 0), cxb = min(icx1 + 1, gx - 1), cya = max(icy0 - 1, 0), cyb = min(icy1 + 1, gy - 1);
    const int ncx = cxb - cxa + 1, ncy = cyb - cya + 1, ncells = ncx * ncy;
    const int nix = icx1 - icx0 + 1, niy = icy1 - icy0 + 1;
    const double* CF = centers + (size_t)f * n * 5;
    const int x = tx0 + (threadIdx.x & 63);
    const int yb = ty0 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (TH / 4);   // wave-uniform: the row arithmetic stays scalar
    const double inc2 = 1.0 / ((double)nc * (double)nc), ins2 = 1.0 / ((double)step * (double)step), fx = x;
    // bound of the rounding error of a screening distance in its affine form: every intermediate is below `mag` in magnitude
    // (colour terms up to 3 * 255^2 / nc^2; tile-relative coordinates up to kSlicTW + 2 cells, TH + 2 cells) and fewer than 32
    // roundings of 2^-53 relative each are involved
    const double ext = (double)(kSlicTW + TH + 4 * cell_px);
    const double delta = 0x1p-48 * (2.0 * 195075.0 * inc2 + 4.0 * ext * ext * ins2 + 1.0);

    // ---- the slow walk: every centre, windows tested as the reference's loop bounds, sums straight to global memory
    auto slow_walk = [&]() {
        for (int r = 0; r < TH / 4; ++r) {
            const int y = yb + r;
            if (x >= cols || y >= rows) continue;
            const size_t p = ((size_t)f * rows + y) * cols + x;
            const uint8_t* px = lab + 3 * p;
            const double p0 = px[0], p1 = px[1], p2 = px[2], fy = y;
            auto probe = [&](const double* C, double& q) -> bool {
                const double cx = C[3], cy = C[4];
                if (__builtin_bit_cast(unsigned long long, cx) == kSlicDead) return false;
                if (x < (int)(cx - (double)step) || !(fx < cx + (double)step) || y < (int)(cy - (double)step) || !(fy < cy + (double)step)) return false;
                const double d0 = C[0] - p0, d1 = C[1] - p1, d2 = C[2] - p2, e0 = cx - fx, e1 = cy - fy;
                q = (d0 * d0 + d1 * d1 + d2 * d2) * inc2 + (e0 * e0 + e1 * e1) * ins2;
                return true;
            };
            double q1 = 1.0e300, q2 = 1.0e300;
            int best = -1;
            for (int j = 0; j < n; ++j) {
                double q;
                if (!probe(CF + (size_t)j * 5, q)) continue;
                if (q < q1) { q2 = q1; q1 = q; best = j; } else if (q < q2) q2 = q;
            }
            if (best < 0) {                             // no window reaches this pixel: it keeps its label and still counts for it
                best = labels[p];
                if (best < 0) continue;
            } else {
                const double band = q1 * (1.0 + 4.0e-12) + 1.0e-300;
                if (!(q2 > band)) {                     // inside the band: the reference's own arithmetic decides, (distance, index)
                    double dbest = 0.0;
                    best = -1;
                    for (int j = 0; j < n; ++j) {
                        double q;
                        if (!probe(CF + (size_t)j * 5, q) || q > band) continue;
                        const double d = slic_dist(CF + (size_t)j * 5, x, y, px, (double)nc, (double)step);
                        if (best < 0 || d < dbest) { dbest = d; best = j; }      // ascending j: the first of equal distances stays
                    }
                }
                labels[p] = best;
            }
            unsigned long long* sj = sums + ((size_t)f * n + best) * 6;
            atomicAdd(&sj[0], (unsigned long long)px[0]); atomicAdd(&sj[1], (unsigned long long)px[1]); atomicAdd(&sj[2], (unsigned long long)px[2]);
            atomicAdd(&sj[3], (unsigned long long)x); atomicAdd(&sj[4], (unsigned long long)y); atomicAdd(&sj[5], 1ull);
        }
    };
    if (overflow[f] != 0 || ncells > kSlicMaxCells || nix > kSlicInnerX || niy > kSlicInnerY) { slow_walk(); return; }   // block-uniform

    // ---- staging
    if (threadIdx.x == 0) { s_slow = 0; s_slow2 = 0; }
    for (int c = threadIdx.x; c < ncells; c += 256) {
        const size_t cell = ((size_t)f * gy + cya + c / ncx) * gx + cxa + c % ncx;
        s_cnt[c] = min(cell_cnt[cell], kSlicCellCap);
    }
    __syncthreads();
    if (threadIdx.x < 64) {                        // exclusive prefix of the counts over the (at most 128) staged cells, one wave, two cells per lane
        const int l = threadIdx.x, own0 = l < ncells ? s_cnt[l] : 0, own1 = l + 64 < ncells ? s_cnt[l + 64] : 0;
        int i0 = own0, i1 = own1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o0 = __shfl_up(i0, d), o1 = __shfl_up(i1, d);
            i0 += l >= d ? o0 : 0; i1 += l >= d ? o1 : 0;
        }
        const int first_half = __shfl(i0, 63);
        s_base[l] = i0 - own0;
        s_base[l + 64] = first_half + i1 - own1;
        if (l == 63) { s_base[128] = first_half + i1; if (first_half + i1 > kSlicEntries) s_slow = 1; }
    }
    __syncthreads();
    if (s_slow) { slow_walk(); return; }                                 // block-uniform
    const int n_entries = s_base[kSlicMaxCells];
    for (int t = threadIdx.x; t < ncells * kSlicCellCap; t += 256) {
        const int c = t / kSlicCellCap, k = t % kSlicCellCap;
        if (k < s_cnt[c]) {
            const int e = s_base[c] + k;
            const size_t cell = ((size_t)f * gy + cya + c / ncx) * gx + cxa + c % ncx;
            const int j = cell_list[cell * kSlicCellCap + k];
            s_idx[e] = j;
            double C[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) C[q] = CF[(size_t)j * 5 + q];
            // q(pixel) = |C_lab - p_lab|^2 / nc^2 + |C_xy - p_xy|^2 / step^2 = A + B . (L, a, b, x, y) + (a term of the pixel alone),
            // with x, y relative to the tile's origin to keep the products small: five fused multiply-adds per candidate
            const double cx = C[3] - (double)tx0, cy = C[4] - (double)ty0;
            s_c[e][0] = -2.0 * C[0] * inc2; s_c[e][1] = -2.0 * C[1] * inc2; s_c[e][2] = -2.0 * C[2] * inc2;
            s_c[e][3] = -2.0 * cx * ins2; s_c[e][4] = -2.0 * cy * ins2;
            s_c[e][5] = (C[0] * C[0] + C[1] * C[1] + C[2] * C[2]) * inc2 + (cx * cx + cy * cy) * ins2;
            int k0 = 0, k1 = 0, l0 = 0, l1 = 0;
            if (!slic_window(C, step, rows, cols, k0, k1, l0, l1)) { k0 = k1 = l0 = l1 = 0; }     // dead or empty: holds no pixel
            s_win[e] = make_int4(k0, k1 - k0, l0, l1 - l0);                                       // origin and extent: one unsigned compare each
#pragma unroll
            for (int q = 0; q < 6; ++q) s_acc[e][q] = 0u;
        }
    }
    // the candidates of the pixels of cell c: the entries of the 3 x 3 staged cells around it, row by row.  One thread per
    // (cell, neighbour): its entries start behind those of the neighbours before it.
    for (int t = threadIdx.x; t < ncells * 9; t += 256) {
        const int c = t / 9, nb = t - 9 * c, ccx = c % ncx, ccy = c / ncx;
        int off = 0, mine = 0, dmine = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int cy = ccy + q / 3 - 1, cx = ccx + q % 3 - 1;
            const bool ok = cy >= 0 && cy < ncy && cx >= 0 && cx < ncx;
            const int d = ok ? cy * ncx + cx : 0;
            const int cnt = ok ? s_cnt[d] : 0;
            off += q < nb ? cnt : 0;
            if (q == nb) { mine = cnt; dmine = d; }
        }
        for (int k = 0; k < mine; ++k) s_list[c][off + k] = (uint8_t)(s_base[dmine] + k);
        if (nb == 8) {
            s_nlist[c] = off + mine;
            if (off + mine > kSlicMaskBits) s_slow2 = 1;
        }
    }
    __syncthreads();
    if (s_slow2) { slow_walk(); return; }                                 // block-uniform
    // which entries of its cell's list hold tile column xx (per cell row iy) / tile row yy (per cell column ix); four entries
    // per trip (their window reads are independent; list bytes past the end are in range and masked out)
    auto window_mask = [&](int c, int v, bool rows_of_window) -> unsigned {
        const int nl = s_nlist[c];
        const unsigned* lw = reinterpret_cast<const unsigned*>(s_list[c]);
        unsigned m = 0u;
        for (int i = 0; i < nl; i += 4) {
            const unsigned w4 = lw[i >> 2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int4 w = s_win[(w4 >> (8 * k)) & (kSlicEntries - 1)];
                const bool in = rows_of_window ? (unsigned)(v - w.z) < (unsigned)w.w : (unsigned)(v - w.x) < (unsigned)w.y;
                m |= (in && i + k < nl ? 1u : 0u) << (i + k);
            }
        }
        return m;
    };
    for (int t = threadIdx.x; t < niy * kSlicTW; t += 256) {
        const int iy = t / kSlicTW, xx = t % kSlicTW, gxx = tx0 + xx;
        s_xmask[iy][xx] = gxx < cols ? window_mask((icy0 + iy - cya) * ncx + (gxx / cell_px - cxa), gxx, false) : 0u;
    }
    for (int t = threadIdx.x; t < nix * TH; t += 256) {
        const int ix = t / TH, yy = t % TH, gyy = ty0 + yy;
        s_ymask[ix][yy] = gyy < rows ? window_mask((gyy / cell_px - cya) * ncx + (icx0 + ix - cxa), gyy, true) : 0u;
    }
    __syncthreads();

    if (x < cols) {
        const int lcx = x / cell_px - cxa, ix = x / cell_px - icx0, xx = x - tx0;
        int run = -1;                                // entry of the current run of equal winners down this column
        unsigned rl = 0, ra = 0, rb = 0, ry = 0, rn = 0;
        auto flush = [&]() {
            if (run >= 0) {
                atomicAdd(&s_acc[run][0], rl); atomicAdd(&s_acc[run][1], ra); atomicAdd(&s_acc[run][2], rb);
                atomicAdd(&s_acc[run][3], rn * (unsigned)x); atomicAdd(&s_acc[run][4], ry); atomicAdd(&s_acc[run][5], rn);
            }
        };
        for (int r = 0; r < TH / 4; ++r) {
            const int y = yb + r;
            if (y >= rows) break;
            const size_t p = ((size_t)f * rows + y) * cols + x;
            const uint8_t* px = lab + 3 * p;
            const unsigned u0 = px[0], u1 = px[1], u2 = px[2];
            const double p0 = u0, p1 = u1, p2 = u2;
            const int cy = y / cell_px;
            const uint8_t* list = s_list[(cy - cya) * ncx + lcx];
            const unsigned cand = s_xmask[cy - icy0][xx] & s_ymask[ix][y - ty0];   // the entries whose windows hold this pixel
            // screening value of entry e without the pixel's own term: A + B . (L, a, b, x, y)
            const double fxl = xx, fyl = y - ty0;
            auto affine = [&](int e) -> double {
                const double* B = s_c[e];
                return __builtin_fma(B[4], fyl, __builtin_fma(B[3], fxl, __builtin_fma(B[2], p2, __builtin_fma(B[1], p1, __builtin_fma(B[0], p0, B[5])))));
            };
            if (cand == 0u) {                           // no window reaches this pixel: it keeps its label and still counts for it
                const int old = labels[p];
                if (old >= 0) {
                    unsigned long long* sj = sums + ((size_t)f * n + old) * 6;
                    atomicAdd(&sj[0], (unsigned long long)u0); atomicAdd(&sj[1], (unsigned long long)u1); atomicAdd(&sj[2], (unsigned long long)u2);
                    atomicAdd(&sj[3], (unsigned long long)x); atomicAdd(&sj[4], (unsigned long long)y); atomicAdd(&sj[5], 1ull);
                }
                continue;
            }
            double a1 = 1.0e300, a2 = 1.0e300;         // smallest and second smallest screening value
            int e1 = -1;                                // entry of the smallest
            for (unsigned m = cand; m != 0u; m &= m - 1u) {
                const int e = list[__builtin_ctz(m)];
                const double a = affine(e);
                a2 = fmin(a2, fmax(a, a1));             // (a == a1 leaves a2 == a1: a tie goes to the exact sweep)
                e1 = a < a1 ? e : e1;
                a1 = fmin(a1, a);
            }
            // the un-rooted distances themselves: + the pixel's own term.  They carry the rounding of the affine form (delta, an
            // absolute bound); the reference's value is the root of the true sum to a few 1e-16 relative.  A minimum that is clear
            // by more than the band has the same winner in the reference's arithmetic.
            const double own = __builtin_fma(__builtin_fma(fyl, fyl, fxl * fxl), ins2, __builtin_fma(p2, p2, __builtin_fma(p1, p1, p0 * p0)) * inc2);
            const double q1 = a1 + own, q2 = a2 + own;
            const double band = q1 * (1.0 + 4.0e-12) + 4.0 * delta;
            if (!(q2 > band)) {                         // inside the band: the reference's own arithmetic decides, (distance, index)
                double dbest = 0.0;
                int jbest = 0x7fffffff;
                e1 = -1;
                for (unsigned m = cand; m != 0u; m &= m - 1u) {
                    const int e = list[__builtin_ctz(m)];
                    if (affine(e) + own > band) continue;
                    const int j = s_idx[e];
                    const double d = slic_dist(CF + (size_t)j * 5, x, y, px, (double)nc, (double)step);
                    if (e1 < 0 || d < dbest || (d == dbest && j < jbest)) { dbest = d; jbest = j; e1 = e; }
                }
            }
            labels[p] = s_idx[e1];
            if (e1 != run) { flush(); run = e1; rl = ra = rb = ry = rn = 0u; }
            rl += u0; ra += u1; rb += u2; ry += (unsigned)y; rn += 1u;
        }
        flush();
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_entries; e += 256) {
        if (s_acc[e][5] != 0u) {
            unsigned long long* sj = sums + ((size_t)f * n + s_idx[e]) * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) atomicAdd(&sj[q], (unsigned long long)s_acc[e][q]);
        }
    }
}

// End of an iteration, one launch: centre = sums / count in f64 (dead without pixels); the sums are left zeroed for the next
// iteration; the new centre is binned into the OTHER cell set (zeroed by the previous launch of this kernel, or by the host
// before the first), and the cell set the assignment has just read is zeroed for the iteration after the next.  Threads
// beyond the centres only zero.  (Separately -- two memsets, a binning and a normalising kernel -- these were four small
// dependent operations per iteration.)
__global__ void k_slic_norm_bin(unsigned long long* __restrict__ sums, double* __restrict__ centers, int n, int batch,
                                int* __restrict__ next_cnt, int* __restrict__ next_list, int* __restrict__ next_overflow,
                                int* __restrict__ used_cnt, int n_used, int cell_px, int gx, int gy)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_used) used_cnt[t] = 0;               // counts and, right behind them, the overflow flags
    if (t >= n * batch) return;
    unsigned long long* s = sums + (size_t)t * 6;
    double* C = centers + (size_t)t * 5;
    if (s[5] == 0) {
        for (int q = 0; q < 5; ++q) C[q] = __builtin_bit_cast(double, kSlicDead);
        return;
    }
    const double cnt = (double)s[5];
    double c[5];
    for (int q = 0; q < 5; ++q) { c[q] = __ddiv_rn((double)s[q], cnt); C[q] = c[q]; }
    for (int q = 0; q < 6; ++q) s[q] = 0ull;
    slic_bin_one(c, t % n, t / n, next_cnt, next_list, next_overflow, cell_px, gx, gy);
}

}  // namespace dcmt
