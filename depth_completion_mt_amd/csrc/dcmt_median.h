// dcmt_median.h -- the exact 5x5 median as time-shared sorting networks (tools/gen_median_shared.py has the scheme and its
// verification, tools/gen_median_3in.py the three-input rewriting of the networks and its proof): every row's five
// horizontal neighbours are sorted once (sort5), every second row a pair of sorted rows is merged (merge55) and the six
// middle order statistics of the 4-row core are extracted (mid20); each window's median is then the 6th smallest of those
// six and the sorted fifth row (final6).  Shared by the streaming kernels (neighbours by DPP) and the staged tile kernel
// (neighbours from LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "median_shared_nets.h"
#include "median_shared_nets3.h"

namespace dcmt {

__device__ __forceinline__ float fmax2(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float fmin2(float a, float b) { return __builtin_fminf(a, b); }

// compile-time loop: f(std::integral_constant<int, P>) for P in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ float fmax3(float a, float b, float c) { return fmax2(fmax2(a, b), c); }   // -> v_max3_f32
__device__ __forceinline__ float fmin3(float a, float b, float c) { return fmin2(fmin2(a, b), c); }   // -> v_min3_f32

#define DCMT_CX(a, b)   { const float lo_ = fmin2(v[a], v[b]); v[b] = fmax2(v[a], v[b]); v[a] = lo_; }
#define DCMT_CMIN(a, b) { v[a] = fmin2(v[a], v[b]); }
#define DCMT_CMAX(a, b) { v[b] = fmax2(v[a], v[b]); }
// sort5 in 12 three-input instructions instead of a 9-exchange network's 18 (min/max/med3 all issue
// at the same rate): sort a triple (min3, med3, max3) and a pair, then merge them in closed form.
// Verified against sorted() on all 5^5 value patterns (ties included).
__device__ __forceinline__ void sort5(float (&v)[5])
{
    const float x0 = fmin3(v[0], v[1], v[2]), x1 = __builtin_amdgcn_fmed3f(v[0], v[1], v[2]), x2 = fmax3(v[0], v[1], v[2]);
    const float y0 = fmin2(v[3], v[4]), y1 = fmax2(v[3], v[4]);
    const float A = fmax2(x0, y0), B = fmin2(x2, y1);
    v[0] = fmin2(x0, y0);
    v[1] = fmin3(A, x1, y1);
    v[2] = __builtin_amdgcn_fmed3f(A, x1, B);
    v[3] = fmax3(B, x1, y0);
    v[4] = fmax2(x2, y1);
}
// The two merge networks in their three-input form (tools/gen_median_3in.py: the exchange networks of
// median_shared_nets.h rewritten with min3 / max3 / med3 using the order knowledge of their sorted inputs, each
// rewrite proven by the 0/1 principle on sorted inputs): 19 instead of 26 and 24 instead of 36 instructions.
// P = merge of two sorted 5-lists
__device__ __forceinline__ void merge55(const float (&a)[5], const float (&b)[5], float (&P)[10])
{
#define DCMT_IN(k) ((k) < 5 ? a[(k) < 5 ? (k) : 0] : b[(k) >= 5 ? (k) - 5 : 0])
#define DCMT_OUT(k) P[k]
    DCMT_MERGE55_3IN(DCMT_IN, DCMT_OUT)
#undef DCMT_IN
#undef DCMT_OUT
}
// C = ranks 8..13 (1-based, ascending) of the union of two sorted 10-lists
__device__ __forceinline__ void mid20(const float (&pa)[10], const float (&pb)[10], float (&C)[6])
{
#define DCMT_IN(k) ((k) < 10 ? pa[(k) < 10 ? (k) : 0] : pb[(k) >= 10 ? (k) - 10 : 0])
#define DCMT_OUT(k) C[k]
    DCMT_MID20_3IN(DCMT_IN, DCMT_OUT)
#undef DCMT_IN
#undef DCMT_OUT
}
#undef DCMT_CX
#undef DCMT_CMIN
#undef DCMT_CMAX
// 6th smallest of sorted C (6) u sorted a (5) = the median of the 25-window:
//     min(C5, max(a0,C4), max(a1,C3), max(a2,C2), max(a3,C1), max(a4,C0))
// folded into five med3: with r >= min(a_i, C_{4-i}) -- which the sortedness of C and a guarantees at every step --
// min(r, max(a_i, C_{4-i})) = med3(a_i, C_{4-i}, r).  Checked on all sorted 0/1 inputs (min / max / med3 commute with
// monotone maps, so that proves it for all inputs) and on random floats with ties (tools/gen_median_3in.py).
__device__ __forceinline__ float final6(const float (&C)[6], const float (&a)[5])
{
    float r = C[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) r = __builtin_amdgcn_fmed3f(a[i], C[4 - i], r);
    return r;
}


// The vertical half of the scheme as a unit: feed the sorted rows of one column top to bottom, one per step; step PP
// (= step number mod 8, compile-time) returns the median of the 5x5 window whose bottom row is the one just fed
// (meaningful from the fifth row on).
struct MedianColumn {
    float SE[4][5];          // sorted even rows, slot (step / 2) & 3
    float SO[5];             // the latest sorted odd row
    float P[2][10];          // merged pairs (rows 2q-1, 2q), slot q & 1
    float C[6];              // middle order statistics of the current 4-row core
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 5; ++k) SE[q][k] = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) SO[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) { P[0][k] = 0.f; P[1][k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = 0.f;
    }
    template <int PP>
    __device__ __forceinline__ float step(const float (&s)[5])
    {
        float m;
        if constexpr ((PP & 1) == 0) {
            constexpr int qs = (PP >> 1) & 3;
            merge55(SO, s, P[(PP >> 1) & 1]);
            mid20(P[((PP >> 1) + 1) & 1], P[(PP >> 1) & 1], C);
            m = final6(C, SE[(qs + 2) & 3]);
#pragma unroll
            for (int k = 0; k < 5; ++k) SE[qs][k] = s[k];
        } else {
            m = final6(C, s);
#pragma unroll
            for (int k = 0; k < 5; ++k) SO[k] = s[k];
        }
        return m;
    }
};

}  // namespace dcmt
