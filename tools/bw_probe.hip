// bw_probe.hip -- what HBM bandwidth do the access patterns of the streaming kernels reach?
// Build: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o /tmp/bw_probe ; run on the GPU box.
//  A: the pattern of k_pre_s: one wave per 64-column strip (48 stored), full height, one dword per
//     lane per row, 8 rows of lookahead
//  B: same compute layout, rows brought in by global_load_lds_dwordx4 (4 rows x 64 columns per
//     wave instruction, wave-private LDS ring), then one ds_read_b32 per row
//  C: plain grid-stride float4 copy (the chip's practical ceiling)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ROWS = 352, COLS = 1216, VW = 48, HL = 8, STRIPS = (COLS + VW - 1) / VW;

__global__ __launch_bounds__(256) void k_a(const float* __restrict__ src, float* __restrict__ dst, int batch)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x, sgs = (STRIPS + 3) / 4;
    const int xcd = b & 7, slot = b >> 3;
    const int f = (slot / sgs) * 8 + xcd, sg = slot % sgs;
    const int strip = sg * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (strip >= STRIPS || f >= batch) return;
    const int gx = strip * VW - HL + lane;
    const int gxc = min(max(gx, 0), COLS - 1);
    const bool outlane = gx >= 0 && gx < COLS && lane >= HL && lane < HL + VW;
    const float* sp = src + (size_t)f * ROWS * COLS + gxc;
    float* op = dst + (size_t)f * ROWS * COLS + gxc;
    float PF[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) PF[q] = sp[(size_t)q * COLS];
    for (int i0 = 0; i0 < ROWS; i0 += 8) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int i = i0 + p;
            const float x = PF[p];
            PF[p] = sp[(size_t)min(i + 8, ROWS - 1) * COLS];
            if (outlane) op[(size_t)i * COLS] = x + 1.0f;
        }
    }
}

// LDS-DMA variant: ring of 32 rows x 64 columns per wave (8 KB), chunks of 4 rows per instruction
__global__ __launch_bounds__(256) void k_b(const float* __restrict__ src, float* __restrict__ dst, int batch)
{
    __shared__ __attribute__((aligned(16))) float ring[4][32 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x, sgs = (STRIPS + 3) / 4;
    const int xcd = b & 7, slot = b >> 3;
    const int f = (slot / sgs) * 8 + xcd, sg = slot % sgs;
    const int strip = sg * 4 + wave;
    if (strip >= STRIPS || f >= batch) return;
    const int gx0 = strip * VW - HL;
    const int gx = gx0 + lane;
    const bool outlane = gx >= 0 && gx < COLS && lane >= HL && lane < HL + VW;
    const float* fp = src + (size_t)f * ROWS * COLS;
    float* op = dst + (size_t)f * ROWS * COLS + min(max(gx, 0), COLS - 1);
    float* myring = ring[wave];
    // lane's source within a 4-row chunk: row (lane >> 4), columns gx0 + 4*(lane & 15) .. +3 (clamped group)
    const int lrow = lane >> 4;
    const int lcol = min(max(gx0 + 4 * (lane & 15), 0), COLS - 4);
    auto issue = [&](int chunk) {     // rows 4*chunk .. 4*chunk+3 -> ring slot (chunk & 7)
        const int r = min(4 * chunk + lrow, ROWS - 1);
        const float* g = fp + (size_t)r * COLS + lcol;
        float* l = myring + (chunk & 7) * 256;    // 4 rows x 64 floats; lane-linear: + lane*4 floats
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    };
    constexpr int NCH = ROWS / 4;     // 88 chunks
#pragma unroll
    for (int c = 0; c < 4; ++c) issue(c);
    for (int c = 0; c < NCH; ++c) {
        if (c + 4 < NCH) { issue(c + 4); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float* l = myring + (c & 7) * 256;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float x = l[rr * 64 + lane];
            if (outlane) op[(size_t)(4 * c + rr) * COLS] = x + 1.0f;
        }
    }
}

__global__ void k_c(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = src[i];
        v.x += 1.0f; v.y += 1.0f; v.z += 1.0f; v.w += 1.0f;
        dst[i] = v;
    }
}


typedef float f4v __attribute__((ext_vector_type(4)));
// D: every thread moves UNR independent 16-byte pieces (all loads issued before the first store), one pass, no grid-stride loop
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_d(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4)
{
    const size_t base = (size_t)blockIdx.x * (256 * UNR) + threadIdx.x;
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n4) { if (NT) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src + i)); v[u] = make_float4(t.x, t.y, t.z, t.w); } else v[u] = src[i]; }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n4) { v[u].x += 1.0f; if (NT) { f4v t = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(dst + i)); } else dst[i] = v[u]; }
    }
}
// R: read only (sum kept so the loads stay), W: write only
template <int UNR>
__global__ __launch_bounds__(256) void k_r(const float4* __restrict__ src, float* __restrict__ sink, size_t n4)
{
    const size_t base = (size_t)blockIdx.x * (256 * UNR) + threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n4) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int UNR>
__global__ __launch_bounds__(256) void k_w(float4* __restrict__ dst, size_t n4)
{
    const size_t base = (size_t)blockIdx.x * (256 * UNR) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n4) dst[i] = make_float4(1.f, 2.f, 3.f, (float)u);
    }
}

int main()
{
    const int batch = 1024;
    const size_t n = (size_t)batch * ROWS * COLS;
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = ((STRIPS + 3) / 4) * batch;
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s %.3f ms  %.0f GB/s algorithmic (read+write %zu MB)\n", name, ms, 2.0 * n * 4 / ms / 1e6, 2 * n * 4 >> 20);
    };
    time("A dword strips", [&] { hipLaunchKernelGGL(k_a, dim3(grid), dim3(256), 0, 0, a, b, batch); });
    time("B lds-dma x4 strips", [&] { hipLaunchKernelGGL(k_b, dim3(grid), dim3(256), 0, 0, a, b, batch); });
    time("C float4 copy", [&] { hipLaunchKernelGGL(k_c, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); });

    const size_t n4 = n / 4;
    auto gridfor = [&](int unr) { return dim3((unsigned)((n4 + 256 * unr - 1) / (256 * unr))); };
    time("D float4 copy unr4", [&] { hipLaunchKernelGGL((k_d<4, false>), gridfor(4), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
    time("D float4 copy unr8", [&] { hipLaunchKernelGGL((k_d<8, false>), gridfor(8), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
    time("D float4 copy unr8 nt", [&] { hipLaunchKernelGGL((k_d<8, true>), gridfor(8), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
    time("D float4 copy unr16", [&] { hipLaunchKernelGGL((k_d<16, false>), gridfor(16), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
    time("hipMemcpyAsync D2D", [&] { (void)hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, 0); });
    time("R read only unr8 (x2 = bytes)", [&] { hipLaunchKernelGGL((k_r<8>), gridfor(8), dim3(256), 0, 0, (const float4*)a, b, n4); });
    time("W write only unr8 (x2 = bytes)", [&] { hipLaunchKernelGGL((k_w<8>), gridfor(8), dim3(256), 0, 0, (float4*)b, n4); });
    // correctness of B vs A
    std::vector<float> ha(ROWS * COLS), hb(ROWS * COLS);
    hipMemset(b, 0, n * 4);
    hipLaunchKernelGGL(k_b, dim3(grid), dim3(256), 0, 0, a, b, batch);
    hipMemcpy(hb.data(), b + (size_t)5 * ROWS * COLS, ROWS * COLS * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (auto v : hb) bad += v != 1.0f;
    printf("B mismatches in frame 5: %zu\n", bad);
    return 0;
}
