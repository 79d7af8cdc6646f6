// Tuning aid (not the product path): what does a DEPENDENT kernel of a decode chain cost before it moves a byte?
// Chains of N launches on one stream, every launch reading what its predecessor wrote, timed with HIP events
// (eager) -- microseconds per launch.  Variants isolate: the bare boundary, the wave launch of a full grid, one
// dependent global round trip behind the boundary, two of them, and the GEMV-like prologue (row -> LDS -> barrier).
//   hipcc --offload-arch=gfx950 -O3 tools/floor_lab.hip -o tools/floor_lab && tools/floor_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty() {}
// one dependent round trip: every thread reads one word of `in` (written by the previous launch), adds, writes `out`
__global__ void k_rt1(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n)
{
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
    out[i] = in[i] + 1;
}
// two dependent round trips: an index word first (the step state of the decoder), then the data
__global__ void k_rt2(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ st, uint32_t n)
{
    const uint32_t off = st[0];
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x + off) % n;
    out[i] = in[i] + 1;
}
// GEMV-like prologue: the whole 8 KB row into LDS, barrier, then lane 0 of every wave writes one word
__global__ void k_row(const uint4* __restrict__ in, uint32_t* __restrict__ out, uint32_t npk)
{
    extern __shared__ uint4 xs[];
    for (uint32_t p = threadIdx.x; p < npk; p += blockDim.x) xs[p] = in[p];
    __syncthreads();
    const uint4 v = xs[(threadIdx.x * 7) % npk];
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % (npk * 4)] = v.x + v.y + 1;
}

// the same with the register footprint of the real kernels (wave launch has to find 120 VGPRs per lane)
__global__ void __launch_bounds__(512) k_row_v120(const uint4* __restrict__ in, uint32_t* __restrict__ out, uint32_t npk)
{
    extern __shared__ uint4 xs[];
    asm volatile("v_mov_b32 v119, 0" ::: "v119");
    for (uint32_t p = threadIdx.x; p < npk; p += blockDim.x) xs[p] = in[p];
    __syncthreads();
    const uint4 v = xs[(threadIdx.x * 7) % npk];
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % (npk * 4)] = v.x + v.y + 1;
}
// ... and with per-wave time stamps: start of the first instruction, row staged, end (s_memrealtime, 100 MHz)
__global__ void __launch_bounds__(512) k_row_tl(const uint4* __restrict__ in, uint32_t* __restrict__ out, uint32_t npk, unsigned long long* tl)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ uint4 xs[];
    asm volatile("v_mov_b32 v119, 0" ::: "v119");
    for (uint32_t p = threadIdx.x; p < npk; p += blockDim.x) xs[p] = in[p];
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const uint4 v = xs[(threadIdx.x * 7) % npk];
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % (npk * 4)] = v.x + v.y + 1;
        const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        tl[w * 2] = t0; tl[w * 2 + 1] = t1;
    }
}

template <typename F>
static float chain(int n, F&& launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 64; i++) launch(i);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; i++) launch(i);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms * 1e3f / n < best ? ms * 1e3f / n : best;
    }
    return best;
}

int main()
{
    const uint32_t n = 4096; // words (16 KB): the size of a hidden row and its neighbours
    uint32_t *a, *b, *st;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&st, 64));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(st, 0, 64));
    const int N = 2000;
    auto rep = [&](const char* name, int wgs, int thr, float us) { printf("{\"kernel\": \"%s\", \"wgs\": %d, \"threads\": %d, \"us_per_launch\": %.2f}\n", name, wgs, thr, us); fflush(stdout); };
    const int grids[][2] = {{1, 64}, {16, 256}, {256, 256}, {256, 512}, {512, 256}, {1024, 256}};
    for (auto& g : grids) {
        const int wgs = g[0], thr = g[1];
        rep("empty", wgs, thr, chain(N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(thr), 0, 0); }));
        rep("rt1", wgs, thr, chain(N, [&](int i) { hipLaunchKernelGGL(k_rt1, dim3(wgs), dim3(thr), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n); }));
        rep("rt2", wgs, thr, chain(N, [&](int i) { hipLaunchKernelGGL(k_rt2, dim3(wgs), dim3(thr), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, st, n); }));
        rep("row8k", wgs, thr, chain(N, [&](int i) { hipLaunchKernelGGL(k_row, dim3(wgs), dim3(thr), 8192, 0, (const uint4*)((i & 1) ? b : a), (i & 1) ? a : b, 512u); }));
    }
    for (int lds : {8192, 13 * 1024, 41 * 1024}) {
        char nm[64]; snprintf(nm, sizeof nm, "row8k v120 lds%dk", lds / 1024);
        rep(nm, 256, 512, chain(N, [&](int i) { hipLaunchKernelGGL(k_row_v120, dim3(256), dim3(512), lds, 0, (const uint4*)((i & 1) ? b : a), (i & 1) ? a : b, 512u); }));
    }
    {
        unsigned long long* tl; CK(hipMalloc(&tl, 2048 * 16));
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_row_tl, dim3(256), dim3(512), 13 * 1024, 0, (const uint4*)((i & 1) ? b : a), (i & 1) ? a : b, 512u, tl);
        CK(hipDeviceSynchronize());
        static unsigned long long h[4096]; CK(hipMemcpy(h, tl, sizeof h, hipMemcpyDeviceToHost));
        unsigned long long mn = ~0ull; for (int w = 0; w < 2048; w++) mn = h[2 * w] < mn ? h[2 * w] : mn;
        double s50 = 0, smax = 0, g50 = 0; int c = 0; static double st[2048], sg[2048];
        for (int w = 0; w < 2048; w++) { st[w] = (h[2 * w] - mn) / 100.0; sg[w] = (h[2 * w + 1] - h[2 * w]) / 100.0; }
        auto med = [](double* a, int n, double q) { for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (a[j] < a[i]) { double t = a[i]; a[i] = a[j]; a[j] = t; } return a[(int)(q * (n - 1))]; };
        printf("{\"kernel\": \"row8k stamps\", \"start_p50\": %.2f, \"start_max\": %.2f, \"staged_minus_start_p50\": %.2f, \"p90\": %.2f}\n", med(st, 2048, 0.5), med(st, 2048, 1.0), med(sg, 2048, 0.5), med(sg, 2048, 0.9));
        (void)s50; (void)smax; (void)g50; (void)c;
    }
    // the same chains replayed from a graph (what mc_decoder_generate does)
    for (auto& g : grids) {
        const int wgs = g[0], thr = g[1];
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_row, dim3(wgs), dim3(thr), 8192, s, (const uint4*)((i & 1) ? b : a), (i & 1) ? a : b, 512u);
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms * 1e3f / 2000 < best ? ms * 1e3f / 2000 : best;
        }
        rep("row8k graph", wgs, thr, best);
    }
    return 0;
}
