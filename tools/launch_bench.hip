// Launch-interval microbenchmark (tuning aid): dependent trivial kernels, several launch paths.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
extern "C" __global__ void triv(int* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += n; }
extern "C" __global__ void triv_wide(int* p, int n) { if (threadIdx.x == 0) p[blockIdx.x] += n; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 2000;
    for (int wide = 0; wide < 2; wide++) {
        // eager, triple-chevron
        for (int rep = 0; rep < 2; rep++) {
            CK(hipStreamSynchronize(s));
            double t0 = now();
            for (int i = 0; i < N; i++) {
                if (wide) hipLaunchKernelGGL(triv_wide, dim3(512), dim3(256), 0, s, d, 1);
                else hipLaunchKernelGGL(triv, dim3(1), dim3(64), 0, s, d, 1);
            }
            double t1 = now();
            CK(hipStreamSynchronize(s));
            double t2 = now();
            if (rep) printf("wide=%d eager GGL: submit %.2f us/launch, total %.2f us/launch\n", wide, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
        }
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; i++) {
            if (wide) hipLaunchKernelGGL(triv_wide, dim3(512), dim3(256), 0, s, d, 1);
            else hipLaunchKernelGGL(triv, dim3(1), dim3(64), 0, s, d, 1);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        printf("wide=%d graph(200 nodes): %.2f us/node\n", wide, (t2 - t0) / 2000 * 1e6);
    }
    // events around a chain to get GPU-side time
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(triv, dim3(1), dim3(64), 0, s, d, 1);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("event-timed eager chain: %.2f us/launch\n", ms * 1e3 / N);
    return 0;
}
