// Developer probe (GPU box): what does a HIP graph buy for a chain of small dependent kernels on this ROCm, and do event-record
// nodes captured from a stream give usable timings?   hipcc --offload-arch=gfx950 -O2 graph_probe.hip -o graph_probe && ./graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_small(int n, const double* __restrict__ a, double* __restrict__ b) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i] * 1.0000001 + 1.0;
}
__global__ void k_big(size_t n, const double2* __restrict__ a, double2* __restrict__ b) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; v.x += 1; v.y += 1; b[i] = v; }
}
int main() {
    const int n = 1 << 14, chain = 10, reps = 200;
    double *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemset(a, 0, n * 8));
    const size_t nb = (size_t)1 << 24; double2 *A, *B; CK(hipMalloc(&A, nb * 16)); CK(hipMalloc(&B, nb * 16)); CK(hipMemset(A, 0, nb * 16));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto d) { return std::chrono::duration<double, std::micro>(d).count(); };
    // plain launches
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < reps; ++r)
            for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, st, n, (k & 1) ? b : a, (k & 1) ? a : b);
        auto t1 = now();
        CK(hipStreamSynchronize(st));
        auto t2 = now();
        if (w) printf("plain: enqueue %.2f us/kernel, total %.2f us/kernel\n", us(t1 - t0) / (reps * chain), us(t2 - t0) / (reps * chain));
    }
    // graph of the same chain, with two event-record nodes around kernel 3 (a bigger kernel)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < chain; ++k) {
        if (k == 3) { CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_big, dim3(2048), dim3(256), 0, st, nb, A, B); CK(hipEventRecord(e1, st)); }
        else hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, st, n, (k & 1) ? b : a, (k & 1) ? a : b);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        auto t1 = now();
        CK(hipStreamSynchronize(st));
        auto t2 = now();
        if (w) printf("graph (10 nodes, one of them 512 MB of traffic): enqueue %.2f us/graph, total %.2f us/graph\n", us(t1 - t0) / reps, us(t2 - t0) / reps);
    }
    float ms = -1; hipError_t e = hipEventElapsedTime(&ms, e0, e1);
    printf("event nodes inside the graph: %s, elapsed %.3f us (expected ~ %.0f us at 5 TB/s)\n", hipGetErrorString(e), ms * 1e3, nb * 32.0 / 5e12 * 1e6);
    // same chain without the big kernel and without events: pure small-kernel graph
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, st, n, (k & 1) ? b : a, (k & 1) ? a : b);
    CK(hipStreamEndCapture(st, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge2, st));
        auto t1 = now();
        CK(hipStreamSynchronize(st));
        auto t2 = now();
        if (w) printf("graph (10 small nodes): enqueue %.2f us/graph, total %.2f us/graph = %.2f us/kernel\n", us(t1 - t0) / reps, us(t2 - t0) / reps, us(t2 - t0) / reps / chain);
    }
    // graph launch followed by a host wait each time (the GMRES pattern: one read-back per iteration)
    {
        auto t0 = now();
        for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge2, st)); CK(hipStreamSynchronize(st)); }
        auto t2 = now();
        printf("graph + sync each: %.2f us/graph\n", us(t2 - t0) / reps);
        t0 = now();
        for (int r = 0; r < reps; ++r) { for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, st, n, (k & 1) ? b : a, (k & 1) ? a : b); CK(hipStreamSynchronize(st)); }
        t2 = now();
        printf("plain + sync each: %.2f us/chain\n", us(t2 - t0) / reps);
    }
    return 0;
}
