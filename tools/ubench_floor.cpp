// Kernel-floor micro-benchmark (gfx950): cost per kernel of a dependent chain replayed from a hipGraph.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_floor.cpp -o gpurun_out/ubench_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_empty(float* p) {}
__global__ void k_store(float* p) { if (threadIdx.x == 0) p[blockIdx.x] = 1.f; }
__global__ void k_load_store(float* p, const float* q) { if (threadIdx.x == 0) p[blockIdx.x] = q[blockIdx.x] + 1.f; }
__global__ void k_chain2(float* p, const int* idx, const float* q) {   // two dependent loads
    if (threadIdx.x == 0) { int i = idx[0]; p[blockIdx.x] = q[i + blockIdx.x] + 1.f; }
}
__global__ void k_lds(float* p) { extern __shared__ float s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (threadIdx.x == 0) p[blockIdx.x] = s[5]; }
__global__ void k_stream(float* p, const float4* q, int n4) {   // each block reads n4 float4 per thread
    float4 a = {0, 0, 0, 0};
    const float4* b = q + (size_t)blockIdx.x * blockDim.x * n4 + threadIdx.x;
    for (int i = 0; i < n4; ++i) { float4 v = b[(size_t)i * blockDim.x]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (a.x + a.y + a.z + a.w == 12345.f) p[0] = 1.f;
}

template <typename F>
double bench(const char* name, F launch, int reps = 2000) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) launch(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %7.3f us/kernel\n", name, ms * 1e3 / reps);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
    return ms * 1e3 / reps;
}

int main() {
    float *p, *q; int* idx;
    size_t big = (size_t)1 << 30;
    CK(hipMalloc(&p, 1 << 20)); CK(hipMalloc(&q, big)); CK(hipMalloc(&idx, 64));
    CK(hipMemset(idx, 0, 64)); CK(hipMemset(q, 0, big));
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int wg : {32, 96, 128, 192, 256}) for (int th : {256, 512, 1024}) {
        char n[128];
        snprintf(n, 128, "empty <<<%d,%d>>>", wg, th); bench(n, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(th), 0, s, p); });
        snprintf(n, 128, "store <<<%d,%d>>>", wg, th); bench(n, [&](hipStream_t s) { hipLaunchKernelGGL(k_store, dim3(wg), dim3(th), 0, s, p); });
        snprintf(n, 128, "load+store <<<%d,%d>>>", wg, th); bench(n, [&](hipStream_t s) { hipLaunchKernelGGL(k_load_store, dim3(wg), dim3(th), 0, s, p, q); });
        snprintf(n, 128, "2 dependent loads+store <<<%d,%d>>>", wg, th); bench(n, [&](hipStream_t s) { hipLaunchKernelGGL(k_chain2, dim3(wg), dim3(th), 0, s, p, idx, q); });
        snprintf(n, 128, "lds 82KB+sync+store <<<%d,%d>>>", wg, th); bench(n, [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(wg), dim3(th), 82 * 1024, s, p); });
    }
    // streaming: 256 blocks x 512 threads, n4 float4 per thread -> bytes = 256*512*16*n4
    for (int n4 : {1, 4, 16, 64, 256}) {
        char n[128]; snprintf(n, 128, "stream %.1f MB <<<256,512>>>", 256.0 * 512 * 16 * n4 / 1e6);
        double us = bench(n, [&](hipStream_t s) { hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), 0, s, p, (const float4*)q, n4); }, 500);
        printf("    -> %.0f GB/s\n", 256.0 * 512 * 16 * n4 / us / 1e3);
    }
    return 0;
}
