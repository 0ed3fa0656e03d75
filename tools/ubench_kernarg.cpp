// Does kernel-argument preloading (gfx940+: the CP writes the first kernarg dwords into SGPRs before the wave starts) shorten a
// dependent launch chain?  Two kernels with the same body -- every workgroup reads words the previous launch wrote and stores its
// own -- one with flat pointer arguments (preloadable), one with the arguments in a by-value struct (as libamt_hip's kernels take
// them: not preloadable), replayed from a hipGraph as a 2000-long dependent chain.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/ubench_kernarg.cpp -o tools/ubench_kernarg.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct P { const float* in; float* out; const float* w; const float* w2; int n; float s; };
__device__ __forceinline__ void body(const float* in, float* out, const float* w, const float* w2, int n, float s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = in[(i * 7) % n] * s + w[i % n] + w2[threadIdx.x];
    v += __shfl_xor(v, 1, 64);
    out[i % n] = v;
}
__global__ __launch_bounds__(1024) void k_flat(const float* in, float* out, const float* w, const float* w2, int n, float s) { body(in, out, w, w2, n, s); }
__global__ __launch_bounds__(1024) void k_struct(P p) { body(p.in, p.out, p.w, p.w2, p.n, p.s); }
template <typename F> double bench(const char* name, F launch, int reps = 2000) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) launch(s, i);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    printf("%-40s %7.3f us/kernel\n", name, best * 1e3 / reps);
    return best;
}
int main() {
    const int n = 256 * 1024;
    float *a, *b, *w, *w2; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&w2, 4096));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(w, 0, n * 4)); CK(hipMemset(w2, 0, 4096));
    for (int wg : {64, 128, 256}) {
        char nm[64];
        snprintf(nm, 64, "flat args (preload) <<<%d,1024>>>", wg);
        bench(nm, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_flat, dim3(wg), dim3(1024), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, w, w2, n, 0.5f); });
        snprintf(nm, 64, "struct arg          <<<%d,1024>>>", wg);
        bench(nm, [&](hipStream_t s, int i) { P p{(i & 1) ? b : a, (i & 1) ? a : b, w, w2, n, 0.5f}; hipLaunchKernelGGL(k_struct, dim3(wg), dim3(1024), 0, s, p); });
    }
    return 0;
}
