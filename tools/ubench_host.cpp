// Host-side cost of hipGraphLaunch per kernel node, and its scaling over two host threads / two streams.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Big { float* p; int pad[60]; };
__global__ void k_small(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void k_big(Big b) { if (threadIdx.x == 0) b.p[blockIdx.x] += 1.f; }
__global__ void k_busy(float* p, int iters) { float a = p[blockIdx.x]; for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f; if (a == 123.f) p[0] = a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct G { hipStream_t s; hipGraph_t g; hipGraphExec_t ge; };
G make(int nodes, int kind, float* p, int busy_iters) {
    G r; CK(hipStreamCreateWithFlags(&r.s, hipStreamNonBlocking));
    CK(hipStreamBeginCapture(r.s, hipStreamCaptureModeThreadLocal));
    Big b{}; b.p = p;
    for (int i = 0; i < nodes; ++i) {
        if (kind == 0) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, r.s, p);
        else if (kind == 1) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, r.s, b);
        else hipLaunchKernelGGL(k_busy, dim3(128), dim3(512), 0, r.s, p, busy_iters);
    }
    CK(hipStreamEndCapture(r.s, &r.g)); CK(hipGraphInstantiate(&r.ge, r.g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(r.ge, r.s)); CK(hipStreamSynchronize(r.s));
    return r;
}
int main() {
    float *p, *q; CK(hipMalloc(&p, 1 << 20)); CK(hipMalloc(&q, 1 << 20)); CK(hipMemset(p, 0, 1 << 20)); CK(hipMemset(q, 0, 1 << 20));
    const int nodes = 392, reps = 20;
    for (int kind = 0; kind < 2; ++kind) {
        G a = make(nodes, kind, p, 0);
        double t0 = now();
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(a.ge, a.s));
        double t1 = now(); CK(hipStreamSynchronize(a.s)); double t2 = now();
        printf("kind=%d 1 thread : host %.2f us/node, total %.2f us/node\n", kind, (t1 - t0) * 1e6 / (nodes * reps), (t2 - t0) * 1e6 / (nodes * reps));
    }
    // two streams, one thread vs two threads; kernels busy ~5 us each so the GPU side can overlap
    for (int iters : {0, 2000}) {
        G a = make(nodes, 2, p, iters), b = make(nodes, 2, q, iters);
        double t0 = now();
        for (int i = 0; i < reps; ++i) { CK(hipGraphLaunch(a.ge, a.s)); }
        CK(hipStreamSynchronize(a.s)); double t1 = now();
        printf("busy iters=%d: single stream %.2f us/node total\n", iters, (t1 - t0) * 1e6 / (nodes * reps));
        t0 = now();
        for (int i = 0; i < reps; ++i) { CK(hipGraphLaunch(a.ge, a.s)); CK(hipGraphLaunch(b.ge, b.s)); }
        CK(hipStreamSynchronize(a.s)); CK(hipStreamSynchronize(b.s)); t1 = now();
        printf("busy iters=%d: two streams, one thread: %.2f us per node-pair total\n", iters, (t1 - t0) * 1e6 / (nodes * reps));
        t0 = now();
        std::thread ta([&] { for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(a.ge, a.s)); CK(hipStreamSynchronize(a.s)); });
        std::thread tb([&] { for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(b.ge, b.s)); CK(hipStreamSynchronize(b.s)); });
        ta.join(); tb.join(); t1 = now();
        printf("busy iters=%d: two streams, two threads: %.2f us per node-pair total\n", iters, (t1 - t0) * 1e6 / (nodes * reps));
    }
    return 0;
}
