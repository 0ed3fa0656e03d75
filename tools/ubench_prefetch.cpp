// Does K/V touched by an EARLIER launch (on the XCD that will consume it) make the decode attention faster?
// prefetch kernel: one workgroup per (clip, head), block id = b*H + h (the attention kernel's (h, b) grid has the same linear
// id, so under round-robin block->XCD placement both land on the same XCD); plain 16-byte loads of the first `frac` of the
// head's K and V rows, values discarded.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench_prefetch.cpp video2music_amd/csrc/attn_decode.hip video2music_amd/csrc/tuning.hip -o tools/ubench_prefetch.bin
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>
#include "../video2music_amd/csrc/kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
void amt_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }

__global__ __launch_bounds__(256) void k_prefetch(const float* __restrict__ k, const float* __restrict__ v, size_t head_stride, int n4, float* sink) {
    // n4 float4 per thread from K and from V of this (clip, head)
    const float4* kp = reinterpret_cast<const float4*>(k + (size_t)blockIdx.x * head_stride) + threadIdx.x;
    const float4* vp = reinterpret_cast<const float4*>(v + (size_t)blockIdx.x * head_stride) + threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < n4; ++i) {
        const float4 a = kp[(size_t)i * 256], b = vp[(size_t)i * 256];
        acc += a.x + b.x;
    }
    if (acc == 123456.f) sink[0] = acc;
}

static double bench(const char* name, std::function<void(hipStream_t, int)> body, int reps) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    body(s, 0); CK(hipStreamSynchronize(s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) body(s, i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-72s %8.3f us per iteration\n", name, ms * 1e3 / reps);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
    return ms * 1e3 / reps;
}

int main() {
    const int B = 32, H = 8, hd = 64, S = 300, nl = 6, d = 512;
    auto falloc = [](size_t n) { float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; };
    float *q = falloc(B * d), *o = falloc(B * d), *sink = falloc(64);
    std::vector<float*> kx(nl), vx(nl);
    for (int l = 0; l < nl; ++l) { kx[l] = falloc((size_t)B * H * S * hd); vx[l] = falloc((size_t)B * H * S * hd); }
    const size_t head_stride = (size_t)S * hd;            // floats per (clip, head)
    auto attn = [&](hipStream_t s, int l) {
        AttnDecodeParams a{};
        a.q = q; a.k = kx[l]; a.v = vx[l]; a.o = o; a.B = B; a.H = H; a.hd = hd; a.cap = S; a.n_keys = S;
        amt_launch_attn_decode(a, s);
    };
    printf("AMT_NT=%s (bit 1 = cross-attention K/V loads non-temporal)\n", getenv("AMT_NT") ? getenv("AMT_NT") : "(default 3)");
    bench("attention alone (cross, 39.3 MB, 6 layers' K/V cycled)", [&](hipStream_t s, int i) { attn(s, i % nl); }, 120);
    for (int pct : {25, 50, 75, 100}) {
        const int n4 = (int)(head_stride / 4 * pct / 100 / 256);      // float4 per thread for that share of the rows
        char name[160];
        snprintf(name, 160, "prefetch %3d%% (%.1f MB) as its own launch, then attention", pct, 2.0 * B * H * n4 * 256 * 16 / 1e6);
        bench(name, [&](hipStream_t s, int i) {
            hipLaunchKernelGGL(k_prefetch, dim3(B * H), dim3(256), 0, s, kx[i % nl], vx[i % nl], head_stride, n4, sink);
            attn(s, i % nl);
        }, 120);
        snprintf(name, 160, "  (that prefetch launch alone)");
        bench(name, [&](hipStream_t s, int i) {
            hipLaunchKernelGGL(k_prefetch, dim3(B * H), dim3(256), 0, s, kx[i % nl], vx[i % nl], head_stride, n4, sink);
        }, 120);
    }
    return 0;
}
