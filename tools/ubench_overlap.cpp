// Experiment (gfx950): can a K/V-prefetching kernel run BESIDE the latency-bound GEMM chain of a decode step?
//   stream kernel S : 256 workgroups x 512 threads, each loads `KB` KiB into registers, then waits (bounded spin) for a
//                     generation counter that the chain publishes, then reduces its registers and writes one float.
//   chain kernel  G : `wg` workgroups x 1024 threads with 66 KiB of LDS (the skinny GEMM's footprint), busy for `us`
//                     microseconds (s_memrealtime), the last one of a chain link arrives on the counter.
// Variants: S alone, chain alone, both in one captured graph (fork / join through events), both on two plain streams.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_overlap.cpp -o tools/ubench_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

template <int NV>   // NV float4 per thread
__global__ __launch_bounds__(512) void k_stream_wait(const float4* __restrict__ src, size_t wg_stride4, const unsigned* flag, unsigned want,
                                                     float* out, unsigned* timeouts, int wait) {
    const float4* p = src + (size_t)blockIdx.x * wg_stride4 + threadIdx.x;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p + (size_t)i * 512));
        v[i] = make_float4(t.x, t.y, t.z, t.w);
    }
    if (wait) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = now();
            bool ok = false;
            while (now() - t0 < 20000ull) {        // 200 us bound
                if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (!ok) atomicAdd(timeouts, 1u);
        }
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    if (s == 123456.789f) out[blockIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = 1.f;
}

__global__ __launch_bounds__(1024) void k_chain(float* p, int ticks, unsigned* flag, int arrive) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = now();
    while (now() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) {
        p[blockIdx.x] = sm[5];
        if (arrive) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const int reps = 200;
    float *out, *pc; unsigned *flag, *timeouts; float4* src;
    const size_t big = (size_t)1 << 30;
    CK(hipMalloc(&src, big)); CK(hipMemset(src, 0, big));
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&pc, 1 << 16)); CK(hipMalloc(&flag, 64)); CK(hipMalloc(&timeouts, 64));
    CK(hipMemset(timeouts, 0, 64));
    CK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));

    const int chain_wg = 192, chain_len = 3, ticks = 300;      // 3 links x 3 us busy
    constexpr int NV = 19;                                        // 19 float4 x 512 threads = 152 KiB per workgroup
    const size_t wg_stride4 = (size_t)NV * 512;                  // contiguous per workgroup
    // cycle through the 1 GiB buffer so that nothing is re-read from the Infinity Cache
    const size_t per_launch4 = wg_stride4 * 256;
    const size_t n_slots = big / 16 / per_launch4;

    auto launch_S = [&](hipStream_t s, int it, unsigned want, int wait) {
        const float4* base = src + (size_t)(it % n_slots) * per_launch4;
        hipLaunchKernelGGL(k_stream_wait<NV>, dim3(256), dim3(512), 0, s, base, wg_stride4, flag, want, out, timeouts, wait);
    };
    auto launch_chain = [&](hipStream_t s, int arrive_last) {
        for (int l = 0; l < chain_len; ++l)
            hipLaunchKernelGGL(k_chain, dim3(chain_wg), dim3(1024), 66 * 1024, s, pc, ticks, flag, (arrive_last && l == chain_len - 1) ? 1 : 0);
    };
    auto time_graph = [&](const char* name, auto body) {
        CK(hipMemset(flag, 0, 64)); CK(hipDeviceSynchronize());
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < reps; ++i) body(i);
        CK(hipStreamEndCapture(sa, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipEventRecord(e0, sa)); CK(hipGraphLaunch(ge, sa)); CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned to; CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
        printf("%-70s %8.2f us/iteration   (timeouts so far %u)\n", name, ms * 1e3 / reps, to);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };

    printf("S = 256 WG x 512 thr x %d KiB (%.1f MB / launch); chain = %d links of %d WG x 1024 thr busy %.1f us\n", NV * 512 * 16 / 1024,
           per_launch4 * 16 / 1e6, chain_len, chain_wg, ticks / 100.0);
    time_graph("S alone (no wait)", [&](int i) { launch_S(sa, i, 0, 0); });
    time_graph("chain alone", [&](int i) { launch_chain(sa, 0); });
    time_graph("serial: chain then S (no wait)", [&](int i) { launch_chain(sa, 0); launch_S(sa, i, 0, 0); });
    // fork/join inside the captured graph: S on branch B (launched at the fork, waits for the chain's last link), chain on branch A
    time_graph("graph fork: S(no wait) || chain, join", [&](int i) {
        CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0));
        launch_S(sb, i, 0, 0);
        launch_chain(sa, 0);
        CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0));
    });
    time_graph("graph fork: S(waits for chain's counter) || chain, join", [&](int i) {
        CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0));
        launch_S(sb, i, (unsigned)(i + 1) * chain_wg, 1);
        launch_chain(sa, 1);
        CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0));
    });
    // the same with the chain launched FIRST in capture order (S enqueued after the chain's kernels)
    time_graph("graph fork, chain captured first: S(waits) || chain, join", [&](int i) {
        CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0));
        launch_chain(sa, 1);
        launch_S(sb, i, (unsigned)(i + 1) * chain_wg, 1);
        CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0));
    });
    // plain streams, no graph
    {
        CK(hipMemset(flag, 0, 64)); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sa));
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0));
            launch_S(sb, i, (unsigned)(i + 1) * chain_wg, 1);
            launch_chain(sa, 1);
            CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0));
        }
        CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned to; CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
        printf("%-70s %8.2f us/iteration   (timeouts so far %u)\n", "two plain streams: S(waits) || chain", ms * 1e3 / reps, to);
    }
    return 0;
}
