// What does a chip-wide hand-off cost inside ONE resident kernel, against the kernel boundary of a captured graph?
// 256 workgroups (one per CU) run N phases; in every phase a workgroup reads rows that ALL workgroups wrote in the
// previous phase (the decode step's skinny-GEMM pattern: 16 rows x K floats per workgroup from a B x K activation
// block), stores its own 256 outputs, and meets the others at a grid barrier.  Variants of the barrier:
//   flat   : one counter, every workgroup's thread 0 adds (release, agent scope) and spins on an acquire load
//   xcd    : per-XCD counters (XCC_ID), the last arriver of an XCD adds to the global one; everybody spins on the global
//   sload  : as flat, but the spin uses a scalar load (s_load_dword glc) so that it never queues behind vector loads
// and the same phase body as N dependent launches replayed from a hipGraph.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_gridbar.cpp -o tools/ubench_gridbar.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NWG = 256, NT = 1024, ROWS = 32;
constexpr unsigned SPIN_MAX = 1u << 22;          // bounded: a missing workgroup ends the kernel with an error flag, not a hang

struct Bar { unsigned* cnt; unsigned* xcd_cnt; int* err; };

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// MODE: 0 acquire-load spin on the one counter (the naive form); 1 relaxed spin + s_sleep back-off, one acquire fence at the end;
// 2 as 1, but the LAST arriver publishes a per-XCD flag word (8 words in 8 different 4 KiB pages) and everybody polls its own XCD's;
// 3 as 1 with a scalar-load spin
template <int MODE, int SLEEP>
__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned target = (phase + 1) * NWG;
        unsigned n = 0;
        if (MODE == 0) {
            __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(b.cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++n < SPIN_MAX) { }
        } else if (MODE == 1) {
            __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(b.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++n < SPIN_MAX) __builtin_amdgcn_s_sleep(SLEEP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else if (MODE == 2) {
            const unsigned old = __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == target) {
#pragma unroll
                for (int x = 0; x < 8; ++x) __hip_atomic_store(b.xcd_cnt + x * 1024, phase + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                unsigned* f = b.xcd_cnt + xcc_id() * 1024;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase + 1 && ++n < SPIN_MAX) __builtin_amdgcn_s_sleep(SLEEP);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned v;
            do {
                asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(b.cnt) : "memory");
                if (v >= target) break;
                __builtin_amdgcn_s_sleep(SLEEP);
            } while (++n < SPIN_MAX);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (n >= SPIN_MAX) *b.err = 1;
    }
    __syncthreads();
}

// phase body: workgroup w reads rows [16*(w&1), +16) x K floats of `in` (written by everybody), adds its own weights word,
// writes 16 rows x 16 columns of `out` (column tile w>>1)
template <int KCH>
__device__ __forceinline__ void body(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ wts, int w) {
    constexpr int K = KCH * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = in + (size_t)(16 * (w & 1) + wave) * K;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(xr + (c * 64 + lane) * 4);
        const float4 q = *reinterpret_cast<const float4*>(wts + ((size_t)w * NT * KCH + (size_t)c * NT + tid) * 4);
        s += v.x * q.x + v.y * q.y + v.z * q.z + v.w * q.w;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
    // 16 columns per workgroup tile: the N = 128 tiles x 16 columns = 2048 >= K columns of the next phase's rows
    if (lane < 16) {
        const int col = (w >> 1) * 16 + lane;
        if (col < K) out[(size_t)(16 * (w & 1) + wave) * K + col] = s * 1e-3f + 1.f;
    }
}

template <int MODE, int SLEEP, int KCH>
__global__ __launch_bounds__(NT) void persistent(float* a, float* bbuf, const float* wts, Bar bar, int phases, unsigned long long* stamps) {
    const int w = blockIdx.x;
    for (int p = 0; p < phases; ++p) {
        if (KCH > 0) body<(KCH > 0 ? KCH : 1)>((p & 1) ? bbuf : a, (p & 1) ? a : bbuf, wts, w);
        unsigned long long t0 = 0;
        if (stamps && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
        grid_barrier<MODE, SLEEP>(bar, (unsigned)p);
        if (stamps && threadIdx.x == 0 && p == phases - 2) { stamps[2 * w] = t0; stamps[2 * w + 1] = __builtin_amdgcn_s_memrealtime(); }
    }
}

template <int KCH>
__global__ __launch_bounds__(NT) void one_phase(const float* in, float* out, const float* wts) { body<KCH>(in, out, wts, blockIdx.x); }

template <int MODE, int SLEEP, int KCH>
void run_persistent(const char* name, float* a, float* b, float* wts, Bar bar, int phases) {
    unsigned long long* st; CK(hipMalloc(&st, NWG * 16));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(bar.cnt, 0, 4, s)); CK(hipMemsetAsync(bar.xcd_cnt, 0, 8 * 4096, s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL((persistent<MODE, SLEEP, KCH>), dim3(NWG), dim3(NT), 0, s, a, b, wts, bar, phases, st);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    int err; CK(hipMemcpy(&err, bar.err, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> h(2 * NWG); CK(hipMemcpy(h.data(), st, NWG * 16, hipMemcpyDeviceToHost));
    unsigned long long last_arr = 0, first_rel = ~0ull, last_rel = 0;
    for (int w = 0; w < NWG; ++w) { last_arr = std::max(last_arr, h[2 * w]); first_rel = std::min(first_rel, h[2 * w + 1]); last_rel = std::max(last_rel, h[2 * w + 1]); }
    printf("%-34s K=%4d  %7.3f us/phase   barrier: last arrival -> first / last release %5.2f / %5.2f us%s\n", name, (KCH > 0 ? KCH : 0) * 256, best * 1e3 / phases,
           (double)((long long)(first_rel - last_arr)) * 0.01, (double)((long long)(last_rel - last_arr)) * 0.01, err ? "   SPIN LIMIT HIT" : "");
    CK(hipFree(st)); CK(hipStreamDestroy(s));
}

template <int KCH>
void run_graph(float* a, float* b, float* wts, int phases) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL((one_phase<KCH>), dim3(NWG), dim3(NT), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, wts);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    printf("%-34s K=%4d  %7.3f us/phase\n", "graph of dependent launches", KCH * 256, best * 1e3 / phases);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("%s, %d CUs\n", pr.name, pr.multiProcessorCount);
    if (pr.multiProcessorCount < NWG) { printf("needs %d CUs for co-residency\n", NWG); return 1; }
    const int phases = 512;
    float *a, *b, *wts; Bar bar;
    CK(hipMalloc(&a, ROWS * 2048 * 4)); CK(hipMalloc(&b, ROWS * 2048 * 4)); CK(hipMalloc(&wts, (size_t)NWG * NT * 6 * 16));
    CK(hipMemset(a, 0, ROWS * 2048 * 4)); CK(hipMemset(b, 0, ROWS * 2048 * 4)); CK(hipMemset(wts, 0, (size_t)NWG * NT * 6 * 16));
    CK(hipMalloc(&bar.cnt, 256)); CK(hipMalloc(&bar.xcd_cnt, 8 * 4096)); CK(hipMalloc(&bar.err, 4)); CK(hipMemset(bar.err, 0, 4));
    run_graph<4>(a, b, wts, phases);
    run_graph<6>(a, b, wts, phases);
    run_persistent<0, 0, 0>("empty body, acquire spin", a, b, wts, bar, phases);
    run_persistent<1, 0, 0>("empty body, relaxed spin sleep 0", a, b, wts, bar, phases);
    run_persistent<1, 1, 0>("empty body, relaxed spin sleep 1", a, b, wts, bar, phases);
    run_persistent<1, 4, 0>("empty body, relaxed spin sleep 4", a, b, wts, bar, phases);
    run_persistent<1, 16, 0>("empty body, relaxed spin sleep 16", a, b, wts, bar, phases);
    run_persistent<2, 1, 0>("empty body, xcd flags sleep 1", a, b, wts, bar, phases);
    run_persistent<2, 4, 0>("empty body, xcd flags sleep 4", a, b, wts, bar, phases);
    run_persistent<3, 1, 0>("empty body, scalar spin sleep 1", a, b, wts, bar, phases);
    run_persistent<3, 8, 0>("empty body, scalar spin sleep 8", a, b, wts, bar, phases);
    run_persistent<1, 1, 4>("relaxed spin sleep 1", a, b, wts, bar, phases);
    run_persistent<1, 4, 4>("relaxed spin sleep 4", a, b, wts, bar, phases);
    run_persistent<1, 16, 4>("relaxed spin sleep 16", a, b, wts, bar, phases);
    run_persistent<2, 1, 4>("xcd flags sleep 1", a, b, wts, bar, phases);
    run_persistent<2, 4, 4>("xcd flags sleep 4", a, b, wts, bar, phases);
    run_persistent<3, 4, 4>("scalar spin sleep 4", a, b, wts, bar, phases);
    run_persistent<1, 4, 6>("relaxed spin sleep 4", a, b, wts, bar, phases);
    run_persistent<2, 4, 6>("xcd flags sleep 4", a, b, wts, bar, phases);
    return 0;
}
