// Stages of a hand-rolled grid barrier on gfx950 (see ubench_gridbar.cpp): release fence | atomic add | spin | acquire fence,
// stamped with s_memrealtime by thread 0 of every workgroup in one phase.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_gridbar2.cpp -o tools/ubench_gridbar2.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NWG = 256;
constexpr unsigned SPIN_MAX = 1u << 22;
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
#define NOW() __builtin_amdgcn_s_memrealtime()

// HIER: 0 one counter; 1 per-XCD counter then global counter (last of each XCD), everybody polls global; 2 as 1, last arriver sets per-XCD flags
template <int HIER, bool FENCES, int NTHR>
__global__ __launch_bounds__(NTHR) void k(unsigned* cnt, unsigned* xc, unsigned* flags, float* data, int phases, unsigned long long* st, int* err) {
    const int w = blockIdx.x;
    for (int p = 0; p < phases; ++p) {
        data[(size_t)w * NTHR + threadIdx.x] = (float)p;      // something dirty in L2 for the release to write back
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t0 = NOW();
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            unsigned long long t1 = NOW();
            const unsigned target = (unsigned)(p + 1);
            bool last = false;
            if (HIER == 0) {
                last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == target * NWG;
            } else {
                const unsigned x = xcc_id();
                const unsigned o = __hip_atomic_fetch_add(xc + x * 1024, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((o + 1) & 31u) == 0u) last = __hip_atomic_fetch_add(cnt, 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 32 == target * NWG;
            }
            unsigned long long t2 = NOW();
            unsigned n = 0;
            if (HIER == 2) {
                if (last) { for (int x = 0; x < 8; ++x) __hip_atomic_store(flags + x * 1024, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else { unsigned* f = flags + xcc_id() * 1024; while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++n < SPIN_MAX) __builtin_amdgcn_s_sleep(2); }
            } else {
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target * NWG && ++n < SPIN_MAX) __builtin_amdgcn_s_sleep(2);
            }
            unsigned long long t3 = NOW();
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            unsigned long long t4 = NOW();
            if (n >= SPIN_MAX) *err = 1;
            if (p == phases - 2) { unsigned long long* o = st + (size_t)w * 8; o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = n; }
        }
        __syncthreads();
    }
}

template <int HIER, bool FENCES, int NTHR>
void run(const char* name) {
    unsigned *cnt, *xc, *flags; float* data; unsigned long long* st; int* err;
    CK(hipMalloc(&cnt, 4096)); CK(hipMalloc(&xc, 8 * 4096)); CK(hipMalloc(&flags, 8 * 4096)); CK(hipMalloc(&data, (size_t)NWG * NTHR * 4)); CK(hipMalloc(&st, NWG * 64)); CK(hipMalloc(&err, 4));
    const int phases = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(cnt, 0, 4096)); CK(hipMemset(xc, 0, 8 * 4096)); CK(hipMemset(flags, 0, 8 * 4096)); CK(hipMemset(err, 0, 4));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<HIER, FENCES, NTHR>), dim3(NWG), dim3(NTHR), 0, 0, cnt, xc, flags, data, phases, st, err);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> h(NWG * 8); CK(hipMemcpy(h.data(), st, NWG * 64, hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull, lastarr = 0;
    for (int w = 0; w < NWG; ++w) { first = std::min(first, h[w * 8]); lastarr = std::max(lastarr, h[w * 8]); }
    const char* nm[4] = {"release fence", "atomic add", "spin", "acquire fence"};
    printf("%-40s %6.2f us/phase%s | entry spread %.2f us |", name, best * 1e3 / phases, herr ? " SPIN LIMIT" : "", (double)(lastarr - first) * 0.01);
    for (int i = 0; i < 4; ++i) {
        std::vector<double> v; for (int w = 0; w < NWG; ++w) v.push_back((double)(h[w * 8 + i + 1] - h[w * 8 + i]) * 0.01);
        std::sort(v.begin(), v.end());
        printf(" %s %.2f/%.2f/%.2f", nm[i], v[0], v[NWG / 2], v[NWG - 1]);
    }
    unsigned long long pollmax = 0; for (int w = 0; w < NWG; ++w) pollmax = std::max(pollmax, h[w * 8 + 5]);
    printf(" | max polls %llu\n", pollmax);
}

int main() {
    run<0, true, 1024>("one counter, fences, 1024 thr");
    run<0, false, 1024>("one counter, NO fences, 1024 thr");
    run<1, true, 1024>("xcd counters, fences");
    run<1, false, 1024>("xcd counters, NO fences");
    run<2, true, 1024>("xcd counters + xcd flags, fences");
    run<2, false, 1024>("xcd counters + xcd flags, NO fences");
    run<2, false, 64>("xcd counters + xcd flags, NO fences, 64 thr");
    run<0, false, 64>("one counter, NO fences, 64 thr");
    return 0;
}
