// Shared device/host helpers for libamt_hip (gfx950 / CDNA4 only: 64-wide wavefronts, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AMT_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- status convention of include/amt_hip.h: 0 ok, <0 bad argument, >0 hipError_t ----
void amt_set_error(const char* fmt, ...);
#define AMT_CHECK_ARG(cond, ...)                \
    do {                                        \
        if (!(cond)) {                          \
            amt_set_error(__VA_ARGS__);         \
            return -1;                          \
        }                                       \
    } while (0)
#define AMT_HIP(expr)                                                                  \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            amt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int32_t)e__;                                                       \
        }                                                                              \
    } while (0)
#define AMT_LAUNCH_CHECK()                                                             \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            amt_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int32_t)e__;                                                       \
        }                                                                              \
    } while (0)

// ---- wavefront reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over the aligned group of G consecutive lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
