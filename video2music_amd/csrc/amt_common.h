// Shared device/host helpers for libamt_hip (gfx950 / CDNA4 only: 64-wide wavefronts, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AMT_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));      // operand pairs of the packed fp32 ops (v_pk_fma_f32)
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
            (void)hipGetLastError();   /* reported here: the next call's launch check must not see it again */ \
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

// ---- wavefront reductions (64 lanes) on the DPP cross-lane path ----
// __shfl_xor lowers to ds_bpermute_b32 (an LDS round trip per step); the row-local DPP modifiers
// below are plain VALU operands.  After quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror and
// row_mirror every lane of a 16-lane row holds its row's total; rows are combined via readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row_sum16(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// sum over the aligned group of G consecutive lanes (G power of two <= 64); every lane gets the total
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if (G >= 2) v += dpp_mov<DPP_XOR1>(v);
    if (G >= 4) v += dpp_mov<DPP_XOR2>(v);
    if (G >= 8) v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    if (G >= 16) v += dpp_mov<DPP_ROW_MIRROR>(v);
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- tuning constants of the launch heuristics ----
// One immutable, process-wide record.  The release build returns the defaults below (what every committed number was measured
// with).  A build with -DAMT_EXPERIMENT (tools/ab_build.sh TAG -DAMT_EXPERIMENT) reads each field ONCE, under std::call_once,
// from the environment variable named beside it -- the A/B switches of DESIGN.md section 9; the release library reads no
// environment variable and keeps no other mutable global besides the per-device attribute flags guarded by mutexes.
struct AmtTuning {
    int kv_pad = -1;              // AMT_KV_PAD: padding rows per (clip, head) slice of the self-attention cache; -1 = make Tcap odd
    int steps_per_graph = 16;     // AMT_STEPS_PER_GRAPH: decode steps per captured hipGraph (a power of two; the first step of a graph keeps its
                                  // separate sampling head, the others take the decision in their first attention)
    int nt_mask = 3;              // AMT_NT: non-temporal K/V loads, bit 0 self-attention, bit 1 cross-attention
    int wide_grouped = 1;         // AMT_WIDE_GROUPED: grouped down-projections on the wide skinny GEMM
    int wide_ntw = 4;             // AMT_WIDE_NTW: column tiles per workgroup of the wide skinny GEMM
    int wide_rb2 = 1;             // AMT_WIDE_RB2: 17-32 rows of a wide product in one workgroup (both row blocks share the weight tiles)
    long gemm_small_m = 4096;     // AMT_GEMM_SMALL_M / AMT_GEMM_SMALL_MN: dense products at or below go to the skinny GEMM
    long gemm_small_mn = 650000;
    int gemm_t64_below = 768;     // AMT_GEMM_T64_BELOW: fewer 128x128 tiles than this take the 64x64 instantiation
    int gemm_pf = 22;             // AMT_GEMM_PF: prefetch distances (tens: big tile, units: small tile)
    int exp_a = 0, exp_b = 0;     // free switches for the experiment of the day (amt_experiment_set, experiment builds only)
    int prepacked = 0;            // AMT_DBG bit 4: amt_decode_linear_fwd finds the packed weight in its scratch (micro-benchmarks)
};
const AmtTuning& amt_tuning();
