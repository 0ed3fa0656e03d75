// Skinny GEMM for one autoregressive decode step:  y[B<=32, N] = epi( LN(x)[B, K] . W[N, K]^T )
//
// At B = 32 clips the projections of a decode step are weight-streaming GEMVs: every weight byte
// is used for 32 rows only, so the kernel is shaped around reading W exactly once, coalesced, and
// around the launch count of the step (the LayerNorm that precedes each projection in the
// post-norm decoder, model/rpr.py:59-69, is folded into the prologue; bias / residual / ReLU /
// q-scaling / KV-cache scatter into the epilogue).
//
//  * weights are pre-packed once at load time into 16(n) x 16(k) tiles laid out in MFMA operand
//    order, so one wave-instruction (64 lanes x 16 B) fetches a whole 1 KiB tile contiguously and
//    feeds 4 v_mfma_f32_16x16x4_f32 per 16-row block without any shuffle;
//  * one 256-thread workgroup per 16 output columns; its 4 waves split K, all their tile loads are
//    issued up front (nothing depends on the LayerNorm prologue), partial tiles are summed in a
//    fixed order through LDS (deterministic, no atomics);
//  * the (normalised) input rows are staged once in LDS ([32][K+8] floats: conflict-free
//    ds_read_b128 A-fragments).
#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int MAXB = 32;
constexpr int XPAD = 8;
constexpr int MAX_TPW = 16;     // k-tiles per wave: K <= 1024

__global__ void pack_weight_kernel(const float* __restrict__ W, float* __restrict__ P, int N, int K) {
    const int kt_n = K / 16;
    const size_t tile = blockIdx.x;                 // nt*kt_n + kt
    const int nt = (int)(tile / kt_n), kt = (int)(tile % kt_n);
    const int lane = threadIdx.x;
    const int n = nt * 16 + (lane & 15), k = kt * 16 + 4 * (lane >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) v = ld4(W + (size_t)n * K + k);
    st4(P + (tile * 64 + lane) * 4, v);
}

__global__ __launch_bounds__(256) void decode_gemm_kernel(DecodeGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = p.K, LD = K + XPAD;
    float* xs = smem;                               // [32][LD]
    float* red = smem + MAXB * LD;                  // [4 waves][2 mt][4 r][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x;
    const int kt_n = K / 16, tpw = kt_n / 4;

    // ---- issue this wave's weight tile loads first: they do not depend on the prologue ----
    float4 wt[MAX_TPW];
    const float* wp = p.Wp + (((size_t)nt * kt_n + (size_t)wave * tpw) * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < MAX_TPW; ++i)
        if (i < tpw) wt[i] = ld4(wp + (size_t)i * 256);

    // ---- prologue: stage (normalised) rows into LDS; wave w owns rows w, w+4, ... ----
    for (int r = wave; r < MAXB; r += 4) {
        float4 v[4];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = (c * 64 + lane) * 4;
            v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < K && r < p.B) v[c] = ld4(p.x + (size_t)r * p.ldx + i);
            s += v[c].x + v[c].y + v[c].z + v[c].w;
        }
        if (p.ln_w && r < p.B) {
            const float* gw = p.ln_w; const float* gb = p.ln_b;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {
                    if (!p.ln2_w) break;
                    gw = p.ln2_w; gb = p.ln2_b;
                    s = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) s += v[c].x + v[c].y + v[c].z + v[c].w;
                }
                const float mean = wave_sum(s) / K;
                float q = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = (c * 64 + lane) * 4;
                    if (i < K) {
                        const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
                        q += dx * dx + dy * dy + dz * dz + dw * dw;
                    }
                }
                const float rstd = 1.0f / sqrtf(wave_sum(q) / K + p.eps);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = (c * 64 + lane) * 4;
                    if (i < K) {
                        const float4 g = ld4(gw + i), h = ld4(gb + i);
                        v[c].x = (v[c].x - mean) * rstd * g.x + h.x; v[c].y = (v[c].y - mean) * rstd * g.y + h.y;
                        v[c].z = (v[c].z - mean) * rstd * g.z + h.z; v[c].w = (v[c].w - mean) * rstd * g.w + h.w;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = (c * 64 + lane) * 4;
            if (i < K) st4(xs + r * LD + i, v[c]);
        }
    }
    __syncthreads();

    // the normalised rows are the residual of the following block: workgroup nt publishes columns 16nt..16nt+15
    if (p.ln_w && p.xn && nt * 16 < K) {
        for (int e = tid; e < MAXB * 16; e += 256) {
            const int r = e >> 4, c = e & 15;
            if (r < p.B) p.xn[(size_t)r * K + nt * 16 + c] = xs[r * LD + nt * 16 + c];
        }
    }

    // ---- main: 4 MFMAs per (k-tile, 16-row block) ----
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* xa = xs + (lane & 15) * LD + (wave * tpw) * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < MAX_TPW; ++i) {
        if (i < tpw) {
            const float4 a0 = ld4(xa + i * 16), a1 = ld4(xa + 16 * LD + i * 16);
            const float4 w = wt[i];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w.w, acc1, 0, 0, 0);
        }
    }
    // ---- cross-wave reduction in fixed order ----
    float* rw = red + wave * 512;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rw[(0 * 4 + r) * 64 + lane] = acc0[r];
        rw[(1 * 4 + r) * 64 + lane] = acc1[r];
    }
    __syncthreads();
    const int t = p.pos ? *p.pos : 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int f = tid + half * 256;             // f = (mt*4 + r)*64 + l
        const int l = f & 63, r = (f >> 6) & 3, mt = f >> 8;
        float v = ((red[f] + red[512 + f]) + red[1024 + f]) + red[1536 + f];
        const int row = mt * 16 + 4 * (l >> 4) + r, n = nt * 16 + (l & 15);
        if (row >= p.B || n >= p.N) continue;
        if (p.bias) v += p.bias[n];
        if (n < p.scale_cols) v *= p.scale;
        if (p.mode == 0) {
            if (p.resid) v += p.resid[(size_t)row * p.ldr + n];
            if (p.relu) v = fmaxf(v, 0.f);
            p.y[(size_t)row * p.ldy + n] = v;
        } else {
            if (n < p.d) {
                p.y[(size_t)row * p.ldy + n] = v;
            } else {
                const int nn = (n < 2 * p.d) ? n - p.d : n - 2 * p.d;
                const int hh = nn / p.hd, cc = nn - hh * p.hd;
                float* dst = (n < 2 * p.d) ? p.kcache : p.vcache;
                dst[(((size_t)row * p.H + hh) * p.cap + t) * p.hd + cc] = v;
            }
        }
    }
}

}  // namespace

int32_t amt_launch_pack_weight(const float* W, float* P, int N, int K, hipStream_t stream) {
    AMT_CHECK_ARG(K % 16 == 0, "pack_weight: K=%d must be a multiple of 16", K);
    const int tiles = cdiv(N, 16) * (K / 16);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(tiles), dim3(64), 0, stream, W, P, N, K);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_decode_gemm(const DecodeGemmParams& p, hipStream_t stream) {
    AMT_CHECK_ARG(p.B > 0 && p.B <= MAXB, "decode_gemm: B=%d outside (0,%d]", p.B, MAXB);
    AMT_CHECK_ARG(p.K % 64 == 0 && p.K <= 16 * 4 * MAX_TPW, "decode_gemm: K=%d must be a multiple of 64 and <= %d", p.K, 16 * 4 * MAX_TPW);
    AMT_CHECK_ARG(p.N > 0 && p.ldx >= p.K && p.ldx % 4 == 0, "decode_gemm: bad N/ldx");
    AMT_CHECK_ARG(p.mode == 0 || (p.kcache && p.vcache && p.d > 0 && p.N == 3 * p.d && p.d == p.H * p.hd), "decode_gemm: bad QKV epilogue");
    const size_t lds = ((size_t)MAXB * (p.K + XPAD) + 4 * 512) * sizeof(float);
    static bool attr_set = false;        // > 64 KiB of dynamic LDS needs the opt-in (gfx950: 160 KiB per CU)
    if (!attr_set) {
        AMT_HIP(hipFuncSetAttribute((const void*)decode_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(decode_gemm_kernel, dim3(cdiv(p.N, 16)), dim3(256), lds, stream, p);
    AMT_LAUNCH_CHECK();
    return 0;
}
