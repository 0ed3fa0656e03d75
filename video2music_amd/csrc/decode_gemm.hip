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
//  * one 1024-thread workgroup per 16 output columns x 16 rows; its 16 waves split K, all their tile loads are
//    issued up front (nothing depends on the LayerNorm prologue), partial tiles are summed in a
//    fixed order through LDS (deterministic, no atomics);
//  * the (normalised) input rows are staged once in LDS ([16][K+8] floats: conflict-free
//    ds_read_b128 A-fragments).
#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int MAXB = 32;
constexpr int XPAD = 8;
constexpr int MAX_TPW = 4;      // k-tiles per wave (16 waves): K <= 1024

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

constexpr int NW = 16;           // waves per workgroup: K is split 16 ways, one staged row per wave
constexpr int MT = 16;           // rows per workgroup (one MFMA row block); blockIdx.y selects the block

// KCH = float4 chunks per lane and row (K <= KCH*256); FULL: K == KCH*256, no lane predicates
template <int KCH, bool FULL>
__global__ __launch_bounds__(NW * 64) void decode_gemm_kernel(DecodeGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = p.K, LD = K + XPAD;
    float* xs = smem;                               // [16][LD]
    float* red = smem;                              // [16 waves][4 r][64 lanes], reuses xs after the MFMA phase
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, m0 = blockIdx.y * MT;
    const int kt_n = K / 16, tpw = (kt_n + NW - 1) / NW;
    const int kt0 = wave * tpw;

    // ---- issue this wave's weight tile loads first: they do not depend on the prologue ----
    float4 wt[MAX_TPW];
    const float* wp = p.Wp + (((size_t)nt * kt_n + (size_t)kt0) * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < MAX_TPW; ++i)
        if (i < tpw && kt0 + i < kt_n) wt[i] = (p.dbg & 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : ld4(wp + (size_t)i * 256);

    // ---- prologue: wave w stages (normalised) row m0+w into LDS ----
    const int r = m0 + wave;
    float4 v[KCH];
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((FULL || i < K) && r < p.B && !(p.dbg & 2)) v[c] = ld4(p.x + (size_t)r * p.ldx + i);
    }
    // epilogue operands do not depend on anything computed here: fetch them now
    const int el = tid & 63, er = (tid >> 6) & 3;
    const int row = m0 + 4 * (el >> 4) + er, n = nt * 16 + (el & 15);
    const bool live = tid < 256 && row < p.B && n < p.N && !(p.dbg & 32);
    float e_bias = 0.f, e_res = 0.f;
    int t = 0;
    if (live) {
        if (p.bias) e_bias = p.bias[n];
        if (p.mode == 0 && p.resid) e_res = p.resid[(size_t)row * p.ldr + n];
        if (p.pos) t = *p.pos;
    }
    if (p.ln_w && !(p.dbg & 4)) {
        const float inv_k = 1.0f / (float)K;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const float* gw = pass == 0 ? p.ln_w : p.ln2_w;
            const float* gb = pass == 0 ? p.ln_b : p.ln2_b;
            if (!gw) break;
            float4 g[KCH], h[KCH];
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int i = (c * 64 + lane) * 4;
                g[c] = (FULL || i < K) ? ld4(gw + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                h[c] = (FULL || i < K) ? ld4(gb + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
            const float mean = wave_sum(s) * inv_k;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int i = (c * 64 + lane) * 4;
                if (FULL || i < K) {
                    const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
                    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            }
            const float rstd = rsqrtf(wave_sum(q) * inv_k + p.eps);
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                v[c].x = (v[c].x - mean) * rstd * g[c].x + h[c].x; v[c].y = (v[c].y - mean) * rstd * g[c].y + h[c].y;
                v[c].z = (v[c].z - mean) * rstd * g[c].z + h[c].z; v[c].w = (v[c].w - mean) * rstd * g[c].w + h[c].w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if ((FULL || i < K) && !(p.dbg & 16)) st4(xs + wave * LD + i, (r < p.B) ? v[c] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __syncthreads();

    // the normalised rows are the residual of the following block: workgroup (nt, m-block) publishes its
    // 16 rows x columns 16nt..16nt+15
    if (p.ln_w && p.xn && nt * 16 < K && tid < 256) {
        const int rr = tid >> 4, c = tid & 15;
        if (m0 + rr < p.B) p.xn[(size_t)(m0 + rr) * K + nt * 16 + c] = xs[rr * LD + nt * 16 + c];
    }

    // ---- main: 4 MFMAs per k-tile ----
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* xa = xs + (lane & 15) * LD + kt0 * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < MAX_TPW; ++i) {
        if (i < tpw && kt0 + i < kt_n && !(p.dbg & 8)) {
            const float4 a0 = ld4(xa + i * 16);
            const float4 w = wt[i];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w.w, acc, 0, 0, 0);
        }
    }
    // ---- cross-wave reduction in fixed order (the partial tiles reuse the xs region) ----
    __syncthreads();
    float* rw = red + wave * 256;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) rw[rr * 64 + lane] = acc[rr];
    __syncthreads();
    if (live) {
        float val = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) val += red[w * 256 + tid];
        val += e_bias;
        if (n < p.scale_cols) val *= p.scale;
        if (p.mode == 0) {
            val += e_res;
            if (p.relu) val = fmaxf(val, 0.f);
            p.y[(size_t)row * p.ldy + n] = val;
        } else if (n < p.d) {
            p.y[(size_t)row * p.ldy + n] = val;
        } else {
            const int nn = (n < 2 * p.d) ? n - p.d : n - 2 * p.d;
            const int hh = nn / p.hd, cc = nn - hh * p.hd;
            float* dst = (n < 2 * p.d) ? p.kcache : p.vcache;
            dst[(((size_t)row * p.H + hh) * p.cap + t) * p.hd + cc] = val;
        }
    }
}

template <int KCH, bool FULL>
int32_t launch_variant(const DecodeGemmParams& p, size_t lds, hipStream_t stream) {
    static bool attr_set = false;        // > 64 KiB of dynamic LDS needs the opt-in (gfx950: 160 KiB per CU)
    if (!attr_set) {
        AMT_HIP(hipFuncSetAttribute((const void*)decode_gemm_kernel<KCH, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((decode_gemm_kernel<KCH, FULL>), dim3(cdiv(p.N, 16), cdiv(p.B, MT)), dim3(NW * 64), lds, stream, p);
    return 0;
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
    AMT_CHECK_ARG(p.K % 64 == 0 && p.K <= 1024, "decode_gemm: K=%d must be a multiple of 64 and <= 1024", p.K);
    AMT_CHECK_ARG(p.N > 0 && p.ldx >= p.K && p.ldx % 4 == 0, "decode_gemm: bad N/ldx");
    AMT_CHECK_ARG(p.mode == 0 || (p.kcache && p.vcache && p.d > 0 && p.N == 3 * p.d && p.d == p.H * p.hd), "decode_gemm: bad QKV epilogue");
    size_t lds = (size_t)MT * (p.K + XPAD) * sizeof(float);
    if (lds < (size_t)NW * 256 * sizeof(float)) lds = (size_t)NW * 256 * sizeof(float);
    int32_t rc;
    switch (p.K) {
        case 256: rc = launch_variant<1, true>(p, lds, stream); break;
        case 512: rc = launch_variant<2, true>(p, lds, stream); break;
        case 768: rc = launch_variant<3, true>(p, lds, stream); break;
        case 1024: rc = launch_variant<4, true>(p, lds, stream); break;
        default:
            if (p.K < 256) rc = launch_variant<1, false>(p, lds, stream);
            else rc = launch_variant<4, false>(p, lds, stream);
    }
    if (rc) return rc;
    AMT_LAUNCH_CHECK();
    return 0;
}
