// Mamba pieces of the regression head `VideoRegression(regModel='bimamba+')`:
//   model/mamba.py:172-175,268-272   depthwise causal Conv1d(kernel d_conv, padding d_conv-1)[..., :L] + SiLU
//   model/mamba.py:291-354           delta = softplus(dt_proj(.) + bias); h_t = exp(delta A) h_{t-1} + delta B_t x_t;
//                                    y_t = h_t . C_t + D x_t                    (the reference evaluates the same
//                                    recurrence with a Blelloch parallel scan, model/pscan.py — rounding differs)
//   model/mamba.py:281-287           Mamba+ gate: out = y * silu(z) + x * (1 - sigmoid(silu(z)))   (use_version 1)
// The backward branch of the bidirectional layer (bimamba.py:171-185) runs the same block on the time-flipped
// sequence and flips the result back; every other op of the block is per-position, so `reverse` here (walk time from
// the end, write results at the original positions) replaces both flips.
//
// Both kernels are tiny and latency-bound at the head's size (B=1, L=300, d_inner=256, N=16 states): the scan keeps
// the N states of a channel on N adjacent lanes (one DPP reduction per step for y_t) and stages TCH time steps of its
// operands in LDS per pass, so global memory is touched once, coalesced over channels, outside the recurrence.
#include "../../include/amt_hip.h"
#include "amt_common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float softplus(float v) { return v > 20.f ? v : log1pf(__expf(v)); }    // torch threshold = 20

__global__ void dwconv_silu_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                   const float* __restrict__ bias, float* __restrict__ y, int B, int L, int C, int K, int reverse) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * L * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int l = (int)((idx / C) % L);
    const int b = (int)(idx / ((size_t)C * L));
    float acc = bias ? bias[c] : 0.f;
    for (int j = 0; j < K; ++j) {
        const int t = reverse ? l + (K - 1 - j) : l - (K - 1) + j;
        if (t >= 0 && t < L) acc += w[c * K + j] * x[((size_t)b * L + t) * ldx + c];
    }
    y[idx] = silu(acc);
}

constexpr int TCH = 32;          // time steps staged per pass
constexpr int CPB = 16;          // channels per 256-thread block (N = 16 lanes each)

template <int N>
__global__ __launch_bounds__(CPB * N) void selective_scan_kernel(ScanParams p) {
    // per staged step: x, delta = softplus(.), silu(z), partial y for the block's channels; B and C rows
    __shared__ float sx[TCH][CPB], sd[TCH][CPB], sz[TCH][CPB], sy[TCH][CPB], sB[TCH][N], sC[TCH][N];
    const int tid = threadIdx.x, n = tid % N, cl = tid / N;
    const int c0 = blockIdx.x * CPB, b = blockIdx.y;
    const int ch = c0 + cl;
    const float A = ch < p.ED ? -__expf(p.A_log[(size_t)ch * N + n]) : 0.f;
    float h = 0.f;
    // global -> registers -> LDS, one pass ahead: the loads of pass k+1 are in flight while pass k runs its recurrence
    constexpr int EPT = TCH * CPB / (CPB * N), BPT = TCH * N / (CPB * N);     // staged elements per thread
    float rx[EPT], rd[EPT], rz[EPT], rB[BPT], rC[BPT];
    auto fetch = [&](int s0) {
        const int steps = min(TCH, p.L - s0);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * CPB * N, s = i / CPB, cc = i % CPB;
            const int t = p.reverse ? p.L - 1 - (s0 + s) : s0 + s;
            const size_t row = (size_t)b * p.L + t;
            const bool ok = s < steps && c0 + cc < p.ED;
            rx[e] = ok ? p.x[row * p.ldx + c0 + cc] : 0.f;
            rd[e] = ok ? p.draw[row * p.ldd + c0 + cc] + p.dt_bias[c0 + cc] : -INFINITY;     // softplus(-inf) = 0: the step is a no-op
            rz[e] = ok ? p.z[row * p.ldz + c0 + cc] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < BPT; ++e) {
            const int i = tid + e * CPB * N, s = i / N, nn = i % N;
            const int t = p.reverse ? p.L - 1 - (s0 + s) : s0 + s;
            const size_t row = (size_t)b * p.L + t;
            rB[e] = s < steps ? p.Bm[row * p.ldbc + nn] : 0.f;
            rC[e] = s < steps ? p.Cm[row * p.ldbc + nn] : 0.f;
        }
    };
    fetch(0);
    for (int s0 = 0; s0 < p.L; s0 += TCH) {
        const int steps = min(TCH, p.L - s0);
        // everything that is not the recurrence itself happens here, in parallel over (step, channel): softplus, SiLU
        // (steps past the end of the sequence are staged as delta = 0, x = 0: they leave h unchanged, so the
        // recurrence below always runs TCH steps and unrolls)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * CPB * N, s = i / CPB, cc = i % CPB;
            sx[s][cc] = rx[e];
            sd[s][cc] = rd[e] == -INFINITY ? 0.f : softplus(rd[e]);
            sz[s][cc] = silu(rz[e]);
        }
#pragma unroll
        for (int e = 0; e < BPT; ++e) {
            const int i = tid + e * CPB * N, s = i / N, nn = i % N;
            sB[s][nn] = rB[e];
            sC[s][nn] = rC[e];
        }
        __syncthreads();
        if (s0 + TCH < p.L) fetch(s0 + TCH);
        // the recurrence: one FMA on the dependency chain per step; exp / LDS reads of later steps overlap (unrolled)
#pragma unroll 8
        for (int s = 0; s < TCH; ++s) {
            const float delta = sd[s][cl];
            h = __expf(delta * A) * h + (delta * sB[s][n]) * sx[s][cl];
            const float ys = group_sum<N>(h * sC[s][n]);
            if (n == 0) sy[s][cl] = ys;
        }
        __syncthreads();
        // + D x, gate, store
        for (int i = tid; i < steps * CPB; i += CPB * N) {
            const int s = i / CPB, cc = i % CPB;
            const int t = p.reverse ? p.L - 1 - (s0 + s) : s0 + s;
            if (c0 + cc < p.ED) {
                const float xv = sx[s][cc], zs = sz[s][cc];
                const float yv = sy[s][cc] + p.D[c0 + cc] * xv;
                p.y[((size_t)b * p.L + t) * p.ldy + c0 + cc] = p.version == 1 ? yv * zs + xv * (1.0f - 1.0f / (1.0f + __expf(-zs))) : yv * zs;
            }
        }
        __syncthreads();
    }
}

// The same scan for wider state spaces, N = 16 * NS states per channel ('moemamba' builds its blocks with d_state = d_hidden,
// video_regression.py:143-147): a channel still owns 16 lanes, each lane carries NS states; operands are staged straight into
// LDS (no register prefetch -- B and C rows alone are 2 * TCH * N floats per pass).
template <int NS>
__global__ __launch_bounds__(CPB * 16) void selective_scan_wide_kernel(ScanParams p) {
    constexpr int N = 16 * NS, NT = CPB * 16, TW = 16;      // TW steps staged per pass (B and C rows: 2 * TW * N floats)
    __shared__ float sx[TW][CPB], sd[TW][CPB], sz[TW][CPB], sy[TW][CPB], sB[TW][N], sC[TW][N];
    const int tid = threadIdx.x, n = tid % 16, cl = tid / 16;
    const int c0 = blockIdx.x * CPB, b = blockIdx.y;
    const int ch = c0 + cl;
    float A[NS], h[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        A[j] = ch < p.ED ? -__expf(p.A_log[(size_t)ch * N + j * 16 + n]) : 0.f;
        h[j] = 0.f;
    }
    for (int s0 = 0; s0 < p.L; s0 += TW) {
        const int steps = min(TW, p.L - s0);
        for (int i = tid; i < TW * CPB; i += NT) {
            const int s = i / CPB, cc = i % CPB;
            const int t = p.reverse ? p.L - 1 - (s0 + s) : s0 + s;
            const size_t row = (size_t)b * p.L + t;
            const bool ok = s < steps && c0 + cc < p.ED;
            sx[s][cc] = ok ? p.x[row * p.ldx + c0 + cc] : 0.f;
            sd[s][cc] = ok ? softplus(p.draw[row * p.ldd + c0 + cc] + p.dt_bias[c0 + cc]) : 0.f;     // delta = 0: the step is a no-op
            sz[s][cc] = ok ? silu(p.z[row * p.ldz + c0 + cc]) : 0.f;
        }
        for (int i = tid; i < TW * N; i += NT) {
            const int s = i / N, nn = i % N;
            const int t = p.reverse ? p.L - 1 - (s0 + s) : s0 + s;
            const size_t row = (size_t)b * p.L + t;
            sB[s][nn] = s < steps ? p.Bm[row * p.ldbc + nn] : 0.f;
            sC[s][nn] = s < steps ? p.Cm[row * p.ldbc + nn] : 0.f;
        }
        __syncthreads();
#pragma unroll 2
        for (int s = 0; s < TW; ++s) {
            const float delta = sd[s][cl], xv = sx[s][cl];
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int nn = j * 16 + n;
                h[j] = __expf(delta * A[j]) * h[j] + (delta * sB[s][nn]) * xv;
                part += h[j] * sC[s][nn];
            }
            const float ys = group_sum<16>(part);
            if (n == 0) sy[s][cl] = ys;
        }
        __syncthreads();
        for (int i = tid; i < steps * CPB; i += NT) {
            const int s = i / CPB, cc = i % CPB;
            const int t = p.reverse ? p.L - 1 - (s0 + s) : s0 + s;
            if (c0 + cc < p.ED) {
                const float xv = sx[s][cc], zs = sz[s][cc];
                const float yv = sy[s][cc] + p.D[c0 + cc] * xv;
                p.y[((size_t)b * p.L + t) * p.ldy + c0 + cc] = p.version == 1 ? yv * zs + xv * (1.0f - 1.0f / (1.0f + __expf(-zs))) : yv * zs;
            }
        }
        __syncthreads();
    }
}

__global__ void concat2_kernel(const float* __restrict__ a, int da, const float* __restrict__ b, int db,
                               float* __restrict__ out, int rows, int ld_out) {
    const int row = blockIdx.x;
    for (int c = threadIdx.x; c < ld_out; c += blockDim.x) {
        float v = 0.f;
        if (c < da) v = a[(size_t)row * da + c];
        else if (c < da + db) v = b[(size_t)row * db + (c - da)];
        out[(size_t)row * ld_out + c] = v;
    }
}

}  // namespace

int32_t amt_launch_dwconv_silu(const float* x, int ldx, const float* w, const float* bias, float* y, int B, int L, int C, int K,
                               int reverse, hipStream_t stream) {
    AMT_CHECK_ARG(B > 0 && L > 0 && C > 0 && K > 0 && ldx >= C, "dwconv: bad shape B=%d L=%d C=%d K=%d ldx=%d", B, L, C, K, ldx);
    const size_t total = (size_t)B * L * C;
    hipLaunchKernelGGL(dwconv_silu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, ldx, w, bias, y, B, L, C, K, reverse);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_selective_scan(const ScanParams& p, hipStream_t stream) {
    AMT_CHECK_ARG(p.B > 0 && p.L > 0 && p.ED > 0, "selective_scan: bad shape B=%d L=%d ED=%d", p.B, p.L, p.ED);
    AMT_CHECK_ARG(p.N == 16 || p.N == 32 || p.N == 64 || p.N == 128 || p.N == 256,
                  "selective_scan: d_state=%d not in {16 (the reference's default), 32, 64, 128, 256}", p.N);
    AMT_CHECK_ARG(p.ldx >= p.ED && p.ldd >= p.ED && p.ldz >= p.ED && p.ldy >= p.ED && p.ldbc >= p.N, "selective_scan: bad leading dimension");
    const dim3 grid(cdiv(p.ED, CPB), p.B), block(CPB * 16);
    switch (p.N) {
        case 16: hipLaunchKernelGGL(selective_scan_kernel<16>, grid, block, 0, stream, p); break;
        case 32: hipLaunchKernelGGL(selective_scan_wide_kernel<2>, grid, block, 0, stream, p); break;
        case 64: hipLaunchKernelGGL(selective_scan_wide_kernel<4>, grid, block, 0, stream, p); break;
        case 128: hipLaunchKernelGGL(selective_scan_wide_kernel<8>, grid, block, 0, stream, p); break;
        default: hipLaunchKernelGGL(selective_scan_wide_kernel<16>, grid, block, 0, stream, p); break;
    }
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_concat2(const float* a, int da, const float* b, int db, float* out, int rows, int ld_out, hipStream_t stream) {
    AMT_CHECK_ARG(rows > 0 && da > 0 && db >= 0 && ld_out >= da + db, "concat2: bad shape");
    hipLaunchKernelGGL(concat2_kernel, dim3(rows), dim3(256), 0, stream, a, da, b, db, out, rows, ld_out);
    AMT_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int32_t amt_dwconv1d_silu_fwd(const float* x, int32_t ldx, const float* w, const float* bias, float* y,
                                         int32_t B, int32_t L, int32_t C, int32_t K, int32_t reverse, void* stream) {
    AMT_CHECK_ARG(x && w && y, "amt_dwconv1d_silu_fwd: null pointer");
    return amt_launch_dwconv_silu(x, ldx, w, bias, y, B, L, C, K, reverse, (hipStream_t)stream);
}

extern "C" int32_t amt_selective_scan_fwd(const float* x, int32_t ldx, const float* delta_raw, int32_t ld_delta, const float* dt_bias,
                                          const float* A_log, const float* Bm, const float* Cm, int32_t ld_bc, const float* D,
                                          const float* z, int32_t ldz, float* y, int32_t ldy, int32_t B, int32_t L, int32_t ED,
                                          int32_t N, int32_t version, int32_t reverse, void* stream) {
    AMT_CHECK_ARG(x && delta_raw && dt_bias && A_log && Bm && Cm && D && z && y, "amt_selective_scan_fwd: null pointer");
    AMT_CHECK_ARG(version == 0 || version == 1, "amt_selective_scan_fwd: version %d not in {0,1}", version);
    ScanParams p{};
    p.x = x; p.ldx = ldx; p.draw = delta_raw; p.ldd = ld_delta; p.dt_bias = dt_bias; p.A_log = A_log; p.Bm = Bm; p.Cm = Cm; p.ldbc = ld_bc;
    p.D = D; p.z = z; p.ldz = ldz; p.y = y; p.ldy = ldy; p.B = B; p.L = L; p.ED = ED; p.N = N; p.version = version; p.reverse = reverse;
    return amt_launch_selective_scan(p, (hipStream_t)stream);
}

extern "C" int32_t amt_concat2_fwd(const float* a, int32_t da, const float* b, int32_t db, float* out, int32_t rows, int32_t ld_out,
                                   void* stream) {
    AMT_CHECK_ARG(a && (b || db == 0) && out, "amt_concat2_fwd: null pointer");
    return amt_launch_concat2(a, da, b, db, out, rows, ld_out, (hipStream_t)stream);
}

extern "C" int32_t amt_linear_ex_fwd(const float* x, int32_t ldx, const float* w, int32_t ldw, const float* bias, const float* resid,
                                     int32_t ldr, float* y, int32_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    AMT_CHECK_ARG(x && w && y, "amt_linear_ex_fwd: null pointer");
    AMT_CHECK_ARG(act >= 0 && act <= 3, "amt_linear_ex_fwd: act %d not in {0 none, 1 relu, 2 sigmoid, 3 silu}", act);
    GemmParams g = gemm_params(x, ldx, w, ldw, y, ldy, M, N, K, bias);
    g.resid = resid; g.ldr = ldr; g.relu = act == 1 ? 1 : (act == 3 ? 2 : 0); g.sigmoid = act == 2;
    return amt_launch_gemm(g, (hipStream_t)stream);
}

extern "C" int32_t amt_layernorm_post_fwd(const float* x, const float* resid, const float* w, const float* b, const float* post,
                                          float* y, int32_t rows, int32_t dim, float eps, void* stream) {
    AMT_CHECK_ARG(x && y, "amt_layernorm_post_fwd: null pointer");
    return amt_launch_layernorm(x, resid, w, b, nullptr, nullptr, y, rows, dim, eps, (hipStream_t)stream, post);
}
