// fp32 GEMM on the CDNA4 matrix cores:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Both operands are K-contiguous (activations row-major, weights in torch's nn.Linear (out,in)
// layout), so one kernel serves every dense projection of the AMT encoder / prefill path
// (Linear_vis, packed in-proj, out-proj, FFN, cross K/V projection, Wout).
//
// Tiling (gfx950): 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave a
// 64x64 sub-tile = 2x2 v_mfma_f32_32x32x2_f32 accumulators (exact fp32, k-ordered fma chain); a 64x64-tile
// instantiation (one accumulator per wave) serves launches with too few 128x128 tiles to fill the GPU.
// K is consumed 32 at a time through a double-buffered LDS image [128][36] (row stride 144 B, so
// a ds_read_b128 lane group touches 16 distinct 16-B bank slots).  One ds_read_b128 per operand
// fragment feeds 4 MFMAs: lane (r, h) holds k = k0+4h+e for e=0..3, MFMA e then sums k0+e (h=0)
// and k0+4+e (h=1) -- any pairing is legal as long as A and B use the same one.
#include <stdlib.h>

#include <type_traits>

#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int BK = 32, LDS_LD = BK + 4;   // floats
// TM x TN = 32x32 MFMA tiles per wave (waves 2x2): workgroup tile 64*TM x 64*TN, i.e. 128x128 (2,2) or 64x64 (1,1)


__device__ __forceinline__ void store_out(const GemmParams& p, int row, int col, float v) {
    if (row >= p.M || col >= p.N) return;
    if (p.bias) v += p.bias[col];
    if (col < p.scale_cols) v *= p.scale;
    if (p.silu_mul) {
        const float g = p.silu_mul[(size_t)row * p.ld_silu + col];
        v *= g / (1.0f + __expf(-g));
    }
    if (p.rowadd) v += p.rowadd[(size_t)(row % p.rowadd_period) * p.N + col];
    if (p.resid) v += p.resid[(size_t)row * p.ldr + col];
    if (p.relu == 1) v = fmaxf(v, 0.f);
    else if (p.relu == 2) v = v / (1.0f + __expf(-v));
    if (p.sigmoid) v = 1.0f / (1.0f + __expf(-v));
    if (p.head_split == 0) {
        p.C[(size_t)row * p.ldc + col] = v;
    } else {
        // row = b*seq + s, col = part*d + h*hd + c  ->  out[part][b][h][s][c]
        int b = row / p.hs_seq, s = row - b * p.hs_seq;
        int part = col / p.hs_d, cc = col - part * p.hs_d;
        int h = cc / p.hs_hd, c = cc - h * p.hs_hd;
        size_t off = (size_t)part * p.hs_part_stride +
                     (((size_t)b * p.hs_heads + h) * p.hs_seq_cap + s) * p.hs_hd + c;
        p.C[off] = v;
    }
}

// GATHER: rows of A come through an index (mixture-of-experts plan).  The dense instantiation has NO guarded load: rows / columns
// past the edge read a clamped address (their results are never stored), because a load under a branch -- or behind the wait for
// a gather index -- makes the compiler drain every outstanding load (s_waitcnt vmcnt(0)) in the middle of the k loop.
// PF: prefetch distance in k-tiles (1: the tile stored to LDS at the end of an iteration was requested at its top; 2: a whole
// iteration earlier, through a second register stage)
template <int TM, int TN, bool GATHER = false, int PF = 2>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BMAX = BM > BN ? BM : BN;
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDS_LD];   // [buf][A rows | W rows][k]
    constexpr int WOFF = BM * LDS_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: consecutive tiles along N share the A panel, keep them on one XCD
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (p.tile_group) {              // grouped mode: per-tile expert weights
        const int g = p.tile_group[m0 / 128];       // groups are planned in 128-row segments
        if (g < 0) return;
        p.W += (size_t)g * p.w_group_stride;
        if (p.bias) p.bias += (size_t)g * p.bias_group_stride;
    }

    // staging: 256 threads x (rows/32) passes x float4 cover a rows x 32 tile (8 float4 per row)
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    // two register stages: the k-tile stored to LDS in an iteration was requested a whole iteration earlier (prefetch distance 2),
    // so the LDS store never waits for global memory behind a single k-tile of MFMA work
    float4 ra[2][2 * TM], rw[2][2 * TN];
    // row pointers, fixed for the whole k loop (gather indices are read once, here); out-of-range rows alias row 0 / the last row
    const float* arow[2 * TM];
    const float* wrow[2 * TN];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) {
        const int gm = min(m0 + srow + i * 32, p.M - 1);
        int am = gm;
        if (GATHER) am = max(p.a_gather[gm], 0);     // -1 = padding slot of the plan: any row, its product is never combined
        arow[i] = p.A + (size_t)am * p.lda + scol;
    }
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) wrow[i] = p.W + (size_t)min(n0 + srow + i * 32, p.N - 1) * p.ldw + scol;
    auto gload = [&](int k0, int st) {
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) ra[st][i] = ld4(arow[i] + k0);
#pragma unroll
        for (int i = 0; i < 2 * TN; ++i) rw[st][i] = ld4(wrow[i] + k0);
    };
    auto lstore = [&](int buf, int st) {
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) st4(&lds[buf][(srow + i * 32) * LDS_LD + scol], ra[st][i]);
#pragma unroll
        for (int i = 0; i < 2 * TN; ++i) st4(&lds[buf][WOFF + (srow + i * 32) * LDS_LD + scol], rw[st][i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const int nk = p.K / BK;
    gload(0, 0);
    if (PF == 2) gload(min(1, nk - 1) * BK, 1);
    lstore(0, 0);
    __syncthreads();
    // (two iterations per trip so that the register stage is a compile-time index)
    auto body = [&](int kt, auto stage) {
        constexpr int ST = decltype(stage)::value;       // stage that holds k-tile kt + 1; k-tile kt + 2 goes into the other
        const int cur = kt & 1;
        gload(min(kt + PF, nk - 1) * BK, PF == 2 ? ST ^ 1 : 0);          // unconditional (the last two trips re-read the last k-tile): a guarded
                                                         // load hides the number of loads in flight and every wait becomes vmcnt(0)
        const float* la = &lds[cur][(wr * 32 * TM + fr) * LDS_LD + fh * 4];
        const float* lw = &lds[cur][WOFF + (wc * 32 * TN + fr) * LDS_LD + fh * 4];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = ld4(la + i * 32 * LDS_LD + kk);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = ld4(lw + j * 32 * LDS_LD + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float a = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
                        const float b = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                    }
        }
        if (kt + 1 < nk) {
            lstore(cur ^ 1, PF == 2 ? ST : 0);
            __syncthreads();
        }
    };
    for (int kt = 0; kt < nk; kt += 2) {
        body(kt, std::integral_constant<int, 1>{});
        if (kt + 1 < nk) body(kt + 1, std::integral_constant<int, 0>{});
    }

    // ---- epilogue ----
    // C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const bool vec_ok = (p.N % 4 == 0) && (p.head_split || p.ldc % 4 == 0) && (!p.resid || p.ldr % 4 == 0) &&
                        (!p.silu_mul || p.ld_silu % 4 == 0);
    if (!vec_ok) {      // odd widths (Wout: N = 159): element-wise stores
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int row = m0 + (wr * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    int col = n0 + (wc * TN + j) * 32 + fr;
                    store_out(p, row, col, acc[i][j][e]);
                }
        return;
    }
    // transpose the tile through LDS (the operand buffers are free now) so that every lane owns 4 consecutive
    // columns: bias / residual / gate loads and the output stores become 16-byte and row-contiguous
    constexpr int CLD = BN + 4;
    static_assert(BM * CLD <= 2 * (BM + BN) * LDS_LD, "output tile does not fit the operand buffers");
    float* ct = &lds[0][0];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (wr * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                ct[r * CLD + (wc * TN + j) * 32 + fr] = acc[i][j][e];
            }
    __syncthreads();
    constexpr int TPR = BN / 4;                      // threads per output row
    const int c4 = (tid % TPR) * 4, col = n0 + c4;
    if (col >= p.N) return;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = ld4(p.bias + col);
    const float sc[4] = {col + 0 < p.scale_cols ? p.scale : 1.f, col + 1 < p.scale_cols ? p.scale : 1.f,
                         col + 2 < p.scale_cols ? p.scale : 1.f, col + 3 < p.scale_cols ? p.scale : 1.f};
    for (int r = tid / TPR; r < BM; r += 256 / TPR) {
        const int row = m0 + r;
        if (row >= p.M) break;
        float4 v = ld4(&ct[r * CLD + c4]);
        v.x = (v.x + bias4.x) * sc[0]; v.y = (v.y + bias4.y) * sc[1]; v.z = (v.z + bias4.z) * sc[2]; v.w = (v.w + bias4.w) * sc[3];
        if (p.silu_mul) {
            const float4 g = ld4(p.silu_mul + (size_t)row * p.ld_silu + col);
            v.x *= g.x / (1.0f + __expf(-g.x)); v.y *= g.y / (1.0f + __expf(-g.y));
            v.z *= g.z / (1.0f + __expf(-g.z)); v.w *= g.w / (1.0f + __expf(-g.w));
        }
        if (p.rowadd) {
            const float4 a = ld4(p.rowadd + (size_t)(row % p.rowadd_period) * p.N + col);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (p.resid) {
            const float4 a = ld4(p.resid + (size_t)row * p.ldr + col);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (p.relu == 2) {
            v.x = v.x / (1.0f + __expf(-v.x)); v.y = v.y / (1.0f + __expf(-v.y));
            v.z = v.z / (1.0f + __expf(-v.z)); v.w = v.w / (1.0f + __expf(-v.w));
        }
        if (p.sigmoid) {
            v.x = 1.0f / (1.0f + __expf(-v.x)); v.y = 1.0f / (1.0f + __expf(-v.y));
            v.z = 1.0f / (1.0f + __expf(-v.z)); v.w = 1.0f / (1.0f + __expf(-v.w));
        }
        if (p.head_split == 0) {
            st4(p.C + (size_t)row * p.ldc + col, v);
        } else {
            // row = b*seq + s, col = part*d + h*hd + c  ->  out[part][b][h][s][c]; hd % 4 == 0 keeps the 4 columns in one head
            const int b = row / p.hs_seq, ss = row - b * p.hs_seq;
            const int part = col / p.hs_d, cc = col - part * p.hs_d;
            const int h = cc / p.hs_hd, c = cc - h * p.hs_hd;
            st4(p.C + (size_t)part * p.hs_part_stride + (((size_t)b * p.hs_heads + h) * p.hs_seq_cap + ss) * p.hs_hd + c, v);
        }
    }
}

}  // namespace

int32_t amt_launch_gemm(const GemmParams& p, hipStream_t stream) {
    AMT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    AMT_CHECK_ARG(p.K % BK == 0, "gemm: K=%d must be a multiple of %d (pad the operands)", p.K, BK);
    AMT_CHECK_ARG(p.lda % 4 == 0 && p.ldw % 4 == 0, "gemm: leading dimensions must be multiples of 4 floats");
    AMT_CHECK_ARG(p.lda >= p.K && p.ldw >= p.K, "gemm: leading dimension smaller than K");
    AMT_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0, "gemm: operands must be 16-byte aligned");
    // Small products (one decode row, the 300 frames of one clip, a table, a few thousand rows of a narrow layer): a 128x128
    // tile costs its full MFMA time on one CU whatever part of it is real rows (27 us at K = 512; ~50 us measured), and with
    // fewer tiles than CUs nothing hides it.  With a plain epilogue they go to the skinny GEMM (16x16 output tiles, all of a
    // workgroup's weight loads in flight at once): ~10 us up to M*N = 256k outputs, break-even with the 64x64-tile variant below
    // near 650k outputs (tools/bench_small_gemm.py).
    const long small_m = amt_tuning().gemm_small_m, small_mn = amt_tuning().gemm_small_mn;
    if (p.M <= small_m && (long)p.M * p.N <= small_mn && p.K % 32 == 0 && p.K <= 1536 && !p.head_split && !p.rowadd && !p.silu_mul &&
        !p.sigmoid && !p.tile_group && !p.a_gather && p.relu != 2) {
        DecodeGemmParams g{};
        g.B = p.M; g.eps = 1e-5f; g.x = p.A; g.ldx = p.lda; g.Wp = p.W; g.ldw = p.ldw; g.bias = p.bias; g.N = p.N; g.K = p.K;
        g.resid = p.resid; g.ldr = p.ldr; g.relu = p.relu; g.scale = p.scale; g.scale_cols = p.scale_cols; g.y = p.C; g.ldy = p.ldc;
        return amt_launch_decode_gemm(g, stream);
    }
    // 64x64 tiles (one accumulator per wave, four times the workgroups) measure faster than 128x128 on every shape of this
    // model (K <= 1312; 114-126 TFLOP/s at M = 32768, 111 at M = 9600 since the loads of the k loop are branch-free and two
    // k-tiles ahead: tools/bench_gemm.py; 103-110 before); the big tile is kept for long-K products with many tiles.
    const int t128 = cdiv(p.M, 128) * cdiv(p.N, 128);
    const int t64_below = amt_tuning().gemm_t64_below;
    const int pf = amt_tuning().gemm_pf;                 // tens: big tile, units: small tile
    const bool big = t128 >= t64_below && p.K >= 2048;
    const dim3 g64(cdiv(p.M, 64) * cdiv(p.N, 64));
    if (p.a_gather) {
        if (big) hipLaunchKernelGGL((gemm_f32_kernel<2, 2, true, 1>), dim3(t128), dim3(256), 0, stream, p);
        else if (pf % 10 == 2) hipLaunchKernelGGL((gemm_f32_kernel<1, 1, true, 2>), g64, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<1, 1, true, 1>), g64, dim3(256), 0, stream, p);
    } else if (big) {
        if (pf / 10 == 2) hipLaunchKernelGGL((gemm_f32_kernel<2, 2, false, 2>), dim3(t128), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<2, 2, false, 1>), dim3(t128), dim3(256), 0, stream, p);
    } else {
        if (pf % 10 == 2) hipLaunchKernelGGL((gemm_f32_kernel<1, 1, false, 2>), g64, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<1, 1, false, 1>), g64, dim3(256), 0, stream, p);
    }
    AMT_LAUNCH_CHECK();
    return 0;
}
