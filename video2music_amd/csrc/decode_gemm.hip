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
#include <stdlib.h>

#include <mutex>

#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int MAXB = 4096;       // rows per launch (16 per workgroup, blockIdx.y); the decode step uses 32
constexpr int XPAD = 8;

// (ldw: row stride of W in floats -- a column range [k0, k0 + K) of a wider matrix is packed as W + k0 with ldw = its full width)
__global__ void pack_weight_kernel(const float* __restrict__ W, float* __restrict__ P, int N, int K, int ldw) {
    const int kt_n = K / 16;
    const size_t tile = blockIdx.x;                 // nt*kt_n + kt
    const int nt = (int)(tile / kt_n), kt = (int)(tile % kt_n);
    const int lane = threadIdx.x;
    const int n = nt * 16 + (lane & 15), k = kt * 16 + 4 * (lane >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) v = ld4(W + (size_t)n * ldw + k);
    st4(P + (tile * 64 + lane) * 4, v);
}

constexpr int NW = 16;           // waves per workgroup: K is split 16 ways, one staged row per wave
constexpr int MT = 16;           // rows per workgroup (one MFMA row block); blockIdx.y selects the block

__device__ __forceinline__ float sum4(const float4 a) { return (a.x + a.y) + (a.z + a.w); }
// (the mean is splat into a full register pair: the packed subtract then has no half it does not define, which
// otherwise picks up a false dependency on whatever load targets the neighbouring register)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sq4(const float4 a, float m) {
    const f32x2 mm = {m, m};
    f32x2 lo = {a.x, a.y}, hi = {a.z, a.w};
    lo -= mm; hi -= mm;
    lo *= lo; hi *= hi;
    return (lo.x + lo.y) + (hi.x + hi.y);
}

// KCH = float4 chunks per lane and row (K <= KCH*256) = weight tiles per wave; FULL: K == KCH*256, no lane
// predicates; PRO: 0 plain rows, 1 LayerNorm prologue, 2 folded-FFN prologue, 3 gated-linear-unit prologue, 4 two LayerNorms,
// 5 folded gated FFN prologue (2 with a gate: up * silu(gate) in place of the ReLU).
//

// epilogue of a paired gate | up tile (DecodeGemmParams::glu_pair): lanes 0-7 of a 16-lane row hold the gate columns, lanes 8-15 the up
// columns of the same 8 hidden columns; every lane gets up * silu(gate), the up lanes store it
__device__ __forceinline__ float glu_pair_value(float val, int c) {
    const float other = __shfl_xor(val, 8, 64);
    const float g = c < 8 ? val : other, u = c < 8 ? other : val;
    return u * (g / (1.0f + __expf(-g)));
}
__device__ __forceinline__ float act(float v, int relu) { return relu == 1 ? fmaxf(v, 0.f) : relu == 2 ? v / (1.0f + __expf(-v)) : v; }
// column of the stacked bias that belongs to interleaved column n (N = stacked width)
__device__ __forceinline__ int glu_pair_bias_col(int n, int N) { const int c = n & 15; return (c < 8 ? 0 : N / 2 - 8) + (n >> 4) * 8 + c; }

// Everything the kernel reads from global memory is issued up front in ONE branch-free sequence: rows, prologue
// vectors, weight tiles, epilogue operands (out-of-range lanes read a clamped address and discard the value).
// Vector loads return in issue order and the compiler can only count them across straight-line code: with the
// rows first and no branch in between, the prologue waits for the (L2-resident) rows alone while the weight tiles
// (Infinity Cache / HBM) are still in flight.  A conditional load anywhere in that sequence degrades every later
// wait to vmcnt(0), i.e. serialises the prologue behind the weights.
#ifdef AMT_STAMPS
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); st_[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int KCH, bool FULL, int PRO>
__global__ __launch_bounds__(NW * 64) void decode_gemm_kernel(DecodeGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef AMT_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);
    const int K = p.K, LD = K + XPAD;
    float* xs = smem;                               // [16][LD]
    // [16 waves][4 r][64 lanes] partial tiles in a region of their own (behind the staged rows and the folded-FFN vectors): a wave stores its
    // partials as soon as its MFMAs are done, with no barrier in between (sharing the rows' region cost one: +0.6 % tokens/s without it)
    constexpr bool FFN = PRO == 2 || PRO == 5;       // folded-FFN prologues: per-column vectors handed over through LDS
    float* red = smem + MT * LD + (PRO == 2 ? 2 * K + 2 * MT : PRO == 5 ? 2 * K + 2 * p.K1 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, m0 = blockIdx.y * MT;
    // column split: tiles below n_split multiply the first K1 input columns by Wp, the others all K by Wp2
    const int nts = p.n_split >> 4;
    const bool high = p.n_split > 0 && nt >= nts;
    const int K1 = p.x2 ? p.K1 : K;                 // first column taken from x2
    const int Kw = (p.n_split > 0 && !high) ? p.K1 : K;
    const int kt_n = Kw / 16, tpw = (kt_n + NW - 1) / NW;
    const int kt0 = wave * tpw;

    // ---- 1. the row this wave stages: row m0+wave (clamped to the last valid row) ----
    const int zg = p.n_groups > 1 ? (int)blockIdx.z : 0;     // grouped launch: one expert per blockIdx.z
    const int r = m0 + wave, rc = min(r, p.B - 1);
    const float* xr = p.x + (size_t)zg * p.x_group_off + (size_t)rc * p.ldx;
    const float* x2r = p.x2 ? p.x2 + (size_t)rc * p.ldx2 - K1 : xr;
    float4 v[KCH];
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int i = (c * 64 + lane) * 4, ic = FULL ? i : min(i, K - 4);
        v[c] = ld4((ic < K1 ? xr : x2r) + ic);
    }
    // PRO 5: at least one 256-column chunk belongs to the u half (two at K = 1536, where the launcher asks for K - K1 >= 512: one
    // float4 more per lane would spill at the 128-VGPR budget of a 16-wave workgroup)
    constexpr int GCH = PRO == 5 ? (KCH == 6 ? 4 : KCH > 1 ? KCH - 1 : 1) : KCH;
    float4 gv[(PRO == 3 || PRO == 5) ? GCH : 1];
    if (PRO == 3) {
        const float* gr = p.glu_gate + (size_t)zg * p.x_group_off + (size_t)rc * p.ldx;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4, ic = FULL ? i : min(i, K - 4);
            gv[c] = ld4(gr + min(ic, K1 - 4));       // (two-source rows: the chunks past K1 are not gated and re-read the last gate chunk)
        }
    }
    if (PRO == 5) {                                   // raw gate columns of the first K1 staged columns (chunks past K1 re-read the last one)
        const float* gr = p.glu_gate + (size_t)rc * p.ldx;
#pragma unroll
        for (int c = 0; c < GCH; ++c) gv[c] = ld4(gr + min((c * 64 + lane) * 4, K1 - 4));
    }
    // ---- 2. prologue vectors ----
    constexpr bool LN1 = PRO == 1 || PRO == 4;      // PRO 4: two LayerNorms in a row (norm3 of the last layer, then decoder.norm)
    float4 g0[LN1 ? KCH : 1], h0[LN1 ? KCH : 1], g1[PRO == 4 ? KCH : 1], h1[PRO == 4 ? KCH : 1];
    if (LN1) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4, ic = FULL ? i : min(i, K - 4);
            g0[c] = ld4(p.ln_w + ic);
            h0[c] = ld4(p.ln_b + ic);
        }
    }
    if (PRO == 4) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4, ic = FULL ? i : min(i, K - 4);
            g1[c] = ld4(p.ln2_w + ic);
            h1[c] = ld4(p.ln2_b + ic);
        }
    }
    // folded FFN: the per-column vectors (same for all 16 rows) go through LDS, one float4 per thread (K/2 <= 1024)
    float4 gs_val = make_float4(0.f, 0.f, 0.f, 0.f);
    const int gs_t = min(tid, K / 2 - 1), gs_vec = gs_t >= K / 4, gs_i = (gs_t - gs_vec * (K / 4)) * 4;
    if (FFN) {
        const float* a = gs_i < K1 ? p.fold_g + gs_i : p.ln_w + (gs_i - K1);
        const float* b = gs_i < K1 ? p.fold_c + gs_i : p.ln_b + (gs_i - K1);
        gs_val = ld4(gs_vec ? b : a);
    }
    // folded gated FFN: the gate's two vectors [g2 (K1) | c2 (K1)], one more float4 for the first K1/2 threads (clamped)
    float4 gs2_val = make_float4(0.f, 0.f, 0.f, 0.f);
    const int gs2_t = min(tid, K1 / 2 - 1), gs2_vec = gs2_t >= K1 / 4, gs2_i = (gs2_t - gs2_vec * (K1 / 4)) * 4;
    if (PRO == 5) gs2_val = ld4((gs2_vec ? p.fold_c2 : p.fold_g2) + gs2_i);
    // ---- 3. this wave's weight tiles (tile index clamped: surplus loads repeat the last tile) ----
    float4 wt[KCH];
    const int grp = p.sel ? *p.sel : zg;            // device-chosen weight group (mixture-of-experts, one token) or the launch's group
    const float* wbase = high ? p.Wp2 + (size_t)(nt - nts) * kt_n * 256 : p.Wp + (size_t)grp * p.sel_w_stride + (size_t)nt * kt_n * 256;
    // un-packed weights (ldw > 0): lane (n = lane & 15, k-quad = lane >> 4) reads its float4 of row n directly: 16 rows x 64 B
    // per tile instead of one 1 KiB run, still one fully used 64-B segment per row
    const float* wrow = p.Wp + (size_t)min(nt * 16 + (lane & 15), p.N - 1) * p.ldw + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int kt = min(kt0 + i, kt_n - 1);
        wt[i] = ld4(p.ldw ? wrow + (size_t)kt * 16 : wbase + ((size_t)kt * 64 + lane) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);              // the scheduler must not sink any of these loads below this point
    STAMP(1);
    const int el = tid & 63, er = (tid >> 6) & 3;
    const int row = m0 + 4 * (el >> 4) + er, n = nt * 16 + (el & 15);
    const bool live = tid < 256 && row < p.B && n < p.N;
    // ---- prologue math ----
    if (PRO == 2) {
        // folded FFN, [ relu((raw - mu*g)*rstd + c) | LayerNorm(u) ] with the statistics of the u half (columns K1..K-1): the rows are
        // staged RAW, the wave leaves its row's (mu, rstd) and the per-column vectors in LDS, and the fix is applied to the A fragments
        // when they are read for the MFMAs -- under the shadow of the weight tiles still in flight, every element exactly once (a wave
        // owns its k range), and with ONE barrier instead of two (round 3: 1.56 us of a 6.5 us launch were this prologue)
        float* gs = smem + MT * LD;                 // [2][K]: g|gamma , c|beta ; then [16][2]: mu, rstd per staged row
        const float inv_n = 1.0f / (float)(K - K1);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            s += ((i >= K1 && (FULL || i < K)) ? 1.f : 0.f) * sum4(v[c]);
        }
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            q += ((i >= K1 && (FULL || i < K)) ? 1.f : 0.f) * sq4(v[c], mean);
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + p.eps);
        gs[2 * K + 2 * wave + (lane & 1)] = (lane & 1) ? rstd : mean;      // (every lane stores one of the two words: no branch)
        st4(gs + gs_vec * K + gs_i, gs_val);        // unconditional (surplus threads repeat the last float4): a guarded
                                                    // store lets the compiler sink the load behind the weight loads
    } else if (PRO == 5) {
        // [ up * silu(gate) | LayerNorm(u) ], both halves of the gated unit finished with the statistics of the u half (columns K1..K-1)
        float* gs = smem + MT * LD;                 // [2][K]: g|gamma , c|beta , then [2][K1]: the gate's g2 , c2
        st4(gs + gs_vec * K + gs_i, gs_val);
        st4(gs + 2 * K + gs2_vec * K1 + gs2_i, gs2_val);
        const float inv_n = 1.0f / (float)(K - K1);
        float um[KCH];                              // 1 on the u half, 0 elsewhere (a multiply keeps the loops branch-free)
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            um[c] = (i >= K1 && (FULL || i < K)) ? 1.f : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) s += um[c] * sum4(v[c]);
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) q += um[c] * sq4(v[c], mean);
        const float rstd = rsqrtf(wave_sum(q) * inv_n + p.eps);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            const int ic = FULL ? i : min(i, K - 4);
            const float4 g = ld4(gs + ic), h = ld4(gs + K + ic);
            // K1 is a multiple of 256 (checked by the launcher): a 256-column chunk lies on one side of it and the side is wave-uniform
            if (256 * c + 256 <= K1) {               // gated half: up * silu(gate)
                if (c < GCH) {
                    const float4 g2 = ld4(gs + 2 * K + ic), h2 = ld4(gs + 2 * K + K1 + ic);
                    const float ux = (v[c].x - mean * g.x) * rstd + h.x, uy = (v[c].y - mean * g.y) * rstd + h.y;
                    const float uz = (v[c].z - mean * g.z) * rstd + h.z, uw = (v[c].w - mean * g.w) * rstd + h.w;
                    const float tx = (gv[c].x - mean * g2.x) * rstd + h2.x, ty = (gv[c].y - mean * g2.y) * rstd + h2.y;
                    const float tz = (gv[c].z - mean * g2.z) * rstd + h2.z, tw = (gv[c].w - mean * g2.w) * rstd + h2.w;
                    v[c].x = ux * (tx / (1.0f + __expf(-tx))); v[c].y = uy * (ty / (1.0f + __expf(-ty)));
                    v[c].z = uz * (tz / (1.0f + __expf(-tz))); v[c].w = uw * (tw / (1.0f + __expf(-tw)));
                }
            } else {                                 // LayerNorm half: (u - mu)*rstd*gamma + beta
                v[c].x = (v[c].x - mean) * rstd * g.x + h.x; v[c].y = (v[c].y - mean) * rstd * g.y + h.y;
                v[c].z = (v[c].z - mean) * rstd * g.z + h.z; v[c].w = (v[c].w - mean) * rstd * g.w + h.w;
            }
        }
    } else if (LN1) {
        const float inv_k = 1.0f / (float)K;
#pragma unroll
        for (int pass = 0; pass < (PRO == 4 ? 2 : 1); ++pass) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int i = (c * 64 + lane) * 4;
                if (FULL || i < K) s += sum4(v[c]);
            }
            const float mean = wave_sum(s) * inv_k;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int i = (c * 64 + lane) * 4;
                if (FULL || i < K) q += sq4(v[c], mean);
            }
            const float rstd = rsqrtf(wave_sum(q) * inv_k + p.eps);
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const float4 g = (PRO == 4 && pass == 1) ? g1[c] : g0[c], h = (PRO == 4 && pass == 1) ? h1[c] : h0[c];
                v[c].x = (v[c].x - mean) * rstd * g.x + h.x; v[c].y = (v[c].y - mean) * rstd * g.y + h.y;
                v[c].z = (v[c].z - mean) * rstd * g.z + h.z; v[c].w = (v[c].w - mean) * rstd * g.w + h.w;
            }
        }
    }
    if (PRO == 3) {
        const float um = p.glu_only ? 0.f : 1.f, uo = p.glu_only ? 1.f : 0.f;      // h = u * silu(g), or silu(g) alone
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            // two-source rows [x * silu(gate) (K1 columns) | x2]: whole 256-column chunks on either side (checked by the launcher)
            if (p.x2 && 256 * c + 256 > K1) continue;
            v[c].x = (v[c].x * um + uo) * (gv[c].x / (1.0f + __expf(-gv[c].x))); v[c].y = (v[c].y * um + uo) * (gv[c].y / (1.0f + __expf(-gv[c].y)));
            v[c].z = (v[c].z * um + uo) * (gv[c].z / (1.0f + __expf(-gv[c].z))); v[c].w = (v[c].w * um + uo) * (gv[c].w / (1.0f + __expf(-gv[c].w)));
        }
    }
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (FULL || i < K) st4(xs + wave * LD + i, v[c]);    // (rows >= B hold a copy of row B-1; their outputs are never stored)
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);

    // ---- 4. epilogue operands: issued behind the prologue so that no register of the prologue's arithmetic sits
    // next to a pending load (packed VALU ops read register pairs).  Absent operands read a zero word instead of
    // being masked after the load: a select on a loaded value would be scheduled early and wait for the weights ----
    const float* bp = high ? p.bias2 : (p.bias ? p.bias + (size_t)grp * p.sel_b_stride : nullptr);      // (grp = blockIdx.z in a grouped launch)
    const bool has_b = live && bp != nullptr, has_r = !FFN && live && !high && p.mode == 0 && p.resid != nullptr;
    const float e_bias = *(has_b ? bp + (high ? n - p.n_split : p.glu_pair ? glu_pair_bias_col(n, p.N) : n) : p.zero);
    float e_res = 0.f;
    if (!FFN) e_res = *(has_r ? p.resid + (size_t)row * p.ldr + n : p.zero);
    const int t = *(p.pos ? p.pos : reinterpret_cast<const int*>(p.zero));

    // the normalised rows are the residual of the following block: workgroup (nt, m-block) publishes its
    // 16 rows x columns 16nt..16nt+15 -- and, when the product has fewer column tiles than the rows have 16-column blocks
    // (N < K: a feed-forward narrower than half the model width), the blocks one grid width further on as well
    if (LN1 && p.xn && tid < 256) {
        const int rr = tid >> 4, c = tid & 15;
        for (int cb = nt * 16; cb < K; cb += (int)gridDim.x * 16)
            if (m0 + rr < p.B) p.xn[(size_t)(m0 + rr) * K + cb + c] = xs[rr * LD + cb + c];
    }
    // folded FFN: the LayerNorm half of the staged row is the residual of the low columns
    if (PRO == 5 && live && !high) e_res = xs[(row - m0) * LD + K1 + n];
    const float* fgs = smem + MT * LD;              // PRO 2: the per-column vectors and the rows' statistics
    float f_mu = 0.f, f_rs = 0.f;
    if (PRO == 2) {
        f_mu = fgs[2 * K + 2 * (lane & 15)]; f_rs = fgs[2 * K + 2 * (lane & 15) + 1];       // A-fragment row of this lane
        if (live && !high) {
            const float* sr = fgs + 2 * K + 2 * (row - m0);
            e_res = fmaf(fmaf(xs[(row - m0) * LD + K1 + n], sr[1], -sr[0] * sr[1]), fgs[K1 + n], fgs[K + K1 + n]);      // (same spelling as the fragments')
        }
    }

    // ---- main: 4 MFMAs per k-tile ----
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* xa = xs + (lane & 15) * LD + kt0 * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        if (i < tpw && kt0 + i < kt_n) {
            float4 a0 = ld4(xa + i * 16);
            if (PRO == 2) {
                // the folded-FFN fix of this fragment.  It is VALU-bound (16 x 1536 elements per workgroup, every workgroup of the launch
                // redoes them: with (a - mu*g)*rstd + c spelled as mul / sub / mul / add / max the fix cost 0.75 us of a 7.5 us launch),
                // so each half is two fused multiply-adds per element, issued as packed pairs (v_pk_fma_f32), with mu*rstd per row:
                //   raw half        relu(a*rstd + (c - mu*rstd*g))        LayerNorm half   (a*rstd - mu*rstd)*gamma + beta
                // The side of K1 is wave-uniform (K1 % 16 == 0).
                const int kc = (kt0 + i) * 16 + 4 * (lane >> 4);
                const float4 g = ld4(fgs + kc), h = ld4(fgs + K + kc);
                const f32x2 a_lo = {a0.x, a0.y}, a_hi = {a0.z, a0.w}, g_lo = {g.x, g.y}, g_hi = {g.z, g.w}, h_lo = {h.x, h.y}, h_hi = {h.z, h.w};
                const f32x2 rs2 = {f_rs, f_rs}, nm2 = {-f_mu * f_rs, -f_mu * f_rs};
                if ((kt0 + i) * 16 < K1) {
                    const f32x2 t_lo = __builtin_elementwise_fma(a_lo, rs2, __builtin_elementwise_fma(nm2, g_lo, h_lo));
                    const f32x2 t_hi = __builtin_elementwise_fma(a_hi, rs2, __builtin_elementwise_fma(nm2, g_hi, h_hi));
                    a0 = make_float4(fmaxf(t_lo.x, 0.f), fmaxf(t_lo.y, 0.f), fmaxf(t_hi.x, 0.f), fmaxf(t_hi.y, 0.f));
                } else {
                    const f32x2 t_lo = __builtin_elementwise_fma(__builtin_elementwise_fma(a_lo, rs2, nm2), g_lo, h_lo);
                    const f32x2 t_hi = __builtin_elementwise_fma(__builtin_elementwise_fma(a_hi, rs2, nm2), g_hi, h_hi);
                    a0 = make_float4(t_lo.x, t_lo.y, t_hi.x, t_hi.y);
                }
            }
            const float4 w = wt[i];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w.w, acc, 0, 0, 0);
        }
    }
    // ---- cross-wave reduction in fixed order ----
#ifdef AMT_STAMPS
    if (acc[0] == 1.2345e-30f) st_[7] = 1;          // the stamp must follow the MFMA results, not just their issue
#endif
    STAMP(4);
    float* rw = red + wave * 256;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) rw[rr * 64 + lane] = acc[rr];
    __syncthreads();
    STAMP(5);
    float val = 0.f;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < NW; ++w) val += red[w * 256 + tid];
        val += e_bias;
    }
    if (p.rope) {
        // interleaved pairs sit in adjacent lanes of the same row (16-column tiles): even column y = x*c - x'*s, odd y = x*c + x'*s
        const float other = __shfl_xor(val, 1, 64);
        if (n < p.rope_cols) {
            const int e2 = (n % p.rope_dim) & ~1;
            const float* rp = p.rope + (size_t)t * p.rope_dim + e2;
            const float c = rp[0], sn = rp[1];
            val = (n & 1) ? val * c + other * sn : val * c - other * sn;
        }
    }
    if (p.glu_pair) {                                // (outside `live`: the exchange takes every lane of the wave)
        val = glu_pair_value(val, n & 15);
        if (live && (n & 8)) p.y[(size_t)row * p.ldy + (n >> 4) * 8 + (n & 7)] = val;
    } else if (live) {
        float* ybase = p.y + (size_t)zg * p.y_group_off;
        if (high) {
            p.y2[(size_t)row * p.ldy2 + (n - p.n_split)] = val;
        } else {
            if (n < p.scale_cols) val *= p.scale;
            if (p.mode == 0) {
                val += e_res;
                val = act(val, p.relu);
                ybase[(size_t)row * p.ldy + n] = val;
            } else if (n < p.d) {
                ybase[(size_t)row * p.ldy + n] = val;
            } else {
                const int nn = (n < 2 * p.d) ? n - p.d : n - 2 * p.d;
                const int hh = nn / p.hd, cc = nn - hh * p.hd;
                float* dst = (n < 2 * p.d) ? p.kcache : p.vcache;
                dst[(((size_t)row * p.H + hh) * p.cap + t) * p.hd + cc] = val;
            }
        }
    }
#ifdef AMT_STAMPS
    STAMP(6);
    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 8; ++i) o[i] = st_[i];
    }
#endif
}

template <int KCH, bool FULL, int PRO>
int32_t launch_one(const DecodeGemmParams& p, size_t lds, hipStream_t stream) {
    // > 64 KiB of dynamic LDS needs the opt-in (gfx950: 160 KiB per CU).  The attribute belongs to the (function, device)
    // pair, so the flag is kept per device ordinal and per instantiation, under a lock (host threads may launch concurrently)
    static bool attr_set[64] = {false};
    static std::mutex mu;
    int dev = 0;
    AMT_HIP(hipGetDevice(&dev));
    AMT_CHECK_ARG(dev >= 0 && dev < 64, "decode_gemm: device ordinal %d out of range", dev);
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            AMT_HIP(hipFuncSetAttribute((const void*)decode_gemm_kernel<KCH, FULL, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL((decode_gemm_kernel<KCH, FULL, PRO>), dim3(cdiv(p.N, 16), cdiv(p.B, MT), p.n_groups > 1 ? p.n_groups : 1), dim3(NW * 64), lds, stream, p);
    AMT_LAUNCH_CHECK();
    return 0;
}

// Wide products (N >= 4096: the stacked gate|up projection of a mixture-of-experts layer, 7 x 2 x dff columns): with one 16-column
// tile per workgroup such a launch is 1792 workgroups of two weight tiles per wave each, i.e. several rounds of the fixed
// stage-rows / barrier / reduce latency over the chip (16.3 us for 29 MB at config 2).  Here a workgroup keeps its 16 staged rows
// and walks NTW column tiles: all NTW x KCH weight tiles of a wave are in flight before the prologue, the rows are staged and
// normalised once, and every thread takes part in the final reduction (thread group j reduces tile j).  Same arithmetic per output
// as decode_gemm_kernel (same k order inside a wave, same fixed wave order in the reduction): results are bit-identical.
// Plain single-source products only: PRO 0 / 1, packed weights, mode 0, no column split, no rotary epilogue.
template <int KCH, int PRO, int NTW>
__global__ __launch_bounds__(NW * 64) void decode_gemm_wide_kernel(DecodeGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = KCH * 256, LD = K + XPAD, kt_n = K / 16;
    float* xs = smem;                               // [16][LD]
    float* red = smem + MT * LD;                    // [NTW][16 waves][256] partial tiles behind the staged rows (no barrier before they are stored)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid: the row blocks of one tile group are 8 linear ids apart, i.e. dispatched back to back onto the SAME XCD (round-robin
    // placement), so the group's weight tiles cross HBM / the Infinity Cache once and the second row block finds them in that L2
    const int mb = (p.B + MT - 1) / MT, n_tiles = (p.N + 15) >> 4;
    const int grp = ((int)blockIdx.x / (8 * mb)) * 8 + ((int)blockIdx.x & 7);
    if (grp * NTW >= n_tiles) return;               // the grid is padded to whole sets of 8 groups (uniform per workgroup)
    const int nt0 = grp * NTW, m0 = (((int)blockIdx.x >> 3) % mb) * MT;
    const int kt0 = wave * KCH;
    const int zg = p.n_groups > 1 ? (int)blockIdx.y : 0;       // grouped launch: one expert per blockIdx.y
    const int r = m0 + wave, rc = min(r, p.B - 1);
    const float* xr = p.x + (size_t)zg * p.x_group_off + (size_t)rc * p.ldx;
    float4 v[KCH];
#pragma unroll
    for (int c = 0; c < KCH; ++c) v[c] = ld4(xr + (c * 64 + lane) * 4);
    float4 gv[PRO == 3 ? KCH : 1];
    if (PRO == 3) {
        const float* gr = p.glu_gate + (size_t)zg * p.x_group_off + (size_t)rc * p.ldx;
#pragma unroll
        for (int c = 0; c < KCH; ++c) gv[c] = ld4(gr + (c * 64 + lane) * 4);
    }
    float4 g0[PRO == 1 ? KCH : 1], h0[PRO == 1 ? KCH : 1];
    if (PRO == 1) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            g0[c] = ld4(p.ln_w + (c * 64 + lane) * 4);
            h0[c] = ld4(p.ln_b + (c * 64 + lane) * 4);
        }
    }
    float4 wt[NTW][KCH];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const float* wbase = p.Wp + (size_t)zg * p.sel_w_stride + (size_t)min(nt0 + j, n_tiles - 1) * kt_n * 256;      // surplus tiles repeat the last one
#pragma unroll
        for (int i = 0; i < KCH; ++i) wt[j][i] = ld4(wbase + ((size_t)(kt0 + i) * 64 + lane) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PRO == 1) {
        const float inv_k = 1.0f / (float)K;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) s += sum4(v[c]);
        const float mean = wave_sum(s) * inv_k;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) q += sq4(v[c], mean);
        const float rstd = rsqrtf(wave_sum(q) * inv_k + p.eps);
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            v[c].x = (v[c].x - mean) * rstd * g0[c].x + h0[c].x; v[c].y = (v[c].y - mean) * rstd * g0[c].y + h0[c].y;
            v[c].z = (v[c].z - mean) * rstd * g0[c].z + h0[c].z; v[c].w = (v[c].w - mean) * rstd * g0[c].w + h0[c].w;
        }
    }
    if (PRO == 3) {
        const float um = p.glu_only ? 0.f : 1.f, uo = p.glu_only ? 1.f : 0.f;      // h = u * silu(g), or silu(g) alone
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            v[c].x = (v[c].x * um + uo) * (gv[c].x / (1.0f + __expf(-gv[c].x))); v[c].y = (v[c].y * um + uo) * (gv[c].y / (1.0f + __expf(-gv[c].y)));
            v[c].z = (v[c].z * um + uo) * (gv[c].z / (1.0f + __expf(-gv[c].z))); v[c].w = (v[c].w * um + uo) * (gv[c].w / (1.0f + __expf(-gv[c].w)));
        }
    }
#pragma unroll
    for (int c = 0; c < KCH; ++c) st4(xs + wave * LD + (c * 64 + lane) * 4, v[c]);
    __syncthreads();

    // thread (jg, e) owns output e of tiles nt0 + jg, nt0 + jg + 4, ...
    constexpr int JN = (NTW + 3) / 4;
    const int jg = tid >> 8, e = tid & 255;
    const int el = e & 63, er = (e >> 6) & 3;
    const int row = m0 + 4 * (el >> 4) + er;
    float e_bias[JN], e_res[JN];
#pragma unroll
    for (int q = 0; q < JN; ++q) {
        const int j = jg + 4 * q, n = (nt0 + j) * 16 + (el & 15);
        const bool live = j < NTW && row < p.B && n < p.N;
        e_bias[q] = *((live && p.bias) ? p.bias + (size_t)zg * p.sel_b_stride + (p.glu_pair ? glu_pair_bias_col(n, p.N) : n) : p.zero);
        e_res[q] = *((live && p.resid) ? p.resid + (size_t)row * p.ldr + n : p.zero);
        if (PRO == 1 && p.xn && j < NTW && (nt0 + j) * 16 < K) {
            const int rr = e >> 4, c = e & 15;
            if (m0 + rr < p.B) p.xn[(size_t)(m0 + rr) * K + (nt0 + j) * 16 + c] = xs[rr * LD + (nt0 + j) * 16 + c];
        }
    }

    f32x4 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xa = xs + (lane & 15) * LD + kt0 * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const float4 a0 = ld4(xa + i * 16);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const float4 w = wt[j][i];
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w.x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w.y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w.z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w.w, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        float* rw = red + (j * NW + wave) * 256;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) rw[rr * 64 + lane] = acc[j][rr];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < JN; ++q) {
        const int j = jg + 4 * q, n = (nt0 + j) * 16 + (el & 15);
        if (j < NTW) {
            float val = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) val += red[(j * NW + w) * 256 + e];
            val += e_bias[q];
            if (p.glu_pair) {                        // (j < NTW is uniform per 256-thread group: whole waves take the exchange)
                val = glu_pair_value(val, n & 15);
                if (row < p.B && n < p.N && (n & 8)) p.y[(size_t)row * p.ldy + (n >> 4) * 8 + (n & 7)] = val;
            } else if (row < p.B && n < p.N) {
                if (n < p.scale_cols) val *= p.scale;
                val += e_res[q];
                val = act(val, p.relu);
                p.y[(size_t)zg * p.y_group_off + (size_t)row * p.ldy + n] = val;
            }
        }
    }
}

// The wide product for up to 32 rows in ONE workgroup (RB2): both 16-row blocks share the workgroup's weight tiles, so a tile
// group's weights cross L2 -> L1 once instead of once per row block, and the launch is half as many workgroups -- the stacked
// gate | up product of a mixture layer at 32 clips (1792 column tiles, K = 512) becomes 224 workgroups = ONE round over the chip
// instead of 448 = two (13.6 -> 11.3 us in the lockstep step's trace; lockstep V2 generate 97.6 -> 96.2 ms alternating in one
// process, profiles/r03_v2_ab.json).  Wave w stages rows w and 16 + w;
// per k-tile a weight fragment feeds two MFMA chains; the partial tiles of the two row blocks go through the same LDS region one
// after the other.  Same arithmetic per output as decode_gemm_wide_kernel (bit-identical results).  K = KCH * 256 <= 512.
template <int KCH, int PRO, int NTW>
__global__ __launch_bounds__(NW * 64) void decode_gemm_wide2_kernel(DecodeGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = KCH * 256, LD = K + XPAD, kt_n = K / 16;
    float* xs = smem;                               // [2][16][LD]
    float* red = smem + 2 * MT * LD;                // [NTW][16 waves][256] partial tiles of ONE row block at a time
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (p.N + 15) >> 4;
    const int grp = (int)blockIdx.x;
    if (grp * NTW >= n_tiles) return;
    const int nt0 = grp * NTW;
    const int kt0 = wave * KCH;
    float4 v[2][KCH];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const float* xr = p.x + (size_t)min(rb * MT + wave, p.B - 1) * p.ldx;
#pragma unroll
        for (int c = 0; c < KCH; ++c) v[rb][c] = ld4(xr + (c * 64 + lane) * 4);
    }
    float4 g0[PRO == 1 ? KCH : 1], h0[PRO == 1 ? KCH : 1];
    if (PRO == 1) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            g0[c] = ld4(p.ln_w + (c * 64 + lane) * 4);
            h0[c] = ld4(p.ln_b + (c * 64 + lane) * 4);
        }
    }
    float4 wt[NTW][KCH];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const float* wbase = p.Wp + (size_t)min(nt0 + j, n_tiles - 1) * kt_n * 256;      // surplus tiles repeat the last one
#pragma unroll
        for (int i = 0; i < KCH; ++i) wt[j][i] = ld4(wbase + ((size_t)(kt0 + i) * 64 + lane) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PRO == 1) {
        const float inv_k = 1.0f / (float)K;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) s += sum4(v[rb][c]);
            const float mean = wave_sum(s) * inv_k;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) q += sq4(v[rb][c], mean);
            const float rstd = rsqrtf(wave_sum(q) * inv_k + p.eps);
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                v[rb][c].x = (v[rb][c].x - mean) * rstd * g0[c].x + h0[c].x; v[rb][c].y = (v[rb][c].y - mean) * rstd * g0[c].y + h0[c].y;
                v[rb][c].z = (v[rb][c].z - mean) * rstd * g0[c].z + h0[c].z; v[rb][c].w = (v[rb][c].w - mean) * rstd * g0[c].w + h0[c].w;
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < KCH; ++c) st4(xs + (rb * MT + wave) * LD + (c * 64 + lane) * 4, v[rb][c]);
    __syncthreads();

    // thread (jg, e) owns output e of tiles nt0 + jg, nt0 + jg + 4, ... of each row block
    constexpr int JN = (NTW + 3) / 4;
    const int jg = tid >> 8, e = tid & 255;
    const int el = e & 63, er = (e >> 6) & 3;
    const int rloc = 4 * (el >> 4) + er;
    float e_bias[JN];
#pragma unroll
    for (int q = 0; q < JN; ++q) {
        const int j = jg + 4 * q, n = (nt0 + j) * 16 + (el & 15);
        e_bias[q] = *((j < NTW && n < p.N && p.bias) ? p.bias + (p.glu_pair ? glu_pair_bias_col(n, p.N) : n) : p.zero);
        if (PRO == 1 && p.xn && j < NTW && (nt0 + j) * 16 < K) {
            const int rr = e >> 4, c = e & 15;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                if (rb * MT + rr < p.B) p.xn[(size_t)(rb * MT + rr) * K + (nt0 + j) * 16 + c] = xs[(rb * MT + rr) * LD + (nt0 + j) * 16 + c];
        }
    }

    f32x4 acc[2][NTW];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xa = xs + (lane & 15) * LD + kt0 * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const float4 a0 = ld4(xa + i * 16), a1 = ld4(xa + MT * LD + i * 16);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const float4 w = wt[j][i];
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w.x, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w.y, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w.z, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w.w, acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w.x, acc[1][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w.y, acc[1][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w.z, acc[1][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w.w, acc[1][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        if (rb) __syncthreads();                    // the first row block's partial tiles are consumed
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            float* rw = red + (j * NW + wave) * 256;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) rw[rr * 64 + lane] = acc[rb][j][rr];
        }
        __syncthreads();
        const int row = rb * MT + rloc;
#pragma unroll
        for (int q = 0; q < JN; ++q) {
            const int j = jg + 4 * q, n = (nt0 + j) * 16 + (el & 15);
            if (j < NTW) {
                float val = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) val += red[(j * NW + w) * 256 + e];
                val += e_bias[q];
                if (p.glu_pair) {
                    val = glu_pair_value(val, n & 15);
                    if (row < p.B && n < p.N && (n & 8)) p.y[(size_t)row * p.ldy + (n >> 4) * 8 + (n & 7)] = val;
                } else if (row < p.B && n < p.N) {
                    if (n < p.scale_cols) val *= p.scale;
                    if (p.resid) val += p.resid[(size_t)row * p.ldr + n];
                    val = act(val, p.relu);
                    p.y[(size_t)row * p.ldy + n] = val;
                }
            }
        }
    }
}

template <int KCH, int PRO, int NTW>
int32_t launch_wide2(const DecodeGemmParams& p, hipStream_t stream) {
    static bool attr_set[64] = {false};
    static std::mutex mu;
    int dev = 0;
    AMT_HIP(hipGetDevice(&dev));
    AMT_CHECK_ARG(dev >= 0 && dev < 64, "decode_gemm: device ordinal %d out of range", dev);
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            AMT_HIP(hipFuncSetAttribute((const void*)decode_gemm_wide2_kernel<KCH, PRO, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = true;
        }
    }
    constexpr size_t lds = (size_t)2 * MT * (KCH * 256 + XPAD) * sizeof(float) + (size_t)NTW * NW * 256 * sizeof(float);      // rows of both blocks | partial tiles
    static_assert(lds <= 160 * 1024, "rows + partial tiles must fit the 160 KiB of LDS");
    hipLaunchKernelGGL((decode_gemm_wide2_kernel<KCH, PRO, NTW>), dim3(cdiv(cdiv(p.N, 16), NTW)), dim3(NW * 64), lds, stream, p);
    AMT_LAUNCH_CHECK();
    return 0;
}

template <int KCH, int PRO, int NTW>
int32_t launch_wide(const DecodeGemmParams& p, hipStream_t stream) {
    static bool attr_set[64] = {false};
    static std::mutex mu;
    int dev = 0;
    AMT_HIP(hipGetDevice(&dev));
    AMT_CHECK_ARG(dev >= 0 && dev < 64, "decode_gemm: device ordinal %d out of range", dev);
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            AMT_HIP(hipFuncSetAttribute((const void*)decode_gemm_wide_kernel<KCH, PRO, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = true;
        }
    }
    const size_t lds = (size_t)MT * (KCH * 256 + XPAD) * sizeof(float) + (size_t)NTW * NW * 256 * sizeof(float);      // rows | partial tiles
    static_assert((size_t)MT * (KCH * 256 + XPAD) * sizeof(float) + (size_t)NTW * NW * 256 * sizeof(float) <= 160 * 1024, "rows + partial tiles must fit the 160 KiB of LDS");
    const int groups8 = cdiv(cdiv(cdiv(p.N, 16), NTW), 8) * 8;
    hipLaunchKernelGGL((decode_gemm_wide_kernel<KCH, PRO, NTW>), dim3(groups8 * cdiv(p.B, MT), p.n_groups > 1 ? p.n_groups : 1), dim3(NW * 64), lds, stream, p);
    AMT_LAUNCH_CHECK();
    return 0;
}

template <int KCH, bool FULL>
int32_t launch_variant(const DecodeGemmParams& p, size_t lds, hipStream_t stream) {
#ifdef AMT_EXPERIMENT
    // timing only (wrong values): what does the folded-FFN prologue cost?  exp_a = 1 launches the same product with the plain staging
    if (p.pro == 1 && amt_tuning().exp_a == 1) return launch_one<KCH, FULL, 0>(p, lds, stream);
#endif
    if (p.pro == 1) return launch_one<KCH, FULL, 2>(p, lds, stream);
    if (p.pro == 2) { if constexpr (FULL && KCH >= 2) return launch_one<KCH, FULL, 5>(p, lds, stream); }
    if (p.glu_gate) return launch_one<KCH, FULL, 3>(p, lds, stream);
    if (p.ln_w && p.ln2_w) { if constexpr (KCH <= 4) return launch_one<KCH, FULL, 4>(p, lds, stream); }
    if (p.ln_w) { if constexpr (KCH <= 4) return launch_one<KCH, FULL, 1>(p, lds, stream); }
    return launch_one<KCH, FULL, 0>(p, lds, stream);
}

}  // namespace

int32_t amt_launch_pack_weight(const float* W, float* P, int N, int K, hipStream_t stream, int ldw) {
    AMT_CHECK_ARG(K % 16 == 0 && (ldw == 0 || (ldw >= K && ldw % 4 == 0)), "pack_weight: K=%d must be a multiple of 16 (ldw=%d)", K, ldw);
    const int tiles = cdiv(N, 16) * (K / 16);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(tiles), dim3(64), 0, stream, W, P, N, K, ldw ? ldw : K);
    AMT_LAUNCH_CHECK();
    return 0;
}

// a few zero words in global memory per device (absent bias / residual / position read these)
static int32_t zero_words(const float** out) {
    static float* buf[64] = {nullptr};
    static std::mutex mu;                            // several host threads may launch concurrently (one stream each)
    int dev = 0;
    AMT_HIP(hipGetDevice(&dev));
    AMT_CHECK_ARG(dev >= 0 && dev < 64, "decode_gemm: device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!buf[dev]) {
        AMT_HIP(hipMalloc((void**)&buf[dev], 256));
        AMT_HIP(hipMemset(buf[dev], 0, 256));
    }
    *out = buf[dev];
    return 0;
}

int32_t amt_decode_gemm_init() { const float* z; return zero_words(&z); }

int32_t amt_launch_decode_gemm(const DecodeGemmParams& p_in, hipStream_t stream) {
    DecodeGemmParams p = p_in;
    if (int32_t zrc = zero_words(&p.zero)) return zrc;
    AMT_CHECK_ARG(p.B > 0 && p.B <= MAXB, "decode_gemm: B=%d outside (0,%d]", p.B, MAXB);
    AMT_CHECK_ARG(p.K % 32 == 0 && p.K <= 1536, "decode_gemm: K=%d must be a multiple of 32 and <= 1536", p.K);
    AMT_CHECK_ARG(p.N > 0 && p.ldx % 4 == 0, "decode_gemm: bad N/ldx");
    AMT_CHECK_ARG(p.mode == 0 || (p.kcache && p.vcache && p.d > 0 && p.N == 3 * p.d && p.d == p.H * p.hd), "decode_gemm: bad QKV epilogue");
    if (p.x2) AMT_CHECK_ARG(p.K1 > 0 && p.K1 < p.K && p.K1 % 16 == 0 && p.ldx >= p.K1 && p.ldx2 >= p.K - p.K1 && p.ldx2 % 4 == 0,
                            "decode_gemm: bad two-source split K1=%d of K=%d", p.K1, p.K);
    else AMT_CHECK_ARG(p.ldx >= p.K, "decode_gemm: ldx=%d < K=%d", p.ldx, p.K);
    if (p.n_split) AMT_CHECK_ARG(p.x2 && p.n_split % 16 == 0 && p.n_split <= p.N && p.mode == 0 && (p.n_split == p.N || (p.Wp2 && p.y2)),
                                 "decode_gemm: bad column split %d of N=%d", p.n_split, p.N);
    AMT_CHECK_ARG(!p.sel || p.n_split == 0, "decode_gemm: a device-selected weight group cannot be combined with a column split");
    AMT_CHECK_ARG(p.ldw == 0 || (p.ldw >= p.K && p.ldw % 4 == 0 && p.n_split == 0 && !p.sel), "decode_gemm: bad un-packed weight (ldw=%d)", p.ldw);
    AMT_CHECK_ARG(p.n_groups <= 1 || (!p.sel && p.n_split == 0 && !p.x2 && p.ldw == 0 && !p.resid && p.mode == 0 && p.n_groups <= 65535), "decode_gemm: a grouped launch takes plain single-source products");
    AMT_CHECK_ARG(!p.glu_gate || p.pro == 2 || (!p.ln_w && p.pro == 0 && (!p.x2 || (p.K1 % 256 == 0 && p.K % 256 == 0 && p.n_groups <= 1 && !p.glu_only))),
                  "decode_gemm: the gated prologue takes no LayerNorm, and a second source only behind whole 256-column chunks");
    AMT_CHECK_ARG(!p.glu_pair || (p.N % 16 == 0 && p.mode == 0 && !p.resid && !p.rope && p.n_split == 0 && !p.sel && p.n_groups <= 1 && p.scale_cols == 0 &&
                                  !p.relu && p.ldy >= p.N / 2), "decode_gemm: the paired gate | up epilogue takes a plain product of N %% 16 == 0 columns");
    AMT_CHECK_ARG(!p.rope || (p.pos && p.rope_dim > 0 && p.rope_dim % 2 == 0 && p.rope_cols % 2 == 0 && p.n_split == 0), "decode_gemm: bad rotary epilogue");
    if (p.pro == 2) {
        AMT_CHECK_ARG(p.x2 && p.glu_gate && p.fold_g && p.fold_c && p.fold_g2 && p.fold_c2 && p.ln_w && p.ln_b && !p.resid && p.K1 % 256 == 0 &&
                      (p.K == 512 || p.K == 768 || p.K == 1024 || (p.K == 1536 && p.K - p.K1 >= 512)) && p.n_groups <= 1 && !p.sel && p.ldw == 0,
                      "decode_gemm: incomplete folded gated-FFN prologue (K1=%d K=%d)", p.K1, p.K);
    } else if (p.pro == 1) {
        AMT_CHECK_ARG(p.x2 && p.fold_g && p.fold_c && p.ln_w && p.ln_b && !p.resid, "decode_gemm: incomplete folded-FFN prologue");
    } else {
        AMT_CHECK_ARG(!p.ln_w || (!p.x2 && p.K <= 1024), "decode_gemm: the LayerNorm prologue takes a single source of K <= 1024");
    }
    AMT_CHECK_ARG(!p.ln2_w || (p.ln_w && p.ln_b && p.ln2_b && p.pro == 0 && !p.glu_gate), "decode_gemm: the second LayerNorm follows a first one");
    // wide plain products: several column tiles per workgroup (see decode_gemm_wide_kernel)
    // the grouped down projections of a mixture layer (gated prologue): more workgroups than the chip holds at once with one
    // tile each, one round with two
    const int wide_grp = amt_tuning().wide_grouped;
    if (wide_grp > 0 && p.n_groups > 1 && !p.ln_w && p.pro == 0 && !p.rope && !p.x2 && p.mode == 0 && p.ldw == 0 && !p.sel &&
        (p.K == 512 || p.K == 1024) && cdiv(p.N, 16) * cdiv(p.B, MT) * p.n_groups > 256) {
        if (p.glu_gate) return p.K == 512 ? launch_wide<2, 3, 2>(p, stream) : launch_wide<4, 3, 2>(p, stream);
        return p.K == 512 ? launch_wide<2, 0, 2>(p, stream) : launch_wide<4, 0, 2>(p, stream);      // (gated already: the producer's paired epilogue)
    }
    const int wide_ntw = amt_tuning().wide_ntw;
    if (wide_ntw > 0 && p.N >= 4096 && !p.x2 && p.n_split == 0 && !p.sel && p.n_groups <= 1 && p.mode == 0 && !p.rope && !p.glu_gate && p.pro == 0 &&
        p.ldw == 0 && (p.K == 512 || p.K == 1024) && (!p.ln_w || p.ln_b) && !p.ln2_w) {
        const bool ln = p.ln_w != nullptr;
        // 17 .. 32 rows: both row blocks in one workgroup (half the workgroups, the weights through L1 once)
        if (p.K == 512 && p.B > MT && p.B <= 2 * MT && amt_tuning().wide_rb2) return ln ? launch_wide2<2, 1, 4>(p, stream) : launch_wide2<2, 0, 4>(p, stream);
        if (p.K == 512) {
            if (wide_ntw == 7) return ln ? launch_wide<2, 1, 7>(p, stream) : launch_wide<2, 0, 7>(p, stream);
            return ln ? launch_wide<2, 1, 4>(p, stream) : launch_wide<2, 0, 4>(p, stream);
        }
        return ln ? launch_wide<4, 1, 4>(p, stream) : launch_wide<4, 0, 4>(p, stream);
    }
    // staged rows | folded-FFN vectors | the waves' partial tiles
    const size_t lds = (size_t)MT * (p.K + XPAD) * sizeof(float) + (p.pro == 1 ? (size_t)(2 * p.K + 2 * MT) * sizeof(float) : 0) +
                       (p.pro == 2 ? (size_t)(2 * p.K + 2 * p.K1) * sizeof(float) : 0) + (size_t)NW * 256 * sizeof(float);
    int32_t rc;
    switch (p.K) {
        case 256: rc = launch_variant<1, true>(p, lds, stream); break;
        case 512: rc = launch_variant<2, true>(p, lds, stream); break;
        case 768: rc = launch_variant<3, true>(p, lds, stream); break;
        case 1024: rc = launch_variant<4, true>(p, lds, stream); break;
        case 1536: rc = launch_variant<6, true>(p, lds, stream); break;
        default:
            if (p.K < 256) rc = launch_variant<1, false>(p, lds, stream);
            else if (p.K < 1024) rc = launch_variant<4, false>(p, lds, stream);
            else rc = launch_variant<6, false>(p, lds, stream);
    }
    if (rc) return rc;
    AMT_LAUNCH_CHECK();
    return 0;
}
