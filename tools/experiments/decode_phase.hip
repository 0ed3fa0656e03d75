// EXPERIMENT (round 2), measured and REJECTED — kept as evidence, not built into libamt_hip.so (results at the end of
// this comment and in profiles/r02_fused_phase_experiment.txt).
//
// Fused decode phase: the skinny GEMM of a decode-step phase and the attention that consumes it, in ONE launch.
//
// The decode step (model/rpr.py:55-70 per layer, one token for B <= 32 clips) is a chain of dependent launches that
// alternates latency-bound weight-streaming GEMMs with HBM-bound K/V-streaming attentions; run one after the other
// the memory system idles during every GEMM and the attention cannot start its stream before the GEMM has drained.
// K/V rows do not depend on the query, though.  So the workgroups of attention j+1 are part of the launch of GEMM j:
//
//   blocks [0, n_gemm)       16 x 16 output tiles of the GEMM (8 waves split K; A fragments straight from L2, no LDS
//                            staging; weights pre-packed in MFMA operand order, decode_gemm.hip), then ONE arrival on the
//                            counter of their 16-row block (outputs written through with sc1 stores, drained first);
//   blocks [n_gemm, ...)     one per (clip, head): pull the head's K/V rows into registers at once (cross-attention: all
//                            300 keys = 150 KiB per workgroup; self-attention: as many as fit), poll the counter of the
//                            clip's row block (one lane, sc1 loads, s_sleep; bounded), read q / the LayerNorm row with sc1
//                            loads, and only then do the arithmetic on resident data.
//
// Forward progress: GEMM tiles never wait; every CU can always take a GEMM workgroup next to one waiting attention
// workgroup (512 threads <= 128 VGPRs each, LDS well under half), and there are at most 256 attention workgroups
// for 256 CUs, so a tile can always be placed whatever the dispatch order.  The spin is bounded anyway: on a timeout the
// workgroup raises the error word (checked by amt_generate_end) instead of hanging the queue.
// The hand-off follows MI355X_MICROARCH.md "Valid forms": all handed-off bytes are stored sc1 and loaded sc1.
#include <mutex>

#include "decode_phase.h"

namespace {

constexpr int PW = 8;                    // waves per workgroup
constexpr int UCH = 2;                   // LayerNorm rows of at most UCH*256 floats (d_model <= 512)
constexpr unsigned long long SPIN_TICKS = 100ull * 1000;   // 1 ms of s_memrealtime (100 MHz)

#ifdef AMT_STAMPS
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); st_[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PSTAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PSTAMP_FLUSH() do { if (P.stamps && threadIdx.x == 0) { \
    st_[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
    for (int i_ = 0; i_ < 8; ++i_) P.stamps[(size_t)blockIdx.x * 8 + i_] = st_[i_]; } } while (0)
#else
#define PSTAMP(i) do { } while (0)
#define PSTAMP_DECL do { } while (0)
#define PSTAMP_FLUSH() do { } while (0)
#endif

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

// --------------------------------------------------------------------------------------------------------------------
// GEMM role.  TPW = k-tiles per wave (8: K <= 1024, 12: K <= 1536); PRO 0 plain rows, 2 folded-FFN prologue.
// --------------------------------------------------------------------------------------------------------------------
template <int TPW, int PRO>
__device__ __forceinline__ void gemm_role(const DecodePhaseParams& P, float* smem, const int wg, const int n_gemm_x) {
    const DecodeGemmParams& p = P.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wg % n_gemm_x, rb = wg / n_gemm_x, m0 = rb * 16;
    const int K = p.K, K1 = p.K1;
    const int nts = p.n_split >> 4;
    const bool high = nt >= nts;
    const int kt_n = (high ? K : K1) >> 4, tpw = (kt_n + PW - 1) / PW, kt0 = wave * tpw;
    const int r16 = lane & 15, kq = lane >> 4;
    const int arow = min(m0 + r16, p.B - 1);
    const float* xr = p.x + (size_t)arow * p.ldx + 4 * kq;
    const float* x2r = p.x2 + (size_t)arow * p.ldx2 + 4 * kq - K1;
    float* gs = smem;                      // PRO 2: [2][K]  g | gamma ,  c | beta
    float* st = smem + 2 * 1536;           // PRO 2: [16][2] mean, rstd of the row block's pre-LN rows
    float* red = smem + 2 * 1536 + 32;     // [PW][256] partial tiles
    PSTAMP_DECL;
    PSTAMP(0);

    // ---- every global load of the role, issued in one branch-free sequence (see decode_gemm.hip on why) ----
    const int du = K - K1;                 // width of the pre-LN sum u (the x2 part of the row)
    float4 uv[2][UCH];
    float4 gsv[2];
    if (PRO == 2) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float* ur = p.x2 + (size_t)min(m0 + 2 * wave + r, p.B - 1) * p.ldx2;
#pragma unroll
            for (int i = 0; i < UCH; ++i) uv[r][i] = ld4(ur + min((i * 64 + lane) * 4, du - 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {      // 2K floats = K/2 float4 <= 768: two per thread, the surplus repeats the last one
            const int f = min(j * 512 + tid, K / 2 - 1), vec = f >= K / 4, i = (f - vec * (K / 4)) * 4;
            const float* a = i < K1 ? p.fold_g + i : p.ln_w + (i - K1);
            const float* b = i < K1 ? p.fold_c + i : p.ln_b + (i - K1);
            gsv[j] = ld4(vec ? b : a);
        }
    }
    // With up to 8 k-tiles per wave every A fragment and weight tile is in flight from the start.  With 12 (K = 1536) that
    // would take more than the 128 VGPRs that let a GEMM and an attention workgroup share a CU, so the tiles go in two
    // halves, the second one issued once the first is in the accumulator: this launch is not the critical path of its
    // phase anyway (the attention part next to it needs longer for its K/V prefetch)
    constexpr int TH = TPW > 8 ? TPW / 2 : TPW;
    float4 af[TH], wf[TH];
    const float* wbase = high ? p.Wp2 + (size_t)(nt - nts) * kt_n * 256 : p.Wp + (size_t)nt * kt_n * 256;
    auto load_half = [&](int h0) {
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            const int k = min(kt0 + h0 + i, kt_n - 1) << 4;
            af[i] = ld4(k < K1 ? xr + k : x2r + k);
        }
#pragma unroll
        for (int i = 0; i < TH; ++i) wf[i] = ld4(wbase + ((size_t)min(kt0 + h0 + i, kt_n - 1) * 64 + lane) * 4);
    };
    load_half(0);
    // epilogue operands
    const int el = tid & 63, er = (tid >> 6) & 3;
    const int row = m0 + 4 * (el >> 4) + er, n = nt * 16 + (el & 15);
    const bool live = tid < 256 && row < p.B && n < p.N;
    const int rowc = min(row, p.B - 1), nc = min(n, p.N - 1);
    const float e_bias = high ? p.bias2[nc - p.n_split] : p.bias[nc];
    float e_res = 0.f;
    if (PRO == 2) e_res = p.x2[(size_t)rowc * p.ldx2 + min(nc, du - 1)];                // u[row][n]: LayerNorm'ed below
    else e_res = p.resid[(size_t)rowc * p.ldr + min(nc, p.n_split - 1)];
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(1);

    float mean = 0.f, rstd = 1.f;
    if (PRO == 2) {
        // statistics of rows 2*wave, 2*wave+1 of u; the per-column vectors through LDS
        const float inv_n = 1.0f / (float)du;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < UCH; ++i)
                if ((i * 64 + lane) * 4 < du) s += (uv[r][i].x + uv[r][i].y) + (uv[r][i].z + uv[r][i].w);
            const float mu = wave_sum(s) * inv_n;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < UCH; ++i)
                if ((i * 64 + lane) * 4 < du) {
                    const float dx = uv[r][i].x - mu, dy = uv[r][i].y - mu, dz = uv[r][i].z - mu, dw = uv[r][i].w - mu;
                    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            const float rs = rsqrtf(wave_sum(q) * inv_n + p.eps);
            if (lane == 0) { st[(2 * wave + r) * 2] = mu; st[(2 * wave + r) * 2 + 1] = rs; }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = min(j * 512 + tid, K / 2 - 1), vec = f >= K / 4, i = (f - vec * (K / 4)) * 4;
            st4(gs + vec * K + i, gsv[j]);
        }
        __syncthreads();
        mean = st[r16 * 2]; rstd = st[r16 * 2 + 1];
        // the LayerNorm half of the row is the residual of the low columns
        if (live && !high) {
            const float m2 = st[(row - m0) * 2], r2 = st[(row - m0) * 2 + 1];
            e_res = (e_res - m2) * r2 * gs[K1 + n] + gs[K + K1 + n];
        }
    }
    // [ relu((raw - mu*g)*rstd + c) | (u - mu)*rstd*gamma + beta ] applied to an A fragment in registers
    auto fix = [&](float4 a, int t) {
        const int k = (min(kt0 + t, kt_n - 1) << 4) + 4 * kq;
        const float4 g = ld4(gs + k), c = ld4(gs + K + k);
        if (k < K1) {
            a.x = fmaxf((a.x - mean * g.x) * rstd + c.x, 0.f); a.y = fmaxf((a.y - mean * g.y) * rstd + c.y, 0.f);
            a.z = fmaxf((a.z - mean * g.z) * rstd + c.z, 0.f); a.w = fmaxf((a.w - mean * g.w) * rstd + c.w, 0.f);
        } else {
            a.x = (a.x - mean) * rstd * g.x + c.x; a.y = (a.y - mean) * rstd * g.y + c.y;
            a.z = (a.z - mean) * rstd * g.z + c.z; a.w = (a.w - mean) * rstd * g.w + c.w;
        }
        return a;
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h0 = 0; h0 < TPW; h0 += TH) {
        if (h0 > 0) {
            __builtin_amdgcn_sched_barrier(0);
            load_half(h0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            const int t = h0 + i;
            const float4 a = PRO == 2 ? fix(af[i], t) : af[i];
            if (t < tpw && kt0 + t < kt_n) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wf[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wf[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wf[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wf[i].w, acc, 0, 0, 0);
            }
        }
    }
    float* rw = red + wave * 256;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) rw[rr * 64 + lane] = acc[rr];
    PSTAMP(2);
    __syncthreads();
    PSTAMP(3);
    const bool handoff = P.a.B > 0;
    if (live) {
        float val = 0.f;
#pragma unroll
        for (int w = 0; w < PW; ++w) val += red[w * 256 + tid];      // fixed order: deterministic
        val += e_bias;
        float* dst;
        if (high) dst = p.y2 + (size_t)row * p.ldy2 + (n - p.n_split);
        else { val += e_res; dst = p.y + (size_t)row * p.ldy + n; }
        if (handoff) st_sc1(dst, val); else *dst = val;
    }
    if (handoff) {
        // every storing wave drains its stores, the workgroup meets, ONE lane arrives (agent-scope relaxed add)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(P.sync + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PSTAMP(4);
    PSTAMP_FLUSH();
}

// --------------------------------------------------------------------------------------------------------------------
// attention roles
// --------------------------------------------------------------------------------------------------------------------
// waits until the `want` tiles of row block rb have arrived; returns false on a timeout (error word raised)
__device__ __forceinline__ bool wait_arrivals(unsigned* sync, int rb, unsigned want, int* flag_lds) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 0;
        for (;;) {
            if (__hip_atomic_load(sync + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = 1; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TICKS) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) __hip_atomic_store(sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag_lds = ok;
    }
    __syncthreads();
    return *flag_lds != 0;
}

// the last attention workgroup to leave clears the launch's counters (the next launch of this phase finds zeros)
__device__ __forceinline__ void leave(unsigned* sync, unsigned n_attn) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == n_attn - 1) {
            __hip_atomic_store(sync + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// 16-byte sc1 loads of handed-off rows: all of them issued, then one wait (also covers the K/V prefetch, which the
// arithmetic needs next anyway)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 f4(v4f t) { return make_float4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void ld4x3_sc1(const float* a, const float* b, const float* c, float4& ra, float4& rb, float4& rc) {
    v4f ta, tb, tc;
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(ta), "=&v"(tb), "=&v"(tc) : "v"(a), "v"(b), "v"(c) : "memory");
    ra = f4(ta); rb = f4(tb); rc = f4(tc);
}

// merge of the per-lane-group softmax states of a workgroup and the store of the head's output row (as attn_decode.hip)
template <int HD>
__device__ __forceinline__ void merge_and_store(float m, float l, float4 o, float* smem, float* out, int c, int sub, int wave) {
    constexpr int LPK = HD / 4;
    float* sm_m = smem;                // [PW]
    float* sm_l = smem + PW;           // [PW]
    float* sm_o = smem + 2 * PW;       // [PW][HD]
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
        float4 o2;
        o2.x = __shfl_xor(o.x, off, 64); o2.y = __shfl_xor(o.y, off, 64);
        o2.z = __shfl_xor(o.z, off, 64); o2.w = __shfl_xor(o.w, off, 64);
        const float mn = fmaxf(m, m2);
        const float a1 = (m == -INFINITY) ? 0.f : __expf(m - mn);
        const float a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
        l = l * a1 + l2 * a2;
        o.x = o.x * a1 + o2.x * a2; o.y = o.y * a1 + o2.y * a2;
        o.z = o.z * a1 + o2.z * a2; o.w = o.w * a1 + o2.w * a2;
        m = mn;
    }
    if (sub == 0) {
        if (c == 0) { sm_m[wave] = m; sm_l[wave] = l; }
        st4(&sm_o[wave * HD + c * 4], o);
    }
    __syncthreads();
    if (wave == 0 && sub == 0) {
        float mn = sm_m[0];
#pragma unroll
        for (int w = 1; w < PW; ++w) mn = fmaxf(mn, sm_m[w]);
        float lt = 0.f;
        float4 ot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < PW; ++w) {
            const float a = (sm_m[w] == -INFINITY) ? 0.f : __expf(sm_m[w] - mn);
            const float4 ow = ld4(&sm_o[w * HD + c * 4]);
            lt += sm_l[w] * a;
            ot.x += ow.x * a; ot.y += ow.y * a; ot.z += ow.z * a; ot.w += ow.w * a;
        }
        const float inv = 1.0f / lt;
        ot.x *= inv; ot.y *= inv; ot.z *= inv; ot.w *= inv;
        st4(out + c * 4, ot);
    }
}

// row statistics of the pre-LN sum u (d <= UCH*256 floats, loaded with sc1 by every wave) and the folded query
struct RowStats { float mean, rstd; };
__device__ __forceinline__ RowStats row_stats(const float4 (&uv)[UCH], int d, int lane, float eps) {
    const float inv_d = 1.0f / (float)d;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < UCH; ++i)
        if ((i * 64 + lane) * 4 < d) s += (uv[i].x + uv[i].y) + (uv[i].z + uv[i].w);
    const float mean = wave_sum(s) * inv_d;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < UCH; ++i)
        if ((i * 64 + lane) * 4 < d) {
            const float dx = uv[i].x - mean, dy = uv[i].y - mean, dz = uv[i].z - mean, dw = uv[i].w - mean;
            qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    return {mean, rsqrtf(wave_sum(qq) * inv_d + eps)};
}

// Cross-attention over the clip's video keys (torch MultiheadAttention at model/rpr.py:62-63; FOLD 1 of attn_decode.hip):
// all n_keys <= MAXR * PW * KPW rows of K and V are register-resident before the query exists.
template <int HD, int MAXR>
__device__ __forceinline__ void cross_attn_role(const DecodePhaseParams& P, float* smem, const int aid, const unsigned want) {
    const AttnDecodeParams& p = P.a;
    constexpr int LPK = HD / 4, KPW = 64 / LPK, RK = PW * KPW;      // keys per round of the workgroup
    const int h = aid % p.H, b = aid / p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % LPK, sub = lane / LPK, c4 = c * 4;
    const float* kb = p.k + ((size_t)b * p.H + h) * p.cap * HD;
    const float* vb = p.v + ((size_t)b * p.H + h) * p.cap * HD;
    const int n_keys = p.n_keys;
    PSTAMP_DECL;
    PSTAMP(0);
    // late start: let the GEMM tiles' own loads get into the memory queues first (pf_rows doubles as the delay, x 0.27 us)
    for (int i = 0; i < P.pf_rows; ++i) __builtin_amdgcn_s_sleep(10);
    float4 kr[MAXR], vr[MAXR];
    // paced: at most 16 loads (16 KiB) in flight per wave, as in attn_decode.hip's double-buffered stream -- issuing all 20 at
    // once (160 KiB per CU) collapsed the stream rate in the first version of this experiment
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int j = r * RK + wave * KPW + sub;
        const unsigned off = (unsigned)((j < n_keys ? j : 0) * HD + c4);
        if (r >= 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        kr[r] = ld4_nt(kb + off);
        vr[r] = ld4_nt(vb + off);
    }
    // query-independent operands of the folded prologue
    const int d = p.d, col = h * HD + c4;
    const float4 gq = ld4(p.fold_g + col), cq = ld4(p.fold_c + col);
    __builtin_amdgcn_sched_barrier(0);

    int* flag = reinterpret_cast<int*>(smem + 2 * PW + PW * HD);
    PSTAMP(1);
    const bool ok = wait_arrivals(P.sync, b >> 4, want, flag);
    PSTAMP(2);
    float4 uv[UCH], rq;
    {
        const float* ub = p.fold_u + (size_t)b * d;
        ld4x3_sc1(ub + min(lane * 4, d - 4), ub + min((64 + lane) * 4, d - 4), p.q + (size_t)b * p.ldq + col, uv[0], uv[1], rq);
    }
    PSTAMP(3);
    const RowStats rs = row_stats(uv, d, lane, p.eps);
    float4 q4;
    q4.x = ((rq.x - rs.mean * gq.x) * rs.rstd + cq.x) * p.q_scale; q4.y = ((rq.y - rs.mean * gq.y) * rs.rstd + cq.y) * p.q_scale;
    q4.z = ((rq.z - rs.mean * gq.z) * rs.rstd + cq.z) * p.q_scale; q4.w = ((rq.w - rs.mean * gq.w) * rs.rstd + cq.w) * p.q_scale;
    if (p.xn && h == 0 && wave == 0) {            // LayerNorm(u[b]): the residual of the following block
#pragma unroll
        for (int i = 0; i < UCH; ++i) {
            const int k = (i * 64 + lane) * 4;
            if (k < d) {
                const float4 w4 = ld4(p.fold_lnw + k), b4 = ld4(p.fold_lnb + k);
                float4 y;
                y.x = (uv[i].x - rs.mean) * rs.rstd * w4.x + b4.x; y.y = (uv[i].y - rs.mean) * rs.rstd * w4.y + b4.y;
                y.z = (uv[i].z - rs.mean) * rs.rstd * w4.z + b4.z; y.w = (uv[i].w - rs.mean) * rs.rstd * w4.w + b4.w;
                st4(p.xn + (size_t)b * d + k, y);
            }
        }
    }
    // exact two-pass softmax over the resident keys of this lane group
    float sc[MAXR];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        float s = q4.x * kr[r].x + q4.y * kr[r].y + q4.z * kr[r].z + q4.w * kr[r].w;
        s = group_sum<LPK>(s);
        const bool valid = r * RK + wave * KPW + sub < n_keys;
        sc[r] = valid ? s : -INFINITY;
        m = fmaxf(m, sc[r]);
    }
    float l = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const float pj = (sc[r] == -INFINITY) ? 0.f : __expf(sc[r] - m);
        l += pj;
        o.x += pj * vr[r].x; o.y += pj * vr[r].y; o.z += pj * vr[r].z; o.w += pj * vr[r].w;
    }
    PSTAMP(4);
    if (ok) merge_and_store<HD>(m, l, o, smem, p.o + ((size_t)b * p.H + h) * HD, c, sub, wave);
    PSTAMP(5);
    leave(P.sync, (unsigned)(p.B * p.H));
    PSTAMP(6);
    PSTAMP_FLUSH();
}

// L2 prefetch role: touch the leading rows of one (clip, head)'s K and V (contiguous [rows][hd] fp32 each) with plain
// 16-byte loads; starts ~1 us late so that the GEMM tiles' own loads are queued first
__device__ __forceinline__ void prefetch_role(const DecodePhaseParams& P, const int j) {
    __builtin_amdgcn_s_sleep(40);
    int rows = P.pf_rows;
    if (P.pf_pos) rows = min(rows, *P.pf_pos + P.pf_pos_add);
    const int n4 = rows * P.pf_row_floats / 4;                      // float4 per tensor
    const float4* kp = reinterpret_cast<const float4*>(P.pf_k + (size_t)j * P.pf_head_stride);
    const float4* vp = reinterpret_cast<const float4*>(P.pf_v + (size_t)j * P.pf_head_stride);
    for (int i = threadIdx.x; i < n4; i += PW * 64) {
        const float4 a = kp[i], b = vp[i];
        asm volatile("" :: "v"(a.x), "v"(b.x));                    // the values are not used: the loads are the point
    }
}

template <int HD, int MAXR, int TPW, int PRO>
__global__ __launch_bounds__(PW * 64, 4) void decode_phase_kernel(DecodePhaseParams P, int n_gemm_x, int n_gemm, int n_pad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wg = blockIdx.x;
    if (wg < n_gemm) {
        gemm_role<TPW, PRO>(P, smem, wg, n_gemm_x);
    } else if (P.pf_n > 0) {
        if (wg >= n_gemm + n_pad) prefetch_role(P, wg - n_gemm - n_pad);
    } else {
        cross_attn_role<HD, MAXR>(P, smem, wg - n_gemm, (unsigned)n_gemm_x);
    }
}

constexpr size_t LDS_BYTES = (2 * 1536 + 32 + PW * 256) * sizeof(float);      // 20.6 KB: GEMM role's vectors + partial tiles

template <int HD, int MAXR, int TPW, int PRO>
int32_t launch(const DecodePhaseParams& P, int n_gemm_x, int n_gemm, int n_attn, hipStream_t stream) {
    // prefetch blocks must keep the linear id of their (clip, head) modulo 8 (= the XCD under round-robin dispatch)
    const int n_pad = P.pf_n > 0 ? (8 - n_gemm % 8) % 8 : 0;
    hipLaunchKernelGGL((decode_phase_kernel<HD, MAXR, TPW, PRO>), dim3(n_gemm + n_attn + (P.pf_n > 0 ? n_pad + P.pf_n : 0)), dim3(PW * 64),
                       LDS_BYTES, stream, P, n_gemm_x, n_gemm, n_pad);
    AMT_LAUNCH_CHECK();
    return 0;
}

template <int HD, int MAXR>
int32_t launch_gemm_variant(const DecodePhaseParams& P, int n_gemm_x, int n_gemm, int n_attn, hipStream_t stream) {
    const bool big = P.g.K > 1024;
    if (P.g.pro == 1) return big ? launch<HD, MAXR, 12, 2>(P, n_gemm_x, n_gemm, n_attn, stream) : launch<HD, MAXR, 8, 2>(P, n_gemm_x, n_gemm, n_attn, stream);
    return big ? launch<HD, MAXR, 12, 0>(P, n_gemm_x, n_gemm, n_attn, stream) : launch<HD, MAXR, 8, 0>(P, n_gemm_x, n_gemm, n_attn, stream);
}

}  // namespace

bool amt_decode_phase_supported(int d, int dff, int hd, int scap) {
    return (hd == 32 || hd == 64) && d % 64 == 0 && d <= UCH * 256 && dff % 16 == 0 && 2 * d <= 1536 && dff + d <= 1536 && scap <= 320;
}

int32_t amt_launch_decode_phase(const DecodePhaseParams& P, hipStream_t stream) {
    const DecodeGemmParams& g = P.g;
    const AttnDecodeParams& a = P.a;
    AMT_CHECK_ARG(g.B > 0 && g.B <= 32, "decode_phase: B=%d outside 1..32", g.B);
    AMT_CHECK_ARG(g.mode == 0 && !g.sel && g.ldw == 0 && !g.ln_w == (g.pro != 1) && !g.relu && g.scale == 1.f, "decode_phase: GEMM part takes the folded chain's launches only");
    AMT_CHECK_ARG(g.x && g.x2 && g.K1 > 0 && g.K1 < g.K && g.K1 % 16 == 0 && g.K % 16 == 0 && g.K <= 1536 && g.K - g.K1 <= UCH * 256 && (g.K - g.K1) % 4 == 0,
                  "decode_phase: bad two-source rows K1=%d K=%d", g.K1, g.K);
    AMT_CHECK_ARG(g.ldx % 4 == 0 && g.ldx2 % 4 == 0 && g.ldx >= g.K1 && g.ldx2 >= g.K - g.K1, "decode_phase: bad row strides");
    AMT_CHECK_ARG(g.n_split > 0 && g.n_split % 16 == 0 && g.n_split < g.N && g.Wp && g.Wp2 && g.y && g.y2 && g.bias && g.bias2 && g.n_split <= g.K - g.K1,
                  "decode_phase: bad column split %d of N=%d", g.n_split, g.N);
    if (g.pro == 1) AMT_CHECK_ARG(g.fold_g && g.fold_c && g.ln_w && g.ln_b && !g.resid, "decode_phase: incomplete folded-FFN prologue");
    else AMT_CHECK_ARG(g.resid && g.ldr >= g.n_split, "decode_phase: the low columns need their residual");
    const int n_gemm_x = cdiv(g.N, 16), n_gemm = n_gemm_x * cdiv(g.B, 16);
    int n_attn = 0;
    AMT_CHECK_ARG(P.pf_n == 0 || (a.B == 0 && P.pf_k && P.pf_v && P.pf_rows > 0 && P.pf_row_floats % 4 == 0 && P.pf_head_stride % 4 == 0),
                  "decode_phase: bad prefetch part");
    if (a.B > 0) {
        AMT_CHECK_ARG(P.sync != nullptr, "decode_phase: the attention part needs the counters");
        AMT_CHECK_ARG(a.B == g.B && a.H > 0 && a.H * a.hd == a.d && a.d <= UCH * 256 && a.d % 4 == 0, "decode_phase: bad attention shape");
        AMT_CHECK_ARG(a.fold_u && a.fold_g && a.fold_c && a.ldq >= a.d && a.ldq % 4 == 0 && (!a.xn || (a.fold_lnw && a.fold_lnb)), "decode_phase: folded prologue operands missing");
        AMT_CHECK_ARG(!a.Er && !a.pos && !a.new_kv && a.n_keys > 0 && a.n_keys <= a.cap && a.n_keys <= 320, "decode_phase: cross-attention part: n_keys=%d", a.n_keys);
        n_attn = a.B * a.H;
    }
    const int hd = a.B > 0 ? a.hd : 64;
    if (hd == 64) return launch_gemm_variant<64, 10>(P, n_gemm_x, n_gemm, n_attn, stream);
    if (hd == 32) return launch_gemm_variant<32, 5>(P, n_gemm_x, n_gemm, n_attn, stream);
    AMT_CHECK_ARG(false, "decode_phase: head_dim %d not in {32,64}", hd);
    return -1;
}
