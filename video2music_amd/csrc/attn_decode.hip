// Single-query attention for the autoregressive decode step (HBM-bound K/V streaming).
//
// One 512-thread workgroup per (clip, head).  The head's K and V rows ([cap][hd] fp32, contiguous)
// are streamed straight to registers with 16-byte lanes: hd/4 lanes cover one key row, so one
// wave-instruction fetches 64/(hd/4) whole rows = 1 KiB fully coalesced.  Each lane group keeps an
// online-softmax state (m, l, o[4]) for the keys it has seen; groups and waves are merged once at
// the end (shuffles, then LDS).  No score row is ever materialised.
//
// Relative-position self-attention (model/rpr.py:391-394 in closed form): for the
// query at position t and key j the score is q.k_j + q.Er[er_len-1-(t-j)], i.e. q.(k_j + e_j) with
// the Er row read from L2 (the table is shared by every clip and head of a layer).
// Cross-attention (torch MultiheadAttention at model/rpr.py:62-63) is the same kernel with Er=null
// and a fixed key count.
#include <stdlib.h>

#include "amt_common.h"
#include "kernels.h"
#include "sample_device.h"

namespace {

constexpr int NW = 8;            // waves per workgroup
constexpr int UNROLL = 4;        // key groups per wave and batch

template <int HD>
struct Batch {                   // one register-resident batch of UNROLL key groups
    float4 k[UNROLL], v[UNROLL], e[UNROLL];
};

__device__ __forceinline__ float4 ld4_stream(const float* p, bool nt) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    if (nt) { v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
    return ld4(p);
}

// kb / vb / eb are wave-uniform bases (SGPR pairs); the lane part is a 32-bit element offset, so a load is
// `global_load_dwordx4 v, v_off, s[base]` with one address VGPR and no 64-bit lane arithmetic
template <int HD, bool NT>
__device__ __forceinline__ void load_kv(Batch<HD>& bt, const float* kb, const float* vb, int j0, int sub, int c4, int n_keys) {
    constexpr int KPW = 64 / (HD / 4);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int j = j0 + u * NW * KPW + sub;
        const unsigned off = (unsigned)((j < n_keys ? j : 0) * HD + c4);
        bt.k[u] = ld4_stream(kb + off, NT);
        bt.v[u] = ld4_stream(vb + off, NT);
    }
}
template <int HD>
__device__ __forceinline__ void load_er(Batch<HD>& bt, const float* eb, int j0, int sub, int c4, int n_keys) {
    constexpr int KPW = 64 / (HD / 4);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int j = j0 + u * NW * KPW + sub;
        bt.e[u] = ld4(eb + (unsigned)((j < n_keys ? j : 0) * HD + c4));
    }
}
template <int HD, bool RPR, bool NT>
__device__ __forceinline__ void load_batch(Batch<HD>& bt, const float* kb, const float* vb, const float* eb,
                                           int j0, int sub, int c4, int n_keys) {
    load_kv<HD, NT>(bt, kb, vb, j0, sub, c4, n_keys);
    if (RPR) load_er<HD>(bt, eb, j0, sub, c4, n_keys);
}

template <int HD, bool RPR>
__device__ __forceinline__ void consume_batch(const Batch<HD>& bt, const float4 q4, int j0, int sub, int n_keys,
                                              float& m, float& l, float4& o) {
    constexpr int LPK = HD / 4, KPW = 64 / LPK;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        float4 k4 = bt.k[u];
        if (RPR) { k4.x += bt.e[u].x; k4.y += bt.e[u].y; k4.z += bt.e[u].z; k4.w += bt.e[u].w; }
        float s = q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
        s = group_sum<LPK>(s);
        if (j0 + u * NW * KPW + sub < n_keys) {
            const float mn = fmaxf(m, s);
            const float alpha = __expf(m - mn), pj = __expf(s - mn);
            l = l * alpha + pj;
            o.x = o.x * alpha + pj * bt.v[u].x; o.y = o.y * alpha + pj * bt.v[u].y;
            o.z = o.z * alpha + pj * bt.v[u].z; o.w = o.w * alpha + pj * bt.v[u].w;
            m = mn;
        }
    }
}

// The double-buffered key stream: `cur` holds a fetched batch, `nxt` receives the next one while `cur` is consumed.
// The next batch is fetched unconditionally (rows past the end are clamped to row 0 and never consumed): a guarded load
// hides the number of outstanding loads from the compiler, which then waits for *all* of them (vmcnt(0)) before the batch
// in hand is consumed, i.e. the double buffer degenerates to one batch in flight per wave.  The scheduling fence keeps the
// consumer's first instruction (which needs a wait) from being hoisted in front of the next batch's load instructions.
template <int HD, bool RPR, bool NT>
__device__ __forceinline__ void stream_keys(Batch<HD>& cur, Batch<HD>& nxt, const float* kb, const float* vb, const float* eb,
                                            const float4 q4, int j0, int sub, int c4, int n_keys, float& m, float& l, float4& o) {
    constexpr int STRIDE = NW * (64 / (HD / 4)) * UNROLL;
    while (j0 < n_keys) {
        load_batch<HD, RPR, NT>(nxt, kb, vb, eb, j0 + STRIDE, sub, c4, n_keys);
        __builtin_amdgcn_sched_barrier(0);
        consume_batch<HD, RPR>(cur, q4, j0, sub, n_keys, m, l, o);
        j0 += STRIDE;
        if (j0 >= n_keys) break;
        load_batch<HD, RPR, NT>(cur, kb, vb, eb, j0 + STRIDE, sub, c4, n_keys);
        __builtin_amdgcn_sched_barrier(0);
        consume_batch<HD, RPR>(nxt, q4, j0, sub, n_keys, m, l, o);
        j0 += STRIDE;
    }
}

constexpr int UCH_MAX = 4;       // folded prologue: the pre-LN row has at most UCH*256 floats (UCH = 2 or 4 by the model's width)

// FOLD: 0 plain query, 1 folded-LayerNorm prologue, 2 the same plus the new key/value of this position, 3 = 1 plus the rotary
// embedding of the query (the lockstep V1/V2 step: q = rope(LayerNorm(u) . Wq^T + b) * scale), 4 = 2 plus the rotary embedding of
// the query and of the new key (the V1/V2 self-attention behind a folded norm3; no relative-position table), 5 = the base model's
// layer-0 self-attention with the PREVIOUS step's sampling decision in its prologue (attn_decode_sample_kernel): wave 0 takes the
// decision of the folded output head for its clip (sample_device.h), every wave then sums its head's q / k / v of the new position
// from the projected input tables, head 0 stores the token and the next input row, the launch's last workgroup advances the
// position -- the sampling head's launch between two steps of a captured graph disappears
#ifdef AMT_STAMPS
#define ASTAMP(i) do { __builtin_amdgcn_sched_barrier(0); st_[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

template <int HD, bool RPR, bool NT, int FOLD, int UCH>
__device__ __forceinline__ void attn_decode_body(const AttnDecodeParams& p, const SampleParams* sp) {
#ifdef AMT_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ASTAMP(0);
    constexpr int LPK = HD / 4;          // lanes per key row
    constexpr int KPW = 64 / LPK;        // keys per wave-instruction
    constexpr int STRIDE = NW * KPW * UNROLL;     // keys per workgroup batch
    __shared__ float sm_m[NW], sm_l[NW];
    __shared__ __attribute__((aligned(16))) float sm_o[NW][HD];

    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % LPK, sub = lane / LPK;
    const int c4 = c * 4;
    const float* kb = p.k + ((size_t)b * p.H + h) * p.cap * HD;      // wave-uniform
    const float* vb = p.v + ((size_t)b * p.H + h) * p.cap * HD;

    // wave w takes key groups w, w+NW, ...; two batches are kept in flight (load i+1 before using i).
    // The first K/V batch and q do not depend on the step position: they are issued before `pos` is
    // read (rows past the current length are fetched but never used; they lie inside the cache).
    Batch<HD> b0, b1;
    int j0 = wave * KPW;
    // issued here: the prologue's stores would pin it behind them.  FOLD 5: *pos is the position the PREVIOUS step processed
    const int t = FOLD == 5 ? *p.pos + 1 : (p.pos ? *p.pos : (p.n_keys - 1));
    float4 q4, kn4 = make_float4(0.f, 0.f, 0.f, 0.f), vn4 = kn4;
    constexpr bool fresh = FOLD == 2 || FOLD == 4 || FOLD == 5;   // key/value of position t live in registers, not in the cache
    const int n_keys = fresh ? t : t + 1;
    const float* eb = RPR ? p.Er + (size_t)(p.er_len - 1 - t) * HD : nullptr;   // Er row of key 0 (wave-uniform)
    // Vector loads return in issue order.  The long prologue of FOLD 2 (11 loads and their address math) goes
    // behind the first K/V batch so that the stream starts at once; the short one of FOLD 1 goes in front of it
    // so that the statistics are computed while the batch is in flight (measured both ways).
    constexpr bool F1 = FOLD == 1 || FOLD == 3;
    __shared__ int s_ra[2];
    if constexpr (FOLD == 5) {
        // ---- the sampling head of the previous step (sample_fold_kernel's work, model/video_music_transformer.py:1070-1123), wave 0,
        // IN FRONT of its share of the key stream: vector loads return in issue order, behind two K/V batches the decision's rows
        // would land microseconds later; the other seven waves start streaming at once ----
        if (wave == 0) {
            const SampleParams& q = *sp;
            int root, attr;
            if (t >= q.n_primer) {
                const int tok = decide_fold_wave<UCH>(q, b, t - 1, lane, h == 0);
                feedback_of(q, tok, root, attr);
                if (h == 0 && lane == 0) {
                    q.tokens[(size_t)b * q.T + t] = tok;
                    q.roots[(size_t)b * q.T + t] = root;
                    q.attrs[(size_t)b * q.T + t] = attr;
                }
            } else {
                // inside the primer the token is given; the logits of the previous position are still owed to a caller who asked for them
                if (q.logits_out && h == 0) (void)decide_fold_wave<UCH>(q, b, t - 1, lane, true);
                root = (int)q.roots[(size_t)b * q.T + t];
                attr = (int)q.attrs[(size_t)b * q.T + t];
            }
            if (lane == 0) { s_ra[0] = root; s_ra[1] = attr; }
        }
    }
    if (!F1) load_kv<HD, NT>(b0, kb, vb, j0, sub, c4, p.cap);
    if (!FOLD) {
        q4 = ld4(p.q + ((size_t)b * p.H + h) * HD + c * 4);
    } else if constexpr (FOLD == 5) {
        // The decision (above) and the table rows behind it are two serial L2 round trips before the query exists: the key stream
        // runs meanwhile -- the Er rows of the first batch and the whole SECOND batch are requested here (rows past the end clamp to
        // row 0), so the HBM stream does not idle behind the prologue
        if (RPR) load_er<HD>(b0, eb, j0, sub, c4, n_keys);
        load_batch<HD, RPR, NT>(b1, kb, vb, eb, j0 + STRIDE, sub, c4, n_keys);
        const SampleParams& q = *sp;
        __syncthreads();
        const int root = s_ra[0], attr = s_ra[1];
        // this head's q / k / v of position t: the decoder input is a sum of table rows, so its projection is one too (the
        // summation order of write_next_input in sample.hip: ((TR[root] + TA[attr]) + key * tk) + TP[t])
        const int d = q.d, d3 = 3 * d, col = h * HD + c * 4;
        const float kv = q.key[b];
        const float* tr = q.tab_r + (size_t)root * d3 + col;
        const float* ta = q.tab_a + (size_t)attr * d3 + col;
        const float* tk = q.tab_k + col;
        const float* tp = q.tab_p + (size_t)t * d3 + col;
        const float4 r0 = ld4(tr), a0 = ld4(ta), k0 = ld4(tk), p0 = ld4(tp);
        const float4 r1 = ld4(tr + d), a1 = ld4(ta + d), k1 = ld4(tk + d), p1 = ld4(tp + d);
        const float4 r2 = ld4(tr + 2 * d), a2 = ld4(ta + 2 * d), k2 = ld4(tk + 2 * d), p2 = ld4(tp + 2 * d);
        __builtin_amdgcn_sched_barrier(0);
        q4.x = (((r0.x + a0.x) + kv * k0.x) + p0.x) * q.q_scale; q4.y = (((r0.y + a0.y) + kv * k0.y) + p0.y) * q.q_scale;
        q4.z = (((r0.z + a0.z) + kv * k0.z) + p0.z) * q.q_scale; q4.w = (((r0.w + a0.w) + kv * k0.w) + p0.w) * q.q_scale;
        kn4.x = ((r1.x + a1.x) + kv * k1.x) + p1.x; kn4.y = ((r1.y + a1.y) + kv * k1.y) + p1.y;
        kn4.z = ((r1.z + a1.z) + kv * k1.z) + p1.z; kn4.w = ((r1.w + a1.w) + kv * k1.w) + p1.w;
        vn4.x = ((r2.x + a2.x) + kv * k2.x) + p2.x; vn4.y = ((r2.y + a2.y) + kv * k2.y) + p2.y;
        vn4.z = ((r2.z + a2.z) + kv * k2.z) + p2.z; vn4.w = ((r2.w + a2.w) + kv * k2.w) + p2.w;
        if (h == 0) {                                 // the next input row x[t] (the residual stream layer 0 starts from)
            for (int cc = threadIdx.x * 4; cc < d; cc += NW * 64 * 4) {
                const float4 pr = ld4(q.PR + (size_t)root * d + cc), pa = ld4(q.PA + (size_t)attr * d + cc);
                const float4 wk = ld4(q.wkey + cc), bb = ld4(q.cbias + cc), pp = ld4(q.pe + (size_t)t * d + cc);
                float4 o4;
                o4.x = ((pr.x + pa.x) + kv * wk.x + bb.x) + pp.x; o4.y = ((pr.y + pa.y) + kv * wk.y + bb.y) + pp.y;
                o4.z = ((pr.z + pa.z) + kv * wk.z + bb.z) + pp.z; o4.w = ((pr.w + pa.w) + kv * wk.w + bb.w) + pp.w;
                st4(q.x_next + (size_t)b * d + cc, o4);
            }
        }
    } else {
        // LayerNorm folded through the projection: q = ((raw - mu*g) * rstd + c) * q_scale with the row
        // statistics of the pre-LN sum u[b] (every wave recomputes them: d floats from L2, two DPP reductions).
        const int d = p.d, col = h * HD + c * 4;
        const float* ub = p.fold_u + (size_t)b * d;
        float4 uv[UCH];
#pragma unroll
        for (int i = 0; i < UCH; ++i) {
            // unguarded (columns past d re-read the row's last float4 and are left out of the sums below): `k < d ? load : 0`
            // makes the compiler write the zero after the load and therefore wait for EVERY load in flight (vmcnt(0)) -- here the
            // whole first K/V batch -- before it issues the rest of the prologue's loads: two serial round trips
            const int k = min((i * 64 + lane) * 4, d - 4);
            uv[i] = ld4(ub + k);
        }
        const float* raw = p.q + (size_t)b * p.ldq + col;
        const float4 rq = ld4(raw), gq = ld4(p.fold_g + col), cq = ld4(p.fold_c + col);
        float4 rk = kn4, gk = kn4, ck = kn4, rv = kn4, gv = kn4, cv = kn4;
        if (fresh) {
            rk = ld4(raw + d); gk = ld4(p.fold_g + d + col); ck = ld4(p.fold_c + d + col);
            rv = ld4(raw + 2 * d); gv = ld4(p.fold_g + 2 * d + col); cv = ld4(p.fold_c + 2 * d + col);
        }
        float4 rcs = make_float4(1.f, 0.f, 1.f, 0.f);
        if (FOLD == 3) rcs = ld4(p.rope + (size_t)(*p.rope_pos) * p.rope_dim + (col % p.rope_dim));     // (cos, sin) of the lane's two pairs
        if (FOLD == 4) rcs = ld4(p.rope + (size_t)t * p.rope_dim + (col % p.rope_dim));
        if (F1) load_kv<HD, NT>(b0, kb, vb, j0, sub, c4, p.cap);
        // the first batch's Er rows need only the position: issued behind the prologue's loads, they land while the
        // statistics are reduced instead of costing one more L2 round trip after the query exists
        if (RPR && fresh) load_er<HD>(b0, eb, j0, sub, c4, n_keys);
        __builtin_amdgcn_sched_barrier(0);             // no consumer of a loaded value moves in front of the loads above
        const float inv_d = 1.0f / (float)d;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < UCH; ++i)
            if ((i * 64 + lane) * 4 < d) s += (uv[i].x + uv[i].y) + (uv[i].z + uv[i].w);
        const float mean = wave_sum(s) * inv_d;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < UCH; ++i) {
            if ((i * 64 + lane) * 4 < d) {
                const float dx = uv[i].x - mean, dy = uv[i].y - mean, dz = uv[i].z - mean, dw = uv[i].w - mean;
                qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float rstd = rsqrtf(wave_sum(qq) * inv_d + p.eps);
        if (FOLD == 3 || FOLD == 4) {
            // interleaved pairs (2i, 2i+1): even y = x*c - x'*s, odd y = x'*c + x*s (the skinny GEMM's rotary epilogue), then the scale
            const float x0 = (rq.x - mean * gq.x) * rstd + cq.x, x1 = (rq.y - mean * gq.y) * rstd + cq.y;
            const float x2 = (rq.z - mean * gq.z) * rstd + cq.z, x3 = (rq.w - mean * gq.w) * rstd + cq.w;
            q4.x = (x0 * rcs.x - x1 * rcs.y) * p.q_scale; q4.y = (x1 * rcs.x + x0 * rcs.y) * p.q_scale;
            q4.z = (x2 * rcs.z - x3 * rcs.w) * p.q_scale; q4.w = (x3 * rcs.z + x2 * rcs.w) * p.q_scale;
        } else {
            q4.x = ((rq.x - mean * gq.x) * rstd + cq.x) * p.q_scale; q4.y = ((rq.y - mean * gq.y) * rstd + cq.y) * p.q_scale;
            q4.z = ((rq.z - mean * gq.z) * rstd + cq.z) * p.q_scale; q4.w = ((rq.w - mean * gq.w) * rstd + cq.w) * p.q_scale;
        }
        if (fresh) {
            kn4.x = (rk.x - mean * gk.x) * rstd + ck.x; kn4.y = (rk.y - mean * gk.y) * rstd + ck.y;
            kn4.z = (rk.z - mean * gk.z) * rstd + ck.z; kn4.w = (rk.w - mean * gk.w) * rstd + ck.w;
            if (FOLD == 4) {                      // the key is rotated like the query (no scale)
                const float k0 = kn4.x, k1 = kn4.y, k2 = kn4.z, k3 = kn4.w;
                kn4.x = k0 * rcs.x - k1 * rcs.y; kn4.y = k1 * rcs.x + k0 * rcs.y;
                kn4.z = k2 * rcs.z - k3 * rcs.w; kn4.w = k3 * rcs.z + k2 * rcs.w;
            }
            vn4.x = (rv.x - mean * gv.x) * rstd + cv.x; vn4.y = (rv.y - mean * gv.y) * rstd + cv.y;
            vn4.z = (rv.z - mean * gv.z) * rstd + cv.z; vn4.w = (rv.w - mean * gv.w) * rstd + cv.w;
        }
        if (p.xn && h == 0 && wave == 0) {        // LayerNorm(u[b]): the residual of the following block
            // all affine vectors requested at once (unguarded, clamped), then the stores: a load under `k < d` per chunk made
            // this one wave of the clip walk UCH serial L2 round trips, and the workgroup's final barrier waits for it
            float4 w4[UCH], b4[UCH];
#pragma unroll
            for (int i = 0; i < UCH; ++i) {
                const int kc = min((i * 64 + lane) * 4, d - 4);
                w4[i] = ld4(p.fold_lnw + kc);
                b4[i] = ld4(p.fold_lnb + kc);
            }
#pragma unroll
            for (int i = 0; i < UCH; ++i) {
                const int k = (i * 64 + lane) * 4;
                if (k < d) {
                    float4 y;
                    y.x = (uv[i].x - mean) * rstd * w4[i].x + b4[i].x; y.y = (uv[i].y - mean) * rstd * w4[i].y + b4[i].y;
                    y.z = (uv[i].z - mean) * rstd * w4[i].z + b4[i].z; y.w = (uv[i].w - mean) * rstd * w4[i].w + b4[i].w;
                    st4(p.xn + (size_t)b * d + k, y);
                }
            }
        }
    }
#ifdef AMT_STAMPS
    if (q4.x == 1.2345e-30f) st_[7] = 1;          // stamp 1 = the query exists (prologue loads landed, statistics done)
#endif
    ASTAMP(1);
    if (fresh && wave == 0 && sub == 0) {
        st4(p.k_new + (((size_t)b * p.H + h) * p.cap + t) * HD + c * 4, kn4);
        st4(p.v_new + (((size_t)b * p.H + h) * p.cap + t) * HD + c * 4, vn4);
    }
    if (RPR && !fresh) load_er<HD>(b0, eb, j0, sub, c4, n_keys);
    // (fresh variants requested the first batch's Er rows inside their prologue)
    float m = -INFINITY, l = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (FOLD == 5) {
        // both batches are in flight since the prologue
        consume_batch<HD, RPR>(b0, q4, j0, sub, n_keys, m, l, o);
        j0 += STRIDE;
        stream_keys<HD, RPR, NT>(b1, b0, kb, vb, eb, q4, j0, sub, c4, n_keys, m, l, o);
    } else if (!RPR) {
        // cross-attention (fixed key count, always several batches): the first half-iteration is peeled, which keeps the
        // wait counts of the loop exact on both halves (measured 7.0 -> 6.8 us); for the self-attention the extra
        // unconditional batch at short lengths costs more than it gains (12.7 -> 13.1 us), so it enters the loop directly
        load_batch<HD, RPR, NT>(b1, kb, vb, eb, j0 + STRIDE, sub, c4, n_keys);
        __builtin_amdgcn_sched_barrier(0);
        consume_batch<HD, RPR>(b0, q4, j0, sub, n_keys, m, l, o);
        j0 += STRIDE;
        stream_keys<HD, RPR, NT>(b1, b0, kb, vb, eb, q4, j0, sub, c4, n_keys, m, l, o);
    } else {
        stream_keys<HD, RPR, NT>(b0, b1, kb, vb, eb, q4, j0, sub, c4, n_keys, m, l, o);
    }
    if (fresh && wave == 0) {                        // the current position's own key (relative distance 0)
        float4 k4 = kn4;
        if (RPR) {
            const float4 e = ld4(p.Er + (size_t)(p.er_len - 1) * HD + c * 4);
            k4.x += e.x; k4.y += e.y; k4.z += e.z; k4.w += e.w;
        }
        float s = q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
        s = group_sum<LPK>(s);
        if (sub == 0) {
            const float mn = fmaxf(m, s);
            const float alpha = __expf(m - mn), pj = __expf(s - mn);
            l = l * alpha + pj;
            o.x = o.x * alpha + pj * vn4.x; o.y = o.y * alpha + pj * vn4.y;
            o.z = o.z * alpha + pj * vn4.z; o.w = o.w * alpha + pj * vn4.w;
            m = mn;
        }
    }

#ifdef AMT_STAMPS
    if (l == 1.2345e-30f) st_[7] = 2;             // stamp 2 = this wave's keys are consumed
#endif
    ASTAMP(2);
    // merge the KPW lane groups of the wave (lanes with equal c)
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
        st4(&sm_o[wave][c * 4], o);
    }
    ASTAMP(3);
    __syncthreads();
    ASTAMP(4);
    if (wave == 0 && sub == 0) {
        float mn = sm_m[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mn = fmaxf(mn, sm_m[w]);
        float lt = 0.f;
        float4 ot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float a = (sm_m[w] == -INFINITY) ? 0.f : __expf(sm_m[w] - mn);
            const float4 ow = ld4(&sm_o[w][c * 4]);
            lt += sm_l[w] * a;
            ot.x += ow.x * a; ot.y += ow.y * a; ot.z += ow.z * a; ot.w += ow.w * a;
        }
        const float inv = 1.0f / lt;
        ot.x *= inv; ot.y *= inv; ot.z *= inv; ot.w *= inv;
        st4(p.o + ((size_t)b * p.H + h) * HD + c * 4, ot);
    }
#ifdef AMT_STAMPS
    ASTAMP(5);
    if (p.stamps && threadIdx.x == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 8; ++i) o[i] = st_[i];
    }
#endif
    if constexpr (FOLD == 5) {
        // every workgroup read *pos at its start; the last one to arrive publishes the position this step processes
        if (threadIdx.x == 0) {
            const unsigned n = __hip_atomic_fetch_add(sp->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n == gridDim.x * gridDim.y - 1) {
                *sp->pos = t;
                __hip_atomic_store(sp->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int HD, bool RPR, bool NT, int FOLD, int UCH>
// hd = 64 at d_model <= 512 (the benchmark's shape) is held at 128 VGPRs = two workgroups per CU: it fits without spilling and a
// launch of more than 256 workgroups (more than 32 clips per chain) then runs in one round (+6-7 % tokens/s at 64-256 clips); the
// other shapes keep the compiler's own choice (the same bound makes the hd = 16 / 32 relative-position variants spill)
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu((HD == 64 && UCH == 2) ? 4 : 1, (HD == 64 && UCH == 2) ? 4 : 8)))
void attn_decode_kernel(AttnDecodeParams p) {
    attn_decode_body<HD, RPR, NT, FOLD, UCH>(p, nullptr);
}

// layer 0 of the base model's folded chain with the previous step's decision in front (FOLD 5).  No 128-VGPR bound here: with the
// first K/V batch in flight the decision and the twelve table rows need more (the bound spilled 36 registers); at 32 clips the launch is
// one workgroup per CU either way, above that this ONE launch of the step's six self-attentions takes two rounds
template <int HD, bool RPR, bool NT, int UCH>
__global__ __launch_bounds__(NW * 64) void attn_decode_sample_kernel(AttnDecodeParams p, SampleParams sp) {
    attn_decode_body<HD, RPR, NT, 5, UCH>(p, &sp);
}

template <int HD, int FOLD, int UCH>
void launch_decode_u(const AttnDecodeParams& p, hipStream_t stream) {
    dim3 grid(p.H, p.B);
    // K/V are streamed once per launch and exceed the 256 MiB Infinity Cache per step: non-temporal loads keep
    // the step's re-used bytes (weights, activations) resident instead (measured +8 % tokens/s at
    // config 2).  AmtTuning::nt_mask: bit 0 = self-attention, bit 1 = cross-attention.
    const int nt_mask = amt_tuning().nt_mask;
    if (p.Er) {
        if constexpr (FOLD != 4) {               // (the rotary self-attention has no relative-position table: checked by the launcher)
            if (nt_mask & 1) hipLaunchKernelGGL((attn_decode_kernel<HD, true, true, FOLD, UCH>), grid, dim3(NW * 64), 0, stream, p);
            else hipLaunchKernelGGL((attn_decode_kernel<HD, true, false, FOLD, UCH>), grid, dim3(NW * 64), 0, stream, p);
        }
    } else {
        if (nt_mask & 2) hipLaunchKernelGGL((attn_decode_kernel<HD, false, true, FOLD, UCH>), grid, dim3(NW * 64), 0, stream, p);
        else hipLaunchKernelGGL((attn_decode_kernel<HD, false, false, FOLD, UCH>), grid, dim3(NW * 64), 0, stream, p);
    }
}

// the prologue keeps the pre-LayerNorm row in UCH float4 per lane: two for d_model <= 512 (config 2), four up to 1024
template <int HD, int FOLD>
void launch_decode(const AttnDecodeParams& p, hipStream_t stream) {
    if (FOLD == 0 || p.d <= 512) launch_decode_u<HD, FOLD, 2>(p, stream);
    else launch_decode_u<HD, FOLD, 4>(p, stream);
}

template <int HD>
void launch_decode_sample(const AttnDecodeParams& p, const SampleParams& sp, hipStream_t stream) {
    dim3 grid(p.H, p.B);
    const bool nt = (amt_tuning().nt_mask & 1) != 0;
#define AMT_LAUNCH_DS(RPR, NTV, UCHV) hipLaunchKernelGGL((attn_decode_sample_kernel<HD, RPR, NTV, UCHV>), grid, dim3(NW * 64), 0, stream, p, sp)
    if (sp.d <= 512) {
        if (p.Er) { if (nt) AMT_LAUNCH_DS(true, true, 2); else AMT_LAUNCH_DS(true, false, 2); }
        else { if (nt) AMT_LAUNCH_DS(false, true, 2); else AMT_LAUNCH_DS(false, false, 2); }
    } else {
        if (p.Er) { if (nt) AMT_LAUNCH_DS(true, true, 4); else AMT_LAUNCH_DS(true, false, 4); }
        else { if (nt) AMT_LAUNCH_DS(false, true, 4); else AMT_LAUNCH_DS(false, false, 4); }
    }
#undef AMT_LAUNCH_DS
}

}  // namespace

// The base model's layer-0 self-attention of a decode step with the previous step's sampling decision in its prologue (FOLD 5):
// p as for the plain layer-0 launch (k / v the cache, pos the device position) plus k_new / v_new; sp as amt_launch_sample takes it
// (folded head: lraw, h1..h4, projected tables).  *pos must hold the position the previous step processed.
int32_t amt_launch_attn_decode_sample(const AttnDecodeParams& p, const SampleParams& sp, hipStream_t stream) {
    AMT_CHECK_ARG(p.B > 0 && p.H > 0 && p.cap > 0 && p.pos && p.pos == sp.pos && p.k_new && p.v_new, "attn_decode_sample: bad attention arguments");
    AMT_CHECK_ARG(sp.lraw && sp.h1 && sp.h2 && sp.h3 && sp.h4 && sp.ln_w && sp.ln_b && sp.tab_r && sp.tab_a && sp.tab_k && sp.tab_p && sp.ticket &&
                  sp.tokens && sp.roots && sp.attrs && sp.x_next && !sp.sample_external && !sp.probs_out,
                  "attn_decode_sample: needs the folded head, the projected input tables and a device-side decision");
    AMT_CHECK_ARG(sp.B == p.B && sp.d == p.H * p.hd && sp.d % 4 == 0 && sp.d <= UCH_MAX * 256, "attn_decode_sample: shapes of the two halves differ");
    AMT_CHECK_ARG(p.Er == nullptr || p.er_len + 1 >= p.cap, "attn_decode_sample: er_len=%d smaller than the key capacity %d", p.er_len, p.cap);
    switch (p.hd) {
        case 16: launch_decode_sample<16>(p, sp, stream); break;
        case 32: launch_decode_sample<32>(p, sp, stream); break;
        case 64: launch_decode_sample<64>(p, sp, stream); break;
        case 128: launch_decode_sample<128>(p, sp, stream); break;
        default: AMT_CHECK_ARG(false, "attn_decode_sample: head_dim %d not in {16,32,64,128}", p.hd);
    }
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_attn_decode(const AttnDecodeParams& p, hipStream_t stream) {
    AMT_CHECK_ARG(p.B > 0 && p.H > 0 && p.cap > 0, "attn_decode: bad shape B=%d H=%d cap=%d", p.B, p.H, p.cap);
    AMT_CHECK_ARG(p.pos != nullptr || (p.n_keys > 0 && p.n_keys <= p.cap), "attn_decode: n_keys=%d outside (0,%d]", p.n_keys, p.cap);
    // (cap is the row count per (clip, head) in memory and may carry one padding row that keeps the heads off a power-of-two stride)
    AMT_CHECK_ARG(p.Er == nullptr || p.er_len + 1 >= p.cap, "attn_decode: er_len=%d smaller than the key capacity %d", p.er_len, p.cap);
    if (p.fold_u) {
        AMT_CHECK_ARG(p.fold_g && p.fold_c && p.d == p.H * p.hd && p.d % 4 == 0 && p.d <= UCH_MAX * 256 && p.ldq >= p.d && p.ldq % 4 == 0,
                      "attn_decode: bad folded prologue (d=%d ldq=%d)", p.d, p.ldq);
        AMT_CHECK_ARG(!p.xn || (p.fold_lnw && p.fold_lnb), "attn_decode: xn needs the LayerNorm affine");
        AMT_CHECK_ARG(!p.new_kv || (p.pos && p.k_new && p.v_new && p.ldq >= 3 * p.d), "attn_decode: new_kv needs pos, the cache and 3d raw columns");
        if (p.new_kv && p.rope) {
            AMT_CHECK_ARG(!p.Er && p.rope_dim > 0 && p.rope_dim % 4 == 0, "attn_decode: the rotary self-attention takes no relative-position table");
            switch (p.hd) {
                case 16: launch_decode<16, 4>(p, stream); break;
                case 32: launch_decode<32, 4>(p, stream); break;
                case 64: launch_decode<64, 4>(p, stream); break;
                case 128: launch_decode<128, 4>(p, stream); break;
                default: AMT_CHECK_ARG(false, "attn_decode: head_dim %d not in {16,32,64,128}", p.hd);
            }
        } else if (p.new_kv) {
            switch (p.hd) {
                case 16: launch_decode<16, 2>(p, stream); break;
                case 32: launch_decode<32, 2>(p, stream); break;
                case 64: launch_decode<64, 2>(p, stream); break;
                case 128: launch_decode<128, 2>(p, stream); break;
                default: AMT_CHECK_ARG(false, "attn_decode: head_dim %d not in {16,32,64,128}", p.hd);
            }
        } else if (p.rope) {
            AMT_CHECK_ARG(p.rope_pos && p.rope_dim > 0 && p.rope_dim % 4 == 0, "attn_decode: bad rotary prologue");
            switch (p.hd) {
                case 16: launch_decode<16, 3>(p, stream); break;
                case 32: launch_decode<32, 3>(p, stream); break;
                case 64: launch_decode<64, 3>(p, stream); break;
                case 128: launch_decode<128, 3>(p, stream); break;
                default: AMT_CHECK_ARG(false, "attn_decode: head_dim %d not in {16,32,64,128}", p.hd);
            }
        } else {
            switch (p.hd) {
                case 16: launch_decode<16, 1>(p, stream); break;
                case 32: launch_decode<32, 1>(p, stream); break;
                case 64: launch_decode<64, 1>(p, stream); break;
                case 128: launch_decode<128, 1>(p, stream); break;
                default: AMT_CHECK_ARG(false, "attn_decode: head_dim %d not in {16,32,64,128}", p.hd);
            }
        }
    } else {
        switch (p.hd) {
            case 16: launch_decode<16, 0>(p, stream); break;
            case 32: launch_decode<32, 0>(p, stream); break;
            case 64: launch_decode<64, 0>(p, stream); break;
            case 128: launch_decode<128, 0>(p, stream); break;
            default: AMT_CHECK_ARG(false, "attn_decode: head_dim %d not in {16,32,64,128}", p.hd);
        }
    }
    AMT_LAUNCH_CHECK();
    return 0;
}
