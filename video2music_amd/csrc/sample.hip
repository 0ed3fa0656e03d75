// Output head of one decode step: final LayerNorms -> Wout -> softmax -> decision -> feedback.
//
// One 1024-thread workgroup per clip.  Restates, on device and without host round trips, the body
// of the reference's generate loop (model/video_music_transformer.py:1070-1131):
//   y = softmax(logits)[:157]                      (END/PAD dropped, not renormalised, :1070-1071)
//   beam==1 : tok = top-1(y); root/attr sequences are left at PAD                     (:1078-1084)
//   beam==0 : y[0]=0 if max_conseq_N==0; y[prev]=0 if the last max_conseq_chord ids are equal;
//             tok = argmax(y / sum(y))  (Categorical.sample replaced by arg-max = oracle G2);
//             (root, attr) = (0,1) for N else ((tok-1)/13+1, (tok-1)%13+1)             (:1085-1123)
// and then builds the next step's decoder input
//   x[t+1] = Linear_chord([E_root[root]+E_attr[attr], key]) + pe[t+1]                  (:984-1001,1029)
// from the tables PR = E_root.Wc[:, :d]^T, PA = E_attr.Wc[:, :d]^T precomputed at weight load.
// The last workgroup to finish advances the device-side position counter that every kernel of
// the captured step graph reads, so one graph replays for all steps.
#include "amt_common.h"
#include "kernels.h"
#include "sample_device.h"

namespace {

constexpr int NWS = 16;                 // waves of the sampling workgroup

__device__ __forceinline__ void write_next_input(const SampleParams& p, int b, int cur, int root, int attr) {
    const float kv = p.key[b];
    for (int c = threadIdx.x * 4; c < p.d; c += blockDim.x * 4) {
        const float4 pr = ld4(p.PR + (size_t)root * p.d + c), pa = ld4(p.PA + (size_t)attr * p.d + c);
        const float4 wk = ld4(p.wkey + c), bb = ld4(p.cbias + c), pp = ld4(p.pe + (size_t)cur * p.d + c);
        float4 o;
        o.x = ((pr.x + pa.x) + kv * wk.x + bb.x) + pp.x;
        o.y = ((pr.y + pa.y) + kv * wk.y + bb.y) + pp.y;
        o.z = ((pr.z + pa.z) + kv * wk.z + bb.z) + pp.z;
        o.w = ((pr.w + pa.w) + kv * wk.w + bb.w) + pp.w;
        st4(p.x_next + (size_t)b * p.d + c, o);
    }
    if (p.tab_r) {          // layer 0's q / k / v of position `cur` (same summation order as x above)
        const int d = p.d, d3 = 3 * p.d;
        for (int c = threadIdx.x * 4; c < d3; c += blockDim.x * 4) {
            const float4 tr = ld4(p.tab_r + (size_t)root * d3 + c), ta = ld4(p.tab_a + (size_t)attr * d3 + c);
            const float4 tk = ld4(p.tab_k + c), tp = ld4(p.tab_p + (size_t)cur * d3 + c);
            float4 o;
            o.x = ((tr.x + ta.x) + kv * tk.x) + tp.x; o.y = ((tr.y + ta.y) + kv * tk.y) + tp.y;
            o.z = ((tr.z + ta.z) + kv * tk.z) + tp.z; o.w = ((tr.w + ta.w) + kv * tk.w) + tp.w;
            if (c < d) {
                o.x *= p.q_scale; o.y *= p.q_scale; o.z *= p.q_scale; o.w *= p.q_scale;
                st4(p.q0 + (size_t)b * d + c, o);
            } else {
                const int nn = c < 2 * d ? c - d : c - 2 * d;
                const int hh = nn / p.hd, cc = nn - hh * p.hd;
                st4((c < 2 * d ? p.kc0 : p.vc0) + (((size_t)b * p.H + hh) * p.cap + cur) * p.hd + cc, o);
            }
        }
    }
}

__device__ __forceinline__ void advance_pos(const SampleParams& p, int t) {
    __syncthreads();
    if (threadIdx.x == 0) {
        // every block has read *pos before it takes a ticket; the last one publishes t+1
        const unsigned n = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == gridDim.x - 1) {
            *p.pos = t + 1;
            __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int KCH>       // float4 chunks per lane covering d (d <= KCH*256)
__global__ __launch_bounds__(NWS * 64) void sample_kernel(SampleParams p) {
    __shared__ __attribute__((aligned(16))) float ys[1024];
    __shared__ float logit[V + 1];
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = p.d;
    const int t = *p.pos, cur = t + 1;

    // Wout rows of the first pass do not depend on the LayerNorm: put them in flight now
    constexpr int RB = KCH <= 2 ? 10 : (KCH == 3 ? 5 : 4);    // rows per wave and pass (16 waves x 10 rows cover the 159 outputs at once)
    float4 wv[RB][KCH];
    float bo[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int n = wave + j * NWS;
        bo[j] = n < V ? p.bout[n] : 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            wv[j][c] = (n < V && i < d) ? ld4(p.Wout + (size_t)n * d + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // ---- LayerNorm(s) of the row, by wave 0 (d <= 1024: up to 4 float4 per lane) ----
    if (wave == 0) {
        const float inv_d = 1.0f / (float)d;
        float4 v[KCH];
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            v[c] = (i < d) ? ld4(p.u + (size_t)b * p.ldu + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const float* gw = pass == 0 ? p.ln_w : p.fn_w;
            const float* gb = pass == 0 ? p.ln_b : p.fn_b;
            if (!gw) continue;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) s += v[c].x + v[c].y + v[c].z + v[c].w;
            const float mean = wave_sum(s) * inv_d;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int i = (c * 64 + lane) * 4;
                if (i < d) {
                    const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
                    q += dx * dx + dy * dy + dz * dz + dw * dw;
                }
            }
            const float rstd = rsqrtf(wave_sum(q) * inv_d + p.eps);
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int i = (c * 64 + lane) * 4;
                if (i < d) {
                    const float4 g = ld4(gw + i), h = ld4(gb + i);
                    v[c].x = (v[c].x - mean) * rstd * g.x + h.x; v[c].y = (v[c].y - mean) * rstd * g.y + h.y;
                    v[c].z = (v[c].z - mean) * rstd * g.z + h.z; v[c].w = (v[c].w - mean) * rstd * g.w + h.w;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            if (i < d) st4(&ys[i], v[c]);
        }
    }
    __syncthreads();

    // ---- logits = y . Wout^T + b : wave w takes rows w, w+16, ... ----
    {
        float4 yv[KCH];
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            yv[c] = (i < d) ? ld4(&ys[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int n0 = wave; n0 < V; n0 += NWS * RB) {
            if (n0 != wave) {                         // later passes (d > 512 only): reload
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const int n = n0 + j * NWS;
                    bo[j] = n < V ? p.bout[n] : 0.f;
#pragma unroll
                    for (int c = 0; c < KCH; ++c) {
                        const int i = (c * 64 + lane) * 4;
                        wv[j][c] = (n < V && i < d) ? ld4(p.Wout + (size_t)n * d + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int n = n0 + j * NWS;
                float sdot = 0.f;
#pragma unroll
                for (int c = 0; c < KCH; ++c)
                    sdot += wv[j][c].x * yv[c].x + wv[j][c].y * yv[c].y + wv[j][c].z * yv[c].z + wv[j][c].w * yv[c].w;
                sdot = wave_sum(sdot);
                if (lane == 0 && n < V) logit[n] = sdot + bo[j];
            }
        }
    }
    __syncthreads();
    if (p.logits_out)
        for (int n = tid; n < V; n += NWS * 64) p.logits_out[((size_t)t * p.B + b) * V + n] = logit[n];

    // ---- decision, by wave 0 ----
    if (wave == 0) {
        float x[3], pr[3];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int n = lane + 64 * k;
            x[k] = (n < V) ? logit[n] : -INFINITY;
            mx = fmaxf(mx, x[k]);
        }
        mx = wave_max(mx);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pr[k] = (lane + 64 * k < V) ? __expf(x[k] - mx) : 0.f;
            se += pr[k];
        }
        se = wave_sum(se);
        int prev = -1;
        if (p.beam == 0) {
            if (cur >= p.max_conseq_chord && cur >= 1) {
                prev = (int)p.tokens[(size_t)b * p.T + cur - 1];
                for (int k = 1; k < p.max_conseq_chord; ++k)
                    if ((int)p.tokens[(size_t)b * p.T + cur - 1 - k] != prev) prev = -1;
            }
        }
        float ps = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int n = lane + 64 * k;
            pr[k] = (n < VP) ? pr[k] / se : 0.f;               // softmax(...)[:157]
            if (p.beam == 0) {
                if (n == 0 && p.max_conseq_N == 0) pr[k] = 0.f;
                if (n == prev) pr[k] = 0.f;
            }
            ps += pr[k];
        }
        ps = wave_sum(ps);
        if (p.probs_out) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (lane + 64 * k < VP) p.probs_out[(size_t)b * VP + lane + 64 * k] = pr[k];
        }
        const int bi = pick_token(p, pr, ps, lane, b, t);
        if (lane == 0) s_tok = bi;
    }
    __syncthreads();

    if (!p.sample_external && cur < p.T) {
        int tok, root, attr;
        if (cur < p.n_primer) {
            tok = (int)p.tokens[(size_t)b * p.T + cur];
            root = (int)p.roots[(size_t)b * p.T + cur];
            attr = (int)p.attrs[(size_t)b * p.T + cur];
        } else {
            tok = s_tok;
            feedback_of(p, tok, root, attr);
            if (tid == 0) {
                p.tokens[(size_t)b * p.T + cur] = tok;
                p.roots[(size_t)b * p.T + cur] = root;
                p.attrs[(size_t)b * p.T + cur] = attr;
            }
        }
        write_next_input(p, b, cur, root, attr);
        advance_pos(p, t);
    }
}

// Output head when the last skinny GEMM already produced the raw logits (SampleParams::lraw): wave 0 computes the two
// LayerNorms' statistics of the clip's row, fixes the 159 raw logits, takes the decision; the whole block then writes
// the next input.  Same decision code as sample_kernel.
template <int KCH>
__global__ __launch_bounds__(256) void sample_fold_kernel(SampleParams p) {
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = p.d;
    const int t = *p.pos, cur = t + 1;
    if (wave == 0) {
        const int bi = decide_fold_wave<KCH>(p, b, t, lane);
        if (lane == 0) s_tok = bi;
    }
    __syncthreads();

    if (!p.sample_external && cur < p.T) {
        int tok, root, attr;
        if (cur < p.n_primer) {
            tok = (int)p.tokens[(size_t)b * p.T + cur];
            root = (int)p.roots[(size_t)b * p.T + cur];
            attr = (int)p.attrs[(size_t)b * p.T + cur];
        } else {
            tok = s_tok;
            feedback_of(p, tok, root, attr);
            if (tid == 0) {
                p.tokens[(size_t)b * p.T + cur] = tok;
                p.roots[(size_t)b * p.T + cur] = root;
                p.attrs[(size_t)b * p.T + cur] = attr;
            }
        }
        write_next_input(p, b, cur, root, attr);
        advance_pos(p, t);
    }
}

// x_next for position *pos from the stored sequences (first step of generate, or after the host
// committed an externally sampled token at position *pos + advance)
__global__ __launch_bounds__(256) void embed_step_kernel(SampleParams p, int advance) {
    const int b = blockIdx.x;
    const int t = *p.pos, cur = t + advance;
    if (cur < p.T) {
        const bool fb = advance && (p.beam == 0 || p.chord_embed) && cur >= p.n_primer;
        if (fb && threadIdx.x == 0) {
            int r, a;
            feedback_of(p, (int)p.tokens[(size_t)b * p.T + cur], r, a);
            p.roots[(size_t)b * p.T + cur] = r;
            p.attrs[(size_t)b * p.T + cur] = a;
        }
        __syncthreads();
        int root, attr;
        if (fb) {
            feedback_of(p, (int)p.tokens[(size_t)b * p.T + cur], root, attr);
        } else {
            root = (int)p.roots[(size_t)b * p.T + cur];
            attr = (int)p.attrs[(size_t)b * p.T + cur];
        }
        write_next_input(p, b, cur, root, attr);
    }
    if (advance) advance_pos(p, t);
}

}  // namespace

namespace {

// Decision of the lockstep V1 / V2 / V3-family step (model/video_music_transformer.py:547-600 of the reference, per clip):
// probs = softmax(logits / temperature)[:157]; beam == 1: top-1, no feedback of root / attr; beam == 0: N / repeat
// suppression, arg-max of the re-normalised probabilities (oracle G2) or the inverse-CDF draw at the supplied uniform
// (Categorical.sample), token -> (root, attr) by the chord.json rule (chord_embed: the id itself feeds back).  One wave per
// clip.  Runs right after amt_v2_step_batch in the same captured graph: state[0] already points at the position being
// decided; the kernel stores the token and leaves that position's (root, attr) in state[1+b], state[1+B+b] for the next step.
__global__ __launch_bounds__(64) void v2_decide_kernel(const float* __restrict__ logits, int ld_logits, int32_t* __restrict__ state,
                                                      int64_t* __restrict__ tokens, int64_t* __restrict__ roots, int64_t* __restrict__ attrs,
                                                      int B, int T, int n_primer, int beam, int max_conseq_N, int max_conseq_chord,
                                                      float inv_temperature, const float* __restrict__ uniforms, int chord_embed) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int cur = state[0];                                  // position decided now (its predecessor's logits are in `logits`)
    if (cur >= T) return;
    if (cur >= n_primer) {
        const float* lg = logits + (size_t)b * ld_logits;
        float z[3];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int n = lane + 64 * k;
            z[k] = n < V ? lg[n] * inv_temperature : -INFINITY;
            m = fmaxf(m, z[k]);
        }
        m = wave_max(m);
        float pr[3];
        float ps = 0.f;
        const int64_t prev = tokens[(size_t)b * T + cur - 1];
        bool rep = beam == 0 && cur >= max_conseq_chord;
        for (int k = 1; rep && k < max_conseq_chord; ++k) rep = tokens[(size_t)b * T + cur - 1 - k] == prev;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int n = lane + 64 * k;
            float e = n < VP ? __expf(z[k] - m) : 0.f;         // the softmax denominator cancels in both decisions
            if (beam == 0 && ((max_conseq_N == 0 && n == 0) || (rep && n == (int)prev))) e = 0.f;
            pr[k] = e;
            ps += e;
        }
        ps = wave_sum(ps);
        SampleParams sp{};
        sp.beam = beam; sp.uniforms = uniforms; sp.B = B;
        const int tok = pick_token(sp, pr, ps, lane, b, cur - 1);
        if (lane == 0) {
            tokens[(size_t)b * T + cur] = tok;
            int root = ROOT_PAD, attr = ATTR_PAD;              // beam == 1: the generated ids never feed back (:547-560)
            if (chord_embed) { root = tok; attr = 0; }
            else if (beam == 0) { root = tok == 0 ? 0 : (tok - 1) / 13 + 1; attr = tok == 0 ? 1 : (tok - 1) % 13 + 1; }
            roots[(size_t)b * T + cur] = root;
            attrs[(size_t)b * T + cur] = attr;
        }
    }
    __syncthreads();
    if (lane == 0) {
        state[1 + b] = (int32_t)roots[(size_t)b * T + cur];
        state[1 + B + b] = (int32_t)attrs[(size_t)b * T + cur];
    }
}

// The same decision as the tail of ONE launch chain (amt_v2_step_decide_batch): state[0] still holds the position whose logits were
// just computed, the kernel decides position cur = state[0] + 1, writes the chord-stream row of that position (the input of the
// next step: what embed_rows_kernel computes at the head of amt_v2_step_batch) and the last block to finish advances state[0]
// (ticket in state[1 + 2B]; every block has read state[0] before it takes its ticket).
__global__ __launch_bounds__(64) void v2_decide_fused_kernel(const float* __restrict__ logits, int ld_logits, int32_t* __restrict__ state,
                                                            int64_t* __restrict__ tokens, int64_t* __restrict__ roots, int64_t* __restrict__ attrs,
                                                            int B, int T, int n_primer, int beam, int max_conseq_N, int max_conseq_chord,
                                                            float inv_temperature, const float* __restrict__ uniforms, int chord_embed,
                                                            const float* __restrict__ keys, const float* __restrict__ PR, const float* __restrict__ PA,
                                                            const float* __restrict__ wkey, const float* __restrict__ bias,
                                                            const float* __restrict__ pe, float* __restrict__ x_next, int d) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int cur = state[0] + 1;
    if (cur < T) {
        // (root, attr) of position cur: decided here (every lane holds the same token after pick_token), or the stored primer's.  They used
        // to travel lane 0 -> global memory -> lane 0 -> LDS: a store / load round trip and two barriers inside a one-wave kernel.
        int root, attr;
        if (cur >= n_primer) {
            const float* lg = logits + (size_t)b * ld_logits;
            // the id history is requested before the logits are reduced (it sat behind the first reduction: one more L2 round trip)
            const int64_t prev = tokens[(size_t)b * T + cur - 1];
            bool rep = beam == 0 && cur >= max_conseq_chord;
            for (int k = 1; rep && k < max_conseq_chord; ++k) rep = tokens[(size_t)b * T + cur - 1 - k] == prev;
            float z[3];
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int n = lane + 64 * k;
                z[k] = n < V ? lg[n] * inv_temperature : -INFINITY;
                m = fmaxf(m, z[k]);
            }
            m = wave_max(m);
            float pr[3];
            float ps = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int n = lane + 64 * k;
                float e = n < VP ? __expf(z[k] - m) : 0.f;
                if (beam == 0 && ((max_conseq_N == 0 && n == 0) || (rep && n == (int)prev))) e = 0.f;
                pr[k] = e;
                ps += e;
            }
            ps = wave_sum(ps);
            SampleParams sp{};
            sp.beam = beam; sp.uniforms = uniforms; sp.B = B;
            const int tok = pick_token(sp, pr, ps, lane, b, cur - 1);
            root = ROOT_PAD; attr = ATTR_PAD;
            if (chord_embed) { root = tok; attr = 0; }
            else if (beam == 0) { root = tok == 0 ? 0 : (tok - 1) / 13 + 1; attr = tok == 0 ? 1 : (tok - 1) % 13 + 1; }
            if (lane == 0) {
                tokens[(size_t)b * T + cur] = tok;
                roots[(size_t)b * T + cur] = root;
                attrs[(size_t)b * T + cur] = attr;
            }
        } else {
            root = (int)roots[(size_t)b * T + cur];
            attr = (int)attrs[(size_t)b * T + cur];
        }
        if (lane == 0) {
            state[1 + b] = root;
            state[1 + B + b] = attr;
        }
        // x_next[b] = chord-stream row of position cur (the summation order of embed_rows_kernel)
        const float kv = keys[b];
        const float* pp_row = pe ? pe + (size_t)cur * d : nullptr;
        for (int c = lane * 4; c < d; c += 256) {
            const float4 pr4 = ld4(PR + (size_t)root * d + c), pa = ld4(PA + (size_t)attr * d + c);
            const float4 wk = ld4(wkey + c), bb = ld4(bias + c);
            const float4 pp = pp_row ? ld4(pp_row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = ((pr4.x + pa.x) + kv * wk.x + bb.x) + pp.x; o.y = ((pr4.y + pa.y) + kv * wk.y + bb.y) + pp.y;
            o.z = ((pr4.z + pa.z) + kv * wk.z + bb.z) + pp.z; o.w = ((pr4.w + pa.w) + kv * wk.w + bb.w) + pp.w;
            st4(x_next + (size_t)b * d + c, o);
        }
    }
    // advance: the last block to arrive publishes the new position and re-arms the ticket
    if (lane == 0) {
        __threadfence();
        const int ticket = atomicAdd(&state[1 + 2 * B], 1);
        if (ticket == B - 1) {
            state[1 + 2 * B] = 0;
            // never past the last position: a replay too many then recomputes position T-1 (cache row T-1 exists, T <= cap is the
            // caller's contract) instead of walking the K/V caches out of bounds
            state[0] = min(cur, T - 1);
        }
    }
}

}  // namespace

int32_t amt_launch_v2_decide_fused(const float* logits, int ld_logits, int32_t* state_dev, int64_t* tokens, int64_t* roots, int64_t* attrs,
                                   int B, int T, int n_primer, int beam, int max_conseq_N, int max_conseq_chord, float temperature,
                                   const float* uniforms, int chord_embed, const float* keys, const float* PR, const float* PA,
                                   const float* wkey, const float* bias, const float* pe, float* x_next, int d, hipStream_t stream) {
    AMT_CHECK_ARG(logits && state_dev && tokens && roots && attrs && keys && PR && PA && wkey && bias && x_next, "v2_decide_fused: null pointer");
    AMT_CHECK_ARG(B > 0 && T > 1 && n_primer >= 1 && n_primer <= T && ld_logits >= 159 && d % 4 == 0, "v2_decide_fused: bad shape");
    AMT_CHECK_ARG((beam == 0 || beam == 1) && max_conseq_chord >= 1 && temperature > 0.f, "v2_decide_fused: bad decision parameters");
    hipLaunchKernelGGL(v2_decide_fused_kernel, dim3(B), dim3(64), 0, stream, logits, ld_logits, state_dev, tokens, roots, attrs, B, T, n_primer, beam,
                       max_conseq_N, max_conseq_chord, 1.0f / temperature, uniforms, chord_embed, keys, PR, PA, wkey, bias, pe, x_next, d);
    AMT_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t amt_v2_decide_batch(const float* logits, int32_t ld_logits, int32_t* state_dev, int64_t* tokens, int64_t* roots,
                                       int64_t* attrs, int32_t B, int32_t T, int32_t n_primer, int32_t beam, int32_t max_conseq_N,
                                       int32_t max_conseq_chord, float temperature, const float* uniforms, int32_t chord_embed,
                                       void* stream) {
    AMT_CHECK_ARG(logits && state_dev && tokens && roots && attrs, "amt_v2_decide_batch: null pointer");
    AMT_CHECK_ARG(B > 0 && T > 1 && n_primer >= 1 && n_primer <= T && ld_logits >= 159, "amt_v2_decide_batch: bad shape");
    AMT_CHECK_ARG((beam == 0 || beam == 1) && max_conseq_chord >= 1 && temperature > 0.f, "amt_v2_decide_batch: bad decision parameters");
    hipLaunchKernelGGL(v2_decide_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ld_logits, state_dev, tokens, roots, attrs, B, T,
                       n_primer, beam, max_conseq_N, max_conseq_chord, 1.0f / temperature, uniforms, chord_embed);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_sample(const SampleParams& p, hipStream_t stream) {
    AMT_CHECK_ARG(p.B > 0 && p.d % 4 == 0 && p.d <= 1024, "sample: bad shape B=%d d=%d", p.B, p.d);
    if (p.lraw) {
        AMT_CHECK_ARG(p.h1 && p.h2 && p.h3 && p.h4 && p.ln_w && p.ln_b && p.ld_lraw >= V, "sample: incomplete folded head");
        switch ((p.d + 255) / 256) {
            case 1: hipLaunchKernelGGL(sample_fold_kernel<1>, dim3(p.B), dim3(256), 0, stream, p); break;
            case 2: hipLaunchKernelGGL(sample_fold_kernel<2>, dim3(p.B), dim3(256), 0, stream, p); break;
            case 3: hipLaunchKernelGGL(sample_fold_kernel<3>, dim3(p.B), dim3(256), 0, stream, p); break;
            default: hipLaunchKernelGGL(sample_fold_kernel<4>, dim3(p.B), dim3(256), 0, stream, p); break;
        }
        AMT_LAUNCH_CHECK();
        return 0;
    }
    switch ((p.d + 255) / 256) {
        case 1: hipLaunchKernelGGL(sample_kernel<1>, dim3(p.B), dim3(NWS * 64), 0, stream, p); break;
        case 2: hipLaunchKernelGGL(sample_kernel<2>, dim3(p.B), dim3(NWS * 64), 0, stream, p); break;
        case 3: hipLaunchKernelGGL(sample_kernel<3>, dim3(p.B), dim3(NWS * 64), 0, stream, p); break;
        default: hipLaunchKernelGGL(sample_kernel<4>, dim3(p.B), dim3(NWS * 64), 0, stream, p); break;
    }
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_embed_step(const SampleParams& p, int advance, hipStream_t stream) {
    AMT_CHECK_ARG(p.B > 0 && p.d % 4 == 0, "embed_step: bad shape");
    hipLaunchKernelGGL(embed_step_kernel, dim3(p.B), dim3(256), 0, stream, p, advance);
    AMT_LAUNCH_CHECK();
    return 0;
}
