// Device-side pieces of the sampling head shared by sample.hip (the stand-alone head) and attn_decode.hip (the decision taken in
// the prologue of the self-attention that starts the next step): chord-id feedback rule, the pick, the folded head's decision.
#pragma once
#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int V = 159, VP = 157;       // CHORD_SIZE, CHORD_END
constexpr int ROOT_PAD = 14, ATTR_PAD = 15;

// (root, attr) the chosen token feeds back as the next input (model/video_music_transformer.py:1107-1123: chord.json's layout,
// plain roots carry attr 1, N = (0, 1); beam == 1 leaves the PAD pair, :1078-1084).  chord_embed (:926-937, 986-987): the chord
// id itself indexes the frozen table, in both branches; the attr slot indexes an all-zero row.
__device__ __forceinline__ void feedback_of(const SampleParams& p, int tok, int& root, int& attr) {
    if (p.chord_embed) { root = tok; attr = 0; }
    else if (p.beam == 0) { root = tok == 0 ? 0 : (tok - 1) / 13 + 1; attr = tok == 0 ? 1 : (tok - 1) % 13 + 1; }
    else { root = ROOT_PAD; attr = ATTR_PAD; }
}

// The decision of one clip by one wave: pr[k] = (masked, un-normalised) probability of token lane + 64k, ps their sum.
// Arg-max of pr / ps (ties -> lowest id), or, with uniforms, the inverse-CDF draw: the first token whose cumulative
// probability reaches u * ps (tokens are ordered lane-major inside k = 0, 1, 2).
__device__ __forceinline__ int pick_token(const SampleParams& p, const float (&pr)[3], float ps, int lane, int b, int t) {
    if (p.beam == 0 && p.uniforms) {
        const float target = p.uniforms[(size_t)t * p.B + b] * ps;
        float base = 0.f;
        int tok = -1, last = -1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float c = pr[k];                                   // inclusive scan over the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float up = __shfl_up(c, off, 64);
                if (lane >= off) c += up;
            }
            const unsigned long long hit = __ballot(pr[k] > 0.f && base + c >= target);
            const unsigned long long pos = __ballot(pr[k] > 0.f);
            if (tok < 0 && hit) tok = 64 * k + __ffsll((long long)hit) - 1;
            if (pos) last = 64 * k + 63 - __clzll((long long)pos);
            base += __shfl(c, 63, 64);
        }
        return tok >= 0 ? tok : (last >= 0 ? last : 0);        // rounding at u -> 1: the last token with positive mass
    }
    float bv = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = lane + 64 * k;
        const float pn = (p.beam == 0) ? pr[k] / ps : pr[k];   // Categorical normalises its probs
        if (n < VP && pn > bv) { bv = pn; bi = n; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    return bi;
}

// The decision of the folded output head for clip b by ONE wave (the last skinny GEMM already produced the raw logits,
// SampleParams::lraw): statistics of the two stacked LayerNorms of the clip's row u, the 159 logits
//   rstd2 * (rstd * (lraw - mu*h1) + h2 - m2*h3) + h4,
// softmax[:157], the N / repeat suppression of the reference loop (model/video_music_transformer.py:1085-1102) and the pick.
// t = position whose logits these are (the token decided is t + 1).  Writes logits_out / probs_out when asked.  Every lane
// returns the token.  Shared by sample_fold_kernel and the decode attention that starts the next step (attn_decode.hip).
template <int KCH>
__device__ __forceinline__ int decide_fold_wave(const SampleParams& p, int b, int t, int lane, bool write_out = true) {
    const int d = p.d, cur = t + 1;
    const float inv_d = 1.0f / (float)d;
    float4 v[KCH];
    // unguarded, clamped loads (columns past d re-read the last float4 and stay out of the sums): a load under a branch, or a
    // `cond ? load : 0`, makes the compiler drain the loads in flight; the LayerNorm affine used to be fetched chunk by chunk
    // inside the pass loop, two serial L2 round trips
    float4 lw[KCH], lb[KCH];
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int i = min((c * 64 + lane) * 4, d - 4);
        v[c] = ld4(p.u + (size_t)b * p.ldu + i);
        lw[c] = ld4(p.ln_w + i);
        lb[c] = ld4(p.ln_b + i);
    }
    float raw[3], a1[3], a2[3], a3[3], a4[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = lane + 64 * k, nc = min(n, V - 1);
        raw[k] = p.lraw[(size_t)b * p.ld_lraw + nc];
        a1[k] = p.h1[nc]; a2[k] = p.h2[nc]; a3[k] = p.h3[nc]; a4[k] = p.h4[nc];
    }
    float mu[2], rs[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c)
            if ((c * 64 + lane) * 4 < d) s += v[c].x + v[c].y + v[c].z + v[c].w;
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
        mu[pass] = mean; rs[pass] = rstd;
        if (pass == 0) {
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int i = (c * 64 + lane) * 4;
                if (i < d) {
                    const float4 g = lw[c], h = lb[c];
                    v[c].x = (v[c].x - mean) * rstd * g.x + h.x; v[c].y = (v[c].y - mean) * rstd * g.y + h.y;
                    v[c].z = (v[c].z - mean) * rstd * g.z + h.z; v[c].w = (v[c].w - mean) * rstd * g.w + h.w;
                }
            }
        }
    }
    float x[3], pr[3];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = lane + 64 * k;
        x[k] = rs[1] * (rs[0] * (raw[k] - mu[0] * a1[k]) + a2[k] - mu[1] * a3[k]) + a4[k];
        if (n >= V) x[k] = -INFINITY;
        else if (p.logits_out && write_out) p.logits_out[((size_t)t * p.B + b) * V + n] = x[k];
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
    if (p.probs_out && write_out) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (lane + 64 * k < VP) p.probs_out[(size_t)b * VP + lane + 64 * k] = pr[k];
    }
    return pick_token(p, pr, ps, lane, b, t);
}

}  // namespace
