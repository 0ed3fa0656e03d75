// C-ABI layer of libamt_hip, operator half: the stateless entry points (dense / skinny GEMM, norms, rotary embedding,
// prefill and decode attention, feature concatenation, chord embedding) that the parity tests and the stand-alone Python
// modules call.  See include/amt_hip.h for the contract and the reference code each entry point replaces; the handle API
// (amt_create ... amt_generate) is in amt_api.hip, the MoE / GQA entry points in moe.hip, the V1 / V2 step in v2_step.hip.
#include <math.h>
#include <stdint.h>

#include "../../include/amt_hip.h"
#include "amt_common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// stateless operator entry points
// ------------------------------------------------------------------------------------------------
extern "C" int32_t amt_linear_fwd(const float* x, const float* w, const float* bias, const float* resid, float* y,
                                  int32_t M, int32_t N, int32_t K, int32_t relu, void* stream) {
    AMT_CHECK_ARG(x && w && y, "amt_linear_fwd: null pointer");
    GemmParams g = gemm_params(x, K, w, K, y, N, M, N, K, bias);
    g.resid = resid; g.ldr = N; g.relu = relu;
    return amt_launch_gemm(g, (hipStream_t)stream);
}

extern "C" int32_t amt_layernorm_fwd(const float* x, const float* resid, const float* w, const float* b, float* y,
                                     int32_t rows, int32_t dim, float eps, void* stream) {
    AMT_CHECK_ARG(x && y, "amt_layernorm_fwd: null pointer");
    return amt_launch_layernorm(x, resid, w, b, nullptr, nullptr, y, rows, dim, eps, (hipStream_t)stream);
}

extern "C" int32_t amt_rmsnorm_fwd(const float* x, const float* w, float* y, int32_t rows, int32_t dim, float eps, void* stream) {
    AMT_CHECK_ARG(x && y, "amt_rmsnorm_fwd: null pointer");
    return amt_launch_rmsnorm(x, w, y, rows, dim, eps, (hipStream_t)stream);
}

extern "C" int32_t amt_rmsnorm_resid_fwd(const float* x, const float* resid, const float* w, float* y, int32_t rows, int32_t dim,
                                         float eps, void* stream) {
    AMT_CHECK_ARG(x && y, "amt_rmsnorm_resid_fwd: null pointer");
    return amt_launch_rmsnorm(x, w, y, rows, dim, eps, (hipStream_t)stream, resid);
}

extern "C" int32_t amt_diff_subln_fwd(const float* o1, const float* o2, const float* w, float* y, int32_t rows, int32_t hd,
                                      float lambda_full, float out_scale, float eps, void* stream) {
    AMT_CHECK_ARG(o1 && o2 && w && y, "amt_diff_subln_fwd: null pointer");
    return amt_launch_diff_subln(o1, o2, w, y, rows, hd, lambda_full, out_scale, eps, (hipStream_t)stream);
}

extern "C" int32_t amt_row_scale_add_fwd(const float* x, const float* row_scale, const float* add, float* y, int32_t rows, int32_t dim,
                                         void* stream) {
    AMT_CHECK_ARG(x && row_scale && y, "amt_row_scale_add_fwd: null pointer");
    return amt_launch_row_scale_add(x, row_scale, add, y, rows, dim, (hipStream_t)stream);
}

extern "C" int32_t amt_add_fwd(const float* a, const float* b, float* y, int64_t n, void* stream) {
    AMT_CHECK_ARG(a && b && y, "amt_add_fwd: null pointer");
    return amt_launch_add(a, b, y, (long)n, (hipStream_t)stream);
}

extern "C" int32_t amt_rope_fwd(const float* x, const float* cache, float* y, int32_t n0, int32_t seq, int32_t n2, int32_t hd,
                                int32_t cache_half, void* stream) {
    AMT_CHECK_ARG(x && cache && y, "amt_rope_fwd: null pointer");
    return amt_launch_rope(x, cache, y, n0, seq, n2, hd, cache_half, (hipStream_t)stream);
}

static AttnParams blh_params(const float* q, const float* k, const float* v, float* o, int B, int H, int Lq, int Lk, int hd) {
    AttnParams a{};
    const size_t E = (size_t)H * hd;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.q_bs = (size_t)Lq * E; a.k_bs = a.v_bs = (size_t)Lk * E; a.o_bs = (size_t)Lq * E;
    a.q_hs = a.k_hs = a.v_hs = a.o_hs = hd;
    a.q_ls = a.k_ls = a.v_ls = a.o_ls = E;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.hd = hd; a.kv_group = 1;
    return a;
}

extern "C" int32_t amt_rpr_attn_fwd(const float* q, const float* k, const float* v, const float* Er, float* o,
                                    int32_t B, int32_t H, int32_t L, int32_t hd, int32_t er_len, void* stream) {
    AMT_CHECK_ARG(q && k && v && Er && o, "amt_rpr_attn_fwd: null pointer");
    AttnParams a = blh_params(q, k, v, o, B, H, L, L, hd);
    a.causal = 1; a.Er = Er; a.er_len = er_len;
    return amt_launch_attn_prefill(a, (hipStream_t)stream);
}

extern "C" int32_t amt_rpr_attn_nomask_fwd(const float* q, const float* k, const float* v, const float* Er, float* o,
                                           int32_t B, int32_t H, int32_t L, int32_t hd, int32_t er_len, void* stream) {
    AMT_CHECK_ARG(q && k && v && Er && o, "amt_rpr_attn_nomask_fwd: null pointer");
    AttnParams a = blh_params(q, k, v, o, B, H, L, L, hd);
    a.causal = 0; a.Er = Er; a.er_len = er_len;
    return amt_launch_attn_prefill(a, (hipStream_t)stream);
}

extern "C" int32_t amt_cross_attn_fwd(const float* q, const float* k, const float* v, float* o,
                                      int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, int32_t causal, void* stream) {
    AMT_CHECK_ARG(q && k && v && o, "amt_cross_attn_fwd: null pointer");
    AttnParams a = blh_params(q, k, v, o, B, H, Lq, Lk, hd);
    a.causal = causal;
    return amt_launch_attn_prefill(a, (hipStream_t)stream);
}

extern "C" int32_t amt_attn_fwd(const float* q, const float* k, const float* v, float* o, const int64_t* strides,
                                int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, int32_t causal, int32_t kv_group,
                                float q_scale, void* stream) {
    AMT_CHECK_ARG(q && k && v && o && strides, "amt_attn_fwd: null pointer");
    AttnParams a{};
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.q_bs = strides[0]; a.q_hs = strides[1]; a.q_ls = strides[2];
    a.k_bs = strides[3]; a.k_hs = strides[4]; a.k_ls = strides[5];
    a.v_bs = strides[6]; a.v_hs = strides[7]; a.v_ls = strides[8];
    a.o_bs = strides[9]; a.o_hs = strides[10]; a.o_ls = strides[11];
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.hd = hd; a.causal = causal; a.kv_group = kv_group > 0 ? kv_group : 1; a.q_scale = q_scale;
    return amt_launch_attn_prefill(a, (hipStream_t)stream);
}

extern "C" int32_t amt_concat_features_fwd(const float* sem, int32_t sem_dim, const float* scene, const float* motion, int32_t motion_dim,
                                           const float* emotion, int32_t emo_dim, float* out, int32_t rows, int32_t ld_out, void* stream) {
    AMT_CHECK_ARG(sem && scene && motion && emotion && out && rows > 0, "amt_concat_features_fwd: bad argument");
    return amt_launch_concat_features(sem, sem_dim, scene, motion, motion_dim, emotion, emo_dim, out, rows, ld_out, (hipStream_t)stream);
}

extern "C" int32_t amt_chord_embed_fwd(const int64_t* root, const int64_t* attr, const float* key, const float* PR, const float* PA,
                                       const float* wkey, const float* bias, const float* pe, float* out,
                                       int32_t B, int32_t L, int32_t d, void* stream) {
    AMT_CHECK_ARG(root && attr && key && PR && PA && wkey && bias && pe && out && B > 0 && L > 0, "amt_chord_embed_fwd: bad argument");
    return amt_launch_chord_embed(root, attr, key, PR, PA, wkey, bias, pe, out, B, L, d, (hipStream_t)stream);
}

extern "C" int32_t amt_attn_decode_fwd(const float* q, const float* kcache, const float* vcache, const float* Er, float* o,
                                       int32_t B, int32_t H, int32_t hd, int32_t cap, int32_t pos, int32_t er_len, void* stream) {
    AMT_CHECK_ARG(q && kcache && vcache && o, "amt_attn_decode_fwd: null pointer");
    AMT_CHECK_ARG(pos >= 0 && pos < cap, "amt_attn_decode_fwd: pos=%d outside 0..%d", pos, cap - 1);
    AttnDecodeParams a{};
    a.q = q; a.k = kcache; a.v = vcache; a.o = o; a.B = B; a.H = H; a.hd = hd; a.cap = cap;
    a.n_keys = pos + 1; a.Er = Er; a.er_len = er_len;
    return amt_launch_attn_decode(a, (hipStream_t)stream);
}

extern "C" int32_t amt_attn_decode_fold_fwd(const float* raw, int32_t ldq, float* kcache, float* vcache, const float* Er,
                                            const float* u, const float* fold_g, const float* fold_c, const float* ln_w,
                                            const float* ln_b, float* xn_out, float* o, int32_t B, int32_t H, int32_t hd,
                                            int32_t cap, const int32_t* pos_dev, int32_t n_keys, int32_t er_len, int32_t new_kv,
                                            float eps, float q_scale, void* stream) {
    AMT_CHECK_ARG(raw && kcache && vcache && u && fold_g && fold_c && o, "amt_attn_decode_fold_fwd: null pointer");
    AMT_CHECK_ARG(pos_dev || (!new_kv && n_keys > 0 && n_keys <= cap), "amt_attn_decode_fold_fwd: need a device position or a key count");
    AttnDecodeParams a{};
    a.q = raw; a.ldq = ldq; a.k = kcache; a.v = vcache; a.o = o; a.B = B; a.H = H; a.hd = hd; a.cap = cap; a.d = H * hd;
    a.pos = (const int*)pos_dev; a.n_keys = n_keys; a.Er = Er; a.er_len = er_len;
    a.fold_u = u; a.fold_g = fold_g; a.fold_c = fold_c; a.fold_lnw = ln_w; a.fold_lnb = ln_b; a.xn = xn_out;
    a.new_kv = new_kv; a.k_new = kcache; a.v_new = vcache; a.eps = eps; a.q_scale = q_scale;
    return amt_launch_attn_decode(a, (hipStream_t)stream);
}

extern "C" int32_t amt_decode_gemm_ex_fwd(const amt_decode_gemm_args* a, void* stream) {
    AMT_CHECK_ARG(a, "amt_decode_gemm_ex_fwd: null argument block");
    AMT_CHECK_ARG(a->x && a->w_low && a->y_low && a->scratch_low && a->n_low > 0 && a->n_low % 16 == 0, "amt_decode_gemm_ex_fwd: bad low part");
    AMT_CHECK_ARG(a->n_high == 0 || (a->w_high && a->y_high && a->scratch_high && a->x2), "amt_decode_gemm_ex_fwd: incomplete high part");
    hipStream_t s = (hipStream_t)stream;
    const int Klow = a->x2 ? a->K1 : a->K;
    int32_t rc;
    if ((rc = amt_launch_pack_weight(a->w_low, a->scratch_low, a->n_low, Klow, s))) return rc;
    if (a->n_high > 0 && (rc = amt_launch_pack_weight(a->w_high, a->scratch_high, a->n_high, a->K, s))) return rc;
    DecodeGemmParams g{};
    g.B = a->B; g.eps = a->eps; g.scale = 1.f; g.x = a->x; g.ldx = a->ldx; g.x2 = a->x2; g.ldx2 = a->ldx2; g.K1 = a->K1; g.K = a->K;
    g.Wp = a->scratch_low; g.bias = a->bias_low; g.resid = a->resid; g.ldr = a->n_low; g.relu = a->relu; g.y = a->y_low; g.ldy = a->n_low;
    g.pro = a->pro; g.fold_g = a->fold_g; g.fold_c = a->fold_c; g.ln_w = a->ln_w; g.ln_b = a->ln_b;
    g.N = a->n_low + a->n_high;
    if (a->x2) { g.n_split = a->n_low; g.Wp2 = a->scratch_high; g.bias2 = a->bias_high; g.y2 = a->y_high; g.ldy2 = a->n_high; }
    return amt_launch_decode_gemm(g, s);
}

extern "C" int32_t amt_decode_linear_fwd(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b,
                                         const float* resid, float* y, float* xn_out, float* w_packed_scratch,
                                         int32_t B, int32_t N, int32_t K, int32_t relu, float eps, void* stream) {
    AMT_CHECK_ARG(x && w && y && w_packed_scratch, "amt_decode_linear_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!amt_tuning().prepacked) {                                 // (micro-benchmarks of an experiment build: scratch already holds the packed weight)
        int32_t rc = amt_launch_pack_weight(w, w_packed_scratch, N, K, s);
        if (rc) return rc;
    }
    DecodeGemmParams g{};
    g.x = x; g.ldx = K; g.Wp = w_packed_scratch; g.bias = bias; g.B = B; g.N = N; g.K = K;
    g.ln_w = ln_w; g.ln_b = ln_b; g.xn = xn_out; g.eps = eps; g.resid = resid; g.ldr = N; g.relu = relu;
    g.scale = 1.f; g.y = y; g.ldy = N;
    return amt_launch_decode_gemm(g, s);
}
