// One KV-cached decode step of VideoMusicTransformer_V2 '2.2' for one clip, issued from a single C call.
//
// The step is the launch sequence of `VideoMusicTransformer_V2._decode_step` (video2music_amd/model/
// video_music_transformer.py; reference model/video_music_transformer.py:437-516 + custom_transformer.py:1250-1292
// restricted to the newest position): chord embedding, then per decoder layer
//   in-proj (q|k|v) -> RoPE(q), RoPE(k) into the cache row, v into the cache row -> attention over keys 0..t
//   -> out-proj + residual -> LayerNorm -> cross q-proj -> RoPE -> attention over the clip's video keys
//   -> out-proj + residual -> LayerNorm -> GLU expert or SharedMoE(6, top-2) -> residual LayerNorm,
// then decoder.norm and Wout.  One row per projection: the dense 128x128-tile GEMM of the prefill path takes ~65 us for such
// a launch (every K step exposes a full memory latency), so the projections run on the skinny decode GEMM over weights the
// host packed once (amt_pack_weight_fwd); the two experts a token is routed to are picked on the device (weight group index
// read by the kernel), so no routing result ever travels to the host.
#include "../../include/amt_hip.h"
#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int G_PTRS = 11, L_PTRS = 48;
enum { G_PR, G_PA, G_WKEY, G_CBIAS, G_ROPE, G_FNW, G_FNB, G_WOUT, G_BOUT, G_SLOT01, G_PE };
enum { L_SAW, L_SAB, L_SAOW, L_SAOB, L_N1W, L_N1B, L_CAW, L_CAB, L_CAOW, L_CAOB, L_N2W, L_N2B, L_N3W, L_N3B,
       L_KC, L_VC, L_KX, L_VX, L_GATEW, L_GATEB, L_W1, L_B1, L_WG, L_BG, L_W2, L_B2, L_SW1, L_SB1, L_SWG, L_SBG, L_SW2, L_SB2,
       // stacked forms for the lockstep step: [gate of every expert (+ shared) | linear1 of every expert (+ shared)] as ONE packed
       // matrix -- when linear1 exists its rows INTERLEAVED in eights before packing (gate rows 8T..8T+7, then linear1 rows 8T..8T+7:
       // DecodeGemmParams::glu_pair) -- and its bias in the stacked order; linear2 of every expert (+ shared) one after the other and
       // their biases (null for a plain GLU layer)
       L_GU, L_GUB, L_W2S, L_B2S,
       // lockstep step, norm1 folded through the cross-attention's query projection (null: separate launches): packed
       // [(Wq o gamma) Wo | Wq o gamma] (E x 2E), its bias (Wq o gamma) bo, g = rowsum(Wq o gamma), c = Wq beta + bq
       L_G1P, L_G1B, L_FQG, L_FQC,
       // lockstep step of a plain GLU layer, norm2 folded through the stacked gate | up product (the base model's
       // G2): packed [(Wgu o gamma2) Wo | Wgu o gamma2] (2 dff x 2E), its bias (Wgu o gamma2) bo, g = rowsum(Wgu o gamma2),
       // c = Wgu beta2 + bgu (gate columns first, then up) -- and norm3 folded through the NEXT layer's QKV projection (the base
       // model's G3): packed [(Wqkv' o gamma3) W2 | Wqkv' o gamma3] (3E x (dff + E)), bias (Wqkv' o gamma3) b2, g, c; null in the
       // last layer.  Null entries: the layer takes the separate launches.
       L_G2P, L_G2B, L_FGUG, L_FGUC, L_G3P, L_G3B, L_FKG, L_FKC };

// (root, attr) either as launch arguments or, for a captured step graph, from device memory (tok[0], tok[1])
__global__ void embed_one_kernel(int root, int attr, const int* __restrict__ tok, float kv, const float* __restrict__ PR,
                                 const float* __restrict__ PA, const float* __restrict__ wkey, const float* __restrict__ bias,
                                 float* __restrict__ out, int d, const float* __restrict__ pe, int t) {
    if (tok) { root = tok[0]; attr = tok[1]; t = tok[-1]; }      // device state {position, root, attr}
    if (pe) pe += (size_t)t * d;                                 // learned positional row (version '2.0' / V1), else none
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        const float4 pr = ld4(PR + (size_t)root * d + c), pa = ld4(PA + (size_t)attr * d + c);
        const float4 wk = ld4(wkey + c), bb = ld4(bias + c);
        const float4 pp = pe ? ld4(pe + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o;        // the summation order of chord_embed_kernel
        o.x = ((pr.x + pa.x) + kv * wk.x + bb.x) + pp.x; o.y = ((pr.y + pa.y) + kv * wk.y + bb.y) + pp.y;
        o.z = ((pr.z + pa.z) + kv * wk.z + bb.z) + pp.z; o.w = ((pr.w + pa.w) + kv * wk.w + bb.w) + pp.w;
        st4(out + c, o);
    }
}

__global__ void advance_kernel(int* pos) { *pos += 1; }

// RoPE of one position's full d_model vector (pair i rotated by the cache row's (cos, sin) i, as rope_kernel does), times
// `scale`, placed either contiguously (query) or as row t of a head-major [H][cap][hd] cache (key); rope == null: plain copy
// (value).  One thread per pair.
__global__ void rope_place_kernel(const float* __restrict__ x, const float* __restrict__ rope, float* __restrict__ dst,
                                  float scale, int E, int hd, int cap, int t, const int* __restrict__ pos, int head_major) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i >= E) return;
    if (pos) { t = *pos; if (rope) rope += (size_t)t * E; }      // captured step: position from device memory, rope = table base
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    float y0 = x0, y1 = x1;
    if (rope) {
        const float c = rope[2 * i], sn = rope[2 * i + 1];
        y0 = x0 * c - x1 * sn;
        y1 = x1 * c + x0 * sn;
    }
    const int e = 2 * i, h = e / hd, cc = e - h * hd;
    float* o = head_major ? dst + ((size_t)h * cap + t) * hd + cc : dst + e;
    o[0] = y0 * scale;
    o[1] = y1 * scale;
}

// h = u * silu(g)   (GLUExpert.forward, moe.py:44-49)
// (u == null: h = silu(g), the Linear -> SiLU -> Linear experts of the V1 family)
__global__ void glu_mul_kernel(const float* __restrict__ u, const float* __restrict__ g, float* __restrict__ h, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < n) {
        const float4 a = u ? ld4(u + i) : make_float4(1.f, 1.f, 1.f, 1.f), b = ld4(g + i);
        float4 o;
        o.x = a.x * (b.x / (1.0f + __expf(-b.x))); o.y = a.y * (b.y / (1.0f + __expf(-b.y)));
        o.z = a.z * (b.z / (1.0f + __expf(-b.z))); o.w = a.w * (b.w / (1.0f + __expf(-b.w)));
        st4(h + i, o);
    }
}

// one query over a head-major cache [H][cap][hd] on the K/V-streaming decode kernel (q already scaled)
int32_t attn_one(const float* q, const float* k, const float* v, float* o, int H, int hd, int cap, int n_keys, const int* pos,
                 hipStream_t s) {
    AttnDecodeParams a{};
    a.q = q; a.k = k; a.v = v; a.o = o; a.B = 1; a.H = H; a.hd = hd; a.cap = cap; a.n_keys = n_keys; a.pos = pos;   // pos: keys 0..*pos
    return amt_launch_attn_decode(a, s);
}

int32_t place(const float* x, const float* rope, float* dst, float scale, int E, int hd, int cap, int t, const int* pos,
              int head_major, hipStream_t s) {
    hipLaunchKernelGGL(rope_place_kernel, dim3(cdiv(E / 2, 256)), dim3(256), 0, s, x, rope, dst, scale, E, hd, cap, t, pos, head_major);
    AMT_LAUNCH_CHECK();
    return 0;
}

// one row through a pre-packed weight on the skinny GEMM: y[N] = x[K] . W^T + b (+ resid); optional device-chosen group
int32_t lin(const float* x, const float* wp, const float* b, const float* resid, float* y, int N, int K, hipStream_t s,
            const int* sel = nullptr, size_t sel_w = 0, int sel_b = 0) {
    DecodeGemmParams g{};
    g.B = 1; g.eps = 1e-5f; g.scale = 1.f; g.x = x; g.ldx = K; g.Wp = wp; g.bias = b; g.N = N; g.K = K;
    g.resid = resid; g.ldr = N; g.y = y; g.ldy = N; g.sel = sel; g.sel_w_stride = sel_w; g.sel_b_stride = sel_b;
    return amt_launch_decode_gemm(g, s);
}

// one row through a layer's norm: LayerNorm(x + resid) (eps 1e-5), or -- bias pointer null -- RMSNorm(x + resid) (eps 1e-6,
// custom_transformer.py:27-45: the V1 family built with rms_norm=True)
int32_t norm_one(const float* x, const float* resid, const float* w, const float* b, float* y, int E, hipStream_t s) {
    if (b) return amt_launch_layernorm(x, resid, w, b, nullptr, nullptr, y, 1, E, 1e-5f, s);
    return amt_launch_rmsnorm(x, w, y, 1, E, 1e-6f, s, resid);
}

// GLUExpert on one row: y = W2 ((W1 x + b1) * silu(Wg x + bg)) + b2 ; scratch 3*dff floats
int32_t glu_one(const float* x, const float* w1, const float* b1, const float* wg, const float* bg, const float* w2, const float* b2,
                float* y, float* scratch, int E, int dff, hipStream_t s, const int* sel = nullptr) {
    float* g = scratch; float* u = g + dff; float* hh = u + dff;
    const size_t sw = (size_t)dff * E;
    int32_t rc;
    if ((rc = lin(x, wg, bg, nullptr, g, dff, E, s, sel, sw, dff))) return rc;
    if (w1 && (rc = lin(x, w1, b1, nullptr, u, dff, E, s, sel, sw, dff))) return rc;
    hipLaunchKernelGGL(glu_mul_kernel, dim3(cdiv(dff, 1024)), dim3(256), 0, s, w1 ? u : nullptr, g, hh, dff);
    AMT_LAUNCH_CHECK();
    return lin(hh, w2, b2, nullptr, y, E, dff, s, sel, sw, E);
}

}  // namespace

extern "C" int64_t amt_v2_step_ws_floats(int32_t E, int32_t dff, int32_t n_exp) {
    (void)n_exp;
    return (int64_t)16 * E + 4 * dff + 64;
}

extern "C" int32_t amt_pack_weight_fwd(const float* w, float* out, int32_t N, int32_t K, void* stream) {
    AMT_CHECK_ARG(w && out && N > 0, "amt_pack_weight_fwd: bad argument");
    return amt_launch_pack_weight(w, out, N, K, (hipStream_t)stream);
}

extern "C" int32_t amt_v2_step(const void* const* tab, int32_t n_layers, int32_t H, int32_t E, int32_t dff, int32_t n_exp,
                               int32_t S, int32_t max_seq, int32_t t, int32_t root, int32_t attr, float key, const int32_t* state_dev,
                               float* logits_out, float* ws, void* stream) {
    AMT_CHECK_ARG(tab && logits_out && ws, "amt_v2_step: null pointer");
    AMT_CHECK_ARG(n_layers > 0 && H > 0 && E % H == 0 && E % 64 == 0 && dff % 64 == 0 && E <= 1536 && dff <= 1536 && t >= 0 && t < max_seq && S > 0,
                  "amt_v2_step: bad shape (E and dff must be multiples of 64, at most 1536)");
    hipStream_t s = (hipStream_t)stream;
    const int hd = E / H;
    const float qscale = 1.0f / sqrtf((float)hd);
    auto G = [&](int i) { return (const float*)tab[i]; };
    float* x = ws; float* y = x + E; float* qkv = y + E; float* q = qkv + 3 * E; float* o = q + E; float* u = o + E;
    float* Y2 = u + E;                                    // two expert outputs [2][E]
    float* ysh = Y2 + 2 * E;                              // shared expert output [E]
    float* ffs = ysh + E;                                 // 3*dff GLU scratch
    float* moe_w = ffs + 3 * dff;                         // routing weights [2], then indices [2]
    int32_t* moe_idx = (int32_t*)(moe_w + 4);
    int32_t rc;
    // state_dev (optional, for a captured step graph): {position, root, attr} in device memory; the step then reads them there
    // and increments the position at its end, so one captured graph serves every token
    const int* pos = (const int*)state_dev;
    const int* tok = pos ? pos + 1 : nullptr;
    hipLaunchKernelGGL(embed_one_kernel, dim3(1), dim3(128), 0, s, root, attr, tok, key, G(G_PR), G(G_PA), G(G_WKEY), G(G_CBIAS), x, E, G(G_PE), t);
    AMT_LAUNCH_CHECK();
    // cache row t: E/2 (cos, sin) pairs; no table (null) = no rotation (version '2.0' / the V1 family)
    const float* rope_row = !G(G_ROPE) ? nullptr : pos ? G(G_ROPE) : G(G_ROPE) + (size_t)t * E;
    for (int l = 0; l < n_layers; ++l) {
        const void* const* L = tab + G_PTRS + (size_t)l * L_PTRS;
        auto P = [&](int i) { return (const float*)L[i]; };
        float* kc = (float*)L[L_KC]; float* vc = (float*)L[L_VC];
        // self-attention: caches are head-major [H][max_seq][hd]; q is scaled here (the decode kernel takes a scaled query)
        if ((rc = lin(x, P(L_SAW), P(L_SAB), nullptr, qkv, 3 * E, E, s))) return rc;
        if ((rc = place(qkv, rope_row, q, qscale, E, hd, 0, 0, pos, 0, s))) return rc;
        if ((rc = place(qkv + E, rope_row, kc, 1.f, E, hd, max_seq, t, pos, 1, s))) return rc;
        if ((rc = place(qkv + 2 * E, nullptr, vc, 1.f, E, hd, max_seq, t, pos, 1, s))) return rc;
        if ((rc = attn_one(q, kc, vc, o, H, hd, max_seq, t + 1, pos, s))) return rc;
        if ((rc = lin(o, P(L_SAOW), P(L_SAOB), x, u, E, E, s))) return rc;
        if ((rc = norm_one(u, nullptr, P(L_N1W), P(L_N1B), x, E, s))) return rc;
        // cross-attention over the clip's (roped) video keys, head-major [H][S][hd]
        if ((rc = lin(x, P(L_CAW), P(L_CAB), nullptr, qkv, E, E, s))) return rc;
        if ((rc = place(qkv, rope_row, q, qscale, E, hd, 0, 0, pos, 0, s))) return rc;
        if ((rc = attn_one(q, P(L_KX), P(L_VX), o, H, hd, S, S, nullptr, s))) return rc;
        if ((rc = lin(o, P(L_CAOW), P(L_CAOB), x, u, E, E, s))) return rc;
        if ((rc = norm_one(u, nullptr, P(L_N2W), P(L_N2B), x, E, s))) return rc;
        // feed-forward: GLU expert (shallow layers) or shared mixture of experts (router, the two chosen experts read
        // their weights through the device-side index, shared expert, weighted sum in expert-index order)
        if (!L[L_GATEW]) {
            if ((rc = glu_one(x, P(L_W1), P(L_B1), P(L_WG), P(L_BG), P(L_W2), P(L_B2), y, ffs, E, dff, s))) return rc;
        } else {
            if ((rc = amt_moe_route_fwd(x, P(L_GATEW), P(L_GATEB), moe_idx, moe_w, 1, E, n_exp, s))) return rc;
            for (int slot = 0; slot < 2; ++slot)
                if ((rc = glu_one(x, P(L_W1), P(L_B1), P(L_WG), P(L_BG), P(L_W2), P(L_B2), Y2 + (size_t)slot * E, ffs, E, dff, s, moe_idx + slot))) return rc;
            const float* shared = nullptr;
            if (L[L_SWG]) {
                if ((rc = glu_one(x, P(L_SW1), P(L_SB1), P(L_SWG), P(L_SBG), P(L_SW2), P(L_SB2), ysh, ffs, E, dff, s))) return rc;
                shared = ysh;
            }
            if ((rc = amt_moe_combine_fwd(Y2, (const int32_t*)tab[G_SLOT01], moe_idx, moe_w, shared, 0.5f, y, 1, E, s))) return rc;
        }
        if ((rc = norm_one(y, x, P(L_N3W), P(L_N3B), u, E, s))) return rc;
        float* tmp = x; x = u; u = tmp;
    }
    if ((rc = norm_one(x, nullptr, G(G_FNW), G(G_FNB), y, E, s))) return rc;
    if ((rc = lin(y, G(G_WOUT), G(G_BOUT), nullptr, logits_out, 159, E, s))) return rc;
    if (pos) { hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, s, (int*)state_dev); AMT_LAUNCH_CHECK(); }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// The same step for B independent clips in lockstep (all at the same position): every projection is one skinny-GEMM launch
// over B rows -- the weights are read once per step instead of once per clip --, the attentions run the decode kernel with
// a clip dimension, and a mixture layer evaluates all its experts on all rows (at B >= 3 every expert's weights are read
// anyway; 3x the flops of the routed pair is nothing at these sizes) and combines each row's two in expert-index order.
// ------------------------------------------------------------------------------------------------------------------------
namespace {

// state {position, root[B], attr[B]} in device memory; keys[B]
__global__ void embed_rows_kernel(const int* __restrict__ state, int B, const float* __restrict__ keys, const float* __restrict__ PR,
                                  const float* __restrict__ PA, const float* __restrict__ wkey, const float* __restrict__ bias,
                                  float* __restrict__ out, int d, const float* __restrict__ pe) {
    const int b = blockIdx.x;
    const int t = state[0], root = state[1 + b], attr = state[1 + B + b];
    const float kv = keys[b];
    if (pe) pe += (size_t)t * d;
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        const float4 pr = ld4(PR + (size_t)root * d + c), pa = ld4(PA + (size_t)attr * d + c);
        const float4 wk = ld4(wkey + c), bb = ld4(bias + c);
        const float4 pp = pe ? ld4(pe + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o;        // the summation order of embed_one_kernel
        o.x = ((pr.x + pa.x) + kv * wk.x + bb.x) + pp.x; o.y = ((pr.y + pa.y) + kv * wk.y + bb.y) + pp.y;
        o.z = ((pr.z + pa.z) + kv * wk.z + bb.z) + pp.z; o.w = ((pr.w + pa.w) + kv * wk.w + bb.w) + pp.w;
        st4(out + (size_t)b * d + c, o);
    }
}

// rope_place_kernel over B rows: x row b at x + b*ldx; dst row b contiguous [B][E] or clip b of a [B][H][cap][hd] cache
__global__ void rope_place_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ rope, float* __restrict__ dst,
                                       float scale, int E, int hd, int cap, const int* __restrict__ pos, int head_major) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (2 * i >= E) return;
    const int t = *pos;
    x += (size_t)b * ldx;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    float y0 = x0, y1 = x1;
    if (rope) {
        const float c = rope[(size_t)t * E + 2 * i], sn = rope[(size_t)t * E + 2 * i + 1];
        y0 = x0 * c - x1 * sn;
        y1 = x1 * c + x0 * sn;
    }
    const int e = 2 * i, h = e / hd, cc = e - h * hd;
    float* o = head_major ? dst + (((size_t)b * (E / hd) + h) * cap + t) * hd + cc : dst + (size_t)b * E + e;
    o[0] = y0 * scale;
    o[1] = y1 * scale;
}

// the three placements of a self-attention in one launch (blockIdx.z: 0 = query -> rotated, scaled, [B][E]; 1 = key -> rotated,
// cache row t; 2 = value -> cache row t); qkv rows [B][3E]
__global__ void place_qkv_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ rope, float* __restrict__ q,
                                      float* __restrict__ kc, float* __restrict__ vc, float qscale, int E, int hd, int cap,
                                      const int* __restrict__ pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, part = blockIdx.z;
    if (2 * i >= E) return;
    const int t = *pos;
    const float* x = qkv + (size_t)b * 3 * E + (size_t)part * E;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    float y0 = x0, y1 = x1;
    if (rope && part < 2) {
        const float c = rope[(size_t)t * E + 2 * i], sn = rope[(size_t)t * E + 2 * i + 1];
        y0 = x0 * c - x1 * sn;
        y1 = x1 * c + x0 * sn;
    }
    const int e = 2 * i, h = e / hd, cc = e - h * hd;
    const float scale = part == 0 ? qscale : 1.f;
    float* o = part == 0 ? q + (size_t)b * E + e : (part == 1 ? kc : vc) + (((size_t)b * (E / hd) + h) * cap + t) * hd + cc;
    o[0] = y0 * scale;
    o[1] = y1 * scale;
}

int32_t lin_rows(const float* x, const float* wp, const float* b, const float* resid, float* y, int B, int N, int K, hipStream_t s,
                 int ldx = 0) {
    DecodeGemmParams g{};
    g.B = B; g.eps = 1e-5f; g.scale = 1.f; g.x = x; g.ldx = ldx ? ldx : K; g.Wp = wp; g.bias = b; g.N = N; g.K = K;
    g.resid = resid; g.ldr = N; g.y = y; g.ldy = N;
    return amt_launch_decode_gemm(g, s);
}

int32_t norm_rows(const float* x, const float* resid, const float* w, const float* b, float* y, int B, int E, hipStream_t s) {
    if (b) return amt_launch_layernorm(x, resid, w, b, nullptr, nullptr, y, B, E, 1e-5f, s);
    return amt_launch_rmsnorm(x, w, y, B, E, 1e-6f, s, resid);
}

int32_t place_rows(const float* x, int ldx, const float* rope, float* dst, float scale, int B, int E, int hd, int cap, const int* pos,
                   int head_major, hipStream_t s) {
    hipLaunchKernelGGL(rope_place_rows_kernel, dim3(cdiv(E / 2, 256), B), dim3(256), 0, s, x, ldx, rope, dst, scale, E, hd, cap, pos, head_major);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t attn_rows(const float* q, const float* k, const float* v, float* o, int B, int H, int hd, int cap, int n_keys, const int* pos,
                  hipStream_t s) {
    AttnDecodeParams a{};
    a.q = q; a.k = k; a.v = v; a.o = o; a.B = B; a.H = H; a.hd = hd; a.cap = cap; a.n_keys = n_keys; a.pos = pos;
    return amt_launch_attn_decode(a, s);
}

}  // namespace

extern "C" int64_t amt_v2_step_batch_ws_floats(int32_t E, int32_t dff, int32_t n_exp, int32_t B) {
    const int64_t ne = n_exp > 0 ? n_exp : 1;
    return (int64_t)B * ((int64_t)(13 + n_exp) * E + 3 * (ne + 1) * dff + 8) + 64;
}

namespace {

// One skinny-GEMM launch of the lockstep step with the fusions of round 2 (each replaces a separate launch):
//   ln_w / ln_b : LayerNorm of the input rows in the prologue, the normalised rows also written to xn
//   rope        : rotary epilogue on the first rope_cols output columns (position from device memory)
//   qkv         : mode-1 epilogue -- q scaled to y, k / v rows of this position into the head-major caches
//   gate        : gated-linear-unit prologue, input row = x * silu(gate) (or silu(gate) when x == null)
//   groups      : one launch for the same product of `groups` experts
struct RowGemm {
    const float* x = nullptr; int ldx = 0; const float* wp = nullptr; const float* b = nullptr; const float* resid = nullptr;
    float* y = nullptr; int N = 0, K = 0;
    const float* ln_w = nullptr; const float* ln_b = nullptr; float* xn = nullptr;
    const float* ln2_w = nullptr; const float* ln2_b = nullptr;
    const float* rope = nullptr; int rope_cols = 0, rope_dim = 0; const int* pos = nullptr; float scale = 1.f; int scale_cols = 0;
    bool qkv = false; float* kc = nullptr; float* vc = nullptr; int H = 0, hd = 0, cap = 0;
    const float* gate = nullptr; bool gate_only = false;
    int glu_pair = 0, relu = 0; int ldy = 0;          // paired gate | up epilogue (y = up * silu(gate), N / 2 columns) ; 2: y = silu(.)
    int groups = 1; size_t x_goff = 0, y_goff = 0, w_gstride = 0; int b_gstride = 0;
};

int32_t row_gemm(const RowGemm& r, int B, hipStream_t s) {
    DecodeGemmParams g{};
    g.B = B; g.eps = 1e-5f; g.scale = r.scale; g.scale_cols = r.scale_cols;
    g.x = r.x ? r.x : r.gate; g.ldx = r.ldx ? r.ldx : r.K; g.Wp = r.wp; g.bias = r.b; g.N = r.N; g.K = r.K;
    g.resid = r.resid; g.ldr = r.N; g.y = r.y; g.ldy = r.ldy ? r.ldy : r.qkv ? r.N / 3 : r.N; g.glu_pair = r.glu_pair; g.relu = r.relu;
    g.ln_w = r.ln_w; g.ln_b = r.ln_b; g.xn = r.xn; g.ln2_w = r.ln2_w; g.ln2_b = r.ln2_b;
    g.rope = r.rope; g.rope_cols = r.rope_cols; g.rope_dim = r.rope_dim; g.pos = r.pos;
    if (r.qkv) { g.mode = 1; g.kcache = r.kc; g.vcache = r.vc; g.H = r.H; g.hd = r.hd; g.cap = r.cap; g.d = r.N / 3; }
    g.glu_gate = r.gate; g.glu_only = r.gate && !r.x;
    g.n_groups = r.groups; g.x_group_off = r.x_goff; g.y_group_off = r.y_goff; g.sel_w_stride = r.w_gstride; g.sel_b_stride = r.b_gstride;
    return amt_launch_decode_gemm(g, s);
}

}  // namespace

// embed: compute the chord-stream rows of this position at the head of the chain (false: the previous call's fused decision left them
// in ws); advance: increment the position at the tail (false: the fused decision does it)
static thread_local int g_last_step_launches = 0;     // launches issued by this thread's last lockstep step (introspection for bench.py)

static int32_t v2_step_batch_impl(const void* const* tab, int32_t n_layers, int32_t H, int32_t E, int32_t dff, int32_t n_exp,
                                  int32_t S, int32_t max_seq, int32_t B, const float* keys_dev, int32_t* state_dev,
                                  float* logits_out, float* ws, void* stream, bool embed, bool advance) {
    AMT_CHECK_ARG(tab && logits_out && ws && keys_dev && state_dev, "amt_v2_step_batch: null pointer");
    AMT_CHECK_ARG(n_layers > 0 && H > 0 && E % H == 0 && E % 64 == 0 && dff % 64 == 0 && E <= 1536 && dff <= 1536 && S > 0 && max_seq > 0,
                  "amt_v2_step_batch: bad shape (E and dff must be multiples of 64, at most 1536)");
    AMT_CHECK_ARG(B >= 1 && B <= 256 && n_exp >= 0 && n_exp <= 64, "amt_v2_step_batch: 1..256 clips, at most 64 experts");
    hipStream_t s = (hipStream_t)stream;
    const int hd = E / H;
    const float qscale = 1.0f / sqrtf((float)hd);
    auto G = [&](int i) { return (const float*)tab[i]; };
    const size_t BE = (size_t)B * E;
    float* x = ws; float* y = x + BE; float* qkv = y + BE; float* q = qkv + 3 * BE; float* o = q + BE; float* u = o + BE;
    float* ysh = u + BE;                                  // shared expert output [B][E]
    float* xa = ysh + BE; float* xb = xa + BE; float* xc = xb + BE;      // LayerNorm outputs written by the fused prologues
    float* Yall = xc + BE;                                // every expert's output [n_exp][B][E]
    float* ffs = Yall + (size_t)(n_exp + 1) * BE;         // 3 * B * (max(n_exp, 1) + 1) * dff expert scratch (gate | up of every expert + shared)
    const int* pos = state_dev;
    int32_t rc;
    int n_launch = 0;                                   // kernel launches of this call (every helper below issues exactly one)
    auto CNT = [&](int32_t r) { ++n_launch; return r; };
    if (embed) {
        ++n_launch;
        hipLaunchKernelGGL(embed_rows_kernel, dim3(B), dim3(128), 0, s, state_dev, B, keys_dev, G(G_PR), G(G_PA), G(G_WKEY), G(G_CBIAS), x, E, G(G_PE));
        AMT_LAUNCH_CHECK();
    }
    const float* rope = G(G_ROPE);
    // A LayerNorm that feeds a projection runs in that projection's prologue (E <= 1024; RMSNorm models keep their own launch).
    // `cur` holds either finished rows (pend_w == null) or the pre-norm sum that the pending LayerNorm (pend_w, pend_b) completes.
    const bool fuse_ln = E <= 1024;
    float* cur = x;
    const float *pend_w = nullptr, *pend_b = nullptr;
    bool raw_qkv = false;                               // `qkv` holds the raw QKV product of the pending norm3 (folded down projection)
    const float *raw_g = nullptr, *raw_c = nullptr;
    // finishes a pending norm into `dst` by its own launch (RMSNorm, or nothing to fuse it into)
    auto settle = [&](float* dst) -> int32_t {
        if (!pend_w) return 0;
        int32_t r2 = CNT(norm_rows(cur, nullptr, pend_w, pend_b, dst, B, E, s));
        cur = dst; pend_w = pend_b = nullptr;
        return r2;
    };
    for (int l = 0; l < n_layers; ++l) {
        const void* const* L = tab + G_PTRS + (size_t)l * L_PTRS;
        auto P = [&](int i) { return (const float*)L[i]; };
        float* kc = (float*)L[L_KC]; float* vc = (float*)L[L_VC];                 // [B][H][max_seq][hd]
        // ---- self-attention: [pending LayerNorm ->] QKV projection -> rotary -> q / cache rows, one launch ----
        if (raw_qkv) {
            // the previous layer's down projection left qkv_raw = u3 (Wqkv o gamma3)^T: the attention kernel finishes q / k / v with
            // u3's row statistics, rotates q and k, appends the cache row and publishes norm3(u3) (this layer's residual stream)
            AttnDecodeParams a{};
            a.k = kc; a.v = vc; a.k_new = kc; a.v_new = vc; a.o = o; a.B = B; a.H = H; a.hd = hd; a.cap = max_seq; a.pos = pos; a.new_kv = 1;
            a.q = qkv; a.ldq = 3 * E; a.d = E; a.fold_u = cur; a.fold_g = raw_g; a.fold_c = raw_c; a.fold_lnw = pend_w; a.fold_lnb = pend_b;
            a.xn = xa; a.eps = 1e-5f; a.q_scale = qscale; a.rope = rope; a.rope_dim = E; a.rope_pos = pos;
            if ((rc = CNT(amt_launch_attn_decode(a, s)))) return rc;
            cur = xa; pend_w = pend_b = nullptr; raw_qkv = false;
        } else {
        if (pend_w && !(fuse_ln && pend_b)) { if ((rc = settle(xa))) return rc; }
        {
            RowGemm r; r.x = cur; r.wp = P(L_SAW); r.b = P(L_SAB); r.y = q; r.N = 3 * E; r.K = E;
            if (pend_w) { r.ln_w = pend_w; r.ln_b = pend_b; r.xn = xa; }
            r.rope = rope; r.rope_cols = 2 * E; r.rope_dim = E; r.pos = pos; r.scale = qscale; r.scale_cols = E;
            r.qkv = true; r.kc = kc; r.vc = vc; r.H = H; r.hd = hd; r.cap = max_seq;
            if ((rc = CNT(row_gemm(r, B, s)))) return rc;
            if (pend_w) { cur = xa; pend_w = pend_b = nullptr; }
        }
        if ((rc = CNT(attn_rows(q, kc, vc, o, B, H, hd, max_seq, 0, pos, s)))) return rc;
        }
        if (P(L_G1P) && fuse_ln && P(L_N1B)) {
            // u = out-proj + residual AND the raw query product of norm1(u) in one launch; the cross-attention finishes the query
            // (row statistics of u, rotary, scale) in its prologue and publishes norm1(u) as the residual of its out-projection
            DecodeGemmParams g{};
            g.B = B; g.eps = 1e-5f; g.scale = 1.f; g.x = o; g.ldx = E; g.x2 = cur; g.ldx2 = E; g.K1 = E; g.K = 2 * E;
            g.Wp = P(L_SAOW); g.bias = P(L_SAOB); g.resid = cur; g.ldr = E; g.y = u; g.ldy = E;
            g.n_split = E; g.N = 2 * E; g.Wp2 = P(L_G1P); g.bias2 = P(L_G1B); g.y2 = q; g.ldy2 = E;
            if ((rc = CNT(amt_launch_decode_gemm(g, s)))) return rc;
            AttnDecodeParams a{};
            a.k = P(L_KX); a.v = P(L_VX); a.o = o; a.B = B; a.H = H; a.hd = hd; a.cap = S; a.n_keys = S;
            a.q = q; a.ldq = E; a.d = E; a.fold_u = u; a.fold_g = P(L_FQG); a.fold_c = P(L_FQC); a.fold_lnw = P(L_N1W); a.fold_lnb = P(L_N1B);
            a.xn = xb; a.eps = 1e-5f; a.q_scale = qscale; a.rope = rope; a.rope_dim = E; a.rope_pos = pos;
            if ((rc = CNT(amt_launch_attn_decode(a, s)))) return rc;
            cur = xb; pend_w = pend_b = nullptr;
        } else {
        if ((rc = CNT(lin_rows(o, P(L_SAOW), P(L_SAOB), cur, u, B, E, E, s)))) return rc;           // u = out-proj + residual
        cur = u; pend_w = P(L_N1W); pend_b = P(L_N1B);
        // ---- cross-attention: [norm1 ->] query projection -> rotary, scale ----
        if (!(fuse_ln && pend_b)) { if ((rc = settle(xb))) return rc; }
        {
            RowGemm r; r.x = cur; r.wp = P(L_CAW); r.b = P(L_CAB); r.y = q; r.N = E; r.K = E;
            if (pend_w) { r.ln_w = pend_w; r.ln_b = pend_b; r.xn = xb; }
            r.rope = rope; r.rope_cols = E; r.rope_dim = E; r.pos = pos; r.scale = qscale; r.scale_cols = E;
            if ((rc = CNT(row_gemm(r, B, s)))) return rc;
            if (pend_w) { cur = xb; pend_w = pend_b = nullptr; }
        }
        if ((rc = CNT(attn_rows(q, P(L_KX), P(L_VX), o, B, H, hd, S, S, nullptr, s)))) return rc;      // [B][H][S][hd]
        }
        if (P(L_G2P) && fuse_ln && P(L_N2B) && !L[L_GATEW]) {
            // plain GLU layer with norm2 and norm3 folded (5 launches per layer like the base model's chain): the cross-attention's
            // out-projection also emits the raw stacked gate | up product of the pre-norm sum; the down projection finishes both
            // halves with that sum's row statistics, applies up * silu(gate), adds norm2(u2) and emits the next layer's raw QKV
            float* GUr = ffs;                                // [B][2 dff]: raw gate columns, then raw up columns
            DecodeGemmParams g{};
            g.B = B; g.eps = 1e-5f; g.scale = 1.f; g.x = o; g.ldx = E; g.x2 = cur; g.ldx2 = E; g.K1 = E; g.K = 2 * E;
            g.Wp = P(L_CAOW); g.bias = P(L_CAOB); g.resid = cur; g.ldr = E; g.y = y; g.ldy = E;
            g.n_split = E; g.N = E + 2 * dff; g.Wp2 = P(L_G2P); g.bias2 = P(L_G2B); g.y2 = GUr; g.ldy2 = 2 * dff;
            if ((rc = CNT(amt_launch_decode_gemm(g, s)))) return rc;
            DecodeGemmParams d3{};
            d3.B = B; d3.eps = 1e-5f; d3.scale = 1.f; d3.pro = 2; d3.x = GUr + dff; d3.glu_gate = GUr; d3.ldx = 2 * dff; d3.x2 = y; d3.ldx2 = E;
            d3.K1 = dff; d3.K = dff + E; d3.fold_g = P(L_FGUG) + dff; d3.fold_c = P(L_FGUC) + dff; d3.fold_g2 = P(L_FGUG); d3.fold_c2 = P(L_FGUC);
            d3.ln_w = P(L_N2W); d3.ln_b = P(L_N2B); d3.Wp = P(L_W2); d3.bias = P(L_B2); d3.y = u; d3.ldy = E; d3.n_split = E; d3.N = E;
            if (P(L_G3P) && l + 1 < n_layers) {
                d3.N = 4 * E; d3.Wp2 = P(L_G3P); d3.bias2 = P(L_G3B); d3.y2 = qkv; d3.ldy2 = 3 * E;
                raw_qkv = true; raw_g = P(L_FKG); raw_c = P(L_FKC);
            }
            if ((rc = CNT(amt_launch_decode_gemm(d3, s)))) return rc;
            cur = u; pend_w = P(L_N3W); pend_b = P(L_N3B);
            continue;
        }
        if ((rc = CNT(lin_rows(o, P(L_CAOW), P(L_CAOB), cur, y, B, E, E, s)))) return rc;
        cur = y; pend_w = P(L_N2W); pend_b = P(L_N2B);
        // ---- feed-forward: [norm2 ->] gate projection; up projection; down projection with the gate applied in its prologue ----
        if (!(fuse_ln && pend_b)) { if ((rc = settle(xc))) return rc; }
        // gate and up projections of the block are ONE product over the stacked matrix [gate | linear1] (of every expert and the
        // shared one in a mixture layer), with norm2 in its prologue.  The stacked matrix is packed with its rows interleaved in eights
        // (tile T = gate columns 8T..8T+7 | up columns 8T..8T+7), so that launch's epilogue writes the hidden rows h = up * silu(gate)
        // themselves, each element once (SiLU experts without an up projection: h = silu(.)); the down projection(s) are plain products.
        // (Round 2 applied the gate in the down projections' prologue: every one of their 128 / 448 workgroups redid 16 x dff
        //  exp / rcp on its four SIMDs, ~2 us of an 8 us launch.)
        const float* ffin;                                   // the normalised rows every projection of the block reads
        const bool has_up = P(L_W1) != nullptr;
        const int ng = L[L_GATEW] ? n_exp + (L[L_SWG] ? 1 : 0) : 1;
        const int Nall = ng * dff, Ngu = has_up ? 2 * Nall : Nall;
        float* Hh = ffs;                                     // [B][Nall]: hidden rows of every expert (+ shared)
        {
            RowGemm rg; rg.x = cur; rg.wp = P(L_GU); rg.b = P(L_GUB); rg.y = Hh; rg.N = Ngu; rg.K = E; rg.ldy = Nall;
            if (has_up) rg.glu_pair = 1; else rg.relu = 2;
            if (pend_w) { rg.ln_w = pend_w; rg.ln_b = pend_b; rg.xn = xc; }
            if ((rc = CNT(row_gemm(rg, B, s)))) return rc;
            if (pend_w) { cur = xc; pend_w = pend_b = nullptr; }
            ffin = cur;
        }
        if (!L[L_GATEW] && P(L_G3P) && fuse_ln && P(L_N3B) && has_up && l + 1 < n_layers) {
            // u = expert(x) + x AND the next layer's raw QKV product of norm3(u) in one launch (norm3 folded through that projection,
            // the base model's G3): rows [h | x], low columns through linear2 over the gated half (+ x), high columns
            // through [(Wqkv' o gamma3) W2 | Wqkv' o gamma3]; the next self-attention finishes q / k / v with u's row statistics
            DecodeGemmParams d3{};
            d3.B = B; d3.eps = 1e-5f; d3.scale = 1.f; d3.x = Hh; d3.ldx = Nall; d3.x2 = ffin; d3.ldx2 = E;
            d3.K1 = dff; d3.K = dff + E; d3.Wp = P(L_W2); d3.bias = P(L_B2); d3.resid = ffin; d3.ldr = E; d3.y = u; d3.ldy = E;
            d3.n_split = E; d3.N = 4 * E; d3.Wp2 = P(L_G3P); d3.bias2 = P(L_G3B); d3.y2 = qkv; d3.ldy2 = 3 * E;
            if ((rc = CNT(amt_launch_decode_gemm(d3, s)))) return rc;
            raw_qkv = true; raw_g = P(L_FKG); raw_c = P(L_FKC);
        } else if (!L[L_GATEW]) {
            RowGemm rd; rd.x = Hh; rd.ldx = Nall; rd.wp = P(L_W2); rd.b = P(L_B2); rd.resid = ffin;
            rd.y = u; rd.N = E; rd.K = dff;
            if ((rc = CNT(row_gemm(rd, B, s)))) return rc;             // u = expert(x) + x : the pre-norm sum of norm3
        } else {
            // the down projections of all experts and the shared one in ONE grouped launch (blockIdx.z = expert)
            RowGemm rd; rd.x = Hh; rd.ldx = Nall; rd.wp = P(L_W2S); rd.b = P(L_B2S); rd.y = Yall; rd.N = E; rd.K = dff;
            rd.groups = ng; rd.x_goff = (size_t)dff; rd.y_goff = BE; rd.w_gstride = (size_t)E * dff; rd.b_gstride = E;
            if ((rc = CNT(row_gemm(rd, B, s)))) return rc;
            const float* shared = L[L_SWG] ? Yall + (size_t)n_exp * BE : nullptr;
            // top-2 routing of every row and the weighted sum of its two experts (+ shared / 2 + residual) in one launch
            if ((rc = CNT(amt_launch_moe_route_combine(ffin, P(L_GATEW), P(L_GATEB), n_exp, Yall, shared, 0.5f, ffin, u, B, E, s)))) return rc;
        }
        cur = u; pend_w = P(L_N3W); pend_b = P(L_N3B);
        // (the next layer's QKV launch consumes `u` before that layer's out-projection writes it again)
        if (l + 1 < n_layers && !(fuse_ln && pend_b)) { if ((rc = settle(x))) return rc; }
    }
    if (fuse_ln && pend_w && pend_b && G(G_FNB)) {                        // norm3 of the last layer AND decoder.norm in the head's prologue
        RowGemm r; r.x = cur; r.wp = G(G_WOUT); r.b = G(G_BOUT); r.y = logits_out; r.N = 159; r.K = E; r.ln_w = pend_w; r.ln_b = pend_b;
        r.ln2_w = G(G_FNW); r.ln2_b = G(G_FNB);
        if ((rc = CNT(row_gemm(r, B, s)))) return rc;
        pend_w = pend_b = nullptr;
    } else if ((rc = settle(x))) {                                        // norm3 of the last layer
        return rc;
    } else if (fuse_ln && G(G_FNB)) {                                     // decoder.norm in the prologue of the output head
        RowGemm r; r.x = cur; r.wp = G(G_WOUT); r.b = G(G_BOUT); r.y = logits_out; r.N = 159; r.K = E; r.ln_w = G(G_FNW); r.ln_b = G(G_FNB);
        if ((rc = CNT(row_gemm(r, B, s)))) return rc;
    } else {
        if ((rc = CNT(norm_rows(cur, nullptr, G(G_FNW), G(G_FNB), y, B, E, s)))) return rc;
        if ((rc = CNT(lin_rows(y, G(G_WOUT), G(G_BOUT), nullptr, logits_out, B, 159, E, s)))) return rc;
    }
    if (advance) {
        ++n_launch;
        hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, s, state_dev);
        AMT_LAUNCH_CHECK();
    }
    g_last_step_launches = n_launch;
    return 0;
}

extern "C" int32_t amt_v2_step_batch(const void* const* tab, int32_t n_layers, int32_t H, int32_t E, int32_t dff, int32_t n_exp,
                                     int32_t S, int32_t max_seq, int32_t B, const float* keys_dev, int32_t* state_dev,
                                     float* logits_out, float* ws, void* stream) {
    return v2_step_batch_impl(tab, n_layers, H, E, dff, n_exp, S, max_seq, B, keys_dev, state_dev, logits_out, ws, stream, true, true);
}

extern "C" int32_t amt_v2_step_decide_batch(const amt_v2_step_args* st, const amt_v2_decide_args* dc, int32_t first, void* stream) {
    AMT_CHECK_ARG(st && dc, "amt_v2_step_decide_batch: null argument block");
    AMT_CHECK_ARG(dc->T <= st->max_seq, "amt_v2_step_decide_batch: T=%d exceeds the caches' %d rows", dc->T, st->max_seq);
    int32_t rc = v2_step_batch_impl(st->tab, st->n_layers, st->H, st->E, st->dff, st->n_exp, st->S, st->max_seq, st->B, st->keys_dev, st->state_dev,
                                    st->logits_out, st->ws, stream, first != 0, false);
    if (rc) return rc;
    auto G = [&](int i) { return (const float*)st->tab[i]; };
    g_last_step_launches += 1;                          // + the decision kernel
    return amt_launch_v2_decide_fused(st->logits_out, 159, st->state_dev, dc->tokens, dc->roots, dc->attrs, st->B, dc->T, dc->n_primer, dc->beam,
                                      dc->max_conseq_N, dc->max_conseq_chord, dc->temperature, dc->uniforms, dc->chord_embed, st->keys_dev, G(G_PR),
                                      G(G_PA), G(G_WKEY), G(G_CBIAS), G(G_PE), st->ws, st->E, (hipStream_t)stream);
}

extern "C" int32_t amt_v2_last_step_launches(void) { return g_last_step_launches; }
