// Internal launcher declarations shared by the C-ABI layer (amt_api.hip) and the kernel files.
#pragma once
#include "amt_common.h"

// ---------------- dense GEMM (gemm_f32.hip) ----------------
struct GemmParams {
    const float* A; int lda;        // [M][K] activations
    const float* W; int ldw;        // [N][K] weights (nn.Linear layout)
    float* C; int ldc;              // [M][N] output (plain mode)
    int M, N, K;
    const float* bias;              // [N] or null
    const float* resid; int ldr;    // [M][N] residual added after bias/scale, or null
    const float* rowadd; int rowadd_period;   // [period][N] row-periodic addend (positional encoding), or null
    float scale; int scale_cols;    // columns [0,scale_cols) are multiplied by scale after the bias (q * hd^-0.5)
    int relu;                       // 1: v = max(v, 0); 2: v = silu(v)
    int sigmoid;                    // v = 1/(1+exp(-v)) last (classifier heads)
    // head-split store: row=b*seq+s, col=part*d+h*hd+c -> C[part*part_stride + ((b*heads+h)*seq_cap+s)*hd+c]
    int head_split, hs_seq, hs_seq_cap, hs_d, hs_hd, hs_heads;
    size_t hs_part_stride;
    // grouped (mixture-of-experts) mode: the 128-row tile tm uses weight group tile_group[tm] (< 0: tile unused)
    const int* tile_group; size_t w_group_stride, bias_group_stride;
    const int* a_gather;            // A row of output row m is a_gather[m] (< 0: zero row), or null
    const float* silu_mul; int ld_silu;   // v *= silu(silu_mul[row][col]) after the bias (GLU gate), or null
};
static inline GemmParams gemm_params(const float* A, int lda, const float* W, int ldw, float* C, int ldc,
                                     int M, int N, int K, const float* bias) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.scale = 1.f; p.rowadd_period = 1;
    return p;
}
int32_t amt_launch_gemm(const GemmParams& p, hipStream_t stream);

// ---------------- normalisation / elementwise (norm.hip) ----------------
// y = LayerNorm(x (+ resid)) * w + b ; optional second LayerNorm (w2,b2) applied on top (decoder.norm)
// `post` (optional, [rows][dim]) is added to the normalised rows: y = LN(x + resid) + post
int32_t amt_launch_layernorm(const float* x, const float* resid, const float* w, const float* b,
                             const float* w2, const float* b2, float* y, int rows, int dim, float eps,
                             hipStream_t stream, const float* post = nullptr);
int32_t amt_launch_rmsnorm(const float* x, const float* w, float* y, int rows, int dim, float eps, hipStream_t stream,
                           const float* resid = nullptr);
// y = RMSNorm_hd(o1 - lambda * o2) * w * out_scale over rows of hd <= 128 (differential attention, custom_transformer.py:818-826)
int32_t amt_launch_diff_subln(const float* o1, const float* o2, const float* w, float* y, int rows, int hd, float lambda,
                              float out_scale, float eps, hipStream_t stream);
int32_t amt_launch_add(const float* a, const float* b, float* y, long n, hipStream_t stream);
int32_t amt_launch_row_scale_add(const float* x, const float* row_scale, const float* add, float* y, int rows, int dim, hipStream_t stream);
// rotary embedding on interleaved pairs: x viewed as [n0][seq][n2][hd], cache [>=seq][cache_half][2];
// reproduces the reference's view(-1, seq, 1, hd/2, 2)[:n0] reinterpretation of the cache
int32_t amt_launch_rope(const float* x, const float* cache, float* y, int n0, int seq, int n2, int hd,
                        int cache_half, hipStream_t stream);
// vf_concat[b*S+s][0:Fpad] = [sem | scene | motion | emotion | 0-pad]
int32_t amt_launch_concat_features(const float* sem, int sem_dim, const float* scene, const float* motion, int motion_dim,
                                   const float* emotion, int emo_dim, float* out, int rows, int ld_out, hipStream_t stream);
// xf[b*L+l] = PR[root] + PA[attr] + key[b]*wkey + bias + pe[l]
int32_t amt_launch_chord_embed(const int64_t* root, const int64_t* attr, const float* key, const float* PR, const float* PA,
                               const float* wkey, const float* bias, const float* pe, float* out,
                               int B, int L, int d, hipStream_t stream);

// ---------------- attention, prefill (attn_prefill.hip) ----------------
struct AttnParams {
    const float* q; const float* k; const float* v; float* o;
    // element strides: tensor[b][h][l][c] = base + b*bs + h*hs + l*ls + c
    size_t q_bs, q_hs, q_ls, k_bs, k_hs, k_ls, v_bs, v_hs, v_ls, o_bs, o_hs, o_ls;
    int B, H, Lq, Lk, hd;
    int causal;                 // keys j > query i masked
    const float* Er; int er_len;   // relative position table [er_len][hd] or null
    int kv_group;               // query head h uses kv head h / kv_group (GQA); 1 for MHA
    float mask_value;           // -inf (additive mask semantics) or finfo.min (masked_fill semantics)
    float q_scale;              // multiplies Q on load (0 = unset = 1): torch MHA scales q by hd^-0.5 after the in-projection
};
int32_t amt_launch_attn_prefill(const AttnParams& p, hipStream_t stream);

// ---------------- attention, decode (attn_decode.hip) ----------------
struct AttnDecodeParams {
    const float* q;             // [B][H*hd] (already scaled)
    const float* k; const float* v;   // [B][H][cap][hd]
    float* o;                   // [B][H*hd]
    int B, H, hd, cap;
    const int* pos;             // device: index of the query token (keys 0..pos); null -> n_keys fixed
    int n_keys;                 // used when pos == null (cross-attention: S)
    const float* Er; int er_len;
    // Folded-LayerNorm prologue (fold_u != null).  The producing GEMM ran on the *un-normalised* sum u and left
    // raw = u . (W o gamma)^T in q (row stride ldq; the head's query at column h*hd, and for self-attention its new
    // key / value at d + h*hd and 2d + h*hd).  With the row statistics (mu, rstd) of u[b] the projection of
    // LayerNorm(u) is  (raw - mu*fold_g) * rstd + fold_c ; the query is then multiplied by q_scale.  Head 0 also
    // publishes LayerNorm(u[b]) to xn (the residual of the following block).  new_kv: the key/value of position
    // *pos come from the prologue (and are written to the cache here); the cache holds keys 0..pos-1 only.
    const float* fold_u; const float* fold_g; const float* fold_c; const float* fold_lnw; const float* fold_lnb;
    float* xn; int ldq, d, new_kv; float eps, q_scale;
    float* k_new; float* v_new;   // the (writable) cache when new_kv
    unsigned long long* stamps;   // diagnostic builds only (-DAMT_STAMPS): [workgroup][8] s_memrealtime stamps; null in the library
    // rotary embedding of the folded query: table [positions][rope_dim] of interleaved (cos, sin), the position in device memory
    // (rope_pos; with new_kv = 1 the position is *pos and the new key is rotated too); column n of the d_model-wide query uses
    // entries (n % rope_dim) & ~1 and that + 1 (decode_gemm's rotary epilogue)
    const float* rope; int rope_dim; const int* rope_pos;
};
int32_t amt_launch_attn_decode(const AttnDecodeParams& p, hipStream_t stream);

// ---------------- decode-step skinny GEMM (decode_gemm.hip) ----------------
// packed weight: tiles of 16(n) x 16(k): P[((nt*(K/16)+kt)*64 + lane)*4 + e] = W[nt*16+(lane&15)][kt*16+4*(lane>>4)+e]
int32_t amt_launch_pack_weight(const float* W, float* P, int N, int K, hipStream_t stream, int ldw = 0);   // ldw: W's row stride (0 = K)
struct DecodeGemmParams {
    const float* x; int ldx;    // [B][K] input rows (pre-LN sum when ln_w != null)
    const float* Wp;            // packed weight
    const float* bias;          // [N]
    int B, N, K;
    // LayerNorm prologue (optional); normalised rows are also written to xn (ld = K)
    const float* ln_w; const float* ln_b; float* xn; float eps;
    // epilogue
    int mode;                   // 0: y = acc+bias (+resid) (relu) ; 1: packed-QKV split into q / K-cache / V-cache
    const float* resid; int ldr;
    int relu;                   // 1: max(v, 0); 2: silu(v)
    float scale; int scale_cols;
    float* y; int ldy;
    // mode 1
    float* kcache; float* vcache; int H, hd, cap; const int* pos; int d;
    // Two-source input rows: columns [0,K1) come from x, [K1,K) from x2 (K1 == 0 or K: single source)
    const float* x2; int ldx2, K1;
    // pro == 1 (FFN-down with LayerNorm folded through FFN-up): x holds raw = u . (W1 o gamma)^T [B][K1], x2 the
    // pre-LN sum u [B][K-K1]; the staged row is [ relu((raw - mu*fold_g)*rstd + fold_c) | LayerNorm(u) ] with
    // (mu, rstd) the statistics of u's row; ln_w / ln_b are that LayerNorm's affine.  The LayerNorm half is also
    // the residual of the columns below n_split.
    int pro; const float* fold_g; const float* fold_c;
    // pro == 2 (gated FFN-down with LayerNorm folded through the stacked gate | up product, the V1 / V2 plain GLU layers): x holds
    // the raw UP columns, glu_gate the raw GATE columns (same row stride; both [B][K1]), x2 the pre-LN sum u; with the statistics of
    // u's row  up = (x - mu*fold_g)*rstd + fold_c,  gate = (glu_gate - mu*fold_g2)*rstd + fold_c2  and the staged row is
    // [ up * silu(gate) | LayerNorm(u) ]  (GLUExpert.forward, moe.py:44-49, over LayerNorm(u)); K1 % 256 == 0.
    const float* fold_g2; const float* fold_c2;
    // Column split: output columns [0,n_split) use the packed weight Wp over the first K1 input columns only
    // (bias, residual, ReLU as in mode 0) and go to y; columns [n_split,N) use Wp2 over all K columns, get
    // bias2[n - n_split] only and go to y2[row*ldy2 + n - n_split].  n_split == 0: no split.
    const float* Wp2; const float* bias2; float* y2; int ldy2, n_split;
    // weight group chosen on the device (one token of a mixture-of-experts layer): Wp / Wp2-less launches only;
    // the packed weight of group *sel starts sel_w_stride floats further, its bias sel_b_stride floats further
    const int* sel; size_t sel_w_stride; int sel_b_stride;
    int ldw;                    // > 0: Wp is NOT packed but a plain nn.Linear weight [N][ldw] (small-M products of the dense paths)
    // Grouped launch (n_groups > 1, blockIdx.z = group; Wp-only launches): group e reads its input rows at x + e*x_group_off
    // (same ldx), its packed weight at Wp + e*sel_w_stride, its bias at bias + e*sel_b_stride, and writes y + e*y_group_off:
    // the down-projections of all experts of a mixture layer in one launch
    int n_groups; size_t x_group_off, y_group_off;
    // Gated-linear-unit prologue (glu_gate != null; single source, no LayerNorm): the staged row is x * silu(glu_gate)
    // (GLUExpert.forward, moe.py:44-49), or silu(glu_gate) alone when glu_only (the Linear -> SiLU -> Linear experts of V1);
    // glu_gate has x's row stride and group offset
    const float* glu_gate; int glu_only;
    // Gated-linear-unit EPILOGUE (glu_pair = 1; round 3): the packed weight is the stacked [gate | up] matrix with its rows interleaved in
    // eights -- column tile T = [gate columns 8T..8T+7 | up columns 8T..8T+7] -- so a tile holds both halves of 8 hidden columns and
    // the launch writes h = up * silu(gate) once per element: y[row*ldy + 8T + c], N/2 columns (N = the stacked width; `bias` keeps the
    // stacked order [gate (N/2) | up (N/2)]).  The gated PROLOGUE above redoes that product in every workgroup of the consuming launch
    // (16 x dff exp / rcp per workgroup on four SIMDs: ~2 us of an 8 us launch).  relu == 2: y = silu(acc + bias) (SiLU experts).
    int glu_pair;
    // Rotary epilogue (rope != null): columns n < rope_cols are rotated as interleaved pairs (2i, 2i+1) by the angles of
    // position *pos in the table rope[pos][rope_dim] = (cos, sin) pairs (custom_transformer.py:1044-1053 as wired: the full
    // d_model vector, pair i by angle i); applied after the bias and before `scale`.  Column n uses table entry n % rope_dim.
    const float* rope; int rope_cols, rope_dim;
    const float* zero;          // set by the launcher: zero words in global memory
    // diagnostic builds only (-DAMT_STAMPS, tools/ubench_chain.cpp): [workgroup][8] s_memrealtime stamps (100 MHz) of the
    // kernel's phases; null and unused in the library build
    const float* ln2_w; const float* ln2_b;   // a second LayerNorm applied to the normalised rows (norm3 of the last layer, then decoder.norm)
    unsigned long long* stamps;
};
int32_t amt_launch_decode_gemm(const DecodeGemmParams& p, hipStream_t stream);
// allocates the per-device zero words (hipMalloc): call once outside any stream capture
int32_t amt_decode_gemm_init();

// mixture-of-experts combine with the layer's residual added (moe.hip): out = sum_e w_e Y[slot_e] (+ shared_scale * shared) + resid;
// dense_B > 0: Y holds every expert's output for every token ([expert][dense_B][d]), slot_pos is not read
int32_t amt_launch_moe_combine(const float* y_rows, const int32_t* slot_pos, const int32_t* idx, const float* wts, const float* shared,
                               float shared_scale, const float* resid, float* out, int n_tok, int d, hipStream_t stream, int dense_B = 0);
// decision + next chord-stream row + position advance of the lockstep V1/V2 step, one launch (sample.hip)
int32_t amt_launch_v2_decide_fused(const float* logits, int ld_logits, int32_t* state_dev, int64_t* tokens, int64_t* roots, int64_t* attrs,
                                   int B, int T, int n_primer, int beam, int max_conseq_N, int max_conseq_chord, float temperature,
                                   const float* uniforms, int chord_embed, const float* keys, const float* PR, const float* PA,
                                   const float* wkey, const float* bias, const float* pe, float* x_next, int d, hipStream_t stream);
// routing (top-2 of x . gate_w^T + gate_b, softmax over the pair) and the combine in one launch; y_all is [expert][n_tok][d]
int32_t amt_launch_moe_route_combine(const float* x, const float* gate_w, const float* gate_b, int n_exp, const float* y_all, const float* shared,
                                     float shared_scale, const float* resid, float* out, int n_tok, int d, hipStream_t stream);

// ---------------- load-time LayerNorm folding (fold.hip) ----------------
int32_t amt_launch_scale_cols(const float* W, const float* gamma, float* out, int N, int K, hipStream_t stream);   // out = W o gamma
int32_t amt_launch_transpose(const float* in, float* out, int R, int C, hipStream_t stream);                     // [R][C] -> [C][R]
// g[n] = sum_k Ws[n][k] ; c[n] = W[n].beta + b[n] ; dv[n] = Ws[n].bo   (fp64 accumulation)
int32_t amt_launch_fold_vectors(const float* W, const float* Ws, const float* beta, const float* b, const float* bo,
                                float* g, float* c, float* dv, int N, int K, hipStream_t stream);

// ---------------- Mamba pieces of the regression head (mamba.hip) ----------------
// y[b][l][c] = silu(bias[c] + sum_j w[c][j] * x[b][l-(K-1)+j][c])   (causal depthwise conv; reverse: the same on the
// time-reversed sequence, results at the original positions)
int32_t amt_launch_dwconv_silu(const float* x, int ldx, const float* w, const float* bias, float* y, int B, int L, int C, int K,
                               int reverse, hipStream_t stream);
struct ScanParams {
    const float* x; int ldx;            // [B*L][ED] conv output
    const float* draw; int ldd;         // [B*L][ED] dt_proj(delta) without bias
    const float* dt_bias;               // [ED]
    const float* A_log;                 // [ED][N]
    const float* Bm; const float* Cm; int ldbc;   // [B*L][N] each (slices of the x_proj output)
    const float* D;                     // [ED]
    const float* z; int ldz;            // [B*L][ED] gate branch (pre-SiLU)
    float* y; int ldy;                  // [B*L][ED] gated output
    int B, L, ED, N, version, reverse;
};
int32_t amt_launch_selective_scan(const ScanParams& p, hipStream_t stream);
// out[row] = [a[row][0:da] | b[row][0:db] | 0-pad to ld_out]
int32_t amt_launch_concat2(const float* a, int da, const float* b, int db, float* out, int rows, int ld_out, hipStream_t stream);

// ---------------- sampling head (sample.hip) ----------------
struct SampleParams {
    const float* u; int ldu;        // [B][d] pre-LN sum of the last decoder layer
    const float* ln_w; const float* ln_b;     // last layer norm3
    const float* fn_w; const float* fn_b;     // decoder.norm
    const float* Wout; const float* bout;     // [159][d], [159]
    float eps;
    int B, d;
    int64_t* tokens; int64_t* roots; int64_t* attrs; int T;   // [B][T] sequences (device)
    int* pos;                        // device step counter (input position); advanced by the last block
    unsigned* ticket;                // device arrival counter (zero between launches)
    int n_primer;                    // positions < n_primer are given, not sampled
    int beam;                        // 0: feedback greedy (G2), 1: verbatim top-1 (G1)
    int max_conseq_N, max_conseq_chord;
    float* logits_out;               // optional [T][B][159] (row pos)
    float* probs_out;                // optional [B][157] decision distribution of this step
    // next-step input embedding
    const float* key; const float* PR; const float* PA; const float* wkey; const float* cbias; const float* pe;
    float* x_next;                   // [B][d]
    int sample_external;             // 1: do not pick a token (host samples from probs_out), only write probs
    // beam == 0 with uniforms != null: draw the token from the decision distribution by inverse CDF with the uniform
    // uniforms[pos*B + b] in [0,1) (the reference's Categorical.sample, :1104-1105); null: arg-max (oracle G2)
    const float* uniforms;
    // Folded output head (lraw != null): the last skinny GEMM already produced lraw = u . (Wout o g3 o gf)^T (ld_lraw
    // floats per clip); with the statistics (mu, rstd) of u and (m2, rstd2) of LayerNorm3(u) the logits are
    //   rstd2 * (rstd * (lraw - mu*h1) + h2 - m2*h3) + h4      (h1..h4: [159] vectors built at weight load)
    const float* lraw; int ld_lraw; const float* h1; const float* h2; const float* h3; const float* h4;
    // Layer-0 QKV of the next position as a table sum (tab_r != null): the decoder input is itself a sum of table
    // rows (root, attr, key, position), so its projection is one too:  qkv = TR[root] + TA[attr] + key*tk + TP[pos]
    // ([.][3d] tables).  q (x q_scale) goes to q0 [B][d], k / v into layer 0's cache row of that position.
    const float* tab_r; const float* tab_a; const float* tab_k; const float* tab_p;
    float* q0; float* kc0; float* vc0; int H, hd, cap; float q_scale;
    int chord_embed;                 // 1: the chosen chord id feeds back as the "root" index (table = chord embedding), attr = 0
};
int32_t amt_launch_sample(const SampleParams& p, hipStream_t stream);
// the base model's layer-0 self-attention with the previous step's sampling decision in its prologue (attn_decode.hip, FOLD 5)
int32_t amt_launch_attn_decode_sample(const AttnDecodeParams& p, const SampleParams& sp, hipStream_t stream);
// writes x_next for position *pos from the token sequences (start of generate / external sampling)
int32_t amt_launch_embed_step(const SampleParams& p, int advance, hipStream_t stream);
