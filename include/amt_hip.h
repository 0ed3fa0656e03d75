/*
 * libamt_hip — C ABI of the MI355X (gfx950) implementation of the Affective Multimodal Transformer
 * forward / generate hot path of khangklj/Video2Music.
 *
 * The reference has no FFI of its own: its seam for this path is the Python nn.Module surface
 *   model/video_music_transformer.py:910-1132   class VideoMusicTransformer (forward, generate)
 *   model/rpr.py:17-455                         TransformerDecoderRPR / MultiheadAttentionRPR / _skew
 *   model/positional_encoding.py:7-23           PositionalEncoding
 *   model/grouped_query_attention.py:172-358    MultiheadGQA
 *   model/moe.py:36-49,150-302                  GLUExpert / MoELayer / SharedMoELayer
 *   model/custom_transformer.py:27-48           RMSNorm
 *   model/rotate_operation.py:50-165            RotaryPositionalEmbeddings
 * Each entry point below names the reference code it replaces.  The Python host module
 * (video2music_amd/model/ python files) binds these with ctypes; INTEGRATION.md shows the stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every function returns int32: 0 = ok, < 0 = bad argument, > 0 = hipError_t;
 *     amt_last_error() returns a thread-local message for the last non-zero return.
 *   - all tensor arguments are raw DEVICE pointers, fp32 (ids int64), row-major, 16-byte aligned;
 *     the caller owns inputs and outputs; weights are copied (and repacked) into library-owned
 *     memory by amt_load_weight, so caller storage may be freed afterwards.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     nothing synchronises the device except amt_finalize / amt_destroy.  amt_generate captures
 *     its decode step into a hipGraph on first use for a given (batch, mode).
 *   - threading: one handle per (process, device); a handle and everything it owns (workspaces, K/V caches, captured graphs,
 *     options) belongs to ONE host thread at a time.  The stateless operator entry points may be called from several host
 *     threads concurrently, each on its own stream, as long as they share no output or scratch buffer.
 *   - global state: the library keeps no mutable global besides (a) the thread-local error message, (b) per-device one-time
 *     initialisations (the > 64 KiB dynamic-LDS opt-in of a kernel, a 256-byte block of zero words), each taken under a mutex,
 *     and (c) the immutable tuning record of csrc/amt_common.h.  The release build reads NO environment variable; a
 *     -DAMT_EXPERIMENT build (tools/ab_build.sh) reads the A/B switches of DESIGN.md once, under std::call_once.
 *   - amt_abi_version() == AMT_ABI_VERSION of the header the caller was built against, or the caller must refuse the library.
 */
#ifndef AMT_HIP_H
#define AMT_HIP_H
#include <stdint.h>

#define AMT_ABI_VERSION 3    /* 2: argument structs for the step / skinny-GEMM calls, options in place of amt_debug_set_skip;
                                3: the lockstep step's stacked gate | linear1 matrix is packed from rows interleaved in eights */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amt_handle amt_handle;

/* Constructor arguments of VideoMusicTransformer (model/video_music_transformer.py:911-914) that
 * shape the computation, plus the batch capacity the reference does not have (it is batch-1). */
typedef struct amt_config {
    int32_t n_layers;            /* encoder and decoder layers (6) */
    int32_t num_heads;           /* 8 */
    int32_t d_model;             /* 512 */
    int32_t dim_feedforward;     /* 1024 */
    int32_t max_sequence_video;  /* 300: rows of positional_encoding_video.pe, key capacity of cross-attention */
    int32_t max_sequence_chord;  /* rows of Er / positional_encoding.pe = longest chord sequence */
    int32_t total_vf_dim;        /* width of the concatenated video feature (generate.py:141-160) */
    int32_t max_batch;           /* clips per call, 1..256: sizes the K/V caches and workspaces (32 = BASELINE.json's per-GPU batch) */
} amt_config;

const char* amt_last_error(void);
int32_t amt_abi_version(void);

/* ---- model lifetime -------------------------------------------------------------------- */
/* Shapes the handle takes (anything else is refused with a message, never computed wrongly): d_model a multiple of 32 with
 * 64 <= d_model <= 1024; head_dim = d_model / num_heads in {16, 32, 64, 128}; dim_feedforward a multiple of 32, <= 8192 (beyond d_model + dim_feedforward = 1536 the decode step runs without folded LayerNorms, beyond
 * dim_feedforward = 1536 its linear2 product runs in column ranges of 1024);
 * max_batch 1..256 clips per decode chain (larger batches are sliced by the caller, video2music_amd/model). */
int32_t amt_create(const amt_config* cfg, amt_handle** out);
int32_t amt_destroy(amt_handle* h);
/* One state_dict entry (reference key names).  `data` may be a host or a
 * device pointer; `shape` has `ndim` entries.  Replaces nn.Module.load_state_dict (generate.py:216). */
int32_t amt_load_weight(amt_handle* h, const char* name, const float* data, int32_t ndim, const int64_t* shape);
/* Validates that every tensor of the hot path is present with the right shape and builds the
 * derived layouts (MFMA-ordered decode weights, Linear_chord tables).  Synchronises the device. */
int32_t amt_finalize(amt_handle* h);

/* ---- VideoMusicTransformer.forward pieces ------------------------------------------------ */
/* Video stream + encoder + per-layer cross-attention K/V (video_music_transformer.py:1005-1033,
 * torch nn.TransformerEncoder).  sem (B,S,sem_dim), scene (B,S), motion (B,S,motion_dim) with
 * motion_dim=1 for the scalar form, emotion (B,S,emo_dim).  Optionally copies the encoder memory
 * (B,S,d) to memory_out.  The handle keeps memory / K / V for amt_prefill and amt_generate. */
int32_t amt_encode(amt_handle* h, int32_t B, int32_t S,
                   const float* sem, int32_t sem_dim, const float* scene,
                   const float* motion, int32_t motion_dim, const float* emotion, int32_t emo_dim,
                   float* memory_out, void* stream);
/* Teacher-forced decoder pass (video_music_transformer.py:984-1001,1027-1044; rpr.py:24-70) over
 * the clips of the last amt_encode: root/attr ids (B,L) int64, key (B) -> logits (B,L,159).
 * layer_out (optional) receives the output of decoder layer `layer_index` (B,L,d). */
int32_t amt_prefill(amt_handle* h, int32_t B, int32_t L, const int64_t* root_ids, const int64_t* attr_ids,
                    const float* key, float* logits_out, float* layer_out, int32_t layer_index, void* stream);

/* ---- VideoMusicTransformer.generate (video_music_transformer.py:1046-1132), batched ------ */
/* Starts a generation over the B clips of the last amt_encode.  primer* are (B,P) int64 when
 * primer_per_clip != 0, else (P) shared by all clips.  beam: 0 = sampling branch, 1 = verbatim
 * top-1 branch.  Resets the KV cache and the device position counter. */
int32_t amt_generate_begin(amt_handle* h, int32_t B, const int64_t* primer, const int64_t* primer_root,
                           const int64_t* primer_attr, int32_t P, int32_t primer_per_clip, const float* key,
                           int32_t T, int32_t beam, int32_t max_conseq_N, int32_t max_conseq_chord, void* stream);
/* Switches the sampling branch (beam=0) of the generation in progress from the arg-max decision to the
 * reference's Categorical draw (video_music_transformer.py:1104-1105), done on device by inverse CDF:
 * uniforms is (T,B) fp32 in [0,1), device memory, copied; the token decided from input position t of clip b
 * is the first id whose cumulative masked probability reaches uniforms[t*B+b] * sum.  null switches back.
 * amt_generate_begin resets to arg-max. */
int32_t amt_generate_set_uniforms(amt_handle* h, const float* uniforms, void* stream);
/* Runs `n_steps` decode steps with the arg-max sampler on device (hipGraph replay; oracle G2 for
 * beam=0, G1 for beam=1).  logits_out (optional) is (T,B,159): row t = logits computed from input
 * position t.  n_steps < 0 runs to the end (T-1 steps in total). */
int32_t amt_generate_run(amt_handle* h, int32_t n_steps, float* logits_out, void* stream);
/* Same steps as amt_generate_run, issued eagerly with a HIP event pair recorded on `stream` around
 * every kernel launch (synchronises once per step).  Accumulates, per kernel class
 * {0: self-attention decode, 1: cross-attention decode, 2: skinny GEMMs, 3: sampling head,
 * 4: an EMPTY event pair per step = the overhead of the two records, to be subtracted}, the
 * summed pair durations (ms) and counts (arrays of 5), and for classes 0/1 the algorithmic K/V bytes
 * (fp32 K and V rows of the keys each launch must read).  Used by bench.py's roofline leg. */
/* Options of the base model that change the input tables (before the first amt_finalize).
 *   "chord_embed" = 1: reference chord_embed=True (model/video_music_transformer.py:931-937, 986-987): the chord id indexes a frozen
 *   table instead of root + attr embeddings.  The caller uploads that table (n_rows, d) under the name "embedding_root.weight" and
 *   an all-zero (16, d) "embedding_attr.weight", passes chord ids as the root ids and zeros as the attr ids; the generated id then
 *   feeds back as the root index in both decision branches. */
/*   "causal_mask" = 0 (any time): amt_prefill runs the decoder self-attention without the subsequent mask (reference
 *   forward(mask=False), :978-982); 1 restores the default.
 *   "decode_chain_plain" = 1 (before the first amt_finalize): the decode step without folded LayerNorms (49 launches instead of
 *   31: model/rpr.py:59-69 operator by operator) -- the chain that shapes outside the fold's range take anyway.
 *   "fuse_sampling_head" = 0 (any time): every step of a captured decode graph ends with its own sampling-head launch (31 launches
 *   per step); 1, the default: inside a graph the head rides in the prologue of the next step's first self-attention (30).
 *   "profile_skip" = 1 | 2 | 3 (any time; results become meaningless): measurement hook of bench.py, leaves the self-attention
 *   (bit 0) and / or cross-attention (bit 1) launches out of the captured decode step, so that what a kernel costs the step is
 *   the difference between two timed generates.  0 restores the real step. */
int32_t amt_set_option(amt_handle* h, const char* name, int32_t value);
/* amt_encode with rows [B*S][d] added to Linear_vis's output before the encoder: reference scene_embed=True
 * (:1016-1027: the scene offset is left out of the feature columns and scene_embedding(offset.int()) is added instead).  The
 * caller keeps the scene column in the features and uploads Linear_vis.weight with a zero column at its position. */
int32_t amt_encode_resid(amt_handle* h, int32_t B, int32_t S, const float* sem, int32_t sem_dim, const float* scene,
                         const float* motion, int32_t motion_dim, const float* emotion, int32_t emo_dim,
                         const float* vis_resid, float* memory_out, void* stream);
int32_t amt_generate_profile(amt_handle* h, int32_t n_steps, double* ms_by_class, int64_t* launches_by_class,
                             int64_t* attn_bytes_by_class, void* stream);
/* One decode step that stops before the decision: writes the decision distribution
 * softmax(logits)[:157] with the suppressions applied, (B,157), for a host-side sampler
 * (torch.multinomial = the reference's Categorical.sample), then amt_generate_commit feeds the
 * chosen ids (B) int64 back. */
int32_t amt_generate_step_probs(amt_handle* h, float* probs_out, void* stream);
int32_t amt_generate_commit(amt_handle* h, const int64_t* chosen, void* stream);
/* Between the steps of a host-driven generation (amt_generate_step_probs / amt_generate_commit): the branch of the following
 * steps, 0 = sampling branch (suppressions applied to the distribution, the committed id feeds back as root / attr), 1 = top-k
 * branch (plain softmax[:157], root / attr of the committed position stay PAD).  This is the per-step choice
 * `random.uniform(0,1) <= beam_chance` of model/video_music_transformer.py:1074-1084; the host keeps the `beam` rows. */
int32_t amt_generate_set_branch(amt_handle* h, int32_t beam);
/* Copies the (B,T) int64 token matrix (PAD=158 beyond the generated length) to tokens_out. */
int32_t amt_generate_end(amt_handle* h, int64_t* tokens_out, void* stream);
/* begin + run(-1) + end. */
int32_t amt_generate(amt_handle* h, int32_t B, const int64_t* primer, const int64_t* primer_root,
                     const int64_t* primer_attr, int32_t P, int32_t primer_per_clip, const float* key,
                     int32_t T, int32_t beam, int32_t max_conseq_N, int32_t max_conseq_chord,
                     int64_t* tokens_out, float* logits_out, void* stream);
/* Timing/roofline introspection for bench.py: algorithmic HBM bytes of the decode-attention
 * launches of one step at key count n_keys (self) — see DESIGN.md. */
int64_t amt_decode_step_bytes(const amt_handle* h, int32_t B, int32_t n_self_keys, int32_t S);

/* ---- stateless operator entry points (used by the parity tests and standalone modules) --- */
/* y[M,N] = x[M,K] . w[N,K]^T + bias (+ resid) ; optional ReLU.  torch.nn.functional.linear. K % 32 == 0. */
int32_t amt_linear_fwd(const float* x, const float* w, const float* bias, const float* resid, float* y,
                       int32_t M, int32_t N, int32_t K, int32_t relu, void* stream);
/* y = LayerNorm(x (+ resid)) (torch.nn.LayerNorm; rpr.py:59-69). */
int32_t amt_layernorm_fwd(const float* x, const float* resid, const float* w, const float* b, float* y,
                          int32_t rows, int32_t dim, float eps, void* stream);
/* RMSNorm.forward (custom_transformer.py:38-45); w may be null. */
int32_t amt_rmsnorm_fwd(const float* x, const float* w, float* y, int32_t rows, int32_t dim, float eps, void* stream);
/* Tail of DifferentialMultiheadAttention (custom_transformer.py:818-826; VideoMusicTransformer_V3): with o1 / o2 the
 * attention outputs of the even / odd heads of a pair over the same values, y = RMSNorm_hd(o1 - lambda_full * o2) * w *
 * out_scale (out_scale = 1 - lambda_init), rows of hd <= 128 values. */
int32_t amt_diff_subln_fwd(const float* o1, const float* o2, const float* w, float* y, int32_t rows, int32_t hd,
                           float lambda_full, float out_scale, float eps, void* stream);
/* y = a + b over n floats (n % 4 == 0): the residual add of the pre-norm layers (custom_transformer.py:1241-1249). */
int32_t amt_add_fwd(const float* a, const float* b, float* y, int64_t n, void* stream);
/* y[r][:] = x[r][:] * row_scale[r] (+ add[r][:] when add != null): the video rows dropped by dropTokenRate in the V1 / V2 / V3
 * classes (model/video_music_transformer.py:193-197, 488-492, 798-802: vf * (rand(B,S) > rate), also in eval mode), with the
 * positional rows that follow (:208) as `add`.  dim a multiple of 4. */
int32_t amt_row_scale_add_fwd(const float* x, const float* row_scale, const float* add, float* y, int32_t rows, int32_t dim,
                              void* stream);
/* RMSNorm(x + resid): the post-norm residual form of the custom layers (custom_transformer.py:1233-1240); resid may be null. */
int32_t amt_rmsnorm_resid_fwd(const float* x, const float* resid, const float* w, float* y, int32_t rows, int32_t dim,
                              float eps, void* stream);
/* RotaryPositionalEmbeddings.forward (rotate_operation.py:111-165), input_pos=None: x is
 * (n0, seq, n2, hd); cache is the module's (max_seq, cache_half, 2) buffer. */
int32_t amt_rope_fwd(const float* x, const float* cache, float* y, int32_t n0, int32_t seq, int32_t n2, int32_t hd,
                     int32_t cache_half, void* stream);
/* Core of multi_head_attention_forward_rpr (rpr.py:387-414) after the projections: q (already
 * scaled), k, v are (B,L,H*hd) row-major; Er (er_len,hd); causal.  o (B,L,H*hd). */
int32_t amt_rpr_attn_fwd(const float* q, const float* k, const float* v, const float* Er, float* o,
                         int32_t B, int32_t H, int32_t L, int32_t hd, int32_t er_len, void* stream);
/* Core of torch MultiheadAttention as used at rpr.py:62-63 / the video encoder: q (B,Lq,H*hd)
 * scaled, k,v (B,Lk,H*hd); no mask (causal=0) or causal. */
/* The same attention without the causal mask (forward(mask=False), model/video_music_transformer.py:978-982): every key is
 * visible; the relative term stays zero for keys j > i, which is what model/rpr.py:439-455 (_skew) produces. */
int32_t amt_rpr_attn_nomask_fwd(const float* q, const float* k, const float* v, const float* Er, float* o,
                                int32_t B, int32_t H, int32_t L, int32_t hd, int32_t er_len, void* stream);
int32_t amt_cross_attn_fwd(const float* q, const float* k, const float* v, float* o,
                           int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, int32_t causal, void* stream);
/* General strided form of the same kernel (used by the V2 stack and MultiheadGQA): tensor[b][h][l][c] lives at
 * base + b*bs + h*hs + l*ls + c with strides = {q_bs,q_hs,q_ls, k_bs,k_hs,k_ls, v_bs,v_hs,v_ls, o_bs,o_hs,o_ls} in
 * elements; q_scale multiplies q (torch MHA: hd^-0.5; 0 = 1); query head h reads kv head h / kv_group. */
int32_t amt_attn_fwd(const float* q, const float* k, const float* v, float* o, const int64_t* strides,
                     int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, int32_t causal, int32_t kv_group,
                     float q_scale, void* stream);
/* Input stage pieces of forward (video_music_transformer.py:984-1030): out[row][0:ld_out] = [sem | scene | motion |
 * emotion | 0-pad]; xf[b*L+l] = PR[root] + PA[attr] + key[b]*wkey + bias + pe[l] with PR = E_root.Wc[:, :d]^T etc. */
int32_t amt_concat_features_fwd(const float* sem, int32_t sem_dim, const float* scene, const float* motion, int32_t motion_dim,
                                const float* emotion, int32_t emo_dim, float* out, int32_t rows, int32_t ld_out, void* stream);
int32_t amt_chord_embed_fwd(const int64_t* root, const int64_t* attr, const float* key, const float* PR, const float* PA,
                            const float* wkey, const float* bias, const float* pe, float* out,
                            int32_t B, int32_t L, int32_t d, void* stream);
/* Decode-step forms: q (B,H*hd) for the token at position pos (host value), K/V caches
 * (B,H,cap,hd); keys 0..pos.  Er may be null (cross-attention: pass pos = S-1). */
int32_t amt_attn_decode_fwd(const float* q, const float* kcache, const float* vcache, const float* Er, float* o,
                            int32_t B, int32_t H, int32_t hd, int32_t cap, int32_t pos, int32_t er_len, void* stream);
/* Decode-step projection: y[B,N] = LN?(x)[B,K] . w[N,K]^T + bias (+resid)(relu); packs w on the fly
 * into `w_packed_scratch` (N_pad16*K floats).  ln_w/ln_b may be null.  B <= 32, K % 64 == 0. */
int32_t amt_decode_linear_fwd(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b,
                              const float* resid, float* y, float* xn_out, float* w_packed_scratch,
                              int32_t B, int32_t N, int32_t K, int32_t relu, float eps, void* stream);
/* The two kernels of the decode step in the form the handle uses them, with a LayerNorm folded through the projection
 * (reference rpr.py:59-69 computes LayerNorm(u) then the projection):
 * amt_attn_decode_fold_fwd: raw (B, ldq) = u . (W o gamma)^T, columns [0,d) query (and [d,2d) key, [2d,3d) value of the new
 *   position when new_kv = 1); u (B, d) the pre-LayerNorm sum; q = ((raw - mu*fold_g) * rstd + fold_c) * q_scale with the row
 *   statistics of u; xn_out (optional) = LayerNorm(u).  new_kv = 1 also writes the key/value of position *pos_dev into the
 *   caches and attends keys 0..pos; new_kv = 0 attends keys 0..n_keys-1 (cross-attention).
 * amt_decode_gemm_ex_fwd: rows [x (K1 columns) | x2 (K-K1 columns)]; y_low (B, n_low) = act(x . w_low^T + b (+resid)) over the
 *   first K1 columns (all K when x2 is null), y_high (B, n_high) = [x|x2] . w_high^T + b_high.  pro = 1: the staged row is
 *   [relu((x - mu*fold_g)*rstd + fold_c) | LayerNorm(x2)] (statistics of x2's row) and LayerNorm(x2) is y_low's residual.
 *   Weights are given in nn.Linear layout and packed into the scratch buffers (ceil(n/16)*16*K floats each). */
int32_t amt_attn_decode_fold_fwd(const float* raw, int32_t ldq, float* kcache, float* vcache, const float* Er,
                                 const float* u, const float* fold_g, const float* fold_c, const float* ln_w,
                                 const float* ln_b, float* xn_out, float* o, int32_t B, int32_t H, int32_t hd,
                                 int32_t cap, const int32_t* pos_dev, int32_t n_keys, int32_t er_len, int32_t new_kv,
                                 float eps, float q_scale, void* stream);
typedef struct amt_decode_gemm_args {
    const float* x;  int32_t ldx;          /* first K1 columns of the rows (all K when x2 is null) */
    const float* x2; int32_t ldx2;         /* remaining K - K1 columns, or null */
    int32_t K1, K;
    const float* w_low;  const float* bias_low;  const float* resid; int32_t relu;   /* y_low = act(x . w_low^T + b (+ resid)) */
    const float* w_high; const float* bias_high;                                     /* y_high = [x | x2] . w_high^T + b_high */
    int32_t n_low, n_high;
    int32_t pro;                           /* 1: folded-FFN prologue (see above) */
    const float* fold_g; const float* fold_c; const float* ln_w; const float* ln_b;
    float* y_low; float* y_high;
    float* scratch_low; float* scratch_high;   /* packed copies of the weights, ceil(n/16)*16*K floats each */
    int32_t B; float eps;
} amt_decode_gemm_args;
int32_t amt_decode_gemm_ex_fwd(const amt_decode_gemm_args* a, void* stream);
/* MultiheadGQA.forward (grouped_query_attention.py:286-358) without RoPE: query/key/value are the
 * caller's (L,B,E) buffers, weights in nn.Linear layout; scratch >= 4*L*B*E floats. */
int32_t amt_gqa_fwd(const float* query, const float* key, const float* value,
                    const float* wq, const float* bq, const float* wk, const float* bk, const float* wv, const float* bv,
                    const float* ln_w, const float* ln_b, const float* wo, const float* bo,
                    float* out, float* scratch, int32_t L, int32_t S, int32_t B, int32_t E, int32_t query_heads,
                    int32_t kv_heads, int32_t is_causal, float ln_eps, void* stream);
/* The same with MultiheadGQA(RoPE=...) (grouped_query_attention.py:316-322): the projected q and k are rotated through the raw
 * (heads, len, B, head_dim) view with the module's cos/sin cache (cache_rows, cache_half, 2) -- cache_half = dim/2 of the
 * RotaryPositionalEmbeddings the caller built, head_dim/2 or a multiple that the view folds (rotate_operation.py:148-149). */
int32_t amt_gqa_rope_fwd(const float* query, const float* key, const float* value,
                         const float* wq, const float* bq, const float* wk, const float* bk, const float* wv, const float* bv,
                         const float* ln_w, const float* ln_b, const float* wo, const float* bo,
                         float* out, float* scratch, int32_t L, int32_t S, int32_t B, int32_t E, int32_t query_heads,
                         int32_t kv_heads, int32_t is_causal, float ln_eps, const float* rope_cache, int32_t cache_rows,
                         int32_t cache_half, void* stream);
/* MoELayer.forward / SharedMoELayer.forward, eval mode (moe.py:167-200,231-302): x (n_tok,d);
 * gate (n_exp,d)+(n_exp); experts' linear1/gate (n_exp,dff,d)+(n_exp,dff), linear2 (n_exp,d,dff)+(n_exp,d);
 * shared_* may be null.  top-k = 2.  idx_out/w_out (optional) receive the routing (n_tok,2).
 * scratch: see amt_moe_scratch_floats. */
int64_t amt_moe_scratch_floats(int32_t n_tok, int32_t d, int32_t dff, int32_t n_exp);
int32_t amt_moe_fwd(const float* x, const float* gate_w, const float* gate_b,
                    const float* w1, const float* b1, const float* wg, const float* bg, const float* w2, const float* b2,
                    const float* sw1, const float* sb1, const float* swg, const float* sbg, const float* sw2, const float* sb2,
                    float* out, int32_t* idx_out, float* w_out, float* scratch,
                    int32_t n_tok, int32_t d, int32_t dff, int32_t n_exp, void* stream);
/* The same layer with n_experts_per_token = k, 1 <= k <= 8 (MoELayer / SharedMoELayer constructor argument, moe.py:150-160,202-215; every
 * reference model passes 2): idx_out / w_out are (n_tok, k), the shared expert enters with 1/k; scratch from amt_moe_topk_scratch_floats. */
int64_t amt_moe_topk_scratch_floats(int32_t n_tok, int32_t d, int32_t dff, int32_t n_exp, int32_t k);
int32_t amt_moe_topk_fwd(const float* x, const float* gate_w, const float* gate_b,
                    const float* w1, const float* b1, const float* wg, const float* bg, const float* w2, const float* b2,
                    const float* sw1, const float* sb1, const float* swg, const float* sbg, const float* sw2, const float* sb2,
                    float* out, int32_t* idx_out, float* w_out, float* scratch,
                    int32_t n_tok, int32_t d, int32_t dff, int32_t n_exp, int32_t k, void* stream);

/* Pieces of the same layer for expert-parallel execution (config 5: one expert group per GPU; the token rows
 * travel by all_to_all between amt_moe_route_fwd and amt_moe_combine_fwd, see video2music_amd/model/moe.py):
 * router (moe.py:180-190), one GLUExpert on n rows (moe.py:44-49; scratch >= 2*n*dff floats), and the
 * weighted sum out[t] = w0*y[slot_pos[t,0]] + w1*y[slot_pos[t,1]] in expert-index order (+ shared_scale*shared). */
int32_t amt_moe_route_fwd(const float* x, const float* gate_w, const float* gate_b, int32_t* idx_out, float* w_out,
                          int32_t n_tok, int32_t d, int32_t n_exp, void* stream);
int32_t amt_glu_expert_fwd(const float* x, const float* w1, const float* b1, const float* wg, const float* bg,
                           const float* w2, const float* b2, float* out, float* scratch,
                           int32_t n, int32_t d, int32_t dff, void* stream);
int32_t amt_moe_combine_fwd(const float* y_rows, const int32_t* slot_pos, const int32_t* idx, const float* wts,
                            const float* shared, float shared_scale, float* out, int32_t n_tok, int32_t d, void* stream);
/* Expert-parallel execution with the plans on the DEVICE (video2music_amd/model/moe.py:expert_parallel_moe):
 * amt_moe_ep_dispatch_plan_fwd: from one rank's routing idx[2*n_tok] the rows per expert counts_out[n_exp], the send-buffer order
 *   perm[2*n_tok] (send row -> token; ordered by expert, exact packing, so the rows of one destination rank are contiguous) and
 *   slot_pos[2*n_tok] (assignment token*2+slot -> send row: what amt_moe_combine_fwd takes).  ints: >= 256 int32 of scratch, zero
 *   before the first call.
 * amt_gather_rows_fwd: dst[i] = src[index[i]] (a negative index gives a zero row): the send buffer x[perm].
 * amt_moe_ep_expert_fwd: the rank's e_local experts (stacked tensors as amt_moe_fwd takes them) on the n_recv rows the all_to_all
 *   delivered, grouped by (source rank, local expert) with the segment sizes in DEVICE memory recv_counts[world][e_local]; results
 *   y_out[n_recv][d] in arrival order.  One grouped GEMM launch per projection; scratch: amt_moe_ep_expert_scratch_floats. */
int32_t amt_moe_ep_dispatch_plan_fwd(const int32_t* idx, int32_t n_tok, int32_t n_exp, int32_t* counts_out, int32_t* perm,
                                     int32_t* slot_pos, int32_t* ints, void* stream);
int32_t amt_gather_rows_fwd(const float* src, const int32_t* index, float* dst, int32_t n_rows, int32_t d, void* stream);
int64_t amt_moe_ep_expert_scratch_floats(int32_t n_recv, int32_t d, int32_t dff, int32_t e_local);
int32_t amt_moe_ep_expert_fwd(const float* rows, const int32_t* recv_counts, int32_t world, int32_t e_local, int32_t n_recv,
                              const float* w1, const float* b1, const float* wg, const float* bg, const float* w2, const float* b2,
                              float* y_out, float* scratch, int32_t d, int32_t dff, void* stream);

/* ---- VideoMusicTransformer_V2 '2.2': one KV-cached decode step of one clip ------------------------
 * Issues, from one call, the launch sequence of the decoder restricted to position t (reference
 * video_music_transformer.py:437-516, custom_transformer.py:1250-1292): embedding of (root, attr, key), per layer RoPE
 * self-attention over the cache (appends row t), RoPE cross-attention over the clip's video keys, GLU expert or
 * SharedMoE(top-2), post-norm LayerNorms; decoder.norm; Wout -> logits_out[159].
 * Every projection handles one row, so it runs on the skinny decode GEMM over weights packed once by
 * amt_pack_weight_fwd (out: ceil(N/16)*16*K floats; the experts of a MoE layer are packed one after the other).
 * tab: device pointers, 11 global (PR, PA, wkey, Linear_chord.bias, rope cache (max_seq, E/2, 2) or null = no rotation,
 * decoder.norm w, b, packed Wout, Wout b, an int32 pair {0, 1}, learned positional table (max_seq, E) added to the
 * embedding of position t or null: version '2.0' has the table and no rotation, :375-380,497-503) then 48 per layer (packed self in_proj, its bias, packed out_proj, b,
 * norm1 w, b, packed cross in_proj rows 0:E, its bias, packed out_proj, b, norm2 w, b, norm3 w, b, self K cache, V cache
 * (head-major: H, max_seq, hd), cross K (roped), V (head-major: H, S, hd), router w (null = plain GLU layer), router b, packed linear1, b, packed gate, b,
 * packed linear2, b (per expert, stacked, for a MoE layer), shared expert's six tensors (packed weights) or null; then, for the
 * lockstep step, the stacked forms: packed [gate of every expert (+ the shared one) | linear1 of every expert (+ shared)] as one matrix
 * -- with linear1 present its rows are interleaved in eights BEFORE packing (gate rows 8T..8T+7, then linear1 rows 8T..8T+7, T = 0, 1, ...:
 * the product's epilogue writes linear1 * silu(gate) itself) -- and its bias in the stacked order [gate | linear1], packed linear2 of every expert (+ shared) one after the other and their biases (both null for a plain GLU layer,
 * whose linear2 is the per-layer entry above); last, for the lockstep step with norm1 folded through the cross-attention's query
 * projection (all four null: separate launches): packed [(Wq o gamma1) Wo | Wq o gamma1] (E x 2E), its bias (Wq o gamma1) bo,
 * g = rowsum(Wq o gamma1), c = Wq beta1 + bq  (Wo, bo: self-attention out-projection; Wq, bq: cross in_proj rows 0:E)); and, for
 * a plain GLU layer of the lockstep step with norm2 and norm3 folded (all eight null: separate launches; E and dff multiples of
 * 256, dff + E <= 1536): packed [(Wgu o gamma2) Wo | Wgu o gamma2] (2 dff x 2E; Wgu = [gate; linear1], Wo the cross-attention's
 * out-projection), its bias (Wgu o gamma2) bo, g = rowsum(Wgu o gamma2), c = Wgu beta2 + bgu, then -- null in the last layer --
 * packed [(Wn o gamma3) W2 | Wn o gamma3] (3E x (dff + E); Wn the NEXT layer's self in_proj, W2 this layer's linear2), its bias
 * (Wn o gamma3) b2, g = rowsum(Wn o gamma3), c = Wn beta3 + bn.
 * A null norm bias selects RMSNorm (eps 1e-6) for that norm; a null linear1 selects Linear -> SiLU -> Linear experts
 * (h = silu(gate-slot projection)): the V1 family (video_music_transformer.py:22-314).
 * ws: amt_v2_step_ws_floats(E, dff, n_exp) floats.  E, dff multiples of 64, at most 1536.
 * state_dev (optional): int32 {position, root, attr} in device memory; when given, t / root / attr are read there and the
 * position is incremented at the end of the step, so that one captured graph of the step can be replayed for every token. */
int32_t amt_pack_weight_fwd(const float* w, float* out, int32_t N, int32_t K, void* stream);
int64_t amt_v2_step_ws_floats(int32_t E, int32_t dff, int32_t n_exp);
int32_t amt_v2_step(const void* const* tab, int32_t n_layers, int32_t H, int32_t E, int32_t dff, int32_t n_exp,
                    int32_t S, int32_t max_seq, int32_t t, int32_t root, int32_t attr, float key, const int32_t* state_dev,
                    float* logits_out, float* ws, void* stream);
/* ---- regression head, recurrent variants (video_regression.py:124-135: torch.nn.LSTM / nn.GRU, batch_first) -------------
 * The recurrence of one layer and direction over projected inputs xproj (B, L, ldxp) whose columns [0, gates*d) hold
 * W_ih x + b_ih (gate order of torch: LSTM i,f,g,o; GRU r,z,n): gates = 4 LSTM, 3 GRU; w_hh (gates*d, d), b_hh (gates*d);
 * y (B, L, ldy) receives h_t in columns [0, d) of the pointer given (pass y + d for the reverse direction of a
 * bidirectional layer); reverse = 1 walks t = L-1 .. 0 and writes h at t.  d a multiple of 8, at most 128.
 * n_dirs = 2 runs both directions of a bidirectional layer in one launch: xproj then holds [forward | reverse] projections
 * side by side (2*gates*d columns), w_hh / b_hh the two directions stacked, y gets [h_forward | h_reverse] (2*d columns). */
int32_t amt_rnn_seq_fwd(const float* xproj, int32_t ldxp, const float* w_hh, const float* b_hh, float* y, int32_t ldy,
                        int32_t B, int32_t L, int32_t d, int32_t gates, int32_t reverse, int32_t n_dirs, void* stream);

/* The same step for B independent clips in lockstep (all at one position): projections are one launch over B rows (weights
 * read once per step, not once per clip), the caches carry a leading clip dimension (self K/V: B, H, max_seq, hd; cross K/V:
 * B, H, S, hd), a mixture layer evaluates all experts on all rows and combines each row's routed pair in expert-index
 * order.  Same pointer table.  keys_dev: B floats; state_dev: int32 {position, root[B], attr[B]} in device memory, the
 * position is incremented at the end; logits_out (B, 159); ws: amt_v2_step_batch_ws_floats(E, dff, n_exp, B) floats. */
int64_t amt_v2_step_batch_ws_floats(int32_t E, int32_t dff, int32_t n_exp, int32_t B);
int32_t amt_v2_step_batch(const void* const* tab, int32_t n_layers, int32_t H, int32_t E, int32_t dff, int32_t n_exp,
                          int32_t S, int32_t max_seq, int32_t B, const float* keys_dev, int32_t* state_dev,
                          float* logits_out, float* ws, void* stream);

/* Per-clip decision of the lockstep step, on the device (reference model/video_music_transformer.py:547-600: temperature
 * softmax[:157], N / repeat suppression, top-1 / arg-max / Categorical draw by inverse CDF at uniforms[(pos)*B + b], id ->
 * (root, attr) feedback; chord_embed: the id feeds back).  Launch right after amt_v2_step_batch on the same stream (it
 * reads the position from state_dev[0], which that call has advanced): stores tokens[b][pos] (and roots / attrs, all
 * [B][T] int64, primer positions pre-filled) and leaves the position's (root, attr) in state_dev for the next step.
 * No host round trip: the pair (step, decide) is captured once and replayed T-1 times. */
int32_t amt_v2_decide_batch(const float* logits, int32_t ld_logits, int32_t* state_dev, int64_t* tokens, int64_t* roots,
                            int64_t* attrs, int32_t B, int32_t T, int32_t n_primer, int32_t beam, int32_t max_conseq_N,
                            int32_t max_conseq_chord, float temperature, const float* uniforms, int32_t chord_embed, void* stream);
/* amt_v2_step_batch and the decision as ONE launch chain (what the lockstep generate replays T-2 times from a captured graph):
 * the step for the position in state_dev[0] WITHOUT its trailing position increment, then one kernel that decides position
 * state_dev[0] + 1 exactly like amt_v2_decide_batch, writes that position's chord-stream row into ws (the input of the next call,
 * which therefore starts at the first projection: pass first = 1 only for the first call of a generation, whose input row is
 * computed from state_dev) and advances state_dev[0].  state_dev: 2 + 2B int32 {position, root[B], attr[B], ticket = 0}.
 * Inside the chain: a mixture layer's routing happens in its combine kernel, the last layer's norm3 and decoder.norm both in the
 * prologue of the output head. */
typedef struct amt_v2_step_args {          /* the lockstep step: see amt_v2_step_batch */
    const void* const* tab;
    int32_t n_layers, H, E, dff, n_exp, S, max_seq, B;
    const float* keys_dev; int32_t* state_dev; float* logits_out; float* ws;
} amt_v2_step_args;
typedef struct amt_v2_decide_args {        /* the decision: see amt_v2_decide_batch */
    int64_t* tokens; int64_t* roots; int64_t* attrs;     /* [B][T] */
    int32_t T, n_primer, beam, max_conseq_N, max_conseq_chord;
    float temperature;
    const float* uniforms;                 /* (T, B) or null */
    int32_t chord_embed;
} amt_v2_decide_args;
/* The caller issues at most T - 1 - n_done calls per generation (positions 0 .. T-2); the position counter in state_dev never
 * advances past T - 1, so a call too many recomputes the last position instead of leaving the caches (T <= max_seq required). */
int32_t amt_v2_step_decide_batch(const amt_v2_step_args* step, const amt_v2_decide_args* decide, int32_t first, void* stream);
/* Introspection for bench.py: kernel launches issued by the calling thread's last amt_v2_step_batch / amt_v2_step_decide_batch
 * call (a count, not a status). */
int32_t amt_v2_last_step_launches(void);

/* ---- regression head VideoRegression(regModel='bimamba+') (model/video_regression.py:104-245) ---- */
/* Depthwise causal Conv1d(kernel K, padding K-1)[..., :L] + SiLU of MambaBlock.forward (mamba.py:172-175,268-272):
 * x (B,L,C) with row stride ldx, w (C,K) = conv1d.weight (C,1,K), y (B,L,C).  reverse = 1 evaluates the block of the
 * time-flipped sequence and returns it un-flipped (the backward branch of bimamba.py:171-185 without the two flips). */
int32_t amt_dwconv1d_silu_fwd(const float* x, int32_t ldx, const float* w, const float* bias, float* y,
                              int32_t B, int32_t L, int32_t C, int32_t K, int32_t reverse, void* stream);
/* MambaBlock.ssm + gate (mamba.py:291-354, 281-287): delta = softplus(delta_raw + dt_bias), A = -exp(A_log) (ED,N),
 * h_t = exp(delta A) h_{t-1} + delta B_t x_t, y_t = h_t . C_t + D x_t, out = y*silu(z) [+ x*(1 - sigmoid(silu(z))) for
 * version 1 = "Mamba+"].  x / delta_raw / z / y are (B*L, ED) with their own row strides, Bm / Cm (B*L, N) with stride
 * ld_bc (slices of the x_proj output).  N = 16.  reverse as above. */
int32_t amt_selective_scan_fwd(const float* x, int32_t ldx, const float* delta_raw, int32_t ld_delta, const float* dt_bias,
                               const float* A_log, const float* Bm, const float* Cm, int32_t ld_bc, const float* D,
                               const float* z, int32_t ldz, float* y, int32_t ldy, int32_t B, int32_t L, int32_t ED,
                               int32_t N, int32_t version, int32_t reverse, void* stream);
/* out[row] = [a[row][0:da] | b[row][0:db] | 0 ...] (ld_out columns): cat(semantic, emotion) of get_feature
 * (video_regression.py:199-216), zero-padded to the GEMM's K step. */
int32_t amt_concat2_fwd(const float* a, int32_t da, const float* b, int32_t db, float* out, int32_t rows, int32_t ld_out,
                        void* stream);
/* amt_linear_fwd with leading dimensions and an activation: y = act(x[M,K](ldx) . w[N,K](ldw)^T + bias (+ resid(ldr)));
 * act 0 none, 1 ReLU, 2 sigmoid (the classifier head, video_regression.py:188-197), 3 SiLU (CNN_GRU's convolution, :88-91).
 * K % 32 == 0. */
int32_t amt_linear_ex_fwd(const float* x, int32_t ldx, const float* w, int32_t ldw, const float* bias, const float* resid,
                          int32_t ldr, float* y, int32_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, void* stream);
/* y = LayerNorm(x (+ resid)) + post : the "norm2(x_b + x) then x_f + x_b" step of bimamba.py:183-188 in one pass. */
int32_t amt_layernorm_post_fwd(const float* x, const float* resid, const float* w, const float* b, const float* post,
                               float* y, int32_t rows, int32_t dim, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMT_HIP_H */
