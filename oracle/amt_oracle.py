"""CPU oracle for the AMT forward / generate hot path.  TEST INFRASTRUCTURE ONLY.

A restatement, in primitive torch-CPU tensor ops, of the arithmetic the reference
``VideoMusicTransformer`` executes (reference files cited per function, paths relative to the
reference tree).  It exists to check the HIP path and to serve as the timed ``cpu_baseline`` of
``bench.py``; nothing in ``video2music_amd/`` may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it.

Parity status: PINNED.  The reference publishes no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, imported on CPU in the build container
by ``oracle/make_goldens.py`` (fixtures in ``tests/golden/*.npz``, checked by
``tests/test_oracle_golden.py``).

Weights are passed as a ``state_dict``-shaped ``{name: tensor}`` with the reference's key names
(``model/video_music_transformer.py:911-976``).  All tensors are batch-first here; the reference's
seq-first ``(L,B,d)`` layout is a memory-layout choice that does not change the numbers.
"""
import math

import torch
import torch.nn.functional as F

CHORD_END = 157
CHORD_PAD = 158
CHORD_SIZE = 159
CHORD_ROOT_PAD = 14
CHORD_ATTR_PAD = 15
LN_EPS = 1e-5


# ----------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------
def positional_encoding(max_len, d_model, dtype=torch.float32):
    """model/positional_encoding.py:13-17: pe[p,2i]=sin(p*w_i), pe[p,2i+1]=cos(p*w_i)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


def layer_norm(x, w, b, eps=LN_EPS):
    """torch.nn.LayerNorm (biased variance, eps inside the sqrt), used at rpr.py:48-50 and by
    torch's TransformerEncoderLayer."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def linear(x, w, b=None):
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def split_heads(x, H):
    """(B,L,d) -> (B,H,L,hd); head h = contiguous hd-slice of the feature dim (rpr.py:351-355)."""
    B, L, d = x.shape
    return x.view(B, L, H, d // H).permute(0, 2, 1, 3)


def merge_heads(x):
    B, H, L, hd = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, H * hd)


def skew(qe):
    """model/rpr.py:439-455 verbatim: zero with the flipped-triu mask, left-pad one column,
    reinterpret (L, L+1) memory as (L+1, L), drop row 0."""
    sz = qe.shape[-2]
    mask = (torch.triu(torch.ones(sz, sz, dtype=qe.dtype)) == 1).to(qe.dtype).flip(0)
    qe = mask * qe
    qe = F.pad(qe, (1, 0))
    qe = qe.reshape(*qe.shape[:-2], qe.shape[-1], qe.shape[-2])
    return qe[..., 1:, :]


def rpr_bias_closed_form(q, Er):
    """Appendix A1 of SURVEY.md: bias[i,j] = q_i . Er[er_len-1-(i-j)] for j<=i, 0 above the
    diagonal.  Used by the tests to cross-check ``skew`` (the production-shaped formula the HIP
    kernels implement)."""
    L = q.shape[-2]
    er_len = Er.shape[0]
    i = torch.arange(L).view(L, 1)
    j = torch.arange(L).view(1, L)
    idx = (er_len - 1 - (i - j)).clamp(0, er_len - 1)
    full = torch.einsum("...ld,lmd->...lm", q, Er[idx])
    return torch.where(j <= i, full, torch.zeros((), dtype=q.dtype))


# ----------------------------------------------------------------------------------------------
# attention blocks
# ----------------------------------------------------------------------------------------------
def rpr_self_attention(x, sd, prefix, H, causal=True):
    """model/rpr.py:201-424 (multi_head_attention_forward_rpr) for q=k=v=x.

    packed in-proj (:253), q *= hd^-0.5 (:328), QK^T (:387), Er[-L:] / einsum / _skew (:391-394),
    additive causal mask (:396-398), softmax (:409), PV (:414), out-proj (:417).
    """
    B, L, d = x.shape
    hd = d // H
    qkv = linear(x, sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q * (float(hd) ** -0.5)
    q, k, v = split_heads(q, H), split_heads(k, H), split_heads(v, H)
    s = q @ k.transpose(-1, -2)
    Er = sd.get(prefix + "Er")          # rpr=False (video_music_transformer.py:903-911): torch's stock decoder layer, no table
    if Er is not None:
        er = Er[max(0, Er.shape[0] - L):, :]                   # _get_valid_embedding, rpr.py:426-437
        qe = torch.einsum("bhld,md->bhlm", q, er)
        s = s + skew(qe)
    if causal:
        mask = torch.triu(torch.full((L, L), float("-inf"), dtype=x.dtype), diagonal=1)
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = merge_heads(p @ v)
    return linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def mha(xq, xkv, sd, prefix, H):
    """torch.nn.MultiheadAttention forward (packed in_proj rows 0:d -> q, d:2d -> k, 2d:3d -> v;
    softmax((q*hd^-0.5) k^T) v; out-proj), as called at rpr.py:62-63 (cross-attention) and inside
    torch's TransformerEncoderLayer (video encoder)."""
    d = xq.shape[-1]
    hd = d // H
    W, b = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    q = linear(xq, W[:d], b[:d]) * (float(hd) ** -0.5)
    k = linear(xkv, W[d:2 * d], b[d:2 * d])
    v = linear(xkv, W[2 * d:], b[2 * d:])
    q, k, v = split_heads(q, H), split_heads(k, H), split_heads(v, H)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o = merge_heads(p @ v)
    return linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


# ----------------------------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------------------------
def n_layers_of(sd, stack):
    n = 0
    while f"transformer.{stack}.layers.{n}.norm1.weight" in sd:
        n += 1
    return n


def video_stream(sd, sem, scene_off, motion, emotion):
    """model/video_music_transformer.py:1005-1030: concat features, Linear_vis, + pe_v."""
    vf = sem.to(sd["Linear_vis.weight"].dtype)
    scene_embed = "scene_embedding.weight" in sd          # scene_embed=True (:926-928): the offset indexes an embedding instead (:1016-1027)
    if not scene_embed:
        vf = torch.cat([vf, scene_off.unsqueeze(-1).to(vf.dtype)], dim=-1)
    if motion.dim() == 2:
        vf = torch.cat([vf, motion.unsqueeze(-1).to(vf.dtype)], dim=-1)
    else:
        vf = torch.cat([vf, motion.to(vf.dtype)], dim=-1)
    vf = torch.cat([vf, emotion.to(vf.dtype)], dim=-1)
    vf = linear(vf, sd["Linear_vis.weight"], sd["Linear_vis.bias"])
    if scene_embed:
        vf = vf + sd["scene_embedding.weight"][scene_off.to(torch.int32).long()]
    S, d = vf.shape[1], vf.shape[2]
    return vf + positional_encoding(S, d, vf.dtype)


def encode(sd, H, sem, scene_off, motion, emotion):
    """Video encoder = torch nn.TransformerEncoder(6 x post-norm ReLU layer) + final LayerNorm
    (constructed at model/video_music_transformer.py:967-971)."""
    x = video_stream(sd, sem, scene_off, motion, emotion)
    for i in range(n_layers_of(sd, "encoder")):
        p = f"transformer.encoder.layers.{i}."
        x = layer_norm(x + mha(x, x, sd, p + "self_attn.", H), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        ff = linear(torch.relu(linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                    sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = layer_norm(x + ff, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    return layer_norm(x, sd["transformer.encoder.norm.weight"], sd["transformer.encoder.norm.bias"])


def chord_stream(sd, x_root, x_attr, key):
    """model/video_music_transformer.py:984-1001,1027-1029: E_root+E_attr, append key, Linear_chord, + pe."""
    if "chord_embedding_model.weight" in sd:             # chord_embed=True (:931-937, 986-987): `x_root` carries the chord ids
        x = sd["chord_embedding_model.weight"][x_root]
    else:
        x = sd["embedding_root.weight"][x_root] + sd["embedding_attr.weight"][x_attr]
    B, L, d = x.shape
    key = key.to(x.dtype).reshape(-1)
    if key.numel() == 1:
        key = key.expand(B)
    kcol = key.view(B, 1, 1).expand(B, L, 1)
    x = linear(torch.cat([x, kcol], dim=-1), sd["Linear_chord.weight"], sd["Linear_chord.bias"])
    return x + positional_encoding(L, d, x.dtype)


def decode(sd, H, x, memory, collect=None, causal=True):
    """model/rpr.py:24-35,55-70: 6 x post-norm (RPR self-attn, cross-attn, ReLU FFN) + final LN."""
    for i in range(n_layers_of(sd, "decoder")):
        p = f"transformer.decoder.layers.{i}."
        x = layer_norm(x + rpr_self_attention(x, sd, p + "self_attn.", H, causal), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        x = layer_norm(x + mha(x, memory, sd, p + "multihead_attn.", H), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        ff = linear(torch.relu(linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                    sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = layer_norm(x + ff, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
        if collect is not None:
            collect.append(x)
    return layer_norm(x, sd["transformer.decoder.norm.weight"], sd["transformer.decoder.norm.bias"])


def forward(sd, H, x_root, x_attr, sem, key, scene_off, motion, emotion, collect=None, mask=True, separated=False):
    """VideoMusicTransformer.forward (model/video_music_transformer.py:978-1044): logits (B,L,159); ``separated`` = IS_SEPERATED
    (:968-973, 1036-1040): the pair (y_root, y_attr) from the Wout_root / Wout_attr heads.  mask=False drops the causal mask (:978-982; the
    relative term stays zero above the diagonal, that is `_skew`'s doing)."""
    memory = encode(sd, H, sem, scene_off, motion, emotion)
    xf = chord_stream(sd, x_root, x_attr, key)
    out = decode(sd, H, xf, memory, collect, causal=mask is True)
    if separated:
        return linear(out, sd["Wout_root.weight"], sd["Wout_root.bias"]), linear(out, sd["Wout_attr.weight"], sd["Wout_attr.bias"])
    return linear(out, sd["Wout.weight"], sd["Wout.bias"])


def root_attr_of(tok):
    """model/video_music_transformer.py:1107-1123 via the chord tables: plain root -> attr 1."""
    if tok == 0:
        return 0, 1
    return (tok - 1) // 13 + 1, (tok - 1) % 13 + 1


def generate(sd, H, sem, key, scene_off, motion, emotion, primer, primer_root, primer_attr,
             target_seq_length=300, beam=0, max_conseq_N=0, max_conseq_chord=2, margins=None, forward_fn=None,
             temperature=1.0, beam_chance=1.0, rng=None):
    """VideoMusicTransformer.generate (model/video_music_transformer.py:1046-1132), one clip,
    full re-forward every step exactly like the reference (no KV cache, encoder re-run).

    beam=1 -> G1 (verbatim top-1 branch :1078-1084: root/attr never updated, no suppression).
    beam=0 -> G2: the sampling branch :1085-1128 with ``Categorical.sample`` replaced by
    arg-max of the normalised probabilities.
    beam>1 / beam_chance<1 (:1074-1084 as written): ``rng.uniform(0, 1) <= beam_chance`` (python's ``random`` in the reference)
    picks the branch per step; the top-k branch replicates row 0 ``beam`` times and writes the k best ids into column cur.
    Returns int64 (max(beam,1), T).  ``margins`` (list) receives top1-top2 of the decision distribution.
    """
    import random as _random
    rng = rng or _random
    T = target_seq_length
    gen = torch.full((1, T), CHORD_PAD, dtype=torch.long)
    gen_root = torch.full((1, T), CHORD_ROOT_PAD, dtype=torch.long)
    gen_attr = torch.full((1, T), CHORD_ATTR_PAD, dtype=torch.long)
    P = len(primer)
    gen[0, :P] = primer
    gen_root[0, :P] = primer_root
    gen_attr[0, :P] = primer_attr
    ids_feed = "chord_embedding_model.weight" in sd       # chord_embed: forward consumes gen_seq itself, in both branches
    if ids_feed:
        gen_root[0, :P] = primer
    cur = P
    while cur < T:
        logits = (forward_fn or forward)(sd, H, gen_root[:, :cur], gen_attr[:, :cur], sem, key, scene_off, motion, emotion)
        y = torch.softmax(logits / temperature, dim=-1)[..., :CHORD_END]
        probs = y[:, cur - 1, :].clone()
        beam_ran = 2.0 if beam == 0 else rng.uniform(0, 1)
        if beam_ran <= beam_chance:
            top_i = torch.topk(probs.flatten(), beam)[1]
            gen = gen[top_i // CHORD_SIZE, :]
            gen[..., cur] = top_i % CHORD_SIZE
            if ids_feed:
                assert beam == 1, "chord_embed with beam > 1 feeds `beam` rows against one clip of features: fails in the reference"
                gen_root[0, cur] = gen[0, cur]
        else:
            if max_conseq_N == 0:
                probs[0, 0] = 0.0
            is_max = cur >= max_conseq_chord
            if is_max:
                pre = int(gen[0, cur - 1])
                for k in range(1, max_conseq_chord):
                    if pre != int(gen[0, cur - 1 - k]):
                        is_max = False
            if is_max:
                probs[0, int(gen[0, cur - 1])] = 0.0
            pn = probs / probs.sum(-1, keepdim=True)        # Categorical(probs=...) normalises
            tok = int(pn.argmax(-1))
            probs = pn
            r, a = (tok, 0) if ids_feed else root_attr_of(tok)
            gen[:, cur] = tok
            gen_root[0, cur] = r
            gen_attr[0, cur] = a
        if margins is not None:
            top2 = torch.topk(probs.flatten(), 2)[0]
            margins.append(float(top2[0] - top2[1]))
        cur += 1
    return gen[:, :cur]


# ----------------------------------------------------------------------------------------------
# VideoMusicTransformer_V2, version '2.2' (model/video_music_transformer.py:316-609)
# ----------------------------------------------------------------------------------------------
def v2_attention(xq, xkv, sd, prefix, H, cache, causal):
    """CustomMultiheadAttention with RoPE (model/custom_transformer.py:864-1218): packed in-proj, RoPE on q and k
    through the raw (H, L, B, hd) view of the seq-first (L, B, E) buffers (:1041-1053), q * hd^-0.5, softmax, out-proj.
    Inputs/outputs are seq-first (L, B, E) because the view above depends on that memory order."""
    L, B, E = xq.shape
    S = xkv.shape[0]
    hd = E // H
    W, b = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    q = linear(xq, W[:E], b[:E]).contiguous()
    k = linear(xkv, W[E:2 * E], b[E:2 * E]).contiguous()
    v = linear(xkv, W[2 * E:], b[2 * E:]).contiguous()
    q = rope(q.view(H, L, B, hd), cache).reshape(L, B, E)
    k = rope(k.view(H, S, B, hd), cache).reshape(S, B, E)
    qh = q.reshape(L, B * H, hd).transpose(0, 1) * math.sqrt(1.0 / float(hd))
    kh = k.reshape(S, B * H, hd).transpose(0, 1)
    vh = v.reshape(S, B * H, hd).transpose(0, 1)
    s = qh @ kh.transpose(-1, -2)
    if causal:
        s = s + torch.triu(torch.full((L, S), float("-inf"), dtype=s.dtype), diagonal=1)
    o = torch.softmax(s, dim=-1) @ vh                                  # (B*H, L, hd)
    o = o.transpose(0, 1).contiguous().view(L * B, E)
    return linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"]).view(L, B, E)


def v2_ff(x, sd, prefix):
    """GLUExpert for the three shallow layers, SharedMoELayer(6 experts, top-2) for the deep ones
    (model/video_music_transformer.py:384-416)."""
    if prefix + "linear1.weight" in sd:
        return glu_expert(x, sd, prefix)
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    n_exp = sub["gate.weight"].shape[0]
    return moe_forward(x, sub, n_exp, k=2, shared=True)


def forward_v2(sd, H, x_root, x_attr, sem, key, scene_off, motion, emotion, max_seq_video=300, mask=True, drop_keep=None):
    """VideoMusicTransformer_V2.forward, version '2.2', chord_embed=False (:427-516): no additive positional
    encoding, RoPE inside every attention, post-norm layers (custom_transformer.py:1228-1240, 1260-1276).  ``mask=False``:
    tgt_mask=None (:440-443).  ``drop_keep`` (B, S) in {0, 1}: the dropTokenRate mask ``torch.rand(B, S) > rate`` (:488-492)."""
    d = sd["Wout.weight"].shape[1]
    cache = rope_cache(d, max_seq_video).to(sd["Wout.weight"].dtype)
    x = sd["embedding_root.weight"][x_root] + sd["embedding_attr.weight"][x_attr]
    B, L, _ = x.shape
    kk = key.to(x.dtype).reshape(-1)
    if kk.numel() == 1:
        kk = kk.expand(B)
    x = linear(torch.cat([x, kk.view(B, 1, 1).expand(B, L, 1)], dim=-1), sd["Linear_chord.weight"], sd["Linear_chord.bias"])
    vf = sem.to(x.dtype)
    vf = torch.cat([vf, scene_off.unsqueeze(-1).to(x.dtype)], dim=-1)
    vf = torch.cat([vf, motion.unsqueeze(-1).to(x.dtype) if motion.dim() == 2 else motion.to(x.dtype)], dim=-1)
    vf = torch.cat([vf, emotion.to(x.dtype)], dim=-1)
    vf = linear(vf, sd["Linear_vis.weight"], sd["Linear_vis.bias"])
    if drop_keep is not None:
        vf = vf * drop_keep.to(vf.dtype).unsqueeze(-1)
    xf, src = x.permute(1, 0, 2).contiguous(), vf.permute(1, 0, 2).contiguous()     # seq-first
    for i in range(n_layers_of(sd, "encoder")):
        p = f"transformer.encoder.layers.{i}."
        src = layer_norm(src + v2_attention(src, src, sd, p + "self_attn.", H, cache, False), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        src = layer_norm(src + v2_ff(src, sd, p + "ff."), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    memory = layer_norm(src, sd["transformer.encoder.norm.weight"], sd["transformer.encoder.norm.bias"])
    t = xf
    for i in range(n_layers_of(sd, "decoder")):
        p = f"transformer.decoder.layers.{i}."
        t = layer_norm(t + v2_attention(t, t, sd, p + "self_attn.", H, cache, mask is True), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        t = layer_norm(t + v2_attention(t, memory, sd, p + "cross_attn.", H, cache, False), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        t = layer_norm(t + v2_ff(t, sd, p + "ff."), sd[p + "norm3.weight"], sd[p + "norm3.bias"])
    t = layer_norm(t, sd["transformer.decoder.norm.weight"], sd["transformer.decoder.norm.bias"])
    return linear(t.permute(1, 0, 2), sd["Wout.weight"], sd["Wout.bias"])


# ----------------------------------------------------------------------------------------------
# standalone modules of configs 4/5 and the kernel-level rows (a10-a12)
# ----------------------------------------------------------------------------------------------
def rms_norm(x, w, eps=1e-6):
    """model/custom_transformer.py:38-45: x * rsqrt(mean(x^2)+eps) * w, computed in fp32."""
    xf = x.float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x)
    return y * w if w is not None else y


def rope_cache(dim, max_seq_len, base=10000):
    """model/rotate_operation.py:88-109: theta_i = base^(-2i/dim); cache (max_seq, dim/2, [cos,sin])."""
    theta = 1.0 / (base ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    idx = torch.arange(max_seq_len, dtype=theta.dtype)
    ang = torch.einsum("i,j->ij", idx, theta).float()
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1)


def rope(x, cache, input_pos=None):
    """model/rotate_operation.py:111-165: x (b, s, n_h, h_d); rotate interleaved pairs (2i,2i+1)
    by the cached angle of position s."""
    seq_len = x.size(1)
    rc = cache[:seq_len] if input_pos is None else cache[input_pos]
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    # when the cache was built for dim != h_d (the reference builds it with dim=d_model,
    # video_music_transformer.py:87,380,660) this view folds the extra frequencies into the
    # leading axis, which is then truncated to x's leading size (rotate_operation.py:148-149)
    rc = rc.contiguous().view(-1, xs.size(1), 1, xs.size(3), 2)
    rc = rc[: xs.size(0), ...]
    out = torch.stack([xs[..., 0] * rc[..., 0] - xs[..., 1] * rc[..., 1],
                       xs[..., 1] * rc[..., 0] + xs[..., 0] * rc[..., 1]], -1)
    return out.flatten(3).type_as(x)


def gqa_forward(query, key, value, sd, query_heads, kv_heads, is_causal=False, layer_norm_eps=1e-5, rope_cache_=None):
    """model/grouped_query_attention.py:286-358 (MultiheadGQA.forward) + :19-170, Appendix A5.

    Inputs are seq-first ``(L,B,E)`` like the reference's callers pass; the reference then
    *reinterprets the memory* as ``(B,L,E)`` with ``.view`` (:316-326), which this restatement
    reproduces with ``reshape`` on the contiguous buffers.  Output ``(L,B,E)``-shaped buffer.
    """
    L, B, E = query.shape
    q = linear(query, sd["q_proj.weight"], sd["q_proj.bias"])
    k = linear(key, sd["k_proj.weight"], sd["k_proj.bias"])
    v = linear(value, sd["v_proj.weight"], sd["v_proj.bias"])
    hd = E // query_heads
    g = query_heads // kv_heads
    src_len = k.shape[0]
    if rope_cache_ is not None:          # RoPE=... (:316-322): rotation through the raw (heads, len, B, hd) view
        q = rope(q.contiguous().view(query_heads, L, B, hd), rope_cache_)
        k = rope(k.contiguous().view(kv_heads, src_len, B, hd), rope_cache_)
    # raw memory reinterpretation (L,B,*) -> (B,L,heads,hd)
    q = q.contiguous().view(B, L, query_heads, hd)
    k = k.contiguous().view(B, src_len, kv_heads, hd)
    v = v.contiguous().view(B, src_len, kv_heads, hd)
    q = q.permute(0, 2, 1, 3) / (hd ** 0.5)                    # b (h g) n d, scale :121-123
    k = k.permute(0, 2, 1, 3)
    v = v.permute(0, 2, 1, 3)
    S = k.shape[2]
    qg = q.reshape(B, kv_heads, g, L, hd)                      # "b (h g) n d -> b g h n d": head = h*g+gi
    sim = torch.einsum("bhgnd,bhsd->bhgns", qg, k)
    if is_causal:
        keep = torch.ones(L, S, dtype=torch.bool).tril_()
        sim = sim.masked_fill(~keep, torch.finfo(sim.dtype).min)
    attn = torch.softmax(sim, dim=-1)
    out = torch.einsum("bhgns,bhsd->bhgnd", attn, v)           # (B,h,g,L,hd)
    out = out.permute(3, 0, 1, 2, 4).reshape(L, B, E)          # "b g h n d -> n b (h g) d" (:159)
    out = layer_norm(out, sd["norm.weight"], sd["norm.bias"], layer_norm_eps)  # MAGNETO LN (:352-354)
    return linear(out, sd["out_proj.weight"], sd["out_proj.bias"])


def glu_expert(x, sd, p):
    """model/moe.py:36-49 GLUExpert: W2((W1 x + b1) * silu(Wg x + bg)) + b2 (dropout = id in eval)."""
    a = linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])
    gte = F.silu(linear(x, sd[p + "gate.weight"], sd[p + "gate.bias"]))
    return linear(a * gte, sd[p + "linear2.weight"], sd[p + "linear2.bias"])


def moe_forward(x, sd, n_experts, k=2, shared=False, routing=None, temperature=1.0):
    """model/moe.py:167-200 (MoELayer) / :231-302 (SharedMoELayer), eval mode, Appendix A6.

    gate logits -> top-k -> softmax over the k logits (fp32) -> sum_e w_e * expert_e(x), accumulated
    in expert-index order; shared: + shared_expert(x)/k.  ``routing`` (dict) receives idx / weights.  ``temperature``: the value
    SharedMoELayer's scheduler holds after its step of this forward (:238-240), dividing the k logits (:288).
    """
    logits = linear(x, sd["gate.weight"], sd.get("gate.bias"))
    w, idx = torch.topk(logits, k, dim=-1)
    w = torch.softmax(w.float() / temperature, dim=-1).to(x.dtype)
    out = torch.zeros_like(x)
    for e in range(n_experts):
        sel = idx == e                                          # (..., k)
        tok = sel.any(-1)
        if not tok.any():
            continue
        we = (w * sel).sum(-1)[tok]
        out[tok] += we.unsqueeze(-1) * glu_expert(x[tok], sd, f"experts.{e}.")
    if shared:
        out = out + (1.0 / k) * glu_expert(x, sd, "shared_expert.")
    if routing is not None:
        routing["idx"], routing["weights"] = idx, w
    return out
