"""``VideoMusicTransformer_V3`` (reference ``model/video_music_transformer.py:611-909``): RMSNorm, differential attention, no KV cache.
Importable from ``video2music_amd.model.video_music_transformer`` like in the reference; split out for size."""
import math

import torch
import torch.nn as nn

from ..utilities.constants import (CHORD_ATTR_PAD, CHORD_ATTR_SIZE, CHORD_END, CHORD_PAD, CHORD_ROOT_PAD, CHORD_ROOT_SIZE, CHORD_SIZE,
                                   SCENE_OFFSET_MAX)
from .vmt_v2 import VideoMusicTransformer_V2, _TransformerParamsV2


class _DiffAttnParams(nn.Module):
    """Keys of custom_transformer.DifferentialMultiheadAttention (:610-647): bias-free q/k (E -> 2E), v, out projections,
    the four lambda vectors and the sub-layer RMSNorm over head_dim."""

    def __init__(self, d_model, head_dim, depth):
        super().__init__()
        for n in ("lambda_q1", "lambda_k1", "lambda_q2", "lambda_k2"):
            setattr(self, n, nn.Parameter(torch.zeros(head_dim).normal_(mean=0, std=0.1)))
        self.k_proj = nn.Linear(d_model, 2 * d_model, bias=False)
        self.q_proj = nn.Linear(d_model, 2 * d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.out_proj = nn.Linear(d_model, d_model, bias=False)
        from .custom_transformer import RMSNorm
        self.subln = RMSNorm(head_dim, eps=1e-5, elementwise_affine=True)
        self.lambda_init = 0.8 - 0.6 * math.exp(-0.3 * depth)                   # lambda_init_fn (:607-608)
        for q in (self.k_proj, self.q_proj, self.v_proj, self.out_proj):
            nn.init.xavier_uniform_(q.weight)

    def lambda_full(self):
        """exp(lq1.lk1) - exp(lq2.lk2) + lambda_init (:818-820); a host scalar, recomputed when a lambda vector changes."""
        ps = (self.lambda_q1, self.lambda_k1, self.lambda_q2, self.lambda_k2)
        sig = tuple((q.data_ptr(), q._version) for q in ps)
        if getattr(self, "_lam_sig", None) != sig:
            q1, k1, q2, k2 = (q.detach().float().cpu() for q in ps)
            self._lam = float(torch.exp(torch.sum(q1 * k1)) - torch.exp(torch.sum(q2 * k2)) + self.lambda_init)
            self._lam_sig = sig
        return self._lam


class VideoMusicTransformer_V3(VideoMusicTransformer_V2):
    """Reference ``VideoMusicTransformer_V3`` (model/video_music_transformer.py:611-909), versions '3.0', '3.1', '3.2', eval
    mode.  RMSNorm everywhere, RoPE built for dim = 2 d_model, three GLU layers then SharedMoELayer(6, top-2, balancing
    buffer) layers; the decoder's (and, for '3.1' / '3.2', the encoder's) attentions are
    ``DifferentialMultiheadAttention`` (custom_transformer.py:610-831) with lambda_init by layer depth; '3.2' is pre-norm.

    Differential attention as the reference wires it: q, k = x Wq^T, x Wk^T (E -> 2E, no bias) rotated through the raw
    (2H, L, B, hd) view, then read through the raw (B, L, 2H, hd) view; v through the raw (B, S, H, hd) view; head pair
    (2h, 2h+1) gives softmax maps A1, A2 and out = RMSNorm_hd((A1 - lambda A2) v_h) (1 - lambda_init); the (B, H, L, hd)
    result is then *viewed* as (L, B, E) (:827) -- which hands row l data of positions other than l, later ones included,
    so a position's output depends on the current length and the model cannot be KV-cached: ``generate`` re-runs the
    decoder on the whole prefix every step (the encoder still runs once).  Here both maps run on the tiled attention
    kernel (two launches over strided views, no copies), the subtraction + sub-norm + scale is one kernel
    (``amt_diff_subln_fwd``) writing the (B, H, L, hd) layout the reference reinterprets.
    """

    def __init__(self, version_name="3.0", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, dropout=0.1,
                 max_sequence_midi=2048, max_sequence_video=300, max_sequence_chord=300, total_vf_dim=0, rms_norm=False,
                 scene_embed=False, chord_embed=False, dropTokenRate=0.0):
        nn.Module.__init__(self)
        if version_name not in ("3.0", "3.1", "3.2"):
            raise ValueError("the reference builds an encoder for '3.0', '3.1' and '3.2' only (:672-690)")
        if n_layers < 3:
            raise IndexError("list index out of range (the reference indexes its n_layers attention modules 0..2, :703-727)")
        from .custom_transformer import RMSNorm
        from .moe import GLUExpert, SharedMoELayer
        from .rotate_operation import RotaryPositionalEmbeddings
        self.nlayers, self.nhead, self.d_model, self.d_ff, self.dropout = n_layers, num_heads, d_model, dim_feedforward, dropout
        self.max_seq_midi, self.max_seq_video, self.max_seq_chord = max_sequence_midi, max_sequence_video, max_sequence_chord
        self.scene_embed, self.chord_embed, self.dropTokenRate, self.version_name = scene_embed, chord_embed, dropTokenRate, version_name
        self.total_vf_dim = total_vf_dim
        self.n_experts, self.n_experts_per_token = 6, 2
        self._learned_pos, self._use_rope = False, True
        self.pre_norm = version_name == "3.2"
        if scene_embed:                          # vf = Linear_vis(features without the scene column) + scene_embedding(offset) (:336-337,481-484)
            self.scene_embedding = nn.Embedding(SCENE_OFFSET_MAX, d_model)
        if chord_embed:
            self.chord_embedding_model = nn.Embedding(CHORD_SIZE, d_model)
            self.chord_embedding_model.weight.requires_grad_(False)
            self._register_load_state_dict_pre_hook(self._resize_chord_table)
        self.embedding = nn.Embedding(CHORD_SIZE, d_model)
        self.embedding_root = nn.Embedding(CHORD_ROOT_SIZE, d_model)
        self.embedding_attr = nn.Embedding(CHORD_ATTR_SIZE, d_model)
        self.Linear_vis = nn.Linear(total_vf_dim, d_model)
        self.Linear_chord = nn.Linear(d_model + 1, d_model)
        self.condition_linear = nn.Linear(1, d_model)
        hd = d_model // num_heads

        def ff(i):
            if i < 3:
                return GLUExpert(d_model, dim_feedforward, dropout)
            return SharedMoELayer(GLUExpert(d_model, dim_feedforward, dropout), d_model, n_experts=self.n_experts,
                                  n_experts_per_token=2, dropout=dropout, balancing=True)

        self.transformer = _TransformerParamsV2(d_model, num_heads, n_layers, ff, norm=RMSNorm)
        for i, lyr in enumerate(self.transformer.decoder.layers):
            lyr.self_attn, lyr.cross_attn = _DiffAttnParams(d_model, hd, i), _DiffAttnParams(d_model, hd, i)
        if version_name != "3.0":                               # '3.0' keeps CustomMultiheadAttention in the encoder (:672-676)
            for i, lyr in enumerate(self.transformer.encoder.layers):
                lyr.self_attn = _DiffAttnParams(d_model, hd, i)
        self.Wout = nn.Linear(d_model, CHORD_SIZE)
        self.softmax = nn.Softmax(dim=-1)
        rope = RotaryPositionalEmbeddings(2 * d_model, max_sequence_video)          # dim = 2 d_model (:658)
        self.register_buffer("_rope_cache", rope.cache.clone(), persistent=False)
        self._max_dec = max_sequence_video
        self._derived_sig = None

    def _attention(self, xq, xkv, a, Lq, Lk, B, causal, resid):
        from .. import ops
        if not isinstance(a, _DiffAttnParams):
            return super()._attention(xq, xkv, a, Lq, Lk, B, causal, resid)
        E, H = self.d_model, self.nhead
        hd = E // H
        q = ops.linear(xq, a.q_proj.weight.detach())                                 # (Lq*B, 2E)
        k = ops.linear(xkv, a.k_proj.weight.detach())
        v = ops.linear(xkv, a.v_proj.weight.detach())                                # (Lk*B, E)
        if getattr(self, "_clip_rows", False):        # independent clips, clip-major rows: per clip the batch-of-one rotation
            q = ops.rope(q.view(B, Lq, 1, 2 * E), self._rope_cache).view(-1)
            k = ops.rope(k.view(B, Lk, 1, 2 * E), self._rope_cache).view(-1)
        else:
            q = ops.rope(q.view(2 * H, Lq, B, hd), self._rope_cache).view(-1)        # raw (2H, L, B, hd) view (:779-785)
            k = ops.rope(k.view(2 * H, Lk, B, hd), self._rope_cache).view(-1)
        # raw (B, L, 2H, hd) / (B, S, H, hd) views of the same memory (:787-789): flat row b*L + l, head j at column j*hd;
        # even heads at head stride 2 hd from offset 0, odd heads from offset hd; outputs (B, H, Lq, hd) contiguous
        o1 = torch.empty(B, H, Lq, hd, device=xq.device, dtype=torch.float32)
        o2 = torch.empty_like(o1)
        st = (Lq * 2 * E, 2 * hd, 2 * E, Lk * 2 * E, 2 * hd, 2 * E, Lk * E, hd, E, H * Lq * hd, Lq * hd, hd)
        scale = hd ** -0.5
        ops.attention(q, k, v, st, B, H, Lq, Lk, hd, causal, scale, o1)
        ops.attention(q[hd:], k[hd:], v, st, B, H, Lq, Lk, hd, causal, scale, o2)
        y = ops.diff_subln(o1, o2, a.subln.weight.detach(), a.lambda_full(), 1.0 - a.lambda_init, eps=a.subln.eps)
        return ops.linear(y.view(Lq * B, E), a.out_proj.weight.detach(), resid=resid)  # attn.view(tgt_len, bsz, E) (:827)

    def _enc_layer(self, src, lyr, S, B):
        if not self.pre_norm:
            return super()._enc_layer(src, lyr, S, B)
        from .. import ops
        h = self._ln(src, lyr.norm1)                                                 # pre-norm (:1241-1249)
        src = self._attention(h, h, lyr.self_attn, S, S, B, False, src)
        return ops.add(src, self._ff(self._ln(src, lyr.norm2), lyr.ff, S, B))

    def _dec_layer(self, t, memory, lyr, L, S, B, causal=True):
        if not self.pre_norm:
            return super()._dec_layer(t, memory, lyr, L, S, B, causal)
        from .. import ops
        h = self._ln(t, lyr.norm1)                                                   # pre-norm (:1277-1292)
        t = self._attention(h, h, lyr.self_attn, L, L, B, causal, t)
        t = self._attention(self._ln(t, lyr.norm2), memory, lyr.cross_attn, L, S, B, False, t)
        return ops.add(t, self._ff(self._ln(t, lyr.norm3), lyr.ff, L, B))

    def generate_batch(self, feature_semantic_list, feature_key, feature_scene_offset, feature_motion, feature_emotion,
                       primer, primer_root, primer_attr, target_seq_length=300, beam=0, beam_chance=1.0, max_conseq_N=0,
                       max_conseq_chord=2, temperature=1.0, sampler="categorical", use_graph=False):
        """`generate` for B clips at once -> (B, T); row b equals `generate` on clip b alone.  V3 has no KV cache (see the
        class docstring): every step re-runs the decoder over the prefix, here for all clips in one pass, each clip computed
        as a batch of one (clip-major rows; the reference's raw views are then the B = 1 ones)."""
        from ..utilities.constants import chord_to_root_attr
        assert (not self.training), "Cannot generate while in training mode"
        if beam > 1 or (beam == 1 and beam_chance < 1.0) or self.dropTokenRate != 0.0:
            return self._generate_clip_by_clip(feature_semantic_list, feature_key, feature_scene_offset, feature_motion, feature_emotion,
                                               primer, primer_root, primer_attr, target_seq_length=target_seq_length, beam=beam,
                                               beam_chance=beam_chance, max_conseq_N=max_conseq_N, max_conseq_chord=max_conseq_chord,
                                               temperature=temperature, sampler=sampler)
        dev = self.Wout.weight.device
        T = int(target_seq_length)
        if T > self._max_dec:
            raise ValueError(f"chord sequence longer than the RoPE cache ({self._max_dec}), like in the reference")
        nb = feature_semantic_list.shape[0]
        prim = [torch.as_tensor(q).long().cpu() for q in (primer, primer_root, primer_attr)]
        prim = [q.unsqueeze(0).expand(nb, -1) if q.dim() == 1 else q for q in prim]
        P = prim[0].shape[1]
        gen = torch.full((nb, T), CHORD_PAD, dtype=torch.long)
        gen_root = torch.full((nb, T), CHORD_ROOT_PAD, dtype=torch.long)
        gen_attr = torch.full((nb, T), CHORD_ATTR_PAD, dtype=torch.long)
        gen[:, :P], gen_root[:, :P], gen_attr[:, :P] = prim
        if self.chord_embed:
            gen_root[:, :P], gen_attr[:, :] = gen[:, :P], 0
        key = feature_key.to(dtype=torch.float32).reshape(-1)
        key = (key.expand(nb) if key.numel() == 1 else key).contiguous()
        memory, _, S = self._encode_memory(feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion, clips=True)
        ra_table = torch.tensor([chord_to_root_attr(i) for i in range(CHORD_END)])

        def next_logits(cur):
            if cur < P:
                return None
            return self._decode(gen_root[:, :cur], gen_attr[:, :cur], key, memory, nb, S, clips=True)[:, cur - 1].cpu()

        n_threads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            return self._lockstep_loop(next_logits, gen, gen_root, gen_attr, ra_table, nb, T, P, beam, max_conseq_N, max_conseq_chord,
                                       temperature, sampler).to(dev)
        finally:
            torch.set_num_threads(n_threads)

    def generate(self, *args, use_cache=False, use_graph=False, **kw):
        """The reference loop; every step re-runs the decoder on the whole prefix (see the class docstring)."""
        if use_cache:
            raise NotImplementedError("V3's attention output view makes earlier rows depend on the sequence length: no KV cache")
        return super().generate(*args, use_cache=False, use_graph=False, **kw)
