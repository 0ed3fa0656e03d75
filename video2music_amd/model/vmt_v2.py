"""``VideoMusicTransformer_V2`` (reference ``model/video_music_transformer.py:316-609``; '2.2' is the reference's default
``music_gen_version``) and the machinery its V1 / V3 siblings inherit: operator composition of the teacher-forced forward, the
KV-cached one-call step, the lockstep batch with the decision on the device.  Importable from
``video2music_amd.model.video_music_transformer`` like in the reference, whose four classes share that module; split out for size."""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from .. import _lib
from ..utilities.constants import (CHORD_ATTR_PAD, CHORD_ATTR_SIZE, CHORD_END, CHORD_PAD, CHORD_ROOT_PAD, CHORD_ROOT_SIZE, CHORD_SIZE,
                                   SCENE_OFFSET_MAX)
from .video_music_transformer import _CAPTURE_LOCK, _AttnParams, _Stack, _TransformerParams


# ==================================================================================================
# VideoMusicTransformer_V2, versions '2.0' / '2.1' / '2.2' (2.2 = the reference's default music_gen_version)
# ==================================================================================================
class _DecoderLayerV2(nn.Module):
    """Keys of custom_transformer.TransformerDecoderLayer (model/custom_transformer.py:1250-1292)."""

    def __init__(self, d_model, head_dim, ff, cross, norm=nn.LayerNorm):
        super().__init__()
        self.self_attn = _AttnParams(d_model, head_dim)
        if cross:
            self.cross_attn = _AttnParams(d_model, head_dim)
        self.ff = ff
        self.norm1 = norm(d_model)
        self.norm2 = norm(d_model)
        if cross:
            self.norm3 = norm(d_model)


class _TransformerParamsV2(nn.Module):
    """Both stacks of the V1 / V2 families: layer i's feed-forward is ``ff(i)``, norms are built by ``norm(d_model)``."""

    def __init__(self, d_model, nhead, n_layers, ff, norm=nn.LayerNorm):
        super().__init__()
        hd = d_model // nhead
        self.encoder = _Stack([_DecoderLayerV2(d_model, hd, ff(i), cross=False, norm=norm) for i in range(n_layers)], d_model, norm)
        self.decoder = _Stack([_DecoderLayerV2(d_model, hd, ff(i), cross=True, norm=norm) for i in range(n_layers)], d_model, norm)
        for q in self.parameters():
            if q.dim() > 1:
                nn.init.xavier_uniform_(q)

    generate_square_subsequent_mask = staticmethod(_TransformerParams.generate_square_subsequent_mask)


class VideoMusicTransformer_V2(nn.Module):
    """Reference ``VideoMusicTransformer_V2`` (model/video_music_transformer.py:316-609), versions '2.2' (generate.py's
    default), '2.1' (same network in eval: its top-k scheduler acts in training only) and '2.0':
    RoPE (cache built for dim=d_model, applied through the raw (H, L, B, hd) view) inside every attention and no additive
    positional encoding -- or, for '2.0', learned positional tables and no rotation; three GLU feed-forward layers then
    three SharedMoELayer(6 experts, top-2) layers in both stacks, post-norm; optionally ``chord_embed=True`` (chord ids
    through a frozen table, the configuration of the Video2music app).  A composition of the library's operator kernels
    (``video2music_amd/ops.py``): ``generate`` runs the video encoder once and the decoder one token at a time over
    cached keys/values (the reference re-runs both stacks on the whole prefix every step, :547-548).
    """

    def __init__(self, version_name="2.0", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, dropout=0.1,
                 max_sequence_midi=2048, max_sequence_video=300, max_sequence_chord=300, total_vf_dim=0, rms_norm=False,
                 scene_embed=False, chord_embed=False, dropTokenRate=0.0, balancing=False):
        super().__init__()
        # the reference matches version strings with `in ('2.0')` -- a substring test on a str -- for the learned
        # positional tables (:375,497) and with tuple membership for RoPE (:379); both rules are kept as they are
        self._learned_pos = version_name in "2.0"
        self._use_rope = (not self._learned_pos) and version_name in ("2.1", "2.2", "2.3")
        if version_name in "2.3":
            raise NotImplementedError("version '2.3' swaps the experts for efficient_kan.KANLinear, a package the reference does not vendor")
        # rms_norm is accepted and has no effect, as in the reference (its RMSNorm branch is commented out, :364-371);
        # '2.1' differs from '2.2' by a top-k scheduler that only acts in training (moe.py:232-236)
        self.nlayers, self.nhead, self.d_model, self.d_ff, self.dropout = n_layers, num_heads, d_model, dim_feedforward, dropout
        self.max_seq_midi, self.max_seq_video, self.max_seq_chord = max_sequence_midi, max_sequence_video, max_sequence_chord
        self.scene_embed, self.chord_embed, self.dropTokenRate, self.version_name = scene_embed, chord_embed, dropTokenRate, version_name
        self.total_vf_dim = total_vf_dim
        self.n_experts, self.n_experts_per_token = 6, 2
        if scene_embed:                          # vf = Linear_vis(features without the scene column) + scene_embedding(offset) (:336-337,481-484)
            self.scene_embedding = nn.Embedding(SCENE_OFFSET_MAX, d_model)
        if chord_embed:
            # the reference fills this frozen table from a gensim Word2Vec file (:340-344); here it arrives with the
            # state_dict (key chord_embedding_model.weight, any number of rows >= the ids fed; vector size = d_model)
            self.chord_embedding_model = nn.Embedding(CHORD_SIZE, d_model)
            self.chord_embedding_model.weight.requires_grad_(False)
            self._register_load_state_dict_pre_hook(self._resize_chord_table)
        self.embedding = nn.Embedding(CHORD_SIZE, d_model)
        self.embedding_root = nn.Embedding(CHORD_ROOT_SIZE, d_model)
        self.embedding_attr = nn.Embedding(CHORD_ATTR_SIZE, d_model)
        self.Linear_vis = nn.Linear(total_vf_dim, d_model)
        self.Linear_chord = nn.Linear(d_model + 1, d_model)
        self.condition_linear = nn.Linear(1, d_model)
        if self._learned_pos:
            self.positional_embedding = nn.Embedding(max_sequence_chord, d_model)
            self.positional_embedding_video = nn.Embedding(max_sequence_video, d_model)
        from .moe import GLUExpert, SharedMoELayer

        def ff(i):
            if i < 3:                                            # rate = 3 shallow layers (:409-414)
                return GLUExpert(d_model, dim_feedforward, dropout)
            return SharedMoELayer(GLUExpert(d_model, dim_feedforward, dropout), d_model, n_experts=self.n_experts,
                                  n_experts_per_token=2, dropout=dropout, balancing=balancing)

        # three shallow layers whatever n_layers says, then n_layers - 3 deep ones (:411-416): n_layers < 3 still builds three
        self.transformer = _TransformerParamsV2(d_model, num_heads, max(3, n_layers), ff)
        self.Wout = nn.Linear(d_model, CHORD_SIZE)
        self.softmax = nn.Softmax(dim=-1)
        if self._use_rope:
            from .rotate_operation import RotaryPositionalEmbeddings
            rope = RotaryPositionalEmbeddings(d_model, max_sequence_video)      # dim = d_model, not head_dim (:380)
            self.register_buffer("_rope_cache", rope.cache.clone(), persistent=False)
        else:
            self._rope_cache = None
        # longest chord sequence: the RoPE cache caps it at max_sequence_video (rotate_operation.py:148), the learned
        # table at max_sequence_chord; with neither there is no cap but the caches need a size
        self._max_dec = max_sequence_video if self._use_rope else max_sequence_chord
        self._derived_sig = None

    def _resize_chord_table(self, state_dict, prefix, *_):
        w = state_dict.get(prefix + "chord_embedding_model.weight")
        if w is not None and tuple(w.shape) != tuple(self.chord_embedding_model.weight.shape):
            if w.dim() != 2 or w.shape[1] != self.d_model:
                raise ValueError("chord_embedding_model.weight must be (n_chords, d_model)")
            cur = self.chord_embedding_model.weight
            self.chord_embedding_model.weight = nn.Parameter(torch.empty(w.shape, dtype=cur.dtype, device=cur.device), requires_grad=False)

    # ---- derived tensors (rebuilt when a parameter changes): Linear_chord tables, padded Linear_vis ----
    def _derived(self):
        from .. import ops
        dev = self.Wout.weight.device
        if dev.type != "cuda":
            raise _lib.AmtError("VideoMusicTransformer_V2 runs on an MI355X only; video2music_amd has no CPU fallback")
        srcs = [self.Linear_chord.weight, self.embedding_root.weight, self.embedding_attr.weight, self.Linear_vis.weight]
        if self.chord_embed:
            srcs.append(self.chord_embedding_model.weight)
        if self._learned_pos:
            srcs += [self.positional_embedding.weight, self.positional_embedding_video.weight]
        sig = tuple((q.data_ptr(), q._version) for q in srcs)
        if sig != self._derived_sig:
            d, F = self.d_model, self.total_vf_dim
            Wc = self.Linear_chord.weight.detach()
            Wc_main = Wc[:, :d].contiguous()
            self._wkey = Wc[:, d].contiguous()
            if self.chord_embed:
                # x = chord_embedding_model(x) (:431-432): one table indexed by the chord id; the attr slot adds a zero row
                self._PR = ops.linear(self.chord_embedding_model.weight.detach().contiguous(), Wc_main)
                self._PA = torch.zeros(CHORD_ATTR_SIZE, d, device=dev)
            else:
                self._PR = ops.linear(self.embedding_root.weight.detach().contiguous(), Wc_main)
                self._PA = ops.linear(self.embedding_attr.weight.detach().contiguous(), Wc_main)
            self._wvis_cache = {}
            # positional rows added to the chord embedding: the learned table of version '2.0' (:497-503) or none
            self._pe_chord = (self.positional_embedding.weight.detach().contiguous() if self._learned_pos
                              else torch.zeros(self._max_dec, d, device=dev))
            self._derived_sig = sig

    def _wvis(self, sem_dim):
        """Linear_vis.weight laid out for the rows of `concat_features` ([semantic | scene | motion | emotion], zero-padded
        to a multiple of 32 columns).  With scene_embed the reference leaves the scene column out of the features (:463-465):
        the weight then gets a zero column at that place, so the same rows serve."""
        if sem_dim not in self._wvis_cache:
            W = self.Linear_vis.weight.detach()
            d, F = W.shape
            cols = F + 1 if self.scene_embed else F
            Fpad = (cols + 31) // 32 * 32
            Wv = torch.zeros(d, Fpad, device=W.device)
            if self.scene_embed:
                Wv[:, :sem_dim], Wv[:, sem_dim + 1:cols] = W[:, :sem_dim], W[:, sem_dim:]
            else:
                Wv[:, :F] = W
            self._wvis_cache[sem_dim] = (Wv, Fpad)
        return self._wvis_cache[sem_dim]

    def _attention(self, xq, xkv, a, Lq, Lk, B, causal, resid):
        """xq (Lq*B, E), xkv (Lk*B, E) seq-first rows; returns out-proj(attn) + resid."""
        from .. import ops
        E, H = self.d_model, self.nhead
        hd = E // H
        W, b = a.in_proj_weight.detach(), a.in_proj_bias.detach()
        q = ops.linear(xq, W[:E], b[:E])
        k = ops.linear(xkv, W[E:2 * E], b[E:2 * E])
        v = ops.linear(xkv, W[2 * E:], b[2 * E:])
        clips = getattr(self, "_clip_rows", False)
        if self._rope_cache is not None and clips:
            # independent clips, rows clip-major (B*L, E): each clip gets what the raw view does for a batch of one, i.e.
            # pair i of the E-wide vector at position l rotated by cache[l][i] -- one launch for all clips
            if self._rope_cache.shape[1] * 2 == E:
                q = ops.rope(q.view(B, Lq, 1, E), self._rope_cache).view(Lq * B, E)
                k = ops.rope(k.view(B, Lk, 1, E), self._rope_cache).view(Lk * B, E)
            else:       # a cache built for another width (V3 '3.0': dim = 2 d_model): the raw batch-of-one view, clip by clip
                for t_, L_ in ((q, Lq), (k, Lk)):
                    for c in range(B):
                        rows = t_[c * L_:(c + 1) * L_].view(H, L_, 1, hd)
                        ops.rope(rows, self._rope_cache, out=rows)
        elif self._rope_cache is not None:
            q = ops.rope(q.view(H, Lq, B, hd), self._rope_cache).view(Lq * B, E)       # raw (H, L, B, hd) view (:1041-1053)
            k = ops.rope(k.view(H, Lk, B, hd), self._rope_cache).view(Lk * B, E)
        o = torch.empty(Lq * B, E, device=xq.device, dtype=torch.float32)
        # b, h, l strides of q, k, v, o: (L, B, E) seq-first buffers, or (B, L, E) clip-major ones
        st = ((Lq * E, hd, E) + (Lk * E, hd, E) * 2 + (Lq * E, hd, E)) if clips else (E, hd, B * E) * 4
        ops.attention(q, k, v, st, B, H, Lq, Lk, hd, causal, 1.0 / math.sqrt(hd), o)
        return ops.linear(o, a.out_proj.weight.detach(), a.out_proj.bias.detach(), resid=resid)

    def _ff(self, x, ff, L, B):
        from .. import ops
        from .moe import GLUExpert, SiLUExpert
        if isinstance(ff, (GLUExpert, SiLUExpert)):
            return ops.glu(x, ff)
        return ff(x.view(L, B, self.d_model)).reshape(L * B, self.d_model)

    def _ln(self, t, n, resid=None):
        from .. import ops
        if not isinstance(n, nn.LayerNorm):                      # RMSNorm (the V1 family with rms_norm=True)
            return ops.rmsnorm(t, n.weight.detach(), resid=resid, eps=n.eps)
        return ops.layernorm(t, n.weight.detach(), n.bias.detach(), resid=resid, eps=n.eps)

    def _encode_memory(self, feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion, clips=False):
        """Video stream + encoder stack (:455-487 and the encoder half of :505): (S*B, d) seq-first rows, B, S.
        clips=True: the B clips are encoded as B independent batches of one (what B calls with one clip each compute; for
        B > 1 the reference's raw RoPE view ties the clips of a batch together) in one pass; rows come back clip-major."""
        from .. import ops
        self._derived()
        dev = self.Wout.weight.device
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        sem, scene, emotion, motion = f32(feature_semantic_list), f32(feature_scene_offset), f32(feature_emotion), f32(feature_motion)
        if motion.dim() == 2:
            motion = motion.unsqueeze(-1).contiguous()
        B, S, d = sem.shape[0], sem.shape[1], self.d_model
        if S > self.max_seq_video and (self._use_rope or self._learned_pos):
            raise ValueError(f"video longer than the positional table ({self.max_seq_video}), like in the reference")
        pos_rows = None
        if self._learned_pos:                                   # vf += positional_embedding_video(arange(S)) (:499-501)
            pos_rows = self.positional_embedding_video.weight.detach()[:S].unsqueeze(0).expand(B, S, d).contiguous().view(B * S, d)
        Wv, Fpad = self._wvis(sem.shape[2])
        if self.scene_embed:                                    # + scene_embedding(feature_scene_offset.int()) (:481-484)
            srows = self.scene_embedding.weight.detach()[scene.to(torch.int32).long()].reshape(B * S, d).contiguous()
            pos_rows = srows if pos_rows is None else ops.add(pos_rows, srows)
        if self.dropTokenRate != 0.0:
            # Drop Tokens (:193-197, 488-492, 798-802): rows of (Linear_vis(.) + scene rows) zeroed by a fresh
            # `torch.rand(B, S) > rate` in EVERY forward, eval mode included; the positional rows come after it.  The draw is the
            # reference's own call (default CPU generator), so torch.manual_seed pins the same mask in both implementations.
            keep = (torch.rand(B, S) > self.dropTokenRate).float().to(dev).reshape(B * S).contiguous()
            srows = None
            if self.scene_embed:
                srows = self.scene_embedding.weight.detach()[scene.to(torch.int32).long()].reshape(B * S, d).contiguous()
            lp = None
            if self._learned_pos:
                lp = self.positional_embedding_video.weight.detach()[:S].unsqueeze(0).expand(B, S, d).contiguous().view(B * S, d)
            vf = ops.linear(ops.concat_features(sem, scene, motion, emotion, Fpad), Wv, self.Linear_vis.bias.detach(), resid=srows)
            vf = ops.row_scale_add(vf, keep, lp)
        else:
            vf = ops.linear(ops.concat_features(sem, scene, motion, emotion, Fpad), Wv, self.Linear_vis.bias.detach(), resid=pos_rows)
        src = vf if clips else vf.view(B, S, d).permute(1, 0, 2).contiguous().view(S * B, d)
        self._clip_rows = bool(clips)
        try:
            for lyr in self.transformer.encoder.layers:
                src = self._enc_layer(src, lyr, S, B)
        finally:
            self._clip_rows = False
        return self._ln(src, self.transformer.encoder.norm), B, S

    def _enc_layer(self, src, lyr, S, B):
        """Post-norm encoder layer (custom_transformer.py:1233-1240)."""
        src = self._ln(self._attention(src, src, lyr.self_attn, S, S, B, False, src), lyr.norm1)
        return self._ln(self._ff(src, lyr.ff, S, B), lyr.norm2, resid=src)

    def _dec_layer(self, t, memory, lyr, L, S, B, causal=True):
        """Post-norm decoder layer (custom_transformer.py:1262-1276)."""
        t = self._ln(self._attention(t, t, lyr.self_attn, L, L, B, causal, t), lyr.norm1)
        t = self._ln(self._attention(t, memory, lyr.cross_attn, L, S, B, False, t), lyr.norm2)
        return self._ln(self._ff(t, lyr.ff, L, B), lyr.norm3, resid=t)

    def _decode(self, x_root, x_attr, feature_key, memory, B, S, clips=False, causal=True):
        """Chord stream + decoder stack + Wout (:437-452, :490-516) over a precomputed encoder memory.  clips=True: the B
        rows are independent clips (each computed as a batch of one), `memory` clip-major as `_encode_memory(clips=True)`
        returns it."""
        from .. import ops
        dev = self.Wout.weight.device
        L, d = x_root.shape[1], self.d_model
        if L > self._max_dec:
            raise ValueError(f"chord sequence longer than the positional table ({self._max_dec}), like in the reference")
        key = feature_key.to(device=dev, dtype=torch.float32).reshape(-1)
        key = key.expand(B).contiguous() if key.numel() == 1 else key.contiguous()
        xf = ops.chord_embed(x_root.to(dev).long().contiguous(), x_attr.to(dev).long().contiguous(), key, self._PR, self._PA,
                             self._wkey, self.Linear_chord.bias.detach(), self._pe_chord)
        t = xf if clips else xf.view(B, L, d).permute(1, 0, 2).contiguous().view(L * B, d)
        self._clip_rows = bool(clips)
        try:
            for lyr in self.transformer.decoder.layers:
                t = self._dec_layer(t, memory, lyr, L, S, B, causal)
        finally:
            self._clip_rows = False
        t = self._ln(t, self.transformer.decoder.norm)
        if not clips:
            t = t.view(L, B, d).permute(1, 0, 2).contiguous().view(B * L, d)
        return ops.linear(t, self.Wout.weight.detach(), self.Wout.bias.detach()).view(B, L, CHORD_SIZE)

    # ---- KV-cached decode of one clip (B = 1) ----------------------------------------------------------------------
    # For B = 1 the raw (H, L, B, hd) RoPE view is ordinary interleaved-pair RoPE over the full d_model vector at the
    # true position, so row t of the decoder depends on tokens <= t only and the K/V rows of earlier
    # positions never change: the decoder can run one token at a time over cached keys/values.  Every kernel computes
    # its rows independently and in the same order as in the full forward, so the step's logits equal row t of `_decode`.
    def _cache_init(self, memory, S):
        """`memory`: one clip's encoder output (S, E), or a list of them for the lockstep step of several clips (every cache then carries a
        leading clip dimension).  Builds the step's pointer table, packed / folded weights and K/V caches: `v2_step_table.build_step_table`."""
        from .v2_step_table import build_step_table
        return build_step_table(self, memory, S)

    def _decode_step_native(self, root, attr, key, t, st, state=None):
        """`_decode_step` issued by one library call (amt_v2_step): logits (159,) for input position t.  With `state`
        (int32 device tensor {position, root, attr}) the step reads those from device memory and increments the position."""
        _lib.call("amt_v2_step", st["tab"], len(self.transformer.decoder.layers), self.nhead, self.d_model, st["dff"], self.n_experts, st["S"],
                  self._max_dec, int(t), int(root), int(attr), float(key), _lib.ptr(state), _lib.ptr(st["logits"]),
                  _lib.ptr(st["ws"]), _lib.stream_ptr())
        return st["logits"]

    def _step_graph(self, key, st, root0, attr0):
        """Runs position 0 eagerly through the device-state form of the step, then captures that step once: every later
        token is `state[1:] = (root, attr)` + one graph replay (≈130 launches at replay cost instead of launch cost)."""
        dev = st["ws"].device
        state = torch.tensor([0, int(root0), int(attr0)], dtype=torch.int32, device=dev)
        self._decode_step_native(0, 0, key, 0, st, state)             # position 0 (also the warm-up the capture needs)
        torch.cuda.current_stream().synchronize()
        g = torch.cuda.CUDAGraph()
        # several clips may be decoded by concurrent host threads (one stream each): captures are serialised, and
        # thread-local capture mode keeps the other threads' launches from invalidating this one
        with _CAPTURE_LOCK, torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._decode_step_native(0, 0, key, 0, st, state)
        return g, state

    def _decode_step(self, root_t, attr_t, key, t, st):
        """Logits (159,) for input position t given the cached positions < t (appends position t to the caches)."""
        from .. import ops
        E, H = self.d_model, self.nhead
        hd = E // H
        scale = 1.0 / math.sqrt(hd)
        strides = (E, hd, E) * 4
        x = ops.chord_embed(root_t, attr_t, key, self._PR, self._PA, self._wkey, self.Linear_chord.bias.detach(), self._pe_chord[t:t + 1])
        rope = self._rope_cache
        for lyr, (kc, vc), (kx, vx) in zip(self.transformer.decoder.layers, st["self"], st["cross"]):
            a = lyr.self_attn
            qkv = ops.linear(x, a.in_proj_weight.detach(), a.in_proj_bias.detach())                     # (1, 3E)
            if rope is not None:
                q = ops.rope(qkv[:, :E].view(1, 1, 1, E), rope, pos=t).view(1, E)
                ops.rope(qkv[:, E:2 * E].view(1, 1, 1, E), rope, pos=t, out=kc[t:t + 1].view(1, 1, 1, E))
            else:
                q = qkv[:, :E].contiguous()
                kc[t:t + 1].copy_(qkv[:, E:2 * E])
            vc[t:t + 1].copy_(qkv[:, 2 * E:])
            o = torch.empty(1, E, device=x.device, dtype=torch.float32)
            ops.attention(q, kc, vc, strides, 1, H, 1, t + 1, hd, False, scale, o)
            x = self._ln(ops.linear(o, a.out_proj.weight.detach(), a.out_proj.bias.detach(), resid=x), lyr.norm1)
            a = lyr.cross_attn
            W, b = a.in_proj_weight.detach(), a.in_proj_bias.detach()
            q = ops.linear(x, W[:E], b[:E])
            if rope is not None:
                q = ops.rope(q.view(1, 1, 1, E), rope, pos=t).view(1, E)
            ops.attention(q, kx, vx, strides, 1, H, 1, st["S"], hd, False, scale, o)
            x = self._ln(ops.linear(o, a.out_proj.weight.detach(), a.out_proj.bias.detach(), resid=x), lyr.norm2)
            x = self._ln(self._ff(x, lyr.ff, 1, 1), lyr.norm3, resid=x)
        x = self._ln(x, self.transformer.decoder.norm)
        return ops.linear(x, self.Wout.weight.detach(), self.Wout.bias.detach())[0]

    def _decode_step_ops(self, root, attr, key, t, st):
        """`_decode_step` with the call shape of `_decode_step_native` (host ints in, logits left in st["logits"])."""
        dev = st["logits"].device
        r = torch.tensor([[int(root)]], device=dev, dtype=torch.long)
        a = torch.tensor([[int(attr)]], device=dev, dtype=torch.long)
        st["logits"].copy_(self._decode_step(r, a, torch.tensor([float(key)], device=dev), int(t), st))
        return st["logits"]

    def forward(self, x, x_root, x_attr, feature_semantic_list, feature_key, feature_scene_offset, feature_motion,
                feature_emotion, mask=True):
        memory, B, S = self._encode_memory(feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion)
        assert x_root.shape[0] == B, f"{x_root.shape[0]} chord sequences but {B} clips of video features"
        if self.chord_embed:                     # the chord ids themselves index the frozen table (:431-432)
            x_root, x_attr = x, torch.zeros_like(x)
        # mask other than True: tgt_mask=None (:440-443), the decoder self-attention sees every position
        return self._decode(x_root, x_attr, feature_key, memory, B, S, causal=mask is True)

    def _generate_clip_by_clip(self, sem, key, scene, motion, emotion, primer, primer_root, primer_attr, **kw):
        """The options whose reference semantics are per call and host-side (top-k branch with beam > 1 or beam_chance < 1:
        python's `random` per step; dropTokenRate: a fresh torch.rand mask per forward): the clips run one after the other
        through `generate`, in order, and row 0 of each result (the top-1 row) is returned."""
        nb = sem.shape[0]
        prim = [torch.as_tensor(q).long().cpu() for q in (primer, primer_root, primer_attr)]
        prim = [q.unsqueeze(0).expand(nb, -1) if q.dim() == 1 else q for q in prim]
        k = key.reshape(-1)
        k = k.expand(nb) if k.numel() == 1 else k
        rows = [self.generate(sem[c:c + 1], k[c:c + 1], scene[c:c + 1], motion[c:c + 1], emotion[c:c + 1], prim[0][c], prim[1][c],
                              prim[2][c], decision="host", **kw)[:1] for c in range(nb)]
        return torch.cat(rows, dim=0)

    def _step_batch(self, st, keys, state):
        _lib.call("amt_v2_step_batch", st["tab"], len(self.transformer.decoder.layers), self.nhead, self.d_model, st["dff"], self.n_experts, st["S"],
                  self._max_dec, st["B"], _lib.ptr(keys), _lib.ptr(state), _lib.ptr(st["logits"]), _lib.ptr(st["ws"]), _lib.stream_ptr())

    def generate_batch(self, feature_semantic_list, feature_key, feature_scene_offset, feature_motion, feature_emotion,
                       primer, primer_root, primer_attr, target_seq_length=300, beam=0, beam_chance=1.0, max_conseq_N=0,
                       max_conseq_chord=2, temperature=1.0, sampler="categorical", use_graph=True, decision="device",
                       uniforms=None):
        """`generate` for B clips at once -> LongTensor (B, T); row b equals `generate` on clip b alone (the reference
        generates one clip per call).  Features (B, S, .), key (B,) / (B, 1); primers (P,) shared or (B, P).

        The clips advance in lockstep through one captured step graph (`amt_v2_step_batch`): each projection reads its
        weights once per step for all clips.  The video encoder runs once over all clips as independent batches of one (for B > 1
        the reference's raw RoPE view would tie the clips of a batch together).

        ``decision="device"`` (default): the per-step decision of the reference loop (:547-600: temperature softmax[:157],
        suppression, top-1 / arg-max / Categorical draw, root / attr feedback) runs in `amt_v2_decide_batch` inside the same
        captured graph, so a generate is T-1 graph replays with no host round trip; the Categorical draw is the inverse CDF at
        ``uniforms`` (T, B) (default ``torch.rand`` on the device: ``torch.manual_seed`` repeats a run).  ``decision="host"``
        keeps the round-1 loop (logits copied to the host every step, torch's own Categorical)."""
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
            raise ValueError(f"chord sequence longer than the positional table ({self._max_dec}), like in the reference")
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
        rows, _, S = self._encode_memory(feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion, clips=True)
        mems = [rows[c * S:(c + 1) * S] for c in range(nb)]
        st = self._cache_init(mems, S)
        if decision not in ("device", "host"):
            raise ValueError(f"unknown decision {decision!r}")
        if (nb == 1 and decision == "host") or not st["native"]:      # layers of unequal width, or the host loop for one clip
            rows = [self.generate(feature_semantic_list[c:c + 1], key[c], feature_scene_offset[c:c + 1], feature_motion[c:c + 1],
                                  feature_emotion[c:c + 1], prim[0][c], prim[1][c], prim[2][c], target_seq_length=T, beam=beam,
                                  beam_chance=beam_chance, max_conseq_N=max_conseq_N, max_conseq_chord=max_conseq_chord,
                                  temperature=temperature, sampler=sampler, use_graph=use_graph, decision="host") for c in range(nb)]
            return torch.cat(rows)
        keys = key.to(dev)
        if decision == "device":
            return self._lockstep_device(st, keys, gen, gen_root, gen_attr, nb, T, P, beam, max_conseq_N, max_conseq_chord,
                                         temperature, sampler, use_graph, uniforms)
        state = torch.zeros(1 + 2 * nb, dtype=torch.int32, device=dev)
        ra_table = torch.tensor([chord_to_root_attr(i) for i in range(CHORD_END)])          # id -> (root, attr) feedback (:578-597)

        def feed(t):
            state[1:] = torch.cat((gen_root[:, t], gen_attr[:, t])).to(torch.int32)

        feed(0)
        self._step_batch(st, keys, state)                               # position 0, eagerly (the warm-up a capture needs)
        graph = None
        if use_graph and T > 2:
            torch.cuda.current_stream().synchronize()
            graph = torch.cuda.CUDAGraph()
            with _CAPTURE_LOCK, torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self._step_batch(st, keys, state)
        # the host decision works on (B, 159) tensors: with torch's intra-op pool awake each such op costs milliseconds
        # (6 ms for the softmax alone, found with cProfile), so the loop runs with one intra-op thread
        n_threads = torch.get_num_threads()
        torch.set_num_threads(1)
        def next_logits(cur):                                           # logits of input position cur-1, (B, 159) on the host
            if cur - 1 > 0:
                feed(cur - 1)
                graph.replay() if graph is not None else self._step_batch(st, keys, state)
            return st["logits"].cpu() if cur >= P else None

        try:
            return self._lockstep_loop(next_logits, gen, gen_root, gen_attr, ra_table, nb, T, P, beam, max_conseq_N,
                                       max_conseq_chord, temperature, sampler).to(dev)
        finally:
            torch.set_num_threads(n_threads)

    def _lockstep_device(self, st, keys, gen, gen_root, gen_attr, nb, T, P, beam, max_conseq_N, max_conseq_chord, temperature,
                         sampler, use_graph, uniforms):
        """Lockstep generate with the decision on the device: amt_v2_step_decide_batch for position 0 issued eagerly, then that
        chain captured once and replayed for positions 1 .. T-2; one synchronisation at the end."""
        dev = keys.device
        if sampler not in ("categorical", "argmax"):
            raise ValueError(f"unknown sampler {sampler!r}")
        tokens, roots, attrs = gen.to(dev).contiguous(), gen_root.to(dev).contiguous(), gen_attr.to(dev).contiguous()
        unif = None
        if beam == 0 and sampler == "categorical":
            unif = (torch.rand(T, nb, device=dev) if uniforms is None else torch.as_tensor(uniforms, dtype=torch.float32).to(dev)).contiguous()
            assert unif.shape == (T, nb), "uniforms must be (target_seq_length, B)"
        state = torch.zeros(2 + 2 * nb, dtype=torch.int32, device=dev)          # {position, root[B], attr[B], ticket}
        state[1:1 + 2 * nb] = torch.cat((gen_root[:, 0], gen_attr[:, 0])).to(torch.int32)

        A = _lib.addr
        step_args = _lib.V2StepArgs(C.cast(st["tab"], C.c_void_p), len(self.transformer.decoder.layers), self.nhead, self.d_model, st["dff"],
                                    self.n_experts, st["S"], self._max_dec, st["B"], A(keys), A(state), A(st["logits"]), A(st["ws"]))
        decide_args = _lib.V2DecideArgs(A(tokens), A(roots), A(attrs), T, P, int(beam), int(max_conseq_N), int(max_conseq_chord),
                                        float(temperature), A(unif), int(bool(self.chord_embed)))

        def step(first):
            # one launch chain: the decoder step, then the decision, the next position's chord-stream row and the position
            # advance in one kernel (amt_v2_step_decide_batch; `first`: the chain starts with the embedding of position 0)
            _lib.call("amt_v2_step_decide_batch", C.byref(step_args), C.byref(decide_args), int(first), _lib.stream_ptr())

        step(True)                                                      # position 0 (the warm-up a capture needs)
        if T > 2:
            if use_graph:
                # the chain is captured twice: STEPS_PER_GRAPH steps in one graph (one host launch per 8 positions) and a single
                # step for the remainder (a step past T - 2 would write a cache row that does not exist)
                k = max(1, min(int(os.environ.get("AMT_V2_STEPS_PER_GRAPH", "8")), T - 2))
                torch.cuda.current_stream().synchronize()
                many, one = torch.cuda.CUDAGraph(), None
                with _CAPTURE_LOCK, torch.cuda.graph(many, capture_error_mode="thread_local"):
                    for _ in range(k):
                        step(False)
                # (capturing executes nothing: the positions start after it)
                for _ in range((T - 2) // k):
                    many.replay()
                rest = (T - 2) % k
                if rest:
                    one = torch.cuda.CUDAGraph()
                    with _CAPTURE_LOCK, torch.cuda.graph(one, capture_error_mode="thread_local"):
                        step(False)
                    for _ in range(rest):
                        one.replay()
            else:
                for _ in range(T - 2):
                    step(False)
        # the captured graphs, `state`, `unif` and the id tables go out of scope with up to T-2 replays still queued: finish them here
        torch.cuda.current_stream().synchronize()
        return tokens

    def _lockstep_loop(self, next_logits, gen, gen_root, gen_attr, ra_table, nb, T, P, beam, max_conseq_N, max_conseq_chord,
                       temperature, sampler):
        """The per-position decision of `generate` (:547-600) for B clips at once; `next_logits(cur)` supplies the logits that
        decide position cur (None inside the primer)."""
        for cur in range(1, T):
            lg = next_logits(cur)
            if lg is None:
                continue
            probs = torch.softmax(lg / temperature, dim=-1)[:, :CHORD_END]     # per row the arithmetic of `generate`
            if beam == 1:
                tok = probs.argmax(-1)                                  # topk(., 1) per clip (:547-560); no root/attr feedback
                gen[:, cur] = tok
                if self.chord_embed:
                    gen_root[:, cur] = tok
                continue
            if max_conseq_N == 0:
                probs[:, 0] = 0.0
            if cur >= max_conseq_chord:
                same = torch.ones(nb, dtype=torch.bool)
                for k in range(1, max_conseq_chord):
                    same &= gen[:, cur - 1] == gen[:, cur - 1 - k]
                probs[same, gen[same, cur - 1]] = 0.0
            if sampler == "argmax":
                tok = (probs / probs.sum(-1, keepdim=True)).argmax(-1)
            else:
                tok = torch.distributions.categorical.Categorical(probs=probs).sample()
            gen[:, cur] = tok
            gen_root[:, cur], gen_attr[:, cur] = (tok, 0) if self.chord_embed else (ra_table[tok, 0], ra_table[tok, 1])
        return gen

    def generate(self, feature_semantic_list=[], feature_key=None, feature_scene_offset=None, feature_motion=None,
                 feature_emotion=None, primer=None, primer_root=None, primer_attr=None, target_seq_length=300, beam=0,
                 beam_chance=1.0, max_conseq_N=0, max_conseq_chord=2, temperature=1.0, sampler="categorical", use_cache=True,
                 use_graph=True, decision="device"):
        """Reference loop (:518-609) for one clip.  The reference re-runs the whole model every step; here the encoder runs
        once and the decoder one token at a time over cached K/V.

        ``decision="device"`` (default, needs `use_cache` and `use_graph`): the clip takes the lockstep step with B = 1 and the
        per-step decision of the reference loop runs inside the captured graph (`generate_batch`): no host round trip per token.
        ``decision="host"``: the round-1 loop — the one-call step with device-routed experts, the decision on the host like
        the reference's python loop (softmax[:157] / temperature, N and repeat suppression, torch's Categorical or arg-max);
        `use_cache=False` keeps the per-step re-forward of the decoder stack; `use_graph=False` issues the cached step eagerly
        instead of replaying a captured graph — required when several host threads generate concurrently (stream capture
        does not tolerate the other threads' synchronisations)."""
        from ..utilities.constants import chord_to_root_attr
        assert (not self.training), "Cannot generate while in training mode"
        import random
        print("Generating sequence of max length:", target_seq_length)
        if decision not in ("device", "host"):
            raise ValueError(f"unknown decision {decision!r}")
        mixed = beam > 1 or (beam == 1 and beam_chance < 1.0)      # the top-k branch as written (:551-561): host loop, (beam, T) rows
        if self.chord_embed and beam > 1:
            raise RuntimeError("chord_embed with beam > 1 feeds `beam` chord rows against one clip of video features: the "
                               "reference fails in the cross-attention at the second step")
        redraw = self.dropTokenRate != 0.0     # every reference step is a full forward with a fresh drop mask (:488-492)
        if redraw:
            use_cache = False
        if decision == "device" and use_cache and use_graph and sampler in ("categorical", "argmax") and not mixed:
            return self.generate_batch(feature_semantic_list, feature_key, feature_scene_offset, feature_motion, feature_emotion,
                                       primer, primer_root, primer_attr, target_seq_length=target_seq_length, beam=beam,
                                       beam_chance=beam_chance, max_conseq_N=max_conseq_N, max_conseq_chord=max_conseq_chord,
                                       temperature=temperature, sampler=sampler, use_graph=True, decision="device")
        dev = self.Wout.weight.device
        T = int(target_seq_length)
        gen = torch.full((1, T), CHORD_PAD, dtype=torch.long)
        gen_root = torch.full((1, T), CHORD_ROOT_PAD, dtype=torch.long)
        gen_attr = torch.full((1, T), CHORD_ATTR_PAD, dtype=torch.long)
        P = len(primer)
        gen[0, :P], gen_root[0, :P], gen_attr[0, :P] = primer.cpu().long(), primer_root.cpu().long(), primer_attr.cpu().long()
        if self.chord_embed:                     # the model input is the chord id (gen_seq, :547-548); the attr slot stays 0
            gen_root[0, :P], gen_attr[0, :] = gen[0, :P], 0
        cur = P
        # the encoder output does not depend on the chords: it is computed once instead of every step (the reference
        # recomputes the identical tensor inside each forward, :547-548)
        memory, B, S = self._encode_memory(feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion)
        assert B == 1, "generate takes one clip, like the reference (:528-530)"
        if T > self._max_dec:
            raise ValueError(f"chord sequence longer than the positional table ({self._max_dec}), like in the reference")
        if use_cache:
            key_val = float(feature_key.reshape(-1)[0])
            st = self._cache_init(memory, S)
            if not st["native"]:
                use_graph = False
            if use_graph:
                graph, state = self._step_graph(key_val, st, gen_root[0, 0], gen_attr[0, 0])      # position 0 done
            else:
                one = self._decode_step_native if st["native"] else self._decode_step_ops
                one(gen_root[0, 0], gen_attr[0, 0], key_val, 0, st)

            def step(t):
                if use_graph:
                    state[1:] = torch.stack((gen_root[0, t], gen_attr[0, t])).to(torch.int32)
                    graph.replay()
                else:
                    one(gen_root[0, t], gen_attr[0, t], key_val, t, st)
            for t in range(1, P - 1):       # primer positions whose logits are not needed: fill the caches
                step(t)
        while cur < T:
            if use_cache:
                if cur - 1 > 0:             # (position 0 already ran; its logits are in st["logits"])
                    step(cur - 1)
                row = st["logits"].cpu()
            else:
                if redraw and cur > P:
                    memory, B, S = self._encode_memory(feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion)
                row = self._decode(gen_root[:, :cur], gen_attr[:, :cur], feature_key, memory, B, S)[0, cur - 1].cpu()
            probs = torch.softmax(row / temperature, dim=-1)[:CHORD_END]
            beam_ran = 2.0 if beam == 0 else random.uniform(0, 1)
            if beam_ran <= beam_chance:
                top_i = torch.topk(probs, beam)[1]          # (:556-561): `beam` copies of row 0, the k best ids in column cur
                gen = gen[top_i // CHORD_SIZE, :]
                gen[..., cur] = top_i % CHORD_SIZE
                if self.chord_embed:        # the ids are the model input here, so the top-1 choice does feed back
                    gen_root[0, cur] = gen[0, cur]
            else:
                if max_conseq_N == 0:
                    probs[0] = 0.0
                if cur >= max_conseq_chord and all(int(gen[0, cur - 1]) == int(gen[0, cur - 1 - k]) for k in range(1, max_conseq_chord)):
                    probs[int(gen[0, cur - 1])] = 0.0
                if sampler == "argmax":
                    tok = int((probs / probs.sum()).argmax())
                else:
                    tok = int(torch.distributions.categorical.Categorical(probs=probs).sample())
                gen[:, cur] = tok
                gen_root[0, cur], gen_attr[0, cur] = (tok, 0) if self.chord_embed else chord_to_root_attr(tok)
            cur += 1
        return gen[:, :cur].to(dev)
