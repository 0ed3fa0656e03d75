"""``VideoMusicTransformer`` — MI355X host module of the Affective Multimodal Transformer.

Drop-in for the reference class of the same name (``model/video_music_transformer.py:910-1132``):
same constructor keywords, same ``state_dict`` keys and shapes (torch's packed ``in_proj_weight``
layout, ``Er`` per decoder layer, the two ``pe`` buffers), same ``forward`` / ``generate``
signatures and return types.  The arithmetic runs in ``libamt_hip.so`` (hand-written gfx950 kernels
behind the C ABI of ``include/amt_hip.h``); PyTorch only owns parameters, device buffers and
streams.  There is no CPU path: calling ``forward``/``generate`` without the HIP library or on a
non-GPU module raises.

Differences a caller can observe, all additive:
  * ``generate`` accepts ``sampler="categorical"`` (default; the reference's
    ``Categorical.sample``, drawn inside the decode graph by inverse CDF from ``torch.rand`` uniforms),
    ``"multinomial"`` (same distribution, ``torch.multinomial`` per step) or ``"argmax"`` (the
    deterministic feedback-greedy decode used for parity, oracle G2);
  * ``generate_batch`` runs B clips at once (the reference is hard-wired to one clip,
    ``:1059-1061``); per clip the result equals a B=1 ``generate``;
  * the video encoder runs once per clip and decoding uses a KV cache — exact for this model
    because its logits at position i do not depend on the current length.
"""
import ctypes as C
import os
import math
import threading

import torch
import torch.nn as nn

from .. import _lib
from ..utilities.constants import (CHORD_ATTR_PAD, CHORD_ATTR_SIZE, CHORD_END, CHORD_PAD, CHORD_ROOT_PAD,
                                   CHORD_ROOT_SIZE, CHORD_SIZE, IS_SEPERATED, SCENE_OFFSET_MAX)

_CAPTURE_LOCK = threading.Lock()
MAX_DECODE_BATCH = 32          # default clips per library call (`model.max_decode_batch`, up to 256); larger batches are processed in slices


class PositionalEncoding(nn.Module):
    """Sinusoidal table of model/positional_encoding.py:7-23 (buffer ``pe`` of shape (max_len,1,d))."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1).contiguous())


class _AttnParams(nn.Module):
    """Parameter container with the key names of torch.nn.MultiheadAttention /
    MultiheadAttentionRPR (model/rpr.py:112-168): in_proj_weight, in_proj_bias, [Er], out_proj.*"""

    def __init__(self, d_model, head_dim, er_len=None):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        if er_len is not None:
            self.Er = nn.Parameter(torch.rand(er_len, head_dim))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _EncoderLayerParams(nn.Module):
    def __init__(self, d_model, head_dim, d_ff):
        super().__init__()
        self.self_attn = _AttnParams(d_model, head_dim)
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _DecoderLayerParams(nn.Module):
    """Keys of TransformerDecoderLayerRPR (model/rpr.py:37-53)."""

    def __init__(self, d_model, head_dim, d_ff, er_len):
        super().__init__()
        self.self_attn = _AttnParams(d_model, head_dim, er_len)
        self.multihead_attn = _AttnParams(d_model, head_dim)
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)


class _Stack(nn.Module):
    def __init__(self, layers, d_model, norm=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = nn.LayerNorm(d_model) if norm is None else norm(d_model)


class _TransformerParams(nn.Module):
    """Key layout of ``nn.Transformer(custom_decoder=TransformerDecoderRPR)`` as built at
    model/video_music_transformer.py:963-971."""

    def __init__(self, d_model, nhead, n_layers, d_ff, er_len):
        super().__init__()
        hd = d_model // nhead
        self.encoder = _Stack([_EncoderLayerParams(d_model, hd, d_ff) for _ in range(n_layers)], d_model)
        self.decoder = _Stack([_DecoderLayerParams(d_model, hd, d_ff, er_len) for _ in range(n_layers)], d_model)
        for p in self.parameters():                 # nn.Transformer._reset_parameters
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    @staticmethod
    def generate_square_subsequent_mask(sz, device=None, dtype=None):
        return torch.triu(torch.full((sz, sz), float("-inf"), dtype=dtype or torch.float32, device=device), diagonal=1)


class VideoMusicTransformer(nn.Module):
    def __init__(self, n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024,
                 dropout=0.1, max_sequence_midi=2048, max_sequence_video=300,
                 max_sequence_chord=300, total_vf_dim=0, rpr=False, scene_embed=False,
                 chord_embed=False):
        super().__init__()
        # rpr=False (the class default; generate.py passes RPR=True) selects torch's stock decoder layers (:957-962): the same
        # layer without the relative-position table.  It runs on the same kernels with an all-zero Er (the bias q.Er is then
        # exactly 0), which is not part of the state_dict.
        self.nlayers = n_layers
        self.nhead = num_heads
        self.d_model = d_model
        self.d_ff = dim_feedforward
        self.dropout = dropout
        self.max_seq_midi = max_sequence_midi
        self.max_seq_video = max_sequence_video
        self.max_seq_chord = max_sequence_chord
        self.rpr = rpr
        self.scene_embed = scene_embed
        self.chord_embed = chord_embed
        self.total_vf_dim = total_vf_dim

        if scene_embed:                          # vf = Linear_vis(features without the scene column) + scene_embedding(offset) (:926-928,1016-1027)
            self.scene_embedding = nn.Embedding(SCENE_OFFSET_MAX, d_model)
        if chord_embed:
            # the reference fills this from a gensim Word2Vec file (:931-935) that the tree does not ship; here the frozen table
            # arrives with the state_dict (key chord_embedding_model.weight, any number of rows >= the ids fed, width d_model)
            self.chord_embedding_model = nn.Embedding(CHORD_SIZE, d_model)
            self.chord_embedding_model.weight.requires_grad_(False)
            self._register_load_state_dict_pre_hook(self._resize_chord_table)
        self.embedding = nn.Embedding(CHORD_SIZE, d_model)
        self.embedding_root = nn.Embedding(CHORD_ROOT_SIZE, d_model)
        self.embedding_attr = nn.Embedding(CHORD_ATTR_SIZE, d_model)
        self.Linear_vis = nn.Linear(total_vf_dim, d_model)
        self.Linear_chord = nn.Linear(d_model + 1, d_model)
        self.positional_encoding = PositionalEncoding(d_model, dropout, max_sequence_chord)
        self.positional_encoding_video = PositionalEncoding(d_model, dropout, max_sequence_video)
        self.condition_linear = nn.Linear(1, d_model)
        self.transformer = _TransformerParams(d_model, num_heads, n_layers, dim_feedforward, max_sequence_chord if rpr else None)
        self.Wout_root = nn.Linear(d_model, CHORD_ROOT_SIZE)
        self.Wout_attr = nn.Linear(d_model, CHORD_ATTR_SIZE)
        self.Wout = nn.Linear(d_model, CHORD_SIZE)
        self.softmax = nn.Softmax(dim=-1)

        self._handle = None
        self._weights_sig = None
        # clips per library call (= per captured decode chain).  The K/V caches and workspaces of the handle are sized for it
        # (config 2: ~2 GB at 32, ~8 GB at 128), so it must be set before the first forward / generate; larger batches are sliced.
        self.max_decode_batch = MAX_DECODE_BATCH

    # ------------------------------------------------------------------------------------------
    # library handle / weight upload
    # ------------------------------------------------------------------------------------------
    def _device(self):
        dev = self.Wout.weight.device
        if dev.type != "cuda":
            raise _lib.AmtError("VideoMusicTransformer runs on an MI355X only: move the module to a GPU "
                                "(`.to('cuda')`); video2music_amd has no CPU fallback")
        return dev

    def _resize_chord_table(self, state_dict, prefix, *_):
        w = state_dict.get(prefix + "chord_embedding_model.weight")
        if w is not None and tuple(w.shape) != tuple(self.chord_embedding_model.weight.shape):
            if w.dim() != 2 or w.shape[1] != self.d_model:
                raise ValueError("chord_embedding_model.weight must be (n_chords, d_model)")
            cur = self.chord_embedding_model.weight
            self.chord_embedding_model.weight = nn.Parameter(torch.empty(w.shape, dtype=cur.dtype, device=cur.device), requires_grad=False)

    def _ensure_handle(self, sem_dim=None):
        dev = self._device()
        torch.cuda.set_device(dev)
        if self.scene_embed:
            # the library keeps the scene column in its feature rows; its weight column is zero and the embedding rows enter as a
            # residual of the Linear_vis product (amt_encode_resid).  The column sits right behind the semantic features.
            if sem_dim is None:
                sem_dim = getattr(self, "_scene_col", None)
            assert sem_dim is not None, "scene_embed: the semantic feature width is known at the first forward / generate"
            if getattr(self, "_scene_col", None) != sem_dim:
                self._scene_col, self._weights_sig = sem_dim, None
        if self._handle is None:
            cfg = _lib.AmtConfig(self.nlayers, self.nhead, self.d_model, self.d_ff, self.max_seq_video,
                                 self.max_seq_chord, self.total_vf_dim + int(bool(self.scene_embed)), int(self.max_decode_batch))
            h = C.c_void_p()
            _lib.call("amt_create", C.byref(cfg), C.byref(h))
            if self.chord_embed:
                _lib.call("amt_set_option", h, b"chord_embed", 1)
            # diagnostic: the decode step without folded LayerNorms (the library itself reads no environment variable)
            if getattr(self, "decode_chain", None) == "plain" or os.environ.get("AMT_DECODE_CHAIN") == "plain":
                _lib.call("amt_set_option", h, b"decode_chain_plain", 1)
            self._handle = h
            self._weights_sig = None
        sig = tuple((k, v.data_ptr(), v._version) for k, v in self.state_dict().items()) + (bool(IS_SEPERATED),)
        if sig != self._weights_sig:
            torch.cuda.synchronize(dev)
            for name, t in self.state_dict().items():
                t = t.detach().to(torch.float32).contiguous()
                if name == "Linear_vis.weight" and self.scene_embed:
                    c = self._scene_col
                    t = torch.cat([t[:, :c], torch.zeros(t.shape[0], 1, device=t.device), t[:, c:]], dim=1).contiguous()
                if IS_SEPERATED and name in ("Wout.weight", "Wout.bias"):
                    # IS_SEPERATED (utilities/constants.py:11; reference :1036-1040): the two small heads ride in the rows of the
                    # library's one output projection (root rows 0..14, attr rows 15..30, the rest zero); forward() splits them
                    part = name.split(".")[1]
                    r, a = getattr(self.Wout_root, part).detach(), getattr(self.Wout_attr, part).detach()
                    t = torch.zeros_like(t)
                    t[:CHORD_ROOT_SIZE], t[CHORD_ROOT_SIZE:CHORD_ROOT_SIZE + CHORD_ATTR_SIZE] = r.to(t), a.to(t)
                if self.chord_embed:                 # the chord table takes the root table's place, the attr table is all zero
                    if name == "embedding_root.weight":
                        t = self.chord_embedding_model.weight.detach().to(torch.float32).contiguous()
                    elif name == "embedding_attr.weight":
                        t = torch.zeros_like(t)
                shape = (C.c_int64 * t.dim())(*t.shape)
                _lib.call("amt_load_weight", self._handle, name.encode(), _lib.ptr(t), t.dim(), shape)
            if not self.rpr:
                zero = torch.zeros(self.max_seq_chord, self.d_model // self.nhead, device=dev)
                shape = (C.c_int64 * 2)(*zero.shape)
                for i in range(self.nlayers):
                    _lib.call("amt_load_weight", self._handle, f"transformer.decoder.layers.{i}.self_attn.Er".encode(), _lib.ptr(zero), 2, shape)
                self._zero_er = zero
            _lib.call("amt_finalize", self._handle)
            self._weights_sig = sig
        return self._handle

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                _lib.call("amt_destroy", h)
            except Exception:
                pass

    # ------------------------------------------------------------------------------------------
    # feature plumbing
    # ------------------------------------------------------------------------------------------
    def _prep_features(self, sem, key, scene, motion, emotion, B=None):
        dev = self._device()

        def f(t):
            return t.to(device=dev, dtype=torch.float32).contiguous()

        sem, scene, emotion = f(sem), f(scene), f(emotion)
        motion = f(motion)
        if sem.dim() == 2:
            sem = sem.unsqueeze(0)
        B = sem.shape[0] if B is None else B
        S = sem.shape[1]
        if motion.dim() == 2:                       # scalar motion per frame (motion_type 0), :1012-1015
            motion = motion.unsqueeze(-1).contiguous()
        key = f(key).reshape(-1)
        if key.numel() == 1:                        # feature_key.item() broadcast, :991-997
            key = key.expand(B).contiguous()
        assert key.numel() == B, f"feature_key has {key.numel()} entries for {B} clips"
        assert scene.shape == (B, S) and motion.shape[:2] == (B, S) and emotion.shape[:2] == (B, S), "feature shapes disagree"
        return sem, key, scene, motion, emotion, B, S

    def _encode(self, h, sem, scene, motion, emotion, sl, memory_out=None):
        B, S = sem[sl].shape[0], sem.shape[1]
        resid = None
        if self.scene_embed:                        # + scene_embedding(feature_scene_offset.int()) (:1026-1027)
            idx = scene[sl].to(torch.int32).long().reshape(-1)
            resid = self.scene_embedding.weight.detach().to(torch.float32)[idx].contiguous()
        _lib.call("amt_encode_resid", h, B, S, _lib.ptr(sem[sl].contiguous()), sem.shape[2], _lib.ptr(scene[sl].contiguous()),
                  _lib.ptr(motion[sl].contiguous()), motion.shape[2], _lib.ptr(emotion[sl].contiguous()), emotion.shape[2],
                  _lib.ptr(resid), _lib.ptr(memory_out), _lib.stream_ptr())

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def forward(self, x, x_root, x_attr, feature_semantic_list, feature_key, feature_scene_offset,
                feature_motion, feature_emotion, mask=True):
        """Teacher-forced pass, reference ``forward`` (:978-1044).  Returns logits (B,L,159) fp32.

        ``x`` is only used for its shape, like in the reference (:985-987 consume root/attr).  ``mask`` other than True
        drops the subsequent mask of the decoder self-attention (:978-982; no reference caller does).  With the module
        constant IS_SEPERATED set, the pair (y_root (B,L,15), y_attr (B,L,16)) of the Wout_root / Wout_attr heads (:1036-1040).
        """
        dev = self._device()
        B, L = x.shape[0], x.shape[1]
        sem, key, scene, motion, emotion, Bf, S = self._prep_features(feature_semantic_list, feature_key,
                                                                     feature_scene_offset, feature_motion, feature_emotion)
        h = self._ensure_handle(sem.shape[2])
        assert Bf == B, f"{B} chord sequences but {Bf} clips of video features"
        if self.chord_embed:                     # x = chord_embedding_model(x) (:986-987): the ids index the table, the attr slot a zero row
            x_root, x_attr = x, torch.zeros_like(x)
        roots = x_root.to(device=dev, dtype=torch.long).contiguous()
        attrs = x_attr.to(device=dev, dtype=torch.long).contiguous()
        logits = torch.empty(B, L, CHORD_SIZE, device=dev, dtype=torch.float32)
        for b0 in range(0, B, self.max_decode_batch):
            sl = slice(b0, min(B, b0 + self.max_decode_batch))
            nb = sl.stop - sl.start
            self._encode(h, sem, scene, motion, emotion, sl)
            out = logits[sl]
            _lib.call("amt_set_option", h, b"causal_mask", int(mask is True))
            try:
                _lib.call("amt_prefill", h, nb, L, _lib.ptr(roots[sl].contiguous()), _lib.ptr(attrs[sl].contiguous()),
                          _lib.ptr(key[sl].contiguous()), _lib.ptr(out), None, -1, _lib.stream_ptr())
            finally:
                _lib.call("amt_set_option", h, b"causal_mask", 1)
        if IS_SEPERATED:
            return (logits[..., :CHORD_ROOT_SIZE].contiguous(),
                    logits[..., CHORD_ROOT_SIZE:CHORD_ROOT_SIZE + CHORD_ATTR_SIZE].contiguous())
        return logits

    def forward_debug(self, x_root, x_attr, feature_semantic_list, feature_key, feature_scene_offset,
                      feature_motion, feature_emotion, layer_index=0):
        """(logits, encoder memory (B,S,d), output of decoder layer ``layer_index`` (B,L,d)) for parity tests."""
        dev = self._device()
        sem, key, scene, motion, emotion, B, S = self._prep_features(feature_semantic_list, feature_key,
                                                                    feature_scene_offset, feature_motion, feature_emotion)
        h = self._ensure_handle(sem.shape[2])
        assert B <= self.max_decode_batch
        L = x_root.shape[1]
        memory = torch.empty(B, S, self.d_model, device=dev)
        layer = torch.empty(B, L, self.d_model, device=dev)
        logits = torch.empty(B, L, CHORD_SIZE, device=dev)
        self._encode(h, sem, scene, motion, emotion, slice(0, B), memory)
        _lib.call("amt_prefill", h, B, L, _lib.ptr(x_root.to(dev).long().contiguous()), _lib.ptr(x_attr.to(dev).long().contiguous()),
                  _lib.ptr(key), _lib.ptr(logits), _lib.ptr(layer), layer_index, _lib.stream_ptr())
        return logits, memory, layer

    # ------------------------------------------------------------------------------------------
    # generate
    # ------------------------------------------------------------------------------------------
    def generate(self, feature_semantic_list=[], feature_key=None, feature_scene_offset=None, feature_motion=None,
                 feature_emotion=None, primer=None, primer_root=None, primer_attr=None, target_seq_length=300,
                 beam=0, beam_chance=1.0, max_conseq_N=0, max_conseq_chord=2, sampler="categorical"):
        """Reference ``generate`` (:1046-1132) for one clip: returns a LongTensor (1, target_seq_length)."""
        assert (not self.training), "Cannot generate while in training mode"
        print("Generating sequence of max length:", target_seq_length)
        if beam > 1 or (beam == 1 and beam_chance < 1.0):      # (beam, T): the k best ids of the last top-k step ride in rows 1..
            return self._generate_mixed(feature_semantic_list, feature_key, feature_scene_offset, feature_motion, feature_emotion,
                                        primer, primer_root, primer_attr, target_seq_length, beam, beam_chance, max_conseq_N,
                                        max_conseq_chord, sampler, clip=0)
        return self.generate_batch(feature_semantic_list, feature_key, feature_scene_offset, feature_motion,
                                   feature_emotion, primer, primer_root, primer_attr, target_seq_length, beam,
                                   beam_chance, max_conseq_N, max_conseq_chord, sampler)[:1]

    def _generate_mixed(self, feature_semantic_list, feature_key, feature_scene_offset, feature_motion, feature_emotion,
                        primer, primer_root, primer_attr, target_seq_length, beam, beam_chance, max_conseq_N, max_conseq_chord,
                        sampler, clip=0):
        """``generate`` with beam > 1 and / or beam_chance < 1 as the reference code behaves (:1074-1084), one clip, host-driven
        (no reference caller takes this path; generate.py:347-349 asserts beam == 0).  Per step ``random.uniform(0, 1) <=
        beam_chance`` (python's ``random``, as in the reference: ``random.seed`` pins it) picks the branch.  Top-k branch: the
        forward consumes root / attr only, so the batch stays 1; ``gen_seq`` becomes ``beam`` copies of its row 0 with the k best
        ids of softmax(...)[:157] in column cur, and root / attr of that position stay PAD.  Sampling branch: suppressions look
        at row 0, the drawn id goes to every row and feeds back.  Returns LongTensor (beam, T)."""
        import random
        if IS_SEPERATED:
            raise TypeError("softmax(): argument 'input' must be Tensor, not tuple (IS_SEPERATED heads, as in the reference)")
        if self.chord_embed and beam > 1:
            raise RuntimeError("chord_embed with beam > 1 feeds `beam` chord rows against one clip of video features: the "
                               "reference fails in the cross-attention at the second step")
        if sampler not in ("categorical", "multinomial", "argmax"):
            raise ValueError(f"unknown sampler {sampler!r}")
        dev = self._device()
        sem, key, scene, motion, emotion, B, S = self._prep_features(feature_semantic_list, feature_key, feature_scene_offset,
                                                                    feature_motion, feature_emotion)
        h = self._ensure_handle(sem.shape[2])
        T = int(target_seq_length)
        prim = [torch.as_tensor(p).to(device=dev, dtype=torch.long).reshape(-1).contiguous() for p in (primer, primer_root, primer_attr)]
        P = prim[0].numel()
        if self.chord_embed:
            prim[1], prim[2] = prim[0], torch.zeros_like(prim[0])
        sl = slice(clip, clip + 1)
        st = _lib.stream_ptr()
        self._encode(h, sem, scene, motion, emotion, sl)
        _lib.call("amt_generate_begin", h, 1, _lib.ptr(prim[0]), _lib.ptr(prim[1]), _lib.ptr(prim[2]), P, 0,
                  _lib.ptr(key[sl].contiguous()), T, 0, int(max_conseq_N), int(max_conseq_chord), st)
        gen = torch.full((1, T), CHORD_PAD, device=dev, dtype=torch.long)
        gen[0, :P] = prim[0]
        probs = torch.empty(1, CHORD_END, device=dev)
        for cur in range(1, T):
            if cur < P:                        # still inside the primer: the commit keeps the given token
                _lib.call("amt_generate_step_probs", h, _lib.ptr(probs), st)
                _lib.call("amt_generate_commit", h, _lib.ptr(gen[0, cur:cur + 1].contiguous()), st)
                continue
            beam_ran = 2.0 if beam == 0 else random.uniform(0, 1)
            top_k = beam_ran <= beam_chance
            _lib.call("amt_generate_set_branch", h, int(top_k))
            _lib.call("amt_generate_step_probs", h, _lib.ptr(probs), st)
            if top_k:
                top_i = torch.topk(probs.flatten(), beam)[1]
                gen = gen[top_i // CHORD_SIZE, :]
                gen[..., cur] = top_i % CHORD_SIZE
            elif sampler == "argmax":
                gen[:, cur] = probs.argmax(-1)
            else:                              # Categorical(probs).sample(), :1104-1105
                gen[:, cur] = torch.multinomial(probs, 1).reshape(())
            _lib.call("amt_generate_commit", h, _lib.ptr(gen[0, cur:cur + 1].contiguous()), st)
        torch.cuda.synchronize(dev)
        return gen

    def _debug_set_skip(self, mask):
        """bench.py only: leave the self- (1) / cross- (2) attention launches out of the decode step."""
        _lib.call("amt_set_option", self._ensure_handle(), b"profile_skip", int(mask))

    def set_option(self, name, value):
        """A handle option of the library (include/amt_hip.h: amt_set_option), e.g. ("fuse_sampling_head", 0)."""
        _lib.call("amt_set_option", self._ensure_handle(), name.encode(), int(value))

    def generate_profile(self, feature_semantic_list, feature_key, feature_scene_offset, feature_motion, feature_emotion,
                         primer, primer_root, primer_attr, target_seq_length=300, max_conseq_N=0, max_conseq_chord=2):
        """One feedback-greedy generate (<= 32 clips) issued eagerly with HIP events around every
        decode kernel launch (``amt_generate_profile``).  Returns ``(tokens, stats)`` where stats maps
        kernel class -> {"ms", "launches", "bytes"}; ``bytes`` = algorithmic fp32 K/V bytes."""
        assert (not self.training), "Cannot generate while in training mode"
        dev = self._device()
        sem, key, scene, motion, emotion, B, S = self._prep_features(feature_semantic_list, feature_key,
                                                                    feature_scene_offset, feature_motion, feature_emotion)
        h = self._ensure_handle(sem.shape[2])
        assert B <= self.max_decode_batch
        T = int(target_seq_length)
        prim = [torch.as_tensor(p).to(device=dev, dtype=torch.long).contiguous() for p in (primer, primer_root, primer_attr)]
        per_clip = prim[0].dim() == 2
        P = prim[0].shape[-1]
        st = _lib.stream_ptr()
        self._encode(h, sem, scene, motion, emotion, slice(0, B))
        _lib.call("amt_generate_begin", h, B, _lib.ptr(prim[0]), _lib.ptr(prim[1]), _lib.ptr(prim[2]), P, int(per_clip),
                  _lib.ptr(key), T, 0, int(max_conseq_N), int(max_conseq_chord), st)
        ms = (C.c_double * 5)()
        launches = (C.c_int64 * 5)()
        nbytes = (C.c_int64 * 2)()
        _lib.call("amt_generate_profile", h, -1, ms, launches, nbytes, st)
        out = torch.empty(B, T, device=dev, dtype=torch.long)
        _lib.call("amt_generate_end", h, _lib.ptr(out), st)
        names = ("self_attn_decode", "cross_attn_decode", "decode_gemm", "sample", "empty_event_pair")
        stats = {n: {"ms": ms[i], "launches": launches[i], "bytes": nbytes[i] if i < 2 else None} for i, n in enumerate(names)}
        return out, stats

    def generate_batch(self, feature_semantic_list, feature_key, feature_scene_offset, feature_motion, feature_emotion,
                       primer, primer_root, primer_attr, target_seq_length=300, beam=0, beam_chance=1.0,
                       max_conseq_N=0, max_conseq_chord=2, sampler="categorical", return_logits=False, one_pass_top1=True,
                       uniforms=None):
        """Batched generate: features (B,S,·), primer (P,) shared or (B,P) per clip -> LongTensor (B,T).

        beam=1 is the reference's deterministic top-1 branch (oracle G1; generated ids never feed
        back, :1078-1084).  beam=0 is the sampling branch (:1085-1128) with ``sampler="categorical"`` (the reference's
        Categorical draw, done on device inside the captured step graph by inverse CDF from ``uniforms`` (T,B) -- default
        ``torch.rand`` on the model's device, so ``torch.manual_seed`` makes a run repeatable), ``"multinomial"`` (the
        same distribution drawn per step on the host side with ``torch.multinomial``) or ``"argmax"`` (oracle G2).

        In the beam=1 branch the model input of every generated position is the PAD root/attr pair, so all T-1 decisions
        follow from ONE teacher-forced forward over (primer, PAD, PAD, ...): ``one_pass_top1`` (default) does exactly that
        instead of T-1 decode steps (``one_pass_top1=False`` or ``return_logits=True`` keep the step loop).
        """
        assert (not self.training), "Cannot generate while in training mode"
        if IS_SEPERATED:
            raise TypeError("softmax(): argument 'input' must be Tensor, not tuple (IS_SEPERATED heads, as in the reference)")
        if sampler not in ("categorical", "multinomial", "argmax"):
            raise ValueError(f"unknown sampler {sampler!r}")
        dev = self._device()
        sem, key, scene, motion, emotion, B, S = self._prep_features(feature_semantic_list, feature_key,
                                                                    feature_scene_offset, feature_motion, feature_emotion)
        if beam > 1 or (beam == 1 and beam_chance < 1.0):
            # the reference's top-k branch is a one-clip affair (see _generate_mixed); clips run one after the other and row 0 of
            # each (the top-1 row, the one every later step looks at) is returned
            assert not return_logits, "return_logits belongs to the device-side step loop"
            per_clip = torch.as_tensor(primer).dim() == 2
            pick = (lambda p, c: torch.as_tensor(p)[c]) if per_clip else (lambda p, c: p)
            return torch.cat([self._generate_mixed(sem, key, scene, motion, emotion, pick(primer, c), pick(primer_root, c),
                                                   pick(primer_attr, c), target_seq_length, beam, beam_chance, max_conseq_N,
                                                   max_conseq_chord, sampler, clip=c)[:1] for c in range(B)], dim=0)
        h = self._ensure_handle(sem.shape[2])
        T = int(target_seq_length)
        prim = [torch.as_tensor(p).to(device=dev, dtype=torch.long).contiguous() for p in (primer, primer_root, primer_attr)]
        per_clip = prim[0].dim() == 2
        P = prim[0].shape[-1]
        assert all(p.shape == prim[0].shape for p in prim), "primer / primer_root / primer_attr shapes differ"
        assert (not per_clip) or prim[0].shape[0] == B
        if self.chord_embed:                     # the ids are the model input (:986-987), in both branches: no one-pass shortcut
            prim[1], prim[2] = prim[0], torch.zeros_like(prim[0])
            one_pass_top1 = False
        tokens = torch.empty(B, T, device=dev, dtype=torch.long)
        if beam == 1 and one_pass_top1 and not return_logits and T > P:
            roots = torch.full((B, T - 1), CHORD_ROOT_PAD, device=dev, dtype=torch.long)
            attrs = torch.full((B, T - 1), CHORD_ATTR_PAD, device=dev, dtype=torch.long)
            roots[:, :P], attrs[:, :P] = prim[1], prim[2]
            lg = torch.empty(B, T - 1, CHORD_SIZE, device=dev, dtype=torch.float32)
            for b0 in range(0, B, self.max_decode_batch):
                sl = slice(b0, min(B, b0 + self.max_decode_batch))
                self._encode(h, sem, scene, motion, emotion, sl)
                _lib.call("amt_prefill", h, sl.stop - sl.start, T - 1, _lib.ptr(roots[sl].contiguous()), _lib.ptr(attrs[sl].contiguous()),
                          _lib.ptr(key[sl].contiguous()), _lib.ptr(lg[sl]), None, -1, _lib.stream_ptr())
            tokens[:, :P] = prim[0]
            tokens[:, P:] = lg[:, P - 1:, :CHORD_END].argmax(dim=-1)          # top-1 of softmax(...)[:157] (:1070-1084)
            return tokens
        logits = torch.zeros(T, B, CHORD_SIZE, device=dev) if return_logits else None
        for b0 in range(0, B, self.max_decode_batch):
            sl = slice(b0, min(B, b0 + self.max_decode_batch))
            nb = sl.stop - sl.start
            self._encode(h, sem, scene, motion, emotion, sl)
            pr = [p[sl].contiguous() if per_clip else p for p in prim]
            lg = torch.zeros(T, nb, CHORD_SIZE, device=dev) if return_logits else None
            out = torch.empty(nb, T, device=dev, dtype=torch.long)
            st = _lib.stream_ptr()
            _lib.call("amt_generate_begin", h, nb, _lib.ptr(pr[0]), _lib.ptr(pr[1]), _lib.ptr(pr[2]), P, int(per_clip),
                      _lib.ptr(key[sl].contiguous()), T, beam, int(max_conseq_N), int(max_conseq_chord), st)
            if beam == 1 or sampler in ("argmax", "categorical"):
                if beam == 0 and sampler == "categorical":
                    u = (torch.rand(T, nb, device=dev) if uniforms is None
                         else torch.as_tensor(uniforms, dtype=torch.float32, device=dev)[:, sl].contiguous())
                    assert u.shape == (T, nb), "uniforms must be (T, B)"
                    _lib.call("amt_generate_set_uniforms", h, _lib.ptr(u), st)
                _lib.call("amt_generate_run", h, -1, _lib.ptr(lg), st)
            else:
                probs = torch.empty(nb, CHORD_END, device=dev)
                for cur in range(1, T):
                    _lib.call("amt_generate_step_probs", h, _lib.ptr(probs), st)
                    if cur >= P:       # Categorical(probs).sample(), :1104-1105
                        chosen = torch.multinomial(probs, 1).reshape(nb).contiguous()
                    else:              # still inside the primer: the commit keeps the given token
                        chosen = torch.zeros(nb, device=dev, dtype=torch.long)
                    _lib.call("amt_generate_commit", h, _lib.ptr(chosen), st)
            _lib.call("amt_generate_end", h, _lib.ptr(out), st)
            tokens[sl] = out
            if return_logits:
                logits[:, sl] = lg
        return (tokens, logits) if return_logits else tokens


# The V2 / V1 / V3 families live in their own files and are importable from here like in the reference, whose four classes share
# this module (V1 and V3 subclass V2).
from .vmt_v2 import VideoMusicTransformer_V2, _DecoderLayerV2, _TransformerParamsV2  # noqa: E402,F401
from .vmt_v1 import VideoMusicTransformer_V1  # noqa: E402,F401
from .vmt_v3 import VideoMusicTransformer_V3, _DiffAttnParams  # noqa: E402,F401
