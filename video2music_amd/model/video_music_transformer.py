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
        """`memory`: one clip's encoder output (S, E), or a list of them for the lockstep step of several clips
        (`generate_batch`: every cache then carries a leading clip dimension)."""
        from .. import ops
        E, H = self.d_model, self.nhead
        hd = E // H
        mems = list(memory) if isinstance(memory, (list, tuple)) else [memory]
        nb = len(mems)
        dev = mems[0].device
        st = {"cross": [], "self": [], "S": S, "B": nb}
        rows = mems[0] if nb == 1 else torch.cat(mems)                  # (nb*S, E): one projection launch for all clips
        cross_all = []
        for lyr in self.transformer.decoder.layers:
            a = lyr.cross_attn
            W, b = a.in_proj_weight.detach(), a.in_proj_bias.detach()
            k = ops.linear(rows, W[E:2 * E], b[E:2 * E])
            if self._rope_cache is not None and self._rope_cache.shape[1] * 2 == E:
                # per clip the B = 1 view, positions 0..S-1 = pair i of the E-wide row at position s rotated by cache[s][i]
                # (`_attention` does the same for the encoder): one launch for all clips
                ops.rope(k.view(nb, S, 1, E), self._rope_cache, out=k.view(nb, S, 1, E))
            elif self._rope_cache is not None:                         # a cache built for another width: clip by clip
                for c in range(nb):
                    kc_ = k[c * S:(c + 1) * S]
                    ops.rope(kc_.view(H, S, 1, hd), self._rope_cache, out=kc_.view(H, S, 1, hd))
            v = ops.linear(rows, W[2 * E:], b[2 * E:])
            cross_all.append((k, v))
            st["cross"].append((k[:S], v[:S]))
            st["self"].append((torch.empty(self._max_dec, E, device=dev), torch.empty(self._max_dec, E, device=dev)))
        # pointer table of amt_v2_step (include/amt_hip.h); `keep` holds every tensor the table points into
        keep, ptrs = [], []

        def add(t):
            if t is None:
                ptrs.append(None)
            else:
                t = t.detach().contiguous()
                keep.append(t)
                ptrs.append(t.data_ptr())

        cache = self.__dict__.setdefault("_pack_cache", {})

        def cached(ident, version, build):
            """One entry per identity (storage pointers + role); a new parameter version (load_state_dict, an optimiser step
            followed by eval) REPLACES the entry, so stale packed copies do not stay resident."""
            hit = cache.get(ident)
            if hit is None or hit[0] != version:
                cache[ident] = hit = (version, build())
            return hit[1]

        def pack_now(w, rows=None):
            src = (w if rows is None else w[:rows]).detach().contiguous()
            N, K = src.shape
            out = torch.empty((N + 15) // 16 * 16 * K, device=dev, dtype=torch.float32)
            _lib.call("amt_pack_weight_fwd", _lib.ptr(src), _lib.ptr(out), N, K, _lib.stream_ptr())
            return out

        def packed(w, rows=None):
            """Weight (N, K) [or its first `rows` rows] in the skinny GEMM's tile order; packed once per parameter version."""
            w = w.detach()
            return cached((w.data_ptr(), rows), w._version, lambda: pack_now(w, rows))

        def packed_experts(experts, name):
            # (the per-expert packed pieces are temporaries: only the concatenation stays resident)
            lins = [expert_parts(e)[name] for e in experts]
            return cached(tuple(l.weight.data_ptr() for l in lins) + (name,), tuple(l.weight._version for l in lins),
                          lambda: torch.cat([pack_now(l.weight) for l in lins]))

        def nbias(n):       # a norm's bias; None (RMSNorm) selects the RMS form inside the step
            return getattr(n, "bias", None)

        for t in (self._PR, self._PA, self._wkey, self.Linear_chord.bias, self._rope_cache, self.transformer.decoder.norm.weight,
                  nbias(self.transformer.decoder.norm), packed(self.Wout.weight), self.Wout.bias,
                  torch.tensor([0, 1], device=dev, dtype=torch.int32), self._pe_chord if self._learned_pos else None):
            add(t)
        from .moe import GLUExpert, SiLUExpert, _stack, expert_dff, expert_parts
        dff, widths = None, set()

        def add_expert(e):
            """linear1 w, b (None for a SiLUExpert), gate w, b, linear2 w, b -- packed weights."""
            q = expert_parts(e)
            for name in ("linear1", "gate", "linear2"):
                add(None if q[name] is None else packed(q[name].weight))
                add(None if q[name] is None else q[name].bias)

        # the one-call step streams K/V with the decode-attention kernel: head-major caches (H, rows, hd)
        st["self_hm"] = [(torch.empty(nb, H, self._max_dec, hd, device=dev), torch.empty(nb, H, self._max_dec, hd, device=dev))
                         for _ in st["self"]]
        st["cross_hm"] = [(k.view(nb, S, H, hd).permute(0, 2, 1, 3).contiguous(), v.view(nb, S, H, hd).permute(0, 2, 1, 3).contiguous())
                          for k, v in cross_all]
        layers = list(self.transformer.decoder.layers)
        for li, (lyr, (kc, vc), (kx, vx)) in enumerate(zip(layers, st["self_hm"], st["cross_hm"])):
            sa, ca = lyr.self_attn, lyr.cross_attn
            for t in (packed(sa.in_proj_weight), sa.in_proj_bias, packed(sa.out_proj.weight), sa.out_proj.bias, lyr.norm1.weight, nbias(lyr.norm1),
                      packed(ca.in_proj_weight, rows=E), ca.in_proj_bias, packed(ca.out_proj.weight), ca.out_proj.bias, lyr.norm2.weight,
                      nbias(lyr.norm2), lyr.norm3.weight, nbias(lyr.norm3), kc, vc, kx, vx):
                add(t)
            ff = lyr.ff

            def stacked(mods, with_down):
                """[gate of every module | linear1 of every module] as ONE packed matrix + bias (the lockstep step's single
                gate/up product), and, for a mixture layer, linear2 of every module one after the other + biases."""
                parts = [expert_parts(e) for e in mods]
                srcs = [q[n] for n in ("gate", "linear1", "linear2") for q in parts if q[n] is not None]

                def build():
                    gu = [q["gate"] for q in parts] + [q["linear1"] for q in parts if q["linear1"] is not None]
                    out = [torch.cat([pack_now(l.weight) for l in gu]), torch.cat([l.bias.detach() for l in gu]).contiguous()]
                    if with_down:
                        out += [torch.cat([pack_now(q["linear2"].weight) for q in parts]),
                                torch.cat([q["linear2"].bias.detach() for q in parts]).contiguous()]
                    else:
                        out += [None, None]
                    return out

                return cached(tuple((l.weight.data_ptr(), l.bias.data_ptr()) for l in srcs) + ("stacked", with_down),
                              tuple((l.weight._version, l.bias._version) for l in srcs), build)

            if isinstance(ff, (GLUExpert, SiLUExpert)):
                layer_dff = expert_dff(ff)
                add(None), add(None)
                add_expert(ff)
                for _ in range(6):
                    add(None)
                for t in stacked([ff], False):
                    add(t)
            else:
                if ff.n_experts_per_token != 2 or getattr(ff, "expert_parallel", False):
                    raise NotImplementedError("the cached V2 step is built for local top-2 MoE layers")
                if hasattr(ff, "temperature_scheduler"):
                    # the scheduler steps in every forward (moe.py:238-240) and rescales the routing logits; the cached step hands
                    # the raw gate to the device, so such a layer must take the per-step re-forward
                    raise NotImplementedError("the cached V2 step does not evaluate a SharedMoELayer temperature_scheduler")
                layer_dff = expert_dff(ff.experts[0])
                add(ff.gate.weight), add(ff.gate.bias)
                for name in ("linear1", "gate", "linear2"):
                    if expert_parts(ff.experts[0])[name] is None:
                        add(None), add(None)
                    else:
                        add(packed_experts(ff.experts, name))
                        add(_stack(ff.experts, name, "bias"))
                if ff.shared:
                    add_expert(ff.shared_expert)
                else:
                    for _ in range(6):
                        add(None)
                for t in stacked(list(ff.experts) + ([ff.shared_expert] if ff.shared else []), True):
                    add(t)
            # out-projection of the self-attention and the cross-attention's query projection in ONE launch of the lockstep step,
            # norm1 folded through the projection (the base model's G1): with u = x + o Wo^T + bo the query is
            # LayerNorm(u) Wq^T + bq = ((u (Wq o gamma)^T) - mean g) rstd + c, and u (Wq o gamma)^T = [o | x] [Wq' Wo | Wq']^T + Wq' bo.
            # The launch produces u and that raw product; the attention kernel finishes the query with u's row statistics.
            if isinstance(lyr.norm1, nn.LayerNorm) and 2 * E <= 1536 and os.environ.get("AMT_V2_FOLD_G1", "1") != "0":
                from .. import ops
                srcs = (sa.out_proj.weight, sa.out_proj.bias, ca.in_proj_weight, ca.in_proj_bias, lyr.norm1.weight, lyr.norm1.bias)

                def build_fold(sa=sa, ca=ca, lyr=lyr):
                    Wo, bo, Wq, bq = sa.out_proj.weight.detach(), sa.out_proj.bias.detach(), ca.in_proj_weight.detach()[:E], ca.in_proj_bias.detach()[:E]
                    gamma, beta = lyr.norm1.weight.detach(), lyr.norm1.bias.detach()
                    Wqg = (Wq * gamma.unsqueeze(0)).contiguous()                              # Wq o gamma
                    A = ops.linear(Wqg, Wo.t().contiguous())                                   # (Wq o gamma) Wo
                    P2 = torch.cat([A, Wqg], dim=1).contiguous()                               # (E, 2E), a temporary
                    return [pack_now(P2), ops.linear(bo.view(1, E).contiguous(), Wqg).view(E).contiguous(),
                            Wqg.sum(dim=1).contiguous(), (ops.linear(beta.view(1, E).contiguous(), Wq.contiguous()).view(E) + bq).contiguous()]

                for t in cached(tuple(t.data_ptr() for t in srcs) + ("g1fold",), tuple(t._version for t in srcs), build_fold):
                    add(t)
            else:
                for _ in range(4):
                    add(None)
            # A plain GLU layer with norm3 folded through the NEXT layer's QKV projection (the base model's G3, DESIGN.md section 5: the
            # down projection also emits the raw QKV product, the next self-attention finishes it: 6 launches per layer instead of 7)
            # and, with AMT_V2_FOLD_FFN=2, norm2 through its stacked gate | up product as well (the base model's G2: 5 launches).
            #   LayerNorm(u2) Wgu^T + bgu = ((u2 (Wgu o gamma2)^T) - mean g) rstd + c,  u2 (Wgu o gamma2)^T = [o | x] [W' Wo | W']^T + W' bo
            #   LayerNorm(u3) Wn^T + bn likewise with u3 = h W2^T + b2 + LayerNorm(u2): [h | xn2] [W'' W2 | W'']^T + W'' b2
            # Widths: whole 256-column chunks on either side of the down projection's staged row [h | xn2] (K = dff + E <= 1536).
            parts = expert_parts(ff) if isinstance(ff, GLUExpert) else None
            fold_mode = int(os.environ.get("AMT_V2_FOLD_FFN", "1"))       # 1 (default): norm3 -> next QKV only; 2: norm2 -> gate | up as well; 0: off
            nxt = layers[li + 1] if li + 1 < len(layers) else None
            fold_ffn = (fold_mode > 0 and parts is not None and parts["linear1"] is not None
                        and all(isinstance(n, nn.LayerNorm) for n in (lyr.norm2, lyr.norm3))
                        and E % 256 == 0 and layer_dff % 256 == 0 and 2 * E <= 1536 and layer_dff + E in (512, 768, 1024, 1536)
                        and (layer_dff + E < 1536 or E >= 512))
            if fold_ffn and fold_mode < 2 and nxt is None:
                fold_ffn = False
            if fold_ffn:
                srcs = [ca.out_proj.weight, ca.out_proj.bias, parts["gate"].weight, parts["gate"].bias, parts["linear1"].weight, parts["linear1"].bias,
                        parts["linear2"].weight, parts["linear2"].bias, lyr.norm2.weight, lyr.norm2.bias, lyr.norm3.weight, lyr.norm3.bias]
                if nxt is not None:
                    srcs += [nxt.self_attn.in_proj_weight, nxt.self_attn.in_proj_bias]

                def build_ffn_fold(ca=ca, lyr=lyr, parts=parts, nxt=nxt, mode=fold_mode):
                    row = lambda v: v.detach().view(1, -1).contiguous()
                    out = [None] * 4
                    if mode >= 2:         # measured slower than the separate launches at d_model 512 (profiles/r03_v2_fold_ab.json): kept for A/B
                        Wo, bo = ca.out_proj.weight.detach(), ca.out_proj.bias.detach()
                        Wgu = torch.cat([parts["gate"].weight.detach(), parts["linear1"].weight.detach()]).contiguous()     # (2 dff, E): gate rows first
                        bgu = torch.cat([parts["gate"].bias.detach(), parts["linear1"].bias.detach()])
                        Wp = (Wgu * lyr.norm2.weight.detach().unsqueeze(0)).contiguous()                                   # Wgu o gamma2
                        P2 = torch.cat([ops.linear(Wp, Wo.t().contiguous()), Wp], dim=1).contiguous()                     # (2 dff, 2E), a temporary
                        out = [pack_now(P2), ops.linear(row(bo), Wp).view(-1).contiguous(), Wp.sum(dim=1).contiguous(),
                               (ops.linear(row(lyr.norm2.bias), Wgu).view(-1) + bgu).contiguous()]
                    if nxt is None:
                        return out + [None] * 4
                    Wn, bn = nxt.self_attn.in_proj_weight.detach(), nxt.self_attn.in_proj_bias.detach()
                    W2, b2 = parts["linear2"].weight.detach(), parts["linear2"].bias.detach()                         # (E, dff)
                    Wq = (Wn * lyr.norm3.weight.detach().unsqueeze(0)).contiguous()                                    # Wqkv' o gamma3
                    P3 = torch.cat([ops.linear(Wq, W2.t().contiguous()), Wq], dim=1).contiguous()                      # (3E, dff + E)
                    return out + [pack_now(P3), ops.linear(row(b2), Wq).view(-1).contiguous(), Wq.sum(dim=1).contiguous(),
                                  (ops.linear(row(lyr.norm3.bias), Wn.contiguous()).view(-1) + bn).contiguous()]

                for t in cached(tuple(t.data_ptr() for t in srcs) + ("ffnfold", fold_mode), tuple(t._version for t in srcs), build_ffn_fold):
                    add(t)
            else:
                for _ in range(8):
                    add(None)
            widths.add(layer_dff)
            dff = layer_dff
        st["tab"] = (C.c_void_p * len(ptrs))(*ptrs)
        st["keep"] = keep
        st["dff"] = dff
        # amt_v2_step lays its scratch out for one feed-forward width; layers of different widths (V1 '1.3.3' / '1.3.4' with
        # dim_feedforward != 2 d_model) take the same cached step issued operator by operator (`_decode_step`)
        # the step kernels take widths that are multiples of 64 up to 1536 (amt_v2_step / amt_v2_step_batch); other widths take the
        # operator path too
        st["native"] = len(widths) == 1 and E % 64 == 0 and dff % 64 == 0 and E <= 1536 and dff <= 1536
        # (one clip may run either step: the one-call step with device-routed experts or the lockstep step with B = 1)
        n_ws = max(_lib.call("amt_v2_step_ws_floats", E, dff, self.n_experts) if nb == 1 else 0,
                   _lib.call("amt_v2_step_batch_ws_floats", E, dff, self.n_experts, nb))
        st["ws"] = torch.empty(n_ws, device=dev, dtype=torch.float32)
        st["logits"] = torch.empty(CHORD_SIZE, device=dev, dtype=torch.float32) if nb == 1 else torch.empty(nb, CHORD_SIZE, device=dev)
        return st

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


# ==================================================================================================
# VideoMusicTransformer_V1 (second widening): the same machinery with another layer plan
# ==================================================================================================
class VideoMusicTransformer_V1(VideoMusicTransformer_V2):
    """Reference ``VideoMusicTransformer_V1`` (model/video_music_transformer.py:22-314), eval mode.  Learned positional
    tables on both streams (:63-65,198-206); every layer's feed-forward a 6-expert top-2 mixture -- ``MoELayer`` for
    '1.0', '1.1', '1.3.4', else ``SharedMoELayer`` -- over ``GLUExpert(d, d_ff)`` ('1.1', '1.3') or
    ``Linear(d, 2d) -> SiLU -> Linear(2d, d)`` experts (:77-85); '1.3.3' / '1.3.4' put three plain GLU layers first
    (:108-125); RoPE inside the attentions when ``version_name in '1.2.3'`` -- the reference's substring test (:86), so
    '1.2' gets it too; ``rms_norm=True`` swaps every LayerNorm for RMSNorm (:69-72).  ``nn.MultiheadAttention`` and
    ``CustomMultiheadAttention`` carry the same parameter names and compute the same attention without RoPE, so one
    code path serves both.  forward / generate / the KV-cached decode step are inherited from the V2 class.
    """

    def __init__(self, version_name="1.1", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, dropout=0.1,
                 max_sequence_midi=2048, max_sequence_video=300, max_sequence_chord=300, total_vf_dim=0, rms_norm=False,
                 scene_embed=False, chord_embed=False, dropTokenRate=0.0):
        nn.Module.__init__(self)
        from .custom_transformer import RMSNorm
        from .moe import GLUExpert, MoELayer, SharedMoELayer, SiLUExpert
        shallow = version_name in ("1.3.3", "1.3.4")
        self.nlayers, self.nhead, self.d_model, self.d_ff, self.dropout = n_layers, num_heads, d_model, dim_feedforward, dropout
        self.max_seq_midi, self.max_seq_video, self.max_seq_chord = max_sequence_midi, max_sequence_video, max_sequence_chord
        self.scene_embed, self.chord_embed, self.dropTokenRate, self.version_name = scene_embed, chord_embed, dropTokenRate, version_name
        self.total_vf_dim = total_vf_dim
        self.n_experts, self.n_experts_per_token = 6, 2
        self._learned_pos = True
        self._use_rope = version_name in "1.2.3"                     # substring test, as written at :86
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
        self.positional_embedding = nn.Embedding(max_sequence_chord, d_model)
        self.positional_embedding_video = nn.Embedding(max_sequence_video, d_model)
        self.condition_linear = nn.Linear(1, d_model)

        def expert():
            if version_name in ("1.1", "1.3"):
                return GLUExpert(d_model, dim_feedforward, dropout)
            return SiLUExpert(d_model, 2 * d_model, dropout)

        def ff(i):
            if shallow and i < 3:
                return GLUExpert(d_model, dim_feedforward, dropout)
            if version_name in ("1.0", "1.1", "1.3.4"):
                return MoELayer(expert(), d_model, self.n_experts, self.n_experts_per_token, dropout)
            return SharedMoELayer(expert(), d_model, n_experts=self.n_experts, n_experts_per_token=self.n_experts_per_token,
                                  balancing=False, dropout=dropout)

        self.transformer = _TransformerParamsV2(d_model, num_heads, max(3, n_layers) if shallow else n_layers, ff,      # (:114-119)
                                                norm=RMSNorm if rms_norm else nn.LayerNorm)
        self.Wout = nn.Linear(d_model, CHORD_SIZE)
        self.softmax = nn.Softmax(dim=-1)
        if self._use_rope:
            from .rotate_operation import RotaryPositionalEmbeddings
            rope = RotaryPositionalEmbeddings(d_model, max_sequence_video)
            self.register_buffer("_rope_cache", rope.cache.clone(), persistent=False)
        else:
            self._rope_cache = None
        # the RoPE cache caps the chord sequence at max_sequence_video, the positional table at max_sequence_chord
        self._max_dec = min(max_sequence_video, max_sequence_chord) if self._use_rope else max_sequence_chord
        self._derived_sig = None


# ==================================================================================================
# VideoMusicTransformer_V3 (last widening): differential attention, not KV-cacheable
# ==================================================================================================
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
