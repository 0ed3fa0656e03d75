"""``MultiheadGQA`` on the HIP path (reference ``model/grouped_query_attention.py:172-358``).

Same constructor, parameter names (q_proj/k_proj/v_proj/norm/out_proj) and ``forward`` signature.
Inputs are the caller's ``(L, B, E)`` buffers; like the reference, the projected q/k/v memory is
reinterpreted as ``(B, L, ·)`` (:316-326), which mixes clips when B > 1 — reproduced exactly.
``attn_mask`` is accepted and ignored like in the reference (:339).  ``RoPE=`` (a ``RotaryPositionalEmbeddings``, deep-copied like
in the reference, :216) rotates the projected q / k through the raw ``(heads, len, B, head_dim)`` view (:316-322) inside
``amt_gqa_rope_fwd``.  ``need_weights`` and ``key_padding_mask`` are not on the hot path and raise; ``dropout > 0`` raises too:
the reference applies ``F.dropout`` to the attention map even in eval (:152-153), i.e. a random result.
"""
import copy

import torch
import torch.nn as nn

from .. import _lib


class MultiheadGQA(nn.Module):
    def __init__(self, embed_dim, query_heads, kv_heads, dropout=0.0, bias=True, layer_norm=True,
                 layer_norm_eps=1e-5, gamma_init=1.0, device=None, dtype=None, RoPE=None):
        super().__init__()
        if query_heads % kv_heads != 0:
            raise ValueError(f"query_heads ({query_heads}) must be divisible by kv_heads ({kv_heads})")
        if embed_dim % query_heads != 0 or embed_dim % kv_heads != 0:
            raise ValueError(f"embed_dim ({embed_dim}) must be divisible by query_heads and kv_heads")
        head_dim = embed_dim // query_heads
        if head_dim % 8 != 0 or head_dim > 128:
            raise ValueError(f"head_dim {head_dim} must be divisible by 8 and <= 128")
        if dropout > 0.0:
            raise NotImplementedError("attention dropout inside MultiheadGQA (random even in eval, :152-153) is outside the hot path")
        self.query_heads, self.kv_heads, self.dropout = query_heads, kv_heads, dropout
        self.layer_norm, self.gamma_init, self.embed_dim = layer_norm, gamma_init, embed_dim
        self.RoPE = copy.deepcopy(RoPE)         # :216
        kv_embed_dim = head_dim * kv_heads
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias, device=device, dtype=dtype)
        self.k_proj = nn.Linear(embed_dim, kv_embed_dim, bias=bias, device=device, dtype=dtype)
        self.v_proj = nn.Linear(embed_dim, kv_embed_dim, bias=bias, device=device, dtype=dtype)
        self.norm = nn.LayerNorm(embed_dim, eps=layer_norm_eps, device=device, dtype=dtype) if layer_norm else None
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias, device=device, dtype=dtype)
        self._reset_parameters()

    def _reset_parameters(self):            # :265-284
        nn.init.xavier_normal_(self.q_proj.weight)
        nn.init.xavier_normal_(self.k_proj.weight)
        nn.init.xavier_normal_(self.v_proj.weight, gain=self.gamma_init)
        nn.init.xavier_normal_(self.out_proj.weight, gain=self.gamma_init)
        for lin in (self.q_proj, self.k_proj, self.v_proj, self.out_proj):
            if lin.bias is not None:
                nn.init.constant_(lin.bias, 0)

    def forward(self, query, key, value, need_weights=False, attn_mask=None, key_padding_mask=None,
                is_causal=False, average_attn_weights=False):
        if need_weights or key_padding_mask is not None:
            raise NotImplementedError("need_weights / key_padding_mask are outside the hot path")
        if query.device.type != "cuda":
            raise _lib.AmtError("MultiheadGQA runs on an MI355X only; video2music_amd has no CPU fallback")
        hd = self.embed_dim // self.query_heads
        if hd not in (16, 32, 64, 128):
            raise NotImplementedError(f"head_dim {hd}: the gfx950 attention kernel is built for 16, 32, 64 and 128")
        L, B, E = query.shape
        S = key.shape[0]
        q, k, v = (t.to(torch.float32).contiguous() for t in (query, key, value))
        out = torch.empty(L, B, E, device=q.device, dtype=torch.float32)
        ekv = hd * self.kv_heads
        scratch = torch.empty(2 * L * B * E + 2 * S * B * ekv, device=q.device, dtype=torch.float32)
        p = _lib.ptr

        def w(lin):
            return p(lin.weight.detach().contiguous()), p(lin.bias.detach().contiguous()) if lin.bias is not None else None

        (wq, bq), (wk, bk), (wv, bv), (wo, bo) = w(self.q_proj), w(self.k_proj), w(self.v_proj), w(self.out_proj)
        lnw = p(self.norm.weight.detach()) if self.norm is not None else None
        lnb = p(self.norm.bias.detach()) if self.norm is not None else None
        eps = float(self.norm.eps) if self.norm is not None else 1e-5
        if self.RoPE is None:
            _lib.call("amt_gqa_fwd", p(q), p(k), p(v), wq, bq, wk, bk, wv, bv, lnw, lnb, wo, bo, p(out), p(scratch),
                      L, S, B, E, self.query_heads, self.kv_heads, int(bool(is_causal)), eps, _lib.stream_ptr())
        else:
            cache = self.RoPE.cache
            if cache.device != q.device or not cache.is_contiguous():
                cache = cache.to(q.device).contiguous()
            if max(L, S) > cache.shape[0]:
                raise ValueError(f"sequence length {max(L, S)} exceeds the rope cache ({cache.shape[0]})")
            _lib.call("amt_gqa_rope_fwd", p(q), p(k), p(v), wq, bq, wk, bk, wv, bv, lnw, lnb, wo, bo, p(out), p(scratch),
                      L, S, B, E, self.query_heads, self.kv_heads, int(bool(is_causal)), eps, p(cache), cache.shape[0], cache.shape[1],
                      _lib.stream_ptr())
        return out, None
