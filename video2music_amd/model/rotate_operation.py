"""``RotaryPositionalEmbeddings`` on the HIP path (reference ``model/rotate_operation.py:50-165``).

The cos/sin cache is built on the host with the reference's formula (:88-109); the rotation of
interleaved pairs (:153-161) runs in ``amt_rope_fwd``.  The reference's call sites pass a
``(n_heads, L, B, head_dim)`` view together with a cache built for ``dim = d_model``
(``custom_transformer.py:1044-1053``): the resulting frequency-band folding (``.view(-1, L, 1,
hd/2, 2)[:n_heads]``, :148-149) is reproduced by the kernel's cache indexing.
"""
import torch
import torch.nn as nn

from .. import _lib


class RotaryPositionalEmbeddings(nn.Module):
    def __init__(self, dim, max_seq_len=4096, base=10_000):
        super().__init__()
        self.dim, self.base, self.max_seq_len = dim, base, max_seq_len
        self._rope_init()

    def reset_parameters(self):
        self._rope_init()

    def _rope_init(self):
        theta = 1.0 / (self.base ** (torch.arange(0, self.dim, 2)[: (self.dim // 2)].float() / self.dim))
        self.register_buffer("theta", theta, persistent=False)
        self.build_rope_cache(self.max_seq_len)

    def build_rope_cache(self, max_seq_len=4096):
        seq_idx = torch.arange(max_seq_len, dtype=self.theta.dtype, device=self.theta.device)
        idx_theta = torch.einsum("i, j -> ij", seq_idx, self.theta).float()
        cache = torch.stack([torch.cos(idx_theta), torch.sin(idx_theta)], dim=-1)
        self.register_buffer("cache", cache, persistent=False)

    def forward(self, x, *, input_pos=None):
        if input_pos is not None:
            raise NotImplementedError("input_pos is not used by the reference's call sites")
        if x.device.type != "cuda":
            raise _lib.AmtError("RotaryPositionalEmbeddings runs on an MI355X only; video2music_amd has no CPU fallback")
        n0, seq, n2, hd = x.shape
        if seq > self.max_seq_len:
            raise ValueError(f"sequence length {seq} exceeds the rope cache ({self.max_seq_len})")
        xf = x.to(torch.float32).contiguous()
        y = torch.empty_like(xf)
        cache = self.cache[:seq].contiguous()
        _lib.call("amt_rope_fwd", _lib.ptr(xf), _lib.ptr(cache), _lib.ptr(y), n0, seq, n2, hd, self.dim // 2, _lib.stream_ptr())
        return y.to(x.dtype)
