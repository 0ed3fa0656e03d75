"""Thin Python wrappers over the stateless C-ABI operator entry points (``include/amt_hip.h``).

Used by the host modules that are compositions of kernels rather than handle-based fast paths
(``VideoMusicTransformer_V2``).  Tensors are contiguous fp32 CUDA tensors; every call raises ``AmtError`` on a
non-zero status.  No torch arithmetic happens here.
"""
import ctypes as C

import torch

from . import _lib

p = _lib.ptr


def _st():
    return _lib.stream_ptr()


def linear(x, w, b=None, resid=None, relu=False):
    """y[M,N] = x[M,K] w[N,K]^T + b (+ resid) (ReLU).  K must be a multiple of 32."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _lib.call("amt_linear_fwd", p(x), p(w), p(b), p(resid), p(y), M, N, K, int(relu), _st())
    return y


def layernorm(x, w, b, resid=None, eps=1e-5):
    rows, dim = x.shape
    y = torch.empty_like(x)
    _lib.call("amt_layernorm_fwd", p(x), p(resid), p(w), p(b), p(y), rows, dim, float(eps), _st())
    return y


def rope(x, cache):
    """x (n0, seq, n2, hd) contiguous, cache (>=seq, cache_half, 2): RotaryPositionalEmbeddings.forward."""
    n0, seq, n2, hd = x.shape
    y = torch.empty_like(x)
    c = cache[:seq].contiguous()
    _lib.call("amt_rope_fwd", p(x), p(c), p(y), n0, seq, n2, hd, cache.shape[1], _st())
    return y


def attention(q, k, v, strides, B, H, Lq, Lk, hd, causal, q_scale, out):
    s = (C.c_int64 * 12)(*strides)
    _lib.call("amt_attn_fwd", p(q), p(k), p(v), p(out), s, B, H, Lq, Lk, hd, int(causal), 1, float(q_scale), _st())
    return out


def glu(x, e):
    """GLUExpert.forward on rows x (n, d): W2((W1 x + b1) * silu(Wg x + bg)) + b2."""
    n, d = x.shape
    dff = e.linear1.out_features
    out = torch.empty(n, d, device=x.device, dtype=torch.float32)
    scratch = torch.empty(2 * n * dff, device=x.device, dtype=torch.float32)
    t = [t_.detach().contiguous() for t_ in (e.linear1.weight, e.linear1.bias, e.gate.weight, e.gate.bias, e.linear2.weight, e.linear2.bias)]
    _lib.call("amt_glu_expert_fwd", p(x), *[p(v) for v in t], p(out), p(scratch), n, d, dff, _st())
    return out


def concat_features(sem, scene, motion, emotion, ld_out):
    B, S, sd_ = sem.shape
    out = torch.empty(B * S, ld_out, device=sem.device, dtype=torch.float32)
    _lib.call("amt_concat_features_fwd", p(sem), sd_, p(scene), p(motion), motion.shape[2], p(emotion), emotion.shape[2], p(out), B * S, ld_out, _st())
    return out


def chord_embed(roots, attrs, key, PR, PA, wkey, bias, pe):
    B, L = roots.shape
    d = PR.shape[1]
    out = torch.empty(B * L, d, device=PR.device, dtype=torch.float32)
    _lib.call("amt_chord_embed_fwd", p(roots), p(attrs), p(key), p(PR), p(PA), p(wkey), p(bias), p(pe), p(out), B, L, d, _st())
    return out
