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


def diff_subln(o1, o2, w, lam, out_scale, eps=1e-5):
    """RMSNorm_hd(o1 - lam * o2) * w * out_scale; o1 / o2 (..., hd) contiguous."""
    hd = o1.shape[-1]
    y = torch.empty_like(o1)
    _lib.call("amt_diff_subln_fwd", p(o1), p(o2), p(w), p(y), o1.numel() // hd, hd, float(lam), float(out_scale), float(eps), _st())
    return y


def add(a, b):
    y = torch.empty_like(a)
    _lib.call("amt_add_fwd", p(a), p(b), p(y), a.numel(), _st())
    return y


def row_scale_add(x, row_scale, add=None):
    """x * row_scale[:, None] (+ add): the dropTokenRate mask of the V1 / V2 / V3 video stream."""
    rows, dim = x.shape
    y = torch.empty_like(x)
    _lib.call("amt_row_scale_add_fwd", p(x), p(row_scale), p(add), p(y), rows, dim, _st())
    return y


def rmsnorm(x, w, resid=None, eps=1e-6):
    """RMSNorm(x (+ resid)) * w (custom_transformer.py:27-45)."""
    rows, dim = x.shape
    y = torch.empty_like(x)
    _lib.call("amt_rmsnorm_resid_fwd", p(x), p(resid), p(w), p(y), rows, dim, float(eps), _st())
    return y


def rope(x, cache, pos=0, out=None):
    """x (n0, seq, n2, hd) contiguous, cache (>=pos+seq, cache_half, 2): RotaryPositionalEmbeddings.forward on the
    cache rows pos..pos+seq-1 (pos > 0: one decode position); `out` receives the result (same numel)."""
    n0, seq, n2, hd = x.shape
    y = torch.empty_like(x) if out is None else out
    c = cache[pos:pos + seq].contiguous()
    _lib.call("amt_rope_fwd", p(x), p(c), p(y), n0, seq, n2, hd, cache.shape[1], _st())
    return y


def attention(q, k, v, strides, B, H, Lq, Lk, hd, causal, q_scale, out):
    s = (C.c_int64 * 12)(*strides)
    _lib.call("amt_attn_fwd", p(q), p(k), p(v), p(out), s, B, H, Lq, Lk, hd, int(causal), 1, float(q_scale), _st())
    return out


def glu(x, e):
    """GLUExpert.forward on rows x (n, d): W2((W1 x + b1) * silu(Wg x + bg)) + b2."""
    from .model.moe import expert_dff, expert_tensors
    n, d = x.shape
    dff = expert_dff(e)
    out = torch.empty(n, d, device=x.device, dtype=torch.float32)
    scratch = torch.empty(2 * n * dff, device=x.device, dtype=torch.float32)
    t = expert_tensors(e)           # a SiLUExpert has no linear1: y = W2 silu(W x + b) + b2
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


def _off(t, col):
    """Pointer to column `col` of row 0 of a contiguous 2-D fp32 tensor (a strided slice for the *_ex entry points)."""
    assert t.is_contiguous() and t.dtype == torch.float32
    return C.c_void_p(t.data_ptr() + 4 * int(col))


def linear_ex(x, w, b=None, resid=None, act=0, x_col=0, K=None, out=None):
    """y = act(x[:, x_col:x_col+K] w[N,K]^T + b (+ resid)); act 0 none / 1 ReLU / 2 sigmoid / 3 SiLU.  K % 32 == 0."""
    M, ldx = x.shape
    N, ldw = w.shape
    K = ldw if K is None else K
    y = torch.empty(M, N, device=x.device, dtype=torch.float32) if out is None else out
    _lib.call("amt_linear_ex_fwd", _off(x, x_col), ldx, p(w), ldw, p(b), p(resid), N, p(y), N, M, N, K, int(act), _st())
    return y


def layernorm_post(x, w, b, resid=None, post=None, eps=1e-5):
    """LayerNorm(x (+ resid)) + post."""
    rows, dim = x.shape
    y = torch.empty_like(x)
    _lib.call("amt_layernorm_post_fwd", p(x), p(resid), p(w), p(b), p(post), p(y), rows, dim, float(eps), _st())
    return y


def concat2(a, b, ld_out):
    """[a | b | 0-pad] row-wise; a (rows, da), b (rows, db)."""
    rows, da = a.shape
    out = torch.empty(rows, ld_out, device=a.device, dtype=torch.float32)
    _lib.call("amt_concat2_fwd", p(a), da, p(b), b.shape[1], p(out), rows, ld_out, _st())
    return out


def dwconv1d_silu(xz, C_, w, bias, B, L, reverse=False):
    """Causal depthwise conv + SiLU over the first C_ columns of xz (B*L, ld); w (C_, K)."""
    y = torch.empty(B * L, C_, device=xz.device, dtype=torch.float32)
    _lib.call("amt_dwconv1d_silu_fwd", p(xz), xz.shape[1], p(w), p(bias), p(y), B, L, C_, w.shape[1], int(reverse), _st())
    return y


def rnn_seq(xproj, w_hh, b_hh, y, col, B, L, d, gates, reverse=False, n_dirs=1):
    """One layer of nn.LSTM (gates=4) or nn.GRU (gates=3) over projected inputs xproj (B*L, n_dirs*gates*d); h_t goes to columns
    [col, col + n_dirs*d) of y (B*L, ldy).  n_dirs=2: both directions at once over stacked w_hh (2, gates*d, d) / b_hh."""
    _lib.call("amt_rnn_seq_fwd", p(xproj), xproj.shape[1], p(w_hh), p(b_hh), _off(y, col), y.shape[1], B, L, d, gates, int(reverse),
              n_dirs, _st())
    return y


def selective_scan(xc, draw, dt_bias, A_log, dbc, R, D, xz, B, L, version=1, reverse=False):
    """Selective scan + gate.  xc, draw (B*L, ED); dbc (B*L, R+2N) = x_proj output (B at column R, C at R+N);
    xz (B*L, 2*ED) = in_proj output (gate branch z at column ED)."""
    ED, N = A_log.shape
    y = torch.empty(B * L, ED, device=xc.device, dtype=torch.float32)
    _lib.call("amt_selective_scan_fwd", p(xc), ED, p(draw), ED, p(dt_bias), p(A_log), _off(dbc, R), _off(dbc, R + N), dbc.shape[1],
              p(D), _off(xz, ED), xz.shape[1], p(y), ED, B, L, ED, N, int(version), int(reverse), _st())
    return y
