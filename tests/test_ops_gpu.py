"""Per-operator parity: HIP kernels through the C ABI vs the CPU oracle, same seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import amt_oracle as O
from video2music_amd import _lib

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.as_tensor(x).cuda().contiguous()


def sp():
    return _lib.stream_ptr()


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize("M,N,K,relu,resid", [(128, 128, 32, 0, False), (300, 159, 512, 0, False), (1000, 1536, 128, 1, True),
                                             (15, 512, 512, 0, False), (257, 130, 1312, 0, True),
                                             (5000, 1024, 512, 1, True),        # 64x64-tile instantiation, many tiles
                                             (4096, 3072, 2048, 0, False)])     # long K, >= 768 tiles: the 128x128 instantiation
def test_linear(M, N, K, relu, resid):
    rs = np.random.RandomState(M + N)
    x, w, b = rnd(rs, M, K), rnd(rs, N, K, scale=K ** -0.5), rnd(rs, N)
    r = rnd(rs, M, N) if resid else None
    ref = O.linear(x.double(), w.double(), b.double())
    if resid:
        ref = ref + r.double()
    if relu:
        ref = torch.relu(ref)
    y = torch.empty(M, N, device="cuda")
    dx, dw, db, dr = dev(x), dev(w), dev(b), dev(r) if resid else None
    _lib.call("amt_linear_fwd", _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(dr), _lib.ptr(y), M, N, K, relu, sp())
    assert (y.cpu().double() - ref).abs().max() < 2e-5      # fp32 fma chain vs fp64


@pytest.mark.parametrize("rows,dim", [(1, 128), (7, 512), (1000, 512), (33, 1024)])
def test_layernorm_rmsnorm(rows, dim):
    rs = np.random.RandomState(rows)
    x, r, w, b = rnd(rs, rows, dim, scale=3.0), rnd(rs, rows, dim), rnd(rs, dim), rnd(rs, dim)
    y = torch.empty(rows, dim, device="cuda")
    dx, dr, dw, db = dev(x), dev(r), dev(w), dev(b)
    _lib.call("amt_layernorm_fwd", _lib.ptr(dx), _lib.ptr(dr), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(y), rows, dim, 1e-5, sp())
    assert (y.cpu() - O.layer_norm(x + r, w, b)).abs().max() < 2e-5
    _lib.call("amt_layernorm_fwd", _lib.ptr(dx), None, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(y), rows, dim, 1e-5, sp())
    assert (y.cpu() - O.layer_norm(x, w, b)).abs().max() < 2e-5
    _lib.call("amt_rmsnorm_fwd", _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(y), rows, dim, 1e-6, sp())
    assert (y.cpu() - O.rms_norm(x, w)).abs().max() < 2e-5


def test_rmsnorm_rope_golden(golden):
    g = golden("g_rms_rope.npz")
    x, w = dev(g["rms_x"]), dev(g["rms_w"])
    y = torch.empty_like(x)
    _lib.call("amt_rmsnorm_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), x.shape[0] * x.shape[1], x.shape[2], 1e-6, sp())
    assert np.abs(y.cpu().numpy() - g["rms_y"]).max() < 1e-5
    cache = dev(O.rope_cache(128, 300))
    for B in (1, 2):                     # the (H,L,B,hd) view of custom_transformer.py:1044-1053 with a dim=d_model cache
        x = dev(g[f"rope_x_B{B}"])
        L = x.shape[0]
        y = torch.empty_like(x)
        _lib.call("amt_rope_fwd", _lib.ptr(x), _lib.ptr(cache), _lib.ptr(y), 4, L, B, 32, 64, sp())
        assert np.abs(y.cpu().numpy() - g[f"rope_y_B{B}"]).max() < 1e-5
    x = dev(g["rope_hd_x"])              # documented (b, s, n_h, h_d) use with a head-dim cache
    y = torch.empty_like(x)
    c2 = dev(O.rope_cache(32, 64))
    _lib.call("amt_rope_fwd", _lib.ptr(x), _lib.ptr(c2), _lib.ptr(y), 2, 12, 4, 32, 16, sp())
    assert np.abs(y.cpu().numpy() - g["rope_hd_y"]).max() < 1e-5


def ref_rpr_attn(q, k, v, Er, H):
    qh, kh, vh = O.split_heads(q, H), O.split_heads(k, H), O.split_heads(v, H)
    L = q.shape[1]
    s = qh @ kh.transpose(-1, -2)
    qe = torch.einsum("bhld,md->bhlm", qh, Er[Er.shape[0] - L:])
    s = s + O.skew(qe) + torch.triu(torch.full((L, L), float("-inf"), dtype=q.dtype), diagonal=1)
    return O.merge_heads(torch.softmax(s, -1) @ vh)


@pytest.mark.parametrize("B,H,L,hd,er_len", [(1, 4, 1, 32, 300), (2, 4, 12, 32, 300), (3, 4, 64, 32, 300), (2, 8, 129, 64, 200),
                                            (1, 8, 300, 64, 300), (2, 2, 257, 64, 1024), (1, 2, 70, 128, 128),
                                            (1, 8, 1024, 64, 1024),        # config 2's full length: L = er_len = 1024
                                            (2, 8, 1, 16, 64), (2, 8, 45, 16, 64), (1, 3, 300, 16, 300)])   # head_dim 16 (d_model 128 with 8 heads)
def test_rpr_attention_prefill(B, H, L, hd, er_len):
    rs = np.random.RandomState(L)
    E = H * hd
    q, k, v = rnd(rs, B, L, E, scale=0.5), rnd(rs, B, L, E), rnd(rs, B, L, E)
    Er = torch.from_numpy(rs.uniform(size=(er_len, hd)).astype(np.float32))
    ref = ref_rpr_attn(q.double(), k.double(), v.double(), Er.double(), H)
    o = torch.empty(B, L, E, device="cuda")
    dq, dk, dv, de = dev(q), dev(k), dev(v), dev(Er)
    _lib.call("amt_rpr_attn_fwd", _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(de), _lib.ptr(o), B, H, L, hd, er_len, sp())
    assert (o.cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("B,H,L,hd,er_len", [(1, 4, 1, 32, 300), (2, 4, 12, 32, 300), (2, 4, 33, 32, 300), (2, 8, 129, 64, 200),
                                            (1, 8, 300, 64, 300), (1, 2, 257, 64, 1024), (1, 2, 70, 128, 128), (1, 8, 1024, 64, 1024),
                                            (2, 8, 45, 16, 64)])
def test_rpr_attention_prefill_without_causal_mask(B, H, L, hd, er_len):
    """forward(mask=False) of the reference (model/video_music_transformer.py:978-982): all keys visible, and `_skew`
    (model/rpr.py:439-455, verbatim in the oracle) leaves the relative term zero above the diagonal."""
    rs = np.random.RandomState(1000 + L)
    E = H * hd
    q, k, v = rnd(rs, B, L, E, scale=0.5), rnd(rs, B, L, E), rnd(rs, B, L, E)
    Er = torch.from_numpy(rs.uniform(size=(er_len, hd)).astype(np.float32))
    qh, kh, vh = (O.split_heads(t.double(), H) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) + O.skew(torch.einsum("bhld,md->bhlm", qh, Er.double()[er_len - L:]))
    ref = O.merge_heads(torch.softmax(s, -1) @ vh)
    o = torch.empty(B, L, E, device="cuda")
    dq, dk, dv, de = dev(q), dev(k), dev(v), dev(Er)
    _lib.call("amt_rpr_attn_nomask_fwd", _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(de), _lib.ptr(o), B, H, L, hd, er_len, sp())
    assert (o.cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("B,H,Lq,Lk,hd,causal", [(2, 4, 5, 300, 32, 0), (1, 8, 300, 300, 64, 0), (2, 8, 130, 77, 64, 0),
                                                (2, 8, 64, 64, 64, 1), (1, 2, 33, 100, 128, 0), (2, 8, 40, 300, 16, 0), (1, 8, 300, 300, 16, 1)])
def test_cross_attention_prefill(B, H, Lq, Lk, hd, causal):
    rs = np.random.RandomState(Lq * 7 + Lk)
    E = H * hd
    q, k, v = rnd(rs, B, Lq, E, scale=0.5), rnd(rs, B, Lk, E), rnd(rs, B, Lk, E)
    qh, kh, vh = (O.split_heads(t.double(), H) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2)
    if causal:
        s = s + torch.triu(torch.full((Lq, Lk), float("-inf"), dtype=s.dtype), diagonal=1)
    ref = O.merge_heads(torch.softmax(s, -1) @ vh)
    o = torch.empty(B, Lq, E, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.call("amt_cross_attn_fwd", _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(o), B, H, Lq, Lk, hd, causal, sp())
    assert (o.cpu().double() - ref).abs().max() < 2e-5


def test_softmax_rescale_branch_forced():
    """Online-softmax rescale: a late key tile carries a much larger score than every earlier one."""
    rs = np.random.RandomState(5)
    B, H, Lq, Lk, hd = 1, 1, 32, 128, 64
    q, k, v = rnd(rs, B, Lq, hd, scale=0.3), rnd(rs, B, Lk, hd, scale=0.3), rnd(rs, B, Lk, hd)
    k[0, 100] = q[0, 7] * 40.0          # spike in the 4th tile for one query row
    k[0, 3] = q[0, 20] * 25.0           # and in the first tile for another
    s = q.double() @ k.double().transpose(-1, -2)
    ref = torch.softmax(s, -1) @ v.double()
    o = torch.empty(B, Lq, hd, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.call("amt_cross_attn_fwd", _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(o), B, H, Lq, Lk, hd, 0, sp())
    assert (o.cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("B,H,hd,cap,pos,rpr", [(1, 4, 32, 300, 0, True), (3, 4, 32, 300, 17, True), (32, 8, 64, 1024, 1023, True),
                                               (5, 8, 64, 1024, 500, True), (32, 8, 64, 300, 299, False), (2, 2, 128, 64, 63, False),
                                               (2, 8, 16, 64, 40, False), (3, 8, 16, 64, 63, True)])
def test_attention_decode(B, H, hd, cap, pos, rpr):
    rs = np.random.RandomState(pos + B)
    q = rnd(rs, B, H * hd, scale=0.5)
    kc, vc = rnd(rs, B, H, cap, hd), rnd(rs, B, H, cap, hd)
    er_len = cap + 5
    Er = torch.from_numpy(rs.uniform(size=(er_len, hd)).astype(np.float32)) if rpr else None
    qh = q.view(B, H, 1, hd).double()
    K, Vv = kc[:, :, :pos + 1].double(), vc[:, :, :pos + 1].double()
    s = qh @ K.transpose(-1, -2)
    if rpr:       # closed form A1: bias[j] = q . Er[er_len-1-(pos-j)]
        idx = er_len - 1 - (pos - torch.arange(pos + 1))
        s = s + torch.einsum("bhqd,jd->bhqj", qh, Er[idx].double())
    ref = (torch.softmax(s, -1) @ Vv).reshape(B, H * hd)
    o = torch.empty(B, H * hd, device="cuda")
    dq, dk, dv, de = dev(q), dev(kc), dev(vc), dev(Er) if rpr else None
    _lib.call("amt_attn_decode_fwd", _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(de), _lib.ptr(o), B, H, hd, cap, pos,
              er_len if rpr else 0, sp())
    assert (o.cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("B,N,K,ln,relu,resid", [(32, 512, 512, True, 0, True), (1, 1536, 512, False, 0, False), (7, 1024, 512, True, 1, False),
                                                (32, 512, 1024, False, 0, True), (3, 384, 128, True, 0, False), (5, 128, 256, False, 1, True),
                                                # wide products (several column tiles per workgroup): the MoE gate|up stack of config 2,
                                                # a tile count that is not a multiple of the tiles per workgroup, K = 1024
                                                (32, 14336, 512, True, 0, False), (20, 4112, 512, False, 1, True), (32, 4096, 1024, True, 0, True),
                                                (9, 6000, 1024, False, 0, False)])
def test_decode_linear(B, N, K, ln, relu, resid):
    rs = np.random.RandomState(N + K + B)
    x, w, b = rnd(rs, B, K, scale=2.0), rnd(rs, N, K, scale=K ** -0.5), rnd(rs, N)
    lw, lb = (rnd(rs, K), rnd(rs, K)) if ln else (None, None)
    r = rnd(rs, B, N) if resid else None
    xn = O.layer_norm(x.double(), lw.double(), lb.double()) if ln else x.double()
    ref = O.linear(xn, w.double(), b.double())
    if resid:
        ref = ref + r.double()
    if relu:
        ref = torch.relu(ref)
    y = torch.empty(B, N, device="cuda")
    xn_out = torch.zeros(B, K, device="cuda")
    scratch = torch.empty(((N + 15) // 16) * 16 * K, device="cuda")
    dx, dw, db = dev(x), dev(w), dev(b)
    dlw, dlb = (dev(lw), dev(lb)) if ln else (None, None)
    dr = dev(r) if resid else None
    _lib.call("amt_decode_linear_fwd", _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(dlw), _lib.ptr(dlb), _lib.ptr(dr),
              _lib.ptr(y), _lib.ptr(xn_out), _lib.ptr(scratch), B, N, K, relu, 1e-5, sp())
    assert (y.cpu().double() - ref).abs().max() < 5e-5
    if ln:
        assert (xn_out.cpu().double() - xn).abs().max() < 2e-5


def test_bad_arguments_fail_loudly():
    x = torch.zeros(4, 48, device="cuda")
    with pytest.raises(_lib.AmtError):       # K not a multiple of the k-step
        _lib.call("amt_linear_fwd", _lib.ptr(x), _lib.ptr(x), None, None, _lib.ptr(x), 4, 4, 48, 0, sp())
    with pytest.raises(_lib.AmtError):       # unsupported head dim
        _lib.call("amt_cross_attn_fwd", _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), 1, 1, 4, 4, 48, 0, sp())


def test_attention_random_shape_sweep():
    """Seeded sweep over ragged shapes (lengths not multiples of the 32-key tile / 128-query block, B and H > 1)."""
    rs = np.random.RandomState(2024)
    for case in range(24):
        hd = int(rs.choice([32, 64, 128]))
        H = int(rs.choice([1, 2, 4]))
        B = int(rs.randint(1, 4))
        rpr = bool(case % 2)
        Lq = int(rs.randint(1, 200))
        Lk = Lq if rpr else int(rs.randint(1, 340))
        causal = 1 if rpr else int(rs.randint(0, 2)) if Lq == Lk else 0
        E = H * hd
        q, k, v = rnd(rs, B, Lq, E, scale=0.5), rnd(rs, B, Lk, E), rnd(rs, B, Lk, E)
        qh, kh, vh = (O.split_heads(t.double(), H) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2)
        er_len = Lq + int(rs.randint(0, 9))
        Er = torch.from_numpy(rs.uniform(size=(er_len, hd)).astype(np.float32))
        if rpr:
            s = s + O.skew(torch.einsum("bhld,md->bhlm", qh, Er[er_len - Lq:].double()))
        if causal:
            s = s + torch.triu(torch.full((Lq, Lk), float("-inf"), dtype=s.dtype), diagonal=1)
        ref = O.merge_heads(torch.softmax(s, -1) @ vh)
        o = torch.empty(B, Lq, E, device="cuda")
        dq, dk, dv, de = dev(q), dev(k), dev(v), dev(Er)
        if rpr:
            _lib.call("amt_rpr_attn_fwd", _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(de), _lib.ptr(o), B, H, Lq, hd, er_len, sp())
        else:
            _lib.call("amt_cross_attn_fwd", _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(o), B, H, Lq, Lk, hd, causal, sp())
        err = (o.cpu().double() - ref).abs().max().item()
        assert err < 3e-5, (case, B, H, Lq, Lk, hd, rpr, causal, err)


def test_attention_decode_random_sweep():
    rs = np.random.RandomState(77)
    for case in range(24):
        hd = int(rs.choice([16, 32, 64, 128]))
        H = int(rs.choice([1, 2, 8]))
        B = int(rs.randint(1, 33))
        cap = int(rs.randint(1, 400))
        pos = int(rs.randint(0, cap))
        rpr = bool(case % 2)
        er_len = cap + int(rs.randint(0, 5))
        q = rnd(rs, B, H * hd, scale=0.5)
        kc, vc = rnd(rs, B, H, cap, hd), rnd(rs, B, H, cap, hd)
        Er = torch.from_numpy(rs.uniform(size=(er_len, hd)).astype(np.float32))
        qh = q.view(B, H, 1, hd).double()
        s = qh @ kc[:, :, :pos + 1].double().transpose(-1, -2)
        if rpr:
            idx = er_len - 1 - (pos - torch.arange(pos + 1))
            s = s + torch.einsum("bhqd,jd->bhqj", qh, Er[idx].double())
        ref = (torch.softmax(s, -1) @ vc[:, :, :pos + 1].double()).reshape(B, H * hd)
        o = torch.empty(B, H * hd, device="cuda")
        dq, dk, dv, de = dev(q), dev(kc), dev(vc), dev(Er) if rpr else None
        _lib.call("amt_attn_decode_fwd", _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(de), _lib.ptr(o), B, H, hd, cap, pos,
                  er_len if rpr else 0, sp())
        err = (o.cpu().double() - ref).abs().max().item()
        assert err < 3e-5, (case, B, H, hd, cap, pos, rpr, err)


def _ln64(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b, mu, 1.0 / torch.sqrt(var + eps)


def test_folded_decode_attention_random_sweep():
    """attn_decode with the LayerNorm folded through the query (and new key/value) projection, against the unfolded
    arithmetic in fp64: q = (LN(u) W^T + b) * scale with W' = W o gamma, g = rowsum(W'), c = W beta + b."""
    rs = np.random.RandomState(5)
    for case in range(16):
        hd = int(rs.choice([32, 64, 128]))
        H = int(rs.choice([1, 4, 8]))
        d = H * hd
        if d > 1024:
            continue
        B = int(rs.randint(1, 33))
        cap = int(rs.randint(2, 300))
        new_kv = case % 2
        rpr = bool((case // 2) % 2) and new_kv
        t = int(rs.randint(0, cap))
        n_out = 3 * d if new_kv else d
        u = rnd(rs, B, d) + float(rs.uniform(-2, 2))                       # a mean far from 0 stresses the cancellation
        gam, bet = 1 + rnd(rs, d, scale=0.2), rnd(rs, d, scale=0.1)
        W, bW = rnd(rs, n_out, d, scale=d ** -0.5), rnd(rs, n_out, scale=0.1)
        kc, vc = rnd(rs, B, H, cap, hd), rnd(rs, B, H, cap, hd)
        er_len = cap + 3
        Er = torch.from_numpy(rs.uniform(size=(er_len, hd)).astype(np.float32))
        scale = hd ** -0.5
        # producer side (fp64 here): raw = u (W o gamma)^T ; vectors g, c
        Wp = W.double() * gam.double()
        raw = (u.double() @ Wp.t()).float()
        g, c = Wp.sum(1).float(), (W.double() @ bet.double() + bW.double()).float()
        # reference
        xn, _, _ = _ln64(u.double(), gam.double(), bet.double())
        proj = xn @ W.double().t() + bW.double()
        q = (proj[:, :d] * scale).view(B, H, 1, hd)
        K, V = kc.double().clone(), vc.double().clone()
        n_keys = t + 1 if new_kv else int(rs.randint(1, cap + 1))
        if new_kv:
            K[:, :, t], V[:, :, t] = proj[:, d:2 * d].view(B, H, hd), proj[:, 2 * d:].view(B, H, hd)
        s = q @ K[:, :, :n_keys].transpose(-1, -2)
        if rpr:
            idx = er_len - 1 - (t - torch.arange(n_keys))
            s = s + torch.einsum("bhqd,jd->bhqj", q, Er[idx].double())
        ref = (torch.softmax(s, -1) @ V[:, :, :n_keys]).reshape(B, d)
        dk, dv = dev(kc), dev(vc)
        o, xo = torch.empty(B, d, device="cuda"), torch.empty(B, d, device="cuda")
        pos = torch.tensor([t], dtype=torch.int32, device="cuda")
        d_raw, d_er, d_u, d_g, d_c, d_gam, d_bet = (dev(v) for v in (raw, Er, u, g, c, gam, bet))    # keep the device copies alive
        _lib.call("amt_attn_decode_fold_fwd", _lib.ptr(d_raw), n_out, _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(d_er) if rpr else None,
                  _lib.ptr(d_u), _lib.ptr(d_g), _lib.ptr(d_c), _lib.ptr(d_gam), _lib.ptr(d_bet), _lib.ptr(xo), _lib.ptr(o),
                  B, H, hd, cap, _lib.ptr(pos) if new_kv else None, n_keys, er_len if rpr else 0, new_kv, 1e-5, scale, sp())
        tag = (case, B, H, hd, cap, t, new_kv, rpr)
        assert (o.cpu().double() - ref).abs().max().item() < 5e-5, tag
        assert (xo.cpu().double() - xn).abs().max().item() < 2e-5, tag
        if new_kv:
            assert (dk.cpu().double()[:, :, t] - K[:, :, t]).abs().max().item() < 3e-5, tag
            assert (dv.cpu().double()[:, :, t] - V[:, :, t]).abs().max().item() < 3e-5, tag
            keep = torch.ones(cap, dtype=torch.bool)
            keep[t] = False
            assert torch.equal(dk.cpu()[:, :, keep], kc[:, :, keep])             # only row t of the cache is touched


def test_skinny_gemm_two_sources_split_and_folded_ffn_prologue():
    """decode_gemm in the three forms of the folded decode chain (G1/G2: two-source rows + column split; G3: folded-FFN
    prologue) against fp64, incl. ragged batch sizes and K = 1536."""
    rs = np.random.RandomState(11)
    for (B, K1, K2, n_low, n_high, pro) in ((32, 512, 512, 512, 512, 0), (7, 128, 128, 128, 256, 0), (32, 1024, 512, 512, 1536, 1),
                                            (19, 256, 128, 128, 384, 1), (5, 256, 128, 128, 0, 1), (32, 1024, 512, 512, 160, 1)):
        K = K1 + K2
        x, x2 = rnd(rs, B, K1), rnd(rs, B, K2) + (0.7 if pro else 0.0)
        wl, bl = rnd(rs, n_low, K1, scale=K1 ** -0.5), rnd(rs, n_low, scale=0.1)
        wh, bh = (rnd(rs, n_high, K, scale=K ** -0.5), rnd(rs, n_high, scale=0.1)) if n_high else (None, None)
        fg, fc = rnd(rs, K1), rnd(rs, K1, scale=0.3)
        gam, bet = 1 + rnd(rs, K2, scale=0.2), rnd(rs, K2, scale=0.1)
        if pro:
            ln, mu, rstd = _ln64(x2.double(), gam.double(), bet.double())
            a = torch.relu((x.double() - mu * fg.double()) * rstd + fc.double())
            rows = torch.cat([a, ln], 1)
            ref_low = a @ wl.double().t() + bl.double() + ln[:, :n_low]
            resid = None
        else:
            rows = torch.cat([x.double(), x2.double()], 1)
            resid = x2[:, :n_low].contiguous() if K2 >= n_low else None
            ref_low = x.double() @ wl.double().t() + bl.double() + (resid.double() if resid is not None else 0)
        ref_high = rows @ wh.double().t() + bh.double() if n_high else None
        yl = torch.empty(B, n_low, device="cuda")
        yh = torch.empty(B, max(n_high, 1), device="cuda")
        sl = torch.empty((n_low + 15) // 16 * 16 * K1, device="cuda")
        sh = torch.empty(max((n_high + 15) // 16 * 16 * K, 1), device="cuda")
        D = {k: dev(v) for k, v in dict(x=x, x2=x2, wl=wl, bl=bl, fg=fg, fc=fc, gam=gam, bet=bet).items()}    # keep the device copies alive
        for k, v in dict(resid=resid, wh=wh, bh=bh).items():
            D[k] = dev(v) if v is not None else None
        A = lambda k, on=True: _lib.addr(D[k]) if on and D[k] is not None else None
        args = _lib.DecodeGemmArgs(x=A("x"), ldx=K1, x2=A("x2"), ldx2=K2, K1=K1, K=K, w_low=A("wl"), bias_low=A("bl"), resid=A("resid"), relu=0,
                                   w_high=A("wh"), bias_high=A("bh"), n_low=n_low, n_high=n_high, pro=pro, fold_g=A("fg", pro), fold_c=A("fc", pro),
                                   ln_w=A("gam", pro), ln_b=A("bet", pro), y_low=_lib.addr(yl), y_high=_lib.addr(yh) if n_high else None,
                                   scratch_low=_lib.addr(sl), scratch_high=_lib.addr(sh) if n_high else None, B=B, eps=1e-5)
        import ctypes
        _lib.call("amt_decode_gemm_ex_fwd", ctypes.byref(args), sp())
        tag = (B, K1, K2, n_low, n_high, pro)
        assert (yl.cpu().double() - ref_low).abs().max().item() < 3e-5, tag
        if n_high:
            assert (yh.cpu().double() - ref_high).abs().max().item() < 3e-5, tag
