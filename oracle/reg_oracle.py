"""CPU oracle of the reference's `VideoRegression(regModel='bimamba+')` forward.

TEST INFRASTRUCTURE: only tests/, bench tools' cpu legs and `oracle/make_goldens_reg.py` import this; the product path
never does.  A restatement in plain torch-CPU fp32 of

    model/video_regression.py:199-245   get_feature / forward: cat(semantic, emotion) -> in_proj -> encoder -> 2 heads
    model/bimamba.py:9-31,102-196       BiMambaEncoder of post-norm BiMambaEncoderLayer_V1
    model/mamba.py:160-323              MambaBlock (use_version=1, "Mamba+"): in_proj, causal depthwise conv + SiLU,
                                        x_proj / dt_proj / softplus, selective scan, the Mamba+ forget gate, out_proj

with the selective scan written as the plain recurrence (`selective_scan_seq`, :325-354); the reference runs the same
recurrence through a Blelloch parallel scan (`pscan.py`), which differs by fp32 rounding only.  PINNED by
tests/golden/g_reg.npz (outputs of the reference class itself, `oracle/make_goldens_reg.py`).
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def layer_norm(x, w, b, eps=LN_EPS):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def causal_dwconv_silu(x, w, b):
    """x (B,L,C); w (C,1,K) depthwise; nn.Conv1d(padding=K-1)[..., :L] (mamba.py:172-175,268-272): output l sees
    inputs l-K+1 .. l."""
    B, L, C = x.shape
    K = w.shape[-1]
    xp = F.pad(x, (0, 0, K - 1, 0))
    y = torch.zeros_like(x)
    for j in range(K):
        y = y + xp[:, j:j + L] * w[:, 0, j]
    return F.silu(y + b)


def selective_scan(x, delta, A, Bm, Cm, D):
    """h_t = exp(delta_t A) h_{t-1} + delta_t B_t x_t ; y_t = h_t . C_t + D x_t   (mamba.py:325-354)."""
    Bsz, L, ED = x.shape
    h = torch.zeros(Bsz, ED, A.shape[1], dtype=x.dtype)
    ys = []
    for t in range(L):
        dA = torch.exp(delta[:, t].unsqueeze(-1) * A)
        h = dA * h + (delta[:, t].unsqueeze(-1) * Bm[:, t].unsqueeze(1)) * x[:, t].unsqueeze(-1)
        ys.append((h * Cm[:, t].unsqueeze(1)).sum(-1))
    return torch.stack(ys, 1) + D * x


def mamba_block(x, sd, p, version=1, collect=None):
    """MambaBlock.forward (mamba.py:257-289) + ssm (:291-323).  sd keys under prefix p."""
    N = sd[p + "A_log"].shape[1]
    R = sd[p + "dt_proj.weight"].shape[1]
    xz = linear(x, sd[p + "in_proj.weight"], sd.get(p + "in_proj.bias"))
    xi, z = xz.chunk(2, dim=-1)
    xc = causal_dwconv_silu(xi, sd[p + "conv1d.weight"], sd[p + "conv1d.bias"])
    dbc = linear(xc, sd[p + "x_proj.weight"])
    dr, Bm, Cm = torch.split(dbc, [R, N, N], dim=-1)
    delta = F.softplus(linear(dr, sd[p + "dt_proj.weight"]) + sd[p + "dt_proj.bias"])
    A = -torch.exp(sd[p + "A_log"].float())
    y = selective_scan(xc, delta, A, Bm, Cm, sd[p + "D"].float())
    zs = F.silu(z)
    out = y * zs + xc * (1 - torch.sigmoid(zs)) if version == 1 else y * zs      # :283-287 (sigmoid of the SiLU'd gate)
    if collect is not None:
        collect.update(xc=xc, delta=delta, y=y, gated=out)
    return linear(out, sd[p + "out_proj.weight"], sd.get(p + "out_proj.bias"))


def bimamba_layer(x, sd, p, version=1):
    """BiMambaEncoderLayer_V1.forward, norm_first=False (bimamba.py:171-196); dropout is identity in eval."""
    xf = mamba_block(x, sd, p + "mamba_forward.", version)
    xf = layer_norm(xf + x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    xb = mamba_block(torch.flip(x, dims=[1]), sd, p + "mamba_backward.", version)
    xb = layer_norm(torch.flip(xb, dims=[1]) + x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    s = xf + xb
    if p + "ffn.gate.weight" in sd:             # a mixture layer in the FFN's place ('moe_bimamba+' / 'sharedmoe_bimamba+')
        from oracle.amt_oracle import moe_forward
        sub = {k[len(p) + 4:]: v for k, v in sd.items() if k.startswith(p + "ffn.")}
        f = moe_forward(s, sub, sub["gate.weight"].shape[0], k=2, shared="shared_expert.gate.weight" in sub)
        return layer_norm(f + s, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
    f = linear(torch.relu(linear(s, sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"])), sd[p + "ffn.3.weight"], sd[p + "ffn.3.bias"])
    return layer_norm(f + s, sd[p + "norm3.weight"], sd[p + "norm3.bias"])


def bimamba_layer_v0(x, sd, p):
    """BiMambaEncoderLayer.forward (bimamba.py:61-100), the layer of use_version 0: an FFN + norm pair per direction;
    `ffn2` is applied to the forward branch's x_f (:94), as written."""
    def ffn(t, q):
        return linear(torch.relu(linear(t, sd[p + q + ".0.weight"], sd[p + q + ".0.bias"])), sd[p + q + ".3.weight"], sd[p + q + ".3.bias"])
    xf = layer_norm(mamba_block(x, sd, p + "mamba_forward.", 0) + x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    xf = layer_norm(ffn(xf, "ffn1") + xf, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    xb = torch.flip(mamba_block(torch.flip(x, dims=[1]), sd, p + "mamba_backward.", 0), dims=[1])
    xb = layer_norm(xb + x, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
    xb = layer_norm(ffn(xf, "ffn2") + xb, sd[p + "norm4.weight"], sd[p + "norm4.bias"])
    return xf + xb


def n_layers_of(sd):
    n = 0
    while any(f"model.layers.{n}.{k}" in sd for k in ("norm1.weight", "norm.weight", "0.norm.weight")):
        n += 1
    return n


def residual_block(x, sd, p, version):
    """ResidualBlock.forward (mamba.py:139-142): mixer(RMSNorm(x)) + x, RMSNorm eps 1e-5 (:472-489)."""
    h = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * sd[p + "norm.weight"]
    return mamba_block(h, sd, p + "mixer.", version) + x


def rnn_stack(x, sd, kind, bidirectional):
    """torch.nn.LSTM / nn.GRU, batch_first, eval (the reference builds them at video_regression.py:124-135; the cell
    equations are torch's documented ones).  x (B, L, d); keys model.weight_ih_l{k}[_reverse] etc.; gate order i,f,g,o / r,z,n."""
    pre = "model.gru." if "model.gru.weight_ih_l0" in sd else "model."
    if pre == "model.gru.":                 # CNN_GRU (video_regression.py:84-103): Conv1d(k=7, padding=3) over time + SiLU first
        W, bc = sd["model.cnn.0.weight"], sd["model.cnn.0.bias"]
        K = W.shape[2]
        xp = F.pad(x, (0, 0, K // 2, K // 2))
        x = F.silu(sum(xp[:, j:j + x.shape[1]] @ W[:, :, j].t() for j in range(K)) + bc)
    n = 0
    while f"{pre}weight_ih_l{n}" in sd:
        n += 1
    for l in range(n):
        outs = []
        for rev in ([False, True] if bidirectional else [False]):
            sfx = f"_l{l}" + ("_reverse" if rev else "")
            Wi, Wh, bi, bh = (sd[f"{pre}{k}{sfx}"] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
            d = Wh.shape[1]
            B, L, _ = x.shape
            h, c = torch.zeros(B, d, dtype=x.dtype), torch.zeros(B, d, dtype=x.dtype)
            ys = [None] * L
            for t in (range(L - 1, -1, -1) if rev else range(L)):
                gx, gh = x[:, t] @ Wi.t() + bi, h @ Wh.t() + bh
                if kind == "lstm":
                    i, f, g, o = (gx + gh).chunk(4, dim=-1)
                    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                    h = torch.sigmoid(o) * torch.tanh(c)
                else:
                    xr, xz, xn = gx.chunk(3, dim=-1)
                    hr, hz, hn = gh.chunk(3, dim=-1)
                    r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
                    h = (1 - z) * torch.tanh(xn + r * hn) + z * h
                ys[t] = h
            outs.append(torch.stack(ys, 1))
        x = torch.cat(outs, dim=-1)
    return x


def forward(sd, sem, emotion, collect=None, reg_model="bimamba+"):
    """VideoRegression.forward (video_regression.py:199-245): returns (loudness_notedensity (B,S,2), instrument (B,S,40)).
    Scene offset and motion are accepted by the reference's signature but not used (:205-213 are commented out).
    reg_model: 'bimamba+' / 'bimamba' (BiMambaEncoder) or 'mamba+' / 'mamba' (Mamba stack); '+' = use_version 1."""
    version = 1 if reg_model.endswith("+") else 0
    reg_model = reg_model.replace("sharedmoe_", "").replace("moe_", "").replace("moemamba", "mamba")   # the mixture shows in the keys
    sd = {k: v.float() for k, v in sd.items()}
    vf = torch.cat([sem.float(), emotion.float()], dim=-1)
    x = linear(vf, sd["in_proj.0.weight"], sd["in_proj.0.bias"])
    if collect is not None:
        collect["in_proj"] = x
    rnn = reg_model in ("lstm", "bilstm", "gru", "bigru", "cnngru", "cnnbigru")
    if rnn:
        x = rnn_stack(x, sd, "lstm" if "lstm" in reg_model else "gru", "bi" in reg_model)
    for l in range(0 if rnn else n_layers_of(sd)):
        p = f"model.layers.{l}."
        if reg_model == "mamba" and p + "0.norm.weight" in sd:          # MoEMamba (mamba.py:102-129): ResidualBlock, then ResidualMoE
            from oracle.amt_oracle import moe_forward
            x = residual_block(x, sd, p + "0.", 0)
            h = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * sd[p + "1.norm.weight"]
            sub = {k[len(p) + 12:]: v for k, v in sd.items() if k.startswith(p + "1.moe_layer.")}
            x = moe_forward(h, sub, sub["gate.weight"].shape[0], k=2, shared=True) + x
        elif reg_model.startswith("bi"):
            x = bimamba_layer(x, sd, p, 1) if version == 1 else bimamba_layer_v0(x, sd, p)
        else:
            x = residual_block(x, sd, p, version)
        if collect is not None:
            collect[f"layer{l}"] = x
    ln_nd = linear(x, sd["regressor.weight"], sd["regressor.bias"])
    inst = torch.sigmoid(linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    return ln_nd, inst


def postprocess(ln_nd):
    """The integer note density and loudness level the callers derive (generate.py:401-409, video2music.py:855-865):
    column 0 = note density -> round, clip [0,40]; column 1 = loudness -> int(x*100), clip [0,50]."""
    y = ln_nd.reshape(-1, 2).numpy()
    import numpy as np
    nd = np.clip(np.round(y[:, 0:1]).astype(int), 0, 40)
    lv = np.clip((y[:, 1:2] * 100).astype(int), 0, 50)
    return nd, lv
