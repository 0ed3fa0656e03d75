"""Regression head: the CPU oracle against the reference's own outputs (tests/golden/g_reg.npz)."""
import ast

import numpy as np
import torch

from oracle import reg_oracle as R
from video2music_amd import synthetic

CASES = ((1, 40), (3, 17), (2, 300))


def reg_sd(g, seed=5):
    shapes = [(str(k), ast.literal_eval(str(s))) for k, s in zip(g["keys"], g["shapes"])]
    return {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=seed).items()}, shapes


def test_oracle_equals_reference_regression_head(golden):
    g = golden("g_reg.npz")
    sd, _ = reg_sd(g)
    for B, S in CASES:
        col = {}
        ln_nd, inst = R.forward(sd, torch.from_numpy(g[f"sem_B{B}_S{S}"]), torch.from_numpy(g[f"emo_B{B}_S{S}"]), collect=col)
        n = R.n_layers_of(sd)
        assert (col[f"layer{n - 1}"] - torch.from_numpy(g[f"feat_B{B}_S{S}"])).abs().max() < 2e-5
        assert (ln_nd - torch.from_numpy(g[f"lnnd_B{B}_S{S}"])).abs().max() < 2e-5
        assert (inst - torch.from_numpy(g[f"inst_B{B}_S{S}"])).abs().max() < 2e-5


def test_scan_is_causal_and_backward_branch_sees_the_future(golden):
    """Structure check of the bidirectional layer: changing the last frame changes every output (backward branch),
    while the forward Mamba block alone is causal."""
    g = golden("g_reg.npz")
    sd, _ = reg_sd(g)
    x = torch.randn(1, 12, 32, generator=torch.Generator().manual_seed(0))
    x2 = x.clone()
    x2[0, -1] += 1.0
    p = "model.layers.0.mamba_forward."
    a, b = R.mamba_block(x, sd, p), R.mamba_block(x2, sd, p)
    assert torch.equal(a[:, :-1], b[:, :-1]) and not torch.equal(a[:, -1], b[:, -1])
    la, lb = R.bimamba_layer(x, sd, "model.layers.0."), R.bimamba_layer(x2, sd, "model.layers.0.")
    assert (la[:, 0] - lb[:, 0]).abs().max() > 0


def alt_sd(g, rm, seed=5):
    from video2music_amd.model.video_regression import VideoRegression
    m = VideoRegression(n_layers=2, d_model=32, d_hidden=64, total_vf_dim=30, regModel=rm)
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert [k for k, _ in shapes] == [str(k) for k in g[f"alt_{rm}_keys"]]       # the reference's key order, too
    return {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=seed).items()}


def test_oracle_equals_reference_for_the_other_mamba_regmodels(golden):
    """'bimamba' (original gate), 'mamba' / 'mamba+' (one-directional ResidualBlock stacks)."""
    g = golden("g_reg.npz")
    for rm in ("bimamba", "mamba", "mamba+", "moe_bimamba+", "sharedmoe_bimamba+", "lstm", "bilstm", "gru", "bigru", "cnngru", "cnnbigru", "moemamba"):
        sd = alt_sd(g, rm)
        ln_nd, inst = R.forward(sd, torch.from_numpy(g["alt_sem"]), torch.from_numpy(g["alt_emo"]), reg_model=rm)
        assert (ln_nd - torch.from_numpy(g[f"alt_{rm}_lnnd"])).abs().max() < 2e-5, rm
        assert (inst - torch.from_numpy(g[f"alt_{rm}_inst"])).abs().max() < 2e-5, rm
