"""Regression head `VideoRegression(regModel='bimamba+')` on the HIP path vs the reference's own
outputs (tests/golden/g_reg.npz) and vs the CPU oracle at the deployed size."""
import numpy as np
import pytest
import torch

from oracle import reg_oracle as R
from video2music_amd import ops, synthetic
from video2music_amd.model.video_regression import VideoRegression
from tests.test_reg_oracle import CASES, reg_sd

pytestmark = pytest.mark.gpu
TOL = 1e-3          # north star tolerance for fp32 paths (observed ~1e-5)


def build(cfg, seed):
    m = VideoRegression(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=seed).items()}
    m.load_state_dict(sd, strict=True)
    return m.cuda(), sd


def test_regression_head_vs_reference_golden(golden):
    g = golden("g_reg.npz")
    m, _ = build(dict(n_layers=2, d_model=32, d_hidden=64, total_vf_dim=30, regModel="bimamba+"), seed=5)
    for B, S in CASES:
        sem, emo = torch.from_numpy(g[f"sem_B{B}_S{S}"]).cuda(), torch.from_numpy(g[f"emo_B{B}_S{S}"]).cuda()
        with torch.no_grad():
            ln_nd, inst = m(sem, None, None, emo)
            feat = m.get_feature(sem, None, None, emo)
        assert ln_nd.shape == (B, S, 2) and inst.shape == (B, S, 40)
        for got, name in ((feat, "feat"), (ln_nd, "lnnd"), (inst, "inst")):
            err = (got.cpu() - torch.from_numpy(g[f"{name}_B{B}_S{S}"])).abs().max().item()
            assert err < 5e-5, (name, B, S, err)


def test_scan_and_conv_kernels_vs_oracle_both_directions():
    """Operator-level: ragged sizes (channels not a multiple of 16, L not a multiple of the 32-step stage), both
    directions, both gate versions."""
    gen = torch.Generator().manual_seed(1)
    for (B, L, ED, K, rk) in ((1, 5, 16, 4, 2), (2, 77, 40, 4, 3), (3, 300, 256, 4, 8), (1, 33, 24, 3, 2)):
        N = 16
        xz = torch.randn(B * L, 2 * ED, generator=gen)
        w, bias = torch.randn(ED, K, generator=gen) * 0.5, torch.randn(ED, generator=gen) * 0.1
        dbc = torch.randn(B * L, (rk + 2 * N + 3) // 4 * 4, generator=gen)
        draw = torch.randn(B * L, ED, generator=gen)
        dtb, A_log, D = torch.randn(ED, generator=gen), torch.randn(ED, N, generator=gen) * 0.5, torch.randn(ED, generator=gen)
        for reverse in (False, True):
            xi = xz[:, :ED].reshape(B, L, ED)
            src = torch.flip(xi, dims=[1]) if reverse else xi
            ref_c = R.causal_dwconv_silu(src, w.view(ED, 1, K), bias)
            got_c = ops.dwconv1d_silu(xz.cuda(), ED, w.cuda(), bias.cuda(), B, L, reverse).cpu().view(B, L, ED)
            rc = torch.flip(ref_c, dims=[1]) if reverse else ref_c
            assert (got_c - rc).abs().max() < 1e-5
            for version in (0, 1):
                fl = (lambda t: torch.flip(t, dims=[1])) if reverse else (lambda t: t)
                xc = fl(rc)
                delta = torch.nn.functional.softplus(fl(draw.view(B, L, ED)) + dtb)
                Bm, Cm = fl(dbc[:, rk:rk + N].reshape(B, L, N)), fl(dbc[:, rk + N:rk + 2 * N].reshape(B, L, N))
                y = R.selective_scan(xc, delta, -torch.exp(A_log), Bm, Cm, D)
                zs = torch.nn.functional.silu(fl(xz[:, ED:].reshape(B, L, ED)))
                ref = fl(y * zs + xc * (1 - torch.sigmoid(zs)) if version == 1 else y * zs)
                got = ops.selective_scan(rc.reshape(B * L, ED).contiguous().cuda(), draw.cuda(), dtb.cuda(), A_log.cuda(), dbc.cuda(), rk,
                                         D.cuda(), xz.cuda(), B, L, version=version, reverse=reverse).cpu().view(B, L, ED)
                scale = ref.abs().max().item() + 1.0
                assert (got - ref).abs().max().item() / scale < 2e-5, (B, L, ED, reverse, version)


def test_deployed_size_vs_oracle_and_postprocessing():
    """video2music.py:651 configuration (6 layers, d_model 128, d_hidden 256, 768+6 features), 300 frames: the HIP
    head against the CPU oracle, and the integer note-density / loudness levels the callers derive from it."""
    cfg = dict(n_layers=6, d_model=128, d_hidden=256, total_vf_dim=774, regModel="bimamba+")
    m, sd = build(cfg, seed=2)
    f = synthetic.synthetic_features(2, seed=9)
    sem, emo = torch.from_numpy(f["semantic"]), torch.from_numpy(f["emotion"])
    with torch.no_grad():
        ln_nd, inst = m(sem.cuda(), None, None, emo.cuda())
    ref_ln, ref_inst = R.forward(sd, sem, emo)
    assert (ln_nd.cpu() - ref_ln).abs().max() < TOL and (inst.cpu() - ref_inst).abs().max() < TOL
    nd, lv = R.postprocess(ln_nd.cpu())
    rnd, rlv = R.postprocess(ref_ln)
    assert (nd != rnd).mean() < 0.01 and (lv != rlv).mean() < 0.01          # rounding boundaries only
    # weights reloaded in place are picked up (padded copies are rebuilt)
    with torch.no_grad():
        m.in_proj[0].weight.mul_(0.5)
        ln2, _ = m(sem.cuda(), None, None, emo.cuda())
    assert (ln2 - ln_nd).abs().max() > 1e-4


def test_rejects_unbuilt_variants():
    with pytest.raises(NotImplementedError):
        VideoRegression(total_vf_dim=774, regModel="minGRU")
    with pytest.raises(ValueError):
        VideoRegression(total_vf_dim=774, d_model=256, regModel="bilstm")       # W_hh is held in registers: d_model <= 128
    with pytest.raises(ValueError):
        m, _ = build(dict(n_layers=1, d_model=32, d_hidden=64, total_vf_dim=30, regModel="bimamba+"), seed=0)
        m(torch.zeros(1, 4, 20).cuda(), None, None, torch.zeros(1, 4, 6).cuda())


@pytest.mark.parametrize("rm", ["bimamba", "mamba", "mamba+", "moe_bimamba+", "sharedmoe_bimamba+", "lstm", "bilstm", "gru", "bigru", "cnngru", "cnnbigru", "moemamba"])
def test_other_mamba_regmodels_vs_reference_golden(golden, rm):
    g = golden("g_reg.npz")
    m, _ = build(dict(n_layers=2, d_model=32, d_hidden=64, total_vf_dim=30, regModel=rm), seed=5)
    assert list(m.state_dict()) == [str(k) for k in g[f"alt_{rm}_keys"]]
    with torch.no_grad():
        ln_nd, inst = m(torch.from_numpy(g["alt_sem"]).cuda(), None, None, torch.from_numpy(g["alt_emo"]).cuda())
    assert (ln_nd.cpu() - torch.from_numpy(g[f"alt_{rm}_lnnd"])).abs().max().item() < 5e-5
    assert (inst.cpu() - torch.from_numpy(g[f"alt_{rm}_inst"])).abs().max().item() < 5e-5


def test_unbuilt_regmodels_say_so():
    for rm in ("minGRU",):
        with pytest.raises(NotImplementedError):
            VideoRegression(total_vf_dim=30, regModel=rm)


@pytest.mark.parametrize("rm", ["bilstm", "bigru", "lstm", "cnnbigru"])
def test_recurrent_heads_at_deployed_width_vs_oracle(rm):
    """d_model = 128 (the callers' -d_model_reg): 2 threads per gate row, 64 weights each in registers; 300 frames, 2 clips."""
    m, sd = build(dict(n_layers=3, d_model=128, d_hidden=256, total_vf_dim=774, regModel=rm), seed=9)
    rs = np.random.RandomState(2)
    sem = torch.from_numpy(rs.standard_normal((2, 300, 768)).astype(np.float32))
    z = rs.standard_normal((2, 300, 6))
    emo = torch.from_numpy((np.exp(z) / np.exp(z).sum(-1, keepdims=True)).astype(np.float32))
    ref_ln, ref_inst = R.forward(sd, sem, emo, reg_model=rm)
    with torch.no_grad():
        ln_nd, inst = m(sem.cuda(), None, None, emo.cuda())
    assert ln_nd.shape == (2, 300, 2) and inst.shape == (2, 300, 40)
    assert (ln_nd.cpu() - ref_ln).abs().max().item() < 1e-4 and (inst.cpu() - ref_inst).abs().max().item() < 1e-4


@pytest.mark.parametrize("gates", [3, 4])
def test_rnn_seq_kernel_shapes_vs_cell_equations(gates):
    """Operator-level: hidden sizes 8..128, one-step and ragged lengths, one direction (forward / backward) and both at once,
    against the cell equations in fp64."""
    gen = torch.Generator().manual_seed(gates)
    for (B, L, d, dirs) in ((1, 1, 8, 1), (3, 2, 16, 2), (2, 77, 64, 2), (1, 300, 128, 2), (2, 5, 128, 1), (1, 33, 40, 2)):
        G = gates * d
        xp = torch.randn(B * L, dirs * G, generator=gen)
        wh, bh = torch.randn(dirs, G, d, generator=gen) * (1.0 / d ** 0.5), torch.randn(dirs, G, generator=gen) * 0.1

        def cell(xrow, h, c, r):
            gx, gh = xrow.double(), h @ wh[r].double().t() + bh[r].double()
            if gates == 4:
                i, f, g, o = (gx + gh).chunk(4, -1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                return torch.sigmoid(o) * torch.tanh(c), c
            xr, xz, xn = gx.chunk(3, -1)
            hr, hz, hn = gh.chunk(3, -1)
            rr, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
            return (1 - z) * torch.tanh(xn + rr * hn) + z * h, c

        ref = torch.zeros(B, L, dirs * d, dtype=torch.float64)
        x3 = xp.view(B, L, dirs * G)
        for r in range(dirs):
            h, c = torch.zeros(B, d, dtype=torch.float64), torch.zeros(B, d, dtype=torch.float64)
            for t in (range(L - 1, -1, -1) if r else range(L)):
                h, c = cell(x3[:, t, r * G:(r + 1) * G], h, c, r)
                ref[:, t, r * d:(r + 1) * d] = h
        y = torch.full((B * L, dirs * d), float("nan")).cuda()
        ops.rnn_seq(xp.cuda(), wh.cuda().contiguous(), bh.cuda().contiguous(), y, 0, B, L, d, gates, n_dirs=dirs)
        assert (y.cpu().double().view(B, L, dirs * d) - ref).abs().max().item() < 2e-5, (B, L, d, dirs)
        if dirs == 2:       # each direction alone, through the single-direction form, writes the same columns
            y1 = torch.full((B * L, 2 * d), float("nan")).cuda()
            for r in range(2):
                ops.rnn_seq(xp[:, r * G:(r + 1) * G].contiguous().cuda(), wh[r].cuda().contiguous(), bh[r].cuda().contiguous(), y1, r * d, B, L, d,
                            gates, reverse=bool(r))
            assert torch.equal(y1, y)


def test_wide_state_scan_vs_oracle():
    """The scan with N = 32 .. 256 states per channel ('moemamba': d_state = d_hidden), both directions and gate versions."""
    gen = torch.Generator().manual_seed(3)
    for (B, L, ED, N) in ((1, 5, 16, 32), (2, 40, 24, 64), (1, 77, 64, 128), (2, 300, 64, 256)):
        rk = 4
        xc = torch.randn(B * L, ED, generator=gen)
        xz = torch.randn(B * L, 2 * ED, generator=gen)
        dbc = torch.randn(B * L, (rk + 2 * N + 3) // 4 * 4, generator=gen) * 0.5
        draw = torch.randn(B * L, ED, generator=gen)
        dtb, A_log, D = torch.randn(ED, generator=gen), torch.randn(ED, N, generator=gen) * 0.5, torch.randn(ED, generator=gen)
        for reverse in (False, True):
            for version in (0, 1):
                x3, d3 = xc.view(B, L, ED).double(), torch.nn.functional.softplus(draw.view(B, L, ED).double() + dtb.double())
                Bm, Cm = dbc.view(B, L, -1)[..., rk:rk + N].double(), dbc.view(B, L, -1)[..., rk + N:rk + 2 * N].double()
                flip = (lambda t: torch.flip(t, dims=[1])) if reverse else (lambda t: t)
                y = flip(R.selective_scan(flip(x3), flip(d3), -torch.exp(A_log.double()), flip(Bm), flip(Cm), D.double()))
                zs = torch.nn.functional.silu(xz.view(B, L, 2 * ED)[..., ED:].double())
                ref = y * zs + x3 * (1 - torch.sigmoid(zs)) if version == 1 else y * zs
                got = ops.selective_scan(xc.cuda(), draw.cuda(), dtb.cuda(), A_log.cuda(), dbc.cuda(), rk, D.cuda(), xz.cuda(), B, L,
                                         version=version, reverse=reverse)
                err = (got.cpu().double().view(B, L, ED) - ref).abs().max().item()
                assert err < 2e-4 * max(1.0, ref.abs().max().item()), (B, L, ED, N, reverse, version, err)
