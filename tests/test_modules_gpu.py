"""Standalone modules of configs 4/5 and the kernel-level rows on the HIP path vs goldens produced by
the reference (tests/golden) and vs the CPU oracle at larger sizes."""
import numpy as np
import pytest
import torch

from oracle import amt_oracle as O
from video2music_amd import synthetic
from video2music_amd.model.custom_transformer import RMSNorm
from video2music_amd.model.grouped_query_attention import MultiheadGQA
from video2music_amd.model.moe import GLUExpert, MoELayer, SharedMoELayer
from video2music_amd.model.rotate_operation import RotaryPositionalEmbeddings
from tests.test_oracle_golden import GQA_SHAPES, moe_shapes

pytestmark = pytest.mark.gpu
TOL = 1e-3          # north star: within 1e-3 fp32 (observed errors are ~1e-5)


def load(module, shapes, seed):
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=seed).items()}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected and all(k == "bias" for k in missing), (missing, unexpected)
    return module.cuda().eval(), sd


@pytest.mark.parametrize("L,B", [(6, 1), (6, 3), (64, 1), (64, 3)])
@pytest.mark.parametrize("causal", [False, True])
def test_gqa_vs_reference_golden(golden, L, B, causal):
    g = golden("g_gqa.npz")
    m, _ = load(MultiheadGQA(256, 8, 2), GQA_SHAPES, 3)
    x = torch.from_numpy(g[f"x_L{L}_B{B}"]).cuda()
    y, w = m(x, x, x, is_causal=causal)
    assert w is None and y.shape == (L, B, 256)
    err = np.abs(y.cpu().numpy() - g[f"y_L{L}_B{B}_c{int(causal)}"]).max()
    assert err < 1e-4, err


@pytest.mark.parametrize("name,dim", [("hd", 32), ("full", 256)])
def test_gqa_rope_vs_reference_golden(golden, name, dim):
    """MultiheadGQA(RoPE=RotaryPositionalEmbeddings(dim, 80)) (grouped_query_attention.py:216,316-322): the rotation through the raw
    (heads, len, B, head_dim) view, cache built for head_dim (broadcast) and for embed_dim (folded and truncated), B in {1,2,3}."""
    g = golden("g_gqa_rope.npz")
    m, _ = load(MultiheadGQA(256, 8, 2, RoPE=RotaryPositionalEmbeddings(dim, 80)), GQA_SHAPES, 3)
    for L, B in ((6, 1), (6, 2), (64, 1), (64, 3)):
        x = torch.from_numpy(g[f"{name}_x_L{L}_B{B}"]).cuda()
        for causal in (False, True):
            y, _ = m(x, x, x, is_causal=causal)
            err = np.abs(y.cpu().numpy() - g[f"{name}_y_L{L}_B{B}_c{int(causal)}"]).max()
            assert err < 1e-4, (L, B, causal, err)
    xq, xk = torch.from_numpy(g[f"{name}_xq"]).cuda(), torch.from_numpy(g[f"{name}_xk"]).cuda()
    y, _ = m(xq, xk, xk)
    assert np.abs(y.cpu().numpy() - g[f"{name}_y_cross"]).max() < 1e-4
    with pytest.raises(ValueError):                    # longer than the cache
        x = torch.zeros(81, 1, 256, device="cuda")
        m(x, x, x)


def test_gqa_config4_vs_oracle():
    """Config 4: MultiheadGQA(512, 8, 2), is_causal, L=2048 (B=1) and the B=4 row-permutation quirk at L=512."""
    shapes = [("q_proj.weight", (512, 512)), ("q_proj.bias", (512,)), ("k_proj.weight", (128, 512)), ("k_proj.bias", (128,)),
              ("v_proj.weight", (128, 512)), ("v_proj.bias", (128,)), ("norm.weight", (512,)), ("norm.bias", (512,)),
              ("out_proj.weight", (512, 512)), ("out_proj.bias", (512,))]
    m, sd = load(MultiheadGQA(512, 8, 2), shapes, 4)
    rs = np.random.RandomState(0)
    for L, B in ((2048, 1), (512, 4)):
        x = torch.from_numpy(rs.standard_normal((L, B, 512)).astype(np.float32))
        ref = O.gqa_forward(x, x, x, sd, 8, 2, is_causal=True)
        y, _ = m(x.cuda(), x.cuda(), x.cuda(), is_causal=True)
        err = (y.cpu() - ref).abs().max().item()
        assert err < TOL, (L, B, err)
    # batch elements mix for B > 1 (raw .view reinterpretation): clip 0 alone differs from clip 0 in the batch
    x = torch.from_numpy(rs.standard_normal((64, 2, 512)).astype(np.float32)).cuda()
    both, _ = m(x, x, x, is_causal=True)
    alone, _ = m(x[:, :1].contiguous(), x[:, :1].contiguous(), x[:, :1].contiguous(), is_causal=True)
    assert (both[:, :1] - alone).abs().max().item() > 1e-2


def test_gqa_head_dim_16_vs_oracle():
    """MultiheadGQA(128, 8, 2): head_dim 16 (the attention's O^T tile is padded to 32 rows of d), causal and not, B = 1 and 3."""
    shapes = [("q_proj.weight", (128, 128)), ("q_proj.bias", (128,)), ("k_proj.weight", (32, 128)), ("k_proj.bias", (32,)),
              ("v_proj.weight", (32, 128)), ("v_proj.bias", (32,)), ("norm.weight", (128,)), ("norm.bias", (128,)),
              ("out_proj.weight", (128, 128)), ("out_proj.bias", (128,))]
    m, sd = load(MultiheadGQA(128, 8, 2), shapes, 6)
    rs = np.random.RandomState(1)
    for L, B, causal in ((300, 1, True), (70, 3, False), (33, 3, True)):
        x = torch.from_numpy(rs.standard_normal((L, B, 128)).astype(np.float32))
        ref = O.gqa_forward(x, x, x, sd, 8, 2, is_causal=causal)
        y, _ = m(x.cuda(), x.cuda(), x.cuda(), is_causal=causal)
        err = (y.cpu() - ref).abs().max().item()
        assert err < TOL, (L, B, causal, err)


@pytest.mark.parametrize("shared,n_exp,k", [(False, 8, 1), (True, 6, 3), (False, 4, 4), (True, 16, 5)])
def test_moe_other_top_k_vs_oracle(shared, n_exp, k):
    """MoELayer / SharedMoELayer with n_experts_per_token other than the class default of 2 (moe.py:150-160: `torch.topk(logits, k)`, softmax over
    the k logits, accumulation in expert order, shared expert / k): routing ids and weights and the output against the oracle."""
    from video2music_amd.model.moe import GLUExpert, MoELayer, SharedMoELayer
    d, dff = 128, 192
    m = (SharedMoELayer(GLUExpert(d, dff), d, n_experts=n_exp, n_experts_per_token=k) if shared else MoELayer(GLUExpert(d, dff), d, n_exp, k)).eval()
    shapes = [(kk, tuple(v.shape)) for kk, v in m.state_dict().items()]
    sd = {kk: torch.from_numpy(v) for kk, v in synthetic.synthetic_state_dict(shapes, seed=11 + k).items()}
    m.load_state_dict(sd)
    m = m.cuda()
    x = torch.from_numpy(np.random.RandomState(k).standard_normal((70, 3, d)).astype(np.float32))
    routing = {}
    ref = O.moe_forward(x, sd, n_exp, k=k, shared=shared, routing=routing)
    with torch.no_grad():
        y = m(x.cuda())
    assert (y.cpu() - ref).abs().max().item() < TOL
    idx, wts = m.last_routing
    assert idx.shape == (70, 3, k) and torch.equal(idx.cpu().long(), routing["idx"])
    assert (wts.cpu() - routing["weights"]).abs().max().item() < 1e-6


@pytest.mark.parametrize("name", ["moe", "shared"])
def test_moe_vs_reference_golden(golden, name):
    g = golden("g_moe.npz")
    shared = name == "shared"
    layer = (SharedMoELayer(GLUExpert(128, 256), 128, n_experts=8, n_experts_per_token=2, balancing=True) if shared
             else MoELayer(GLUExpert(128, 256), 128, n_experts=8, n_experts_per_token=2))
    m, _ = load(layer, moe_shapes(8, 128, 256, shared), 5)
    y = m(torch.from_numpy(g["x"]).cuda())
    idx, w = m.last_routing
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), g[f"idx_{name}"])       # selected expert ids, torch.topk order
    assert np.abs(w.cpu().numpy() - g[f"w_{name}"]).max() < 1e-5
    err = np.abs(y.cpu().numpy() - g[f"y_{name}"]).max()
    assert err < 1e-4, err


@pytest.mark.parametrize("shared", [False, True])
def test_moe_config5_vs_oracle(shared):
    """Config 5: 8 experts, top-2, d=512, d_ff=1024, x (1024, 4, 512)."""
    layer = (SharedMoELayer(GLUExpert(512, 1024), 512) if shared else MoELayer(GLUExpert(512, 1024), 512))
    m, sd = load(layer, moe_shapes(8, 512, 1024, shared), 6)
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.standard_normal((1024, 4, 512)).astype(np.float32))
    routing = {}
    ref = O.moe_forward(x, sd, 8, k=2, shared=shared, routing=routing)
    y = m(x.cuda())
    idx, w = m.last_routing
    same = (idx.cpu().long() == routing["idx"]).all(-1)
    assert same.float().mean() > 0.999          # routing ties within fp32 rounding may flip a token
    err = (y.cpu() - ref)[same].abs().max().item()
    assert err < TOL, err
    with pytest.raises(ValueError):
        m(x[:, 0].cuda())                       # 2-D input is rejected like the reference (moe.py:193)


def test_rmsnorm_and_rope_modules(golden):
    g = golden("g_rms_rope.npz")
    rms = RMSNorm(128).cuda()
    rms.weight.data = torch.from_numpy(g["rms_w"]).cuda()
    assert np.abs(rms(torch.from_numpy(g["rms_x"]).cuda()).cpu().numpy() - g["rms_y"]).max() < 1e-5
    rope = RotaryPositionalEmbeddings(128, 300).cuda()
    for B in (1, 2):
        x = torch.from_numpy(g[f"rope_x_B{B}"]).cuda()
        L = x.shape[0]
        y = rope(x.view(4, L, B, 32)).view(L, B, 128)
        assert np.abs(y.cpu().numpy() - g[f"rope_y_B{B}"]).max() < 1e-5
    rope_hd = RotaryPositionalEmbeddings(32, 64).cuda()
    y = rope_hd(torch.from_numpy(g["rope_hd_x"]).cuda())
    assert np.abs(y.cpu().numpy() - g["rope_hd_y"]).max() < 1e-5


def test_generate_cli_reads_feature_files_and_writes_lab(tmp_path):
    """`python -m video2music_amd.generate` on a miniature MuVi-Sync tree: features come
    from the files, ids go to `<id>_chords.lab`, and the ids equal a direct model.generate on the loaded tensors."""
    import os
    from tests.helpers_features import write_mini_dataset, mini_dataset_content
    from video2music_amd import generate as G, synthetic
    from video2music_amd.dataset import vevo_features as VF
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer
    from video2music_amd.utilities import constants as C_
    root, out = str(tmp_path / "data"), str(tmp_path / "out")
    write_mini_dataset(root, mini_dataset_content(seed=11))
    argv = ["-dataset_dir", root, "-output_dir", out, "--test_ids", "split:test", "--synthetic_weights", "-music_gen_version", "None",
            "-n_layers", "2", "-num_heads", "4", "-d_model", "128", "-dim_feedforward", "256", "-target_seq_length_chord", "24",
            "-max_sequence_chord", "300", "--sampler", "argmax", "-motion_type", "1", "--regression", "--midi", "-n_layers_reg", "2",
            "-d_model_reg", "32", "-dim_feedforward_reg", "64"]
    toks = G.main(argv).cpu()
    assert toks.shape == (2, 24)
    assert open(os.path.join(out, "003_chords.mid"), "rb").read(4) == b"MThd"
    rows = open(os.path.join(out, "017_loudness_density.csv")).read().splitlines()
    assert rows[0] == "frame,note_density,loudness_level" and len(rows) == 301 and all(0 <= int(r.split(",")[2]) <= 50 for r in rows[1:])
    for i, fid in enumerate(("003", "017")):
        chord, _, _, _, last = VF.read_chords(os.path.join(out, f"{fid}_chords.lab"), 300)
        assert last == 23 and chord[:24].tolist() == toks[i].tolist()
    f = VF.load_clips(root, ["003", "017"], motion_type=1)
    m = VideoMusicTransformer(n_layers=2, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300,
                              total_vf_dim=24 + 1 + 512 + 6, rpr=True).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()}, strict=False)
    m = m.cuda()
    key = torch.tensor([[VF.key_from_emotion(e)] for e in f["emotion"]])
    prim = torch.tensor([G.default_primer(k) for k in key[:, 0]])
    t = lambda k: torch.from_numpy(f[k]).cuda()
    ref = m.generate_batch(t("semantic"), key.cuda(), t("scene_offset"), t("motion"), t("emotion"), prim[:, 0:1], prim[:, 1:2], prim[:, 2:3],
                           target_seq_length=24, beam=0, sampler="argmax")
    assert torch.equal(ref.cpu(), toks)
    # custom primer typed the user's way, and the clips' own first chords (generate.py:286-344, :367-379)
    base = [a for a in argv if a not in ("--regression", "--midi")]
    custom = G.main(base + ["--primer", "C Am Dm G"]).cpu()
    assert custom[:, :4].tolist() == [[C_.CHORD_DIC[n] for n in ("C", "A:min", "D:min", "G")]] * 2
    rows = torch.tensor([C_.primer_from_user_chords(["C", "Am", "Dm", "G"])] * 2)
    ref = m.generate_batch(t("semantic"), key.cuda(), t("scene_offset"), t("motion"), t("emotion"), rows[:, :, 0], rows[:, :, 1], rows[:, :, 2],
                           target_seq_length=24, beam=0, sampler="argmax")
    assert torch.equal(ref.cpu(), custom)
    own = G.main(base + ["--primer_from_dataset", "-num_prime_chord", "5"]).cpu()
    assert np.array_equal(own[:, :5].numpy(), f["chord"][:, :5])
    pr = [torch.from_numpy(f[k][:, :5]).cuda() for k in ("chord", "chord_root", "chord_attr")]
    ref = m.generate_batch(t("semantic"), key.cuda(), t("scene_offset"), t("motion"), t("emotion"), *pr, target_seq_length=24, beam=0, sampler="argmax")
    assert torch.equal(ref.cpu(), own)


def test_bench_line_contract(tmp_path):
    """bench.py prints ONE JSON line with the driver's keys, the roofline object and the cpu_baseline object (small config)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--seq", "48", "--batch", "3",
                          "--layers", "2", "--d_model", "128"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "chord_tokens_per_sec_generated" and d["n_gpus"] == 1 and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and r["avg_launch_us"] > 0 and r["whole_step"]["algorithmic_bytes_per_step"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - 3 * 47 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-2


def test_shared_moe_temperature_scheduler_vs_reference_golden(golden):
    """SharedMoELayer(temperature_scheduler=...): stepped in every forward, eval included, the two routing logits divided by the
    temperature before their softmax (reference model/moe.py:238-240, 288) -- three consecutive calls of the reference layer."""
    from video2music_amd.model.moe import TemperatureScheduler
    g = golden("g_opts.npz")
    sched = TemperatureScheduler(temperature_min=0.7, temperature_max=0.9, temperature_step=0.15)
    layer = SharedMoELayer(GLUExpert(128, 256), 128, n_experts=8, n_experts_per_token=2, balancing=True, temperature_scheduler=sched)
    m, _ = load(layer, moe_shapes(8, 128, 256, True), 5)
    x = torch.from_numpy(g["moe_t_x"]).cuda()
    for c in range(3):
        y = m(x)
        assert abs(sched.getT() - float(g[f"moe_t_t{c}"])) < 1e-9
        err = np.abs(y.cpu().numpy() - g[f"moe_t_y{c}"]).max()
        assert err < 1e-4, (c, err)
