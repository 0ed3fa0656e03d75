"""VideoMusicTransformer_V3 (differential attention, RMSNorm, pre-norm for '3.2') on the HIP
operator kernels vs goldens produced by the reference's own V3 class (oracle/make_goldens_v3.py)."""
import numpy as np
import pytest
import torch

from video2music_amd import ops, synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V3
from tests.helpers import feats_t

pytestmark = pytest.mark.gpu
CFG = dict(n_layers=4, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300, total_vf_dim=synthetic.total_vf_dim(1))


def build(version, **over):
    m = VideoMusicTransformer_V3(version_name=version, **dict(CFG, **over)).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return m.cuda()


@pytest.mark.parametrize("tag,version", [("v30", "3.0"), ("v31", "3.1"), ("v32", "3.2")])
def test_v3_vs_reference_golden(golden, tag, version):
    g = golden("g_v3.npz")
    m = build(version)
    assert len(m.state_dict()) == int(g[f"{tag}_n_keys"])
    key = np.array([[0.0], [1.0], [0.0]], dtype=np.float32)
    feats = synthetic.synthetic_features(3, seed=1234)
    for sl, suffix in ((slice(0, 2), ""), (slice(2, 3), "1")):          # B = 2 (clips and positions mix), B = 1 with L = 24
        f = {k: v.cuda() for k, v in feats_t(feats, sl, key=key).items()}
        root, attr = torch.from_numpy(g[f"{tag}_root{suffix}"]), torch.from_numpy(g[f"{tag}_attr{suffix}"])
        with torch.no_grad():
            y = m(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
        err = np.abs(y.cpu().numpy() - g[f"{tag}_logits{suffix}"]).max()
        assert err < 1e-3, (tag, suffix, err)
    f = {k: v.cuda() for k, v in feats_t(feats, slice(0, 1), key=key).items()}
    kw = dict(feature_semantic_list=f["semantic"], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"],
              feature_motion=f["motion"], feature_emotion=f["emotion"], primer=torch.tensor([1]), primer_root=torch.tensor([1]),
              primer_attr=torch.tensor([0]), target_seq_length=16)
    with torch.no_grad():
        assert np.array_equal(m.generate(beam=1, **kw).cpu().numpy(), g[f"{tag}_g1"])
        assert np.array_equal(m.generate(beam=0, sampler="argmax", **kw).cpu().numpy(), g[f"{tag}_g2"])
        with pytest.raises(NotImplementedError):
            m.generate(beam=0, use_cache=True, **kw)


@pytest.mark.parametrize("hd", [32, 64, 128])
def test_diff_subln_kernel(hd):
    rs = np.random.RandomState(hd)
    o1, o2 = (torch.from_numpy(rs.standard_normal((3, 5, 7, hd)).astype(np.float32)) for _ in range(2))
    w = torch.from_numpy(rs.uniform(0.5, 1.5, hd).astype(np.float32))
    lam, scale = 0.37, 0.8
    d = (o1 - lam * o2).double()
    ref = (d * torch.rsqrt(d.pow(2).mean(-1, keepdim=True) + 1e-5) * w.double() * scale).float()
    y = ops.diff_subln(o1.cuda(), o2.cuda(), w.cuda(), lam, scale, eps=1e-5).cpu()
    assert (y - ref).abs().max().item() < 2e-6
    a, b = torch.randn(1000, 12), torch.randn(1000, 12)
    assert torch.equal(ops.add(a.cuda(), b.cuda()).cpu(), a + b)


def test_cli_runs_v3(tmp_path):
    from video2music_amd import generate as G
    argv = ["--synthetic", "--n_clips", "2", "-n_layers", "4", "-num_heads", "4", "-d_model", "128", "-dim_feedforward", "256",
            "-target_seq_length_chord", "12", "--sampler", "argmax", "-music_gen_version", "3.1", "-output_dir", str(tmp_path)]
    a = G.main(argv).cpu()
    assert a.shape == (2, 12) and int(a[:, 1:].min()) >= 1


@pytest.mark.parametrize("version,B,P", [("3.0", 3, 1), ("3.1", 4, 2), ("3.2", 2, 1)])
def test_v3_generate_batch_equals_per_clip_generate(version, B, P):
    """B clips re-forwarded together each step (each as a batch of one) give, clip by clip, the ids of `generate`."""
    m = build(version)
    T = 20
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(B, seed=31)).items()}
    rs = np.random.RandomState(B)
    ids = torch.from_numpy(rs.randint(1, 157, size=(B, P)))
    from video2music_amd.utilities.constants import chord_to_root_attr
    ra = torch.tensor([[chord_to_root_attr(int(i)) for i in row] for row in ids])
    pr = (ids, ra[:, :, 0], ra[:, :, 1])
    args = (f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
    with torch.no_grad():
        for kw in (dict(beam=0, sampler="argmax"), dict(beam=1), dict(beam=0, sampler="argmax", temperature=0.8)):
            got = m.generate_batch(*args, *pr, target_seq_length=T, **kw)
            assert got.shape == (B, T)
            for c in range(B):
                one = m.generate(f["semantic"][c:c + 1], f["key"][c], f["scene_offset"][c:c + 1], f["motion"][c:c + 1], f["emotion"][c:c + 1],
                                 pr[0][c], pr[1][c], pr[2][c], target_seq_length=T, **kw)
                assert torch.equal(one[0], got[c]), (kw, c)


def test_v3_options_vs_reference_golden(golden):
    """dropTokenRate, forward(mask=False), beam=2 / beam_chance=0.5 of the reference V3 class ('3.0') -- tests/golden/g_opts.npz."""
    from tests.test_v2_gpu import check_family_options
    check_family_options(golden, "v30", build("3.0"), lambda **over: build("3.0", **over))
