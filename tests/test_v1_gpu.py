"""VideoMusicTransformer_V1 family on the HIP operator kernels vs goldens produced by the
reference's own V1 class (oracle/make_goldens_v1.py): every layer plan its version strings select."""
import numpy as np
import pytest
import torch

from video2music_amd import synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V1
from tests.helpers import feats_t

pytestmark = pytest.mark.gpu
CFG = dict(n_layers=4, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300, total_vf_dim=synthetic.total_vf_dim(1))
CASES = [("v10", "1.0", False), ("v11", "1.1", False), ("v12", "1.2", False), ("v13", "1.3", False),
         ("v133", "1.3.3", False), ("v134", "1.3.4", False), ("v11rms", "1.1", True)]


def build(version, rms=False, **over):
    m = VideoMusicTransformer_V1(version_name=version, rms_norm=rms, **dict(CFG, **over)).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return m.cuda()


def gen_kw(f, T=20):
    return dict(feature_semantic_list=f["semantic"], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"],
                feature_motion=f["motion"], feature_emotion=f["emotion"], primer=torch.tensor([1]), primer_root=torch.tensor([1]),
                primer_attr=torch.tensor([0]), target_seq_length=T)


@pytest.mark.parametrize("tag,version,rms", CASES)
def test_v1_vs_reference_golden(golden, tag, version, rms):
    g = golden("g_v1.npz")
    m = build(version, rms)
    assert len(m.state_dict()) == int(g[f"{tag}_n_keys"])
    assert (m._rope_cache is not None) == (version == "1.2")            # '1.2' in '1.2.3': the reference's substring test
    key = np.array([[0.0], [1.0], [0.0]], dtype=np.float32)
    feats = synthetic.synthetic_features(3, seed=1234)
    f = {k: v.cuda() for k, v in feats_t(feats, slice(0, 2), key=key).items()}
    root, attr = torch.from_numpy(g[f"{tag}_root"]), torch.from_numpy(g[f"{tag}_attr"])
    with torch.no_grad():
        y = m(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
    err = np.abs(y.cpu().numpy() - g[f"{tag}_logits"]).max()
    assert y.shape == (2, 12, 159) and err < 1e-3, (tag, err)
    f = {k: v.cuda() for k, v in feats_t(feats, slice(0, 1), key=key).items()}
    with torch.no_grad():
        assert np.array_equal(m.generate(beam=1, **gen_kw(f)).cpu().numpy(), g[f"{tag}_g1"])
        for graph in (True, False):
            assert np.array_equal(m.generate(beam=0, sampler="argmax", use_graph=graph, **gen_kw(f)).cpu().numpy(), g[f"{tag}_g2"])
        assert np.array_equal(m.generate(beam=0, sampler="argmax", use_cache=False, **gen_kw(f)).cpu().numpy(), g[f"{tag}_g2"])


@pytest.mark.parametrize("version,over", [("1.3.3", dict(dim_feedforward=192)), ("1.0", dict(n_layers=2)), ("1.2", dict(max_sequence_chord=64))])
def test_v1_cached_decode_equals_reforward(version, over):
    """Shapes the goldens do not hold: GLU layers narrower than the SiLU experts ('1.3.3' with d_ff != 2d: the cached step
    is then issued operator by operator), a 2-layer model, a positional table shorter than the RoPE cache."""
    m = build(version, **over)
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(1, seed=5)).items()}
    T = 40
    with torch.no_grad():
        a = m.generate(beam=0, sampler="argmax", **gen_kw(f, T))
        b = m.generate(beam=0, sampler="argmax", use_cache=False, **gen_kw(f, T))
    assert torch.equal(a, b) and a.shape == (1, T)
    if version == "1.2":
        with pytest.raises(ValueError):
            m.generate(beam=0, sampler="argmax", **gen_kw(f, 65))


@pytest.mark.parametrize("version", ["1.3", "2.0"])
def test_cli_runs_the_other_families(tmp_path, version):
    """`python -m video2music_amd.generate -music_gen_version 1.x / 2.0`: the script picks the class by the version prefix
    (generate.py:209-238); concurrent and one-at-a-time clips give the same ids."""
    from video2music_amd import generate as G
    base = ["--synthetic", "--n_clips", "3", "-n_layers", "4", "-num_heads", "4", "-d_model", "128", "-dim_feedforward", "256",
            "-target_seq_length_chord", "24", "--sampler", "argmax", "-music_gen_version", version]
    a = G.main(base + ["-output_dir", str(tmp_path / "a"), "--v2_batch", "1"]).cpu()
    b = G.main(base + ["-output_dir", str(tmp_path / "b"), "--v2_batch", "2"]).cpu()
    assert a.shape == (3, 24) and torch.equal(a, b)
    assert (tmp_path / "a" / "clip000_chords.lab").exists()


def _v2(version, **over):
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2
    from tests.helpers import CFG_V2
    m = VideoMusicTransformer_V2(**dict(CFG_V2, version_name=version, **over)).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()}, strict=False)
    return m.cuda()


@pytest.mark.parametrize("make,B,P", [(lambda: _v2("2.2"), 5, 1), (lambda: _v2("2.0"), 3, 3), (lambda: _v2("2.2", chord_embed=True), 2, 1),
                                      (lambda: build("1.0"), 4, 1), (lambda: build("1.1", True), 3, 2), (lambda: build("1.2"), 2, 1)])
def test_lockstep_generate_batch_equals_per_clip_generate(make, B, P):
    """generate_batch (B clips through one captured lockstep step, all experts evaluated on all rows) gives, clip by clip,
    the ids of `generate` on that clip alone -- greedy and top-1 branches, shared and per-clip primers."""
    m = make()
    T = 36
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(B, seed=77)).items()}
    rs = np.random.RandomState(B)
    ids = torch.from_numpy(rs.randint(1, 157, size=(B, P)))
    from video2music_amd.utilities.constants import chord_to_root_attr
    ra = torch.tensor([[chord_to_root_attr(int(i)) for i in row] for row in ids])
    pr = (ids, ra[:, :, 0], ra[:, :, 1])
    args = (f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
    with torch.no_grad():
        for kw in (dict(beam=0, sampler="argmax"), dict(beam=1), dict(beam=0, sampler="argmax", temperature=0.8, max_conseq_N=1, max_conseq_chord=3)):
            got = m.generate_batch(*args, *pr, target_seq_length=T, **kw)
            assert got.shape == (B, T)
            for c in range(B):
                one = m.generate(f["semantic"][c:c + 1], f["key"][c], f["scene_offset"][c:c + 1], f["motion"][c:c + 1], f["emotion"][c:c + 1],
                                 pr[0][c], pr[1][c], pr[2][c], target_seq_length=T, decision="host", **kw)
                assert torch.equal(one[0], got[c]), (kw, c)
        dev_ids = m.generate_batch(*args, *pr, target_seq_length=T, beam=0, sampler="argmax")
        eager = m.generate_batch(*args, *pr, target_seq_length=T, beam=0, sampler="argmax", use_graph=False)
        assert torch.equal(eager, dev_ids)
        # the round-1 loop with the decision on the host gives the same ids as the on-device decision
        assert torch.equal(m.generate_batch(*args, *pr, target_seq_length=T, beam=0, sampler="argmax", decision="host"), dev_ids)
        rnd = m.generate_batch(*args, pr[0][0], pr[1][0], pr[2][0], target_seq_length=T, beam=0)       # shared primer, random draw
        assert rnd.shape == (B, T) and int(rnd[:, P:].min()) >= 1 and int(rnd.max()) < 157
        assert not bool(((rnd[:, 2:] == rnd[:, 1:-1]) & (rnd[:, 1:-1] == rnd[:, :-2]))[:, max(P - 2, 0):].any())   # repeat suppression
        # the device draw is the inverse CDF at the supplied uniforms: same uniforms -> same ids; u = 0 -> the lowest allowed ids
        u = torch.rand(T, B, generator=torch.Generator().manual_seed(3))
        a = m.generate_batch(*args, pr[0][0], pr[1][0], pr[2][0], target_seq_length=T, beam=0, uniforms=u)
        assert torch.equal(a, m.generate_batch(*args, pr[0][0], pr[1][0], pr[2][0], target_seq_length=T, beam=0, uniforms=u))
        lo = m.generate_batch(*args, pr[0][0], pr[1][0], pr[2][0], target_seq_length=T, beam=0, uniforms=torch.zeros(T, B))
        assert set(lo[:, P + 1:].flatten().tolist()) <= {1, 2}
        # one clip through the lockstep step (B = 1, decision on the device) equals the one-call step with the host decision
        one = [a[:1] for a in args]
        assert torch.equal(m.generate_batch(*one, pr[0][0], pr[1][0], pr[2][0], target_seq_length=T, beam=0, sampler="argmax"),
                           m.generate_batch(*one, pr[0][0], pr[1][0], pr[2][0], target_seq_length=T, beam=0, sampler="argmax", decision="host"))


def test_v1_options_vs_reference_golden(golden):
    """dropTokenRate, forward(mask=False), beam=2 / beam_chance=0.5 of the reference V1 class ('1.1') -- tests/golden/g_opts.npz."""
    from tests.test_v2_gpu import check_family_options
    check_family_options(golden, "v11", build("1.1"), lambda **over: build("1.1", **over))


@pytest.mark.parametrize("family,version", [("v2", "2.2"), ("v1", "1.0"), ("v1", "1.1"), ("v1", "1.3.3")])
def test_lockstep_at_bench_width_equals_per_clip_generate(family, version):
    """The lockstep step at the width the V2 bench runs (d_model 512, d_ff 1024, 32 clips): there its wide products take the
    several-tiles-per-workgroup kernels (stacked gate|up matrix, grouped down projections -- with GLU experts and with the V1
    family's Linear -> SiLU -> Linear experts, with and without a shared expert) and the self-attention out-projection rides with
    the folded cross-attention query projection; clip by clip the ids must still equal `generate` on that clip alone (one-call
    step with device-routed experts, separate launches)."""
    from tests.helpers import CFG_V2
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2
    B, T = 32, 20
    wide = dict(n_layers=4, num_heads=8, d_model=512, dim_feedforward=1024)
    if family == "v2":
        m = VideoMusicTransformer_V2(**dict(CFG_V2, version_name=version, **wide)).eval()
    else:
        m = VideoMusicTransformer_V1(version_name=version, **dict(CFG, **wide)).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0, recipe="feedback").items()})
    m = m.cuda()
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(B, seed=78)).items()}
    pr = (torch.tensor([1]), torch.tensor([1]), torch.tensor([0]))
    with torch.no_grad():
        got = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], *pr, target_seq_length=T, beam=0, sampler="argmax")
        assert len(set(got[:, 1:].flatten().tolist())) >= 4
        for c in (0, 13, 31):
            one = m.generate(f["semantic"][c:c + 1], f["key"][c], f["scene_offset"][c:c + 1], f["motion"][c:c + 1], f["emotion"][c:c + 1],
                             *pr, target_seq_length=T, beam=0, sampler="argmax", decision="host")
            assert torch.equal(one[0], got[c]), c
