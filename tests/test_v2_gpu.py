"""VideoMusicTransformer_V2 '2.2'  on the HIP operator kernels vs goldens produced by the
reference's own V2 class, and vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import amt_oracle as O
from video2music_amd import synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2
from tests.helpers import CFG_V2, feats_t, synthetic_sd_v2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def v2():
    m = VideoMusicTransformer_V2(**CFG_V2).eval()
    sd = synthetic_sd_v2(CFG_V2)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return m.cuda(), sd


@pytest.mark.parametrize("B,L", [(1, 1), (1, 12), (2, 1), (2, 12)])
def test_v2_forward_vs_reference_golden(golden, v2, B, L):
    m, _ = v2
    g = golden("g_v2_cfg1.npz")
    key = golden("g_fwd_cfg1.npz")["key"]
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(3, seed=1234), slice(0, B), key=key).items()}
    root, attr = torch.from_numpy(g[f"root_B{B}_L{L}"]), torch.from_numpy(g[f"attr_B{B}_L{L}"])
    with torch.no_grad():
        y = m(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
    err = np.abs(y.cpu().numpy() - g[f"logits_B{B}_L{L}"]).max()
    assert y.shape == (B, L, 159) and err < 1e-3, err


def test_v2_generate_vs_reference_golden(golden, v2):
    m, _ = v2
    g = golden("g_v2_cfg1.npz")
    key = golden("g_fwd_cfg1.npz")["key"]
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(3, seed=1234), slice(0, 1), key=key).items()}
    kw = dict(feature_semantic_list=f["semantic"], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"],
              feature_motion=f["motion"], feature_emotion=f["emotion"], primer=torch.tensor([1]), primer_root=torch.tensor([1]),
              primer_attr=torch.tensor([0]), target_seq_length=24)
    with torch.no_grad():
        assert np.array_equal(m.generate(beam=1, **kw).cpu().numpy(), g["g1"])
        assert np.array_equal(m.generate(beam=0, sampler="argmax", **kw).cpu().numpy(), g["g2"])
        # the round-1 loop (one-call step with device-routed experts, decision on the host) stays available
        assert np.array_equal(m.generate(beam=1, decision="host", **kw).cpu().numpy(), g["g1"])
        assert np.array_equal(m.generate(beam=0, sampler="argmax", decision="host", **kw).cpu().numpy(), g["g2"])


def test_v2_longer_sequence_vs_oracle(v2):
    """L = 150 chords over 300 frames, B = 2 (the RoPE band mapping mixes heads and clips for B > 1)."""
    m, sd = v2
    fc = feats_t(synthetic.synthetic_features(2, seed=17))
    rs = np.random.RandomState(5)
    root = torch.from_numpy(rs.randint(0, 13, size=(2, 150)))
    attr = torch.from_numpy(rs.randint(0, 14, size=(2, 150)))
    ref = O.forward_v2(sd, 4, root, attr, fc["semantic"], fc["key"], fc["scene_offset"], fc["motion"], fc["emotion"])
    f = {k: v.cuda() for k, v in fc.items()}
    with torch.no_grad():
        y = m(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
    err = (y.cpu() - ref).abs().max().item()
    assert err < 1e-3, err


def test_v2_cached_decode_equals_per_step_reforward(v2):
    """generate over cached K/V (one token per step) gives the ids of the per-step re-forward of the decoder stack
    (the reference's loop), for a 3-chord primer and 120 tokens through the GLU and the MoE layers, and the step
    logits equal the last row of the teacher-forced forward."""
    m, _ = v2
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(1, seed=21)).items()}
    kw = dict(feature_semantic_list=f["semantic"], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"],
              feature_motion=f["motion"], feature_emotion=f["emotion"], primer=torch.tensor([1, 66, 122]),
              primer_root=torch.tensor([1, 6, 10]), primer_attr=torch.tensor([0, 0, 5]), target_seq_length=120, beam=0, sampler="argmax")
    with torch.no_grad():
        a = m.generate(use_cache=True, **kw)
        b = m.generate(use_cache=False, **kw)
    assert a.shape == (1, 120) and torch.equal(a, b)
    # step logits vs the full forward over the generated prefix
    from video2music_amd.utilities.constants import chord_to_root_attr
    ids = a[0].cpu()
    ra = [chord_to_root_attr(int(t)) for t in ids]
    root = torch.tensor([[1, 6, 10] + [r for r, _ in ra[3:]]])
    attr = torch.tensor([[0, 0, 5] + [x for _, x in ra[3:]]])
    with torch.no_grad():
        full = m(ids[None], root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])[0]
        memory, B, S = m._encode_memory(f["semantic"], f["scene_offset"], f["motion"], f["emotion"])
        st, st2 = m._cache_init(memory, S), m._cache_init(memory, S)
        key = f["key"].reshape(-1)[:1].contiguous()
        for t in range(119):
            row = m._decode_step(root[:, t:t + 1].cuda(), attr[:, t:t + 1].cuda(), key, t, st)
            assert (row - full[t]).abs().max().item() < 1e-5, t
            # the same step issued by one library call (amt_v2_step) over packed weights on the skinny GEMM, the two
            # routed experts picked on the device: same values up to the summation order of the projections
            nat = m._decode_step_native(int(root[0, t]), int(attr[0, t]), float(key[0]), t, st2)
            assert (nat - row).abs().max().item() < 2e-5, t


def test_cli_decodes_several_v2_clips_concurrently(tmp_path):
    """`python -m video2music_amd.generate` with the default V2 model: clips decoded together in lockstep give the ids of the
    one-at-a-time run."""
    from video2music_amd import generate as G
    base = ["--synthetic", "--n_clips", "6", "-n_layers", "4", "-num_heads", "4", "-d_model", "128", "-dim_feedforward", "256",
            "-target_seq_length_chord", "40", "--sampler", "argmax", "-music_gen_version", "2.2"]
    a = G.main(base + ["-output_dir", str(tmp_path / "a"), "--v2_batch", "1"]).cpu()
    b = G.main(base + ["-output_dir", str(tmp_path / "b"), "--v2_batch", "4"]).cpu()
    assert a.shape == (6, 40) and torch.equal(a, b)


# ---- the other V2 variants: '2.0' (learned positional tables, no RoPE), '2.1', and chord_embed=True -----------------
def _variant(version, chord_embed=False, **extra):
    cfg = dict(CFG_V2, version_name=version, chord_embed=chord_embed, **extra)
    m = VideoMusicTransformer_V2(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return m.cuda()


@pytest.mark.parametrize("tag,version,ce", [("v20", "2.0", False), ("v21", "2.1", False), ("v22ce", "2.2", True), ("v22se", "2.2", False)])
def test_v2_variants_vs_reference_golden(golden, tag, version, ce):
    g = golden("g_v2_variants.npz")
    extra = dict(scene_embed=True, total_vf_dim=CFG_V2["total_vf_dim"] - 1) if tag == "v22se" else {}
    m = _variant(version, ce, rms_norm=(version == "2.1"), **extra)          # rms_norm has no effect in V2, as in the reference
    assert ("scene_embedding.weight" in m.state_dict()) == (tag == "v22se")
    assert ("positional_embedding.weight" in m.state_dict()) == (version == "2.0")
    assert ("chord_embedding_model.weight" in m.state_dict()) == ce
    key = np.array([[0.0], [1.0], [0.0]], dtype=np.float32)
    feats = synthetic.synthetic_features(3, seed=1234)
    for B in (1, 2):
        f = {k: v.cuda() for k, v in feats_t(feats, slice(0, B), key=key).items()}
        x, root, attr = (torch.from_numpy(g[f"{tag}_{n}_B{B}"]) for n in ("x", "root", "attr"))
        with torch.no_grad():
            y = m(x, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
        err = np.abs(y.cpu().numpy() - g[f"{tag}_logits_B{B}"]).max()
        assert err < 1e-3, (tag, B, err)
    f = {k: v.cuda() for k, v in feats_t(feats, slice(0, 1), key=key).items()}
    kw = dict(feature_semantic_list=f["semantic"], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"],
              feature_motion=f["motion"], feature_emotion=f["emotion"], primer=torch.tensor([1]), primer_root=torch.tensor([1]),
              primer_attr=torch.tensor([0]), target_seq_length=24)
    with torch.no_grad():
        assert np.array_equal(m.generate(beam=1, **kw).cpu().numpy(), g[f"{tag}_g1"])
        for graph in (True, False):
            assert np.array_equal(m.generate(beam=0, sampler="argmax", use_graph=graph, **kw).cpu().numpy(), g[f"{tag}_g2"])
        assert np.array_equal(m.generate(beam=0, sampler="argmax", temperature=0.7, **kw).cpu().numpy(), g[f"{tag}_g2_t"])
        assert np.array_equal(m.generate(beam=0, sampler="argmax", use_cache=False, **kw).cpu().numpy(), g[f"{tag}_g2"])


def test_v2_version_21_equals_22_in_eval(golden):
    assert np.array_equal(golden("g_v2_variants.npz")["v21_g2"], golden("g_v2_cfg1.npz")["g2"])


def test_v2_chord_table_rows_follow_the_state_dict():
    """The frozen chord table comes from a Word2Vec file in the reference; its row count is whatever that file holds."""
    m = VideoMusicTransformer_V2(**dict(CFG_V2, chord_embed=True)).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["chord_embedding_model.weight"] = torch.randn(170, CFG_V2["d_model"])
    m.load_state_dict(sd)
    assert m.chord_embedding_model.weight.shape == (170, CFG_V2["d_model"]) and not m.chord_embedding_model.weight.requires_grad


def test_v2_unbuilt_variants_say_so():
    with pytest.raises(NotImplementedError):
        VideoMusicTransformer_V2(**dict(CFG_V2, version_name="2.3"))


@pytest.mark.parametrize("clip", [0, 1])
def test_v2_well_conditioned_generate_vs_reference_golden(golden, clip):
    """g_v2_hi.npz: the reference's V2 class with the "feedback" weight recipe — every greedy decision has a top-1 / top-2 margin
    >= 1e-2 (the round-1 fixture's minimum is 4.2e-4): G2 at temperature 1.0 and 0.8 (with N allowed and 3-fold repeat
    suppression), G1, and the forward logits along the generated sequence; also through the lockstep batch (device decision)."""
    g = golden("g_v2_hi.npz")
    for name in ("t10", "t08"):
        assert g[f"g2_{name}_margins_clip{clip}"].min() >= 1e-2
    m = VideoMusicTransformer_V2(**CFG_V2).eval()
    sd = synthetic_sd_v2(CFG_V2, seed=int(g["seed"]), recipe="feedback")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    m = m.cuda()
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(3, seed=1234), slice(clip, clip + 1), key=g["key"]).items()}
    pr, prr, pra = (torch.tensor([int(v)]) for v in g[f"primer_clip{clip}"])
    kw = dict(feature_semantic_list=f["semantic"], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"],
              feature_motion=f["motion"], feature_emotion=f["emotion"], primer=pr, primer_root=prr, primer_attr=pra, target_seq_length=48)
    with torch.no_grad():
        assert np.array_equal(m.generate(beam=0, sampler="argmax", **kw).cpu().numpy(), g[f"g2_t10_clip{clip}"])
        assert np.array_equal(m.generate(beam=0, sampler="argmax", temperature=0.8, max_conseq_N=1, max_conseq_chord=3, **kw).cpu().numpy(),
                              g[f"g2_t08_clip{clip}"])
        assert np.array_equal(m.generate(beam=1, **kw).cpu().numpy(), g[f"g1_clip{clip}"])
        # two copies of the clip through the lockstep step with the decision on the device
        f2 = {k: torch.cat([v, v]) for k, v in f.items()}
        both = m.generate_batch(f2["semantic"], f2["key"], f2["scene_offset"], f2["motion"], f2["emotion"], pr, prr, pra,
                                target_seq_length=48, beam=0, sampler="argmax", temperature=0.8, max_conseq_N=1, max_conseq_chord=3)
        assert np.array_equal(both[0].cpu().numpy(), g[f"g2_t08_clip{clip}"][0]) and torch.equal(both[0], both[1])
        if clip == 0:
            root, attr = torch.from_numpy(g["fwd_root"]), torch.from_numpy(g["fwd_attr"])
            y = m(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
            assert np.abs(y.cpu().numpy() - g["fwd_logits"]).max() < 1e-3


# ---------------- rarely used options (tests/golden/g_opts.npz, oracle/make_goldens_opts.py; VERDICT r1 missing #5) ----------------

def check_family_options(golden, tag, m, mid):
    """dropTokenRate (forward under torch.manual_seed(5), generate under torch.manual_seed(9): a fresh mask per step), forward
    (mask=False), generate(beam=2, beam_chance=0.5) with python's `random` seeded -- against the reference class's own outputs."""
    import random
    g = golden("g_opts.npz")
    key = g["key"]
    feats = synthetic.synthetic_features(3, seed=1234)
    f = {k: v.cuda() for k, v in feats_t(feats, slice(0, 2), key=key).items()}
    root, attr = torch.from_numpy(g["fam_root"]), torch.from_numpy(g["fam_attr"])
    f1 = {k: v.cuda() for k, v in feats_t(feats, slice(0, 1), key=key).items()}
    gkw = dict(feature_semantic_list=f1["semantic"], feature_key=f1["key"][0], feature_scene_offset=f1["scene_offset"], feature_motion=f1["motion"],
               feature_emotion=f1["emotion"], primer=torch.tensor([1]), primer_root=torch.tensor([1]), primer_attr=torch.tensor([0]))
    with torch.no_grad():
        y = m(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], mask=False)
        assert np.abs(y.cpu().numpy() - g[f"{tag}_nomask_logits"]).max() < 1e-3
        random.seed(13)
        ids = m.generate(beam=2, beam_chance=0.5, target_seq_length=24, sampler="argmax", **gkw)
        assert ids.shape == (2, 24) and np.array_equal(ids.cpu().numpy(), g[f"{tag}_beam2_c05"])
        random.seed(13)
        rows = m.generate_batch(f1["semantic"], f1["key"], f1["scene_offset"], f1["motion"], f1["emotion"], torch.tensor([1]), torch.tensor([1]),
                                torch.tensor([0]), target_seq_length=24, beam=2, beam_chance=0.5, sampler="argmax")
        assert np.array_equal(rows.cpu().numpy(), g[f"{tag}_beam2_c05"][:1])
        md = mid(dropTokenRate=0.3)
        torch.manual_seed(5)
        y = md(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
        assert np.abs(y.cpu().numpy() - g[f"{tag}_drop_logits"]).max() < 1e-3
        torch.manual_seed(9)
        ids = md.generate(beam=0, target_seq_length=14, sampler="argmax", **gkw)
        assert np.array_equal(ids.cpu().numpy(), g[f"{tag}_drop_g2"])
        torch.manual_seed(9)
        rows = md.generate_batch(f1["semantic"], f1["key"], f1["scene_offset"], f1["motion"], f1["emotion"], torch.tensor([1]), torch.tensor([1]),
                                 torch.tensor([0]), target_seq_length=14, beam=0, sampler="argmax")
        assert np.array_equal(rows.cpu().numpy(), g[f"{tag}_drop_g2"])


@pytest.mark.parametrize("tag,version", [("v22", "2.2"), ("v20", "2.0")])
def test_v2_options_vs_reference_golden(golden, tag, version):
    def mid(**over):
        cfg = dict(CFG_V2, version_name=version, n_layers=4, **over)
        m = VideoMusicTransformer_V2(**cfg).eval()
        shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()})
        return m.cuda()
    check_family_options(golden, tag, mid(), mid)


def test_v2_at_bench_width_vs_reference_golden(golden):
    """The reference VideoMusicTransformer_V2('2.2') itself at d_model 512 / 8 heads / d_ff 1024 / 6 layers (tests/golden/
    g_v2_wide.npz, oracle/make_goldens_v2_wide.py): forward logits, G1 and G2 ids -- per clip, and with the clip as row 5 of a
    32-clip lockstep batch, where the step takes its wide-product kernels and the folded out-projection + query projection."""
    g = golden("g_v2_wide.npz")
    cfg = dict(CFG_V2, version_name="2.2", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024)
    m = VideoMusicTransformer_V2(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=int(g["seed"]), recipe="feedback").items()})
    m = m.cuda()
    assert g["g2_margins"].min() >= 1e-2 and len(set(g["g2"].flatten().tolist())) >= 8
    feats = synthetic.synthetic_features(2, seed=4321)
    f = {k: v.cuda() for k, v in feats_t(feats, slice(0, 1), key=g["key"]).items()}
    T = g["g2"].shape[1]
    pr = tuple(torch.tensor([int(v)]) for v in g["primer"])
    with torch.no_grad():
        root, attr = torch.from_numpy(g["fwd_root"]), torch.from_numpy(g["fwd_attr"])
        y = m(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
        assert np.abs(y.cpu().numpy() - g["fwd_logits"]).max() < 1e-3           # logits of magnitude ~130
        kw = dict(feature_semantic_list=f["semantic"], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"], feature_motion=f["motion"],
                  feature_emotion=f["emotion"], primer=pr[0], primer_root=pr[1], primer_attr=pr[2], target_seq_length=T)
        assert np.array_equal(m.generate(beam=0, sampler="argmax", **kw).cpu().numpy(), g["g2"])
        assert np.array_equal(m.generate(beam=1, **kw).cpu().numpy(), g["g1"])
        big = synthetic.synthetic_features(32, seed=99)
        for k in big:
            big[k][5] = feats[k][0]
        key32 = big["key"].copy()
        key32[5] = g["key"][0]
        fb = {k: v.cuda() for k, v in feats_t(big, key=key32).items()}
        toks = m.generate_batch(fb["semantic"], fb["key"], fb["scene_offset"], fb["motion"], fb["emotion"], *pr, target_seq_length=T,
                                beam=0, sampler="argmax")
        assert np.array_equal(toks[5].cpu().numpy(), g["g2"][0])


@pytest.mark.parametrize("B,clips", [(24, (3, 20)), (40, (17, 39))])
def test_lockstep_wide_gate_up_kernels_every_row_block(B, clips):
    """At d_model 512 the stacked gate | up product of a mixture layer (7 x 2 x dff columns >= 4096) takes the wide skinny-GEMM kernels -- both
    16-row blocks in one workgroup for 17 ... 32 clips, one row block per workgroup otherwise -- whose epilogue writes up * silu(gate) from
    the interleaved tiles (round 3).  Clips of the first, second and third row block give the ids of `generate` on that clip alone."""
    cfg = dict(CFG_V2, version_name="2.2", n_layers=2, num_heads=8, d_model=512, dim_feedforward=512, max_sequence_chord=40)
    m = VideoMusicTransformer_V2(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=5, recipe="feedback").items()})
    m = m.cuda()
    f = {k: v.cuda() for k, v in feats_t(synthetic.synthetic_features(B, seed=B)).items()}
    pr = tuple(torch.tensor([v]) for v in (1, 1, 0))
    T = 16
    with torch.no_grad():
        got = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], *pr, target_seq_length=T, beam=0, sampler="argmax")
        assert got.shape == (B, T) and len(set(got[:, 1:].flatten().tolist())) >= 4
        for c in clips:
            one = m.generate(f["semantic"][c:c + 1], f["key"][c], f["scene_offset"][c:c + 1], f["motion"][c:c + 1], f["emotion"][c:c + 1],
                             pr[0], pr[1], pr[2], target_seq_length=T, beam=0, sampler="argmax", decision="host")
            assert torch.equal(one[0], got[c]), (c, one[0], got[c])


@pytest.mark.parametrize("d,H,ff,nl", [(256, 4, 64, 4), (512, 4, 192, 3), (32, 1, 320, 3)])     # d_model 32: outside the step kernels' widths
def test_v2_feed_forward_narrower_than_half_the_model_width(d, H, ff, nl):
    """2 * dim_feedforward < d_model: the stacked gate | up product of a GLU layer then has fewer column tiles than its LayerNorm
    prologue has 16-column blocks to publish as the block's residual (found by tests/fuzz_parity.py v2: columns past 2 * d_ff of the
    residual were never written).  Lockstep generate_batch and the one-clip generate against the oracle, clip by clip.  A model width
    that is not a multiple of 64 takes the cached step operator by operator instead of being refused at generate time."""
    cfg = dict(CFG_V2, version_name="2.2", n_layers=nl, num_heads=H, d_model=d, dim_feedforward=ff)
    m = VideoMusicTransformer_V2(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=d + ff, recipe="feedback").items()}
    m.load_state_dict(sd)
    m = m.cuda()
    fc = feats_t(synthetic.synthetic_features(3, seed=ff))
    f = {k: v.cuda() for k, v in fc.items()}
    pr = (torch.tensor([1]), torch.tensor([1]), torch.tensor([0]))
    T = 10
    with torch.no_grad():
        toks = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], *pr, target_seq_length=T,
                                beam=0, sampler="argmax").cpu()
        one = m.generate(feature_semantic_list=f["semantic"][:1], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"][:1],
                         feature_motion=f["motion"][:1], feature_emotion=f["emotion"][:1], primer=pr[0], primer_root=pr[1], primer_attr=pr[2],
                         target_seq_length=T, beam=0, sampler="argmax").cpu()
    for b in range(3):
        c = {k: v[b:b + 1] for k, v in fc.items()}
        margins = []
        ref = O.generate(sd, H, c["semantic"], c["key"], c["scene_offset"], c["motion"], c["emotion"], *pr, target_seq_length=T, beam=0,
                         forward_fn=O.forward_v2, margins=margins)
        assert min(margins) > 1e-3, margins
        assert torch.equal(toks[b:b + 1], ref), (b, toks[b], ref)
        if b == 0:
            assert torch.equal(one, ref)


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("d,H,ff,nl,version", [(256, 4, 256, 4, "2.2"), (256, 4, 512, 4, "2.2"), (512, 8, 512, 4, "2.2"), (512, 8, 1024, 4, "2.0")])
def test_v2_plain_layers_with_folded_norms(monkeypatch, d, H, ff, nl, version, mode):
    """The lockstep step of a plain GLU layer with norm3 folded through the next layer's QKV projection (mode 1, the default: the down
    projection takes rows [up * silu(gate) | x] and emits the raw QKV product, the next self-attention finishes q / k / v with the row
    statistics, rotates q and the new key -- attention prologue 4; '2.0' has no rotation and takes prologue 2) and, mode 2, norm2 through
    the stacked gate | up product as well (skinny-GEMM prologue 5 at K = 512 / 768 / 1024 / 1536): ids equal to the chain with the
    folds switched off and, for '2.2', to the oracle clip by clip."""
    cfg = dict(CFG_V2, version_name=version, n_layers=nl, num_heads=H, d_model=d, dim_feedforward=ff)
    fc = feats_t(synthetic.synthetic_features(3, seed=d + ff))
    f = {k: v.cuda() for k, v in fc.items()}
    pr = (torch.tensor([1]), torch.tensor([1]), torch.tensor([0]))
    T = 14
    outs = {}
    for fold in (mode, "0"):
        monkeypatch.setenv("AMT_V2_FOLD_FFN", fold)
        m = VideoMusicTransformer_V2(**cfg).eval()
        shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=d + ff, recipe="feedback").items()}
        m.load_state_dict(sd)
        m = m.cuda()
        with torch.no_grad():
            outs[fold] = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], *pr, target_seq_length=T,
                                          beam=0, sampler="argmax").cpu()
            if fold != "0":
                st = m._cache_init([torch.zeros(300, d, device="cuda")], 300)
                n_qkv = sum(1 for li in range(nl) if st["tab"][11 + 48 * li + 44] is not None)
                n_gu = sum(1 for li in range(nl) if st["tab"][11 + 48 * li + 40] is not None)
                assert (n_qkv, n_gu) == (3, 3 if mode == "2" else 0), (n_qkv, n_gu)      # the three shallow GLU layers carry the folded entries
    assert torch.equal(outs[mode], outs["0"])
    assert len(set(outs[mode][:, 1:].flatten().tolist())) >= 4
    if version != "2.2":
        return                      # (the oracle's V2 forward is the rotary one; '2.0' is pinned by its own goldens)
    for b in range(3):
        c = {k: v[b:b + 1] for k, v in fc.items()}
        margins = []
        ref = O.generate(sd, H, c["semantic"], c["key"], c["scene_offset"], c["motion"], c["emotion"], *pr, target_seq_length=T, beam=0,
                         forward_fn=O.forward_v2, margins=margins)
        assert min(margins) > 1e-3, margins
        assert torch.equal(outs[mode][b:b + 1], ref), (b, outs[mode][b], ref)
