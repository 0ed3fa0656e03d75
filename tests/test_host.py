"""Host-side logic that needs no GPU: module surface, flags, chord tables, sharding arithmetic, and
that the C-ABI library loads and exports every symbol include/amt_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.helpers import CFG1, CFG2, amt_named_shapes
from video2music_amd import _lib, dist as vdist, synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer
from video2music_amd.utilities import constants as C
from video2music_amd.utilities.argument_generate_funcs import parse_generate_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "amt_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(amt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/amt_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().amt_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define AMT_ABI_VERSION (\d+)", header).group(1))
    # release build: no debug entry points, and no environment variable is read by the library (include/amt_hip.h, conventions)
    import subprocess
    syms = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "amt_debug" not in syms
    assert " U getenv" not in syms and " U secure_getenv" not in syms


def test_bindings_refuse_a_library_of_another_abi_version(monkeypatch):
    import importlib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.AmtError, match="ABI version"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load() is not None


def test_bad_config_is_rejected_without_a_gpu():
    cfg = _lib.AmtConfig(6, 8, 500, 1024, 300, 300, 1287, 32)       # d_model not divisible by heads*... (500/8)
    h = ctypes.c_void_p()
    with pytest.raises(_lib.AmtError, match="amt_create"):
        _lib.call("amt_create", ctypes.byref(cfg), ctypes.byref(h))
    cfg = _lib.AmtConfig(6, 8, 512, 1024, 300, 300, 1287, 257)       # more than 256 clips per decode chain (32 in round 1)
    with pytest.raises(_lib.AmtError, match="max_batch"):
        _lib.call("amt_create", ctypes.byref(cfg), ctypes.byref(h))


@pytest.mark.parametrize("cfg", [CFG1, CFG2])
def test_state_dict_matches_reference_key_set(cfg):
    m = VideoMusicTransformer(**cfg)
    want = dict(amt_named_shapes(**cfg))
    want["positional_encoding.pe"] = (cfg["max_sequence_chord"], 1, cfg["d_model"])
    want["positional_encoding_video.pe"] = (300, 1, cfg["d_model"])
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    n_params = sum(p.numel() for p in m.parameters())
    if cfg is CFG2:
        assert abs(n_params - 33.4e6) < 1.0e6          # ~33 M parameters at config 2 (F=1287)
    assert callable(m.transformer.generate_square_subsequent_mask)
    mask = m.transformer.generate_square_subsequent_mask(3)
    assert mask[0, 1] == float("-inf") and mask[1, 0] == 0


def test_v2_state_dict_matches_reference_key_set():
    from tests.helpers import CFG_V2, v2_named_shapes
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2
    m = VideoMusicTransformer_V2(**CFG_V2)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(v2_named_shapes(**CFG_V2))
    with pytest.raises(NotImplementedError):
        VideoMusicTransformer_V2(version_name="2.3", total_vf_dim=1287)
    # '2.0': learned positional tables instead of RoPE (the default version_name of the class, :317)
    v20 = VideoMusicTransformer_V2(**dict(CFG_V2, version_name="2.0", max_sequence_chord=64))
    extra = set(v20.state_dict()) - set(m.state_dict())
    assert extra == {"positional_embedding.weight", "positional_embedding_video.weight"}
    assert v20.positional_embedding.weight.shape == (64, CFG_V2["d_model"]) and v20._rope_cache is None


def test_v1_state_dict_matches_reference_key_counts(golden):
    """Key counts of the reference's V1 class for every layer plan (recorded by oracle/make_goldens_v1.py); names and shapes
    are exercised by the GPU parity tests, which load procedural weights by name."""
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V1
    g = golden("g_v1.npz")
    cfg = dict(n_layers=4, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300, total_vf_dim=1287)
    for tag, version, rms in (("v10", "1.0", False), ("v11", "1.1", False), ("v12", "1.2", False), ("v13", "1.3", False),
                              ("v133", "1.3.3", False), ("v134", "1.3.4", False), ("v11rms", "1.1", True)):
        m = VideoMusicTransformer_V1(version_name=version, rms_norm=rms, **cfg)
        assert len(m.state_dict()) == int(g[f"{tag}_n_keys"]), tag
    sd = VideoMusicTransformer_V1(version_name="1.0", **cfg).state_dict()
    assert sd["transformer.decoder.layers.0.ff.experts.5.3.weight"].shape == (128, 256)      # Linear(d,2d) -> SiLU -> Linear(2d,d)
    assert "transformer.decoder.layers.0.ff.shared_expert.0.weight" not in sd                 # '1.0': MoELayer, no shared expert


def test_v3_state_dict_matches_reference_key_counts(golden):
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V3
    g = golden("g_v3.npz")
    cfg = dict(n_layers=4, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300, total_vf_dim=1287)
    for tag, version in (("v30", "3.0"), ("v31", "3.1"), ("v32", "3.2")):
        m = VideoMusicTransformer_V3(version_name=version, **cfg)
        assert len(m.state_dict()) == int(g[f"{tag}_n_keys"]), tag
    sd = m.state_dict()
    assert sd["transformer.decoder.layers.3.cross_attn.q_proj.weight"].shape == (256, 128)
    assert sd["transformer.decoder.layers.3.self_attn.subln.weight"].shape == (32,) and sd["transformer.decoder.layers.3.ff.bias"].shape == (6, 1)
    assert "transformer.decoder.layers.0.norm1.bias" not in sd               # RMSNorm
    assert abs(m.transformer.decoder.layers[2].self_attn.lambda_init - (0.8 - 0.6 * np.exp(-0.6))) < 1e-12


def test_unsupported_constructor_options_raise():
    plain = VideoMusicTransformer(rpr=False, **{k: v for k, v in CFG1.items() if k != "rpr"})       # torch's stock decoder layers: no Er
    assert not any(k.endswith(".Er") for k in plain.state_dict())
    assert len(plain.state_dict()) == len(VideoMusicTransformer(**CFG1).state_dict()) - CFG1["n_layers"]
    both = VideoMusicTransformer(total_vf_dim=1286, rpr=True, chord_embed=True, scene_embed=True)   # round 2: built (:926-937)
    assert {"chord_embedding_model.weight", "scene_embedding.weight"} <= set(both.state_dict())


def test_no_cpu_fallback():
    m = VideoMusicTransformer(**CFG1).eval()
    z = torch.zeros(1, 2, dtype=torch.long)
    with pytest.raises(_lib.AmtError, match="no CPU fallback"):
        m(z, z, z, torch.zeros(1, 300, 768), torch.zeros(1), torch.zeros(1, 300), torch.zeros(1, 300, 512), torch.zeros(1, 300, 6))


def test_chord_tables_and_primers():
    assert C.CHORD_SIZE == 159 and C.CHORD_ROOT_SIZE == 15 and C.CHORD_ATTR_SIZE == 16
    assert C.CHORD_DIC["C"] == 1 and C.CHORD_DIC["A:min"] == 122 and C.CHORD_INV_DIC["156"] == "B:maj7"
    assert C.primer_from_name("C") == (1, 1, 0) and C.primer_from_name("A:min") == (122, 10, 5)
    assert C.chord_to_root_attr(0) == (0, 1) and C.chord_to_root_attr(1) == (1, 1) and C.chord_to_root_attr(122) == (10, 5)
    for i in range(1, C.CHORD_END):
        r, a = C.chord_to_root_attr(i)
        assert 1 + 13 * (r - 1) + (a - 1) == i


def test_generate_flags_and_feature_widths():
    args = parse_generate_args([])[0]
    assert (args.target_seq_length_chord, args.beam, args.n_layers, args.num_heads, args.d_model, args.dim_feedforward) == (300, 0, 6, 8, 512, 1024)
    assert (args.max_sequence_video, args.max_sequence_chord, args.motion_type, args.rpr) == (300, 300, 1, True)
    assert args.music_gen_version == "2.2"                  # reference default (argument_generate_funcs.py:82)
    from video2music_amd.generate import total_vf_dim_of, default_primer
    assert total_vf_dim_of(args) == 1287 == synthetic.total_vf_dim(1)
    args.motion_type = 0
    assert total_vf_dim_of(args) == 776 == synthetic.total_vf_dim(0)
    assert default_primer(0.0) == (1, 1, 0) and default_primer(1.0) == (122, 10, 5)


def test_synthetic_recipes_are_reproducible():
    a = synthetic.synthetic_features(2, seed=7)
    b = synthetic.synthetic_features(2, seed=7)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["semantic"].shape == (2, 300, 768) and a["motion"].shape == (2, 300, 512) and a["emotion"].shape == (2, 300, 6)
    assert np.allclose(a["emotion"].sum(-1), 1.0, atol=1e-5) and a["scene_offset"].min() == 0
    assert np.array_equal(synthetic.fill_tensor("Wout.weight", (159, 128), 0), synthetic.fill_tensor("Wout.weight", (159, 128), 0))
    er = synthetic.fill_tensor("transformer.decoder.layers.0.self_attn.Er", (300, 32), 0)
    assert er.min() >= 0 and er.max() < 1
    e = np.zeros((1, 300, 6), dtype=np.float32); e[0, 0, 2] = 1.0
    assert synthetic.feature_key_from_emotion(e)[0] == 1.0       # generate.py:199-205 flat arg-max rule
    e[:] = 0; e[0, 5, 2] = 1.0
    assert synthetic.feature_key_from_emotion(e)[0] == 0.0


def test_shard_bounds_cover_all_clips():
    for n in (1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [vdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_user_primer_chords_follow_the_reference_rules():
    """generate.py:291-312: flats -> sharps, quality split after the root, the five short qualities expanded."""
    want = {"C": "C", "Am": "A:min", "Dm": "D:min", "G": "G", "F#": "F#", "F#m": "F#:min", "Bb": "A#", "Bbm7": "A#:min7",
            "Ebm6": "D#:min6", "CM7": "C:maj7", "GM6": "G:maj6", "Cdim": "C:dim", "C#sus4": "C#:sus4", "Abmaj7": "G#:maj7"}
    for typed, name in want.items():
        assert C.normalise_user_chord(typed) == name, typed
        assert name in C.CHORD_DIC, name
    rows = C.primer_from_user_chords(["C", "Am", "Dm", "G"])                 # the reference's custumPrimer (:53)
    assert rows == [C.primer_from_name(n) for n in ("C", "A:min", "D:min", "G")]
    assert rows[0] == (1, 1, 0) and rows[1] == (122, 10, 5)                   # plain roots carry attr 0 in a primer (:315-318)
    a = parse_generate_args(["--primer", "C Am"])[0]
    assert a.primer == "C Am" and a.num_prime_chord == 30 and not a.primer_from_dataset


def test_moe_schedulers_follow_the_reference():
    """moe.py:160-179: MoELayer's schedulers act in training only (kept, no effect here); SharedMoELayer's temperature
    scheduler steps in eval too and divides the routing logits (:238-240,288)."""
    from video2music_amd.model.moe import GLUExpert, MoELayer, SharedMoELayer
    sched = object()
    a = MoELayer(GLUExpert(16, 32), 16, topk_scheduler=sched, temperature_scheduler=sched)
    assert a.topk_scheduler is sched and a.temperature_scheduler is sched
    b = SharedMoELayer(GLUExpert(16, 32), 16, topk_scheduler=sched)
    assert b.topk_scheduler is sched
    from video2music_amd.model.moe import TemperatureScheduler, TopKScheduler
    ts = TemperatureScheduler(temperature_min=0.7, temperature_max=0.9, temperature_step=0.15)
    c = SharedMoELayer(GLUExpert(16, 32), 16, temperature_scheduler=ts)
    assert [round(c._temperature(), 6) for _ in range(3)] == [0.85, 0.9, 0.9] and b._temperature() == 1.0 and a._temperature() == 1.0
    k = TopKScheduler(n_experts=4, min_n_experts_per_token=2, update_step=2)
    ks = []
    for _ in range(6):
        k.step(); ks.append(k.getK())
    assert ks == [4, 3, 3, 2, 2, 2]


def test_v2_builds_three_shallow_layers_whatever_n_layers_says():
    """model/video_music_transformer.py:411-416 of the reference: `rate = 3` shallow layers + (n_layers - 3) deep ones, so n_layers = 2
    still gives three (117 state_dict entries for this configuration, counted on the reference class)."""
    from video2music_amd import synthetic
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2, VideoMusicTransformer_V3
    m = VideoMusicTransformer_V2(version_name="2.2", n_layers=2, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300,
                                 total_vf_dim=synthetic.total_vf_dim(1))
    assert len(m.transformer.decoder.layers) == 3 and len(m.transformer.encoder.layers) == 3 and len(m.state_dict()) == 117
    with pytest.raises(IndexError):
        VideoMusicTransformer_V3(version_name="3.1", n_layers=2, num_heads=4, d_model=128, dim_feedforward=256,
                                 total_vf_dim=synthetic.total_vf_dim(1))
