"""Feature-file reader on a miniature on-disk dataset; CPU only."""
import os

import numpy as np
import pytest

from video2music_amd.dataset import vevo_features as V
from video2music_amd.utilities import constants as C


@pytest.fixture()
def root(tmp_path):
    r = str(tmp_path)
    for d in ("vevo_chord/lab_v2_norm/origin", "vevo_semantic/origin/2d/clip_l14p", "vevo_scene_offset/origin",
              "vevo_motion/origin", "vevo_motion/option1", "vevo_emotion/6c_l14p/origin", "vevo_meta/split/v1"):
        os.makedirs(os.path.join(r, d))
    rng = np.random.default_rng(0)
    n = 7
    np.save(os.path.join(r, "vevo_semantic/origin/2d/clip_l14p/001.npy"), rng.standard_normal((n, 768)).astype(np.float32))
    np.save(os.path.join(r, "vevo_semantic/origin/2d/clip_l14p/002.npy"), rng.standard_normal((40, 768)).astype(np.float32))
    np.save(os.path.join(r, "vevo_motion/option1/001.npy"), rng.standard_normal((n, 512)).astype(np.float32))
    np.save(os.path.join(r, "vevo_motion/option1/002.npy"), rng.standard_normal((40, 512)).astype(np.float32))
    for fid, m in (("001", n), ("002", 40)):
        with open(os.path.join(r, f"vevo_scene_offset/origin/{fid}.lab"), "w") as f:
            f.write("".join(f"{t} {t // 3}\n" for t in range(m)))
        with open(os.path.join(r, f"vevo_motion/origin/{fid}.lab"), "w") as f:
            f.write("".join(f"{t} {0.25 * t:.4f}\n" for t in range(m)))
        with open(os.path.join(r, f"vevo_emotion/6c_l14p/origin/{fid}.lab"), "w") as f:
            f.write("time exciting_prob fearful_prob tense_prob sad_prob relaxing_prob neutral_prob\n")
            f.write("".join(f"{t} 0.1 0.2 0.3 0.1 0.2 {0.1 + 0.01 * t:.2f}\n" for t in range(m)))
    with open(os.path.join(r, "vevo_chord/lab_v2_norm/origin/001.lab"), "w") as f:
        f.write("key A minor\n0 N\n1 A:min\n2 A:min\n3 G\n4 C:maj7\n5 F#:hdim7\n")
    with open(os.path.join(r, "vevo_meta/split/v1/test.txt"), "w") as f:
        f.write("001\n002\n")
    return r


def test_load_clip_padding_and_chord_rules(root):
    c = V.load_clip(root, "001", motion_type=1, max_seq_video=12, max_seq_chord=12)
    assert c["semantic"].shape == (12, 768) and not c["semantic"][7:].any() and c["semantic"][:7].any()
    assert c["scene_offset"].tolist() == [1, 1, 1, 2, 2, 2, 3, 0, 0, 0, 0, 0]            # ids stored +1, 0 = pad
    assert c["motion"].shape == (12, 512) and not c["motion"][7:].any()
    assert c["emotion"].shape == (12, 6) and np.allclose(c["emotion"][6], [0.1, 0.2, 0.3, 0.1, 0.2, 0.16]) and not c["emotion"][7:].any()
    assert c["key"].tolist() == [1.0]                                                   # "A minor"
    assert c["chord"][:7].tolist() == [0, C.CHORD_DIC["A:min"], C.CHORD_DIC["A:min"], C.CHORD_DIC["G"], C.CHORD_DIC["C:maj7"],
                                       C.CHORD_DIC["F#:hdim7"], C.CHORD_END]
    assert c["chord_root"][:7].tolist() == [0, 10, 10, 8, 1, 7, C.CHORD_ROOT_END]
    assert c["chord_attr"][:7].tolist() == [0, 5, 5, 1, 13, 10, C.CHORD_ATTR_END]           # plain root -> 1 ("maj"), N -> 0


def test_truncation_follows_max_seq_chord(root):
    """The text streams stop at max_seq_chord and the .npy motion is cut to max_seq_chord rows (reference quirk)."""
    c = V.load_clip(root, "002", motion_type=1, max_seq_video=30, max_seq_chord=20)
    assert c["semantic"].shape == (30, 768) and c["semantic"][29].any()                    # cut to max_seq_video
    assert c["scene_offset"].shape == (30,) and c["scene_offset"][19] == 7 and not c["scene_offset"][20:].any()
    assert c["emotion"][19].any() and not c["emotion"][20:].any()
    assert c["motion"].shape == (20, 512)
    assert (c["chord"] == C.CHORD_PAD).all() and c["chord"].shape == (20,)                 # no chord file for this clip
    c0 = V.load_clip(root, "002", motion_type=0, max_seq_video=30, max_seq_chord=30)
    assert c0["motion"].shape == (30,) and c0["motion"][29] == pytest.approx(7.25)


def test_key_rule_of_the_generate_script():
    e = np.full((5, 6), 0.1, dtype=np.float32)
    e[0, 2] = 0.9
    assert V.key_from_emotion(e) == 1.0             # flat arg-max index 2
    e[3, 2] = 0.95
    assert V.key_from_emotion(e) == 0.0             # a later frame wins: flat index 20 is not in (1, 2, 3)
    e[:] = 0.1
    e[0, 0] = 0.9
    assert V.key_from_emotion(e) == 0.0


def test_load_clips_layout_and_lab_round_trip(root, tmp_path):
    ids = V.read_split(root, "test")
    assert ids == ["001", "002"]
    f = V.load_clips(root, ids, motion_type=1, max_seq_video=10, max_seq_chord=10)
    assert f["semantic"].shape == (2, 10, 768) and f["key"].shape == (2, 1) and f["scene_offset"].shape == (2, 10)
    assert f["motion"].shape == (2, 10, 512) and f["emotion"].shape == (2, 10, 6) and f["chord"].shape == (2, 10)
    out = str(tmp_path / "gen.lab")
    V.write_lab(out, [1, C.CHORD_DIC["A:min"], 0])
    assert open(out).read() == "key ?\n0 C\n1 A:min\n2 N\n"
    chord, root_ids, attr_ids, key, last = V.read_chords(out, 8)
    assert chord[:3].tolist() == [1, C.CHORD_DIC["A:min"], 0] and last == 2 and key == 1.0     # "key ?" has no "major"


def test_reader_equals_reference_dataset_class(golden, tmp_path):
    """g_features.npz: the reference's own VevoDataset.createSample run on a miniature dataset (a 40 s and a 320 s
    clip); the reader must return the same tensors for motion_type 0 and 1."""
    from tests.helpers_features import write_mini_dataset
    g = golden("g_features.npz")
    content = {k[3:]: g[k] for k in g if k.startswith("in_")}
    content["ids"] = [str(i) for i in g["ids"]]
    write_mini_dataset(str(tmp_path), content)
    for mt in (0, 1):
        for fid in content["ids"]:
            c = V.load_clip(str(tmp_path), fid, motion_type=mt, max_seq_video=300, max_seq_chord=300)
            ref = lambda k: g[f"ref_mt{mt}_{fid}_{k}"]
            assert np.array_equal(c["semantic"], ref("semanticList"))
            assert np.array_equal(c["scene_offset"], ref("scene_offset"))
            assert np.array_equal(c["motion"], ref("motion"))
            assert np.array_equal(c["emotion"], ref("emotion"))
            assert np.array_equal(c["key"], ref("key"))
            for k in ("chord", "chord_root", "chord_attr"):
                assert np.array_equal(c[k], ref(k))
            assert np.array_equal(c["chord"][:299], ref("x")) and np.array_equal(c["chord_root"][:299], ref("x_root"))
