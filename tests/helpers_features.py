"""A miniature MuVi-Sync dataset for the feature-reader tests (test infrastructure)."""
import json
import os

import numpy as np

from video2music_amd.utilities import constants as C

CHORDS = ["N", "C", "A:min", "G", "F:maj7", "D:min7", "E:7", "B:hdim7", "F#:dim", "G#:sus4", "A#:maj6", "C#:aug", "D#:sus2"]


def mini_dataset_content(seed=7):
    """Clip "003": 40 s (shorter than every limit); clip "017": 320 s (longer than the 300-row limits)."""
    rng = np.random.default_rng(seed)
    c = {"ids": ["003", "017"]}
    for fid, n, key in (("003", 40, "A minor"), ("017", 320, "G major")):
        c[f"{fid}_semantic"] = (rng.integers(-64, 64, size=(n, 24)) / 16).astype(np.float32)      # any width is read as is
        c[f"{fid}_motion1"] = (rng.integers(-8, 8, size=(n, 512)) / 4).astype(np.float32)         # few distinct values: small fixture
        c[f"{fid}_motion0"] = np.round(rng.random(n) * 50, 4)
        c[f"{fid}_scene"] = np.cumsum(rng.random(n) < 0.2).astype(np.int64)
        e = rng.random((n, 6)).astype(np.float64)
        c[f"{fid}_emotion"] = np.round(e / e.sum(1, keepdims=True), 4)
        c[f"{fid}_chords"] = rng.integers(0, len(CHORDS), size=n).astype(np.int64)
        c[f"{fid}_key"] = np.array(key)
    return c


def write_mini_dataset(root, c, with_targets=False):
    """Writes the files of `mini_dataset_content` in the reference's directory layout.  with_targets also writes the
    regression-target files and the un-normalised chord file the reference's dataset class insists on."""
    def d(*p):
        path = os.path.join(root, *p)
        os.makedirs(path, exist_ok=True)
        return path
    meta = d("vevo_meta")
    for name, table in (("chord.json", C.CHORD_DIC), ("chord_root.json", C.CHORD_ROOT_DIC), ("chord_attr.json", C.CHORD_ATTR_DIC)):
        with open(os.path.join(meta, name), "w") as f:
            json.dump(table, f)
    with open(os.path.join(d("vevo_meta", "split", "v1"), "test.txt"), "w") as f:
        f.write("\n".join(c["ids"]) + "\n")
    for fid in c["ids"]:
        n = len(c[f"{fid}_chords"])
        lab = f"key {c[f'{fid}_key']}\n" + "".join(f"{t} {CHORDS[int(k)]}\n" for t, k in enumerate(c[f"{fid}_chords"]))
        with open(os.path.join(d("vevo_chord", "lab_v2_norm", "origin"), fid + ".lab"), "w") as f:
            f.write(lab)
        np.save(os.path.join(d("vevo_semantic", "origin", "2d", "clip_l14p"), fid + ".npy"), c[f"{fid}_semantic"])
        np.save(os.path.join(d("vevo_motion", "option1"), fid + ".npy"), c[f"{fid}_motion1"])
        with open(os.path.join(d("vevo_motion", "origin"), fid + ".lab"), "w") as f:
            f.write("".join(f"{t} {v:.4f}\n" for t, v in enumerate(c[f"{fid}_motion0"])))
        with open(os.path.join(d("vevo_scene_offset", "origin"), fid + ".lab"), "w") as f:
            f.write("".join(f"{t} {int(v)}\n" for t, v in enumerate(c[f"{fid}_scene"])))
        with open(os.path.join(d("vevo_emotion", "6c_l14p", "origin"), fid + ".lab"), "w") as f:
            f.write("time exciting_prob fearful_prob tense_prob sad_prob relaxing_prob neutral_prob\n")
            f.write("".join(f"{t} " + " ".join(f"{p:.4f}" for p in row) + "\n" for t, row in enumerate(c[f"{fid}_emotion"])))
        if with_targets:
            with open(os.path.join(d("vevo_chord", "lab_v2", "origin"), fid + ".lab"), "w") as f:
                f.write(lab)
            for sub in ("vevo_loudness", "vevo_note_density"):
                with open(os.path.join(d(sub, "origin"), fid + ".lab"), "w") as f:
                    f.write("".join(f"{t} 0.5\n" for t in range(n)))
            with open(os.path.join(d("vevo_instrument", "thresholding"), fid + ".csv"), "w") as f:
                f.write(",".join(f"i{j}" for j in range(40)) + "\n" + "".join(",".join(["0"] * 40) + "\n" for _ in range(n)))
