"""Randomised sweep of the feature-file reader (video2music_amd/dataset/vevo_features.py) against the REFERENCE's own
`VevoDataset.createSample` (dataset/vevo_dataset.py) on random miniature datasets: clip lengths 1 ... 400 s around the 300-row limits,
random chord sequences over the whole vocabulary spellings the helper writes, random keys (major / minor), both motion types,
the 300-row limits of the callers.

TEST INFRASTRUCTURE; runs only in the build container (it imports /root/reference like oracle/make_goldens.py):
    PYTHONDONTWRITEBYTECODE=1 python oracle/sweep_features_vs_reference.py [n_datasets] [seed]
Prints one line per dataset and a summary; exit 1 on any mismatch.
"""
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from video2music_amd.dataset import vevo_features as V     # noqa: E402
from tests.helpers_features import write_mini_dataset, CHORDS     # noqa: E402


def random_content(rng, n_clips):
    c = {"ids": [f"{i:03d}" for i in rng.choice(900, size=n_clips, replace=False)]}
    for fid in c["ids"]:
        n = int(rng.choice([1, 2, 39, 150, 299, 300, 301, 320, 400]))
        key = f"{rng.choice(['C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B'])} {rng.choice(['major', 'minor'])}"
        c[f"{fid}_semantic"] = (rng.integers(-64, 64, size=(n, int(rng.choice([8, 24])))) / 16).astype(np.float32)
        c[f"{fid}_motion1"] = (rng.integers(-8, 8, size=(n, 512)) / 4).astype(np.float32)
        c[f"{fid}_motion0"] = np.round(rng.random(n) * 50, 4)
        c[f"{fid}_scene"] = np.cumsum(rng.random(n) < 0.2).astype(np.int64)
        e = rng.random((n, 6)).astype(np.float64)
        c[f"{fid}_emotion"] = np.round(e / e.sum(1, keepdims=True), 4)
        c[f"{fid}_chords"] = rng.integers(0, len(CHORDS), size=n).astype(np.int64)
        c[f"{fid}_key"] = np.array(key)
    return c


def main():
    n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    import make_goldens as G
    G.import_reference()                                    # chdirs into the reference tree, stubs off-path modules
    from dataset.vevo_dataset import VevoDataset
    bad = total = ref_fails = 0
    for s in range(n_sets):
        content = random_content(rng, int(rng.integers(1, 4)))
        tmp = tempfile.mkdtemp(prefix="vevo_sweep_")
        try:
            write_mini_dataset(tmp, content, with_targets=True)
            for mt in (0, 1):
                msc = 300          # (the reference's createSample indexes past a shorter chord limit: vevo_dataset.py:492)
                try:
                    ds = VevoDataset(dataset_root=tmp + "/", split="test", split_ver="v1", vis_models="2d/clip_l14p", emo_model="6c_l14p",
                                     motion_type=mt, max_seq_chord=msc, max_seq_video=300, random_seq=False, is_video=True)
                except IndexError as e:                 # a clip of exactly 300 rows: the reference writes CHORD_END at index 299 of a 299-row target (:326)
                    ref_fails += 1
                    for fid in content["ids"]:
                        V.load_clip(tmp, fid, motion_type=mt, max_seq_video=300, max_seq_chord=msc)      # the reader itself must not fail
                    print(f"set {s}: the reference's dataset class raises ({e}) for clips {[len(content[f + '_chords']) for f in content['ids']]}")
                    continue
                for i, fid in enumerate(content["ids"]):
                    ref = ds[i]
                    got = V.load_clip(tmp, fid, motion_type=mt, max_seq_video=300, max_seq_chord=msc)
                    pairs = [("semantic", "semanticList"), ("scene_offset", "scene_offset"), ("motion", "motion"), ("emotion", "emotion"),
                             ("key", "key"), ("chord", "chord"), ("chord_root", "chord_root"), ("chord_attr", "chord_attr")]
                    diffs = [a for a, b in pairs if not np.array_equal(np.asarray(got[a]), ref[b].numpy())]
                    total += 1
                    if diffs:
                        bad += 1
                        print(f"MISMATCH set {s} clip {fid} n={len(content[fid + '_chords'])} motion_type={mt} max_seq_chord={msc}: {diffs}")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        print(f"set {s}: clips {[len(content[f + '_chords']) for f in content['ids']]} ok", flush=True)
    print({"clip_reads": total, "mismatches": bad, "datasets_the_reference_itself_fails_on": ref_fails})
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
