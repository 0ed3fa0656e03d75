"""Golden fixture for the feature-file reader (tests/golden/g_features.npz), produced by the REFERENCE's own
`VevoDataset` (dataset/vevo_dataset.py) run in the build container on a miniature dataset written to a temp dir.

TEST INFRASTRUCTURE.  The npz holds the *content* of the miniature dataset (so the test can rebuild the files
anywhere) and the tensors the reference's `createSample` returned for it.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_features.py
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from video2music_amd.utilities import constants as C        # noqa: E402
from tests.helpers_features import write_mini_dataset, mini_dataset_content     # noqa: E402


def main():
    content = mini_dataset_content(seed=7)
    tmp = tempfile.mkdtemp(prefix="vevo_mini_")
    write_mini_dataset(tmp, content, with_targets=True)
    import make_goldens as G
    G.import_reference()                                    # chdirs into the reference tree, stubs off-path modules
    from dataset.vevo_dataset import VevoDataset
    out = {}
    for mt in (0, 1):
        ds = VevoDataset(dataset_root=tmp + "/", split="test", split_ver="v1", vis_models="2d/clip_l14p", emo_model="6c_l14p",
                         motion_type=mt, max_seq_chord=300, max_seq_video=300, random_seq=False, is_video=True)
        assert len(ds) == len(content["ids"])
        for i, fid in enumerate(content["ids"]):
            s = ds[i]
            for k in ("chord", "chord_root", "chord_attr", "x", "x_root", "x_attr", "semanticList", "key", "scene_offset", "motion", "emotion"):
                out[f"ref_mt{mt}_{fid}_{k}"] = s[k].numpy()
    for k, v in content.items():
        if k != "ids":
            out["in_" + k] = v
    out["ids"] = np.array(content["ids"])
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "g_features.npz"), **out)
    print("wrote g_features.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
