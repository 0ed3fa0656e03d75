"""Golden fixture of the base VideoMusicTransformer built with rpr=False (the class default: torch's stock decoder layers,
model/video_music_transformer.py:957-962), from the REFERENCE class itself on CPU: forward logits, G1, G2.

TEST INFRASTRUCTURE; runs only in the build container:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_norpr.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG                                   # noqa: E402
from video2music_amd import synthetic                       # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    ref = MG.import_reference()
    from torch.distributions.categorical import Categorical
    orig_sample = Categorical.sample
    t = MG.t
    feats = synthetic.synthetic_features(3, seed=1234)
    key = np.array([[0.0], [1.0], [0.0]], dtype=np.float32)
    m = ref.vmt.VideoMusicTransformer(n_layers=2, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300,
                                      total_vf_dim=synthetic.total_vf_dim(1), rpr=False).eval()
    MG.load_synthetic(m, seed=0)
    out = {"n_keys": np.array(len(m.state_dict()))}
    rs = np.random.RandomState(41)
    B, L = 3, 20
    rootv = rs.randint(0, 13, size=(B, L)).astype(np.int64)
    attrv = rs.randint(0, 14, size=(B, L)).astype(np.int64)
    y = m(torch.zeros_like(t(rootv)), t(rootv), t(attrv), t(feats["semantic"]), t(key), t(feats["scene_offset"]), t(feats["motion"]),
          t(feats["emotion"]))
    out["root"], out["attr"], out["logits"] = rootv, attrv, y.numpy()
    kw = dict(feature_semantic_list=t(feats["semantic"][:1]), feature_key=t(key[0]), feature_scene_offset=t(feats["scene_offset"][:1]),
              feature_motion=t(feats["motion"][:1]), feature_emotion=t(feats["emotion"][:1]),
              primer=torch.tensor([1]), primer_root=torch.tensor([1]), primer_attr=torch.tensor([0]), target_seq_length=48)
    out["g1"] = m.generate(beam=1, beam_chance=1.0, **kw).numpy()
    margins = []

    def argmax_sample(self, sample_shape=torch.Size()):
        top2 = torch.topk(self.probs.flatten(), 2)[0]
        margins.append(float(top2[0] - top2[1]))
        return self.probs.argmax(-1)

    Categorical.sample = argmax_sample
    try:
        out["g2"] = m.generate(beam=0, **kw).numpy()
    finally:
        Categorical.sample = orig_sample
    print("keys", len(m.state_dict()), "G1 unique", len(set(out["g1"].flatten().tolist())), "G2 unique", len(set(out["g2"].flatten().tolist())),
          "min margin", min(margins))
    np.savez_compressed(os.path.join(MG.OUT, "g_norpr.npz"), **out)


if __name__ == "__main__":
    main()
