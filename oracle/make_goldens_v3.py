"""Golden fixtures of VideoMusicTransformer_V3, from the REFERENCE class itself on CPU:
versions '3.0' (differential attention in the decoder), '3.1' (in both stacks) and '3.2' (pre-norm); B = 2 exercises the
raw-view reinterpretations that mix clips and positions.

TEST INFRASTRUCTURE; runs only in the build container:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_v3.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG                                   # noqa: E402  (import recipe, procedural weights)
from video2music_amd import synthetic                       # noqa: E402

CASES = (("v30", "3.0", False), ("v31", "3.1", False), ("v32", "3.2", False))
CFG = dict(n_layers=4, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300)


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    ref = MG.import_reference()
    from torch.distributions.categorical import Categorical
    orig_sample = Categorical.sample
    t = MG.t
    feats = synthetic.synthetic_features(3, seed=1234)
    key = np.array([[0.0], [1.0], [0.0]], dtype=np.float32)
    out = {}
    for tag, version, rms in CASES:
        m = ref.vmt.VideoMusicTransformer_V3(version_name=version, rms_norm=rms, total_vf_dim=synthetic.total_vf_dim(1), **CFG).eval()
        MG.load_synthetic(m, seed=0)
        rs = np.random.RandomState(37)
        B, L = 2, 12
        rootv = rs.randint(0, 13, size=(B, L)).astype(np.int64)
        attrv = rs.randint(0, 14, size=(B, L)).astype(np.int64)
        sl = slice(0, B)
        y = m(torch.zeros_like(t(rootv)), t(rootv), t(attrv), t(feats["semantic"][sl]), t(key[sl]), t(feats["scene_offset"][sl]),
              t(feats["motion"][sl]), t(feats["emotion"][sl]))
        out[f"{tag}_root"], out[f"{tag}_attr"], out[f"{tag}_logits"] = rootv, attrv, y.numpy()
        r1, a1 = rs.randint(0, 13, size=(1, 24)).astype(np.int64), rs.randint(0, 14, size=(1, 24)).astype(np.int64)
        y1 = m(torch.zeros_like(t(r1)), t(r1), t(a1), t(feats["semantic"][2:3]), t(key[2:3]), t(feats["scene_offset"][2:3]),
               t(feats["motion"][2:3]), t(feats["emotion"][2:3]))
        out[f"{tag}_root1"], out[f"{tag}_attr1"], out[f"{tag}_logits1"] = r1, a1, y1.numpy()
        kw = dict(feature_semantic_list=t(feats["semantic"][:1]), feature_key=t(key[0]), feature_scene_offset=t(feats["scene_offset"][:1]),
                  feature_motion=t(feats["motion"][:1]), feature_emotion=t(feats["emotion"][:1]),
                  primer=torch.tensor([1]), primer_root=torch.tensor([1]), primer_attr=torch.tensor([0]), target_seq_length=16)
        out[f"{tag}_g1"] = m.generate(beam=1, beam_chance=1.0, **kw).numpy()
        margins = []

        def argmax_sample(self, sample_shape=torch.Size()):
            top2 = torch.topk(self.probs.flatten(), 2)[0]
            margins.append(float(top2[0] - top2[1]))
            return self.probs.argmax(-1)

        Categorical.sample = argmax_sample
        try:
            out[f"{tag}_g2"] = m.generate(beam=0, **kw).numpy()
        finally:
            Categorical.sample = orig_sample
        out[f"{tag}_g2_margins"] = np.array(margins, dtype=np.float64)
        out[f"{tag}_n_keys"] = np.array(len(m.state_dict()))
        print(tag, "keys", len(m.state_dict()), "G1 unique", len(set(out[f"{tag}_g1"].flatten().tolist())),
              "G2 unique", len(set(out[f"{tag}_g2"].flatten().tolist())), "min margin", min(margins))
    np.savez_compressed(os.path.join(MG.OUT, "g_v3.npz"), **out)


if __name__ == "__main__":
    main()
