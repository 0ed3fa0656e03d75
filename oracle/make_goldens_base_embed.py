"""Base VideoMusicTransformer with chord_embed=True / scene_embed=True — TEST INFRASTRUCTURE, build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_base_embed.py

The reference class (`model/video_music_transformer.py:910-1132`), config 1 with rpr=True:
  * scene_embed=True (:926-928,1016-1027): total_vf_dim without the scene column, scene_embedding(offset.int()) added;
  * chord_embed=True (:931-937,986-987): that branch loads a gensim Word2Vec file the tree does not ship, so the model is built
    with chord_embed=False and the two attributes the branch would have set (`chord_embed`, `chord_embedding_model`) are set by
    hand to a procedural frozen table — the same recipe as oracle/make_goldens_v2x.py;
  * both together.
Per variant: forward logits (B = 1, 2; L = 12), G1, G2 (T = 32) with margins.  -> tests/golden/g_base_embed.npz"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG                                  # noqa: E402
from video2music_amd import synthetic                      # noqa: E402

t = MG.t


def main():
    ref = MG.import_reference()
    torch.set_grad_enabled(False)
    from torch.distributions.categorical import Categorical
    orig = Categorical.sample
    feats = synthetic.synthetic_features(3, seed=1234)
    key = np.array([[0.0], [1.0], [0.0]], dtype=np.float32)
    out = {"key": key}
    for tag, chord_embed, scene_embed in (("ce", True, False), ("se", False, True), ("cese", True, True)):
        cfg = dict(n_layers=2, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300,
                   total_vf_dim=synthetic.total_vf_dim(1) - int(scene_embed), rpr=True, scene_embed=scene_embed)
        m = ref.vmt.VideoMusicTransformer(**cfg).eval()
        MG.load_synthetic(m, seed=0)
        if chord_embed:
            table = torch.from_numpy(synthetic.fill_tensor("chord_embedding_model.weight", (159, 128), 0))
            m.chord_embed = True
            m.chord_embedding_model = torch.nn.Embedding.from_pretrained(table, freeze=True)
        out[f"{tag}_n_keys"] = np.array(len(m.state_dict()))
        rs = np.random.RandomState(31)
        for B in (1, 2):
            L = 12
            ids = rs.randint(0, 157, size=(B, L)).astype(np.int64)
            rootv = rs.randint(0, 13, size=(B, L)).astype(np.int64)
            attrv = rs.randint(0, 14, size=(B, L)).astype(np.int64)
            sl = slice(0, B)
            y = m(t(ids), t(rootv), t(attrv), t(feats["semantic"][sl]), t(key[sl]), t(feats["scene_offset"][sl]),
                  t(feats["motion"][sl]), t(feats["emotion"][sl]))
            out[f"{tag}_x_B{B}"], out[f"{tag}_root_B{B}"], out[f"{tag}_attr_B{B}"], out[f"{tag}_logits_B{B}"] = ids, rootv, attrv, y.numpy()
        kw = dict(feature_semantic_list=t(feats["semantic"][:1]), feature_key=t(key[0]), feature_scene_offset=t(feats["scene_offset"][:1]),
                  feature_motion=t(feats["motion"][:1]), feature_emotion=t(feats["emotion"][:1]),
                  primer=torch.tensor([1]), primer_root=torch.tensor([1]), primer_attr=torch.tensor([0]), target_seq_length=32)
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
            Categorical.sample = orig
        out[f"{tag}_g2_margins"] = np.array(margins, dtype=np.float64)
        print(tag, "keys", int(out[f"{tag}_n_keys"]), "G1 unique", len(set(out[f"{tag}_g1"].flatten().tolist())), "G2 unique",
              len(set(out[f"{tag}_g2"].flatten().tolist())), "min margin", min(margins), flush=True)
    np.savez_compressed(os.path.join(MG.OUT, "g_base_embed.npz"), **out)
    print("wrote g_base_embed.npz")


if __name__ == "__main__":
    main()
