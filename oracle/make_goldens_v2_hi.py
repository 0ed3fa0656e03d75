"""Round-2 V2 fixture with well-conditioned decisions — TEST INFRASTRUCTURE, build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_v2_hi.py

The round-1 V2 golden (g_v2_cfg1.npz) has a minimum top-1 / top-2 margin of 4.2e-4: one fp32 reordering away from a flaky id
test.  This one loads the reference's VideoMusicTransformer_V2('2.2') (`model/video_music_transformer.py:316-609`)
with the "feedback" weight recipe (video2music_amd/synthetic.py) and records forward logits, G1, and G2 at temperature 1.0 and
0.8 for two clips, T=48, with every step's margin; the script searches the recipe's seed for a model whose margins are all
>= 1e-2 and prints distinct-id counts and minimum margins."""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG                                  # noqa: E402
from video2music_amd import synthetic                      # noqa: E402
from video2music_amd.utilities import constants as C       # noqa: E402

t = MG.t


def main():
    ref = MG.import_reference()
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    feats = synthetic.synthetic_features(3, seed=1234)
    key = np.array([[0.0], [1.0], [0.0]], dtype=np.float32)
    cfg = dict(version_name="2.2", n_layers=6, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300,
               total_vf_dim=synthetic.total_vf_dim(1))
    m = ref.vmt.VideoMusicTransformer_V2(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    Categorical = torch.distributions.categorical.Categorical
    orig = Categorical.sample
    T = 48

    def g2(clip, prim, **kw):
        margins = []

        def argmax_sample(self, sample_shape=torch.Size()):
            top2 = torch.topk(self.probs.flatten(), 2)[0]
            margins.append(float((top2[0] - top2[1]) / self.probs.sum()))      # margin of the re-normalised decision distribution
            return self.probs.argmax(-1)

        sl = slice(clip, clip + 1)
        Categorical.sample = argmax_sample
        try:
            ids = m.generate(feature_semantic_list=t(feats["semantic"][sl]), feature_key=t(key[clip]), feature_scene_offset=t(feats["scene_offset"][sl]),
                             feature_motion=t(feats["motion"][sl]), feature_emotion=t(feats["emotion"][sl]), primer=torch.tensor([prim[0]]),
                             primer_root=torch.tensor([prim[1]]), primer_attr=torch.tensor([prim[2]]), target_seq_length=T, beam=0, **kw)
        finally:
            Categorical.sample = orig
        return ids.numpy(), np.array(margins, dtype=np.float64)

    best = None
    for seed in range(0, 12):
        sd = synthetic.synthetic_state_dict(shapes, seed=seed, recipe="feedback")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        runs = {}
        ok = True
        for clip, prim in ((0, C.primer_from_name("C")), (1, C.primer_from_name("A:min"))):
            for name, kw in (("t10", {}), ("t08", dict(temperature=0.8, max_conseq_N=1, max_conseq_chord=3))):
                ids, mg = g2(clip, prim, **kw)
                runs[(clip, name)] = (ids, mg)
                ok = ok and mg.min() >= 1e-2
        stats = {k: (len(set(v[0].flatten().tolist())), round(float(v[1].min()), 4)) for k, v in runs.items()}
        print("seed", seed, stats, flush=True)
        if ok:
            best = (seed, runs)
            break
    assert best is not None, "no seed with all margins >= 1e-2"
    seed, runs = best
    out = {"seed": np.array(seed), "key": key}
    for (clip, name), (ids, mg) in runs.items():
        out[f"g2_{name}_clip{clip}"], out[f"g2_{name}_margins_clip{clip}"] = ids, mg
    for clip, prim in ((0, C.primer_from_name("C")), (1, C.primer_from_name("A:min"))):
        sl = slice(clip, clip + 1)
        out[f"primer_clip{clip}"] = np.array(prim, dtype=np.int64)
        out[f"g1_clip{clip}"] = m.generate(feature_semantic_list=t(feats["semantic"][sl]), feature_key=t(key[clip]), feature_scene_offset=t(feats["scene_offset"][sl]),
                                           feature_motion=t(feats["motion"][sl]), feature_emotion=t(feats["emotion"][sl]), primer=torch.tensor([prim[0]]),
                                           primer_root=torch.tensor([prim[1]]), primer_attr=torch.tensor([prim[2]]), target_seq_length=T, beam=1,
                                           beam_chance=1.0).numpy()
    # forward logits along clip 0's generated sequence
    ids = out["g2_t10_clip0"][0]
    ra = np.array([C.chord_to_root_attr(int(i)) for i in ids], dtype=np.int64)
    ra[0] = (out["primer_clip0"][1], out["primer_clip0"][2])
    out["fwd_root"], out["fwd_attr"] = ra[None, :, 0].copy(), ra[None, :, 1].copy()
    out["fwd_logits"] = m(torch.zeros(1, T, dtype=torch.long), t(out["fwd_root"]), t(out["fwd_attr"]), t(feats["semantic"][:1]), t(key[:1]),
                          t(feats["scene_offset"][:1]), t(feats["motion"][:1]), t(feats["emotion"][:1])).numpy()
    np.savez_compressed(os.path.join(MG.OUT, "g_v2_hi.npz"), **out)
    print("wrote g_v2_hi.npz with seed", seed, flush=True)


if __name__ == "__main__":
    main()
