"""A/B of the lockstep V2 step's plain-layer folds in ONE process (box-to-box and run-to-run noise is larger than the effect):
AMT_V2_FOLD_FFN = 0 (separate launches), 1 (norm3 -> next QKV), 2 (+ norm2 -> gate | up), alternating, B = 32, T = 300."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video2music_amd import _lib, synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2

B, T = 32, 300
cfg = dict(version_name="2.2", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=T, total_vf_dim=1287)
m = VideoMusicTransformer_V2(**cfg).eval()
shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()})
m = m.cuda()
f = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(B, seed=5).items()}
pr = [torch.tensor([v]) for v in (1, 1, 0)]
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2"]
res = {k: [] for k in modes}
launches, ids = {}, {}
with torch.no_grad():
    for rnd in range(8):
        for k in modes:
            os.environ["AMT_V2_FOLD_FFN"] = k
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], *pr, target_seq_length=T, beam=0, sampler="argmax")
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if rnd >= 2:
                res[k].append(dt)
            launches[k] = int(_lib.call("amt_v2_last_step_launches"))
            ids[k] = out.cpu()
assert all(torch.equal(ids[k], ids[modes[0]]) for k in modes)
print(json.dumps({k: {"launches_per_step": launches[k], "generate_ms_median": round(1e3 * float(np.median(v)), 2), "generate_ms_min": round(1e3 * min(v), 2),
                      "tokens_per_s_median": round(B * (T - 1) / float(np.median(v)), 1)} for k, v in res.items()}))
