"""Teacher-forced forward at config 2 (B=32, L=1024): wall time and per-kernel flop rates (run under rocprofv3 for the split)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video2music_amd import synthetic
from bench import make_model

B, L = int(os.environ.get("B", 32)), int(os.environ.get("L", 1024))
cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=1024,
           total_vf_dim=synthetic.total_vf_dim(1), rpr=True)
model, _ = make_model(cfg, "cuda")
f = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(B, seed=1).items()}
rs = np.random.RandomState(0)
root = torch.from_numpy(rs.randint(1, 13, size=(B, L))).cuda()
attr = torch.from_numpy(rs.randint(1, 14, size=(B, L))).cuda()
with torch.no_grad():
    for _ in range(2):
        model(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        model(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
d, S, H = 512, 300, 8
print(json.dumps({"forward_ms": round(dt * 1e3, 2), "token_positions_per_s": round(B * L / dt),
                  "cross_attn_gflop_per_layer": 4 * B * L * S * d / 1e9, "self_attn_gflop_per_layer_dense": 6 * B * L * L * d / 1e9}))
