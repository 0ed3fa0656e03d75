"""Timing of the regression head VideoRegression($REGMODEL, default 'bimamba+') at the deployed size (not a test; run on the GPU box)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import synthetic
from video2music_amd.model.video_regression import VideoRegression
from oracle import reg_oracle as R

cfg = dict(n_layers=6, d_model=128, d_hidden=256, total_vf_dim=774, regModel=os.environ.get("REGMODEL", "bimamba+"))
m = VideoRegression(**cfg).eval()
shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=2).items()}
m.load_state_dict(sd)
m = m.cuda()
res = {}
for B in (1, 32):
    f = synthetic.synthetic_features(B, seed=9)
    sem, emo = torch.from_numpy(f["semantic"]).cuda(), torch.from_numpy(f["emotion"]).cuda()
    with torch.no_grad():
        for _ in range(3): m(sem, None, None, emo)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(sem, None, None, emo)
        torch.cuda.synchronize(); res[f"hip_ms_B{B}"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        # device time only: capture one forward in a graph
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m(sem, None, None, emo)
            with torch.cuda.graph(g, stream=s):
                m(sem, None, None, emo)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); res[f"hip_graph_ms_B{B}"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
torch.set_num_threads(16)
f = synthetic.synthetic_features(1, seed=9)
t0 = time.perf_counter(); R.forward(sd, torch.from_numpy(f["semantic"]), torch.from_numpy(f["emotion"]), reg_model=cfg["regModel"]); res["cpu_oracle_ms_B1"] = round((time.perf_counter() - t0) * 1e3, 1)
print(json.dumps(res))
