"""Evidence behind bench.py's cpu_baseline thread count: one step of the CPU oracle (the
reference loop's full re-forward, B = 1, config 2) at L = 512 chord positions, timed with 4 ... 128 intra-op threads on the GPU
box's host.  Prints one JSON object; run as `python tools/cpu_thread_sweep.py > gpurun_out/cpu_thread_sweep.json`."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import amt_oracle as O                      # noqa: E402  (measurement of the checker itself, like bench.py's cpu_baseline)
from video2music_amd import synthetic                   # noqa: E402
from video2music_amd.model.video_music_transformer import VideoMusicTransformer  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=1024, total_vf_dim=synthetic.total_vf_dim(1), rpr=True)
    m = VideoMusicTransformer(**cfg)
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()}
    f = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_features(1, seed=99).items()}
    rs = np.random.RandomState(0)
    root = torch.from_numpy(rs.randint(1, 13, size=(1, L)))
    attr = torch.from_numpy(rs.randint(1, 14, size=(1, L)))
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = None
    out = {"L": L, "host_cpus": os.cpu_count(), "sched_affinity": affinity, "torch_default_threads": torch.get_num_threads(), "seconds_per_forward": {}}
    try:
        out["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        pass
    with torch.no_grad():
        for n in (4, 8, 16, 32, 64, 128):
            if n > (os.cpu_count() or 1):
                break
            torch.set_num_threads(n)
            best = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                O.forward(sd, cfg["num_heads"], root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
                best = min(best, time.perf_counter() - t0)
            out["seconds_per_forward"][str(n)] = round(best, 4)
    best_n = min(out["seconds_per_forward"], key=lambda k: out["seconds_per_forward"][k])
    out["best_threads"] = int(best_n)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
