"""Lockstep V2 generate (bench.py's v2_lockstep leg) with 4 ... 64 steps per captured graph, alternating inside one process."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
out = {}
for rnd in range(2):
    for k in (4, 8, 16, 32, 64):
        os.environ["AMT_V2_STEPS_PER_GRAPH"] = str(k)
        r = bench.v2_lockstep_leg(torch.device("cuda", 0))
        out.setdefault(str(k), []).append((r["generate_ms"], r["us_per_step_from_slope"]))
print(json.dumps(out))
