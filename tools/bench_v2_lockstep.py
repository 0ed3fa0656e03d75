"""The `v2_lockstep` leg of bench.py on its own (V2 '2.2', 32 clips, T = 300): one JSON line.  AMT_V2_FOLD_FFN=0 switches the
plain-layer folds off (A/B)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WORLD_SIZE", "1")           # import bench.py as a module without its launcher branch
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    print(json.dumps(bench.v2_lockstep_leg(torch.device("cuda", 0))))
