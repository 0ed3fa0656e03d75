"""Experiment builds only: set one tuning field (amt_experiment_set) and run bench.py's generate a few times -- a rocprofv3 target
(python3 tools/exp_bench_field.py FIELD VALUE [reps]) for per-kernel durations under a switch."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from video2music_amd import _lib, synthetic  # noqa: E402
from video2music_amd.utilities import constants as C  # noqa: E402

field, value = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = _lib.load()
lib.amt_experiment_set.argtypes = [ctypes.c_char_p, ctypes.c_int32]
assert lib.amt_experiment_set(field.encode(), value) == 0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
T = 1024
cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=T, total_vf_dim=synthetic.total_vf_dim(1), rpr=True)
model, _ = bench.make_model(cfg, dev)
f = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.synthetic_features(32, seed=1234).items()}
prim = tuple(torch.tensor([v], device=dev) for v in C.primer_from_name("C"))
with torch.no_grad():
    for _ in range(reps):
        model.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], *prim, target_seq_length=T, beam=0, sampler="argmax")
torch.cuda.synchronize()
print("done")
