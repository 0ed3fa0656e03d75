import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import synthetic, _lib
from video2music_amd.utilities import constants as C
from bench import make_model
cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=1024, total_vf_dim=1287, rpr=True)
m, _ = make_model(cfg, "cuda")
f = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(32, seed=1).items()}
pr, prr, pra = (torch.tensor([v], device="cuda") for v in C.primer_from_name("C"))
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pr, prr, pra, target_seq_length=1024, beam=0, sampler="argmax")
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3
for _ in range(3): print("enqueue ms %.1f  total ms %.1f" % run())
