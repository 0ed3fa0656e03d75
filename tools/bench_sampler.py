import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from video2music_amd import synthetic
from video2music_amd.utilities import constants as C
cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=1024, total_vf_dim=synthetic.total_vf_dim(1), rpr=True)
dev = torch.device("cuda:0")
model, sd = bench.make_model(cfg, dev)
f = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.synthetic_features(32, seed=1234).items()}
pr, prr, pra = (torch.tensor([v], device=dev) for v in C.primer_from_name("C"))
for sampler in ("argmax", "categorical"):
    with torch.no_grad():
        model.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pr, prr, pra, target_seq_length=1024, beam=0, sampler=sampler)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pr, prr, pra, target_seq_length=1024, beam=0, sampler=sampler)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(sampler, round(dt * 1e3, 1), "ms", round(32 * 1023 / dt), "tok/s", "distinct ids", len(set(out.flatten().tolist())))
