"""Chord-tokens/s of the config-2 generate against the number of clips decoded as ONE chain (model.max_decode_batch): the fixed
per-launch latencies of the 31-launch step are shared by more clips, the K/V streaming grows with them."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import synthetic
from video2music_amd.utilities import constants as C
from bench import make_model

T = int(os.environ.get("T", 1024))
cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=T, total_vf_dim=synthetic.total_vf_dim(1), rpr=True)
pr, prr, pra = (torch.tensor([v], device="cuda") for v in C.primer_from_name("C"))
res = {}
for B in (16, 32, 64, 128, 256):
    model, _ = make_model(cfg, "cuda")
    model.max_decode_batch = B
    f = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(B, seed=1234).items()}
    def gen():
        return model.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pr, prr, pra,
                                    target_seq_length=T, beam=0, sampler="argmax")
    with torch.no_grad():
        gen(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2): gen()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    res[f"B{B}"] = {"ms_per_generate": round(dt * 1e3, 1), "us_per_step": round(dt * 1e6 / (T - 1), 1), "tokens_per_s": round(B * (T - 1) / dt)}
    print(json.dumps({f"B{B}": res[f"B{B}"]}), flush=True)
    del model, f
    torch.cuda.empty_cache()
print(json.dumps(res))
