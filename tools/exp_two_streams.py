"""Experiment: G independent handles x (32/G clips) on G HIP streams vs one handle x 32 clips."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import synthetic
from video2music_amd.utilities import constants as C
from bench import make_model

def main():
    cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=1024, total_vf_dim=1287, rpr=True)
    B, T = 32, 1024
    feats = synthetic.synthetic_features(B, seed=1234)
    pr, prr, pra = (torch.tensor([v], device="cuda") for v in C.primer_from_name("C"))
    def run(G):
        models = [make_model(cfg, "cuda")[0] for _ in range(G)]
        streams = [torch.cuda.Stream() for _ in range(G)]
        per = B // G
        fs = [{k: torch.from_numpy(v[g * per:(g + 1) * per]).cuda() for k, v in feats.items()} for g in range(G)]
        outs = [None] * G
        def once():
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    f = fs[g]
                    outs[g] = models[g].generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pr, prr, pra,
                                                       target_seq_length=T, beam=0, sampler="argmax")
            torch.cuda.synchronize()
        once(); once()
        t0 = time.perf_counter(); n = 3
        for _ in range(n): once()
        dt = (time.perf_counter() - t0) / n
        print(f"G={G}: {dt*1e3:.1f} ms per generate of {B} clips -> {B*(T-1)/dt:.0f} tok/s", flush=True)
        return torch.cat(outs)
    with torch.no_grad():
        a = run(1); b = run(2); c = run(4)
    print("ids equal:", torch.equal(a, b), torch.equal(a, c))


if __name__ == "__main__":
    main()
