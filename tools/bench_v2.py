import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2
cfg = dict(version_name="2.2", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=300, total_vf_dim=1287)
m = VideoMusicTransformer_V2(**cfg).eval()
shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()})
m = m.cuda()
f = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(1, seed=3).items()}
kw = dict(primer=torch.tensor([1]), primer_root=torch.tensor([1]), primer_attr=torch.tensor([0]), beam=0, sampler="argmax")
with torch.no_grad():
    m.generate(f["semantic"], f["key"][0], f["scene_offset"], f["motion"], f["emotion"], target_seq_length=16, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.generate(f["semantic"], f["key"][0], f["scene_offset"], f["motion"], f["emotion"], target_seq_length=300, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    root = torch.randint(1, 13, (8, 300)); attr = torch.randint(1, 14, (8, 300))
    f8 = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(8, seed=4).items()}
    m(root, root, attr, f8["semantic"], f8["key"], f8["scene_offset"], f8["motion"], f8["emotion"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m(root, root, attr, f8["semantic"], f8["key"], f8["scene_offset"], f8["motion"], f8["emotion"])
    torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 3
    batch = {}
    for nb in (1, 4, 16, 32):                   # lockstep generate_batch: clips per second and per-step time
        fb = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(nb, seed=5).items()}
        args = (fb["semantic"], fb["key"], fb["scene_offset"], fb["motion"], fb["emotion"], kw["primer"], kw["primer_root"], kw["primer_attr"])
        m.generate_batch(*args, target_seq_length=8, beam=0, sampler="argmax")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ob = m.generate_batch(*args, target_seq_length=300, beam=0, sampler="argmax")
        torch.cuda.synchronize(); db = time.perf_counter() - t0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        oh = m.generate_batch(*args, target_seq_length=300, beam=0, sampler="argmax", decision="host")
        torch.cuda.synchronize(); dh = time.perf_counter() - t0
        assert torch.equal(ob, oh)
        batch[f"B{nb}"] = {"s": round(db, 3), "tokens_per_s": round(nb * 299 / db, 1), "ms_per_step": round(db / 299 * 1e3, 3),
                           "host_decision_s": round(dh, 3), "host_decision_tokens_per_s": round(nb * 299 / dh, 1)}
print(json.dumps({"v2_generate_T300_s": round(dt, 3), "tokens_per_s": round(299 / dt, 1), "unique_ids": len(set(out.flatten().tolist())),
                  "v2_forward_B8_L300_ms": round(df * 1e3, 2),
                  "v2_generate_batch_T300": batch}))
