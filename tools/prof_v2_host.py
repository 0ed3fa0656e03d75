import os, sys, time, json, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2
cfg = dict(version_name="2.2", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=300, total_vf_dim=1287)
m = VideoMusicTransformer_V2(**cfg).eval()
shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()})
m = m.cuda()
nb = 32
fb = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(nb, seed=5).items()}
pr = [torch.tensor([v]) for v in (1, 1, 0)]
args = (fb["semantic"], fb["key"], fb["scene_offset"], fb["motion"], fb["emotion"], *pr)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    m.generate_batch(*args, target_seq_length=8, beam=0, sampler="argmax")
    m.generate_batch(*args, target_seq_length=8, beam=0, sampler="argmax")
    for rep in range(2):
        t0 = sync()
        rows, _, S = m._encode_memory(fb["semantic"], fb["scene_offset"], fb["motion"], fb["emotion"], clips=True)
        t1h = time.perf_counter(); t1 = sync()
        st = m._cache_init([rows[c * S:(c + 1) * S] for c in range(nb)], S)
        t2h = time.perf_counter(); t2 = sync()
        out = m.generate_batch(*args, target_seq_length=3, beam=0, sampler="argmax")
        t3 = sync()
    print(json.dumps({"encode_ms": (t1 - t0) * 1e3, "encode_host_ms": (t1h - t0) * 1e3, "cache_init_ms": (t2 - t1) * 1e3, "cache_init_host_ms": (t2h - t1) * 1e3, "generate_T3_ms": (t3 - t2) * 1e3}))
    pr_ = cProfile.Profile(); pr_.enable()
    out = m.generate_batch(*args, target_seq_length=3, beam=0, sampler="argmax"); torch.cuda.synchronize()
    pr_.disable()
    pstats.Stats(pr_).sort_stats("cumulative").print_stats(28)
