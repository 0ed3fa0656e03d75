"""rocprofv3 --kernel-trace target: the V2 '2.2' video encoder + cache initialisation for 32 clips (the fixed part of a lockstep generate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2
nb = int(os.environ.get("NB", "32"))
cfg = dict(version_name="2.2", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=300, total_vf_dim=1287)
m = VideoMusicTransformer_V2(**cfg).eval()
shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()})
m = m.cuda()
fb = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(nb, seed=5).items()}
with torch.no_grad():
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rows, _, S = m._encode_memory(fb["semantic"], fb["scene_offset"], fb["motion"], fb["emotion"], clips=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st = m._cache_init([rows[c * S:(c + 1) * S] for c in range(nb)], S)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("encode ms", round((t1 - t0) * 1e3, 3), "cache_init ms", round((t2 - t1) * 1e3, 3))
