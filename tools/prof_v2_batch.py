"""rocprofv3 --kernel-trace --stats target: lockstep generate_batch of VideoMusicTransformer_V2('2.2') (B = $NB clips, T = 100)."""
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
pr = [torch.tensor([v]) for v in (1, 1, 0)]
args = (fb["semantic"], fb["key"], fb["scene_offset"], fb["motion"], fb["emotion"], *pr)
with torch.no_grad():
    m.generate_batch(*args, target_seq_length=8, beam=0, sampler="argmax")
    for g in (True, False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.generate_batch(*args, target_seq_length=100, beam=0, sampler="argmax", use_graph=g)
        torch.cuda.synchronize(); print("graph", g, "s", round(time.perf_counter() - t0, 3))
