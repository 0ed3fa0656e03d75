"""Wall-clock phases of the lockstep V2 generate at B = 32 (encoder, cache initialisation, step chain), synchronised between them."""
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
nb = 32
fb = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(nb, seed=5).items()}
pr = [torch.tensor([v]) for v in (1, 1, 0)]
args = (fb["semantic"], fb["key"], fb["scene_offset"], fb["motion"], fb["emotion"], *pr)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    m.generate_batch(*args, target_seq_length=8, beam=0, sampler="argmax")
    res = {}
    for rep in range(3):
        t0 = sync()
        rows, _, S = m._encode_memory(fb["semantic"], fb["scene_offset"], fb["motion"], fb["emotion"], clips=True)
        t1 = sync()
        st = m._cache_init([rows[c * S:(c + 1) * S] for c in range(nb)], S)
        t2 = sync()
        out = m.generate_batch(*args, target_seq_length=300, beam=0, sampler="argmax")
        t3 = sync()
        out = m.generate_batch(*args, target_seq_length=100, beam=0, sampler="argmax")
        t4 = sync()
        res = {"encode_ms": (t1 - t0) * 1e3, "cache_init_ms": (t2 - t1) * 1e3, "generate_T300_ms": (t3 - t2) * 1e3, "generate_T100_ms": (t4 - t3) * 1e3,
               "step_us_from_slope": ((t3 - t2) - (t4 - t3)) / 200 * 1e6}
    print(json.dumps({k: round(v, 2) for k, v in res.items()}))
