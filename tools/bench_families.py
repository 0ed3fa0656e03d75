"""Wall time of generate for one representative of each model family at d=512, 6 layers, 300 tokens (run on the GPU box)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V1, VideoMusicTransformer_V2, VideoMusicTransformer_V3

cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=300, total_vf_dim=1287)
res = {}
pr = [torch.tensor([v]) for v in (1, 1, 0)]
for name, cls, ver in (("V1_1.1", VideoMusicTransformer_V1, "1.1"), ("V1_1.0", VideoMusicTransformer_V1, "1.0"), ("V2_2.0", VideoMusicTransformer_V2, "2.0"),
                       ("V3_3.1", VideoMusicTransformer_V3, "3.1")):
    m = cls(version_name=ver, **cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()})
    m = m.cuda()
    f = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(16, seed=3).items()}
    one = (f["semantic"][:1], f["key"][0], f["scene_offset"][:1], f["motion"][:1], f["emotion"][:1], *pr)
    with torch.no_grad():
        m.generate(*one, target_seq_length=8, beam=0, sampler="argmax")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.generate(*one, target_seq_length=300, beam=0, sampler="argmax")
        torch.cuda.synchronize(); res[name + "_one_clip_s"] = round(time.perf_counter() - t0, 3)
        if True:
            args = (f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], *pr)
            m.generate_batch(*args, target_seq_length=8, beam=0, sampler="argmax")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.generate_batch(*args, target_seq_length=300, beam=0, sampler="argmax")
            torch.cuda.synchronize(); res[name + "_16_clips_s"] = round(time.perf_counter() - t0, 3)
    del m
print(json.dumps(res))
