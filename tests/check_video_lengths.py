"""One-off GPU check (test infrastructure): max_sequence_video other than the callers' 300 -- clips of up to 500 frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import amt_oracle as O
from video2music_amd import synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer
from video2music_amd.utilities import constants as C
from tests.helpers import CFG1, synthetic_sd, feats_t

for msv, S, B in ((500, 450, 2), (64, 40, 3), (301, 301, 1), (1024, 1000, 1)):
    cfg = dict(CFG1, max_sequence_video=msv)
    m = VideoMusicTransformer(**cfg).eval()
    sd = synthetic_sd(cfg, seed=msv, recipe="feedback")
    m.load_state_dict(sd, strict=False)
    m = m.cuda()
    fc = feats_t(synthetic.synthetic_features(B, seed=msv, n_frames=S))
    f = {k: v.cuda() for k, v in fc.items()}
    pr, prr, pra = (torch.tensor([v]) for v in C.primer_from_name("C"))
    T = 14
    with torch.no_grad():
        out = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pr, prr, pra, target_seq_length=T, beam=0, sampler="argmax").cpu()
    for b in range(B):
        one = {k: v[b:b + 1] for k, v in fc.items()}
        ref = O.generate(sd, 4, one["semantic"], one["key"], one["scene_offset"], one["motion"], one["emotion"], pr, prr, pra, target_seq_length=T, beam=0)
        assert torch.equal(out[b:b + 1], ref), (msv, S, b, out[b], ref)
    rs = np.random.RandomState(1)
    root, attr = torch.from_numpy(rs.randint(0, 13, size=(B, 9))), torch.from_numpy(rs.randint(0, 14, size=(B, 9)))
    with torch.no_grad():
        lg = m(root, root.cuda(), attr.cuda(), f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"]).cpu()
    ref = O.forward(sd, 4, root, attr, fc["semantic"], fc["key"], fc["scene_offset"], fc["motion"], fc["emotion"])
    err = float((lg - ref).abs().max())
    assert err < 1e-3, (msv, S, err)
    print("max_sequence_video", msv, "frames", S, "clips", B, "ok, forward err", err, flush=True)
