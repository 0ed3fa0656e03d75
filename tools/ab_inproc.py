"""In-process A/B of a tuning field (experiment build of the library only: tools/ab_build.sh exp -DAMT_EXPERIMENT, then
AMT_LIB=video2music_amd/lib/libamt_hip.exp.so python tools/ab_inproc.py v2|base FIELD V0,V1[,V2] [rounds]; FIELD = opt:NAME switches a
handle option of the base model instead and needs no experiment build).  The variants alternate
inside one process -- box-to-box and process-to-process noise (+-1.5 %) is larger than most effects worth keeping.
v2: lockstep V2 '2.2' generate (32 clips, T = 300; the step graph is re-captured per generate, so the field acts at once).
base: bench.py's configuration (32 clips, T = 1024); one model per variant, each captured under its value."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video2music_amd import _lib, synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer, VideoMusicTransformer_V2

which, field, values = sys.argv[1], sys.argv[2], [int(v) for v in sys.argv[3].split(",")]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
lib = _lib.load()
models = {}


def put(v):
    if field.startswith("opt:"):                 # a handle option of the release library (base model only): amt_set_option(h, name, v)
        _lib.call("amt_set_option", models[v]._ensure_handle(), field[4:].encode(), v)
        return
    setter = lib.amt_experiment_set              # a tuning field: experiment build only
    setter.argtypes = [ctypes.c_char_p, ctypes.c_int32]
    assert setter(field.encode(), v) == 0, field


B = 32
pr = [torch.tensor([v]) for v in (1, 1, 0)]
if which == "v2":
    T = 300
    cfg = dict(version_name="2.2", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=T, total_vf_dim=1287)
    m = VideoMusicTransformer_V2(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()})
    models.update({v: m.cuda() for v in values})
    f = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(B, seed=5).items()}
else:
    T = 1024
    cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=T, total_vf_dim=synthetic.total_vf_dim(1), rpr=True)
    for v in values:
        m = VideoMusicTransformer(**cfg).eval()
        shapes = [(k, tuple(t.shape)) for k, t in m.state_dict().items()]
        m.load_state_dict({k: torch.from_numpy(t) for k, t in synthetic.synthetic_state_dict(shapes, seed=0).items()}, strict=False)
        models[v] = m.cuda()
    f = {k: torch.from_numpy(v).cuda() for k, v in synthetic.synthetic_features(B, seed=1234).items()}
res = {v: [] for v in values}
ids = {}
with torch.no_grad():
    for rnd in range(rounds + 2):
        for v in values:
            put(v)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = models[v].generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], *pr, target_seq_length=T, beam=0, sampler="argmax")
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if rnd >= 2:
                res[v].append(dt)
            ids[v] = out.cpu()
same = all(torch.equal(ids[v], ids[values[0]]) for v in values)
print(json.dumps({"which": which, "field": field, "ids_equal": same,
                  "generate_ms": {str(v): {"median": round(1e3 * float(np.median(r)), 2), "min": round(1e3 * min(r), 2), "max": round(1e3 * max(r), 2),
                                           "tokens_per_s": round(B * (T - 1) / float(np.median(r)), 1)} for v, r in res.items()}}))
