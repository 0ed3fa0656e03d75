"""Golden fixture of the regression head (tests/golden/g_reg.npz) from the REFERENCE's own `VideoRegression`.

TEST INFRASTRUCTURE; runs only in the build container.  Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_reg.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from video2music_amd import synthetic                      # noqa: E402

CFG = dict(n_layers=2, d_model=32, d_hidden=64, total_vf_dim=24 + 6, regModel="bimamba+")


def main():
    import types
    import make_goldens as G
    G.import_reference()

    class _Absent:                      # off-path third-party module of model/minGRULM.py, never executed for 'bimamba+'
        def __init__(self, *a, **k):
            raise RuntimeError("off-path third-party module is stubbed")
    pkg = types.ModuleType("minGRU_pytorch")
    sub = types.ModuleType("minGRU_pytorch.minGRU")
    sub.minGRU = _Absent
    pkg.minGRU = sub
    sys.modules["minGRU_pytorch"], sys.modules["minGRU_pytorch.minGRU"] = pkg, sub
    from model.video_regression import VideoRegression
    torch.manual_seed(0)
    m = VideoRegression(max_sequence_video=300, **CFG).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=5).items()}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    out = {"keys": np.array([k for k, _ in shapes]), "shapes": np.array([str(s) for _, s in shapes])}
    rs = np.random.RandomState(3)
    for B, S in ((1, 40), (3, 17), (2, 300)):
        sem = torch.from_numpy(rs.standard_normal((B, S, 24)).astype(np.float32))
        z = rs.standard_normal((B, S, 6))
        emo = torch.from_numpy((np.exp(z) / np.exp(z).sum(-1, keepdims=True)).astype(np.float32))
        with torch.no_grad():
            ln_nd, inst = m(sem, torch.zeros(B, S), torch.zeros(B, S, 512), emo)
            feat = m.get_feature(sem, None, None, emo)
        out.update({f"sem_B{B}_S{S}": sem.numpy(), f"emo_B{B}_S{S}": emo.numpy(), f"lnnd_B{B}_S{S}": ln_nd.numpy(),
                    f"inst_B{B}_S{S}": inst.numpy(), f"feat_B{B}_S{S}": feat.numpy()})
    # the other Mamba regModels: same inputs as the (3, 17) case, one golden each
    rs = np.random.RandomState(11)
    sem = torch.from_numpy(rs.standard_normal((3, 50, 24)).astype(np.float32))
    z = rs.standard_normal((3, 50, 6))
    emo = torch.from_numpy((np.exp(z) / np.exp(z).sum(-1, keepdims=True)).astype(np.float32))
    out["alt_sem"], out["alt_emo"] = sem.numpy(), emo.numpy()
    for rm in ("bimamba", "mamba", "mamba+", "moe_bimamba+", "sharedmoe_bimamba+", "lstm", "bilstm", "gru", "bigru", "cnngru", "cnnbigru", "moemamba"):
        mm = VideoRegression(max_sequence_video=300, **dict(CFG, regModel=rm)).eval()
        shp = [(k, tuple(v.shape)) for k, v in mm.state_dict().items()]
        mm.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shp, seed=5).items()}, strict=True)
        with torch.no_grad():
            ln_nd, inst = mm(sem, torch.zeros(3, 50), torch.zeros(3, 50, 512), emo)
        out[f"alt_{rm}_lnnd"], out[f"alt_{rm}_inst"] = ln_nd.numpy(), inst.numpy()
        out[f"alt_{rm}_keys"] = np.array([k for k, _ in shp])
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "g_reg.npz"), **out)
    print("wrote g_reg.npz", len(out), [(k, v) for k, v in shapes][:6])


if __name__ == "__main__":
    main()
