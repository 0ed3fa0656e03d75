"""Experiment: G handles x (32/G clips), graph chunks submitted round-robin on G streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from video2music_amd import synthetic, _lib
from video2music_amd.utilities import constants as K
from bench import make_model

def main():
    cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=1024, total_vf_dim=1287, rpr=True)
    B, T = 32, 1024
    feats = synthetic.synthetic_features(B, seed=1234)
    prim = [torch.tensor([v], device="cuda") for v in K.primer_from_name("C")]
    hip = C.CDLL("libamdhip64.so")
    class RawStream:
        def __init__(self):
            h = C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0
            self.cuda_stream = h.value
    RAW = os.environ.get("RAW", "0") == "1"
    def sp(s): return C.c_void_p(s.cuda_stream)
    def run(G, chunk):
        models = [make_model(cfg, "cuda")[0] for _ in range(G)]
        streams = [torch.cuda.ExternalStream(RawStream().cuda_stream) if RAW else torch.cuda.Stream() for _ in range(G)]
        per = B // G
        fs = [{k: torch.from_numpy(v[g * per:(g + 1) * per]).cuda() for k, v in feats.items()} for g in range(G)]
        outs = [torch.empty(per, T, dtype=torch.long, device="cuda") for _ in range(G)]
        hs = [m._ensure_handle() for m in models]
        def once():
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    f = fs[g]; m = models[g]
                    sem, key, scene, motion, emotion, _, S = m._prep_features(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
                    m._encode(hs[g], sem, scene, motion, emotion, slice(0, per))
                    _lib.call("amt_generate_begin", hs[g], per, _lib.ptr(prim[0]), _lib.ptr(prim[1]), _lib.ptr(prim[2]), 1, 0, _lib.ptr(key), T, 0, 0, 2, sp(streams[g]))
            left = T - 1
            while left > 0:
                n = min(chunk, left)
                for g in range(G):
                    _lib.call("amt_generate_run", hs[g], n, None, sp(streams[g]))
                left -= n
            for g in range(G):
                _lib.call("amt_generate_end", hs[g], _lib.ptr(outs[g]), sp(streams[g]))
            torch.cuda.synchronize()
        once(); once()
        t0 = time.perf_counter(); n = 3
        for _ in range(n): once()
        dt = (time.perf_counter() - t0) / n
        print(f"G={G} chunk={chunk}: {dt*1e3:.1f} ms per generate of {B} clips -> {B*(T-1)/dt:.0f} tok/s", flush=True)
        return torch.cat(outs)
    with torch.no_grad():
        a = run(1, 8); b = run(2, 8); c = run(2, 1); d = a
    print("ids equal:", torch.equal(a, b), torch.equal(a, c), torch.equal(a, d))


if __name__ == "__main__":
    main()
