"""Experiment (round 3): config 2's 32 clips as ONE lockstep chain of 32 against TWO independent chains of 16 on two streams
(two handles, two host threads).  The question: do the bandwidth-bound attention launches of one chain overlap the latency-bound
skinny GEMMs and the launch boundaries of the other?  Prints one JSON line; ids of both forms are compared.

    python tools/ab_two_chains.py [T=1024] [reps=3] [chains=2]
"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from video2music_amd import synthetic  # noqa: E402
from video2music_amd.utilities import constants as C  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n_chains = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B = 32
    cfg = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=T, total_vf_dim=synthetic.total_vf_dim(1), rpr=True)
    feats = synthetic.synthetic_features(B, seed=1234)
    f = {k: torch.from_numpy(v).to(dev) for k, v in feats.items()}
    prim = tuple(torch.tensor([v], device=dev) for v in C.primer_from_name("C"))

    def gen(model, lo, hi):
        return model.generate_batch(f["semantic"][lo:hi], f["key"][lo:hi], f["scene_offset"][lo:hi], f["motion"][lo:hi], f["emotion"][lo:hi],
                                    *prim, target_seq_length=T, beam=0, sampler="argmax")

    out = {"T": T, "clips": B}
    with torch.no_grad():
        one, _ = bench.make_model(cfg, dev)
        gen(one, 0, B)
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids_one = gen(one, 0, B)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out["one_chain_ms"] = round(1e3 * best, 3)
        del one

        models = [bench.make_model(cfg, dev)[0] for _ in range(n_chains)]
        streams = [torch.cuda.Stream(dev) for _ in range(n_chains)]
        per = B // n_chains
        res = [None] * n_chains

        def work(i):
            torch.cuda.set_device(dev)
            with torch.no_grad(), torch.cuda.stream(streams[i]):
                res[i] = gen(models[i], i * per, (i + 1) * per)
                streams[i].synchronize()

        def both():
            th = [threading.Thread(target=work, args=(i,)) for i in range(n_chains)]
            for t in th:
                t.start()
            for t in th:
                t.join()

        both()
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            both()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out[f"{n_chains}_chains_ms"] = round(1e3 * best, 3)
        # one chain of B / n alone (what each chain costs when the chip is its own)
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            work(0)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out[f"chain_of_{per}_alone_ms"] = round(1e3 * best, 3)
        out["ids_equal"] = bool(torch.equal(torch.cat([r.cpu() for r in res]), ids_one.cpu()))
        out["tokens_per_s_one"] = round(B * (T - 1) / (out["one_chain_ms"] * 1e-3), 1)
        out["tokens_per_s_chains"] = round(B * (T - 1) / (out[f"{n_chains}_chains_ms"] * 1e-3), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
