#!/usr/bin/env python3
"""Measurement entry of BASELINE.json config 5 ("moe.py 8-expert FFN variant, d_model=512 seq=1024, 8 x MI355X with per-GPU expert
placement"): `MoELayer(GLUExpert(512, 1024), 512, n_experts=8, n_experts_per_token=2)` with `enable_expert_parallel()`, one layer
call over x (1024, B, 512) per rank; whole-job tokens/s, bytes per all_to_all, milliseconds per phase.

    python tools/bench_moe_ep.py [--gpus N --batch B --shared --steps K --warmup W]

N > 1 starts N ranks itself (like bench.py: the parent touches no GPU, no exec).  With fewer GPUs than ranks the ranks SHARE the
visible GPU(s) and the collectives run over gloo through the host (AMT_DIST_BACKEND=gloo is set for them): a functional rehearsal of
the N-rank path whose exchange phases say nothing about xGMI.  N == 1 runs the expert-parallel code path with one member over
RCCL (backend "nccl": every collective is a self-copy) next to the local layer -- the device-side cost of the orchestration.
Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="clips per rank: x is (1024, batch, 512)")
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--shared", action="store_true", help="SharedMoELayer (shared expert on every rank)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    return ap.parse_args()


def launch(args):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    n_dev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True).stdout.strip() or 0)
    if n_dev < args.gpus:
        env["AMT_DIST_BACKEND"] = "gloo"            # ranks share a GPU: RCCL wants one device per rank
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    sys.exit(rc if line or rc else 1)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse()
    if _a.gpus > 1:
        launch(_a)

import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import torch.distributed as dist                              # noqa: E402
from video2music_amd import dist as vdist, synthetic          # noqa: E402
from video2music_amd.model import moe as M                    # noqa: E402


def moe_shapes(n_exp, d, dff, shared):
    out = [("gate.weight", (n_exp, d)), ("gate.bias", (n_exp,))]
    for p in [f"experts.{e}." for e in range(n_exp)] + (["shared_expert."] if shared else []):
        out += [(p + "linear1.weight", (dff, d)), (p + "linear1.bias", (dff,)), (p + "gate.weight", (dff, d)), (p + "gate.bias", (dff,)),
                (p + "linear2.weight", (d, dff)), (p + "linear2.bias", (d,))]
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ:              # one member: still a process group, so that the expert-parallel path runs over RCCL
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
        s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
    rank, world, local = vdist.init()
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if not dist.is_initialized():                   # (vdist.init leaves a single process without a group)
        backend = os.environ.get("AMT_DIST_BACKEND") or "nccl"
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    d, dff, n_exp, L, B = 512, 1024, 8, args.seq, args.batch
    layer = (M.SharedMoELayer if args.shared else M.MoELayer)(M.GLUExpert(d, dff), d, n_experts=n_exp, n_experts_per_token=2)
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(moe_shapes(n_exp, d, dff, args.shared), seed=5).items()}
    layer.load_state_dict(sd, strict=False)
    layer = layer.to(device).eval()
    x = torch.from_numpy(np.random.RandomState(7 + rank).standard_normal((L, B, d)).astype(np.float32)).to(device)
    n_tok = L * B

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        return (time.perf_counter() - t0) / steps, out

    with torch.no_grad():
        local_s, ref = timed(lambda: layer(x), args.steps, args.warmup)
        layer.enable_expert_parallel()
        ep_s, got = timed(lambda: layer(x), args.steps, args.warmup)
        err = (got - ref).abs().max().item()
        ph = M._Phases(device)
        for _ in range(3):
            layer._run_ep(x, phases=ph)
        te = torch.tensor([ep_s, local_s], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
    if rank == 0:
        ep_s, local_s = float(te[0]), float(te[1])
        print(json.dumps({
            "metric": "moe_tokens_per_sec_expert_parallel", "value": round(world * n_tok / ep_s, 1), "unit": "tokens/s", "n_gpus": world,
            "backend": dist.get_backend(), "ranks_share_a_gpu": world > torch.cuda.device_count(), "ms_per_layer_call": round(1e3 * ep_s, 3),
            "same_layer_without_expert_parallelism_ms": round(1e3 * local_s, 3), "max_abs_diff_vs_local_layer": err,
            "config": {"workload": f"{'SharedMoELayer' if args.shared else 'MoELayer'}(GLUExpert(512, 1024), 512, n_experts=8, top-2), x ({L}, {B}, 512) per rank, "
                                   f"{n_exp // world} experts per rank", "tokens_per_rank": n_tok},
            "bytes_per_all_to_all_rank0": ph.bytes_per_all_to_all,
            "phase_ms_rank0": {k: round(v / 3, 3) for k, v in ph.ms.items() if k != "start"},
            "phase_note": "phases are separated by device synchronisations (3 timed calls averaged), so they add up to more than an untimed call",
            "host_syncs_per_call": 1, "dtype": "f32", "data": "synthetic"}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
