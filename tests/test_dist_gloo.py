"""world_size-2 test of the data-parallel path on CPU (gloo): clip sharding + the one all_gather of
generated id matrices.  The per-rank generate itself is GPU-only, so ranks fabricate their shard's
ids deterministically; the test checks that every rank ends with the global matrix in clip order."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["AMT_ROOT"])
import torch
import torch.distributed as dist
from video2music_amd import dist as vdist

rank, world, _ = vdist.init("gloo")
n_clips, T = int(os.environ["N_CLIPS"]), 16
lo, hi = vdist.shard_bounds(n_clips, rank, world)
local = (torch.arange(lo, hi).view(-1, 1) * 1000 + torch.arange(T).view(1, -1)).long()     # stands in for generate_batch
full = vdist.all_gather_sequences(local, n_clips)
want = (torch.arange(n_clips).view(-1, 1) * 1000 + torch.arange(T).view(1, -1)).long()
assert torch.equal(full, want), (rank, full[:, 0])
# the bench's timing reduction: MAX over ranks
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == world
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_clips", [8, 5])       # even and ragged shards
def test_two_rank_gather(tmp_path, n_clips):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AMT_ROOT=ROOT, N_CLIPS=str(n_clips), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts the ranks itself and relays rank 0's
    line (the first SCALE run died on an assert).  Rehearsal mode: gloo, fabricated ids, no GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AMT_BENCH_REHEARSAL="1", AMT_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--batch", "3", "--seq", "8"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["gathered"] == [6, 8]


def test_bench_relays_a_failing_rank():
    """A rank that raises (here: a negative shard size) must make the parent exit non-zero with no result line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AMT_BENCH_REHEARSAL="1", AMT_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--batch", "-1", "--seq", "8"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and not p.stdout.decode().strip()
