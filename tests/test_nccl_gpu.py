"""RCCL ("nccl" backend) runs of the two multi-GPU paths, one process per GPU — only when the box shows >= 2 GPUs
(the builder's gpurun box has one; the driver's scaling node has eight), skipped otherwise.

* data-parallel generate: every rank generates its contiguous shard of the clips (video2music_amd.dist.shard_bounds)
  and ONE all_gather over RCCL yields the global id matrix, compared on every rank with the ids of a single-rank
  run over all clips;
* expert-parallel MoE (config 5): MoELayer.enable_expert_parallel() with the all_to_all over RCCL equals the
  same layer without expert parallelism;
* the library on a second device after the first (the >64 KiB dynamic-LDS opt-in of the skinny GEMM is
  per device): a model moved to cuda:1 after one ran on cuda:0 gives the same ids.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs for an RCCL run")

DP_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["AMT_ROOT"])
import numpy as np, torch, torch.distributed as dist
from video2music_amd import synthetic, dist as vdist
from video2music_amd.model.video_music_transformer import VideoMusicTransformer
from tests.helpers import CFG1, synthetic_sd, feats_t

backend = os.environ.get("AMT_TEST_BACKEND", "nccl")          # "gloo": the same worker with both ranks on one GPU
rank, world, local = vdist.init(backend)
dev = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(dev)
m = VideoMusicTransformer(**CFG1).eval()
m.load_state_dict(synthetic_sd(CFG1), strict=False)
m = m.to(dev)
n_clips, T = int(os.environ["N_CLIPS"]), 48
feats = feats_t(synthetic.synthetic_features(n_clips, seed=4242))
pr, prr, pra = (torch.tensor([v]) for v in (1, 1, 0))
def gen(sl):
    f = {k: v[sl].to(dev) for k, v in feats.items()}
    return m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pr, prr, pra,
                            target_seq_length=T, beam=0, sampler="argmax")
lo, hi = vdist.shard_bounds(n_clips, rank, world)
full = vdist.all_gather_sequences(gen(slice(lo, hi)), n_clips)
assert full.is_cuda and full.shape == (n_clips, T)
want = gen(slice(0, n_clips))                      # the single-rank run over all clips
assert torch.equal(full, want), (rank, (full != want).nonzero()[:4])
t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
dist.all_reduce(t, op=dist.ReduceOp.MAX)           # the bench's MAX-over-ranks reduction
assert float(t) == world
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""

EP_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["AMT_ROOT"])
import numpy as np, torch, torch.distributed as dist
from video2music_amd import synthetic, dist as vdist
from video2music_amd.model.moe import GLUExpert, MoELayer, SharedMoELayer
from tests.test_oracle_golden import moe_shapes

rank, world, local = vdist.init("nccl")
shared = os.environ["SHARED"] == "1"
d, dff = 128, 256
layer = SharedMoELayer(GLUExpert(d, dff), d) if shared else MoELayer(GLUExpert(d, dff), d)
sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(moe_shapes(8, d, dff, shared), seed=5).items()}
layer.load_state_dict(sd, strict=False)
layer = layer.cuda(local).eval()
rs = np.random.RandomState(7 + rank)
x = torch.from_numpy(rs.standard_normal((40 + 8 * rank, 3, d)).astype(np.float32)).cuda(local)
ref = layer(x).clone()
layer.enable_expert_parallel()
got = layer(x)
err = (got - ref).abs().max().item()
assert err < 2e-5, (rank, err)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", err)
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(tmp_path, text, world=2, **extra):
    script = tmp_path / "worker.py"
    script.write_text(text)
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AMT_ROOT=ROOT, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        if "AMT_DIST_BACKEND" not in extra:
            env.pop("AMT_DIST_BACKEND", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@needs_two_gpus
@pytest.mark.parametrize("n_clips", [6, 5])        # even and ragged shards
def test_data_parallel_generate_over_rccl(tmp_path, n_clips):
    run_ranks(tmp_path, DP_WORKER, N_CLIPS=str(n_clips))


@pytest.mark.parametrize("n_clips", [6, 5])
def test_data_parallel_generate_two_ranks_one_gpu_gloo(tmp_path, n_clips):
    """The same worker on a one-GPU box: both ranks generate their shard on cuda:0, the gather goes through gloo (host staging)."""
    run_ranks(tmp_path, DP_WORKER, N_CLIPS=str(n_clips), AMT_TEST_BACKEND="gloo", AMT_DIST_BACKEND="gloo")


@needs_two_gpus
@pytest.mark.parametrize("shared", [False, True])
def test_expert_parallel_moe_over_rccl(tmp_path, shared):
    run_ranks(tmp_path, EP_WORKER, SHARED=str(int(shared)))


@needs_two_gpus
def test_second_device_after_the_first():
    """One process, cuda:0 then cuda:1: config 2's K >= 1024 skinny GEMMs need the dynamic-LDS opt-in on each device."""
    from tests.helpers import CFG2, synthetic_sd, feats_t
    from video2music_amd import synthetic
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer
    feats = feats_t(synthetic.synthetic_features(3, seed=77))
    pr, prr, pra = (torch.tensor([v]) for v in (1, 1, 0))
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        with torch.cuda.device(dev):
            m = VideoMusicTransformer(**CFG2).eval()
            m.load_state_dict(synthetic_sd(CFG2), strict=False)
            m = m.to(dev)
            f = {k: v.to(dev) for k, v in feats.items()}
            outs.append(m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pr, prr, pra,
                                         target_seq_length=40, beam=0, sampler="argmax").cpu())
    assert torch.equal(outs[0], outs[1])


RCCL_ONE_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["AMT_ROOT"])
import numpy as np, torch, torch.distributed as dist
from video2music_amd import synthetic
from video2music_amd.model.moe import GLUExpert, MoELayer, SharedMoELayer
from tests.test_oracle_golden import moe_shapes

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)     # a real RCCL communicator, one member
assert dist.get_backend() == "nccl"
# the collectives of the data-parallel generate (video2music_amd/dist.py, bench.py) on HBM tensors
ids = (torch.arange(32, device=dev).view(-1, 1) * 10000 + torch.arange(1024, device=dev).view(1, -1)).long()
out = torch.empty_like(ids)
dist.all_gather_into_tensor(out, ids)
assert torch.equal(out, ids)
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 3.5
dist.barrier()
# the expert-parallel exchange (counts + rows out + rows back: three all_to_all_single per call) with every expert local
d, dff = 128, 256
for shared in (False, True):
    layer = SharedMoELayer(GLUExpert(d, dff), d) if shared else MoELayer(GLUExpert(d, dff), d)
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(moe_shapes(8, d, dff, shared), seed=5).items()}
    layer.load_state_dict(sd, strict=False)
    layer = layer.cuda(0).eval()
    x = torch.from_numpy(np.random.RandomState(7).standard_normal((40, 3, d)).astype(np.float32)).cuda(0)
    ref = layer(x).clone()
    layer.enable_expert_parallel()
    err = (layer(x) - ref).abs().max().item()
    assert err < 2e-5, (shared, err)
dist.barrier(); dist.destroy_process_group()
print("rccl one-rank ok")
"""


def test_rccl_communicator_and_collectives_one_rank(tmp_path):
    """What a one-GPU box can show of the RCCL path: backend "nccl" loads, creates a communicator bound to the device and runs the
    path's collectives (all_gather_into_tensor, all_reduce MAX, barrier, all_to_all_single) on HBM tensors; the expert-parallel
    MoE orchestration over it equals the local layer.  The exchange itself (two members and more) needs the tests above."""
    run_ranks(tmp_path, RCCL_ONE_RANK_WORKER, world=1)
