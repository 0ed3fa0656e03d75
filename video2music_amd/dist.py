"""Data-parallel sharding of video clips over the GPUs of one node (one process per GPU).

The reference has no multi-device concept (``utilities/device.py:8-9`` hard-codes ``cuda:0``).
Clips are independent units: each rank encodes and decodes its own CONTIGUOUS shard (SURVEY.md §8(e)
suggested the interleave ``c mod world``; contiguous shards give the same balance and make the gathered
matrix come out in clip order without a permutation) with replicated weights
and the only exchange is ONE ``all_gather`` of the generated ``(B_local, T)`` int64 id matrices at
the end (256 KiB per rank at B_local=32, T=1024 — latency-bound).
``torch.distributed`` backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialises the default process group from the torchrun environment (no-op for world size 1).

    ``AMT_DIST_BACKEND=gloo`` forces gloo (rehearsals of the multi-rank path on a box with fewer GPUs
    than ranks; ids are then staged through host memory for the gather)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("AMT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_bounds(n_clips, rank, world):
    """Contiguous shard [lo, hi) of ``n_clips`` for ``rank``; sizes differ by at most one."""
    base, rem = divmod(n_clips, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_sequences(tokens, n_clips=None):
    """Gathers the per-rank ``(B_local, T)`` id matrices into the global ``(n_clips, T)`` matrix on
    every rank, in rank order (= clip order under ``shard_bounds``) with ONE collective.

    Shard sizes are never exchanged: with ``n_clips`` given they follow from ``shard_bounds`` on every
    rank (ragged shards are padded to the largest one for the collective and trimmed afterwards);
    without it every rank MUST hold the same number of clips (the bench's case: a precondition, not checked by an
    exchange; with ``AMT_DIST_CHECK=1`` the shard sizes are compared with one small all-reduce first)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tokens
    if tokens.is_cuda and dist.get_backend() == "gloo":      # rehearsal path: gloo gathers host tensors
        return all_gather_sequences(tokens.cpu(), n_clips).to(tokens.device)
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_clips is None:
        sizes = [tokens.shape[0]] * world
        if os.environ.get("AMT_DIST_CHECK", "0") == "1":       # debug aid: unequal shards would hang or corrupt the collective
            n = torch.tensor([tokens.shape[0], -tokens.shape[0]], dtype=torch.int64, device=tokens.device)
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            if int(n[0]) != -int(n[1]):
                raise ValueError(f"all_gather_sequences without n_clips: shards of {-int(n[1])}..{int(n[0])} clips; pass n_clips")
    else:
        sizes = [hi - lo for lo, hi in (shard_bounds(n_clips, r, world) for r in range(world))]
        if tokens.shape[0] != sizes[rank]:
            raise ValueError(f"rank {rank} holds {tokens.shape[0]} clips, shard_bounds says {sizes[rank]}")
    mx = max(sizes)
    pad = tokens
    if tokens.shape[0] < mx:
        pad = torch.cat([tokens, tokens.new_zeros(mx - tokens.shape[0], tokens.shape[1])])
    out = torch.empty(world * mx, tokens.shape[1], dtype=tokens.dtype, device=tokens.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    if mx * world == sum(sizes):
        return out
    return torch.cat([out.view(world, mx, -1)[r, :sizes[r]] for r in range(world)])
