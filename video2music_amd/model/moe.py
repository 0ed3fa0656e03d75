"""``GLUExpert`` / ``MoELayer`` / ``SharedMoELayer`` on the HIP path (reference ``model/moe.py:36-49,
150-200, 202-302``), eval mode.

Same parameter names (``experts.{e}.linear1|linear2|gate``, ``gate``, ``shared_expert``, buffer
``bias`` when ``balancing``).  The reference's module-global logging side effects
(``update_maxvio`` / ``update_expert_counts``) are deliberately not replicated.
Training-time behaviour (top-k / temperature schedulers, balancing-bias updates) is out of scope:
``forward`` raises in training mode.
"""
import copy

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib


class _Phases:
    """Optional wall-clock phases of one expert-parallel layer call (tools/bench_moe_ep.py): each mark synchronises the device,
    so a timed call is slower than an untimed one; `None` costs nothing."""

    def __init__(self, device):
        import time
        self.t, self.dev, self.ms, self.clock = None, device, {}, time.perf_counter

    def mark(self, name):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        now = self.clock()
        if self.t is not None:
            self.ms[name] = self.ms.get(name, 0.0) + 1e3 * (now - self.t)
        self.t = now


def expert_parallel_moe(x, idx, ops, n_experts, group=None, phases=None):
    """Expert-parallel top-2 MoE over ``torch.distributed`` (experts placed per GPU).

    Rank r owns experts ``[r*E/W, (r+1)*E/W)``.  Every rank routes its own tokens (``idx`` (n_tok, 2) int32), sends each
    (token, slot) row to the owner of its expert with ONE ``all_to_all_single`` (the send buffer is ordered by expert, so the
    per-rank segments are contiguous), runs its experts on what it received, returns the results with a second
    ``all_to_all_single`` and combines per token in expert-index order.

    Everything that touches rows runs on the device through ``ops`` (HIP kernels in production -- ``_HipEpOps`` below: the plan
    kernels of the local layer, a row gather, ONE grouped expert launch per projection --, torch on the CPU in the gloo test):

        ops.plan(idx, counts_out)            -> perm (2 n_tok,), slot_pos (2 n_tok,) [send row -> token; assignment -> send row]; rows per expert into counts_out (E,) int32
        ops.gather(x, perm)                  -> send rows (2 n_tok, d)
        ops.overlap()                        -> anything independent of the exchange (the shared expert); runs while rows travel
        ops.experts(rows, recv_counts, n)    -> results (n, d) in arrival order; recv_counts (W, E/W) int32 on the rows' device
        ops.combine(y_sorted, slot_pos)      -> (n_tok, d)

    The host learns ONE thing per call, in ONE synchronisation: the split sizes (rows per destination / source rank), which the
    variable-size collective needs as Python ints.  The per-expert counts are exchanged on the device first (a (W, E/W) int32
    all_to_all) and both count matrices come to the host together.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert n_experts % world == 0, "experts must divide evenly over ranks"
    e_local = n_experts // world
    n_tok, d = x.shape
    # gloo moves host tensors: the rehearsal path (two ranks on one GPU, CPU tests) stages through the host
    via_host = dist.get_backend(group) == "gloo" and x.is_cuda
    stage = (lambda t: t.cpu()) if via_host else (lambda t: t)
    unstage = (lambda t: t.to(x.device)) if via_host else (lambda t: t)
    # both count matrices in ONE buffer: [0] rows per (destination rank, its local expert) from the plan, [1] what the peers send here
    both_dev = torch.empty(2, world, e_local, dtype=torch.int32, device=x.device)
    perm, slot_pos = ops.plan(idx, both_dev[0].view(-1))
    send_rows = ops.gather(x, perm)
    if phases: phases.mark("route_plan_gather")
    # 1) sizes: exchanged where the rows live, then ONE copy to the host
    if via_host:
        sc_h = both_dev[0].cpu()
        rc_h = torch.empty_like(sc_h)
        dist.all_to_all_single(rc_h, sc_h, group=group)
        both_dev[1].copy_(rc_h)
        both = torch.stack([sc_h, rc_h])
    else:
        dist.all_to_all_single(both_dev[1], both_dev[0], group=group)
        both = both_dev.cpu()                                         # the one host synchronisation of the call
    recv_counts = both_dev[1]
    out_splits, in_splits = both[0].sum(1).tolist(), both[1].sum(1).tolist()
    n_recv = sum(in_splits)
    if phases: phases.mark("counts_exchange_and_host_sync")
    # 2) rows to their experts' owners; whatever does not depend on them runs meanwhile
    recv_rows = torch.empty(n_recv, d, dtype=x.dtype, device="cpu" if via_host else x.device)
    work = dist.all_to_all_single(recv_rows, stage(send_rows), output_split_sizes=in_splits, input_split_sizes=out_splits, group=group,
                                  async_op=True)
    ops.overlap()
    work.wait()
    recv_rows = unstage(recv_rows)
    if phases: phases.mark("dispatch_all_to_all")
    y_recv = ops.experts(recv_rows, recv_counts, n_recv)
    if phases: phases.mark("local_experts")
    # 3) results back to the token owners, landing at the rows they were sent from
    y_sorted = torch.empty(2 * n_tok, d, dtype=x.dtype, device="cpu" if via_host else x.device)
    dist.all_to_all_single(y_sorted, stage(y_recv), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    y_sorted = unstage(y_sorted)
    if phases: phases.mark("return_all_to_all")
    out = ops.combine(y_sorted, slot_pos)
    if phases:
        phases.mark("combine")
        phases.bytes_per_all_to_all = {"sent": int(sum(out_splits)) * d * 4, "received": int(n_recv) * d * 4,
                                       "leaving_this_rank": int(sum(out_splits) - out_splits[rank]) * d * 4}
    return out


class _HipEpOps:
    """`expert_parallel_moe`'s row operations on the library's kernels (no torch indexing kernels in the layer)."""

    def __init__(self, layer, xf, idx, wts, rank, e_local):
        self.layer, self.xf, self.idx, self.wts, self.rank, self.e_local = layer, xf, idx, wts, rank, e_local
        self.n_tok, self.d = xf.shape
        self.dff = expert_dff(layer.experts[0])
        self.shared = None
        if getattr(layer, "_ep_ints", None) is None or layer._ep_ints.device != xf.device:
            layer._ep_ints = torch.zeros(256, dtype=torch.int32, device=xf.device)      # plan scratch: zero between calls

    def plan(self, idx, counts):
        n, dev = self.n_tok, self.xf.device
        perm = torch.empty(2 * n, dtype=torch.int32, device=dev)
        slot_pos = torch.empty(2 * n, dtype=torch.int32, device=dev)
        p = _lib.ptr
        _lib.call("amt_moe_ep_dispatch_plan_fwd", p(idx), n, self.layer.n_experts, p(counts), p(perm), p(slot_pos), p(self.layer._ep_ints),
                  _lib.stream_ptr())
        return perm, slot_pos

    def gather(self, x, perm):
        rows = torch.empty(perm.numel(), self.d, device=x.device, dtype=torch.float32)
        _lib.call("amt_gather_rows_fwd", _lib.ptr(x), _lib.ptr(perm), _lib.ptr(rows), perm.numel(), self.d, _lib.stream_ptr())
        return rows

    def _glu(self, rows, m):
        out = torch.empty(rows.shape[0], self.d, device=rows.device, dtype=torch.float32)
        scratch = torch.empty(2 * rows.shape[0] * self.dff, device=rows.device, dtype=torch.float32)
        p = _lib.ptr
        _lib.call("amt_glu_expert_fwd", p(rows), *[p(v) for v in expert_tensors(m)], p(out), p(scratch), rows.shape[0], self.d, self.dff,
                  _lib.stream_ptr())
        return out

    def overlap(self):
        if self.layer.shared:
            self.shared = self._glu(self.xf, self.layer.shared_expert)

    def experts(self, rows, recv_counts, n_recv):
        y = torch.empty(n_recv, self.d, device=rows.device, dtype=torch.float32)
        if n_recv == 0:
            return y
        lo = self.rank * self.e_local
        w = [None if t is None else t[lo:lo + self.e_local].contiguous() for t in self.layer._stacked_expert_weights()]
        world = recv_counts.shape[0]
        scratch = torch.empty(_lib.call("amt_moe_ep_expert_scratch_floats", n_recv, self.d, self.dff, self.e_local), device=rows.device,
                              dtype=torch.float32)
        p = _lib.ptr
        _lib.call("amt_moe_ep_expert_fwd", p(rows.contiguous()), p(recv_counts.contiguous()), world, self.e_local, n_recv, *[p(t) for t in w], p(y),
                  p(scratch), self.d, self.dff, _lib.stream_ptr())
        return y

    def combine(self, y_sorted, slot_pos):
        out = torch.empty(self.n_tok, self.d, device=self.xf.device, dtype=torch.float32)
        p = _lib.ptr
        _lib.call("amt_moe_combine_fwd", p(y_sorted.contiguous()), p(slot_pos), p(self.idx), p(self.wts), p(self.shared),
                  1.0 / self.layer.n_experts_per_token, p(out), self.n_tok, self.d, _lib.stream_ptr())
        return out


class GLUExpert(nn.Module):
    def __init__(self, d_model, d_ff=2048, dropout=0.1):
        super().__init__()
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.gate = nn.Linear(d_model, d_ff)
        self.dropout = nn.Dropout(dropout)


class SiLUExpert(nn.Sequential):
    """The other expert of the V1 family (video_music_transformer.py:80-85): ``nn.Sequential(Linear(d, d_ff), SiLU, Dropout,
    Linear(d_ff, d))`` -- state_dict keys ``0.weight, 0.bias, 3.weight, 3.bias``."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__(nn.Linear(d_model, d_ff), nn.SiLU(), nn.Dropout(dropout), nn.Linear(d_ff, d_model))


def expert_parts(e):
    """{slot: nn.Linear or None} of an expert in the library's terms: y = W2 ((W1 x + b1) * silu(Wg x + bg)) + b2, with
    linear1 = None for a SiLUExpert (y = W2 silu(Wg x + bg) + b2: its first Linear sits in the gate slot)."""
    if isinstance(e, SiLUExpert):
        return {"linear1": None, "gate": e[0], "linear2": e[3]}
    return {"linear1": e.linear1, "gate": e.gate, "linear2": e.linear2}


def expert_dff(e):
    """Hidden width as the kernels see it: a multiple of 32 (the GEMM's K step).  An odd width -- the regression head's
    GLUExpert(d, 2d + 1), video_regression.py:149,166,174 -- is zero-padded: the extra hidden units compute
    (0 + 0) * silu(0) = 0 and meet zero columns of linear2, so the result is unchanged."""
    return (expert_parts(e)["gate"].out_features + 31) // 32 * 32


def _slot_tensor(lin, name, field, width):
    """weight / bias of one slot, hidden dimension zero-padded to `width`."""
    t = getattr(lin, field).detach()
    have = lin.out_features if name != "linear2" else lin.in_features
    if have == width or (name == "linear2" and field == "bias"):
        return t.contiguous()
    if name == "linear2":                                   # (d, dff): pad columns
        out = torch.zeros(t.shape[0], width, dtype=t.dtype, device=t.device)
        out[:, :have] = t
    else:                                                   # (dff, d) or (dff,): pad rows
        out = torch.zeros((width,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        out[:have] = t
    return out


def expert_tensors(e):
    """(w1, b1, wg, bg, w2, b2) contiguous, None where the expert has no such tensor."""
    q, width = expert_parts(e), expert_dff(e)
    out = []
    for name in ("linear1", "gate", "linear2"):
        out += [None, None] if q[name] is None else [_slot_tensor(q[name], name, "weight", width), _slot_tensor(q[name], name, "bias", width)]
    return out


def _stack(mods, attr, field):
    lins = [expert_parts(m)[attr] for m in mods]
    if lins[0] is None:
        return None
    width = expert_dff(mods[0])
    return torch.stack([_slot_tensor(l, attr, field, width) for l in lins]).contiguous()


class TopKScheduler(nn.Module):
    """Reference ``TopKScheduler`` (model/moe.py:66-82): k falls from n_experts to the minimum, one per `update_step` calls.
    The layers consult it in training mode only."""

    def __init__(self, n_experts=8, min_n_experts_per_token=2, update_step=16):
        super().__init__()
        self.n_experts, self.min_n_experts_per_token, self.k = n_experts, min_n_experts_per_token, n_experts
        self.update_step, self.counting_step = update_step, 0

    def step(self):
        self.counting_step += 1
        if self.counting_step % self.update_step == 0:
            self.k = max(self.min_n_experts_per_token, self.k - 1)

    def getK(self):
        return self.k


class TemperatureScheduler(nn.Module):
    """Reference ``TemperatureScheduler`` (model/moe.py:84-97): t climbs from temperature_min by temperature_step per call up to
    temperature_max.  ``SharedMoELayer`` steps it in every forward, eval included (:238-240)."""

    def __init__(self, temperature_min=0.8, temperature_max=1.1, temperature_step=0.0005):
        super().__init__()
        self.temperature_min, self.temperature_max, self.temperature_step = temperature_min, temperature_max, temperature_step
        self.t = self.temperature_min

    def step(self):
        self.t += self.temperature_step
        self.t = min(self.t, self.temperature_max)

    def getT(self):
        return self.t


class _MoEBase(nn.Module):
    shared = False
    ep_group = None
    expert_parallel = False

    def enable_expert_parallel(self, group=None):
        """Config 5: rank r runs experts [r*E/W, (r+1)*E/W); token rows travel by all_to_all (RCCL over xGMI)."""
        assert dist.is_initialized(), "expert parallelism needs an initialised process group"
        self.expert_parallel, self.ep_group = True, group
        return self

    def _run_ep(self, x, phases=None):
        L, B, d = x.shape
        n_tok, n_exp = L * B, self.n_experts
        xf = x.to(torch.float32).contiguous().view(n_tok, d)
        p, st = _lib.ptr, _lib.stream_ptr
        if phases: phases.mark("start")
        idx = torch.empty(n_tok, 2, device=x.device, dtype=torch.int32)
        wts = torch.empty(n_tok, 2, device=x.device, dtype=torch.float32)
        gw, gb = self.gate.weight.detach().contiguous(), self.gate.bias.detach().contiguous()
        t = self._temperature()
        if t != 1.0:
            gw, gb = (gw / t).contiguous(), (gb / t).contiguous()
        _lib.call("amt_moe_route_fwd", p(xf), p(gw), p(gb), p(idx), p(wts), n_tok, d, n_exp, st())
        world, rank = dist.get_world_size(self.ep_group), dist.get_rank(self.ep_group)
        ops = _HipEpOps(self, xf, idx, wts, rank, n_exp // world)
        out = expert_parallel_moe(xf, idx, ops, n_exp, self.ep_group, phases)
        self.last_routing = (idx.view(L, B, 2), wts.view(L, B, 2))
        return out.view(L, B, d)

    def _temperature(self):
        """Routing temperature of this forward: 1 unless the layer steps a temperature scheduler in eval mode (SharedMoELayer)."""
        return 1.0

    def _stacked_expert_weights(self):
        """(n_exp, ...) contiguous copies of the experts' tensors for the grouped GEMMs, rebuilt only when a
        parameter changed (the library takes one base pointer + a per-expert stride)."""
        sig = tuple((p.data_ptr(), p._version) for e in self.experts for p in e.parameters())
        if getattr(self, "_stack_sig", None) != sig:
            self._stack = (_stack(self.experts, "linear1", "weight"), _stack(self.experts, "linear1", "bias"),
                           _stack(self.experts, "gate", "weight"), _stack(self.experts, "gate", "bias"),
                           _stack(self.experts, "linear2", "weight"), _stack(self.experts, "linear2", "bias"))
            self._stack_sig = sig
        return self._stack

    def _run(self, x):
        if self.training:
            raise NotImplementedError("MoE training (schedulers, balancing updates) is outside the hot path")
        if x.dim() != 3:
            raise ValueError("MoE layers take (L, B, d) input (moe.py:193,292 unpack three indices)")
        if x.device.type != "cuda":
            raise _lib.AmtError("MoE layers run on an MI355X only; video2music_amd has no CPU fallback")
        k = int(self.n_experts_per_token)
        if not 1 <= k <= min(8, self.n_experts):
            raise NotImplementedError(f"n_experts_per_token={k}: the gfx950 MoE path routes to 1..8 experts per token")
        if self.expert_parallel:
            if k != 2:
                raise NotImplementedError("expert-parallel execution is built for top-2 routing (the class default)")
            return self._run_ep(x)
        L, B, d = x.shape
        n_tok, n_exp = L * B, self.n_experts
        dff = expert_dff(self.experts[0])
        xf = x.to(torch.float32).contiguous()
        p = _lib.ptr
        w1, b1, wg, bg, w2, b2 = self._stacked_expert_weights()
        sh = [None] * 6
        if self.shared:
            sh = expert_tensors(self.shared_expert)
        out = torch.empty(n_tok, d, device=x.device, dtype=torch.float32)
        idx = torch.empty(n_tok, k, device=x.device, dtype=torch.int32)
        wts = torch.empty(n_tok, k, device=x.device, dtype=torch.float32)
        scratch = torch.empty(_lib.call("amt_moe_topk_scratch_floats", n_tok, d, dff, n_exp, k), device=x.device, dtype=torch.float32)
        gw, gb = self.gate.weight.detach().contiguous(), self.gate.bias.detach().contiguous()
        t = self._temperature()
        if t != 1.0:
            # softmax(top-2 logits / t) (moe.py:288): the top-2 of logits / t are the top-2 of the logits, so the temperature is
            # folded into the router's weight and bias
            gw, gb = (gw / t).contiguous(), (gb / t).contiguous()
        _lib.call("amt_moe_topk_fwd", p(xf), p(gw), p(gb), p(w1), p(b1), p(wg), p(bg), p(w2), p(b2),
                  *[p(t) for t in sh], p(out), p(idx), p(wts), p(scratch), n_tok, d, dff, n_exp, k, _lib.stream_ptr())
        self.last_routing = (idx.view(L, B, k), wts.view(L, B, k))
        return out.view(L, B, d)


class MoELayer(_MoEBase):
    def __init__(self, expert, d_model, n_experts=8, n_experts_per_token=2, dropout=0.1, topk_scheduler=None,
                 temperature_scheduler=None):
        super().__init__()
        self.n_experts, self.n_experts_per_token, self.d_model = n_experts, n_experts_per_token, d_model
        self.dropout = nn.Dropout(dropout)
        self.experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(n_experts)])    # _get_clones, :157
        self.gate = nn.Linear(d_model, n_experts)
        # moe.py:167-179: both schedulers act under `self.training` only; this build is inference-only (forward refuses
        # training mode), so they are kept as attributes like the reference does and have no effect
        if topk_scheduler is not None:
            self.topk_scheduler = topk_scheduler
        if temperature_scheduler is not None:
            self.temperature_scheduler = temperature_scheduler

    def forward(self, x):
        return self._run(x)


class SharedMoELayer(_MoEBase):
    shared = True

    def __init__(self, expert, d_model, n_experts=8, n_experts_per_token=2, dropout=0.1, balancing=False,
                 topk_scheduler=None, temperature_scheduler=None, use_KAN=False):
        super().__init__()
        if use_KAN:
            raise NotImplementedError("KAN gates need efficient_kan (absent); outside the hot path")
        if temperature_scheduler is not None:   # stepped in EVERY forward, eval included (:238-240), and applied at :288
            self.temperature_scheduler = temperature_scheduler
        if topk_scheduler is not None:          # training-only (:232-236)
            self.topk_scheduler = topk_scheduler
        self.n_experts, self.n_experts_per_token, self.d_model = n_experts, n_experts_per_token, d_model
        self.dropout = nn.Dropout(dropout)
        self.experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(n_experts)])    # :209
        self.balancing = balancing
        self.gate = nn.Linear(d_model, n_experts)
        if balancing:                           # routing bias: in the state_dict, ignored in eval (:257-268)
            self.register_buffer("bias", torch.zeros((n_experts, 1)))
            self.update_rate = 0.001
        self.shared_expert = copy.deepcopy(expert)                                         # :229

    def _temperature(self):
        if hasattr(self, "temperature_scheduler"):
            self.temperature_scheduler.step()
            return float(self.temperature_scheduler.getT())
        return 1.0

    def forward(self, x):
        return self._run(x)
