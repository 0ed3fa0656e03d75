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


def expert_parallel_moe(x, idx, wts, n_experts, run_local_experts, combine, group=None):
    """Expert-parallel top-2 MoE over ``torch.distributed`` (experts placed per GPU).

    Rank r owns experts ``[r*E/W, (r+1)*E/W)``.  Every rank routes its own tokens (``idx``/``wts`` (n_tok,2)),
    sends each (token, slot) row to the owner of its expert with ONE ``all_to_all_single`` (rows sorted by
    expert, so per-rank segments are contiguous), runs its experts on what it received, returns the
    results with a second ``all_to_all_single`` and combines per token in expert-index order.

    ``run_local_experts(rows, counts_per_local_expert) -> rows`` and ``combine(y_sorted, slot_pos) -> out``
    are supplied by the caller (HIP kernels in production, the CPU oracle in the gloo test).
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert n_experts % world == 0, "experts must divide evenly over ranks"
    e_local = n_experts // world
    n_tok, d = x.shape
    flat_e = idx.reshape(-1).long()
    order = torch.argsort(flat_e, stable=True)                     # sorted (token, slot) assignments
    send_rows = x.index_select(0, order // 2).contiguous()
    counts = torch.bincount(flat_e, minlength=n_experts)            # rows per expert on this rank
    send_counts = counts.view(world, e_local).cpu()
    # 1) sizes: who sends how many rows for which of my experts
    recv_counts = torch.empty_like(send_counts)
    cpu_group_ok = dist.get_backend(group) == "gloo"
    sc_dev = send_counts if cpu_group_ok else send_counts.to(x.device)
    rc_dev = recv_counts if cpu_group_ok else recv_counts.to(x.device)
    dist.all_to_all_single(rc_dev, sc_dev, group=group)
    recv_counts = rc_dev.cpu()
    in_splits = recv_counts.sum(1).tolist()
    out_splits = send_counts.sum(1).tolist()
    # 2) rows to their experts' owners
    stage = (lambda t: t.cpu()) if (cpu_group_ok and x.is_cuda) else (lambda t: t)
    recv_rows = torch.empty(sum(in_splits), d, dtype=x.dtype, device="cpu" if (cpu_group_ok and x.is_cuda) else x.device)
    dist.all_to_all_single(recv_rows, stage(send_rows), output_split_sizes=in_splits, input_split_sizes=out_splits, group=group)
    recv_rows = recv_rows.to(x.device)
    # received rows are grouped by source rank, inside a source by local expert: regroup by local expert
    src_e = torch.repeat_interleave(torch.arange(world * e_local) % e_local, recv_counts.reshape(-1)).to(x.device)
    by_e = torch.argsort(src_e, stable=True)
    per_local = recv_counts.sum(0).tolist()
    y_local = run_local_experts(recv_rows.index_select(0, by_e).contiguous(), per_local)
    y_recv = torch.empty_like(y_local)
    y_recv[by_e] = y_local                                          # back to arrival order
    # 3) results back to the token owners
    y_sorted = torch.empty(send_rows.shape[0], d, dtype=x.dtype, device="cpu" if (cpu_group_ok and x.is_cuda) else x.device)
    dist.all_to_all_single(y_sorted, stage(y_recv.contiguous()), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    y_sorted = y_sorted.to(x.device)
    slot_pos = torch.empty_like(order)
    slot_pos[order] = torch.arange(order.numel(), device=order.device)   # row of assignment (token*2+slot) in y_sorted
    return combine(y_sorted, slot_pos.to(torch.int32).view(n_tok, 2))


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

    def _run_ep(self, x):
        L, B, d = x.shape
        n_tok, n_exp = L * B, self.n_experts
        dff = expert_dff(self.experts[0])
        xf = x.to(torch.float32).contiguous().view(n_tok, d)
        p, st = _lib.ptr, _lib.stream_ptr
        idx = torch.empty(n_tok, 2, device=x.device, dtype=torch.int32)
        wts = torch.empty(n_tok, 2, device=x.device, dtype=torch.float32)
        gw, gb = self.gate.weight.detach().contiguous(), self.gate.bias.detach().contiguous()
        t = self._temperature()
        if t != 1.0:
            gw, gb = (gw / t).contiguous(), (gb / t).contiguous()
        _lib.call("amt_moe_route_fwd", p(xf), p(gw), p(gb), p(idx), p(wts), n_tok, d, n_exp, st())
        world, rank = dist.get_world_size(self.ep_group), dist.get_rank(self.ep_group)
        e_local = n_exp // world

        def glu(rows, e):
            m = self.experts[e] if e is not None else self.shared_expert
            out = torch.empty(rows.shape[0], d, device=rows.device, dtype=torch.float32)
            if rows.shape[0] == 0:
                return out
            scratch = torch.empty(2 * rows.shape[0] * dff, device=rows.device, dtype=torch.float32)
            t = expert_tensors(m)
            _lib.call("amt_glu_expert_fwd", p(rows), *[p(v) for v in t], p(out), p(scratch), rows.shape[0], d, dff, st())
            return out

        def run_local(rows, per_local):
            outs, o = [], 0
            for j, n in enumerate(per_local):
                outs.append(glu(rows[o:o + n].contiguous(), rank * e_local + j))
                o += n
            return torch.cat(outs) if outs else rows

        def combine(y_sorted, slot_pos):
            out = torch.empty(n_tok, d, device=x.device, dtype=torch.float32)
            shared = glu(xf, None) if self.shared else None
            _lib.call("amt_moe_combine_fwd", p(y_sorted.contiguous()), p(slot_pos.contiguous()), p(idx), p(wts), p(shared),
                      1.0 / self.n_experts_per_token, p(out), n_tok, d, st())
            return out

        out = expert_parallel_moe(xf, idx, wts, n_exp, run_local, combine, self.ep_group)
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
