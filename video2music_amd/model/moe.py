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
import torch.nn as nn

from .. import _lib


class GLUExpert(nn.Module):
    def __init__(self, d_model, d_ff=2048, dropout=0.1):
        super().__init__()
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.gate = nn.Linear(d_model, d_ff)
        self.dropout = nn.Dropout(dropout)


def _stack(mods, attr, field):
    return torch.stack([getattr(getattr(m, attr), field).detach() for m in mods]).contiguous()


class _MoEBase(nn.Module):
    shared = False

    def _run(self, x):
        if self.training:
            raise NotImplementedError("MoE training (schedulers, balancing updates) is outside the hot path")
        if x.dim() != 3:
            raise ValueError("MoE layers take (L, B, d) input (moe.py:193,292 unpack three indices)")
        if x.device.type != "cuda":
            raise _lib.AmtError("MoE layers run on an MI355X only; video2music_amd has no CPU fallback")
        if self.n_experts_per_token != 2:
            raise NotImplementedError("the gfx950 MoE path is built for top-2 routing (class default)")
        L, B, d = x.shape
        n_tok, n_exp = L * B, self.n_experts
        dff = self.experts[0].linear1.out_features
        xf = x.to(torch.float32).contiguous()
        p = _lib.ptr
        w1, b1 = _stack(self.experts, "linear1", "weight"), _stack(self.experts, "linear1", "bias")
        wg, bg = _stack(self.experts, "gate", "weight"), _stack(self.experts, "gate", "bias")
        w2, b2 = _stack(self.experts, "linear2", "weight"), _stack(self.experts, "linear2", "bias")
        sh = [None] * 6
        if self.shared:
            e = self.shared_expert
            sh = [t.detach().contiguous() for t in (e.linear1.weight, e.linear1.bias, e.gate.weight, e.gate.bias,
                                                     e.linear2.weight, e.linear2.bias)]
        out = torch.empty(n_tok, d, device=x.device, dtype=torch.float32)
        idx = torch.empty(n_tok, 2, device=x.device, dtype=torch.int32)
        wts = torch.empty(n_tok, 2, device=x.device, dtype=torch.float32)
        scratch = torch.empty(_lib.call("amt_moe_scratch_floats", n_tok, d, dff, n_exp), device=x.device, dtype=torch.float32)
        gw, gb = self.gate.weight.detach().contiguous(), self.gate.bias.detach().contiguous()
        _lib.call("amt_moe_fwd", p(xf), p(gw), p(gb), p(w1), p(b1), p(wg), p(bg), p(w2), p(b2),
                  *[p(t) for t in sh], p(out), p(idx), p(wts), p(scratch), n_tok, d, dff, n_exp, _lib.stream_ptr())
        self.last_routing = (idx.view(L, B, 2), wts.view(L, B, 2))
        return out.view(L, B, d)


class MoELayer(_MoEBase):
    def __init__(self, expert, d_model, n_experts=8, n_experts_per_token=2, dropout=0.1, topk_scheduler=None,
                 temperature_scheduler=None):
        super().__init__()
        self.n_experts, self.n_experts_per_token, self.d_model = n_experts, n_experts_per_token, d_model
        self.dropout = nn.Dropout(dropout)
        self.experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(n_experts)])    # _get_clones, :157
        self.gate = nn.Linear(d_model, n_experts)

    def forward(self, x):
        return self._run(x)


class SharedMoELayer(_MoEBase):
    shared = True

    def __init__(self, expert, d_model, n_experts=8, n_experts_per_token=2, dropout=0.1, balancing=False,
                 topk_scheduler=None, temperature_scheduler=None, use_KAN=False):
        super().__init__()
        if use_KAN:
            raise NotImplementedError("KAN gates need efficient_kan (absent); outside the hot path")
        self.n_experts, self.n_experts_per_token, self.d_model = n_experts, n_experts_per_token, d_model
        self.dropout = nn.Dropout(dropout)
        self.experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(n_experts)])    # :209
        self.balancing = balancing
        self.gate = nn.Linear(d_model, n_experts)
        if balancing:                           # routing bias: in the state_dict, ignored in eval (:257-268)
            self.register_buffer("bias", torch.zeros((n_experts, 1)))
            self.update_rate = 0.001
        self.shared_expert = copy.deepcopy(expert)                                         # :229

    def forward(self, x):
        return self._run(x)
