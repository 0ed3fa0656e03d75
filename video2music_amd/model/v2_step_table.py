"""Pointer table, packed / folded weights and K/V caches of the cached decode step of the V1 / V2 families
(``VideoMusicTransformer_V2._cache_init``; consumed by ``amt_v2_step`` / ``amt_v2_step_batch`` / ``amt_v2_step_decide_batch``,
layout in ``include/amt_hip.h``).  Split out of ``video_music_transformer.py``: this is load-time plumbing -- which tensors the step
reads, in which packed form, and which LayerNorms are folded through which projection -- not model code."""
import ctypes as C
import os

import torch
import torch.nn as nn

from .. import _lib
from ..utilities.constants import CHORD_SIZE


def build_step_table(self, memory, S):
    """`memory`: one clip's encoder output (S, E), or a list of them for the lockstep step of several clips
    (`generate_batch`: every cache then carries a leading clip dimension)."""
    from .. import ops
    E, H = self.d_model, self.nhead
    hd = E // H
    mems = list(memory) if isinstance(memory, (list, tuple)) else [memory]
    nb = len(mems)
    dev = mems[0].device
    st = {"cross": [], "self": [], "S": S, "B": nb}
    rows = mems[0] if nb == 1 else torch.cat(mems)                  # (nb*S, E): one projection launch for all clips
    cross_all = []
    for lyr in self.transformer.decoder.layers:
        a = lyr.cross_attn
        W, b = a.in_proj_weight.detach(), a.in_proj_bias.detach()
        k = ops.linear(rows, W[E:2 * E], b[E:2 * E])
        if self._rope_cache is not None and self._rope_cache.shape[1] * 2 == E:
            # per clip the B = 1 view, positions 0..S-1 = pair i of the E-wide row at position s rotated by cache[s][i]
            # (`_attention` does the same for the encoder): one launch for all clips
            ops.rope(k.view(nb, S, 1, E), self._rope_cache, out=k.view(nb, S, 1, E))
        elif self._rope_cache is not None:                         # a cache built for another width: clip by clip
            for c in range(nb):
                kc_ = k[c * S:(c + 1) * S]
                ops.rope(kc_.view(H, S, 1, hd), self._rope_cache, out=kc_.view(H, S, 1, hd))
        v = ops.linear(rows, W[2 * E:], b[2 * E:])
        cross_all.append((k, v))
        st["cross"].append((k[:S], v[:S]))
        st["self"].append((torch.empty(self._max_dec, E, device=dev), torch.empty(self._max_dec, E, device=dev)))
    # pointer table of amt_v2_step (include/amt_hip.h); `keep` holds every tensor the table points into
    keep, ptrs = [], []

    def add(t):
        if t is None:
            ptrs.append(None)
        else:
            t = t.detach().contiguous()
            keep.append(t)
            ptrs.append(t.data_ptr())

    cache = self.__dict__.setdefault("_pack_cache", {})

    def cached(ident, version, build):
        """One entry per identity (storage pointers + role); a new parameter version (load_state_dict, an optimiser step
        followed by eval) REPLACES the entry, so stale packed copies do not stay resident."""
        hit = cache.get(ident)
        if hit is None or hit[0] != version:
            cache[ident] = hit = (version, build())
        return hit[1]

    def pack_now(w, rows=None):
        src = (w if rows is None else w[:rows]).detach().contiguous()
        N, K = src.shape
        out = torch.empty((N + 15) // 16 * 16 * K, device=dev, dtype=torch.float32)
        _lib.call("amt_pack_weight_fwd", _lib.ptr(src), _lib.ptr(out), N, K, _lib.stream_ptr())
        return out

    def packed(w, rows=None):
        """Weight (N, K) [or its first `rows` rows] in the skinny GEMM's tile order; packed once per parameter version."""
        w = w.detach()
        return cached((w.data_ptr(), rows), w._version, lambda: pack_now(w, rows))

    def packed_experts(experts, name):
        # (the per-expert packed pieces are temporaries: only the concatenation stays resident)
        lins = [expert_parts(e)[name] for e in experts]
        return cached(tuple(l.weight.data_ptr() for l in lins) + (name,), tuple(l.weight._version for l in lins),
                      lambda: torch.cat([pack_now(l.weight) for l in lins]))

    def nbias(n):       # a norm's bias; None (RMSNorm) selects the RMS form inside the step
        return getattr(n, "bias", None)

    for t in (self._PR, self._PA, self._wkey, self.Linear_chord.bias, self._rope_cache, self.transformer.decoder.norm.weight,
              nbias(self.transformer.decoder.norm), packed(self.Wout.weight), self.Wout.bias,
              torch.tensor([0, 1], device=dev, dtype=torch.int32), self._pe_chord if self._learned_pos else None):
        add(t)
    from .moe import GLUExpert, SiLUExpert, _stack, expert_dff, expert_parts
    dff, widths = None, set()

    def add_expert(e):
        """linear1 w, b (None for a SiLUExpert), gate w, b, linear2 w, b -- packed weights."""
        q = expert_parts(e)
        for name in ("linear1", "gate", "linear2"):
            add(None if q[name] is None else packed(q[name].weight))
            add(None if q[name] is None else q[name].bias)

    # the one-call step streams K/V with the decode-attention kernel: head-major caches (H, rows, hd)
    st["self_hm"] = [(torch.empty(nb, H, self._max_dec, hd, device=dev), torch.empty(nb, H, self._max_dec, hd, device=dev))
                     for _ in st["self"]]
    st["cross_hm"] = [(k.view(nb, S, H, hd).permute(0, 2, 1, 3).contiguous(), v.view(nb, S, H, hd).permute(0, 2, 1, 3).contiguous())
                      for k, v in cross_all]
    layers = list(self.transformer.decoder.layers)
    for li, (lyr, (kc, vc), (kx, vx)) in enumerate(zip(layers, st["self_hm"], st["cross_hm"])):
        sa, ca = lyr.self_attn, lyr.cross_attn
        for t in (packed(sa.in_proj_weight), sa.in_proj_bias, packed(sa.out_proj.weight), sa.out_proj.bias, lyr.norm1.weight, nbias(lyr.norm1),
                  packed(ca.in_proj_weight, rows=E), ca.in_proj_bias, packed(ca.out_proj.weight), ca.out_proj.bias, lyr.norm2.weight,
                  nbias(lyr.norm2), lyr.norm3.weight, nbias(lyr.norm3), kc, vc, kx, vx):
            add(t)
        ff = lyr.ff

        def stacked(mods, with_down):
            """[gate of every module | linear1 of every module] as ONE packed matrix (rows interleaved in eights when linear1 exists)
            + bias (the lockstep step's single gate/up product), and, for a mixture layer, linear2 of every module one after the
            other + biases."""
            parts = [expert_parts(e) for e in mods]
            srcs = [q[n] for n in ("gate", "linear1", "linear2") for q in parts if q[n] is not None]

            def build():
                gu = [q["gate"] for q in parts] + [q["linear1"] for q in parts if q["linear1"] is not None]
                if len(gu) == 2 * len(parts):
                    # gate and linear1 rows interleaved in eights (column tile T of the product = gate columns 8T..8T+7 | linear1 columns
                    # 8T..8T+7): the launch's epilogue then holds both halves of a hidden column and writes up * silu(gate) itself
                    # (DecodeGemmParams::glu_pair, csrc/kernels.h); the bias keeps the stacked order
                    g_rows = torch.cat([q["gate"].weight.detach() for q in parts])
                    u_rows = torch.cat([q["linear1"].weight.detach() for q in parts])
                    w = torch.stack([g_rows.view(-1, 8, g_rows.shape[1]), u_rows.view(-1, 8, u_rows.shape[1])], dim=1)
                    wp = pack_now(w.reshape(-1, g_rows.shape[1]).contiguous())
                else:
                    wp = torch.cat([pack_now(l.weight) for l in gu])
                out = [wp, torch.cat([l.bias.detach() for l in gu]).contiguous()]
                if with_down:
                    out += [torch.cat([pack_now(q["linear2"].weight) for q in parts]),
                            torch.cat([q["linear2"].bias.detach() for q in parts]).contiguous()]
                else:
                    out += [None, None]
                return out

            return cached(tuple((l.weight.data_ptr(), l.bias.data_ptr()) for l in srcs) + ("stacked", with_down),
                          tuple((l.weight._version, l.bias._version) for l in srcs), build)

        if isinstance(ff, (GLUExpert, SiLUExpert)):
            layer_dff = expert_dff(ff)
            add(None), add(None)
            add_expert(ff)
            for _ in range(6):
                add(None)
            for t in stacked([ff], False):
                add(t)
        else:
            if ff.n_experts_per_token != 2 or getattr(ff, "expert_parallel", False):
                raise NotImplementedError("the cached V2 step is built for local top-2 MoE layers")
            if hasattr(ff, "temperature_scheduler"):
                # the scheduler steps in every forward (moe.py:238-240) and rescales the routing logits; the cached step hands
                # the raw gate to the device, so such a layer must take the per-step re-forward
                raise NotImplementedError("the cached V2 step does not evaluate a SharedMoELayer temperature_scheduler")
            layer_dff = expert_dff(ff.experts[0])
            add(ff.gate.weight), add(ff.gate.bias)
            for name in ("linear1", "gate", "linear2"):
                if expert_parts(ff.experts[0])[name] is None:
                    add(None), add(None)
                else:
                    add(packed_experts(ff.experts, name))
                    add(_stack(ff.experts, name, "bias"))
            if ff.shared:
                add_expert(ff.shared_expert)
            else:
                for _ in range(6):
                    add(None)
            for t in stacked(list(ff.experts) + ([ff.shared_expert] if ff.shared else []), True):
                add(t)
        # out-projection of the self-attention and the cross-attention's query projection in ONE launch of the lockstep step,
        # norm1 folded through the projection (the base model's G1): with u = x + o Wo^T + bo the query is
        # LayerNorm(u) Wq^T + bq = ((u (Wq o gamma)^T) - mean g) rstd + c, and u (Wq o gamma)^T = [o | x] [Wq' Wo | Wq']^T + Wq' bo.
        # The launch produces u and that raw product; the attention kernel finishes the query with u's row statistics.
        if isinstance(lyr.norm1, nn.LayerNorm) and 2 * E <= 1536 and os.environ.get("AMT_V2_FOLD_G1", "1") != "0":
            from .. import ops
            srcs = (sa.out_proj.weight, sa.out_proj.bias, ca.in_proj_weight, ca.in_proj_bias, lyr.norm1.weight, lyr.norm1.bias)

            def build_fold(sa=sa, ca=ca, lyr=lyr):
                Wo, bo, Wq, bq = sa.out_proj.weight.detach(), sa.out_proj.bias.detach(), ca.in_proj_weight.detach()[:E], ca.in_proj_bias.detach()[:E]
                gamma, beta = lyr.norm1.weight.detach(), lyr.norm1.bias.detach()
                Wqg = (Wq * gamma.unsqueeze(0)).contiguous()                              # Wq o gamma
                A = ops.linear(Wqg, Wo.t().contiguous())                                   # (Wq o gamma) Wo
                P2 = torch.cat([A, Wqg], dim=1).contiguous()                               # (E, 2E), a temporary
                return [pack_now(P2), ops.linear(bo.view(1, E).contiguous(), Wqg).view(E).contiguous(),
                        Wqg.sum(dim=1).contiguous(), (ops.linear(beta.view(1, E).contiguous(), Wq.contiguous()).view(E) + bq).contiguous()]

            for t in cached(tuple(t.data_ptr() for t in srcs) + ("g1fold",), tuple(t._version for t in srcs), build_fold):
                add(t)
        else:
            for _ in range(4):
                add(None)
        # A plain GLU layer with norm3 folded through the NEXT layer's QKV projection (the base model's G3, DESIGN.md section 5: the
        # down projection also emits the raw QKV product, the next self-attention finishes it: 6 launches per layer instead of 7)
        # and, with AMT_V2_FOLD_FFN=2, norm2 through its stacked gate | up product as well (the base model's G2: 5 launches).
        #   LayerNorm(u2) Wgu^T + bgu = ((u2 (Wgu o gamma2)^T) - mean g) rstd + c,  u2 (Wgu o gamma2)^T = [o | x] [W' Wo | W']^T + W' bo
        #   LayerNorm(u3) Wn^T + bn likewise with u3 = h W2^T + b2 + LayerNorm(u2): [h | xn2] [W'' W2 | W'']^T + W'' b2
        # Widths: whole 256-column chunks on either side of the down projection's staged row [h | xn2] (K = dff + E <= 1536).
        parts = expert_parts(ff) if isinstance(ff, GLUExpert) else None
        fold_mode = int(os.environ.get("AMT_V2_FOLD_FFN", "1"))       # 1 (default): norm3 -> next QKV only; 2: norm2 -> gate | up as well; 0: off
        nxt = layers[li + 1] if li + 1 < len(layers) else None
        fold_ffn = (fold_mode > 0 and parts is not None and parts["linear1"] is not None
                    and all(isinstance(n, nn.LayerNorm) for n in (lyr.norm2, lyr.norm3))
                    and E % 256 == 0 and layer_dff % 256 == 0 and 2 * E <= 1536 and layer_dff + E in (512, 768, 1024, 1536)
                    and (layer_dff + E < 1536 or E >= 512))
        if fold_ffn and fold_mode < 2 and nxt is None:
            fold_ffn = False
        if fold_ffn:
            srcs = [ca.out_proj.weight, ca.out_proj.bias, parts["gate"].weight, parts["gate"].bias, parts["linear1"].weight, parts["linear1"].bias,
                    parts["linear2"].weight, parts["linear2"].bias, lyr.norm2.weight, lyr.norm2.bias, lyr.norm3.weight, lyr.norm3.bias]
            if nxt is not None:
                srcs += [nxt.self_attn.in_proj_weight, nxt.self_attn.in_proj_bias]

            def build_ffn_fold(ca=ca, lyr=lyr, parts=parts, nxt=nxt, mode=fold_mode):
                row = lambda v: v.detach().view(1, -1).contiguous()
                out = [None] * 4
                if mode >= 2:         # measured slower than the separate launches at d_model 512 (profiles/r03_v2_fold_ab.json): kept for A/B
                    Wo, bo = ca.out_proj.weight.detach(), ca.out_proj.bias.detach()
                    Wgu = torch.cat([parts["gate"].weight.detach(), parts["linear1"].weight.detach()]).contiguous()     # (2 dff, E): gate rows first
                    bgu = torch.cat([parts["gate"].bias.detach(), parts["linear1"].bias.detach()])
                    Wp = (Wgu * lyr.norm2.weight.detach().unsqueeze(0)).contiguous()                                   # Wgu o gamma2
                    P2 = torch.cat([ops.linear(Wp, Wo.t().contiguous()), Wp], dim=1).contiguous()                     # (2 dff, 2E), a temporary
                    out = [pack_now(P2), ops.linear(row(bo), Wp).view(-1).contiguous(), Wp.sum(dim=1).contiguous(),
                           (ops.linear(row(lyr.norm2.bias), Wgu).view(-1) + bgu).contiguous()]
                if nxt is None:
                    return out + [None] * 4
                Wn, bn = nxt.self_attn.in_proj_weight.detach(), nxt.self_attn.in_proj_bias.detach()
                W2, b2 = parts["linear2"].weight.detach(), parts["linear2"].bias.detach()                         # (E, dff)
                Wq = (Wn * lyr.norm3.weight.detach().unsqueeze(0)).contiguous()                                    # Wqkv' o gamma3
                P3 = torch.cat([ops.linear(Wq, W2.t().contiguous()), Wq], dim=1).contiguous()                      # (3E, dff + E)
                return out + [pack_now(P3), ops.linear(row(b2), Wq).view(-1).contiguous(), Wq.sum(dim=1).contiguous(),
                              (ops.linear(row(lyr.norm3.bias), Wn.contiguous()).view(-1) + bn).contiguous()]

            for t in cached(tuple(t.data_ptr() for t in srcs) + ("ffnfold", fold_mode), tuple(t._version for t in srcs), build_ffn_fold):
                add(t)
        else:
            for _ in range(8):
                add(None)
        widths.add(layer_dff)
        dff = layer_dff
    st["tab"] = (C.c_void_p * len(ptrs))(*ptrs)
    st["keep"] = keep
    st["dff"] = dff
    # amt_v2_step lays its scratch out for one feed-forward width; layers of different widths (V1 '1.3.3' / '1.3.4' with
    # dim_feedforward != 2 d_model) take the same cached step issued operator by operator (`_decode_step`)
    # the step kernels take widths that are multiples of 64 up to 1536 (amt_v2_step / amt_v2_step_batch); other widths take the
    # operator path too
    st["native"] = len(widths) == 1 and E % 64 == 0 and dff % 64 == 0 and E <= 1536 and dff <= 1536
    # (one clip may run either step: the one-call step with device-routed experts or the lockstep step with B = 1)
    n_ws = max(_lib.call("amt_v2_step_ws_floats", E, dff, self.n_experts) if nb == 1 else 0,
               _lib.call("amt_v2_step_batch_ws_floats", E, dff, self.n_experts, nb))
    st["ws"] = torch.empty(n_ws, device=dev, dtype=torch.float32)
    st["logits"] = torch.empty(CHORD_SIZE, device=dev, dtype=torch.float32) if nb == 1 else torch.empty(nb, CHORD_SIZE, device=dev)
    return st
