"""`VideoRegression` (reference model/video_regression.py:104-245) for the Mamba regModels: 'bimamba+' -- the default
regression head of both callers (utilities/argument_generate_funcs.py:87-91, video2music.py:651) --, 'bimamba' (the same
encoder over the original Mamba gate), and the one-directional stacks 'mamba' / 'mamba+' (mamba.py:66-100,131-147:
x + MambaBlock(RMSNorm(x)) per layer), 'moe_bimamba+' / 'sharedmoe_bimamba+' (a mixture layer in the FFN's place), and the
recurrent heads 'lstm' / 'bilstm' / 'gru' / 'bigru' / 'cnngru' / 'cnnbigru' (torch's nn.LSTM / nn.GRU, kept as parameter
containers; the recurrence runs in csrc/rnn.hip).  Per video frame it predicts (note density, loudness) and 40 instrument
probabilities from cat(semantic, emotion) — SURVEY.md §8 row f2.

The module keeps the reference's parameter names (so `load_state_dict(torch.load(path))` works) and composes the
library's kernels; no torch arithmetic runs in `forward`:

    cat + zero-pad            amt_concat2_fwd
    every nn.Linear           amt_linear_ex_fwd (bias / residual / ReLU / sigmoid in the GEMM epilogue)
    Conv1d + SiLU             amt_dwconv1d_silu_fwd
    softplus, scan, gate      amt_selective_scan_fwd
    LayerNorms                amt_layernorm_post_fwd (residual in, `x_f + x_b` out)
    LSTM / GRU recurrence     amt_rnn_seq_fwd (both directions of a layer in one launch)

The backward Mamba block runs with `reverse=1` instead of `torch.flip` before and after (bimamba.py:171-185).
"""
import math

import torch
import torch.nn as nn

from .. import ops

INSTRUMENT_SIZE = 40            # utilities/constants.py:84


class _MambaBlockParams(nn.Module):
    """Parameters of MambaBlock (mamba.py:160-231) with bias=True, conv_bias=True, no inner layernorms."""

    def __init__(self, d_model, d_state=16, expand=2, d_conv=4):
        super().__init__()
        self.d_inner, self.d_state, self.d_conv = expand * d_model, d_state, d_conv
        self.dt_rank = math.ceil(d_model / 16)
        self.in_proj = nn.Linear(d_model, 2 * self.d_inner)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, d_conv, groups=self.d_inner, padding=d_conv - 1)
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner)
        self.A_log = nn.Parameter(torch.log(torch.arange(1, d_state + 1, dtype=torch.float32).repeat(self.d_inner, 1)))
        self.D = nn.Parameter(torch.ones(self.d_inner))
        self.out_proj = nn.Linear(self.d_inner, d_model)


class _BiMambaLayerParams(nn.Module):
    """BiMambaEncoderLayer_V1 (bimamba.py:102-133): plain FFN, or a copy of `moe` (a mixture layer) in its place."""

    def __init__(self, d_model, d_hidden, moe=None):
        super().__init__()
        self.mamba_forward = _MambaBlockParams(d_model)
        self.mamba_backward = _MambaBlockParams(d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        if moe is None:
            self.ffn = nn.Sequential(nn.Linear(d_model, d_hidden), nn.ReLU(), nn.Dropout(0.0), nn.Linear(d_hidden, d_model))
        else:
            import copy
            self.ffn = copy.deepcopy(moe)


class _RMSWeight(nn.Module):
    """mamba.py's RMSNorm (:472-489): a gain vector, eps 1e-5."""

    def __init__(self, d_model, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d_model))


class _ResidualBlockParams(nn.Module):
    """ResidualBlock (mamba.py:131-147): keys mixer.*, norm.weight."""

    def __init__(self, d_model, d_state=16, d_conv=4):
        super().__init__()
        self.mixer = _MambaBlockParams(d_model, d_state=d_state, d_conv=d_conv)
        self.norm = _RMSWeight(d_model)


class _ResidualMoEParams(nn.Module):
    """ResidualMoE (mamba.py:117-129): keys moe_layer.*, norm.weight."""

    def __init__(self, moe, d_model):
        super().__init__()
        import copy
        self.moe_layer = copy.deepcopy(moe)
        self.norm = _RMSWeight(d_model)


class _MoEMambaParams(nn.Module):
    """MoEMamba (mamba.py:102-115): layers.{i}.0 = ResidualBlock, layers.{i}.1 = ResidualMoE."""

    def __init__(self, moe, d_model, d_state, d_conv, n_layers):
        super().__init__()
        self.layers = nn.ModuleList([nn.Sequential(_ResidualBlockParams(d_model, d_state, d_conv), _ResidualMoEParams(moe, d_model))
                                     for _ in range(n_layers)])


class _MambaStackParams(nn.Module):
    def __init__(self, d_model, n_layers):
        super().__init__()
        self.layers = nn.ModuleList([_ResidualBlockParams(d_model) for _ in range(n_layers)])


class _BiMambaLayerV0Params(nn.Module):
    """BiMambaEncoderLayer (bimamba.py:33-59), the layer of use_version 0 ('bimamba'): one FFN + norm pair per direction."""

    def __init__(self, d_model, d_hidden):
        super().__init__()
        self.mamba_forward = _MambaBlockParams(d_model)
        self.mamba_backward = _MambaBlockParams(d_model)
        self.norm1, self.norm2, self.norm3, self.norm4 = (nn.LayerNorm(d_model) for _ in range(4))
        self.ffn1 = nn.Sequential(nn.Linear(d_model, d_hidden), nn.ReLU(), nn.Dropout(0.0), nn.Linear(d_hidden, d_model))
        self.ffn2 = nn.Sequential(nn.Linear(d_model, d_hidden), nn.ReLU(), nn.Dropout(0.0), nn.Linear(d_hidden, d_model))


class _BiMambaEncoderParams(nn.Module):
    def __init__(self, d_model, d_hidden, n_layers, version=1, moe=None):
        super().__init__()
        if version == 1:                                                                # bimamba.py:16-19
            self.layers = nn.ModuleList([_BiMambaLayerParams(d_model, d_hidden, moe) for _ in range(n_layers)])
        else:
            self.layers = nn.ModuleList([_BiMambaLayerV0Params(d_model, d_hidden) for _ in range(n_layers)])


class VideoRegression(nn.Module):
    def __init__(self, n_layers=2, d_model=64, d_hidden=1024, dropout=0.1, use_KAN=False, max_sequence_video=300,
                 total_vf_dim=0, regModel="bilstm", scene_embed=False, chord_embed=False):
        super().__init__()
        if regModel not in ("bimamba+", "bimamba", "mamba", "mamba+", "moe_bimamba+", "sharedmoe_bimamba+", "lstm", "bilstm", "gru", "bigru",
                            "cnngru", "cnnbigru", "moemamba"):
            raise NotImplementedError("built: every regModel of video_regression.py:124-178 ('bimamba+' is the callers' default); 'minGRU' has "
                                      "no branch there either")
        self._version = 1 if regModel.endswith("+") else 0           # MambaConfig.use_version: 1 = the Mamba+ gate
        self._bidirectional = "bimamba" in regModel
        moe = None
        if regModel in ("moe_bimamba+", "sharedmoe_bimamba+", "moemamba"):     # GLUExpert(d, 2d + 1), 6 experts, top-2 (:143-178)
            from .moe import GLUExpert, MoELayer, SharedMoELayer
            cls = MoELayer if regModel == "moe_bimamba+" else SharedMoELayer
            moe = cls(GLUExpert(d_model, d_model * 2 + 1), d_model, n_experts=6, n_experts_per_token=2, dropout=dropout)
        if use_KAN or scene_embed or chord_embed:
            raise NotImplementedError("use_KAN / scene_embed / chord_embed are outside this path")
        if d_model % 32 != 0 or d_hidden % 32 != 0:
            raise ValueError("d_model and d_hidden must be multiples of 32 (GEMM K step)")
        self.n_layers, self.d_model, self.d_hidden = n_layers, d_model, d_hidden
        self.max_seq_video, self.total_vf_dim, self.regModel = max_sequence_video, total_vf_dim, regModel
        self._rnn = regModel in ("lstm", "bilstm", "gru", "bigru", "cnngru", "cnnbigru")
        if self._rnn:
            # torch's own modules as parameter containers (same keys: weight_ih_l0, ..., *_reverse); their forward is never called
            if d_model > 128 or d_model % 8:
                raise ValueError("the recurrent heads keep W_hh in registers: d_model must be a multiple of 8, at most 128")
            cls = nn.LSTM if "lstm" in regModel else nn.GRU
            self._dirs = 2 if "bi" in regModel else 1
            rnn = cls(d_model, d_model, n_layers, bidirectional=self._dirs == 2, dropout=dropout, batch_first=True)
            if regModel.startswith("cnn"):                # CNN_GRU (:84-103): Conv1d(k=7, pad=3) + SiLU, then the GRU
                self.model = nn.Module()
                self.model.cnn = nn.Sequential(nn.Conv1d(d_model, d_model, kernel_size=7, stride=1, padding=3), nn.SiLU(), nn.Dropout(dropout))
                self.model.gru = rnn
            else:
                self.model = rnn
            self.__dict__["_rnn_mod"] = rnn          # an alias outside the module registry (no second set of keys)
        elif regModel == "moemamba":                      # MoEMamba over blocks with d_state = d_hidden, d_conv = 8 (:143-150)
            if d_hidden not in (16, 32, 64, 128, 256):
                raise ValueError("'moemamba' uses d_state = d_hidden: the scan kernel takes 16, 32, 64, 128 or 256 states per channel")
            self.model = _MoEMambaParams(moe, d_model, d_hidden, 8, n_layers)
        else:
            self.model = (_BiMambaEncoderParams(d_model, d_hidden, n_layers, self._version, moe) if self._bidirectional
                          else _MambaStackParams(d_model, n_layers))
        self.in_proj = nn.Sequential(nn.Linear(total_vf_dim, d_model), nn.Dropout(dropout))
        width = d_model * (self._dirs if self._rnn else 1)           # bidirectional recurrent heads: 2 d_model (:201-206)
        self.regressor = nn.Linear(width, 2)
        self.classifier = nn.Sequential(nn.Linear(width, INSTRUMENT_SIZE), nn.Sigmoid())
        self._derived_sig = None

    # zero-padded copies of the weights whose K (or row count) does not fit the GEMM's steps (rebuilt when they change)
    def _derived(self):
        if self._rnn:                                   # only the in-projection needs a padded copy
            w = self.in_proj[0].weight
            sig = (w.data_ptr(), w._version)
            if sig != self._derived_sig:
                self._Fpad = (self.total_vf_dim + 31) // 32 * 32
                self._Win = torch.zeros(self.d_model, self._Fpad, device=w.device)
                self._Win[:, :self.total_vf_dim] = w.detach()
                self._derived_sig = sig
            return
        blocks = ([m for l in self.model.layers for m in (l.mamba_forward, l.mamba_backward)] if self._bidirectional
                  else [(l[0] if isinstance(l, nn.Sequential) else l).mixer for l in self.model.layers])
        ws = [self.in_proj[0].weight] + [m.dt_proj.weight for m in blocks]
        xs = [m.x_proj.weight for m in blocks]
        sig = tuple((w.data_ptr(), w._version) for w in ws + xs)
        if sig != self._derived_sig:
            dev = ws[0].device
            F = self.total_vf_dim
            self._Fpad = (F + 31) // 32 * 32
            self._Win = torch.zeros(self.d_model, self._Fpad, device=dev)
            self._Win[:, :F] = ws[0].detach()
            self._Wdt = []
            for w in ws[1:]:                       # (d_inner, dt_rank) -> (d_inner, 32): the GEMM reads dt | B | C[:..] x zeros
                t = torch.zeros(w.shape[0], 32, device=dev)
                t[:, :w.shape[1]] = w.detach()
                self._Wdt.append(t)
            self._Wx = []
            for w in xs:                           # (dt_rank + 2N, d_inner): rows padded to a multiple of 4 (row stride of dbc)
                t = torch.zeros((w.shape[0] + 3) // 4 * 4, w.shape[1], device=dev)
                t[:w.shape[0]] = w.detach()
                self._Wx.append(t)
            self._derived_sig = sig

    def _conv7_silu(self, x, B, S):
        """CNN_GRU's Conv1d(d, d, kernel 7, padding 3) + SiLU over time (:88-91, :97-100) as seven accumulating GEMMs: the clips
        are laid end to end with 3 zero frames on each side, tap j of output row r reads row r + j; the SiLU rides on the last one."""
        conv = self.model.cnn[0]
        d, K = self.d_model, conv.kernel_size[0]
        pad = K // 2
        W = conv.weight.detach()                                   # (d_out, d_in, K)
        xp = torch.zeros(B, S + 2 * pad, d, device=x.device, dtype=torch.float32)
        xp[:, pad:pad + S] = x.view(B, S, d)
        flat = xp.view(B * (S + 2 * pad), d)
        M = flat.shape[0] - 2 * pad
        acc = None
        for j in range(K):
            acc = ops.linear_ex(flat[j:j + M], W[:, :, j].contiguous(), conv.bias.detach() if j == 0 else None, resid=acc,
                                act=3 if j == K - 1 else 0)
        out = torch.zeros(B, S + 2 * pad, d, device=x.device, dtype=torch.float32)
        out.view(-1, d)[:M] = acc
        return out[:, :S].contiguous().view(B * S, d)

    def _rnn_layer(self, l):
        """(W_ih, b_ih, W_hh, b_hh) of layer l with the directions stacked along the rows; rebuilt when a parameter changes."""
        names = [n + f"_l{l}" + sfx for sfx in (("", "_reverse") if self._dirs == 2 else ("",))
                 for n in ("weight_ih", "bias_ih", "weight_hh", "bias_hh")]
        ps = [getattr(self._rnn_mod, n) for n in names]
        sig = tuple((q.data_ptr(), q._version) for q in ps)
        cache = self.__dict__.setdefault("_rnn_cache", {})
        if cache.get(l, (None,))[0] != sig:
            per = [[q.detach() for q in ps[4 * r:4 * r + 4]] for r in range(self._dirs)]
            cache[l] = (sig, tuple(torch.cat([per[r][k] for r in range(self._dirs)]).contiguous() for k in range(4)))
        return cache[l][1]

    def _mamba(self, x, m, wdt, wx, B, L, resid, reverse):
        """MambaBlock.forward on rows x (B*L, d) + the layer's residual add (out_proj epilogue)."""
        R, N = m.dt_rank, m.d_state
        if R + 2 * N < 32:
            raise ValueError("dt_rank + 2*d_state < 32 is not supported")
        xz = ops.linear_ex(x, m.in_proj.weight.detach(), m.in_proj.bias.detach())
        xc = ops.dwconv1d_silu(xz, m.d_inner, m.conv1d.weight.detach().reshape(m.d_inner, m.d_conv).contiguous(),
                               m.conv1d.bias.detach(), B, L, reverse)
        dbc = ops.linear_ex(xc, wx)
        draw = ops.linear_ex(dbc, wdt, K=32)
        g = ops.selective_scan(xc, draw, m.dt_proj.bias.detach(), m.A_log.detach(), dbc, R, m.D.detach(), xz, B, L,
                               version=self._version, reverse=reverse)
        return ops.linear_ex(g, m.out_proj.weight.detach(), m.out_proj.bias.detach(), resid=resid)

    def get_feature(self, feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion):
        """video_regression.py:199-238: (B, S, d_model) encoder output.  Scene offset and motion are not used."""
        self._derived()
        dev = self.regressor.weight.device
        if dev.type != "cuda":
            raise RuntimeError("video2music_amd has no CPU path: move the module to the GPU")
        sem = feature_semantic_list.to(device=dev, dtype=torch.float32).contiguous()
        emo = feature_emotion.to(device=dev, dtype=torch.float32).contiguous()
        B, S = sem.shape[0], sem.shape[1]
        if sem.shape[2] + emo.shape[2] != self.total_vf_dim:
            raise ValueError(f"semantic ({sem.shape[2]}) + emotion ({emo.shape[2]}) features != total_vf_dim ({self.total_vf_dim})")
        vf = ops.concat2(sem.view(B * S, -1), emo.view(B * S, -1), self._Fpad)
        x = ops.linear_ex(vf, self._Win, self.in_proj[0].bias.detach())
        if self._rnn:                                   # nn.LSTM / nn.GRU (:124-135): per layer and direction, input GEMM + recurrence
            gates, d = (4 if isinstance(self._rnn_mod, nn.LSTM) else 3), self.d_model
            if self._rnn_mod is not self.model:
                x = self._conv7_silu(x, B, S)
            for l in range(self.n_layers):               # both directions: one input GEMM, one recurrence launch
                wi, bi, wh, bh = self._rnn_layer(l)
                out = torch.empty(B * S, self._dirs * d, device=dev, dtype=torch.float32)
                ops.rnn_seq(ops.linear_ex(x, wi, bi), wh, bh, out, 0, B, S, d, gates, n_dirs=self._dirs)
                x = out
            return x.view(B, S, self._dirs * d)
        if not self._bidirectional:                     # Mamba.forward (mamba.py:73-78): x = mixer(norm(x)) + x per layer
            for i, lyr in enumerate(self.model.layers):
                blk = lyr[0] if isinstance(lyr, nn.Sequential) else lyr
                h = ops.rmsnorm(x, blk.norm.weight.detach(), eps=blk.norm.eps)
                x = self._mamba(h, blk.mixer, self._Wdt[i], self._Wx[i], B, S, x, False)
                if blk is not lyr:                      # ResidualMoE: moe_layer(norm(x)) + x (mamba.py:126-128)
                    h = ops.rmsnorm(x, lyr[1].norm.weight.detach(), eps=lyr[1].norm.eps)
                    x = ops.add(lyr[1].moe_layer(h.view(B, S, self.d_model)).reshape(B * S, self.d_model), x)
            return x.view(B, S, self.d_model)
        if self._version == 0:                          # BiMambaEncoderLayer.forward (bimamba.py:61-100)
            ln = lambda t, n, post=None: ops.layernorm_post(t, n.weight.detach(), n.bias.detach(), post=post, eps=n.eps)
            ffn = lambda t, f, resid: ops.linear_ex(ops.linear_ex(t, f[0].weight.detach(), f[0].bias.detach(), act=1),
                                                    f[3].weight.detach(), f[3].bias.detach(), resid=resid)
            for i, lyr in enumerate(self.model.layers):
                xf = ln(self._mamba(x, lyr.mamba_forward, self._Wdt[2 * i], self._Wx[2 * i], B, S, x, False), lyr.norm1)
                xf = ln(ffn(xf, lyr.ffn1, xf), lyr.norm2)
                xb = ln(self._mamba(x, lyr.mamba_backward, self._Wdt[2 * i + 1], self._Wx[2 * i + 1], B, S, x, True), lyr.norm3)
                x = ln(ffn(xf, lyr.ffn2, xb), lyr.norm4, post=xf)      # ffn2 reads x_f, as written at :94; then x_f + x_b
            return x.view(B, S, self.d_model)
        for i, lyr in enumerate(self.model.layers):
            xf = ops.layernorm_post(self._mamba(x, lyr.mamba_forward, self._Wdt[2 * i], self._Wx[2 * i], B, S, x, False),
                                    lyr.norm1.weight.detach(), lyr.norm1.bias.detach(), eps=lyr.norm1.eps)
            s = ops.layernorm_post(self._mamba(x, lyr.mamba_backward, self._Wdt[2 * i + 1], self._Wx[2 * i + 1], B, S, x, True),
                                   lyr.norm2.weight.detach(), lyr.norm2.bias.detach(), post=xf, eps=lyr.norm2.eps)      # x_f + x_b
            if isinstance(lyr.ffn, nn.Sequential):
                h = ops.linear_ex(s, lyr.ffn[0].weight.detach(), lyr.ffn[0].bias.detach(), act=1)
                f, r = ops.linear_ex(h, lyr.ffn[3].weight.detach(), lyr.ffn[3].bias.detach(), resid=s), None
            else:                                       # the mixture layer in the FFN's place; the residual enters the norm
                f, r = lyr.ffn(s.view(B, S, self.d_model)).reshape(B * S, self.d_model), s
            x = ops.layernorm_post(f, lyr.norm3.weight.detach(), lyr.norm3.bias.detach(), resid=r, eps=lyr.norm3.eps)
        return x.view(B, S, self.d_model)

    def forward(self, feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion):
        """-> (loudness_notedensity (B,S,2), instrument (B,S,40)) like video_regression.py:240-245."""
        out = self.get_feature(feature_semantic_list, feature_scene_offset, feature_motion, feature_emotion).contiguous()
        B, S, d = out.shape
        rows = out.view(B * S, d)
        ln_nd = ops.linear_ex(rows, self.regressor.weight.detach(), self.regressor.bias.detach())
        inst = ops.linear_ex(rows, self.classifier[0].weight.detach(), self.classifier[0].bias.detach(), act=2)
        return ln_nd.view(B, S, 2), inst.view(B, S, INSTRUMENT_SIZE)
