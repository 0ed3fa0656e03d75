"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np
import torch

from video2music_amd import synthetic

CFG1 = dict(n_layers=2, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300,
            total_vf_dim=synthetic.total_vf_dim(1), rpr=True)
CFG2 = dict(n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, max_sequence_chord=1024,
            total_vf_dim=synthetic.total_vf_dim(1), rpr=True)


def amt_named_shapes(n_layers, num_heads, d_model, dim_feedforward, max_sequence_chord, total_vf_dim, **_):
    """(name, shape) list of the reference VideoMusicTransformer state_dict."""
    d, ff, F = d_model, dim_feedforward, total_vf_dim
    hd = d // num_heads
    out = [("embedding.weight", (159, d)), ("embedding_root.weight", (15, d)), ("embedding_attr.weight", (16, d)),
           ("Linear_vis.weight", (d, F)), ("Linear_vis.bias", (d,)),
           ("Linear_chord.weight", (d, d + 1)), ("Linear_chord.bias", (d,)),
           ("condition_linear.weight", (d, 1)), ("condition_linear.bias", (d,))]

    def attn(p, er):
        o = [(p + "in_proj_weight", (3 * d, d)), (p + "in_proj_bias", (3 * d,))]
        if er:
            o.append((p + "Er", (max_sequence_chord, hd)))
        return o + [(p + "out_proj.weight", (d, d)), (p + "out_proj.bias", (d,))]

    def ffn(p, nn):
        o = [(p + "linear1.weight", (ff, d)), (p + "linear1.bias", (ff,)),
             (p + "linear2.weight", (d, ff)), (p + "linear2.bias", (d,))]
        for i in range(1, nn + 1):
            o += [(p + f"norm{i}.weight", (d,)), (p + f"norm{i}.bias", (d,))]
        return o

    for i in range(n_layers):
        p = f"transformer.encoder.layers.{i}."
        out += attn(p + "self_attn.", False) + ffn(p, 2)
    out += [("transformer.encoder.norm.weight", (d,)), ("transformer.encoder.norm.bias", (d,))]
    for i in range(n_layers):
        p = f"transformer.decoder.layers.{i}."
        out += attn(p + "self_attn.", True) + attn(p + "multihead_attn.", False) + ffn(p, 3)
    out += [("transformer.decoder.norm.weight", (d,)), ("transformer.decoder.norm.bias", (d,)),
            ("Wout_root.weight", (15, d)), ("Wout_root.bias", (15,)),
            ("Wout_attr.weight", (16, d)), ("Wout_attr.bias", (16,)),
            ("Wout.weight", (159, d)), ("Wout.bias", (159,))]
    return out


def synthetic_sd(cfg, seed=0, dtype=torch.float32, recipe="default"):
    sd = synthetic.synthetic_state_dict(amt_named_shapes(**cfg), seed=seed, recipe=recipe)
    return {k: torch.from_numpy(v).to(dtype) for k, v in sd.items()}


def feats_t(feats, sl=slice(None), key=None, dtype=torch.float32):
    f = {k: torch.from_numpy(np.ascontiguousarray(v[sl])).to(dtype) for k, v in feats.items()}
    if key is not None:
        f["key"] = torch.from_numpy(np.ascontiguousarray(key[sl])).to(dtype)
    return f


CFG_V2 = dict(version_name="2.2", n_layers=6, num_heads=4, d_model=128, dim_feedforward=256, max_sequence_chord=300,
              total_vf_dim=synthetic.total_vf_dim(1))


def v2_named_shapes(n_layers, num_heads, d_model, dim_feedforward, total_vf_dim, n_experts=6, **_):
    """(name, shape) list of the reference VideoMusicTransformer_V2('2.2') state_dict."""
    d, ff, F = d_model, dim_feedforward, total_vf_dim
    out = [("embedding.weight", (159, d)), ("embedding_root.weight", (15, d)), ("embedding_attr.weight", (16, d)),
           ("Linear_vis.weight", (d, F)), ("Linear_vis.bias", (d,)),
           ("Linear_chord.weight", (d, d + 1)), ("Linear_chord.bias", (d,)),
           ("condition_linear.weight", (d, 1)), ("condition_linear.bias", (d,))]

    def attn(p):
        return [(p + "in_proj_weight", (3 * d, d)), (p + "in_proj_bias", (3 * d,)), (p + "out_proj.weight", (d, d)), (p + "out_proj.bias", (d,))]

    def glu(p):
        return [(p + "linear1.weight", (ff, d)), (p + "linear1.bias", (ff,)), (p + "linear2.weight", (d, ff)), (p + "linear2.bias", (d,)),
                (p + "gate.weight", (ff, d)), (p + "gate.bias", (ff,))]

    def ffn(p, deep):
        if not deep:
            return glu(p)
        o = []
        for e in range(n_experts):
            o += glu(p + f"experts.{e}.")
        return o + [(p + "gate.weight", (n_experts, d)), (p + "gate.bias", (n_experts,))] + glu(p + "shared_expert.")

    def norms(p, n):
        o = []
        for i in range(1, n + 1):
            o += [(p + f"norm{i}.weight", (d,)), (p + f"norm{i}.bias", (d,))]
        return o

    for stack, nn_ in (("encoder", 2), ("decoder", 3)):
        for i in range(n_layers):
            p = f"transformer.{stack}.layers.{i}."
            out += attn(p + "self_attn.")
            if stack == "decoder":
                out += attn(p + "cross_attn.")
            out += ffn(p + "ff.", i >= 3) + norms(p, nn_)
        out += [(f"transformer.{stack}.norm.weight", (d,)), (f"transformer.{stack}.norm.bias", (d,))]
    return out + [("Wout.weight", (159, d)), ("Wout.bias", (159,))]


def synthetic_sd_v2(cfg, seed=0, recipe="default"):
    sd = synthetic.synthetic_state_dict(v2_named_shapes(**cfg), seed=seed, recipe=recipe)
    return {k: torch.from_numpy(v) for k, v in sd.items()}
