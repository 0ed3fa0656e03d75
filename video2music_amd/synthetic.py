"""Procedural (numpy ``RandomState``) weights and video features for the AMT hot path.

There are no trained weights or dataset features in the reference tree, so parity tests, goldens
and ``bench.py`` all use these recipes.  Nothing here depends on
torch's RNG: every tensor gets its own frozen legacy ``RandomState`` stream seeded from
``crc32(name) + seed`` so the values do not depend on iteration order.
"""
import zlib

import numpy as np

# feature widths of the video stream (video2music.py:609-610, generate.py:141-160)
SEM_DIM = 768
EMO_DIM = 6
MOTION_DIMS = {0: 1, 1: 512, 2: 768}


def total_vf_dim(motion_type: int = 1) -> int:
    """generate.py:141-160: semantic + scene offset + motion + emotion."""
    return SEM_DIM + 1 + MOTION_DIMS[motion_type] + EMO_DIM


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) + seed) % (2 ** 32))


def fill_tensor(name: str, shape, seed: int = 0) -> np.ndarray:
    """Value of one ``state_dict`` entry under the procedural recipe (fp32)."""
    rs = _rs(name, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    if leaf == "Er":                                  # rpr.py:148 uses torch.rand -> [0,1)
        t = rs.uniform(0.0, 1.0, size=shape)
    elif leaf == "A_log":                             # mamba.py:208-219: S4D-real init log(1..N), here with a little jitter
        t = np.log(np.arange(1, shape[-1] + 1, dtype=np.float64))[None, :] + rs.uniform(-0.1, 0.1, size=shape)
    elif name.endswith("dt_proj.bias"):               # mamba.py:198-204: softplus^-1 of dt log-uniform in [1e-3, 1e-1]
        dt = np.exp(rs.uniform(np.log(1e-3), np.log(1e-1), size=shape))
        t = dt + np.log(-np.expm1(-dt))
    elif leaf == "D" and len(shape) == 1:             # mamba.py:221: ones
        t = 1.0 + rs.uniform(-0.2, 0.2, size=shape)
    elif "norm" in name and leaf in ("weight", "scale"):
        t = 1.0 + rs.uniform(-0.2, 0.2, size=shape)
    elif "norm" in name and leaf == "bias":
        t = rs.uniform(-0.1, 0.1, size=shape)
    elif name.startswith("embedding") or "embedding" in name.split(".")[0]:
        t = rs.standard_normal(size=shape)
    elif leaf == "bias" or leaf.endswith("_bias"):
        t = rs.uniform(-0.05, 0.05, size=shape)
    elif len(shape) >= 2:
        fan_in = shape[-1]
        a = np.sqrt(3.0 / fan_in)
        t = rs.uniform(-a, a, size=shape)
        if name.startswith("Wout"):
            t = t * 8.0                               # keeps arg-max sequences non-degenerate
    else:
        t = rs.uniform(-0.05, 0.05, size=shape)
    return np.ascontiguousarray(t, dtype=np.float32)


# "feedback" recipe (round 2): the default recipe's greedy sequences are dominated by a constant component of the
# decoder output (4-6 distinct ids in 64 tokens, mostly the repeat-suppression pattern a,a,b), which says little about
# the root/attr feedback.  Stronger chord embeddings and weaker decoder sub-layers (the residual stream keeps the
# token; the video memory's constant pull through the cross-attention is turned down) give >= 20 distinct ids in 64
# tokens with top-1/top-2 margins >= 1e-2 at config 1 and config 2 (found by a random search over these five factors
# with the CPU oracle, then confirmed on the reference: oracle/make_goldens_cfg2.py prints both statistics).
FEEDBACK_SCALES = {"embedding": 8.0, "wout": 4.0, "self_attn": 0.5, "cross_attn": 0.1, "ffn": 0.5}


def _feedback_scale(name: str) -> float:
    leaf = name.split(".")[-1]
    if name in ("embedding_root.weight", "embedding_attr.weight"):
        return FEEDBACK_SCALES["embedding"]
    if name == "Wout.weight":
        return FEEDBACK_SCALES["wout"]
    if name.startswith("transformer.decoder.layers") and leaf.endswith("weight") and "norm" not in name:
        if ".self_attn." in name:
            return FEEDBACK_SCALES["self_attn"]
        if ".multihead_attn." in name or ".cross_attn." in name:
            return FEEDBACK_SCALES["cross_attn"]
        return FEEDBACK_SCALES["ffn"]
    return 1.0


def synthetic_state_dict(named_shapes, seed: int = 0, skip=("pe",), recipe: str = "default"):
    """``{name: ndarray}`` for an iterable of ``(name, shape)``; buffers named ``*.pe`` are skipped.
    ``recipe="feedback"`` rescales a few tensors of the default fill (see ``FEEDBACK_SCALES``)."""
    assert recipe in ("default", "feedback")
    out = {}
    for name, shape in named_shapes:
        if name.split(".")[-1] in skip:
            continue
        t = fill_tensor(name, shape, seed)
        if recipe == "feedback":
            sc = _feedback_scale(name)
            if sc != 1.0:
                t = np.ascontiguousarray(t * np.float32(sc), dtype=np.float32)
        out[name] = t
    return out


def synthetic_features(n_clips: int, seed: int = 1234, n_frames: int = 300, motion_type: int = 1):
    """Random 300-frame video features for ``n_clips`` clips.

    Returns a dict of fp32 arrays: ``semantic (B,S,768)`` ~ N(0,1); ``scene_offset (B,S)`` running
    offsets reset at random cuts (dataset/vevo_dataset.py:331-345); ``motion (B,S,512)`` ~ U[0,1)
    (``(B,S)`` for motion_type 0); ``emotion (B,S,6)`` softmax rows; ``key (B,1)`` in {0.,1.} by the
    generate.py:199-205 rule applied per clip.
    """
    rs = np.random.RandomState(seed)
    B, S = n_clips, n_frames
    sem = rs.standard_normal((B, S, SEM_DIM)).astype(np.float32)
    scene = np.zeros((B, S), dtype=np.float32)
    for b in range(B):
        cuts = rs.uniform(size=S) < 0.06
        off = 0
        for s in range(S):
            if cuts[s]:
                off = 0
            scene[b, s] = off
            off += 1
    if motion_type == 0:
        motion = rs.uniform(size=(B, S)).astype(np.float32)
    else:
        motion = rs.uniform(size=(B, S, MOTION_DIMS[motion_type])).astype(np.float32)
    z = rs.standard_normal((B, S, EMO_DIM))
    e = np.exp(z - z.max(-1, keepdims=True))
    emotion = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    key = np.stack([feature_key_from_emotion(emotion[b:b + 1]) for b in range(B)]).astype(np.float32)
    return {"semantic": sem, "scene_offset": scene, "motion": motion, "emotion": emotion, "key": key}


def feature_key_from_emotion(feature_emotion: np.ndarray) -> np.ndarray:
    """generate.py:199-205 as shipped: ``argmax(feature_emotion.mean(dim=0))`` on the ``(1,S,6)``
    tensor is an arg-max over the *flattened* ``(S,6)`` mean, so the key is minor (1.) only when
    that flat index is 1, 2 or 3."""
    flat = feature_emotion.mean(axis=0).reshape(-1)
    idx = int(np.argmax(flat))
    return np.array([1.0 if idx in (1, 2, 3) else 0.0], dtype=np.float32)
