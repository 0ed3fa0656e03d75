"""Randomised parity sweep of the HIP path against the CPU oracle (test infrastructure -- it lives under tests/ because it drives the oracle --, GPU box only).

Draws model shapes, batch sizes, clip lengths, primer lengths and target lengths at random (seeded), and for every draw compares
  * the teacher-forced forward logits with oracle.forward (<= 1e-3, BASELINE.json's tolerance),
  * the feedback-greedy ids (beam=0, argmax = oracle G2) and the top-1 ids (beam=1 = oracle G1) of every clip with oracle.generate;
    an id mismatch is only accepted as a near-tie when the oracle's own top-1 / top-2 margin at the first differing position is below 1e-4,
  * the decode-path logits with the forward's on the generated sequence (<= 1e-3, as for the forward).
Usage: python tests/fuzz_parity.py [n_cases] [seed] [v2]      -> one JSON line per case (third argument v2: the V2 '2.2' family against the oracle; families: V1 / V2 / V3 cached decode against their own re-forward; modules: MultiheadGQA / MoELayer / SharedMoELayer against the oracle; reg: VideoRegression against oracle/reg_oracle.py), a summary at the end, exit 1 on any failure.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import amt_oracle as O
from video2music_amd import synthetic
from video2music_amd.model.video_music_transformer import VideoMusicTransformer
from video2music_amd.utilities import constants as C
from tests.helpers import synthetic_sd, feats_t


LONG = False         # "long": 100 ... 300 tokens on small models (one or two clips)
ODD = False          # "odd": model widths that are multiples of 32 but not of 64 (96, 160, 224, ...), feed-forward widths likewise
BIG = False          # "big": model widths 576 ... 1024 (the library's upper range, incl. the plain chain beyond d + dff = 1536), up to 40 clips


def draw(rs):
    hd = int(rs.choice([16, 32, 64, 128]))
    H = int(rs.choice([1, 2, 3, 4, 8]))
    while H * hd > 512:
        H = max(1, H // 2)
    d = H * hd
    if BIG:
        hd = int(rs.choice([64, 128]))
        H = int(rs.choice([5, 6, 8, 12, 16]))
        while H * hd > 1024:
            H -= 1
        d = H * hd
    ff = int(rs.choice([d, 2 * d, 3 * d, 64, 256 + 64 * int(rs.randint(0, 8))]))
    ff = max(64, min(ff, 1536) // 64 * 64)
    if ODD:
        hd = int(rs.choice([32, 32, 16]))
        H = int(rs.choice([3, 3, 5, 7, 9, 11])) if hd == 32 else int(rs.choice([6, 10, 14]))
        d = H * hd
        ff = int(rs.choice([32, 96, 160, 224, 288, 64, 128, 2 * d, 3 * d]))
    cfg = dict(n_layers=int(rs.randint(1, 4)), num_heads=H if not ODD else int(H), d_model=d, dim_feedforward=ff,
               max_sequence_chord=int(rs.choice([24, 40, 64, 100, 300])), total_vf_dim=synthetic.total_vf_dim(int(rs.randint(0, 2))),
               rpr=bool(rs.rand() < 0.8))
    B = int(rs.choice([1, 1, 2, 3, 5, 7]))
    if BIG:
        cfg["n_layers"] = int(rs.randint(1, 3))
        B = int(rs.choice([1, 2, 33, 40]))
    S = int(rs.choice([300, 300, 1, 2, 17, 120, 299]))
    T = int(rs.randint(2, min(cfg["max_sequence_chord"], 36) + 1))
    if LONG:
        cfg.update(max_sequence_chord=300, n_layers=int(rs.randint(1, 3)))
        if cfg["d_model"] > 128:
            cfg.update(d_model=128, num_heads=int(rs.choice([1, 2, 4, 8])), dim_feedforward=int(rs.choice([64, 128, 256])))
        B, T = int(rs.choice([1, 2])), int(rs.randint(100, 301))
    P = int(rs.randint(1, min(4, T) + 1))
    recipe = str(rs.choice(["default", "feedback"]))
    return cfg, B, S, T, P, recipe


def run_case(i, rs):
    cfg, B, S, T, P, recipe = draw(rs)
    motion_type = 1 if cfg["total_vf_dim"] == synthetic.total_vf_dim(1) else 0
    info = dict(case=i, cfg={k: v for k, v in cfg.items() if k != "total_vf_dim"}, motion_type=motion_type, B=B, S=S, T=T, P=P, recipe=recipe)
    m = VideoMusicTransformer(**cfg).eval()
    sd = synthetic_sd(cfg, seed=100 + i, recipe=recipe)
    if not cfg["rpr"]:                                # the oracle follows the state_dict: no Er rows = torch's plain decoder layer
        sd = {k: v for k, v in sd.items() if not k.endswith(".Er")}
    m.load_state_dict(sd, strict=False)
    m = m.cuda()
    feats = synthetic.synthetic_features(B, seed=500 + i, motion_type=motion_type)
    fc = feats_t(feats)
    fc = {k: (v[:, :S].contiguous() if v.dim() > 1 and v.shape[1] == 300 else v) for k, v in fc.items()}
    f = {k: v.cuda() for k, v in fc.items()}
    vocab = sorted(C.CHORD_DIC, key=C.CHORD_DIC.get)
    vocab = [n for n in vocab if 0 < C.CHORD_DIC[n] < C.CHORD_END]
    # primers: the same chords for every clip, or (30 % of the draws) different ones per clip (generate_batch takes (B, P) primers)
    per_clip = bool(rs.rand() < 0.3)
    prim_all = torch.tensor([[C.primer_from_name(str(rs.choice(vocab))) for _ in range(P)] for _ in range(B if per_clip else 1)])   # (B or 1, P, 3)
    mcn, mcc = int(rs.choice([0, 0, 1])), int(rs.choice([2, 2, 1, 3]))
    info.update(per_clip_primers=per_clip, max_conseq_N=mcn, max_conseq_chord=mcc)
    prim = prim_all[0]
    pb = (lambda j: prim_all[:, :, j]) if per_clip else (lambda j: prim_all[0, :, j])
    fails = []
    # forward
    L = int(rs.randint(1, T + 1))
    root = torch.from_numpy(rs.randint(0, 13, size=(B, L)))
    attr = torch.from_numpy(rs.randint(0, 14, size=(B, L)))
    with torch.no_grad():
        ref = O.forward(sd, cfg["num_heads"], root, attr, fc["semantic"], fc["key"], fc["scene_offset"], fc["motion"], fc["emotion"])
        got = m(root, root.cuda(), attr.cuda(), f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"]).cpu()
    info["fwd_err"] = float((got - ref).abs().max())
    if not info["fwd_err"] < 1e-3:
        fails.append("forward")
    # generate, both branches
    near = 0
    for beam in (0, 1):
        kw = dict(sampler="argmax") if beam == 0 else {}
        with torch.no_grad():
            out = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pb(0), pb(1), pb(2),
                                   target_seq_length=T, beam=beam, max_conseq_N=mcn, max_conseq_chord=mcc, **kw).cpu()
        for b in (range(B) if B <= 8 else (0, 31, 32, B - 1)):
            one = {k: v[b:b + 1] for k, v in fc.items()}
            prim = prim_all[b if per_clip else 0]
            ref_ids = O.generate(sd, cfg["num_heads"], one["semantic"], one["key"], one["scene_offset"], one["motion"], one["emotion"],
                                 prim[:, 0], prim[:, 1], prim[:, 2], target_seq_length=T, beam=beam, max_conseq_N=mcn, max_conseq_chord=mcc)
            if torch.equal(out[b:b + 1], ref_ids):
                continue
            j = int((out[b] != ref_ids[0]).nonzero()[0])
            # the oracle's margin at the first differing decision (position j is decided from the prefix of length j)
            ids = ref_ids[0, :j]
            if beam == 0:
                rr = torch.tensor([[O.root_attr_of(int(t))[0] if k >= P else int(prim[k, 1]) for k, t in enumerate(ids)]])
                aa = torch.tensor([[O.root_attr_of(int(t))[1] if k >= P else int(prim[k, 2]) for k, t in enumerate(ids)]])
            else:
                rr = torch.tensor([[int(prim[k, 1]) if k < P else 14 for k in range(j)]])
                aa = torch.tensor([[int(prim[k, 2]) if k < P else 15 for k in range(j)]])
            lg = O.forward(sd, cfg["num_heads"], rr, aa, one["semantic"], one["key"], one["scene_offset"], one["motion"], one["emotion"])[0, -1, :157]
            top = torch.topk(torch.softmax(lg, -1), 3).values
            margin = float(top[0] - top[1])
            if margin < 1e-4:
                near += 1
            else:
                fails.append(f"ids beam={beam} clip={b} pos={j} got={int(out[b, j])} want={int(ref_ids[0, j])} margin={margin:.2e}")
    info["near_ties"] = near
    # decode logits vs forward on the generated sequence
    prim = prim_all[0]
    with torch.no_grad():
        toks, lg = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], prim[:, 0], prim[:, 1], prim[:, 2],
                                    target_seq_length=T, beam=0, sampler="argmax", return_logits=True)
        tk = toks.cpu()
        roots = torch.tensor([[int(prim[k, 1]) if k < P else O.root_attr_of(int(t))[0] for k, t in enumerate(row)] for row in tk])
        attrs = torch.tensor([[int(prim[k, 2]) if k < P else O.root_attr_of(int(t))[1] for k, t in enumerate(row)] for row in tk])
        fwd = m(toks, roots.cuda(), attrs.cuda(), f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"])
    if T - 1 > P - 1:
        dec = lg[P - 1:T - 1]                        # logits of the positions the decode path decided
        info["decode_vs_forward"] = float((fwd[:, P - 1:T - 1].permute(1, 0, 2) - dec).abs().max())
        if not info["decode_vs_forward"] < 1e-3:      # BASELINE.json's logit tolerance (observed: <= 3.4e-4 on logits of magnitude ~100)
            fails.append("decode_vs_forward")
    # the Categorical draw done on the device: every id must be the inverse CDF of the reference's decision distribution at its uniform
    if T > P and mcc >= 1:
        from tests.test_model_gpu import _check_draws
        u = torch.from_numpy(rs.rand(T, B).astype(np.float32))
        with torch.no_grad():
            tk, lgc = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], pb(0), pb(1), pb(2),
                                       target_seq_length=T, beam=0, sampler="categorical", uniforms=u, return_logits=True,
                                       max_conseq_N=mcn, max_conseq_chord=mcc)
        try:
            _check_draws(tk, lgc, u, P, mcn, mcc)
        except AssertionError as e:
            fails.append(f"categorical draw {str(e)[:120]}")
    info["fails"] = fails
    del m
    return info


def run_case_v2(i, rs):
    """The same for VideoMusicTransformer_V2 '2.2' (the reference's default family): forward logits against
    oracle.forward_v2, the lockstep generate_batch (decision on the device) against oracle.generate(forward_fn=forward_v2) clip by clip."""
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V2
    hd = int(rs.choice([32, 64, 128]))
    H = int(rs.choice([1, 2, 4, 8]))
    while H * hd > 512:
        H //= 2
    d = H * hd
    ff = int(rs.choice([d, 2 * d, 192, 320]))
    cfg = dict(version_name="2.2", n_layers=int(rs.choice([3, 4, 6])), num_heads=H, d_model=d, dim_feedforward=ff,
               max_sequence_chord=int(rs.choice([40, 300])), total_vf_dim=synthetic.total_vf_dim(1))
    B, S = int(rs.choice([1, 2, 3, 5])), int(rs.choice([300, 300, 120, 17]))
    T = int(rs.randint(2, 25))
    P = int(rs.randint(1, min(3, T) + 1))
    temperature = float(rs.choice([1.0, 1.0, 0.7]))
    recipe = str(rs.choice(["default", "feedback"]))
    info = dict(case=i, family="V2", cfg={k: v for k, v in cfg.items() if k != "total_vf_dim"}, B=B, S=S, T=T, P=P, temperature=temperature, recipe=recipe)
    m = VideoMusicTransformer_V2(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=300 + i, recipe=recipe).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    m = m.cuda()
    fc = feats_t(synthetic.synthetic_features(B, seed=700 + i))
    fc = {k: (v[:, :S].contiguous() if v.dim() > 1 and v.shape[1] == 300 else v) for k, v in fc.items()}
    f = {k: v.cuda() for k, v in fc.items()}
    prim = torch.tensor([C.primer_from_name(n) for n in ["C", "A:min", "D:min"][:P]])
    fails = []
    L = int(rs.randint(1, T + 1))
    root = torch.from_numpy(rs.randint(0, 13, size=(B, L)))
    attr = torch.from_numpy(rs.randint(0, 14, size=(B, L)))
    with torch.no_grad():
        ref = O.forward_v2(sd, H, root, attr, fc["semantic"], fc["key"], fc["scene_offset"], fc["motion"], fc["emotion"])
        got = m(root, root, attr, f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"]).cpu()
    info["fwd_err"] = float((got - ref).abs().max())
    if not info["fwd_err"] < 1e-3:
        fails.append("forward")
    near = 0
    for beam in (0, 1):
        kw = dict(sampler="argmax") if beam == 0 else {}
        with torch.no_grad():
            out = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], prim[:, 0], prim[:, 1], prim[:, 2],
                                   target_seq_length=T, beam=beam, temperature=temperature, **kw).cpu()
        for b in range(B):
            one = {k: v[b:b + 1] for k, v in fc.items()}
            margins = []
            ref_ids = O.generate(sd, H, one["semantic"], one["key"], one["scene_offset"], one["motion"], one["emotion"],
                                 prim[:, 0], prim[:, 1], prim[:, 2], target_seq_length=T, beam=beam, forward_fn=O.forward_v2,
                                 temperature=temperature, margins=margins if beam == 0 else None)
            if torch.equal(out[b:b + 1], ref_ids):
                continue
            j = int((out[b] != ref_ids[0]).nonzero()[0])
            margin = margins[j - P] if beam == 0 and 0 <= j - P < len(margins) else None
            if margin is not None and margin < 1e-4:
                near += 1
            else:
                fails.append(f"ids beam={beam} clip={b} pos={j} got={int(out[b, j])} want={int(ref_ids[0, j])} margin={margin}")
    info["near_ties"] = near
    info["fails"] = fails
    del m
    return info


def run_case_options(i, rs):
    """The base model's rarely used options at random shapes against the oracle: chord_embed / scene_embed (state_dict built like
    oracle/make_goldens_base_embed.py builds the reference's), forward(mask=False), and generate with beam > 1 / beam_chance < 1 under a seeded
    python `random` (model/video_music_transformer.py:1074-1084; Categorical replaced by arg-max)."""
    import random
    from video2music_amd.utilities.constants import SCENE_OFFSET_MAX
    cfg, B, S, T, P, recipe = draw(rs)
    ce, se = bool(rs.rand() < 0.4), bool(rs.rand() < 0.4)
    cfg["total_vf_dim"] = synthetic.total_vf_dim(1) - int(se)
    info = dict(case=i, cfg={k: v for k, v in cfg.items() if k != "total_vf_dim"}, chord_embed=ce, scene_embed=se, B=B, S=S, T=T, P=P, recipe=recipe)
    sd = dict(synthetic_sd(cfg, seed=200 + i, recipe=recipe))
    if not cfg["rpr"]:
        sd = {k: v for k, v in sd.items() if not k.endswith(".Er")}
    if se:
        sd["scene_embedding.weight"] = torch.from_numpy(synthetic.fill_tensor("scene_embedding.weight", (SCENE_OFFSET_MAX, cfg["d_model"]), 200 + i))
    if ce:
        sd["chord_embedding_model.weight"] = torch.from_numpy(synthetic.fill_tensor("chord_embedding_model.weight", (159, cfg["d_model"]), 200 + i))
    m = VideoMusicTransformer(**cfg, chord_embed=ce, scene_embed=se).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(".pe") for k in missing), (missing, unexpected)
    m = m.cuda()
    fc = feats_t(synthetic.synthetic_features(B, seed=600 + i))
    fc = {k: (v[:, :S].contiguous() if v.dim() > 1 and v.shape[1] == 300 else v) for k, v in fc.items()}
    f = {k: v.cuda() for k, v in fc.items()}
    fails = []
    H = cfg["num_heads"]
    L = int(rs.randint(1, T + 1))
    ids = torch.from_numpy(rs.randint(0, 157, size=(B, L)))
    root = torch.from_numpy(rs.randint(0, 13, size=(B, L)))
    attr = torch.from_numpy(rs.randint(0, 14, size=(B, L)))
    for mask in (True, False):
        with torch.no_grad():
            ref = O.forward(sd, H, ids if ce else root, attr, fc["semantic"], fc["key"], fc["scene_offset"], fc["motion"], fc["emotion"], mask=mask)
            got = m(ids.cuda(), root.cuda(), attr.cuda(), f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], mask=mask).cpu()
        err = float((got - ref).abs().max())
        info["fwd_err" if mask else "fwd_nomask_err"] = err
        if not err < 1e-3:
            fails.append("forward mask=%s" % mask)
    prim = torch.tensor([C.primer_from_name(n) for n in ["C", "A:min", "D:min", "G"][:P]])
    for beam, chance in ((0, 1.0), (1, 1.0), (int(rs.randint(2, 5)), float(rs.choice([1.0, 0.5]))), (1, 0.4)):
        if ce and beam > 1:
            continue                                   # the reference itself fails there (cross-attention shape at the second step)
        one = {k: v[:1] for k, v in fc.items()}
        seed = int(rs.randint(0, 10000))
        ref_ids = O.generate(sd, H, one["semantic"], one["key"], one["scene_offset"], one["motion"], one["emotion"], prim[:, 0], prim[:, 1], prim[:, 2],
                             target_seq_length=T, beam=beam, beam_chance=chance, rng=random.Random(seed))
        random.seed(seed)
        with torch.no_grad():
            out = m.generate(feature_semantic_list=f["semantic"][:1], feature_key=f["key"][0], feature_scene_offset=f["scene_offset"][:1],
                             feature_motion=f["motion"][:1], feature_emotion=f["emotion"][:1], primer=prim[:, 0], primer_root=prim[:, 1],
                             primer_attr=prim[:, 2], target_seq_length=T, beam=beam, beam_chance=chance, sampler="argmax").cpu()
        if out.shape != ref_ids.shape or not torch.equal(out, ref_ids):
            fails.append(f"generate beam={beam} chance={chance}: shape {tuple(out.shape)} vs {tuple(ref_ids.shape)}"
                         + ("" if out.shape != ref_ids.shape else f" first diff at {[int(v) for v in (out != ref_ids).nonzero()[0]]}"))
    info["fails"] = fails
    del m
    return info


def run_case_families(i, rs):
    """V1 ('1.0' ... '1.3.4', rms_norm on / off), V2 ('2.0', '2.1', '2.2') and V3 ('3.0' - '3.2') at random shapes.  The oracle holds
    only V2 '2.2'; the other families are pinned by goldens of the reference classes at their default shapes (tests/golden/g_v1, g_v2_variants,
    g_v3), so here the cached / lockstep decode (skinny-GEMM step kernels) is compared with the per-step re-forward of the prefix on the
    operator kernels (use_cache=False: the reference's loop, the path those goldens pin) clip by clip."""
    from video2music_amd.model.video_music_transformer import VideoMusicTransformer_V1, VideoMusicTransformer_V2, VideoMusicTransformer_V3
    version = str(rs.choice(["1.0", "1.1", "1.2", "1.3", "1.3.3", "1.3.4", "2.0", "2.1", "2.2", "3.0", "3.1", "3.2"]))
    cls = {"1": VideoMusicTransformer_V1, "2": VideoMusicTransformer_V2, "3": VideoMusicTransformer_V3}[version[0]]
    hd = int(rs.choice([16, 32, 64, 128]))
    H = int(rs.choice([1, 2, 4, 8]))
    while H * hd > 512:
        H //= 2
    while H * hd < 64:
        H *= 2
    d = H * hd
    ff = int(rs.choice([d, 2 * d, 64, 192, 320]))
    cfg = dict(version_name=version, n_layers=int(rs.choice([3, 4, 6])), num_heads=H, d_model=d, dim_feedforward=ff,
               max_sequence_chord=int(rs.choice([40, 300])), total_vf_dim=synthetic.total_vf_dim(1))
    if version[0] == "1" and rs.rand() < 0.3:
        cfg["rms_norm"] = True
    if rs.rand() < 0.3:
        cfg["chord_embed"] = True                     # chord ids through a frozen table; the ids themselves feed back
    if rs.rand() < 0.3:
        cfg["scene_embed"] = True                     # scene offsets through an embedding instead of a feature column
        cfg["total_vf_dim"] = synthetic.total_vf_dim(1) - 1
    B, S = int(rs.choice([1, 2, 3, 5])), int(rs.choice([300, 300, 120, 17]))
    msv = int(rs.choice([300, 300, 64, 400]))          # max_sequence_video: video positional table / RoPE cache length
    cfg["max_sequence_video"] = msv
    S = min(S, msv) if msv < 300 else (int(rs.choice([S, 350])) if msv > 300 else S)
    T = int(rs.randint(2, 25))
    P = int(rs.randint(1, min(3, T) + 1))
    info = dict(case=i, family="V" + version[0], cfg={k: v for k, v in cfg.items() if k != "total_vf_dim"}, B=B, S=S, T=T, P=P)
    m = cls(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=900 + i, recipe="feedback").items()}
    m.load_state_dict(sd, strict=False)
    m = m.cuda()
    fc = feats_t(synthetic.synthetic_features(B, seed=1100 + i, n_frames=S))
    f = {k: v.cuda() for k, v in fc.items()}
    prim = torch.tensor([C.primer_from_name(n) for n in ["C", "A:min", "D:min"][:P]])
    fails = []
    with torch.no_grad():
        for beam in (0, 1):
            kw = dict(sampler="argmax") if beam == 0 else {}
            out = m.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"], prim[:, 0], prim[:, 1], prim[:, 2],
                                   target_seq_length=T, beam=beam, **kw).cpu()
            for b in range(B):
                one = {k: v[b:b + 1] for k, v in f.items()}
                gk = dict(feature_semantic_list=one["semantic"], feature_key=one["key"][0], feature_scene_offset=one["scene_offset"],
                          feature_motion=one["motion"], feature_emotion=one["emotion"], primer=prim[:, 0], primer_root=prim[:, 1],
                          primer_attr=prim[:, 2], target_seq_length=T, beam=beam, **kw)
                ref = m.generate(use_cache=False, **gk).cpu()
                if not torch.equal(out[b:b + 1], ref):
                    j = int((out[b] != ref[0]).nonzero()[0])
                    fails.append(f"lockstep vs re-forward beam={beam} clip={b} pos={j} got={int(out[b, j])} want={int(ref[0, j])}")
                if b == 0 and version[0] != "3":
                    cached = m.generate(**gk).cpu()
                    if not torch.equal(cached, ref):
                        fails.append(f"cached one-clip vs re-forward beam={beam}")
    info["fails"] = fails
    del m
    return info


def run_case_modules(i, rs):
    """The stand-alone modules of configs 4 / 5 at random shapes against the oracle: MultiheadGQA (head_dim 16 ... 128, every kv-head
    grouping, self- and cross-shaped inputs, causal or not, the (L, B) memory reinterpretation for B > 1) and MoELayer / SharedMoELayer
    (2 ... 16 experts, top-1 ... top-4, expert widths that are not multiples of 64)."""
    from video2music_amd.model.grouped_query_attention import MultiheadGQA
    from video2music_amd.model.moe import GLUExpert, MoELayer, SharedMoELayer
    fails = []
    if rs.rand() < 0.5:
        hd = int(rs.choice([16, 32, 64, 128]))
        qh = int(rs.choice([1, 2, 4, 8, 16]))
        while qh * hd > 1024:
            qh //= 2
        kvh = int(rs.choice([h for h in (1, 2, 4, 8, 16) if qh % h == 0]))
        E = qh * hd
        L, B = int(rs.choice([1, 7, 33, 128, 300, 700])), int(rs.choice([1, 2, 3]))
        causal = bool(rs.rand() < 0.5)
        S = L if causal or rs.rand() < 0.5 else int(rs.choice([5, 64, 300]))
        info = dict(case=i, module="MultiheadGQA", E=E, query_heads=qh, kv_heads=kvh, L=L, S=S, B=B, causal=causal)
        m = MultiheadGQA(E, qh, kvh).eval()
        shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=40 + i).items()}
        m.load_state_dict(sd)
        m = m.cuda()
        q = torch.from_numpy(rs.standard_normal((L, B, E)).astype(np.float32))
        kv = q if S == L else torch.from_numpy(rs.standard_normal((S, B, E)).astype(np.float32))
        ref = O.gqa_forward(q, kv, kv, sd, qh, kvh, is_causal=causal)
        with torch.no_grad():
            y, _ = m(q.cuda(), kv.cuda(), kv.cuda(), is_causal=causal)
        info["err"] = float((y.cpu() - ref).abs().max())
        info["scale"] = float(ref.abs().max())
    else:
        d = int(rs.choice([64, 128, 192, 256, 512]))
        dff = int(rs.choice([d, 2 * d, 96, 200, 320]))
        ne = int(rs.choice([2, 4, 6, 8, 16]))
        k = int(rs.choice([1, 2, 2, 3, 4]))
        k = min(k, ne)
        shared = bool(rs.rand() < 0.5)
        L, B = int(rs.choice([1, 5, 64, 300, 1024])), int(rs.choice([1, 2, 4]))
        info = dict(case=i, module="SharedMoELayer" if shared else "MoELayer", d=d, dff=dff, n_experts=ne, k=k, L=L, B=B)
        m = (SharedMoELayer(GLUExpert(d, dff), d, n_experts=ne, n_experts_per_token=k) if shared
             else MoELayer(GLUExpert(d, dff), d, ne, k)).eval()
        shapes = [(kk, tuple(v.shape)) for kk, v in m.state_dict().items()]
        sd = {kk: torch.from_numpy(v) for kk, v in synthetic.synthetic_state_dict(shapes, seed=60 + i).items()}
        m.load_state_dict(sd)
        m = m.cuda()
        x = torch.from_numpy(rs.standard_normal((L, B, d)).astype(np.float32))
        ref = O.moe_forward(x, sd, ne, k=k, shared=shared)
        with torch.no_grad():
            y = m(x.cuda())
        info["err"] = float((y.cpu() - ref).abs().max())
        info["scale"] = float(ref.abs().max())
    if not info["err"] < 1e-4 * max(1.0, info["scale"]):
        fails.append("output")
    info["fails"] = fails
    return info


def run_case_reg(i, rs):
    """VideoRegression (row f2) at random sizes against oracle/reg_oracle.py: the Mamba / BiMamba heads in both gate versions and the
    recurrent heads, 1-4 layers, clips of 1 ... 300 frames."""
    from oracle import reg_oracle as R
    from video2music_amd.model.video_regression import VideoRegression
    reg = str(rs.choice(["bimamba+", "bimamba", "mamba+", "mamba", "lstm", "bilstm", "gru", "bigru", "cnngru", "cnnbigru"]))
    rnn = "lstm" in reg or "gru" in reg
    cfg = dict(n_layers=int(rs.randint(1, 5)), d_model=int(rs.choice([16, 32, 64, 128])),
               d_hidden=int(rs.choice([16, 32, 64, 128, 256])), total_vf_dim=774, regModel=reg)
    B, S = int(rs.choice([1, 2, 3])), int(rs.choice([1, 2, 31, 32, 33, 120, 299, 300]))
    info = dict(case=i, module="VideoRegression", cfg=cfg, B=B, S=S)
    m = VideoRegression(**cfg).eval()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=80 + i).items()}
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    f = synthetic.synthetic_features(B, seed=90 + i)
    sem, emo = torch.from_numpy(f["semantic"][:, :S].copy()), torch.from_numpy(f["emotion"][:, :S].copy())
    with torch.no_grad():
        ln_nd, inst = m(sem.cuda(), None, None, emo.cuda())
    ref_ln, ref_inst = R.forward(sd, sem, emo, reg_model=reg)
    info["err"] = float(max((ln_nd.cpu() - ref_ln).abs().max(), (inst.cpu() - ref_inst).abs().max()))
    info["scale"] = float(ref_ln.abs().max())
    info["fails"] = [] if info["err"] < 1e-4 * max(1.0, info["scale"]) else ["output"]
    return info


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3 and sys.argv[3] == "v2":
        globals()["run_case"] = run_case_v2
    if len(sys.argv) > 3 and sys.argv[3] == "long":
        globals()["LONG"] = True
    if len(sys.argv) > 3 and sys.argv[3] == "odd":
        globals()["ODD"] = True
    if len(sys.argv) > 3 and sys.argv[3] == "options":
        globals()["run_case"] = run_case_options
    if len(sys.argv) > 3 and sys.argv[3] == "big":
        globals()["BIG"] = True
    if len(sys.argv) > 3 and sys.argv[3] == "reg":
        globals()["run_case"] = run_case_reg
    if len(sys.argv) > 3 and sys.argv[3] == "modules":
        globals()["run_case"] = run_case_modules
    if len(sys.argv) > 3 and sys.argv[3] == "families":
        globals()["run_case"] = run_case_families
    rs = np.random.RandomState(seed)
    bad = 0
    t0 = time.time()
    for i in range(n):
        state = rs.get_state()
        try:
            info = run_case(i, rs)
        except Exception as e:                        # a refused shape is reported with its message, not hidden
            msg = f"{type(e).__name__}: {str(e)[:200]}"
            if run_case.__name__ == "run_case":       # the base sweep: say which shape it was
                rs2 = np.random.RandomState()
                rs2.set_state(state)
                c, B, S, T, P, _ = draw(rs2)
                msg += f" | draw: {dict((k, v) for k, v in c.items() if k != 'total_vf_dim')} B={B} S={S} T={T} P={P}"
            refused = "amt_create failed" in msg and "d_model must be" in msg      # the library's documented shape caps
            info = dict(case=i, refused=msg) if refused else dict(case=i, fails=["exception " + msg])
            info.setdefault("fails", [])
        print(json.dumps(info), flush=True)
        bad += bool(info["fails"])
    print(json.dumps({"cases": n, "failed": bad, "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
