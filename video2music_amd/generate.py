"""``generate.py`` entry point for the AMT hot path (reference ``generate.py:86-392``).

Reproduces the part of the reference script that feeds ``model.generate``: flag names/defaults
(``parse_generate_args``), the ``total_vf_dim`` rule (:141-160), the batch-1 feature shapes
(:181-189), the key rule (:199-207), the default primer ("C" / "A:min", :246-284) and the
``model.generate`` keyword names (:368-392), the chord ``.lab`` it writes (:440-444), and with ``--regression`` / ``--midi`` the
regression head and the voiced-arpeggio MIDI (:393-607).  Audio and video rendering (FluidSynth, moviepy: :608-709) are out of scope.  Inputs come from
``--synthetic`` or from MuVi-Sync feature files (``-dataset_dir`` + ``--test_ids``, ``dataset/vevo_features.py``); under torchrun the clips are sharded over ranks and the ids all-gathered.

    python -m video2music_amd.generate --synthetic --n_clips 4 -target_seq_length_chord 64 -beam 0
"""
import json
import os
import sys

import numpy as np
import torch

from . import dist as vdist
from . import synthetic
from .model.video_music_transformer import (VideoMusicTransformer, VideoMusicTransformer_V1, VideoMusicTransformer_V2,
                                            VideoMusicTransformer_V3)
from .utilities import constants as C
from .utilities.argument_generate_funcs import parse_generate_args
from .utilities.device import get_device

max_conseq_N = 0          # generate.py:65-66
max_conseq_chord = 2


def total_vf_dim_of(args, sem_dim=synthetic.SEM_DIM):
    """generate.py:141-160."""
    total = sem_dim if args.is_video else 0
    total += 1
    total += {0: 1, 1: 512, 2: 768}[args.motion_type]
    total += 6 if args.emo_model.startswith("6c") else 5
    return total


def default_primer(feature_key):
    """generate.py:246-269: "C" for a major key, "A:min" for a minor one; plain roots get attr 0."""
    return C.primer_from_name("C" if int(feature_key) == 0 else "A:min")


def regression_levels(args, f, device):
    """generate.py:394-409 / video2music.py:849-865: the regression head on (semantic, emotion); per frame the integer
    note density (round, clip [0,40]) and loudness level (int(100 x), clip [0,50]) the MIDI renderer consumes."""
    from .model.video_regression import VideoRegression
    reg = VideoRegression(n_layers=args.n_layers_reg, d_model=args.d_model_reg, d_hidden=args.dim_feedforward_reg,
                          max_sequence_video=args.max_sequence_video, total_vf_dim=f["semantic"].shape[-1] + f["emotion"].shape[-1],
                          regModel=args.regModel).eval()
    if args.synthetic or args.synthetic_weights:
        shapes = [(k, tuple(v.shape)) for k, v in reg.state_dict().items()]
        reg.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=1).items()})
    else:
        reg.load_state_dict(torch.load(args.modelReg_weights, map_location="cpu"))
    reg = reg.to(device)
    ln_nd, _ = reg(f["semantic"], f["scene_offset"], f["motion"], f["emotion"])
    y = ln_nd.cpu().numpy()
    nd = np.clip(np.round(y[..., 0]).astype(int), 0, 40)
    lv = np.clip((y[..., 1] * 100).astype(int), 0, 50)
    return list(zip(nd, lv))


def main(argv=None):
    args = parse_generate_args(argv)[0]
    if args.music_gen_version in ("None", "none", ""):
        args.music_gen_version = None
    if args.music_gen_version is not None and not args.music_gen_version.startswith(("1.", "2.", "3.")):
        raise SystemExit("music_gen_version must be None (base AMT) or start with '1.', '2.' or '3.' (generate.py:209-238)")
    if args.force_cpu:
        raise SystemExit("--force_cpu: video2music_amd has no CPU path (the CPU oracle lives in oracle/ for tests only)")
    rank, world, local = vdist.init()
    device = get_device()
    if device.type != "cuda":
        raise SystemExit("no GPU visible: video2music_amd runs on MI355X only")
    if args.synthetic:
        feats = synthetic.synthetic_features(args.n_clips, seed=args.seed, n_frames=args.max_sequence_video,
                                             motion_type=args.motion_type)
        names = [f"clip{i:03d}" for i in range(args.n_clips)]
    else:
        # feature files of the MuVi-Sync layout (dataset/vevo_dataset.py:241-554); key bit by the script's own rule (:199-207)
        from .dataset import vevo_features as VF
        if not args.test_ids:
            raise SystemExit("--test_ids (or --synthetic) is required")
        names = (VF.read_split(args.dataset_dir, args.test_ids[6:], "v1") if args.test_ids.startswith("split:")
                 else [t.strip() for t in args.test_ids.split(",") if t.strip()])
        feats = VF.load_clips(args.dataset_dir, names, vis_models="2d/clip_l14p", emo_model=args.emo_model, motion_type=args.motion_type,
                              max_seq_video=args.max_sequence_video, max_seq_chord=args.max_sequence_chord)
        feats["key"] = np.array([[VF.key_from_emotion(e)] for e in feats["emotion"]], dtype=np.float32)
        own_chords = np.stack([feats["chord"], feats["chord_root"], feats["chord_attr"]], axis=-1)     # (n, Tc, 3)
        feats = {k: feats[k] for k in ("semantic", "key", "scene_offset", "motion", "emotion")}
        args.n_clips = len(names)
    common = dict(n_layers=args.n_layers, num_heads=args.num_heads, d_model=args.d_model, dim_feedforward=args.dim_feedforward,
                  max_sequence_midi=args.max_sequence_midi, max_sequence_video=args.max_sequence_video,
                  max_sequence_chord=args.max_sequence_chord,
                  total_vf_dim=total_vf_dim_of(args, sem_dim=feats["semantic"].shape[-1]))     # generate.py:141-143: widths of the loaded features
    if args.music_gen_version is None:                 # generate.py:209-216
        model = VideoMusicTransformer(rpr=args.rpr, **common)
    elif args.music_gen_version.startswith("1."):      # generate.py:221-226
        model = VideoMusicTransformer_V1(version_name=args.music_gen_version, rms_norm=args.rms_norm, **common)
    elif args.music_gen_version.startswith("3."):      # generate.py:233-238
        model = VideoMusicTransformer_V3(version_name=args.music_gen_version, rms_norm=args.rms_norm, **common)
    else:                                              # generate.py:227-232
        model = VideoMusicTransformer_V2(version_name=args.music_gen_version, rms_norm=args.rms_norm, **common)
    if args.synthetic or args.synthetic_weights:
        shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        sd = {k: torch.from_numpy(v) for k, v in synthetic.synthetic_state_dict(shapes, seed=0).items()}
        model.load_state_dict(sd, strict=False)
    else:
        model.load_state_dict(torch.load(args.model_weights, map_location="cpu"))
    model = model.to(device).eval()
    lo, hi = vdist.shard_bounds(args.n_clips, rank, world)
    f = {k: torch.from_numpy(v[lo:hi]).to(device) for k, v in feats.items()}
    # primer (chord, root, attr) rows per clip, (n_local, P, 3): generate.py:246-344
    if args.primer:                                    # custom chords, typed the user's way ("C Am Dm G")
        rows = C.primer_from_user_chords(args.primer.replace(",", " ").split())
        prim = torch.tensor([rows] * (hi - lo), device=device).reshape(hi - lo, len(rows), 3)
    elif args.primer_from_dataset:                     # the clip's own first chords (:367-379)
        if args.synthetic:
            raise SystemExit("--primer_from_dataset needs chord files: use -dataset_dir / --test_ids")
        prim = torch.from_numpy(own_chords[lo:hi, :args.num_prime_chord]).to(device)
    else:                                              # "C" for a major key, "A:min" for a minor one (:246-284)
        prim = torch.tensor([[default_primer(k)] for k in feats["key"][lo:hi, 0]], device=device).reshape(hi - lo, 1, 3)
    with torch.set_grad_enabled(False):
        if args.beam > 1:
            assert False, "No Beam sampling method implemented yet..."     # generate.py:347-349
        print("RAND DIST" if args.beam == 0 else "BEAM: 1")
        if args.music_gen_version is None:
            toks = model.generate_batch(f["semantic"], f["key"], f["scene_offset"], f["motion"], f["emotion"],
                                        prim[:, :, 0].contiguous(), prim[:, :, 1].contiguous(), prim[:, :, 2].contiguous(),
                                        target_seq_length=args.target_seq_length_chord, beam=args.beam,
                                        max_conseq_N=max_conseq_N, max_conseq_chord=max_conseq_chord, sampler=args.sampler)
        else:
            # the V1 / V2 / V3 classes generate one clip per call in the reference; here the clips of this rank advance in
            # lockstep through one captured step graph, --v2_batch clips at a time (each projection then reads its weights
            # once per step for all of them); per clip the ids equal the one-clip generate
            n_local = hi - lo
            rows = []
            for c0 in range(0, n_local, max(1, args.v2_batch)):
                sl = slice(c0, min(n_local, c0 + max(1, args.v2_batch)))
                rows.append(model.generate_batch(f["semantic"][sl], f["key"][sl], f["scene_offset"][sl], f["motion"][sl], f["emotion"][sl],
                                                 prim[sl, :, 0], prim[sl, :, 1], prim[sl, :, 2],
                                                 target_seq_length=args.target_seq_length_chord, beam=args.beam, max_conseq_N=max_conseq_N,
                                                 max_conseq_chord=max_conseq_chord, sampler=args.sampler))
            toks = torch.cat(rows) if rows else torch.empty(0, args.target_seq_length_chord, dtype=torch.long, device=device)
        toks = vdist.all_gather_sequences(toks, args.n_clips)
        reg_rows = None
        if args.regression:
            reg_rows = regression_levels(args, f, device)
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        out = toks.cpu().numpy()
        from .dataset.vevo_features import write_lab
        for name, row in zip(names, out):
            write_lab(os.path.join(args.output_dir, f"{name}_chords.lab"), row)      # generate.py:440-444
        if args.midi:
            from .render import chord_midi
            for i, (name, row) in enumerate(zip(names, out)):
                levels = None
                if reg_rows is not None and lo <= i < hi:
                    lv = reg_rows[i - lo][1]
                    levels = [int(lv[min(j, len(lv) - 1)]) for j in range(len(row))]
                chord_midi.write_midi(os.path.join(args.output_dir, f"{name}_chords.mid"), chord_midi.arrange(row, levels))
        if reg_rows is not None:            # this rank's clips only (the head is cheap; no gather)
            for name, (nd, lv) in zip(names[lo:hi], reg_rows):
                with open(os.path.join(args.output_dir, f"{name}_loudness_density.csv"), "w") as fh:
                    fh.write("frame,note_density,loudness_level\n")
                    fh.write("".join(f"{i},{int(a)},{int(b)}\n" for i, (a, b) in enumerate(zip(nd, lv))))
        print(json.dumps({"clips": int(out.shape[0]), "length": int(out.shape[1]), "first": out[0, :16].tolist()}))
    return toks


if __name__ == "__main__":
    main(sys.argv[1:])
