"""MuVi-Sync ("vevo") feature files -> the tensors `VideoMusicTransformer.forward/generate` take.

Host-side reader for the on-disk layout the reference's `VevoDataset.createSample` consumes
(`dataset/vevo_dataset.py:58-236` for the directory tree, `:241-554` for the parsing rules).  Only the
inputs of the chord model are read (chords, key, semantic, scene offset, motion, emotion); the
regression targets (loudness, note density, instrument) belong to SURVEY.md §8 row f2.

    <root>/vevo_chord/lab_v2_norm/origin/<id>.lab     "key C major" then "<t> <chord>" per second
    <root>/vevo_semantic/origin/<p1>/<p2>/<id>.npy    (n, 768) float            (vis_models "p1/p2")
    <root>/vevo_scene_offset/origin/<id>.lab          "<t> <scene id>"
    <root>/vevo_motion/origin/<id>.lab                "<t> <motion>"             (motion_type 0)
    <root>/vevo_motion/option1|option2/<id>.npy       (n, 512|768) float         (motion_type 1|2)
    <root>/vevo_emotion/<emo_model>/origin/<id>.lab   "time ..." header, then "<t> p1 .. p6|p5"

Quirks kept on purpose (they decide what the model sees):
  * every per-second text stream stops at the first line whose time is >= max_seq_**chord**
    (`:334,355,383,401,430`), not max_seq_video;
  * the `.npy` motion features are padded/cut to max_seq_**chord** rows (`:361-374`);
  * scene ids are stored +1 so that 0 is the pad value (`:337`);
  * a plain-root chord ("G") gets attribute id 1 = "maj", "N" gets (0, 0) (`:270-286`);
  * a chord file that ends before max_seq_chord leaves END (not PAD) right after its last chord: the reference sets
    END in the shifted target *view*, which aliases the chord tensor (`:318-330`);
  * key = 0 iff the normalised chord file's key line contains "major" (`:291-294`).
"""
import os

import numpy as np

from ..utilities import constants as C

SEMANTIC_PAD = SCENE_OFFSET_PAD = MOTION_PAD = EMOTION_PAD = 0.0      # utilities/constants.py:65-75
MOTION_DIM = {1: 512, 2: 768}


def _rows(path):
    """Yields the whitespace-split fields of every non-empty line."""
    with open(path, encoding="utf-8") as fh:
        for line in fh:
            fields = line.strip().split(" ")
            if fields and fields[0] != "":
                yield fields


def read_series(path, length, limit, width=None, pad=0.0, transform=float, header=None):
    """A "<time> v1 [.. vw]" text stream as a dense array of `length` rows (pad-filled).

    Reading stops at the first time >= `limit`, like the reference loops.  `header`: first field of lines to skip.
    """
    out = np.full((length,) if width is None else (length, width), pad, dtype=np.float64)
    for fields in _rows(path):
        if header is not None and fields[0] == header:
            continue
        t = int(fields[0])
        if t >= limit:
            break
        if width is None:
            out[t] = transform(fields[1])
        else:
            vals = [transform(v) for v in fields[1:]]
            if len(vals) != width:
                raise ValueError(f"{path}: line for t={t} has {len(vals)} values, expected {width}")
            out[t] = vals
    return out.astype(np.float32)


def read_chords(path, max_seq_chord):
    """Chord `.lab` -> (chord ids, root ids, attr ids) of length max_seq_chord (PAD-filled), key flag (0 major / 1
    minor) and the index of the last chord read (the reference puts END there in the targets)."""
    chord = np.full(max_seq_chord, C.CHORD_PAD, dtype=np.int64)
    root = np.full(max_seq_chord, C.CHORD_ROOT_PAD, dtype=np.int64)
    attr = np.full(max_seq_chord, C.CHORD_ATTR_PAD, dtype=np.int64)
    key, last = "", -1
    for fields in _rows(path):
        if fields[0] == "key":
            key = " ".join(fields[1:3])
            continue
        t = int(fields[0])
        if t >= max_seq_chord:
            break
        name = fields[1]
        chord[t] = C.CHORD_DIC[name]
        parts = name.split(":")
        root[t] = C.CHORD_ROOT_DIC[parts[0]]
        attr[t] = C.CHORD_ATTR_DIC[parts[1]] if len(parts) == 2 else (C.CHORD_ATTR_DIC["N"] if parts[0] == "N" else 1)
        last = t
    else:
        # the file ended before the limit: the reference writes END into its shifted *target view*
        # (`tgt[time] = CHORD_END`, :327-330), which aliases the chord tensor one position later
        if 0 <= last < max_seq_chord - 1:
            chord[last + 1], root[last + 1], attr[last + 1] = C.CHORD_END, C.CHORD_ROOT_END, C.CHORD_ATTR_END
    return chord, root, attr, (0.0 if "major" in key else 1.0), last


def _pad_rows(a, rows, pad):
    out = np.full((rows, a.shape[1]), pad, dtype=np.float32)
    n = min(rows, a.shape[0])
    out[:n] = a[:n]
    return out


def clip_paths(dataset_root, fid, vis_models="2d/clip_l14p", emo_model="6c_l14p", motion_type=1):
    p1, p2 = vis_models.split(" ")[0].split("/")
    motion_dir = {0: "origin", 1: "option1", 2: "option2"}[motion_type]
    return {
        "chord": os.path.join(dataset_root, "vevo_chord", "lab_v2_norm", "origin", fid + ".lab"),
        "semantic": os.path.join(dataset_root, "vevo_semantic", "origin", p1, p2, fid + ".npy"),
        "scene_offset": os.path.join(dataset_root, "vevo_scene_offset", "origin", fid + ".lab"),
        "motion": os.path.join(dataset_root, "vevo_motion", motion_dir, fid + (".lab" if motion_type == 0 else ".npy")),
        "emotion": os.path.join(dataset_root, "vevo_emotion", emo_model, "origin", fid + ".lab"),
    }


def load_clip(dataset_root, fid, vis_models="2d/clip_l14p", emo_model="6c_l14p", motion_type=1,
              max_seq_video=300, max_seq_chord=300):
    """One clip as numpy arrays: semantic (S,768), scene_offset (S,), motion (Tc,512|768) or (S,), emotion (S,6|5),
    key (1,), chord/chord_root/chord_attr (Tc,) int64 (PAD where the file has no chord; all PAD without a file)."""
    if len(vis_models.split(" ")) != 1:
        raise NotImplementedError("one semantic model per run, as in the reference's defaults (vis_models='2d/clip_l14p')")
    p = clip_paths(dataset_root, fid, vis_models, emo_model, motion_type)
    S, Tc = max_seq_video, max_seq_chord
    out = {"semantic": _pad_rows(np.load(p["semantic"]).astype(np.float32), S, SEMANTIC_PAD)}
    out["scene_offset"] = read_series(p["scene_offset"], S, Tc, pad=SCENE_OFFSET_PAD, transform=lambda v: int(v) + 1)
    if motion_type == 0:
        out["motion"] = read_series(p["motion"], S, Tc, pad=MOTION_PAD)
    else:
        m = np.load(p["motion"]).astype(np.float32)
        if m.shape[1] != MOTION_DIM[motion_type]:
            raise ValueError(f"{p['motion']}: {m.shape[1]} motion features, motion_type {motion_type} has {MOTION_DIM[motion_type]}")
        out["motion"] = _pad_rows(m, Tc, 0.0)
    out["emotion"] = read_series(p["emotion"], S, Tc, width=6 if emo_model.startswith("6c") else 5, pad=EMOTION_PAD, header="time")
    if os.path.exists(p["chord"]):
        chord, root, attr, key, last = read_chords(p["chord"], Tc)
    else:       # a clip without a chord file: no primer chords, key from the emotion stream (generate.py's rule)
        chord = np.full(Tc, C.CHORD_PAD, dtype=np.int64)
        root = np.full(Tc, C.CHORD_ROOT_PAD, dtype=np.int64)
        attr = np.full(Tc, C.CHORD_ATTR_PAD, dtype=np.int64)
        key = None
    out.update(chord=chord, chord_root=root, chord_attr=attr)
    out["key"] = np.array([key_from_emotion(out["emotion"]) if key is None else key], dtype=np.float32)
    return out


def key_from_emotion(emotion):
    """The key flag `generate.py:199-207` derives from the (S, 6) emotion probabilities: 1 (minor) iff the arg-max over
    the *flattened* array is index 1, 2 or 3 — the script calls `torch.argmax` without a dim on the (S, 6) mean over its
    batch axis, so the rule only ever fires when the clip's largest probability is frame 0's fearful / tense / sad
    entry; everything else is major (0).  Reproduced as is: it decides which primer and key bit the model gets."""
    return 1.0 if int(np.argmax(np.asarray(emotion, dtype=np.float32).reshape(-1))) in (1, 2, 3) else 0.0


def load_clips(dataset_root, ids, **kw):
    """Stacks `load_clip` over ids in the layout of `synthetic.synthetic_features`: semantic (B,S,768), key (B,1),
    scene_offset (B,S), motion (B,S,·) / (B,S), emotion (B,S,6), plus chord / chord_root / chord_attr (B,Tc)."""
    clips = [load_clip(dataset_root, fid, **kw) for fid in ids]
    return {k: np.stack([c[k] for c in clips]) for k in clips[0]}


def read_split(dataset_root, split="test", split_ver="v1"):
    """Clip ids of `vevo_meta/split/<ver>/<split>.txt`."""
    with open(os.path.join(dataset_root, "vevo_meta", "split", split_ver, split + ".txt")) as fh:
        return [line.strip() for line in fh if line.strip()]


def write_lab(path, chord_ids):
    """The chord `.lab` the reference writes next to the MIDI (generate.py:440-444): "key ?" then "<i> <chord>"."""
    with open(path, "w", encoding="utf-8") as fh:
        fh.write("key ?\n")
        for i, cid in enumerate(chord_ids):
            fh.write(f"{i} {C.chord_name(int(cid))}\n")
