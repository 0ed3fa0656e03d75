"""Vocabulary constants and chord tables of the AMT hot path.

Mirrors the values of the reference's ``utilities/constants.py:50-62,96`` and the
``dataset/vevo_meta/{chord,chord_inv,chord_root,chord_attr}.json`` tables that
``VideoMusicTransformer.generate`` opens (``model/video_music_transformer.py:1052-1057``).
The tables are rebuilt arithmetically here (``id = 1 + 13*(root-1) + (attr-1)``, ``N = 0``);
``oracle/make_goldens.py`` asserts they equal the reference's JSON files entry by entry.
"""

# chord vocabulary (utilities/constants.py:50-52)
CHORD_END = 157
CHORD_PAD = CHORD_END + 1
CHORD_SIZE = CHORD_PAD + 1

# chord root vocabulary (utilities/constants.py:55-57)
CHORD_ROOT_END = 13
CHORD_ROOT_PAD = CHORD_ROOT_END + 1
CHORD_ROOT_SIZE = CHORD_ROOT_PAD + 1

# chord attribute vocabulary (utilities/constants.py:60-62)
CHORD_ATTR_END = 14
CHORD_ATTR_PAD = CHORD_ATTR_END + 1
CHORD_ATTR_SIZE = CHORD_ATTR_PAD + 1

SCENE_OFFSET_MAX = 300

# the reference trains/generates with a single 159-way head (utilities/constants.py:11)
IS_SEPERATED = False
RPR = True
IS_VIDEO = True
VERSION = "AMT"

ROOT_NAMES = ["N", "C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]
ATTR_NAMES = ["N", "maj", "dim", "sus4", "min7", "min", "sus2", "aug", "dim7",
              "maj6", "hdim7", "7", "min6", "maj7"]

# dataset/vevo_meta/chord_root.json, chord_attr.json
CHORD_ROOT_DIC = {name: i for i, name in enumerate(ROOT_NAMES)}
CHORD_ATTR_DIC = {name: i for i, name in enumerate(ATTR_NAMES)}


def chord_name(chord_id: int) -> str:
    """Name of chord id 0..156 as in dataset/vevo_meta/chord_inv.json ("C", "C:min", "N")."""
    if chord_id == 0:
        return "N"
    root = (chord_id - 1) // 13 + 1
    attr = (chord_id - 1) % 13 + 1
    return ROOT_NAMES[root] if attr == 1 else ROOT_NAMES[root] + ":" + ATTR_NAMES[attr]


# dataset/vevo_meta/chord_inv.json (str id -> name) and chord.json (name -> id)
CHORD_INV_DIC = {str(i): chord_name(i) for i in range(CHORD_END)}
CHORD_DIC = {v: int(k) for k, v in CHORD_INV_DIC.items()}


def chord_to_root_attr(chord_id: int):
    """(root, attr) ids that ``generate`` feeds back for a sampled chord id.

    Follows model/video_music_transformer.py:1107-1123: a plain root (no ":") gets attr 1,
    including "N" -> (0, 1).
    """
    if chord_id == 0:
        return 0, 1
    return (chord_id - 1) // 13 + 1, (chord_id - 1) % 13 + 1


def primer_from_name(name: str):
    """(chord, root, attr) ids of a primer chord as generate.py:246-284 encodes it.

    Unlike the feedback rule above, a plain-root primer gets attr 0 (generate.py:264).
    """
    cid = CHORD_DIC[name]
    parts = name.split(":")
    if len(parts) == 1:
        return cid, CHORD_ROOT_DIC[parts[0]], 0
    return cid, CHORD_ROOT_DIC[parts[0]], CHORD_ATTR_DIC[parts[1]]


FLAT_TO_SHARP = {"Db": "C#", "Eb": "D#", "Gb": "F#", "Ab": "G#", "Bb": "A#"}          # generate.py:57-63
_SUFFIX = {"m": "min", "m6": "min6", "m7": "min7", "M6": "maj6", "M7": "maj7"}


def normalise_user_chord(name: str) -> str:
    """A chord as a user types it ("Am", "Bbm7", "F#", "CM7") -> the vocabulary's spelling ("A:min", "A#:min7", "F#",
    "C:maj7"), by the rules of generate.py:291-312: flats become sharps, the quality is split off after the root, the five
    short qualities are expanded, an empty quality leaves the bare root; anything else passes through ("Cdim" -> "C:dim")."""
    p = name
    if len(p) > 1:
        if p[1] == "b":
            p = FLAT_TO_SHARP[p[0:2]] + p[2:]
        t = 2 if p[1] == "#" else 1
        root, q = p[:t], p[t:]
        q = _SUFFIX.get(q, q)
        p = root if q == "" else root + ":" + q
    return p


def primer_from_user_chords(names):
    """[(chord, root, attr)] of a custom primer (generate.py:286-329)."""
    return [primer_from_name(normalise_user_chord(n)) for n in names]
