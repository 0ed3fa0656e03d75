"""Golden fixture for the chord -> pitch and voicing functions (tests/golden/g_chord_midi.npz) from the REFERENCE's own
`utilities/chord_to_midi.py` (`Chord(...).getMIDI`, `voice`).  TEST INFRASTRUCTURE; build container only.
`midiutil` (absent, imported at the file's top, never executed by these two functions) is stubbed for the import.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_midi.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from video2music_amd.utilities import constants as C      # noqa: E402


def main():
    m = types.ModuleType("midiutil")
    m.MIDIFile = object
    sys.modules["midiutil"] = m
    sys.path.insert(0, "/root/reference")
    from utilities.chord_to_midi import Chord, voice
    out = {}
    pitches = []
    for cid in range(C.CHORD_END):
        name = C.chord_name(cid)
        p = [] if name == "N" else Chord(name.replace(":", "")).getMIDI("c", 4)      # generate.py:452-456
        pitches.append(p)
        out[f"p{cid}"] = np.array(p, dtype=np.int64)
    rs = np.random.RandomState(0)
    for k in range(6):
        seq = rs.randint(0, C.CHORD_END, size=60)
        seq[rs.uniform(size=60) < 0.3] = seq[0]                 # runs and returns
        seq = np.where(rs.uniform(size=60) < 0.1, 0, seq)       # some "N"
        v = voice([list(pitches[int(c)]) for c in seq])
        out[f"seq{k}"] = seq.astype(np.int64)
        out[f"voiced{k}_len"] = np.array([len(x) for x in v], dtype=np.int64)
        out[f"voiced{k}"] = np.array([n for x in v for n in x], dtype=np.int64)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "g_chord_midi.npz"), **out)
    print("wrote g_chord_midi.npz", len(out))


if __name__ == "__main__":
    main()
