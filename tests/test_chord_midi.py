"""Chord ids -> MIDI: pitches and voicing against the reference's own parser / voicing function
(tests/golden/g_chord_midi.npz), the arrangement rules and the MIDI file writer.  CPU only."""
import struct

import numpy as np

from video2music_amd.render import chord_midi as M
from video2music_amd.utilities import constants as C


def test_pitches_of_every_chord_equal_the_reference_parser(golden):
    g = golden("g_chord_midi.npz")
    for cid in range(C.CHORD_END):
        assert M.chord_pitches(C.chord_name(cid)) == g[f"p{cid}"].tolist(), C.chord_name(cid)


def test_voicing_equals_the_reference_on_random_progressions(golden):
    g = golden("g_chord_midi.npz")
    for k in range(6):
        seq = g[f"seq{k}"]
        v = M.voice([M.chord_pitches(C.chord_name(int(c))) for c in seq])
        assert [len(x) for x in v] == g[f"voiced{k}_len"].tolist()
        assert [n for x in v for n in x] == g[f"voiced{k}"].tolist()


def test_arrangement_rules():
    assert M.run_offsets(["C", "C", "G", "C", "C", "C"]) == [0, 1, 0, 0, 1, 2]
    assert M.velocity_of(0) == 49 and M.velocity_of(50) == 112 and M.velocity_of(10) == int(round(0.2 ** 0.3 * 63 + 49))
    assert [M.loudness_class(v) for v in (0, 5, 6, 10, 15, 16, 20, 21, 50)] == [0, 0, 1, 1, 2, 3, 3, 4, 4]
    ids = [C.CHORD_DIC["C"], C.CHORD_DIC["C"], 0, C.CHORD_DIC["A:min7"]]
    block = M.arrange(ids, [50, 50, 50, 50], arpeggio=False, voiced=False)
    assert [n for n in block if n[1] == 0] == [(48, 0, 2, 112), (60, 0, 2, 112), (64, 0, 2, 112), (67, 0, 2, 112)]
    assert not [n for n in block if n[1] == 4]                                   # "N": silence
    arp = M.arrange(ids, [3, 3, 3, 22], arpeggio=True, voiced=False)
    assert [(p, on) for p, on, _, _ in arp if on < 2] == [(48, 0), (60, 1)]         # class 0, even position: notes 0, 1
    assert [(p, on) for p, on, _, _ in arp if 2 <= on < 4] == [(64, 2), (67, 3)]   # odd position: notes 2, 3
    last = [(p, on - 6) for p, on, _, _ in arp if on >= 6]
    am7 = M.chord_pitches("A:min7")
    assert last == [(am7[i], o) for i, o in M.FIGURES[4][0]] and len(last) == 8


def test_midi_file_round_trip(tmp_path):
    notes = M.arrange([1, 1, 66, 0, 122], [10, 30, 50, 5, 20])
    path = str(tmp_path / "x.mid")
    M.write_midi(path, notes)
    raw = open(path, "rb").read()
    assert raw[:4] == b"MThd" and struct.unpack(">IHHH", raw[4:14]) == (6, 1, 2, 960)
    # walk the chunks, decode the note track
    pos, tracks = 14, []
    while pos < len(raw):
        assert raw[pos:pos + 4] == b"MTrk"
        n = struct.unpack(">I", raw[pos + 4:pos + 8])[0]
        tracks.append(raw[pos + 8:pos + 8 + n])
        pos += 8 + n
    assert len(tracks) == 2 and tracks[0][:4] == bytes([0x00, 0xFF, 0x51, 0x03]) and int.from_bytes(tracks[0][4:7], "big") == 500000
    t, i, on, got = 0, 0, {}, []
    data = tracks[1]
    while i < len(data):
        dt = 0
        while True:
            b = data[i]; i += 1
            dt = (dt << 7) | (b & 0x7F)
            if not b & 0x80:
                break
        t += dt
        st = data[i]
        if st == 0xFF:
            assert data[i:i + 3] == bytes([0xFF, 0x2F, 0x00])
            break
        pitch, vel = data[i + 1], data[i + 2]
        i += 3
        if st == 0x90:
            on.setdefault(pitch, []).append((t, vel))
        else:
            t0, v0 = on[pitch].pop(0)
            got.append((pitch, t0 / 960, (t - t0) / 960, v0))
    assert sorted(got) == sorted((p, float(o), float(d), v) for p, o, d, v in notes)
