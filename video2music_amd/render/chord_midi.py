"""Chord ids -> MIDI (reference generate.py:410-607 + utilities/chord_to_midi.py).

Host-side post-processing of the generated chord sequence, restated from the reference's behaviour:

  * `chord_pitches`: the pitches the reference's chord-name parser (`Chord(name).getMIDI("c", 4)`,
    chord_to_midi.py:196-316) returns for the 157 chords of the vocabulary — bass an octave below the root, then
    root / third / fifth / extension.  The parser's quirks are part of the behaviour and are kept: "sus4"/"sus2" repeat
    the suspended note, "dim7" carries a minor (not diminished) seventh, "hdim7" comes out as a dominant seventh.
    Pinned for every chord by `tests/golden/g_chord_midi.npz` (produced by the reference's parser itself).
  * `voice`: the voice-leading pass (chord_to_midi.py:132-194), pinned by the same fixture on random progressions.
  * `arrange`: velocities from the regression head's loudness level (generate.py:409-418), the five arpeggio
    figures chosen by loudness class and the chord's position inside a run of equal chords (:420-433, :462-603), or
    block chords (:604-607).  These figures are restated from the script's text (it cannot be run here: it needs the
    dataset, `midiutil`, FluidSynth and moviepy), so unlike the two functions above they are not pinned by an execution of
    the reference.
  * `write_midi`: a minimal Standard MIDI File writer (format 1, 960 ticks per quarter, tempo 120) in place of `midiutil`.
"""
import math
import struct

from ..utilities import constants as C

PITCH_CLASS = {"C": 12, "C#": 13, "D": 14, "D#": 15, "E": 16, "F": 17, "F#": 18, "G": 19, "G#": 20, "A": 9, "A#": 10, "B": 11}
# semitones above the root after the (bass, root) pair, per chord quality of the vocabulary ("" = plain major triad)
INTERVALS = {"": (4, 7), "maj": (4, 7), "dim": (3, 6), "sus4": (5, 7, 5), "min7": (3, 7, 10), "min": (3, 7), "sus2": (2, 7, 2),
             "aug": (4, 8), "dim7": (3, 6, 10), "maj6": (4, 7, 9), "hdim7": (4, 7, 10), "7": (4, 7, 10), "min6": (3, 7, 9),
             "maj7": (4, 7, 11)}
DURATION, TEMPO, TICKS = 2, 120, 960                      # generate.py:40-41; midiutil's default resolution
MIN_VEL, MAX_VEL, MAX_LOUD, EXPONENT = 49, 112, 50, 0.3   # generate.py:47-50,411

# arpeggio figures (generate.py:462-603): per loudness class, for the even / odd position inside a run of equal chords,
# (index into the voiced chord, onset in beats)
FIGURES = {
    0: (((0, 0), (1, 1)), ((2, 0), (3, 1))),
    1: (((0, 0), (1, .5), (2, 1)), ((3, 0), (1, .5), (2, 1))),
    2: (((0, 0), (1, .5), (2, 1), (3, 1.5)), ((2, 0), (1, .5), (2, 1), (3, 1.5))),
    3: (((0, 0), (1, .25), (2, .5), (1, .75), (3, 1), (2, 1.5)), ((1, 0), (0, .25), (1, .5), (2, .75), (3, 1), (2, 1.5))),
    4: (((0, 0), (1, .25), (2, .5), (1, .75), (3, 1), (2, 1.25), (1, 1.5), (2, 1.75)),
        ((1, 0), (0, .25), (1, .5), (2, .75), (3, 1), (2, 1.25), (1, 1.5), (2, 1.75))),
}


def chord_pitches(name, octave=4):
    """MIDI pitches of a vocabulary chord ("N" -> [])."""
    if name == "N":
        return []
    root_name, _, quality = name.partition(":")
    root = PITCH_CLASS[root_name] + 12 * octave
    return [root - 12, root] + [root + iv for iv in INTERVALS[quality]]


def voice(chords):
    """Voice leading over a list of pitch lists (empty = no chord): the first chord is kept, every later chord keeps
    its bass within a seventh of the previous bass (octave shift when that is closer) and moves each upper note to the
    octave nearest the previous chord's closest pitch class neighbour, unless that leaves the register (center + 8)."""
    out, prev, center = [], None, 0
    for cur in chords:
        if not cur:
            out.append([])
            continue
        if prev is None:
            out.append(list(cur))
            prev, center = list(cur), cur[1] + 3
            continue
        v = []
        for i, note in enumerate(cur):
            if i == 0:
                p = prev[0]
                best = note
                if abs(note - p) > 7:           # more than a fifth away: the octave on the other side is always closer
                    best = note + 12 if note < p else note - 12
                v.append(best)
                continue
            neighbour, allowance = None, -1
            while neighbour is None:
                allowance += 1
                for p in prev[1:]:
                    if abs(note - p) % 12 in (allowance, 12 - allowance):
                        neighbour = p
                        break
            if note <= neighbour:
                best = note + ((neighbour - note + 6) // 12) * 12
            else:
                best = note + math.ceil((neighbour - note - 6) / 12) * 12
            if not (abs(best - center) <= 8 or allowance > 2):
                best = note
            v.append(best)
        v.sort()
        out.append(v)
        prev = v
    return out


def run_offsets(names):
    """Position of each chord inside its run of equal chords (generate.py:74-84)."""
    out, cur, off = [], None, 0
    for n in names:
        if n != cur:
            cur, off = n, 0
        out.append(off)
        off += 1
    return out


def velocity_of(level):
    """generate.py:411-418."""
    return int(round((level / MAX_LOUD) ** EXPONENT * (MAX_VEL - MIN_VEL) + MIN_VEL))


def loudness_class(level):
    """generate.py:420-433 (the script classes the *loudness* level, not the note density)."""
    return 0 if level <= 5 else 1 if level <= 10 else 2 if level <= 15 else 3 if level <= 20 else 4


def arrange(chord_ids, loudness_levels=None, arpeggio=True, voiced=True):
    """[(pitch, onset beats, duration beats, velocity)] for a chord-id sequence; loudness_levels: per chord integer
    level 0..50 from the regression head (None: level 25 everywhere)."""
    names = [C.chord_name(int(c)) for c in chord_ids]
    chords = [chord_pitches(n) for n in names]
    if voiced:
        chords = voice(chords)
    offs = run_offsets(names)
    notes = []
    for i, ch in enumerate(chords):
        level = 25 if loudness_levels is None else int(loudness_levels[i])
        vel = velocity_of(level)
        if not arpeggio:
            notes += [(p, i * DURATION, DURATION, vel) for p in ch]
        elif len(ch) in (4, 5):
            for idx, on in FIGURES[loudness_class(level)][offs[i] % 2]:
                notes.append((ch[idx], i * DURATION + on, DURATION, vel))
    return notes


def _vlq(n):
    b = [n & 0x7F]
    n >>= 7
    while n:
        b.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(b))


def write_midi(path, notes, tempo=TEMPO):
    """Standard MIDI File, format 1: a tempo track and one note track (channel 0)."""
    ev = []
    for pitch, on, dur, vel in notes:
        t0, t1 = int(round(on * TICKS)), int(round((on + dur) * TICKS))
        ev.append((t0, 1, bytes([0x90, pitch & 0x7F, vel & 0x7F])))
        ev.append((t1, 0, bytes([0x80, pitch & 0x7F, 0])))
    ev.sort(key=lambda e: (e[0], e[1]))                   # note-offs before note-ons at the same tick
    body, last = b"", 0
    for t, _, msg in ev:
        body += _vlq(t - last) + msg
        last = t
    body += b"\x00\xff\x2f\x00"
    tempo_track = b"\x00\xff\x51\x03" + struct.pack(">I", 60000000 // tempo)[1:] + b"\x00\xff\x2f\x00"
    with open(path, "wb") as fh:
        fh.write(b"MThd" + struct.pack(">IHHH", 6, 1, 2, TICKS))
        fh.write(b"MTrk" + struct.pack(">I", len(tempo_track)) + tempo_track)
        fh.write(b"MTrk" + struct.pack(">I", len(body)) + body)
