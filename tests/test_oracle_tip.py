"""Downstream frame validation: the oracle's restatement of the reference's MATLAB
checkParity.m / daytimeDecode.m, pinned on the reference's own bundled capture."""
import os

import numpy as np
import pytest

from oracle import binding as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_frames(name="clip.c10000.txt"):
    """(time, 104 bytes, complete) of every line of a golden minorFrames file (reference output)."""
    out = []
    for line in open(os.path.join(HERE, "golden", name)).read().split("\n"):
        p = line.split()
        if len(p) < 2:
            continue
        b = bytes(int(x, 16) for x in p[1:])
        out.append((float(p[0].rstrip("i")), b, len(b) == 104))
    return out


def test_real_capture_satisfies_every_parity_equation():
    """5sec_clip.wav is a real NOAA-15 pass: all 5 x 47 even-parity equations of its complete frames hold
    under the word/bit mapping of checkParity.m:20-86 -- which pins that mapping."""
    frames = golden_frames()
    sm, rec = orc.tip_check(frames)
    assert sm["frames_checked"] == 47 and sm["good_frames"] == 47
    assert sm["good_chunks"] == 5 * 47 and sm["bad_chunks"] == 0
    assert sm["spacecraft"] == 8                                   # NOAA-15 (daytimeDecode.m:82)
    ids = rec["minor_id"][rec["checked"] == 1]
    assert np.all(np.diff(ids.astype(int)) % 320 == 1)             # consecutive 9-bit minor-frame counters
    assert rec["checked"][-1] == 0                                 # the partial last frame is not a matrix row
    assert sm["time_frames"] == int(np.sum((ids == 0)))


def test_major_frame_time_fields():
    frames = golden_frames()
    sm, rec = orc.tip_check(frames)
    t = rec[rec["has_time"] == 1]
    if len(t):                                                     # the 5-second clip holds at most one major-frame start
        assert sm["day"] == int(t["day"][0])
    # a synthetic major-frame start: day 249 -> bytes 8/9, 56 242 685 ms -> bytes 9..12
    b = bytearray(frames[0][1])
    b[4] &= 0xFE; b[5] = 0
    ms = 56242685
    b[8] = 249 >> 1
    b[9] = ((249 & 1) << 7) | ((ms >> 24) & 7)
    b[10], b[11], b[12] = (ms >> 16) & 255, (ms >> 8) & 255, ms & 255
    sm2, rec2 = orc.tip_check([(12.5, bytes(b), True)])
    assert rec2["has_time"][0] == 1 and rec2["day_ms"][0] == ms
    assert rec2["day"][0] == ((249 >> 1) << 1) + 1                 # the reference's "| 128" always yields an odd day
    assert sm2["t0_ms"] == ms - 12500
    b[9] |= 7; b[10] = 255                                         # >= 86 400 000 ms: rejected (daytimeDecode.m:24)
    _, rec3 = orc.tip_check([(12.5, bytes(b), True)])
    assert rec3["day_ms"][0] == -1


@pytest.mark.parametrize("byte,group", [(2, 0), (18, 0), (19, 1), (35, 1), (36, 2), (52, 2), (53, 3), (69, 3), (70, 4), (86, 4)])
def test_single_bit_errors_are_attributed_to_their_group(byte, group):
    frames = golden_frames()[:3]
    t, b, c = frames[1]
    bad = bytearray(b)
    bad[byte] ^= 0x10
    sm, rec = orc.tip_check([frames[0], (t, bytes(bad), c), frames[2]])
    assert list(rec["parity"]) == [0, 1 << group, 0]
    assert sm["good_frames"] == 2 and sm["bad_chunks"] == 1


def test_unchecked_words_and_parity_bits():
    t, b, c = golden_frames()[0]
    for byte in (0, 1, 87, 100, 102):                              # words outside the five groups
        bad = bytearray(b); bad[byte] ^= 1
        assert orc.tip_check([(t, bytes(bad), c)])[1]["parity"][0] == 0
    for g, shift in enumerate((5, 4, 3, 2, 1)):                    # flipping a parity bit itself
        bad = bytearray(b); bad[103] ^= 1 << shift
        assert orc.tip_check([(t, bytes(bad), c)])[1]["parity"][0] == 1 << g
    for shift in (7, 6, 0):                                        # CPU flags and the unchecked sixth group
        bad = bytearray(b); bad[103] ^= 1 << shift
        assert orc.tip_check([(t, bytes(bad), c)])[1]["parity"][0] == 0


def test_hand_derived_known_answers():
    """tests/tip_kat.py: nine frames whose expected records were derived by hand from checkParity.m / daytimeDecode.m --
    the pin of the oracle's restatement that does not depend on the oracle itself (there is no MATLAB here)."""
    import tip_kat
    sm, rec = orc.tip_check([(t, b, True) for _, t, b, _ in tip_kat.VECTORS])
    tip_kat.check(rec, sm)
