"""The oracle restatement against the committed golden fixtures (outputs of the reference's own
DSP objects, tests/golden/make_golden.py) and the survey's recorded md5 for config 1."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_text

STAGE_IDS = {"iq": 0, "time": 1, "pll": 2, "lock": 3, "fir": 4, "agc": 5, "sym": 6, "symt": 7, "bits": 8, "bitt": 9,
             "taps": 11}


def test_clip_md5_is_the_surveys_golden():
    # demodPOES 5sec_clip.wav, real main.c, gcc -O2
    assert hashlib.md5(golden_text("clip.c10000.txt")).hexdigest() == "d3c496d003a29eeee061c01b00ce025c"
    assert hashlib.md5(open(os.path.join(GOLDEN, "5sec_clip.wav"), "rb").read()).hexdigest() == "58ba9b19585ff59067e5aa84e08d779f"


@pytest.mark.parametrize("chunk", [10000, 1000, 260000])
def test_oracle_clip_text(orc, clip, chunk):
    rate, iq = clip
    o = orc.Oracle(orc.POES, rate, iq, chunk=chunk, keep_stages=False)
    assert o.text() == golden_text(f"clip.c{chunk}.txt")


def test_chunk_size_is_observable():
    # Q-list: a single 260000-sample chunk changes two time stamps
    assert golden_text("clip.c10000.txt") == golden_text("clip.c1000.txt")
    assert golden_text("clip.c10000.txt") != golden_text("clip.c260000.txt")


def test_oracle_clip_stage_digests(orc, clip, golden):
    rate, iq = clip
    o = orc.Oracle(orc.POES, rate, iq)
    for name, digest in golden["stages"]["clip"].items():
        got = hashlib.sha256(o.stage(STAGE_IDS[name]).tobytes()).hexdigest()
        assert got == digest, f"stage {name} differs from the reference's dump"
    assert o.lock_sample == 16528
    assert f"{o.lock_freq_hz:0.2f}" == "-3466.19"
    assert f"{o.norm_factor:f}" == "17.583342"


def test_oracle_norm_override(orc, clip):
    rate, iq = clip
    o = orc.Oracle(orc.POES, rate, iq, norm_override=12.5, keep_stages=False)
    assert o.text() == golden_text("clip.n12.txt")


@pytest.mark.parametrize("fs", [18750, 32000, 50000, 100000, 250000])
def test_oracle_synthetic_poes(orc, pdt, golden, fs):
    p = golden["params"]
    iq = pdt.synth_capture(0, fs, p["poes_seconds"], seed=p["poes_seed"])
    assert hashlib.sha256(iq.tobytes()).hexdigest() == golden["synth"][f"poes_{fs}"], "generator is not bit-reproducible"
    o = orc.Oracle(orc.POES, fs, iq)
    assert o.text() == golden_text(f"poes_{fs}.txt")
    for name, digest in golden["stages"][f"poes_{fs}"].items():
        assert hashlib.sha256(o.stage(STAGE_IDS[name]).tobytes()).hexdigest() == digest, name
    taps = np.fromfile(os.path.join(GOLDEN, f"taps_poes_{fs}.f32"), dtype=np.float32)
    assert o.stage(orc.ST_TAPS).tobytes() == taps.tobytes()
    assert len(taps) == 26 * o.interp


def test_oracle_synthetic_argos(orc, pdt, golden):
    p = golden["params"]
    iq = pdt.synth_capture(1, 32000, p["argos_seconds"], seed=p["argos_seed"])
    assert hashlib.sha256(iq.tobytes()).hexdigest() == golden["synth"]["argos_32000"]
    o = orc.Oracle(orc.ARGOS, 32000, iq)
    assert o.text() == golden_text("argos_32000.txt")
    for name, digest in golden["stages"]["argos_32000"].items():
        assert hashlib.sha256(o.stage(STAGE_IDS[name]).tobytes()).hexdigest() == digest, name
    o2 = orc.Oracle(orc.ARGOS, 32000, iq, chunk=1000, keep_stages=False)
    assert o2.text() == golden_text("argos_32000.c1000.txt")


def test_round_trip_poes(orc, pdt):
    """encode -> demodulate: every complete decoded frame equals a transmitted frame, in order."""
    fs = 50000
    iq = pdt.synth_capture(0, fs, 8.0, seed=77)
    par = pdt.synth_params(0, fs, 1000.0, 77)
    o = orc.Oracle(orc.POES, fs, iq, keep_stages=False)
    frames = [f for f in o.frames() if f.complete]
    assert len(frames) >= 70
    # the first decoded frame tells which transmitted frame we locked onto
    sent = {bytes(pdt.synth_poes_frame(par, k)): k for k in range(0, 90)}
    idx = [sent.get(bytes(f.bytes[:104])) for f in frames]
    assert all(i is not None for i in idx), "decoded frame not among the transmitted ones"
    assert idx == list(range(idx[0], idx[0] + len(idx)))


def test_empty_and_tiny_inputs(orc):
    for n in (0, 1, 7, 9999, 10000, 10001, 20000):
        iq = np.zeros((n, 2), dtype=np.int16)
        o = orc.Oracle(orc.POES, 50000, iq, keep_stages=False)
        assert o.text() == b""
        assert o.totals()[0] == n


def test_cross_check_against_the_older_builds_sample_output():
    """SURVEY 4 cross-check (not a golden): POESTIPdemod/minorFrame.txt is the sample output an OLDER build of the
    reference produced from the bundled 5sec_clip.wav (six-decimal times, older time base).  Today's reference --
    and so the oracle and the GPU path -- gives the same frames from the second one on, byte for byte, except
    byte 22 of two frames (01->51, 03->5B).  Any wider drift in the restatement would show up here."""
    old = [l.split() for l in open(os.path.join(GOLDEN, "old_build_minorFrame.txt")).read().splitlines() if l.strip()]
    new = [l.split() for l in golden_text("clip.c10000.txt").decode().splitlines() if l.strip()]
    assert len(old) == 47 and len(new) == 48
    diffs = []
    for k, o in enumerate(old):
        n = new[k + 1]
        assert len(o) == len(n)
        assert abs(float(o[0]) - float(n[0])) < 0.01           # older time base, same 0.1 s frame slot
        diffs += [(k, i - 1, a, b) for i, (a, b) in enumerate(zip(o, n)) if i > 0 and a != b]
    assert diffs == [(16, 22, "01", "51"), (36, 22, "03", "5B")]


@pytest.mark.parametrize("piece", [4160, 1000, 77, 19137])
def test_bytesync_on_the_references_own_harness_bits(orc, piece):
    """Known answer for the synchroniser alone: the literal bit string of the reference's commented-out harness
    (POESTIPdemod/ByteSync.c:6-14), frames produced by the reference's ByteSync object (make_bytesync_vector.py)."""
    bits = np.frombuffer(golden_text("bytesync_harness_bits.txt").strip(), dtype=np.uint8)
    text, frames = orc.bytesync(orc.POES, bits, piece)
    assert text == golden_text("bytesync_harness_frames.txt")
    assert len(frames) == 23 and all(f[4] for f in frames)
