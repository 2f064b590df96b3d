"""Stage-level parity of the sync-word search / frame extraction kernels (pdt_stage_bytesync)
with the byte synchroniser of the oracle on crafted bit strings -- the kind of harness the
reference itself carries, commented out, at POESTIPdemod/ByteSync.c:6-14."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SYNC_POES = "1110110111100010000"
SYNC_ARGOS = "0001011110000"


def to_bits(s: str) -> np.ndarray:
    return np.frombuffer(s.encode(), dtype=np.uint8).copy()


def rand_bits(rng, n):
    return "".join(rng.choice(["0", "1"], size=n))


def check(pdt, orc, mode, bitstr, piece=4160):
    bits = to_bits(bitstr)
    text, frames = orc.bytesync(orc.ARGOS if mode == pdt.MODE_ARGOS else orc.POES, bits, piece)
    with pdt.Demodulator(mode, 50000) as d:
        d.bytesync(bits)
        got = d.frames_array()
        assert d.text() == text
        assert len(got) == len(frames)
        for g, f in zip(got, frames):
            assert (g["time"], g["bit_index"], g["inverted"], g["nbytes"], g["complete"]) == f[:5]
            assert bytes(g["bytes"]) == f[5]
        return d.stats().sync_overflow, len(frames)


def inv(s):
    return s.translate(str.maketrans("01", "10"))


def test_contiguous_frames_and_truncation(pdt, orc):
    rng = np.random.default_rng(1)
    body = lambda: rand_bits(rng, 813)
    s = rand_bits(rng, 200) + "".join(SYNC_POES + body() for _ in range(40)) + SYNC_POES + rand_bits(rng, 300)
    ov, n = check(pdt, orc, pdt.MODE_POES, s)
    assert n >= 41 and ov == 0
    for cut in (0, 1, 18, 19, 20, 24, 25, 32, 832, 833, 850):
        check(pdt, orc, pdt.MODE_POES, s[:200 + cut])


def test_inverse_frames_and_mixed(pdt, orc):
    rng = np.random.default_rng(2)
    parts = [rand_bits(rng, 77)]
    for k in range(30):
        fr = SYNC_POES + rand_bits(rng, 813)
        parts.append(inv(fr) if k % 3 == 1 else fr)
        if k % 5 == 4:
            parts.append(rand_bits(rng, int(rng.integers(1, 40))))
    check(pdt, orc, pdt.MODE_POES, "".join(parts))


def test_sync_inside_a_frame_is_ignored_and_reopens_exactly_at_the_end(pdt, orc):
    rng = np.random.default_rng(3)
    payload = list(rand_bits(rng, 813))
    payload[100:119] = SYNC_POES                       # a sync word inside the frame: ignored (Q10)
    payload[813 - 19:813] = SYNC_POES                  # one that completes on the frame's last bit: opens the next frame
    s = rand_bits(rng, 50) + SYNC_POES + "".join(payload) + rand_bits(rng, 900)
    check(pdt, orc, pdt.MODE_POES, s)


def test_early_bits_against_the_zero_history(pdt, orc):
    # the history ring starts as all '0' (ByteSync.c:39): "...10000" can complete a sync within the first 19 bits only
    # for the inverse word 0001001000011101111 whose leading zeros match the initial history
    s = "1001000011101111" + "0" * 900
    check(pdt, orc, pdt.MODE_POES, s)
    check(pdt, orc, pdt.MODE_POES, "")
    check(pdt, orc, pdt.MODE_POES, "1")


def test_dense_hits_overflow_path(pdt, orc):
    """ARGOS frames are 13 + 56 bits: a 4096-bit tile holds up to 59 of them -> more than 31 hits,
    the generic (sorted append) path must take over and agree."""
    rng = np.random.default_rng(4)
    s = "".join(SYNC_ARGOS + rand_bits(rng, 56) for _ in range(400)) + SYNC_ARGOS + rand_bits(rng, 30)
    ov, n = check(pdt, orc, pdt.MODE_ARGOS, s)
    assert ov > 0 and n >= 390
    # sparse ARGOS packets: tile path, more than 7 hits in some tiles
    s = "".join(SYNC_ARGOS + rand_bits(rng, 56) + "1" * int(rng.integers(100, 300)) for _ in range(300))
    ov, n = check(pdt, orc, pdt.MODE_ARGOS, s)
    assert ov == 0 and n >= 290


def test_random_bit_soup(pdt, orc):
    rng = np.random.default_rng(5)
    for n in (5000, 100000, 1_000_000):
        check(pdt, orc, pdt.MODE_POES, rand_bits(rng, n))
        check(pdt, orc, pdt.MODE_ARGOS, rand_bits(rng, n))


def test_more_hits_than_the_lds_filter_holds(pdt, orc):
    """> 8190 sync hits (a capture longer than ~14 minutes): the successor-link / pointer-doubling filter runs out of
    global memory instead of LDS.  9 500 frames, some of them overlapping candidates and inverse ones."""
    rng = np.random.default_rng(11)
    parts = [rand_bits(rng, 333)]
    for i in range(9500):
        body = rand_bits(rng, 813)
        if i % 97 == 0:
            body = body[:400] + SYNC_POES + body[419:]          # a sync word inside an open frame: ignored
        parts.append((inv(SYNC_POES) if i % 41 == 0 else SYNC_POES) + (inv(body) if i % 41 == 0 else body))
        if i % 13 == 0:
            parts.append(rand_bits(rng, int(rng.integers(1, 50))))
    ov, n = check(pdt, orc, pdt.MODE_POES, "".join(parts))
    assert n >= 9500 and ov == 0


@pytest.mark.parametrize("piece", [4160, 77])
def test_the_references_own_harness_bits(pdt, orc, piece):
    """The literal bit string of the reference's commented-out harness (POESTIPdemod/ByteSync.c:6-14) against the
    frames the reference's ByteSync object printed for it (tests/golden/make_bytesync_vector.py)."""
    from conftest import golden_text
    bitstr = golden_text("bytesync_harness_bits.txt").strip().decode()
    check(pdt, orc, pdt.MODE_POES, bitstr, piece)
    with pdt.Demodulator(pdt.MODE_POES, 50000) as d:
        d.bytesync(to_bits(bitstr))
        assert d.text() == golden_text("bytesync_harness_frames.txt")
