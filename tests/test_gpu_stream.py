"""Streaming front end (pdt_stream_*): pushing a capture block by block yields, in order and exactly
once, the frames of one whole-capture demodulation -- the guarantee stated in include/pdt.h."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def stream_all(d, iq, block):
    d.stream_begin()
    parts, when = [], []
    for i in range(0, len(iq), block):
        new = d.stream_push(iq[i:i + block])
        if len(new):
            parts.append(new)
            when.append(i + block)
    tail = d.stream_end()
    parts.append(tail)
    return np.concatenate(parts) if parts else tail, when, len(tail)


@pytest.mark.parametrize("block", [2400, 10000, 7777, 50001])
def test_clip_in_blocks_equals_one_shot(pdt, clip, block):
    rate, iq = clip
    with pdt.Demodulator(pdt.MODE_POES, rate) as ref:
        ref.demod(iq)
        want = ref.frames_array()
        want_text = ref.text()
    with pdt.Demodulator(pdt.MODE_POES, rate) as d:
        got, when, n_tail = stream_all(d, iq, block)
        assert got.tobytes() == want.tobytes()
        assert d.text() == want_text                      # after stream_end the context describes the whole capture
        if block <= 10000:
            assert len(when) >= 5 and n_tail <= 3         # frames arrive while the capture is still being pushed
        got2, _, _ = stream_all(d, iq, block)             # a context can stream again
        assert got2.tobytes() == want.tobytes()


def test_synthetic_minute_in_portaudio_blocks(pdt):
    """2 400-frame blocks as the reference's real-time twin reads them (POESTIPdemodPortAudio/main.c:326)."""
    fs = 50000
    iq = pdt.synth_capture(0, fs, 20.0, seed=17)
    with pdt.Demodulator(pdt.MODE_POES, fs) as ref:
        ref.demod(iq)
        want = ref.frames_array()
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        got, when, n_tail = stream_all(d, iq, 2400)
    assert got.tobytes() == want.tobytes() and len(want) >= 198
    # latency: every frame is reported within ~3 chunks (0.6 s) of its end
    assert n_tail <= 4


def test_raw_float_and_argos_streams(pdt):
    fs = 50000
    iq = pdt.synth_capture(0, fs, 6.0, seed=3)
    raw = iq.astype(np.float32) / np.float32(32768.0)
    with pdt.Demodulator(pdt.MODE_POES, fs) as ref:
        ref.demod_raw(raw)
        want = ref.frames_array()
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        got, _, _ = stream_all(d, raw, 4096)
        assert got.tobytes() == want.tobytes() and len(want) >= 55
        d.stream_begin()
        d.stream_push(raw[:100])
        with pytest.raises(pdt.PdtError):
            d.stream_push(iq[:100])                       # the sample format is fixed by the first push
    a = pdt.synth_capture(1, 32000, 12.0, f0_hz=130.0, seed=9)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as ref:
        ref.demod(a)
        want = ref.frames_array()
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        got, _, _ = stream_all(d, a, 2400)
        assert got.tobytes() == want.tobytes() and len(want) >= 5
        with pytest.raises(pdt.PdtError):
            d.stream_begin().stream_push(np.zeros((10, 2), dtype=np.float32))   # ARGOS refuses RAW input


def test_empty_stream(pdt):
    with pdt.Demodulator(pdt.MODE_POES, 50000) as d:
        d.stream_begin()
        assert len(d.stream_end()) == 0 and d.stats().frames == 0
