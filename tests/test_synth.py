"""Synthetic capture generator: integer-only, position-independent, pinned by digests."""
import hashlib
import zlib

import numpy as np


def test_sine_table_pinned(pdt, golden):
    tab = np.ctypeslib.as_array(pdt.synth_lib().pdt_synth_sine_table(), shape=(65536,))
    assert zlib.crc32(tab.tobytes()) == golden["synth"]["sine_table_crc32"]
    assert tab[0] == 0 and tab[16384] == 32767 and tab[49152] == -32767


def test_random_access_equals_sequential(pdt):
    a = pdt.synth_capture(0, 50000, 1.0, seed=9)
    b = pdt.synth_capture(0, 50000, 0.5, seed=9, start=25000)
    assert np.array_equal(a[25000:], b)
    c = pdt.synth_capture(1, 32000, 2.0, seed=9)
    d = pdt.synth_capture(1, 32000, 1.0, seed=9, start=32000)
    assert np.array_equal(c[32000:], d)


def test_seeds_differ_and_levels(pdt):
    a = pdt.synth_capture(0, 50000, 0.2, seed=1)
    b = pdt.synth_capture(0, 50000, 0.2, seed=2)
    assert not np.array_equal(a, b)
    amp = np.hypot(a[:, 0].astype(float), a[:, 1].astype(float))
    assert 9000 < amp.mean() < 10700          # 0.3 full scale carrier, 20 dB noise


def test_wav_header(pdt, tmp_path):
    iq = pdt.synth_capture(0, 50000, 0.1, seed=1)
    p = tmp_path / "x.wav"
    pdt.write_wav(str(p), 50000, iq)
    rate, back = pdt.read_wav(str(p))
    assert rate == 50000 and np.array_equal(back, iq)
    raw = p.read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data" and len(raw) == 44 + 4 * len(iq)
