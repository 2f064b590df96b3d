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


def test_pass_shape_is_a_pure_function_of_the_index(pdt):
    """Round 6: noise lead and tail, linear Doppler ramp (integer phase accumulator in closed form), amplitude envelope -- any
    sample is still a pure function of its index, and the shape is what it says: no carrier outside [signal_start, signal_end),
    the carrier offset at the two ends of the ramp, the envelope's floor at the horizon and 1 at culmination."""
    import ctypes as C
    fs, n = 250000, 500000
    p = pdt.synth_params(0, fs, 1000.0, 77)
    pdt.synth_lib().pdt_synth_set_pass(C.byref(p), 50000, 450000, 3000.0, -3000.0, 0.25)
    p.noise_gain = 0                                                   # (the carrier alone)
    whole = np.zeros((n, 2), dtype="<i2")
    pdt.synth_lib().pdt_synth_fill(C.byref(p), 0, n, whole.ctypes.data)
    part = np.zeros((1234, 2), dtype="<i2")
    pdt.synth_lib().pdt_synth_fill(C.byref(p), 333333, 1234, part.ctypes.data)
    assert np.array_equal(whole[333333:333333 + 1234], part)
    assert not whole[:50000].any() and not whole[450000:].any() and whole[50000:450000].any()
    z = whole[:, 0].astype(float) + 1j * whole[:, 1].astype(float)
    amp = np.abs(z)
    assert abs(amp[50010] / 9830 - 0.25) < 0.01 and abs(amp[250000] / 9830 - 1.0) < 0.01 and abs(amp[449990] / 9830 - 0.25) < 0.01

    def carrier_hz(at):                                                # the modulation is +-m: its square has twice the carrier's phase ramp
        w = z[at:at + 4000] ** 2
        return float(np.angle(np.sum(w[1:] * np.conj(w[:-1]))) / 2.0 * fs / (2 * np.pi))
    assert abs(carrier_hz(50000) - 3000.0) < 60 and abs(carrier_hz(248000)) < 60 and abs(carrier_hz(445000) + 2950.0) < 80
    # every field off: the captures of rounds 1 - 5 bit for bit (the goldens under tests/golden depend on it)
    q = pdt.synth_params(0, 50000, 1000.0, 9)
    assert (q.signal_end, q.doppler_q32, q.env_floor_q15) == (0, 0, 0)
