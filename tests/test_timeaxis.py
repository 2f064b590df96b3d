"""Closed-form time axis (pdt_time_axis) vs the brute-force running sum the reference performs
(common/wave.c:91,96-97,167-168), including the binade crossings and the stall (Q1)."""
import numpy as np
import pytest


def brute_f32(rate, n):
    ts = np.float32(1.0 / float(np.float32(rate)))
    return np.cumsum(np.full(n, ts, dtype=np.float32), dtype=np.float32)     # sequential float32 accumulation


@pytest.mark.parametrize("rate,n", [(50000, 27_000_000), (250000, 35_000_000), (18750, 2_000_000), (32000, 4_000_000),
                                    (48000, 3_000_000), (44100, 3_000_000)])
def test_float_axis_matches_running_sum(pdt, rate, n):
    t = brute_f32(rate, n)
    rng = np.random.default_rng(rate)
    idx = np.unique(np.concatenate([np.arange(1, 2000), rng.integers(1, n, 4000), [n - 1, n],
                                    # neighbourhoods of every power-of-two crossing
                                    *[np.flatnonzero(np.diff(np.frexp(t)[1]))[:64] + d for d in (-1, 0, 1, 2, 3)]]))
    idx = idx[(idx >= 1) & (idx <= n)]
    for m in idx:
        assert np.float32(pdt.time_axis(pdt.MODE_POES, rate, int(m))) == t[m - 1], f"m={m}"
    assert pdt.time_axis(pdt.MODE_POES, rate, 0) == 0.0


def test_stall_values(pdt):
    # the float accumulator stalls at 512.0 s (50 ksps) and 128.0 s (250 ksps)
    assert pdt.time_axis(pdt.MODE_POES, 50000, 10**12) == 512.0
    assert pdt.time_axis(pdt.MODE_POES, 250000, 10**12) == 128.0
    t = brute_f32(50000, 26_500_000)
    first = int(np.argmax(t == np.float32(512.0)))
    assert t[-1] == 512.0 and t[first - 1] < 512.0
    assert first < 25_600_000          # the accumulator runs fast in the upper binades, then stops
    assert np.float32(pdt.time_axis(pdt.MODE_POES, 50000, first)) == t[first - 1]


def test_double_axis_matches_running_sum(pdt):
    rate, n = 32000, 3_000_000
    t = np.cumsum(np.full(n, 1.0 / rate, dtype=np.float64))
    rng = np.random.default_rng(0)
    for m in np.concatenate([np.arange(1, 500), rng.integers(1, n, 3000), [n]]):
        assert pdt.time_axis(pdt.MODE_ARGOS, rate, int(m)) == t[m - 1]
