"""Batched many-capture mode (pdt_demod_batch_device): several captures demodulated together -- one launch
per stage for the whole batch, the capture index in blockIdx.z -- give, per context, exactly what the ORACLE computes for
each capture alone: every intermediate stream, the text, the counters."""
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import check_all_stages

pytestmark = pytest.mark.gpu


def to_dev(iq):
    return torch.from_numpy(iq.reshape(-1).copy()).to("cuda:0") if len(iq) else torch.zeros(4, dtype=torch.int16, device="cuda:0")


def test_batch_matches_the_oracle_per_capture(pdt, orc, clip):
    rate, clip_iq = clip
    caps = [pdt.synth_capture(0, 50000, secs, seed=seed) for secs, seed in ((6.0, 1), (9.5, 2), (4.0, 3))]
    caps.append(np.ascontiguousarray(clip_iq))                       # rate 50000 as well
    caps.append(np.zeros((0, 2), dtype=np.int16))                    # an empty capture in the middle of a batch
    caps.append(pdt.synth_capture(0, 50000, 7.0, seed=4))
    caps.append(pdt.synth_capture(0, 50000, 0.35, seed=5))           # shorter than four chunks: the sequential sampler's plan
    oracles = [orc.Oracle(orc.POES, 50000, iq) for iq in caps]
    dev = [to_dev(iq) for iq in caps]
    torch.cuda.synchronize()
    ds = [pdt.Demodulator(pdt.MODE_POES, 50000) for _ in caps]
    try:
        for rep in range(2):                                         # contexts are reusable
            pdt.demod_batch(ds, [t.data_ptr() for t in dev], [len(iq) for iq in caps])
            for d, o in zip(ds, oracles):
                check_all_stages(pdt, orc, d, o)
        with pytest.raises(pdt.PdtError):
            pdt.demod_batch([ds[0], ds[0]], [dev[0].data_ptr()] * 2, [len(caps[0])] * 2)     # one context per capture
        pdt.demod_batch([], [], [])
    finally:
        for d in ds:
            d.close()


def test_batch_of_equal_captures_and_weak_signal(pdt, orc):
    """Eight slots, two of them noisy enough for seam repairs: the repair rounds of one capture must not disturb the
    others that share its launches."""
    import ctypes as C
    fs, secs = 50000, 12.0
    n = int(fs * secs)
    caps = []
    for k in range(8):
        p = pdt.synth_params(0, fs, 1000.0 - 250.0 * k, 100 + k)
        if k in (2, 5):
            p.noise_gain = int(p.noise_gain * 6)
        iq = np.zeros((n, 2), dtype="<i2")
        pdt.synth_lib().pdt_synth_fill(C.byref(p), 0, n, iq.ctypes.data)
        caps.append(iq)
    oracles = [orc.Oracle(orc.POES, fs, iq) for iq in caps]
    dev = [to_dev(iq) for iq in caps]
    torch.cuda.synchronize()
    # (without the walkers' consensus -- round 4 -- the noisy slots lose walkers and their repairs really run; with it, the default,
    # the same result with fewer of them)
    for lone in (True, False):
        if lone:
            os.environ["PDT_PLL_NOCONSENSUS"] = "1"
        try:
            ds = [pdt.Demodulator(pdt.MODE_POES, fs) for _ in caps]
        finally:
            os.environ.pop("PDT_PLL_NOCONSENSUS", None)
        try:
            pdt.demod_batch(ds, [t.data_ptr() for t in dev], [n] * len(caps))
            for d, o in zip(ds, oracles):
                check_all_stages(pdt, orc, d, o)
            if lone:
                assert ds[2].stats().pll_seam_fixes + ds[5].stats().pll_seam_fixes >= 2
        finally:
            for d in ds:
                d.close()


def test_batch_with_fades_and_noise_tails(pdt, orc):
    """Round 6, last build: in a batch the tail pass (k_pll_tail) takes the g-th stretch of capture z in workgroup
    (g - 41 z) mod 1024, so that the long walkers of a batch of passes do not meet on one shader engine.  Six slots, each with its own
    fades and its own loss of signal (small PLL blocks: every stretch spans many seams), every stage of every slot equal to the
    oracle on that slot's capture, and the tail pass ran."""
    import ctypes as C
    fs, seg_s = 50000, 1.5
    n = int(fs * seg_s)
    shapes = ([1, 1, 0, 1, 1, 0.06, 1, 0, 0], [1, 1, 1, 1, 0, 0, 0, 0, 0], [1, 0.06, 0.06, 1, 1, 1, 0, 1, 1],
              [1, 1, 1, 1, 1, 1, 1, 1, 0], [1, 0, 1, 0, 1, 0, 1, 0, 1], [1, 1, 1, 0.06, 0, 0.06, 1, 1, 0])
    caps = []
    for z, amps in enumerate(shapes):
        parts = []
        for k, amp in enumerate(amps):
            p = pdt.synth_params(0, fs, -1500.0 + 500.0 * z, 4800 + z)
            p.amplitude = int(round(p.amplitude * amp))
            iq = np.zeros((n, 2), dtype="<i2")
            pdt.synth_lib().pdt_synth_fill(C.byref(p), k * n, n, iq.ctypes.data)
            parts.append(iq)
        caps.append(np.concatenate(parts))
    oracles = [orc.Oracle(orc.POES, fs, iq) for iq in caps]
    dev = [to_dev(iq) for iq in caps]
    torch.cuda.synchronize()
    ds = [pdt.Demodulator(pdt.MODE_POES, fs, pll_block=1024, profile=(z == 0)) for z in range(len(caps))]
    try:
        pdt.demod_batch(ds, [t.data_ptr() for t in dev], [len(iq) for iq in caps])
        for d, o in zip(ds, oracles):
            check_all_stages(pdt, orc, d, o)
        assert "pll_tail" in ds[0].kernel_times()
        assert all(d.stats().pll_seam_fixes >= 20 for d in ds), [d.stats().pll_seam_fixes for d in ds]
    finally:
        for d in ds:
            d.close()


def test_batch_mixed_modes(pdt, orc):
    p = pdt.synth_capture(0, 50000, 5.0, seed=11)
    a = pdt.synth_capture(1, 32000, 10.0, f0_hz=140.0, seed=12)
    a2 = pdt.synth_capture(1, 32000, 13.0, f0_hz=110.0, seed=13)
    op = orc.Oracle(orc.POES, 50000, p)
    oa = orc.Oracle(orc.ARGOS, 32000, a, math_mode=orc.MATH_LIBM)
    oa2 = orc.Oracle(orc.ARGOS, 32000, a2, math_mode=orc.MATH_LIBM)
    with pdt.Demodulator(pdt.MODE_POES, 50000) as d0, pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d1, \
            pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d2:
        tp, ta, ta2 = to_dev(p), to_dev(a), to_dev(a2)
        torch.cuda.synchronize()
        pdt.demod_batch([d1, d0, d2], [ta.data_ptr(), tp.data_ptr(), ta2.data_ptr()], [len(a), len(p), len(a2)])
        check_all_stages(pdt, orc, d0, op)
        check_all_stages(pdt, orc, d1, oa)
        check_all_stages(pdt, orc, d2, oa2)
        assert d0.stats().frames > 0 and d1.stats().frames > 0 and d2.stats().frames > 0
