"""Batched many-capture mode (pdt_demod_batch_device): several captures enqueued together give, per
context, exactly the result of demodulating each one alone."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batch_equals_individual_runs(pdt, clip):
    rate, clip_iq = clip
    caps = [pdt.synth_capture(0, 50000, secs, seed=seed) for secs, seed in ((6.0, 1), (9.5, 2), (4.0, 3))]
    caps.append(np.ascontiguousarray(clip_iq))                       # rate 50000 as well
    caps.append(np.zeros((0, 2), dtype=np.int16))                    # an empty capture in the middle of a batch
    caps.append(pdt.synth_capture(0, 50000, 7.0, seed=4))
    want = []
    for iq in caps:
        with pdt.Demodulator(pdt.MODE_POES, 50000) as d:
            d.demod(iq)
            want.append((d.text(), d.frames_array().tobytes(), d.stats().symbols))
    dev = [torch.from_numpy(iq.reshape(-1).copy()).to("cuda:0") if len(iq) else torch.zeros(4, dtype=torch.int16, device="cuda:0")
           for iq in caps]
    torch.cuda.synchronize()
    ds = [pdt.Demodulator(pdt.MODE_POES, 50000) for _ in caps]
    try:
        for rep in range(2):                                         # contexts are reusable
            pdt.demod_batch(ds, [t.data_ptr() for t in dev], [len(iq) for iq in caps])
            for d, w in zip(ds, want):
                assert (d.text(), d.frames_array().tobytes(), d.stats().symbols) == w
        with pytest.raises(pdt.PdtError):
            pdt.demod_batch([ds[0], ds[0]], [dev[0].data_ptr()] * 2, [len(caps[0])] * 2)     # one context per capture
        pdt.demod_batch([], [], [])
    finally:
        for d in ds:
            d.close()


def test_batch_mixed_modes(pdt):
    p = pdt.synth_capture(0, 50000, 5.0, seed=11)
    a = pdt.synth_capture(1, 32000, 10.0, f0_hz=140.0, seed=12)
    with pdt.Demodulator(pdt.MODE_POES, 50000) as d0, pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d1:
        d0.demod(p); d1.demod(a)
        want = (d0.text(), d1.text())
        tp = torch.from_numpy(p.reshape(-1).copy()).to("cuda:0")
        ta = torch.from_numpy(a.reshape(-1).copy()).to("cuda:0")
        torch.cuda.synchronize()
        pdt.demod_batch([d1, d0], [ta.data_ptr(), tp.data_ptr()], [len(a), len(p)])
        assert (d0.text(), d1.text()) == want and len(want[0]) > 0 and len(want[1]) > 0
