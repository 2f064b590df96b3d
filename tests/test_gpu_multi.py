"""Multi-capture launcher (bin/demodMulti): capture i on GPU i, every capture demodulated by its own context in its own
thread, frame records gathered with RCCL (libpdtgather: all-gather of the counts, padded all-gather of the records), one
output file per capture -- identical to the oracle's text for that capture.  On a one-GPU box the captures go through in
waves of one (the gather then runs on a one-rank communicator); on an 8-GPU node the same command spreads over the GPUs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_text

pytestmark = pytest.mark.gpu


def test_demodmulti_writes_one_file_per_capture(pdt, orc, tmp_path):
    caps = []
    clip = tmp_path / "clip.wav"
    clip.write_bytes(open(os.path.join(GOLDEN, "5sec_clip.wav"), "rb").read())
    caps.append((str(clip), golden_text("clip.c10000.txt")))
    for k, (secs, seed) in enumerate(((6.0, 41), (3.5, 42))):
        iq = pdt.synth_capture(0, 50000, secs, seed=seed)
        path = tmp_path / f"synth{k}.wav"
        pdt.write_wav(str(path), 50000, iq)
        caps.append((str(path), orc.Oracle(orc.POES, 50000, iq).text()))
    silent = tmp_path / "silence.wav"
    pdt.write_wav(str(silent), 50000, np.zeros((30000, 2), dtype=np.int16))
    r = subprocess.run([os.path.join(ROOT, "bin", "demodMulti")] + [c[0] for c in caps] + [str(silent)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for path, want in caps:
        assert open(path + ".frames.txt", "rb").read() == want
    assert not os.path.exists(str(silent) + ".frames.txt")           # no frame, no file (POESTIPdemod/main.c:508-512)
    assert "4 capture(s)" in r.stdout


def test_demodmulti_argos(pdt, orc, tmp_path):
    a = pdt.synth_capture(1, 32000, 12.0, f0_hz=130.0, seed=9)
    path = tmp_path / "argos.wav"
    pdt.write_wav(str(path), 32000, a)
    r = subprocess.run([os.path.join(ROOT, "bin", "demodMulti"), "-a", str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(str(path) + ".frames.txt", "rb").read() == orc.Oracle(orc.ARGOS, 32000, a).text()
