"""The chunk loop's progress / quality line (POESTIPdemod/main.c:457-481, ARGOSdemod/main.c:286-296): CarrierTrackPLL's return
value (averagePhase after every chunk, CarrierTrackingPLL.c:277) bit for bit, the per-chunk symbol / bit / frame counts, and the
text the host program prints -- against golden vectors made by the reference's own objects (tests/golden/make_progress_golden.py)
and, where oracle/_ref travelled with the repository, against those objects run on fresh captures."""
import os
import subprocess

import numpy as np
import pytest
import torch        # before libpdt is loaded (as in test_gpu_batch.py): torch brings its own HIP runtime

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
REF_POES = os.path.join(ROOT, "oracle/_ref/ref_demodPOES")
REF_ARGOS = os.path.join(ROOT, "oracle/_ref/ref_demodARGOS")
have_ref = os.path.exists(REF_POES) and os.path.exists(REF_ARGOS)


def golden(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def progress_of(stdout: bytes) -> bytes:
    """the progress lines of a demodPOES / demodARGOS run: from the first carriage return to the newline in front of the summary"""
    a = stdout.index(b"\r")
    b = stdout.index(b"\n100.0% ", a)
    return stdout[a:b]


def check_reports(rep, avg_ref, counts, dt):
    nc = len(rep)
    assert len(avg_ref) in (nc, nc + 1)             # + the loop's extra pass with zero samples (data end on a chunk boundary)
    assert rep["avg_phase"].astype(dt).tobytes() == avg_ref[:nc].tobytes(), "averagePhase differs from the reference's"
    if len(avg_ref) == nc + 1:
        assert avg_ref[nc].tobytes() == avg_ref[nc - 1].tobytes()
    if counts is not None:
        counts = counts[:nc]
        assert np.array_equal(rep["samples"], counts[:, 0])
        assert np.array_equal(rep["symbols"], counts[:, 1])
        assert np.array_equal(rep["bits"], counts[:, 2])


@pytest.mark.parametrize("chunk", [10000, 1000])
def test_clip_average_phase_golden(pdt, clip, chunk):
    rate, iq = clip
    avg_ref = np.frombuffer(golden(f"clip.c{chunk}.avg.f32"), dtype="<f4")
    with pdt.Demodulator(pdt.MODE_POES, rate, chunk=chunk) as d:
        d.keep_quality().demod(iq)
        rep = d.chunk_reports()
        assert len(rep) == (len(iq) + chunk - 1) // chunk
        check_reports(rep, avg_ref, None, "<f4")
        st = d.stats()
        assert rep["symbols"].sum() == st.symbols and rep["bits"].sum() == st.bits and rep["frames"].sum() == st.frames
        # off again: no reports, same frames
        text = d.text()
        d.keep_quality(False).demod(iq)
        assert len(d.chunk_reports()) == 0 and d.text() == text


def test_argos_average_phase_golden(pdt):
    iq = pdt.synth_capture(1, 32000, 13.0, seed=99)
    avg_ref = np.frombuffer(golden("argos_32000.avg.f64"), dtype="<f8")
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        d.keep_quality().demod(iq)
        check_reports(d.chunk_reports(), avg_ref, None, "<f8")


@pytest.mark.parametrize("name,args", [("clip.c10000", []), ("clip.c1000", ["-c", "1000"])])
def test_cli_progress_lines_clip(tmp_path, name, args):
    r = subprocess.run([os.path.join(ROOT, "bin", "demodPOES"), *args, "-o", str(tmp_path / "o.txt"), os.path.join(GOLDEN, "5sec_clip.wav")],
                       capture_output=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert progress_of(r.stdout) == golden(name + ".progress")
    assert (tmp_path / "o.txt").read_bytes() == golden(name + ".txt")


def test_cli_progress_lines_exact_multiple(pdt, tmp_path):
    """150 000 samples = 15 chunks exactly: the reference's loop runs a 16th time with zero samples and prints the line again"""
    iq = pdt.synth_capture(0, 50000, 3.0, seed=1234)
    wav = tmp_path / "p.wav"
    pdt.write_wav(str(wav), 50000, iq)
    r = subprocess.run([os.path.join(ROOT, "bin", "demodPOES"), "-o", str(tmp_path / "o.txt"), str(wav)], capture_output=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert progress_of(r.stdout) == golden("poes_50000.progress")
    # -P: no progress lines, same output file
    r2 = subprocess.run([os.path.join(ROOT, "bin", "demodPOES"), "-P", "-o", str(tmp_path / "o2.txt"), str(wav)], capture_output=True)
    assert r2.returncode == 0 and b"\r" not in r2.stdout
    assert (tmp_path / "o2.txt").read_bytes() == (tmp_path / "o.txt").read_bytes() == golden("poes_50000.txt")


def test_cli_progress_lines_argos(pdt, tmp_path):
    iq = pdt.synth_capture(1, 32000, 13.0, seed=99)
    wav = tmp_path / "a.wav"
    pdt.write_wav(str(wav), 32000, iq)
    r = subprocess.run([os.path.join(ROOT, "bin", "demodARGOS"), "-o", str(tmp_path / "o.txt"), str(wav)], capture_output=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    a = out.index(b"\r")
    b = out.index(b"\n", a)
    assert out[a:b] == golden("argos_32000.progress")


def ref_dump(binary, wav, tmp_path, extra=()):
    dump = tmp_path / "refdump"
    subprocess.run([binary, *extra, "-d", str(dump), str(wav), str(tmp_path / "ref.txt")], check=True, capture_output=True)
    counts = np.loadtxt(f"{dump}.counts", dtype=np.int64, ndmin=2)
    return dump, counts


@pytest.mark.skipif(not have_ref, reason="oracle/_ref not built")
@pytest.mark.parametrize("fs,seconds,seed,f0,chunk,noise", [
    (50000, 20.0, 21, -2300.0, 10000, 1.0),        # 100 chunks: many EMA blocks after the lock
    (250000, 6.0, 22, 3100.0, 10000, 1.0),
    (50000, 8.0, 23, 900.0, 777, 1.0),             # chunk ends inside EMA blocks, short last chunk
    (50000, 6.0, 24, 1500.0, 10000, 40.0),         # noise only: never locks, the acquisition delivers every value
])
def test_average_phase_and_counts_vs_reference_objects(pdt, tmp_path, fs, seconds, seed, f0, chunk, noise):
    iq = pdt.synth_capture(0, fs, seconds, f0_hz=f0, seed=seed)
    if noise != 1.0:
        rng = np.random.default_rng(seed)
        iq = np.clip(rng.normal(0, 3000, iq.shape), -32768, 32767).astype("<i2")
    wav = tmp_path / "s.wav"
    pdt.write_wav(str(wav), fs, iq)
    dump, counts = ref_dump(REF_POES, wav, tmp_path, ["-c", str(chunk)])
    avg_ref = np.fromfile(f"{dump}.avg", dtype="<f4")
    with pdt.Demodulator(pdt.MODE_POES, fs, chunk=chunk) as d:
        d.keep_quality().demod(iq)
        rep = d.chunk_reports()
        check_reports(rep, avg_ref, counts, "<f4")
        if noise != 1.0:
            assert d.stats().lock_sample < 0
    r = subprocess.run([os.path.join(ROOT, "bin", "demodPOES"), "-c", str(chunk), "-o", str(tmp_path / "o.txt"), str(wav)], capture_output=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert progress_of(r.stdout) == open(f"{dump}.progress", "rb").read()


@pytest.mark.skipif(not have_ref, reason="oracle/_ref not built")
@pytest.mark.parametrize("chunk", [2400, 1000])
def test_argos_reports_vs_reference_objects(pdt, tmp_path, chunk):
    iq = pdt.synth_capture(1, 32000, 20.0, f0_hz=160.0, seed=31)
    wav = tmp_path / "a.wav"
    pdt.write_wav(str(wav), 32000, iq)
    dump, counts = ref_dump(REF_ARGOS, wav, tmp_path, ["-c", str(chunk)])
    avg_ref = np.fromfile(f"{dump}.avg", dtype="<f8")
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000, chunk=chunk) as d:
        d.keep_quality().demod(iq)
        check_reports(d.chunk_reports(), avg_ref, counts, "<f8")
    r = subprocess.run([os.path.join(ROOT, "bin", "demodARGOS"), "-c", str(chunk), "-o", str(tmp_path / "o.txt"), str(wav)], capture_output=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    a = out.index(b"\r")
    b = out.index(b"\n", a)
    assert out[a:b] == open(f"{dump}.progress", "rb").read()


def test_batch_with_quality(pdt, clip):
    """the batched entry point delivers the same reports"""
    rate, iq = clip
    iq = np.ascontiguousarray(iq)
    avg_ref = np.frombuffer(golden("clip.c10000.avg.f32"), dtype="<f4")
    caps = [iq, iq[:123456], iq]
    dev = [torch.from_numpy(c.reshape(-1).copy()).to("cuda:0") for c in caps]
    torch.cuda.synchronize()
    ds = [pdt.Demodulator(pdt.MODE_POES, rate) for _ in caps]
    try:
        for d in ds:
            d.keep_quality()
        pdt.demod_batch(ds, [t.data_ptr() for t in dev], [len(c) for c in caps])
        check_reports(ds[0].chunk_reports(), avg_ref, None, "<f4")
        check_reports(ds[2].chunk_reports(), avg_ref, None, "<f4")
        assert len(ds[1].chunk_reports()) == 13
        # causal: the first twelve (whole) chunks of the cut capture see what the full one sees
        assert ds[1].chunk_reports()["avg_phase"][:12].astype("<f4").tobytes() == avg_ref[:12].tobytes()
        # a batch whose contexts differ in what they keep (plans of different shapes): every context still gets its own
        text = [d.text() for d in ds]
        ds[1].keep_quality(False)
        pdt.demod_batch(ds, [t.data_ptr() for t in dev], [len(c) for c in caps])
        check_reports(ds[0].chunk_reports(), avg_ref, None, "<f4")
        assert len(ds[1].chunk_reports()) == 0
        check_reports(ds[2].chunk_reports(), avg_ref, None, "<f4")
        assert [d.text() for d in ds] == text
    finally:
        for d in ds:
            d.close()
