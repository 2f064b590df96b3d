"""bin/demodMulti's scheduling on a machine without a GPU: host/demod_multi.c linked against tests/fake_pdt.c (stand-ins for
the library calls: a "capture" says how long its demodulation takes and how many frames it yields).  What is checked is the
host logic of round 4: one worker per GPU taking the next capture from a shared queue (no waves), contexts reused from capture to
capture (re-opened when the sample rate changes), ONE gatherer for the whole run, one output file per capture with that
capture's records, failures confined to the capture they happen in."""
import os
import re
import struct
import subprocess

import pytest

from conftest import ROOT


def wav(path, ms, frames, rate=250000, pad=4000, channels=2, ingest_ms=0):
    hdr = b"RIFF" + struct.pack("<I", 36 + pad) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * 4, 4, 16)
    hdr += b"data" + struct.pack("<I", pad)
    assert len(hdr) == 44
    with open(path, "wb") as f:
        f.write(hdr + struct.pack("<III", ms, frames, ingest_ms) + bytes(pad - 12))


def expected_text(nsamples, frames):
    out = []
    for k in range(frames):
        b = lambda i: (nsamples * 7 + k * 13 + i) & 0xFF
        out.append("%.5f %02X %02X %02X\n" % (k * 0.1, b(0), b(1), b(103)))
    return "".join(out)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("multi") / "demodMulti_fake")
    src = [os.path.join(ROOT, "project-desert-tortoise_amd", "host", "demod_multi.c"), os.path.join(ROOT, "tests", "fake_pdt.c")]
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", out] + src + ["-lpthread"], check=True)
    return out


def run(exe, files, devices, extra=("-l", "1")):
    # (-l 1: one context per GPU, the scheduling the round-4 tests below describe; the two-lane default has its own test)
    env = dict(os.environ, FAKE_DEVICES=str(devices))
    r = subprocess.run([exe, *extra, *files], capture_output=True, text=True, env=env, timeout=120)
    took = {}
    for m in re.finditer(r"^GPU (\d+): (\S+):", r.stdout, re.M):
        took[m.group(2)] = int(m.group(1))
    return r, took


def test_a_free_gpu_takes_the_next_capture(exe, tmp_path):
    """One long capture and five short ones on two GPUs: the GPU that is not stuck with the long one demodulates all the
    others (waves of one capture per GPU would have given it two and left it idle for most of the run)."""
    files = []
    for k, (ms, fr) in enumerate([(900, 5), (40, 3), (40, 0), (40, 7), (40, 2), (40, 4)]):
        p = str(tmp_path / f"c{k}.wav")
        wav(p, ms, fr, pad=4000 + 40 * k)
        files.append(p)
    r, took = run(exe, files, 2)
    assert r.returncode == 0, r.stdout + r.stderr
    assert set(took) == set(files)                                   # every capture once
    long_gpu = took[files[0]]
    assert [took[f] for f in files[1:]] == [1 - long_gpu] * 5
    assert "1 gatherer(s)" in r.stderr and "2 context(s) opened, 2 closed" in r.stderr
    for k, (ms, fr) in enumerate([(900, 5), (40, 3), (40, 0), (40, 7), (40, 2), (40, 4)]):
        name = files[k] + ".frames.txt"
        if fr == 0:
            assert not os.path.exists(name)                          # no frame, no file (main.c:508-512)
        else:
            assert open(name).read() == expected_text((4000 + 40 * k) // 4, fr)
    # the run lasts about as long as the long capture, not as long as three waves
    secs = float(re.search(r"in ([0-9.]+) s", r.stdout).group(1))
    assert secs < 1.5


def test_more_gpus_than_captures_and_a_rate_change(exe, tmp_path):
    a, b, c = (str(tmp_path / n) for n in ("a.wav", "b.wav", "c.wav"))
    wav(a, 10, 2)
    wav(b, 10, 3, rate=50000)
    wav(c, 10, 1)
    r, took = run(exe, [a, b, c], 8)
    assert r.returncode == 0 and len(took) == 3
    assert "on 3 MI355X GPU(s)" in r.stdout                          # no more workers than captures
    r, took = run(exe, [a, b, c], 1)                                 # one GPU: its context is re-opened when the rate changes
    assert r.returncode == 0 and set(took.values()) == {0}
    assert "3 context(s) opened, 3 closed" in r.stderr
    for p, n in ((a, 2), (b, 3), (c, 1)):
        assert open(p + ".frames.txt").read() == expected_text(1000, n)


def test_a_bad_capture_costs_only_itself(exe, tmp_path):
    good, bad, missing = str(tmp_path / "g.wav"), str(tmp_path / "mono.wav"), str(tmp_path / "nothing.wav")
    wav(good, 10, 4)
    wav(bad, 10, 4, channels=1)
    r, took = run(exe, [bad, good, missing], 2)
    assert r.returncode == 1
    assert list(took) == [good]
    assert r.stdout.count("unsupported WAV format") == 2
    assert open(good + ".frames.txt").read() == expected_text(1000, 4)


def test_no_gpu(exe, tmp_path):
    p = str(tmp_path / "x.wav")
    wav(p, 1, 1)
    r, _ = run(exe, [p], 0)
    assert r.returncode == 1 and "GPU demodulator unavailable" in r.stdout


def test_two_contexts_per_gpu_hide_the_chain_behind_the_next_ingest(exe, tmp_path):
    """Round 5: two lanes per GPU.  A capture is link time first (its ingest: one per GPU at a time) and GPU time second; with
    one context per GPU a queue of six captures costs 6 x (ingest + chain), with two the chain of capture k runs while capture
    k + 1 arrives: 6 x ingest + one chain.  Same files, same text."""
    import time
    files = []
    for k in range(6):
        p = str(tmp_path / f"q{k}.wav")
        wav(p, 80, 2 + k, pad=4000 + 40 * k, ingest_ms=100)
        files.append(p)
    walls = {}
    for lanes in ("1", "2"):
        t0 = time.perf_counter()
        r, took = run(exe, files, 1, extra=("-l", lanes))
        walls[lanes] = time.perf_counter() - t0
        assert r.returncode == 0, r.stdout + r.stderr
        assert set(took) == set(files) and set(took.values()) == {0}
        assert f"{lanes} context(s) per GPU" in r.stdout
        for k, p in enumerate(files):
            assert open(p + ".frames.txt").read() == expected_text((os.path.getsize(p) - 44) // 4, 2 + k)
        assert f"fake: {lanes} context(s) opened, {lanes} closed, 1 gatherer(s)" in r.stderr
    # 6 x (100 + 80) ms against 6 x 100 + 80 ms: the RELATION of the two runs (absolute bounds failed on loaded hosts) --
    # two lanes hide most of the five chains that have a next ingest to hide behind (5 x 80 ms = 0.4 s; 0.25 s asked for)
    assert walls["1"] > 6 * 0.18 and walls["2"] < walls["1"] - 0.25, walls


def test_lane_option_is_validated(exe, tmp_path):
    p = str(tmp_path / "x.wav")
    wav(p, 1, 1)
    for bad in ("0", "-1", "3", "two"):
        r, _ = run(exe, [p], 1, extra=("-l", bad))
        assert r.returncode == 2 and "contexts per GPU, 1 or 2" in r.stderr, (bad, r.stdout, r.stderr)
    r, took = run(exe, [p], 1, extra=("-l", "2"))                     # fewer captures than lanes: said in so many words
    assert r.returncode == 0 and "one context per GPU" in r.stdout and "1 context(s) per GPU" in r.stdout


def test_repeated_passes_and_the_json_line(exe, tmp_path):
    """-R: the whole queue several times with the contexts and the gatherer kept (what bench.py --gpus N times: the second
    pass on is warm); -J: every pass's clock -- first open to last output file closed, the gather inside -- and every capture's
    ingest and GPU time in one line.  Eight GPUs, eight captures, two passes: eight contexts, one gatherer."""
    import json
    files = []
    for k in range(8):
        p = str(tmp_path / f"r{k}.wav")
        wav(p, 30, 3 + k, pad=4000 + 40 * k, ingest_ms=60)
        files.append(p)
    r, took = run(exe, files, 8, extra=("-l", "2", "-R", "2", "-J"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fake: 8 context(s) opened, 8 closed, 1 gatherer(s)" in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith('{"demodMulti"')]
    assert len(line) == 1
    doc = json.loads(line[0])["demodMulti"]
    assert doc["gpus"] == 8 and doc["lanes"] == 1 and doc["captures"] == 8 and len(doc["passes"]) == 2
    for pss in doc["passes"]:
        assert pss["failed"] == 0 and 0.09 <= pss["until_last_gpu_s"] <= pss["wall_s"] < 1.0
    last = doc["last_pass"]
    assert sorted(c["gpu"] for c in last) == list(range(8)) and all(c["ingest_ms"] == 60.0 and c["gpu_ms"] == 30.0 for c in last)
    for k, p in enumerate(files):
        assert open(p + ".frames.txt").read() == expected_text((os.path.getsize(p) - 44) // 4, 3 + k)


def test_bench_accounts_every_gpus_ingest_in_the_e2e_line(exe, tmp_path):
    """bench.py --gpus N reports BASELINE's end-to-end metric through this launcher (bench.e2e_multi): with the stand-in library
    on eight "GPUs", eight captures -- the line's figure is samples / median wall of the warm passes, every GPU's ingest (bytes,
    ms, GB/s) is accounted, and the gather sits inside the pass's clock."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    files = []
    for k in range(8):
        p = str(tmp_path / f"b{k}.wav")
        wav(p, 20, 5, pad=400000, ingest_ms=50)
        files.append(p)
    doc = bench.e2e_multi(exe, files, 8, passes=3, env=dict(os.environ, FAKE_DEVICES="8"))
    assert "error" not in doc, doc
    assert doc["gpus"] == 8 and doc["captures"] == 8 and len(doc["passes_ms"]) == 3
    assert sorted(doc["per_gpu"]) == [str(g) for g in range(8)]
    for g in doc["per_gpu"].values():
        assert g["captures"] == 1 and g["bytes"] == 400000 and g["ingest_ms"] == 50.0 and g["ingest_GBps"] == round(400000 / 0.05 / 1e9, 2)
    assert doc["samples"] == 8 * 100000 and abs(doc["value"] - doc["samples"] / (doc["ms"] * 1e-3) / 1e6) < 0.01 * doc["value"]
    assert doc["ms"] >= doc["until_last_gpu_ms"] >= 70.0 and doc["ms"] in doc["passes_ms"][1:]
