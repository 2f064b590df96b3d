"""End-to-end A/B on ONE box, interleaved: the c3 WAV on tmpfs -> minor-frame file through pdt_demod_file, one context per
configuration (the developer switches are read when a context is opened), the configurations taken round robin for a number of
rounds so that the neighbours' noise on a shared host hits all of them alike.  Prints median / min / all per configuration.
Usage: python tools/e2e_ab.py [rounds] name=ENV1:VAL1,ENV2:VAL2 ...   (e.g. plain=PDT_NO_OVERLAP:1 s4=PDT_OVERLAP_SPLIT:0.42/0.27/0.18/0.13;
AB_QUALITY:1 in a configuration = pdt_keep_quality on, the per-chunk reports the CLI programs ask for)"""
import importlib, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pdt = importlib.import_module("project-desert-tortoise_amd")

rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
specs = [a for a in sys.argv[1:] if "=" in a] or ["plain=PDT_NO_OVERLAP:1", "default="]
fs, n = 250000, 900_000_000
secs = float(os.environ.get("AB_SECONDS", "3600"))
n = int(secs * fs)
tmp = tempfile.mkdtemp(dir=os.environ.get("AB_DIR", "/dev/shm"), prefix="pdt_ab_")      # (AB_DIR=/tmp: the capture on a disk file system)
wav = os.path.join(tmp, "c3.wav")
par = bench.capture_params(pdt, "c3", 1234)
bench.make_capture(pdt, par, n, 32, wav_path=wav, fs=fs)
ctxs = {}
evict = set()
if os.environ.get("AB_DIR"):
    fdw = os.open(wav, os.O_RDWR)
    os.fsync(fdw)
    os.close(fdw)
for spec in specs:
    name, _, envs = spec.partition("=")
    env = {}
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition(":")
        env[k] = v.replace("/", ",")
    quality = env.pop("AB_QUALITY", None)                 # (not a switch of the library: pdt_keep_quality, what the CLI runs with)
    if env.pop("AB_EVICT", None):                         # (not a switch either: the file's pages dropped from the page cache before every run)
        evict.add(name)
    os.environ.update(env)
    ctxs[name] = pdt.Demodulator(0, fs, device=0).keep_pll(False)
    if quality:
        ctxs[name].keep_quality()
    for k in env:
        os.environ.pop(k)
times = {k: [] for k in ctxs}
texts = {}
for r in range(rounds + 1):
    for name, d in ctxs.items():
        outp = os.path.join(tmp, f"o_{name}.txt")
        t0 = time.perf_counter()
        fd = os.open(wav, os.O_RDONLY)
        if name in evict:
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            t0 = time.perf_counter()
        fo = os.open(outp, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
        d.demod_file_text(fd, 44, n, fo, 0)
        os.close(fd)
        os.close(fo)
        dt = (time.perf_counter() - t0) * 1e3
        if r:
            times[name].append(dt)
        else:
            texts[name] = open(outp, "rb").read()
        os.unlink(outp)
same = len(set(texts.values())) == 1
for name, t in times.items():
    s = sorted(t)
    st = ctxs[name].stats()
    print(f"{name:>14}: median {s[len(s) // 2]:7.2f}  min {s[0]:7.2f}  ingest {st.ingest_ms:6.1f} (direct {st.ingest_direct}, node {st.ingest_numa_node})  all " + " ".join(f"{x:.1f}" for x in t))
print("texts identical:", same)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
