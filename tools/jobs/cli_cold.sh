#!/bin/bash
# the C host program from process start, with its own timing (-T), default (per-chunk reports) and -P
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/cli
export TMPDIR=/tmp
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cli/cli_cold.txt
import importlib, json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import bench
pdt = importlib.import_module("project-desert-tortoise_amd")
tmp = tempfile.mkdtemp(dir="/dev/shm", prefix="pdt_cli_")
wav = os.path.join(tmp, "c3.wav")
n = 900_000_000
bench.make_capture(pdt, bench.capture_params(pdt, "c3", 1234), n, 32, wav_path=wav, fs=250000)
for args in ([], ["-P"]):
    for rep in range(3):
        out = os.path.join(tmp, "o.txt")
        time.sleep(3.0)          # (the previous process's buffers go back to the driver in the background)
        t0 = time.perf_counter()
        r = subprocess.run(["bin/demodPOES", "-T"] + args + ["-o", out, wav], capture_output=True, text=True)
        dt = (time.perf_counter() - t0) * 1e3
        line = [l for l in r.stderr.splitlines() if l.startswith('{"timing_ms"')]
        sp = json.loads(line[0])["timing_ms"] if line else None
        print("demodPOES", " ".join(args), f"wall {dt:.1f} ms rc {r.returncode}", "loader", round(dt - sp["total"], 1) if sp else None, sp)
import shutil; shutil.rmtree(tmp)
PY
