#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s of the demodulation hot path on N MI355X, with roofline, end-to-end and CPU figures.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  For N > 1 the driver launches it as
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``; started by hand without that wrapper
(no WORLD_SIZE in the environment) it re-executes itself under torch.distributed.run, one rank per GPU.

Workloads (``--config``; BASELINE.json ``configs``) -- THE SAME default for every N, so that value(N) / value(1) is a
like-for-like weak-scaling figure:
  c3    (default)  configs[2] = the per-GPU capture of configs[4]: 250 ksps, 60 min (900 M samples, 3.6 GB), POES chain;
                   the largest single-GPU configuration of BASELINE.json
  c2               configs[1]: synthetic 50 ksps complex-IQ capture, 10 min (30 M samples, 120 MB), POES chain
  argos            configs[3]: synthetic ARGOS capture, 32 ksps, 5 min (9.6 M samples), double precision chain
  aos              (not in BASELINE) 250 ksps, 10 min, the first 60 s noise only: the receiver is switched
                   on before the satellite rises, the PLL sweeps for a minute before its one-time lock
  weak             (not in BASELINE) 250 ksps, 10 min at six times the noise amplitude (4.4 dB SNR in the sampled band)
  pass             (not in BASELINE) 250 ksps, 15 min as a receiver records a pass: a minute of noise, the signal with a
                   Doppler ramp from +3 kHz to -3 kHz, an amplitude envelope of 0.25 .. 1 and a 20 s fade, a minute of noise
  i8, c2h           the interpolating filter at scale: 18.75 ksps (interp 8) and 50 ksps (interp 3), 60 min each
  --captures N     the batched many-capture mode (pdt_demod_batch_device), every slot its own capture, every slot's text checked
One "step" = one pass of the whole hot path (StaticGain, PLL, FIR, AGC, Gardner, Manchester, ByteSync, frame records +
time stamps) over one capture that is ALREADY RESIDENT IN HBM when the timed region starts: that is ``value``, and the
``metric`` string says so.  The figure from the WAV file to the closed output file is ``e2e`` (N = 1).  With N GPUs every
rank demodulates its own independent capture (different seed): weak scaling, no data-path collective; the decoded frame
records are gathered on rank 0 with one padded all_gather (RCCL) after the timed region and checked there.

The JSON line carries, besides the contract keys:
  roofline     the kernel group that dominates the step: algorithmic bytes / live HIP-event duration (libpdt's profile
               mode, events on the stream the kernels run on) against 8 TB/s; ``traffic`` from the committed rocprofv3
               counter passes, printed only when that file was measured with the build that is running (``build`` tags)
  stages       every kernel group: ms per step, algorithmic bytes per step, GB/s, fraction of 8 TB/s
  fir_pll_stage  the north star's stage figure: 4 B in + 4 * interp B out per sample over the PLL + FIR critical path
  parity       what was compared with what in THIS run (a failure is fatal: exit code 1, no JSON line)
  e2e          what the reference program does, timed: open the WAV (tmpfs) -> header -> pdt_demod_fd (threaded read into
               pinned memory, copy to HBM, all kernels, frame records back) -> text -> output file written and closed
  e2e_cli      wall time of the C host program bin/demodPOES|demodARGOS on the same file (process start and HIP
               initialisation included)
  cpu_baseline the reference's own DSP objects (oracle/_ref, kind "reference") or the CPU restatement (kind "port"),
               single thread on this host, on a bounded sample of the same capture: DSP-only rate (``value``, comparable
               with the resident GPU number) and end-to-end rate (``e2e_value``: file read and text output included)
  cpu_baseline_8proc  8 concurrent single-thread CPU processes, one capture each (BASELINE.md section 3, configs[4])
  secondary    the other workloads measured on the same GPU, each a process of its own: c2, argos (with their CPU legs), pass, i8 and
               the batched ARGOS mode (64 captures per step)
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import ctypes as C
import importlib
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

def _cfg(kind, fs, seconds, cpu_seconds, baseline, lead_s=0.0, tail_s=0.0, noise_x=1.0, f_end_hz=None, env_floor=0.0, fade=None):
    return dict(kind=kind, fs=fs, seconds=seconds, cpu_seconds=cpu_seconds, baseline=baseline, lead_s=lead_s, tail_s=tail_s,
                noise_x=noise_x, f_end_hz=f_end_hz, env_floor=env_floor, fade=fade)


CONFIGS = {
    "c2":    _cfg(0, 50000, 600.0, 600.0, "configs[1]"),
    "c3":    _cfg(0, 250000, 3600.0, 600.0, "configs[2] (= one GPU's capture of configs[4])"),
    "argos": _cfg(1, 32000, 300.0, 300.0, "configs[3]"),
    "aos":   _cfg(0, 250000, 600.0, 600.0, "none (receiver on 60 s before the signal rises)", lead_s=60.0),
    "weak":  _cfg(0, 250000, 600.0, 600.0, "none (noise amplitude x6)", noise_x=6.0),
    # a pass as a receiver sees it: a minute of noise, the signal rising out of it -- carrier offset ramping from
    # +3 kHz to -3 kHz (Doppler), amplitude from a quarter at the horizon to full at culmination and back --, a minute of noise
    # ... and a fade of 20 s in the middle (the signal 30 dB down: the loop is on noise there too, and finds the signal again)
    "pass":  _cfg(0, 250000, 900.0, 900.0, "none (15 min pass: 60 s noise, Doppler +3 -> -3 kHz, amplitude envelope 0.25 .. 1, a 20 s fade, 60 s noise)",
                  lead_s=60.0, tail_s=60.0, f_end_hz=-3000.0, env_floor=0.25, fade=(0.5, 20.0, 0.03)),
    # the interpolating filter at scale (north star "FIR + 8x interpolator"): interp = rint(150000 / Fs)
    # (POESTIPdemod/main.c:347) is 8 at 18.75 ksps and 3 at 50 ksps
    "i8":    _cfg(0, 18750, 3600.0, 1200.0, "none (interp 8: 18.75 ksps x 60 min = 67.5 M samples, 540 M filter outputs)"),
    "c2h":   _cfg(0, 50000, 3600.0, 1200.0, "none (configs[1]'s rate for 60 min: interp 3, 180 M samples, 540 M filter outputs)"),
}
DEFAULT_CONFIG = "c3"


# ---------------------------------------------------------------------------------------------------------------------------
def relaunch_under_torchrun(n: int) -> None:
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def capture_params(pdt, cfg: str, seed: int, seconds: float | None = None):
    c = CONFIGS[cfg]
    kind, fs = c["kind"], c["fs"]
    n = int(round((seconds if seconds else c["seconds"]) * fs))
    p = pdt.synth_params(kind, fs, 1000.0 if kind == 0 else 120.0, seed)
    p.signal_start = int(round(c["lead_s"] * fs))
    if c["f_end_hz"] is not None or c["tail_s"] or c["env_floor"]:
        # the pass keeps its shape when --seconds shortens it: lead and tail shrink with it
        scale = min(1.0, n / (c["seconds"] * fs))
        lead, tail = int(round(c["lead_s"] * fs * scale)), int(round(c["tail_s"] * fs * scale))
        f_start = 3000.0 if c["f_end_hz"] is not None else 1000.0
        pdt.synth_lib().pdt_synth_set_pass(C.byref(p), lead, n - tail, f_start, c["f_end_hz"] if c["f_end_hz"] is not None else f_start,
                                           c["env_floor"])
    if c["fade"]:
        at, secs_f, level = c["fade"]                                   # (position as a fraction of the capture, seconds at full length, level)
        scale = min(1.0, n / (c["seconds"] * fs))
        p.fade_start = int(round(at * n))
        p.fade_len = int(round(secs_f * fs * scale))
        p.fade_q15 = int(round(level * 32768))
    p.noise_gain = int(round(p.noise_gain * c["noise_x"]))
    return p


def make_capture(pdt, p, n: int, threads: int, device=None, wav_path: str | None = None, fs: int = 0):
    """The synthetic capture, generated slice by slice by a few host threads (every sample is a pure function of its
    index) and handed on at once -- to HBM (`device`), to a WAV file (`wav_path`) or both -- so that the host never holds
    more than one 256 MB slice of it, whatever the number of ranks on the node."""
    import torch
    S = pdt.synth_lib()
    S.pdt_synth_sine_table()                                             # build the table before the threads start
    slice_n = 1 << 26
    piece = 1 << 21
    d_iq = torch.empty(2 * n, dtype=torch.int16, device=device) if device is not None else None
    f = None
    if wav_path:
        f = open(wav_path, "wb")
        hdr = C.create_string_buffer(44)
        S.pdt_synth_wav_header(hdr, fs, n)
        f.write(hdr.raw)
    buf = np.empty((min(slice_n, max(n, 1)), 2), dtype="<i2")
    if device is not None:
        try:
            tbuf = torch.from_numpy(buf.reshape(-1)).pin_memory()       # (a pinned slice: the copy runs at PCIe speed)
            buf = tbuf.numpy().reshape(-1, 2)
        except RuntimeError:
            tbuf = torch.from_numpy(buf.reshape(-1))
    with cf.ThreadPoolExecutor(max(1, threads)) as ex:
        for s0 in range(0, n, slice_n):
            c0 = min(slice_n, n - s0)

            def fill(off, s0=s0, c0=c0):
                c = min(piece, c0 - off)
                S.pdt_synth_fill(C.byref(p), s0 + off, c, buf[off:off + c].ctypes.data)

            list(ex.map(fill, range(0, c0, piece)))
            if d_iq is not None:
                d_iq[2 * s0:2 * (s0 + c0)].copy_(tbuf[:2 * c0])
                torch.cuda.synchronize()
            if f:
                f.write(buf[:c0].reshape(-1).view(np.uint8))
    if f:
        f.close()
    return d_iq


_gather_lib = None


def gather_lib():
    """libpdtgather.so: its two host-only functions state the exchange format (pdt_gather_plan / pdt_gather_unpad, include/
    pdt_gather.h); bin/demodMulti's RCCL gather and this one (torch.distributed across processes) share them."""
    global _gather_lib
    if _gather_lib is None:
        import ctypes as C
        import importlib
        pdt = importlib.import_module("project-desert-tortoise_amd")
        C.CDLL(pdt.LIBPDT_PATH, mode=C.RTLD_GLOBAL)
        L = C.CDLL(os.path.join(os.path.dirname(pdt.LIBPDT_PATH), "libpdtgather.so"))
        L.pdt_gather_plan.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]
        L.pdt_gather_unpad.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
        _gather_lib = L
    return _gather_lib


def gather_frames(frames: np.ndarray, device):
    """All ranks contribute a (ragged) array of pdt_frame records; rank 0 gets the list per rank.
    Two collectives: counts (all_gather of one int64) and the records padded to the maximum."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    rec = frames.dtype.itemsize
    n = torch.tensor([len(frames)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = np.array([int(c.item()) for c in counts], dtype=np.uint64)
    L = gather_lib()
    nmax = C.c_uint64(0)
    offsets = np.zeros(world + 1, dtype=np.uint64)
    assert L.pdt_gather_plan(counts.ctypes.data, world, C.byref(nmax), offsets.ctypes.data) == 0
    nmax = int(nmax.value)
    buf = torch.zeros(nmax * rec, dtype=torch.uint8, device=device)
    if len(frames):
        buf[: len(frames) * rec] = torch.from_numpy(frames.view(np.uint8).reshape(-1).copy()).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    if dist.get_rank() != 0:
        return None
    padded = np.ascontiguousarray(torch.stack(out).cpu().numpy())                 # world x (nmax records)
    flat = np.zeros(int(offsets[world]) * rec + 1, dtype=np.uint8)
    assert L.pdt_gather_unpad(padded.ctypes.data, counts.ctypes.data, world, nmax, rec, flat.ctypes.data) == 0
    return [flat[int(offsets[r]) * rec: int(offsets[r + 1]) * rec].view(frames.dtype) for r in range(world)]


def transmitted_check(pdt, p, frames: np.ndarray, n: int, fs: int) -> dict:
    """Size-independent property at full size: the complete decoded POES frames are frames the generator transmitted, in
    ascending order, and nearly all of the transmitted ones arrive (the frame during which the PLL locks, and a frame hit by a
    noise peak once in an hour, may be damaged: at most 0.5 % unmatched)."""
    complete = frames[frames["complete"] == 1]
    start = int(p.signal_start * 10 // fs)
    expect = int(n / fs * 10.0) - start                                  # 10 minor frames per second of signal
    sent = {bytes(pdt.synth_poes_frame(p, k)): k for k in range(start, start + expect + 2)}
    idx = [sent.get(bytes(b)) for b in complete["bytes"]]
    got = [k for k in idx if k is not None]
    ok = (len(got) >= expect - 12 and len(idx) - len(got) <= max(2, len(idx) // 200)
          and all(b > a for a, b in zip(got, got[1:])))
    return {"ok": bool(ok), "complete": int(len(complete)), "matched": len(got), "expected_about": expect}


# Algorithmic bytes per step of each kernel group: what the group must read
# and write if its input and output streams are materialised exactly once (f = 4-byte float / 8-byte double).
# The serial pieces (acquisition, head, seam repair, chain) move only a few KB: 0.
def stage_bytes(n: int, interp: int, nsym: int, nbits: int, f: int, chunk: int):
    return {
        "static_gain": 4 * chunk,                        # first chunk of int16 I/Q
        "pll_theta": (4 + f) * n,                        # I/Q in, theta out
        "pll_acquire": 0,
        "pll_phase": (f + f) * n,                        # theta in, phase out
        "pll_head": 0,
        "pll_fix": 0,
        "pll_mix": (4 + f + f) * n,                      # I/Q + phase in, mixed sample out
        "mix_fir": (4 + f + f * interp) * n,             # fused: I/Q + phase in, filtered (interpolated) stream out
        "lock_ema": 2 * f * n,
        "fir": (f + f * interp) * n,                     # f B in + f*interp B out per input sample
        "agc_block": 2 * f * interp * n,
        "agc_fix": 0,
        "gardner_table": f * interp * n,                 # the AGC stream, once
        "gardner_chain": 0,
        "gardner": f * interp * n + (f + 8) * nsym,      # the AGC stream + symbol value/index out
        "manchester": 2 * f * nsym + 5 * nbits,
        "bytesync": 2 * nbits,
    }


def group_kernel(group: str, dt: str, interp: int) -> str:
    """kernel group -> the kernel that dominates it, as rocprofv3 / tools/pmc_traffic.py name it"""
    return {
        "pll_phase": f"k_pll_phase<{dt}, false>", "pll_acquire": f"k_pll_acquire_pipe<{dt}, false, true>",
        "pll_head": f"k_pll_head<{dt}, false, true>", "pll_fix": f"k_pll_fix<{dt}, false>", "pll_theta": f"k_pll_theta<{dt}>",
        "pll_mix": f"k_pll_mix<{dt}, {'true' if dt == 'double' else 'false'}>", "lock_ema": f"k_lock_ema<{dt}>",
        "fir": f"k_fir_interp_rt<{dt}, {interp}, 26>" if dt == "float" else f"k_fir_plain<{dt}>",
        "mix_fir": "k_mix_fir<26, 0, 8, false>",
        "agc_block": f"k_agc_block<{dt}>", "gardner_table": "k_gardner_table_merge<2048>",
        "gardner": "k_gardner<float, 2048, 256>" if dt == "float" else "k_gardner_ring<double, 2560, 6, 256>",
        "static_gain": f"k_static_gain<{dt}>", "manchester": f"k_manch_emit<{dt}>", "bytesync": "k_sync_frames_tiles",
        "gardner_chain": "k_gardner_chain", "agc_fix": f"k_agc_fix<{dt}>",
    }.get(group, group)


def pmc_traffic(cfg: str, kernel: str, build: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes (FETCH_SIZE x2 + WRITE_SIZE, see
    tools/pmc_traffic.py and profiles/r3/README.md) -- only from a file measured with the library build that is running now
    (its `build` tag); otherwise None: a figure of another build says nothing about this one."""
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        path = os.path.join(ROOT, "profiles", rnd, f"pmc_hbm_traffic_bench_{cfg}.json")
        try:
            doc = json.load(open(path))
            rows = doc["kernels"] if isinstance(doc, dict) else doc
            tag = doc.get("build") if isinstance(doc, dict) else None
            if tag != build:
                continue
            for row in rows:
                if row["kernel"] == kernel:
                    return int(row["hbm_bytes"])
        except (OSError, ValueError, KeyError):
            pass
    return None


HBM_STREAM_GBS = 5800.0     # what streaming kernels reach on this chip (k_pll_theta, the AGC walkers: 5.8 - 6.0 TB/s of 8)


def sq_valu_active(cfg: str, kernel: str, build: str):
    """Fraction of the kernel's wave cycles in which its wavefronts issue vector-ALU instructions, from the committed SQ counter
    pass of this build (profiles/r4/sq_counters_bench_<cfg>.json: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES), or None."""
    try:
        for rnd in ("r6", "r5", "r4"):
            path = os.path.join(ROOT, "profiles", rnd, f"sq_counters_bench_{cfg}.json")
            if not os.path.exists(path):
                continue
            doc = json.load(open(path))
            if doc.get("build") != build:
                continue
            v = doc["kernels"][kernel]
            return v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        pass
    return None


def roofline_bound(ms: float, traffic, alg_bytes: int, valu_active=None) -> str:
    """What holds the dominant kernel up, from the evidence at hand: "hbm" when the bytes it moves (counter traffic of this build
    if committed, its algorithmic bytes otherwise) already go at 85 % of the rate streaming kernels reach here -- whatever else is
    scarce, that roof is the nearest (k_pll_phase at c3: 27.7 GB of warm-up re-reads at 5.2 TB/s, with its wavefronts issuing in
    79 % of their cycles as well); "issue" when the committed SQ pass of this build shows its wavefronts issuing vector
    instructions in more than 60 % of their cycles (lone wavefronts per SIMD: the instruction count of the serial step is the
    time); "hbm" again when the bytes take more than half its time at that rate; "latency" otherwise."""
    moved = traffic if traffic else alg_bytes
    rate = moved / (ms * 1e-3) / 1e9 if ms and moved else 0.0
    if rate >= 0.85 * HBM_STREAM_GBS:
        return "hbm"
    if valu_active is not None and valu_active >= 0.6:
        return "issue"
    return "hbm" if rate >= 0.5 * HBM_STREAM_GBS else "latency"


def cpu_exe(kind: int):
    exe = "ref_demodARGOS" if kind else "ref_demodPOES"
    ref = os.path.join(ROOT, "oracle", "_ref", exe)
    port = os.path.join(ROOT, "oracle", "oracle_demod")
    if os.path.exists(ref):
        return "reference", [ref]
    if not os.path.exists(port):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so", "oracle_demod"], check=True, capture_output=True)
    return "port", [port] + (["-a"] if kind else [])


def dump_wav(pdt, d_iq, n: int, fs: int, path: str):
    """The capture that is resident in HBM, written out as the canonical WAV the host programs read (slice by slice: the host
    never holds more than 256 MB of it)."""
    S = pdt.synth_lib()
    hdr = C.create_string_buffer(44)
    S.pdt_synth_wav_header(hdr, fs, n)
    with open(path, "wb") as f:
        f.write(hdr.raw)
        step = 1 << 27                                                    # int16 elements per slice
        for a in range(0, 2 * n, step):
            f.write(d_iq[a:min(2 * n, a + step)].cpu().numpy().view(np.uint8))


def e2e_multi(exe: str, wavs: list[str], ngpu: int, passes: int = 4, env=None, timeout: float = 1800.0) -> dict:
    """BASELINE's metric through the product launcher: `bin/demodMulti -g N -R passes -J` over N capture files -- one process,
    two contexts per GPU, every capture read from its file into its GPU's HBM, demodulated, the frame records gathered on GPU 0
    with RCCL, one text file per capture written.  A pass's clock runs from the moment the first capture may be opened until the
    last output file is closed: the gather is INSIDE it.  The first pass opens the contexts, allocates their buffers and sets
    the communicators up; the figure is the median of the others.  (POESTIPdemod/main.c:373-492 per capture; BASELINE
    configs[4].)"""
    r = subprocess.run([exe, "-g", str(ngpu), "-R", str(passes), "-J"] + list(wavs), capture_output=True, text=True, env=env, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"demodMulti"')]
    if r.returncode != 0 or not lines:
        return {"error": (r.stdout[-400:] + r.stderr[-400:]).strip(), "rc": r.returncode}
    doc = json.loads(lines[-1])["demodMulti"]
    warm = doc["passes"][1:] or doc["passes"]
    walls = sorted(p["wall_s"] for p in warm)
    med = walls[len(walls) // 2]
    mp = next(p for p in warm if p["wall_s"] == med)
    samples = sum(c["samples"] for c in doc["last_pass"])
    per_gpu = {}
    for c in doc["last_pass"]:
        g = per_gpu.setdefault(c["gpu"], {"captures": 0, "bytes": 0, "ingest_ms": 0.0, "gpu_ms": 0.0, "direct": c.get("direct", 0),
                                           "numa_node": c.get("numa_node", -1)})
        g["captures"] += 1
        g["bytes"] += c["bytes"]
        g["ingest_ms"] += c["ingest_ms"]
        g["gpu_ms"] += c["gpu_ms"]
    for g in per_gpu.values():
        g["ingest_GBps"] = round(g["bytes"] / (g["ingest_ms"] * 1e-3) / 1e9, 2) if g["ingest_ms"] > 0 else None
        g["ingest_ms"] = round(g["ingest_ms"], 2)
        g["gpu_ms"] = round(g["gpu_ms"], 2)
    return {"value": round(samples / med / 1e6, 3), "unit": "Msamples/s", "ms": round(med * 1e3, 3), "gpus": doc["gpus"],
            "contexts_per_gpu": doc["lanes"], "captures": doc["captures"], "samples": samples,
            "passes_ms": [round(p["wall_s"] * 1e3, 3) for p in doc["passes"]], "statistic": f"median of passes 2..{len(doc['passes'])} (pass 1 opens the contexts and the communicators)",
            "until_last_gpu_ms": round(mp["until_last_gpu_s"] * 1e3, 3), "gather_ms": mp["gather_ms"], "write_ms": mp["write_ms"],
            "per_gpu": {str(k): per_gpu[k] for k in sorted(per_gpu)},
            "ingest_GBps_all": round(sum(g["bytes"] for g in per_gpu.values()) / max(g["ingest_ms"] for g in per_gpu.values()) / 1e6, 2) if per_gpu else None,
            "includes": "per capture: open, 44-byte header, threaded read into pinned staging, copies to HBM, the chain (large POES files in "
                        "overlapped segments), frame records to the host; then ONE RCCL gather of all GPUs' records on GPU 0 (counts + padded "
                        "all-gather) and one text file per capture written and closed; contexts already open (passes 2..)"}


def cpu_cores(k: int):
    """One core per CPU-baseline process (pinned): the last k cores this process may run on, each process
    bound to its own with sched_setaffinity before exec; None where the platform cannot say."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return [None] * k
    return [avail[-1 - (i % len(avail))] for i in range(k)]


def cpu_baseline(kind: int, wavs: list[str], n_sample: int, tmp: str):
    """The reference CPU path on this host, one thread per process: len(wavs) concurrent processes, one capture each.
    Returns (per-process DSP seconds, wall seconds of the slowest, texts, kind)."""
    what, cmd = cpu_exe(kind)
    outs = [os.path.join(tmp, f"cpu_out_{i}.txt") for i in range(len(wavs))]
    cores = cpu_cores(len(wavs))
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + [w, o], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              preexec_fn=(lambda c=c: os.sched_setaffinity(0, {c})) if c is not None else None)
             for w, o, c in zip(wavs, outs, cores)]
    dsp, walls = [], []
    for p in procs:
        _, err = p.communicate()
        walls.append(time.perf_counter() - t0)
        if p.returncode != 0:
            raise RuntimeError(f"CPU baseline failed: {err[-500:]}")
        m = re.search(r"dsp_seconds ([0-9.]+)", err)
        dsp.append(float(m.group(1)) if m else walls[-1])
    texts = [open(o, "rb").read() if os.path.exists(o) else b"" for o in outs]
    cpu_baseline.last_cores = cores
    return dsp, max(walls), texts, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # (5 steps behind 1 of warm-up read 2 % slow: the clocks are still settling)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=DEFAULT_CONFIG,
                    help=f"workload (default {DEFAULT_CONFIG} = BASELINE configs[2], the same for every N)")
    ap.add_argument("--seconds", type=float, default=None, help="capture length override (parity / smoke runs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline, e2e and CLI legs")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the c2 / argos records measured beside the headline")
    ap.add_argument("--e2e-only", action="store_true", help="developer runs: keep the in-process e2e leg, skip the CLI and CPU legs")
    ap.add_argument("--captures", type=int, default=1,
                    help="captures demodulated together per GPU and step through pdt_demod_batch_device (default 1 = the "
                         "BASELINE workload; >1 is the batched many-capture mode, reported as such)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)                                # does not return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU implementation of the product path)")
    # PDT_BENCH_BACKEND=gloo: dry run of the multi-rank logic on a box with fewer GPUs than ranks (ranks share
    # devices, collectives on host tensors); the real runs use RCCL, one rank per GPU
    backend = os.environ.get("PDT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    elif world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")              # where collective payloads live
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cfg = args.config
    c = CONFIGS[cfg]
    kind, fs, seconds, cpu_seconds, baseline_cfg = c["kind"], c["fs"], c["seconds"], c["cpu_seconds"], c["baseline"]
    if args.seconds:
        seconds = args.seconds
        cpu_seconds = min(cpu_seconds, seconds)
    mode = 1 if kind else 0
    dt_name, fbytes = ("double", 8) if kind else ("float", 4)
    legs = (not args.no_cpu) and world == 1 and args.captures == 1       # e2e / CLI / CPU legs belong to the N = 1 line

    pdt = importlib.import_module("project-desert-tortoise_amd")
    build_tag = pdt.build_tag()
    n = int(round(seconds * fs))
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(dir=shm, prefix="pdt_bench_") if ((legs or (args.captures > 1 and not args.no_cpu and world == 1)) and rank == 0) else None
    try:
        run(args, pdt, torch, dist, rank, local, world, dev, cdev, cfg, kind, fs, seconds, cpu_seconds, baseline_cfg, mode, dt_name,
            fbytes, legs, build_tag, n, threads, tmp)
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)


def run(args, pdt, torch, dist, rank, local, world, dev, cdev, cfg, kind, fs, seconds, cpu_seconds, baseline_cfg, mode, dt_name,
        fbytes, legs, build_tag, n, threads, tmp):
    ncap = max(args.captures, 1)
    par = capture_params(pdt, cfg, 1234 + rank * ncap, seconds)            # one independent capture per rank (and per slot of a batch)
    wav = os.path.join(tmp, f"{cfg}.wav") if (tmp and legs) else None
    d_iq = make_capture(pdt, par, n, threads, device=dev, wav_path=wav, fs=fs)    # resident in HBM before timing
    dm = pdt.Demodulator(mode, fs, device=local, profile=True).keep_pll(False)      # as the host programs run it
    dm.set_stream(torch.cuda.current_stream().cuda_stream)
    extra = [pdt.Demodulator(mode, fs, device=local, profile=True).keep_pll(False) for _ in range(ncap - 1)]
    # batched many-capture mode: every slot its own capture (seeds 1234 + rank * captures + slot)
    pars_b = [par] + [capture_params(pdt, cfg, 1234 + rank * ncap + i, seconds) for i in range(1, ncap)]
    d_iqs = [d_iq] + [make_capture(pdt, pars_b[i], n, threads, device=dev) for i in range(1, ncap)]

    def step():
        if extra:
            pdt.demod_batch([dm] + extra, [d.data_ptr() for d in d_iqs], [n] * ncap)
        else:
            dm.demod_device(d_iq.data_ptr(), n)

    for _ in range(args.warmup):
        step()
    ktot: dict[str, float] = {}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, (_, ms) in dm.kernel_times().items():
            ktot[k] = ktot.get(k, 0.0) + ms
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    own_ms = dt / args.steps * 1e3
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        each = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(each, torch.tensor([own_ms], dtype=torch.float64, device=cdev))
        per_rank_ms = [round(float(e.item()), 3) for e in each]
    else:
        per_rank_ms = [round(own_ms, 3)]

    st = dm.stats()
    frames = dm.frames_array()
    gpu_text = dm.text()
    gathered = gather_frames(frames, cdev) if world > 1 else [frames]
    # ---- BASELINE's metric at N GPUs: N capture files through the product launcher (bin/demodMulti), gather inside the clock
    multi = None
    multi_same = None
    exe_multi = os.path.join(ROOT, "bin", "demodMulti")
    if not args.no_cpu and ncap == 1 and kind == 0 and os.path.exists(exe_multi) and not args.e2e_only:
        mdir = None
        if rank == 0:
            mdir = tmp if tmp else tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, prefix="pdt_bench_m_")
        if world > 1:
            box = [mdir]
            dist.broadcast_object_list(box, src=0)
            mdir = box[0]
        my_wav = wav if wav else os.path.join(mdir, f"rank{rank}.wav")
        if not wav:
            dump_wav(pdt, d_iq, n, fs, my_wav)
        if world > 1:
            dist.barrier()
        if rank == 0:
            wavs_m = [my_wav] + [os.path.join(mdir, f"rank{r}.wav") for r in range(1, world)]
            try:
                multi = e2e_multi(exe_multi, wavs_m, world)
            except Exception as e:                                         # (never fatal for the resident figure)
                multi = {"error": str(e)[:300]}
            open(os.path.join(mdir, "launcher_done"), "w").close()
        elif world > 1:
            # (the other ranks wait on the HOST for the launcher: inside an RCCL barrier they would each keep a few CUs of their GPU
            # spinning beside the launcher's kernels for as long as it runs)
            t_wait = time.perf_counter()
            while not os.path.exists(os.path.join(mdir, "launcher_done")) and time.perf_counter() - t_wait < 1900.0:
                time.sleep(0.05)
        if world > 1:
            dist.barrier()
        out_m = my_wav + ".frames.txt"
        mine = bool(os.path.exists(out_m) and open(out_m, "rb").read() == gpu_text)
        if os.path.exists(out_m):
            os.unlink(out_m)
        if world > 1:
            flags = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
            dist.all_gather(flags, torch.tensor([int(mine)], dtype=torch.int64, device=cdev))
            multi_same = [bool(int(f.item())) for f in flags]
            if not wav and os.path.exists(my_wav):
                os.unlink(my_wav)
            dist.barrier()
            if rank == 0 and not tmp:
                shutil.rmtree(mdir, ignore_errors=True)
        else:
            multi_same = [mine]
    # batched many-capture mode: every slot's text against the reference CPU path on that slot's capture (a few processes at a time)
    batch_same = None
    if ncap > 1 and tmp and not legs:
        slot_texts = [d.text() for d in [dm] + extra]
        group = max(1, min(8, (os.cpu_count() or 8) // 2))
        batch_same = []
        for g0 in range(0, ncap, group):
            wavs_b = []
            for i in range(g0, min(ncap, g0 + group)):
                wavs_b.append(os.path.join(tmp, f"slot_{i}.wav"))
                make_capture(pdt, pars_b[i], n, threads, wav_path=wavs_b[-1], fs=fs)
            _, _, texts_b, _ = cpu_baseline(kind, wavs_b, n, tmp)
            batch_same += [bool(t == u) for t, u in zip(texts_b, slot_texts[g0:g0 + group])]
            for w in wavs_b:
                os.unlink(w)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * ncap * n * args.steps / dt / 1e6
        chunk = 2400 if kind else 10000
        sb = stage_bytes(n * ncap, st.interp, st.symbols * ncap, st.bits * ncap, fbytes, chunk)
        stages = {}
        for k, ms in ktot.items():
            per = ms / args.steps
            gbs = (sb.get(k, 0) / (per * 1e-3) / 1e9) if per > 0 else 0.0
            stages[k] = {"ms": round(per, 4), "alg_bytes": sb.get(k, 0), "GBps": round(gbs, 2),
                         "frac_hbm": round(gbs / HBM_PEAK_GBS, 6)}
        dom = max(stages, key=lambda k: stages[k]["ms"])
        dom_kernel = group_kernel(dom, dt_name, st.interp)
        # FIR+PLL stage (north-star target): critical path through the two concurrent streams
        g = lambda k: stages.get(k, {"ms": 0.0})["ms"]
        front_ms = (g("pll_theta") + max(g("pll_phase"), g("pll_acquire") + g("pll_head")) + g("pll_fix") + g("pll_mix")
                    + g("lock_ema") + g("fir") + g("mix_fir"))
        front_bytes = (4 + fbytes * st.interp) * n * ncap   # fused FIR+PLL stage: 4 B in + f*interp B out per sample
        hbm_bound = ["pll_theta", "pll_mix", "fir", "mix_fir"]   # the groups that are pure streaming kernels
        traffic = pmc_traffic(cfg, dom_kernel, build_tag) if ncap == 1 else None
        parity = {}
        if batch_same is not None:
            parity["batch_slots_text_equals_cpu_baseline"] = batch_same
        if multi is not None and "error" not in multi:
            parity["demodMulti_text_equals_resident_full_size"] = multi_same
        out = {
            "metric": "IQ Msamples/s end-to-end (WAV\u2192minorframes), 1-GPU + %HBM roofline",
            "value_is": f"the bench contract's `value`: the whole hot path over a {cfg} capture ALREADY RESIDENT IN HBM when the timed "
                        "region starts (capture -> minor-frame records, K timed steps); the PCIe-inclusive figure the metric's name "
                        "describes -- WAV file on tmpfs -> minorframes file closed -- is `value_e2e` / `ms_e2e` in this same line "
                        "(N = 1), and the C host program from process start is `e2e_cli`",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if kind else "f32",
            "data": "synthetic",
            "build": build_tag,
            "config": {"workload": f"{cfg} = BASELINE {baseline_cfg}: synthetic {fs / 1000:g} ksps complex-IQ capture, {seconds:g} s "
                                   f"({n} samples) per GPU, {'ARGOS' if kind else 'POES'} chain, chunk {chunk}, input resident in HBM "
                                   "when the timed region starts",
                       "samples_per_gpu": n * ncap, "captures": world * ncap,
                       "parallelism": f"{ncap} capture(s) per GPU x{world}"
                                      + (" (batched many-capture mode: one launch per stage for all captures, every slot its own capture)" if ncap > 1 else "")},
            "roofline": {"bound": roofline_bound(stages[dom]["ms"], traffic, stages[dom]["alg_bytes"], sq_valu_active(cfg, dom_kernel, build_tag) if ncap == 1 else None),
                         "valu_active": sq_valu_active(cfg, dom_kernel, build_tag) if ncap == 1 else None, "kernel": dom_kernel, "group": dom,
                         "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": stages[dom]["frac_hbm"], "traffic": traffic,
                         "alg_bytes": stages[dom]["alg_bytes"], "ms": stages[dom]["ms"],
                         "note": "achieved = algorithmic bytes / live HIP-event duration of the group's launch; traffic = FETCH_SIZE x2 + "
                                 "WRITE_SIZE per launch from the committed counter passes of this build (profiles/), null when the "
                                 "committed passes belong to another build"},
            "streaming_kernels": {k: {"GBps": stages[k]["GBps"], "frac_hbm": stages[k]["frac_hbm"],
                                      "traffic": pmc_traffic(cfg, group_kernel(k, dt_name, st.interp), build_tag) if ncap == 1 else None}
                                  for k in hbm_bound if k in stages},
            "pipeline_hbm_frac": round(4 * n * ncap / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
            "fir_pll_stage": {"ms": round(front_ms, 4), "alg_bytes": front_bytes,
                              "GBps": round(front_bytes / (front_ms * 1e-3) / 1e9, 2) if front_ms else None,
                              "frac_hbm": round(front_bytes / (front_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if front_ms else None,
                              "target_fir_pll": 0.40,
                              "why_not": "issue/latency-bound, not HBM-bound: the (phase, frequency) recurrence is an exact serial float loop "
                                         "walked one lane per block (16 vector operations a step, W + B steps per lane) and the one-lane head "
                                         "behind the lock is mandatory (3.5 ms at c3); DESIGN 5.2",
                              "note": "critical path: theta + max(phase, acquire + head) + fix + mix + fir"},
            "stages": stages,
            "per_rank_ms": per_rank_ms,
            "frames_per_capture": [int(len(g)) for g in gathered],
            "pll_seam_fixes": int(st.pll_seam_fixes), "agc_seam_fixes": int(st.agc_seam_fixes),
            "gardner_walked": int(st.gardner_walked), "gardner_candidates": int(st.gardner_candidates),
            "lock_sample": int(st.lock_sample),
        }
        if multi is not None:
            # the figure BASELINE.json's metric names, at this N: files -> frame files through bin/demodMulti, RCCL gather inside
            out["e2e_multi"] = multi
            if "error" not in multi:
                out["metric_e2e"] = (f"IQ Msamples/s end-to-end (WAV files on tmpfs -> minorframes files closed), {world} GPU(s): {world} capture "
                                     "file(s) through bin/demodMulti, one per GPU, two contexts per GPU, the RCCL gather of the frame records "
                                     "inside the clock, contexts open; median of passes 2..4")
                out["value_e2e"] = multi["value"]
                out["ms_e2e"] = multi["ms"]
        # ---- full size, every rank: the decoded frames are the transmitted ones (POES; a size-independent property)
        # (a weak signal loses frames, and at 18.75 ksps -- 1.13 samples per Manchester symbol -- the reference itself decodes two
        # frames in three with bit errors once the symbol clock has drifted against the sampling grid: the CPU sample is their gate)
        if kind == 0 and CONFIGS[cfg]["noise_x"] == 1.0 and not CONFIGS[cfg]["env_floor"] and fs >= 50000:
            tx = [transmitted_check(pdt, capture_params(pdt, cfg, 1234 + r, seconds), gathered[r], n, fs) for r in range(world)]
            parity["frames_equal_transmitted_full_size"] = [t["ok"] for t in tx]
            parity["frames_complete_matched_expected"] = [[t["complete"], t["matched"], t["expected_about"]] for t in tx]
            if world > 1:
                parity["gathered_rank0_equals_own"] = bool(pdt.format_frames(gathered[0]) == gpu_text)
        if legs:
            # ---- end to end, in process: what POESTIPdemod/main.c:284-512 does with the file
            e2e_ms, split = [], []
            with pdt.Demodulator(mode, fs, device=local, profile=bool(os.environ.get("PDT_DEBUG_OVERLAP"))).keep_pll(False) as de:
                for rep in range(6):                                      # (rep 0 allocates the context's buffers: reported apart)
                    outp = os.path.join(tmp, f"e2e_out{rep}.txt")         # (a new file every time, as the host program's)
                    t1 = time.perf_counter()
                    fd = os.open(wav, os.O_RDONLY)
                    hdr = os.pread(fd, 44, 0)
                    rate = int.from_bytes(hdr[24:28], "little")
                    nfr = (os.fstat(fd).st_size - 44) // 4
                    fo = os.open(outp, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
                    t2 = time.perf_counter()
                    de.demod_file_text(fd, 44, nfr, fo, 0)               # pdt_demod_file: capture file in, frame text out
                    t3 = time.perf_counter()
                    os.close(fd)
                    os.close(fo)
                    t5 = time.perf_counter()
                    e2e_ms.append((t5 - t1) * 1e3)
                    split.append({"open_header": round((t2 - t1) * 1e3, 3), "demod_file": round((t3 - t2) * 1e3, 3),
                                  "close": round((t5 - t3) * 1e3, 3)})
                    assert rate == fs and nfr == n
                    if rep < 5:
                        os.unlink(outp)
                e2e_text = open(outp, "rb").read()                       # (the parity legs compare the FILE's bytes)
                assert e2e_text == de.text()
                e2e_gpu_ms = de.stats().gpu_ms
            first_ms, e2e_ms, split = e2e_ms[0], e2e_ms[1:], split[1:]
            med = sorted(e2e_ms)[len(e2e_ms) // 2]
            out["e2e"] = {"ms": round(med, 3), "value": round(n / med / 1e3, 3), "unit": "Msamples/s", "statistic": "median of 5 (one untimed run before them)",
                          "runs_ms": [round(x, 3) for x in e2e_ms], "first_run_ms": round(first_ms, 3),
                          "gpu_ms_last_segment": round(e2e_gpu_ms, 3), "file_bytes": 44 + 4 * n, "split_ms": split[e2e_ms.index(med)],
                          "includes": "open WAV on tmpfs, header, output file created, pdt_demod_file (threaded pread into pinned memory + "
                                      "copies to HBM; the chain in three unequal segments -- 55 / 28 / 17 % -- each as soon as its samples "
                                      "have arrived; frame records to the host, time stamps, each segment's text formatted and written "
                                      "while the next one runs), files closed; context already open (HIP initialised)"}
            # the figure BASELINE.json's metric names, beside `value` (which is the resident rate)
            # (`value_e2e` is the launcher's figure -- the same definition at every N, `e2e_multi` -- whenever that leg ran; this one is
            # the library call in process: pdt_demod_file, text written while the segments run)
            if "value_e2e" not in out:
                out["metric_e2e"] = ("IQ Msamples/s end-to-end (WAV file on tmpfs -> minorframes file closed), 1 GPU, in process, HIP "
                                     "already initialised, median of 5; `e2e_cli` is the C host program from process start")
                out["value_e2e"] = out["e2e"]["value"]
                out["ms_e2e"] = out["e2e"]["ms"]
            out["value_e2e_in_process"] = out["e2e"]["value"]
            parity["e2e_text_equals_resident_full_size"] = bool(e2e_text == gpu_text)
            if args.e2e_only:
                out["parity"] = parity
                print(json.dumps(out), flush=True)
                return
            # ---- the C host program itself
            exe = os.path.join(ROOT, "bin", "demodARGOS" if kind else "demodPOES")
            if os.path.exists(exe):
                cli_out = os.path.join(tmp, "cli_out.txt")
                # (this process has just closed a context with ~25 GB of buffers: the driver hands that memory back in the background,
                # and a process that starts meanwhile waits for it in its runtime start-up and its first allocations -- up to 0.7 s)
                torch.cuda.synchronize()
                time.sleep(3.0)
                t1 = time.perf_counter()
                r = subprocess.run([exe, "-T", "-d", str(local), "-o", cli_out, wav], capture_output=True, text=True)
                cli_s = time.perf_counter() - t1
                cli_text = open(cli_out, "rb").read() if os.path.exists(cli_out) else b""
                split = None
                for line in r.stderr.splitlines():
                    if line.startswith('{"timing_ms"'):
                        split = json.loads(line)["timing_ms"]
                        split["process_start_and_loader"] = round(cli_s * 1e3 - split["total"], 2)
                out["e2e_cli"] = {"seconds": round(cli_s, 3), "value": round(n / cli_s / 1e6, 3), "unit": "Msamples/s", "rc": r.returncode,
                                  "split_ms": split,
                                  "includes": "process start, HIP initialisation, everything of `e2e`, the reference's per-chunk "
                                              "progress / quality lines (averagePhase EMA); split_ms from the program's own clock (-T): "
                                              "`demod_call` contains `of_it_alloc` (device + pinned buffers of a first run) and "
                                              "`of_it_ingest`"}
                # ... and without the per-chunk reports (-P): the overlapped path of `e2e`, from process start
                os.unlink(cli_out)
                time.sleep(3.0)                                      # (the first program's buffers, likewise)
                t1 = time.perf_counter()
                r2 = subprocess.run([exe, "-T", "-P", "-d", str(local), "-o", cli_out, wav], capture_output=True, text=True)
                cli2_s = time.perf_counter() - t1
                split2 = None
                for line in r2.stderr.splitlines():
                    if line.startswith('{"timing_ms"'):
                        split2 = json.loads(line)["timing_ms"]
                        split2["process_start_and_loader"] = round(cli2_s * 1e3 - split2["total"], 2)
                out["e2e_cli_noprogress"] = {"seconds": round(cli2_s, 3), "rc": r2.returncode, "split_ms": split2,
                                             "text_equals_resident": bool(os.path.exists(cli_out) and open(cli_out, "rb").read() == gpu_text)}
                parity["cli_text_equals_resident_full_size"] = bool(cli_text == gpu_text)
            os.unlink(wav)                                               # (3.6 GB of tmpfs back before the CPU legs)
            # ---- CPU baseline on a bounded sample: the first n_cpu samples of this capture, one thread ...
            n_cpu = min(n, int(round(cpu_seconds * fs)))
            cpu_wav = os.path.join(tmp, "sample_0.wav")
            make_capture(pdt, par, n_cpu, threads, wav_path=cpu_wav, fs=fs)
            dsp, wall, texts, what = cpu_baseline(kind, [cpu_wav], n_cpu, tmp)
            out["cpu_baseline"] = {"value": round(n_cpu / dsp[0] / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": what,
                                   "e2e_value": round(n_cpu / wall / 1e6, 3),
                                   "sample": f"the first {n_cpu} samples ({n_cpu / fs:g} s) of rank 0's capture as a WAV on tmpfs: "
                                             f"{dsp[0]:.2f} s in the DSP stages (value), {wall:.2f} s wall for the whole program with "
                                             "file read and text output (e2e_value)",
                                   "pinned_to_core": cpu_baseline.last_cores[0], "host_cpus": os.cpu_count()}
            with pdt.Demodulator(mode, fs, device=local) as ds:
                ds.demod_device(d_iq.data_ptr(), n_cpu)                 # the same samples, through the product path
                parity["sample_text_equals_cpu_baseline"] = bool(ds.text() == texts[0])
                parity["sample"] = f"first {n_cpu} samples, {len(texts[0])} bytes of text"
                # ---- ... and 8 concurrent single-thread processes, one capture each (configs[4]'s CPU counterpart)
                if kind == 0 and cfg == DEFAULT_CONFIG and not args.seconds:
                    wavs, pars = [cpu_wav], [par]
                    for r in range(1, 8):
                        pr = capture_params(pdt, cfg, 1234 + r)
                        w = os.path.join(tmp, f"sample_{r}.wav")
                        make_capture(pdt, pr, n_cpu, threads, wav_path=w, fs=fs)
                        wavs.append(w); pars.append(pr)
                    dsp8, wall8, texts8, _ = cpu_baseline(kind, wavs, n_cpu, tmp)
                    out["cpu_baseline_8proc"] = {"value": round(8 * n_cpu / wall8 / 1e6, 3), "unit": "Msamples/s", "cores": 8, "kind": what,
                                                 "per_process_dsp_s": [round(x, 2) for x in dsp8], "wall_s": round(wall8, 2),
                                                 "pinned_to_cores": cpu_baseline.last_cores,
                                                 "sample": f"8 concurrent single-thread processes, the first {n_cpu} samples of the 8 "
                                                           "captures of configs[4] (seeds 1234..1241), whole program (file on tmpfs -> text)"}
                    same8 = []
                    for r in range(1, 8):
                        d_r = make_capture(pdt, pars[r], n_cpu, threads, device=dev)
                        ds.demod_device(d_r.data_ptr(), n_cpu)
                        same8.append(bool(ds.text() == texts8[r]))
                        del d_r
                    parity["other_seeds_sample_text_equals_cpu"] = same8
        out["parity"] = parity
        out["parity_with_cpu_baseline"] = parity.get("sample_text_equals_cpu_baseline")
        if legs and not args.no_secondary and cfg == DEFAULT_CONFIG and not args.seconds:
            del d_iq
            torch.cuda.empty_cache()
            out["secondary"] = {}
            # the other workloads on the same GPU, each a process of its own: BASELINE configs[1] and [3] with their CPU legs; the pass-shaped
            # capture, the interpolating filter at scale and the batched ARGOS mode resident only
            for name, extra in (("c2", ["--config", "c2", "--steps", "10", "--warmup", "2"]),
                                ("argos", ["--config", "argos", "--steps", "10", "--warmup", "2"]),
                                ("pass", ["--config", "pass", "--steps", "2", "--warmup", "1", "--no-cpu"]),
                                ("i8", ["--config", "i8", "--steps", "5", "--warmup", "2", "--no-cpu"]),
                                ("argos_x64", ["--config", "argos", "--captures", "64", "--steps", "3", "--warmup", "1", "--no-cpu"])):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-secondary"] + extra, capture_output=True, text=True, timeout=900)
                    ref = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                    out["secondary"][name] = {k: ref.get(k) for k in ("value", "unit", "ms_per_step", "dtype", "roofline", "fir_pll_stage",
                                                                       "e2e", "e2e_cli", "value_e2e", "ms_e2e", "cpu_baseline", "parity")}
                    out["secondary"][name]["workload"] = ref["config"]["workload"]
                    out["secondary"][name]["stages_ms"] = {k: v["ms"] for k, v in ref.get("stages", {}).items()}
                except Exception as e:                                     # (never fatal for the headline line)
                    out["secondary"][name] = {"error": (str(e) + " " + (r.stderr[-300:] if "r" in dir() else ""))[:500]}
        bad = [k for k, v in parity.items() if v is False or (isinstance(v, list) and v and isinstance(v[0], bool) and not all(v))]
        if bad:
            sys.stderr.write("bench.py: parity failure -- no result line: " + json.dumps({k: parity[k] for k in bad}) + "\n")
            sys.stderr.write(json.dumps({k: out[k] for k in ("parity", "e2e", "e2e_cli") if k in out}) + "\n")
            if world > 1:
                dist.destroy_process_group()
            sys.exit(1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
