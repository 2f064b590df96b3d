"""Large 250 ksps capture (BASELINE configs[2] scale-down: PDT_SECS seconds) on the GPU against the reference's own
CPU objects (oracle/_ref/ref_demodPOES) -- byte-identical minor-frame file; run on a GPU box."""
import importlib, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
rate = int(os.environ.get("PDT_RATE", "250000"))
secs = float(os.environ.get("PDT_SECS", "600"))
t0 = time.time()
iq = pdt.synth_capture(0, rate, secs, seed=31)
print(f"synth {len(iq)} samples in {time.time() - t0:.1f} s", flush=True)
d = pdt.Demodulator(pdt.MODE_POES, rate, profile=True)
d.demod(iq); t1 = time.time(); d.demod(iq); t2 = time.time()
s = d.stats(); kt = d.kernel_times()
print(f"gpu: {s.frames} frames, gpu_ms {s.gpu_ms:.2f} ({len(iq) / s.gpu_ms / 1e3:.0f} Msamples/s), wall incl. H2D {t2 - t1:.3f} s, pll fixes {s.pll_seam_fixes}, "
      f"agc fixes {s.agc_seam_fixes}, walked {s.gardner_walked}, parallel {s.gardner_parallel}, cand {s.gardner_candidates}")
print("  " + " ".join(f"{k} {v[1]:.2f}" for k, v in kt.items()), flush=True)
ref = os.path.join(ROOT, "oracle", "_ref", "ref_demodPOES")
if os.path.exists(ref) and not os.environ.get("PDT_NO_REF"):
    with tempfile.TemporaryDirectory() as tmp:
        wav = os.path.join(tmp, "c3.wav"); out = os.path.join(tmp, "o.txt")
        pdt.write_wav(wav, rate, iq)
        t0 = time.time(); subprocess.run([ref, wav, out], check=True, capture_output=True); dt = time.time() - t0
        text = open(out, "rb").read()
    same = text == d.text()
    print(f"reference CPU: {dt:.1f} s ({len(iq) / dt / 1e6:.2f} Msamples/s); output identical: {same} ({len(text)} bytes)")
    sys.exit(0 if same else 1)
