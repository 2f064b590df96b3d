"""How long a warm-up do the block-parallel PLL / AGC need?  Seam repairs vs warm-up length."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
for name, fs, iq in (("synth50k_120s", 50000, pdt.synth_capture(0, 50000, 120.0, seed=1234)),
                     ("synth250k_24s", 250000, pdt.synth_capture(0, 250000, 24.0, seed=7)),
                     ("clip", 50000, pdt.read_wav(os.path.join(ROOT, "tests/golden/5sec_clip.wav"))[1])):
    interp = round(150000 / fs)
    print("==", name)
    for wp in (0.04, 0.08, 0.12, 0.16, 0.2, 0.3):
        d = pdt.Demodulator(pdt.MODE_POES, fs, profile=True, pll_block=int(0.1 * fs), pll_warm=int(wp * fs))
        d.demod(iq); s = d.stats(); kt = d.kernel_times()
        print(f"  pll warm {wp:.2f}s block 0.1s: blocks {s.pll_blocks} fixes {s.pll_seam_fixes}  phase {kt['pll_phase'][1]:.2f} ms fix {kt['pll_fix'][1]:.2f} ms")
        d.close()
    for wa in (0.15, 0.25, 0.35, 0.5, 0.7, 1.0):
        d = pdt.Demodulator(pdt.MODE_POES, fs, profile=True, agc_block=int(0.125 * fs * interp), agc_warm=int(wa * fs * interp))
        d.demod(iq); s = d.stats(); kt = d.kernel_times()
        print(f"  agc warm {wa:.2f}s block 0.125s: blocks {s.agc_blocks} fixes {s.agc_seam_fixes}  agc {kt['agc_block'][1]:.2f} ms fix {kt['agc_fix'][1]:.2f} ms")
        d.close()
