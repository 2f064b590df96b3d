#!/usr/bin/env python3
"""Golden vectors for the chunk loop's progress / quality line (POESTIPdemod/main.c:457-481, ARGOSdemod/main.c:286-296):
the text the reference prints per chunk and CarrierTrackPLL's return value of every pass of the loop, produced by the
reference's own DSP objects through oracle/_ref (ref_driver.c -d: <dump>.progress, <dump>.avg).  DATA only.  Runs where
/root/reference exists; the GPU box uses the committed files.

    python tests/golden/make_progress_golden.py
"""
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
REF_POES = os.path.join(ROOT, "oracle/_ref/ref_demodPOES")
REF_ARGOS = os.path.join(ROOT, "oracle/_ref/ref_demodARGOS")


def run(binary, wav, name, extra, tmp, ext):
    dump = os.path.join(tmp, name)
    subprocess.run([binary, *extra, "-d", dump, wav, os.path.join(tmp, name + ".txt")], check=True, capture_output=True)
    shutil.copy(dump + ".progress", os.path.join(HERE, name + ".progress"))
    shutil.copy(dump + ".avg", os.path.join(HERE, name + ".avg." + ext))


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    with tempfile.TemporaryDirectory() as tmp:
        clip = os.path.join(HERE, "5sec_clip.wav")
        run(REF_POES, clip, "clip.c10000", ["-c", "10000"], tmp, "f32")
        run(REF_POES, clip, "clip.c1000", ["-c", "1000"], tmp, "f32")
        # 3 s at 50 ksps = 15 chunks exactly: the loop's extra pass with zero samples prints the last line
        iq = pdt.synth_capture(0, 50000, 3.0, seed=1234)
        wav = os.path.join(tmp, "poes_50000.wav")
        pdt.write_wav(wav, 50000, iq)
        run(REF_POES, wav, "poes_50000", [], tmp, "f32")
        iq = pdt.synth_capture(1, 32000, 13.0, seed=99)
        wav = os.path.join(tmp, "argos.wav")
        pdt.write_wav(wav, 32000, iq)
        run(REF_ARGOS, wav, "argos_32000", [], tmp, "f64")


if __name__ == "__main__":
    main()
