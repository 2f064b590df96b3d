#!/usr/bin/env python3
"""Known-answer vector for the byte synchroniser (the reference's only unit-test-like artefact).

POESTIPdemod/ByteSync.c:6-14 keeps a commented-out harness whose input is a literal string of demodulated bits
(about 19 kbit, several minor frames).  This script lifts that DATA (the bit string, nothing else) into
bytesync_harness_bits.txt and runs the reference's own ByteSync object over it (oracle/_ref/ref_demodPOES -B) in
pieces of 4160, 1000 and 77 bits -- the answer must not depend on the piece size -- to produce
bytesync_harness_frames.txt.  Runs only in the build container (needs /root/reference); the outputs are committed."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = "/root/reference/POESTIPdemod/ByteSync.c"
REF = os.path.join(ROOT, "oracle/_ref/ref_demodPOES")


def main():
    text = open(SRC).read()
    strings = re.findall(r'dataStreamBits\[\]\s*=\s*"([01]+)"', text)
    bits = max(strings, key=len)                      # the first, un-commented-in-the-comment literal is the long one
    assert len(bits) > 15000
    bpath = os.path.join(HERE, "bytesync_harness_bits.txt")
    with open(bpath, "w") as f:
        f.write(bits + "\n")
    outs = []
    for piece in (4160, 1000, 77):
        out = os.path.join(HERE, "bytesync_harness_frames.txt")
        subprocess.run([REF, "-B", "-c", str(piece), bpath, out], check=True, capture_output=True)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 0
    print(len(bits), "bits ->", outs[0].count(b"\n"), "lines")


if __name__ == "__main__":
    sys.exit(main())
