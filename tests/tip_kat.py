"""Hand-derived known-answer vectors for the frame validation.  There is no MATLAB / Octave in this image,
so the expected values below were worked out BY HAND from the reference's MATLAB source, line by line (MATLAB is 1-indexed:
its minorFrames(f, w) is bytes[w - 1]); neither the oracle nor the GPU code was used to produce them.

checkParity.m:20-86   parity(f, g) = number of one bits in words 3..19 / 20..36 / 37..53 / 54..70 / 71..87 (bytes 2..18,
                      19..35, 36..52, 53..69, 70..86); the check passes when mod(count, 2) equals bit 5 / 4 / 3 / 2 / 1
                      (bitshift(byte104, -5) & 1, ...) of word 104 (bytes[103]).
daytimeDecode.m:4     minorFrameID = ((bytes[4] & 1) << 8) | bytes[5]
daytimeDecode.m:16    spaceCraft = bytes[2]
daytimeDecode.m:19    dayNum = (bytes[8] << 1) + ((bytes[9] | 128) >> 7)          (the "| 128" makes the second term always 1)
daytimeDecode.m:22-29 ms = ((bytes[9] & 7) << 24) + (bytes[10] << 16) + (bytes[11] << 8) + bytes[12], kept when < 86 400 000,
                      else -1; T0 = ms - frameTime * 1000

Each vector: (name, frame time, 104 bytes, expected dict).  `parity` is this repository's bit mask: bit g set = check g
(g = 0..4 for the five groups in the order above) failed."""


def _frame(**at):
    b = bytearray(104)
    b[0], b[1] = 0xED, 0xE2
    for k, v in at.items():
        b[int(k[1:])] = v
    return bytes(b)


VECTORS = [
    # 1. all payload bytes zero: every group holds 0 ones (even), every parity bit of bytes[103] is 0 -> all five checks pass.
    #    minorFrameID = 0 -> a major-frame start: day = (0 << 1) + ((0 | 128) >> 7) = 1, ms = 0 (< 86 400 000).
    ("zeros", 3.25, _frame(), dict(parity=0, minor_id=0, spacecraft=0, has_time=1, day=1, day_ms=0)),
    # 2. bytes 2..18 all 0xFF: 17 * 8 = 136 ones, even; bit 5 of bytes[103] is 0 -> check 0 passes.  (136 is even: 0 == 0.)
    #    minorFrameID = ((0xFF & 1) << 8) | 0xFF = 511; spacecraft = 0xFF.
    ("group0_even", 1.0, _frame(**{f"b{k}": 0xFF for k in range(2, 19)}),
     dict(parity=0, minor_id=511, spacecraft=255, has_time=0)),
    # 3. as 2 with bytes[2] = 0xFE: 135 ones, odd; bit 5 of bytes[103] still 0 -> 1 != 0: check 0 FAILS, the others pass.
    ("group0_odd_unflagged", 1.0, _frame(**{**{f"b{k}": 0xFF for k in range(3, 19)}, "b2": 0xFE}),
     dict(parity=0b00001, minor_id=511, spacecraft=254, has_time=0)),
    # 4. as 3 with bytes[103] = 0x20 (bit 5 set): 1 == 1 -> passes again.
    ("group0_odd_flagged", 1.0, _frame(**{**{f"b{k}": 0xFF for k in range(3, 19)}, "b2": 0xFE, "b103": 0x20}),
     dict(parity=0, minor_id=511, spacecraft=254, has_time=0)),
    # 5. one frame exercising all five groups:
    #    group 0 (bytes 2..18):  0x0D (3 ones) + 0x01 (1) + 0x2C = 0b00101100 (3) = 7 ones, odd  -> needs bit 5 (0x20)
    #    group 1 (bytes 19..35): 0x01 = 1 one, odd                                                  -> needs bit 4 (0x10)
    #    group 2 (bytes 36..52): 0x03 = 2 ones, even                                                -> bit 3 clear
    #    group 3 (bytes 53..69): 0x07 = 3 ones, odd                                                 -> needs bit 2 (0x04)
    #    group 4 (bytes 70..86): 0x0F = 4 ones, even                                                -> bit 1 clear
    #    bytes[103] = 0x20 + 0x10 + 0x04 = 0x34 -> all pass.  minorFrameID = (1 << 8) | 0x2C = 300; spacecraft 13 = NOAA-18.
    ("all_groups_good", 7.5, _frame(b2=0x0D, b4=0x01, b5=0x2C, b19=0x01, b36=0x03, b53=0x07, b70=0x0F, b103=0x34),
     dict(parity=0, minor_id=300, spacecraft=13, has_time=0)),
    # 6. as 5 with bit 1 of bytes[103] flipped (0x36): group 4 has 4 ones (even) but its bit now says odd -> check 4 fails.
    ("group4_bit_flipped", 7.5, _frame(b2=0x0D, b4=0x01, b5=0x2C, b19=0x01, b36=0x03, b53=0x07, b70=0x0F, b103=0x36),
     dict(parity=0b10000, minor_id=300, spacecraft=13, has_time=0)),
    # 7. as 5 with bits 7, 6 and 0 of bytes[103] set as well (0x34 | 0xC1 = 0xF5): CPU flags and the sixth parity group are not
    #    looked at by checkParity.m -> still all pass.  Bytes outside every group (87..102) are free: bytes[95] = 0xFF.
    ("unchecked_bits", 7.5, _frame(b2=0x0D, b4=0x01, b5=0x2C, b19=0x01, b36=0x03, b53=0x07, b70=0x0F, b95=0xFF, b103=0xF5),
     dict(parity=0, minor_id=300, spacecraft=13, has_time=0)),
    # 8. major-frame start with a time code: bytes[4] & 1 = 0, bytes[5] = 0 -> id 0.
    #    day = (0x7C << 1) + ((0x83 | 128) >> 7) = 248 + 1 = 249
    #    ms  = ((0x83 & 7) << 24) + (0x5A << 16) + (0x33 << 8) + 0xBD = 50 331 648 + 5 898 240 + 13 056 + 189 = 56 243 133
    #    T0  = 56 243 133 - 12.5 * 1000 = 56 230 633
    #    parity: group 0 = bytes 2..18 = 0x08 (1) + 0x7C (5) + 0x83 (3) + 0x5A (4) + 0x33 (4) + 0xBD (6) = 23 ones, odd -> 0x20.
    ("time_code", 12.5, _frame(b2=0x08, b8=0x7C, b9=0x83, b10=0x5A, b11=0x33, b12=0xBD, b103=0x20),
     dict(parity=0, minor_id=0, spacecraft=8, has_time=1, day=249, day_ms=56243133, t0_ms=56230633)),
    # 9. as 8 with the top three bits of the ms counter all set and bytes[10] = 0xFF: ms = (7 << 24) + (0xFF << 16) + ... =
    #    117 440 512 + 16 711 680 + 13 056 + 189 = 134 165 437 >= 86 400 000 -> rejected (-1).  day = (0x7C << 1) + 1 = 249.
    #    group 0: 0x08 (1) + 0x7C (5) + 0x87 (4) + 0xFF (8) + 0x33 (4) + 0xBD (6) = 28 ones, even -> bit 5 clear.
    ("time_code_out_of_range", 12.5, _frame(b2=0x08, b8=0x7C, b9=0x87, b10=0xFF, b11=0x33, b12=0xBD, b103=0x00),
     dict(parity=0, minor_id=0, spacecraft=8, has_time=1, day=249, day_ms=-1)),
]

# Summary of the nine vectors as one capture (checkParity.m:88-92, daytimeDecode.m:34,82,95):
#   frames checked 9; vectors 3 and 6 have one failing check each -> 7 error-free frames, 2 bad chunks, 43 good chunks
#   spacecraft ids: 0, 255, 254, 254, 13, 13, 13, 8, 8 -> mode 13 (three times)
#   days of the major-frame starts: 1, 249, 249 -> mode 249
#   T0 over the positive ones: vector 1 gives 0 - 3250 < 0 (dropped), vector 8 gives 56 230 633, vector 9 is -1 -> 56 230 633
SUMMARY = dict(frames_checked=9, good_frames=7, bad_chunks=2, good_chunks=43, spacecraft=13, day=249, t0_ms=56230633, time_frames=3)


def check(records, summary):
    """records: structured array with the fields of pdt_tip_frame, in vector order; summary: dict"""
    for (name, _t, _b, want), r in zip(VECTORS, records):
        assert int(r["checked"]) == 1, name
        for k in ("parity", "minor_id", "spacecraft", "has_time"):
            assert int(r[k]) == want[k], (name, k, int(r[k]), want[k])
        if want["has_time"]:
            assert int(r["day"]) == want["day"], name
            assert int(r["day_ms"]) == want["day_ms"], name
    for k, v in SUMMARY.items():
        assert int(summary[k]) == v, (k, summary[k], v)
