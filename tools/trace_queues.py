"""Per-hardware-queue occupancy of a rocprofv3 --kernel-trace CSV: which queue ran what when, how busy each queue was.
usage: python tools/trace_queues.py <kernel_trace.csv> [kernels_in_last_step]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
seg = rows[-n:]
s = min(int(r["Start_Timestamp"]) for r in seg)
e = max(int(r["End_Timestamp"]) for r in seg)
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"{len(seg)} dispatches on {len(set(r['Stream_Id'] for r in seg))} streams / {len(set(r['Queue_Id'] for r in seg))} hardware queues: "
      f"span {(e - s) / 1e6:.3f} ms, sum of kernel durations {tot / 1e6:.3f} ms")
byq = collections.defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]) - s, int(r["End_Timestamp"]) - s, r["Stream_Id"],
                               re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("pdt::", "")[:40]))
for q in sorted(byq):
    l = byq[q]
    busy = sum(b - a for a, b, _, _ in l)
    print(f"queue {q}: {len(l)} dispatches from streams {sorted(set(x[2] for x in l))}, busy {busy / 1e6:.3f} ms = {100.0 * busy / (e - s):.0f} % of the span")
q0 = sorted(byq)[0]
print(f"\ntimeline of queue {q0} (ms from the first dispatch):")
for a, b, st, nm in sorted(byq[q0]):
    print(f"{a / 1e6:9.3f} {b / 1e6:9.3f} {(b - a) / 1e3:9.1f} us  stream {st:>3s}  {nm}")
