"""Reduce a rocprofv3 PC-sampling CSV to a per-kernel, per-instruction histogram (round 4: attribution of the walkers' stalls).
usage: pc_hist.py <pc_sampling.csv> <kernel_trace.csv> <out.json> [kernel-name substrings...]
Columns differ between the host-trap and the stochastic method; whatever is there is used:
  Instruction / Instruction_Comment (decoded text, when the tool could read the code object), Dispatch_Id or Correlation_Id
  (-> kernel name through the kernel trace), Wave_Issued_Instruction, Instruction_Type, Stall_Reason (stochastic only)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import kernel_name

pc_csv, kt_csv, out_json = sys.argv[1:4]
want = sys.argv[4:]
csv.field_size_limit(1 << 30)

names = {}
if kt_csv and os.path.exists(kt_csv):
    for r in csv.DictReader(open(kt_csv)):
        k = kernel_name(r.get("Kernel_Name", ""))
        for key in ("Dispatch_Id", "Correlation_Id"):
            if r.get(key):
                names[(key, r[key])] = k

rd = csv.DictReader(open(pc_csv))
cols = rd.fieldnames
print("columns:", cols)
inst_col = next((c for c in cols if c == "Instruction"), None) or next((c for c in cols if "nstruction" in c and "Type" not in c and "Issued" not in c and "Comment" not in c), None)
com_col = next((c for c in cols if "Comment" in c), None)
stall_col = next((c for c in cols if "Stall" in c), None)
issued_col = next((c for c in cols if "Issued" in c), None)
type_col = next((c for c in cols if "Instruction_Type" in c), None)
id_col = next((c for c in ("Dispatch_Id", "Correlation_Id") if c in cols), None)
exec_col = next((c for c in cols if "Exec" in c), None)

hist = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.Counter()))
tot = collections.defaultdict(collections.Counter)
n = 0
for r in rd:
    n += 1
    k = names.get((id_col, r.get(id_col, "")), "dispatch " + str(r.get(id_col)))
    if want and not any(w in k for w in want):
        continue
    inst = (r.get(inst_col) or "?").strip()
    com = (r.get(com_col) or "").strip() if com_col else ""
    key = inst + ("   ; " + com if com else "")
    issued = r.get(issued_col, "") if issued_col else ""
    stall = r.get(stall_col, "") if stall_col else ""
    tag = ("issued" if str(issued) in ("1", "True", "true") else (stall or "not_issued"))
    hist[k][key][tag] += 1
    tot[k][tag] += 1
    tot[k]["_samples"] += 1
    if type_col:
        tot[k]["type:" + r.get(type_col, "")] += 1
print("samples:", n)
res = {}
for k in sorted(tot, key=lambda k: -tot[k]["_samples"]):
    s = tot[k]["_samples"]
    print(f"\n=== {k}: {s} samples; " + ", ".join(f"{t} {c * 100.0 / s:.1f}%" for t, c in tot[k].most_common() if t != "_samples"))
    rows = sorted(hist[k].items(), key=lambda kv: -sum(kv[1].values()))
    res[k] = {"samples": s, "by_reason": dict(tot[k]), "top": []}
    for key, c in rows[:60]:
        tt = sum(c.values())
        res[k]["top"].append({"inst": key, "samples": tt, "reasons": dict(c)})
        print(f"  {tt * 100.0 / s:6.2f}%  {key[:110]:110s} " + " ".join(f"{t}:{v}" for t, v in c.most_common(4)))
json.dump(res, open(out_json, "w"), indent=1)
