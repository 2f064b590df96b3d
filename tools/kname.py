"""Readable kernel names from rocprofv3 output.  Every kernel is entered through the k_run<&body, threads> wrapper
(csrc/pdt_rt.h), whose mangled name the profiler cannot demangle; the body's name and template arguments are inside it:
..._ZN3pdt11k_pll_phaseIfLb0EEEv...  ->  k_pll_phase<float, false>"""
import re


def kernel_name(raw: str) -> str:
    raw = re.sub(r"\(.*", "", raw).replace("void ", "").replace("pdt::", "").strip()
    m = re.search(r"_ZN3pdt(\d+)", raw)
    if not m or "k_run" not in raw:
        return raw
    n = int(m.group(1))
    pos = m.end()
    name = raw[pos:pos + n]
    pos += n
    args = []
    if raw[pos:pos + 1] == "I":
        pos += 1
        while pos < len(raw) and raw[pos] != "E":
            c = raw[pos]
            if c == "f":
                args.append("float"); pos += 1
            elif c == "d":
                args.append("double"); pos += 1
            elif c == "L":
                e = raw.index("E", pos)
                lit = raw[pos + 1:e]
                args.append({"b0": "false", "b1": "true"}.get(lit, lit[1:] if lit[:1] in "ijlm" else lit))
                pos = e + 1
            else:
                break
    return name + ("<" + ", ".join(args) + ">" if args else "")


if __name__ == "__main__":
    import sys
    for line in sys.stdin:
        print(kernel_name(line.strip()))
