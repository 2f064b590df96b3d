"""bench.py's host-side rules (no GPU): what `roofline.bound` says for a kernel, from the evidence the line carries."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_rules_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_bound_rule():
    b = load_bench()
    alg = 7_200_000_000
    # the bytes already move at >= 85 % of what streaming kernels reach: "hbm", whatever the issue counters say
    assert b.roofline_bound(5.33, 27_700_000_000, alg, 0.79) == "hbm"
    # the same kernel with less traffic per launch: the wavefronts issue in > 60 % of their cycles -> "issue"
    assert b.roofline_bound(5.44, 24_070_000_000, alg, 0.76) == "issue"
    # no SQ pass of this build committed: the traffic alone decides between "hbm" (more than half the time at that rate) and "latency"
    assert b.roofline_bound(5.44, 24_070_000_000, alg, None) == "hbm"
    assert b.roofline_bound(5.44, None, alg, None) == "latency"          # algorithmic bytes only: 1.3 TB/s
    assert b.roofline_bound(0.0, None, alg, None) == "latency"
    # a streaming kernel is "hbm" by its algorithmic bytes alone
    assert b.roofline_bound(1.28, None, alg, None) == "hbm"
