"""debug: one shape of tests/test_sort_fused.py with the fused-path trace on"""
import os, sys
os.environ.setdefault("ARES_RTC_ASYNC", "0")
os.environ["ARES_HR_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import harness as H
import test_sort_fused as T
name = sys.argv[1] if len(sys.argv) > 1 else "sum8_two_dims"
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5000, 1, 40000, 700]
shape = [s for s in T.SHAPES if s.name == name][0]
hip = H.hip_backend()
rng = np.random.default_rng(sum(map(ord, shape.name)))
batches = [T.make_batch(rng, shape, n) for n in sizes]
for i, bt in enumerate(batches):
    print("== batch", i, flush=True)
got, kernels = T._kernels_of(hip, lambda: T.run_sequence(hip, shape, batches))
print([ (e["kept"], e["groups"]) for e in got])
print(sorted(kernels))
