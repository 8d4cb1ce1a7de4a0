"""One-off wider sweep of the sequence fuzzer (tests/test_sequence_fuzz.py) on the GPU box:
python tools/fuzz_sweep.py FIRST COUNT  — replays seeds on the HIP libraries and on the oracle, prints mismatching seeds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H
import test_sequence_fuzz as T

first, count = int(sys.argv[1]), int(sys.argv[2])
hip, oracle = H.hip_backend(), H.oracle_backend()
bad, t0 = [], time.time()
for seed in range(first, first + count):
    try:
        p = T.Program(seed)
        T._same(p.run(hip), p.run(oracle), seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
    except Exception as e:  # noqa: BLE001
        bad.append((seed, f"{type(e).__name__}: {e}"[:200]))
print(f"env {dict((k, v) for k, v in os.environ.items() if k.startswith('ARES_'))}: {count} seeds from {first} in {time.time() - t0:.0f} s, mismatches: {bad}")
