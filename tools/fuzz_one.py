"""Replays one seed of tests/test_sequence_fuzz.py on the HIP libraries and on the oracle and reports the
first observation that differs (debugging aid for the fuzzer):  python tools/fuzz_one.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H
import test_sequence_fuzz as T

seed = int(sys.argv[1])
a, b = T.Program(seed).run(H.hip_backend()), T.Program(seed).run(H.oracle_backend())
print("env", {k: v for k, v in os.environ.items() if k.startswith("ARES_")}, "observations", len(a), len(b))
for i, (x, y) in enumerate(zip(a, b)):
    if x[:2] != y[:2]:
        print("kind differs at", i, x[:2], y[:2]); break
    if isinstance(x[2], dict):
        bad = [(k, x[2].get(k), v) for k, v in y[2].items() if x[2].get(k) != v]
        extra = [k for k in x[2] if k not in y[2]]
        if bad or extra:
            print("table differs at", i, x[:2], "groups", len(x[2]), len(y[2]), "wrong", len(bad), "extra", len(extra), bad[:3]); break
    elif x[2] != y[2]:
        print("differs at", i, x[:2], (x[2], y[2]) if not isinstance(x[2], bytes) else (len(x[2]), len(y[2]))); break
else:
    print("identical")
