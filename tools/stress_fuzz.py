"""Race hunting: the sequence fuzzer's four-thread round, repeated in ONE process; mismatches are counted, not fatal.
usage: python tools/stress_fuzz.py --iters 60 [--threads 4] [--tag name]   (environment selects the library variant)"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ARES_RTC_ASYNC", "0")

import harness as H  # noqa: E402
from test_sequence_fuzz import PROFILES, Program  # noqa: E402


def first_diff(a, b):
    if len(a) != len(b):
        return ("length", len(a), len(b))
    for k, (x, y) in enumerate(zip(a, b)):
        if x[:2] != y[:2]:
            return ("kind", k, x[:2], y[:2])
        if isinstance(x[2], dict):
            if x[2].keys() != y[2].keys():
                return (x[0], x[1], "keys", len(x[2]), len(y[2]))
            for key, v in y[2].items():
                g = x[2][key]
                if not (g == v or abs(float(g) - float(v)) <= 1e-6 * max(1.0, abs(float(v)))):
                    return (x[0], x[1], "value")
        elif x[2] != y[2]:
            if isinstance(x[2], bytes):
                n = sum(1 for p, q in zip(x[2], y[2]) if p != q)
                f = next(i for i, (p, q) in enumerate(zip(x[2], y[2])) if p != q)
                return (x[0], x[1], "bytes", len(x[2]), "differ", n, "first", f, x[2][f], y[2][f])
            return (x[0], x[1], x[2], y[2])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--tag", default="run")
    ap.add_argument("--seeds", type=int, default=48)
    ap.add_argument("--profiles", action="store_true", help="cycle the fuzzer's profiles (narrow columns, drifting time filters) over the seeds")
    ap.add_argument("--kernels", action="store_true", help="report the launches per kernel (the library's profiler: HIP events around every launch)")
    ap.add_argument("--budget-s", type=float, default=0.0, help="stop starting new rounds after this many seconds (0: run every round)")
    a = ap.parse_args()
    prof = (lambda s: PROFILES[s % len(PROFILES)]) if a.profiles else (lambda s: ())
    hip, oracle = (H.oracle_backend() if os.environ.get("STRESS_ON_ORACLE") else H.hip_backend()), H.oracle_backend()
    want = {}
    if a.kernels and getattr(hip, "has_profiler", False):
        hip.profiler_enable(True)
    t0 = time.time()
    bad, errs, programs = [], [], 0
    for it in range(a.iters):
        if a.budget_s and time.time() - t0 > a.budget_s:
            break
        seeds = [2000 + (it * a.threads + t) % a.seeds for t in range(a.threads)]
        for s in seeds:
            if s not in want:
                want[s] = Program(s, prof(s)).run(oracle)
        got = [None] * a.threads

        def work(t):
            try:
                got[t] = Program(seeds[t], prof(seeds[t])).run(hip, expect=want[seeds[t]])
            except Exception as e:  # noqa: BLE001
                errs.append((it, seeds[t], repr(e)[:200]))
        ths = [threading.Thread(target=work, args=(t,)) for t in range(a.threads)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        for t in range(a.threads):
            programs += 1
            if got[t] is not None:
                d = first_diff(got[t], want[seeds[t]])
                if d:
                    bad.append((it, seeds[t], d))
    kernels = None
    if a.kernels and getattr(hip, "has_profiler", False):
        hip.wait()
        kernels = {k: v[0] for k, v in hip.profiler_report().items()}
        hip.profiler_enable(False)
    print(json.dumps({"kernels": kernels, "tag": a.tag, "iters": a.iters, "programs": programs, "mismatches": len(bad), "errors": len(errs),
                      "seconds": round(time.time() - t0, 1), "first": [list(map(str, b)) for b in bad[:6]], "errs": errs[:4]}))


if __name__ == "__main__":
    main()
