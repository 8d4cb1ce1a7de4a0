#!/usr/bin/env python3
"""Write-tracking sweep: the sequence fuzzer (tests/test_sequence_fuzz.py) with ARES_MEM_VERIFY_CLEAN=1 — libmem.so then
checks, on the device, every block it hands out as "still cleared" and aborts on stale bytes (a kernel output that
libalgorithm.so did not report with AresMemNoteWrite).  One process, so the environment is set before the
libraries load:

    python tools/verify_clean_sweep.py [--first 1000] [--seeds 120] [--thread-rounds 3]

Every program's observable buffers are compared with the oracle's as in the test; exit code 0 = clean.
"""
import argparse
import os
import sys
import threading

os.environ["ARES_MEM_VERIFY_CLEAN"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import harness as H  # noqa: E402
from test_sequence_fuzz import Program, _same  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--seeds", type=int, default=120)
    ap.add_argument("--thread-rounds", type=int, default=3, help="rounds of four programs on four host threads / streams")
    args = ap.parse_args()
    hip, oracle = H.hip_backend(), H.oracle_backend()
    for seed in range(args.first, args.first + args.seeds):
        p = Program(seed)
        _same(p.run(hip), p.run(oracle), seed)
    for rnd in range(args.thread_rounds):
        seeds = [5000 + 10 * rnd + t for t in range(4)]
        want = [Program(s).run(oracle) for s in seeds]
        got, errs = [None] * 4, []

        def work(t):
            try:
                got[t] = Program(seeds[t]).run(hip)
            except Exception as e:  # noqa: BLE001
                errs.append((seeds[t], repr(e)))
        threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if errs:
            raise SystemExit(f"errors: {errs}")
        for t in range(4):
            _same(got[t], want[t], seeds[t])
    print(f"verify-clean sweep: {args.seeds} programs + {args.thread_rounds} x 4 threaded programs clean")


if __name__ == "__main__":
    main()
