"""Race hunting, part 2: is HOST memory being corrupted?  (The fuzz failures of rounds 3/4 turned out to be single
elements of the *input* numpy arrays differing between the oracle run and the library run: +-1 on a 16-byte aligned
word.)  A thread phase (variant-dependent), then a single-threaded loop of fuzz programs with canary arrays that are
checked after every program.  One process per variant.

  --threads none | fuzz | libmem | hip | hipnull      what the four short-lived threads of the thread phase do
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ARES_RTC_ASYNC", "0")

import harness as H  # noqa: E402
from test_sequence_fuzz import Program  # noqa: E402
from stress_fuzz import first_diff  # noqa: E402

PATTERN = 0x55555555


def thread_phase(kind, hip, oracle, rounds):
    if kind == "none":
        return
    hiplib = None
    if kind in ("hip", "hipnull"):
        hiplib = C.CDLL("libamdhip64.so")
        hiplib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hiplib.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hiplib.hipStreamSynchronize.argtypes = [C.c_void_p]
        hiplib.hipFree.argtypes = [C.c_void_p]
        hiplib.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        hiplib.hipStreamDestroy.argtypes = [C.c_void_p]
    for rnd in range(rounds):
        def work(t):
            if kind == "fuzz":
                Program(2000 + 10 * rnd + t).run(hip)
            elif kind == "libmem":
                for k in range(200):
                    a = np.arange(30000, dtype=np.uint32) + k
                    p = hip.device_alloc(a.nbytes)
                    hip.h2d(p, a.ctypes.data_as(C.c_void_p), a.nbytes)
                    hip.wait()
                    b = np.empty_like(a)
                    hip.d2h(b.ctypes.data_as(C.c_void_p), p, a.nbytes)
                    hip.wait()
                    assert np.array_equal(a, b)
                    hip.device_free(p)
            else:  # the HIP runtime alone: no library of ours is involved
                s = C.c_void_p(0)
                if kind == "hip":
                    assert hiplib.hipStreamCreateWithFlags(C.byref(s), 1) == 0
                for k in range(200):
                    a = np.arange(30000, dtype=np.uint32) + k
                    p = C.c_void_p(0)
                    assert hiplib.hipMalloc(C.byref(p), a.nbytes) == 0
                    assert hiplib.hipMemcpyAsync(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1, s) == 0
                    assert hiplib.hipStreamSynchronize(s) == 0
                    b = np.empty_like(a)
                    assert hiplib.hipMemcpyAsync(b.ctypes.data_as(C.c_void_p), p, a.nbytes, 2, s) == 0
                    assert hiplib.hipStreamSynchronize(s) == 0
                    assert np.array_equal(a, b)
                    assert hiplib.hipFree(p) == 0
                if kind == "hip":
                    assert hiplib.hipStreamDestroy(s) == 0
        ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="fuzz")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--programs", type=int, default=320)
    ap.add_argument("--tag", default="run")
    a = ap.parse_args()
    hip, oracle = (H.oracle_backend() if os.environ.get("STRESS_ON_ORACLE") else H.hip_backend()), H.oracle_backend()
    t0 = time.time()
    thread_phase(a.threads, hip, oracle, a.rounds)
    # canaries: heap arrays of the sizes the fuzzer's inputs have, allocated after the threads are gone
    canaries = [np.full(int(n), PATTERN, np.uint32) for n in np.random.default_rng(7).integers(2000, 31000, 600)]
    hits, bad = [], []
    for k in range(a.programs):
        seed = 1000 + k % 160
        p = Program(seed)
        want = p.run(oracle)
        got = p.run(hip, expect=want)
        d = first_diff(got, want)
        if d:
            bad.append((k, seed, [str(x) for x in d][:8]))
        for ci, c in enumerate(canaries):
            w = np.flatnonzero(c != PATTERN)
            for i in w[:4]:
                addr = c.ctypes.data + 4 * int(i)
                hits.append({"program": k, "canary": ci, "index": int(i), "addr_mod_64": addr % 64, "value": hex(int(c[i])),
                             "delta": int(c[i]) - PATTERN})
            if len(w):
                c[w] = PATTERN
        if k % 40 == 0:  # churn: free and reallocate a tenth of the canaries
            for ci in range(k % 10, len(canaries), 10):
                canaries[ci] = np.full(len(canaries[ci]), PATTERN, np.uint32)
    print(json.dumps({"tag": a.tag, "threads": a.threads, "programs": a.programs, "fuzz_mismatches": len(bad), "canary_hits": len(hits),
                      "seconds": round(time.time() - t0, 1), "hits": hits[:12], "bad": bad[:6]}))


if __name__ == "__main__":
    main()
