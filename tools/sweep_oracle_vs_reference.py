"""Wider differential sweeps of the C oracle against the reference's own HOST build (oracle/_ref), CPU only:

    python tools/sweep_oracle_vs_reference.py FAMILY FIRST COUNT

FAMILY: transform | lookup | sort | hash | hll | geo (the case generators of tests/cases.py with fresh seeds) or
sequence (the whole-query programs of tests/test_sequence_fuzz.py).  Prints the mismatching seeds.  Round 4: 20 000 seeds
of each case family from 200 000 and 20 000 sequence programs from 10 000 — no mismatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import harness as H  # noqa: E402
import test_sequence_fuzz as T  # noqa: E402

fam, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
o, r = H.oracle_backend(), H.ref_backend()
bad, t0 = [], time.time()
for seed in range(first, first + count):
    try:
        if fam == "transform":
            for arity in (1, 2):
                for as_filter in (False, True):
                    c = cases.TransformCase(seed * 4 + arity * 2 + int(as_filter), arity, as_filter)
                    cases.assert_same(c.run(o), c.run(r), repr(c))
        elif fam == "lookup":
            c = cases.HashLookupCase(seed)
            cases.assert_same(c.run(o), c.run(r), f"HashLookupCase({seed})")
        elif fam == "sort":
            c = cases.GroupByCase(seed)
            cases.assert_same(c.run_sort_reduce(o), c.run_sort_reduce(r), f"GroupByCase({seed}) sort/reduce")
        elif fam == "hash":
            c = cases.GroupByCase(seed)
            cases.assert_same(c.run_hash_reduce(o), c.run_hash_reduce(r), f"GroupByCase({seed}) hash reduce")
        elif fam == "hll":
            c = cases.HllCase(seed)
            cases.assert_same(c.run(o), c.run(r), repr(c))
        elif fam == "geo":
            c = cases.GeoCase(seed)
            cases.assert_same(c.run(o), c.run(r), repr(c))
        elif fam == "sequence":
            p = T.Program(seed)
            T._same(p.run(r), p.run(o), seed)
        else:
            raise SystemExit(f"unknown family {fam}")
    except AssertionError as e:
        bad.append((seed, str(e)[:300]))
    except Exception as e:  # noqa: BLE001
        bad.append((seed, f"{type(e).__name__}: {e}"[:300]))
print(f"{fam}: {count} seeds from {first} in {time.time() - t0:.0f} s, mismatches: {bad[:10]} ({len(bad)})", flush=True)
