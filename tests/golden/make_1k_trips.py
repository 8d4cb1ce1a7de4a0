"""Builds tests/golden/1k_trips.json — BASELINE config C1 (examples/1k_trips) as a committed fixture.

Run in the authoring container only (needs /root/reference and the reference HOST build in
oracle/_ref).  The 1 000 rows come from the reference's examples/1k_trips/data/trips.csv; the
loader's random request times (examples/utils/example_utils.go:40-55: uniformly inside the last
day) are replaced by a seeded draw against a fixed "now".  The expected results of the two example
queries (examples/1k_trips/queries/total_trips.aql, total_fare.aql) are produced by running the
reference's own libalgorithm.so (QUERY_MODE=HOST) through the ABI call sequence of the Go batch
executor, and cross-checked against a plain numpy group-by before they are written.

    python tests/golden/make_1k_trips.py
"""
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("ARES_REFERENCE", "/root/reference")
NOW = 1546300800  # 2019-01-01T00:00:00Z, "this quarter-hour" boundary


def main():
    import harness as H
    from trips_queries import run_trips_query, trips_plans
    rows = list(csv.DictReader(open(os.path.join(REF, "examples", "1k_trips", "data", "trips.csv"))))
    rng = np.random.default_rng(1000)
    request_at = (NOW - 86400 + rng.integers(0, 86400, len(rows))).astype(np.uint32)
    enum = {}
    status = np.array([enum.setdefault(r["status"], len(enum)) for r in rows], np.uint8)  # SmallEnum ids by first appearance
    city = np.array([int(r["city_id"]) for r in rows], np.uint16)
    fare = np.array([float(r["fare"]) for r in rows], np.float32)
    data = {"now": NOW, "enum_status": enum, "request_at": request_at.tolist(), "status": status.tolist(),
            "city_id": city.tolist(), "fare_f32_bits": fare.view(np.uint32).tolist()}
    ref = H.ref_backend()
    expected = {}
    for name, plan in trips_plans(data).items():
        got = run_trips_query(ref, plan, data)
        # independent numpy group-by
        keep = (status == enum["completed"]) & (request_at >= NOW - 86400) & (request_at < NOW)
        want = {}
        for ts, f in zip(request_at[keep], fare[keep]):
            k = int(ts - ts % 3600)
            want[k] = want.get(k, 0) + (1 if name == "total_trips" else float(f))
        assert got.keys() == want.keys(), name
        for k in want:
            assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (name, k, got[k], want[k])
        expected[name] = {str(k): (int(v) if name == "total_trips" else float(v)) for k, v in sorted(got.items())}
    data["expected"] = expected
    json.dump(data, open(os.path.join(HERE, "1k_trips.json"), "w"))
    print("wrote", len(rows), "rows;", {k: len(v) for k, v in expected.items()}, "groups")


if __name__ == "__main__":
    main()
