"""BASELINE config C1 (examples/1k_trips, plumbing): both example queries on every backend against
the committed fixture, whose expected results were produced by the reference's own HOST build
(tests/golden/make_1k_trips.py)."""
import json
import os

import pytest

from trips_queries import run_trips_query, trips_plans

DATA = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "1k_trips.json")))


@pytest.mark.parametrize("query", ["total_trips", "total_fare"])
def test_1k_trips_example_queries(be, query):
    got = run_trips_query(be, trips_plans(DATA)[query], DATA)
    want = {int(k): v for k, v in DATA["expected"][query].items()}
    assert got.keys() == want.keys()
    for k, v in want.items():
        assert abs(got[k] - v) <= 1e-6 * max(1.0, abs(v)), (k, got[k], v)
    assert sum(1 for _ in got) > 10  # a day of trips spreads over its hours
