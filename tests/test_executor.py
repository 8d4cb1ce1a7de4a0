"""Query-level replay: the Go batch executor's call sequence (aresdb_amd/executor.py) against
every backend; results must agree with the oracle and, independently, with a numpy group-by."""
import numpy as np
import pytest

import harness as H
from aresdb_amd import abi, smoke
from aresdb_amd.executor import Binary, Col, Const, DimensionSpec, QueryPlan, Unary


def numpy_c3(batches, with_filter=True):
    out = {}
    for cols, valid in batches:
        ts, d1, d2, d3, m = (cols[k][1] for k in ("ts", "d1", "d2", "d3", "m"))
        n = len(ts)
        ok = {k: (np.ones(n, bool) if valid[k] is None else valid[k]) for k in cols}
        keep = (d1 < 90) & ok["d1"] if with_filter else np.ones(n, bool)
        for i in np.nonzero(keep)[0]:
            def dim(v, k):
                return (np.uint32(v if ok[k][i] else 0).tobytes(), int(ok[k][i]))
            # a null input yields value 0 / validity 0 in the dimension vector — except Noop, which
            # passes the stored value through (query/functor.hpp:345-351)
            key = (dim(ts[i] - ts[i] % 3600, "ts"),
                   (np.uint32(d1[i]).tobytes(), int(ok["d1"][i])),
                   (np.uint32(d2[i]).tobytes(), int(ok["d2"][i])),
                   (np.uint32(d3[i]).tobytes(), int(ok["d3"][i])))
            out[key] = out.get(key, 0.0) + (float(m[i]) if ok["m"][i] else 0.0)
    return out


@pytest.mark.parametrize("use_hash", [True, False])
def test_c3_shape_matches_numpy_group_by(be, use_hash):
    rng = np.random.default_rng(3)
    data = [smoke.synth_batch(rng, 5000, null_fraction=0.02) for _ in range(3)]
    got, calls = smoke.run_query(be, smoke.c3_plan(use_hash), data)
    want = numpy_c3(data)
    smoke.compare_results(got, want)
    assert calls > 0


@pytest.mark.parametrize("use_hash", [True, False])
def test_c3_shape_matches_oracle(be, use_hash):
    rng = np.random.default_rng(11)
    data = [smoke.synth_batch(rng, 3000) for _ in range(2)]
    got, _ = smoke.run_query(be, smoke.c3_plan(use_hash), data)
    want, _ = smoke.run_query(H.oracle_backend(), smoke.c3_plan(use_hash), data)
    smoke.compare_results(got, want)


def test_count_star_with_inner_expression(be):
    """COUNT(*) = sum of constant 1 as AGGR_SUM_UNSIGNED over the sort path, with a nested filter
    (ts % 7 == 3) that needs a scratch frame (aql_compiler.go:1191-1197)."""
    rng = np.random.default_rng(5)
    n = 4000
    ts = rng.integers(0, 1 << 20, n).astype(np.uint32)
    d1 = rng.integers(0, 9, n).astype(np.uint32)
    plan = QueryPlan(
        filters=[Binary(abi.Equal, Binary(abi.Mod, Col("ts"), Const(7), abi.Int32), Const(3)),
                 Unary(abi.IsNotNull, Col("d1"))],
        dimensions=[DimensionSpec(Col("d1"), abi.Uint32)], measure=Const(1),
        agg=abi.AGGR_SUM_UNSIGNED, measure_type=abi.Uint32, use_hash_reduction=False)
    valid = rng.random(n) > 0.1
    batches = [({"ts": (abi.Uint32, ts), "d1": (abi.Uint32, d1)}, {"ts": None, "d1": valid})]
    got, _ = smoke.run_query(be, plan, batches)
    keep = (ts.astype(np.int32) % 7 == 3) & valid
    want = {}
    for v in d1[keep]:
        k = ((np.uint32(v).tobytes(), 1),)
        want[k] = want.get(k, 0) + 1
    assert {k: int(v) for k, v in got.items()} == want
