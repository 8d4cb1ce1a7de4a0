"""Row f3 of SURVEY.md 8: host batches through libaresdriver.so's transfer pipeline (upload of batch
k+1 overlapped with the execution of batch k on the query's second stream, columns freed by the
executor, query/aql_processor.go:850-881) and its device-resident column cache — results must equal
the resident path's, on every backend; the GPU variant runs two streams and pinned memory for real."""
import numpy as np
import pytest

import harness as H
from aresdb_amd import abi, smoke
from aresdb_amd.driver import ColumnCache, HostColumn, NativeQuery

NAMES = ["ts", "d1", "d2", "d3", "m"]


def _host_batches(be, data, keyed):
    out = []
    for b, (cols, valid) in enumerate(data):
        hcs = [HostColumn(be, cols[k][0], cols[k][1], valid=valid[k], cache_key=(1000 * (b + 1) + i + 1) if keyed else 0)
               for i, k in enumerate(NAMES)]
        out.append((hcs, len(cols["ts"][1])))
    return out


def _result(q, plan):
    dims, valids, meas = q.fetch()
    n = q.result_size
    m = meas.view(np.float64)
    return {tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids)): m[r]
            for r in range(n)}


@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_overlapped_host_batches_equal_the_resident_path(be, use_hash):
    rng = np.random.default_rng(31)
    data = [smoke.synth_batch(rng, n, null_fraction=0.02) for n in (9000, 1, 14000, 6000, 11000)]
    plan = smoke.c3_plan(use_hash)
    want, _ = smoke.run_query(be, plan, data)
    streams = [be.call("CreateCudaStream", 0) for _ in range(2)]
    hb = _host_batches(be, data, keyed=False)
    q = NativeQuery(be, plan, NAMES, streams=streams)
    stats = q.run_host_batches(hb)
    smoke.compare_results(_result(q, plan), want)
    assert stats["uploads"] == 5 * len(data) and stats["cache_hits"] == 0
    assert stats["uploaded_bytes"] == sum(hc.nbytes for cols, _ in hb for hc in cols)
    q.release()
    for cols, _ in hb:
        for hc in cols:
            hc.free()
    for s in streams:
        be.call("DestroyCudaStream", s, 0)


def test_column_cache_keeps_hot_batches_resident(be):
    rng = np.random.default_rng(32)
    data = [smoke.synth_batch(rng, 8000, null_fraction=0.01) for _ in range(4)]
    plan = smoke.c3_plan(True)
    want, _ = smoke.run_query(be, plan, data)
    hb = _host_batches(be, data, keyed=True)
    per_batch = sum(hc.nbytes for hc in hb[0][0])
    cache = ColumnCache(be, 0, budget_bytes=int(2.5 * per_batch))  # room for two of the four batches
    seen = []
    for _ in range(3):
        q = NativeQuery(be, plan, NAMES)
        seen.append(q.run_host_batches(hb, cache))
        smoke.compare_results(_result(q, plan), want)
        q.release()
    assert seen[0]["cache_hits"] == 0 and seen[0]["uploads"] == 20
    assert all(s["cache_bytes"] <= 2.5 * per_batch for s in seen)
    big = ColumnCache(be, 0, budget_bytes=8 * per_batch)  # everything fits: the second query uploads nothing
    q = NativeQuery(be, plan, NAMES)
    first = q.run_host_batches(hb, big)
    q.release()
    q = NativeQuery(be, plan, NAMES)
    second = q.run_host_batches(hb, big)
    smoke.compare_results(_result(q, plan), want)
    q.release()
    assert first["uploads"] == 20 and second["uploads"] == 0 and second["cache_hits"] == 20 and second["uploaded_bytes"] == 0
    cache.destroy()
    big.destroy()
    for cols, _ in hb:
        for hc in cols:
            hc.free()


@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_resident_batches_in_one_driver_call_equal_one_call_per_batch(be, use_hash):
    """AresQueryRunResidentBatches (what bench.py times as a step): the per-batch call sequence over a list of resident
    batches without returning to the caller in between — same result, same number of ABI calls as one
    AresQueryRunBatch per batch."""
    from aresdb_amd.columns import DeviceColumn
    if be.name == "hip":
        pytest.skip("covered on the GPU by bench.py's timed path and smoke(); this test pins the driver logic on the CPU backends")
    rng = np.random.default_rng(33)
    data = [smoke.synth_batch(rng, n, null_fraction=0.02) for n in (9000, 1, 14000, 6000)]
    plan = smoke.c3_plan(use_hash)
    want, _ = smoke.run_query(be, plan, data)
    dev = [({k: DeviceColumn(be, t, v, valid=valid[k]) for k, (t, v) in cols.items()}, len(cols["ts"][1])) for cols, valid in data]
    one = NativeQuery(be, plan, NAMES)
    for cols, n in dev:
        one.run({k: c.vp for k, c in cols.items()}, n)
    smoke.compare_results(_result(one, plan), want)
    many = NativeQuery(be, plan, NAMES)
    packed = many.pack_batches([({k: c.vp for k, c in cols.items()}, n) for cols, n in dev])
    many.run_batches(packed)
    smoke.compare_results(_result(many, plan), want)
    assert many.result_size == one.result_size and many.calls == one.calls
    one.release()
    many.release()
    for cols, _ in dev:
        for c in cols.values():
            c.free()
