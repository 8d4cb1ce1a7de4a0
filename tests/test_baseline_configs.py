"""BASELINE.json configs C2 and C4 as GPU parity cases (C3 at full size is bench.py; its small-size
parity lives in test_executor.py).  Sizes are reduced; results are checked against independent numpy
computations (size-independent properties: exact counts, exact integer sums, key sets)."""
import numpy as np
import pytest

import cases
import harness as H
from aresdb_amd import abi, queries
from aresdb_amd.columns import DeviceColumn
from aresdb_amd.driver import NativeQuery
from aresdb_amd.executor import Binary, Col, Const, DimensionSpec, ForeignTable, QueryPlan

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("selectivity", [0.1, 0.5, 0.9])
def test_c2_filter_count(selectivity):
    """C2: single uint32 predicate + COUNT(*) — no dimensions, measure literal 1, AGGR_SUM_UNSIGNED,
    sort path (query/aql_compiler.go:1191-1197); 3 batches, 1 % nulls."""
    be = H.hip_backend()
    rng = np.random.default_rng(1)
    n, hi = 4_000_000, 86400 * 30
    thr = int(hi * selectivity)
    q = NativeQuery(be, queries.c2_plan(thr), ["ts"])
    want = 0
    for _ in range(3):
        ts = rng.integers(0, hi, n).astype(np.uint32)
        valid = rng.random(n) >= 0.01
        col = DeviceColumn(be, abi.Uint32, ts, valid=valid)
        q.run({"ts": col.vp}, n)
        col.free()
        want += int(((ts < thr) & valid).sum())
    dims, valids, meas = q.fetch()
    assert q.result_size == 1
    assert int(meas.view(np.uint32)[0]) == want
    q.release()


def test_c4_join_high_cardinality_sort_reduce():
    """C4: foreign-key HashLookup into a dimension table + high-cardinality group-by through
    Sort + Reduce (one group per key), SUM of uint32 — exact against numpy."""
    be = H.hip_backend()
    rng = np.random.default_rng(4)
    nkeys, per_batch, n = 20000, 8192, 1_500_000
    keys = rng.choice(1 << 24, nkeys, replace=False).astype(np.uint32)
    attr = rng.integers(0, 1000, nkeys).astype(np.uint32)
    seeds = [int(x) for x in rng.integers(0, 1 << 32, 4)]
    nb = (nkeys + per_batch - 1) // per_batch
    table, placed = cases.build_cuckoo([int(k).to_bytes(4, "little") for k in keys], 4, nkeys // 6, seeds, rng,
                                       record_of=lambda i: (1 + i // per_batch, i % per_batch))
    usable = np.array([i for i, k in enumerate(keys) if int(k).to_bytes(4, "little") in placed])  # a few keys overflow the stash
    assert len(usable) > nkeys * 0.99
    tb = H.Buf(be, table)
    idx = abi.CuckooHashIndex()
    idx.buckets = tb.ptr
    for i, s in enumerate(seeds):
        idx.seeds[i] = s
    idx.keyBytes, idx.numHashes, idx.numBuckets = 4, 4, nkeys // 6
    dcols = [H.Column(be, abi.Uint32, attr[b * per_batch:(b + 1) * per_batch]) for b in range(nb)]
    ft = ForeignTable(join_column="fk", index=idx, batches={"attr": [c.vp for c in dcols]},
                      data_types={"attr": abi.Uint32}, base_batch_id=1,
                      num_records_in_last_batch=nkeys - (nb - 1) * per_batch)
    plan = QueryPlan(filters=[], foreign_tables=[ft], foreign_filters=[],
                     dimensions=[DimensionSpec(Col("fk"), abi.Uint32), DimensionSpec(Col("attr", table=1), abi.Uint32)],
                     measure=Col("amount"), agg=abi.AGGR_SUM_UNSIGNED, measure_type=abi.Uint32, use_hash_reduction=False)
    q = NativeQuery(be, plan, ["fk", "amount"])
    sums = np.zeros(nkeys, np.int64)
    for _ in range(2):
        pick = usable[rng.integers(0, len(usable), n)]
        amount = rng.integers(0, 100, n).astype(np.uint32)
        cf, ca = DeviceColumn(be, abi.Uint32, keys[pick]), DeviceColumn(be, abi.Uint32, amount)
        q.run({"fk": cf.vp, "amount": ca.vp}, n)
        cf.free(); ca.free()
        sums += np.bincount(pick, weights=amount, minlength=nkeys).astype(np.int64)
    dims, valids, meas = q.fetch()
    g = q.result_size
    present = np.zeros(nkeys, bool)
    present[usable] = True  # every usable key is drawn many times (n >> nkeys)
    assert g == int(present.sum())
    got_fk, got_attr, got_sum = dims[0].view(np.uint32), dims[1].view(np.uint32), meas.view(np.uint32)
    assert valids[0].all() and valids[1].all()
    order = np.argsort(got_fk)
    want_keys = np.sort(keys[present])
    assert np.array_equal(got_fk[order], want_keys)
    key_to_i = {int(k): i for i, k in enumerate(keys)}
    ii = np.array([key_to_i[int(k)] for k in got_fk[order]])
    assert np.array_equal(got_attr[order], attr[ii])
    assert np.array_equal(got_sum[order].astype(np.int64), sums[ii])
    q.release()
    for b in [tb] + dcols:
        b.free()
