"""Sort + Reduce without sorting rows (aresdb_amd/csrc/algo/sort_reduce_fused.hip): the Go host's call sequence of the
reference's DEFAULT aggregation path (query/aql_batchexecutor.go:236-251: InitIndexVector over previous result + batch,
Sort, Reduce — every aggregate but SUM_SIGNED / SUM_FLOAT, and those too unless enable_hash_reduction is set) replayed at
the ABI, batch after batch with the result buffers ping-ponged as query/aql_processor.go:718-724 does.  The HIP library
DEFINES Sort's outputs while the batch's transforms are still pending and Reduce aggregates by the 64-bit row hash; what a
host can observe must be what the oracle (and the reference build) leave: the groups in ascending hash order, bit for bit
— and, for a host that does look, the hash and index vectors between and after the two calls (materialised on demand)."""
import os

import numpy as np
import pytest

import harness as H
from aresdb_amd import abi

pytestmark = pytest.mark.gpu

U32 = (0, 0, 1, 0, 0)


def _kernels_of(hip, fn):
    hip.profiler_enable(True)
    try:
        res = fn()
        hip.wait()
        return res, hip.profiler_report()
    finally:
        hip.profiler_enable(False)


def _fusion_on():
    return all(os.environ.get(k, "1") != "0" for k in ("ARES_FUSE", "ARES_DEFER", "ARES_SORT_FUSE", "ARES_RTC"))


class Shape:
    """One query shape: columns, filters (column, functor, constant), dimensions (column, functor or None, constant, output
    type), measure (None = COUNT(*): the literal 1; else a column), aggregate, value bytes."""

    def __init__(self, name, cols, filters, dims, measure, agg, value_type, ndw):
        self.name, self.cols, self.filters, self.dims, self.measure, self.agg, self.value_type, self.ndw = \
            name, cols, filters, dims, measure, agg, value_type, ndw
        self.value_bytes = abi.DATA_TYPE_BYTES[value_type]


def make_batch(rng, shape, n, null_fraction=0.03):
    cols = {}
    for name, (dtype, hi) in shape.cols.items():
        np_t = {abi.Uint32: np.uint32, abi.Int32: np.int32, abi.Uint16: np.uint16, abi.Uint8: np.uint8, abi.Int16: np.int16}[dtype]
        lo = -hi if dtype in (abi.Int32, abi.Int16) else 0
        vals = rng.integers(lo, hi, n).astype(np_t)
        valid = (rng.random(n) >= null_fraction) if null_fraction > 0 else None
        cols[name] = (dtype, vals, valid)
    return cols


def _column(b, dtype, vals, valid, rle=False):
    """A batch column on backend b: plain (mode 1 / 2) or — rle — run-length encoded (mode 3: counts, one validity bit and one
    value per run; query/iterator.hpp:117-126), the way an archive batch stores a sorted column."""
    if not rle:
        return H.Column(b, dtype, vals, valid=valid)
    ok = np.ones(len(vals), bool) if valid is None else valid
    cut = np.flatnonzero((vals[1:] != vals[:-1]) | (ok[1:] != ok[:-1])) + 1
    starts = np.concatenate([[0], cut])
    counts = np.concatenate([starts, [len(vals)]]).astype(np.uint32)
    return H.Column(b, dtype, vals[starts], valid=ok[starts], counts=counts)


SHAPES = [
    # C3's dimensions, COUNT(*) — the shape of the reference's first example query (examples/1k_trips/queries/total_trips.aql)
    Shape("count_c3", {"ts": (abi.Uint32, 86400 * 7), "d1": (abi.Uint32, 100), "d2": (abi.Uint32, 50), "d3": (abi.Uint32, 2)},
          [("d1", abi.LessThan, 90)],
          [("ts", abi.Floor, 3600, abi.Uint32), ("d1", None, 0, abi.Uint32), ("d2", None, 0, abi.Uint32), ("d3", None, 0, abi.Uint32)],
          None, abi.AGGR_SUM_UNSIGNED, abi.Uint32, (0, 0, 4, 0, 0)),
    # enable_hash_reduction = false: SUM over an unsigned column goes through Sort + Reduce into 8 bytes (Int64 output type)
    Shape("sum8_two_dims", {"ts": (abi.Uint32, 86400), "d1": (abi.Uint32, 40), "m": (abi.Uint32, 1000)},
          [("ts", abi.GreaterThanOrEqual, 3600), ("ts", abi.LessThan, 80000)],
          [("ts", abi.Floor, 600, abi.Uint32), ("d1", None, 0, abi.Uint32)],
          "m", abi.AGGR_SUM_UNSIGNED, abi.Int64, (0, 0, 2, 0, 0)),
    # the trips schema: city_id Uint16 in a 2-byte slot, status Uint8 filter (examples/1k_trips/schema/trips.json), COUNT(*)
    Shape("count_trips", {"request_at": (abi.Uint32, 86400 * 3), "city_id": (abi.Uint16, 300), "status": (abi.Uint8, 4)},
          [("request_at", abi.GreaterThanOrEqual, 1000), ("request_at", abi.LessThan, 200000), ("status", abi.Equal, 2)],
          [("request_at", abi.Floor, 3600, abi.Uint32), ("city_id", None, 0, abi.Uint16)],
          None, abi.AGGR_SUM_UNSIGNED, abi.Uint32, (0, 0, 1, 1, 0)),
    # MIN / MAX of a signed column, one dimension, no filter at all
    Shape("min_signed", {"d1": (abi.Uint32, 3000), "m": (abi.Int32, 100000)}, [],
          [("d1", None, 0, abi.Uint32)], "m", abi.AGGR_MIN_SIGNED, abi.Int32, (0, 0, 1, 0, 0)),
    Shape("max_unsigned", {"d1": (abi.Uint32, 7), "d2": (abi.Uint32, 9), "m": (abi.Uint32, 1 << 30)}, [("d2", abi.NotEqual, 3)],
          [("d1", abi.Plus, 5, abi.Uint32), ("d2", None, 0, abi.Uint32)], "m", abi.AGGR_MAX_UNSIGNED, abi.Uint32, (0, 0, 2, 0, 0)),
]


def run_sequence(b, shape, batches, read=frozenset(), cap_slack=10, hash_reduce=False, eager=False):
    """The Go host's per-batch sequence; returns every observable the test compares.  read: any of "iota" (the index
    vector between InitIndexVector and Sort), "sorted" (hash + index vector between Sort and Reduce), "after" (input hash /
    index vector and output index vector after Reduce), "inputs" (input dimension and measure rows after Reduce).
    eager: the host looks at the batch's dimension and measure rows before it sorts (so they are written: what a joined column
    or a generic expression leads to as well)."""
    cap = sum(len(next(iter(bt.values()))[1]) for bt in batches) + cap_slack
    vb = shape.value_bytes
    dv = [H.DimVector(b, cap, shape.ndw, True, False) for _ in range(2)]   # dimension + hash vector pairs: swapped per batch
    iv = [H.Buf(b, nbytes=4 * cap) for _ in range(2)]                      # dimIndexVectorD: NOT swapped
    vv = [H.Buf(b, nbytes=vb * cap) for _ in range(2)]
    res = 0
    log = []
    vtype = {abi.Uint32: np.uint32, abi.Int32: np.int32, abi.Int64: np.int64}[shape.value_type]

    def vec(i, index):
        s = dv[i].struct()
        s.IndexVector = index.ptr
        return s

    for bt in batches:
        n = len(next(iter(bt.values()))[1])
        cols = {k: _column(b, *spec) for k, spec in bt.items()}
        idx, pred = H.Buf(b, nbytes=4 * n), H.Buf(b, nbytes=n)
        b.call("InitIndexVector", idx.ptr, 0, n, None, 0)
        kept = n
        for col, ft, k in shape.filters:
            kept = b.call("BinaryFilter", cols[col].input(), H.const_int(k), idx.ptr, pred.ptr, kept, None, 0, None, 0, ft, None, 0)
        offs = dv[0].dim_offsets()
        # dimensions in vector (descending width) order = the order of shape.dims here
        for d, (col, ft, k, otype) in enumerate(shape.dims):
            out = H.dimension_output(dv[0].values.ptr + offs[d][0] + offs[d][2] * res, dv[0].values.ptr + offs[d][1] + res, otype)
            if kept <= 0:
                continue
            if ft is None:
                b.call("UnaryTransform", cols[col].input(), out, idx.ptr, kept, None, 0, abi.Noop, None, 0)
            else:
                b.call("BinaryTransform", cols[col].input(), H.const_int(k), out, idx.ptr, kept, None, 0, ft, None, 0)
        if kept > 0:
            mout = H.measure_output(vv[0].ptr + vb * res, shape.value_type, shape.agg)
            if shape.measure is None:
                b.call("UnaryTransform", H.const_int(1), mout, idx.ptr, kept, None, 0, abi.Noop, None, 0)
            else:
                b.call("UnaryTransform", cols[shape.measure].input(), mout, idx.ptr, kept, None, 0, abi.Noop, None, 0)
        b.wait()
        if eager and kept > 0:
            dv[0].rows(res + kept), vv[0].read(vtype, res + kept)
        for c in cols.values():  # cleanupBeforeAggregation
            c.free()
        idx.free(), pred.free()
        length = res + kept
        entry = {"kept": kept}
        kin, kout = vec(0, iv[0]), vec(1, iv[1])
        if hash_reduce:  # (enable_hash_reduction: the same buffers through HashReduce; the order of its output is arbitrary)
            groups = b.call("HashReduce", kin, vv[0].ptr, kout, vv[1].ptr, vb, length, shape.agg, None, 0)
            b.wait()
            entry["groups"] = groups
            entry["table"] = dict(zip(dv[1].rows(groups), (int(x) for x in vv[1].read(vtype, groups))))
            log.append(entry)
            res = groups
            dv[0], dv[1] = dv[1], dv[0]
            vv[0], vv[1] = vv[1], vv[0]
            continue
        b.call("InitIndexVector", iv[0].ptr, 0, length, None, 0)
        if "iota" in read:
            entry["iota"] = iv[0].read(np.uint32, length)
        b.call("Sort", kin, length, None, 0)
        if "sorted" in read:
            entry["hash"] = dv[0].hash.read(np.uint64, length)
            entry["index"] = iv[0].read(np.uint32, length)
        groups = b.call("Reduce", kin, vv[0].ptr, kout, vv[1].ptr, vb, length, shape.agg, None, 0)
        b.wait()
        entry["groups"] = groups
        if "after" in read:
            entry["hash_after"] = dv[0].hash.read(np.uint64, length)
            entry["index_after"] = iv[0].read(np.uint32, length)
            entry["out_index"] = iv[1].read(np.uint32, groups)
        if "inputs" in read:
            entry["in_rows"] = dv[0].rows(length)
            entry["in_values"] = vv[0].read(vtype, length)
        entry["rows"] = dv[1].rows(groups)
        entry["values"] = vv[1].read(vtype, groups)
        log.append(entry)
        res = groups
        dv[0], dv[1] = dv[1], dv[0]
        vv[0], vv[1] = vv[1], vv[0]
    for x in dv + iv + vv:
        x.free()
    return log


def assert_same(got, want, what):
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        assert set(g) == set(w), (what, k)
        for key in w:
            if isinstance(w[key], np.ndarray):
                assert np.array_equal(g[key], w[key]), (what, "batch", k, key, g[key][:8], w[key][:8])
            else:
                assert g[key] == w[key], (what, "batch", k, key)


@pytest.mark.parametrize("shape", SHAPES, ids=[s.name for s in SHAPES])
def test_sort_reduce_consumes_pending_transforms(shape):
    """Four batches (one of them a single row, one large enough for several partitions); the output rows IN ORDER, their
    values and the group counts are the oracle's; on the HIP side nothing is sorted and no transform is launched."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(sum(map(ord, shape.name)))
    batches = [make_batch(rng, shape, n) for n in (5000, 1, 40000, 700)]
    got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape, batches))
    want = run_sequence(oracle, shape, batches)
    assert_same(got, want, shape.name)
    if _fusion_on():
        assert any(k.startswith("sr_scan_rtc") for k in kernels) and any(k.startswith("sr_merge_kernel") for k in kernels), sorted(kernels)
        if all(e["kept"] > 0 for e in got):  # (a batch without survivors queues no transforms: Sort + Reduce run over the previous result)
            assert not any(k.startswith(("radix_pass_kernel", "transform_", "reduce_kernel", "filter_pred")) for k in kernels), sorted(kernels)


@pytest.mark.parametrize("read", [("iota",), ("sorted",), ("after",), ("inputs",), ("sorted", "after"), ("iota", "sorted", "after", "inputs")],
                         ids=lambda r: "+".join(r))
@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[2], SHAPES[1]], ids=lambda s: s.name)
def test_lazily_defined_sort_materialises_for_a_host_that_looks(shape, read):
    """A host that reads the index vector before Sort, the hash / index vector between Sort and Reduce, or — after Reduce — the
    input's hash / index vector, the output's index vector (the groups' representatives) or the input's dimension and measure
    rows, sees what the reference leaves there: the definition is run (or replayed) on demand."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(99 + len(read))
    batches = [make_batch(rng, shape, n) for n in (3000, 9000, 50)]
    got = run_sequence(hip, shape, batches, read=frozenset(read))
    want = run_sequence(oracle, shape, batches, read=frozenset(read))
    assert_same(got, want, (shape.name, read))


def make_archive_batch(rng, n):
    """An archive batch sorted by (ts, d3): both run-length encoded (ts in runs of ~40 rows, nulls in runs too), the other
    columns plain."""
    ts = np.sort(rng.integers(0, 86400 * 2, max(1, n // 40)).astype(np.uint32))[rng.integers(0, max(1, n // 40), n)]
    ts.sort()
    d3 = ((np.arange(n) // 23) % 3).astype(np.uint32)
    ok_ts = np.repeat(rng.random((n + 99) // 100) >= 0.04, 100)[:n]
    ok_d3 = np.repeat(rng.random((n + 22) // 23) >= 0.03, 23)[:n]
    ts = np.where(ok_ts, ts, 0).astype(np.uint32)
    plain = lambda hi: (rng.integers(0, hi, n).astype(np.uint32), rng.random(n) >= 0.03)
    d1, o1 = plain(100)
    d2, o2 = plain(50)
    m, om = plain(1000)
    return {"ts": (abi.Uint32, ts, ok_ts, True), "d3": (abi.Uint32, d3, ok_d3, True), "d1": (abi.Uint32, d1, o1), "d2": (abi.Uint32, d2, o2),
            "m": (abi.Uint32, m, om)}


_ARCHIVE_FILTERS = [("ts", abi.GreaterThanOrEqual, 3600), ("ts", abi.LessThan, 150000), ("d1", abi.LessThan, 90)]
_ARCHIVE_DIMS = [("ts", abi.Floor, 3600, abi.Uint32), ("d1", None, 0, abi.Uint32), ("d2", None, 0, abi.Uint32), ("d3", None, 0, abi.Uint32)]
ARCHIVE_SUM = Shape("archive_sum", {}, _ARCHIVE_FILTERS, _ARCHIVE_DIMS, "m", abi.AGGR_SUM_UNSIGNED, abi.Uint32, (0, 0, 4, 0, 0))
ARCHIVE_COUNT = Shape("archive_count", {}, _ARCHIVE_FILTERS, _ARCHIVE_DIMS, None, abi.AGGR_SUM_UNSIGNED, abi.Uint32, (0, 0, 4, 0, 0))


@pytest.mark.parametrize("hash_reduce", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_archive_batches_decode_run_length_columns_once_and_stay_on_the_fast_path(hash_reduce):
    """Archive batches (query/aql_processor.go:571-625): the sort columns arrive run-length encoded (mode 3).  The time
    filters and a dimension read ts, another dimension reads d3: each is decoded ONCE per batch into a stream temporary
    (expand_runs_kernel) and the row-space filters, the queued transforms and the fused scan read the copy — nothing is
    materialised, the result is the oracle's (ordered, on the Sort + Reduce path)."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(71)
    batches = [make_archive_batch(rng, n) for n in (30000, 45000, 1200)]
    shape = ARCHIVE_SUM if hash_reduce else ARCHIVE_COUNT
    want = run_sequence(oracle, shape, batches, hash_reduce=hash_reduce)
    for attempt in range(2):
        got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape, batches, hash_reduce=hash_reduce))
        assert_same(got, want, shape.name)
    if _fusion_on():
        assert kernels["expand_runs_kernel"][0] == 2 * len(batches), kernels.get("expand_runs_kernel")  # ts and d3, once per batch
        assert any(k.startswith(("hr_scan_rtc", "hr_table_scan_rtc", "sr_scan_rtc", "hr_fused_scan")) for k in kernels), sorted(kernels)
        assert not any(k.startswith(("transform", "filter_pred", "filter_kernel", "radix_pass")) for k in kernels), sorted(kernels)


# MAX_DIMENSIONS = 8 (query/time_series_aggregate.h:36-37): six 4-byte slots, a 2-byte and a 1-byte one, in vector order
# (widest first: query/aql_compiler.go:1341-1362), every column with nulls, a filter on a column no dimension reads
_EIGHT_COLS = {**{c: (abi.Uint32, 3) for c in "abcdef"}, "g": (abi.Uint16, 3), "h": (abi.Uint8, 2), "m": (abi.Uint32, 1000), "k": (abi.Uint32, 10)}
_EIGHT_DIMS = [("a", abi.Floor, 2, abi.Uint32)] + [(c, None, 0, abi.Uint32) for c in "bcdef"] + [("g", None, 0, abi.Uint16), ("h", None, 0, abi.Uint8)]
EIGHT_SUM = Shape("eight_dims_sum", _EIGHT_COLS, [("k", abi.LessThan, 8)], _EIGHT_DIMS, "m", abi.AGGR_SUM_UNSIGNED, abi.Uint32, (0, 0, 6, 1, 1))
EIGHT_COUNT = Shape("eight_dims_count", _EIGHT_COLS, [("k", abi.LessThan, 8)], _EIGHT_DIMS, None, abi.AGGR_SUM_UNSIGNED, abi.Uint32, (0, 0, 6, 1, 1))


def test_eight_dimensions_through_sort_reduce():
    """The ABI's limit of dimensions on the fused Sort + Reduce path: ordered output, bit for bit."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(88)
    batches = [make_batch(rng, EIGHT_COUNT, n) for n in (20000, 30000, 900)]
    for shape in (EIGHT_COUNT, EIGHT_SUM):
        got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape, batches, read=frozenset(("after",))))
        want = run_sequence(oracle, shape, batches, read=frozenset(("after",)))
        assert_same(got, want, shape.name)
        got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape, batches))
        if _fusion_on():
            assert any(k.startswith("sr_scan_rtc") for k in kernels) and not any(k.startswith(("radix_pass", "transform_")) for k in kernels), sorted(kernels)


# ... and eight 4-byte slots (the all-4-byte shortcuts of the generators pack four validity bytes into a word: not beyond four)
_ALL4_COLS = {**{c: (abi.Uint32, 3) for c in "abcdefgh"}, "m": (abi.Uint32, 1000), "k": (abi.Uint32, 10)}
EIGHT_ALL4 = Shape("eight_u32_dims_sum", _ALL4_COLS, [("k", abi.LessThan, 8)], [(c, None, 0, abi.Uint32) for c in "abcdefgh"], "m",
                   abi.AGGR_SUM_UNSIGNED, abi.Uint32, (0, 0, 8, 0, 0))


@pytest.mark.parametrize("shape8", [EIGHT_SUM, EIGHT_ALL4], ids=lambda s: s.name)
@pytest.mark.parametrize("lean", [False, True], ids=["table_scan", "direct_scan"])
def test_eight_dimensions_through_hash_reduce(lean, shape8, monkeypatch):
    """... and on the fused HashReduce path (kFusedDims 8, kFusedCols 10: round 6): more than four dimensions run on the
    kernels generated for the plan's shape only — the kernel log proves that nothing was materialised."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(89)
    batches = [make_batch(rng, shape8, n) for n in (20000, 30000, 900, 25000)]
    want = run_sequence(oracle, shape8, batches, hash_reduce=True)
    if lean:
        monkeypatch.setenv("ARES_LEAN_MIN_GROUPS", "0")
        monkeypatch.setenv("ARES_MIN_PART_BITS", "2")
    hip.reload_env()
    try:
        for attempt in range(2):  # (the first pass of a shape may meet kernels that are still to be generated)
            got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape8, batches, hash_reduce=True))
            assert_same(got, want, "eight_dims_hash")
    finally:
        monkeypatch.undo()
        hip.reload_env()
    if _fusion_on():
        assert any(k.startswith(("hr_scan_rtc", "hr_table_scan_rtc")) for k in kernels) and any(k.startswith("hr_merge_rtc") for k in kernels), sorted(kernels)
        # (hr_partition_kernel — layout-generic — only re-partitions PREVIOUS groups that are not grouped by partition yet)
        assert not any(k.startswith(("transform_", "hr_partition4", "hr_fused")) for k in kernels), sorted(kernels)


def test_many_groups_per_partition_fall_back_to_the_real_sort(monkeypatch):
    """More groups than a partition's table orders (ARES_SR_MAX_GROUPS lowers the limit to 100 here): the first batch overflows
    inside the merge, the second is declined up front (the previous result alone is too large) — either way the ordinary
    Sort + Reduce runs over the same buffers and the result is the oracle's."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    shape = Shape("distinct", {"d1": (abi.Uint32, 1 << 30)}, [], [("d1", None, 0, abi.Uint32)], None, abi.AGGR_SUM_UNSIGNED, abi.Uint32, U32)
    rng = np.random.default_rng(5)
    batches = [make_batch(rng, shape, n, null_fraction=0) for n in (30000, 30000)]
    monkeypatch.setenv("ARES_SR_MAX_GROUPS", "100")
    hip.reload_env()
    try:
        got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape, batches, read=frozenset(("after",))))
    finally:
        monkeypatch.delenv("ARES_SR_MAX_GROUPS")
        hip.reload_env()
    want = run_sequence(oracle, shape, batches, read=frozenset(("after",)))
    assert_same(got, want, "distinct")
    assert any(k.startswith("radix_pass_kernel") for k in kernels), sorted(kernels)


def test_float_sums_keep_the_real_sort():
    """SUM over a float measure depends on the order of additions (ascending hash, then row): never consumed."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(8)
    n = 6000
    d1 = rng.integers(0, 50, n).astype(np.uint32)
    m = (rng.integers(0, 400, n) / 4).astype(np.float32)

    def seq(b):
        cap = n + 10
        cols = {"d1": H.Column(b, abi.Uint32, d1), "m": H.Column(b, abi.Float32, m)}
        idx = H.Buf(b, nbytes=4 * n)
        din, dout = H.DimVector(b, cap, U32), H.DimVector(b, cap, U32)
        vin, vout = H.Buf(b, nbytes=8 * cap), H.Buf(b, nbytes=8 * cap)
        b.call("InitIndexVector", idx.ptr, 0, n, None, 0)
        offs = din.dim_offsets()
        b.call("UnaryTransform", cols["d1"].input(), H.dimension_output(din.values.ptr + offs[0][0], din.values.ptr + offs[0][1], abi.Uint32),
               idx.ptr, n, None, 0, abi.Noop, None, 0)
        b.call("UnaryTransform", cols["m"].input(), H.measure_output(vin.ptr, abi.Float64, abi.AGGR_SUM_FLOAT), idx.ptr, n, None, 0, abi.Noop, None, 0)
        b.wait()
        b.call("InitIndexVector", din.index.ptr, 0, n, None, 0)
        b.call("Sort", din.struct(), n, None, 0)
        g = b.call("Reduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, 8, n, abi.AGGR_SUM_FLOAT, None, 0)
        b.wait()
        out = (g, dout.rows(g), vout.read(np.float64, g))
        for x in list(cols.values()) + [idx, din, dout, vin, vout]:
            x.free()
        return out

    got, want = seq(hip), seq(oracle)
    assert got[0] == want[0] and got[1] == want[1] and np.array_equal(got[2], want[2])


# ---- Sort + Reduce over rows that exist (fused_sort_reduce_vectors: the wide layout) -----------------------------------------
_VECTOR_SHAPES = [SHAPES[0], SHAPES[3], SHAPES[4], SHAPES[1], SHAPES[2], EIGHT_COUNT, EIGHT_SUM,  # (8-byte values; 2- and 1-byte slots)
                  Shape("eight_u32_dims_count", _ALL4_COLS, [("k", abi.LessThan, 8)], [(c, None, 0, abi.Uint32) for c in "abcdefgh"], None,
                        abi.AGGR_SUM_UNSIGNED, abi.Uint32, (0, 0, 8, 0, 0))]


@pytest.mark.parametrize("part_bits", [None, 7, 12], ids=["default_bits", "128_partitions", "4096_partitions"])
@pytest.mark.parametrize("shape", _VECTOR_SHAPES, ids=[s.name for s in _VECTOR_SHAPES])
def test_sort_reduce_over_materialised_rows_orders_groups_not_rows(shape, part_bits, monkeypatch):
    """The batch's rows were written before Sort (an eager host; a joined column): Sort is defined all the same and Reduce
    aggregates by row hash in the wide layout — level-1 scan, split into 2^bits partitions (forced here: one level, and two
    levels with a fan-out of 8), small tables, the previous result taken from the row hashes kept beside it.  Ordered output,
    bit for bit; no radix pass, no reduce kernel."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(sum(map(ord, shape.name)) + 7)
    batches = [make_batch(rng, shape, n) for n in (5000, 1, 40000, 700)]
    want = run_sequence(oracle, shape, batches, eager=True)
    if part_bits is not None:
        monkeypatch.setenv("ARES_SRV_PART_BITS", str(part_bits))
    hip.reload_env()
    try:
        got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape, batches, eager=True))
    finally:
        monkeypatch.undo()
        hip.reload_env()
    assert_same(got, want, shape.name)
    if _fusion_on() and os.environ.get("ARES_SORT_VECTORS", "1") != "0":
        assert any(k.startswith("sr_split_kernel") for k in kernels) and any(k.startswith("sr_merge_kernel") for k in kernels), sorted(kernels)
        assert any(k.startswith("sr_bounds_kernel") for k in kernels), sorted(kernels)  # (the previous result was recognised)
        # (63 groups over 512 forced level-1 partitions: a few streams take everything and overflow — that batch is sorted for real)
        if not (shape.name == "max_unsigned" and part_bits == 12):
            assert not any(k.startswith(("radix_pass_kernel", "reduce_kernel")) for k in kernels), sorted(kernels)


@pytest.mark.parametrize("read", [("sorted",), ("after",), ("inputs",), ("iota", "sorted", "after", "inputs")], ids=lambda r: "+".join(r))
def test_sort_over_materialised_rows_materialises_for_a_host_that_looks(read):
    hip, oracle = H.hip_backend(), H.oracle_backend()
    shape = SHAPES[0]
    rng = np.random.default_rng(199 + len(read))
    batches = [make_batch(rng, shape, n) for n in (3000, 9000, 50)]
    got = run_sequence(hip, shape, batches, read=frozenset(read), eager=True)
    want = run_sequence(oracle, shape, batches, read=frozenset(read), eager=True)
    assert_same(got, want, (shape.name, read))


def test_materialised_rows_with_too_many_groups_fall_back_to_the_real_sort(monkeypatch):
    """A partition's table overflows (ARES_SR_MAX_GROUPS = 20, one partition forced): the ordinary Sort + Reduce runs over the
    same buffers."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    shape = Shape("distinct", {"d1": (abi.Uint32, 1 << 30)}, [], [("d1", None, 0, abi.Uint32)], None, abi.AGGR_SUM_UNSIGNED, abi.Uint32, U32)
    rng = np.random.default_rng(6)
    batches = [make_batch(rng, shape, n, null_fraction=0) for n in (600, 500)]
    want = run_sequence(oracle, shape, batches, read=frozenset(("after",)), eager=True)
    monkeypatch.setenv("ARES_SR_MAX_GROUPS", "20")
    monkeypatch.setenv("ARES_SRV_PART_BITS", "1")
    hip.reload_env()
    try:
        got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape, batches, read=frozenset(("after",)), eager=True))
    finally:
        monkeypatch.undo()
        hip.reload_env()
    assert_same(got, want, "fallback")
    assert any(k.startswith("radix_pass_kernel") or k.startswith("sort_") for k in kernels), sorted(kernels)


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[1], SHAPES[2]], ids=lambda s: s.name)
def test_scan_fed_path_declining_hands_over_to_the_wide_layout(shape, monkeypatch):
    """What the scan-fed path declines (ARES_SR_SCAN_FED=0 here; in production: a previous result of more groups than its 512
    tables order) is not sorted row by row either: the pending transforms are launched and the groups ordered over the rows
    they wrote.  Ordered output, and — for a host that looks — the hash / index vectors a sort would have left."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(41)
    batches = [make_batch(rng, shape, n) for n in (6000, 20000, 300)]
    want = run_sequence(oracle, shape, batches)
    want_read = run_sequence(oracle, shape, batches, read=frozenset(("sorted", "after")))
    monkeypatch.setenv("ARES_SR_SCAN_FED", "0")
    hip.reload_env()
    try:
        got, kernels = _kernels_of(hip, lambda: run_sequence(hip, shape, batches))
        got_read = run_sequence(hip, shape, batches, read=frozenset(("sorted", "after")))
    finally:
        monkeypatch.undo()
        hip.reload_env()
    assert_same(got, want, shape.name)
    assert_same(got_read, want_read, shape.name)
    if _fusion_on() and os.environ.get("ARES_SORT_VECTORS", "1") != "0":
        assert any(k.startswith("sr_split_kernel") for k in kernels) and not any(k.startswith("radix_pass_kernel") for k in kernels), sorted(kernels)


def _temp_stats(b):
    import ctypes as C
    out, cached = C.c_size_t(0), C.c_size_t(0)
    b._algo.AresTempStats.argtypes, b._algo.AresTempStats.restype = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)], None
    b._algo.AresTempStats(C.byref(out), C.byref(cached))
    return out.value, cached.value


@pytest.mark.parametrize("hash_reduce", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_decoded_columns_do_not_pile_up(hash_reduce):
    """Regression (round 6): a consumed queue kept its list of decoded run-length columns, so every archive batch of a stream's
    life stayed allocated (two 300 MB buffers per 64 Mi-row batch: 86 GB after 143 batches, and a hipMalloc per decode).  The
    library's stream temporaries that are handed out must not grow with the number of batches."""
    hip = H.hip_backend()
    rng = np.random.default_rng(72)
    shape = ARCHIVE_SUM if hash_reduce else ARCHIVE_COUNT
    few = [make_archive_batch(rng, 20000) for _ in range(3)]
    many = [make_archive_batch(rng, 20000) for _ in range(24)]
    run_sequence(hip, shape, few, hash_reduce=hash_reduce)
    hip.wait()
    base, _ = _temp_stats(hip)
    run_sequence(hip, shape, many, hash_reduce=hash_reduce)
    hip.wait()
    after, _ = _temp_stats(hip)
    # (a decoded column of 20 000 rows is 80 KB + validity: 24 batches x 2 columns piling up would be ~4 MB)
    assert after <= base + (1 << 20), (base, after)
