"""Query-level replay: the Go batch executor's call sequence (aresdb_amd/executor.py) against
every backend; results must agree with the oracle and, independently, with a numpy group-by."""
import os

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


# ---- the C++ host driver (libaresdriver.so) must issue the same calls and produce the same result ----
@pytest.mark.parametrize("use_hash", [True, False])
def test_native_driver_matches_python_executor(be, use_hash):
    rng = np.random.default_rng(21)
    data = [smoke.synth_batch(rng, 4000, null_fraction=0.02) for _ in range(3)]
    plan = smoke.c3_plan(use_hash)
    want, calls_py = smoke.run_query(be, plan, data)
    got, calls_cc = smoke.run_query_native(be, plan, data)
    smoke.compare_results(got, want)
    assert calls_cc == calls_py  # one ABI call per AST node per batch, exactly like the Go host


def test_native_driver_inner_expressions(be):
    rng = np.random.default_rng(6)
    n = 3000
    ts = rng.integers(0, 1 << 20, n).astype(np.uint32)
    d1 = rng.integers(0, 9, n).astype(np.uint32)
    plan = QueryPlan(
        filters=[Binary(abi.Equal, Binary(abi.Mod, Col("ts"), Const(7), abi.Int32), Const(3)),
                 Unary(abi.IsNotNull, Col("d1"))],
        dimensions=[DimensionSpec(Binary(abi.Plus, Binary(abi.Multiply, Col("d1"), Const(3), abi.Int32), Const(1)), abi.Int32),
                    DimensionSpec(Col("d1"), abi.Uint32)],
        measure=Const(1), agg=abi.AGGR_SUM_UNSIGNED, measure_type=abi.Uint32, use_hash_reduction=False)
    valid = rng.random(n) > 0.1
    batches = [({"ts": (abi.Uint32, ts), "d1": (abi.Uint32, d1)}, {"ts": None, "d1": valid})] * 2
    want, calls_py = smoke.run_query(be, plan, batches)
    got, calls_cc = smoke.run_query_native(be, plan, batches)
    assert {k: int(v) for k, v in got.items()} == {k: int(v) for k, v in want.items()}
    assert calls_cc == calls_py


# ---- HyperLogLog queries (countdistincthll): query/aql_batchexecutor.go:221-233 -------------------------
def _hll_plan():
    return QueryPlan(filters=[Binary(abi.LessThan, Col("d1"), Const(90))],
                     dimensions=[DimensionSpec(Binary(abi.Floor, Col("ts"), Const(86400)), abi.Uint32),
                                 DimensionSpec(Col("d3"), abi.Uint32)],
                     measure=Unary(abi.GetHLLValue, Col("user")), agg=abi.AGGR_HLL, measure_type=abi.Uint32)


def _hll_batches(rng, sizes, null_fraction=0.02):
    out = []
    for n in sizes:
        cols, valid = smoke.synth_batch(rng, n, null_fraction=null_fraction)
        cols["user"] = (abi.Uint32, rng.integers(0, 30000, n).astype(np.uint32))
        valid["user"] = (rng.random(n) >= null_fraction) if null_fraction else None
        out.append((cols, valid))
    return out


def _numpy_hll(batches):
    """Registers per group, computed independently: murmur3_x64_128 of the 4 value bytes, register =
    low 14 bits, rho = trailing zeros of the rest (query/functor.hpp:431-466); a null user counts as
    value 0 / rho 0 into register 0 (query/utils.hpp:169-184: the HLL identity is 0)."""
    import ctypes
    lib = ctypes.CDLL(H.ORACLE_SO)
    lib.oracle_murmur3_128.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
    h = (ctypes.c_uint64 * 2)()
    out = {}
    for cols, valid in batches:
        ts, d1, d3, user = (cols[k][1] for k in ("ts", "d1", "d3", "user"))
        n = len(ts)
        ok = {k: (np.ones(n, bool) if valid[k] is None else valid[k]) for k in cols}
        for i in np.nonzero((d1 < 90) & ok["d1"])[0]:
            key = ((np.uint32(ts[i] - ts[i] % 86400 if ok["ts"][i] else 0).tobytes(), int(ok["ts"][i])),
                   (np.uint32(d3[i]).tobytes(), int(ok["d3"][i])))
            reg, rho = 0, 0
            if ok["user"][i]:
                lib.oracle_murmur3_128(np.uint32(user[i]).tobytes(), 4, 0, h)
                reg = h[0] & 0x3FFF
                rest = h[0] >> 14
                while rho + 14 < 32 and not (rest >> rho) & 1:  # the reference tests a 32-bit mask
                    rho += 1
            g = out.setdefault(key, {})
            g[reg] = max(g.get(reg, 0), rho + 1)
    return {k: sorted(v.items()) for k, v in out.items()}


def test_hll_query_matches_independent_registers(be):
    rng = np.random.default_rng(31)
    data = _hll_batches(rng, [3000, 0, 2500, 1200])
    got, _ = smoke.run_hll_query(be, _hll_plan(), data)
    assert got == _numpy_hll(data)


def test_hll_query_native_driver_matches_python_executor(be):
    rng = np.random.default_rng(32)
    data = _hll_batches(rng, [4000, 4000, 100])
    want, calls_py = smoke.run_hll_query(be, _hll_plan(), data)
    got, calls_cc = smoke.run_hll_query(be, _hll_plan(), data, native=True)
    assert got == want
    assert calls_cc == calls_py


# ---- geo intersection queries (geography_intersects join): query/aql_batchexecutor.go:146-196 -------------
def _geo_query(be, rng, in_or_out, with_dim):
    from aresdb_amd.executor import GeoIntersection
    # three disjoint axis-parallel-free polygons (closed rings), the second with a hole
    rings = [[(10, 10), (20, 12), (18, 22), (9, 19)],
             [(-30, -30), (-10, -28), (-8, -8), (-29, -11)], [(-22, -22), (-16, -21), (-17, -15), (-23, -16)],
             [(30, -20), (45, -18), (38, -5)]]
    shape_of_ring = [0, 1, 1, 2]
    lats, longs, sidx = [], [], []
    flt_max = float(np.finfo(np.float32).max)
    for r, (ring, sh) in enumerate(zip(rings, shape_of_ring)):
        if r and shape_of_ring[r - 1] == sh:
            lats.append(flt_max), longs.append(flt_max), sidx.append(sh)
        for la, lo in ring + [ring[0]]:
            lats.append(la), longs.append(lo), sidx.append(sh)
    shapes = H.GeoShapes(be, np.float32(lats), np.float32(longs), sidx, 3)
    dims = [DimensionSpec(Col("d3"), abi.Uint32)]
    if with_dim:
        dims.append(DimensionSpec(Const(0), abi.Uint8))  # placeholder: written by WriteGeoShapeDim
    plan = QueryPlan(filters=[Binary(abi.LessThan, Col("d1"), Const(90))], dimensions=dims, measure=Const(1),
                     agg=abi.AGGR_SUM_UNSIGNED, measure_type=abi.Uint32, use_hash_reduction=False,
                     geo=GeoIntersection(shapes.buf.ptr, 3, len(lats), "pt", 0, in_or_out, 1 if with_dim else -1))
    batches = []
    for n in (3000, 2000):
        cols, valid = smoke.synth_batch(rng, n, null_fraction=0.02)
        cols["pt"] = (abi.GeoPoint, (rng.integers(-200, 200, (n, 2)) / 4 + 0.125).astype(np.float32))
        valid["pt"] = rng.random(n) >= 0.05
        batches.append((cols, valid))

    def inside(lat, lng):  # even-odd rule in float64; the test points sit off every edge
        hit = []
        for sh in range(3):
            c = 0
            for ring, s in zip(rings, shape_of_ring):
                if s != sh:
                    continue
                for (la1, lo1), (la2, lo2) in zip(ring, ring[1:] + ring[:1]):
                    if (lo1 > lng) != (lo2 > lng) and lat < (la2 - la1) * (lng - lo1) / (lo2 - lo1) + la1:
                        c ^= 1
            if c:
                hit.append(sh)
        return hit

    want = {}
    for cols, valid in batches:
        d1, d3, pt = cols["d1"][1], cols["d3"][1], cols["pt"][1]
        for i in range(len(d1)):
            if not (valid["d1"][i] and d1[i] < 90):
                continue
            if not valid["pt"][i]:  # a null point never survives, whichever way the filter asks
                continue            # (query/iterator.hpp:1372-1381)
            hit = inside(float(pt[i, 0]), float(pt[i, 1]))
            if in_or_out != bool(hit):
                continue
            key = [(np.uint32(d3[i]).tobytes(), int(valid["d3"][i]))]
            if with_dim:
                key.append((bytes([hit[0]]), 1))
            want[tuple(key)] = want.get(tuple(key), 0) + 1
    return plan, batches, want, shapes


@pytest.mark.parametrize("in_or_out,with_dim", [(True, True), (True, False), (False, False)])
def test_geo_query_matches_independent_point_in_polygon(be, in_or_out, with_dim):
    plan, batches, want, shapes = _geo_query(be, np.random.default_rng(41), in_or_out, with_dim)
    got, calls_py = smoke.run_query(be, plan, batches)
    assert {k: int(v) for k, v in got.items()} == want
    got_cc, calls_cc = smoke.run_query_native(be, plan, batches)
    assert {k: int(v) for k, v in got_cc.items()} == want
    assert calls_cc == calls_py
    shapes.free()


def _join_fixture(be, rng, n, keep):
    """Fact table with a foreign key into a 2-batch dimension table reached through a cuckoo index
    (memstore/cuckoo_index.go layout), as prepareForeignTable uploads it (aql_processor.go:398-457)."""
    import cases
    from aresdb_amd.executor import ForeignTable
    nkeys, per_batch = 300, 200
    keys = rng.choice(100000, nkeys, replace=False).astype(np.uint32)
    region = rng.integers(0, 7, nkeys).astype(np.uint32)
    region_valid = rng.random(nkeys) > 0.1
    seeds = [int(x) for x in rng.integers(0, 1 << 32, 4)]
    table, placed = cases.build_cuckoo([int(k).to_bytes(4, "little") for k in keys], 4, 80, seeds, rng,
                                       record_of=lambda i: (1 + i // per_batch, i % per_batch))
    tb = H.Buf(be, table)
    idx = abi.CuckooHashIndex()
    idx.buckets = tb.ptr
    for i, s in enumerate(seeds):
        idx.seeds[i] = s
    idx.keyBytes, idx.numHashes, idx.numBuckets = 4, 4, 80
    cols = [H.Column(be, abi.Uint32, region[b * per_batch:(b + 1) * per_batch],
                     valid=region_valid[b * per_batch:(b + 1) * per_batch]) for b in range(2)]
    keep.extend([tb] + cols)
    ft = ForeignTable(join_column="fk", index=idx, batches={"region": [c.vp for c in cols]},
                      data_types={"region": abi.Uint32}, base_batch_id=1,
                      num_records_in_last_batch=nkeys - per_batch)
    fk = keys[rng.integers(0, nkeys, n)].copy()
    miss = rng.random(n) < 0.15
    fk[miss] = rng.integers(200000, 300000, int(miss.sum()))  # keys the dimension table does not hold
    lookup = {int(k): (int(r), bool(v)) for k, r, v in zip(keys, region, region_valid) if int(k).to_bytes(4, "little") in placed}
    return ft, fk, lookup


@pytest.mark.parametrize("native", [False, True], ids=["python", "cxx"])
def test_join_group_by_foreign_column(be, native):
    """fact JOIN dim ON fk: filter on a joined column, group by it — HashLookup, RecordID vectors
    carried through the filters, foreign-column transforms (BASELINE config C4's shape, small)."""
    rng = np.random.default_rng(31)
    n = 5000
    keep = []
    ft, fk, lookup = _join_fixture(be, rng, n, keep)
    amount = rng.integers(1, 50, n).astype(np.uint32)
    plan = QueryPlan(
        filters=[Binary(abi.GreaterThan, Col("amount"), Const(5))],
        foreign_tables=[ft],
        foreign_filters=[Binary(abi.NotEqual, Col("region", table=1), Const(2))],
        dimensions=[DimensionSpec(Col("region", table=1), abi.Uint32)],
        measure=Col("amount"), agg=abi.AGGR_SUM_UNSIGNED, measure_type=abi.Uint32, use_hash_reduction=False)
    batches = [({"fk": (abi.Uint32, fk), "amount": (abi.Uint32, amount)}, {"fk": None, "amount": None})]
    run = smoke.run_query_native if native else smoke.run_query
    got, _ = run(be, plan, batches)
    want = {}
    for k, a in zip(fk, amount):
        if a <= 5:
            continue
        r, ok = lookup.get(int(k), (0, False))
        if not (ok and r != 2):  # a null (missing or invalid) joined value fails the filter
            continue
        key = ((np.uint32(r).tobytes(), 1),)
        want[key] = want.get(key, 0) + int(a)
    assert {k: int(v) for k, v in got.items()} == want
    for b in keep:
        b.free()


# ---- fused extension (include/ares_extensions.h): one call per batch, same groups and sums -------------
def _fused_plan(variant, use_fused):
    dims_by_variant = [
        [DimensionSpec(Binary(abi.Floor, Col("ts"), Const(3600)), abi.Uint32), DimensionSpec(Col("d1"), abi.Uint32),
         DimensionSpec(Col("d2"), abi.Uint32), DimensionSpec(Col("d3"), abi.Uint32)],
        [DimensionSpec(Binary(abi.Plus, Col("d1"), Const(7)), abi.Int32), DimensionSpec(Col("d3"), abi.Uint32)],
        [DimensionSpec(Binary(abi.Mod, Col("ts"), Const(11)), abi.Int32)],
        [DimensionSpec(Col("d2"), abi.Uint32), DimensionSpec(Binary(abi.Multiply, Col("m"), Const(2.0)), abi.Float32),
         DimensionSpec(Binary(abi.Divide, Col("ts"), Const(86400)), abi.Uint32)],
    ]
    filters_by_variant = [
        [Binary(abi.LessThan, Col("d1"), Const(90))],
        [],
        [Binary(abi.GreaterThanOrEqual, Col("ts"), Const(1000)), Binary(abi.NotEqual, Col("d2"), Const(0)),
         Binary(abi.LessThan, Col("m"), Const(80.5))],
        [Binary(abi.Equal, Col("d3"), Const(1))],
    ]
    measures = [(Col("m"), abi.AGGR_SUM_FLOAT, abi.Float64), (Col("d1"), abi.AGGR_SUM_SIGNED, abi.Int64),
                (Binary(abi.Plus, Col("d2"), Const(1)), abi.AGGR_SUM_SIGNED, abi.Int32), (Col("m"), abi.AGGR_SUM_FLOAT, abi.Float64)]
    m, agg, mt = measures[variant]
    return QueryPlan(filters=filters_by_variant[variant], dimensions=dims_by_variant[variant], measure=m, agg=agg,
                     measure_type=mt, use_hash_reduction=True, use_fused_extension=use_fused)


@pytest.mark.gpu
@pytest.mark.parametrize("nulls", [0.0, 0.05], ids=["mode1", "mode2"])
@pytest.mark.parametrize("variant", range(4))
def test_fused_extension_matches_unfused_sequence(variant, nulls):
    """The fused call must produce exactly the groups (keys bit-exact) and sums of the ordinary
    per-node sequence run on the oracle, over several batches (previous results are re-reduced)."""
    hip = H.hip_backend()
    rng = np.random.default_rng(100 + variant)
    sizes = [5000, 17, 40001, 1]
    data = [smoke.synth_batch(rng, n, null_fraction=nulls) for n in sizes]
    got, calls = smoke.run_query_native(hip, _fused_plan(variant, True), data)
    assert smoke.run_query_native.last_fused_batches == len(sizes)
    assert calls == len(sizes)
    want, _ = smoke.run_query(H.oracle_backend(), _fused_plan(variant, False), data)
    smoke.compare_results(got, want)
    unfused, _ = smoke.run_query_native(hip, _fused_plan(variant, False), data)
    smoke.compare_results(unfused, want)


@pytest.mark.gpu
def test_fused_extension_declines_unsupported_plans():
    """A plan outside the fusable shape (nested expression) silently takes the ordinary sequence."""
    hip = H.hip_backend()
    rng = np.random.default_rng(9)
    data = [smoke.synth_batch(rng, 3000) for _ in range(2)]
    plan = QueryPlan(filters=[Binary(abi.Equal, Binary(abi.Mod, Col("ts"), Const(7), abi.Int32), Const(3))],
                     dimensions=[DimensionSpec(Col("d1"), abi.Uint32)], measure=Col("m"), agg=abi.AGGR_SUM_FLOAT,
                     measure_type=abi.Float64, use_hash_reduction=True, use_fused_extension=True)
    got, _ = smoke.run_query_native(hip, plan, data)
    assert smoke.run_query_native.last_fused_batches == 0
    plan.use_fused_extension = False
    want, _ = smoke.run_query(H.oracle_backend(), plan, data)
    smoke.compare_results(got, want)


# ---- second-stage fusion inside the unchanged ABI (include/ares_extensions.h) -------------------------
def _kernels_of(hip, fn):
    hip.profiler_enable(True)
    try:
        res = fn()
        hip.wait()
        return res, hip.profiler_report()
    finally:
        hip.profiler_enable(False)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", range(4))
@pytest.mark.parametrize("native", [True, False], ids=["cpp_driver", "python_mirror"])
def test_pending_transforms_are_consumed_by_hash_reduce(variant, native):
    """The Go call sequence, columns freed between project() and reduce(): the batch's dimension and
    measure transforms never launch — HashReduce evaluates them on the fly — and the result is the
    oracle's."""
    hip = H.hip_backend()
    rng = np.random.default_rng(300 + variant)
    data = [smoke.synth_batch(rng, n, null_fraction=0.03) for n in (6000, 1, 45000, 300)]
    plan = _fused_plan(variant, False)
    run = smoke.run_query_native if native else smoke.run_query
    (got, _), kernels = _kernels_of(hip, lambda: run(hip, plan, data))
    want, _ = smoke.run_query(H.oracle_backend(), plan, data)
    smoke.compare_results(got, want)
    if os.environ.get("ARES_FUSE", "1") != "0" and os.environ.get("ARES_DEFER", "1") != "0":
        assert any(k.startswith(("hr_fused_scan_kernel", "hr_scan_rtc", "hr_table_scan_rtc")) for k in kernels), kernels
        assert not any(k.startswith("transform_") for k in kernels), kernels


@pytest.mark.gpu
def test_skipped_transform_outputs_materialise_on_copy():
    """A host that copies the INPUT dimension / measure vectors back after HashReduce (the Go host
    never does) still sees what the transforms would have written: the skipped work is launched
    before the copy."""
    be, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(8)
    n = 20000
    d1 = rng.integers(0, 100, n).astype(np.uint32)
    ts = rng.integers(0, 86400 * 7, n).astype(np.uint32)
    m = (rng.integers(0, 400, n) / 4).astype(np.float32)
    valid = rng.random(n) > 0.05

    def sequence(b, read_inputs):
        cap = n + 10
        cols = {"d1": H.Column(b, abi.Uint32, d1, valid=valid), "ts": H.Column(b, abi.Uint32, ts), "m": H.Column(b, abi.Float32, m)}
        idx, pred = H.Buf(b, nbytes=4 * n), H.Buf(b, nbytes=n)
        din, dout = H.DimVector(b, cap, (0, 0, 2, 0, 0), False, False), H.DimVector(b, cap, (0, 0, 2, 0, 0), False, False)
        vin, vout = H.Buf(b, nbytes=8 * cap), H.Buf(b, nbytes=8 * cap)
        b.call("InitIndexVector", idx.ptr, 0, n, None, 0)
        kept = b.call("BinaryFilter", cols["d1"].input(), H.const_int(90), idx.ptr, pred.ptr, n, None, 0, None, 0, abi.LessThan, None, 0)
        offs = din.dim_offsets()
        b.call("BinaryTransform", cols["ts"].input(), H.const_int(3600),
               H.dimension_output(din.values.ptr + offs[0][0], din.values.ptr + offs[0][1], abi.Uint32), idx.ptr, kept, None, 0,
               abi.Floor, None, 0)
        b.call("UnaryTransform", cols["d1"].input(),
               H.dimension_output(din.values.ptr + offs[1][0], din.values.ptr + offs[1][1], abi.Uint32), idx.ptr, kept, None, 0,
               abi.Noop, None, 0)
        b.call("UnaryTransform", cols["m"].input(), H.measure_output(vin.ptr, abi.Float64, abi.AGGR_SUM_FLOAT), idx.ptr, kept,
               None, 0, abi.Noop, None, 0)
        b.wait()
        for c in cols.values():  # cleanupBeforeAggregation: columns, index vector, predicate vector
            c.free()
        idx.free(), pred.free()
        groups = b.call("HashReduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, 8, kept, abi.AGGR_SUM_FLOAT, None, 0)
        b.wait()
        res = {"kept": kept, "groups": groups}
        if read_inputs:
            res["in_dims"] = din.values.read(np.uint8)
            res["in_measures"] = vin.read(np.float64, kept)
        rows = dout.rows(groups)
        sums = vout.read(np.float64, groups)
        res["map"] = {r: float(v) for r, v in zip(rows, sums)}
        for x in (din, dout, vin, vout):
            x.free()
        return res

    (got, kernels) = _kernels_of(be, lambda: sequence(be, True))
    want = sequence(oracle, True)
    assert got["kept"] == want["kept"] and got["groups"] == want["groups"]
    assert got["map"] == want["map"]
    assert np.array_equal(got["in_dims"], want["in_dims"])
    assert np.array_equal(got["in_measures"], want["in_measures"])
    if os.environ.get("ARES_FUSE", "1") != "0" and os.environ.get("ARES_DEFER", "1") != "0":
        assert any(k.startswith(("hr_fused_scan_kernel", "hr_scan_rtc", "hr_table_scan_rtc")) for k in kernels), kernels
        assert any(k.startswith("transform_") for k in kernels), kernels


@pytest.mark.gpu
def test_filter_column_overwritten_before_reduce():
    """The host overwrites the column a filter has read (a plain H2D copy into the same buffer) before
    HashReduce: the survivors must still be those of the filter call, not a re-evaluation."""
    be, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(14)
    n = 12000
    x = rng.integers(0, 100, n).astype(np.uint32)
    y = rng.integers(0, 30, n).astype(np.uint32)
    m = rng.integers(0, 1000, n).astype(np.uint32)

    def sequence(b):
        cap = n + 8
        cx, cy, cm = H.Column(b, abi.Uint32, x), H.Column(b, abi.Uint32, y), H.Column(b, abi.Uint32, m)
        idx, pred = H.Buf(b, nbytes=4 * n), H.Buf(b, nbytes=n)
        din, dout = H.DimVector(b, cap, (0, 0, 1, 0, 0), False, False), H.DimVector(b, cap, (0, 0, 1, 0, 0), False, False)
        vin, vout = H.Buf(b, nbytes=8 * cap), H.Buf(b, nbytes=8 * cap)
        b.call("InitIndexVector", idx.ptr, 0, n, None, 0)
        kept = b.call("BinaryFilter", cx.input(), H.const_int(50), idx.ptr, pred.ptr, n, None, 0, None, 0, abi.LessThan, None, 0)
        cx.buf.write(np.zeros(n, np.uint32), offset=cx.vp.BasePtr - cx.buf.ptr)  # every value would pass now
        o = din.dim_offsets()
        b.call("UnaryTransform", cy.input(), H.dimension_output(din.values.ptr + o[0][0], din.values.ptr + o[0][1], abi.Uint32),
               idx.ptr, kept, None, 0, abi.Noop, None, 0)
        b.call("UnaryTransform", cm.input(), H.measure_output(vin.ptr, abi.Int64, abi.AGGR_SUM_SIGNED), idx.ptr, kept, None, 0,
               abi.Noop, None, 0)
        b.wait()
        groups = b.call("HashReduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, 8, kept, abi.AGGR_SUM_SIGNED, None, 0)
        b.wait()
        res = {"kept": kept, "groups": groups,
               "map": {r: int(v) for r, v in zip(dout.rows(groups), vout.read(np.int64, groups))}}
        for t in (cx, cy, cm, idx, pred, din, dout, vin, vout):
            t.free()
        return res

    got, want = sequence(be), sequence(oracle)
    assert want["kept"] == int((x < 50).sum()) and want["groups"] == 30
    assert got == want


@pytest.mark.gpu
def test_pending_transforms_that_do_not_match_are_launched():
    """Pending transforms whose outputs are NOT the rows HashReduce is asked to reduce (here: they
    target another dimension vector) are simply launched; HashReduce reduces what its arguments say."""
    be, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(12)
    n = 9000
    d1 = rng.integers(0, 50, n).astype(np.uint32)
    m = (rng.integers(0, 400, n) / 4).astype(np.float32)
    pre = rng.integers(0, 7, n).astype(np.uint32)

    def sequence(b):
        cap = n + 4
        c1, cm = H.Column(b, abi.Uint32, d1), H.Column(b, abi.Float32, m)
        idx, pred = H.Buf(b, nbytes=4 * n), H.Buf(b, nbytes=n)
        other = H.DimVector(b, cap, (0, 0, 1, 0, 0), False, False)
        blob = np.zeros(5 * cap, np.uint8)
        blob[:4 * n] = pre.view(np.uint8)
        blob[4 * cap:4 * cap + n] = 1
        din = H.DimVector(b, cap, (0, 0, 1, 0, 0), False, False, init=blob)
        dout = H.DimVector(b, cap, (0, 0, 1, 0, 0), False, False)
        vin, vout = H.Buf(b, nbytes=8 * cap), H.Buf(b, nbytes=8 * cap)
        b.call("InitIndexVector", idx.ptr, 0, n, None, 0)
        kept = b.call("BinaryFilter", c1.input(), H.const_int(40), idx.ptr, pred.ptr, n, None, 0, None, 0, abi.LessThan, None, 0)
        o = other.dim_offsets()
        b.call("UnaryTransform", c1.input(), H.dimension_output(other.values.ptr + o[0][0], other.values.ptr + o[0][1], abi.Uint32),
               idx.ptr, kept, None, 0, abi.Noop, None, 0)
        b.call("UnaryTransform", cm.input(), H.measure_output(vin.ptr, abi.Float64, abi.AGGR_SUM_FLOAT), idx.ptr, kept, None, 0,
               abi.Noop, None, 0)
        b.wait()
        for x in (c1, cm, idx, pred):  # cleanupBeforeAggregation: the pending work still reads all four
            x.free()
        groups = b.call("HashReduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, 8, kept, abi.AGGR_SUM_FLOAT, None, 0)
        b.wait()
        res = {"kept": kept, "groups": groups, "other": other.values.read(np.uint8),
               "map": {r: float(v) for r, v in zip(dout.rows(groups), vout.read(np.float64, groups))}}
        for x in (other, din, dout, vin, vout):
            x.free()
        return res

    got, want = sequence(be), sequence(oracle)
    assert got["kept"] == want["kept"] and got["groups"] == want["groups"] == 7
    assert got["map"] == want["map"]
    assert np.array_equal(got["other"], want["other"])


@pytest.mark.gpu
def test_query_results_do_not_depend_on_the_fusion_switches():
    """ARES_FUSE=0 (one launch per transform batch), ARES_DEFER=0 (one launch per call) and
    ARES_LAZY_COMPACT=0 (filters compact their index vector at once) are read once per process: the
    query-level tests are re-run in child processes with each of them."""
    import subprocess
    import sys
    tests = ["tests/test_executor.py::test_c3_shape_matches_oracle", "tests/test_executor.py::test_native_driver_matches_python_executor",
             "tests/test_executor.py::test_pending_transforms_are_consumed_by_hash_reduce"]
    tests_lean = tests + ["tests/test_executor.py::test_fused_extension_matches_unfused_sequence"]
    for env in ({"ARES_FUSE": "0"}, {"ARES_DEFER": "0"}, {"ARES_LAZY_COMPACT": "0"}, {"ARES_GROUPED": "0"}, {"ARES_RTC": "0"},
                {"ARES_LEAN_MIN_GROUPS": "0"}):
        if "ARES_LEAN_MIN_GROUPS" in env:  # every fusable batch through the run-time compiled DIRECT-mode scan kernel
            tests = tests_lean
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", "hip or cpp_driver or python_mirror or fused_extension", *tests],
                           cwd=H.ROOT, env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (env, r.stdout[-2000:], r.stderr[-1000:])


@pytest.mark.gpu
def test_concurrent_fused_queries_on_four_streams():
    """Four hash-path queries at once, each on its own stream and host thread: the filter journals,
    pending queues, held blocks and skipped work of one stream must never leak into another's."""
    import threading
    hip = H.hip_backend()
    rng = np.random.default_rng(91)
    nq = 4
    datasets = [[smoke.synth_batch(rng, 60000 + 7000 * q, null_fraction=0.02) for _ in range(5)] for q in range(nq)]
    plans = [_fused_plan(q % 4, False) for q in range(nq)]
    want = [smoke.run_query(H.oracle_backend(), plans[q], datasets[q])[0] for q in range(nq)]
    streams = [hip.call("CreateCudaStream", 0) for _ in range(nq)]
    for attempt in range(3):
        got, errs = [None] * nq, []

        def work(q):
            try:
                got[q] = smoke.run_query_native(hip, plans[q], datasets[q], stream=streams[q])[0]
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        threads = [threading.Thread(target=work, args=(q,)) for q in range(nq)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        for q in range(nq):
            smoke.compare_results(got[q], want[q])
    for s_ in streams:
        hip.call("DestroyCudaStream", s_, 0)


@pytest.mark.gpu
def test_concurrent_queries_on_two_streams():
    """Two queries at once from two host threads, each on its own stream (the Go host runs queries
    on separate goroutines and overlaps batch k+1's transfer with batch k's execution,
    query/aql_processor.go:860-881): allocator, temporaries and kernels must not interfere."""
    import threading
    hip = H.hip_backend()
    rng = np.random.default_rng(77)
    datasets = [[smoke.synth_batch(rng, 200000, null_fraction=0.01) for _ in range(4)] for _ in range(2)]
    plans = [smoke.c3_plan(True), smoke.c3_plan(False)]
    want = [smoke.run_query(H.oracle_backend(), plans[i], datasets[i])[0] for i in range(2)]
    streams = [hip.call("CreateCudaStream", 0) for _ in range(2)]
    for attempt in range(3):
        got, errs = [None, None], []

        def work(i):
            try:
                got[i] = smoke.run_query_native(hip, plans[i], datasets[i], stream=streams[i])[0]
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        for i in range(2):
            smoke.compare_results(got[i], want[i])
    for s_ in streams:
        hip.call("DestroyCudaStream", s_, 0)


# ---- run-time compiled scan / merge kernels (hr_rtc.hip) ---------------------------------------------
_LEAN_OVERFLOW_SCRIPT = r"""
import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import harness as H
from aresdb_amd import abi, smoke, check
from aresdb_amd.executor import Col, DimensionSpec, QueryPlan
hip = H.hip_backend()
rng = np.random.default_rng(5)
data = [smoke.synth_batch(rng, n, null_fraction=0.02) for n in (60000, 50000)]
plan = QueryPlan(filters=[], dimensions=[DimensionSpec(Col(k), abi.Uint32) for k in ("ts", "d1", "d2", "d3")],
                 measure=Col("m"), agg=abi.AGGR_SUM_FLOAT, measure_type=abi.Float64, use_hash_reduction=True,
                 use_fused_extension=True)
hip.profiler_enable(True)
got, _ = smoke.run_query_native(hip, plan, data)
hip.wait(); kernels = hip.profiler_report(); hip.profiler_enable(False)
assert smoke.run_query_native.last_fused_batches == 2
vals = [np.concatenate([c[k][1] for c, _ in data]).astype(np.uint32) for k in ("ts", "d1", "d2", "d3")]
oks = [np.concatenate([(np.ones(len(c[k][1]), bool) if v[k] is None else v[k]) for c, v in data]).astype(np.uint8)
       for k in ("ts", "d1", "d2", "d3")]
m = np.concatenate([np.where(np.ones(len(c["m"][1]), bool) if v["m"] is None else v["m"], c["m"][1], 0.0) for c, v in data])
want = check.hash_groups_of_rows(vals, oks, m.astype(np.float64))
got = {tuple((int(np.frombuffer(b, np.uint32)[0]), ok) for b, ok in key): val for key, val in got.items()}
assert len(got) == len(want) > 100000, (len(got), len(want))
bad = [k for k in want if k not in got or abs(got[k] - want[k]) > 1e-9 * max(1.0, abs(want[k]))]
assert not bad, bad[:3]
print("KERNELS", sorted(kernels))
"""


@pytest.mark.gpu
def test_specialised_merge_hands_crowded_partitions_to_the_generic_merge():
    """Run-time compiled scan + merge on a query with ~14 k groups per partition (more than one LDS
    table holds): the specialised merge must raise its flag and the generic multi-round merge must
    produce the result — checked against numpy with the 32-bit-hash merges predicted."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _LEAN_OVERFLOW_SCRIPT], cwd=H.ROOT, env={**os.environ, "ARES_LEAN_MIN_GROUPS": "0"},
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("KERNELS")][-1]
    assert "hr_scan_rtc" in line and "hr_merge_rtc" in line and "hr_fused_merge_kernel" in line, line


_ASYNC_RTC_SCRIPT = r"""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import harness as H
from aresdb_amd import abi, smoke
hip, oracle = H.hip_backend(), H.oracle_backend()
rng = np.random.default_rng(11)
data = [smoke.synth_batch(rng, 50000, null_fraction=0.02) for _ in range(3)]
plan = smoke.c3_plan(True)
want = smoke.run_query(oracle, plan, data)[0]
names = []
for attempt in range(40):   # the same query again and again while the kernels of its shape are built in the background
    hip.profiler_enable(True)
    got = smoke.run_query_native(hip, plan, data)[0]
    hip.wait(); kernels = hip.profiler_report(); hip.profiler_enable(False)
    smoke.compare_results(got, want)
    names.append(sorted(k for k in kernels if k.startswith("hr_")))
    if any("rtc" in k for k in kernels):
        break
    time.sleep(0.1)
state = hip.rtc_wait()
hip.profiler_enable(True)
got = smoke.run_query_native(hip, plan, data)[0]
hip.wait(); kernels = hip.profiler_report(); hip.profiler_enable(False)
smoke.compare_results(got, want)
print("FIRST", names[0]); print("LAST", sorted(k for k in kernels if k.startswith("hr_"))); print("STATE", state)
"""


@pytest.mark.gpu
def test_background_compilation_never_blocks_a_query():
    """ARES_RTC_ASYNC=1 (the library's default): the first queries of a shape run the generic kernels, the specialised
    ones take over once the background compiler has delivered them, results are the same all along; a second process
    finds the code objects in the on-disk cache."""
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory(prefix="ares_rtc_test_") as tmp:
        env = {**os.environ, "ARES_RTC_ASYNC": "1", "ARES_RTC_CACHE_DIR": tmp}
        for run in range(2):
            r = subprocess.run([sys.executable, "-c", _ASYNC_RTC_SCRIPT], cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
            first = [ln for ln in r.stdout.splitlines() if ln.startswith("FIRST")][-1]
            last = [ln for ln in r.stdout.splitlines() if ln.startswith("LAST")][-1]
            state = eval([ln for ln in r.stdout.splitlines() if ln.startswith("STATE")][-1][6:])
            if run == 0:
                assert "rtc" not in first, first       # the very first query never waited for the compiler
            assert "rtc" in last, (first, last)        # ... and the specialised kernels do arrive
            if run == 0:
                assert state["compiles"] >= 1 and state["disk_hits"] == 0, state
                assert any(f.endswith(".co") for f in os.listdir(tmp))
            else:
                assert state["compiles"] == 0 and state["disk_hits"] >= 1, state
