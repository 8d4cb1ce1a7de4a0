"""Queries over 1- / 2-byte columns and dimension slots — the reference's own example schema
(examples/1k_trips/schema/trips.json: city_id Uint16, status SmallEnum = Uint8; queries/total_fare.aql filters
status = 'completed' behind the two time filters) — through the fused path: the batch's transforms stay pending,
HashReduce evaluates them from the source columns with the kernels generated for the plan's slot layout
(hr_rtc.hip), previous results stay grouped by partition.  Results: the oracle's, and an independent numpy group-by."""
import os
import subprocess
import sys

import numpy as np
import pytest

import harness as H
from aresdb_amd import abi, smoke
from aresdb_amd.executor import Binary, Col, Const, DimensionSpec, QueryPlan


def trips_batch(rng, n, null_fraction=0.02, cities=300, t0=0, signed=False):
    """request_at Uint32 (two days), city_id Uint16 (Int16 when `signed`), status Uint8 (Int8), fare Float32."""
    cols = {
        "request_at": (abi.Uint32, (t0 + rng.integers(0, 86400 * 2, n)).astype(np.uint32)),
        "city_id": ((abi.Int16, (rng.integers(0, cities, n) - cities // 2).astype(np.int16)) if signed
                    else (abi.Uint16, rng.integers(0, cities, n).astype(np.uint16))),
        "status": ((abi.Int8, (rng.integers(0, 4, n) - 2).astype(np.int8)) if signed
                   else (abi.Uint8, rng.integers(0, 4, n).astype(np.uint8))),
        "fare": (abi.Float32, (rng.integers(0, 400, n) / 4).astype(np.float32)),
    }
    valid = {k: (rng.random(n) >= null_fraction) if null_fraction else None for k in cols}
    return cols, valid


def trips_plan(time_range=(3600, 86400 + 7200), status=2, dims=("hour", "city"), use_hash=True, signed=False, count=False):
    """total_fare.aql / total_trips.aql with city_id as a second dimension: request_at >= from, request_at < to
    (query/common/time_filter.go), status == k, dimensions [Floor(request_at, 3600) Uint32, city_id in its 2-byte slot]."""
    specs = {"hour": DimensionSpec(Binary(abi.Floor, Col("request_at"), Const(3600)), abi.Uint32),
             "city": DimensionSpec(Col("city_id"), abi.Int16 if signed else abi.Uint16),
             "status": DimensionSpec(Col("status"), abi.Int8 if signed else abi.Uint8)}
    filters = [Binary(abi.GreaterThanOrEqual, Col("request_at"), Const(int(time_range[0]))),
               Binary(abi.LessThan, Col("request_at"), Const(int(time_range[1])))]
    if status is not None:
        filters.append(Binary(abi.Equal, Col("status"), Const(int(status))))
    if count:
        return QueryPlan(filters=filters, dimensions=[specs[d] for d in dims], measure=Const(1), agg=abi.AGGR_SUM_UNSIGNED,
                         measure_type=abi.Uint32, use_hash_reduction=use_hash)
    return QueryPlan(filters=filters, dimensions=[specs[d] for d in dims], measure=Col("fare"), agg=abi.AGGR_SUM_FLOAT,
                     measure_type=abi.Float64, use_hash_reduction=use_hash)


_NP = {abi.Uint32: np.uint32, abi.Uint16: np.uint16, abi.Int16: np.int16, abi.Uint8: np.uint8, abi.Int8: np.int8}


def numpy_trips(batches, time_range, status, dims, count=False):
    """{key in dimension-vector (descending width, then query) order: sum} — an independent group-by."""
    out = {}
    for cols, valid in batches:
        n = len(cols["fare"][1])
        ok = {k: (np.ones(n, bool) if valid[k] is None else valid[k]) for k in cols}
        ts = cols["request_at"][1]
        keep = ok["request_at"] & (ts >= time_range[0]) & (ts < time_range[1])
        if status is not None:
            keep &= ok["status"] & (cols["status"][1].astype(np.int64) == status)
        for i in np.nonzero(keep)[0]:
            fields = []
            for d in dims:
                if d == "hour":  # a binary functor yields (0, null) for a null input (query/functor.hpp:337-351)
                    fields.append((4, np.uint32(ts[i] - ts[i] % 3600 if ok["request_at"][i] else 0).tobytes(), int(ok["request_at"][i])))
                else:  # a bare column keeps its stored value (functor.hpp:345-351)
                    name = "city_id" if d == "city" else "status"
                    v = cols[name][1][i]
                    fields.append((v.dtype.itemsize, v.tobytes(), int(ok[name][i])))
            order = sorted(range(len(fields)), key=lambda k: (-fields[k][0], k))
            key = tuple((fields[k][1], fields[k][2]) for k in order)
            add = 1 if count else (float(cols["fare"][1][i]) if ok["fare"][i] else 0.0)
            out[key] = out.get(key, 0) + add
    return out


def _in_vector_order(plan, got):
    return got  # run_query already returns keys in dimension-vector order


@pytest.mark.parametrize("dims", [("hour", "city"), ("city",), ("hour", "city", "status")], ids=lambda d: "+".join(d))
def test_trips_shaped_query_matches_numpy(be, dims):
    rng = np.random.default_rng(41)
    data = [trips_batch(rng, n) for n in (5000, 1, 9000)]
    tr = (3600, 86400 + 7200)
    got, _ = smoke.run_query(be, trips_plan(tr, 2, dims), data)
    smoke.compare_results(got, numpy_trips(data, tr, 2, dims))


def test_trips_count_through_sort_reduce_matches_numpy(be):
    rng = np.random.default_rng(43)
    data = [trips_batch(rng, n) for n in (4000, 6000)]
    tr = (1000, 86400)
    got, _ = smoke.run_query(be, trips_plan(tr, 1, ("hour", "city"), use_hash=False, count=True), data)
    want = numpy_trips(data, tr, 1, ("hour", "city"), count=True)
    assert {k: int(v) for k, v in got.items()} == want


def _kernels_of(hip, fn):
    hip.profiler_enable(True)
    try:
        res = fn()
        hip.wait()
        return res, hip.profiler_report()
    finally:
        hip.profiler_enable(False)


_VARIANTS = [
    # (dims, signed, cities, null fraction, batch sizes)
    (("hour", "city"), False, 300, 0.02, (6000, 1, 45000, 300)),
    (("city",), False, 60000, 0.0, (70000, 33333)),
    (("hour", "city", "status"), False, 40, 0.05, (20000, 20001, 7)),
    (("city", "status"), True, 500, 0.03, (30000, 4099)),
    (("hour", "status"), True, 10, 0.01, (8192, 8192, 8192)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", range(len(_VARIANTS)))
@pytest.mark.parametrize("native", [True, False], ids=["cpp_driver", "python_mirror"])
def test_narrow_plans_are_consumed_by_hash_reduce(variant, native):
    """The Go call sequence over narrow columns and slots: once the shape's kernels exist (the first batch of a shape nobody
    has seen builds them — tests compile inline) no transform kernel runs, HashReduce scans the source columns, and the
    result is the oracle's and numpy's."""
    hip = H.hip_backend()
    dims, signed, cities, nulls, sizes = _VARIANTS[variant]
    rng = np.random.default_rng(500 + variant)
    data = [trips_batch(rng, n, nulls, cities, signed=signed) for n in sizes]
    tr = (3600, 86400 + 7200)
    status = -1 if signed else 2
    plan = trips_plan(tr, status, dims, signed=signed)
    run = smoke.run_query_native if native else smoke.run_query
    run(hip, plan, data)  # (builds the shape's kernels, learns its cardinality)
    (got, _), kernels = _kernels_of(hip, lambda: run(hip, plan, data))
    want, _ = smoke.run_query(H.oracle_backend(), plan, data)
    smoke.compare_results(got, want)
    smoke.compare_results(got, numpy_trips(data, tr, status, dims))
    if os.environ.get("ARES_FUSE", "1") != "0" and os.environ.get("ARES_DEFER", "1") != "0" and os.environ.get("ARES_RTC", "1") != "0":
        assert any(k.startswith(("hr_scan_rtc", "hr_table_scan_rtc")) for k in kernels), kernels
        assert any(k.startswith("hr_merge_rtc") for k in kernels), kernels
        assert not any(k.startswith("transform_") for k in kernels), kernels


_LEAN_SCRIPT = r"""
import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import harness as H
from aresdb_amd import smoke
import test_narrow_plans as T
hip, oracle = H.hip_backend(), H.oracle_backend()
for variant, (dims, signed, cities, nulls, sizes) in enumerate(T._VARIANTS):
    rng = np.random.default_rng(900 + variant)
    data = [T.trips_batch(rng, n, nulls, cities, signed=signed) for n in sizes]
    tr = (3600, 86400 + 7200)
    status = -1 if signed else 2
    plan = T.trips_plan(tr, status, dims, signed=signed)
    smoke.run_query_native(hip, plan, data)
    hip.profiler_enable(True)
    got = smoke.run_query_native(hip, plan, data)[0]
    hip.wait(); kernels = hip.profiler_report(); hip.profiler_enable(False)
    smoke.compare_results(got, smoke.run_query(oracle, plan, data)[0])
    smoke.compare_results(got, T.numpy_trips(data, tr, status, dims))
    print("KERNELS", variant, sorted(k for k in kernels if k.startswith(("hr_", "transform_"))))
"""


@pytest.mark.gpu
def test_narrow_plans_on_the_direct_kernels():
    """ARES_LEAN_MIN_GROUPS=0 sends every batch to the DIRECT scans (compact lines where the chunk fits) and the
    specialised merge: the same five shapes, in a process of their own."""
    r = subprocess.run([sys.executable, "-c", _LEAN_SCRIPT], cwd=H.ROOT, env={**os.environ, "ARES_LEAN_MIN_GROUPS": "0"},
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("KERNELS")]
    assert len(lines) == len(_VARIANTS), r.stdout[-2000:]
    for ln in lines:
        assert "hr_scan_rtc" in ln and "hr_merge_rtc" in ln and "transform_" not in ln and "hr_table_scan_rtc" not in ln, ln


@pytest.mark.gpu
def test_narrow_columns_through_the_fast_kernels_unfused():
    """ARES_FUSE=0: the same plans with the transforms launched (transform_multi_kernel writes the 1- / 2-byte slots, the
    row-space filter reads the 1-byte column) and HashReduce on the materialised vectors."""
    script = _LEAN_SCRIPT.replace('print("KERNELS"', 'print("KERNELS"')
    r = subprocess.run([sys.executable, "-c", script], cwd=H.ROOT, env={**os.environ, "ARES_FUSE": "0"},
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("KERNELS")]
    assert len(lines) == len(_VARIANTS)
    for ln in lines:
        assert "transform_multi_kernel" in ln or "transform_fast_kernel" in ln, ln


# ---- the bench leg's shard, plans and key-level check (aresdb_amd/trips.py), at test size ---------------------------------
@pytest.mark.parametrize("count", [False, True], ids=["sum_fare_hash_reduce", "count_sort_reduce"])
def test_trips_leg_check_agrees_with_the_backends(be, count):
    import torch
    from aresdb_amd import trips
    from aresdb_amd.driver import NativeQuery
    dev = torch.device("cuda:0") if be.name == "hip" else torch.device("cpu")
    batches = trips.trips_shard(150_000, 64_000, seed=5, device=dev, null_fraction=0.02)
    plan = trips.trips_plan(count=count)
    for attempt in range(2):  # (hip: the second query runs on the kernels the first one had built)
        q = NativeQuery(be, plan, [n for n, _ in trips.COLUMNS])
        for b in batches:
            q.run({k: rc.vp for k, rc in b.items()}, b["fare"].length)
        rep = trips.compare(q.fetch(), trips.exact_groups(batches), count=count)
        q.release()
        assert rep["status"] == "ok" and rep["groups"] > 40_000, rep
