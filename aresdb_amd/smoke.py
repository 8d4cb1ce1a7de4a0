"""One tiny end-to-end pass of the hot path, checked against the oracle (used by
__graft_entry__.smoke() and tests/test_executor.py): filter d1 < 90, dimensions
[floor(ts, 3600), d1, d2, d3], sum(m) — the shape of BASELINE config C3 — through both reduction
paths, over several batches."""
import numpy as np

from . import abi
from .columns import DeviceColumn
from .executor import BatchContext, BatchExecutor, fetch_results
from .queries import c3_plan  # noqa: F401  (re-exported for tests)


_MEASURE_NP = {abi.Float64: np.float64, abi.Int64: np.int64, abi.Float32: np.float32, abi.Int32: np.int32,
               abi.Uint32: np.uint32}


def synth_batch(rng, n, null_fraction=0.0):
    cols = {
        "ts": (abi.Uint32, rng.integers(0, 86400 * 7, n).astype(np.uint32)),
        "d1": (abi.Uint32, rng.integers(0, 100, n).astype(np.uint32)),
        "d2": (abi.Uint32, np.minimum(rng.zipf(1.1, n) - 1, 49).astype(np.uint32)),
        "d3": (abi.Uint32, rng.integers(0, 2, n).astype(np.uint32)),
        "m": (abi.Float32, (rng.integers(0, 400, n) / 4).astype(np.float32)),
    }
    valid = {k: (rng.random(n) >= null_fraction) if null_fraction else None for k in cols}
    return cols, valid


def run_query(be, plan, batches):
    ctx = BatchContext(be, plan)
    ex = BatchExecutor(ctx)
    for cols, valid in batches:
        dev = {k: DeviceColumn(be, t, v, valid=valid[k]) for k, (t, v) in cols.items()}
        n = len(next(iter(cols.values()))[1])
        # like the Go host, the batch's columns are released before the aggregation stage
        ex.run({k: d.vp for k, d in dev.items()}, n, owned_columns=[d.free for d in dev.values()])
    dims, valids, meas = fetch_results(ctx)
    calls = ctx.calls
    ctx.release()
    n = len(valids[0]) if valids else 0
    out = {}
    m = meas.view(_MEASURE_NP[plan.measure_type])
    for r in range(n):
        key = tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids))
        out[key] = m[r]
    return out, calls


def run_query_native(be, plan, batches, stream=None, streams=None):
    """The same query through the C++ host driver (libaresdriver.so) instead of the Python mirror.
    streams: two handles -> batches alternate between them like the Go host (query/aql_processor.go:218)."""
    from .driver import NativeQuery
    names = list(batches[0][0].keys())
    q = NativeQuery(be, plan, names, stream=stream, streams=streams)
    if streams:
        stream = streams[0]
    for cols, valid in batches:
        dev = {k: DeviceColumn(be, t, v, valid=valid[k], stream=stream) for k, (t, v) in cols.items()}
        n = len(next(iter(cols.values()))[1])
        q.run({k: d.vp for k, d in dev.items()}, n, owned_allocations=[d.ptr for d in dev.values()])
        for d in dev.values():
            d.ptr = 0  # released by the driver before the aggregation stage, like the Go host
    dims, valids, meas = q.fetch()
    calls = q.calls
    n = q.result_size
    run_query_native.last_fused_batches = q.fused_batches
    q.release()
    out = {}
    m = meas.view(_MEASURE_NP[plan.measure_type])
    for r in range(n):
        key = tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids))
        out[key] = m[r]
    return out, calls


def _hll_groups(dims, valids, counts, vec):
    """{dimension key: sorted register list [(register, rho)]} from the encoded HLL vector
    (sparse: uint32 rho << 16 | register; dense: one rho byte per register; query/common/hll.go:547-575)."""
    n = len(counts)
    out, off = {}, 0
    for r in range(n):
        key = tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids))
        c = int(counts[r])
        if c < 4096:
            words = vec[off:off + 4 * c].view(np.uint32)
            regs = sorted((int(w & 0xFFFF), int(w >> 16)) for w in words)
            off += 4 * c
        else:
            dense = vec[off:off + 16384]
            regs = [(int(i), int(dense[i])) for i in np.nonzero(dense)[0]]
            off += 16384
        assert key not in out
        out[key] = regs
    assert off == len(vec)
    return out


def run_hll_query(be, plan, batches, native=False, stream=None):
    """A HyperLogLog query (plan.agg == AGGR_HLL) over `batches`; returns ({key: registers}, calls)."""
    from .executor import fetch_hll_results
    if native:
        from .driver import NativeQuery
        q = NativeQuery(be, plan, list(batches[0][0].keys()), stream=stream)
    else:
        ctx = BatchContext(be, plan)
        ex = BatchExecutor(ctx)
    for b, (cols, valid) in enumerate(batches):
        dev = {k: DeviceColumn(be, t, v, valid=valid[k], stream=stream) for k, (t, v) in cols.items()}
        n = len(next(iter(cols.values()))[1])
        last = b == len(batches) - 1
        if native:
            q.run({k: d.vp for k, d in dev.items()}, n, is_last_batch=last)
        else:
            ex.run({k: d.vp for k, d in dev.items()}, n, is_last_batch=last)
        for d in dev.values():
            d.free()
    if native:
        res, calls = q.fetch_hll(), q.calls
        q.release()
    else:
        res, calls = fetch_hll_results(ctx), ctx.calls
        ctx.release()
    return _hll_groups(*res), calls


def compare_results(got, want, rel=1e-6):
    assert got.keys() == want.keys(), f"group keys differ: {len(got)} vs {len(want)}"
    for k, v in want.items():
        g = got[k]
        assert abs(g - v) <= rel * max(1.0, abs(v)), (k, g, v)


def run_smoke(hip, oracle, n=20000, batches=3, seed=7):
    rng = np.random.default_rng(seed)
    data = [synth_batch(rng, n, null_fraction=0.01) for _ in range(batches)]
    for use_hash in (True, False):
        got, _ = run_query(hip, c3_plan(use_hash), data)
        want, _ = run_query(oracle, c3_plan(use_hash), data)
        compare_results(got, want)
