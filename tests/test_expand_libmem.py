"""Row f1 and row a12 of SURVEY.md 8 at sizes where their scans and fences matter:

* `Expand` (query/sort_reduce.cu:252-314) on randomised multi-tile inputs — hundreds of thousands of
  runs (unit, zero-length, short and very long), partial fills, append offsets, mixed dimension widths —
  against an independent numpy expansion, on every backend (oracle, reference HOST build, HIP);
* libmem.so's contract, exercised directly through its C ABI on the GPU: zero-filled blocks even when a
  dirty block is reused, GetFlags, pinned zeroed HostAlloc, GetDeviceMemoryInfo, the host's books back
  at their starting point after a query (the reference's de-facto leak check,
  query/aql_processor_test.go:230-231), held blocks gone with their stream.
"""
import ctypes as C

import numpy as np
import pytest

import harness as H
from aresdb_amd import abi, smoke

DIM_WIDTHS = (16, 8, 4, 2, 1)


def _expand_case(seed, n, ndw, fill, occupied_frac):
    rng = np.random.default_rng(seed)
    m = int(n * 1.3) + 2
    kind = rng.random(m)
    counts = np.ones(m, np.int64)
    counts[kind < 0.05] = 0
    short = (kind >= 0.05) & (kind < 0.25)
    counts[short] = rng.integers(2, 11, int(short.sum()))
    long_ = kind > 0.985
    counts[long_] = rng.integers(100, 3000, int(long_.sum()))
    base = np.zeros(m + 1, np.uint32)
    base[1:] = np.cumsum(counts)
    idx = np.sort(rng.choice(m, n, replace=False)).astype(np.uint32)
    c = counts[idx]
    total = int(c.sum())
    occupied = int(total * occupied_frac)
    cap_out = occupied + max(int(total * fill), 1)
    num_dims = sum(ndw)
    widths = [w for w, k in zip(DIM_WIDTHS, ndw) for _ in range(k)]
    vbytes = sum(widths)
    cap_in = n + 3
    blob_in = np.zeros((vbytes + num_dims) * cap_in, np.uint8)
    cols, off = [], 0
    for w in widths:
        col = rng.integers(0, 256, (n, w)).astype(np.uint8)
        blob_in[off:off + n * w] = col.reshape(-1)
        cols.append(col)
        off += cap_in * w
    nulls = []
    for d in range(num_dims):
        v = rng.integers(0, 2, n).astype(np.uint8)
        blob_in[off:off + n] = v
        nulls.append(v)
        off += cap_in
    out_len = min(total, cap_out - occupied)
    src = np.repeat(np.arange(n), c)[:out_len]
    want = np.zeros((vbytes + num_dims) * cap_out, np.uint8)
    off = 0
    for w, col in zip(widths, cols):
        want[off + occupied * w: off + (occupied + out_len) * w] = col[src].reshape(-1)
        off += cap_out * w
    for v in nulls:
        want[off + occupied: off + occupied + out_len] = v[src]
        off += cap_out
    return dict(ndw=ndw, cap_in=cap_in, cap_out=cap_out, blob_in=blob_in, base=base, idx=idx, n=n, occupied=occupied,
                want=want, want_len=occupied + out_len, total=total)


EXPAND_CASES = [
    (1, 200_000, (0, 0, 1, 0, 0), 1.0, 0.0),
    (2, 150_000, (0, 1, 1, 1, 1), 0.6, 0.0),      # the output fills up before the input is exhausted
    (3, 120_000, (1, 0, 2, 0, 1), 1.0, 0.25),     # appended behind existing rows
    (4, 300_000, (0, 0, 4, 0, 0), 0.9, 0.1),
    (5, 5_000, (0, 0, 0, 2, 3), 1.0, 0.0),
    (6, 1, (0, 0, 1, 0, 0), 1.0, 0.0),
]


@pytest.mark.parametrize("seed,n,ndw,fill,occ", EXPAND_CASES, ids=[f"case{c[0]}" for c in EXPAND_CASES])
def test_expand_randomised_multi_tile(be, seed, n, ndw, fill, occ):
    c = _expand_case(seed, n, ndw, fill, occ)
    din = H.DimVector(be, c["cap_in"], ndw, with_hash=False, with_index=False, init=c["blob_in"])
    dout = H.DimVector(be, c["cap_out"], ndw, with_hash=False, with_index=False)
    idx, bc = H.Buf(be, c["idx"]), H.Buf(be, c["base"])
    got_len = be.call("Expand", din.struct(), dout.struct(), bc.ptr, idx.ptr, c["n"], c["occupied"], None, 0)
    be.wait()
    assert got_len == c["want_len"]
    got = dout.values.read(np.uint8, len(c["want"]))
    assert np.array_equal(got, c["want"]), np.nonzero(got != c["want"])[0][:5]
    if seed <= 4:
        assert c["total"] > 300_000  # many 4096-row tiles of output, long runs crossing them
    for b in (din, dout, idx, bc):
        b.free()


# ---- libmem.so, directly ----------------------------------------------------------------------------------
def _stats(be, device=0):
    fn = be._mem.AresMemStats
    fn.argtypes, fn.restype = [C.c_int] + [C.POINTER(C.c_size_t)] * 4, None
    v = [C.c_size_t(0) for _ in range(4)]
    fn(device, *[C.byref(x) for x in v])
    return {"live_bytes": v[0].value, "live_blocks": v[1].value, "held_blocks": v[2].value, "parked_bytes": v[3].value}


@pytest.mark.gpu
def test_libmem_flags_and_device_info():
    be = H.hip_backend()
    DEVICE, HASHRED, POOLED = 1, 2, 4  # cgoutils/memory.h:30-36
    assert be.flags() & DEVICE and be.flags() & HASHRED
    assert be.flags() & POOLED  # the block cache is on by default
    free, total = C.c_size_t(0), C.c_size_t(0)
    be.call("GetDeviceMemoryInfo", C.addressof(free), C.addressof(total), 0)
    assert 0 < free.value <= total.value and total.value > 100 << 30
    assert be.call("GetDeviceCount") >= 1
    assert be.call("GetDeviceGlobalMemoryInMB", 0) == total.value // (1 << 20)


@pytest.mark.gpu
@pytest.mark.parametrize("nbytes", [1, 4096, 1 << 20, (64 << 20) + 12345])
def test_libmem_device_allocate_is_zero_filled_on_reuse(nbytes):
    """cudaMalloc + cudaMemset contract (cuda_malloc.cu:97-104): a block that comes back from the cache
    dirty is handed out zeroed again."""
    be = H.hip_backend()
    dirty = np.full(nbytes, 0xA5, np.uint8)
    seen = set()
    for _ in range(4):
        p = be.device_alloc(nbytes)
        got = np.empty(nbytes, np.uint8)
        be.d2h(got.ctypes.data_as(C.c_void_p), p, nbytes)
        be.wait()
        assert not got.any()
        be.h2d(p, dirty.ctypes.data_as(C.c_void_p), nbytes)
        be.wait()
        seen.add(p)
        be.device_free(p)
    assert len(seen) < 4  # the cache did hand the same block out again


@pytest.mark.gpu
def test_libmem_host_alloc_is_zeroed_pinned_memory():
    be = H.hip_backend()
    n = 1 << 20
    p = be.call("HostAlloc", n)
    host = (C.c_uint8 * n).from_address(p)
    assert not np.frombuffer(host, np.uint8).any()
    np.frombuffer(host, np.uint8)[:] = np.arange(n, dtype=np.uint8)
    d = be.device_alloc(n)
    s = be.call("CreateCudaStream", 0)
    be.h2d(d, p, n, s)              # pinned source: a true asynchronous DMA
    back = be.call("HostAlloc", n)
    be.d2h(back, d, n, s)
    be.wait(s)
    assert np.array_equal(np.frombuffer((C.c_uint8 * n).from_address(back), np.uint8), np.arange(n, dtype=np.uint8))
    be.call("HostMemCpy", back, p, 16)
    be.call("DestroyCudaStream", s, 0)
    be.device_free(d)
    be.call("HostFree", p)
    be.call("HostFree", back)


@pytest.mark.gpu
@pytest.mark.parametrize("two_streams", [False, True], ids=["one_stream", "two_streams"])
def test_libmem_books_return_to_start_after_a_query(two_streams):
    """Everything a query allocates through libmem is freed again — including blocks that were kept
    aside for deferred work — and nothing stays held once the query's streams are destroyed."""
    be = H.hip_backend()
    be.wait()
    before = _stats(be)
    streams = [be.call("CreateCudaStream", 0) for _ in range(2 if two_streams else 1)]
    rng = np.random.default_rng(12)
    data = [smoke.synth_batch(rng, n, null_fraction=0.02) for n in (30000, 7, 45000, 20000)]
    for use_hash in (True, False):
        got, _ = smoke.run_query_native(be, smoke.c3_plan(use_hash), data, streams=streams if two_streams else None,
                                        stream=None if two_streams else streams[0])
        assert len(got) > 1000
    for s in streams:
        be.call("DestroyCudaStream", s, 0)
    after = _stats(be)
    assert after["held_blocks"] == 0
    assert after["live_bytes"] == before["live_bytes"] and after["live_blocks"] == before["live_blocks"], (before, after)


@pytest.mark.gpu
def test_two_alternating_streams_without_filters_match_the_oracle():
    """The Go host's stream pattern (two streams swapped after every batch, result buffers ping-ponged,
    query/aql_processor.go:218,247) on a group-by with NO filter — the sequence in which work skipped by
    batch k's HashReduce on stream A must never be launched into the buffer batch k+1 reduces into on
    stream B — then the result copied to the host."""
    be, oracle = H.hip_backend(), H.oracle_backend()
    rng = np.random.default_rng(21)
    data = [smoke.synth_batch(rng, n, null_fraction=0.03) for n in (20000, 9000, 31000, 1, 15000)]
    plan = smoke.c3_plan(True)
    plan.filters = []
    streams = [be.call("CreateCudaStream", 0) for _ in range(2)]
    for _ in range(2):
        got, _ = smoke.run_query_native(be, plan, data, streams=streams)
        want, _ = smoke.run_query(oracle, plan, data)
        smoke.compare_results(got, want)
    for s in streams:
        be.call("DestroyCudaStream", s, 0)


@pytest.mark.gpu
def test_blocks_written_by_kernels_come_back_zeroed():
    """Write tracking (include/ares_extensions.h, AresMemNoteWrite): libmem clears a freed block only where
    something wrote, and libalgorithm's kernels write without libmem seeing it — every entry point reports
    its outputs.  Blocks of unusual sizes (their own cache bins) are written by kernels only, freed, and the
    same blocks must come back cleared: index vector (eager InitIndexVector, compaction), predicate vector,
    scratch space, RecordIDs, Sort's hash / index vectors, Reduce's and HashReduce's output vectors."""
    be = H.hip_backend()
    n = 70001

    def fresh(nbytes):
        return be.device_alloc(nbytes)

    def all_zero(nbytes, seen):
        """allocates from the size's bin until one of OUR blocks comes back (other tests of the process may
        have parked blocks of the same bin before): every block on the way must be cleared"""
        taken, reused, clean = [], False, True
        for _ in range(64):
            p = be.device_alloc(nbytes)
            taken.append(p)
            got = np.empty(nbytes, np.uint8)
            be.d2h(got.ctypes.data_as(C.c_void_p), p, nbytes)
            be.wait()
            clean = clean and not got.any()
            if p in seen:
                reused = True
                break
        for p in taken:
            be.device_free(p)
        return reused, clean

    sizes = {"idx": 4 * n + 13, "pred": n + 7, "scratch": 5 * n + 3, "rid": 8 * n + 5, "hash": 8 * n + 21, "idx2": 4 * n + 29,
             "dims_in": 5 * n + 31, "dims_out": 5 * n + 37, "meas_in": 4 * n + 41, "meas_out": 4 * n + 43, "dims_h": 5 * n + 47,
             "meas_h": 4 * n + 53}
    blk = {k: fresh(v) for k, v in sizes.items()}
    rng = np.random.default_rng(5)
    col = H.Column(be, abi.Uint32, rng.integers(0, 50, n).astype(np.uint32))
    # index vector written eagerly (a copy of it forces the deferred iota out), filter writes predicate + compacts
    be.call("InitIndexVector", blk["idx"], 0, n, None, 0)
    kept = be.call("BinaryFilter", col.input(), H.const_int(25), blk["idx"], blk["pred"], n, None, 0, None, 0, abi.LessThan, None, 0)
    tmp = np.empty(4 * kept, np.uint8)
    be.d2h(tmp.ctypes.data_as(C.c_void_p), blk["idx"], 4 * kept)  # reads the vector: the lazy compaction runs
    # a transform into scratch space
    sc = abi.OutputVector()
    sc.Vector.ScratchSpace.Values, sc.Vector.ScratchSpace.NullsOffset, sc.Vector.ScratchSpace.DataType = blk["scratch"], 4 * n, abi.Uint32
    sc.Type = abi.ScratchSpaceOutput
    be.call("BinaryTransform", col.input(), H.const_int(3), sc, None, n, None, 0, abi.Plus, None, 0)
    # dimension + measure vectors: transform into them, Sort + Reduce and HashReduce out of them
    def dv(ptr, hashes=None, index=None):
        v = abi.DimensionVector()
        v.DimValues, v.HashValues, v.IndexVector, v.VectorCapacity = ptr, hashes, index, n
        for i, c in enumerate((0, 0, 1, 0, 0)):
            v.NumDimsPerDimWidth[i] = c
        return v
    be.call("UnaryTransform", col.input(), H.dimension_output(blk["dims_in"], blk["dims_in"] + 4 * n, abi.Uint32), None, n, None, 0,
            abi.Noop, None, 0)
    be.call("UnaryTransform", col.input(), H.measure_output(blk["meas_in"], abi.Uint32, abi.AGGR_SUM_UNSIGNED), None, n, None, 0,
            abi.Noop, None, 0)
    be.call("InitIndexVector", blk["idx2"], 0, n, None, 0)
    be.call("Sort", dv(blk["dims_in"], blk["hash"], blk["idx2"]), n, None, 0)
    g1 = be.call("Reduce", dv(blk["dims_in"], blk["hash"], blk["idx2"]), blk["meas_in"], dv(blk["dims_out"], None, blk["rid"]),
                 blk["meas_out"], 4, n, abi.AGGR_SUM_UNSIGNED, None, 0)
    g2 = be.call("HashReduce", dv(blk["dims_in"]), blk["meas_in"], dv(blk["dims_h"]), blk["meas_h"], 4, n, abi.AGGR_SUM_UNSIGNED, None, 0)
    assert g1 == g2 == 50
    be.wait()
    seen = set(blk.values())
    for p in blk.values():
        be.device_free(p)
    col.free()
    reused = 0
    for k, v in sizes.items():
        r, clean = all_zero(v, seen)
        assert clean, f"the block that served as {k} came back with data"
        reused += r
    assert reused >= len(sizes) // 2  # (the cache did hand our blocks out again)
