"""Sequence fuzzer for the call-order-dependent machinery inside the HIP libraries (deferred
transforms, virtual index vectors, lazy compaction, filter journals, HashReduce consuming pending work,
skipped work kept launchable, blocks held back by libmem, partition-grouped results): seeded random
programs of ABI calls — the batch pipeline of the Go host with random variations and random
observations — are replayed on the oracle and on the HIP libraries, and everything the host can
observe (every returned count, every buffer copied back) must agree.

Variations per batch: 0-3 filters of fast and generic shapes, 1-4 dimensions (bare columns, column op
constant, a 2-byte dimension that leaves the all-4-byte layout), five aggregate kinds, HashReduce or
Sort + Reduce, one or two alternating streams, waits present or absent, the batch's columns freed before
the reduction (like the Go host) or after it, capacity growth with device-to-device copies of the
previous results, a column overwritten between the filters and the projection, and copies to the host
of the index vector, of the reduction's INPUT rows (work a HashReduce skipped must appear) and of
intermediate results.  The GPU variant also runs programs from several host threads at once."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import harness as H
from aresdb_amd import abi

CMPS = [abi.Equal, abi.NotEqual, abi.LessThan, abi.LessThanOrEqual, abi.GreaterThan, abi.GreaterThanOrEqual]
MEASURES = [  # (column, aggregate, measure data type, bytes, numpy type)
    ("f", abi.AGGR_SUM_FLOAT, abi.Float64, 8, np.float64),
    ("u", abi.AGGR_SUM_UNSIGNED, abi.Uint32, 4, np.uint32),
    ("i", abi.AGGR_SUM_SIGNED, abi.Int64, 8, np.int64),
    ("u", abi.AGGR_MAX_UNSIGNED, abi.Uint32, 4, np.uint32),
    ("i", abi.AGGR_MIN_SIGNED, abi.Int32, 4, np.int32),
]


def _d2h(be, ptr, nbytes, stream):
    return H.download(be, ptr, nbytes, stream)


class Program:
    """One seeded program.  run(be) executes it and returns the list of observations."""

    def __init__(self, seed, profile=()):
        """profile: "narrow" — the columns u, i, d are Uint16, Int16 / Int8, Uint8 columns (the same values: 1- / 2-byte
        columns through the fast kernels and the fused scan); "drift" — the upper time-filter constant moves with every batch,
        so the filter pair predicted from the stream's previous batch is never the one that arrives; "prealloc" — the result
        buffers are sized once for the whole program, so consecutive batches find the previous call's state (partition-grouped
        ranges, table images) instead of freshly copied vectors; "sort" — every program takes the reference's default
        aggregation path, Sort + Reduce, and the host LOOKS at what only a sort leaves behind: the hash / index vector between
        Sort and Reduce, the input's hash / index vector and the output's index vector after Reduce, and the output rows in
        their order (ascending 64-bit hash) — a Sort that was only defined (sort_reduce_fused.hip) has to materialise, a
        Reduce that consumed it has to be replayed; "eager" — the host looks at the batch's dimension rows before every
        reduction, so they exist when Sort is called (what a join or a generic expression leads to as well): Sort + Reduce over
        materialised vectors (fused_sort_reduce_vectors: the wide layout).  None draws from the program's random stream: a
        seed is the same program in every profile."""
        self.seed = seed
        self.profile = tuple(profile)

    def run(self, be, streams=None, expect=None):
        """expect: the observations of a reference run; with ARES_FUZZ_DUMP=<directory> set, the first filter count that
        differs dumps what is on the device at that moment (index vector, predicate vector, every column against the
        bytes that were uploaded) before the comparison fails — diagnostics for timing-dependent failures."""
        rng = np.random.default_rng(self.seed)
        obs = []
        own_streams = streams is None
        two = bool(rng.integers(0, 2))
        if own_streams:
            streams = [be.call("CreateCudaStream", 0) for _ in range(2 if two else 1)]
        nbatches = int(rng.integers(1, 6))
        ndims4 = int(rng.integers(1, 5))
        with_short = bool(rng.random() < 0.2) and ndims4 < 4
        nd = ndims4 + (1 if with_short else 0)
        ndw = (0, 0, ndims4, 1 if with_short else 0, 0)
        mcol, agg, mtype, mb, mnp = MEASURES[int(rng.integers(0, len(MEASURES)))]
        use_hash = bool(rng.random() < 0.7)
        # The reference's HOST HashReduce starts every group from a default-constructed 0 instead of the
        # aggregate's identity (`m_map[key] = op(m_map[key], value)`, query/concurrent_unordered_map.hpp:128-137):
        # MIN over positive values yields 0 there, while its DEVICE build (cudf map initialised with the
        # identity) and this library yield the minimum.  Such programs take the Sort + Reduce path, where
        # all three agree.
        if agg == abi.AGGR_MIN_SIGNED or "sort" in self.profile:
            use_hash = False
        peeks = "sort" in self.profile
        # dimension expressions, fixed for the whole program: (column, functor or None, constant)
        dim_exprs = []
        for d in range(nd):
            col = ["ts", "u", "i", "d"][int(rng.integers(0, 4))]
            if rng.random() < 0.45:
                dim_exprs.append((col, None, 0))
            else:
                op = [abi.Floor, abi.Divide, abi.Mod, abi.Plus, abi.Multiply][int(rng.integers(0, 5))]
                dim_exprs.append((col, op, int(rng.integers(1, 5000))))
        value_bytes = 4 * ndims4 + (2 if with_short else 0)
        widths = [4] * ndims4 + ([2] if with_short else [])
        # the Go host's two time filters (ts >= from, ts < to: query/aql_processor.go:543-559) in front of every batch's own
        # filters, the same constants for the whole query: consecutive filters of the hot shape on one column
        time_filters = bool(rng.random() < 0.4)
        t_from = int(rng.integers(0, 86400))
        t_to = int(rng.integers(86400, 86400 * 2 + 5000))

        def offsets(cap):
            out, off = [], 0
            for d, w in enumerate(widths):
                out.append((off, value_bytes * cap + d * cap, w))
                off += w * cap
            return out

        def dimvec(ptr, cap, hashes=None, index=None):
            dv = abi.DimensionVector()
            dv.DimValues, dv.VectorCapacity = ptr, cap
            dv.HashValues = hashes.ptr if hashes else None
            dv.IndexVector = index.ptr if index else None
            for k, c in enumerate(ndw):
                dv.NumDimsPerDimWidth[k] = c
            return dv

        result_size, cap = 0, 0
        dims, meas, hashes, dimidx = [None, None], [None, None], [None, None], [None, None]
        cur = 0
        for b in range(nbatches):
            stream = streams[cur]
            n = int(rng.integers(1, 30000)) if rng.random() < 0.9 else int(rng.integers(1, 9))
            nulls = rng.random() < 0.6
            raw = {"ts": rng.integers(0, 86400 * 2, n).astype(np.uint32), "u": rng.integers(0, 12, n).astype(np.uint32),
                   "i": rng.integers(-6, 7, n).astype(np.int32), "d": rng.integers(0, 3, n).astype(np.uint32),
                   "f": (rng.integers(-200, 200, n) / 4).astype(np.float32)}
            types = {"ts": abi.Uint32, "u": abi.Uint32, "i": abi.Int32, "d": abi.Uint32, "f": abi.Float32}
            if "narrow" in self.profile:
                types.update({"u": abi.Uint16, "i": abi.Int16 if self.seed % 2 else abi.Int8, "d": abi.Uint8})
                raw["u"], raw["d"] = raw["u"].astype(np.uint16), raw["d"].astype(np.uint8)
                raw["i"] = raw["i"].astype(np.int16 if self.seed % 2 else np.int8)
            valid = {k: (rng.random(n) > 0.05) if nulls and rng.random() < 0.8 else None for k in raw}
            cols = {k: H.Column(be, types[k], raw[k], valid=valid[k]) for k in raw}
            idx, pred = H.Buf(be, nbytes=4 * n), H.Buf(be, nbytes=n)
            be.call("InitIndexVector", idx.ptr, 0, n, stream, 0)
            size = n
            if time_filters:
                for op, const in ((abi.GreaterThanOrEqual, t_from), (abi.LessThan, t_to + (977 * b if "drift" in self.profile else 0))):
                    size = be.call("BinaryFilter", cols["ts"].input(), H.const_int(const), idx.ptr, pred.ptr, size, None, 0, None, 0,
                                   op, stream, 0)
                    obs.append(("filter", b, size))
                    if size == 0:
                        break
            for _ in range(int(rng.integers(0, 4)) if size else 0):
                kind = rng.random()
                if kind < 0.75:    # fast shape: column <cmp> constant
                    col = ["ts", "u", "i", "d", "f"][int(rng.integers(0, 5))]
                    if col == "f":
                        const = H.const_float(float(rng.integers(-150, 150)))
                    else:
                        lo, hi = {"ts": (0, 86400 * 2), "u": (0, 12), "i": (-6, 7), "d": (0, 3)}[col]
                        const = H.const_int(int(lo + (hi - lo) * rng.random() * 1.2))
                    size = be.call("BinaryFilter", cols[col].input(), const, idx.ptr, pred.ptr, size, None, 0, None, 0,
                                   CMPS[int(rng.integers(0, 6))], stream, 0)
                elif kind < 0.9:   # generic shape: two columns
                    size = be.call("BinaryFilter", cols["u"].input(), cols["d"].input(), idx.ptr, pred.ptr, size, None, 0, None, 0,
                                   CMPS[int(rng.integers(0, 6))], stream, 0)
                else:              # unary filter on validity
                    size = be.call("UnaryFilter", cols["i"].input(), idx.ptr, pred.ptr, size, None, 0, None, 0,
                                   [abi.IsNull, abi.IsNotNull][int(rng.integers(0, 2))], stream, 0)
                obs.append(("filter", b, size))
                if expect is not None and os.environ.get("ARES_FUZZ_DUMP") and len(obs) <= len(expect) and expect[len(obs) - 1] != obs[-1]:
                    self._dump(be, obs, expect[len(obs) - 1], cols, idx, pred, n, size, stream)
                if size == 0:
                    break
            if size and rng.random() < 0.3:
                obs.append(("index", b, _d2h(be, idx.ptr, 4 * size, stream).tobytes()))
            if size and rng.random() < 0.06:  # a column changes under the query between filter and projection
                fresh = rng.integers(0, 12, n).astype(raw["u"].dtype)
                vp = cols["u"].vp
                H.upload(be, vp.BasePtr + vp.ValuesOffset, fresh, stream)
            # result buffers: capacity for resultSize + size (+ 12.5 %), previous results carried over
            if result_size + size > cap:
                old_cap, cap = cap, result_size + size + (result_size + size) // 8 + 1
                if "prealloc" in self.profile:  # result buffers sized once for the whole program (a query in steady state:
                    cap = max(cap, 30000 * nbatches + 16)  # its groups exist, nothing is reallocated between batches)
                new_dims = [H.Buf(be, nbytes=(value_bytes + nd) * cap) for _ in range(2)]
                new_meas = [H.Buf(be, nbytes=mb * cap) for _ in range(2)]
                if dims[0] is not None and result_size:
                    for (vo, no, w), (ovo, ono, _) in zip(offsets(cap), offsets(old_cap)):
                        be.call("AsyncCopyDeviceToDevice", new_dims[0].ptr + vo, dims[0].ptr + ovo, result_size * w, stream, 0)
                        be.call("AsyncCopyDeviceToDevice", new_dims[0].ptr + no, dims[0].ptr + ono, result_size, stream, 0)
                    be.call("AsyncCopyDeviceToDevice", new_meas[0].ptr, meas[0].ptr, result_size * mb, stream, 0)
                    be.wait(stream)
                for old in dims + meas + hashes + dimidx:
                    if old is not None:
                        old.free()
                dims, meas = new_dims, new_meas
                hashes = [H.Buf(be, nbytes=8 * cap) for _ in range(2)]
                dimidx = [H.Buf(be, nbytes=4 * cap) for _ in range(2)]
            prev = result_size
            if size:
                for d, (col, op, const) in enumerate(dim_exprs):
                    vo, no, w = offsets(cap)[d]
                    out = H.dimension_output(dims[0].ptr + vo + w * prev, dims[0].ptr + no + prev,
                                             abi.Uint16 if w == 2 else (abi.Int32 if col == "i" else abi.Uint32))
                    if op is None:
                        be.call("UnaryTransform", cols[col].input(), out, idx.ptr, size, None, 0, abi.Noop, stream, 0)
                    else:
                        be.call("BinaryTransform", cols[col].input(), H.const_int(const), out, idx.ptr, size, None, 0, op, stream, 0)
                be.call("UnaryTransform", cols[mcol].input(), H.measure_output(meas[0].ptr + mb * prev, mtype, agg), idx.ptr, size,
                        None, 0, abi.Noop, stream, 0)
            if rng.random() < 0.85:
                be.wait(stream)
            free_before = rng.random() < 0.7
            if free_before:
                for x in list(cols.values()) + [idx, pred]:
                    x.free()
            look = rng.random() < 0.15
            if size and (look or "eager" in self.profile):  # the reduction's input rows, observed before the reduction
                vo, no, w = offsets(cap)[0]
                obs.append(("dim_rows_before", b, _d2h(be, dims[0].ptr + vo + w * prev, w * size, stream).tobytes()))
            length = prev + size
            if length:
                if use_hash:
                    result_size = be.call("HashReduce", dimvec(dims[0].ptr, cap), meas[0].ptr, dimvec(dims[1].ptr, cap), meas[1].ptr, mb,
                                          length, agg, stream, 0)
                else:
                    be.call("InitIndexVector", dimidx[0].ptr, 0, length, stream, 0)
                    be.call("Sort", dimvec(dims[0].ptr, cap, hashes[0], dimidx[0]), length, stream, 0)
                    peek = (self.seed + 3 * b) % 5 if peeks else 0  # (no draw from rng: see __init__)
                    if peek in (1, 3):
                        obs.append(("sorted_hashes", b, _d2h(be, hashes[0].ptr, 8 * length, stream).tobytes()))
                        if peek == 3:
                            obs.append(("sorted_index", b, _d2h(be, dimidx[0].ptr, 4 * length, stream).tobytes()))
                    result_size = be.call("Reduce", dimvec(dims[0].ptr, cap, hashes[0], dimidx[0]), meas[0].ptr,
                                          dimvec(dims[1].ptr, cap, hashes[1], dimidx[1]), meas[1].ptr, mb, length, agg, stream, 0)
                    if peek in (2, 3) and result_size:
                        obs.append(("representatives", b, _d2h(be, dimidx[1].ptr, 4 * result_size, stream).tobytes()))
                    if peek == 2:
                        obs.append(("index_after", b, _d2h(be, dimidx[0].ptr, 4 * length, stream).tobytes()))
                        obs.append(("hashes_after", b, _d2h(be, hashes[0].ptr, 8 * length, stream).tobytes()))
                    if peeks and result_size and mnp != np.float64:  # the output rows in their order (dimension 0 and the values)
                        vo, no, w = offsets(cap)[0]
                        obs.append(("ordered_rows", b, _d2h(be, dims[1].ptr + vo, w * result_size, stream).tobytes()))
                        obs.append(("ordered_values", b, _d2h(be, meas[1].ptr, mb * result_size, stream).tobytes()))
            obs.append(("groups", b, result_size))
            if size and rng.random() < 0.12:  # ... and after it (work the reduction skipped must materialise)
                vo, no, w = offsets(cap)[nd - 1]
                obs.append(("dim_rows_after", b, _d2h(be, dims[0].ptr + vo + w * prev, w * size, stream).tobytes()))
                obs.append(("measure_rows_after", b, _d2h(be, meas[0].ptr + mb * prev, mb * size, stream).tobytes()))
            be.wait(stream)
            if not free_before:
                for x in list(cols.values()) + [idx, pred]:
                    x.free()
            dims.reverse(); meas.reverse(); hashes.reverse()
            if rng.random() < 0.25 and result_size:
                obs.append(("result", b, self._table(be, dims[0], meas[0], offsets(cap), result_size, mb, mnp, stream)))
            if len(streams) > 1:
                cur ^= 1
        if dims[0] is not None:
            obs.append(("final", nbatches, self._table(be, dims[0], meas[0], offsets(cap), result_size, mb, mnp, streams[cur])))
        for x in dims + meas + hashes + dimidx:
            if x is not None:
                x.free()
        if own_streams:
            for s in streams:
                be.call("DestroyCudaStream", s, 0)
        return obs

    def _dump(self, be, obs, want, cols, idx, pred, n, size, stream):
        out = os.path.join(os.environ["ARES_FUZZ_DUMP"], f"fuzz_mismatch_{os.getpid()}_{self.seed}_{len(obs)}")
        lines = [f"seed {self.seed} observation {len(obs) - 1}: got {obs[-1]} want {want}; n {n} size {size}"]
        arrays = {}
        for name, col in cols.items():
            dev = _d2h(be, col.buf.ptr, len(col.blob), stream)
            diff = np.flatnonzero(dev != col.blob)
            lines.append(f"column {name}: {len(col.blob)} bytes, {len(diff)} differ from the upload, first at {diff[:8].tolist()}")
            arrays["col_" + name + "_device"] = dev
            arrays["col_" + name + "_upload"] = col.blob
        arrays["pred"] = _d2h(be, pred.ptr, n, stream)
        arrays["idx"] = _d2h(be, idx.ptr, 4 * n, stream).view(np.uint32)  # (runs the pending compaction)
        lines.append(f"pred nonzero {int(np.count_nonzero(arrays['pred']))}; idx head {arrays['idx'][:8].tolist()}")
        np.savez_compressed(out + ".npz", **arrays)
        with open(out + ".txt", "w") as f:
            f.write("\n".join(lines) + "\n")

    @staticmethod
    def _table(be, dims, meas, offs, n, mb, mnp, stream):
        cols = [np.frombuffer(_d2h(be, dims.ptr + vo, w * n, stream), np.uint8).reshape(n, w) for vo, no, w in offs]
        oks = [_d2h(be, dims.ptr + no, n, stream) for vo, no, w in offs]
        m = _d2h(be, meas.ptr, mb * n, stream).view(mnp)
        return {tuple((c[r].tobytes(), int(o[r])) for c, o in zip(cols, oks)): m[r] for r in range(n)}


def _same(a, b, seed):
    assert len(a) == len(b), (seed, len(a), len(b))
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[1] == y[1], (seed, x[:2], y[:2])
        if isinstance(x[2], dict):
            assert x[2].keys() == y[2].keys(), (seed, x[0], x[1], len(x[2]), len(y[2]))
            for k, v in y[2].items():
                g = x[2][k]
                assert g == v or abs(float(g) - float(v)) <= 1e-6 * max(1.0, abs(float(v))), (seed, x[0], x[1], k, g, v)
        else:
            assert x[2] == y[2], (seed, x[0], x[1])


SEEDS = list(range(1000, 1160))
PROFILES = [(), ("narrow",), ("drift",), ("narrow", "drift"), ("prealloc",), ("prealloc", "narrow"), ("prealloc", "drift"),
            ("sort",), ("sort", "narrow"), ("sort", "prealloc"), ("sort", "drift"), ("sort", "eager"), ("sort", "eager", "prealloc")]


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(8))
def test_random_programs_match_the_oracle(chunk):
    hip, oracle = H.hip_backend(), H.oracle_backend()
    for k, seed in enumerate(SEEDS[chunk::8]):
        p = Program(seed, PROFILES[k % len(PROFILES)])
        want = p.run(oracle)
        _same(p.run(hip, expect=want), want, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("profile", [("sort", "eager"), ("sort", "eager", "prealloc")], ids=lambda p: "+".join(p))
def test_random_programs_sorting_rows_that_exist(profile):
    """48 more programs in the "eager" profiles alone: every reduction is a Sort + Reduce over rows the host has looked at,
    with the hash / index vectors and the output order observed (fused_sort_reduce_vectors and its replay)."""
    hip, oracle = H.hip_backend(), H.oracle_backend()
    for seed in range(3000, 3048):
        p = Program(seed, profile)
        want = p.run(oracle)
        _same(p.run(hip, expect=want), want, seed)


def test_random_programs_reference_build_matches_the_oracle():
    """The same programs on the reference's own HOST build: the fuzzer's expectations are the reference's."""
    if not H.have_ref():
        pytest.skip("reference HOST build absent")
    ref, oracle = H.ref_backend(), H.oracle_backend()
    for k, seed in enumerate(SEEDS[:48]):
        p = Program(seed, PROFILES[k % len(PROFILES)])
        _same(p.run(ref), p.run(oracle), seed)


@pytest.mark.gpu
def test_random_programs_from_four_host_threads():
    hip, oracle = H.hip_backend(), H.oracle_backend()
    for rnd in range(3):
        seeds = [2000 + 10 * rnd + t for t in range(4)]
        want = [Program(s).run(oracle) for s in seeds]
        got, errs = [None] * 4, []

        def work(t):
            try:
                got[t] = Program(seeds[t]).run(hip, expect=want[t])
            except Exception as e:  # noqa: BLE001
                errs.append((seeds[t], e))
        threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errs, errs
        for t in range(4):
            _same(got[t], want[t], seeds[t])


@pytest.mark.gpu
def test_soak_four_threads_stream_churn_clean_blocks():
    """The soak the deferral state machine's coverage used to live in, inside the suite the driver runs: 1024 programs from
    four host threads in ONE process (every program creates and destroys its own streams: stream churn with fences, error
    watches and lazy work outstanding), all four profiles — narrow columns, time-filter pairs whose prediction fails every
    batch —, ARES_MEM_VERIFY_CLEAN=1 (every block handed out as cleared is checked on the device: a writer that does not
    report what it wrote aborts the process).  No mismatch, no error, no abort."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "stress_fuzz.py"), "--iters", "256", "--threads", "4", "--seeds", "128",
                        "--profiles", "--tag", "soak"], cwd=H.ROOT, env={**os.environ, "ARES_MEM_VERIFY_CLEAN": "1"},
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    rep = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rep["programs"] >= 1000 and rep["mismatches"] == 0 and rep["errors"] == 0, rep


@pytest.mark.gpu
def test_random_programs_on_the_direct_kernels_with_table_images():
    """ARES_LEAN_MIN_GROUPS=0 sends every fusable batch of the (small) fuzz programs to the DIRECT-mode kernels and
    ARES_MIN_PART_BITS=2 gives them the four partitions the generated merges need, hence the table images: merges that start from the previous call's image, measure vectors that are only defined until the
    program copies them back (it does, at random points and at the end), result buffers that are reallocated and copied
    mid-query or sized once (the "prealloc" profiles), all seven profiles — 140 programs, four threads, every block checked clean."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "stress_fuzz.py"), "--iters", "35", "--threads", "4", "--seeds", "140",
                        "--profiles", "--kernels", "--tag", "images"], cwd=H.ROOT,
                       env={**os.environ, "ARES_MEM_VERIFY_CLEAN": "1", "ARES_LEAN_MIN_GROUPS": "0", "ARES_MIN_PART_BITS": "2"},
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    rep = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rep["programs"] == 140 and rep["mismatches"] == 0 and rep["errors"] == 0, rep
    k = rep["kernels"] or {}
    # the programs did go where this test means them to: generated scans and merges (a merge that STARTS from an image needs
    # two consecutive fusable batches on the same result buffers — the "prealloc" profiles; a handful of the 140 programs:
    # tests/test_table_image.py and the scale-parity variants are where that path is exercised batch after batch)
    assert k.get("hr_scan_rtc", 0) > 40 and k.get("hr_merge_rtc", 0) > 40, k


@pytest.mark.gpu
def test_random_programs_on_the_wide_sort_layout_with_two_levels():
    """ARES_SR_SCAN_FED=0 hands every Sort + Reduce of the "sort" profiles to the wide layout (rows written first), and
    ARES_SRV_PART_BITS=11 gives the small fuzz batches what production-sized ones get: 512 level-1 partitions dealt out to 2048
    (sr_count_kernel / sr_split_kernel with a fan-out of four), previous results found by their row hashes or hashed again —
    130 programs, four threads, every block checked clean, hash / index vectors and output order observed by the profiles."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "stress_fuzz.py"), "--iters", "33", "--threads", "4", "--seeds", "130",
                        "--profiles", "--kernels", "--tag", "wide"], cwd=H.ROOT,
                       env={**os.environ, "ARES_MEM_VERIFY_CLEAN": "1", "ARES_SR_SCAN_FED": "0", "ARES_SRV_PART_BITS": "11"},
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    rep = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rep["mismatches"] == 0 and rep["errors"] == 0 and rep["programs"] >= 130, rep
    k = rep["kernels"] or {}
    assert k.get("sr_split_kernel", 0) > 40 and k.get("sr_merge_kernel", 0) > 40, k
