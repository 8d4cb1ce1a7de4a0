"""Seeded, backend-agnostic test cases for the ABI.

Every case is a callable `run(be) -> dict[str, np.ndarray | int | ...]` that builds its inputs
with `be`'s own allocator, calls the ABI, and returns the observable outputs.  Two backends are
at parity on a case when the returned dicts are equal (see `assert_same`).  Inputs avoid what is
undefined behaviour in the reference itself (integer division by zero, INT_MIN / -1,
out-of-range float->int conversions), as the reference's HOST and DEVICE builds already
disagree there.
"""
import ctypes as C

import numpy as np

import harness as H
from aresdb_amd import abi

INT_TYPES = [abi.Int8, abi.Uint8, abi.Int16, abi.Uint16, abi.Int32, abi.Uint32]
COL_TYPES = [abi.Bool] + INT_TYPES + [abi.Float32]
UNARY_INT = [abi.Negate, abi.Not, abi.BitwiseNot, abi.IsNull, abi.IsNotNull, abi.Noop,
             abi.GetWeekStart, abi.GetMonthStart, abi.GetQuarterStart, abi.GetYearStart,
             abi.GetDayOfMonth, abi.GetDayOfYear, abi.GetMonthOfYear, abi.GetQuarterOfYear]
UNARY_FLOAT = [abi.Negate, abi.Not, abi.IsNull, abi.IsNotNull, abi.Noop, abi.GetMonthStart]
BINARY_ALL = list(range(abi.And, abi.Floor + 1))
BINARY_FLOAT_OK = list(range(abi.And, abi.Divide + 1)) + [abi.Mod, abi.Floor]  # last two: "return lhs"


def _rand_values(rng, dtype, n, small=False, nonzero=False, nonneg=False):
    if dtype == abi.Bool:
        return rng.integers(0, 2, n).astype(bool)
    if dtype == abi.Float32:
        v = (rng.random(n) * (90 if small else 1000) + (1 if nonzero else 0)).astype(np.float32)
        if not nonneg:
            v = v * rng.choice([-1, 1], n).astype(np.float32)
        v = np.round(v * 4) / 4  # exactly representable quarters
        return v.astype(np.float32)
    info = np.iinfo(H._NP_OF[dtype])
    lo, hi = int(info.min), int(info.max)
    if small:
        lo, hi = max(lo, -100), min(hi, 100)
    else:
        lo, hi = max(lo, -(1 << 30)), min(hi, (1 << 30))
    if nonneg:
        lo = max(lo, 0)
    v = rng.integers(lo, hi + 1, n)
    if nonzero:
        v[v == 0] = 1
    return v.astype(H._NP_OF[dtype])


class ColSpec:
    """Logical description of a column; materialised per backend."""

    def __init__(self, rng, dtype, mode, rows, **kw):
        self.dtype, self.mode, self.rows = dtype, mode, rows
        self.starting_index = int(rng.integers(0, 8)) if mode in (2, 3) or dtype == abi.Bool else 0
        if mode == 1 and dtype != abi.Bool:
            self.starting_index = 0
        self.default = None
        self.values = self.valid = self.counts = None
        if mode == 0:
            if rng.random() < 0.8:
                self.default = _rand_values(rng, dtype, 1, **kw)[0]
                if dtype != abi.Bool and dtype != abi.Float32:
                    self.default = int(self.default)
        elif mode in (1, 2):
            self.values = _rand_values(rng, dtype, rows, **kw)
            if mode == 2:
                self.valid = rng.random(rows) > 0.25
        else:  # run-length compressed: `runs` runs covering `rows` logical rows
            runs = max(1, int(rng.integers(1, max(2, rows // 2 + 1))))
            cuts = np.sort(rng.choice(np.arange(1, rows), size=min(runs - 1, rows - 1), replace=False)) \
                if rows > 1 else np.array([], np.int64)
            self.counts = np.concatenate([[0], cuts, [rows]]).astype(np.uint32)
            nruns = len(self.counts) - 1
            self.values = _rand_values(rng, dtype, nruns, **kw)
            self.valid = rng.random(nruns) > 0.25

    def build(self, be):
        if self.mode == 0:
            return H.Column(be, self.dtype, default=self.default)
        return H.Column(be, self.dtype, self.values, valid=self.valid, counts=self.counts,
                        starting_index=self.starting_index)


class OperandSpec:
    KINDS = ("col", "scratch", "cint", "cfloat")

    def __init__(self, rng, kind, rows, n, dtype=None, mode=None, **kw):
        self.kind, self.n = kind, n
        if kind == "col":
            self.col = ColSpec(rng, dtype if dtype is not None else rng.choice(COL_TYPES),
                               mode if mode is not None else int(rng.integers(0, 4)), rows, **kw)
        elif kind == "scratch":
            self.dtype = dtype if dtype in (abi.Int32, abi.Uint32, abi.Float32) else \
                [abi.Int32, abi.Uint32, abi.Float32][int(rng.integers(0, 3))]
            self.values = _rand_values(rng, self.dtype, n, **kw)
            self.valid = rng.random(n) > 0.25
        elif kind == "cint":
            self.value = int(_rand_values(rng, abi.Int32, 1, **kw)[0])
            self.valid = bool(rng.random() > 0.1)
        else:
            self.value = float(_rand_values(rng, abi.Float32, 1, **kw)[0])
            self.valid = bool(rng.random() > 0.1)

    def is_float(self):
        if self.kind == "col":
            return self.col.dtype == abi.Float32
        if self.kind == "scratch":
            return self.dtype == abi.Float32
        return self.kind == "cfloat"

    def uses_rle(self):
        return self.kind == "col" and self.col.mode == 3

    def build(self, be, keep):
        if self.kind == "col":
            c = self.col.build(be)
            keep.append(c)
            return c.input()
        if self.kind == "scratch":
            s = H.Scratch(be, self.n, self.dtype, self.values, self.valid)
            keep.append(s)
            return s.input()
        if self.kind == "cint":
            return H.const_int(self.value, self.valid)
        return H.const_float(self.value, self.valid)


class OutSpec:
    def __init__(self, kind, dtype, agg=None, offset=0):
        # offset: the output starts `offset` elements into its buffers (the Go host writes batch
        # k's dimensions at row offset resultSize, so value / validity pointers are rarely aligned)
        self.kind, self.dtype, self.agg, self.offset = kind, dtype, agg, offset

    def build(self, be, n, keep):
        if self.kind == "scratch":
            s = H.Scratch(be, n, self.dtype)
            keep.append(s)
            return s.output(), lambda: {"values": s.buf.read(np.uint8, 4 * n), "valid": s.valid()}
        w = abi.DATA_TYPE_BYTES[self.dtype]
        off = self.offset
        if self.kind == "dim":
            vb = H.Buf(be, nbytes=w * (n + off) + 8)
            nb = H.Buf(be, nbytes=n + off + 8)
            keep.extend([vb, nb])
            # the whole buffers are compared: bytes around the output range must stay untouched
            return H.dimension_output(vb.ptr + w * off, nb.ptr + off, self.dtype), \
                lambda: {"values": vb.read(np.uint8, w * (n + off) + 8), "valid": nb.read(np.uint8, n + off + 8)}
        vb = H.Buf(be, nbytes=w * (n + off) + 8)
        keep.append(vb)
        return H.measure_output(vb.ptr + w * off, self.dtype, self.agg), \
            lambda: {"values": vb.read(np.uint8, w * (n + off) + 8)}


def make_index(rng, rows, style):
    """Index vectors as the filters leave them (sorted subsets) plus a permuted one."""
    if style == "identity":
        return np.arange(rows, dtype=np.uint32)
    if style == "subset":
        keep = rng.random(rows) > 0.4
        idx = np.nonzero(keep)[0].astype(np.uint32)
        return idx if len(idx) else np.array([0], np.uint32)
    return rng.permutation(rows).astype(np.uint32)


class TransformCase:
    """Unary/BinaryTransform and Unary/BinaryFilter."""

    def __init__(self, seed, arity, as_filter, rows=None, index_style=None):
        rng = np.random.default_rng(seed)
        self.seed, self.arity, self.as_filter = seed, arity, as_filter
        self.rows = rows if rows is not None else int(rng.integers(1, 300))
        style = index_style or ["identity", "subset", "perm"][int(rng.integers(0, 3))]
        if as_filter and style == "perm":
            style = "subset"
        self.index = make_index(rng, self.rows, style)
        self.n = len(self.index)
        n = self.n
        self.base_counts = None
        self.start_count = 0
        self.pred_offset = 0
        if arity == 1:
            kind = ["col", "col", "col", "scratch", "cint"][int(rng.integers(0, 5))]
            a = OperandSpec(rng, kind, self.rows, n, small=True, nonneg=True)
            self.ops = [a]
            self.functor = int(rng.choice(UNARY_FLOAT if a.is_float() else UNARY_INT))
            if kind == "col" and a.col.dtype not in (abi.Float32, abi.Bool) and \
                    self.functor >= abi.GetWeekStart:
                # calendar functors: realistic epoch seconds
                a.col.values = None if a.col.mode == 0 else \
                    rng.integers(0, 1 << 31, len(a.col.values)).astype(H._NP_OF[a.col.dtype]) \
                    if a.col.dtype in (abi.Int32, abi.Uint32) else a.col.values
        else:
            ka = ["col", "col", "scratch", "cint", "cfloat"][int(rng.integers(0, 5))]
            kb = ["col", "scratch", "cint", "cfloat"][int(rng.integers(0, 4))]
            if ka in ("cint", "cfloat") and kb in ("cint", "cfloat"):
                kb = "col"
            a = OperandSpec(rng, ka, self.rows, n, small=True)
            b = OperandSpec(rng, kb, self.rows, n, small=True, nonzero=True)
            if b.kind == "col" and b.col.dtype == abi.Bool:  # bool divisors may be 0
                b = OperandSpec(rng, "col", self.rows, n, dtype=abi.Int16, small=True, nonzero=True)
            self.ops = [a, b]
            anyf = a.is_float() or b.is_float()
            self.functor = int(rng.choice(BINARY_FLOAT_OK if anyf else BINARY_ALL))
        if any(o.uses_rle() for o in self.ops):
            # logical row r of the batch maps to compressed position via baseCounts/startCount
            if rng.random() < 0.5:
                self.start_count = 0
            else:
                # identity baseCounts vector (uncompressed base column)
                self.base_counts = np.arange(self.rows + 1, dtype=np.uint32)
        anyf = any(o.is_float() for o in self.ops)
        if as_filter:
            self.out = None
            self.num_foreign = int(rng.integers(0, 3))
            self.rids = [np.stack([rng.integers(1, 5, n), rng.integers(0, 1000, n)], 1).astype(np.uint32)
                         for _ in range(self.num_foreign)]
        else:
            choice = int(rng.integers(0, 3))
            if choice == 0:
                dt = [abi.Int32, abi.Uint32, abi.Float32][int(rng.integers(0, 3))]
                self.out = OutSpec("scratch", dt)
            elif choice == 1:
                dts = [abi.Bool, abi.Int8, abi.Uint8, abi.Int16, abi.Uint16, abi.Int32, abi.Uint32, abi.Float32]
                self.out = OutSpec("dim", dts[int(rng.integers(0, len(dts)))])
            else:
                dts = [abi.Int32, abi.Uint32, abi.Float32, abi.Int64, abi.Float64]
                aggs = [abi.AGGR_SUM_UNSIGNED, abi.AGGR_SUM_SIGNED, abi.AGGR_SUM_FLOAT,
                        abi.AGGR_MIN_UNSIGNED, abi.AGGR_MIN_SIGNED, abi.AGGR_MAX_UNSIGNED,
                        abi.AGGR_MAX_SIGNED, abi.AGGR_AVG_FLOAT]
                agg = aggs[int(rng.integers(0, len(aggs)))]
                dt = dts[int(rng.integers(0, len(dts)))]
                if agg == abi.AGGR_AVG_FLOAT:
                    dt = abi.Float64
                self.out = OutSpec("measure", dt, agg)
                if self.base_counts is None and rng.random() < 0.5 and not any(o.uses_rle() for o in self.ops):
                    # compressed base column: run lengths multiply SUM/AVG
                    lens = rng.integers(1, 5, self.rows)
                    self.base_counts = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
            # out-of-range float -> integer conversions are UB in the reference (negative ->
            # unsigned, |x| >= 2^(bits-1)): floats only meet 32/64-bit signed or float sinks
            if anyf and self.out.dtype in (abi.Int8, abi.Uint8, abi.Int16, abi.Uint16, abi.Uint32):
                self.out.dtype = abi.Int32

    def __repr__(self):
        return f"TransformCase(seed={self.seed}, arity={self.arity}, filter={self.as_filter})"

    def run(self, be):
        keep = []
        ins = [o.build(be, keep) for o in self.ops]
        idx = H.Buf(be, self.index)
        bc = H.Buf(be, self.base_counts) if self.base_counts is not None else None
        bcp = bc.ptr if bc else None
        n = self.n
        res = {}
        if self.as_filter:
            po = self.pred_offset
            pred = H.Buf(be, nbytes=n + po + 8)
            rbufs = [H.Buf(be, r) for r in self.rids]
            vecs = (C.c_void_p * max(1, len(rbufs)))(*[b.ptr for b in rbufs])
            name = "UnaryFilter" if self.arity == 1 else "BinaryFilter"
            cnt = be.call(name, *ins, idx.ptr, pred.ptr + po, n, C.addressof(vecs) if rbufs else None,
                          len(rbufs), bcp, self.start_count, self.functor, None, 0)
            res["count"] = cnt
            res["pred"] = pred.read(np.uint8, n + po + 8)
            res["index"] = idx.read(np.uint32, cnt)
            for i, b in enumerate(rbufs):
                res[f"rid{i}"] = b.read(np.uint32, 2 * cnt)
            keep.extend([pred] + rbufs)
        else:
            ov, reader = self.out.build(be, n, keep)
            name = "UnaryTransform" if self.arity == 1 else "BinaryTransform"
            res["ret"] = be.call(name, *ins, ov, idx.ptr, n, bcp, self.start_count, self.functor,
                                 None, 0)
            res.update(reader())
        for k in keep + [idx] + ([bc] if bc else []):
            k.free()
        return res


def fast_path_case(seed, as_filter, rows, style):
    """The hot live-batch shape the HIP library serves with its vectorised kernels: a 4-byte column
    (values only or validity + values) against a constant (or bare, for transforms), written to a
    4-byte dimension / scratch vector or a measure at an arbitrary row offset."""
    rng = np.random.default_rng(77000 + seed)
    c = TransformCase(seed, 2, as_filter, rows=rows, index_style=style)
    dtype = [abi.Uint32, abi.Int32, abi.Float32][seed % 3]
    mode = 1 + (seed // 3) % 2
    a = OperandSpec(rng, "col", c.rows, c.n, dtype=dtype, mode=mode, small=True, nonneg=(seed % 5 != 0))
    c.base_counts, c.start_count = None, 0
    unary = (not as_filter) and seed % 4 == 0
    if unary:
        c.arity, c.ops, c.functor = 1, [a], abi.Noop
    else:
        kb = "cfloat" if (dtype == abi.Float32 or seed % 7 == 0) else "cint"
        b = OperandSpec(rng, kb, c.rows, c.n, small=True, nonzero=True)
        b.valid = seed % 11 != 0
        c.arity, c.ops = 2, [a, b]
        anyf = a.is_float() or b.is_float()
        if as_filter:
            c.functor = int(rng.integers(abi.Equal, abi.GreaterThanOrEqual + 1))
        else:
            c.functor = int(rng.choice(BINARY_FLOAT_OK if anyf else BINARY_ALL))
    anyf = any(o.is_float() for o in c.ops)
    if as_filter:
        c.pred_offset = seed % 4
        c.out = None
    else:
        off = int(rng.integers(0, 9))
        k = seed % 3
        if k == 0:
            c.out = OutSpec("scratch", abi.Int32 if anyf else [abi.Int32, abi.Uint32][seed % 2])
        elif k == 1:
            c.out = OutSpec("dim", abi.Int32 if anyf else [abi.Uint32, abi.Int32][seed % 2], offset=off)
        else:
            dt = [abi.Float64, abi.Int64, abi.Float32, abi.Int32][(seed // 3) % 4] if anyf else \
                [abi.Float64, abi.Uint32, abi.Int64, abi.Int32, abi.Float32][(seed // 3) % 5]
            agg = [abi.AGGR_SUM_FLOAT, abi.AGGR_SUM_SIGNED, abi.AGGR_MIN_SIGNED, abi.AGGR_MAX_UNSIGNED][seed % 4]
            c.out = OutSpec("measure", dt, agg, offset=off)
    return c


# ---- cuckoo index builder (layout of memstore/cuckoo_index.go; see oracle HashLookup) ----------
def _murmur3_32(key: bytes, seed: int) -> int:
    h = seed & 0xFFFFFFFF
    nb = len(key) // 4
    for i in range(nb):
        k = int.from_bytes(key[4 * i:4 * i + 4], "little")
        k = (k * 0xcc9e2d51) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * 0x1b873593) & 0xFFFFFFFF
        h ^= k
        h = ((h << 13) | (h >> 19)) & 0xFFFFFFFF
        h = (h * 5 + 0xe6546b64) & 0xFFFFFFFF
    tail = key[4 * nb:]
    k = 0
    if len(tail) >= 3:
        k ^= tail[2] << 16
    if len(tail) >= 2:
        k ^= tail[1] << 8
    if len(tail) >= 1:
        k ^= tail[0]
        k = (k * 0xcc9e2d51) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * 0x1b873593) & 0xFFFFFFFF
        h ^= k
    h ^= len(key)
    h ^= h >> 16
    h = (h * 0x85ebca6b) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xc2b2ae35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def build_cuckoo(keys, key_bytes, num_buckets, seeds, rng, record_of=None):
    """Places each key in the first free slot among its 4 candidate buckets, else the stash.
    (Insertion policy does not matter to the probe; only the layout does.)"""
    bucket_bytes = 8 * (8 + 1 + key_bytes)
    table = np.zeros((num_buckets + 1) * bucket_bytes, np.uint8)
    placed = {}
    for ki, key in enumerate(keys):
        rec = record_of(ki) if record_of else (1 + ki // 1000, ki % 1000)
        done = False
        order = list(range(len(seeds)))
        rng.shuffle(order)
        for h in order:
            hv = _murmur3_32(key, seeds[h])
            b = hv % num_buckets
            sig = max(1, hv >> 24)
            base = b * bucket_bytes
            for j in range(8):
                if table[base + 64 + j] == 0:
                    table[base + 8 * j: base + 8 * j + 8] = np.array(rec, np.uint32).view(np.uint8)
                    table[base + 64 + j] = sig
                    table[base + 72 + j * key_bytes: base + 72 + (j + 1) * key_bytes] = \
                        np.frombuffer(key, np.uint8)
                    done = True
                    break
            if done:
                break
        if not done:
            base = num_buckets * bucket_bytes
            for j in range(4):
                if table[base + 64 + j] == 0:
                    table[base + 8 * j: base + 8 * j + 8] = np.array(rec, np.uint32).view(np.uint8)
                    table[base + 64 + j] = 1
                    table[base + 72 + j * key_bytes: base + 72 + (j + 1) * key_bytes] = \
                        np.frombuffer(key, np.uint8)
                    done = True
                    break
        if done:
            placed[key] = rec
    return table, placed


class HashLookupCase:
    def __init__(self, seed, n=None, nkeys=None):
        rng = np.random.default_rng(seed)
        self.seed = seed
        self.dtype = [abi.Uint32, abi.Int32, abi.Uint16, abi.Uint8, abi.Int64, abi.UUID][int(rng.integers(0, 6))]
        self.key_bytes = {abi.Uint32: 4, abi.Int32: 4, abi.Uint16: 2, abi.Uint8: 1, abi.Int64: 8,
                          abi.UUID: 16}[self.dtype]
        nkeys = nkeys or int(rng.integers(1, 200))
        if self.key_bytes == 1:
            nkeys = min(nkeys, 100)
        space = min(1 << (8 * min(self.key_bytes, 4)), 1 << 20)
        raw = rng.choice(space, size=min(nkeys, space // 2), replace=False)
        self.keys = [int(k).to_bytes(self.key_bytes, "little") for k in raw]
        self.num_buckets = max(1, len(self.keys) // 5)  # dense enough to exercise the stash
        self.seeds = [int(x) for x in rng.integers(0, 1 << 32, 4)]
        self.table, self.placed = build_cuckoo(self.keys, self.key_bytes, self.num_buckets,
                                               self.seeds, rng)
        self.rows = n or int(rng.integers(1, 400))
        probe = rng.choice(space, size=self.rows)  # hits and misses
        hit = rng.random(self.rows) < 0.6
        probe[hit] = rng.choice(raw, size=int(hit.sum()))
        self.probe = probe
        self.valid = rng.random(self.rows) > 0.1
        self.index = make_index(rng, self.rows, "subset")

    def run(self, be):
        if self.dtype in (abi.Int64, abi.UUID):
            w = abi.DATA_TYPE_BYTES[self.dtype]
            rawv = b"".join(int(p).to_bytes(w, "little") for p in self.probe)
            col = H.Column(be, self.dtype, raw_values=rawv, valid=self.valid)
        else:
            col = H.Column(be, self.dtype, self.probe.astype(np.int64).astype(H._NP_OF[self.dtype]),
                           valid=self.valid)
        tb = H.Buf(be, self.table)
        hi = abi.CuckooHashIndex()
        hi.buckets = tb.ptr
        for i, s in enumerate(self.seeds):
            hi.seeds[i] = s
        hi.keyBytes, hi.numHashes, hi.numBuckets = self.key_bytes, 4, self.num_buckets
        idx = H.Buf(be, self.index)
        n = len(self.index)
        out = H.Buf(be, np.full(2 * n, 0xAAAAAAAA, np.uint32))
        be.call("HashLookup", col.input(), out.ptr, idx.ptr, n, None, 0, hi, None, 0)
        res = {"rids": out.read(np.uint32, 2 * n)}
        for b in (col, tb, idx, out):
            b.free()
        return res


class GroupByCase:
    """Dimension vector + measures -> Sort/Reduce and HashReduce."""

    def __init__(self, seed, length=None, groups=None, ndw=None, agg=None, value_bytes=None,
                 capacity_slack=None):
        rng = np.random.default_rng(seed)
        self.seed = seed
        self.ndw = ndw or tuple(int(x) for x in rng.integers(0, 3, 5))
        if sum(self.ndw) == 0:
            self.ndw = (0, 0, 1, 0, 0)
        while sum(w * c for w, c in zip(H.DIM_WIDTHS, self.ndw)) + sum(self.ndw) > 32:
            self.ndw = tuple(max(0, c - 1) if i < 2 else c for i, c in enumerate(self.ndw))
        self.length = length if length is not None else int(rng.integers(1, 2000))
        self.capacity = self.length + (capacity_slack if capacity_slack is not None else int(rng.integers(0, 50)))
        ngroups = groups or max(1, int(rng.integers(1, max(2, self.length // 3 + 1))))
        # distinct group rows
        self.offsets = []
        value_bytes_total = sum(w * c for w, c in zip(H.DIM_WIDTHS, self.ndw))
        ndims = sum(self.ndw)
        # distinct prototype rows, vectorised: per dimension an (ngroups, width) byte matrix (10% of
        # the values all-zero) and a validity column (10% nulls)
        widths = [w for w, c in zip(H.DIM_WIDTHS, self.ndw) for _ in range(c)]
        proto_vals = []
        for w in widths:
            m = rng.integers(0, 256, (ngroups, w), dtype=np.uint8)
            m[rng.random(ngroups) >= 0.9] = 0
            proto_vals.append(m)
        proto_valid = (rng.random((ngroups, ndims)) > 0.1).astype(np.uint8)
        pick = rng.integers(0, ngroups, self.length)
        blob = np.zeros((value_bytes_total + ndims) * self.capacity, np.uint8)
        off = 0
        for d, w in enumerate(widths):
            blob[off:off + w * self.length] = proto_vals[d][pick].reshape(-1)
            noff = value_bytes_total * self.capacity + d * self.capacity
            blob[noff:noff + self.length] = proto_valid[pick, d]
            off += w * self.capacity
        self.blob = blob
        aggs = [abi.AGGR_SUM_UNSIGNED, abi.AGGR_SUM_SIGNED, abi.AGGR_SUM_FLOAT, abi.AGGR_MIN_UNSIGNED,
                abi.AGGR_MIN_SIGNED, abi.AGGR_MIN_FLOAT, abi.AGGR_MAX_UNSIGNED, abi.AGGR_MAX_SIGNED,
                abi.AGGR_MAX_FLOAT, abi.AGGR_AVG_FLOAT]
        self.agg = agg if agg is not None else aggs[int(rng.integers(0, len(aggs)))]
        is_sum = self.agg in (abi.AGGR_SUM_UNSIGNED, abi.AGGR_SUM_SIGNED, abi.AGGR_SUM_FLOAT)
        self.value_bytes = value_bytes or (int(rng.choice([4, 8])) if is_sum else
                                           (8 if self.agg == abi.AGGR_AVG_FLOAT else 4))
        n = self.length
        if self.agg == abi.AGGR_AVG_FLOAT:
            v = np.zeros(2 * n, np.uint32)
            v[0::2] = (rng.integers(0, 400, n) / 4).astype(np.float32).view(np.uint32)
            v[1::2] = rng.integers(1, 4, n)
            self.values = v.view(np.uint8)
        elif self.agg in (abi.AGGR_SUM_FLOAT, abi.AGGR_MIN_FLOAT, abi.AGGR_MAX_FLOAT):
            f = (rng.integers(-4000, 4000, n) / 8)
            self.values = (f.astype(np.float64) if self.value_bytes == 8 else f.astype(np.float32)).view(np.uint8)
        elif self.agg in (abi.AGGR_SUM_SIGNED, abi.AGGR_MIN_SIGNED, abi.AGGR_MAX_SIGNED):
            i = rng.integers(-100000, 100000, n)
            self.values = (i.astype(np.int64) if self.value_bytes == 8 else i.astype(np.int32)).view(np.uint8)
        else:
            i = rng.integers(0, 100000, n)
            self.values = (i.astype(np.uint64) if self.value_bytes == 8 else i.astype(np.uint32)).view(np.uint8)
        self.vb = 8 if self.value_bytes == 8 else 4

    def _dims(self, be, with_sort):
        din = H.DimVector(be, self.capacity, self.ndw, with_hash=with_sort, with_index=with_sort,
                          init=self.blob)
        dout = H.DimVector(be, self.capacity, self.ndw, with_hash=with_sort, with_index=with_sort)
        return din, dout

    def run_sort_reduce(self, be):
        din, dout = self._dims(be, True)
        vin = H.Buf(be, self.values)
        vout = H.Buf(be, nbytes=self.vb * self.capacity + 8)
        be.call("InitIndexVector", din.index.ptr, 0, self.length, None, 0)
        be.call("Sort", din.struct(), self.length, None, 0)
        res = {"sorted_hash": din.hash.read(np.uint64, self.length),
               "sorted_index": din.index.read(np.uint32, self.length)}
        g = be.call("Reduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, self.value_bytes,
                    self.length, self.agg, None, 0)
        res["groups"] = g
        res["out_index"] = dout.index.read(np.uint32, g)
        res["out_values"] = vout.read(np.uint8, self.vb * g)
        res["out_dims"] = dout.rows(g)
        for b in (din, dout, vin, vout):
            b.free()
        return res

    def run_hash_reduce(self, be):
        din, dout = self._dims(be, False)
        vin = H.Buf(be, self.values)
        vout = H.Buf(be, nbytes=self.vb * self.capacity + 8)
        g = be.call("HashReduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, self.value_bytes,
                    self.length, self.agg, None, 0)
        vals = vout.read(np.uint8, self.vb * g).reshape(g, self.vb)
        rows = dout.rows(g)
        res = {"groups": g, "map": {r: bytes(v) for r, v in zip(rows, vals)}}
        for b in (din, dout, vin, vout):
            b.free()
        return res

    def value_dtype(self):
        if self.agg == abi.AGGR_AVG_FLOAT:
            return "avg"
        if self.agg in (abi.AGGR_SUM_FLOAT, abi.AGGR_MIN_FLOAT, abi.AGGR_MAX_FLOAT):
            return np.float64 if self.vb == 8 else np.float32
        if self.agg in (abi.AGGR_SUM_SIGNED, abi.AGGR_MIN_SIGNED, abi.AGGR_MAX_SIGNED):
            return np.int64 if self.vb == 8 else np.int32
        return np.uint64 if self.vb == 8 else np.uint32


class HllCase:
    """Several batches through HyperLogLog, replaying the Go host's protocol
    (query/aql_batchexecutor.go:219-233, query/aql_processor.go:717-723): dimension rows of the
    batch behind the previous results in dim[0], hll values in measure[1], index vectors
    re-initialised per batch, dim / measure / hash vectors swapped after every call."""

    def __init__(self, seed, batches=None, batch_rows=None, groups=None, registers=None, ndw=None):
        rng = np.random.default_rng(seed)
        self.seed = seed
        self.ndw = ndw or [(0, 0, 1, 0, 0), (0, 0, 0, 0, 1), (0, 0, 1, 1, 1), (0, 1, 0, 0, 2), (1, 0, 1, 0, 0)][seed % 5]
        self.widths = [w for w, c in zip(H.DIM_WIDTHS, self.ndw) for _ in range(c)]
        nb = batches or int(rng.integers(1, 5))
        self.sizes = [int(batch_rows if batch_rows is not None else rng.integers(0, 3000)) for _ in range(nb)]
        ngroups = groups or int(rng.integers(1, 40))
        nregs = registers or int(rng.choice([3, 50, 5000, 1 << 14]))
        ndims = len(self.widths)
        proto_vals = [rng.integers(0, 256, (ngroups, w), dtype=np.uint8) for w in self.widths]
        proto_valid = (rng.random((ngroups, ndims)) > 0.1).astype(np.uint8)
        for d in range(ndims):  # a null dimension carries zero bytes, as the transforms write it
            proto_vals[d][proto_valid[:, d] == 0] = 0
        self.batches = []
        for n in self.sizes:
            pick = rng.integers(0, ngroups, n)
            if ngroups > 1 and n:  # group 0 is hot: it may cross the dense threshold
                pick[rng.random(n) < 0.5] = 0
            regs = rng.integers(0, nregs, n).astype(np.uint32)
            rho = rng.integers(0, 51, n).astype(np.uint32)
            vals = (rho << 16) | regs
            self.batches.append(([proto_vals[d][pick] for d in range(ndims)], proto_valid[pick], vals))
        self.capacity = sum(self.sizes) + int(rng.integers(1, 20))

    def __repr__(self):
        return f"HllCase(seed={self.seed}, ndw={self.ndw}, sizes={self.sizes})"

    def run(self, be, keep_state=True):
        cap = self.capacity
        dims = [H.DimVector(be, cap, self.ndw), H.DimVector(be, cap, self.ndw)]
        meas = [H.Buf(be, nbytes=4 * cap), H.Buf(be, nbytes=4 * cap)]
        offs = dims[0].dim_offsets()
        res, size = {}, 0
        for b, (vals, valid, hll) in enumerate(self.batches):
            n = len(hll)
            last = b == len(self.batches) - 1
            for d, (vo, no, w) in enumerate(offs):
                if n:
                    dims[0].values.write(vals[d].reshape(-1), offset=vo + size * w)
                    dims[0].values.write(valid[:, d].copy(), offset=no + size)
            if n:
                meas[1].write(hll)
            # the index vectors are never swapped: [0] numbers the previous results, [1] the batch
            be.call("InitIndexVector", dims[0].index.ptr, 0, size, None, 0)
            be.call("InitIndexVector", dims[1].index.ptr, size, n, None, 0)
            prev, cur = dims[0], dims[1]
            dv_prev, dv_cur = prev.struct(), cur.struct()
            dv_prev.IndexVector, dv_cur.IndexVector = dims[0].index.ptr, dims[1].index.ptr
            g, vec, reg = H.hyperloglog(be, _Fixed(dv_prev), _Fixed(dv_cur), meas[0], meas[1], size, n, last)
            res[f"size{b}"] = g
            if not (last and size + n > 0) and keep_state:
                res[f"hash{b}"] = cur.hash.read(np.uint64, g)
                res[f"values{b}"] = meas[1].read(np.uint32, g)
                res[f"index{b}"] = dims[1].index.read(np.uint32, g)
            res[f"dims{b}"] = cur.rows(g) if g < 4000 else cur.values.read(np.uint8)
            if vec is not None:
                res["hll"], res["reg_counts"] = vec, reg
            size = g
            # swapResultBufferForNextBatch: everything but the index vectors
            idx0, idx1 = dims[0].index, dims[1].index
            dims.reverse()
            meas.reverse()
            dims[0].index, dims[1].index = idx0, idx1
        for x in dims + meas:
            x.free()
        return res


class GeoCase:
    """Random polygons (one or two rings each) x random points through GeoBatchIntersects and
    WriteGeoShapeDim: non-identity index vectors, null points, RecordID vectors that must be
    compacted alongside, up to 200 shapes (shape numbers >= 128 wrap negative in the reference's
    int8 predicate iterator), points read from the main table or through a join."""

    def __init__(self, seed, rows=None, shapes=None, foreign_points=None):
        rng = np.random.default_rng(seed)
        self.seed = seed
        self.num_shapes = shapes or int(rng.choice([1, 3, 33, 70, 200]))
        lats, longs, sidx = [], [], []
        flt_max = np.finfo(np.float32).max
        for sh in range(self.num_shapes):
            cx, cy = rng.uniform(-50, 50, 2)
            for ring in range(int(rng.integers(1, 3))):
                k = int(rng.integers(3, 9))
                ang = np.sort(rng.uniform(0, 2 * np.pi, k))
                rad = rng.uniform(2, 30) / (1 + 2 * ring)
                ys = (cy + rad * np.sin(ang)).astype(np.float32)
                xs = (cx + rad * np.cos(ang)).astype(np.float32)
                if ring:
                    lats.append(flt_max), longs.append(flt_max), sidx.append(sh)
                lats += list(ys) + [ys[0]]
                longs += list(xs) + [xs[0]]
                sidx += [sh] * (k + 1)
        self.lats, self.longs, self.sidx = np.float32(lats), np.float32(longs), np.uint8(sidx)
        self.length = int(rows if rows is not None else rng.integers(1, 3000))
        self.points = rng.uniform(-60, 60, (self.length, 2)).astype(np.float32)
        # a few points exactly on polygon vertices
        for i in rng.integers(0, self.length, min(5, self.length)):
            j = int(rng.integers(0, len(lats)))
            if lats[j] < flt_max:
                self.points[i] = (lats[j], longs[j])
        self.valid = None if seed % 4 == 0 else (rng.random(self.length) > 0.1)
        self.starting_index = int(rng.integers(0, 8)) if self.valid is not None else 0
        style = ["identity", "subset", "perm"][seed % 3]
        self.foreign_points = bool(seed % 5 == 3) if foreign_points is None else foreign_points
        if self.foreign_points:
            style = "identity"
        self.index = make_index(rng, self.length, style)
        self.n = len(self.index)
        self.in_or_out = bool(seed % 2 == 0)
        self.num_foreign = int(rng.integers(0, 3))
        self.rids = [H.record_id_array([(int(b), int(x)) for b, x in zip(rng.integers(-5, 5, self.n), rng.integers(0, 1000, self.n))])
                     for _ in range(self.num_foreign)]
        self.words = (self.num_shapes + 31) // 32
        self.prefill = rng.integers(0, 1 << 32, self.n * self.words, dtype=np.uint64).astype(np.uint32) \
            if seed % 7 == 0 else None
        self.batch_of = rng.integers(0, 3, self.length)  # joined table: 3 batches

    def __repr__(self):
        return f"GeoCase(seed={self.seed}, shapes={self.num_shapes}, rows={self.length}, n={self.n})"

    def run(self, be):
        shapes = H.GeoShapes(be, self.lats, self.longs, self.sidx, self.num_shapes)
        keep = [shapes]
        idx = H.Buf(be, self.index)
        rid_bufs = [H.Buf(be, r) for r in self.rids]
        vecs = (C.c_void_p * max(self.num_foreign, 1))(*[b.ptr for b in rid_bufs]) if self.num_foreign else None
        pred = H.Buf(be, self.prefill) if self.prefill is not None else H.Buf(be, nbytes=4 * self.n * self.words)
        if self.foreign_points:
            # three batches hold the points; entry i joins record (batch_of[i], i); batch 0 of the
            # join is a constant (mode 0) batch without a default: its points are null
            slices = (abi.VectorPartySlice * 3)()
            for b in range(3):
                if b == 0:
                    slices[b].BasePtr, slices[b].DataType = None, abi.GeoPoint
                    continue
                col = H.geo_column(be, self.points, valid=self.valid, starting_index=self.starting_index)
                keep.append(col)
                slices[b] = col.vp
            point_rids = H.Buf(be, H.record_id_array([(7 + int(self.batch_of[r]), int(r)) for r in self.index]))
            keep.append(point_rids)
            iv = abi.InputVector()
            f = iv.Vector.ForeignVP
            f.RecordIDs, f.Batches = point_rids.ptr, C.addressof(slices)
            f.BaseBatchID, f.NumBatches, f.NumRecordsInLastBatch = 7, 3, self.length
            f.TimezoneLookup, f.TimezoneLookupSize, f.DataType = None, 0, abi.GeoPoint
            iv.Type = abi.ForeignColumnInput
        else:
            col = H.geo_column(be, self.points, valid=self.valid, starting_index=self.starting_index)
            keep.append(col)
            iv = col.input()
        kept = be.call("GeoBatchIntersects", shapes.struct(), iv, idx.ptr, self.n, 0,
                       C.addressof(vecs) if self.num_foreign else None, self.num_foreign, pred.ptr, self.in_or_out,
                       None, 0)
        dim = H.Buf(be, nbytes=2 * self.n + 16)
        dv = abi.DimensionOutputVector()
        dv.DimValues, dv.DimNulls, dv.DataType = dim.ptr, dim.ptr + self.n + 8, abi.Uint8
        be.call("WriteGeoShapeDim", self.words, dv, self.n, pred.ptr, None, 0)
        be.wait()
        words = pred.read(np.uint32, self.n * self.words).reshape(self.n, self.words)
        first = np.full(self.n, -1, np.int64)
        for w in range(self.words - 1, -1, -1):
            col_w = words[:, w]
            low = np.where(col_w != 0, np.log2((col_w & -col_w.astype(np.int64)).astype(np.float64) + (col_w == 0)), -1)
            first = np.where(col_w != 0, w * 32 + low.astype(np.int64), first)
        inside = int(np.count_nonzero((first >= 0) & (first < 128)))
        res = {"kept": kept, "pred": words.copy(), "index": idx.read(np.uint32, kept), "inside": inside,
               "dim_values": dim.read(np.uint8, inside), "dim_nulls": dim.read(np.uint8, inside, self.n + 8)}
        for t, b in enumerate(rid_bufs):
            res[f"rids{t}"] = b.read(np.uint64, kept)
        for b in keep + [idx, pred, dim] + rid_bufs:
            b.free()
        return res


def hll_estimate_check(be, per_batch, distinct, tolerance=0.04):
    """GetHLLValue measure transform + HyperLogLog over two batches of a one-dimension query whose
    groups have known distinct counts; decodes the sparse / dense registers like the host does
    (query/common/hll.go:547-575) and checks the classic HyperLogLog estimate (p = 14)."""
    rng = np.random.default_rng(5)
    cap = 2 * per_batch + 64
    dims = [H.DimVector(be, cap, (0, 0, 1, 0, 0)), H.DimVector(be, cap, (0, 0, 1, 0, 0))]
    meas = [H.Buf(be, nbytes=4 * cap), H.Buf(be, nbytes=4 * cap)]
    size = 0
    seen = []
    for b in range(2):
        group = rng.integers(0, len(distinct), per_batch).astype(np.uint32)
        user = (rng.integers(0, 1 << 62, per_batch) % np.array(distinct)[group]).astype(np.uint32)
        seen.append(group.astype(np.uint64) << np.uint64(32) | user)
        col = H.Column(be, abi.Uint32, user)
        dims[0].values.write(group, offset=4 * size)
        dims[0].values.write(np.ones(per_batch, np.uint8), offset=4 * cap + size)
        be.call("InitIndexVector", dims[1].index.ptr, 0, per_batch, None, 0)
        be.call("UnaryTransform", col.input(), H.measure_output(meas[1].ptr, abi.Uint32, abi.AGGR_HLL),
                dims[1].index.ptr, per_batch, None, 0, abi.GetHLLValue, None, 0)
        be.call("InitIndexVector", dims[0].index.ptr, 0, size, None, 0)
        be.call("InitIndexVector", dims[1].index.ptr, size, per_batch, None, 0)
        size, vec, reg = H.hyperloglog(be, dims[0], dims[1], meas[0], meas[1], size, per_batch, b == 1)
        idx0, idx1 = dims[0].index, dims[1].index
        dims.reverse()
        meas.reverse()
        dims[0].index, dims[1].index = idx0, idx1
        col.free()
    assert size == len(distinct)
    truth = np.bincount((np.unique(np.concatenate(seen)) >> np.uint64(32)).astype(np.int64), minlength=len(distinct))
    keys = dims[0].values.read(np.uint32, size)
    assert sorted(keys.tolist()) == list(range(len(distinct)))
    off = 0
    for g, count in zip(keys.tolist(), reg.tolist()):
        if count < 4096:
            words = vec[off:off + 4 * count].view(np.uint32)
            regs = np.zeros(1 << 14, np.uint8)
            regs[words & 0xFFFF] = (words >> 16).astype(np.uint8)
            off += 4 * count
        else:
            regs = vec[off:off + (1 << 14)]
            off += 1 << 14
        m = float(1 << 14)
        est = 0.7213 / (1 + 1.079 / m) * m * m / np.sum(2.0 ** -regs.astype(np.float64))
        zeros = int(np.count_nonzero(regs == 0))
        if est <= 2.5 * m and zeros:
            est = m * np.log(m / zeros)
        assert abs(est - truth[g]) / truth[g] < tolerance, (g, est, truth[g])
    assert off == len(vec)
    for x in dims + meas:
        x.free()


class _Fixed:
    """A DimensionVector struct prepared by the caller, with the DimVector.struct() interface."""

    def __init__(self, dv):
        self.dv = dv

    def struct(self):
        return self.dv


def assert_same(a, b, what=""):
    assert a.keys() == b.keys(), what
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray):
            assert x.shape == y.shape and np.array_equal(x, y), \
                f"{what}: field {k!r} differs\n{x[:32]}\n{y[:32]}"
        else:
            assert x == y, f"{what}: field {k!r} differs: {x!r} != {y!r}"
