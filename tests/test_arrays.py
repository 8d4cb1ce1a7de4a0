"""Array columns (ArrayVectorPartyInput; ArrayLength / ArrayContains / ArrayElementAt): the reference's
three known-answer tests (query/functor_unittest.cu:1171-1300, ArrayLengthTest / ArrayContainsTest /
ArrayElementAtTest) replayed through the ABI on every backend, and seeded random array columns of all
eleven element types (null arrays, empty arrays, null elements) on which every backend must agree with
the oracle — the reference build included, which pins the restatement."""
import ctypes as C

import numpy as np
import pytest

import harness as H
from aresdb_amd import abi

OFFSET_LENGTH = [0, 2, 16, 1, 32, 3, 0, 0, 0xFFFFFFFF, 0, 56, 1]
VALUES = [2, 1, 2, 0x03, 1, 1, 0x01, 0, 3, 1, 2, 3, 0x07, 0, 1, 1, 0x01, 0]


def _array_input(buf, dtype, length, adj=0):
    iv = abi.InputVector()
    iv.Vector.ArrayVP.OffsetLengthVector = buf.ptr
    iv.Vector.ArrayVP.ValueOffsetAdj = adj
    iv.Vector.ArrayVP.DataType = dtype
    iv.Vector.ArrayVP.Length = length
    iv.Type = abi.ArrayVectorPartyInput
    return iv


def _golden_column(be):
    vals = np.zeros(72, np.uint32)
    vals[:len(VALUES)] = VALUES
    blob = np.concatenate([np.array(OFFSET_LENGTH, np.uint32), vals]).view(np.uint8)
    return H.Buf(be, blob)


def test_array_length_known_answer(be):
    col = _golden_column(be)
    out = H.Scratch(be, 6, abi.Uint32)
    be.call("UnaryTransform", _array_input(col, abi.Uint32, 6), out.output(), None, 6, None, 0, abi.ArrayLength, None, 0)
    be.wait()
    assert out.values().tolist() == [2, 1, 3, 0, 0, 1]
    assert out.valid().tolist() == [1, 1, 1, 0, 1, 1]
    col.free(); out.free()


def test_array_contains_known_answer(be):
    col = _golden_column(be)
    dim = H.Buf(be, nbytes=16)
    nulls = H.Buf(be, nbytes=16)
    be.call("BinaryTransform", _array_input(col, abi.Uint32, 6), H.const_int(2), H.dimension_output(dim.ptr, nulls.ptr, abi.Bool),
            None, 6, None, 0, abi.ArrayContains, None, 0)
    be.wait()
    assert dim.read(np.uint8, 6).tolist() == [1, 0, 1, 0, 0, 0]
    assert nulls.read(np.uint8, 6).tolist() == [1, 1, 1, 0, 1, 1]
    # the same as a filter: rows 0 and 2 survive
    idx, pred = H.Buf(be, np.arange(6, dtype=np.uint32)), H.Buf(be, nbytes=8)
    n = be.call("BinaryFilter", _array_input(col, abi.Uint32, 6), H.const_int(2), idx.ptr, pred.ptr, 6, None, 0, None, 0,
                abi.ArrayContains, None, 0)
    be.wait()
    assert n == 2 and idx.read(np.uint32, 2).tolist() == [0, 2]
    for b in (col, dim, nulls, idx, pred):
        b.free()


def test_array_element_at_known_answer(be):
    col = _golden_column(be)
    out = H.Scratch(be, 6, abi.Uint32)
    be.call("BinaryTransform", _array_input(col, abi.Uint32, 6), H.const_int(1), out.output(), None, 6, None, 0,
            abi.ArrayElementAt, None, 0)
    be.wait()
    assert out.values().tolist() == [2, 0, 2, 0, 0, 0]
    assert out.valid().tolist() == [1, 0, 1, 0, 0, 0]
    col.free(); out.free()


ELEM = {abi.Bool: (1, np.uint8), abi.Int8: (1, np.int8), abi.Uint8: (1, np.uint8), abi.Int16: (2, np.int16),
        abi.Uint16: (2, np.uint16), abi.Int32: (4, np.int32), abi.Uint32: (4, np.uint32), abi.Float32: (4, np.float32),
        abi.Int64: (8, np.int64), abi.GeoPoint: (8, np.float32), abi.UUID: (16, np.uint64)}


def _random_column(rng, dtype, n):
    w, npt = ELEM[dtype]
    ol = np.zeros((n, 2), np.uint32)
    chunks, off = [], 0
    for i in range(n):
        r = rng.random()
        if r < 0.1:
            continue                                  # null array: offset 0, length 0
        if r < 0.2:
            ol[i] = (0xFFFFFFFF, 0)                   # empty array: any non-zero offset, length 0
            continue
        ln = int(rng.integers(1, 10))
        if dtype == abi.Bool:
            elems = rng.integers(0, 2, ln).astype(np.uint8)
        elif dtype == abi.Float32:
            elems = (rng.integers(-8, 8, ln) / 2).astype(np.float32)
        elif dtype == abi.GeoPoint:
            elems = (rng.integers(0, 3, 2 * ln)).astype(np.float32)
        elif dtype == abi.UUID:
            elems = rng.integers(0, 2, 2 * ln).astype(np.uint64)
        else:
            elems = rng.integers(-4 if npt().dtype.kind == "i" else 0, 5, ln).astype(npt)
        valid = np.packbits(rng.random(ln) > 0.2, bitorder="little")
        body = np.concatenate([np.array([ln], np.uint32).view(np.uint8), elems.view(np.uint8), valid])
        body = np.concatenate([body, np.zeros(-len(body) % 8, np.uint8)])
        ol[i] = (off, ln)
        chunks.append(body)
        off += len(body)
    values = np.concatenate(chunks) if chunks else np.zeros(8, np.uint8)
    return np.concatenate([ol.reshape(-1).view(np.uint8), values, np.zeros(16, np.uint8)])


def _run_array_ops(be, seed):
    rng = np.random.default_rng(seed)
    dtype = list(ELEM)[seed % len(ELEM)]
    n = int(rng.integers(1, 3000))
    col = H.Buf(be, _random_column(rng, dtype, n))
    inp = _array_input(col, dtype, n)
    out = {}
    s = H.Scratch(be, n, abi.Uint32)
    be.call("UnaryTransform", inp, s.output(), None, n, None, 0, abi.ArrayLength, None, 0)
    be.wait()
    out["length"] = (s.values().tobytes(), s.valid().tobytes())
    s.free()
    if dtype == abi.UUID:
        const = abi.InputVector()
        const.Vector.Constant.Value.UUIDVal.p1, const.Vector.Constant.Value.UUIDVal.p2 = 1, int(rng.integers(0, 2))
        const.Vector.Constant.IsValid, const.Vector.Constant.DataType, const.Type = True, abi.ConstUUID, abi.ConstantInput
    elif dtype == abi.GeoPoint:
        const = abi.InputVector()
        const.Vector.Constant.Value.GeoPointVal.Lat, const.Vector.Constant.Value.GeoPointVal.Long = 1.0, float(rng.integers(0, 3))
        const.Vector.Constant.IsValid, const.Vector.Constant.DataType, const.Type = True, abi.ConstGeoPoint, abi.ConstantInput
    elif dtype == abi.Float32 or rng.random() < 0.3:
        const = H.const_float(float(rng.integers(-2, 3)))
    else:
        const = H.const_int(int(rng.integers(-2, 4)))
    dim, nulls = H.Buf(be, nbytes=n + 8), H.Buf(be, nbytes=n + 8)
    be.call("BinaryTransform", inp, const, H.dimension_output(dim.ptr, nulls.ptr, abi.Bool), None, n, None, 0, abi.ArrayContains,
            None, 0)
    be.wait()
    out["contains"] = (dim.read(np.uint8, n).tobytes(), nulls.read(np.uint8, n).tobytes())
    idx, pred = H.Buf(be, np.arange(n, dtype=np.uint32)), H.Buf(be, nbytes=n + 8)
    kept = be.call("BinaryFilter", inp, const, idx.ptr, pred.ptr, n, None, 0, None, 0, abi.ArrayContains, None, 0)
    be.wait()
    out["filter"] = (kept, idx.read(np.uint32, kept).tobytes())
    for b in (dim, nulls, idx, pred):
        b.free()
    index = H.const_int(int(rng.integers(-4, 5)))
    w = ELEM[dtype][0]
    if dtype in (abi.UUID, abi.GeoPoint, abi.Int64):
        s = H.Scratch(be, n, dtype, width=w)
        be.call("BinaryTransform", inp, index, s.output(), None, n, None, 0, abi.ArrayElementAt, None, 0)
        be.wait()
        out["element_wide"] = (s.buf.read(np.uint8, w * n).tobytes(), s.valid().tobytes())
        s.free()
    else:
        # static_cast<O>(element); a negative float into uint32 is undefined in C++ (x86 and the GPU differ)
        for ot in ((abi.Int32, abi.Float32) if dtype == abi.Float32 else (abi.Uint32, abi.Int32, abi.Float32)):
            s = H.Scratch(be, n, ot)
            be.call("BinaryTransform", inp, index, s.output(), None, n, None, 0, abi.ArrayElementAt, None, 0)
            be.wait()
            out[f"element_{ot}"] = (s.buf.read(np.uint8, 4 * n).tobytes(), s.valid().tobytes())
            s.free()
        dim, nulls = H.Buf(be, nbytes=w * n + 8), H.Buf(be, nbytes=n + 8)   # a dimension of the element's own type
        be.call("BinaryTransform", inp, index, H.dimension_output(dim.ptr, nulls.ptr, dtype), None, n, None, 0, abi.ArrayElementAt,
                None, 0)
        be.wait()
        out["element_dim"] = (dim.read(np.uint8, w * n).tobytes(), nulls.read(np.uint8, n).tobytes())
        m = H.Buf(be, nbytes=8 * n + 8)                                         # ... and a float64 sum measure
        be.call("BinaryTransform", inp, index, H.measure_output(m.ptr, abi.Float64, abi.AGGR_SUM_FLOAT), None, n, None, 0,
                abi.ArrayElementAt, None, 0)
        be.wait()
        out["element_measure"] = m.read(np.uint8, 8 * n).tobytes()
        for b in (dim, nulls, m):
            b.free()
    col.free()
    return out


@pytest.mark.parametrize("chunk", range(4))
def test_random_array_columns_match_the_oracle(be, chunk):
    oracle = H.oracle_backend()
    for seed in range(chunk, 66, 4):
        got, want = _run_array_ops(be, seed), _run_array_ops(oracle, seed)
        assert got.keys() == want.keys()
        for k in want:
            assert got[k] == want[k], (seed, k)


def test_array_operand_rules(be):
    """binder.hpp:469-558: the second operand of an array functor is a constant of a fitting type."""
    col = _golden_column(be)
    out = H.Scratch(be, 6, abi.Uint32)
    other = H.Column(be, abi.Uint32, np.arange(6, dtype=np.uint32))
    with pytest.raises(abi.AresError):
        be.call("BinaryTransform", _array_input(col, abi.Uint32, 6), other.input(), out.output(), None, 6, None, 0,
                abi.ArrayElementAt, None, 0)
    for b in (col, out, other):
        b.free()
