"""Known-answer tests of the reference's own gtest suite, replayed through the C ABI.

Each test cites the reference test it restates (query/algorithm_unittest.cu).  The same vectors
run against the C oracle, the reference's HOST build (when oracle/_ref exists) and — under
`-m gpu` — the HIP library, so the oracle is pinned by the reference's goldens and the product
is pinned by both.
"""
import calendar
import ctypes as C
import json
import os

import numpy as np
import pytest

import harness as H
from aresdb_amd import abi


def ts(y, m, d):
    return calendar.timegm((y, m, d, 0, 0, 0))


# ---- UnaryTransformTest (algorithm_unittest.cu:79-268) ---------------------------------------
def test_unary_transform_check_int(be):
    col = H.Column(be, abi.Int32, [-1, 1, 0], valid=[1, 1, 0])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    out = H.Scratch(be, 3, abi.Int32)
    n = be.call("UnaryTransform", col.input(), out.output(), idx.ptr, 3, None, 0, abi.Negate, None, 0)
    assert n == 3
    assert out.values().tolist() == [1, -1, 0]
    assert out.valid().tolist() == [1, 1, 0]


def test_unary_transform_check_constant(be):
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    out = H.Scratch(be, 3, abi.Int32)
    be.call("UnaryTransform", H.const_int(1), out.output(), idx.ptr, 3, None, 0, abi.Negate, None, 0)
    assert out.values().tolist() == [-1, -1, -1]
    assert out.valid().tolist() == [1, 1, 1]


def test_unary_transform_measure_output_sum_and_avg(be):
    """CheckMeasureOutputIteratorForAvg (:153-214): RLE counts multiply SUM, AVG packs (avg,count)."""
    col = H.Column(be, abi.Int32, [-1, 1, 0], valid=[1, 1, 0])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    base_counts = H.Buf(be, np.array([0, 3, 9, 12], np.uint32))
    outv = H.Buf(be, nbytes=24)
    mo = H.measure_output(outv.ptr, abi.Int64, abi.AGGR_SUM_SIGNED)
    be.call("UnaryTransform", col.input(), mo, idx.ptr, 3, base_counts.ptr, 0, abi.Negate, None, 0)
    assert outv.read(np.int64, 3).tolist() == [3, -6, 0]
    be.call("UnaryTransform", col.input(), mo, idx.ptr, 3, base_counts.ptr, 0, abi.Noop, None, 0)
    assert outv.read(np.int64, 3).tolist() == [-3, 6, 0]
    mo2 = H.measure_output(outv.ptr, abi.Float64, abi.AGGR_AVG_FLOAT)
    be.call("UnaryTransform", col.input(), mo2, idx.ptr, 3, base_counts.ptr, 0, abi.Noop, None, 0)
    raw = outv.read(np.uint32, 6)
    assert raw.view(np.float32)[[0, 2, 4]].tolist() == [-1.0, 1.0, 0.0]
    assert raw[[1, 3, 5]].tolist() == [3, 6, 0]


def test_unary_transform_dimension_output(be):
    """CheckDimensionOutputIterator (:217-268): int16 dim values + validity bytes."""
    col = H.Column(be, abi.Int16, [-1, 1, 0], valid=[1, 1, 0])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    out = H.Buf(be, nbytes=9)
    do = H.dimension_output(out.ptr, out.ptr + 6, abi.Int16)
    be.call("UnaryTransform", col.input(), do, idx.ptr, 3, None, 0, abi.Negate, None, 0)
    assert out.read(np.uint8, 9).tolist() == [1, 0, 0xFF, 0xFF, 0, 0, 1, 1, 0]
    be.call("UnaryTransform", col.input(), do, idx.ptr, 3, None, 0, abi.Noop, None, 0)
    assert out.read(np.uint8, 9).tolist() == [0xFF, 0xFF, 1, 0, 0, 0, 1, 1, 0]


# ---- UnaryFilterTest (:271-353) -----------------------------------------------------------------
def test_unary_filter_check_filter(be):
    inp = H.Scratch(be, 3, abi.Int32, [1, 0, 1], [1, 0, 1])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    pred = H.Buf(be, nbytes=3)
    rids = H.Buf(be, H.record_id_array([(0, 0), (0, 1), (0, 2)]))
    vecs = (C.c_void_p * 1)(rids.ptr)
    n = be.call("UnaryFilter", inp.input(), idx.ptr, pred.ptr, 3, C.addressof(vecs), 1, None, 0,
                abi.IsNotNull, None, 0)
    assert n == 2
    assert idx.read(np.uint32, 2).tolist() == [0, 2]
    got = rids.read(np.uint32, 4).tolist()
    assert got == [0, 0, 0, 2]


def test_unary_filter_all_empty(be):
    inp = H.Scratch(be, 3, abi.Int32, [0, 0, 0], [1, 1, 1])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    pred = H.Buf(be, nbytes=3)
    n = be.call("UnaryFilter", inp.input(), idx.ptr, pred.ptr, 3, None, 0, None, 0, abi.Negate, None, 0)
    assert n == 0


# ---- BinaryTransformTest / BinaryFilterTest (:356-716) -----------------------------------------
def test_binary_transform_check_int(be):
    lhs = H.Column(be, abi.Int32, [-1, 1, 0], valid=[1, 1, 0])
    rhs = H.Column(be, abi.Int32, [0, 1, -1], valid=[0, 1, 1])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    out = H.Scratch(be, 3, abi.Int32)
    be.call("BinaryTransform", lhs.input(), rhs.input(), out.output(), idx.ptr, 3, None, 0,
            abi.Plus, None, 0)
    assert out.values().tolist() == [0, 2, 0]
    assert out.valid().tolist() == [0, 1, 0]


def test_binary_transform_float_and_scratch(be):
    lhs = H.Scratch(be, 3, abi.Int32, [-1, 1, 0], [1, 1, 1])
    rhs = H.Scratch(be, 3, abi.Float32, [1.1, -1.1, 0.1], [1, 1, 1])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    out = H.Scratch(be, 3, abi.Float32)
    be.call("BinaryTransform", lhs.input(), rhs.input(), out.output(), idx.ptr, 3, None, 0,
            abi.Minus, None, 0)
    exp = np.array([-1, 1, 0], np.float32) - np.array([1.1, -1.1, 0.1], np.float32)
    assert np.array_equal(out.values(), exp)
    assert out.valid().tolist() == [1, 1, 1]
    be.call("BinaryTransform", lhs.input(), rhs.input(), out.output(), idx.ptr, 3, None, 0,
            abi.Multiply, None, 0)
    exp = np.array([-1, 1, 0], np.float32) * np.array([1.1, -1.1, 0.1], np.float32)
    assert np.array_equal(out.values(), exp)


def test_binary_transform_constant(be):
    lhs = H.Scratch(be, 3, abi.Int32, [-1, 1, 0], [1, 1, 1])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    out = H.Scratch(be, 3, abi.Float32)
    be.call("BinaryTransform", lhs.input(), H.const_float(0.1), out.output(), idx.ptr, 3, None, 0,
            abi.Plus, None, 0)
    exp = np.array([-1, 1, 0], np.float32) + np.float32(0.1)
    assert np.array_equal(out.values(), exp)
    assert out.valid().tolist() == [1, 1, 1]


def test_binary_filter_check_filter(be):
    lhs = H.Scratch(be, 3, abi.Int32, [0, 1, 2], [1, 1, 1])
    rhs = H.Scratch(be, 3, abi.Float32, [0.1, 0.9, 1.9], [1, 1, 1])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    pred = H.Buf(be, nbytes=3)
    n = be.call("BinaryFilter", lhs.input(), rhs.input(), idx.ptr, pred.ptr, 3, None, 0, None, 0,
                abi.GreaterThan, None, 0)
    assert n == 2
    assert idx.read(np.uint32, 2).tolist() == [1, 2]


def test_binary_transform_measure_output(be):
    """CheckMeasureOutputIterator (:664-716): (lhs - rhs) * runLength, null -> identity 0."""
    lhs = H.Scratch(be, 3, abi.Int32, [-1, 1, 0], [1, 1, 0])
    rhs = H.Scratch(be, 3, abi.Float32, [1.1, -1.1, 0.1], [1, 1, 1])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    base_counts = H.Buf(be, np.array([0, 3, 9, 10], np.uint32))
    outv = H.Buf(be, nbytes=12)
    mo = H.measure_output(outv.ptr, abi.Float32, abi.AGGR_SUM_FLOAT)
    be.call("BinaryTransform", lhs.input(), rhs.input(), mo, idx.ptr, 3, base_counts.ptr, 0,
            abi.Minus, None, 0)
    exp = (np.array([-1, 1], np.float32) - np.array([1.1, -1.1], np.float32)) * np.array([3, 6], np.float32)
    got = outv.read(np.float32, 3)
    assert np.array_equal(got[:2], exp) and got[2] == 0.0


def test_binary_geo_point_equal_and_error(be):
    """CheckGeoPoint (:417-486)."""
    geo = np.array([[1, 1], [1, 0], [0, 0]], np.float32)
    col = H.Column(be, abi.GeoPoint, raw_values=geo.tobytes(), valid=[1, 1, 0])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    out = H.Scratch(be, 3, abi.Uint32)
    rhs = abi.InputVector()
    rhs.Vector.Constant.Value.GeoPointVal.Lat = 1.0
    rhs.Vector.Constant.Value.GeoPointVal.Long = 1.0
    rhs.Vector.Constant.IsValid = True
    rhs.Vector.Constant.DataType = abi.ConstGeoPoint
    rhs.Type = abi.ConstantInput
    be.call("BinaryTransform", col.input(), rhs, out.output(), idx.ptr, 3, None, 0, abi.Equal, None, 0)
    assert out.values().tolist() == [1, 0, 0]
    assert out.valid().tolist() == [1, 1, 0]
    other = H.Column(be, abi.Int32, [0, 1, -1], valid=[0, 1, 1])
    with pytest.raises(abi.AresError):
        be.call("BinaryTransform", col.input(), other.input(), out.output(), idx.ptr, 3, None, 0,
                abi.Equal, None, 0)


def test_init_index_vector(be):
    idx = H.Buf(be, nbytes=12)
    be.call("InitIndexVector", idx.ptr, 0, 3, None, 0)
    assert idx.read(np.uint32, 3).tolist() == [0, 1, 2]
    be.call("InitIndexVector", idx.ptr, 7, 3, None, 0)
    assert idx.read(np.uint32, 3).tolist() == [7, 8, 9]


# ---- HashLookupTest (:731-883): cuckoo table bytes generated by the Go index ------------------
GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                     "cuckoo_tables.json")))
BUCKETS_4B = GOLDEN["lookup4"]["buckets"]
SEEDS_4B = GOLDEN["lookup4"]["seeds"]


def _cuckoo(be, buckets, seeds, key_bytes, num_hashes, num_buckets):
    buf = H.Buf(be, np.array(buckets, np.uint8))
    hi = abi.CuckooHashIndex()
    hi.buckets = buf.ptr
    for i, s in enumerate(seeds):
        hi.seeds[i] = s
    hi.keyBytes, hi.numHashes, hi.numBuckets = key_bytes, num_hashes, num_buckets
    return hi, buf


def test_hash_lookup_check_lookup(be):
    assert len(BUCKETS_4B) == 312
    hi, _keep = _cuckoo(be, BUCKETS_4B, SEEDS_4B, 4, 4, 2)
    n = 18
    col = H.Column(be, abi.Int32, list(range(n)), valid=[1] * n)
    idx = H.Buf(be, np.arange(n, dtype=np.uint32))
    out = H.Buf(be, nbytes=8 * n)
    be.call("HashLookup", col.input(), out.ptr, idx.ptr, n, None, 0, hi, None, 0)
    got = out.read(np.uint32, 2 * n).reshape(n, 2)
    assert got[:, 0].tolist() == [0] * n
    assert got[:, 1].tolist() == list(range(n))


def test_hash_lookup_miss_and_null(be):
    hi, _keep = _cuckoo(be, BUCKETS_4B, SEEDS_4B, 4, 4, 2)
    col = H.Column(be, abi.Int32, [5, 1000, 7, 12345], valid=[1, 1, 0, 1])
    idx = H.Buf(be, np.arange(4, dtype=np.uint32))
    out = H.Buf(be, np.full(8, 0xFFFFFFFF, np.uint32))
    be.call("HashLookup", col.input(), out.ptr, idx.ptr, 4, None, 0, hi, None, 0)
    got = out.read(np.uint32, 8).reshape(4, 2)
    assert got.tolist() == [[0, 5], [0, 0], [0, 0], [0, 0]]


def test_hash_lookup_uuid(be):
    """CheckUUID (:803-883): 16-byte keys."""
    g = GOLDEN["uuid16"]
    hi, _keep = _cuckoo(be, g["buckets"], g["seeds"], 16, 4, 2)
    vals = np.array(g["keys"], np.uint64)
    col = H.Column(be, abi.UUID, raw_values=vals.tobytes(), valid=[1, 1, 1])
    idx = H.Buf(be, np.arange(3, dtype=np.uint32))
    out = H.Buf(be, nbytes=24)
    be.call("HashLookup", col.input(), out.ptr, idx.ptr, 3, None, 0, hi, None, 0)
    got = out.read(np.uint32, 6).reshape(3, 2)
    assert got.tolist() == [[0, 0], [0, 1], [0, 2]]


# ---- ForeignTableColumnTransformTest (:885-976) -------------------------------------------------
def test_foreign_table_column_transform(be):
    base_batch = -2147483648
    n = 5
    cols = [H.Column(be, abi.Int32, [(i + 1) if j == i else 0 for j in range(5)], valid=[1] * 5)
            for i in range(5)]
    batches = (abi.VectorPartySlice * n)(*[c.vp for c in cols])
    rids = H.Buf(be, H.record_id_array([(i + base_batch, i) for i in range(n)]))
    iv = abi.InputVector()
    f = iv.Vector.ForeignVP
    f.RecordIDs = rids.ptr
    f.Batches = C.addressof(batches)
    f.BaseBatchID, f.NumBatches, f.NumRecordsInLastBatch = base_batch, n, 5
    f.TimezoneLookup, f.TimezoneLookupSize = None, 0
    f.DataType = abi.Int32
    iv.Type = abi.ForeignColumnInput
    out = H.Scratch(be, n, abi.Int32)
    be.call("UnaryTransform", iv, out.output(), None, n, None, 0, abi.Negate, None, 0)
    assert out.values().tolist() == [-1, -2, -3, -4, -5]
    assert out.valid().tolist() == [1] * 5


# ---- Sort / Reduce (:979-1226) --------------------------------------------------------------------
def test_sort_dim_column_vector(be):
    keys = np.zeros(30, np.uint8)
    keys[0:12].view(np.uint32)[:] = [1, 2, 1]
    keys[12:18].view(np.uint16)[:] = [1, 2, 1]
    keys[18:21] = [1, 2, 1]
    dv = H.DimVector(be, 3, (0, 0, 1, 1, 1), init=keys)
    dv.index.write(np.array([0, 1, 2], np.uint32))
    be.call("Sort", dv.struct(), 3, None, 0)
    got = dv.index.read(np.uint32, 3).tolist()
    assert got in ([1, 0, 2], [0, 2, 1])
    hv = dv.hash.read(np.uint64, 3)
    assert hv[0] <= hv[1] <= hv[2]


REDUCE_DIMS = [1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 1, 0, 0, 0, 1, 0, 2, 0,
               3, 0, 2, 0, 3, 0, 1, 0, 1, 2, 3, 2, 3, 1] + [1] * 18


def test_reduce_dim_column_vector(be):
    din = H.DimVector(be, 6, (0, 0, 1, 1, 1), init=REDUCE_DIMS)
    din.hash.write(np.array([1, 1, 2, 2, 3, 3], np.uint64))
    din.index.write(np.array([1, 3, 2, 4, 0, 5], np.uint32))
    vin = H.Buf(be, np.array([5, 1, 3, 2, 4, 6], np.uint32))
    dout = H.DimVector(be, 6, (0, 0, 1, 1, 1))
    vout = H.Buf(be, nbytes=24)
    n = be.call("Reduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, 4, 6,
                abi.AGGR_SUM_UNSIGNED, None, 0)
    assert n == 3
    assert vout.read(np.uint32, 3).tolist() == [3, 7, 11]
    assert dout.index.read(np.uint32, 3).tolist() == [1, 2, 0]
    exp = [2, 0, 0, 0, 3, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
           0, 0, 0, 0, 2, 0, 3, 0, 1, 0, 0, 0, 0, 0, 0, 0, 2, 3, 1, 0,
           0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0]
    assert dout.values.read(np.uint8, 60).tolist() == exp


def test_reduce_by_avg(be):
    dims = [1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 1, 0, 0, 0] + [1] * 6
    din = H.DimVector(be, 6, (0, 0, 1, 0, 0), init=dims)
    din.hash.write(np.array([1, 1, 2, 2, 3, 3], np.uint64))
    din.index.write(np.array([1, 3, 2, 4, 0, 5], np.uint32))
    vals = np.zeros(12, np.uint32)
    vals[0::2] = np.array([5, 1, 3, 2, 4, 6], np.float32).view(np.uint32)
    vals[1::2] = 1
    vin = H.Buf(be, vals)
    dout = H.DimVector(be, 6, (0, 0, 1, 0, 0))
    vout = H.Buf(be, nbytes=48)
    n = be.call("Reduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, 8, 6,
                abi.AGGR_AVG_FLOAT, None, 0)
    assert n == 3
    raw = vout.read(np.uint32, 6)
    assert raw[0::2].view(np.float32).tolist() == [1.5, 3.5, 5.5]
    assert raw[1::2].tolist() == [2, 2, 2]
    assert dout.index.read(np.uint32, 3).tolist() == [1, 2, 0]
    exp = [2, 0, 0, 0, 3, 0, 0, 0, 1, 0, 0, 0] + [0] * 12 + [1, 1, 1, 0, 0, 0]
    assert dout.values.read(np.uint8, 30).tolist() == exp


def test_sort_and_reduce_check_hash(be):
    """SortAndReduceTest.CheckHash (:1160-1217): pins murmur3_x64_128 ordering."""
    dims = [2, 1, 0, 3, 0, 1, 2, 3] + [1] * 8
    din = H.DimVector(be, 8, (0, 0, 0, 0, 1), init=dims)
    be.call("InitIndexVector", din.index.ptr, 0, 8, None, 0)
    vin = H.Buf(be, np.ones(8, np.uint32))
    dout = H.DimVector(be, 8, (0, 0, 0, 0, 1))
    vout = H.Buf(be, nbytes=32)
    be.call("Sort", din.struct(), 8, None, 0)
    hv = din.hash.read(np.uint64, 8)
    # known-answer hashes recorded from the reference build
    assert sorted(set(hv.tolist())) == [0x60e187b4814392c4, 0x7cb3f5c58dab264c,
                                        0xb73e42bb654cee53, 0xca410abc0a9d4c6b]
    n = be.call("Reduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, 4, 8,
                abi.AGGR_SUM_UNSIGNED, None, 0)
    assert n == 4
    assert dout.values.read(np.uint8, 16).tolist() == [2, 0, 3, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0]
    assert vout.read(np.uint32, 8).tolist() == [2, 2, 2, 2, 0, 0, 0, 0]
    assert dout.index.read(np.uint32, 4).tolist() == [0, 2, 3, 1]


# ---- HashReductionTest.CheckReduce (:1957-2056) ------------------------------------------------
def test_hash_reduction_check_reduce(be):
    din = H.DimVector(be, 6, (0, 0, 1, 1, 1), with_hash=False, with_index=False, init=REDUCE_DIMS)
    vin = H.Buf(be, np.array([5, 1, 3, 2, 4, 6], np.uint32))
    dout = H.DimVector(be, 6, (0, 0, 1, 1, 1), with_hash=False, with_index=False)
    vout = H.Buf(be, nbytes=24)
    n = be.call("HashReduce", din.struct(), vin.ptr, dout.struct(), vout.ptr, 4, 6,
                abi.AGGR_SUM_UNSIGNED, None, 0)
    assert n == 3
    rows = dout.rows(3)
    vals = vout.read(np.uint32, 3).tolist()
    got = {r: v for r, v in zip(rows, vals)}

    def key(x):
        return ((np.uint32(x).tobytes(), np.uint16(x).tobytes(), np.uint8(x).tobytes()), (1, 1, 1))
    assert got == {key(2): 3, key(1): 11, key(3): 7}


# ---- DateFunctorsTest (:1420-1500) -----------------------------------------------------------------
def test_date_functors(be):
    be.call("BootstrapDevice")
    inp = H.Scratch(be, 3, abi.Int32, np.array([0, ts(2018, 6, 11), ts(1970, 1, 1)], np.uint32).view(np.int32),
                    [0, 1, 1])
    idx = H.Buf(be, np.array([0, 1, 2], np.uint32))
    out = H.Scratch(be, 3, abi.Int32)
    for functor, exp in ((abi.GetMonthStart, ts(2018, 6, 1)), (abi.GetQuarterStart, ts(2018, 4, 1)),
                         (abi.GetYearStart, ts(2018, 1, 1))):
        be.call("UnaryTransform", inp.input(), out.output(), idx.ptr, 3, None, 0, functor, None, 0)
        assert out.values().view(np.uint32).tolist() == [0, exp, 0]
        assert out.valid().tolist() == [0, 1, 1]


# ---- ExpandTest (:1747-1950) -----------------------------------------------------------------------
EXPAND_DIMS = ([1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 1, 0, 0, 0,
                1, 0, 2, 0, 3, 0, 2, 0, 3, 0, 1, 0, 1, 2, 3, 2, 3, 1] + [1] * 18)
EXPAND_BASE_COUNTS = [0, 0, 1, 3, 4, 7, 8, 10, 13, 14, 16, 19, 22, 23, 26, 30]


def _expand(be, index, length, occupied):
    din = H.DimVector(be, 6, (0, 0, 1, 1, 1), with_hash=False, with_index=False, init=EXPAND_DIMS)
    dout = H.DimVector(be, 10, (0, 0, 1, 1, 1), with_hash=False, with_index=False)
    idx = H.Buf(be, np.array(index, np.uint32))
    bc = H.Buf(be, np.array(EXPAND_BASE_COUNTS, np.uint32))
    n = be.call("Expand", din.struct(), dout.struct(), bc.ptr, idx.ptr, length, occupied, None, 0)
    return n, dout.values.read(np.uint8, 100).tolist()


def test_expand_overfill(be):
    n, got = _expand(be, [1, 2, 4, 7, 9, 12], 6, 0)
    exp = [1, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 3, 0, 0, 0,
           3, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0,
           1, 0, 2, 0, 2, 0, 3, 0, 3, 0, 3, 0, 2, 0, 2, 0, 2, 0, 3, 0,
           1, 2, 2, 3, 3, 3, 2, 2, 2, 3] + [1] * 30
    assert n == 10 and got == exp


def test_expand_append(be):
    n, got = _expand(be, [1, 2, 4, 7, 9, 12], 6, 5)
    exp = [0] * 20 + [1, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 3, 0, 0, 0] + \
          [0] * 10 + [1, 0, 2, 0, 2, 0, 3, 0, 3, 0] + [0] * 5 + [1, 2, 2, 3, 3] + \
          ([0] * 5 + [1] * 5) * 3
    assert n == 10 and got == exp


def test_expand_fill_partial(be):
    n, got = _expand(be, [1, 2, 4], 3, 0)
    exp = [1, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 3, 0, 0, 0,
           3, 0, 0, 0] + [0] * 16 + [1, 0, 2, 0, 2, 0, 3, 0, 3, 0, 3, 0] + [0] * 8 + \
          [1, 2, 2, 3, 3, 3, 0, 0, 0, 0] + ([1] * 6 + [0] * 4) * 3
    assert n == 6 and got == exp


# ---- HyperLogLogTest (:1229-1420) -------------------------------------------------------------
def test_hyperloglog_sparse_mode(be):
    """HyperLogLogTest.CheckSparseMode (:1229-1302)."""
    prev = H.DimVector(be, 8, (0, 0, 0, 0, 1), init=[1, 1, 2, 2, 3, 3, 4, 4] + [1] * 8)
    cur = H.DimVector(be, 8, (0, 0, 0, 0, 1))
    cur.index.write(np.arange(8, dtype=np.uint32))
    pv = H.Buf(be, nbytes=32)
    cv = H.Buf(be, np.array([0x010001, 0x020002, 0x010002, 0x020002, 0x010003, 0x020003, 0x010004, 0x020004],
                            np.uint32))
    n, hll, reg = H.hyperloglog(be, prev, cur, pv, cv, 0, 8, True)
    assert n == 4
    assert hll.tolist() == [2, 0, 3, 0, 4, 0, 3, 0, 3, 0, 3, 0, 1, 0, 2, 0, 2, 0, 3, 0]
    assert reg.tolist() == [1, 1, 1, 2]
    assert cur.values.read(np.uint8, 16).tolist() == [2, 4, 3, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0]


def test_hyperloglog_dense_mode(be):
    """HyperLogLogTest.CheckDenseMode (:1305-1420)."""
    dims = np.zeros(10000, np.uint8)
    dims[0:4] = [1, 1, 2, 2]
    dims[5000:] = 1
    prev = H.DimVector(be, 5000, (0, 0, 0, 0, 1), init=dims)
    cur = H.DimVector(be, 5000, (0, 0, 0, 0, 1))
    cur.index.write(np.arange(5000, dtype=np.uint32))
    vals = np.zeros(5000, np.uint32)
    vals[0:4] = [0x010001, 0x020002, 0x010002, 0x020002]
    vals[4:] = 0x010000 | np.arange(4996, dtype=np.uint32)
    pv = H.Buf(be, nbytes=20000)
    cv = H.Buf(be, vals)
    n, hll, reg = H.hyperloglog(be, prev, cur, pv, cv, 0, 5000, True)
    assert n == 3
    assert len(hll) == 16396
    exp = np.zeros(16396, np.uint8)
    exp[0:4] = [2, 0, 3, 0]
    exp[4:5000] = 2
    exp[16388:] = [1, 0, 2, 0, 2, 0, 3, 0]
    assert np.array_equal(hll, exp)
    assert reg.tolist() == [1, 4996, 2]
    exp_dims = np.zeros(10000, np.uint8)
    exp_dims[0:3] = [2, 0, 1]
    exp_dims[5000:5003] = 1
    assert np.array_equal(cur.values.read(np.uint8, 10000), exp_dims)


# ---- GeoBatchIntersectTest / GeoBatchIntersectionJoinTest (:1502-1745) -------------------------
FLT_MAX = np.finfo(np.float32).max
# 1. square (1,1),(1,-1),(-1,-1),(-1,1)  2. triangle (3,3),(2,2),(4,2)
# 3. square (0,6),(3,6),(3,3),(0,3) with the hole (1,5),(2,5),(2,4),(1,4)
GEO_LATS = [1, 1, -1, -1, 1, 3, 2, 4, 3, 0, 3, 3, 0, 0, FLT_MAX, 1, 2, 2, 1, 1]
GEO_LONGS = [1, -1, -1, 1, 1, 3, 2, 2, 3, 6, 6, 3, 3, 6, FLT_MAX, 5, 5, 4, 4, 5]
GEO_SHAPE = [0] * 5 + [1] * 4 + [2] * 11
GEO_POINTS = [[0, 0], [3, 2.5], [1.5, 3.5], [1.5, 4.5], [0, 0]]  # in 1, in 2, in 3, in the hole, null


@pytest.mark.parametrize("in_or_out,kept,pred", [(True, 3, [1, 2, 4, 0, 0]), (False, 1, [1, 2, 4, 0, 1])])
def test_geo_batch_intersects(be, in_or_out, kept, pred):
    """CheckInShape (:1502-1552) and CheckNotInShape (:1636-1685)."""
    shapes = H.GeoShapes(be, GEO_LATS, GEO_LONGS, GEO_SHAPE, 3)
    col = H.geo_column(be, GEO_POINTS, valid=[1, 1, 1, 1, 0])
    idx = H.Buf(be, np.arange(5, dtype=np.uint32))
    out = H.Buf(be, nbytes=20)
    n = be.call("GeoBatchIntersects", shapes.struct(), col.input(), idx.ptr, 5, 0, None, 0, out.ptr, in_or_out, None, 0)
    assert n == kept
    assert out.read(np.uint32, 5).tolist() == pred


def test_geo_batch_intersects_record_id_join(be):
    """CheckRecordIDJoinIterator (:1554-1634): the points come from a joined table, one per batch."""
    shapes = H.GeoShapes(be, GEO_LATS, GEO_LONGS, GEO_SHAPE, 3)
    base_batch = -2147483648
    rids = H.Buf(be, H.record_id_array([(base_batch + i, i) for i in range(5)]))
    cols, slices = [], (abi.VectorPartySlice * 5)()
    for i in range(5):
        pts = np.zeros((5, 2), np.float32)
        pts[i] = GEO_POINTS[i]
        c = H.geo_column(be, pts, valid=[1, 1, 1, 1, 0])
        cols.append(c)
        slices[i] = c.vp
    iv = abi.InputVector()
    f = iv.Vector.ForeignVP
    f.RecordIDs, f.Batches = rids.ptr, C.addressof(slices)
    f.BaseBatchID, f.NumBatches, f.NumRecordsInLastBatch = base_batch, 5, 5
    f.TimezoneLookup, f.TimezoneLookupSize, f.DataType = None, 0, abi.GeoPoint
    iv.Type = abi.ForeignColumnInput
    idx = H.Buf(be, np.arange(5, dtype=np.uint32))
    out = H.Buf(be, nbytes=20)
    n = be.call("GeoBatchIntersects", shapes.struct(), iv, idx.ptr, 5, 0, None, 0, out.ptr, True, None, 0)
    assert n == 3
    assert out.read(np.uint32, 5).tolist() == [1, 2, 4, 0, 0]


def test_geo_shape_dimension_writing(be):
    """GeoBatchIntersectionJoinTest.DimensionWriting (:1687-1745)."""
    lats = [1, 1, -1, -1, 1, 2, 2, -2, -2, 2, 1.6, 1.6, 1.4, 1.4, 1.6]
    longs = [1, -1, -1, 1, 1, 2, -2, -2, 2, 2, 3.6, 3.4, 3.6, 3.4, 3.6]
    shapes = H.GeoShapes(be, np.float32(lats), np.float32(longs), [0] * 5 + [1] * 5 + [2] * 5, 3)
    col = H.geo_column(be, [[1.5, 1.5], [0, 0], [1.5, 4.5], [1.5, 3.5], [0, 0]], valid=[1, 1, 1, 1, 0])
    idx = H.Buf(be, np.arange(5, dtype=np.uint32))
    out = H.Buf(be, nbytes=20)
    dim = H.Buf(be, nbytes=16)
    n = be.call("GeoBatchIntersects", shapes.struct(), col.input(), idx.ptr, 5, 0, None, 0, out.ptr, True, None, 0)
    assert n == 3
    dv = abi.DimensionOutputVector()
    dv.DimValues, dv.DimNulls, dv.DataType = dim.ptr, dim.ptr + 5, abi.Uint8
    be.call("WriteGeoShapeDim", 1, dv, 5, out.ptr, None, 0)
    be.wait()
    assert out.read(np.uint32, 5).tolist() == [2, 3, 0, 4, 0]
    assert dim.read(np.uint8, 5).tolist() == [1, 0, 2, 0, 0]
    assert dim.read(np.uint8, 5, 5).tolist() == [1, 1, 1, 0, 0]
