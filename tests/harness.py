"""Shared test plumbing: backends behind the C ABI and buffer helpers.

Three implementations of the same ABI can be driven with identical inputs:
  * "hip"    — the product (aresdb_amd/lib/libalgorithm.so + libmem.so), device pointers;
  * "oracle" — oracle/_build/liboracle.so, the plain-C restatement (host pointers);
  * "ref"    — oracle/_ref/libalgorithm.so + libmem.so, the reference's own sources compiled in
               QUERY_MODE=HOST by oracle/Makefile.ref (host pointers; present only when it was
               built in the authoring container — the prebuilt .so travels to the GPU box).
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

from aresdb_amd import abi

ROOT = abi.REPO_ROOT
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF_ALGO = os.path.join(ROOT, "oracle", "_ref", "libalgorithm.so")
REF_MEM = os.path.join(ROOT, "oracle", "_ref", "libmem.so")

_cache = {}


def oracle_backend():
    if "oracle" not in _cache:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        _cache["oracle"] = abi.Backend("oracle", ORACLE_SO, ORACLE_SO, device_memory=False)
    return _cache["oracle"]


def have_ref():
    return os.path.exists(REF_ALGO) and os.path.exists(REF_MEM)


def ref_backend():
    if "ref" not in _cache:
        _cache["ref"] = abi.Backend("ref", REF_ALGO, REF_MEM, device_memory=False)
    return _cache["ref"]


def hip_backend():
    if "hip" not in _cache:
        _cache["hip"] = abi.load_hip_backend()
        _cache["hip"].call("BootstrapDevice")
    return _cache["hip"]


def get_backend(name):
    return {"oracle": oracle_backend, "ref": ref_backend, "hip": hip_backend}[name]()


class _PinnedPool:
    """Host staging buffers from the backend's own HostAlloc (pinned, portable memory on the HIP backend — what the Go
    host's memutils.HostAlloc hands out and what every one of its uploads starts from, memstore/vectors/vector.go:75-84),
    cached by size class per thread.  ARES_TEST_PAGEABLE=1 switches back to plain numpy memory (pageable: the HIP runtime
    stages such copies itself)."""

    def __init__(self):
        self.tls = threading.local()

    def take(self, be, nbytes):
        size = 256
        while size < nbytes:
            size *= 2
        free = getattr(self.tls, "free", None)
        if free is None:
            free = self.tls.free = {}
        lst = free.setdefault((be.name, size), [])
        ptr = lst.pop() if lst else be.call("HostAlloc", size)
        return ptr, size

    def give(self, be, ptr, size):
        self.tls.free[(be.name, size)].append(ptr)


_pinned = _PinnedPool()
_PAGEABLE = os.environ.get("ARES_TEST_PAGEABLE") == "1"


def upload(be, dst, data, stream=None):
    """host numpy array -> 'device' memory of the backend, complete on return"""
    data = np.ascontiguousarray(data)
    if not data.nbytes:
        return
    if _PAGEABLE:
        be.h2d(dst, data.ctypes.data_as(C.c_void_p), data.nbytes, stream)
        be.wait(stream)
        return
    ptr, size = _pinned.take(be, data.nbytes)
    C.memmove(ptr, data.ctypes.data, data.nbytes)
    be.h2d(dst, ptr, data.nbytes, stream)
    be.wait(stream)
    _pinned.give(be, ptr, size)


def download(be, src, nbytes, stream=None):
    """'device' memory of the backend -> a fresh numpy uint8 array, complete on return"""
    nbytes = int(nbytes)
    out = np.empty(nbytes, np.uint8)
    if not nbytes:
        return out
    if _PAGEABLE:
        be.d2h(out.ctypes.data_as(C.c_void_p), src, nbytes, stream)
        be.wait(stream)
        return out
    ptr, size = _pinned.take(be, nbytes)
    be.d2h(ptr, src, nbytes, stream)
    be.wait(stream)
    C.memmove(out.ctypes.data, ptr, nbytes)
    _pinned.give(be, ptr, size)
    return out


class Buf:
    """A 'device' allocation of one backend, filled from / read back into numpy."""

    def __init__(self, be, data=None, nbytes=None):
        self.be = be
        if data is not None:
            data = np.ascontiguousarray(data)
            nbytes = data.nbytes
        self.nbytes = int(nbytes)
        self.ptr = be.device_alloc(self.nbytes + 16)  # DeviceAllocate zero-fills
        if data is not None and self.nbytes:
            self.write(data)

    def write(self, data, offset=0):
        upload(self.be, self.ptr + offset, data)

    def read(self, dtype=np.uint8, count=None, offset=0):
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset) // dtype.itemsize
        return download(self.be, self.ptr + offset, count * dtype.itemsize).view(dtype)

    def free(self):
        if self.ptr:
            self.be.device_free(self.ptr)
            self.ptr = 0


def align(x, a=8):
    return (x + a - 1) // a * a


def pack_bits(bools, starting_index=0):
    """LSB-first bit packing used for validity vectors and Bool values (1 = valid)."""
    bools = np.asarray(bools, dtype=bool)
    padded = np.concatenate([np.zeros(starting_index, dtype=bool), bools])
    return np.packbits(padded, bitorder="little")


_NP_OF = {abi.Int8: np.int8, abi.Uint8: np.uint8, abi.Int16: np.int16, abi.Uint16: np.uint16,
          abi.Int32: np.int32, abi.Uint32: np.uint32, abi.Float32: np.float32,
          abi.Int64: np.int64}


def make_default(dtype=abi.Int32, value=None):
    dv = abi.DefaultValue()
    dv.HasDefault = value is not None
    if value is not None:
        if dtype == abi.Bool:
            dv.Value.BoolVal = bool(value)
        elif dtype in (abi.Int8, abi.Int16, abi.Int32):
            dv.Value.Int32Val = int(value)
        elif dtype in (abi.Uint8, abi.Uint16, abi.Uint32):
            dv.Value.Uint32Val = int(value)
        elif dtype == abi.Float32:
            dv.Value.FloatVal = float(value)
        elif dtype == abi.Int64:
            dv.Value.Int64Val = int(value)
    return dv


class Column:
    """One VectorParty slice laid out as [counts u32 x(len+1)][validity bitmap][values]
    (reference query/aql_processor.go:1415-1429), sections aligned like the reference's tests."""

    def __init__(self, be, dtype, values=None, valid=None, counts=None, starting_index=0,
                 default=None, raw_values=None, alignment=8):
        self.be, self.dtype = be, dtype
        self.buf = None
        vp = abi.VectorPartySlice()
        vp.DataType = dtype
        vp.DefaultValue = make_default(dtype, default)
        vp.StartingIndex = starting_index
        if values is None and raw_values is None:  # mode 0
            vp.BasePtr = None
            vp.Length = 0
        else:
            if raw_values is not None:
                vbytes = np.frombuffer(bytes(raw_values), dtype=np.uint8)
                length = len(vbytes) // abi.DATA_TYPE_BYTES[dtype]
            elif dtype == abi.Bool:
                vbytes = pack_bits(values, starting_index)
                length = len(values)
            else:
                arr = np.asarray(values).astype(_NP_OF[dtype])
                vbytes = arr.view(np.uint8)
                length = len(arr)
            cbytes = np.zeros(0, np.uint8) if counts is None else \
                np.asarray(counts, dtype=np.uint32).view(np.uint8)
            nbytes_ = np.zeros(0, np.uint8) if valid is None else pack_bits(valid, starting_index)
            if counts is not None and valid is None:
                raise ValueError("mode 3 needs a validity vector")
            co, no = 0, align(len(cbytes), alignment)
            vo = no + align(len(nbytes_), alignment)
            blob = np.zeros(vo + align(len(vbytes), alignment) + alignment, np.uint8)
            blob[co:co + len(cbytes)] = cbytes
            blob[no:no + len(nbytes_)] = nbytes_
            blob[vo:vo + len(vbytes)] = vbytes
            self.blob = blob  # what was uploaded (diagnostics compare the device copy with it)
            self.buf = Buf(be, blob)
            if counts is not None:      # mode 3
                vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = self.buf.ptr, no, vo
            elif valid is not None:     # mode 2
                vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = self.buf.ptr + no, 0, vo - no
            else:                       # mode 1
                vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = self.buf.ptr + vo, 0, 0
            vp.Length = length if counts is None else len(counts) - 1
        self.vp = vp

    def input(self):
        iv = abi.InputVector()
        iv.Vector.VP = self.vp
        iv.Type = abi.VectorPartyInput
        return iv

    def free(self):
        if self.buf:
            self.buf.free()


def const_int(v, valid=True):
    iv = abi.InputVector()
    iv.Vector.Constant.Value.IntVal = int(v)
    iv.Vector.Constant.IsValid = valid
    iv.Vector.Constant.DataType = abi.ConstInt
    iv.Type = abi.ConstantInput
    return iv


def const_float(v, valid=True):
    iv = abi.InputVector()
    iv.Vector.Constant.Value.FloatVal = float(v)
    iv.Vector.Constant.IsValid = valid
    iv.Vector.Constant.DataType = abi.ConstFloat
    iv.Type = abi.ConstantInput
    return iv


class Scratch:
    """Scratch-space vector: values[n] (4 bytes each) then one validity byte per row."""

    def __init__(self, be, n, dtype, values=None, valid=None, width=4):
        self.be, self.n, self.dtype, self.width = be, n, dtype, width
        self.nulls_offset = align(width * n, 8)
        blob = np.zeros(self.nulls_offset + align(n, 8) + 8, np.uint8)
        if values is not None:
            arr = np.asarray(values).astype(_NP_OF[dtype]).view(np.uint8)
            blob[:len(arr)] = arr
        if valid is not None:
            blob[self.nulls_offset:self.nulls_offset + n] = np.asarray(valid, dtype=np.uint8)
        self.buf = Buf(be, blob)

    def _vec(self):
        s = abi.ScratchSpaceVector()
        s.Values, s.NullsOffset, s.DataType = self.buf.ptr, self.nulls_offset, self.dtype
        return s

    def input(self):
        iv = abi.InputVector()
        iv.Vector.ScratchSpace = self._vec()
        iv.Type = abi.ScratchSpaceInput
        return iv

    def output(self):
        ov = abi.OutputVector()
        ov.Vector.ScratchSpace = self._vec()
        ov.Type = abi.ScratchSpaceOutput
        return ov

    def values(self):
        return self.buf.read(_NP_OF[self.dtype], self.n)

    def valid(self):
        return self.buf.read(np.uint8, self.n, self.nulls_offset)

    def free(self):
        self.buf.free()


def measure_output(ptr, dtype, agg):
    ov = abi.OutputVector()
    ov.Vector.Measure.Values, ov.Vector.Measure.DataType, ov.Vector.Measure.AggFunc = ptr, dtype, agg
    ov.Type = abi.MeasureOutput
    return ov


def dimension_output(values_ptr, nulls_ptr, dtype):
    ov = abi.OutputVector()
    ov.Vector.Dimension.DimValues = values_ptr
    ov.Vector.Dimension.DimNulls = nulls_ptr
    ov.Vector.Dimension.DataType = dtype
    ov.Type = abi.DimensionOutput
    return ov


DIM_WIDTHS = (16, 8, 4, 2, 1)


class DimVector:
    """Columnar dimension store (reference query/common/dimval.go:122-144)."""

    def __init__(self, be, capacity, num_dims_per_width, with_hash=True, with_index=True, init=None):
        self.be, self.capacity = be, capacity
        self.ndw = tuple(num_dims_per_width)
        self.num_dims = sum(self.ndw)
        self.value_bytes = sum(w * c for w, c in zip(DIM_WIDTHS, self.ndw))
        self.nbytes = (self.value_bytes + self.num_dims) * capacity
        if init is not None:
            blob = np.zeros(self.nbytes, np.uint8)
            init = np.asarray(init, dtype=np.uint8)
            blob[:len(init)] = init
            self.values = Buf(be, blob)
        else:
            self.values = Buf(be, nbytes=self.nbytes)
        self.hash = Buf(be, nbytes=8 * capacity) if with_hash else None
        self.index = Buf(be, nbytes=4 * capacity) if with_index else None

    def dim_offsets(self):
        """(value byte offset, validity byte offset, width) per dimension."""
        out, off, d = [], 0, 0
        for w, c in zip(DIM_WIDTHS, self.ndw):
            for _ in range(c):
                out.append([off, self.value_bytes * self.capacity + d * self.capacity, w])
                off += w * self.capacity
                d += 1
        return out

    def struct(self):
        dv = abi.DimensionVector()
        dv.DimValues = self.values.ptr
        dv.HashValues = self.hash.ptr if self.hash else None
        dv.IndexVector = self.index.ptr if self.index else None
        dv.VectorCapacity = self.capacity
        for i, c in enumerate(self.ndw):
            dv.NumDimsPerDimWidth[i] = c
        return dv

    def rows(self, n):
        """First n rows as a list of tuples ((value bytes...), (valid...))."""
        blob = self.values.read(np.uint8)
        res = []
        offs = self.dim_offsets()
        for r in range(n):
            vals = tuple(bytes(blob[vo + r * w: vo + (r + 1) * w]) for vo, _, w in offs)
            valid = tuple(int(blob[no + r]) for _, no, _ in offs)
            res.append((vals, valid))
        return res

    def free(self):
        for b in (self.values, self.hash, self.index):
            if b:
                b.free()


def record_id_array(pairs):
    arr = np.zeros(len(pairs), dtype=[("batchID", np.int32), ("index", np.uint32)])
    for i, (b, x) in enumerate(pairs):
        arr[i] = (b, x)
    return arr


def read_device_bytes(be, ptr, nbytes):
    return download(be, ptr, nbytes)


def hyperloglog(be, prev, cur, prev_values, cur_values, prev_size, batch_size, last):
    """One HyperLogLog call (query/hll.cu:21-60).  Returns (result size, hll vector bytes,
    registers per dimension) — the last two are None unless the call produced them; the buffers the
    callee allocated are released with DeviceFree, as the Go host does
    (query/aql_postprocessor.go:217-219)."""
    vec, size, counts = C.c_void_p(0), C.c_size_t(0), C.c_void_p(0)
    n = be.call("HyperLogLog", prev.struct(), cur.struct(), prev_values.ptr, cur_values.ptr, prev_size, batch_size,
                bool(last), C.addressof(vec), C.addressof(size), C.addressof(counts), None, 0)
    be.wait()
    hll, reg = None, None
    if vec.value:
        hll = read_device_bytes(be, vec.value, size.value)
        be.device_free(vec.value)
    if counts.value:
        reg = read_device_bytes(be, counts.value, 2 * n).view(np.uint16)
        be.device_free(counts.value)
    return n, hll, reg


class GeoShapes:
    """GeoShapeBatch on a backend: [latitudes f32][longitudes f32][shape index u8] of all polygon
    points (query/time_series_aggregate.h:316-330, query/unittest_utils.hpp:270-293)."""

    def __init__(self, be, lats, longs, shape_index, num_shapes):
        lats = np.asarray(lats, np.float32)
        longs = np.asarray(longs, np.float32)
        shape_index = np.asarray(shape_index, np.uint8)
        self.num_points = len(lats)
        self.total_words = (num_shapes + 31) // 32
        blob = np.concatenate([lats.view(np.uint8), longs.view(np.uint8), shape_index])
        self.buf = Buf(be, blob)

    def struct(self):
        g = abi.GeoShapeBatch()
        g.LatLongs, g.TotalNumPoints, g.TotalWords = self.buf.ptr, self.num_points, self.total_words
        return g

    def free(self):
        self.buf.free()


def geo_column(be, points, valid=None, starting_index=0):
    return Column(be, abi.GeoPoint, raw_values=np.asarray(points, np.float32).tobytes(), valid=valid,
                  starting_index=starting_index)
