"""ctypes mirror of the drop-in C ABI (include/ares_algorithm.h, include/ares_memory.h).

This is the Python-side equivalent of the reference's cgo glue
(query/time_series_aggregate.go:17-19, cgoutils/memory.go:17-19, cgoutils/utils.go:26-34):
struct layouts, enum values, the CGoCallResHandle error convention, and a `Backend` that binds
one (libalgorithm, libmem) pair.  Nothing in here computes anything; a missing shared library is
a hard error (`load_hip_backend` never falls back to a CPU implementation).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
# ARES_LIB_DIR: another build of the same libraries (diagnostics: the AddressSanitizer build of `make asan`)
LIB_DIR = os.environ.get("ARES_LIB_DIR") or os.path.join(_HERE, "lib")

# ---- enums (include/ares_algorithm.h; positional, part of the ABI) -------------------------
AGGR_SUM_UNSIGNED, AGGR_SUM_SIGNED, AGGR_SUM_FLOAT = 1, 2, 3
AGGR_MIN_UNSIGNED, AGGR_MIN_SIGNED, AGGR_MIN_FLOAT = 4, 5, 6
AGGR_MAX_UNSIGNED, AGGR_MAX_SIGNED, AGGR_MAX_FLOAT = 7, 8, 9
AGGR_HLL, AGGR_AVG_FLOAT = 10, 11

(Bool, Int8, Uint8, Int16, Uint16, Int32, Uint32, Float32, Int64, Uint64, Float64, GeoPoint,
 UUID) = range(13)
ConstInt, ConstFloat, ConstGeoPoint, ConstUUID = range(4)
(Negate, Not, BitwiseNot, IsNull, IsNotNull, Noop, GetWeekStart, GetMonthStart, GetQuarterStart,
 GetYearStart, GetDayOfMonth, GetDayOfYear, GetMonthOfYear, GetQuarterOfYear, GetHLLValue,
 ArrayLength) = range(16)
(And, Or, Equal, NotEqual, LessThan, LessThanOrEqual, GreaterThan, GreaterThanOrEqual, Plus, Minus,
 Multiply, Divide, Mod, BitwiseAnd, BitwiseOr, BitwiseXor, Floor, ArrayContains,
 ArrayElementAt) = range(19)
(VectorPartyInput, ScratchSpaceInput, ConstantInput, ForeignColumnInput,
 ArrayVectorPartyInput) = range(5)
ScratchSpaceOutput, MeasureOutput, DimensionOutput = range(3)
NUM_DIM_WIDTH = 5

DEVICE_MEMORY_IMPLEMENTATION_FLAG, POOLED_MEMORY_FLAG, HASH_REDUCTION_SUPPORT = 1, 2, 4

DATA_TYPE_BYTES = {Bool: 1, Int8: 1, Uint8: 1, Int16: 2, Uint16: 2, Int32: 4, Uint32: 4,
                   Float32: 4, Int64: 8, Uint64: 8, Float64: 8, GeoPoint: 8, UUID: 16}


# ---- structs ---------------------------------------------------------------------------------
class CGoCallResHandle(C.Structure):
    _fields_ = [("res", C.c_void_p), ("pStrErr", C.c_void_p)]


class RecordID(C.Structure):
    _fields_ = [("batchID", C.c_int32), ("index", C.c_uint32)]


class CuckooHashIndex(C.Structure):
    _fields_ = [("buckets", C.c_void_p), ("seeds", C.c_uint32 * 4), ("keyBytes", C.c_int),
                ("numHashes", C.c_int), ("numBuckets", C.c_int)]


class GeoPointT(C.Structure):
    _fields_ = [("Lat", C.c_float), ("Long", C.c_float)]


class UUIDT(C.Structure):
    _fields_ = [("p1", C.c_uint64), ("p2", C.c_uint64)]


class _DefaultValueUnion(C.Union):
    _fields_ = [("BoolVal", C.c_bool), ("Int32Val", C.c_int32), ("Uint32Val", C.c_uint32),
                ("FloatVal", C.c_float), ("Int64Val", C.c_int64), ("GeoPointVal", GeoPointT),
                ("UUIDVal", UUIDT)]


class DefaultValue(C.Structure):
    _fields_ = [("HasDefault", C.c_bool), ("Value", _DefaultValueUnion)]


class VectorPartySlice(C.Structure):
    _fields_ = [("BasePtr", C.c_void_p), ("NullsOffset", C.c_uint32), ("ValuesOffset", C.c_uint32),
                ("StartingIndex", C.c_uint8), ("DataType", C.c_int), ("DefaultValue", DefaultValue),
                ("Length", C.c_uint32)]


class ScratchSpaceVector(C.Structure):
    _fields_ = [("Values", C.c_void_p), ("NullsOffset", C.c_uint32), ("DataType", C.c_int)]


class _ConstUnion(C.Union):
    _fields_ = [("IntVal", C.c_int32), ("FloatVal", C.c_float), ("GeoPointVal", GeoPointT),
                ("UUIDVal", UUIDT)]


class ConstantVector(C.Structure):
    _fields_ = [("Value", _ConstUnion), ("IsValid", C.c_bool), ("DataType", C.c_int)]


class ForeignColumnVector(C.Structure):
    _fields_ = [("RecordIDs", C.c_void_p), ("Batches", C.c_void_p), ("BaseBatchID", C.c_int32),
                ("NumBatches", C.c_int32), ("NumRecordsInLastBatch", C.c_int32),
                ("TimezoneLookup", C.c_void_p), ("TimezoneLookupSize", C.c_int16),
                ("DataType", C.c_int), ("DefaultValue", DefaultValue)]


class ArrayVectorPartySlice(C.Structure):
    _fields_ = [("OffsetLengthVector", C.c_void_p), ("ValueOffsetAdj", C.c_uint32),
                ("DataType", C.c_int), ("Length", C.c_uint32)]


class _InputUnion(C.Union):
    _fields_ = [("Constant", ConstantVector), ("VP", VectorPartySlice),
                ("ScratchSpace", ScratchSpaceVector), ("ForeignVP", ForeignColumnVector),
                ("ArrayVP", ArrayVectorPartySlice)]


class InputVector(C.Structure):
    _fields_ = [("Vector", _InputUnion), ("Type", C.c_int)]


class DimensionVector(C.Structure):
    _fields_ = [("DimValues", C.c_void_p), ("HashValues", C.c_void_p), ("IndexVector", C.c_void_p),
                ("VectorCapacity", C.c_int), ("NumDimsPerDimWidth", C.c_uint8 * NUM_DIM_WIDTH)]


class DimensionOutputVector(C.Structure):
    _fields_ = [("DimValues", C.c_void_p), ("DimNulls", C.c_void_p), ("DataType", C.c_int)]


class MeasureOutputVector(C.Structure):
    _fields_ = [("Values", C.c_void_p), ("DataType", C.c_int), ("AggFunc", C.c_int)]


class _OutputUnion(C.Union):
    _fields_ = [("ScratchSpace", ScratchSpaceVector), ("Dimension", DimensionOutputVector),
                ("Measure", MeasureOutputVector)]


class OutputVector(C.Structure):
    _fields_ = [("Vector", _OutputUnion), ("Type", C.c_int)]


class GeoShapeBatch(C.Structure):
    _fields_ = [("LatLongs", C.c_void_p), ("TotalNumPoints", C.c_int32), ("TotalWords", C.c_uint8)]


# sizes pinned by include/ares_algorithm.h's static assertions
ABI_SIZES = {CGoCallResHandle: 16, RecordID: 8, CuckooHashIndex: 40, DefaultValue: 24,
             VectorPartySlice: 56, ScratchSpaceVector: 16, ConstantVector: 24,
             ForeignColumnVector: 72, ArrayVectorPartySlice: 24, InputVector: 80,
             DimensionVector: 40, DimensionOutputVector: 24, MeasureOutputVector: 16,
             OutputVector: 32, GeoShapeBatch: 16}

_VP, _S, _I = C.c_void_p, C.c_void_p, C.c_int

ALGORITHM_SYMBOLS = {
    "InitIndexVector": [_VP, C.c_uint32, _I, _S, _I],
    "HashLookup": [InputVector, _VP, _VP, _I, _VP, C.c_uint32, CuckooHashIndex, _S, _I],
    "UnaryTransform": [InputVector, OutputVector, _VP, _I, _VP, C.c_uint32, _I, _S, _I],
    "UnaryFilter": [InputVector, _VP, _VP, _I, _VP, _I, _VP, C.c_uint32, _I, _S, _I],
    "BinaryTransform": [InputVector, InputVector, OutputVector, _VP, _I, _VP, C.c_uint32, _I, _S, _I],
    "BinaryFilter": [InputVector, InputVector, _VP, _VP, _I, _VP, _I, _VP, C.c_uint32, _I, _S, _I],
    "Sort": [DimensionVector, _I, _S, _I],
    "Reduce": [DimensionVector, _VP, DimensionVector, _VP, _I, _I, _I, _S, _I],
    "HashReduce": [DimensionVector, _VP, DimensionVector, _VP, _I, _I, _I, _S, _I],
    "Expand": [DimensionVector, DimensionVector, _VP, _VP, _I, _I, _S, _I],
    "HyperLogLog": [DimensionVector, DimensionVector, _VP, _VP, _I, _I, C.c_bool, _VP, _VP, _VP, _S, _I],
    "GeoBatchIntersects": [GeoShapeBatch, InputVector, _VP, _I, C.c_uint32, _VP, _I, _VP, C.c_bool, _S, _I],
    "WriteGeoShapeDim": [_I, DimensionOutputVector, _I, _VP, _S, _I],
    "BootstrapDevice": [],
}

MEMORY_SYMBOLS = {
    "HostAlloc": [C.c_size_t],
    "HostFree": [_VP],
    "HostMemCpy": [_VP, _VP, C.c_size_t],
    "CreateCudaStream": [_I],
    "WaitForCudaStream": [_S, _I],
    "DestroyCudaStream": [_S, _I],
    "DeviceAllocate": [C.c_size_t, _I],
    "DeviceFree": [_VP, _I],
    "AsyncCopyHostToDevice": [_VP, _VP, C.c_size_t, _S, _I],
    "AsyncCopyDeviceToDevice": [_VP, _VP, C.c_size_t, _S, _I],
    "AsyncCopyDeviceToHost": [_VP, _VP, C.c_size_t, _S, _I],
    "GetDeviceCount": [],
    "GetDeviceGlobalMemoryInMB": [_I],
    "CudaProfilerStart": [],
    "CudaProfilerStop": [],
    "GetDeviceMemoryInfo": [_VP, _VP, _I],
    "deviceMalloc": [_VP, C.c_size_t],
    "deviceFree": [_VP],
    "deviceMemset": [_VP, _I, C.c_size_t],
    "asyncCopyHostToDevice": [_VP, _VP, C.c_size_t, _S],
    "asyncCopyDeviceToHost": [_VP, _VP, C.c_size_t, _S],
    "waitForCudaStream": [_S],
}


class AresError(RuntimeError):
    """Raised where the Go host would panic (cgoutils/utils.go:26-34)."""


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _check(handle):
    """DoCGoCall: turn pStrErr into an exception and free it; return `res` as an int."""
    if handle.pStrErr:
        msg = C.string_at(handle.pStrErr).decode("utf-8", "replace")
        _libc.free(handle.pStrErr)
        raise AresError(msg.strip())
    return handle.res or 0


class Backend:
    """One (libalgorithm, libmem) pair bound through the C ABI.

    `device_memory` says whether pointers handed to the algorithm library are device pointers
    (HIP build) or plain host pointers (reference QUERY_MODE=HOST / the C oracle).
    """

    def __init__(self, name, algorithm_path, memory_path, device_memory):
        self.name = name
        self.device_memory = device_memory
        for p in (memory_path, algorithm_path):
            if not os.path.exists(p):
                raise FileNotFoundError(
                    f"{name}: shared library {p} is missing — build it first "
                    f"(python -c 'import __graft_entry__ as g; g.build()')")
        # RTLD_LOCAL: several backends export the same symbol names in one process
        self._mem = C.CDLL(memory_path, mode=os.RTLD_NOW | os.RTLD_LOCAL)
        self._algo = self._mem if algorithm_path == memory_path else \
            C.CDLL(algorithm_path, mode=os.RTLD_NOW | os.RTLD_LOCAL)
        self.algorithm_path, self.memory_path = algorithm_path, memory_path
        for sym, argtypes in ALGORITHM_SYMBOLS.items():
            fn = getattr(self._algo, sym)  # AttributeError if the export is missing
            fn.argtypes, fn.restype = argtypes, CGoCallResHandle
            setattr(self, "_" + sym, fn)
        for sym, argtypes in MEMORY_SYMBOLS.items():
            fn = getattr(self._mem, sym)
            fn.argtypes, fn.restype = argtypes, CGoCallResHandle
            setattr(self, "_" + sym, fn)
        self._mem.GetFlags.argtypes, self._mem.GetFlags.restype = [], C.c_uint32
        # optional extensions of the MI355X build (include/ares_extensions.h)
        self.has_profiler = hasattr(self._algo, "AresProfilerEnable")
        if self.has_profiler:
            self._algo.AresProfilerEnable.argtypes, self._algo.AresProfilerEnable.restype = [C.c_int], None
            self._algo.AresProfilerReport.argtypes = [C.c_char_p, C.c_size_t]
            self._algo.AresProfilerReport.restype = C.c_size_t

    def call(self, sym, *args):
        return _check(getattr(self, "_" + sym)(*args))

    def mem_driver_calls(self, device=0):
        """{mallocs, frees, trims}: driver calls libmem.so's block cache could not avoid (None: not this libmem)."""
        if not hasattr(self._mem, "AresMemDriverCalls"):
            return None
        m, f, t = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._mem.AresMemDriverCalls.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self._mem.AresMemDriverCalls.restype = None
        self._mem.AresMemDriverCalls(device, C.byref(m), C.byref(f), C.byref(t))
        return {"mallocs": m.value, "frees": f.value, "trims": t.value}

    def rtc_wait(self):
        """Blocks until no kernel is being compiled in the background; {kernels, compiles, disk_hits, evictions}."""
        if not hasattr(self._algo, "AresRtcWait"):
            return None
        c = (C.c_long * 3)()
        self._algo.AresRtcWait.argtypes, self._algo.AresRtcWait.restype = [C.POINTER(C.c_long)], C.c_size_t
        n = self._algo.AresRtcWait(c)
        return {"kernels": int(n), "compiles": int(c[0]), "disk_hits": int(c[1]), "evictions": int(c[2])}

    def reload_env(self):
        """The library's latched environment switches (ARES_HASH_REDUCE, ...) are read again on next use."""
        if hasattr(self._algo, "AresReloadEnv"):
            self._algo.AresReloadEnv.argtypes, self._algo.AresReloadEnv.restype = [], None
            self._algo.AresReloadEnv()

    def flags(self):
        return self._mem.GetFlags()

    def profiler_enable(self, on=True):
        self._algo.AresProfilerEnable(1 if on else 0)

    def profiler_report(self):
        """{kernel name: (launches, total ms)} since the last profiler_enable(True); the caller has
        synchronised its streams."""
        need = self._algo.AresProfilerReport(None, 0)
        buf = C.create_string_buffer(need + 16)
        self._algo.AresProfilerReport(buf, need + 16)
        out = {}
        for line in buf.value.decode().splitlines():
            name, launches, ms = line.rsplit(" ", 2)
            out[name] = (int(launches), float(ms))
        return out

    # -- convenience wrappers used by tests / host code ----------------------------------------
    def device_alloc(self, nbytes, device=0):
        return self.call("DeviceAllocate", max(int(nbytes), 1), device)

    def device_free(self, ptr, device=0):
        self.call("DeviceFree", ptr, device)

    def h2d(self, dst, src_buf, nbytes, stream=None, device=0):
        self.call("AsyncCopyHostToDevice", dst, src_buf, nbytes, stream, device)

    def d2h(self, dst_buf, src, nbytes, stream=None, device=0):
        self.call("AsyncCopyDeviceToHost", dst_buf, src, nbytes, stream, device)

    def wait(self, stream=None, device=0):
        self.call("WaitForCudaStream", stream, device)


def hip_library_paths():
    return os.path.join(LIB_DIR, "libalgorithm.so"), os.path.join(LIB_DIR, "libmem.so")


def load_hip_backend():
    """The product: hand-written HIP libalgorithm.so + libmem.so.  No fallback of any kind."""
    # One HIP runtime per process: torch bundles its own libamdhip64 and must be the first to load it —
    # loaded after ours (which resolves to /opt/rocm), torch finds "no ROCm-capable device".
    if not os.environ.get("ARES_NO_TORCH"):  # (diagnostics run without torch: then /opt/rocm's runtime is the only one)
        import torch  # noqa: F401
    algo, mem = hip_library_paths()
    return Backend("hip", algo, mem, device_memory=True)
