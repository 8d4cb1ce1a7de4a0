"""ctypes binding of libaresdriver.so — the C++ host side of the batch pipeline
(aresdb_amd/csrc/host/ares_driver.cpp, a mirror of the reference's Go batch executor).

`NativeQuery` has the same surface as executor.BatchContext/BatchExecutor/fetch_results, so tests
run one plan through the Python mirror and the C++ driver and compare; bench.py uses the C++ driver
so that the host side of the timed region is compiled code, as it is in the reference."""
import ctypes as C
import os

import numpy as np

from . import abi
from .executor import DIM_WIDTHS, Binary, Col, Const, QueryPlan, Unary

DRIVER_PATH = os.path.join(abi.LIB_DIR, "libaresdriver.so")

(NODE_COLUMN, NODE_CONST_INT, NODE_CONST_FLOAT, NODE_UNARY, NODE_BINARY) = range(5)


class PlanNode(C.Structure):
    _fields_ = [("kind", C.c_int), ("op", C.c_int), ("lhs", C.c_int), ("rhs", C.c_int), ("table", C.c_int),
                ("column", C.c_int), ("ival", C.c_int32), ("fval", C.c_float), ("outType", C.c_int)]


class ForeignTableC(C.Structure):
    _fields_ = [("joinColumn", C.c_int), ("index", abi.CuckooHashIndex), ("numColumns", C.c_int),
                ("numBatches", C.c_int), ("slices", C.POINTER(abi.VectorPartySlice)), ("dataTypes", C.POINTER(C.c_int)),
                ("baseBatchID", C.c_int32), ("numRecordsInLastBatch", C.c_int32)]


class GeoIntersectionC(C.Structure):
    _fields_ = [("shapeLatLongs", C.c_void_p), ("numShapes", C.c_int), ("totalNumPoints", C.c_int),
                ("pointTable", C.c_int), ("pointColumn", C.c_int), ("inOrOut", C.c_int), ("dimIndex", C.c_int)]


class QueryPlanC(C.Structure):
    _fields_ = [("nodes", C.POINTER(PlanNode)), ("numNodes", C.c_int),
                ("filters", C.POINTER(C.c_int)), ("numFilters", C.c_int),
                ("foreignFilters", C.POINTER(C.c_int)), ("numForeignFilters", C.c_int),
                ("dimNodes", C.POINTER(C.c_int)), ("dimTypes", C.POINTER(C.c_int)), ("numDims", C.c_int),
                ("measureNode", C.c_int), ("aggFunc", C.c_int), ("measureType", C.c_int),
                ("useHashReduction", C.c_int),
                ("foreignTables", C.POINTER(ForeignTableC)), ("numForeignTables", C.c_int),
                ("useFusedExtension", C.c_int), ("geo", C.POINTER(GeoIntersectionC))]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p,
                          C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p)


class HostColumnC(C.Structure):
    _fields_ = [("host", C.c_void_p), ("bytes", C.c_size_t), ("slice", abi.VectorPartySlice), ("cacheKey", C.c_uint64)]


_lib = None


def _driver():
    global _lib
    if _lib is None:
        if not os.path.exists(DRIVER_PATH):
            raise FileNotFoundError(f"{DRIVER_PATH} is missing — build it first "
                                    f"(python -c 'import __graft_entry__ as g; g.build()')")
        lib = C.CDLL(DRIVER_PATH, mode=os.RTLD_NOW | os.RTLD_LOCAL)
        lib.AresDriverOpen.argtypes, lib.AresDriverOpen.restype = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int], C.c_void_p
        lib.AresDriverClose.argtypes = [C.c_void_p]
        lib.AresQueryCreate.argtypes = [C.c_void_p, C.POINTER(QueryPlanC), C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        lib.AresQueryCreate.restype = C.c_void_p
        lib.AresQueryRunBatch.argtypes = [C.c_void_p, C.POINTER(abi.VectorPartySlice), C.c_int, C.c_int, C.c_void_p,
                                          C.c_uint32, C.c_char_p, C.c_int]
        lib.AresQueryRunBatch.restype = C.c_int
        lib.AresQueryRunResidentBatches.argtypes = [C.c_void_p, C.POINTER(abi.VectorPartySlice), C.c_int, C.POINTER(C.c_int), C.c_int,
                                                    C.c_char_p, C.c_int]
        lib.AresQueryRunResidentBatches.restype = C.c_int
        for name, res in (("AresQueryResultSize", C.c_int), ("AresQueryResultCapacity", C.c_int),
                          ("AresQueryDimensionVector", C.c_void_p), ("AresQueryMeasureVector", C.c_void_p),
                          ("AresQueryNumCalls", C.c_long), ("AresQueryNumFusedBatches", C.c_long)):
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = [C.c_void_p], res
        lib.AresQueryFetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        lib.AresQueryFetch.restype = C.c_int
        lib.AresQueryDestroy.argtypes = [C.c_void_p]
        lib.AresQuerySetLastBatch.argtypes = [C.c_void_p, C.c_int]
        lib.AresQuerySetSecondStream.argtypes = [C.c_void_p, C.c_void_p]
        lib.AresQueryAdoptColumns.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]
        lib.AresCommCreate.argtypes, lib.AresCommCreate.restype = [C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p], C.c_void_p
        lib.AresCommRcclUniqueId.argtypes, lib.AresCommRcclUniqueId.restype = [C.c_void_p, C.c_char_p, C.c_int], C.c_int
        lib.AresCommCreateRccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        lib.AresCommCreateRccl.restype = C.c_void_p
        lib.AresCommDestroy.argtypes = [C.c_void_p]
        lib.AresQueryMergeShards.argtypes, lib.AresQueryMergeShards.restype = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int], C.c_int
        lib.AresQueryMergeShardsPartitioned.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_char_p, C.c_int]
        lib.AresQueryMergeShardsPartitioned.restype = C.c_int
        lib.AresCommSetAllToAll.argtypes, lib.AresCommSetAllToAll.restype = [C.c_void_p, ALLTOALL_FN], None
        lib.AresColumnCacheCreate.argtypes, lib.AresColumnCacheCreate.restype = [C.c_void_p, C.c_int, C.c_size_t], C.c_void_p
        lib.AresColumnCacheDestroy.argtypes = [C.c_void_p]
        lib.AresQueryRunHostBatches.argtypes = [C.c_void_p, C.POINTER(HostColumnC), C.c_int, C.POINTER(C.c_int), C.c_int,
                                                C.c_void_p, C.POINTER(C.c_uint64), C.c_char_p, C.c_int]
        lib.AresQueryRunHostBatches.restype = C.c_int
        lib.AresQueryHLLVectorSize.argtypes, lib.AresQueryHLLVectorSize.restype = [C.c_void_p], C.c_int64
        lib.AresQueryFetchHLL.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        lib.AresQueryFetchHLL.restype = C.c_int
        _lib = lib
    return _lib


_handles = {}


def _open(be: abi.Backend):
    key = (be.algorithm_path, be.memory_path)
    if key not in _handles:
        err = C.create_string_buffer(512)
        h = _driver().AresDriverOpen(be.algorithm_path.encode(), be.memory_path.encode(), err, 512)
        if not h:
            raise abi.AresError(err.value.decode())
        _handles[key] = h
    return _handles[key]


class NativeQuery:
    """One query on one device, executed by the C++ driver."""

    def __init__(self, be: abi.Backend, plan: QueryPlan, column_names, device=0, stream=None,
                 foreign_column_names=None, streams=None):
        """streams: two stream handles -> the driver alternates them per batch like the Go host
        (query/aql_processor.go:218,247); `stream` alone: every batch on that stream."""
        if streams:
            stream = streams[0]
        self.be, self.plan, self.device, self.stream = be, plan, device, stream
        self.column_names = list(column_names)
        self.foreign_column_names = foreign_column_names or [sorted(ft.batches) for ft in plan.foreign_tables]
        self._keep = []
        nodes = []

        def emit(e):
            n = PlanNode()
            if isinstance(e, Col):
                n.kind, n.table = NODE_COLUMN, e.table
                n.column = self.column_names.index(e.name) if e.table == 0 else \
                    self.foreign_column_names[e.table - 1].index(e.name)
            elif isinstance(e, Const):
                if isinstance(e.value, float):
                    n.kind, n.fval = NODE_CONST_FLOAT, e.value
                else:
                    n.kind, n.ival = NODE_CONST_INT, int(e.value)
            elif isinstance(e, Unary):
                n.kind, n.op, n.lhs, n.outType = NODE_UNARY, e.op, emit(e.arg), e.out_type
            elif isinstance(e, Binary):
                lhs = emit(e.lhs)
                rhs = emit(e.rhs)
                n.kind, n.op, n.lhs, n.rhs, n.outType = NODE_BINARY, e.op, lhs, rhs, e.out_type
            else:
                raise TypeError(f"unsupported expression node {e!r}")
            nodes.append(n)
            return len(nodes) - 1

        filters = [emit(f) for f in plan.filters]
        ffilters = [emit(f) for f in plan.foreign_filters]
        dim_nodes = [emit(d.expr) for d in plan.dimensions]
        measure = emit(plan.measure)

        def arr(ctype, values):
            a = (ctype * max(len(values), 1))(*values)
            self._keep.append(a)
            return a

        fts = []
        for ft, names in zip(plan.foreign_tables, self.foreign_column_names):
            nb = len(ft.batches[names[0]])
            slices = arr(abi.VectorPartySlice, [s for name in names for s in ft.batches[name]])
            types = arr(C.c_int, [ft.data_types[name] for name in names])
            f = ForeignTableC()
            f.joinColumn = self.column_names.index(ft.join_column)
            f.index = ft.index
            f.numColumns, f.numBatches = len(names), nb
            f.slices, f.dataTypes = slices, types
            f.baseBatchID, f.numRecordsInLastBatch = ft.base_batch_id, ft.num_records_in_last_batch
            fts.append(f)

        pc = QueryPlanC()
        pc.nodes, pc.numNodes = arr(PlanNode, nodes), len(nodes)
        pc.filters, pc.numFilters = arr(C.c_int, filters), len(filters)
        pc.foreignFilters, pc.numForeignFilters = arr(C.c_int, ffilters), len(ffilters)
        pc.dimNodes, pc.numDims = arr(C.c_int, dim_nodes), len(dim_nodes)
        pc.dimTypes = arr(C.c_int, [d.data_type for d in plan.dimensions])
        pc.measureNode, pc.aggFunc, pc.measureType = measure, plan.agg, plan.measure_type
        pc.useHashReduction = int(plan.use_hash_reduction)
        pc.foreignTables, pc.numForeignTables = arr(ForeignTableC, fts), len(fts)
        pc.useFusedExtension = int(getattr(plan, "use_fused_extension", False))
        geo = getattr(plan, "geo", None)
        if geo is not None:
            g = GeoIntersectionC()
            g.shapeLatLongs, g.numShapes, g.totalNumPoints = geo.shape_lat_longs, geo.num_shapes, geo.total_num_points
            g.pointTable, g.inOrOut, g.dimIndex = geo.point_table, int(geo.in_or_out), geo.dim_index
            g.pointColumn = self.column_names.index(geo.point_column) if geo.point_table == 0 else \
                self.foreign_column_names[geo.point_table - 1].index(geo.point_column)
            self._keep.append(g)
            pc.geo = C.pointer(g)
        err = C.create_string_buffer(512)
        self._q = _driver().AresQueryCreate(_open(be), C.byref(pc), device, stream, err, 512)
        if not self._q:
            raise abi.AresError(err.value.decode())
        if streams and len(streams) > 1:
            _driver().AresQuerySetSecondStream(self._q, streams[1])
        self._err = C.create_string_buffer(1024)

    def run(self, columns, size, base_counts=None, start_row=0, is_last_batch=False, owned_allocations=()):
        """columns: {name: VectorPartySlice} of the main table for this batch.  owned_allocations:
        device pointers of the batch's columns that the driver frees before the aggregation stage,
        like the Go host."""
        _driver().AresQuerySetLastBatch(self._q, int(is_last_batch))
        owned = (C.c_void_p * max(len(owned_allocations), 1))(*owned_allocations)
        _driver().AresQueryAdoptColumns(self._q, owned, len(owned_allocations))
        cols = (abi.VectorPartySlice * len(self.column_names))(*[columns[n] for n in self.column_names])
        rc = _driver().AresQueryRunBatch(self._q, cols, len(self.column_names), size, base_counts, start_row,
                                         self._err, 1024)
        if rc != 0:
            raise abi.AresError(self._err.value.decode().strip())

    def pack_batches(self, batches):
        """[(columns dict, size), ...] -> an opaque argument for run_batches (built once, reused for every step)"""
        nc = len(self.column_names)
        cols = (abi.VectorPartySlice * (nc * len(batches)))(*[b[0][n] for b in batches for n in self.column_names])
        sizes = (C.c_int * len(batches))(*[int(b[1]) for b in batches])
        return cols, nc, sizes, len(batches)

    def run_batches(self, packed):
        """every batch of a device-resident shard in one call of the C++ driver (no interpreter between batches)"""
        cols, nc, sizes, nb = packed
        rc = _driver().AresQueryRunResidentBatches(self._q, cols, nc, sizes, nb, self._err, 1024)
        if rc != 0:
            raise abi.AresError(self._err.value.decode().strip())

    @property
    def result_size(self):
        return _driver().AresQueryResultSize(self._q)

    @property
    def result_capacity(self):
        return _driver().AresQueryResultCapacity(self._q)

    @property
    def dim_vector(self):
        return _driver().AresQueryDimensionVector(self._q)

    @property
    def measure_vector(self):
        return _driver().AresQueryMeasureVector(self._q)

    @property
    def calls(self):
        return _driver().AresQueryNumCalls(self._q)

    @property
    def fused_batches(self):
        return _driver().AresQueryNumFusedBatches(self._q)

    # -- the attributes shard_merge.merge_shard_results reads from a batch context --------------------
    @property
    def ndw(self):
        return self.plan.num_dims_per_width()

    @property
    def dim_index(self):
        return self.plan.dim_vector_index()

    @property
    def dim_vec(self):
        return [self.dim_vector, 0]

    @property
    def measure_vec(self):
        return [self.measure_vector, 0]

    def call(self, sym, *args):
        return self.be.call(sym, *args)

    def fetch(self):
        """(dims, valids, measures) in query dimension order, like executor.fetch_results."""
        plan, n = self.plan, self.result_size
        widths = sorted([d.width for d in plan.dimensions], reverse=True)
        dims_blob = np.empty(max(n * (sum(widths) + len(widths)), 1), np.uint8)
        meas = np.empty(max(n * plan.measure_bytes, 1), np.uint8)
        rc = _driver().AresQueryFetch(self._q, dims_blob.ctypes.data_as(C.c_void_p), meas.ctypes.data_as(C.c_void_p),
                                      self._err, 1024)
        if rc != 0:
            raise abi.AresError(self._err.value.decode().strip())
        order = plan.dim_vector_index()
        starts = np.concatenate([[0], np.cumsum([w * n for w in widths])])
        null_base = int(starts[-1])
        dims, valids = [], []
        for q in range(len(plan.dimensions)):
            d = order[q]
            dims.append(dims_blob[starts[d]:starts[d] + widths[d] * n].copy())
            valids.append(dims_blob[null_base + d * n: null_base + (d + 1) * n].copy())
        return dims, valids, meas[:n * plan.measure_bytes].copy()

    def fetch_hll(self):
        """(dims, valids, registers per dimension, encoded HLL vector) of a HyperLogLog query."""
        dims, valids, _ = self.fetch()
        n = self.result_size
        counts = np.empty(max(n, 1), np.uint16)
        vec = np.empty(max(_driver().AresQueryHLLVectorSize(self._q) if n else 0, 1), np.uint8)
        rc = _driver().AresQueryFetchHLL(self._q, counts.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p),
                                         self._err, 1024)
        if rc != 0:
            raise abi.AresError(self._err.value.decode().strip())
        size = _driver().AresQueryHLLVectorSize(self._q) if n else 0
        return dims, valids, counts[:n].copy(), vec[:size].copy()

    def merge_shards(self, comm):
        """Replaces this query's result by the merged result of every rank of `comm` (all-gather of
        the partial group tables + re-reduce, inside libaresdriver.so)."""
        rc = _driver().AresQueryMergeShards(self._q, comm.handle, self._err, 1024)
        if rc != 0:
            raise abi.AresError(self._err.value.decode().strip())

    def merge_shards_partitioned(self, comm):
        """Hash-partitioned merge: this query's result becomes the groups whose 64-bit row hash falls into
        this rank's share of the hash range (no rank holds the whole table).  Returns the size of the whole
        result over all ranks."""
        total = C.c_int64(0)
        rc = _driver().AresQueryMergeShardsPartitioned(self._q, comm.handle, C.byref(total), self._err, 1024)
        if rc != 0:
            raise abi.AresError(self._err.value.decode().strip())
        return total.value

    def run_host_batches(self, batches, cache=None):
        """batches: [(list of HostColumn in column order, rows)] living in pinned host memory; the
        driver uploads batch k+1 while batch k executes.  Returns {uploaded_bytes, uploads, cache_hits,
        cache_bytes}."""
        ncol = len(self.column_names)
        flat = (HostColumnC * (ncol * len(batches)))()
        sizes = (C.c_int * len(batches))()
        for b, (cols, n) in enumerate(batches):
            sizes[b] = n
            for c, hc in enumerate(cols):
                e = flat[b * ncol + c]
                e.host, e.bytes, e.slice, e.cacheKey = hc.host_ptr, hc.nbytes, hc.slice, hc.cache_key
        stats = (C.c_uint64 * 4)()
        rc = _driver().AresQueryRunHostBatches(self._q, flat, ncol, sizes, len(batches), cache.handle if cache else None,
                                               stats, self._err, 1024)
        if rc != 0:
            raise abi.AresError(self._err.value.decode().strip())
        return {"uploaded_bytes": stats[0], "uploads": stats[1], "cache_hits": stats[2], "cache_bytes": stats[3]}

    def release(self):
        if self._q:
            _driver().AresQueryDestroy(self._q)
            self._q = None

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001
            pass


class NativeComm:
    """Communicator of libaresdriver.so's shard merge.  `rccl`: ranks on GPUs (the 128-byte id of rank
    0 reaches the others through `broadcast`, e.g. torch.distributed); `torch_group`: any
    torch.distributed group on host memory (gloo) — what the multi-process CPU tests use."""

    def __init__(self, handle, keep=None):
        self.handle, self._keep = handle, keep

    @classmethod
    def rccl(cls, rank, world, device, broadcast):
        lib = _driver()
        ident = (C.c_uint8 * 128)()
        err = C.create_string_buffer(512)
        failed = rank == 0 and lib.AresCommRcclUniqueId(ident, err, 512) != 0
        raw = broadcast(bytes(128) if failed else bytes(ident))  # every rank takes part, whatever rank 0 found
        if raw == bytes(128):
            raise abi.AresError("rank 0 could not obtain an RCCL id: " + err.value.decode())
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        h = lib.AresCommCreateRccl(ident, rank, world, device, err, 512)
        if not h:
            raise abi.AresError(err.value.decode())
        return cls(h)

    @classmethod
    def local(cls, nranks, device_memory):
        """nranks communicators for nranks threads of THIS process (the reference's process model: one device per
        shard inside the server, query/device_manager.go:185-218); blocks are copied peer to peer."""
        lib = _driver()
        lib.AresCommCreateLocal.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        lib.AresCommCreateLocal.restype = C.c_int
        out = (C.c_void_p * nranks)()
        err = C.create_string_buffer(512)
        if lib.AresCommCreateLocal(nranks, 1 if device_memory else 0, out, err, 512) != 0:
            raise abi.AresError(err.value.decode())
        return [cls(out[r]) for r in range(nranks)]

    @classmethod
    def torch_group(cls, group=None, all_to_all=False, device_backend=None, device=0):
        """`group`: a torch.distributed group for host tensors (gloo).  With `device_backend` (an abi.Backend
        whose memory is device memory) the buffers are staged through host memory with its copy entry
        points — the transport of last resort for ranks on GPUs when RCCL cannot be bound directly."""
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def all_gather(user, send, recv, nbytes, stream):
            try:
                if device_backend is not None:
                    mine = torch.empty(nbytes, dtype=torch.uint8)
                    device_backend.call("AsyncCopyDeviceToHost", mine.data_ptr(), send, nbytes, stream, device)
                    device_backend.call("WaitForCudaStream", stream, device)
                else:
                    mine = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
                out = torch.empty(world * nbytes, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, mine, group=group)
                if device_backend is not None:
                    device_backend.call("AsyncCopyHostToDevice", recv, out.data_ptr(), world * nbytes, stream, device)
                    device_backend.call("WaitForCudaStream", stream, device)
                else:
                    C.memmove(recv, out.data_ptr(), world * nbytes)
                return 0
            except Exception:  # noqa: BLE001
                return 1
        cb = ALLGATHER_FN(all_gather)
        handle = _driver().AresCommCreate(rank, world, cb, None)
        keep = [cb]
        if all_to_all:
            def exchange(user, send, send_bytes, send_off, recv, recv_bytes, recv_off, stream):
                try:
                    outs = [torch.frombuffer((C.c_uint8 * max(send_bytes[r], 1)).from_address(send + send_off[r]),
                                             dtype=torch.uint8)[:send_bytes[r]].clone() for r in range(world)]
                    ins = [torch.empty(recv_bytes[r], dtype=torch.uint8) for r in range(world)]
                    # gloo has no all_to_all: pairwise exchanges, lower rank sends first
                    for peer in range(world):
                        if peer == rank:
                            ins[peer].copy_(outs[peer])
                        elif rank < peer:
                            dist.send(outs[peer], peer, group=group)
                            dist.recv(ins[peer], peer, group=group)
                        else:
                            dist.recv(ins[peer], peer, group=group)
                            dist.send(outs[peer], peer, group=group)
                    for r in range(world):
                        if recv_bytes[r]:
                            C.memmove(recv + recv_off[r], ins[r].data_ptr(), recv_bytes[r])
                    return 0
                except Exception:  # noqa: BLE001
                    return 1
            a2a = ALLTOALL_FN(exchange)
            _driver().AresCommSetAllToAll(handle, a2a)
            keep.append(a2a)
        return cls(handle, keep=keep)

    def destroy(self):
        if self.handle:
            _driver().AresCommDestroy(self.handle)
            self.handle = None


class HostColumn:
    """One column of one batch in pinned host memory (HostAlloc), laid out like the device allocation
    the Go host uploads: [validity bitmap, 64-byte padded][values] (query/aql_processor.go:1415-1429)."""

    def __init__(self, be, data_type, values, valid=None, cache_key=0):
        import numpy as np
        values = np.ascontiguousarray(values)
        vbytes = values.view(np.uint8).reshape(-1)
        nb = np.zeros(0, np.uint8) if valid is None else np.packbits(np.asarray(valid, bool), bitorder="little")
        vo = (len(nb) + 63) // 64 * 64
        self.nbytes = vo + (len(vbytes) + 63) // 64 * 64
        self.be = be
        self.host_ptr = be.call("HostAlloc", self.nbytes)
        buf = np.frombuffer((C.c_uint8 * self.nbytes).from_address(self.host_ptr), np.uint8)
        buf[:len(nb)] = nb
        buf[vo:vo + len(vbytes)] = vbytes
        vp = abi.VectorPartySlice()
        vp.DataType, vp.Length, vp.StartingIndex = data_type, len(values), 0
        if valid is not None:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = 0, 0, vo
        else:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = vo, 0, 0  # BasePtr = byte offset inside the allocation
        self.slice, self.cache_key = vp, cache_key

    @classmethod
    def from_blob(cls, be, data_type, blob, values_off, length, has_nulls, cache_key=0):
        """From an allocation image that already has the upload layout (aresdb_amd.workload.ResidentColumn)."""
        import numpy as np
        self = cls.__new__(cls)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.be, self.nbytes = be, len(blob)
        self.host_ptr = be.call("HostAlloc", self.nbytes)
        np.frombuffer((C.c_uint8 * self.nbytes).from_address(self.host_ptr), np.uint8)[:] = blob
        vp = abi.VectorPartySlice()
        vp.DataType, vp.Length, vp.StartingIndex = data_type, length, 0
        if has_nulls:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = 0, 0, values_off
        else:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = values_off, 0, 0
        self.slice, self.cache_key = vp, cache_key
        return self

    def free(self):
        if self.host_ptr:
            self.be.call("HostFree", self.host_ptr)
            self.host_ptr = 0


class ColumnCache:
    """Device-resident column cache of libaresdriver.so (least recently used out, byte budget)."""

    def __init__(self, be, device=0, budget_bytes=1 << 30):
        self.handle = _driver().AresColumnCacheCreate(_open(be), device, budget_bytes)

    def destroy(self):
        if self.handle:
            _driver().AresColumnCacheDestroy(self.handle)
            self.handle = None
