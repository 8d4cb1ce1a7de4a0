"""Host side of the AQL batch pipeline, above the C ABI.

A Python mirror of the reference's Go batch executor — same stage names, same call order, same
buffer ownership — so that a query replayed here issues exactly the ABI calls the Go host would:

  oopkBatchContext        query/aql_context.go + query/aql_processor.go:690-804
  BatchExecutorImpl       query/aql_batchexecutor.go:103-273   (preExec/filter/join/project/reduce/postExec)
  processExpression       query/time_series_aggregate.go:491-593 (AST walk, one ABI call per node)
  GetDimensionStartOffsets query/common/dim_util.go

The Go toolchain is not available in this environment, so this module — not cgo —
drives the libraries in tests and benchmarks.  It is backend-agnostic by construction (it only
speaks the ABI of aresdb_amd.abi.Backend); the product entry points always bind the HIP backend.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import abi

DIM_WIDTHS = (16, 8, 4, 2, 1)


# ---- expressions (the subset of query/expr that reaches the ABI) ----------------------------------
@dataclass
class Col:
    name: str
    table: int = 0  # 0 = main table, k>0 = foreign table k-1


@dataclass
class Const:
    value: object  # int or float


@dataclass
class Unary:
    op: int  # abi.UnaryFunctorType
    arg: object
    out_type: int = abi.Int32  # scratch data type when used as an inner node


@dataclass
class Binary:
    op: int  # abi.BinaryFunctorType
    lhs: object
    rhs: object
    out_type: int = abi.Int32


@dataclass
class DimensionSpec:
    expr: object
    data_type: int = abi.Uint32  # type of the dimension output vector

    @property
    def width(self):
        return abi.DATA_TYPE_BYTES[self.data_type]


@dataclass
class ForeignTable:
    """Dimension table joined on a main-table column (query/aql_processor.go:398-457)."""
    join_column: str
    index: abi.CuckooHashIndex
    batches: Dict[str, Sequence[abi.VectorPartySlice]]  # column name -> one slice per batch
    data_types: Dict[str, int]
    base_batch_id: int
    num_records_in_last_batch: int


@dataclass
class GeoIntersection:
    """geoIntersection (query/aql_context.go:327-353) after the processor uploaded the shapes."""
    shape_lat_longs: int        # device pointer: [lats f32][longs f32][shape index u8]
    num_shapes: int
    total_num_points: int
    point_column: str
    point_table: int = 0        # 0 = main table, k > 0 = foreign table k-1
    in_or_out: bool = True
    dim_index: int = -1         # query dimension that is the shape number, < 0: filter only


@dataclass
class QueryPlan:
    filters: List[object]
    dimensions: List[DimensionSpec]
    measure: object
    agg: int
    measure_type: int  # data type of the measure output vector
    use_hash_reduction: bool = False
    use_fused_extension: bool = False  # C++ driver only: one fused call per batch where the plan allows
    foreign_tables: List[ForeignTable] = field(default_factory=list)
    foreign_filters: List[object] = field(default_factory=list)
    geo: Optional[GeoIntersection] = None

    @property
    def measure_bytes(self):
        return abi.DATA_TYPE_BYTES[self.measure_type]

    @property
    def is_hll(self):
        """OOPKContext.IsHLL (query/aql_context.go:421-424)."""
        return self.agg == abi.AGGR_HLL

    def num_dims_per_width(self):
        return tuple(sum(1 for d in self.dimensions if d.width == w) for w in DIM_WIDTHS)

    def dim_vector_index(self):
        """Position of each query dimension inside the width-ordered dimension vector
        (query/aql_compiler.go:1341-1359: by width 16..1, then query order)."""
        order = sorted(range(len(self.dimensions)), key=lambda i: (-self.dimensions[i].width, i))
        pos = [0] * len(self.dimensions)
        for p, i in enumerate(order):
            pos[i] = p
        return pos

    @property
    def dim_row_bytes(self):
        return sum(d.width for d in self.dimensions) + len(self.dimensions)


def dimension_start_offsets(ndw, dim_index, capacity):
    """(value offset, validity offset) of dimension `dim_index` — query/common/dim_util.go."""
    widths = [w for w, c in zip(DIM_WIDTHS, ndw) for _ in range(c)]
    value_off = sum(widths[:dim_index]) * capacity
    null_off = sum(widths) * capacity + dim_index * capacity
    return value_off, null_off


def column_input(vp: abi.VectorPartySlice) -> abi.InputVector:
    iv = abi.InputVector()
    iv.Vector.VP = vp
    iv.Type = abi.VectorPartyInput
    return iv


def constant_input(value) -> abi.InputVector:
    iv = abi.InputVector()
    if isinstance(value, float):
        iv.Vector.Constant.Value.FloatVal = value
        iv.Vector.Constant.DataType = abi.ConstFloat
    else:
        iv.Vector.Constant.Value.IntVal = int(value)
        iv.Vector.Constant.DataType = abi.ConstInt
    iv.Vector.Constant.IsValid = True
    iv.Type = abi.ConstantInput
    return iv


class BatchContext:
    """oopkBatchContext: per-query device state that survives across batches."""

    def __init__(self, be: abi.Backend, plan: QueryPlan, device=0, stream=None):
        self.be, self.plan, self.device, self.stream = be, plan, device, stream
        self.ndw = plan.num_dims_per_width()
        self.dim_index = plan.dim_vector_index()
        self.result_size = 0
        self.result_capacity = 0
        self.dim_vec = [0, 0]
        self.measure_vec = [0, 0]
        self.hash_vec = [0, 0]
        self.dim_index_vec = [0, 0]
        self.size = 0
        self.index_vec = 0
        self.pred_vec = 0
        self.base_counts = None
        self.start_row = 0
        self.columns: Dict[str, abi.VectorPartySlice] = {}
        self.stack: List[int] = []
        self.foreign_rids: List[int] = []
        self.geo_predicate_vec = 0
        self.owned_columns = []  # callables that free the batch's device columns
        self._keepalive = []
        self.calls = 0  # ABI calls issued (for tests / stats)
        # HyperLogLog queries: buffers the library allocates on the last batch
        # (query/aql_context.go:289-294), released with DeviceFree like every other buffer
        self.hll_vector = 0
        self.hll_dim_reg_count = 0
        self.hll_vector_size = 0

    # -- allocation helpers (device_allocator.go semantics: every byte is tracked and freed) --------
    def _alloc(self, nbytes):
        return self.be.device_alloc(nbytes, self.device)

    def _free(self, ptr):
        if ptr:
            self.be.device_free(ptr, self.device)

    def call(self, sym, *args):
        self.calls += 1
        return self.be.call(sym, *args)

    # -- aql_processor.go:726-739 --------------------------------------------------------------------
    def prepare_for_filtering(self, columns, size, base_counts=None, start_row=0):
        self.columns = columns
        self.size = size
        self.start_row = start_row
        self.base_counts = base_counts
        self.index_vec = self._alloc(size * 4)
        self.pred_vec = self._alloc(size)

    # -- aql_processor.go:743-804 --------------------------------------------------------------------
    def prepare_for_dim_and_measure_eval(self):
        plan = self.plan
        if self.result_size + self.size <= self.result_capacity:
            return
        old_capacity = self.result_capacity
        self.result_capacity = self.result_size + self.size
        self.result_capacity += self.result_capacity // 8
        cap = self.result_capacity

        def realloc(pair, unit, copy):
            old = list(pair)
            pair[0] = self._alloc(cap * unit)
            pair[1] = self._alloc(cap * unit)
            if copy and old[0]:
                copy(pair[0], old[0])
            if old[0] or old[1]:
                self.be.wait(self.stream, self.device)
            self._free(old[0])
            self._free(old[1])

        def copy_dims(to, frm):  # asyncCopyDimensionVector: per-dim strided D2D copies
            widths = [w for w, c in zip(DIM_WIDTHS, self.ndw) for _ in range(c)]
            for d, w in enumerate(widths):
                nv, _ = dimension_start_offsets(self.ndw, d, cap)
                ov, _ = dimension_start_offsets(self.ndw, d, old_capacity)
                if self.result_size:
                    self.be.call("AsyncCopyDeviceToDevice", to + nv, frm + ov, self.result_size * w,
                                 self.stream, self.device)
            for d in range(len(widths)):
                _, nn = dimension_start_offsets(self.ndw, d, cap)
                _, on = dimension_start_offsets(self.ndw, d, old_capacity)
                if self.result_size:
                    self.be.call("AsyncCopyDeviceToDevice", to + nn, frm + on, self.result_size,
                                 self.stream, self.device)

        realloc(self.dim_vec, max(plan.dim_row_bytes, 1), copy_dims)
        if plan.is_hll or not plan.use_hash_reduction:
            realloc(self.dim_index_vec, 4, None)
        if plan.is_hll:  # the merged keys of the earlier batches live in hash vector [0]
            realloc(self.hash_vec, 8, lambda to, frm: self.result_size and self.be.call(
                "AsyncCopyDeviceToDevice", to, frm, self.result_size * 8, self.stream, self.device))
        elif not plan.use_hash_reduction:
            realloc(self.hash_vec, 8, None)
        mb = plan.measure_bytes
        realloc(self.measure_vec, mb, lambda to, frm: self.result_size and self.be.call(
            "AsyncCopyDeviceToDevice", to, frm, self.result_size * mb, self.stream, self.device))

    # -- time_series_aggregate.go:745-769 ------------------------------------------------------------
    def allocate_stack_frame(self, data_type):
        w = 8 if data_type in (abi.Int64, abi.Uint64, abi.Float64, abi.GeoPoint) else \
            16 if data_type == abi.UUID else 4
        values = self._alloc((w + 1) * self.size)
        self.stack.append(values)
        return values, w * self.size

    def shrink_stack_frame(self):
        self.stack[-1], self.stack[-2] = self.stack[-2], self.stack[-1]
        self._free(self.stack.pop())

    def cleanup_before_aggregation(self):
        # the batch's input columns go first (query/aql_processor.go:695-699)
        for release in self.owned_columns:
            release()
        self.owned_columns = []
        self._free(self.index_vec)
        self._free(self.pred_vec)
        self.index_vec = self.pred_vec = 0
        for p in self.foreign_rids:
            self._free(p)
        self.foreign_rids = []
        self._free(self.geo_predicate_vec)
        self.geo_predicate_vec = 0
        for p in self.stack:
            self._free(p)
        self.stack = []
        self._keepalive = []

    def swap_result_buffers(self):
        self.size = 0
        for pair in (self.dim_vec, self.measure_vec, self.hash_vec):
            pair[0], pair[1] = pair[1], pair[0]

    def release(self):
        for pair in (self.dim_vec, self.measure_vec, self.hash_vec, self.dim_index_vec):
            for i in (0, 1):
                self._free(pair[i])
                pair[i] = 0
        self.result_capacity = 0
        self._free(self.hll_vector)
        self._free(self.hll_dim_reg_count)
        self.hll_vector = self.hll_dim_reg_count = 0

    # -- processExpression (time_series_aggregate.go:491-593) ---------------------------------------
    def _foreign_input(self, e: Col):
        ft = self.plan.foreign_tables[e.table - 1]
        slices = ft.batches[e.name]
        arr = (abi.VectorPartySlice * len(slices))(*slices)
        self._keepalive.append(arr)
        iv = abi.InputVector()
        f = iv.Vector.ForeignVP
        f.RecordIDs = self.foreign_rids[e.table - 1]
        f.Batches = C.addressof(arr)
        f.BaseBatchID = ft.base_batch_id
        f.NumBatches = len(slices)
        f.NumRecordsInLastBatch = ft.num_records_in_last_batch
        f.TimezoneLookup, f.TimezoneLookupSize = None, 0
        f.DataType = ft.data_types[e.name]
        iv.Type = abi.ForeignColumnInput
        return iv

    def process_expression(self, e, action):
        if isinstance(e, Col):
            iv = column_input(self.columns[e.name]) if e.table == 0 else self._foreign_input(e)
            if action:
                action(abi.Noop, [iv])
                return None
            return iv
        if isinstance(e, Const):
            iv = constant_input(e.value)
            if action:
                action(abi.Noop, [iv])
                return None
            return iv
        if isinstance(e, Unary):
            iv = self.process_expression(e.arg, None)
            if action:
                action(e.op, [iv])
                return None
            values, nulls_off = self.allocate_stack_frame(e.out_type)
            ov = abi.OutputVector()
            ov.Vector.ScratchSpace.Values = values
            ov.Vector.ScratchSpace.NullsOffset = nulls_off
            ov.Vector.ScratchSpace.DataType = e.out_type
            ov.Type = abi.ScratchSpaceOutput
            self.call("UnaryTransform", iv, ov, self.index_vec, self.size, self.base_counts,
                      self.start_row, e.op, self.stream, self.device)
            if iv.Type == abi.ScratchSpaceInput:
                self.shrink_stack_frame()
            return self._scratch_input(values, nulls_off, e.out_type)
        if isinstance(e, Binary):
            lhs = self.process_expression(e.lhs, None)
            rhs = self.process_expression(e.rhs, None)
            if action:
                action(e.op, [lhs, rhs])
                return None
            values, nulls_off = self.allocate_stack_frame(e.out_type)
            ov = abi.OutputVector()
            ov.Vector.ScratchSpace.Values = values
            ov.Vector.ScratchSpace.NullsOffset = nulls_off
            ov.Vector.ScratchSpace.DataType = e.out_type
            ov.Type = abi.ScratchSpaceOutput
            self.call("BinaryTransform", lhs, rhs, ov, self.index_vec, self.size, self.base_counts,
                      self.start_row, e.op, self.stream, self.device)
            if rhs.Type == abi.ScratchSpaceInput:
                self.shrink_stack_frame()
            if lhs.Type == abi.ScratchSpaceInput:
                self.shrink_stack_frame()
            return self._scratch_input(values, nulls_off, e.out_type)
        raise TypeError(f"unsupported expression node {e!r}")

    @staticmethod
    def _scratch_input(values, nulls_off, data_type):
        iv = abi.InputVector()
        iv.Vector.ScratchSpace.Values = values
        iv.Vector.ScratchSpace.NullsOffset = nulls_off
        iv.Vector.ScratchSpace.DataType = data_type
        iv.Type = abi.ScratchSpaceInput
        return iv

    # -- root actions (time_series_aggregate.go:369-459) -----------------------------------------------
    def filter_action(self, functor, inputs):
        if self.size <= 0:
            return
        nf = len(self.foreign_rids)
        vecs = (C.c_void_p * max(nf, 1))(*self.foreign_rids) if nf else None
        args = [self.index_vec, self.pred_vec, self.size, C.addressof(vecs) if nf else None, nf,
                self.base_counts, self.start_row, functor, self.stream, self.device]
        name = "UnaryFilter" if len(inputs) == 1 else "BinaryFilter"
        self.size = self.call(name, *inputs, *args)

    def measure_action(self, functor, inputs):
        if self.size <= 0:
            return
        plan = self.plan
        ov = abi.OutputVector()
        ov.Vector.Measure.Values = self.measure_vec[0] + self.result_size * plan.measure_bytes
        if plan.is_hll:  # hll values of the batch go to measure vector [1] (time_series_aggregate.go:404-408)
            ov.Vector.Measure.Values = self.measure_vec[1]
        ov.Vector.Measure.DataType = plan.measure_type
        ov.Vector.Measure.AggFunc = plan.agg
        ov.Type = abi.MeasureOutput
        name = "UnaryTransform" if len(inputs) == 1 else "BinaryTransform"
        self.call(name, *inputs, ov, self.index_vec, self.size, self.base_counts, self.start_row, functor,
                  self.stream, self.device)

    def make_dimension_action(self, dim: DimensionSpec, value_off, null_off, prev_result_size):
        def action(functor, inputs):
            if self.size <= 0:
                return
            ov = abi.OutputVector()
            ov.Vector.Dimension.DimValues = self.dim_vec[0] + value_off + dim.width * prev_result_size
            ov.Vector.Dimension.DimNulls = self.dim_vec[0] + null_off + prev_result_size
            ov.Vector.Dimension.DataType = dim.data_type
            ov.Type = abi.DimensionOutput
            name = "UnaryTransform" if len(inputs) == 1 else "BinaryTransform"
            self.call(name, *inputs, ov, self.index_vec, self.size, self.base_counts, self.start_row, functor,
                      self.stream, self.device)
        return action

    # -- geo (time_series_aggregate.go:596-660) ------------------------------------------------------------
    def geo_intersect(self):
        geo = self.plan.geo
        if self.size <= 0 or not geo.shape_lat_longs:
            return
        shapes = abi.GeoShapeBatch()
        shapes.LatLongs, shapes.TotalNumPoints = geo.shape_lat_longs, geo.total_num_points
        shapes.TotalWords = (geo.num_shapes + 31) // 32
        points = column_input(self.columns[geo.point_column]) if geo.point_table == 0 else \
            self._foreign_input(Col(geo.point_column, geo.point_table))
        nf = len(self.foreign_rids)
        vecs = (C.c_void_p * max(nf, 1))(*self.foreign_rids) if nf else None
        self.size = self.call("GeoBatchIntersects", shapes, points, self.index_vec, self.size, self.start_row,
                              C.addressof(vecs) if nf else None, nf, self.geo_predicate_vec, geo.in_or_out,
                              self.stream, self.device)

    def write_geo_shape_dim(self, value_off, null_off, size_before_geo, prev_result_size):
        geo = self.plan.geo
        if self.size <= 0 or not geo.shape_lat_longs:
            return
        dv = abi.DimensionOutputVector()
        dv.DimValues = self.dim_vec[0] + value_off + prev_result_size
        dv.DimNulls = self.dim_vec[0] + null_off + prev_result_size
        dv.DataType = abi.Uint8
        self.call("WriteGeoShapeDim", (geo.num_shapes + 31) // 32, dv, size_before_geo, self.geo_predicate_vec,
                  self.stream, self.device)

    def dimension_vector(self, which):
        dv = abi.DimensionVector()
        dv.DimValues = self.dim_vec[which]
        dv.HashValues = self.hash_vec[which] or None
        dv.IndexVector = self.dim_index_vec[which] or None
        dv.VectorCapacity = self.result_capacity
        for i, c in enumerate(self.ndw):
            dv.NumDimsPerDimWidth[i] = c
        return dv


class BatchExecutor:
    """BatchExecutorImpl.Run: preExec, filter, join, project, reduce, postExec."""

    def __init__(self, ctx: BatchContext):
        self.ctx = ctx
        self.is_last_batch = False
        self.size_before_geo = 0

    def run(self, columns: Dict[str, abi.VectorPartySlice], size: int, base_counts=None, start_row=0,
            is_last_batch=False, owned_columns=()):
        """owned_columns: callables releasing the batch's device columns; the Go host frees them in
        cleanupBeforeAggregation, i.e. between project() and reduce()."""
        c = self.ctx
        c.owned_columns = list(owned_columns)
        self.is_last_batch = is_last_batch
        c.prepare_for_filtering(columns, size, base_counts, start_row)
        self.pre_exec()
        self.filter()
        self.join()
        self.project()
        self.reduce()
        self.post_exec()

    def pre_exec(self):
        c = self.ctx
        if c.index_vec and c.size > 0:
            c.call("InitIndexVector", c.index_vec, 0, c.size, c.stream, c.device)

    def filter(self):
        c = self.ctx
        for f in c.plan.filters:
            c.process_expression(f, c.filter_action)

    def join(self):
        c = self.ctx
        for ft in c.plan.foreign_tables:
            rids = c._alloc(8 * max(c.size, 1))
            c.foreign_rids.append(rids)
            if c.size > 0:
                c.call("HashLookup", column_input(c.columns[ft.join_column]), rids, c.index_vec, c.size,
                       c.base_counts, c.start_row, ft.index, c.stream, c.device)
        for f in c.plan.foreign_filters:
            c.process_expression(f, c.filter_action)
        if c.plan.geo is not None:  # query/aql_batchexecutor.go:146-165
            words = (c.plan.geo.num_shapes + 31) // 32
            c.geo_predicate_vec = c._alloc(max(c.size, 1) * 4 * words)
        self.size_before_geo = c.size
        if c.plan.geo is not None:
            c.geo_intersect()

    def project(self):
        c = self.ctx
        c.prepare_for_dim_and_measure_eval()
        prev = c.result_size
        for i, dim in enumerate(c.plan.dimensions):
            vo, no = dimension_start_offsets(c.ndw, c.dim_index[i], c.result_capacity)
            if c.plan.geo is not None and c.plan.geo.dim_index == i:
                c.write_geo_shape_dim(vo, no, self.size_before_geo, prev)
                continue
            c.process_expression(dim.expr, c.make_dimension_action(dim, vo, no, prev))
        c.process_expression(c.plan.measure, c.measure_action)
        c.be.wait(c.stream, c.device)
        c.cleanup_before_aggregation()

    def reduce(self):
        c = self.ctx
        plan = c.plan
        length = c.result_size + c.size
        if plan.is_hll:  # query/aql_batchexecutor.go:221-233, query/time_series_aggregate.go:661-680
            c.call("InitIndexVector", c.dim_index_vec[0], 0, c.result_size, c.stream, c.device)
            c.call("InitIndexVector", c.dim_index_vec[1], c.result_size, length, c.stream, c.device)
            vec, size, counts = C.c_void_p(0), C.c_size_t(0), C.c_void_p(0)
            c.result_size = c.call("HyperLogLog", c.dimension_vector(0), c.dimension_vector(1), c.measure_vec[0],
                                   c.measure_vec[1], c.result_size, c.size, bool(self.is_last_batch),
                                   C.addressof(vec), C.addressof(size), C.addressof(counts), c.stream, c.device)
            if vec.value or counts.value:
                c._free(c.hll_vector)
                c._free(c.hll_dim_reg_count)
                c.hll_vector, c.hll_dim_reg_count, c.hll_vector_size = vec.value or 0, counts.value or 0, size.value
        elif plan.use_hash_reduction:
            c.result_size = c.call("HashReduce", c.dimension_vector(0), c.measure_vec[0], c.dimension_vector(1),
                                   c.measure_vec[1], plan.measure_bytes, length, plan.agg, c.stream, c.device)
        else:
            c.call("InitIndexVector", c.dim_index_vec[0], 0, length, c.stream, c.device)
            c.call("Sort", c.dimension_vector(0), length, c.stream, c.device)
            c.result_size = c.call("Reduce", c.dimension_vector(0), c.measure_vec[0], c.dimension_vector(1),
                                   c.measure_vec[1], plan.measure_bytes, length, plan.agg, c.stream, c.device)
        c.be.wait(c.stream, c.device)

    def post_exec(self):
        self.ctx.swap_result_buffers()


def fetch_results(ctx: BatchContext):
    """D2H of the final result (query/aql_processor.go:641-671, :145-154): per-dimension value
    arrays + validity arrays + the measure vector, in query dimension order."""
    be, plan, n = ctx.be, ctx.plan, ctx.result_size
    dims, valids = [], []
    for i, dim in enumerate(plan.dimensions):
        vo, no = dimension_start_offsets(ctx.ndw, ctx.dim_index[i], ctx.result_capacity)
        v = np.empty(n * dim.width, np.uint8)
        m = np.empty(n, np.uint8)
        if n:
            be.d2h(v.ctypes.data_as(C.c_void_p), ctx.dim_vec[0] + vo, v.nbytes, ctx.stream, ctx.device)
            be.d2h(m.ctypes.data_as(C.c_void_p), ctx.dim_vec[0] + no, m.nbytes, ctx.stream, ctx.device)
        dims.append(v)
        valids.append(m)
    meas = np.empty(n * plan.measure_bytes, np.uint8)
    if n:
        be.d2h(meas.ctypes.data_as(C.c_void_p), ctx.measure_vec[0], meas.nbytes, ctx.stream, ctx.device)
    be.wait(ctx.stream, ctx.device)
    return dims, valids, meas


def fetch_hll_results(ctx: BatchContext):
    """What SerializeHLL copies to the host (query/hll.go:52-63): the dimension columns, the
    registers-per-dimension counts (uint16) and the encoded HLL vector."""
    be, plan, n = ctx.be, ctx.plan, ctx.result_size
    dims, valids = [], []
    for i, dim in enumerate(plan.dimensions):
        vo, no = dimension_start_offsets(ctx.ndw, ctx.dim_index[i], ctx.result_capacity)
        v = np.empty(n * dim.width, np.uint8)
        m = np.empty(n, np.uint8)
        if n:
            be.d2h(v.ctypes.data_as(C.c_void_p), ctx.dim_vec[0] + vo, v.nbytes, ctx.stream, ctx.device)
            be.d2h(m.ctypes.data_as(C.c_void_p), ctx.dim_vec[0] + no, m.nbytes, ctx.stream, ctx.device)
        dims.append(v)
        valids.append(m)
    counts = np.empty(n, np.uint16)
    vec = np.empty(ctx.hll_vector_size if n else 0, np.uint8)
    if n:
        be.d2h(counts.ctypes.data_as(C.c_void_p), ctx.hll_dim_reg_count, counts.nbytes, ctx.stream, ctx.device)
        be.d2h(vec.ctypes.data_as(C.c_void_p), ctx.hll_vector, vec.nbytes, ctx.stream, ctx.device)
    be.wait(ctx.stream, ctx.device)
    return dims, valids, counts, vec
