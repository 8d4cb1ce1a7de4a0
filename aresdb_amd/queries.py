"""Query plans of the BASELINE configurations, as the AQL compiler would
hand them to the batch executor."""
from . import abi
from .executor import Binary, Col, Const, DimensionSpec, QueryPlan


def c3_plan(use_hash_reduction=True, with_filter=True, dims=("ts", "d1", "d2", "d3"), d1_below=90, ts_range=None, sort_measure=None,
            eight_dims=False):
    """BASELINE config C3; `dims` selects a subset of its four group-by dimensions (lower-cardinality variants of the
    same query: the filter and the measure stay), `d1_below` the filter constant.  ts_range = (from, to): the two time
    filters the Go host puts in front of every fact-table query's own filters — ts >= from, ts < to
    (query/aql_processor.go:543-559, query/common/time_filter.go:371-397).  sort_measure: the same group-by through the
    reference's DEFAULT path, Sort + Reduce (config/ares.yaml:11 enable_hash_reduction: false; query/aql_context.go:426-434):
    "count" = COUNT(*), a column name = SUM over that unsigned column."""
    specs = {"ts": DimensionSpec(Binary(abi.Floor, Col("ts"), Const(3600)), abi.Uint32),
             "d1": DimensionSpec(Col("d1"), abi.Uint32), "d2": DimensionSpec(Col("d2"), abi.Uint32),
             "d3": DimensionSpec(Col("d3"), abi.Uint32)}
    time_filters = [] if ts_range is None else [Binary(abi.GreaterThanOrEqual, Col("ts"), Const(int(ts_range[0]))),
                                                Binary(abi.LessThan, Col("ts"), Const(int(ts_range[1])))]
    filters = time_filters + ([Binary(abi.LessThan, Col("d1"), Const(d1_below))] if with_filter else [])
    dimensions = [specs[d] for d in dims]
    if eight_dims:  # MAX_DIMENSIONS (query/time_series_aggregate.h:36-37): four more, functions of the first four — the same groups
        dimensions = dimensions + [DimensionSpec(Binary(abi.Plus, Col("d1"), Const(5)), abi.Uint32),
                                   DimensionSpec(Binary(abi.Multiply, Col("d2"), Const(3)), abi.Uint32),
                                   DimensionSpec(Binary(abi.Plus, Col("d3"), Const(1)), abi.Uint32),
                                   DimensionSpec(Binary(abi.Mod, Col("d2"), Const(7)), abi.Uint32)]
    if sort_measure == "count":  # COUNT(*): SUM_UNSIGNED over the literal 1 into 4 bytes, never hash-reduced (aql_compiler.go:1191-1197)
        return QueryPlan(filters=filters, dimensions=dimensions, measure=Const(1), agg=abi.AGGR_SUM_UNSIGNED,
                         measure_type=abi.Uint32, use_hash_reduction=False)
    if sort_measure:  # SUM over an unsigned column: 8 bytes, output type Int64 (time_series_aggregate.go:352-361); Sort + Reduce
        return QueryPlan(filters=filters, dimensions=dimensions, measure=Col(sort_measure), agg=abi.AGGR_SUM_UNSIGNED,
                         measure_type=abi.Int64, use_hash_reduction=False)
    return QueryPlan(
        filters=filters,
        dimensions=dimensions,
        measure=Col("m"), agg=abi.AGGR_SUM_FLOAT, measure_type=abi.Float64,
        use_hash_reduction=use_hash_reduction)


def c2_plan(threshold):
    """BASELINE config C2: single uint32 predicate + COUNT(*) — measure literal 1, AGGR_SUM_UNSIGNED,
    4-byte measure, no dimensions, sort path (query/aql_compiler.go:1191-1197)."""
    return QueryPlan(filters=[Binary(abi.LessThan, Col("ts"), Const(threshold))], dimensions=[],
                     measure=Const(1), agg=abi.AGGR_SUM_UNSIGNED, measure_type=abi.Uint32,
                     use_hash_reduction=False)
