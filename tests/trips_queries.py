"""BASELINE config C1: the two example queries of examples/1k_trips as the AQL compiler hands them to
the batch executor (examples/1k_trips/queries/total_trips.aql, total_fare.aql):

  rowFilter  status = 'completed'          -> Equal(status: SmallEnum/Uint8, enum id)
  timeFilter request_at in [from, to)      -> GreaterThanOrEqual / LessThan on the time column
                                              (query/aql_processor.go:553-559, common/time_filter.go)
  dimension  request_at, bucket "hour"     -> Floor(request_at, 3600) (query/time_bucketizer.go:157-173)
  measure    count(*) -> literal 1, AGGR_SUM_UNSIGNED, Uint32; sum(fare) -> AGGR_SUM_FLOAT, Float64
  hash reduction is off by default (config/ares.yaml:11): Sort + Reduce.
"""
import numpy as np

from aresdb_amd import abi
from aresdb_amd.columns import DeviceColumn
from aresdb_amd.driver import NativeQuery
from aresdb_amd.executor import Binary, Col, Const, DimensionSpec, QueryPlan


def trips_plans(data):
    now = data["now"]
    filters = [Binary(abi.Equal, Col("status"), Const(data["enum_status"]["completed"])),
               Binary(abi.GreaterThanOrEqual, Col("request_at"), Const(now - 86400)),
               Binary(abi.LessThan, Col("request_at"), Const(now))]
    dims = [DimensionSpec(Binary(abi.Floor, Col("request_at"), Const(3600)), abi.Uint32)]
    return {
        "total_trips": QueryPlan(filters=filters, dimensions=dims, measure=Const(1), agg=abi.AGGR_SUM_UNSIGNED,
                                 measure_type=abi.Uint32, use_hash_reduction=False),
        "total_fare": QueryPlan(filters=filters, dimensions=dims, measure=Col("fare"), agg=abi.AGGR_SUM_FLOAT,
                                measure_type=abi.Float64, use_hash_reduction=False),
    }


def run_trips_query(be, plan, data, batches=(400, 600)):
    """Runs the plan over the fixture split into live batches; returns {hour bucket: value}."""
    cols = {"request_at": (abi.Uint32, np.array(data["request_at"], np.uint32)),
            "status": (abi.Uint8, np.array(data["status"], np.uint8)),
            "fare": (abi.Float32, np.array(data["fare_f32_bits"], np.uint32).view(np.float32))}
    q = NativeQuery(be, plan, list(cols))
    start = 0
    for n in batches:
        dev = {k: DeviceColumn(be, t, v[start:start + n]) for k, (t, v) in cols.items()}
        q.run({k: d.vp for k, d in dev.items()}, n)
        for d in dev.values():
            d.free()
        start += n
    dims, valids, meas = q.fetch()
    n = q.result_size
    keys = dims[0].view(np.uint32)
    vals = meas.view(np.float64 if plan.measure_type == abi.Float64 else np.uint32)
    assert all(valids[0][:n])
    q.release()
    return {int(k): (float(v) if plan.measure_type == abi.Float64 else int(v)) for k, v in zip(keys[:n], vals[:n])}
