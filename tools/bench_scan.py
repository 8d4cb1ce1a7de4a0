"""Filter / transform kernels alone on a 100 M-row resident column (BASELINE config C2 shape):
per-kernel HIP-event time and the HBM traffic each call must move (development aid)."""
import json, os, sys
os.environ.setdefault("ARES_FUSE", "0")  # time every transform launch on its own (no second-stage fusion)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aresdb_amd import abi, workload
from aresdb_amd.executor import column_input, constant_input


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    be = abi.load_hip_backend(); be.call("BootstrapDevice")
    dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(1)
    for nullf in (0.0, 0.01):
        b = workload.c3_batch(n, g, dev, null_fraction=nullf)
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        pred = torch.empty(n, dtype=torch.uint8, device=dev)
        dimv = torch.empty(n, dtype=torch.int32, device=dev); dimn = torch.empty(n, dtype=torch.uint8, device=dev)
        meas = torch.empty(n, dtype=torch.float64, device=dev)
        def timed(fn, reps=5):
            best = {}
            for _ in range(reps):
                be.profiler_enable(True); r = fn(); be.wait(); rep = be.profiler_report(); be.profiler_enable(False)
                for k, (c, ms) in rep.items():
                    best[k] = min(best.get(k, 1e9), ms / c)
            return r, best
        for sel, thr in ((1.0, 1000), (0.9, 90), (0.5, 50), (0.1, 10)):
            def f():
                be.call("InitIndexVector", idx.data_ptr(), 0, n, None, 0)
                return be.call("BinaryFilter", column_input(b["d1"].vp), constant_input(thr), idx.data_ptr(), pred.data_ptr(), n,
                               None, 0, None, 0, abi.LessThan, None, 0)
            cnt, k = timed(f)
            fbytes = n * (4 + 4 + (0.125 if nullf else 0) + 1) + cnt * 4
            ms = sum(v for name, v in k.items() if name.startswith("filter_"))
            print(json.dumps({"op": "filter", "nulls": nullf, "sel": sel, "rows": n, "survivors": cnt, "ms": round(ms, 4),
                              "traffic_GBps": round(fbytes / ms / 1e6, 1), "rows_per_s_G": round(n / ms / 1e6, 1),
                              "init_ms": round(k.get("init_index_kernel", 0), 4),
                              "parts": {name: round(v, 4) for name, v in k.items() if name.startswith("filter_")}}), flush=True)
            ov = abi.OutputVector(); ov.Vector.Dimension.DimValues, ov.Vector.Dimension.DimNulls = dimv.data_ptr(), dimn.data_ptr()
            ov.Vector.Dimension.DataType = abi.Uint32; ov.Type = abi.DimensionOutput
            _, k = timed(lambda: be.call("BinaryTransform", column_input(b["ts"].vp), constant_input(3600), ov, idx.data_ptr(), cnt,
                                         None, 0, abi.Floor, None, 0))
            ms = k["transform_fast_kernel"]; tb = cnt * (4 + 4 + 5)
            print(json.dumps({"op": "floor->dim", "nulls": nullf, "sel": sel, "rows": cnt, "ms": round(ms, 4),
                              "traffic_GBps": round(tb / ms / 1e6, 1), "rows_per_s_G": round(cnt / ms / 1e6, 1)}), flush=True)
            _, k = timed(lambda: be.call("UnaryTransform", column_input(b["d2"].vp), ov, idx.data_ptr(), cnt,
                                         None, 0, abi.Noop, None, 0))
            ms = k["transform_fast_kernel"]
            print(json.dumps({"op": "noop->dim", "nulls": nullf, "sel": sel, "rows": cnt, "ms": round(ms, 4),
                              "traffic_GBps": round(tb / ms / 1e6, 1), "rows_per_s_G": round(cnt / ms / 1e6, 1)}), flush=True)
            mo = abi.OutputVector(); mo.Vector.Measure.Values, mo.Vector.Measure.DataType, mo.Vector.Measure.AggFunc = meas.data_ptr(), abi.Float64, abi.AGGR_SUM_FLOAT
            mo.Type = abi.MeasureOutput
            _, k = timed(lambda: be.call("UnaryTransform", column_input(b["m"].vp), mo, idx.data_ptr(), cnt, None, 0, abi.Noop, None, 0))
            ms = k["transform_fast_kernel"]; tb = cnt * (4 + 4 + 8)
            print(json.dumps({"op": "noop->measure f64", "nulls": nullf, "sel": sel, "rows": cnt, "ms": round(ms, 4),
                              "traffic_GBps": round(tb / ms / 1e6, 1), "rows_per_s_G": round(cnt / ms / 1e6, 1)}), flush=True)
        del b
main()
