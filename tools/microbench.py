"""Per-entry-point timings of the HIP library on synthetic device-resident data (development aid;
bench.py is the contract benchmark).  Usage: python tools/microbench.py [rows]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aresdb_amd import abi  # noqa: E402
from aresdb_amd.columns import slice_from_pointer  # noqa: E402
from aresdb_amd.executor import column_input, constant_input  # noqa: E402


def timed(be, fn, reps=5):
    fn()
    be.wait()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        be.wait()
        best = min(best, time.perf_counter() - t0)
    return best, r


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    be = abi.load_hip_backend()
    be.call("BootstrapDevice")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    ts = torch.randint(0, 86400 * 7, (n,), dtype=torch.int32, device=dev, generator=g)
    d1 = torch.randint(0, 100, (n,), dtype=torch.int32, device=dev, generator=g)
    m = torch.rand((n,), dtype=torch.float32, device=dev, generator=g) * 100
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    pred = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    out = []

    def rec(name, secs, bytes_per_row, rows=n, extra=None):
        r = {"op": name, "rows": rows, "ms": secs * 1e3, "rows_per_s": rows / secs,
             "GBps_algorithmic": rows * bytes_per_row / secs / 1e9}
        if extra:
            r.update(extra)
        out.append(r)
        print(json.dumps(r), flush=True)

    t, _ = timed(be, lambda: be.call("InitIndexVector", idx.data_ptr(), 0, n, None, 0))
    rec("InitIndexVector", t, 4)
    for sel, thr in ((0.1, 10), (0.5, 50), (0.9, 90)):
        def f():
            be.call("InitIndexVector", idx.data_ptr(), 0, n, None, 0)
            return be.call("BinaryFilter", column_input(slice_from_pointer(d1.data_ptr(), abi.Uint32, n)),
                           constant_input(thr), idx.data_ptr(), pred.data_ptr(), n, None, 0, None, 0,
                           abi.LessThan, None, 0)
        t, cnt = timed(be, f)
        t0, _ = timed(be, lambda: be.call("InitIndexVector", idx.data_ptr(), 0, n, None, 0))
        rec(f"BinaryFilter sel={sel}", t - t0, 4, extra={"survivors": cnt})
    # dimension transform (Floor) + measure transform on the surviving index vector
    cnt = f()
    dimv = torch.empty(cnt, dtype=torch.int32, device=dev)
    dimn = torch.empty(cnt, dtype=torch.uint8, device=dev)
    meas = torch.empty(cnt, dtype=torch.float64, device=dev)
    ov = abi.OutputVector()
    ov.Vector.Dimension.DimValues, ov.Vector.Dimension.DimNulls = dimv.data_ptr(), dimn.data_ptr()
    ov.Vector.Dimension.DataType = abi.Uint32
    ov.Type = abi.DimensionOutput
    t, _ = timed(be, lambda: be.call("BinaryTransform", column_input(slice_from_pointer(ts.data_ptr(), abi.Uint32, n)),
                                     constant_input(3600), ov, idx.data_ptr(), cnt, None, 0, abi.Floor, None, 0))
    rec("BinaryTransform Floor->dim", t, 4, rows=cnt)
    mo = abi.OutputVector()
    mo.Vector.Measure.Values, mo.Vector.Measure.DataType, mo.Vector.Measure.AggFunc = meas.data_ptr(), abi.Float64, abi.AGGR_SUM_FLOAT
    mo.Type = abi.MeasureOutput
    t, _ = timed(be, lambda: be.call("UnaryTransform", column_input(slice_from_pointer(m.data_ptr(), abi.Float32, n)),
                                     mo, idx.data_ptr(), cnt, None, 0, abi.Noop, None, 0))
    rec("UnaryTransform Noop->measure f64", t, 4, rows=cnt)

    # group-by: 4 x u32 dims, f64 measure; ~1M groups
    for rows, groups_hint in ((min(n, 1 << 26), "1.68M"), (min(n, 1 << 22), "1.68M")):
        cap = rows
        dims = torch.empty(cap * 20, dtype=torch.uint8, device=dev)
        dv = dims[: cap * 16].view(torch.int32).view(4, cap)
        dv[0] = (ts[:rows] // 3600)
        dv[1] = d1[:rows]
        dv[2] = torch.randint(0, 50, (rows,), dtype=torch.int32, device=dev, generator=g)
        dv[3] = torch.randint(0, 2, (rows,), dtype=torch.int32, device=dev, generator=g)
        dims[cap * 16:] = 1
        mval = (m[:rows]).double()
        outd = torch.empty_like(dims)
        outm = torch.empty(cap, dtype=torch.float64, device=dev)
        hashes = torch.empty(cap, dtype=torch.int64, device=dev)
        hashes2 = torch.empty(cap, dtype=torch.int64, device=dev)
        ix = torch.empty(cap, dtype=torch.int32, device=dev)
        ix2 = torch.empty(cap, dtype=torch.int32, device=dev)

        def dvec(d, h, i):
            v = abi.DimensionVector()
            v.DimValues, v.HashValues, v.IndexVector, v.VectorCapacity = d.data_ptr(), h.data_ptr(), i.data_ptr(), cap
            for k, c in enumerate((0, 0, 4, 0, 0)):
                v.NumDimsPerDimWidth[k] = c
            return v
        torch.cuda.synchronize()
        t, grp = timed(be, lambda: be.call("HashReduce", dvec(dims, hashes, ix), mval.data_ptr(), dvec(outd, hashes2, ix2),
                                           outm.data_ptr(), 8, rows, abi.AGGR_SUM_FLOAT, None, 0), reps=3)
        rec("HashReduce 4xu32+f64", t, 28, rows=rows, extra={"groups": grp})

        def sr():
            be.call("InitIndexVector", ix.data_ptr(), 0, rows, None, 0)
            be.call("Sort", dvec(dims, hashes, ix), rows, None, 0)
        t, _ = timed(be, sr, reps=3)
        rec("InitIndex+Sort", t, 20, rows=rows)
        t, grp = timed(be, lambda: be.call("Reduce", dvec(dims, hashes, ix), mval.data_ptr(), dvec(outd, hashes2, ix2),
                                           outm.data_ptr(), 8, rows, abi.AGGR_SUM_FLOAT, None, 0), reps=3)
        rec("Reduce", t, 12 + 28, rows=rows, extra={"groups": grp})
        del dims, outd, outm, hashes, hashes2, ix, ix2, mval
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/microbench.json", "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
