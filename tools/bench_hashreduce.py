"""HashReduce alone on C3-shaped dimension vectors (development aid)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aresdb_amd import abi

def main():
    rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 26
    be = abi.load_hip_backend(); be.call("BootstrapDevice")
    dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(1)
    cap = rows
    def mk(groups_mode):
        dims = torch.empty(cap * 20, dtype=torch.uint8, device=dev)
        dv = dims[: cap * 16].view(torch.int32).view(4, cap)
        if groups_mode == "c3":
            dv[0] = torch.randint(0, 168, (rows,), dtype=torch.int32, device=dev, generator=g) * 3600
            dv[1] = torch.randint(0, 90, (rows,), dtype=torch.int32, device=dev, generator=g)
            dv[2] = torch.randint(0, 50, (rows,), dtype=torch.int32, device=dev, generator=g)
            dv[3] = torch.randint(0, 2, (rows,), dtype=torch.int32, device=dev, generator=g)
        else:
            dv[0] = torch.randint(0, int(groups_mode), (rows,), dtype=torch.int32, device=dev, generator=g)
            dv[1] = 1; dv[2] = 2; dv[3] = 3
        dims[cap * 16:] = 1
        return dims
    def dvec(d):
        v = abi.DimensionVector(); v.DimValues, v.VectorCapacity = d.data_ptr(), cap
        for k, c in enumerate((0, 0, 4, 0, 0)): v.NumDimsPerDimWidth[k] = c
        return v
    mval = torch.rand((rows,), dtype=torch.float64, device=dev, generator=g)
    outm = torch.empty(cap, dtype=torch.float64, device=dev)
    for mode in ("c3",):
        dims = mk(mode); outd = torch.empty_like(dims); torch.cuda.synchronize()
        for env in [("lds", "0"), ("lds", "8"), ("lds", "16"), ("lds", "24")]:
            if env[0] == "global": os.environ["ARES_HASH_REDUCE"] = "global"
            else: os.environ.pop("ARES_HASH_REDUCE", None)
            os.environ["ARES_HR_DEBUG"] = env[1]
            best = 1e9
            for _ in range(3):
                be.profiler_enable(True)
                t0 = time.perf_counter()
                grp = be.call("HashReduce", dvec(dims), mval.data_ptr(), dvec(outd), outm.data_ptr(), 8, rows, abi.AGGR_SUM_FLOAT, None, 0)
                be.wait(); dt = time.perf_counter() - t0
                rep = be.profiler_report(); be.profiler_enable(False)
                best = min(best, dt)
            print(json.dumps({"groups_mode": mode, "path": env[0], "debug": env[1], "rows": rows, "ms": best * 1e3, "groups": grp,
                              "kernels": {k: round(v[1] / v[0], 3) for k, v in rep.items()}}), flush=True)
        del dims, outd
main()
