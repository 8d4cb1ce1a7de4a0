import json, os, sys
import torch
sys.path.insert(0, "/root/repo")
from aresdb_amd import abi, workload
from aresdb_amd.executor import column_input, constant_input
n = 100_000_000
be = abi.load_hip_backend(); be.call("BootstrapDevice")
dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(1)
b = workload.c3_batch(n, g, dev, null_fraction=0.0)
idx = torch.empty(n, dtype=torch.int32, device=dev); pred = torch.empty(n, dtype=torch.uint8, device=dev)
for dbg in ("0", "4", "5", "7"):
    os.environ["ARES_F_DEBUG"] = dbg
    best = 1e9
    for _ in range(5):
        be.call("InitIndexVector", idx.data_ptr(), 0, n, None, 0)
        be.profiler_enable(True)
        try:
            be.call("BinaryFilter", column_input(b["d1"].vp), constant_input(50), idx.data_ptr(), pred.data_ptr(), n, None, 0, None, 0, abi.LessThan, None, 0)
        except Exception as e:
            pass
        be.wait(); rep = be.profiler_report(); be.profiler_enable(False)
        best = min(best, rep["filter_fast_kernel"][1])
    print(json.dumps({"debug": dbg, "ms": best}), flush=True)
