"""One resident C3 batch through the ABI call sequence (in-ABI fusion: filter kernels + fused scan +
merge); prints per-kernel HIP-event times.  ARES_HR_DEBUG selects timing experiments."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aresdb_amd import abi, workload
from aresdb_amd.executor import BatchContext, BatchExecutor
from aresdb_amd.queries import c3_plan

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 26
be = abi.load_hip_backend(); be.call("BootstrapDevice")
dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(1)
batch = workload.c3_batch(n, g, dev, null_fraction=0.01)
torch.cuda.synchronize()
best = {}
for rep in range(4):
    ctx = BatchContext(be, c3_plan(use_hash_reduction=True)); ex = BatchExecutor(ctx)
    be.profiler_enable(True)
    for _ in range(2):
        ex.run({k: rc.vp for k, rc in batch.items()}, n)
    be.wait(); k = be.profiler_report(); be.profiler_enable(False)
    for name, (c, ms) in k.items():
        best[name] = min(best.get(name, 1e9), ms / c)
    groups = ctx.result_size
    ctx.release()
print(json.dumps({"debug": os.environ.get("ARES_HR_DEBUG", "0"), "groups": groups, "kernels": {k: round(v, 3) for k, v in best.items()}}))
