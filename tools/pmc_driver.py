"""One pass of the C3 hot path over a single resident batch (default 64 Mi rows) — the workload the
rocprofv3 --pmc passes of tools/gpu_pmc.sh profile (per-kernel HBM bytes, SQ stall counters)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aresdb_amd import abi, workload
from aresdb_amd.executor import BatchContext, BatchExecutor
from aresdb_amd.queries import c3_plan

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 26
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
be = abi.load_hip_backend(); be.call("BootstrapDevice")
dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(1)
batch = workload.c3_batch(n, g, dev, null_fraction=0.01)
torch.cuda.synchronize()
ctx = BatchContext(be, c3_plan(use_hash_reduction=True)); ex = BatchExecutor(ctx)
for _ in range(reps):
    ex.run({k: rc.vp for k, rc in batch.items()}, n)
print("groups", ctx.result_size)
ctx.release()
