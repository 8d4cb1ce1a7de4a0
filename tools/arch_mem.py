import sys, time, torch
sys.path.insert(0, ".")
from aresdb_amd import workload
dev = torch.device("cuda:0")
t = time.time()
b = workload.c3_shard(int(1e9), 1 << 26, 2, dev, 0.01, archive=True)
torch.cuda.synchronize()
print("shard s", round(time.time() - t, 1), "reserved GB", torch.cuda.memory_reserved() / 1e9, "allocated GB", torch.cuda.memory_allocated() / 1e9)
free, total = torch.cuda.mem_get_info()
print("free GB", free / 1e9, "total", total / 1e9)
print({k: (v.length, getattr(v, "runs", None), v.blob.numel() / 1e6) for k, v in b[0].items()})
