import sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from aresdb_amd import abi
be = abi.load_hip_backend(); be.call("BootstrapDevice")
dev = torch.device("cuda:0")
for length in (500000, 1980314, 4000001):
    cap = length + 100
    hashes = torch.full((cap,), 0x1234567, dtype=torch.int64, device=dev)
    idx = torch.arange(cap, dtype=torch.int32, device=dev)
    vals = torch.ones(cap, dtype=torch.int32, device=dev); vals[0] = 1000000
    dims = torch.zeros(cap * 5, dtype=torch.uint8, device=dev)
    outd = torch.zeros_like(dims); outh = torch.zeros_like(hashes); outi = torch.zeros_like(idx)
    outv = torch.zeros(cap, dtype=torch.int32, device=dev)
    def dvec(d, h, i, nd):
        v = abi.DimensionVector(); v.DimValues, v.HashValues, v.IndexVector, v.VectorCapacity = d.data_ptr(), h.data_ptr(), i.data_ptr(), cap
        for k, c in enumerate((0, 0, nd, 0, 0)): v.NumDimsPerDimWidth[k] = c
        return v
    torch.cuda.synchronize()
    for nd in (0, 1):
        bad = 0
        for t in range(10):
            g = be.call("Reduce", dvec(dims, hashes, idx, nd), vals.data_ptr(), dvec(outd, outh, outi, nd), outv.data_ptr(), 4, length, abi.AGGR_SUM_UNSIGNED, None, 0)
            be.wait()
            got = int(outv[0].item())
            if g != 1 or got != 1000000 + length - 1: bad += 1; last = (g, got)
        print(length, "nd", nd, "bad", bad, "want", 1000000 + length - 1, last if bad else "", flush=True)
