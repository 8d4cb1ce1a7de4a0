import sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import harness as H
from aresdb_amd import abi, queries
from aresdb_amd.columns import DeviceColumn
from aresdb_amd.executor import BatchContext, BatchExecutor
be = H.hip_backend()
rng = np.random.default_rng(1)
hi = 86400 * 30
n = 4000000
thr = hi // 2
ts = rng.integers(0, hi, n).astype(np.uint32)
valid = rng.random(n) >= 0.01
want1 = int(((ts < thr) & valid).sum())
def rd(ptr, count, dt):
    out = np.empty(count, dt)
    be.d2h(out.ctypes.data_as(C.c_void_p), ptr, out.nbytes, stream); be.wait(stream)
    return out
def reduce_variant(ex, variant):
    c = ex.ctx
    length = c.result_size + c.size
    c.call("InitIndexVector", c.dim_index_vec[0], 0, length, c.stream, c.device)
    if variant & 1: be.wait()
    c.call("Sort", c.dimension_vector(0), length, c.stream, c.device)
    if variant & 2: be.wait()
    c.result_size = c.call("Reduce", c.dimension_vector(0), c.measure_vec[0], c.dimension_vector(1), c.measure_vec[1], 4, length, abi.AGGR_SUM_UNSIGNED, c.stream, c.device)
    be.wait(c.stream)
use_stream = os.environ.get("DBG_STREAM") == "1"
stream = be.call("CreateCudaStream", 0) if use_stream else None
for variant in (0, 0):
    bad = 0
    for trial in range(4):
        ctx = BatchContext(be, queries.c2_plan(thr), stream=stream); ex = BatchExecutor(ctx)
        for b in range(3):
            col = DeviceColumn(be, abi.Uint32, ts, valid=valid, stream=stream)
            ctx.prepare_for_filtering({"ts": col.vp}, n)
            ex.pre_exec(); ex.filter(); ex.join(); ex.project()
            reduce_variant(ex, variant)
            ex.post_exec()
            col.free()
        out = rd(ctx.measure_vec[0], 1, np.uint32)
        if int(out[0]) != 3 * want1: bad += 1
        ctx.release()
    print("variant", variant, "bad", bad, "of 4", flush=True)
