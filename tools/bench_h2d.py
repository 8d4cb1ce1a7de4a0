"""C3 with the batch pipeline of the Go host (query/aql_processor.go:513-540, :860-881): every batch's
columns start in pinned host memory (libmem HostAlloc), are uploaded with AsyncCopyHostToDevice on a
transfer stream into freshly allocated device buffers, and the upload of batch k+1 overlaps the
execution of batch k on the query stream; the driver frees the columns before the aggregation stage.
Reports the PCIe-inclusive rows/s (never the bench `value`: that one has the shard resident in HBM)."""
import ctypes as C
import json, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aresdb_amd import abi, workload
from aresdb_amd.driver import NativeQuery
from aresdb_amd.queries import c3_plan

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 26
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
be = abi.load_hip_backend(); be.call("BootstrapDevice")
dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(1)
batch = workload.c3_batch(n, g, dev, null_fraction=0.01)
names = [k for k, _ in workload.C3_COLUMNS]
# the batch in pinned host memory, column blobs exactly as they are uploaded
host = {}
for k in names:
    rc = batch[k]
    nbytes = rc.blob.numel()
    hp = be.call("HostAlloc", nbytes)
    torch.cuda.synchronize()
    be.call("AsyncCopyDeviceToHost", hp, rc.blob.data_ptr(), nbytes, None, 0); be.wait()
    host[k] = (hp, nbytes, rc)
del batch
torch.cuda.synchronize()
xfer, query = be.call("CreateCudaStream", 0), be.call("CreateCudaStream", 0)
bytes_per_batch = sum(v[1] for v in host.values())


def upload():
    """transferLiveBatch: one device allocation per column + async H2D on the transfer stream."""
    cols, ptrs = {}, []
    for k in names:
        hp, nbytes, rc = host[k]
        dp = be.call("DeviceAllocate", nbytes, 0)
        be.call("AsyncCopyHostToDevice", dp, hp, nbytes, xfer, 0)
        vp = abi.VectorPartySlice()
        if rc.has_nulls:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = dp, 0, rc.values_off
        else:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = dp + rc.values_off, 0, 0
        vp.DataType, vp.Length, vp.StartingIndex = rc.data_type, rc.length, 0
        cols[k] = vp
        ptrs.append(dp)
    be.call("WaitForCudaStream", xfer, 0)
    return cols, ptrs


def run(nb):
    q = NativeQuery(be, c3_plan(use_hash_reduction=True), names, device=0, stream=query)
    nxt = upload()
    for b in range(nb):
        cols, ptrs = nxt
        holder = {}
        t = None
        if b + 1 < nb:  # the Go host uploads batch k+1 on its own goroutine while batch k executes
            t = threading.Thread(target=lambda: holder.setdefault("x", upload()))
            t.start()
        q.run(cols, n, owned_allocations=ptrs)
        if t:
            t.join()
            nxt = holder["x"]
    groups = q.result_size
    q.release()
    return groups

run(2)
torch.cuda.synchronize()
t0 = time.perf_counter(); groups = run(batches); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"rows_per_batch": n, "batches": batches, "bytes_per_batch": bytes_per_batch, "groups": groups,
                  "ms_per_batch": dt / batches * 1e3, "rows_per_s": n * batches / dt,
                  "h2d_GBps": bytes_per_batch * batches / dt / 1e9}))
