"""Host cost of libmem's calls as the Go host issues them per batch: DeviceAllocate + DeviceFree of an index-vector-sized block with
two query streams registered (every free fences the block on the null stream and on each registered stream), WaitForCudaStream
on an idle stream, and an empty stream synchronise — microseconds per call (tools/ubench_libmem.py [bytes])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aresdb_amd import abi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8 << 20
be = abi.load_hip_backend(); be.call("BootstrapDevice")
streams = [be.call("CreateCudaStream", 0) for _ in range(2)]
for reps in (200, 2000):
    ptrs = [be.device_alloc(n, 0) for _ in range(4)]
    for p in ptrs: be.device_free(p, 0)
    t0 = time.perf_counter()
    for _ in range(reps):
        p = be.device_alloc(n, 0)
    t1 = time.perf_counter()
    # (allocate / free pairs: the block cache serves every allocation after the first few)
    for _ in range(reps):
        be.device_free(be.device_alloc(n, 0), 0)
    t2 = time.perf_counter()
    for _ in range(reps):
        be.wait(streams[0], 0)
    t3 = time.perf_counter()
    print(f"reps {reps}: DeviceAllocate (fresh blocks) {(t1 - t0) / reps * 1e6:.1f} us, allocate + free pair {(t2 - t1) / reps * 1e6:.1f} us, "
          f"WaitForCudaStream on an idle stream {(t3 - t2) / reps * 1e6:.1f} us")
    break
