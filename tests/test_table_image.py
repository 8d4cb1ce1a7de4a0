"""Table images (hash_reduce_lds.hip): the partitions' LDS tables persist between the HashReduce calls of a query; a merge
starts from the previous call's image, appends new groups only, and leaves the measure vector defined-but-unwritten until
somebody reads it."""
import os
import subprocess
import sys

import pytest

import harness as H

_SCRIPT = r"""
import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import harness as H
from aresdb_amd import smoke
hip, oracle = H.hip_backend(), H.oracle_backend()
rng = np.random.default_rng(77)
data = [smoke.synth_batch(rng, n, null_fraction=0.02) for n in (40000, 30000, 1, 52000, 9000, 45000, 300, 41000)]
plan = smoke.c3_plan(True)
want = smoke.run_query(oracle, plan, data)[0]
streams = [hip.call("CreateCudaStream", 0) for _ in range(2)]
for attempt in range(2):   # the second query finds every kernel of the shape loaded
    hip.profiler_enable(True)
    got = smoke.run_query_native(hip, plan, data, streams=streams)[0]
    hip.wait(); kernels = hip.profiler_report(); hip.profiler_enable(False)
    smoke.compare_results(got, want)
py = smoke.run_query(hip, plan, data)[0]   # the Python mirror of the executor (one stream)
smoke.compare_results(py, want)
print("KERNELS", {k: v[0] for k, v in kernels.items() if k.startswith("hr_")})
"""


@pytest.mark.gpu
@pytest.mark.parametrize("verify", ["0", "1"], ids=["", "verify_clean"])
def test_merges_start_from_the_previous_image(verify):
    r = subprocess.run([sys.executable, "-c", _SCRIPT], cwd=H.ROOT,
                       env={**os.environ, "ARES_LEAN_MIN_GROUPS": "0", "ARES_MEM_VERIFY_CLEAN": verify}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("KERNELS")][-1]
    kernels = eval(line[8:])
    # eight batches: most through the specialised merge; the measure vector written when the result is fetched
    # (a batch whose total length crosses a power of two changes the partition count: that one takes the generic merge)
    assert kernels.get("hr_merge_rtc", 0) >= 5 and 1 <= kernels.get("hr_image_values_kernel", 0) <= 3, kernels


@pytest.mark.gpu
def test_results_do_not_depend_on_the_image_switch():
    r = subprocess.run([sys.executable, "-c", _SCRIPT], cwd=H.ROOT, env={**os.environ, "ARES_LEAN_MIN_GROUPS": "0", "ARES_IMAGE": "0"},
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("KERNELS")][-1]
    assert "hr_image_values_kernel" not in line, line


_FORK_SCRIPT = r"""
import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import harness as H
from aresdb_amd import abi
NDW = (0, 0, 2, 0, 0)

def batch(rng, n, lo, hi):
    return {"d1": rng.integers(lo, hi, n).astype(np.uint32), "d2": rng.integers(0, 40, n).astype(np.uint32), "m": rng.integers(0, 1000, n).astype(np.uint32)}

def reduce_into(b, src, prev, dst, bt, cap):
    # one batch of the Go sequence: transforms into rows [prev, prev + n) of `src`, then HashReduce(src -> dst)
    n = len(bt["d1"])
    cols = {k: H.Column(b, abi.Uint32, v) for k, v in bt.items()}
    idx = H.Buf(b, nbytes=4 * n)
    b.call("InitIndexVector", idx.ptr, 0, n, None, 0)
    offs = src[0].dim_offsets()
    for d, name in enumerate(("d1", "d2")):
        b.call("UnaryTransform", cols[name].input(), H.dimension_output(src[0].values.ptr + offs[d][0] + 4 * prev, src[0].values.ptr + offs[d][1] + prev, abi.Uint32),
               idx.ptr, n, None, 0, abi.Noop, None, 0)
    b.call("UnaryTransform", cols["m"].input(), H.measure_output(src[1].ptr + 4 * prev, abi.Uint32, abi.AGGR_SUM_UNSIGNED), idx.ptr, n, None, 0, abi.Noop, None, 0)
    b.wait()
    for c in cols.values():
        c.free()
    idx.free()
    g = b.call("HashReduce", src[0].struct(), src[1].ptr, dst[0].struct(), dst[1].ptr, 4, prev + n, abi.AGGR_SUM_UNSIGNED, None, 0)
    b.wait()
    return g

def table(buf, g):
    rows = buf[0].rows(g)
    vals = buf[1].read(np.uint32, g)
    return {r: int(v) for r, v in zip(rows, vals)}

def run(b):
    rng = np.random.default_rng(12)
    cap = 200000
    mk = lambda: (H.DimVector(b, cap, NDW, False, False), H.Buf(b, nbytes=4 * cap))
    Z, A, B, C = mk(), mk(), mk(), mk()
    gA = reduce_into(b, Z, 0, A, batch(rng, 30000, 0, 300), cap)          # A: the query's first result (leaves an image)
    X, Y = batch(rng, 20000, 300, 330), batch(rng, 25000, 1000, 1400)     # X adds few new groups, Y many others
    # both forks read A's rows [0, gA) and write their batch behind them: the second fork's transforms overwrite the first's
    gB = reduce_into(b, A, gA, B, X, cap)                                 # A -> B
    gC = reduce_into(b, A, gA, C, Y, cap)                                 # A -> C   (a fork: same lineage as B, not its ancestor)
    gD = reduce_into(b, C, gC, B, batch(rng, 15000, 0, 1400), cap)        # C -> B's buffer
    out = {"A": gA, "B": gB, "C": gC, "D": gD, "table": table(B, gD)}
    for x in (Z, A, B, C):
        x[0].free(); x[1].free()
    return out

hip, oracle = H.hip_backend(), H.oracle_backend()
want = run(oracle)
for attempt in range(2):  # (the second time every generated kernel is loaded)
    hip.profiler_enable(True)
    got = run(hip)
    kernels = hip.profiler_report(); hip.profiler_enable(False)
    assert {k: got[k] for k in "ABCD"} == {k: want[k] for k in "ABCD"}, (got["A"], got["B"], got["C"], got["D"], want["D"])
    assert got["table"] == want["table"]
print("KERNELS", {k: v[0] for k, v in kernels.items() if k.startswith("hr_")})
"""


@pytest.mark.gpu
def test_a_forked_result_is_not_taken_for_an_ancestor():
    """Two results forked from one input (A -> B with one batch, A -> C with another), then C reduced into B's buffer: B
    shares C's lineage and is no larger, but its rows behind A's are another batch's groups — the merge must not keep
    them as "the first rows of C" (round-5 advisor finding; legal through the C ABI, never issued by the Go host)."""
    r = subprocess.run([sys.executable, "-c", _FORK_SCRIPT], cwd=H.ROOT,
                       env={**os.environ, "ARES_LEAN_MIN_GROUPS": "0", "ARES_MIN_PART_BITS": "2"}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("KERNELS")][-1]
    assert eval(line[8:]).get("hr_merge_rtc", 0) >= 3, line  # the calls did take the image-mode merges
