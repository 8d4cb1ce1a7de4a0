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
