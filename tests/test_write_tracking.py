"""Write tracking under ARES_MEM_VERIFY_CLEAN=1 (libmem.so checks on the device every block it hands out as
"still cleared" and aborts on stale bytes): DeviceAllocate's zero-fill contract
(reference cgoutils/memory/cuda_malloc.cu:97-104) must hold although a freed block is cleared only where a
reported write touched it.  Child processes: the variable is read when the library loads."""
import json
import os
import subprocess
import sys

import pytest

import harness as H

pytestmark = pytest.mark.gpu

ENV = {"ARES_MEM_VERIFY_CLEAN": "1"}


@pytest.mark.parametrize("first", [1000, 1060])
def test_sequence_fuzzer_under_verify_clean(first):
    """120 random ABI programs (one stream) plus four-thread / four-stream rounds, every observable buffer equal
    to the oracle's, no stale block handed out."""
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "verify_clean_sweep.py"), "--first", str(first),
                        "--seeds", "60", "--thread-rounds", "2"], cwd=H.ROOT, env={**os.environ, **ENV},
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "clean" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("name,env,args", [
    ("fused_two_streams", {}, ["--rows", str(40 << 20), "--batch-rows", str(16 << 20), "--streams", "2"]),
    ("unfused", {"ARES_FUSE": "0"}, ["--rows", str(24 << 20), "--batch-rows", str(8 << 20)]),
    ("live_batches", {}, ["--rows", str(16 << 20), "--batch-rows", str(2 << 20), "--streams", "2"]),
], ids=["fused_two_streams", "unfused", "live_batches"])
def test_c3_at_scale_under_verify_clean(name, env, args):
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tests", "scale_check.py"), *args], cwd=H.ROOT,
                       env={**os.environ, **ENV, **env}, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    report = json.loads(lines[-1])
    assert r.returncode == 0 and report["status"] == "ok", report
    assert report["env"].get("ARES_MEM_VERIFY_CLEAN") == "1"
