"""tools/prototypes/: algorithms prepared for the next round that are not part of the libraries yet.  Their logic is
host+device code; the host side is checked here so that the GPU time of the next round goes into measuring, not
debugging."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not on PATH")
def test_sort_topbits_fixup_equals_a_full_stable_sort(tmp_path):
    """Radix sort of the top 32 bits + detect / sort fix-up == stable sort by all 64 bits, or the fallback flag is raised
    (segments beyond the walk bound); thread order does not matter; listed segments are disjoint."""
    src = os.path.join(ROOT, "tools", "prototypes", "sort_topbits_fixup_test.cpp")
    exe = tmp_path / "fixup_test"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", str(exe), src], check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 failed" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
