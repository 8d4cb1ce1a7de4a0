"""The kernels libalgorithm.so generates at run time (hr_rtc.hip: scan and merge compiled per query shape, and
their vector-sourced variants) must compile for gfx950 — checked without a GPU: tools/rtc_check.cpp asks the
library for the sources of the C3 plan and of the vector shapes and hands them to hiprtc."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_generated_kernels_compile_for_gfx950(tmp_path):
    lib = os.path.join(ROOT, "aresdb_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libalgorithm.so")):
        pytest.skip("libalgorithm.so not built")
    exe = tmp_path / "rtc_check"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "aresdb_amd", "csrc", "algo"), "-o", str(exe),
                    os.path.join(ROOT, "tools", "rtc_check.cpp"), "-L" + lib, "-lalgorithm", "-lhiprtc", "-Wl,-rpath," + lib],
                   check=True, timeout=600)
    out = subprocess.run([str(exe), str(tmp_path / "k")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    for what in ("scan", "merge", "compact scan", "compact merge", "table scan", "compact scan (2 dims)",
                 "compact merge (2 dims)", "table scan (2 dims, 1 partition)", "narrow compact scan", "narrow compact merge",
                 "narrow scan", "narrow merge", "narrow table scan", "narrow region-A merge", "signed narrow compact scan",
                 "signed narrow compact merge", "compact merge + image out", "compact merge from image", "merge from image",
                 "narrow compact merge from image", "sort scan (COUNT)", "sort scan (SUM into 8 bytes)", "narrow sort scan (COUNT)",
                 "8-dimension compact scan", "8-dimension compact merge", "8-dimension compact merge from image", "8-dimension table scan",
                 "8-dimension region-A merge", "8-dimension sort scan (COUNT)", "vector sort scan nd 2", "vector sort scan nd 8",
                 "vector sort scan nd 1, one partition", "vector sort scan, slots 4 4 2 1"):
        assert f"{what} compile rc 0" in out.stdout, what
    for nd in (1, 4):
        for vw in (4, 8):
            assert f"vector scan nd {nd} vw {vw} compile rc 0" in out.stdout
            assert f"vector merge nd {nd} vw {vw} compile rc 0" in out.stdout
    # Budgets per generated kernel FAMILY (static vector-instruction count of the code object, registers, LDS): the scans are
    # bound by vector-ALU issue, so an instruction count that creeps up is a slowdown that no test of results
    # sees.  Limits = the count at the end of round 6 + ~5 %.  DIRECT scan with 16-byte lines / compact lines, TABLE scan,
    # the merges (16-byte, compact, from a table image, region A), their narrow variants (1- / 2-byte columns and slots), and
    # the 64-bit-key scans of the Sort + Reduce path (murmur3_x64_128: ~30 % more than the 32-bit scan of the same shape).
    import re
    budgets = {"k": 755, "k_compact": 840, "k_table": 1500, "k_merge": 3650, "k_cmerge": 5550, "k_cmerge_img2": 5400, "k_namerge": 3320,
               "k_nlines": 660, "k_ncompact": 750, "k_ntable": 1320, "k_nmerge": 3250, "k_ncmerge": 5150,
               "k_sort_count": 990, "k_sort_sum8": 1015, "k_sort_trips": 745, "k_8compact": 1025, "k_8sort": 1245,
               # the vector-sourced sort scans (Sort + Reduce over materialised rows: C4's is k_vsort2).  The eight-dimension one
               # (k_vsort8: 128 VGPRs and 12 bytes of scratch per lane) compiles, above, and is not held to "no scratch"
               "k_vsort2": 845, "k_vsort1": 630, "k_vsort_narrow": 975}
    for tag, limit in budgets.items():
        co = str(tmp_path / f"{tag}.co")
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True)
        dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True)
        if notes.returncode != 0 or dis.returncode != 0 or ".vgpr_count" not in notes.stdout:
            continue  # (tools absent: the compile check above still holds)
        assert ".private_segment_fixed_size: 0" in notes.stdout, tag  # no scratch: registers and LDS hold the working set
        vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", notes.stdout).group(1))
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", notes.stdout).group(1))
        assert vgprs <= 128 and lds <= 160 * 1024, (tag, vgprs, lds)  # 1024 lanes = 4 wavefronts per SIMD; one workgroup per CU
        valu = sum(1 for ln in dis.stdout.splitlines() if ln.strip().startswith("v_"))
        assert valu <= limit, (tag, valu, limit)
    # the plan-sourced kernels keep their whole working set in registers and LDS: no scratch
    for tag in ("k", "k_compact", "k_cmerge", "k_table"):
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(tmp_path / f"{tag}.co")], capture_output=True, text=True)
        if notes.returncode == 0 and ".private_segment_fixed_size" in notes.stdout:
            assert ".private_segment_fixed_size: 0" in notes.stdout, tag
            # 1024 lanes per workgroup = 4 wavefronts per SIMD: at most 128 VGPRs each; one workgroup's LDS fits a CU's 160 KB
            import re
            vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", notes.stdout).group(1))
            lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", notes.stdout).group(1))
            assert vgprs <= 128 and lds <= 160 * 1024, (tag, vgprs, lds)
    # murmur's h * 5 + c must not come back as the 64-bit multiply-add the compiler prefers (20 per tile and lane: 4 % of
    # the compact scan, profiles/r3_experiments.md); a handful remain outside the tile loop
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", str(tmp_path / "k_compact.co")], capture_output=True, text=True)
    if dis.returncode == 0 and "v_mul_lo_u32" in dis.stdout:
        assert dis.stdout.count("v_mad_u64_u32") <= 10, dis.stdout.count("v_mad_u64_u32")
        # The compact scan is bound by vector-ALU issue (~750 VALU instructions per lane and 4096-row tile, four
        # waves per SIMD, four cycles each): its instruction count is its speed.  798 at the end of round 4.
        valu = sum(1 for ln in dis.stdout.splitlines() if ln.strip().startswith("v_"))
        assert valu <= 840, valu
        # the columns are read once: streaming (non-temporal) loads, -3 % on the scan (profiles/r4_experiments.md)
        assert sum(1 for ln in dis.stdout.splitlines() if "global_load_dwordx4" in ln and " nt" in ln) >= 5
