"""Aggregates rocprofv3 counter_collection CSVs (one directory per pass) into a per-kernel table."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "ares::" not in name and "_rtc" not in name:
            continue
        short = (name.split("ares::")[1] if "ares::" in name else name).replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[short][r["Counter_Name"]] += 1
counters = sorted({c for k in acc.values() for c in k})
print("| kernel | " + " | ".join(counters) + " |")
print("|---|" + "---|" * len(counters))
for k in sorted(acc):
    row = []
    for c in counters:
        n = calls[k].get(c, 0)
        row.append(f"{acc[k][c] / n:.4g} (x{n})" if n else "-")
    print(f"| {k} | " + " | ".join(row) + " |")
print()
print("Values are per-dispatch averages.  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them;")
print("MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, i.e.")
print("HBM read bytes = 2 x FETCH_SIZE x 1024 for 16 B/lane streams (narrower accesses uncalibrated).")

# machine-readable per-launch HBM traffic (gfx950 correction applied to the read side)
import json
out = {}
for k in sorted(acc):
    f = acc[k].get("FETCH_SIZE"); w = acc[k].get("WRITE_SIZE")
    if f is None or w is None:
        continue
    fpl = f / calls[k]["FETCH_SIZE"]; wpl = w / calls[k]["WRITE_SIZE"]
    out[k] = {"fetch_size_KiB_per_launch": fpl, "write_size_KiB_per_launch": wpl,
              "hbm_bytes_per_launch": 2 * fpl * 1024 + wpl * 1024,
              "note": "2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request on wide reads) + WRITE_SIZE"}
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 67108864
import hashlib
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aresdb_amd", "lib", "libalgorithm.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
json.dump({"rows_per_batch": rows, "libalgorithm_sha256_16": sha, "kernels": out}, open(os.path.join(root, "traffic.json"), "w"), indent=1)
