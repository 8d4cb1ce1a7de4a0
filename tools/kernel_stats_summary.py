"""profiles/r<N>_kernel_stats_summary.md from the rocprofv3 `--kernel-trace --stats` CSVs that tools/gpu_r<N>_evidence.sh leaves
(copied to profiles/r<N>_<tag>_kernel_stats_rocprofv3.csv):  [ARES_ROUND=5] python tools/kernel_stats_summary.py"""
import csv
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = n.strip('"').replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("ares::")[-1].split("<")[0]


ROUND = os.environ.get("ARES_ROUND", "6")
HEAD = {
    "5": ["# Round 5 — rocprofv3 `--kernel-trace --stats` summaries (tools/gpu_r5_evidence.sh; one MI355X box per call)", "",
          "Durations in microseconds per launch; `share` = of all kernel time of the traced command.  Commands: C3 = `bench.py --steps 5 --warmup 2 --no-legs`",
          "(1 B rows per step, 15 batches of 64 Mi rows; the priming passes on the generic kernels included); live = the same shard as 477 batches of 2 Mi rows;",
          "trips = `tools/bench_configs.py trips` at 1 B rows (both queries); C2 = `tools/bench_configs.py c2` (100 M rows, three selectivities, one and four batches);",
          "C4 = `tools/bench_configs.py c4spec` (1 B rows, 50 M-key cuckoo join, Sort + Reduce).  C3 and live: the library as committed at the end of the",
          "round; the trips and C2 traces were taken while the merges' result words still went through the pinned slot by default (`ARES_RESULT_PINNED=1`,",
          "since made opt-in: it cost the live leg 6 ms — `r5_evidence_ab.txt`; it adds a few microseconds to `hr_merge_rtc`, nothing to the other kernels).", ""],
    "6": ["# Round 6 — rocprofv3 `--kernel-trace --stats` summaries (tools/gpu_r6_evidence.sh; ONE MI355X box, one call, the library as committed)", "",
          "Durations in microseconds per launch; `share` = of all kernel time of the traced command.  Commands: C3 = `bench.py --steps 5 --warmup 2 --no-legs`",
          "(1 B rows per step, 15 batches of 64 Mi rows; the priming passes included); sort = the same shard, COUNT(*) through Sort + Reduce (`--sort-path count`);",
          "archive = the same shard sorted by (ts, d3), both run-length encoded (`--archive`); live = the shard as 512 batches of 2 Mi rows; trips =",
          "`tools/bench_configs.py trips` at 1 B rows (SUM(fare) through HashReduce, COUNT(*) through Sort + Reduce); C2 = `tools/bench_configs.py c2`;",
          "C4 = `tools/bench_configs.py c4spec` (1 B rows, 50 M-key cuckoo join, 50 M groups through Sort + Reduce over materialised vectors).", ""],
}[ROUND]
SECTIONS = (("c3", "C3 headline"), ("sort", "C3 dimensions, COUNT(*) through Sort + Reduce (scan-fed: sr_scan_rtc, sr_merge_kernel, sr_emit_kernel)"),
            ("archive", "archive batches (mode-3 ts and d3 decoded once per batch: expand_runs_kernel)"), ("live", "C3 as 2 Mi-row live batches"),
            ("trips", "trips-shaped leg (SUM(fare) via HashReduce + COUNT(*) via Sort + Reduce)"), ("c2", "C2"), ("c4", "C4 at its stated size"))
out = list(HEAD)
for tag, title in SECTIONS:
    f = os.path.join(ROOT, "profiles", f"r{ROUND}_{tag}_kernel_stats_rocprofv3.csv")
    if not os.path.exists(f):
        continue
    out += [f"## {title}", "", "| kernel | launches | avg | min | max | share % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        if not any(k in r["Name"] for k in ("ares::", "_rtc")) or float(r["Percentage"]) < 0.3:
            continue
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                   f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    out.append("")
open(os.path.join(ROOT, "profiles", f"r{ROUND}_kernel_stats_summary.md"), "w").write("\n".join(out))
