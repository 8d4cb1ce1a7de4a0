#!/usr/bin/env python3
"""Key-level parity of the C3 query at sizes only the GPU reaches, as a child process (the fusion /
path switches ARES_FUSE, ARES_DEFER, ARES_GROUPED, ARES_HASH_REDUCE are read once per process):

    python tests/scale_check.py --rows R --batch-rows B [--fused-extension] [--streams 2]

Runs the query through the C++ host driver on the HIP libraries, then compares group count, every
(dimension row -> sum) and every representative with the independent exact group-by of
aresdb_amd/check.py (32-bit-hash merges predicted).  Prints one JSON report; exit code 1 on mismatch."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from aresdb_amd import abi, check, workload  # noqa: E402
from aresdb_amd.driver import NativeQuery  # noqa: E402
from aresdb_amd.queries import c3_plan  # noqa: E402

NAMES = [n for n, _ in workload.C3_COLUMNS]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=float(1 << 26))
    ap.add_argument("--batch-rows", type=float, default=float(1 << 25))
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--null-fraction", type=float, default=0.01)
    ap.add_argument("--fused-extension", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="2: alternate two streams per batch like the Go host")
    ap.add_argument("--ts-range", default="", help="from,to: the Go host's two time filters in front of the query's own")
    ap.add_argument("--sort-path", default="", choices=["", "count", "sum"],
                    help="the reference's default aggregation path, Sort + Reduce: COUNT(*) / SUM(d2) as AGGR_SUM_UNSIGNED into 8 bytes; "
                         "the result must also come in ascending order of the 64-bit row hash")
    args = ap.parse_args()
    ts_range = tuple(int(x) for x in args.ts_range.split(",")) if args.ts_range else None
    dev = torch.device("cuda:0")
    be = abi.load_hip_backend()
    be.call("BootstrapDevice")
    streams = [be.call("CreateCudaStream", 0) for _ in range(args.streams)]
    batches = workload.c3_shard(int(args.rows), int(args.batch_rows), seed=args.seed, device=dev,
                                null_fraction=args.null_fraction)
    torch.cuda.synchronize()
    sort_measure = {"": None, "count": "count", "sum": "d2"}[args.sort_path]
    plan = c3_plan(use_hash_reduction=True, ts_range=ts_range, sort_measure=sort_measure)
    plan.use_fused_extension = args.fused_extension
    ctx = NativeQuery(be, plan, NAMES, device=0, stream=streams[0], streams=streams)
    sizes = []
    be.profiler_enable(True)
    for b in batches:
        ctx.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
        sizes.append(ctx.result_size)
    kernels = sorted(be.profiler_report())
    be.profiler_enable(False)
    fetched = ctx.fetch()
    if args.sort_path:
        report = check.compare_result(fetched, check.exact_groups(batches, ts_range=ts_range, measure=sort_measure), hash_identity=False,
                                      ordered=True, measure_dtype={"count": "<u4", "sum": "<i8"}[args.sort_path])
    else:
        report = check.compare_result(fetched, check.exact_groups(batches, ts_range=ts_range), hash_identity=True)
    report["kernels"] = kernels
    report.update({"rows": int(args.rows), "batch_rows": int(args.batch_rows), "result_sizes": sizes,
                   "fused_batches": ctx.fused_batches,
                   "env": {k: v for k, v in os.environ.items() if k.startswith("ARES_")}})
    ctx.release()
    for s in streams:
        be.call("DestroyCudaStream", s, 0)
    print(json.dumps(report), flush=True)
    sys.exit(0 if report["status"] == "ok" else 1)


if __name__ == "__main__":
    main()
