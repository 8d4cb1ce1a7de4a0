#!/usr/bin/env python3
"""Contract benchmark: BASELINE config C3 — a 1 B-row filter -> 4-dimension / 1-measure
time-bucketised SUM group-by through the hash-reduction path — driven through the C ABI of the
HIP libalgorithm.so / libmem.so exactly as the Go batch executor would
(query/aql_batchexecutor.go:103-273), with the fact-table shard already resident in HBM.

One "step" = one full pass of the query over the rank's shard (all batches, reduce included; for
N > 1 also the cross-device merge of the per-shard group tables).  N ranks = N shards (weak
scaling): `value` = rows of all shards / max-over-ranks step time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--batch-rows B]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from aresdb_amd import abi, workload  # noqa: E402
from aresdb_amd.driver import NativeQuery  # noqa: E402
from aresdb_amd.queries import c3_plan  # noqa: E402
from aresdb_amd.workload import C3_COLUMNS  # noqa: E402

COLUMN_NAMES = [name for name, _ in C3_COLUMNS]

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a float4 copy)


def run_shard(be, plan, batches, device, stream):
    """ProcessQuery for one shard (query/aql_processor.go:49-161): every batch through
    preExec/filter/join/project/reduce/postExec; results accumulate on the device."""
    ctx = NativeQuery(be, plan, COLUMN_NAMES, device=device, stream=stream)  # the C++ host driver
    for b in batches:
        ctx.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
    return ctx


def expected_total(batches):
    """sum(m) over rows passing `d1 < 90` with d1 and m valid — float64, any order is exact because
    the synthetic measures are multiples of 0.25."""
    tot = torch.zeros((), dtype=torch.float64, device=batches[0]["m"].blob.device)
    rows = 0
    for b in batches:
        keep = b["d1"].values() < 90
        if b["d1"].has_nulls:
            keep &= b["d1"].valid()
        rows += int(keep.sum())
        mv = b["m"].values().to(torch.float64)
        if b["m"].has_nulls:
            mv = mv * b["m"].valid()
        tot += (mv * keep).sum()
    return float(tot), rows


def result_total(ctx, dims_ptr=None):
    n = ctx.result_size
    out = torch.empty(max(n, 1), dtype=torch.float64, device=f"cuda:{ctx.device}")
    ctx.be.call("AsyncCopyDeviceToDevice", out.data_ptr(), ctx.measure_vector, n * 8, ctx.stream, ctx.device)
    ctx.be.wait(ctx.stream, ctx.device)
    return float(out[:n].sum())


def cpu_baseline(batch, plan_factory, budget_s=15.0):
    """The reference's own sources in QUERY_MODE=HOST (oracle/_ref, kind "reference") or, when that
    build is absent, the C restatement (kind "port"), single-threaded like thrust::host
    (query/utils.hpp:236-241), on a bounded sample of the same workload.  HOST HashReduce's
    extraction is O(groups^2) (query/concurrent_unordered_map.hpp:154-159), so the sample goes
    through the HOST Sort+Reduce path."""
    from aresdb_amd.columns import DeviceColumn
    ref_algo = os.path.join(ROOT, "oracle", "_ref", "libalgorithm.so")
    ref_mem = os.path.join(ROOT, "oracle", "_ref", "libmem.so")
    port = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if os.path.exists(ref_algo) and os.path.exists(ref_mem):
        be, kind = abi.Backend("ref", ref_algo, ref_mem, device_memory=False), "reference"
    elif os.path.exists(port):
        be, kind = abi.Backend("oracle", port, port, device_memory=False), "port"
    else:
        return None
    chunk = 1 << 21  # the reference's example live-batch size (examples/1k_trips/schema/trips.json:46)
    cols, valid = workload.batch_to_host(batch, limit=32 * chunk)
    total_rows = len(next(iter(cols.values()))[1])
    plan = plan_factory(use_hash_reduction=False)
    ctx = NativeQuery(be, plan, COLUMN_NAMES)
    spent, rows, nb = 0.0, 0, 0
    for start in range(0, total_rows, chunk):
        n = min(chunk, total_rows - start)
        dev = {k: DeviceColumn(be, t, v[start:start + n], valid=None if valid[k] is None else valid[k][start:start + n])
               for k, (t, v) in cols.items()}
        t0 = time.perf_counter()
        ctx.run({k: d.vp for k, d in dev.items()}, n)
        spent += time.perf_counter() - t0
        for d in dev.values():
            d.free()
        rows += n
        nb += 1
        if spent > budget_s:
            break
    groups = ctx.result_size
    ctx.release()
    return {"value": rows / spent, "unit": "rows/s", "cores": 1, "kind": kind,
            "sample": f"{rows} rows of the same C3 shard as {nb} live batches of {chunk} rows, "
                      f"QUERY_MODE=HOST filter+transforms+Sort+Reduce, {groups} groups, {spent:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=float, default=1e9, help="rows per GPU (shard size)")
    ap.add_argument("--batch-rows", type=float, default=float(1 << 26))
    ap.add_argument("--null-fraction", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-leg", action="store_true", help="skip the extra AresFusedFilterHashReduce measurement")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    tdev = torch.device(f"cuda:{local_rank}")
    # ARES_BENCH_FORCE_DIST=1: exercise the multi-rank code path (RCCL init, merge) with one rank
    force_dist = os.environ.get("ARES_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
            os.environ["NCCL_DEBUG"] = "WARN"  # keep RCCL's version banner out of stdout (one JSON line)
        dist.init_process_group("nccl", device_id=tdev)

    be = abi.load_hip_backend()  # raises if the HIP libraries are missing — no fallback
    if world == 1:
        be.call("BootstrapDevice")  # touches every visible device (reference utils.cu:63-85): one process per GPU skips it
    stream = be.call("CreateCudaStream", local_rank)

    rows, batch_rows = int(args.rows), int(args.batch_rows)
    batches = workload.c3_shard(rows, batch_rows, seed=1 + rank, device=tdev, null_fraction=args.null_fraction)
    torch.cuda.synchronize()
    plan = c3_plan(use_hash_reduction=True)

    def step():
        ctx = run_shard(be, plan, batches, local_rank, stream)
        merged = None
        if world > 1 or force_dist:
            from aresdb_amd.shard_merge import merge_shard_results
            merged = merge_shard_results(ctx, tdev)
        return ctx, merged

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()

    for _ in range(args.warmup):
        ctx, _ = step()
        ctx.release()
    sync()
    if be.has_profiler:
        be.profiler_enable(True)
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        if last is not None:
            last[0].release()
        last = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if be.has_profiler:
        kernels = be.profiler_report()
        be.profiler_enable(False)
    else:
        kernels = {}
    if world > 1 or force_dist:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    ctx, merged = last
    # outside the timed region: the aggregate must account for every surviving row
    want, kept = expected_total(batches)
    got = result_total(ctx)
    groups = ctx.result_size
    check = abs(got - want) <= 1e-9 * max(1.0, abs(want))
    merged_groups = merged.size if merged is not None else None
    ctx.release()

    bytes_per_row = 5 * 4 + (5 / 8 if args.null_fraction > 0 else 0)
    total_rows_rank = rows * args.steps
    dominant = None
    kern_out = {}
    if kernels:
        tot_ms = sum(v[1] for v in kernels.values())
        for name, (launches, ms) in sorted(kernels.items(), key=lambda kv: -kv[1][1]):
            kern_out[name] = {"launches": launches, "avg_ms": ms / launches, "share": ms / tot_ms}
        dominant = max(kernels.items(), key=lambda kv: kv[1][1])
    # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, corrected
    # as MI355X_MICROARCH.md prescribes), when they were taken at this batch size
    pmc_kernels = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pmc.get("rows_per_batch") == batch_rows:
            pmc_kernels = pmc["kernels"]
    except (OSError, ValueError, KeyError):
        pass
    for name, k in kern_out.items():  # measured traffic rate of every kernel (not the algorithmic roofline)
        base = name.split("<")[0]
        if base in pmc_kernels:
            k["hbm_bytes_per_launch"] = pmc_kernels[base]["hbm_bytes_per_launch"]
            k["hbm_GBps"] = k["hbm_bytes_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9
    roofline = None
    if dominant:
        name, (launches, ms) = dominant
        traffic = pmc_kernels.get(name.split("<")[0], {}).get("hbm_bytes_per_launch")
        achieved = bytes_per_row * total_rows_rank / (ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": bytes_per_row * total_rows_rank / launches,
                    "avg_launch_ms": ms / launches, "launches": launches,
                    "note": "algorithmic = 20 B/row of compulsory column reads (+5 validity bits/row), "
                            "SURVEY.md 8d; traffic from rocprofv3 PMC passes is in profiles/"}

    # Extra leg (reported separately, never `value`): the same query through the fused extension entry
    # point (include/ares_extensions.h) — one call per batch instead of the per-node ABI sequence.
    fused = None
    if not args.no_fused_leg and be.has_profiler:
        fplan = c3_plan(use_hash_reduction=True)
        fplan.use_fused_extension = True
        fctx = run_shard(be, fplan, batches, local_rank, stream)
        fctx.release()
        sync()
        be.profiler_enable(True)
        t1 = time.perf_counter()
        for k in range(args.steps):
            fctx = run_shard(be, fplan, batches, local_rank, stream)
            if k + 1 < args.steps:
                fctx.release()
        torch.cuda.synchronize()
        felapsed = time.perf_counter() - t1
        fk = be.profiler_report()
        be.profiler_enable(False)
        fgot = result_total(fctx)
        fused = {"rows_per_sec_per_gpu": rows * args.steps / felapsed, "ms_per_step": felapsed / args.steps * 1e3,
                 "fused_batches": fctx.fused_batches, "groups": fctx.result_size,
                 "check_sum_of_measures": "ok" if abs(fgot - want) <= 1e-9 * max(1.0, abs(want)) else f"MISMATCH {fgot} vs {want}",
                 "algorithmic_GBps": rows * args.steps / felapsed * bytes_per_row / 1e9,
                 "kernels": {n: {"launches": c, "avg_ms": ms / c} for n, (c, ms) in sorted(fk.items(), key=lambda kv: -kv[1][1])}}
        fctx.release()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(batches[0], c3_plan, args.cpu_budget)

    if rank == 0:
        value = rows * world * args.steps / elapsed
        out = {
            "metric": "rows/sec, 1B-row filter -> group-by-agg (whole job)", "value": value, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 keys / f64 sums", "data": "synthetic",
            "config": {"workload": "C3: filter d1<90 -> dims [floor(ts,3600), d1, d2, d3] (4 x uint32) -> "
                                   "SUM(m float32 -> float64) via HashReduce, validity bitmaps with "
                                   f"{args.null_fraction:.0%} nulls, shard resident in HBM",
                       "rows_per_gpu": rows, "batch_rows": batch_rows, "batches": len(batches),
                       "groups_per_shard": groups, "merged_groups": merged_groups, "rows_after_filter": kept,
                       "parallelism": f"{world} shard(s), one per GPU" + (", RCCL all_gather merge" if (world > 1 or force_dist) else "")},
            "rows_per_sec_per_gpu": value / world,
            "algorithmic_GBps_end_to_end": value / world * bytes_per_row / 1e9,
            "check_sum_of_measures": "ok" if check else f"MISMATCH got {got} want {want}",
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kern_out, "fused_extension": fused,
        }
        print(json.dumps(out), flush=True)
        if not check:
            sys.exit(1)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
