#!/usr/bin/env python3
"""Contract benchmark: BASELINE config C3 — a 1 B-row filter -> 4-dimension / 1-measure
time-bucketised SUM group-by through the hash-reduction path — driven through the C ABI of the
HIP libalgorithm.so / libmem.so exactly as the Go batch executor would
(query/aql_batchexecutor.go:103-273, two streams alternating per batch as query/aql_processor.go:218),
with the fact-table shard already resident in HBM.

One "step" = one full pass of the query over the rank's shard (all batches, reduce included; for
N > 1 also the cross-device merge of the per-shard group tables).  N ranks = N shards (weak
scaling): `value` = rows of all shards / max-over-ranks step time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--batch-rows B]

`--gpus N` with N > 1 and no torchrun environment spawns the N ranks itself (one process per GPU).
Outside the timed region the final group table is verified key by key (aresdb_amd/check.py), the
reference's own HOST build is timed on a bounded sample (cpu_baseline) and compared key by key with
the same independent group-by, and a few secondary legs are reported (never `value`): the same query
with the in-ABI fusion stages switched off, at the reference's live-batch size, and through the
explicit fused extension.
"""
import argparse
import gc
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from aresdb_amd import abi, check, workload  # noqa: E402
from aresdb_amd.driver import NativeQuery  # noqa: E402
from aresdb_amd.queries import c3_plan  # noqa: E402
from aresdb_amd.workload import C3_COLUMNS  # noqa: E402

COLUMN_NAMES = [name for name, _ in C3_COLUMNS]

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a float4 copy)
LIVE_BATCH_ROWS = 1 << 21  # the reference's example live-batch size (examples/1k_trips/schema/trips.json:46)


def run_shard(be, plan, vps, device, streams):
    """ProcessQuery for one shard (query/aql_processor.go:49-161): every batch through
    preExec/filter/join/project/reduce/postExec; results accumulate on the device."""
    ctx = NativeQuery(be, plan, COLUMN_NAMES, device=device, streams=streams)  # the C++ host driver
    packed = _PACKED.get(id(vps))
    if packed is None:  # the slices of the resident shard, marshalled once: a step is ONE call into the driver
        packed = _PACKED[id(vps)] = ctx.pack_batches(vps)
    ctx.run_batches(packed)
    return ctx


_PACKED = {}


def cpu_baseline(batches, plan_factory, budget_s=15.0, max_chunks=32):
    """The reference's own sources in QUERY_MODE=HOST (oracle/_ref, kind "reference") or, when that
    build is absent, the C restatement (kind "port"), single-threaded like thrust::host
    (query/utils.hpp:236-241), on a bounded sample of the same workload: live-batch-sized chunks taken round-robin from
    EVERY batch of the shard (chunk j comes from batch j mod #batches), so the sample — and the key-level comparison of the
    reference's result with the independent group-by — covers the whole shard, not its first batch.  HOST HashReduce's
    extraction is O(groups^2) (query/concurrent_unordered_map.hpp:154-159), so the sample goes
    through the HOST Sort+Reduce path.  Returns (report, fetched result, slices)."""
    from aresdb_amd.columns import DeviceColumn
    ref_algo = os.path.join(ROOT, "oracle", "_ref", "libalgorithm.so")
    ref_mem = os.path.join(ROOT, "oracle", "_ref", "libmem.so")
    port = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if os.path.exists(ref_algo) and os.path.exists(ref_mem):
        be, kind = abi.Backend("ref", ref_algo, ref_mem, device_memory=False), "reference"
    elif os.path.exists(port):
        be, kind = abi.Backend("oracle", port, port, device_memory=False), "port"
    else:
        return None, None, []
    chunk = LIVE_BATCH_ROWS
    nb = len(batches)
    plan = plan_factory(use_hash_reduction=False)
    ctx = NativeQuery(be, plan, COLUMN_NAMES)
    spent, rows, slices = 0.0, 0, []
    for j in range(max_chunks):
        bi, lo = j % nb, (j // nb) * chunk
        length = next(iter(batches[bi].values())).length
        if lo >= length:
            continue
        cols, valid = workload.batch_to_host(batches[bi], limit=chunk, lo=lo)
        n = len(next(iter(cols.values()))[1])
        dev = {k: DeviceColumn(be, t, v, valid=valid[k]) for k, (t, v) in cols.items()}
        t0 = time.perf_counter()
        ctx.run({k: d.vp for k, d in dev.items()}, n)
        spent += time.perf_counter() - t0
        for d in dev.values():
            d.free()
        rows += n
        slices.append((bi, lo, n))
        if spent > budget_s:
            break
    groups = ctx.result_size
    fetched = ctx.fetch()
    ctx.release()
    report = {"value": rows / spent, "unit": "rows/s", "cores": 1, "kind": kind,
              "sample": f"{rows} rows of the same C3 shard: {len(slices)} live batches of {chunk} rows taken round-robin from all "
                        f"{nb} batches, QUERY_MODE=HOST filter+transforms+Sort+Reduce, {groups} groups, {spent:.1f} s"}
    return report, fetched, slices


def host_batch_leg(be, plan, batches, device, streams, max_batches=4):
    """PCIe-inclusive secondary leg (never `value`): the first batches of the shard handed over as pinned
    host buffers — upload of batch k+1 overlapped with the execution of batch k, like the Go host
    (query/aql_processor.go:850-881) — then the same query again with the columns resident in the
    driver's HBM column cache."""
    from aresdb_amd.driver import ColumnCache, HostColumn
    take = batches[:max_batches]
    hb = []
    for b, cols in enumerate(take):
        hcs = [HostColumn.from_blob(be, cols[k].data_type, cols[k].blob.cpu().numpy(), cols[k].values_off, cols[k].length,
                                    cols[k].has_nulls, cache_key=100 * (b + 1) + i + 1) for i, k in enumerate(COLUMN_NAMES)]
        hb.append((hcs, cols[COLUMN_NAMES[0]].length))
    rows = sum(n for _, n in hb)
    total_bytes = sum(hc.nbytes for cols, _ in hb for hc in cols)
    cache = ColumnCache(be, device, budget_bytes=2 * total_bytes)
    out = {"rows": rows, "batches": len(hb)}
    for name, use_cache in (("pcie_inclusive", None), ("cache_fill", cache), ("cache_resident", cache)):
        ctx = NativeQuery(be, plan, COLUMN_NAMES, device=device, streams=streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = ctx.run_host_batches(hb, use_cache)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rep = check.compare_result(ctx.fetch(), check.exact_groups(take), hash_identity=True)
        ctx.release()
        out[name] = {"rows_per_sec": rows / dt, "ms": dt * 1e3, "uploaded_GB": stats["uploaded_bytes"] / 1e9,
                     "h2d_GBps": stats["uploaded_bytes"] / dt / 1e9, "cache_hits": stats["cache_hits"],
                     "check_groups": rep["status"]}
    cache.destroy()
    for cols, _ in hb:
        for hc in cols:
            hc.free()
    return out


def library_sha():
    h = hashlib.sha256()
    with open(abi.hip_library_paths()[0], "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def kernel_table(kernels):
    out = {}
    tot = sum(v[1] for v in kernels.values()) or 1.0
    for name, (launches, ms) in sorted(kernels.items(), key=lambda kv: -kv[1][1]):
        out[name] = {"launches": launches, "avg_ms": ms / launches, "share": ms / tot}
    return out


def dominant_roofline(kernels, bytes_per_row, rows_total):
    """Algorithmic bytes of `rows_total` rows over the summed duration of the kernel that took longest."""
    if not kernels:
        return None
    name, (launches, ms) = max(kernels.items(), key=lambda kv: kv[1][1])
    achieved = bytes_per_row * rows_total / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": bytes_per_row * rows_total / launches,
            "avg_launch_ms": ms / launches, "launches": launches}


def measure_traffic(batch_rows, timeout=240):
    """HBM bytes per launch of every kernel of the C3 hot path, measured now: two rocprofv3 --pmc passes
    (FETCH_SIZE, WRITE_SIZE: they do not fit one pass; kernel-trace only, never combined with API tracing) over
    tools/pmc_driver.py — three passes of the query over one resident batch of `batch_rows` rows.  The read side is
    doubled as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE tallies 64 B per 128-B request on wide
    coalesced reads).  Returns ({kernel: {...}}, note)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="ares_pmc_", dir="/tmp")
    acc = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_driver.py"), str(batch_rows), "3"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp", "ARES_RTC_ASYNC": "0"},
                                   capture_output=True, text=True, timeout=timeout)
            except subprocess.TimeoutExpired:
                return {}, f"rocprofv3 --pmc {counter}: timeout"
            if r.returncode != 0:
                return {}, f"rocprofv3 --pmc {counter}: rc {r.returncode}: {r.stderr[-200:]}"
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    if "ares::" not in name and "_rtc" not in name:
                        continue
                    short = (name.split("ares::")[1] if "ares::" in name else name).replace("(anonymous namespace)::", "")
                    short = short.split("(")[0].split("<")[0]
                    if row.get("Counter_Name") != counter:
                        continue
                    a = acc.setdefault(short, {})
                    a[counter] = a.get(counter, 0.0) + float(row["Counter_Value"])
                    a[counter + "_n"] = a.get(counter + "_n", 0) + 1
    finally:
        shutil.rmtree(out, ignore_errors=True)
    kernels = {}
    for k, a in acc.items():
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            f = a["FETCH_SIZE"] / a["FETCH_SIZE_n"] * 1024.0
            w = a["WRITE_SIZE"] / a["WRITE_SIZE_n"] * 1024.0
            kernels[k] = {"fetch_bytes_per_launch": 2 * f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": 2 * f + w,
                          "launches_profiled": a["FETCH_SIZE_n"]}
    return kernels, ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes taken inside this run (tools/pmc_driver.py, "
                     f"{batch_rows} rows per batch; read side x2: gfx950 counts 64 B per 128-B request)")


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: one process per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE from torch.distributed.run), rendezvous on 127.0.0.1."""
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < n:
        raise SystemExit(f"bench.py --gpus {n}: only {visible} GPU(s) visible — refusing to report a smaller job as n_gpus={n}")
    port = os.environ.get("MASTER_PORT", "29517")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__), *argv]
    raise SystemExit(subprocess.call(cmd, env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}))


def run_leg(env, argv, timeout=900):
    """A secondary measurement in a child process (the fusion switches are read once per process):
    this script with --leg, printing one JSON line."""
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", *argv]
    try:
        r = subprocess.run(cmd, env={**os.environ, **env}, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"rc {r.returncode}", "stderr": r.stderr[-400:]}
    return json.loads(lines[-1])


def single_process(args, backend=None, tensor_device=None):
    """--single-process: one thread per shard / GPU inside this process; every thread runs its shard through the ABI on
    its own device and streams, then all of them merge the per-shard tables inside libaresdriver.so (blocks copied peer
    to peer with hipMemcpyAsync).  Prints the same JSON line as the contract launch (parallelism says which it is)."""
    import threading
    from aresdb_amd.driver import NativeComm
    on_gpu = backend is None
    n = max(1, args.gpus)
    if on_gpu and (not torch.cuda.is_available() or torch.cuda.device_count() < n):
        raise SystemExit(f"bench.py --single-process --gpus {n}: needs {n} visible GPU(s)")
    be = backend if backend is not None else abi.load_hip_backend()
    if on_gpu:
        be.call("BootstrapDevice")
    rows, batch_rows = int(args.rows), int(args.batch_rows)
    dims = tuple(d for d in args.dims.split(",") if d)
    ts_range = tuple(int(x) for x in args.ts_range.split(",")) if args.ts_range else None
    plan = c3_plan(use_hash_reduction=True, dims=dims, d1_below=args.d1_below, ts_range=ts_range)
    comms = NativeComm.local(n, on_gpu)
    barrier = threading.Barrier(n)
    state = {"elapsed": [0.0] * n, "merged": [0] * n, "errors": [], "checks": [None] * n}

    def work(r):
        try:
            dev = r if on_gpu else 0
            tdev = torch.device(f"cuda:{r}") if on_gpu else torch.device(tensor_device or "cpu")
            if on_gpu:
                torch.cuda.set_device(r)
            streams = [be.call("CreateCudaStream", dev) for _ in range(2 if on_gpu else 1)]
            batches = workload.c3_shard(rows, batch_rows, seed=1 + r, device=tdev, null_fraction=args.null_fraction)
            vps = [({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length) for b in batches]

            def sync():
                if on_gpu:
                    torch.cuda.synchronize(tdev)
                barrier.wait()
            ctx = run_shard(be, plan, vps, dev, streams)  # priming pass (kernels of the shape get compiled)
            ctx.merge_shards(comms[r])
            ctx.release()
            be.rtc_wait()
            for _ in range(args.warmup):
                ctx = run_shard(be, plan, vps, dev, streams)
                ctx.merge_shards(comms[r])
                ctx.release()
            sync()
            t0 = time.perf_counter()
            last = None
            for _ in range(args.steps):
                if last is not None:
                    last.release()
                last = run_shard(be, plan, vps, dev, streams)
                last.merge_shards(comms[r])
            sync()
            state["elapsed"][r] = time.perf_counter() - t0
            state["merged"][r] = last.result_size
            if args.verify_merged and r == 0:
                every = []
                for q in range(n):
                    every += workload.c3_shard(rows, batch_rows, seed=1 + q, device=tdev, null_fraction=args.null_fraction)
                state["checks"][0] = check.compare_result(last.fetch(), check.exact_groups(every, dims=dims, d1_below=args.d1_below, ts_range=ts_range),
                                                          hash_identity=True, dims=dims)
            last.release()
            for s_ in streams:
                be.call("DestroyCudaStream", s_, dev)
        except Exception as e:  # noqa: BLE001
            import traceback
            state["errors"].append((r, f"{type(e).__name__}: {e}", traceback.format_exc()[-800:]))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass
    threads = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in comms:
        c.destroy()
    if state["errors"]:
        print(f"bench.py --single-process: {state['errors']}", file=sys.stderr)
        sys.exit(1)
    elapsed = max(state["elapsed"])
    value = rows * n * args.steps / elapsed
    ok = state["checks"][0] is None or state["checks"][0]["status"] == "ok"
    print(json.dumps({
        "metric": "rows/sec, 1B-row filter -> group-by-agg (whole job)", "value": value, "unit": "rows/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32 keys / f64 sums", "data": "synthetic",
        "config": {"workload": "C3 (see the contract launch)", "rows_per_gpu": rows, "batch_rows": batch_rows,
                   "merged_groups": state["merged"][0],
                   "parallelism": f"{n} shard(s), one THREAD per GPU in one process (query/device_manager.go:185-218), "
                                  "peer-to-peer all-gather + re-reduce merge in libaresdriver.so"},
        "rows_per_sec_per_gpu": value / n, "check_merged_groups": state["checks"][0],
        "per_rank_ms_per_step": [e / args.steps * 1e3 for e in state["elapsed"]]}), flush=True)
    if not ok:
        sys.exit(1)
    return 0


HEADLINE_MAX_BYTES = 4096  # the driver keeps a bounded line: round 5's 22 KB line came back unparsed (BENCH_r05.json)


def _round_floats(x, digits=6):
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _round_floats(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round_floats(v, digits) for v in x]
    return x


def _brief_check(rep):
    if not isinstance(rep, dict):
        return rep
    return {k: rep[k] for k in ("status", "groups", "rows", "batches_sampled") if k in rep}


def headline(out):
    """The compact record the driver parses: contract keys + roofline + cpu_baseline + a short summary of the legs.
    Everything else (per-kernel table, legs, per-step times) goes to bench_full.json (see emit)."""
    cfg = out.get("config") or {}
    roof = out.get("roofline")
    cpu = out.get("cpu_baseline")
    summ = dict(out.get("summary") or {})
    for k in ("value_rows_per_s", "ms_per_step", "check_groups", "dominant_kernel", "traffic_over_algorithmic",
              "cpu_baseline_rows_per_s", "cpu_baseline_kind", "cpu_baseline_sample", "roofline_frac_dominant_kernel"):
        summ.pop(k, None)  # already in the line proper
    line = {
        "metric": out["metric"], "value": out["value"], "unit": out["unit"], "n_gpus": out["n_gpus"],
        "steps": out["steps"], "warmup": out["warmup"], "ms_per_step": out["ms_per_step"],
        "higher_is_better": True, "scaling": out["scaling"], "vs_baseline": out["vs_baseline"],
        "dtype": out["dtype"], "data": out["data"],
        "config": {k: cfg.get(k) for k in ("workload", "rows_per_gpu", "batch_rows", "batches", "streams_per_query",
                                           "groups_per_shard", "merged_groups", "merge_transport", "parallelism")},
        "roofline": None if roof is None else {k: roof.get(k) for k in (
            "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
            "algorithmic_bytes_per_launch", "avg_launch_ms", "launches")},
        "cpu_baseline": None if cpu is None else {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample")},
        "check_groups": _brief_check(out.get("check_groups")), "check_merged_groups": _brief_check(out.get("check_merged_groups")),
        "reference_host_check": _brief_check(out.get("reference_host_check")),
        "per_rank_ms_per_step": out.get("per_rank_ms_per_step"),
        "median_ms_per_step": out.get("median_ms_per_step"), "max_ms_per_step": out.get("max_ms_per_step"),
        "summary": summ, "full_record": out.get("full_record"),
    }
    line = _round_floats(line)
    if line["cpu_baseline"] and isinstance(line["cpu_baseline"].get("sample"), str):
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:200]
    # never outgrow the bound: drop the optional parts, least important first
    for victim in ("legs_dominant_kernel_frac", "trips_shaped", "legs_ms_per_1B_rows", None):
        if len(json.dumps(line)) < HEADLINE_MAX_BYTES:
            break
        if victim is None:
            line["summary"] = {}
        else:
            line["summary"].pop(victim, None)
    return line


def emit(out, full_line=False):
    """Rank 0's output.  A secondary leg (--leg) prints its whole record for the parent to read; the contract run writes
    the whole record to bench_full.json (gpurun_out/ when that exists, so it comes back from the GPU box, and the
    working directory) and prints ONE compact JSON line, the last line of stdout."""
    if full_line:
        print(json.dumps(out), flush=True)
        return
    written = []
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            if d != ROOT and not os.path.isdir(d):
                continue
            path = os.path.join(d, "bench_full.json")
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            written.append(os.path.relpath(path, ROOT))
        except OSError:
            pass
    out["full_record"] = written[0] if written else None
    text = json.dumps(headline(out))
    assert len(text) < HEADLINE_MAX_BYTES, len(text)
    print(text, flush=True)


def main(argv=None, backend=None, tensor_device=None):
    """backend / tensor_device: injected by the multi-process CPU test of this file's distributed
    path (tests/test_bench_distributed.py); the benchmark itself always loads the HIP libraries and
    fails without a GPU — there is no CPU fallback."""
    argv = sys.argv[1:] if argv is None else argv
    t_process0 = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=float, default=1e9, help="rows per GPU (shard size)")
    ap.add_argument("--batch-rows", type=float, default=float(1 << 26))
    ap.add_argument("--null-fraction", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary legs (fusion off, live batches, extension)")
    ap.add_argument("--legs", default="all", help="comma list of substrings: only the secondary legs whose name contains one of them")
    ap.add_argument("--leg-budget", type=float, default=240.0,
                    help="seconds for the secondary legs (0 = no limit): they run in the order below until the budget is spent, "
                         "the rest is marked skipped; a leg that needs more than what is left is not started")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--one-stream", action="store_true", help="every batch on one stream (the Go host alternates two)")
    ap.add_argument("--verify-merged", action="store_true",
                    help="N > 1: rank 0 regenerates every shard and checks the merged table key by key (small sizes)")
    ap.add_argument("--dims", default="ts,d1,d2,d3",
                    help="group-by dimensions, a subset of C3's four (lower-cardinality variants of the same query: secondary legs)")
    ap.add_argument("--d1-below", type=int, default=90, help="constant of the filter d1 < K")
    ap.add_argument("--ts-range", default="", help="from,to: the Go host's two time filters (ts >= from, ts < to) in front of the "
                                                   "query's own filter — the shape every fact-table query has (secondary leg)")
    ap.add_argument("--sort-path", default="", choices=["", "count", "sum"],
                    help="the same group-by through the reference's DEFAULT aggregation path, Sort + Reduce (config/ares.yaml:11 "
                         "enable_hash_reduction: false): count = COUNT(*), sum = SUM(d2) as AGGR_SUM_UNSIGNED into 8 bytes (secondary legs)")
    ap.add_argument("--eight-dims", action="store_true",
                    help="MAX_DIMENSIONS: four more dimensions (functions of the first four: the same groups) — the fused path at the ABI's limit (secondary leg)")
    ap.add_argument("--archive", action="store_true",
                    help="archive batches: rows sorted by (ts, d3), both run-length encoded (mode 3) — decoded once per batch inside the "
                         "ABI, everything else on the same fast path (secondary leg)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--single-process", action="store_true",
                    help="N shards on N GPUs as N threads of THIS process (the reference's process model, "
                         "query/device_manager.go:185-218), merged through the in-process communicator: not the contract "
                         "launch (one rank per GPU under torch.distributed.run), a second way to run the same job")
    ap.add_argument("--cold", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--fused-extension", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)

    if args.single_process:
        return single_process(args, backend, tensor_device)
    # Cold-start legs first, while this process has not touched the GPU yet: a fresh process alone on the device — what a
    # restarted server is.  (The 0.3-0.7 s stalls these legs showed in rounds 3 and 4 were one driver allocation: the
    # DIRECT-mode workspace was sized for a region it does not write, 4.3 GB of fresh device memory in a query's second
    # or third batch — profiles/r4_experiments.md "cold start".)
    early_legs = {}
    if backend is None and args.gpus <= 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not (args.leg or args.cold or args.no_legs) and \
            (args.legs == "all" or any("cold" in w for w in args.legs.split(","))):
        import tempfile
        cold_argv = ["--rows", str(int(args.rows)), "--null-fraction", str(args.null_fraction), "--steps", "3", "--warmup", "1",
                     "--batch-rows", str(int(args.batch_rows)), "--cold"]
        with tempfile.TemporaryDirectory(prefix="ares_rtc_cache_") as tmp:
            for name in ("cold_process", "cold_process_warm_disk_cache"):  # empty on-disk cache, then a second process that finds it filled
                t0 = time.perf_counter()
                early_legs[name] = run_leg({"ARES_RTC_CACHE_DIR": tmp}, cold_argv)
                if isinstance(early_legs[name], dict):
                    early_legs[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)


    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and backend is None:
        spawn_ranks(args.gpus, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started {world} rank(s)")
    on_gpu = backend is None
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if on_gpu:
        torch.cuda.set_device(local_rank)
        tdev = torch.device(f"cuda:{local_rank}")
    else:
        tdev = torch.device(tensor_device or "cpu")
    # ARES_BENCH_FORCE_DIST=1: exercise the multi-rank code path (RCCL init, merge) with one rank
    force_dist = os.environ.get("ARES_BENCH_FORCE_DIST") == "1"
    distributed = world > 1 or force_dist
    created_group = False
    if distributed and not dist.is_initialized():
        created_group = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
            os.environ["NCCL_DEBUG"] = "WARN"  # keep RCCL's version banner out of stdout (one JSON line)
        if on_gpu:
            dist.init_process_group("nccl", device_id=tdev)
        else:
            dist.init_process_group("gloo")
    if distributed and dist.get_world_size() != world:
        raise SystemExit(f"the process group has {dist.get_world_size()} ranks, expected {world}")

    be = backend if backend is not None else abi.load_hip_backend()  # raises if the HIP libraries are missing
    device_index = local_rank if on_gpu else 0
    if on_gpu and world == 1:
        be.call("BootstrapDevice")  # touches every visible device (reference utils.cu:63-85): one process per GPU skips it
    n_streams = 1 if args.one_stream or not on_gpu else 2
    streams = [be.call("CreateCudaStream", device_index) for _ in range(n_streams)]

    rows, batch_rows = int(args.rows), int(args.batch_rows)
    batches = workload.c3_shard(rows, batch_rows, seed=1 + rank, device=tdev, null_fraction=args.null_fraction, archive=args.archive)
    vps = [({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length) for b in batches]
    dims = tuple(d for d in args.dims.split(",") if d)
    assert dims and all(d in check.ALL_DIMS for d in dims), args.dims
    ts_range = tuple(int(x) for x in args.ts_range.split(",")) if args.ts_range else None
    sort_measure = {"": None, "count": "count", "sum": "d2"}[args.sort_path]
    plan = c3_plan(use_hash_reduction=True, dims=dims, d1_below=args.d1_below, ts_range=ts_range, sort_measure=sort_measure,
                   eight_dims=args.eight_dims)
    plan.use_fused_extension = bool(args.fused_extension)
    # columns the plan reads (dimensions + measure + the filter's d1 [+ ts of the time filters]): the algorithmic bytes per row
    measure_columns = {"": {"m"}, "count": set(), "sum": {"d2"}}[args.sort_path]
    plan_columns = sorted(set(dims) | measure_columns | {"d1"} | ({"ts"} if ts_range else set()))

    # the independent group-by a result is compared with, and the comparison: HashReduce identifies groups by their 32-bit
    # hash; Sort + Reduce by the 64-bit one (exact groups at these cardinalities) and leaves them in ascending hash order
    def expected_groups(bs, d1_below=args.d1_below, **kw):
        return check.exact_groups(bs, dims=dims, d1_below=d1_below, ts_range=ts_range, measure=(sort_measure or "m"), **kw)

    def compare_fetched(fetched, expected):
        if not args.sort_path:
            return check.compare_result(fetched, expected, hash_identity=True, dims=dims, eight=args.eight_dims)
        return check.compare_result(fetched, expected, hash_identity=False, dims=dims, ordered=True, eight=args.eight_dims,
                                    measure_dtype={"count": "<u4", "sum": "<i8"}[args.sort_path])

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    # N > 1: the per-shard group tables meet in ONE exchange inside libaresdriver.so — all-gather of the
    # padded columnar partials (RCCL on the query's stream over xGMI; gloo in the CPU test of this file)
    # and a re-reduce with the library's own HashReduce; aresdb_amd/shard_merge.py is its test mirror
    comm, merge_transport = None, None
    if distributed:
        from aresdb_amd.driver import NativeComm
        if on_gpu:
            def bcast(raw):
                t = torch.tensor(list(raw), dtype=torch.uint8, device=tdev)
                dist.broadcast(t, 0)
                return bytes(t.cpu().tolist())
            host_group = dist.new_group(backend="gloo")  # collective: made whether or not it is needed
            why = ""
            try:
                if os.environ.get("ARES_BENCH_NO_RCCL") == "1":  # exercise the fallback
                    raise RuntimeError("ARES_BENCH_NO_RCCL=1")
                comm = NativeComm.rccl(rank, world, device_index, bcast)
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
            everyone = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=tdev)
            dist.all_reduce(everyone, op=dist.ReduceOp.MIN)
            merge_transport = "ncclAllGather on the query stream (librccl bound by libaresdriver.so)"
            if int(everyone.item()) == 0:  # some rank could not bind RCCL: the same merge over host-staged gloo
                if comm is not None:
                    comm.destroy()
                print(f"[rank {rank}] RCCL communicator unavailable ({why or 'on another rank'}): shard merge staged through "
                      "host memory", file=sys.stderr, flush=True)
                comm = NativeComm.torch_group(host_group, device_backend=be, device=device_index)
                merge_transport = "gloo all-gather staged through host memory (RCCL could not be bound)"
        else:
            comm = NativeComm.torch_group()
            merge_transport = "gloo all-gather (CPU test)"

    def merge(ctx):
        ctx.merge_shards(comm)
        return ctx.result_size

    if args.cold:  # secondary leg (own process): the very first query of a process, then the same shape with another constant
        cold = {}

        def run_timed(pl):  # a query, batch by batch: where a slow first query spends its time
            per = []
            ctx_ = NativeQuery(be, pl, COLUMN_NAMES, device=device_index, streams=streams)
            for cols_, n_ in vps:
                tb = time.perf_counter()
                ctx_.run(cols_, n_)
                per.append(round((time.perf_counter() - tb) * 1e3, 3))
            return ctx_, per
        sync()
        t0 = time.perf_counter()
        ctx, cold["cold_first_query_batch_ms"] = run_timed(plan)
        sync()
        cold["cold_first_query_ms"] = (time.perf_counter() - t0) * 1e3
        rep = compare_fetched(ctx.fetch(), expected_groups(batches))
        cold["cold_check_groups"] = rep["status"]
        ctx.release()
        cold["rtc_after_cold_query"] = be.rtc_wait()  # (waits for the kernels the first query asked for)
        for _ in range(2):
            ctx = run_shard(be, plan, vps, device_index, streams)
            ctx.release()
        sync()
        t0 = time.perf_counter()
        ctx = run_shard(be, plan, vps, device_index, streams)
        sync()
        cold["warm_query_ms"] = (time.perf_counter() - t0) * 1e3
        ctx.release()
        plan2 = c3_plan(use_hash_reduction=True, dims=dims, d1_below=args.d1_below - 7, ts_range=ts_range)  # another constant, never seen
        sync()
        t0 = time.perf_counter()
        ctx, cold["new_constants_batch_ms"] = run_timed(plan2)
        sync()
        cold["new_constants_query_ms"] = (time.perf_counter() - t0) * 1e3
        rep = check.compare_result(ctx.fetch(), check.exact_groups(batches, dims=dims, d1_below=args.d1_below - 7, ts_range=ts_range), hash_identity=True, dims=dims)
        cold["new_constants_check_groups"] = rep["status"]
        cold["rtc_after_new_constants"] = be.rtc_wait()
        ctx.release()
        print(json.dumps(cold), flush=True)
        sys.exit(0 if cold["cold_check_groups"] == "ok" and cold["new_constants_check_groups"] == "ok" else 1)

    # The scan / merge kernels of this query shape are compiled in the background the first time the shape is seen
    # (a query never waits for hiprtc: it runs the generic kernels meanwhile).  One untimed priming pass, then wait
    # for the compiler — what a server's first query of the shape does for every later one.
    rtc_state, compiles = None, -1
    for _ in range(4):  # (a query's first batch, its later batches and the merges that start from a table image are different
        ctx = run_shard(be, plan, vps, device_index, streams)  # kernels, each requested when the query first gets there)
        ctx.release()
        rtc_state = be.rtc_wait() if on_gpu else None
        if rtc_state is None or rtc_state["compiles"] == compiles:
            break
        compiles = rtc_state["compiles"]
    # ---- the timed region: W warm-up steps, then exactly K steps between barrier + synchronize ----
    for _ in range(args.warmup):
        ctx = run_shard(be, plan, vps, device_index, streams)
        if distributed:
            merge(ctx)
        ctx.release()
    sync()
    # A step is one call into the C++ driver (AresQueryRunResidentBatches: the batch loop of ProcessQuery): the
    # interpreter allocates a few dozen objects per step, so its garbage collector — whose generation-2 pass over
    # torch's ~10^6 objects took 30-50 ms and landed in one step out of ~20 in round 2 — has no reason to run inside the
    # timed region, and is left alone (round 3 switched it off there).
    gc.collect()
    drv0 = be.mem_driver_calls(device_index)
    shard_s = merge_s = 0.0
    t0 = time.perf_counter()
    last = merged = None
    step_ms = []
    for _ in range(args.steps):
        if last is not None:
            last.release()
        t1 = time.perf_counter()
        last = run_shard(be, plan, vps, device_index, streams)
        step_ms.append((time.perf_counter() - t1) * 1e3)  # host view of the step (it ends with a blocking read-back)
        if distributed:
            if on_gpu:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            merged = merge(last)
            if on_gpu:
                torch.cuda.synchronize()
            shard_s += t2 - t1
            merge_s += time.perf_counter() - t2
    sync()
    elapsed = time.perf_counter() - t0
    drv1 = be.mem_driver_calls(device_index)
    # ---- per-kernel durations: the same steps once more with HIP events around every launch (on the launch's own
    # stream, inside the library) — outside the timed region, so that the events cost `value` nothing; the pass's
    # own wall time is reported next to ms_per_step
    kernels, profiled_ms, prof_steps = {}, None, 0
    if be.has_profiler:
        prof_steps = max(1, min(args.steps, 5))
        sync()
        be.profiler_enable(True)
        tp = time.perf_counter()
        for _ in range(prof_steps):
            c = run_shard(be, plan, vps, device_index, streams)
            c.release()
        sync()
        profiled_ms = (time.perf_counter() - tp) / prof_steps * 1e3
        kernels = be.profiler_report()
        be.profiler_enable(False)
    rank_times = None
    if distributed:
        dist.barrier()
        t = torch.tensor([elapsed, shard_s, merge_s], dtype=torch.float64, device=tdev)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        rank_times = [[float(x) / args.steps * 1e3 for x in g] for g in gathered]
        elapsed = max(float(g[0]) for g in gathered)
    ctx = last

    # ---- outside the timed region: key-level verification of this rank's final group table ----
    merged_groups = merged_check = None
    if distributed:  # `ctx` holds the merged table of all shards: check it, then redo the shard alone
        merged_groups = int(merged)
        if args.verify_merged and rank == 0:
            every = []
            for r in range(world):
                every += workload.c3_shard(rows, batch_rows, seed=1 + r, device=tdev, null_fraction=args.null_fraction)
            merged_check = compare_fetched(ctx.fetch(), expected_groups(every))
        ctx.release()
        ctx = run_shard(be, plan, vps, device_index, streams)
    report = compare_fetched(ctx.fetch(), expected_groups(batches))
    groups = ctx.result_size
    ok = report["status"] == "ok" and report["groups"] == groups
    if merged_check is not None:
        ok = ok and merged_check["status"] == "ok" and merged_check["groups"] == merged_groups
    ctx.release()

    bytes_per_row = len(plan_columns) * (4 + (1 / 8 if args.null_fraction > 0 else 0))
    total_rows_rank = rows * max(prof_steps, 1)  # rows the profiled pass (the `kernels` table) went over
    if args.leg:  # a secondary leg: the measurement and its check, nothing else
        print(json.dumps({"rows_per_sec_per_gpu": rows * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                          "batches": len(vps), "groups": groups, "check_groups": report["status"],
                          "algorithmic_GBps": rows * args.steps / elapsed * bytes_per_row / 1e9,
                          "columns_read": plan_columns, "algorithmic_bytes_per_row": bytes_per_row,
                          "roofline": dominant_roofline(kernels, bytes_per_row, total_rows_rank),
                          "kernels": {n: {"launches": c, "avg_ms": ms / c} for n, (c, ms) in
                                      sorted(kernels.items(), key=lambda kv: -kv[1][1])}}), flush=True)
        sys.exit(0 if ok else 1)

    kern_out = kernel_table(kernels)
    # HBM bytes per launch: measured now by two rocprofv3 --pmc passes (rank 0 of a 1-GPU run), else — same build of
    # libalgorithm.so and same batch size only — from the committed passes under profiles/, else null
    pmc_kernels, pmc_note = {}, "not measured (--no-pmc)"
    if rank == 0 and world == 1 and on_gpu and not args.no_pmc and not args.leg:
        t_pmc0 = time.perf_counter()
        pmc_kernels, pmc_note = measure_traffic(batch_rows)
        pmc_note += f" ({time.perf_counter() - t_pmc0:.0f} s)"
    if not pmc_kernels:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r6_c3_pmc_traffic.json")))
            if pmc.get("rows_per_batch") == batch_rows and pmc.get("libalgorithm_sha256_16") == library_sha() and on_gpu:
                pmc_kernels = pmc["kernels"]
                pmc_note += f"; taken from profiles/r6_c3_pmc_traffic.json (build {pmc['libalgorithm_sha256_16']})"
        except (OSError, ValueError, KeyError):
            pass
    for name, k in kern_out.items():  # measured traffic rate of every kernel (not the algorithmic roofline)
        base = name.split("<")[0]
        if base in pmc_kernels:
            k["hbm_bytes_per_launch"] = pmc_kernels[base]["hbm_bytes_per_launch"]
            k["hbm_GBps"] = k["hbm_bytes_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9
    roofline = chain = None
    if kernels:
        roofline = dominant_roofline(kernels, bytes_per_row, total_rows_rank)
        base = roofline["kernel"].split("<")[0]
        traffic = pmc_kernels.get(base, {}).get("hbm_bytes_per_launch")
        roofline.update({"traffic": traffic, "traffic_source": pmc_note,
                         "traffic_over_algorithmic": None if traffic is None else traffic / roofline["algorithmic_bytes_per_launch"],
                         "note": "algorithmic = 20 B/row of compulsory column reads (+5 validity bits/row), SURVEY.md 8d"})
        all_ms = sum(v[1] for v in kernels.values())
        chain = {"kernel_ms_per_step": all_ms / prof_steps,
                 "achieved": bytes_per_row * total_rows_rank / (all_ms * 1e-3) / 1e9, "unit": "GB/s",
                 "frac": bytes_per_row * total_rows_rank / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "note": "the same algorithmic bytes over the summed time of EVERY kernel of the step"}

    cpu = ref_check = None
    legs = {}
    if rank == 0 and world == 1 and on_gpu:
        if not args.no_cpu_baseline:
            cpu, ref_fetched, ref_slices = cpu_baseline(batches, c3_plan, args.cpu_budget)
            if cpu is not None:  # the reference's HOST result of the sample, key by key against the same group-by
                ref_check = check.compare_result(ref_fetched, check.exact_groups(batches, slices=ref_slices), hash_identity=False)
                ref_check["rows"] = sum(n for _, _, n in ref_slices)
                ref_check["batches_sampled"] = len({bi for bi, _, _ in ref_slices})
                ok = ok and ref_check["status"] == "ok"
        if not args.no_legs:
            common = ["--rows", str(rows), "--null-fraction", str(args.null_fraction), "--steps", "3", "--warmup", "1"]
            big = common + ["--batch-rows", str(batch_rows)]
            wanted = [w for w in args.legs.split(",") if w]

            legs_t0 = time.perf_counter()

            def want(name, needs_s):
                """selected by --legs, and enough of --leg-budget left for what the leg usually takes"""
                if not (args.legs == "all" or any(w in name for w in wanted)):
                    return False
                left = args.leg_budget - (time.perf_counter() - legs_t0)
                if args.leg_budget > 0 and left < needs_s:
                    legs[name] = {"skipped": f"needs ~{needs_s} s, {max(left, 0):.0f} s of --leg-budget {args.leg_budget:.0f} s left "
                                             "(run with --leg-budget 0 for every leg)"}
                    return False
                return True

            def leg(name, env, argv, needs_s=15):
                if want(name, needs_s):
                    t0 = time.perf_counter()
                    legs[name] = run_leg(env, argv)
                    if isinstance(legs[name], dict):
                        legs[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)

            # BASELINE configs C2 (100 M rows, one predicate + COUNT(*)) and C4 at its stated size (1 B rows, 50 M-key cuckoo
            # join, 50 M groups through Sort + Reduce): tools/bench_configs.py, each checked (count / every key -> sum)
            def tool_leg(name, which, timeout, needs_s):
                if not want(name, needs_s):
                    return
                t0 = time.perf_counter()
                try:
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), which], capture_output=True,
                                       text=True, timeout=timeout, cwd=ROOT)
                    rows_ = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
                    for row in rows_:
                        row["leg_wall_s"] = round(time.perf_counter() - t0, 1)
                    legs[name] = rows_ if r.returncode == 0 and rows_ else {"error": f"rc {r.returncode}", "stderr": r.stderr[-300:]}
                except subprocess.TimeoutExpired:
                    legs[name] = {"error": "timeout"}

            # in the order of what the round's review asks about first; the two long ones last
            # the query shape the Go host really issues: ts >= from, ts < to in front of the query's own filter
            leg("time_filters_ts_ge_lt_then_d1", {}, big + ["--ts-range", "3600,601200"])
            # the reference's DEFAULT aggregation path (enable_hash_reduction: false, and COUNT(*) always): Sort + Reduce over the
            # same four dimensions — hash-keyed inside the ABI (sort_reduce_fused.hip), rows in ascending 64-bit hash order
            leg("c3_sort_path_count", {}, big + ["--sort-path", "count"])
            leg("c3_sort_path_sum_unsigned", {}, big + ["--sort-path", "sum"])
            # ... and with the scan-fed path declining everything: the batch's transforms are launched and Reduce orders the groups
            # over the rows they wrote (the wide layout of sort_reduce_fused.hip — what a join or a generic expression gets)
            leg("c3_sort_path_count_materialised_rows", {"ARES_SR_SCAN_FED": "0"}, big + ["--sort-path", "count"])
            # the ABI's limit of dimensions (MAX_DIMENSIONS = 8) on the fused path: generated kernels only
            leg("c3_eight_dimensions", {}, big + ["--eight-dims"])
            # archive batches (mode-3 run-length sort columns ts and d3): decoded once per batch, then the headline's path
            # (two warm-up passes: sorted rows overflow record streams on whichever batch is the most skewed — the slack grows and
            # the larger workspace comes from the driver once, 60-300 ms of hipMalloc that a timed pass must not contain)
            leg("archive_batches_rle_ts_d3", {}, big + ["--archive", "--ts-range", "3600,601200", "--warmup", "2"])
            leg(f"live_batches_{LIVE_BATCH_ROWS}_rows", {}, common + ["--batch-rows", str(LIVE_BATCH_ROWS)])
            # lower-cardinality variants of the same query (same filter and measure; fewer group-by dimensions)
            # the reference's own example table and queries at 1 B rows (examples/1k_trips: request_at Uint32, city_id Uint16 in
            # a 2-byte dimension slot, status Uint8): SUM(fare) through HashReduce, COUNT(*) through Sort + Reduce, key-level checked
            tool_leg("trips_shaped_1B_rows_u16_dim_u8_filter", "trips", 600, 30)
            # BASELINE config C4 at its stated size ahead of the variants of C3 (a leg that does not fit the budget is skipped)
            tool_leg("c4_spec_1B_rows_50M_keys", "c4spec", 600, 90)
            leg("groups_100_dims_d2_d3", {}, big + ["--dims", "d2,d3"])        # TABLE-mode scan, ~100 groups, 4 columns read
            leg("groups_15k_dims_ts_d1", {}, big + ["--dims", "ts,d1"])        # DIRECT-mode kernels (> 6000 groups)
            tool_leg("c2_100M_rows_filter_count", "c2", 300, 15)
            # first query of a fresh process (kernels compiled in the background: empty on-disk cache), the same process
            # warm, and the same shape with a comparison constant never seen before: measured at the start of this run
            legs.update(early_legs)
            leg("fused_extension", {}, big + ["--fused-extension"])
            leg("groups_4k6_dims_d1_d2", {}, big + ["--dims", "d1,d2"])        # TABLE-mode scan, table well filled
            leg("groups_90_dims_d1", {}, big + ["--dims", "d1"])               # TABLE-mode scan, ~100 groups, 2 columns read
            leg("unfused_abi_ARES_FUSE=0", {"ARES_FUSE": "0"}, big)
            leg("eager_abi_ARES_DEFER=0", {"ARES_DEFER": "0"}, big)
            if want("host_batches", 30):
                t0 = time.perf_counter()
                try:
                    legs["host_batches"] = host_batch_leg(be, plan, batches, device_index, streams)
                    legs["host_batches"]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
                except Exception as e:  # noqa: BLE001
                    legs["host_batches"] = {"error": f"{type(e).__name__}: {e}"}
            legs["legs_wall_s"] = round(time.perf_counter() - legs_t0, 1)

    if rank == 0:
        value = rows * world * args.steps / elapsed
        out = {
            "metric": "rows/sec, 1B-row filter -> group-by-agg (whole job)", "value": value, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "median_ms_per_step": float(sorted(step_ms)[len(step_ms) // 2]), "max_ms_per_step": float(max(step_ms)),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 keys / f64 sums", "data": "synthetic",
            "config": {"workload": "C3: filter d1<90 -> dims [floor(ts,3600), d1, d2, d3] (4 x uint32) -> "
                                   "SUM(m float32 -> float64) via HashReduce, validity bitmaps with "
                                   f"{args.null_fraction:.0%} nulls, shard resident in HBM",
                       "rows_per_gpu": rows, "batch_rows": batch_rows, "batches": len(batches),
                       "streams_per_query": n_streams,
                       "groups_per_shard": groups, "merged_groups": merged_groups, "merge_transport": merge_transport,
                       "host_ms_of_each_step": [round(x, 2) for x in step_ms],
                       "rtc_kernels_after_priming_pass": rtc_state,
                       "profiled_pass_ms_per_step": profiled_ms,
                       "libmem_driver_calls_in_timed_region": None if drv0 is None else
                       {k: drv1[k] - drv0[k] for k in drv0},
                       "parallelism": f"{world} shard(s), one per GPU" + (", all-gather + re-reduce merge in libaresdriver.so" if distributed else "")},
            "rows_per_sec_per_gpu": value / world,
            "algorithmic_GBps_end_to_end": value / world * bytes_per_row / 1e9,
            "check_groups": report, "check_merged_groups": merged_check, "reference_host_check": ref_check,
            "per_rank_ms_per_step": None if rank_times is None else [
                {"step": t[0], "shard": t[1], "merge": t[2]} for t in rank_times],
            "roofline": roofline, "roofline_all_kernels": chain, "cpu_baseline": cpu, "kernels": kern_out, "legs": legs,
            "wall_s_whole_run": round(time.perf_counter() - t_process0, 1),
        }
        # the numbers a reader looks for first, once more at the very END of the line: a log that keeps only the tail of
        # this (long) line still has them
        def _leg(name, key="ms_per_step"):
            v = legs.get(name)
            if isinstance(v, dict):
                return v.get(key, v.get("skipped") or v.get("error"))
            return None
        c2 = legs.get("c2_100M_rows_filter_count")
        c4 = legs.get("c4_spec_1B_rows_50M_keys")
        tr = legs.get("trips_shaped_1B_rows_u16_dim_u8_filter")
        out["summary"] = {
            "value_rows_per_s": value, "ms_per_step": out["ms_per_step"], "check_groups": report["status"],
            "roofline_frac_dominant_kernel": None if roofline is None else round(roofline["frac"], 4),
            "dominant_kernel": None if roofline is None else roofline["kernel"],
            "traffic_over_algorithmic": None if roofline is None else roofline.get("traffic_over_algorithmic"),
            "roofline_frac_all_kernels": None if chain is None else round(chain["frac"], 4),
            "cpu_baseline_rows_per_s": None if cpu is None else cpu["value"], "cpu_baseline_kind": None if cpu is None else cpu["kind"],
            "cpu_baseline_sample": None if cpu is None else cpu["sample"],
            "legs_ms_per_1B_rows": {n: _leg(n) for n in legs if isinstance(legs.get(n), dict) and "ms_per_step" in legs[n]},
            "legs_dominant_kernel_frac": {n: round(legs[n]["roofline"]["frac"], 3) for n in legs
                                          if isinstance(legs.get(n), dict) and isinstance(legs[n].get("roofline"), dict)},
            "cold_first_query_ms": _leg("cold_process", "cold_first_query_ms"),
            "cold_warm_disk_first_query_ms": _leg("cold_process_warm_disk_cache", "cold_first_query_ms"),
            "cold_new_constants_ms": _leg("cold_process_warm_disk_cache", "new_constants_query_ms"),
            "c2_ms_per_query": [round(r["ms"], 3) for r in c2] if isinstance(c2, list) else c2,
            "c4_spec_ms": [round(r["ms"], 1) for r in c4] if isinstance(c4, list) else c4,
            "trips_shaped": [{"query": r["query"], "ms_per_1B_rows": round(r["ms_per_step"], 2), "rows_per_s": round(r["rows_per_s"]),
                              "check": r["key_level_check"]} for r in tr] if isinstance(tr, list) else tr,
            "host_batches_pcie_inclusive_rows_per_s": (legs.get("host_batches") or {}).get("pcie_inclusive", {}).get("rows_per_sec")
            if isinstance(legs.get("host_batches"), dict) else None,
        }
        emit(out, full_line=bool(args.leg))
    if distributed:
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=tdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(int(okt))
        if comm is not None:
            comm.destroy()
        if backend is None or created_group:  # (left to the interpreter's exit, gloo's threads abort the process now and then)
            dist.barrier()
            dist.destroy_process_group()
    for s in streams:
        be.call("DestroyCudaStream", s, device_index)
    if not ok:
        if rank == 0:
            print(f"bench.py: verification failed: {report['status']} / {ref_check}", file=sys.stderr)
        sys.exit(1)
    return 0


if __name__ == "__main__":
    main()
