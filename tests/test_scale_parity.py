"""Configuration-level parity at sizes only the GPU reaches (BASELINE configs C2 / C3 / C4):

* C3 — key-level comparison (group count, every dimension row -> sum, every representative) against
  the independent exact group-by of aresdb_amd/check.py with the reference's 32-bit-hash merges
  predicted, for every HashReduce path: fused (the default: pending transforms consumed by
  HashReduce, previous groups read from the partition-grouped previous result), two alternating
  streams like the Go host, the explicit fused extension, ARES_FUSE=0 (transforms materialised,
  partition + merge), ARES_DEFER=0 (one launch per call), ARES_GROUPED=0 (previous groups
  re-partitioned every batch), the global-table fallback, and many small (2 Mi-row) batches with
  result-buffer reallocation.  Each runs in a child process: the switches are read once per process.
* C2 — 100 M rows, one predicate + COUNT(*), exact.
* C4 — 16.7 M-key cuckoo index (> 256 MB of buckets, beyond the Infinity Cache) joined to a fact
  table and grouped by key through Sort + Reduce: 16.7 M groups, every key / joined attribute / sum
  exact against numpy.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import harness as H
from aresdb_amd import abi, queries
from aresdb_amd.driver import NativeQuery
from aresdb_amd.executor import Col, DimensionSpec, ForeignTable, QueryPlan

pytestmark = pytest.mark.gpu

BIG = ["--rows", str(48 << 20), "--batch-rows", str(32 << 20)]          # 32 Mi + 16 Mi rows
MID = ["--rows", str(20 << 20), "--batch-rows", str(8 << 20)]           # 8 + 8 + 4 Mi rows
LIVE = ["--rows", str(24 << 20), "--batch-rows", str(2 << 20)]          # 12 live-batch-sized batches

C3_VARIANTS = [
    ("fused", {}, BIG),
    ("unfused", {"ARES_FUSE": "0"}, BIG),
    ("global_table", {"ARES_HASH_REDUCE": "global"}, BIG),
    ("two_streams", {}, MID + ["--streams", "2"]),
    ("extension", {}, MID + ["--fused-extension"]),
    ("eager", {"ARES_DEFER": "0"}, MID),
    ("ungrouped", {"ARES_GROUPED": "0"}, MID),
    ("live_batches", {}, LIVE + ["--streams", "2"]),
    # table images (the default for DIRECT-mode batches): the same run with every block that is handed out as cleared checked
    # on the device (an image-mode merge leaves the measure vector unwritten and must not report it written), and with the
    # images switched off (the merge re-inserts the previous groups from their partition-grouped ranges, as round 4 did)
    ("live_batches_verify_clean", {"ARES_MEM_VERIFY_CLEAN": "1"}, LIVE + ["--streams", "2"]),
    ("live_batches_no_image", {"ARES_IMAGE": "0"}, LIVE + ["--streams", "2"]),
    # the query shape the Go host really issues: ts >= from, ts < to in front of the query's own filter — counted in row
    # space, the second one predicted from the stream's previous batch (five batches per stream); and the same with the
    # filters taking the predicate-vector path of rounds 1-3
    ("time_filters", {}, ["--rows", str(40 << 20), "--batch-rows", str(4 << 20), "--streams", "2", "--ts-range", "40000,560000"]),
    ("time_filters_pred_vectors", {"ARES_FILTER_ROWSPACE": "0"},
     ["--rows", str(40 << 20), "--batch-rows", str(8 << 20), "--streams", "2", "--ts-range", "40000,560000"]),
    # the reference's DEFAULT aggregation path, Sort + Reduce (COUNT(*) always goes there; every SUM too unless
    # enable_hash_reduction is set): hash-keyed inside the ABI (sort_reduce_fused.hip) — every key -> count / sum exact AND the
    # rows in ascending order of the 64-bit row hash; the same with the path switched off (rows are sorted for real)
    ("sort_path_count", {}, BIG + ["--streams", "2", "--sort-path", "count"]),
    ("sort_path_sum_time_filters", {}, ["--rows", str(40 << 20), "--batch-rows", str(8 << 20), "--streams", "2", "--ts-range", "40000,560000",
                                        "--sort-path", "sum"]),
    ("sort_path_count_live_batches", {}, LIVE + ["--streams", "2", "--sort-path", "count"]),
    ("sort_path_count_real_sort", {"ARES_SORT_FUSE": "0"}, MID + ["--sort-path", "count"]),
    # ... and with the scan-fed path declining everything: the transforms are launched and the groups ordered over the rows they
    # wrote (the wide layout: two partition levels at this size, ~26 rows per group and batch, the previous result found by
    # the row hashes kept beside it)
    ("sort_path_count_materialised_rows", {"ARES_SR_SCAN_FED": "0"}, BIG + ["--streams", "2", "--sort-path", "count"]),
]


@pytest.mark.parametrize("name,env,args", C3_VARIANTS, ids=[v[0] for v in C3_VARIANTS])
def test_c3_key_level_parity_at_scale(name, env, args):
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tests", "scale_check.py"), *args], cwd=H.ROOT,
                       env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    report = json.loads(lines[-1])
    assert r.returncode == 0 and report["status"] == "ok", report
    assert report["groups"] == report["expected_groups"] > 1_000_000
    assert report["result_sizes"][-1] == report["groups"]
    if name in ("fused", "two_streams", "live_batches", "time_filters"):
        assert report["fused_batches"] == 0  # (the extension counter: these go through the plain ABI)
    if name == "extension":
        assert report["fused_batches"] == len(report["result_sizes"])
    if name.startswith("sort_path"):  # which path ran
        fused = any(k.startswith("sr_merge_kernel") for k in report["kernels"])
        sorted_rows = any(k.startswith("radix_pass_kernel") for k in report["kernels"])
        assert (fused, sorted_rows) == ((False, True) if name.endswith("real_sort") else (True, False)), report["kernels"]
        if name.endswith("materialised_rows"):
            assert any(k.startswith("sr_split_kernel") for k in report["kernels"]) and not any(k.startswith("sr_scan_rtc") for k in report["kernels"])


def test_c2_filter_count_at_spec_size():
    """C2 at its stated size: 100 M rows, single uint32 predicate + COUNT(*)."""
    import torch
    from aresdb_amd import workload
    be = H.hip_backend()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    n, hi = 100_000_000, 86400 * 30
    for selectivity in (0.1, 0.9):
        thr = int(hi * selectivity)
        ts = torch.randint(0, hi, (n,), dtype=torch.int32, device=dev, generator=gen)
        valid = torch.rand((n,), dtype=torch.float32, device=dev, generator=gen) >= 0.01
        ts = torch.where(valid, ts, torch.zeros((), dtype=torch.int32, device=dev))
        col = workload._pack_column(ts, valid, abi.Uint32)
        want = int(((ts < thr) & valid).sum())
        torch.cuda.synchronize()
        q = NativeQuery(be, queries.c2_plan(thr), ["ts"])
        q.run({"ts": col.vp}, n)
        dims, valids, meas = q.fetch()
        assert q.result_size == 1
        assert int(meas.view(np.uint32)[0]) == want
        q.release()
        del col, ts, valid


def _murmur3_32_u32(keys, seed):
    c1, c2 = np.uint32(0xcc9e2d51), np.uint32(0x1b873593)
    with np.errstate(over="ignore"):
        k = (keys.astype(np.uint32) * c1).astype(np.uint32)
        k = ((k << np.uint32(15)) | (k >> np.uint32(17))).astype(np.uint32)
        k = (k * c2).astype(np.uint32)
        h = (np.uint32(seed) ^ k).astype(np.uint32)
        h = ((h << np.uint32(13)) | (h >> np.uint32(19))).astype(np.uint32)
        h = (h * np.uint32(5) + np.uint32(0xe6546b64)).astype(np.uint32)
        h ^= np.uint32(4)
        h ^= h >> np.uint32(16)
        h = (h * np.uint32(0x85ebca6b)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        h = (h * np.uint32(0xc2b2ae35)).astype(np.uint32)
        h ^= h >> np.uint32(16)
    return h


def build_cuckoo_u32(keys, num_buckets, seeds, per_batch, base_batch_id=1):
    """Vectorised builder of the reference's cuckoo layout (memstore/cuckoo_index.go:42-48: bucket =
    [RecordID x 8][signature u8 x 8][key x 8]) for 4-byte keys: every key goes to the first of its 4
    candidate buckets that still has a free slot (the insertion policy does not matter to the probe,
    only the layout does).  Returns (table bytes, placed mask)."""
    bucket_bytes = 8 * (8 + 1 + 4)
    table = np.zeros((num_buckets + 1) * bucket_bytes, np.uint8)
    t32 = table.view(np.uint32)
    fill = np.zeros(num_buckets, np.int64)
    remaining = np.arange(len(keys), dtype=np.int64)
    placed = np.zeros(len(keys), bool)
    for s in seeds:
        if not len(remaining):
            break
        hv = _murmur3_32_u32(keys[remaining], s)
        b = (hv % np.uint32(num_buckets)).astype(np.int64)
        order = np.argsort(b, kind="stable")
        bs, rem = b[order], remaining[order]
        head = np.ones(len(bs), bool)
        head[1:] = bs[1:] != bs[:-1]
        start = np.nonzero(head)[0]
        rank = np.arange(len(bs)) - np.repeat(start, np.diff(np.append(start, len(bs))))
        slot = fill[bs] + rank
        ok = slot < 8
        i, sb, ss, hh = rem[ok], bs[ok], slot[ok], hv[order][ok]
        base = sb * bucket_bytes
        t32[(base + 8 * ss) // 4] = (base_batch_id + i // per_batch).astype(np.uint32)
        t32[(base + 8 * ss) // 4 + 1] = (i % per_batch).astype(np.uint32)
        table[base + 64 + ss] = np.maximum(1, hh >> np.uint32(24)).astype(np.uint8)
        t32[(base + 72 + 4 * ss) // 4] = keys[i]
        np.add.at(fill, sb, 1)
        placed[i] = True
        remaining = rem[~ok]
    return table, placed


def test_c4_join_sort_reduce_at_16m_groups():
    """C4 towards its stated shape: a 16.7 M-key cuckoo index (348 MB of buckets: every probe is an
    HBM access) joined to the fact table, group by (fk, joined attribute) through Sort + Reduce —
    one group per key.  Batch 1 holds every key once, batches 2 and 3 draw keys at random (the result buffers grow before
    batch 2, so its previous result is hashed again; batch 3 finds the row hashes batch 2 left beside its result).  On the
    HIP side the reduction orders GROUPS, not rows (fused_sort_reduce_vectors, the wide layout: 2^15 partitions here) — the
    16.7 M output rows must be in ascending order of their 64-bit row hash, like a sort's."""
    import torch
    be = H.hip_backend()
    rng = np.random.default_rng(6)
    nkeys, per_batch = 1 << 24, 1 << 20
    keys = rng.permutation(np.arange(1, (1 << 26) + 1, 4, dtype=np.uint32))[:nkeys]
    assert len(keys) == nkeys
    attr = (keys * np.uint32(2654435761) >> np.uint32(20)).astype(np.uint32)
    seeds = [int(x) for x in rng.integers(0, 1 << 32, 4)]
    num_buckets = nkeys // 5
    table, placed = build_cuckoo_u32(keys, num_buckets, seeds, per_batch)
    assert placed.mean() > 0.999
    usable = np.nonzero(placed)[0]
    nb = nkeys // per_batch
    tb = H.Buf(be, table)
    idx = abi.CuckooHashIndex()
    idx.buckets = tb.ptr
    for i, s in enumerate(seeds):
        idx.seeds[i] = s
    idx.keyBytes, idx.numHashes, idx.numBuckets = 4, 4, num_buckets
    dcols = [H.Column(be, abi.Uint32, attr[b * per_batch:(b + 1) * per_batch]) for b in range(nb)]
    ft = ForeignTable(join_column="fk", index=idx, batches={"attr": [c.vp for c in dcols]},
                      data_types={"attr": abi.Uint32}, base_batch_id=1, num_records_in_last_batch=per_batch)
    plan = QueryPlan(filters=[], foreign_tables=[ft], foreign_filters=[],
                     dimensions=[DimensionSpec(Col("fk"), abi.Uint32), DimensionSpec(Col("attr", table=1), abi.Uint32)],
                     measure=Col("amount"), agg=abi.AGGR_SUM_UNSIGNED, measure_type=abi.Uint32, use_hash_reduction=False)
    q = NativeQuery(be, plan, ["fk", "amount"])
    sums = np.zeros(nkeys, np.int64)
    from aresdb_amd.columns import DeviceColumn
    be.profiler_enable(True)
    for batch in range(3):
        pick = rng.permutation(usable) if batch == 0 else usable[rng.integers(0, len(usable), 1 << 24)]
        amount = rng.integers(0, 100, len(pick)).astype(np.uint32)
        cf, ca = DeviceColumn(be, abi.Uint32, keys[pick]), DeviceColumn(be, abi.Uint32, amount)
        q.run({"fk": cf.vp, "amount": ca.vp}, len(pick))
        cf.free(); ca.free()
        sums += np.bincount(pick, weights=amount, minlength=nkeys).astype(np.int64)
    be.wait()
    kernels = be.profiler_report()
    be.profiler_enable(False)
    dims, valids, meas = q.fetch()
    assert q.result_size == len(usable)
    from aresdb_amd import check
    hashes = check.row_hashes_of_fetched(dims, valids)
    assert (hashes[1:] > hashes[:-1]).all(), "output rows not in ascending order of the row hash"
    if all(os.environ.get(k, "1") != "0" for k in ("ARES_FUSE", "ARES_DEFER", "ARES_SORT_FUSE", "ARES_RTC", "ARES_SORT_VECTORS")):
        assert "sr_split_kernel" in kernels and "sr_bounds_kernel" in kernels and "radix_pass_kernel" not in kernels, sorted(kernels)
    got_fk, got_attr, got_sum = dims[0].view(np.uint32), dims[1].view(np.uint32), meas.view(np.uint32)
    assert valids[0].all() and valids[1].all()
    order = np.argsort(got_fk)
    korder = usable[np.argsort(keys[usable])]
    assert np.array_equal(got_fk[order], keys[korder])
    assert np.array_equal(got_attr[order], attr[korder])
    assert np.array_equal(got_sum[order].astype(np.int64), sums[korder])
    q.release()
    for b in [tb] + dcols:
        b.free()
