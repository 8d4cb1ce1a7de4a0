"""N>1 path on CPU: two processes (gloo), one shard each, partial group tables merged with
aresdb_amd.shard_merge — the same code bench.py runs over RCCL.  The library behind the ABI is the
oracle here (host memory); the merged table must equal one process reducing both shards
(the reference's "append previous results + re-reduce" contract,
query/aql_batchexecutor.go:236-251; combine rules broker/result_merge.go:77-94)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_shard(be, plan, batches):
    from aresdb_amd.executor import BatchContext, BatchExecutor
    ctx = BatchContext(be, plan)
    ex = BatchExecutor(ctx)
    for b in batches:
        ex.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
    return ctx


def _single_process_result(be, plan, all_batches):
    from aresdb_amd.executor import fetch_results
    ctx = _run_shard(be, plan, all_batches)
    dims, valids, meas = fetch_results(ctx)
    n = ctx.result_size
    m = meas.view(np.float64) if plan.measure_bytes == 8 else meas.view(np.uint32)
    out = {}
    for r in range(n):
        key = tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids))
        out[key] = m[r]
    ctx.release()
    return out


def _native_worker(rank, world, port, use_hash, q):
    """The merge inside libaresdriver.so (C++), the collective supplied by torch.distributed / gloo."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    from aresdb_amd import workload
    from aresdb_amd.driver import NativeComm, NativeQuery
    from aresdb_amd.queries import c3_plan
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        be = H.oracle_backend()
        plan = c3_plan(use_hash_reduction=use_hash)
        names = [n for n, _ in workload.C3_COLUMNS]
        shards = [workload.c3_shard(5000 + 700 * r, 2048, seed=11 + r, device="cpu") for r in range(world)]
        ctx = NativeQuery(be, plan, names)
        for b in shards[rank]:
            ctx.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
        comm = NativeComm.torch_group()
        ctx.merge_shards(comm)
        dims, valids, meas = ctx.fetch()
        n = ctx.result_size
        m = meas.view(np.float64)
        got = {tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids)): m[r]
               for r in range(n)}
        want = _single_process_result(be, plan, [b for s in shards for b in s])
        assert got.keys() == want.keys(), (len(got), len(want))
        for k, v in want.items():
            assert abs(got[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, got[k], v)
        comm.destroy()
        ctx.release()
        q.put((rank, "ok", len(got)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}", 0))
    finally:
        dist.destroy_process_group()


def _partitioned_worker(rank, world, port, use_hash, all_to_all, q):
    """Hash-partitioned merge inside libaresdriver.so: the ranks' shares are disjoint, their union equals
    one process reducing every shard."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    from aresdb_amd import workload
    from aresdb_amd.driver import NativeComm, NativeQuery
    from aresdb_amd.queries import c3_plan
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        be = H.oracle_backend()
        plan = c3_plan(use_hash_reduction=use_hash)
        names = [n for n, _ in workload.C3_COLUMNS]
        # the last rank's shard is empty when there are three: a rank with nothing to send
        rows = [5000 + 700 * r if (world < 3 or r < world - 1) else 0 for r in range(world)]
        shards = [workload.c3_shard(rows[r], 2048, seed=11 + r, device="cpu") if rows[r] else [] for r in range(world)]
        ctx = NativeQuery(be, plan, names)
        for b in shards[rank]:
            ctx.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
        comm = NativeComm.torch_group(all_to_all=all_to_all)
        total = ctx.merge_shards_partitioned(comm)
        n = ctx.result_size
        got = {}
        if n:
            dims, valids, meas = ctx.fetch()
            m = meas.view(np.float64)
            got = {tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids)): m[r]
                   for r in range(n)}
        everyone = [None] * world
        dist.all_gather_object(everyone, got)
        union = {}
        for part in everyone:
            assert not (union.keys() & part.keys()), "two ranks hold the same group"
            union.update(part)
        want = _single_process_result(be, plan, [b for s in shards for b in s])
        assert total == len(want) == len(union), (total, len(want), len(union))
        assert union.keys() == want.keys()
        for k, v in want.items():
            assert abs(union[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, union[k], v)
        comm.destroy()
        ctx.release()
        q.put((rank, "ok", n))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,all_to_all", [(2, True), (3, True), (3, False), (8, True)],
                         ids=["2_all_to_all", "3_all_to_all", "3_all_gathers", "8_all_to_all"])
@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_hash_partitioned_shard_merge_gloo(use_hash, world, all_to_all):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_partitioned_worker, args=(r, world, port, use_hash, all_to_all, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
    assert sum(r[2] for r in results) > 0 and all(r[2] > 0 for r in results)  # every rank owns a share


@pytest.mark.parametrize("world", [2, 3, 8])  # 8: one node's worth of shards (BASELINE config C5)
@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_native_shard_merge_gloo(use_hash, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, use_hash, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
    assert len({r[2] for r in results}) == 1 and results[0][2] > 0


def _hll_worker(rank, world, port, q):
    """HyperLogLog shards: every rank runs its batches un-finalised, the merge exchanges the (group, register, rho)
    entries and finalises; every rank's encoded result must equal one process over all the batches — register by
    register (the broker's register-max, broker/result_merge.go:95-104)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    from aresdb_amd import abi, smoke
    from aresdb_amd.columns import DeviceColumn
    from aresdb_amd.driver import NativeComm, NativeQuery
    from test_executor import _hll_batches, _hll_plan
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        be = H.oracle_backend()
        plan = _hll_plan()
        rng = np.random.default_rng(77)
        # rank 2 of three has an empty shard; a dense group (> 4096 registers) needs many users of one (day, d3) pair
        sizes = [[4000, 2500], [3000, 60000], []] if world == 3 else [[4000, 2500, 30000], [3000, 45000]]
        if world > 3:  # eight shards: one dense, one empty, the rest small
            sizes = [[3000, 45000]] + [[1500 + 300 * r] for r in range(1, world - 1)] + [[]]
        shards = [_hll_batches(rng, sz) for sz in sizes]
        for sh in shards:  # dense groups: few distinct group keys
            for cols, valid in sh:
                if len(cols["ts"][1]) > 20000:
                    cols["ts"] = (cols["ts"][0], (cols["ts"][1] % 86400).astype(np.uint32))
        q_ = NativeQuery(be, plan, list(shards[0][0][0].keys()) if shards[0] else ["ts", "d1", "d2", "d3", "m", "user"])
        for cols, valid in shards[rank]:
            dev = {k: DeviceColumn(be, t, v, valid=valid[k]) for k, (t, v) in cols.items()}
            q_.run({k: d.vp for k, d in dev.items()}, len(next(iter(cols.values()))[1]), is_last_batch=False)
            for d in dev.values():
                d.free()
        comm = NativeComm.torch_group()
        q_.merge_shards(comm)
        got = smoke._hll_groups(*q_.fetch_hll())
        want, _ = smoke.run_hll_query(be, plan, [b for sh in shards for b in sh], native=True)
        assert got == want, (len(got), len(want))
        assert any(len(v) >= 4096 for v in got.values()), "no dense group in the test data"
        comm.destroy()
        q_.release()
        q.put((rank, "ok", len(got)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e} {traceback.format_exc()[-600:]}", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_hll_shard_merge_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hll_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
    assert len({r[2] for r in results}) == 1 and results[0][2] > 0


def _threads_merge(be, device_memory, shards_of, plan, names, nranks, make_stream=None, device_of=lambda r: 0):
    """nranks shards as nranks threads of this process (the reference's process model), merged inside
    libaresdriver.so through the in-process communicator; returns every rank's merged table."""
    import threading
    from aresdb_amd.driver import NativeComm, NativeQuery
    comms = NativeComm.local(nranks, device_memory)
    got, errs = [None] * nranks, []

    def work(r):
        try:
            stream = make_stream() if make_stream else None
            ctx = NativeQuery(be, plan, names, device=device_of(r), stream=stream)
            for b in shards_of(r):
                ctx.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
            ctx.merge_shards(comms[r])
            dims, valids, meas = ctx.fetch()
            n = ctx.result_size
            m = meas.view(np.float64)
            got[r] = {tuple((bytes(d[i * len(d) // n:(i + 1) * len(d) // n]), int(v[i])) for d, v in zip(dims, valids)): m[i]
                      for i in range(n)}
            ctx.release()
            if stream is not None:
                be.call("DestroyCudaStream", stream, 0)
        except Exception as e:  # noqa: BLE001
            import traceback
            errs.append((r, f"{type(e).__name__}: {e}", traceback.format_exc()[-500:]))
    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for c in comms:
        c.destroy()
    assert not errs, errs
    return got


@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_shards_as_threads_of_one_process(use_hash):
    """Three shards, three threads, one process (query/device_manager.go:185-218), the oracle behind the ABI."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    from aresdb_amd import workload
    from aresdb_amd.queries import c3_plan
    be = H.oracle_backend()
    plan = c3_plan(use_hash_reduction=use_hash)
    names = [n for n, _ in workload.C3_COLUMNS]
    shards = [workload.c3_shard(4000 + 900 * r, 2048, seed=21 + r, device="cpu") for r in range(3)]
    got = _threads_merge(be, False, lambda r: shards[r], plan, names, 3)
    want = _single_process_result(be, plan, [b for s in shards for b in s])
    for r in range(3):
        assert got[r].keys() == want.keys(), (r, len(got[r]), len(want))
        for k, v in want.items():
            assert abs(got[r][k] - v) <= 1e-9 * max(1.0, abs(v)), (r, k)


@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_eight_shard_threads_on_eight_device_indices(use_hash):
    """BASELINE config C5's shape in the reference's process model: eight shards, eight threads, device index r for
    shard r (query/device_manager.go:185-218) — on the oracle, which has one simulated device, so what this exercises is
    the host side: every device-keyed structure of libaresdriver.so and of the Python driver sees indices 0 .. 7, and
    the in-process communicator runs with eight ranks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    from aresdb_amd import workload
    from aresdb_amd.queries import c3_plan
    be = H.oracle_backend()
    plan = c3_plan(use_hash_reduction=use_hash)
    names = [n for n, _ in workload.C3_COLUMNS]
    shards = [workload.c3_shard(2500 + 300 * r, 1024, seed=41 + r, device="cpu") for r in range(8)]
    got = _threads_merge(be, False, lambda r: shards[r], plan, names, 8, device_of=lambda r: r)
    want = _single_process_result(be, plan, [b for s in shards for b in s])
    for r in range(8):
        assert got[r].keys() == want.keys(), (r, len(got[r]), len(want))
        for k, v in want.items():
            assert abs(got[r][k] - v) <= 1e-9 * max(1.0, abs(v)), (r, k)


@pytest.mark.gpu
def test_shards_as_threads_of_one_process_on_the_gpu():
    """The same with the HIP libraries: two shard threads (both on device 0 — this box has one GPU), each with its own
    stream, blocks copied with hipMemcpyAsync on the query streams."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    from aresdb_amd import workload
    from aresdb_amd.queries import c3_plan
    be = H.hip_backend()
    dev = torch.device("cuda:0")
    plan = c3_plan(use_hash_reduction=True)
    names = [n for n, _ in workload.C3_COLUMNS]
    shards = [workload.c3_shard(300000 + 50000 * r, 1 << 17, seed=31 + r, device=dev) for r in range(2)]
    torch.cuda.synchronize()
    got = _threads_merge(be, True, lambda r: shards[r], plan, names, 2, make_stream=lambda: be.call("CreateCudaStream", 0))
    from aresdb_amd.driver import NativeQuery
    ctx = NativeQuery(be, plan, names)
    for b in [b for s in shards for b in s]:
        ctx.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
    dims, valids, meas = ctx.fetch()
    n = ctx.result_size
    m = meas.view(np.float64)
    want = {tuple((bytes(d[i * len(d) // n:(i + 1) * len(d) // n]), int(v[i])) for d, v in zip(dims, valids)): m[i] for i in range(n)}
    ctx.release()
    for r in range(2):
        assert got[r].keys() == want.keys()
        assert all(abs(got[r][k] - v) <= 1e-9 * max(1.0, abs(v)) for k, v in want.items())


def _worker(rank, world, port, use_hash, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    from aresdb_amd import workload
    from aresdb_amd.queries import c3_plan
    from aresdb_amd.shard_merge import merge_shard_results, merged_to_dict
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        be = H.oracle_backend()
        plan = c3_plan(use_hash_reduction=use_hash)
        shards = [workload.c3_shard(6000 + 500 * r, 2048, seed=1 + r, device="cpu") for r in range(world)]
        ctx = _run_shard(be, plan, shards[rank])
        merged = merge_shard_results(ctx, "cpu")
        got = merged_to_dict(ctx, merged)
        want = _single_process_result(be, plan, [b for s in shards for b in s])
        assert got.keys() == want.keys(), (len(got), len(want))
        for k, v in want.items():
            assert abs(got[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, got[k], v)
        ctx.release()
        q.put((rank, "ok", len(got)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_two_shard_merge_gloo(use_hash):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, use_hash, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
    assert results[0][2] == results[1][2] > 0


def _gpu_worker(port, use_hash, q):
    """One rank over RCCL on the GPU box: all_gather + re-reduce through the HIP library on the
    query's own (non-blocking) stream; the merged table must equal the shard's own result."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import harness as H
        from aresdb_amd import workload
        from aresdb_amd.driver import NativeQuery
        from aresdb_amd.queries import c3_plan
        from aresdb_amd.shard_merge import merge_shard_results, merged_to_dict
        os.environ["NCCL_DEBUG"] = "WARN"
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        be = H.hip_backend()
        stream = be.call("CreateCudaStream", 0)
        plan = c3_plan(use_hash_reduction=use_hash)
        batches = workload.c3_shard(300000, 1 << 17, seed=3, device=dev)
        names = [n for n, _ in workload.C3_COLUMNS]
        ctx = NativeQuery(be, plan, names, device=0, stream=stream)
        for b in batches:
            ctx.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
        dims, valids, meas = ctx.fetch()
        n = ctx.result_size
        m = meas.view(np.float64)
        want = {tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids)): m[r]
                for r in range(n)}
        merged = merge_shard_results(ctx, dev)
        got = merged_to_dict(ctx, merged)
        assert got.keys() == want.keys(), (len(got), len(want))
        for k, v in want.items():
            assert abs(got[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, got[k], v)
        # the same merge inside libaresdriver.so: ncclAllGather on the query's stream, re-reduce, fetch
        from aresdb_amd.driver import NativeComm

        def bcast(raw):
            t = torch.tensor(list(raw), dtype=torch.uint8, device=dev)
            dist.broadcast(t, 0)
            return bytes(t.cpu().tolist())
        comm = NativeComm.rccl(0, 1, 0, bcast)
        ctx.merge_shards(comm)
        dims, valids, meas = ctx.fetch()
        n2 = ctx.result_size
        m2 = meas.view(np.float64)
        got2 = {tuple((bytes(d[r * len(d) // n2:(r + 1) * len(d) // n2]), int(v[r])) for d, v in zip(dims, valids)): m2[r]
                for r in range(n2)}
        assert got2.keys() == want.keys() and all(abs(got2[k] - v) <= 1e-9 * max(1.0, abs(v)) for k, v in want.items())
        # ... and the hash-partitioned merge: Sort + Reduce of the table, one ncclSend / ncclRecv pair (to itself)
        total = ctx.merge_shards_partitioned(comm)
        dims, valids, meas = ctx.fetch()
        n3 = ctx.result_size
        m3 = meas.view(np.float64)
        got3 = {tuple((bytes(d[r * len(d) // n3:(r + 1) * len(d) // n3]), int(v[r])) for d, v in zip(dims, valids)): m3[r]
                for r in range(n3)}
        assert total == n3 == len(want)
        assert got3.keys() == want.keys() and all(abs(got3[k] - v) <= 1e-9 * max(1.0, abs(v)) for k, v in want.items())
        comm.destroy()
        ctx.release()
        dist.destroy_process_group()
        q.put(("ok", len(got)))
    except Exception as e:  # noqa: BLE001
        q.put((f"{type(e).__name__}: {e}", 0))


@pytest.mark.gpu
@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_single_rank_merge_rccl(use_hash):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_gpu_worker, args=(_free_port(), use_hash, q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert res[0] == "ok" and res[1] > 0, res


# ---- RCCL with more than one rank: one process per GPU, the contract's launch shape --------------------------------------
def _rccl_worker(rank, world, port, use_hash, q):
    """Rank `rank` on GPU `rank`: its shard through the HIP library, then the three merges of libaresdriver.so over RCCL —
    all-gather + re-reduce, hash-partitioned all-to-all (ncclSend / ncclRecv), and the union check across ranks — against
    one oracle process reducing every shard (shards are generated on the CPU so that every rank and the oracle see the
    same rows)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import harness as H
        from aresdb_amd import workload
        from aresdb_amd.driver import NativeComm, NativeQuery
        from aresdb_amd.queries import c3_plan
        os.environ["NCCL_DEBUG"] = "WARN"
        torch.cuda.set_device(rank)
        dev = torch.device(f"cuda:{rank}")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
        be = H.hip_backend()
        stream = be.call("CreateCudaStream", rank)
        plan = c3_plan(use_hash_reduction=use_hash)
        names = [n for n, _ in workload.C3_COLUMNS]
        rows = [60000 + 7000 * r for r in range(world)]
        shards = [workload.c3_shard(rows[r], 1 << 15, seed=21 + r, device="cpu") for r in range(world)]
        mine = [{k: workload.ResidentColumn(rc.blob.to(dev), rc.values_off, rc.data_type, rc.length, rc.has_nulls) for k, rc in b.items()}
                for b in shards[rank]]

        def bcast(raw):
            t = torch.tensor(list(raw), dtype=torch.uint8, device=dev)
            dist.broadcast(t, 0)
            return bytes(t.cpu().tolist())

        def run_shard():
            ctx = NativeQuery(be, plan, names, device=rank, stream=stream)
            for b in mine:
                ctx.run({k: rc.vp for k, rc in b.items()}, next(iter(b.values())).length)
            return ctx

        def table(ctx):
            n = ctx.result_size
            if not n:
                return {}
            dims, valids, meas = ctx.fetch()
            m = meas.view(np.float64)
            return {tuple((bytes(d[r * len(d) // n:(r + 1) * len(d) // n]), int(v[r])) for d, v in zip(dims, valids)): m[r] for r in range(n)}

        want = _single_process_result(H.oracle_backend(), plan, [b for s in shards for b in s])
        comm = NativeComm.rccl(rank, world, rank, bcast)
        ctx = run_shard()
        ctx.merge_shards(comm)  # ncclAllGather on the query's stream + re-reduce: every rank ends with the whole table
        got = table(ctx)
        assert got.keys() == want.keys(), (len(got), len(want))
        assert all(abs(got[k] - v) <= 1e-9 * max(1.0, abs(v)) for k, v in want.items())
        ctx.release()
        ctx = run_shard()
        total = ctx.merge_shards_partitioned(comm)  # ncclSend / ncclRecv pairs: disjoint shares whose union is the table
        part = table(ctx)
        everyone = [None] * world
        dist.all_gather_object(everyone, part)
        union = {}
        for p in everyone:
            assert not (union.keys() & p.keys()), "two ranks hold the same group"
            union.update(p)
        assert total == len(want) == len(union), (total, len(want), len(union))
        assert all(abs(union[k] - v) <= 1e-9 * max(1.0, abs(v)) for k, v in want.items())
        ctx.release()
        comm.destroy()
        dist.destroy_process_group()
        q.put((rank, "ok", len(got)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}", 0))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4], ids=["1_rank", "2_ranks", "4_ranks"])
@pytest.mark.parametrize("use_hash", [True, False], ids=["hash_reduce", "sort_reduce"])
def test_shard_merges_over_rccl(use_hash, world):
    """Skipped where the box has fewer GPUs than ranks (the 1-rank case always runs: same code path, RCCL talking to itself)."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (RCCL over xGMI), this box has {torch.cuda.device_count()}")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, use_hash, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
    assert len({r[2] for r in results}) == 1 and results[0][2] > 0
