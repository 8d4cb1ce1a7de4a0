"""bench.py's multi-GPU contract, as far as a box without GPUs can prove it: `--gpus N` is honoured
(spawned ranks or a loud failure, never a silent N = 1), and the N-rank code path of the file —
per-rank shards, barrier-bracketed timing, max over ranks, all_gather + re-reduce merge, key-level
check of the merged table — runs under torch.distributed with world_size 2 (gloo, oracle backend
injected by tests/bench_dist_harness.py)."""
import json
import os
import subprocess
import sys

import harness as H


def test_gpus_flag_is_never_silently_ignored():
    """No GPU here: `bench.py --gpus 2` must refuse, not run one rank and print n_gpus 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--rows", "1000"],
                       cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_launcher_world_size_must_match_gpus_flag():
    env = {**os.environ, "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "4", "--rows", "1000"],
                       cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "rank" in (r.stderr + r.stdout)


def test_two_rank_path_merges_and_verifies(tmp_path):
    H.oracle_backend()  # builds liboracle.so once, before two ranks race for it
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(H.ROOT, "tests", "bench_dist_harness.py"), "--gpus", "2", "--rows", "6000",
           "--batch-rows", "2500", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-legs", "--verify-merged"]
    r = subprocess.run(cmd, cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["check_groups"]["status"] == "ok"
    assert out["check_merged_groups"]["status"] == "ok"
    assert out["config"]["merged_groups"] == out["check_merged_groups"]["groups"] >= out["config"]["groups_per_shard"]
    assert len(out["per_rank_ms_per_step"]) == 2


def test_eight_rank_path_merges_and_verifies():
    """The launch the driver uses on an 8-GPU node (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`), with
    gloo and the oracle backend: eight shards, one merge, the merged table checked key by key."""
    H.oracle_backend()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(H.ROOT, "tests", "bench_dist_harness.py"), "--gpus", "8", "--rows", "5000",
           "--batch-rows", "2500", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-legs", "--verify-merged"]
    r = subprocess.run(cmd, cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["steps"] == 2
    assert out["check_groups"]["status"] == "ok" and out["check_merged_groups"]["status"] == "ok"
    assert len(out["per_rank_ms_per_step"]) == 8
    assert out["config"]["merged_groups"] == out["check_merged_groups"]["groups"] >= out["config"]["groups_per_shard"]


def test_single_process_threads_path_merges_and_verifies(capsys):
    """bench.py --single-process: shards as threads of one process (the reference's process model), here two threads
    on the oracle backend; the merged table is checked key by key."""
    import sys as _sys
    _sys.path.insert(0, H.ROOT)
    import bench
    rc = bench.main(["--gpus", "2", "--single-process", "--rows", "6000", "--batch-rows", "2500", "--steps", "1", "--warmup", "0",
                     "--verify-merged"], backend=H.oracle_backend(), tensor_device="cpu")
    out = json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1])
    assert rc == 0 and out["n_gpus"] == 2 and out["check_merged_groups"]["status"] == "ok"
    assert out["config"]["merged_groups"] == out["check_merged_groups"]["groups"] > 1000
    assert "THREAD" in out["config"]["parallelism"]
