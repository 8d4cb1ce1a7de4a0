#!/bin/bash
# cold-start legs as bench.py runs them: children of a process that holds a shard on the same GPU
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
python - <<'PY'
import os, subprocess, sys, tempfile, json, torch
x = torch.zeros(5 << 30, dtype=torch.uint8, device="cuda:0")  # the parent's shard
torch.cuda.synchronize()
def leg(tag, env, tmp):
    e = {**os.environ, "ARES_RTC_CACHE_DIR": tmp, "ARES_RTC_TRACE": f"gpurun_out/r4/rtc_trace_{tag}.log", **env}
    try: os.remove(e["ARES_RTC_TRACE"])
    except OSError: pass
    r = subprocess.run([sys.executable, "bench.py", "--leg", "--cold", "--rows", "1e9", "--batch-rows", "67108864", "--steps", "3", "--warmup", "1"],
                       env=e, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
    print(tag, {k: (v if not isinstance(v, list) else [round(x, 1) for x in v[:4]]) for k, v in d.items() if "ms" in k or "error" in k})
    try: print(open(e["ARES_RTC_TRACE"]).read())
    except OSError: pass
for variant, env in (("async", {}), ("sync", {"ARES_RTC_ASYNC": "0"})):
    with tempfile.TemporaryDirectory() as tmp:
        leg(variant + "_empty", env, tmp)
        leg(variant + "_warmdisk", env, tmp)
PY
