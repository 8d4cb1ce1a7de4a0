"""bench.py's stdout contract: the LAST line is one compact JSON object (< 4 KB) carrying the contract keys, `roofline`
and `cpu_baseline`; the long record (legs, kernel table) goes to bench_full.json.  Round 5's single 22 KB line came back
from the driver unparsed (BENCH_r05.json: parsed null)."""
import json
import os
import sys

import harness as H

sys.path.insert(0, H.ROOT)
import bench  # noqa: E402


def _recorded():
    """A full record as round 5's bench produced it (profiles/r5_final_bench_n1.json), the largest one on file."""
    path = os.path.join(H.ROOT, "profiles", "r5_final_bench_n1.json")
    text = [ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1]
    return json.loads(text)


def test_headline_is_compact_and_complete(tmp_path, capsys, monkeypatch):
    out = _recorded()
    assert len(json.dumps(out)) > 15000  # the input really is the long record
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(out)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(lines) == 1
    assert len(lines[0]) < 4096
    line = json.loads(lines[0])  # strict JSON
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "check_groups"):
        assert k in line, k
    assert line["config"]["workload"].startswith("C3")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["check_groups"]["status"] == "ok"
    assert len(json.dumps(line["summary"])) < 2048
    assert "legs_ms_per_1B_rows" in line["summary"]
    full = json.load(open(os.path.join(str(tmp_path), "bench_full.json")))
    assert "legs" in full and "kernels" in full


def test_headline_sheds_optional_parts_before_it_outgrows_the_bound():
    out = _recorded()
    out["summary"]["legs_ms_per_1B_rows"] = {f"leg_{i}_with_a_long_descriptive_name": float(i) for i in range(200)}
    line = bench.headline(out)
    assert len(json.dumps(line)) < bench.HEADLINE_MAX_BYTES
    assert line["roofline"] and line["cpu_baseline"]
