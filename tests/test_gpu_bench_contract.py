"""bench.py prints ONE JSON line with the fields the driver reads (a small workload here; the headline run uses the defaults)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--D", "4", "--T", "2", "--H", "96",
                          "--W", "128", "--cpu-frames", "1", "--loss-steps", "1", "--no-stage2"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Mpix/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 96 * 128 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the line stays under 6 KB, its LAST key is the flat summary (what survives in a record that keeps the tail), the legs in full are in
    # the side file the line names, and the build is stated
    assert len(lines[0]) < 6000
    keys = list(d)
    assert keys[-1] == "summary" and keys.index("roofline") < keys.index("cpu_baseline") < keys.index("summary")
    assert d["build"]["translation_units"] >= 10 and "hipcc" in d["build"]["mode"]
    detail = json.load(open(os.path.join(ROOT, d["detail_file"])))
    assert abs(d["summary"]["loss720_ref_it_s"] - detail["loss"]["ref"]["iters_per_s"]) <= 1e-3 * detail["loss"]["ref"]["iters_per_s"]
    assert all(not isinstance(v, (dict, list)) for k, v in d["summary"].items() if k != "errors")
    assert "errors" not in d["summary"], d["summary"].get("errors")
    assert d["grad_sums"][1] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mpix/s" and "sample" in c
