"""The contract line of bench.py: the driver keeps only the tail of stdout, so the LAST line must be small and carry the keys it parses
(a 23 KB line gave `parsed: null`).  Built here from a recorded full result (profiles/r05_bench_default_f16x3.json) -- no GPU."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RECORDED = os.path.join(ROOT, "profiles", "r05_bench_default_f16x3.json")


def _recorded():
    d = json.load(open(RECORDED))
    ref = dict(d["cpu_baseline_reference"])
    d["cpu_baseline"] = bench.merge_cpu_baseline(d["cpu_baseline"], ref)
    return d, ref


def test_contract_line_is_small_and_complete():
    d, _ = _recorded()
    line = bench.contract_line(d)
    assert len(line) < 6000 and len(line) <= bench.LINE_BUDGET and "\n" not in line
    p = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in p, k
    assert p["value"] == d["value"] and p["ms_per_step"] == d["ms_per_step"]
    assert "workload" in p["config"] and "model" not in p["config"]
    r = p["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert 0 < r["frac_on_evaluated_pairs"] < r["frac"]
    c = p["cpu_baseline"]
    assert c["value"] > 0 and c["unit"] == "rays/s" and c["cores"] >= 1 and c["kind"] in ("reference", "port") and c["sample"]
    assert set(p["parity"]) >= {"c1", "c2"} and p["parity"]["c1"]["sign_flips"] == 0


def test_last_stdout_line_is_the_contract_line(tmp_path, monkeypatch):
    d, _ = _recorded()
    monkeypatch.setenv("O2345_BENCH_EXTRA_FILE", str(tmp_path / "bench_extra.json"))
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(d)
    lines = buf.getvalue().strip().split("\n")
    assert len(lines) == 2 and len(lines[-1]) <= bench.LINE_BUDGET
    last = json.loads(lines[-1])
    assert last["roofline"]["frac"] > 0 and last["cpu_baseline"]["value"] > 0
    assert json.loads(lines[0])["dropin"] and json.load(open(tmp_path / "bench_extra.json"))["trained_regime"]
    # the driver's tail window (BENCH_r05.json kept 8.3 KB of stdout): the contract line fits whole
    assert len(lines[-1]) < 8000


def test_cpu_baseline_prefers_the_reference_on_the_same_host_type():
    d, ref = _recorded()
    port = {"value": 390.0, "unit": "rays/s", "cores": 32, "kind": "port", "sample": "x"}
    m = bench.merge_cpu_baseline(port, ref)
    assert m["kind"] == "reference" and m["value"] == ref["value"] and m["port_rays_per_s"] == 390.0 and m["cores"] == ref["torch_threads"] and m["same_host_type"]
    other = dict(ref, same_host_type=False)
    m = bench.merge_cpu_baseline(port, other)
    assert m["kind"] == "port" and m["value"] == 390.0 and m["reference"]["value"] is None and "measured on" in m["reference"]["refused"]
    m = bench.merge_cpu_baseline(port, {"refused": "no file"})
    assert m["kind"] == "port" and m["reference"]["refused"] == "no file"


def test_line_sheds_detail_before_it_exceeds_the_budget():
    d, _ = _recorded()
    d["config"]["workload"] = "x" * 3000                      # truncated to 400
    d["cpu_baseline"]["sample"] = "y" * 1800
    line = bench.contract_line(d)
    assert len(line) <= bench.LINE_BUDGET
    p = json.loads(line)
    assert "dropped_for_size" in p and "roofline" in p and "cpu_baseline" in p
