"""bench.py's stdout line must stay parseable by the driver (10 KB stdout tail): the line builder is
run on a canned full record (round 3's 23 KB record, which the driver could NOT parse) and on a
worst case with every optional block at its largest."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
            "cpu_baseline")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r03_c_bench.json")) as fh:
        return json.load(fh)


def test_compact_line_fits_and_carries_the_contract():
    import bench
    full = _canned()
    assert len(json.dumps(full)) > 10000          # the record that broke the driver's parser
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert d["learner_steps_per_sec"] == full["learner_steps_per_sec"]
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    for name in ("ppo", "sac"):
        oc = d["other_configs"][name]
        assert {"value", "ms_per_step", "roofline", "cpu_baseline"} <= set(oc)
        assert "frac" in oc["roofline"]


def test_compact_line_drops_optional_blocks_before_it_overflows():
    import bench
    full = _canned()
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["other_configs"]["ppo"] = {"error": "e" * 5000}
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k


def test_error_records_of_other_configs_stay_short():
    import bench
    c = bench._compact_other({"error": "z" * 1000})
    assert len(c["error"]) <= 160
    assert bench._compact_other(None)["error"] == "missing"
