"""The bench line the driver parses: the committed result of the last GPU run (profiles/r01_bench_final.json) must carry
every field of the contract, consistent with BASELINE.json, and bench.py must still emit those keys (static check)."""
import json, os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_final.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac",
        "traffic")) <= set(r) and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert abs(d["value"] - d["steps"] * d["config"]["pairs_per_step"] / (d["ms_per_step"] * d["steps"] * 1e-3) / 1e6) < 0.01 * d["value"]   # value = pairs / wall time
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base.get("metric", d["metric"])


def test_bench_source_still_emits_the_contract_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ("\"metric\"", "\"value\"", "\"unit\"", "\"n_gpus\"", "\"steps\"", "\"warmup\"", "\"ms_per_step\"", "\"higher_is_better\"",
        "\"scaling\"", "\"vs_baseline\"", "\"dtype\"", "\"data\"", "\"config\"", "\"roofline\"", "\"cpu_baseline\""):
        assert k in src, k
    assert re.search(r"--gpus", src) and re.search(r"--steps", src) and re.search(r"--warmup", src)
