"""The committed bench line (newest profiles/r*_bench.json, written by `python bench.py` on an MI355X) carries every
key the driver's contract names, with the types and relations it states."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line"
    lines = [l for l in open(files[-1]).read().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def test_top_level_keys():
    b = newest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in b, k
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert b["unit"] == "Mrays/s" and b["metric"].startswith("Mrays/s")
    assert str(baseline.get("metric", "Mrays/s")).lower().startswith("mrays")
    assert b["n_gpus"] == 1 and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert b["data"] == "synthetic" and b["dtype"] == "f32"
    assert "workload" in b["config"] and "model" not in b["config"]
    # value = incoherent rays (depth >= 1) per second; the coherent camera rays are reported beside it
    assert b["value"] > 0 and abs(b["value"] - (b["mrays_secondary_closest"] + b["mrays_secondary_any"])) < 1e-6 * b["value"]
    assert abs(b["mrays_all"] - (b["value"] + b["mrays_primary"])) < 1e-6 * b["mrays_all"]
    assert abs(b["mrays_all"] * 1e6 / (b["samples_per_s"] * b["rays_per_sample"]) - 1) < 1e-6
    assert abs(b["mrays_primary"] * 1e6 / b["samples_per_s"] - 1) < 1e-6


def test_shade_roofline_object():
    r = newest_line()["roofline_shade"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_vertex", "vertices_per_launch", "avg_launch_ms"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 192.0 <= r["bytes_per_vertex"] <= 384.0
    assert abs(r["achieved"] - r["bytes_per_vertex"] * r["vertices_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]


def test_roofline_object():
    r = newest_line()["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["peak"] == 8000.0  # HBM, MI355X_MICROARCH.md
    # achieved = algorithmic bytes per launch / average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["algorithmic_bytes_per_launch"] - r["bytes_per_ray"] * r["rays_per_launch"]) < 1e-6 * r["algorithmic_bytes_per_launch"]
    assert r["traffic"] is None or r["traffic"] > 0


def test_cpu_baseline_object():
    c = newest_line()["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == "Mrays/s" and c["cores"] >= 1 and c["value"] > 0
