"""The committed bench lines (profiles/r04/bench.json at N = 1, the 2- / 4-rank gloo rehearsals) keep the driver's contract and are
internally consistent: every field of the JSON-line contract, `roofline` and `cpu_baseline` as the task statement spells them,
one regime per figure (kernel time <= step time), frac = achieved / peak, algorithmic bytes = 12 Ne + 64 Nn + 8, value = elements /
step time.  A CPU test: it reads records, it measures nothing."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R04 = os.path.join(ROOT, "profiles", "r04")


def _line(name):
    with open(os.path.join(R04, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_single_gpu_line_keeps_the_contract_and_is_consistent():
    d = _line("bench.json")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None           # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic"
    c = d["config"]
    assert "workload" in c and "model" not in c
    ne, nn = c["elements"], c["nodes"]
    assert d["value"] == pytest.approx(ne / (d["ms_per_step"] * 1e-3), rel=1e-9)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["alg_bytes_per_launch"] == 12 * ne + 64 * nn + 8
    assert r["achieved"] == pytest.approx(r["alg_bytes_per_launch"] / (r["kernel_us"] * 1e-6) / 1e9, rel=1e-9)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-12)
    assert r["kernel_us"] <= d["ms_per_step"] * 1e3, "the top-level roofline is the regime the timed step runs in"
    assert r["traffic"] is None or 0.9 <= r["traffic"] / r["alg_bytes_per_launch"] <= 1.2, "traffic far from the algorithmic bytes"
    assert "measured in this run" in r["traffic_kind"] or "committed" in r["traffic_kind"]
    h = r["hbm_regime"]
    assert h["kernel_us"] >= r["kernel_us"] and h["frac"] == pytest.approx(r["alg_bytes_per_launch"] / (h["kernel_us"] * 1e-6) / 8e12, rel=1e-9)
    assert h["kernel_us"] <= h["step"]["ms_per_step"] * 1e3
    b = d["cpu_baseline"]
    assert b["kind"] in ("port", "reference") and b["cores"] >= 1 and b["value"] > 0 and b["unit"] == d["unit"] and b["sample"]
    assert b["loss_rel_err_vs_gpu"] <= 1e-12
    # the legs the round added are present and plausible
    assert 0.9 < c["lbfgs_step_1gpu"]["ms_per_inner_iteration"] < 2.0 and len(c["lbfgs_sharded_emulated"]) == 3
    assert c["fp32_rows"]["kernel_us"] < c["fp32_rows"]["fp64_arithmetic"]["kernel_us"]
    assert c["train_step_1gpu"]["one_launch"]["us_per_iteration"] < c["train_step_1gpu"]["two_launch"]["us_per_iteration"]
    assert {e["key"] for e in c["extra"]} >= {"Q1M", "T2M", "cfg5", "cfg5auto", "cfg5u"}
    assert "wall_s" in c and c["wall_s"]["cpu_baseline"] < 450, "the default run fits the driver's limit with room to spare"


@pytest.mark.parametrize("n", [2, 4])
def test_rehearsed_multi_rank_lines(n):
    d = _line(f"bench_{n}rank_gloo_one_gpu.json")
    c = d["config"]
    assert d["n_gpus"] == n and d["scaling"] == "weak" and c["elements"] == n * c["elements_per_gpu"]
    assert d["value"] == pytest.approx(c["elements"] / (d["ms_per_step"] * 1e-3), rel=1e-9)      # whole-job aggregate
    assert "strong_scaling" in c and "lbfgs_step" in c and c["peer_exchange"]["state"].startswith("verified in this run")
    assert c["plan_cache"]["per_rank_model_plan_and_sharded_plan"][0] == ["miss", "miss"]
    assert all(v == ["hit", "hit"] for v in c["plan_cache"]["per_rank_model_plan_and_sharded_plan"][1:]), "only rank 0 builds host plans"
