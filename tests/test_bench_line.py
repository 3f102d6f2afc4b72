"""CPU-side checks of the line bench.py hands the driver (BENCH_r05.json had `parsed: null` -- the line had grown to
21 KB and the driver keeps the last 8 081 bytes of stdout).  The assembler is a pure function: it is fed the committed round-5 line."""
import json
import os

import pytest

from conftest import ROOT

import bench


def canned():
    return json.loads(open(os.path.join(ROOT, "profiles", "r05d_bench.json")).read().strip().splitlines()[-1])


def test_the_round5_line_was_too_long_and_its_summary_is_not():
    full = canned()
    assert len(json.dumps(full)) > 8081                      # what broke the driver's parse
    line = json.dumps(bench.headline_line(full, {"bench_full": "bench_full.json"}))
    assert len(line) < bench.HEADLINE_MAX_BYTES < 8081
    d = json.loads(line)                                     # round-trips
    assert json.dumps(d) == line


def test_summary_keeps_the_contract():
    full = canned()
    d = bench.headline_line(full)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_learner", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-5) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    for k in ("traffic", "kernel_ms", "agents_per_launch", "frac_algorithmic", "frac_physical_stored"):
        assert k in r, k
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and len(c["sample"]) <= 200
    assert c["reference_python"]["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    assert len(d["other_configs"]) == len(full["other_configs"])
    assert all("value" in r or "error" in r for r in d["other_configs"])


def test_summary_survives_pathological_inputs():
    """Long strings anywhere, a hundred other-config rows: the line still fits (rows are dropped before the contract's keys are)."""
    full = canned()
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["other_configs"] = full["other_configs"] * 12
    d = bench.headline_line(full)
    assert len(json.dumps(d)) < bench.HEADLINE_MAX_BYTES
    assert "roofline" in d and "cpu_baseline" in d and "value" in d


def test_side_files_hold_the_whole_result(tmp_path, monkeypatch):
    full = canned()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    names = bench.write_side_files(full)
    assert json.load(open(tmp_path / names["bench_full"])) == full
    assert json.load(open(tmp_path / names["bench_other_configs"])) == full["other_configs"]


def test_step_bytes_model_adds_up_to_what_the_counters_saw():
    """bench.step_bytes_model: the per-plane bytes of a step launch.  Written bytes at configs[1] = the 282 B per agent-step that
    WRITE_SIZE reported (profiles/r06_pmc.txt; the counter is exact on gfx950: profiles/r06_counter_calibration.txt); the totals stay
    within 10 % of the counted traffic at every size a profile was committed for."""
    m = bench.step_bytes_model(80, 16384, policy=True, records=True)
    assert m["written_bytes"] == 176 + 80 + 16 + 4 + 2 + 4 == 282
    assert abs(sum(m["read"].values()) - m["read_bytes"]) < 1e-9 and abs(m["total"] - m["read_bytes"] - m["written_bytes"]) < 1e-9
    summ = json.load(open(os.path.join(ROOT, "profiles", "summary.json")))
    got = summ["envs16384_batch16384_dqn_packed"]
    assert got["k_step_policy_WRITE_SIZE_KB"] * 1024 / 16384 == pytest.approx(282, rel=0.01)
    assert got["k_step_policy_traffic_bytes_per_launch"] / 16384 == pytest.approx(m["total"], rel=0.10)
    for key, n in (("envonly65536_packed", 65536), ("envonly262144_packed", 262144)):
        e = bench.step_bytes_model(80, n, policy=False, records=False)
        assert summ[key]["k_step_traffic_bytes_per_launch"] / n == pytest.approx(e["total"], rel=0.10), key
