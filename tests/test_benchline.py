"""The stdout line of bench.py is babyjubjub-rs_amd/benchline.compact(detail): these tests hold it to its budget on a RECORDED
detail record (tests/golden/bench_detail_r05.json = the 24 KB line of round 5 that the round-end driver could not parse) and
on inflated ones.  CPU only."""
import copy
import json
import os

from conftest import ROOT

from babyjubjub_rs_amd import benchline


def _detail():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_r05.json")))


def test_recorded_round5_record_compacts_under_the_budget():
    d = _detail()
    assert len(json.dumps(d)) > 20000                      # what the driver was handed in round 5
    c = benchline.compact(d, "bench_detail.json")
    line = benchline.dumps(c)
    assert len(line) <= benchline.BUDGET < benchline.HARD_LIMIT and "\n" not in line and "shed_blocks" not in c
    assert json.loads(line) == c


def test_compact_line_keeps_the_contract():
    d = _detail()
    c = benchline.compact(d)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_sample_ok"):
        assert k in c, k
    assert c["unit"] == "scalar mults/s" and c["dtype"] == "u32" and c["scaling"] == "weak" and c["vs_baseline"] is None
    assert c["config"]["workload"] == d["config"]["workload"] and "configs[1]" in c["config"]["workload"]
    assert c["config"]["window_bits"] == 28 and c["config"]["table_bytes"] == 154618823808 and c["config"]["streams"] == 2
    r = c["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["kernel"] == "bjj_k_mul_fixed_base"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and abs(r["traffic"] / d["roofline"]["traffic"] - 1) < 1e-5
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-2
    cb = c["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 16 and cb["value"] > 0 and isinstance(cb["sample"], str) and len(cb["sample"]) <= 150
    # value and time per step survive the rounding to the driver's consistency tolerance
    assert abs(c["value"] / d["value"] - 1) < 1e-5 and abs(c["ms_per_step"] / d["ms_per_step"] - 1) < 1e-5
    assert abs(c["value"] * c["ms_per_step"] * 1e-3 / c["config"]["batch_per_gpu"] - 1) < 1e-4
    # numbers only: no prose blocks reach stdout
    for k in ("note", "traffic_source", "protocol"):
        assert k not in c["roofline"]
    assert "clock" not in c and "streams_note" not in c and "overlap_detail" not in c
    assert c["valu"] == {"frac": 0.663999, "frac_at_measured_clock": 0.723815, "counter_derived_frac": 0.781632}
    assert c["roofline_overlapped"]["kernel"] == "bjj_k_mul_fixed_base_2x256"


def test_every_other_workload_is_one_small_row():
    c = benchline.compact(_detail())
    for k in ("verify", "var_base"):
        row = c["also"][k]
        assert set(row) >= {"value", "unit", "ms_per_step", "roofline_frac", "valu_frac", "kernel"}
        assert len(json.dumps(row)) < 450
        assert all(not isinstance(v, dict) or kk == "cpu_baseline" for kk, v in row.items())
    assert c["also"]["fixed_base_window_bits_23"]["value"] == 1433950000
    h = c["also"]["host_api"]
    assert h["fixed_base"]["value"] > 0 and h["verify"]["vs_device_one_launch"] > 0 and h["pageable"]["var_base"] > 0 and "note" not in h
    s = c["strong"]["verify_16M_total_cfg5"]
    assert s["total_items"] == 1 << 24 and s["ms_per_step"] > 0 and "mode" not in s


def test_multi_rank_strong_rows_keep_what_the_scaling_session_reads():
    d = _detail()
    d["n_gpus"], d["rccl_ranks"], d["backend"], d["devices"] = 8, 8, "nccl", list(range(8))
    d["strong"]["verify_16M_total_cfg5"]["rank0_resident"] = {
        "value": 4.1e8, "unit": "verifies/s", "ms_per_step": 40.9, "gathered_verdicts_ok": True, "pieces_per_peer_block": 4,
        "mode": "x" * 300, "serial_schedule": {"value": 3.5e8, "ms_per_step": 47.9, "scatter_ms": 5.0, "kernel_ms": 36.0, "gather_ms": 1.0,
                                               "mode": "y" * 200}}
    c = benchline.compact(d)
    assert c["strong"]["verify_16M_total_cfg5"]["rank0_resident"] == {"value": 410000000, "ms_per_step": 40.9,
                                                                      "gathered_verdicts_ok": True, "serial_ms": 47.9}
    assert c["devices"] == list(range(8)) and len(benchline.dumps(c)) <= benchline.BUDGET


def test_a_record_that_outgrows_the_budget_sheds_blocks_instead_of_printing_a_long_line():
    d = _detail()
    for i in range(40):                                        # forty more workloads under `also`
        d["also"]["extra_%d" % i] = copy.deepcopy(d["also"]["verify"])
    c = benchline.compact(d)
    assert len(benchline.dumps(c)) <= benchline.BUDGET and c["shed_blocks"] >= 1
    for k in ("metric", "value", "ms_per_step", "config", "roofline", "cpu_baseline", "parity_sample_ok"):
        assert k in c


def test_abandoned_and_failed_records():
    # the watchdog's record: headline + note, no optional sections; a failed parity publishes no value
    c = benchline.compact({"metric": "m", "value": None, "parity_sample_ok": False, "optional_sections": "abandoned after 240 s " + "z" * 400})
    assert c["value"] is None and c["parity_sample_ok"] is False and len(c["optional_sections"]) <= 200 and "also" not in c
    assert benchline.compact({"metric": "m", "value": float("nan")})["value"] is None      # json has no NaN


def test_native_multi_record():
    res = {"metric": "native", "value": 1.23456789e8, "unit": "verifies/s", "n_gpus": 8, "devices": list(range(8)), "transport": "rccl",
           "mode": "w" * 300, "config": {"workload": "BASELINE configs[4] shape", "window_bits": 23}, "parity_sample_ok": True,
           "results": {"verify": {"value": 1.23456789e8, "ms_per_step": 135.9, "scatter_ms": 20.1, "compute_ms": 100.0, "gather_ms": 1.0,
                                  "total_ms": 130.0, "wall_ms": 131.0, "chunks": 4, "rccl_version": 22105, "unit": "verifies/s",
                                  "items": 1 << 24, "steps": 3, "parity_sample_ok": True, "note": "n" * 300,
                                  "serial_schedule": {"value": 1e8, "ms_per_step": 160.0},
                                  "one_context_one_launch": {"value": 6e7, "ms_per_step": 280.0, "parity_sample_ok": True}}}}
    c = benchline.compact(res)
    v = c["results"]["verify"]
    assert v["serial_ms"] == 160.0 and v["one_launch_ms"] == 280.0 and v["value"] == 123457000 and "note" not in v
    assert c["transport"] == "rccl" and len(benchline.dumps(c)) < 1500
