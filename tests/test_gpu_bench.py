"""bench.py's N > 1 path on the single-GPU box: `python bench.py --gpus 2` must start its two ranks itself (child
processes, before the parent touches a GPU), shard the batch, and report n_gpus == 2.  The ranks share the one GPU through
the developer mode (gloo; RCCL refuses two ranks per device); on an N-GPU box the same command needs no launcher either."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra, env_extra=None, timeout=1200):
    """runs bench.py; returns (process, DETAIL record or None).  The detail record (--detail-out) is what the assertions below read;
    the stdout contract is checked HERE for every run: exactly one JSON line, printed last, under benchline.HARD_LIMIT bytes, and
    equal to benchline.compact(detail) -- the round-5 line was 24 KB and the driver could not parse it.  `r.compact` keeps it."""
    from babyjubjub_rs_amd import benchline
    env = dict(os.environ)
    env.update(env_extra or {})
    with tempfile.TemporaryDirectory() as td:
        det = os.path.join(td, "detail.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + ["--detail-out", det], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        detail = json.load(open(det)) if os.path.exists(det) else None
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    r.compact = None
    if detail is None:
        assert lines == []
        return r, None
    assert len(lines) == 1 and r.stdout.rstrip("\n").splitlines()[-1] == lines[0], r.stdout[-2000:]
    assert len(lines[0]) < benchline.HARD_LIMIT, len(lines[0])
    r.compact = json.loads(lines[0])
    assert r.compact == json.loads(benchline.dumps(benchline.compact(detail, det)))
    assert "shed_blocks" not in r.compact                      # the budget holds without dropping anything
    assert ("bench_detail: {" in r.stderr)
    return r, detail


def test_bench_gpus_2_self_launch_shared_gpu():
    r, j = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--warmup-seconds", "0", "--batch", "65536", "--batches", "2",
                 "--window-bits", "16", "--strong-total", str(1 << 18), "--no-cpu-baseline"],
                {"BJJ_BENCH_SHARE_GPU": "1", "BJJ_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and len(j["devices"]) == 2 and j["backend"] == "gloo"
    assert j["parity_sample_ok"] is True and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["global_batch"] == 2 * 65536
    assert j["also"]["verify"]["parity_sample_ok"] and j["also"]["var_base"]["parity_sample_ok"]
    s = j["strong"]
    assert s["fixed_base_1M_total"]["total_items"] == 1 << 20
    rr = s["verify_total"]["rank0_resident"]
    assert s["verify_total"]["total_items"] == 1 << 18 and rr["gathered_verdicts_ok"]
    # the pipelined schedule is the reported one, the serial one is timed next to it; 2^17 items per peer block = 4 pieces
    assert rr["pieces_per_peer_block"] == 4 and rr["serial_schedule"]["ms_per_step"] > 0 and rr["ms_per_step"] > 0


def test_bench_gpus_3_ragged_strong_total():
    """three ranks and a total that does not divide: exact-size blocks out, verdicts back, every verdict checked on rank 0"""
    r, j = _run(["--gpus", "3", "--steps", "2", "--warmup", "1", "--warmup-seconds", "0", "--batch", "20000", "--batches", "1",
                 "--window-bits", "12", "--strong-total", "100003", "--no-cpu-baseline", "--no-also"],
                {"BJJ_BENCH_SHARE_GPU": "1", "BJJ_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert j["n_gpus"] == 3 and j["parity_sample_ok"] is True and len(j["devices"]) == 3
    v = j["strong"]["verify_total"]
    assert v["total_items"] == 100003 and v["rank0_resident"]["gathered_verdicts_ok"]


def test_bench_gpus_8_node_shape_shared_gpu():
    """eight ranks -- the node size of the scaling bench and of BASELINE configs[4] -- sharing the one GPU: every rank its own
    context, a ragged total cut into 8 blocks, peer blocks in pieces, every gathered verdict checked on rank 0"""
    r, j = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--warmup-seconds", "0", "--batch", "8192", "--batches", "1",
                 "--window-bits", "12", "--strong-total", str((1 << 19) + 5), "--no-cpu-baseline", "--no-also"],
                {"BJJ_BENCH_SHARE_GPU": "1", "BJJ_BENCH_BACKEND": "gloo"}, 1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and len(j["devices"]) == 8 and j["parity_sample_ok"] is True
    assert j["config"]["global_batch"] == 8 * 8192 and j["scaling"] == "weak"
    v = j["strong"]["verify_total"]
    assert v["total_items"] == (1 << 19) + 5 and v["rank0_resident"]["gathered_verdicts_ok"]
    assert v["rank0_resident"]["pieces_per_peer_block"] == 2          # 65 537-item blocks: two pieces of >= 32 768 items


def test_bench_rejects_world_size_mismatch():
    r, j = _run(["--gpus", "2", "--no-also", "--no-strong"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, 300)
    assert r.returncode != 0 and j is None and "WORLD_SIZE=1 but --gpus 2" in r.stderr


def test_bench_one_gpu_line_has_every_block():
    r, j = _run(["--steps", "20", "--warmup", "5", "--warmup-seconds", "0.2", "--batch", str(1 << 16), "--window-bits", "16",
                 "--strong-total", str(1 << 18)])
    assert r.returncode == 0, r.stderr[-3000:]
    assert j["n_gpus"] == 1 and j["parity_sample_ok"] and j["roofline"]["frac"] > 0 and j["cpu_baseline"]["value"] > 0
    # the compact stdout line: the contract's keys with numbers, every other workload as one small row
    c = r.compact
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_sample_ok"):
        assert k in c, k
    assert c["dtype"] == "u32" and "configs[1]" in c["config"]["workload"] and c["config"]["kernel"] == "bjj_k_mul_fixed_base_2x256"
    assert set(c["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg"}
    assert set(c["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and len(c["cpu_baseline"]["sample"]) <= 150
    assert abs(c["value"] / j["value"] - 1) < 1e-5 and abs(c["ms_per_step"] / j["ms_per_step"] - 1) < 1e-5
    for k in ("verify", "var_base"):
        assert set(c["also"][k]) >= {"value", "unit", "ms_per_step", "roofline_frac", "valu_frac", "kernel"} or j["also"][k].get("valu") is None
    assert c["also"]["fixed_base_window_bits_23"]["value"] > 0 and c["also"]["host_api"]["fixed_base"]["value"] > 0
    assert c["strong"]["fixed_base_1M_total"]["value"] > 0 and len(json.dumps(c)) < 4096
    assert j["rotating_batches"] == 4 and j["single_batch_kernel_ms"] > 0
    # headline: two-stream protocol, the one-stream control (per-launch HIP events, median) next to it; roofline describes ONE launch
    assert j["streams"] == 2 and j["device_ms_per_launch"] > 0
    assert j["single_stream"]["per_launch_event_ms"]["median_ms"] > 0 and j["single_stream"]["per_launch_event_ms"]["value_from_median"] > 0
    assert abs(j["roofline"]["kernel_ms_avg"] - j["single_stream"]["kernel_ms_avg"]) < 1e-9
    for k in ("verify", "var_base"):
        assert j["also"][k]["roofline"]["kernel_ms_avg"] > 0 and j["also"][k]["cpu_baseline"]["cores"] >= 1
        # two-stream protocol for the kernels whose launch is a non-integral number of rounds, one-stream control next to it
        assert j["also"][k]["streams"] == 2 and j["also"][k]["single_stream"]["kernel_ms_avg"] > 0
    assert j["also"]["fixed_base_window_bits_23"]["parity_sample_ok"]
    assert j["config"]["init_ms"] > 0 and j["config"]["table_bytes"] > 0
    # round 4: clock / power sampled across the timed regions; counters only for the build they were taken from; PCIe-inclusive
    # host-pointer rates beside the line
    assert "clock_mhz" in j and "socket_w" in j and isinstance(j["clock"], dict)
    if j["clock"].get("available"):
        assert 300 < j["clock_mhz"] < 3000 and j["socket_w"] > 50 and j["clock"]["samples"] >= 1
        assert j["single_stream"]["clock"]["sclk_mhz"] > 300
    assert j["roofline"]["traffic"] is None          # 2^16-item batch / 16-bit table: the committed counters describe another configuration
    h = j["also"]["host_api"]
    assert h["parity_sample_ok"] and h["fixed_base"]["value"] > 0 and h["verify"]["value"] > 0 and "never" in h["note"]
    # round 5: the kernel that makes `value` is named and has its own roofline block next to the one-stream one; the host-pointer
    # rows are measured on pinned caller memory (copied directly) and on pageable memory (staged by the copy workers)
    assert j["config"]["streams"] == 2 and j["config"]["kernel"] == "bjj_k_mul_fixed_base_2x256" == j["kernel"]
    assert j["roofline"]["kernel"] == "bjj_k_mul_fixed_base" == j["single_stream"]["kernel"]
    ro = j["roofline_overlapped"]
    assert ro["kernel"] == "bjj_k_mul_fixed_base_2x256" and ro["streams"] == 2 and ro["frac"] > 0
    assert abs(ro["span_ms_per_launch"] - j["device_ms_per_launch"]) < 1e-9
    # one overlapped launch takes longer than the span per launch (two are co-resident); the separate per-launch pass agrees with the timed region
    assert ro["kernel_ms_avg"] > 0.8 * ro["span_ms_per_launch"] and 0.5 < ro["per_launch_pass"]["span_ms_per_launch"] / ro["span_ms_per_launch"] < 2.0
    assert j["also"]["var_base"]["roofline_overlapped"]["kernel"] == "bjj_k_mul_var_base" and j["also"]["var_base"]["roofline"]["kernel"] == "bjj_k_mul_var_base_tiles"
    assert j["also"]["verify"]["roofline_overlapped"]["kernel"] == "bjj_k_eddsa_verify_groups"
    assert (h["fixed_base"]["arrays_direct"], h["fixed_base"]["arrays_staged"]) == (2, 0) and (h["verify"]["arrays_direct"], h["verify"]["arrays_staged"]) == (5, 0)
    hp = h["pageable"]
    assert (hp["fixed_base"]["arrays_direct"], hp["fixed_base"]["arrays_staged"]) == (0, 2) and hp["verify"]["arrays_staged"] == 5 and h["copy_threads"] >= 1
    # ... the verify row on the cfg-4 workload (off-curve items included), held against one device-pointer launch on the same inputs;
    # the variable-base row (Point::mul_scalar on the caller's own points)
    assert h["verify"]["device_one_launch_ms"] > 0 and 0.5 < h["verify"]["vs_device_one_launch"] < 5.0 and "cfg-4" in h["note"]
    assert (h["var_base"]["arrays_direct"], h["var_base"]["arrays_staged"]) == (3, 0) and hp["var_base"]["arrays_staged"] == 3
    assert h["var_base"]["value"] > 0 and h["var_base"]["vs_device_one_launch"] > 0.5
    # round 6: the compressed-output form of K1 on device pointers (same two-stream protocol) and through the host-pointer pipeline
    fc = j["also"]["fixed_base_compressed"]
    assert fc["parity_sample_ok"] and fc["kernel"] == "bjj_k_mul_fixed_base_2x256_c32" and fc["value"] > 0.5 * j["value"]
    assert h["fixed_base_compressed"]["value"] > 0 and h["fixed_base_compressed"]["bytes_moved"] == (1 << 16) * 64
    assert (h["fixed_base_compressed"]["arrays_direct"], hp["fixed_base_compressed"]["arrays_staged"]) == (2, 2)
    assert r.compact["also"]["host_api"]["fixed_base_compressed"]["value"] > 0 and r.compact["also"]["fixed_base_compressed"]["value"] > 0
    assert h["verify_compressed"]["value"] > 0 and 0.5 < h["verify_compressed"]["vs_device_one_launch"] < 5.0 and hp["verify_compressed"]["arrays_staged"] == 4
    w23 = j["also"]["fixed_base_window_bits_23"]
    assert w23["kernel"] == "bjj_k_mul_fixed_base" and w23["streams"] == 1 and "clock_mhz" in w23 and w23["init_ms"] > 0
    if j["clock"].get("available"):
        assert j["clock"]["poll_period_ms"] >= 1.0 and j["clock"]["polled_during_timed_region"] is True


def test_bench_point_add_and_compress_workloads():
    """the reference's remaining criterion cases (benches/bench_babyjubjub.rs:26-31, 40-41) as bench workloads"""
    for wl, unit in (("point_add", "point additions/s"), ("compress", "points/s")):
        r, j = _run(["--workload", wl, "--steps", "10", "--warmup", "2", "--warmup-seconds", "0.1", "--batch", str(1 << 16), "--window-bits", "16",
                     "--no-also", "--no-strong"])
        assert r.returncode == 0, r.stderr[-3000:]
        assert j["parity_sample_ok"] and j["unit"] == unit and j["roofline"]["achieved"] > 0 and j["cpu_baseline"]["value"] > 0


def test_bench_headline_survives_unfinished_optional_sections():
    """the watchdog of bench.Headline end to end: with a deadline the optional sections cannot meet, the line still carries the
    measured headline (value, roofline, parity of the headline's own oracle sample), a note, and the exit code is 4 (cut short:
    neither success nor the parity failure's 3)"""
    r, j = _run(["--steps", "10", "--warmup", "2", "--warmup-seconds", "0.1", "--batch", str(1 << 16), "--window-bits", "16",
                 "--strong-total", str(1 << 18), "--no-cpu-baseline"], {"BJJ_BENCH_OPTIONAL_DEADLINE_S": "0.001"}, 600)
    assert r.returncode == 4, r.stderr[-3000:]
    assert j["value"] > 0 and j["parity_sample_ok"] is True and j["roofline"]["frac"] > 0
    assert "abandoned" in j["optional_sections"] and "strong" not in j
