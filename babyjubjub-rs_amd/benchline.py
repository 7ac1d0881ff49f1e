"""The ONE line bench.py prints on stdout.

bench.py measures far more than a driver needs to parse (per-launch event statistics, clock blocks, notes, the one-stream
control next to the two-stream protocol, ...).  Round 5's line grew to 24 KB and the round-end driver, which keeps a bounded
tail of stdout, could no longer parse it: the headline went unmeasured in the only record the builder does not write.  So:

  * everything measured goes to a DETAIL file (bench_detail.json next to bench.py, `--detail-out`) and to stderr;
  * stdout gets exactly one line, `compact(detail)`: the contract's keys + `roofline` + `cpu_baseline` with numbers only,
    every other workload reduced to {value, unit, ms_per_step, roofline_frac, valu_frac, kernel}.  Budget: <= BUDGET bytes
    (HARD_LIMIT is what the tests enforce); if a future block pushes the line over the budget, `compact` sheds the optional
    blocks (strong detail, also detail, ...) in a fixed order rather than print a long line.

No GPU, no torch: tests/test_benchline.py drives this on recorded detail files.
"""
import json

BUDGET = 4096       # what compact() aims for
HARD_LIMIT = 6000   # what tests assert on every line bench.py prints


def _num(x, sig=6):
    """floats to `sig` significant digits (a 64-bit repr is 18 characters; 6 digits is beyond every measurement's noise)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    r = float("%.*g" % (sig, x))
    return int(r) if r.is_integer() and abs(r) >= 1e6 else r


def _round(o):
    if isinstance(o, dict):
        return {k: _round(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round(v) for v in o]
    return _num(o)


def _pick(d, keys):
    d = d or {}
    return {k: d[k] for k in keys if k in d}


def _short(s, limit):
    if not isinstance(s, str) or len(s) <= limit:
        return s
    return s[:limit - 3] + "..."


def _roofline(r):
    return _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "algorithmic_bytes_per_launch",
                     "stale_profile"))


def _valu(v):
    if not v:
        return None
    out = _pick(v, ("frac", "frac_at_measured_clock"))
    cd = (v.get("counter_derived") or {}).get("frac")
    if cd is not None:
        out["counter_derived_frac"] = cd
    return out


def _cpu_baseline(c):
    if not c:
        return None
    out = _pick(c, ("value", "unit", "cores", "kind", "single_thread_value"))
    out["sample"] = _short(c.get("sample_short") or c.get("sample"), 150)
    return out


def _workload_row(w):
    """one `also` workload: {value, unit, ms_per_step, roofline_frac, valu_frac, kernel} (+ the two facts a reader needs to price
    it: the one-launch kernel time behind roofline_frac and the CPU port's rate)"""
    out = _pick(w, ("value", "unit", "ms_per_step", "kernel", "streams", "parity_sample_ok", "kernel_ms_avg", "table_bytes"))
    rf = w.get("roofline") or {}
    if "frac" in rf:
        out["roofline_frac"] = rf["frac"]
        out["kernel_ms_avg"] = rf.get("kernel_ms_avg")
        if rf.get("traffic") is not None:
            out["traffic"] = rf["traffic"]
    ro = w.get("roofline_overlapped") or {}
    if "frac" in ro:
        out["roofline_overlapped_frac"] = ro["frac"]
    v = w.get("valu") or {}
    if "frac" in v:
        out["valu_frac"] = v.get("frac_at_measured_clock", v["frac"])
    cb = w.get("cpu_baseline") or {}
    if "value" in cb:
        out["cpu_baseline"] = _pick(cb, ("value", "cores", "kind"))
    return out


def _host_api(h):
    def rows(d):
        return {k: _pick(d[k], ("value", "ms_per_call", "vs_device_one_launch")) for k in d
                if isinstance(d.get(k), dict) and "value" in d[k]}
    out = rows(h)
    if isinstance(h.get("pageable"), dict):
        out["pageable"] = {k: v["value"] for k, v in rows(h["pageable"]).items()}
    if isinstance(h.get("single_call"), dict):      # what ONE call of ONE item costs, microseconds (first table)
        out["single_call_us"] = _pick(h["single_call"], ("fixed_base", "var_base", "poseidon5", "verify"))
    out.update(_pick(h, ("parity_sample_ok", "copy_threads")))
    out["unit"] = "items/s, PCIe-inclusive (pinned host pointers); never `value`"
    return out


def _strong_row(s):
    out = _pick(s, ("value", "unit", "total_items", "steps", "ms_per_step"))
    rr = s.get("rank0_resident")
    if rr:
        out["rank0_resident"] = _pick(rr, ("value", "ms_per_step", "gathered_verdicts_ok"))
        ser = rr.get("serial_schedule") or {}
        if "ms_per_step" in ser:
            out["rank0_resident"]["serial_ms"] = ser["ms_per_step"]
    return out


def _native_row(r):
    """one workload of `bench.py --native-multi` (bjj_*_multi_dev): the pipelined schedule's spans, the serial schedule and one
    launch of one context beside it"""
    out = _pick(r, ("value", "unit", "items", "steps", "ms_per_step", "scatter_ms", "compute_ms", "gather_ms", "total_ms", "chunks",
                    "rccl_version", "parity_sample_ok"))
    if "ms_per_step" in (r.get("serial_schedule") or {}):
        out["serial_ms"] = r["serial_schedule"]["ms_per_step"]
    if "ms_per_step" in (r.get("one_context_one_launch") or {}):
        out["one_launch_ms"] = r["one_context_one_launch"]["ms_per_step"]
    return out


def compact(full, detail_path=None):
    """the dict bench.py prints: `full` (the detail record) reduced to the contract's keys.  Pure function of `full`."""
    c = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                     "vs_baseline", "dtype", "data", "rccl_ranks", "backend"))
    cfg = _pick(full.get("config"), ("workload", "batch_per_gpu", "global_batch", "window_bits", "table_bytes", "init_ms", "streams",
                                     "kernel", "resident_batches", "library_default_window_bits", "parallelism", "devices",
                                     "transport", "mode"))
    c["config"] = cfg
    if full.get("roofline"):
        c["roofline"] = _roofline(full["roofline"])
    ro = full.get("roofline_overlapped")
    if ro:
        c["roofline_overlapped"] = _pick(ro, ("kernel", "span_ms_per_launch", "frac", "traffic"))
    v = _valu(full.get("valu"))
    if v:
        c["valu"] = v
    for k in ("clock_mhz", "socket_w"):
        if full.get(k) is not None:
            c[k] = full[k]
    ss = full.get("single_stream") or {}
    if "value_this_rank" in ss:
        c["single_stream"] = {"value": ss["value_this_rank"], "kernel_ms_avg": ss.get("kernel_ms_avg"), "kernel": ss.get("kernel")}
    cb = _cpu_baseline(full.get("cpu_baseline"))
    if cb:
        c["cpu_baseline"] = cb
    c["parity_sample_ok"] = full.get("parity_sample_ok")
    also = full.get("also") or {}
    if also:
        a = {}
        for k, w in also.items():
            if k == "host_api":
                a[k] = _host_api(w)
            elif isinstance(w, dict):
                a[k] = _workload_row(w)
        c["also"] = a
    strong = full.get("strong") or {}
    if strong:
        c["strong"] = {k: _strong_row(s) for k, s in strong.items() if isinstance(s, dict)}
    if isinstance(full.get("results"), dict):
        c["results"] = {k: _native_row(r) for k, r in full["results"].items() if isinstance(r, dict)}
        c.update(_pick(full, ("transport",)))
    for k in ("devices", "optional_sections"):
        if full.get(k) is not None:
            c[k] = _short(full[k], 200) if isinstance(full[k], str) else full[k]
    if detail_path:
        c["detail"] = detail_path
    c = _round(c)

    # never a long line: shed optional blocks, least important first, until the budget holds
    def size(d):
        return len(json.dumps(d, separators=(", ", ": ")))
    shed = [lambda d: [w.pop("cpu_baseline", None) for w in d.get("also", {}).values() if isinstance(w, dict)],
            lambda d: (d.get("also", {}).get("host_api") or {}).pop("pageable", None),
            lambda d: [w.pop(k, None) for w in d.get("also", {}).values() if isinstance(w, dict)
                       for k in ("traffic", "kernel_ms_avg", "roofline_overlapped_frac", "streams")],
            lambda d: [s.pop("rank0_resident", None) for s in d.get("strong", {}).values()],
            lambda d: d.pop("single_stream", None),
            lambda d: d.pop("strong", None),
            lambda d: d.pop("also", None),
            lambda d: d.pop("roofline_overlapped", None)]
    dropped = 0
    for f in shed:
        if size(c) <= BUDGET:
            break
        f(c)
        dropped += 1
    if dropped:
        c["shed_blocks"] = dropped
    return c


def dumps(c):
    return json.dumps(c)
