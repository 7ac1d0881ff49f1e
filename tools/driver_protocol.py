#!/usr/bin/env python3
"""Developer tool: A/B under the DRIVER'S protocol.  The round-end record (BENCH_rNN.json) comes from
    python3 bench.py --gpus 1 --steps 20 --warmup 5
so that exact command is what K1 is tuned under: this tool runs it ROUNDS times per variant, interleaved, each run a fresh
process (fresh context, fresh clocks), and prints one row per run plus per-variant medians.

usage: tools/driver_protocol.py [--rounds 8] [--workload fixed_base] [--full] name=ENV1=V1,ENV2=V2[:extra bench args] ...
  a variant is `name=` followed by comma-separated environment assignments (may be empty) and, after a colon, extra bench.py
  arguments, e.g.   shipped=   v0=BJJ_K1_VARIANT=0   v1=BJJ_K1_VARIANT=1   one=:--streams 1   lib=BJJ_LIB_PATH=tools/ab_x.so
  --full keeps the optional sections of the line (also / strong / cpu_baseline: ~90 s per run); without it they are
  switched off -- the headline is measured first either way, under the identical protocol."""
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, workload, full, variants = 8, "fixed_base", False, []
    steps, warmup = "20", "5"
    while args:
        a = args.pop(0)
        if a == "--rounds":
            rounds = int(args.pop(0))
        elif a == "--workload":
            workload = args.pop(0)
        elif a == "--steps":
            steps = args.pop(0)
        elif a == "--warmup":
            warmup = args.pop(0)
        elif a == "--full":
            full = True
        else:
            name, _, rest = a.partition("=")
            envs, _, extra = rest.partition(":")
            env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
            variants.append((name, env, extra.split() if extra else []))
    if not variants:
        variants = [("shipped", {}, [])]
    rows = {name: [] for name, _, _ in variants}
    for r in range(rounds):
        for name, env, extra in variants:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", steps, "--warmup", warmup]
            if workload != "fixed_base":
                cmd += ["--workload", workload]
            if not full:
                cmd += ["--no-also", "--no-strong", "--no-cpu-baseline"]
            cmd += extra
            e = dict(os.environ)
            for k, v in env.items():
                e[k] = os.path.join(ROOT, v) if k == "BJJ_LIB_PATH" and not os.path.isabs(v) else v
            t0 = time.time()
            p = subprocess.run(cmd, env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            # the full record is on stderr (`bench_detail: {...}`); stdout carries the compact line the driver parses
            line = [ln[len("bench_detail: "):] for ln in p.stderr.splitlines() if ln.startswith("bench_detail: {")]
            if p.returncode != 0 or not line:
                print("round %d %-10s FAILED rc=%d %s" % (r + 1, name, p.returncode, p.stderr[-300:]))
                continue
            d = json.loads(line[-1])
            ss = d.get("single_stream") or {}
            pl = (ss.get("per_launch_event_ms") or d.get("per_launch_event_ms") or {})
            row = {"value": d["value"], "ms_per_step": d["ms_per_step"], "dev_ms": d.get("device_ms_per_launch"),
                   "one_stream_value": ss.get("value_this_rank"), "one_stream_median_ms": pl.get("median_ms"),
                   "kernel_ms_avg": d["roofline"]["kernel_ms_avg"], "parity": d.get("parity_sample_ok"),
                   "clock_mhz": (d.get("clock") or {}).get("sclk_mhz"), "socket_w": (d.get("clock") or {}).get("socket_w"),
                   "wall_s": time.time() - t0}
            rows[name].append(row)
            f = lambda v, fmt: (fmt % v) if v is not None else "   -   "  # noqa: E731
            print("round %d %-10s value %s  ms/step %s  dev ms/launch %s | one stream: %s  median launch %s ms | kernel_ms_avg %s  "
                  "sclk %s W %s parity %s (%.0f s)"
                  % (r + 1, name, f(row["value"] / 1e6, "%8.2f M/s"), f(row["ms_per_step"], "%.4f"), f(row["dev_ms"], "%.4f"),
                     f(row["one_stream_value"] and row["one_stream_value"] / 1e6, "%8.2f M/s"), f(row["one_stream_median_ms"], "%.4f"),
                     f(row["kernel_ms_avg"], "%.4f"), f(row["clock_mhz"], "%.0f"), f(row["socket_w"], "%.0f"), row["parity"], row["wall_s"]))
            sys.stdout.flush()
    print("---- medians over %d rounds (driver protocol: --steps %s --warmup %s, fresh process per run)" % (rounds, steps, warmup))
    base = None
    for name, _, _ in variants:
        v = [x["value"] for x in rows[name]]
        if not v:
            continue
        med = statistics.median(v)
        base = base or med
        o = [x["one_stream_value"] for x in rows[name] if x["one_stream_value"]]
        print("%-10s value median %8.2f M/s  (min %8.2f, max %8.2f)  %+5.1f %% vs %s | one-stream control median %s"
              % (name, med / 1e6, min(v) / 1e6, max(v) / 1e6, (med / base - 1) * 100, variants[0][0],
                 ("%8.2f M/s" % (statistics.median(o) / 1e6)) if o else "-"))


if __name__ == "__main__":
    main()
