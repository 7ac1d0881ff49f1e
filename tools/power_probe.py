#!/usr/bin/env python3
"""Developer tool: sustained clock / power while one kernel runs back to back (is the kernel power-limited?).
usage: tools/power_probe.py <workload> [seconds]   -- launches bench.py's workload in a loop and samples rocm-smi."""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import babyjubjub_rs_amd as bjj
import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "fixed_base"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dev = torch.device("cuda", 0)
ctx = bjj.Context(0, int(os.environ.get("W", "28")))
n = 1 << 20
ctx.reserve(n)
st = torch.cuda.Stream(device=dev)
wl = bench.Workload(ctx, kind, n, 0, dev, st)
if os.environ.get("REPEAT") == "1" and kind == "fixed_base":   # every lane multiplies by the same scalar
    wl.batches[0].d_sc.view(n, 32)[:] = wl.batches[0].d_sc.view(n, 32)[0].clone()
    torch.cuda.synchronize()
for _ in range(3):
    wl.launch()
st.synchronize()
samples = []
stop = False


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
        keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l or "junction" in l.lower()) and "GPU[0]" in l]
        samples.append(keep)
        time.sleep(0.5)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
launches = 0
while time.time() - t0 < secs:
    for _ in range(50):
        wl.launch()
    st.synchronize()
    launches += 50
dt = time.time() - t0
stop = True
th.join()
print("%s: %d launches in %.2f s -> %.3f ms per launch (W=%d)" % (kind, launches, dt, dt / launches * 1e3, ctx.info().window_bits))
for s in samples[1:-1][:6]:
    print("   ", " | ".join(s))
ctx.close()
