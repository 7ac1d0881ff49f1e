#!/usr/bin/env python3
"""Developer tool: variable-base kernel time against (a) the point population (prime-order subgroup only vs the whole
group, SURVEY 8d cfg 3) and (b) the number of resident batches the timed loop rotates over."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import babyjubjub_rs_amd as bjj
import bench
from babyjubjub_rs_amd import workload as w

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = bjj.Context(0, int(os.environ.get("W", "28")))
n = 1 << 20
ctx.reserve(n)
st = torch.cuda.Stream(device=dev)
full = bench.Workload(ctx, "var_base", n, 0, dev, st, nb=4)
# subgroup-only variant of batch 0: P = k * B8
sub = bench.Workload.__new__(bench.Workload); sub.__dict__.update(full.__dict__)
B = bench.Batch(); B.__dict__.update(full.batches[0].__dict__)
d_k = torch.from_numpy(w.random_u256(w.SEED_POINTS, n, 0).reshape(-1)).to(dev)
B.d_pts = torch.empty(n * 64, dtype=torch.uint8, device=dev)
ctx.mul_fixed_base_dev(d_k.data_ptr(), n, B.d_pts.data_ptr(), 0); ctx.sync()
sub.batches = [B]
one = bench.Workload.__new__(bench.Workload); one.__dict__.update(full.__dict__); one.batches = full.batches[:1]
two = bench.Workload.__new__(bench.Workload); two.__dict__.update(full.__dict__); two.batches = full.batches[:2]
for rnd in range(3):
    for name, wl in (("subgroup points, 1 batch", sub), ("whole group, 1 batch", one), ("whole group, 2 batches", two), ("whole group, 4 batches", full)):
        dt, km = bench.timed_steps(wl, 20, 3, 1, 1.0)
        print("round %d  %-26s kernel %.3f ms" % (rnd, name, km), flush=True)
