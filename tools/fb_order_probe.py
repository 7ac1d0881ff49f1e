#!/usr/bin/env python3
"""Developer probe: does the ORDER of host-pointer fixed-base calls in a process matter?  affine / compressed / affine / compressed ..., pinned, 2^20 items."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
n = 1 << 20
W = int(sys.argv[1]) if len(sys.argv) > 1 else 23
first = sys.argv[2] if len(sys.argv) > 2 else "affine"
seq = sys.argv[3].split(",") if len(sys.argv) > 3 else None      # e.g. "a1,c,a,c": a/c = timed blocks, a1/c1 = one call, s = small affine call
ctx = bjj.Context(0, W)
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
h_sc, o32, o64 = ctx.host_empty(n * 32), ctx.host_empty(n * 32), ctx.host_empty(n * 64)
h_sc[:] = sc
def best(f, reps=9, warm_s=0.4):
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < warm_s: f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3
import glob, threading
import torch
_pr = torch.cuda.get_device_properties(0)
_BUS = "%04x:%02x:%02x.0" % (_pr.pci_domain_id, _pr.pci_bus_id, _pr.pci_device_id)
def dpm():
    out = []
    for f in sorted(glob.glob("/sys/bus/pci/devices/%s/pp_dpm_*" % _BUS)) + sorted(glob.glob("/sys/bus/pci/devices/%s/current_link_*" % _BUS)):
        if "current_link" in f:
            try: out.append("%s=%s" % (os.path.basename(f)[8:], open(f).read().strip()))
            except Exception: pass
            continue
        try:
            cur = [l.strip() for l in open(f) if l.strip().endswith("*")]
            out.append("%s=%s" % (os.path.basename(f)[7:], cur[0].split(":")[1].strip(" *") if cur else "?"))
        except Exception:
            pass
    return " ".join(out)
class Sampler(threading.Thread):
    def __init__(self): super().__init__(daemon=True); self.rows = []; self.on = True
    def run(self):
        while self.on:
            self.rows.append(dpm()); time.sleep(0.05)
calls = {"affine": lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, n, o64.ctypes.data), "a"),
         "compressed": lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base_compressed(ctx.handle, h_sc.ctypes.data, n, o32.ctypes.data), "c")}
order = [first, "compressed" if first == "affine" else "affine"] * 3
if seq:
    names = {"a": "affine", "c": "compressed"}
    for op in seq:
        if op in names:
            smp = Sampler(); smp.start()
            b = best(calls[names[op]])
            smp.on = False; smp.join()
            import collections
            print("W=%d block  %-10s %.3f ms (median %.3f)   dpm during the block: %s" % (W, names[op], b[0], b[1], collections.Counter(smp.rows).most_common(2)), flush=True)
        elif op in ("a1", "c1"):
            calls[names[op[0]]](); print("   one %s call" % names[op[0]], flush=True)
        elif op == "s":
            ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, 64, o64.ctypes.data), "s"); print("   one 64-item affine call", flush=True)
        elif op.startswith("m"):     # m15 / m16 / m18: one affine call of 2^k items
            k = int(op[1:])
            ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, 1 << k, o64.ctypes.data), "m"); print("   one 2^%d-item affine call" % k, flush=True)
        elif op == "sleep":
            time.sleep(1.0); print("   sleep 1 s", flush=True)
    sys.exit(0)
trace = os.environ.get("BJJ_PIPE_TRACE") == "1"
for k in order:
    if trace:
        sys.stderr.write("#### %s (one call after 0.3 s of them)\n" % k); sys.stderr.flush()
        devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2); os.dup2(devnull, 2)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3: calls[k]()
        os.dup2(saved, 2); os.close(devnull); os.close(saved)
        calls[k]()
        continue
    b = best(calls[k])
    print("W=%d %-10s %.3f ms (median %.3f) chunks %d zero-copy %d" % (W, k, b[0], b[1], ctx.info().last_host_chunks, ctx.info().last_host_zero_copy), flush=True)
