#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the BabyJubJub hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fixed_base|var_base|verify|poseidon5]
                  [--batch B] [--scatter] [--no-cpu-baseline]

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM (SplitMix64 streams of SURVEY.md 8d).  Default workload = BASELINE.json
configs[1]: 2^20 fixed-base scalar multiplications per GPU.  With N > 1 (launched by
torch.distributed.run, one rank per GPU, RCCL) every rank owns its own 2^20-item block of the
global batch -- the path shards with no data-path collective (weak scaling) -- and `value` is
the whole-job rate: N * batch * K / max-over-ranks time.  `--scatter` instead times BASELINE
cfg 5's shape (rank 0 holds everything; RCCL scatter -> kernel -> gather).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : algorithmic bytes per launch / average kernel duration (HIP events on the
                 launch stream) against the 8 TB/s HBM peak
  cpu_baseline : the oracle's C restatement of the reference algorithm ("port": the Rust
                 reference cannot be built in this image) timed on this box's host cores on
                 a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALGO_BYTES = {"fixed_base": 96, "var_base": 160, "verify": 193, "poseidon5": 192,
              "verify_compressed": 129, "decompress": 97, "sign": 160}  # 8(f) row 1: 32 pk + 64 sig + 32 msg -> 1; 32 -> 64 + 1
UNITS = {"fixed_base": "scalar mults/s", "var_base": "scalar mults/s", "verify": "verifies/s", "poseidon5": "hashes/s",
         "verify_compressed": "verifies/s", "decompress": "points/s", "sign": "signatures/s"}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for fixed_base, 20 otherwise)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 50 for fixed_base, 3 otherwise)")
    ap.add_argument("--warmup-seconds", type=float, default=1.0,
                    help="after the W warm-up steps keep launching (untimed) until this much time has passed, so that the "
                         "timed region runs at the sustained clocks: a 0.6 ms kernel is otherwise timed during the DVFS ramp "
                         "(profiles/r01j_bench_warmup_effect.txt)")
    ap.add_argument("--workload", default="fixed_base", choices=sorted(ALGO_BYTES))
    ap.add_argument("--batch", type=int, default=1 << 20, help="items per GPU per step")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--scatter", action="store_true", help="cfg 5 shape: rank-0 resident, RCCL scatter/gather timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workload lines")
    return ap.parse_args()


class Workload:
    """Device-resident synthetic inputs for one rank, and the launch closure."""

    def __init__(self, ctx, kind, n, offset, dev, stream):
        from babyjubjub_rs_amd import workload as w
        self.kind, self.n, self.ctx, self.stream = kind, n, ctx, stream
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)  # noqa: E731
        self.host = {}
        if kind == "fixed_base":
            self.host["scalars"] = w.scalars_254(n, offset)
            self.d_sc = up(self.host["scalars"])
            self.d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        elif kind == "var_base":
            # cfg 3 points: random multiples of B8 made by the (parity-tested) fixed-base kernel
            self.host["scalars"] = w.scalars_254(n, offset)
            self.host["points"] = ctx.mul_fixed_base(w.random_u256(w.SEED_POINTS, n, offset))
            self.d_sc, self.d_pts = up(self.host["scalars"]), up(self.host["points"])
            self.d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        elif kind == "poseidon5":
            self.host["in"] = w.random_u256(w.SEED_MSGS, 5 * n, 5 * offset, top_bits_cleared=3).reshape(n, 160)
            self.d_in = up(self.host["in"])
            self.d_out = torch.empty(n * 32, dtype=torch.uint8, device=dev)
        elif kind == "sign":
            self.host["keys"] = w.random_u256(w.SEED_KEYS, n, offset)
            self.host["msgs"] = w.random_u256(w.SEED_MSGS, n, offset, top_bits_cleared=3)
            self.d_keys, self.d_msgs = up(self.host["keys"]), up(self.host["msgs"])
            self.d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
            self.d_s = torch.empty(n * 32, dtype=torch.uint8, device=dev)
            self.d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
        elif kind == "decompress":
            pts = ctx.mul_fixed_base(w.random_u256(w.SEED_POINTS, n, offset))
            self.host["comp"] = ctx.compress_points(pts)
            self.host["points"] = pts
            self.d_in = up(self.host["comp"])
            self.d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
            self.d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
        else:  # verify / verify_compressed: cfg 4 signatures, 1/64 corrupted
            make_signatures, corrupt = w.make_signatures, w.corrupt
            A, R, S, msg = make_signatures(ctx.mul_fixed_base, ctx.poseidon5, n, offset)
            if kind == "verify":
                self.bad = corrupt(A, R, S, msg, n, offset)
                self.host.update(pk=A, r=R, s=S, msg=msg)
                self.d_pk, self.d_r, self.d_s, self.d_msg = up(A), up(R), up(S), up(msg)
            else:  # wire format: 32-byte pk, 64-byte signature; corruption lands in s / msg / the y bytes
                z = np.zeros((n, 32), np.uint8)
                A_t = np.concatenate([ctx.compress_points(A), z], axis=1)   # corrupt() flips A[i, :32] ...
                R_t = np.concatenate([z, ctx.compress_points(R)], axis=1)   # ... and R[i, 32:]: the compressed bytes
                self.bad = corrupt(A_t, R_t, S, msg, n, offset)
                pkc = np.ascontiguousarray(A_t[:, :32])
                sig = np.concatenate([R_t[:, 32:], S], axis=1)
                self.host.update(pk=pkc, sig=sig, msg=msg)
                self.d_pk, self.d_sig, self.d_msg = up(pkc), up(sig), up(msg)
            self.d_out = torch.empty(n, dtype=torch.uint8, device=dev)

    def launch(self):
        c, n, s = self.ctx, self.n, self.stream.cuda_stream
        if self.kind == "fixed_base":
            c.mul_fixed_base_dev(self.d_sc.data_ptr(), n, self.d_out.data_ptr(), s)
        elif self.kind == "var_base":
            c.mul_var_base_dev(self.d_pts.data_ptr(), self.d_sc.data_ptr(), n, self.d_out.data_ptr(), s)
        elif self.kind == "poseidon5":
            c.poseidon5_dev(self.d_in.data_ptr(), n, self.d_out.data_ptr(), s)
        elif self.kind == "sign":
            c.sign_dev(self.d_keys.data_ptr(), self.d_msgs.data_ptr(), n, self.d_out.data_ptr(), self.d_s.data_ptr(),
                       self.d_ok.data_ptr(), s)
        elif self.kind == "decompress":
            c.decompress_points_dev(self.d_in.data_ptr(), n, self.d_out.data_ptr(), self.d_ok.data_ptr(), s)
        elif self.kind == "verify_compressed":
            c.eddsa_verify_compressed_dev(self.d_pk.data_ptr(), self.d_sig.data_ptr(), self.d_msg.data_ptr(), n,
                                          self.d_out.data_ptr(), s)
        else:
            c.eddsa_verify_dev(self.d_pk.data_ptr(), self.d_r.data_ptr(), self.d_s.data_ptr(), self.d_msg.data_ptr(), n,
                               self.d_out.data_ptr(), s)

    def check_sample(self, orc, count=512):
        """byte-compare a strided sample of the last step's output with the oracle"""
        n = self.n
        idx = np.unique(np.linspace(0, n - 1, min(count, n)).astype(np.int64))
        h = self.host
        if self.kind == "fixed_base":
            got = self.d_out.view(n, 64)[torch.from_numpy(idx).to(self.d_out.device)].cpu().numpy()
            return bool((got == orc.mul_fixed_base(h["scalars"][idx])).all())
        if self.kind == "var_base":
            got = self.d_out.view(n, 64)[torch.from_numpy(idx).to(self.d_out.device)].cpu().numpy()
            return bool((got == orc.mul_var_base(h["points"][idx], h["scalars"][idx])).all())
        if self.kind == "poseidon5":
            got = self.d_out.view(n, 32)[torch.from_numpy(idx).to(self.d_out.device)].cpu().numpy()
            return bool((got == orc.poseidon5(h["in"][idx])).all())
        if self.kind == "sign":
            t = torch.from_numpy(idx).to(self.d_out.device)
            ro, so, oko = orc.sign(h["keys"][idx], h["msgs"][idx])
            return bool((self.d_out.view(n, 64)[t].cpu().numpy() == ro).all()) and \
                bool((self.d_s.view(n, 32)[t].cpu().numpy() == so).all()) and bool(self.d_ok.cpu().numpy().all())
        if self.kind == "decompress":
            got = self.d_out.view(n, 64).cpu().numpy()
            return bool((got == h["points"]).all()) and bool(self.d_ok.cpu().numpy().all()) and \
                bool((got[idx] == orc.decompress(h["comp"][idx])[0]).all())
        if self.kind == "verify_compressed":
            got = self.d_out.cpu().numpy()
            return bool((got[~self.bad] == 1).all()) and bool((got[self.bad] != 1).all()) and \
                bool((got[idx] == orc.verify_compressed(h["pk"][idx], h["sig"][idx], h["msg"][idx])).all())
        got = self.d_out.cpu().numpy()
        ok_mask = bool((got == (~self.bad).astype(np.uint8)).all())
        return ok_mask and bool((got[idx] == orc.verify(h["pk"][idx], h["r"][idx], h["s"][idx], h["msg"][idx])).all())


def timed_steps(wl, steps, warmup, world, warm_s=0.0):
    """W untimed + K timed launches; returns (wall seconds for K steps, mean kernel ms from HIP events).
    warm_s: extra untimed launches until that many seconds have passed (clock warm-up, see --warmup-seconds)."""
    st = wl.stream
    t_w = time.perf_counter()
    for _ in range(warmup):
        wl.launch()
    st.synchronize()
    while time.perf_counter() - t_w < warm_s:
        for _ in range(8):
            wl.launch()
        st.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record(st)
    for k in range(steps):
        wl.launch()
        evs[k + 1].record(st)
    st.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(steps)]
    return dt, float(np.mean(kernel_ms))


def cpu_baseline(kind, wl, budget_cpu_s=25.0):
    """oracle (C restatement of the reference algorithm, "port") on the host cores, bounded sample.
    The thread count is chosen by a short probe (containers often expose more logical CPUs than
    their CPU quota lets run at once); `cores` reports the threads actually used."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    orc = Oracle()
    logical = os.cpu_count() or 1
    h = wl.host

    def run(o, m):
        t0 = time.perf_counter()
        if kind == "fixed_base":
            o.mul_fixed_base(h["scalars"][:m])
        elif kind == "var_base":
            o.mul_var_base(h["points"][:m], h["scalars"][:m])
        elif kind == "poseidon5":
            o.poseidon5(h["in"][:m])
        elif kind == "sign":
            o.sign(h["keys"][:m], h["msgs"][:m])
        elif kind == "decompress":
            o.decompress(h["comp"][:m])
        elif kind == "verify_compressed":
            o.verify_compressed(h["pk"][:m], h["sig"][:m], h["msg"][:m])
        else:
            o.verify(h["pk"][:m], h["r"][:m], h["s"][:m], h["msg"][:m])
        return time.perf_counter() - t0

    # single-thread rate on a small slice (also calibrates the sample size)
    orc.threads = 1
    m1 = {"fixed_base": 1024, "var_base": 1024, "verify": 384, "poseidon5": 2048, "verify_compressed": 384,
          "decompress": 2048, "sign": 512}[kind]
    m1 = min(m1, wl.n)
    dt1 = run(orc, m1)
    rate1 = m1 / dt1
    # probe thread counts with ~0.25 s of single-thread-equivalent work per thread
    best_t, best_rate = 1, rate1
    for t in sorted({2, 4, 8, 16, 32, 64, 128, logical}):
        if t > logical:
            continue
        orc.threads = t
        m = int(min(wl.n, max(t * 8, rate1 * 0.25 * t)))
        r = m / run(orc, m)
        if r > best_rate:
            best_t, best_rate = t, r
    orc.threads = best_t
    sample = int(min(wl.n, max(best_t * 8, rate1 * budget_cpu_s)))
    dt = run(orc, sample)
    return {"value": sample / dt, "unit": UNITS[kind], "cores": best_t, "kind": "port",
            "sample": "first %d items of the same %s batch, oracle/bjj_ref.c (reference algorithm: bit-serial "
                      "double-and-add with unified adds, binary-Euclid inversions, plain Poseidon), %d pthreads "
                      "(best of a thread-count probe; %d logical CPUs visible)" % (sample, kind, best_t, logical),
            "single_thread_value": rate1}, orc


def load_traffic(kind, field="bytes_per_launch"):
    """per-launch HBM bytes (or VALU wave-instructions) from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic.json, written by tools/summarize_profile.py), or None"""
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(kind, {}).get(field)
        except Exception:
            return None
    return None


# the unit that actually bounds this path: VALU issue.  Peak = 1024 SIMDs x 2.4 GHz / 4.54 cycles per
# v_mad_u64_u32 wave-instruction (tools/ubench, profiles/r01_ubench_valu_rates.txt).
VALU_PEAK_GINST = 1024 * 2.4 / 4.54


def main():
    args = parse()
    if args.steps is None:
        args.steps = 200 if args.workload == "fixed_base" else 20
    if args.warmup is None:
        args.warmup = 50 if args.workload == "fixed_base" else 3
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # BJJ_BENCH_BACKEND=gloo + BJJ_BENCH_SHARE_GPU=1 is a developer mode that runs several ranks on ONE
    # GPU to exercise the N > 1 control flow on a single-GPU box (RCCL refuses two ranks per device).
    backend = os.environ.get("BJJ_BENCH_BACKEND", "nccl")
    if os.environ.get("BJJ_BENCH_SHARE_GPU", "0") == "1":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    import babyjubjub_rs_amd as bjj
    ctx = bjj.Context(local_rank, args.window_bits)
    n = args.batch
    ctx.reserve(n)
    stream = torch.cuda.Stream(device=dev)
    kind = args.workload
    wl = Workload(ctx, kind, n, rank * n, dev, stream)

    extra = {}
    if args.scatter and world > 1:
        # cfg 5 shape: time scatter + kernel + gather of rank-0 resident data
        from babyjubjub_rs_amd import shard
        rows = {"fixed_base": [32], "var_base": [64, 32], "verify": [64, 64, 32, 32], "poseidon5": [160],
                "verify_compressed": [32, 64, 32]}[kind]
        outb = {"fixed_base": 64, "var_base": 64, "verify": 1, "poseidon5": 32, "verify_compressed": 1}[kind]
        total = n * world
        full = None
        if rank == 0:
            full = [torch.zeros(total * rb, dtype=torch.uint8, device=dev) for rb in rows]
        tensors = {"fixed_base": ["d_sc"], "var_base": ["d_pts", "d_sc"], "verify": ["d_pk", "d_r", "d_s", "d_msg"],
                   "poseidon5": ["d_in"], "verify_compressed": ["d_pk", "d_sig", "d_msg"]}[kind]
        # assemble the global batch on rank 0 from every rank's block (untimed setup)
        for ti, (name, rb) in enumerate(zip(tensors, rows)):
            g = shard.gather_rows(getattr(wl, name), total, rb, dev)
            if rank == 0:
                full[ti] = g

        def step():
            shards = [shard.scatter_rows(full[i] if rank == 0 else None, total, rb, dev) for i, rb in enumerate(rows)]
            for name, s in zip(tensors, shards):
                getattr(wl, name).copy_(s)
            torch.cuda.current_stream().synchronize()
            wl.launch()
            wl.stream.synchronize()
            return shard.gather_rows(wl.d_out, total, outb, dev)

        for _ in range(args.warmup):
            step()
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        dist.barrier(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kernel_ms = None
        extra["mode"] = "rank0-resident scatter/kernel/gather (BASELINE cfg 5 shape)"
    else:
        dt, kernel_ms = timed_steps(wl, args.steps, args.warmup, world, args.warmup_seconds)

    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    result = None
    if rank == 0:
        info = ctx.info()
        value = world * n * args.steps / dt_max
        result = {
            "metric": "BabyJubJub %s, %d-item batch per GPU" % (
                {"fixed_base": "fixed-base scalar mults/sec", "var_base": "variable-base scalar mults/sec",
                 "verify": "EdDSA-Poseidon verifies/sec", "poseidon5": "Poseidon(t=6) hashes/sec",
                 "verify_compressed": "EdDSA-Poseidon verifies/sec (compressed pk + signature)",
                 "decompress": "point decompressions/sec", "sign": "EdDSA-Poseidon signatures/sec"}[kind], n),
            "value": value, "unit": UNITS[kind], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": {"fixed_base": "1M fixed-base scalar mults (generator B8), BASELINE configs[1]",
                                    "var_base": "1M variable-base scalar mults, BASELINE configs[2]",
                                    "verify": "1M EdDSA-Poseidon verifies, 1/64 corrupted, BASELINE configs[3]",
                                    "poseidon5": "Poseidon t=6 hashes (component of configs[3])",
                                    "verify_compressed": "1M EdDSA-Poseidon verifies on wire-format inputs (SURVEY 8f row 1)",
                                    "decompress": "1M decompress_point (SURVEY 8f row 1)",
                                    "sign": "1M PrivateKey::sign (Blake-512 x2, 2 fixed-base mults, Poseidon; SURVEY 8f row 2)"}[kind],
                       "batch_per_gpu": n, "global_batch": n * world, "window_bits": info.window_bits,
                       "fixed_base_table_mb": info.table_bytes / 1e6,
                       "limbs": "9 x 29-bit, 64-bit column accumulators (v_mad_u64_u32)",
                       "warmup_seconds": args.warmup_seconds,
                       "parallelism": "independent shards, one process per GPU, no data-path collective"},
        }
        result.update(extra)
        if kernel_ms is not None:
            algo = ALGO_BYTES[kind] * n
            ach = algo / (kernel_ms * 1e-3) / 1e9
            result["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": ach / HBM_PEAK_GBPS,
                                  "traffic": load_traffic(kind) if (n == (1 << 20) and load_traffic(kind, "window_bits") in (None, info.window_bits)) else None,
                                  "kernel": {"fixed_base": info.kernel_fixed_base, "var_base": info.kernel_var_base,
                                             "verify": info.kernel_verify, "poseidon5": info.kernel_poseidon5,
                                             "verify_compressed": info.kernel_verify,
                                             "decompress": b"bjj_k_decompress_points", "sign": b"bjj_k_sign"}[kind].decode(),
                                  "kernel_ms_avg": kernel_ms, "algorithmic_bytes_per_launch": algo,
                                  "note": "integer-ALU bound path (see DESIGN.md): HBM fraction is reported as measured"}
            vi = load_traffic(kind, "valu_insts_per_launch")
            if vi and n == (1 << 20) and load_traffic(kind, "window_bits") in (None, info.window_bits):
                va = vi / (kernel_ms * 1e-3) / 1e9
                result["valu"] = {"insts_per_launch": vi, "achieved": va, "peak": VALU_PEAK_GINST, "frac": va / VALU_PEAK_GINST,
                                  "unit": "G wave-instructions/s",
                                  "note": "SQ_INSTS_VALU of the rocprofv3 PMC pass / live kernel time; peak = measured v_mad_u64_u32 issue rate"}
        orc = None
        if not args.no_cpu_baseline and world == 1:
            cb, orc = cpu_baseline(kind, wl)
            result["cpu_baseline"] = cb
        if orc is None:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from conftest import Oracle
            orc = Oracle()
        result["parity_sample_ok"] = wl.check_sample(orc)
        if not args.no_also and not args.scatter and world == 1:  # secondary lines only in the 1-GPU run
            also = {}
            for k2, n2, s2 in (("verify", n, 5), ("var_base", n, 5)):
                if k2 == kind:
                    continue
                w2 = Workload(ctx, k2, n2, rank * n2, dev, stream)
                d2, km2 = timed_steps(w2, s2, 1, 1, 0.5)
                also[k2] = {"value_one_gpu": n2 * s2 / d2, "unit": UNITS[k2], "kernel_ms_avg": km2, "batch": n2,
                            "parity_sample_ok": w2.check_sample(orc, 128)}
                del w2
            result["also"] = also
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
    ctx.close()


if __name__ == "__main__":
    main()
