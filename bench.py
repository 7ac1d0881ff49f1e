#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the BabyJubJub hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fixed_base|var_base|verify|poseidon5|...]
                  [--batch B] [--window-bits W] [--no-cpu-baseline] [--no-also] [--no-strong] [--native-multi]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM (SplitMix64
streams of SURVEY.md 8d; the timed loop rotates over 4 distinct resident batches).  Default workload = BASELINE.json
configs[1]: 2^20 fixed-base scalar multiplications per GPU, 28-bit windows (154.6 GB table, passed explicitly -- the
library's own default is 23 bits).

N > 1: one process per GPU over RCCL.  `python bench.py --gpus N` launches its N ranks itself (child processes through
torch.distributed.run, before this process touches a GPU); under an external launcher (WORLD_SIZE set) it is a rank.
Every rank owns its own 2^20-item block of the global batch -- the path shards with no data-path collective (weak
scaling) -- and `value` is the whole-job rate: N * batch * K / max-over-ranks time.

OUTPUT.  stdout gets exactly ONE line, printed last: the compact record of babyjubjub-rs_amd/benchline.py (<= 4 KB: the
contract's keys, `roofline`, `cpu_baseline`, every other workload as {value, unit, ms_per_step, roofline_frac, valu_frac,
kernel}).  Everything measured -- per-launch event statistics, clock blocks, notes -- goes to bench_detail.json
(`--detail-out`) and to stderr as one `bench_detail: {...}` line.  The detail record carries
  also    : the EdDSA-verify and variable-base halves of BASELINE's metric (1 M per GPU), with their own roofline /
            cpu_baseline blocks (1-GPU run) -- same steps / warm-up protocol
  strong  : fixed total work split over the N ranks: 2^20 fixed-base mults, and BASELINE configs[4] = 2^24 verifies,
            pre-sharded and in cfg 5's literal shape (rank 0 holds everything; RCCL scatter -> kernels -> gather)
  roofline / cpu_baseline / valu : see DESIGN.md section 6
The process exits non-zero when any oracle sample comparison fails (a miscomputing build must not publish a number).
`--native-multi` instead drives all N GPUs from ONE process through the C ABI's bjj_multi_* entry points
(ncclCommInitAll + grouped ncclScatter / ncclGather inside libbjj_hip.so).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALGO_BYTES = {"fixed_base": 96, "var_base": 160, "verify": 193, "poseidon5": 192,
              "verify_compressed": 129, "decompress": 97, "sign": 160,  # 8(f) row 1: 32 pk + 64 sig + 32 msg -> 1; 32 -> 64 + 1
              "point_add": 192, "compress": 96,  # the reference's other criterion cases: 2 x 64 -> 64; 64 -> 32
              "fixed_base_compressed": 64}  # 8(f) row 1 "compress on output": 32 -> 32
UNITS = {"fixed_base": "scalar mults/s", "var_base": "scalar mults/s", "verify": "verifies/s", "poseidon5": "hashes/s",
         "verify_compressed": "verifies/s", "decompress": "points/s", "sign": "signatures/s", "point_add": "point additions/s",
         "compress": "points/s", "fixed_base_compressed": "scalar mults/s"}
METRIC = {"fixed_base": "fixed-base scalar mults/sec", "var_base": "variable-base scalar mults/sec",
          "verify": "EdDSA-Poseidon verifies/sec", "poseidon5": "Poseidon(t=6) hashes/sec",
          "verify_compressed": "EdDSA-Poseidon verifies/sec (compressed pk + signature)",
          "decompress": "point decompressions/sec", "sign": "EdDSA-Poseidon signatures/sec",
          "point_add": "projective add + affine (criterion case `add`)/sec", "compress": "point compressions/sec",
          "fixed_base_compressed": "fixed-base scalar mults/sec, compressed output"}
WORKLOAD_TEXT = {"fixed_base": "1M fixed-base scalar mults (generator B8), BASELINE configs[1]",
                 "var_base": "1M variable-base scalar mults on random group points, BASELINE configs[2]",
                 "verify": "1M EdDSA-Poseidon verifies, 1/64 corrupted, BASELINE configs[3]",
                 "poseidon5": "Poseidon t=6 hashes (component of configs[3])",
                 "verify_compressed": "1M EdDSA-Poseidon verifies on wire-format inputs (SURVEY 8f row 1)",
                 "decompress": "1M decompress_point (SURVEY 8f row 1)",
                 "sign": "1M PrivateKey::sign (Blake-512 x2, 2 fixed-base mults, Poseidon; SURVEY 8f row 2)",
                 "point_add": "1M p.projective().add(&q.projective()).affine() (benches/bench_babyjubjub.rs:26-31)",
                 "compress": "1M Point::compress (benches/bench_babyjubjub.rs:40-41)",
                 "fixed_base_compressed": "1M B8.mul_scalar(n).compress(), compression fused into K1's epilogue (SURVEY 8f row 1)"}
# the kernel a launch that runs ALONE on one stream gets (what roofline / valu and the rocprofv3 summaries describe)
KERNEL = {"fixed_base": "bjj_k_mul_fixed_base", "var_base": "bjj_k_mul_var_base_tiles", "verify": "bjj_k_eddsa_verify_groups",
          "poseidon5": "bjj_k_poseidon5", "verify_compressed": "bjj_k_eddsa_verify_groups", "decompress": "bjj_k_decompress_points",
          "sign": "bjj_k_sign", "point_add": "bjj_k_point_add", "compress": "bjj_k_compress_points",
          "fixed_base_compressed": "bjj_k_mul_fixed_base_c32"}
# Streams the timed loop alternates over by default.  The context keeps one scratch set per stream (two sets), so with two
# streams consecutive launches overlap: the head of launch k+1 fills the partly empty last wave-round of launch k (verify is
# 8.1 rounds of the resident waves, variable base 5.3), and for K1 the library switches to its two-workgroups-per-CU shape,
# in which one launch's serial section (the workgroup-wide inversion, the epilogue) is covered by the other launch's main loop.
# `value` is the rate of the K timed launches under that protocol; `roofline` / `valu` keep describing ONE launch on one stream
# (per-launch HIP events; what the rocprofv3 summaries under profiles/ profile), and `single_stream` carries its rate.
DEFAULT_STREAMS = {"fixed_base": 2, "verify": 2, "var_base": 2, "verify_compressed": 2, "fixed_base_compressed": 2}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
T8 = (4342719913949491028786768530115087822524712248835451589697801404893164183326,
      4826523245007015323400664741523384119579596407052839571721035538011798951543)  # a point of order 8 (SURVEY.md 8d cfg 3)


class GpuTelemetry:
    """sclk / socket power / junction temperature of one HIP device, sampled from the amdgpu hwmon files in a background thread
    (without the clock and the power in the line, numbers from different boxes cannot be compared -- this
    path is power-capped, and the sustained clock differs from box to box and with how warm the package is).
    Read-only sysfs; no root needed.  `window(t0, t1)` = statistics of the samples whose perf_counter timestamp lies in
    [t0, t1]; when the region is too short to hold two samples the window is widened backwards (and says so)."""

    def __init__(self, device_index, period_s=None):
        import collections
        import glob
        import threading
        # 5 ms by default (BJJ_BENCH_TELEMETRY_MS): the SMU refreshes its metrics about once per millisecond, and a 2 kHz
        # Python poller next to the launch loop competes for the GIL and queries the SMU 2 000 times a second per rank --
        # enough to perturb the number it annotates when 8 ranks share a host.  0 switches the poller off.
        if period_s is None:
            period_s = float(os.environ.get("BJJ_BENCH_TELEMETRY_MS", "5")) * 1e-3
        self.ok, self.why, self.period, self.paused = False, None, period_s, False
        self.samples = collections.deque(maxlen=200000)     # appended by the poller, snapshotted under the lock by window()
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self._thread = None
        if period_s <= 0:
            self.why = "switched off (BJJ_BENCH_TELEMETRY_MS=0)"
            return
        try:
            # the PCI address from torch's device properties (NOT a second dlopen of libamdhip64: torch ships its own copy of
            # the runtime, and a process must not end up with two)
            pr = torch.cuda.get_device_properties(int(device_index))
            bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            hw = sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bus))
            if not hw:
                raise RuntimeError("no hwmon directory for %s" % bus)
            self.dir, self.bus = hw[0], bus
            lab = {}
            for f in glob.glob(self.dir + "/*_label"):
                lab[open(f).read().strip()] = f[:-len("_label")]
            self.f_sclk = (lab.get("sclk") or self.dir + "/freq1") + "_input"
            self.f_power = self.dir + ("/power1_input" if os.path.exists(self.dir + "/power1_input") else "/power1_average")
            self.f_temp = (lab.get("junction") or self.dir + "/temp2") + "_input"
            self.cap_w = None
            try:
                self.cap_w = int(open(self.dir + "/power1_cap").read()) / 1e6
            except Exception:
                pass
            self._read()
            self.ok = True
        except Exception as e:   # no telemetry is not an error of the benchmark
            self.why = "%s: %s" % (type(e).__name__, e)

    def _read(self):
        def rd(p):
            try:
                with open(p) as f:
                    return float(f.read())
            except Exception:
                return float("nan")
        return (time.perf_counter(), rd(self.f_sclk) / 1e6, rd(self.f_power) / 1e6, rd(self.f_temp) / 1e3)

    def start(self):
        import threading
        if not self.ok or self._thread is not None:
            return self

        def loop():
            while not self._stop.is_set():
                if self.paused:                      # the CPU baseline has the host cores to itself
                    time.sleep(0.01)
                    continue
                r = self._read()
                with self._lock:
                    self.samples.append(r)
                time.sleep(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)

    def window(self, t0, t1):
        if not self.ok:
            return {"available": False, "reason": self.why}
        note = None
        with self._lock:
            snap = list(self.samples)
        rows = [x for x in snap if t0 <= x[0] <= t1]
        if len(rows) < 2:   # a 12 ms region may fall between two updates: take the last 100 ms up to its end
            rows = [x for x in snap if t1 - 0.1 <= x[0] <= t1 + 0.01]
            note = "timed region shorter than two samples: window widened to the 100 ms before its end (warm-up launches of the same kernel)"
        if not rows:
            return {"available": False, "reason": "no sample in the window"}
        a = np.array([r[1:] for r in rows], dtype=np.float64)
        a = a[~np.isnan(a).any(axis=1)] if len(a) else a
        if not len(a):
            return {"available": False, "reason": "hwmon files unreadable"}
        out = {"available": True, "sclk_mhz": float(a[:, 0].mean()), "sclk_mhz_min": float(a[:, 0].min()),
               "sclk_mhz_max": float(a[:, 0].max()), "socket_w": float(a[:, 1].mean()), "socket_w_max": float(a[:, 1].max()),
               "junction_c": float(a[:, 2].mean()), "power_cap_w": self.cap_w, "samples": int(len(a)),
               "window_ms": (rows[-1][0] - rows[0][0]) * 1e3,
               "poll_period_ms": self.period * 1e3, "polled_during_timed_region": True,
               "source": "sysfs hwmon of %s (freq1 = sclk, power1 = socket power, junction temperature), polled every ~%.1f ms "
                         "by a host thread across the timed region" % (self.bus, self.period * 1e3)}
        if note:
            out["note"] = note
        return out


TELEMETRY = None   # set in main() (rank-local GPU)


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
    ap.add_argument("--batches", type=int, default=4, help="distinct resident input batches the timed loop rotates over")
    ap.add_argument("--streams", type=int, default=None, choices=[1, 2],
                    help="HIP streams the timed launches alternate over (default: 2 for verify / var_base, 1 otherwise; with 2 "
                         "the single-stream protocol is measured as well and printed next to it)")
    ap.add_argument("--window-bits", type=int, default=int(os.environ.get("BJJ_BENCH_WINDOW_BITS", "28")),
                    help="fixed-base window width passed to bjj_init (28 = 154.6 GB table; 0 = the library default, 23; -1 = auto); "
                         "default from BJJ_BENCH_WINDOW_BITS when set (the driver's command line is fixed: 8 ranks on one host "
                         "can be given a smaller table this way, tools/scale_session.sh)")
    ap.add_argument("--strong-total", type=int, default=1 << 24, help="total items of the cfg-5 strong-scaling line")
    ap.add_argument("--signer-constant-time", action="store_true",
                    help="bjj_set_signer_constant_time(ctx, 1): the signer workloads (sign) scan the small 4-bit table instead of "
                         "indexing the fixed-base table with secret digits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workload lines")
    ap.add_argument("--no-strong", action="store_true", help="skip the fixed-total-work (strong scaling / cfg 5) lines")
    ap.add_argument("--native-multi", action="store_true",
                    help="ONE process, all --gpus devices through bjj_multi_* (RCCL inside the library)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "peer"],
                    help="--native-multi: grouped ncclScatter / ncclGather, or hipMemcpyPeerAsync of the same blocks")
    ap.add_argument("--chunks", type=int, default=4,
                    help="--native-multi: pieces a peer's block travels in (bjj_multi_set_chunks); the serial schedule "
                         "(1 piece) is timed next to it")
    ap.add_argument("--devices", default=None,
                    help="--native-multi: comma-separated device list (default 0..gpus-1; a device may repeat with --transport peer)")
    ap.add_argument("--detail-out", default=os.environ.get("BJJ_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="file that receives EVERYTHING measured (the stdout line is the compact record of babyjubjub-rs_amd/benchline.py: "
                         "<= 4 KB, numbers only); '' = no file.  The same record goes to stderr as one 'bench_detail: {...}' line")
    return ap.parse_args()


DETAIL_OUT = os.path.join(ROOT, "bench_detail.json")   # set from --detail-out in main()


def publish(full):
    """rank 0's output: the full record to DETAIL_OUT and to stderr, then -- as the LAST thing written to stdout -- the one compact
    line a driver parses (benchline.compact: <= 4 KB by construction)"""
    from babyjubjub_rs_amd import benchline
    shown = None
    if DETAIL_OUT:
        try:
            with open(DETAIL_OUT, "w") as fh:
                json.dump(full, fh, indent=1)
                fh.write("\n")
            shown = os.path.relpath(DETAIL_OUT, ROOT) if os.path.abspath(DETAIL_OUT).startswith(ROOT + os.sep) else DETAIL_OUT
        except OSError as e:      # a read-only checkout must not cost the line
            sys.stderr.write("bench.py: cannot write %s: %s\n" % (DETAIL_OUT, e))
    sys.stderr.write("bench_detail: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    sys.stdout.write(benchline.dumps(benchline.compact(full, shown)) + "\n")
    sys.stdout.flush()


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own ranks (never exec from a process that touched the GPU)
# ---------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(args):
    have = torch.cuda.device_count()   # counting devices does not initialise the GPU
    if have < args.gpus and os.environ.get("BJJ_BENCH_SHARE_GPU", "0") != "1":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (BJJ_BENCH_SHARE_GPU=1 + BJJ_BENCH_BACKEND=gloo is the "
                         "single-GPU developer mode)" % (args.gpus, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------------------------
# device-resident synthetic inputs
# ---------------------------------------------------------------------------------------------------------------
class Batch:
    """one resident input batch + its output buffers"""
    pass


class Workload:
    """Device-resident synthetic inputs for one rank (`nb` distinct batches), and the launch closure."""

    def __init__(self, ctx, kind, n, offset, dev, stream, nb=1):
        from babyjubjub_rs_amd import workload as w
        self.kind, self.n, self.ctx, self.stream, self.dev = kind, n, ctx, stream, dev
        self.streams = [stream]          # launch k goes to streams[k % len(streams)] (timed_steps sets this)
        self.batches = []
        for b in range(nb):
            self.batches.append(self._make(w, offset + b * n * 1009))   # distinct SplitMix64 windows per batch
        self.last = 0

    def _up(self, a):
        return torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(self.dev)

    def _empty(self, nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device=self.dev)

    def _make(self, w, offset):
        c, n, kind, s = self.ctx, self.n, self.kind, 0
        B = Batch()
        B.offset = offset
        if kind in ("fixed_base", "fixed_base_compressed"):
            B.d_sc = self._up(w.scalars_254(n, offset))
            B.d_out = self._empty(n * (64 if kind == "fixed_base" else 32))
        elif kind == "var_base":
            # cfg 3 points: k*B8 + c*T8 (the whole group, cofactor components included), made by the parity-tested kernels
            B.d_sc = self._up(w.scalars_254(n, offset))
            d_k = self._up(w.random_u256(w.SEED_POINTS, n, offset))
            d_kb = self._empty(n * 64)
            c.mul_fixed_base_dev(d_k.data_ptr(), n, d_kb.data_ptr(), s)
            tors = c.mul_var_base([T8] * 8, list(range(8)))                       # (8, 64): j * T8
            cidx = (w.splitmix64(w.SEED_POINTS ^ 0x77, n, offset) & np.uint64(7)).astype(np.int64)
            d_t = self._up(tors)[None].reshape(8, 64)[torch.from_numpy(cidx).to(self.dev)].reshape(-1).contiguous()
            torch.cuda.synchronize()      # torch's gather ran on torch's stream; the library launches on its own
            B.d_pts = self._empty(n * 64)
            c.point_add_dev(d_kb.data_ptr(), d_t.data_ptr(), n, B.d_pts.data_ptr(), s)
            c.sync()
            B.d_out = self._empty(n * 64)
        elif kind == "poseidon5":
            B.d_in = self._up(w.random_u256(w.SEED_MSGS, 5 * n, 5 * offset, top_bits_cleared=3))
            B.d_out = self._empty(n * 32)
        elif kind == "sign":
            B.d_keys = self._up(w.random_u256(w.SEED_KEYS, n, offset))
            B.d_msgs = self._up(w.random_u256(w.SEED_MSGS, n, offset, top_bits_cleared=3))
            B.d_out, B.d_s, B.d_ok = self._empty(n * 64), self._empty(n * 32), self._empty(n)
        elif kind in ("point_add", "compress"):
            # random group points k*B8 (the criterion cases use one fixed point; a batch uses n different ones)
            d_k = self._up(w.random_u256(w.SEED_POINTS, n, offset))
            B.d_p = self._empty(n * 64)
            c.mul_fixed_base_dev(d_k.data_ptr(), n, B.d_p.data_ptr(), s)
            if kind == "point_add":
                d_k2 = self._up(w.random_u256(w.SEED_POINTS ^ 0x5151, n, offset))
                B.d_q = self._empty(n * 64)
                c.mul_fixed_base_dev(d_k2.data_ptr(), n, B.d_q.data_ptr(), s)
            c.sync()
            B.d_out = self._empty(n * (64 if kind == "point_add" else 32))
        elif kind == "decompress":
            d_k = self._up(w.random_u256(w.SEED_POINTS, n, offset))
            B.d_pts = self._empty(n * 64)
            c.mul_fixed_base_dev(d_k.data_ptr(), n, B.d_pts.data_ptr(), s)
            B.d_in = self._empty(n * 32)
            c.compress_points_dev(B.d_pts.data_ptr(), n, B.d_in.data_ptr(), s)
            c.sync()
            B.d_out, B.d_ok = self._empty(n * 64), self._empty(n)
        else:
            # verify / verify_compressed: cfg 4 signatures produced on the device by the (oracle-checked) signer kernels --
            # A = public(key), (R, S) = sign(key, msg), src/lib.rs:304-342 -- then 1 in 64 corrupted in place
            d_keys = self._up(w.random_u256(w.SEED_KEYS, n, offset))
            B.d_msg = self._up(w.random_u256(w.SEED_MSGS, n, offset, top_bits_cleared=3))
            B.d_pk, B.d_r, B.d_s = self._empty(n * 64), self._empty(n * 64), self._empty(n * 32)
            d_f = self._empty(n)
            c.public_keys_dev(d_keys.data_ptr(), n, B.d_pk.data_ptr(), s)
            c.sign_dev(d_keys.data_ptr(), B.d_msg.data_ptr(), n, B.d_r.data_ptr(), B.d_s.data_ptr(), d_f.data_ptr(), s)
            c.sync()
            assert bool(d_f.all())
            if kind == "verify":
                B.bad = w.corrupt(B.d_pk.view(n, 64), B.d_r.view(n, 64), B.d_s.view(n, 32), B.d_msg.view(n, 32), n, offset)
            else:  # wire format: 32-byte pk, 64-byte signature; corruption lands in s / msg / the compressed bytes
                d_pkc, d_rc = self._empty(n * 32), self._empty(n * 32)
                c.compress_points_dev(B.d_pk.data_ptr(), n, d_pkc.data_ptr(), s)
                c.compress_points_dev(B.d_r.data_ptr(), n, d_rc.data_ptr(), s)
                c.sync()
                z = torch.zeros(n, 32, dtype=torch.uint8, device=self.dev)
                A_t = torch.cat([d_pkc.view(n, 32), z], dim=1)          # corrupt() flips A[i, :32] ...
                R_t = torch.cat([z, d_rc.view(n, 32)], dim=1)           # ... and R[i, 32:]: the compressed bytes
                B.bad = w.corrupt(A_t, R_t, B.d_s.view(n, 32), B.d_msg.view(n, 32), n, offset)
                B.d_pk = A_t[:, :32].contiguous().reshape(-1)
                B.d_sig = torch.cat([R_t[:, 32:], B.d_s.view(n, 32)], dim=1).contiguous().reshape(-1)
            B.d_out = self._empty(n)
        torch.cuda.synchronize()          # everything torch did to the inputs (uploads, corruption) is complete
        return B

    def launch(self, k=0):
        c, n, s = self.ctx, self.n, self.streams[k % len(self.streams)].cuda_stream
        B = self.batches[k % len(self.batches)]
        self.last = k % len(self.batches)
        kind = self.kind
        if kind == "fixed_base":
            c.mul_fixed_base_dev(B.d_sc.data_ptr(), n, B.d_out.data_ptr(), s)
        elif kind == "fixed_base_compressed":
            c.mul_fixed_base_compressed_dev(B.d_sc.data_ptr(), n, B.d_out.data_ptr(), s)
        elif kind == "var_base":
            c.mul_var_base_dev(B.d_pts.data_ptr(), B.d_sc.data_ptr(), n, B.d_out.data_ptr(), s)
        elif kind == "poseidon5":
            c.poseidon5_dev(B.d_in.data_ptr(), n, B.d_out.data_ptr(), s)
        elif kind == "sign":
            c.sign_dev(B.d_keys.data_ptr(), B.d_msgs.data_ptr(), n, B.d_out.data_ptr(), B.d_s.data_ptr(), B.d_ok.data_ptr(), s)
        elif kind == "point_add":
            c.point_add_dev(B.d_p.data_ptr(), B.d_q.data_ptr(), n, B.d_out.data_ptr(), s)
        elif kind == "compress":
            c.compress_points_dev(B.d_p.data_ptr(), n, B.d_out.data_ptr(), s)
        elif kind == "decompress":
            c.decompress_points_dev(B.d_in.data_ptr(), n, B.d_out.data_ptr(), B.d_ok.data_ptr(), s)
        elif kind == "verify_compressed":
            c.eddsa_verify_compressed_dev(B.d_pk.data_ptr(), B.d_sig.data_ptr(), B.d_msg.data_ptr(), n, B.d_out.data_ptr(), s)
        else:
            c.eddsa_verify_dev(B.d_pk.data_ptr(), B.d_r.data_ptr(), B.d_s.data_ptr(), B.d_msg.data_ptr(), n, B.d_out.data_ptr(), s)

    # ---- host views of a few rows (the oracle runs on the host) ----
    def rows(self, t, width, idx, B=None):
        B = B or self.batches[self.last]
        v = getattr(B, t).view(self.n, width)
        if isinstance(idx, slice):
            return v[idx].cpu().numpy()
        return v[torch.from_numpy(np.asarray(idx)).to(self.dev)].cpu().numpy()

    def oracle_run(self, orc, idx, B=None):
        """the oracle's outputs for rows `idx` of batch B (also the cpu_baseline's unit of work)"""
        k = self.kind
        r = lambda t, wd: self.rows(t, wd, idx, B)  # noqa: E731
        if k == "fixed_base":
            return orc.mul_fixed_base(r("d_sc", 32))
        if k == "fixed_base_compressed":
            return orc.compress(orc.mul_fixed_base(r("d_sc", 32)))
        if k == "var_base":
            return orc.mul_var_base(r("d_pts", 64), r("d_sc", 32))
        if k == "poseidon5":
            return orc.poseidon5(r("d_in", 160))
        if k == "sign":
            return orc.sign(r("d_keys", 32), r("d_msgs", 32))
        if k == "decompress":
            return orc.decompress(r("d_in", 32))
        if k == "point_add":
            return orc.point_add(r("d_p", 64), r("d_q", 64))
        if k == "compress":
            return orc.compress(r("d_p", 64))
        if k == "verify_compressed":
            return orc.verify_compressed(r("d_pk", 32), r("d_sig", 64), r("d_msg", 32))
        return orc.verify(r("d_pk", 64), r("d_r", 64), r("d_s", 32), r("d_msg", 32))

    def check_sample(self, orc, count=512):
        """byte-compare a strided sample of the LAST launched batch's output with the oracle (verify: plus the full
        verdict vector against the known corruption mask)"""
        n, k = self.n, self.kind
        B = self.batches[self.last]
        idx = np.unique(np.linspace(0, n - 1, min(count, n)).astype(np.int64))
        want = self.oracle_run(orc, idx, B)
        if k in ("fixed_base", "var_base", "point_add"):
            return bool((self.rows("d_out", 64, idx, B) == want).all())
        if k in ("poseidon5", "compress", "fixed_base_compressed"):
            return bool((self.rows("d_out", 32, idx, B) == want).all())
        if k == "sign":
            return bool((self.rows("d_out", 64, idx, B) == want[0]).all()) and bool((self.rows("d_s", 32, idx, B) == want[1]).all()) \
                and bool(B.d_ok.all())
        if k == "decompress":
            return bool(torch.equal(B.d_out, B.d_pts)) and bool(B.d_ok.all()) and bool((self.rows("d_out", 64, idx, B) == want[0]).all())
        got = B.d_out.cpu().numpy()
        if k == "verify_compressed":
            return bool((got[~B.bad] == 1).all()) and bool((got[B.bad] != 1).all()) and bool((got[idx] == want).all())
        return bool((got == (~B.bad).astype(np.uint8)).all()) and bool((got[idx] == want).all())


def timed_steps(wl, steps, warmup, world, warm_s=0.0, streams=None):
    """W untimed + K timed launches; returns (wall seconds for K steps, mean device ms per launch from HIP events on the
    launch streams).  warm_s: extra untimed launches until that many seconds have passed (clock warm-up, see
    --warmup-seconds).
    streams = [s0] (default: the workload's stream): launches back to back on one stream; the device time of launch k is
    the HIP-event interval around it.
    streams = [s0, s1]: launch k goes to stream k % 2 -- the context gives each stream its own scratch set, so the head of
    launch k+1 fills the partly empty last wave-round of launch k.  Launch intervals then overlap, so
    the per-launch device time is (first start .. last end over all streams) / K, from HIP events as well."""
    sts = streams or [wl.stream]
    wl.streams = sts
    t_w = time.perf_counter()
    for k in range(warmup):
        wl.launch(k)
    for st in sts:
        st.synchronize()
    while time.perf_counter() - t_w < warm_s:
        for k in range(8):
            wl.launch(k)
        for st in sts:
            st.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if len(sts) == 1:
        st = sts[0]
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        evs[0].record(st)
        for k in range(steps):
            wl.launch(k)
            evs[k + 1].record(st)
        st.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dt = t1 - t0
        per = [evs[k].elapsed_time(evs[k + 1]) for k in range(steps)]
        wl.step_ms = per
        wl.clock = TELEMETRY.window(t0, t1) if TELEMETRY else None
        return dt, float(np.mean(per))
    ev0 = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in sts]
    t0 = time.perf_counter()
    ev0.record(sts[0])
    for st in sts[1:]:
        st.wait_event(ev0)              # every stream starts behind the same point
    for k in range(steps):
        wl.launch(k)
    for st, e in zip(sts, ends):
        e.record(st)
    for st in sts:
        st.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dt = t1 - t0
    dev_ms = max(ev0.elapsed_time(e) for e in ends)
    wl.step_ms = None
    wl.clock = TELEMETRY.window(t0, t1) if TELEMETRY else None
    return dt, dev_ms / steps


def overlapped_launch_detail(wl, steps, sts):
    """A SEPARATE pass of the two-stream protocol with a HIP-event pair around every launch on its own stream (the timed
    region that produces `value` carries no per-launch events): duration of each overlapped launch -- what rocprofv3's kernel
    trace reports per dispatch, about twice the span per launch because two launches are co-resident -- and the span of the
    pass.  Returns {"kernel_ms_avg", "kernel_ms_median", "span_ms_per_launch", "launches"}."""
    wl.streams = sts
    for k in range(4):
        wl.launch(k)
    for st in sts:
        st.synchronize()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    ev0.record(sts[0])
    for st in sts[1:]:
        st.wait_event(ev0)
    for k in range(steps):
        st = sts[k % len(sts)]
        pairs[k][0].record(st)
        wl.launch(k)
        pairs[k][1].record(st)
    for st in sts:
        st.synchronize()
    per = np.array([a.elapsed_time(b) for a, b in pairs], dtype=np.float64)
    span = max(ev0.elapsed_time(b) for _, b in pairs)
    return {"kernel_ms_avg": float(per.mean()), "kernel_ms_median": float(np.median(per)), "kernel_ms_min": float(per.min()),
            "kernel_ms_max": float(per.max()), "span_ms_per_launch": span / steps, "launches": steps,
            "note": "separate pass of the same two-stream protocol with an event pair around every launch (HIP events on the "
                    "launch's own stream)"}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline, roofline, VALU model
# ---------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    info = {"logical_cpus": os.cpu_count() or 1}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # cgroup v2 CPU quota: "max 100000" = unlimited, "800000 100000" = 8 CPUs
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        info["cgroup_cpu_quota"] = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:
        info["cgroup_cpu_quota"] = "unknown"
    return info


def get_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    return Oracle()


def cpu_baseline(kind, wl, orc, budget_cpu_s=12.0):
    """oracle (C restatement of the reference algorithm, "port") on the host cores, bounded sample.
    The thread count is chosen by a short probe (containers often expose more logical CPUs than
    their CPU quota lets run at once); `cores` reports the threads actually used."""
    hc = host_cpu_info()
    logical = hc.get("affinity_cpus", hc["logical_cpus"])
    B = wl.batches[0]
    if TELEMETRY:
        TELEMETRY.paused = True       # no 2 kHz sysfs polling next to the timed host threads (resumed below)

    def run(m):
        t0 = time.perf_counter()
        wl.oracle_run(orc, slice(0, m), B)
        return time.perf_counter() - t0

    orc.threads = 1
    m1 = min({"fixed_base": 1024, "fixed_base_compressed": 1024, "var_base": 1024, "verify": 384, "poseidon5": 2048, "verify_compressed": 384,
              "decompress": 2048, "sign": 512, "point_add": 16384, "compress": 1 << 18}[kind], wl.n)
    run(min(m1, 64))            # page in
    rate1 = m1 / run(m1)
    best_t, best_rate = 1, rate1
    for t in sorted({2, 4, 8, 16, 32, 64, 128, logical}):
        if t > logical:
            continue
        orc.threads = t
        m = int(min(wl.n, max(t * 8, rate1 * 0.2 * t)))
        r = m / run(m)
        if r > best_rate:
            best_t, best_rate = t, r
    orc.threads = best_t
    sample = int(min(wl.n, max(best_t * 8, best_rate * budget_cpu_s)))
    dt = run(sample)
    quota = hc.get("cgroup_cpu_quota")
    if TELEMETRY:
        TELEMETRY.paused = False
    return {"value": sample / dt, "unit": UNITS[kind], "cores": best_t, "kind": "port",
            "sample": "first %d items of the same %s batch, oracle/bjj_ref.c (reference algorithm: bit-serial "
                      "double-and-add with unified adds, binary-Euclid inversions, plain Poseidon), %d pthreads = best of a "
                      "thread-count probe; host: %d logical CPUs, %s usable by this process, cgroup CPU quota %s"
                      % (sample, kind, best_t, hc["logical_cpus"], hc.get("affinity_cpus", "?"),
                         "none" if quota is None else quota),
            "sample_short": "first %d items of the same batch; oracle/bjj_ref.c (C restatement of the reference algorithm), %d pthreads, "
                            "cgroup CPU quota %s" % (sample, best_t, "none" if quota is None else quota),
            "single_thread_value": rate1, "host": hc}


def load_profile_json(name):
    p = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(p))
    except Exception:
        return {}


def source_hash_now():
    """fingerprint of babyjubjub-rs_amd/csrc as it is in this tree (srchash.py); None if it cannot be computed"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "babyjubjub-rs_amd"))
        import srchash
        return srchash.tree_hash()
    except Exception:
        return None


def profile_is_current(stored_hash):
    """the committed counters are quoted only for the build they were taken from (whole-tree fingerprint: entries of rounds 3-5)"""
    now = source_hash_now()
    return stored_hash is not None and now is not None and stored_hash == now


_KERNEL_HASHES = None


def kernel_hash_now(kernel):
    """fingerprint of the MACHINE CODE of one kernel of the library in this tree (srchash.kernel_hashes); None if unknown"""
    global _KERNEL_HASHES
    if _KERNEL_HASHES is None:
        try:
            sys.path.insert(0, os.path.join(ROOT, "babyjubjub-rs_amd"))
            import srchash
            _KERNEL_HASHES = srchash.kernel_hashes()
        except Exception:
            _KERNEL_HASHES = {}
    return _KERNEL_HASHES.get(kernel)


def profile_entry_is_current(entry, kernel=None):
    """Do these committed counters describe the kernel that ships?  An entry that carries `code_hash` (round 6: the fingerprint of
    the profiled kernel's machine code, babyjubjub-rs_amd/srchash.py) is current exactly when the library in this tree holds the same
    code for that kernel -- an edit of another translation unit, or a new kernel next to it, does not stale it.  Older entries go by
    the whole-tree fingerprint."""
    if not entry:
        return False
    k = kernel or entry.get("kernel")
    if entry.get("code_hash") and k:
        now = kernel_hash_now(k)
        return now is not None and now == entry["code_hash"]
    return profile_is_current(entry.get("source_hash") or entry.get("_source_hash"))


def mix_is_current(mixj, kernel):
    """profiles/isa_mix.json: the static instruction mix of `kernel` was taken from the code that ships"""
    m = mixj.get(kernel) or {}
    if m.get("code_hash"):
        return kernel_hash_now(kernel) == m["code_hash"]
    return profile_is_current(mixj.get("_source_hash"))


def fingerprint_text(entry):
    return ("kernel code fingerprint %s" % entry["code_hash"]) if entry.get("code_hash") else ("source fingerprint %s" % entry.get("source_hash"))


def roofline_blocks(kind, kernel_ms, n, info, extra):
    """(roofline, roofline_overlapped or None).  `roofline` describes ONE launch on one stream -- the kernel named in it is the
    kernel that protocol launched; when the timed launches alternated over two streams, `roofline_overlapped` describes the
    protocol (and the kernel form) that produced `value`."""
    one = (extra.get("single_stream") or {}).get("kernel") or extra.get("kernel")
    r = roofline_block(kind, kernel_ms, n, info, one)
    r["protocol"] = "one launch at a time on one stream (per-launch HIP events)"
    ro = None
    if extra.get("streams") == 2:
        ro = roofline_overlapped_block(kind, extra["device_ms_per_launch"], extra.get("overlap_detail"), n, info, extra.get("kernel"))
    return r, ro


def roofline_block(kind, kernel_ms, n, info, kernel=None):
    algo = ALGO_BYTES[kind] * n
    ach = algo / (kernel_ms * 1e-3) / 1e9
    tr = load_profile_json("hbm_traffic.json").get(kind, {})
    same_cfg = n == (1 << 20) and tr.get("window_bits") in (None, info.window_bits)
    stale = bool(tr) and not profile_entry_is_current(tr)
    quote = same_cfg and bool(tr) and not stale
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
           "traffic": tr.get("bytes_per_launch") if quote else None,
           "traffic_source": ("static: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command at %s, committed; "
                              "not re-measured in this run)" % (tr.get("source"), fingerprint_text(tr))) if quote else None,
           "kernel": kernel or KERNEL[kind], "kernel_ms_avg": kernel_ms, "algorithmic_bytes_per_launch": algo,
           "note": "integer-ALU bound path (see DESIGN.md): HBM fraction is reported as measured"}
    if kernel and tr.get("kernel") and tr["kernel"] != kernel:   # the committed counters describe another kernel form
        out["traffic"], out["traffic_source"] = None, None
    if stale:
        out["stale_profile"] = True
        out["stale_profile_note"] = ("profiles/hbm_traffic.json was taken at %s, this tree has %s: counters not quoted (re-run "
                                     "tools/profile_r.sh + tools/summarize_profile.py)"
                                     % (fingerprint_text(tr), kernel_hash_now(tr.get("kernel")) if tr.get("code_hash") else source_hash_now()))
    return out


def kernel_that_ran(kind, info, overlapped):
    """the kernel symbol the context's LAST launch of this workload used (bjj_get_info: last_* fields)"""
    dec = lambda b, alt: b.decode() if b else alt   # noqa: E731  (an older A/B build fills fewer fields of bjj_info)
    if kind == "fixed_base":
        return dec(info.kernel_fixed_base_overlap, "bjj_k_mul_fixed_base_2x256") if info.last_fixed_base_shape == 1 else dec(info.kernel_fixed_base, KERNEL[kind])
    if kind == "fixed_base_compressed":
        return "bjj_k_mul_fixed_base_2x256_c32" if info.last_fixed_base_shape == 1 else KERNEL[kind]
    if kind == "var_base":
        return dec(info.kernel_var_base_overlap, "bjj_k_mul_var_base") if info.last_var_base_form == 0 else dec(info.kernel_var_base, KERNEL[kind])
    if kind in ("verify", "verify_compressed"):
        return "bjj_k_eddsa_verify" if info.last_verify_dispatch == 0 else "bjj_k_eddsa_verify_groups"
    return KERNEL[kind]


def roofline_overlapped_block(kind, span_ms, detail, n, info, kernel):
    """The protocol that produces `value`: K launches alternating over two streams.  `achieved` = algorithmic bytes per launch /
    (span of the K launches / K) -- two launches are co-resident, so ONE launch takes about twice that (kernel_ms_avg, what a
    kernel trace shows per dispatch).  traffic: PMC passes of the same two-stream command (profiles/*_two_stream_summary.md)."""
    algo = ALGO_BYTES[kind] * n
    ach = algo / (span_ms * 1e-3) / 1e9
    tr = load_profile_json("hbm_traffic.json").get(kind + "_two_stream", {})
    stale = bool(tr) and not profile_entry_is_current(tr)
    quote = bool(tr) and not stale and n == (1 << 20) and tr.get("window_bits") in (None, info.window_bits) and tr.get("kernel") == kernel
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
           "kernel": kernel, "streams": 2, "span_ms_per_launch": span_ms, "algorithmic_bytes_per_launch": algo,
           "traffic": tr.get("bytes_per_launch") if quote else None,
           "traffic_source": tr.get("source") if quote else None,
           "trace": ({"span_us_per_launch": tr.get("trace_span_us_per_launch"), "kernel_us_avg": tr.get("trace_kernel_us_avg"),
                      "unprofiled_ms_per_step": tr.get("unprofiled_ms_per_step"), "source": tr.get("source")} if quote else None),
           "note": "two launches co-resident: achieved = algorithmic bytes / (span of the K timed launches / K)"}
    if detail:
        out["kernel_ms_avg"] = detail["kernel_ms_avg"]
        out["per_launch_pass"] = detail
    # VALU utilisation of the overlap-form kernel from its own PMC pass (counters of one dispatch; the profiler serialises
    # dispatches) and its own static instruction mix -- the same two figures `valu` gives for the one-stream kernel
    mixj = load_profile_json("isa_mix.json")
    mix = mixj.get(kernel or "", {})
    if quote and mix and tr.get("valu_insts_per_launch") and tr.get("sq_busy_cycles") and mix_is_current(mixj, kernel):
        cyc = mix["avg_issue_cycles_per_valu_inst"]
        ipc = tr["valu_insts_per_launch"] / 1024.0 / (tr["sq_busy_cycles"] / 32.0)
        peak = 1024 * 2.4 / cyc
        va = tr["valu_insts_per_launch"] / (span_ms * 1e-3) / 1e9
        out["valu"] = {"insts_per_launch": tr["valu_insts_per_launch"], "achieved": va, "peak": peak, "frac": va / peak,
                       "unit": "G wave-instructions/s", "avg_issue_cycles_per_inst": cyc,
                       "counter_derived": {"valu_insts_per_simd_per_busy_cycle": ipc, "frac": ipc * cyc,
                                           "note": "counters of ONE dispatch of the overlap form running alone (PMC pass)"},
                       "note": "achieved = instructions per launch / (span of the K timed launches / K): two co-resident launches share the SIMDs"}
    if stale:
        out["stale_profile"] = True
    return out


def valu_block(kind, kernel_ms, n, info, clock=None):
    """The unit that actually bounds this path is VALU issue.  Peak = 1024 SIMDs x 2.4 GHz / (issue cycles per
    wave-instruction of THIS kernel's instruction mix): the static ISA histogram of the kernel (tools/isa_histogram.py ->
    profiles/isa_mix.json) weighted with the measured per-class issue cycles (profiles/r01_ubench_valu_rates.txt: 4.54 for
    v_mad_u64_u32 chains, ~4.2 for the other quarter-rate classes, ~2.4 for plain 32-bit VOP1/VOP2)."""
    tr = load_profile_json("hbm_traffic.json").get(kind, {})
    mix = load_profile_json("isa_mix.json").get(KERNEL[kind], {})
    vi = tr.get("valu_insts_per_launch")
    if not vi or not mix or n != (1 << 20) or tr.get("window_bits") not in (None, info.window_bits):
        return None
    if not profile_entry_is_current(tr) or not mix_is_current(load_profile_json("isa_mix.json"), KERNEL[kind]):
        return {"stale_profile": True, "frac": None,
                "note": "the committed instruction counters / ISA mix describe another build (%s / mix %s; this tree: kernel %s, source %s)"
                        % (fingerprint_text(tr), mix.get("code_hash") or load_profile_json("isa_mix.json").get("_source_hash"),
                           kernel_hash_now(KERNEL[kind]), source_hash_now())}
    cyc = mix["avg_issue_cycles_per_valu_inst"]
    peak = 1024 * 2.4 / cyc
    va = vi / (kernel_ms * 1e-3) / 1e9
    out = {"insts_per_launch": vi, "achieved": va, "peak": peak, "frac": va / peak, "unit": "G wave-instructions/s",
           "avg_issue_cycles_per_inst": cyc, "quarter_rate_share": mix.get("quarter_rate_share"),
           "note": "insts: static SQ_INSTS_VALU of the committed rocprofv3 PMC pass (%s) / live kernel time; peak: issue-cycle "
                   "model from the kernel's static ISA mix (profiles/isa_mix.json) x measured per-class rates at the nominal "
                   "2.4 GHz; frac_at_measured_clock prices the same ceiling at the sclk sampled across the timed region "
                   "(the package runs into its power cap)" % tr.get("source")}
    if tr.get("sq_busy_cycles"):
        # the same figure from hardware counters alone: instructions per SIMD per busy cycle of the profiled
        # launches x this kernel's average issue cost = the share of its busy cycles in which a SIMD issued VALU work.  Both
        # counters come from the one PMC pass, so the ratio does not depend on the clock of either run.
        ipc = vi / 1024.0 / (tr["sq_busy_cycles"] / 32.0)
        out["counter_derived"] = {"valu_insts_per_simd_per_busy_cycle": ipc, "frac": ipc * cyc,
                                  "formula": "SQ_INSTS_VALU / 1024 SIMDs / (SQ_BUSY_CYCLES / 32) x avg_issue_cycles_per_inst",
                                  "source": tr.get("source")}
    if clock and clock.get("available") and clock.get("sclk_mhz"):
        out["clock_mhz"] = clock["sclk_mhz"]
        out["peak_at_measured_clock"] = peak * clock["sclk_mhz"] / 2400.0
        out["frac_at_measured_clock"] = va / out["peak_at_measured_clock"]
    return out


def one_stream_clock(extra):
    """telemetry of the ONE-stream timed region -- the protocol roofline.kernel_ms_avg and valu are computed from"""
    return (extra.get("single_stream") or {}).get("clock") or extra.get("clock")


def step_stats(step_ms, n, world):
    """per-launch HIP-event times of the timed region (single-stream protocol): the median is robust against one DVFS or
    scheduling hiccup in a window of a few milliseconds; `value_from_median` is the rate it implies for this rank x world"""
    a = np.asarray(step_ms, dtype=np.float64)
    med = float(np.median(a))
    return {"median_ms": med, "mean_ms": float(a.mean()), "min_ms": float(a.min()), "max_ms": float(a.max()),
            "p90_ms": float(np.percentile(a, 90)), "value_from_median": world * n / (med * 1e-3)}


def measure(ctx, kind, n, offset, dev, stream, steps, warmup, warm_s, world, nb, orc, with_cpu, rank, stream2=None):
    """one full measurement block of a workload on this rank: (wl, dt over the K timed steps, device ms per launch,
    extras, parity_ok, cpu_baseline).  With stream2 the timed launches alternate over two streams (overlapping launches,
    see timed_steps) and the classic one-stream protocol is measured first and reported in extras["single_stream"]."""
    wl = Workload(ctx, kind, n, offset, dev, stream, nb=max(nb, 2 if stream2 is not None else 1))
    dt, kernel_ms = timed_steps(wl, steps, warmup, world, warm_s)
    extra = {"streams": 1, "clock": wl.clock, "kernel": kernel_that_ran(kind, ctx.info(), False)}
    if wl.step_ms:
        extra["per_launch_event_ms"] = step_stats(wl.step_ms, n, world)
    if stream2 is not None:
        dt1, k1, clock1 = dt, kernel_ms, wl.clock
        kernel_one = kernel_that_ran(kind, ctx.info(), False)
        dt, km2 = timed_steps(wl, steps, 2, world, 0.2, streams=[stream, stream2])
        kernel_two = kernel_that_ran(kind, ctx.info(), True)
        clock2 = wl.clock
        detail = overlapped_launch_detail(wl, steps, [stream, stream2]) if rank == 0 else None
        wl.clock = clock2
        # kernel_ms stays the ONE-launch time (per-launch HIP events on one stream): roofline / valu describe a launch, and the
        # rocprofv3 summaries profile exactly that; the overlapped protocol's device time per launch is reported next to it
        extra = {"streams": 2, "device_ms_per_launch": km2, "clock": wl.clock,
                 "kernel": kernel_two, "overlap_detail": detail,
                 "single_stream": {"wall_s": dt1, "kernel_ms_avg": k1, "value_this_rank": n * steps / dt1, "clock": clock1,
                                   "kernel": kernel_one,
                                   "per_launch_event_ms": extra.get("per_launch_event_ms")},
                 "streams_note": "timed launches alternate over two HIP streams; the context keeps one scratch set per stream, "
                                 "so consecutive launches overlap.  device_ms_per_launch = (first start .. last end over both "
                                 "streams, HIP events) / steps; single_stream = the same K launches back to back on one stream, "
                                 "which is also what roofline.kernel_ms_avg and valu are computed from"}
    if len(wl.batches) > 1 and rank == 0:  # Infinity-Cache control: the same protocol on ONE repeated batch
        one = Workload.__new__(Workload)
        one.__dict__.update(wl.__dict__)
        one.batches = wl.batches[:1]
        _, k1b = timed_steps(one, max(10, steps // 2), 2, 1, 0.3)
        extra["single_batch_kernel_ms"] = k1b
        extra["rotating_batches"] = len(wl.batches)
    ok = wl.check_sample(orc)
    cb = cpu_baseline(kind, wl, orc) if with_cpu else None
    return wl, dt, kernel_ms, extra, ok, cb


def host_api_block(ctx, n, orc):
    """bjj_mul_fixed_base / bjj_eddsa_verify on HOST pointers (the signatures a Rust caller holding BigInts binds to,
    include/bjj_hip.h): the call copies in, computes and copies out.  Measured on both kinds of caller memory:
      pinned    arrays from bjj_host_alloc -- what the crate's *_batch marshalling writes its records into (rust/src/gpu.rs):
                copied from / to directly, every byte crosses PCIe once and nothing else moves it
      pageable  ordinary numpy memory, already touched: staged through the context's pinned ring by its copy workers
    best of a few calls each; byte-compared with the oracle on a sample.  NEVER `value`."""
    from babyjubjub_rs_amd import workload as w
    sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
    idx = np.unique(np.linspace(0, n - 1, 64).astype(np.int64))
    # verify: the workload of `--workload verify` (BASELINE configs[3]: valid signatures from the device signer, 1 in 64 corrupted, half
    # of those with pk or R off the curve = the ~3x longer exact items) -- not random points, which hold no exact item at all
    dev = torch.device("cuda", torch.cuda.current_device())
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)  # noqa: E731
    d_keys, d_m = up(w.random_u256(w.SEED_KEYS, n, 0)), up(w.random_u256(w.SEED_MSGS, n, 0, top_bits_cleared=3))
    d_pk, d_r, d_s, d_f = (torch.empty(k, dtype=torch.uint8, device=dev) for k in (n * 64, n * 64, n * 32, n))
    ctx.public_keys_dev(d_keys.data_ptr(), n, d_pk.data_ptr(), 0)
    ctx.sign_dev(d_keys.data_ptr(), d_m.data_ptr(), n, d_r.data_ptr(), d_s.data_ptr(), d_f.data_ptr(), 0)
    ctx.sync()
    # the same signatures in wire format (32-byte pk, 64-byte signature), corrupted the way `--workload verify_compressed` corrupts them -- made BEFORE the
    # affine copies are corrupted (a corrupted point does not compress to anything meaningful)
    d_pkc, d_rc = torch.empty(n * 32, dtype=torch.uint8, device=dev), torch.empty(n * 32, dtype=torch.uint8, device=dev)
    ctx.compress_points_dev(d_pk.data_ptr(), n, d_pkc.data_ptr(), 0)
    ctx.compress_points_dev(d_r.data_ptr(), n, d_rc.data_ptr(), 0)
    ctx.sync()
    z_ = torch.zeros(n, 32, dtype=torch.uint8, device=dev)
    A_t, R_t = torch.cat([d_pkc.view(n, 32), z_], dim=1), torch.cat([z_, d_rc.view(n, 32)], dim=1)
    d_sw, d_mw = d_s.clone(), d_m.clone()
    bad_w = w.corrupt(A_t, R_t, d_sw.view(n, 32), d_mw.view(n, 32), n, 0)
    d_pkw = A_t[:, :32].contiguous().reshape(-1)
    d_sigw = torch.cat([R_t[:, 32:], d_sw.view(n, 32)], dim=1).contiguous().reshape(-1)
    w_host = [t.cpu().numpy() for t in (d_pkw, d_sigw, d_mw)]
    bad = w.corrupt(d_pk.view(n, 64), d_r.view(n, 64), d_s.view(n, 32), d_m.view(n, 32), n, 0)
    v_host = [t.cpu().numpy() for t in (d_pk, d_r, d_s, d_m)]
    d_ok = torch.empty(n, dtype=torch.uint8, device=dev)

    def best(f, reps, warm_s=0.5):
        # calls back to back for warm_s first: the section before this one leaves the GPU idle for a few hundred milliseconds
        # (a context torn down, inputs generated on the host), and the first calls after that run during the DVFS ramp --
        # 2.7 ms instead of 1.6 for 2^20 fixed-base multiplications (tools/host_api_probe.py, profiles/r05_host_pipeline.txt)
        t_w = time.perf_counter()
        f()
        while time.perf_counter() - t_w < warm_s:
            f()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t)
        return min(ts), float(np.median(ts))

    def run(kind_mem):
        if kind_mem == "pinned":
            alloc = ctx.host_empty
        else:
            alloc = lambda nb: np.zeros(nb, np.uint8)  # noqa: E731
        h_sc, out, ok = alloc(n * 32), alloc(n * 64), alloc(n)
        h_sc[:] = sc
        out[:] = 0
        ok[:] = 0
        t_fb, t_fb_med = best(lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, n, out.ctypes.data), "bjj_mul_fixed_base"), 7)
        i_fb = ctx.info()
        good = bool((np.asarray(out).reshape(n, 64)[idx] == orc.mul_fixed_base(sc.reshape(n, 32)[idx])).all())
        # ... and the same multiplications leaving as Point::compress records (src/lib.rs:166-178): half the bytes back across PCIe
        out32 = alloc(n * 32)
        out32[:] = 0
        t_fc, t_fc_med = best(lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base_compressed(ctx.handle, h_sc.ctypes.data, n, out32.ctypes.data), "bjj_mul_fixed_base_compressed"), 7)
        i_fc = ctx.info()
        good = good and bool((np.asarray(out32).reshape(n, 32)[idx] == orc.compress(orc.mul_fixed_base(sc.reshape(n, 32)[idx]))).all())
        # variable base (Point::mul_scalar on a caller's own points, src/lib.rs:149-164): the points just computed, the same scalars
        h_pts, out_vb = alloc(n * 64), alloc(n * 64)
        h_pts[:] = out
        out_vb[:] = 0
        t_vb, t_vb_med = best(lambda: ctx._ck(ctx.lib.bjj_mul_var_base(ctx.handle, h_pts.ctypes.data, h_sc.ctypes.data, n, out_vb.ctypes.data), "bjj_mul_var_base"), 3)
        i_vb = ctx.info()
        good = good and bool((np.asarray(out_vb).reshape(n, 64)[idx] == orc.mul_var_base(np.asarray(h_pts).reshape(n, 64)[idx], sc.reshape(n, 32)[idx])).all())
        hv = [alloc(a.size) for a in v_host]
        for b, a in zip(hv, v_host):
            b[:] = a
        t_v, t_v_med = best(lambda: ctx._ck(ctx.lib.bjj_eddsa_verify(ctx.handle, hv[0].ctypes.data, hv[1].ctypes.data, hv[2].ctypes.data, hv[3].ctypes.data, n,
                                                                      ok.ctypes.data), "bjj_eddsa_verify"), 3)
        i_v = ctx.info()
        vi = np.unique(np.concatenate([idx, np.nonzero(bad)[0][:64]]))                      # incl. corrupted items
        good = good and bool((np.asarray(ok) == (~bad).astype(np.uint8)).all())             # every verdict against the corruption mask
        good = good and bool((np.asarray(ok)[vi] == orc.verify(v_host[0].reshape(n, 64)[vi], v_host[1].reshape(n, 64)[vi], v_host[2].reshape(n, 32)[vi],
                                                               v_host[3].reshape(n, 32)[vi])).all())
        hw = [alloc(a.size) for a in w_host]
        for b, a in zip(hw, w_host):
            b[:] = a
        okw = alloc(n)
        okw[:] = 0
        t_vc, t_vc_med = best(lambda: ctx._ck(ctx.lib.bjj_eddsa_verify_compressed(ctx.handle, hw[0].ctypes.data, hw[1].ctypes.data, hw[2].ctypes.data, n,
                                                                               okw.ctypes.data), "bjj_eddsa_verify_compressed"), 3)
        i_vc = ctx.info()
        gw = np.asarray(okw)
        good = good and bool((gw[~bad_w] == 1).all()) and bool((gw[bad_w] != 1).all())
        wi = np.unique(np.concatenate([idx, np.nonzero(bad_w)[0][:64]]))
        good = good and bool((gw[wi] == orc.verify_compressed(w_host[0].reshape(n, 32)[wi], w_host[1].reshape(n, 64)[wi], w_host[2].reshape(n, 32)[wi])).all())
        res = {"fixed_base": {"value": n / t_fb, "unit": UNITS["fixed_base"], "ms_per_call": t_fb * 1e3, "ms_per_call_median": t_fb_med * 1e3,
                              "items": n, "bytes_moved": n * 96, "entry_point": "bjj_mul_fixed_base",
                              "arrays_direct": i_fb.last_host_direct_arrays, "arrays_staged": i_fb.last_host_staged_arrays,
                              "chunks": i_fb.last_host_chunks},
               "fixed_base_compressed": {"value": n / t_fc, "unit": UNITS["fixed_base"], "ms_per_call": t_fc * 1e3, "ms_per_call_median": t_fc_med * 1e3,
                                         "items": n, "bytes_moved": n * 64, "entry_point": "bjj_mul_fixed_base_compressed",
                                         "arrays_direct": i_fc.last_host_direct_arrays, "arrays_staged": i_fc.last_host_staged_arrays,
                                         "chunks": i_fc.last_host_chunks},
               "var_base": {"value": n / t_vb, "unit": UNITS["var_base"], "ms_per_call": t_vb * 1e3, "ms_per_call_median": t_vb_med * 1e3,
                            "items": n, "bytes_moved": n * 160, "entry_point": "bjj_mul_var_base",
                            "arrays_direct": i_vb.last_host_direct_arrays, "arrays_staged": i_vb.last_host_staged_arrays,
                            "chunks": i_vb.last_host_chunks, "device_one_launch_ms": t_dev_vb * 1e3, "vs_device_one_launch": t_vb / t_dev_vb},
               "verify": {"value": n / t_v, "unit": UNITS["verify"], "ms_per_call": t_v * 1e3, "ms_per_call_median": t_v_med * 1e3,
                          "items": n, "bytes_moved": n * 193, "entry_point": "bjj_eddsa_verify",
                          "arrays_direct": i_v.last_host_direct_arrays, "arrays_staged": i_v.last_host_staged_arrays,
                          "chunks": i_v.last_host_chunks, "device_one_launch_ms": t_dev * 1e3, "vs_device_one_launch": t_v / t_dev},
               "verify_compressed": {"value": n / t_vc, "unit": UNITS["verify"], "ms_per_call": t_vc * 1e3, "ms_per_call_median": t_vc_med * 1e3,
                                     "items": n, "bytes_moved": n * 129, "entry_point": "bjj_eddsa_verify_compressed",
                                     "arrays_direct": i_vc.last_host_direct_arrays, "arrays_staged": i_vc.last_host_staged_arrays,
                                     "chunks": i_vc.last_host_chunks, "device_one_launch_ms": t_dev_vc * 1e3, "vs_device_one_launch": t_vc / t_dev_vc},
               "parity_sample_ok": good}
        if kind_mem == "pinned":
            for a in [h_sc, out, out32, ok, okw, h_pts, out_vb] + hv + hw:
                ctx.host_free(a)
        return res

    # What ONE call of ONE item costs (the reference's single-item API -- B8.mul_scalar / p.mul_scalar / POSEIDON.hash / verify, src/lib.rs:149, :400,
    # :395 -- is served by n = 1 calls of the batch entry points): pinned host pointers, median of 40 calls after 0.2 s of them, each result
    # against the oracle.  Short calls run the several-lanes-per-item kernels (csrc/k_small.hip).
    def single_calls():
        import ctypes as C
        one = C.c_size_t(1)
        h_p, h_s, h_o = ctx.host_empty(64), ctx.host_empty(32), ctx.host_empty(64)
        h_5, h_h = ctx.host_empty(160), ctx.host_empty(32)
        hv = [ctx.host_empty(k) for k in (64, 64, 32, 32)]
        h_ok = ctx.host_empty(1)
        h_s[:] = sc[:32]
        pt = orc.mul_fixed_base(sc.reshape(n, 32)[1:2])
        h_p[:] = pt.reshape(-1)
        h_5[:] = np.concatenate([pt.reshape(-1), sc[:96]])
        h_5[31::32] &= 0x1f
        good_i = int(np.nonzero(~bad)[0][0])
        for b, a, wdt in zip(hv, v_host, (64, 64, 32, 32)):
            b[:] = a[good_i * wdt:(good_i + 1) * wdt]
        calls = {"fixed_base": lambda: ctx.lib.bjj_mul_fixed_base(ctx.handle, h_s.ctypes.data, one, h_o.ctypes.data),
                 "var_base": lambda: ctx.lib.bjj_mul_var_base(ctx.handle, h_p.ctypes.data, h_s.ctypes.data, one, h_o.ctypes.data),
                 "poseidon5": lambda: ctx.lib.bjj_poseidon5(ctx.handle, h_5.ctypes.data, one, h_h.ctypes.data),
                 "verify": lambda: ctx.lib.bjj_eddsa_verify(ctx.handle, hv[0].ctypes.data, hv[1].ctypes.data, hv[2].ctypes.data, hv[3].ctypes.data, one, h_ok.ctypes.data)}
        res, good = {}, True
        for k, f in calls.items():
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.2:
                ctx._ck(f(), k)
            ts = []
            for _ in range(40):
                t = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t)
            res[k] = float(np.median(ts)) * 1e6
            if k == "fixed_base":
                good = good and bool((np.asarray(h_o) == orc.mul_fixed_base(sc.reshape(n, 32)[0:1]).reshape(-1)).all())
            elif k == "var_base":
                good = good and bool((np.asarray(h_o) == orc.mul_var_base(pt, sc.reshape(n, 32)[0:1]).reshape(-1)).all())
            elif k == "poseidon5":
                good = good and bool((np.asarray(h_h) == orc.poseidon5(np.asarray(h_5).reshape(1, 160)).reshape(-1)).all())
            else:
                good = good and int(np.asarray(h_ok)[0]) == 1
        i = ctx.info()
        for b in [h_p, h_s, h_o, h_5, h_h, h_ok] + hv:
            ctx.host_free(b)
        return dict(res, unit="microseconds per n = 1 call, pinned host pointers, median of 40", parity_ok=good,
                    forms={"var_base": i.last_var_base_form, "poseidon5": i.last_poseidon_form, "verify": i.last_verify_dispatch})

    # the same inputs through the device-pointer entry point, ONE launch at a time: what a synchronous call can be held against
    # (the two-stream rate of `also.verify` lives on the overlap of consecutive launches, which a synchronous call cannot have)
    def one_launch():
        ctx.eddsa_verify_dev(d_pk.data_ptr(), d_r.data_ptr(), d_s.data_ptr(), d_m.data_ptr(), n, d_ok.data_ptr(), 0)
        ctx.sync()
    t_dev, _ = best(one_launch, 3)
    d_sc = up(sc)
    d_pts, d_vb = torch.empty(n * 64, dtype=torch.uint8, device=dev), torch.empty(n * 64, dtype=torch.uint8, device=dev)
    ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, d_pts.data_ptr(), 0)

    def one_launch_vb():
        ctx.mul_var_base_dev(d_pts.data_ptr(), d_sc.data_ptr(), n, d_vb.data_ptr(), 0)
        ctx.sync()
    t_dev_vb, _ = best(one_launch_vb, 3)

    def one_launch_vc():
        ctx.eddsa_verify_compressed_dev(d_pkw.data_ptr(), d_sigw.data_ptr(), d_mw.data_ptr(), n, d_ok.data_ptr(), 0)
        ctx.sync()
    t_dev_vc, _ = best(one_launch_vc, 3)
    # the hwmon poller reads the SMU every 5 ms; round 6 found that such reads can hold the SOC clock -- and with it the copy engines'
    # device-to-host rate -- up (profiles/r06_host_d2h_power_states.txt): the host-pointer rows are measured WITHOUT it
    if TELEMETRY:
        TELEMETRY.paused = True
    try:
        pinned = run("pinned")
        pageable = run("pageable")
        single = single_calls()
    finally:
        if TELEMETRY:
            TELEMETRY.paused = False
    return {"telemetry_paused": True, "single_call": single,
            "note": "PCIe-inclusive: host pointers in, host pointers out, synchronous call.  fixed_base / verify = caller arrays in "
                    "pinned memory (bjj_host_alloc): copied directly, chunked pipeline (2^15 items first, doubling to 2^18; "
                    "verify 2^16 to 2^19, its off-curve items as one launch beside the chunks'), chunk kernels alternating over the context's two "
                    "compute streams; verify inputs = the cfg-4 workload of `also.verify` (1 in 64 corrupted), `vs_device_one_launch` = the call "
                    "against one device-pointer launch on the same inputs; `pageable` = the same calls on ordinary memory, "
                    "staged through pinned buffers by %d copy workers.  Calls back to back for 0.5 s before the timed ones (sustained "
                    "clocks), then best of 7 (fixed_base) / 3 (verify) and the median.  Reported beside the line, never as `value`."
                    % ctx.info().host_copy_threads,
            "fixed_base": pinned["fixed_base"], "fixed_base_compressed": pinned["fixed_base_compressed"], "var_base": pinned["var_base"],
            "verify": pinned["verify"], "verify_compressed": pinned["verify_compressed"],
            "pageable": {"fixed_base": pageable["fixed_base"], "fixed_base_compressed": pageable["fixed_base_compressed"],
                         "var_base": pageable["var_base"], "verify": pageable["verify"], "verify_compressed": pageable["verify_compressed"]},
            "copy_threads": ctx.info().host_copy_threads,
            "parity_sample_ok": pinned["parity_sample_ok"] and pageable["parity_sample_ok"] and single["parity_ok"]}


def all_max(x, world, red_dev):
    t = torch.tensor([x], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_and(flag, world, red_dev):
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


# ---------------------------------------------------------------------------------------------------------------
def run_native_multi(args):
    """ONE process, all GPUs through the C ABI: bjj_multi_init + bjj_*_multi_dev (RCCL scatter / kernels / gather inside
    libbjj_hip.so).  Times BASELINE cfg 5's shape: the batch lives in the HBM of device 0."""
    import babyjubjub_rs_amd as bjj
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    g = len(devs)
    m = bjj.MultiContext(devs, args.window_bits, args.transport)
    dev = torch.device("cuda", m.device(0))
    torch.cuda.set_device(dev)
    ctx0 = m.ctx(0)
    stream = torch.cuda.Stream(device=dev)
    orc = get_oracle()
    out = {}
    ok_all = True
    for kind, n, steps in (("fixed_base", args.batch * g, 10), ("verify", args.strong_total if not args.no_strong else args.batch * g, 3)):
        wl = Workload(ctx0, kind, n, 0, dev, stream, nb=1)
        B = wl.batches[0]
        if kind == "fixed_base":
            call = lambda: m.mul_fixed_base_dev(B.d_sc.data_ptr(), n, B.d_out.data_ptr())  # noqa: E731
        else:
            call = lambda: m.eddsa_verify_dev(B.d_pk.data_ptr(), B.d_r.data_ptr(), B.d_s.data_ptr(), B.d_msg.data_ptr(), n, B.d_out.data_ptr())  # noqa: E731
        def timed(chunks):
            m.set_chunks(chunks)
            call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tim = []
            for _ in range(steps):
                call()
                tim.append(m.last_timing())
            dt = time.perf_counter() - t0
            mean = lambda k: float(np.mean([t[k] for t in tim]))  # noqa: E731
            return {"value": n * steps / dt, "ms_per_step": dt / steps * 1e3, "scatter_ms": mean("scatter_ms"),
                    "compute_ms": mean("compute_ms"), "gather_ms": mean("gather_ms"), "total_ms": mean("total_ms"),
                    "wall_ms": mean("wall_ms"), "chunks": tim[-1]["chunks"], "rccl_version": tim[-1]["rccl_version"]}
        serial = timed(1)                 # scatter everything -> kernels -> gather everything
        piped = timed(args.chunks)        # peer blocks in pieces, transfers behind the kernels
        ok = wl.check_sample(orc)
        ok_all = ok_all and ok
        # reference point: the same batch as ONE launch of the first context (no handle, no blocks, no transfers)
        wl.launch(0)
        stream.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            wl.launch(0)
        stream.synchronize()
        dt1 = time.perf_counter() - t0
        ok1 = wl.check_sample(orc)
        ok_all = ok_all and ok1
        out[kind] = dict(piped, unit=UNITS[kind], items=n, steps=steps, parity_sample_ok=ok,
                         serial_schedule=serial,
                         one_context_one_launch={"value": n * steps / dt1, "ms_per_step": dt1 / steps * 1e3, "parity_sample_ok": ok1},
                         note="spans are HIP-event intervals (max over devices); in the pipelined schedule they overlap: "
                              "total_ms is what the call took, scatter + compute + gather what a serial schedule pays")
        del wl
    res = {"metric": "BabyJubJub native multi-GPU (bjj_multi_*): fixed-base mults/sec and EdDSA verifies/sec, batch resident on device 0",
           "value": out["verify"]["value"], "unit": "verifies/s", "n_gpus": g, "devices": [m.device(i) for i in range(g)],
           "mode": ("single process, ncclCommInitAll; peer blocks in pieces: one group of ncclSend / ncclRecv pairs per piece, kernels "
                    "per piece behind its arrival, results back per piece -- inside libbjj_hip.so" if args.transport == "rccl"
                    else "single process, hipMemcpyPeerAsync of the pieces on the peers' transfer streams / kernels per piece / "
                         "hipMemcpyPeerAsync of the results -- inside libbjj_hip.so"),
           "transport": args.transport,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4] shape", "window_bits": ctx0.info().window_bits},
           "results": out, "parity_sample_ok": ok_all}
    publish(res)
    m.close()
    return 0 if ok_all else 3


# ---------------------------------------------------------------------------------------------------------------
# The headline line must not be hostage to the optional sections (`also`, `strong`): those are the only code of this file that
# has never run over real multi-GPU RCCL (point-to-point groups of cfg 5's rank-0-resident shape).  Once the headline is
# measured, a watchdog bounds everything that follows: if the optional sections raise on this rank, or are not finished by
# the deadline (a rank stuck in a collective), rank 0 prints the headline line with what is complete plus a note, and every rank
# leaves through os._exit(4) -- no further collective, no destructor that could block on a wedged communicator; the exit code
# says that the run was cut short (3 stays reserved for a failed parity comparison).
# ---------------------------------------------------------------------------------------------------------------
class Headline:
    def __init__(self, rank, result, deadline_s):
        import threading
        self.rank, self.result, self.deadline_s = rank, result, deadline_s
        self.sections = {}            # name -> dict, filled as the optional sections complete
        self.parity = True
        self.lock = threading.Lock()
        self.printed = False
        self.done = threading.Event()
        self.t0 = time.monotonic()
        self.thread = threading.Thread(target=self._watch, daemon=True)
        self.thread.start()

    def emit(self, note=None):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.rank == 0:
                r = self.result
                for k, v in self.sections.items():
                    if v:
                        r[k] = v
                r["parity_sample_ok"] = self.parity
                if note:
                    r["optional_sections"] = note
                if not self.parity:
                    r["value"] = None      # a miscomputing build publishes no number
                publish(r)

    def abandon(self, note):
        """print what is complete and leave at once (called from the watchdog, or after an exception in an optional section)"""
        self.emit(note)
        sys.stderr.write("bench.py rank %d: %s\n" % (self.rank, note))
        sys.stderr.flush()
        # 3 = a parity comparison failed (no number published); 4 = the headline is complete and correct but an optional section
        # raised or hung -- a process that has touched the GPU and is abandoned must not report success (CI, torchrun and the
        # session scripts only see the exit code)
        os._exit(4 if self.parity else 3)

    def _watch(self):
        if not self.done.wait(self.deadline_s):
            self.abandon("abandoned after %.0f s (BJJ_BENCH_OPTIONAL_DEADLINE_S): headline complete, optional sections incomplete"
                         % (time.monotonic() - self.t0))


def main():
    args = parse()
    global DETAIL_OUT
    DETAIL_OUT = args.detail_out
    if args.steps is None:
        args.steps = 200 if args.workload == "fixed_base" else 20
    if args.warmup is None:
        args.warmup = 50 if args.workload == "fixed_base" else 3
    if args.native_multi:
        return run_native_multi(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_children(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
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
    try:
        ctx = bjj.Context(local_rank, args.window_bits)
    except bjj.BjjError as e:
        if "cannot allocate the fixed-base table" not in str(e):
            raise
        # the requested table does not fit next to the GPU's other tenants: take the widest that does, and say so in `config`
        ctx = bjj.Context(local_rank, bjj.WINDOW_AUTO)
    if args.signer_constant_time:
        ctx.set_signer_constant_time(True)
    n = args.batch
    ctx.reserve(max(n, 1))
    stream = torch.cuda.Stream(device=dev)
    stream_b = torch.cuda.Stream(device=dev)
    # HIP binds a stream to one of the four hardware queues of its priority when the stream is first USED: both timing streams
    # are touched now, before anything else creates streams, so that they get a queue each (two streams that end up on one
    # hardware queue run their launches one after the other: profiles/r05_host_pipeline.txt)
    for st_ in (stream, stream_b):
        with torch.cuda.stream(st_):
            torch.zeros(64, device=dev).add_(1)
    torch.cuda.synchronize()

    def second_stream(k):
        ns = args.streams if args.streams is not None else DEFAULT_STREAMS.get(k, 1)
        return stream_b if ns == 2 else None

    kind = args.workload
    orc = get_oracle()
    one_gpu = world == 1
    parity = True
    global TELEMETRY
    TELEMETRY = GpuTelemetry(local_rank).start()

    # ---- headline: weak scaling, every rank its own block(s) of the global batch
    wl, dt, kernel_ms, extra, ok, cb = measure(ctx, kind, n, rank * n, dev, stream, args.steps, args.warmup, args.warmup_seconds,
                                               world, args.batches, orc, one_gpu and not args.no_cpu_baseline, rank,
                                               stream2=second_stream(kind))
    parity = parity and ok
    dt_max = all_max(dt, world, red_dev)
    info = ctx.info()
    result = None
    if rank == 0:
        result = {
            "metric": "BabyJubJub %s, %d-item batch per GPU" % (METRIC[kind], n),
            "value": world * n * args.steps / dt_max, "unit": UNITS[kind], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": backend if world > 1 else None,
            "config": {"workload": WORKLOAD_TEXT[kind], "batch_per_gpu": n, "global_batch": n * world,
                       "window_bits": info.window_bits, "window_bits_requested": args.window_bits,
                       "fixed_base_table_mb": info.table_bytes / 1e6, "table_bytes": info.table_bytes,
                       "init_ms": info.init_ms, "library_default_window_bits": 23,
                       "limbs": "9 x 29-bit, 64-bit column accumulators (v_mad_u64_u32)",
                       "warmup_seconds": args.warmup_seconds, "resident_batches": args.batches,
                       "signer_constant_time": bool(info.signer_constant_time),
                       "streams": extra.get("streams", 1), "kernel": extra.get("kernel"),
                       "parallelism": "independent shards, one process per GPU, no data-path collective"},
        }
        result["roofline"], ro = roofline_blocks(kind, kernel_ms, n, info, extra)
        if ro:
            result["roofline_overlapped"] = ro
        result.update(extra)
        ck = extra.get("clock") or {}
        result["clock_mhz"] = ck.get("sclk_mhz")      # sampled across the timed region that produced `value`
        result["socket_w"] = ck.get("socket_w")
        vb = valu_block(kind, kernel_ms, n, info, one_stream_clock(extra))
        if vb:
            result["valu"] = vb
        if cb:
            result["cpu_baseline"] = cb
    del wl
    hl = Headline(rank, result, float(os.environ.get("BJJ_BENCH_OPTIONAL_DEADLINE_S", "240")))
    hl.parity = parity
    try:
        parity, devices = optional_sections(args, hl, ctx, bjj, kind, n, rank, local_rank, world, dev, red_dev, stream, stream_b,
                                            second_stream, orc, one_gpu, info, parity)
    except Exception as e:   # never a collective after a failure: the other ranks' watchdogs end them
        import traceback
        traceback.print_exc()
        hl.abandon("optional section failed on rank %d: %s: %s -- headline complete" % (rank, type(e).__name__, e))
    hl.done.set()
    hl.parity = parity
    if rank == 0:
        result["devices"] = devices
    hl.emit()
    TELEMETRY.stop()
    ctx.close()
    return 0 if parity else 3


def optional_sections(args, hl, ctx, bjj, kind, n, rank, local_rank, world, dev, red_dev, stream, stream_b, second_stream, orc,
                      one_gpu, info, parity):
    """`also` and `strong` of the bench line (module docstring); fills hl.sections as each completes; returns (parity, devices)"""
    # ---- the other halves of BASELINE's metric, same protocol (20 steps / 3 warm-up / warm-up by time)
    also = {}
    if not args.no_also:
        for k2 in ("verify", "var_base"):
            if k2 == kind:
                continue
            s2, w2 = 20, 3
            nb2 = min(args.batches, 2)
            wl2, d2, km2, ex2, ok2, cb2 = measure(ctx, k2, n, rank * n, dev, stream, s2, w2, args.warmup_seconds, world, nb2, orc,
                                                  one_gpu and not args.no_cpu_baseline, rank, stream2=second_stream(k2))
            parity = parity and ok2
            d2m = all_max(d2, world, red_dev)
            if rank == 0:
                also[k2] = {"metric": "BabyJubJub %s, %d-item batch per GPU" % (METRIC[k2], n), "value": world * n * s2 / d2m,
                            "unit": UNITS[k2], "steps": s2, "warmup": w2, "ms_per_step": d2m / s2 * 1e3, "batch_per_gpu": n,
                            "workload": WORKLOAD_TEXT[k2], "parity_sample_ok": ok2}
                also[k2]["roofline"], ro2 = roofline_blocks(k2, km2, n, info, ex2)
                if ro2:
                    also[k2]["roofline_overlapped"] = ro2
                also[k2].update(ex2)
                ck2 = ex2.get("clock") or {}
                also[k2]["clock_mhz"], also[k2]["socket_w"] = ck2.get("sclk_mhz"), ck2.get("socket_w")
                vb2 = valu_block(k2, km2, n, info, one_stream_clock(ex2))
                if vb2:
                    also[k2]["valu"] = vb2
                if cb2:
                    also[k2]["cpu_baseline"] = cb2
            del wl2
        # reference value of the headline kernel with the library's DEFAULT table (23-bit windows, 5.9 GB)
        if kind == "fixed_base" and one_gpu and info.window_bits != 23:
            c23 = bjj.Context(local_rank, 23)
            c23.reserve(n)
            w23 = Workload(c23, "fixed_base", n, rank * n, dev, stream, nb=2)
            d23, k23 = timed_steps(w23, 100, 20, 1, 0.5)
            ok23 = w23.check_sample(orc)
            parity = parity and ok23
            ck23 = w23.clock or {}
            also["fixed_base_window_bits_23"] = {"value": n * 100 / d23, "unit": UNITS["fixed_base"], "kernel_ms_avg": k23,
                                                 "table_bytes": c23.info().table_bytes, "init_ms": c23.info().init_ms,
                                                 "streams": 1, "kernel": kernel_that_ran("fixed_base", c23.info(), False),
                                                 "clock_mhz": ck23.get("sclk_mhz"), "socket_w": ck23.get("socket_w"), "clock": ck23,
                                                 "parity_sample_ok": ok23}
            del w23
            c23.close()

        # K1 with Point::compress fused into its epilogue (bjj_mul_fixed_base_compressed_dev): same protocol as the headline
        if kind == "fixed_base" and one_gpu:
            wc = Workload(ctx, "fixed_base_compressed", n, rank * n, dev, stream, nb=2)
            wc.streams = [stream, stream_b]
            dc, kc = timed_steps(wc, 100, 20, 1, 0.5, streams=[stream, stream_b])
            okc = wc.check_sample(orc)
            parity = parity and okc
            also["fixed_base_compressed"] = {"value": n * 100 / dc, "unit": UNITS["fixed_base"], "ms_per_step": dc / 100 * 1e3, "device_ms_per_launch": kc,
                                             "streams": 2, "kernel": kernel_that_ran("fixed_base_compressed", ctx.info(), True),
                                             "workload": WORKLOAD_TEXT["fixed_base_compressed"], "parity_sample_ok": okc}
            del wc

        # PCIe-inclusive rates of the host-pointer API -- what a host holding its data in ordinary (pageable) memory gets from
        # bjj_mul_fixed_base / bjj_eddsa_verify: the chunked host-pointer pipeline around the same kernels (pinned memory copied directly,
        # pageable memory staged by copy workers).  NEVER `value`.
        if one_gpu and rank == 0:
            also["host_api"] = host_api_block(ctx, n, orc)
            parity = parity and also["host_api"]["parity_sample_ok"]

    hl.sections["also"] = also
    hl.parity = parity

    # ---- strong scaling: fixed total work over the N ranks
    strong = {}
    if not args.no_strong:
        from babyjubjub_rs_amd import shard, workload as w
        # (1) 2^20 fixed-base multiplications in total
        tot = 1 << 20
        lo, hi = w.shard_bounds(tot, world, rank)
        ws1 = Workload(ctx, "fixed_base", hi - lo, lo, dev, stream, nb=2)
        s1 = 100
        d1, km1 = timed_steps(ws1, s1, 10, world, 0.3)
        ok1 = ws1.check_sample(orc, 128)
        parity = parity and ok1
        d1m = all_max(d1, world, red_dev)
        strong["fixed_base_1M_total"] = {"value": tot * s1 / d1m, "unit": UNITS["fixed_base"], "total_items": tot, "steps": s1,
                                         "ms_per_step": d1m / s1 * 1e3, "kernel_ms_rank0": km1}
        del ws1
        # (2) BASELINE configs[4]: 2^24 verifies in total, contiguous ceil(n/G) blocks
        tot = args.strong_total
        lo, hi = w.shard_bounds(tot, world, rank)
        m = hi - lo
        ctx.reserve(m)
        ws2 = Workload(ctx, "verify", m, lo, dev, stream, nb=1)
        s2 = 3
        d2, km2 = timed_steps(ws2, s2, 1, world, 0.0)
        ok2 = ws2.check_sample(orc, 128)
        parity = parity and ok2
        d2m = all_max(d2, world, red_dev)
        line = {"value": tot * s2 / d2m, "unit": UNITS["verify"], "total_items": tot, "steps": s2, "ms_per_step": d2m / s2 * 1e3,
                "kernel_ms_rank0": km2, "mode": "pre-sharded: every rank holds its block"}
        if world > 1:
            # cfg 5's literal shape: rank 0 holds the whole batch; exact-size blocks out, verdicts back, one posted group each way
            B = ws2.batches[0]
            names, rows_b = ["d_pk", "d_r", "d_s", "d_msg"], [64, 64, 32, 32]
            fulls = [shard.gather_array(getattr(B, nm), tot, rb, dev) for nm, rb in zip(names, rows_b)]   # untimed setup
            recv = [getattr(B, nm) for nm in names]                       # the blocks land in the rank's own input tensors
            out_full = torch.empty(tot, dtype=torch.uint8, device=dev) if rank == 0 else None
            t_parts = []
            for it in range(1 + 2):
                dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                blocks = shard.scatter_arrays(fulls if rank == 0 else [None] * 4, tot, rows_b, dev, 0, recv)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ctx.eddsa_verify_dev(blocks[0].data_ptr(), blocks[1].data_ptr(), blocks[2].data_ptr(), blocks[3].data_ptr(), m,
                                     B.d_out.data_ptr(), stream.cuda_stream)
                stream.synchronize()
                t2 = time.perf_counter()
                shard.gather_array(B.d_out, tot, 1, dev, 0, out_full)
                dist.barrier(); torch.cuda.synchronize()
                t3 = time.perf_counter()
                if it:
                    t_parts.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
            tp = np.mean(np.array(t_parts), axis=0)
            tmax = all_max(float(tp[3]), world, red_dev)
            # the same shape PIPELINED (shard.scatter_compute_gather_pipelined = the schedule of bjj_multi_* in the library):
            # peer blocks travel in pieces, a peer verifies piece c while piece c+1 arrives (pieces alternate over two streams =
            # the context's two scratch sets), rank 0 verifies its own block from t = 0 while its sends are in flight
            def compute_piece(arrays, count, out_view, k):
                st = (stream, stream_b)[k % 2]
                ctx.eddsa_verify_dev(arrays[0].data_ptr(), arrays[1].data_ptr(), arrays[2].data_ptr(), arrays[3].data_ptr(), count,
                                     out_view.data_ptr(), st.cuda_stream)

            t_pipe = []
            for it in range(1 + 2):
                if rank == 0:
                    out_full.zero_()
                dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                shard.scatter_compute_gather_pipelined(fulls if rank == 0 else [None] * 4, rows_b, tot, compute_piece, 1, dev, 0,
                                                       pieces=4, recv=recv, res=B.d_out, out=out_full, streams=[stream, stream_b])
                stream.synchronize(); stream_b.synchronize()
                dist.barrier(); torch.cuda.synchronize()
                if it:
                    t_pipe.append(time.perf_counter() - t0)
            tpipe = all_max(float(np.mean(t_pipe)), world, red_dev)
            if rank == 0:   # every verdict of the gathered vector against the corruption mask of the GLOBAL batch
                r_ = w.splitmix64(w.SEED_BAD, tot, 0)
                okg = bool(((out_full.cpu().numpy() == 1) == ((r_ & np.uint64(63)) != 0)).all())
                parity = parity and okg
                line["rank0_resident"] = {"value": tot / tpipe, "unit": UNITS["verify"], "ms_per_step": tpipe * 1e3,
                                          "gathered_verdicts_ok": okg, "pieces_per_peer_block": len(w.piece_bounds(m, 4)),
                                          "mode": "rank 0 holds all inputs; pipelined: peer blocks in pieces (one RCCL group of exact "
                                                  "send / receive pairs per piece), kernels per piece behind its arrival, results back "
                                                  "per piece, rank 0 computes from t = 0",
                                          "serial_schedule": {"value": tot / tmax, "ms_per_step": tmax * 1e3, "scatter_ms": float(tp[0]) * 1e3,
                                                              "kernel_ms": float(tp[1]) * 1e3, "gather_ms": float(tp[2]) * 1e3,
                                                              "mode": "scatter (exact blocks, one group) -> kernels -> gather, "
                                                                      "synchronised between the phases"}}
            del fulls
        strong["verify_16M_total_cfg5" if tot == (1 << 24) else "verify_total"] = line
        del ws2

    hl.sections["strong"] = strong
    hl.parity = parity

    parity = all_and(parity, world, red_dev)
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, torch.cuda.current_device())
        dist.barrier()
        dist.destroy_process_group()
    else:
        devices = [torch.cuda.current_device()]
    return parity, devices


if __name__ == "__main__":
    sys.exit(main())
