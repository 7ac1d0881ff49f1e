#!/usr/bin/env python3
"""Developer tool (tools/scale_session.sh): the one-process multi-GPU form of the C ABI (bjj_*_multi_dev) over a transport, at the
batch shapes the block arithmetic has to get right -- n a multiple of G, ragged n, n < G (empty ranks) -- with 1 and 4 pieces per
peer block; every result is compared with ONE context's result on the same inputs (bit-exact) and a sample with the oracle.
With --expect-failure (set BJJ_MULTI_INJECT_FAIL_GROUP=k in the environment) the first *_multi_dev call must fail with the
injected error, the handle must refuse every later call, bjj_multi_free must return, and a fresh handle must work.

  python3 tools/native_multi_cases.py --devices 0,1,2,3,4,5,6,7 --transport rccl [--window-bits 23] [--out profiles/x.json]
  python3 tools/native_multi_cases.py --devices 0,0,0,0,0,0,0,0 --transport peer      (one-GPU dry run of the same code)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default=None, help="comma separated (default: every visible device)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "peer"])
    ap.add_argument("--window-bits", type=int, default=int(os.environ.get("BJJ_BENCH_WINDOW_BITS", "23")))
    ap.add_argument("--per-gpu", type=int, default=1 << 18, help="items per device of the even case")
    ap.add_argument("--expect-failure", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    from conftest import Oracle
    devs = [int(d) for d in a.devices.split(",")] if a.devices else list(range(torch.cuda.device_count()))
    G = len(devs)
    orc = Oracle()
    m = bjj.MultiContext(devs, a.window_bits, a.transport)
    dev0 = torch.device("cuda", m.device(0))
    torch.cuda.set_device(dev0)
    ref = m.ctx(0)                      # one context, one launch: the reference result
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x).reshape(-1)).to(dev0)  # noqa: E731
    res = {"devices": devs, "transport": a.transport, "window_bits": ref.info().window_bits, "cases": []}

    if a.expect_failure:
        n = G * 4096
        d_sc, d_out = up(w.scalars_254(n)), torch.zeros(n * 64, dtype=torch.uint8, device=dev0)
        m.set_chunks(1)
        t0 = time.perf_counter()
        try:
            m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
            raise SystemExit("expected the injected failure (BJJ_MULTI_INJECT_FAIL_GROUP=%s), the call succeeded" % os.environ.get("BJJ_MULTI_INJECT_FAIL_GROUP"))
        except bjj.BjjError as e:
            assert "injected failure" in str(e) and "unusable" in str(e), str(e)
            first = str(e)
        try:
            m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
            raise SystemExit("a retired handle accepted another call")
        except bjj.BjjError:
            pass
        m.close()                                           # must return (bounded drain + ncclCommAbort)
        dt = time.perf_counter() - t0
        os.environ.pop("BJJ_MULTI_INJECT_FAIL_GROUP", None)
        m2 = bjj.MultiContext(devs, a.window_bits, a.transport)   # the process is still healthy: new communicators work
        m2.set_chunks(1)
        m2.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
        idx = np.arange(0, n, max(1, n // 97))
        ok = bool((d_out.cpu().numpy().reshape(n, 64)[idx] == orc.mul_fixed_base(w.scalars_254(n)[idx])).all())
        m2.close()
        res["injected_failure"] = {"error": first[:300], "retire_and_free_s": dt, "fresh_handle_ok": ok}
        print(json.dumps(res))
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)
        return 0 if ok else 3

    ok_all = True
    for label, n in (("even", G * a.per_gpu), ("ragged", G * a.per_gpu + 333), ("n_lt_G", max(1, G - 1))):
        sc = w.scalars_254(n, offset=7)
        d_sc = up(sc)
        d_pts = torch.zeros(n * 64, dtype=torch.uint8, device=dev0)
        ref.mul_fixed_base_dev(d_sc.data_ptr(), n, d_pts.data_ptr(), 0)
        ref.sync()
        want_fb = d_pts.clone()
        # verify inputs: valid signatures made by the signer kernels of the first context, 1 in 64 corrupted
        keys, msg = up(w.random_u256(w.SEED_KEYS, n, 0)), up(w.random_u256(w.SEED_MSGS, n, 0, top_bits_cleared=3))
        d_pk, d_r, d_s, d_f = (torch.zeros(n * 64, dtype=torch.uint8, device=dev0), torch.zeros(n * 64, dtype=torch.uint8, device=dev0),
                               torch.zeros(n * 32, dtype=torch.uint8, device=dev0), torch.zeros(n, dtype=torch.uint8, device=dev0))
        ref.public_keys_dev(keys.data_ptr(), n, d_pk.data_ptr(), 0)
        ref.sign_dev(keys.data_ptr(), msg.data_ptr(), n, d_r.data_ptr(), d_s.data_ptr(), d_f.data_ptr(), 0)
        ref.sync()
        bad = w.corrupt(d_pk.view(n, 64), d_r.view(n, 64), d_s.view(n, 32), msg.view(n, 32), n, 0)
        torch.cuda.synchronize()
        for chunks in (1, 4):
            m.set_chunks(chunks)
            d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev0)
            d_ok = torch.full((max(n, 16),), 9, dtype=torch.uint8, device=dev0)
            t0 = time.perf_counter()
            m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
            t_fb = time.perf_counter() - t0
            tim_fb = m.last_timing()
            t0 = time.perf_counter()
            m.eddsa_verify_dev(d_pk.data_ptr(), d_r.data_ptr(), d_s.data_ptr(), msg.data_ptr(), n, d_ok.data_ptr())
            t_v = time.perf_counter() - t0
            tim_v = m.last_timing()
            fb_ok = bool(torch.equal(d_out, want_fb))
            v_ok = bool((d_ok[:n].cpu().numpy() == (~bad).astype(np.uint8)).all())
            idx = np.unique(np.linspace(0, n - 1, min(n, 64)).astype(np.int64))
            orc_ok = bool((d_out.cpu().numpy().reshape(n, 64)[idx] == orc.mul_fixed_base(sc[idx])).all())
            ok_all = ok_all and fb_ok and v_ok and orc_ok
            res["cases"].append({"case": label, "n": n, "chunks_requested": chunks, "chunks_used": tim_v["chunks"],
                                 "fixed_base_equals_one_context": fb_ok, "verdicts_equal_corruption_mask": v_ok, "oracle_sample_ok": orc_ok,
                                 "fixed_base_wall_ms": t_fb * 1e3, "verify_wall_ms": t_v * 1e3, "fixed_base_timing": tim_fb, "verify_timing": tim_v,
                                 "blocks": [m.shard_bounds(n, r) for r in range(G)]})
    res["ok"] = ok_all
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    m.close()
    return 0 if ok_all else 3


if __name__ == "__main__":
    sys.exit(main())
