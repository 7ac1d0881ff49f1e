#!/usr/bin/env python3
"""
Regenerates tests/golden/*.json.

  reference_kats.json   the known-answer vectors held by the reference's own tests
                        (data only: inputs and expected outputs, with the src/lib.rs
                        line range each came from)
  oracle_vectors.json   seeded random + edge-case vectors produced by the pure-Python
                        oracle (oracle/bjj_oracle.py), whose behaviour is pinned by the
                        KATs above.  Inputs use the SplitMix64 streams of SURVEY.md 8(d),
                        so they are a prefix of the benchmark workload.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bjj_oracle as o  # noqa: E402

Q = o.Q


def hx(v):
    return "0x%064x" % v


def kats():
    P = (17777552123799933955779906779655732241715742912184938656739573121738514868268,
         2626589144620713026669568689430873010625803728049924121243784502389097019475)
    Qp = (16540640123574156134436876038791482806971768689494387082833631921987005038935,
          20819045374670962167435360035096875258406992893633759881276124905556507972311)
    return {
        "add_same_point": {"src": "src/lib.rs:421-459", "p": P, "q": P, "sum": (
            6890855772600357754907169075114257697580319025794532037257385534741338397365,
            4338620300185947561074059802482547481416142213883829469920100239455078257889)},
        "add_different_points": {"src": "src/lib.rs:461-499", "p": P, "q": Qp, "sum": (
            7916061937171219682591368294088513039687205273691143098332585753343424131937,
            14035240266687799601661095864649209771790948434046947201833777492504781204499)},
        "mul_scalar": {"src": "src/lib.rs:502-552", "p": P, "cases": [
            {"n": 3, "out": (19372461775513343691590086534037741906533799473648040012278229434133483800898,
                             9458658722007214007257525444427903161243386465067105737478306991484593958249)},
            {"n": 14035240266687799601661095864649209771790948434046947201833777492504781204499,
             "out": (17070357974431721403481313912716834497662307308519659060910483826664480189605,
                     4014745322800118607127020275658861516666525056516280575712425373174125159339)}]},
        "point_compress": {"src": "src/lib.rs:575-594", "p": P,
                           "hex": "53b81ed5bffe9545b54016234682e7b2f699bd42a5e9eae27ff4051bc698ce85"},
        "point_decompress": {"src": "src/lib.rs:597-632", "cases": [
            {"y_bytes": "b5328f8791d48f20bec6e481d91c7ada235f1facf22547901c18656b6c3e042f",
             "x_le_bytes": "b86cc8d9c97daef0afe1a4753c54fb2d8a530dc74c7eee4e72b3fdf2496d2113"},
            {"y_bytes": "70552d3ff548e09266ded29b33ce75139672b062b02aa66bb0d9247ffecf1d0b",
             "x_le_bytes": "30f1635ba7d56f9cb32c3ffbe6dca508a68c7f43936af11a23c785ce98cb3404"}]},
        "circomlib_testvector": {
            "src": "src/lib.rs:689-738",
            "key": "0001020304050607080900010203040506070809000102030405060708090001",
            "blake512": "c992db23d6290c70ffcc02f7abeb00b9d00fa8b43e55d7949c28ba6be7545d3253882a61bd004a236ef1cdba01b27ba0aedfb08eefdbfb7c19657c880b43ddf1",
            "scalar_key": 6466070937662820620902051049739362987537906109895538826186780010858059362905,
            "pk": (0x1d5ac1f31407018b7d413a4f52c8f74463b30e6ac2238220ad8b254de4eaa3a2,
                   0x1e1de8a908826c3f9ac2e0ceee929ecd0caf3b99b3ef24523aaab796a6f733c4),
            "msg": int.from_bytes(bytes.fromhex("00010203040506070809"), "little"),
            "r_b8": (0x192b4e51adf302c8139d356d0e08e2404b5ace440ef41fc78f5c4f2428df0765,
                     0x2202bebcf57b820863e0acc88970b6ca7d987a0d513c2ddeb42e3f5d31b4eddf),
            "s": 1672775540645840396591609181675628451599263765380031905495115170613215233181,
            "verify": True},
        "bench_inputs": {"src": "benches/bench_babyjubjub.rs:15-38", "p": P, "scalars": [
            3, 2626589144620713026669568689430873010625803728049924121243784502389097019475]},
        "poseidon_public": {"src": "public circomlib/iden3 values (not in /root/reference); SURVEY.md Appendix C",
                            "cases": [
                                {"in": [1, 2, 0, 0, 0], "out": 1018317224307729531995786483840663576608797660851238720571059489595066344487},
                                {"in": [1, 2, 3, 4, 5], "out": 6183221330272524995739186171720101788151706631170188140075976616310159254464}]},
    }


def oracle_vectors():
    out = {"generator": "SplitMix64, seeds per SURVEY.md 8(d)"}
    rs = o.SplitMix64(o.SEED_SCALARS)
    scalars = [rs.u256() & ((1 << 254) - 1) for _ in range(48)]
    edge_scalars = [0, 1, 2, 7, 8, 9, 15, 16, 17, o.SUBORDER - 1, o.SUBORDER, o.SUBORDER + 1, o.ORDER - 1, o.ORDER,
                    o.ORDER + 1, Q, (1 << 254) - 1, 1 << 254, 1 << 255, (1 << 256) - 1,
                    int("8" * 64, 16), int("7" * 64, 16), int("f" * 63, 16)]
    out["fixed_base"] = [{"n": hx(n), "out": [hx(v) for v in o.mul_scalar(o.B8, n)]} for n in scalars[:24] + edge_scalars]

    rp = o.SplitMix64(o.SEED_POINTS)
    pts = []
    for i in range(10):
        k = rp.u256() % o.SUBORDER
        c = rp.next() & 7
        P = o.mul_scalar(o.B8, k)
        T = o.mul_scalar(o.T8, c)
        pts.append(o.proj_affine(o.proj_add(P + (1,), T + (1,))))
    pts += [(0, 1), (0, Q - 1), o.T8, o.B8]
    off = [(0, 0), (1, 2), (5, 7), (Q - 1, Q - 1), (0, 2), (rp.u256() % Q, rp.u256() % Q)]
    vb = []
    for i, P in enumerate(pts + off):
        for n in [scalars[24 + (i % 24)], edge_scalars[i % len(edge_scalars)], edge_scalars[(i * 7 + 3) % len(edge_scalars)]]:
            vb.append({"p": [hx(P[0]), hx(P[1])], "on_curve": o.on_curve(P), "n": hx(n),
                       "out": [hx(v) for v in o.mul_scalar(P, n)]})
    out["var_base"] = vb

    rm = o.SplitMix64(o.SEED_MSGS)
    pos = []
    for i in range(12):
        ins = [rm.u256() % Q for _ in range(5)]
        pos.append({"in": [hx(v) for v in ins], "out": hx(o.poseidon(ins))})
    for ins in ([0] * 5, [Q - 1] * 5, [1, 0, 0, 0, 0], [0, 0, 0, 0, 1]):
        pos.append({"in": [hx(v) for v in ins], "out": hx(o.poseidon(ins))})
    out["poseidon5"] = pos

    adds = []
    allp = pts + off
    for i in range(len(allp)):
        p, q = allp[i], allp[(i * 5 + 1) % len(allp)]
        adds.append({"p": [hx(p[0]), hx(p[1])], "q": [hx(q[0]), hx(q[1])],
                     "out": [hx(v) for v in o.proj_affine(o.proj_add(p + (1,), q + (1,)))]})
    out["point_add"] = adds

    rk, rn = o.SplitMix64(o.SEED_KEYS), o.SplitMix64(o.SEED_NONCES)
    ver = []

    def rec(A, R, S, m, note):
        ver.append({"pk": [hx(A[0]), hx(A[1])], "r_b8": [hx(R[0]), hx(R[1])], "s": hx(S % (1 << 256)), "msg": hx(m),
                    "ok": bool(o.verify(A, R, S % (1 << 256), m)), "note": note})

    for i in range(6):
        k, rho, m = rk.u256() % o.SUBORDER, rn.u256() % o.SUBORDER, rm.u256() % Q
        A, R, S = o.sign_with_scalars(k, rho, m)
        rec(A, R, S, m, "valid")
        if i == 0:
            rec(A, R, S ^ 1, m, "bit flipped in s")
            rec(A, R, S + o.SUBORDER, m, "s + l (unreduced s is accepted: no s < l check, lib.rs:405)")
            rec(A, R, S + o.ORDER, m, "s + 8l")
            rec(A, R, S, (m + 1) % Q, "wrong msg")
            rec((A[0], (A[1] + 1) % Q), R, S, m, "pk off curve")
            rec(A, (R[0], (R[1] + 1) % Q), S, m, "R off curve")
            rec(A, R, S, Q + 5, "msg > Q -> false (lib.rs:396-398)")
            rec(A, (Q - R[0], R[1]), S, m, "-R")
    A0, R0, S0 = o.sign_with_scalars(5, 7, 0)
    rec(A0, R0, S0, 0, "msg = 0")
    rec(A0, R0, S0, Q, "msg == Q accepted and wraps to 0 (lib.rs:396-399)")
    rec((0, 0), (0, 0), 0, 0, "all zero (off curve)")
    rec((0, 1), (0, 1), 0, 0, "identity everywhere")
    rec(A0, (0, 1), 0, 0, "R = identity, s = 0")
    rec(o.T8, R0, S0, 0, "pk of order 8")
    out["verify"] = ver

    # ---- codec row: compress / decompress_point / verify on compressed inputs
    rc = o.SplitMix64(o.SEED_POINTS ^ 0xC0DEC)
    comp = []
    for P in pts + [(Q - p[0], p[1]) for p in pts[:6]]:
        comp.append({"in": o.compress(P).hex(), "note": "on-curve point"})
    for raw, note in ((o.to_le32(Q), "y == Q -> Err"), (o.to_le32(Q - 1), "y = Q-1 = -1: x^2 == 0 -> Err (modsqrt rejects 0)"),
                      (o.to_le32(1), "y = 1: x^2 == 0 -> Err"), (bytes([0xff] * 32), "all ones: y >= Q after clearing the sign"),
                      (bytes(32), "y = 0"), (o.to_le32(2), "y = 2"), (o.to_le32((1 << 255) | 2), "y = 2 with sign bit")):
        comp.append({"in": raw.hex(), "note": note})
    for i in range(24):
        v = (rc.u256() % Q) | ((rc.next() & 1) << 255)
        comp.append({"in": o.to_le32(v).hex(), "note": "random y"})
    for c in comp:
        d = o.decompress_point(bytes.fromhex(c["in"]))
        c["ok"] = d is not None
        c["out"] = [hx(d[0]), hx(d[1])] if d else [hx(0), hx(0)]
        if d:
            assert o.compress(d) == bytes.fromhex(c["in"])
    out["decompress"] = comp
    vc = []
    for rec in ver[:9] + ver[14:16]:
        A = tuple(int(v, 16) for v in rec["pk"]); R = tuple(int(v, 16) for v in rec["r_b8"])
        if not (o.on_curve(A) and o.on_curve(R)) or A[0] == 0 or R[0] == 0:
            continue  # only points that have a compressed form
        s_int = int(rec["s"], 16)
        vc.append({"pk": o.compress(A).hex(), "sig": o.compress_signature(R, s_int).hex(), "msg": rec["msg"],
                   "ok": 1 if rec["ok"] else 0, "note": rec["note"]})
    good = vc[0]
    vc.append({"pk": o.to_le32(Q).hex(), "sig": good["sig"], "msg": good["msg"], "ok": 2, "note": "pk does not decompress"})
    vc.append({"pk": good["pk"], "sig": o.to_le32(3).hex() + good["sig"][64:], "msg": good["msg"],
               "ok": 2 if o.decompress_point(o.to_le32(3)) is None else 0, "note": "R replaced by y = 3"})
    out["verify_compressed"] = vc

    # ---- signer row: scalar_key / public / sign (lib.rs:284-342), keys from the SEED_KEYS stream
    rs = o.SplitMix64(o.SEED_KEYS ^ 0x5167)
    sg = []
    for i in range(10):
        key = o.to_le32(rs.u256())
        m = [0, 1, Q, Q - 1, 5, 123456789012345678901234567890][i] if i < 6 else rm.u256() % Q
        R, S = o.sign(key, m)
        sg.append({"key": key.hex(), "msg": hx(m), "scalar_key": hx(o.scalar_key(key)), "pk": [hx(v) for v in o.public(key)],
                   "r_b8": [hx(R[0]), hx(R[1])], "s": hx(S), "ok": True})
    sg.append({"key": sg[0]["key"], "msg": hx(Q + 1), "scalar_key": sg[0]["scalar_key"], "pk": sg[0]["pk"],
               "r_b8": [hx(0), hx(0)], "s": hx(0), "ok": False})
    out["sign"] = sg

    # ---- Schnorr variant (lib.rs:344-385): nonce from the SEED_NONCES stream instead of thread_rng
    sch = []
    rk2 = o.SplitMix64(o.SEED_NONCES ^ 0x5C)
    for i in range(5):
        key = o.to_le32(rs.u256())
        m = 123456789012345678901234567890 if i == 0 else rm.u256() % Q
        k = rk2.u256() | (rk2.u256() << 256) | (rk2.u256() << 512) | (rk2.u256() << 768)   # 1024-bit nonce, lib.rs:347-348
        r, s_big = o.sign_schnorr_with_nonce(key, m, k)
        pkp = o.public(key)
        base = {"pk": [hx(pkp[0]), hx(pkp[1])], "r": [hx(r[0]), hx(r[1])], "msg": hx(m), "s_unreduced_bits": s_big.bit_length()}
        s_red = s_big % o.ORDER
        assert o.verify_schnorr(pkp, m, r, s_big) is True and o.verify_schnorr(pkp, m, r, s_red) is True
        sch.append(dict(base, s=hx(s_red), ok=1, note="valid (s reduced mod 8l)"))
        if i == 0:
            sch.append(dict(base, s=hx(s_red ^ 1), ok=0, note="bit flipped in s"))
            sch.append(dict(base, s=hx(s_red), msg=hx((m + 1) % Q), ok=0, note="wrong msg"))
            sch.append(dict(base, s=hx(s_red), msg=hx(Q + 1), ok=2, note="msg > Q -> Err (lib.rs:365-367)"))
            for pkx, note in (((1, 2), "pk off curve"), ((0, 1), "pk = identity")):
                sch.append(dict(base, pk=[hx(pkx[0]), hx(pkx[1])], s=hx(s_red), ok=int(o.verify_schnorr(pkx, m, r, s_red)), note=note))
            sch.append(dict(base, r=[hx(3), hx(4)], s=hx(s_red), ok=int(o.verify_schnorr(pkp, m, (3, 4), s_red)), note="r off curve"))
    out["schnorr"] = sch
    return out


def main():
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(kats(), f, indent=1)
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump(oracle_vectors(), f, indent=1)
    print("wrote reference_kats.json, oracle_vectors.json")


if __name__ == "__main__":
    main()
