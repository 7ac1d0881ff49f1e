"""The reference crate's own unit tests for the accelerated path, re-stated against the
host-side mirror (babyjubjub_rs_amd.Point / Signature / verify) so that they read like
/root/reference/src/lib.rs:421-572, 689-738.  Every numeric step runs in libbjj_hip.so."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(gpu_ctx):
    import babyjubjub_rs_amd as bjj
    bjj.api._DEFAULT = gpu_ctx
    return bjj


P = (17777552123799933955779906779655732241715742912184938656739573121738514868268,
     2626589144620713026669568689430873010625803728049924121243784502389097019475)


def test_add_same_point(api):  # src/lib.rs:421-459
    p = api.PointProjective(*P, 1)
    q = api.PointProjective(*P, 1)
    res = p.add(q).affine()
    assert res.x == 6890855772600357754907169075114257697580319025794532037257385534741338397365
    assert res.y == 4338620300185947561074059802482547481416142213883829469920100239455078257889


def test_add_different_points(api):  # src/lib.rs:461-499
    p = api.PointProjective(*P, 1)
    q = api.PointProjective(16540640123574156134436876038791482806971768689494387082833631921987005038935,
                            20819045374670962167435360035096875258406992893633759881276124905556507972311, 1)
    res = p.add(q).affine()
    assert res.x == 7916061937171219682591368294088513039687205273691143098332585753343424131937
    assert res.y == 14035240266687799601661095864649209771790948434046947201833777492504781204499


def test_mul_scalar(api):  # src/lib.rs:502-552
    p = api.Point(*P)
    res_m = p.mul_scalar(3)
    res_a = p.projective().add(p.projective())
    res_a = res_a.add(p.projective()).affine()
    assert res_m.x == res_a.x
    assert res_m.x == 19372461775513343691590086534037741906533799473648040012278229434133483800898
    assert res_m.y == 9458658722007214007257525444427903161243386465067105737478306991484593958249
    n = 14035240266687799601661095864649209771790948434046947201833777492504781204499
    res2 = p.mul_scalar(n)
    assert res2.x == 17070357974431721403481313912716834497662307308519659060910483826664480189605
    assert res2.y == 4014745322800118607127020275658861516666525056516280575712425373174125159339
    assert p.mul_scalar(-n).equals(res2)  # sign dropped, src/lib.rs:156
    assert p.mul_scalar(0).equals(api.Point(0, 1))


def test_new_key_sign_verify(api, golden):  # src/lib.rs:555-572 (the signature is a fixture: tests/golden/make_gpu_expected.py)
    from conftest import ints
    for c in golden["gpu_expected"]["sign_with_scalars"]:
        msg, A, R, S = ints(c["msg"]), ints(c["A"]), ints(c["R"]), ints(c["S"])
        pk = api.Point(*A)
        assert pk.equals(api.Point(*api.B8).mul_scalar(ints(c["k"])))
        sig = api.Signature(api.Point(*R), S)
        assert api.verify(pk, sig, msg) is True
        assert api.verify(pk, sig, msg + 1) is False


def test_circomlib_testvector(api):  # src/lib.rs:689-738 (public key, R, S and verify == true)
    scalar_key = 6466070937662820620902051049739362987537906109895538826186780010858059362905
    pk = api.Point(*api.B8).mul_scalar(scalar_key)  # PrivateKey::public, src/lib.rs:304-306
    assert pk.x == 0x1d5ac1f31407018b7d413a4f52c8f74463b30e6ac2238220ad8b254de4eaa3a2
    assert pk.y == 0x1e1de8a908826c3f9ac2e0ceee929ecd0caf3b99b3ef24523aaab796a6f733c4
    msg = int.from_bytes(bytes.fromhex("00010203040506070809"), "little")
    sig = api.Signature(
        api.Point(0x192b4e51adf302c8139d356d0e08e2404b5ace440ef41fc78f5c4f2428df0765,
                  0x2202bebcf57b820863e0acc88970b6ca7d987a0d513c2ddeb42e3f5d31b4eddf),
        1672775540645840396591609181675628451599263765380031905495115170613215233181)
    assert api.verify(pk, sig, msg) is True
    assert api.verify(pk, api.Signature(sig.r_b8, sig.s + 1), msg) is False
    assert api.verify(pk, sig, api.Q + 1) is False  # src/lib.rs:396-398
