// The reference's own unit tests for the accelerated path (src/lib.rs:421-552, 689-738)
// re-stated in C++ against babyjubjub.hpp -> libbjj_hip.so.  Run on a GPU box by
// tests/test_gpu_cpp_api.py.  Exit code 0 = all assertions held.
#include <cstdio>
#include "../../babyjubjub-rs_amd/csrc/babyjubjub.hpp"
using namespace babyjubjub_rs;
static int failures = 0;
#define ASSERT_EQ(a, b) do { if (!((a) == (b))) { printf("FAIL %s:%d  %s != %s\n", __FILE__, __LINE__, #a, #b); failures++; } } while (0)
#define ASSERT_TRUE(a) do { if (!(a)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #a); failures++; } } while (0)

static Point P() {
  return Point{Fr::from_str("17777552123799933955779906779655732241715742912184938656739573121738514868268"),
               Fr::from_str("2626589144620713026669568689430873010625803728049924121243784502389097019475")};
}
static void test_add_same_point() {  // lib.rs:421-459
  PointProjective p = P().projective(), q = P().projective();
  Point res = p.add(q).affine();
  ASSERT_EQ(res.x, Fr::from_str("6890855772600357754907169075114257697580319025794532037257385534741338397365"));
  ASSERT_EQ(res.y, Fr::from_str("4338620300185947561074059802482547481416142213883829469920100239455078257889"));
}
static void test_add_different_points() {  // lib.rs:461-499
  PointProjective p = P().projective();
  PointProjective q{Fr::from_str("16540640123574156134436876038791482806971768689494387082833631921987005038935"),
                    Fr::from_str("20819045374670962167435360035096875258406992893633759881276124905556507972311"), Fr(1)};
  Point res = p.add(q).affine();
  ASSERT_EQ(res.x, Fr::from_str("7916061937171219682591368294088513039687205273691143098332585753343424131937"));
  ASSERT_EQ(res.y, Fr::from_str("14035240266687799601661095864649209771790948434046947201833777492504781204499"));
}
static void test_mul_scalar() {  // lib.rs:502-552
  Point p = P();
  Point res_m = p.mul_scalar(U256(3));
  Point res_a = p.projective().add(p.projective()).add(p.projective()).affine();
  ASSERT_EQ(res_m.x, res_a.x);
  ASSERT_EQ(res_m.x, Fr::from_str("19372461775513343691590086534037741906533799473648040012278229434133483800898"));
  ASSERT_EQ(res_m.y, Fr::from_str("9458658722007214007257525444427903161243386465067105737478306991484593958249"));
  // the chained add goes through z != 1 (raw PointProjective::add); a scalar wider than 256 bits: 3 + 8l * 2^300
  ASSERT_TRUE(!(p.projective().add(p.projective()).z == Fr(1)));
  ASSERT_TRUE((PointProjective{Fr(1), Fr(2), Fr(0)}.affine().equals(Point{Fr(0), Fr(0)})));   // lib.rs:71-76
  {
    U256 order = U256::from_str("21888242871839275222246405745257275088614511777268538073601725287587578984328");
    std::vector<uint8_t> wide(96, 0);   // 8l << 296 occupies bytes 37.. ; + 3
    for (int i = 0; i < 32; i++) wide[37 + i] = order.le[i];
    wide[0] = 3;
    ASSERT_TRUE(p.mul_scalar_wide(wide).equals(res_m));
  }
  U256 n = U256::from_str("14035240266687799601661095864649209771790948434046947201833777492504781204499");
  Point res2 = p.mul_scalar(n);
  ASSERT_EQ(res2.x, Fr::from_str("17070357974431721403481313912716834497662307308519659060910483826664480189605"));
  ASSERT_EQ(res2.y, Fr::from_str("4014745322800118607127020275658861516666525056516280575712425373174125159339"));
}
static void test_circomlib_testvector() {  // lib.rs:689-738 (everything downstream of Blake-512)
  U256 scalar_key = U256::from_str("6466070937662820620902051049739362987537906109895538826186780010858059362905");
  Point pk = B8().mul_scalar(scalar_key);  // PrivateKey::public, lib.rs:304-306
  ASSERT_EQ(pk.x.to_hex(), std::string("1d5ac1f31407018b7d413a4f52c8f74463b30e6ac2238220ad8b254de4eaa3a2"));
  ASSERT_EQ(pk.y.to_hex(), std::string("1e1de8a908826c3f9ac2e0ceee929ecd0caf3b99b3ef24523aaab796a6f733c4"));
  U256 msg; for (int i = 0; i < 10; i++) msg.le[i] = (uint8_t)i;  // BigInt::from_bytes_le("00010203040506070809")
  // the signer half of the same test (lib.rs:699-735): import -> scalar_key -> public -> sign
  std::vector<uint8_t> kb(32); for (int i = 0; i < 32; i++) kb[i] = (uint8_t)(i % 10);
  PrivateKey sk = PrivateKey::import(kb);
  ASSERT_EQ(sk.scalar_key(), scalar_key);
  ASSERT_TRUE(sk.public_key().equals(pk));
  Signature made = sk.sign(msg);
  ASSERT_EQ(made.r_b8.x.to_hex(), std::string("192b4e51adf302c8139d356d0e08e2404b5ace440ef41fc78f5c4f2428df0765"));
  ASSERT_EQ(made.s, U256::from_str("1672775540645840396591609181675628451599263765380031905495115170613215233181"));
  ASSERT_TRUE(verify(pk, made, msg));
  Signature sig{Point{Fr::from_hex("192b4e51adf302c8139d356d0e08e2404b5ace440ef41fc78f5c4f2428df0765"),
                      Fr::from_hex("2202bebcf57b820863e0acc88970b6ca7d987a0d513c2ddeb42e3f5d31b4eddf")},
                U256::from_str("1672775540645840396591609181675628451599263765380031905495115170613215233181")};
  ASSERT_TRUE(verify(pk, sig, msg));
  Signature bad = sig; bad.s.le[0] ^= 1;
  ASSERT_TRUE(!verify(pk, bad, msg));
  U256 big = U256::from_str("21888242871839275222246405745257275088548364400416034343698204186575808495618");  // Q + 1
  ASSERT_TRUE(!verify(pk, sig, big));  // lib.rs:396-398
}
static void test_compressed_outputs() {  // the circomlib vector (lib.rs:689-738) through the fused wire-format outputs
  std::vector<uint8_t> kb(32); for (int i = 0; i < 32; i++) kb[i] = (uint8_t)(i % 10);
  PrivateKey sk = PrivateKey::import(kb);
  U256 msg; for (int i = 0; i < 10; i++) msg.le[i] = (uint8_t)i;
  std::vector<PrivateKey> ks(300, sk); std::vector<U256> ms(300, msg);
  ms[7] = U256::from_str("21888242871839275222246405745257275088548364400416034343698204186575808495618");  // Q + 1: Err
  auto pkc = public_keys_compressed_batch(ks);
  ASSERT_TRUE(pkc[0] == sk.public_key().compress() && pkc[299] == pkc[0]);
  std::vector<uint8_t> ok;
  auto sigc = sign_compressed_batch(ks, ms, ok);
  Signature made = sk.sign(msg);
  ASSERT_TRUE(ok[0] == 1 && sigc[0] == made.compress() && sigc[299] == sigc[0]);
  std::array<uint8_t, 64> zero{}; ASSERT_TRUE(ok[7] == 0 && sigc[7] == zero);
  auto back = decompress_signature(sigc[0]);                   // lib.rs:260-268
  ASSERT_TRUE(back.r_b8.equals(made.r_b8)); ASSERT_EQ(back.s, made.s);
  ASSERT_TRUE(verify_compressed_batch({pkc[0]}, {sigc[0]}, {msg})[0] == 1);
  std::vector<U256> n(100, sk.scalar_key());
  ASSERT_TRUE(mul_fixed_base_compressed_batch(n)[99] == pkc[0]);     // public() == B8.mul_scalar(scalar_key), lib.rs:304-306
}
static void test_point_compress_decompress() {  // lib.rs:575-594
  Point p = P();
  auto c = p.compress();
  U256 cc; std::memcpy(cc.le.data(), c.data(), 32);
  // hex::encode(p_comp) is byte order; U256::to_hex prints big-endian, so compare bytes
  const char* want = "53b81ed5bffe9545b54016234682e7b2f699bd42a5e9eae27ff4051bc698ce85";
  char got[65]; for (int i = 0; i < 32; i++) snprintf(got + 2 * i, 3, "%02x", c[i]);
  ASSERT_EQ(std::string(got), std::string(want));
  Point p2 = decompress_point(c);
  ASSERT_EQ(p.x, p2.x); ASSERT_EQ(p.y, p2.y);
  Signature sig{p, U256(12345)};
  Signature sig2 = decompress_signature(sig.compress());   // lib.rs:657-675 shape
  ASSERT_TRUE(sig2.r_b8.equals(p)); ASSERT_EQ(sig2.s, sig.s);
  bool threw = false;
  try { std::array<uint8_t, 32> bad; bad.fill(0xff); decompress_point(bad); } catch (const std::invalid_argument&) { threw = true; }
  ASSERT_TRUE(threw);
}
static void test_schnorr_signature() {  // lib.rs:678-686: sign_schnorr -> verify_schnorr == true
  std::vector<uint8_t> kb(32); for (int i = 0; i < 32; i++) kb[i] = (uint8_t)(7 * i + 1);
  PrivateKey sk = PrivateKey::import(kb);
  Point pk = sk.public_key();
  U256 msg = U256::from_str("123456789012345678901234567890");
  std::array<uint8_t, BJJ_SCHNORR_NONCE_BYTES> k; for (size_t i = 0; i < k.size(); i++) k[i] = (uint8_t)(251 * i + 17);
  auto rs = sk.sign_schnorr(msg, k);
  ASSERT_TRUE(verify_schnorr(pk, msg, rs.first, rs.second.data(), rs.second.size()));
  auto tampered = rs.second; tampered[0] ^= 1;
  ASSERT_TRUE(!verify_schnorr(pk, msg, rs.first, tampered.data(), tampered.size()));
  ASSERT_TRUE(rs.first.equals(B8().mul_scalar(reduce_mod_order(k.data(), k.size()))));  // r = k*B8, lib.rs:351
  bool threw = false;
  U256 big = U256::from_str("21888242871839275222246405745257275088548364400416034343698204186575808495618");  // Q + 1
  try { sk.sign_schnorr(big, k); } catch (const std::invalid_argument&) { threw = true; }
  ASSERT_TRUE(threw);
}
static void test_batch() {
  std::vector<U256> n; for (uint64_t i = 0; i < 1000; i++) n.push_back(U256(i * 0x9E3779B97F4A7C15ULL + 1));
  std::vector<Point> a = mul_fixed_base_batch(n);
  std::vector<Point> b = mul_scalar_batch(std::vector<Point>(n.size(), B8()), n);
  for (size_t i = 0; i < n.size(); i++) ASSERT_TRUE(a[i].equals(b[i]));
  ASSERT_TRUE(a[0].equals(B8()));  // n = 1
}
static void test_multi() {  // the single-process multi-GPU handle (bjj_multi_*) from a compiled-language host: two contexts on device 0
  MultiContext m({0, 0}, 12);
  ASSERT_EQ(m.size(), 2);
  std::vector<uint8_t> kb(32); for (int i = 0; i < 32; i++) kb[i] = (uint8_t)(3 * i + 5);
  PrivateKey sk = PrivateKey::import(kb);
  Point pk = sk.public_key();
  std::vector<Point> pks; std::vector<Signature> sigs; std::vector<U256> msgs;
  for (uint64_t i = 0; i < 101; i++) {   // odd count: ragged blocks of 51 and 50
    U256 msg(1000 + i);
    Signature sg = sk.sign(msg);
    if (i % 7 == 3) sg.s.le[1] ^= 4;    // corrupt some
    pks.push_back(pk); sigs.push_back(sg); msgs.push_back(msg);
  }
  std::vector<uint8_t> got = m.verify_batch(pks, sigs, msgs), want = verify_batch(pks, sigs, msgs);
  ASSERT_TRUE(got == want);
  for (size_t i = 0; i < got.size(); i++) ASSERT_EQ((int)got[i], (i % 7 == 3) ? 0 : 1);
  std::vector<U256> n; for (uint64_t i = 0; i < 77; i++) n.push_back(U256(i * 0x9E3779B97F4A7C15ULL + 3));
  std::vector<Point> a = m.mul_fixed_base_batch(n), b = mul_fixed_base_batch(n);
  for (size_t i = 0; i < n.size(); i++) ASSERT_TRUE(a[i].equals(b[i]));
}
static void test_batch_pinned() {  // the batch forms on page-locked vectors: copied directly, same bytes as the staged path
  const size_t n = 70001;
  PinnedVector<U256> sc(n);
  for (size_t i = 0; i < n; i++) { U256 v(i * 2654435761ull + 17); v.le[20] = (uint8_t)(i >> 3); v.le[31] = (uint8_t)(i & 0x1f); sc[i] = v; }
  PinnedVector<Point> out;
  mul_fixed_base_batch(sc, out);
  bjj_info info; info.struct_size = sizeof(info);
  ASSERT_EQ(bjj_get_info(Context::global().handle(), &info), 0);
  ASSERT_EQ(info.last_host_direct_arrays, 2u);
  ASSERT_EQ(info.last_host_staged_arrays, 0u);
  std::vector<U256> sc2(sc.begin(), sc.end());
  std::vector<Point> want = mul_fixed_base_batch(sc2);
  ASSERT_EQ(bjj_get_info(Context::global().handle(), &info), 0);
  ASSERT_EQ(info.last_host_staged_arrays, 2u);
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) if (!(out[i].x == want[i].x) || !(out[i].y == want[i].y)) bad++;
  ASSERT_EQ(bad, (size_t)0);
  // verify on pinned arrays: the points above as public keys AND as R, the scalars as s and msg -- verdicts only have to agree
  PinnedVector<U256> msg(sc.begin(), sc.end());
  PinnedVector<uint8_t> ok;
  verify_batch(out, out, sc, msg, ok);
  ASSERT_EQ(bjj_get_info(Context::global().handle(), &info), 0);
  ASSERT_EQ(info.last_host_direct_arrays, 5u);
  std::vector<Signature> sigs(n);
  for (size_t i = 0; i < n; i++) { sigs[i].r_b8 = want[i]; sigs[i].s = sc2[i]; }
  std::vector<uint8_t> ok2 = verify_batch(want, sigs, sc2);
  bad = 0;
  for (size_t i = 0; i < n; i++) if (ok[i] != ok2[i]) bad++;
  ASSERT_EQ(bad, (size_t)0);
}

int main() {
  try {
    test_add_same_point(); test_add_different_points(); test_mul_scalar(); test_circomlib_testvector(); test_point_compress_decompress(); test_compressed_outputs(); test_schnorr_signature(); test_batch(); test_batch_pinned(); test_multi();
  } catch (const std::exception& e) { printf("EXCEPTION %s\n", e.what()); return 2; }
  printf(failures ? "FAILED %d\n" : "ok (reference tests re-stated in C++)\n", failures);
  return failures ? 1 : 0;
}
