// babyjubjub.hpp -- header-only C++ mirror of the reference crate's API for the accelerated
// path, on top of the C ABI (include/bjj_hip.h).  The reference is Rust and no Rust toolchain
// exists in this image, so this is the compiled-language host side; names, argument meaning
// and error behaviour follow /root/reference/src/lib.rs:
//
//   Fr::from_str (decimal)                      used all over lib.rs:28-60 and the tests
//   Point { x, y }, projective(), mul_scalar(), equals()      lib.rs:134-164, 180-185
//   PointProjective { x, y, z }, add(), affine()              lib.rs:62-131
//   Signature { r_b8, s }                                     lib.rs:239-243
//   verify(pk, sig, msg) -> bool                              lib.rs:395-412
//   Point::compress, decompress_point, Signature::compress,
//   decompress_signature (wire format)                        lib.rs:166-178, 192-224, 245-268
//   PrivateKey { key }, import, scalar_key, public, sign       lib.rs:270-342
//   + *_batch forms (what the GPU is for)
//
// Every arithmetic operation runs in libbjj_hip.so on the GPU; this header only marshals
// integers to the ABI's 32-byte little-endian records.  mul_scalar/add are infallible and
// verify folds every failure into `false`, as in the reference; std::runtime_error is thrown
// only for API misuse or HIP runtime errors.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bjj_hip.h"

namespace babyjubjub_rs {

// Unsigned 256-bit integer, little-endian bytes: stands in for both `Fr` values at the
// boundary (canonical, < r) and `BigInt` scalars (`n`, `s`, `msg`).
struct U256 {
  std::array<uint8_t, 32> le{};
  U256() = default;
  explicit U256(uint64_t v) { for (int i = 0; i < 8; i++) le[i] = (uint8_t)(v >> (8 * i)); }
  // decimal string, like Fr::from_str / BigInt::parse_bytes(.., 10); throws on overflow / bad digit
  static U256 from_str(const std::string& dec) {
    if (dec.empty()) throw std::runtime_error("U256::from_str: empty string");
    U256 r;
    for (char ch : dec) {
      if (ch < '0' || ch > '9') throw std::runtime_error("U256::from_str: not a decimal digit");
      unsigned carry = (unsigned)(ch - '0');
      for (int i = 0; i < 32; i++) { unsigned v = r.le[i] * 10u + carry; r.le[i] = (uint8_t)v; carry = v >> 8; }
      if (carry) throw std::runtime_error("U256::from_str: value does not fit 256 bits");
    }
    return r;
  }
  static U256 from_hex(const std::string& hex) {  // big-endian hex, optional 0x
    size_t p = (hex.size() > 1 && hex[0] == '0' && (hex[1] == 'x' || hex[1] == 'X')) ? 2 : 0;
    U256 r; size_t nd = hex.size() - p;
    if (nd == 0 || nd > 64) throw std::runtime_error("U256::from_hex: bad length");
    for (size_t k = 0; k < nd; k++) {
      char c = hex[hex.size() - 1 - k]; unsigned v;
      if (c >= '0' && c <= '9') v = c - '0'; else if (c >= 'a' && c <= 'f') v = c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v = c - 'A' + 10; else throw std::runtime_error("U256::from_hex: bad digit");
      r.le[k / 2] |= (uint8_t)(v << (4 * (k & 1)));
    }
    return r;
  }
  std::string to_hex() const {  // 64 hex chars, big-endian, like ff's to_hex (lib.rs:169, 336, 406)
    static const char* d = "0123456789abcdef"; std::string s(64, '0');
    for (int i = 0; i < 32; i++) { s[63 - 2 * i] = d[le[i] & 15]; s[62 - 2 * i] = d[le[i] >> 4]; }
    return s;
  }
  bool operator==(const U256& o) const { return le == o.le; }
  bool operator!=(const U256& o) const { return !(le == o.le); }
};
using Fr = U256;

class Context {  // RAII over bjj_ctx (one GPU + stream + fixed-base table)
 public:
  explicit Context(int device = 0, int window_bits = 0) {
    int rc = bjj_init(device, window_bits, &h_);
    if (rc != BJJ_OK) throw std::runtime_error(std::string("bjj_init: ") + bjj_last_error());
  }
  ~Context() { bjj_free(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  bjj_ctx* handle() const { return h_; }
  // signer hardening (bjj_hip.h): no secret-dependent address or branch in public / sign / sign_schnorr; same results
  void set_signer_constant_time(bool on) {
    if (bjj_set_signer_constant_time(h_, on ? 1 : 0) != BJJ_OK) throw std::runtime_error(std::string("bjj_set_signer_constant_time: ") + bjj_last_error());
  }
  static Context& global() { static Context c; return c; }
 private:
  bjj_ctx* h_ = nullptr;
};
inline void check(int rc, const char* what) {
  if (rc != BJJ_OK) throw std::runtime_error(std::string(what) + ": " + bjj_last_error());
}

struct Point;
struct PointProjective {  // lib.rs:62-67
  Fr x, y, z;
  Point affine() const;                                   // lib.rs:70-85 (z == 0 -> (0, 0))
  PointProjective add(const PointProjective& q) const;    // lib.rs:88-131: the raw (x, y, z), any z
};
struct Point {  // lib.rs:134-138
  Fr x, y;
  PointProjective projective() const { return PointProjective{x, y, Fr(1)}; }  // lib.rs:141-147
  Point mul_scalar(const U256& n) const;                                       // lib.rs:149-164
  // `n: &BigInt` is unbounded (lib.rs:149, 156-157): little-endian magnitude of any width (padded to 32 k bytes here)
  Point mul_scalar_wide(const std::vector<uint8_t>& n_le) const;
  bool equals(const Point& p) const { return x == p.x && y == p.y; }           // lib.rs:180-185
  std::array<uint8_t, 32> compress() const;                                    // lib.rs:166-178
};
struct Signature {  // lib.rs:239-243
  Point r_b8;
  U256 s;
  std::array<uint8_t, 64> compress() const {  // lib.rs:245-258
    std::array<uint8_t, 64> b{}; auto r = r_b8.compress();
    std::memcpy(b.data(), r.data(), 32); std::memcpy(b.data() + 32, s.le.data(), 32); return b;
  }
};

struct PrivateKey {  // lib.rs:270-342
  std::array<uint8_t, 32> key{};
  static PrivateKey import(const std::vector<uint8_t>& b) {  // lib.rs:275-282
    if (b.size() != 32) throw std::invalid_argument("imported key can not be bigger than 32 bytes");
    PrivateKey k; std::memcpy(k.key.data(), b.data(), 32); return k;
  }
  U256 scalar_key() const;               // lib.rs:284-302
  Point public_key() const;              // lib.rs:304-306 (`public` is a C++ keyword)
  Signature sign(const U256& msg) const; // lib.rs:308-342; throws std::invalid_argument where the crate returns Err
  // lib.rs:344-361 with the 1024-bit nonce k (128 little-endian bytes) supplied by the caller -- the crate draws it from
  // rand::thread_rng (:347-348).  Returns (r, s) with s the crate's UNREDUCED integer k + scalar_key*h, 160 bytes LE.
  std::pair<Point, std::array<uint8_t, BJJ_SCHNORR_S_BYTES>> sign_schnorr(const U256& m,
                                                                         const std::array<uint8_t, BJJ_SCHNORR_NONCE_BYTES>& k) const;
};

inline const Point& B8() {  // lib.rs:37-46
  static const Point p{Fr::from_str("5299619240641551281634865583518297030282874472190772894086521144482721001553"),
                       Fr::from_str("16950150798460657717958625567821834550301663161624707787222815936182638968203")};
  return p;
}

// ---- page-locked storage for the batch forms ---------------------------------------------
// A std::vector whose memory comes from bjj_host_alloc: the host-pointer entry points copy such arrays straight over PCIe
// instead of staging them (include/bjj_hip.h, "pinned host memory").  Marshal records into a PinnedVector once and reuse it:
// page-locking costs ~60 us per MB.  The memory belongs to the process (any Context may use it).
template <class T>
struct PinnedAllocator {
  using value_type = T;
  PinnedAllocator() = default;
  template <class U> PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(size_t n) {
    void* p = nullptr;
    check(bjj_host_alloc(Context::global().handle(), n * sizeof(T), &p), "bjj_host_alloc");
    return (T*)p;
  }
  void deallocate(T* p, size_t) noexcept { bjj_host_free(Context::global().handle(), p); }
  template <class U> bool operator==(const PinnedAllocator<U>&) const { return true; }
  template <class U> bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
template <class T> using PinnedVector = std::vector<T, PinnedAllocator<T>>;

// ---- batch forms --------------------------------------------------------------------
// (each has a PinnedVector overload below: same call, arrays copied directly)
inline std::vector<Point> mul_fixed_base_batch(const std::vector<U256>& n, Context& c = Context::global()) {
  std::vector<Point> out(n.size());
  static_assert(sizeof(U256) == 32 && sizeof(Point) == 64, "records must be tightly packed");
  check(bjj_mul_fixed_base(c.handle(), (const uint8_t*)n.data(), n.size(), (uint8_t*)out.data()), "bjj_mul_fixed_base");
  return out;
}
inline void mul_fixed_base_batch(const PinnedVector<U256>& n, PinnedVector<Point>& out, Context& c = Context::global()) {
  out.resize(n.size());
  check(bjj_mul_fixed_base(c.handle(), (const uint8_t*)n.data(), n.size(), (uint8_t*)out.data()), "bjj_mul_fixed_base");
}
inline std::vector<Point> mul_scalar_batch(const std::vector<Point>& p, const std::vector<U256>& n,
                                           Context& c = Context::global()) {
  if (p.size() != n.size()) throw std::runtime_error("mul_scalar_batch: length mismatch");
  std::vector<Point> out(n.size());
  check(bjj_mul_var_base(c.handle(), (const uint8_t*)p.data(), (const uint8_t*)n.data(), n.size(), (uint8_t*)out.data()),
        "bjj_mul_var_base");
  return out;
}
inline std::vector<Fr> poseidon5_batch(const std::vector<std::array<Fr, 5>>& in, Context& c = Context::global()) {
  std::vector<Fr> out(in.size());
  check(bjj_poseidon5(c.handle(), (const uint8_t*)in.data(), in.size(), (uint8_t*)out.data()), "bjj_poseidon5");
  return out;
}
inline std::vector<uint8_t> verify_batch(const std::vector<Point>& pk, const std::vector<Signature>& sig,
                                         const std::vector<U256>& msg, Context& c = Context::global()) {
  size_t n = pk.size();
  if (sig.size() != n || msg.size() != n) throw std::runtime_error("verify_batch: length mismatch");
  std::vector<Point> r(n); std::vector<U256> s(n);
  for (size_t i = 0; i < n; i++) { r[i] = sig[i].r_b8; s[i] = sig[i].s; }
  std::vector<uint8_t> ok(n);
  check(bjj_eddsa_verify(c.handle(), (const uint8_t*)pk.data(), (const uint8_t*)r.data(), (const uint8_t*)s.data(),
                         (const uint8_t*)msg.data(), n, ok.data()), "bjj_eddsa_verify");
  return ok;
}
// the same on page-locked arrays (R and s of the signatures as arrays of their own: no re-packing, no staging)
inline void verify_batch(const PinnedVector<Point>& pk, const PinnedVector<Point>& r_b8, const PinnedVector<U256>& s,
                         const PinnedVector<U256>& msg, PinnedVector<uint8_t>& ok, Context& c = Context::global()) {
  const size_t n = pk.size();
  if (r_b8.size() != n || s.size() != n || msg.size() != n) throw std::runtime_error("verify_batch: length mismatch");
  ok.resize(n);
  check(bjj_eddsa_verify(c.handle(), (const uint8_t*)pk.data(), (const uint8_t*)r_b8.data(), (const uint8_t*)s.data(),
                         (const uint8_t*)msg.data(), n, ok.data()), "bjj_eddsa_verify");
}

// wire format, batch: ok[i] == 0 where decompress_point returns Err
inline std::vector<std::array<uint8_t, 32>> compress_batch(const std::vector<Point>& p, Context& c = Context::global()) {
  std::vector<std::array<uint8_t, 32>> out(p.size());
  check(bjj_compress_points(c.handle(), (const uint8_t*)p.data(), p.size(), (uint8_t*)out.data()), "bjj_compress_points");
  return out;
}
inline std::vector<Point> decompress_batch(const std::vector<std::array<uint8_t, 32>>& in, std::vector<uint8_t>& ok,
                                           Context& c = Context::global()) {
  std::vector<Point> out(in.size()); ok.assign(in.size(), 0);
  check(bjj_decompress_points(c.handle(), (const uint8_t*)in.data(), in.size(), (uint8_t*)out.data(), ok.data()),
        "bjj_decompress_points");
  return out;
}
// 1 = valid, 0 = invalid, 2 = pk or R does not decompress (the crate returns Err there)
inline std::vector<uint8_t> verify_compressed_batch(const std::vector<std::array<uint8_t, 32>>& pk,
                                                    const std::vector<std::array<uint8_t, 64>>& sig,
                                                    const std::vector<U256>& msg, Context& c = Context::global()) {
  if (sig.size() != pk.size() || msg.size() != pk.size()) throw std::runtime_error("verify_compressed_batch: length mismatch");
  std::vector<uint8_t> ok(pk.size());
  check(bjj_eddsa_verify_compressed(c.handle(), (const uint8_t*)pk.data(), (const uint8_t*)sig.data(),
                                    (const uint8_t*)msg.data(), pk.size(), ok.data()), "bjj_eddsa_verify_compressed");
  return ok;
}

// signer side, batch: ok[i] == 0 where sign returns Err (msg > Q)
inline std::vector<Point> public_keys_batch(const std::vector<PrivateKey>& k, Context& c = Context::global()) {
  std::vector<Point> out(k.size());
  check(bjj_public_keys(c.handle(), (const uint8_t*)k.data(), k.size(), (uint8_t*)out.data()), "bjj_public_keys");
  return out;
}
inline std::vector<Signature> sign_batch(const std::vector<PrivateKey>& k, const std::vector<U256>& msg,
                                         std::vector<uint8_t>& ok, Context& c = Context::global()) {
  if (k.size() != msg.size()) throw std::runtime_error("sign_batch: length mismatch");
  std::vector<Point> r(k.size()); std::vector<U256> s(k.size()); ok.assign(k.size(), 0);
  check(bjj_sign(c.handle(), (const uint8_t*)k.data(), (const uint8_t*)msg.data(), k.size(), (uint8_t*)r.data(),
                 (uint8_t*)s.data(), ok.data()), "bjj_sign");
  std::vector<Signature> out(k.size());
  for (size_t i = 0; i < k.size(); i++) out[i] = Signature{r[i], s[i]};
  return out;
}

// wire-format OUTPUT, one pass each (the compression is fused into the producing kernel): sk.public().compress() (lib.rs:304-306 +
// 166-178), B8.mul_scalar(n).compress() (149-164 + 166-178), sk.sign(msg)?.compress() (308-342 + 245-258; ok[i] == 0 and an all-zero
// record where sign returns Err)
inline std::vector<std::array<uint8_t, 32>> public_keys_compressed_batch(const std::vector<PrivateKey>& k, Context& c = Context::global()) {
  std::vector<std::array<uint8_t, 32>> out(k.size());
  check(bjj_public_keys_compressed(c.handle(), (const uint8_t*)k.data(), k.size(), (uint8_t*)out.data()), "bjj_public_keys_compressed");
  return out;
}
inline std::vector<std::array<uint8_t, 32>> mul_fixed_base_compressed_batch(const std::vector<U256>& n, Context& c = Context::global()) {
  std::vector<std::array<uint8_t, 32>> out(n.size());
  check(bjj_mul_fixed_base_compressed(c.handle(), (const uint8_t*)n.data(), n.size(), (uint8_t*)out.data()), "bjj_mul_fixed_base_compressed");
  return out;
}
inline std::vector<std::array<uint8_t, 64>> sign_compressed_batch(const std::vector<PrivateKey>& k, const std::vector<U256>& msg,
                                                                 std::vector<uint8_t>& ok, Context& c = Context::global()) {
  if (k.size() != msg.size()) throw std::runtime_error("sign_compressed_batch: length mismatch");
  std::vector<std::array<uint8_t, 64>> out(k.size()); ok.assign(k.size(), 0);
  check(bjj_sign_compressed(c.handle(), (const uint8_t*)k.data(), (const uint8_t*)msg.data(), k.size(), (uint8_t*)out.data(), ok.data()),
        "bjj_sign_compressed");
  return out;
}

// Schnorr signer, batch (lib.rs:344-361): ok[i] == 0 where sign_schnorr returns Err (msg > Q)
inline std::vector<Point> sign_schnorr_batch(const std::vector<PrivateKey>& k, const std::vector<U256>& msg,
                                             const std::vector<std::array<uint8_t, BJJ_SCHNORR_NONCE_BYTES>>& nonces,
                                             std::vector<std::array<uint8_t, BJJ_SCHNORR_S_BYTES>>& s, std::vector<uint8_t>& ok,
                                             Context& c = Context::global()) {
  if (k.size() != msg.size() || k.size() != nonces.size()) throw std::runtime_error("sign_schnorr_batch: length mismatch");
  std::vector<Point> r(k.size()); s.resize(k.size()); ok.assign(k.size(), 0);
  check(bjj_sign_schnorr(c.handle(), (const uint8_t*)k.data(), (const uint8_t*)msg.data(), (const uint8_t*)nonces.data(), k.size(),
                         (uint8_t*)r.data(), (uint8_t*)s.data(), ok.data()), "bjj_sign_schnorr");
  return r;
}

// ---- scalar (single-item) forms, same signatures as the crate ------------------------------
inline U256 PrivateKey::scalar_key() const {
  U256 out; check(bjj_scalar_keys(Context::global().handle(), key.data(), 1, out.le.data()), "bjj_scalar_keys"); return out;
}
inline Point PrivateKey::public_key() const { return public_keys_batch({*this})[0]; }
inline Signature PrivateKey::sign(const U256& msg) const {
  std::vector<uint8_t> ok; Signature s = sign_batch({*this}, {msg}, ok)[0];
  if (!ok[0]) throw std::invalid_argument("msg outside the Finite Field");
  return s;
}
inline std::pair<Point, std::array<uint8_t, BJJ_SCHNORR_S_BYTES>> PrivateKey::sign_schnorr(
    const U256& m, const std::array<uint8_t, BJJ_SCHNORR_NONCE_BYTES>& k) const {
  std::vector<std::array<uint8_t, BJJ_SCHNORR_S_BYTES>> s; std::vector<uint8_t> ok;
  Point r = sign_schnorr_batch({*this}, {m}, {k}, s, ok)[0];
  if (!ok[0]) throw std::invalid_argument("msg outside the Finite Field");
  return {r, s[0]};
}
inline std::array<uint8_t, 32> Point::compress() const { return compress_batch({*this})[0]; }
// decompress_point(bb) -> Result<Point, String>: throws std::invalid_argument for Err (lib.rs:192-224)
inline Point decompress_point(const std::array<uint8_t, 32>& bb) {
  std::vector<uint8_t> ok; Point p = decompress_batch({bb}, ok)[0];
  if (!ok[0]) throw std::invalid_argument("decompress_point: y outside the field or x^2 not a non-zero square");
  return p;
}
inline Signature decompress_signature(const std::array<uint8_t, 64>& b) {  // lib.rs:260-268
  std::array<uint8_t, 32> r; std::memcpy(r.data(), b.data(), 32);
  Signature s; s.r_b8 = decompress_point(r); std::memcpy(s.s.le.data(), b.data() + 32, 32); return s;
}
inline Point Point::mul_scalar(const U256& n) const {
  if (equals(B8())) return mul_fixed_base_batch({n})[0];
  return mul_scalar_batch({*this}, {n})[0];
}
inline Point Point::mul_scalar_wide(const std::vector<uint8_t>& n_le) const {
  size_t nbytes = ((n_le.size() + 31) / 32) * 32;
  if (nbytes == 0) nbytes = 32;
  if (nbytes > BJJ_MAX_SCALAR_BYTES) throw std::runtime_error("mul_scalar_wide: scalar wider than BJJ_MAX_SCALAR_BYTES");
  std::vector<uint8_t> sc(nbytes, 0);
  std::memcpy(sc.data(), n_le.data(), n_le.size());
  Point o;
  check(bjj_mul_var_base_wide(Context::global().handle(), (const uint8_t*)this, sc.data(), nbytes, 1, (uint8_t*)&o),
        "bjj_mul_var_base_wide");
  return o;
}
inline PointProjective PointProjective::add(const PointProjective& q) const {
  static_assert(sizeof(PointProjective) == 96, "records must be tightly packed");
  PointProjective o;
  check(bjj_proj_add(Context::global().handle(), (const uint8_t*)this, (const uint8_t*)&q, 1, (uint8_t*)&o), "bjj_proj_add");
  return o;
}
inline Point PointProjective::affine() const {
  Point o;
  check(bjj_proj_affine(Context::global().handle(), (const uint8_t*)this, 1, (uint8_t*)&o), "bjj_proj_affine");
  return o;
}
inline std::vector<PointProjective> proj_add_batch(const std::vector<PointProjective>& p, const std::vector<PointProjective>& q,
                                                   Context& c = Context::global()) {
  if (p.size() != q.size()) throw std::runtime_error("proj_add_batch: length mismatch");
  std::vector<PointProjective> out(p.size());
  check(bjj_proj_add(c.handle(), (const uint8_t*)p.data(), (const uint8_t*)q.data(), p.size(), (uint8_t*)out.data()), "bjj_proj_add");
  return out;
}

// every GPU of the node behind one handle (bjj_multi_*): contiguous ceil(n/G) blocks per device
class MultiContext {
 public:
  explicit MultiContext(const std::vector<int>& devices = {}, int window_bits = 0) {
    int rc = bjj_multi_init(devices.empty() ? nullptr : devices.data(), (int)devices.size(), window_bits, &h_);
    if (rc != BJJ_OK) throw std::runtime_error(std::string("bjj_multi_init: ") + bjj_last_error());
  }
  ~MultiContext() { bjj_multi_free(h_); }
  MultiContext(const MultiContext&) = delete;
  MultiContext& operator=(const MultiContext&) = delete;
  bjj_multi* handle() const { return h_; }
  int size() const { return bjj_multi_size(h_); }
  // pipeline depth of the device-resident form (bjj_multi_set_chunks): pieces per peer block, 1 = serial schedule
  void set_chunks(int chunks, size_t min_chunk_items = (size_t)1 << 15) {
    check(bjj_multi_set_chunks(h_, chunks, min_chunk_items), "bjj_multi_set_chunks");
  }
  std::vector<uint8_t> verify_batch(const std::vector<Point>& pk, const std::vector<Signature>& sig, const std::vector<U256>& msg) {
    size_t n = pk.size();
    if (sig.size() != n || msg.size() != n) throw std::runtime_error("verify_batch: length mismatch");
    std::vector<Point> r(n); std::vector<U256> s(n);
    for (size_t i = 0; i < n; i++) { r[i] = sig[i].r_b8; s[i] = sig[i].s; }
    std::vector<uint8_t> ok(n);
    check(bjj_eddsa_verify_multi(h_, (const uint8_t*)pk.data(), (const uint8_t*)r.data(), (const uint8_t*)s.data(),
                                 (const uint8_t*)msg.data(), n, ok.data()), "bjj_eddsa_verify_multi");
    return ok;
  }
  std::vector<Point> mul_fixed_base_batch(const std::vector<U256>& n) {
    std::vector<Point> out(n.size());
    check(bjj_mul_fixed_base_multi(h_, (const uint8_t*)n.data(), n.size(), (uint8_t*)out.data()), "bjj_mul_fixed_base_multi");
    return out;
  }
 private:
  bjj_multi* h_ = nullptr;
};
// little-endian integer of any width mod ORDER = 8l (the group order): bit-serial shift-and-subtract on the host.
// Exact for the scalar of B8.mul_scalar (lib.rs:377) because B8 lies on the curve.
inline U256 reduce_mod_order(const uint8_t* le, size_t nbytes) {
  static const U256 order = U256::from_str("21888242871839275222246405745257275088614511777268538073601725287587578984328");
  uint8_t acc[33] = {0};  // < 2 * ORDER < 2^256, plus the shifted-in bit
  for (size_t bit = nbytes * 8; bit-- > 0;) {
    unsigned carry = (le[bit >> 3] >> (bit & 7)) & 1u;
    for (int i = 0; i < 33; i++) { unsigned v = ((unsigned)acc[i] << 1) | carry; acc[i] = (uint8_t)v; carry = v >> 8; }
    bool ge = acc[32] != 0;
    if (!ge) { ge = true; for (int i = 31; i >= 0; i--) if (acc[i] != order.le[i]) { ge = acc[i] > order.le[i]; break; } }
    if (ge) { int borrow = 0; for (int i = 0; i < 33; i++) { int v = (int)acc[i] - (i < 32 ? order.le[i] : 0) - borrow; borrow = v < 0; acc[i] = (uint8_t)(v + (borrow << 8)); } }
  }
  U256 r; std::memcpy(r.le.data(), acc, 32); return r;
}
// verify_schnorr(pk, m, r, s) -> Result<bool, String> (lib.rs:375-385); s: the signer's integer, any width (little-endian);
// throws std::invalid_argument where the crate returns Err (msg > Q)
inline bool verify_schnorr(const Point& pk, const U256& m, const Point& r, const uint8_t* s_le, size_t s_bytes) {
  U256 s = reduce_mod_order(s_le, s_bytes);
  uint8_t ok = 0;
  check(bjj_schnorr_verify(Context::global().handle(), (const uint8_t*)&pk, (const uint8_t*)&r, s.le.data(), m.le.data(), 1, &ok),
        "bjj_schnorr_verify");
  if (ok == 2) throw std::invalid_argument("msg outside the Finite Field");
  return ok == 1;
}
inline bool verify(const Point& pk, const Signature& sig, const U256& msg) {  // lib.rs:395-412
  return verify_batch({pk}, {sig}, {msg})[0] != 0;
}

}  // namespace babyjubjub_rs
