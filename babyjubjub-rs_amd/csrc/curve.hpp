// BabyJubJub group law for the GPU path.
//
// Reference semantics being reproduced (file:line into /root/reference):
//   PointProjective::add     src/lib.rs:88-131   (ref_add below: same op order)
//   PointProjective::affine  src/lib.rs:70-85
//   Point::mul_scalar        src/lib.rs:149-164
//
// Fast path.  The reference curve  A x^2 + y^2 = 1 + D x^2 y^2  (A = 168700 a
// square, D = 168696 a non-square => complete addition law) is isomorphic, via
// x' = F x with F = sqrt(-A), to  -x'^2 + y^2 = 1 + D' x'^2 y^2,  D' = -D/A (still a
// non-square, and -1 is a square in F_r, so the law stays complete).  On that
// curve extended coordinates (X:Y:Z:T), T = XY/Z, give 7M mixed additions
// (madd-2008-hwcd-3), 8M additions and 4M+4S doublings, against the reference's
// 12M+1S unified projective add.  Canonical affine coordinates of a group element
// are unique, so for ON-CURVE inputs any correct evaluation of n*P is bit-identical
// to the reference's LSB-first double-and-add.  Off-curve inputs (the reference
// never validates points, src/lib.rs:134-138, 395-412) take the exact path:
// ref_mul_scalar replays the reference's operation sequence.
#pragma once
#include "fr.hpp"

namespace bjj {

struct Consts {
  Fr A, D;          // reference curve constants, Montgomery
  Fr F;             // sqrt(-A), Montgomery
  Fr FINV_PLAIN;    // 1/F, canonical (so that mont_mul(x_mont, FINV_PLAIN) is canonical x/F)
  Fr FINV;          // 1/F, Montgomery
  Fr L_R1, L_R2;    // plain 2^261 mod l, 2^522 mod l (Montgomery constants of the mod-l arithmetic)
  Fr DP, D2P, DPINV;  // D', 2D', 1/D' Montgomery
  Fr B8X, B8Y;      // generator, Montgomery (reference curve)
  Fr TS_G;          // generator of the 2^28-order subgroup (Tonelli-Shanks), Montgomery
  Fr HALFQ;         // plain integer (r-1)/2: the sign threshold of compress/decompress
  Fr ORDER, ORDER2, ORDER4;  // plain integers 8l, 16l, 32l in 29-bit limbs
  Fr L, L2, L4;              // plain integers l, 2l, 4l
  // Poseidon t=6 in its sparse-partial-round form (gen_tables.py: poseidon_sparse_constants)
  Fr PCF[48];       // full-round constants: rounds 0-3 then 64-67 (round 64 adjusted), Montgomery
  Fr PKP[60];       // scalar constant of each partial round
  Fr PSP[660];      // per partial round: m00, v[5], what[5]
  Fr PAL[25];       // 5x5 block applied after the last partial round
  Fr PM[36];        // MDS, row-major
  Fr PCAB[30];      // per pair of partial rounds (2p, 2p+1): sum_j v_b[j] * what_a[j]
  // Pohlig-Hellman tables of the square root (gen_tables.py): digit stripping / half-exponent factors / hash
  Fr TSN[384];
  Fr TSH[448];
  unsigned char TSHASH[2048];
};

struct Ext { Fr X, Y, Z, T; };            // a' = -1 curve, extended
struct Niels { Fr ymx, ypx, t2d; };       // affine point (x',y): y-x', y+x', 2D'x'y
struct PNiels { Fr ymx, ypx, t2d, z2; };  // projective: Y-X, Y+X, 2D'T, 2Z
struct RefProj { Fr x, y, z; };           // reference PointProjective, src/lib.rs:62-67

BJJ_HD Ext ext_identity() {
  Ext e; e.X = fr_zero(); e.Y = fr_one(); e.Z = fr_one(); e.T = fr_zero(); return e;
}

// Limb discipline inside the formulas below: products of fr_mul/fr_sqr are N-form
// (limbs < 2^29); sums/differences that feed exactly one multiplication stay carry-less
// ("lazy", limbs < 2^30 for a sum, < 1.5*2^30 for a difference) and only the one operand per
// formula that would break the 64-bit column bound is carried.  Value bounds in comments.

// P + Q, Q affine-precomputed (entries N-form, < 2r).  P coords N-form < 2r.  7M (6M when the caller
// does not need T: the last addition of a chain).
template <bool NEED_T = true>
BJJ_HD Ext ext_madd(const Ext& p, const Niels& q) {
  Fr a = fr_mul(fr_sub_lazy(p.Y, p.X), q.ymx);   // (Y-X) < 6r
  Fr b = fr_mul(fr_add_lazy(p.Y, p.X), q.ypx);   // (Y+X) < 4r
  Fr c = fr_mul(p.T, q.t2d);
  Fr d = fr_add_lazy(p.Z, p.Z);                  // 2Z < 4r, limbs < 2^30
  Fr e = fr_sub_lazy(b, a);                      // < 6r,  limbs < 1.5*2^30
  Fr f = fr_sub(d, c);                           // < 8r,  carried
  Fr g = fr_add_lazy(d, c);                      // < 6r,  limbs < 1.5*2^30
  Fr h = fr_add_lazy(b, a);                      // < 4r,  limbs < 2^30
  Ext r;
  r.X = fr_mul(e, f); r.Y = fr_mul(g, h); r.Z = fr_mul(f, g);
  if (NEED_T) r.T = fr_mul(e, h); else r.T = fr_zero();
  return r;
}
// P + Q, Q projective-precomputed (entries carried).  8M; 7M when the caller does not need T (need_t is wave-uniform in
// the kernels: a doubling follows, which never reads T).
BJJ_HD Ext ext_add_pn(const Ext& p, const PNiels& q, bool need_t = true) {
  Fr a = fr_mul(fr_sub_lazy(p.Y, p.X), q.ymx);
  Fr b = fr_mul(fr_add_lazy(p.Y, p.X), q.ypx);
  Fr c = fr_mul(p.T, q.t2d);
  Fr d = fr_mul(p.Z, q.z2);
  Fr e = fr_sub_lazy(b, a);                      // limbs < 1.5*2^30
  Fr f = fr_sub(d, c);                           // carried
  Fr g = fr_add_lazy(d, c);                      // limbs < 2^30
  Fr h = fr_add_lazy(b, a);                      // limbs < 2^30
  Ext r;
  r.X = fr_mul(e, f); r.Y = fr_mul(g, h); r.Z = fr_mul(f, g);
  r.T = fr_zero();
  if (need_t) r.T = fr_mul(e, h);
  return r;
}
// 2P (dbl-2008-hwcd with a = -1; all four outputs negated, which is the same
// projective point, so that only subtractions of small operands are needed).
template <bool NEED_T>
BJJ_HD Ext ext_dbl(const Ext& p) {
  Fr a = fr_sqr(p.X), b = fr_sqr(p.Y), zz = fr_sqr(p.Z);
  Fr h = fr_add_lazy(a, b);                      // H' = A + B           < 4r, limbs < 2^30
  Fr s = fr_sqr(fr_add_lazy(p.X, p.Y));
  Fr e = fr_sub8_of_lazy(s, h);                  // E = (X+Y)^2 - A - B  < 10r, carried
  Fr g = fr_sub_lazy(b, a);                      // G = B - A            < 6r, limbs < 1.5*2^30
  Fr f = fr_sub(fr_add_lazy(fr_add_lazy(zz, zz), a), b);  // F' = 2Z^2 - G = 2Z^2 + A - B  < 10r, carried
  Ext r;
  r.X = fr_mul(e, f); r.Y = fr_mul(g, h); r.Z = fr_mul(f, g);
  if (NEED_T) r.T = fr_mul(e, h); else r.T = fr_zero();
  return r;
}
BJJ_HD PNiels ext_to_pniels(const Ext& p, const Consts& K) {
  PNiels n;
  n.ymx = fr_sub(p.Y, p.X); n.ypx = fr_add(p.Y, p.X);
  n.t2d = fr_mul(p.T, K.D2P); n.z2 = fr_dbl(p.Z);
  return n;
}
// conditional negation: -(x,y) = (-x,y)  => swap ymx/ypx, negate t2d.  The negation stays carry-less (limbs < 2^30, value
// < 4r): the entry's t2d only ever feeds ONE multiplication whose other operand (T, or the constant 1/D') is N-form -- 9
// subtractions instead of 9 + a 24-instruction carry sweep, twice per window of the windowed loops.  Entries: t2d N-form < 4r.
BJJ_HD PNiels pniels_cneg(const PNiels& n, bool neg) {
  PNiels r;
  r.ymx = fr_select(neg, n.ypx, n.ymx);
  r.ypx = fr_select(neg, n.ymx, n.ypx);
  r.t2d = fr_select(neg, fr_sub_lazy(fr_zero(), n.t2d), n.t2d);
  r.z2 = n.z2;
  return r;
}
// affine point of the REFERENCE curve (Montgomery x, y) -> extended point on the a'=-1 curve
BJJ_HD Ext ext_from_ref_affine(const Fr& x, const Fr& y, const Consts& K) {
  Ext e;
  e.X = fr_mul(x, K.F); e.Y = y; e.Z = fr_one(); e.T = fr_mul(e.X, y);
  return e;
}
// on-curve test on the reference curve: A x^2 + y^2 == 1 + D x^2 y^2  (x, y Montgomery, < 2r)
BJJ_HD bool ref_on_curve(const Fr& x, const Fr& y, const Consts& K) {
  Fr x2 = fr_sqr(x), y2 = fr_sqr(y);
  Fr lhs = fr_add(fr_mul(K.A, x2), y2);                      // < 4r
  Fr rhs = fr_add(fr_one(), fr_mul(K.D, fr_mul(x2, y2)));    // < 3r
  return fr_eq(lhs, rhs);
}

// ---- reference-exact projective arithmetic (src/lib.rs:88-131) -------------
BJJ_HD RefProj ref_add(const RefProj& p, const RefProj& q, const Consts& K) {
  Fr a = fr_mul(p.z, q.z);
  Fr b = fr_sqr(a);
  Fr c = fr_mul(p.x, q.x);
  Fr d = fr_mul(p.y, q.y);
  Fr e = fr_mul(fr_mul(K.D, c), d);
  Fr f = fr_sub(b, e);                       // < 6r
  Fr g = fr_add(b, e);                       // < 4r
  Fr aux = fr_mul(fr_add(p.x, p.y), fr_add(q.x, q.y));
  aux = fr_sub(fr_sub(aux, c), d);           // < 10r
  Fr x3 = fr_mul(fr_mul(a, f), aux);
  Fr dac = fr_sub(d, fr_mul(K.A, c));        // < 6r
  Fr y3 = fr_mul(fr_mul(a, g), dac);
  RefProj r; r.x = x3; r.y = y3; r.z = fr_mul(f, g);
  return r;
}

// ---- scalars ----------------------------------------------------------------
// 256-bit scalar (8 words) mod 8l, returned as 8 words (< 2^254).  Exact for every
// on-curve point because the group order is 8l.
BJJ_HD void scalar_mod_order(const u32 w[8], u32 out[8], const Consts& K) {
  Fr s = fr_from_words(w);
  s = fr_cond_sub_kr(s, K.ORDER4.v);
  s = fr_cond_sub_kr(s, K.ORDER2.v);
  s = fr_cond_sub_kr(s, K.ORDER.v);
  fr_to_words(s, out);
}
// canonical field element (plain N-form, < r < 8l) mod l
BJJ_HD Fr plain_mod_l(const Fr& v, const Consts& K) {
  Fr s = fr_cond_sub_kr(v, K.L4.v);
  s = fr_cond_sub_kr(s, K.L2.v);
  s = fr_cond_sub_kr(s, K.L.v);
  return s;
}
// W-bit window j of a 256-bit little-endian integer (W <= 32); bits past 255 read as 0
BJJ_HD u32 scalar_window(const u32 w[8], int j, int W) {
  const int bit = j * W, wi = bit >> 5, sh = bit & 31;
  if (wi >= 8) return 0;
  u64 two = (u64)w[wi] | ((u64)(wi + 1 < 8 ? w[wi + 1] : 0u) << 32);
  return (u32)(two >> sh) & ((1u << W) - 1u);
}

}  // namespace bjj
