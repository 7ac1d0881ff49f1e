// Fr: the BN254 scalar field (reference `type Fr`, src/lib.rs:7; third-party
// ff_ce derive) re-designed for the gfx950 VALU.
//
// Measured on MI355X (profiles/r01_ubench_valu_rates.txt): v_mad_u64_u32 costs
// ~4.5 cycles per wave-instruction -- the same as v_addc_co_u32, v_lshl_add_u64
// or v_mul_lo_u32 -- so a saturated 8x32-bit Montgomery multiplier spends more
// issue slots on carries than on multiplies.  This field therefore uses an
// UNSATURATED representation:
//
//     value = sum_{i<9} v[i] * 2^(29 i),   Montgomery radix R = 2^261
//
// Limb products are < 2^58, so a whole column (<= 9 a_i*b_j plus <= 9 m_i*n_j)
// accumulates in one 64-bit register with a single v_mad_u64_u32 per product and
// no carry instructions.  Because R / r ~ 2^7.4, values may float well above r
// between multiplications ("lazy reduction"); the contract is:
//
//   N-form  : v[0..7] < 2^29, v[8] < 2^26      (all functions return N-form)
//   fr_mul  : requires limbs < 2^30 and value(a)*value(b) < r*2^261
//             (e.g. both < 13 r); returns a value < 2 r
//   fr_add  : value adds; fr_sub(a,b): a + 4r - b (b < 4r); fr_sub8: a + 8r - b
//
// Everything is __host__ __device__ so the kernel bodies can also be executed on
// the CPU by the debug harness (tests/emul); the shipped C-ABI only ever launches
// the device code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BJJ_HD __host__ __device__ __forceinline__
#define BJJ_HD_NOINLINE static __host__ __device__ __noinline__   // internal linkage: the library has several translation units
#else
#define BJJ_HD inline
#define BJJ_HD_NOINLINE
#endif

#if defined(BJJ_DEBUG_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
#define BJJ_ASSERT(x) assert(x)
#else
#define BJJ_ASSERT(x) ((void)0)
#endif

namespace bjj {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int NL = 9;
constexpr u32 MASK29 = 0x1fffffffu;

struct Fr { u32 v[NL]; };

// r in 29-bit limbs
#define BJJ_N0 0x10000001u
#define BJJ_N1 0x1f0fac9fu
#define BJJ_N2 0x0e5c2450u
#define BJJ_N3 0x07d090f3u
#define BJJ_N4 0x1585d283u
#define BJJ_N5 0x02db40c0u
#define BJJ_N6 0x00a6e141u
#define BJJ_N7 0x0e5c2634u
#define BJJ_N8 0x0030644eu
#define BJJ_NINV29 0x0fffffffu  // -r^-1 mod 2^29

BJJ_HD u32 fr_modlimb(int i) {
  constexpr u32 N[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  return N[i];
}

// k*r written with "borrowed" limbs so that limbwise (C - b) never underflows
// for an N-form b below (k r - 2^232):  c0 = d0 + 2^29, ci = di + 2^29 - 1, c8 = d8 - 1.
BJJ_HD u32 fr_c4limb(int i) {
  constexpr u32 C[NL] = {0x4u + 0x20000000u, 0x1c3eb27eu + 0x1fffffffu, 0x19709143u + 0x1fffffffu,
                         0x1f4243cdu + 0x1fffffffu, 0x16174a0cu + 0x1fffffffu, 0x0b6d0302u + 0x1fffffffu,
                         0x029b8504u + 0x1fffffffu, 0x197098d0u + 0x1fffffffu, 0x00c19139u - 1u};
  return C[i];
}
BJJ_HD u32 fr_c8limb(int i) {
  constexpr u32 C[NL] = {0x8u + 0x20000000u, 0x187d64fcu + 0x1fffffffu, 0x12e12287u + 0x1fffffffu,
                         0x1e84879bu + 0x1fffffffu, 0x0c2e9419u + 0x1fffffffu, 0x16da0605u + 0x1fffffffu,
                         0x05370a08u + 0x1fffffffu, 0x12e131a0u + 0x1fffffffu, 0x01832273u - 1u};
  return C[i];
}

BJJ_HD Fr fr_zero() { Fr r; for (int i = 0; i < NL; i++) r.v[i] = 0; return r; }

// carry propagation to N-form (limb 8 keeps whatever is left)
BJJ_HD void fr_carry(Fr& a) {
#pragma unroll
  for (int i = 0; i < NL - 1; i++) {
    a.v[i + 1] += a.v[i] >> 29;
    a.v[i] &= MASK29;
  }
}

BJJ_HD Fr fr_add(const Fr& a, const Fr& b) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
  fr_carry(r);
  return r;
}
// a + 4r - b ; needs b < 4r - 2^232 (anything that is < 2r, or a sum of two canonical values)
BJJ_HD Fr fr_sub(const Fr& a, const Fr& b) {
  Fr r;
  BJJ_ASSERT(b.v[8] <= fr_c4limb(8));
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + fr_c4limb(i) - b.v[i];
  fr_carry(r);
  return r;
}
// a + 8r - b ; needs b < 8r - 2^232
BJJ_HD Fr fr_sub8(const Fr& a, const Fr& b) {
  Fr r;
  BJJ_ASSERT(b.v[8] <= fr_c8limb(8));
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + fr_c8limb(i) - b.v[i];
  fr_carry(r);
  return r;
}
// ---- carry-less ("lazy") forms: limbs may exceed 29 bits; only legal where the consumer's
// column bound holds (fr_mul: sum_k a_i*b_{k-i} + 9*2^58 + 2^36 < 2^64; checked in the debug
// harness).  Typical use: one operand of a multiplication.
BJJ_HD Fr fr_add_lazy(const Fr& a, const Fr& b) {  // limbs: a_i + b_i
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
BJJ_HD Fr fr_sub_lazy(const Fr& a, const Fr& b) {  // a + 4r - b, b N-form < 4r; limbs < a_i + 2^30
  Fr r;
  BJJ_ASSERT(b.v[8] <= fr_c4limb(8));
#pragma unroll
  for (int i = 0; i < NL; i++) { BJJ_ASSERT(b.v[i] <= fr_c4limb(i)); r.v[i] = a.v[i] + fr_c4limb(i) - b.v[i]; }
  return r;
}
// 8r with limbs >= 2^30 - 2 (i < 8), so that a lazy (limbs < 2^30) subtrahend never underflows
BJJ_HD u32 fr_k8limb(int i) {
  constexpr u32 C[NL] = {0x8u + 0x40000000u, 0x187d64fcu + 0x3ffffffeu, 0x12e12287u + 0x3ffffffeu,
                         0x1e84879bu + 0x3ffffffeu, 0x0c2e9419u + 0x3ffffffeu, 0x16da0605u + 0x3ffffffeu,
                         0x05370a08u + 0x3ffffffeu, 0x12e131a0u + 0x3ffffffeu, 0x01832273u - 2u};
  return C[i];
}
// a + 8r - b with b lazy (limbs <= 2^30 - 2, value < 8r - 2^233); result N-form
BJJ_HD Fr fr_sub8_of_lazy(const Fr& a, const Fr& b) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) { BJJ_ASSERT(b.v[i] <= fr_k8limb(i)); r.v[i] = a.v[i] + fr_k8limb(i) - b.v[i]; }
  fr_carry(r);
  return r;
}
BJJ_HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
BJJ_HD Fr fr_dbl(const Fr& a) { return fr_add(a, a); }

BJJ_HD Fr fr_select(bool c, const Fr& a, const Fr& b) {  // c ? a : b
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

#if defined(BJJ_DEBUG_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
inline void fr_check_mul_operands(const Fr& a, const Fr& b) {
  // every 64-bit column accumulator must hold: sum_i a_i*b_{k-i} + 9 reduction products + carry-in
  for (int k = 0; k < 2 * NL - 1; k++) {
    unsigned __int128 col = 0;
    for (int i = 0; i < NL; i++) { int j = k - i; if (j >= 0 && j < NL) col += (unsigned __int128)a.v[i] * b.v[j]; }
    col += (unsigned __int128)9 * 0x1fffffffULL * 0x1fffffffULL + ((unsigned __int128)1 << 36);
    assert((col >> 64) == 0);
  }
  // value(a)*value(b) < r * 2^261, compared in units of 2^406 using the top two limbs (+1 for the tail)
  unsigned __int128 ah = (((unsigned __int128)a.v[8] << 29) + a.v[7]) + 1;
  unsigned __int128 bh = (((unsigned __int128)b.v[8] << 29) + b.v[7]) + 1;
  // r * 2^261 / 2^406 = r / 2^145 = (r >> 144) >> 1
  const unsigned __int128 lim = (((unsigned __int128)0x30644e72e131ULL << 64) | 0xa029b85045b68181ULL) >> 1;
  assert(ah * bh <= lim);
}
#define BJJ_CHECK_MUL(a, b) fr_check_mul_operands(a, b)
#else
#define BJJ_CHECK_MUL(a, b) ((void)0)
#endif

// One multiply-accumulate step of a limb column: acc += x * y (64-bit, never overflows by
// the column bound above); hipcc selects v_mad_u64_u32.  Left to itself hipcc splits the
// columns into 17 accumulators joined by v_lshl_add_u64 (as expensive as a multiply on gfx950)
// and 34 extra VGPRs.  On the device fr_mul / fr_sqr therefore come from fr_mul_columns.inc
// (gen_fr_asm.py): ONE inline-asm statement per column part, a single accumulator chain --
// 208 instead of 233 VALU instructions and 30 instead of 58 VGPRs per multiply; +4..10 % on
// every kernel (profiles/r01_ab_column_asm_multiplier.txt).  One statement per multiply-add
// was measured first and rejected: hipcc pads every asm statement with an s_nop.
// -DBJJ_NO_ASM_COLUMNS selects this portable form on the device too (A/B, and the host).
#define BJJ_MAD(acc, x, y) acc += (u64)(x) * (y)
#define BJJ_MAD_K(acc, x, k) acc += (u64)(x) * (k)

#if defined(__HIP_DEVICE_COMPILE__) && !defined(BJJ_NO_ASM_COLUMNS)
#include "fr_mul_columns.inc"  // gen_fr_asm.py: one asm statement per limb-column part
#endif

// Montgomery product a*b*2^-261 mod r, product-scanning, one 64-bit accumulator.
BJJ_HD Fr fr_mul(const Fr& a, const Fr& b) {
  BJJ_CHECK_MUL(a, b);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BJJ_NO_ASM_COLUMNS)
  return fr_mul_columns(a, b);
#endif
  u32 m[NL];
  Fr r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) BJJ_MAD(acc, a.v[i], b.v[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) BJJ_MAD_K(acc, m[i], fr_modlimb(k - i));
    m[k] = ((u32)acc * BJJ_NINV29) & MASK29;
    BJJ_MAD_K(acc, m[k], BJJ_N0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - (NL - 1); i < NL; i++) BJJ_MAD(acc, a.v[i], b.v[k - i]);
#pragma unroll
    for (int i = k - (NL - 1); i < NL; i++) BJJ_MAD_K(acc, m[i], fr_modlimb(k - i));
    r.v[k - NL] = (u32)acc & MASK29;
    acc >>= 29;
  }
  r.v[NL - 1] = (u32)acc;
  return r;
}

// Montgomery square: cross terms once, against a pre-doubled copy.
BJJ_HD Fr fr_sqr(const Fr& a) {
  BJJ_CHECK_MUL(a, a);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BJJ_NO_ASM_COLUMNS)
  return fr_sqr_columns(a);
#endif
  u32 m[NL], a2[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) a2[i] = a.v[i] << 1;
  Fr r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) BJJ_MAD(acc, a2[i], a.v[k - i]);
    if ((k & 1) == 0) BJJ_MAD(acc, a.v[k / 2], a.v[k / 2]);
#pragma unroll
    for (int i = 0; i < k; i++) BJJ_MAD_K(acc, m[i], fr_modlimb(k - i));
    m[k] = ((u32)acc * BJJ_NINV29) & MASK29;
    BJJ_MAD_K(acc, m[k], BJJ_N0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - (NL - 1); 2 * i < k; i++) BJJ_MAD(acc, a2[i], a.v[k - i]);
    if ((k & 1) == 0) BJJ_MAD(acc, a.v[k / 2], a.v[k / 2]);
#pragma unroll
    for (int i = k - (NL - 1); i < NL; i++) BJJ_MAD_K(acc, m[i], fr_modlimb(k - i));
    r.v[k - NL] = (u32)acc & MASK29;
    acc >>= 29;
  }
  r.v[NL - 1] = (u32)acc;
  return r;
}

// value-preserving (mod r) weak reduction of an N-form value: result < r + 2^233.
// q = floor(top limb / (floor(r / 2^232) + 1)) never exceeds floor(x / r).
BJJ_HD Fr fr_reduce_weak(const Fr& x) {
  const u32 q = x.v[8] / 3171407u;  // floor(r / 2^232) = 0x30644e = 3171406
  Fr r;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    c += (int64_t)x.v[i] - (int64_t)((u64)q * fr_modlimb(i));
    r.v[i] = (i < NL - 1) ? ((u32)c & MASK29) : (u32)c;
    c >>= 29;
  }
  return r;
}

// ---- constants in Montgomery form (R = 2^261) ----------------------------
BJJ_HD Fr fr_one() {  // 2^261 mod r
  Fr r = {{0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u}};
  return r;
}
BJJ_HD Fr fr_r2() {  // 2^522 mod r  (to-Montgomery multiplier)
  Fr r = {{0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au}};
  return r;
}
BJJ_HD Fr fr_one_plain() { Fr r = fr_zero(); r.v[0] = 1; return r; }

// ---- I/O: 32-byte little-endian canonical integers -----------------------
// 8 x u32 words -> 9 x 29-bit limbs (plain value, any 256-bit integer)
BJJ_HD Fr fr_from_words(const u32 w[8]) {
  Fr r;
  r.v[0] = w[0] & MASK29;
#pragma unroll
  for (int i = 1; i < 8; i++) {
    // limb i holds bits [29i, 29i+29): spans words (29i)/32 and possibly the next
    const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
    u64 two = (u64)w[wi] | ((u64)(wi + 1 < 8 ? w[wi + 1] : 0u) << 32);
    r.v[i] = (u32)(two >> sh) & MASK29;
  }
  r.v[8] = w[7] >> 8;  // bits 232..255
  return r;
}
// N-form plain value < 2^256 -> 8 x u32 words
BJJ_HD void fr_to_words(const Fr& a, u32 w[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int bit = 32 * j, li = bit / 29, sh = bit - 29 * li;  // word j starts inside limb li at offset sh
    u64 acc = (u64)a.v[li] >> sh;
    int have = 29 - sh;
    acc |= (u64)a.v[li + 1] << have;
    have += 29;
    if (have < 32 && li + 2 < NL) acc |= (u64)a.v[li + 2] << have;
    w[j] = (u32)acc;
  }
}

// x - r if x >= r (x N-form with limbs < 2^29, top limb arbitrary)
BJJ_HD Fr fr_cond_sub_kr(const Fr& x, const u32 kr[NL]) {
  Fr d;
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u32 t = x.v[i] - kr[i] - borrow;
    borrow = t >> 31;
    d.v[i] = (i < NL - 1) ? (t & MASK29) : t;
  }
  return fr_select(borrow != 0, x, d);
}
// full reduction to [0, r) of an N-form value < 16 r
BJJ_HD Fr fr_canon(const Fr& x) {
  constexpr u32 R8[NL] = {0x8u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u};
  constexpr u32 R4[NL] = {0x4u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
  constexpr u32 R2[NL] = {0x2u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
  constexpr u32 R1[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  Fr t = fr_cond_sub_kr(x, R8);
  t = fr_cond_sub_kr(t, R4);
  t = fr_cond_sub_kr(t, R2);
  t = fr_cond_sub_kr(t, R1);
  return t;
}
BJJ_HD bool fr_is_zero_canon(const Fr& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) o |= a.v[i];
  return o == 0;
}
BJJ_HD bool fr_is_zero(const Fr& a) { return fr_is_zero_canon(fr_canon(a)); }  // a < 16 r
BJJ_HD bool fr_eq(const Fr& a, const Fr& b) { return fr_is_zero(fr_sub(a, b)); }  // a < 12 r, b < 4r

// bytes (as 8 words, any 256-bit integer) -> Montgomery form, value < 2r
BJJ_HD Fr fr_to_mont_words(const u32 w[8]) { return fr_mul(fr_from_words(w), fr_r2()); }
// Montgomery form -> canonical words
BJJ_HD void fr_from_mont_words(const Fr& a, u32 w[8]) {
  constexpr u32 R1[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  Fr t = fr_mul(a, fr_one_plain());  // in [0, r]
  t = fr_cond_sub_kr(t, R1);
  fr_to_words(t, w);
}

// a^(r-2): Fermat inversion, plain left-to-right square-and-multiply.  0 -> 0.
// Kept as the independent cross-check of fr_inv_gcd (tests/emul); the kernels use fr_inv.
BJJ_HD_NOINLINE Fr fr_inv_fermat(const Fr& a) {
  // r - 2, 32-bit words, little-endian
  const u32 E[8] = {0xefffffffu, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  Fr x = a;  // top bit (bit 253) is set
  for (int bit = 252; bit >= 0; bit--) {
    x = fr_sqr(x);
    if ((E[bit >> 5] >> (bit & 31)) & 1) x = fr_mul(x, a);
  }
  return x;
}

// ---------------------------------------------------------------------------------------
// Low-latency inversion: binary GCD with 29-bit "jumps" (after T. Pornin, "Optimized Binary
// GCD for Modular Inversion", 2020, with k - 1 = 29 inner steps per round so that the exact
// division by 2^29 is a limb shift in this representation).  18 rounds x 29 steps >= 2*254 - 1.
// Invariants: a == u*y/K, b == v*y/K (mod r) with K = 2^522 (= R^2); at the end b == 1, so
// v == R^2 / y, which for y = x*R is exactly the Montgomery form of 1/x.  0 -> 0.
// ~10^4 plain instructions instead of a chain of 381 dependent Montgomery multiplications:
// the point is LATENCY -- the workgroup-wide inversion of the affine epilogue is a serial
// section during which the other waves of the workgroup wait.
// ---------------------------------------------------------------------------------------
// (u, v) are kept as SIGNED numbers -- limbs 0..7 in [0, 2^29), limb 8 a signed 32-bit top -- and are not reduced
// between rounds: the transition matrices satisfy |f| + |g| <= 2^29 per row, so one round grows max(|u|, |v|) by at
// most r (the q*r term), 18 rounds keep them below 19 r, and a single normalisation at the end replaces the three
// conditional subtractions per operand and round of the straightforward form.  The matrix entries fit 32 bits.
BJJ_HD_NOINLINE Fr fr_inv_gcd(const Fr& x) {
  constexpr u32 R32[NL] = {0x20u, 0x01f593f0u, 0x0b848a1fu, 0x1a121e6eu, 0x10ba5067u, 0x1b681815u, 0x14dc2822u, 0x0b84c680u, 0x060c89ceu};
  constexpr u32 R16[NL] = {0x10u, 0x10fac9f8u, 0x05c2450fu, 0x1d090f37u, 0x185d2833u, 0x0db40c0au, 0x0a6e1411u, 0x05c26340u, 0x030644e7u};
  constexpr u32 R8[NL] = {0x8u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u};
  constexpr u32 R4[NL] = {0x4u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
  constexpr u32 R2c[NL] = {0x2u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
  constexpr u32 R1[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  Fr a = fr_canon(x);  // y, in [0, r)
  Fr b, u = fr_r2(), v = fr_zero();
#pragma unroll
  for (int i = 0; i < NL; i++) b.v[i] = fr_modlimb(i);
#pragma unroll 1
  for (int outer = 0; outer < 18; outer++) {
    // ---- 60-bit approximations (k = 30): the low k-1 = 29 bits (exact: one limb) and the top
    //      k+1 = 31 bits of the longer of (a, b); exact values once both fit in 60 bits.
    u32 ah = 0, al = 0, all_ = 0, bh = 0, bl = 0, bll = 0;
    int top = 0;  // index of the highest limb in which a | b is non-zero (>= 2), 0 if none
#pragma unroll
    for (int i = NL - 1; i >= 2; i--) {
      const bool hit = (top == 0) && ((a.v[i] | b.v[i]) != 0);
      ah = hit ? a.v[i] : ah; al = hit ? a.v[i - 1] : al; all_ = hit ? a.v[i - 2] : all_;
      bh = hit ? b.v[i] : bh; bl = hit ? b.v[i - 1] : bl; bll = hit ? b.v[i - 2] : bll;
      top = hit ? i : top;
    }
    const int L = (top == 0) ? 0 : 32 - __builtin_clz(ah | bh);  // significant bits of the top limb
    // Both forms are computed and one is SELECTED: no branch of this function depends on the operand (the signer's
    // constant-time option inverts a secret-derived Z product; the trip counts are fixed).
    const bool big = 29 * top + L > 60, l2 = L >= 2;
    const int sh = l2 ? L - 2 : 0;
    const u64 ha = ((u64)ah << 29) | al, hb = ((u64)bh << 29) | bl;  // L + 29 significant bits
    const u64 ta_s = ha >> sh, tb_s = hb >> sh;
    const u64 ta_l = (ha << 1) | (all_ >> 28), tb_l = (hb << 1) | (bll >> 28);
    const u64 ta = l2 ? ta_s : ta_l, tb = l2 ? tb_s : tb_l;
    const u64 xa_big = (ta << 29) | a.v[0], xb_big = (tb << 29) | b.v[0];
    const u64 xa_small = ((u64)a.v[2] << 58) | ((u64)a.v[1] << 29) | a.v[0];   // both below 2^60: exact
    const u64 xb_small = ((u64)b.v[2] << 58) | ((u64)b.v[1] << 29) | b.v[0];
    u64 xa = big ? xa_big : xa_small, xb = big ? xb_big : xb_small;
    // ---- 29 binary-GCD steps on the approximations, recording the transition matrix (|entries| <= 2^29)
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 1
    for (int i = 0; i < 29; i++) {
      const bool odd = (xa & 1) != 0;
      const bool swp = odd && (xa < xb);
      const u64 ta = swp ? xb : xa, tb = swp ? xa : xb;
      const int32_t tf0 = swp ? f1 : f0, tf1 = swp ? f0 : f1, tg0 = swp ? g1 : g0, tg1 = swp ? g0 : g1;
      xa = (odd ? ta - tb : ta) >> 1; xb = tb;
      f0 = odd ? tf0 - tf1 : tf0; g0 = odd ? tg0 - tg1 : tg0;
      f1 = tf1 * 2; g1 = tg1 * 2;
    }
    // ---- (a, b) <- (a f0 + b g0, a f1 + b g1) / 2^29   (exact), then make both non-negative
    Fr na, nb;
    int64_t ca = 0, cb = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      ca += (int64_t)a.v[i] * f0 + (int64_t)b.v[i] * g0;
      cb += (int64_t)a.v[i] * f1 + (int64_t)b.v[i] * g1;
      if (i > 0) { na.v[i - 1] = (u32)ca & MASK29; nb.v[i - 1] = (u32)cb & MASK29; }
      ca >>= 29; cb >>= 29;
    }
    na.v[NL - 1] = (u32)ca & MASK29; nb.v[NL - 1] = (u32)cb & MASK29;
    const bool nega = ca < 0, negb = cb < 0;
    {  // conditional negation (two's complement over the 9 x 29-bit limbs)
      u32 c1 = 1, c2 = 1;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const u32 ia = (~na.v[i] & MASK29) + c1, ib = (~nb.v[i] & MASK29) + c2;
        c1 = ia >> 29; c2 = ib >> 29;
        a.v[i] = nega ? (ia & MASK29) : na.v[i];
        b.v[i] = negb ? (ib & MASK29) : nb.v[i];
      }
    }
    f0 = nega ? -f0 : f0; g0 = nega ? -g0 : g0;
    f1 = negb ? -f1 : f1; g1 = negb ? -g1 : g1;
    // ---- (u, v) <- (u f0 + v g0 + qu r, u f1 + v g1 + qv r) / 2^29   (exact; signed, unreduced)
    const u32 lu = (u.v[0] * (u32)f0 + v.v[0] * (u32)g0) & MASK29;
    const u32 lv = (u.v[0] * (u32)f1 + v.v[0] * (u32)g1) & MASK29;
    const u32 qu = (lu * BJJ_NINV29) & MASK29, qv = (lv * BJJ_NINV29) & MASK29;
    Fr nu, nv;
    int64_t cu = 0, cv = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int64_t ui = (i < NL - 1) ? (int64_t)u.v[i] : (int64_t)(int32_t)u.v[i];   // limb 8 is signed
      const int64_t vi = (i < NL - 1) ? (int64_t)v.v[i] : (int64_t)(int32_t)v.v[i];
      cu += ui * f0 + vi * g0 + (int64_t)((u64)qu * fr_modlimb(i));
      cv += ui * f1 + vi * g1 + (int64_t)((u64)qv * fr_modlimb(i));
      if (i > 0) { nu.v[i - 1] = (u32)cu & MASK29; nv.v[i - 1] = (u32)cv & MASK29; }
      cu >>= 29; cv >>= 29;
    }
    nu.v[NL - 1] = (u32)(int32_t)cu; nv.v[NL - 1] = (u32)(int32_t)cv;
    BJJ_ASSERT(cu > -(1 << 28) && cu < (1 << 28) && cv > -(1 << 28) && cv < (1 << 28));
    u = nu; v = nv;
  }
  // b == 1 now (or a == b... for x == 0: v == 0): v == R^2 / y (mod r) with |v| < 19 r.  Add 32 r, then reduce.
  Fr t;
  {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { const u32 w = v.v[i] + R32[i] + c; t.v[i] = w & MASK29; c = w >> 29; }
    t.v[NL - 1] = (u32)((int32_t)v.v[NL - 1] + (int32_t)R32[NL - 1] + (int32_t)c);   // > 0, < 2^28
  }
  t = fr_cond_sub_kr(t, R32); t = fr_cond_sub_kr(t, R16); t = fr_cond_sub_kr(t, R8);
  t = fr_cond_sub_kr(t, R4); t = fr_cond_sub_kr(t, R2c); t = fr_cond_sub_kr(t, R1);
  return t;
}

// 1/x in Montgomery form (0 -> 0): the low-latency binary-GCD inversion.
BJJ_HD Fr fr_inv(const Fr& x) { return fr_inv_gcd(x); }

}  // namespace bjj
