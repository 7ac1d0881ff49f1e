/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of the reference
 * algorithm of arnaucube/babyjubjub-rs v0.0.11 for the hot path this repo
 * accelerates.  Not shipped, not linked into libbjj_hip.so.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * What it follows (file:line into /root/reference):
 *   constants              src/lib.rs:28-60
 *   PointProjective::affine src/lib.rs:70-85
 *   PointProjective::add   src/lib.rs:88-131   (add-2008-bbjlp, same op order)
 *   Point::mul_scalar      src/lib.rs:149-164  (LSB-first, bit-serial, unified
 *                                               add also used for doubling)
 *   test_bit               src/lib.rs:188-190
 *   verify                 src/lib.rs:395-412
 *   schnorr_hash / verify_schnorr   src/lib.rs:364-385
 *   blh / scalar_key / public / sign   src/lib.rs:226-237, 284-342 (Blake-512: third-party
 *                          blake-hash 0.4.0, Cargo.toml:17; restated from the BLAKE specification and
 *                          pinned by the digest KAT of src/lib.rs:695-696)
 *   compress / decompress_point / decompress_signature
 *                          src/lib.rs:166-178, 192-224, 245-268 with utils.rs:11-29 (modinv),
 *                          109-160 (Tonelli-Shanks modsqrt), 215-223 (legendre_symbol)
 *   Fr                     third-party ff_ce 0.11 derive (Cargo.toml:12): 4 x u64
 *                          Montgomery limbs, R = 2^256; restated from the
 *                          published algorithm
 *   Poseidon::hash         third-party poseidon-rs 0.0.8 (Cargo.toml:20);
 *                          restated from the published algorithm (SURVEY.md
 *                          Appendix B), constants from gen_oracle_constants.py
 *
 * Parity status: PINNED.  tests/test_oracle_kat.py checks this library against
 * every known-answer vector the reference's tests hold for the path
 * (src/lib.rs:421-552, 689-738) and against oracle/bjj_oracle.py.
 *
 * All values cross this file's C boundary as 32-byte little-endian canonical
 * integers (exactly Fr::into_repr().0 as [u64;4]).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include "bjj_ref_constants.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr_t;            /* Montgomery form, value < r */
typedef struct { fr_t x, y, z; } proj_t;           /* lib.rs:62-67 */
typedef struct { fr_t x, y; } point_t;             /* lib.rs:134-138 */

/* r = Q, lib.rs:33-36 */
static const fr_t MODULUS = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL,
                              0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
#define INV64 0xc2e1f593efffffffULL /* -r^-1 mod 2^64 */
static const fr_t R1 = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL,
                         0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}}; /* 2^256 mod r */
static const fr_t R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL,
                         0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}; /* 2^512 mod r */
static const fr_t ZERO = {{0, 0, 0, 0}};

/* ---- Fr (ff_ce-style) -------------------------------------------------- */
static int fr_geq(const fr_t *a, const fr_t *b) {
  for (int i = 3; i >= 0; i--) {
    if (a->l[i] > b->l[i]) return 1;
    if (a->l[i] < b->l[i]) return 0;
  }
  return 1;
}
static int fr_is_zero(const fr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a, b, sizeof(fr_t)) == 0; }
static uint64_t add_nc(fr_t *a, const fr_t *b) { /* a += b, returns carry */
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; a->l[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t sub_nb(fr_t *a, const fr_t *b) { /* a -= b, returns borrow */
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - br;
    a->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}
static void fr_add(fr_t *a, const fr_t *b) { /* add_assign */
  add_nc(a, b);
  if (fr_geq(a, &MODULUS)) sub_nb(a, &MODULUS);
}
static void fr_sub(fr_t *a, const fr_t *b) { /* sub_assign */
  if (!fr_geq(a, b)) add_nc(a, &MODULUS);
  sub_nb(a, b);
}
static void fr_mul(fr_t *a, const fr_t *b) { /* mul_assign: schoolbook + Montgomery reduce */
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->l[i] * b->l[j] + t[i + j];
      t[i + j] = (uint64_t)c; c >>= 64;
    }
    t[i + 4] = (uint64_t)c;
  }
  uint64_t carry2 = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t m = t[i] * INV64;
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)m * MODULUS.l[j] + t[i + j];
      t[i + j] = (uint64_t)c; c >>= 64;
    }
    c += (u128)t[i + 4] + carry2;
    t[i + 4] = (uint64_t)c; carry2 = (uint64_t)(c >> 64);
  }
  fr_t r = {{t[4], t[5], t[6], t[7]}};
  if (carry2 || fr_geq(&r, &MODULUS)) sub_nb(&r, &MODULUS);
  *a = r;
}
static void fr_square(fr_t *a) { fr_t b = *a; fr_mul(a, &b); }
static void div2(fr_t *a) {
  for (int i = 0; i < 3; i++) a->l[i] = (a->l[i] >> 1) | (a->l[i + 1] << 63);
  a->l[3] >>= 1;
}
/* Fr::inverse: binary extended Euclid as published in ff_ce's derive output;
 * works directly on the Montgomery representation (b starts at R2). Returns 0
 * for zero input ("None"). */
static int fr_inverse(fr_t *out, const fr_t *a) {
  if (fr_is_zero(a)) return 0;
  static const fr_t ONE_REPR = {{1, 0, 0, 0}};
  fr_t u = *a, v = MODULUS, b = R2, c = ZERO;
  while (!fr_eq(&u, &ONE_REPR) && !fr_eq(&v, &ONE_REPR)) {
    while ((u.l[0] & 1) == 0) {
      div2(&u);
      if ((b.l[0] & 1) == 0) div2(&b);
      else { uint64_t cy = add_nc(&b, &MODULUS); div2(&b); b.l[3] |= cy << 63; }
    }
    while ((v.l[0] & 1) == 0) {
      div2(&v);
      if ((c.l[0] & 1) == 0) div2(&c);
      else { uint64_t cy = add_nc(&c, &MODULUS); div2(&c); c.l[3] |= cy << 63; }
    }
    if (!fr_geq(&v, &u)) { sub_nb(&u, &v); fr_sub(&b, &c); }
    else { sub_nb(&v, &u); fr_sub(&c, &b); }
  }
  *out = fr_eq(&u, &ONE_REPR) ? b : c;
  return 1;
}
static void fr_from_le(fr_t *o, const uint8_t *b) { /* reduces mod r like from_str's wrap */
  fr_t t;
  memcpy(t.l, b, 32);
  /* value < 2^256 < 6r: bring below r, then to Montgomery form */
  while (fr_geq(&t, &MODULUS)) sub_nb(&t, &MODULUS);
  fr_mul(&t, &R2);
  *o = t;
}
static void fr_to_le(uint8_t *b, const fr_t *a) { /* into_repr */
  fr_t t = *a;
  static const fr_t ONE_REPR = {{1, 0, 0, 0}};
  fr_mul(&t, &ONE_REPR);
  memcpy(b, t.l, 32);
}
static void fr_from_u64(fr_t *o, uint64_t v) {
  fr_t t = {{v, 0, 0, 0}};
  fr_mul(&t, &R2);
  *o = t;
}

/* ---- lazily initialised constants (lib.rs:28-60) ----------------------- */
static fr_t C_A, C_D;
static point_t C_B8;
static fr_t PC[408], PM[36];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_consts(void) {
  fr_from_u64(&C_A, 168700); /* lib.rs:30 */
  fr_from_u64(&C_D, 168696); /* lib.rs:28 */
  static const uint64_t b8x[4] = {0x2893f3f6bb957051ULL, 0x2ab8d8010534e0b6ULL,
                                  0x4eacb2e09d6277c1ULL, 0x0bb77a6ad63e739bULL};
  static const uint64_t b8y[4] = {0x4b3c257a872d7d8bULL, 0xfce0051fb9e13377ULL,
                                  0x25572e1cd16bf9edULL, 0x25797203f7a0b249ULL};
  fr_from_le(&C_B8.x, (const uint8_t *)b8x); /* lib.rs:37-46 */
  fr_from_le(&C_B8.y, (const uint8_t *)b8y);
  for (int i = 0; i < 408; i++) fr_from_le(&PC[i], (const uint8_t *)BJJREF_POSEIDON_C[i]);
  for (int i = 0; i < 36; i++) fr_from_le(&PM[i], (const uint8_t *)BJJREF_POSEIDON_M[i]);
}
static void ensure_init(void) { pthread_once(&g_once, init_consts); }

/* ---- PointProjective ---------------------------------------------------- */
static void proj_add(proj_t *o, const proj_t *p, const proj_t *q) { /* lib.rs:88-131 */
  fr_t a = p->z;  fr_mul(&a, &q->z);
  fr_t b = a;     fr_square(&b);
  fr_t c = p->x;  fr_mul(&c, &q->x);
  fr_t d = p->y;  fr_mul(&d, &q->y);
  fr_t e = C_D;   fr_mul(&e, &c); fr_mul(&e, &d);
  fr_t f = b;     fr_sub(&f, &e);
  fr_t g = b;     fr_add(&g, &e);
  fr_t x1y1 = p->x; fr_add(&x1y1, &p->y);
  fr_t x2y2 = q->x; fr_add(&x2y2, &q->y);
  fr_t aux = x1y1; fr_mul(&aux, &x2y2); fr_sub(&aux, &c); fr_sub(&aux, &d);
  fr_t x3 = a;    fr_mul(&x3, &f); fr_mul(&x3, &aux);
  fr_t ac = C_A;  fr_mul(&ac, &c);
  fr_t dac = d;   fr_sub(&dac, &ac);
  fr_t y3 = a;    fr_mul(&y3, &g); fr_mul(&y3, &dac);
  fr_t z3 = f;    fr_mul(&z3, &g);
  o->x = x3; o->y = y3; o->z = z3;
}
static void proj_affine(point_t *o, const proj_t *p) { /* lib.rs:70-85 */
  if (fr_is_zero(&p->z)) { o->x = ZERO; o->y = ZERO; return; }
  fr_t zinv; fr_inverse(&zinv, &p->z);
  o->x = p->x; fr_mul(&o->x, &zinv);
  o->y = p->y; fr_mul(&o->y, &zinv);
}
static int test_bit(const uint8_t *b, size_t i) { return (b[i / 8] & (1 << (i % 8))) != 0; } /* lib.rs:188-190 */
static size_t bit_length(const uint8_t *b, size_t nbytes) {
  for (size_t i = nbytes; i-- > 0;)
    if (b[i]) { size_t n = i * 8; uint8_t v = b[i]; while (v) { n++; v >>= 1; } return n; }
  return 0;
}
static void mul_scalar(point_t *o, const point_t *p, const uint8_t *n, size_t nbytes) { /* lib.rs:149-164 */
  proj_t r = {ZERO, R1, R1};
  proj_t exp = {p->x, p->y, R1};
  size_t bits = bit_length(n, nbytes);
  for (size_t i = 0; i < bits; i++) {
    if (test_bit(n, i)) { proj_t t; proj_add(&t, &r, &exp); r = t; }
    proj_t t2; proj_add(&t2, &exp, &exp); exp = t2;
  }
  proj_affine(o, &r);
}

/* ---- Poseidon t=6 ------------------------------------------------------- */
static void sbox5(fr_t *x) { fr_t x2 = *x; fr_square(&x2); fr_t x4 = x2; fr_square(&x4); fr_mul(x, &x4); }
static void poseidon5(fr_t *out, const fr_t in[5]) {
  fr_t st[6];
  st[0] = ZERO;
  for (int i = 0; i < 5; i++) st[i + 1] = in[i];
  for (int r = 0; r < 68; r++) {
    for (int j = 0; j < 6; j++) fr_add(&st[j], &PC[r * 6 + j]);
    if (r < 4 || r >= 64) { for (int j = 0; j < 6; j++) sbox5(&st[j]); }
    else sbox5(&st[0]);
    fr_t nw[6];
    for (int i = 0; i < 6; i++) {
      nw[i] = ZERO;
      for (int j = 0; j < 6; j++) { fr_t t = PM[i * 6 + j]; fr_mul(&t, &st[j]); fr_add(&nw[i], &t); }
    }
    memcpy(st, nw, sizeof(st));
  }
  *out = st[0];
}

/* ---- verify (lib.rs:395-412) ------------------------------------------- */
static int verify1(const uint8_t *pk, const uint8_t *rb8, const uint8_t *s, const uint8_t *msg) {
  fr_t m; memcpy(m.l, msg, 32);
  /* msg > Q -> false (lib.rs:396-398); msg == Q passes and wraps to 0 (lib.rs:399) */
  if (fr_geq(&m, &MODULUS) && !fr_eq(&m, &MODULUS)) return 0;
  point_t A_, R_;
  fr_from_le(&A_.x, pk); fr_from_le(&A_.y, pk + 32);
  fr_from_le(&R_.x, rb8); fr_from_le(&R_.y, rb8 + 32);
  fr_t in[5] = {R_.x, R_.y, A_.x, A_.y, ZERO};
  fr_from_le(&in[4], msg);
  fr_t hm; poseidon5(&hm, in);                       /* lib.rs:400-404 */
  point_t l; mul_scalar(&l, &C_B8, s, 32);           /* lib.rs:405 */
  uint8_t hm8[33]; uint8_t hmb[32];
  fr_to_le(hmb, &hm);                                /* lib.rs:406 */
  unsigned carry = 0;                                /* 8 * hm_b, lib.rs:410 */
  for (int i = 0; i < 32; i++) { unsigned v = ((unsigned)hmb[i] << 3) | carry; hm8[i] = (uint8_t)v; carry = v >> 8; }
  hm8[32] = (uint8_t)carry;
  point_t t; mul_scalar(&t, &A_, hm8, 33);
  proj_t rp = {R_.x, R_.y, R1}, tp = {t.x, t.y, R1}, sum;
  proj_add(&sum, &rp, &tp);                          /* lib.rs:407-410 */
  point_t ra; proj_affine(&ra, &sum);                /* lib.rs:411 */
  return fr_eq(&l.x, &ra.x) && fr_eq(&l.y, &ra.y);   /* lib.rs:180-185 */
}


/* ---- Blake-512 (original BLAKE; lib.rs:226-237 via blake-hash) ---------- */
static const uint64_t BLAKE_IV[8] = {0x6A09E667F3BCC908ULL, 0xBB67AE8584CAA73BULL, 0x3C6EF372FE94F82BULL, 0xA54FF53A5F1D36F1ULL,
                                     0x510E527FADE682D1ULL, 0x9B05688C2B3E6C1FULL, 0x1F83D9ABFB41BD6BULL, 0x5BE0CD19137E2179ULL};
static const uint64_t BLAKE_C[16] = {0x243F6A8885A308D3ULL, 0x13198A2E03707344ULL, 0xA4093822299F31D0ULL, 0x082EFA98EC4E6C89ULL,
                                     0x452821E638D01377ULL, 0xBE5466CF34E90C6CULL, 0xC0AC29B7C97C50DDULL, 0x3F84D5B5B5470917ULL,
                                     0x9216D5D98979FB1BULL, 0xD1310BA698DFB5ACULL, 0x2FFD72DBD01ADFB7ULL, 0xB8E1AFED6A267E96ULL,
                                     0xBA7C9045F12C7F99ULL, 0x24A19947B3916CF7ULL, 0x0801F2E2858EFC16ULL, 0x636920D871574E69ULL};
static const uint8_t BLAKE_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static void blake_compress(uint64_t h[8], const uint8_t blk[128], uint64_t t) {
  uint64_t m[16], v[16];
  for (int i = 0; i < 16; i++) { m[i] = 0; for (int b = 0; b < 8; b++) m[i] = (m[i] << 8) | blk[8 * i + b]; }
  for (int i = 0; i < 8; i++) v[i] = h[i];
  for (int i = 0; i < 4; i++) v[8 + i] = BLAKE_C[i];
  v[12] = BLAKE_C[4] ^ t; v[13] = BLAKE_C[5] ^ t; v[14] = BLAKE_C[6]; v[15] = BLAKE_C[7];
#define BG(a, b, c, d, i)                                                   \
  do { int j = BLAKE_SIGMA[r % 10][2 * (i)], k = BLAKE_SIGMA[r % 10][2 * (i) + 1]; \
    v[a] += v[b] + (m[j] ^ BLAKE_C[k]); v[d] = ror64(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = ror64(v[b] ^ v[c], 25); \
    v[a] += v[b] + (m[k] ^ BLAKE_C[j]); v[d] = ror64(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror64(v[b] ^ v[c], 11); } while (0)
  for (int r = 0; r < 16; r++) {
    BG(0, 4, 8, 12, 0); BG(1, 5, 9, 13, 1); BG(2, 6, 10, 14, 2); BG(3, 7, 11, 15, 3);
    BG(0, 5, 10, 15, 4); BG(1, 6, 11, 12, 5); BG(2, 7, 8, 13, 6); BG(3, 4, 9, 14, 7);
  }
#undef BG
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
static void blake512(const uint8_t *msg, size_t len, uint8_t out[64]) { /* len < 2^61 */
  uint64_t h[8]; memcpy(h, BLAKE_IV, sizeof(h));
  uint64_t nbits = (uint64_t)len * 8, done = 0;
  uint8_t blk[128];
  while (len - done >= 128) { blake_compress(h, msg + done, (done + 128) * 8); done += 128; }
  size_t rem = len - done;
  memset(blk, 0, 128); memcpy(blk, msg + done, rem);
  if (rem <= 111) {
    blk[rem] |= 0x80; blk[111] |= 0x01;
    for (int i = 0; i < 8; i++) blk[127 - i] = (uint8_t)(nbits >> (8 * i));
    blake_compress(h, blk, rem ? nbits : 0);
  } else {
    blk[rem] |= 0x80;
    blake_compress(h, blk, nbits);
    memset(blk, 0, 128); blk[111] = 0x01;
    for (int i = 0; i < 8; i++) blk[127 - i] = (uint8_t)(nbits >> (8 * i));
    blake_compress(h, blk, 0);
  }
  for (int i = 0; i < 8; i++) for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(h[i] >> (56 - 8 * b));
}

/* ---- PrivateKey::scalar_key / public / sign (lib.rs:284-342) ------------ */
static const uint64_t SUBORDER_L[4] = {0x677297dc392126f1ULL, 0xab3eedb83920ee0aULL, 0x370a08b6d0302b0bULL, 0x060c89ce5c263405ULL};
/* x (nw 64-bit words, LE) mod l by binary long division; out 4 words */
static void mod_l(const uint64_t *x, int nw, uint64_t out[4]) {
  uint64_t r[5] = {0, 0, 0, 0, 0};
  for (int bit = nw * 64 - 1; bit >= 0; bit--) {
    for (int i = 4; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
    r[0] = (r[0] << 1) | ((x[bit / 64] >> (bit % 64)) & 1);
    /* r >= l ? */
    int ge = r[4] != 0;
    if (!ge) { ge = 1; for (int i = 3; i >= 0; i--) { if (r[i] > SUBORDER_L[i]) break; if (r[i] < SUBORDER_L[i]) { ge = 0; break; } } }
    if (ge) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r[i] - SUBORDER_L[i] - br; r[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } r[4] -= br; }
  }
  memcpy(out, r, 32);
}
static void scalar_key_pruned(const uint8_t key[32], uint8_t h64[64], uint8_t pruned[32]) {
  blake512(key, 32, h64);                         /* lib.rs:292 */
  memcpy(pruned, h64, 32);
  pruned[0] &= 0xF8; pruned[31] &= 0x7F; pruned[31] |= 0x40;   /* lib.rs:297-299 */
}
static void shr3(const uint8_t in[32], uint8_t out[32]) {         /* sk >> 3, lib.rs:301-302 */
  for (int i = 0; i < 32; i++) out[i] = (uint8_t)((in[i] >> 3) | (i < 31 ? (in[i + 1] << 5) : 0));
}
/* returns 1 = Ok, 0 = Err ("msg outside the Finite Field", lib.rs:309-311) */
static int sign1(const uint8_t key[32], const uint8_t msg[32], uint8_t out_r[64], uint8_t out_s[32]) {
  fr_t m; memcpy(m.l, msg, 32);
  if (fr_geq(&m, &MODULUS) && !fr_eq(&m, &MODULUS)) return 0;
  uint8_t h[64], pruned[32], sk[32];
  scalar_key_pruned(key, h, pruned);              /* lib.rs:316 */
  uint8_t rb[64]; memcpy(rb, h + 32, 32); memcpy(rb + 32, msg, 32);   /* lib.rs:318-325 */
  uint8_t rh[64]; blake512(rb, 64, rh);           /* lib.rs:326 */
  uint64_t rw[8], r[4]; memcpy(rw, rh, 64);       /* from_bytes_le */
  mod_l(rw, 8, r);                                /* lib.rs:328 */
  point_t rp, ap;
  mul_scalar(&rp, &C_B8, (const uint8_t *)r, 32); /* lib.rs:329 */
  shr3(pruned, sk);
  mul_scalar(&ap, &C_B8, sk, 32);                 /* lib.rs:330 (public) */
  fr_t in[5] = {rp.x, rp.y, ap.x, ap.y, ZERO};
  fr_from_le(&in[4], msg);
  fr_t hm; poseidon5(&hm, in);                    /* lib.rs:332-333 */
  uint8_t hmb[32]; fr_to_le(hmb, &hm);
  /* s = r + hm * (sk << 3) mod l ; sk << 3 == pruned (its low 3 bits are already 0) */
  uint64_t a[4], b[4], prod[9] = {0};
  memcpy(a, hmb, 32); memcpy(b, pruned, 32);
  for (int i = 0; i < 4; i++) { u128 c = 0; for (int j = 0; j < 4; j++) { c += (u128)a[i] * b[j] + prod[i + j]; prod[i + j] = (uint64_t)c; c >>= 64; } prod[i + 4] = (uint64_t)c; }
  u128 c = 0;
  for (int i = 0; i < 9; i++) { c += (u128)prod[i] + (i < 4 ? r[i] : 0); prod[i] = (uint64_t)c; c >>= 64; }
  uint64_t s[4]; mod_l(prod, 9, s);               /* lib.rs:335-339 */
  fr_to_le(out_r, &rp.x); fr_to_le(out_r + 32, &rp.y);
  memcpy(out_s, s, 32);
  return 1;
}

/* ---- codec (lib.rs:166-178, 192-224; utils.rs:109-160, 215-223) -------- */
static void fr_pow(fr_t *out, const fr_t *base, const uint64_t e[4]) {
  fr_t r = R1;
  for (int i = 255; i >= 0; i--) {
    fr_square(&r);
    if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&r, base);
  }
  *out = r;
}
static int legendre_symbol(const fr_t *a) { /* utils.rs:215-223: -1 only for a non-residue */
  static const uint64_t HALF[4] = {0xa1f0fac9f8000000ULL, 0x9419f4243cdcb848ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL}; /* (r-1)/2 */
  fr_t ls; fr_pow(&ls, a, HALF);
  fr_t m1 = ZERO; fr_sub(&m1, &R1); /* -1 */
  return fr_eq(&ls, &m1) ? -1 : 1;
}
static fr_t TS_G;          /* n^s for the smallest non-residue n */
static const uint64_t TS_S[4] = {0x9b9709143e1f593fULL, 0x181585d2833e8487ULL, 0x131a029b85045b68ULL, 0x000000030644e72eULL};       /* s = (r-1) >> 28 */
static const uint64_t TS_S1H[4] = {0xcdcb848a1f0faca0ULL, 0x0c0ac2e9419f4243ULL, 0x098d014dc2822db4ULL, 0x0000000183227397ULL};     /* (s+1) >> 1 */
static pthread_once_t g_once_ts = PTHREAD_ONCE_INIT;
static void init_ts(void) {
  uint64_t n = 2; fr_t f;
  for (;; n++) { fr_from_u64(&f, n); if (legendre_symbol(&f) == -1) break; }   /* utils.rs:127-130 */
  fr_pow(&TS_G, &f, TS_S);
}
static void fr_pow2k(fr_t *x, unsigned k) { while (k--) fr_square(x); }
static int modsqrt(fr_t *out, const fr_t *a) { /* utils.rs:109-160 */
  pthread_once(&g_once_ts, init_ts);
  if (legendre_symbol(a) != 1 || fr_is_zero(a)) return 0;   /* "not a mod p square" */
  fr_t y, b, g = TS_G;
  fr_pow(&y, a, TS_S1H);
  fr_pow(&b, a, TS_S);
  unsigned r = 28;
  for (;;) {
    fr_t t = b; unsigned m = 0;
    while (!fr_eq(&t, &R1)) { fr_square(&t); m++; }
    if (m == 0) { *out = y; return 1; }
    t = g; fr_pow2k(&t, r - m - 1);
    fr_pow2k(&g, r - m);
    fr_mul(&y, &t);
    fr_mul(&b, &g);
    r = m;
  }
}
static int decompress_point(point_t *o, const uint8_t bb[32]) { /* lib.rs:192-224 */
  uint8_t b[32]; memcpy(b, bb, 32);
  int sign = (b[31] & 0x80) != 0;
  b[31] &= 0x7F;
  fr_t yr; memcpy(yr.l, b, 32);
  if (fr_geq(&yr, &MODULUS)) return 0;                      /* lib.rs:201-203 */
  fr_t y; fr_from_le(&y, b);
  fr_t y2 = y; fr_square(&y2);
  fr_t den = C_D; fr_mul(&den, &y2);
  fr_t t = C_A; fr_sub(&t, &den);                           /* A - D y^2 */
  fr_t deninv;
  if (!fr_inverse(&deninv, &t)) return 0;                   /* modinv(0) -> Err, utils.rs:13-15 */
  fr_t num = R1; fr_sub(&num, &y2);
  fr_mul(&num, &deninv);                                    /* x^2, lib.rs:214 */
  fr_t x;
  if (!modsqrt(&x, &num)) return 0;                         /* lib.rs:215 */
  uint8_t xb[32]; fr_to_le(xb, &x);
  fr_t xr; memcpy(xr.l, xb, 32);
  static const fr_t HALFQ = {{0xa1f0fac9f8000000ULL, 0x9419f4243cdcb848ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL}}; /* Q >> 1 */
  int gt = fr_geq(&xr, &HALFQ) && !fr_eq(&xr, &HALFQ);      /* x > Q/2 */
  if ((sign && !gt) || (!sign && gt)) { fr_t z = ZERO; fr_sub(&z, &x); x = z; }   /* lib.rs:217-219 */
  o->x = x; o->y = y;
  return 1;
}
static void compress_point(uint8_t r[32], const point_t *p) { /* lib.rs:166-178 */
  uint8_t xb[32]; fr_to_le(xb, &p->x); fr_to_le(r, &p->y);
  fr_t xr; memcpy(xr.l, xb, 32);
  static const fr_t HALFQ = {{0xa1f0fac9f8000000ULL, 0x9419f4243cdcb848ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL}};
  if (fr_geq(&xr, &HALFQ) && !fr_eq(&xr, &HALFQ)) r[31] |= 0x80;
}

/* ---- Schnorr variant (lib.rs:364-385) ----------------------------------- */
/* 1 = Ok(true), 0 = Ok(false), 2 = Err ("msg outside the Finite Field") */
static int verify_schnorr1(const uint8_t *pk, const uint8_t *msg, const uint8_t *rb, const uint8_t *s) {
  fr_t m; memcpy(m.l, msg, 32);
  point_t sg; mul_scalar(&sg, &C_B8, s, 32);          /* lib.rs:377 (computed before the hash, as there) */
  if (fr_geq(&m, &MODULUS) && !fr_eq(&m, &MODULUS)) return 2;   /* lib.rs:365-367 */
  point_t A_, R_;
  fr_from_le(&A_.x, pk); fr_from_le(&A_.y, pk + 32);
  fr_from_le(&R_.x, rb); fr_from_le(&R_.y, rb + 32);
  fr_t in[5] = {A_.x, A_.y, R_.x, R_.y, ZERO};        /* lib.rs:369 */
  fr_from_le(&in[4], msg);
  fr_t h; poseidon5(&h, in);
  uint8_t hb[32]; fr_to_le(hb, &h);
  point_t t; mul_scalar(&t, &A_, hb, 32);             /* lib.rs:381 */
  proj_t rp = {R_.x, R_.y, R1}, tp = {t.x, t.y, R1}, sum;
  proj_add(&sum, &rp, &tp);                           /* lib.rs:382 */
  point_t ra; proj_affine(&ra, &sum);
  return fr_eq(&sg.x, &ra.x) && fr_eq(&sg.y, &ra.y);  /* lib.rs:384 */
}

/* ======================= exported C entry points ======================== */
#define EXPORT __attribute__((visibility("default")))

EXPORT void bjjref_fr_mul(const uint8_t *a, const uint8_t *b, uint8_t *o) {
  fr_t x, y; fr_from_le(&x, a); fr_from_le(&y, b); fr_mul(&x, &y); fr_to_le(o, &x);
}
EXPORT void bjjref_fr_add(const uint8_t *a, const uint8_t *b, uint8_t *o) {
  fr_t x, y; fr_from_le(&x, a); fr_from_le(&y, b); fr_add(&x, &y); fr_to_le(o, &x);
}
EXPORT void bjjref_fr_sub(const uint8_t *a, const uint8_t *b, uint8_t *o) {
  fr_t x, y; fr_from_le(&x, a); fr_from_le(&y, b); fr_sub(&x, &y); fr_to_le(o, &x);
}
EXPORT int bjjref_fr_inverse(const uint8_t *a, uint8_t *o) {
  fr_t x, r; fr_from_le(&x, a);
  if (!fr_inverse(&r, &x)) { memset(o, 0, 32); return 0; }
  fr_to_le(o, &r); return 1;
}
/* p, q, out: 96-byte projective (x,y,z) */
EXPORT void bjjref_proj_add(const uint8_t *p, const uint8_t *q, uint8_t *out) {
  ensure_init();
  proj_t a, b, o;
  fr_from_le(&a.x, p); fr_from_le(&a.y, p + 32); fr_from_le(&a.z, p + 64);
  fr_from_le(&b.x, q); fr_from_le(&b.y, q + 32); fr_from_le(&b.z, q + 64);
  proj_add(&o, &a, &b);
  fr_to_le(out, &o.x); fr_to_le(out + 32, &o.y); fr_to_le(out + 64, &o.z);
}
EXPORT void bjjref_proj_affine(const uint8_t *p, uint8_t *out) {
  ensure_init();
  proj_t a; point_t o;
  fr_from_le(&a.x, p); fr_from_le(&a.y, p + 32); fr_from_le(&a.z, p + 64);
  proj_affine(&o, &a);
  fr_to_le(out, &o.x); fr_to_le(out + 32, &o.y);
}
/* pt: 64-byte affine point, or NULL for B8; scalar: nbytes little-endian */
EXPORT void bjjref_mul_scalar(const uint8_t *pt, const uint8_t *scalar, size_t nbytes, uint8_t *out) {
  ensure_init();
  point_t p, o;
  if (pt) { fr_from_le(&p.x, pt); fr_from_le(&p.y, pt + 32); } else p = C_B8;
  mul_scalar(&o, &p, scalar, nbytes);
  fr_to_le(out, &o.x); fr_to_le(out + 32, &o.y);
}
EXPORT void bjjref_poseidon5(const uint8_t *in, uint8_t *out) {
  ensure_init();
  fr_t v[5], h;
  for (int i = 0; i < 5; i++) fr_from_le(&v[i], in + 32 * i);
  poseidon5(&h, v);
  fr_to_le(out, &h);
}
EXPORT int bjjref_verify(const uint8_t *pk, const uint8_t *rb8, const uint8_t *s, const uint8_t *msg) {
  ensure_init();
  return verify1(pk, rb8, s, msg);
}


/* out: 64-byte point, returns 1 = Ok, 0 = Err (out zeroed) */
EXPORT int bjjref_decompress_point(const uint8_t *in, uint8_t *out) {
  ensure_init();
  point_t p;
  if (!decompress_point(&p, in)) { memset(out, 0, 64); return 0; }
  fr_to_le(out, &p.x); fr_to_le(out + 32, &p.y);
  return 1;
}
EXPORT void bjjref_compress_point(const uint8_t *in, uint8_t *out) {
  ensure_init();
  point_t p; fr_from_le(&p.x, in); fr_from_le(&p.y, in + 32);
  compress_point(out, &p);
}
/* decompress pk (32 B) and the signature (64 B: compressed R, then s) as lib.rs:192-224, 260-268,
 * then verify (lib.rs:395-412).  1 = true, 0 = false, 2 = a point failed to decompress (Err). */
EXPORT int bjjref_verify_compressed(const uint8_t *pk32, const uint8_t *sig64, const uint8_t *msg) {
  ensure_init();
  uint8_t pk[64], r[64];
  if (!bjjref_decompress_point(pk32, pk)) return 2;
  if (!bjjref_decompress_point(sig64, r)) return 2;
  return verify1(pk, r, sig64 + 32, msg);
}

EXPORT void bjjref_blake512(const uint8_t *msg, size_t len, uint8_t *out) { blake512(msg, len, out); }
EXPORT void bjjref_scalar_key(const uint8_t *key, uint8_t *out32) {
  uint8_t h[64], pr[32]; scalar_key_pruned(key, h, pr); shr3(pr, out32);
}
EXPORT void bjjref_public(const uint8_t *key, uint8_t *out_xy) {
  uint8_t sk[32]; bjjref_scalar_key(key, sk);
  bjjref_mul_scalar(NULL, sk, 32, out_xy);
}
EXPORT int bjjref_sign(const uint8_t *key, const uint8_t *msg, uint8_t *out_r, uint8_t *out_s) {
  ensure_init();
  if (!sign1(key, msg, out_r, out_s)) { memset(out_r, 0, 64); memset(out_s, 0, 32); return 0; }
  return 1;
}

EXPORT int bjjref_verify_schnorr(const uint8_t *pk, const uint8_t *msg, const uint8_t *rb, const uint8_t *s) {
  ensure_init();
  return verify_schnorr1(pk, msg, rb, s);
}

/* ---- threaded batch drivers (CPU baseline + bulk expected values) ------ */
typedef struct {
  int kind; /* 0 fixed-base, 1 var-base, 2 poseidon5, 3 verify, 4 decompress, 5 compress, 6 verify-compressed, 7 sign, 8 public, 9 verify_schnorr, 10 point add */
  const uint8_t *a, *b, *c, *d; uint8_t *out; size_t lo, hi;
} job_t;
static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  for (size_t i = j->lo; i < j->hi; i++) {
    switch (j->kind) {
      case 0: bjjref_mul_scalar(NULL, j->a + 32 * i, 32, j->out + 64 * i); break;
      case 1: bjjref_mul_scalar(j->a + 64 * i, j->b + 32 * i, 32, j->out + 64 * i); break;
      case 2: bjjref_poseidon5(j->a + 160 * i, j->out + 32 * i); break;
      case 3: j->out[i] = (uint8_t)verify1(j->a + 64 * i, j->b + 64 * i, j->c + 32 * i, j->d + 32 * i); break;
      case 4: ((uint8_t *)j->b)[i] = (uint8_t)bjjref_decompress_point(j->a + 32 * i, j->out + 64 * i); break;
      case 5: bjjref_compress_point(j->a + 64 * i, j->out + 32 * i); break;
      case 6: j->out[i] = (uint8_t)bjjref_verify_compressed(j->a + 32 * i, j->b + 64 * i, j->c + 32 * i); break;
      case 7: ((uint8_t *)j->d)[i] = (uint8_t)bjjref_sign(j->a + 32 * i, j->b + 32 * i, j->out + 64 * i, (uint8_t *)j->c + 32 * i); break;
      case 8: bjjref_public(j->a + 32 * i, j->out + 64 * i); break;
      case 9: j->out[i] = (uint8_t)verify_schnorr1(j->a + 64 * i, j->d + 32 * i, j->b + 64 * i, j->c + 32 * i); break;
      case 10: {  /* p.projective().add(&q.projective()).affine(): the criterion case `add` of benches/bench_babyjubjub.rs:26-31 */
        uint8_t pa[96] = {0}, qa[96] = {0}, sum[96];
        memcpy(pa, j->a + 64 * i, 64); pa[64] = 1;   /* Point::projective: z = 1, src/lib.rs:141-147 */
        memcpy(qa, j->b + 64 * i, 64); qa[64] = 1;
        bjjref_proj_add(pa, qa, sum);
        bjjref_proj_affine(sum, j->out + 64 * i);
      } break;
    }
  }
  return NULL;
}
static void run_batch(int kind, const uint8_t *a, const uint8_t *b, const uint8_t *c, const uint8_t *d,
                      uint8_t *out, size_t n, int nthreads) {
  ensure_init();
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
  job_t *jobs = (job_t *)malloc(sizeof(job_t) * nthreads);
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (job_t){kind, a, b, c, d, out, n * t / nthreads, n * (t + 1) / nthreads};
    if (nthreads == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
}
EXPORT void bjjref_mul_fixed_base_batch(const uint8_t *scalars, size_t n, uint8_t *out, int nthreads) {
  run_batch(0, scalars, NULL, NULL, NULL, out, n, nthreads);
}
EXPORT void bjjref_mul_var_base_batch(const uint8_t *pts, const uint8_t *scalars, size_t n, uint8_t *out, int nthreads) {
  run_batch(1, pts, scalars, NULL, NULL, out, n, nthreads);
}
EXPORT void bjjref_poseidon5_batch(const uint8_t *in, size_t n, uint8_t *out, int nthreads) {
  run_batch(2, in, NULL, NULL, NULL, out, n, nthreads);
}
EXPORT void bjjref_verify_batch(const uint8_t *pk, const uint8_t *rb8, const uint8_t *s, const uint8_t *msg,
                                size_t n, uint8_t *ok, int nthreads) {
  run_batch(3, pk, rb8, s, msg, ok, n, nthreads);
}
EXPORT void bjjref_decompress_batch(const uint8_t *in, size_t n, uint8_t *out_xy, uint8_t *ok, int nthreads) {
  run_batch(4, in, ok, NULL, NULL, out_xy, n, nthreads);
}
EXPORT void bjjref_compress_batch(const uint8_t *in_xy, size_t n, uint8_t *out, int nthreads) {
  run_batch(5, in_xy, NULL, NULL, NULL, out, n, nthreads);
}
EXPORT void bjjref_verify_compressed_batch(const uint8_t *pk32, const uint8_t *sig64, const uint8_t *msg, size_t n,
                                           uint8_t *ok, int nthreads) {
  run_batch(6, pk32, sig64, msg, NULL, ok, n, nthreads);
}
EXPORT void bjjref_sign_batch(const uint8_t *keys, const uint8_t *msgs, size_t n, uint8_t *out_r, uint8_t *out_s,
                              uint8_t *ok, int nthreads) {
  run_batch(7, keys, msgs, out_s, ok, out_r, n, nthreads);
}
EXPORT void bjjref_public_batch(const uint8_t *keys, size_t n, uint8_t *out_xy, int nthreads) {
  run_batch(8, keys, NULL, NULL, NULL, out_xy, n, nthreads);
}
EXPORT void bjjref_verify_schnorr_batch(const uint8_t *pk, const uint8_t *rb, const uint8_t *s, const uint8_t *msg,
                                        size_t n, uint8_t *ok, int nthreads) {
  run_batch(9, pk, rb, s, msg, ok, n, nthreads);
}
EXPORT void bjjref_point_add_batch(const uint8_t *p_xy, const uint8_t *q_xy, size_t n, uint8_t *out_xy, int nthreads) {
  run_batch(10, p_xy, q_xy, NULL, NULL, out_xy, n, nthreads);
}
