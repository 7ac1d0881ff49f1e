// Signer side: PrivateKey::scalar_key / public / sign,
// src/lib.rs:284-342, with blh() = Blake-512 (src/lib.rs:226-237; third-party blake-hash 0.4.0,
// Cargo.toml:17 -- the ORIGINAL BLAKE, not BLAKE2; restated from the published specification and
// pinned by the digest KAT of src/lib.rs:695-696).
#pragma once
#include "bjj_device.hpp"

namespace bjj {

// ---------------------------------------------------------------------------
// Blake-512, one 128-byte block (both uses here -- a 32-byte key and a 64-byte
// h[32..64] || msg32 -- fit one padded block).  Words are big-endian 64-bit.
// ---------------------------------------------------------------------------
BJJ_HD u64 blake_ror(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
BJJ_HD u64 blake_c(int i) {
  constexpr u64 C[16] = {0x243F6A8885A308D3ULL, 0x13198A2E03707344ULL, 0xA4093822299F31D0ULL, 0x082EFA98EC4E6C89ULL,
                         0x452821E638D01377ULL, 0xBE5466CF34E90C6CULL, 0xC0AC29B7C97C50DDULL, 0x3F84D5B5B5470917ULL,
                         0x9216D5D98979FB1BULL, 0xD1310BA698DFB5ACULL, 0x2FFD72DBD01ADFB7ULL, 0xB8E1AFED6A267E96ULL,
                         0xBA7C9045F12C7F99ULL, 0x24A19947B3916CF7ULL, 0x0801F2E2858EFC16ULL, 0x636920D871574E69ULL};
  return C[i];
}
BJJ_HD int blake_sigma(int r, int i) {
  constexpr unsigned char S[10][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
  return S[r % 10][i];
}
#define BJJ_BLAKE_G(a, b, c, d, r, i)                                                   \
  do {                                                                                  \
    const int j_ = blake_sigma(r, 2 * (i)), k_ = blake_sigma(r, 2 * (i) + 1);           \
    v[a] += v[b] + (m[j_] ^ blake_c(k_)); v[d] = blake_ror(v[d] ^ v[a], 32);            \
    v[c] += v[d]; v[b] = blake_ror(v[b] ^ v[c], 25);                                    \
    v[a] += v[b] + (m[k_] ^ blake_c(j_)); v[d] = blake_ror(v[d] ^ v[a], 16);            \
    v[c] += v[d]; v[b] = blake_ror(v[b] ^ v[c], 11);                                    \
  } while (0)
// digest of a message of `nbytes` (<= 111) bytes given as the big-endian words of its padded block
BJJ_HD void blake512_one_block(const u64 m[16], u64 nbits, u64 out[8]) {
  constexpr u64 IV[8] = {0x6A09E667F3BCC908ULL, 0xBB67AE8584CAA73BULL, 0x3C6EF372FE94F82BULL, 0xA54FF53A5F1D36F1ULL,
                         0x510E527FADE682D1ULL, 0x9B05688C2B3E6C1FULL, 0x1F83D9ABFB41BD6BULL, 0x5BE0CD19137E2179ULL};
  u64 v[16];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = IV[i];
#pragma unroll
  for (int i = 0; i < 4; i++) v[8 + i] = blake_c(i);
  v[12] = blake_c(4) ^ nbits; v[13] = blake_c(5) ^ nbits; v[14] = blake_c(6); v[15] = blake_c(7);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    BJJ_BLAKE_G(0, 4, 8, 12, r, 0); BJJ_BLAKE_G(1, 5, 9, 13, r, 1); BJJ_BLAKE_G(2, 6, 10, 14, r, 2); BJJ_BLAKE_G(3, 7, 11, 15, r, 3);
    BJJ_BLAKE_G(0, 5, 10, 15, r, 4); BJJ_BLAKE_G(1, 6, 11, 12, r, 5); BJJ_BLAKE_G(2, 7, 8, 13, r, 6); BJJ_BLAKE_G(3, 4, 9, 14, r, 7);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = IV[i] ^ v[i] ^ v[i + 8];
}
BJJ_HD u32 bswap32(u32 x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }
// message bytes arrive as little-endian u32 words (memory order); BLAKE reads big-endian u64 words
BJJ_HD u64 be64_from_le_words(u32 w0, u32 w1) { return ((u64)bswap32(w0) << 32) | bswap32(w1); }
// Blake-512 of `nw` (8 or 16) little-endian u32 words = 32 or 64 message bytes; digest as 16 LE u32 words
BJJ_HD void blake512_words(const u32* w, int nw, u32 dig[16]) {
  u64 m[16], h[8];
#pragma unroll
  for (int i = 0; i < 16; i++) m[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) if (2 * i < nw) m[i] = be64_from_le_words(w[2 * i], w[2 * i + 1]);
  m[nw / 2] = 0x8000000000000000ULL;          // the 1-bit right after the message
  m[13] |= 1;                                 // the 1-bit that ends the padding (byte 111)
  m[15] = (u64)nw * 32;                       // bit length (high word m[14] = 0)
  blake512_one_block(m, (u64)nw * 32, h);
#pragma unroll
  for (int i = 0; i < 8; i++) { dig[2 * i] = bswap32((u32)(h[i] >> 32)); dig[2 * i + 1] = bswap32((u32)h[i]); }
}

// bits [lo, lo+261) of a little-endian word array as 9 x 29-bit limbs
BJJ_HD Fr limbs_from_bits(const u32* w, int nw, int lo) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = lo + 29 * i, wi = bit >> 5, sh = bit & 31;
    u64 two = 0;
    if (wi < nw) two = w[wi];
    if (wi + 1 < nw) two |= (u64)w[wi + 1] << 32;
    r.v[i] = (u32)(two >> sh) & MASK29;
  }
  return r;
}
// ---------------------------------------------------------------------------
// PrivateKey::scalar_key (src/lib.rs:284-302): Blake-512(key)[..32], pruned, >> 3
// `pruned` is the value before the shift (== scalar_key << 3, what lib.rs:335 multiplies by);
// hi = digest words 8..15 (h[32..64], the nonce prefix of sign).
// ---------------------------------------------------------------------------
BJJ_HD void scalar_key_words(const u32 key[8], u32 sk[8], u32 pruned[8], u32 hi[8]) {
  u32 dig[16];
  blake512_words(key, 8, dig);
#pragma unroll
  for (int i = 0; i < 8; i++) { pruned[i] = dig[i]; hi[i] = dig[8 + i]; }
  pruned[0] &= 0xfffffff8u;                         // h[0] &= 0xF8
  pruned[7] = (pruned[7] & 0x7fffffffu) | 0x40000000u;   // h[31] &= 0x7F; h[31] |= 0x40
#pragma unroll
  for (int i = 0; i < 8; i++) sk[i] = (pruned[i] >> 3) | (i < 7 ? (pruned[i + 1] << 29) : 0u);
}

// PrivateKey::sign (src/lib.rs:308-342).  Returns false where the reference returns Err
// (msg > Q, :309-311; the outputs are then meaningless -- straight-line like verify_fast_t, so that the
// whole wave reaches the fixed-base gathers).  R comes out as canonical words, s as a canonical integer mod l.
template <class G>
BJJ_HD bool sign_item(const u32 key[8], const u32 msg[8], const G& fb_table, int W, int nwin, u32 out_rx[8],
                      u32 out_ry[8], u32 out_s[8], const Consts& K) {
  const bool good = !words_gt_modulus(msg);
  u32 sk[8], pruned[8], buf[16], dig[16];
  scalar_key_words(key, sk, pruned, buf);                        // :316 (h), buf[0..8) = h[32..64]
#pragma unroll
  for (int i = 0; i < 8; i++) buf[8 + i] = msg[i];               // :318-325  h[32..64] || msg32
  blake512_words(buf, 16, dig);                                  // :326
  // r = from_bytes_le(digest) mod l (:327-328):  X = X0 + 2^261 X1
  Fr r = fl_canon4(fr_add(fl_mul(limbs_from_bits(dig, 16, 0), K.L_R1, K), fl_mul(limbs_from_bits(dig, 16, 261), K.L_R2, K)), K);
  u32 rw[8];
  fr_to_words(r, rw);
  Ext Rp = fixed_base_mul(fb_table, W, nwin, rw, K);             // :329
  Ext Ap = fixed_base_mul(fb_table, W, nwin, sk, K);             // :330 (public)
  // both to affine Montgomery coordinates of the reference curve with ONE inversion
  Fr zi = fr_inv(fr_mul(Rp.Z, Ap.Z));
  Fr zr = fr_mul(zi, Ap.Z), za = fr_mul(zi, Rp.Z);
  Fr h[5];
  h[0] = fr_mul(fr_mul(Rp.X, zr), K.FINV); h[1] = fr_mul(Rp.Y, zr);
  h[2] = fr_mul(fr_mul(Ap.X, za), K.FINV); h[3] = fr_mul(Ap.Y, za);
  h[4] = fr_to_mont_words(msg);                                  // :321
  Fr hm = poseidon5(h, K);                                       // :332-333 (partial rounds one at a time: faster here, A/B'd)
  Fr hm_plain = fr_canon(fr_mul(hm, fr_one_plain()));            // :336
  // s = r + hm * (scalar_key << 3) mod l   (:335-339);  scalar_key << 3 == pruned
  Fr t = fl_mul(fr_from_words(pruned), K.L_R2, K);               // pruned * 2^261
  Fr s = fl_canon4(fr_add(fl_mul(hm_plain, t, K), r), K);
  fr_from_mont_words(h[0], out_rx); fr_from_mont_words(h[1], out_ry);
  fr_to_words(s, out_s);
  return good;
}
BJJ_HD bool sign_item(const u32 key[8], const u32 msg[8], const u32* fb_table, int W, int nwin, u32 out_rx[8],
                      u32 out_ry[8], u32 out_s[8], const Consts& K) {
  return sign_item(key, msg, GatherPerLane{fb_table}, W, nwin, out_rx, out_ry, out_s, K);
}

// PrivateKey::sign_schnorr (src/lib.rs:344-361) with the 1024-bit nonce k supplied by the caller (the
// reference draws it from rand::thread_rng, :347-348; randomness stays on the host).  r = k*B8 (:351),
// h = schnorr_hash(pk, m, r) (:355, hash input order (pk, r, m), :369), s = k + scalar_key*h as a PLAIN
// integer -- the reference never reduces it (:359) -- written as 40 little-endian words (< 2^1025).
// Returns false where the reference returns Err (msg > Q, :365-367).
constexpr int SCHNORR_K_WORDS = 32;   // 1024-bit nonce
constexpr int SCHNORR_S_WORDS = 40;   // 160-byte record of s
template <class G>
BJJ_HD bool sign_schnorr_item(const u32 key[8], const u32 msg[8], const u32 k[SCHNORR_K_WORDS], const G& fb_table, int W,
                              int nwin, u32 out_rx[8], u32 out_ry[8], u32 out_s[SCHNORR_S_WORDS], const Consts& K) {
  const bool good = !words_gt_modulus(msg);                      // straight-line: see sign_item
  u32 sk[8], pruned[8], hi[8];
  scalar_key_words(key, sk, pruned, hi);                         // :358 (and :354 through public())
  // k mod l for the fixed-base engine (B8 has order l):  k = X0 + 2^261 X1 + 2^522 X2 + 2^783 X3
  Fr t1 = fl_mul(limbs_from_bits(k, SCHNORR_K_WORDS, 261), K.L_R2, K);
  Fr t2 = fl_mul(fl_mul(limbs_from_bits(k, SCHNORR_K_WORDS, 522), K.L_R2, K), K.L_R2, K);
  Fr t3 = fl_mul(fl_mul(fl_mul(limbs_from_bits(k, SCHNORR_K_WORDS, 783), K.L_R2, K), K.L_R2, K), K.L_R2, K);
  Fr t0 = fl_mul(limbs_from_bits(k, SCHNORR_K_WORDS, 0), K.L_R1, K);
  Fr kr = fl_canon4(fr_add(fl_canon4(fr_add(t0, t1), K), fl_canon4(fr_add(t2, t3), K)), K);  // each term < 2l
  u32 kw[8];
  fr_to_words(kr, kw);
  Ext Rp = fixed_base_mul(fb_table, W, nwin, kw, K);             // :351
  Ext Ap = fixed_base_mul(fb_table, W, nwin, sk, K);             // :354
  Fr zi = fr_inv(fr_mul(Rp.Z, Ap.Z));
  Fr zr = fr_mul(zi, Ap.Z), za = fr_mul(zi, Rp.Z);
  Fr h[5];
  h[0] = fr_mul(fr_mul(Ap.X, za), K.FINV); h[1] = fr_mul(Ap.Y, za);   // pk first, :369
  h[2] = fr_mul(fr_mul(Rp.X, zr), K.FINV); h[3] = fr_mul(Rp.Y, zr);
  h[4] = fr_to_mont_words(msg);                                  // :368
  Fr hm = poseidon5(h, K);
  u32 hw[8];
  fr_from_mont_words(hm, hw);                                    // canonical integer, :371
  // s = k + sk * h over the integers (:359): 8 x 8 word product accumulated onto k
  u64 acc = 0;
#pragma unroll
  for (int c = 0; c < SCHNORR_S_WORDS; c++) {
    u64 carry = 0;
    if (c < SCHNORR_K_WORDS) acc += k[c];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int j = c - i;
      if (j >= 0 && j < 8) {
        const u64 p = (u64)sk[i] * hw[j];
        acc += (u32)p;
        carry += p >> 32;
      }
    }
    out_s[c] = (u32)acc;
    acc = (acc >> 32) + carry;
  }
  fr_from_mont_words(h[2], out_rx); fr_from_mont_words(h[3], out_ry);
  return good;
}
BJJ_HD bool sign_schnorr_item(const u32 key[8], const u32 msg[8], const u32 k[SCHNORR_K_WORDS], const u32* fb_table, int W,
                              int nwin, u32 out_rx[8], u32 out_ry[8], u32 out_s[SCHNORR_S_WORDS], const Consts& K) {
  return sign_schnorr_item(key, msg, k, GatherPerLane{fb_table}, W, nwin, out_rx, out_ry, out_s, K);
}

}  // namespace bjj
