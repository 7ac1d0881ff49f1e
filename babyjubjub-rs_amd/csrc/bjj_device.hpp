// Per-item bodies of the hot path (one item per lane).  __host__ __device__ so
// that tests/emul can run exactly this code on the CPU with bound assertions; the
// shipped library only instantiates them inside HIP kernels (bjj_hip.hip).
#pragma once
#include "poseidon.hpp"
#include "bjj_constants.inc"  // BJJ_L_NINV29 (macros only; safe to include repeatedly)

namespace bjj {

struct alignas(16) U4 { u32 x, y, z, w; };

// 32-byte little-endian integer <-> 8 words (16-byte aligned memory)
BJJ_HD void load_w8(const void* p, u32 w[8]) {
  const U4* q = (const U4*)p;
  U4 a = q[0], b = q[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
BJJ_HD void store_w8(void* p, const u32 w[8]) {
  U4* q = (U4*)p;
  U4 a = {w[0], w[1], w[2], w[3]}, b = {w[4], w[5], w[6], w[7]};
  q[0] = a; q[1] = b;
}
// ---- fixed-base table: entry = Niels in 32 words (128 B = one cache line) ----
constexpr int NIELS_WORDS = 32;  // 27 used
BJJ_HD Niels load_niels(const u32* p) {
  const U4* q = (const U4*)p;
  U4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
  Niels n;
  n.ymx = Fr{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x}};
  n.ypx = Fr{{c.y, c.z, c.w, d.x, d.y, d.z, d.w, e.x, e.y}};
  n.t2d = Fr{{e.z, e.w, f.x, f.y, f.z, f.w, g.x, g.y, g.z}};
  return n;
}
BJJ_HD void store_niels(u32* p, const Niels& n) {
  U4* q = (U4*)p;
  q[0] = U4{n.ymx.v[0], n.ymx.v[1], n.ymx.v[2], n.ymx.v[3]};
  q[1] = U4{n.ymx.v[4], n.ymx.v[5], n.ymx.v[6], n.ymx.v[7]};
  q[2] = U4{n.ymx.v[8], n.ypx.v[0], n.ypx.v[1], n.ypx.v[2]};
  q[3] = U4{n.ypx.v[3], n.ypx.v[4], n.ypx.v[5], n.ypx.v[6]};
  q[4] = U4{n.ypx.v[7], n.ypx.v[8], n.t2d.v[0], n.t2d.v[1]};
  q[5] = U4{n.t2d.v[2], n.t2d.v[3], n.t2d.v[4], n.t2d.v[5]};
  q[6] = U4{n.t2d.v[6], n.t2d.v[7], n.t2d.v[8], 0};
  q[7] = U4{0, 0, 0, 0};
}
// ---- per-lane variable-base table -------------------------------------------------------------------
// {0..8} * P in projective-Niels form, one table per lane in global scratch.  BJJ_PNIELS_LAYOUT selects the entry format
// per translation unit (the host side allocates VB_TABLE_WORDS_MAX words per table, which serves either):
//   0  raw:    the 4 x 9 limb words as they are (144 B, nine 16-byte quarters), 9 entries incl. a stored identity
//              (1 296 B per table); no packing arithmetic, entries straddle cache lines
//   1  packed: four 256-bit integers per entry (128 B = one cache line; Y-X weakly reduced first so that every component
//              is below 2^256), digit 0 selects the constant identity (1, 1, 0, 2): 8 entries, 1 KB per table
//   2  packed entries at the raw stride (36 words, 9 slots) -- what rounds 1-2 shipped by accident of two disagreeing
//              guards; kept so that the round-3 A/B has the old build as its control
// Interleaved A/B of the three on one MI355X (profiles/r03_ab_pniels_layout_verify_block.txt): the verify kernel is 3.7-5 %
// faster with raw entries than with either packed form (two tables per lane, 68 entry loads per item: the ~100 plain
// instructions of an unpack cost more than the bytes), the variable-base kernel 0.5-0.9 % faster with packed than with raw
// entries and 1.5-2.2 % faster than with the old hybrid.  So: k_verify.hip = raw, k_var.hip = packed, default = raw.
#ifndef BJJ_PNIELS_LAYOUT
#define BJJ_PNIELS_LAYOUT 0
#endif
constexpr int VB_TABLE_WORDS_MAX = 36 * 9;   // the largest layout: what the host allocates per table
#if BJJ_PNIELS_LAYOUT == 1
constexpr int PNIELS_WORDS = 32;
constexpr int VB_TABLE_ENTRIES = 8;  // 1*P .. 8*P
#else
constexpr int PNIELS_WORDS = 36;
constexpr int VB_TABLE_ENTRIES = 9;  // 0*P .. 8*P
#endif
constexpr int VB_TABLE_WORDS = PNIELS_WORDS * VB_TABLE_ENTRIES;
constexpr int VB_VERIFY_WORDS = 2 * VB_TABLE_WORDS;  // verify keeps two per-lane tables (-8A and -+R)
BJJ_HD PNiels pniels_identity() {
  PNiels id; id.ymx = fr_one(); id.ypx = fr_one(); id.t2d = fr_zero(); id.z2 = fr_dbl(fr_one());
  return id;
}
#if BJJ_PNIELS_LAYOUT == 0
BJJ_HD PNiels load_pniels_raw(const u32* p) {
  const U4* q = (const U4*)p;
  U4 t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = q[i];
  const u32* w = (const u32*)t;
  PNiels n;
#pragma unroll
  for (int i = 0; i < 9; i++) { n.ymx.v[i] = w[i]; n.ypx.v[i] = w[9 + i]; n.t2d.v[i] = w[18 + i]; n.z2.v[i] = w[27 + i]; }
  return n;
}
BJJ_HD void store_pniels_raw(u32* p, const PNiels& n) {
  u32 w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) { w[i] = n.ymx.v[i]; w[9 + i] = n.ypx.v[i]; w[18 + i] = n.t2d.v[i]; w[27 + i] = n.z2.v[i]; }
  U4* q = (U4*)p;
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = U4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
}
// entry k (0..8) of a per-lane table
BJJ_HD PNiels vb_table_load(const u32* tbl, u32 k) { return load_pniels_raw(tbl + k * PNIELS_WORDS); }
BJJ_HD void vb_table_store(u32* tbl, u32 k, const PNiels& n) { store_pniels_raw(tbl + k * PNIELS_WORDS, n); }
BJJ_HD void vb_table_store_identity(u32* tbl) { store_pniels_raw(tbl, pniels_identity()); }
#else
BJJ_HD PNiels vb_table_load(const u32* tbl, u32 k) {
  const u32* p = tbl + (k ? k - 1 : 0) * PNIELS_WORDS;
  u32 w[8];
  PNiels n;
  load_w8(p, w);      n.ymx = fr_from_words(w);
  load_w8(p + 8, w);  n.ypx = fr_from_words(w);
  load_w8(p + 16, w); n.t2d = fr_from_words(w);
  load_w8(p + 24, w); n.z2 = fr_from_words(w);
  const PNiels id = pniels_identity();
  const bool z = k == 0;
  n.ymx = fr_select(z, id.ymx, n.ymx); n.ypx = fr_select(z, id.ypx, n.ypx);
  n.t2d = fr_select(z, id.t2d, n.t2d); n.z2 = fr_select(z, id.z2, n.z2);
  return n;
}
BJJ_HD void vb_table_store(u32* tbl, u32 k, const PNiels& n) {  // k = 1..8; components N-form: ymx < 8r, the others < 4r
  u32* p = tbl + (k - 1) * PNIELS_WORDS;
  u32 w[8];
  fr_to_words(fr_reduce_weak(n.ymx), w); store_w8(p, w);
  fr_to_words(n.ypx, w); store_w8(p + 8, w);
  fr_to_words(n.t2d, w); store_w8(p + 16, w);
  fr_to_words(n.z2, w);  store_w8(p + 24, w);
}
BJJ_HD void vb_table_store_identity(u32*) {}
#endif

// =============================================================================
// fixed base:  acc + n * B8   with the precomputed signed-window table
//   n is first reduced mod l (B8 has order l, so this is exact) and recoded into nwin = ceil(252 / W)
//   signed digits d_j in [-2^(W-1), 2^(W-1)];  table[j][k] = Niels( k * 2^(W j) * B8 ), k = 0 .. 2^(W-1)
//   (k = 0 is the identity entry), -k*P by swapping y-x / y+x and negating 2D'xy.
//   WINDOW 0 stores 2x'y in place of 2D'x'y ("T form"): a multiplication from scratch starts by lifting the entry of
//   window 0 to extended coordinates (2x' : 2y : 2 : 2x'y), which then costs no multiplication at all; the
//   accumulating form (verify) multiplies that one entry by D' instead.
// =============================================================================
BJJ_HD int fixed_nwin(int W) { return (252 + W - 1) / W; }            // l < 2^251: the top digit absorbs the last carry
BJJ_HD size_t fixed_stride(int W) { return ((size_t)1 << (W - 1)) + 1; }  // entries per window
// 256-bit integer mod l (8 words in, 8 words out, < l).  Quotient estimate from the top byte:
// floor(2^264 / l) = 10834, so q = (top byte * 10834) >> 16 satisfies q*l <= s < (q + 1.17) l.
BJJ_HD void scalar_mod_l(const u32 w[8], u32 out[8], const Consts& K) {
  Fr s = fr_from_words(w);
  const u32 q = ((w[7] >> 24) * 10834u) >> 16;  // <= 42
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    c += (int64_t)s.v[i] - (int64_t)((u64)q * K.L.v[i]);
    s.v[i] = (i < NL - 1) ? ((u32)c & MASK29) : (u32)c;
    c >>= 29;
  }
  s = fr_cond_sub_kr(s, K.L.v);
  fr_to_words(s, out);
}
// signed digit j of sc (< l): returns the table slot, sets neg; carry threads through the windows LSB-first
BJJ_HD size_t fixed_digit_slot(const u32 sc[8], int j, int W, u32& carry, bool& neg) {
  const u32 u = scalar_window(sc, j, W) + carry;
  neg = u > (1u << (W - 1));
  carry = neg ? 1u : 0u;
  const u32 d = neg ? (1u << W) - u : u;
  return (size_t)j * fixed_stride(W) + d;
}
// The same digits as a STREAM (windows 0, 1, 2, .. in order): the scalar sits in a shift register of 8 words that moves down by
// W bits per window, so that a window is always the low W bits of word 0 -- eight funnel shifts per window instead of two
// 8-way selections of words by a run-time window index (what scalar_window costs inside a rolled loop: ~60 instructions
// per window in K1; profiles/r04_ab_digit_stream.txt).  4 <= W <= 28.
struct DigitStream {
  u32 w[8];
  u32 carry;
  size_t base;      // slot of digit 0 of the current window
  size_t stride;
  u32 mask, half;
  int W;
};
BJJ_HD DigitStream digit_stream(const u32 sc[8], int W) {
  DigitStream d;
#pragma unroll
  for (int i = 0; i < 8; i++) d.w[i] = sc[i];
  d.carry = 0; d.base = 0; d.stride = fixed_stride(W); d.mask = (1u << W) - 1u; d.half = 1u << (W - 1); d.W = W;
  return d;
}
BJJ_HD size_t digit_next(DigitStream& d, bool& neg) {
  const u32 u = (d.w[0] & d.mask) + d.carry;
  neg = u > d.half;
  d.carry = neg ? 1u : 0u;
  const u32 dig = neg ? (d.mask + 1u) - u : u;
  const size_t slot = d.base + dig;
  d.base += d.stride;
#pragma unroll
  for (int i = 0; i < 7; i++) d.w[i] = (d.w[i] >> d.W) | (d.w[i + 1] << (32 - d.W));   // v_alignbit_b32
  d.w[7] >>= d.W;
  return slot;
}
// -(x', y) = (-x', y): swap y-x' / y+x', negate 2D'x'y.  The negation stays carry-less (limbs < 2^30): the entry
// only ever feeds one multiplication whose other operand is N-form.
BJJ_HD Niels niels_cneg_lazy(const Niels& n, bool neg) {
  Niels r;
  r.ymx = fr_select(neg, n.ypx, n.ymx);
  r.ypx = fr_select(neg, n.ymx, n.ypx);
  r.t2d = fr_select(neg, fr_sub_lazy(fr_zero(), n.t2d), n.t2d);
  return r;
}
// ---- gather policies ---------------------------------------------------------------------------------
// How a lane obtains table entry `slot`.  A policy has a `Pending` handle type and
//     void  issue(size_t slot, Pending& p, int buf);   // start the gather (buf: staging buffer 0 / 1)
//     Niels finish(Pending& p, int buf);               // complete it, return the entry
//     static int wave_max(int v);                      // max of v over the lanes that gather together (identity if lane-private)
// The loops below issue gather j+1, run the 7 multiplications of addition j, and only then finish.
//  * GatherPerLane (here): every lane reads its own 128-byte entry (7 x dwordx4).  Works in divergent code and
//    on the host (tests/emul).  hipcc sinks the loads to their first use, so nothing overlaps, and each of
//    the 7 load instructions touches 64 different lines/pages: fine for cache- or TLB-resident tables.
//  * GatherCoopLds (bjj_hip.hip, device only): the 64 lanes of a wave fetch their 64 entries together, 8 full
//    lines per instruction, straight into LDS -- what makes the 43 / 155 GB tables pay off.
struct GatherPerLane {
  struct Pending { Niels e; };
  static constexpr int kBuffers = 2;   // gathers that may be in flight at once
  const u32* table;
  static BJJ_HD int wave_max(int v) { return v; }   // a lane-private policy makes no assumption about the other lanes
  BJJ_HD void issue(size_t slot, Pending& p, int) const { p.e = load_niels(table + slot * NIELS_WORDS); }
  BJJ_HD Niels finish(Pending& p, int) const { return p.e; }
};
//  * GatherScan (here; the signer's constant-time option): a lane reads EVERY entry of the window its slot lies in -- the
//    addresses depend on the window index alone, never on the digit -- and keeps the one its digit names by arithmetic
//    selection.  For the small table the signer builds for this purpose (4-bit windows: 9 entries of 128 B per window,
//    73 KB in all, cache-resident); 9 loads + 8 x 27 selects per addition instead of one gather.
struct GatherScan {
  struct Pending { Niels e; };
  static constexpr int kBuffers = 2;
  const u32* table;
  u32 stride;   // entries per window (fixed_stride(W))
  static BJJ_HD int wave_max(int v) { return v; }
  BJJ_HD void issue(size_t slot, Pending& p, int) const {
    const u32 s = (u32)slot, base = (s / stride) * stride, d = s - base;   // base = window index * stride: public
    Niels e = load_niels(table + (size_t)base * NIELS_WORDS);
#pragma unroll 1
    for (u32 k = 1; k < stride; k++) {
      const Niels t = load_niels(table + (size_t)(base + k) * NIELS_WORDS);
      const bool hit = d == k;
      e.ymx = fr_select(hit, t.ymx, e.ymx); e.ypx = fr_select(hit, t.ypx, e.ypx); e.t2d = fr_select(hit, t.t2d, e.t2d);
    }
    p.e = e;
  }
  BJJ_HD Niels finish(Pending& p, int) const { return p.e; }
};
#if defined(__HIP_DEVICE_COMPILE__)
#define BJJ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   // nothing is scheduled across: keeps `finish` behind the addition
#else
#define BJJ_SCHED_FENCE() ((void)0)
#endif

// acc + sc * B8, sc < l.  The result's T is not computed (callers only compare or convert X, Y, Z).
template <class G>
BJJ_HD Ext fixed_base_accumulate(Ext acc, const G& g, int W, int nwin, const u32 sc[8], const Consts& K) {
  DigitStream ds = digit_stream(sc, W);
  bool neg;
  typename G::Pending p;
  g.issue(digit_next(ds, neg), p, 0);
  Niels cur = niels_cneg_lazy(g.finish(p, 0), neg);
  cur.t2d = fr_mul(cur.t2d, K.DP);                                 // window 0 is stored in T form
#pragma unroll 1
  for (int j = 0; j + 1 < nwin; j++) {
    g.issue(digit_next(ds, neg), p, (j + 1) & 1);                  // in flight during this window's 7 multiplications
    acc = ext_madd(acc, cur);
    BJJ_SCHED_FENCE();
    cur = niels_cneg_lazy(g.finish(p, (j + 1) & 1), neg);
  }
  return ext_madd<false>(acc, cur);
}
BJJ_HD Ext fixed_base_accumulate(Ext acc, const u32* table, int W, int nwin, const u32 sc[8], const Consts& K) {
  return fixed_base_accumulate(acc, GatherPerLane{table}, W, nwin, sc, K);
}

// n * B8 from scratch (n any 256-bit integer): window 0's entry (T form) is lifted directly to extended coordinates
// (X:Y:Z:T) = (2x' : 2y : 2 : 2x'y) -- additions and subtractions only -- instead of a 7M addition to the identity.
template <class G>
BJJ_HD Ext fixed_base_mul(const G& g, int W, int nwin, const u32 raw[8], const Consts& K) {
  u32 sc[8];
  scalar_mod_l(raw, sc, K);
  DigitStream ds = digit_stream(sc, W);
  bool neg0, neg;
  typename G::Pending p0, p;
  g.issue(digit_next(ds, neg0), p0, 0);
  const size_t slot1 = digit_next(ds, neg);
  if (G::kBuffers >= 2) g.issue(slot1, p, 1);
  const Niels n0 = niels_cneg_lazy(g.finish(p0, 0), neg0);
  if (G::kBuffers < 2) g.issue(slot1, p, 1);
  Ext acc;
  acc.X = fr_reduce_weak(fr_sub(n0.ypx, n0.ymx));  // ext_madd wants coordinates < 2r
  acc.Y = fr_add(n0.ypx, n0.ymx);
  acc.Z = fr_add(fr_one(), fr_one()); acc.T = fr_add(n0.t2d, fr_zero());   // carry sweep only (T feeds one multiplication)
  BJJ_SCHED_FENCE();
  Niels cur = niels_cneg_lazy(g.finish(p, 1), neg);
#pragma unroll 1
  for (int j = 1; j + 1 < nwin; j++) {
    g.issue(digit_next(ds, neg), p, (j + 1) & 1);
    acc = ext_madd(acc, cur);
    BJJ_SCHED_FENCE();
    cur = niels_cneg_lazy(g.finish(p, (j + 1) & 1), neg);
  }
  return ext_madd<false>(acc, cur);  // nwin >= 9: the last window's addition, T not needed by the epilogue
}
BJJ_HD Ext fixed_base_mul(const u32* table, int W, int nwin, const u32 raw[8], const Consts& K) {
  return fixed_base_mul(GatherPerLane{table}, W, nwin, raw, K);
}

// =============================================================================
// variable base, on-curve fast path:  n * P  (n < 2^254 already reduced mod 8l)
// signed 4-bit windows, per-lane table tbl[0..8] = {0, P, .., 8P} in thread-private
// memory (global scratch on the GPU).
// =============================================================================
// affine_base: P has Z == 1 (it came from ext_from_ref_affine): k*P = (k-1)*P + P is then a MIXED addition (7M, the 2Z
// term is an addition) instead of the general one (8M).
BJJ_HD void vb_build_table(const Ext& P, u32* tbl, const Consts& K, bool affine_base = false) {
  vb_table_store_identity(tbl);
  PNiels p1 = ext_to_pniels(P, K);
  vb_table_store(tbl, 1, p1);
  const Niels p1a = {p1.ymx, p1.ypx, p1.t2d};
  Ext cur = P;
#pragma unroll 1
  for (int k = 2; k <= 8; k++) {
    cur = affine_base ? ext_madd(cur, p1a) : ext_add_pn(cur, p1);
    vb_table_store(tbl, (u32)k, ext_to_pniels(cur, K));
  }
}
// A table entry as an extended point (all four coordinates doubled: the same projective point):
// (Y+X) - (Y-X) = 2X, (Y+X) + (Y-X) = 2Y, 2Z, and 2T = (2D'T) / D'.  Replaces "identity + entry" (8M) at the top
// window of a windowed loop by one multiplication -- none when T is not needed.
BJJ_HD Ext pniels_to_ext(const PNiels& e, const Consts& K, bool need_t) {
  Ext r;
  r.X = fr_reduce_weak(fr_sub8(e.ypx, e.ymx));     // entries: ymx < 8r, ypx < 4r
  r.Y = fr_reduce_weak(fr_add(e.ypx, e.ymx));
  r.Z = fr_reduce_weak(e.z2);
  r.T = fr_zero();
  if (need_t) r.T = fr_mul(e.t2d, K.DPINV);
  return r;
}
// nwin windows of 4 bits, most significant first; needs sc < 2^(4*nwin - 2) so that the
// signed recoding (add 0x88..8, digit = nibble - 8) cannot carry out of the top window: the top
// nibble plus an incoming carry must stay below 8 (callers pass nwin = 64 with sc < 2^254).
// The T coordinate of an addition is only read by a following ADDITION: every addition here is followed by the four
// doublings of the next window (which never read T), so it is computed only for the very last one, and only when the caller
// goes on adding (final_t).  The top window is peeled off the loop -- inside the loop the compiler cannot see that the
// "no doublings yet" case never recurs, and would keep every T alive.
BJJ_HD Ext vb_mul_windowed(const u32* tbl, const u32 sc[8], int nwin, const Consts& K, bool final_t = false) {
  u32 t[8];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (u64)sc[i] + 0x88888888u; t[i] = (u32)c; c >>= 32; }
  Ext acc;
  {
    const int j = nwin - 1;
    const int d = (int)((t[j >> 3] >> ((j & 7) * 4)) & 15u) - 8;
    const PNiels e = vb_table_load(tbl, (u32)(d < 0 ? -d : d));
    acc = pniels_to_ext(pniels_cneg(e, d < 0), K, final_t && nwin == 1);
  }
#pragma unroll 1
  for (int j = nwin - 2; j >= 0; j--) {
    int d = (int)((t[j >> 3] >> ((j & 7) * 4)) & 15u) - 8;
    bool neg = d < 0;
    u32 idx = (u32)(neg ? -d : d);
    PNiels e = vb_table_load(tbl, idx);  // issued ahead of the doublings
#pragma unroll 1
    for (int k = 0; k < 3; k++) acc = ext_dbl<false>(acc);
    acc = ext_dbl<true>(acc);
    acc = ext_add_pn(acc, pniels_cneg(e, neg), final_t && j == 0);
  }
  return acc;
}

// =============================================================================
// exact path: the reference's own operation sequence (src/lib.rs:149-164, 70-85)
// x, y Montgomery; scalar = nw words, little-endian.  Returns affine Montgomery.
// =============================================================================
BJJ_HD void ref_mul_scalar(const Fr& x, const Fr& y, const u32* sc, int nw, Fr& ox, Fr& oy, const Consts& K) {
  int bits = 0;
  for (int i = nw - 1; i >= 0; i--)
    if (sc[i]) { bits = 32 * i + 32 - __builtin_clz(sc[i]); break; }
  RefProj r; r.x = fr_zero(); r.y = fr_one(); r.z = fr_one();
  RefProj e; e.x = x; e.y = y; e.z = fr_one();
#pragma unroll 1
  for (int i = 0; i < bits; i++) {
    if ((sc[i >> 5] >> (i & 31)) & 1) r = ref_add(r, e, K);
    e = ref_add(e, e, K);
  }
  if (fr_is_zero(r.z)) { ox = fr_zero(); oy = fr_zero(); return; }  // src/lib.rs:71-76
  Fr zi = fr_inv(r.z);
  ox = fr_mul(r.x, zi); oy = fr_mul(r.y, zi);
}

// fixed-base table entry (j, k) = Niels( k * 2^(W j) * B8 ), fully reduced; tform: third word = 2x'y instead of 2D'x'y
BJJ_HD Niels fixed_table_entry(u32 k, int j, int W, const Consts& K, bool tform = false) {
  Ext base = ext_from_ref_affine(K.B8X, K.B8Y, K);
  PNiels bn = ext_to_pniels(base, K);
  PNiels idn = pniels_identity();
  Ext acc = ext_identity();
#pragma unroll 1
  for (int b = W - 1; b >= 0; b--) {
    acc = ext_dbl<true>(acc);
    const bool bit = (k >> b) & 1;
    PNiels sel;
    sel.ymx = fr_select(bit, bn.ymx, idn.ymx); sel.ypx = fr_select(bit, bn.ypx, idn.ypx);
    sel.t2d = fr_select(bit, bn.t2d, idn.t2d); sel.z2 = fr_select(bit, bn.z2, idn.z2);
    acc = ext_add_pn(acc, sel);
  }
#pragma unroll 1
  for (int d = 0; d < W * j; d++) acc = ext_dbl<true>(acc);
  Fr zi = fr_inv(acc.Z);
  Fr x = fr_mul(acc.X, zi), y = fr_mul(acc.Y, zi);
  Niels n;
  n.ymx = fr_canon(fr_sub(y, x)); n.ypx = fr_canon(fr_add(y, x));
  n.t2d = tform ? fr_canon(fr_dbl(fr_mul(x, y))) : fr_canon(fr_mul(fr_mul(x, y), K.D2P));
  return n;
}

// ---- table construction in chains ------------------------------------------------------------------
// One thread owns `cnt` consecutive digits k0 .. k0+cnt-1 of window j (slots slot0 ..): it walks
// k*P_j by repeated mixed addition of P_j, parks (X, Y, Z, running product of Z) of every step in the entry's
// own 128-byte slot (4 x 8 words), inverts the final product once and walks back (Montgomery's trick),
// overwriting each slot with the affine Niels form.  ~14 multiplications per entry instead of a
// W(j+1)-step ladder and an inversion per entry (fixed_table_entry, kept as the independent definition).
BJJ_HD void store_chain_slot(u32* p, const Fr& x, const Fr& y, const Fr& z, const Fr& pre) {
  u32 w[8];
  fr_to_words(x, w);   store_w8(p, w);        // N-form values < 2r < 2^256: 8 words hold them exactly
  fr_to_words(y, w);   store_w8(p + 8, w);
  fr_to_words(z, w);   store_w8(p + 16, w);
  fr_to_words(pre, w); store_w8(p + 24, w);
}
BJJ_HD Fr load_chain_word(const u32* p) {
  u32 w[8];
  load_w8(p, w);
  return fr_from_words(w);
}
BJJ_HD Niels niels_from_affine(const Fr& x, const Fr& y, const Consts& K, bool tform) {
  Niels n;
  n.ymx = fr_canon(fr_sub(y, x)); n.ypx = fr_canon(fr_add(y, x));
  n.t2d = tform ? fr_canon(fr_dbl(fr_mul(x, y))) : fr_canon(fr_mul(fr_mul(x, y), K.D2P));
  return n;
}
BJJ_HD void fixed_table_chain(u32* table, const Niels& base, size_t slot0, u32 k0, u32 cnt, int W, const Consts& K) {
  const bool tform = slot0 < fixed_stride(W);   // window 0
  // k0 * P_j, MSB first; the addition is computed unconditionally and selected (uniform control flow)
  Ext acc = ext_identity();
#pragma unroll 1
  for (int b = W - 1; b >= 0; b--) {
    acc = ext_dbl<true>(acc);
    Ext sum = ext_madd(acc, base);
    const bool bit = (k0 >> b) & 1;
    acc.X = fr_select(bit, sum.X, acc.X); acc.Y = fr_select(bit, sum.Y, acc.Y);
    acc.Z = fr_select(bit, sum.Z, acc.Z); acc.T = fr_select(bit, sum.T, acc.T);
  }
  Fr pre = fr_one();
#pragma unroll 1
  for (u32 i = 0; i < cnt; i++) {
    pre = fr_mul(pre, acc.Z);
    store_chain_slot(table + (slot0 + i) * NIELS_WORDS, acc.X, acc.Y, acc.Z, pre);
    acc = ext_madd(acc, base);
  }
  Fr inv = fr_inv(pre);
#pragma unroll 1
  for (u32 i = cnt; i-- > 0;) {
    u32* slot = table + (slot0 + i) * NIELS_WORDS;
    Fr X = load_chain_word(slot), Y = load_chain_word(slot + 8), Z = load_chain_word(slot + 16);
    Fr prev = i > 0 ? load_chain_word(slot - NIELS_WORDS + 24) : fr_one();
    Fr zi = fr_mul(inv, prev);
    inv = fr_mul(inv, Z);
    store_niels(slot, niels_from_affine(fr_mul(X, zi), fr_mul(Y, zi), K, tform));
  }
}
// Link check of the finished table (proof by induction that every entry is k * 2^(W j) * B8):
//   T[j][0] = identity, T[j][1] = P_j, T[j][k] + P_j = T[j][k+1], P_{j+1} = 2 * T[j][2^(W-1)], P_0 = B8,
//   every entry canonical with 2D'x'y consistent.  Returns the number of violated conditions for slot (j, k).
BJJ_HD Ext niels_lift(const Niels& n, const Consts& K, bool tform) {  // (2x' : 2y : 2 : 2x'y)
  Ext e;
  e.X = fr_reduce_weak(fr_sub(n.ypx, n.ymx)); e.Y = fr_add(n.ypx, n.ymx);
  e.Z = fr_add(fr_one(), fr_one()); e.T = tform ? n.t2d : fr_mul(n.t2d, K.DPINV);
  return e;
}
BJJ_HD bool ext_equals_niels(const Ext& p, const Niels& n) {  // p == (x', y) of n, projectively
  Fr two_x = fr_sub(n.ypx, n.ymx), two_y = fr_add(n.ypx, n.ymx);
  return fr_eq(fr_mul(two_x, p.Z), fr_dbl(p.X)) && fr_eq(fr_mul(two_y, p.Z), fr_dbl(p.Y));
}
BJJ_HD bool niels_limbs_equal(const Niels& a, const Niels& b) {
  bool eq = true;
  for (int i = 0; i < NL; i++) eq = eq && a.ymx.v[i] == b.ymx.v[i] && a.ypx.v[i] == b.ypx.v[i] && a.t2d.v[i] == b.t2d.v[i];
  return eq;
}
BJJ_HD int fixed_table_check_slot(const u32* table, const u32* bases, int j, u32 k, int W, int nwin, const Consts& K) {
  const size_t stride = fixed_stride(W);
  const Niels e = load_niels(table + ((size_t)j * stride + k) * NIELS_WORDS);
  const Niels base = load_niels(bases + (size_t)j * NIELS_WORDS);
  const bool tform = j == 0;   // window 0 holds 2x'y in the third word, the bases (and every other window) 2D'x'y
  int bad = 0;
  // canonical limbs and values
  for (int i = 0; i < NL; i++) bad += (e.ymx.v[i] >> 29) != 0 || (e.ypx.v[i] >> 29) != 0 || (e.t2d.v[i] >> 29) != 0;
  bad += !niels_limbs_equal(e, Niels{fr_canon(e.ymx), fr_canon(e.ypx), fr_canon(e.t2d)});
  // 2 * t2d == D' * (ypx^2 - ymx^2)      (4x'y = (y+x')^2 - (y-x')^2);  T form: without the D'
  const Fr dsq = fr_sub(fr_sqr(e.ypx), fr_sqr(e.ymx));
  bad += !fr_eq(tform ? dsq : fr_mul(dsq, K.DP), fr_dbl(e.t2d));
  if (k == 0) bad += !niels_limbs_equal(e, Niels{fr_one(), fr_one(), fr_zero()});
  if (k == 1) bad += !niels_limbs_equal(Niels{e.ymx, e.ypx, tform ? fr_canon(fr_mul(e.t2d, K.DP)) : e.t2d}, base);
  if (k + 1 < stride) {
    const Niels nx = load_niels(table + ((size_t)j * stride + k + 1) * NIELS_WORDS);
    bad += !ext_equals_niels(ext_madd(niels_lift(e, K, tform), base), nx);
  } else if (j + 1 < nwin) {
    const Niels nb = load_niels(bases + (size_t)(j + 1) * NIELS_WORDS);
    bad += !ext_equals_niels(ext_dbl<false>(niels_lift(e, K, tform)), nb);
  }
  if (j == 0 && k == 1) {
    Ext g = ext_from_ref_affine(K.B8X, K.B8Y, K);
    bad += !ext_equals_niels(g, e);
  }
  return bad;
}

// one variable-base item: (x, y) Montgomery on the reference curve, raw 256-bit scalar.
// Fast path (on-curve): extended point on the a'=-1 curve.  Off-curve points need the
// reference's exact formula sequence; the kernels defer them to a compacted second launch
// (a wave that contains one such lane would otherwise execute both paths for all 64 lanes).
BJJ_HD Ext var_base_fast(const Fr& x, const Fr& y, const u32 sc[8], u32* tbl, const Consts& K) {
  u32 red[8];
  scalar_mod_order(sc, red, K);
  Ext P = ext_from_ref_affine(x, y, K);
  vb_build_table(P, tbl, K, true);
  return vb_mul_windowed(tbl, red, 64, K);
}
// exact replay of the reference's bit-serial loop; result mapped onto the a'=-1 curve with
// Z = 1 so that the shared affine epilogue maps it back unchanged.
BJJ_HD Ext var_base_exact(const Fr& x, const Fr& y, const u32 sc[8], const Consts& K) {
  Fr ox, oy;
  ref_mul_scalar(x, y, sc, 8, ox, oy, K);
  Ext p;
  p.X = fr_mul(ox, K.F); p.Y = oy; p.Z = fr_one(); p.T = fr_zero();
  return p;
}
BJJ_HD Ext var_base_item(const Fr& x, const Fr& y, const u32 sc[8], u32* tbl, const Consts& K) {
  if (ref_on_curve(x, y, K)) return var_base_fast(x, y, sc, tbl, K);
  return var_base_exact(x, y, sc, K);
}

// =============================================================================
// codec row: Point::compress (src/lib.rs:166-178), decompress_point
// (src/lib.rs:192-224 with utils.rs modinv / modsqrt).  decompress_point's result does not
// depend on which square root Tonelli-Shanks returns (the sign rule of lib.rs:217-219 picks
// by the sign bit), so any correct root is bit-identical to the reference.
// =============================================================================
// a^((s-1)/2), s = (r-1)/2^28: fixed 225-bit exponent.  Sliding 3-bit windows over the CONSTANT exponent (program
// from gen_tables.py: 223 squarings + 51 multiplications by a, a^3, a^5 or a^7, + 4 for the table) instead of plain
// square-and-multiply (224 + 98).  The schedule is the same for every lane.
BJJ_HD_NOINLINE Fr fr_pow_ts(const Fr& a) {
  constexpr unsigned char PROG[2 * BJJ_TS_POW_STEPS] = BJJ_TS_POW_PROG;
  const Fr a2 = fr_sqr(a);
  const Fr p3 = fr_mul(a, a2), p5 = fr_mul(p3, a2), p7 = fr_mul(p5, a2);
  Fr x = fr_zero();
#pragma unroll 1
  for (int st = 0; st < BJJ_TS_POW_STEPS; st++) {
    const int nsq = PROG[2 * st], idx = PROG[2 * st + 1];
    const Fr m = fr_select(idx & 2, fr_select(idx & 1, p7, p5), fr_select(idx & 1, p3, a));
#pragma unroll 1
    for (int k = 0; k < nsq; k++) x = fr_sqr(x);
    x = st == 0 ? m : fr_mul(x, m);
  }
#pragma unroll 1
  for (int k = 0; k < BJJ_TS_POW_TAIL; k++) x = fr_sqr(x);
  return x;
}
// Square root in F_r (r - 1 = 2^28 s).  With w = a^((s-1)/2): x = a w satisfies x^2 = a b, b = a^s in the
// subgroup <G> of order 2^28.  The discrete log e of b (four 7-bit digits, Pohlig-Hellman: 21 + 14 + 7
// squarings and a 128-entry lookup per digit, tables from gen_tables.py) gives the root x G^(-e/2);
// e is even exactly for residues.  Data-independent schedule (no lane divergence).  Returns false for
// a == 0 (the reference's modsqrt errors on 0, utils.rs:117-119) and for non-residues; root < 2r.
BJJ_HD u32 ts_digit(const Fr& c, const Consts& K) {  // c in <H> (order 128) -> its exponent
  Fr cc = fr_canon(c);
  return K.TSHASH[(cc.v[0] * BJJ_TS_HASH_MAGIC) >> 21];
}
BJJ_HD bool fr_sqrt(const Fr& a, Fr& root, const Consts& K) {
  Fr w = fr_pow_ts(a);
  Fr x = fr_mul(a, w);   // a^((s+1)/2)
  Fr b = fr_mul(x, w);   // a^s
  Fr c = b;
#pragma unroll 1
  for (int i = 0; i < 21; i++) c = fr_sqr(c);
  const u32 e0 = ts_digit(c, K);
  b = fr_mul(b, K.TSN[e0]);
  c = b;
#pragma unroll 1
  for (int i = 0; i < 14; i++) c = fr_sqr(c);
  const u32 e1 = ts_digit(c, K);
  b = fr_mul(b, K.TSN[128 + e1]);
  c = b;
#pragma unroll 1
  for (int i = 0; i < 7; i++) c = fr_sqr(c);
  const u32 e2 = ts_digit(c, K);
  b = fr_mul(b, K.TSN[256 + e2]);
  const u32 e3 = ts_digit(b, K);
  x = fr_mul(x, K.TSH[e0 >> 1]);
  x = fr_mul(x, K.TSH[64 + e1]);
  x = fr_mul(x, K.TSH[192 + e2]);
  x = fr_mul(x, K.TSH[320 + e3]);
  root = x;
  return fr_eq(fr_sqr(x), a) && !fr_is_zero(a);   // also rejects odd e0 (non-residues) and garbage digits
}
// plain canonical N-form value > (r-1)/2 ?
BJJ_HD bool plain_gt_halfq(const Fr& v, const Consts& K) {
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) { u32 t = K.HALFQ.v[i] - v.v[i] - borrow; borrow = t >> 31; }
  return borrow != 0;
}
BJJ_HD bool words_ge_modulus(const u32 w[8]) {
  const u32 M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  bool gt = false, eq = true;
#pragma unroll
  for (int i = 7; i >= 0; i--) { gt = gt || (eq && w[i] > M[i]); eq = eq && (w[i] == M[i]); }
  return gt || eq;
}
// decompress_point (src/lib.rs:192-224): 32 bytes -> canonical (x, y) words; false = Err
BJJ_HD bool decompress_item(const u32 in[8], u32 ox[8], u32 oy[8], const Consts& K) {
  u32 w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = in[i];
  const bool sign = (w[7] >> 31) != 0;                       // :196-199
  w[7] &= 0x7fffffffu;
  bool ok = !words_ge_modulus(w);                            // :201-203
  Fr y = fr_to_mont_words(w);
  Fr y2 = fr_sqr(y);
  Fr den = fr_sub(K.A, fr_mul(K.D, y2));                     // A - D y^2 (never 0: A/D is a non-residue)
  Fr num = fr_sub(fr_one(), y2);
  Fr x2 = fr_mul(num, fr_inv(den));                          // :207-214
  Fr x;
  ok = fr_sqrt(x2, x, K) && ok;                              // :215
  constexpr u32 R1[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  Fr xc = fr_cond_sub_kr(fr_mul(x, fr_one_plain()), R1);     // canonical integer
  const bool gt = plain_gt_halfq(xc, K);
  if (sign != gt) {                                          // :217-219: x <- -x
    Fr neg;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) { u32 t = R1[i] - xc.v[i] - borrow; borrow = t >> 31; neg.v[i] = (i < NL - 1) ? (t & MASK29) : t; }
    xc = neg;
  }
  fr_to_words(xc, ox);
#pragma unroll
  for (int i = 0; i < 8; i++) { oy[i] = ok ? w[i] : 0u; ox[i] = ok ? ox[i] : 0u; }
  return ok;
}
// Point::compress (src/lib.rs:166-178): y little-endian, bit 255 = (x > (r-1)/2); inputs >= r are reduced
BJJ_HD void compress_item(const u32 xw[8], const u32 yw[8], u32 out[8], const Consts& K) {
  Fr x = fr_canon(fr_from_words(xw)), y = fr_canon(fr_from_words(yw));
  fr_to_words(y, out);
  if (plain_gt_halfq(x, K)) out[7] |= 0x80000000u;
}

// msg > Q ?   (src/lib.rs:396-398; msg == Q is accepted and wraps to 0)
BJJ_HD bool words_gt_modulus(const u32 w[8]) {
  const u32 M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  bool gt = false, eq = true;
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    gt = gt || (eq && w[i] > M[i]);
    eq = eq && (w[i] == M[i]);
  }
  return gt;
}

// =============================================================================
// one EdDSA-Poseidon verification (src/lib.rs:395-412)
// =============================================================================
struct VerifyIn { const void *pk, *r, *s, *msg; };  // this item's 64/64/32/32-byte records

// true when verify() must take the exact path: msg in range and pk or R off the curve
BJJ_HD bool verify_needs_exact(const VerifyIn& in, const Consts& K) {
  u32 w[8];
  load_w8(in.msg, w);
  if (words_gt_modulus(w)) return false;  // verdict is `false` either way (:396-398)
  load_w8(in.r, w);                    Fr rx = fr_to_mont_words(w);
  load_w8((const char*)in.r + 32, w);  Fr ry = fr_to_mont_words(w);
  if (!ref_on_curve(rx, ry, K)) return true;
  load_w8(in.pk, w);                   Fr ax = fr_to_mont_words(w);
  load_w8((const char*)in.pk + 32, w); Fr ay = fr_to_mont_words(w);
  return !ref_on_curve(ax, ay, K);
}
// ---------------------------------------------------------------------------
// arithmetic mod l (the prime subgroup order) in the same 9 x 29-bit limb form, Montgomery
// radix 2^261.  Only three products per signature, so it is not tuned.
// ---------------------------------------------------------------------------
BJJ_HD Fr fl_mul(const Fr& a, const Fr& b, const Consts& K) {  // a*b*2^-261 mod l, needs a*b < l*2^261; result < 2l
  u32 m[NL];
  Fr r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * K.L.v[k - i];
    m[k] = ((u32)acc * BJJ_L_NINV29) & MASK29;
    acc += (u64)m[k] * K.L.v[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - (NL - 1); i < NL; i++) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - (NL - 1); i < NL; i++) acc += (u64)m[i] * K.L.v[k - i];
    r.v[k - NL] = (u32)acc & MASK29;
    acc >>= 29;
  }
  r.v[NL - 1] = (u32)acc;
  return r;
}
// canonical value mod l of a plain N-form x < 4l
BJJ_HD Fr fl_canon4(const Fr& x, const Consts& K) {
  Fr t = fr_cond_sub_kr(x, K.L2.v);
  return fr_cond_sub_kr(t, K.L.v);
}


// =============================================================================
// Half-size scalars for EdDSA verification (Antipa, Brown, Gallant, Lambert, Struik, Vanstone:
// "Accelerated verification of ECDSA signatures", SAC 2005).  The check
//     D := s*B8 - 8*kappa*A - R == O            (kappa = hm mod l)
// is multiplied by an ODD integer v that is non-zero mod l -- hence invertible modulo the group
// order 8l, so v*D == O <=> D == O exactly, whatever torsion component R or A carry -- chosen
// such that u = v*kappa mod l is small as well.  Then
//     v*D = (v*s mod l)*B8 + u*(-8A) + v*(-R)
// needs one fixed-base multiplication and ONE joint double-and-add over ~126-bit u, |v| instead
// of a 251-bit variable-base multiplication.  (u, v) comes from the classical extended Euclid
// sequence r_i = s_i*l + t_i*kappa, stopped at the first r_i < 2^126: (r_i, t_i) if t_i is odd,
// otherwise the better of the previous pair (t odd because consecutive t's are never both even:
// t_{i+1} r_i - t_i r_{i+1} = +-l is odd) and the next pair (rejected when it degenerates to t = +-l).
// =============================================================================
BJJ_HD bool limbs_lt(const Fr& a, const Fr& b) {  // a < b, both N-form plain integers
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) { u32 t = a.v[i] - b.v[i] - borrow; borrow = t >> 31; }
  return borrow != 0;
}
BJJ_HD bool limbs_ge_2p126(const Fr& a) {  // 2^126 = bit 10 of limb 4
  return (a.v[8] | a.v[7] | a.v[6] | a.v[5] | (a.v[4] >> 10)) != 0;
}
BJJ_HD bool limbs_is_zero(const Fr& a) { return fr_is_zero_canon(a); }
BJJ_HD double limbs_to_double(const Fr& a) {  // rounded to 53 bits
  double d = (double)a.v[8];
#pragma unroll
  for (int i = NL - 2; i >= 0; i--) d = d * 536870912.0 + (double)a.v[i];
  return d;
}
BJJ_HD int limbs_bits(const Fr& a) {  // bit length
  int bits = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) if (a.v[i]) bits = 29 * i + 32 - __builtin_clz(a.v[i]);
  return bits;
}
BJJ_HD void limbs_submul(Fr& a, u32 q, const Fr& b) {  // a -= q*b  (result >= 0 by the caller's choice of q)
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    c += (int64_t)a.v[i] - (int64_t)((u64)q * b.v[i]);
    a.v[i] = (i < NL - 1) ? ((u32)c & MASK29) : (u32)c;
    c >>= 29;
  }
}
BJJ_HD void limbs_addmul(Fr& a, u32 q, const Fr& b) {  // a += q*b
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    c += (u64)a.v[i] + (u64)q * b.v[i];
    a.v[i] = (i < NL - 1) ? ((u32)c & MASK29) : (u32)c;
    c >>= 29;
  }
}
BJJ_HD Fr limbs_shift_up(const Fr& a, int k) {  // a * 2^(29 k), k in 0..8 (high limbs fall off: callers keep it in range)
  Fr r = a;
#pragma unroll
  for (int s = 0; s < NL - 1; s++) {
    if (s < k) {
#pragma unroll
      for (int i = NL - 1; i > 0; i--) r.v[i] = r.v[i - 1];
      r.v[0] = 0;
    }
  }
  return r;
}
// The quotient estimate below is only an UNDER-estimate -- which is what keeps every step exact -- under IEEE-754 division and
// multiplication: the (1 - 2^-48) factor covers two correctly rounded operations on 53-bit roundings of the operands, not an
// approximate reciprocal.  -ffast-math / -freciprocal-math would break verdicts silently, so such a build does not compile.
#if defined(__FAST_MATH__) || defined(__RECIPROCAL_MATH__)
#error "bjj: build without -ffast-math / -freciprocal-math (euclid_partial_step relies on IEEE f64 division)"
#endif
// one "partial quotient" step of Euclid on (r0 >= r1 > 0): r0 -= q r1, |t0| += q |t1| with
// 1 <= q <= floor(r0 / r1), where q = m * 2^(29 k) is the leading 29-bit digit of a 53-bit
// floating-point UNDER-estimate of the quotient (so even a 2^250 quotient takes <= 9 steps)
BJJ_HD void euclid_partial_step(Fr& r0, const Fr& r1, Fr& t0, const Fr& t1) {
  double qf = limbs_to_double(r0) / limbs_to_double(r1) * (1.0 - 1.0 / 281474976710656.0);  // * (1 - 2^-48)
  int k = 0;
#pragma unroll 1
  while (qf >= 536870912.0) { qf *= (1.0 / 536870912.0); k++; }                             // qf in [.., 2^29)
  u32 m = (u32)qf;
  m = m < 1u ? 1u : m;                                                                      // k == 0 here: q = 1 <= true quotient
  if (k == 0) {  // the common case: quotient below 2^29
    limbs_submul(r0, m, r1);
    limbs_addmul(t0, m, t1);
  } else {
    limbs_submul(r0, m, limbs_shift_up(r1, k));
    limbs_addmul(t0, m, limbs_shift_up(t1, k));
  }
}
// u (>= 0), |v|, sign(v) with u == v*kappa (mod l), v odd and != 0 (mod l); kappa plain canonical < l
#ifdef BJJ_EXP_LATTICE_NOINLINE   // experiment (profiles/r04_verify_register_file.txt): the Euclid phase as a real call
BJJ_HD_NOINLINE void lattice_short_pair(const Fr& kappa, Fr& u, Fr& vmag, bool& vneg, const Consts& K) {
#else
BJJ_HD void lattice_short_pair(const Fr& kappa, Fr& u, Fr& vmag, bool& vneg, const Consts& K) {
#endif
  Fr r0 = K.L, r1 = kappa, t0 = fr_zero(), t1 = fr_one_plain();
  bool s1 = false;  // sign of t1; t0 has the opposite sign (or is 0)
  while (limbs_ge_2p126(r1)) {
    euclid_partial_step(r0, r1, t0, t1);
    if (limbs_lt(r0, r1)) { Fr x = r0; r0 = r1; r1 = x; x = t0; t0 = t1; t1 = x; s1 = !s1; }
  }
  if (t1.v[0] & 1) { u = r1; vmag = t1; vneg = s1; return; }
  // t1 even (so r1 != 0 and t0 is odd): previous pair P = (r0, t0), next pair N = (r0 mod r1, t0 + q t1)
  const Fr pr = r0, pt = t0;
  while (!limbs_lt(r0, r1)) euclid_partial_step(r0, r1, t0, t1);
  const bool takeN = !limbs_is_zero(r0) && (limbs_bits(t0) < limbs_bits(pr));
  u = fr_select(takeN, r0, pr);
  vmag = fr_select(takeN, t0, pt);
  vneg = !s1;
}
// signed 4-bit recoding of a plain N-form scalar < 2^(4*nwin - 2): returns the words of sc + 0x88..8
BJJ_HD void recode_signed4(const Fr& sc, u32 t[8]) {
  u32 w[8];
  fr_to_words(sc, w);
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (u64)w[i] + 0x88888888u; t[i] = (u32)c; c >>= 32; }
}
// W = acc0 + u*P1 + |v|*P2 with per-lane tables tbl1 / tbl2 ({0..8}*P in PNiels form)
BJJ_HD Ext joint_mul_windowed(const u32* tbl1, const u32* tbl2, const Fr& u, const Fr& vmag, int nwin, const Consts& K) {
  u32 tu[8], tv[8];
  recode_signed4(u, tu);
  recode_signed4(vmag, tv);
  Ext acc;
  {  // top window, peeled (see vb_mul_windowed): two additions to the identity, no doublings
    const int j = nwin - 1;
    const int du = (int)((tu[j >> 3] >> ((j & 7) * 4)) & 15u) - 8;
    const int dv = (int)((tv[j >> 3] >> ((j & 7) * 4)) & 15u) - 8;
    const PNiels e1 = vb_table_load(tbl1, (u32)(du < 0 ? -du : du));
    const PNiels e2 = vb_table_load(tbl2, (u32)(dv < 0 ? -dv : dv));
    acc = pniels_to_ext(pniels_cneg(e1, du < 0), K, true);
    acc = ext_add_pn(acc, pniels_cneg(e2, dv < 0), nwin == 1);
  }
#pragma unroll 1
  for (int j = nwin - 2; j >= 0; j--) {
    const int du = (int)((tu[j >> 3] >> ((j & 7) * 4)) & 15u) - 8;
    const int dv = (int)((tv[j >> 3] >> ((j & 7) * 4)) & 15u) - 8;
    PNiels e1 = vb_table_load(tbl1, (u32)(du < 0 ? -du : du));
    PNiels e2 = vb_table_load(tbl2, (u32)(dv < 0 ? -dv : dv));
#pragma unroll 1
    for (int k = 0; k < 3; k++) acc = ext_dbl<false>(acc);
    acc = ext_dbl<true>(acc);
    acc = ext_add_pn(acc, pniels_cneg(e1, du < 0));
    acc = ext_add_pn(acc, pniels_cneg(e2, dv < 0), j == 0);   // T only for the caller's next addition (the fixed-base part)
  }
  return acc;
}

// u*P1 + |v|*P2 for the short pair (u, |v|) of the EdDSA fast path: per-lane tables of P1 (any Z) and P2 (Z == 1) in vb_tbl
// (VB_VERIFY_WORDS words), then the joint loop over the number of windows the WIDEST item of the wave needs.
// Signed recoding needs top nibble + carry < 8, i.e. scalars < 2^(4*jw - 2): an item needs ceil((bits + 2) / 4) windows --
// 32 for the 92 % of the pairs of at most 126 bits, 33 up to 130 bits (99.9 %), 63 for the odd kappa = (l+1)/2 with its
// 250-bit u, 64 (the cap) from 251 bits on.  Leading zero digits select the identity entry, so more windows than an item needs never change its result:
// 33 windows for 96 % of the waves, where rounds 1-3 ran a flat 34 (profiles/r04_ab_wave_windows.txt).  G::wave_max is the
// identity for the per-lane policies (host harness: every item runs exactly its own minimum, which is also the stricter test
// of the recoding bound); under the wave-cooperative policy it is a cross-lane maximum, i.e. a data-dependent trip count
// shared by the wave -- tests/devfuzz/joint.hip drives THIS function on the device with chosen bit lengths per lane
// (one lane at 250 bits next to lanes at 1, 126, 127, 130, 131 bits, partly filled waves).  *windows_run reports jw.
template <class G>
BJJ_HD Ext joint_short_pair(const Ext& p1, const Ext& p2, const Fr& u, const Fr& vmag, u32* vb_tbl, const Consts& K,
                            int* windows_run = nullptr) {
  u32* tbl2 = vb_tbl + VB_TABLE_WORDS;
  vb_build_table(p1, vb_tbl, K);             // -8A: three doublings, Z != 1
  vb_build_table(p2, tbl2, K, true);         // -+R: affine
  const int ub = limbs_bits(u), vb = limbs_bits(vmag);
  const int mb = ub > vb ? ub : vb;
  const int need = mb <= 2 ? 1 : (mb + 5) >> 2;
  const int jw = G::wave_max(need > 64 ? 64 : need);
  if (windows_run) *windows_run = jw;
  return joint_mul_windowed(vb_tbl, tbl2, u, vmag, jw, K);
}

// Fast path of verify (src/lib.rs:395-412) and, with SCHNORR, of verify_schnorr (src/lib.rs:375-385:
// hash input order (pk, R, msg) instead of (R, pk, msg), the hash is NOT multiplied by 8, and msg > Q is
// an Err -- verdict 2 -- rather than `false`).  Verdict 0 / 1 / 2; need_exact is set when pk or R is off
// the curve (the item then belongs to the exact path and the verdict returned here is meaningless).
// Straight-line on purpose: items whose verdict is already known (msg > Q, off-curve) run through the same
// arithmetic on their meaningless data instead of leaving early, so that all lanes of a wave reach the
// fixed-base part together -- the cooperative gather policy needs the whole wave, and a wave executes the
// instructions of its slowest lane anyway.
template <bool SCHNORR, class G>
BJJ_HD int verify_fast_t(const VerifyIn& in, const G& fb, int W, int nwin, u32* vb_tbl, const Consts& K,
                         bool& need_exact) {
  u32 w[8];
  load_w8(in.msg, w);
  const bool msg_gt = words_gt_modulus(w);                      // :396-398 / :365-367
  Fr h[5];
  h[4] = fr_to_mont_words(w);                                   // :399
  Fr rx, ry, ax, ay;
  load_w8(in.r, w);                    rx = fr_to_mont_words(w);
  load_w8((const char*)in.r + 32, w);  ry = fr_to_mont_words(w);
  load_w8(in.pk, w);                   ax = fr_to_mont_words(w);
  load_w8((const char*)in.pk + 32, w); ay = fr_to_mont_words(w);
  need_exact = !msg_gt && !(ref_on_curve(rx, ry, K) && ref_on_curve(ax, ay, K));
  if (SCHNORR) { h[0] = ax; h[1] = ay; h[2] = rx; h[3] = ry; }  // :369
  else         { h[0] = rx; h[1] = ry; h[2] = ax; h[3] = ay; }  // :400
  Fr hm = poseidon5_t<true>(h, K);                              // :400-404
  Fr hm_plain = fr_canon(fr_mul(hm, fr_one_plain()));           // canonical integer, :406
#ifdef BJJ_EXP_RELOAD_INPUTS   // experiment: the four coordinates are re-read (4 multiplications) instead of living through the hash
  load_w8(in.r, w);                    rx = fr_to_mont_words(w);
  load_w8((const char*)in.r + 32, w);  ry = fr_to_mont_words(w);
  load_w8(in.pk, w);                   ax = fr_to_mont_words(w);
  load_w8((const char*)in.pk + 32, w); ay = fr_to_mont_words(w);
#endif
  u32 sw[8];
  load_w8(in.s, sw);
  int verdict;
  if (SCHNORR) {
    // s*B8 == R + hm*A  <=>  hm*(-A) + s*B8 == R   (hm < r < 8l: no reduction; A may carry torsion and
    // hm may be even, so the half-size trick below does not apply)
    u32 kw[8];
    fr_to_words(hm_plain, kw);
    Ext negA = ext_from_ref_affine(fr_neg(ax), ay, K);
    vb_build_table(negA, vb_tbl, K, true);
    Ext q = vb_mul_windowed(vb_tbl, kw, 64, K, true);           // scalar < 2^254; T for the addition chain that follows
    u32 sl[8];
    scalar_mod_l(sw, sl, K);                                    // B8 has order l
    q = fixed_base_accumulate(q, fb, W, nwin, sl, K);           // + s*B8   (:377)
    Fr fx = fr_mul(rx, K.F);                                    // compare with R on the a'=-1 curve
    verdict = (fr_eq(q.X, fr_mul(fx, q.Z)) && fr_eq(q.Y, fr_mul(ry, q.Z))) ? 1 : 0;
    return msg_gt ? 2 : verdict;
  }
  // EdDSA: v*(s*B8 - 8*kappa*A - R) == O  with the short odd pair (u, v), u = v*kappa mod l (see above)
  Fr u, vmag;
  bool vneg;
  lattice_short_pair(plain_mod_l(hm_plain, K), u, vmag, vneg, K);
  // c = v*s mod l for the fixed-base part (B8 has order l): three Montgomery products mod l
  Fr sl = fl_mul(fr_from_words(sw), K.L_R1, K);                                  // s mod l (< 2l)
  Fr c = fl_canon4(fl_mul(vmag, fl_mul(sl, K.L_R2, K), K), K);                   // |v|*s mod l
  if (vneg && !limbs_is_zero(c)) { Fr t = K.L; limbs_submul(t, 1u, c); c = t; }  // l - c
  u32 cw[8];
  fr_to_words(c, cw);
  // P1 = -8A, P2 = -sign(v) R   (both on the a'=-1 curve)
  Ext p1 = ext_from_ref_affine(fr_neg(ax), ay, K);
  p1 = ext_dbl<false>(p1); p1 = ext_dbl<false>(p1); p1 = ext_dbl<true>(p1);
  Ext p2 = ext_from_ref_affine(vneg ? rx : fr_neg(rx), ry, K);
  Ext q = joint_short_pair<G>(p1, p2, u, vmag, vb_tbl, K);      // u*(-8A) + |v|*(-+R)
  q = fixed_base_accumulate(q, fb, W, nwin, cw, K);             // + (v s mod l)*B8
  verdict = (fr_is_zero(q.X) && fr_eq(q.Y, q.Z)) ? 1 : 0;       // projective identity (0 : z : z)
  return msg_gt ? 0 : verdict;
}
// Exact path (pk or R off the curve): replays src/lib.rs:395-412 (or :375-385) for every operation whose result
// depends on the formula sequence, and ONLY for those.  The reference computes
//     l = B8.mul_scalar(s),  t = pk.mul_scalar(8 hm),  r = (R + t).affine(),  l == r.
// l never depends on the inputs' being on the curve (B8 is): it comes from the fixed-base table.  t is a canonical
// affine point of the group whenever pk is on the curve -- then any correct evaluation is bit-identical and the
// windowed one is used (scalar 8 (hm mod l), resp. hm for Schnorr: both below 2^254) -- and is replayed bit by bit
// with the reference's unified additions only when pk itself is off the curve.  The final R + t and affine() are
// always the reference's formulas (R may be off the curve; z == 0 -> (0, 0)).  Cost: ~1.3x a fast item when only
// R is off the curve, ~2.1x when pk is (the bit-serial 257-bit multiplication), instead of 4.4x for both.
template <bool SCHNORR>
BJJ_HD int verify_exact_t(const VerifyIn& in, const u32* fb_table, int W, int nwin, u32* vb_tbl, const Consts& K) {
  u32 w[8];
  load_w8(in.msg, w);
  if (words_gt_modulus(w)) return SCHNORR ? 2 : 0;
  Fr h[5];
  h[4] = fr_to_mont_words(w);
  Fr rx, ry, ax, ay;
  load_w8(in.r, w);                    rx = fr_to_mont_words(w);
  load_w8((const char*)in.r + 32, w);  ry = fr_to_mont_words(w);
  load_w8(in.pk, w);                   ax = fr_to_mont_words(w);
  load_w8((const char*)in.pk + 32, w); ay = fr_to_mont_words(w);
  if (SCHNORR) { h[0] = ax; h[1] = ay; h[2] = rx; h[3] = ry; }
  else         { h[0] = rx; h[1] = ry; h[2] = ax; h[3] = ay; }
  Fr hm = poseidon5(h, K);
  Fr hm_plain = fr_canon(fr_mul(hm, fr_one_plain()));
  u32 sw[8];
  load_w8(in.s, sw);
  // l = s * B8 (:405 / :377): fixed-base table, affine on the reference curve (Montgomery form)
  Ext L = fixed_base_mul(GatherPerLane{fb_table}, W, nwin, sw, K);
  Fr tx, ty, zi;
  if (ref_on_curve(ax, ay, K)) {
    // t = (8 hm) * pk = 8 (hm mod l) * pk, resp. hm * pk: windowed, shares one inversion with l
    u32 kw[8];
    if (SCHNORR) {
      fr_to_words(hm_plain, kw);                                // hm < r < 2^254
    } else {
      u32 kp[8];
      fr_to_words(plain_mod_l(hm_plain, K), kp);
      kw[0] = kp[0] << 3;
#pragma unroll
      for (int i = 1; i < 8; i++) kw[i] = (kp[i] << 3) | (kp[i - 1] >> 29);   // 8 (hm mod l) < 8 l < 2^254
    }
    vb_build_table(ext_from_ref_affine(ax, ay, K), vb_tbl, K, true);
    Ext T = vb_mul_windowed(vb_tbl, kw, 64, K);
    zi = fr_inv(fr_mul(L.Z, T.Z));
    const Fr zt = fr_mul(zi, L.Z);
    tx = fr_mul(fr_mul(T.X, zt), K.FINV); ty = fr_mul(T.Y, zt);
    zi = fr_mul(zi, T.Z);
  } else {
    u32 h8[9], hw[8];
    fr_to_words(hm_plain, hw);
    if (SCHNORR) {
#pragma unroll
      for (int i = 0; i < 8; i++) h8[i] = hw[i];
      h8[8] = 0;                                                // pk.mul_scalar(&h), :381
    } else {
      h8[0] = hw[0] << 3;
#pragma unroll
      for (int i = 1; i < 8; i++) h8[i] = (hw[i] << 3) | (hw[i - 1] >> 29);
      h8[8] = hw[7] >> 29;                                      // 8 * hm_b, :410
    }
    ref_mul_scalar(ax, ay, h8, 9, tx, ty, K);
    zi = fr_inv(L.Z);
  }
  const Fr lx = fr_mul(fr_mul(L.X, zi), K.FINV), ly = fr_mul(L.Y, zi);
  RefProj rp; rp.x = rx; rp.y = ry; rp.z = fr_one();
  RefProj tp; tp.x = tx; tp.y = ty; tp.z = fr_one();
  RefProj sum = ref_add(rp, tp, K);                             // :407-410 / :382
  Fr qx, qy;
  if (fr_is_zero(sum.z)) { qx = fr_zero(); qy = fr_zero(); }    // :71-76
  else { Fr z2 = fr_inv(sum.z); qx = fr_mul(sum.x, z2); qy = fr_mul(sum.y, z2); }
  return (fr_eq(lx, qx) && fr_eq(ly, qy)) ? 1 : 0;              // :411 / :384
}
BJJ_HD bool verify_fast(const VerifyIn& in, const u32* fb_table, int W, int nwin, u32* vb_tbl, const Consts& K,
                        bool& need_exact) {
  return verify_fast_t<false>(in, GatherPerLane{fb_table}, W, nwin, vb_tbl, K, need_exact) == 1;
}
BJJ_HD bool verify_exact(const VerifyIn& in, const u32* fb_table, int W, int nwin, u32* vb_tbl, const Consts& K) {
  return verify_exact_t<false>(in, fb_table, W, nwin, vb_tbl, K) == 1;
}
BJJ_HD bool verify_item(const VerifyIn& in, const u32* fb_table, int W, int nwin, u32* vb_tbl, const Consts& K) {
  bool need_exact;
  bool ok = verify_fast(in, fb_table, W, nwin, vb_tbl, K, need_exact);
  return need_exact ? verify_exact(in, fb_table, W, nwin, vb_tbl, K) : ok;
}
BJJ_HD int verify_schnorr_item(const VerifyIn& in, const u32* fb_table, int W, int nwin, u32* vb_tbl, const Consts& K) {
  bool need_exact;
  int v = verify_fast_t<true>(in, GatherPerLane{fb_table}, W, nwin, vb_tbl, K, need_exact);
  return need_exact ? verify_exact_t<true>(in, fb_table, W, nwin, vb_tbl, K) : v;
}

}  // namespace bjj
