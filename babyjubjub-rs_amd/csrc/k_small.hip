// libbjj_hip.so, kernel unit 6: Point::mul_scalar (src/lib.rs:149-164) for SHORT calls -- four lanes per item.
//
// K2 puts one item on one lane: a call of 1 .. a few thousand items costs one lane's serial chain, ~1.15 ms, whatever its size --
// seven times what the reference's own loop takes on one CPU core for ONE item, and a single `p.mul_scalar(&n)` through the
// drop-in is exactly such a call (first table).  The chain is 252 doublings and 63 additions of 7-8 field
// multiplications each, but inside one doubling or addition the multiplications come in TWO levels of four independent ones:
//     addition  (a' = -1, extended + precomputed):  A = (Y-X) ymx   B = (Y+X) ypx   D = Z z2   C = T t2d
//     doubling  (dbl-2008-hwcd):                    A = X^2         B = Y^2         ZZ = Z^2   S = (X+Y)^2
//     both:     X3 = E F   Y3 = G H   Z3 = F G   T3 = E H        with E, F, G, H sums and differences of level one
// Here the lanes 4k .. 4k+3 of a wave hold X, Y, Z, T of item k: every level is ONE multiplication per lane, the operands of the
// next level come from the quad's other lanes by DPP quad permutes (one VALU move per limb, no LDS), and the chain is
// 2 multiplications + ~200 cheap instructions per point operation instead of 7-8 multiplications.  The same formulas with the same
// operand forms as curve.hpp (ext_dbl, ext_add_pn), the same signed 4-bit windows as vb_mul_windowed; the item's table of 0 .. 8 P
// lives in LDS, one component per lane.  The result is the same group element, and the output its canonical affine coordinates:
// byte-identical to K2's (tests/test_gpu_small_calls.py: against K2 and the oracle).
// Total work is the same as K2's (4 lanes x 2 multiplications); what it buys is latency, so the host uses it for calls that do not
// fill the chip one item per lane anyway (bjj_hip.hip: var_base_launch, BJJ_VB_QUAD_MAX).
// Off-curve points: as in K2 -- appended to `slow` for the exact kernel K6, their output slots not touched.
#include "k_common.hpp"

#define BJJ_QUAD_BLOCK 64
#define BJJ_QUAD_ITEMS (BJJ_QUAD_BLOCK / 4)
#define QTBL_ENTRY_WORDS (NL * 4)                  // one table entry of one item: 9 limbs x 4 lanes
#define QTBL_ITEM_WORDS (9 * QTBL_ENTRY_WORDS)     // entries 0 .. 8

// the value lane (4k + SRC) holds, on every lane of quad k
template <int SRC>
__device__ __forceinline__ Fr quad_bcast(const Fr& f) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)f.v[i], SRC * 0x55, 0xf, 0xf, true);   // quad_perm:[SRC,SRC,SRC,SRC]
  return r;
}
// level two of both formulas: lane 0: E F, lane 1: G H, lane 2: F G, lane 3: E H
__device__ __forceinline__ Fr quad_finish(int q, const Fr& e, const Fr& f, const Fr& g, const Fr& h) {
  const Fr u = fr_select(q == 1, g, fr_select(q == 2, f, e));
  const Fr v = fr_select(q == 0, f, fr_select(q == 2, g, h));
  return fr_mul(u, v);
}
// 2P: ext_dbl (curve.hpp), c = this lane's coordinate of P
__device__ __forceinline__ Fr quad_dbl(int q, const Fr& c) {
  const Fr x = quad_bcast<0>(c), y = quad_bcast<1>(c);
  const Fr s = fr_sqr(fr_select(q == 3, fr_add_lazy(x, y), c));   // X^2, Y^2, Z^2, (X+Y)^2
  const Fr a = quad_bcast<0>(s), b = quad_bcast<1>(s), zz = quad_bcast<2>(s), ss = quad_bcast<3>(s);
  const Fr h = fr_add_lazy(a, b);
  const Fr e = fr_sub8_of_lazy(ss, h);
  const Fr g = fr_sub_lazy(b, a);
  const Fr f = fr_sub(fr_add_lazy(fr_add_lazy(zz, zz), a), b);
  return quad_finish(q, e, f, g, h);
}
// P + Q: ext_add_pn (curve.hpp); n = this lane's component of Q's precomputed form (lane 0: Y-X, 1: Y+X, 2: 2Z, 3: 2D'T)
__device__ __forceinline__ Fr quad_add(int q, const Fr& c, const Fr& n) {
  const Fr x = quad_bcast<0>(c), y = quad_bcast<1>(c);
  const Fr op = fr_select(q == 0, fr_sub_lazy(y, x), fr_select(q == 1, fr_add_lazy(y, x), c));
  const Fr m = fr_mul(op, n);
  const Fr a = quad_bcast<0>(m), b = quad_bcast<1>(m), d = quad_bcast<2>(m), cc = quad_bcast<3>(m);
  const Fr e = fr_sub_lazy(b, a);
  const Fr f = fr_sub(d, cc);
  const Fr g = fr_add_lazy(d, cc);
  const Fr h = fr_add_lazy(b, a);
  return quad_finish(q, e, f, g, h);
}
// this lane's component of the precomputed form of P (ext_to_pniels)
__device__ __forceinline__ Fr quad_entry(int q, const Fr& c) {
  const Fr x = quad_bcast<0>(c), y = quad_bcast<1>(c);
  const Fr t = fr_mul(c, c_K.D2P);
  return fr_select(q == 0, fr_sub(y, x), fr_select(q == 1, fr_add(y, x), fr_select(q == 2, fr_dbl(c), t)));
}
__device__ __forceinline__ void qtbl_store(u32* t, int k, int q, const Fr& n) {
#pragma unroll
  for (int i = 0; i < NL; i++) t[k * QTBL_ENTRY_WORDS + i * 4 + q] = n.v[i];
}
// entry |d| with the sign of d: -(x, y) = (-x, y) swaps Y-X and Y+X (lane 0 reads lane 1's words and the other way round) and negates 2D'T
__device__ __forceinline__ Fr qtbl_load(const u32* t, int d, int q) {
  const bool neg = d < 0;
  const int k = neg ? -d : d;
  const int qq = (neg && q < 2) ? (q ^ 1) : q;
  Fr n;
#pragma unroll
  for (int i = 0; i < NL; i++) n.v[i] = t[k * QTBL_ENTRY_WORDS + i * 4 + qq];
  return fr_select(neg && q == 3, fr_sub_lazy(fr_zero(), n), n);
}

__global__ void __launch_bounds__(BJJ_QUAD_BLOCK) bjj_k_mul_var_base_quad(const uint8_t* __restrict__ pts, const uint8_t* __restrict__ scalars,
                                                                      size_t n, uint8_t* __restrict__ out, u32* __restrict__ slow) {
  constexpr u32 R1[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  __shared__ u32 tbl_all[BJJ_QUAD_ITEMS * QTBL_ITEM_WORDS];   // 20.7 KB
  const int lane = threadIdx.x, q = lane & 3;
  u32* tbl = tbl_all + (lane >> 2) * QTBL_ITEM_WORDS;
  const size_t item = (size_t)blockIdx.x * BJJ_QUAD_ITEMS + (size_t)(lane >> 2);
  const bool live = item < n;
  const size_t i = live ? item : n - 1;          // a quad beyond the batch repeats the last item and stores nothing
  u32 w[8], sc[8], red[8];
  load_w8(pts + i * 64, w);      Fr x = fr_to_mont_words(w);
  load_w8(pts + i * 64 + 32, w); Fr y = fr_to_mont_words(w);
  const bool on = ref_on_curve(x, y, c_K);
  if (live && !on && q == 0 && slow) slow[8 + atomicAdd(&slow[0], 1u)] = (u32)i;   // K6's item
  x = fr_select(on, x, fr_zero());               // ... and this quad walks the identity instead: control flow stays uniform
  y = fr_select(on, y, fr_one());
  load_w8(scalars + i * 32, sc);
  scalar_mod_order(sc, red, c_K);
  // P on the a' = -1 curve (ext_from_ref_affine), one coordinate per lane
  const Fr X = fr_mul(x, c_K.F);
  const Fr T = fr_mul(X, y);
  Fr c = fr_select(q == 0, X, fr_select(q == 1, y, fr_select(q == 2, fr_one(), T)));
  // table 0 .. 8 P (vb_build_table)
  {
    const PNiels id = pniels_identity();
    qtbl_store(tbl, 0, q, fr_select(q == 0, id.ymx, fr_select(q == 1, id.ypx, fr_select(q == 2, id.z2, id.t2d))));
    const Fr p1 = quad_entry(q, c);
    qtbl_store(tbl, 1, q, p1);
    Fr cur = c;
#pragma unroll 1
    for (int k = 2; k <= 8; k++) {
      cur = quad_add(q, cur, p1);
      qtbl_store(tbl, k, q, quad_entry(q, cur));
    }
  }
  __syncthreads();                               // one wave: orders the table stores before the loads of the quad's other lanes
  // signed 4-bit windows, most significant first (vb_mul_windowed)
  u32 t[8];
  {
    u64 cy = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { cy += (u64)red[k] + 0x88888888u; t[k] = (u32)cy; cy >>= 32; }
  }
  {  // top window: the entry itself as a point (pniels_to_ext; T is not needed: doublings follow)
    const int d = (int)((t[7] >> 28) & 15u) - 8;
    const Fr e = qtbl_load(tbl, d, q);
    const Fr ymx = quad_bcast<0>(e), ypx = quad_bcast<1>(e);
    c = fr_select(q == 0, fr_reduce_weak(fr_sub8(ypx, ymx)), fr_select(q == 1, fr_reduce_weak(fr_add(ypx, ymx)), fr_select(q == 2, fr_reduce_weak(e), fr_zero())));
  }
#pragma unroll 1
  for (int j = 62; j >= 0; j--) {
    const int d = (int)((t[j >> 3] >> ((j & 7) * 4)) & 15u) - 8;
    const Fr e = qtbl_load(tbl, d, q);           // issued ahead of the doublings
#pragma unroll 1
    for (int k = 0; k < 4; k++) c = quad_dbl(q, c);
    c = quad_add(q, c, e);
  }
  // affine, back on the reference curve, canonical (epilogue_finish): x = X / (Z F), y = Y / Z
  const Fr zi = fr_mul(fr_inv(quad_bcast<2>(c)), fr_one_plain());          // plain 1/Z
  const Fr m = fr_select(q == 0, fr_mul(zi, c_K.FINV), zi);
  const Fr v = fr_cond_sub_kr(fr_mul(c, m), R1);
  if (live && on && q < 2) {
    fr_to_words(v, w);
    store_w8(out + i * 64 + (size_t)q * 32, w);
  }
}

namespace bjjk {
// slow: the list K6 reads (reset by the caller), or nullptr when somebody else has made it (the scan of the split form)
hipError_t mul_var_base_quad(hipStream_t st, const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out, u32* slow) {
  const size_t grid = (n + BJJ_QUAD_ITEMS - 1) / BJJ_QUAD_ITEMS;
  BJJ_LAUNCH(bjj_k_mul_var_base_quad, dim3((unsigned)(grid ? grid : 1)), dim3(BJJ_QUAD_BLOCK), 0, st, pts, scalars, n, out, slow);
  return hipGetLastError();
}
}  // namespace bjjk

// =====================================================================================================================================
// Poseidon t = 6 (src/lib.rs:400-404; poseidon.hpp) for SHORT calls: six lanes per hash (groups of eight lanes, two of them spares).
//
// One lane per hash is ~1 000 multiplication-equivalents in a row, 0.51 ms, whatever the call's size: six times what the CPU needs
// for ONE hash.  The dependent chain of the permutation is much shorter: in a (sparse-form) partial round only
//     x0 = (st0 + k)^5                        3 multiplications
//     st0' = m00 x0 + (v . st[1..5])          1 multiplication once x0 is there -- the dot product over the OLD st[1..5] does not wait for x0
// are in sequence; the five updates st_j += what_j x0 and the five products of the dot product are independent of one another.  Lane j of a
// group owns st_j: the round is FOUR multiplication slots for the whole group -- slot 1: lane 0 squares, lanes 1..5 form v_j st_j; slots 2, 3:
// lane 0 finishes x0; x0 is broadcast, the five products are summed across the lanes (butterfly); slot 4: lane 0 m00 x0, lanes 1..5 what_j x0.
// A full round is three slots for the six S-boxes side by side, then every lane gathers the state and forms ITS row of M . st.
// The same constants (bjj_constants.inc: the sparse form gen_tables.py derives) and the same field elements round by round as poseidon5_t<false>,
// so the hash is the same canonical integer: tests/test_gpu_small_calls.py, against bjj_k_poseidon5 and the oracle.
// =====================================================================================================================================
#define BJJ_P5C_BLOCK 64
#define BJJ_P5C_GROUP 8
__device__ __forceinline__ Fr grp_get(const Fr& f, int k) {          // the value lane k of this lane's group holds
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = (u32)__shfl((int)f.v[i], k, BJJ_P5C_GROUP);
  return r;
}
__device__ __forceinline__ Fr grp_sum(Fr f) {                        // sum over the group, on every lane (carried; <= 8 terms of < 2r)
#pragma unroll
  for (int m = 1; m < BJJ_P5C_GROUP; m <<= 1) {
    Fr o;
#pragma unroll
    for (int i = 0; i < NL; i++) o.v[i] = (u32)__shfl_xor((int)f.v[i], m, BJJ_P5C_GROUP);
    f = fr_add(f, o);
  }
  return f;
}
__device__ __forceinline__ Fr p5c_const(const Fr* base, int idx) {   // a per-lane element of the constant block
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = base[idx].v[i];
  return r;
}
// S-boxes side by side, then this lane's row of M . state.  LAST: only row 0 is needed (the hash is element 0)
__device__ __forceinline__ Fr p5c_full_round(int j, const Fr& mine, int r) {
  const Fr x = fr_pow5(fr_add(mine, p5c_const(c_K.PCF, r * 6 + j)));
  Fr s[6], row[6];
#pragma unroll
  for (int k = 0; k < 6; k++) { s[k] = grp_get(x, k); row[k] = p5c_const(c_K.PM, j * 6 + k); }
  return fr_dot<6>(row, s);
}

// the permutation over one group: lane j (0 .. 5; the spare lanes pass j = 5) brings element j of the state [0, in0 .. in4] (Montgomery, < 2r);
// returns the final state's element of this lane -- the hash is lane 0's
__device__ __forceinline__ Fr p5c_permute(int j, int gl, Fr st) {
#pragma unroll 1
  for (int r = 0; r < 4; r++) st = p5c_full_round(j, st, r);
  // 60 partial rounds, sparse form, one at a time (poseidon5_t<false>); the constants of round p + 1 are fetched while round p runs
  const int i1 = j, i2 = j ? 5 + j : 0;
  Fr c1 = p5c_const(c_K.PSP, i1), c2 = p5c_const(c_K.PSP, i2);
#pragma unroll 1
  for (int p = 0; p < 60; p++) {
    const int pn = p < 59 ? p + 1 : 59;
    const Fr c1n = p5c_const(c_K.PSP, pn * 11 + i1), c2n = p5c_const(c_K.PSP, pn * 11 + i2);
    const Fr a = fr_add(st, c_K.PKP[p]);                             // (lane 0's is the one that counts)
    const Fr r1 = fr_mul(fr_select(j == 0, a, c1), fr_select(j == 0, a, st));       // lane 0: a^2       lanes 1..5: v_j st_j
    const Fr t2 = fr_mul(r1, r1);                                                     // lane 0: a^4
    const Fr x0 = grp_get(fr_mul(t2, a), 0);                                          // lane 0: a^5, to every lane
    const Fr V = grp_sum(fr_select(gl >= 1 && gl <= 5, r1, fr_zero()));               // v . st[1..5] (the old elements)
    const Fr m = fr_mul(c2, x0);                                                      // lane 0: m00 x0    lanes 1..5: what_j x0
    st = fr_select(j == 0, fr_reduce_weak(fr_add(m, V)), fr_add(st, m));
    if ((p & 3) == 3) st = fr_reduce_weak(st);                        // the lazily growing elements stay below ~10 r
    c1 = c1n; c2 = c2n;
  }
  {  // diag(1, A_last): lanes 1..5 take their row of the 5 x 5 block over st[1..5]
    Fr s[5], row[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { s[k] = grp_get(st, k + 1); row[k] = p5c_const(c_K.PAL, (j ? j - 1 : 0) * 5 + k); }
    const Fr d = fr_dot<5>(row, s);
    st = fr_select(j == 0, st, d);
  }
#pragma unroll 1
  for (int r = 4; r < 8; r++) st = p5c_full_round(j, st, r);         // (round 7: only lane 0's row is the hash)
  return st;
}

__global__ void __launch_bounds__(BJJ_P5C_BLOCK) bjj_k_poseidon5_coop(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
  const int lane = threadIdx.x, gl = lane & (BJJ_P5C_GROUP - 1);
  const int j = gl < 6 ? gl : 5;                                     // the spare lanes shadow lane 5 (their values are never read)
  const size_t item = (size_t)blockIdx.x * (BJJ_P5C_BLOCK / BJJ_P5C_GROUP) + (size_t)(lane >> 3);
  const bool live = item < n;
  const size_t i = live ? item : n - 1;
  u32 w[8];
  load_w8(in + i * 160 + (size_t)(j ? j - 1 : 0) * 32, w);
  const Fr st = p5c_permute(j, gl, fr_select(j == 0, fr_zero(), fr_to_mont_words(w)));   // state = [0, in0 .. in4]
  if (live && gl == 0) {
    fr_from_mont_words(st, w);
    store_w8(out + i * 32, w);
  }
}

// =====================================================================================================================================
// verify(pk, sig, msg) (src/lib.rs:395-412) for SHORT calls: eight lanes per signature.
//
// The fast path of K4 (bjj_device.hpp: verify_fast_t<false>) with its two serial parts spread over lanes: the hash runs on six of the group's
// lanes (p5c_permute), and  u (-8A) + |v| (-+R) + (v s mod l) B8 == O  -- the short odd pair (u, v) of the lattice step, the joint windowed loop, then
// the fixed-base windows -- runs with X, Y, Z, T of the accumulator on the four lanes of a quad (both quads of the group compute it; the lower one's
// verdict is stored).  Same scalars, same tables of 0 .. 8 P, same windows, same table entries of B8 as the lane form: the same group element, the
// same verdict.  An item whose pk or R is off the curve is not this kernel's: the scan has put it on the list and the exact launch owns its
// verdict, as with K4's bulk workgroups (k_verify.hip).
// =====================================================================================================================================
#define BJJ_VS_BLOCK 64
__device__ __forceinline__ void qtbl_build(int q, u32* tbl, const Fr& c) {    // 0 .. 8 P for P = c (any Z): vb_build_table
  const PNiels id = pniels_identity();
  qtbl_store(tbl, 0, q, fr_select(q == 0, id.ymx, fr_select(q == 1, id.ypx, fr_select(q == 2, id.z2, id.t2d))));
  const Fr p1 = quad_entry(q, c);
  qtbl_store(tbl, 1, q, p1);
  Fr cur = c;
#pragma unroll 1
  for (int k = 2; k <= 8; k++) {
    cur = quad_add(q, cur, p1);
    qtbl_store(tbl, k, q, quad_entry(q, cur));
  }
}
// this lane's component of entry `slot` of the fixed-base table (load_niels: Y-X words 0..8, Y+X 9..17, 2D'xy 18..26; Z = 1: 2Z = 2), with the digit's sign
__device__ __forceinline__ Fr qfb_load(const u32* __restrict__ table, size_t slot, bool neg, int q) {
  const int qq = (neg && q < 2) ? (q ^ 1) : q;
  const u32* e = table + slot * NIELS_WORDS + (qq == 3 ? 18 : qq * 9);
  Fr n;
#pragma unroll
  for (int i = 0; i < NL; i++) n.v[i] = e[i];
  n = fr_select(q == 2, fr_dbl(fr_one()), n);
  return fr_select(neg && q == 3, fr_sub_lazy(fr_zero(), n), n);
}
__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
  return v;
}

// Touch the table entries a fixed-base multiplication by sc (< l) is going to read -- one word of each, all loads in flight together -- so that the gathers
// of the multiplication itself, which the compiler sinks to their first use one after the other, find their lines and page-table entries in the caches:
// a lone lane's gather from the 5.9 GB (or 155 GB) table costs several microseconds when it misses the TLB, 11 to 22 of them in a row per item.
__device__ __forceinline__ void qfb_prefetch(const u32* __restrict__ table, int W, int nwin, const u32 sc[8]) {
  DigitStream ds = digit_stream(sc, W);
  u32 v[32];                                     // (tables of fewer than 32 windows: W >= 8; narrower ones are prefetched in part)
#pragma unroll
  for (int k = 0; k < 32; k++) {
    bool neg;
    const size_t slot = digit_next(ds, neg);
    v[k] = k < nwin ? table[slot * NIELS_WORDS] : 0u;
  }
  u32 sink = 0;
#pragma unroll
  for (int k = 0; k < 32; k++) sink ^= v[k];
  asm volatile("" ::"v"(sink));
}
// TWO waves per workgroup share the curve arithmetic of their eight signatures (they are on different SIMDs: truly side by side, not in lock-step): both hash and
// derive the scalars (the same work at the same time costs no time), then wave 0 builds the table of -8A and walks u over it, wave 1 builds the table of -+R, walks |v|
// over it and adds the fixed-base windows; wave 1's sum goes through LDS to wave 0, which adds it to its own and tests for the identity.  One scalar per wave instead of
// two in a joint loop: 128 doublings + 33 additions in sequence instead of 128 + 66, 11 table operations instead of 19 (profiles/r06_small_calls.txt).
__global__ void __launch_bounds__(2 * BJJ_VS_BLOCK) bjj_k_eddsa_verify_small(const u32* __restrict__ table, int W, int nwin,
                                                                            const uint8_t* __restrict__ pk, const uint8_t* __restrict__ rb8,
                                                                            const uint8_t* __restrict__ sg, const uint8_t* __restrict__ msg, size_t n,
                                                                            uint8_t* __restrict__ ok) {
  __shared__ u32 tbl_all[2 * (BJJ_VS_BLOCK / 4) * QTBL_ITEM_WORDS];    // one table per quad and wave: 41.5 KB
  __shared__ u32 xch[(BJJ_VS_BLOCK / 4) * 4 * NL];                     // wave 1's sum, one coordinate per lane
  const int role = threadIdx.x >> 6, lane = threadIdx.x & 63, gl = lane & 7, q = lane & 3;
  const int j = gl < 6 ? gl : 5;
  u32* tbl = tbl_all + (role * (BJJ_VS_BLOCK / 4) + (lane >> 2)) * QTBL_ITEM_WORDS;
  u32* mine = xch + ((lane >> 2) * 4 + q) * NL;
  const size_t item = (size_t)blockIdx.x * (BJJ_VS_BLOCK / 8) + (size_t)(lane >> 3);
  const bool live = item < n;
  const size_t i = live ? item : n - 1;
  u32 w[8];
  load_w8(msg + i * 32, w);
  const bool msg_gt = words_gt_modulus(w);                           // :396-398
  const Fr m5 = fr_to_mont_words(w);
  load_w8(rb8 + i * 64, w);      const Fr rx = fr_to_mont_words(w);
  load_w8(rb8 + i * 64 + 32, w); const Fr ry = fr_to_mont_words(w);
  load_w8(pk + i * 64, w);       const Fr ax = fr_to_mont_words(w);
  load_w8(pk + i * 64 + 32, w);  const Fr ay = fr_to_mont_words(w);
  const bool need_exact = !msg_gt && !(ref_on_curve(rx, ry, c_K) && ref_on_curve(ax, ay, c_K));
  // hm = H(R.x, R.y, A.x, A.y, msg) (:400-404), six lanes
  const Fr st0 = fr_select(j == 0, fr_zero(), fr_select(j == 1, rx, fr_select(j == 2, ry, fr_select(j == 3, ax, fr_select(j == 4, ay, m5)))));
  const Fr hm = grp_get(p5c_permute(j, gl, st0), 0);
  const Fr hm_plain = fr_canon(fr_mul(hm, fr_one_plain()));
  // the short odd pair and the fixed-base scalar (verify_fast_t): every lane for itself
  Fr u, vmag;
  bool vneg;
  lattice_short_pair(plain_mod_l(hm_plain, c_K), u, vmag, vneg, c_K);
  u32 sw[8], cw[8];
  load_w8(sg + i * 32, sw);
  {
    const Fr sl = fl_mul(fr_from_words(sw), c_K.L_R1, c_K);
    Fr c = fl_canon4(fl_mul(vmag, fl_mul(sl, c_K.L_R2, c_K), c_K), c_K);
    if (vneg && !limbs_is_zero(c)) { Fr t = c_K.L; limbs_submul(t, 1u, c); c = t; }
    fr_to_words(c, cw);
  }
  if (role) qfb_prefetch(table, W, nwin, cw);
  // this wave's point on the a' = -1 curve, one coordinate per lane, and its table: wave 0 -8A, wave 1 -sign(v) R
  {
    const Fr px = role ? fr_select(vneg, rx, fr_neg(rx)) : fr_neg(ax), py = role ? ry : ay;
    const Fr X = fr_mul(px, c_K.F);
    Fr c = fr_select(q == 0, X, fr_select(q == 1, py, fr_select(q == 2, fr_one(), fr_mul(X, py))));
    if (!role) {
#pragma unroll 1
      for (int k = 0; k < 3; k++) c = quad_dbl(q, c);
    }
    qtbl_build(q, tbl, c);
  }
  __syncthreads();
  // this wave's scalar over its table, as many windows as the widest pair of the wave needs (joint_short_pair: the same count for both waves)
  const int ub = limbs_bits(u), vb = limbs_bits(vmag);
  const int mb = ub > vb ? ub : vb;
  const int need = mb <= 2 ? 1 : (mb + 5) >> 2;
  const int jw = wave_max_int(need > 64 ? 64 : need);
  u32 t[8];
  recode_signed4(role ? vmag : u, t);
  Fr acc;
  {
    const int jj = jw - 1;
    const int d = (int)((t[jj >> 3] >> ((jj & 7) * 4)) & 15u) - 8;
    const Fr e = qtbl_load(tbl, d, q);
    const Fr ymx = quad_bcast<0>(e), ypx = quad_bcast<1>(e);         // pniels_to_ext with T (additions may follow at once): 2T = (2D'T) / D'
    acc = fr_select(q == 0, fr_reduce_weak(fr_sub8(ypx, ymx)), fr_select(q == 1, fr_reduce_weak(fr_add(ypx, ymx)),
                    fr_select(q == 2, fr_reduce_weak(e), fr_mul(e, c_K.DPINV))));
  }
#pragma unroll 1
  for (int jj = jw - 2; jj >= 0; jj--) {
    const int d = (int)((t[jj >> 3] >> ((jj & 7) * 4)) & 15u) - 8;
    const Fr e = qtbl_load(tbl, d, q);
#pragma unroll 1
    for (int k = 0; k < 4; k++) acc = quad_dbl(q, acc);
    acc = quad_add(q, acc, e);
  }
  if (role) {  // + (v s mod l) B8: the fixed-base windows (fixed_base_accumulate); window 0 is stored in T form
    DigitStream ds = digit_stream(cw, W);
    bool neg;
    size_t slot = digit_next(ds, neg);
    Fr cur = qfb_load(table, slot, neg, q);
    cur = fr_select(q == 3, fr_mul(cur, c_K.DP), cur);
#pragma unroll 1
    for (int k = 0; k + 1 < nwin; k++) {
      slot = digit_next(ds, neg);
      const Fr nxt = qfb_load(table, slot, neg, q);                   // in flight during this window's addition
      acc = quad_add(q, acc, cur);
      cur = nxt;
    }
    acc = quad_add(q, acc, cur);
#pragma unroll
    for (int k = 0; k < NL; k++) mine[k] = acc.v[k];
  }
  __syncthreads();
  if (!role) {
    Fr other;
#pragma unroll
    for (int k = 0; k < NL; k++) other.v[k] = mine[k];
    acc = quad_add(q, acc, quad_entry(q, other));
    // the projective identity (0 : z : z)
    const Fr X = quad_bcast<0>(acc), Y = quad_bcast<1>(acc), Z = quad_bcast<2>(acc);
    const int verdict = (fr_is_zero(X) && fr_eq(Y, Z)) ? 1 : 0;
    if (live && !need_exact && gl == 0) ok[i] = (uint8_t)(msg_gt ? 0 : verdict);
  }
}

// ONE wave per eight signatures, both scalars in a joint loop (joint_mul_windowed): eight lanes per signature instead of sixteen -- the form for calls that put more than a
// wave or two on every CU (2^12 ... 2^13 signatures), where lanes are what is scarce: 643 us per call at 2^12 against 897 for the two-wave form, which wins below (587 vs 627 at 2^11).
__global__ void __launch_bounds__(BJJ_VS_BLOCK) bjj_k_eddsa_verify_small_joint(const u32* __restrict__ table, int W, int nwin,
                                                                        const uint8_t* __restrict__ pk, const uint8_t* __restrict__ rb8,
                                                                        const uint8_t* __restrict__ sg, const uint8_t* __restrict__ msg, size_t n,
                                                                        uint8_t* __restrict__ ok) {
  __shared__ u32 tbl_all[(BJJ_VS_BLOCK / 4) * 2 * QTBL_ITEM_WORDS];    // two tables per quad: 41.5 KB
  const int lane = threadIdx.x, gl = lane & 7, q = lane & 3;
  const int j = gl < 6 ? gl : 5;
  u32* tbl1 = tbl_all + (lane >> 2) * 2 * QTBL_ITEM_WORDS;
  u32* tbl2 = tbl1 + QTBL_ITEM_WORDS;
  const size_t item = (size_t)blockIdx.x * (BJJ_VS_BLOCK / 8) + (size_t)(lane >> 3);
  const bool live = item < n;
  const size_t i = live ? item : n - 1;
  u32 w[8];
  load_w8(msg + i * 32, w);
  const bool msg_gt = words_gt_modulus(w);                           // :396-398
  const Fr m5 = fr_to_mont_words(w);
  load_w8(rb8 + i * 64, w);      const Fr rx = fr_to_mont_words(w);
  load_w8(rb8 + i * 64 + 32, w); const Fr ry = fr_to_mont_words(w);
  load_w8(pk + i * 64, w);       const Fr ax = fr_to_mont_words(w);
  load_w8(pk + i * 64 + 32, w);  const Fr ay = fr_to_mont_words(w);
  const bool need_exact = !msg_gt && !(ref_on_curve(rx, ry, c_K) && ref_on_curve(ax, ay, c_K));
  // hm = H(R.x, R.y, A.x, A.y, msg) (:400-404), six lanes
  const Fr st0 = fr_select(j == 0, fr_zero(), fr_select(j == 1, rx, fr_select(j == 2, ry, fr_select(j == 3, ax, fr_select(j == 4, ay, m5)))));
  const Fr hm = grp_get(p5c_permute(j, gl, st0), 0);
  const Fr hm_plain = fr_canon(fr_mul(hm, fr_one_plain()));
  // the short odd pair and the fixed-base scalar (verify_fast_t): every lane for itself
  Fr u, vmag;
  bool vneg;
  lattice_short_pair(plain_mod_l(hm_plain, c_K), u, vmag, vneg, c_K);
  u32 sw[8], cw[8];
  load_w8(sg + i * 32, sw);
  {
    const Fr sl = fl_mul(fr_from_words(sw), c_K.L_R1, c_K);
    Fr c = fl_canon4(fl_mul(vmag, fl_mul(sl, c_K.L_R2, c_K), c_K), c_K);
    if (vneg && !limbs_is_zero(c)) { Fr t = c_K.L; limbs_submul(t, 1u, c); c = t; }
    fr_to_words(c, cw);
  }
  qfb_prefetch(table, W, nwin, cw);
  // P1 = -8A, P2 = -sign(v) R on the a' = -1 curve, one coordinate per lane; their tables
  {
    const Fr X = fr_mul(fr_neg(ax), c_K.F);
    Fr c = fr_select(q == 0, X, fr_select(q == 1, ay, fr_select(q == 2, fr_one(), fr_mul(X, ay))));
#pragma unroll 1
    for (int k = 0; k < 3; k++) c = quad_dbl(q, c);
    qtbl_build(q, tbl1, c);
  }
  {
    const Fr X = fr_mul(fr_select(vneg, rx, fr_neg(rx)), c_K.F);
    const Fr c = fr_select(q == 0, X, fr_select(q == 1, ry, fr_select(q == 2, fr_one(), fr_mul(X, ry))));
    qtbl_build(q, tbl2, c);
  }
  __syncthreads();
  // the joint windowed loop (joint_short_pair / joint_mul_windowed): as many windows as the widest pair of the wave needs
  const int ub = limbs_bits(u), vb = limbs_bits(vmag);
  const int mb = ub > vb ? ub : vb;
  const int need = mb <= 2 ? 1 : (mb + 5) >> 2;
  const int jw = wave_max_int(need > 64 ? 64 : need);
  u32 tu[8], tv[8];
  recode_signed4(u, tu);
  recode_signed4(vmag, tv);
  Fr acc;
  {
    const int jj = jw - 1;
    const int du = (int)((tu[jj >> 3] >> ((jj & 7) * 4)) & 15u) - 8;
    const int dv = (int)((tv[jj >> 3] >> ((jj & 7) * 4)) & 15u) - 8;
    const Fr e1 = qtbl_load(tbl1, du, q), e2 = qtbl_load(tbl2, dv, q);
    const Fr ymx = quad_bcast<0>(e1), ypx = quad_bcast<1>(e1);       // pniels_to_ext with T (an addition follows): 2T = (2D'T) / D'
    acc = fr_select(q == 0, fr_reduce_weak(fr_sub8(ypx, ymx)), fr_select(q == 1, fr_reduce_weak(fr_add(ypx, ymx)),
                    fr_select(q == 2, fr_reduce_weak(e1), fr_mul(e1, c_K.DPINV))));
    acc = quad_add(q, acc, e2);
  }
#pragma unroll 1
  for (int jj = jw - 2; jj >= 0; jj--) {
    const int du = (int)((tu[jj >> 3] >> ((jj & 7) * 4)) & 15u) - 8;
    const int dv = (int)((tv[jj >> 3] >> ((jj & 7) * 4)) & 15u) - 8;
    const Fr e1 = qtbl_load(tbl1, du, q), e2 = qtbl_load(tbl2, dv, q);
#pragma unroll 1
    for (int k = 0; k < 4; k++) acc = quad_dbl(q, acc);
    acc = quad_add(q, acc, e1);
    acc = quad_add(q, acc, e2);
  }
  // + (v s mod l) B8: the fixed-base windows (fixed_base_accumulate); window 0 is stored in T form
  {
    DigitStream ds = digit_stream(cw, W);
    bool neg;
    size_t slot = digit_next(ds, neg);
    Fr cur = qfb_load(table, slot, neg, q);
    cur = fr_select(q == 3, fr_mul(cur, c_K.DP), cur);
#pragma unroll 1
    for (int k = 0; k + 1 < nwin; k++) {
      slot = digit_next(ds, neg);
      const Fr nxt = qfb_load(table, slot, neg, q);                   // in flight during this window's addition
      acc = quad_add(q, acc, cur);
      cur = nxt;
    }
    acc = quad_add(q, acc, cur);
  }
  // the projective identity (0 : z : z)
  const Fr X = quad_bcast<0>(acc), Y = quad_bcast<1>(acc), Z = quad_bcast<2>(acc);
  const int verdict = (fr_is_zero(X) && fr_eq(Y, Z)) ? 1 : 0;
  if (live && !need_exact && gl == 0) ok[i] = (uint8_t)(msg_gt ? 0 : verdict);
}

// verify_schnorr (src/lib.rs:364-385) for SHORT calls: the fast path of verify_fast_t<true> on eight lanes -- hash input order (pk, R, msg), the hash NOT multiplied by 8,
// hm (-A) over 64 signed windows with the accumulator on a quad, + (s mod l) B8, compared with R on the a' = -1 curve; verdict 2 = Err (msg > Q).
__global__ void __launch_bounds__(BJJ_VS_BLOCK) bjj_k_schnorr_verify_small(const u32* __restrict__ table, int W, int nwin,
                                                                          const uint8_t* __restrict__ pk, const uint8_t* __restrict__ rb8,
                                                                          const uint8_t* __restrict__ sg, const uint8_t* __restrict__ msg, size_t n,
                                                                          uint8_t* __restrict__ ok) {
  __shared__ u32 tbl_all[(BJJ_VS_BLOCK / 4) * QTBL_ITEM_WORDS];
  const int lane = threadIdx.x, gl = lane & 7, q = lane & 3;
  const int j = gl < 6 ? gl : 5;
  u32* tbl = tbl_all + (lane >> 2) * QTBL_ITEM_WORDS;
  const size_t item = (size_t)blockIdx.x * (BJJ_VS_BLOCK / 8) + (size_t)(lane >> 3);
  const bool live = item < n;
  const size_t i = live ? item : n - 1;
  u32 w[8];
  load_w8(msg + i * 32, w);
  const bool msg_gt = words_gt_modulus(w);                           // :365-367
  const Fr m5 = fr_to_mont_words(w);
  load_w8(rb8 + i * 64, w);      const Fr rx = fr_to_mont_words(w);
  load_w8(rb8 + i * 64 + 32, w); const Fr ry = fr_to_mont_words(w);
  load_w8(pk + i * 64, w);       const Fr ax = fr_to_mont_words(w);
  load_w8(pk + i * 64 + 32, w);  const Fr ay = fr_to_mont_words(w);
  const bool need_exact = !msg_gt && !(ref_on_curve(rx, ry, c_K) && ref_on_curve(ax, ay, c_K));
  const Fr st0 = fr_select(j == 0, fr_zero(), fr_select(j == 1, ax, fr_select(j == 2, ay, fr_select(j == 3, rx, fr_select(j == 4, ry, m5)))));   // :369
  const Fr hm = grp_get(p5c_permute(j, gl, st0), 0);
  u32 kw[8], sw[8], sl[8];
  fr_to_words(fr_canon(fr_mul(hm, fr_one_plain())), kw);             // hm < r < 2^254: no reduction
  load_w8(sg + i * 32, sw);
  scalar_mod_l(sw, sl, c_K);                                         // B8 has order l
  qfb_prefetch(table, W, nwin, sl);
  {
    const Fr X = fr_mul(fr_neg(ax), c_K.F);                          // -A on the a' = -1 curve
    qtbl_build(q, tbl, fr_select(q == 0, X, fr_select(q == 1, ay, fr_select(q == 2, fr_one(), fr_mul(X, ay)))));
  }
  __syncthreads();
  u32 t[8];
  {
    u64 cy = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { cy += (u64)kw[k] + 0x88888888u; t[k] = (u32)cy; cy >>= 32; }
  }
  Fr acc;
  {
    const int d = (int)((t[7] >> 28) & 15u) - 8;
    const Fr e = qtbl_load(tbl, d, q);
    const Fr ymx = quad_bcast<0>(e), ypx = quad_bcast<1>(e);
    acc = fr_select(q == 0, fr_reduce_weak(fr_sub8(ypx, ymx)), fr_select(q == 1, fr_reduce_weak(fr_add(ypx, ymx)), fr_select(q == 2, fr_reduce_weak(e), fr_zero())));
  }
#pragma unroll 1
  for (int jj = 62; jj >= 0; jj--) {
    const int d = (int)((t[jj >> 3] >> ((jj & 7) * 4)) & 15u) - 8;
    const Fr e = qtbl_load(tbl, d, q);
#pragma unroll 1
    for (int k = 0; k < 4; k++) acc = quad_dbl(q, acc);
    acc = quad_add(q, acc, e);
  }
  {  // + (s mod l) B8 (:377)
    DigitStream ds = digit_stream(sl, W);
    bool neg;
    size_t slot = digit_next(ds, neg);
    Fr cur = qfb_load(table, slot, neg, q);
    cur = fr_select(q == 3, fr_mul(cur, c_K.DP), cur);
#pragma unroll 1
    for (int k = 0; k + 1 < nwin; k++) {
      slot = digit_next(ds, neg);
      const Fr nxt = qfb_load(table, slot, neg, q);
      acc = quad_add(q, acc, cur);
      cur = nxt;
    }
    acc = quad_add(q, acc, cur);
  }
  const Fr X = quad_bcast<0>(acc), Y = quad_bcast<1>(acc), Z = quad_bcast<2>(acc);
  const int verdict = (fr_eq(X, fr_mul(fr_mul(rx, c_K.F), Z)) && fr_eq(Y, fr_mul(ry, Z))) ? 1 : 0;
  if (live && !need_exact && gl == 0) ok[i] = (uint8_t)(msg_gt ? 2 : verdict);
}

// =====================================================================================================================================
// PrivateKey::sign (src/lib.rs:308-342) for SHORT calls: eight lanes per signature.  sign_item (sign.hpp) as it is -- the two Blake-512 digests, r, the two
// fixed-base multiplications, one inversion -- on every lane of the group for itself, and the Poseidon hash, two thirds of the lane form's time, on six of them
// (p5c_permute).  Same field elements, same outputs; the constant-time signer option keeps its own kernels.  C64: Signature::compress (k_sign.hip).
// =====================================================================================================================================
template <bool C64>
__device__ __forceinline__ void sign_small_body(const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ keys,
                                                const uint8_t* __restrict__ msgs, size_t n, uint8_t* __restrict__ out_r, uint8_t* __restrict__ out_s,
                                                uint8_t* __restrict__ ok) {
  const int lane = threadIdx.x, gl = lane & 7;
  const int j = gl < 6 ? gl : 5;
  const size_t item = (size_t)blockIdx.x * (BJJ_VS_BLOCK / 8) + (size_t)(lane >> 3);
  const bool live = item < n;
  const size_t i = live ? item : n - 1;
  u32 key[8], msg[8];
  load_w8(keys + i * 32, key); load_w8(msgs + i * 32, msg);
  const bool good = !words_gt_modulus(msg);                      // :309-311
  u32 sk[8], pruned[8], buf[16], dig[16];
  scalar_key_words(key, sk, pruned, buf);                        // :316
#pragma unroll
  for (int k = 0; k < 8; k++) buf[8 + k] = msg[k];               // :318-325
  blake512_words(buf, 16, dig);                                  // :326
  const Fr r = fl_canon4(fr_add(fl_mul(limbs_from_bits(dig, 16, 0), c_K.L_R1, c_K), fl_mul(limbs_from_bits(dig, 16, 261), c_K.L_R2, c_K)), c_K);   // :327-328
  u32 rw[8];
  fr_to_words(r, rw);
  const GatherPerLane fb = {table};
  {
    u32 rl[8], sl[8];
    scalar_mod_l(rw, rl, c_K); scalar_mod_l(sk, sl, c_K);        // (what fixed_base_mul reduces its scalar to)
    qfb_prefetch(table, W, nwin, rl); qfb_prefetch(table, W, nwin, sl);
  }
  const Ext Rp = fixed_base_mul(fb, W, nwin, rw, c_K);           // :329
  const Ext Ap = fixed_base_mul(fb, W, nwin, sk, c_K);           // :330
  const Fr zi = fr_inv(fr_mul(Rp.Z, Ap.Z));
  const Fr zr = fr_mul(zi, Ap.Z), za = fr_mul(zi, Rp.Z);
  const Fr h0 = fr_mul(fr_mul(Rp.X, zr), c_K.FINV), h1 = fr_mul(Rp.Y, zr);
  const Fr h2 = fr_mul(fr_mul(Ap.X, za), c_K.FINV), h3 = fr_mul(Ap.Y, za);
  const Fr h4 = fr_to_mont_words(msg);                           // :321
  const Fr st0 = fr_select(j == 0, fr_zero(), fr_select(j == 1, h0, fr_select(j == 2, h1, fr_select(j == 3, h2, fr_select(j == 4, h3, h4)))));
  const Fr hm = grp_get(p5c_permute(j, gl, st0), 0);             // :332-333
  const Fr hm_plain = fr_canon(fr_mul(hm, fr_one_plain()));      // :336
  const Fr t = fl_mul(fr_from_words(pruned), c_K.L_R2, c_K);
  const Fr s = fl_canon4(fr_add(fl_mul(hm_plain, t, c_K), r), c_K);   // :335-339
  if (live && gl == 0) {
    u32 rx[8], ry[8], sw[8];
    fr_from_mont_words(h0, rx); fr_from_mont_words(h1, ry);
    fr_to_words(s, sw);
#pragma unroll
    for (int k = 0; k < 8; k++) { rx[k] = good ? rx[k] : 0u; ry[k] = good ? ry[k] : 0u; sw[k] = good ? sw[k] : 0u; }
    if constexpr (C64) {
      u32 c[8];
      compress_item(rx, ry, c, c_K);
      store_w8(out_r + i * 64, c); store_w8(out_r + i * 64 + 32, sw);
    } else {
      store_w8(out_r + i * 64, rx); store_w8(out_r + i * 64 + 32, ry); store_w8(out_s + i * 32, sw);
    }
    ok[i] = good ? 1 : 0;
  }
}
#define SIGN_SMALL_ARGS const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ keys, const uint8_t* __restrict__ msgs, \
                        size_t n, uint8_t* __restrict__ out_r, uint8_t* __restrict__ out_s, uint8_t* __restrict__ ok
__global__ void __launch_bounds__(BJJ_VS_BLOCK) bjj_k_sign_small(SIGN_SMALL_ARGS) { sign_small_body<false>(table, W, nwin, keys, msgs, n, out_r, out_s, ok); }
__global__ void __launch_bounds__(BJJ_VS_BLOCK) bjj_k_sign_small_c64(SIGN_SMALL_ARGS) { sign_small_body<true>(table, W, nwin, keys, msgs, n, out_r, out_s, ok); }

// =====================================================================================================================================
// B8.mul_scalar(n) / PrivateKey::public (src/lib.rs:149-164, 304-306) for SHORT calls: four lanes per item.  fixed_base_mul (bjj_device.hpp) with the accumulator on a
// quad: window 0's entry lifted to a point, one quad addition per further window (each table entry read one component per lane, all of them touched up front), then the
// affine conversion of THIS item alone -- no workgroup-wide inversion, no stash -- and, with COMPRESS, Point::compress (src/lib.rs:166-178) of it.
// =====================================================================================================================================
template <bool COMPRESS>
__device__ __forceinline__ void fixed_base_quad_body(const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ scalars, size_t n,
                                                     uint8_t* __restrict__ out) {
  constexpr u32 R1[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  const int lane = threadIdx.x, q = lane & 3;
  const size_t item = (size_t)blockIdx.x * BJJ_QUAD_ITEMS + (size_t)(lane >> 2);
  const bool live = item < n;
  const size_t i = live ? item : n - 1;
  u32 raw[8], sc[8], w[8];
  load_w8(scalars + i * 32, raw);
  scalar_mod_l(raw, sc, c_K);
  qfb_prefetch(table, W, nwin, sc);
  DigitStream ds = digit_stream(sc, W);
  bool neg;
  size_t slot = digit_next(ds, neg);
  Fr c;
  {  // window 0 is stored in T form: (2x' : 2y : 2 : 2x'y) costs no multiplication (fixed_base_mul)
    const Fr e0 = qfb_load(table, slot, neg, q);
    const Fr ymx = quad_bcast<0>(e0), ypx = quad_bcast<1>(e0);
    c = fr_select(q == 0, fr_reduce_weak(fr_sub(ypx, ymx)), fr_select(q == 1, fr_add(ypx, ymx), fr_select(q == 2, fr_add(fr_one(), fr_one()), fr_add(e0, fr_zero()))));
  }
  slot = digit_next(ds, neg);
  Fr cur = qfb_load(table, slot, neg, q);
#pragma unroll 1
  for (int k = 1; k + 1 < nwin; k++) {
    slot = digit_next(ds, neg);
    const Fr nxt = qfb_load(table, slot, neg, q);
    c = quad_add(q, c, cur);
    cur = nxt;
  }
  c = quad_add(q, c, cur);
  const Fr zi = fr_mul(fr_inv(quad_bcast<2>(c)), fr_one_plain());
  const Fr m = fr_select(q == 0, fr_mul(zi, c_K.FINV), zi);
  const Fr v = fr_cond_sub_kr(fr_mul(c, m), R1);                     // lane 0: x, lane 1: y, canonical
  if (COMPRESS) {
    const int big = __builtin_amdgcn_mov_dpp((int)plain_gt_halfq(v, c_K), 0x00, 0xf, 0xf, true);   // lane 0's x > (Q - 1) / 2, on the quad
    if (live && q == 1) {
      fr_to_words(v, w);
      if (big) w[7] |= 0x80000000u;
      store_w8(out + i * 32, w);
    }
  } else if (live && q < 2) {
    fr_to_words(v, w);
    store_w8(out + i * 64 + (size_t)q * 32, w);
  }
}
__global__ void __launch_bounds__(BJJ_QUAD_BLOCK) bjj_k_mul_fixed_base_quad(const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ scalars,
                                                                        size_t n, uint8_t* __restrict__ out) {
  fixed_base_quad_body<false>(table, W, nwin, scalars, n, out);
}
__global__ void __launch_bounds__(BJJ_QUAD_BLOCK) bjj_k_mul_fixed_base_quad_c32(const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ scalars,
                                                                            size_t n, uint8_t* __restrict__ out32) {
  fixed_base_quad_body<true>(table, W, nwin, scalars, n, out32);
}

namespace bjjk {
// the bulk of a short verify call; the scan (before) and the exact launch (behind) are K4's (k_verify.hip)
hipError_t verify_small(hipStream_t st, bool schnorr, const u32* table, int W, int nwin, const uint8_t* pk, const uint8_t* rb8, const uint8_t* s, const uint8_t* msg,
                        size_t n, uint8_t* ok) {
  const size_t per = BJJ_VS_BLOCK / 8, grid = (n + per - 1) / per;
  if (schnorr) BJJ_LAUNCH(bjj_k_schnorr_verify_small, dim3((unsigned)(grid ? grid : 1)), dim3(BJJ_VS_BLOCK), 0, st, table, W, nwin, pk, rb8, s, msg, n, ok);
  else if (n <= ((size_t)1 << 11))
    BJJ_LAUNCH(bjj_k_eddsa_verify_small, dim3((unsigned)(grid ? grid : 1)), dim3(2 * BJJ_VS_BLOCK), 0, st, table, W, nwin, pk, rb8, s, msg, n, ok);
  else
    BJJ_LAUNCH(bjj_k_eddsa_verify_small_joint, dim3((unsigned)(grid ? grid : 1)), dim3(BJJ_VS_BLOCK), 0, st, table, W, nwin, pk, rb8, s, msg, n, ok);
  return hipGetLastError();
}
hipError_t mul_fixed_base_quad(hipStream_t st, const u32* table, int W, int nwin, const uint8_t* scalars, size_t n, uint8_t* out, bool compressed) {
  const size_t grid = (n + BJJ_QUAD_ITEMS - 1) / BJJ_QUAD_ITEMS;
  if (compressed) BJJ_LAUNCH(bjj_k_mul_fixed_base_quad_c32, dim3((unsigned)(grid ? grid : 1)), dim3(BJJ_QUAD_BLOCK), 0, st, table, W, nwin, scalars, n, out);
  else BJJ_LAUNCH(bjj_k_mul_fixed_base_quad, dim3((unsigned)(grid ? grid : 1)), dim3(BJJ_QUAD_BLOCK), 0, st, table, W, nwin, scalars, n, out);
  return hipGetLastError();
}
// out_s == nullptr: the compressed form (out_r = 64-byte Signature::compress records), as bjjk::sign
hipError_t sign_small(hipStream_t st, const u32* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs, size_t n, uint8_t* out_r, uint8_t* out_s,
                      uint8_t* ok) {
  const size_t per = BJJ_VS_BLOCK / 8, grid = (n + per - 1) / per;
  if (out_s) BJJ_LAUNCH(bjj_k_sign_small, dim3((unsigned)(grid ? grid : 1)), dim3(BJJ_VS_BLOCK), 0, st, table, W, nwin, keys, msgs, n, out_r, out_s, ok);
  else BJJ_LAUNCH(bjj_k_sign_small_c64, dim3((unsigned)(grid ? grid : 1)), dim3(BJJ_VS_BLOCK), 0, st, table, W, nwin, keys, msgs, n, out_r, out_s, ok);
  return hipGetLastError();
}
hipError_t poseidon5_coop(hipStream_t st, const uint8_t* in, size_t n, uint8_t* out) {
  const size_t per = BJJ_P5C_BLOCK / BJJ_P5C_GROUP, grid = (n + per - 1) / per;
  BJJ_LAUNCH(bjj_k_poseidon5_coop, dim3((unsigned)(grid ? grid : 1)), dim3(BJJ_P5C_BLOCK), 0, st, in, n, out);
  return hipGetLastError();
}
}  // namespace bjjk
