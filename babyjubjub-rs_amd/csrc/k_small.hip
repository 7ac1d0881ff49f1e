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
