// Poseidon t = 6 (5 inputs), as the reference calls it for EdDSA:
//   POSEIDON.hash(vec![R.x, R.y, A.x, A.y, msg])   src/lib.rs:400-404 (also :333, :370)
// The permutation itself lives in the third-party crate poseidon-rs 0.0.8
// (Cargo.toml:20); round structure per SURVEY.md Appendix B: state = [0, in0..in4];
// 68 rounds (4 full, 60 partial, 4 full) of { add round constants; x^5 on all
// (full) or on state[0] only (partial); state <- M . state }; output state[0].
//
// GPU mapping: one hash per lane, state in 54 VGPRs (6 x 9 limbs), round
// constants / MDS read with wave-uniform indices (scalar loads).  The MDS row is a
// 6-term dot product accumulated in the unsaturated limb columns and reduced ONCE
// (fr_dot6): 6*81 + 90 multiply-adds instead of 6*171.  S-box and row loops rotate
// the state through registers so the loop bodies stay small (instruction cache).
#pragma once
#include "curve.hpp"

namespace bjj {

// sum_j a[j]*b[j] * 2^-261 mod r with one Montgomery reduction.
// Needs N-form limbs (< 2^29) on both sides and sum of value products < r*2^261.
BJJ_HD Fr fr_dot6(const Fr* a, const Fr* b) {
#if defined(BJJ_DEBUG_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
  for (int j = 0; j < 6; j++)
    for (int i = 0; i < NL; i++) { assert(a[j].v[i] < (1u << 29) || i == NL - 1); assert(b[j].v[i] < (1u << 29) || i == NL - 1); }
#endif
  u32 m[NL];
  Fr r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
      for (int i = 0; i <= k; i++) acc += (u64)a[j].v[i] * b[j].v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * fr_modlimb(k - i);
    m[k] = ((u32)acc * BJJ_NINV29) & MASK29;
    acc += (u64)m[k] * BJJ_N0;
    acc >>= 29;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
      for (int i = k - (NL - 1); i < NL; i++) acc += (u64)a[j].v[i] * b[j].v[k - i];
#pragma unroll
    for (int i = k - (NL - 1); i < NL; i++) acc += (u64)m[i] * fr_modlimb(k - i);
    r.v[k - NL] = (u32)acc & MASK29;
    acc >>= 29;
  }
  r.v[NL - 1] = (u32)acc;
  return r;
}

BJJ_HD Fr fr_pow5(const Fr& x) {  // x < 4r
  Fr x2 = fr_sqr(x);
  Fr x4 = fr_sqr(x2);
  return fr_mul(x4, x);
}

// in[0..4] Montgomery (< 2r each); returns Montgomery hash (< 2r)
BJJ_HD Fr poseidon5(const Fr in[5], const Consts& K) {
  Fr st[6];
  st[0] = fr_zero();
#pragma unroll
  for (int j = 0; j < 5; j++) st[j + 1] = in[j];
#pragma unroll 1
  for (int r = 0; r < 68; r++) {
    const Fr* C = &K.PC[r * 6];
    const bool full = (r < 4) || (r >= 64);
    if (full) {
      // S-box on every element: process st[0], rotate left, six times
#pragma unroll 1
      for (int j = 0; j < 6; j++) {
        Fr x = fr_pow5(fr_add(st[0], C[j]));
        st[0] = st[1]; st[1] = st[2]; st[2] = st[3]; st[3] = st[4]; st[4] = st[5]; st[5] = x;
      }
    } else {
      st[0] = fr_pow5(fr_add(st[0], C[0]));
#pragma unroll
      for (int j = 1; j < 6; j++) st[j] = fr_add(st[j], C[j]);
    }
    // state <- M . state   (new[i] = sum_j M[i][j] st[j]); rows produced through a rotating window
    Fr nw[6];
#pragma unroll
    for (int j = 0; j < 6; j++) nw[j] = fr_zero();
#pragma unroll 1
    for (int i = 0; i < 6; i++) {
      Fr d = fr_dot6(&K.PM[i * 6], st);
      nw[0] = nw[1]; nw[1] = nw[2]; nw[2] = nw[3]; nw[3] = nw[4]; nw[4] = nw[5]; nw[5] = d;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) st[j] = nw[j];
  }
  return st[0];
}

}  // namespace bjj
