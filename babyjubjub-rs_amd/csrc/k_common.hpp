// Device-side code shared by the kernel translation units (k_*.hip): the constant block, the
// workgroup-wide simultaneous inversion + affine epilogue (K5), and the wave-cooperative gather
// policy of the fixed-base table.  The library is split into one translation unit per kernel
// family so that the ~140 s single-file device compile runs in parallel; every unit carries its
// own (internal-linkage) copy of the 63 KB constant block.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "sign.hpp"
#include "bjj_constants.inc"
#include "bjj_launch.hpp"

using namespace bjj;

// Every launcher returns hipGetLastError() right after its launch.  That call reports the thread's LAST error, which may be a
// stale, harmless one (hipErrorNotReady of an event query): the slate is wiped before the launch so that the
// status a launcher returns is the status of its own launch.
#define BJJ_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

static __constant__ Consts c_K = {
    BJJ_K_A, BJJ_K_D, BJJ_K_F, BJJ_K_FINV_PLAIN, BJJ_K_FINV, BJJ_K_L_R1, BJJ_K_L_R2, BJJ_K_DP, BJJ_K_D2P, BJJ_K_DPINV, BJJ_K_B8X, BJJ_K_B8Y, BJJ_K_TS_G, BJJ_K_HALFQ,
    BJJ_K_ORDER, BJJ_K_ORDER2, BJJ_K_ORDER4, BJJ_K_L, BJJ_K_L2, BJJ_K_L4,
    BJJ_K_POSEIDON_CF, BJJ_K_POSEIDON_KP, BJJ_K_POSEIDON_SP, BJJ_K_POSEIDON_AL, BJJ_K_POSEIDON_M, BJJ_K_POSEIDON_CAB,
    BJJ_K_TS_NEG, BJJ_K_TS_HALF, BJJ_K_TS_HASH};

// ---------------------------------------------------------------------------
// workgroup-wide simultaneous inversion: every thread passes x (Montgomery, != 0,
// < 2r) and receives 1/x.  The workgroup is cut into 64 groups of G = BJJ_EPI_BLOCK/64
// consecutive lanes: prefix and suffix products inside a group by cross-lane shuffles
// (log2 G steps each, no barrier), the 64 group products go through LDS to ONE wave
// whose 64 lanes invert one group product each (binary GCD), and every thread
// finishes with 1/x = (1/group product) * (product of the lanes before it) * (after it).
// ---------------------------------------------------------------------------
template <int GROUP>
__device__ __forceinline__ Fr fr_shfl_up(const Fr& f, int d) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = __shfl_up(f.v[i], d, GROUP);
  return r;
}
template <int GROUP>
__device__ __forceinline__ Fr fr_shfl_down(const Fr& f, int d) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = __shfl_down(f.v[i], d, GROUP);
  return r;
}
template <int BLOCK>
__device__ Fr block_invert(const Fr& x, u32* lds /* NL * 64 words */) {
  constexpr int GROUP = BLOCK / 64;
  const int t = threadIdx.x, gl = t & (GROUP - 1), grp = t / GROUP;
  Fr pre = x, suf = x;
#pragma unroll 1
  for (int d = 1; d < GROUP; d <<= 1) {  // inclusive prefix / suffix products inside the group
    Fr yp = fr_shfl_up<GROUP>(pre, d), ys = fr_shfl_down<GROUP>(suf, d);
    pre = fr_mul(pre, fr_select(gl >= d, yp, fr_one()));
    suf = fr_mul(suf, fr_select(gl + d < GROUP, ys, fr_one()));
  }
  Fr epre = fr_select(gl > 0, fr_shfl_up<GROUP>(pre, 1), fr_one());                    // exclusive versions
  Fr esuf = fr_select(gl + 1 < GROUP, fr_shfl_down<GROUP>(suf, 1), fr_one());
  if (gl == GROUP - 1) {
#pragma unroll
    for (int i = 0; i < NL; i++) lds[i * 64 + grp] = pre.v[i];  // limb-major: conflict-free
  }
  __syncthreads();
  // the inverting wave rotates with the workgroup index so that co-resident workgroups do not queue on one SIMD
  if ((t >> 6) == (int)((blockIdx.x + (blockIdx.x >> 8)) % (BLOCK / 64))) {
    const int l = t & 63;
    Fr tot;
#pragma unroll
    for (int i = 0; i < NL; i++) tot.v[i] = lds[i * 64 + l];
    Fr inv = fr_inv(tot);
#pragma unroll
    for (int i = 0; i < NL; i++) lds[i * 64 + l] = inv.v[i];
  }
  __syncthreads();
  Fr ginv;
#pragma unroll
  for (int i = 0; i < NL; i++) ginv.v[i] = lds[i * 64 + grp];
  __syncthreads();
  return fr_mul(fr_mul(ginv, epre), esuf);
}

// Phase-1 record for the affine epilogue: X, Y (as 2 x 32-byte integers) go to `xy_item` -- the item's final output slot when the
// result is the 64-byte affine point, a slot of the scratch set's XY area when the result is the 32-byte compressed point
// (EPI_COMPRESS) -- Z and the lane's running prefix product go to scratch.
__device__ __forceinline__ void epilogue_stash(const Ext& p, Fr& run, uint8_t* xy_item, u32* scr_item) {
  u32 w[8];
  fr_to_words(p.X, w); store_w8(xy_item, w);
  fr_to_words(p.Y, w); store_w8(xy_item + 32, w);
  fr_to_words(p.Z, w); store_w8(scr_item, w);
  fr_to_words(run, w); store_w8(scr_item + 8, w);
  run = fr_mul(run, p.Z);
}
// An item this launch does NOT finish (EPI_SKIPPABLE kernels; K2: the point is off the curve, the exact kernel K6 owns the item's
// output slot and may be writing it while this launch runs): Z = 0 marks it -- the Z of a computed point is never 0 -- and the
// running product passes it by.  Nothing is written to the output slot, neither here nor in phase 2.
__device__ __forceinline__ void epilogue_stash_skipped(u32* scr_item) {
  const u32 z[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  store_w8(scr_item, z);
}
// Epilogue forms (template flags; the default -- affine output, no skipped items -- is what K1 and the signer use, and its code
// does not change when the other forms are instantiated next to it)
//   EPI_STASH_APART: the affine form with the phase-1 stash in `xy` when that is given (run time) instead of in the output slots -- for
//   an output array that must be written exactly once and never read: the caller's pinned HOST memory behind its device mapping
enum : unsigned { EPI_AFFINE = 0u, EPI_COMPRESS = 1u, EPI_SKIPPABLE = 2u, EPI_STASH_APART = 4u };
// Phase-2: given inv = 1 / (product of this lane's Z_0..Z_i) as a PLAIN (non-Montgomery) integer, finish item i
// and step inv down to 1 / (Z_0..Z_{i-1}).  A Montgomery product of a plain and a Montgomery operand is the plain
// product, so Y * (1/Z) lands directly on the canonical output integer.  Output: reference-curve (x, y), or with EPI_COMPRESS
// Point::compress of it (src/lib.rs:166-178: y as 32 little-endian bytes, bit 255 set when x > Q >> 1) -- x is canonical right
// here, so the sign costs one comparison instead of a second pass over 64-byte points.
template <unsigned FORM = EPI_AFFINE>
__device__ __forceinline__ void epilogue_finish(Fr& inv, uint8_t* out_item, uint8_t* xy_item, const u32* scr_item) {
  constexpr u32 R1[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  u32 w[8];
  load_w8(scr_item, w);
  if (FORM & EPI_SKIPPABLE) {
    if ((w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0u) return;
  }
  Fr Z = fr_from_words(w);
  load_w8(scr_item + 8, w); Fr P = fr_from_words(w);
  load_w8(xy_item, w);      Fr X = fr_from_words(w);
  load_w8(xy_item + 32, w); Fr Y = fr_from_words(w);
  Fr zinv = fr_mul(inv, P);               // plain 1/Z
  inv = fr_mul(inv, Z);
  Fr c2 = fr_mul(zinv, c_K.FINV);         // plain 1/(Z F): maps x' back to the reference curve
  Fr x = fr_cond_sub_kr(fr_mul(X, c2), R1);
  Fr y = fr_cond_sub_kr(fr_mul(Y, zinv), R1);
  if (FORM & EPI_COMPRESS) {
    fr_to_words(y, w);
    if (plain_gt_halfq(x, c_K)) w[7] |= 0x80000000u;
    store_w8(out_item, w);
  } else {
    fr_to_words(x, w); store_w8(out_item, w);
    fr_to_words(y, w); store_w8(out_item + 32, w);
  }
}
// out: 64 bytes per item (32 with EPI_COMPRESS); xy: where phase 1 stashed X, Y, 64 bytes per item (== out for the affine form)
template <int BLOCK = BJJ_EPI_BLOCK, unsigned FORM = EPI_AFFINE>
__device__ __forceinline__ void epilogue_run(Fr run, size_t n, size_t tid, size_t nthreads, uint8_t* out, u32* scratch,
                                             u32* lds, uint8_t* xy = nullptr) {
  Fr inv = fr_mul(block_invert<BLOCK>(run, lds), fr_one_plain());  // out of Montgomery form once per lane
  if (tid >= n) return;
  size_t cnt = (n - tid + nthreads - 1) / nthreads;
#pragma unroll 1
  for (size_t m = cnt; m-- > 0;) {
    size_t i = tid + m * nthreads;
    if (FORM & EPI_COMPRESS) epilogue_finish<FORM>(inv, out + i * 32, xy + i * 64, scratch + i * 16);
    else if (FORM & EPI_STASH_APART) epilogue_finish<FORM>(inv, out + i * 64, (xy ? xy : out) + i * 64, scratch + i * 16);
    else epilogue_finish<FORM>(inv, out + i * 64, out + i * 64, scratch + i * 16);
  }
}

// Wave-cooperative gather straight into LDS (policy interface: bjj_device.hpp "gather policies").
// A lane-private gather costs 7 load instructions x 64 lanes, every lane in its own 128-byte line and, for tables beyond
// the TLB reach, its own page: measured alone (tools/ubench/gather_bench.hip) that pattern sustains 12.7 G gathers/s on a
// 5.9 GB table and 10.8 G/s on 155 GB, the one used here 48 and 46 G/s.  Load instruction k of a wave fetches the 8 FULL
// lines of the entries owned by lanes 8k .. 8k+7: lane L moves one 16-byte chunk of the entry of lane e = 8k + L/8, whose slot
// number it obtains by a cross-lane read, with global_load_lds_dwordx4 (no VGPR staging; LDS address = M0 + 16 L), so an
// instruction touches 8 lines, each exactly once.  Chunk c of entry e lands at position c ^ ((e >> 1) & 7) of the entry's
// 128-byte LDS row, which makes the read-back of one's own entry (7 x ds_read_b128, lane stride 128 B) bank-conflict free.
// hipcc does not track LDS-DMA completion, hence the explicit s_waitcnt.  ALL 64 lanes of the wave must call issue/finish
// together (cross-lane reads).  NBUF = 2 staging areas per wave let two gathers be in flight (the start of a fresh
// multiplication in K1); with NBUF = 1 the loops still overlap gather j+1 with addition j, whose entry is in registers by then.
#define FB_STAGE_WORDS (64 * NIELS_WORDS)   // 8 KB: one staged entry per lane
template <int NBUF>
struct GatherCoopLds {
  struct Pending {};
  static constexpr int kBuffers = NBUF;
  const u32* table;
  u32* wlds;   // this wave's staging area
  int lane;
  // all 64 lanes run together under this policy: wave-wide maximum (butterfly over the lanes)
  static __device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
  }
  __device__ __forceinline__ void issue(size_t slot, Pending&, int buf) const {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int e = 8 * k + (lane >> 3);
      const u32 s = (u32)__shfl((int)(u32)slot, e, 64);
      const int c = (lane & 7) ^ ((e >> 1) & 7);
      __builtin_amdgcn_global_load_lds(table + (size_t)s * NIELS_WORDS + c * 4,
                                       (__attribute__((address_space(3))) void*)(wlds + (NBUF > 1 ? buf : 0) * FB_STAGE_WORDS + k * (8 * NIELS_WORDS)),
                                       16, 0, 0);
    }
  }
  __device__ __forceinline__ Niels finish(Pending&, int buf) const {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const U4* q = (const U4*)(wlds + (NBUF > 1 ? buf : 0) * FB_STAGE_WORDS + lane * NIELS_WORDS);
    const int x = (lane >> 1) & 7;
    const U4 a = q[0 ^ x], b = q[1 ^ x], c = q[2 ^ x], d = q[3 ^ x], e = q[4 ^ x], f = q[5 ^ x], h = q[6 ^ x];
    Niels n;
    n.ymx = Fr{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x}};
    n.ypx = Fr{{c.y, c.z, c.w, d.x, d.y, d.z, d.w, e.x, e.y}};
    n.t2d = Fr{{e.z, e.w, f.x, f.y, f.z, f.w, h.x, h.y, h.z}};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are free again before the next issue overwrites them
    return n;
  }
};

#include "slot_queue.hpp"   // per-XCD rings of free table slots (verify groups, K2 tiles)
