// libbjj_hip.so -- HIP kernels (gfx950) and the extern "C" boundary declared in
// include/bjj_hip.h.  There is no CPU fallback anywhere in this file: every entry
// point launches device code or returns an error.
//
// Kernel map (SURVEY.md section 2 "kernel inventory"; bodies in bjj_device.hpp / sign.hpp):
//   bjj_k_build_fixed_table   init-time: window table of B8 multiples (affine Niels form)
//   bjj_k_mul_fixed_base      K1  B8.mul_scalar(n)               src/lib.rs:149-164, 37-46
//   bjj_k_mul_var_base        K2  P.mul_scalar(n), on-curve P    src/lib.rs:149-164
//   bjj_k_mul_var_base_exact  K6  the reference's exact op sequence for off-curve P
//   bjj_k_poseidon5           K3  POSEIDON.hash([a,b,c,d,e])     src/lib.rs:400-404
//   bjj_k_eddsa_verify_scan / bjj_k_eddsa_verify / bjj_k_schnorr_verify
//                             K4  verify / verify_schnorr        src/lib.rs:395-412, 375-385
//   bjj_k_point_add               PointProjective::add + affine  src/lib.rs:88-131, 70-85
//   bjj_k_compress_points / bjj_k_decompress_points / bjj_k_merge_codec_flags
//                                 wire format                    src/lib.rs:166-224, 260-268
//   bjj_k_scalar_keys / bjj_k_sign / bjj_k_sign_schnorr  signer side   src/lib.rs:284-361
// K5 (batched affine conversion) is the epilogue of K1/K2: Montgomery's trick per lane over its
// items, then across the 512-lane workgroup (shuffle scans inside groups of 8 lanes, one wave inverting the
// 64 group products with one binary-GCD inversion per lane): block_invert below.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <new>
#include <thread>

#include "../../include/bjj_hip.h"
#include "sign.hpp"
#include "bjj_constants.inc"

using namespace bjj;

#define BJJ_VERSION_STRING "bjj-hip 0.1.0 gfx950"
#define BJJ_BLOCK 256
// Fixed-base window width when bjj_init is given 0: the widest of these whose table fits in 60 % of the device's
// free memory -- 28 bits = 9 signed digits, 9 x (2^27 + 1) entries = 154.6 GB of the 288 GB of HBM; 26 = 10 digits,
// 42.9 GB; 23 = 11 digits, 5.9 GB; 21 = 12 digits, 1.6 GB; 16 = 16 digits, 67 MB.  One addition less per step:
// the kernel is VALU-bound and the cooperative gathers keep the table reads off the critical path at any size
// (profiles/r01_ablation_signed_windows.txt for the earlier lane-private gathers; r01i_* for these).
static const int kAutoWindowBits[] = {28, 26, 23, 21, 16};
#define BJJ_MAX_WINDOW_BITS 28
// Workgroup size of the kernels that end in the shared-inversion epilogue: one binary-GCD
// inversion (executed by one wave) is amortised over the whole workgroup.
#define BJJ_EPI_BLOCK 512

__constant__ Consts c_K = {
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

// Phase-1 record for the affine epilogue: X, Y (as 2 x 32-byte integers) go to the
// item's final output slot; Z and the lane's running prefix product go to scratch.
__device__ __forceinline__ void epilogue_stash(const Ext& p, Fr& run, uint8_t* out_item, u32* scr_item) {
  u32 w[8];
  fr_to_words(p.X, w); store_w8(out_item, w);
  fr_to_words(p.Y, w); store_w8(out_item + 32, w);
  fr_to_words(p.Z, w); store_w8(scr_item, w);
  fr_to_words(run, w); store_w8(scr_item + 8, w);
  run = fr_mul(run, p.Z);
}
// Phase-2: given inv = 1 / (product of this lane's Z_0..Z_i) as a PLAIN (non-Montgomery) integer, finish item i
// and step inv down to 1 / (Z_0..Z_{i-1}).  A Montgomery product of a plain and a Montgomery operand is the plain
// product, so Y * (1/Z) lands directly on the canonical output integer.  Output: reference-curve (x, y).
__device__ __forceinline__ void epilogue_finish(Fr& inv, uint8_t* out_item, const u32* scr_item) {
  constexpr u32 R1[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  u32 w[8];
  load_w8(scr_item, w);     Fr Z = fr_from_words(w);
  load_w8(scr_item + 8, w); Fr P = fr_from_words(w);
  load_w8(out_item, w);     Fr X = fr_from_words(w);
  load_w8(out_item + 32, w); Fr Y = fr_from_words(w);
  Fr zinv = fr_mul(inv, P);               // plain 1/Z
  inv = fr_mul(inv, Z);
  Fr c2 = fr_mul(zinv, c_K.FINV);         // plain 1/(Z F): maps x' back to the reference curve
  Fr x = fr_cond_sub_kr(fr_mul(X, c2), R1);
  Fr y = fr_cond_sub_kr(fr_mul(Y, zinv), R1);
  fr_to_words(x, w); store_w8(out_item, w);
  fr_to_words(y, w); store_w8(out_item + 32, w);
}
__device__ __forceinline__ void epilogue_run(Fr run, size_t n, size_t tid, size_t nthreads, uint8_t* out, u32* scratch,
                                             u32* lds) {
  Fr inv = fr_mul(block_invert<BJJ_EPI_BLOCK>(run, lds), fr_one_plain());  // out of Montgomery form once per lane
  if (tid >= n) return;
  size_t cnt = (n - tid + nthreads - 1) / nthreads;
#pragma unroll 1
  for (size_t m = cnt; m-- > 0;) {
    size_t i = tid + m * nthreads;
    epilogue_finish(inv, out + i * 64, scratch + i * 16);
  }
}

// ---------------------------------------------------------------------------
// init: fixed-base table (layout and recoding: bjj_device.hpp "fixed base").
//   bases:  one thread per window j -> P_j = 2^(W j) * B8 (ladder + inversion; nwin threads)
//   build:  one thread per chain of `chain` consecutive digits of one window (fixed_table_chain)
//   check:  one thread per entry, link conditions of fixed_table_check_slot (bjj_check_table)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) bjj_k_fixed_window_bases(u32* bases, int W, int nwin) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nwin) return;
  store_niels(bases + (size_t)j * NIELS_WORDS, fixed_table_entry(1u, j, W, c_K));
}
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_build_fixed_table(u32* table, const u32* __restrict__ bases, int W, int nwin,
                                                                 u32 chain) {
  const size_t stride = fixed_stride(W);
  const size_t cpw = (stride + chain - 1) / chain;  // chains per window
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cpw * (size_t)nwin) return;
  const int j = (int)(t / cpw);
  const size_t k0 = (t % cpw) * chain;
  const u32 cnt = (u32)(stride - k0 < chain ? stride - k0 : chain);
  fixed_table_chain(table, load_niels(bases + (size_t)j * NIELS_WORDS), (size_t)j * stride + k0, (u32)k0, cnt, W, c_K);
}
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_check_fixed_table(const u32* __restrict__ table, const u32* __restrict__ bases,
                                                                 int W, int nwin, unsigned long long* bad) {
  const size_t stride = fixed_stride(W);
  const size_t total = stride * (size_t)nwin, nthreads = (size_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
#pragma unroll 1
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += nthreads)
    mine += (unsigned long long)fixed_table_check_slot(table, bases, (int)(e / stride), (u32)(e % stride), W, nwin, c_K);
  if (mine) atomicAdd(bad, mine);
}

// ---------------------------------------------------------------------------
// K1: fixed base
// ---------------------------------------------------------------------------
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

__global__ void __launch_bounds__(BJJ_EPI_BLOCK) bjj_k_mul_fixed_base(const u32* __restrict__ table, int W, int nwin,
                                                                  const uint8_t* __restrict__ scalars, size_t n,
                                                                  uint8_t* __restrict__ out, u32* __restrict__ scratch) {
  __shared__ u32 lds[NL * 64];
  __shared__ __attribute__((aligned(16))) u32 stage[(BJJ_EPI_BLOCK / 64) * 2 * FB_STAGE_WORDS];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const GatherCoopLds<2> fb = {table, stage + (threadIdx.x >> 6) * 2 * FB_STAGE_WORDS, lane};
  Fr run = fr_one();
#pragma unroll 1
  for (size_t i = tid; i - lane < n; i += nthreads) {  // wave-uniform trip count: the gathers are cooperative
    const bool valid = i < n;
    u32 sc[8];
    load_w8(scalars + (valid ? i : n - 1) * 32, sc);
    Ext p = fixed_base_mul(fb, W, nwin, sc, c_K);
    if (valid) epilogue_stash(p, run, out + i * 64, scratch + i * 16);
  }
  epilogue_run(run, n, tid, nthreads, out, scratch, lds);
}

// ---------------------------------------------------------------------------
// K2: variable base.  Off-curve points are appended to `slow` (slow[0] = count, item indices from
// slow[8]) and finished by K6 (bjj_k_mul_var_base_exact) right after this kernel.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BJJ_EPI_BLOCK) bjj_k_mul_var_base(const uint8_t* __restrict__ pts,
                                                                const uint8_t* __restrict__ scalars, size_t n,
                                                                uint8_t* __restrict__ out, u32* __restrict__ scratch,
                                                                u32* __restrict__ vb_tables, u32* __restrict__ slow) {
  __shared__ u32 lds[NL * 64];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  u32* tbl = vb_tables + tid * VB_TABLE_WORDS;
  Fr run = fr_one();
#pragma unroll 1
  for (size_t i = tid; i < n; i += nthreads) {
    u32 w[8], sc[8];
    load_w8(pts + i * 64, w);      Fr x = fr_to_mont_words(w);
    load_w8(pts + i * 64 + 32, w); Fr y = fr_to_mont_words(w);
    load_w8(scalars + i * 32, sc);
    Ext p = ext_identity();
    if (ref_on_curve(x, y, c_K)) p = var_base_fast(x, y, sc, tbl, c_K);
    else slow[8 + atomicAdd(&slow[0], 1u)] = (u32)i;  // placeholder result; K6 overwrites the output
    epilogue_stash(p, run, out + i * 64, scratch + i * 16);
  }
  epilogue_run(run, n, tid, nthreads, out, scratch, lds);
}
// K6: exact replay of the reference's loop for the (rare) off-curve inputs, one per lane.
__global__ void __launch_bounds__(64) bjj_k_mul_var_base_exact(const uint8_t* __restrict__ pts,
                                                               const uint8_t* __restrict__ scalars,
                                                               uint8_t* __restrict__ out, const u32* __restrict__ slow) {
  const u32 cnt = slow[0];
  for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += gridDim.x * blockDim.x) {
    const size_t i = slow[8 + j];
    u32 w[8], sc[8];
    load_w8(pts + i * 64, w);      Fr x = fr_to_mont_words(w);
    load_w8(pts + i * 64 + 32, w); Fr y = fr_to_mont_words(w);
    load_w8(scalars + i * 32, sc);
    Fr ox, oy;
    ref_mul_scalar(x, y, sc, 8, ox, oy, c_K);
    fr_from_mont_words(ox, w); store_w8(out + i * 64, w);
    fr_from_mont_words(oy, w); store_w8(out + i * 64 + 32, w);
  }
}

// ---------------------------------------------------------------------------
// K3: Poseidon, 5 inputs
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_poseidon5(const uint8_t* __restrict__ in, size_t n,
                                                             uint8_t* __restrict__ out) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = tid; i < n; i += nthreads) {
    Fr h[5];
    u32 w[8];
#pragma unroll
    for (int j = 0; j < 5; j++) { load_w8(in + i * 160 + j * 32, w); h[j] = fr_to_mont_words(w); }
    Fr r = poseidon5(h, c_K);
    fr_from_mont_words(r, w);
    store_w8(out + i * 32, w);
  }
}

// ---------------------------------------------------------------------------
// K4: EdDSA-Poseidon verify, two launches:
//  (1) bjj_k_eddsa_verify_scan: on-curve tests only (14 multiplications per item); items whose
//      pk or R is off the curve -- they need the reference's exact, ~3x longer, strictly serial
//      formula sequence -- are appended to the work list `wl`.
//  (2) bjj_k_eddsa_verify: waves pull work through atomic cursors: first 64-item groups of the
//      exact list (so the long items start at t = 0 and overlap everything else), then 64-item
//      chunks of the whole batch on the fast path.  No wave ever runs both paths for one group.
// wl layout (u32 words): [0] exact count, [2..3] exact cursor (u64), [4..5] batch cursor (u64),
// [8..] exact item indices.
// ---------------------------------------------------------------------------
#define WL_HDR 8
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_eddsa_verify_scan(const uint8_t* __restrict__ pk,
                                                                     const uint8_t* __restrict__ rb8,
                                                                     const uint8_t* __restrict__ msg, size_t n,
                                                                     u32* __restrict__ wl) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    VerifyIn in = {pk + i * 64, rb8 + i * 64, nullptr, msg + i * 32};
    if (verify_needs_exact(in, c_K)) wl[WL_HDR + atomicAdd(&wl[0], 1u)] = (u32)i;
  }
}
__device__ __forceinline__ unsigned long long wave_grab(u32* cursor_words, int lane) {
  unsigned long long c = 0;
  if (lane == 0) c = atomicAdd((unsigned long long*)cursor_words, 64ULL);
  return __shfl(c, 0, 64);
}
template <bool SCHNORR>
__device__ __forceinline__ void verify_kernel_body(const u32* __restrict__ table, int W, int nwin,
                                                   const uint8_t* __restrict__ pk, const uint8_t* __restrict__ rb8,
                                                   const uint8_t* __restrict__ s, const uint8_t* __restrict__ msg, size_t n,
                                                   uint8_t* __restrict__ ok, u32* __restrict__ vb_tables,
                                                   u32* __restrict__ wl) {
  __shared__ __attribute__((aligned(16))) u32 stage[(BJJ_BLOCK / 64) * FB_STAGE_WORDS];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const GatherCoopLds<1> fb = {table, stage + (threadIdx.x >> 6) * FB_STAGE_WORDS, lane};
  u32* tbl = vb_tables + tid * VB_VERIFY_WORDS;
  const unsigned long long nexact = wl[0];
#pragma unroll 1
  for (;;) {  // exact-path groups first
    const unsigned long long c = wave_grab(wl + 2, lane);
    if (c >= nexact) break;
    if (c + lane < nexact) {
      const size_t i = wl[WL_HDR + c + lane];
      VerifyIn in = {pk + i * 64, rb8 + i * 64, s + i * 32, msg + i * 32};
      ok[i] = (uint8_t)verify_exact_t<SCHNORR>(in, table, W, nwin, tbl, c_K);
    }
  }
#pragma unroll 1
  for (;;) {  // then the bulk
    const unsigned long long c = wave_grab(wl + 4, lane);
    if (c >= n) break;
    const size_t i = c + lane, ic = i < n ? i : n - 1;  // every lane runs (cooperative gathers); the tail repeats the last item
    VerifyIn in = {pk + ic * 64, rb8 + ic * 64, s + ic * 32, msg + ic * 32};
    bool need_exact;
    const int v = verify_fast_t<SCHNORR>(in, fb, W, nwin, tbl, c_K, need_exact);
    if (i < n && !need_exact) ok[i] = (uint8_t)v;  // exact items were written by the first loop
  }
}
// verify_schnorr (src/lib.rs:375-385): same structure, verdict 2 = Err (msg > Q)
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_schnorr_verify(const u32* __restrict__ table, int W, int nwin,
                                                                     const uint8_t* __restrict__ pk,
                                                                     const uint8_t* __restrict__ rb8,
                                                                     const uint8_t* __restrict__ s,
                                                                     const uint8_t* __restrict__ msg, size_t n,
                                                                     uint8_t* __restrict__ ok, u32* __restrict__ vb_tables,
                                                                     u32* __restrict__ wl) {
  verify_kernel_body<true>(table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl);
}
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_eddsa_verify(const u32* __restrict__ table, int W, int nwin,
                                                                   const uint8_t* __restrict__ pk,
                                                                   const uint8_t* __restrict__ rb8,
                                                                   const uint8_t* __restrict__ s,
                                                                   const uint8_t* __restrict__ msg, size_t n,
                                                                   uint8_t* __restrict__ ok, u32* __restrict__ vb_tables,
                                                                   u32* __restrict__ wl) {
  verify_kernel_body<false>(table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl);
}

// ---------------------------------------------------------------------------
// PointProjective::add on affine inputs followed by affine()  (reference-exact)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_point_add(const uint8_t* __restrict__ p, const uint8_t* __restrict__ q,
                                                             size_t n, uint8_t* __restrict__ out) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = tid; i < n; i += nthreads) {
    u32 w[8];
    RefProj a, b;
    load_w8(p + i * 64, w); a.x = fr_to_mont_words(w);
    load_w8(p + i * 64 + 32, w); a.y = fr_to_mont_words(w); a.z = fr_one();
    load_w8(q + i * 64, w); b.x = fr_to_mont_words(w);
    load_w8(q + i * 64 + 32, w); b.y = fr_to_mont_words(w); b.z = fr_one();
    RefProj r = ref_add(a, b, c_K);
    Fr ox = fr_zero(), oy = fr_zero();
    if (!fr_is_zero(r.z)) { Fr zi = fr_inv(r.z); ox = fr_mul(r.x, zi); oy = fr_mul(r.y, zi); }
    fr_from_mont_words(ox, w); store_w8(out + i * 64, w);
    fr_from_mont_words(oy, w); store_w8(out + i * 64 + 32, w);
  }
}

// ---------------------------------------------------------------------------
// codec row: Point::compress, decompress_point, decompress_signature
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_compress_points(const uint8_t* __restrict__ in_xy, size_t n,
                                                                   uint8_t* __restrict__ out) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    u32 x[8], y[8], o[8];
    load_w8(in_xy + i * 64, x); load_w8(in_xy + i * 64 + 32, y);
    compress_item(x, y, o, c_K);
    store_w8(out + i * 32, o);
  }
}
// in: records of `stride` bytes whose first 32 bytes are a compressed point.  When out_s is
// given (signatures, stride 64: lib.rs:260-268) bytes 32..63 are copied there unchanged.
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_decompress_points(const uint8_t* __restrict__ in, size_t stride,
                                                                        size_t n, uint8_t* __restrict__ out_xy,
                                                                        uint8_t* __restrict__ ok,
                                                                        uint8_t* __restrict__ out_s) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    u32 w[8], ox[8], oy[8];
    load_w8(in + i * stride, w);
    const bool good = decompress_item(w, ox, oy, c_K);
    store_w8(out_xy + i * 64, ox); store_w8(out_xy + i * 64 + 32, oy);
    ok[i] = good ? 1 : 0;
    if (out_s) { load_w8(in + i * stride + 32, w); store_w8(out_s + i * 32, w); }
  }
}
// verdict byte of the compressed-input verify: 2 where a point failed to decompress (the
// reference returns Err there and never reaches verify), else verify()'s 1 / 0.
__global__ void bjj_k_merge_codec_flags(uint8_t* __restrict__ ok, const uint8_t* __restrict__ f_pk,
                                        const uint8_t* __restrict__ f_r, size_t n) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads)
    if (!(f_pk[i] && f_r[i])) ok[i] = 2;
}

// ---------------------------------------------------------------------------
// signer row: PrivateKey::scalar_key / public / sign, src/lib.rs:284-342
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_scalar_keys(const uint8_t* __restrict__ keys, size_t n,
                                                               uint8_t* __restrict__ out) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    u32 k[8], sk[8], pr[8], hi[8];
    load_w8(keys + i * 32, k);
    scalar_key_words(k, sk, pr, hi);
    store_w8(out + i * 32, sk);
  }
}
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_sign(const u32* __restrict__ table, int W, int nwin,
                                                           const uint8_t* __restrict__ keys,
                                                           const uint8_t* __restrict__ msgs, size_t n,
                                                           uint8_t* __restrict__ out_r, uint8_t* __restrict__ out_s,
                                                           uint8_t* __restrict__ ok) {
  __shared__ __attribute__((aligned(16))) u32 stage[(BJJ_BLOCK / 64) * FB_STAGE_WORDS];
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const GatherCoopLds<1> fb = {table, stage + (threadIdx.x >> 6) * FB_STAGE_WORDS, lane};
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i - lane < n; i += nthreads) {  // wave-uniform trip count
    const size_t ic = i < n ? i : n - 1;
    u32 k[8], m[8], rx[8], ry[8], s[8];
    load_w8(keys + ic * 32, k); load_w8(msgs + ic * 32, m);
    const bool good = sign_item(k, m, fb, W, nwin, rx, ry, s, c_K);
    if (i < n) {
#pragma unroll
      for (int j = 0; j < 8; j++) { rx[j] = good ? rx[j] : 0u; ry[j] = good ? ry[j] : 0u; s[j] = good ? s[j] : 0u; }
      store_w8(out_r + i * 64, rx); store_w8(out_r + i * 64 + 32, ry); store_w8(out_s + i * 32, s);
      ok[i] = good ? 1 : 0;
    }
  }
}

// PrivateKey::sign_schnorr (src/lib.rs:344-361) with caller-supplied 1024-bit nonces (128 B each); s is the
// reference's unreduced integer k + scalar_key*h in a 160-byte little-endian record.
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_sign_schnorr(const u32* __restrict__ table, int W, int nwin,
                                                                   const uint8_t* __restrict__ keys,
                                                                   const uint8_t* __restrict__ msgs,
                                                                   const uint8_t* __restrict__ nonces, size_t n,
                                                                   uint8_t* __restrict__ out_r, uint8_t* __restrict__ out_s,
                                                                   uint8_t* __restrict__ ok) {
  __shared__ __attribute__((aligned(16))) u32 stage[(BJJ_BLOCK / 64) * FB_STAGE_WORDS];
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const GatherCoopLds<1> fb = {table, stage + (threadIdx.x >> 6) * FB_STAGE_WORDS, lane};
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i - lane < n; i += nthreads) {  // wave-uniform trip count
    const size_t ic = i < n ? i : n - 1;
    u32 k[8], m[8], rx[8], ry[8], kn[SCHNORR_K_WORDS], s[SCHNORR_S_WORDS];
    load_w8(keys + ic * 32, k); load_w8(msgs + ic * 32, m);
#pragma unroll
    for (int j = 0; j < SCHNORR_K_WORDS / 8; j++) load_w8(nonces + ic * (SCHNORR_K_WORDS * 4) + j * 32, kn + 8 * j);
    const bool good = sign_schnorr_item(k, m, kn, fb, W, nwin, rx, ry, s, c_K);
    if (i < n) {
#pragma unroll
      for (int j = 0; j < 8; j++) { rx[j] = good ? rx[j] : 0u; ry[j] = good ? ry[j] : 0u; }
#pragma unroll
      for (int j = 0; j < SCHNORR_S_WORDS; j++) s[j] = good ? s[j] : 0u;
      store_w8(out_r + i * 64, rx); store_w8(out_r + i * 64 + 32, ry);
#pragma unroll
      for (int j = 0; j < SCHNORR_S_WORDS / 8; j++) store_w8(out_s + i * (SCHNORR_S_WORDS * 4) + j * 32, s + 8 * j);
      ok[i] = good ? 1 : 0;
    }
  }
}

// ===========================================================================
// host side: context + extern "C" boundary
// ===========================================================================
static thread_local std::string g_err;
static int set_err(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCK(call)                                                                                  \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return set_err(BJJ_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                  \
  } while (0)

struct bjj_ctx {
  int device = 0;
  int cus = 0;
  int W = 16, nwin = 16;
  // resident workgroups per CU of each kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor):
  // grids are sized to exactly one resident wave of workgroups, items are grid-strided
  int occ_fixed = 1, occ_var = 1, occ_poseidon = 1, occ_verify = 1, occ_scan = 1, occ_add = 1;
  hipStream_t stream = nullptr;
  u32* table = nullptr;      // [window][digit 0 .. 2^(W-1)] x 128 B
  u32* bases = nullptr;      // P_j = 2^(W j) * B8, one Niels entry per window
  size_t table_bytes = 0;
  u32* scratch = nullptr;      // n * 64 B (Z, prefix)
  size_t scratch_items = 0;
  u32* vb_tables = nullptr;    // grid threads * VB_TABLE_WORDS * 4 B
  size_t vb_threads = 0;
  u32* slow = nullptr;         // [0] = count, [1..] item indices deferred to the exact-path kernels
  size_t slow_items = 0;
  uint8_t* codec = nullptr;    // verify_compressed: n * (64 pk + 64 R + 32 s + 2 flags) bytes
  size_t codec_items = 0;
  int occ_decomp = 1, occ_sign = 1, occ_sign_schnorr = 1;
  // host-pointer API: chunked pipeline  user memory -> pinned[b] -H2D-> dstage[b] -kernel-> dstage[b] -D2H-> pinned[b] -> user
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  uint8_t* pinned[2] = {nullptr, nullptr};
  uint8_t* dstage[2] = {nullptr, nullptr};
  size_t pipe_bytes = 0;
};

static int grid_for(const bjj_ctx* c, size_t n, int blocks_per_cu, int block = BJJ_BLOCK) {
  size_t want = (n + block - 1) / block;
  size_t cap = (size_t)c->cus * blocks_per_cu;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}
template <typename K>
static int occupancy_of(K kernel, int block) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) nb = 1;
  return nb;
}

static int ensure_scratch(bjj_ctx* c, size_t n) {
  HIPCK(hipSetDevice(c->device));
  if (n > c->scratch_items) {
    if (c->scratch) { HIPCK(hipStreamSynchronize(c->stream)); HIPCK(hipFree(c->scratch)); c->scratch = nullptr; }
    HIPCK(hipMalloc((void**)&c->scratch, n * 64));
    c->scratch_items = n;
  }
  if (n > c->slow_items) {
    if (c->slow) { HIPCK(hipStreamSynchronize(c->stream)); HIPCK(hipFree(c->slow)); c->slow = nullptr; }
    HIPCK(hipMalloc((void**)&c->slow, (n + 16) * sizeof(u32)));
    c->slow_items = n;
  }
  size_t tv = (size_t)c->occ_var * BJJ_EPI_BLOCK, te = (size_t)c->occ_verify * BJJ_BLOCK * 2;  // verify: 2 tables per lane
  size_t threads = (size_t)c->cus * (tv > te ? tv : te);
  if (threads > c->vb_threads) {
    if (c->vb_tables) { HIPCK(hipStreamSynchronize(c->stream)); HIPCK(hipFree(c->vb_tables)); c->vb_tables = nullptr; }
    HIPCK(hipMalloc((void**)&c->vb_tables, threads * VB_TABLE_WORDS * sizeof(u32)));
    c->vb_threads = threads;
  }
  return BJJ_OK;
}
// ---------------------------------------------------------------------------
// Host-pointer API plumbing.  Pageable hipMemcpy runs at ~3 GB/s and would dominate every call
// (29.8 ms of copies around a 1.1 ms kernel for 2^20 fixed-base multiplications), so batches are
// cut into chunks that flow through two pinned staging buffers: while the kernels of chunk c run
// on the context's stream, chunk c+1 is copied in (s_in) and chunk c-1 is copied out (s_out).
// Kernels of different chunks stay on ONE stream, so the context's scratch is never shared.
// ---------------------------------------------------------------------------
#define BJJ_PIPE_CHUNK ((size_t)1 << 18)
struct PipeSpec {
  int n_in, n_out;
  const uint8_t* in[4]; size_t in_stride[4];
  uint8_t* out[4];      size_t out_stride[4];
};
static size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }
// Staging copies between the caller's pageable memory and the pinned buffers: one thread moves ~30 GB/s, PCIe 54 GB/s
// (tools/ubench/pcie_probe.cpp), so copies of 4 MB and more are split over up to four threads.
static void staged_copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
  const size_t kMin = (size_t)4 << 20;
  unsigned hw = std::thread::hardware_concurrency();
  size_t parts = bytes / kMin;
  if (parts > 4) parts = 4;
  if (hw && parts > hw) parts = hw;
  if (parts < 2) { memcpy(dst, src, bytes); return; }
  const size_t per = (bytes / parts + 63) & ~(size_t)63;
  std::thread th[3];
  size_t started = 0;
  for (size_t i = 1; i < parts; i++) {
    const size_t lo = i * per, len = (i + 1 == parts) ? bytes - lo : per;
    try { th[started] = std::thread([=] { memcpy(dst + lo, src + lo, len); }); started++; }
    catch (...) { memcpy(dst + lo, src + lo, len); }   // no thread available: copy here
  }
  memcpy(dst, src, per);
  for (size_t i = 0; i < started; i++) th[i].join();
}
static int ensure_pipe(bjj_ctx* c, size_t bytes) {
  HIPCK(hipSetDevice(c->device));
  if (!c->s_in) {
    HIPCK(hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    HIPCK(hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) {
      HIPCK(hipEventCreateWithFlags(&c->ev_in[b], hipEventDisableTiming));
      HIPCK(hipEventCreateWithFlags(&c->ev_k[b], hipEventDisableTiming));
      HIPCK(hipEventCreateWithFlags(&c->ev_out[b], hipEventDisableTiming));
    }
  }
  if (bytes > c->pipe_bytes) {
    HIPCK(hipStreamSynchronize(c->stream));
    for (int b = 0; b < 2; b++) {
      if (c->pinned[b]) { HIPCK(hipHostFree(c->pinned[b])); c->pinned[b] = nullptr; }
      if (c->dstage[b]) { HIPCK(hipFree(c->dstage[b])); c->dstage[b] = nullptr; }
      HIPCK(hipHostMalloc((void**)&c->pinned[b], bytes, hipHostMallocDefault));
      HIPCK(hipMalloc((void**)&c->dstage[b], bytes));
    }
    c->pipe_bytes = bytes;
  }
  return BJJ_OK;
}
// launch(d_in[], d_out[], count, stream) enqueues the kernels of one chunk on `stream`
template <typename Launch>
static int run_pipelined(bjj_ctx* c, size_t n, const PipeSpec& sp, Launch launch) {
  const size_t chunk = n < BJJ_PIPE_CHUNK ? n : BJJ_PIPE_CHUNK;
  size_t off_in[4], off_out[4], tot = 0, in_bytes = 0;
  for (int i = 0; i < sp.n_in; i++) { off_in[i] = tot; tot += up16(chunk * sp.in_stride[i]); }
  in_bytes = tot;
  for (int i = 0; i < sp.n_out; i++) { off_out[i] = tot; tot += up16(chunk * sp.out_stride[i]); }
  int rc = ensure_pipe(c, tot); if (rc) return rc;
  const size_t nchunks = (n + chunk - 1) / chunk;
  auto drain = [&](size_t ch) -> int {  // copy the finished outputs of chunk `ch` to user memory
    const int b = (int)(ch & 1);
    const size_t lo = ch * chunk, cnt = (lo + chunk <= n ? chunk : n - lo);
    HIPCK(hipEventSynchronize(c->ev_out[b]));
    for (int i = 0; i < sp.n_out; i++) staged_copy(sp.out[i] + lo * sp.out_stride[i], c->pinned[b] + off_out[i], cnt * sp.out_stride[i]);
    return BJJ_OK;
  };
  for (size_t ch = 0; ch < nchunks; ch++) {
    const int b = (int)(ch & 1);
    const size_t lo = ch * chunk, cnt = (lo + chunk <= n ? chunk : n - lo);
    if (ch >= 2) { rc = drain(ch - 2); if (rc) return rc; }     // frees pinned[b] and dstage[b]
    for (int i = 0; i < sp.n_in; i++) staged_copy(c->pinned[b] + off_in[i], sp.in[i] + lo * sp.in_stride[i], cnt * sp.in_stride[i]);
    HIPCK(hipMemcpyAsync(c->dstage[b], c->pinned[b], in_bytes, hipMemcpyHostToDevice, c->s_in));
    HIPCK(hipEventRecord(c->ev_in[b], c->s_in));
    HIPCK(hipStreamWaitEvent(c->stream, c->ev_in[b], 0));
    void* d_in[4]; void* d_out[4];
    for (int i = 0; i < sp.n_in; i++) d_in[i] = c->dstage[b] + off_in[i];
    for (int i = 0; i < sp.n_out; i++) d_out[i] = c->dstage[b] + off_out[i];
    rc = launch(d_in, d_out, cnt, (void*)c->stream); if (rc) return rc;
    HIPCK(hipEventRecord(c->ev_k[b], c->stream));
    HIPCK(hipStreamWaitEvent(c->s_out, c->ev_k[b], 0));
    HIPCK(hipMemcpyAsync(c->pinned[b] + in_bytes, c->dstage[b] + in_bytes, tot - in_bytes, hipMemcpyDeviceToHost, c->s_out));
    HIPCK(hipEventRecord(c->ev_out[b], c->s_out));
  }
  if (nchunks >= 2) { rc = drain(nchunks - 2); if (rc) return rc; }
  return drain(nchunks - 1);
}
static int ensure_codec(bjj_ctx* c, size_t n) {  // 162 bytes per item of intermediate records
  HIPCK(hipSetDevice(c->device));
  if (n > c->codec_items) {
    if (c->codec) { HIPCK(hipStreamSynchronize(c->stream)); HIPCK(hipFree(c->codec)); c->codec = nullptr; }
    HIPCK(hipMalloc((void**)&c->codec, n * 162 + 64));
    c->codec_items = n;
  }
  return BJJ_OK;
}
static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" {
#pragma GCC visibility push(default)

const char* bjj_version(void) { return BJJ_VERSION_STRING; }
const char* bjj_last_error(void) { return g_err.c_str(); }

int bjj_init(int device, int window_bits, bjj_ctx** out_ctx) {
  if (!out_ctx) return set_err(BJJ_E_INVALID, "bjj_init: out_ctx is NULL");
  *out_ctx = nullptr;
  if (window_bits != 0 && (window_bits < 4 || window_bits > BJJ_MAX_WINDOW_BITS))
    return set_err(BJJ_E_INVALID, "bjj_init: window_bits must be 0 (auto) or 4..28");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return set_err(BJJ_E_NO_DEVICE, "bjj_init: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return set_err(BJJ_E_INVALID, "bjj_init: device index out of range");
  HIPCK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, device));
  bjj_ctx* c = new (std::nothrow) bjj_ctx();
  if (!c) return set_err(BJJ_E_NOMEM, "bjj_init: out of host memory");
  c->device = device;
  c->cus = prop.multiProcessorCount;
  int W = window_bits;
  if (W == 0) {  // auto: widest table that leaves 40 % of the free memory to the caller (a second context gets 26 bits)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    W = kAutoWindowBits[sizeof(kAutoWindowBits) / sizeof(int) - 1];
    for (int cand : kAutoWindowBits) {
      const size_t need = fixed_stride(cand) * (size_t)fixed_nwin(cand) * NIELS_WORDS * sizeof(u32);
      if (need <= free_b / 5 * 3) { W = cand; break; }
    }
  }
  c->W = W;
  c->nwin = fixed_nwin(W);
  c->occ_fixed = occupancy_of(bjj_k_mul_fixed_base, BJJ_EPI_BLOCK);
  c->occ_var = occupancy_of(bjj_k_mul_var_base, BJJ_EPI_BLOCK);
  c->occ_poseidon = occupancy_of(bjj_k_poseidon5, BJJ_BLOCK);
  c->occ_verify = occupancy_of(bjj_k_eddsa_verify, BJJ_BLOCK);
  c->occ_scan = occupancy_of(bjj_k_eddsa_verify_scan, BJJ_BLOCK);
  c->occ_add = occupancy_of(bjj_k_point_add, BJJ_BLOCK);
  c->occ_decomp = occupancy_of(bjj_k_decompress_points, BJJ_BLOCK);
  c->occ_sign = occupancy_of(bjj_k_sign, BJJ_BLOCK);
  c->occ_sign_schnorr = occupancy_of(bjj_k_sign_schnorr, BJJ_BLOCK);
  hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (se != hipSuccess) { delete c; return set_err(BJJ_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(se)); }
  size_t entries = 0;
  for (;;) {
    entries = fixed_stride(c->W) * (size_t)c->nwin;
    c->table_bytes = entries * NIELS_WORDS * sizeof(u32);
    se = hipMalloc((void**)&c->table, c->table_bytes);
    if (se == hipSuccess || window_bits != 0) break;
    (void)hipGetLastError();   // auto mode: the free-memory estimate was too optimistic, take the next narrower table
    int next = 0;
    for (int cand : kAutoWindowBits) if (cand < c->W) { next = cand; break; }
    if (!next) break;
    c->W = W = next;
    c->nwin = fixed_nwin(W);
    c->table = nullptr;
  }
  if (se == hipSuccess) se = hipMalloc((void**)&c->bases, (size_t)c->nwin * NIELS_WORDS * sizeof(u32));
  if (se != hipSuccess) {
    if (c->table) hipFree(c->table);
    hipStreamDestroy(c->stream); delete c;
    return set_err(BJJ_E_NOMEM, "bjj_init: cannot allocate the fixed-base table");
  }
  // chain length: long enough to amortise the start ladder and the inversion, short enough to fill the GPU
  size_t chain = entries >> 18;
  chain = chain < 4 ? 4 : (chain > 256 ? 256 : chain);
  const size_t chains = ((fixed_stride(W) + chain - 1) / chain) * (size_t)c->nwin;
  hipLaunchKernelGGL(bjj_k_fixed_window_bases, dim3((c->nwin + 63) / 64), dim3(64), 0, c->stream, c->bases, c->W, c->nwin);
  hipLaunchKernelGGL(bjj_k_build_fixed_table, dim3((unsigned)((chains + BJJ_BLOCK - 1) / BJJ_BLOCK)), dim3(BJJ_BLOCK), 0, c->stream,
                     c->table, c->bases, c->W, c->nwin, (u32)chain);
  se = hipGetLastError();
  if (se == hipSuccess) se = hipStreamSynchronize(c->stream);
  if (se != hipSuccess) {
    hipFree(c->table); hipFree(c->bases); hipStreamDestroy(c->stream); delete c;
    return set_err(BJJ_E_HIP, std::string("bjj_init: table build failed: ") + hipGetErrorString(se));
  }
  *out_ctx = c;
  return BJJ_OK;
}

void bjj_free(bjj_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->table) hipFree(c->table);
  if (c->bases) hipFree(c->bases);
  if (c->scratch) hipFree(c->scratch);
  if (c->vb_tables) hipFree(c->vb_tables);
  if (c->slow) hipFree(c->slow);
  if (c->codec) hipFree(c->codec);
  for (int b = 0; b < 2; b++) {
    if (c->pinned[b]) hipHostFree(c->pinned[b]);
    if (c->dstage[b]) hipFree(c->dstage[b]);
    if (c->ev_in[b]) hipEventDestroy(c->ev_in[b]);
    if (c->ev_k[b]) hipEventDestroy(c->ev_k[b]);
    if (c->ev_out[b]) hipEventDestroy(c->ev_out[b]);
  }
  if (c->s_in) hipStreamDestroy(c->s_in);
  if (c->s_out) hipStreamDestroy(c->s_out);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int bjj_sync(bjj_ctx* c) {
  if (!c) return set_err(BJJ_E_INVALID, "bjj_sync: ctx is NULL");
  HIPCK(hipStreamSynchronize(c->stream));
  return BJJ_OK;
}
void* bjj_stream(bjj_ctx* c) { return c ? (void*)c->stream : nullptr; }

int bjj_reserve(bjj_ctx* c, size_t n) {
  if (!c) return set_err(BJJ_E_INVALID, "bjj_reserve: ctx is NULL");
  return ensure_scratch(c, n ? n : 1);
}

int bjj_get_info(bjj_ctx* c, bjj_info* info) {
  if (!c || !info) return set_err(BJJ_E_INVALID, "bjj_get_info: NULL argument");
  info->device = c->device;
  info->compute_units = c->cus;
  info->window_bits = c->W;
  info->n_windows = c->nwin;
  info->table_bytes = c->table_bytes;
  info->scratch_bytes = c->scratch_items * 64 + c->vb_threads * VB_TABLE_WORDS * sizeof(u32) + c->slow_items * 4;
  info->kernel_fixed_base = "bjj_k_mul_fixed_base";
  info->kernel_var_base = "bjj_k_mul_var_base";
  info->kernel_poseidon5 = "bjj_k_poseidon5";
  info->kernel_verify = "bjj_k_eddsa_verify";
  return BJJ_OK;
}

int bjj_check_table(bjj_ctx* c, uint64_t* n_bad) {
  if (!c || !n_bad) return set_err(BJJ_E_INVALID, "bjj_check_table: NULL argument");
  HIPCK(hipSetDevice(c->device));
  unsigned long long* d_bad = nullptr;
  HIPCK(hipMalloc((void**)&d_bad, sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(bjj_k_check_fixed_table, dim3((unsigned)(c->cus * 8)), dim3(BJJ_BLOCK), 0, c->stream, c->table, c->bases, c->W,
                       c->nwin, d_bad);
    e = hipGetLastError();
  }
  unsigned long long h = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d_bad, sizeof(h), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_bad);
  if (e != hipSuccess) return set_err(BJJ_E_HIP, std::string("bjj_check_table: ") + hipGetErrorString(e));
  *n_bad = (uint64_t)h;
  return BJJ_OK;
}

// ---- device-pointer API ------------------------------------------------------
// item indices travel as 32-bit words in the work lists and cursors
#define CHECK_N(n) if ((n) >> 32) return set_err(BJJ_E_INVALID, "batches are limited to 2^32 - 1 items per call")
#define CHECK_CTX(c, name) if (!(c)) return set_err(BJJ_E_INVALID, name ": ctx is NULL")
#define CHECK_PTR(p, name) if (!(p) || !aligned16(p)) return set_err(BJJ_E_INVALID, name ": NULL or not 16-byte aligned device pointer")

int bjj_mul_fixed_base_dev(bjj_ctx* c, const void* d_scalars, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_mul_fixed_base_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_scalars, "bjj_mul_fixed_base_dev"); CHECK_PTR(d_out, "bjj_mul_fixed_base_dev");
  int rc = ensure_scratch(c, n); if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(bjj_k_mul_fixed_base, dim3(grid_for(c, n, c->occ_fixed, BJJ_EPI_BLOCK)), dim3(BJJ_EPI_BLOCK), 0, st, c->table, c->W,
                     c->nwin, (const uint8_t*)d_scalars, n, (uint8_t*)d_out, c->scratch);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}
int bjj_mul_var_base_dev(bjj_ctx* c, const void* d_pts, const void* d_scalars, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_mul_var_base_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_pts, "bjj_mul_var_base_dev"); CHECK_PTR(d_scalars, "bjj_mul_var_base_dev"); CHECK_PTR(d_out, "bjj_mul_var_base_dev");
  int rc = ensure_scratch(c, n); if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  HIPCK(hipMemsetAsync(c->slow, 0, 8 * sizeof(u32), st));
  hipLaunchKernelGGL(bjj_k_mul_var_base, dim3(grid_for(c, n, c->occ_var, BJJ_EPI_BLOCK)), dim3(BJJ_EPI_BLOCK), 0, st, (const uint8_t*)d_pts,
                     (const uint8_t*)d_scalars, n, (uint8_t*)d_out, c->scratch, c->vb_tables, c->slow);
  HIPCK(hipGetLastError());
  hipLaunchKernelGGL(bjj_k_mul_var_base_exact, dim3(c->cus * 4), dim3(64), 0, st, (const uint8_t*)d_pts,
                     (const uint8_t*)d_scalars, (uint8_t*)d_out, c->slow);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}
int bjj_poseidon5_dev(bjj_ctx* c, const void* d_in, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_poseidon5_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_in, "bjj_poseidon5_dev"); CHECK_PTR(d_out, "bjj_poseidon5_dev");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(bjj_k_poseidon5, dim3(grid_for(c, n, c->occ_poseidon)), dim3(BJJ_BLOCK), 0, st, (const uint8_t*)d_in, n,
                     (uint8_t*)d_out);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}
static int verify_launch(bjj_ctx* c, bool schnorr, const void* d_pk, const void* d_r, const void* d_s, const void* d_msg,
                         size_t n, void* d_ok, void* stream, const char* who) {
  if (!c) return set_err(BJJ_E_INVALID, std::string(who) + ": ctx is NULL");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  if (!d_pk || !d_r || !d_s || !d_msg || !aligned16(d_pk) || !aligned16(d_r) || !aligned16(d_s) || !aligned16(d_msg))
    return set_err(BJJ_E_INVALID, std::string(who) + ": NULL or not 16-byte aligned device pointer");
  if (!d_ok) return set_err(BJJ_E_INVALID, std::string(who) + ": d_ok is NULL");
  int rc = ensure_scratch(c, n); if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  HIPCK(hipMemsetAsync(c->slow, 0, WL_HDR * sizeof(u32), st));
  hipLaunchKernelGGL(bjj_k_eddsa_verify_scan, dim3(grid_for(c, n, c->occ_scan)), dim3(BJJ_BLOCK), 0, st,
                     (const uint8_t*)d_pk, (const uint8_t*)d_r, (const uint8_t*)d_msg, n, c->slow);
  HIPCK(hipGetLastError());
  if (schnorr)
    hipLaunchKernelGGL(bjj_k_schnorr_verify, dim3(grid_for(c, n, c->occ_verify)), dim3(BJJ_BLOCK), 0, st, c->table, c->W,
                       c->nwin, (const uint8_t*)d_pk, (const uint8_t*)d_r, (const uint8_t*)d_s, (const uint8_t*)d_msg, n,
                       (uint8_t*)d_ok, c->vb_tables, c->slow);
  else
    hipLaunchKernelGGL(bjj_k_eddsa_verify, dim3(grid_for(c, n, c->occ_verify)), dim3(BJJ_BLOCK), 0, st, c->table, c->W,
                       c->nwin, (const uint8_t*)d_pk, (const uint8_t*)d_r, (const uint8_t*)d_s, (const uint8_t*)d_msg, n,
                       (uint8_t*)d_ok, c->vb_tables, c->slow);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}
int bjj_eddsa_verify_dev(bjj_ctx* c, const void* d_pk, const void* d_r, const void* d_s, const void* d_msg, size_t n,
                         void* d_ok, void* stream) {
  return verify_launch(c, false, d_pk, d_r, d_s, d_msg, n, d_ok, stream, "bjj_eddsa_verify_dev");
}
int bjj_schnorr_verify_dev(bjj_ctx* c, const void* d_pk, const void* d_r, const void* d_s, const void* d_msg, size_t n,
                           void* d_ok, void* stream) {
  return verify_launch(c, true, d_pk, d_r, d_s, d_msg, n, d_ok, stream, "bjj_schnorr_verify_dev");
}
int bjj_point_add_dev(bjj_ctx* c, const void* d_p, const void* d_q, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_point_add_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_p, "bjj_point_add_dev"); CHECK_PTR(d_q, "bjj_point_add_dev"); CHECK_PTR(d_out, "bjj_point_add_dev");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(bjj_k_point_add, dim3(grid_for(c, n, c->occ_add)), dim3(BJJ_BLOCK), 0, st, (const uint8_t*)d_p,
                     (const uint8_t*)d_q, n, (uint8_t*)d_out);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}

int bjj_compress_points_dev(bjj_ctx* c, const void* d_pts, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_compress_points_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_pts, "bjj_compress_points_dev"); CHECK_PTR(d_out, "bjj_compress_points_dev");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(bjj_k_compress_points, dim3(grid_for(c, n, 8)), dim3(BJJ_BLOCK), 0, st, (const uint8_t*)d_pts, n,
                     (uint8_t*)d_out);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}
int bjj_decompress_points_dev(bjj_ctx* c, const void* d_in, size_t n, void* d_out_xy, void* d_ok, void* stream) {
  CHECK_CTX(c, "bjj_decompress_points_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_in, "bjj_decompress_points_dev"); CHECK_PTR(d_out_xy, "bjj_decompress_points_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_decompress_points_dev: d_ok is NULL");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(bjj_k_decompress_points, dim3(grid_for(c, n, c->occ_decomp)), dim3(BJJ_BLOCK), 0, st,
                     (const uint8_t*)d_in, (size_t)32, n, (uint8_t*)d_out_xy, (uint8_t*)d_ok, (uint8_t*)nullptr);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}
int bjj_eddsa_verify_compressed_dev(bjj_ctx* c, const void* d_pk32, const void* d_sig64, const void* d_msg, size_t n,
                                    void* d_ok, void* stream) {
  CHECK_CTX(c, "bjj_eddsa_verify_compressed_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_pk32, "bjj_eddsa_verify_compressed_dev"); CHECK_PTR(d_sig64, "bjj_eddsa_verify_compressed_dev");
  CHECK_PTR(d_msg, "bjj_eddsa_verify_compressed_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_eddsa_verify_compressed_dev: d_ok is NULL");
  int rc0 = ensure_codec(c, n); if (rc0) return rc0;
  uint8_t* pk_xy = c->codec;
  uint8_t* r_xy = pk_xy + n * 64;
  uint8_t* s32 = r_xy + n * 64;
  uint8_t* f_pk = s32 + n * 32;
  uint8_t* f_r = f_pk + n;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  const int g = grid_for(c, n, c->occ_decomp);
  hipLaunchKernelGGL(bjj_k_decompress_points, dim3(g), dim3(BJJ_BLOCK), 0, st, (const uint8_t*)d_pk32, (size_t)32, n,
                     pk_xy, f_pk, (uint8_t*)nullptr);
  hipLaunchKernelGGL(bjj_k_decompress_points, dim3(g), dim3(BJJ_BLOCK), 0, st, (const uint8_t*)d_sig64, (size_t)64, n,
                     r_xy, f_r, s32);
  HIPCK(hipGetLastError());
  int rc = bjj_eddsa_verify_dev(c, pk_xy, r_xy, s32, d_msg, n, d_ok, (void*)st);
  if (rc) return rc;
  hipLaunchKernelGGL(bjj_k_merge_codec_flags, dim3(grid_for(c, n, 8)), dim3(BJJ_BLOCK), 0, st, (uint8_t*)d_ok, f_pk, f_r, n);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}

int bjj_scalar_keys_dev(bjj_ctx* c, const void* d_keys, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_scalar_keys_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_keys, "bjj_scalar_keys_dev"); CHECK_PTR(d_out, "bjj_scalar_keys_dev");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(bjj_k_scalar_keys, dim3(grid_for(c, n, 4)), dim3(BJJ_BLOCK), 0, st, (const uint8_t*)d_keys, n,
                     (uint8_t*)d_out);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}
int bjj_public_keys_dev(bjj_ctx* c, const void* d_keys, size_t n, void* d_out_xy, void* stream) {
  CHECK_CTX(c, "bjj_public_keys_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_keys, "bjj_public_keys_dev"); CHECK_PTR(d_out_xy, "bjj_public_keys_dev");
  int rc = ensure_codec(c, n); if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  rc = bjj_scalar_keys_dev(c, d_keys, n, c->codec, (void*)st); if (rc) return rc;   // B8.mul_scalar(&self.scalar_key())
  return bjj_mul_fixed_base_dev(c, c->codec, n, d_out_xy, (void*)st);
}
int bjj_sign_dev(bjj_ctx* c, const void* d_keys, const void* d_msgs, size_t n, void* d_out_r, void* d_out_s, void* d_ok,
                 void* stream) {
  CHECK_CTX(c, "bjj_sign_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_keys, "bjj_sign_dev"); CHECK_PTR(d_msgs, "bjj_sign_dev"); CHECK_PTR(d_out_r, "bjj_sign_dev");
  CHECK_PTR(d_out_s, "bjj_sign_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_sign_dev: d_ok is NULL");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(bjj_k_sign, dim3(grid_for(c, n, c->occ_sign)), dim3(BJJ_BLOCK), 0, st, c->table, c->W, c->nwin,
                     (const uint8_t*)d_keys, (const uint8_t*)d_msgs, n, (uint8_t*)d_out_r, (uint8_t*)d_out_s, (uint8_t*)d_ok);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}

int bjj_sign_schnorr_dev(bjj_ctx* c, const void* d_keys, const void* d_msgs, const void* d_nonces, size_t n, void* d_out_r,
                         void* d_out_s, void* d_ok, void* stream) {
  CHECK_CTX(c, "bjj_sign_schnorr_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_keys, "bjj_sign_schnorr_dev"); CHECK_PTR(d_msgs, "bjj_sign_schnorr_dev"); CHECK_PTR(d_nonces, "bjj_sign_schnorr_dev");
  CHECK_PTR(d_out_r, "bjj_sign_schnorr_dev"); CHECK_PTR(d_out_s, "bjj_sign_schnorr_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_sign_schnorr_dev: d_ok is NULL");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(bjj_k_sign_schnorr, dim3(grid_for(c, n, c->occ_sign_schnorr)), dim3(BJJ_BLOCK), 0, st, c->table, c->W, c->nwin,
                     (const uint8_t*)d_keys, (const uint8_t*)d_msgs, (const uint8_t*)d_nonces, n, (uint8_t*)d_out_r,
                     (uint8_t*)d_out_s, (uint8_t*)d_ok);
  HIPCK(hipGetLastError());
  return BJJ_OK;
}

// ---- host-pointer API: chunked pinned-staging pipeline around the *_dev entry points ----------
#define HOST_PROLOGUE(name, cond)                                             \
  CHECK_CTX(c, name);                                                         \
  if (n == 0) return BJJ_OK;                                                  \
  if (cond) return set_err(BJJ_E_INVALID, name ": NULL buffer")

int bjj_mul_fixed_base(bjj_ctx* c, const uint8_t* scalars, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_mul_fixed_base", !scalars || !out);
  PipeSpec sp = {1, 1, {scalars}, {32}, {out}, {64}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_mul_fixed_base_dev(c, i[0], cnt, o[0], st); });
}
int bjj_mul_var_base(bjj_ctx* c, const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_mul_var_base", !pts || !scalars || !out);
  PipeSpec sp = {2, 1, {pts, scalars}, {64, 32}, {out}, {64}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_mul_var_base_dev(c, i[0], i[1], cnt, o[0], st); });
}
int bjj_poseidon5(bjj_ctx* c, const uint8_t* in, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_poseidon5", !in || !out);
  PipeSpec sp = {1, 1, {in}, {160}, {out}, {32}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_poseidon5_dev(c, i[0], cnt, o[0], st); });
}
int bjj_eddsa_verify(bjj_ctx* c, const uint8_t* pk, const uint8_t* r, const uint8_t* s, const uint8_t* msg, size_t n,
                     uint8_t* ok) {
  HOST_PROLOGUE("bjj_eddsa_verify", !pk || !r || !s || !msg || !ok);
  PipeSpec sp = {4, 1, {pk, r, s, msg}, {64, 64, 32, 32}, {ok}, {1}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_eddsa_verify_dev(c, i[0], i[1], i[2], i[3], cnt, o[0], st); });
}
int bjj_schnorr_verify(bjj_ctx* c, const uint8_t* pk, const uint8_t* r, const uint8_t* s, const uint8_t* msg, size_t n,
                       uint8_t* ok) {
  HOST_PROLOGUE("bjj_schnorr_verify", !pk || !r || !s || !msg || !ok);
  PipeSpec sp = {4, 1, {pk, r, s, msg}, {64, 64, 32, 32}, {ok}, {1}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_schnorr_verify_dev(c, i[0], i[1], i[2], i[3], cnt, o[0], st); });
}
int bjj_point_add(bjj_ctx* c, const uint8_t* p, const uint8_t* q, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_point_add", !p || !q || !out);
  PipeSpec sp = {2, 1, {p, q}, {64, 64}, {out}, {64}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_point_add_dev(c, i[0], i[1], cnt, o[0], st); });
}
int bjj_compress_points(bjj_ctx* c, const uint8_t* pts, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_compress_points", !pts || !out);
  PipeSpec sp = {1, 1, {pts}, {64}, {out}, {32}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_compress_points_dev(c, i[0], cnt, o[0], st); });
}
int bjj_decompress_points(bjj_ctx* c, const uint8_t* in, size_t n, uint8_t* out_xy, uint8_t* ok) {
  HOST_PROLOGUE("bjj_decompress_points", !in || !out_xy || !ok);
  PipeSpec sp = {1, 2, {in}, {32}, {out_xy, ok}, {64, 1}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_decompress_points_dev(c, i[0], cnt, o[0], o[1], st); });
}
int bjj_eddsa_verify_compressed(bjj_ctx* c, const uint8_t* pk32, const uint8_t* sig64, const uint8_t* msg, size_t n,
                                uint8_t* ok) {
  HOST_PROLOGUE("bjj_eddsa_verify_compressed", !pk32 || !sig64 || !msg || !ok);
  PipeSpec sp = {3, 1, {pk32, sig64, msg}, {32, 64, 32}, {ok}, {1}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_eddsa_verify_compressed_dev(c, i[0], i[1], i[2], cnt, o[0], st); });
}
int bjj_scalar_keys(bjj_ctx* c, const uint8_t* keys, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_scalar_keys", !keys || !out);
  PipeSpec sp = {1, 1, {keys}, {32}, {out}, {32}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_scalar_keys_dev(c, i[0], cnt, o[0], st); });
}
int bjj_public_keys(bjj_ctx* c, const uint8_t* keys, size_t n, uint8_t* out_xy) {
  HOST_PROLOGUE("bjj_public_keys", !keys || !out_xy);
  PipeSpec sp = {1, 1, {keys}, {32}, {out_xy}, {64}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_public_keys_dev(c, i[0], cnt, o[0], st); });
}
int bjj_sign(bjj_ctx* c, const uint8_t* keys, const uint8_t* msgs, size_t n, uint8_t* out_r, uint8_t* out_s, uint8_t* ok) {
  HOST_PROLOGUE("bjj_sign", !keys || !msgs || !out_r || !out_s || !ok);
  PipeSpec sp = {2, 3, {keys, msgs}, {32, 32}, {out_r, out_s, ok}, {64, 32, 1}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_sign_dev(c, i[0], i[1], cnt, o[0], o[1], o[2], st); });
}
int bjj_sign_schnorr(bjj_ctx* c, const uint8_t* keys, const uint8_t* msgs, const uint8_t* nonces, size_t n, uint8_t* out_r,
                     uint8_t* out_s, uint8_t* ok) {
  HOST_PROLOGUE("bjj_sign_schnorr", !keys || !msgs || !nonces || !out_r || !out_s || !ok);
  PipeSpec sp = {3, 3, {keys, msgs, nonces}, {32, 32, BJJ_SCHNORR_NONCE_BYTES}, {out_r, out_s, ok}, {64, BJJ_SCHNORR_S_BYTES, 1}};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_sign_schnorr_dev(c, i[0], i[1], i[2], cnt, o[0], o[1], o[2], st); });
}

#pragma GCC visibility pop
}  // extern "C"
