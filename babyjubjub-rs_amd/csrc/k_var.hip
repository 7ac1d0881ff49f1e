// libbjj_hip.so, kernel unit 2: K2 / K6 Point::mul_scalar for arbitrary points (src/lib.rs:149-164) and the raw
// PointProjective::add / affine (src/lib.rs:88-131, 70-85).
// per-lane table entries of this unit: packed, 128 B (bjj_device.hpp "per-lane variable-base table")
#ifndef BJJ_K2_PNIELS_LAYOUT
#define BJJ_K2_PNIELS_LAYOUT 1
#endif
#define BJJ_PNIELS_LAYOUT BJJ_K2_PNIELS_LAYOUT
#include "k_common.hpp"

// Workgroup size / resident workgroups per CU of K2.  Three 256-lane workgroups per CU = 3 waves per SIMD at 168 VGPRs
// (+112 B of scratch per lane) issue 1.2-2.5 % more than one 512-lane workgroup = 2 waves per SIMD at 229 VGPRs, although
// 2^20 items no longer divide evenly over the 196 608 resident lanes; 4 waves per SIMD (128 VGPRs, 400 B of scratch) lose
// 2 % (profiles/r02_ab_occupancy.txt).  K1 and the verify kernel were measured the same way and stay at 2 waves per SIMD.
#ifndef BJJ_K2_BLOCK
#define BJJ_K2_BLOCK 256
#endif
#ifndef BJJ_K2_MIN_BLOCKS
#define BJJ_K2_MIN_BLOCKS 3
#endif
// K2 comes in two forms, both compiled, picked per call by the host (bjj_hip.hip: var_base_variant):
//   0  one resident set of workgroups, items grid-strided (rounds 1-2): best while ANOTHER launch of the context is in flight
//      -- its long-lived workgroups let the other launch fill the tail, and the two launches' per-lane tables are not all
//      live at once (2 x 200 MB of tables interleaved would fall out of the 256 MB Infinity Cache);
//   1  one 256-item tile per workgroup (below): best for a launch that runs alone -- the hardware balances 4 096 short
//      workgroups over the chip, so a 2^20-item launch costs ~5.4 item-times instead of 6.
// Interleaved A/B, one MI355X (profiles/r03_ab_k2_tiles.txt): alone 15.0 vs 15.4 ms (tiles +2.6 %); on two streams 72.7 vs
// 74.1 M mults/s (strided +1.9 %).  BJJ_K2_VARIANT=0|1 forces one form.

// (n >> 3) mod l of a little-endian integer of nw words, as 8 words -> n mod 8l = 8*that + (n & 7) < 2^254.
// Horner over 261-bit chunks, most significant first: acc <- acc * 2^261 + chunk (mod l), with the mod-l Montgomery
// products of the signer row (fl_mul(acc, 2^522) = acc * 2^261, fl_mul(chunk, 2^261) = chunk; each < 2l).
__device__ void wide_scalar_mod_order(const u32* __restrict__ w, int nw, u32 out[8]) {
  const int bits = nw * 32 - 3;
  const int chunks = (bits + 260) / 261;
  Fr acc = fr_zero();
#pragma unroll 1
  for (int c = chunks - 1; c >= 0; c--) {
    Fr hi = fl_mul(acc, c_K.L_R2, c_K);
    Fr lo = fl_mul(limbs_from_bits(w, nw, 3 + 261 * c), c_K.L_R1, c_K);
    acc = fl_canon4(fr_add(hi, lo), c_K);
  }
  u32 q[8];
  fr_to_words(acc, q);   // < l < 2^251
  out[0] = (q[0] << 3) | (w[0] & 7u);
#pragma unroll
  for (int i = 1; i < 8; i++) out[i] = (q[i] << 3) | (q[i - 1] >> 29);
}

// ---------------------------------------------------------------------------
// K2: variable base.  An off-curve point (the reference's Point has pub fields and no check, src/lib.rs:134-138) is not this
// kernel's item: K6 (bjj_k_mul_var_base_exact) replays the reference's loop for it and owns its output slot, which K2 never
// touches (epilogue_stash_skipped) -- so K6 may run BESIDE K2.  Two ways the list of those items comes about:
//   slow != nullptr   K2 appends them as it meets them (slow[0] = count, item indices from slow[8]); K6 follows on the same stream.
//                     What a batch of on-curve points costs nothing extra for; one off-curve item then adds K6's ~4.5 ms
//                     (a strictly serial 254-step chain on one lane) BEHIND the launch.
//   slow == nullptr   somebody else has made the list (bjj_k_var_base_scan, on another stream) and K6 is already running.
// WIDE: scalars are records of `sc_words` 32-bit words (a multiple of 8; `n: &BigInt` is unbounded, src/lib.rs:149,
// 156-157); for an on-curve point n*P == (n mod 8l)*P exactly.
// ---------------------------------------------------------------------------
// The items of one lane are tid, tid + nthreads, ... below n; tbl = this lane's table scratch.
template <bool WIDE, bool APART = false>
__device__ __forceinline__ void var_base_body(const uint8_t* __restrict__ pts, const uint8_t* __restrict__ scalars, int sc_words,
                                              size_t n, uint8_t* __restrict__ out, u32* __restrict__ scratch,
                                              u32* __restrict__ tbl, u32* __restrict__ slow, u32* lds, size_t tid, size_t nthreads,
                                              uint8_t* __restrict__ xy) {
  // APART (the *_zc kernels): X, Y of phase 1 are stashed in `xy` and `out` is written once, by phase 2, and never read -- `out` may
  // then be the caller's pinned host array behind its device mapping (the host-pointer pipeline: no copy-out stage, bjj_hip.hip).
  // A compile-time form: the kernels every device-pointer caller gets do not carry the second pointer.
  uint8_t* const stash = APART ? xy : out;
  Fr run = fr_one();
#pragma unroll 1
  for (size_t i = tid; i < n; i += nthreads) {
    u32 w[8], sc[8];
    load_w8(pts + i * 64, w);      Fr x = fr_to_mont_words(w);
    load_w8(pts + i * 64 + 32, w); Fr y = fr_to_mont_words(w);
    if (ref_on_curve(x, y, c_K)) {
      Ext p;
      if (WIDE) {
        wide_scalar_mod_order((const u32*)(scalars + i * (size_t)sc_words * 4), sc_words, sc);
        Ext P = ext_from_ref_affine(x, y, c_K);
        vb_build_table(P, tbl, c_K, true);
        p = vb_mul_windowed(tbl, sc, 64, c_K);
      } else {
        load_w8(scalars + i * 32, sc);
        p = var_base_fast(x, y, sc, tbl, c_K);
      }
      epilogue_stash(p, run, stash + i * 64, scratch + i * 16);
    } else {
      if (slow) slow[8 + atomicAdd(&slow[0], 1u)] = (u32)i;
      epilogue_stash_skipped(scratch + i * 16);                 // K6's item: its output slot is not ours
    }
  }
  epilogue_run<BJJ_K2_BLOCK, APART ? (EPI_SKIPPABLE | EPI_STASH_APART) : EPI_SKIPPABLE>(run, n, tid, nthreads, out, scratch, lds, xy);
}
__global__ void __launch_bounds__(BJJ_K2_BLOCK, BJJ_K2_MIN_BLOCKS) bjj_k_mul_var_base(const uint8_t* __restrict__ pts,
                                                                const uint8_t* __restrict__ scalars, size_t n,
                                                                uint8_t* __restrict__ out, u32* __restrict__ scratch,
                                                                u32* __restrict__ vb_tables, u32* __restrict__ slow) {
  __shared__ u32 lds[NL * 64];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  var_base_body<false>(pts, scalars, 8, n, out, scratch, vb_tables + tid * VB_TABLE_WORDS, slow, lds, tid, (size_t)gridDim.x * blockDim.x, nullptr);
}
__global__ void __launch_bounds__(BJJ_K2_BLOCK, BJJ_K2_MIN_BLOCKS) bjj_k_mul_var_base_wide(const uint8_t* __restrict__ pts,
                                                                     const uint8_t* __restrict__ scalars, int sc_words, size_t n,
                                                                     uint8_t* __restrict__ out, u32* __restrict__ scratch,
                                                                     u32* __restrict__ vb_tables, u32* __restrict__ slow) {
  __shared__ u32 lds[NL * 64];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  var_base_body<true>(pts, scalars, sc_words, n, out, scratch, vb_tables + tid * VB_TABLE_WORDS, slow, lds, tid, (size_t)gridDim.x * blockDim.x, nullptr);
}
// ---- dispatch mode 1: one tile of BJJ_K2_BLOCK consecutive items per workgroup ------------------------------------------
// The grid-strided form above is one resident set of workgroups whose lanes own 5 or 6 items of a 2^20-item batch: the
// launch costs 6 item-times, and a second launch on another stream can only use what the 5-item workgroups free at the end.
// Here a workgroup does ONE item per lane (its own workgroup-wide inversion: one wave inverts while the CU's other two
// workgroups, which are in other phases, keep the SIMDs busy), retires, and the hardware dispatches the next tile -- of
// this launch or of the other one in flight: the pair is work-conserving.  The per-lane table scratch comes from a
// per-XCD slot queue (k_common.hpp), one slot = the tables of one workgroup.
template <bool WIDE, bool APART = false>
__device__ __forceinline__ void var_base_tile(const uint8_t* __restrict__ pts, const uint8_t* __restrict__ scalars, int sc_words, size_t n,
                                              uint8_t* __restrict__ out, u32* __restrict__ scratch, u32* __restrict__ vb_tables,
                                              u32* __restrict__ slow, u32* __restrict__ slotq, u32 cap_nx, uint8_t* __restrict__ xy) {
  __shared__ u32 lds[NL * 64];
  __shared__ u32 sh_slot;
  u32* q = slot_queue_of_this_xcd(slotq, cap_nx);
  if (threadIdx.x == 0) sh_slot = slot_pop_one(q, cap_nx);
  __syncthreads();
  const u32 slot = sh_slot;
  const size_t base = (size_t)blockIdx.x * BJJ_K2_BLOCK;
  const size_t hi = base + BJJ_K2_BLOCK < n ? base + BJJ_K2_BLOCK : n;
  var_base_body<WIDE, APART>(pts, scalars, sc_words, hi, out, scratch, vb_tables + ((size_t)slot * BJJ_K2_BLOCK + threadIdx.x) * VB_TABLE_WORDS, slow,
                      lds, base + threadIdx.x, (size_t)BJJ_K2_BLOCK, xy);
  slot_release_wave();   // every wave of the tile wrote tables into the slot: all acknowledged before the barrier, the push behind it
  __syncthreads();
  if (threadIdx.x == 0) slot_push_one(q, cap_nx, slot);
}
__global__ void __launch_bounds__(BJJ_K2_BLOCK, BJJ_K2_MIN_BLOCKS) bjj_k_mul_var_base_tiles(const uint8_t* __restrict__ pts,
    const uint8_t* __restrict__ scalars, size_t n, uint8_t* __restrict__ out, u32* __restrict__ scratch, u32* __restrict__ vb_tables,
    u32* __restrict__ slow, u32* __restrict__ slotq, u32 cap) {
  var_base_tile<false>(pts, scalars, 8, n, out, scratch, vb_tables, slow, slotq, cap, nullptr);
}
__global__ void __launch_bounds__(BJJ_K2_BLOCK, BJJ_K2_MIN_BLOCKS) bjj_k_mul_var_base_wide_tiles(const uint8_t* __restrict__ pts,
    const uint8_t* __restrict__ scalars, int sc_words, size_t n, uint8_t* __restrict__ out, u32* __restrict__ scratch,
    u32* __restrict__ vb_tables, u32* __restrict__ slow, u32* __restrict__ slotq, u32 cap) {
  var_base_tile<true>(pts, scalars, sc_words, n, out, scratch, vb_tables, slow, slotq, cap, nullptr);
}
// The tiles with the phase-1 stash apart from the output array (zero-copy outputs of the host-pointer pipeline)
__global__ void __launch_bounds__(BJJ_K2_BLOCK, BJJ_K2_MIN_BLOCKS) bjj_k_mul_var_base_tiles_zc(const uint8_t* __restrict__ pts,
    const uint8_t* __restrict__ scalars, size_t n, uint8_t* __restrict__ out, u32* __restrict__ scratch, u32* __restrict__ vb_tables,
    u32* __restrict__ slow, u32* __restrict__ slotq, u32 cap, uint8_t* __restrict__ xy) {
  var_base_tile<false, true>(pts, scalars, 8, n, out, scratch, vb_tables, slow, slotq, cap, xy);
}
__global__ void __launch_bounds__(BJJ_K2_BLOCK, BJJ_K2_MIN_BLOCKS) bjj_k_mul_var_base_wide_tiles_zc(const uint8_t* __restrict__ pts,
    const uint8_t* __restrict__ scalars, int sc_words, size_t n, uint8_t* __restrict__ out, u32* __restrict__ scratch,
    u32* __restrict__ vb_tables, u32* __restrict__ slow, u32* __restrict__ slotq, u32 cap, uint8_t* __restrict__ xy) {
  var_base_tile<true, true>(pts, scalars, sc_words, n, out, scratch, vb_tables, slow, slotq, cap, xy);
}
// The on-curve scan of items first .. end-1 (2 conversions + 5 multiplications per item against K2's ~3 000): the items K6 owns,
// appended to `list` (same layout as `slow`; somebody else has reset it).  Runs on the priority stream while K2 fills the chip:
// 64-lane workgroups of few registers take whatever slot frees up.
__global__ void __launch_bounds__(64) bjj_k_var_base_scan(const uint8_t* __restrict__ pts, size_t first, size_t end, u32* __restrict__ list) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += nthreads) {
    u32 w[8];
    load_w8(pts + i * 64, w);      Fr x = fr_to_mont_words(w);
    load_w8(pts + i * 64 + 32, w); Fr y = fr_to_mont_words(w);
    if (!ref_on_curve(x, y, c_K)) list[8 + atomicAdd(&list[0], 1u)] = (u32)i;
  }
}
// The reference's loop (src/lib.rs:157-162), one item on TWO lanes.  Per bit it does  if bit { r = r + e }  e = e + e  with ONE addition formula
// (PointProjective::add) for both; the two chains only meet in `r + e`.  One lane per item runs both chains -- and, in a wave whose lanes hold
// different scalars, BOTH additions at every bit: 2 x 254 formula evaluations, ~4.9 ms.  Here lane 2k keeps e (doubles it every step), lane 2k+1
// keeps r: in step i both evaluate the SAME formula once -- e_i + e_i on the even lane, r + e_i on the odd one, e_i fetched from the partner lane --
// and keep the sum or not (a select: no divergence).  254 evaluations per item instead of 508, the same field operations in the same order on the
// same values: bit-exact by construction.  All 64 lanes of the wave call this together.
__device__ __forceinline__ Fr fr_from_partner(const Fr& f) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = (u32)__shfl_xor((int)f.v[i], 1, 64);
  return r;
}
__device__ void ref_mul_scalar_pair(bool valid, const Fr& x, const Fr& y, const u32* sc, int nw, bool accumulator, Fr& ox, Fr& oy) {
  int bits = 0;
  if (valid)
    for (int i = nw - 1; i >= 0; i--)
      if (sc[i]) { bits = 32 * i + 32 - __builtin_clz(sc[i]); break; }
  int steps = bits;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(steps, d, 64); steps = o > steps ? o : steps; }
  RefProj acc;                                   // even lane: e = P;  odd lane: r = (0, 1, 1)
  acc.x = fr_select(accumulator, fr_zero(), x); acc.y = fr_select(accumulator, fr_one(), y); acc.z = fr_one();
#pragma unroll 1
  for (int i = 0; i < steps; i++) {
    RefProj e;                                   // the pair's e_i (the even lane's accumulator; on the even lane: its own)
    e.x = fr_select(accumulator, fr_from_partner(acc.x), acc.x);
    e.y = fr_select(accumulator, fr_from_partner(acc.y), acc.y);
    e.z = fr_select(accumulator, fr_from_partner(acc.z), acc.z);
    const RefProj sum = ref_add(acc, e, c_K);    // even: e_i + e_i;  odd: r + e_i
    const bool in_range = i < bits;
    const bool keep = accumulator ? (in_range && ((sc[i >> 5] >> (i & 31)) & 1u)) : in_range;
    acc.x = fr_select(keep, sum.x, acc.x); acc.y = fr_select(keep, sum.y, acc.y); acc.z = fr_select(keep, sum.z, acc.z);
  }
  ox = fr_zero(); oy = fr_zero();
  if (accumulator && !fr_is_zero(acc.z)) {       // src/lib.rs:71-76: z == 0 -> (0, 0)
    const Fr zi = fr_inv(acc.z);
    ox = fr_mul(acc.x, zi); oy = fr_mul(acc.y, zi);
  }
}
#ifdef BJJ_K6_EXPERIMENT
__device__ int bjj_k6_xmode = 0;
// mode 4 / 5: instead of K6 a kernel WITHOUT scratch (and without LDS) that sleeps ~4.9 ms: one wave (4) or 169 waves (5)
__global__ void __launch_bounds__(64) bjj_k_xsleep(const u32* __restrict__ slow) {
  if (slow[0] == 0) return;
  for (int k = 0; k < 1400; k++) __builtin_amdgcn_s_sleep(127);
}
// mode 6: the same with 2 KB of LDS; mode 7: the same with 160 VGPRs allocated (a clobbered high register)
__global__ void __launch_bounds__(64) bjj_k_xsleep_lds(const u32* __restrict__ slow, u32* sink) {
  __shared__ u32 pad[512];
  pad[threadIdx.x] = slow[1];
  if (slow[0] == 0) return;
  for (int k = 0; k < 1400; k++) __builtin_amdgcn_s_sleep(127);
  if (pad[(threadIdx.x + 1) & 63] == 0xdeadbeefu) sink[0] = 1;
}
__global__ void __launch_bounds__(64) bjj_k_xsleep_vgpr(const u32* __restrict__ slow) {
  if (slow[0] == 0) return;
  asm volatile("v_mov_b32 v159, 0" ::: "v159");
  for (int k = 0; k < 1400; k++) __builtin_amdgcn_s_sleep(127);
}
#endif
// K6: exact replay of the reference's loop for the (rare) off-curve inputs, one item per PAIR of lanes (ref_mul_scalar_pair); sc_words words per scalar.
//   patch == nullptr : result j goes to its item's slot out + i * 64
//   patch != nullptr : result j goes to patch + j * 64, next to its index slow[8 + j] (the host-pointer pipeline: a chunk's results
//                      may have left the device before K6 is done; the host lays the few patched results over them, bjj_hip.hip)
//   seen  != nullptr : device-visible HOST word that receives the count -- what the next call's choice of form goes by
__global__ void __launch_bounds__(64) bjj_k_mul_var_base_exact(const uint8_t* __restrict__ pts,
                                                               const uint8_t* __restrict__ scalars, int sc_words,
                                                               uint8_t* __restrict__ out, const u32* __restrict__ slow,
                                                               uint8_t* __restrict__ patch, u32* __restrict__ seen) {
  // A K6 wave is one serial chain of ~3 000 dependent multiplications (254 evaluations of the reference's addition formula; 508 and ~4.9 ms
  // until the chains were split over lane pairs).  Beside K2 it shares its SIMD with two or three K2 waves; it runs at raised wave priority.
  // What K6 beside K2 costs K2 -- 0.7-0.9 ms while any K6 wave is busy -- and what does NOT cause it: profiles/r06_k6_beside_experiments.txt.
#ifdef BJJ_K6_EXPERIMENT   // A/B build only (make EXTRA=-DBJJ_K6_EXPERIMENT): what about K6 beside K2 costs K2 ~0.25 ms per busy K6 wave?
  const int xmode = bjj_k6_xmode;
  if (xmode != 1) __builtin_amdgcn_s_setprio(3);
  if (xmode >= 2) {           // 2: resident and asleep for ~4.8 ms; 3: resident and spinning on the scalar unit -- no arithmetic, no memory, wrong results
    if (blockIdx.x * blockDim.x < slow[0])
      for (int k = 0; k < (xmode == 2 ? 1400 : 350000); k++) { if (xmode == 2) __builtin_amdgcn_s_sleep(127); else asm volatile("s_nop 15"); }
    return;
  }
#else
  __builtin_amdgcn_s_setprio(3);
#endif
  const u32 cnt = slow[0];
  if (seen && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(seen, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int lane = threadIdx.x & 63;
  const bool accumulator = (lane & 1) != 0;
#pragma unroll 1
  for (u32 base = blockIdx.x * 32u; base < cnt; base += gridDim.x * 32u) {   // wave-uniform trip count: the lanes of a pair exchange registers
    const u32 j = base + (u32)(lane >> 1);
    const bool valid = j < cnt;
    const size_t i = slow[8 + (valid ? j : base)];
    u32 w[8];
    load_w8(pts + i * 64, w);      Fr x = fr_to_mont_words(w);
    load_w8(pts + i * 64 + 32, w); Fr y = fr_to_mont_words(w);
    Fr ox, oy;
    ref_mul_scalar_pair(valid, x, y, (const u32*)(scalars + i * (size_t)sc_words * 4), sc_words, accumulator, ox, oy);
    if (valid && accumulator) {
      uint8_t* dst = patch ? patch + (size_t)j * 64 : out + i * 64;
      fr_from_mont_words(ox, w); store_w8(dst, w);
      fr_from_mont_words(oy, w); store_w8(dst + 32, w);
    }
  }
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
// The raw PointProjective::add (src/lib.rs:88-131): (x, y, z) records of 96 bytes in and out, any z, no normalisation.
// The result is the canonical value of each of the reference's three output field elements.
__device__ __forceinline__ RefProj load_proj(const uint8_t* p) {
  u32 w[8];
  RefProj a;
  load_w8(p, w);      a.x = fr_to_mont_words(w);
  load_w8(p + 32, w); a.y = fr_to_mont_words(w);
  load_w8(p + 64, w); a.z = fr_to_mont_words(w);
  return a;
}
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_proj_add(const uint8_t* __restrict__ p, const uint8_t* __restrict__ q, size_t n,
                                                            uint8_t* __restrict__ out) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    const RefProj r = ref_add(load_proj(p + i * 96), load_proj(q + i * 96), c_K);
    u32 w[8];
    fr_from_mont_words(r.x, w); store_w8(out + i * 96, w);
    fr_from_mont_words(r.y, w); store_w8(out + i * 96 + 32, w);
    fr_from_mont_words(r.z, w); store_w8(out + i * 96 + 64, w);
  }
}
// PointProjective::affine (src/lib.rs:70-85): z == 0 -> (0, 0), else (x / z, y / z).
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_proj_affine(const uint8_t* __restrict__ p, size_t n, uint8_t* __restrict__ out) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    const RefProj a = load_proj(p + i * 96);
    Fr ox = fr_zero(), oy = fr_zero();
    if (!fr_is_zero(a.z)) { Fr zi = fr_inv(a.z); ox = fr_mul(a.x, zi); oy = fr_mul(a.y, zi); }
    u32 w[8];
    fr_from_mont_words(ox, w); store_w8(out + i * 64, w);
    fr_from_mont_words(oy, w); store_w8(out + i * 64 + 32, w);
  }
}

namespace bjjk {
int var_base_block() { return BJJ_K2_BLOCK; }
int var_base_lanes_per_cu() {   // resident lanes of K2 per CU (sizes the per-lane table scratch and the grid): the least of all forms
  int a = occupancy_of(bjj_k_mul_var_base_tiles, BJJ_K2_BLOCK), b = occupancy_of(bjj_k_mul_var_base_wide_tiles, BJJ_K2_BLOCK);
  const int az = occupancy_of(bjj_k_mul_var_base_tiles_zc, BJJ_K2_BLOCK), bz = occupancy_of(bjj_k_mul_var_base_wide_tiles_zc, BJJ_K2_BLOCK);
  a = a < az ? a : az; b = b < bz ? b : bz;
  const int a0 = occupancy_of(bjj_k_mul_var_base, BJJ_K2_BLOCK), b0 = occupancy_of(bjj_k_mul_var_base_wide, BJJ_K2_BLOCK);
  a = a < a0 ? a : a0; b = b < b0 ? b : b0;
  return (a < b ? a : b) * BJJ_K2_BLOCK;   // one grid size (and one per-lane table allocation) serves both kernels
}
int occ_point_add() {
  const int a = occupancy_of(bjj_k_point_add, BJJ_BLOCK), b = occupancy_of(bjj_k_proj_add, BJJ_BLOCK),
            c = occupancy_of(bjj_k_proj_affine, BJJ_BLOCK);
  return a < b ? (a < c ? a : c) : (b < c ? b : c);
}
// K2 alone.  slow != nullptr: the list is reset here and filled by the kernel (K6 must follow on `st`: mul_var_base_exact);
// slow == nullptr: the off-curve items are somebody else's (var_base_scan + mul_var_base_exact on another stream).
hipError_t mul_var_base_main(hipStream_t st, int cus, int lanes_per_cu, int variant, const uint8_t* pts, const uint8_t* scalars, int sc_words, size_t n,
                             uint8_t* out, u32* scratch, u32* vb_tables, u32* slow, u32* slotq, u32 slot_cap, uint8_t* xy) {
  if (slow) {
    hipError_t e = hipMemsetAsync(slow, 0, 8 * sizeof(u32), st);
    if (e != hipSuccess) return e;
  }
  const size_t want = (n + BJJ_K2_BLOCK - 1) / BJJ_K2_BLOCK, cap = (size_t)cus * (size_t)(lanes_per_cu / BJJ_K2_BLOCK);
  if (xy && variant != 1) return hipErrorInvalidValue;   // the stash apart exists for the tiles only (what the pipeline launches)
  if (variant == 1 && xy) {
  if (sc_words == 8)
    BJJ_LAUNCH(bjj_k_mul_var_base_tiles_zc, dim3((unsigned)(want ? want : 1)), dim3(BJJ_K2_BLOCK), 0, st, pts, scalars, n, out, scratch, vb_tables, slow,
                       slotq, slot_cap, xy);
  else
    BJJ_LAUNCH(bjj_k_mul_var_base_wide_tiles_zc, dim3((unsigned)(want ? want : 1)), dim3(BJJ_K2_BLOCK), 0, st, pts, scalars, sc_words, n, out, scratch,
                       vb_tables, slow, slotq, slot_cap, xy);
  } else if (variant == 1) {
  if (sc_words == 8)
    BJJ_LAUNCH(bjj_k_mul_var_base_tiles, dim3((unsigned)(want ? want : 1)), dim3(BJJ_K2_BLOCK), 0, st, pts, scalars, n, out, scratch, vb_tables, slow,
                       slotq, slot_cap);
  else
    BJJ_LAUNCH(bjj_k_mul_var_base_wide_tiles, dim3((unsigned)(want ? want : 1)), dim3(BJJ_K2_BLOCK), 0, st, pts, scalars, sc_words, n, out, scratch,
                       vb_tables, slow, slotq, slot_cap);
  } else {
  const int grid = (int)(want < cap ? (want ? want : 1) : cap);
  if (sc_words == 8)
    BJJ_LAUNCH(bjj_k_mul_var_base, dim3(grid), dim3(BJJ_K2_BLOCK), 0, st, pts, scalars, n, out, scratch, vb_tables, slow);
  else
    BJJ_LAUNCH(bjj_k_mul_var_base_wide, dim3(grid), dim3(BJJ_K2_BLOCK), 0, st, pts, scalars, sc_words, n, out, scratch,
                       vb_tables, slow);
  }
  return hipGetLastError();
}
hipError_t var_base_list_reset(hipStream_t st, u32* list) { return hipMemsetAsync(list, 0, 8 * sizeof(u32), st); }
hipError_t var_base_scan(hipStream_t st, int grid, const uint8_t* pts, size_t first, size_t end, u32* list) {
  BJJ_LAUNCH(bjj_k_var_base_scan, dim3(grid), dim3(64), 0, st, pts, first, end, list);
  return hipGetLastError();
}
hipError_t mul_var_base_exact(hipStream_t st, int grid_exact, const uint8_t* pts, const uint8_t* scalars, int sc_words, uint8_t* out, const u32* slow,
                              uint8_t* patch, u32* seen) {
#ifdef BJJ_K6_EXPERIMENT
  { const char* e = getenv("BJJ_K6_XMODE"); const int v = e ? atoi(e) : 0;
    if (v == 4 || v == 5) { BJJ_LAUNCH(bjj_k_xsleep, dim3(v == 4 ? 1 : 169), dim3(64), 0, st, slow); return hipGetLastError(); }
    if (v == 6) { BJJ_LAUNCH(bjj_k_xsleep_lds, dim3(4), dim3(64), 0, st, slow, (u32*)slow + 4); return hipGetLastError(); }
    if (v == 7) { BJJ_LAUNCH(bjj_k_xsleep_vgpr, dim3(4), dim3(64), 0, st, slow); return hipGetLastError(); } }
  { static const int m = [] { const char* e = getenv("BJJ_K6_XMODE"); const int v = e ? atoi(e) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(bjj_k6_xmode), &v, sizeof(int)); return v; }(); (void)m; }
#endif
  BJJ_LAUNCH(bjj_k_mul_var_base_exact, dim3(grid_exact), dim3(64), 0, st, pts, scalars, sc_words, out, slow, patch, seen);
  return hipGetLastError();
}
int occ_var_base_scan() { return occupancy_of(bjj_k_var_base_scan, 64); }   // resident scan waves per CU
hipError_t point_add(hipStream_t st, int grid, const uint8_t* p, const uint8_t* q, size_t n, uint8_t* out) {
  BJJ_LAUNCH(bjj_k_point_add, dim3(grid), dim3(BJJ_BLOCK), 0, st, p, q, n, out);
  return hipGetLastError();
}
hipError_t proj_add(hipStream_t st, int grid, const uint8_t* p, const uint8_t* q, size_t n, uint8_t* out) {
  BJJ_LAUNCH(bjj_k_proj_add, dim3(grid), dim3(BJJ_BLOCK), 0, st, p, q, n, out);
  return hipGetLastError();
}
hipError_t proj_affine(hipStream_t st, int grid, const uint8_t* p, size_t n, uint8_t* out) {
  BJJ_LAUNCH(bjj_k_proj_affine, dim3(grid), dim3(BJJ_BLOCK), 0, st, p, n, out);
  return hipGetLastError();
}
}  // namespace bjjk
