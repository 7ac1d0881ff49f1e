// libbjj_hip.so, kernel unit 4: K4, verify(pk, sig, msg) (src/lib.rs:395-412) and verify_schnorr (:375-385).
// per-lane table entries of this unit: raw, 144 B (bjj_device.hpp "per-lane variable-base table")
#ifndef BJJ_VERIFY_PNIELS_LAYOUT
#define BJJ_VERIFY_PNIELS_LAYOUT 0
#endif
#define BJJ_PNIELS_LAYOUT BJJ_VERIFY_PNIELS_LAYOUT
#include "k_common.hpp"

// Waves per SIMD the verify kernels are compiled for (the second __launch_bounds__ argument of HIP is the minimum number of
// waves per execution unit; A/B knob): 2 = 256 VGPRs; 3 (168 VGPRs, more scratch) was measured slower
// (profiles/r02_ab_occupancy.txt).
#ifndef BJJ_VERIFY_MIN_BLOCKS
#define BJJ_VERIFY_MIN_BLOCKS 2
#endif

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
// How the main kernel's work is handed out -- both forms are compiled, the host picks per call (bjj_hip.hip: enqueue_verify;
// BJJ_VERIFY_DISPATCH=0|1 in the environment forces one):
//   0  persistent waves: the grid is one resident set, waves pull 64-item groups through atomic cursors (rounds 1-2).  Its
//      waves stay in lock-step over the 70 K-instruction body, which is worth ~4 % in a launch of many rounds: the form for
//      one launch of more than 2^21 items that runs alone (profiles/r03_ab_verify_group_dispatch.txt: 59.5 vs 57.1 M/s at 2^22);
//   1  one group per 64-lane workgroup: the grid is (exact-list groups) + (batch chunks), the hardware's workgroup dispatcher
//      does the scheduling, and a per-XCD queue hands every running workgroup a slot of the per-lane table scratch: better at
//      2^20 items (the last, partly empty round is cheaper) and the form that makes overlapping launches work-conserving.
// With persistent waves two launches that overlap (two streams, two scratch sets) split the chip half and half for their
// whole life and each pays its own partly empty last round (profiles/r03_ab_inkernel_scan_rejected.txt); workgroups that
// retire after one group give every freed slot to whichever launch has work pending, which is what makes the pair
// work-conserving.  The exact-list groups have the lowest block indices, so they are still dispatched first.
// 64-lane workgroups: one light wave (73 VGPRs) fits into any slot a retiring wave of the main kernel frees.
#ifndef BJJ_SCAN_BLOCK
#define BJJ_SCAN_BLOCK 64
#endif
__global__ void __launch_bounds__(BJJ_SCAN_BLOCK) bjj_k_eddsa_verify_scan(const uint8_t* __restrict__ pk,
                                                                     const uint8_t* __restrict__ rb8,
                                                                     const uint8_t* __restrict__ msg, size_t first, size_t n,
                                                                     u32* __restrict__ wl) {
  // items first .. n-1 of the arrays (first > 0: the host-pointer pipeline scans a batch chunk by chunk, as the chunks arrive,
  // into ONE list of batch-wide indices)
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    VerifyIn in = {pk + i * 64, rb8 + i * 64, nullptr, msg + i * 32};
    if (verify_needs_exact(in, c_K)) wl[WL_HDR + atomicAdd(&wl[0], 1u)] = (u32)i;
  }
}
// ---- dispatch mode 0: persistent waves + atomic cursors (rounds 1-2; still the better form for one very large launch)
// NOTE on this cursor idiom (`if (lane == 0) c = atomicAdd(..); c = __shfl(c, 0);`): it is safe in the two loops below,
// whose bodies contain no other `if (lane == 0)` block.  In a loop whose body ENDS with such a block hipcc threads lanes
// 1..63 from the end of the body straight into the next iteration's cross-lane read while lane 0 is still away -- they read
// an inactive lane, get 0 and never leave (round 3, profiles/r03_ab_inkernel_scan_rejected.txt).  The branch-free form --
// every lane issues the atomic, lane 0 adds the step and the others 0, v_readfirstlane for the result -- is immune.
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
  __shared__ __attribute__((aligned(16))) u32 stage[(BJJ_VERIFY_BLOCK / 64) * FB_STAGE_WORDS];
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
__global__ void __launch_bounds__(BJJ_VERIFY_BLOCK, BJJ_VERIFY_MIN_BLOCKS) bjj_k_schnorr_verify(const u32* __restrict__ table, int W, int nwin,
                                                                     const uint8_t* __restrict__ pk,
                                                                     const uint8_t* __restrict__ rb8,
                                                                     const uint8_t* __restrict__ s,
                                                                     const uint8_t* __restrict__ msg, size_t n,
                                                                     uint8_t* __restrict__ ok, u32* __restrict__ vb_tables,
                                                                     u32* __restrict__ wl) {
  verify_kernel_body<true>(table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl);
}
__global__ void __launch_bounds__(BJJ_VERIFY_BLOCK, BJJ_VERIFY_MIN_BLOCKS) bjj_k_eddsa_verify(const u32* __restrict__ table, int W, int nwin,
                                                                   const uint8_t* __restrict__ pk,
                                                                   const uint8_t* __restrict__ rb8,
                                                                   const uint8_t* __restrict__ s,
                                                                   const uint8_t* __restrict__ msg, size_t n,
                                                                   uint8_t* __restrict__ ok, u32* __restrict__ vb_tables,
                                                                   u32* __restrict__ wl) {
  verify_kernel_body<false>(table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl);
}

__global__ void bjj_k_probe_xcc(u32* out) {
  if (threadIdx.x == 0) atomicMax(out, xcc_id() + 1u);
}
// ---- dispatch mode 1: one 64-item group per workgroup --------------------------------------------------------------
template <bool SCHNORR>
__device__ __forceinline__ void verify_group_body(const u32* __restrict__ table, int W, int nwin,
                                                  const uint8_t* __restrict__ pk, const uint8_t* __restrict__ rb8,
                                                  const uint8_t* __restrict__ s, const uint8_t* __restrict__ msg, size_t n,
                                                  uint8_t* __restrict__ ok, u32* __restrict__ vb_tables,
                                                  const u32* __restrict__ wl, u32* __restrict__ slotq, u32 cap_nx, u32 exact_wgs) {
  __shared__ __attribute__((aligned(16))) u32 stage[FB_STAGE_WORDS];
  const int lane = threadIdx.x & 63;
  // grid = exact_wgs workgroups for the exact list (lowest indices: dispatched first) + the rest for the batch; both kinds
  // stride over their groups, so that the grid stays small for huge batches (one group each up to 2^21 items)
  const size_t nchunks = (n + 63) / 64, b = blockIdx.x;
  const bool exact = b < exact_wgs;
  const unsigned long long nexact = wl[0];
  if (exact && b * 64 >= nexact) return;                     // wave-uniform: nothing on the list for this block
  u32* q = slot_queue_of_this_xcd(slotq, cap_nx);
  const u32 slot = slot_pop(q, cap_nx, lane);
  u32* tbl = vb_tables + ((size_t)slot * 64 + lane) * VB_VERIFY_WORDS;
#ifdef BJJ_EXP_NO_EXACT_IN_BULK   // experiment (resource usage only: the exact path out of the bulk kernel's allocation)
  if (false) {
#else
  if (exact) {
#endif
#pragma unroll 1
    for (size_t c = b * 64; c < nexact; c += (size_t)exact_wgs * 64) {
      if (c + lane < nexact) {
        const size_t i = wl[WL_HDR + c + lane];
        VerifyIn in = {pk + i * 64, rb8 + i * 64, s + i * 32, msg + i * 32};
        ok[i] = (uint8_t)verify_exact_t<SCHNORR>(in, table, W, nwin, tbl, c_K);
      }
    }
  } else {
    const GatherCoopLds<1> fb = {table, stage, lane};
    const size_t bulk_wgs = gridDim.x - exact_wgs;
#pragma unroll 1
    for (size_t ch = b - exact_wgs; ch < nchunks; ch += bulk_wgs) {
      const size_t i = ch * 64 + lane, ic = i < n ? i : n - 1;   // every lane runs (cooperative gathers)
      VerifyIn in = {pk + ic * 64, rb8 + ic * 64, s + ic * 32, msg + ic * 32};
      bool need_exact;
      const int v = verify_fast_t<SCHNORR>(in, fb, W, nwin, tbl, c_K, need_exact);
      if (i < n && !need_exact) ok[i] = (uint8_t)v;
    }
  }
  slot_push(q, cap_nx, slot, lane);
}
__global__ void __launch_bounds__(64, BJJ_VERIFY_MIN_BLOCKS) bjj_k_schnorr_verify_groups(const u32* __restrict__ table, int W, int nwin,
    const uint8_t* __restrict__ pk, const uint8_t* __restrict__ rb8, const uint8_t* __restrict__ s, const uint8_t* __restrict__ msg,
    size_t n, uint8_t* __restrict__ ok, u32* __restrict__ vb_tables, const u32* __restrict__ wl, u32* __restrict__ slotq, u32 cap,
    u32 exact_wgs) {
  verify_group_body<true>(table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl, slotq, cap, exact_wgs);
}
__global__ void __launch_bounds__(64, BJJ_VERIFY_MIN_BLOCKS) bjj_k_eddsa_verify_groups(const u32* __restrict__ table, int W, int nwin,
    const uint8_t* __restrict__ pk, const uint8_t* __restrict__ rb8, const uint8_t* __restrict__ s, const uint8_t* __restrict__ msg,
    size_t n, uint8_t* __restrict__ ok, u32* __restrict__ vb_tables, const u32* __restrict__ wl, u32* __restrict__ slotq, u32 cap,
    u32 exact_wgs) {
  verify_group_body<false>(table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl, slotq, cap, exact_wgs);
}

namespace bjjk {
int probe_xccs(hipStream_t st, u32* d_word) {   // number of XCDs = highest XCC_ID seen by a few thousand workgroups + 1
  if (hipMemsetAsync(d_word, 0, sizeof(u32), st) != hipSuccess) return 0;
  BJJ_LAUNCH(bjj_k_probe_xcc, dim3(4096), dim3(64), 0, st, d_word);
  u32 h = 0;
  if (hipMemcpyAsync(&h, d_word, sizeof(u32), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 0;
  return (int)h;
}
int occ_verify() {   // resident workgroups per CU; one grid size (and one per-lane table allocation) serves both kernels
  int a = occupancy_of(bjj_k_eddsa_verify_groups, 64), b = occupancy_of(bjj_k_schnorr_verify_groups, 64);
  const int a0 = occupancy_of(bjj_k_eddsa_verify, BJJ_VERIFY_BLOCK) * (BJJ_VERIFY_BLOCK / 64), b0 = occupancy_of(bjj_k_schnorr_verify, BJJ_VERIFY_BLOCK) * (BJJ_VERIFY_BLOCK / 64);
  a = a < a0 ? a : a0; b = b < b0 ? b : b0;
  return a < b ? a : b;   // resident WAVES per CU, the least of all four kernels
}
int occ_verify_scan() { return occupancy_of(bjj_k_eddsa_verify_scan, BJJ_SCAN_BLOCK) * BJJ_SCAN_BLOCK / 64; }   // resident scan WAVES per CU
int verify_scan_block() { return BJJ_SCAN_BLOCK; }
// The on-curve scan (memset of the list header + scan kernel) and the main kernel are launched separately so that the host
// side can route the scan through a high-priority stream (bjj_hip.hip: verify_launch).
hipError_t verify_scan(hipStream_t st, int grid_scan, const uint8_t* pk, const uint8_t* rb8, const uint8_t* msg, size_t n, u32* wl) {
  hipError_t e = hipMemsetAsync(wl, 0, WL_HDR * sizeof(u32), st);
  if (e != hipSuccess) return e;
  BJJ_LAUNCH(bjj_k_eddsa_verify_scan, dim3(grid_scan), dim3(BJJ_SCAN_BLOCK), 0, st, pk, rb8, msg, (size_t)0, n, wl);
  return hipGetLastError();
}
// The scan of items first .. end-1 appended to a list that somebody else has reset (verify_list_reset): the host-pointer
// pipeline, one range per chunk.
hipError_t verify_list_reset(hipStream_t st, u32* wl) { return hipMemsetAsync(wl, 0, WL_HDR * sizeof(u32), st); }
hipError_t verify_scan_range(hipStream_t st, int grid_scan, const uint8_t* pk, const uint8_t* rb8, const uint8_t* msg, size_t first, size_t end, u32* wl) {
  BJJ_LAUNCH(bjj_k_eddsa_verify_scan, dim3(grid_scan), dim3(BJJ_SCAN_BLOCK), 0, st, pk, rb8, msg, first, end, wl);
  return hipGetLastError();
}
hipError_t verify_main(hipStream_t st, int mode, int grid, bool schnorr, const u32* table, int W, int nwin, const uint8_t* pk,
                       const uint8_t* rb8, const uint8_t* s, const uint8_t* msg, size_t n, uint8_t* ok, u32* vb_tables, u32* wl,
                       u32* slotq, u32 slot_cap, int part) {
  if (mode == 1) {
  // part (VERIFY_BOTH / VERIFY_BULK / VERIFY_EXACT, bjj_launch.hpp): the two kinds of workgroup of this form are independent --
  // a bulk workgroup skips the items of the list, an exact workgroup touches nothing else -- so they can be two launches.  The
  // host-pointer pipeline uses that: a chunk's bulk launch lasts as long as its items take, not as long as the ~3x longer exact
  // items among them, and ONE exact launch serves the whole batch (bjj_hip.hip: VerifyPipe).
  const size_t nchunks = (n + 63) / 64;
  const unsigned exact_wgs = part == VERIFY_BULK ? 0u : (unsigned)(nchunks < 4096 ? nchunks : 4096);      // most of them find nothing on the list and exit at once
  const unsigned bulk_wgs = part == VERIFY_EXACT ? 0u : (unsigned)(nchunks < 32768 ? nchunks : 32768);    // one 64-item chunk each up to 2^21 items, strided beyond
  const unsigned groups = exact_wgs + bulk_wgs;
  if (schnorr)
    BJJ_LAUNCH(bjj_k_schnorr_verify_groups, dim3(groups), dim3(64), 0, st, table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl, slotq, slot_cap, exact_wgs);
  else
    BJJ_LAUNCH(bjj_k_eddsa_verify_groups, dim3(groups), dim3(64), 0, st, table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl, slotq, slot_cap, exact_wgs);
  } else {
  if (part != VERIFY_BOTH) return hipErrorInvalidValue;   // the persistent form is one launch
  if (schnorr)
    BJJ_LAUNCH(bjj_k_schnorr_verify, dim3(grid), dim3(BJJ_VERIFY_BLOCK), 0, st, table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl);
  else
    BJJ_LAUNCH(bjj_k_eddsa_verify, dim3(grid), dim3(BJJ_VERIFY_BLOCK), 0, st, table, W, nwin, pk, rb8, s, msg, n, ok, vb_tables, wl);
  }
  return hipGetLastError();
}
}  // namespace bjjk
