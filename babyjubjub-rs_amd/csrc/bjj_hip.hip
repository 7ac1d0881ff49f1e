// libbjj_hip.so -- host side: contexts and the extern "C" boundary declared in include/bjj_hip.h.
// There is no CPU fallback anywhere in this library: every entry point launches device code or
// returns an error.
//
// Kernel map (SURVEY.md section 2 "kernel inventory"; kernels in k_*.hip, bodies in bjj_device.hpp / sign.hpp):
//   k_fixed.hip       bjj_k_build_fixed_table   init-time: window table of B8 multiples (affine Niels form)
//                     bjj_k_mul_fixed_base      K1  B8.mul_scalar(n)               src/lib.rs:149-164, 37-46
//   k_var.hip         bjj_k_mul_var_base[_wide] K2  P.mul_scalar(n), on-curve P    src/lib.rs:149-164
//                     bjj_k_mul_var_base_exact  K6  the reference's exact op sequence for off-curve P
//                     bjj_k_point_add / bjj_k_proj_add / bjj_k_proj_affine         src/lib.rs:88-131, 70-85
//   k_hash_codec.hip  bjj_k_poseidon5           K3  POSEIDON.hash([a,b,c,d,e])     src/lib.rs:400-404
//                     bjj_k_compress_points / bjj_k_decompress_points / bjj_k_merge_codec_flags   src/lib.rs:166-224, 260-268
//                     bjj_k_scalar_keys                                            src/lib.rs:284-302
//   k_verify.hip      bjj_k_eddsa_verify_scan / bjj_k_eddsa_verify / bjj_k_schnorr_verify
//                                               K4  verify / verify_schnorr        src/lib.rs:395-412, 375-385
//   k_sign.hip        bjj_k_sign / bjj_k_sign_schnorr                              src/lib.rs:308-361
//   k_small.hip       bjj_k_mul_var_base_quad / bjj_k_poseidon5_coop / bjj_k_eddsa_verify_small
//                                               K2 / K3 / K4 for SHORT calls: four lanes per item, six per hash, eight per signature
// K5 (batched affine conversion) is the epilogue of K1/K2 (k_common.hpp: block_invert).
// Multi-GPU: bjj_multi_* (bjj_multi.inc, included at the end of this file) -- one context per device,
// contiguous ceil(n/G) blocks; the device-resident form scatters inputs / gathers results with RCCL (bound with dlopen on
// first use) or with peer copies.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bjj_hip.h"
#include "bjj_device.hpp"
#include "bjj_launch.hpp"
#include "copy_pool.hpp"

using namespace bjj;

#ifdef BJJ_TEST_HOOKS   // `make hooks`: the same library plus the failure injection of bjj_multi.inc; never the shipped one
#define BJJ_VERSION_STRING "bjj-hip 0.6.0 gfx950 +test-hooks"
#else
#define BJJ_VERSION_STRING "bjj-hip 0.6.0 gfx950"
#endif
// Fixed-base window width.  window_bits = 0 (default) is a modest 23 bits = 11 signed digits, 5.9 GB: a library that
// is linked into a process with other tenants of the GPU must not take half of the HBM unasked.  The wide tables are
// opt-in: an explicit width (28 bits = 9 digits, 9 x (2^27 + 1) entries = 154.6 GB of the 288 GB; 26 = 10 digits,
// 42.9 GB) or BJJ_WINDOW_AUTO (-1) = the widest of these whose table fits in 60 % of the device's free memory.  One
// addition less per step: the kernel is VALU-bound and the cooperative gathers keep the table reads off the critical
// path at any size (profiles/r01i_ablation_window_bits_coop_gather.txt).
static const int kAutoWindowBits[] = {28, 26, 23, 21, 16};
#define BJJ_DEFAULT_WINDOW_BITS 23
#define BJJ_MAX_WINDOW_BITS 28

// ===========================================================================
// context
// ===========================================================================
static thread_local std::string g_err;
static int set_err(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCK(call)                                                                                  \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return set_err(BJJ_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                  \
  } while (0)

// Every entry point runs on its context's device and leaves the calling thread's current HIP device as it found it (a host
// that shares the thread with torch or another HIP user must not find its default device changed).
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev == dev) prev = -1; else err = hipSetDevice(dev);
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ENTER_DEVICE(dev)                                                                                       \
  DeviceGuard dg_(dev);                                                                                         \
  if (dg_.err != hipSuccess) return set_err(BJJ_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(dg_.err))

#define BJJ_PIPE_BUFS 4
#define BJJ_SCRATCH_SETS 2
#define BJJ_STREAM_MARKS 8
struct ScratchSet {
  u32* scratch = nullptr;      // n * 64 B (Z, prefix)
  size_t scratch_items = 0;
  u32* vb_tables = nullptr;    // grid threads * VB_TABLE_WORDS * 4 B
  size_t vb_threads = 0;
  u32* slow = nullptr;         // [0] = count, [1..] item indices deferred to the exact-path kernels
  size_t slow_items = 0;
  // verify: the on-curve scan of a call is routed through this stream of the HIGHEST priority, so that while another launch
  // of the context fills the chip its few light waves get the next slots that free up instead of queueing behind that
  // launch's pending workgroups for a whole launch (profiles/r03_ab_verify_group_dispatch.txt)
  hipStream_t scan_stream = nullptr;
  hipEvent_t ev_scan_in = nullptr, ev_scan_out = nullptr;
  u32* slotq2 = nullptr;       // variable base, one tile per workgroup: the same per-XCD rings, one slot = one workgroup's tables
  u32 slot_cap2 = 0;
  u32* slotq = nullptr;        // verify, one group per workgroup: per-XCD ring of free per-lane-table slots (k_verify.hip)
  u32 slot_cap = 0;            // slots per XCD
  uint8_t* codec = nullptr;    // verify_compressed: n * (64 pk + 64 R + 32 s + 2 flags) bytes; public_keys: the scalar keys
  size_t codec_items = 0;
  uint8_t* xy = nullptr;       // K1 with compressed output: n * 64 B, the phase-1 stash of X, Y (the 32-byte output slot cannot hold it)
  size_t xy_items = 0;
  // Ordering of the set: every call that uses it records `ev_last` on its stream after enqueueing, and a call on a
  // DIFFERENT stream first makes its stream wait for it.  Calls return before the work runs, so "serialised by the
  // caller" alone would not order execution.
  hipEvent_t ev_last = nullptr;
  hipStream_t last_stream = nullptr;
  bool have_last = false;
  uint64_t last_use = 0;
  uint64_t last_set_use = 0;   // value of bjj_ctx::set_uses when this set was taken last
};
struct StreamMark {
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;
  bool used = false;
  uint64_t last_use = 0;
};
struct bjj_ctx {
  int device = 0;
  int cus = 0;
  int W = 16, nwin = 16;
  double init_ms = 0.0;
  // residency of each kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor): grids are sized to exactly one resident set
  // of workgroups, items are grid-strided.  K1 / K2: resident LANES per CU (their workgroup size is the kernel unit's
  // business); the others: resident 256-lane workgroups per CU
  int lanes_fixed = 512, lanes_var = 512;
  int lanes_fixed_2x256 = 512;   // resident lanes per CU of K1's two-workgroup shape
  int verify_mode = -1;          // -1 = per call (persistent waves for one launch > 2^21 items that runs alone, groups otherwise), 0 / 1 = forced (BJJ_VERIFY_DISPATCH)
  int k2_variant = -1;           // -1 = per call (tiles for a launch that runs alone, grid-strided while another is in flight), 0 / 1 = forced (BJJ_K2_VARIANT)
  int k1_variant = -1;           // -1 = per call (two-workgroup shape while another launch of the context is in flight), 0 / 1 = forced (BJJ_K1_VARIANT)
  int occ_poseidon = 1, occ_verify = 1, occ_scan = 1, occ_add = 1;
  // Variable base, off-curve points (k_var.hip): K6 runs BEHIND K2 on the caller's stream (nothing extra for a clean batch) or,
  // with the list made by a scan, BESIDE it on the set's priority stream.  -1 = per call by what the previous calls met
  // (vb_seen: one device-visible pinned word per scratch set, written by K6 with the number of items it had), 0 / 1 = forced
  // (BJJ_VB_SPLIT)
  int vb_split = -1;
  int occ_vb_scan = 1;
  u32* vb_seen = nullptr;                  // [BJJ_SCRATCH_SETS] per set (device-pointer calls) + [1] the host-pointer pipeline
  int last_vb_split = -1;
  uint8_t* patch_host = nullptr;           // host-pointer pipeline: K6's results on their way into the caller's array (pinned)
  size_t patch_host_bytes = 0;
  int xccs = 1;                // XCDs of the device (probed at init; sizes the verify kernels' slot queues)
  int occ_decomp = 1, occ_sign = 1, occ_sign_schnorr = 1;
  // signer hardening (bjj_set_signer_constant_time): a second, small fixed-base table (4-bit windows: 63 x 9 entries) that
  // the signer kernels SCAN instead of indexing; built on first use
  bool ct_signer = false;
  u32* ct_table = nullptr;
  u32* ct_bases = nullptr;
  int occ_sign_ct = 1, occ_sign_schnorr_ct = 1;
  hipStream_t stream = nullptr;
  u32* table = nullptr;      // [window][digit 0 .. 2^(W-1)] x 128 B
  u32* bases = nullptr;      // P_j = 2^(W j) * B8, one Niels entry per window
  size_t table_bytes = 0;
  // Scratch is kept in BJJ_SCRATCH_SETS independent sets so that calls on two streams can be in flight at once
  // (pick_set): a launch whose last wave-round is only partly filled -- 2^20 verifications are 8.1 rounds of the 2 048
  // resident waves, 2^20 variable-base multiplications 5.3 rounds of the 196 608 resident lanes -- then shares the chip
  // with the head of the next launch instead of idling it.  A single-stream caller only ever touches (and allocates) set 0.
  ScratchSet set[BJJ_SCRATCH_SETS];
  uint64_t use_counter = 0;
  uint64_t set_uses = 0;         // calls that took a scratch set (streams_alternate)
  // calls that use no scratch (Poseidon, point add, codec, sign) are not ordered behind anything; their completion
  // events are only kept so that bjj_sync can wait for them: one slot per distinct caller stream
  StreamMark marks[BJJ_STREAM_MARKS];
  // host-pointer API (run_pipelined): chunked pipeline
  //   caller's array -[H2D, s_in]-> dstage -[kernels on the lanes stream / stream2, chunks alternating]-> dstage -[D2H, s_out]-> caller's array
  // straight from / to the caller's memory when that is pinned (bjj_host_alloc / bjj_host_register / any hipHostMalloc'd or
  // hipHostRegister'ed range); a PAGEABLE array goes through the rings pin_in[] / pin_out[], moved by the context's copy workers
  // (CopyPool), never by the enqueueing thread
  hipStream_t s_in = nullptr, s_out = nullptr, stream2 = nullptr;
  std::vector<hipEvent_t> ev_in, ev_k, ev_out;   // per chunk of a super-batch (grown on demand)
  std::vector<hipEvent_t> ev_dec;          // wire-format verifier through the pipeline: behind a chunk's decompressions (VerifyCompressedPipe)
  hipEvent_t ev_tail = nullptr;            // behind the extra stages of a call (PipeExtra: verify's batch-wide exact launch)
  u32* pipe_wl = nullptr;                  // verify through the pipeline: ONE list of off-curve items per super-batch (VerifyPipe)
  size_t pipe_wl_items = 0;
  uint8_t* dstage = nullptr;   // device staging for one super-batch: every array contiguous
  size_t pipe_bytes = 0;
  uint8_t* pin_in[BJJ_PIPE_BUFS] = {};     // pinned rings of the staged path (allocated on the first call that has a pageable array)
  uint8_t* pin_out[BJJ_PIPE_BUFS] = {};
  size_t pin_in_bytes = 0, pin_out_bytes = 0;
  size_t pipe_budget = 0;                  // bytes of device staging a call may take (BJJ_PIPE_STAGING_MB)
  CopyPool* pool = nullptr;
  size_t pipe_chunk = 0, pipe_first = 0;   // chunk schedule (items): first chunk, doubling up to pipe_chunk
  bool pipe_env_schedule = false;          // ... given in the environment: it overrides the entry points' own schedules too
  bool pipe_ready = false;                 // ensure_pipe's first-use block ran to its end
  bool pipe_small_direct = true;           // BJJ_PIPE_SMALL_DIRECT=0: short calls on pinned memory keep their copies (A/B, tests)
  bool pipe_zero_copy_in = true;           // BJJ_PIPE_ZERO_COPY_IN=0: inputs always arrive through the copy engines (A/B, tests)
  bool k1_half_now = false;                // a host-pointer call is enqueueing chunk launches of K1 that take ONE workgroup slot per CU each
  bool pipe_zero_copy = true;              // BJJ_PIPE_ZERO_COPY=0: results always leave through the copy engines (A/B, tests)
  u32 last_host_zero_copy = 0;
  bool in_pipeline = false;                // a host-pointer call is enqueueing (enqueue_verify: where the scans run)
  bool pipe_scan_inline = true;            // BJJ_PIPE_SCAN=prio (developer): scans of the pipeline's chunks on the priority streams
  bool force_staged = false;               // BJJ_HOST_FORCE_STAGED=1: treat every host array as pageable (A/B, tests)
  // what the last calls did (bjj_get_info; tests and the bench line read these)
  int last_k1 = -1, last_k2 = -1, last_verify_mode = -1;
  size_t fb_quad_max = (size_t)1 << 15;        // fixed base / public keys: calls of at most this many items run four lanes per item (BJJ_FB_QUAD_MAX; 0 = never)
  size_t sign_small_max = (size_t)1 << 13;     // sign: calls of at most this many signatures run eight lanes per signature (BJJ_SIGN_SMALL_MAX; 0 = never)
  int last_sign = -1;
  size_t verify_small_max = (size_t)1 << 13;   // EdDSA verify: calls of at most this many signatures run eight lanes per signature (BJJ_VERIFY_SMALL_MAX; 0 = never)
  size_t p5_coop_max = (size_t)1 << 14;    // Poseidon: calls of at most this many hashes run six lanes per hash (BJJ_P5_COOP_MAX; 0 = never)
  int last_p5 = -1;
  size_t vb_quad_max = (size_t)1 << 14;    // variable base: calls of at most this many items run four lanes per item (BJJ_VB_QUAD_MAX; 0 = never)
  int idle_alternations = 0;               // expect_overlap: consecutive alternating calls that found the other set idle
  u32 last_host_direct = 0, last_host_staged = 0, last_host_chunks = 0;
  u32* slot_block = nullptr;               // all slot-queue rings of the context in one device allocation (slot_block_make)
  size_t slot_block_words = 0;
  u32* err_words = nullptr;                // pinned: the ring block as read back by ctx_check_slot_queues
  bool rings_used = false;                 // a kernel that pops / pushes slots has been launched since the last check
};

static int grid_for(const bjj_ctx* c, size_t n, int blocks_per_cu, int block = BJJ_BLOCK) {
  size_t want = (n + block - 1) / block;
  size_t cap = (size_t)c->cus * blocks_per_cu;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

// Which scratch set a call on stream `st` uses:
//  (1) the set this stream used last -- calls of one stream are ordered by the stream itself, nothing to wait for;
//  (2) else a set that was never used;  (3) else the least recently used one (the caller then waits for its last call).
// Two streams that alternate therefore settle on one set each and overlap; one stream never leaves set 0.
static ScratchSet* pick_set(bjj_ctx* c, hipStream_t st) {
  for (ScratchSet& S : c->set) if (S.have_last && S.last_stream == st) return &S;
  for (ScratchSet& S : c->set) if (!S.have_last) return &S;
  ScratchSet* lru = &c->set[0];
  for (ScratchSet& S : c->set) if (S.last_use < lru->last_use) lru = &S;
  return lru;
}
// Makes `st` safe to touch set S: waits (on the device) for the set's last call if that ran on another stream.
static int set_enter(bjj_ctx* c, ScratchSet* S, hipStream_t st) {
  if (!S->ev_last) HIPCK(hipEventCreateWithFlags(&S->ev_last, hipEventDisableTiming));
  if (S->have_last && S->last_stream != st) HIPCK(hipStreamWaitEvent(st, S->ev_last, 0));
  return BJJ_OK;
}
static int set_leave(bjj_ctx* c, ScratchSet* S, hipStream_t st) {
  HIPCK(hipEventRecord(S->ev_last, st));
  S->last_stream = st;
  S->have_last = true;
  S->last_use = ++c->use_counter;
  S->last_set_use = ++c->set_uses;
  return BJJ_OK;
}
// Is a launch of this context that uses ANOTHER scratch set still queued or running?  (event query: no synchronisation)
// hipEventQuery leaves hipErrorNotReady as the thread's last error; it is cleared on EVERY path so that a launcher that reads
// hipGetLastError() after its launch never reports it.
static bool other_launch_in_flight(bjj_ctx* c, const ScratchSet* mine) {
  bool busy = false;
  for (const ScratchSet& S : c->set)
    if (&S != mine && S.have_last && hipEventQuery(S.ev_last) == hipErrorNotReady) busy = true;
  (void)hipGetLastError();   // hipErrorNotReady is not an error
  return busy;
}
// Does the caller ALTERNATE over streams (one scratch set each)?  True when another set took one of the last
// BJJ_SCRATCH_SETS calls that used scratch.  Unlike the event query above this does not depend on how far the device has got
// when the host enqueues: a caller that ping-pongs over two streams gets the same kernel shape for every launch of the run,
// a one-stream caller never sees it (profiles/r04_driver_protocol.txt).
static bool streams_alternate(const bjj_ctx* c, const ScratchSet* mine) {
  for (const ScratchSet& S : c->set)
    if (&S != mine && S.have_last && c->set_uses - S.last_set_use < BJJ_SCRATCH_SETS) return true;
  return false;
}
// A launch of more than 2^21 items gains nothing from sharing the chip with another launch of the context (its partly empty last
// round is a fraction of a percent) but would lose the kernel form that is best for a launch that runs alone (persistent verify
// waves, K2 tiles: 2-4 % at >= 2^22 items, profiles/r03_throughput_vs_batch.txt).  Such a launch therefore queues BEHIND the other
// scratch sets' last calls (a device-side wait, no host synchronisation) and is then enqueued as a launch that runs alone.
#define BJJ_LARGE_LAUNCH ((size_t)1 << 21)
static int wait_for_other_sets(bjj_ctx* c, const ScratchSet* mine, hipStream_t st) {
  for (const ScratchSet& S : c->set)
    if (&S != mine && S.have_last && S.last_stream != st) HIPCK(hipStreamWaitEvent(st, S.ev_last, 0));
  return BJJ_OK;
}
// Will this launch share the chip with another launch of the context?  The kernels come in a form for a launch that runs alone
// and a form for overlapping launches (K1: one 512-lane workgroup per CU / two of 256 lanes; K2: tiles / grid-strided; verify:
// scan in line / on the priority stream), and the wrong form costs: K1's two-workgroup shape run ALONE takes 0.611-0.623 ms instead
// of 0.602-0.605 ms (+1.5 ... 3 %), K2's grid-strided form alone 14.63 instead of 14.13 ms (+3.5 %; profiles/r05_driver_protocol.txt,
// r05_host_pipeline.txt).  Pattern AND state decide:
//   * another set's launch is still queued or running (event query)          -> overlap
//   * else the caller alternates over streams (streams_alternate) and this is the FIRST such call that finds the other set
//     idle -- the first launch behind a synchronisation point of a caller that ping-pongs, e.g. launch 0 of a timed region:
//     the pattern says the next launch follows at once (profiles/r04_driver_protocol.txt)                   -> overlap
//   * else (one stream; or a caller that alternates but synchronises between its launches, so that the second call in a row
//     finds the other set idle)                                                                             -> alone
// One rule for the three entry points (enqueue_verify used the bare event query until round 4).
static bool expect_overlap(bjj_ctx* c, const ScratchSet* S) {
  if (other_launch_in_flight(c, S)) { c->idle_alternations = 0; return true; }
  if (!streams_alternate(c, S)) { c->idle_alternations = 0; return false; }
  if (c->idle_alternations < 2) c->idle_alternations++;
  return c->idle_alternations < 2;
}
static int fixed_base_variant(bjj_ctx* c, const ScratchSet* S) {
  const int kv = c->k1_half_now ? 1 : (c->k1_variant >= 0 ? c->k1_variant : (expect_overlap(c, S) ? 1 : 0));
  c->last_k1 = kv;
  return kv;
}
// lanes per CU the launch may take: everything its shape can hold, or -- chunk launches of a host-pointer call, PipeSpec::k1_half --
// one 256-lane workgroup per CU, so that the neighbouring chunk's launch has the other slot
static int fixed_base_lanes(const bjj_ctx* c, int kv) {
  if (c->k1_half_now && kv == 1) return 256;
  return kv ? c->lanes_fixed_2x256 : c->lanes_fixed;
}
// completion mark of a call that used no scratch (bjj_sync waits for these)
static int mark_stream(bjj_ctx* c, hipStream_t st) {
  if (st == c->stream) return BJJ_OK;   // bjj_sync synchronises the context's own stream anyway
  StreamMark* m = nullptr;
  for (StreamMark& k : c->marks) if (k.used && k.stream == st) { m = &k; break; }
  if (!m) for (StreamMark& k : c->marks) if (!k.used) { m = &k; break; }
  if (!m) {   // more caller streams than slots: retire the oldest mark (wait for it, then reuse the slot)
    m = &c->marks[0];
    for (StreamMark& k : c->marks) if (k.last_use < m->last_use) m = &k;
    HIPCK(hipEventSynchronize(m->ev));
  }
  if (!m->ev) HIPCK(hipEventCreateWithFlags(&m->ev, hipEventDisableTiming));
  HIPCK(hipEventRecord(m->ev, st));
  m->stream = st; m->used = true; m->last_use = ++c->use_counter;
  return BJJ_OK;
}

// Per-XCD rings of free table slots (k_common.hpp): [0] head ticket, [1] tail ticket, [2] pops that gave up (SLOTQ_ERR),
// [16 + i] = slot id + 1.  (Re)built on the host: every slot free.
#define BJJ_SLOTQ_HDR 16
#define BJJ_SLOTQ_ERR 2
static int slot_queue_fill(bjj_ctx* c, u32* d_q, u32 cap) {
  const size_t stride = BJJ_SLOTQ_HDR + cap;
  std::vector<u32> h((size_t)c->xccs * stride, 0u);
  for (int x = 0; x < c->xccs; x++)
    for (u32 i = 0; i < cap; i++) h[(size_t)x * stride + BJJ_SLOTQ_HDR + i] = (u32)x * cap + i + 1u;   // slot id + 1
  HIPCK(hipMemcpyAsync(d_q, h.data(), h.size() * sizeof(u32), hipMemcpyHostToDevice, c->stream));   // pageable source: staged before the call returns
  HIPCK(hipStreamSynchronize(c->stream));
  return BJJ_OK;
}
// All rings of a context -- two per scratch set: verify's (cap slots per XCD) and the variable-base tiles' (cap2) -- live in ONE
// device block, so that the check below reads them back with a single copy.  The capacities are constants of the context
// (occupancy x CUs per XCD): the block is made once, when the first scratch set is sized.
static size_t slot_ring_words(const bjj_ctx* c, u32 cap) { return (size_t)c->xccs * (BJJ_SLOTQ_HDR + cap); }
static int slot_block_make(bjj_ctx* c, u32 cap, u32 cap2) {
  if (c->slot_block) return BJJ_OK;
  const size_t per_set = slot_ring_words(c, cap) + slot_ring_words(c, cap2);
  u32* blk = nullptr;
  HIPCK(hipMalloc((void**)&blk, BJJ_SCRATCH_SETS * per_set * sizeof(u32)));
  u32* host = nullptr;
  if (hipHostMalloc((void**)&host, BJJ_SCRATCH_SETS * per_set * sizeof(u32), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(blk);
    return set_err(BJJ_E_NOMEM, "slot queues: no pinned memory for the read-back buffer");
  }
  for (int k = 0; k < BJJ_SCRATCH_SETS; k++) {
    ScratchSet& S = c->set[k];
    S.slotq = blk + (size_t)k * per_set;           S.slot_cap = cap;
    S.slotq2 = S.slotq + slot_ring_words(c, cap);  S.slot_cap2 = cap2;
    int rc = slot_queue_fill(c, S.slotq, cap);
    if (!rc) rc = slot_queue_fill(c, S.slotq2, cap2);
    if (rc) { for (ScratchSet& T : c->set) { T.slotq = T.slotq2 = nullptr; T.slot_cap = T.slot_cap2 = 0; } (void)hipFree(blk); (void)hipHostFree(host); return rc; }
  }
  c->slot_block = blk; c->slot_block_words = BJJ_SCRATCH_SETS * per_set; c->err_words = host;
  return BJJ_OK;
}
// After the launches in question have completed: did a pop ever give up waiting for a slot?  ONE asynchronous copy of the ring
// block (35 KB) on the context's own non-blocking stream into pinned memory -- not four synchronous pageable copies through
// the legacy null stream, which also made the caller wait for other tenants' blocking streams -- and only when a
// kernel that uses the rings has been launched since the last check (rings_used: 2^20 fixed-base multiplications on host
// pointers take 1.5 ms, a check that costs 0.17 ms has no business there).  A ring is rebuilt only when its counter is
// non-zero.  Every entry point that synchronises for the caller ends with this check (bjj_sync, the host-pointer pipeline, the
// multi-GPU pipeline): a launch that worked on the overflow slot must never be reported as BJJ_OK.
static int ctx_check_slot_queues(bjj_ctx* c, const char* who) {
  if (!c->slot_block || !c->rings_used) return BJJ_OK;
  HIPCK(hipMemcpyAsync(c->err_words, c->slot_block, c->slot_block_words * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPCK(hipStreamSynchronize(c->stream));
  c->rings_used = false;
  unsigned long long starved = 0;
  for (ScratchSet& S : c->set) {
    struct Ring { u32* q; u32 cap; } rings[2] = {{S.slotq, S.slot_cap}, {S.slotq2, S.slot_cap2}};
    for (const Ring& r : rings) {
      const u32* h = c->err_words + (r.q - c->slot_block);
      unsigned long long bad = 0;
      for (int x = 0; x < c->xccs; x++) bad += h[(size_t)x * (BJJ_SLOTQ_HDR + r.cap) + BJJ_SLOTQ_ERR];
      if (bad) { int rc = slot_queue_fill(c, r.q, r.cap); if (rc) return rc; }
      starved += bad;
    }
  }
  if (starved)
    return set_err(BJJ_E_HIP, std::string(who) + ": " + std::to_string(starved) + " workgroup(s) gave up waiting for a per-lane table slot; the results of "
                   "the verify / variable-base launches since the last synchronising call are not valid (slot queues rebuilt)");
  return BJJ_OK;
}

static int ensure_scratch(bjj_ctx* c, ScratchSet* S, size_t n) {
  if (n > S->scratch_items) {
    if (S->scratch) { HIPCK(hipDeviceSynchronize()); HIPCK(hipFree(S->scratch)); S->scratch = nullptr; S->scratch_items = 0; }
    HIPCK(hipMalloc((void**)&S->scratch, n * 64));
    S->scratch_items = n;
  }
  if (n > S->slow_items) {
    if (S->slow) { HIPCK(hipDeviceSynchronize()); HIPCK(hipFree(S->slow)); S->slow = nullptr; S->slow_items = 0; }
    HIPCK(hipMalloc((void**)&S->slow, (n + 16) * sizeof(u32)));
    S->slow_items = n;
  }
  // slots of per-lane tables: K2 needs its resident lanes; verify (2 tables per lane) the waves that can be resident, rounded
  // up to a whole number per XCD (the slot queues are per XCD); plus ONE overflow slot per XCD behind the regular ones (what a
  // pop that gave up waiting continues on, k_common.hpp)
  const size_t cu_per_xcc = ((size_t)c->cus + c->xccs - 1) / c->xccs;
  const u32 cap = (u32)(cu_per_xcc * c->occ_verify);   // occ_verify counts waves
  const u32 cap2 = (u32)(cu_per_xcc * (size_t)(c->lanes_var / bjjk::var_base_block()));
  const size_t tv = (size_t)c->xccs * ((size_t)cap2 + 1) * (size_t)bjjk::var_base_block();
  const size_t tv_strided = cu_per_xcc * c->xccs * (size_t)c->lanes_var;          // the grid-strided form indexes by global lane
  const size_t te = (size_t)c->xccs * ((size_t)cap + 1) * 64 * 2;
  size_t threads = tv > te ? tv : te;
  if (tv_strided > threads) threads = tv_strided;
  { int rc = slot_block_make(c, cap, cap2); if (rc) return rc; }
  if (threads > S->vb_threads) {
    if (S->vb_tables) { HIPCK(hipDeviceSynchronize()); HIPCK(hipFree(S->vb_tables)); S->vb_tables = nullptr; S->vb_threads = 0; }
    HIPCK(hipMalloc((void**)&S->vb_tables, threads * VB_TABLE_WORDS_MAX * sizeof(u32)));
    S->vb_threads = threads;
  }
  return BJJ_OK;
}
// ---------------------------------------------------------------------------
// Host-pointer API plumbing.
//
// A call is cut into chunks; chunk k flows
//     H2D (copy stream s_in)  ->  kernels on lane k % 2 (stream / stream2)  ->  D2H (copy stream s_out)
// through device staging that holds the WHOLE batch (every array contiguous, up to BJJ_PIPE_STAGING_MB = 1 GB per call; a
// larger batch runs as consecutive super-batches), so nothing on the device is ever reused inside a call.
// * Pinned caller memory (bjj_host_alloc, bjj_host_register, or anything the HIP runtime reports as pinned host memory) is
//   copied from / to DIRECTLY: no staging copy at all, the bound is the slower PCIe direction (2^20 fixed-base
//   multiplications: 64 MB of results, 1.19 ms).  Round 4 staged everything through two pinned buffers with a memcpy on the
//   calling thread and ran at 17 % of the device rate.
// * A pageable array goes through a ring of pinned buffers; the memcpy between ring and caller memory is done by the
//   context's copy workers (CopyPool) while the calling thread only enqueues.  (A pageable hipMemcpy runs at ~3 GB/s.)
// * Two lanes = the context's two scratch sets: the kernels of consecutive chunks overlap on the chip like the two-stream
//   launches of the device-pointer API (verify in 2^18-item chunks on ONE stream ran 6.76 ms per chunk, on two 4.47 ms:
//   profiles/r04_throughput_vs_batch.txt).  The lanes carry kernels only.
// * What orders the stages is chosen so that NO mapping of streams onto hardware queues can hurt.  HIP binds a stream, at its
//   first use, to one of FOUR hardware queues of its priority (a fifth stream shares one), packets of a hardware queue run in
//   order, and an event record or a cross-stream wait behind an SDMA copy -- or a wait for a kernel on a copy stream -- is a
//   barrier packet there: a kernel of ANOTHER stream that lands behind such a packet waits for whatever the packet waits for.
//   The forms of this pipeline, in the order they were measured (profiles/r05_host_pipeline.txt; 2^20 fixed-base, pinned):
//     normal-priority copy streams, one event per chunk and stage: every kernel ran behind the previous chunk's D2H   2.7 ms
//     D2H in stream order behind the kernels (no events; the runtime then copies with a shader blit at half rate)       1.65 ms
//     copy streams and second lane of the HIGHEST priority (their own queue pool), s_out waiting for the kernels'
//     events on the device: 1.62 ms in a fresh process, but 12 % slower verifications in a process whose other
//     high-priority streams (the verify scan streams) had pushed the second lane onto s_out's hardware queue
//     the same with the wait moved to the HOST (ships): the calling thread, idle anyway, watches the chunks' kernel
//     events in order and enqueues each D2H when its kernels are done -- s_out's queue holds no packet at all          1.58 ms
//   What remains on the device: one event per chunk behind the H2D on s_in, which completes early (copies on s_in run in
//   order and far ahead of the kernels), and the lanes' kernels.  Apart from the context's own stream the pipeline occupies
//   no normal-priority queue: a library that parks streams there leaves the caller's streams to share what is left (two torch
//   streams first used after a host-pointer call landed on ONE hardware queue and their launches ran one after the other).
//   The verify scans of the pipeline's chunks run in line (on their priority streams they were no faster here: 19.0 vs 19.1 ms).
// * The first chunk is small (2^15 items: the copy-out engine, which bounds a copy-bound call, starts 0.1 ms after the call)
//   and the size doubles up to 2^18; a remainder below half a chunk is merged into the
//   last chunk (a small last launch leaves the chip half empty).  BJJ_PIPE_FIRST_CHUNK / BJJ_PIPE_CHUNK (items) override.
// ---------------------------------------------------------------------------
// The cap was 2^18 until round 6.  With 2^18-item chunks (16 MB copy-outs for 64-byte results) the copy engines' device-to-host rate has two
// states per process -- 56 or 44 GB/s, the SOC clock domain awake or asleep between calls, depending on what the process did before
// (profiles/r06_host_d2h_power_states.txt): 2^20 fixed-base multiplications take 1.575 or 1.92 ms.  With 2^17-item chunks the slow state
// does not occur (1.565-1.60 ms in every sequence tried) and the fast state loses nothing.  The compressed fixed-base forms keep 2^18
// (1.155 ms in the fast state against 1.26 with 2^17; 1.30-1.36 in the slow state with either).
#define BJJ_PIPE_CHUNK ((size_t)1 << 17)
#define BJJ_PIPE_FIRST_CHUNK ((size_t)1 << 15)
struct PipeSpec {
  int n_in, n_out;
  const uint8_t* in[4]; size_t in_stride[4];
  uint8_t* out[4];      size_t out_stride[4];
  bool secret;          // inputs are key material: wipe the staging buffers when the call is done
  struct PipeExtra* extra = nullptr;   // stages beside the chunk launches (below)
  bool out_at_end = false;             // outputs leave the device once, after everything (they are not final chunk by chunk)
  size_t first_chunk = 0, max_chunk = 0;   // chunk schedule of this entry point (items; 0 = the context's, which the environment overrides)
  size_t extra_dev_per_item = 0;           // bytes of device staging per item for `extra` (PipeExtra::d_extra), beside the arrays
  // Kernel-bound calls whose launches are work-conserving among themselves (one tile / group per workgroup):
  //   tail_chunk            the LAST chunk has at most this many items (0 = no rule): its copy-out is the only one nothing hides
  //   last_on_priority_lane the lanes are the context's stream (normal priority) and stream2 (highest): the hardware serves the
  //                         priority lane's workgroups first, so ITS chain of launches ends first and the other lane's last launch
  //                         runs its tail alone.  With the last chunk on the priority lane the other lane's launches fill every tail
  //                         but the very last one -- as in ONE launch (profiles/r06_var_base_host_schedule.txt)
  size_t tail_chunk = 0;
  bool last_on_priority_lane = false;
  // The kernels of this entry point write every output byte exactly once and never read it (their scratch is elsewhere): when ALL
  // output arrays are pinned, the launches get the arrays' DEVICE MAPPINGS instead of staging and there is no copy-out stage at all.
  // Only for kernel-bound calls: a kernel's stores to mapped host memory run at the copy engines' rate (54.9 GB/s), but a workgroup
  // holds its slot until PCIe has taken them -- K1, which is copy-bound, lost 5 % this way (profiles/r05_host_pipeline.txt).
  bool zero_copy_out = false;
  // The first kernel of a chunk reads every input byte exactly once, coalesced, at the start of an item's work: when ALL input arrays are
  // pinned, the launches read them through the arrays' DEVICE MAPPINGS and there is no copy-in stage -- the first kernel starts at once
  // instead of behind a copy, and the chain of launches is no longer paced by the copy engine (43 GB/s over the small copies of a
  // schedule) but by the kernels.  For launch chains that outrun their copy-in: K1, whose 2^20 items are 0.55 ms of kernel behind
  // 0.77 ms of copy-in (profiles/r06_fb_zero_copy_in.txt).  Not with `extra` stages (they wait for copies).
  bool zero_copy_in = false;
  // K1 chunk launches (fixed base, public keys): a launch of K1 is persistent -- every lane walks its items, then ONE epilogue per
  // workgroup (inversion, phase 2) -- and its ramp, epilogue and tail cost about one round of multiplications (~70 us) whatever
  // its size.  A chunk launch that fills both workgroup slots of every CU runs that fixed part with nothing beside it; one that
  // takes ONE slot per CU (256 lanes) shares each CU with the neighbouring chunk's launch on the other lane, out of phase by the
  // pacing of the copies, and the fixed part of one is covered by the main loop of the other -- as on two caller streams
  // (profiles/r03_ab_k1_2x256_two_streams.txt).  Calls of >= 2 chunks only (profiles/r06_fb_host_half_slots.txt).
  bool k1_half = false;
  // Short calls (a call of ONE item is the reference's single-item API): with at most this many items, one chunk and every array pinned and 16-byte aligned,
  // the kernels read the inputs AND store the results through the arrays' device mappings -- no copy, no event hop, the host does not come back between
  // the kernel and a copy-out: launch and synchronise (20-30 us of a 100-600 us call).  Only for entry points whose short calls run kernels that write every
  // output byte once and read none of it back (csrc/k_small.hip).  0 = never.
  size_t small_direct_max = 0;
};
// Work of a pipelined call that does not belong to ONE chunk.  All three run on the calling thread while it enqueues:
//   begin          once per super-batch, before the first copy; d_in / d_out = the staging arrays of the whole super-batch
//   chunk_arrived  behind the H2D of items lo .. lo+cnt-1 (ev_in of the chunk has been recorded: make a stream wait for it)
//   all_arrived    behind the last chunk; whatever it enqueues is covered by ev_tail, which the call waits for before it copies
//                  `out_at_end` outputs and returns
//   all_launched   behind the last chunk's launch -- for stages that need what the chunks' own launches produce (the wire-format verifier scans
//                  points its chunks' launches have decompressed); a stage that uses it records ev_tail there instead of in all_arrived
//   finish         last: every copy of the super-batch has landed in the caller's arrays (host_out = where its outputs begin), every
//                  stream of the pipeline and ev_tail are done -- host-side work on the results
struct PipeExtra {
  void* d_extra = nullptr;     // n * PipeSpec::extra_dev_per_item bytes of the super-batch's device staging (set before begin)
  bool zero_copy = false;      // begin's d_out[] are the device mappings of the caller's pinned output arrays (PipeSpec::zero_copy_out)
  virtual int begin(size_t n, void** d_in, void** d_out) = 0;
  virtual int chunk_arrived(size_t lo, size_t cnt, hipEvent_t arrived) = 0;
  virtual int all_arrived(hipEvent_t ev_tail) = 0;
  virtual int all_launched(hipEvent_t ev_tail) { (void)ev_tail; return BJJ_OK; }   // behind the LAST chunk's launch (what it enqueues is covered by ev_tail, too)
  virtual int finish(uint8_t* const* host_out, size_t n) { (void)host_out; (void)n; return BJJ_OK; }
  virtual ~PipeExtra() {}
};
static size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }
static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
static void secure_bzero(void* p, size_t n) {
  memset(p, 0, n);
  __asm__ __volatile__("" : : "r"(p) : "memory");   // the stores must not be elided as dead
}

// ---- pinned host memory ---------------------------------------------------------------------------------------------------
// Ranges handed out by bjj_host_alloc (hipHostMalloc) or pinned in place by bjj_host_register (hipHostRegister): process-wide,
// any context may copy from / to them.  Memory pinned by somebody else (torch's pin_memory, the caller's own hipHostMalloc) is
// recognised through the driver's pointer attributes.  A wrong answer can only cost speed: hipMemcpyAsync accepts any host
// pointer, and every call waits for its copies before it returns.
struct HostRange { uintptr_t lo, hi; bool owned; };
static std::mutex g_host_mu;
static std::vector<HostRange> g_host_ranges;
static bool host_range_registered(const void* p, size_t bytes) {
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  std::lock_guard<std::mutex> lk(g_host_mu);
  for (const HostRange& r : g_host_ranges) if (lo >= r.lo && hi <= r.hi) return true;
  return false;
}
static bool driver_attr(const void* p, hipPointerAttribute_t* a) {
  memset(a, 0, sizeof(*a));
  if (hipPointerGetAttributes(a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // an ordinary malloc'd pointer
  return true;
}
static bool host_range_pinned(const void* p, size_t bytes) {
  if (!bytes || host_range_registered(p, bytes)) return true;
  hipPointerAttribute_t a0, a1;
  if (!driver_attr(p, &a0) || a0.type != hipMemoryTypeHost) return false;
  if (!driver_attr((const uint8_t*)p + bytes - 1, &a1) || a1.type != hipMemoryTypeHost) return false;
  // both ends are pinned: one mapping?  (two pinned allocations with a pageable hole between them would pass the end test)
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (a0.devicePointer && hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a0.devicePointer) == hipSuccess)
    return (uintptr_t)a0.devicePointer - (uintptr_t)base + bytes <= size;
  (void)hipGetLastError();
  return a0.devicePointer && a1.devicePointer && (uintptr_t)a1.devicePointer - (uintptr_t)a0.devicePointer == bytes - 1;
}

static size_t env_items(const char* name, size_t dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const unsigned long long v = strtoull(e, nullptr, 0);
  return v >= 64 && v <= ((size_t)1 << 24) ? ((size_t)v + 63) & ~(size_t)63 : dflt;
}
// The context's high-priority streams, created AND first used in a fixed order when the context is made.  HIP binds a stream to a
// hardware queue at its first use: the streams of one priority share FOUR queues, a fifth stream takes the queue of the first, and
// packets of streams that share a queue run one after the other.  Created on demand, the assignment depended on what the caller
// did first: after two-stream verifications (both sets' scan streams in use) the pipeline's second lane landed on the first scan
// stream's queue -- the stream the host-pointer pipelines run their scans and exact launches on, whose small kernels wait for
// wave slots while the lanes' kernels fill the chip -- and a chunk's launch on that lane queued behind them: 2^20 variable-base
// multiplications on pinned memory took 17.0 ms instead of 15.9 (profiles/r06_var_base_host_schedule.txt).  Now: copy-in, copy-out,
// second lane, first scan stream = queues 0..3, always; the second set's scan stream (two-stream device-pointer callers only) is
// the one that shares -- with the copy-in stream, which carries nothing while a device-pointer launch is what the caller is doing.
//   * the second lane is a high-priority stream as well: a library that parks two normal-priority streams in the four queues of
//     that priority (the context's own stream is one) leaves the caller's streams to share what is left -- two torch streams first
//     used after a host-pointer call landed on ONE hardware queue and their launches ran one after the other
//     (tools/queue_map_probe.py, profiles/r05_host_pipeline.txt)
static int ensure_scan_stream(ScratchSet* S);
static int ensure_pipe_streams(bjj_ctx* c) {
  if (c->s_in && c->s_out && c->stream2 && c->ev_tail && c->set[0].scan_stream) return BJJ_OK;
  int least = 0, greatest = 0;
  HIPCK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  int prio_in = greatest, prio_out = greatest;
  if (const char* e = getenv("BJJ_PIPE_COPY_PRIORITY")) {   // developer: "nn" / "hn" / "nh" / "hh" = normal / high for s_in, s_out
    if (e[0] == 'n') prio_in = 0;
    if (e[0] && e[1] == 'n') prio_out = 0;
  }
  if (!c->ev_tail) HIPCK(hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
  hipStream_t* order[3] = {&c->s_in, &c->s_out, &c->stream2};
  const int prio[3] = {prio_in, prio_out, greatest};
  for (int k = 0; k < 3; k++)
    if (!*order[k]) {
      HIPCK(hipStreamCreateWithPriority(order[k], hipStreamNonBlocking, prio[k]));
      HIPCK(hipEventRecord(c->ev_tail, *order[k]));              // first use: binds the hardware queue
      HIPCK(hipStreamSynchronize(*order[k]));
    }
  if (!c->set[0].scan_stream) {
    int rc = ensure_scan_stream(&c->set[0]); if (rc) return rc;
    HIPCK(hipEventRecord(c->ev_tail, c->set[0].scan_stream));
    HIPCK(hipStreamSynchronize(c->set[0].scan_stream));
  }
  return BJJ_OK;
}
// streams of the pipeline, events for `chunks` chunks, dev_bytes of device staging and -- only when a pageable array takes
// part -- the pinned rings (in_ring / out_ring bytes per slot) and the copy workers
static int ensure_pipe(bjj_ctx* c, size_t chunks, size_t dev_bytes, size_t in_ring, size_t out_ring) {
  ENTER_DEVICE(c->device);
  // first use: the knobs first (they cannot fail), then whatever of streams / event is still missing; `pipe_ready` only when all of it
  // exists -- a call after a failed first use comes through here again instead of running with a zero chunk size
  if (!c->pipe_ready) {
    c->pipe_env_schedule = getenv("BJJ_PIPE_CHUNK") || getenv("BJJ_PIPE_FIRST_CHUNK");
    c->pipe_chunk = env_items("BJJ_PIPE_CHUNK", BJJ_PIPE_CHUNK);
    c->pipe_first = env_items("BJJ_PIPE_FIRST_CHUNK", BJJ_PIPE_FIRST_CHUNK);
    if (c->pipe_first > c->pipe_chunk) c->pipe_first = c->pipe_chunk;
    c->pipe_budget = (size_t)1 << 30;
    if (const char* e = getenv("BJJ_PIPE_STAGING_MB")) { const long v = atol(e); if (v >= 1 && v <= 65536) c->pipe_budget = (size_t)v << 20; }
    if (const char* e = getenv("BJJ_HOST_FORCE_STAGED")) c->force_staged = e[0] == '1';
    if (const char* e = getenv("BJJ_PIPE_SCAN")) c->pipe_scan_inline = e[0] != 'p';
    if (const char* e = getenv("BJJ_PIPE_ZERO_COPY")) c->pipe_zero_copy = e[0] != '0';
    if (const char* e = getenv("BJJ_PIPE_ZERO_COPY_IN")) c->pipe_zero_copy_in = e[0] != '0';
    if (const char* e = getenv("BJJ_PIPE_SMALL_DIRECT")) c->pipe_small_direct = e[0] != '0';
    { int rc = ensure_pipe_streams(c); if (rc) return rc; }
    c->pipe_ready = true;
  }
  try {
    while (c->ev_in.size() < chunks) {
      hipEvent_t e = nullptr;
      HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_in.push_back(e);
      e = nullptr;
      HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_k.push_back(e);
      e = nullptr;
      HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_out.push_back(e);
    }
  } catch (...) { return set_err(BJJ_E_NOMEM, "host-pointer pipeline: out of host memory"); }
  if (dev_bytes > c->pipe_bytes) {
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipStreamSynchronize(c->stream2));
    if (c->dstage) { HIPCK(hipFree(c->dstage)); c->dstage = nullptr; c->pipe_bytes = 0; }
    HIPCK(hipMalloc((void**)&c->dstage, dev_bytes));
    c->pipe_bytes = dev_bytes;
  }
  auto grow_ring = [&](uint8_t** ring, size_t* have, size_t want) -> int {
    if (want <= *have) return BJJ_OK;
    for (int b = 0; b < BJJ_PIPE_BUFS; b++) {
      if (ring[b]) { secure_bzero(ring[b], *have); HIPCK(hipHostFree(ring[b])); ring[b] = nullptr; }
    }
    *have = 0;
    for (int b = 0; b < BJJ_PIPE_BUFS; b++) HIPCK(hipHostMalloc((void**)&ring[b], want, hipHostMallocDefault));
    *have = want;
    return BJJ_OK;
  };
  { int rc = grow_ring(c->pin_in, &c->pin_in_bytes, in_ring); if (rc) return rc; }
  { int rc = grow_ring(c->pin_out, &c->pin_out_bytes, out_ring); if (rc) return rc; }
  if ((in_ring || out_ring) && !c->pool) {
    int want = 4;
    if (const char* e = getenv("BJJ_STAGE_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 32) want = v; }
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && (unsigned)want > hw) want = (int)hw;
    c->pool = new (std::nothrow) CopyPool();
    if (!c->pool || !c->pool->start(want)) { delete c->pool; c->pool = nullptr; return set_err(BJJ_E_NOMEM, "host-pointer pipeline: cannot start the copy workers"); }
  }
  return BJJ_OK;
}
// One super-batch of at most `cap` items (everything fits in the device staging).  launch(d_in[], d_out[], count, stream)
// enqueues the kernels of one chunk on `stream`.
template <typename Launch>
static int run_super_batch_body(bjj_ctx* c, size_t n, const PipeSpec& sp, const bool* in_direct, const bool* out_direct, u32 n_staged, Launch& launch,
                                u32* chunks_out);
// The body allocates (chunk schedule, copy groups, the workers' queue): a std::bad_alloc must not cross the extern "C" boundary,
// and nothing may still be copying into the caller's memory or out of the rings when the call returns.
template <typename Launch>
static int run_super_batch(bjj_ctx* c, size_t n, const PipeSpec& sp, const bool* in_direct, const bool* out_direct, u32 n_staged, Launch& launch,
                           u32* chunks_out) {
  try {
    return run_super_batch_body(c, n, sp, in_direct, out_direct, n_staged, launch, chunks_out);
  } catch (...) {
    // only the allocations at the head of the body can throw (the chunk schedule, the copy groups, the timing marks): no worker
    // holds a task yet (CopyPool::submit does not throw), the device has at most a list reset queued
    hipDeviceSynchronize();
    (void)hipGetLastError();
    return set_err(BJJ_E_NOMEM, "host-pointer pipeline: out of host memory");
  }
}
template <typename Launch>
static int run_super_batch_body(bjj_ctx* c, size_t n, const PipeSpec& sp, const bool* in_direct, const bool* out_direct, u32 n_staged, Launch& launch,
                                u32* chunks_out) {
  // ---- chunk schedule: first, 2 first, 4 first ... capped at pipe_chunk; a remainder below half a chunk joins the last chunk
  std::vector<size_t> lo_of;     // lo_of[ch] .. lo_of[ch + 1]
  {
    const size_t sz_max = (sp.max_chunk && !c->pipe_env_schedule) ? sp.max_chunk : c->pipe_chunk;
    size_t lo = 0, sz = (sp.first_chunk && !c->pipe_env_schedule) ? sp.first_chunk : c->pipe_first;
    if (sz > sz_max) sz = sz_max;
    // a separate small last chunk only when there is a schedule to speak of in front of it
    const size_t tail = (sp.tail_chunk && !c->pipe_env_schedule && n >= 4 * sp.tail_chunk) ? sp.tail_chunk : 0;
    const size_t body_n = n - tail;
    while (lo < body_n) {
      size_t take = sz < body_n - lo ? sz : body_n - lo;
      if (body_n - lo - take < sz / 2) take = body_n - lo;          // what would be left is small: take it along
      lo_of.push_back(lo);
      lo += take;
      if (sz < sz_max) sz = sz * 2 < sz_max ? sz * 2 : sz_max;
    }
    if (tail) lo_of.push_back(body_n);
    lo_of.push_back(n);
    // developer: BJJ_PIPE_SCHEDULE="a,b,c,..." = the chunk sizes themselves (items; the last one repeats, the remainder joins the last chunk)
    static const std::vector<size_t> forced = [] {
      std::vector<size_t> v;
      if (const char* e = getenv("BJJ_PIPE_SCHEDULE"))
        for (const char* p = e; *p;) { char* q = nullptr; const unsigned long long x = strtoull(p, &q, 0); if (q == p) break; if (x >= 64) v.push_back((size_t)x & ~(size_t)63); p = *q ? q + 1 : q; }
      return v;
    }();
    if (!forced.empty()) {
      lo_of.clear();
      size_t at = 0, k = 0;
      while (at < n) { lo_of.push_back(at); at += forced[k < forced.size() ? k : forced.size() - 1]; k++; }
      lo_of.push_back(n);
    }
  }
  const size_t nchunks = lo_of.size() - 1;
  static const int parity_env = [] { const char* e = getenv("BJJ_PIPE_LANE_PARITY"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();   // developer A/B
  const size_t lane_flip = parity_env >= 0 ? (size_t)parity_env : (sp.last_on_priority_lane ? ((nchunks - 1) & 1) ^ 1 : 0);   // chunk ch runs on lane (ch + flip) & 1; lane 1 = stream2
  size_t max_chunk = 0;
  for (size_t ch = 0; ch < nchunks; ch++) if (lo_of[ch + 1] - lo_of[ch] > max_chunk) max_chunk = lo_of[ch + 1] - lo_of[ch];
  // device staging: array i of the whole super-batch at d_off[i]; pinned ring slots: the staged arrays of ONE chunk
  size_t d_in_off[4], d_out_off[4], r_in_off[4], r_out_off[4], dev_tot = 0, in_ring = 0, out_ring = 0;
  for (int i = 0; i < sp.n_in; i++) { d_in_off[i] = dev_tot; dev_tot += up256(n * sp.in_stride[i]); }
  for (int i = 0; i < sp.n_out; i++) { d_out_off[i] = dev_tot; dev_tot += up256(n * sp.out_stride[i]); }
  const size_t d_extra_off = dev_tot;
  dev_tot += up256(n * sp.extra_dev_per_item);
  for (int i = 0; i < sp.n_in; i++) if (!in_direct[i]) { r_in_off[i] = in_ring; in_ring += up16(max_chunk * sp.in_stride[i]); }
  // outputs that leave at the end go through ONE ring slot that holds the whole array
  for (int i = 0; i < sp.n_out; i++) if (!out_direct[i]) { r_out_off[i] = out_ring; out_ring += up16((sp.out_at_end ? n : max_chunk) * sp.out_stride[i]); }
  { int rc = ensure_pipe(c, nchunks, dev_tot, in_ring, out_ring); if (rc) return rc; }
  const bool chunk_out_ring = out_ring && !sp.out_at_end;   // pageable outputs travel chunk by chunk through the ring
  // zero-copy outputs: every output array pinned AND mapped into the device's address space
  uint8_t* mapped_out[4] = {nullptr, nullptr, nullptr, nullptr};
  const bool small_direct = sp.small_direct_max && n <= sp.small_direct_max && nchunks == 1 && !sp.extra && c->pipe_small_direct;
  bool zc = (sp.zero_copy_out || small_direct) && !out_ring && sp.n_out > 0 && !sp.out_at_end && c->pipe_zero_copy;
  for (int i = 0; i < sp.n_out && zc; i++) {
    void* dp = nullptr;
    if (!out_direct[i] || ((uintptr_t)sp.out[i] & 15u) || hipHostGetDevicePointer(&dp, sp.out[i], 0) != hipSuccess || !dp) { (void)hipGetLastError(); zc = false; }   // (the kernels move 16-byte words)
    mapped_out[i] = (uint8_t*)dp;
  }
  uint8_t* mapped_in[4] = {nullptr, nullptr, nullptr, nullptr};
  // Calls of one or two chunks: every chunk.  Longer calls: the FIRST chunk only -- the head of the chain of launches starts at once instead
  // of behind its copy, and the copy-in of the chunks behind it, which keeps the two lanes' launches out of phase, starts earlier too.
  bool zi = (sp.zero_copy_in || small_direct) && !in_ring && sp.n_in > 0 && !sp.extra && c->pipe_zero_copy_in;
  for (int i = 0; i < sp.n_in && zi; i++) {
    void* dp = nullptr;
    if (!in_direct[i] || ((uintptr_t)sp.in[i] & 15u) || hipHostGetDevicePointer(&dp, (void*)sp.in[i], 0) != hipSuccess || !dp) { (void)hipGetLastError(); zi = false; }   // (the kernels move 16-byte words)
    mapped_in[i] = (uint8_t*)dp;
  }
  static const int zi_first_env = [] { const char* e = getenv("BJJ_PIPE_ZERO_COPY_IN_FIRST"); return e ? atoi(e) : 1; }();   // developer A/B: leading chunks of a long call that read in place
  if (small_direct && !(zi && zc)) { zi = zi && sp.zero_copy_in; zc = zc && sp.zero_copy_out; }   // both directions or the entry point's own rule
  const size_t zi_chunks = !zi ? 0 : (nchunks <= 2 ? nchunks : (size_t)zi_first_env);
  c->last_host_zero_copy = (zc ? 1u : 0u) | (zi_chunks == nchunks ? 2u : 0u);
  c->k1_half_now = sp.k1_half && nchunks >= 2;
  if (sp.extra) {
    void* bi[4]; void* bo[4];
    for (int i = 0; i < sp.n_in; i++) bi[i] = c->dstage + d_in_off[i];
    for (int i = 0; i < sp.n_out; i++) bo[i] = zc ? (void*)mapped_out[i] : (void*)(c->dstage + d_out_off[i]);
    sp.extra->zero_copy = zc;
    sp.extra->d_extra = sp.extra_dev_per_item ? c->dstage + d_extra_off : nullptr;
    int rc = sp.extra->begin(n, bi, bo); if (rc) return rc;
  }
  *chunks_out += (u32)nchunks;
  CopyPool* pool = c->pool;
  std::vector<CopyGroup> g_in(nchunks), g_out(nchunks);
  auto cnt_of = [&](size_t ch) { return lo_of[ch + 1] - lo_of[ch]; };
  // BJJ_PIPE_TRACE=1 (developer): host-side timestamps of every enqueue step on stderr -- tells a host thread that blocks inside
  // an "asynchronous" runtime call from a device-side dependency, which a kernel / copy trace cannot
  static const bool trace = [] { const char* e = getenv("BJJ_PIPE_TRACE"); return e && e[0] == '1'; }();
  const auto tr0 = std::chrono::steady_clock::now();
  auto tr = [&](const char* what, size_t ch) {
    if (trace) fprintf(stderr, "[pipe] %8.1f us  chunk %zu  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(), ch, what);
  };
  // pageable inputs of chunk ch -> pin_in[ch % BUFS] (workers); the slot is free once the H2D of chunk ch - BUFS has run
  auto submit_in = [&](size_t ch) -> int {
    const int b = (int)(ch % BJJ_PIPE_BUFS);
    if (!in_ring) return BJJ_OK;
    if (ch >= BJJ_PIPE_BUFS) HIPCK(hipEventSynchronize(c->ev_in[ch - BJJ_PIPE_BUFS]));
    for (int i = 0; i < sp.n_in; i++)
      if (!in_direct[i]) pool->submit(c->pin_in[b] + r_in_off[i], sp.in[i] + lo_of[ch] * sp.in_stride[i], cnt_of(ch) * sp.in_stride[i], &g_in[ch]);
    return BJJ_OK;
  };
  // chunk ch has left the device: pageable outputs pin_out[ch % BUFS] -> caller (workers)
  size_t harvested = 0;           // chunks whose copy-out has been submitted (in order)
  auto harvest = [&](size_t ch, bool block) -> int {
    const int b = (int)(ch % BJJ_PIPE_BUFS);
    if (block) HIPCK(hipEventSynchronize(c->ev_out[ch]));
    else {
      const hipError_t q = hipEventQuery(c->ev_out[ch]);
      (void)hipGetLastError();
      if (q != hipSuccess) return 1;   // not yet
    }
    for (int i = 0; i < sp.n_out; i++)
      if (!out_direct[i]) pool->submit(sp.out[i] + lo_of[ch] * sp.out_stride[i], c->pin_out[b] + r_out_off[i], cnt_of(ch) * sp.out_stride[i], &g_out[ch]);
    return BJJ_OK;
  };
  // the pageable results of chunk ch are in the caller's memory: its pin_out slot is free
  auto finish_out = [&](size_t ch) -> int {
    while (harvested <= ch) { int r = harvest(harvested, true); if (r) return r; harvested++; }
    pool->wait(&g_out[ch]);
    return BJJ_OK;
  };
  // ... and, in the same developer mode, timing events behind every stage (device-side timeline of the call, printed at its end)
  std::vector<hipEvent_t> tev(1 + 3 * nchunks, nullptr);
  auto tmark = [&](size_t slot, hipStream_t st) {
    if (!trace) return;
    if (hipEventCreate(&tev[slot]) == hipSuccess) hipEventRecord(tev[slot], st);
  };
  tmark(0, c->s_in);   // t = 0
  // H2D of chunk ch on s_in, its kernels on lane ch % 2 behind the copy's event, ev_k[ch] behind the kernels
  auto enqueue_front = [&](size_t ch) -> int {
    const int b = (int)(ch % BJJ_PIPE_BUFS);
    const size_t lo = lo_of[ch], cnt = cnt_of(ch);
    tr("enqueue begin", ch);
    const bool zi = ch < zi_chunks;   // this chunk's launch reads the caller's arrays in place
    for (int i = 0; i < sp.n_in && !zi; i++)
      HIPCK(hipMemcpyAsync(c->dstage + d_in_off[i] + lo * sp.in_stride[i], in_direct[i] ? sp.in[i] + lo * sp.in_stride[i] : c->pin_in[b] + r_in_off[i],
                           cnt * sp.in_stride[i], hipMemcpyHostToDevice, c->s_in));
    if (!zi) HIPCK(hipEventRecord(c->ev_in[ch], c->s_in));
    tmark(1 + 3 * ch, c->s_in);                                // H2D done
    if (sp.extra) {
      int r = sp.extra->chunk_arrived(lo, cnt, c->ev_in[ch]); if (r) return r;
      if (ch + 1 == nchunks) { r = sp.extra->all_arrived(c->ev_tail); if (r) return r; }
    }
    hipStream_t lane = ((ch + lane_flip) & 1) ? c->stream2 : c->stream;
    if (!zi) HIPCK(hipStreamWaitEvent(lane, c->ev_in[ch], 0));
    void* d_in[4]; void* d_out[4];
    for (int i = 0; i < sp.n_in; i++) d_in[i] = (zi ? mapped_in[i] : c->dstage + d_in_off[i]) + lo * sp.in_stride[i];
    for (int i = 0; i < sp.n_out; i++) d_out[i] = (zc ? mapped_out[i] : c->dstage + d_out_off[i]) + lo * sp.out_stride[i];
    int r = launch(d_in, d_out, cnt, (void*)lane); if (r) return r;
    if (sp.extra && ch + 1 == nchunks) { r = sp.extra->all_launched(c->ev_tail); if (r) return r; }
    HIPCK(hipEventRecord(c->ev_k[ch], lane));                  // behind a kernel: its completion signal, no extra packet
    tmark(2 + 3 * ch, lane);                                   // kernels done
    tr("kernels enqueued", ch);
    return BJJ_OK;
  };
  // D2H of chunk ch, enqueued by the HOST once it has seen the chunk's kernels complete: the copy carries no device-side wait,
  // so s_out's hardware queue never holds a packet that another stream's kernel could get stuck behind
  auto enqueue_out = [&](size_t ch) -> int {
    const int b = (int)(ch % BJJ_PIPE_BUFS);
    const size_t lo = lo_of[ch], cnt = cnt_of(ch);
    if (sp.out_at_end || zc) { tr("kernels seen complete", ch); return BJJ_OK; }   // nothing leaves chunk by chunk / the kernels wrote the caller's arrays themselves
    if (chunk_out_ring && ch >= BJJ_PIPE_BUFS) { int r = finish_out(ch - BJJ_PIPE_BUFS); if (r) return r; }   // frees pin_out[b]
    for (int i = 0; i < sp.n_out; i++)
      HIPCK(hipMemcpyAsync(out_direct[i] ? sp.out[i] + lo * sp.out_stride[i] : c->pin_out[b] + r_out_off[i], c->dstage + d_out_off[i] + lo * sp.out_stride[i],
                           cnt * sp.out_stride[i], hipMemcpyDeviceToHost, c->s_out));
    if (chunk_out_ring) HIPCK(hipEventRecord(c->ev_out[ch], c->s_out));   // only the staged path needs to know when ONE chunk has arrived
    tmark(3 + 3 * ch, c->s_out);                               // D2H done
    tr("D2H enqueued", ch);
    return BJJ_OK;
  };
  size_t outs = 0;                // chunks whose D2H has been enqueued (in order)
  auto drain_kernels = [&](bool block, size_t upto) -> int {   // D2H for the chunks < upto whose kernels have completed; block: wait for them
    while (outs < upto) {
      if (block) HIPCK(hipEventSynchronize(c->ev_k[outs]));
      else {
        const hipError_t q = hipEventQuery(c->ev_k[outs]);
        (void)hipGetLastError();
        if (q != hipSuccess) break;
      }
      { int r = enqueue_out(outs); if (r) return r; }
      outs++;
      while (chunk_out_ring && harvested < outs) {   // results that have already arrived: start their copy-out, do not wait
        const int r = harvest(harvested, false);
        if (r == 1) break;
        if (r) return r;
        harvested++;
      }
    }
    return BJJ_OK;
  };
  size_t fronts = 0;              // chunks whose H2D + kernels have been enqueued
  auto body = [&]() -> int {
    size_t next_in = 0;           // next chunk whose pageable inputs are handed to the workers (one chunk ahead of the enqueue)
    for (size_t ch = 0; ch < nchunks; ch++) {
      while (next_in < nchunks && next_in <= ch + 1) { int r = submit_in(next_in); if (r) return r; next_in++; }
      if (in_ring) pool->wait(&g_in[ch]);
      { int r = enqueue_front(ch); if (r) return r; }
      fronts = ch + 1;
      { int r = drain_kernels(false, fronts); if (r) return r; }   // only chunks that are enqueued can be drained
    }
    { int r = drain_kernels(true, nchunks); if (r) return r; }
    if (chunk_out_ring) for (size_t ch = harvested; ch < nchunks; ch++) { int r = finish_out(ch); if (r) return r; }
    if (chunk_out_ring) for (size_t ch = 0; ch < nchunks; ch++) pool->wait(&g_out[ch]);
    if (sp.extra) { HIPCK(hipEventSynchronize(c->ev_tail)); tr("extra stages complete", nchunks); }
    if (sp.out_at_end) {   // the whole output arrays, once: pinned -> directly, pageable -> ring slot 0 -> workers
      for (int i = 0; i < sp.n_out; i++)
        HIPCK(hipMemcpyAsync(out_direct[i] ? sp.out[i] : c->pin_out[0] + r_out_off[i], c->dstage + d_out_off[i], n * sp.out_stride[i], hipMemcpyDeviceToHost, c->s_out));
      if (out_ring) {
        HIPCK(hipStreamSynchronize(c->s_out));
        for (int i = 0; i < sp.n_out; i++) if (!out_direct[i]) pool->submit(sp.out[i], c->pin_out[0] + r_out_off[i], n * sp.out_stride[i], &g_out[0]);
        pool->wait(&g_out[0]);
      }
    }
    HIPCK(hipStreamSynchronize(c->s_out));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipStreamSynchronize(c->stream2));
    return BJJ_OK;
  };
  int rc = body();
  tr("all chunks finished", nchunks);
  if (!rc && sp.extra) { rc = sp.extra->finish(sp.out, n); tr("extra stages finished on the host", nchunks); }
  if (trace && !rc) {
    for (size_t ch = 0; ch < nchunks; ch++) {
      float t[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++) if (tev[0] && tev[1 + 3 * ch + k]) hipEventElapsedTime(&t[k], tev[0], tev[1 + 3 * ch + k]);
      fprintf(stderr, "[pipe-dev] chunk %zu (%7zu items)  H2D done %7.1f us  kernels done %7.1f us  D2H done %7.1f us\n", ch, cnt_of(ch), t[0] * 1e3, t[1] * 1e3, t[2] * 1e3);
    }
  }
  for (hipEvent_t e : tev) if (e) hipEventDestroy(e);
  if (rc) {   // error path: nothing may still be writing into the caller's memory or reading the rings when we return
    hipStreamSynchronize(c->s_in); hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2); hipStreamSynchronize(c->s_out);
    if (sp.extra) hipDeviceSynchronize();   // whatever the extra stages had enqueued on their own streams
    (void)hipGetLastError();
    if (pool) for (size_t ch = 0; ch < nchunks; ch++) { pool->wait(&g_in[ch]); pool->wait(&g_out[ch]); }
  }
  if (sp.secret) {  // key material went through the staging levels: wipe them (also on the error path) -- what THIS call can have used of them: the
    // rings keep the size of the largest call so far, and wiping all of it made a one-key call after a large one cost 0.3 ms more than its kernel
    hipStreamSynchronize(c->s_in); hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2); hipStreamSynchronize(c->s_out);
    if (c->dstage) hipMemsetAsync(c->dstage, 0, dev_tot, c->stream);
    const size_t slots = nchunks < (size_t)BJJ_PIPE_BUFS ? nchunks : (size_t)BJJ_PIPE_BUFS;
    for (size_t b = 0; b < slots; b++) {
      if (c->pin_in[b] && in_ring) secure_bzero(c->pin_in[b], in_ring < c->pin_in_bytes ? in_ring : c->pin_in_bytes);
      if (c->pin_out[b] && out_ring) secure_bzero(c->pin_out[b], out_ring < c->pin_out_bytes ? out_ring : c->pin_out_bytes);
    }
    hipStreamSynchronize(c->stream);
  }
  (void)n_staged;
  return rc;
}
// Short calls on pinned memory without copies (PipeSpec::small_direct_max): as long as the entry point's short-call kernel is the one that runs
static size_t small_direct_items(size_t kernel_max) { return kernel_max < 256 ? kernel_max : 256; }
template <typename Launch>
static int run_pipelined(bjj_ctx* c, size_t n, const PipeSpec& sp, Launch launch) {
  ENTER_DEVICE(c->device);
  { int rc = ensure_pipe(c, 0, 0, 0, 0); if (rc) return rc; }   // streams, the chunk schedule, the knobs
  // ---- which arrays are pinned (copied directly) and which go through the pinned rings
  bool in_direct[4], out_direct[4];
  u32 n_direct = 0, n_staged = 0;
  size_t per_item = 0;
  for (int i = 0; i < sp.n_in; i++) { in_direct[i] = !c->force_staged && host_range_pinned(sp.in[i], n * sp.in_stride[i]); (in_direct[i] ? n_direct : n_staged)++; per_item += sp.in_stride[i]; }
  for (int i = 0; i < sp.n_out; i++) { out_direct[i] = !c->force_staged && host_range_pinned(sp.out[i], n * sp.out_stride[i]); (out_direct[i] ? n_direct : n_staged)++; per_item += sp.out_stride[i]; }
  per_item += sp.extra_dev_per_item;
  // ---- super-batches: what fits into the device staging budget at once (2^20 verifications are 202 MB)
  size_t cap = c->pipe_budget / per_item;
  cap = cap > c->pipe_chunk ? cap / c->pipe_chunk * c->pipe_chunk : c->pipe_chunk;
  if (!cap) return set_err(BJJ_E_HIP, "host-pointer pipeline: not initialised (chunk size 0)");   // the loop below would never advance
  u32 chunks = 0;
  int rc = BJJ_OK;
  c->in_pipeline = true;
  for (size_t lo = 0; lo < n && !rc; lo += cap) {
    const size_t cnt = n - lo < cap ? n - lo : cap;
    PipeSpec sub = sp;
    for (int i = 0; i < sp.n_in; i++) sub.in[i] = sp.in[i] + lo * sp.in_stride[i];
    for (int i = 0; i < sp.n_out; i++) sub.out[i] = sp.out[i] + lo * sp.out_stride[i];
    rc = run_super_batch(c, cnt, sub, in_direct, out_direct, n_staged, launch, &chunks);
  }
  c->in_pipeline = false;
  c->k1_half_now = false;
  c->last_host_direct = n_direct; c->last_host_staged = n_staged; c->last_host_chunks = chunks;
  if (!sp.zero_copy_out && !sp.small_direct_max) c->last_host_zero_copy &= ~1u;
  if (!sp.zero_copy_in && !sp.small_direct_max) c->last_host_zero_copy &= ~2u;
  // the call has synchronised for the caller: a verify / variable-base workgroup that gave up waiting for a table slot makes
  // it an error here, not at some later bjj_sync.  Only the pipeline's OWN streams have been waited for: while a
  // device-pointer launch of the caller is still in flight on one of the sets, its workgroups are popping and pushing the rings --
  // reading them now could miss an error that comes later (and clear rings_used, so that bjj_sync would skip the check), and
  // rebuilding one would hand a slot out twice.  The check then stays pending for the next synchronising call.
  if (!rc) {
    bool busy = false;
    for (const ScratchSet& S : c->set)
      if (S.have_last && hipEventQuery(S.ev_last) == hipErrorNotReady) busy = true;
    (void)hipGetLastError();
    if (!busy) rc = ctx_check_slot_queues(c, "host-pointer call");
  }
  return rc;
}
static int ensure_codec(bjj_ctx* c, ScratchSet* S, size_t n) {  // 162 bytes per item of intermediate records
  if (n > S->codec_items) {
    if (S->codec) { HIPCK(hipDeviceSynchronize()); HIPCK(hipFree(S->codec)); S->codec = nullptr; S->codec_items = 0; }
    HIPCK(hipMalloc((void**)&S->codec, n * 162 + 64));
    S->codec_items = n;
  }
  return BJJ_OK;
}
static int ensure_xy(bjj_ctx* c, ScratchSet* S, size_t n) {     // 64 bytes per item: X, Y of phase 1 when the output slot is 32 bytes
  if (n > S->xy_items) {
    if (S->xy) { HIPCK(hipDeviceSynchronize()); HIPCK(hipFree(S->xy)); S->xy = nullptr; S->xy_items = 0; }
    HIPCK(hipMalloc((void**)&S->xy, n * 64));
    S->xy_items = n;
  }
  return BJJ_OK;
}
static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static void ctx_destroy(bjj_ctx* c) {
  DeviceGuard dg_(c->device);
  hipDeviceSynchronize();
  // key-derived material may sit in the codec scratch (scalar keys) and in the staging buffers: zero before release
  for (ScratchSet& S : c->set) {
    if (S.codec) { hipMemset(S.codec, 0, S.codec_items * 162 + 64); hipFree(S.codec); }
    if (S.scratch) hipFree(S.scratch);
    if (S.vb_tables) hipFree(S.vb_tables);
    if (S.slow) hipFree(S.slow);
    if (S.xy) hipFree(S.xy);
    if (S.ev_scan_in) hipEventDestroy(S.ev_scan_in);
    if (S.ev_scan_out) hipEventDestroy(S.ev_scan_out);
    if (S.scan_stream) hipStreamDestroy(S.scan_stream);
    if (S.ev_last) hipEventDestroy(S.ev_last);
  }
  for (StreamMark& k : c->marks) if (k.ev) hipEventDestroy(k.ev);
  if (c->table) hipFree(c->table);
  if (c->bases) hipFree(c->bases);
  if (c->ct_table) hipFree(c->ct_table);
  if (c->ct_bases) hipFree(c->ct_bases);
  delete c->pool;   // joins the copy workers
  c->pool = nullptr;
  for (int b = 0; b < BJJ_PIPE_BUFS; b++) {
    if (c->pin_in[b]) { secure_bzero(c->pin_in[b], c->pin_in_bytes); hipHostFree(c->pin_in[b]); }
    if (c->pin_out[b]) { secure_bzero(c->pin_out[b], c->pin_out_bytes); hipHostFree(c->pin_out[b]); }
  }
  if (c->dstage) { hipMemset(c->dstage, 0, c->pipe_bytes); hipFree(c->dstage); }
  for (hipEvent_t e : c->ev_in) hipEventDestroy(e);
  for (hipEvent_t e : c->ev_out) hipEventDestroy(e);
  if (c->ev_tail) hipEventDestroy(c->ev_tail);
  if (c->pipe_wl) hipFree(c->pipe_wl);
  for (hipEvent_t e : c->ev_k) hipEventDestroy(e);
  for (hipEvent_t e : c->ev_dec) hipEventDestroy(e);
  if (c->err_words) hipHostFree(c->err_words);
  if (c->vb_seen) hipHostFree(c->vb_seen);
  if (c->patch_host) hipHostFree(c->patch_host);
  if (c->slot_block) hipFree(c->slot_block);
  if (c->s_in) hipStreamDestroy(c->s_in);
  if (c->s_out) hipStreamDestroy(c->s_out);
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

extern "C" {
#pragma GCC visibility push(default)

const char* bjj_version(void) { return BJJ_VERSION_STRING; }
const char* bjj_last_error(void) { return g_err.c_str(); }

int bjj_init(int device, int window_bits, bjj_ctx** out_ctx) {
  if (!out_ctx) return set_err(BJJ_E_INVALID, "bjj_init: out_ctx is NULL");
  *out_ctx = nullptr;
  if (window_bits != 0 && window_bits != BJJ_WINDOW_AUTO && (window_bits < 4 || window_bits > BJJ_MAX_WINDOW_BITS))
    return set_err(BJJ_E_INVALID, "bjj_init: window_bits must be 0 (default, 23), BJJ_WINDOW_AUTO (-1) or 4..28");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return set_err(BJJ_E_NO_DEVICE, "bjj_init: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return set_err(BJJ_E_INVALID, "bjj_init: device index out of range");
  ENTER_DEVICE(device);
  const auto t0 = std::chrono::steady_clock::now();
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, device));
  bjj_ctx* c = new (std::nothrow) bjj_ctx();
  if (!c) return set_err(BJJ_E_NOMEM, "bjj_init: out of host memory");
  c->device = device;
  c->cus = prop.multiProcessorCount;
  const bool autow = window_bits == BJJ_WINDOW_AUTO;
  int W = window_bits == 0 ? BJJ_DEFAULT_WINDOW_BITS : window_bits;
  if (autow) {  // widest table that leaves 40 % of the free memory to the caller (a second context gets 26 bits)
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
  c->lanes_fixed = bjjk::fixed_base_lanes_per_cu(0);
  c->lanes_fixed_2x256 = bjjk::fixed_base_lanes_per_cu(1);
  if (const char* e = getenv("BJJ_VERIFY_DISPATCH")) {
    if (e[0] == '0' || e[0] == '1') c->verify_mode = e[0] - '0';
  }
  if (const char* e = getenv("BJJ_K2_VARIANT")) {
    if (e[0] == '0' || e[0] == '1') c->k2_variant = e[0] - '0';
  }
  if (const char* e = getenv("BJJ_FB_QUAD_MAX")) {  // tests / A-B
    char* q = nullptr; const unsigned long long v = strtoull(e, &q, 0);
    if (q != e && v <= ((unsigned long long)1 << 20)) c->fb_quad_max = (size_t)v;
  }
  if (const char* e = getenv("BJJ_SIGN_SMALL_MAX")) {  // tests / A-B
    char* q = nullptr; const unsigned long long v = strtoull(e, &q, 0);
    if (q != e && v <= ((unsigned long long)1 << 20)) c->sign_small_max = (size_t)v;
  }
  if (const char* e = getenv("BJJ_VERIFY_SMALL_MAX")) {  // tests / A-B
    char* q = nullptr; const unsigned long long v = strtoull(e, &q, 0);
    if (q != e && v <= ((unsigned long long)1 << 20)) c->verify_small_max = (size_t)v;
  }
  if (const char* e = getenv("BJJ_P5_COOP_MAX")) {  // tests / A-B
    char* q = nullptr; const unsigned long long v = strtoull(e, &q, 0);
    if (q != e && v <= ((unsigned long long)1 << 20)) c->p5_coop_max = (size_t)v;
  }
  if (const char* e = getenv("BJJ_VB_QUAD_MAX")) {  // tests / A-B: variable-base calls of at most this many items run four lanes per item (0 = never)
    char* q = nullptr; const unsigned long long v = strtoull(e, &q, 0);
    if (q != e && v <= ((unsigned long long)1 << 20)) c->vb_quad_max = (size_t)v;
  }
  if (const char* e = getenv("BJJ_K1_VARIANT")) {   // tests / A-B: force one shape of the fixed-base kernel
    if (e[0] == '0' || e[0] == '1') c->k1_variant = e[0] - '0';
  }
  if (const char* e = getenv("BJJ_VB_SPLIT")) {     // tests / A-B: K6 always behind K2 (0) or always beside it, behind a scan (1)
    if (e[0] == '0' || e[0] == '1') c->vb_split = e[0] - '0';
  }
  c->occ_vb_scan = bjjk::occ_var_base_scan();
  c->lanes_var = bjjk::var_base_lanes_per_cu();
  c->occ_poseidon = bjjk::occ_poseidon5();
  c->occ_verify = bjjk::occ_verify();
  c->occ_scan = bjjk::occ_verify_scan();
  c->occ_add = bjjk::occ_point_add();
  c->occ_decomp = bjjk::occ_decompress();
  c->occ_sign = bjjk::occ_sign();
  c->occ_sign_schnorr = bjjk::occ_sign_schnorr();
  hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (se == hipSuccess) se = hipHostMalloc((void**)&c->vb_seen, (BJJ_SCRATCH_SETS + 1) * sizeof(u32), hipHostMallocDefault);
  if (se != hipSuccess) { (void)hipGetLastError(); ctx_destroy(c); return set_err(BJJ_E_HIP, std::string("bjj_init: stream/event: ") + hipGetErrorString(se)); }
  memset(c->vb_seen, 0, (BJJ_SCRATCH_SETS + 1) * sizeof(u32));
  { const int rc_ = ensure_pipe_streams(c); if (rc_) { ctx_destroy(c); return rc_; } }   // the hardware-queue plan of the context, fixed now
  {  // XCDs of this device (the verify kernels keep one queue of table slots per XCD)
    u32* d_word = nullptr;
    if (hipMalloc((void**)&d_word, sizeof(u32)) == hipSuccess) {
      const int x = bjjk::probe_xccs(c->stream, d_word);
      hipFree(d_word);
      c->xccs = x >= 1 && x <= 16 ? x : 1;
    }
  }
  for (;;) {
    c->table_bytes = fixed_stride(c->W) * (size_t)c->nwin * NIELS_WORDS * sizeof(u32);
    se = hipMalloc((void**)&c->table, c->table_bytes);
    if (se == hipSuccess || !autow) break;
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
    (void)hipGetLastError();
    c->table_bytes = 0;
    ctx_destroy(c);
    return set_err(BJJ_E_NOMEM, "bjj_init: cannot allocate the fixed-base table");
  }
  se = bjjk::build_fixed_table(c->stream, c->table, c->bases, c->W, c->nwin);
  if (se == hipSuccess) se = hipStreamSynchronize(c->stream);
  if (se != hipSuccess) {
    ctx_destroy(c);
    return set_err(BJJ_E_HIP, std::string("bjj_init: table build failed: ") + hipGetErrorString(se));
  }
  c->init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *out_ctx = c;
  return BJJ_OK;
}

void bjj_free(bjj_ctx* c) {
  if (!c) return;
  ctx_destroy(c);
}

int bjj_sync(bjj_ctx* c) {
  if (!c) return set_err(BJJ_E_INVALID, "bjj_sync: ctx is NULL");
  ENTER_DEVICE(c->device);
  for (ScratchSet& S : c->set)      // work enqueued on a caller's stream
    if (S.have_last && S.last_stream != c->stream) HIPCK(hipEventSynchronize(S.ev_last));
  for (StreamMark& k : c->marks)
    if (k.used) HIPCK(hipEventSynchronize(k.ev));
  HIPCK(hipStreamSynchronize(c->stream));
  // everything the context enqueued has run: a slot-queue pop that gave up waiting (slot_queue.hpp) is an ERROR of the launches
  // just completed -- reported here instead of a hung GPU; the rings are rebuilt so that the context stays usable
  return ctx_check_slot_queues(c, "bjj_sync");
}
void* bjj_stream(bjj_ctx* c) { return c ? (void*)c->stream : nullptr; }

// Sizes the first `sets` scratch sets for n items (the codec records only on request).  bjj_reserve is the public form: set 0
// plus whatever a second stream has already brought into use.  The multi-GPU pipeline calls this with sets = 2 before it
// enqueues anything when a peer's block travels in more than one piece -- pieces alternate over two streams, and the second
// set must not be allocated (hipMalloc, a synchronous copy of the slot queues) in the middle of the pipeline.
static int reserve_sets(bjj_ctx* c, size_t n, int sets, bool with_codec) {
  ENTER_DEVICE(c->device);
  int k = 0;
  for (ScratchSet& S : c->set) {
    const bool wanted = k < sets || S.have_last;
    k++;
    if (!wanted) continue;
    int rc = ensure_scratch(c, &S, n ? n : 1); if (rc) return rc;
    if (with_codec) { rc = ensure_codec(c, &S, n ? n : 1); if (rc) return rc; }
  }
  return BJJ_OK;
}
int bjj_reserve(bjj_ctx* c, size_t n) {
  if (!c) return set_err(BJJ_E_INVALID, "bjj_reserve: ctx is NULL");
  return reserve_sets(c, n, 1, true);
}
// ---- pinned host memory for the host-pointer entry points (include/bjj_hip.h) ----------------------------------------------
int bjj_host_alloc(bjj_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return set_err(BJJ_E_INVALID, "bjj_host_alloc: NULL argument");
  *out = nullptr;
  if (!bytes) return set_err(BJJ_E_INVALID, "bjj_host_alloc: zero bytes");
  ENTER_DEVICE(c->device);
  void* p = nullptr;
  const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
  if (e != hipSuccess) { (void)hipGetLastError(); return set_err(BJJ_E_NOMEM, std::string("bjj_host_alloc: ") + hipGetErrorString(e)); }
  try {
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_ranges.push_back({(uintptr_t)p, (uintptr_t)p + bytes, true});
  } catch (...) { hipHostFree(p); return set_err(BJJ_E_NOMEM, "bjj_host_alloc: out of host memory"); }
  *out = p;
  return BJJ_OK;
}
static int host_range_drop(void* p, bool owned, const char* who) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  for (size_t i = 0; i < g_host_ranges.size(); i++)
    if (g_host_ranges[i].lo == (uintptr_t)p && g_host_ranges[i].owned == owned) {
      g_host_ranges.erase(g_host_ranges.begin() + (long)i);
      return BJJ_OK;
    }
  return set_err(BJJ_E_INVALID, std::string(who) + ": not a pointer this library " + (owned ? "allocated" : "registered"));
}
int bjj_host_free(bjj_ctx* c, void* p) {
  if (!c) return set_err(BJJ_E_INVALID, "bjj_host_free: ctx is NULL");
  if (!p) return BJJ_OK;
  { int rc = host_range_drop(p, true, "bjj_host_free"); if (rc) return rc; }
  ENTER_DEVICE(c->device);
  HIPCK(hipHostFree(p));
  return BJJ_OK;
}
int bjj_host_register(bjj_ctx* c, void* p, size_t bytes) {
  if (!c || !p || !bytes) return set_err(BJJ_E_INVALID, "bjj_host_register: NULL argument or zero bytes");
  ENTER_DEVICE(c->device);
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
  if (e != hipSuccess) { (void)hipGetLastError(); return set_err(BJJ_E_HIP, std::string("bjj_host_register: ") + hipGetErrorString(e)); }
  try {
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_ranges.push_back({(uintptr_t)p, (uintptr_t)p + bytes, false});
  } catch (...) { hipHostUnregister(p); return set_err(BJJ_E_NOMEM, "bjj_host_register: out of host memory"); }
  return BJJ_OK;
}
int bjj_host_unregister(bjj_ctx* c, void* p) {
  if (!c) return set_err(BJJ_E_INVALID, "bjj_host_unregister: ctx is NULL");
  if (!p) return BJJ_OK;
  { int rc = host_range_drop(p, false, "bjj_host_unregister"); if (rc) return rc; }
  ENTER_DEVICE(c->device);
  HIPCK(hipHostUnregister(p));
  return BJJ_OK;
}
int bjj_host_is_pinned(bjj_ctx* c, const void* p, size_t bytes) {
  if (!c || !p) return set_err(BJJ_E_INVALID, "bjj_host_is_pinned: NULL argument");
  ENTER_DEVICE(c->device);
  return host_range_pinned(p, bytes) ? 1 : 0;
}

int bjj_get_info(bjj_ctx* c, bjj_info* out) {
  if (!c || !out) return set_err(BJJ_E_INVALID, "bjj_get_info: NULL argument");
  const size_t cap = out->struct_size;
  if (cap < 8) return set_err(BJJ_E_INVALID, "bjj_get_info: set info.struct_size = sizeof(bjj_info) before the call");
  bjj_info full;
  memset(&full, 0, sizeof(full));
  bjj_info* info = &full;
  info->device = c->device;
  info->compute_units = c->cus;
  info->window_bits = c->W;
  info->n_windows = c->nwin;
  info->table_bytes = c->table_bytes;
  info->scratch_bytes = (uint64_t)c->pipe_bytes;
  for (const ScratchSet& S : c->set)
    info->scratch_bytes += S.scratch_items * 64 + S.vb_threads * VB_TABLE_WORDS_MAX * sizeof(u32) + S.slow_items * 4 +
                           (S.codec_items ? S.codec_items * 162 + 64 : 0) + S.xy_items * 64;
  info->kernel_fixed_base = "bjj_k_mul_fixed_base";
  info->kernel_var_base = "bjj_k_mul_var_base_tiles";   // the form a launch that runs alone gets (k_var.hip)
  info->kernel_poseidon5 = "bjj_k_poseidon5";
  info->kernel_verify = "bjj_k_eddsa_verify_groups";
  info->init_ms = c->init_ms;
  info->signer_constant_time = c->ct_signer ? 1 : 0;
  info->last_fixed_base_shape = c->last_k1;
  info->last_var_base_form = c->last_k2;
  info->last_var_base_split = c->last_vb_split;
  info->last_host_zero_copy = c->last_host_zero_copy;
  info->last_poseidon_form = c->last_p5;
  info->last_sign_form = c->last_sign;
  info->last_verify_dispatch = c->last_verify_mode;
  info->last_host_direct_arrays = c->last_host_direct;
  info->last_host_staged_arrays = c->last_host_staged;
  info->last_host_chunks = c->last_host_chunks;
  info->host_copy_threads = c->pool ? (int)c->pool->th.size() : 0;
  info->kernel_fixed_base_overlap = "bjj_k_mul_fixed_base_2x256";   // the forms overlapping launches get (expect_overlap)
  info->kernel_var_base_overlap = "bjj_k_mul_var_base";
  const size_t fill = cap < sizeof(full) ? cap : sizeof(full);   // never past the caller's struct
  full.struct_size = (uint32_t)fill;
  memcpy(out, &full, fill);
  return BJJ_OK;
}

int bjj_check_table(bjj_ctx* c, uint64_t* n_bad) {
  if (!c || !n_bad) return set_err(BJJ_E_INVALID, "bjj_check_table: NULL argument");
  ENTER_DEVICE(c->device);
  unsigned long long* d_bad = nullptr;
  HIPCK(hipMalloc((void**)&d_bad, sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), c->stream);
  if (e == hipSuccess) e = bjjk::check_fixed_table(c->stream, c->cus * 8, c->table, c->bases, c->W, c->nwin, d_bad);
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
// Every *_dev entry selects the context's device (a second context on another GPU of the same process must not launch
// on a foreign device).  Entries that use scratch (SET_ENTER / SET_LEAVE) pick a scratch set for their stream, size it,
// order themselves behind the set's previous user and record; the others (DEV_ENTER / DEV_LEAVE) only leave a completion
// mark for bjj_sync.
#define DEV_ENTER(c, stream)                                              \
  hipStream_t st = (stream) ? (hipStream_t)(stream) : (c)->stream;        \
  ENTER_DEVICE((c)->device)
#define DEV_LEAVE(c) return mark_stream((c), st)
#define SET_ENTER(c, stream, n, with_codec)                               \
  hipStream_t st = (stream) ? (hipStream_t)(stream) : (c)->stream;        \
  ENTER_DEVICE((c)->device);                                              \
  ScratchSet* S = pick_set((c), st);                                      \
  { int rc_ = ensure_scratch((c), S, (n)); if (rc_) return rc_;           \
    if (with_codec) { rc_ = ensure_codec((c), S, (n)); if (rc_) return rc_; } \
    rc_ = set_enter((c), S, st); if (rc_) return rc_; }
#define SET_LEAVE(c) return set_leave((c), S, st)
#define LAUNCHCK_S(expr, who)                                                                       \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return set_err(BJJ_E_HIP, std::string(who) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define LAUNCHCK(expr, name)                                                                        \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return set_err(BJJ_E_HIP, std::string(name ": ") + hipGetErrorString(e_)); \
  } while (0)

static int ensure_scan_stream(ScratchSet* S) {
  if (!S->scan_stream) {
    int least = 0, greatest = 0;
    HIPCK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    static const bool normal = [] { const char* e = getenv("BJJ_SCAN_STREAM_PRIORITY"); return e && e[0] == 'n'; }();   // developer A/B
    HIPCK(hipStreamCreateWithPriority(&S->scan_stream, hipStreamNonBlocking, normal ? 0 : greatest));
    HIPCK(hipEventCreateWithFlags(&S->ev_scan_in, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&S->ev_scan_out, hipEventDisableTiming));
  }
  return BJJ_OK;
}
// compressed: d_out holds 32-byte Point::compress records (src/lib.rs:166-178), the compression fused into K1's epilogue
static int fixed_base_launch(bjj_ctx* c, const void* d_scalars, size_t n, void* d_out, void* stream, bool compressed, const char* who) {
  if (!c) return set_err(BJJ_E_INVALID, std::string(who) + ": ctx is NULL");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  if (!d_scalars || !d_out || !aligned16(d_scalars) || !aligned16(d_out))
    return set_err(BJJ_E_INVALID, std::string(who) + ": NULL or not 16-byte aligned device pointer");
  if (n <= c->fb_quad_max && c->k1_variant < 0) {   // short calls (a single B8.mul_scalar is one): four lanes per item, no scratch (k_small.hip)
    DEV_ENTER(c, stream);
    c->last_k1 = 2;
    LAUNCHCK_S(bjjk::mul_fixed_base_quad(st, c->table, c->W, c->nwin, (const uint8_t*)d_scalars, n, (uint8_t*)d_out, compressed), who);
    DEV_LEAVE(c);
  }
  SET_ENTER(c, stream, n, false);
  if (compressed) { int rc_ = ensure_xy(c, S, n); if (rc_) return rc_; }
  const int kv = fixed_base_variant(c, S);
  LAUNCHCK(bjjk::mul_fixed_base(st, c->cus, fixed_base_lanes(c, kv), kv, c->table, c->W, c->nwin, (const uint8_t*)d_scalars, n,
                                (uint8_t*)d_out, S->scratch, compressed ? S->xy : nullptr), "bjj_mul_fixed_base_dev");
  SET_LEAVE(c);
}
int bjj_mul_fixed_base_dev(bjj_ctx* c, const void* d_scalars, size_t n, void* d_out, void* stream) {
  return fixed_base_launch(c, d_scalars, n, d_out, stream, false, "bjj_mul_fixed_base_dev");
}
int bjj_mul_fixed_base_compressed_dev(bjj_ctx* c, const void* d_scalars, size_t n, void* d_out32, void* stream) {
  return fixed_base_launch(c, d_scalars, n, d_out32, stream, true, "bjj_mul_fixed_base_compressed_dev");
}
// Off-curve points (k_var.hip): which form does this call take?  Forced by BJJ_VB_SPLIT, else by what K6 found in the calls that have
// COMPLETED so far (K6 writes its item count into the set's pinned word; a call that is still in flight has not reported yet, so a
// caller that enqueues many launches without synchronising gets one form for all of them).
static bool vb_want_split(const bjj_ctx* c) {
  if (c->vb_split >= 0) return c->vb_split == 1;
  u32 any = 0;
  for (int k = 0; k <= BJJ_SCRATCH_SETS; k++) any |= __atomic_load_n(&c->vb_seen[k], __ATOMIC_RELAXED);
  return any != 0;
}
static int var_base_check(bjj_ctx* c, const void* d_pts, const void* d_scalars, size_t scalar_bytes, size_t n, void* d_out, const char* who) {
  if (!c) return set_err(BJJ_E_INVALID, std::string(who) + ": ctx is NULL");
  if (scalar_bytes == 0 || (scalar_bytes & 31) || scalar_bytes > BJJ_MAX_SCALAR_BYTES)
    return set_err(BJJ_E_INVALID, std::string(who) + ": scalar_bytes must be a multiple of 32 in 32..BJJ_MAX_SCALAR_BYTES");
  if (n >> 32) return set_err(BJJ_E_INVALID, "batches are limited to 2^32 - 1 items per call");
  if (n && (!d_pts || !d_scalars || !d_out || !aligned16(d_pts) || !aligned16(d_scalars) || !aligned16(d_out)))
    return set_err(BJJ_E_INVALID, std::string(who) + ": NULL or not 16-byte aligned device pointer");
  return BJJ_OK;
}
// k_var.hip: the two forms of K2 -- tiles for a launch that runs alone, grid-strided for overlapping launches (expect_overlap).
// A large launch queues behind the other sets and runs alone (tiles).
static int var_base_form(bjj_ctx* c, ScratchSet* S, hipStream_t st, size_t n, int* kv_out) {
  bool overlap = c->k2_variant < 0 && expect_overlap(c, S);
  if (overlap && n > BJJ_LARGE_LAUNCH) { int rc_ = wait_for_other_sets(c, S, st); if (rc_) return rc_; overlap = false; }
  const int kv = c->k2_variant >= 0 ? c->k2_variant : (overlap ? 0 : 1);
  c->last_k2 = kv;
  if (kv == 1) c->rings_used = true;      // the tiles take their table scratch from the slot queues
  *kv_out = kv;
  static const bool trace = [] { const char* e = getenv("BJJ_PIPE_TRACE"); return e && e[0] == '1'; }();
  if (trace) fprintf(stderr, "[k2] %zu items on set %d: %s\n", n, (int)(S - c->set), kv ? "tiles" : "grid-strided");
  return BJJ_OK;
}
// One device-pointer call.  Clean batches (every point on the curve -- what the history says): K2 makes the list of the items it
// skips, K6 follows on the caller's stream and finds it empty: nothing extra.  Once a completed call has met an off-curve point:
//   caller's stream:   ------------------------- K2 (skips them, no list) ------------------------ wait --
//   priority stream:   wait -- scan (the list) -- K6 (its ~4.5 ms serial chains, a few waves) ---- record
// so that the malformed items cost their share of the chip instead of a launch-long tail: one off-curve point in 4 096 was +30 % on
// a 2^20-item launch with K6 behind K2 (profiles/r05_var_base_offcurve.txt); beside it, see profiles/r06_var_base_offcurve.txt.
static int var_base_launch(bjj_ctx* c, const void* d_pts, const void* d_scalars, size_t scalar_bytes, size_t n, void* d_out,
                           void* stream, const char* who) {
  { int rc_ = var_base_check(c, d_pts, d_scalars, scalar_bytes, n, d_out, who); if (rc_) return rc_; }
  if (n == 0) return BJJ_OK;
  SET_ENTER(c, stream, n, false);
  // Short calls (a single Point::mul_scalar is one): four lanes per item (k_small.hip) -- 0.50 ms per call up to 2^12 items, 0.56 ms at 2^14, against
  // K2's 1.15-1.22 ms, which is one lane's serial chain whatever the call's size; from 2^15 items on K2 is ahead (1.27 vs 1.38 ms;
  // profiles/r06_small_calls.txt).  32-byte scalars; off-curve points go to K6 as from K2.
  const bool quad = scalar_bytes == 32 && n <= c->vb_quad_max && c->k2_variant < 0;   // (BJJ_K2_VARIANT forces K2, in that form)
  int kv = 2;
  if (quad) c->last_k2 = 2;
  else { int rc_ = var_base_form(c, S, st, n, &kv); if (rc_) return rc_; }
  const uint8_t* pts = (const uint8_t*)d_pts;
  const uint8_t* sc = (const uint8_t*)d_scalars;
  const int sc_words = (int)(scalar_bytes / 4);
  u32* seen = &c->vb_seen[S - c->set];
  const bool split = vb_want_split(c);
  c->last_vb_split = split ? 1 : 0;
  if (split) {
    { int rc_ = ensure_scan_stream(S); if (rc_) return rc_; }
    hipStream_t xs = S->scan_stream;   // (the copy streams or the second lane instead: no difference, profiles/r06_var_base_beside_ab.txt)
    HIPCK(hipEventRecord(S->ev_scan_in, st));
    HIPCK(hipStreamWaitEvent(xs, S->ev_scan_in, 0));
    LAUNCHCK(bjjk::var_base_list_reset(xs, S->slow), "variable-base list");
    LAUNCHCK(bjjk::var_base_scan(xs, grid_for(c, n, c->occ_vb_scan, 64), pts, 0, n, S->slow), "variable-base scan");
    LAUNCHCK(bjjk::mul_var_base_exact(xs, c->cus * 4, pts, sc, sc_words, (uint8_t*)d_out, S->slow, nullptr, seen), "variable-base (exact)");
    HIPCK(hipEventRecord(S->ev_scan_out, xs));
  }
  if (quad) {
    if (!split) HIPCK(hipMemsetAsync(S->slow, 0, 8 * sizeof(u32), st));
    LAUNCHCK_S(bjjk::mul_var_base_quad(st, pts, sc, n, (uint8_t*)d_out, split ? nullptr : S->slow), who);
  } else
  LAUNCHCK_S(bjjk::mul_var_base_main(st, c->cus, c->lanes_var, kv, pts, sc, sc_words, n, (uint8_t*)d_out, S->scratch, S->vb_tables, split ? nullptr : S->slow,
                                     S->slotq2, S->slot_cap2 | ((u32)c->xccs << 16)), who);
  if (split) HIPCK(hipStreamWaitEvent(st, S->ev_scan_out, 0));
  else LAUNCHCK(bjjk::mul_var_base_exact(st, c->cus * 4, pts, sc, sc_words, (uint8_t*)d_out, S->slow, nullptr, seen), "variable-base (exact)");
  SET_LEAVE(c);
}
// ---- variable base through the host-pointer pipeline: the cure verify got in round 5 (VerifyPipe below) -----------------------
//   per chunk, behind its H2D:  the on-curve scan of its points on the priority stream, appending batch-wide indices to ONE list;
//                               K2 over its items on its lane -- K2 skips an off-curve item by itself and never writes its slot
//   behind the last scan:       ONE K6 launch over the list on the priority stream, results COMPACT beside the list (d_extra): a
//                               chunk's outputs leave the device when its K2 is done, long before K6 is
//   when everything has landed: the host lays the few exact results over their slots in the caller's array
// Before: K6 behind K2 in EVERY chunk's lane -- one off-curve point in 4 096 cost 2^20 items on pinned memory +71 % (15.9 ->
// 27.3 ms, profiles/r05_var_base_offcurve.txt).
static int var_base_bulk_launch(bjj_ctx* c, const void* d_pts, const void* d_scalars, size_t scalar_bytes, size_t n, void* d_out, uint8_t* xy, void* stream) {
  SET_ENTER(c, stream, n, false);
  // one tile per workgroup, whatever else is in flight: the chunks' launches are then work-conserving among themselves -- a
  // workgroup retires after its tile and the slot goes to whichever launch has tiles pending -- where the grid-strided form's
  // resident sets (the better form for two whole 2^20-item launches that overlap) made the call's length depend on how the chunks'
  // partly filled rounds happened to interleave (15.6 ... 18.0 ms for the same 2^20 items, profiles/r06_var_base_host_schedule.txt)
  const int kv = c->k2_variant >= 0 ? c->k2_variant : 1;
  c->last_k2 = kv;
  if (kv == 1) c->rings_used = true;
  c->last_vb_split = 1;
  LAUNCHCK(bjjk::mul_var_base_main(st, c->cus, c->lanes_var, kv, (const uint8_t*)d_pts, (const uint8_t*)d_scalars, (int)(scalar_bytes / 4), n, (uint8_t*)d_out,
                                   S->scratch, S->vb_tables, nullptr, S->slotq2, S->slot_cap2 | ((u32)c->xccs << 16), xy), "variable base (bulk)");
  SET_LEAVE(c);
}
struct VarBasePipe : PipeExtra {
  bjj_ctx* c;
  int sc_words;
  const uint8_t *pts = nullptr, *scalars = nullptr;
  uint8_t* out = nullptr;      // zero_copy: the device mapping of the caller's pinned output array (K6 writes its items' slots itself)
  size_t n = 0;
  VarBasePipe(bjj_ctx* c_, size_t scalar_bytes) : c(c_), sc_words((int)(scalar_bytes / 4)) {}
  int begin(size_t n_, void** d_in, void** d_out) override {
    out = (uint8_t*)d_out[0];
    n = n_;
    pts = (const uint8_t*)d_in[0]; scalars = (const uint8_t*)d_in[1];
    ScratchSet* S = &c->set[0];
    { int rc_ = ensure_scan_stream(S); if (rc_) return rc_; }
    if (n > c->pipe_wl_items) {
      if (c->pipe_wl) { HIPCK(hipDeviceSynchronize()); HIPCK(hipFree(c->pipe_wl)); c->pipe_wl = nullptr; c->pipe_wl_items = 0; }
      HIPCK(hipMalloc((void**)&c->pipe_wl, (n + 16) * sizeof(u32)));
      c->pipe_wl_items = n;
    }
    LAUNCHCK(bjjk::var_base_list_reset(S->scan_stream, c->pipe_wl), "variable-base list");
    return BJJ_OK;
  }
  int chunk_arrived(size_t lo, size_t cnt, hipEvent_t arrived) override {
    ScratchSet* S = &c->set[0];
    HIPCK(hipStreamWaitEvent(S->scan_stream, arrived, 0));
    LAUNCHCK(bjjk::var_base_scan(S->scan_stream, grid_for(c, cnt, c->occ_vb_scan, 64), pts, lo, lo + cnt, c->pipe_wl), "variable-base scan");
    return BJJ_OK;
  }
  int all_arrived(hipEvent_t ev_tail) override {
    ScratchSet* S = &c->set[0];
    // zero-copy: d_extra is K2's stash and K6 stores straight into the caller's array; else d_extra takes K6's results compactly
    LAUNCHCK(bjjk::mul_var_base_exact(S->scan_stream, c->cus * 4, pts, scalars, sc_words, zero_copy ? out : nullptr, c->pipe_wl,
                                      zero_copy ? nullptr : (uint8_t*)d_extra, &c->vb_seen[BJJ_SCRATCH_SETS]), "variable-base (exact)");
    HIPCK(hipEventRecord(ev_tail, S->scan_stream));
    return BJJ_OK;
  }
  int finish(uint8_t* const* host_out, size_t) override {
    const size_t cnt = __atomic_load_n(&c->vb_seen[BJJ_SCRATCH_SETS], __ATOMIC_ACQUIRE);   // K6 is done (ev_tail): its count is in
    if (!cnt || zero_copy) return BJJ_OK;
    if (cnt > n) return set_err(BJJ_E_HIP, "variable base: the exact list is longer than the batch");
    const size_t need = cnt * 68;
    if (need > c->patch_host_bytes) {
      if (c->patch_host) { HIPCK(hipHostFree(c->patch_host)); c->patch_host = nullptr; c->patch_host_bytes = 0; }
      const size_t want = need < ((size_t)1 << 16) ? (size_t)1 << 16 : need + need / 2;
      HIPCK(hipHostMalloc((void**)&c->patch_host, want, hipHostMallocDefault));
      c->patch_host_bytes = want;
    }
    HIPCK(hipMemcpyAsync(c->patch_host, d_extra, cnt * 64, hipMemcpyDeviceToHost, c->s_out));
    HIPCK(hipMemcpyAsync(c->patch_host + cnt * 64, c->pipe_wl + 8, cnt * 4, hipMemcpyDeviceToHost, c->s_out));
    HIPCK(hipStreamSynchronize(c->s_out));
    const u32* idx = (const u32*)(c->patch_host + cnt * 64);
    for (size_t j = 0; j < cnt; j++) {
      if (idx[j] >= n) return set_err(BJJ_E_HIP, "variable base: index out of range on the exact list");
      memcpy(host_out[0] + (size_t)idx[j] * 64, c->patch_host + j * 64, 64);
    }
    return BJJ_OK;
  }
};
static int var_base_host(bjj_ctx* c, const uint8_t* pts, const uint8_t* scalars, size_t scalar_bytes, size_t n, uint8_t* out) {
  PipeSpec sp = {2, 1, {pts, scalars}, {64, scalar_bytes}, {out}, {64}, false};
  if (scalar_bytes == 32 && c->k2_variant < 0) sp.small_direct_max = small_direct_items(c->vb_quad_max);
  sp.first_chunk = (size_t)1 << 16;   // 14 ms of kernels over 3 ms of copies: a 2^15-item launch holds its lane for a whole round with a quarter of the chip
  sp.max_chunk = (size_t)1 << 18;
  static const bool per_chunk = [] { const char* e = getenv("BJJ_PIPE_VAR_BASE_SPLIT"); return e && e[0] == '0'; }();   // developer: the round-5 form
  { ENTER_DEVICE(c->device); int rc_ = ensure_pipe(c, 0, 0, 0, 0); if (rc_) return rc_; }
  const size_t first = c->pipe_env_schedule ? c->pipe_first : sp.first_chunk;
  // a call of ONE chunk is a device-pointer launch with copies around it (a single Point::mul_scalar, src/lib.rs:149, is such a call)
  if (per_chunk || c->vb_split == 0 || n < first + first / 2)
    return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return var_base_launch(c, i[0], i[1], scalar_bytes, cnt, o[0], st, "bjj_mul_var_base"); });
  VarBasePipe vp(c, scalar_bytes);
  sp.extra = &vp;
  sp.extra_dev_per_item = 64;         // K6's compact results (as many as there are items, at worst) -- or, zero-copy, K2's phase-1 stash
  sp.last_on_priority_lane = true;
  // Pinned output array: the kernels store into it themselves (64 MB of results over the 15 ms the kernels take anyway) -- no copy-out
  // stage, so nothing favours small chunks once the inputs are in: the chunks double without a cap (2^16, 2^17, 2^18, the rest), and
  // the call ends when the last tile does, like one device-pointer launch.  Pageable output: copies chunk by chunk as before, the last
  // chunk small (its copy-out is the one nothing hides).
  if (c->pipe_zero_copy && c->k2_variant != 0 && !c->force_staged && host_range_pinned(out, n * 64)) { sp.zero_copy_out = true; sp.max_chunk = (size_t)1 << 24; }   // (the stash apart exists for the tiles)
  else { sp.tail_chunk = (size_t)1 << 16; sp.max_chunk = (size_t)1 << 18; }
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) {
    const size_t lo = (size_t)((const uint8_t*)i[0] - vp.pts) / 64;           // this chunk's first item within the super-batch
    return var_base_bulk_launch(c, i[0], i[1], scalar_bytes, cnt, o[0], vp.zero_copy ? (uint8_t*)vp.d_extra + lo * 64 : nullptr, st); });
}
int bjj_mul_var_base_dev(bjj_ctx* c, const void* d_pts, const void* d_scalars, size_t n, void* d_out, void* stream) {
  return var_base_launch(c, d_pts, d_scalars, 32, n, d_out, stream, "bjj_mul_var_base_dev");
}
int bjj_mul_var_base_wide_dev(bjj_ctx* c, const void* d_pts, const void* d_scalars, size_t scalar_bytes, size_t n, void* d_out,
                              void* stream) {
  return var_base_launch(c, d_pts, d_scalars, scalar_bytes, n, d_out, stream, "bjj_mul_var_base_wide_dev");
}
int bjj_poseidon5_dev(bjj_ctx* c, const void* d_in, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_poseidon5_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_in, "bjj_poseidon5_dev"); CHECK_PTR(d_out, "bjj_poseidon5_dev");
  DEV_ENTER(c, stream);
  // Short calls (a single POSEIDON.hash is one): six lanes per hash (k_small.hip) -- the permutation's dependent chain instead of all of
  // its ~1 000 multiplications in a row on one lane (profiles/r06_small_calls.txt)
  c->last_p5 = n <= c->p5_coop_max ? 1 : 0;
  if (c->last_p5) LAUNCHCK(bjjk::poseidon5_coop(st, (const uint8_t*)d_in, n, (uint8_t*)d_out), "bjj_poseidon5_dev");
  else
  LAUNCHCK(bjjk::poseidon5(st, grid_for(c, n, c->occ_poseidon), (const uint8_t*)d_in, n, (uint8_t*)d_out), "bjj_poseidon5_dev");
  DEV_LEAVE(c);
}
// scan (priority stream) -> main kernel (the caller's stream), both ordered behind what `st` has queued so far
static int enqueue_verify(bjj_ctx* c, ScratchSet* S, hipStream_t st, bool schnorr, const uint8_t* pk, const uint8_t* r, const uint8_t* s,
                          const uint8_t* msg, size_t n, uint8_t* ok) {
  { int rc_ = ensure_scan_stream(S); if (rc_) return rc_; }
  const int scan_grid = grid_for(c, n, c->occ_scan, 64) * 64 / bjjk::verify_scan_block();   // occ_scan counts waves
  // Short calls (a single `verify` is one): the bulk with eight lanes per signature (k_small.hip) between K4's scan and K4's exact
  // launch -- the hash on six lanes, the curve arithmetic on four, instead of one lane's serial chain (profiles/r06_small_calls.txt)
  if (n <= c->verify_small_max && c->verify_mode < 0) {
    c->last_verify_mode = 2;
    c->rings_used = true;               // the exact launch takes its table scratch from the slot queues
    LAUNCHCK(bjjk::verify_scan(st, scan_grid, pk, r, msg, n, S->slow), "verify scan");
    LAUNCHCK(bjjk::verify_small(st, schnorr, c->table, c->W, c->nwin, pk, r, s, msg, n, ok), "verify (short call)");
    LAUNCHCK(bjjk::verify_main(st, 1, 0, schnorr, c->table, c->W, c->nwin, pk, r, s, msg, n, ok, S->vb_tables, S->slow, S->slotq,
                               S->slot_cap | ((u32)c->xccs << 16), bjjk::VERIFY_EXACT), "verify (exact)");
    return BJJ_OK;
  }
  bool busy = expect_overlap(c, S);
  if (busy && c->verify_mode < 0 && n > BJJ_LARGE_LAUNCH) { int rc_ = wait_for_other_sets(c, S, st); if (rc_) return rc_; busy = false; }
  // k_verify.hip: persistent waves for ONE large launch that runs alone, one group per workgroup otherwise
  const int mode = c->verify_mode >= 0 ? c->verify_mode : ((!busy && n > BJJ_LARGE_LAUNCH) ? 0 : 1);
  c->last_verify_mode = mode;
  if (mode == 1) c->rings_used = true;    // one group per workgroup: table scratch from the slot queues
  // inside the host-pointer pipeline the scan stays in line: its lanes, copy streams and the scan streams would be five
  // co-active high-priority streams on four hardware queues, and a scan that lands behind the copy-out stream's wait for the
  // previous chunk's kernels would serialise the lanes
  if (busy && !(c->in_pipeline && c->pipe_scan_inline)) {   // the chip is (about to be) full of another launch's workgroups: priority stream
    HIPCK(hipEventRecord(S->ev_scan_in, st));
    HIPCK(hipStreamWaitEvent(S->scan_stream, S->ev_scan_in, 0));
    LAUNCHCK(bjjk::verify_scan(S->scan_stream, scan_grid, pk, r, msg, n, S->slow), "verify scan");
    HIPCK(hipEventRecord(S->ev_scan_out, S->scan_stream));
    HIPCK(hipStreamWaitEvent(st, S->ev_scan_out, 0));
  } else {                              // nothing to compete with: in line, no event hops
    LAUNCHCK(bjjk::verify_scan(st, scan_grid, pk, r, msg, n, S->slow), "verify scan");
  }
  LAUNCHCK(bjjk::verify_main(st, mode, grid_for(c, n, c->occ_verify, 64) * 64 / BJJ_VERIFY_BLOCK, schnorr, c->table, c->W, c->nwin, pk, r, s, msg, n, ok,
                             S->vb_tables, S->slow, S->slotq, S->slot_cap | ((u32)c->xccs << 16)), "verify");
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
  SET_ENTER(c, stream, n, false);
  { int rc_ = enqueue_verify(c, S, st, schnorr, (const uint8_t*)d_pk, (const uint8_t*)d_r, (const uint8_t*)d_s, (const uint8_t*)d_msg, n,
                             (uint8_t*)d_ok); if (rc_) return rc_; }
  SET_LEAVE(c);
}
int bjj_eddsa_verify_dev(bjj_ctx* c, const void* d_pk, const void* d_r, const void* d_s, const void* d_msg, size_t n,
                         void* d_ok, void* stream) {
  return verify_launch(c, false, d_pk, d_r, d_s, d_msg, n, d_ok, stream, "bjj_eddsa_verify_dev");
}
// ---- verify through the host-pointer pipeline -----------------------------------------------------------------------
// Items whose pk or R is off the curve take the reference's exact formula sequence, ~3x as long as a bulk item and strictly
// serial per lane (k_verify.hip).  A device-pointer launch starts them first and they are long done when the bulk is; a
// CHUNK of the pipeline that carries its own exact items lasts at least as long as they do -- 6.6 ms for the first 2^15 items
// instead of 2.3 -- and the next chunk of its lane waits behind it: 2^20 verifications of BASELINE configs[3] (1 in 64
// corrupted, half of those off the curve) took 25 ms on pinned host memory against 18.7 ms on device pointers
// (profiles/r05_host_verify_exact_split.txt).  Here the two kinds of work are separate launches:
//   per chunk, behind its H2D:  the on-curve scan of its items on the priority stream, appending batch-wide indices to ONE list
//                               the bulk launch of its items on its lane (no scan needed: a bulk workgroup recognises and skips
//                               an off-curve item by itself)
//   behind the last scan:       ONE exact launch over the list, on the priority stream (its few workgroups take the next slots
//                               that free up); it shares the first scratch set's slot queue with that set's bulk launches -- the
//                               queue holds a slot for every workgroup that can be resident, whatever launch it belongs to
// The verdicts leave once, after both (PipeSpec::out_at_end: 1 byte per item).
static int verify_bulk_launch(bjj_ctx* c, bool schnorr, const void* d_pk, const void* d_r, const void* d_s, const void* d_msg, size_t n,
                              void* d_ok, void* stream) {
  SET_ENTER(c, stream, n, false);
  c->rings_used = true;
  c->last_verify_mode = 1;
  LAUNCHCK(bjjk::verify_main(st, 1, 0, schnorr, c->table, c->W, c->nwin, (const uint8_t*)d_pk, (const uint8_t*)d_r, (const uint8_t*)d_s,
                             (const uint8_t*)d_msg, n, (uint8_t*)d_ok, S->vb_tables, S->slow, S->slotq, S->slot_cap | ((u32)c->xccs << 16),
                             bjjk::VERIFY_BULK), "verify (bulk)");
  SET_LEAVE(c);
}
struct VerifyPipe : PipeExtra {
  bjj_ctx* c;
  bool schnorr;
  const uint8_t *pk = nullptr, *r = nullptr, *s = nullptr, *msg = nullptr;
  uint8_t* ok = nullptr;
  size_t n = 0;
  VerifyPipe(bjj_ctx* c_, bool schnorr_) : c(c_), schnorr(schnorr_) {}
  int begin(size_t n_, void** d_in, void** d_out) override {
    n = n_;
    pk = (const uint8_t*)d_in[0]; r = (const uint8_t*)d_in[1]; s = (const uint8_t*)d_in[2]; msg = (const uint8_t*)d_in[3];
    ok = (uint8_t*)d_out[0];
    ScratchSet* S = &c->set[0];
    { int rc_ = ensure_scratch(c, S, 1); if (rc_) return rc_; }      // the set's tables and slot queues exist
    { int rc_ = ensure_scan_stream(S); if (rc_) return rc_; }
    // the exact launch works in this set's tables without being one of its calls (set_enter / set_leave): whatever used the set
    // last on another stream -- a variable-base launch a caller has enqueued and not waited for lays ITS tables over the same
    // memory -- must be done first.  Afterwards only this call's own bulk launches touch the set (slot queue), and the call
    // does not return before the exact launch has completed (ev_tail).
    { int rc_ = set_enter(c, S, S->scan_stream); if (rc_) return rc_; }
    if (n > c->pipe_wl_items) {
      if (c->pipe_wl) { HIPCK(hipDeviceSynchronize()); HIPCK(hipFree(c->pipe_wl)); c->pipe_wl = nullptr; c->pipe_wl_items = 0; }
      HIPCK(hipMalloc((void**)&c->pipe_wl, (n + 16) * sizeof(u32)));
      c->pipe_wl_items = n;
    }
    LAUNCHCK(bjjk::verify_list_reset(S->scan_stream, c->pipe_wl), "verify list");
    return BJJ_OK;
  }
  int chunk_arrived(size_t lo, size_t cnt, hipEvent_t arrived) override {
    ScratchSet* S = &c->set[0];
    HIPCK(hipStreamWaitEvent(S->scan_stream, arrived, 0));
    const int grid = grid_for(c, cnt, c->occ_scan, 64) * 64 / bjjk::verify_scan_block();
    LAUNCHCK(bjjk::verify_scan_range(S->scan_stream, grid, pk, r, msg, lo, lo + cnt, c->pipe_wl), "verify scan");
    return BJJ_OK;
  }
  int all_arrived(hipEvent_t ev_tail) override {
    ScratchSet* S = &c->set[0];
    c->rings_used = true;
    LAUNCHCK(bjjk::verify_main(S->scan_stream, 1, 0, schnorr, c->table, c->W, c->nwin, pk, r, s, msg, n, ok, S->vb_tables, c->pipe_wl, S->slotq,
                               S->slot_cap | ((u32)c->xccs << 16), bjjk::VERIFY_EXACT), "verify (exact)");
    HIPCK(hipEventRecord(ev_tail, S->scan_stream));
    return BJJ_OK;
  }
};
static int verify_host(bjj_ctx* c, bool schnorr, const uint8_t* pk, const uint8_t* r, const uint8_t* s, const uint8_t* msg, size_t n, uint8_t* ok) {
  PipeSpec sp = {4, 1, {pk, r, s, msg}, {64, 64, 32, 32}, {ok}, {1}, false};
  if (c->verify_mode < 0) sp.small_direct_max = small_direct_items(c->verify_small_max);
  static const bool per_chunk = [] { const char* e = getenv("BJJ_PIPE_VERIFY_SPLIT"); return e && e[0] == '0'; }();   // developer: the round-5 form
  // a call of ONE chunk is a device-pointer launch with copies around it: its exact groups start first inside the launch, nothing
  // waits behind them, and three launches instead of two would only add latency (a single `verify`, src/lib.rs:395, is such a call)
  { ENTER_DEVICE(c->device); int rc_ = ensure_pipe(c, 0, 0, 0, 0); if (rc_) return rc_; }
  const size_t first = c->pipe_env_schedule ? c->pipe_first : (size_t)1 << 16;
  const bool one_chunk = n < first + first / 2;
  sp.first_chunk = first;
  sp.max_chunk = (size_t)1 << 19;
  if (c->verify_mode == 0 || per_chunk || one_chunk)   // (the persistent form, when forced, is one launch per chunk, exact items included)
    return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) {
      return schnorr ? bjj_schnorr_verify_dev(c, i[0], i[1], i[2], i[3], cnt, o[0], st) : bjj_eddsa_verify_dev(c, i[0], i[1], i[2], i[3], cnt, o[0], st); });
  VerifyPipe vp(c, schnorr);
  sp.extra = &vp;
  sp.out_at_end = true;
  // 18 ms of kernels against 3.9 ms of H2D for 2^20 items: the copy-out never bounds this call, what costs is the ramp (the
  // chip is part empty until the first chunks have arrived) and every launch's partly empty last round -- fewer, larger chunks
  // than the copy-bound default (2^16 / 2^19: 19.17 ms, 2^15 / 2^18: 19.36 ms; profiles/r05_host_verify_exact_split.txt)
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return verify_bulk_launch(c, schnorr, i[0], i[1], i[2], i[3], cnt, o[0], st); });
}
// ---- the wire-format verifier through the pipeline (round 6) ------------------------------------------------------------------
// bjj_eddsa_verify_compressed on host pointers used to be the device entry point per chunk: decompressions, scan and a launch that carried the
// chunk's own exact groups -- the form the plain verifiers left in round 5 -- and with the pipeline's default chunks it took 31.7 ms for 2^20
// signatures of which 1 in 64 is corrupted (39.5 ms with the 2^17-item cap of round 6) against 23.5 ms for one device-pointer launch.  Now, like
// VerifyPipe, with one more stage in front:
//   per chunk, on its lane:     decompress pk, decompress R (+ s) into the super-batch's staging (d_extra: 162 B per item), an event;
//                               behind the event, on the priority stream, the on-curve scan of the chunk into ONE batch-wide list;
//                               the bulk launch of the chunk on its lane
//   behind the last chunk:      ONE exact launch over the list, then ONE pass that writes verdict 2 where pk or R did not decompress (such an item
//                               decompresses to (0, 0), which is off the curve: it is on the list, the exact launch writes it last, the pass after it)
struct VerifyCompressedPipe : PipeExtra {
  bjj_ctx* c;
  const uint8_t *pk32 = nullptr, *sig64 = nullptr, *msg = nullptr;
  uint8_t *ok = nullptr, *pk_xy = nullptr, *r_xy = nullptr, *s32 = nullptr, *f_pk = nullptr, *f_r = nullptr;
  size_t n = 0, launched = 0;
  explicit VerifyCompressedPipe(bjj_ctx* c_) : c(c_) {}
  int begin(size_t n_, void** d_in, void** d_out) override {
    n = n_; launched = 0;
    pk32 = (const uint8_t*)d_in[0]; sig64 = (const uint8_t*)d_in[1]; msg = (const uint8_t*)d_in[2];
    ok = (uint8_t*)d_out[0];
    pk_xy = (uint8_t*)d_extra; r_xy = pk_xy + n * 64; s32 = r_xy + n * 64; f_pk = s32 + n * 32; f_r = f_pk + n;
    ScratchSet* S = &c->set[0];
    { int rc_ = ensure_scratch(c, S, 1); if (rc_) return rc_; }
    { int rc_ = ensure_scan_stream(S); if (rc_) return rc_; }
    { int rc_ = set_enter(c, S, S->scan_stream); if (rc_) return rc_; }     // (as VerifyPipe: the exact launch works in this set's tables)
    if (n > c->pipe_wl_items) {
      if (c->pipe_wl) { HIPCK(hipDeviceSynchronize()); HIPCK(hipFree(c->pipe_wl)); c->pipe_wl = nullptr; c->pipe_wl_items = 0; }
      HIPCK(hipMalloc((void**)&c->pipe_wl, (n + 16) * sizeof(u32)));
      c->pipe_wl_items = n;
    }
    LAUNCHCK(bjjk::verify_list_reset(S->scan_stream, c->pipe_wl), "verify list");
    return BJJ_OK;
  }
  int chunk_arrived(size_t, size_t, hipEvent_t) override { return BJJ_OK; }
  int all_arrived(hipEvent_t) override { return BJJ_OK; }
  // one chunk on its lane (called by the pipeline's launch closure): items lo .. lo + cnt - 1 of the super-batch
  int launch_chunk(size_t lo, size_t cnt, hipStream_t lane) {
    ScratchSet* S = &c->set[0];
    const int g = grid_for(c, cnt, c->occ_decomp);
    LAUNCHCK(bjjk::decompress_points(lane, g, pk32 + lo * 32, 32, cnt, pk_xy + lo * 64, f_pk + lo, nullptr), "decompress(pk)");
    LAUNCHCK(bjjk::decompress_points(lane, g, sig64 + lo * 64, 64, cnt, r_xy + lo * 64, f_r + lo, s32 + lo * 32), "decompress(sig)");
    try {
      while (c->ev_dec.size() <= launched) { hipEvent_t e = nullptr; HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_dec.push_back(e); }
    } catch (...) { return set_err(BJJ_E_NOMEM, "host-pointer pipeline: out of host memory"); }
    hipEvent_t ev = c->ev_dec[launched++];
    HIPCK(hipEventRecord(ev, lane));
    HIPCK(hipStreamWaitEvent(S->scan_stream, ev, 0));
    const int sg = grid_for(c, cnt, c->occ_scan, 64) * 64 / bjjk::verify_scan_block();
    LAUNCHCK(bjjk::verify_scan_range(S->scan_stream, sg, pk_xy, r_xy, msg, lo, lo + cnt, c->pipe_wl), "verify scan");
    return verify_bulk_launch(c, false, pk_xy + lo * 64, r_xy + lo * 64, s32 + lo * 32, msg + lo * 32, cnt, ok + lo, (void*)lane);
  }
  int all_launched(hipEvent_t ev_tail) override {
    ScratchSet* S = &c->set[0];
    c->rings_used = true;
    LAUNCHCK(bjjk::verify_main(S->scan_stream, 1, 0, false, c->table, c->W, c->nwin, pk_xy, r_xy, s32, msg, n, ok, S->vb_tables, c->pipe_wl, S->slotq,
                               S->slot_cap | ((u32)c->xccs << 16), bjjk::VERIFY_EXACT), "verify (exact)");
    LAUNCHCK(bjjk::merge_codec_flags(S->scan_stream, grid_for(c, n, 8), ok, f_pk, f_r, n), "merge_codec_flags");
    HIPCK(hipEventRecord(ev_tail, S->scan_stream));
    return BJJ_OK;
  }
};
static int verify_compressed_host(bjj_ctx* c, const uint8_t* pk32, const uint8_t* sig64, const uint8_t* msg, size_t n, uint8_t* ok) {
  PipeSpec sp = {3, 1, {pk32, sig64, msg}, {32, 64, 32}, {ok}, {1}, false};
  static const bool per_chunk = [] { const char* e = getenv("BJJ_PIPE_VERIFY_SPLIT"); return e && e[0] == '0'; }();   // developer: the form until round 6
  { ENTER_DEVICE(c->device); int rc_ = ensure_pipe(c, 0, 0, 0, 0); if (rc_) return rc_; }
  const size_t first = c->pipe_env_schedule ? c->pipe_first : (size_t)1 << 16;
  sp.first_chunk = first;
  sp.max_chunk = (size_t)1 << 19;          // the verifiers' schedule: 23 ms of kernels hide 2.5 ms of copies several times over
  if (c->verify_mode == 0 || per_chunk || n < first + first / 2)
    return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_eddsa_verify_compressed_dev(c, i[0], i[1], i[2], cnt, o[0], st); });
  VerifyCompressedPipe vp(c);
  sp.extra = &vp;
  sp.extra_dev_per_item = 162;             // decompressed pk (64) + R (64) + s (32) + the two decompression flags
  sp.out_at_end = true;
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) {
    (void)o;
    return vp.launch_chunk((size_t)((const uint8_t*)i[0] - vp.pk32) / 32, cnt, (hipStream_t)st); });
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
  DEV_ENTER(c, stream);
  LAUNCHCK(bjjk::point_add(st, grid_for(c, n, c->occ_add), (const uint8_t*)d_p, (const uint8_t*)d_q, n, (uint8_t*)d_out), "bjj_point_add_dev");
  DEV_LEAVE(c);
}
int bjj_proj_add_dev(bjj_ctx* c, const void* d_p, const void* d_q, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_proj_add_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_p, "bjj_proj_add_dev"); CHECK_PTR(d_q, "bjj_proj_add_dev"); CHECK_PTR(d_out, "bjj_proj_add_dev");
  DEV_ENTER(c, stream);
  LAUNCHCK(bjjk::proj_add(st, grid_for(c, n, c->occ_add), (const uint8_t*)d_p, (const uint8_t*)d_q, n, (uint8_t*)d_out), "bjj_proj_add_dev");
  DEV_LEAVE(c);
}
int bjj_proj_affine_dev(bjj_ctx* c, const void* d_p, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_proj_affine_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_p, "bjj_proj_affine_dev"); CHECK_PTR(d_out, "bjj_proj_affine_dev");
  DEV_ENTER(c, stream);
  LAUNCHCK(bjjk::proj_affine(st, grid_for(c, n, c->occ_add), (const uint8_t*)d_p, n, (uint8_t*)d_out), "bjj_proj_affine_dev");
  DEV_LEAVE(c);
}

int bjj_compress_points_dev(bjj_ctx* c, const void* d_pts, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_compress_points_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_pts, "bjj_compress_points_dev"); CHECK_PTR(d_out, "bjj_compress_points_dev");
  DEV_ENTER(c, stream);
  LAUNCHCK(bjjk::compress_points(st, grid_for(c, n, 8), (const uint8_t*)d_pts, n, (uint8_t*)d_out), "bjj_compress_points_dev");
  DEV_LEAVE(c);
}
int bjj_decompress_points_dev(bjj_ctx* c, const void* d_in, size_t n, void* d_out_xy, void* d_ok, void* stream) {
  CHECK_CTX(c, "bjj_decompress_points_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_in, "bjj_decompress_points_dev"); CHECK_PTR(d_out_xy, "bjj_decompress_points_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_decompress_points_dev: d_ok is NULL");
  DEV_ENTER(c, stream);
  LAUNCHCK(bjjk::decompress_points(st, grid_for(c, n, c->occ_decomp), (const uint8_t*)d_in, 32, n, (uint8_t*)d_out_xy, (uint8_t*)d_ok,
                                   nullptr), "bjj_decompress_points_dev");
  DEV_LEAVE(c);
}
int bjj_eddsa_verify_compressed_dev(bjj_ctx* c, const void* d_pk32, const void* d_sig64, const void* d_msg, size_t n,
                                    void* d_ok, void* stream) {
  CHECK_CTX(c, "bjj_eddsa_verify_compressed_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_pk32, "bjj_eddsa_verify_compressed_dev"); CHECK_PTR(d_sig64, "bjj_eddsa_verify_compressed_dev");
  CHECK_PTR(d_msg, "bjj_eddsa_verify_compressed_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_eddsa_verify_compressed_dev: d_ok is NULL");
  SET_ENTER(c, stream, n, true);
  uint8_t* pk_xy = S->codec;
  uint8_t* r_xy = pk_xy + n * 64;
  uint8_t* s32 = r_xy + n * 64;
  uint8_t* f_pk = s32 + n * 32;
  uint8_t* f_r = f_pk + n;
  const int g = grid_for(c, n, c->occ_decomp);
  LAUNCHCK(bjjk::decompress_points(st, g, (const uint8_t*)d_pk32, 32, n, pk_xy, f_pk, nullptr), "decompress(pk)");
  LAUNCHCK(bjjk::decompress_points(st, g, (const uint8_t*)d_sig64, 64, n, r_xy, f_r, s32), "decompress(sig)");
  { int rc_ = enqueue_verify(c, S, st, false, pk_xy, r_xy, s32, (const uint8_t*)d_msg, n, (uint8_t*)d_ok); if (rc_) return rc_; }
  LAUNCHCK(bjjk::merge_codec_flags(st, grid_for(c, n, 8), (uint8_t*)d_ok, f_pk, f_r, n), "merge_codec_flags");
  SET_LEAVE(c);
}

// ---- signer hardening ---------------------------------------------------------------------------------------------
#define BJJ_CT_W 4
static int ensure_ct_table(bjj_ctx* c) {
  if (c->ct_table) return BJJ_OK;
  ENTER_DEVICE(c->device);
  const int nwin = fixed_nwin(BJJ_CT_W);
  const size_t bytes = fixed_stride(BJJ_CT_W) * (size_t)nwin * NIELS_WORDS * sizeof(u32);
  u32 *t = nullptr, *b = nullptr;
  unsigned long long* d_bad = nullptr;
  unsigned long long bad = 1;
  hipError_t e = hipMalloc((void**)&t, bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&b, (size_t)nwin * NIELS_WORDS * sizeof(u32));
  if (e == hipSuccess) e = hipMalloc((void**)&d_bad, sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), c->stream);
  if (e == hipSuccess) e = bjjk::build_fixed_table(c->stream, t, b, BJJ_CT_W, nwin);
  if (e == hipSuccess) e = bjjk::check_fixed_table(c->stream, 8, t, b, BJJ_CT_W, nwin, d_bad);   // the same induction proof as the big table
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (d_bad) hipFree(d_bad);
  if (e != hipSuccess || bad != 0) {
    if (t) hipFree(t);
    if (b) hipFree(b);
    return e != hipSuccess ? set_err(BJJ_E_HIP, std::string("constant-time signer table: ") + hipGetErrorString(e))
                           : set_err(BJJ_E_HIP, "constant-time signer table failed its self-check");
  }
  c->ct_table = t; c->ct_bases = b;
  c->occ_sign_ct = bjjk::occ_sign_ct();
  c->occ_sign_schnorr_ct = bjjk::occ_sign_schnorr_ct();
  return BJJ_OK;
}
int bjj_set_signer_constant_time(bjj_ctx* c, int on) {
  CHECK_CTX(c, "bjj_set_signer_constant_time");
  if (on) { int rc = ensure_ct_table(c); if (rc) return rc; }
  c->ct_signer = on != 0;
  return BJJ_OK;
}

int bjj_scalar_keys_dev(bjj_ctx* c, const void* d_keys, size_t n, void* d_out, void* stream) {
  CHECK_CTX(c, "bjj_scalar_keys_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_keys, "bjj_scalar_keys_dev"); CHECK_PTR(d_out, "bjj_scalar_keys_dev");
  DEV_ENTER(c, stream);
  LAUNCHCK(bjjk::scalar_keys(st, grid_for(c, n, 4), (const uint8_t*)d_keys, n, (uint8_t*)d_out), "bjj_scalar_keys_dev");
  DEV_LEAVE(c);
}
// compressed: d_out holds the 32-byte records of sk.public().compress() (src/lib.rs:304-306, 166-178)
static int public_keys_launch(bjj_ctx* c, const void* d_keys, size_t n, void* d_out_xy, void* stream, bool compressed, const char* who) {
  if (!c) return set_err(BJJ_E_INVALID, std::string(who) + ": ctx is NULL");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  if (!d_keys || !d_out_xy || !aligned16(d_keys) || !aligned16(d_out_xy))
    return set_err(BJJ_E_INVALID, std::string(who) + ": NULL or not 16-byte aligned device pointer");
  SET_ENTER(c, stream, n, true);
  if (compressed) { int rc_ = ensure_xy(c, S, n); if (rc_) return rc_; }
  uint8_t* xy = compressed ? S->xy : nullptr;
  // B8.mul_scalar(&self.scalar_key()), src/lib.rs:304-306; the scalar keys live in the codec scratch only for the
  // duration of the multiplication and are wiped on the same stream right behind it
  LAUNCHCK(bjjk::scalar_keys(st, grid_for(c, n, 4), (const uint8_t*)d_keys, n, S->codec), "scalar_keys");
  if (c->ct_signer) {
    LAUNCHCK(bjjk::mul_fixed_base_scan(st, c->cus, c->ct_table, BJJ_CT_W, fixed_nwin(BJJ_CT_W), S->codec, n, (uint8_t*)d_out_xy, S->scratch, xy),
             "mul_fixed_base_scan");
  } else if (n <= c->fb_quad_max && c->k1_variant < 0) {   // short calls: four lanes per key (k_small.hip)
    c->last_k1 = 2;
    LAUNCHCK(bjjk::mul_fixed_base_quad(st, c->table, c->W, c->nwin, S->codec, n, (uint8_t*)d_out_xy, compressed), "mul_fixed_base (short call)");
  } else {
    const int kv = fixed_base_variant(c, S);
    LAUNCHCK(bjjk::mul_fixed_base(st, c->cus, fixed_base_lanes(c, kv), kv, c->table, c->W, c->nwin, S->codec, n, (uint8_t*)d_out_xy,
                                  S->scratch, xy), "mul_fixed_base");
  }
  HIPCK(hipMemsetAsync(S->codec, 0, n * 32, st));
  SET_LEAVE(c);
}
int bjj_public_keys_dev(bjj_ctx* c, const void* d_keys, size_t n, void* d_out_xy, void* stream) {
  return public_keys_launch(c, d_keys, n, d_out_xy, stream, false, "bjj_public_keys_dev");
}
int bjj_public_keys_compressed_dev(bjj_ctx* c, const void* d_keys, size_t n, void* d_out32, void* stream) {
  return public_keys_launch(c, d_keys, n, d_out32, stream, true, "bjj_public_keys_compressed_dev");
}
// d_out_s == NULL: the compressed form -- d_out_r holds 64-byte Signature::compress records (src/lib.rs:245-258)
static int sign_launch(bjj_ctx* c, const void* d_keys, const void* d_msgs, size_t n, void* d_out_r, void* d_out_s, void* d_ok, void* stream) {
  DEV_ENTER(c, stream);
  c->last_sign = (!c->ct_signer && n <= c->sign_small_max) ? 1 : 0;
  if (c->last_sign)   // short calls (a single sk.sign(msg) is one): eight lanes per signature, the hash on six of them (k_small.hip)
    LAUNCHCK(bjjk::sign_small(st, c->table, c->W, c->nwin, (const uint8_t*)d_keys, (const uint8_t*)d_msgs, n, (uint8_t*)d_out_r, (uint8_t*)d_out_s,
                              (uint8_t*)d_ok), "bjj_sign_dev (short call)");
  else
  if (c->ct_signer)
    LAUNCHCK(bjjk::sign_ct(st, grid_for(c, n, c->occ_sign_ct), c->ct_table, BJJ_CT_W, fixed_nwin(BJJ_CT_W), (const uint8_t*)d_keys,
                           (const uint8_t*)d_msgs, n, (uint8_t*)d_out_r, (uint8_t*)d_out_s, (uint8_t*)d_ok), "bjj_sign_dev (constant-time)");
  else
  LAUNCHCK(bjjk::sign(st, grid_for(c, n, c->occ_sign), c->table, c->W, c->nwin, (const uint8_t*)d_keys, (const uint8_t*)d_msgs, n,
                      (uint8_t*)d_out_r, (uint8_t*)d_out_s, (uint8_t*)d_ok), "bjj_sign_dev");
  DEV_LEAVE(c);
}
int bjj_sign_dev(bjj_ctx* c, const void* d_keys, const void* d_msgs, size_t n, void* d_out_r, void* d_out_s, void* d_ok,
                 void* stream) {
  CHECK_CTX(c, "bjj_sign_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_keys, "bjj_sign_dev"); CHECK_PTR(d_msgs, "bjj_sign_dev"); CHECK_PTR(d_out_r, "bjj_sign_dev");
  CHECK_PTR(d_out_s, "bjj_sign_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_sign_dev: d_ok is NULL");
  return sign_launch(c, d_keys, d_msgs, n, d_out_r, d_out_s, d_ok, stream);
}
int bjj_sign_compressed_dev(bjj_ctx* c, const void* d_keys, const void* d_msgs, size_t n, void* d_out_sig64, void* d_ok, void* stream) {
  CHECK_CTX(c, "bjj_sign_compressed_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_keys, "bjj_sign_compressed_dev"); CHECK_PTR(d_msgs, "bjj_sign_compressed_dev"); CHECK_PTR(d_out_sig64, "bjj_sign_compressed_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_sign_compressed_dev: d_ok is NULL");
  return sign_launch(c, d_keys, d_msgs, n, d_out_sig64, nullptr, d_ok, stream);
}

int bjj_sign_schnorr_dev(bjj_ctx* c, const void* d_keys, const void* d_msgs, const void* d_nonces, size_t n, void* d_out_r,
                         void* d_out_s, void* d_ok, void* stream) {
  CHECK_CTX(c, "bjj_sign_schnorr_dev");
  if (n == 0) return BJJ_OK;
  CHECK_N(n);
  CHECK_PTR(d_keys, "bjj_sign_schnorr_dev"); CHECK_PTR(d_msgs, "bjj_sign_schnorr_dev"); CHECK_PTR(d_nonces, "bjj_sign_schnorr_dev");
  CHECK_PTR(d_out_r, "bjj_sign_schnorr_dev"); CHECK_PTR(d_out_s, "bjj_sign_schnorr_dev");
  if (!d_ok) return set_err(BJJ_E_INVALID, "bjj_sign_schnorr_dev: d_ok is NULL");
  DEV_ENTER(c, stream);
  if (c->ct_signer)
    LAUNCHCK(bjjk::sign_schnorr_ct(st, grid_for(c, n, c->occ_sign_schnorr_ct), c->ct_table, BJJ_CT_W, fixed_nwin(BJJ_CT_W), (const uint8_t*)d_keys,
                                   (const uint8_t*)d_msgs, (const uint8_t*)d_nonces, n, (uint8_t*)d_out_r, (uint8_t*)d_out_s, (uint8_t*)d_ok),
             "bjj_sign_schnorr_dev (constant-time)");
  else
  LAUNCHCK(bjjk::sign_schnorr(st, grid_for(c, n, c->occ_sign_schnorr), c->table, c->W, c->nwin, (const uint8_t*)d_keys,
                              (const uint8_t*)d_msgs, (const uint8_t*)d_nonces, n, (uint8_t*)d_out_r, (uint8_t*)d_out_s, (uint8_t*)d_ok),
           "bjj_sign_schnorr_dev");
  DEV_LEAVE(c);
}

// ---- host-pointer API: chunked pinned-staging pipeline around the *_dev entry points ----------
#define HOST_PROLOGUE(name, cond)                                             \
  CHECK_CTX(c, name);                                                         \
  if (n == 0) return BJJ_OK;                                                  \
  if (cond) return set_err(BJJ_E_INVALID, name ": NULL buffer")

// Chunk schedule of the K1 entry points with 32-byte results (the copy-out does not bound them: the launches do): chunks of 2^17
// items behind a first one of 2^16 -- one round of multiplications for every lane of a launch that takes one workgroup slot per CU
// (PipeSpec::k1_half).  2^18 .. 2^20 items: 8-15 % less than the schedule of the rounds before (doubling to 2^18, a 2^15-item last
// chunk; BJJ_PIPE_K1_HALF=0 brings it back for an A/B), profiles/r06_fb_host_half_slots.txt.  The affine forms keep the default
// schedule and full launches: their copy-out (64 bytes per item) is what bounds them, and they measured the same either way.
// Long calls (more than 1.5 * 2^20 items) take chunks of 2^18: their steady state is the copy engines', which move 8 MB blocks faster
// than 4 MB blocks (2^21 .. 2^23 items: 5-15 % less than with 2^17; 2^20 items: 14 % more).
static void k1_chunk_schedule(PipeSpec* sp, size_t n) {
  static const bool half = [] { const char* e = getenv("BJJ_PIPE_K1_HALF"); return !(e && e[0] == '0'); }();
  if (half) { sp->k1_half = true; sp->first_chunk = (size_t)1 << 16; sp->max_chunk = (size_t)1 << (n > ((size_t)3 << 19) ? 18 : 17); }
  else { sp->tail_chunk = (size_t)1 << 15; sp->max_chunk = (size_t)1 << 18; }
}
int bjj_mul_fixed_base(bjj_ctx* c, const uint8_t* scalars, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_mul_fixed_base", !scalars || !out);
  PipeSpec sp = {1, 1, {scalars}, {32}, {out}, {64}, false};
  sp.zero_copy_in = true;
  if (c->k1_variant < 0) sp.small_direct_max = small_direct_items(c->fb_quad_max);
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_mul_fixed_base_dev(c, i[0], cnt, o[0], st); });
}
// 32 bytes per result across PCIe instead of 64: the copy-out is what bounds the affine form (1.19 ms of 1.58 per 2^20 items)
int bjj_mul_fixed_base_compressed(bjj_ctx* c, const uint8_t* scalars, size_t n, uint8_t* out32) {
  HOST_PROLOGUE("bjj_mul_fixed_base_compressed", !scalars || !out32);
  PipeSpec sp = {1, 1, {scalars}, {32}, {out32}, {32}, false};
  if (c->k1_variant < 0) sp.small_direct_max = small_direct_items(c->fb_quad_max);
  k1_chunk_schedule(&sp, n);
  static const bool zc = [] { const char* e = getenv("BJJ_FB_COMPRESSED_ZERO_COPY"); return e && e[0] == '1'; }();   // experiment (tools/fb_compressed_sweep.py)
  if (zc) { sp.zero_copy_out = true; sp.tail_chunk = 0; }
  sp.zero_copy_in = true;
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_mul_fixed_base_compressed_dev(c, i[0], cnt, o[0], st); });
}
int bjj_mul_var_base(bjj_ctx* c, const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_mul_var_base", !pts || !scalars || !out);
  return var_base_host(c, pts, scalars, 32, n, out);
}
int bjj_mul_var_base_wide(bjj_ctx* c, const uint8_t* pts, const uint8_t* scalars, size_t scalar_bytes, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_mul_var_base_wide", !pts || !scalars || !out);
  if (scalar_bytes == 0 || (scalar_bytes & 31) || scalar_bytes > BJJ_MAX_SCALAR_BYTES)
    return set_err(BJJ_E_INVALID, "bjj_mul_var_base_wide: scalar_bytes must be a multiple of 32 in 32..BJJ_MAX_SCALAR_BYTES");
  return var_base_host(c, pts, scalars, scalar_bytes, n, out);
}
int bjj_poseidon5(bjj_ctx* c, const uint8_t* in, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_poseidon5", !in || !out);
  PipeSpec sp = {1, 1, {in}, {160}, {out}, {32}, false};
  sp.small_direct_max = small_direct_items(c->p5_coop_max);
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_poseidon5_dev(c, i[0], cnt, o[0], st); });
}
int bjj_eddsa_verify(bjj_ctx* c, const uint8_t* pk, const uint8_t* r, const uint8_t* s, const uint8_t* msg, size_t n,
                     uint8_t* ok) {
  HOST_PROLOGUE("bjj_eddsa_verify", !pk || !r || !s || !msg || !ok);
  return verify_host(c, false, pk, r, s, msg, n, ok);
}
int bjj_schnorr_verify(bjj_ctx* c, const uint8_t* pk, const uint8_t* r, const uint8_t* s, const uint8_t* msg, size_t n,
                       uint8_t* ok) {
  HOST_PROLOGUE("bjj_schnorr_verify", !pk || !r || !s || !msg || !ok);
  return verify_host(c, true, pk, r, s, msg, n, ok);
}
int bjj_point_add(bjj_ctx* c, const uint8_t* p, const uint8_t* q, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_point_add", !p || !q || !out);
  PipeSpec sp = {2, 1, {p, q}, {64, 64}, {out}, {64}, false};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_point_add_dev(c, i[0], i[1], cnt, o[0], st); });
}
int bjj_proj_add(bjj_ctx* c, const uint8_t* p, const uint8_t* q, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_proj_add", !p || !q || !out);
  PipeSpec sp = {2, 1, {p, q}, {96, 96}, {out}, {96}, false};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_proj_add_dev(c, i[0], i[1], cnt, o[0], st); });
}
int bjj_proj_affine(bjj_ctx* c, const uint8_t* p, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_proj_affine", !p || !out);
  PipeSpec sp = {1, 1, {p}, {96}, {out}, {64}, false};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_proj_affine_dev(c, i[0], cnt, o[0], st); });
}
int bjj_compress_points(bjj_ctx* c, const uint8_t* pts, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_compress_points", !pts || !out);
  PipeSpec sp = {1, 1, {pts}, {64}, {out}, {32}, false};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_compress_points_dev(c, i[0], cnt, o[0], st); });
}
int bjj_decompress_points(bjj_ctx* c, const uint8_t* in, size_t n, uint8_t* out_xy, uint8_t* ok) {
  HOST_PROLOGUE("bjj_decompress_points", !in || !out_xy || !ok);
  PipeSpec sp = {1, 2, {in}, {32}, {out_xy, ok}, {64, 1}, false};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_decompress_points_dev(c, i[0], cnt, o[0], o[1], st); });
}
int bjj_eddsa_verify_compressed(bjj_ctx* c, const uint8_t* pk32, const uint8_t* sig64, const uint8_t* msg, size_t n,
                                uint8_t* ok) {
  HOST_PROLOGUE("bjj_eddsa_verify_compressed", !pk32 || !sig64 || !msg || !ok);
  return verify_compressed_host(c, pk32, sig64, msg, n, ok);
}
// signer side: the inputs are key material, so the staging buffers are wiped when the call is done (PipeSpec::secret)
int bjj_scalar_keys(bjj_ctx* c, const uint8_t* keys, size_t n, uint8_t* out) {
  HOST_PROLOGUE("bjj_scalar_keys", !keys || !out);
  PipeSpec sp = {1, 1, {keys}, {32}, {out}, {32}, true};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_scalar_keys_dev(c, i[0], cnt, o[0], st); });
}
int bjj_public_keys(bjj_ctx* c, const uint8_t* keys, size_t n, uint8_t* out_xy) {
  HOST_PROLOGUE("bjj_public_keys", !keys || !out_xy);
  PipeSpec sp = {1, 1, {keys}, {32}, {out_xy}, {64}, true};
  if (c->k1_variant < 0 && !c->ct_signer) sp.small_direct_max = small_direct_items(c->fb_quad_max);
  sp.zero_copy_in = true;
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_public_keys_dev(c, i[0], cnt, o[0], st); });
}
int bjj_public_keys_compressed(bjj_ctx* c, const uint8_t* keys, size_t n, uint8_t* out32) {
  HOST_PROLOGUE("bjj_public_keys_compressed", !keys || !out32);
  PipeSpec sp = {1, 1, {keys}, {32}, {out32}, {32}, true};
  if (c->k1_variant < 0 && !c->ct_signer) sp.small_direct_max = small_direct_items(c->fb_quad_max);
  k1_chunk_schedule(&sp, n);
  sp.zero_copy_in = true;
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_public_keys_compressed_dev(c, i[0], cnt, o[0], st); });
}
int bjj_sign_compressed(bjj_ctx* c, const uint8_t* keys, const uint8_t* msgs, size_t n, uint8_t* out_sig64, uint8_t* ok) {
  HOST_PROLOGUE("bjj_sign_compressed", !keys || !msgs || !out_sig64 || !ok);
  PipeSpec sp = {2, 2, {keys, msgs}, {32, 32}, {out_sig64, ok}, {64, 1}, true};
  if (!c->ct_signer) sp.small_direct_max = small_direct_items(c->sign_small_max);
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_sign_compressed_dev(c, i[0], i[1], cnt, o[0], o[1], st); });
}
int bjj_sign(bjj_ctx* c, const uint8_t* keys, const uint8_t* msgs, size_t n, uint8_t* out_r, uint8_t* out_s, uint8_t* ok) {
  HOST_PROLOGUE("bjj_sign", !keys || !msgs || !out_r || !out_s || !ok);
  PipeSpec sp = {2, 3, {keys, msgs}, {32, 32}, {out_r, out_s, ok}, {64, 32, 1}, true};
  if (!c->ct_signer) sp.small_direct_max = small_direct_items(c->sign_small_max);
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_sign_dev(c, i[0], i[1], cnt, o[0], o[1], o[2], st); });
}
int bjj_sign_schnorr(bjj_ctx* c, const uint8_t* keys, const uint8_t* msgs, const uint8_t* nonces, size_t n, uint8_t* out_r,
                     uint8_t* out_s, uint8_t* ok) {
  HOST_PROLOGUE("bjj_sign_schnorr", !keys || !msgs || !nonces || !out_r || !out_s || !ok);
  PipeSpec sp = {3, 3, {keys, msgs, nonces}, {32, 32, BJJ_SCHNORR_NONCE_BYTES}, {out_r, out_s, ok}, {64, BJJ_SCHNORR_S_BYTES, 1}, true};
  return run_pipelined(c, n, sp, [&](void** i, void** o, size_t cnt, void* st) { return bjj_sign_schnorr_dev(c, i[0], i[1], i[2], cnt, o[0], o[1], o[2], st); });
}

#pragma GCC visibility pop
}  // extern "C"

#include "bjj_multi.inc"
