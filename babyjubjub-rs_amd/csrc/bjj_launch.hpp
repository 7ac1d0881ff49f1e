// Interface between the host side of libbjj_hip.so (bjj_hip.hip: contexts + the extern "C" boundary) and its kernel
// translation units (k_*.hip).  Plain pointers and sizes only; every launcher enqueues on `st` and returns the launch status.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define BJJ_BLOCK 256
// Workgroup size of the kernels that end in the shared-inversion epilogue: one binary-GCD
// inversion (executed by one wave) is amortised over the whole workgroup.
#define BJJ_EPI_BLOCK 512

// Workgroup size of the verify kernels.  Their waves never cooperate beyond the wave (per-wave staging rows, atomic work
// cursors), so a workgroup is ONE wave: a wave that runs out of work retires its slot at once instead of waiting for the
// slowest of four, which is what lets the head of the next launch (second scratch set, second stream) fill a tail round.
#ifndef BJJ_VERIFY_BLOCK
#define BJJ_VERIFY_BLOCK 64
#endif

namespace bjjk {

template <typename K>
static inline int occupancy_of(K kernel, int block) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) nb = 1;
  return nb;
}

// resident workgroups per CU -- resident LANES per CU for K1 / K2 (queried on the current device)
int fixed_base_lanes_per_cu(int variant);   // variant 0: one 512-lane workgroup per CU, 1: two 256-lane workgroups
int var_base_lanes_per_cu();
int var_base_block();
int occ_point_add();
int occ_poseidon5();
int occ_decompress();
int occ_verify();           // resident verify WAVES per CU
int probe_xccs(hipStream_t st, uint32_t* d_word);   // number of XCDs of the current device (0 on error)
int occ_verify_scan();      // resident scan WAVES per CU
int verify_scan_block();
int occ_sign();
int occ_sign_schnorr();

// k_fixed.hip
hipError_t build_fixed_table(hipStream_t st, uint32_t* table, uint32_t* bases, int W, int nwin);
hipError_t check_fixed_table(hipStream_t st, int grid, const uint32_t* table, const uint32_t* bases, int W, int nwin,
                             unsigned long long* d_bad);
// xy == nullptr: out = 64-byte affine points; xy != nullptr: out = 32-byte Point::compress records, xy = 64 B / item of stash
hipError_t mul_fixed_base(hipStream_t st, int cus, int lanes_per_cu, int variant, const uint32_t* table, int W, int nwin,
                          const uint8_t* scalars, size_t n, uint8_t* out, uint32_t* scratch, uint8_t* xy = nullptr);
hipError_t mul_fixed_base_scan(hipStream_t st, int cus, const uint32_t* table, int W, int nwin, const uint8_t* scalars, size_t n,
                               uint8_t* out, uint32_t* scratch, uint8_t* xy = nullptr);   // constant-time form over the small 4-bit table
// k_var.hip (sc_words: 32-bit words per scalar record, 8 for the 32-byte form)
// K2 (slow != nullptr: appends the off-curve items it skips; nullptr: they are on somebody else's list), the on-curve scan that makes
// such a list, and K6 over a list (patch: compact results beside the indices instead of the items' own slots; seen: host word <- count)
hipError_t mul_var_base_main(hipStream_t st, int cus, int lanes_per_cu, int variant, const uint8_t* pts, const uint8_t* scalars, int sc_words, size_t n,
                             uint8_t* out, uint32_t* scratch, uint32_t* vb_tables, uint32_t* slow, uint32_t* slotq, uint32_t slot_cap,
                             uint8_t* xy = nullptr);   // xy: phase-1 stash apart from `out` (which may then be mapped host memory)
hipError_t var_base_list_reset(hipStream_t st, uint32_t* list);
hipError_t var_base_scan(hipStream_t st, int grid, const uint8_t* pts, size_t first, size_t end, uint32_t* list);
hipError_t mul_var_base_exact(hipStream_t st, int grid_exact, const uint8_t* pts, const uint8_t* scalars, int sc_words, uint8_t* out,
                              const uint32_t* slow, uint8_t* patch, uint32_t* seen);
int occ_var_base_scan();
// k_small.hip: four lanes per item, for calls that do not fill the chip one item per lane (32-byte scalars); slow as for mul_var_base_main
// ... Poseidon with six lanes per hash, and the bulk of verify with eight lanes per signature (scan and exact launch: k_verify.hip)
hipError_t verify_small(hipStream_t st, bool schnorr, const uint32_t* table, int W, int nwin, const uint8_t* pk, const uint8_t* rb8, const uint8_t* s, const uint8_t* msg,
                        size_t n, uint8_t* ok);
hipError_t mul_fixed_base_quad(hipStream_t st, const uint32_t* table, int W, int nwin, const uint8_t* scalars, size_t n, uint8_t* out, bool compressed);   // B8.mul_scalar, four lanes per item
hipError_t sign_small(hipStream_t st, const uint32_t* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs, size_t n, uint8_t* out_r, uint8_t* out_s,
                      uint8_t* ok);   // PrivateKey::sign, eight lanes per signature (out_s == nullptr: compressed records)
hipError_t poseidon5_coop(hipStream_t st, const uint8_t* in, size_t n, uint8_t* out);
hipError_t mul_var_base_quad(hipStream_t st, const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out, uint32_t* slow);
hipError_t point_add(hipStream_t st, int grid, const uint8_t* p, const uint8_t* q, size_t n, uint8_t* out);
hipError_t proj_add(hipStream_t st, int grid, const uint8_t* p, const uint8_t* q, size_t n, uint8_t* out);
hipError_t proj_affine(hipStream_t st, int grid, const uint8_t* p, size_t n, uint8_t* out);
// k_hash_codec.hip
hipError_t poseidon5(hipStream_t st, int grid, const uint8_t* in, size_t n, uint8_t* out);
hipError_t compress_points(hipStream_t st, int grid, const uint8_t* in_xy, size_t n, uint8_t* out);
hipError_t decompress_points(hipStream_t st, int grid, const uint8_t* in, size_t stride, size_t n, uint8_t* out_xy, uint8_t* ok,
                             uint8_t* out_s);
hipError_t merge_codec_flags(hipStream_t st, int grid, uint8_t* ok, const uint8_t* f_pk, const uint8_t* f_r, size_t n);
hipError_t scalar_keys(hipStream_t st, int grid, const uint8_t* keys, size_t n, uint8_t* out);
// k_verify.hip
hipError_t verify_scan(hipStream_t st, int grid_scan, const uint8_t* pk, const uint8_t* rb8, const uint8_t* msg, size_t n, uint32_t* wl);
hipError_t verify_list_reset(hipStream_t st, uint32_t* wl);
hipError_t verify_scan_range(hipStream_t st, int grid_scan, const uint8_t* pk, const uint8_t* rb8, const uint8_t* msg, size_t first, size_t end,
                             uint32_t* wl);
enum { VERIFY_BOTH = 0, VERIFY_BULK = 1, VERIFY_EXACT = 2 };   // which workgroups of the one-group-per-workgroup form a launch holds
hipError_t verify_main(hipStream_t st, int mode, int grid, bool schnorr, const uint32_t* table, int W, int nwin, const uint8_t* pk,
                       const uint8_t* rb8, const uint8_t* s, const uint8_t* msg, size_t n, uint8_t* ok, uint32_t* vb_tables,
                       uint32_t* wl, uint32_t* slotq, uint32_t slot_cap, int part = VERIFY_BOTH);
// k_sign.hip
hipError_t sign(hipStream_t st, int grid, const uint32_t* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs, size_t n,
                uint8_t* out_r, uint8_t* out_s, uint8_t* ok);
hipError_t sign_schnorr(hipStream_t st, int grid, const uint32_t* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs,
                        const uint8_t* nonces, size_t n, uint8_t* out_r, uint8_t* out_s, uint8_t* ok);
// the same with the scanning (constant-time) gather policy over the small 4-bit table; ct = true in the occupancy queries
hipError_t sign_ct(hipStream_t st, int grid, const uint32_t* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs, size_t n,
                   uint8_t* out_r, uint8_t* out_s, uint8_t* ok);
hipError_t sign_schnorr_ct(hipStream_t st, int grid, const uint32_t* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs,
                           const uint8_t* nonces, size_t n, uint8_t* out_r, uint8_t* out_s, uint8_t* ok);
int occ_sign_ct();
int occ_sign_schnorr_ct();

}  // namespace bjjk
