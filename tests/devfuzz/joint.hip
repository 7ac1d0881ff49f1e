// TEST-ONLY kernel (not part of libbjj_hip.so): the joint part of the EdDSA fast path as it ships -- joint_short_pair<G>
// (csrc/bjj_device.hpp: per-lane tables of P1 / P2, the window count of the wave, joint_mul_windowed) under the policy the
// verify kernels use, GatherCoopLds<1>, whose wave_max is a cross-lane maximum: the trip count of the joint loop is data
// dependent and shared by the 64 lanes of a wave.  The test (tests/test_gpu_devfuzz.py::test_joint_loop_*) chooses the bit
// lengths of (u, |v|) per lane -- one lane at 250 bits (what kappa = (l+1)/2 produces: 64 windows) next to lanes at 1, 126,
// 127, 130, 131 bits, partly filled last waves -- and compares u*P1 + |v|*P2 with the oracle.
// Same per-lane table layout as the verify unit (BJJ_VERIFY_PNIELS_LAYOUT) and one 64-lane wave per workgroup, like
// bjj_k_eddsa_verify_groups.  `per_lane` = 1 runs the same items with the identity wave_max (every lane its own minimum).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../babyjubjub-rs_amd/csrc/bjj_launch.hpp"
#define BJJ_PNIELS_LAYOUT 0   // = BJJ_VERIFY_PNIELS_LAYOUT of k_verify.hip: raw 144-byte entries
#include "../../babyjubjub-rs_amd/csrc/k_common.hpp"

// u, v: 32-byte little-endian integers (< 2^252); a_xy, p2_xy: affine points of the reference curve (64 bytes each).
// P1 = 8*A by three doublings (Z != 1: the general-Z table build, as verify's -8A), P2 affine (as verify's -+R).
// out: X, Y, Z of the result on the internal a' = -1 curve as canonical plain integers (3 x 32 bytes; the host divides and maps
// x back with 1/F); windows[i] = the number of windows the item's wave ran.
template <class G>
__device__ __forceinline__ void joint_item(const uint8_t* u, const uint8_t* v, const uint8_t* a_xy, const uint8_t* p2_xy, size_t i, size_t n,
                                           u32* tbl, uint8_t* out, int* windows) {
  const size_t ic = i < n ? i : n - 1;    // every lane runs (cross-lane maximum); the tail repeats the last item
  u32 w[8];
  load_w8(u + ic * 32, w);  const Fr uu = fr_from_words(w);
  load_w8(v + ic * 32, w);  const Fr vv = fr_from_words(w);
  load_w8(a_xy + ic * 64, w);       const Fr ax = fr_to_mont_words(w);
  load_w8(a_xy + ic * 64 + 32, w);  const Fr ay = fr_to_mont_words(w);
  load_w8(p2_xy + ic * 64, w);      const Fr px = fr_to_mont_words(w);
  load_w8(p2_xy + ic * 64 + 32, w); const Fr py = fr_to_mont_words(w);
  Ext p1 = ext_from_ref_affine(ax, ay, c_K);
  p1 = ext_dbl<false>(p1); p1 = ext_dbl<false>(p1); p1 = ext_dbl<true>(p1);
  const Ext p2 = ext_from_ref_affine(px, py, c_K);
  int jw = 0;
  const Ext q = joint_short_pair<G>(p1, p2, uu, vv, tbl, c_K, &jw);
  if (i < n) {
    fr_to_words(fr_canon(fr_mul(q.X, fr_one_plain())), w); store_w8(out + i * 96, w);
    fr_to_words(fr_canon(fr_mul(q.Y, fr_one_plain())), w); store_w8(out + i * 96 + 32, w);
    fr_to_words(fr_canon(fr_mul(q.Z, fr_one_plain())), w); store_w8(out + i * 96 + 64, w);
    windows[i] = jw;
  }
}
__global__ void __launch_bounds__(64) jt_kernel(const uint8_t* __restrict__ u, const uint8_t* __restrict__ v, const uint8_t* __restrict__ a_xy,
                                                const uint8_t* __restrict__ p2_xy, size_t n, u32* __restrict__ tables, uint8_t* __restrict__ out,
                                                int* __restrict__ windows, int per_lane) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  u32* tbl = tables + i * VB_VERIFY_WORDS;
  if (per_lane) joint_item<GatherPerLane>(u, v, a_xy, p2_xy, i, n, tbl, out, windows);
  else joint_item<GatherCoopLds<1>>(u, v, a_xy, p2_xy, i, n, tbl, out, windows);
}
// tables: ceil(n / 64) * 64 * jt_table_words() words of device scratch
extern "C" __attribute__((visibility("default"))) int jt_run(const uint8_t* u, const uint8_t* v, const uint8_t* a_xy, const uint8_t* p2_xy, size_t n,
                                                              uint32_t* tables, uint8_t* out, int* windows, int per_lane, void* stream) {
  if (!n) return 0;
  hipLaunchKernelGGL(jt_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, u, v, a_xy, p2_xy, n, tables, out, windows, per_lane);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) int jt_table_words(void) { return VB_VERIFY_WORDS; }
