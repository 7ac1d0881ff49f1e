// Developer experiment: the north_star's Poseidon layout -- one STATE ELEMENT per
// lane, 6 lanes per hash (10 hashes per wave, lanes 60..63 idle), the MDS mix fetching the other five state words of
// the hash across lanes -- against the layout that ships: one HASH per lane, state in registers, sparse partial rounds,
// matrix operands in scalar registers (bjj_k_poseidon5's body, poseidon5()).  Both produce the same 32 output bytes
// (checked here for every hash); the timing goes to profiles/r02_ab_poseidon_layout.txt.
//
// Lane-per-element notes.  6 is not a power of two, so DPP row/bank permutes do not apply: the mix uses ds_bpermute_b32
// (__shfl with an arbitrary source lane), 6 x 9 limb moves per round.  The S-box of a PARTIAL round touches state[0] only,
// but the 6 lanes of a hash execute together: the x^5 costs the same wall time as in a full round (5 of 6 lanes discard
// it).  The round constants and the matrix row differ per lane, so they are VGPR / LDS operands, not scalar registers.
// The sparse partial-round factorisation does not help this layout: its per-round critical path (x^5, a 6-term dot
// product for element 0, then a multiply-add for elements 1..5) is LONGER than the dense round's (x^5, one 6-term dot
// product) once every lane executes the union of both roles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../babyjubjub-rs_amd/csrc/bjj_device.hpp"
#include "../../babyjubjub-rs_amd/csrc/bjj_constants.inc"
#include "poseidon_plain_t6.inc"
using namespace bjj;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static __constant__ Consts c_K = {
    BJJ_K_A, BJJ_K_D, BJJ_K_F, BJJ_K_FINV_PLAIN, BJJ_K_FINV, BJJ_K_L_R1, BJJ_K_L_R2, BJJ_K_DP, BJJ_K_D2P, BJJ_K_DPINV, BJJ_K_B8X, BJJ_K_B8Y, BJJ_K_TS_G, BJJ_K_HALFQ,
    BJJ_K_ORDER, BJJ_K_ORDER2, BJJ_K_ORDER4, BJJ_K_L, BJJ_K_L2, BJJ_K_L4,
    BJJ_K_POSEIDON_CF, BJJ_K_POSEIDON_KP, BJJ_K_POSEIDON_SP, BJJ_K_POSEIDON_AL, BJJ_K_POSEIDON_M, BJJ_K_POSEIDON_CAB,
    BJJ_K_TS_NEG, BJJ_K_TS_HALF, BJJ_K_TS_HASH};
static __device__ const Fr g_C[PL_ROUNDS * 6] = PL_C;
static __device__ const Fr g_M[36] = PL_M;

// ---- A: the layout that ships (one hash per lane)
__global__ void __launch_bounds__(256, 2) k_lane_per_hash(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    Fr h[5];
    u32 w[8];
#pragma unroll
    for (int j = 0; j < 5; j++) { load_w8(in + i * 160 + j * 32, w); h[j] = fr_to_mont_words(w); }
    Fr r = poseidon5(h, c_K);
    fr_from_mont_words(r, w);
    store_w8(out + i * 32, w);
  }
}

// ---- B: one state element per lane, 6 lanes per hash, dense rounds, cross-lane mix
__device__ __forceinline__ Fr fr_shfl_lane(const Fr& f, int src) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = (u32)__shfl((int)f.v[i], src, 64);
  return r;
}
__global__ void __launch_bounds__(256) k_lane_per_element(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int hsub = lane / 6, e = lane - 6 * hsub;          // hash slot 0..9 (10 = idle lanes 60..63), element 0..5
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int base = 6 * (hsub < 10 ? hsub : 9);              // idle lanes shadow slot 9 (their results are discarded)
  Fr mrow[6];
#pragma unroll
  for (int j = 0; j < 6; j++) mrow[j] = g_M[6 * e + j];    // this lane's matrix row, in registers
#pragma unroll 1
  for (size_t g = wave; g * 10 < n; g += nwaves) {
    const size_t hidx = g * 10 + (hsub < 10 ? hsub : 9);
    const bool live = hsub < 10 && hidx < n;
    const size_t hi = hidx < n ? hidx : n - 1;
    Fr st = fr_zero();
    if (e > 0) { u32 w[8]; load_w8(in + hi * 160 + (e - 1) * 32, w); st = fr_to_mont_words(w); }
#pragma unroll 1
    for (int r = 0; r < PL_ROUNDS; r++) {
      st = fr_add(st, g_C[6 * r + e]);                                        // ark (per-lane constant)
      const bool full = r < 4 || r >= 4 + PL_RP;
      Fr p5 = fr_pow5(st);                                                    // every lane pays for the S-box
      st = fr_select(full || e == 0, p5, st);
      Fr x[6];
#pragma unroll
      for (int j = 0; j < 6; j++) x[j] = fr_shfl_lane(st, base + j);          // the other state words of this hash
      st = fr_dot3<6, 0, 0>(mrow, x, mrow, x, mrow, x);                       // one row of M . state, single reduction
    }
    if (live && e == 0) { u32 w[8]; fr_from_mont_words(st, w); store_w8(out + hidx * 32, w); }
  }
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : (size_t)1 << 20;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::vector<uint8_t> h_in(n * 160);
  uint64_t s = 0x424A4A5F4D534753ULL;
  for (size_t i = 0; i < h_in.size(); i += 8) { s += 0x9E3779B97F4A7C15ULL; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31; memcpy(&h_in[i], &z, 8); }
  for (size_t i = 31; i < h_in.size(); i += 32) h_in[i] &= 0x1f;   // values < 2^253 < r
  uint8_t *d_in, *d_a, *d_b;
  CK(hipMalloc(&d_in, n * 160)); CK(hipMalloc(&d_a, n * 32)); CK(hipMalloc(&d_b, n * 32));
  CK(hipMemcpy(d_in, h_in.data(), n * 160, hipMemcpyHostToDevice));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int occA = 1, occB = 1;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occA, k_lane_per_hash, 256, 0));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occB, k_lane_per_element, 256, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("Poseidon t=6, %zu hashes, %d CUs; resident workgroups per CU: lane-per-hash %d, lane-per-element %d\n", n, prop.multiProcessorCount, occA, occB);
  for (int round = 0; round < 3; round++) {
    for (int which = 0; which < 2; which++) {
      const int grid = prop.multiProcessorCount * (which ? occB : occA);
      auto launch = [&] {
        if (which) hipLaunchKernelGGL(k_lane_per_element, dim3(grid), dim3(256), 0, 0, d_in, n, d_b);
        else hipLaunchKernelGGL(k_lane_per_hash, dim3(grid), dim3(256), 0, 0, d_in, n, d_a);
      };
      launch(); launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; r++) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("round %d  %-38s %8.3f ms per launch  %8.2f M hashes/s\n", round, which ? "B: lane per state element (6 lanes/hash)" : "A: lane per hash (ships)", ms / reps, n / (ms / reps) / 1e3);
    }
  }
  std::vector<uint8_t> a(n * 32), b(n * 32);
  CK(hipMemcpy(a.data(), d_a, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, n * 32, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) bad += memcmp(&a[i * 32], &b[i * 32], 32) != 0;
  printf("outputs of the two layouts differ in %zu of %zu hashes\n", bad, n);
  return bad != 0;
}
