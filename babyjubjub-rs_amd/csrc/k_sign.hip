// libbjj_hip.so, kernel unit 5: signer side, PrivateKey::sign / sign_schnorr (src/lib.rs:308-361).
// Every kernel exists in two forms: the fast one gathers from the context's big fixed-base table with the digits of the nonce
// and of the scalar key as indices (addresses depend on secrets); the constant-time one (CT, bjj_set_signer_constant_time)
// goes through the scanning policy over the context's small 4-bit table -- every entry of a window is read, the digit only
// selects -- and costs ~8x more per fixed-base multiplication (62 additions instead of 8).
#include "k_common.hpp"

// C64: the signature leaves as Signature::compress (src/lib.rs:245-258): 64 bytes at out_r + i * 64 = Point::compress(R) || s as
// 32 little-endian bytes (s < l: the reference's min(len, 32) copy never truncates); out_s is not used.
template <bool CT, bool C64 = false>
__device__ __forceinline__ void sign_body(const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ keys,
                                          const uint8_t* __restrict__ msgs, size_t n, uint8_t* __restrict__ out_r,
                                          uint8_t* __restrict__ out_s, uint8_t* __restrict__ ok) {
  __shared__ __attribute__((aligned(16))) u32 stage[CT ? 4 : (BJJ_BLOCK / 64) * FB_STAGE_WORDS];
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i - lane < n; i += nthreads) {  // wave-uniform trip count
    const size_t ic = i < n ? i : n - 1;
    u32 k[8], m[8], rx[8], ry[8], s[8];
    load_w8(keys + ic * 32, k); load_w8(msgs + ic * 32, m);
    bool good;
    if constexpr (CT) {
      good = sign_item(k, m, GatherScan{table, (u32)fixed_stride(W)}, W, nwin, rx, ry, s, c_K);
    } else {
      const GatherCoopLds<1> fb = {table, stage + (threadIdx.x >> 6) * FB_STAGE_WORDS, lane};
      good = sign_item(k, m, fb, W, nwin, rx, ry, s, c_K);
    }
    if (i < n) {
#pragma unroll
      for (int j = 0; j < 8; j++) { rx[j] = good ? rx[j] : 0u; ry[j] = good ? ry[j] : 0u; s[j] = good ? s[j] : 0u; }
      if constexpr (C64) {
        u32 c[8];
        compress_item(rx, ry, c, c_K);
        store_w8(out_r + i * 64, c); store_w8(out_r + i * 64 + 32, s);
      } else {
        store_w8(out_r + i * 64, rx); store_w8(out_r + i * 64 + 32, ry); store_w8(out_s + i * 32, s);
      }
      ok[i] = good ? 1 : 0;
    }
  }
}
#define SIGN_ARGS const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ keys, const uint8_t* __restrict__ msgs, \
                  size_t n, uint8_t* __restrict__ out_r, uint8_t* __restrict__ out_s, uint8_t* __restrict__ ok
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_sign(SIGN_ARGS) { sign_body<false>(table, W, nwin, keys, msgs, n, out_r, out_s, ok); }
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_sign_ct(SIGN_ARGS) { sign_body<true>(table, W, nwin, keys, msgs, n, out_r, out_s, ok); }
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_sign_c64(SIGN_ARGS) { sign_body<false, true>(table, W, nwin, keys, msgs, n, out_r, out_s, ok); }
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_sign_ct_c64(SIGN_ARGS) { sign_body<true, true>(table, W, nwin, keys, msgs, n, out_r, out_s, ok); }

// PrivateKey::sign_schnorr (src/lib.rs:344-361) with caller-supplied 1024-bit nonces (128 B each); s is the
// reference's unreduced integer k + scalar_key*h in a 160-byte little-endian record.
template <bool CT>
__device__ __forceinline__ void sign_schnorr_body(const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ keys,
                                                  const uint8_t* __restrict__ msgs, const uint8_t* __restrict__ nonces, size_t n,
                                                  uint8_t* __restrict__ out_r, uint8_t* __restrict__ out_s, uint8_t* __restrict__ ok) {
  __shared__ __attribute__((aligned(16))) u32 stage[CT ? 4 : (BJJ_BLOCK / 64) * FB_STAGE_WORDS];
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i - lane < n; i += nthreads) {  // wave-uniform trip count
    const size_t ic = i < n ? i : n - 1;
    u32 k[8], m[8], rx[8], ry[8], kn[SCHNORR_K_WORDS], s[SCHNORR_S_WORDS];
    load_w8(keys + ic * 32, k); load_w8(msgs + ic * 32, m);
#pragma unroll
    for (int j = 0; j < SCHNORR_K_WORDS / 8; j++) load_w8(nonces + ic * (SCHNORR_K_WORDS * 4) + j * 32, kn + 8 * j);
    bool good;
    if constexpr (CT) {
      good = sign_schnorr_item(k, m, kn, GatherScan{table, (u32)fixed_stride(W)}, W, nwin, rx, ry, s, c_K);
    } else {
      const GatherCoopLds<1> fb = {table, stage + (threadIdx.x >> 6) * FB_STAGE_WORDS, lane};
      good = sign_schnorr_item(k, m, kn, fb, W, nwin, rx, ry, s, c_K);
    }
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
#define SCHNORR_ARGS const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ keys, const uint8_t* __restrict__ msgs, \
                     const uint8_t* __restrict__ nonces, size_t n, uint8_t* __restrict__ out_r, uint8_t* __restrict__ out_s,            \
                     uint8_t* __restrict__ ok
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_sign_schnorr(SCHNORR_ARGS) {
  sign_schnorr_body<false>(table, W, nwin, keys, msgs, nonces, n, out_r, out_s, ok);
}
__global__ void __launch_bounds__(BJJ_BLOCK, 2) bjj_k_sign_schnorr_ct(SCHNORR_ARGS) {
  sign_schnorr_body<true>(table, W, nwin, keys, msgs, nonces, n, out_r, out_s, ok);
}

namespace bjjk {
int occ_sign() { const int a = occupancy_of(bjj_k_sign, BJJ_BLOCK), b = occupancy_of(bjj_k_sign_c64, BJJ_BLOCK); return a < b ? a : b; }
int occ_sign_schnorr() { return occupancy_of(bjj_k_sign_schnorr, BJJ_BLOCK); }
int occ_sign_ct() { const int a = occupancy_of(bjj_k_sign_ct, BJJ_BLOCK), b = occupancy_of(bjj_k_sign_ct_c64, BJJ_BLOCK); return a < b ? a : b; }
int occ_sign_schnorr_ct() { return occupancy_of(bjj_k_sign_schnorr_ct, BJJ_BLOCK); }
// out_s == nullptr: the compressed form (out_r = 64-byte Signature::compress records)
hipError_t sign(hipStream_t st, int grid, const u32* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs, size_t n,
                uint8_t* out_r, uint8_t* out_s, uint8_t* ok) {
  if (out_s) BJJ_LAUNCH(bjj_k_sign, dim3(grid), dim3(BJJ_BLOCK), 0, st, table, W, nwin, keys, msgs, n, out_r, out_s, ok);
  else BJJ_LAUNCH(bjj_k_sign_c64, dim3(grid), dim3(BJJ_BLOCK), 0, st, table, W, nwin, keys, msgs, n, out_r, out_s, ok);
  return hipGetLastError();
}
hipError_t sign_ct(hipStream_t st, int grid, const u32* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs, size_t n,
                   uint8_t* out_r, uint8_t* out_s, uint8_t* ok) {
  if (out_s) BJJ_LAUNCH(bjj_k_sign_ct, dim3(grid), dim3(BJJ_BLOCK), 0, st, table, W, nwin, keys, msgs, n, out_r, out_s, ok);
  else BJJ_LAUNCH(bjj_k_sign_ct_c64, dim3(grid), dim3(BJJ_BLOCK), 0, st, table, W, nwin, keys, msgs, n, out_r, out_s, ok);
  return hipGetLastError();
}
hipError_t sign_schnorr(hipStream_t st, int grid, const u32* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs,
                        const uint8_t* nonces, size_t n, uint8_t* out_r, uint8_t* out_s, uint8_t* ok) {
  BJJ_LAUNCH(bjj_k_sign_schnorr, dim3(grid), dim3(BJJ_BLOCK), 0, st, table, W, nwin, keys, msgs, nonces, n, out_r, out_s, ok);
  return hipGetLastError();
}
hipError_t sign_schnorr_ct(hipStream_t st, int grid, const u32* table, int W, int nwin, const uint8_t* keys, const uint8_t* msgs,
                           const uint8_t* nonces, size_t n, uint8_t* out_r, uint8_t* out_s, uint8_t* ok) {
  BJJ_LAUNCH(bjj_k_sign_schnorr_ct, dim3(grid), dim3(BJJ_BLOCK), 0, st, table, W, nwin, keys, msgs, nonces, n, out_r, out_s, ok);
  return hipGetLastError();
}
}  // namespace bjjk
