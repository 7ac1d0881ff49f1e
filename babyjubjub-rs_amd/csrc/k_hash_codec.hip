// libbjj_hip.so, kernel unit 3: K3 Poseidon t=6 (src/lib.rs:400-404), the wire-format codec
// (src/lib.rs:166-224, 260-268) and PrivateKey::scalar_key (src/lib.rs:284-302).
#include "k_common.hpp"

// resident 256-lane workgroups per CU the Poseidon kernel is compiled for (A/B knob)
#ifndef BJJ_POSEIDON_MIN_BLOCKS
#define BJJ_POSEIDON_MIN_BLOCKS 2
#endif

// ---------------------------------------------------------------------------
// K3: Poseidon, 5 inputs
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BJJ_BLOCK, BJJ_POSEIDON_MIN_BLOCKS) bjj_k_poseidon5(const uint8_t* __restrict__ in, size_t n,
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

namespace bjjk {
int occ_poseidon5() { return occupancy_of(bjj_k_poseidon5, BJJ_BLOCK); }
int occ_decompress() { return occupancy_of(bjj_k_decompress_points, BJJ_BLOCK); }
hipError_t poseidon5(hipStream_t st, int grid, const uint8_t* in, size_t n, uint8_t* out) {
  BJJ_LAUNCH(bjj_k_poseidon5, dim3(grid), dim3(BJJ_BLOCK), 0, st, in, n, out);
  return hipGetLastError();
}
hipError_t compress_points(hipStream_t st, int grid, const uint8_t* in_xy, size_t n, uint8_t* out) {
  BJJ_LAUNCH(bjj_k_compress_points, dim3(grid), dim3(BJJ_BLOCK), 0, st, in_xy, n, out);
  return hipGetLastError();
}
hipError_t decompress_points(hipStream_t st, int grid, const uint8_t* in, size_t stride, size_t n, uint8_t* out_xy, uint8_t* ok,
                             uint8_t* out_s) {
  BJJ_LAUNCH(bjj_k_decompress_points, dim3(grid), dim3(BJJ_BLOCK), 0, st, in, stride, n, out_xy, ok, out_s);
  return hipGetLastError();
}
hipError_t merge_codec_flags(hipStream_t st, int grid, uint8_t* ok, const uint8_t* f_pk, const uint8_t* f_r, size_t n) {
  BJJ_LAUNCH(bjj_k_merge_codec_flags, dim3(grid), dim3(BJJ_BLOCK), 0, st, ok, f_pk, f_r, n);
  return hipGetLastError();
}
hipError_t scalar_keys(hipStream_t st, int grid, const uint8_t* keys, size_t n, uint8_t* out) {
  BJJ_LAUNCH(bjj_k_scalar_keys, dim3(grid), dim3(BJJ_BLOCK), 0, st, keys, n, out);
  return hipGetLastError();
}
}  // namespace bjjk
