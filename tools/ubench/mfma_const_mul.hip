// Microbenchmark (developer tool): can the MATRIX cores take the constant half of a Montgomery
// product off the VALU?  Half of every fr_mul is m x N with N = r a compile-time constant (81 of the 171 multiply-adds),
// i.e. a batch-of-lanes x Toeplitz(constant) contraction -- the one place on this path where MFMA is even thinkable.
//
// What is built: for 64 items per wave (one per lane, as every kernel of the library holds them), the 18-column product
//   P = m * N,  m = 9 x 29-bit limbs per lane, N = r
// (a) on the VALU: 81 v_mad_u64_u32 with constant operands into 64-bit column accumulators (the shipped form), and
// (b) on the matrix cores with v_mfma_i32_16x16x64_i8: m cut into 38 slices of 7 bits (i8 operands are signed), N likewise,
//     out[c] = sum_k m_k n_(c-k) for c = 0..74 as OUT(75 x items) = TOEP(75 x 38) x SLICES(38 x items).
// The data stays where the kernels have it (an item's 38 slices live in ITS lane).  The B operand of the MFMA wants the 64
// k-values of one matrix column spread over the four lanes j, j+16, j+32, j+48; instead of shuffling slices between lanes,
// the constant A operand is zero outside ONE of the four k-blocks: an MFMA then multiplies 16 output columns x 16 slices for
// the 16 items of lane group q -- a quarter of the instruction's work is useful.  Per wave and product: 4 lane groups x 11
// non-zero (column tile, slice quad) pairs = 44 MFMAs, accumulating into 20 result quads.  (With a transposed data layout it
// would be 20 MFMAs -- timed as well, issue only, as the bound no layout change could beat.)
// What the MFMA form still owes and is NOT charged in the timing (all of it VALU work): recombining the 75 seven-bit-spaced
// i32 columns into 29-bit limbs (~75 shift-adds + carries), gathering an item's columns from the 4 lanes x 5 tiles they
// land in (C/D layout: lane l holds rows 4(l/16)..+3 of column l%16), and -- for a whole Montgomery product -- the same
// again for m = (T mod R) N' mod R.  The slicing of m (38 bit-field extracts + packing) IS charged.
// Correctness: the MFMA columns are recombined through LDS (untimed) and compared limb for limb with the VALU columns on
// 10^7 random m; the timed kernels keep a data dependence from each product into the next m.
//
// Output: products per second for (a), (b), (b) issue-only with 44 and with 20 MFMAs, and for the two full-multiplier
// shapes (c) 162 VALU multiply-adds  vs  (d) 81 VALU multiply-adds + slicing + 44 MFMAs in the same instruction stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../babyjubjub-rs_amd/csrc/fr.hpp"
using namespace bjj;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NSL = 38;    // 7-bit slices of a 261-bit value
constexpr int NCOL = 75;   // columns of the slice product

struct SliceTab { signed char n[NSL]; };
// 7-bit slices of r (host, once)
static SliceTab make_n_slices() {
  const u32 N[NL] = {BJJ_N0, BJJ_N1, BJJ_N2, BJJ_N3, BJJ_N4, BJJ_N5, BJJ_N6, BJJ_N7, BJJ_N8};
  SliceTab t;
  for (int k = 0; k < NSL; k++) {
    const int bit = 7 * k, li = bit / 29, off = bit % 29;
    u64 two = (u64)N[li] | ((u64)(li + 1 < NL ? N[li + 1] : 0u) << 29);
    t.n[k] = (signed char)((two >> off) & 127u);
  }
  return t;
}
__constant__ SliceTab c_n;

__device__ __forceinline__ Fr rnd_m(u32 seed) {   // random 9 x 29-bit limbs (top limb 26 bits: an N-form value)
  Fr m;
  u32 s = seed * 2654435761u + 12345u;
#pragma unroll
  for (int i = 0; i < NL; i++) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; m.v[i] = s & (i < NL - 1 ? MASK29 : 0x3ffffffu); }
  return m;
}

// ---- (a) VALU: 18 columns of m x N, 81 multiply-adds with constant operands ----
__device__ __forceinline__ void valu_columns(const Fr& m, u64 col[2 * NL - 1]) {
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) { const int j = k - i; if (j >= 0 && j < NL) acc += (u64)m.v[i] * fr_modlimb(j); }
    col[k] = acc;
  }
}
// 18 raw columns -> 18 canonical 29-bit limbs (the value is < 2^516)
__device__ __forceinline__ void normalise(const u64 col[2 * NL - 1], u32 limb[2 * NL]) {
  u64 c = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) { c += col[k]; limb[k] = (u32)c & MASK29; c >>= 29; }
  limb[2 * NL - 1] = (u32)c;
}

// ---- (b) MFMA ----
// slices of m packed 4 per word, 16 per quad: quad s holds slices 16 s .. 16 s + 15 (the last quad is padded with zeros)
__device__ __forceinline__ void slice_m(const Fr& m, v4i b[3]) {
  u32 w[12];
#pragma unroll
  for (int i = 0; i < 12; i++) w[i] = 0;
#pragma unroll
  for (int k = 0; k < NSL; k++) {
    const int bit = 7 * k, li = bit / 29, off = bit % 29;
    u32 sl = m.v[li] >> off;
    if (off > 22 && li + 1 < NL) sl |= m.v[li + 1] << (29 - off);
    sl &= 127u;
    w[k >> 2] |= sl << (8 * (k & 3));
  }
#pragma unroll
  for (int s = 0; s < 3; s++) b[s] = v4i{(int)w[4 * s], (int)w[4 * s + 1], (int)w[4 * s + 2], (int)w[4 * s + 3]};
}
// A operand for lane l, column-tile minus slice-quad d = ct - s (0..3): byte u = n[16 d + (l % 16) - u], zero outside 0..37
__device__ __forceinline__ v4i toeplitz_quad(int lane, int d) {
  u32 w[4] = {0, 0, 0, 0};
  const int i = lane & 15;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int idx = 16 * d + i - u;
    const u32 v = (idx >= 0 && idx < NSL) ? (u32)(unsigned char)c_n.n[idx] : 0u;
    w[u >> 2] |= v << (8 * (u & 3));
  }
  return v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
}
struct AOps { v4i a[4][4]; };   // [lane group q][d]: the Toeplitz quad where l / 16 == q, zero elsewhere
__device__ __forceinline__ AOps make_a(int lane) {
  AOps o;
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const v4i t = toeplitz_quad(lane, d);
      o.a[q][d] = (lane >> 4) == q ? t : v4i{0, 0, 0, 0};
    }
  return o;
}
// 44 MFMAs: D[q][ct] = sum_s A[q][ct - s] x B[s] over the (ct, s) pairs whose Toeplitz block is not all zero
__device__ __forceinline__ void mfma_columns(const AOps& A, const v4i b[3], v4i D[4][5]) {
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int ct = 0; ct < 5; ct++) {
      v4i acc = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 3; s++) {
        const int d = ct - s;
        if (d >= 0 && d <= 3 && !(ct == 4 && s == 0)) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A.a[q][d], b[s], acc, 0, 0, 0);
      }
      D[q][ct] = acc;
    }
}

__global__ void __launch_bounds__(256) k_valu(u32* out, int iters) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr m = rnd_m(t);
  u32 sink = 0;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    u64 col[2 * NL - 1];
    valu_columns(m, col);
    u32 x = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) x ^= (u32)col[k] ^ (u32)(col[k] >> 32);
    m.v[it % NL] = (m.v[it % NL] ^ x) & MASK29;   // the next product depends on this one
    sink ^= x;
  }
  out[t] = sink;
}
template <int MODE>   // 0: slicing + 44 MFMAs; 1: 44 MFMAs, operands fixed; 2: 20 MFMAs, operands fixed
__global__ void __launch_bounds__(256) k_mfma(u32* out, int iters) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  Fr m = rnd_m(t);
  const AOps A = make_a(lane);
  v4i b[3];
  slice_m(m, b);
  u32 sink = 0;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) slice_m(m, b);
    v4i D[4][5];
    if (MODE == 2) {   // the MFMA count of an ideal transposed layout: 4 lane groups x 5 column tiles, K fully used
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int ct = 0; ct < 5; ct++) D[q][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A.a[q][ct & 3], b[ct % 3], v4i{0, 0, 0, 0}, 0, 0, 0);
    } else {
      mfma_columns(A, b, D);
    }
    u32 x = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int ct = 0; ct < 5; ct++) x ^= (u32)(D[q][ct].x ^ D[q][ct].y ^ D[q][ct].z ^ D[q][ct].w);
    if (MODE == 0) m.v[it % NL] = (m.v[it % NL] ^ x) & MASK29;
    else b[it % 3].x ^= (int)(x & 0x01010101u);        // keeps the chain (operand bytes stay below 128)
    sink ^= x;
  }
  out[t] = sink;
}
// full-multiplier shapes: (c) 162 VALU multiply-adds; (d) 81 VALU multiply-adds + slicing + 44 MFMAs
__global__ void __launch_bounds__(256) k_full_valu(u32* out, int iters) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr a = rnd_m(t), bb = rnd_m(t ^ 0x5555u);
  u32 sink = 0;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    u64 col[2 * NL - 1], col2[2 * NL - 1];
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      u64 acc = 0;
#pragma unroll
      for (int i = 0; i < NL; i++) { const int j = k - i; if (j >= 0 && j < NL) acc += (u64)a.v[i] * bb.v[j]; }
      col[k] = acc;
    }
    Fr m;
#pragma unroll
    for (int i = 0; i < NL; i++) m.v[i] = (u32)col[i] & MASK29;
    valu_columns(m, col2);
    u32 x = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) x ^= (u32)(col[k] + col2[k]) ^ (u32)((col[k] + col2[k]) >> 32);
    a.v[it % NL] = (a.v[it % NL] ^ x) & MASK29;
    sink ^= x;
  }
  out[t] = sink;
}
__global__ void __launch_bounds__(256) k_full_mixed(u32* out, int iters) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  Fr a = rnd_m(t), bb = rnd_m(t ^ 0x5555u);
  const AOps A = make_a(lane);
  u32 sink = 0;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    u64 col[2 * NL - 1];
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      u64 acc = 0;
#pragma unroll
      for (int i = 0; i < NL; i++) { const int j = k - i; if (j >= 0 && j < NL) acc += (u64)a.v[i] * bb.v[j]; }
      col[k] = acc;
    }
    Fr m;
#pragma unroll
    for (int i = 0; i < NL; i++) m.v[i] = (u32)col[i] & MASK29;
    v4i b[3];
    slice_m(m, b);
    v4i D[4][5];
    mfma_columns(A, b, D);
    u32 x = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) x ^= (u32)col[k] ^ (u32)(col[k] >> 32);
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int ct = 0; ct < 5; ct++) x ^= (u32)(D[q][ct].x ^ D[q][ct].y ^ D[q][ct].z ^ D[q][ct].w);
    a.v[it % NL] = (a.v[it % NL] ^ x) & MASK29;
    sink ^= x;
  }
  out[t] = sink;
}

// ---- correctness: MFMA columns recombined through LDS (one wave per block), compared with the VALU columns ----
__global__ void __launch_bounds__(64) k_check(unsigned long long* bad, u32 seed0, int rounds) {
  __shared__ int cols[64][80];
  const int lane = threadIdx.x;
  const AOps A = make_a(lane);
  unsigned long long mine = 0;
#pragma unroll 1
  for (int r = 0; r < rounds; r++) {
    const Fr m = rnd_m(seed0 + (u32)((blockIdx.x * rounds + r) * 64 + lane));
    v4i b[3];
    slice_m(m, b);
    v4i D[4][5];
    mfma_columns(A, b, D);
    // C/D layout of the 16x16 family: lane l holds rows 4 (l / 16) + {0..3} of column l % 16; with A confined to k-block q the
    // column belongs to item (l % 16) + 16 q, the rows are slice-product columns 16 ct + 4 (l / 16) + {0..3}
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int ct = 0; ct < 5; ct++) {
        const int item = (lane & 15) + 16 * q, c0 = 16 * ct + 4 * (lane >> 4);
        cols[item][c0 + 0] = D[q][ct].x; cols[item][c0 + 1] = D[q][ct].y; cols[item][c0 + 2] = D[q][ct].z; cols[item][c0 + 3] = D[q][ct].w;
      }
    __syncthreads();
    // sum_c cols[c] 2^(7 c) into 29-bit limb buckets, then carries
    u64 bucket[2 * NL];
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) bucket[i] = 0;
#pragma unroll
    for (int c = 0; c < NCOL; c++) {
      const int bit = 7 * c, li = bit / 29, off = bit % 29;
      bucket[li] += (u64)(u32)cols[lane][c] << off;
    }
    u32 got[2 * NL], want[2 * NL];
    u64 cy = 0;
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) { cy += bucket[i]; got[i] = (i < 2 * NL - 1) ? ((u32)cy & MASK29) : (u32)cy; cy >>= 29; }
    u64 col[2 * NL - 1];
    valu_columns(m, col);
    normalise(col, want);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) ok = ok && got[i] == want[i];
    mine += ok ? 0 : 1;
  }
  if (mine) atomicAdd(bad, mine);
}

typedef void (*kern_t)(u32*, int);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  SliceTab t = make_n_slices();
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_n), &t, sizeof(t)));
  // ---- correctness on 10^7 products
  unsigned long long* d_bad; CK(hipMalloc(&d_bad, 8)); CK(hipMemset(d_bad, 0, 8));
  const int rounds = 77, blocks = 2048;   // 2048 x 77 x 64 = 10 092 544 products
  hipLaunchKernelGGL(k_check, dim3(blocks), dim3(64), 0, 0, d_bad, 0x1234567u, rounds);
  CK(hipDeviceSynchronize());
  unsigned long long h_bad = 0; CK(hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost));
  printf("correctness: m x N through v_mfma_i32_16x16x64_i8 (7-bit slices, recombined) vs 81 v_mad_u64_u32: %llu mismatches in %d products\n",
         h_bad, blocks * rounds * 64);
  // ---- timing
  u32* d_out; CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(u32)));
  struct { const char* name; kern_t k; } tests[] = {
      {"(a) VALU   81 mad (m x N, constant operands)", k_valu},
      {"(b) MFMA   slicing + 44 mfma_i32_16x16x64_i8", k_mfma<0>},
      {"    MFMA   44 mfma, operands fixed (issue only)", k_mfma<1>},
      {"    MFMA   20 mfma, operands fixed (ideal layout bound)", k_mfma<2>},
      {"(c) VALU   81 mad a x b + 81 mad m x N", k_full_valu},
      {"(d) mixed  81 mad a x b + slicing + 44 mfma", k_full_mixed}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  printf("%-58s %14s %14s %14s   (G products/s chip-wide; nominal-clock cycles per wave-product per SIMD at 2 w/SIMD)\n", "variant", "1 w/SIMD", "2 w/SIMD", "cycles @2w");
  for (auto& tst : tests) {
    printf("%-58s", tst.name);
    double cyc = 0;
    for (int wps = 1; wps <= 2; wps++) {
      const int nb = cus * wps;
      hipLaunchKernelGGL(tst.k, dim3(nb), dim3(256), 0, 0, d_out, 10); CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(tst.k, dim3(nb), dim3(256), 0, 0, d_out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf(" %14.2f", (double)iters * nb * 256 / (best * 1e-3) / 1e9);
      cyc = best * 1e-3 * prop.clockRate * 1e3 / ((double)iters * wps);
    }
    printf(" %14.1f\n", cyc);
  }
  return h_bad ? 2 : 0;
}
