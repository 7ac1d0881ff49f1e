// Developer microbenchmark: how fast can the chip do the fixed-base kernel's memory pattern alone?
// Each lane performs G dependent-free random 128-byte-line reads (7 x dwordx4 = 112 B used, like load_niels)
// from a table of `bytes` bytes and xors them into one word.  Reports ms per 2^20 items x G gathers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
struct alignas(16) U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
template <int Q>
__global__ void __launch_bounds__(512) k_gather(const U4* __restrict__ tab, size_t lines, int G, size_t n, uint32_t* out, int items_per_lane) {
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (size_t i = tid; i < n; i += nth) {
    for (int g = 0; g < G; g++) {
      size_t line = mix(i * 16 + g) % lines;
      const U4* p = tab + line * 8;
#pragma unroll
      for (int q = 0; q < Q; q++) { U4 v = p[q]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  out[tid] = acc;
}
// cooperative pattern: instruction k of a wave reads the 8 full lines of entries 8k..8k+7 (lane L: chunk L%8 of
// entry 8k + L/8), so each load instruction touches 8 distinct lines instead of 64
__global__ void __launch_bounds__(512) k_gather_coop(const U4* __restrict__ tab, size_t lines, int G, size_t n, uint32_t* out) {
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  uint32_t acc = 0;
  for (size_t i = tid; i < n; i += nth) {
    for (int g = 0; g < G; g++) {
      uint32_t line = (uint32_t)(mix(i * 16 + g) % lines);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        uint32_t l2 = __shfl(line, 8 * k + (lane >> 3), 64);
        U4 v = tab[(size_t)l2 * 8 + (lane & 7)];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  out[tid] = acc;
}
int main(int argc, char** argv) {
  double gb = argc > 1 ? atof(argv[1]) : 5.9;
  int G = argc > 2 ? atoi(argv[2]) : 11;
  int wg = argc > 3 ? atoi(argv[3]) : 256;       // workgroups of 512
  int Q = argc > 4 ? atoi(argv[4]) : 7;
  size_t lines = (size_t)(gb * 1e9 / 128), n = 1 << 20;
  U4* tab; uint32_t* out;
  if (hipMalloc((void**)&tab, lines * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(tab, 1, lines * 128);
  hipMalloc((void**)&out, (size_t)wg * 512 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(a);
    for (int it = 0; it < 10; it++) {
      if (Q == 0) hipLaunchKernelGGL(k_gather_coop, dim3(wg), dim3(512), 0, 0, tab, lines, G, n, out);
      else if (Q == 7) hipLaunchKernelGGL(k_gather<7>, dim3(wg), dim3(512), 0, 0, tab, lines, G, n, out, 0);
      else if (Q == 4) hipLaunchKernelGGL(k_gather<4>, dim3(wg), dim3(512), 0, 0, tab, lines, G, n, out, 0);
      else hipLaunchKernelGGL(k_gather<8>, dim3(wg), dim3(512), 0, 0, tab, lines, G, n, out, 0);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("table %.1f GB  G=%d  wg=%d  Q=%d: %.3f ms per 2^20 items  (%.2f TB/s of 128-B lines, %.2f G gathers/s)\n", gb, G, wg, Q, ms / 10,
           (double)n * G * 128 / (ms / 10 * 1e-3) / 1e12, (double)n * G / (ms / 10 * 1e-3) / 1e9);
  }
  return 0;
}
