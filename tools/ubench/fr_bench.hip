// Field-operation throughput microbenchmark (developer tool): dependent chains of fr_mul /
// fr_sqr / ext_madd / ext_dbl per lane, at 1..4 waves per SIMD.  Build twice to A/B the
// multiplier: default (one asm statement per column part) and -DBJJ_NO_ASM_COLUMNS (compiler form).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../babyjubjub-rs_amd/csrc/curve.hpp"
using namespace bjj;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_mul(const Fr* in, Fr* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr x = in[t & 1023], y = in[(t + 1) & 1023];
#pragma unroll 1
  for (int i = 0; i < iters; i++) { x = fr_mul(x, y); y = fr_mul(y, x); x = fr_mul(x, y); y = fr_mul(y, x); }
  out[t] = fr_add(x, y);
}
__global__ void __launch_bounds__(256) k_sqr(const Fr* in, Fr* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr x = in[t & 1023];
#pragma unroll 1
  for (int i = 0; i < iters; i++) { x = fr_sqr(x); x = fr_sqr(x); x = fr_sqr(x); x = fr_sqr(x); }
  out[t] = x;
}
__global__ void __launch_bounds__(256) k_addsub(const Fr* in, Fr* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr x = in[t & 1023], y = in[(t + 1) & 1023];
#pragma unroll 1
  for (int i = 0; i < iters; i++) { x = fr_add(x, y); y = fr_sub(y, x); x = fr_reduce_weak(x); y = fr_reduce_weak(y); }
  out[t] = fr_add(x, y);
}
__global__ void __launch_bounds__(256) k_madd(const Fr* in, Fr* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Ext p; p.X = in[t & 1023]; p.Y = in[(t + 1) & 1023]; p.Z = in[(t + 2) & 1023]; p.T = in[(t + 3) & 1023];
  Niels n; n.ymx = in[(t + 4) & 1023]; n.ypx = in[(t + 5) & 1023]; n.t2d = in[(t + 6) & 1023];
#pragma unroll 1
  for (int i = 0; i < iters; i++) p = ext_madd(p, n);
  out[t] = fr_add(fr_add(p.X, p.Y), fr_add(p.Z, p.T));
}
__global__ void __launch_bounds__(256) k_dbl(const Fr* in, Fr* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Ext p; p.X = in[t & 1023]; p.Y = in[(t + 1) & 1023]; p.Z = in[(t + 2) & 1023]; p.T = in[(t + 3) & 1023];
#pragma unroll 1
  for (int i = 0; i < iters; i++) p = ext_dbl<true>(p);
  out[t] = fr_add(fr_add(p.X, p.Y), fr_add(p.Z, p.T));
}
typedef void (*kern_t)(const Fr*, Fr*, int);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  Fr* h = new Fr[1024];
  unsigned s = 12345;
  for (int i = 0; i < 1024; i++) { for (int j = 0; j < 9; j++) { s = s * 1664525u + 1013904223u; h[i].v[j] = (s >> 3) & (j < 8 ? MASK29 : 0x1fffffu); } }
  Fr *din, *dout; CK(hipMalloc(&din, 1024 * sizeof(Fr))); CK(hipMalloc(&dout, (size_t)cus * 8 * 256 * sizeof(Fr)));
  CK(hipMemcpy(din, h, 1024 * sizeof(Fr), hipMemcpyHostToDevice));
  struct { const char* name; kern_t k; double ops; } tests[] = {
    {"fr_mul  (x4 per iter)", k_mul, 4}, {"fr_sqr  (x4 per iter)", k_sqr, 4}, {"add+sub+2 weak reduce", k_addsub, 1},
    {"ext_madd (7M)", k_madd, 1}, {"ext_dbl<T> (4M+4S)", k_dbl, 1}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef BJJ_NO_ASM_COLUMNS
  printf("variant: compiler-scheduled columns\n");
#else
  printf("variant: one asm statement per column part (BJJ_ASM_COLUMNS)\n");
#endif
  printf("%-24s %12s %12s %12s %12s   (nominal-clock cycles per op per SIMD; Gops/s chip-wide at 4 w/SIMD)\n", "op", "1w/SIMD", "2w/SIMD", "3w/SIMD", "4w/SIMD");
  const int iters = 2000;
  for (auto& t : tests) {
    printf("%-24s", t.name);
    double gops = 0;
    for (int wps = 1; wps <= 4; wps++) {
      int blocks = cus * wps;
      hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, din, dout, 10); CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double ops_per_simd = (double)iters * t.ops * wps;
      printf(" %12.1f", best * 1e-3 * prop.clockRate * 1e3 / ops_per_simd);
      gops = (double)iters * t.ops * blocks * 256 / (best * 1e-3) / 1e9;
    }
    printf("   %8.2f\n", gops);
  }
  return 0;
}
