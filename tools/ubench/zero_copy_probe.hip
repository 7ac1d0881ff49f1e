// Developer probe: can a kernel's own stores / loads move a result array over PCIe as fast as the copy engines?
// (the idea left open in profiles/r05_host_pipeline.txt: K1's epilogue storing its 64-byte records straight into mapped host memory)
//   hipcc -O2 --offload-arch=gfx950 -o zero_copy_probe zero_copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
typedef uint4 U4;
// pattern 0: lane L of a wave writes the 16-byte pieces k = 0..3 of ITS OWN 64-byte record (4 store instructions, each touching
//            64 different 64-byte lines partially) -- what epilogue_finish does today
// pattern 1: the same bytes, transposed: store instruction k of a wave writes 1 KB contiguous (lane L -> bytes 16 L of block k)
__global__ void wr(U4* dst, size_t recs, int pattern) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i - lane < recs; i += nthreads) {
    const size_t wave0 = i - lane;           // first record of this wave's 64
    U4 v = {(unsigned)i, 1u, 2u, 3u};
    if (i < recs) {
      if (pattern == 0) { for (int k = 0; k < 4; k++) dst[i * 4 + k] = v; }
      else { for (int k = 0; k < 4; k++) dst[wave0 * 4 + k * 64 + lane] = v; }
    }
  }
}
__global__ void rd(const U4* src, size_t n16, unsigned* sink) {   // 32-byte scalars read by the kernel itself
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += nthreads) acc ^= src[i].x;
  if (acc == 0x12345678u) *sink = acc;
}
int main() {
  const size_t recs = 1 << 20, bytes = recs * 64;
  U4 *h, *d; unsigned* sink;
  CK(hipHostMalloc((void**)&h, bytes, hipHostMallocPortable)); memset(h, 0, bytes);
  CK(hipMalloc((void**)&d, bytes)); CK(hipMalloc((void**)&sink, 4));
  U4* hd = nullptr; CK(hipHostGetDevicePointer((void**)&hd, h, 0));
  for (int grid : {256, 1024, 4096}) {
    for (int pattern : {0, 1}) {
      for (U4* target : {hd, d}) {
        double best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
          CK(hipDeviceSynchronize());
          double t0 = now();
          hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, target, recs, pattern);
          CK(hipDeviceSynchronize());
          best = std::min(best, now() - t0);
        }
        printf("write 64 MB of 64-byte records to %-12s pattern %d (%s), grid %4d x 256: %.3f ms (%.1f GB/s)\n", target == hd ? "HOST (mapped)" : "HBM", pattern,
               pattern ? "1 KB contiguous per store" : "16 B per lane, 64 B apart", grid, best * 1e3, bytes / best / 1e9);
      }
    }
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipDeviceSynchronize());
      double t0 = now();
      hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, hd, bytes / 2 / 16, sink);
      CK(hipDeviceSynchronize());
      best = std::min(best, now() - t0);
    }
    printf("read 32 MB from HOST (mapped), 16 B per lane contiguous, grid %4d x 256: %.3f ms (%.1f GB/s)\n", grid, best * 1e3, bytes / 2 / best / 1e9);
  }
  return 0;
}
