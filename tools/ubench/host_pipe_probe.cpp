// Developer probe for the host-pointer pipeline (bjj_hip.hip: run_pipelined): (1) raw PCIe copy rates per flavour of pinned
// memory and per copy size, (2) bjj_mul_fixed_base / bjj_eddsa_verify on each flavour of caller memory.
//   hipcc -O2 -o host_pipe_probe host_pipe_probe.cpp -I../../include -L../../babyjubjub-rs_amd/csrc -lbjj_hip -Wl,-rpath,'$ORIGIN/../../babyjubjub-rs_amd/csrc'
//   ./host_pipe_probe [window_bits=23] [calls=5] [what=all|fb_pinned|fb_pageable|v_pinned]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "bjj_hip.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define BK(x) do { int r_ = (x); if (r_) { printf("%s: %d %s\n", #x, r_, bjj_last_error()); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 23, calls = argc > 2 ? atoi(argv[2]) : 5;
  const char* what = argc > 3 ? argv[3] : "all";
  const bool all = !strcmp(what, "all");
  const size_t MB = 1 << 20, n = 1 << 20;
  bjj_ctx* c = nullptr;
  BK(bjj_init(0, W, &c));
  char *d_in, *d_out;
  CK(hipMalloc((void**)&d_in, 192 * MB)); CK(hipMalloc((void**)&d_out, 64 * MB));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  struct Flavour { const char* name; char* p; };
  std::vector<Flavour> fl;
  { char* p; CK(hipHostMalloc((void**)&p, 256 * MB, hipHostMallocDefault)); fl.push_back({"hipHostMalloc(Default)", p}); }
  { char* p; CK(hipHostMalloc((void**)&p, 256 * MB, hipHostMallocPortable)); fl.push_back({"hipHostMalloc(Portable)", p}); }
  { char* p; CK(hipHostMalloc((void**)&p, 256 * MB, hipHostMallocNonCoherent)); fl.push_back({"hipHostMalloc(NonCoherent)", p}); }
  { void* p; BK(bjj_host_alloc(c, 256 * MB, &p)); fl.push_back({"bjj_host_alloc", (char*)p}); }
  { char* p = (char*)aligned_alloc(4096, 256 * MB); memset(p, 1, 256 * MB); BK(bjj_host_register(c, p, 256 * MB)); fl.push_back({"bjj_host_register(malloc)", p}); }
  char* pageable = (char*)aligned_alloc(4096, 256 * MB); memset(pageable, 1, 256 * MB);
  for (auto& f : fl) memset(f.p, 1, 256 * MB);
  if (all || !strcmp(what, "prio")) {   // the same copies on a HIGH-priority stream, and behind a kernel-side event
    int least = 0, greatest = 0; CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t hp; CK(hipStreamCreateWithPriority(&hp, hipStreamNonBlocking, greatest));
    char* big; CK(hipMalloc((void**)&big, 512 * MB));
    for (hipStream_t st : {s1, hp}) {
      for (char* src : {d_out, big + 200 * MB}) {
        double best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
          double t0 = now();
          for (size_t o = 0; o < 64 * MB; o += 16 * MB) CK(hipMemcpyAsync(fl[3].p + o, src + o, 16 * MB, hipMemcpyDeviceToHost, st));
          CK(hipStreamSynchronize(st));
          best = std::min(best, now() - t0);
        }
        printf("D2H 64 MB in 16 MB pieces, %s stream, source %s: %.3f ms (%.1f GB/s)\n", st == hp ? "HIGH-priority" : "normal", src == d_out ? "own 64 MB allocation" : "inside a 512 MB allocation",
               best * 1e3, 64 * MB / best / 1e9);
      }
    }
  }
  if (all) {
    for (auto& f : fl) {
      for (size_t piece : {64 * MB, 16 * MB, 4 * MB, 1 * MB}) {
        double best_d = 1e9, best_h = 1e9, best_b = 1e9;
        for (int rep = 0; rep < 4; rep++) {
          double t0 = now();
          for (size_t o = 0; o < 64 * MB; o += piece) CK(hipMemcpyAsync(f.p + o, d_out + o, piece, hipMemcpyDeviceToHost, s1));
          CK(hipStreamSynchronize(s1));
          double t1 = now();
          for (size_t o = 0; o < 32 * MB; o += piece < 32 * MB ? piece : 32 * MB) CK(hipMemcpyAsync(d_in + o, f.p + 64 * MB + o, piece < 32 * MB ? piece : 32 * MB, hipMemcpyHostToDevice, s2));
          CK(hipStreamSynchronize(s2));
          double t2 = now();
          for (size_t o = 0; o < 64 * MB; o += piece) {   // both directions at once
            CK(hipMemcpyAsync(f.p + o, d_out + o, piece, hipMemcpyDeviceToHost, s1));
            if (o < 32 * MB) CK(hipMemcpyAsync(d_in + o, f.p + 64 * MB + o, piece < 32 * MB ? piece : 32 * MB, hipMemcpyHostToDevice, s2));
          }
          CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
          double t3 = now();
          best_d = std::min(best_d, t1 - t0); best_h = std::min(best_h, t2 - t1); best_b = std::min(best_b, t3 - t2);
        }
        printf("%-28s pieces of %3zu MB: D2H 64 MB %.3f ms (%.1f GB/s)  H2D 32 MB %.3f ms (%.1f GB/s)  both at once %.3f ms\n", f.name, piece / MB,
               best_d * 1e3, 64 * MB / best_d / 1e9, best_h * 1e3, 32 * MB / best_h / 1e9, best_b * 1e3);
      }
    }
  }
  // the library calls
  auto fb = [&](const char* name, char* p) {
    uint8_t* sc = (uint8_t*)p; uint8_t* out = (uint8_t*)p + 64 * MB;
    for (size_t i = 0; i < n * 32; i++) sc[i] = (uint8_t)(i * 2654435761u >> 13);
    for (size_t i = 0; i < n; i++) sc[i * 32 + 31] &= 0x3f;
    double best = 1e9, tot = 0;
    BK(bjj_mul_fixed_base(c, sc, n, out));
    for (int k = 0; k < calls; k++) { double t0 = now(); BK(bjj_mul_fixed_base(c, sc, n, out)); double dt = now() - t0; best = std::min(best, dt); tot += dt; }
    bjj_info inf; inf.struct_size = sizeof(inf); BK(bjj_get_info(c, &inf));
    printf("bjj_mul_fixed_base 2^20, %-28s best %.3f ms (%.0f M/s), mean %.3f ms; direct %u staged %u chunks %u\n", name, best * 1e3, n / best / 1e6, tot / calls * 1e3,
           inf.last_host_direct_arrays, inf.last_host_staged_arrays, inf.last_host_chunks);
  };
  auto vf = [&](const char* name, char* p) {
    uint8_t* pk = (uint8_t*)p; uint8_t* sc = pk + 64 * MB; uint8_t* msg = pk + 96 * MB; uint8_t* ok = pk + 128 * MB; uint8_t* tmp = pk + 160 * MB;
    for (size_t i = 0; i < n * 32; i++) { sc[i] = (uint8_t)(i * 2654435761u >> 13); msg[i] = (uint8_t)(i * 40503u >> 7); }
    for (size_t i = 0; i < n; i++) { sc[i * 32 + 31] &= 0x3f; msg[i * 32 + 31] &= 0x1f; }
    BK(bjj_mul_fixed_base(c, sc, n, tmp)); memcpy(pk, tmp, 64 * MB);
    double best = 1e9;
    BK(bjj_eddsa_verify(c, pk, pk, sc, msg, n, ok));
    for (int k = 0; k < std::max(2, calls / 2); k++) { double t0 = now(); BK(bjj_eddsa_verify(c, pk, pk, sc, msg, n, ok)); best = std::min(best, now() - t0); }
    printf("bjj_eddsa_verify 2^20,   %-28s best %.3f ms (%.1f M/s)\n", name, best * 1e3, n / best / 1e6);
  };
  if (all || !strcmp(what, "fb_pinned")) for (auto& f : fl) { if (all || !strcmp(f.name, "bjj_host_alloc")) fb(f.name, f.p); }
  if (all || !strcmp(what, "fb_pageable")) fb("pageable (aligned_alloc)", pageable);
  if (all || !strcmp(what, "v_pinned")) vf("bjj_host_alloc", fl[3].p);
  if (all) vf("pageable (aligned_alloc)", pageable);
  bjj_free(c);
  return 0;
}
