// Instruction-rate microbenchmark for the integer/FP64 VALU ops a 256-bit
// Montgomery multiplier can be built from on gfx950.  Standalone tool (not part
// of the product): hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
// Prints cycles per wave-instruction per SIMD at 1/2/4/8 waves per SIMD,
// for independent (4 chains) and dependent (1 chain) streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// Each kernel: ITER loop iterations of 64 instructions of the op under test.
#define KERNEL_BEGIN(name) \
  __global__ void __launch_bounds__(256) name(u32* out, int iters, u32 s0, u32 s1) { \
    u32 a = threadIdx.x * 2654435761u + s0, b = threadIdx.x * 40503u + s1; \
    u64 q0 = a, q1 = b, q2 = a ^ b, q3 = a + b; \
    u32 r0 = a, r1 = b, r2 = a ^ b, r3 = a + b; \
    double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, da = 1.0000001, db = 0.5; \
    for (int it = 0; it < iters; ++it) {
#define KERNEL_END \
    } \
    u32 acc = r0 ^ r1 ^ r2 ^ r3 ^ (u32)q0 ^ (u32)q1 ^ (u32)q2 ^ (u32)q3 ^ (u32)(q0 >> 32) ^ (u32)(q1>>32) ^ (u32)(q2>>32) ^ (u32)(q3>>32); \
    acc ^= (u32)(d0 + d1 + d2 + d3); \
    if (acc == 0x12345678u) out[threadIdx.x] = acc; \
  }

// ---- 4 independent chains -------------------------------------------------
KERNEL_BEGIN(k_mad64_ind)
  REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a), "v"(b) : "vcc");)
KERNEL_END
KERNEL_BEGIN(k_mad64_dep)
  REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(a), "v"(b) : "vcc");)
KERNEL_END
// mad64 + addc (the 96-bit accumulate pair)
KERNEL_BEGIN(k_mad64_addc)
  REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(q0), "+v"(q1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "vcc");)
KERNEL_END
KERNEL_BEGIN(k_mullo_ind)
  REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_mulhi_ind)
  REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_mad24_ind)
  REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_mulhi24_ind)
  REP16(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_mad_u32_u16_ind)
  REP16(asm volatile("v_mad_u32_u16 %0, %0, %4, %1\n v_mad_u32_u16 %1, %1, %4, %2\n v_mad_u32_u16 %2, %2, %4, %3\n v_mad_u32_u16 %3, %3, %4, %0" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_add_ind)
  REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_addc_chain)
  REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a) : "vcc");)
KERNEL_END
KERNEL_BEGIN(k_add3_ind)
  REP16(asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_lshladd64_ind)
  REP16(asm volatile("v_lshl_add_u64 %0, %1, 0, %0\n v_lshl_add_u64 %1, %2, 0, %1\n v_lshl_add_u64 %2, %3, 0, %2\n v_lshl_add_u64 %3, %0, 0, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
KERNEL_END
KERNEL_BEGIN(k_lshr64_ind)
  REP16(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
KERNEL_END
KERNEL_BEGIN(k_mov_ind)
  REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)
KERNEL_END
KERNEL_BEGIN(k_alignbit_ind)
  REP16(asm volatile("v_alignbit_b32 %0, %1, %0, 29\n v_alignbit_b32 %1, %2, %1, 29\n v_alignbit_b32 %2, %3, %2, 29\n v_alignbit_b32 %3, %0, %3, 29" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)
KERNEL_END
KERNEL_BEGIN(k_fma64_ind)
  REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db));)
KERNEL_END
KERNEL_BEGIN(k_fma64_dep)
  REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(da), "v"(db));)
KERNEL_END
KERNEL_BEGIN(k_mul64f_ind)
  REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da));)
KERNEL_END
KERNEL_BEGIN(k_add64f_ind)
  REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da));)
KERNEL_END
KERNEL_BEGIN(k_fma32_ind)
  REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_dot4_ind)
  REP16(asm volatile("v_dot4_u32_u8 %0, %0, %4, %1\n v_dot4_u32_u8 %1, %1, %4, %2\n v_dot4_u32_u8 %2, %2, %4, %3\n v_dot4_u32_u8 %3, %3, %4, %0" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));)
KERNEL_END
KERNEL_BEGIN(k_cvt_f64_u32)
  REP16(asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));)
KERNEL_END
// mixed: mad64 interleaved with 1 plain add each (does the add hide in the mul shadow?)
KERNEL_BEGIN(k_mad64_plus_add)
  REP16(asm volatile("v_mad_u64_u32 %0, vcc, %6, %7, %0\n v_add_u32 %2, %2, %6\n v_mad_u64_u32 %1, vcc, %6, %7, %1\n v_add_u32 %3, %3, %6\n v_add_u32 %4, %4, %6\n v_add_u32 %5, %5, %6" : "+v"(q0), "+v"(q1), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "vcc");)
KERNEL_END

typedef void (*kern_t)(u32*, int, u32, u32);
struct Test { const char* name; kern_t k; double ops_per_iter; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  u32* d; CK(hipMalloc(&d, 4096));
  std::vector<Test> tests = {
    {"v_mad_u64_u32 indep x4", k_mad64_ind, 64}, {"v_mad_u64_u32 dependent", k_mad64_dep, 64},
    {"mad_u64_u32+addc pairs (count mads)", k_mad64_addc, 32},
    {"mad_u64_u32 x2 + 4 adds (count groups of 6)", k_mad64_plus_add, 16},
    {"v_mul_lo_u32", k_mullo_ind, 64}, {"v_mul_hi_u32", k_mulhi_ind, 64},
    {"v_mad_u32_u24", k_mad24_ind, 64}, {"v_mul_hi_u32_u24", k_mulhi24_ind, 64},
    {"v_mad_u32_u16", k_mad_u32_u16_ind, 64},
    {"v_add_u32", k_add_ind, 64}, {"v_add_co/addc chain", k_addc_chain, 64}, {"v_add3_u32", k_add3_ind, 64},
    {"v_lshl_add_u64", k_lshladd64_ind, 64}, {"v_lshrrev_b64", k_lshr64_ind, 64}, {"v_mov_b32", k_mov_ind, 64},
    {"v_alignbit_b32", k_alignbit_ind, 64},
    {"v_fma_f64 indep x4", k_fma64_ind, 64}, {"v_fma_f64 dependent", k_fma64_dep, 64},
    {"v_mul_f64", k_mul64f_ind, 64}, {"v_add_f64", k_add64f_ind, 64}, {"v_fma_f32", k_fma32_ind, 64},
    {"v_dot4_u32_u8", k_dot4_ind, 64}, {"v_cvt_f64_u32", k_cvt_f64_u32, 64},
  };
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-46s %10s %10s %10s %10s   (cycles per wave-instruction per SIMD @ nominal %.2f GHz)\n", "op", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD", prop.clockRate / 1e6);
  for (auto& t : tests) {
    printf("%-46s", t.name);
    for (int wps : {1, 2, 4, 8}) {
      // blocks of 256 threads = 4 waves = 1 per SIMD; wps blocks per CU
      int blocks = cus * wps;
      hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 10, 1u, 2u);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, 2u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double total_wave_instr_per_simd = (double)iters * t.ops_per_iter * wps;
      double cycles = best * 1e-3 * (prop.clockRate * 1e3);
      printf(" %10.2f", cycles / total_wave_instr_per_simd);
    }
    printf("\n");
  }
  return 0;
}
