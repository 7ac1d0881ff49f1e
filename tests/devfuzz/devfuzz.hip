// TEST-ONLY kernels (not part of libbjj_hip.so): run the field / curve primitives that SHIP -- the inline-asm column
// multiplier fr_mul_columns / fr_sqr_columns, the asm dot products with scalar-register matrix operands, fr_inv_gcd,
// ext_madd / ext_dbl / ext_add_pn -- on raw limb vectors chosen by the test, so that the lazy-reduction contract of
// fr.hpp:14-20 ("limbs < 2^30, values < 13 r") is asserted on the device code itself and not only on the portable form
// the CPU harness (tests/emul) executes.  Built twice from the product's own headers:
//     libbjj_devfuzz_asm.so        default flags (what ships)
//     libbjj_devfuzz_portable.so   -DBJJ_NO_ASM_COLUMNS (compiler-scheduled columns)
// tests/test_gpu_devfuzz.py compares both bit for bit and checks samples against Python integers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../babyjubjub-rs_amd/csrc/bjj_device.hpp"
#include "../../babyjubjub-rs_amd/csrc/bjj_constants.inc"

using namespace bjj;

static __constant__ Consts c_K = {
    BJJ_K_A, BJJ_K_D, BJJ_K_F, BJJ_K_FINV_PLAIN, BJJ_K_FINV, BJJ_K_L_R1, BJJ_K_L_R2, BJJ_K_DP, BJJ_K_D2P, BJJ_K_DPINV, BJJ_K_B8X, BJJ_K_B8Y, BJJ_K_TS_G, BJJ_K_HALFQ,
    BJJ_K_ORDER, BJJ_K_ORDER2, BJJ_K_ORDER4, BJJ_K_L, BJJ_K_L2, BJJ_K_L4,
    BJJ_K_POSEIDON_CF, BJJ_K_POSEIDON_KP, BJJ_K_POSEIDON_SP, BJJ_K_POSEIDON_AL, BJJ_K_POSEIDON_M, BJJ_K_POSEIDON_CAB,
    BJJ_K_TS_NEG, BJJ_K_TS_HALF, BJJ_K_TS_HASH};

__device__ __forceinline__ Fr ld(const u32* p) { Fr f; for (int i = 0; i < NL; i++) f.v[i] = p[i]; return f; }
__device__ __forceinline__ void st(u32* p, const Fr& f) { for (int i = 0; i < NL; i++) p[i] = f.v[i]; }

enum { FZ_MUL = 0, FZ_SQR = 1, FZ_INV = 2, FZ_DOT6 = 3, FZ_DOT15 = 4, FZ_DOT151 = 5, FZ_DOT2ADD = 6, FZ_MADD = 7, FZ_DBL = 8,
       FZ_ADDPN = 9, FZ_DBL_NOT = 10, FZ_MADD_NOT = 11, FZ_CONSTS = 20 };

// a, b, c: per-item records of wa / wb / wc limb-words; out: wo words per item.  `row` selects the wave-uniform
// constant operands of the dot products (Poseidon matrix rows / sparse-round vectors), as in the kernels that ship.
template <int OP>
__global__ void __launch_bounds__(256) fz_kernel(const u32* __restrict__ a, const u32* __restrict__ b, const u32* __restrict__ c,
                                                 u32* __restrict__ out, size_t n, int row) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    if (OP == FZ_MUL) st(out + i * 9, fr_mul(ld(a + i * 9), ld(b + i * 9)));
    if (OP == FZ_SQR) st(out + i * 9, fr_sqr(ld(a + i * 9)));
    if (OP == FZ_INV) st(out + i * 9, fr_inv(ld(a + i * 9)));
    if (OP == FZ_DOT6) {
      Fr x[6];
      for (int j = 0; j < 6; j++) x[j] = ld(b + i * 54 + j * 9);
      st(out + i * 9, pos_dot6(c_K.PM + 6 * (row % 6), x));
    }
    if (OP == FZ_DOT15) {
      Fr x[6];
      for (int j = 0; j < 6; j++) x[j] = ld(b + i * 54 + j * 9);
      st(out + i * 9, pos_dot15(c_K.PSP[11 * row], x[0], c_K.PSP + 11 * row + 1, x + 1));
    }
    if (OP == FZ_DOT151) {
      Fr x[7];
      for (int j = 0; j < 7; j++) x[j] = ld(b + i * 63 + j * 9);
      st(out + i * 9, pos_dot151(c_K.PSP[11 * row], x[0], c_K.PSP + 11 * row + 1, x + 1, c_K.PCAB[row % 30], x[6]));
    }
    if (OP == FZ_DOT2ADD) {
      st(out + i * 9, pos_dot2_add(c_K.PSP[11 * row + 6], ld(a + i * 9), c_K.PSP[11 * row + 7], ld(b + i * 9), ld(c + i * 9)));
    }
    if (OP == FZ_MADD || OP == FZ_MADD_NOT || OP == FZ_DBL || OP == FZ_DBL_NOT || OP == FZ_ADDPN) {
      Ext p; p.X = ld(a + i * 36); p.Y = ld(a + i * 36 + 9); p.Z = ld(a + i * 36 + 18); p.T = ld(a + i * 36 + 27);
      Ext r;
      if (OP == FZ_MADD || OP == FZ_MADD_NOT) {
        Niels q; q.ymx = ld(b + i * 27); q.ypx = ld(b + i * 27 + 9); q.t2d = ld(b + i * 27 + 18);
        r = OP == FZ_MADD ? ext_madd<true>(p, q) : ext_madd<false>(p, q);
      } else if (OP == FZ_ADDPN) {
        PNiels q; q.ymx = ld(b + i * 36); q.ypx = ld(b + i * 36 + 9); q.t2d = ld(b + i * 36 + 18); q.z2 = ld(b + i * 36 + 27);
        r = ext_add_pn(p, q);
      } else {
        r = OP == FZ_DBL ? ext_dbl<true>(p) : ext_dbl<false>(p);
      }
      st(out + i * 36, r.X); st(out + i * 36 + 9, r.Y); st(out + i * 36 + 18, r.Z); st(out + i * 36 + 27, r.T);
    }
  }
}
// the constant operands the dot-product ops use: PM (36), PSP (660), PCAB (30), then DP, D2P, F as raw limb vectors
__global__ void fz_consts(u32* out) {
  if (blockIdx.x || threadIdx.x) return;
  size_t o = 0;
  for (int j = 0; j < 36; j++, o += 9) st(out + o, c_K.PM[j]);
  for (int j = 0; j < 660; j++, o += 9) st(out + o, c_K.PSP[j]);
  for (int j = 0; j < 30; j++, o += 9) st(out + o, c_K.PCAB[j]);
  st(out + o, c_K.DP); o += 9;
  st(out + o, c_K.D2P); o += 9;
  st(out + o, c_K.F);
}

#define FZ_CASE(OP) case OP: hipLaunchKernelGGL(fz_kernel<OP>, dim3(grid), dim3(256), 0, st_, a, b, c, out, n, row); break

extern "C" __attribute__((visibility("default"))) int fz_run(int op, const uint32_t* a, const uint32_t* b, const uint32_t* c,
                                                              uint32_t* out, size_t n, int row, void* stream) {
  hipStream_t st_ = (hipStream_t)stream;
  size_t want = (n + 255) / 256;
  const int grid = (int)(want < 2048 ? (want ? want : 1) : 2048);
  switch (op) {
    FZ_CASE(FZ_MUL); FZ_CASE(FZ_SQR); FZ_CASE(FZ_INV); FZ_CASE(FZ_DOT6); FZ_CASE(FZ_DOT15); FZ_CASE(FZ_DOT151);
    FZ_CASE(FZ_DOT2ADD); FZ_CASE(FZ_MADD); FZ_CASE(FZ_DBL); FZ_CASE(FZ_ADDPN); FZ_CASE(FZ_DBL_NOT); FZ_CASE(FZ_MADD_NOT);
    case FZ_CONSTS: hipLaunchKernelGGL(fz_consts, dim3(1), dim3(64), 0, st_, out); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) const char* fz_variant(void) {
#if defined(BJJ_NO_ASM_COLUMNS)
  return "portable";
#else
  return "asm";
#endif
}
