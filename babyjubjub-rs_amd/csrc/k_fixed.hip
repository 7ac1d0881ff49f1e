// libbjj_hip.so, kernel unit 1: the fixed-base table (build / check) and K1, B8.mul_scalar(n)
// (src/lib.rs:149-164 with self = B8, src/lib.rs:37-46; the engine of PrivateKey::public, :304-306).
#include "k_common.hpp"

// workgroup size / resident workgroups per CU / staging areas per wave of K1 (A/B knobs; default: one 512-lane workgroup
// per CU = 2 waves per SIMD, two 8 KB staging areas per wave)
#ifndef BJJ_K1_BLOCK
#define BJJ_K1_BLOCK BJJ_EPI_BLOCK
#endif
#ifndef BJJ_K1_MIN_BLOCKS
#define BJJ_K1_MIN_BLOCKS 1
#endif
#ifndef BJJ_K1_NBUF
#define BJJ_K1_NBUF 2
#endif

// ---------------------------------------------------------------------------
// init: fixed-base table (layout and recoding: bjj_device.hpp "fixed base").
//   bases:  one thread per window j -> P_j = 2^(W j) * B8 (ladder + inversion; nwin threads)
//   build:  one thread per chain of `chain` consecutive digits of one window (fixed_table_chain)
//   check:  one thread per entry, link conditions of fixed_table_check_slot (bjj_check_table)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) bjj_k_fixed_window_bases(u32* bases, int W, int nwin) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nwin) return;
  store_niels(bases + (size_t)j * NIELS_WORDS, fixed_table_entry(1u, j, W, c_K));
}
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_build_fixed_table(u32* table, const u32* __restrict__ bases, int W, int nwin,
                                                                 u32 chain) {
  const size_t stride = fixed_stride(W);
  const size_t cpw = (stride + chain - 1) / chain;  // chains per window
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cpw * (size_t)nwin) return;
  const int j = (int)(t / cpw);
  const size_t k0 = (t % cpw) * chain;
  const u32 cnt = (u32)(stride - k0 < chain ? stride - k0 : chain);
  fixed_table_chain(table, load_niels(bases + (size_t)j * NIELS_WORDS), (size_t)j * stride + k0, (u32)k0, cnt, W, c_K);
}
__global__ void __launch_bounds__(BJJ_BLOCK) bjj_k_check_fixed_table(const u32* __restrict__ table, const u32* __restrict__ bases,
                                                                 int W, int nwin, unsigned long long* bad) {
  const size_t stride = fixed_stride(W);
  const size_t total = stride * (size_t)nwin, nthreads = (size_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
#pragma unroll 1
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += nthreads)
    mine += (unsigned long long)fixed_table_check_slot(table, bases, (int)(e / stride), (u32)(e % stride), W, nwin, c_K);
  if (mine) atomicAdd(bad, mine);
}


// FORM = EPI_AFFINE: out holds 64-byte points and doubles as the phase-1 stash (xy == out); EPI_COMPRESS: out holds 32-byte
// Point::compress records (src/lib.rs:166-178) and the stash is the scratch set's XY area.
template <int BLOCK, int NBUF, unsigned FORM = EPI_AFFINE>
__device__ __forceinline__ void mul_fixed_base_body(const u32* __restrict__ table, int W, int nwin, const uint8_t* __restrict__ scalars,
                                                    size_t n, uint8_t* __restrict__ out, u32* __restrict__ scratch,
                                                    uint8_t* __restrict__ xy = nullptr) {
  __shared__ u32 lds[NL * 64];
  __shared__ __attribute__((aligned(16))) u32 stage[(BLOCK / 64) * NBUF * FB_STAGE_WORDS];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const GatherCoopLds<NBUF> fb = {table, stage + (threadIdx.x >> 6) * NBUF * FB_STAGE_WORDS, lane};
  Fr run = fr_one();
#pragma unroll 1
  for (size_t i = tid; i - lane < n; i += nthreads) {  // wave-uniform trip count: the gathers are cooperative
    const bool valid = i < n;
    u32 sc[8];
    load_w8(scalars + (valid ? i : n - 1) * 32, sc);
    Ext p = fixed_base_mul(fb, W, nwin, sc, c_K);
    if (valid) epilogue_stash(p, run, ((FORM & EPI_COMPRESS) ? xy : out) + i * 64, scratch + i * 16);
  }
  epilogue_run<BLOCK, FORM>(run, n, tid, nthreads, out, scratch, lds, xy);
}
// Two shapes of the same kernel:
//  * bjj_k_mul_fixed_base: ONE workgroup of BJJ_K1_BLOCK = 512 lanes per CU (2 waves per SIMD, two staging areas per wave,
//    150 KB of LDS): one workgroup-wide inversion per CU and launch.  Best when launches run one after the other.
//  * bjj_k_mul_fixed_base_2x256: TWO workgroups of 256 lanes per CU (the same 2 waves per SIMD).  Alone it is ~2 % slower (two
//    inversions per CU); but a launch only ever needs ONE of the two workgroup slots of a CU, so when a second launch of
//    the context is in flight (another stream, the other scratch set) the hardware gives each launch one slot per CU, the
//    two run half a period out of phase, and one launch's serial section -- the inversion by one wave while the other waves of
//    its workgroup wait, then the epilogue -- is covered by the other launch's main loop: 1.84 vs 1.75 G mults/s on two streams
//    (profiles/r03_ab_k1_2x256_two_streams.txt).  The host picks the shape per call (bjj_hip.hip: fixed_base_variant).
__global__ void __launch_bounds__(BJJ_K1_BLOCK, BJJ_K1_MIN_BLOCKS) bjj_k_mul_fixed_base(const u32* __restrict__ table, int W, int nwin,
                                                                  const uint8_t* __restrict__ scalars, size_t n,
                                                                  uint8_t* __restrict__ out, u32* __restrict__ scratch) {
  mul_fixed_base_body<BJJ_K1_BLOCK, BJJ_K1_NBUF>(table, W, nwin, scalars, n, out, scratch);
}
// The signer's constant-time option (bjj_set_signer_constant_time; PrivateKey::public, src/lib.rs:304-306): the multiplication
// through the scanning policy over the context's small 4-bit table -- no address depends on a digit of the scalar.
template <unsigned FORM>
__device__ __forceinline__ void mul_fixed_base_scan_body(const u32* __restrict__ table, int W, const uint8_t* __restrict__ scalars, size_t n,
                                                         uint8_t* __restrict__ out, u32* __restrict__ scratch, uint8_t* __restrict__ xy, int nwin) {
  __shared__ u32 lds[NL * 64];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const GatherScan fb = {table, (u32)fixed_stride(W)};
  Fr run = fr_one();
#pragma unroll 1
  for (size_t i = tid; i < n; i += nthreads) {
    u32 sc[8];
    load_w8(scalars + i * 32, sc);
    Ext p = fixed_base_mul(fb, W, nwin, sc, c_K);
    epilogue_stash(p, run, ((FORM & EPI_COMPRESS) ? xy : out) + i * 64, scratch + i * 16);
  }
  epilogue_run<BJJ_EPI_BLOCK, FORM>(run, n, tid, nthreads, out, scratch, lds, xy);
}
__global__ void __launch_bounds__(BJJ_EPI_BLOCK, 1) bjj_k_mul_fixed_base_scan(const u32* __restrict__ table, int W, int nwin,
                                                                        const uint8_t* __restrict__ scalars, size_t n,
                                                                        uint8_t* __restrict__ out, u32* __restrict__ scratch) {
  mul_fixed_base_scan_body<EPI_AFFINE>(table, W, scalars, n, out, scratch, nullptr, nwin);
}
__global__ void __launch_bounds__(256, 2) bjj_k_mul_fixed_base_2x256(const u32* __restrict__ table, int W, int nwin,
                                                                    const uint8_t* __restrict__ scalars, size_t n,
                                                                    uint8_t* __restrict__ out, u32* __restrict__ scratch) {
  mul_fixed_base_body<256, 2>(table, W, nwin, scalars, n, out, scratch);
}
// ---- the same three kernels with Point::compress fused into the epilogue (src/lib.rs:166-178): 32 bytes per result instead of 64.
// `sk.public().compress()` is what a key server ships, and the host-pointer form of K1 is bound by the copy-out (64 MB of
// affine points per 2^20 items = 1.2 ms of PCIe against 0.6 ms of kernel): half the bytes, and no second pass over the points.
__global__ void __launch_bounds__(BJJ_K1_BLOCK, BJJ_K1_MIN_BLOCKS) bjj_k_mul_fixed_base_c32(const u32* __restrict__ table, int W, int nwin,
                                                                      const uint8_t* __restrict__ scalars, size_t n,
                                                                      uint8_t* __restrict__ out32, u32* __restrict__ scratch,
                                                                      uint8_t* __restrict__ xy) {
  mul_fixed_base_body<BJJ_K1_BLOCK, BJJ_K1_NBUF, EPI_COMPRESS>(table, W, nwin, scalars, n, out32, scratch, xy);
}
__global__ void __launch_bounds__(256, 2) bjj_k_mul_fixed_base_2x256_c32(const u32* __restrict__ table, int W, int nwin,
                                                                        const uint8_t* __restrict__ scalars, size_t n,
                                                                        uint8_t* __restrict__ out32, u32* __restrict__ scratch,
                                                                        uint8_t* __restrict__ xy) {
  mul_fixed_base_body<256, 2, EPI_COMPRESS>(table, W, nwin, scalars, n, out32, scratch, xy);
}
__global__ void __launch_bounds__(BJJ_EPI_BLOCK, 1) bjj_k_mul_fixed_base_scan_c32(const u32* __restrict__ table, int W, int nwin,
                                                                            const uint8_t* __restrict__ scalars, size_t n,
                                                                            uint8_t* __restrict__ out32, u32* __restrict__ scratch,
                                                                            uint8_t* __restrict__ xy) {
  mul_fixed_base_scan_body<EPI_COMPRESS>(table, W, scalars, n, out32, scratch, xy, nwin);
}

// ---- launchers (declared in bjj_launch.hpp) ------------------------------------------------------------
namespace bjjk {
int fixed_base_lanes_per_cu(int variant) {   // one grid size serves the affine and the compressed form: the lesser of the two
  const int a = variant ? occupancy_of(bjj_k_mul_fixed_base_2x256, 256) * 256 : occupancy_of(bjj_k_mul_fixed_base, BJJ_K1_BLOCK) * BJJ_K1_BLOCK;
  const int b = variant ? occupancy_of(bjj_k_mul_fixed_base_2x256_c32, 256) * 256 : occupancy_of(bjj_k_mul_fixed_base_c32, BJJ_K1_BLOCK) * BJJ_K1_BLOCK;
  return a < b ? a : b;
}
hipError_t build_fixed_table(hipStream_t st, u32* table, u32* bases, int W, int nwin) {
  const size_t entries = fixed_stride(W) * (size_t)nwin;
  // chain length: long enough to amortise the start ladder and the inversion, short enough to fill the GPU
  size_t chain = entries >> 18;
  chain = chain < 4 ? 4 : (chain > 256 ? 256 : chain);
  const size_t chains = ((fixed_stride(W) + chain - 1) / chain) * (size_t)nwin;
  BJJ_LAUNCH(bjj_k_fixed_window_bases, dim3((nwin + 63) / 64), dim3(64), 0, st, bases, W, nwin);
  BJJ_LAUNCH(bjj_k_build_fixed_table, dim3((unsigned)((chains + BJJ_BLOCK - 1) / BJJ_BLOCK)), dim3(BJJ_BLOCK), 0, st,
                     table, bases, W, nwin, (u32)chain);
  return hipGetLastError();
}
hipError_t check_fixed_table(hipStream_t st, int grid, const u32* table, const u32* bases, int W, int nwin,
                             unsigned long long* d_bad) {
  BJJ_LAUNCH(bjj_k_check_fixed_table, dim3((unsigned)grid), dim3(BJJ_BLOCK), 0, st, table, bases, W, nwin, d_bad);
  return hipGetLastError();
}
// xy != nullptr: the compressed form (out = 32-byte records, xy = the 64-byte-per-item stash)
hipError_t mul_fixed_base_scan(hipStream_t st, int cus, const u32* table, int W, int nwin, const uint8_t* scalars, size_t n,
                               uint8_t* out, u32* scratch, uint8_t* xy) {
  const size_t want = (n + BJJ_EPI_BLOCK - 1) / BJJ_EPI_BLOCK;
  const size_t cap = (size_t)cus * (size_t)(xy ? occupancy_of(bjj_k_mul_fixed_base_scan_c32, BJJ_EPI_BLOCK) : occupancy_of(bjj_k_mul_fixed_base_scan, BJJ_EPI_BLOCK));
  const int grid = (int)(want < cap ? (want ? want : 1) : cap);
  if (xy)
    BJJ_LAUNCH(bjj_k_mul_fixed_base_scan_c32, dim3(grid), dim3(BJJ_EPI_BLOCK), 0, st, table, W, nwin, scalars, n, out, scratch, xy);
  else
    BJJ_LAUNCH(bjj_k_mul_fixed_base_scan, dim3(grid), dim3(BJJ_EPI_BLOCK), 0, st, table, W, nwin, scalars, n, out, scratch);
  return hipGetLastError();
}
hipError_t mul_fixed_base(hipStream_t st, int cus, int lanes_per_cu, int variant, const u32* table, int W, int nwin, const uint8_t* scalars,
                          size_t n, uint8_t* out, u32* scratch, uint8_t* xy) {
  const int block = variant ? 256 : BJJ_K1_BLOCK;
  const size_t want = (n + block - 1) / block, cap = (size_t)cus * (size_t)(lanes_per_cu / block);
  const int grid = (int)(want < cap ? (want ? want : 1) : cap);
  if (xy) {
    if (variant)
      BJJ_LAUNCH(bjj_k_mul_fixed_base_2x256_c32, dim3(grid), dim3(256), 0, st, table, W, nwin, scalars, n, out, scratch, xy);
    else
      BJJ_LAUNCH(bjj_k_mul_fixed_base_c32, dim3(grid), dim3(BJJ_K1_BLOCK), 0, st, table, W, nwin, scalars, n, out, scratch, xy);
  } else if (variant)
    BJJ_LAUNCH(bjj_k_mul_fixed_base_2x256, dim3(grid), dim3(256), 0, st, table, W, nwin, scalars, n, out, scratch);
  else
    BJJ_LAUNCH(bjj_k_mul_fixed_base, dim3(grid), dim3(BJJ_K1_BLOCK), 0, st, table, W, nwin, scalars, n, out, scratch);
  return hipGetLastError();
}
}  // namespace bjjk
