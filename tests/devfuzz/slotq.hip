// TEST-ONLY kernel (not part of libbjj_hip.so): the slot-queue hand-over of csrc/slot_queue.hpp, built with a small
// BJJ_SLOT_SPIN_LIMIT so that the bounded wait can be driven into its give-up path in milliseconds
// (tests/test_gpu_devfuzz.py::test_slot_queue_*).  Every workgroup pops a slot, marks it, holds it for a while, pushes it back.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../babyjubjub-rs_amd/csrc/slot_queue.hpp"

// owner[slot] counts concurrent holders (must never exceed 1 for a regular slot); got[b] = the slot workgroup b worked on
__global__ void __launch_bounds__(64) sq_kernel(u32* slotq, u32 cap_nx, u32* owner, u32* got, u32* clash, int hold) {
  const int lane = threadIdx.x & 63;
  u32* q = slot_queue_of_this_xcd(slotq, cap_nx);
  const u32 slot = slot_pop(q, cap_nx, lane);
  const u32 cap = cap_nx & 0xffffu, nx = cap_nx >> 16;
  if (lane == 0) {
    got[blockIdx.x] = slot;
    if (slot < nx * cap) {
      if (atomicAdd(&owner[slot], 1u) != 0u) atomicAdd(clash, 1u);   // somebody else holds this slot right now
      for (int i = 0; i < hold; i++) __builtin_amdgcn_s_sleep(8);
      atomicSub(&owner[slot], 1u);
    }
  }
  slot_push(q, cap_nx, slot, lane);
}
extern "C" __attribute__((visibility("default"))) int sq_run(uint32_t* slotq, uint32_t cap_nx, uint32_t* owner, uint32_t* got,
                                                              uint32_t* clash, int blocks, int hold, void* stream) {
  hipLaunchKernelGGL(sq_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, slotq, cap_nx, owner, got, clash, hold);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) unsigned sq_spin_limit(void) { return BJJ_SLOT_SPIN_LIMIT; }
