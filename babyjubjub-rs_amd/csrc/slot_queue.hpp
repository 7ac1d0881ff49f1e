// Per-XCD rings of free per-lane-table slots: the device side (included by k_common.hpp; also built into the test-only
// library tests/devfuzz with a small BJJ_SLOT_SPIN_LIMIT, tests/test_gpu_devfuzz.py).  The host side -- allocation, initial
// fill, the check in bjj_sync -- is in bjj_hip.hip (slot_queue_fill / slot_queue_check).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bjj { typedef uint32_t u32; }
using bjj::u32;

// ---------------------------------------------------------------------------
// Slot queue (one per XCD: a table slot never migrates between XCDs, whose L2s are not coherent with each other inside a
// kernel): words [0] head ticket, [1] tail ticket, [SLOTQ_HDR + i] = slot id + 1, or 0 while the slot is out.  Tickets make
// it a ring: a pop takes entry (head++ mod cap), a push refills entry (tail++ mod cap).  There are exactly as many slots
// per XCD as waves can be resident there, so a pop finds its entry full except for the instant in which the push that
// refills it is still in flight (it then spins on that one word).  The tickets wrap at cap by themselves (atomicInc).
#define SLOTQ_HDR 16
#define SLOTQ_ERR 2          // header word: pops that gave up waiting (bjj_sync reports them and re-initialises the rings)
// A pop finds its ring entry full except for the instant in which the push that refills it is in flight.  If a slot is never
// pushed back (a kernel that was aborted, residency assumptions that no longer hold) the wait would never end and every later
// launch on the scratch set would hang the GPU.  The wait is therefore bounded: after BJJ_SLOT_SPIN_LIMIT polls (seconds) the
// pop counts itself in SLOTQ_ERR and continues on its XCD's OVERFLOW slot -- one extra slot per XCD behind the regular ones,
// never queued.  Results of such a launch are not trustworthy (two starved workgroups may share the overflow slot); the host
// sees the count at the next bjj_sync, returns an error instead of hanging, and rebuilds the rings.
#ifndef BJJ_SLOT_SPIN_LIMIT
#define BJJ_SLOT_SPIN_LIMIT (1u << 22)
#endif
__device__ __forceinline__ u32 xcc_id() { return (u32)__builtin_amdgcn_s_getreg(6164); }   // hwreg(HW_REG_XCC_ID, 0, 4)
// The kernels get (slots per XCD) | (number of XCDs << 16) in one word.  The XCD count was probed at bjj_init; an id beyond it
// (a partition mode that changed since) wraps onto an existing queue instead of indexing past the allocation.
__device__ __forceinline__ u32* slot_queue_of_this_xcd(u32* slotq, u32 cap_nx) {
  const u32 cap = cap_nx & 0xffffu, nx = cap_nx >> 16;
  return slotq + (size_t)(xcc_id() % nx) * (SLOTQ_HDR + cap);
}
__device__ __forceinline__ u32 slot_overflow_of_this_xcd(u32 cap_nx) {
  const u32 cap = cap_nx & 0xffffu, nx = cap_nx >> 16;
  return nx * cap + xcc_id() % nx;
}
// one thread takes / returns a slot.  Hand-over ordering: the holder's stores to its table slot are complete
// before the slot number is published (fence + the push), and the next holder's accesses start after its pop (fence), so that
// nothing of the previous holder can land on top of the new holder's table.
// BJJ_SLOT_FENCE_SCOPE = "workgroup" (s_waitcnt vmcnt(0): the stores are acknowledged by the L2) is what ships: a slot never
// leaves its XCD, i.e. holder and successor share ONE L2 -- the push / pop atomics and all table traffic meet there, vL1D is
// write-through, and the successor reads only bytes it has itself written in this tenancy.  "agent" -- what the HIP memory model
// asks for between workgroups in general -- adds an L2 write-back of every resident workgroup's dirty tables per hand-over and
// costs verify 1.6 % (profiles/r04_ab_slot_fences.txt); it can be selected with -DBJJ_SLOT_FENCE_SCOPE='"agent"'.
#ifndef BJJ_SLOT_FENCE_SCOPE
#define BJJ_SLOT_FENCE_SCOPE "workgroup"
#endif
__device__ __forceinline__ u32 slot_pop_one(u32* q, u32 cap_nx) {
  const u32 cap = cap_nx & 0xffffu;
  const u32 t = atomicInc(&q[0], cap - 1u);            // ticket in [0, cap): wraps by itself
  u32 v, polls = 0;
  do { v = atomicExch(&q[SLOTQ_HDR + t], 0u); } while (v == 0u && ++polls < BJJ_SLOT_SPIN_LIMIT);
  if (v == 0u) {                                       // starved: flag it, go on with the overflow slot
    atomicAdd(&q[SLOTQ_ERR], 1u);
    v = slot_overflow_of_this_xcd(cap_nx) + 1u;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, BJJ_SLOT_FENCE_SCOPE);
  return v - 1u;
}
__device__ __forceinline__ void slot_push_one(u32* q, u32 cap_nx, u32 slot) {
  const u32 cap = cap_nx & 0xffffu, nx = cap_nx >> 16;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, BJJ_SLOT_FENCE_SCOPE);
  if (slot >= nx * cap) return;                        // the overflow slot is never queued
  const u32 t = atomicInc(&q[1], cap - 1u);
  u32 polls = 0;
  while (atomicCAS(&q[SLOTQ_HDR + t], 0u, slot + 1u) != 0u && ++polls < BJJ_SLOT_SPIN_LIMIT) {}
  if (polls >= BJJ_SLOT_SPIN_LIMIT) atomicAdd(&q[SLOTQ_ERR], 1u);
}
// a WAVE takes / returns a slot (lane 0 does it, every lane gets the number)
__device__ __forceinline__ u32 slot_pop(u32* q, u32 cap_nx, int lane) {
  u32 v = 0;
  if (lane == 0) v = slot_pop_one(q, cap_nx);
  return (u32)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void slot_push(u32* q, u32 cap_nx, u32 slot, int lane) {
  if (lane == 0) slot_push_one(q, cap_nx, slot);
}
