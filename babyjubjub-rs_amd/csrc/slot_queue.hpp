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
// -- acknowledged by the XCD's L2 -- before the slot number is published, and the next holder's accesses start after its pop
// has returned, so that nothing of the previous holder can land on top of the new holder's table.
// What enforces it is the EXPLICIT `s_waitcnt vmcnt(0)` of slot_release_wave() in front of the push: holder and successor may
// sit on different CUs of the XCD, and a CU's requests to different L2 channels are not ordered among each other, so the table
// stores must have been acknowledged before the push atomic is issued.  A workgroup-scope release fence alone does NOT emit
// that wait on gfx950 (outside tgsplit mode the memory model only orders a workgroup's accesses through its own CU: the ISA of
// `store; fence(release, "workgroup"); atomic` is store, s_waitcnt lgkmcnt(0), atomic -- ADVICE r04 found round 4 shipping
// exactly that); tests/test_isa_checks.py asserts the wait in the shipped code objects.  Every WAVE that wrote to the slot
// calls slot_release_wave() (vmcnt counts per wave); in a multi-wave workgroup a barrier follows before one thread pushes.
// No cache maintenance is needed on top: a slot never leaves its XCD, i.e. holder and successor share ONE L2 -- the push / pop
// atomics and all table traffic meet there, vL1D is write-through, and the successor reads only bytes it has itself written
// in this tenancy.  BJJ_SLOT_FENCE_SCOPE = "workgroup" therefore ships (compiler ordering only); "agent" -- what the HIP memory
// model asks for between workgroups in general -- adds an L2 write-back of every resident workgroup's dirty tables per
// hand-over and costs verify 1.6 % (profiles/r04_ab_slot_fences.txt, r05_ab_slot_fences.txt); -DBJJ_SLOT_FENCE_SCOPE='"agent"'.
#ifndef BJJ_SLOT_FENCE_SCOPE
#define BJJ_SLOT_FENCE_SCOPE "workgroup"
#endif
// all outstanding vector-memory operations of THIS wave (its table stores among them) have been acknowledged
__device__ __forceinline__ void slot_release_wave() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
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
  slot_release_wave();                                 // the pushing wave's own stores: acknowledged by the L2
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
