// TEST DOUBLE (tests only; never shipped, never linked): the ten RCCL entry points libbjj_hip.so binds with dlsym, emulated
// inside ONE process with device-to-device copies, so that the RCCL branch of bjj_multi_* (csrc/bjj_multi.inc) -- which
// buffers, offsets, counts, roots and streams it hands to ncclScatter / ncclGather / ncclSend / ncclRecv, and how it groups
// them -- can be executed for G = 2, 3, 8 ranks on a box with ONE GPU (real RCCL refuses two ranks on one device).
// Selected with BJJ_RCCL_LIBRARY=<this .so>.  Semantics follow /opt/rocm/include/rccl/rccl.h:
//   ncclScatter  rank i receives block i (recvcount elements) of the root's sendbuff; in place when
//                recvbuff == sendbuff + rank * recvcount (rccl.h:754-770)
//   ncclGather   the root receives sendcount elements from rank i at offset i * sendcount; in place when
//                sendbuff == recvbuff + rank * sendcount (rccl.h:729-748)
//   ncclSend / ncclRecv   matched point to point inside a group, in posting order per (sender, receiver) pair
//   ncclGroupStart / End  every call between them is collected and executed at the (outermost) GroupEnd
// It also CHECKS what real RCCL would reject or deadlock on: every rank of a collective must post it inside the same
// group with the same count and root; every send needs a matching receive of the same size; no call outside a group.
// A violated rule makes the call return ncclInvalidUsage (5) and the message is kept for ncclGetErrorString.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

typedef int ncclResult_t;
typedef int ncclDataType_t;
struct ncclComm { int rank, nranks, device; };
typedef ncclComm* ncclComm_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

namespace {
struct Op { int kind; const void* send; void* recv; size_t count; int root_or_peer; ncclComm_t comm; hipStream_t stream; };
enum { SCATTER, GATHER, SEND, RECV };
int g_depth = 0;
std::vector<Op> g_ops;
std::string g_error = "no error";
long g_groups = 0, g_scatter = 0, g_gather = 0, g_send = 0, g_recv = 0;
ncclResult_t fail(const std::string& m) { g_error = "fake RCCL: " + m; fprintf(stderr, "%s\n", g_error.c_str()); return ncclInvalidUsage; }
size_t elem(ncclDataType_t t) { return (t == 0 || t == 1) ? 1 : 4; }   // ncclInt8 / ncclUint8 (the library only moves bytes)
ncclResult_t copy(void* dst, const void* src, size_t bytes, ncclComm_t on, hipStream_t st) {
  if (!bytes || dst == src) return ncclSuccess;
  if (hipSetDevice(on->device) != hipSuccess) return ncclUnhandledCudaError;
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
// failure injection (tests of the library's error path): FAKE_RCCL_FAIL_AT = k makes the k-th posted operation (1-based,
// counted over the life of the process) fail synchronously, the way a real RCCL call rejects a bad argument.  Like real RCCL
// the double then discards the whole group: its GroupEnd reports the error and launches nothing.
long g_posted = 0;
bool g_group_failed = false;
long fail_at() { const char* e = getenv("FAKE_RCCL_FAIL_AT"); return e && *e ? atol(e) : 0; }
ncclResult_t post(const Op& op) {
  if (g_depth == 0) return fail("collective / p2p call outside ncclGroupStart/End in a single-process multi-rank clique");
  if (++g_posted == fail_at()) { g_group_failed = true; return fail("injected failure of a posted operation"); }
  g_ops.push_back(op);
  return ncclSuccess;
}
ncclResult_t run_group() {
  std::vector<Op> ops;
  ops.swap(g_ops);
  std::vector<bool> done(ops.size(), false);
  // collectives: all nranks calls with the same kind / count / root, in posting order
  for (size_t i = 0; i < ops.size(); i++) {
    if (done[i] || (ops[i].kind != SCATTER && ops[i].kind != GATHER)) continue;
    const int n = ops[i].comm->nranks, kind = ops[i].kind, root = ops[i].root_or_peer;
    const size_t count = ops[i].count;
    std::vector<int> idx((size_t)n, -1);
    for (size_t j = i; j < ops.size(); j++) {
      if (done[j] || ops[j].kind != kind) continue;
      const int r = ops[j].comm->rank;
      if (idx[(size_t)r] >= 0) continue;                     // a later collective of the same rank
      if (ops[j].count != count || ops[j].root_or_peer != root) return fail("ranks disagree on count / root of a collective");
      idx[(size_t)r] = (int)j;
    }
    for (int r = 0; r < n; r++) if (idx[(size_t)r] < 0) return fail("a rank did not post its part of a collective inside the group (real RCCL would hang)");
    const Op& ro = ops[(size_t)idx[(size_t)root]];
    for (int r = 0; r < n; r++) {
      const Op& o = ops[(size_t)idx[(size_t)r]];
      ncclResult_t rc;
      if (kind == SCATTER) {
        if (!ro.send) return fail("ncclScatter: sendbuff is NULL on the root");
        if (!o.recv) return fail("ncclScatter: recvbuff is NULL");
        rc = copy(o.recv, (const char*)ro.send + (size_t)r * count, count, o.comm, o.stream);
      } else {
        if (!ro.recv) return fail("ncclGather: recvbuff is NULL on the root");
        if (!o.send) return fail("ncclGather: sendbuff is NULL");
        rc = copy((char*)ro.recv + (size_t)r * count, o.send, count, o.comm, o.stream);
      }
      if (rc) return rc;
      done[(size_t)idx[(size_t)r]] = true;
    }
  }
  // point to point: the k-th send a -> b pairs with the k-th receive b <- a
  for (size_t i = 0; i < ops.size(); i++) {
    if (done[i] || ops[i].kind != SEND) continue;
    const int a = ops[i].comm->rank, b = ops[i].root_or_peer;
    size_t j = 0;
    for (; j < ops.size(); j++)
      if (!done[j] && ops[j].kind == RECV && ops[j].comm->rank == b && ops[j].root_or_peer == a) break;
    if (j == ops.size()) return fail("ncclSend without a matching ncclRecv in the group (real RCCL would hang)");
    if (ops[j].count != ops[i].count) return fail("ncclSend / ncclRecv sizes differ");
    // the copy is ordered on the RECEIVER's stream, after both sides' prior work: make the receiver wait for the sender's stream
    hipEvent_t ev;
    if (hipSetDevice(ops[i].comm->device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    hipEventRecord(ev, ops[i].stream);
    hipSetDevice(ops[j].comm->device);
    hipStreamWaitEvent(ops[j].stream, ev, 0);
    ncclResult_t rc = copy(ops[j].recv, ops[i].send, ops[i].count, ops[j].comm, ops[j].stream);
    // ... and the sender's stream must not run ahead of the transfer either
    hipEvent_t ev2; hipEventCreateWithFlags(&ev2, hipEventDisableTiming); hipEventRecord(ev2, ops[j].stream);
    hipSetDevice(ops[i].comm->device); hipStreamWaitEvent(ops[i].stream, ev2, 0);
    hipEventDestroy(ev); hipEventDestroy(ev2);
    if (rc) return rc;
    done[i] = done[j] = true;
  }
  for (size_t i = 0; i < ops.size(); i++) if (!done[i]) return fail("ncclRecv without a matching ncclSend in the group (real RCCL would hang)");
  return ncclSuccess;
}
}  // namespace

extern "C" {
#define API __attribute__((visibility("default")))
API ncclResult_t ncclGetVersion(int* v) { *v = 99999; return ncclSuccess; }   // recognisable: not a real RCCL
API const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : g_error.c_str(); }
API ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist) {
  if (!comm || ndev < 1) return fail("ncclCommInitAll: bad arguments");
  for (int i = 0; i < ndev; i++) comm[i] = new ncclComm{i, ndev, devlist ? devlist[i] : i};
  return ncclSuccess;
}
long g_aborted = 0;
API ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
API ncclResult_t ncclCommAbort(ncclComm_t c) { g_aborted++; delete c; return ncclSuccess; }
API long fake_rccl_aborted(void) { return g_aborted; }
API ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
API ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
  if (--g_depth) return ncclSuccess;
  g_groups++;
  if (g_group_failed) { g_group_failed = false; g_ops.clear(); return fail("group discarded: a call inside it had failed"); }
  return run_group();
}
API ncclResult_t ncclScatter(const void* s, void* r, size_t n, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st) {
  g_scatter++; return post(Op{SCATTER, s, r, n * elem(t), root, c, st});
}
API ncclResult_t ncclGather(const void* s, void* r, size_t n, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st) {
  g_gather++; return post(Op{GATHER, s, r, n * elem(t), root, c, st});
}
API ncclResult_t ncclSend(const void* s, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  g_send++; return post(Op{SEND, s, nullptr, n * elem(t), peer, c, st});
}
API ncclResult_t ncclRecv(void* r, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  g_recv++; return post(Op{RECV, nullptr, r, n * elem(t), peer, c, st});
}
// test introspection: how often each entry point was used since the last call
API void fake_rccl_counters(long out[5]) {
  out[0] = g_groups; out[1] = g_scatter; out[2] = g_gather; out[3] = g_send; out[4] = g_recv;
  g_groups = g_scatter = g_gather = g_send = g_recv = 0;
}
}
