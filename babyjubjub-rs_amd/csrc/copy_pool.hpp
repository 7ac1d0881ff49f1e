// copy_pool.hpp -- the worker threads of the host-pointer pipeline (bjj_hip.hip: run_pipelined).  Plain C++ (no HIP): also
// built for the CPU with ThreadSanitizer / AddressSanitizer by tests/test_emul_sanitizers.py (tests/emul/emul_copy_pool.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

// Worker threads that move PAGEABLE caller memory to / from the pinned staging buffers of the host-pointer pipeline, in
// slices, so that the thread that enqueues copies and kernels never executes a memcpy itself (the staging
// memcpy, serialised with the chunk loop on the calling thread, was what bounded bjj_mul_fixed_base at 17 % of the device
// rate -- not PCIe).  One pool per context, started on the first call that needs it; BJJ_STAGE_THREADS (default 4) workers.
struct CopyGroup { int pending = 0; };   // guarded by CopyPool::mu
struct CopyPool {
  struct Task { uint8_t* dst; const uint8_t* src; size_t len; CopyGroup* g; };
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_task, cv_done;
  std::deque<Task> q;
  bool stop = false;
  static constexpr size_t kSlice = (size_t)1 << 20;
  bool start(int n) {
    for (int i = 0; i < n; i++) {
      try { th.emplace_back([this] { run(); }); } catch (...) { break; }
    }
    return !th.empty();
  }
  void run() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_task.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty()) return;   // stop
        t = q.front(); q.pop_front();
      }
      memcpy(t.dst, t.src, t.len);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--t.g->pending == 0) cv_done.notify_all();
      }
    }
  }
  // Never throws: if the queue cannot grow (std::bad_alloc), what has been queued stays queued -- `pending` counts exactly the
  // queued slices -- and the calling thread copies the rest itself (an exception must not cross the C boundary, and
  // a group's count must never run ahead of its tasks).
  void submit(uint8_t* dst, const uint8_t* src, size_t bytes, CopyGroup* g) noexcept {
    if (!bytes) return;
    size_t o = 0;
    {
      std::lock_guard<std::mutex> lk(mu);
      try {
        for (; o < bytes; o += kSlice) { q.push_back({dst + o, src + o, bytes - o < kSlice ? bytes - o : kSlice, g}); g->pending++; }
      } catch (...) {}
    }
    cv_task.notify_all();
    if (o < bytes) memcpy(dst + o, src + o, bytes - o);
  }
  void wait(CopyGroup* g) {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [g] { return g->pending == 0; });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_task.notify_all();
    for (auto& t : th) t.join();
  }
};

