// Developer probe: what is the cheapest way to move a caller's pageable buffers?  (a) memcpy into pinned staging with
// 1..8 threads, (b) hipHostRegister the caller's pages and DMA in place, (c) plain pageable hipMemcpy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t MB = 1 << 20, nin = 32 * MB, nout = 64 * MB;
  char* uin = (char*)aligned_alloc(4096, nin); char* uout = (char*)aligned_alloc(4096, nout);
  memset(uin, 1, nin); memset(uout, 2, nout);
  char *pin, *din, *dout; hipHostMalloc((void**)&pin, nout); hipMalloc((void**)&din, nin); hipMalloc((void**)&dout, nout);
  for (int rep = 0; rep < 2; rep++) {
    for (int th : {1, 2, 4, 8}) {
      double t0 = now();
      std::vector<std::thread> ts; size_t per = nout / th;
      for (int i = 0; i < th; i++) ts.emplace_back([=] { memcpy(pin + i * per, uout + i * per, per); });
      for (auto& t : ts) t.join();
      double dt = now() - t0; printf("memcpy 64 MB pageable->pinned, %d threads: %.2f ms (%.1f GB/s)\n", th, dt * 1e3, nout / dt / 1e9);
    }
    double t0 = now(); hipMemcpy(din, uin, nin, hipMemcpyHostToDevice); double t1 = now(); hipMemcpy(uout, dout, nout, hipMemcpyDeviceToHost); double t2 = now();
    printf("pageable hipMemcpy: H2D 32 MB %.2f ms, D2H 64 MB %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    t0 = now(); hipHostRegister(uin, nin, hipHostRegisterDefault); hipHostRegister(uout, nout, hipHostRegisterDefault); t1 = now();
    hipMemcpy(din, uin, nin, hipMemcpyHostToDevice); t2 = now(); hipMemcpy(uout, dout, nout, hipMemcpyDeviceToHost); double t3 = now();
    hipHostUnregister(uin); hipHostUnregister(uout); double t4 = now();
    printf("hipHostRegister 96 MB: %.2f ms; H2D 32 MB %.2f ms; D2H 64 MB %.2f ms; unregister %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    t0 = now(); hipMemcpy(din, pin, nin, hipMemcpyHostToDevice); t1 = now(); hipMemcpy(pin, dout, nout, hipMemcpyDeviceToHost); t2 = now();
    printf("pinned hipMemcpy: H2D 32 MB %.2f ms, D2H 64 MB %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
  }
  return 0;
}
