// TEST INFRASTRUCTURE: the copy workers of the host-pointer pipeline (babyjubjub-rs_amd/csrc/copy_pool.hpp) driven the way
// run_pipelined drives them -- copy-in groups one chunk ahead, copy-out groups harvested in order, a 4-deep ring of staging
// buffers that is recycled -- on the CPU, under ThreadSanitizer / AddressSanitizer (tests/test_emul_sanitizers.py).
// Exit code 0 = every byte arrived; prints "copy_pool ok <bytes>".
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../babyjubjub-rs_amd/csrc/copy_pool.hpp"

int main(int argc, char** argv) {
  const int workers = argc > 1 ? atoi(argv[1]) : 4;
  const size_t n = argc > 2 ? (size_t)atoll(argv[2]) : ((size_t)37 << 20) + 12345;   // bytes; not a multiple of the slice
  const size_t chunk = ((size_t)5 << 20) + 77;
  const int NB = 4;
  std::vector<uint8_t> src(n), dst(n, 0);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; src[i] = (uint8_t)x; }
  std::vector<std::vector<uint8_t>> ring((size_t)NB, std::vector<uint8_t>(chunk));
  const size_t nchunks = (n + chunk - 1) / chunk;
  std::vector<CopyGroup> g_in(nchunks), g_out(nchunks);
  {
    CopyPool pool;
    if (!pool.start(workers)) { fprintf(stderr, "no worker thread\n"); return 2; }
    auto lo = [&](size_t ch) { return ch * chunk; };
    auto cnt = [&](size_t ch) { return lo(ch) + chunk <= n ? chunk : n - lo(ch); };
    size_t next_in = 0, harvested = 0;
    for (size_t ch = 0; ch < nchunks; ch++) {
      while (next_in < nchunks && next_in <= ch + 1) {
        if (next_in >= (size_t)NB) {             // the ring slot must be free: its previous chunk copied out completely
          const size_t old = next_in - NB;
          while (harvested <= old) { pool.submit(dst.data() + lo(harvested), ring[harvested % NB].data(), cnt(harvested), &g_out[harvested]); harvested++; }
          pool.wait(&g_out[old]);
        }
        pool.submit(ring[next_in % NB].data(), src.data() + lo(next_in), cnt(next_in), &g_in[next_in]);
        next_in++;
      }
      pool.wait(&g_in[ch]);                       // "enqueue": the device would now read ring[ch % NB]
      for (size_t i = 0; i < cnt(ch); i += 4097) ring[ch % NB][i] ^= 0x5a;   // the "kernel" touches the buffer on this thread
    }
    while (harvested < nchunks) { pool.submit(dst.data() + lo(harvested), ring[harvested % NB].data(), cnt(harvested), &g_out[harvested]); harvested++; }
    for (size_t ch = 0; ch < nchunks; ch++) pool.wait(&g_out[ch]);
    // an empty group and a zero-byte submit return at once
    CopyGroup e;
    pool.submit(dst.data(), src.data(), 0, &e);
    pool.wait(&e);
  }   // ~CopyPool joins the workers
  for (size_t ch = 0; ch < nchunks; ch++)
    for (size_t i = 0; i < (ch * chunk + chunk <= n ? chunk : n - ch * chunk); i++) {
      const uint8_t want = (uint8_t)(src[ch * chunk + i] ^ (i % 4097 == 0 ? 0x5a : 0));
      if (dst[ch * chunk + i] != want) { fprintf(stderr, "mismatch at chunk %zu byte %zu\n", ch, i); return 1; }
    }
  printf("copy_pool ok %zu\n", n);
  return 0;
}
