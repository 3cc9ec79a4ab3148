// How long faulting in and freeing a multi-gigabyte malloc block takes with and without MADV_HUGEPAGE, and what /proc says about it.
// g++ -O2 -pthread tools/thp_bench.cpp -o /tmp/thp_bench && /tmp/thp_bench [GB]
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static long anon_huge_kb() {
  FILE* f = std::fopen("/proc/self/smaps_rollup", "r"); char line[256]; long v = -1;
  while (f && std::fgets(line, sizeof line, f)) if (std::sscanf(line, "AnonHugePages: %ld kB", &v) == 1) break;
  if (f) std::fclose(f);
  return v;
}
int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? std::atoi(argv[1]) : 8, n = gb << 30;
  for (int mode = 0; mode < 5; mode++) {
    double t0 = now();
    char* p = (mode == 2 || mode == 4) ? (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0) : (char*)std::malloc(n);
    if (mode >= 1) madvise((void*)(((uintptr_t)p + 4095) & ~(uintptr_t)4095), n - 8192, MADV_HUGEPAGE);
    double t1 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < 16; t++) th.emplace_back([=] { std::memset(p + n / 16 * t, 1, n / 16); });
    for (auto& x : th) x.join();
    double t2 = now();
    const long huge = anon_huge_kb();
    if (mode >= 3) {   // hand the pages back from 16 threads first (MADV_DONTNEED takes the address-space lock shared)
      std::vector<std::thread> tf;
      const uintptr_t lo = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)p + n) & ~(uintptr_t)4095;
      const size_t per = ((hi - lo) / 16 + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
      for (int t = 0; t < 16; t++) tf.emplace_back([=] { uintptr_t a = lo + per * t, b = a + per < hi ? a + per : hi; if (a < b) madvise((void*)a, b - a, MADV_DONTNEED); });
      for (auto& x : tf) x.join();
    }
    if (mode == 2 || mode == 4) munmap(p, n); else std::free(p);
    double t3 = now();
    std::printf("%s %zu GB: alloc %.3f s, touch (16 threads) %.3f s, free %.3f s, AnonHugePages %ld MB\n",
                mode == 0 ? "malloc" : mode == 1 ? "malloc+MADV_HUGEPAGE" : mode == 2 ? "mmap+MADV_HUGEPAGE" : mode == 3 ? "malloc+HUGEPAGE, parallel DONTNEED" : "mmap+HUGEPAGE, parallel DONTNEED", gb, t1 - t0, t2 - t1, t3 - t2, huge >> 10);
  }
}
