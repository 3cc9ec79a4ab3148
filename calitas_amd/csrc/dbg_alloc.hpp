// dbg_alloc.hpp -- debugging aid, compiled in only with `make DBG_ALLOC=1` (-DCALITAS_ALLOC_DEBUG).
//
// Records every hipMalloc / hipHostMalloc / hipFree / hipHostFree this library makes and, when the process aborts (the HIP
// runtime aborts after printing "Memory access fault by GPU ... on address X"), prints the live allocations and the most recent
// frees sorted by address, so that X can be placed next to the buffer it overran or the buffer it outlived.
#pragma once
#include <hip/hip_runtime.h>

#include <csignal>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

namespace calitas_dbg {

struct Entry { size_t size; const char* file; int line; bool host; uint64_t serial; };
struct Freed { uintptr_t p; Entry e; uint64_t at; const char* file; int line; };

struct Registry {
  std::mutex mu;
  std::map<uintptr_t, Entry> live;
  std::vector<Freed> freed;     // ring
  size_t freed_next = 0;
  uint64_t serial = 0;
  bool hooked = false;
};
inline Registry& reg() { static Registry* r = new Registry(); return *r; }

inline void dump(int) {
  Registry& r = reg();
  std::fprintf(stderr, "[calitas-dbg] %zu live allocations (serial %llu):\n", r.live.size(), (unsigned long long)r.serial);
  for (auto& kv : r.live)
    std::fprintf(stderr, "[calitas-dbg]   live %s %012llx .. %012llx (%zu bytes) #%llu %s:%d\n", kv.second.host ? "host" : "dev ",
                 (unsigned long long)kv.first, (unsigned long long)(kv.first + kv.second.size), kv.second.size,
                 (unsigned long long)kv.second.serial, kv.second.file, kv.second.line);
  for (auto& f : r.freed)
    std::fprintf(stderr, "[calitas-dbg]   freed@%llu %s %012llx .. %012llx (%zu bytes) #%llu %s:%d by %s:%d\n", (unsigned long long)f.at,
                 f.e.host ? "host" : "dev ", (unsigned long long)f.p, (unsigned long long)(f.p + f.e.size), f.e.size,
                 (unsigned long long)f.e.serial, f.e.file, f.e.line, f.file, f.line);
  std::fflush(stderr);
  std::signal(SIGABRT, SIG_DFL);
}

inline void note_alloc(void* p, size_t n, bool host, const char* file, int line) {
  Registry& r = reg();
  std::lock_guard<std::mutex> lk(r.mu);
  if (!r.hooked) { std::signal(SIGABRT, dump); r.hooked = true; }
  r.live[(uintptr_t)p] = Entry{n, file, line, host, ++r.serial};
}
inline void note_free(void* p, const char* file, int line) {
  if (!p) return;
  Registry& r = reg();
  std::lock_guard<std::mutex> lk(r.mu);
  auto it = r.live.find((uintptr_t)p);
  if (it == r.live.end()) return;
  Freed f{(uintptr_t)p, it->second, ++r.serial, file, line};
  r.freed.push_back(f);   // everything: a stale pointer may be many contexts old
  r.live.erase(it);
}

inline hipError_t d_malloc(void** p, size_t n, const char* file, int line) {
  hipError_t e = ::hipMalloc(p, n);
  if (e == hipSuccess) note_alloc(*p, n, false, file, line);
  return e;
}
inline hipError_t h_malloc(void** p, size_t n, unsigned flags, const char* file, int line) {
  hipError_t e = ::hipHostMalloc(p, n, flags);
  if (e == hipSuccess) note_alloc(*p, n, true, file, line);
  return e;
}
inline hipError_t d_free(void* p, const char* file, int line) { note_free(p, file, line); return ::hipFree(p); }
inline hipError_t h_free(void* p, const char* file, int line) { note_free(p, file, line); return ::hipHostFree(p); }

}  // namespace calitas_dbg

#define hipMalloc(p, n) calitas_dbg::d_malloc((void**)(p), (n), __FILE__, __LINE__)
#define hipHostMalloc(p, n, f) calitas_dbg::h_malloc((void**)(p), (n), (f), __FILE__, __LINE__)
#define hipFree(p) calitas_dbg::d_free((void*)(p), __FILE__, __LINE__)
#define hipHostFree(p) calitas_dbg::h_free((void*)(p), __FILE__, __LINE__)
