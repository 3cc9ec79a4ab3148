// mailbox.hpp -- a few counters from the device to the host without a copy command: a one-thread kernel writes them into
// page-locked host memory the device can address and bumps a sequence word behind a system-scope fence; the host polls that word.
// A D2H hipMemcpyAsync + hipStreamQuery loop costs 30-40 us per round trip on this stack (copy command, completion signal, query);
// the mailbox costs the kernel launch (~3 us on the stream) and a PCIe write.
#pragma once
#include <hip/hip_runtime_api.h>

#include <sched.h>
#include <time.h>

#include <chrono>
#include <cstdint>

namespace calitas {

// One step of a host wait on the critical path: pure spinning for the first ~50 us (a round trip through the mailbox is 10-30 us and a
// blocking wait would add its wake-up latency to each of them), then the core is offered to whoever else can run on it (eight ranks
// of a job, each with a caller and a lane thread, may share sixteen cores), and from 5 ms on -- a long kernel, a 20 GB copy -- the
// thread sleeps between looks.
struct Backoff {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  long long waited_us() const { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
  void pause() {
    if (++spins < 256) { __builtin_ia32_pause(); return; }   // (the first ~10 us: not even a look at the clock)
    const long long us = waited_us();
    if (us < 50) __builtin_ia32_pause();
    else if (us < 5000) sched_yield();
    else { timespec ts{0, 50000}; nanosleep(&ts, nullptr); }
  }
};

constexpr int MAILBOX_WORDS = 21;

struct Mailbox {
  volatile uint32_t* host = nullptr;   // [0] sequence, [1..] payload
  uint32_t* dev = nullptr;             // the same memory as the device sees it
  uint32_t seq = 0;
};

hipError_t mailbox_open(Mailbox& mb);
void mailbox_close(Mailbox& mb);
// queues the publication of src[0..n) (device memory, n <= MAILBOX_WORDS) on `stream`
hipError_t mailbox_post(Mailbox& mb, const uint32_t* src, int n, hipStream_t stream);
// waits for the last post; the payload is then in mb.host[1..].  Also watches the stream: a failed kernel ends the wait with its error.
hipError_t mailbox_wait(Mailbox& mb, hipStream_t stream);

}  // namespace calitas
