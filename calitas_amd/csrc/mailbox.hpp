// mailbox.hpp -- a few counters from the device to the host without a copy command: a one-thread kernel writes them into
// page-locked host memory the device can address and bumps a sequence word behind a system-scope fence; the host polls that word.
// A D2H hipMemcpyAsync + hipStreamQuery loop costs 30-40 us per round trip on this stack (copy command, completion signal, query);
// the mailbox costs the kernel launch (~3 us on the stream) and a PCIe write.
#pragma once
#include <hip/hip_runtime_api.h>

#include "parallel.hpp"   // Backoff

#include <sched.h>
#include <time.h>

#include <chrono>
#include <cstdint>

namespace calitas {

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
