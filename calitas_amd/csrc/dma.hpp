// dma.hpp -- device -> host copies on the SDMA engines, through the HSA runtime underneath HIP.
//
// hipMemcpyAsync from device memory to device-visible pinned host memory is carried out by a blit kernel
// (__amd_rocclr_copyBuffer in a rocprofv3 trace): it shares the CUs with the search kernels of the other lanes for as long as
// PCIe takes.  hsa_amd_memory_async_copy between the GPU agent and a CPU agent uses a DMA engine instead and leaves the CUs alone.
#pragma once
#include <cstddef>
#include <vector>

namespace calitas {

// One per context.  open() attaches to the HSA runtime; a copy takes its agents from the two buffers (the GPU that owns the source
// allocation, the CPU that owns the page-locked destination) and returns false when anything is not as expected -- callers then
// use hipMemcpyAsync.
class DmaCopier {
 public:
  bool open(int device);
  bool usable() const { return ok_; }
  // Blocking copy of n bytes from device memory to page-locked host memory; the source must be complete (the caller waited for the
  // producing kernels).  Thread-safe: every call uses its own completion signal.  Returns false on failure.
  bool copy_to_host(void* dst_host, const void* src_dev, size_t n) const;
  // The same in two halves, so that the caller can work on what has arrived while the next piece is on the bus: start() queues the
  // copy and returns a ticket (0: declined, see last_reason), finish() waits for it (false: the runtime reports a failed copy).
  unsigned long long start(void* dst_host, const void* src_dev, size_t n) const;
  bool finish(unsigned long long ticket) const;
  // One copy as pieces of `piece` bytes queued back to back (a ticket each, in the order of the bytes): the caller hands the pieces on
  // as they land.  false: declined, nothing is on its way.
  bool start_pieces(void* dst_host, const void* src_dev, size_t n, size_t piece, std::vector<unsigned long long>& tickets) const;
  static const char* last_reason();   // why the last copy_to_host of this thread declined ("" otherwise)

 private:
  bool ok_ = false;
};

}  // namespace calitas
