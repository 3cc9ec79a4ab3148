// dma.cpp -- see dma.hpp.
#include "dma.hpp"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <cstdint>

namespace calitas {

bool DmaCopier::open(int /*device*/) {
  // reference counted; HIP initialised the runtime already.  The agents are taken from the buffers themselves at copy time, so no
  // assumption is made about how HIP's device ordinals map to HSA's agent enumeration (HIP_VISIBLE_DEVICES reorders one, not the other).
  ok_ = hsa_init() == HSA_STATUS_SUCCESS;
  return ok_;
}

thread_local const char* g_dma_reason = "";
const char* DmaCopier::last_reason() { return g_dma_reason; }

bool DmaCopier::copy_to_host(void* dst_host, const void* src_dev, size_t n) const {
  if (n == 0) { g_dma_reason = ""; return ok_; }
  const unsigned long long t = start(dst_host, src_dev, n);
  return t != 0 && finish(t);
}

bool DmaCopier::finish(unsigned long long ticket) const {
  hsa_signal_t sig;
  sig.handle = ticket;
  hsa_signal_value_t v;
  do { v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE); } while (v >= 1);
  hsa_signal_destroy(sig);
  return v == 0;                                            // negative: the runtime reports a failed copy
}

unsigned long long DmaCopier::start(void* dst_host, const void* src_dev, size_t n) const {
  std::vector<unsigned long long> t;
  if (n == 0 || !start_pieces(dst_host, src_dev, n, n, t)) return 0;
  return t[0];
}

bool DmaCopier::start_pieces(void* dst_host, const void* src_dev, size_t n, size_t piece, std::vector<unsigned long long>& tickets) const {
  g_dma_reason = "";
  tickets.clear();
  if (!ok_ || n == 0 || piece == 0) return false;
  hsa_amd_pointer_info_t si{}, di{};
  si.size = sizeof(si); di.size = sizeof(di);
  if (hsa_amd_pointer_info(const_cast<void*>(src_dev), &si, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
      hsa_amd_pointer_info(dst_host, &di, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS)
    { g_dma_reason = "pointer info unavailable"; return false; }
  // both ends must be allocations of the runtime itself (hipMalloc / hipHostMalloc).  A host range that is merely *locked* -- which
  // is what the runtime's own pageable copies leave behind, possibly mapped read-only -- is left to hipMemcpyAsync.
  if (si.type != HSA_EXT_POINTER_TYPE_HSA || di.type != HSA_EXT_POINTER_TYPE_HSA) {
    g_dma_reason = di.type == HSA_EXT_POINTER_TYPE_LOCKED ? "destination is a locked pageable range" : "an end is not a runtime allocation";
    return false;
  }
  {
    const char* base = (const char*)(di.hostBaseAddress ? di.hostBaseAddress : di.agentBaseAddress);
    if ((const char*)dst_host < base || (const char*)dst_host + n > base + di.sizeInBytes) { g_dma_reason = "destination range leaves its allocation"; return false; }
  }
  if ((const char*)src_dev < (const char*)si.agentBaseAddress || (const char*)src_dev + n > (const char*)si.agentBaseAddress + si.sizeInBytes) {
    g_dma_reason = "source range leaves its allocation";
    return false;
  }
  hsa_device_type_t st, dt;
  if (hsa_agent_get_info(si.agentOwner, HSA_AGENT_INFO_DEVICE, &st) != HSA_STATUS_SUCCESS || st != HSA_DEVICE_TYPE_GPU) return false;
  if (hsa_agent_get_info(di.agentOwner, HSA_AGENT_INFO_DEVICE, &dt) != HSA_STATUS_SUCCESS || dt != HSA_DEVICE_TYPE_CPU) return false;
  for (size_t off = 0; off < n; off += piece) {
    const size_t len = std::min(piece, n - off);
    hsa_signal_t sig;
    bool ok = hsa_signal_create(1, 0, nullptr, &sig) == HSA_STATUS_SUCCESS;
    if (ok && (hsa_amd_memory_async_copy((char*)dst_host + off, di.agentOwner, (const char*)src_dev + off, si.agentOwner, len, 0, nullptr, sig) != HSA_STATUS_SUCCESS ||
               sig.handle == 0)) {
      hsa_signal_destroy(sig);
      ok = false;
    }
    if (!ok) {
      for (unsigned long long t : tickets) (void)finish(t);   // (what is on its way lands before the caller frees anything)
      tickets.clear();
      g_dma_reason = "hsa_amd_memory_async_copy failed";
      return false;
    }
    tickets.push_back(sig.handle);
  }
  return true;
}

}  // namespace calitas
