// dma.cpp -- see dma.hpp.
#include "dma.hpp"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <cstdint>

namespace calitas {

bool DmaCopier::open(int /*device*/) {
  // reference counted; HIP initialised the runtime already.  The agents are taken from the buffers themselves at copy time, so no
  // assumption is made about how HIP's device ordinals map to HSA's agent enumeration (HIP_VISIBLE_DEVICES reorders one, not the other).
  ok_ = hsa_init() == HSA_STATUS_SUCCESS;
  return ok_;
}

bool DmaCopier::copy_to_host(void* dst_host, const void* src_dev, size_t n) const {
  if (!ok_) return false;
  if (n == 0) return true;
  hsa_amd_pointer_info_t si{}, di{};
  si.size = sizeof(si); di.size = sizeof(di);
  if (hsa_amd_pointer_info(const_cast<void*>(src_dev), &si, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
      hsa_amd_pointer_info(dst_host, &di, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS)
    return false;
  if (si.type != HSA_EXT_POINTER_TYPE_HSA || (di.type != HSA_EXT_POINTER_TYPE_HSA && di.type != HSA_EXT_POINTER_TYPE_LOCKED)) return false;
  hsa_device_type_t st, dt;
  if (hsa_agent_get_info(si.agentOwner, HSA_AGENT_INFO_DEVICE, &st) != HSA_STATUS_SUCCESS || st != HSA_DEVICE_TYPE_GPU) return false;
  if (hsa_agent_get_info(di.agentOwner, HSA_AGENT_INFO_DEVICE, &dt) != HSA_STATUS_SUCCESS || dt != HSA_DEVICE_TYPE_CPU) return false;
  hsa_signal_t sig;
  if (hsa_signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) return false;
  bool good = hsa_amd_memory_async_copy(dst_host, di.agentOwner, src_dev, si.agentOwner, n, 0, nullptr, sig) == HSA_STATUS_SUCCESS;
  if (good) {
    hsa_signal_value_t v;
    do { v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE); } while (v >= 1);
    good = v == 0;                                          // negative: the runtime reports a failed copy
  }
  hsa_signal_destroy(sig);
  return good;
}

}  // namespace calitas
