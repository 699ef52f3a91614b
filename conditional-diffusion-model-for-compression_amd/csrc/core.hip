// ABI version, status strings, launch counter.
#include <atomic>

#include "common.h"

namespace cdx {
static std::atomic<uint64_t> g_launches{0};
void count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
}  // namespace cdx

extern "C" int cdx_abi_version(void) { return CDX_ABI_VERSION; }

extern "C" uint64_t cdx_launch_count(void) { return cdx::g_launches.load(std::memory_order_relaxed); }

extern "C" const char* cdx_strerror(int status) {
    switch (status) {
        case CDX_OK: return "ok";
        case CDX_EINVAL: return "CDX_EINVAL: bad shape, null or misaligned pointer";
        case CDX_ENOSPC: return "CDX_ENOSPC: workspace too small";
        case CDX_ELAUNCH: return "CDX_ELAUNCH: kernel launch failed";
        case CDX_ENOTSUP: return "CDX_ENOTSUP: no kernel built for this request";
        default: return "unknown cdx status";
    }
}
