// Library-level entry points of libppoaf_hip.so: ABI version, error string,
// device query.  See include/ppoaf_hip.h.
#include "common.hpp"

#include <cstring>

namespace ppoaf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace ppoaf

extern "C" int ppoaf_abi_version(void) { return PPOAF_ABI_VERSION; }

extern "C" const char* ppoaf_last_error(void) { return ppoaf::g_err; }

extern "C" int ppoaf_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { ppoaf::set_error("no HIP device"); return PPOAF_E_LAUNCH; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { ppoaf::set_error("hipGetDeviceProperties failed"); return PPOAF_E_LAUNCH; }
    return prop.multiProcessorCount;
}
