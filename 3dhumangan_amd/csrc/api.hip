// libh3d: version / error plumbing of the C ABI (include/h3d.h).
#include "common.hpp"
#include <string.h>

namespace h3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int compute_units() {
    static int cached[64] = {0};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cached[dev & 63];
    if (c) return c;
    c = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    return c;
}

}  // namespace h3d

extern "C" int h3d_version(void) { return H3D_VERSION; }

extern "C" const char* h3d_last_error(void) { return h3d::g_err; }

extern "C" int h3d_device_info(int* n_cu, char* name, int name_len) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        h3d::set_error("h3d_device_info: no HIP device");
        return H3D_ELAUNCH;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (name && name_len > 0) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return H3D_OK;
}
