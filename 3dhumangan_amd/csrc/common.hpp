// Shared host/device helpers for libh3d (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/h3d.h"

namespace h3d {

void set_error(const char* fmt, ...);

// Drop any stale (sticky) error another library left in the HIP runtime before we launch.
inline void pre_launch() { (void)hipGetLastError(); }

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return H3D_ELAUNCH;
    }
    return H3D_OK;
}

#define H3D_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) {                                 \
            h3d::set_error(__VA_ARGS__);               \
            return H3D_EINVAL;                         \
        }                                              \
    } while (0)

constexpr int kWave = 64;   // CDNA wavefront

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Number of CUs of the current device (cached per device; falls back to the MI355X value).
int compute_units();

// Function attributes are per device: raise the dynamic-LDS limit of `kernel` to the full 160 KB once per
// (expansion site = kernel instantiation, device).  Idempotent, so a race between two host threads is harmless.
#define H3D_ALLOW_MAX_LDS(kernel)                                                                         \
    do {                                                                                                  \
        static unsigned long long done_ = 0;                                                              \
        int dev_ = 0;                                                                                     \
        (void)hipGetDevice(&dev_);                                                                        \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                              \
        if (!(done_ & bit_)) {                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);            \
            done_ |= bit_;                                                                                \
        }                                                                                                 \
    } while (0)

}  // namespace h3d
