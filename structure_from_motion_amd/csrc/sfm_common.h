// Shared host/device helpers of the libsfm_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sfm_hip.h"

namespace sfmhost {

// thread-local last-error text shared by all translation units (defined in sfm_kernels.hip)
char* error_buffer();
constexpr int kErrorBytes = 512;

inline int fail(int code, const char* msg) {
    snprintf(error_buffer(), kErrorBytes, "%s", msg);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(error_buffer(), kErrorBytes, "%s: %s", what, hipGetErrorString(err));
        return SFM_EHIP;
    }
    return SFM_OK;
}

inline unsigned grid_for(int64_t work, int block, int64_t cap = 1 << 20) {
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace sfmhost

constexpr int kWave = 64;

// One correspondence in K-normalised coordinates: 32 bytes, read as two 16-byte loads.
struct alignas(32) Corr {
    double xa, ya, xb, yb;
};
