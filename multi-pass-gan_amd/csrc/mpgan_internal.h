// Internal helpers shared by the HIP translation units of libmpgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "mpgan.h"

namespace mpg {

void set_error(const char* fmt, ...);

inline int hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MPG_ERR_HIP;
    }
    return MPG_OK;
}

#define MPG_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            mpg::set_error(__VA_ARGS__);  \
            return MPG_ERR_ARG;           \
        }                                 \
    } while (0)

#define MPG_LAUNCH_CHECK(name) return mpg::hip_check(hipGetLastError(), name)

// Zero `bytes` (a multiple of 4) of device memory with a kernel on `stream`.  Used instead of
// hipMemsetAsync for buffers that atomics accumulate into: memset nodes of a captured hipGraph were
// observed to lose their ordering against the neighbouring kernels on replay (ROCm 7.2), a plain
// kernel node does not.
hipError_t zero_async(void* ptr, size_t bytes, hipStream_t stream);

// 256 zero bytes of device memory (per device, allocated on first use outside any capture): the source of LDS-DMA lanes
// that must deliver zeros.  nullptr when the allocation failed.
const char* zero_page();

__device__ __forceinline__ float apply_act(float v, int act, float leak) {
    // MPG_ACT_RELU: tf.nn.relu; MPG_ACT_LRELU: 0.5(1+leak) x + 0.5(1-leak)|x| (GAN.py:733-737)
    if (act == MPG_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MPG_ACT_LRELU) return 0.5f * (1.f + leak) * v + 0.5f * (1.f - leak) * fabsf(v);
    if (act == MPG_ACT_TANH) return tanhf(v);
    return v;
}

// power-of-two scale that brings a tensor with absolute maximum `amax` into [2^8, 2^9): fp16 hi/lo splits of
// gradients (1e-4 .. 1e-8 in magnitude) would otherwise land in the fp16 subnormal range
__device__ __forceinline__ float pow2_scale(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    frexpf(amax, &e);
    return ldexpf(1.f, 9 - e);
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device): hipFuncSetAttribute is a driver call and the
// value never has to shrink, so a launch only pays for it the first time a size above the recorded one is asked.
// `slot` is a per-kernel static array of 64 ints (one per device), zero-initialised.
inline hipError_t ensure_dyn_lds(const void* kern, int bytes, int* slot) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    int seen = __atomic_load_n(&slot[dev], __ATOMIC_ACQUIRE);
    if (bytes <= seen) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    while (seen < bytes && !__atomic_compare_exchange_n(&slot[dev], &seen, bytes, false, __ATOMIC_RELEASE, __ATOMIC_ACQUIRE)) {
    }
    return hipSuccess;
}

}  // namespace mpg
