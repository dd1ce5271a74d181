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

// channel-chunk plan of one conv segment; shared by the packer and the kernel launch
struct SegPlan {
    int kc;       // input channels per LDS chunk (multiple of 8)
    int g;        // 8-channel groups per chunk
    int nchunks;  // chunks over cin
    int sc;       // weight stages per chunk
    int ps;       // LDS bytes per halo pixel per plane
};

inline int pick_kc(int cin, int kc_max) {
    const int cin8 = (cin + 7) & ~7;
    if (kc_max <= 0) kc_max = 32;
    if (cin8 <= kc_max) return cin8;
    const int cand[4] = {32, 24, 16, 8};
    for (int i = 0; i < 4; ++i)
        if (cand[i] <= kc_max && cin8 % cand[i] == 0) return cand[i];
    return 8;
}

inline SegPlan make_plan(int kh, int kw, int cin, int kc_max, int ks) {
    SegPlan p;
    if (kh == 1 && kw == 1) kc_max = 32;   // no halo: a 256-pixel image, the widest chunk always fits
    p.kc = pick_kc(cin, kc_max);
    p.g = p.kc / 8;
    const int cin8 = (cin + 7) & ~7;
    p.nchunks = (cin8 + p.kc - 1) / p.kc;
    const int tg = kh * kw * p.g;          // 8-channel groups per chunk over all taps
    const int ksteps = (tg + 1) / 2;       // one MFMA k-step (K=16) eats two groups
    p.sc = (ksteps + ks - 1) / ks;
    p.ps = (p.g & 1) ? 16 * p.g : 16 * (p.g + 1);   // odd number of 16-B slots => conflict-free b128 reads
    return p;
}

}  // namespace mpg
