// Error state and device probe of libmpgan_hip.so.
#include "mpgan_internal.h"

#include <cstring>

namespace mpg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace mpg

extern "C" const char* mpg_last_error(void) { return mpg::g_err; }

extern "C" const char* mpg_version(void) { return "mpgan-hip 0.1 (gfx950)"; }

extern "C" int mpg_device_info(int* cu_count, char* arch_name, int arch_name_len) {
    int dev = 0;
    if (int rc = mpg::hip_check(hipGetDevice(&dev), "hipGetDevice")) return rc;
    hipDeviceProp_t prop;
    if (int rc = mpg::hip_check(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties")) return rc;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mpg::set_error("device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        return MPG_ERR_UNSUPPORTED;
    }
    return MPG_OK;
}

namespace {
__global__ void zero_words_kernel(unsigned int* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace

hipError_t mpg::zero_async(void* ptr, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned int*)ptr, n);
    return hipGetLastError();
}
