// Stand-alone two-kernel reproducer for profiles/r02/packed_fp32_mfma_interference.md (round 3):
// does a wave executing v_pk_fma_f32 return wrong sums when it shares a SIMD with a matrix-core wave of ANOTHER kernel?
//   victim  : every thread runs the same FMA chain twice, once with v_pk_fma_f32 and once with v_fma_f32 (both exact
//             fused multiply-adds: bit-identical by definition), and counts lanes whose two results differ.  No LDS, no
//             memory traffic in the loop: only the packed-fp32 datapath is exercised.  64 VGPRs: up to four victim waves
//             fit on a SIMD beside one 256-register matrix wave.
//   corunner: 256 threads, 256 VGPRs (one wave per SIMD and block, like conv_mfma_f8_kernel<1>), 75 KB of LDS, a dense
//             loop of v_mfma_f32_32x32x16_f16 + v_mfma_scale_f32_32x32x64_f8f6f4 (fp8) with LDS-DMA pieces in between.
// Three streams as in the finding: one of co-runners, two of victims, back to back for ~2 s per mode.
// build: hipcc --offload-arch=gfx950 -O3 -o pkfma_vs_mfma pkfma_vs_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void victim(unsigned* bad, int iters, int use_lds) {
    __shared__ float tab[1024];
    const int t = threadIdx.x + blockIdx.x * 256;
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = 1.f + (float)((i * 7) % 13) * 0.03125f;
    __syncthreads();
    f2 p[6], m = {0.999f, 1.0009f}, c = {0.125f, -0.0625f};
    float s[12];
    for (int i = 0; i < 6; ++i) { p[i].x = 1.f + (float)((t + i) % 97) * 0.01f; p[i].y = 2.f - (float)((t * 3 + i) % 89) * 0.01f; s[2 * i] = p[i].x; s[2 * i + 1] = p[i].y; }
    for (int it = 0; it < iters; ++it) {
        if (use_lds) { const float w = tab[(it * 17 + threadIdx.x) & 1023]; m.x = w * 0.999f; m.y = w; }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[2 * i]) : "v"(m.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[2 * i + 1]) : "v"(m.y), "v"(c.y));
        }
        if ((it & 63) == 63)
            for (int i = 0; i < 6; ++i) {     // compare and re-seed (keeps the chain in range)
                if (__float_as_uint(p[i].x) != __float_as_uint(s[2 * i]) || __float_as_uint(p[i].y) != __float_as_uint(s[2 * i + 1])) atomicAdd(bad, 1u);
                p[i].x = s[2 * i] = 1.f + (float)((t + i + it) % 97) * 0.01f;
                p[i].y = s[2 * i + 1] = 2.f - (float)((t * 3 + i + it) % 89) * 0.01f;
            }
    }
}

__global__ __launch_bounds__(256, 2) void corunner(const char* src, float* sink, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f32x16 acc[12];                                   // 192 accumulator registers + operands: 256 VGPRs, one wave per SIMD
    for (int k = 0; k < 12; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    half8 a, b; v8i a8, b8;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((float)((threadIdx.x + i) % 17) * 0.125f - 1.f); b[i] = (_Float16)((float)((threadIdx.x * 3 + i) % 13) * 0.25f - 1.5f); a8[i] = 0x38383838 ^ (threadIdx.x * 2654435761u + i); b8[i] = a8[i] * 31 + 7; }
    asm volatile("" : "+v"(a), "+v"(b), "+v"(a8), "+v"(b8));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
            if (mode >= 1 && (k & 1)) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[k], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (mode >= 2 && (k & 3) == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((it * 12 + k) & 63) * 4096 + threadIdx.x * 16),
                                                 (__attribute__((address_space(3))) void*)(lds + ((k >> 2) * 4 + wave) * 1024), 16, 0, 0);
        }
        if (mode >= 2) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); a[0] = *reinterpret_cast<_Float16*>(lds + (threadIdx.x & 63) * 2); }
    }
    float s = 0.f;
    for (int k = 0; k < 12; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (s == 123.456f) sink[0] = s;
}

int main() {
    unsigned* bad; char* src; float* sink;
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&src, 64 * 4096 + 4096)); CK(hipMalloc(&sink, 4)); CK(hipMemset(src, 0x3c, 64 * 4096 + 4096));
    CK(hipFuncSetAttribute((const void*)corunner, hipFuncAttributeMaxDynamicSharedMemorySize, 75 * 1024));
    hipStream_t st[3];
    for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    const char* names[4] = {"victims alone", "beside f16 MFMA waves", "beside f16 + fp8-scaled MFMA waves", "beside f16 + fp8 MFMA waves with LDS-DMA"};
    for (int lds = 0; lds < 2; ++lds)
        for (int mode = -1; mode < 3; ++mode) {
            CK(hipMemset(bad, 0, 4));
            for (int rep = 0; rep < 40; ++rep) {
                if (mode >= 0) hipLaunchKernelGGL(corunner, dim3(512), dim3(256), 75 * 1024, st[0], src, sink, 6000, mode);
                for (int v = 1; v < 3; ++v)
                    for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(victim, dim3(1024), dim3(256), 0, st[v], bad, 20000, lds);
            }
            CK(hipDeviceSynchronize());
            unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            printf("victim %s, %-42s: %u mismatching (packed vs scalar) checks of %.3g\n", lds ? "with LDS reads" : "registers only ", names[mode + 1], h,
                   40.0 * 8 * 1024 * 256 * 6 * (20000 / 64));
        }
    return 0;
}
