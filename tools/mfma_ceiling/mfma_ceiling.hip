// Sustained MFMA rate of gfx950 for the instruction mixes of the fused convolutions, operands in registers
// only (no LDS, no memory): the ceiling the K loops can be compared with.  Development probe, not part of
// the library.   hipcc --offload-arch=gfx950 -O3 mfma_ceiling.hip -o mfma_ceiling && ./mfma_ceiling
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

// MIX 0: fp16 32x32x16 only; 1: fp8 (MX scale) 32x32x64 only; 2: 4 fp16 + 2 fp8 per 64 K (MPG_PREC_F16F8);
template <int MIX, int TILES>
__global__ __launch_bounds__(512, 2) void mfma_loop(int iters, float* out) {
    f32x16 acc[TILES];
    for (int t = 0; t < TILES; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    half8 a, b;
    v8i a8, b8;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * (threadIdx.x + i));
        b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i));
        a8[i] = 0x38383838 + (int)threadIdx.x;
        b8[i] = 0x30303030 + i;
    }
    for (int it = 0; it < iters; ++it) {
        if (MIX == 0 || MIX == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        }
        if (MIX == 1 || MIX == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    float s = 0.f;
    for (int t = 0; t < TILES; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 123.456f) out[0] = s;
}

template <int MIX>
static void run(const char* name, int iters, float* out) {
    const int blocks = 256, threads = 512;                  // one block of 8 waves per CU, 2 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<MIX, 8>), dim3(blocks), dim3(threads), 0, 0, iters / 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mfma_loop<MIX, 8>), dim3(blocks), dim3(threads), 0, 0, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // fp16-equivalent units: a 32x32x16 fp16 MFMA = 1 unit = 32768 flop at the 2.5 PFLOP/s dense fp16 peak;
    // a 32x32x64 fp8 MFMA = 2 units (4x the MACs at twice the rate)
    const double units_per_iter = (MIX == 0 ? 4 : MIX == 1 ? 4 : 8) * 8.0;
    const double units = units_per_iter * iters * (double)blocks * (threads / 64);
    const double tf = units * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-34s %8.3f ms  %8.1f TFLOP/s fp16-equivalent issue  (%.3f of 2500)\n", name, ms, tf, tf / 2500.0);
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("fp16 32x32x16", 20000, out);
        run<1>("fp8 MX 32x32x64", 20000, out);
        run<2>("4 fp16 + 2 fp8 per K=64 (F16F8)", 10000, out);
    }
    // sustained over ~1 s: clocks settle under the power cap
    run<2>("F16F8 mix, long run", 400000, out);
    run<0>("fp16, long run", 800000, out);
    return 0;
}
