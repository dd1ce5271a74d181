// Stand-alone two-kernel reproducer for profiles/r02/packed_fp32_mfma_interference.md (round 3):
// does a wave executing v_pk_fma_f32 return wrong sums when it shares a SIMD with a matrix-core wave of ANOTHER kernel?
//   victim  : every thread runs the same FMA chain twice, once with v_pk_fma_f32 and once with v_fma_f32 (both exact
//             fused multiply-adds: bit-identical by definition), and counts lanes whose two results differ.  No LDS, no
//             memory traffic in the loop: only the packed-fp32 datapath is exercised.  64 VGPRs: up to four victim waves
//             fit on a SIMD beside one 256-register matrix wave.
//   corunner: 256 threads, 256 VGPRs (one wave per SIMD and block, like conv_mfma_f8_kernel<1>), 100 KB of LDS (one block, i.e. one matrix wave per SIMD, per CU), a dense
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

// victim 2: the instruction pattern of the failing build of conv_small_kernel<1,8> (hipcc -O3 with the SLP vectoriser):
// the 64-bit source of every v_pk_fma_f32 is assembled by two v_mov_b32 and REWRITTEN by the next two v_mov_b32 right
// behind the packed FMA that read it (write after read, back to back):
//     v_mov_b32 v134, a0 ; v_mov_b32 v135, b0 ; v_pk_fma_f32 acc, v[134:135], w, acc ; v_mov_b32 v134, a1 ; ...
// The same sums are formed with scalar v_fma_f32 from the same registers; wrong lanes are recorded by quarter wave.
// SEL = 0: op_sel_hi:[1,0,1], both halves take the LOW word of w (the form tried in round 3 first); SEL = 1: op_sel:[0,1,0],
// both halves take the HIGH word of w -- the one form whose in-place replacement by two v_fma_f32 makes the failing library
// build pass (tools/experiments/slp_asm_edit.py expand_form=sel010)
template <int SEL>
__global__ __launch_bounds__(256) void victim_war(unsigned* bad, int iters) {
    const int t = threadIdx.x + blockIdx.x * 256;
    float a[8], b[8];
    f2 w[4];
    for (int i = 0; i < 8; ++i) { a[i] = 0.5f + (float)((t * 7 + i * 3) % 61) * 0.015625f; b[i] = 1.5f - (float)((t * 5 + i * 11) % 53) * 0.015625f; }
    for (int i = 0; i < 4; ++i) { w[i].x = 0.25f + 0.03125f * i; w[i].y = -0.125f + 0.0625f * i; }
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    for (int it = 0; it < iters; ++it) {
        f2 acc = {0.f, 0.f};
        float s0 = 0.f, s1 = 0.f;
        // packed: acc.x += a[i] * w.x ; acc.y += b[i] * w.x  (op_sel_hi:[1,0,1]: both halves take the LOW word of w); the
        // source pair v[62:63] is rewritten right behind every packed FMA that read it, as in the failing build
#define STEP(i) "v_mov_b32 v62, %[a" #i "]\n\tv_mov_b32 v63, %[b" #i "]\n\tv_pk_fma_f32 %[acc], v[62:63], %[w" #i "], %[acc] op_sel_hi:[1,0,1]\n\t"
#define STEPH(i) "v_mov_b32 v62, %[a" #i "]\n\tv_mov_b32 v63, %[b" #i "]\n\tv_pk_fma_f32 %[acc], v[62:63], %[w" #i "], %[acc] op_sel:[0,1,0]\n\t"
        if (SEL == 1)
        asm volatile(STEPH(0) STEPH(1) STEPH(2) STEPH(3) STEPH(4) STEPH(5) STEPH(6) STEPH(7)
                     : [acc] "+v"(acc)
                     : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),
                       [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7]),
                       [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[0]), [w5] "v"(w[1]), [w6] "v"(w[2]), [w7] "v"(w[3])
                     : "v62", "v63");
        else
        asm volatile(STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
                     : [acc] "+v"(acc)
                     : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),
                       [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7]),
                       [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[0]), [w5] "v"(w[1]), [w6] "v"(w[2]), [w7] "v"(w[3])
                     : "v62", "v63");
#undef STEP
#undef STEPH
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(a[i]), "v"(SEL ? w[i & 3].y : w[i & 3].x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(b[i]), "v"(SEL ? w[i & 3].y : w[i & 3].x));
        }
        if (__float_as_uint(acc.x) != __float_as_uint(s0) || __float_as_uint(acc.y) != __float_as_uint(s1)) atomicAdd(&bad[1 + ((threadIdx.x & 63) >> 4)], 1u);
        a[it & 7] += 0.0078125f; b[(it + 3) & 7] -= 0.00390625f;
        if ((it & 255) == 255) for (int i = 0; i < 8; ++i) { a[i] = 0.5f + (float)((t * 7 + i * 3 + it) % 61) * 0.015625f; b[i] = 1.5f - (float)((t * 5 + i * 11 + it) % 53) * 0.015625f; }
    }
}

// victim 3: which part of the pattern is needed?  FORM 0: op_sel:[0,1,0] with fixed source pairs (no v_mov, nothing rewritten);
// FORM 1: op_sel:[1,0,0] (the HIGH word of source 0 for the low half); FORM 2: v_pk_mul_f32 op_sel:[0,1] + plain v_pk_add_f32;
// FORM 3: op_sel:[0,1,0] op_sel_hi:[1,0,1] (halves of source 1 crossed); FORM 4: v_pk_mov_b32 op_sel:[1,0] (low = source 0 HIGH word,
// high = source 1 LOW word: the form hipcc emits in this library's wgrad_ring_kernel); FORM 5: v_pk_mov_b32 op_sel:[0,1] (low = source 0
// low, high = source 1 high: the compiler's 64-bit register move).  Scalar twins from the same registers as above.  A thread stops
// after 64 wrong sums (a twin that does not match the instruction's meaning must not turn into 1e12 atomics).
template <int FORM>
__global__ __launch_bounds__(256) void victim_form(unsigned* bad, int iters) {
    const int t = threadIdx.x + blockIdx.x * 256;
    f2 ab[8], w[4];
    for (int i = 0; i < 8; ++i) { ab[i].x = 0.5f + (float)((t * 7 + i * 3) % 61) * 0.015625f; ab[i].y = 1.5f - (float)((t * 5 + i * 11) % 53) * 0.015625f; }
    for (int i = 0; i < 4; ++i) { w[i].x = 0.25f + 0.03125f * i; w[i].y = -0.125f + 0.0625f * i; }
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    int nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f2 acc = {0.f, 0.f}, prod;
        float s0 = 0.f, s1 = 0.f, q0, q1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(ab[i]), "v"(w[i & 3]));
            if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(ab[i]), "v"(w[i & 3]));
            if (FORM == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(prod) : "v"(ab[i]), "v"(w[i & 3]));
                             asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(prod)); }
            if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(ab[i]), "v"(w[i & 3]));
            if (FORM == 4) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(prod) : "v"(ab[i]), "v"(w[i & 3]));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc.x) : "v"(prod.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc.y) : "v"(prod.y)); }
            if (FORM == 5) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(prod) : "v"(ab[i]), "v"(w[i & 3]));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc.x) : "v"(prod.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc.y) : "v"(prod.y)); }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (FORM == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(ab[i].x), "v"(w[i & 3].y));
                             asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(ab[i].y), "v"(w[i & 3].y)); }
            if (FORM == 1) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(ab[i].y), "v"(w[i & 3].x));
                             asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(ab[i].y), "v"(w[i & 3].y)); }
            if (FORM == 2) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(ab[i].x), "v"(w[i & 3].y));
                             asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(ab[i].y), "v"(w[i & 3].y));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(q0));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(q1)); }
            if (FORM == 3) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(ab[i].x), "v"(w[i & 3].y));
                             asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(ab[i].y), "v"(w[i & 3].x)); }
            if (FORM == 4) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(ab[i].y)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(w[i & 3].x)); }
            if (FORM == 5) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(ab[i].x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(w[i & 3].y)); }
        }
        if (__float_as_uint(acc.x) != __float_as_uint(s0) || __float_as_uint(acc.y) != __float_as_uint(s1)) {
            atomicAdd(&bad[1 + ((threadIdx.x & 63) >> 4)], 1u);
            if (++nbad >= 64) break;
        }
        ab[it & 7].x += 0.0078125f; ab[(it + 3) & 7].y -= 0.00390625f;
        if ((it & 255) == 255) for (int i = 0; i < 8; ++i) { ab[i].x = 0.5f + (float)((t * 7 + i * 3 + it) % 61) * 0.015625f; ab[i].y = 1.5f - (float)((t * 5 + i * 11 + it) % 53) * 0.015625f; }
    }
}

// victim 4: WHAT does a wrong lane hold?  One v_pk_mul_f32 op_sel:[0,1] per check: low half = a.x * w.y, high half = a.y * w.y.
// cnt[0] low half wrong, cnt[1] of those equal to a.x * w.x (the op_sel bit of source 1 not applied: the LOW word was read),
// cnt[2] high half wrong, cnt[3] low-half errors outside lanes 48-63.
__global__ __launch_bounds__(256) void victim_value(unsigned* cnt, int iters, float* samples) {
    const int t = threadIdx.x + blockIdx.x * 256;
    f2 a = {0.5f + (float)(t % 61) * 0.015625f, 1.5f - (float)(t % 53) * 0.015625f}, w = {0.25f + 0.03125f * (float)(t & 3), -0.125f - 0.0625f * (float)(t & 7)};
    asm volatile("" : "+v"(w));
    for (int it = 0; it < iters; ++it) {
        f2 p; float lo, hi, alt;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(a), "v"(w));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(w.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(w.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(alt) : "v"(a.x), "v"(w.x));
        if (__float_as_uint(p.x) != __float_as_uint(lo)) {
            atomicAdd(&cnt[0], 1u);
            if (__float_as_uint(p.x) == __float_as_uint(alt)) atomicAdd(&cnt[1], 1u);
            if ((threadIdx.x & 63) < 48) atomicAdd(&cnt[3], 1u);
            const unsigned k = atomicAdd(&cnt[4], 1u);
            if (k < 24) { float* r = samples + 6 * k; r[0] = p.x; r[1] = a.x; r[2] = w.x; r[3] = w.y; r[4] = (float)(threadIdx.x & 63); r[5] = p.y; }
        }
        if (__float_as_uint(p.y) != __float_as_uint(hi)) atomicAdd(&cnt[2], 1u);
        a.x += 0.0078125f; a.y -= 0.00390625f;
        if ((it & 255) == 255) { a.x = 0.5f + (float)((t + it) % 61) * 0.015625f; a.y = 1.5f - (float)((t + it) % 53) * 0.015625f; }
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
    setvbuf(stdout, nullptr, _IOLBF, 0);
    unsigned* bad; char* src; float* sink;
    CK(hipMalloc(&bad, 32)); CK(hipMalloc(&src, 64 * 4096 + 4096)); CK(hipMalloc(&sink, 4)); CK(hipMemset(src, 0x3c, 64 * 4096 + 4096));
    CK(hipFuncSetAttribute((const void*)corunner, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipStream_t st[3];
    for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    const char* names[4] = {"victims alone", "beside f16 MFMA waves", "beside f16 + fp8-scaled MFMA waves", "beside f16 + fp8 MFMA waves with LDS-DMA"};
    for (int lds = 0; lds < 2; ++lds)
        for (int mode = -1; mode < 3; ++mode) {
            CK(hipMemset(bad, 0, 4));
            for (int rep = 0; rep < 40; ++rep) {
                if (mode >= 0) hipLaunchKernelGGL(corunner, dim3(512), dim3(256), 100 * 1024, st[0], src, sink, 6000, mode);
                for (int v = 1; v < 3; ++v)
                    for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(victim, dim3(1024), dim3(256), 0, st[v], bad, 20000, lds);
            }
            CK(hipDeviceSynchronize());
            unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            printf("victim %s, %-42s: %u mismatching (packed vs scalar) checks of %.3g\n", lds ? "with LDS reads" : "registers only ", names[mode + 1], h,
                   40.0 * 8 * 1024 * 256 * 6 * (20000 / 64));
        }
    // the write-after-read pattern of the failing build
    for (int sel = 0; sel < 2; ++sel)
    for (int mode = -1; mode < 3; ++mode) {
        CK(hipMemset(bad, 0, 32));
        for (int rep = 0; rep < 40; ++rep) {
            if (mode >= 0) hipLaunchKernelGGL(corunner, dim3(512), dim3(256), 100 * 1024, st[0], src, sink, 6000, mode);
            for (int v = 1; v < 3; ++v)
                for (int q = 0; q < 4; ++q) {
                    if (sel) hipLaunchKernelGGL(victim_war<1>, dim3(1024), dim3(256), 0, st[v], bad, 20000);
                    else hipLaunchKernelGGL(victim_war<0>, dim3(1024), dim3(256), 0, st[v], bad, 20000);
                }
        }
        CK(hipDeviceSynchronize());
        unsigned h[8]; CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
        printf("victim v_mov / v_pk_fma_f32 %s / v_mov (source rewritten behind the packed FMA), %-42s: wrong sums by quarter wave [lanes 0-15, 16-31, 32-47, 48-63] = %u %u %u %u of %.3g checks\n",
               sel ? "op_sel:[0,1,0]" : "op_sel_hi:[1,0,1]", names[mode + 1], h[1], h[2], h[3], h[4], 40.0 * 8 * 1024 * 256 * 20000);
    }
    // which part of the pattern is needed (alone, and beside the strongest co-runner)
    const char* forms[6] = {"v_pk_fma_f32 op_sel:[0,1,0], fixed sources (no v_mov)", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_mul_f32 op_sel:[0,1] + v_pk_add_f32",
                            "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mov_b32 op_sel:[0,1]"};
    for (int form = 0; form < 6; ++form)
        for (int mode = -1; mode < 3; mode += 3) {
            CK(hipMemset(bad, 0, 32));
            for (int rep = 0; rep < 40; ++rep) {
                if (mode >= 0) hipLaunchKernelGGL(corunner, dim3(512), dim3(256), 100 * 1024, st[0], src, sink, 6000, mode);
                for (int v = 1; v < 3; ++v)
                    for (int q = 0; q < 4; ++q) {
                        if (form == 0) hipLaunchKernelGGL(victim_form<0>, dim3(1024), dim3(256), 0, st[v], bad, 20000);
                        if (form == 1) hipLaunchKernelGGL(victim_form<1>, dim3(1024), dim3(256), 0, st[v], bad, 20000);
                        if (form == 2) hipLaunchKernelGGL(victim_form<2>, dim3(1024), dim3(256), 0, st[v], bad, 20000);
                        if (form == 3) hipLaunchKernelGGL(victim_form<3>, dim3(1024), dim3(256), 0, st[v], bad, 20000);
                        if (form == 4) hipLaunchKernelGGL(victim_form<4>, dim3(1024), dim3(256), 0, st[v], bad, 20000);
                        if (form == 5) hipLaunchKernelGGL(victim_form<5>, dim3(1024), dim3(256), 0, st[v], bad, 20000);
                    }
            }
            CK(hipDeviceSynchronize());
            unsigned h[8]; CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
            printf("victim %-54s, %-42s: wrong sums by quarter wave = %u %u %u %u of %.3g checks\n", forms[form], names[mode + 1], h[1], h[2], h[3], h[4],
                   40.0 * 8 * 1024 * 256 * 20000);
        }
    {
        float* samples; CK(hipMalloc(&samples, 24 * 6 * 4)); CK(hipMemset(samples, 0, 24 * 6 * 4));
        CK(hipMemset(bad, 0, 32));
        for (int rep = 0; rep < 40; ++rep) {
            hipLaunchKernelGGL(corunner, dim3(512), dim3(256), 100 * 1024, st[0], src, sink, 6000, 2);
            for (int v = 1; v < 3; ++v)
                for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(victim_value, dim3(1024), dim3(256), 0, st[v], bad, 20000, samples);
        }
        CK(hipDeviceSynchronize());
        unsigned h[8]; CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
        printf("one v_pk_mul_f32 op_sel:[0,1] per check beside MFMA waves: low half wrong %u times, %u of them = a.x * w.x (LOW word of source 1 read instead of the HIGH word), "
               "%u outside lanes 48-63; high half wrong %u times\n", h[0], h[1], h[3], h[2]);
        float hs[24 * 6]; CK(hipMemcpy(hs, samples, sizeof(hs), hipMemcpyDeviceToHost));
        for (int k = 0; k < 24; ++k)
            printf("  sample: lane %2.0f  a.x %.7g  w = (%.7g, %.7g)  low half %.9g (a.x*w.y = %.9g; low half / a.x = %.7g)  high half %.9g\n", hs[6 * k + 4], hs[6 * k + 1], hs[6 * k + 2],
                   hs[6 * k + 3], hs[6 * k], hs[6 * k + 1] * hs[6 * k + 3], hs[6 * k] / hs[6 * k + 1], hs[6 * k + 5]);
    }
    return 0;
}
