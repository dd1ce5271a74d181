// Stand-alone probe (round 3): semantics and rates of the gfx950 bf6 (e3m2) path the F16F6 convolution relies on.
//   T1  v_cvt_scalef32_pk32_bf6_f16: bit layout of the 32 codes, rounding, saturation, direction of the scale
//   T2  v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 3: lane -> (row / column, k) map of the 6-register operands
//   T3  block scales: which byte op_sel picks, E8M0 meaning
//   T4  cycles per MFMA: f16 32x32x16, fp8 32x32x64, bf6 32x32x64 (one wave per SIMD, s_memtime)
//   T5  cycles per v_cvt_scalef32_pk32_bf6_f16
// build: hipcc --offload-arch=gfx950 -O3 -o probe_bf6 probe_bf6.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float e3m2_decode(int c) {
    const int s = (c >> 5) & 1, E = (c >> 2) & 7, M = c & 3;
    const float v = E == 0 ? ldexpf((float)M / 4.f, -2) : ldexpf(1.f + (float)M / 4.f, E - 3);
    return s ? -v : v;
}
static int e3m2_encode(float x) {   // round to nearest even, saturating
    const int s = std::signbit(x) ? 1 : 0;
    float a = fabsf(x);
    if (!(a == a)) return s << 5 | 31;
    if (a >= 28.f) return s << 5 | 31;
    int e;
    frexpf(a, &e);                   // a = f * 2^e, f in [0.5, 1)
    int E = e - 1 + 3;               // biased exponent of 1.m form
    if (E < 1) E = 0;
    const float step = E == 0 ? ldexpf(1.f, -4) : ldexpf(1.f, E - 3 - 2);
    float q = nearbyintf(a / step);  // RNE in the default rounding mode
    float v = q * step;
    if (v >= 28.f) v = 28.f;
    // re-derive the code from v
    if (v == 0.f) return s << 5;
    frexpf(v, &e);
    E = e - 1 + 3;
    int M;
    if (E < 1) { E = 0; M = (int)(v / ldexpf(1.f, -4)); }
    else M = (int)((v / ldexpf(1.f, E - 3) - 1.f) * 4.f);
    return s << 5 | E << 2 | M;
}
static int get6(const int* w, int i) {
    const int bit = 6 * i;
    unsigned long long lo = (unsigned)w[bit >> 5];
    if ((bit >> 5) + 1 < 6) lo |= (unsigned long long)(unsigned)w[(bit >> 5) + 1] << 32;
    return (int)((lo >> (bit & 31)) & 63);
}
static void put6(int* w, int i, int c) {
    const int bit = 6 * i;
    unsigned long long v = (unsigned long long)(c & 63) << (bit & 31);
    w[bit >> 5] |= (int)(unsigned)v;
    if ((bit >> 5) + 1 < 8) w[(bit >> 5) + 1] |= (int)(unsigned)(v >> 32);
}

__global__ void k_cvt(const h32* x, v6i* y, const float* s) {
    y[threadIdx.x] = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(x[threadIdx.x], s[threadIdx.x]);
}

__global__ void k_mfma(const v8i* a, const v8i* b, f32x16* c, const int* sa, const int* sb, int mode) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const v8i av = a[threadIdx.x], bv = b[threadIdx.x];
    const int sav = sa[threadIdx.x], sbv = sb[threadIdx.x];
    if (mode == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 3, 3, 0, sav, 0, sbv);
    else if (mode == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 3, 3, 1, sav, 0, sbv);
    else if (mode == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 3, 3, 0, sav, 1, sbv);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 3, 3, 2, sav, 3, sbv);
    c[threadIdx.x] = acc;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(unsigned long long* out, int iters, int seed) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x3c003c00 ^ (threadIdx.x * 2654435761u + seed + i * 97); b[i] = a[i] * 31 + 7; }
    half8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)((float)((threadIdx.x + i) % 17) * 0.125f - 1.f); bh[i] = (_Float16)((float)((threadIdx.x * 3 + i) % 13) * 0.25f - 1.5f); }
    asm volatile("" : "+v"(a), "+v"(b), "+v"(ah), "+v"(bh));
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            if (KIND == 1) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (KIND == 2) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (KIND == 3) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 3, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (KIND == 4) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_cvt_rate(unsigned long long* out, int iters) {
    h32 x;
    for (int i = 0; i < 32; ++i) x[i] = (_Float16)((float)((threadIdx.x + i) % 29) * 0.5f - 7.f);
    asm volatile("" : "+v"(x));
    v6i acc = {0, 0, 0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
        v6i y = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(x, 1.0f);
        asm volatile("" : "+v"(y));
        acc ^= y;
        x[0] = (_Float16)((float)x[0] + 0.f);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (acc[0] == 0x12345678) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}


// T6: 4 f16 MFMAs + NC conversions per iteration (do the conversions hide under the matrix pipe?)
template <int NC>
__global__ __launch_bounds__(256) void k_mix_rate(unsigned long long* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    half8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)((float)((threadIdx.x + i) % 17) * 0.125f - 1.f); bh[i] = (_Float16)((float)((threadIdx.x * 3 + i) % 13) * 0.25f - 1.5f); }
    h32 x[2];
    for (int c = 0; c < 2; ++c) for (int i = 0; i < 32; ++i) x[c][i] = (_Float16)((float)((threadIdx.x + i + c) % 29) * 0.5f - 7.f);
    asm volatile("" : "+v"(ah), "+v"(bh), "+v"(x[0]), "+v"(x[1]));
    v6i keep = {0, 0, 0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            if (t < NC) {
                v6i y = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(x[t & 1], 1.0f);
                asm volatile("" : "+v"(y));
                keep ^= y;
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.678f || keep[0] == 0x1234567) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// T7: v_pk_maximum3_f16 / v_pk_minimum3_f16 by inline asm: per-lane max |x| over 32 halves in 19 instructions
__global__ void k_max3(const h32* x, int* out) {
    typedef int v16i __attribute__((ext_vector_type(16)));
    h32 v = x[threadIdx.x];
    v16i r = __builtin_bit_cast(v16i, v);
    int mx[8], mn[8];
#define MAX3(d, a, b, c) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define MIN3(d, a, b, c) asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
    MAX3(mx[0], r[0], r[1], r[2]); MAX3(mx[1], r[3], r[4], r[5]); MAX3(mx[2], r[6], r[7], r[8]); MAX3(mx[3], r[9], r[10], r[11]);
    MAX3(mx[4], r[12], r[13], r[14]); MAX3(mx[5], mx[0], mx[1], r[15]); MAX3(mx[6], mx[2], mx[3], mx[4]); MAX3(mx[7], mx[5], mx[6], mx[6]);
    MIN3(mn[0], r[0], r[1], r[2]); MIN3(mn[1], r[3], r[4], r[5]); MIN3(mn[2], r[6], r[7], r[8]); MIN3(mn[3], r[9], r[10], r[11]);
    MIN3(mn[4], r[12], r[13], r[14]); MIN3(mn[5], mn[0], mn[1], r[15]); MIN3(mn[6], mn[2], mn[3], mn[4]); MIN3(mn[7], mn[5], mn[6], mn[6]);
    int m2, m1;
    asm volatile("v_pk_max_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(m2) : "v"(mx[7]), "v"(mn[7]));
    asm volatile("v_pk_max_f16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m1) : "v"(m2));
    out[threadIdx.x] = m1 & 0xffff;
}

int main() {
    // ---------------- T1 ----------------
    {
        std::vector<float> vals = {0.f, 0.0625f, 0.03125f, 0.09375f, 0.125f, 0.25f, 0.3f, 0.4375f, 0.5f, 0.75f, 1.f, 1.125f, 1.25f, 1.375f,
                                   1.5f, 1.75f, 2.f, 3.f, 4.f, 5.f, 7.f, 9.f, 12.f, 14.f, 16.f, 20.f, 24.f, 26.f, 28.f, 30.f, 100.f, -3.5f};
        h32* dx; v6i* dy; float* ds;
        CK(hipMalloc(&dx, 64 * sizeof(h32))); CK(hipMalloc(&dy, 64 * sizeof(v6i))); CK(hipMalloc(&ds, 64 * sizeof(float)));
        std::vector<_Float16> hx(64 * 32);
        std::vector<float> hs(64);
        const float scales[4] = {1.f, 2.f, 0.25f, 1.5f};
        for (int l = 0; l < 64; ++l) {
            hs[l] = scales[l & 3];
            for (int i = 0; i < 32; ++i) hx[l * 32 + i] = (_Float16)(vals[(i + l / 4) % 32] * ((l / 4) & 1 ? -1.f : 1.f));
        }
        CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(ds, hs.data(), 64 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dx, dy, ds);
        CK(hipDeviceSynchronize());
        const int VS = (int)(sizeof(v6i) / 4);
        std::vector<int> hy(64 * VS);
        CK(hipMemcpy(hy.data(), dy, 64 * sizeof(v6i), hipMemcpyDeviceToHost));
        int bad_div = 0, bad_mul = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 32; ++i) {
                const float v = (float)hx[l * 32 + i];
                const float sc = ldexpf(1.f, (int)floorf(log2f(hs[l])));   // only the exponent of the scale is expected to count
                const int got = get6(&hy[l * VS], i);
                if (got != e3m2_encode(v / sc) && !(v == 0.f && (got & 31) == 0)) ++bad_div;
                if (got != e3m2_encode(v * sc) && !(v == 0.f && (got & 31) == 0)) ++bad_mul;
            }
        printf("T1 cvt_scalef32_pk32_bf6_f16: mismatches if dst = src / 2^floor(log2 scale): %d ; if dst = src * ...: %d (of 2048)\n", bad_div, bad_mul);
        for (int l = 0; l < 4; ++l) {
            printf("   lane %d scale %.2f:", l, hs[l]);
            for (int i = 0; i < 32; ++i) printf(" %g->%g", (float)hx[l * 32 + i], e3m2_decode(get6(&hy[l * VS], i)));
            printf("\n");
        }
    }
    // ---------------- T2 / T3 ----------------
    {
        // A[m][k], B[k][n] small exactly representable values; assumed map: lane l holds row/col l & 31, k = 32 (l >> 5) + i
        std::vector<float> A(32 * 64), B(64 * 32);
        const float tab[8] = {0.f, 0.25f, 0.5f, 1.f, -1.f, 1.5f, 2.f, -0.75f};
        for (int m = 0; m < 32; ++m) for (int k = 0; k < 64; ++k) A[m * 64 + k] = tab[(m * 5 + k * 3 + 1) & 7];
        for (int k = 0; k < 64; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = tab[(k * 7 + n * 11 + 2) & 7];
        std::vector<int> ha(64 * 8, 0), hb(64 * 8, 0), hsa(64), hsb(64);
        for (int l = 0; l < 64; ++l) {
            for (int i = 0; i < 32; ++i) {
                put6(&ha[l * 8], i, e3m2_encode(A[(l & 31) * 64 + 32 * (l >> 5) + i]));
                put6(&hb[l * 8], i, e3m2_encode(B[(32 * (l >> 5) + i) * 32 + (l & 31)]));
            }
            // byte 0: 2^0, byte 1: 2^1 (row-dependent: +1 for odd rows), byte 2: 2^-2, byte 3: 2^3
            hsa[l] = 127 | (128 + (l & 1)) << 8 | 125 << 16 | 130 << 24;
            hsb[l] = 127 | (129 - (l & 1)) << 8 | 126 << 16 | 124 << 24;
        }
        v8i *da, *db; f32x16* dc; int *dsa, *dsb;
        CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dc, 64 * 64)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
        CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb, mode);
            CK(hipDeviceSynchronize());
            std::vector<float> hc(64 * 16);
            CK(hipMemcpy(hc.data(), dc, 64 * 64, hipMemcpyDeviceToHost));
            const int ba[4] = {0, 1, 0, 2}, bb[4] = {0, 0, 1, 3};
            double maxerr = 0, maxref = 0;
            for (int l = 0; l < 64; ++l)
                for (int rg = 0; rg < 16; ++rg) {
                    const int n = l & 31, m = (rg & 3) + 8 * (rg >> 2) + 4 * (l >> 5);
                    double ref = 0;
                    for (int half = 0; half < 2; ++half) {
                        // scale of A: lane (m, half); scale of B: lane (n, half)
                        const int la = m + 32 * half, lb = n + 32 * half;
                        const double s = ldexp(1.0, ((hsa[la] >> (8 * ba[mode])) & 255) - 127) * ldexp(1.0, ((hsb[lb] >> (8 * bb[mode])) & 255) - 127);
                        double p = 0;
                        for (int i = 0; i < 32; ++i) p += (double)A[m * 64 + 32 * half + i] * B[(32 * half + i) * 32 + n];
                        ref += p * s;
                    }
                    maxerr = fmax(maxerr, fabs(ref - hc[l * 16 + rg]));
                    maxref = fmax(maxref, fabs(ref));
                }
            printf("T2/T3 mfma bf6 x bf6, op_sel a=%d b=%d: max |err| %.3g (max |ref| %.3g)\n", ba[mode], bb[mode], maxerr, maxref);
        }
    }
    // ---------------- T4 / T5 ----------------
    {
        unsigned long long* d; CK(hipMalloc(&d, 16)); CK(hipMemset(d, 0, 16));
        const int iters = 20000;
        const char* names[5] = {"f16 32x32x16", "fp8xfp8 32x32x64", "bf6xbf6 32x32x64", "bf6xfp8 32x32x64", "fp4xfp4 32x32x64"};
        for (int rep = 0; rep < 2; ++rep)
            for (int kind = 0; kind < 5; ++kind) {
                unsigned long long h[2];
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, d, iters, 1); break;
                    case 1: hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, d, iters, 1); break;
                    case 2: hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(256), 0, 0, d, iters, 1); break;
                    case 3: hipLaunchKernelGGL(k_rate<3>, dim3(256), dim3(256), 0, 0, d, iters, 1); break;
                    default: hipLaunchKernelGGL(k_rate<4>, dim3(256), dim3(256), 0, 0, d, iters, 1); break;
                }
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
                printf("T4 %-18s: %.1f cycles per MFMA (s_memtime ticks / MFMA, one wave per SIMD), %.3f ms wall\n", names[kind], (double)h[0] / (iters * 4.0), ms);
            }
        for (int rep = 0; rep < 2; ++rep) {
            unsigned long long h[2];
            hipLaunchKernelGGL(k_cvt_rate, dim3(256), dim3(256), 0, 0, d, 20000);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            printf("T5 cvt_scalef32_pk32_bf6_f16 (+1 xor chain of 6, 1 cvt f16): %.1f ticks per iteration\n", (double)h[0] / 20000.0);
        }
    }
    // ---------------- T6 ----------------
    {
        unsigned long long* d; CK(hipMalloc(&d, 16)); CK(hipMemset(d, 0, 16));
        const int iters = 20000;
        for (int nc = 0; nc <= 4; ++nc) {
            unsigned long long h[2];
            switch (nc) {
                case 0: hipLaunchKernelGGL(k_mix_rate<0>, dim3(256), dim3(256), 0, 0, d, iters); break;
                case 1: hipLaunchKernelGGL(k_mix_rate<1>, dim3(256), dim3(256), 0, 0, d, iters); break;
                case 2: hipLaunchKernelGGL(k_mix_rate<2>, dim3(256), dim3(256), 0, 0, d, iters); break;
                case 3: hipLaunchKernelGGL(k_mix_rate<3>, dim3(256), dim3(256), 0, 0, d, iters); break;
                default: hipLaunchKernelGGL(k_mix_rate<4>, dim3(256), dim3(256), 0, 0, d, iters); break;
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            printf("T6 4 f16 MFMAs + %d cvt_pk32_bf6_f16 (+6 xor each) per iteration: %.1f ticks per iteration (128 = MFMA only)\n", nc, (double)h[0] / iters);
        }
    }
    // ---------------- T7 ----------------
    {
        h32* dx; int* dout;
        CK(hipMalloc(&dx, 64 * sizeof(h32))); CK(hipMalloc(&dout, 256));
        std::vector<_Float16> hx(64 * 32);
        std::vector<float> want(64);
        for (int l = 0; l < 64; ++l) {
            float m = 0.f;
            for (int i = 0; i < 32; ++i) {
                float v = ldexpf((float)(((l * 37 + i * 11) % 23) - 11) / 8.f, (l % 9) - 4);
                if (i == (l % 32)) v = (l & 1) ? -3.f * ldexpf(1.f, (l % 9) - 4) : 2.5f * ldexpf(1.f, (l % 9) - 4);
                hx[l * 32 + i] = (_Float16)v;
                m = fmaxf(m, fabsf((float)hx[l * 32 + i]));
            }
            want[l] = m;
        }
        CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_max3, dim3(1), dim3(64), 0, 0, dx, dout);
        CK(hipDeviceSynchronize());
        int ho[64]; CK(hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            _Float16 g; unsigned short u = (unsigned short)ho[l]; memcpy(&g, &u, 2);
            if ((float)g != want[l]) { if (bad < 4) printf("   lane %d: got %g want %g\n", l, (float)g, want[l]); ++bad; }
        }
        printf("T7 max|x| over 32 halves by pk_maximum3/minimum3: %d of 64 lanes wrong\n", bad);
    }
    return 0;
}
