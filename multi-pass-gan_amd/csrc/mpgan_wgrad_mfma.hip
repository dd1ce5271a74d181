// Weight gradient of the stride-1 SAME convolution on the matrix cores (gfx950).
//
//   dW[ky][kx][ci][co] = wscale * sum_{b,oy,ox} x[b, oy+ky-pt, ox+kx-pl, ci] * dy[b, oy, ox, co]
//
// is, per filter tap, a GEMM whose contraction runs over PIXELS.  v_mfma_f32_32x32x16_f16 wants the
// 8 contraction elements of a lane contiguous, so both operands are first rewritten channel-major
// ("P16": [N][H][C][2 planes hi,lo][Wp] fp16, Wp = W rounded up to 8, scaled by a power of two taken
// from the tensor's absolute maximum so that small gradients stay in the fp16 normal range).  Then
//   * a block (4 waves, one per SIMD, so each wave may hold up to 512 registers) owns one filter row ky,
//     up to 128 input x 128 output channels, and a range of output rows; it walks the rows in chunks
//     of 32*KS pixels, staging the x piece (with an 8-pixel halo on both sides) and the dy piece in
//     LDS, double buffered through registers;
//   * a wave owns a 32-channel ci tile, COTW co tiles and all KW taps of the row: KW*COTW accumulator
//     tiles (20 for a 5x5 128->128 layer).  Per 16-pixel k-step it reads one 24-pixel window of x per plane (3 x ds_read_b128) and
//     derives the KW shifted A fragments in registers (v_alignbit for odd shifts), so x is read from
//     LDS once for all taps; dy fragments are single aligned ds_read_b128;
//   * precision 3 adds the hi*lo and lo*hi products (fp32-grade, like MPG_PREC_F16X3); precision 1
//     keeps hi*hi only;
//   * partial sums of the pixel ranges are combined with fp32 atomics into dW.
#include "mpgan_internal.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BLK = 256;

using mpg::pow2_scale;

__global__ void absmax_kernel(const float* __restrict__ x, size_t n, unsigned int* __restrict__ out) {
    __shared__ float red[BLK];
    float m = 0.f;
    const size_t n4 = ((((uintptr_t)x) & 15) == 0) ? n / 4 : 0;          // 16-byte loads over the aligned bulk
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLK) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK)
        m = fmaxf(m, fabsf(x[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = BLK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(red[0]));   // non-negative floats order like their bit patterns
}

__global__ void amax_init_kernel(float* __restrict__ amax, const float* __restrict__ x_amax, const float* __restrict__ dy_amax) {
    const int i = threadIdx.x;
    if (i < 64) amax[i] = i == 0 ? (x_amax ? *x_amax : 0.f) : (i == 1 ? (dy_amax ? *dy_amax : 0.f) : 0.f);
}

// fp32 NHWC -> P16.  block: one image row, 64 pixels, 64 channels through an LDS transpose.
__global__ __launch_bounds__(256) void to_p16_kernel(const float* __restrict__ x, int h, int w, int c, int wp,
                                                     const float* __restrict__ amax, _Float16* __restrict__ out) {
    __shared__ float tile[64][65];
    const int row = blockIdx.x;            // b*h + y
    const int x0 = blockIdx.y * 64;
    const int c0 = blockIdx.z * 64;
    const float scale = pow2_scale(*amax);
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int px = e / 64, ch = e % 64;
        float v = 0.f;
        if (x0 + px < w && c0 + ch < c) v = x[((size_t)row * w + x0 + px) * c + c0 + ch];
        tile[px][ch] = v * scale;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 8; e += 256) {
        const int u = e % 8, ch = e / 8;
        if (c0 + ch >= c || x0 + 8 * u >= wp) continue;
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = tile[8 * u + i][ch];
            const _Float16 hh = (_Float16)v;
            hi[i] = hh;
            lo[i] = (_Float16)(v - (float)hh);
        }
        _Float16* o = out + (((size_t)row * c + c0 + ch) * 2) * wp + x0 + 8 * u;
        *reinterpret_cast<half8*>(o) = hi;
        *reinterpret_cast<half8*>(o + wp) = lo;
    }
}

#ifndef MPG_WG_DIAG
#define MPG_WG_DIAG 0
#endif
#if MPG_WG_DIAG
__device__ unsigned long long g_wg_diag[8];
#define WG_T() __builtin_readcyclecounter()
#endif

struct WgArgs {
    const _Float16* xp;      // P16 of x  [N][H][cin_total][2][wp]
    const _Float16* dp;      // P16 of dy [N][H][cout_total][2][wp]
    const float* amax;       // [0] x, [1] dy
    const char* zeros;       // >= 16 zero bytes: the source of padding / out-of-row units of the LDS-DMA copy
    float* dw;               // [kh][kw][cin_total][cout_total]
    int n, h, w, wp;
    int cin_total, cout_total, ci0, co0, cin, cout;   // channel window of this launch (cin, cout <= 128)
    int windows;                                       // equal cout windows handled by this launch (>= 1)
    int kh, pt, pl;
    int cit, cog, ks;        // waves = cit * cog * ks = 4
    int chunk;               // pixels per chunk = 32 * ks
    int xrowb, drowb;        // LDS row strides in bytes
    int nsplit, rows_per_split;
    float wscale;
};

// prefetch registers for the x piece: 10 units (a 64-pixel chunk of 128 channels, hi + lo) where the
// accumulators leave room, else 6 (32-pixel chunks)
constexpr int x_units(int kw, int cotw) { return kw * cotw > 12 ? 6 : 10; }
// with the pad unit that ends every LDS row (bank spread), per thread
constexpr int x_units_dma(int kw, int cotw) { return x_units(kw, cotw) + 2; }
constexpr int D_UNITS_DMA = 5;

template <int KW, int COTW, int PREC>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // equal output-channel windows of one layer run as one launch: linear block id = (jj * windows + window) * 8 + xcd, so
    // the windows of one row range are dispatched together and to the same XCD, and share the x rows through its L2
    // instead of streaming them from memory once per window
    const int nwin = a.windows;
    const int co_base = a.co0 + ((blockIdx.x / 8) % nwin) * a.cout;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // XCD-aware order: the kh blocks of one row range land on the same XCD (shared L2)
    const int xcd = blockIdx.x % 8, jj = blockIdx.x / (8 * nwin);
    const int ky = jj % a.kh;
    const int split = (jj / a.kh) * 8 + xcd;
    if (split >= a.nsplit) return;
    const int cit = wave % a.cit;
    const int cog = (wave / a.cit) % a.cog;
    const int ks = wave / (a.cit * a.cog);

    const int nplane = PREC == 3 ? 2 : 1;
    const int xch = a.cit * 32, dch = a.cog * COTW * 32;
    const int xbytes = xch * nplane * a.xrowb, dbytes = dch * nplane * a.drowb;
    char* xs[2] = {lds, lds + xbytes + dbytes};
    char* dsm[2] = {lds + xbytes, lds + 2 * xbytes + dbytes};

    // global -> LDS copy: LDS-DMA, 16 bytes per lane, every wave instruction fills 1 KiB of the image linearly.  The image
    // is [plane][channel][units of 8 pixels + one pad unit] (the pad spreads the rows over the banks; it is never
    // written or read), so unit u = (plane * channels + channel) * units_per_row + offset lands at byte 16 u.
    const int xupr = (a.chunk + 16) / 8 + 1, dupr = a.chunk / 8 + 1;  // units per row incl. the pad unit
    const int xunits = xch * nplane * xupr, dunits = dch * nplane * dupr;
    constexpr int XU = x_units_dma(KW, COTW), DU = D_UNITS_DMA;        // DMA instructions per wave and chunk
    const int total_rows = a.n * a.h;
    const int row_begin = split * a.rows_per_split;
    const int row_end = min(total_rows, row_begin + a.rows_per_split);
    const int chunks_per_row = (a.w + a.chunk - 1) / a.chunk;
    const int nwork = (row_end - row_begin) * chunks_per_row;

    f32x16 acc[KW][COTW];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int t = 0; t < COTW; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[k][t][v] = 0.f;

    // copy plan of this thread, formed once: per unit its offset inside a P16 row block (halves) and its pixel offset
    // inside the chunk.  (A first version staged the chunk through registers: 10 global loads + 10 ds_write_b128 per thread
    // and chunk took 2100 + 1700 cycles next to 3840 cycles of MFMA work, with nothing to hide them behind at one wave
    // per SIMD -- measured with MPG_WG_DIAG; the wide LDS stores alone run at a third of the read rate.)
    constexpr int NOPX = 1 << 28;                 // pixel offset of a unit that delivers zeros whatever the chunk
    constexpr int SKIP = -(1 << 28);              // ... of a unit that is not copied at all (pad unit, beyond the image)
    int xg[XU], xpo[XU], dg[DU], dpo[DU];
#pragma unroll
    for (int i = 0; i < XU; ++i) {
        const int u = tid + i * 256;
        const int rr = u / xupr, off = u % xupr;
        const int pln = rr / xch, ch = rr % xch;
        xg[i] = ((a.ci0 + ch) * 2 + pln) * a.wp + off * 8 - 8;
        xpo[i] = (u >= xunits || off == xupr - 1) ? SKIP : (ch < a.cin ? off * 8 - 8 : NOPX);
    }
#pragma unroll
    for (int i = 0; i < DU; ++i) {
        const int u = tid + i * 256;
        const int rr = u / dupr, off = u % dupr;
        const int pln = rr / dch, ch = rr % dch;
        dg[i] = ((co_base + ch) * 2 + pln) * a.wp + off * 8;
        dpo[i] = (u >= dunits || off == dupr - 1) ? SKIP : (ch < a.cout ? off * 8 : NOPX);
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // (the copy of a chunk is written in NPART parts so that variants can place them; the CU's address unit takes ~40 cycles
    // per LDS-DMA instruction of this shape: 3000 cycles for the 68 instructions of a chunk, measured)
    constexpr int NPART = 4;
    struct ChunkAt { const _Float16* xrow; const _Float16* drow; int x0; bool row_ok; };
    auto chunk_at = [&](int wi) {
        const int row = row_begin + wi / chunks_per_row;        // output row b*h + oy
        const int x0 = (wi % chunks_per_row) * a.chunk;
        const int oy = row % a.h;
        const int iy = oy + ky - a.pt;
        ChunkAt c;
        c.x0 = x0;
        c.row_ok = iy >= 0 && iy < a.h;
        c.xrow = a.xp + (size_t)(row - oy + iy) * a.cin_total * 2 * a.wp + x0;
        c.drow = a.dp + (size_t)row * a.cout_total * 2 * a.wp + x0;
        return c;
    };
    auto dma_part = [&](const ChunkAt& c, int buf, auto part_c) {
        constexpr int P = decltype(part_c)::value;
        const int x0 = c.x0;
        const bool row_ok = c.row_ok;
        const _Float16* xrow = c.xrow;
        const _Float16* drow = c.drow;
#pragma unroll
        for (int i = P; i < XU; i += NPART) {
            if (xpo[i] != SKIP) {
                const int px = x0 + xpo[i];
                const char* src = (row_ok && px >= 0 && px < a.wp) ? reinterpret_cast<const char*>(xrow + xg[i]) : a.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs[buf] + (i * 4 + wave_u) * 1024),
                                                 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = P; i < DU; i += NPART) {
            if (dpo[i] != SKIP) {
                const char* src = (row_ok && x0 + dpo[i] < a.wp) ? reinterpret_cast<const char*>(drow + dg[i]) : a.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dsm[buf] + (i * 4 + wave_u) * 1024),
                                                 16, 0, 0);
            }
        }
    };

    auto dma_parts_from = [&](const ChunkAt& c, int buf, int first) {      // parts first .. NPART-1 (first is wave-uniform)
        if (first <= 0) dma_part(c, buf, std::integral_constant<int, 0>{});
        if (first <= 1) dma_part(c, buf, std::integral_constant<int, 1>{});
        if (first <= 2) dma_part(c, buf, std::integral_constant<int, 2>{});
        if (first <= 3) dma_part(c, buf, std::integral_constant<int, 3>{});
    };
    if (nwork > 0) dma_parts_from(chunk_at(0), 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int ksteps = a.chunk / 16;
#if MPG_WG_DIAG
    unsigned long long d_fetch = 0, d_comp = 0, d_stash = 0, d_bar = 0, d_wait = 0;
    const unsigned long long t_begin = WG_T();
#endif
    for (int wi = 0; wi < nwork; ++wi) {
        const int buf = wi & 1;
#if MPG_WG_DIAG
        const unsigned long long t0 = WG_T();
#endif
        // the whole next chunk at once: spreading the LDS-DMA instructions between the MFMA phases was measured SLOWER
        // (9900 against 9000 cycles per chunk): with one wave per SIMD every instruction stalls the wave while the CU's
        // address unit works, and the matrix pipe drains each time
        if (wi + 1 < nwork) dma_parts_from(chunk_at(wi + 1), buf ^ 1, 0);
#if MPG_WG_DIAG
        const unsigned long long t1 = WG_T();
#endif
        const char* xb = xs[buf] + (cit * 32 + r) * a.xrowb + hh * 16;
        const char* db = dsm[buf] + ((cog * COTW) * 32 + r) * a.drowb + hh * 16;
        for (int j = ks; j < ksteps && ks < a.ks; j += a.ks) {
            // 24-pixel window of x per plane: pixels [16j + 8h - 8, 16j + 8h + 16) relative to the chunk start
            unsigned int win[2][12];
#pragma unroll
            for (int pln = 0; pln < nplane; ++pln) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(xb + pln * xch * a.xrowb + j * 32 + q * 16);
                    win[pln][4 * q + 0] = v[0]; win[pln][4 * q + 1] = v[1];
                    win[pln][4 * q + 2] = v[2]; win[pln][4 * q + 3] = v[3];
                }
            }
            half8 bfr[2][COTW];
#pragma unroll
            for (int pln = 0; pln < nplane; ++pln)
#pragma unroll
                for (int t = 0; t < COTW; ++t)
                    bfr[pln][t] = *reinterpret_cast<const half8*>(db + pln * dch * a.drowb + t * 32 * a.drowb + j * 32);
            // the shifted A fragments of all taps first, then the products in product-major order: consecutive MFMAs go
            // to different accumulators (three back-to-back products into one accumulator wait for each other with one
            // wave per SIMD)
            half8 afr[KW][2];
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                // pixels [8 + s, 16 + s) of the window, s = kx - pl (|s| <= 3)
                const int s = kx - (KW - 1) / 2;
#pragma unroll
                for (int pln = 0; pln < nplane; ++pln) {
                    u32x4 f;
                    if ((s & 1) == 0) {
                        const int q = (8 + s) / 2;
                        f[0] = win[pln][q]; f[1] = win[pln][q + 1]; f[2] = win[pln][q + 2]; f[3] = win[pln][q + 3];
                    } else {
                        const int q = (7 + s) / 2;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            f[i] = __builtin_amdgcn_alignbit(win[pln][q + i + 1], win[pln][q + i], 16);
                    }
                    afr[kx][pln] = __builtin_bit_cast(half8, f);
                }
            }
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                for (int t = 0; t < COTW; ++t)
                    acc[kx][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[kx][0], bfr[0][t], acc[kx][t], 0, 0, 0);
            if (PREC == 3) {
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                    for (int t = 0; t < COTW; ++t)
                        acc[kx][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[kx][0], bfr[1][t], acc[kx][t], 0, 0, 0);
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                    for (int t = 0; t < COTW; ++t)
                        acc[kx][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[kx][1], bfr[0][t], acc[kx][t], 0, 0, 0);
            }
        }
#if MPG_WG_DIAG
        const unsigned long long t2a = WG_T();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = WG_T();
        d_wait += t2 - t2a;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next chunk has landed in the other buffer
#if MPG_WG_DIAG
        const unsigned long long t3 = WG_T();
#endif
        __syncthreads();
#if MPG_WG_DIAG
        const unsigned long long t4 = WG_T();
        d_fetch += t1 - t0; d_comp += t2a - t1; d_stash += t3 - t2; d_bar += t4 - t3;
#endif
    }
#if MPG_WG_DIAG
    const unsigned long long t_loop = WG_T();
#endif

    // epilogue: D row = (v&3) + 8*(v>>2) + 4*(lane>>5) is the input channel, column lane&31 the output channel
    const float unscale = a.wscale / (pow2_scale(a.amax[0]) * pow2_scale(a.amax[1]));
#pragma unroll
    for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int t = 0; t < COTW; ++t) {
            const int co = (cog * COTW + t) * 32 + r;
            if (co >= a.cout) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ci = cit * 32 + (v & 3) + 8 * (v >> 2) + 4 * hh;
                const float val = acc[kx][t][v];
                if (ci < a.cin && val != 0.f)
                    atomicAdd(a.dw + ((size_t)(ky * KW + kx) * a.cin_total + a.ci0 + ci) * a.cout_total + co_base + co,
                              val * unscale);
            }
        }
#if MPG_WG_DIAG
    if (lane == 0 && blockIdx.x % 64 == 0) {
        const unsigned long long t_end = WG_T();
        atomicAdd(&g_wg_diag[0], d_fetch); atomicAdd(&g_wg_diag[1], d_comp); atomicAdd(&g_wg_diag[2], d_stash);
        atomicAdd(&g_wg_diag[3], d_bar); atomicAdd(&g_wg_diag[4], t_loop - t_begin); atomicAdd(&g_wg_diag[5], d_wait);
        atomicAdd(&g_wg_diag[6], 1ull); atomicAdd(&g_wg_diag[7], (unsigned long long)nwork);
    }
#endif
}

template <int KW, int COTW, int PREC>
hipError_t launch(hipStream_t s, const WgArgs& a, int blocks, size_t lds_bytes, int windows) {
    auto kern = wgrad_mfma_kernel<KW, COTW, PREC>;
    static int lds_limit[64] = {0};
    hipError_t e = mpg::ensure_dyn_lds((const void*)kern, 160 * 1024, lds_limit);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks * windows), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

inline size_t p16_elems(int n, int h, int w, int c) { return (size_t)n * h * c * 2 * ((w + 7) & ~7); }

}  // namespace

#if MPG_WG_DIAG
extern "C" int mpg_debug_wg_diag(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wg_diag), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wg_diag), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int mpg_absmax(mpg_stream_t stream, const float* x, size_t n, float* out) {
    MPG_REQUIRE(x && out, "mpg_absmax: null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = mpg::zero_async(out, sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_absmax: zero");
    if (n == 0) return MPG_OK;
    size_t b = (n + BLK * 16 - 1) / (BLK * 16);
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)b), dim3(BLK), 0, s, x, n, (unsigned int*)out);
    MPG_LAUNCH_CHECK("absmax_kernel");
}

extern "C" size_t mpg_conv2d_wgrad_mfma_ws_bytes(int n, int h, int w, int cin, int cout) {
    if (n < 1 || h < 1 || w < 1 || cin < 1 || cout < 1) return 0;
    return 256 + (p16_elems(n, h, w, cin) + p16_elems(n, h, w, cout)) * sizeof(_Float16);
}

extern "C" int mpg_conv2d_wgrad_mfma(mpg_stream_t stream, const float* x, int n, int h, int w, int cin,
                                     const float* dy, int cout, int kh, int kw, float wscale, int prec,
                                     void* workspace, size_t workspace_bytes, const float* dy_amax, const float* x_amax,
                                     float* dw) {
    MPG_REQUIRE(x && dy && dw && workspace, "mpg_conv2d_wgrad_mfma: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && cin >= 1 && cout >= 1, "mpg_conv2d_wgrad_mfma: bad shape");
    MPG_REQUIRE(kh >= 1 && kh <= 7 && (kw == 1 || kw == 3 || kw == 4 || kw == 5),
                "mpg_conv2d_wgrad_mfma: filter %dx%d not built", kh, kw);
    MPG_REQUIRE(prec == MPG_PREC_F16X1 || prec == MPG_PREC_F16X3, "mpg_conv2d_wgrad_mfma: prec %d", prec);
    MPG_REQUIRE(workspace_bytes >= mpg_conv2d_wgrad_mfma_ws_bytes(n, h, w, cin, cout) &&
                    (((uintptr_t)workspace) & 255) == 0,
                "mpg_conv2d_wgrad_mfma: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    const char* zeros = mpg::zero_page();
    MPG_REQUIRE(zeros != nullptr, "mpg_conv2d_wgrad_mfma: could not allocate the zero page");
    const int wp = (w + 7) & ~7;
    float* amax = (float*)workspace;
    _Float16* xp = (_Float16*)((char*)workspace + 256);
    _Float16* dp = xp + p16_elems(n, h, w, cin);
    // amax[0] / amax[1]: the callers' values where given (one small kernel: device-to-device copies are slower graph
    // nodes than a launch), zero where the reductions below fill them in
    hipLaunchKernelGGL(amax_init_kernel, dim3(1), dim3(64), 0, s, amax, x_amax, dy_amax);
    hipError_t e = mpg::zero_async(dw, (size_t)kh * kw * cin * cout * sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_conv2d_wgrad_mfma: memset");
    const size_t nx = (size_t)n * h * w * cin, nd = (size_t)n * h * w * cout;
    auto am_grid = [](size_t n) { const size_t b = (n + BLK * 16 - 1) / (BLK * 16); return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); };
    if (x_amax == nullptr)       // else: e.g. a forward activation whose scale the caller fixes (no reduction pass over x)
        hipLaunchKernelGGL(absmax_kernel, dim3(am_grid(nx)), dim3(BLK), 0, s, x, nx, (unsigned int*)amax);
    if (dy_amax == nullptr)      // else: the caller already has max |dy| (it scales the data gradient with it too)
        hipLaunchKernelGGL(absmax_kernel, dim3(am_grid(nd)), dim3(BLK), 0, s, dy, nd, (unsigned int*)(amax + 1));
    hipLaunchKernelGGL(to_p16_kernel, dim3(n * h, (w + 63) / 64, (cin + 63) / 64), dim3(256), 0, s, x, h, w, cin, wp,
                       amax, xp);
    hipLaunchKernelGGL(to_p16_kernel, dim3(n * h, (w + 63) / 64, (cout + 63) / 64), dim3(256), 0, s, dy, h, w, cout, wp,
                       amax + 1, dp);

    // a wave keeps KW * COTW accumulator tiles of 16 registers in the 256 AccVGPRs: 16 tiles at most, so a
    // 5-wide filter row takes 64 output channels per launch (x is then staged twice, from L2)
    const int co_step = kw >= 5 ? 64 : 128;
    const bool merged = cout > co_step && cout % co_step == 0;     // equal windows: one launch, windows on grid.y
    for (int ci0 = 0; ci0 < cin; ci0 += 128)
        for (int co0 = 0; co0 < (merged ? 1 : cout); co0 += co_step) {
            WgArgs a;
            a.xp = xp; a.dp = dp; a.amax = amax; a.dw = dw; a.zeros = zeros;
            a.n = n; a.h = h; a.w = w; a.wp = wp;
            a.cin_total = cin; a.cout_total = cout; a.ci0 = ci0; a.co0 = co0;
            a.cin = cin - ci0 < 128 ? cin - ci0 : 128;
            a.cout = cout - co0 < co_step ? cout - co0 : co_step;
            a.kh = kh; a.pt = (kh - 1) / 2; a.pl = (kw - 1) / 2;
            a.wscale = wscale;
            const int nci = (a.cin + 31) / 32, nco = (a.cout + 31) / 32;
            a.cit = nci > 2 ? 4 : nci;                           // 1, 2, 4 ci tiles, one per wave
            const int avail = 4 / a.cit;                          // waves left for output-channel groups
            int cog = nco >= 3 ? 4 : nco;                         // 1, 2, 4
            if (cog > avail) cog = avail;
            a.cog = cog;
            int cotw = (nco + cog - 1) / cog;                     // co tiles per wave: 1, 2, 4
            if (cotw == 3) cotw = 4;
            const int npl = prec == MPG_PREC_F16X3 ? 2 : 1;
            a.ks = 4 / (a.cit * a.cog);                           // waves left over split the 16-pixel k-steps of a chunk
            a.chunk = 32 * a.ks < 64 ? 64 : 32 * a.ks;            // fewer barriers per pixel with 64-pixel chunks
            const int xu = x_units_dma(kw, cotw);
            // short rows, and at most xu (x) / D_UNITS_DMA (dy) LDS-DMA instructions per wave in the copy plan
            while (a.chunk > 32 && (a.chunk / 2 >= wp || a.cit * 32 * npl * ((a.chunk + 16) / 8 + 1) > xu * 256 ||
                                    a.cog * cotw * 32 * npl * (a.chunk / 8 + 1) > D_UNITS_DMA * 256))
                a.chunk /= 2;
            if (a.chunk / 16 < a.ks) a.ks = a.chunk / 16;         // the other waves idle (tiny layers)
            a.xrowb = (a.chunk + 16) * 2 + 16;
            a.drowb = a.chunk * 2 + 16;
            const size_t lds = 2 * ((size_t)a.cit * 32 * npl * a.xrowb + (size_t)a.cog * cotw * 32 * npl * a.drowb);
            MPG_REQUIRE(lds <= 160 * 1024, "mpg_conv2d_wgrad_mfma: LDS plan %zu bytes", lds);
            MPG_REQUIRE(a.cit * 32 * npl * ((a.chunk + 16) / 8 + 1) <= xu * 256 &&
                            a.cog * cotw * 32 * npl * (a.chunk / 8 + 1) <= D_UNITS_DMA * 256,
                        "mpg_conv2d_wgrad_mfma: copy plan");
            const int rows = n * h;
            int want = 1024 / kh;
            if (want < 1) want = 1;
            int nsplit = want < rows ? want : rows;
            a.rows_per_split = (rows + nsplit - 1) / nsplit;
            a.nsplit = (rows + a.rows_per_split - 1) / a.rows_per_split;
            const int blocks = ((a.nsplit + 7) / 8) * 8 * kh;
            const int windows = merged ? cout / co_step : 1;
            a.windows = windows;
            hipError_t le = hipErrorInvalidValue;
#define MPG_WGM(K, C)                                                                        \
    if (kw == K && cotw == C)                                                                \
        le = prec == MPG_PREC_F16X3 ? launch<K, C, 3>(s, a, blocks, lds, windows) : launch<K, C, 1>(s, a, blocks, lds, windows)
            MPG_WGM(1, 1); MPG_WGM(1, 2); MPG_WGM(1, 4); MPG_WGM(3, 1); MPG_WGM(3, 2); MPG_WGM(3, 4);
            MPG_WGM(4, 1); MPG_WGM(4, 2); MPG_WGM(4, 4); MPG_WGM(5, 1); MPG_WGM(5, 2);
#undef MPG_WGM
            if (le != hipSuccess) return mpg::hip_check(le, "wgrad_mfma_kernel");
        }
    MPG_LAUNCH_CHECK("mpg_conv2d_wgrad_mfma");
}
