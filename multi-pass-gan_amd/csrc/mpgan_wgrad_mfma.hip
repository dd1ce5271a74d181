// Weight gradient of the stride-1 SAME convolution on the matrix cores (gfx950).
//
//   dW[ky][kx][ci][co] = wscale * sum_{b,oy,ox} x[b, oy+ky-pt, ox+kx-pl, ci] * dy[b, oy, ox, co]
//
// is, per filter tap, a GEMM whose contraction runs over PIXELS.  v_mfma_f32_32x32x16_f16 wants the
// 8 contraction elements of a lane in one register quad, while the tensors of the convolution kernels (G8:
// [N][C/8][2 planes hi,lo][H][W][8 fp16], scaled by a power of two taken from the tensor's absolute maximum where it
// is a gradient) keep 8 CHANNELS of a pixel together.  The kernel reads G8 as it is: the pixel rows of a channel
// group are copied to LDS unchanged and the fragments come out of ds_read_b64_tr_b16, the transposing LDS read of
// gfx950 (a first version rewrote both operands channel-major in memory first, "P16": two more passes over x and dy
// and 2.3 of the 3.4 GB a call moved at 128x128 channels).  So the layer's forward input and the scaled dy of the
// data-gradient convolution are shared with this kernel, without a conversion of their own.
//   * a block (8 waves, two per SIMD, each with at most 6 accumulator tiles; rounds 1-2 ran 4 waves with up to 16) owns one
//     filter row ky,
//     up to 128 input x 128 output channels, and a range of output rows; it walks the rows in chunks
//     of 32*KS pixels, staging the x piece (with an 8-pixel halo on both sides) and the dy piece in
//     LDS by LDS-DMA, double buffered;
//   * a wave owns a 32-channel ci tile, COTW co tiles and all KW taps of the row: KW*COTW accumulator
//     tiles (at most 6: 256 registers per wave).  Per 16-pixel k-step it reads one 24-pixel window of x per plane (6 transposed
//     reads of 4 pixels) and derives the KW shifted A fragments in registers (v_alignbit for odd shifts), so x is read
//     from LDS once for all taps; a dy fragment is two transposed reads;
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

#ifndef MPG_WG_DIAG
#define MPG_WG_DIAG 0
#endif
#define WG_MFMA(a_, b_, c_, x_, y_, z_) ((MPG_WG_EXP & 2) ? (c_) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, x_, y_, z_))
#ifndef MPG_WG_EXP         // timing experiments (results are garbage): 1 no LDS waits, 2 no MFMAs, 4 no copy of the next chunk
#define MPG_WG_EXP 0
#endif
// MPG_WG_SPREAD 1 (development A/B, off): one LDS-DMA instruction of the next chunk's copy in front of each product of the
// k-steps instead of the whole copy at the head of the chunk.  Measured on the 5x5 128->128 layer, 16 tiles of 256^2
// (profiles/r03/wgrad_variants.md): 3.62 ms against 2.98 ms.  A wave waits for the CU's address unit at every LDS-DMA
// instruction (~35 cycles each when the unit is free, 64 instructions per chunk and CU), and spread out, the waits of the
// two waves of a SIMD fall into the MFMA phase of both; at the head of the chunk they cost 2100 cycles once, with idle
// matrix pipes, and the k-steps then run undisturbed (4500 cycles for 3840 of MFMA work).
#ifndef MPG_WG_SPREAD
#define MPG_WG_SPREAD 0
#endif
#ifndef MPG_WG_RING        // 0: square 3x3 / 4x4 / 5x5 filters on the one-filter-row kernel too (development A/B)
#define MPG_WG_RING 1
#endif
#if MPG_WG_DIAG
__device__ unsigned long long g_wg_diag[8];
#define WG_T() __builtin_readcyclecounter()
#endif

struct WgArgs {
    const char* xg;          // G8 of x  [N][x_cg][2][H][W][8]
    const char* dg;          // G8 of dy [N][d_cg][2][H][W][8]
    const float* x_amax;     // max |x| / max |dy| the G8 tensors were scaled with (null: unscaled)
    const float* d_amax;
    const char* zeros;       // >= 16 zero bytes: the source of padding / out-of-row units of the LDS-DMA copy
    float* dw;               // [kh][kw][cin_total][cout_total]
    int n, h, w;
    int x_cg, d_cg;          // channel groups of the two tensors
    int cin_total, cout_total, ci0, co0, cin, cout;   // channel window of this launch (cin, cout <= 128)
    int windows;                                       // equal cout windows handled by this launch (>= 1)
    int kh, pt, pl;
    int cit, cog, ks;        // waves = cit * cog * ks = WG_WAVES
    int chunk;               // pixels per chunk = 32 * ks
    int xp, dp;              // pixels (16-byte units) per LDS row of a channel group, pad included
    int nsplit, rows_per_split;
    float wscale;
};

constexpr int WG_WAVES = 8, WG_THREADS = WG_WAVES * 64;
// LDS-DMA instructions per wave for the x piece of a chunk: 6 cover a 64-pixel chunk of 128 channels, hi + lo, with the
// halo and the pad units (2 * 16 * 84 units of 16 bytes); 5 the dy piece of 128 channels (2 * 16 * 68)
constexpr int X_UNITS_DMA = 6;
constexpr int D_UNITS_DMA = 5;
constexpr int LDS_ROW_PAD = 4;       // units: a row of 16 k + 4 units starts 16 banks after its neighbour (see the reads)

typedef __fp16 h4raw __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 4 rows (pixels) x 16 columns (channels: two G8 groups) per 16 lanes, delivered column-major: lane i of the 16 gets channel i
// of the 4 pixels.  Every lane must be active and its address 8-byte aligned.
// The reads and their waits are volatile asm, in program order, with the waits counted by hand: to the compiler an LDS read
// behind an LDS-DMA instruction may alias the DMA's target, and it puts `s_waitcnt vmcnt(0)` in front of the read -- the
// builtin form of this read made every k-step wait for the copy of the NEXT chunk to land (the copy goes to the other
// buffer; the barrier at the end of the chunk is what orders it against its readers).
template <int OFF>
__device__ __forceinline__ void lds_read_tr(u32x2& dst, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds_read_b64_tr_b16 offset");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// wait until at most N of the LDS reads issued so far are outstanding (they return in order)
template <int N>
__device__ __forceinline__ void lgkm_wait() {
#if !(MPG_WG_EXP & 1)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
#endif
}
// no instruction: makes every later use of `frag` depend on the preceding (volatile) wait
template <class T>
__device__ __forceinline__ void tie(T& frag) {
    asm volatile("" : "+v"(frag));
}

__device__ __forceinline__ void tr_read_at(u32x2& dst, unsigned addr, int m) {      // m is a constant after unrolling
    switch (m) {
        case 0: lds_read_tr<0>(dst, addr); break;
        case 1: lds_read_tr<64>(dst, addr); break;
        case 2: lds_read_tr<128>(dst, addr); break;
        case 3: lds_read_tr<192>(dst, addr); break;
        case 4: lds_read_tr<256>(dst, addr); break;
        default: lds_read_tr<320>(dst, addr); break;
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for_wg(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_wg<I + 1, N>(f);
    }
}

template <int KW, int COTW, int PREC>
__global__ __launch_bounds__(WG_THREADS) void wgrad_mfma_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // equal output-channel windows of one layer run as one launch: linear block id = (jj * windows + window) * 8 + xcd, so
    // the windows of one row range are dispatched together and to the same XCD, and share the x rows through its L2
    // instead of streaming them from memory once per window
    const int nwin = a.windows;
    const int co_base = a.co0 + ((blockIdx.x / 8) % nwin) * a.cout;
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave number as a scalar: everything derived from it (tile ownership, loop bounds, the placement of the copy
    // instructions) then stays on the scalar unit; taken from threadIdx it is a vector value to the compiler
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int gx = xch / 8, gd = dch / 8;                        // channel groups of the two LDS images
    const int xbytes = gx * nplane * a.xp * 16, dbytes = gd * nplane * a.dp * 16;
    // the two buffers, x image then dy image each (address arithmetic, not a pointer table: indexed by the buffer number at
    // run time such a table lives in scratch memory)
    auto xs = [&](int buf) { return lds + buf * (xbytes + dbytes); };
    auto dsm = [&](int buf) { return lds + buf * (xbytes + dbytes) + xbytes; };

    // global -> LDS copy: LDS-DMA, 16 bytes per lane (the 8 channels of one pixel, as G8 keeps them), every wave instruction
    // fills 1 KiB of the image linearly.  The image is [plane][channel group][pixels of the chunk + pad units] (the pad
    // spreads the rows over the banks; it is never written or read), so unit u = (plane * groups + group) * units_per_row
    // + pixel lands at byte 16 u.
    const int xupr = a.xp, dupr = a.dp;                               // units per row incl. the pad units
    const int xpx = a.chunk + 16, dpx = a.chunk;                      // ... that are pixels
    const int xunits = gx * nplane * xupr, dunits = gd * nplane * dupr;
    constexpr int XU = X_UNITS_DMA, DU = D_UNITS_DMA;                  // DMA instructions per wave and chunk
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

    // copy plan of this thread, formed once: per unit its offset (in 16-byte units) from the chunk's first pixel in plane
    // 0 of group 0 of its image, and its pixel offset inside the chunk.  (A first version staged the chunk through registers: 10 global loads + 10 ds_write_b128 per thread
    // and chunk took 2100 + 1700 cycles next to 3840 cycles of MFMA work, with nothing to hide them behind at one wave
    // per SIMD -- measured with MPG_WG_DIAG; the wide LDS stores alone run at a third of the read rate.)
    constexpr int NOPX = 1 << 28;                 // pixel offset of a unit that delivers zeros whatever the chunk
    constexpr int SKIP = -(1 << 28);              // ... of a unit that is not copied at all (pad unit, beyond the image)
    const int plane_px = a.h * a.w;
    int xg[XU], xpo[XU], dg[DU], dpo[DU];
#pragma unroll
    for (int i = 0; i < XU; ++i) {
        const int u = tid + i * WG_THREADS;
        const int rr = u / xupr, off = u % xupr;
        const int pln = rr / gx, g = a.ci0 / 8 + rr % gx;
        xg[i] = (g * 2 + pln) * plane_px + off - 8;
        xpo[i] = (u >= xunits || off >= xpx) ? SKIP : (g < a.x_cg ? off - 8 : NOPX);
    }
#pragma unroll
    for (int i = 0; i < DU; ++i) {
        const int u = tid + i * WG_THREADS;
        const int rr = u / dupr, off = u % dupr;
        const int pln = rr / gd, g = co_base / 8 + rr % gd;
        dg[i] = (g * 2 + pln) * plane_px + off;
        dpo[i] = (u >= dunits || off >= dpx) ? SKIP : (g < a.d_cg ? off : NOPX);
    }
    const int wave_u = wave;
    // (the copy of a chunk is written in NPART parts so that variants can place them; the CU's address unit takes ~40 cycles
    // per LDS-DMA instruction of this shape: 3000 cycles for the 68 instructions of a chunk, measured)
    constexpr int NPART = XU + DU;             // one LDS-DMA instruction per part: x pieces first, then the dy pieces
    // position of the next chunk to copy: image b, output row oy, chunk ci of the row; stepped from chunk to chunk (the
    // divisions of a position formed from the chunk number were 800 cycles per chunk and wave: hipcc does them on the
    // vector unit).  Plain scalars, not a struct handed around by reference: that one ended up in scratch memory, and
    // every copy instruction then waited for a scratch load -- and with it for all the copies in flight.
    int nb = row_begin / a.h, noy = row_begin % a.h, nci = 0;
    // source of the chunk at (nb, noy, nci): rows of x and dy, first pixel, and whether the x row exists
    const char* c_xrow = nullptr;
    const char* c_drow = nullptr;
    int c_x0 = 0;
    bool c_ok = false;
    auto take_chunk = [&]() {
        const int iy = noy + ky - a.pt;
        c_x0 = nci * a.chunk;
        c_ok = iy >= 0 && iy < a.h;
        c_xrow = a.xg + (((size_t)nb * a.x_cg * 2) * plane_px + (size_t)iy * a.w + c_x0) * 16;
        c_drow = a.dg + (((size_t)nb * a.d_cg * 2) * plane_px + (size_t)noy * a.w + c_x0) * 16;
        if (++nci == chunks_per_row) {
            nci = 0;
            if (++noy == a.h) { noy = 0; ++nb; }
        }
    };
    auto dma_part = [&](int buf, auto part_c) {
        constexpr int P = decltype(part_c)::value;
        const int x0 = c_x0;
        const bool row_ok = c_ok;
        const char* xrow = c_xrow;
        const char* drow = c_drow;
        if constexpr (P < XU) {
            constexpr int i = P;
            if (xpo[i] != SKIP) {
                const int px = x0 + xpo[i];
                const char* src = (row_ok && px >= 0 && px < a.w) ? xrow + (ptrdiff_t)xg[i] * 16 : a.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs(buf) + (i * WG_WAVES + wave_u) * 1024),
                                                 16, 0, 0);
            }
        }
        if constexpr (P >= XU) {
            constexpr int i = P - XU;
            if (dpo[i] != SKIP) {
                const char* src = (row_ok && x0 + dpo[i] < a.w) ? drow + (ptrdiff_t)dg[i] * 16 : a.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dsm(buf) + (i * WG_WAVES + wave_u) * 1024),
                                                 16, 0, 0);
            }
        }
    };

    auto dma_parts = [&](int buf, int first, int end) {      // parts first .. end-1 (wave-uniform bounds)
        static_for_wg<0, NPART>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            if (first <= P && P < end) dma_part(buf, pc);
        });
    };
    if (nwork > 0) { take_chunk(); dma_parts(0, 0, NPART); }
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
        // the whole copy of the next chunk first (see MPG_WG_SPREAD), then the k-steps
        const bool has_next = wi + 1 < nwork && !(MPG_WG_EXP & 4);
        if (has_next) take_chunk();
        const int my_steps = ks < a.ks ? (ksteps - ks + a.ks - 1) / a.ks : 0;
        int next_part = 0;
#if !MPG_WG_SPREAD
        if (has_next) { dma_parts(buf ^ 1, 0, NPART); next_part = NPART; }
#endif
#if MPG_WG_DIAG
        const unsigned long long t1 = WG_T();
#endif
        // transposed reads: the 16 lanes (lane & 48) own 16 channels = two groups; lane 4q+p of them addresses pixel q,
        // channels 4p .. 4p+3 of the block and receives 4 pixels of channel (lane & 15).  Rows of 16 k + 4 units put the
        // four groups a 32-lane half reads 16 banks apart, the pixels 4 banks: no conflicts.
        const int sub = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
        const char* xb = xs(buf) + ((cit * 4 + sub * 2 + (tp >> 1)) * a.xp + 8 * hh + tq) * 16 + (tp & 1) * 8;
        const char* db = dsm(buf) + ((cog * COTW * 4 + sub * 2 + (tp >> 1)) * a.dp + 8 * hh + tq) * 16 + (tp & 1) * 8;
        const int xplane = gx * a.xp * 16, dplane = gd * a.dp * 16, dtile = 4 * a.dp * 16;
        // Per 16-pixel k-step: a 24-pixel window of x per plane -- pixels [16j + 8h - 8, 16j + 8h + 16) relative to the chunk
        // start -- as 6 transposed reads of 4 pixels, and a dy fragment (8 pixels) per cout tile and plane as two.  The reads
        // run one product ahead of their MFMAs: in issue order hi window, hi dy | lo dy, lo window, and behind the second
        // product of a k-step the hi reads of the NEXT k-step, behind the third its lo reads, so that every wait finds its
        // reads issued 5..10 MFMAs earlier.  (All 16 reads of a k-step in front of its first product, every wave of the block
        // at the same moment, took 1780 cycles per k-step for 480 of MFMA work: the waves queue at the LDS.)  The last k-step
        // of the chunk issues its look-ahead reads too, at its own addresses: the waits count them.
        constexpr int NB = 2 * COTW;
        const unsigned xa0 = lds_off(xb), da0 = lds_off(db);
        u32x2 w0[6], w1[6], b0[COTW][2], b1[COTW][2];
        auto read_hi = [&](u32x2 (&w)[6], u32x2 (&b)[COTW][2], int j) {
            const unsigned xa = xa0 + j * 256, da = da0 + j * 256;
#pragma unroll
            for (int m = 0; m < 6; ++m) tr_read_at(w[m], xa, m);
#pragma unroll
            for (int t = 0; t < COTW; ++t) {
                lds_read_tr<0>(b[t][0], da + t * dtile);
                lds_read_tr<64>(b[t][1], da + t * dtile);
            }
        };
        auto read_lo = [&](u32x2 (&w)[6], u32x2 (&b)[COTW][2], int j) {
            const unsigned xa = xa0 + xplane + j * 256, da = da0 + dplane + j * 256;
#pragma unroll
            for (int t = 0; t < COTW; ++t) {
                lds_read_tr<0>(b[t][0], da + t * dtile);
                lds_read_tr<64>(b[t][1], da + t * dtile);
            }
#pragma unroll
            for (int m = 0; m < 6; ++m) tr_read_at(w[m], xa, m);
        };
        // the KW shifted A fragments of a window: pixels [8 + s, 16 + s), s = kx - pl (|s| <= 3); register q of the window is
        // w[q / 2][q % 2]; odd shifts through v_alignbit
        auto shifted = [&](const u32x2 (&w)[6], half8 (&afr)[KW]) {
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int s = kx - (KW - 1) / 2;
                u32x4 f;
                if ((s & 1) == 0) {
                    const int q = (8 + s) / 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[i] = w[(q + i) / 2][(q + i) % 2];
                } else {
                    const int q = (7 + s) / 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        f[i] = __builtin_amdgcn_alignbit(w[(q + i + 1) / 2][(q + i + 1) % 2], w[(q + i) / 2][(q + i) % 2], 16);
                }
                afr[kx] = __builtin_bit_cast(half8, f);
            }
        };
        auto frag_of = [](const u32x2 (&b)[2]) {
            u32x4 f;
            f[0] = b[0][0]; f[1] = b[0][1]; f[2] = b[1][0]; f[3] = b[1][1];
            return __builtin_bit_cast(half8, f);
        };
        // one k-step: its hi fragments in (wc, bc), the look-ahead hi reads go to (wn, bn) -- the two sets alternate, so that no
        // register is copied while its read is still in flight (nothing but the counted waits orders these reads)
        auto kstep = [&](u32x2 (&wc)[6], u32x2 (&bc)[COTW][2], u32x2 (&wn)[6], u32x2 (&bn)[COTW][2], int j) {
            auto dma_slot = [&]() {       // the next chunk's copy: one LDS-DMA instruction in front of each product
                if (has_next && next_part < NPART) {
                    dma_parts(buf ^ 1, next_part, next_part + 1);
                    ++next_part;
                }
            };
            dma_slot();
            const int jn = j + a.ks < ksteps ? j + a.ks : j;
            half8 ah[KW], bh[COTW];
            lgkm_wait<(PREC == 3 ? NB + 6 : 0)>();
#pragma unroll
            for (int m = 0; m < 6; ++m) tie(wc[m]);
#pragma unroll
            for (int t = 0; t < COTW; ++t) { tie(bc[t][0]); tie(bc[t][1]); bh[t] = frag_of(bc[t]); }
            shifted(wc, ah);
            // the products in product-major order: consecutive MFMAs go to different accumulators
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                for (int t = 0; t < COTW; ++t)
                    acc[kx][t] = WG_MFMA(ah[kx], bh[t], acc[kx][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // the products stay in front of the reads and waits behind them
            if (PREC == 3) {
                half8 bl[COTW], al[KW];
                dma_slot();
                lgkm_wait<6>();
#pragma unroll
                for (int t = 0; t < COTW; ++t) { tie(b1[t][0]); tie(b1[t][1]); bl[t] = frag_of(b1[t]); }
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                    for (int t = 0; t < COTW; ++t)
                        acc[kx][t] = WG_MFMA(ah[kx], bl[t], acc[kx][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                read_hi(wn, bn, jn);
                dma_slot();
                lgkm_wait<6 + NB>();
#pragma unroll
                for (int m = 0; m < 6; ++m) tie(w1[m]);
                shifted(w1, al);
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                    for (int t = 0; t < COTW; ++t)
                        acc[kx][t] = WG_MFMA(al[kx], bh[t], acc[kx][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                read_lo(w1, b1, jn);
            } else {
                read_hi(wn, bn, jn);
            }
        };
        u32x2 w0b[6], b0b[COTW][2];
        if (my_steps > 0) {
            read_hi(w0, b0, ks);
            if (PREC == 3) read_lo(w1, b1, ks);
        }
        for (int t = 0, j = ks; t < my_steps; t += 2, j += 2 * a.ks) {
            kstep(w0, b0, w0b, b0b, j);
            if (t + 1 < my_steps) kstep(w0b, b0b, w0, b0, j + a.ks);
        }
        if (my_steps > 0) lgkm_wait<0>();      // the look-ahead reads of the last k-step
        if (has_next) dma_parts(buf ^ 1, next_part, NPART);     // waves without k-steps, or whatever is left
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
    const float unscale = a.wscale / ((a.x_amax ? pow2_scale(*a.x_amax) : 1.f) * (a.d_amax ? pow2_scale(*a.d_amax) : 1.f));
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

// ------------------------------------------------------------------------------------------------------------------
// The same product with ALL filter rows of a tile in one block ("ring" form; square 3x3, 4x4 and 5x5 filters).
//
// The kernel above gives a block ONE filter row: every x row is then staged once per filter row and cout window (10 times
// for a 5x5 128 -> 128 layer), every dy row once per filter row, 60 KB per chunk of 64 pixels for 3840 cycles of MFMA work
// -- and the L2 -> LDS copy is what the launch waits for (10 B/clk/CU, profiles/r03/wgrad_variants.md).  Here a block owns a
// 32-channel ci tile x a cout window x all KH x KW taps over a column chunk of 64 pixels and sweeps a range of output
// rows: the KH x rows an output row needs stay in an LDS ring of KH + 1 row slots, so ONE new x row and one dy row are
// copied per output row: 28 KB for 4800 cycles of MFMA work.  The KH * KW * COTW accumulator tiles (50 at 5x5 with a
// 64-wide window) are dealt to the 8 waves in (ky, kx, cout tile) order, 6 or 7 each; a wave reads the 24-pixel windows
// of the one or two filter rows its tiles touch.  Which tiles a wave owns is a compile-time property of its number: the
// k-step body exists once per wave (`switch` on the scalar wave number), registers are statically indexed.
struct WrArgs {
    const char* xg;
    const char* dg;
    const float* x_amax;
    const float* d_amax;
    const char* zeros;
    float* dw;
    int n, h, w;
    int x_cg, d_cg;
    int cin_total, cout_total;
    int pt;
    int chunks;              // column chunks of WR_PX pixels per row
    int ranges, rows_per_range;
    int n_ci, n_cow;         // ci tiles of 32, cout windows of 32 * COTW
    int n_outer;             // n * ranges * chunks
    float wscale;
};

constexpr int WR_PX = 64;                      // pixels of a column chunk
constexpr int WR_XP = WR_PX + 16 + LDS_ROW_PAD, WR_DP = WR_PX + LDS_ROW_PAD;      // units per LDS row of a group
constexpr int WR_XSLOT = 2 * 4 * WR_XP * 16;   // an x row: 2 planes x 4 groups
constexpr int wr_pairs(int kh, int kw, int cotw) { return kh * kw * cotw; }
constexpr int wr_first(int wave, int np) { return wave * np / WG_WAVES; }        // first (ky, kx, cot) pair of a wave
constexpr int wr_max_pairs(int np) { return (np + WG_WAVES - 1) / WG_WAVES; }

template <int KH, int KW, int COTW, int PREC>
__global__ __launch_bounds__(WG_THREADS) void wgrad_ring_kernel(WrArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NP = wr_pairs(KH, KW, COTW), MAXP = wr_max_pairs(NP), RX = KH + 1;
    constexpr int GD = COTW * 4;                                   // channel groups of the dy image
    constexpr int DBUF = 2 * GD * WR_DP * 16;                      // one dy buffer (both planes laid out; PREC 1 reads plane 0)
    constexpr int XUNITS = 2 * 4 * WR_XP, DUNITS = 2 * GD * WR_DP;
    constexpr int XU = (XUNITS + WG_THREADS - 1) / WG_THREADS, DU = (DUNITS + WG_THREADS - 1) / WG_THREADS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    // block -> (image, row range, column chunk) x (ci tile, cout window); the combinations of one outer index run next to
    // each other on one XCD (they share the x rows / the dy rows through its L2)
    const int ncombo = a.n_ci * a.n_cow;
    const int xcd = blockIdx.x % 8, t = blockIdx.x / 8;
    const int combo = t % ncombo, outer = (t / ncombo) * 8 + xcd;
    if (outer >= a.n_outer) return;
    const int cit = combo % a.n_ci, cow = combo / a.n_ci;
    const int chunk = outer % a.chunks, rng = (outer / a.chunks) % a.ranges, b = outer / (a.chunks * a.ranges);
    const int x0 = chunk * WR_PX;
    const int row0 = rng * a.rows_per_range, row1 = min(a.h, row0 + a.rows_per_range);
    if (row0 >= row1) return;
    const int plane_px = a.h * a.w;
    char* xs = lds;
    char* dsm = lds + RX * WR_XSLOT;

    // copy plan: per 16-byte unit of this thread its offset (in units) from pixel x0 of plane 0 of group 0 of the row, or
    // NOCOPY.  What a unit holds for the whole launch is decided here: pad units and the unused lo plane are never touched;
    // channel groups the tensor does not have (8 -> 128: three of the four groups of the ci tile) and pixels outside the
    // image's columns (the block owns ONE column chunk) are zeroed once, in every ring slot / both dy buffers.
    constexpr int NOCOPY = -(1 << 30);
    int xg[XU], dg[DU];
#pragma unroll
    for (int i = 0; i < XU; ++i) {
        const int u = tid + i * WG_THREADS;
        const int rr = u / WR_XP, off = u % WR_XP;
        const int pln = rr / 4, g = cit * 4 + rr % 4;
        const int px = x0 + off - 8;
        const bool unused = u >= XUNITS || off >= WR_PX + 16 || (PREC != 3 && pln == 1);
        const bool zero = !unused && (g >= a.x_cg || px < 0 || px >= a.w);
        xg[i] = (unused || zero) ? NOCOPY : (g * 2 + pln) * plane_px + off - 8;
        if (zero)
            for (int slot = 0; slot < RX; ++slot)
                *reinterpret_cast<u32x4*>(xs + slot * WR_XSLOT + u * 16) = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < DU; ++i) {
        const int u = tid + i * WG_THREADS;
        const int rr = u / WR_DP, off = u % WR_DP;
        const int pln = rr / GD, g = cow * GD + rr % GD;
        const bool unused = u >= DUNITS || off >= WR_PX || (PREC != 3 && pln == 1);
        const bool zero = !unused && (g >= a.d_cg || x0 + off >= a.w);
        dg[i] = (unused || zero) ? NOCOPY : (g * 2 + pln) * plane_px + off;
        if (zero)
            for (int bufi = 0; bufi < 2; ++bufi)
                *reinterpret_cast<u32x4*>(dsm + bufi * DBUF + u * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    const char* ximg = a.xg + ((size_t)b * a.x_cg * 2) * plane_px * 16;
    const char* dimg = a.dg + ((size_t)b * a.d_cg * 2) * plane_px * 16;
    auto copy_x_row = [&](int iy, int slot) {          // image row iy (zeros outside the image) -> ring slot
        const bool ok = iy >= 0 && iy < a.h;
        const char* row = ximg + ((size_t)(ok ? iy : 0) * a.w + x0) * 16;
#pragma unroll
        for (int i = 0; i < XU; ++i)
            if (xg[i] != NOCOPY) {
                const char* src = ok ? row + (ptrdiff_t)xg[i] * 16 : a.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs + slot * WR_XSLOT + (i * WG_WAVES + wave) * 1024),
                                                 16, 0, 0);
            }
    };
    auto copy_d_row = [&](int oy, int buf) {
        const char* row = dimg + ((size_t)oy * a.w + x0) * 16;
#pragma unroll
        for (int i = 0; i < DU; ++i)
            if (dg[i] != NOCOPY)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + (ptrdiff_t)dg[i] * 16),
                                                 (__attribute__((address_space(3))) void*)(dsm + buf * DBUF + (i * WG_WAVES + wave) * 1024),
                                                 16, 0, 0);
    };

    f32x16 acc[MAXP];
#pragma unroll
    for (int s = 0; s < MAXP; ++s)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[s][v] = 0.f;

    // ring: x row iy = oy + ky - pt lives in slot (oy + ky) % RX
    int base = row0 % RX;                              // slot of filter row 0 of the current output row
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) copy_x_row(row0 + ky - a.pt, (base + ky) % RX);
    copy_d_row(row0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int ksteps = (min(WR_PX, a.w - x0) + 15) / 16;
    // lane part of the transposed-read addresses (see wgrad_mfma_kernel): 16 lanes own 16 channels = two groups
    const int sub = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned xlane = lds_off(xs) + ((sub * 2 + (tp >> 1)) * WR_XP + 8 * hh + tq) * 16 + (tp & 1) * 8;
    const unsigned dlane = lds_off(dsm) + ((sub * 2 + (tp >> 1)) * WR_DP + 8 * hh + tq) * 16 + (tp & 1) * 8;
    constexpr int XPLANE = 4 * WR_XP * 16, DPLANE = GD * WR_DP * 16, DTILE = 4 * WR_DP * 16;

    // the k-steps of one output row for wave W: compile-time list of its (ky, kx, cot) pairs.  LDS reads run ahead of their
    // MFMAs: in issue order hi windows, hi dy, lo dy, lo windows; the hi windows of the NEXT k-step go out behind the second
    // product (the last reader of this k-step's), the rest behind the third.  The last k-step issues its look-ahead reads at
    // its own addresses: the waits count them.
    auto row_body = [&](auto wc, unsigned dbase) {
        constexpr int W = decltype(wc)::value;
        constexpr int P0 = wr_first(W, NP), P1 = wr_first(W + 1, NP), CNT = P1 - P0;
        constexpr int KY0 = P0 / (KW * COTW), KY1 = (P1 - 1) / (KW * COTW), NKY = KY1 - KY0 + 1;
        static_assert(CNT >= 1 && CNT <= MAXP && NKY <= 2, "tile split");
        u32x2 wh[NKY][6], wl[NKY][6], bhq[COTW][2], blq[COTW][2];
        unsigned xa0[NKY];
#pragma unroll
        for (int k = 0; k < NKY; ++k) {
            int slot = base + KY0 + k;
            slot = slot >= RX ? slot - RX : slot;
            xa0[k] = xlane + slot * WR_XSLOT;
        }
        const unsigned da0 = dlane + dbase;
        auto read_wh = [&](int j) {
#pragma unroll
            for (int k = 0; k < NKY; ++k)
#pragma unroll
                for (int m = 0; m < 6; ++m) tr_read_at(wh[k][m], xa0[k] + j * 256, m);
        };
        auto read_rest = [&](int j) {
            const unsigned da = da0 + j * 256;
#pragma unroll
            for (int c = 0; c < COTW; ++c) {
                lds_read_tr<0>(bhq[c][0], da + c * DTILE);
                lds_read_tr<64>(bhq[c][1], da + c * DTILE);
            }
            if (PREC == 3) {
#pragma unroll
                for (int c = 0; c < COTW; ++c) {
                    lds_read_tr<0>(blq[c][0], da + DPLANE + c * DTILE);
                    lds_read_tr<64>(blq[c][1], da + DPLANE + c * DTILE);
                }
#pragma unroll
                for (int k = 0; k < NKY; ++k)
#pragma unroll
                    for (int m = 0; m < 6; ++m) tr_read_at(wl[k][m], xa0[k] + XPLANE + j * 256, m);
            }
        };
        auto frag_of = [](const u32x2 (&q)[2]) {
            u32x4 f;
            f[0] = q[0][0]; f[1] = q[0][1]; f[2] = q[1][0]; f[3] = q[1][1];
            return __builtin_bit_cast(half8, f);
        };
        // pixels [8 + s, 16 + s) of a window, s = kx - pl
        auto shifted = [&](const u32x2 (&wq)[6], auto kxc) {
            constexpr int kx = decltype(kxc)::value, s = kx - (KW - 1) / 2;
            u32x4 f;
            if constexpr ((s & 1) == 0) {
                constexpr int q = (8 + s) / 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = wq[(q + i) / 2][(q + i) % 2];
            } else {
                constexpr int q = (7 + s) / 2;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    f[i] = __builtin_amdgcn_alignbit(wq[(q + i + 1) / 2][(q + i + 1) % 2], wq[(q + i) / 2][(q + i) % 2], 16);
            }
            return __builtin_bit_cast(half8, f);
        };
        constexpr int NW = 6 * NKY, NB = 2 * COTW;
        read_wh(0);
        read_rest(0);
        for (int j = 0; j < ksteps; ++j) {
            const int jn = j + 1 < ksteps ? j + 1 : j;
            // in flight, oldest first: wh, bh [, bl, wl]
            lgkm_wait<(PREC == 3 ? NB + NW : 0)>();
#pragma unroll
            for (int k = 0; k < NKY; ++k)
#pragma unroll
                for (int m = 0; m < 6; ++m) tie(wh[k][m]);
            half8 bh[COTW], bl[COTW];
#pragma unroll
            for (int c = 0; c < COTW; ++c) { tie(bhq[c][0]); tie(bhq[c][1]); bh[c] = frag_of(bhq[c]); }
            // product-major: consecutive MFMAs go to different accumulators
            static_for_wg<0, CNT>([&](auto sc) {
                constexpr int p = P0 + decltype(sc)::value, ky = p / (KW * COTW), kx = (p / COTW) % KW, c = p % COTW;
                acc[decltype(sc)::value] = WG_MFMA(shifted(wh[ky - KY0], std::integral_constant<int, kx>{}), bh[c], acc[decltype(sc)::value], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
            if (PREC == 3) {
                lgkm_wait<NW>();
#pragma unroll
                for (int c = 0; c < COTW; ++c) { tie(blq[c][0]); tie(blq[c][1]); bl[c] = frag_of(blq[c]); }
                static_for_wg<0, CNT>([&](auto sc) {
                    constexpr int p = P0 + decltype(sc)::value, ky = p / (KW * COTW), kx = (p / COTW) % KW, c = p % COTW;
                    acc[decltype(sc)::value] = WG_MFMA(shifted(wh[ky - KY0], std::integral_constant<int, kx>{}), bl[c], acc[decltype(sc)::value], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
                read_wh(jn);                        // behind the last reader of this k-step's hi windows
                lgkm_wait<NW>();                    // ... the lo windows are there
#pragma unroll
                for (int k = 0; k < NKY; ++k)
#pragma unroll
                    for (int m = 0; m < 6; ++m) tie(wl[k][m]);
                static_for_wg<0, CNT>([&](auto sc) {
                    constexpr int p = P0 + decltype(sc)::value, ky = p / (KW * COTW), kx = (p / COTW) % KW, c = p % COTW;
                    acc[decltype(sc)::value] = WG_MFMA(shifted(wl[ky - KY0], std::integral_constant<int, kx>{}), bh[c], acc[decltype(sc)::value], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
                read_rest(jn);
            } else {
                read_wh(jn);
                read_rest(jn);
            }
        }
        lgkm_wait<0>();                             // the look-ahead reads of the last k-step
    };

    for (int oy = row0; oy < row1; ++oy) {
        const int buf = (oy - row0) & 1;
        // the x row the NEXT output row adds and its dy row, whole at the head of the step (see MPG_WG_SPREAD above): the
        // slot it goes to held filter row 0 of the previous output row
        if (oy + 1 < row1) {
            int slot = base + KH;
            slot = slot >= RX ? slot - RX : slot;
            copy_x_row(oy + KH - a.pt, slot);
            copy_d_row(oy + 1, buf ^ 1);
        }
        const unsigned dbase = buf * DBUF;
        switch (wave) {
            case 0: row_body(std::integral_constant<int, 0>{}, dbase); break;
            case 1: row_body(std::integral_constant<int, 1>{}, dbase); break;
            case 2: row_body(std::integral_constant<int, 2>{}, dbase); break;
            case 3: row_body(std::integral_constant<int, 3>{}, dbase); break;
            case 4: row_body(std::integral_constant<int, 4>{}, dbase); break;
            case 5: row_body(std::integral_constant<int, 5>{}, dbase); break;
            case 6: row_body(std::integral_constant<int, 6>{}, dbase); break;
            default: row_body(std::integral_constant<int, 7>{}, dbase); break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        base = base + 1 >= RX ? 0 : base + 1;
    }

    // epilogue: D row = (v&3) + 8*(v>>2) + 4*(lane>>5) is the input channel, column lane&31 the output channel
    const float unscale = a.wscale / ((a.x_amax ? pow2_scale(*a.x_amax) : 1.f) * (a.d_amax ? pow2_scale(*a.d_amax) : 1.f));
    const int p0 = wave * NP / WG_WAVES, p1 = (wave + 1) * NP / WG_WAVES;
#pragma unroll
    for (int s = 0; s < MAXP; ++s) {
        const int p = p0 + s;
        if (p >= p1) break;
        const int ky = p / (KW * COTW), kx = (p / COTW) % KW, c = p % COTW;
        const int co = (cow * COTW + c) * 32 + r;
        if (co >= a.cout_total) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = cit * 32 + (v & 3) + 8 * (v >> 2) + 4 * hh;
            const float val = acc[s][v];
            if (ci < a.cin_total && val != 0.f)
                atomicAdd(a.dw + ((size_t)(ky * KW + kx) * a.cin_total + ci) * a.cout_total + co, val * unscale);
        }
    }
}

template <int KH, int KW, int COTW, int PREC>
hipError_t launch_ring(hipStream_t s, const WrArgs& a, int blocks) {
    auto kern = wgrad_ring_kernel<KH, KW, COTW, PREC>;
    static int lds_limit[64] = {0};
    constexpr size_t lds = (size_t)(KH + 1) * WR_XSLOT + 2 * (size_t)(2 * COTW * 4 * WR_DP * 16);
    static_assert(lds <= 160 * 1024, "ring kernel LDS plan");
    hipError_t e = mpg::ensure_dyn_lds((const void*)kern, 160 * 1024, lds_limit);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WG_THREADS), lds, s, a);
    return hipGetLastError();
}

template <int KW, int COTW, int PREC>
hipError_t launch(hipStream_t s, const WgArgs& a, int blocks, size_t lds_bytes, int windows) {
    auto kern = wgrad_mfma_kernel<KW, COTW, PREC>;
    static int lds_limit[64] = {0};
    hipError_t e = mpg::ensure_dyn_lds((const void*)kern, 160 * 1024, lds_limit);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks * windows), dim3(WG_THREADS), lds_bytes, s, a);
    return hipGetLastError();
}

inline size_t g8_bytes(int n, int h, int w, int c) { return (size_t)n * ((c + 7) / 8) * 2 * h * w * 16; }

// all launches of one weight gradient; dw is zeroed first (the row ranges are combined with atomics)
int wgrad_launches(hipStream_t s, const char* xg, const char* dg, const float* x_amax, const float* d_amax, int n, int h,
                   int w, int cin, int cout, int kh, int kw, float wscale, int prec, float* dw) {
    const char* zeros = mpg::zero_page();
    MPG_REQUIRE(zeros != nullptr, "mpg_conv2d_wgrad: could not allocate the zero page");
    MPG_REQUIRE((size_t)((cin > cout ? cin : cout) + 7) / 8 * 2 * h * w < (1u << 30), "mpg_conv2d_wgrad: image too large");
    hipError_t e = mpg::zero_async(dw, (size_t)kh * kw * cin * cout * sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_conv2d_wgrad: memset");
    // all filter rows of a tile in one block, x rows in an LDS ring (wgrad_ring_kernel): 1.02 .. 3x faster than the
    // one-filter-row kernel on every 3x3 / 4x4 / 5x5 shape of the training steps (tools/probe_wgrad_shapes.py,
    // profiles/r03/wgrad_variants.md); other filter shapes stay on that kernel
    if (MPG_WG_RING && kh == kw && (kh == 3 || kh == 4 || kh == 5)) {
        const int nco = (cout + 31) / 32;
        // cout tiles per window: as many as the registers of a wave take beside its windows (5x5 with a 64-wide window is 7
        // tiles on two of the waves and compiled to 28 spilled registers: 32-wide windows there, 4 tiles on one wave, 3 on
        // the others; 3x3 with a 128-wide window at three products: 7 spilled registers, 64-wide there)
        const int cotw = kh == 3 ? (nco >= 3 && prec != MPG_PREC_F16X3 ? 4 : (nco >= 2 ? 2 : 1)) : (kh == 4 && nco >= 2 ? 2 : 1);
        WrArgs a;
        a.xg = xg; a.dg = dg; a.x_amax = x_amax; a.d_amax = d_amax; a.zeros = zeros; a.dw = dw;
        a.n = n; a.h = h; a.w = w;
        a.x_cg = (cin + 7) / 8; a.d_cg = (cout + 7) / 8;
        a.cin_total = cin; a.cout_total = cout;
        a.pt = (kh - 1) / 2;
        a.wscale = wscale;
        a.chunks = (w + WR_PX - 1) / WR_PX;
        a.n_ci = (cin + 31) / 32;
        a.n_cow = (cout + 32 * cotw - 1) / (32 * cotw);
        // row ranges: a block pays for KH rows before its first output row, so ranges are as long as still leaves ~1024
        // blocks (4 per CU) to balance the chip
        const int per_range = n * a.chunks * a.n_ci * a.n_cow;
        int rs = (1024 + per_range - 1) / per_range;
        const int max_rs = h / 8 > 1 ? h / 8 : 1;
        if (rs > max_rs) rs = max_rs;
        if (rs < 1) rs = 1;
        a.rows_per_range = (h + rs - 1) / rs;
        a.ranges = (h + a.rows_per_range - 1) / a.rows_per_range;
        a.n_outer = n * a.ranges * a.chunks;
        const int blocks = ((a.n_outer + 7) / 8) * 8 * a.n_ci * a.n_cow;
        hipError_t le = hipErrorInvalidValue;
#define MPG_WR(K, C)                                                                      \
    if (kh == K && cotw == C)                                                             \
        le = prec == MPG_PREC_F16X3 ? launch_ring<K, K, C, 3>(s, a, blocks) : launch_ring<K, K, C, 1>(s, a, blocks)
        MPG_WR(3, 1); MPG_WR(3, 2); MPG_WR(4, 1); MPG_WR(4, 2); MPG_WR(5, 1);
        if (kh == 3 && cotw == 4) le = launch_ring<3, 3, 4, 1>(s, a, blocks);
#undef MPG_WR
        if (le != hipSuccess) return mpg::hip_check(le, "wgrad_ring_kernel");
        MPG_LAUNCH_CHECK("mpg_conv2d_wgrad (ring)");
    }
    // a wave keeps KW * COTW accumulator tiles of 16 registers, 6 at most beside its fragments (256 registers per wave at two
    // waves per SIMD; 8 tiles compiled to 3 spilled registers), so a 4- or 5-wide filter row takes 64 output channels per
    // window (x is then staged twice, from L2)
    const int co_step = kw >= 4 ? 64 : 128;
    const bool merged = cout > co_step && cout % co_step == 0;     // equal windows: one launch, windows on grid.y
    for (int ci0 = 0; ci0 < cin; ci0 += 128)
        for (int co0 = 0; co0 < (merged ? 1 : cout); co0 += co_step) {
            WgArgs a;
            a.xg = xg; a.dg = dg; a.x_amax = x_amax; a.d_amax = d_amax; a.dw = dw; a.zeros = zeros;
            a.n = n; a.h = h; a.w = w;
            a.x_cg = (cin + 7) / 8; a.d_cg = (cout + 7) / 8;
            a.cin_total = cin; a.cout_total = cout; a.ci0 = ci0; a.co0 = co0;
            a.cin = cin - ci0 < 128 ? cin - ci0 : 128;
            a.cout = cout - co0 < co_step ? cout - co0 : co_step;
            a.kh = kh; a.pt = (kh - 1) / 2; a.pl = (kw - 1) / 2;
            a.wscale = wscale;
            const int nci = (a.cin + 31) / 32, nco = (a.cout + 31) / 32;
            a.cit = nci > 2 ? 4 : nci;                           // 1, 2, 4 ci tiles, one per wave
            const int avail = WG_WAVES / a.cit;                   // waves left for output-channel groups
            int cog = nco >= 3 ? 4 : nco;                         // 1, 2, 4
            if (cog > avail) cog = avail;
            a.cog = cog;
            const int cotw = (nco + cog - 1) / cog;               // co tiles per wave: 1, 2
            const int npl = prec == MPG_PREC_F16X3 ? 2 : 1;
            a.ks = WG_WAVES / (a.cit * a.cog);                    // waves left over split the 16-pixel k-steps of a chunk
            a.chunk = 32 * a.ks < 64 ? 64 : 32 * a.ks;            // fewer barriers per pixel with 64-pixel chunks
            const int xu = X_UNITS_DMA;
            const int gx = a.cit * 4, gd = a.cog * cotw * 4;      // channel groups of the LDS images
            // short rows, and at most xu (x) / D_UNITS_DMA (dy) LDS-DMA instructions per wave in the copy plan
            auto x_units = [&](int chunk) { return gx * npl * (chunk + 16 + LDS_ROW_PAD); };
            auto d_units = [&](int chunk) { return gd * npl * (chunk + LDS_ROW_PAD); };
            while (a.chunk > 32 && (a.chunk / 2 >= w || x_units(a.chunk) > xu * WG_THREADS || d_units(a.chunk) > D_UNITS_DMA * WG_THREADS))
                a.chunk /= 2;
            if (a.chunk / 16 < a.ks) a.ks = a.chunk / 16;         // the other waves idle (tiny layers)
            a.xp = a.chunk + 16 + LDS_ROW_PAD;
            a.dp = a.chunk + LDS_ROW_PAD;
            const size_t lds = 2 * ((size_t)x_units(a.chunk) + d_units(a.chunk)) * 16;
            MPG_REQUIRE(lds <= 160 * 1024, "mpg_conv2d_wgrad: LDS plan %zu bytes", lds);
            MPG_REQUIRE(x_units(a.chunk) <= xu * WG_THREADS && d_units(a.chunk) <= D_UNITS_DMA * WG_THREADS, "mpg_conv2d_wgrad: copy plan");
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
            MPG_WGM(1, 1); MPG_WGM(1, 2); MPG_WGM(3, 1); MPG_WGM(3, 2); MPG_WGM(4, 1); MPG_WGM(5, 1);
#undef MPG_WGM
            if (le != hipSuccess) return mpg::hip_check(le, "wgrad_mfma_kernel");
        }
    MPG_LAUNCH_CHECK("mpg_conv2d_wgrad");
}

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
    return 256 + g8_bytes(n, h, w, cin) + g8_bytes(n, h, w, cout);
}

static int wgrad_check(int n, int h, int w, int cin, int cout, int kh, int kw, int prec) {
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && cin >= 1 && cout >= 1, "mpg_conv2d_wgrad: bad shape");
    MPG_REQUIRE(kh >= 1 && kh <= 7 && (kw == 1 || kw == 3 || kw == 4 || kw == 5), "mpg_conv2d_wgrad: filter %dx%d not built", kh, kw);
    MPG_REQUIRE(prec == MPG_PREC_F16X1 || prec == MPG_PREC_F16X3, "mpg_conv2d_wgrad: prec %d", prec);
    return MPG_OK;
}

// both operands already in G8 (the layer's forward input as the convolution kernel read it, and the scaled dy of the
// data-gradient convolution): no pass over either before the matrix kernel
extern "C" int mpg_conv2d_wgrad_g8(mpg_stream_t stream, const void* x_g8, int n, int h, int w, int cin, const void* dy_g8, int cout,
                                   int kh, int kw, float wscale, int prec, const float* x_amax, const float* dy_amax, float* dw) {
    MPG_REQUIRE(x_g8 && dy_g8 && dw, "mpg_conv2d_wgrad_g8: null pointer");
    MPG_REQUIRE(((((uintptr_t)x_g8) | ((uintptr_t)dy_g8)) & 15) == 0, "mpg_conv2d_wgrad_g8: operands must be 16-byte aligned");
    const int rc = wgrad_check(n, h, w, cin, cout, kh, kw, prec);
    if (rc != MPG_OK) return rc;
    return wgrad_launches((hipStream_t)stream, (const char*)x_g8, (const char*)dy_g8, x_amax, dy_amax, n, h, w, cin, cout, kh, kw,
                          wscale, prec, dw);
}

extern "C" int mpg_conv2d_wgrad_mfma(mpg_stream_t stream, const float* x, int n, int h, int w, int cin,
                                     const float* dy, int cout, int kh, int kw, float wscale, int prec,
                                     void* workspace, size_t workspace_bytes, const float* dy_amax, const float* x_amax,
                                     float* dw) {
    MPG_REQUIRE(x && dy && dw && workspace, "mpg_conv2d_wgrad_mfma: null pointer");
    const int rc = wgrad_check(n, h, w, cin, cout, kh, kw, prec);
    if (rc != MPG_OK) return rc;
    MPG_REQUIRE(workspace_bytes >= mpg_conv2d_wgrad_mfma_ws_bytes(n, h, w, cin, cout) &&
                    (((uintptr_t)workspace) & 255) == 0,
                "mpg_conv2d_wgrad_mfma: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    float* amax = (float*)workspace;
    char* xg = (char*)workspace + 256;
    char* dg = xg + g8_bytes(n, h, w, cin);
    // amax[0] / amax[1]: the callers' values where given (one small kernel: device-to-device copies are slower graph
    // nodes than a launch), zero where the reductions below fill them in
    hipLaunchKernelGGL(amax_init_kernel, dim3(1), dim3(64), 0, s, amax, x_amax, dy_amax);
    const size_t nx = (size_t)n * h * w * cin, nd = (size_t)n * h * w * cout;
    auto am_grid = [](size_t n) { const size_t b = (n + BLK * 16 - 1) / (BLK * 16); return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); };
    if (x_amax == nullptr)       // else: e.g. a forward activation whose scale the caller fixes (no reduction pass over x)
        hipLaunchKernelGGL(absmax_kernel, dim3(am_grid(nx)), dim3(BLK), 0, s, x, nx, (unsigned int*)amax);
    if (dy_amax == nullptr)      // else: the caller already has max |dy| (it scales the data gradient with it too)
        hipLaunchKernelGGL(absmax_kernel, dim3(am_grid(nd)), dim3(BLK), 0, s, dy, nd, (unsigned int*)(amax + 1));
    int e = mpg_f32_to_g8_scaled(stream, x, n, h, w, cin, 0, cin, MPG_G8_F16, amax, xg);
    if (e != MPG_OK) return e;
    e = mpg_f32_to_g8_scaled(stream, dy, n, h, w, cout, 0, cout, MPG_G8_F16, amax + 1, dg);
    if (e != MPG_OK) return e;
    return wgrad_launches(s, xg, dg, amax, amax + 1, n, h, w, cin, cout, kh, kw, wscale, prec, dw);
}
