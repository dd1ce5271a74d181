// Backward / training companions of the convolution:
// weight gradient (register-tiled fp32 contraction over pixels, split over the pixel range,
// accumulated with fp32 atomics), data gradient for strided / even-sized filters, batch-norm
// with batch statistics (forward + backward), activation / pixel-norm / resize / pool backward,
// lerp, per-channel sums and the TF-flavoured Adam update.  All tensors fp32 NHWC on the device.
#include "mpgan_internal.h"

namespace {

constexpr int BLK = 256;

inline unsigned grid_for(size_t n) { return (unsigned)((n + BLK - 1) / BLK); }

struct ConvGeom {
    int n, h, w, cin, oh, ow, cout, kh, kw, sh, sw, pt, pl;
};

// ---------------------------------------------------------------- weight gradient
// dw[ky][kx][ci][co] += wscale * sum_p x[p shifted by (ky,kx)][ci] * dy[p][co]
// block = one tap, a (16*MI) x (16*NI) tile of (ci, co), one slice of the pixel range.
template <int MI, int NI>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                    float* __restrict__ dw, ConvGeom g, float wscale,
                                                    int ci_tiles, int co_tiles, size_t pix_per_split) {
    constexpr int TM = 16 * MI, TN = 16 * NI, TK = 8;
    __shared__ float xs[TK][TM + 4];
    __shared__ float ds[TK][TN + 4];
    __shared__ long long xoff[TK];
    __shared__ long long doff[TK];

    int bid = blockIdx.x;
    const int co_t = bid % co_tiles; bid /= co_tiles;
    const int ci_t = bid % ci_tiles; bid /= ci_tiles;
    const int tap = bid;
    const int ky = tap / g.kw, kx = tap % g.kw;
    const int ci0 = ci_t * TM, co0 = co_t * TN;
    const size_t P = (size_t)g.n * g.oh * g.ow;
    const size_t p_begin = (size_t)blockIdx.y * pix_per_split;
    const size_t p_end = min(P, p_begin + pix_per_split);
    const int tid = threadIdx.x;
    const int ty = tid / 16, tx = tid % 16;

    float acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = 0.f;

    for (size_t p0 = p_begin; p0 < p_end; p0 += TK) {
        if (tid < TK) {
            const size_t p = p0 + tid;
            long long xo = -1, dofs = -1;
            if (p < p_end) {
                const int ox = (int)(p % g.ow);
                const size_t r = p / g.ow;
                const int oy = (int)(r % g.oh);
                const int b = (int)(r / g.oh);
                const int iy = oy * g.sh + ky - g.pt, ix = ox * g.sw + kx - g.pl;
                dofs = (long long)p * g.cout;
                if (iy >= 0 && iy < g.h && ix >= 0 && ix < g.w) xo = (((long long)b * g.h + iy) * g.w + ix) * g.cin;
            }
            xoff[tid] = xo;
            doff[tid] = dofs;
        }
        __syncthreads();
        for (int e = tid; e < TK * TM; e += BLK) {
            const int pp = e / TM, c = e % TM;
            const long long o = xoff[pp];
            xs[pp][c] = (o >= 0 && ci0 + c < g.cin) ? x[o + ci0 + c] : 0.f;
        }
        for (int e = tid; e < TK * TN; e += BLK) {
            const int pp = e / TN, c = e % TN;
            const long long o = doff[pp];
            ds[pp][c] = (o >= 0 && xoff[pp] >= 0 && co0 + c < g.cout) ? dy[o + co0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            float a[MI], bq[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = xs[k][ty * MI + i];
#pragma unroll
            for (int j = 0; j < NI; ++j) bq[j] = ds[k][tx * NI + j];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = fmaf(a[i], bq[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ci = ci0 + ty * MI + i;
        if (ci >= g.cin) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int co = co0 + tx * NI + j;
            if (co < g.cout && acc[i][j] != 0.f)
                atomicAdd(dw + ((size_t)tap * g.cin + ci) * g.cout + co, acc[i][j] * wscale);
        }
    }
}

template <int MI, int NI>
int launch_wgrad(hipStream_t s, const float* x, const float* dy, float* dw, const ConvGeom& g, float wscale) {
    constexpr int TM = 16 * MI, TN = 16 * NI;
    const int ci_tiles = (g.cin + TM - 1) / TM, co_tiles = (g.cout + TN - 1) / TN;
    const size_t tiles = (size_t)g.kh * g.kw * ci_tiles * co_tiles;
    const size_t P = (size_t)g.n * g.oh * g.ow;
    size_t split = (2048 + tiles - 1) / tiles;
    const size_t max_split = (P + 127) / 128;
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    size_t pps = (P + split - 1) / split;
    pps = (pps + 7) & ~(size_t)7;
    split = (P + pps - 1) / pps;
    hipLaunchKernelGGL((wgrad_kernel<MI, NI>), dim3((unsigned)tiles, (unsigned)split), dim3(BLK), 0, s, x, dy, dw, g,
                       wscale, ci_tiles, co_tiles, pps);
    return 0;
}

inline int micro(int c) { return c > 64 ? 8 : c > 32 ? 4 : c > 16 ? 2 : 1; }

// ---------------------------------------------------------------- data gradient (any stride / filter)
// dx[b,iy,ix,ci] = wscale * sum_{ky,kx,co} dy[b,oy,ox,co] * w[ky,kx,ci,co],  oy*sh + ky - pt == iy
// wt is the filter with the channel axes swapped, [kh,kw,cout,cin]: the lanes of a wave (adjacent ci)
// read adjacent weights while dy[co] is a broadcast
__global__ void dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wt, float* __restrict__ dx,
                             ConvGeom g, float wscale) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)g.n * g.h * g.w * g.cin;
    if (idx >= total) return;
    const int ci = idx % g.cin;
    size_t p = idx / g.cin;
    const int ix = p % g.w; p /= g.w;
    const int iy = p % g.h;
    const int b = p / g.h;
    float acc = 0.f;
    for (int ky = 0; ky < g.kh; ++ky) {
        const int ny = iy + g.pt - ky;
        if (ny < 0 || ny % g.sh) continue;
        const int oy = ny / g.sh;
        if (oy >= g.oh) continue;
        for (int kx = 0; kx < g.kw; ++kx) {
            const int nx = ix + g.pl - kx;
            if (nx < 0 || nx % g.sw) continue;
            const int ox = nx / g.sw;
            if (ox >= g.ow) continue;
            const float* dp = dy + (((size_t)b * g.oh + oy) * g.ow + ox) * g.cout;
            const float* wp = wt + (size_t)(ky * g.kw + kx) * g.cout * g.cin + ci;
            for (int co = 0; co < g.cout; ++co) acc = fmaf(dp[co], wp[(size_t)co * g.cin], acc);
        }
    }
    dx[idx] = acc * wscale;
}

#ifndef MPG_BN_TWO_PASS
#define MPG_BN_TWO_PASS 0
#endif
// ---------------------------------------------------------------- per-channel sums over pixels
// MODE 0: sum x            MODE 1: sum (x - m)^2 with m = aux0[c] * inv_n
// MODE 2: sum a, sum a*(x - mean)*invstd  (a = dy; two outputs)
// MODE 3: sum (x - k), sum (x - k)^2 with k = bn_shift(x, ch): the channel's mean over four pixels spread through the
//         batch: both batch moments in ONE pass over x.  With k within a few sigma of the mean,
//         var = E[(x-k)^2] - E[x-k]^2 loses nothing to cancellation (k = the first pixel's value did: a border pixel
//         can sit many sigma away, and the gradients through the normalisation felt it)
__device__ __forceinline__ float bn_shift(const float* __restrict__ x, size_t npix, int c, int ch) {
    const size_t q = npix >> 3;
    return 0.25f * (x[q * c + ch] + x[3 * q * c + ch] + x[5 * q * c + ch] + x[7 * q * c + ch]);
}

template <int MODE>
__global__ __launch_bounds__(256) void chan_sum_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                       size_t npix, int c, int lanes, const float* __restrict__ aux0,
                                                       const float* __restrict__ aux1, float inv_n, float eps,
                                                       float* __restrict__ out0, float* __restrict__ out1,
                                                       size_t pix_per_block, float* __restrict__ partials) {
    __shared__ float red0[BLK];
    __shared__ float red1[BLK];
    const int tid = threadIdx.x;
    const int ppi = BLK / lanes;                 // pixels per iteration
    const int lane = tid % lanes, row = tid / lanes;
    const int ch = blockIdx.y * lanes + lane;
    const size_t p_begin = (size_t)blockIdx.x * pix_per_block;
    const size_t p_end = min(npix, p_begin + pix_per_block);
    float s0 = 0.f, s1 = 0.f;
    if (ch < c) {
        float m = 0.f, is = 0.f;
        if (MODE == 1) m = aux0[ch] * inv_n;
        if (MODE == 2) { m = aux0[ch]; is = rsqrtf(aux1[ch] + eps); }
        if (MODE == 3) m = bn_shift(a, npix, c, ch);
        for (size_t p = p_begin + row; p < p_end; p += ppi) {
            const float v = a[p * c + ch];
            if (MODE == 0) s0 += v;
            if (MODE == 1) { const float d = v - m; s0 = fmaf(d, d, s0); }
            if (MODE == 2) { s0 += v; s1 = fmaf(v, (x[p * c + ch] - m) * is, s1); }
            if (MODE == 3) { const float d = v - m; s0 += d; s1 = fmaf(d, d, s1); }
        }
    }
    red0[tid] = s0;
    red1[tid] = s1;
    __syncthreads();
    if (row == 0 && ch < c) {
        for (int r = 1; r < ppi; ++r) { s0 += red0[r * lanes + lane]; s1 += red1[r * lanes + lane]; }
        if (partials != nullptr) {           // ordered form: the block's sums are kept, bn_finalize_kernel adds them in block order
            partials[((size_t)blockIdx.x * c + ch) * 2] = s0;
            partials[((size_t)blockIdx.x * c + ch) * 2 + 1] = s1;
        } else {
            atomicAdd(out0 + ch, s0);
            if (MODE == 2 || MODE == 3) atomicAdd(out1 + ch, s1);
        }
    }
}

// The same sums with four channels per thread (c % 4 == 0, 16-byte loads, four pixels in flight per thread): the scalar
// kernel above moves 1.7 TB/s on a 128-channel tensor, one 4-byte load per dependent iteration.
template <int MODE>
__global__ __launch_bounds__(256) void chan_sum4_kernel(const float* __restrict__ a, const float* __restrict__ x, size_t npix,
                                                        int c, int lanes, const float* __restrict__ aux0,
                                                        const float* __restrict__ aux1, float inv_n, float eps,
                                                        float* __restrict__ out0, float* __restrict__ out1,
                                                        size_t pix_per_block, float* __restrict__ partials) {
    __shared__ float4 red0[BLK];
    __shared__ float4 red1[BLK];
    const int tid = threadIdx.x;
    const int ppi = BLK / lanes;                 // pixels per iteration
    const int lane = tid % lanes, row = tid / lanes;
    const int ch = (blockIdx.y * lanes + lane) * 4;
    const size_t p_begin = (size_t)blockIdx.x * pix_per_block;
    const size_t p_end = min(npix, p_begin + pix_per_block);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (ch < c) {
        float m[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 1) m[j] = aux0[ch + j] * inv_n;
            if (MODE == 2) { m[j] = aux0[ch + j]; is[j] = rsqrtf(aux1[ch + j] + eps); }
            if (MODE == 3) m[j] = bn_shift(a, npix, c, ch + j);
        }
        auto take = [&](const float4 v4, const float4 x4) {
            const float v[4] = {v4.x, v4.y, v4.z, v4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 0) s0[j] += v[j];
                if (MODE == 1) { const float d = v[j] - m[j]; s0[j] = fmaf(d, d, s0[j]); }
                if (MODE == 2) { s0[j] += v[j]; s1[j] = fmaf(v[j], (xv[j] - m[j]) * is[j], s1[j]); }
                if (MODE == 3) { const float d = v[j] - m[j]; s0[j] += d; s1[j] = fmaf(d, d, s1[j]); }
            }
        };
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        size_t p = p_begin + row;
        constexpr int UN = MODE == 2 ? 4 : 8;      // pixels in flight per thread
        for (; p + (UN - 1) * (size_t)ppi < p_end; p += UN * (size_t)ppi) {
            float4 v[UN], xv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                v[u] = *reinterpret_cast<const float4*>(a + (p + (size_t)u * ppi) * c + ch);
                xv[u] = MODE == 2 ? *reinterpret_cast<const float4*>(x + (p + (size_t)u * ppi) * c + ch) : z4;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) take(v[u], xv[u]);
        }
        for (; p < p_end; p += ppi)
            take(*reinterpret_cast<const float4*>(a + p * c + ch),
                 MODE == 2 ? *reinterpret_cast<const float4*>(x + p * c + ch) : z4);
    }
    red0[tid] = make_float4(s0[0], s0[1], s0[2], s0[3]);
    red1[tid] = make_float4(s1[0], s1[1], s1[2], s1[3]);
    __syncthreads();
    if (row == 0 && ch < c) {
        float4 t0 = red0[tid], t1 = red1[tid];
        for (int r = 1; r < ppi; ++r) {
            const float4 u0 = red0[r * lanes + lane], u1 = red1[r * lanes + lane];
            t0.x += u0.x; t0.y += u0.y; t0.z += u0.z; t0.w += u0.w;
            t1.x += u1.x; t1.y += u1.y; t1.z += u1.z; t1.w += u1.w;
        }
        if (partials != nullptr) {
            float* pp = partials + ((size_t)blockIdx.x * c + ch) * 2;
            pp[0] = t0.x; pp[1] = t1.x; pp[2] = t0.y; pp[3] = t1.y; pp[4] = t0.z; pp[5] = t1.z; pp[6] = t0.w; pp[7] = t1.w;
        } else {
            atomicAdd(out0 + ch, t0.x); atomicAdd(out0 + ch + 1, t0.y); atomicAdd(out0 + ch + 2, t0.z); atomicAdd(out0 + ch + 3, t0.w);
            if (MODE == 2 || MODE == 3) {
                atomicAdd(out1 + ch, t1.x); atomicAdd(out1 + ch + 1, t1.y); atomicAdd(out1 + ch + 2, t1.z); atomicAdd(out1 + ch + 3, t1.w);
            }
        }
    }
}

template <int MODE>
int launch_chan_sum(hipStream_t s, const float* a, const float* x, size_t npix, int c, const float* aux0,
                    const float* aux1, float inv_n, float eps, float* out0, float* out1, float* partials = nullptr) {
    if (c >= 16 && (c % 4) == 0 && (((uintptr_t)a) & 15) == 0 && (x == nullptr || (((uintptr_t)x) & 15) == 0)) {
        int lanes = 1;
        while (lanes < c / 4 && lanes < BLK) lanes <<= 1;
        const int cblocks = (c / 4 + lanes - 1) / lanes;
        const int ppi = BLK / lanes;
        // every block ends in one atomic per channel: 2048 blocks queued ~100 us of atomics on each address
        size_t blocks = (npix + (size_t)ppi * 16 - 1) / ((size_t)ppi * 16);
        if (blocks > 512) blocks = 512;
        if (blocks < 1) blocks = 1;
        const size_t ppb = (npix + blocks - 1) / blocks;
        blocks = (npix + ppb - 1) / ppb;
        hipLaunchKernelGGL((chan_sum4_kernel<MODE>), dim3((unsigned)blocks, cblocks), dim3(BLK), 0, s, a, x, npix, c, lanes,
                           aux0, aux1, inv_n, eps, out0, out1, ppb, partials);
        return (int)blocks;
    }
    int lanes = 1;
    while (lanes < c && lanes < BLK) lanes <<= 1;
    const int cblocks = (c + lanes - 1) / lanes;
    const int ppi = BLK / lanes;
    size_t blocks = (npix + (size_t)ppi * 16 - 1) / ((size_t)ppi * 16);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const size_t ppb = (npix + blocks - 1) / blocks;
    blocks = (npix + ppb - 1) / ppb;
    hipLaunchKernelGGL((chan_sum_kernel<MODE>), dim3((unsigned)blocks, cblocks), dim3(BLK), 0, s, a, x, npix, c, lanes,
                       aux0, aux1, inv_n, eps, out0, out1, ppb, partials);
    return (int)blocks;
}
constexpr int CHAN_SUM_MAX_BLOCKS = 1024;       // pixel blocks of either kernel (the size of a `partials` buffer: blocks x c x 2)

// The ordered form of the sums: the blocks' partial sums ([block][channel][2]) added in a FIXED order -- 16 interleaved runs
// of blocks per channel, then the 16 runs in sequence -- so that the result does not depend on the order the blocks ran in
// (atomics: it does, at 1e-7 relative, enough to flip a ReLU mask element next to zero now and then).
__global__ __launch_bounds__(256) void sum_partials_kernel(const float2* __restrict__ partials, int nblocks, int c,
                                                           float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float2 red[256];
    const int tid = threadIdx.x, cl = tid & 15, seg = tid >> 4;
    const int ch = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (ch < c) {
        int b = seg;
        for (; b + 7 * 16 < nblocks; b += 8 * 16) {          // eight loads in flight, added in block order
            float2 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = partials[(size_t)(b + 16 * u) * c + ch];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += p[u].x; s1 += p[u].y; }
        }
        for (; b < nblocks; b += 16) {
            const float2 p = partials[(size_t)b * c + ch];
            s0 += p.x;
            s1 += p.y;
        }
    }
    red[tid] = make_float2(s0, s1);
    __syncthreads();
    if (seg == 0 && ch < c) {
        for (int k = 1; k < 16; ++k) { s0 += red[k * 16 + cl].x; s1 += red[k * 16 + cl].y; }
        out0[ch] = s0;
        out1[ch] = s1;
    }
}

// sums -> batch mean / biased variance, and the moving averages of tf.contrib batch_norm
// (moving = decay * moving + (1 - decay) * batch) when their pointers are given
__global__ void bn_finalize_kernel(float* mean, float* var, int c, float inv_n, float* moving_mean,
                                   float* moving_var, float decay, const float* __restrict__ shift, size_t shift_npix) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= c) return;
    float m = mean[i] * inv_n, v = var[i] * inv_n;
    if (shift != nullptr) {         // sums of (x - k) and (x - k)^2 over the `shift` tensor of shift_npix pixels
        v = fmaxf(v - m * m, 0.f);
        m += bn_shift(shift, shift_npix, c, i);
    }
    mean[i] = m;
    var[i] = v;
    if (moving_mean) moving_mean[i] = decay * moving_mean[i] + (1.f - decay) * m;
    if (moving_var) moving_var[i] = decay * moving_var[i] + (1.f - decay) * v;
}

// four channels per thread (c % 4 == 0, c <= 512): the per-channel vectors (possibly unaligned views into a flat parameter
// buffer) are staged in LDS once per block -- 16 scalar parameter loads per thread made this kernel slower than the
// one-element version -- and every thread streams 16 bytes of x per step of a grid-stride loop
constexpr int BN4_CMAX = 512;

__global__ __launch_bounds__(256) void bn_apply4_kernel(const float* __restrict__ x, size_t total4, int c,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        int act, float leak, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float par[4][BN4_CMAX];
    for (int i = threadIdx.x; i < c; i += BLK) {
        par[0][i] = mean[i];
        par[1][i] = rsqrtf(var[i] + eps);
        par[2][i] = gamma[i];
        par[3][i] = beta[i];
    }
    __syncthreads();
    for (size_t i4 = (size_t)blockIdx.x * BLK + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * BLK) {
        const int ch = (int)((i4 * 4) % (size_t)c);
        const float4 v = reinterpret_cast<const float4*>(x)[i4];
        const float4 m = *reinterpret_cast<const float4*>(&par[0][ch]), is = *reinterpret_cast<const float4*>(&par[1][ch]);
        const float4 g = *reinterpret_cast<const float4*>(&par[2][ch]), bt = *reinterpret_cast<const float4*>(&par[3][ch]);
        float4 o;
        o.x = mpg::apply_act((v.x - m.x) * is.x * g.x + bt.x, act, leak);
        o.y = mpg::apply_act((v.y - m.y) * is.y * g.y + bt.y, act, leak);
        o.z = mpg::apply_act((v.z - m.z) * is.z * g.z + bt.z, act, leak);
        o.w = mpg::apply_act((v.w - m.w) * is.w * g.w + bt.w, act, leak);
        reinterpret_cast<float4*>(y)[i4] = o;
    }
}

// y = act((x - mean) * rsqrt(var + eps) * gamma + beta)
__global__ void bn_apply_kernel(const float* __restrict__ x, size_t total, int c, const float* __restrict__ mean,
                                const float* __restrict__ var, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int act, float leak, float* __restrict__ y) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= total) return;
    const int ch = idx % c;
    const float v = (x[idx] - mean[ch]) * rsqrtf(var[ch] + eps) * gamma[ch] + beta[ch];
    y[idx] = mpg::apply_act(v, act, leak);
}

// max |v| of a block into *out (float bits of a non-negative value order like unsigned ints): wave shuffles, LDS across the
// waves, ONE atomic per block.  The kernels that use it run grid-stride with a capped grid, so a tensor costs a few
// thousand atomics (one per wave on a single address made bn_bwd_apply_kernel 9x slower: measured, round 2).
constexpr unsigned AMAX_GRID = 2048;

__device__ __forceinline__ void block_absmax_to(float m, unsigned int* __restrict__ out) {
    __shared__ float wave_max[BLK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = wave_max[0];
#pragma unroll
        for (int w = 1; w < BLK / 64; ++w) b = fmaxf(b, wave_max[w]);
        atomicMax(out, __float_as_uint(b));
    }
}

// dx = gamma * invstd * (dy - dbeta/N - xhat * dgamma/N)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, size_t total, int c,
                                    const float* __restrict__ mean, const float* __restrict__ var,
                                    const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, float eps, float inv_n, float* __restrict__ dx,
                                    unsigned int* __restrict__ amax) {
    float m = 0.f;
    for (size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * BLK) {
        const int ch = idx % c;
        const float is = rsqrtf(var[ch] + eps);
        const float xh = (x[idx] - mean[ch]) * is;
        const float v = gamma[ch] * is * (dy[idx] - dbeta[ch] * inv_n - xh * dgamma[ch] * inv_n);
        dx[idx] = v;
        m = fmaxf(m, fabsf(v));
    }
    if (amax != nullptr) block_absmax_to(m, amax);
}

// ---------------------------------------------------------------- elementwise backward
// derivative expressed through the activation OUTPUT y (relu: y>0; lrelu: slope 1 / leak by sign of y,
// 0.5(1+leak) at 0 as tf.abs has a zero gradient there, GAN.py:733-737; tanh: 1 - y^2)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, size_t n, int act,
                               float leak, float* __restrict__ dx, unsigned int* __restrict__ amax) {
    float m = 0.f;
    auto one = [&](float g, float o) {
        float d = 1.f;
        if (act == MPG_ACT_RELU) d = o > 0.f ? 1.f : 0.f;
        else if (act == MPG_ACT_LRELU) d = o > 0.f ? 1.f : (o < 0.f ? leak : 0.5f * (1.f + leak));
        else if (act == MPG_ACT_TANH) d = 1.f - o * o;
        const float v = g * d;
        m = fmaxf(m, fabsf(v));
        return v;
    };
    const bool vec = ((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)dx)) & 15) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLK) {
        const float4 g = reinterpret_cast<const float4*>(dy)[i], o = reinterpret_cast<const float4*>(y)[i];
        reinterpret_cast<float4*>(dx)[i] = make_float4(one(g.x, o.x), one(g.y, o.y), one(g.z, o.z), one(g.w, o.w));
    }
    for (size_t idx = n4 * 4 + (size_t)blockIdx.x * BLK + threadIdx.x; idx < n; idx += (size_t)gridDim.x * BLK)
        dx[idx] = one(dy[idx], y[idx]);
    if (amax != nullptr) block_absmax_to(m, amax);
}

// y = x * r, r = rsqrt(mean_c x^2 + eps);  dx = r * (dy - y * mean_c(dy * y)); `lanes` consecutive lanes per pixel
__global__ void pixel_norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, size_t npix, int c,
                                      int lanes, float eps, float* __restrict__ dx) {
    const size_t gid = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t pix = gid / lanes;
    const int l = (int)(gid % lanes);
    const bool ok = pix < npix;
    const float* xp = x + (ok ? pix : 0) * c;
    const float* dp = dy + (ok ? pix : 0) * c;
    float ss = 0.f, dot = 0.f;
    if (ok)
        for (int i = l; i < c; i += lanes) { ss = fmaf(xp[i], xp[i], ss); dot = fmaf(dp[i], xp[i], dot); }
    for (int m = lanes >> 1; m > 0; m >>= 1) { ss += __shfl_xor(ss, m); dot += __shfl_xor(dot, m); }
    if (!ok) return;
    const float r = rsqrtf(ss / c + eps);
    const float k = dot * r * r / c;
    for (int i = l; i < c; i += lanes) dx[pix * c + i] = r * (dp[i] - xp[i] * k);
}

// nearest upsample by integer factors: dx[iy,ix] = sum of the fy x fx block of dy
__global__ void resize_nearest_bwd_kernel(const float* __restrict__ dy, int n, int oh, int ow, int c,
                                          float* __restrict__ dx, int h, int w, float scale) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * h * w * c;
    if (idx >= total) return;
    const int ch = idx % c;
    size_t p = idx / c;
    const int ix = p % w; p /= w;
    const int iy = p % h;
    const int b = p / h;
    const int fy = oh / h, fx = ow / w;
    float s = 0.f;
    for (int dyy = 0; dyy < fy; ++dyy)
        for (int dxx = 0; dxx < fx; ++dxx)
            s += dy[(((size_t)b * oh + iy * fy + dyy) * ow + ix * fx + dxx) * c + ch];
    dx[idx] = s * scale;
}

// 2x2 average pool backward: every input pixel receives a quarter of its output pixel
__global__ void avg_pool2_bwd_kernel(const float* __restrict__ dy, int n, int h, int w, int c,
                                     float* __restrict__ dx) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * h * w * c;
    if (idx >= total) return;
    const int ch = idx % c;
    size_t p = idx / c;
    const int ix = p % w; p /= w;
    const int iy = p % h;
    const int b = p / h;
    const int oh = h / 2, ow = w / 2;
    const int oy = iy / 2, ox = ix / 2;
    dx[idx] = (oy < oh && ox < ow) ? 0.25f * dy[(((size_t)b * oh + oy) * ow + ox) * c + ch] : 0.f;
}

// out = x + (y - x) * t   (x may be null: zeros)
__global__ void lerp_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t n, float t,
                            float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= n) return;
    const float a = x ? x[idx] : 0.f;
    out[idx] = a + (y[idx] - a) * t;
}

// tf.train.AdamOptimizer: p -= lr_t * m / (sqrt(v) + eps), lr_t = lr * sqrt(1-b2^t) / (1-b1^t)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ gr, float* __restrict__ m,
                            float* __restrict__ v, size_t n, const float* __restrict__ lr_ptr, float b1, float b2,
                            float eps) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= n) return;
    const float lr_t = *lr_ptr;
    const float g = gr[idx];
    const float mm = m[idx] + (g - m[idx]) * (1.f - b1);
    const float vv = v[idx] + (g * g - v[idx]) * (1.f - b2);
    m[idx] = mm;
    v[idx] = vv;
    p[idx] -= lr_t * mm / (sqrtf(vv) + eps);
}

// long contractions (the discriminator's flatten -> 1: k = 262144 at 256^2 tiles) with one block per row leave the chip
// empty (16 blocks, 320 us): split K over blockIdx.z, partial sums by atomics into a zeroed y, bias + activation after
__global__ __launch_bounds__(256) void fc_splitk_kernel(const float* __restrict__ x, const float* __restrict__ w, int k, int cout,
                                                        int kchunk, float wscale, float* __restrict__ y) {
    __shared__ float red[BLK];
    const int row = blockIdx.x, o0 = blockIdx.y * 64, tid = threadIdx.x;
    const int no = min(64, cout - o0);
    const int k0 = blockIdx.z * kchunk, k1 = min(k, k0 + kchunk);
    const float* xr = x + (size_t)row * k;
    if (no == 1) {
        float s = 0.f;
        for (int i = k0 + tid; i < k1; i += BLK) s = fmaf(xr[i], w[(size_t)i * cout + o0], s);
        red[tid] = s;
        __syncthreads();
        for (int st = BLK / 2; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) atomicAdd(y + (size_t)row * cout + o0, red[0] * wscale);
        return;
    }
    const int o = tid % 64, kl = tid / 64;
    float s = 0.f;
    if (o < no)
        for (int i = k0 + kl; i < k1; i += 4) s = fmaf(xr[i], w[(size_t)i * cout + o0 + o], s);
    red[tid] = s;
    __syncthreads();
    if (kl == 0 && o < no)
        atomicAdd(y + (size_t)row * cout + o0 + o, (red[o] + red[64 + o] + red[128 + o] + red[192 + o]) * wscale);
}

__global__ void fc_finish_kernel(float* __restrict__ y, const float* __restrict__ bias, size_t n, int cout, int act, float leak) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= n) return;
    y[idx] = mpg::apply_act(y[idx] + (bias ? bias[idx % cout] : 0.f), act, leak);
}

// fully connected layer: y[r][o] = act(wscale * sum_k x[r][k] * w[k][o] + b[o]); one block per (row, 64 outputs),
// the K range is strided over the block and reduced through LDS
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int k, int cout, float wscale,
                                                     int act, float leak, float* __restrict__ y) {
    __shared__ float red[BLK];
    const int row = blockIdx.x;
    const int o0 = blockIdx.y * 64;
    const int no = min(64, cout - o0);
    const int tid = threadIdx.x;
    const float* xr = x + (size_t)row * k;
    if (no == 1) {
        float s = 0.f;
        for (int i = tid; i < k; i += BLK) s = fmaf(xr[i], w[(size_t)i * cout + o0], s);
        red[tid] = s;
        __syncthreads();
        for (int st = BLK / 2; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) y[(size_t)row * cout + o0] = mpg::apply_act(red[0] * wscale + (bias ? bias[o0] : 0.f), act, leak);
        return;
    }
    // 4 k-lanes x 64 outputs
    const int o = tid % 64, kl = tid / 64;
    float s = 0.f;
    if (o < no)
        for (int i = kl; i < k; i += 4) s = fmaf(xr[i], w[(size_t)i * cout + o0 + o], s);
    red[tid] = s;
    __syncthreads();
    if (kl == 0 && o < no) {
        s = red[o] + red[64 + o] + red[128 + o] + red[192 + o];
        y[(size_t)row * cout + o0 + o] = mpg::apply_act(s * wscale + (bias ? bias[o0 + o] : 0.f), act, leak);
    }
}

// out[0] += sum_i |a_i - b_i| (mode 0) or sum_i (a_i - b_i)^2 (mode 1); block partials, one atomic per block
__global__ __launch_bounds__(256) void pair_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          size_t n, int mode, float* __restrict__ out) {
    __shared__ float red[BLK];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
        const float d = a[i] - (b ? b[i] : 0.f);
        s += mode == 0 ? fabsf(d) : d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = BLK / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, red[0]);
}

// tensorResample (multipassGAN-4x.py:398-441), 2D: bilinear look-up of value[b] at pos[b,i,j] = (y, x) given in
// cell-centred coordinates (sample k sits at k + 0.5).  floor / ceil indices are clamped to the grid when
// `clamp` (the script default, :125) and the weights 1 - |p - 0.5 - idx| use the CLAMPED index, as the
// reference does; without clamping samples outside the grid read zero.
__device__ __forceinline__ void resample_taps(float p, int n, int clamp, int (&idx)[2], float (&wt)[2]) {
    const float f = p - 0.5f;
    int i0 = (int)floorf(f), i1 = i0 + 1;
    if (clamp) {
        i0 = min(max(i0, 0), n - 1);
        i1 = min(max(i1, 0), n - 1);
    }
    idx[0] = i0; idx[1] = i1;
    wt[0] = 1.f - fabsf(f - (float)i0);
    wt[1] = 1.f - fabsf(f - (float)i1);
}

__global__ void resample_kernel(const float* __restrict__ v, const float* __restrict__ pos, int n, int h, int w, int c,
                                int clamp, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * h * w * c;
    if (idx >= total) return;
    const int ch = idx % c;
    const size_t pix = idx / c;
    const int b = pix / ((size_t)h * w);
    int iy[2], ix[2];
    float wy[2], wx[2];
    resample_taps(pos[pix * 2], h, clamp, iy, wy);
    resample_taps(pos[pix * 2 + 1], w, clamp, ix, wx);
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (iy[a] >= 0 && iy[a] < h && ix[e] >= 0 && ix[e] < w)
                s = fmaf(v[(((size_t)b * h + iy[a]) * w + ix[e]) * c + ch], wy[a] * wx[e], s);
    out[idx] = s;
}

// gradient with respect to value: scatter-add of dy with the same taps (dv zeroed by the caller)
__global__ void resample_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pos, int n, int h, int w,
                                    int c, int clamp, float* __restrict__ dv) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * h * w * c;
    if (idx >= total) return;
    const int ch = idx % c;
    const size_t pix = idx / c;
    const int b = pix / ((size_t)h * w);
    int iy[2], ix[2];
    float wy[2], wx[2];
    resample_taps(pos[pix * 2], h, clamp, iy, wy);
    resample_taps(pos[pix * 2 + 1], w, clamp, ix, wx);
    const float g = dy[idx];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (iy[a] >= 0 && iy[a] < h && ix[e] >= 0 && ix[e] < w)
                atomicAdd(dv + (((size_t)b * h + iy[a]) * w + ix[e]) * c + ch, g * wy[a] * wx[e]);
}

int fill_geom(ConvGeom& g, int n, int h, int w, int cin, int cout, int kh, int kw, int sh, int sw) {
    g.n = n; g.h = h; g.w = w; g.cin = cin; g.cout = cout; g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw;
    g.oh = (h + sh - 1) / sh;
    g.ow = (w + sw - 1) / sw;
    int ph = (g.oh - 1) * sh + kh - h; if (ph < 0) ph = 0;
    int pw = (g.ow - 1) * sw + kw - w; if (pw < 0) pw = 0;
    g.pt = ph / 2;
    g.pl = pw / 2;
    return 0;
}


// ---------------------------------------------------------------------------------------------
// GAN.advect (tools_wscale/GAN.py:173-418), 2D, square fields: HBM-bound gathers
// ---------------------------------------------------------------------------------------------
// legacy tf.image.resize_images(method 0) value of channel `ch` of vel[b] at output pixel (i, j); 0 past the grid
__device__ __forceinline__ float adv_bilinear(const float* __restrict__ vel, int b, int hv, int wv, int cv, int ch, int i,
                                              int j, int h, int w) {
    if (i >= h || j >= w) return 0.f;
    const float sy = (float)i * ((float)hv / (float)h), sx = (float)j * ((float)wv / (float)w);
    const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    const int y1 = min(y0 + 1, hv - 1), x1 = min(x0 + 1, wv - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const float* base = vel + (size_t)b * hv * wv * cv + ch;
    const float v00 = base[((size_t)y0 * wv + x0) * cv], v01 = base[((size_t)y0 * wv + x1) * cv];
    const float v10 = base[((size_t)y1 * wv + x0) * cv], v11 = base[((size_t)y1 * wv + x1) * cv];
    const float top = v00 + (v01 - v00) * fx, bot = v10 + (v11 - v10) * fx;
    return top + (bot - top) * fy;
}

// :376-396 in one pass: channels (x,y) -> (y,x), bilinear resize to [h,w], times the resolution ratio, MAC -> centre
// (average with the successor along the component's own axis, zero past the end), times dt * (+1, 0, -1)[b % 3]
__global__ void advect_velocity_kernel(const float* __restrict__ vel, int n, int hv, int wv, int cv, int h, int w, float dt,
                                       float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * h * w;
    if (idx >= total) return;
    const int j = idx % w;
    const int i = (idx / w) % h;
    const int b = idx / ((size_t)w * h);
    const float rh = (float)h / (float)hv, rw = (float)w / (float)wv;
    const float up = fmaxf(rh, rw);
    const float vy = adv_bilinear(vel, b, hv, wv, cv, 1, i, j, h, w) * up;
    const float vy_n = adv_bilinear(vel, b, hv, wv, cv, 1, i + 1, j, h, w) * up;
    const float vx = adv_bilinear(vel, b, hv, wv, cv, 0, i, j, h, w) * up;
    const float vx_n = adv_bilinear(vel, b, hv, wv, cv, 0, i, j + 1, h, w) * up;
    const int ph = b % 3;
    const float step = ph == 0 ? dt : (ph == 1 ? 0.f : -dt);
    out[idx * 2] = 0.5f * (vy + vy_n) * step;
    out[idx * 2 + 1] = 0.5f * (vx + vx_n) * step;
}

struct AdvCorner {
    int y[2], x[2];
    float wy[2], wx[2];
};
// :175-190: p = (i + 1, j + 1) - sign * vel; q = p - 0.5; indices floor(q), floor(q) + 1 clamped; weights 1 - |q - index|
__device__ __forceinline__ AdvCorner adv_corners(const float* __restrict__ vel, size_t pix, int i, int j, int h, int w, float sign) {
    AdvCorner c;
    const float qy = ((float)i + 1.0f - sign * vel[pix * 2]) - 0.5f;
    const float qx = ((float)j + 1.0f - sign * vel[pix * 2 + 1]) - 0.5f;
    const int y0 = (int)floorf(qy), x0 = (int)floorf(qx);
    c.y[0] = min(max(y0, 0), h - 1); c.y[1] = min(max(y0 + 1, 0), h - 1);
    c.x[0] = min(max(x0, 0), w - 1); c.x[1] = min(max(x0 + 1, 0), w - 1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        c.wy[k] = 1.0f - fabsf(qy - (float)c.y[k]);
        c.wx[k] = 1.0f - fabsf(qx - (float)c.x[k]);
    }
    return c;
}

__global__ void semi_lagrange_kernel(const float* __restrict__ src, const float* __restrict__ vel, int n, int h, int w, int c,
                                     float sign, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * h * w * c;
    if (idx >= total) return;
    const int ch = idx % c;
    const size_t pix = idx / c;
    const int j = pix % w;
    const int i = (pix / w) % h;
    const int b = pix / ((size_t)w * h);
    const AdvCorner k = adv_corners(vel, pix, i, j, h, w, sign);
    const float* s = src + (size_t)b * h * w * c + ch;
    float acc = 0.f;
    // corner order of the reference: bit 0 selects the upper index on axis 0 (y), bit 1 on axis 1 (x)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += s[((size_t)k.y[q & 1] * w + k.x[q >> 1]) * c] * (k.wy[q & 1] * k.wx[q >> 1]);
    out[idx] = acc;
}

__global__ void semi_lagrange_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ vel, int n, int h, int w, int c,
                                         float sign, float* __restrict__ dsrc) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * h * w * c;
    if (idx >= total) return;
    const int ch = idx % c;
    const size_t pix = idx / c;
    const int j = pix % w;
    const int i = (pix / w) % h;
    const int b = pix / ((size_t)w * h);
    const AdvCorner k = adv_corners(vel, pix, i, j, h, w, sign);
    float* d = dsrc + (size_t)b * h * w * c + ch;
    const float g = dy[idx];
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(d + ((size_t)k.y[q & 1] * w + k.x[q >> 1]) * c, g * (k.wy[q & 1] * k.wx[q >> 1]));
}

// MacCormackCorrect + MacCormackClamp (:206-343), one channel: corrected = forward + strength/2 (source - backward) in
// fluid cells (flags < 0.2); kept only if it lies inside [min, max] of source over the fluid cells among the 2x2
// neighbourhood of the truncated look-up position, else the semi-Lagrangian value.  Index clipping as the reference
// writes it: the first corner is clipped per axis (h-1, w-1), the other three -- batch index included -- to w-1.
__global__ void maccormack_kernel(const float* __restrict__ src, const float* __restrict__ fwd, const float* __restrict__ bwd,
                                  const float* __restrict__ flags, const float* __restrict__ vel, int n, int h, int w,
                                  float strength, float* __restrict__ out, float* __restrict__ keep) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * h * w;
    if (idx >= total) return;
    const int j = idx % w;
    const int i = (idx / w) % h;
    const int b = idx / ((size_t)w * h);
    const float f = fwd[idx];
    const bool fluid = flags[idx] < 0.2f;
    const float corr = fluid ? f + strength * 0.5f * (src[idx] - bwd[idx]) : f;
    const int cy = (int)(((float)i + 1.0f) - vel[idx * 2]), cx = (int)(((float)j + 1.0f) - vel[idx * 2 + 1]);   // truncation
    const int i0 = min(max(cy, 0), h - 1), j0 = min(max(cx, 0), w - 1);
    const float big = 9223372036854775807.0f;
    float lo = big, hi = -big - 1.0f;
    const float lo_i = lo, hi_i = hi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int di = q & 1, dj = q >> 1;
        int bb = b, ii = i0 + di, jj = j0 + dj;
        if (q != 0) { bb = min(max(bb, 0), w - 1); ii = min(max(ii, 0), w - 1); jj = min(max(jj, 0), w - 1); }
        const size_t at = ((size_t)bb * h + ii) * w + jj;
        if (flags[at] < 0.2f) {
            const float sv = src[at];
            lo = fminf(lo, sv);
            hi = fmaxf(hi, sv);
        }
    }
    const bool reject = corr < lo || corr > hi || lo == lo_i || hi == hi_i;
    out[idx] = reject ? f : corr;
    if (keep != nullptr) keep[idx] = (!reject && fluid) ? 1.f : 0.f;   // where the correction term carries gradient
}


// ---------------------------------------------------------------------------------------------
// staged Adam with dynamic loss scaling (multipassGAN-8x.py:490-541, 1305-1362), all decisions on the device so that a
// captured hipGraph of the iteration replays them.  state (8 floats): [0] ls_var (log2 of the loss scale), [1] coef =
// exp(-ls_var ln 2) / total_grads, [2] 1 if every scaled gradient is finite, [3] t = number of applied updates,
// [4] lr_t of the update being attempted.
// ---------------------------------------------------------------------------------------------
__global__ void ls_begin_kernel(float* __restrict__ st, const float* __restrict__ lr, float inv_total, int use_ls, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st[1] = use_ls ? inv_total * expf(-st[0] * 0.69314718f) : 1.f;      // undo_loss_scaling(1/total_grads), :528-530
    st[2] = 1.f;
    const float t = st[3] + 1.f;
    st[4] = *lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
}

__global__ void ls_check_kernel(const float* __restrict__ g, const float* __restrict__ mask, size_t n, float* __restrict__ st) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= n) return;
    if (mask != nullptr && mask[idx] == 0.f) return;
    if (!isfinite(g[idx] * st[1])) st[2] = 0.f;                          // tf.reduce_all(tf.is_finite(g)), :533-534
}

// masked Adam on the scaled gradient; skipped as a whole when the finite check failed (tf.cond, :537-539)
__global__ void adam_staged_kernel(float* __restrict__ p, const float* __restrict__ gr, float* __restrict__ m,
                                   float* __restrict__ v, const float* __restrict__ mask, size_t n,
                                   const float* __restrict__ st, float b1, float b2, float eps, float* __restrict__ shadow,
                                   float ema_decay) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= n) return;
    if (st[2] == 0.f) return;
    if (mask != nullptr && mask[idx] == 0.f) return;
    const float g = gr[idx] * st[1];
    const float mm = m[idx] + (g - m[idx]) * (1.f - b1);
    const float vv = v[idx] + (g * g - v[idx]) * (1.f - b2);
    m[idx] = mm;
    v[idx] = vv;
    const float pn = p[idx] - st[4] * mm / (sqrtf(vv) + eps);
    p[idx] = pn;
    // tf.contrib.opt.MovingAverageOptimizer: shadow -= (1 - decay) (shadow - var) after the update (:1356)
    if (shadow != nullptr) shadow[idx] -= (1.f - ema_decay) * (shadow[idx] - pn);
}

__global__ void ls_end_kernel(float* __restrict__ st, int use_ls, float inc, float dec) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st[2] != 0.f) {
        st[3] += 1.f;
        if (use_ls) st[0] += inc;                                        // tf.assign_add(ls_var, loss_scaling_inc)
    } else if (use_ls) {
        st[0] -= dec;                                                    // tf.assign_sub(ls_var, loss_scaling_dec)
    }
}

}  // namespace

#define MPG_GEOM_CHECK(NAME)                                                                                  \
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && cin >= 1 && cout >= 1, NAME ": bad shape");                     \
    MPG_REQUIRE(kh >= 1 && kw >= 1 && kh <= 16 && kw <= 16, NAME ": bad filter %dx%d", kh, kw);               \
    MPG_REQUIRE(stride_h >= 1 && stride_w >= 1, NAME ": bad stride")

extern "C" int mpg_conv2d_wgrad(mpg_stream_t stream, const float* x, int n, int h, int w, int cin, const float* dy,
                                int cout, int kh, int kw, int stride_h, int stride_w, float wscale, float* dw) {
    MPG_REQUIRE(x && dy && dw, "mpg_conv2d_wgrad: null pointer");
    MPG_GEOM_CHECK("mpg_conv2d_wgrad");
    ConvGeom g;
    fill_geom(g, n, h, w, cin, cout, kh, kw, stride_h, stride_w);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = mpg::zero_async(dw, (size_t)kh * kw * cin * cout * sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_conv2d_wgrad: memset");
    const int mi = micro(cin), ni = micro(cout);
#define MPG_WG(M, N) if (mi == M && ni == N) launch_wgrad<M, N>(s, x, dy, dw, g, wscale)
    MPG_WG(1, 1); MPG_WG(1, 2); MPG_WG(1, 4); MPG_WG(1, 8);
    MPG_WG(2, 1); MPG_WG(2, 2); MPG_WG(2, 4); MPG_WG(2, 8);
    MPG_WG(4, 1); MPG_WG(4, 2); MPG_WG(4, 4); MPG_WG(4, 8);
    MPG_WG(8, 1); MPG_WG(8, 2); MPG_WG(8, 4); MPG_WG(8, 8);
#undef MPG_WG
    MPG_LAUNCH_CHECK("wgrad_kernel");
}

extern "C" int mpg_conv2d_dgrad(mpg_stream_t stream, const float* dy, int n, int h, int w, int cin,
                                const float* w_hwoi, int cout, int kh, int kw, int stride_h, int stride_w,
                                float wscale, float* dx) {
    MPG_REQUIRE(dy && w_hwoi && dx, "mpg_conv2d_dgrad: null pointer");
    MPG_GEOM_CHECK("mpg_conv2d_dgrad");
    ConvGeom g;
    fill_geom(g, n, h, w, cin, cout, kh, kw, stride_h, stride_w);
    const size_t total = (size_t)n * h * w * cin;
    hipLaunchKernelGGL(dgrad_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, dy, w_hwoi, dx, g,
                       wscale);
    MPG_LAUNCH_CHECK("dgrad_kernel");
}

extern "C" int mpg_fc_forward(mpg_stream_t stream, const float* x, int rows, int k, const float* w, int cout,
                              float wscale, const float* bias, int act, float leak, float* y) {
    MPG_REQUIRE(x && w && y, "mpg_fc_forward: null pointer");
    MPG_REQUIRE(rows >= 1 && k >= 1 && cout >= 1, "mpg_fc_forward: bad shape");
    MPG_REQUIRE(act >= MPG_ACT_NONE && act <= MPG_ACT_TANH, "mpg_fc_forward: bad activation %d", act);
    const int ogroups = (cout + 63) / 64;
    if (k >= 16384 && (size_t)rows * ogroups < 512) {
        int splits = (int)(1024 / ((size_t)rows * ogroups));
        if (splits > k / 2048) splits = k / 2048;                       // at least 2048 terms per block
        if (splits > 1) {
            const int kchunk = ((k + splits - 1) / splits + 255) / 256 * 256;
            splits = (k + kchunk - 1) / kchunk;
            hipError_t e = mpg::zero_async(y, (size_t)rows * cout * sizeof(float), (hipStream_t)stream);
            if (e != hipSuccess) return mpg::hip_check(e, "mpg_fc_forward: zero");
            hipLaunchKernelGGL(fc_splitk_kernel, dim3(rows, ogroups, splits), dim3(BLK), 0, (hipStream_t)stream, x, w, k, cout,
                               kchunk, wscale, y);
            hipLaunchKernelGGL(fc_finish_kernel, dim3(grid_for((size_t)rows * cout)), dim3(BLK), 0, (hipStream_t)stream, y, bias,
                               (size_t)rows * cout, cout, act, leak);
            MPG_LAUNCH_CHECK("fc_splitk_kernel");
        }
    }
    hipLaunchKernelGGL(fc_fwd_kernel, dim3(rows, ogroups), dim3(BLK), 0, (hipStream_t)stream, x, w, bias, k,
                       cout, wscale, act, leak, y);
    MPG_LAUNCH_CHECK("fc_fwd_kernel");
}

extern "C" int mpg_channel_sum(mpg_stream_t stream, const float* x, size_t npix, int c, float* out) {
    MPG_REQUIRE(x && out, "mpg_channel_sum: null pointer");
    MPG_REQUIRE(npix >= 1 && c >= 1, "mpg_channel_sum: bad shape");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = mpg::zero_async(out, (size_t)c * sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_channel_sum: memset");
    launch_chan_sum<0>(s, x, nullptr, npix, c, nullptr, nullptr, 0.f, 0.f, out, nullptr);
    MPG_LAUNCH_CHECK("chan_sum_kernel");
}

// ... with the blocks' sums kept in `partials` (mpg_bn_partials_floats(c) floats; the second float of a pair receives a
// copy) and added in a fixed order
extern "C" int mpg_channel_sum_ordered(mpg_stream_t stream, const float* x, size_t npix, int c, float* out, float* partials,
                                       size_t partials_floats) {
    MPG_REQUIRE(x && out && partials, "mpg_channel_sum_ordered: null pointer");
    MPG_REQUIRE(npix >= 1 && c >= 1, "mpg_channel_sum_ordered: bad shape");
    MPG_REQUIRE(partials_floats >= (size_t)CHAN_SUM_MAX_BLOCKS * c * 2 + (size_t)c, "mpg_channel_sum_ordered: partials buffer too small");
    hipStream_t s = (hipStream_t)stream;
    float* spare = partials + (size_t)CHAN_SUM_MAX_BLOCKS * c * 2;      // sum_partials_kernel writes two vectors: the second goes here
    const int nb = launch_chan_sum<0>(s, x, nullptr, npix, c, nullptr, nullptr, 0.f, 0.f, out, nullptr, partials);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((c + 15) / 16), dim3(256), 0, s, (const float2*)partials, nb, c, out, spare);
    MPG_LAUNCH_CHECK("chan_sum_kernel (ordered)");
}

extern "C" size_t mpg_bn_partials_floats(int c) { return c >= 1 ? (size_t)CHAN_SUM_MAX_BLOCKS * c * 2 : 0; }

static int bn_train_fwd_impl(mpg_stream_t stream, const float* x, size_t npix, int c, const float* gamma, const float* beta,
                             float eps, int act, float leak, float* y, float* batch_mean, float* batch_var, float* moving_mean,
                             float* moving_var, float decay, float* partials);

extern "C" int mpg_bn_train_fwd(mpg_stream_t stream, const float* x, size_t npix, int c, const float* gamma,
                                const float* beta, float eps, int act, float leak, float* y, float* batch_mean,
                                float* batch_var, float* moving_mean, float* moving_var, float decay) {
    return bn_train_fwd_impl(stream, x, npix, c, gamma, beta, eps, act, leak, y, batch_mean, batch_var, moving_mean, moving_var,
                             decay, nullptr);
}

extern "C" int mpg_bn_train_fwd_ordered(mpg_stream_t stream, const float* x, size_t npix, int c, const float* gamma,
                                        const float* beta, float eps, int act, float leak, float* y, float* batch_mean,
                                        float* batch_var, float* moving_mean, float* moving_var, float decay, float* partials,
                                        size_t partials_floats) {
    MPG_REQUIRE(partials != nullptr && partials_floats >= mpg_bn_partials_floats(c), "mpg_bn_train_fwd_ordered: partials buffer too small");
    return bn_train_fwd_impl(stream, x, npix, c, gamma, beta, eps, act, leak, y, batch_mean, batch_var, moving_mean, moving_var,
                             decay, partials);
}

static int bn_train_fwd_impl(mpg_stream_t stream, const float* x, size_t npix, int c, const float* gamma, const float* beta,
                             float eps, int act, float leak, float* y, float* batch_mean, float* batch_var, float* moving_mean,
                             float* moving_var, float decay, float* partials) {
    MPG_REQUIRE(x && gamma && beta && y && batch_mean && batch_var, "mpg_bn_train_fwd: null pointer");
    MPG_REQUIRE(npix >= 1 && c >= 1, "mpg_bn_train_fwd: bad shape");
    MPG_REQUIRE(act >= MPG_ACT_NONE && act <= MPG_ACT_TANH, "mpg_bn_train_fwd: bad activation %d", act);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (partials != nullptr) {
        // ordered form: sum_partials_kernel writes every sum, nothing to clear
    } else if (batch_var == batch_mean + c) {
        e = mpg::zero_async(batch_mean, (size_t)2 * c * sizeof(float), s);
    } else {
        e = mpg::zero_async(batch_mean, (size_t)c * sizeof(float), s);
        if (e == hipSuccess) e = mpg::zero_async(batch_var, (size_t)c * sizeof(float), s);
    }
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_bn_train_fwd: memset");
    const float inv_n = 1.f / (float)npix;
#if MPG_BN_TWO_PASS
    launch_chan_sum<0>(s, x, nullptr, npix, c, nullptr, nullptr, 0.f, 0.f, batch_mean, nullptr);
    launch_chan_sum<1>(s, x, nullptr, npix, c, batch_mean, nullptr, inv_n, 0.f, batch_var, nullptr);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(grid_for(c)), dim3(BLK), 0, s, batch_mean, batch_var, c, inv_n,
                       moving_mean, moving_var, decay, (const float*)nullptr, (size_t)0);
#else
    const int nb = launch_chan_sum<3>(s, x, nullptr, npix, c, nullptr, nullptr, 0.f, 0.f, batch_mean, batch_var, partials);
    if (partials != nullptr)
        hipLaunchKernelGGL(sum_partials_kernel, dim3((c + 15) / 16), dim3(256), 0, s, (const float2*)partials, nb, c, batch_mean,
                           batch_var);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(grid_for(c)), dim3(BLK), 0, s, batch_mean, batch_var, c, inv_n,
                       moving_mean, moving_var, decay, x, npix);
#endif
    const size_t total = npix * c;
    const uintptr_t al = (uintptr_t)x | (uintptr_t)y;
    if ((c % 4) == 0 && c <= BN4_CMAX && (al & 15) == 0) {
        unsigned g = grid_for(total / 4);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(bn_apply4_kernel, dim3(g), dim3(BLK), 0, s, x, total / 4, c, batch_mean, batch_var,
                           gamma, beta, eps, act, leak, y);
    }
    else
        hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total)), dim3(BLK), 0, s, x, total, c, batch_mean, batch_var,
                           gamma, beta, eps, act, leak, y);
    MPG_LAUNCH_CHECK("bn_train_fwd");
}

static int bn_train_bwd_impl(mpg_stream_t stream, const float* dy, const float* x, size_t npix, int c, const float* batch_mean,
                             const float* batch_var, const float* gamma, float eps, float* dx, float* dgamma, float* dbeta,
                             float* amax, float* partials);

extern "C" int mpg_bn_train_bwd(mpg_stream_t stream, const float* dy, const float* x, size_t npix, int c,
                                const float* batch_mean, const float* batch_var, const float* gamma, float eps,
                                float* dx, float* dgamma, float* dbeta, float* amax) {
    return bn_train_bwd_impl(stream, dy, x, npix, c, batch_mean, batch_var, gamma, eps, dx, dgamma, dbeta, amax, nullptr);
}

extern "C" int mpg_bn_train_bwd_ordered(mpg_stream_t stream, const float* dy, const float* x, size_t npix, int c,
                                        const float* batch_mean, const float* batch_var, const float* gamma, float eps,
                                        float* dx, float* dgamma, float* dbeta, float* amax, float* partials,
                                        size_t partials_floats) {
    MPG_REQUIRE(partials != nullptr && partials_floats >= mpg_bn_partials_floats(c), "mpg_bn_train_bwd_ordered: partials buffer too small");
    return bn_train_bwd_impl(stream, dy, x, npix, c, batch_mean, batch_var, gamma, eps, dx, dgamma, dbeta, amax, partials);
}

static int bn_train_bwd_impl(mpg_stream_t stream, const float* dy, const float* x, size_t npix, int c, const float* batch_mean,
                             const float* batch_var, const float* gamma, float eps, float* dx, float* dgamma, float* dbeta,
                             float* amax, float* partials) {
    MPG_REQUIRE(dy && x && batch_mean && batch_var && gamma && dx && dgamma && dbeta,
                "mpg_bn_train_bwd: null pointer");
    MPG_REQUIRE(npix >= 1 && c >= 1, "mpg_bn_train_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (partials != nullptr) {
        // ordered form: sum_partials_kernel writes every sum, nothing to clear
    } else if (dbeta == dgamma + c) {
        e = mpg::zero_async(dgamma, (size_t)2 * c * sizeof(float), s);
    } else {
        e = mpg::zero_async(dgamma, (size_t)c * sizeof(float), s);
        if (e == hipSuccess) e = mpg::zero_async(dbeta, (size_t)c * sizeof(float), s);
    }
    if (e == hipSuccess && amax != nullptr) e = mpg::zero_async(amax, sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_bn_train_bwd: memset");
    const int nb = launch_chan_sum<2>(s, dy, x, npix, c, batch_mean, batch_var, 0.f, eps, dbeta, dgamma, partials);
    if (partials != nullptr)      // the blocks' sums in a fixed order (and no atomics queueing on 2 c addresses)
        hipLaunchKernelGGL(sum_partials_kernel, dim3((c + 15) / 16), dim3(256), 0, s, (const float2*)partials, nb, c, dbeta, dgamma);
    const size_t total = npix * c;
    unsigned g = grid_for(total);
    if (amax != nullptr && g > AMAX_GRID) g = AMAX_GRID;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(g), dim3(BLK), 0, s, dy, x, total, c, batch_mean,
                       batch_var, gamma, dgamma, dbeta, eps, 1.f / (float)npix, dx, (unsigned int*)amax);
    MPG_LAUNCH_CHECK("bn_train_bwd");
}

extern "C" int mpg_act_bwd(mpg_stream_t stream, const float* dy, const float* y, size_t n, int act, float leak,
                           float* dx, float* amax) {
    MPG_REQUIRE(dy && y && dx, "mpg_act_bwd: null pointer");
    MPG_REQUIRE(act >= MPG_ACT_NONE && act <= MPG_ACT_TANH, "mpg_act_bwd: bad activation %d", act);
    if (amax != nullptr) {
        hipError_t e = mpg::zero_async(amax, sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return mpg::hip_check(e, "mpg_act_bwd: zero");
    }
    if (n == 0) return MPG_OK;
    unsigned g = grid_for((n + 3) / 4);
    if (g > AMAX_GRID) g = AMAX_GRID;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(g), dim3(BLK), 0, (hipStream_t)stream, dy, y, n, act, leak, dx, (unsigned int*)amax);
    MPG_LAUNCH_CHECK("act_bwd_kernel");
}

extern "C" int mpg_pixel_norm_bwd(mpg_stream_t stream, const float* dy, const float* x, size_t npix, int c, float eps,
                                  float* dx) {
    MPG_REQUIRE(dy && x && dx, "mpg_pixel_norm_bwd: null pointer");
    MPG_REQUIRE(npix >= 1 && c >= 1, "mpg_pixel_norm_bwd: bad shape");
    int lanes = 1;
    while (lanes * 2 <= c && lanes < 64) lanes <<= 1;
    hipLaunchKernelGGL(pixel_norm_bwd_kernel, dim3(grid_for(npix * lanes)), dim3(BLK), 0, (hipStream_t)stream, dy, x, npix,
                       c, lanes, eps, dx);
    MPG_LAUNCH_CHECK("pixel_norm_bwd_kernel");
}

extern "C" int mpg_resize_nearest_bwd(mpg_stream_t stream, const float* dy, int n, int oh, int ow, int c, float* dx,
                                      int h, int w) {
    MPG_REQUIRE(dy && dx, "mpg_resize_nearest_bwd: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1 && oh >= h && ow >= w, "mpg_resize_nearest_bwd: bad shape");
    MPG_REQUIRE(oh % h == 0 && ow % w == 0, "mpg_resize_nearest_bwd: only integer factors (%dx%d -> %dx%d)", h, w, oh, ow);
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(resize_nearest_bwd_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, dy, n, oh,
                       ow, c, dx, h, w, 1.f);
    MPG_LAUNCH_CHECK("resize_nearest_bwd_kernel");
}

extern "C" int mpg_avg_pool2_bwd(mpg_stream_t stream, const float* dy, int n, int h, int w, int c, float* dx) {
    MPG_REQUIRE(dy && dx, "mpg_avg_pool2_bwd: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 2 && w >= 2 && c >= 1, "mpg_avg_pool2_bwd: bad shape");
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(avg_pool2_bwd_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, dy, n, h, w, c,
                       dx);
    MPG_LAUNCH_CHECK("avg_pool2_bwd_kernel");
}

extern "C" int mpg_lerp(mpg_stream_t stream, const float* x, const float* y, size_t n, float t, float* out) {
    MPG_REQUIRE(y && out, "mpg_lerp: null pointer");
    if (n == 0) return MPG_OK;
    hipLaunchKernelGGL(lerp_kernel, dim3(grid_for(n)), dim3(BLK), 0, (hipStream_t)stream, x, y, n, t, out);
    MPG_LAUNCH_CHECK("lerp_kernel");
}

extern "C" int mpg_tensor_resample(mpg_stream_t stream, const float* value, const float* pos, int n, int h, int w,
                                   int c, int clamp, float* out) {
    MPG_REQUIRE(value && pos && out, "mpg_tensor_resample: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "mpg_tensor_resample: bad shape");
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(resample_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, value, pos, n, h, w, c,
                       clamp, out);
    MPG_LAUNCH_CHECK("resample_kernel");
}

extern "C" int mpg_tensor_resample_bwd(mpg_stream_t stream, const float* dy, const float* pos, int n, int h, int w,
                                       int c, int clamp, float* dvalue) {
    MPG_REQUIRE(dy && pos && dvalue, "mpg_tensor_resample_bwd: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "mpg_tensor_resample_bwd: bad shape");
    const size_t total = (size_t)n * h * w * c;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = mpg::zero_async(dvalue, total * sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_tensor_resample_bwd: zero");
    hipLaunchKernelGGL(resample_bwd_kernel, dim3(grid_for(total)), dim3(BLK), 0, s, dy, pos, n, h, w, c, clamp, dvalue);
    MPG_LAUNCH_CHECK("resample_bwd_kernel");
}

extern "C" int mpg_pair_reduce(mpg_stream_t stream, const float* a, const float* b, size_t n, int mode, float* out) {
    MPG_REQUIRE(a && out, "mpg_pair_reduce: null pointer");
    MPG_REQUIRE(mode == 0 || mode == 1, "mpg_pair_reduce: mode %d", mode);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = mpg::zero_async(out, sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_pair_reduce: zero");
    if (n == 0) return MPG_OK;
    size_t blocks = (n + BLK * 8 - 1) / (BLK * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pair_reduce_kernel, dim3((unsigned)blocks), dim3(BLK), 0, s, a, b, n, mode, out);
    MPG_LAUNCH_CHECK("pair_reduce_kernel");
}

extern "C" int mpg_adam_step(mpg_stream_t stream, float* p, const float* grad, float* m, float* v, size_t n,
                             const float* lr_t, float beta1, float beta2, float eps) {
    MPG_REQUIRE(p && grad && m && v && lr_t, "mpg_adam_step: null pointer");
    if (n == 0) return MPG_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(BLK), 0, (hipStream_t)stream, p, grad, m, v, n, lr_t,
                       beta1, beta2, eps);
    MPG_LAUNCH_CHECK("adam_kernel");
}

extern "C" int mpg_adam_step_staged(mpg_stream_t stream, float* p, const float* grad, float* m, float* v, const float* mask,
                                    size_t n, float* state, const float* lr, int total_grads, int use_loss_scaling,
                                    float beta1, float beta2, float eps, float ls_inc, float ls_dec, float* ema_shadow,
                                    float ema_decay) {
    MPG_REQUIRE(p && grad && m && v && state && lr, "mpg_adam_step_staged: null pointer");
    MPG_REQUIRE(total_grads >= 1, "mpg_adam_step_staged: total_grads %d", total_grads);
    if (n == 0) return MPG_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ls_begin_kernel, dim3(1), dim3(64), 0, s, state, lr, 1.f / (float)total_grads, use_loss_scaling, beta1, beta2);
    if (use_loss_scaling) hipLaunchKernelGGL(ls_check_kernel, dim3(grid_for(n)), dim3(BLK), 0, s, grad, mask, n, state);
    hipLaunchKernelGGL(adam_staged_kernel, dim3(grid_for(n)), dim3(BLK), 0, s, p, grad, m, v, mask, n, state, beta1, beta2, eps,
                       ema_shadow, ema_decay);
    hipLaunchKernelGGL(ls_end_kernel, dim3(1), dim3(64), 0, s, state, use_loss_scaling, ls_inc, ls_dec);
    MPG_LAUNCH_CHECK("adam_staged_kernel");
}

extern "C" int mpg_advect_velocity(mpg_stream_t stream, const float* vel, int n, int hv, int wv, int cv, int h, int w, float dt,
                                   float* out) {
    MPG_REQUIRE(vel && out, "mpg_advect_velocity: null pointer");
    MPG_REQUIRE(n >= 1 && hv >= 1 && wv >= 1 && cv >= 2 && h >= 1 && w >= 1, "mpg_advect_velocity: bad shape");
    MPG_REQUIRE(n % 3 == 0, "mpg_advect_velocity: the batch (%d) must hold whole frame triples", n);
    const size_t total = (size_t)n * h * w;
    hipLaunchKernelGGL(advect_velocity_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, vel, n, hv, wv, cv, h, w,
                       dt, out);
    MPG_LAUNCH_CHECK("advect_velocity_kernel");
}

extern "C" int mpg_semi_lagrange(mpg_stream_t stream, const float* source, const float* vel, int n, int h, int w, int c,
                                 float vel_sign, float* out) {
    MPG_REQUIRE(source && vel && out, "mpg_semi_lagrange: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "mpg_semi_lagrange: bad shape");
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(semi_lagrange_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, source, vel, n, h, w, c,
                       vel_sign, out);
    MPG_LAUNCH_CHECK("semi_lagrange_kernel");
}

extern "C" int mpg_semi_lagrange_bwd(mpg_stream_t stream, const float* dy, const float* vel, int n, int h, int w, int c,
                                     float vel_sign, float* dsource) {
    MPG_REQUIRE(dy && vel && dsource, "mpg_semi_lagrange_bwd: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "mpg_semi_lagrange_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)n * h * w * c;
    hipError_t e = mpg::zero_async(dsource, total * sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_semi_lagrange_bwd: zero");
    hipLaunchKernelGGL(semi_lagrange_bwd_kernel, dim3(grid_for(total)), dim3(BLK), 0, s, dy, vel, n, h, w, c, vel_sign, dsource);
    MPG_LAUNCH_CHECK("semi_lagrange_bwd_kernel");
}

extern "C" int mpg_maccormack(mpg_stream_t stream, const float* source, const float* forward, const float* backward,
                              const float* flags, const float* vel, int n, int h, int w, float strength, float* out,
                              float* keep) {
    MPG_REQUIRE(source && forward && backward && flags && vel && out, "mpg_maccormack: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1, "mpg_maccormack: bad shape");
    const size_t total = (size_t)n * h * w;
    hipLaunchKernelGGL(maccormack_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, source, forward, backward,
                       flags, vel, n, h, w, strength, out, keep);
    MPG_LAUNCH_CHECK("maccormack_kernel");
}
