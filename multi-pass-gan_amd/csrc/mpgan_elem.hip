// HBM-bound companions of the convolution: legacy-TF1 resampling, pooling,
// pixel norm, the axis zoom and the volume <-> slice-batch transposes, plus a
// plain fp32 direct convolution on the vector ALUs (strided / odd shapes and an
// independent check of the MFMA kernel).  One thread per output element with
// the channel (or x) index fastest, so wave accesses are contiguous.
#include "mpgan_internal.h"

namespace {

constexpr int BLK = 256;

inline unsigned grid_for(size_t n) { return (unsigned)((n + BLK - 1) / BLK); }

// ---------------------------------------------------------------- direct conv (tf.nn.conv2d SAME, GAN.py:686-691)
__global__ void conv_direct_kernel(const float* __restrict__ x, int n, int h, int w, int cin,
                                   const float* __restrict__ wt, int kh, int kw, int cout, int sh, int sw,
                                   int oh, int ow, int pt, int pl, float wscale,
                                   const float* __restrict__ cscale, const float* __restrict__ bias,
                                   int act, float leak, float* __restrict__ y) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * oh * ow * cout;
    if (idx >= total) return;
    const int co = idx % cout;
    size_t pix = idx / cout;
    const int ox = pix % ow; pix /= ow;
    const int oy = pix % oh;
    const int b = pix / oh;
    float acc = 0.f;
    for (int dy = 0; dy < kh; ++dy) {
        const int iy = oy * sh - pt + dy;
        if (iy < 0 || iy >= h) continue;
        for (int dx = 0; dx < kw; ++dx) {
            const int ix = ox * sw - pl + dx;
            if (ix < 0 || ix >= w) continue;
            const float* xp = x + (((size_t)b * h + iy) * w + ix) * cin;
            const float* wp = wt + ((size_t)(dy * kw + dx) * cin) * cout + co;
            for (int ci = 0; ci < cin; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * cout], acc);
        }
    }
    float v = acc * wscale;
    if (cscale) v *= cscale[co];
    if (bias) v += bias[co];
    y[idx] = mpg::apply_act(v, act, leak);
}

// ---------------------------------------------------------------- resize (legacy TF1 coordinates)
__global__ void resize_nearest_kernel(const float* __restrict__ x, int n, int h, int w, int c,
                                      float* __restrict__ y, int oh, int ow, float sy, float sx) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * oh * ow * c;
    if (idx >= total) return;
    const int ch = idx % c;
    size_t p = idx / c;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    const int iy = min((int)floorf(oy * sy), h - 1);
    const int ix = min((int)floorf(ox * sx), w - 1);
    y[idx] = x[(((size_t)b * h + iy) * w + ix) * c + ch];
}

__global__ void resize_bilinear_kernel(const float* __restrict__ x, int n, int h, int w, int c,
                                       float* __restrict__ y, int oh, int ow, float sy, float sx) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * oh * ow * c;
    if (idx >= total) return;
    const int ch = idx % c;
    size_t p = idx / c;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    const float fy = oy * sy, fx = ox * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - y0, lx = fx - x0;
    const float* base = x + (size_t)b * h * w * c + ch;
    const float tl = base[((size_t)y0 * w + x0) * c], tr = base[((size_t)y0 * w + x1) * c];
    const float bl = base[((size_t)y1 * w + x0) * c], br = base[((size_t)y1 * w + x1) * c];
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    y[idx] = top + (bot - top) * ly;
}

// TF 1.x ResizeBicubic: Keys A = -0.75, fraction quantised to 1/1024, taps clamped.
__device__ __forceinline__ void bicubic_taps(int o, float scale, int in_size, int idx[4], float wt[4]) {
    const float A = -0.75f;
    const float s = o * scale;
    const int i = (int)floorf(s);
    const float delta = s - (float)i;
    const int off = (int)lrintf(delta * 1024.f);
    const float x0 = (float)off / 1024.f;
    const float x1 = (float)(1024 - off) / 1024.f;
    const float xa = x0 + 1.f, xb = x1 + 1.f;
    wt[0] = ((A * xa - 5.f * A) * xa + 8.f * A) * xa - 4.f * A;
    wt[1] = ((A + 2.f) * x0 - (A + 3.f)) * x0 * x0 + 1.f;
    wt[2] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    wt[3] = ((A * xb - 5.f * A) * xb + 8.f * A) * xb - 4.f * A;
#pragma unroll
    for (int t = 0; t < 4; ++t) idx[t] = min(max(i - 1 + t, 0), in_size - 1);
}

__global__ void resize_bicubic_kernel(const float* __restrict__ x, int n, int h, int w, int c,
                                      float* __restrict__ y, int oh, int ow, float sy, float sx) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * oh * ow * c;
    if (idx >= total) return;
    const int ch = idx % c;
    size_t p = idx / c;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    int iy[4], ix[4];
    float wy[4], wx[4];
    bicubic_taps(oy, sy, h, iy, wy);
    bicubic_taps(ox, sx, w, ix, wx);
    const float* base = x + (size_t)b * h * w * c + ch;
    float out = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float row = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) row += base[((size_t)iy[r] * w + ix[t]) * c] * wx[t];
        out += row * wy[r];
    }
    y[idx] = out;
}

__global__ void avg_pool2_kernel(const float* __restrict__ x, int n, int h, int w, int c, float* __restrict__ y) {
    const int oh = h / 2, ow = w / 2;
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * oh * ow * c;
    if (idx >= total) return;
    const int ch = idx % c;
    size_t p = idx / c;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    const float* base = x + (((size_t)b * h + 2 * oy) * w + 2 * ox) * c + ch;
    y[idx] = (base[0] + base[c] + base[(size_t)w * c] + base[(size_t)w * c + c]) * 0.25f;
}

// tf.nn.max_pool VALID, window k x k, stride s.  arg (optional) receives the window position (dy * k + dx) of the
// FIRST maximum in scan order: where TensorFlow's MaxPoolGrad sends the gradient.
__global__ void max_pool_kernel(const float* __restrict__ x, int n, int h, int w, int c, int k, int s,
                                float* __restrict__ y, unsigned char* __restrict__ arg) {
    const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * oh * ow * c;
    if (idx >= total) return;
    const int ch = idx % c;
    size_t p = idx / c;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    const float* base = x + (((size_t)b * h + (size_t)s * oy) * w + (size_t)s * ox) * c + ch;
    float best = base[0];
    int at = 0;
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) {
            const float v = base[((size_t)dy * w + dx) * c];
            if (v > best) { best = v; at = dy * k + dx; }
        }
    y[idx] = best;
    if (arg != nullptr) arg[idx] = (unsigned char)at;
}

// gradient: every output sends dy to its arg position (windows overlap when s < k: atomics; dx is zeroed first)
__global__ void max_pool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg, int n, int h, int w,
                                    int c, int k, int s, float* __restrict__ dx) {
    const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * oh * ow * c;
    if (idx >= total) return;
    const int ch = idx % c;
    size_t p = idx / c;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    const int at = arg[idx];
    float* dst = dx + (((size_t)b * h + (size_t)s * oy + at / k) * w + (size_t)s * ox + at % k) * c + ch;
    if (s >= k) *dst = dy[idx];
    else atomicAdd(dst, dy[idx]);
}

// `lanes` (a power of two <= 64) consecutive lanes share one pixel and stride its channels, so a wave reads
// contiguous memory; the sum of squares is folded with shuffles inside the lane group
__global__ void pixel_norm_kernel(const float* __restrict__ x, size_t npix, int c, int lanes, float eps,
                                  float* __restrict__ y) {
    const size_t gid = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t pix = gid / lanes;
    const int l = (int)(gid % lanes);
    const bool ok = pix < npix;
    const float* p = x + (ok ? pix : 0) * c;
    float ss = 0.f;
    if (ok)
        for (int i = l; i < c; i += lanes) ss = fmaf(p[i], p[i], ss);
    for (int m = lanes >> 1; m > 0; m >>= 1) ss += __shfl_xor(ss, m);
    if (!ok) return;
    const float sc = rsqrtf(ss / (float)c + eps);
    float* q = y + pix * c;
    for (int i = l; i < c; i += lanes) q[i] = p[i] * sc;
}

// GAN.minibatch_stddev_layer (GAN.py:476-488): batch index n = g * M + m (G groups members, M groups);
// stat[m] = mean over (h, w, c) of sqrt(var over g + 1e-8).  Pass 1 accumulates the per-m sums.
__global__ __launch_bounds__(256) void mbstd_stat_kernel(const float* __restrict__ x, int g, int m, size_t hwc,
                                                         float* __restrict__ stat) {
    __shared__ float red[BLK];
    const int mi = blockIdx.y;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < hwc; i += (size_t)gridDim.x * BLK) {
        float mean = 0.f;
        for (int k = 0; k < g; ++k) mean += x[((size_t)k * m + mi) * hwc + i];
        mean /= (float)g;
        float var = 0.f;
        for (int k = 0; k < g; ++k) {
            const float d = x[((size_t)k * m + mi) * hwc + i] - mean;
            var = fmaf(d, d, var);
        }
        s += sqrtf(var / (float)g + 1e-8f);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = BLK / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(stat + mi, red[0]);
}

// pass 2: y[n, p, 0..c-1] = x, y[n, p, c] = stat[n % m] / hwc
__global__ void mbstd_concat_kernel(const float* __restrict__ x, const float* __restrict__ stat, int m, size_t npix_per,
                                    int c, size_t total, float inv_hwc, float* __restrict__ y) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= total) return;
    const int ch = idx % (c + 1);
    const size_t pix = idx / (c + 1);
    const int n = (int)(pix / npix_per);
    y[idx] = ch < c ? x[pix * c + ch] : stat[n % m] * inv_hwc;
}

// backward.  dstat[m] = sum over the group's members and pixels of dy[.., c] (the broadcast statistic's gradient).
__global__ __launch_bounds__(256) void mbstd_bwd_stat_kernel(const float* __restrict__ dy, int n, int m, size_t npix_per, int c,
                                                             float* __restrict__ dstat) {
    __shared__ float red[BLK];
    const int mi = blockIdx.y;
    const int g = n / m;
    float s = 0.f;
    const size_t total = (size_t)g * npix_per;
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < total; i += (size_t)gridDim.x * BLK) {
        const size_t k = i / npix_per, p = i - k * npix_per;
        s += dy[(((size_t)k * m + mi) * npix_per + p) * (c + 1) + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = BLK / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(dstat + mi, red[0]);
}

// dx[k,i] = dy[k,i (channels < c)] + dstat[m] / hwc * (x[k,i] - mean_i) / (g * sqrt(var_i + 1e-8))
__global__ void mbstd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ dstat,
                                 int g, int m, size_t hwc, int c, float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * BLK + threadIdx.x;
    const int mi = blockIdx.y;
    if (i >= hwc) return;
    float mean = 0.f;
    for (int k = 0; k < g; ++k) mean += x[((size_t)k * m + mi) * hwc + i];
    mean /= (float)g;
    float var = 0.f;
    for (int k = 0; k < g; ++k) {
        const float d = x[((size_t)k * m + mi) * hwc + i] - mean;
        var = fmaf(d, d, var);
    }
    const float coef = dstat[mi] / ((float)hwc * (float)g * sqrtf(var / (float)g + 1e-8f));
    const size_t pix = i / c;
    const int ch = (int)(i - pix * c);
    for (int k = 0; k < g; ++k) {
        const size_t row = (size_t)k * m + mi;
        dx[row * hwc + i] = dy[(row * (hwc / c) + pix) * (c + 1) + ch] + coef * (x[row * hwc + i] - mean);
    }
}

__global__ void add_act_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, int act,
                               float leak, float* __restrict__ y) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= n) return;
    float v = a[idx];
    if (b) v += b[idx];
    y[idx] = mpg::apply_act(v, act, leak);
}

// ---------------------------------------------------------------- volume marshalling
__global__ void axis_zoom_kernel(const float* __restrict__ v, size_t outer, int n, size_t inner,
                                 float* __restrict__ out, int big) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = outer * (size_t)big * inner;
    if (idx >= total) return;
    const size_t in = idx % inner;
    size_t p = idx / inner;
    const int o = p % big;
    const size_t ou = p / big;
    // scipy.ndimage.zoom(order=1): src = o*(n-1)/(big-1), computed in double like scipy
    const double s = big > 1 ? (double)o * (double)(n - 1) / (double)(big - 1) : 0.0;
    int i0 = (int)floor(s);
    if (i0 > n - 1) i0 = n - 1;
    const int i1 = min(i0 + 1, n - 1);
    const double t = s - (double)i0;
    const double a = v[(ou * n + i0) * inner + in];
    const double b = v[(ou * n + i1) * inner + in];
    out[idx] = (float)(a * (1.0 - t) + b * t);
}

struct PermArgs {
    int din[3];      // input dims
    int perm[3];     // out axis k takes input axis perm[k]
    int cmap[8];
    int c;
    float cutoff;
};

// generic: one thread per output element (small, multi-channel volumes)
__global__ void transpose_generic_kernel(const float* __restrict__ v, PermArgs a, float* __restrict__ out) {
    const int dout[3] = {a.din[a.perm[0]], a.din[a.perm[1]], a.din[a.perm[2]]};
    const size_t total = (size_t)dout[0] * dout[1] * dout[2] * a.c;
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= total) return;
    const int ch = idx % a.c;
    size_t p = idx / a.c;
    int o[3];
    o[2] = p % dout[2]; p /= dout[2];
    o[1] = p % dout[1];
    o[0] = p / dout[1];
    int i[3];
    i[a.perm[0]] = o[0]; i[a.perm[1]] = o[1]; i[a.perm[2]] = o[2];
    float val = v[(((size_t)i[0] * a.din[1] + i[1]) * a.din[2] + i[2]) * a.c + a.cmap[ch]];
    if (a.cutoff > 0.f && val < a.cutoff) val = 0.f;
    out[idx] = val;
}

// single-channel volumes whose innermost axis moves: 32x32 LDS tile between the input's
// innermost axis b (=2) and the axis a that becomes innermost in the output (perm[2]); both
// the reads and the writes are contiguous 128-byte rows.  r is the remaining axis.
struct TileArgs {
    int da, db, dr;
    size_t sin_a, sin_r;            // input strides (axis b has stride 1)
    size_t sout_b, sout_r;          // output strides (axis a has stride 1)
    int tiles_a, tiles_b;
    float cutoff;
};

// 64 x 64 tile through LDS: a wave reads / writes whole 256-byte rows in both directions (two 128-byte lines per row
// instead of one: half as many separate DRAM bursts per byte as the 32 x 32 tile it replaces, which averaged 3.3 TB/s
// on the 256^3 permutations, profiles/r01), rows padded to 65 words (conflict-free column reads).  Tiles are walked in
// an XCD-interleaved order so that the blocks resident at one time spread over the strided axis.
#ifndef MPG_TR_TILE
#define MPG_TR_TILE 64
#endif
constexpr int TRT = MPG_TR_TILE;           // tile edge: 64 (256-byte rows both ways) or 128 (512-byte rows, 66 KB of LDS)

__global__ __launch_bounds__(256) void transpose_tiled_kernel(const float* __restrict__ v, TileArgs a, float* __restrict__ out) {
    extern __shared__ float tile_lds[];
    float (*tile)[TRT + 1] = reinterpret_cast<float (*)[TRT + 1]>(tile_lds);
    int bid = blockIdx.x;
    const int tb = bid % a.tiles_b; bid /= a.tiles_b;
    const int ta = bid % a.tiles_a;
    const int rr = bid / a.tiles_a;
    constexpr int RP = 256 / TRT;                              // tile rows per pass of the 256 threads
    const int tx = threadIdx.x % TRT, ty = threadIdx.x / TRT;
    const int a0 = ta * TRT, b0 = tb * TRT;
    const float* src = v + (size_t)rr * a.sin_r;
#pragma unroll 4
    for (int k = 0; k < TRT / RP; ++k) {
        const int ia = a0 + ty + RP * k, ib = b0 + tx;
        if (ia < a.da && ib < a.db) tile[ty + RP * k][tx] = __builtin_nontemporal_load(src + (size_t)ia * a.sin_a + ib);
    }
    __syncthreads();
    float* dst = out + (size_t)rr * a.sout_r;
#pragma unroll 4
    for (int k = 0; k < TRT / RP; ++k) {
        const int ib = b0 + ty + RP * k, ia = a0 + tx;
        if (ia < a.da && ib < a.db) {
            float val = tile[tx][ty + RP * k];
            if (a.cutoff > 0.f && val < a.cutoff) val = 0.f;
            __builtin_nontemporal_store(val, dst + (size_t)ib * a.sout_b + ia);
        }
    }
}

// The same tile with 16-byte accesses in BOTH directions (round 3): 4-byte-per-lane loads and stores reach ~4 TB/s on
// this chip (profiles/r02/hbm_kernels.md), 16-byte ones the copy rate.  Load: thread (row a, quad b4) reads a float4
// along the input's innermost axis b (a wave instruction = four 256-byte rows) and stores it to LDS at [a][b4 ^ (a >> 2)].
// Store: thread (quad ag of a, quad bq of b) reads the four float4 tile[4 ag + i][bq], transposes the 4 x 4 block in
// registers and writes four float4 along the output's innermost axis a (again four 256-byte rows per wave instruction).
// The XOR swizzle makes both the LDS writes (16 lanes of one row: 16 distinct quads) and the LDS reads (16 lanes with
// ag = 0..15 and one bq: quads bq ^ ag, all distinct) conflict-free without padding.  Needs every extent and stride
// to be a multiple of 4 elements and 16-byte aligned pointers; the scalar kernel above takes the rest.
typedef float f4t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void transpose_tiled4_kernel(const float* __restrict__ v, TileArgs a, float* __restrict__ out) {
    __shared__ f4t tile4[64][16];
    int bid = blockIdx.x;
    const int tb = bid % a.tiles_b; bid /= a.tiles_b;
    const int ta = bid % a.tiles_a;
    const int rr = bid / a.tiles_a;
    const int a0 = ta * 64, b0 = tb * 64;
    const float* src = v + (size_t)rr * a.sin_r;
    {
        const int b4 = threadIdx.x & 15, ar = threadIdx.x >> 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int la = ar + 16 * k;
            const int ia = a0 + la, ib = b0 + 4 * b4;
            f4t x = {0.f, 0.f, 0.f, 0.f};
            if (ia < a.da && ib < a.db) x = __builtin_nontemporal_load(reinterpret_cast<const f4t*>(src + (size_t)ia * a.sin_a + ib));
            tile4[la][b4 ^ (la >> 2)] = x;
        }
    }
    __syncthreads();
    float* dst = out + (size_t)rr * a.sout_r;
    const int ag = threadIdx.x & 15, bq = threadIdx.x >> 4;
    f4t r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = tile4[4 * ag + i][bq ^ ag];
    const int ia = a0 + 4 * ag;
    if (ia < a.da) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ib = b0 + 4 * bq + j;
            if (ib < a.db) {
                f4t o = {r[0][j], r[1][j], r[2][j], r[3][j]};
                if (a.cutoff > 0.f) {
                    o.x = o.x < a.cutoff ? 0.f : o.x; o.y = o.y < a.cutoff ? 0.f : o.y;
                    o.z = o.z < a.cutoff ? 0.f : o.z; o.w = o.w < a.cutoff ? 0.f : o.w;
                }
                __builtin_nontemporal_store(o, reinterpret_cast<f4t*>(dst + (size_t)ib * a.sout_b + ia));
            }
        }
    }
}

__global__ void add_adjacent_kernel(const float* __restrict__ in, int s_total, size_t hw, int c, int s_off,
                                    int s_cnt, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)s_cnt * hw * (c + 2);
    if (idx >= total) return;
    const int ch = idx % (c + 2);
    size_t p = idx / (c + 2);
    const size_t px = p % hw;
    const int s = (int)(p / hw) + s_off;
    float v;
    if (ch < c) v = in[((size_t)s * hw + px) * c + ch];
    else if (ch == c) v = s > 0 ? in[((size_t)(s - 1) * hw + px) * c] : 0.f;
    else v = s + 1 < s_total ? in[((size_t)(s + 1) * hw + px) * c] : 0.f;
    out[idx] = v;
}

__global__ void cutoff_kernel(const float* __restrict__ v, size_t n, float cutoff, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= n) return;
    const float x = v[idx];
    out[idx] = x < cutoff ? 0.f : x;
}

typedef float f4v __attribute__((ext_vector_type(4)));

// 16 bytes per lane, streamed once (n4 = n / 4; the tail goes through cutoff_kernel)
__global__ void cutoff4_kernel(const f4v* __restrict__ v, size_t n4, float cutoff, f4v* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (idx >= n4) return;
    f4v x = __builtin_nontemporal_load(v + idx);
    x.x = x.x < cutoff ? 0.f : x.x; x.y = x.y < cutoff ? 0.f : x.y;
    x.z = x.z < cutoff ? 0.f : x.z; x.w = x.w < cutoff ? 0.f : x.w;
    __builtin_nontemporal_store(x, out + idx);
}

// transposes that keep the last axis (perm (1,0,2)) are row copies: out[i1][i0][:] = in[i0][i1][:], 16 bytes per lane
__global__ void swap01_rows_kernel(const f4v* __restrict__ v, int d0, int d1, int row4, float cutoff, f4v* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)d0 * d1 * row4;
    if (idx >= total) return;
    const int q = idx % row4;
    const size_t r = idx / row4;           // output row = i1 * d0 + i0
    const int i0 = r % d0;
    const int i1 = r / d0;
    f4v x = __builtin_nontemporal_load(v + ((size_t)i0 * d1 + i1) * row4 + q);
    if (cutoff > 0.f) {
        x.x = x.x < cutoff ? 0.f : x.x; x.y = x.y < cutoff ? 0.f : x.y;
        x.z = x.z < cutoff ? 0.f : x.z; x.w = x.w < cutoff ? 0.f : x.w;
    }
    __builtin_nontemporal_store(x, out + idx);
}


// tf.nn.conv2d_transpose(x, W[kh,kw,cout,cin], output_shape [n, h*sh, w*sw, cout], strides, "SAME") (GAN.py:703-708):
// the gradient of the SAME convolution that maps the OUTPUT grid to the input grid, written as a gather.  Forward
// geometry on the output grid: out = ceil(H/s), pad_total = max((out-1) s + k - H, 0), pad_before = pad_total / 2;
// y[oy] = sum over (iy, ky) with iy*s + ky - pad_before == oy of x[iy] * W[ky].  One thread per output element,
// cout fastest: the lanes of a pixel share the x reads (broadcast), W rows are contiguous over cin.
__global__ void conv_transpose_kernel(const float* __restrict__ x, int n, int h, int w, int cin,
                                      const float* __restrict__ wt, int kh, int kw, int cout, int sh, int sw,
                                      int pad_t, int pad_l, float wscale, const float* __restrict__ bias, int act,
                                      float leak, float* __restrict__ y) {
    const int oh = h * sh, ow = w * sw;
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)n * oh * ow * cout;
    if (idx >= total) return;
    const int co = idx % cout;
    size_t p = idx / cout;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    float acc = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
        const int ty = oy + pad_t - ky;
        if (ty < 0 || ty % sh) continue;
        const int iy = ty / sh;
        if (iy >= h) continue;
        for (int kx = 0; kx < kw; ++kx) {
            const int tx = ox + pad_l - kx;
            if (tx < 0 || tx % sw) continue;
            const int ix = tx / sw;
            if (ix >= w) continue;
            const float* xr = x + (((size_t)b * h + iy) * w + ix) * cin;
            const float* wr = wt + (((size_t)ky * kw + kx) * cout + co) * cin;
            for (int ci = 0; ci < cin; ++ci) acc = fmaf(xr[ci], wr[ci], acc);
        }
    }
    acc = acc * wscale + (bias != nullptr ? bias[co] : 0.f);
    y[idx] = mpg::apply_act(acc, act, leak);
}

// tf.depth_to_space(x, r) (GAN.pixel_shuffle, GAN.py:554-560): y[b, r*q + i, r*p + j, c] = x[b, q, p, (i*r + j)*C + c]
__global__ void depth_to_space_kernel(const float* __restrict__ x, int n, int h, int w, int c_out, int r,
                                      float* __restrict__ y) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const int oh = h * r, ow = w * r;
    const size_t total = (size_t)n * oh * ow * c_out;
    if (idx >= total) return;
    const int c = idx % c_out;
    size_t p = idx / c_out;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    const int q = oy / r, i = oy - q * r, pp = ox / r, j = ox - pp * r;
    y[idx] = x[(((size_t)b * h + q) * w + pp) * ((size_t)c_out * r * r) + (size_t)(i * r + j) * c_out + c];
}

}  // namespace

extern "C" int mpg_conv2d_direct(mpg_stream_t stream, const float* x, int n, int h, int w, int cin,
                                 const float* w_hwio, int kh, int kw, int cout, int stride_h, int stride_w,
                                 float wscale, const float* cout_scale, const float* bias, int act, float leak,
                                 float* y) {
    MPG_REQUIRE(x && w_hwio && y, "mpg_conv2d_direct: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && cin >= 1 && cout >= 1, "mpg_conv2d_direct: bad shape");
    MPG_REQUIRE(kh >= 1 && kw >= 1 && stride_h >= 1 && stride_w >= 1, "mpg_conv2d_direct: bad kernel/stride");
    const int oh = (h + stride_h - 1) / stride_h, ow = (w + stride_w - 1) / stride_w;
    int pad_h = (oh - 1) * stride_h + kh - h; if (pad_h < 0) pad_h = 0;
    int pad_w = (ow - 1) * stride_w + kw - w; if (pad_w < 0) pad_w = 0;
    const size_t total = (size_t)n * oh * ow * cout;
    hipLaunchKernelGGL(conv_direct_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, x, n, h, w, cin,
                       w_hwio, kh, kw, cout, stride_h, stride_w, oh, ow, pad_h / 2, pad_w / 2, wscale, cout_scale,
                       bias, act, leak, y);
    MPG_LAUNCH_CHECK("conv_direct_kernel");
}

#define MPG_RESIZE_ENTRY(NAME, KERNEL)                                                                        \
    extern "C" int NAME(mpg_stream_t stream, const float* x, int n, int h, int w, int c, float* y, int oh,    \
                        int ow) {                                                                             \
        MPG_REQUIRE(x && y, #NAME ": null pointer");                                                          \
        MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1 && oh >= 1 && ow >= 1, #NAME ": bad shape");         \
        const size_t total = (size_t)n * oh * ow * c;                                                         \
        const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;                                     \
        hipLaunchKernelGGL(KERNEL, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, x, n, h, w, c, y, \
                           oh, ow, sy, sx);                                                                   \
        MPG_LAUNCH_CHECK(#KERNEL);                                                                            \
    }

MPG_RESIZE_ENTRY(mpg_resize_nearest, resize_nearest_kernel)
MPG_RESIZE_ENTRY(mpg_resize_bilinear, resize_bilinear_kernel)
MPG_RESIZE_ENTRY(mpg_resize_bicubic, resize_bicubic_kernel)

extern "C" int mpg_max_pool(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int k, int s, float* y,
                            unsigned char* arg) {
    MPG_REQUIRE(x && y, "mpg_max_pool: null pointer");
    MPG_REQUIRE(n >= 1 && c >= 1 && k >= 1 && k <= 15 && s >= 1 && h >= k && w >= k, "mpg_max_pool: bad shape");
    const size_t total = (size_t)n * ((h - k) / s + 1) * ((w - k) / s + 1) * c;
    hipLaunchKernelGGL(max_pool_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, x, n, h, w, c, k, s, y, arg);
    MPG_LAUNCH_CHECK("max_pool_kernel");
}

extern "C" int mpg_max_pool_bwd(mpg_stream_t stream, const float* dy, const unsigned char* arg, int n, int h, int w, int c,
                                int k, int s, float* dx) {
    MPG_REQUIRE(dy && arg && dx, "mpg_max_pool_bwd: null pointer");
    MPG_REQUIRE(n >= 1 && c >= 1 && k >= 1 && k <= 15 && s >= 1 && h >= k && w >= k, "mpg_max_pool_bwd: bad shape");
    hipError_t e = mpg::zero_async(dx, (size_t)n * h * w * c * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_max_pool_bwd: zero");
    const size_t total = (size_t)n * ((h - k) / s + 1) * ((w - k) / s + 1) * c;
    hipLaunchKernelGGL(max_pool_bwd_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, dy, arg, n, h, w, c, k,
                       s, dx);
    MPG_LAUNCH_CHECK("max_pool_bwd_kernel");
}

extern "C" int mpg_avg_pool2(mpg_stream_t stream, const float* x, int n, int h, int w, int c, float* y) {
    MPG_REQUIRE(x && y, "mpg_avg_pool2: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 2 && w >= 2 && c >= 1, "mpg_avg_pool2: bad shape");
    const size_t total = (size_t)n * (h / 2) * (w / 2) * c;
    hipLaunchKernelGGL(avg_pool2_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, x, n, h, w, c, y);
    MPG_LAUNCH_CHECK("avg_pool2_kernel");
}

extern "C" int mpg_pixel_norm(mpg_stream_t stream, const float* x, size_t npix, int c, float eps, float* y) {
    MPG_REQUIRE(x && y, "mpg_pixel_norm: null pointer");
    MPG_REQUIRE(npix >= 1 && c >= 1, "mpg_pixel_norm: bad shape");
    int lanes = 1;
    while (lanes * 2 <= c && lanes < 64) lanes <<= 1;
    hipLaunchKernelGGL(pixel_norm_kernel, dim3(grid_for(npix * lanes)), dim3(BLK), 0, (hipStream_t)stream, x, npix, c, lanes,
                       eps, y);
    MPG_LAUNCH_CHECK("pixel_norm_kernel");
}

extern "C" int mpg_minibatch_stddev(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int group_size,
                                    float* stat, float* y) {
    MPG_REQUIRE(x && stat && y, "mpg_minibatch_stddev: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1 && group_size >= 1, "mpg_minibatch_stddev: bad shape");
    const int g = group_size < n ? group_size : n;
    MPG_REQUIRE(n % g == 0, "mpg_minibatch_stddev: batch %d is not divisible by the group size %d", n, g);
    const int m = n / g;
    const size_t hwc = (size_t)h * w * c;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = mpg::zero_async(stat, (size_t)m * sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_minibatch_stddev: zero");
    unsigned bx = (unsigned)((hwc + BLK * 4 - 1) / (BLK * 4));
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(mbstd_stat_kernel, dim3(bx, m), dim3(BLK), 0, s, x, g, m, hwc, stat);
    const size_t total = (size_t)n * h * w * (c + 1);
    hipLaunchKernelGGL(mbstd_concat_kernel, dim3(grid_for(total)), dim3(BLK), 0, s, x, stat, m, (size_t)h * w, c, total,
                       1.f / (float)hwc, y);
    MPG_LAUNCH_CHECK("mbstd kernels");
}

extern "C" int mpg_minibatch_stddev_bwd(mpg_stream_t stream, const float* x, const float* dy, int n, int h, int w, int c,
                                        int group_size, float* dstat, float* dx) {
    MPG_REQUIRE(x && dy && dstat && dx, "mpg_minibatch_stddev_bwd: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1 && group_size >= 1, "mpg_minibatch_stddev_bwd: bad shape");
    const int g = group_size < n ? group_size : n;
    MPG_REQUIRE(n % g == 0, "mpg_minibatch_stddev_bwd: batch %d is not divisible by the group size %d", n, g);
    const int m = n / g;
    const size_t hwc = (size_t)h * w * c;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = mpg::zero_async(dstat, (size_t)m * sizeof(float), s);
    if (e != hipSuccess) return mpg::hip_check(e, "mpg_minibatch_stddev_bwd: zero");
    const size_t per_group = (size_t)g * h * w;
    unsigned bx = (unsigned)((per_group + BLK * 4 - 1) / (BLK * 4));
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(mbstd_bwd_stat_kernel, dim3(bx, m), dim3(BLK), 0, s, dy, n, m, (size_t)h * w, c, dstat);
    hipLaunchKernelGGL(mbstd_bwd_kernel, dim3(grid_for(hwc), m), dim3(BLK), 0, s, x, dy, dstat, g, m, hwc, c, dx);
    MPG_LAUNCH_CHECK("mbstd backward kernels");
}

extern "C" int mpg_add_act(mpg_stream_t stream, const float* a, const float* b, size_t n, int act, float leak,
                           float* y) {
    MPG_REQUIRE(a && y, "mpg_add_act: null pointer");
    if (n == 0) return MPG_OK;
    hipLaunchKernelGGL(add_act_kernel, dim3(grid_for(n)), dim3(BLK), 0, (hipStream_t)stream, a, b, n, act, leak, y);
    MPG_LAUNCH_CHECK("add_act_kernel");
}

extern "C" int mpg_axis_zoom_linear(mpg_stream_t stream, const float* v, size_t outer, int n, size_t inner,
                                    float* out, int big) {
    MPG_REQUIRE(v && out, "mpg_axis_zoom_linear: null pointer");
    MPG_REQUIRE(outer >= 1 && n >= 1 && inner >= 1 && big >= 1, "mpg_axis_zoom_linear: bad shape");
    const size_t total = outer * (size_t)big * inner;
    hipLaunchKernelGGL(axis_zoom_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, v, outer, n, inner,
                       out, big);
    MPG_LAUNCH_CHECK("axis_zoom_kernel");
}

extern "C" int mpg_volume_transpose(mpg_stream_t stream, const float* v, int d0, int d1, int d2, int c,
                                    const int* perm, const int* chan_map, float cutoff, float* out) {
    MPG_REQUIRE(v && out && perm, "mpg_volume_transpose: null pointer");
    MPG_REQUIRE(d0 >= 1 && d1 >= 1 && d2 >= 1 && c >= 1 && c <= 8, "mpg_volume_transpose: bad shape (c must be 1..8)");
    int seen = 0;
    for (int k = 0; k < 3; ++k) {
        MPG_REQUIRE(perm[k] >= 0 && perm[k] < 3, "mpg_volume_transpose: perm[%d]=%d", k, perm[k]);
        seen |= 1 << perm[k];
    }
    MPG_REQUIRE(seen == 7, "mpg_volume_transpose: perm is not a permutation");
    PermArgs a;
    a.din[0] = d0; a.din[1] = d1; a.din[2] = d2;
    bool ident_map = true;
    for (int k = 0; k < 3; ++k) a.perm[k] = perm[k];
    for (int k = 0; k < 8; ++k) {
        a.cmap[k] = (chan_map && k < c) ? chan_map[k] : k;
        if (k < c) {
            MPG_REQUIRE(a.cmap[k] >= 0 && a.cmap[k] < c, "mpg_volume_transpose: chan_map[%d]=%d", k, a.cmap[k]);
            if (a.cmap[k] != k) ident_map = false;
        }
    }
    a.c = c; a.cutoff = cutoff;
    const size_t total = (size_t)d0 * d1 * d2 * c;
    if (c == 1 && ident_map && perm[2] != 2) {
        const int ax_a = perm[2];
        const int ax_r = 1 - ax_a;
        const int dout[3] = {a.din[perm[0]], a.din[perm[1]], a.din[perm[2]]};
        size_t sin[3] = {(size_t)d1 * d2, (size_t)d2, 1};
        size_t sout_of_in[3];   // output stride of input axis q
        for (int k = 0; k < 3; ++k) {
            size_t sd = 1;
            for (int m = k + 1; m < 3; ++m) sd *= dout[m];
            sout_of_in[perm[k]] = sd;
        }
        TileArgs t;
        t.da = a.din[ax_a]; t.db = d2; t.dr = a.din[ax_r];
        t.sin_a = sin[ax_a]; t.sin_r = sin[ax_r];
        t.sout_b = sout_of_in[2]; t.sout_r = sout_of_in[ax_r];
        t.tiles_a = (t.da + TRT - 1) / TRT; t.tiles_b = (t.db + TRT - 1) / TRT;
        t.cutoff = cutoff;
        if (TRT == 64 && (t.da & 3) == 0 && (t.db & 3) == 0 && (t.sin_a & 3) == 0 && (t.sin_r & 3) == 0 && (t.sout_b & 3) == 0 &&
            (t.sout_r & 3) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)out & 15) == 0) {
            const size_t nb4 = (size_t)t.tiles_a * t.tiles_b * t.dr;
            MPG_REQUIRE(nb4 < (1UL << 31), "mpg_volume_transpose: grid too large");
            hipLaunchKernelGGL(transpose_tiled4_kernel, dim3((unsigned)nb4), dim3(BLK), 0, (hipStream_t)stream, v, t, out);
            MPG_LAUNCH_CHECK("transpose_tiled4_kernel");
        }
        const size_t nblk = (size_t)t.tiles_a * t.tiles_b * t.dr;
        MPG_REQUIRE(nblk < (1UL << 31), "mpg_volume_transpose: grid too large");
        constexpr size_t tile_bytes = (size_t)TRT * (TRT + 1) * sizeof(float);
        if (tile_bytes > 48 * 1024) {
            static int lds_limit[64] = {0};
            hipError_t e = mpg::ensure_dyn_lds(reinterpret_cast<const void*>(&transpose_tiled_kernel), (int)tile_bytes, lds_limit);
            if (e != hipSuccess) return mpg::hip_check(e, "mpg_volume_transpose: dynamic LDS");
        }
        hipLaunchKernelGGL(transpose_tiled_kernel, dim3((unsigned)nblk), dim3(BLK), tile_bytes, (hipStream_t)stream, v, t, out);
        MPG_LAUNCH_CHECK("transpose_tiled_kernel");
    }
    if (c == 1 && ident_map && perm[0] == 1 && perm[1] == 0 && perm[2] == 2 && d2 % 4 == 0 && ((uintptr_t)v & 15) == 0 &&
        ((uintptr_t)out & 15) == 0) {
        hipLaunchKernelGGL(swap01_rows_kernel, dim3(grid_for(total / 4)), dim3(BLK), 0, (hipStream_t)stream,
                           reinterpret_cast<const f4v*>(v), d0, d1, d2 / 4, cutoff, reinterpret_cast<f4v*>(out));
        MPG_LAUNCH_CHECK("swap01_rows_kernel");
    }
    hipLaunchKernelGGL(transpose_generic_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, v, a, out);
    MPG_LAUNCH_CHECK("transpose_generic_kernel");
}

extern "C" int mpg_add_adjacent(mpg_stream_t stream, const float* in, int s_total, size_t hw, int c, int s_off,
                                int s_cnt, float* out) {
    MPG_REQUIRE(in && out, "mpg_add_adjacent: null pointer");
    MPG_REQUIRE(s_total >= 1 && hw >= 1 && c >= 1 && s_off >= 0 && s_cnt >= 1 && s_off + s_cnt <= s_total,
                "mpg_add_adjacent: bad slice range");
    const size_t total = (size_t)s_cnt * hw * (c + 2);
    hipLaunchKernelGGL(add_adjacent_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, in, s_total, hw,
                       c, s_off, s_cnt, out);
    MPG_LAUNCH_CHECK("add_adjacent_kernel");
}

// out[p][j] = (src_j[p] * scale[j]) * scale2[j]: channel j of the output is channel map[j] of a (map[j] < ca) or map[j] - ca
// of b; two factors applied one after the other, as the reference scales the velocities twice (4x.py:278, 283)
struct GatherArgs { int map[MPG_GATHER_MAX_C]; float scale[MPG_GATHER_MAX_C]; float scale2[MPG_GATHER_MAX_C]; };

__global__ void channel_gather_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb, size_t npix,
                                      int cout, GatherArgs g, float* __restrict__ out) {
    const size_t total = npix * cout;
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < total; i += (size_t)gridDim.x * BLK) {
        const size_t p = i / cout;
        const int j = (int)(i - p * cout);
        const int m = g.map[j];
        const float v = m < ca ? a[p * ca + m] : b[p * cb + (m - ca)];
        out[i] = (v * g.scale[j]) * g.scale2[j];
    }
}

extern "C" int mpg_channel_gather(mpg_stream_t stream, const float* a, int ca, const float* b, int cb, size_t npix,
                                  const int* map, const float* scale, const float* scale2, int cout, float* out) {
    MPG_REQUIRE(a && map && out, "mpg_channel_gather: null pointer");
    MPG_REQUIRE(ca >= 1 && cb >= 0 && (cb == 0 || b != nullptr) && cout >= 1 && cout <= MPG_GATHER_MAX_C,
                "mpg_channel_gather: bad channel counts %d + %d -> %d", ca, cb, cout);
    GatherArgs g;
    for (int j = 0; j < MPG_GATHER_MAX_C; ++j) {
        g.map[j] = j < cout ? map[j] : 0;
        g.scale[j] = (j < cout && scale) ? scale[j] : 1.f;
        g.scale2[j] = (j < cout && scale2) ? scale2[j] : 1.f;
        MPG_REQUIRE(g.map[j] >= 0 && g.map[j] < ca + cb, "mpg_channel_gather: map[%d] = %d outside %d + %d channels", j, g.map[j], ca, cb);
    }
    if (npix == 0) return MPG_OK;
    size_t blocks = (npix * cout + BLK - 1) / BLK;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(channel_gather_kernel, dim3((unsigned)blocks), dim3(BLK), 0, (hipStream_t)stream, a, ca, b, cb, npix, cout, g, out);
    MPG_LAUNCH_CHECK("channel_gather_kernel");
}

extern "C" int mpg_cutoff(mpg_stream_t stream, const float* v, size_t n, float cutoff, float* out) {
    MPG_REQUIRE(v && out, "mpg_cutoff: null pointer");
    if (n == 0) return MPG_OK;
    if (n >= 1024 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        const size_t n4 = n / 4;
        hipLaunchKernelGGL(cutoff4_kernel, dim3(grid_for(n4)), dim3(BLK), 0, (hipStream_t)stream,
                           reinterpret_cast<const f4v*>(v), n4, cutoff, reinterpret_cast<f4v*>(out));
        if (n4 * 4 < n)
            hipLaunchKernelGGL(cutoff_kernel, dim3(1), dim3(BLK), 0, (hipStream_t)stream, v + n4 * 4, n - n4 * 4, cutoff, out + n4 * 4);
        MPG_LAUNCH_CHECK("cutoff4_kernel");
    }
    hipLaunchKernelGGL(cutoff_kernel, dim3(grid_for(n)), dim3(BLK), 0, (hipStream_t)stream, v, n, cutoff, out);
    MPG_LAUNCH_CHECK("cutoff_kernel");
}

extern "C" int mpg_conv2d_transpose(mpg_stream_t stream, const float* x, int n, int h, int w, int cin, const float* w_hwoi,
                                    int kh, int kw, int cout, int stride_h, int stride_w, float wscale, const float* bias,
                                    int act, float leak, float* y) {
    MPG_REQUIRE(x && w_hwoi && y, "mpg_conv2d_transpose: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && cin >= 1 && cout >= 1 && kh >= 1 && kw >= 1 && stride_h >= 1 && stride_w >= 1,
                "mpg_conv2d_transpose: bad shape");
    MPG_REQUIRE(act >= MPG_ACT_NONE && act <= MPG_ACT_TANH, "mpg_conv2d_transpose: bad activation %d", act);
    const int pad_h = kh - stride_h > 0 ? kh - stride_h : 0, pad_w = kw - stride_w > 0 ? kw - stride_w : 0;
    const size_t total = (size_t)n * h * stride_h * w * stride_w * cout;
    hipLaunchKernelGGL(conv_transpose_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, x, n, h, w, cin, w_hwoi,
                       kh, kw, cout, stride_h, stride_w, pad_h / 2, pad_w / 2, wscale, bias, act, leak, y);
    MPG_LAUNCH_CHECK("conv_transpose_kernel");
}

extern "C" int mpg_depth_to_space(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int r, float* y) {
    MPG_REQUIRE(x && y, "mpg_depth_to_space: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && r >= 1 && c >= r * r && c % (r * r) == 0,
                "mpg_depth_to_space: %d channels are not a multiple of %d^2", c, r);
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(depth_to_space_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, x, n, h, w, c / (r * r),
                       r, y);
    MPG_LAUNCH_CHECK("depth_to_space_kernel");
}
