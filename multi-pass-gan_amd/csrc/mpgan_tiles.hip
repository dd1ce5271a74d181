// Device side of the training-tile supply: the array work of tools_wscale/tilecreator_t.py on frames
// that stay resident in HBM -- tile cutting (:403-450,576-642), the scipy.ndimage resamplings of the augmentation
// (zoom :808-845, affine_transform :858-879; order 1, mode 'constant'), quarter turns / flips with the velocity
// components following the grid (:700-806), and the semi-Lagrangian look-up positions of coherent triples (:1293-1378).
// All of it is HBM-bound gather work on small tiles; one thread per output element, channels fastest.
// The random decisions (which frame, which offset, retries on the density test, augmentation parameters) stay on the
// host in multi-pass-gan_amd/tiles_device.py so that the reference's random streams are consumed identically.
#include "mpgan_internal.h"

namespace {

constexpr int BLK = 256;
inline unsigned grid_for(size_t n) { return (unsigned)((n + BLK - 1) / BLK); }

// one row per tile: frame, first channel, z0, y0, x0 (low-res or high-res offsets, as the caller computed them)
__global__ void tile_gather_kernel(const float* __restrict__ frames, int Z, int Y, int X, int Cf, const int* __restrict__ table,
                                   int B, int tz, int ty, int tx, int C, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t per = (size_t)tz * ty * tx * C;
    if (idx >= per * B) return;
    const int b = idx / per;
    size_t r = idx - (size_t)b * per;
    const int c = r % C; r /= C;
    const int x = r % tx; r /= tx;
    const int y = r % ty;
    const int z = r / ty;
    const int* t = table + b * 5;
    out[idx] = frames[((((size_t)t[0] * Z + t[2] + z) * Y + t[3] + y) * X + t[4] + x) * Cf + t[1] + c];
}

struct ResampleArgs {
    int zs, ys, xs, c;          // source [zs,ys,xs,c]
    int zd, yd, xd;             // destination
    double m[9], off[3];        // source coordinate = m * (z,y,x)_dst + off   (float64, like scipy)
    float mix[12 * 12];         // out channel i = sum_k mix[i*c + k] * interpolated channel k
    int use_mix;
};

// scipy.ndimage order-1 interpolation with mode 'constant', cval 0: a destination element whose source coordinate leaves
// [0, n-1] on any axis is 0 (no tolerance -- scipy.ndimage.zoom itself zeroes its last row when (n-1)/(N-1) * (N-1)
// rounds above n-1); inside, linear weights with the upper neighbour clamped to n-1 (its weight is 0 there).
__global__ void resample_kernel(const float* __restrict__ src, ResampleArgs a, float* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)a.zd * a.yd * a.xd;
    if (idx >= total) return;
    const int x = idx % a.xd;
    const int y = (idx / a.xd) % a.yd;
    const int z = idx / ((size_t)a.xd * a.yd);
    const double o[3] = {(double)z, (double)y, (double)x};
    const int dims[3] = {a.zs, a.ys, a.xs};
    int i0[3], i1[3];
    float w[3];
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double cc = a.m[3 * k] * o[0] + a.m[3 * k + 1] * o[1] + a.m[3 * k + 2] * o[2] + a.off[k];
        if (cc < 0.0 || cc > (double)(dims[k] - 1)) inside = false;
        const double fl = floor(cc);
        i0[k] = (int)fl;
        i1[k] = min(i0[k] + 1, dims[k] - 1);
        w[k] = (float)(cc - fl);
    }
    float val[12];
    for (int c = 0; c < a.c; ++c) val[c] = 0.f;
    if (inside) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int zz = (q & 4) ? i1[0] : i0[0], yy = (q & 2) ? i1[1] : i0[1], xx = (q & 1) ? i1[2] : i0[2];
            const float ww = ((q & 4) ? w[0] : 1.f - w[0]) * ((q & 2) ? w[1] : 1.f - w[1]) * ((q & 1) ? w[2] : 1.f - w[2]);
            if (ww != 0.f) {
                const float* s = src + (((size_t)zz * a.ys + yy) * a.xs + xx) * a.c;
                for (int c = 0; c < a.c; ++c) val[c] += ww * s[c];
            }
        }
    }
    float* d = dst + idx * a.c;
    if (!a.use_mix) {
        for (int c = 0; c < a.c; ++c) d[c] = val[c];
    } else {
        for (int i = 0; i < a.c; ++i) {
            float acc = 0.f;
            for (int k = 0; k < a.c; ++k) acc += a.mix[i * a.c + k] * val[k];
            d[i] = acc;
        }
    }
}

struct OrientArgs {
    int zs, ys, xs, c;          // source array
    int z0, y0, x0;             // crop offset in the source
    int dz, dy, dx;             // output size
    int perm[3];                // output axis k runs along source axis perm[k] of the CROPPED, not yet turned array
    int flip[3];                // ... backwards if flip[k]
    int cz, cy, cx;             // crop size in source axis order
    int cmap[12];
    float csign[12];
};

// the final crop of a tile, followed by the composed quarter turns / flips (an axis permutation with reversals) and the
// matching permutation / sign changes of the vector components, written straight into slot `slot` of the batch
__global__ void orient_kernel(const float* __restrict__ src, OrientArgs a, float* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)a.dz * a.dy * a.dx * a.c;
    if (idx >= total) return;
    const int c = idx % a.c;
    size_t r = idx / a.c;
    const int ox = r % a.dx; r /= a.dx;
    const int oy = r % a.dy;
    const int oz = r / a.dy;
    const int o[3] = {oz, oy, ox};
    const int csz[3] = {a.cz, a.cy, a.cx};
    int s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ax = a.perm[k];
        s[ax] = a.flip[k] ? csz[ax] - 1 - o[k] : o[k];
    }
    const float v = src[(((size_t)(a.z0 + s[0]) * a.ys + a.y0 + s[1]) * a.xs + a.x0 + s[2]) * a.c + a.cmap[c]];
    dst[idx] = a.csign[c] * v;
}

// getSemiLagrPosBatch (:1345-1378), 2D: positions (y, x) - v * dt on an n x n grid from MAC velocities [b,1,h,w,3]
// (x,y,z): the velocity is resampled to the output grid (map_coordinates order 1, mode 'nearest', at (i + 0.5) * h / n --
// the reference feeds cell-centre coordinates as array indices), centred (mean with the successor along its own axis, the
// last one repeated), divided by the resolution ratio and multiplied by dt[b]
__device__ __forceinline__ float sl_sample(const float* __restrict__ vel, int b, int h, int w, int comp, double cy, double cx) {
    cy = fmin(fmax(cy, 0.0), (double)(h - 1));
    cx = fmin(fmax(cx, 0.0), (double)(w - 1));
    const int y0 = (int)floor(cy), x0 = (int)floor(cx);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float fy = (float)(cy - y0), fx = (float)(cx - x0);
    const float* base = vel + (size_t)b * h * w * 3 + comp;
    const float v00 = base[((size_t)y0 * w + x0) * 3], v01 = base[((size_t)y0 * w + x1) * 3];
    const float v10 = base[((size_t)y1 * w + x0) * 3], v11 = base[((size_t)y1 * w + x1) * 3];
    return (1.f - fy) * ((1.f - fx) * v00 + fx * v01) + fy * ((1.f - fx) * v10 + fx * v11);
}

__global__ void semilagr_pos_kernel(const float* __restrict__ vel, const float* __restrict__ dt, int B, int h, int w, int n,
                                    float* __restrict__ pos) {
    const size_t idx = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t total = (size_t)B * n * n;
    if (idx >= total) return;
    const int j = idx % n;
    const int i = (idx / n) % n;
    const int b = idx / ((size_t)n * n);
    float vy, vx;
    if (n == w) {       // same resolution: no resampling (:1370-1372)
        const int i1 = min(i + 1, h - 1), j1 = min(j + 1, w - 1);
        const float* v = vel + (size_t)b * h * w * 3;
        vy = 0.5f * (v[((size_t)i * w + j) * 3 + 1] + v[((size_t)i1 * w + j) * 3 + 1]);
        vx = 0.5f * (v[((size_t)i * w + j) * 3] + v[((size_t)i * w + j1) * 3]);
    } else {
        const double ry = (double)h / n, rx = (double)w / n;
        const int i1 = min(i + 1, n - 1), j1 = min(j + 1, n - 1);
        const float fx = (float)w / (float)n;
        const float a = sl_sample(vel, b, h, w, 1, (i + 0.5) * ry, (j + 0.5) * rx);
        const float a1 = sl_sample(vel, b, h, w, 1, (i1 + 0.5) * ry, (j + 0.5) * rx);
        const float c = sl_sample(vel, b, h, w, 0, (i + 0.5) * ry, (j + 0.5) * rx);
        const float c1 = sl_sample(vel, b, h, w, 0, (i + 0.5) * ry, (j1 + 0.5) * rx);
        vy = 0.5f * (a + a1) / fx;
        vx = 0.5f * (c + c1) / fx;
    }
    pos[idx * 2] = ((float)i + 0.5f) - vy * dt[b];
    pos[idx * 2 + 1] = ((float)j + 0.5f) - vx * dt[b];
}

}  // namespace

extern "C" int mpg_tile_gather(mpg_stream_t stream, const float* frames, int n_frames, int z, int y, int x, int cf,
                               const int* table, int n_tiles, int tz, int ty, int tx, int c, float* out) {
    MPG_REQUIRE(frames && table && out, "mpg_tile_gather: null pointer");
    MPG_REQUIRE(n_frames >= 1 && z >= 1 && y >= 1 && x >= 1 && cf >= 1 && n_tiles >= 1 && tz >= 1 && ty >= 1 && tx >= 1 && c >= 1 &&
                    tz <= z && ty <= y && tx <= x && c <= cf, "mpg_tile_gather: bad shape");
    const size_t total = (size_t)n_tiles * tz * ty * tx * c;
    hipLaunchKernelGGL(tile_gather_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, frames, z, y, x, cf, table,
                       n_tiles, tz, ty, tx, c, out);
    MPG_LAUNCH_CHECK("tile_gather_kernel");
}

extern "C" int mpg_resample_affine(mpg_stream_t stream, const float* src, int zs, int ys, int xs, int c, float* dst, int zd, int yd,
                                   int xd, const double* matrix9, const double* offset3, const float* channel_mix) {
    MPG_REQUIRE(src && dst && matrix9 && offset3, "mpg_resample_affine: null pointer");
    MPG_REQUIRE(zs >= 1 && ys >= 1 && xs >= 1 && zd >= 1 && yd >= 1 && xd >= 1 && c >= 1 && c <= 12,
                "mpg_resample_affine: bad shape (1..12 channels)");
    ResampleArgs a;
    a.zs = zs; a.ys = ys; a.xs = xs; a.c = c; a.zd = zd; a.yd = yd; a.xd = xd;
    for (int i = 0; i < 9; ++i) a.m[i] = matrix9[i];
    for (int i = 0; i < 3; ++i) a.off[i] = offset3[i];
    a.use_mix = channel_mix != nullptr;
    for (int i = 0; i < 144; ++i) a.mix[i] = (channel_mix != nullptr && i < c * c) ? channel_mix[i] : 0.f;
    const size_t total = (size_t)zd * yd * xd;
    hipLaunchKernelGGL(resample_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, src, a, dst);
    MPG_LAUNCH_CHECK("resample_kernel");
}

extern "C" int mpg_tile_orient(mpg_stream_t stream, const float* src, int zs, int ys, int xs, int c, const int* crop_off3,
                               const int* crop_size3, const int* perm3, const int* flip3, const int* chan_map,
                               const float* chan_sign, float* dst) {
    MPG_REQUIRE(src && dst && crop_off3 && crop_size3 && perm3 && flip3, "mpg_tile_orient: null pointer");
    MPG_REQUIRE(c >= 1 && c <= 12, "mpg_tile_orient: 1..12 channels");
    OrientArgs a;
    a.zs = zs; a.ys = ys; a.xs = xs; a.c = c;
    a.z0 = crop_off3[0]; a.y0 = crop_off3[1]; a.x0 = crop_off3[2];
    a.cz = crop_size3[0]; a.cy = crop_size3[1]; a.cx = crop_size3[2];
    int seen = 0;
    for (int k = 0; k < 3; ++k) {
        MPG_REQUIRE(perm3[k] >= 0 && perm3[k] < 3, "mpg_tile_orient: perm");
        a.perm[k] = perm3[k]; a.flip[k] = flip3[k] != 0; seen |= 1 << perm3[k];
    }
    MPG_REQUIRE(seen == 7, "mpg_tile_orient: perm is not a permutation");
    MPG_REQUIRE(a.z0 >= 0 && a.y0 >= 0 && a.x0 >= 0 && a.z0 + a.cz <= zs && a.y0 + a.cy <= ys && a.x0 + a.cx <= xs,
                "mpg_tile_orient: crop leaves the source");
    const int csz[3] = {a.cz, a.cy, a.cx};
    a.dz = csz[a.perm[0]]; a.dy = csz[a.perm[1]]; a.dx = csz[a.perm[2]];
    for (int k = 0; k < 12; ++k) {
        a.cmap[k] = (chan_map != nullptr && k < c) ? chan_map[k] : k;
        a.csign[k] = (chan_sign != nullptr && k < c) ? chan_sign[k] : 1.f;
        if (k < c) MPG_REQUIRE(a.cmap[k] >= 0 && a.cmap[k] < c, "mpg_tile_orient: chan_map[%d]", k);
    }
    const size_t total = (size_t)a.dz * a.dy * a.dx * c;
    hipLaunchKernelGGL(orient_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, src, a, dst);
    MPG_LAUNCH_CHECK("orient_kernel");
}

extern "C" int mpg_semilagr_positions(mpg_stream_t stream, const float* vel, const float* dt, int n_batch, int h, int w, int n_out,
                                      float* pos) {
    MPG_REQUIRE(vel && dt && pos, "mpg_semilagr_positions: null pointer");
    MPG_REQUIRE(n_batch >= 1 && h >= 1 && w >= 1 && n_out >= 1, "mpg_semilagr_positions: bad shape");
    const size_t total = (size_t)n_batch * n_out * n_out;
    hipLaunchKernelGGL(semilagr_pos_kernel, dim3(grid_for(total)), dim3(BLK), 0, (hipStream_t)stream, vel, dt, n_batch, h, w, n_out,
                       pos);
    MPG_LAUNCH_CHECK("semilagr_pos_kernel");
}
