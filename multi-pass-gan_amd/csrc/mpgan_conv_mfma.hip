// Fused implicit-GEMM convolution on the gfx950 matrix cores.
//
// One workgroup (4 waves) produces an 8 x 32 pixel tile of ALL output
// channels (<= 128).  The contraction index K runs over (segment, channel
// chunk, tap, 8-channel group); for every channel chunk the input tile with
// its halo is staged ONCE in LDS as fp16 [pixel][channel] (im2col-free: the
// 25 taps of a 5x5 kernel are 25 shifted windows of the same LDS image) and
// the pre-packed fp16 weights are streamed through a double-buffered LDS
// stage.  v_mfma_f32_32x32x16_f16 computes D[cout][pixel] += W[cout][k] * X[k][pixel]:
// the weights are the A operand, the pixels the B operand, so that each lane
// ends up with 4 consecutive output channels of one pixel per accumulator
// quad (16-byte NHWC stores).
//
// Replaces tf.nn.conv2d + bias + batch_norm + activation (+ residual 1x1 conv,
// + pixel_norm, + nearest upsample, + channel concat) of
// tools_wscale/GAN.py:80-119,472-474,501-541 and GAN/multipassGAN-4x.py:505-526,
// GAN/multipassGAN-out.py:220-237,357 (reference tree).
#include "mpgan_internal.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 8;     // tile rows
constexpr int TW = 32;    // tile cols == MFMA N dimension
constexpr int TAPOFF_BYTES = 1024;

struct SegArgs {
    const float* x;
    const char* w;
    int cin, cin_stride, c_off, kh, kw, up;
    int kc, g, nchunks, sc, ps;
    int oy, ox;     // window origin of this segment inside the halo tile
    int hs, ws;     // source height / width (h >> up, w >> up)
    int vec4;       // 16-byte aligned channel vectors
};

struct ConvArgs {
    int n, h, w, cout, nseg;
    SegArgs seg[MPG_MAX_SEG];
    const float* bias;
    int act;
    float leak;
    int pn;
    float pn_eps;
    const float* post_add;
    int pa_stride, pa_coff;
    float* y;
    int halo_h, halo_w, pad_t, pad_l;
    int in_plane;   // bytes of one LDS halo plane (max over segments)
    int tiles_x, tiles_y;
};

// The argument block is read through the kernarg segment pointer (constant address space,
// scalar loads) so that the runtime-indexed segment table never lands in scratch.
typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
typedef const __attribute__((address_space(4))) SegArgs* KSeg;

template <int NT, int PREC, int KS>
__global__ __launch_bounds__(256, (NT >= 3 ? 2 : 3)) void conv_mfma_kernel(const ConvArgs a_unused) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KArgs ap = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const auto& a = *ap;
    constexpr int NPL = (PREC == 3) ? 2 : 1;
    constexpr int WPLANE = KS * NT * 1024;          // bytes of one weight plane per stage
    constexpr int WSTAGE = WPLANE * NPL;            // bytes per stage (hi [+ lo])
    constexpr int NPASS = (WSTAGE + 4095) / 4096;   // 16-B copies per thread per stage

    int* tapoff = reinterpret_cast<int*>(smem);
    char* in_lds = smem + TAPOFF_BYTES;
    char* w_lds = in_lds + a.in_plane * NPL;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int hh = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give
    // each XCD a contiguous run of tiles => neighbouring halos hit the same L2.
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int tx = bid % a.tiles_x;
    const int t2 = bid / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    f32x16 acc[2][NT];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pt][nt][i] = 0.f;

    const int halo_px = a.halo_h * a.halo_w;

    for (int s = 0; s < a.nseg; ++s) {
        const auto& sg = ap->seg[s];
        const int G = sg.g;
        const int TG = sg.kh * sg.kw * G;
        const int ps = sg.ps;

        // tap/group -> LDS byte offset table (all waves passed the previous stage barrier)
        for (int q = tid; q < sg.sc * KS * 2; q += 256) {
            int off = 0;
            if (q < TG) {
                const int tap = q / G;
                const int g = q - tap * G;
                const int dy = tap / sg.kw;
                const int dx = tap - dy * sg.kw;
                off = ((dy + sg.oy) * a.halo_w + dx + sg.ox) * ps + g * 16;
            }
            tapoff[q] = off;
        }
        const int pixb0 = ((2 * wave) * a.halo_w + r) * ps;
        const int pixb1 = pixb0 + a.halo_w * ps;

        // weight stages are contiguous over (chunk, stage): stream them through a 2-deep LDS ring,
        // always one stage ahead (the prefetch of the last stage re-reads it: no branch, no hazard)
        const int total_stages = sg.nchunks * sg.sc;
        auto stage_off = [&](int i) -> int {   // byte offset of this thread's i-th 16-B piece, clamped
            const int o = (i * 256 + tid) * 16;
            return o < WSTAGE ? o : WSTAGE - 16;
        };
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int o = stage_off(i);
            *reinterpret_cast<uint4*>(w_lds + o) = *reinterpret_cast<const uint4*>(sg.w + o);
        }
        int cur = 0;
        for (int ch = 0; ch < sg.nchunks; ++ch) {
            // ---- stage the input halo tile of this channel chunk (fp32 -> fp16 hi[/lo]) ----
            for (int idx = tid; idx < halo_px * G; idx += 256) {
                const int p = idx / G;
                const int g = idx - p * G;
                const int hy = p / a.halo_w;
                const int hx = p - hy * a.halo_w;
                const int yy = y0 - a.pad_t + hy;
                const int xx = x0 - a.pad_l + hx;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
                if (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) {
                    const int c = ch * sg.kc + g * 8;
                    const int rem = sg.cin - c;
                    const float* src = sg.x +
                        ((size_t)(n * sg.hs + (yy >> sg.up)) * sg.ws + (xx >> sg.up)) * sg.cin_stride + sg.c_off + c;
                    if (rem >= 8 && sg.vec4) {
                        const float4 a0 = *reinterpret_cast<const float4*>(src);
                        const float4 a1 = *reinterpret_cast<const float4*>(src + 4);
                        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w;
                        v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (j < rem) v[j] = src[j];
                    }
                }
                half8 hi;
#pragma unroll
                for (int j = 0; j < 8; ++j) hi[j] = (_Float16)v[j];
                *reinterpret_cast<half8*>(in_lds + p * ps + g * 16) = hi;
                if (PREC == 3) {
                    half8 lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) lo[j] = (_Float16)(v[j] - (float)hi[j]);
                    *reinterpret_cast<half8*>(in_lds + a.in_plane + p * ps + g * 16) = lo;
                }
            }
            __syncthreads();

            for (int st = 0; st < sg.sc; ++st) {
                const int gst = ch * sg.sc + st;
                const int nst = gst + 1 < total_stages ? gst + 1 : gst;
                uint4 pre[NPASS];
                {
                    const char* wn = sg.w + (size_t)nst * WSTAGE;
#pragma unroll
                    for (int i = 0; i < NPASS; ++i) pre[i] = *reinterpret_cast<const uint4*>(wn + stage_off(i));
                }
                const char* wb = w_lds + cur * WSTAGE;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int toff = tapoff[2 * (st * KS + ks) + hh];
                    half8 b_hi[2], b_lo[2], a_hi[NT], a_lo[NT];
                    b_hi[0] = *reinterpret_cast<const half8*>(in_lds + pixb0 + toff);
                    b_hi[1] = *reinterpret_cast<const half8*>(in_lds + pixb1 + toff);
                    if (PREC == 3) {
                        b_lo[0] = *reinterpret_cast<const half8*>(in_lds + a.in_plane + pixb0 + toff);
                        b_lo[1] = *reinterpret_cast<const half8*>(in_lds + a.in_plane + pixb1 + toff);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        a_hi[nt] = *reinterpret_cast<const half8*>(wb + ((ks * NT + nt) * 64 + lane) * 16);
                        if (PREC == 3)
                            a_lo[nt] = *reinterpret_cast<const half8*>(wb + WPLANE + ((ks * NT + nt) * 64 + lane) * 16);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) {
                            acc[pt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[nt], b_hi[pt], acc[pt][nt], 0, 0, 0);
                            if (PREC == 3) {
                                acc[pt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[nt], b_hi[pt], acc[pt][nt], 0, 0, 0);
                                acc[pt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[nt], b_lo[pt], acc[pt][nt], 0, 0, 0);
                            }
                        }
                }
                {
                    char* wd = w_lds + (cur ^ 1) * WSTAGE;
#pragma unroll
                    for (int i = 0; i < NPASS; ++i) *reinterpret_cast<uint4*>(wd + stage_off(i)) = pre[i];
                }
                __syncthreads();
                cur ^= 1;
            }
        }
    }

    // ---------------- epilogue: bias, activation, pixel norm, post add, NHWC store ----------------
    // accumulator element i of n-tile nt: output channel nt*32 + 8*(i>>2) + 4*hh + (i&3), pixel r.
    const int px = x0 + r;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int py = y0 + 2 * wave + pt;
        float ss = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = nt * 32 + 8 * (i >> 2) + 4 * hh + (i & 3);
                float v = acc[pt][nt][i];
                if (a.bias != nullptr && co < a.cout) v += a.bias[co];
                v = mpg::apply_act(v, a.act, a.leak);
                if (co >= a.cout) v = 0.f;
                acc[pt][nt][i] = v;
                ss += v * v;
            }
        if (a.pn) {
            ss += __shfl_xor(ss, 32);
            const float sc = rsqrtf(ss / (float)a.cout + a.pn_eps);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[pt][nt][i] *= sc;
        }
        if (py < a.h && px < a.w) {
            const size_t pix = ((size_t)n * a.h + py) * a.w + px;
            float* dst = a.y + pix * a.cout;
            const float* pa = a.post_add ? a.post_add + pix * a.pa_stride + a.pa_coff : nullptr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int co0 = nt * 32 + 8 * q4 + 4 * hh;
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[i] = acc[pt][nt][4 * q4 + i];
                        if (pa != nullptr && co0 + i < a.cout) o[i] += pa[co0 + i];
                    }
                    if ((a.cout & 3) == 0) {
                        if (co0 < a.cout) *reinterpret_cast<float4*>(dst + co0) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (co0 + i < a.cout) dst[co0 + i] = o[i];
                    }
                }
        }
    }
}

// weights HWIO fp32 -> per (chunk, stage) fragment-ordered fp16 hi [lo] planes
__global__ void pack_weights_kernel(const float* __restrict__ w, int kh, int kw, int cin_total, int c_off,
                                    int cin, int cout, float wscale, const float* __restrict__ cscale,
                                    int NT, int KS, int NPL, int kc, int g, int nchunks, int sc,
                                    _Float16* __restrict__ out) {
    const long total = (long)nchunks * sc * KS * NT * 512;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7;
    const int lane = (idx >> 3) & 63;
    long rest = idx >> 9;
    const int nt = rest % NT; rest /= NT;
    const int ks = rest % KS; rest /= KS;
    const int st = rest % sc;
    const int c = rest / sc;
    const int r = lane & 31, hh = lane >> 5;
    const int q = 2 * (st * KS + ks) + hh;
    float v = 0.f;
    if (q < kh * kw * g) {
        const int tap = q / g;
        const int gg = q - tap * g;
        const int chn = c * kc + gg * 8 + j;
        const int co = nt * 32 + r;
        if (chn < cin && co < cout) {
            v = w[((size_t)tap * cin_total + c_off + chn) * cout + co] * wscale;
            if (cscale != nullptr) v *= cscale[co];
        }
    }
    const long plane = (long)KS * NT * 512;
    const long stage = (long)c * sc + st;
    const long off = ((long)(ks * NT + nt) * 64 + lane) * 8 + j;
    const _Float16 hi = (_Float16)v;
    out[stage * plane * NPL + off] = hi;
    if (NPL == 2) out[stage * plane * NPL + plane + off] = (_Float16)(v - (float)hi);
}

template <int NT, int PREC>
int launch_ks(int ks, dim3 grid, size_t lds, hipStream_t st, const ConvArgs& a) {
    // dynamic LDS beyond the 64 KiB default needs the per-function opt-in
    if (ks == 2) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<NT, PREC, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((conv_mfma_kernel<NT, PREC, 2>), grid, dim3(256), lds, st, a);
    } else {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<NT, PREC, 4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((conv_mfma_kernel<NT, PREC, 4>), grid, dim3(256), lds, st, a);
    }
    return MPG_OK;
}

template <int PREC>
int launch_nt(int nt, int ks, dim3 grid, size_t lds, hipStream_t st, const ConvArgs& a) {
    switch (nt) {
        case 1: return launch_ks<1, PREC>(ks, grid, lds, st, a);
        case 2: return launch_ks<2, PREC>(ks, grid, lds, st, a);
        case 3: return launch_ks<3, PREC>(ks, grid, lds, st, a);
        default: return launch_ks<4, PREC>(ks, grid, lds, st, a);
    }
}

int default_ks(int prec, int ks) {
    if (ks == 2 || ks == 4) return ks;
    return prec == MPG_PREC_F16X3 ? 2 : 4;
}

int default_kc(int prec, int kc_max) {
    if (kc_max == 8 || kc_max == 16 || kc_max == 24 || kc_max == 32) return kc_max;
    return prec == MPG_PREC_F16X3 ? 16 : 32;
}

}  // namespace

extern "C" size_t mpg_conv_pack_size(int kh, int kw, int cin, int cout, int prec, int kc_max, int ks) {
    if (kh < 1 || kw < 1 || cin < 1 || cout < 1 || cout > 128) return 0;
    ks = default_ks(prec, ks);
    const mpg::SegPlan p = mpg::make_plan(kh, kw, cin, default_kc(prec, kc_max), ks);
    const int nt = (cout + 31) / 32;
    const int npl = prec == MPG_PREC_F16X3 ? 2 : 1;
    return (size_t)p.nchunks * p.sc * ks * nt * 1024 * npl;
}

extern "C" int mpg_conv_pack_weights(mpg_stream_t stream, const float* w_hwio, int kh, int kw,
                                     int w_cin_total, int w_c_off, int cin, int cout, float wscale,
                                     const float* cout_scale, int prec, int kc_max, int ks, void* out,
                                     size_t out_bytes) {
    MPG_REQUIRE(w_hwio && out, "mpg_conv_pack_weights: null pointer");
    MPG_REQUIRE(prec == MPG_PREC_F16X1 || prec == MPG_PREC_F16X3, "mpg_conv_pack_weights: bad prec %d", prec);
    MPG_REQUIRE(kh >= 1 && kh <= 7 && kw >= 1 && kw <= 7, "mpg_conv_pack_weights: kernel %dx%d unsupported", kh, kw);
    MPG_REQUIRE(cin >= 1 && w_c_off >= 0 && w_c_off + cin <= w_cin_total, "mpg_conv_pack_weights: channel range");
    MPG_REQUIRE(cout >= 1 && cout <= 128, "mpg_conv_pack_weights: cout %d not in 1..128", cout);
    const size_t need = mpg_conv_pack_size(kh, kw, cin, cout, prec, kc_max, ks);
    MPG_REQUIRE(out_bytes >= need, "mpg_conv_pack_weights: out buffer %zu < %zu", out_bytes, need);
    ks = default_ks(prec, ks);
    const mpg::SegPlan p = mpg::make_plan(kh, kw, cin, default_kc(prec, kc_max), ks);
    const int nt = (cout + 31) / 32;
    const int npl = prec == MPG_PREC_F16X3 ? 2 : 1;
    const long total = (long)p.nchunks * p.sc * ks * nt * 512;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, kh, kw,
                       w_cin_total, w_c_off, cin, cout, wscale, cout_scale, nt, ks, npl, p.kc, p.g, p.nchunks, p.sc,
                       (_Float16*)out);
    MPG_LAUNCH_CHECK("pack_weights_kernel");
}

extern "C" int mpg_conv2d_fused(mpg_stream_t stream, const mpg_conv_desc* d) {
    MPG_REQUIRE(d != nullptr, "mpg_conv2d_fused: null desc");
    MPG_REQUIRE(d->n >= 1 && d->h >= 1 && d->w >= 1, "mpg_conv2d_fused: bad shape %d x %d x %d", d->n, d->h, d->w);
    MPG_REQUIRE(d->cout >= 1 && d->cout <= 128, "mpg_conv2d_fused: cout %d not in 1..128", d->cout);
    MPG_REQUIRE(d->nseg >= 1 && d->nseg <= MPG_MAX_SEG, "mpg_conv2d_fused: nseg %d", d->nseg);
    MPG_REQUIRE(d->y != nullptr, "mpg_conv2d_fused: null output");
    MPG_REQUIRE(d->prec == MPG_PREC_F16X1 || d->prec == MPG_PREC_F16X3, "mpg_conv2d_fused: bad prec %d", d->prec);
    MPG_REQUIRE(d->act >= MPG_ACT_NONE && d->act <= MPG_ACT_TANH, "mpg_conv2d_fused: bad act %d", d->act);
    const int ks = default_ks(d->prec, d->ks);
    const int kc_max = default_kc(d->prec, d->kc_max);
    const int npl = d->prec == MPG_PREC_F16X3 ? 2 : 1;
    const int nt = (d->cout + 31) / 32;

    ConvArgs a;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cout = d->cout; a.nseg = d->nseg;
    int khm = 1, kwm = 1;
    for (int s = 0; s < d->nseg; ++s) {
        const mpg_conv_seg& g = d->seg[s];
        MPG_REQUIRE(g.x && g.wpack, "mpg_conv2d_fused: segment %d null pointer", s);
        MPG_REQUIRE(g.kh >= 1 && g.kh <= 7 && g.kw >= 1 && g.kw <= 7, "mpg_conv2d_fused: segment %d kernel %dx%d", s, g.kh, g.kw);
        MPG_REQUIRE(g.cin >= 1 && g.c_off >= 0 && g.c_off + g.cin <= g.cin_stride, "mpg_conv2d_fused: segment %d channel range", s);
        MPG_REQUIRE(g.up_log2 >= 0 && g.up_log2 <= 4, "mpg_conv2d_fused: segment %d up_log2 %d", s, g.up_log2);
        MPG_REQUIRE((d->h % (1 << g.up_log2)) == 0 && (d->w % (1 << g.up_log2)) == 0,
                    "mpg_conv2d_fused: segment %d: %dx%d not divisible by upsample %d", s, d->h, d->w, 1 << g.up_log2);
        khm = g.kh > khm ? g.kh : khm;
        kwm = g.kw > kwm ? g.kw : kwm;
    }
    a.pad_t = (khm - 1) / 2;
    a.pad_l = (kwm - 1) / 2;
    a.halo_h = TH + khm - 1;
    a.halo_w = TW + kwm - 1;
    int max_ps = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const mpg_conv_seg& g = d->seg[s];
        const mpg::SegPlan p = mpg::make_plan(g.kh, g.kw, g.cin, kc_max, ks);
        MPG_REQUIRE(p.sc * ks * 2 <= TAPOFF_BYTES / 4, "mpg_conv2d_fused: segment %d tap table too large", s);
        SegArgs& o = a.seg[s];
        o.x = g.x; o.w = (const char*)g.wpack;
        o.cin = g.cin; o.cin_stride = g.cin_stride; o.c_off = g.c_off; o.kh = g.kh; o.kw = g.kw; o.up = g.up_log2;
        o.kc = p.kc; o.g = p.g; o.nchunks = p.nchunks; o.sc = p.sc; o.ps = p.ps;
        o.oy = a.pad_t - (g.kh - 1) / 2;
        o.ox = a.pad_l - (g.kw - 1) / 2;
        o.hs = d->h >> g.up_log2; o.ws = d->w >> g.up_log2;
        o.vec4 = ((g.cin_stride & 3) == 0 && (g.c_off & 3) == 0 && (((uintptr_t)g.x) & 15) == 0) ? 1 : 0;
        max_ps = p.ps > max_ps ? p.ps : max_ps;
    }
    for (int s = d->nseg; s < MPG_MAX_SEG; ++s) a.seg[s] = a.seg[0];
    a.bias = d->bias; a.act = d->act; a.leak = d->leak; a.pn = d->pixel_norm; a.pn_eps = d->pn_eps;
    a.post_add = d->post_add; a.pa_stride = d->post_add_stride; a.pa_coff = d->post_add_coff;
    MPG_REQUIRE(!d->post_add || d->post_add_coff + d->cout <= d->post_add_stride, "mpg_conv2d_fused: post_add channel range");
    a.y = d->y;
    a.in_plane = ((a.halo_h * a.halo_w * max_ps) + 15) & ~15;
    a.tiles_x = (d->w + TW - 1) / TW;
    a.tiles_y = (d->h + TH - 1) / TH;
    const long nblk = (long)d->n * a.tiles_x * a.tiles_y;
    MPG_REQUIRE(nblk < (1L << 31), "mpg_conv2d_fused: grid too large");
    const size_t lds = TAPOFF_BYTES + (size_t)a.in_plane * npl + 2 * (size_t)ks * nt * 1024 * npl;
    MPG_REQUIRE(lds <= 160 * 1024, "mpg_conv2d_fused: LDS budget %zu exceeds 160 KiB", lds);
    const dim3 grid((unsigned)nblk);
    if (d->prec == MPG_PREC_F16X3)
        launch_nt<3>(nt, ks, grid, lds, (hipStream_t)stream, a);
    else
        launch_nt<1>(nt, ks, grid, lds, (hipStream_t)stream, a);
    MPG_LAUNCH_CHECK("conv_mfma_kernel");
}
