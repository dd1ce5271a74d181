// Fused implicit-GEMM convolution on the gfx950 matrix cores.
//
// Data layout.  Activations travel between fused convolutions as "G8" tensors:
//     [N][CG = ceil(C/8)][2 planes: hi, lo][H][W][8 x fp16]
// (value = hi + lo, both fp16, exact to 2^-22; channels beyond C are zero).  A channel group of
// a tile row is therefore a contiguous run of 16-byte pixels, so the input tile with its halo AND
// the pre-packed weights both stream into LDS by LDS-DMA (global_load_lds_dwordx4): no VALU, no
// staging registers, deep prefetch.  fp32 NHWC enters / leaves through mpg_f32_to_g8 and the
// optional fp32 output of the epilogue.
//
// Work decomposition.  One workgroup = 4 waves = (4*PT) x 32 output pixels x all NT*32 output
// channels; wave w owns tile rows [PT*w, PT*w+PT) (PT pixel tiles of 32 pixels) x NT cout tiles.
// v_mfma_f32_32x32x16_f16 computes D[cout][pixel] += W[cout][k] * X[k][pixel] (weights = A operand,
// pixels = B operand), so a lane ends with 4 consecutive channels of one pixel per accumulator quad.
// K runs over (segment, chunk of CGC channel groups, tap, group): per chunk the halo image of the
// tile is DMA'd once ([group][plane][pixel][16 B], conflict-free ds_read_b128) and double-buffered,
// the kh*kw taps are shifted windows of that image (im2col-free); weight stages (KS k-steps of 16)
// stream through a ring of R LDS slots, D = R-1 stages ahead of the MFMAs, one barrier per stage.
//
// Replaces tf.nn.conv2d + bias + batch_norm + activation (+ residual 1x1 conv, + pixel_norm,
// + nearest upsample, + channel concat) of tools_wscale/GAN.py:80-119,472-474,501-541 and
// GAN/multipassGAN-4x.py:505-526, GAN/multipassGAN-out.py:220-237,357 (reference tree).
#include <mutex>
#include <type_traits>

#include "mpgan_internal.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef MPG_AH
#define MPG_AH 2
#endif
// development switches of the F16F6 K loop (tools/build_variants.sh builds one library per setting, tools/probe_variants.py
// times them against each other on one box):
//   MPG_WD          bf6 weight planes read MPG_WD correction steps ahead of their MFMAs (MPG_WD + 1 register buffers)
//   MPG_DIAG6       timing-only builds (results are garbage): 1 = no correction phase at all, 2 = no block-scale / conversion
//                   VALU work (the bf6 operands are whatever the fp16 fragments hold), 4 = no image copies in the K loop,
//                   8 = no weight copies in the K loop
#ifndef MPG_WD
#define MPG_WD 1
#endif
// (measured and dropped in round 3, like in round 2: four cout tiles as 4 waves x (4 tile rows x 4 cout tiles), one wave per
// SIMD with the 256 accumulators in AccVGPRs: hipcc allocates 142 VGPRs + 256 AGPRs but keeps 1088 bytes of scratch per lane
// in the K loop -- 15.8 ms against 0.64 ms; that shape needs hand-allocated registers: profiles/r03/kloop_variants.md)
#ifndef MPG_DIAG6
#define MPG_DIAG6 0
#endif
//   MPG_W0 0        the weight planes of correction step 0 are read at the head of the correction phase, not in front of the
//                   last fp16 group
#ifndef MPG_W0
#define MPG_W0 1
#endif
//   MPG_PIECES_AFTER 0   the LDS-DMA pieces of an fp16 group are issued in front of the group's operand wait (rounds 2-3)
//   MPG_IMG_LATE 1  the image pieces of a stage may land during the next stage (measured slower: off)
#ifndef MPG_IMG_LATE
#define MPG_IMG_LATE 0
#endif
#ifndef MPG_PIECES_AFTER
#define MPG_PIECES_AFTER 1
#endif
//   MPG_ALT 1       experiment, off: the second half of the waves of an 8-wave block runs a stage's correction steps BEFORE its
//                   fp16 groups (complementary phases on a SIMD).  As compiled the corrections-first order keeps 265 spilled
//                   registers at four cout tiles (profiles/r03/kloop_variants.md): not measured on the hardware
#ifndef MPG_ALT
#define MPG_ALT 0
#endif
// (measured and dropped: the NT steps `w_lo6 x a_hi6` riding in the last NT fp16 groups of the stage, operands prefetched like
// the fp16 ones, only `w_hi6 x a_lo6` left as a separate phase: b1.B 650 us against 639, profiles/r03/kloop_variants.md)
//   MPG_STAMPS 1    diagnostic build only: every wave accumulates, over the stages of its K loop, the s_memtime cycles from
//                   the barrier release to (0) its first MFMA wait satisfied, (1) the end of the fp16 groups, (2) the end
//                   of the correction steps, (3) the release of the next barrier, and writes the four sums to
//                   y[(block * WAVES + wave) * 4 ..] when desc.reserved has bit 3 set (tools/probe_stamps.py)
#ifndef MPG_STAMPS
#define MPG_STAMPS 0
#endif
#if MPG_STAMPS
#define MPG_STAMP(v) asm volatile("s_memtime %0" : "=s"(v))
#else
#define MPG_STAMP(v)
#endif
// the a_hi correction step (0 .. NT-1) behind whose MFMAs the a_lo codes of tile row pt are made
constexpr int lo_step(int pt, int nt, int ptn) {
    const int s = nt - ptn + pt - 1;
    return s < 0 ? 0 : (s > nt - 1 ? nt - 1 : s);
}
constexpr int TW = 32;              // tile cols == MFMA N dimension
constexpr int TAPOFF_BYTES = 1024;  // 256 tap offsets

struct SegArgs {
    const char* x;        // G8 tensor
    const char* w;        // packed weights
    int cg_seg;           // channel groups consumed
    int cg_total, g_off;  // groups of the tensor, first group consumed
    int kh, kw, up;
    int cgc, nchunks, sc; // groups per chunk, chunks, weight stages per chunk
    int ih, iw;           // LDS image: (TH + kh - 1) x (TW + kw - 1) pixels
    int pt, pl;           // SAME padding before
    int hs, ws;           // source height / width (h >> up, w >> up)
    int np;               // pixels per image plane, padded to a multiple of 64
    int ni_img;           // image DMA instructions per thread per chunk
    int direct;           // F16F6, 1x1 over >= 2 groups: B fragments straight from memory, K runs over groups
    int tp;               // F16F6: tap slots per channel group (kh*kw, or rounded up to 8 when below 16)
    int pref;             // F16F6: the B fragments of stage st + 1 may be read during stage st (seg_shape_f6)
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
    float* y;             // fp32 NHWC output or null
    char* y_g8;           // G8 output (planes hi16, lo16) or null
    const float* in_amax; // inputs were multiplied by pow2_scale(*in_amax): the accumulators are divided by it
    const char* zeros;    // >= 16 zero bytes (source of out-of-image pixels)
    int img_bytes;        // bytes of one LDS image buffer (max over segments)
    int tap_bytes;        // F16F6: bytes of the tap-offset table at the start of LDS
    int tiles_x, tiles_y;
    int dbg;              // development probes: 1 skip K loop, 2 skip stores
};

typedef const __attribute__((address_space(4))) ConvArgs* KArgs;

// Per (NT, PREC) pipeline shape (host mirror: pipe_shape()).
template <int NT, int PREC>
struct Pipe {
    static constexpr int NPL = (PREC == 3) ? 2 : 1;
    static constexpr int PT = (NT >= 3) ? 2 : 4;       // pixel tiles (tile rows) per wave
    static constexpr int TH = 4 * PT;                  // tile rows per workgroup
    static constexpr int KS = (PREC == 3) ? ((NT == 4 || NT == 2) ? 1 : 2) : ((NT == 4 || NT == 2) ? 2 : 4);
    static constexpr int R = (NT == 3) ? 3 : 4;
    static constexpr int D = R - 1;
    static constexpr int WPLANE = KS * NT * 1024;
    static constexpr int WSTAGE = WPLANE * NPL;
    static constexpr int NI = WSTAGE / 4096;
    static_assert(WSTAGE % 4096 == 0, "stage must be a whole number of 256 x 16-byte pieces");
};

template <int N>
__device__ __forceinline__ void wait_dma_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
// ... with `extra` (0..7, wave-uniform) more of the newest operations allowed in flight
__device__ __forceinline__ void wait_dma_rt(int base, int extra) {
    switch (base + extra) {
#define MPG_W(n) case n: wait_dma_and_barrier<n>(); break;
        MPG_W(0) MPG_W(1) MPG_W(2) MPG_W(3) MPG_W(4) MPG_W(5) MPG_W(6) MPG_W(7) MPG_W(8) MPG_W(9) MPG_W(10) MPG_W(11)
        MPG_W(12) MPG_W(13) MPG_W(14) MPG_W(15)
#undef MPG_W
        default: wait_dma_and_barrier<0>(); break;
    }
}

__device__ __forceinline__ void dma16_stream(const char* src, char* lds_wave_base) {
    // same, with the non-temporal hint: activation tiles are read once or twice and should not push the weights
    // (re-read by every tile) out of L2
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

__device__ __forceinline__ void dma16(const char* src, char* lds_wave_base) {
    // lane l of the wave copies 16 bytes from its own `src` to lds_wave_base + 16*l
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int NT, int PT>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[PT][NT], const KArgs ap, char* smem, int n, int y0, int x0,
                                              int wave, int lane) {
    const auto& a = *ap;
    const int r = lane & 31;
    const int hh = lane >> 5;
    // accumulator element i of n-tile nt: output channel nt*32 + 8*(i>>2) + 4*hh + (i&3), pixel r.
    constexpr int ROWF = NT * 32 + 4;
    float* stg = reinterpret_cast<float*>(smem + TAPOFF_BYTES) + wave * (32 * ROWF);
    const int cg_out = (a.cout + 7) >> 3;
    const float unscale = a.in_amax != nullptr ? 1.f / mpg::pow2_scale(*a.in_amax) : 1.f;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int py = y0 + PT * wave + pt;
        float ss = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int co0 = nt * 32 + 8 * q4 + 4 * hh;
                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.bias != nullptr) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (co0 + i < a.cout) b4[i] = a.bias[co0 + i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = mpg::apply_act(acc[pt][nt][4 * q4 + i] * unscale + b4[i], a.act, a.leak);
                    if (co0 + i >= a.cout) v = 0.f;
                    acc[pt][nt][4 * q4 + i] = v;
                    ss += v * v;
                }
            }
        if (a.pn) {
            ss += __shfl_xor(ss, 32);
            const float sc = rsqrtf(ss / (float)a.cout + a.pn_eps);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[pt][nt][i] *= sc;
        }
        if (a.y == nullptr && a.post_add == nullptr) {
            // G8 output only (every launch between two fused convolutions): no LDS staging.  An accumulator quad holds
            // channels 8 q + 4 hh .. + 3 of pixel r, i.e. the two lanes (r, hh = 0 / 1) share every 8-channel group.
            // v_permlane32_swap trades the upper half-wave of one quad register with the lower half-wave of another:
            // after four swaps lane (r, 0) holds all 8 channels of group gA and lane (r, 1) all 8 of group gB, ready
            // to be split into the hi / lo planes and stored as 512-byte runs per plane and half-wave.
            const int npx = min(32, a.w - x0);
            if (py < a.h && r < npx && !(a.dbg & 2)) {
                const size_t plane_px = (size_t)a.h * a.w;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
                            const v2u_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[pt][nt][8 * qp + i]),
                                                                              __float_as_uint(acc[pt][nt][8 * qp + 4 + i]), false, false);
                            v[i] = __uint_as_float(sw[0]);
                            v[4 + i] = __uint_as_float(sw[1]);
                        }
                        const int cg = nt * 4 + 2 * qp + hh;
                        if (cg < cg_out) {
                            char* dst = a.y_g8 + ((((size_t)n * cg_out + cg) * 2) * plane_px + (size_t)py * a.w + x0 + r) * 16;
                            half8 hi, lo;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                hi[j] = (_Float16)v[j];
                                lo[j] = (_Float16)(v[j] - (float)hi[j]);
                            }
                            __builtin_nontemporal_store(hi, reinterpret_cast<half8*>(dst));
                            __builtin_nontemporal_store(lo, reinterpret_cast<half8*>(dst + plane_px * 16));
                        }
                    }
            }
            continue;
        }
        // stage this wave's 32 pixels x cout through LDS ([pixel][cout] rows padded by 16 B); the 32
        // pixels of a tile row are contiguous in every output layout, so all stores are whole runs
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int co0 = nt * 32 + 8 * q4 + 4 * hh;
                *reinterpret_cast<float4*>(stg + r * ROWF + co0) =
                    make_float4(acc[pt][nt][4 * q4], acc[pt][nt][4 * q4 + 1], acc[pt][nt][4 * q4 + 2], acc[pt][nt][4 * q4 + 3]);
            }
        if (py < a.h && !(a.dbg & 2)) {
            const int npx = min(32, a.w - x0);
            const size_t pix0 = ((size_t)n * a.h + py) * a.w + x0;
            if (a.post_add != nullptr) {
                // add into the staged tile first, so both output formats carry it
                const float* pa = a.post_add + pix0 * a.pa_stride + a.pa_coff;
                for (int f = lane; f < npx * a.cout; f += 64) {
                    const int p = f / a.cout;
                    const int c = f - p * a.cout;
                    stg[p * ROWF + c] += pa[(size_t)p * a.pa_stride + c];
                }
            }
            if (a.y != nullptr) {
                float* dst = a.y + pix0 * a.cout;
                const int total = npx * a.cout;
                if ((a.cout & 3) == 0) {
                    for (int f = lane * 4; f < total; f += 256) {
                        const int p = f / a.cout;
                        const int c = f - p * a.cout;
                        *reinterpret_cast<float4*>(dst + f) = *reinterpret_cast<const float4*>(stg + p * ROWF + c);
                    }
                } else {
                    for (int f = lane; f < total; f += 64) {
                        const int p = f / a.cout;
                        dst[f] = stg[p * ROWF + (f - p * a.cout)];
                    }
                }
            }
            if (a.y_g8 != nullptr) {
                // lane (pixel r, half hh) converts channel group 2 i + hh and writes BOTH of its planes (hi16, lo16): no
                // divergence between the halves, 512-byte runs per plane and half-wave; streamed (read once or twice by
                // the next launch), so the stores do not push the weights out of L2
                const size_t plane_px = (size_t)a.h * a.w;
                if (r < npx) {
                    for (int cg = hh; cg < cg_out; cg += 2) {
                        const float4 v0 = *reinterpret_cast<const float4*>(stg + r * ROWF + cg * 8);
                        const float4 v1 = *reinterpret_cast<const float4*>(stg + r * ROWF + cg * 8 + 4);
                        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                        char* dst = a.y_g8 + ((((size_t)n * cg_out + cg) * 2) * plane_px + (size_t)py * a.w + x0 + r) * 16;
                        half8 hi, lo;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            hi[j] = (_Float16)v[j];
                            lo[j] = (_Float16)(v[j] - (float)hi[j]);
                        }
                        __builtin_nontemporal_store(hi, reinterpret_cast<half8*>(dst));
                        __builtin_nontemporal_store(lo, reinterpret_cast<half8*>(dst + plane_px * 16));
                    }
                }
            }
        }
    }
}

template <int NT, int PREC>
__global__ __launch_bounds__(256, (NT >= 2 ? 2 : 3)) void conv_mfma_kernel(const ConvArgs a_unused) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KArgs ap = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const auto& a = *ap;
    using P = Pipe<NT, PREC>;
    constexpr int PT = P::PT, TH = P::TH, KS = P::KS, WPLANE = P::WPLANE, WSTAGE = P::WSTAGE;
    constexpr int NI = P::NI, R = P::R, D = P::D;

    int* tapoff = reinterpret_cast<int*>(smem);
    char* img_lds = smem + TAPOFF_BYTES;              // two image buffers
    char* w_lds = img_lds + 2 * a.img_bytes;          // ring of R stage slots

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r = lane & 31;
    const int hh = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD
    // a contiguous run of tiles => neighbouring halos hit the same L2.
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int tx = bid % a.tiles_x;
    const int t2 = bid / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    f32x16 acc[PT][NT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pt][nt][i] = 0.f;

    for (int s = 0; s < ((a.dbg & 1) ? 0 : a.nseg); ++s) {
        const auto& sg = ap->seg[s];
        const int CGC = sg.cgc;
        const int TG = sg.kh * sg.kw * CGC;
        const int plane_b = sg.np * 16;             // bytes of one image plane
        const int group_b = plane_b * 2;            // hi + lo
        const int ppg = sg.np >> 6;                 // 1-KiB pieces per plane

        // tap/group -> LDS byte offset inside an image buffer
        for (int q = tid; q < sg.sc * KS * 2; q += 256) {
            int off = 0;
            if (q < TG) {
                const int tap = q / CGC;
                const int g = q - tap * CGC;
                const int dy = tap / sg.kw;
                const int dx = tap - dy * sg.kw;
                off = (dy * sg.iw + dx) * 16 + g * group_b;
            }
            tapoff[q] = off;
        }
        int pixb[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) pixb[pt] = ((PT * wave + pt) * sg.iw + r) * 16;

        // ---- image DMA: piece pc = 4*i + wave covers 64 pixels of one plane of one group ----
        const size_t plane_px = (size_t)sg.hs * sg.ws;
        auto dma_image = [&](int chunk) {
            char* buf = img_lds + (chunk & 1) * a.img_bytes;
            for (int i = 0; i < sg.ni_img; ++i) {
                const int pc = 4 * i + wave_u;                 // piece of this wave
                const int g = pc / (2 * ppg);                  // group within the chunk
                const int rem = pc - g * 2 * ppg;
                const int pl = rem / ppg;                      // plane: 0 hi, 1 lo
                const int p = (rem - pl * ppg) * 64 + lane;    // pixel of the halo image
                const int hy = p / sg.iw;
                const int hx = p - hy * sg.iw;
                const int yy = y0 - sg.pt + hy;
                const int xx = x0 - sg.pl + hx;
                const int grp = chunk * CGC + g;
                const char* src = a.zeros;
                if (g < CGC && grp < sg.cg_seg && hy < sg.ih && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w &&
                    (PREC == 3 || pl == 0))
                    src = sg.x + ((((size_t)n * sg.cg_total + sg.g_off + grp) * 2 + pl) * plane_px +
                                  (size_t)(yy >> sg.up) * sg.ws + (xx >> sg.up)) * 16;
                dma16(src, buf + pc * 1024);
            }
        };
        // ---- weight DMA: stage -> ring slot, linear copy ----
        const int total_stages = sg.nchunks * sg.sc;
        auto dma_stage = [&](int stage) {
            const int sidx = stage < total_stages ? stage : total_stages - 1;   // tail: harmless re-read
            const char* src = sg.w + (size_t)sidx * WSTAGE + tid * 16;
            char* dst = w_lds + (stage % R) * WSTAGE + wave_u * 1024;
#pragma unroll
            for (int i = 0; i < NI; ++i) dma16(src + i * 4096, dst + i * 4096);
        };

        dma_image(0);
#pragma unroll
        for (int d = 0; d < D; ++d) dma_stage(d);

        for (int ch = 0; ch < sg.nchunks; ++ch) {
            // short chunks: the image of this chunk was issued fewer than D-1 stages ago
            if (sg.sc < D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* img = img_lds + (ch & 1) * a.img_bytes;
            for (int st = 0; st < sg.sc; ++st) {
                const int gst = ch * sg.sc + st;
                // stage gst (and everything older, incl. this chunk's image) has landed; all waves are
                // done with stage gst-1 and, at st == 0, with the previous chunk's image
                wait_dma_and_barrier<(D - 1) * NI>();
                if (st == 0 && ch + 1 < sg.nchunks) dma_image(ch + 1);
                dma_stage(gst + D);
                const char* wb = w_lds + (gst % R) * WSTAGE;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int toff = tapoff[2 * (st * KS + ks) + hh];
                    half8 b_hi[PT], b_lo[PT], a_hi[NT], a_lo[NT];
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        b_hi[pt] = *reinterpret_cast<const half8*>(img + pixb[pt] + toff);
                        if (PREC == 3) b_lo[pt] = *reinterpret_cast<const half8*>(img + plane_b + pixb[pt] + toff);
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
                        for (int pt = 0; pt < PT; ++pt) {
                            acc[pt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[nt], b_hi[pt], acc[pt][nt], 0, 0, 0);
                            if (PREC == 3) {
                                acc[pt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[nt], b_hi[pt], acc[pt][nt], 0, 0, 0);
                                acc[pt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[nt], b_lo[pt], acc[pt][nt], 0, 0, 0);
                            }
                        }
                }
            }
        }
        // drain the tail re-reads before the buffers (or the epilogue staging) are reused
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    conv_epilogue<NT, PT>(acc, ap, smem, n, y0, x0, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// MPG_PREC_F16F6: one fp16 product a_hi*w_hi plus the two correction products a_lo*w_hi and a_hi*w_lo as
// block-scaled bf6 (e3m2) products: v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 3 runs K = 64 in the
// 32 cycles of ONE fp16 32x32x16 (K = 16), so the two corrections cost half an fp16 product together:
// 1.5 fp16-equivalent matrix units per MAC (the fp8 form of round 1-2 cost 2: a mixed or fp8 operand pair
// runs at half this rate; tools/probes/probe_bf6.hip).
//
// Activations are read in the exact G8 flavour (hi16, lo16).  The bf6 operands of a lane -- the 32 K values
// (4 tap slots x 8 channels) it holds for the K = 64 instruction -- are made in registers from the very fp16
// fragments the fp16 products use: v_cvt_scalef32_pk32_bf6_f16 converts 32 values in one instruction with a
// power-of-two block scale, which is PER LANE here: 2^-18 times the binade of the largest |a_hi| among the
// lane's 32 values (three-input packed max / min trees), and 2^-12 of that for the a_lo block.  The E8M0
// bytes of both go to the MFMA's scale operand: true MX block scaling, no per-tensor exponent, nothing the
// producer has to know -- an activation tensor of any range and any mix of channel scales keeps the
// corrections (the fixed exponents of the fp8 flavour lost them outside |v| in 1e-2..112).
// Weights: bf6 planes packed per stage with one E8M0 byte per (output channel, K block of 32) and plane.
//
// One workgroup = WAVES waves = 16 tile rows x 32 pixels; a weight stage is one macro-step of 8 tap slots
// (K = 64): [4 fp16 k-steps][w_hi6][w_lo6].  The slots of a segment form one stream over its channel groups
// (slot = group * tp + tap), so a stage may end one group and begin the next; every group has its own LDS image.
// ---------------------------------------------------------------------------------------------
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
// LDS reads in the K loops use clang vector types only: a read through HIP's struct `int4` makes the compiler put an
// `s_waitcnt vmcnt(0)` in front of it while LDS-DMA pieces are in flight (it cannot tell the read from the DMA's
// destination), which serialises every stage behind its own weight / image DMAs; ext_vector_type reads do not.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));

// 32 bytes of LDS as two 16-byte reads ([half][lane][16 B]: consecutive lanes read consecutive 16-byte words,
// which ds_read_b128 serves without bank conflicts; a [lane][32 B] layout is 2-way conflicted)
__device__ __forceinline__ v8i lds_read32(const char* p) {
    const v4i lo = *reinterpret_cast<const v4i*>(p);
    const v4i hi = *reinterpret_cast<const v4i*>(p + 1024);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- explicit LDS reads / counted waits ----
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// The reads and waits are volatile asm: they stay in program order, which is what the counted waits count.
template <int OFF, class T>
__device__ __forceinline__ void ds_read16(T& dst, unsigned addr) {
    static_assert(sizeof(T) == 16 && OFF >= 0 && OFF < 65536, "one ds_read_b128");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// wait until at most N of the LDS reads issued so far are outstanding (they return in order)
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
}
// no instruction: makes every later use of `frag` depend on the preceding (volatile) wait
template <class T>
__device__ __forceinline__ void tie(T& frag) {
    asm volatile("" : "+v"(frag));
}

// bookkeeping of the explicit schedule (all compile-time): fp16 group h = (k-step h / NT, cout tile h % NT) issues
// [PT B fragments when h % NT == 0] + 1 A fragment
constexpr int kx_cum(int h, int NT, int PT) { return h + PT * ((h + NT - 1) / NT); }
// reads that may still be outstanding when group g's MFMAs start: everything issued after group g's own fragments
constexpr int kx_allowed(int g, int NT, int PT, int AH) {
    const int G16 = 4 * NT;
    const int hi = g + AH + 1 < G16 ? g + AH + 1 : G16;
    return kx_cum(hi, NT, PT) - kx_cum(g + 1, NT, PT);
}

__device__ __forceinline__ half32 cat32(const half8& a, const half8& b, const half8& c, const half8& d) {
    const half16 lo = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const half16 hi = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23,
                                   24, 25, 26, 27, 28, 29, 30, 31);
}

// biased fp16 exponent (0..30) of the largest |x| among the 32 halves of a lane: packed three-input max and min
// trees (8 + 8 instructions; v_pk_maximum3_f16 has no |x| modifier), max(max, -min), the larger half, its exponent
__device__ __forceinline__ int block_exp16(const half32& v) {
    const v16i r = __builtin_bit_cast(v16i, v);
    int a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7, m2, m1;
#define MPG_MAX3(d, x, y, z) asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z))
#define MPG_MIN3(d, x, y, z) asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z))
    MPG_MAX3(a0, r[0], r[1], r[2]); MPG_MAX3(a1, r[3], r[4], r[5]); MPG_MAX3(a2, r[6], r[7], r[8]); MPG_MAX3(a3, r[9], r[10], r[11]);
    MPG_MAX3(a4, r[12], r[13], r[14]); MPG_MAX3(a5, a0, a1, r[15]); MPG_MAX3(a6, a2, a3, a4); MPG_MAX3(a7, a5, a6, a6);
    MPG_MIN3(b0, r[0], r[1], r[2]); MPG_MIN3(b1, r[3], r[4], r[5]); MPG_MIN3(b2, r[6], r[7], r[8]); MPG_MIN3(b3, r[9], r[10], r[11]);
    MPG_MIN3(b4, r[12], r[13], r[14]); MPG_MIN3(b5, b0, b1, r[15]); MPG_MIN3(b6, b2, b3, b4); MPG_MIN3(b7, b5, b6, b6);
#undef MPG_MAX3
#undef MPG_MIN3
    asm("v_pk_max_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(m2) : "v"(a7), "v"(b7));
    asm("v_pk_max_f16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m1) : "v"(m2));
    return (m1 >> 10) & 31;
}

// bf6 operands of a lane's 32 activation values.  With e = block_exp16 (every |a_hi| < 2^(e-14)):
//   hi block: codes of a_hi * 2^(18-e)  (< 16; the e3m2 range ends at 28), E8M0 byte e + 109
//   lo block: codes of a_lo * 2^(30-e)  (|a_lo| <= half an ulp of a_hi <= 2^(e-26)), E8M0 byte e + 97
// v_cvt_scalef32_pk32_bf6_f16 divides by its f32 scale operand (a power of two), rounds to nearest even and saturates.
#ifndef MPG_CVT_DIVIDES
#define MPG_CVT_DIVIDES 1
#endif
__device__ __forceinline__ float pow2_from_byte(int e8m0) {
#if MPG_CVT_DIVIDES
    return __builtin_bit_cast(float, e8m0 << 23);
#else
    return __builtin_bit_cast(float, (254 - e8m0) << 23);
#endif
}
__device__ __forceinline__ v8i widen6(const v6i& v) {
    return __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, -1, -1);
}
__device__ __forceinline__ v8i bf6_of(const half32& v, int e8m0) {
    return widen6(__builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, pow2_from_byte(e8m0)));
}
constexpr int BF6 = 3;     // cbsz / blgp code of e3m2

template <int NT>
struct Pipe6 {
    static constexpr int WAVES = (NT == 1) ? 4 : 8;
    static constexpr int PT = 16 / WAVES;                  // tile rows per wave
    static constexpr int TH = 16;
    static constexpr int WF16 = 4 * NT * 1024;             // four fp16 k-steps
    static constexpr int WF6 = NT * 2048;                  // one bf6 plane: NT x [2 halves][64 lanes][16 B]: 24 B of codes, scales, pad
    static constexpr int WSTAGE = WF16 + 2 * WF6;          // 8 * NT KiB
    static constexpr int R = 3;
    static constexpr int D = R - 1;
    static constexpr int NI = WSTAGE / (WAVES * 1024);
    static_assert(WSTAGE % (WAVES * 1024) == 0, "stage must be a whole number of per-wave pieces");
};

template <int NT>
__global__ __launch_bounds__(Pipe6<NT>::WAVES * 64, 2) void conv_mfma_f6_kernel(const ConvArgs a_unused) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KArgs ap = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const auto& a = *ap;
    using P = Pipe6<NT>;
    constexpr int WAVES = P::WAVES, PT = P::PT, TH = P::TH, WF16 = P::WF16, WF6 = P::WF6, WSTAGE = P::WSTAGE;
    constexpr int NI = P::NI, R = P::R, D = P::D, THREADS = WAVES * 64;

    int* tap16 = reinterpret_cast<int*>(smem);
    char* img_lds = smem + a.tap_bytes;
    char* w_lds = img_lds + 2 * a.img_bytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r = lane & 31;
    const int hh = lane >> 5;

    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int tx = bid % a.tiles_x;
    const int t2 = bid / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    f32x16 acc[PT][NT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pt][nt][i] = 0.f;

    // The K segments are independent partial sums.  Blocks that share a CU (workgroups go round-robin over the 8
    // XCDs, then over the 32 CUs of an XCD: co-resident blocks differ in bit 8 of the id) walk them in opposite
    // orders, so one block's HBM-bound direct 1x1 segment runs under the other's LDS / MFMA-bound 5x5 segment.
    const int seg_flip = (NT == 1 && a.nseg > 1) ? ((int)(blockIdx.x >> 8) & 1) : 0;
    for (int s0 = 0; s0 < ((a.dbg & 1) ? 0 : a.nseg); ++s0) {
        const int s = seg_flip ? a.nseg - 1 - s0 : s0;
        const auto& sg = ap->seg[s];
        if (NT <= 2 && sg.direct) {
            // 1x1 segment over cg_seg >= 2 channel groups: no halo, so no LDS image.  The G8 rows are already
            // fragment-shaped (16 B per pixel and group): lane (pixel r, half hh) loads its B operands from
            // memory, a weight stage is one macro-step of 8 GROUPS (K = 64) instead of 8 taps.  Rows / columns
            // past the image edge re-read the last valid pixel; their outputs are never stored.
            const unsigned plane_bytes = (unsigned)(sg.hs * sg.ws) * 16u;   // host: 16 planes < 2^31 bytes
            const unsigned gstride = 2u * plane_bytes;
            const char* xb = sg.x + ((size_t)n * sg.cg_total + sg.g_off) * gstride;   // uniform
            unsigned pixo[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                int yy = y0 + PT * wave + pt, xx = x0 + r;
                yy = (yy < a.h ? yy : a.h - 1) >> sg.up;
                xx = (xx < a.w ? xx : a.w - 1) >> sg.up;
                pixo[pt] = (unsigned)(yy * sg.ws + xx) * 16u;
            }
            const int glast = sg.cg_seg - 1;
            auto dma_stage_d = [&](int stage) {
                const int sidx = stage < sg.sc ? stage : sg.sc - 1;
                const char* src = sg.w + (size_t)sidx * WSTAGE + tid * 16;
                char* dst = w_lds + (stage % R) * WSTAGE + wave_u * 1024;
#pragma unroll
                for (int i = 0; i < NI; ++i) dma16(src + i * (THREADS * 16), dst + i * (THREADS * 16));
            };
#pragma unroll
            for (int d = 0; d < D; ++d) dma_stage_d(d);
            for (int st = 0; st < sg.sc; ++st) {
                wait_dma_and_barrier<(D - 1) * NI>();
                dma_stage_d(st + D);
                const char* wb = w_lds + (st % R) * WSTAGE;
                const char* xs = xb + (size_t)st * 8 * gstride;   // uniform: first group of this macro-step
                const int grem = glast - st * 8;
                // one (four tile rows per wave) or two tile rows at a time: all 32 operand fragments of a macro-step would not
                // fit next to the accumulators (the weights are re-read from LDS for every part)
                constexpr int PH = PT == 4 ? 1 : 2;
                static_for<0, PT / PH>([&](auto hc) {
                    constexpr int p0 = decltype(hc)::value * PH;
                    half8 b_hi[4][PH], b_lo[4][PH];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int g = 2 * j + hh;
                        g = g < grem ? g : grem;                 // groups past the segment: zero weights
#pragma unroll
                        for (int q = 0; q < PH; ++q) {
                            b_hi[j][q] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(xs + (pixo[p0 + q] + g * gstride)));   // read once
                            b_lo[j][q] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(xs + (pixo[p0 + q] + g * gstride + plane_bytes)));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const half8 a_hi = *reinterpret_cast<const half8*>(wb + ((j * NT + nt) * 64 + lane) * 16);
#pragma unroll
                            for (int q = 0; q < PH; ++q)
                                acc[p0 + q][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi[j][q], acc[p0 + q][nt], 0, 0, 0);
                        }
                    }
                    v8i hi6[PH], lo6[PH];
                    int sb[PH];
#pragma unroll
                    for (int q = 0; q < PH; ++q) {
                        const half32 bh = cat32(b_hi[0][q], b_hi[1][q], b_hi[2][q], b_hi[3][q]);
                        const half32 bl = cat32(b_lo[0][q], b_lo[1][q], b_lo[2][q], b_lo[3][q]);
                        const int e = block_exp16(bh);
                        hi6[q] = bf6_of(bh, e + 109);
                        lo6[q] = bf6_of(bl, e + 97);
                        sb[q] = (e + 109) | (e + 97) << 8;
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const v8i w_hi = lds_read32(wb + WF16 + nt * 2048 + lane * 16);
                        const v8i w_lo = lds_read32(wb + WF16 + WF6 + nt * 2048 + lane * 16);
#pragma unroll
                        for (int q = 0; q < PH; ++q) {
                            acc[p0 + q][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w_lo, hi6[q], acc[p0 + q][nt], BF6, BF6, 1, w_lo[6], 0, sb[q]);
                            acc[p0 + q][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w_hi, lo6[q], acc[p0 + q][nt], BF6, BF6, 0, w_hi[6], 1, sb[q]);
                        }
                    }
                });
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            continue;
        }
        // K is ONE stream of tap slots over the channel groups of the segment: slot q = (group q / tp, tap q % tp),
        // eight slots per weight stage, so a stage may finish one group and start the next (25 taps x 16 groups =
        // 50 full stages instead of 16 x 4 with 7 empty slots each).  Group g's halo image lives in LDS buffer
        // g & 1; the offset table carries the buffer with the tap.
        const int T = sg.kh * sg.kw;
        const int G = sg.nchunks;
        const int NS = sg.sc;
        const int plane_b = sg.np * 16;
        const int ppg = sg.np >> 6;

        for (int q = tid; q < NS * 8; q += THREADS) {
            const int g = q / sg.tp;
            const int t = q - g * sg.tp;
            // Padding slots (t >= T, or past the last group) have zero weights but their pixels still enter the lane's
            // block maximum, i.e. the scale of the REAL values of the block: they must read stable data.  They read tap
            // (0, 0) of the image of the group they pad -- resident for the whole stage -- never the other buffer, which
            // may be receiving the next group's image by DMA at that moment (run-to-run differences in the last bits).
            int off = ((g < G ? g : G - 1) & 1) * a.img_bytes;
            if (g < G && t < T) {
                const int dy = t / sg.kw;
                const int dx = t - dy * sg.kw;
                off += (dy * sg.iw + dx) * 16;
            }
            tap16[(q >> 3) * 8 + (q & 1) * 4 + ((q & 7) >> 1)] = off;             // [stage][half][k-step]
        }
        int pixb[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) pixb[pt] = ((PT * wave + pt) * sg.iw + r) * 16;

        const size_t plane_px = (size_t)sg.hs * sg.ws;
        // Every group's image has the same per-lane source offsets (only the group's base address differs): they are
        // worked out once per segment, so that inside the stage loop an image piece costs a select and one DMA
        // instruction.  -1: the pixel lies outside the image (or past the halo rows): it is fetched from the zero page.
        constexpr int MAXI = (WAVES == 8) ? 4 : 7;       // pieces per wave and image: 2 planes x <= 14 KiB over WAVES waves
        int img_src[MAXI];
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int pc = WAVES * i + wave_u;
            const int pl = pc / ppg;
            const int p = (pc - pl * ppg) * 64 + lane;
            const int hy = p / sg.iw;
            const int hx = p - hy * sg.iw;
            const int yy = y0 - sg.pt + hy;
            const int xx = x0 - sg.pl + hx;
            const bool ok = pl < 2 && hy < sg.ih && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
            img_src[i] = ok ? (int)(((size_t)pl * plane_px + (size_t)(yy >> sg.up) * sg.ws + (xx >> sg.up)) * 16) : -1;
        }
        const size_t group_bytes = 2 * plane_px * 16;
        const char* const x_first = sg.x + ((size_t)n * sg.cg_total + sg.g_off) * group_bytes;   // uniform
        // piece i (compile-time) of the image of channel group `chunk` (uniform)
        auto img_piece = [&](int chunk, auto ic) {
            constexpr int i = decltype(ic)::value;
            if ((MPG_DIAG6 & 4) && chunk > 0) return;          // timing only: no image copies in the K loop
            if (i < sg.ni_img) {
                const int off = img_src[i];
                const char* src = (off >= 0 && chunk < sg.cg_seg) ? x_first + (size_t)chunk * group_bytes + off : a.zeros;
                dma16_stream(src, img_lds + (chunk & 1) * a.img_bytes + (WAVES * i + wave_u) * 1024);
            }
        };
        auto w_piece = [&](int stage, auto ic) {
            constexpr int i = decltype(ic)::value;
            const int sidx = stage < NS ? stage : NS - 1;
            if ((MPG_DIAG6 & 8) && stage >= D) return;         // timing only: no weight copies in the K loop
            dma16(sg.w + (size_t)sidx * WSTAGE + tid * 16 + i * (THREADS * 16),
                  w_lds + (stage % R) * WSTAGE + wave_u * 1024 + i * (THREADS * 16));
        };
        static_for<0, MAXI>([&](auto ic) { img_piece(0, ic); });
#pragma unroll
        for (int d = 0; d < D; ++d) static_for<0, NI>([&](auto ic) { w_piece(d, ic); });
        int g_next = 1;                                  // next group image to fetch

        // B fragments of a stage: lane (pixel r of tile row pt, half hh) holds, for k-step j, the 8 channels of slot
        // 2 j + hh: bh[pt][j] is both the fp16 B operand of k-step j and a quarter of the lane's bf6 block.
        half8 bh[PT][4];
        v4i o16n = {0, 0, 0, 0};
        const unsigned i_base = lds_off(img_lds);
#if MPG_STAMPS
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts3_prev = 0;
        unsigned sum_head = 0, sum_f16 = 0, sum_f6 = 0, sum_bar = 0;
#endif

        auto run_stages = [&](auto oc) {
        constexpr bool CORR_FIRST = decltype(oc)::value != 0;
        int img_in_flight = 0;      // image pieces of the previous stage that may still be in flight behind this barrier
        for (int st = 0; st < NS; ++st) {
            // stage st (and everything older, incl. the images issued before it) has landed; all waves are done
            // with stage st-1.  (MPG_IMG_LATE: the image pieces of stage st-1 -- in front of its weight pieces in issue
            // order, so `vmcnt` can count them with the weights -- may stay in flight when their group is first read two
            // or more stages later.  Measured: b1.B 652 against 636 us, b2.A 229 against 231: off.)
            if constexpr (MPG_IMG_LATE) wait_dma_rt((D - 1) * NI, img_in_flight);
            else wait_dma_and_barrier<(D - 1) * NI>();
#if MPG_STAMPS
            if (st > 0) {       // the barrier's lgkmcnt(0) completed every stamp of the previous stage
                sum_head += (unsigned)(ts1 - ts0);
                sum_f16 += (unsigned)(ts2 - ts1);
                sum_f6 += (unsigned)(ts3 - ts2);
                if (st > 1) sum_bar += (unsigned)(ts0 - ts3_prev);
                ts3_prev = ts3;
            }
            MPG_STAMP(ts0);
#endif
            // the tap table is constant over the segment: stage st + 1's entry is read at the head of stage st
            if (st == 0) {
                ds_read16<0>(o16n, lds_off(tap16 + hh * 4));
                lgkm_wait<0>();
            }
            tie(o16n);
            const v4i o16 = o16n;
            const int to16[4] = {o16.x, o16.y, o16.z, o16.w};
            {
                const int sn = st + 1 < NS ? st + 1 : st;
                ds_read16<0>(o16n, lds_off(tap16 + (sn * 2 + hh) * 4));
            }
            // image of group g_next goes into the buffer of group g_next - 2: free once no slot of this or a later
            // stage belongs to that group.  The LDS-DMA pieces of this stage are spread over its MFMA groups (one piece
            // at a time between the MFMAs) instead of being issued as a burst behind the barrier: a burst of WAVES x
            // (NI + image pieces) 1-KiB pieces queues up in the CU's address unit for ~1000-1500 cycles in which no
            // wave issues an MFMA (profiles/r02/kloop_analysis.md).  Order within the stage: the image pieces in the
            // first half of the groups, then the NI weight pieces of stage st + D, so `vmcnt((D-1) NI)` at the next
            // barrier still covers them.
            const bool do_img = g_next < G && st * 8 >= (g_next - 1) * sg.tp;
            const int img_chunk = g_next;
            // first stage that reads group g_next: the one holding slot g_next * tp
            img_in_flight = (MPG_IMG_LATE && do_img && (g_next * sg.tp) / 8 >= st + 2) ? sg.ni_img : 0;
            if (do_img) ++g_next;
            const char* wb = w_lds + (st % R) * WSTAGE;
            const unsigned a_base = lds_off(wb) + (unsigned)lane * 16u;
            constexpr int AH = MPG_AH, G16 = 4 * NT;
            half8 aq[AH + 1];
            v8i hi6[PT], lo6[PT];
            int sb[PT], e16[PT];
            // fp16 group g = (k-step g / NT, cout tile g % NT): its reads are [the PT B fragments of the k-step when
            // g % NT == 0] + one A fragment, issued AH groups ahead of its MFMAs
            auto read_group = [&](auto gc) {
                constexpr int g = decltype(gc)::value, j = g / NT;
                if constexpr (g % NT == 0)
                    static_for<0, PT>([&](auto pc) {
                        constexpr int pt = decltype(pc)::value;
                        ds_read16<0>(bh[pt][j], i_base + (unsigned)(pixb[pt] + to16[j]));
                    });
                ds_read16<g * 1024>(aq[g % (AH + 1)], a_base);
            };
            auto make_hi6 = [&](auto pc) {
                constexpr int pt = decltype(pc)::value;
#if MPG_DIAG6 & 2
                const v4i q0 = __builtin_bit_cast(v4i, bh[pt][0]), q1 = __builtin_bit_cast(v4i, bh[pt][1]);
                hi6[pt] = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
                e16[pt] = 15;
                sb[pt] = 0x7f7f7f7f;
#else
                const half32 b32 = cat32(bh[pt][0], bh[pt][1], bh[pt][2], bh[pt][3]);
                e16[pt] = block_exp16(b32);
                hi6[pt] = bf6_of(b32, e16[pt] + 109);
                sb[pt] = (e16[pt] + 109) | (e16[pt] + 97) << 8;
#endif
            };
            constexpr int HALF = G16 / 2;
            constexpr int IPG = (MAXI + HALF - 1) / HALF;    // image pieces per group (first half of the groups)
            constexpr int WPG = (NI + HALF - 1) / HALF;      // weight pieces per group (second half)
            // bf6 weight planes of the correction steps: step k < NT reads w_lo6[k], step k >= NT w_hi6[k - NT]
            constexpr int KS6 = 2 * NT;                            // correction steps
            constexpr int WD = MPG_WD < KS6 - 1 ? MPG_WD : KS6 - 1;
            v4i wq[WD + 1][2];
            auto read_w6 = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int off = WF16 + (k < NT ? WF6 + k * 2048 : (k - NT) * 2048);
                ds_read16<off>(wq[k % (WD + 1)][0], a_base);
                ds_read16<off + 1024>(wq[k % (WD + 1)][1], a_base);
            };
            // ---- the fp16 product: G16 groups of PT MFMAs ----
            auto fp16_phase = [&](auto mk) {
                constexpr bool MAKE_HI6 = decltype(mk)::value != 0;
                // fp16 groups first, corrections behind them: the weight planes of correction step 0 are read in front of
                // the last group (the correction phase then starts on operands that are there: its head was ~200 cycles of
                // LDS latency, and the younger wave of a SIMD runs that phase alone)
                constexpr bool W0_AHEAD = MAKE_HI6 && MPG_W0;
                static_for<0, (AH < G16 ? AH : G16)>([&](auto gc) { read_group(gc); });
                static_for<0, G16>([&](auto gc) {
                    constexpr int g = decltype(gc)::value, j = g / NT, nt = g % NT;
                    // the LDS-DMA pieces of the group: a wave waits 60-185 cycles per piece for the CU's address unit.  Behind
                    // the group's MFMAs that wait runs under them; in front of the group (rounds 2-3) it delayed the group's
                    // own operand wait -- with one cout tile (three image pieces per group) the head of a stage was 830 cycles
                    auto pieces = [&]() {
                        if constexpr (g < HALF) {
                            if (do_img)
                                static_for<0, IPG>([&](auto kc) {
                                    constexpr int i = g * IPG + decltype(kc)::value;
                                    if constexpr (i < MAXI) img_piece(img_chunk, std::integral_constant<int, i>{});
                                });
                        } else {
                            static_for<0, WPG>([&](auto kc) {
                                constexpr int i = (g - HALF) * WPG + decltype(kc)::value;
                                if constexpr (i < NI) w_piece(st + D, std::integral_constant<int, i>{});
                            });
                        }
                    };
                    if constexpr (!MPG_PIECES_AFTER) pieces();
                    if constexpr (g + AH < G16) read_group(std::integral_constant<int, g + AH>{});
                    if constexpr (W0_AHEAD && g == G16 - 1) read_w6(std::integral_constant<int, 0>{});
                    lgkm_wait<kx_allowed(g, NT, PT, AH) + (W0_AHEAD && g == G16 - 1 ? 2 : 0)>();   // reads issued behind group g's own
                    if constexpr (g == 0 && MAKE_HI6) { MPG_STAMP(ts1); }
                    tie(aq[g % (AH + 1)]);
                    if constexpr (nt == 0)
                        static_for<0, PT>([&](auto pc) { tie(bh[decltype(pc)::value][j]); });
                    static_for<0, PT>([&](auto pc) {
                        constexpr int pt = decltype(pc)::value;
                        acc[pt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[g % (AH + 1)], bh[pt][j], acc[pt][nt], 0, 0, 0);
                    });
                    if constexpr (MPG_PIECES_AFTER) pieces();
                    // the a_hi block scales and codes (VALU work under the matrix pipe): tile row g - 3 NT behind each group
                    // of the last k-step, whatever is left behind the last group
                    if constexpr (MAKE_HI6) {
                        if constexpr (g >= 3 * NT && g - 3 * NT < PT) make_hi6(std::integral_constant<int, g - 3 * NT>{});
                        if constexpr (g == G16 - 1 && NT < PT) static_for<NT, PT>([&](auto pc) { make_hi6(pc); });
                    }
                });
            };
            // ---- the two bf6 corrections: step k < NT is w_lo6[k] x a_hi6, step k >= NT is w_hi6[k - NT] x a_lo6 ----
            // LDS reads in order: W(0), the a_lo fragments (into the registers of the a_hi ones, which the conversions
            // above have consumed), W(1) .. W(WD), then W(k + WD) ahead of step k.
            auto bf6_phase = [&](auto w0c) {
            constexpr bool W0_DONE = decltype(w0c)::value != 0;    // W(0) was read in front of the last fp16 group
#if !(MPG_DIAG6 & 1)
            constexpr int PB = PT;                                 // all a_lo fragments at once: they land in the a_hi registers
            auto read_bl = [&](auto pc) {
                constexpr int pt = decltype(pc)::value;
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    ds_read16<0>(bh[pt][j], i_base + (unsigned)(plane_b + pixb[pt] + to16[j]));
                });
            };
            auto make_lo6 = [&](auto pc) {
                constexpr int pt = decltype(pc)::value;
                static_for<0, 4>([&](auto jc) { tie(bh[pt][decltype(jc)::value]); });
#if MPG_DIAG6 & 2
                const v4i q0 = __builtin_bit_cast(v4i, bh[pt][0]), q1 = __builtin_bit_cast(v4i, bh[pt][1]);
                lo6[pt] = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
#else
                lo6[pt] = bf6_of(cat32(bh[pt][0], bh[pt][1], bh[pt][2], bh[pt][3]), e16[pt] + 97);
#endif
            };
            if constexpr (!W0_DONE) read_w6(std::integral_constant<int, 0>{});
            static_for<0, PB>([&](auto pc) { read_bl(pc); });
            static_for<1, WD + 1>([&](auto kc) { read_w6(kc); });
            static_for<0, KS6>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k >= 1 && k + WD < KS6) read_w6(std::integral_constant<int, k + WD>{});
                // reads issued behind W(k): at k = 0 the a_lo fragments and W(1 .. WD), later W(k + 1 .. k + WD)
                constexpr int ahead = (k + WD < KS6 ? k + WD : KS6 - 1) - k;
                lgkm_wait<(k == 0 ? 4 * PB : 0) + 2 * ahead>();
                tie(wq[k % (WD + 1)][0]);
                tie(wq[k % (WD + 1)][1]);
                const v8i w6 = __builtin_shufflevector(wq[k % (WD + 1)][0], wq[k % (WD + 1)][1], 0, 1, 2, 3, 4, 5, 6, 7);
                static_for<0, PT>([&](auto pc) {
                    constexpr int pt = decltype(pc)::value;
                    if constexpr (k < NT)
                        acc[pt][k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w6, hi6[pt], acc[pt][k], BF6, BF6, 1, w6[6], 0, sb[pt]);
                    else
                        acc[pt][k - NT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w6, lo6[pt], acc[pt][k - NT], BF6, BF6, 0, w6[6], 1, sb[pt]);
                });
                // the a_lo codes of tile row pt (VALU: block maximum, scale, 32-value conversion) behind the MFMAs of the a_hi
                // step lo_step(pt), as late as still finishes in front of the first a_lo step NT: the conversions of the
                // rows run under the matrix work of different steps instead of in one piece in front of step 1
                if constexpr (k < NT) {
                    if constexpr (k == 0 && lo_step(0, NT, PT) == 0) lgkm_wait<2 * ahead>();     // the a_lo fragments are there
                    static_for<0, PT>([&](auto pc) {
                        if constexpr (lo_step(decltype(pc)::value, NT, PT) == k) make_lo6(pc);
                    });
                }
            });
#else
            asm volatile("" ::"v"(hi6[0]), "v"(sb[0]), "v"(e16[0]));
#endif
            };
            // The two waves of a SIMD (wave w and w + WAVES / 2 of an 8-wave block) walk the stage in OPPOSITE orders: the
            // first half runs the matrix-dense fp16 groups first and the latency- and VALU-bound correction steps last,
            // the second half the other way round (all operands of a stage are there at its barrier).  Run in the same
            // order the older wave of a SIMD wins every arbitration, finishes after ~3300 cycles and idles, and the
            // younger one ends the stage alone in its correction steps at < 50 % matrix duty (profiles/r03/kloop_stamps.txt).
            if constexpr (CORR_FIRST) {
                static_for<0, PT>([&](auto pc) {
                    constexpr int pt = decltype(pc)::value;
                    static_for<0, 4>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        ds_read16<0>(bh[pt][j], i_base + (unsigned)(pixb[pt] + to16[j]));
                    });
                });
                lgkm_wait<0>();
                static_for<0, PT>([&](auto pc) {
                    static_for<0, 4>([&](auto jc) { tie(bh[decltype(pc)::value][decltype(jc)::value]); });
                    make_hi6(pc);
                });
                MPG_STAMP(ts1);
                bf6_phase(std::integral_constant<int, 0>{});
                MPG_STAMP(ts2);
                fp16_phase(std::integral_constant<int, 0>{});
            } else {
                fp16_phase(std::integral_constant<int, 1>{});
                MPG_STAMP(ts2);
                bf6_phase(std::integral_constant<int, MPG_W0>{});
            }
            MPG_STAMP(ts3);
        }
        };
        // one loop per order, chosen once per segment (a branch inside the stage body would merge the two register
        // allocations at every stage: 390 spilled registers when tried)
#if MPG_ALT
        if (WAVES == 8 && wave_u >= WAVES / 2) run_stages(std::integral_constant<int, 1>{});
        else
#endif
            run_stages(std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
#if MPG_STAMPS
        if ((a.dbg & 8) && a.y != nullptr && s == 0 && lane == 0) {
            unsigned* o = reinterpret_cast<unsigned*>(a.y) + ((size_t)blockIdx.x * WAVES + wave) * 4;
            o[0] = sum_head; o[1] = sum_f16; o[2] = sum_f6; o[3] = sum_bar;
        }
#endif
    }
    conv_epilogue<NT, PT>(acc, ap, smem, n, y0, x0, wave, lane);
}

// e3m2 code (sign, 3 exponent bits, bias 3, 2 mantissa bits; no infinities) of x, round to nearest even, saturating at 28
__device__ __forceinline__ int e3m2_encode(float x) {
    const int s = (__builtin_bit_cast(unsigned, x) >> 31) << 5;
    const float ax = fabsf(x);
    if (!(ax < 28.f)) return s | 31;
    const int e = ax >= 0.25f ? ilogbf(ax) : -2;          // the binade whose step is used; subnormals share the step of [0.25, 0.5)
    const float step = ldexpf(1.f, e - 2);
    const float v = rintf(ax / step) * step;              // may reach the next binade
    if (v == 0.f) return s;
    const int e2 = ilogbf(v);
    if (e2 < -2) return s | (int)(v * 16.f);             // subnormal: M * 2^-4
    return s | (e2 + 3) << 2 | (int)((v * ldexpf(1.f, -e2) - 1.f) * 4.f);
}

// F16F6 weight image: per stage (8 consecutive tap slots of the segment's slot stream; fold: 8 channel groups):
//   [4 k-steps][NT][64 lanes][8 x fp16]  |  w_hi6: [NT][2 halves][64 lanes][16 B]  |  w_lo6: same
// Lane (row r of cout tile nt, half hh) holds the 32 values (k-step j, element e) -> slot 2 j + hh, channel e: the fp16 A
// fragments of the four k-steps AND, in that order, the K block of 32 of the bf6 instruction.  A bf6 plane keeps the
// lane's 32 six-bit codes (value i at bits 6 i .. 6 i + 5 of 24 bytes) in bytes 0-15 of half 0 and 0-7 of half 1; bytes
// 8-11 of half 1 are the scale word {E8M0 of the w_hi block, E8M0 of the w_lo block, 0, 0} (in both planes), 12-15 zero.
// Block scale: 2^(floor(log2 max|v|) - 3), i.e. the largest code magnitude lies in [8, 16).
__global__ void pack_weights_f6_kernel(const float* __restrict__ w, int kh, int kw, int cin_total, int c_off,
                                       int cin, int cout, float wscale, const float* __restrict__ cscale,
                                       int NT, int sc, int fold, int tp, char* __restrict__ out) {
    // fold (direct 1x1 segments): the "taps" of a macro-step are 8 consecutive channel groups
    const int T = kh * kw;
    const long total = (long)sc * NT * 64;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int nt = (idx >> 6) % NT;
    const int st = (int)((idx >> 6) / NT);
    const int r = lane & 31, hh = lane >> 5;
    const int co = nt * 32 + r;
    char* base = out + (size_t)st * 8 * NT * 1024;
    auto weight = [&](int slot, int j) -> float {
        // slot within the stream: fold: channel group `slot`; else group slot / tp, tap slot % tp
        int tap, chn;
        if (fold) { tap = 0; chn = slot * 8 + j; }
        else { const int g = slot / tp; tap = slot - g * tp; chn = g * 8 + j; }
        if (tap >= T || chn >= cin || co >= cout) return 0.f;
        float v = w[((size_t)tap * cin_total + c_off + chn) * cout + co] * wscale;
        if (cscale != nullptr) v *= cscale[co];
        return v;
    };
    float v[32], lo[32];
    float mh = 0.f, ml = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        v[i] = weight(st * 8 + 2 * (i >> 3) + hh, i & 7);
        const _Float16 h = (_Float16)v[i];
        lo[i] = v[i] - (float)h;
        mh = fmaxf(mh, fabsf(v[i]));
        ml = fmaxf(ml, fabsf(lo[i]));
        reinterpret_cast<_Float16*>(base)[((size_t)((i >> 3) * NT + nt) * 64 + lane) * 8 + (i & 7)] = h;
    }
    int eh = mh > 0.f ? ilogbf(mh) - 3 : 0, el = ml > 0.f ? ilogbf(ml) - 3 : 0;
    eh = eh < -126 ? -126 : eh > 120 ? 120 : eh;
    el = el < -126 ? -126 : el > 120 ? 120 : el;
    const int scale_word = (eh + 127) | (el + 127) << 8;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        unsigned d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float inv = ldexpf(1.f, -(pl ? el : eh));
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const unsigned long long c = (unsigned long long)e3m2_encode((pl ? lo[i] : v[i]) * inv) << ((6 * i) & 31);
            d[(6 * i) >> 5] |= (unsigned)c;
            if (((6 * i) >> 5) + 1 < 6) d[((6 * i) >> 5) + 1] |= (unsigned)(c >> 32);
        }
        d[6] = (unsigned)scale_word;
        char* p6 = base + 4 * NT * 1024 + (size_t)pl * NT * 2048 + (size_t)nt * 2048 + (size_t)lane * 16;
        *reinterpret_cast<uint4*>(p6) = make_uint4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<uint4*>(p6 + 1024) = make_uint4(d[4], d[5], d[6], d[7]);
    }
}

// weights HWIO fp32 -> per (chunk, stage) fragment-ordered fp16 hi [lo] planes
__global__ void pack_weights_kernel(const float* __restrict__ w, int kh, int kw, int cin_total, int c_off,
                                    int cin, int cout, float wscale, const float* __restrict__ cscale,
                                    int NT, int KS, int NPL, int cgc, int nchunks, int sc,
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
    if (q < kh * kw * cgc) {
        const int tap = q / cgc;
        const int gg = q - tap * cgc;
        const int chn = (c * cgc + gg) * 8 + j;
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

// fp32 table [tap][ci 8][co 8] of a small layer (zero padded), scaled like the MFMA pack
__global__ void pack_small_kernel(const float* __restrict__ w, int taps, int cin_total, int c_off, int cin, int cout,
                                  float wscale, const float* __restrict__ cscale, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= taps * 64) return;
    const int co = idx & 7, ci = (idx >> 3) & 7, tap = idx >> 6;
    float v = 0.f;
    if (ci < cin && co < cout) {
        v = w[((size_t)tap * cin_total + c_off + ci) * cout + co] * wscale;
        if (cscale != nullptr) v *= cscale[co];
    }
    out[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// conv_small_kernel: fused convolution for layers with <= 8 input and <= 8 output channels per
// segment.  One thread per output pixel holds the 8 output channels; every tap is one 16-byte
// read per plane of the single G8 channel group (hi + lo -> fp32), the weights are wave-uniform
// scalar loads.  HBM / L1 bound, fp32 arithmetic on fp32-grade activations.
// ---------------------------------------------------------------------------------------------
struct SmallSeg {
    const char* x;
    const float* w;       // [tap][8][8]
    int cg_total, g_off, kh, kw, up, pt, pl, hs, ws, cin;
};

struct SmallArgs {
    int n, h, w, cout, nseg;
    int tile_floats;      // floats of the halo tile of the largest segment; the weight table follows it in LDS
    SmallSeg seg[MPG_MAX_SEG];
    const float* bias;
    const float* in_amax;
    int act;
    float leak;
    float* y;
    char* y_g8;
};

__device__ __forceinline__ void g8_load8(const char* src, size_t plane_bytes, float (&v)[8]) {
    const half8 hi = *reinterpret_cast<const half8*>(src);
    const half8 lo = *reinterpret_cast<const half8*>(src + plane_bytes);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)hi[j] + (float)lo[j];
}

// block = 64 x 16 output pixels; a thread owns a COLUMN of four of them (rows 4 yg .. 4 yg + 3 at column lx), so that
// consecutive lanes read consecutive 16-byte LDS words (no bank conflicts) and the rows a thread reads for one filter
// column serve all of its pixels: (4 + kh - 1) pixel reads per kx instead of 4 kh, and every tap's weight block --
// LDS broadcasts of the dense [tap][CINB][COUT] table -- feeds four pixels.  Per segment the halo tile is converted to
// fp32 ONCE into LDS as planes of four channels ([plane][row][col] float4, zero outside the image).
#ifndef MPG_SM_RPT
#define MPG_SM_RPT 4
#endif
constexpr int SM_TW = 64, SM_RPT = MPG_SM_RPT, SM_TH = 4 * SM_RPT, SM_KMAX = 7;
#ifndef MPG_DIAG_SMALL
#define MPG_DIAG_SMALL 0
#endif
#ifndef MPG_SMALL_INV
#define MPG_SMALL_INV 0
#endif
#if MPG_DIAG_SMALL
__device__ unsigned g_small_diag[2];
#endif

// COUT / CINB: output channels / input channels per segment rounded up to 1, 2, 4, 8 (compile-time loop bounds: the
// weight table holds zeros beyond cin and cout, a G8 group holds zeros beyond its channels)
template <int COUT, int CINB>
__global__ __launch_bounds__(256) void conv_small_kernel(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float small_lds[];
    constexpr int PL = (CINB + 3) / 4;                 // planes of four channels
    constexpr int CP = CINB < 4 ? CINB : 4;            // channels used of a plane
    float4* tile = reinterpret_cast<float4*>(small_lds);
    float* wl = small_lds + a.tile_floats;
    const int tid = threadIdx.x;
    const int lx = tid % SM_TW, yg = tid / SM_TW;
    const int x0 = blockIdx.x * SM_TW, y0 = blockIdx.y * SM_TH, b = blockIdx.z;
#if MPG_SMALL_INV
    asm volatile("buffer_inv sc0 sc1" ::: "memory");
#endif
    float acc[SM_RPT][COUT];
#pragma unroll
    for (int j = 0; j < SM_RPT; ++j)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[j][co] = 0.f;
    for (int s = 0; s < a.nseg; ++s) {
        const SmallSeg& g = a.seg[s];
        const size_t plane_bytes = (size_t)g.hs * g.ws * 16;
        const char* base = g.x + ((size_t)b * g.cg_total + g.g_off) * 2 * plane_bytes;
        const int tw = SM_TW + g.kw - 1, th = SM_TH + g.kh - 1;
        if (s > 0) __syncthreads();
        for (int p = tid; p < g.kh * g.kw * CINB * COUT; p += 256) {
            const int co = p % COUT, ci = (p / COUT) % CINB, tap = p / (COUT * CINB);
            wl[p] = g.w[tap * 64 + ci * 8 + co];
        }
        for (int p = tid; p < tw * th; p += 256) {
            const int hy = p / tw, hx = p - hy * tw;
            const int yy = y0 - g.pt + hy, xx = x0 - g.pl + hx;
            float v[8];
            if (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) {
                g8_load8(base + ((size_t)(yy >> g.up) * g.ws + (xx >> g.up)) * 16, plane_bytes, v);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 0.f;
            }
            tile[p] = make_float4(v[0], v[1], v[2], v[3]);
            if (PL > 1) tile[th * tw + p] = make_float4(v[4], v[5], v[6], v[7]);
        }
        __syncthreads();
        const int kh = g.kh;
        for (int kx = 0; kx < g.kw; ++kx) {
            // the rows this thread's four pixels see through filter column kx
            float rows[SM_RPT + SM_KMAX - 1][CINB];
#pragma unroll
            for (int r = 0; r < SM_RPT + SM_KMAX - 1; ++r) {
                if (r < SM_RPT + kh - 1) {
                    const float4* src = tile + (yg * SM_RPT + r) * tw + lx + kx;
                    const float4 p0 = src[0];
                    rows[r][0] = p0.x;
                    if constexpr (CP > 1) rows[r][1] = p0.y;
                    if constexpr (CP > 2) { rows[r][2] = p0.z; rows[r][3] = p0.w; }
                    if constexpr (PL > 1) {
                        const float4 p1 = src[th * tw];
                        rows[r][4] = p1.x; rows[r][5] = p1.y; rows[r][6] = p1.z; rows[r][7] = p1.w;
                    }
                }
            }
#pragma unroll
            for (int ky = 0; ky < SM_KMAX; ++ky) {
                if (ky < kh) {
                    const float* wt = wl + (ky * g.kw + kx) * (CINB * COUT);
#pragma unroll
                    for (int ci = 0; ci < CINB; ++ci) {
                        float wv[COUT];
                        if (COUT >= 4) {
#pragma unroll
                            for (int q = 0; q < COUT / 4; ++q) {
                                const float4 w4 = *reinterpret_cast<const float4*>(wt + ci * COUT + 4 * q);
                                wv[4 * q] = w4.x; wv[4 * q + 1] = w4.y; wv[4 * q + 2] = w4.z; wv[4 * q + 3] = w4.w;
                            }
                        } else if (COUT == 2) {
                            const float2 w2 = *reinterpret_cast<const float2*>(wt + ci * 2);
                            wv[0] = w2.x; wv[1] = w2.y;
                        } else {
                            wv[0] = wt[ci];
                        }
#pragma unroll
                        for (int j = 0; j < SM_RPT; ++j)
#pragma unroll
                            for (int co = 0; co < COUT; ++co) acc[j][co] = fmaf(rows[j + ky][ci], wv[co], acc[j][co]);
                    }
                }
            }
        }
#if MPG_DIAG_SMALL
        {   // diagnostic: is this block's LDS still what it wrote?  (foreign writes into the allocation; checked for EVERY
            // segment right after its sums, before the next segment overwrites the table and the tile)
            __syncthreads();
            unsigned bad_w = 0, bad_t = 0;
            for (int p = tid; p < g.kh * g.kw * CINB * COUT; p += 256) {
                const int co = p % COUT, ci = (p / COUT) % CINB, tap = p / (COUT * CINB);
                if (wl[p] != g.w[tap * 64 + ci * 8 + co]) ++bad_w;
            }
            for (int p = tid; p < tw * th; p += 256) {
                const int hy = p / tw, hx = p - hy * tw;
                const int yy = y0 - g.pt + hy, xx = x0 - g.pl + hx;
                float v[8];
                if (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) {
                    g8_load8(base + ((size_t)(yy >> g.up) * g.ws + (xx >> g.up)) * 16, plane_bytes, v);
                } else {
                    for (int q = 0; q < 8; ++q) v[q] = 0.f;
                }
                const float4 t0 = tile[p];
                if (t0.x != v[0] || t0.y != v[1] || t0.z != v[2] || t0.w != v[3]) ++bad_t;
                if (PL > 1) {
                    const float4 t1 = tile[th * tw + p];
                    if (t1.x != v[4] || t1.y != v[5] || t1.z != v[6] || t1.w != v[7]) ++bad_t;
                }
            }
            if (bad_w) atomicAdd(&g_small_diag[0], bad_w);
            if (bad_t) atomicAdd(&g_small_diag[1], bad_t);
        }
#endif
    }
    const int x = x0 + lx;
    if (x >= a.w) return;
    const float unscale = a.in_amax != nullptr ? 1.f / mpg::pow2_scale(*a.in_amax) : 1.f;
    const size_t plane_px = (size_t)a.h * a.w;
#pragma unroll
    for (int j = 0; j < SM_RPT; ++j) {
        const int y = y0 + yg * SM_RPT + j;
        if (y >= a.h) break;
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            o[q] = (q < COUT && q < a.cout) ? mpg::apply_act(acc[j][q < COUT ? q : 0] * unscale + (a.bias != nullptr ? a.bias[q] : 0.f),
                                                              a.act, a.leak)
                                            : 0.f;
        const size_t pix = (size_t)y * a.w + x;
        if (a.y != nullptr) {
            float* dst = a.y + ((size_t)b * plane_px + pix) * a.cout;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < a.cout) dst[q] = o[q];
        }
        if (a.y_g8 != nullptr) {
            half8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                hi[q] = (_Float16)o[q];
                lo[q] = (_Float16)(o[q] - (float)hi[q]);
            }
            char* dst = a.y_g8 + ((size_t)b * 2 * plane_px + pix) * 16;
            *reinterpret_cast<half8*>(dst) = hi;
            *reinterpret_cast<half8*>(dst + plane_px * 16) = lo;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv_small_pair_kernel: a residual block whose three convolutions all have <= 8 channels on either side -- the first
// and the last block of gen_resnet, relu(convB(relu(convA(x))) + conv1x1(x)) with 1 -> 2 -> 8 and 8 -> 2 -> 1 channels
// (GAN/multipassGAN-4x.py:505-526,560,564) -- as ONE launch.  The block's middle tensor never leaves the CU: stage A is
// evaluated on the output tile plus the halo of filter B (68 x 20 pixels for a 64 x 16 tile and a 5x5 filter) into LDS,
// stage B and the shortcut read it and the input tile from there.  As two launches the middle tensor made a round
// trip through HBM in the G8 layout (32 bytes per pixel written and read for two channels) and the second launch
// waited for the last block of the first.
// A thread owns a column of SM_RPT pixels in both stages (see conv_small_kernel).  Middle pixels outside the image
// are zero: filter B sees the SAME padding of a tensor of the image's size, not an extension of stage A.
// ---------------------------------------------------------------------------------------------
struct PairArgs {
    int n, h, w;
    const char* x;                 // G8 input, one channel group
    int cg_total, g_off, up, hs, ws;
    const float *wa, *wb, *wsc;    // [tap][8][8] tables: stage A (cin -> cmid), stage B (cmid -> cout), shortcut (cin -> cout) or null
    int kha, kwa, pta, pla;        // filter A and its SAME padding before
    int khb, kwb, ptb, plb;
    int khs, kws, pts, pls;
    const float *bias_a, *bias_b;
    int act_a, act_b;
    float leak_a, leak_b;
    int cout;
    float* y;
    char* y_g8;
    int x_px, mid_px;              // pixels of the input tile / of the middle tile (LDS plane sizes)
};

// sums of one filter over a column of SM_RPT pixels: tile = planes of four channels [plane][row][col] (float4), the
// thread's first row is `row0`, its column `col` (both in tile coordinates of the first tap); wl = [tap][CINB][COUTB]
template <int CINB, int COUTB>
__device__ __forceinline__ void small_column(const float4* tile, int tw, int plane_px, int row0, int col, const float* wl,
                                             int kh, int kw, float (&acc)[SM_RPT][COUTB]) {
    constexpr int PL = (CINB + 3) / 4, CP = CINB < 4 ? CINB : 4;
    for (int kx = 0; kx < kw; ++kx) {
        float rows[SM_RPT + SM_KMAX - 1][CINB];
#pragma unroll
        for (int r = 0; r < SM_RPT + SM_KMAX - 1; ++r) {
            if (r < SM_RPT + kh - 1) {
                const float4* src = tile + (row0 + r) * tw + col + kx;
                const float4 p0 = src[0];
                rows[r][0] = p0.x;
                if constexpr (CP > 1) rows[r][1] = p0.y;
                if constexpr (CP > 2) { rows[r][2] = p0.z; rows[r][3] = p0.w; }
                if constexpr (PL > 1) {
                    const float4 p1 = src[plane_px];
                    rows[r][4] = p1.x; rows[r][5] = p1.y; rows[r][6] = p1.z; rows[r][7] = p1.w;
                }
            }
        }
#pragma unroll
        for (int ky = 0; ky < SM_KMAX; ++ky) {
            if (ky < kh) {
                const float* wt = wl + (ky * kw + kx) * (CINB * COUTB);
#pragma unroll
                for (int ci = 0; ci < CINB; ++ci) {
                    float wv[COUTB];
#pragma unroll
                    for (int co = 0; co < COUTB; ++co) wv[co] = wt[ci * COUTB + co];
#pragma unroll
                    for (int j = 0; j < SM_RPT; ++j)
#pragma unroll
                        for (int co = 0; co < COUTB; ++co) acc[j][co] = fmaf(rows[j + ky][ci], wv[co], acc[j][co]);
                }
            }
        }
    }
}

template <int CINB, int CMIDB, int COUTB>
__global__ __launch_bounds__(256) void conv_small_pair_kernel(PairArgs a) {
    extern __shared__ __attribute__((aligned(16))) float small_lds[];
    constexpr int PLX = (CINB + 3) / 4, PLM = (CMIDB + 3) / 4;
    float4* xt = reinterpret_cast<float4*>(small_lds);               // input tile: PLX planes of x_px pixels
    float4* mt = xt + PLX * a.x_px;                                   // middle tile: PLM planes of mid_px pixels
    float* wla = reinterpret_cast<float*>(mt + PLM * a.mid_px);       // [tap][CINB][CMIDB]
    float* wlb = wla + a.kha * a.kwa * CINB * CMIDB;                  // [tap][CMIDB][COUTB]
    float* wls = wlb + a.khb * a.kwb * CMIDB * COUTB;                 // [tap][CINB][COUTB]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * SM_TW, y0 = blockIdx.y * SM_TH, b = blockIdx.z;
    const int mw = SM_TW + a.kwb - 1, mh = SM_TH + a.khb - 1;         // middle tile
    const int xw = mw + a.kwa - 1, xh = mh + a.kha - 1;               // input tile
    const int mx0 = x0 - a.plb, my0 = y0 - a.ptb;                     // image coordinates of the tiles' first pixels
    const int xx0 = mx0 - a.pla, xy0 = my0 - a.pta;
    for (int p = tid; p < a.kha * a.kwa * CINB * CMIDB; p += 256) {
        const int co = p % CMIDB, ci = (p / CMIDB) % CINB, tap = p / (CMIDB * CINB);
        wla[p] = a.wa[tap * 64 + ci * 8 + co];
    }
    for (int p = tid; p < a.khb * a.kwb * CMIDB * COUTB; p += 256) {
        const int co = p % COUTB, ci = (p / COUTB) % CMIDB, tap = p / (COUTB * CMIDB);
        wlb[p] = a.wb[tap * 64 + ci * 8 + co];
    }
    if (a.wsc != nullptr)
        for (int p = tid; p < a.khs * a.kws * CINB * COUTB; p += 256) {
            const int co = p % COUTB, ci = (p / COUTB) % CINB, tap = p / (COUTB * CINB);
            wls[p] = a.wsc[tap * 64 + ci * 8 + co];
        }
    {
        const size_t plane_bytes = (size_t)a.hs * a.ws * 16;
        const char* base = a.x + ((size_t)b * a.cg_total + a.g_off) * 2 * plane_bytes;
        for (int p = tid; p < xw * xh; p += 256) {
            const int hy = p / xw, hx = p - hy * xw;
            const int yy = xy0 + hy, xx = xx0 + hx;
            float v[8];
            if (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) {
                g8_load8(base + ((size_t)(yy >> a.up) * a.ws + (xx >> a.up)) * 16, plane_bytes, v);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 0.f;
            }
            xt[p] = make_float4(v[0], v[1], v[2], v[3]);
            if (PLX > 1) xt[a.x_px + p] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
    __syncthreads();
    // ---- stage A on the middle tile: columns of SM_RPT pixels, mw x ceil(mh / SM_RPT) of them ----
    const int mgroups = (mh + SM_RPT - 1) / SM_RPT;
    for (int t = tid; t < mw * mgroups; t += 256) {
        const int col = t % mw, rg = t / mw;
        float acc[SM_RPT][CMIDB];
#pragma unroll
        for (int j = 0; j < SM_RPT; ++j)
#pragma unroll
            for (int c = 0; c < CMIDB; ++c) acc[j][c] = 0.f;
        small_column<CINB, CMIDB>(xt, xw, a.x_px, rg * SM_RPT, col, wla, a.kha, a.kwa, acc);
#pragma unroll
        for (int j = 0; j < SM_RPT; ++j) {
            const int row = rg * SM_RPT + j;
            if (row < mh) {
                const int yy = my0 + row, xx = mx0 + col;
                const bool in = yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
                float o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    o[c] = (c < CMIDB && in) ? mpg::apply_act(acc[j][c < CMIDB ? c : 0] + (a.bias_a != nullptr ? a.bias_a[c] : 0.f), a.act_a, a.leak_a) : 0.f;
                mt[row * mw + col] = make_float4(o[0], o[1], o[2], o[3]);
                if (PLM > 1) mt[a.mid_px + row * mw + col] = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
    }
    __syncthreads();
    // ---- stage B + shortcut on the output tile ----
    const int lx = tid % SM_TW, yg = tid / SM_TW;
    float acc[SM_RPT][COUTB];
#pragma unroll
    for (int j = 0; j < SM_RPT; ++j)
#pragma unroll
        for (int c = 0; c < COUTB; ++c) acc[j][c] = 0.f;
    small_column<CMIDB, COUTB>(mt, mw, a.mid_px, yg * SM_RPT, lx, wlb, a.khb, a.kwb, acc);
    if (a.wsc != nullptr)     // the shortcut reads the input tile; its first tap sits at (ptb + pta - pts, plb + pla - pls) of it
        small_column<CINB, COUTB>(xt, xw, a.x_px, yg * SM_RPT + a.ptb + a.pta - a.pts, lx + a.plb + a.pla - a.pls, wls, a.khs, a.kws, acc);
    const int x = x0 + lx;
    if (x >= a.w) return;
    const size_t plane_px = (size_t)a.h * a.w;
#pragma unroll
    for (int j = 0; j < SM_RPT; ++j) {
        const int y = y0 + yg * SM_RPT + j;
        if (y >= a.h) break;
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            o[q] = (q < COUTB && q < a.cout) ? mpg::apply_act(acc[j][q < COUTB ? q : 0] + (a.bias_b != nullptr ? a.bias_b[q] : 0.f), a.act_b, a.leak_b) : 0.f;
        const size_t pix = (size_t)y * a.w + x;
        if (a.y != nullptr) {
            float* dst = a.y + ((size_t)b * plane_px + pix) * a.cout;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < a.cout) dst[q] = o[q];
        }
        if (a.y_g8 != nullptr) {
            half8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                hi[q] = (_Float16)o[q];
                lo[q] = (_Float16)(o[q] - (float)hi[q]);
            }
            char* dst = a.y_g8 + ((size_t)b * 2 * plane_px + pix) * 16;
            *reinterpret_cast<half8*>(dst) = hi;
            *reinterpret_cast<half8*>(dst + plane_px * 16) = lo;
        }
    }
}

// fp32 NHWC -> G8
__global__ void f32_to_g8_kernel(const float* __restrict__ x, int n, int h, int w, int c, int c_off, int cin,
                                 const float* __restrict__ amax, _Float16* __restrict__ out) {
    const float scale = amax != nullptr ? mpg::pow2_scale(*amax) : 1.f;
    const int cg_n = (cin + 7) >> 3;
    const size_t plane_px = (size_t)h * w;
    const size_t total = (size_t)n * cg_n * plane_px;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t px = idx % plane_px;
    const size_t t = idx / plane_px;
    const int cg = t % cg_n;
    const int b = t / cg_n;
    const float* src = x + ((size_t)b * plane_px + px) * c + c_off + cg * 8;
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (cg * 8 + j < cin) ? src[j] * scale : 0.f;
        hi[j] = (_Float16)v;
        lo[j] = (_Float16)(v - (float)hi[j]);
    }
    _Float16* dst = out + ((((size_t)b * cg_n + cg) * 2) * plane_px + px) * 8;
    *reinterpret_cast<half8*>(dst) = hi;
    *reinterpret_cast<half8*>(dst + plane_px * 8) = lo;
}

// The same conversion for wide tensors (cin >= 32, 16-byte aligned rows): the kernel above reads 32 bytes per thread at a
// stride of c floats (2.0 TB/s on a 128-channel tensor).  Here a block takes 32 pixels x up to 128 channels: the pixel
// rows are read as float4 by 32 consecutive threads (512 contiguous bytes), staged in LDS, and every (pixel, group) pair is
// then converted by one thread that writes 16 bytes per plane next to its neighbour pixel's.
constexpr int G8T_PX = 32, G8T_CH = 128, G8T_STRIDE = G8T_CH + 4;

__global__ __launch_bounds__(256) void f32_to_g8_tiled_kernel(const float* __restrict__ x, int n, int h, int w, int c, int c_off,
                                                              int cin, const float* __restrict__ amax,
                                                              _Float16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float tile[G8T_PX * G8T_STRIDE];
    const float scale = amax != nullptr ? mpg::pow2_scale(*amax) : 1.f;
    const int cg_n = (cin + 7) >> 3;
    const size_t plane_px = (size_t)h * w;
    const size_t px0 = (size_t)blockIdx.x * G8T_PX;          // first pixel of the tile within image b
    const int ch0 = blockIdx.y * G8T_CH;                      // first channel (of the cin window) of the tile
    const int b = blockIdx.z;
    const int nch = min(G8T_CH, cin - ch0);                   // channels of this tile
    const int tid = threadIdx.x;
    const int q = tid & 31, pr = tid >> 5;                    // channel quad, pixel sub-row
#pragma unroll
    for (int i = 0; i < G8T_PX / 8; ++i) {
        const int p = pr + 8 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (px0 + p < plane_px && q * 4 < nch) {
            const float* src = x + ((size_t)b * plane_px + px0 + p) * c + c_off + ch0 + q * 4;
            if (q * 4 + 3 < nch) {
                v = *reinterpret_cast<const float4*>(src);
            } else {
                v.x = src[0];
                if (q * 4 + 1 < nch) v.y = src[1];
                if (q * 4 + 2 < nch) v.z = src[2];
            }
        }
        *reinterpret_cast<float4*>(tile + p * G8T_STRIDE + q * 4) = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }
    __syncthreads();
    const int ng = (nch + 7) >> 3;
    for (int u = tid; u < ng * G8T_PX; u += 256) {
        const int p = u % G8T_PX, g = u / G8T_PX;
        if (px0 + p >= plane_px) continue;
        const float4 a0 = *reinterpret_cast<const float4*>(tile + p * G8T_STRIDE + g * 8);
        const float4 a1 = *reinterpret_cast<const float4*>(tile + p * G8T_STRIDE + g * 8 + 4);
        float vv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (_Float16)vv[j];
            lo[j] = (_Float16)(vv[j] - (float)hi[j]);
        }
        _Float16* dst = out + ((((size_t)b * cg_n + (ch0 >> 3) + g) * 2) * plane_px + px0 + p) * 8;
        *reinterpret_cast<half8*>(dst) = hi;
        *reinterpret_cast<half8*>(dst + plane_px * 8) = lo;
    }
}

// G8 -> fp32 NHWC
__global__ void g8_to_f32_kernel(const _Float16* __restrict__ g, int n, int h, int w, int c, float* __restrict__ y) {
    const int cg_n = (c + 7) >> 3;
    const size_t plane_px = (size_t)h * w;
    const size_t total = (size_t)n * plane_px * c;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ch = idx % c;
    const size_t t = idx / c;
    const size_t px = t % plane_px;
    const int b = t / plane_px;
    const _Float16* src = g + ((((size_t)b * cg_n + (ch >> 3)) * 2) * plane_px + px) * 8 + (ch & 7);
    y[idx] = (float)src[0] + (float)src[plane_px * 8];
}

struct Shape {
    int pt, th, ks, r, ni;
};

// host mirror of Pipe<NT, PREC>
Shape pipe_shape(int nt, int prec) {
    Shape s;
    const int npl = prec == MPG_PREC_F16X3 ? 2 : 1;
    s.pt = nt >= 3 ? 2 : 4;
    s.th = 4 * s.pt;
    if (prec == MPG_PREC_F16X3) s.ks = (nt == 4 || nt == 2) ? 1 : 2;
    else s.ks = (nt == 4 || nt == 2) ? 2 : 4;
    s.r = nt == 3 ? 3 : 4;
    s.ni = s.ks * nt * 1024 * npl / 4096;
    return s;
}

struct SegShape {
    int cgc, nchunks, sc, np, ni_img, img_bytes, direct, tp, pref;
};

// groups per chunk: two when that removes the half-empty k-step of an odd tap count and the
// double-buffered images still leave room for two workgroups per CU
SegShape seg_shape(int kh, int kw, int cin, int nt, int prec) {
    const Shape ps = pipe_shape(nt, prec);
    const int npl = prec == MPG_PREC_F16X3 ? 2 : 1;
    const int cg = (cin + 7) / 8;
    const int px = (ps.th + kh - 1) * (TW + kw - 1);
    const int np = (px + 63) & ~63;
    const int ring = ps.r * ps.ks * nt * 1024 * npl;
    SegShape s;
    s.direct = 0;
    s.tp = 0;
    s.pref = 0;
    s.np = np;
    s.cgc = 1;
    if (cg >= 2 && ((kh * kw) & 1)) {
        const int img2 = ((2 * 2 * (np / 64) + 3) & ~3) * 1024;
        if (TAPOFF_BYTES + 2 * img2 + ring <= 80 * 1024) s.cgc = 2;
    }
    const int pieces = s.cgc * 2 * (np / 64);
    s.ni_img = (pieces + 3) / 4;
    s.img_bytes = s.ni_img * 4 * 1024;
    s.nchunks = (cg + s.cgc - 1) / s.cgc;
    const int ksteps = (kh * kw * s.cgc + 1) / 2;
    s.sc = (ksteps + ps.ks - 1) / ps.ks;
    return s;
}

// ---- MPG_PREC_F16F6 shapes (host mirror of Pipe6<NT>) ----
bool f6_supported(int nt) { return nt >= 1 && nt <= 4; }
int f6_waves(int nt) { return nt == 1 ? 4 : 8; }

SegShape seg_shape_f6(int kh, int kw, int cin, int nt) {
    // nchunks = channel groups (one LDS halo image each), sc = weight stages of the whole segment
    SegShape s;
    s.direct = 0;
    s.tp = 0;
    s.pref = 0;
    if (kh == 1 && kw == 1 && cin > 8 && nt <= 2) {   // conv_mfma_f6_kernel's direct path: K over channel groups
        s.direct = 1; s.cgc = 1; s.np = 0; s.ni_img = 0; s.img_bytes = 0;
        s.nchunks = 1;
        s.sc = ((cin + 7) / 8 + 7) / 8;
        return s;
    }
    const int waves = f6_waves(nt);
    const int px = (16 + kh - 1) * (TW + kw - 1);
    s.np = (px + 63) & ~63;
    s.cgc = 1;
    const int pieces = 2 * (s.np / 64);
    s.ni_img = (pieces + waves - 1) / waves;
    s.img_bytes = s.ni_img * waves * 1024;
    s.nchunks = (cin + 7) / 8;
    // tap slots per group: the taps themselves when a group spans >= 2 stages; below that padded to 8, 12 or 16 so
    // that the image of group g+1 (issued once no stage touches group g-1) has >= 1 stage of flight before its
    // first slot: floor(tp (g+1) / 8) - ceil(tp g / 8) >= 1 for every g
    const int T = kh * kw;
    s.tp = T >= 16 ? T : T <= 8 ? 8 : T <= 12 ? 12 : 16;
    s.sc = (s.nchunks * s.tp + 7) / 8;
    // The image of group g is issued in stage ceil((g - 1) tp / 8) (the first stage that no longer touches group g - 2)
    // and its first slot lies in stage floor(g tp / 8).  The kernel may read stage st + 1's B fragments during stage st
    // only if every image is then already behind a barrier that followed its DMA wait: two stages between issue and
    // first use, for every group.
    s.pref = 1;
    for (int g = 1; g < s.nchunks; ++g)
        if ((g * s.tp) / 8 - ((g - 1) * s.tp + 7) / 8 < 2) s.pref = 0;
    return s;
}

template <int NT>
hipError_t launch_f6(dim3 grid, size_t lds, hipStream_t st, const ConvArgs& a) {
    static int lds_limit[64] = {0};
    if (lds > 48 * 1024) {
        hipError_t e = mpg::ensure_dyn_lds(reinterpret_cast<const void*>(&conv_mfma_f6_kernel<NT>), (int)lds, lds_limit);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv_mfma_f6_kernel<NT>), grid, dim3(Pipe6<NT>::WAVES * 64), lds, st, a);
    return hipSuccess;
}

template <int NT, int PREC>
hipError_t launch_one(dim3 grid, size_t lds, hipStream_t st, const ConvArgs& a) {
    static int lds_limit[64] = {0};
    if (lds > 48 * 1024) {
        hipError_t e = mpg::ensure_dyn_lds(reinterpret_cast<const void*>(&conv_mfma_kernel<NT, PREC>), (int)lds, lds_limit);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<NT, PREC>), grid, dim3(256), lds, st, a);
    return hipSuccess;
}

template <int PREC>
hipError_t launch_nt(int nt, dim3 grid, size_t lds, hipStream_t st, const ConvArgs& a) {
    switch (nt) {
        case 1: return launch_one<1, PREC>(grid, lds, st, a);
        case 2: return launch_one<2, PREC>(grid, lds, st, a);
        case 3: return launch_one<3, PREC>(grid, lds, st, a);
        default: return launch_one<4, PREC>(grid, lds, st, a);
    }
}

// 256 zero bytes per device: the DMA source of out-of-image halo pixels.  Allocated on the first launch on a
// device under a lock; the fill is a blocking hipMemset followed by a device synchronise, so the page is zero before
// ANY stream (torch's side streams are non-blocking) can run a kernel that reads it.  Later calls are a lock-free
// read.  Must first happen outside a stream capture (Session / Trainer run one eager step before capturing).
const char* zero_buffer() {
    static std::mutex mu;
    static char* per_dev[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    char* p = __atomic_load_n(&per_dev[dev], __ATOMIC_ACQUIRE);
    if (p) return p;
    std::lock_guard<std::mutex> lock(mu);
    p = per_dev[dev];
    if (!p) {
        if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        __atomic_store_n(&per_dev[dev], p, __ATOMIC_RELEASE);
    }
    return p;
}

}  // namespace

namespace mpg {
const char* zero_page() { return zero_buffer(); }
}  // namespace mpg

extern "C" size_t mpg_g8_bytes(int n, int h, int w, int c) {
    if (n < 1 || h < 1 || w < 1 || c < 1) return 0;
    return (size_t)n * ((c + 7) / 8) * 2 * h * w * 16;
}

extern "C" int mpg_f32_to_g8_scaled(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int c_off, int cin,
                                    int flavour, const float* amax, void* out) {
    MPG_REQUIRE(flavour == MPG_G8_F16, "mpg_f32_to_g8: bad flavour %d", flavour);
    MPG_REQUIRE(x && out, "mpg_f32_to_g8: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1 && c_off >= 0 && cin >= 1 && c_off + cin <= c, "mpg_f32_to_g8: bad shape");
    const size_t total = (size_t)n * ((cin + 7) / 8) * h * w;
    if (cin >= 32 && (c % 4) == 0 && (c_off % 4) == 0 && (((uintptr_t)x) & 15) == 0 && n <= 65535) {
        const dim3 grid((unsigned)(((size_t)h * w + G8T_PX - 1) / G8T_PX), (unsigned)((cin + G8T_CH - 1) / G8T_CH), (unsigned)n);
        hipLaunchKernelGGL(f32_to_g8_tiled_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, c_off, cin,
                           amax, (_Float16*)out);
        MPG_LAUNCH_CHECK("f32_to_g8_tiled_kernel");
    }
    hipLaunchKernelGGL(f32_to_g8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, h,
                       w, c, c_off, cin, amax, (_Float16*)out);
    MPG_LAUNCH_CHECK("f32_to_g8_kernel");
}

extern "C" int mpg_f32_to_g8(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int c_off, int cin,
                             int flavour, void* out) {
    return mpg_f32_to_g8_scaled(stream, x, n, h, w, c, c_off, cin, flavour, nullptr, out);
}

extern "C" int mpg_g8_to_f32(mpg_stream_t stream, const void* g8, int n, int h, int w, int c, float* y) {
    MPG_REQUIRE(g8 && y, "mpg_g8_to_f32: null pointer");
    MPG_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "mpg_g8_to_f32: bad shape");
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(g8_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)g8, n, h, w, c, y);
    MPG_LAUNCH_CHECK("g8_to_f32_kernel");
}

// Layers with at most 8 input and 8 output channels (the first and last residual blocks of gen_resnet:
// 1->2->8 and 8->2->1) are not matrix work: they run on conv_small_kernel, which reads a plain fp32
// table [tap][ci 8][co 8] appended to the packed weights.
static inline bool small_layer(int cin, int cout) { return cin <= 8 && cout <= 8; }
static size_t pack_base_bytes(int kh, int kw, int cin, int cout, int prec);

extern "C" size_t mpg_conv_pack_size(int kh, int kw, int cin, int cout, int prec) {
    const size_t base = pack_base_bytes(kh, kw, cin, cout, prec);
    if (base == 0) return 0;
    return base + (small_layer(cin, cout) ? (size_t)kh * kw * 64 * sizeof(float) : 0);
}

static size_t pack_base_bytes(int kh, int kw, int cin, int cout, int prec) {
    if (kh < 1 || kw < 1 || kh > 7 || kw > 7 || cin < 1 || cout < 1 || cout > 128) return 0;
    const int nt = (cout + 31) / 32;
    if (prec == MPG_PREC_F16F6) {
        if (!f6_supported(nt)) return 0;
        const SegShape ss = seg_shape_f6(kh, kw, cin, nt);
        // a segment whose tables + two images + ring cannot fit the 160 KiB of LDS is "not available at this
        // precision" (callers then pack for MPG_PREC_F16X3), e.g. 7x7 with four cout tiles
        const size_t tabs = ss.direct ? TAPOFF_BYTES : (((size_t)ss.sc * 8 * 4 + 1023) / 1024) * 1024;
        if ((tabs < (size_t)TAPOFF_BYTES ? (size_t)TAPOFF_BYTES : tabs) + 2 * (size_t)ss.img_bytes + (size_t)3 * 8 * nt * 1024 > 160 * 1024)
            return 0;
        return (size_t)ss.sc * 8 * nt * 1024;
    }
    if (prec != MPG_PREC_F16X1 && prec != MPG_PREC_F16X3) return 0;
    const Shape ps = pipe_shape(nt, prec);
    const SegShape ss = seg_shape(kh, kw, cin, nt, prec);
    const int npl = prec == MPG_PREC_F16X3 ? 2 : 1;
    return (size_t)ss.nchunks * ss.sc * ps.ks * nt * 1024 * npl;
}

extern "C" int mpg_conv_pack_weights(mpg_stream_t stream, const float* w_hwio, int kh, int kw, int w_cin_total,
                                     int w_c_off, int cin, int cout, float wscale, const float* cout_scale, int prec,
                                     void* out, size_t out_bytes) {
    MPG_REQUIRE(w_hwio && out, "mpg_conv_pack_weights: null pointer");
    MPG_REQUIRE(prec == MPG_PREC_F16X1 || prec == MPG_PREC_F16X3 || prec == MPG_PREC_F16F6,
                "mpg_conv_pack_weights: bad prec %d", prec);
    MPG_REQUIRE(kh >= 1 && kh <= 7 && kw >= 1 && kw <= 7, "mpg_conv_pack_weights: kernel %dx%d unsupported", kh, kw);
    MPG_REQUIRE(cin >= 1 && w_c_off >= 0 && w_c_off + cin <= w_cin_total, "mpg_conv_pack_weights: channel range");
    MPG_REQUIRE(cout >= 1 && cout <= 128, "mpg_conv_pack_weights: cout %d not in 1..128", cout);
    const size_t need = mpg_conv_pack_size(kh, kw, cin, cout, prec);
    MPG_REQUIRE(need > 0, "mpg_conv_pack_weights: %dx%d %d->%d not available at prec %d", kh, kw, cin, cout, prec);
    MPG_REQUIRE(out_bytes >= need, "mpg_conv_pack_weights: out buffer %zu < %zu", out_bytes, need);
    const int nt = (cout + 31) / 32;
    if (prec == MPG_PREC_F16F6) {
        const SegShape ss = seg_shape_f6(kh, kw, cin, nt);
        const long total = (long)ss.sc * nt * 64;
        hipLaunchKernelGGL(pack_weights_f6_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                           w_hwio, kh, kw, w_cin_total, w_c_off, cin, cout, wscale, cout_scale, nt, ss.sc, ss.direct, ss.tp,
                           (char*)out);
        if (small_layer(cin, cout))
            hipLaunchKernelGGL(pack_small_kernel, dim3((kh * kw * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_hwio,
                               kh * kw, w_cin_total, w_c_off, cin, cout, wscale, cout_scale,
                               (float*)((char*)out + pack_base_bytes(kh, kw, cin, cout, prec)));
        MPG_LAUNCH_CHECK("pack_weights_f6_kernel");
    }
    const Shape ps = pipe_shape(nt, prec);
    const SegShape ss = seg_shape(kh, kw, cin, nt, prec);
    const int npl = prec == MPG_PREC_F16X3 ? 2 : 1;
    const long total = (long)ss.nchunks * ss.sc * ps.ks * nt * 512;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, kh, kw,
                       w_cin_total, w_c_off, cin, cout, wscale, cout_scale, nt, ps.ks, npl, ss.cgc, ss.nchunks, ss.sc,
                       (_Float16*)out);
    if (small_layer(cin, cout))
        hipLaunchKernelGGL(pack_small_kernel, dim3((kh * kw * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_hwio,
                           kh * kw, w_cin_total, w_c_off, cin, cout, wscale, cout_scale,
                           (float*)((char*)out + pack_base_bytes(kh, kw, cin, cout, prec)));
    MPG_LAUNCH_CHECK("pack_weights_kernel");
}

// One residual block of <= 8-channel convolutions as a single launch (conv_small_pair_kernel).
template <int CI, int CM, int CO>
static hipError_t launch_pair(dim3 grid, size_t lds, hipStream_t st, const PairArgs& a) {
    static int lds_limit[64] = {0};
    if (lds > 48 * 1024) {
        hipError_t e = mpg::ensure_dyn_lds(reinterpret_cast<const void*>(&conv_small_pair_kernel<CI, CM, CO>), (int)lds, lds_limit);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv_small_pair_kernel<CI, CM, CO>), grid, dim3(256), lds, st, a);
    return hipSuccess;
}

extern "C" int mpg_conv2d_small_pair(mpg_stream_t stream, const mpg_small_pair_desc* d) {
    MPG_REQUIRE(d != nullptr, "mpg_conv2d_small_pair: null desc");
    MPG_REQUIRE(d->n >= 1 && d->h >= 1 && d->w >= 1 && d->n <= 65535, "mpg_conv2d_small_pair: bad shape %d x %d x %d", d->n, d->h, d->w);
    MPG_REQUIRE(d->cin >= 1 && d->cin <= 8 && d->cmid >= 1 && d->cmid <= 8 && d->cout >= 1 && d->cout <= 8,
                "mpg_conv2d_small_pair: %d -> %d -> %d channels (1..8 each)", d->cin, d->cmid, d->cout);
    MPG_REQUIRE(d->x && d->wpack_a && d->wpack_b, "mpg_conv2d_small_pair: null pointer");
    MPG_REQUIRE(d->y != nullptr || d->y_g8 != nullptr, "mpg_conv2d_small_pair: no output requested");
    MPG_REQUIRE(d->kh_a >= 1 && d->kh_a <= 7 && d->kw_a >= 1 && d->kw_a <= 7 && d->kh_b >= 1 && d->kh_b <= 7 && d->kw_b >= 1 && d->kw_b <= 7,
                "mpg_conv2d_small_pair: kernel sizes 1..7");
    MPG_REQUIRE((d->kh_a & 1) && (d->kw_a & 1) && (d->kh_b & 1) && (d->kw_b & 1), "mpg_conv2d_small_pair: odd filters only");
    MPG_REQUIRE(d->wpack_s == nullptr || (d->kh_s >= 1 && d->kh_s <= d->kh_a + d->kh_b - 1 && d->kw_s >= 1 && d->kw_s <= d->kw_a + d->kw_b - 1 &&
                                          (d->kh_s & 1) && (d->kw_s & 1) && ((d->kh_a + d->kh_b) & 1) == 0 && ((d->kw_a + d->kw_b) & 1) == 0),
                "mpg_conv2d_small_pair: the shortcut filter must be odd and fit inside the input tile of two odd filters");
    MPG_REQUIRE(d->g_off >= 0 && d->g_off < d->cgroups, "mpg_conv2d_small_pair: channel-group range");
    MPG_REQUIRE(d->up_log2 >= 0 && d->up_log2 <= 4 && (d->h % (1 << d->up_log2)) == 0 && (d->w % (1 << d->up_log2)) == 0,
                "mpg_conv2d_small_pair: upsample %d", d->up_log2);
    MPG_REQUIRE(d->act_a >= MPG_ACT_NONE && d->act_a <= MPG_ACT_TANH && d->act_b >= MPG_ACT_NONE && d->act_b <= MPG_ACT_TANH, "mpg_conv2d_small_pair: bad act");
    MPG_REQUIRE(d->prec == MPG_PREC_F16X1 || d->prec == MPG_PREC_F16X3 || d->prec == MPG_PREC_F16F6, "mpg_conv2d_small_pair: bad prec %d", d->prec);
    MPG_REQUIRE((((uintptr_t)d->x) & 15) == 0 && (((uintptr_t)d->y_g8) & 15) == 0, "mpg_conv2d_small_pair: misaligned tensor");
    PairArgs a;
    a.n = d->n; a.h = d->h; a.w = d->w;
    a.x = (const char*)d->x; a.cg_total = d->cgroups; a.g_off = d->g_off; a.up = d->up_log2;
    a.hs = d->h >> d->up_log2; a.ws = d->w >> d->up_log2;
    a.wa = (const float*)((const char*)d->wpack_a + pack_base_bytes(d->kh_a, d->kw_a, d->cin, d->cmid, d->prec));
    a.wb = (const float*)((const char*)d->wpack_b + pack_base_bytes(d->kh_b, d->kw_b, d->cmid, d->cout, d->prec));
    a.wsc = d->wpack_s ? (const float*)((const char*)d->wpack_s + pack_base_bytes(d->kh_s, d->kw_s, d->cin, d->cout, d->prec)) : nullptr;
    a.kha = d->kh_a; a.kwa = d->kw_a; a.pta = (d->kh_a - 1) / 2; a.pla = (d->kw_a - 1) / 2;
    a.khb = d->kh_b; a.kwb = d->kw_b; a.ptb = (d->kh_b - 1) / 2; a.plb = (d->kw_b - 1) / 2;
    a.khs = d->wpack_s ? d->kh_s : 1; a.kws = d->wpack_s ? d->kw_s : 1; a.pts = (a.khs - 1) / 2; a.pls = (a.kws - 1) / 2;
    a.bias_a = d->bias_a; a.bias_b = d->bias_b; a.act_a = d->act_a; a.act_b = d->act_b; a.leak_a = d->leak_a; a.leak_b = d->leak_b;
    a.cout = d->cout; a.y = d->y; a.y_g8 = (char*)d->y_g8;
    const int mw = SM_TW + a.kwb - 1, mh = SM_TH + a.khb - 1;
    const int mgroups = (mh + SM_RPT - 1) / SM_RPT;
    const int xw = mw + a.kwa - 1, xh = mgroups * SM_RPT + a.kha - 1;      // rows past the middle tile are read, never used
    a.x_px = xw * xh;
    a.mid_px = mw * (mgroups * SM_RPT + a.khb - 1);
    // compile-time channel bounds: cin in {1, 4, 8}, cmid in {2, 8}, cout in {1, 8}
    const int ci = d->cin == 1 ? 1 : d->cin <= 4 ? 4 : 8, cm = d->cmid <= 2 ? 2 : 8, co = d->cout == 1 ? 1 : 8;
    const size_t lds = ((size_t)((ci + 3) / 4) * a.x_px + (size_t)((cm + 3) / 4) * a.mid_px) * 16 +
                       ((size_t)a.kha * a.kwa * ci * cm + (size_t)a.khb * a.kwb * cm * co + (size_t)a.khs * a.kws * ci * co) * sizeof(float);
    MPG_REQUIRE(lds <= 160 * 1024, "mpg_conv2d_small_pair: LDS budget %zu exceeds 160 KiB", lds);
    const dim3 grid((unsigned)((d->w + SM_TW - 1) / SM_TW), (unsigned)((d->h + SM_TH - 1) / SM_TH), (unsigned)d->n);
    hipError_t le = hipSuccess;
    switch (ci * 100 + cm * 10 + co) {
#define MPG_PAIR(CI, CM, CO) case CI * 100 + CM * 10 + CO: le = launch_pair<CI, CM, CO>(grid, lds, (hipStream_t)stream, a); break;
        MPG_PAIR(1, 2, 1) MPG_PAIR(1, 2, 8) MPG_PAIR(1, 8, 1) MPG_PAIR(1, 8, 8)
        MPG_PAIR(4, 2, 1) MPG_PAIR(4, 2, 8) MPG_PAIR(4, 8, 1) MPG_PAIR(4, 8, 8)
        MPG_PAIR(8, 2, 1) MPG_PAIR(8, 2, 8) MPG_PAIR(8, 8, 1) MPG_PAIR(8, 8, 8)
#undef MPG_PAIR
        default: break;
    }
    if (le != hipSuccess) return mpg::hip_check(le, "mpg_conv2d_small_pair: hipFuncSetAttribute(dynamic LDS)");
    MPG_LAUNCH_CHECK("conv_small_pair_kernel");
}

#if MPG_DIAG_SMALL
extern "C" int mpg_debug_small_diag(unsigned* out2) {
    return hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_small_diag), 8) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int mpg_conv2d_fused(mpg_stream_t stream, const mpg_conv_desc* d) {
    MPG_REQUIRE(d != nullptr, "mpg_conv2d_fused: null desc");
    MPG_REQUIRE(d->n >= 1 && d->h >= 1 && d->w >= 1, "mpg_conv2d_fused: bad shape %d x %d x %d", d->n, d->h, d->w);
    MPG_REQUIRE(d->cout >= 1 && d->cout <= 128, "mpg_conv2d_fused: cout %d not in 1..128", d->cout);
    MPG_REQUIRE(d->nseg >= 1 && d->nseg <= MPG_MAX_SEG, "mpg_conv2d_fused: nseg %d", d->nseg);
    MPG_REQUIRE(d->y != nullptr || d->y_g8 != nullptr, "mpg_conv2d_fused: no output requested");
    MPG_REQUIRE(d->prec == MPG_PREC_F16X1 || d->prec == MPG_PREC_F16X3 || d->prec == MPG_PREC_F16F6,
                "mpg_conv2d_fused: bad prec %d", d->prec);
    MPG_REQUIRE(d->act >= MPG_ACT_NONE && d->act <= MPG_ACT_TANH, "mpg_conv2d_fused: bad act %d", d->act);
    {   // small-channel layers: conv_small_kernel
        bool small = d->cout <= 8 && !d->pixel_norm && d->post_add == nullptr && d->reserved == 0;
        for (int s = 0; s < d->nseg && small; ++s) small = d->seg[s].cin <= 8;
        if (small) {
            SmallArgs sa;
            sa.n = d->n; sa.h = d->h; sa.w = d->w; sa.cout = d->cout; sa.nseg = d->nseg;
            for (int s = 0; s < d->nseg; ++s) {
                const mpg_conv_seg& g = d->seg[s];
                MPG_REQUIRE(g.x && g.wpack, "mpg_conv2d_fused: segment %d null pointer", s);
                MPG_REQUIRE(g.kh >= 1 && g.kh <= 7 && g.kw >= 1 && g.kw <= 7, "mpg_conv2d_fused: segment %d kernel %dx%d", s, g.kh, g.kw);
                MPG_REQUIRE(g.cin >= 1 && g.g_off >= 0 && g.g_off < g.cgroups, "mpg_conv2d_fused: segment %d channel-group range", s);
                MPG_REQUIRE(g.up_log2 >= 0 && g.up_log2 <= 4 && (d->h % (1 << g.up_log2)) == 0 && (d->w % (1 << g.up_log2)) == 0,
                            "mpg_conv2d_fused: segment %d upsample %d", s, g.up_log2);
                MPG_REQUIRE((((uintptr_t)g.x) & 15) == 0 && (((uintptr_t)g.wpack) & 15) == 0, "mpg_conv2d_fused: segment %d misaligned", s);
                SmallSeg& o = sa.seg[s];
                o.x = (const char*)g.x;
                o.w = (const float*)((const char*)g.wpack + pack_base_bytes(g.kh, g.kw, g.cin, d->cout, d->prec));
                o.cg_total = g.cgroups; o.g_off = g.g_off; o.kh = g.kh; o.kw = g.kw; o.up = g.up_log2;
                o.pt = g.pad_hi ? g.kh / 2 : (g.kh - 1) / 2; o.pl = g.pad_hi ? g.kw / 2 : (g.kw - 1) / 2;
                o.hs = d->h >> g.up_log2; o.ws = d->w >> g.up_log2; o.cin = g.cin;
            }
            for (int s = d->nseg; s < MPG_MAX_SEG; ++s) sa.seg[s] = sa.seg[0];
            sa.bias = d->bias; sa.in_amax = d->in_amax; sa.act = d->act; sa.leak = d->leak;
            sa.y = d->y; sa.y_g8 = (char*)d->y_g8;
            MPG_REQUIRE((((uintptr_t)d->y_g8) & 15) == 0, "mpg_conv2d_fused: misaligned output");
            const size_t total = (size_t)d->n * d->h * d->w;
            (void)total;
            const dim3 sg((unsigned)((d->w + SM_TW - 1) / SM_TW), (unsigned)((d->h + SM_TH - 1) / SM_TH), (unsigned)d->n);
            int cmax = 1, tmax = 1, tile_px = 0;
            for (int s = 0; s < d->nseg; ++s) {
                const mpg_conv_seg& g = d->seg[s];
                cmax = g.cin > cmax ? g.cin : cmax;
                tmax = g.kh * g.kw > tmax ? g.kh * g.kw : tmax;
                const int px = (SM_TH + g.kh - 1) * (SM_TW + g.kw - 1);
                tile_px = px > tile_px ? px : tile_px;
            }
            const int cob = d->cout == 1 ? 1 : d->cout == 2 ? 2 : d->cout <= 4 ? 4 : 8;
            const int cib = cmax == 1 ? 1 : cmax == 2 ? 2 : cmax <= 4 ? 4 : 8;
            sa.tile_floats = tile_px * 4 * ((cib + 3) / 4);       // one or two planes of four channels
#ifndef MPG_SM_PAD
#define MPG_SM_PAD 0
#endif
            const size_t small_lds = ((size_t)sa.tile_floats + (size_t)tmax * cib * cob) * sizeof(float) + MPG_SM_PAD;
            switch (cob * 16 + cib) {
#define MPG_SMALL(CO, CI) \
    case CO * 16 + CI: hipLaunchKernelGGL((conv_small_kernel<CO, CI>), sg, dim3(256), small_lds, (hipStream_t)stream, sa); break;
                MPG_SMALL(1, 1) MPG_SMALL(1, 2) MPG_SMALL(1, 4) MPG_SMALL(1, 8)
                MPG_SMALL(2, 1) MPG_SMALL(2, 2) MPG_SMALL(2, 4) MPG_SMALL(2, 8)
                MPG_SMALL(4, 1) MPG_SMALL(4, 2) MPG_SMALL(4, 4) MPG_SMALL(4, 8)
                MPG_SMALL(8, 1) MPG_SMALL(8, 2) MPG_SMALL(8, 4) MPG_SMALL(8, 8)
#undef MPG_SMALL
                default: break;
            }
            MPG_LAUNCH_CHECK("conv_small_kernel");
        }
    }
    const int nt = (d->cout + 31) / 32;
    const bool f8 = d->prec == MPG_PREC_F16F6;
    if (f8 && !f6_supported(nt)) {
        mpg::set_error("mpg_conv2d_fused: MPG_PREC_F16F6 not available for cout %d", d->cout);
        return MPG_ERR_UNSUPPORTED;
    }
    Shape ps = pipe_shape(nt, f8 ? MPG_PREC_F16X3 : d->prec);
    if (f8) { ps.th = 16; ps.pt = 16 / f6_waves(nt); }
    const int npl = d->prec == MPG_PREC_F16X3 ? 2 : 1;

    ConvArgs a;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cout = d->cout; a.nseg = d->nseg;
    int max_img = 0, max_slots = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const mpg_conv_seg& g = d->seg[s];
        MPG_REQUIRE(g.x && g.wpack, "mpg_conv2d_fused: segment %d null pointer", s);
        MPG_REQUIRE(g.kh >= 1 && g.kh <= 7 && g.kw >= 1 && g.kw <= 7, "mpg_conv2d_fused: segment %d kernel %dx%d", s, g.kh, g.kw);
        MPG_REQUIRE(g.cin >= 1 && g.g_off >= 0 && g.g_off + (g.cin + 7) / 8 <= g.cgroups,
                    "mpg_conv2d_fused: segment %d channel-group range", s);
        MPG_REQUIRE(g.up_log2 >= 0 && g.up_log2 <= 4, "mpg_conv2d_fused: segment %d up_log2 %d", s, g.up_log2);
        MPG_REQUIRE((d->h % (1 << g.up_log2)) == 0 && (d->w % (1 << g.up_log2)) == 0,
                    "mpg_conv2d_fused: segment %d: %dx%d not divisible by upsample %d", s, d->h, d->w, 1 << g.up_log2);
        MPG_REQUIRE((((uintptr_t)g.x) & 15) == 0 && (((uintptr_t)g.wpack) & 15) == 0, "mpg_conv2d_fused: segment %d misaligned", s);
        const SegShape ss = f8 ? seg_shape_f6(g.kh, g.kw, g.cin, nt) : seg_shape(g.kh, g.kw, g.cin, nt, d->prec);
        MPG_REQUIRE(f8 || ss.sc * ps.ks * 2 <= TAPOFF_BYTES / 4, "mpg_conv2d_fused: segment %d tap table too large", s);
        if (f8 && !ss.direct) max_slots = ss.sc * 8 > max_slots ? ss.sc * 8 : max_slots;
        SegArgs& o = a.seg[s];
        o.x = (const char*)g.x; o.w = (const char*)g.wpack;
        o.cg_seg = (g.cin + 7) / 8; o.cg_total = g.cgroups; o.g_off = g.g_off;
        o.kh = g.kh; o.kw = g.kw; o.up = g.up_log2;
        o.cgc = ss.cgc; o.nchunks = ss.nchunks; o.sc = ss.sc;
        o.ih = ps.th + g.kh - 1; o.iw = TW + g.kw - 1;
        o.pt = g.pad_hi ? g.kh / 2 : (g.kh - 1) / 2; o.pl = g.pad_hi ? g.kw / 2 : (g.kw - 1) / 2;
        o.hs = d->h >> g.up_log2; o.ws = d->w >> g.up_log2;
        o.np = ss.np; o.ni_img = ss.ni_img;
        o.direct = ss.direct;
        o.tp = ss.tp;
        MPG_REQUIRE(!ss.direct || (size_t)(d->h >> g.up_log2) * (d->w >> g.up_log2) * 16 * 16 < ((size_t)1 << 31),
                    "mpg_conv2d_fused: segment %d: %dx%d too large for the 1x1 path (32-bit group offsets)", s, d->h, d->w);
        o.pref = ss.pref;
        max_img = ss.img_bytes > max_img ? ss.img_bytes : max_img;
    }
    for (int s = d->nseg; s < MPG_MAX_SEG; ++s) a.seg[s] = a.seg[0];
    a.bias = d->bias; a.in_amax = d->in_amax; a.act = d->act; a.leak = d->leak; a.pn = d->pixel_norm; a.pn_eps = d->pn_eps;
    a.post_add = d->post_add; a.pa_stride = d->post_add_stride; a.pa_coff = d->post_add_coff;
    MPG_REQUIRE(!d->post_add || d->post_add_coff + d->cout <= d->post_add_stride, "mpg_conv2d_fused: post_add channel range");
    a.y = d->y;
    a.y_g8 = (char*)d->y_g8;
    MPG_REQUIRE((((uintptr_t)d->y) & 15) == 0 && (((uintptr_t)d->y_g8) & 15) == 0,
                "mpg_conv2d_fused: misaligned output");
    a.zeros = zero_buffer();
    MPG_REQUIRE(a.zeros != nullptr, "mpg_conv2d_fused: could not allocate the zero page");
    a.img_bytes = max_img;
    a.tiles_x = (d->w + TW - 1) / TW;
    a.tiles_y = (d->h + ps.th - 1) / ps.th;
    a.dbg = d->reserved;
    const long nblk = (long)d->n * a.tiles_x * a.tiles_y;
    MPG_REQUIRE(nblk < (1L << 31), "mpg_conv2d_fused: grid too large");
    const int waves = f8 ? f6_waves(nt) : 4;
    const size_t ring = f8 ? (size_t)3 * 8 * nt * 1024 : (size_t)ps.r * ps.ks * nt * 1024 * npl;
    // F16F6: the tap-offset table ([stage][half][k-step]) of max_slots entries, 1 KiB granules
    a.tap_bytes = f8 ? ((max_slots * 4 + 1023) / 1024) * 1024 : TAPOFF_BYTES;
    if (a.tap_bytes < TAPOFF_BYTES) a.tap_bytes = TAPOFF_BYTES;
    const size_t lds_loop = (size_t)a.tap_bytes + 2 * (size_t)max_img + ring;
    const size_t lds_epi = TAPOFF_BYTES + (size_t)waves * 32 * (nt * 32 + 4) * sizeof(float);
    const size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    MPG_REQUIRE(lds <= 160 * 1024, "mpg_conv2d_fused: LDS budget %zu exceeds 160 KiB", lds);
    const dim3 grid((unsigned)nblk);
    hipError_t le;
    if (f8) {
        switch (nt) {
            case 1: le = launch_f6<1>(grid, lds, (hipStream_t)stream, a); break;
            case 2: le = launch_f6<2>(grid, lds, (hipStream_t)stream, a); break;
            case 3: le = launch_f6<3>(grid, lds, (hipStream_t)stream, a); break;
            default: le = launch_f6<4>(grid, lds, (hipStream_t)stream, a); break;
        }
    } else if (d->prec == MPG_PREC_F16X3)
        le = launch_nt<3>(nt, grid, lds, (hipStream_t)stream, a);
    else
        le = launch_nt<1>(nt, grid, lds, (hipStream_t)stream, a);
    if (le != hipSuccess) return mpg::hip_check(le, "mpg_conv2d_fused: hipFuncSetAttribute(dynamic LDS)");
    MPG_LAUNCH_CHECK("conv_mfma_kernel");
}
