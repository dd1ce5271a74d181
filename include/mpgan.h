/*
 * mpgan.h -- C ABI of the MI355X (gfx950) multi-pass GAN hot path.
 *
 * The reference (maxwerhahn/Multi-pass-GAN) is pure Python over TensorFlow 1.x
 * and has no FFI of its own; every entry point below
 * therefore names the TF / numpy / scipy call site it replaces.  Citations are
 * relative to the reference tree.
 *
 * Conventions
 *   - tensors are dense NHWC float32 in device memory (HBM) unless noted (the
 *     fused convolution reads its inputs in the G8 layout defined below);
 *     pointers are borrowed, the caller (PyTorch-ROCm or any HIP program) owns
 *     the memory;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work
 *     on it and is re-entrant per stream;
 *   - return value: MPG_OK or an MPG_ERR_* code; mpg_last_error() returns a
 *     thread-local message for the last failing call.
 */
#ifndef MPGAN_H
#define MPGAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mpg_stream_t;

enum { MPG_OK = 0, MPG_ERR_ARG = 1, MPG_ERR_HIP = 2, MPG_ERR_UNSUPPORTED = 3 };

/* activation ids: tf.nn.relu / lrelu (tools_wscale/GAN.py:733-737) / tf.nn.tanh */
enum { MPG_ACT_NONE = 0, MPG_ACT_RELU = 1, MPG_ACT_LRELU = 2, MPG_ACT_TANH = 3 };

/* arithmetic of the MFMA convolution: fp16 operands, fp32 accumulation.
 * F16X1: one fp16 term per operand (fast, ~3e-4 relative error per layer);
 * F16X3: hi/lo fp16 split of both operands, three MFMA products
 *        (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo), fp32-equivalent (~1e-6). */
enum { MPG_PREC_F16X1 = 1, MPG_PREC_F16F6 = 2, MPG_PREC_F16X3 = 3 };
/* F16F6: the fp16 product a_hi*w_hi plus the two correction products a_lo*w_hi and a_hi*w_lo
 *        on the block-scaled bf6 (e3m2) MFMA path at four times the fp16 rate per K, with true MX
 *        block scales (per lane and 32 K values, formed in registers from the fp16 fragments):
 *        ~2^-14 per operand at any input range, half of the F16X3 matrix work. */

/* flavour of a G8 tensor's second plane: fp16 lo (every launch reads it since round 3; the
 * {8 x fp8 hi | 8 x fp8 lo} flavour of rounds 1-2 is gone) */
enum { MPG_G8_F16 = 0 };

const char* mpg_last_error(void);
/* library / device probe: returns MPG_OK when a gfx950 device is usable. */
int mpg_device_info(int* cu_count, char* arch_name, int arch_name_len);
const char* mpg_version(void);

/* ------------------------------------------------------------------------
 * "G8" activation tensors: what fused convolutions exchange.
 *     [N][CG = ceil(C/8)][2 planes: hi, lo][H][W][8 x fp16],   value = hi + lo
 * (two fp16 numbers per value, exact to 2^-22; channels beyond C are zero).
 * Channel-group-major, so a tile row of one group is a contiguous run of
 * 16-byte pixels that LDS-DMA can stream.
 * ------------------------------------------------------------------------ */
size_t mpg_g8_bytes(int n, int h, int w, int c);
/* fp32 NHWC x[..., c_off : c_off+cin] -> G8 (flavour MPG_G8_*) with ceil(cin/8) groups */
int mpg_f32_to_g8(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int c_off, int cin,
                  int flavour, void* out);
/* G8 (flavour MPG_G8_F16, c channels) -> fp32 NHWC */
int mpg_g8_to_f32(mpg_stream_t stream, const void* g8, int n, int h, int w, int c, float* y);

/* ------------------------------------------------------------------------
 * Fused convolution  (replaces GAN.convolutional_layer = tf.nn.conv2d SAME +
 * bias + batch_norm(inference) + activation, tools_wscale/GAN.py:80-119,686-691;
 * the residual sum relu(convB(.) + conv1x1(.)) of resBlock,
 * GAN/multipassGAN-4x.py:517-523, GAN/multipassGAN-out.py:227-235;
 * GAN.pixel_norm, GAN.py:472-474; the nearest upsample in front of a block,
 * GAN.py:501-523,541; the channel concat of x_in_2, multipassGAN-out.py:357.)
 *
 *   y = post( act( sum_s conv_SAME(up_s(x_s)[groups g_off_s ..], W_s) + bias ) ) + post_add
 *
 * stride 1, SAME padding (pad_before = (k-1)/2, extra pad bottom/right).
 * Each segment s contributes one K-slice of the implicit GEMM: a residual
 * shortcut is a 1x1 segment, a channel concat is two segments with the same
 * kernel size, a fused nearest upsample is up_log2 > 0.
 * The result is written as fp32 NHWC (y), as G8 (y_g8), or both.
 * ------------------------------------------------------------------------ */
#define MPG_MAX_SEG 4

typedef struct mpg_conv_seg {
    const void* x;       /* G8 [N][cgroups][2][H>>up_log2][W>>up_log2][8] */
    const void* wpack;   /* from mpg_conv_pack_weights (same cout and prec as the launch) */
    int32_t cin;         /* channels consumed, starting at channel 8*g_off of x */
    int32_t cgroups;     /* channel groups of x */
    int32_t g_off;       /* first group consumed */
    int32_t kh, kw;      /* kernel size, 1..7 */
    int32_t up_log2;     /* fused nearest upsample: src = (y >> up_log2, x >> up_log2) */
    int32_t reserved0;   /* must be 0 */
    int32_t pad_hi;           /* 0: TF SAME, pad_before = (k-1)/2; 1: pad_before = k/2 (the data gradient of an even filter) */
} mpg_conv_seg;

typedef struct mpg_conv_desc {
    int32_t n, h, w;          /* output batch / height / width */
    int32_t cout;             /* 1..128 */
    int32_t nseg;             /* 1..MPG_MAX_SEG */
    mpg_conv_seg seg[MPG_MAX_SEG];
    const float* bias;        /* [cout] effective bias (bias and folded batch norm) or NULL */
    int32_t act;              /* MPG_ACT_* */
    float   leak;             /* lrelu leak (0.2 in the reference) */
    int32_t pixel_norm;       /* 1: y *= rsqrt(mean_c(y^2) + pn_eps) after act */
    float   pn_eps;           /* 1e-8 */
    const float* post_add;    /* optional fp32 [N,H,W,post_add_stride], channels post_add_coff.. added last */
    int32_t post_add_stride;
    int32_t post_add_coff;
    float*  y;                /* fp32 NHWC [N,H,W,cout] or NULL */
    void*   y_g8;             /* G8 (MPG_G8_F16) with ceil(cout/8) groups or NULL; at least one of the two outputs */
    int32_t prec;             /* MPG_PREC_* */
    int32_t reserved;         /* must be 0 */
    const float* in_amax;     /* NULL, or device pointer to the absolute maximum the segment inputs were scaled by
                               * (mpg_f32_to_g8_scaled): the sum is divided by the same power of two before bias
                               * and activation.  Used for gradients, whose magnitude is below the fp16 normal range. */
} mpg_conv_desc;

/* bytes of the packed weight image of one segment; 0 when the shape is not available at `prec` (kernel larger
 * than 7x7, cout > 128, or -- MPG_PREC_F16F6 only -- LDS images that do not fit, e.g. 7x7 with cout > 96:
 * pack such a segment for MPG_PREC_F16X3 instead). */
size_t mpg_conv_pack_size(int kh, int kw, int cin, int cout, int prec);

/* Pack W[kh,kw,w_cin_total,cout] (HWIO fp32, device; GAN.py:93) channels
 * [w_c_off, w_c_off+cin) into the MFMA fragment order, multiplying by the
 * equalised-LR constant `wscale` (GAN.py:664-668) and an optional per-output
 * channel scale (folded batch norm gamma/sqrt(var+eps), GAN.py:110). */
int mpg_conv_pack_weights(mpg_stream_t stream, const float* w_hwio, int kh, int kw,
                          int w_cin_total, int w_c_off, int cin, int cout,
                          float wscale, const float* cout_scale,
                          int prec, void* out, size_t out_bytes);

int mpg_conv2d_fused(mpg_stream_t stream, const mpg_conv_desc* desc);

/* A whole residual block whose three convolutions have <= 8 channels on either side, as ONE launch:
 *     y = act_b( conv_b( act_a( conv_a(up(x)) + bias_a ) ) + conv_s(up(x)) + bias_b )
 * (resBlock 0 and 3 of gen_resnet, 1 -> 2 -> 8 and 8 -> 2 -> 1 channels: GAN/multipassGAN-4x.py:505-526,560,564).
 * The middle tensor stays in LDS (as two mpg_conv2d_fused launches it made a round trip through HBM).  fp32
 * arithmetic on the exact (hi16 + lo16) inputs whatever `prec` says; wpack_* come from mpg_conv_pack_weights
 * with that `prec` (the fp32 tables of small layers follow the matrix-core image).  Odd filters only. */
typedef struct mpg_small_pair_desc {
    int32_t n, h, w;          /* output batch / height / width */
    const void* x;            /* G8 input [N][cgroups][2][H>>up_log2][W>>up_log2][8]; one channel group is read */
    int32_t cin, cgroups, g_off, up_log2;
    const void* wpack_a;      /* conv_a: kh_a x kw_a, cin -> cmid */
    int32_t kh_a, kw_a, cmid;
    const float* bias_a;      /* [cmid] or NULL */
    int32_t act_a;
    float   leak_a;
    const void* wpack_b;      /* conv_b: kh_b x kw_b, cmid -> cout */
    int32_t kh_b, kw_b;
    const void* wpack_s;      /* shortcut conv_s: kh_s x kw_s, cin -> cout, or NULL */
    int32_t kh_s, kw_s;
    const float* bias_b;      /* [cout] or NULL (bias of conv_b plus bias of conv_s) */
    int32_t act_b;
    float   leak_b;
    int32_t cout;
    float*  y;                /* fp32 NHWC [N,H,W,cout] or NULL */
    void*   y_g8;             /* G8 (one group) or NULL; at least one of the two */
    int32_t prec;             /* MPG_PREC_* the weights were packed for */
    int32_t reserved;         /* must be 0 */
} mpg_small_pair_desc;
int mpg_conv2d_small_pair(mpg_stream_t stream, const mpg_small_pair_desc* desc);

/* out[0] = max |x_i| (device float).  mpg_f32_to_g8_scaled converts x * 2^k with 2^k = the power of two that brings
 * that maximum into [2^8, 2^9); a convolution over such an input takes the same pointer in desc.in_amax. */
int mpg_absmax(mpg_stream_t stream, const float* x, size_t n, float* out);
int mpg_f32_to_g8_scaled(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int c_off, int cin,
                         int flavour, const float* amax, void* out);

/* Plain fp32 direct convolution on the vector ALUs, any stride / kernel /
 * channel count (tf.nn.conv2d SAME, GAN.py:686-691; used for the strided
 * discriminator convs GAN/multipassGAN-4x.py:607-614 and as an independent
 * check of the MFMA kernel).  w is HWIO fp32, multiplied by wscale on the fly.
 * y = act(conv(x, w*wscale) * cout_scale + bias). */
int mpg_conv2d_direct(mpg_stream_t stream, const float* x, int n, int h, int w, int cin,
                      const float* w_hwio, int kh, int kw, int cout, int stride_h, int stride_w,
                      float wscale, const float* cout_scale, const float* bias,
                      int act, float leak, float* y);

/* ------------------------------------------------------------------------
 * Resampling (legacy TF1 coordinates: src = dst * in/out, no half-pixel centres)
 * ------------------------------------------------------------------------ */
/* tf.image.resize_images(method=1) / kb.resize_images (GAN.py:517,541; multipassGAN-out.py:357) */
int mpg_resize_nearest(mpg_stream_t stream, const float* x, int n, int h, int w, int c,
                       float* y, int oh, int ow);
/* tf.image.resize_images(method=0) (GAN.py:541) */
int mpg_resize_bilinear(mpg_stream_t stream, const float* x, int n, int h, int w, int c,
                        float* y, int oh, int ow);
/* tf.image.resize_images(method=2): ResizeBicubic A=-0.75, 1/1024 weight table
 * (avg_depool(mode=2), multipassGAN-out.py:330) */
int mpg_resize_bicubic(mpg_stream_t stream, const float* x, int n, int h, int w, int c,
                       float* y, int oh, int ow);
/* tf.nn.max_pool VALID, k x k window, stride s (GAN.max_pool, GAN.py:152-159).  arg (may be NULL; one byte per output)
 * receives the window position dy * k + dx of the first maximum in scan order; mpg_max_pool_bwd sends dy there
 * (dx [n,h,w,c] is overwritten). */
int mpg_max_pool(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int k, int s, float* y,
                 unsigned char* arg);
int mpg_max_pool_bwd(mpg_stream_t stream, const float* dy, const unsigned char* arg, int n, int h, int w, int c,
                     int k, int s, float* dx);
/* tf.nn.avg_pool 2x2 VALID (GAN.py:162-169) */
int mpg_avg_pool2(mpg_stream_t stream, const float* x, int n, int h, int w, int c, float* y);
/* GAN.pixel_norm standalone (GAN.py:472-474) */
int mpg_pixel_norm(mpg_stream_t stream, const float* x, size_t npix, int c, float eps, float* y);
/* GAN.minibatch_stddev_layer (GAN.py:476-488): y[n,h,w,c+1] = concat(x, s[n % M]) with s[m] the mean over
 * (h,w,c) of the standard deviation over the G = min(group_size, n) members {g*M + m} of group m, M = n / G.
 * stat: M floats of scratch. */
int mpg_minibatch_stddev(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int group_size,
                         float* stat, float* y);
/* its gradient: dx[n,h,w,c] from dy[n,h,w,c+1] (the pass-through channels plus the statistic's dependence on
 * every member of the group); dstat: M floats of scratch. */
int mpg_minibatch_stddev_bwd(mpg_stream_t stream, const float* x, const float* dy, int n, int h, int w, int c,
                             int group_size, float* dstat, float* dx);
/* y = act(a + b) elementwise (tf.nn.relu(tf.add(..)), multipassGAN-4x.py:523); b may be NULL */
int mpg_add_act(mpg_stream_t stream, const float* a, const float* b, size_t n, int act, float leak, float* y);

/* ------------------------------------------------------------------------
 * Volume marshalling (generate3DUniForNewNetwork)
 * ------------------------------------------------------------------------ */
/* scipy.ndimage.zoom(v, factor on ONE axis, order=1, mode='constant')
 * (multipassGAN-out.py:401-421,465-485,527-547; multipassGAN-4x.py:1095-1103):
 * v viewed as [outer, n, inner] -> [outer, big, inner], src = o*(n-1)/(big-1). */
int mpg_axis_zoom_linear(mpg_stream_t stream, const float* v, size_t outer, int n, size_t inner,
                         float* out, int big);

/* numpy transpose of a [d0,d1,d2,c] array to axes order perm (a permutation of
 * 0,1,2; channels stay last), fused with the per-pass velocity channel
 * permutation chan_map (out[..., k] = in[..., chan_map[k]], NULL = identity;
 * multipassGAN-out.py:402-419,472-475) and the storage cutoff
 * (x < cutoff -> 0, multipassGAN-out.py:614-615; cutoff <= 0 disables). */
int mpg_volume_transpose(mpg_stream_t stream, const float* v, int d0, int d1, int d2, int c,
                         const int* perm, const int* chan_map, float cutoff, float* out);

/* add_adj_idcs channels (multipassGAN-out.py:423-436): out[i] = concat(in[i],
 * in[i-1][...,0], in[i+1][...,0]) with zeros at the ends.  in: [s, hw, c] -> out: [s, hw, c+2].
 * s_off/s_total let a rank build only its slice range of a sharded pass. */
int mpg_add_adjacent(mpg_stream_t stream, const float* in, int s_total, size_t hw, int c,
                     int s_off, int s_cnt, float* out);

/* Channel marshalling between the passes (multipassGAN-4x.py:278-283, 1095-1119: the velocity channels are cut out of the
 * low-res array, scaled by the upres factor and the velocity scale, and concatenated behind the density slices of the
 * previous pass): out[p][j] = (s[p][map[j]] * scale[j]) * scale2[j], where s is the channel-wise concatenation of
 * a [npix, ca] and b [npix, cb] (b may be NULL with cb = 0); map, scale and scale2 are HOST arrays of
 * cout <= MPG_GATHER_MAX_C entries (a NULL scale = all ones; two factors because the reference multiplies twice). */
#define MPG_GATHER_MAX_C 8
int mpg_channel_gather(mpg_stream_t stream, const float* a, int ca, const float* b, int cb, size_t npix,
                       const int* map, const float* scale, const float* scale2, int cout, float* out);

/* out[i] = v[i] < cutoff ? 0 : v[i]   (multipassGAN-4x.py:1156-1157) */
int mpg_cutoff(mpg_stream_t stream, const float* v, size_t n, float cutoff, float* out);

/* tf.nn.conv2d_transpose(x, W, output_shape = [n, h*stride_h, w*stride_w, cout], strides, "SAME") + bias + activation:
 * GAN.deconv2d / GAN.deconvolutional_layer (GAN.py:566-619, 703-708).  w_hwoi[kh,kw,cout,cin] is TensorFlow's
 * transposed-convolution filter layout (output channels before input channels).  y[n, h*stride_h, w*stride_w, cout].
 * fp32 vector-ALU gather, any stride / filter; the matrix-core route for stride 1 and 2 is composed above the ABI from
 * mpg_conv2d_fused (flipped / sub-pixel filters) and mpg_depth_to_space (ops.conv2d_transpose). */
int mpg_conv2d_transpose(mpg_stream_t stream, const float* x, int n, int h, int w, int cin, const float* w_hwoi,
                         int kh, int kw, int cout, int stride_h, int stride_w, float wscale, const float* bias,
                         int act, float leak, float* y);
/* tf.depth_to_space(x, r) (GAN.pixel_shuffle, GAN.py:554-560): x[n,h,w,c] -> y[n, h*r, w*r, c / r^2] */
int mpg_depth_to_space(mpg_stream_t stream, const float* x, int n, int h, int w, int c, int r, float* y);

/* ------------------------------------------------------------------------
 * Training step: what tf.gradients produces for
 * the layers of GAN.py, and the optimiser update.  fp32 NHWC device tensors;
 * geometry is tf.nn.conv2d SAME of an [n,h,w,cin] input (GAN.py:686-691).
 * ------------------------------------------------------------------------ */
/* d loss / d W for W stored unscaled (GAN.py:664-668): dw[kh,kw,cin,cout] =
 * wscale * sum_p x[p + tap] * dy[p].  dy is [n, ceil(h/sh), ceil(w/sw), cout]; dw is overwritten. */
int mpg_conv2d_wgrad(mpg_stream_t stream, const float* x, int n, int h, int w, int cin, const float* dy,
                     int cout, int kh, int kw, int stride_h, int stride_w, float wscale, float* dw);
/* The same weight gradient on the matrix cores for stride-1 filters (kh <= 7, kw in 1,3,4,5):
 * x and dy are converted to G8 (fp16 hi + lo planes, power-of-two scaled; mpg_f32_to_g8_scaled) inside
 * `workspace` (>= mpg_conv2d_wgrad_mfma_ws_bytes, 256-byte aligned), then contracted over pixels
 * with v_mfma_f32_32x32x16_f16, the pixel-major fragments read out of the channel-grouped G8 rows with the
 * transposing LDS read (ds_read_b64_tr_b16).  prec MPG_PREC_F16X3 (three products, fp32-grade) or MPG_PREC_F16X1.
 * dy_amax / x_amax (device scalars, may be NULL): max |dy| / max |x| when the caller already has them (the kernel that
 * produced dy returns it; a forward activation needs no scaling: pass a scalar holding 256.0f = scale 1) -- without
 * them each costs a reduction pass over its tensor. */
size_t mpg_conv2d_wgrad_mfma_ws_bytes(int n, int h, int w, int cin, int cout);
int mpg_conv2d_wgrad_mfma(mpg_stream_t stream, const float* x, int n, int h, int w, int cin,
                          const float* dy, int cout, int kh, int kw, float wscale, int prec,
                          void* workspace, size_t workspace_bytes, const float* dy_amax, const float* x_amax,
                          float* dw);
/* ... with both operands already in G8 (MPG_G8_F16, 16-byte aligned): the layer's forward input as mpg_conv2d_fused
 * read it, and the scaled dy the data-gradient convolution reads -- no conversion pass and no workspace.
 * x_amax / dy_amax: the device scalars the tensors were scaled with by mpg_f32_to_g8_scaled, NULL for an unscaled one.
 * This is what tf.gradients' Conv2DBackpropFilter does for the layers of GAN.py:686-691 under multipassGAN-4x.py:880-902. */
int mpg_conv2d_wgrad_g8(mpg_stream_t stream, const void* x_g8, int n, int h, int w, int cin, const void* dy_g8, int cout,
                        int kh, int kw, float wscale, int prec, const float* x_amax, const float* dy_amax, float* dw);
/* dy_amax: NULL, or a device float holding max |dy| already computed with mpg_absmax */
/* d loss / d x of the same convolution, any stride / filter size (the strided 4x4 discriminator
 * convs, multipassGAN-4x.py:607-614).  The filter is passed with its channel axes swapped,
 * w_hwoi[kh,kw,cout,cin].  Stride-1 filters can instead run mpg_conv2d_fused on dy with the
 * flipped w_hwoi and pad_hi = 1. */
int mpg_conv2d_dgrad(mpg_stream_t stream, const float* dy, int n, int h, int w, int cin,
                     const float* w_hwoi, int cout, int kh, int kw, int stride_h, int stride_w,
                     float wscale, float* dx);
/* GAN.fully_connected_layer (GAN.py:438-456): y[rows,cout] = act(x[rows,k] @ (w[k,cout] * wscale) + bias).
 * Its gradients are mpg_conv2d_wgrad / mpg_conv2d_dgrad with h = w = 1. */
int mpg_fc_forward(mpg_stream_t stream, const float* x, int rows, int k, const float* w, int cout,
                   float wscale, const float* bias, int act, float leak, float* y);
/* out[c] = sum over pixels of x[p, c]  (bias gradient, GAN.py:683) */
int mpg_channel_sum(mpg_stream_t stream, const float* x, size_t npix, int c, float* out);
/* ... with the blocks' sums added in a fixed order: partials holds >= mpg_bn_partials_floats(c) + c floats */
int mpg_channel_sum_ordered(mpg_stream_t stream, const float* x, size_t npix, int c, float* out, float* partials,
                            size_t partials_floats);
/* tf.contrib.layers.batch_norm(is_training=True) (GAN.py:110): batch mean / biased variance over
 * all pixels, y = act((x - mean) * rsqrt(var + eps) * gamma + beta); the moments are returned, and
 * the moving averages (the UPDATE_OPS, multipassGAN-4x.py:773-776) are advanced in place when
 * moving_mean / moving_var are given: moving = decay * moving + (1 - decay) * batch. */
int mpg_bn_train_fwd(mpg_stream_t stream, const float* x, size_t npix, int c, const float* gamma,
                     const float* beta, float eps, int act, float leak, float* y, float* batch_mean,
                     float* batch_var, float* moving_mean, float* moving_var, float decay);
/* The same with the blocks' partial sums kept in `partials` (>= mpg_bn_partials_floats(c) floats of device memory) and
 * added in block order instead of by atomics: the batch statistics -- and with them every ReLU mask of the step -- are
 * then the same bits on every run (a pre-activation within 1e-6 of zero otherwise changes side now and then, and one
 * flipped mask element moves the gradients upstream of it by 1 / sqrt(elements): DESIGN section 10). */
size_t mpg_bn_partials_floats(int c);
int mpg_bn_train_fwd_ordered(mpg_stream_t stream, const float* x, size_t npix, int c, const float* gamma,
                             const float* beta, float eps, int act, float leak, float* y, float* batch_mean,
                             float* batch_var, float* moving_mean, float* moving_var, float decay, float* partials,
                             size_t partials_floats);
/* gradient of the normalisation above (dy is taken before the activation).  amax (may be NULL) receives max |dx|:
 * the data- and weight-gradient convolutions that consume dx scale it by a power of two before the fp16 split
 * (mpg_absmax would re-read the tensor for it). */
int mpg_bn_train_bwd(mpg_stream_t stream, const float* dy, const float* x, size_t npix, int c,
                     const float* batch_mean, const float* batch_var, const float* gamma, float eps,
                     float* dx, float* dgamma, float* dbeta, float* amax);
/* ... with the blocks' partial sums of dbeta / dgamma kept in `partials` (mpg_bn_partials_floats(c) floats) and added in a
 * fixed order, as mpg_bn_train_fwd_ordered does for the statistics */
int mpg_bn_train_bwd_ordered(mpg_stream_t stream, const float* dy, const float* x, size_t npix, int c,
                             const float* batch_mean, const float* batch_var, const float* gamma, float eps,
                             float* dx, float* dgamma, float* dbeta, float* amax, float* partials, size_t partials_floats);
/* dx = dy * act'(.) written through the activation OUTPUT y (relu, lrelu GAN.py:733-737, tanh); amax as above */
int mpg_act_bwd(mpg_stream_t stream, const float* dy, const float* y, size_t n, int act, float leak, float* dx,
                float* amax);
/* gradient of GAN.pixel_norm (GAN.py:472-474) */
int mpg_pixel_norm_bwd(mpg_stream_t stream, const float* dy, const float* x, size_t npix, int c, float eps,
                       float* dx);
/* gradient of the nearest upsample by integer factors (GAN.py:517; dy is [n,oh,ow,c]) */
int mpg_resize_nearest_bwd(mpg_stream_t stream, const float* dy, int n, int oh, int ow, int c, float* dx,
                           int h, int w);
/* gradient of mpg_avg_pool2 (dx is [n,h,w,c], dy [n,h/2,w/2,c]) */
int mpg_avg_pool2_bwd(mpg_stream_t stream, const float* dy, int n, int h, int w, int c, float* dx);
/* lerp(x, y, t) = x + (y - x) * t with t already clipped to [0,1] (multipassGAN-8x.py:598-599); x NULL = zeros */
int mpg_lerp(mpg_stream_t stream, const float* x, const float* y, size_t n, float t, float* out);
/* tensorResample (multipassGAN-4x.py:398-441; 8x.py:547-595), 2D: out[b,i,j,:] = bilinear look-up of
 * value[b] at pos[b,i,j] = (y, x) in cell-centred coordinates, indices clamped to the grid when `clamp`
 * (script default); the advection step of the temporal discriminator inputs.  _bwd: gradient with
 * respect to value (dvalue is overwritten). */
int mpg_tensor_resample(mpg_stream_t stream, const float* value, const float* pos, int n, int h, int w, int c,
                        int clamp, float* out);
int mpg_tensor_resample_bwd(mpg_stream_t stream, const float* dy, const float* pos, int n, int h, int w, int c,
                            int clamp, float* dvalue);
/* One optimiser call of the 8x training loop (multipassGAN-8x.py:1305-1362 with the dynamic loss scaling of :490-541):
 * Adam on the elements of the flat buffer whose `mask` entry is non-zero (the variables of the current growing
 * stage; NULL = all), on the gradient times coef = exp(-ls_var ln 2) / total_grads (use_loss_scaling; the gradient
 * buffer holds d(loss * 2^ls_var)), applied only if every such product is finite -- otherwise nothing moves and
 * ls_var -= ls_dec; after an applied update ls_var += ls_inc and the optimiser's own step count t advances, from which
 * lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) is formed (lr: one float in DEVICE memory).  ema_shadow (or NULL):
 * MovingAverageOptimizer shadow of p, pulled towards the new values with (1 - ema_decay) on applied updates.
 * state: 8 floats of device memory owned by the optimiser, [0] = ls_var, [3] = t (initialise to 64 / 0).
 * Everything is decided on the device: a captured hipGraph of the iteration replays correctly. */
int mpg_adam_step_staged(mpg_stream_t stream, float* p, const float* grad, float* m, float* v, const float* mask,
                         size_t n, float* state, const float* lr, int total_grads, int use_loss_scaling,
                         float beta1, float beta2, float eps, float ls_inc, float ls_dec, float* ema_shadow,
                         float ema_decay);
/* GAN.advect (GAN.py:347-418): semi-Lagrangian / MacCormack advection of [n,h,w,c] fields (2D, h == w) for the
 * temporal-coherence branch with adv_mode 1 / 2 (multipassGAN-8x.py:1199,1225).
 *   mpg_advect_velocity: vel[n,hv,wv,cv] with channels (x, y, ..) -> out[n,h,w,2] = the (y, x) displacement the look-up
 *     uses: legacy-bilinear resize to [h,w], times max(h/hv, w/wv), averaged with its successor along its own axis
 *     (MAC -> cell centre, zero past the end), times dt * (+1, 0, -1)[b % 3] (:376-396); n is a multiple of 3.
 *   mpg_semi_lagrange: out[b,i,j,:] = sum over the 2x2 cells around q = (i + 0.5, j + 0.5) - sign * vel[b,i,j] of
 *     source * prod(1 - |q - index|), indices clamped to the grid, weights from the clamped indices (:175-204).
 *     _bwd: gradient with respect to source (dsource is overwritten).
 *   mpg_maccormack (one channel): the MacCormack correction forward + strength/2 (source - backward) where
 *     flags < 0.2, clamped back to `forward` where it leaves the [min, max] of the fluid cells around the truncated
 *     look-up position (:206-343, index clipping as written there: for three of the four corners the batch index is
 *     clipped to w-1 too).  keep (may be NULL) receives 1 where the correction term survived: the only place where
 *     d out / d source differs from the semi-Lagrangian one (TensorFlow differentiates tf.where by its branches). */
int mpg_advect_velocity(mpg_stream_t stream, const float* vel, int n, int hv, int wv, int cv, int h, int w, float dt,
                        float* out);
int mpg_semi_lagrange(mpg_stream_t stream, const float* source, const float* vel, int n, int h, int w, int c,
                      float vel_sign, float* out);
int mpg_semi_lagrange_bwd(mpg_stream_t stream, const float* dy, const float* vel, int n, int h, int w, int c,
                          float vel_sign, float* dsource);
int mpg_maccormack(mpg_stream_t stream, const float* source, const float* forward, const float* backward,
                   const float* flags, const float* vel, int n, int h, int w, float strength, float* out, float* keep);
/* ------------------------------------------------------------------------
 * Training-tile supply on device-resident frames (tools_wscale/tilecreator_t.py).  The random
 * decisions stay on the host (multi-pass-gan_amd/tiles_device.py); these are the array operations.
 *   mpg_tile_gather: out[b] = frames[f][z0:z0+tz, y0:y0+ty, x0:x0+tx, c0:c0+c] with (f, c0, z0, y0, x0) = table[5 b ..]
 *     (cutTile / getDatum, :403-450,562-574); frames [n_frames, z, y, x, cf], table in device memory.
 *   mpg_resample_affine: scipy.ndimage.affine_transform / zoom with order 1, mode 'constant', cval 0 of a [zs,ys,xs,c]
 *     array (:808-879): source coordinate = matrix9 * (z,y,x)_dst + offset3 in float64; channel_mix (c x c, or NULL)
 *     is applied to the interpolated channels (vector components rotate / scale with the grid, :700-760,846-856).
 *   mpg_tile_orient: crop [off, off + size) of src, axes permuted / reversed (the composed np.rot90 / np.flip of
 *     :762-806), channels permuted / negated (chan_map, chan_sign: the vector components follow the turn).
 *   mpg_semilagr_positions: getSemiLagrPosBatch (:1345-1378), 2D: vel [n_batch,h,w,3] (MAC, x,y,z), dt [n_batch] ->
 *     pos [n_batch, n_out, n_out, 2] = (y, x) - v dt. */
int mpg_tile_gather(mpg_stream_t stream, const float* frames, int n_frames, int z, int y, int x, int cf,
                    const int* table, int n_tiles, int tz, int ty, int tx, int c, float* out);
int mpg_resample_affine(mpg_stream_t stream, const float* src, int zs, int ys, int xs, int c, float* dst, int zd, int yd,
                        int xd, const double* matrix9, const double* offset3, const float* channel_mix);
int mpg_tile_orient(mpg_stream_t stream, const float* src, int zs, int ys, int xs, int c, const int* crop_off3,
                    const int* crop_size3, const int* perm3, const int* flip3, const int* chan_map,
                    const float* chan_sign, float* dst);
int mpg_semilagr_positions(mpg_stream_t stream, const float* vel, const float* dt, int n_batch, int h, int w, int n_out,
                           float* pos);

/* the reductions of the generator losses (multipassGAN-4x.py:754,764-765): out[0] = sum |a - b| (mode 0,
 * tf.reduce_mean(tf.abs(..)) after division by n) or sum (a - b)^2 (mode 1, 2 * tf.nn.l2_loss); b NULL = 0 */
int mpg_pair_reduce(mpg_stream_t stream, const float* a, const float* b, size_t n, int mode, float* out);
/* tf.train.AdamOptimizer update on a flat parameter buffer (multipassGAN-4x.py:880-902):
 * m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= lr_t * m / (sqrt(v) + eps), with
 * lr_t = lr * sqrt(1-b2^t)/(1-b1^t) computed by the caller and read from DEVICE memory (one float),
 * so that a captured hipGraph of the iteration can be replayed with a new step size. */
int mpg_adam_step(mpg_stream_t stream, float* p, const float* grad, float* m, float* v, size_t n,
                  const float* lr_t, float beta1, float beta2, float eps);

#ifdef __cplusplus
}
#endif
#endif /* MPGAN_H */
