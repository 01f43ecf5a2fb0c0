/* libflair_hip.so -- C ABI of the MI355X (gfx950) kernels behind FLAIR's sampling hot path.
 *
 * Drop-in boundary.  The reference has no FFI on this path: its
 * Python calls ATen / torchvision / flash-attn / mmcv operators.  Each entry point below
 * names the reference call site(s) it replaces (paths relative to the reference root,
 * guided_diffusion/...).  The only native precedent in the reference is the pybind
 * module dcn/src/deform_conv_ext.cpp:150-163, whose conventions are kept: the caller owns
 * and pre-allocates every buffer, inputs are contiguous, work is enqueued on the caller's
 * stream, and nothing synchronises with the host.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers;
 *   - every function returns FLAIR_OK (0) or a negative flair_status and records a
 *     message retrievable with flair_last_error() (thread-local);
 *   - functions are asynchronous on `stream`, re-entrant, allocate nothing, and may be
 *     captured into a hipGraph;
 *   - activations are "clip tensors": [T][H][W][C] with C innermost ("NHWC"), element
 *     type FLAIR_F32 or FLAIR_BF16; per-channel parameters (bias, norm scale) are f32;
 *   - `ld` arguments are the element distance between consecutive pixels, so a tensor
 *     may be a channel slice of a wider buffer (this is how torch.cat is avoided).
 */
#ifndef FLAIR_HIP_H
#define FLAIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

typedef enum {
    FLAIR_OK = 0,
    FLAIR_ERR_ARG = -1, /* bad shape / alignment / null pointer */
    FLAIR_ERR_HIP = -2  /* HIP runtime refused the launch */
} flair_status;

typedef enum { FLAIR_F32 = 0, FLAIR_BF16 = 1 } flair_dtype;

typedef enum {
    FLAIR_ACT_NONE = 0,
    FLAIR_ACT_RELU = 1,
    FLAIR_ACT_LRELU01 = 2, /* LeakyReLU(0.1) */
    FLAIR_ACT_SILU = 3,
    /* offset / mask activation of SecondOrderDeformableAlignment (unet_new.py:879-885) applied by the
     * convolution that PRODUCES the 27*G tap-major offset channels: with period = 3*G (act_period),
     * channel co -> act_param * tanh(v) if co % period < 2*period/3 (the (dy,dx) residues, act_param =
     * max_residue_magnitude), else sigmoid(v) (the modulation mask).  flair_dcn_align then takes the
     * finished values (raw_activated = 1) instead of re-deriving them per gathered channel group. */
    FLAIR_ACT_DCN_OFFSETS = 4,
    FLAIR_ACT_LRELU02 = 5, /* LeakyReLU(0.2): SFT scale / shift branches, codeformer.py:579-588 */
    FLAIR_ACT_GELU = 6     /* exact (erf) GELU: transformer MLP, codeformer.py:520-528 */
} flair_act;

/* Last error message of the calling thread ("" if none). */
const char* flair_last_error(void);
/* ABI version of this header: bumped whenever entry points are added or a struct changes
 * (3: round 2; 4: + face crop / paste entries, flair_bcast_weights; 5: round 4 entries; 6: + calibration launches).
 * The library may be used from several devices of one process: per-kernel launch attributes and
 * CU counts are cached per device. */
int flair_abi_version(void);

/* ------------------------------------------------------------------ convolution
 * Y = act(conv(X, W) + bias) + res0 + res1, times out_scale.  "same" zero padding,
 * stride 1, odd kernel sizes; KT > 1 convolves across frames (Conv3d), KT == 1 treats
 * frames as a batch (Conv2d / Conv1d-1x1 / Linear on pixels).
 * Replaces: unet_new.py:240-295 (ResBlock conv_nd 2-D/3-D, 1x1 skip), :359,:367,:409,:417
 * (qkv / proj_out Conv1d), :455-459 (TemporalAttention linears/proj), :659-668
 * (ResidualBlocksWithInputConv trunks, conv_last), :859-867 (offset conv stack), :993,
 * :1220 (stem / head convs); mmedit SPyNet 7x7 convs (unet_new.py:985).
 *   x[i]   : input segment i, [T][H][W] pixels of seg_c[i] channels, pixel stride seg_ld[i];
 *            the segments are the channel-wise concatenation the reference builds with
 *            th.cat (seg_c[i] must be a multiple of 32 (bf16) / 16 (f32); pad with zeros)
 *   w      : [Cout][KT*KH*KW][sum seg_c] in the activation dtype
 *   bias   : [Cout] f32 or NULL;  res0/res1: [T][H][W][Cout-slice] or NULL
 *   frame_bias : [T][frame_bias_ld] f32 or NULL, added before the activation (the per-frame
 *            embedding terms `h + emb_out` of unet.py:246 and sr3.py:82)
 *   y      : [T][H][W] pixels, y_ld elements apart, Cout (multiple of 4) written per pixel
 *   workspace : optional device scratch of flair_conv_workspace_bytes(p) bytes; when given,
 *            deep-K convolutions on few pixels (the 16x16..4x4 levels) are split over K
 *            (f32 partial sums + a reduce/epilogue launch) */
typedef struct {
    int dtype;
    int T, H, W;
    int KT, KH, KW;
    int Cout;
    int nseg;
    int seg_c[4];
    int seg_ld[4];
    int y_ld;
    int res_ld[2];
    int act;
    float out_scale;
    int frame_bias_ld;
    int stride; /* spatial stride 1 (default when 0) or 2: T,H,W describe the INPUT, the output is
                 * ceil(H/stride) x ceil(W/stride) (PyTorch Conv2d(k, stride, padding=k//2)) */
    float act_param; /* FLAIR_ACT_DCN_OFFSETS: max_residue_magnitude */
    int act_period;  /* FLAIR_ACT_DCN_OFFSETS: 3 * deform_groups (multiple of 24) */
    int asym_pad;    /* 1 (stride 2 only): taps at stride*i .. stride*i + K-1, zeros past the bottom / right edge =
                      * F.pad(x, (0, 1, 0, 1)) + Conv2d(k, stride 2, padding 0), CodeFormer's Downsample
                      * (codeformer.py:138-149); 0: taps centred (padding k//2) */
    int reflect_pad; /* 1: taps that fall outside the frame read the mirrored pixel (nn.ReflectionPad2d(k//2) + Conv2d
                      * without padding: ParseNet's ConvLayer, facelib/parsing/parsenet.py:92-104) instead of zeros;
                      * im2col kernels only (any stride) */
} flair_conv_params;

size_t flair_conv_workspace_bytes(const flair_conv_params* p);
int flair_conv_nhwc(const flair_conv_params* p, const void* const* x, const void* w,
                    const float* bias, const float* frame_bias, const void* res0, const void* res1,
                    void* y, void* workspace, size_t workspace_bytes, hipStream_t stream);
/* Which kernel variant flair_conv_nhwc launches for these parameters (profiling aid):
 * im2col tiles 0 = 128 couts x 128 pixels, 1 = 64 x 128, 2 = 64 x 64 per workgroup;
 * halo kernel (3x3 spatial taps, W % 32 == 0) 3 / 4 / 5 = 8 / 4 / 2 image rows per workgroup;
 * 6 / 7 = K-split halo kernel with 8 / 4 rows (launches of exactly 256 workgroups: one frame). */
int flair_conv_variant(const flair_conv_params* p);


/* ------------------------------------------------- fused chain of two 3x3 convolutions
 * M = actA(conv3x3(cat(x), WA) + biasA)   (C = 64 or 128 channels, never leaves the LDS)
 * Y = (actB(conv3x3(M, WB) + biasB) + res0 + res1) * out_scale        (CoutB channels)
 * per frame, zero "same" padding on both convolutions.  With WA == NULL there is no stage A: the
 * single input segment (C channels) is staged once per tile and kept resident while CoutB is walked
 * in blocks of 64 (wide-output convolutions: the c -> 27*G offset convolution).  For bf16, C = 64, W % 32 == 0,
 * H % 8 == 0, CoutB % 8 == 0 and no residual inputs this form runs on conv_resident_kernel (round 4: halo by LDS-DMA,
 * three-stage weight ring, the epilogue of one 64-cout block issued between the MFMAs of the next); same results.
 * Replaces pairs of dependent launches of the BasicVSR++ recurrence: conv_offset[2]+[4] and
 * conv_offset[4]+[6] of SecondOrderDeformableAlignment (unet_new.py:859-867), conv1+conv2 of mmedit's
 * ResidualBlockNoBN inside ResidualBlocksWithInputConv with its residual adds (unet_new.py:659-668,
 * :731-739) -- same rounding points as the two launches (the intermediate is rounded to the element type).
 *   x[i], seg_c, seg_ld : input segments as in flair_conv_nhwc (T frames of H x W, W % 8 == 0)
 *   wA : [C][9][sum seg_c] or NULL;  wB : [CoutB][9][C];  biases f32 or NULL
 *   res0 / res1 : [T][H][W][CoutB-slice] or NULL, added after actB;  y : pixel stride y_ld */
typedef struct {
    int dtype;
    int T, H, W;
    int C;      /* channels of the intermediate (= output channels of stage A) */
    int CoutB;
    int nseg;
    int seg_c[4];
    int seg_ld[4];
    int y_ld;
    int res_ld[2];
    int actA, actB;
    float out_scale;
    float act_param; /* actB == FLAIR_ACT_DCN_OFFSETS: max_residue_magnitude */
    int act_period;  /* actB == FLAIR_ACT_DCN_OFFSETS: 3 * deform_groups */
} flair_chain_params;

int flair_conv_chain(const flair_chain_params* p, const void* const* x, const void* wA, const float* biasA,
                     const void* wB, const float* biasB, const void* res0, const void* res1, void* y,
                     hipStream_t stream);


/* ------------------------------------------------------- GroupNorm + SiLU (+ FiLM)
 * y = act( GroupNorm(x) * (1 + scale) + shift ), statistics joint over
 * (C/groups) x frames_per_stat x H x W (frames_per_stat = T for the FLAIR video UNet,
 * 1 for per-frame norms).  The input may be the channel concatenation of two stored
 * tensors (x0: first c0 channels, x1: the remaining C - c0).  resample: 0 none,
 * 1 = 2x2 average pooling of the activated result, 2 = nearest 2x upsampling; `raw`
 * (optional) receives the identically resampled un-normalised input.
 * Replaces: nn_new.py:17-19 GroupNorm32 behind nn.py:359-367 LazyReshaper3D, with the
 * SiLU / (1+scale)*h+shift / Upsample / Downsample steps of unet_new.py:237-329, the
 * attention norms (:358,:408,:461) and the head (:1216-1222).
 *   gamma, beta : [C] f32;  film : [F][film_ld] f32 rows of (scale[C] | shift[C]) or NULL
 *   workspace   : flair_groupnorm_workspace_bytes(p) bytes of device scratch
 *   limits      : C <= 2048 (f32) / 4096 (bf16): one 16-byte channel piece per thread of a <= 512-thread row group
 *                 (FLAIR_ERR_ARG "too wide" beyond that); C a multiple of 4 (f32) / 8 (bf16). */
typedef struct {
    int dtype;
    int C, c0;   /* channels in total / in segment 0 */
    int ld0, ld1;
    int groups;
    int F, H, W;
    int frames_per_stat;
    float eps;
    int act;
    int resample;
    int y_ld, raw_ld;
    int film_ld;
} flair_gn_params;

/* y (and raw) must not alias x0 / x1. */
size_t flair_groupnorm_workspace_bytes(const flair_gn_params* p);
int flair_groupnorm_nhwc(const flair_gn_params* p, const void* x0, const void* x1,
                         const float* gamma, const float* beta, const float* film, void* y,
                         void* raw, void* workspace, hipStream_t stream);

/* ----------------------------------------------------------------- API-edge layout
 * (N,C,H,W) f32 <-> channel slice [coff, coff+C) of an NHWC clip tensor.  Replaces the
 * rearrange/cat/type casts at unet_new.py:1330-1331,1353,1361-1362. */
int flair_nchw_f32_to_nhwc(const float* src, int N, int C, int H, int W, void* dst, int dtype,
                           int dst_ld, int dst_coff, hipStream_t stream);
int flair_nhwc_to_nchw_f32(const void* src, int dtype, int src_ld, int src_coff, int N, int C,
                           int H, int W, float* dst, hipStream_t stream);

/* timestep_embedding (nn_new.py:103-121): out[n] = [cos(t_n f_i) | sin(t_n f_i)], f32;
 * sin_first != 0 gives sr3's PositionalEncoding order [sin | cos] (sr3.py:45-60). */
int flair_timestep_embedding(const float* t, int N, int dim, float max_period, int sin_first,
                             float* out, hipStream_t stream);

/* y = act_out(act_in(x) @ w^T + bias), f32, M <= 32 rows (one per frame).  Replaces the
 * time_embed MLP (unet_new.py:980-984,1351) and every emb_layers Linear (:258-264,:399). */
int flair_linear_f32(const float* x, int M, int K, const float* w, const float* bias, int N,
                     int act_in, int act_out, float* y, int y_ld, hipStream_t stream);

/* ------------------------------------------------------------------- sampler step
 * flair_predict_xstart: x0 = clamp(c_recip*x - c_recipm1*eps) with eps = first C of Cm
 * model channels (gaussian_diffusion.py:278-327).  flair_sampler_update: data
 * consistency x0 -= gamma*restored (+clamp), aux blend w*x0 + (1-w)*clamp(aux), prev_recon
 * pinning of the first prev_frames frames, eps' and the generalised DDIM update
 * (gaussian_diffusion.py:465-515).  All tensors (N,C,H,W) f32, flat length n. */
typedef struct {
    float gamma, w_aux;
    float sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod;
    float sqrt_alphas_cumprod_prev, sqrt_one_minus_alphas_cumprod_prev;
    float sqrt_one_minus_rho, sqrt_rho;
    int clip_denoised, nonzero;
    long frame_elems;
    int frames, prev_frames;
} flair_sampler_coefs;

int flair_predict_xstart(const float* x, const float* model_out, int N, int C, int Cm, int H, int W,
                         float c_recip, float c_recipm1, int clip, float* x0, hipStream_t stream);
int flair_sampler_update(const flair_sampler_coefs* c, const float* x, float* x0,
                         const float* restored, const float* aux, const float* z,
                         const float* prev_recon, long n, float* x_prev, hipStream_t stream);
/* out = clamp(a*x + b*y, lo, hi) on flat f32 tensors (y may be NULL): q_sample,
 * posterior mean, eps-from-x0 (gaussian_diffusion.py:206-248,361-365). */
int flair_axpby_f32(const float* x, const float* y, float a, float b, float lo, float hi, long n,
                    float* out, hipStream_t stream);
/* LEARNED_RANGE model variance (gaussian_diffusion.py:285-292) from channels [C,2C) of the
 * (N,2C,H,W) model output. */
int flair_learned_range_variance(const float* model_out, int N, int C, int H, int W, float min_log,
                                 float max_log, float* variance, float* log_variance,
                                 hipStream_t stream);
/* y[p][c] = (clamp(x[p][c]*a + b, lo, hi) - sub[c]) * mul[c] on f32 pixels (SPyNet input
 * normalisation: unet_new.py:1300 and mmedit SPyNet's mean/std). */
int flair_affine_channels_f32(const float* x, int x_ld, int C, long P, float a, float b, float lo,
                              float hi, const float* sub, const float* mul, float* y, int y_ld,
                              hipStream_t stream);
/* y = x + sigmoid(gate[f][c]) * (m - x): TemporalWrapper2's emb-gated residual mix
 * (sr3.py:203-226).  x, m, y: [F][HW] pixels of C channels; gate: [F][gate_ld] f32 logits. */
int flair_gated_blend(const void* x, int x_ld, const void* m, int m_ld, const float* gate, int gate_ld,
                      int dtype, int C, int F, long HW, void* y, int y_ld, hipStream_t stream);
/* y = act(x0 + x1) on P pixels of C channels (x1 may be NULL): the residual sum of the RetinaFace detector's ResNet-50
 * Bottleneck (`out += identity; out = relu(out)`: torchvision resnet.py as built by
 * facelib/detection/retinaface/retinaface.py:99-102) and FPN's lateral sums (retinaface_net.py:88-94). */
int flair_add_act_nhwc(const void* x0, int x0_ld, const void* x1, int x1_ld, int dtype, int C, long P,
                       int act, void* y, int y_ld, hipStream_t stream);
/* nn.MaxPool2d(3, stride 2, padding 1) on [F][H][W][C] -> [F][(H+1)/2][(W+1)/2][C] (the stem of that ResNet-50). */
int flair_maxpool3x3s2_nhwc(const void* x, int x_ld, int dtype, int F, int H, int W, int C, void* y,
                            int y_ld, hipStream_t stream);
/* x[f][p][c] += bias[f][c]  (AttentionbottleBlock h + emb_out, unet_new.py:426-428). */
int flair_add_frame_bias(void* x, int dtype, int ld, int C, int F, long HW, const float* bias,
                         int bias_ld, hipStream_t stream);
/* dst[p][coff+c] = cast(src[p][c]): f32 flow fields into a conv input segment (th.cat at
 * unet_new.py:875). */
int flair_cast_channels(const float* src, int src_ld, int C, long P, void* dst, int dtype,
                        int dst_ld, int dst_coff, hipStream_t stream);
/* x[p][:] *= wmap[p]  (BasicVSR++ per-pixel vsrpp_weights, unet_new.py:739). */
int flair_scale_pixels(void* x, int dtype, int ld, int C, long P, const float* wmap,
                       hipStream_t stream);

/* ---------------------------------------------------------------------- attention
 * Spatial attention over the L tokens of each frame, per head of width 64
 * (unet_new.py:540-605).  qkv: [frames][L][ld]; q/k/v of head h start at channel
 * {q,k,v}_off + h*head_stride (legacy order: 0/64/128 + h*192; new order: 0/C/2C + h*64);
 * out: [frames][L][out_ld], channel h*64 + d.  scale multiplies q.k (1/sqrt(64)). */
typedef struct {
    int dtype;
    int frames, L, heads, head_dim;
    int ld, out_ld;
    int q_off, k_off, v_off, head_stride;
    float scale;
} flair_attn_params;
int flair_qkv_attention(const flair_attn_params* p, const void* qkv, void* out, hipStream_t stream);

/* Temporal window attention per pixel (unet_new.py:473-515 + nn.py:370-386): query = own
 * frame, keys/values = the window-1 neighbouring frames (replicate padded), softmax scale
 * `scale`.  qkv: [T][H*W][ld] holding q|k|v (C each); kpos: [window-1][C] f32 added to the
 * keys of each window slot (W_k applied to the positional code of that offset);
 * round_fp16 reproduces the reference's fp16 cast of q/k/v and of the result. */
typedef struct {
    int dtype;
    int T, H, W, C, window;
    int ld, out_ld;
    int round_fp16;
    float scale;
} flair_tattn_params;
int flair_temporal_attention(const flair_tattn_params* p, const void* qkv, const float* kpos,
                             void* out, hipStream_t stream);

/* ------------------------------------------------------------------ warps / resize
 * flow_warp (mmedit; unet_new.py:706,718,719): y[p] = bilinear(x, p + flow[p]),
 * flow [F][H][W] pixels of (dx,dy) f32, flow_ld floats apart; align_corners=True, zeros
 * (border=0) or border padding. */
int flair_flow_warp(const void* x, int dtype, int x_ld, const float* flow, int flow_ld, int F,
                    int H, int W, int C, int border, void* y, int y_ld, hipStream_t stream);
/* out = f1 + warp(f2, f1) on flow fields (unet_new.py:716-718). */
int flair_flow_compose(const float* f1, const float* f2, int F, int H, int W, float* out,
                       hipStream_t stream);
/* The two warps of one BasicVSR++ propagation step with both flows given (unet_new.py:706,719):
 * cond1 = warp(prop, flow1), cond2 = warp(feat2, flow2); flow2 == NULL: first-order step (cond1 only).
 * The second-order flow (unet_new.py:716-718) depends on the optical flows alone, so the caller composes it
 * once per clip (flair_vsrpp_prep on the first denoising step) and reuses it for the other steps. */
int flair_vsrpp_warp2(const void* prop, int prop_ld, const void* feat2, int feat2_ld, const float* flow1,
                      const float* flow2, int dtype, int H, int W, int C, void* cond1, int cond1_ld,
                      void* cond2, int cond2_ld, hipStream_t stream);
/* One BasicVSR++ propagation step's alignment inputs in one launch (unet_new.py:704-722):
 * cond1 = warp(prop, flow1); flow2 = flow1 + warp(flow_prev, flow1); cond2 = warp(feat2, flow2);
 * flowpad[p][0..3] = (flow1, flow2) in the activation dtype.  flow_prev == NULL: first-order
 * step (cond2 / flow2_out untouched, flowpad[2..3] = 0).  One frame: tensors are [H][W][*]. */
int flair_vsrpp_prep(const void* prop, int prop_ld, const void* feat2, int feat2_ld,
                     const float* flow1, const float* flow_prev, int dtype, int H, int W, int C,
                     void* cond1, int cond1_ld, void* cond2, int cond2_ld, float* flow2_out,
                     void* flowpad, int pad_ld, hipStream_t stream);
/* mode 0/1: bilinear (align_corners False/True), 2: bicubic (A=-0.75), 3: 2x2 avg-pool,
 * 4: nearest;
 * channel 0 / 1 of the result are multiplied by scale_c0 / scale_c1 (flow rescaling). */
int flair_resize_nhwc(const void* x, int dtype, int x_ld, int F, int Hi, int Wi, int C, int mode,
                      int Ho, int Wo, void* y, int y_ld, float scale_c0, float scale_c1,
                      hipStream_t stream);

/* ------------------------------------------------- deformable alignment (DCNv2)
 * Fused offset/mask activation + modulated deformable 3x3 conv (unet_new.py:877-898;
 * same operator as dcn/src/deform_conv_cuda_kernel.cu:571-633 + deform_conv_cuda.cpp:540-560).
 *   x0|x1 : the two halves of the 2c input channels (feat_prop | feat_n2)
 *   raw   : conv_offset output, 27*G channels, pixel stride raw_ld, in TAP-MAJOR order
 *           (tap k owns the 3*G contiguous channels [3*G*k, 3*G*(k+1))):
 *             raw[3*G*k + 2*g + {0,1}] = (dy,dx) pre-activation of group g / tap k,
 *             raw[3*G*k + 2*G + g]     = mask pre-activation
 *           (the reference's o1|o2|mask order is 2*(g*9+k)+{0,1} and 18*G + g*9 + k; the
 *           caller permutes the output channels of the last conv_offset convolution once,
 *           when it packs that layer's weights).  G in {8, 16}; Cin/G a power of two.
 *   flow1, flow2 : [F][H][W][2] f32 or NULL (zero);  w : [Cout][9][Cin];  bias f32 */
typedef struct {
    int dtype;
    int F, H, W;
    int Cin, Cout, G;
    int x_ld[2];
    int raw_ld, y_ld;
    float max_residue_magnitude;
    int raw_activated; /* 1: raw already holds max_residue_magnitude*tanh / sigmoid values
                        * (produced with FLAIR_ACT_DCN_OFFSETS); 0: pre-activations */
} flair_dcn_params;
int flair_dcn_align(const flair_dcn_params* p, const void* x0, const void* x1, const void* raw,
                    const float* flow1, const float* flow2, const void* w, const float* bias,
                    void* y, hipStream_t stream);

/* ------------------------------------------------- degradation operators (restore_fn)
 * Images here are the sampler's (N,C,H,W) f32 tensors.
 *
 * flair_depthwise_filter: pseudoSR's Filter_Layer (pseudoSR.py:15-44,174-246): every plane is
 * cross-correlated with one kh x kw filter after replication padding:
 *   out[i][j] = sum_{u,v} K[u][v] * IN(i*out_stride + out_offset + u - pad, j*... + v - pad)
 * where IN clamps to the virtual image of size (Hin*stuff, Win*stuff) whose samples are zero
 * except at coordinates == stuff_offset (mod stuff) (zero-stuffing up-sampler).
 *   Down  : pad 4, out_stride 4, out_offset pre_stride;  InvHtH: pad 19;  Up: stuff 4.
 * reflect != 0 uses F.pad(mode="reflect") instead of replication (imresize_efficient, the
 * forward operator A: imresize_pseudoSR.py:163-178). */
int flair_depthwise_filter(const float* x, int planes, int Hin, int Win, const float* filt, int kh,
                           int kw, int pad, int out_stride, int out_offset, int stuff,
                           int stuff_offset, int Hout, int Wout, int reflect, float* y,
                           hipStream_t stream);
/* jpeg_decode(jpeg_encode(x, qf), qf) of jpeg.py:72-167 on (N,3,S,S) images in [-1,1].
 * q_luma, q_chroma, dct8 are HOST arrays of 64 floats (quantisation tables of
 * general_quant_matrix, jpeg.py:35-65, and the 8x8 orthonormal DCT-II matrix); workspace:
 * N*3*S*S device floats.  luma_q / chroma_q (both or neither, may be NULL): the quantised integer
 * levels round(DCT / q) that jpeg_encode returns (jpeg.py:108-114), as f32 planes (N,1,S,S) and
 * (N,2,S/2,S/2) in the reference's block layout. */
int flair_jpeg_roundtrip(const float* x, int N, int S, const float* q_luma, const float* q_chroma,
                         const float* dct8, float* workspace, float* y, float* luma_q, float* chroma_q,
                         hipStream_t stream);
/* C[b] = A[b] (MxK) * B[b] (KxN), row-major f32; a stride of 0 shares the matrix across the
 * batch.  SRConv's separable U/V products (restore_util.py:102-227). */
int flair_matmul_f32(const float* A, long a_batch_stride, const float* B, long b_batch_stride,
                     float* C, int batch, int M, int N, int K, hipStream_t stream);
/* Resizer (resizer.py:54-73) along one axis of a tensor viewed [outer][Lin][inner]:
 * y[o][i][n] = sum_k w[k][i] * x[o][fov[k][i]][n]; fov int32 [taps][Lout], w f32 [taps][Lout]. */
int flair_gather_mac_f32(const float* x, long outer, int Lin, long inner, const int* fov,
                         const float* w, int taps, int Lout, float* y, hipStream_t stream);

/* ------------------------------------------------------------- CodeFormer auxiliary prior
 * (guided_diffusion/codeformer.py, called from gaussian_diffusion.py:471-496 as
 * aux_model(pred_xstart, t, x)).  Its convolutions, GroupNorms, 1x1 projections and 8 x 64 multi-head attention
 * run on the entry points above; these are the pieces only the prior needs. */
/* nn.LayerNorm(C) over the channels of each of `rows` pixels (codeformer.py:541-542, :639), eps inside the
 * square root.  Optional second output y2 = y + pos[row % pos_rows][C] (q = k = norm(x) + pos, :561-562). */
int flair_layernorm_nhwc(const void* x, int dtype, int x_ld, long rows, int C, const float* gamma,
                         const float* beta, float eps, void* y, int y_ld, const float* pos, int pos_rows,
                         void* y2, int y2_ld, hipStream_t stream);
/* Attention with heads of any width (AttnBlock.forward, codeformer.py:217-241: one head of C = 512 over the 256
 * pixels of the 16x16 level).  Same layout contract as flair_qkv_attention; head_dim a multiple of 8 (bf16) / 4
 * (f32), 64 * (head_dim + L) bytes of LDS <= 128 KiB. */
int flair_attention_wide(const flair_attn_params* p, const void* qkv, void* out, hipStream_t stream);
/* idx[row] = argmax_n logits[row][n] (softmax + topk(1) of codeformer.py:727-728; first index on ties) and
 * y[row][0..D) = codebook[idx[row]] (VectorQuantizer.get_codebook_feat, :82-94).  codebook: [N][D] f32.
 * forced_idx (or NULL): use these indices instead of the arg-max (tests).  idx may be NULL. */
int flair_argmax_codebook(const void* logits, int dtype, int ld, long rows, int N, const float* codebook, int D,
                          const int* forced_idx, int* idx, void* y, int y_ld, hipStream_t stream);
/* adaptive_instance_normalization(content, style) of codeformer.py:437-470 on [frames][HW][C] tensors:
 * per (frame, channel) mean and sqrt(unbiased variance + eps) of both, y = (content - mc) / sc * ss + ms. */
int flair_adain_nhwc(const void* content, int c_ld, const void* style, int s_ld, int dtype, int frames, int HW,
                     int C, float eps, void* y, int y_ld, hipStream_t stream);
/* Fuse_sft_block tail (codeformer.py:595-596): y = dec + w * (dec * scale + shift) over n dense elements. */
int flair_sft_fuse(const void* dec, const void* scale, const void* shift, float w, int dtype, long n, void* y,
                   hipStream_t stream);

/* ------------------------------------------------------------- un-aligned prior branch: face crop / inverse paste
 * (second half; gaussian_diffusion.py:476-493 calls facelib/utils/face_restoration_helper.py:225-254
 * and :264-335 every denoising step and round-trips the frames through numpy / OpenCV).  Images are the sampler's
 * (N,C,H,W) f32 tensors; the affine matrices are an INPUT (computed once per window, scripts/video_sample.py:446-448).
 * OpenCV's arithmetic is restated from its published algorithm (cv2 absent here: parity unpinned). */
/* cv2.warpAffine(src, M, (Wd, Hd), flags=INTER_CUBIC, borderMode=BORDER_CONSTANT, borderValue=border) per image n.
 * minv: DEVICE [N][6] doubles, the dst -> src matrix warpAffine derives from M (its inverse, computed in double on the
 * host exactly as imgwarp.cpp does).  src: [N][C][Hs][Ws] f32, or f64 when src_is_f64 (masks).  border: HOST floats [C].
 * pre = 1 applies clamp((x + 1) / 2, 0, 1) * 255 to every source sample (face_restoration_helper.py:230), post = 1 applies
 * clamp((y / 255 - 0.5) / 0.5, -1, 1) to the result (:246-253); dst: [N][C][Hd][Wd] f32.  C <= 4. */
int flair_warp_affine_cubic(const void* src, int src_is_f64, int N, int C, int Hs, int Ws, const double* minv,
                            int Hd, int Wd, const float* border, int pre, int post, float* dst, hipStream_t stream);
/* The paste mask of inverse_faces (face_restoration_helper.py:283-317): mask = lut[parse_idx] (MASK_COLORMAP, DEVICE
 * doubles [nlut]), `repeats` x cv2.GaussianBlur(mask, (ksize, ksize), sigma) in float64 with BORDER_REFLECT_101 (kern:
 * DEVICE doubles [ksize] = cv2.getGaussianKernel), the `edge` outermost pixels zeroed, / div.  parse_idx: [N][H][W]
 * int32 (flair_argmax_codebook's idx); tmp, mask: [N][H][W] doubles (mask is the result). */
int flair_face_mask_blur(const int* parse_idx, int N, int H, int W, const double* lut, int nlut, const double* kern,
                         int ksize, int repeats, int edge, double div, double* tmp, double* mask, hipStream_t stream);
/* x0 * (1 - mask) + face * mask (gaussian_diffusion.py:491); x0, face, out: [N][C][H][W] f32; mask: [N][1][H][W] f32. */
int flair_face_blend(const float* x0, const float* face, const float* mask, int N, int C, int H, int W, float* out,
                     hipStream_t stream);

/* ------------------------------------------------------------- multi-GPU start-up
 * The only collective of the path: in-place RCCL broadcast of the kernel-native weight blob (flair_amd.checkpoint.export_packed:
 * bf16 [Cout][taps][Cin] conv weights, batched embedding matrix, f32 biases ...) from `root` to every rank of `rccl_comm`
 * (an ncclComm_t of the host process), enqueued on `stream`.  Replaces dist_util.py:40-79 (pickled-chunk load_state_dict +
 * one broadcast per parameter).  librccl is bound with dlopen at the first call.  Clips are independent after that: no
 * data-path collective exists. */
int flair_bcast_weights(void* blob, size_t bytes, int root, void* rccl_comm, hipStream_t stream);

/* ------------------------------------------------------------- calibration launches (measurement only)
 * No counterpart in the reference: they exist so that bench.py can print, beside every roofline fraction against the data-sheet
 * peaks, what the device it ran on sustains with nothing in the way (profiles/r04_attainable_peaks.txt).
 * flair_probe_matrix_rate: workgroups_per_cu x CU-count workgroups of four waves, each wave `iters` x 16 independent
 *   v_mfma_f32_32x32x16_bf16 out of registers; *flop_out (host, optional) = FLOPs of the launch; sink: >= 4 bytes of device memory
 *   (never written).  flair_probe_stream_rate: mode 0 reads `bytes` from src, 1 writes `bytes` to dst, 2 copies src -> dst
 *   (16-byte accesses, grid-stride).  The caller times them with events on `stream`. */
int flair_probe_matrix_rate(int iters, int workgroups_per_cu, float* sink, double* flop_out, hipStream_t stream);
int flair_probe_stream_rate(const void* src, void* dst, size_t bytes, int mode, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FLAIR_HIP_H */
