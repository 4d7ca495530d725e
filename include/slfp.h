/*
 * slfp.h -- C ABI of libslfp_hip.so: the MI355X (gfx950) SLFP<3,4> / SFP<3,3> quantized
 * conv2d forward path.
 *
 * This is the drop-in boundary for the hot path of happyxtt/CNNs_SLFP_quantization:
 *   utils/sfp_quant.py    quantize_act / quantize_weight (fake-quant codecs)
 *   utils/conv2d_func.py  Conv2d_Q.forward (conv2d_Q, conv2d_Q_bias), Linear_Q.forward
 * Every entry point takes plain device pointers and sizes (no torch types), is
 * asynchronous on the hipStream_t passed as `void* stream` (NULL = the default stream),
 * never allocates, never synchronises, never throws, and returns an int status
 * (0 = ok, < 0 = error; slfp_last_error() gives the text).  The reference raises Python
 * exceptions (an assert at sfp_quant.py:138 and shape errors from F.conv2d); the host
 * binding maps non-zero statuses to the same exception types.
 *
 * Threading: no mutable state on the launch path except a thread-local last-error string and lock-protected caches of
 * derived constants (threshold tables per scale, occupancy / CU count per device); safe to call from any host thread; one
 * device per process is assumed (multi-GPU = one process per GPU, as torch.distributed/RCCL launches them).
 * Environment: a handful of SLFP_* experiment switches (csrc/slfp_host.hpp: Switches) are read ONCE when the library is
 * loaded -- never per launch; slfp_debug_reload_switches() re-reads them (profiling tools only).
 */
#ifndef SLFP_H_
#define SLFP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLFP_ABI_VERSION 1

/* status codes */
#define SLFP_OK 0
#define SLFP_ERR_BAD_ARG (-1)     /* null pointer, bad enum, non-positive size, bad Qbits */
#define SLFP_ERR_SHAPE (-2)       /* inconsistent conv geometry (what F.conv2d would reject) */
#define SLFP_ERR_UNSUPPORTED (-3) /* valid but not implemented on this path (e.g. dilation with NCHW) */
#define SLFP_ERR_ALIGNMENT (-4)   /* a pointer is not aligned as the kernel needs (16 B) */
#define SLFP_ERR_HIP (-5)         /* a HIP runtime call failed (launch error etc.) */

/* codec formats (the `fmt` argument) */
#define SLFP_FMT_ACT8 0 /* quantize_act(8):    SLFP<3,4> activations, utils/sfp_quant.py:80-96 */
#define SLFP_FMT_W8 1   /* quantize_weight(8): SLFP<3,4> weights,     utils/sfp_quant.py:32-47 */
#define SLFP_FMT_SFP7 2 /* quantize_{act,weight}(7): SFP<3,3>,        utils/sfp_quant.py:14-30,63-78 */
#define SLFP_FMT_EXT 4  /* OR-able: extended code points (exact zero = 0x01, clamp literal =
                           sign|0x02) so that decode(encode(x)) == quantize(x) bit for bit */

/* tensor layouts */
#define SLFP_LAYOUT_NCHW 0 /* the reference's layout (contiguous NCHW) */
#define SLFP_LAYOUT_NHWC 1 /* channels-last: the native layout of the HIP kernels */

/* pointwise/implicit-GEMM MFMA operand precision (slfp_conv2d_desc.mfma_passes) */
#define SLFP_MFMA_DEFAULT 0 /* library default = SLFP_MFMA_F16X1 (meets the 1e-3 parity bar) */
#define SLFP_MFMA_F16X1 1   /* one fp16 MFMA pass: ~2.5e-4 tensor-relative error on SLFP<3,4> */
#define SLFP_MFMA_F16X3 3   /* hi/lo split, three passes: float32-equivalent (~1e-6) */

int slfp_version(void);
/* Re-reads the SLFP_* experiment switches from the environment (profiles/variants.py); not for production use. */
void slfp_debug_reload_switches(void);
/* Text of the last error on the calling thread ("" if none). Never NULL. */
const char* slfp_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unavailable). */
int slfp_device_count(void);

/* ---- codec: replaces quantize_act(k) / quantize_weight(k) .forward ------------------
 * q = x[i] / scale_div is an IEEE float32 division, exactly as `input/self.Ka`
 * (utils/conv2d_func.py:21-22) with the float64 0-dim scale cast to float32.           */

/* code[i] = canonical code of Q_fmt(q): sign<<7 | (E+4)<<4 | m  (Qbits 8)
 *                                        sign<<6 | (E+4)<<3 | m  (Qbits 7)
 * (bit layout from the comments at utils/sfp_quant.py:95 and :125). */
int slfp_encode_f32(const float* x, uint8_t* code, size_t n, float scale_div, int fmt, void* stream);
/* y[i] = float32 value of code[i] (inverse of the above; with SLFP_FMT_EXT exact). */
int slfp_decode_f32(const uint8_t* code, float* y, size_t n, int fmt, void* stream);
/* y[i] = Q_fmt(q) as float32: bit-identical to what qfn.forward returns
 * (utils/sfp_quant.py:10-48, :59-97).  x == y (in place) is allowed. */
int slfp_quantize_f32(const float* x, float* y, size_t n, float scale_div, int fmt, void* stream);

/* y[i] = SFP<4,4> layer-output quantizer: replaces quantize_layerout(k <= 8).forward
 * (utils/sfp_quant.py:108-127), bit-identical including its quirks (the reference's `2^(-8)` is an
 * integer XOR, so only RNE to 5 significant bits, the >= 248 clamp and NaN-for-exact-zero are live). */
int slfp_quantize_layerout_f32(const float* x, float* y, size_t n, void* stream);
/* *out = max_i |x[i]| (device pointer to one float32): the per-layer statistic of the reference's
 * calibration pass get_scale_factor (cifar100_train_eval.py:261-271); Ka = max / 15.5. */
int slfp_absmax_f32(const float* x, size_t n, float* out, void* stream);

/* ---- conv2d: replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 and :41-47) ------ */

typedef struct slfp_conv2d_desc {
    int64_t n, c_in, h, w;    /* input  N x C_in x H x W (logical NCHW sizes)              */
    int64_t c_out, kh, kw;    /* weight C_out x (C_in/groups) x KH x KW                     */
    int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups;
    int32_t x_layout, y_layout; /* SLFP_LAYOUT_*: memory layout of x and of y                 */
    int32_t qbits;            /* 8 = SLFP<3,4>, 7 = SFP<3,3>  (32 is a host-side passthrough) */
    float ka, kw_scale;       /* float32(Ka), float32(Kw): the module's calibration scales    */
    int32_t mfma_passes;      /* SLFP_MFMA_*                                                   */
    int32_t reserved;
} slfp_conv2d_desc;

/* Validates the geometry; writes the output spatial size. */
int slfp_conv2d_out_shape(const slfp_conv2d_desc* d, int64_t* h_out, int64_t* w_out);
/* Which kernel family slfp_conv2d_fwd will run for this descriptor (for logs/tests):
 * a static string such as "dw3x3_nhwc", "pw_mfma_f16x3", "dense_mfma_f16x1", "direct_nhwc". */
const char* slfp_conv2d_kernel_name(const slfp_conv2d_desc* d);

/* Bytes of the prepared-weight blob for this layer (device memory the caller owns). */
size_t slfp_conv2d_wprep_bytes(const slfp_conv2d_desc* d);
/* Quantize the float32 OIHW weights once: weight_q = QW(w / Kw) (conv2d_func.py:22) and
 * lay them out for the kernel this descriptor selects.  weight_q_oihw (optional, may be
 * NULL) receives the reference's `self.weight_q` tensor (float32, OIHW).  The reference
 * re-quantizes the weights on every forward; callers cache the blob and call this again
 * whenever the weight tensor changes. */
int slfp_conv2d_prepare_weights(const slfp_conv2d_desc* d, const float* w_oihw, void* wprep,
                                float* weight_q_oihw, void* stream);
/* The same blob from the weights' 1-byte codes: codes_oihw[i] = slfp_encode_f32(w, Kw, fmt | SLFP_FMT_EXT)[i] with fmt =
 * SLFP_FMT_W8 (qbits 8) or SLFP_FMT_SFP7 (qbits 7).  decode(encode(x)) == quantize(x) bit for bit, so the blob is
 * identical to slfp_conv2d_prepare_weights' -- this is what a rank builds from the broadcast of SURVEY 8(e)
 * (quantized weights travel as u8 codes, 1 B per weight; each rank lays them out for its own kernels). */
int slfp_conv2d_prepare_weights_codes(const slfp_conv2d_desc* d, const uint8_t* codes_oihw, void* wprep,
                                      float* weight_q_oihw, void* stream);
/* Bytes of scratch slfp_conv2d_fwd needs for this descriptor: 0 for the HBM-bound NHWC families
 * (depthwise, pointwise, 3x3 stems); non-zero when a layout conversion is involved and for the
 * two MFMA families of compute-bound layers ("dense_mfma_*": k x k with C_in >= 16,
 * "stem_mfma_*": C_in <= 4 with KH*KW >= 25), which encode the input once to fp16 into it.
 * The contents are scratch: nothing persists between calls. */
size_t slfp_conv2d_workspace_bytes(const slfp_conv2d_desc* d);
/* y = conv2d(QA(x/Ka), weight_q, bias/Ka/Kw) * Ka * Kw.
 * bias: NULL for conv2d_Q (conv2d_func.py:20-25), float32[C_out] for conv2d_Q_bias (:41-47).
 * input_q (optional, may be NULL): receives QA(x/Ka) in x's layout (the reference's
 * `self.input_q`).  workspace: slfp_conv2d_workspace_bytes(d) bytes or NULL if that is 0. */
int slfp_conv2d_fwd(const slfp_conv2d_desc* d, const float* x, const void* wprep, const float* bias,
                    float* y, float* input_q, void* workspace, void* stream);

/* Same, with the eval-mode BatchNorm2d (+ ReLU) that follows every Conv2d_Q in the reference
 * nets (nets_imgnet/mobilenetv1.py:24-41, resnet50.py:74-88) fused into the epilogue (SURVEY 8f
 * rank 1):   y = act(conv_q_out * post_scale[c] + post_shift[c]),  act = max(., 0) if relu.
 * post_scale = gamma / sqrt(running_var + eps), post_shift = beta - running_mean * post_scale,
 * float32[C_out], 16-byte aligned; both NULL = no affine.  The conv result itself keeps the
 * reference's (out * Ka) * Kw roundings; the affine is one fused multiply-add on top. */
#define SLFP_POST_RELU 1      /* `relu` argument of slfp_conv2d_fwd_post: max(., 0) last                         */
#define SLFP_POST_LAYEROUT 2  /* SFP<4,4> layer-output quantizer (slfp_quantize_layerout_f32) between the affine
                                 and the ReLU: the [Conv2d_Q, BatchNorm2d, layerout_quantize_func, ReLU] blocks of
                                 MobileNetV1_swish / VGG16_gelu / ShuffleNetV2 (nets_cifar/mobilenetv1.py:196-231);
                                 needs post_scale / post_shift                                                  */
int slfp_conv2d_fwd_post(const slfp_conv2d_desc* d, const float* x, const void* wprep, const float* bias,
                         const float* post_scale, const float* post_shift, int relu, float* y,
                         float* input_q, void* workspace, void* stream);

/* ---- one MobileNet block in one launch (second half) ---------------------------------------------
 * nets_imgnet/mobilenetv1.py:24-41's conv_dw block -- Conv2d_Q(3x3, groups = C) -> BatchNorm2d -> ReLU -> Conv2d_Q(1x1)
 * -> BatchNorm2d -> ReLU -- with the depthwise result quantized for the pointwise layer where it is produced and handed
 * to the matrix cores through LDS as fp16: the intermediate tensor never goes to HBM.  Bit-identical to
 * slfp_conv2d_fwd_post(dw) followed by slfp_conv2d_fwd_post(pw) in the single-pass MFMA mode.  `dw` / `pw` are the two
 * layers' descriptors (NHWC; pw's input size = dw's output size), wprep_* their prepared weights, post1_* the folded
 * BatchNorm of the depthwise layer (required), post2_* that of the pointwise layer (or NULL), relu*: 0 / 1.
 * slfp_dwpw_supported: 1 if the pair can be fused (3x3 depthwise stride 1 / 2, C in {32, 64, 128}, N in {64, 128, 256},
 * C * N * 2 B <= 64 KiB, single-pass mode, threshold tables available for both Ka), else 0. */
int slfp_dwpw_supported(const slfp_conv2d_desc* dw, const slfp_conv2d_desc* pw);
int slfp_dwpw_fwd(const slfp_conv2d_desc* dw, const slfp_conv2d_desc* pw, const float* x, const void* wprep_dw,
                  const float* post1_scale, const float* post1_shift, int relu1, const void* wprep_pw,
                  const float* bias_pw, const float* post2_scale, const float* post2_shift, int relu2, float* y,
                  void* stream);

/* ---- inter-layer activations as 1-byte codes ("... + next layer's encode") ---------------------------
 * Every Conv2d_Q of the reference nets feeds BatchNorm2d -> ReLU -> the next Conv2d_Q, whose first step is
 * input_q = quantize_act(input / Ka) (utils/conv2d_func.py:21; nets_imgnet/mobilenetv1.py:24-33).  With `y_codes` the
 * producer applies the CONSUMER's quantizer in its epilogue and stores, per element, the byte
 *     slfp_encode_f32(out, y_ka, fmt(y_qbits) | SLFP_FMT_EXT)
 * instead of the float32 value; with `x_codes` a layer reads such bytes (written for ITS Ka and qbits) in place of the
 * float32 tensor and skips its own quantizer.  Classes and values are the same as on the float32 interface, so a chain
 * of layers linked this way returns bit for bit what slfp_conv2d_fwd_post returns layer by layer (single-pass MFMA mode /
 * SFP<3,3>), while activations cross HBM as 1 B per element instead of 4.  NaN has no code (it becomes 0x00, as in
 * slfp_encode_f32).  Both tensors are NHWC; x, y and wprep 16-byte aligned; wprep is the blob of
 * slfp_conv2d_prepare_weights for the same descriptor.  Supported: 3x3 depthwise (x_codes), 1x1 (x_codes; C_in a
 * multiple of 32, C_out of 16 with y_codes), and the 3x3 stride-2 RGB stem -> 32 channels (float32 in, y_codes):
 * every layer of nets_imgnet/mobilenetv1.py:43-57; the 3x3 RGB stem -> 64 channels of VGG-16 (float32 in, y_codes); dense k x k
 * layers through slfp_conv2d_fwd_codes_ws (below).
 * slfp_conv2d_codes_supported answers 1 / 0 without device work. */
typedef struct slfp_conv2d_io {
    int32_t x_codes;  /* 0: x is float32 (as slfp_conv2d_fwd); 1: x is uint8 codes of QA(. / d->ka), format of d->qbits */
    int32_t y_codes;  /* 0: y is float32; 1: y receives uint8 codes of QA(out / y_ka) in the format of y_qbits        */
    float y_ka;       /* float32(Ka) of the layer that will read y                                                   */
    int32_t y_qbits;  /* its q_bit: 8 = SLFP<3,4>, 7 = SFP<3,3>                                                       */
} slfp_conv2d_io;
int slfp_conv2d_codes_supported(const slfp_conv2d_desc* d, const slfp_conv2d_io* io, int has_bias, int relu);
/* post_scale / post_shift / relu as in slfp_conv2d_fwd_post (SLFP_POST_LAYEROUT is not supported here). */
int slfp_conv2d_fwd_codes(const slfp_conv2d_desc* d, const slfp_conv2d_io* io, const void* x, const void* wprep,
                          const float* bias, const float* post_scale, const float* post_shift, int relu, void* y,
                          void* stream);
/* The same with a workspace, for the layers whose float32 form needs one: dense k x k convolutions on the matrix cores
 * (VGG-16 / ResNet-50 3x3, SqueezeNet expand3x3; x_codes and / or y_codes; C_out a multiple of 16 with y_codes).  `workspace`:
 * slfp_conv2d_workspace_bytes(d) bytes, 16-byte aligned; may be NULL for the layers slfp_conv2d_fwd_codes takes.  The codes
 * are decoded into the same fp16 operand copy the float32 interface builds from its input: results are bit-identical. */
int slfp_conv2d_fwd_codes_ws(const slfp_conv2d_desc* d, const slfp_conv2d_io* io, const void* x, const void* wprep,
                             const float* bias, const float* post_scale, const float* post_shift, int relu, void* y,
                             void* workspace, void* stream);
/* nn.MaxPool2d (floor mode, dilation 1) on a tensor of activation codes (NHWC, C a multiple of 4; 16-byte aligned): the class
 * of a window's largest input is the highest class among its codes in the order of the classes' pre-images (the clamp literal
 * of Qbits 8 ranks above the top regular code, "tiny" above exact zero, signed codes by decreasing magnitude), so
 * y == slfp_encode_f32(max_pool2d(t), ka, fmt | SLFP_FMT_EXT) wherever x == slfp_encode_f32(t, ka, fmt | SLFP_FMT_EXT), bit for
 * bit -- the pools between VGG-16's stages (nets_cifar/vgg16.py:39,49,63,78,92) can stay inside a code chain. */
int slfp_maxpool2d_codes(const uint8_t* x, uint8_t* y, int64_t n, int64_t h, int64_t w, int64_t c, int kh, int kw, int sh, int sw,
                         int ph, int pw, int qbits, void* stream);
/* Self-check of the producer side: sweeps ALL 2^32 float32 inputs on the device; out3[0] = inputs whose table-driven code
 * (signed variant) differs from slfp_encode_f32(.., fmt | SLFP_FMT_EXT), out3[1] = the same for the unsigned variant a
 * producer with a ReLU epilogue runs (inputs >= +0 and -0), out3[2] = code bytes whose decode-table entries (float32 and
 * fp16 operand) differ from slfp_decode_f32.  All three must be 0.  fmt: SLFP_FMT_ACT8 or SLFP_FMT_SFP7. */
int slfp_debug_code_mismatches(float scale_div, int fmt, unsigned long long* out3, void* stream);

/* ---- linear: replaces Linear_Q.forward (utils/conv2d_func.py:60-65) -------------------
 * out = linear(QA(x/Ka), QW(w/Kw), bias/Kw/Ka) * Kw * Ka   (note the Kw-first order).
 * x: [batch, in_f] row-major, w: [out_f, in_f] row-major, bias: [out_f] or NULL.  The weights
 * are re-quantized on every call (as the reference does) into `workspace`
 * (slfp_linear_workspace_bytes bytes of device memory).                                   */
size_t slfp_linear_workspace_bytes(int64_t batch, int64_t in_f, int64_t out_f);
int slfp_linear_fwd(const float* x, const float* w, const float* bias, float* y, int64_t batch,
                    int64_t in_f, int64_t out_f, float ka, float kw_scale, int qbits, int mfma_passes,
                    void* workspace, void* stream);
/* The same with the weights quantized ONCE (inference: AlexNet's 9216x4096 and VGG-16's 25088x4096
 * fully-connected layers would otherwise be re-encoded on every forward):
 * slfp_linear_prepare_weights fills `wprep` (slfp_linear_workspace_bytes(1, in_f, out_f) bytes; the
 * blob depends on kw_scale, qbits and mfma_passes, not on the batch), slfp_linear_fwd_prepared
 * consumes it.  slfp_linear_fwd == prepare + fwd_prepared on the caller's workspace. */
int slfp_linear_prepare_weights(const float* w, void* wprep, int64_t in_f, int64_t out_f, float kw_scale,
                                int qbits, int mfma_passes, void* stream);
int slfp_linear_fwd_prepared(const float* x, const void* wprep, const float* bias, float* y, int64_t batch,
                             int64_t in_f, int64_t out_f, float ka, float kw_scale, int qbits,
                             int mfma_passes, void* stream);

/* ---- self-check ------------------------------------------------------------------------
 * The kernels compute x / scale_div with an FMA correction chain on a host-computed
 * reciprocal instead of the IEEE division macro (csrc/slfp_device.hpp, ScaleDiv).  This
 * sweeps ALL 2^32 float32 inputs on the device and writes to out2 (device memory, 2 x u64):
 * out2[0] = inputs with |x| and |x/scale| in [1e-20, 1e20] whose quotient differs from IEEE `/`,
 * out2[1] = inputs (all 2^32) for which any quantizer result would differ.  Both must be 0. */
int slfp_debug_div_mismatches(float scale_div, unsigned long long* out2, void* stream);

/* The conv kernels evaluate QA(x / Ka) through a per-Ka threshold table (csrc/slfp_enc.hpp: one approximate
 * multiply picks a bin, one exact compare in x-space picks the side) instead of the long form above.  The table is
 * built and proven on the host per (Ka, format); this sweeps ALL 2^32 float32 inputs on the device and writes
 * out2[0] = inputs whose float32 result differs from the long form, out2[1] = inputs whose fp16 MFMA-operand
 * result differs (+0 / -0 counted as equal).  Both must be 0.  fmt: SLFP_FMT_ACT8 or SLFP_FMT_SFP7.
 * Returns SLFP_ERR_UNSUPPORTED if no table can be proven for this scale (the kernels then use the long form). */
int slfp_debug_enc_mismatches(float scale_div, int fmt, unsigned long long* out2, void* stream);
/* The same for the hi / lo fp16 operand pair of the three-pass (float32-equivalent) pointwise mode (round 3): *out1 = inputs
 * (all 2^32) whose pair from the two tables differs from hi = fp16(16 Q), lo = fp16(16 Q - fp32(hi)).  Must be 0. */
int slfp_debug_enc_hl_mismatches(float scale_div, int fmt, unsigned long long* out1, void* stream);
/* 1 if the threshold table exists for this scale / format (host-only query, no device work). */
int slfp_enc_table_ok(float scale_div, int fmt);

/* ---- layout helpers (the reference is NCHW; the kernels are NHWC) -------------------- */
int slfp_nchw_to_nhwc_f32(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, void* stream);
int slfp_nhwc_to_nchw_f32(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLFP_H_ */
