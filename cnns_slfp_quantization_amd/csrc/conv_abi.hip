// conv_abi.hip -- C ABI for the quantized conv2d / linear forward (include/slfp.h):
// descriptor validation, kernel selection, weight preparation, dispatch.
#include <cstring>
#include "slfp_device.hpp"
#include "slfp_codes.hpp"
#include "slfp_host.hpp"

namespace slfp {

static size_t round256(size_t b) { return (b + 255) & ~(size_t)255; }

static int make_plan_core(const slfp_conv2d_desc* d, ConvPlan* plan) {
    if (!d || !plan) return fail(SLFP_ERR_BAD_ARG, "conv2d: null descriptor");
    if (d->n <= 0 || d->c_in <= 0 || d->h <= 0 || d->w <= 0 || d->c_out <= 0 || d->kh <= 0 || d->kw <= 0)
        return fail(SLFP_ERR_SHAPE, "conv2d: non-positive size");
    if (d->stride_h <= 0 || d->stride_w <= 0 || d->dil_h <= 0 || d->dil_w <= 0 || d->pad_h < 0 || d->pad_w < 0)
        return fail(SLFP_ERR_SHAPE, "conv2d: bad stride/dilation/padding");
    if (d->groups <= 0 || d->c_in % d->groups || d->c_out % d->groups)
        return fail(SLFP_ERR_SHAPE, "conv2d: groups=%d does not divide C_in=%lld / C_out=%lld", d->groups,
                    (long long)d->c_in, (long long)d->c_out);
    if (d->qbits != 8 && d->qbits != 7)
        return fail(SLFP_ERR_BAD_ARG, "conv2d: qbits must be 8 (SLFP<3,4>) or 7 (SFP<3,3>), got %d", d->qbits);
    if ((d->x_layout != SLFP_LAYOUT_NCHW && d->x_layout != SLFP_LAYOUT_NHWC) ||
        (d->y_layout != SLFP_LAYOUT_NCHW && d->y_layout != SLFP_LAYOUT_NHWC))
        return fail(SLFP_ERR_BAD_ARG, "conv2d: unknown layout");
    if (!(d->ka > 0.f) || !(d->kw_scale > 0.f)) return fail(SLFP_ERR_BAD_ARG, "conv2d: Ka and Kw must be > 0");
    if (!scale_div_ok(d->ka) || !scale_div_ok(d->kw_scale))
        return fail(SLFP_ERR_UNSUPPORTED, "conv2d: Ka and Kw must be within [1e-30, 1e30]");
    if (d->mfma_passes != SLFP_MFMA_DEFAULT && d->mfma_passes != SLFP_MFMA_F16X1 && d->mfma_passes != SLFP_MFMA_F16X3)
        return fail(SLFP_ERR_BAD_ARG, "conv2d: mfma_passes must be 0, 1 or 3");
    const int64_t eh = d->h + 2 * (int64_t)d->pad_h - (int64_t)d->dil_h * (d->kh - 1) - 1;
    const int64_t ew = d->w + 2 * (int64_t)d->pad_w - (int64_t)d->dil_w * (d->kw - 1) - 1;
    if (eh < 0 || ew < 0) return fail(SLFP_ERR_SHAPE, "conv2d: kernel larger than the padded input");
    plan->h_out = eh / d->stride_h + 1;
    plan->w_out = ew / d->stride_w + 1;
    if (d->n > 0x7FFFFFFF || d->h > 0x7FFFFFFF || d->w > 0x7FFFFFFF || d->c_in > 0x7FFFFFFF || d->c_out > 0x7FFFFFFF)
        return fail(SLFP_ERR_UNSUPPORTED, "conv2d: dimension exceeds 2^31");
    plan->fmt_act = d->qbits == 8 ? kFmtAct8 : kFmtSfp7;
    plan->fmt_w = d->qbits == 8 ? kFmtW8 : kFmtSfp7;
    plan->passes = d->qbits == 7 ? 1 : (d->mfma_passes == SLFP_MFMA_F16X3 ? 3 : 1);
    plan->k_pad = plan->n_pad = 0;
    plan->repad = false;
    plan->cpi = d->c_in; plan->cpo = d->c_out;
    plan->s1 = d->ka;
    plan->s2 = d->kw_scale;
    const int64_t cg = d->c_in / d->groups;
    const bool sq_stride = d->stride_h == d->stride_w;
    if (d->groups == d->c_in && d->c_out == d->c_in && d->kh == 3 && d->kw == 3 && d->dil_h == 1 && d->dil_w == 1 &&
        sq_stride && (d->stride_h == 1 || d->stride_h == 2) && d->pad_h == d->pad_w && d->pad_h <= 2 &&
        (d->c_in % 2) == 0) {   // multiples of 4: 16-byte lanes; other even counts (58): 8-byte lanes
        plan->family = kDw3x3;
        plan->wprep_bytes = round256((size_t)9 * d->c_in * sizeof(float));
    } else if (d->kh == 1 && d->kw == 1 && d->groups == 1 && d->pad_h == 0 && d->pad_w == 0 && sq_stride &&
               (((d->c_in % 4) == 0 && (d->c_out % 4) == 0) ||
                // even channel counts (ShuffleNetV2: 58): the LDS-resident stream kernel with 8-byte accesses
                ((d->c_in % 2) == 0 && (d->c_out % 2) == 0 && d->c_in >= 8 &&
                 pointwise_stream_fits(ceil_div(d->c_in, 64) * 64, ceil_div(d->c_out, 64) * 64, plan->passes)))) {
        plan->family = kPointwise;
        plan->k_pad = ceil_div(d->c_in, 64) * 64;
        plan->n_pad = ceil_div(d->c_out, 64) * 64;
        plan->wprep_bytes = round256((size_t)2 * plan->k_pad * plan->n_pad * sizeof(_Float16));
    } else if (stem_small_applicable(*d, plan->passes)) {
        plan->family = kStemSmall;
        plan->n_pad = stem_small_tiles(*d);  // 16-channel tiles
        plan->wprep_bytes = round256((size_t)plan->n_pad * 1024);
    } else if (stem_mfma_applicable(*d, plan->passes)) {
        plan->family = kStemMfma;
        int ksub, nt;
        stem_mfma_blob_shape(*d, &ksub, &nt);
        plan->k_pad = ksub;  // k-steps per tap row
        plan->n_pad = nt;    // 16-channel tiles
        plan->wprep_bytes = round256((size_t)d->kh * ksub * nt * 1024);
    } else if (dense_mfma_applicable(*d, plan->passes)) {
        plan->family = kDenseMfma;
        plan->k_pad = ceil_div(d->c_in, 64) * 64;
        plan->n_pad = ceil_div(d->c_out, 16) * 16;
        // one fp16 plane, plus the residual plane in float32-equivalent mode
        plan->wprep_bytes = round256((size_t)(plan->passes == 3 ? 2 : 1) * d->kh * d->kw * plan->k_pad * plan->n_pad * sizeof(_Float16));
    } else {
        plan->family = kDirect;
        plan->wprep_bytes = round256((size_t)d->kh * d->kw * cg * d->c_out * sizeof(float));
    }
    return SLFP_OK;
}

static int64_t round4(int64_t c) { return (c + 3) & ~(int64_t)3; }

// The descriptor of the channel-padded launch, if this layer qualifies (see ConvPlan::repad).
static bool padded_desc(const slfp_conv2d_desc& d, slfp_conv2d_desc* d2) {
    if ((d.c_in % 4) == 0 && (d.c_out % 4) == 0) return false;
    *d2 = d;
    d2->x_layout = d2->y_layout = SLFP_LAYOUT_NHWC;
    if (d.groups == d.c_in && d.c_out == d.c_in) {          // depthwise
        d2->c_in = d2->c_out = round4(d.c_in);
        d2->groups = (int32_t)d2->c_in;
    } else if (d.groups == 1 && d.kh == 1 && d.kw == 1 && d.c_in >= 8) {  // pointwise
        d2->c_in = round4(d.c_in);
        d2->c_out = round4(d.c_out);
    } else {
        return false;
    }
    return true;
}

int make_plan(const slfp_conv2d_desc* d, ConvPlan* plan) {
    int rc = make_plan_core(d, plan);
    if (rc != SLFP_OK || plan->family != kDirect) return rc;
    slfp_conv2d_desc d2;
    if (!padded_desc(*d, &d2)) return rc;
    ConvPlan inner;
    if (make_plan_core(&d2, &inner) != SLFP_OK || (inner.family != kDw3x3 && inner.family != kPointwise)) return rc;
    const int64_t ho = plan->h_out, wo = plan->w_out;
    *plan = inner;
    plan->h_out = ho; plan->w_out = wo;
    plan->repad = true;
    plan->cpi = d2.c_in; plan->cpo = d2.c_out;
    return SLFP_OK;
}

// dst[row][c] = c < cs ? src[row][c] : 0 for c < cd  (channel re-padding of an NHWC tensor, either way).
// V = floats per thread: 2 when both widths are even (58 <-> 60: 8-byte accesses), else 1.
template <int V>
__global__ __launch_bounds__(256) void k_repad(const float* __restrict__ src, float* __restrict__ dst, int64_t total_v,
                                               int cs, int cd) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total_v) return;
    const int cdv = cd / V;
    const int64_t row = idx / cdv;
    const int c = (int)(idx - row * cdv) * V;
    if (V == 2) {
        float2 v = make_float2(0.f, 0.f);
        if (c < cs) v = *reinterpret_cast<const float2*>(src + row * cs + c);
        *reinterpret_cast<float2*>(dst + row * cd + c) = v;
    } else {
        dst[idx] = c < cs ? src[row * cs + c] : 0.f;
    }
}

static int launch_repad(const float* src, float* dst, int64_t rows, int64_t cs, int64_t cd, hipStream_t stream) {
    if ((cs % 2) == 0 && (cd % 2) == 0) {
        const int64_t total = rows * (cd / 2);
        hipLaunchKernelGGL(k_repad<2>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream, src, dst, total, (int)cs, (int)cd);
    } else {
        const int64_t total = rows * cd;
        hipLaunchKernelGGL(k_repad<1>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream, src, dst, total, (int)cs, (int)cd);
    }
    return check_launch("slfp channel re-pad kernel");
}

// One thread per OIHW weight element: weight_q = QW(w / Kw) (utils/conv2d_func.py:22), written
// in the layout the selected kernel family reads.
// CODES: `w` is not float32 weights but their 1-byte extended codes (slfp_encode_f32 with SLFP_FMT_EXT): the value is
// decoded instead of quantized (decode(encode(x)) == quantize(x) bit for bit), so a blob built from broadcast codes
// is identical to one built from the weights themselves (multi-GPU: sharding.py).
template <int FMT, bool CODES = false>
__global__ __launch_bounds__(256) void k_prepare(const float* __restrict__ w, void* __restrict__ prep,
                                                 float* __restrict__ wq_oihw, int64_t total, int O, int Cg, int KH,
                                                 int KW, const ScaleDiv sd, int family, int KS, int64_t plane, int ldo) {
    __shared__ uint32_t sT[16];
    lut_fill<FMT>(sT);   // W8 / SFP7: the identity table the decoder indexes as well
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float q;
    if constexpr (CODES) q = __uint_as_float(decode_bits<FMT>(reinterpret_cast<const uint8_t*>(w)[idx], true, sT));
    else q = quantize_scaled<FMT>(w[idx], sd, sT);
    if (wq_oihw) wq_oihw[idx] = q;
    int64_t r = idx;
    const int kw = (int)(r % KW); r /= KW;
    const int kh = (int)(r % KH); r /= KH;
    const int ci = (int)(r % Cg); r /= Cg;
    const int o = (int)r;
    if (family == kDw3x3) {
        reinterpret_cast<float*>(prep)[(size_t)(kh * 3 + kw) * ldo + o] = q;  // [9][C (padded)]
    } else if (family == kDirect) {
        reinterpret_cast<float*>(prep)[((size_t)(kh * KW + kw) * Cg + ci) * O + o] = q;  // [KH][KW][Cg][O]
    } else if (family == kStemSmall) {
        // one 32-deep k-step: k = (kh*KW + kw)*C_in + ci; A-fragment lane (k/8, o%16), element k%8: conv_stem_small.hip
        const int k = (kh * KW + kw) * Cg + ci;
        const size_t at = (((size_t)(o >> 4) * 64) + (size_t)((k >> 3) * 16 + (o & 15))) * 8 + (k & 7);
        reinterpret_cast<_Float16*>(prep)[at] = (_Float16)(16.0f * q);
    } else if (family == kStemMfma) {
        // K = (kh, run element r = kw*C_in + ci padded to 32*KS): k-step kh*KS + r/32, lane-quarter (r%32)/8;
        // `plane` carries the number of channel tiles: conv_stem_mfma.hip
        const int r = kw * Cg + ci, ks = kh * KS + (r >> 5);
        const size_t at = ((((size_t)ks * plane + (o >> 4)) * 64) + (size_t)(((r & 31) >> 3) * 16 + (o & 15))) * 8 + (r & 7);
        reinterpret_cast<_Float16*>(prep)[at] = (_Float16)(16.0f * q);
    } else if (family == kDenseMfma) {
        // tap-major copy of the pointwise fragment order (single fp16 plane): conv_dense.hip
        const int nt = o >> 4, row = o & 15, ks = ci >> 5, kk = ci & 31;
        const int kq = (kk & 15) >> 2, j = (kk >> 4) * 4 + (kk & 3);
        const size_t ntiles = (size_t)(plane / ((int64_t)KS * 32 * 16));
        const size_t at = ((((size_t)(kh * KW + kw) * ntiles + nt) * KS + ks) * 64 + (size_t)(kq * 16 + row)) * 8 + j;
        const float v = 16.0f * q;
        const _Float16 hi = (_Float16)v;
        reinterpret_cast<_Float16*>(prep)[at] = hi;
        if (ldo) reinterpret_cast<_Float16*>(prep)[(size_t)KH * KW * plane + at] = (_Float16)(v - (float)hi);  // ldo != 0: residual plane
    } else {
        // MFMA 16x16x32 A-fragment order: tile (o/16, k/32), lane = kq*16 + o%16 where lane-quarter kq
        // holds k%32 in {kq*4..kq*4+3} (elements 0-3) and {16+kq*4..16+kq*4+3} (elements 4-7): conv_pw.hip
        const int nt = o >> 4, row = o & 15, ks = ci >> 5, kk = ci & 31;
        const int kq = (kk & 15) >> 2, j = (kk >> 4) * 4 + (kk & 3);
        const size_t at = (((size_t)nt * KS + ks) * 64 + (size_t)(kq * 16 + row)) * 8 + j;
        const float v = 16.0f * q;  // 2^4 pre-scale (exact), see conv_pw.hip
        const _Float16 hi = (_Float16)v;
        _Float16* blob = reinterpret_cast<_Float16*>(prep);
        blob[at] = hi;
        blob[plane + at] = (_Float16)(v - (float)hi);
    }
}

int launch_prepare_weights(const slfp_conv2d_desc& d, const ConvPlan& p, const float* w_oihw, void* wprep,
                           float* weight_q_oihw, hipStream_t stream, bool codes) {
    const int Cg = (int)(d.c_in / d.groups);
    const int64_t total = d.c_out * Cg * d.kh * d.kw;
    if (p.family == kPointwise || p.family == kDenseMfma || p.family == kStemMfma || p.family == kStemSmall || p.repad) {
        if (hipMemsetAsync(wprep, 0, p.wprep_bytes, stream) != hipSuccess) return check_launch("hipMemsetAsync(wprep)");
    }
    const int64_t plane = p.family == kStemMfma ? p.n_pad : p.k_pad * p.n_pad;
    const int KS = p.family == kStemMfma ? (int)p.k_pad : (int)(p.k_pad / 32);
    const unsigned grid = (unsigned)ceil_div(total, 256);
    const ScaleDiv sd = make_scale_div(d.kw_scale);
    // depthwise: row pitch of the [9][C] table (padded channel count); dense MFMA: 1 = also write the residual plane
    const int ldo = p.family == kDenseMfma ? (p.passes == 3 ? 1 : 0) : (int)p.cpo;
#define SLFP_PREP(FF, CC) hipLaunchKernelGGL((k_prepare<FF, CC>), dim3(grid), dim3(256), 0, stream, w_oihw, wprep, weight_q_oihw, total, \
                                             (int)d.c_out, Cg, (int)d.kh, (int)d.kw, sd, (int)p.family, KS, plane, ldo)
    if (p.fmt_w == kFmtW8) { if (codes) SLFP_PREP(kFmtW8, true); else SLFP_PREP(kFmtW8, false); }
    else { if (codes) SLFP_PREP(kFmtSfp7, true); else SLFP_PREP(kFmtSfp7, false); }
#undef SLFP_PREP
    return check_launch("slfp weight prepare kernel");
}

static const char* family_name(const ConvPlan& p, const slfp_conv2d_desc& d) {
    if (p.family == kDirect && stem_applicable(d)) return "stem_nhwc";
    if (p.repad) {  // the same kernels on channel-padded copies (ConvPlan::repad)
        if (p.family == kDw3x3) return "repad+dw3x3_nhwc";
        return p.fmt_act == kFmtSfp7 ? "repad+pw_mfma_f16_exact" : (p.passes == 3 ? "repad+pw_mfma_f16x3" : "repad+pw_mfma_f16x1");
    }
    switch (p.family) {
        case kDw3x3: return "dw3x3_nhwc";
        case kPointwise: return p.fmt_act == kFmtSfp7 ? "pw_mfma_f16_exact" : (p.passes == 3 ? "pw_mfma_f16x3" : "pw_mfma_f16x1");
        case kDenseMfma: return p.fmt_act == kFmtSfp7 ? "dense_mfma_f16_exact" : (p.passes == 3 ? "dense_mfma_f16x3" : "dense_mfma_f16x1");
        case kStemMfma: return p.fmt_act == kFmtSfp7 ? "stem_mfma_f16_exact" : "stem_mfma_f16x1";
        case kStemSmall: return p.fmt_act == kFmtSfp7 ? "stem_small_mfma_f16_exact" : "stem_small_mfma_f16x1";
        default: return "direct_nhwc";
    }
}

}  // namespace slfp

using namespace slfp;

extern "C" {

int slfp_conv2d_out_shape(const slfp_conv2d_desc* d, int64_t* h_out, int64_t* w_out) {
    ConvPlan p;
    const int rc = make_plan(d, &p);
    if (rc != SLFP_OK) return rc;
    if (h_out) *h_out = p.h_out;
    if (w_out) *w_out = p.w_out;
    return SLFP_OK;
}

const char* slfp_conv2d_kernel_name(const slfp_conv2d_desc* d) {
    ConvPlan p;
    if (make_plan(d, &p) != SLFP_OK) return "invalid";
    return family_name(p, *d);
}

size_t slfp_conv2d_wprep_bytes(const slfp_conv2d_desc* d) {
    ConvPlan p;
    if (make_plan(d, &p) != SLFP_OK) return 0;
    return p.wprep_bytes;
}

int slfp_conv2d_prepare_weights(const slfp_conv2d_desc* d, const float* w_oihw, void* wprep, float* weight_q_oihw,
                                void* stream) {
    ConvPlan p;
    const int rc = make_plan(d, &p);
    if (rc != SLFP_OK) return rc;
    if (!w_oihw || !wprep) return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_prepare_weights: null pointer");
    if (!aligned16(wprep)) return fail(SLFP_ERR_ALIGNMENT, "slfp_conv2d_prepare_weights: wprep must be 16-byte aligned");
    return launch_prepare_weights(*d, p, w_oihw, wprep, weight_q_oihw, as_stream(stream));
}

int slfp_conv2d_prepare_weights_codes(const slfp_conv2d_desc* d, const uint8_t* codes_oihw, void* wprep, float* weight_q_oihw,
                                      void* stream) {
    ConvPlan p;
    const int rc = make_plan(d, &p);
    if (rc != SLFP_OK) return rc;
    if (!codes_oihw || !wprep) return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_prepare_weights_codes: null pointer");
    if (!aligned16(wprep)) return fail(SLFP_ERR_ALIGNMENT, "slfp_conv2d_prepare_weights_codes: wprep must be 16-byte aligned");
    return launch_prepare_weights(*d, p, reinterpret_cast<const float*>(codes_oihw), wprep, weight_q_oihw, as_stream(stream), true);
}

static size_t workspace_bytes_for(const slfp_conv2d_desc* d, const ConvPlan& p) {
    size_t b = 0;
    if (d->x_layout == SLFP_LAYOUT_NCHW) b += round256((size_t)d->n * d->c_in * d->h * d->w * sizeof(float));
    if (d->y_layout == SLFP_LAYOUT_NCHW) b += round256((size_t)d->n * d->c_out * p.h_out * p.w_out * sizeof(float));
    if (p.family == kDenseMfma) b += dense_mfma_workspace_bytes(*d, p.passes);  // the input encoded once to fp16
    if (p.family == kStemMfma) b += stem_mfma_workspace_bytes(*d, p.w_out);
    if (p.repad) {
        if (p.cpi != d->c_in) b += round256((size_t)d->n * d->h * d->w * p.cpi * sizeof(float));
        if (p.cpo != d->c_out) b += round256((size_t)d->n * p.h_out * p.w_out * p.cpo * sizeof(float));
        b += 3 * round256((size_t)p.cpo * sizeof(float));  // bias / post_scale / post_shift, padded
    }
    return b;
}

size_t slfp_conv2d_workspace_bytes(const slfp_conv2d_desc* d) {
    ConvPlan p;
    if (make_plan(d, &p) != SLFP_OK) return 0;
    return workspace_bytes_for(d, p);
}

int slfp_conv2d_fwd(const slfp_conv2d_desc* d, const float* x, const void* wprep, const float* bias, float* y,
                    float* input_q, void* workspace, void* stream) {
    return slfp_conv2d_fwd_post(d, x, wprep, bias, nullptr, nullptr, 0, y, input_q, workspace, stream);
}

int slfp_conv2d_fwd_post(const slfp_conv2d_desc* d, const float* x, const void* wprep, const float* bias,
                         const float* post_scale, const float* post_shift, int relu, float* y, float* input_q,
                         void* workspace, void* stream) {
    ConvPlan p;
    int rc = make_plan(d, &p);
    if (rc != SLFP_OK) return rc;
    if (!x || !wprep || !y) return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd: null pointer");
    if ((post_scale == nullptr) != (post_shift == nullptr))
        return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd_post: post_scale and post_shift must be given together");
    if (post_scale && (!aligned16(post_scale) || !aligned16(post_shift)))
        return fail(SLFP_ERR_ALIGNMENT, "slfp_conv2d_fwd_post: post_scale / post_shift must be 16-byte aligned");
    if ((relu & ~(SLFP_POST_RELU | SLFP_POST_LAYEROUT)) != 0)
        return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd_post: unknown flag bits in `relu`");
    if ((relu & SLFP_POST_LAYEROUT) && !post_scale)
        return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd_post: SLFP_POST_LAYEROUT needs post_scale / post_shift");
    const PostOp post{post_scale, post_shift, (relu & SLFP_POST_RELU) ? 1 : 0, (relu & SLFP_POST_LAYEROUT) ? 1 : 0};
    if (!aligned16(x) || !aligned16(y) || !aligned16(wprep) || (bias && !aligned16(bias)))
        return fail(SLFP_ERR_ALIGNMENT, "slfp_conv2d_fwd: x, y, wprep and bias must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (input_q) {  // the reference's self.input_q (utils/conv2d_func.py:21), in x's layout
        rc = launch_quantize(x, input_q, (size_t)d->n * d->c_in * d->h * d->w, d->ka, p.fmt_act, st);
        if (rc != SLFP_OK) return rc;
    }
    const size_t ws_need = workspace_bytes_for(d, p);   // the plan is built once per call
    if (ws_need && (!workspace || !aligned16(workspace)))
        return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd: %zu bytes of 16-byte aligned workspace required (slfp_conv2d_workspace_bytes)", ws_need);
    unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
    const float* x_nhwc = x;
    float* y_nhwc = y;
    if (d->x_layout == SLFP_LAYOUT_NCHW) {
        float* t = reinterpret_cast<float*>(ws);
        ws += round256((size_t)d->n * d->c_in * d->h * d->w * sizeof(float));
        rc = slfp_nchw_to_nhwc_f32(x, t, d->n, d->c_in, d->h, d->w, stream);
        if (rc != SLFP_OK) return rc;
        x_nhwc = t;
    }
    if (d->y_layout == SLFP_LAYOUT_NCHW) {
        y_nhwc = reinterpret_cast<float*>(ws);
        ws += round256((size_t)d->n * d->c_out * p.h_out * p.w_out * sizeof(float));
    }
    if (p.repad) {
        slfp_conv2d_desc d2;
        padded_desc(*d, &d2);
        const float* xin = x_nhwc;
        float* yout = y_nhwc;
        if (p.cpi != d->c_in) {
            float* xp = reinterpret_cast<float*>(ws);
            ws += round256((size_t)d->n * d->h * d->w * p.cpi * sizeof(float));
            rc = launch_repad(x_nhwc, xp, d->n * d->h * d->w, d->c_in, p.cpi, st);
            if (rc != SLFP_OK) return rc;
            xin = xp;
        }
        if (p.cpo != d->c_out) {
            yout = reinterpret_cast<float*>(ws);
            ws += round256((size_t)d->n * p.h_out * p.w_out * p.cpo * sizeof(float));
        }
        const float* vec[3] = {bias, post_scale, post_shift};  // per-channel vectors are read 16 bytes at a time
        for (int i = 0; i < 3; ++i) {
            float* vp = reinterpret_cast<float*>(ws);
            ws += round256((size_t)p.cpo * sizeof(float));
            if (!vec[i] || p.cpo == d->c_out) continue;
            rc = launch_repad(vec[i], vp, 1, d->c_out, p.cpo, st);
            if (rc != SLFP_OK) return rc;
            vec[i] = vp;
        }
        const PostOp post2{vec[1], vec[2], post.relu, post.layerout};
        if (p.family == kDw3x3) rc = launch_dw3x3(d2, p, xin, reinterpret_cast<const float*>(wprep), vec[0], post2, yout, st);
        else rc = launch_pointwise(d2, p, xin, wprep, vec[0], post2, yout, st);
        if (rc != SLFP_OK) return rc;
        if (p.cpo != d->c_out) rc = launch_repad(yout, y_nhwc, d->n * p.h_out * p.w_out, p.cpo, d->c_out, st);
        if (rc != SLFP_OK) return rc;
        if (d->y_layout == SLFP_LAYOUT_NCHW) rc = slfp_nhwc_to_nchw_f32(y_nhwc, y, d->n, d->c_out, p.h_out, p.w_out, stream);
        return rc;
    }
    switch (p.family) {
        case kDw3x3: rc = launch_dw3x3(*d, p, x_nhwc, reinterpret_cast<const float*>(wprep), bias, post, y_nhwc, st); break;
        case kPointwise: rc = launch_pointwise(*d, p, x_nhwc, wprep, bias, post, y_nhwc, st); break;
        case kDenseMfma: rc = launch_dense_mfma(*d, p, x_nhwc, wprep, bias, post, y_nhwc, ws, st); break;
        case kStemMfma: rc = launch_stem_mfma(*d, p, x_nhwc, wprep, bias, post, y_nhwc, ws, st); break;
        case kStemSmall: rc = launch_stem_small(*d, p, x_nhwc, wprep, bias, post, y_nhwc, st); break;
        default: rc = launch_direct(*d, p, x_nhwc, reinterpret_cast<const float*>(wprep), bias, post, y_nhwc, st); break;
    }
    if (rc != SLFP_OK) return rc;
    if (d->y_layout == SLFP_LAYOUT_NCHW) rc = slfp_nhwc_to_nchw_f32(y_nhwc, y, d->n, d->c_out, p.h_out, p.w_out, stream);
    return rc;
}

}  // extern "C"

// ---- 1-byte activation codes between layers (include/slfp.h; csrc/slfp_codes.hpp) ----
static int codes_route(const slfp_conv2d_desc* d, const slfp_conv2d_io* io, bool has_bias, int relu, ConvPlan* p) {
    // 0: unsupported; 1: depthwise on codes; 2: pointwise on codes; 3: image stem with code output; 4: dense k x k (needs workspace); 5: small-K MFMA stem with code output
    if (!d || !io) return 0;
    if (make_plan(d, p) != SLFP_OK) return 0;
    if (d->x_layout != SLFP_LAYOUT_NHWC || d->y_layout != SLFP_LAYOUT_NHWC) return 0;
    if ((relu & ~(SLFP_POST_RELU | SLFP_POST_LAYEROUT)) != 0 || (relu & SLFP_POST_LAYEROUT)) return 0;
    if (io->y_codes) {
        if (io->y_qbits != 8 && io->y_qbits != 7) return 0;
        if (!(io->y_ka > 0.f) || !scale_div_ok(io->y_ka)) return 0;
        if (!enc_table(io->y_ka, io->y_qbits == 8 ? kFmtAct8 : kFmtSfp7, kEncCode)->valid) return 0;
    }
    if (long_encode_forced()) return 0;
    if (io->x_codes) {
        if (dwc_applicable(*d, *p, has_bias ? reinterpret_cast<const float*>(1) : nullptr, relu)) return 1;
        if (pwc_applicable(*d, *p, relu, io->y_codes != 0)) return 2;
        if (dense_codes_applicable(*d, *p, relu, io->y_codes != 0)) return 4;
        return 0;
    }
    if (io->y_codes && stem_codes_applicable(*d, *p, relu)) return 3;
    if (io->y_codes && dense_codes_applicable(*d, *p, relu, true)) return 4;
    if (io->y_codes && stem_small_codes_applicable(*d, *p, relu)) return 5;
    return 0;
}

extern "C" int slfp_conv2d_codes_supported(const slfp_conv2d_desc* d, const slfp_conv2d_io* io, int has_bias, int relu) {
    ConvPlan p;
    return codes_route(d, io, has_bias != 0, relu, &p) != 0 ? 1 : 0;
}

extern "C" int slfp_conv2d_fwd_codes(const slfp_conv2d_desc* d, const slfp_conv2d_io* io, const void* x, const void* wprep,
                                     const float* bias, const float* post_scale, const float* post_shift, int relu, void* y,
                                     void* stream) {
    return slfp_conv2d_fwd_codes_ws(d, io, x, wprep, bias, post_scale, post_shift, relu, y, nullptr, stream);
}

extern "C" int slfp_conv2d_fwd_codes_ws(const slfp_conv2d_desc* d, const slfp_conv2d_io* io, const void* x, const void* wprep,
                                        const float* bias, const float* post_scale, const float* post_shift, int relu, void* y,
                                        void* workspace, void* stream) {
    if (!d || !io) return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd_codes: null descriptor");
    ConvPlan p;
    int rc = make_plan(d, &p);
    if (rc != SLFP_OK) return rc;
    if (!x || !wprep || !y) return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd_codes: null pointer");
    if ((post_scale == nullptr) != (post_shift == nullptr))
        return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd_codes: post_scale and post_shift must be given together");
    if (!aligned16(x) || !aligned16(y) || !aligned16(wprep) || (bias && !aligned16(bias)) ||
        (post_scale && (!aligned16(post_scale) || !aligned16(post_shift))))
        return fail(SLFP_ERR_ALIGNMENT, "slfp_conv2d_fwd_codes: pointers must be 16-byte aligned");
    const int route = codes_route(d, io, bias != nullptr, relu, &p);
    if (route == 0)
        return fail(SLFP_ERR_UNSUPPORTED, "slfp_conv2d_fwd_codes: this layer / io combination has no code-path kernel "
                                          "(slfp_conv2d_codes_supported); use slfp_conv2d_fwd_post");
    const PostOp post{post_scale, post_shift, (relu & SLFP_POST_RELU) ? 1 : 0, 0};
    const int y_fmt = io->y_qbits == 7 ? kFmtSfp7 : kFmtAct8;
    hipStream_t st = as_stream(stream);
    if (route == 1)
        return launch_dwc(*d, p, reinterpret_cast<const uint8_t*>(x), reinterpret_cast<const float*>(wprep), post, y,
                          io->y_codes != 0, io->y_ka, y_fmt, st);
    if (route == 2)
        return launch_pwc(*d, p, reinterpret_cast<const uint8_t*>(x), wprep, bias, post, y, io->y_codes != 0, io->y_ka, y_fmt, st);
    if (route == 4) {
        const size_t ws_need = workspace_bytes_for(d, p);
        if (ws_need && (!workspace || !aligned16(workspace)))
            return fail(SLFP_ERR_BAD_ARG, "slfp_conv2d_fwd_codes_ws: %zu bytes of 16-byte aligned workspace required (slfp_conv2d_workspace_bytes)", ws_need);
        const CodeIo cio{io->x_codes != 0, io->y_codes != 0, io->y_ka, y_fmt};
        return launch_dense_mfma_io(*d, p, x, wprep, bias, post, y, workspace, cio, st);
    }
    const CodeIo cio{false, true, io->y_ka, y_fmt};
    if (route == 5) return launch_stem_small_io(*d, p, reinterpret_cast<const float*>(x), wprep, bias, post, y, cio, st);
    return launch_stem_codes(*d, p, reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(wprep), bias, post, y, cio, st);
}

extern "C" {

size_t slfp_linear_workspace_bytes(int64_t batch, int64_t in_f, int64_t out_f) {
    slfp_conv2d_desc d;
    memset(&d, 0, sizeof(d));
    d.n = batch; d.c_in = in_f; d.h = 1; d.w = 1; d.c_out = out_f; d.kh = 1; d.kw = 1;
    d.stride_h = d.stride_w = d.dil_h = d.dil_w = d.groups = 1;
    d.x_layout = d.y_layout = SLFP_LAYOUT_NHWC; d.qbits = 8; d.ka = d.kw_scale = 1.f;
    ConvPlan p;
    if (make_plan_core(&d, &p) != SLFP_OK) return 0;  // the plan slfp_linear_fwd uses (no channel re-padding)
    return p.wprep_bytes;
}

static void linear_desc(slfp_conv2d_desc* d, int64_t batch, int64_t in_f, int64_t out_f, float ka, float kw_scale,
                        int qbits, int mfma_passes) {
    // Linear_Q.forward (utils/conv2d_func.py:60-65) = a 1x1 convolution over `batch` pixels,
    // except that the reference divides the bias by Kw first and rescales by Kw first.
    memset(d, 0, sizeof(*d));
    d->n = batch; d->c_in = in_f; d->h = 1; d->w = 1; d->c_out = out_f; d->kh = 1; d->kw = 1;
    d->stride_h = d->stride_w = d->dil_h = d->dil_w = d->groups = 1;
    d->x_layout = d->y_layout = SLFP_LAYOUT_NHWC; d->qbits = qbits; d->ka = ka; d->kw_scale = kw_scale;
    d->mfma_passes = mfma_passes;
}

int slfp_linear_prepare_weights(const float* w, void* wprep, int64_t in_f, int64_t out_f, float kw_scale, int qbits,
                                int mfma_passes, void* stream) {
    slfp_conv2d_desc d;
    linear_desc(&d, 1, in_f, out_f, 1.0f, kw_scale, qbits, mfma_passes);
    ConvPlan p;
    const int rc = make_plan_core(&d, &p);  // no channel re-padding here: odd feature counts take the direct kernel
    if (rc != SLFP_OK) return rc;
    if (!w || !wprep) return fail(SLFP_ERR_BAD_ARG, "slfp_linear_prepare_weights: null pointer");
    if (!aligned16(wprep)) return fail(SLFP_ERR_ALIGNMENT, "slfp_linear_prepare_weights: wprep must be 16-byte aligned");
    return launch_prepare_weights(d, p, w, wprep, nullptr, as_stream(stream));
}

int slfp_linear_fwd_prepared(const float* x, const void* wprep, const float* bias, float* y, int64_t batch, int64_t in_f,
                             int64_t out_f, float ka, float kw_scale, int qbits, int mfma_passes, void* stream) {
    slfp_conv2d_desc d;
    linear_desc(&d, batch, in_f, out_f, ka, kw_scale, qbits, mfma_passes);
    ConvPlan p;
    const int rc = make_plan_core(&d, &p);
    if (rc != SLFP_OK) return rc;
    if (!x || !wprep || !y) return fail(SLFP_ERR_BAD_ARG, "slfp_linear_fwd: null pointer");
    if (!aligned16(x) || !aligned16(y) || !aligned16(wprep) || (bias && !aligned16(bias)))
        return fail(SLFP_ERR_ALIGNMENT, "slfp_linear_fwd: pointers must be 16-byte aligned");
    p.s1 = kw_scale;
    p.s2 = ka;
    const PostOp none{nullptr, nullptr, 0, 0};
    hipStream_t st = as_stream(stream);
    if (p.family == kPointwise) return launch_pointwise(d, p, x, wprep, bias, none, y, st);
    return launch_direct(d, p, x, reinterpret_cast<const float*>(wprep), bias, none, y, st);
}

int slfp_linear_fwd(const float* x, const float* w, const float* bias, float* y, int64_t batch, int64_t in_f,
                    int64_t out_f, float ka, float kw_scale, int qbits, int mfma_passes, void* workspace,
                    void* stream) {
    if (!workspace) return fail(SLFP_ERR_BAD_ARG, "slfp_linear_fwd: null pointer");
    // the reference re-quantizes the weights on every call; so does this entry point
    const int rc = slfp_linear_prepare_weights(w, workspace, in_f, out_f, kw_scale, qbits, mfma_passes, stream);
    if (rc != SLFP_OK) return rc;
    return slfp_linear_fwd_prepared(x, workspace, bias, y, batch, in_f, out_f, ka, kw_scale, qbits, mfma_passes, stream);
}

}  // extern "C"
