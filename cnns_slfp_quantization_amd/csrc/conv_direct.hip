// conv_direct.hip -- generic SLFP-quantized direct convolution (any kernel size, stride,
// padding, dilation, groups, channel count), NHWC, float32 FMA, gfx950.
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) for every geometry
// the specialised kernels do not take: the 3x3/7x7/11x11 stems (C_in = 3), dense kxk
// layers, odd channel counts (ShuffleNetV2's 58), dilation.  One workgroup = one image,
// an 8x8 output tile, one group, 64 output channels; the input halo tile is quantized
// ONCE per element into LDS in C_in chunks; each thread keeps a 4-pixel x 4-channel
// register tile and reads its weights (prepared as [KH][KW][C_in/g][C_out], 16-byte
// loads, broadcast through L1) straight from global memory.
#include "slfp_device.hpp"
#include "slfp_host.hpp"

namespace slfp {

constexpr int kDirThreads = 256;
constexpr int kDirTH = 8, kDirTW = 8, kDirOC = 64;

struct DirParams {
    int N, H, W, C, O, KH, KW;
    int sh, sw, ph, pw, dh, dw;
    int groups, Cg, Og, Ho, Wo;
    int tiles_h, tiles_w, oc_chunks;
    int CC, IH, IW;
    ScaleDiv sd;
    float s1, s2;
    uint32_t nblocks;
};

template <int FMT>
__global__ __launch_bounds__(kDirThreads) void k_direct(const float* __restrict__ x, const float* __restrict__ wq,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        const DirParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sT = reinterpret_cast<uint32_t*>(smem);
    float* tile = reinterpret_cast<float*>(smem + 64);  // [IH][IW][CC]
    lut_fill<FMT>(sT);

    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const int oc = b % p.oc_chunks; b /= p.oc_chunks;
    const int g = b % p.groups; b /= p.groups;
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b;

    const int cgi = threadIdx.x & 15;  // which 4 output channels of the 64-chunk
    const int pg = threadIdx.x >> 4;   // which 4 pixels of the 8x8 tile (consecutive in w)
    const int oh = pg >> 1, ow0 = (pg & 1) * 4;
    const int o_in_g = oc * kDirOC + cgi * 4;          // first output channel inside the group
    const int o0 = g * p.Og + o_in_g;                  // global output channel
    int n_o = p.Og - o_in_g;                           // live channels of this thread (<= 4)
    n_o = n_o < 0 ? 0 : (n_o > 4 ? 4 : n_o);
    const bool o_vec = (n_o == 4) && ((p.O & 3) == 0) && ((o0 & 3) == 0);

    const int h_in0 = th * kDirTH * p.sh - p.ph, w_in0 = tw * kDirTW * p.sw - p.pw;

    float acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;

    for (int cc0 = 0; cc0 < p.Cg; cc0 += p.CC) {
        const int cc = (p.Cg - cc0) < p.CC ? (p.Cg - cc0) : p.CC;
        __syncthreads();  // previous chunk fully consumed (also orders the LUT fill)
        const int n_in = p.IH * p.IW * cc;
        for (int item = threadIdx.x; item < n_in; item += kDirThreads) {
            const int c = item % cc;
            const int pix = item / cc;
            const int iw = pix % p.IW, ih = pix / p.IW;
            const int gh = h_in0 + ih, gw = w_in0 + iw;
            float v = 0.f;
            if (gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) {
                const float raw = x[(((size_t)n * p.H + gh) * p.W + gw) * p.C + g * p.Cg + cc0 + c];
                v = quantize_scaled<FMT>(raw, p.sd, sT);
            }
            tile[(ih * p.IW + iw) * p.CC + c] = v;
        }
        __syncthreads();
        if (n_o > 0) {
            for (int kh = 0; kh < p.KH; ++kh) {
                for (int kw = 0; kw < p.KW; ++kw) {
                    const float* arow = tile + ((oh * p.sh + kh * p.dh) * p.IW + ow0 * p.sw + kw * p.dw) * p.CC;
                    const float* wrow = wq + ((size_t)(kh * p.KW + kw) * p.Cg + cc0) * p.O + o0;
                    for (int c = 0; c < cc; ++c) {
                        float w[4];
                        if (o_vec) {
                            const float4 w4 = *reinterpret_cast<const float4*>(wrow + (size_t)c * p.O);
                            w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) w[r] = r < n_o ? wrow[(size_t)c * p.O + r] : 0.f;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float a = arow[q * p.sw * p.CC + c];
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[q][r] = fmaf(a, w[r], acc[q][r]);
                        }
                    }
                }
            }
        }
    }

    if (n_o == 0) return;
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < n_o) bq[r] = (bias[o0 + r] / p.s1) / p.s2;  // bias_q (conv2d_func.py:44)
    }
    const int goh = th * kDirTH + oh;
    if (goh >= p.Ho) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gow = tw * kDirTW + ow0 + q;
        if (gow >= p.Wo) continue;
        float r4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) r4[r] = ((acc[q][r] + bq[r]) * p.s1) * p.s2;
        float* dst = y + (((size_t)n * p.Ho + goh) * p.Wo + gow) * p.O + o0;
        if (o_vec) {
            *reinterpret_cast<float4*>(dst) = make_float4(r4[0], r4[1], r4[2], r4[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < n_o) dst[r] = r4[r];
        }
    }
}

int launch_direct(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const float* wq_hwio,
                  const float* bias, float* y, hipStream_t stream) {
    DirParams p;
    p.N = (int)d.n; p.H = (int)d.h; p.W = (int)d.w; p.C = (int)d.c_in; p.O = (int)d.c_out;
    p.KH = (int)d.kh; p.KW = (int)d.kw;
    p.sh = d.stride_h; p.sw = d.stride_w; p.ph = d.pad_h; p.pw = d.pad_w; p.dh = d.dil_h; p.dw = d.dil_w;
    p.groups = d.groups; p.Cg = p.C / p.groups; p.Og = p.O / p.groups;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out;
    p.tiles_h = (int)ceil_div(p.Ho, kDirTH);
    p.tiles_w = (int)ceil_div(p.Wo, kDirTW);
    p.oc_chunks = (int)ceil_div(p.Og, kDirOC);
    p.IH = (kDirTH - 1) * p.sh + (p.KH - 1) * p.dh + 1;
    p.IW = (kDirTW - 1) * p.sw + (p.KW - 1) * p.dw + 1;
    const int budget = 12 * 1024;  // floats (48 KiB)
    int cc = budget / (p.IH * p.IW);
    if (cc < 1) return fail(SLFP_ERR_UNSUPPORTED, "direct conv: %dx%d input tile does not fit LDS", p.IH, p.IW);
    if (cc > 32) cc = 32;
    if (cc > p.Cg) cc = p.Cg;
    p.CC = cc;
    p.sd = make_scale_div(d.ka); p.s1 = plan.s1; p.s2 = plan.s2;
    const int64_t nblocks = (int64_t)p.N * p.tiles_h * p.tiles_w * p.groups * p.oc_chunks;
    if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "direct conv: grid too large");
    p.nblocks = (uint32_t)nblocks;
    const size_t lds = 64 + (size_t)p.IH * p.IW * p.CC * sizeof(float);
    if (plan.fmt_act == kFmtAct8)
        hipLaunchKernelGGL((k_direct<kFmtAct8>), dim3(p.nblocks), dim3(kDirThreads), lds, stream, x, wq_hwio, bias, y, p);
    else
        hipLaunchKernelGGL((k_direct<kFmtSfp7>), dim3(p.nblocks), dim3(kDirThreads), lds, stream, x, wq_hwio, bias, y, p);
    return check_launch("slfp direct conv kernel");
}

}  // namespace slfp
