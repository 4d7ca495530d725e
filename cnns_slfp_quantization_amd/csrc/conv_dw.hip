// conv_dw.hip -- SLFP-quantized depthwise 3x3 convolution, NHWC, gfx950.
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25) for groups == C_in == C_out,
// 3x3 kernels (the 13 "dw" layers of MobileNetV1, nets_imgnet/mobilenetv1.py:27).
//
// HBM-bound (1.7 flop/B): algorithmic traffic is 4 B read + 4 B written per element.
//   * one workgroup = one (image, TH x TW output tile, CB-channel group);
//   * LOAD phase: the (TH-1)*S+3 x (TW-1)*S+3 input halo tile is read from HBM with
//     16-byte loads, channels across lanes (a pixel's CB channels are one contiguous
//     128/256-byte segment), x/Ka + SLFP encode applied ONCE per element inline, and the
//     dequantized float32 written to LDS (zero where the conv pads);
//   * COMPUTE phase: every thread owns 4 channels (its 9x4 weights live in registers),
//     reads 9 ds_read_b128 per output float4, FMAs in float32, rescales *Ka*Kw with the
//     reference's two roundings, and stores 16 bytes.
// For stride 2 the tile's columns are stored de-interleaved (even columns, then odd) so
// that the 8 pixels a wave reads for one tap are contiguous in LDS (no bank conflicts).
// Tried and dropped (same-box A/B, profiles/ab_compare.sh): keeping the 16-entry encode table in a VGPR and
// reading it with ds_bpermute so that the exactly-32-KiB 16x16x32 tile lets five workgroups share a CU
// instead of four: 2.7 % slower (the crossbar lookup costs more than the fifth workgroup brings); unrolling the
// compute loop by two (18 tap reads in flight per thread): 3 % slower at the same 94 VGPRs.
// rocprof (profiles/r01a) showed HBM traffic already ideal (halo re-reads hit L2) and the
// kernel VALU-bound, so all index arithmetic is incremental (no runtime div/mod in loops).
#include "slfp_device.hpp"
#include "slfp_host.hpp"

namespace slfp {

constexpr int kDwThreads = 256;

struct DwParams {
    int N, H, W, C, Ho, Wo;
    int TH, TW, tiles_h, tiles_w;
    int CB, cb4_shift, cgroups;  // channels per block (4 << cb4_shift), #channel groups
    int pad;
    int IH, IW, IWh;             // input tile extent; IWh = (IW+1)/2 (stride-2 de-interleave)
    int out_step_h, out_step_w;  // same for the output tile
    ScaleDiv sd;
    float ka, kw;
    PostOp post;
    uint32_t nblocks;
    EncArgs enc;                 // threshold table of QA(x / Ka) (TAB kernels; slfp_enc.hpp)
};

// V floats per lane: 4 (16-byte accesses) or 2 (channel counts that are even but not a multiple of 4,
// ShuffleNetV2's 58: pixel rows are only 8-byte aligned).
template <int V>
struct alignas(V * 4) DwVec {
    float v[V];
};

// CBT / IWT > 0: channel-group width and input-tile width known at compile time (the MobileNetV1
// tiles): every LDS offset becomes an immediate and the row-wrap arithmetic folds away.
// LO: the fused SFP<4,4> layer-output quantizer (PostOp::layerout) is compiled in; kept out of the default
// instantiations, whose store loop it slowed by 1 % (same-box A/B) even when not taken.
// TAB: the input quantizer is the threshold table of slfp_enc.hpp (7 VALU instructions per element) instead of
// the long form (22): the kernel was VALU-issue-bound on it (profiles/r01h).
template <int FMT, int S, int CBT, int IWT, int V = 4, bool LO = false, bool TAB = false>
__global__ __launch_bounds__(kDwThreads) void k_dw3x3(const float* __restrict__ x, const float* __restrict__ wq,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      const DwParams p) {
    using Vec = DwVec<V>;
    constexpr int VSH = V == 4 ? 0 : 1;   // log2(4 / V): lanes per pixel double with 2-float lanes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sT = reinterpret_cast<uint32_t*>(smem);      // 16 dwords (long form) or the 2 KiB threshold table
    constexpr int kTabBytes = TAB ? ((kEncEntries * 8 + 15) & ~15) : 64;
    float* tile = reinterpret_cast<float*>(smem + kTabBytes);     // [IH][IW][CB]
    if constexpr (TAB) enc_fill<kDwThreads>(reinterpret_cast<uint2*>(smem), p.enc);
    else lut_fill<FMT>(sT);
    const int CB = CBT > 0 ? CBT : p.CB;
    const int IW = IWT > 0 ? IWT : p.IW;
    const int IWh = (IW + 1) / 2;
    const int cb4_shift = (CBT == 32 ? 3 : (CBT == 64 ? 4 : p.cb4_shift)) + VSH;  // log2(lanes per pixel)
    const int dp = kDwThreads >> cb4_shift;  // pixels between two items of a thread
    const int in_step_h = dp / IW, in_step_w = dp - in_step_h * IW;

    // logical block id -> (channel group, tile_w, tile_h, image); XCD-contiguous so that
    // tiles sharing a halo are served by the same L2.
    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const int cg = b % p.cgroups; b /= p.cgroups;
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b;

    const int cb4 = 1 << cb4_shift;
    const int my_c4 = threadIdx.x & (cb4 - 1);  // constant per thread: cb4 divides the block size
    const int my_c = cg * CB + my_c4 * V;
    const bool c_live = my_c < p.C;
    const int pix0 = threadIdx.x >> cb4_shift;
    const int h_in0 = th * p.TH * S - p.pad, w_in0 = tw * p.TW * S - p.pad;
    Vec zero;
#pragma unroll
    for (int e = 0; e < V; ++e) zero.v[e] = 0.f;

    // this thread's V channels x 9 taps: issued first so that their (L2) latency hides under the tile loads
    Vec wt[9];
    Vec bq = zero;
    // fused BN/ReLU vectors of this thread's channels: loaded once, not per output
    Vec psc, psh;
#pragma unroll
    for (int e = 0; e < V; ++e) { psc.v[e] = 1.f; psh.v[e] = 0.f; }
    if (c_live) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const Vec*>(wq + (size_t)t * p.C + my_c);
        if (bias) {  // bias_q = bias / Ka / Kw (utils/conv2d_func.py:44)
            const Vec bb = *reinterpret_cast<const Vec*>(bias + my_c);
#pragma unroll
            for (int e = 0; e < V; ++e) bq.v[e] = (bb.v[e] / p.ka) / p.kw;
        }
        if (p.post.scale) {
            psc = *reinterpret_cast<const Vec*>(p.post.scale + my_c);
            psh = *reinterpret_cast<const Vec*>(p.post.shift + my_c);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = zero;
    }
    __syncthreads();

    // ---------------- LOAD + ENCODE phase ----------------
    // 32-bit element offsets advanced incrementally (64-bit multiplies per item cost more than
    // the encode itself: profiles/r01c).
    {
        const int n_pix = p.IH * IW;
        int pix = pix0;
        int ih = pix0 / IW, iw = pix0 - ih * IW;  // the only division: once per thread
        int gh = h_in0 + ih, gw = w_in0 + iw;
        const float* xn = x + (size_t)n * p.H * p.W * p.C + my_c;
        int goff = (gh * p.W + gw) * p.C;                               // may be negative outside the image
        int lrow = ih * IW * CB + my_c4 * V;                         // LDS offset of the tile row
        const int g_step = (in_step_h * p.W + in_step_w) * p.C, g_wrap = (p.W - IW) * p.C;
        const int l_step = in_step_h * IW * CB, l_wrap = IW * CB;
        constexpr int U = 4;  // loads kept in flight per thread: 4 keeps the kernel at 94 VGPRs (5 waves/SIMD where LDS allows); 8 (one batch per 16x16x32 tile, 114-124 VGPRs) measured 1 % slower in a same-box A/B
        while (pix < n_pix) {
            Vec v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = pix < n_pix;
                const bool inb = live && c_live && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
                const int slot = (S == 2) ? ((iw & 1) * IWh + (iw >> 1)) : iw;
                dst[u] = live ? lrow + slot * CB : -1;
                v[u] = zero;
                if (inb) v[u] = *reinterpret_cast<const Vec*>(xn + (uint32_t)goff);
                pix += dp;
                gh += in_step_h; gw += in_step_w; iw += in_step_w;
                goff += g_step; lrow += l_step;
                if (in_step_w != 0 && iw >= IW) { iw -= IW; gw -= IW; ++gh; goff += g_wrap; lrow += l_wrap; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (dst[u] < 0) continue;
                Vec q;
                // Q(0/Ka) == 0, so padded / out-of-range zeros go through the same path.
                if constexpr (TAB) {
#pragma unroll
                    for (int e = 0; e < V; ++e) q.v[e] = enc_f32(v[u].v[e], p.enc.r1, p.enc.lo, p.enc.hi, smem);
                    bool un = __builtin_isunordered(v[u].v[0], v[u].v[1]);
                    if constexpr (V == 4) un |= __builtin_isunordered(v[u].v[2], v[u].v[3]);
                    if (__builtin_expect(un, 0)) {   // NaN in -> NaN out (never taken on real activations)
#pragma unroll
                        for (int e = 0; e < V; ++e) q.v[e] = v[u].v[e] != v[u].v[e] ? __uint_as_float(kBitsQNaN) : q.v[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e) q.v[e] = quantize_scaled<FMT>(v[u].v[e], p.sd, sT);
                }
                *reinterpret_cast<Vec*>(tile + dst[u]) = q;
            }
        }
    }

    __syncthreads();

    // ---------------- COMPUTE phase ----------------
    {
        const int n_pix = p.TH * p.TW;
        int pix = pix0;
        int oh = pix0 / p.TW, ow = pix0 - oh * p.TW;
        int goh = th * p.TH + oh, gow = tw * p.TW + ow;
        float* yn = y + (size_t)n * p.Ho * p.Wo * p.C + my_c;
        int yoff = (goh * p.Wo + gow) * p.C;
        int lbase = ((oh * S) * IW + (S == 1 ? ow : 0)) * CB + my_c4 * V;  // tile offset of the window's first tap
        const int y_step = (p.out_step_h * p.Wo + p.out_step_w) * p.C, y_wrap = (p.Wo - p.TW) * p.C;
        const int l_step = (p.out_step_h * S * IW + (S == 1 ? p.out_step_w : 0)) * CB;
        const int l_wrap = (S * IW - (S == 1 ? p.TW : 0)) * CB;
        const int row_pitch = IW * CB;
        for (; pix < n_pix; pix += dp) {
            if (goh < p.Ho && gow < p.Wo && c_live) {
                Vec acc = bq;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* row = tile + lbase + kh * row_pitch;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        int o;
                        if (S == 1) {
                            o = kw * CB;
                        } else {
                            const int iw = ow * 2 + kw;
                            o = ((iw & 1) * IWh + (iw >> 1)) * CB;
                        }
                        const Vec a = *reinterpret_cast<const Vec*>(row + o);
                        const Vec w = wt[kh * 3 + kw];
#pragma unroll
                        for (int e = 0; e < V; ++e) acc.v[e] = fmaf(a.v[e], w.v[e], acc.v[e]);
                    }
                }
                Vec r;  // (out * Ka) * Kw: two float32 roundings, as utils/conv2d_func.py:24; then the fused BN / ReLU
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float t = (acc.v[e] * p.ka) * p.kw;
                    if (p.post.scale) {
                        t = __builtin_fmaf(t, psc.v[e], psh.v[e]);
                        if constexpr (LO) t = layerout1(t);
                    }
                    if (p.post.relu) t = fmaxf(t, 0.f);
                    r.v[e] = t;
                }
                *reinterpret_cast<Vec*>(yn + (uint32_t)yoff) = r;
            }
            oh += p.out_step_h; goh += p.out_step_h;
            ow += p.out_step_w; gow += p.out_step_w;
            yoff += y_step; lbase += l_step;
            if (ow >= p.TW) { ow -= p.TW; gow -= p.TW; ++oh; ++goh; yoff += y_wrap; lbase += l_wrap; }
        }
    }
}

int launch_dw3x3(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const float* wq9c,
                 const float* bias, const PostOp& post, float* y, hipStream_t stream) {
    if (dw3x3_tile_applicable(d, plan, bias, post)) return launch_dw3x3_tile(d, plan, x, wq9c, post, y, stream);
    DwParams p;
    p.post = post;
    p.N = (int)d.n; p.H = (int)d.h; p.W = (int)d.w; p.C = (int)d.c_in;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out;
    const int S = d.stride_h;
    // channel group: largest power-of-two multiple of 4 (<= 64) dividing C; channel counts that are
    // not a multiple of 32 (ShuffleNetV2: 24, 116, 232) get 32-wide groups with a ragged last one
    // (masked lanes) rather than 4- or 8-channel groups whose pixels are 16-32 byte fragments
    int CB = 4;
    while (CB < 64 && p.C % (CB * 2) == 0) CB *= 2;
    if (CB < 32) {
        CB = 4;
        while (CB < 32 && CB < p.C) CB *= 2;
    }
    const int tmax = (S == 1) ? 14 : 7;
    p.TH = p.Ho < tmax ? p.Ho : tmax;
    p.TW = p.Wo < tmax ? p.Wo : tmax;
    p.tiles_h = (int)ceil_div(p.Ho, p.TH);
    p.tiles_w = (int)ceil_div(p.Wo, p.TW);
    p.IH = (p.TH - 1) * S + 3;
    p.IW = (p.TW - 1) * S + 3;
    // keep the halo tile within 40 KiB of LDS (>= 3 workgroups per CU)
    while (CB > 4 && (size_t)p.IH * p.IW * CB * sizeof(float) > 40 * 1024) CB /= 2;
    p.CB = CB;
    p.cb4_shift = 0;   // log2(CB / 4); the 2-float kernels add 1
    while ((4 << p.cb4_shift) < CB) ++p.cb4_shift;
    p.cgroups = (int)ceil_div(p.C, CB);
    p.pad = d.pad_h;
    p.IWh = (p.IW + 1) / 2;
    const bool v2 = (p.C % 4) != 0;   // even channel count (58): 8-byte lanes, twice as many lanes per pixel
    const int dp = kDwThreads >> (p.cb4_shift + (v2 ? 1 : 0));
    p.out_step_h = dp / p.TW; p.out_step_w = dp % p.TW;
    p.sd = make_scale_div(d.ka);
    p.ka = d.ka; p.kw = d.kw_scale;
    if ((int64_t)p.H * p.W * p.C >= (1ll << 29) || (int64_t)p.Ho * p.Wo * p.C >= (1ll << 29))
        return fail(SLFP_ERR_UNSUPPORTED, "dw3x3: one image exceeds 2^29 elements");
    const int64_t nblocks = (int64_t)p.N * p.tiles_h * p.tiles_w * p.cgroups;
    if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "dw3x3: grid too large");
    p.nblocks = (uint32_t)nblocks;
    const EncArgs* tab = act_table(d.ka, plan.fmt_act, kEncF32);
    if (tab) p.enc = *tab;
    const size_t lds = (tab ? ((kEncEntries * 8 + 15) & ~15) : 64) + (size_t)p.IH * p.IW * p.CB * sizeof(float);
    if (lds > 64 * 1024) return fail(SLFP_ERR_UNSUPPORTED, "dw3x3: tile needs %zu B of LDS", lds);
    const bool a8 = plan.fmt_act == kFmtAct8;
#define SLFP_DW_LAUNCH(FMT, SS, CBT, IWT, VV) \
    do { if (tab) hipLaunchKernelGGL((k_dw3x3<FMT, SS, CBT, IWT, VV, false, true>), dim3(p.nblocks), dim3(kDwThreads), lds, stream, x, wq9c, bias, y, p); \
         else hipLaunchKernelGGL((k_dw3x3<FMT, SS, CBT, IWT, VV>), dim3(p.nblocks), dim3(kDwThreads), lds, stream, x, wq9c, bias, y, p); } while (0)
#define SLFP_DW_LAUNCH_LO(FMT, SS, VV) \
    do { if (tab) hipLaunchKernelGGL((k_dw3x3<FMT, SS, 0, 0, VV, true, true>), dim3(p.nblocks), dim3(kDwThreads), lds, stream, x, wq9c, bias, y, p); \
         else hipLaunchKernelGGL((k_dw3x3<FMT, SS, 0, 0, VV, true>), dim3(p.nblocks), dim3(kDwThreads), lds, stream, x, wq9c, bias, y, p); } while (0)
#define SLFP_DW_BY_FMT(SS, CBT, IWT, VV) \
    do { if (a8) SLFP_DW_LAUNCH(kFmtAct8, SS, CBT, IWT, VV); else SLFP_DW_LAUNCH(kFmtSfp7, SS, CBT, IWT, VV); } while (0)
    if (p.post.layerout && p.post.scale) {   // fused layer-output quantizer: the generic-geometry instantiations only
        if (a8) { if (S == 1) { if (v2) SLFP_DW_LAUNCH_LO(kFmtAct8, 1, 2); else SLFP_DW_LAUNCH_LO(kFmtAct8, 1, 4); }
                  else        { if (v2) SLFP_DW_LAUNCH_LO(kFmtAct8, 2, 2); else SLFP_DW_LAUNCH_LO(kFmtAct8, 2, 4); } }
        else    { if (S == 1) { if (v2) SLFP_DW_LAUNCH_LO(kFmtSfp7, 1, 2); else SLFP_DW_LAUNCH_LO(kFmtSfp7, 1, 4); }
                  else        { if (v2) SLFP_DW_LAUNCH_LO(kFmtSfp7, 2, 2); else SLFP_DW_LAUNCH_LO(kFmtSfp7, 2, 4); } }
    }
    else if (v2) {
        if (S == 1 && p.CB == 32 && p.IW == 16) SLFP_DW_BY_FMT(1, 32, 16, 2);
        else if (S == 1) SLFP_DW_BY_FMT(1, 0, 0, 2);
        else SLFP_DW_BY_FMT(2, 0, 0, 2);
    }
    else if (S == 1 && p.CB == 32 && p.IW == 16) SLFP_DW_BY_FMT(1, 32, 16, 4);      // 14x14 tiles (112..14 px layers)
    else if (S == 1 && p.CB == 64 && p.IW == 9) SLFP_DW_BY_FMT(1, 64, 9, 4);  // 7x7 images, >= 64 channels
    else if (S == 2 && p.CB == 32 && p.IW == 15) SLFP_DW_BY_FMT(2, 32, 15, 4); // 7x7 output tiles of stride-2 layers
    else if (S == 1) SLFP_DW_BY_FMT(1, 0, 0, 4);
    else SLFP_DW_BY_FMT(2, 0, 0, 4);
#undef SLFP_DW_BY_FMT
#undef SLFP_DW_LAUNCH_LO
#undef SLFP_DW_LAUNCH
    return check_launch("slfp dw3x3 kernel");
}

}  // namespace slfp
