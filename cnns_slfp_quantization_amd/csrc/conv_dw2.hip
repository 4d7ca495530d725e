// conv_dw2.hip -- SLFP-quantized depthwise 3x3 convolution, NHWC, gfx950: the straight-line tile kernel.
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25) for groups == C_in == C_out, 3x3, stride 1 / 2,
// C a multiple of 32 (the 13 "dw" layers of MobileNetV1, nets_imgnet/mobilenetv1.py:27).  Same arithmetic
// as conv_dw.hip (the general kernel, which keeps the odd shapes): input quantized once per element on the
// load path, float32 FMAs in (kh, kw) order, (acc * Ka) * Kw with the reference's two roundings.
//
// What changed, and why (profiles/r02a, r02b; profiles/variants.py):
//   * every global access is a buffer_load / buffer_store through a per-image descriptor: out-of-image halo
//     pixels and ragged tile edges get an out-of-range offset (the range check returns 0 / drops the store),
//     so the body is ONE straight line: all 8 x 16-byte loads of the halo tile are issued back to back and
//     hipcc counts them exactly (conv_dw.hip's per-item bounds branches made it wait vmcnt(0) before every
//     encode and split the encode stream into 4-element pieces);
//   * the quantizer is the threshold table of slfp_enc.hpp (7 VALU instructions per element, was 22);
//   * halo and output geometry are fixed at 16 (8) pixel slots per row (lane = (pixel slot, 4 channels)):
//     every LDS and global offset is `per-thread constant + immediate`, no index arithmetic is left in the loops.
// One workgroup = one (image, 14x14 | 7x7 output tile, 32-channel group), many small workgroups in flight: a
// persistent, software-pipelined variant (loads of tile t+1 in flight under the convolution of tile t, 4 / 8 /
// 16 tiles per workgroup) was built and measured 2-12 % SLOWER on every layer -- the hardware's workgroup
// dispatcher balances 16 k small workgroups better than a static partition, and four co-resident workgroups in
// different phases already overlap loads with compute.  Same-box, interleaved (us per layer, batch 256):
// conv_dw.hip + table 1058 total, this kernel 984 (5.2 TB/s algorithmic), persistent x16 1126.
// Neighbouring tiles share 2-pixel halos; logical workgroup ids are dealt to XCDs contiguously (xcd_remap) so
// the re-reads hit the same L2 (HBM reads = algorithmic bytes, profiles/r02c).
#include <cstdlib>
#include "slfp_device.hpp"
#include "slfp_enc.hpp"
#include "slfp_host.hpp"

namespace slfp {

constexpr int kDw2Threads = 256;
constexpr int kDw2Tab = (kEncEntries * 8 + 15) & ~15;   // LDS bytes of the threshold table
constexpr uint32_t kOob = 0xFFFFFFF0u;                  // byte offset beyond any descriptor: reads 0, stores nothing

struct Dw2Params {
    int N, H, W, C, Ho, Wo;
    int tiles_h, tiles_w, cgroups;
    int pad;
    uint32_t ntiles;      // N * tiles_h * tiles_w: tiles per channel group
    uint32_t nblocks;
    float ka, kw;
    int nt_out;           // store the output with the nt hint (set by the launcher: outputs too large for the Infinity Cache)
    PostOp post;
    EncArgs enc;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// S: stride (1, 2); TH x TW: output tile (14x14 or 7x7 for stride 1, 7x7 for stride 2); POST: fused
// per-channel scale/shift (eval-mode BatchNorm) present.
template <int S, int TH, int TW, bool POST>
__global__ __launch_bounds__(kDw2Threads, 4) void k_dw3x3_tile(const float* __restrict__ x, const float* __restrict__ wq,
                                                               float* __restrict__ y, const Dw2Params p) {
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    static_assert(IW <= 16 && IH <= 16, "halo tile is at most 16 x 16 pixel slots");
    constexpr int NI = (IH + 1) / 2;            // load items per thread: rows 2i + (slot >> 4)
    constexpr int ROWB = 16 * 32 * 4;           // LDS bytes per halo row: 16 pixel slots x 32 channels

    // static LDS: the table sits at LDS address 0, a compile-time constant, so every table lookup is `ds_read_b64 v, bin`
    // without the add of a (link-time) dynamic-LDS base that hipcc otherwise emits per lookup
    __shared__ __attribute__((aligned(16))) unsigned char smem[kDw2Tab + 2 * NI * ROWB];
    unsigned char* tile = smem + kDw2Tab;       // [2 * NI rows][16 slots][32 ch] float32
    enc_fill<kDw2Threads>(reinterpret_cast<uint2*>(smem), p.enc);
    const float r1 = p.enc.r1, lo = p.enc.lo, hi = p.enc.hi;

    const uint32_t lb = xcd_remap(blockIdx.x, p.nblocks);
    const int cg = (int)(lb % (uint32_t)p.cgroups);
    const uint32_t t = lb / (uint32_t)p.cgroups;   // tile: (image, tile row, tile column)

    const int c4 = threadIdx.x & 7, slot = threadIdx.x >> 3;   // 8 lanes per pixel, 32 pixel slots
    const int c = cg * 32 + c4 * 4;
    const int iw = slot & 15, ihh = slot >> 4;

    // this thread's 4 channels x 9 taps, fused BN vectors (layers with a bias take conv_dw.hip)
    f32x4 wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const f32x4*>(wq + (size_t)k * p.C + c);
    f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (POST) {
        psc = *reinterpret_cast<const f32x4*>(p.post.scale + c);
        psh = *reinterpret_cast<const f32x4*>(p.post.shift + c);
    }

    // per-thread constants of the halo load / LDS write
    const int colslot = (S == 2) ? ((iw & 1) * 8 + (iw >> 1)) : iw;          // stride 2: even columns, then odd
    const uint32_t lds_w = (uint32_t)((ihh * 16 + colslot) * 128 + c4 * 16);  // + i * 2 * ROWB
    const bool col_live = iw < IW;
    // outputs: the 32 pixel slots are CW columns x 32/CW rows per step (16 x 2 for 14-wide tiles, 8 x 4 for 7-wide
    // ones), so every per-step offset is `constant + step * immediate` and no per-output index math is left
    constexpr int CW = TW > 8 ? 16 : 8, RPI = 32 / CW, NO = (TH + RPI - 1) / RPI;
    const int ow = slot & (CW - 1), ohb = slot / CW;
    const bool ocol_live = ow < TW;
    const uint32_t lds_r0 = (uint32_t)(((ohb * S) * 16 + ow) * 128 + c4 * 16);           // + j * RPI * S * ROWB
    const uint32_t out_rel0 = (uint32_t)((ohb * p.Wo + ow) * p.C + c) * 4u;              // + j * RPI * Wo * C * 4
    const uint32_t out_step = (uint32_t)(RPI * p.Wo * p.C) * 4u;
    const uint32_t img_in_bytes = (uint32_t)p.H * p.W * p.C * 4u, img_out_bytes = (uint32_t)p.Ho * p.Wo * p.C * 4u;
    const uint32_t tiles_per_img = (uint32_t)(p.tiles_h * p.tiles_w);

    const uint32_t n = t / tiles_per_img, tr = t - n * tiles_per_img;
    const int th = (int)(tr / (uint32_t)p.tiles_w), tw = (int)(tr - (uint32_t)th * p.tiles_w);

    // ---- all halo loads of the tile, back to back
    f32x4 v[NI];
    {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(x + (size_t)n * p.H * p.W * p.C, img_in_bytes);
        const int gw = tw * TW * S - p.pad + iw;
        const int gh0 = th * TH * S - p.pad + ihh;
        const bool w_ok = col_live && (unsigned)gw < (unsigned)p.W;
        const uint32_t off0 = (uint32_t)((gh0 * p.W + gw) * p.C + c) * 4u;   // wraps for negative rows: only used when in range
        const uint32_t step = (uint32_t)(2 * p.W * p.C) * 4u;
        uint32_t voff[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            bool ok = w_ok && (unsigned)(gh0 + 2 * i) < (unsigned)p.H;
            if (2 * i + 1 >= IH) ok = ok && ihh == 0;          // the odd last halo row does not exist
            voff[i] = ok ? off0 + (uint32_t)i * step : kOob;
            asm volatile("" : "+v"(voff[i]));                   // a value, not control flow: hipcc otherwise branches around the load
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
            v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, (SLFP_NT_DW & 1) ? 2 : 0));
    }
    __syncthreads();   // threshold table visible

    // ---- quantize the tile into LDS: Q(0) == 0, so padding and dead slots go through the same path
    bool any_nan = false;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float4 xi = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        any_nan |= enc_has_nan4(xi);
        *reinterpret_cast<float4*>(tile + lds_w + (uint32_t)i * 2u * ROWB) = enc4_f32_raw(xi, r1, lo, hi, smem);
    }
    if (__builtin_expect(any_nan, 0)) {   // NaN in -> NaN out; never taken on real activations
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[i][e] != v[i][e]) reinterpret_cast<uint32_t*>(tile + lds_w + (uint32_t)i * 2u * ROWB)[e] = kBitsQNaN;
    }
    __syncthreads();

    // ---- convolve + store
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + (size_t)n * p.Ho * p.Wo * p.C, img_out_bytes);
    const int oh0 = th * TH, ow0 = tw * TW;
    const uint32_t org = (uint32_t)((oh0 * p.Wo + ow0) * p.C) * 4u;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tap = j * RPI * S * ROWB + ((S == 1) ? (kh * ROWB + kw * 128) : (kh * ROWB + ((kw & 1) * 8 + (kw >> 1)) * 128));
                const f32x4 a = *reinterpret_cast<const f32x4*>(tile + lds_r0 + tap);
                const f32x4 w = wt[kh * 3 + kw];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(a[e], w[e], acc[e]);
            }
        }
        f32x4 rr;   // (out * Ka) * Kw: two float32 roundings, as utils/conv2d_func.py:24; then the fused BN / ReLU
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = (acc[e] * p.ka) * p.kw;
            if constexpr (POST) u = __builtin_fmaf(u, psc[e], psh[e]);
            if (p.post.relu) u = fmaxf(u, 0.f);
            rr[e] = u;
        }
        const bool live = ocol_live && (oh0 + ohb + j * RPI) < p.Ho && (j * RPI + ohb) < TH && (ow0 + ow) < p.Wo;
        uint32_t so = live ? org + out_rel0 + (uint32_t)j * out_step : kOob;
        asm volatile("" : "+v"(so));
        // nt for outputs the 256 MiB Infinity Cache cannot hold anyway (the store then does not displace what the next
        // kernel reads); smaller outputs are stored normally so that the layer that consumes them finds them in the cache
        typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
        if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, rr), ry, so, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, rr), ry, so, 0, 0);
    }
}

bool dw3x3_tile_applicable(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* bias, const PostOp& post) {
    if (bias) return false;
    if (long_encode_forced() || dw_old_forced()) return false;
    if (plan.family != kDw3x3 || plan.repad) return false;
    if (d.c_in % 32 != 0 || post.layerout) return false;
    if (d.pad_h > 2) return false;
    if (d.stride_h == 1 && plan.h_out <= 7 && plan.w_out <= 7) return false;   // 7x7 images: conv_dw.hip's 64-channel tiles are faster
    if ((int64_t)d.h * d.w * d.c_in >= (1ll << 29) || plan.h_out * plan.w_out * d.c_in >= (1ll << 29)) return false;
    return act_table(d.ka, plan.fmt_act, kEncF32) != nullptr;
}

int launch_dw3x3_tile(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const float* wq9c,
                      const PostOp& post, float* y, hipStream_t stream) {
    Dw2Params p;
    p.post = post;
    p.N = (int)d.n; p.H = (int)d.h; p.W = (int)d.w; p.C = (int)d.c_in;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out;
    const int S = d.stride_h;
    const int TH = S == 2 ? 7 : 14, TW = TH;
    p.tiles_h = (int)ceil_div(p.Ho, TH);
    p.tiles_w = (int)ceil_div(p.Wo, TW);
    p.cgroups = p.C / 32;
    p.pad = d.pad_h;
    const int64_t ntiles = (int64_t)p.N * p.tiles_h * p.tiles_w;
    if (ntiles * p.cgroups > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "dw3x3: grid too large");
    p.ntiles = (uint32_t)ntiles;
    p.nblocks = (uint32_t)(ntiles * p.cgroups);
    p.ka = d.ka; p.kw = d.kw_scale;
    {   // SLFP_NT_DW bit 1 enables the policy; SLFP_DW_NT_MIN_MB moves the threshold (experiment switch; 0 = always nt)
        const int64_t min_mb = switches().dw_nt_min_mb;
        p.nt_out = ((SLFP_NT_DW & 2) && (int64_t)p.N * p.Ho * p.Wo * p.C * 4 >= (min_mb << 20)) ? 1 : 0;
    }
    p.enc = *act_table(d.ka, plan.fmt_act, kEncF32);
#define SLFP_DW2(SS, TT) \
    do { if (post.scale) hipLaunchKernelGGL((k_dw3x3_tile<SS, TT, TT, true>), dim3(p.nblocks), dim3(kDw2Threads), 0, stream, x, wq9c, y, p); \
         else hipLaunchKernelGGL((k_dw3x3_tile<SS, TT, TT, false>), dim3(p.nblocks), dim3(kDw2Threads), 0, stream, x, wq9c, y, p); } while (0)
    if (S == 2) SLFP_DW2(2, 7);
    else SLFP_DW2(1, 14);
#undef SLFP_DW2
    return check_launch("slfp dw3x3 (tile) kernel");
}

}  // namespace slfp
