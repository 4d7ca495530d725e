// conv_dwpw.hip -- one MobileNet block in one kernel: depthwise 3x3 Conv2d_Q -> eval BatchNorm -> ReLU ->
// pointwise 1x1 Conv2d_Q (-> BatchNorm -> ReLU), the intermediate tensor never leaving the CU (gfx950).
//
// The reference runs nets_imgnet/mobilenetv1.py:24-41's conv_dw block as six modules: Conv2d_Q(3x3, groups = C),
// BatchNorm2d, ReLU, Conv2d_Q(1x1), BatchNorm2d, ReLU.  fusion.fuse_bn_relu folds each BatchNorm/ReLU into the
// preceding conv's epilogue; fusion.fuse_dw_pw then pairs the two convs here (second half): the
// depthwise result is quantized for the pointwise layer (QA(. / Ka_pw), utils/conv2d_func.py:21) right where it is
// produced and goes to the MFMA through LDS as fp16, instead of a float32 round trip through HBM
// (4 B written + 4 B read per element = 40 % of the block's traffic for 32->64 at 112x112).
//
// Arithmetic is EXACTLY the two separate kernels' (conv_dw2.hip, conv_pw.hip: k_pw_stream), step for step, so the
// output is bit-identical to running them back to back: input quantized through the threshold table, float32 FMAs in
// (kh, kw) order, (acc * Ka) * Kw, fma(scale, shift), max(0), threshold-table quantizer in its packed-fp16 form,
// v_mfma_f32_16x16x32_f16 over k in blob order, ((acc + 256 bq) * s1/256) * s2, fma, max.
//
//   workgroup (512 threads) = one image x one 14x14 (stride 1) or 7x7 (stride 2) tile of depthwise outputs x ALL channels:
//   phase 1, per 32-channel group: halo tile -> LDS (as conv_dw2.hip; the loads of the next group are issued before
//            this group is convolved), 3x3 conv, BN/ReLU, quantize -> fp16 row [pixel][K] in LDS;
//   phase 2: the tile's 13 (4) sixteen-pixel units x all output channels on the matrix cores, W resident in LDS
//            (K * N * 2 B <= 64 KiB: MobileNetV1's first four blocks, which carry 70 % of the net's conv traffic),
//            outputs staged to 128-byte pieces (the staging area reuses the halo tile).
//
// Measured (profiles/dwpw_bench.py, batch 256, cold): the kernel removes 40 % (32->64) to 33 % of the pair's HBM
// traffic but runs 1-2 workgroups per CU with its phases in series, so it only breaks even on 32@112->64
// (399 vs 407 us) and is slower on the K = 64 / 128 pairs (412 vs 352, 420 vs 337, 316 vs 204 us).  fusion.fuse_dw_pw is
// therefore opt-in; DESIGN.md lists what the next version needs (fp16 halo tile, k-outer accumulation, >= 3 workgroups/CU).
#include <cstdlib>
#include "slfp_device.hpp"
#include "slfp_enc.hpp"
#include "slfp_host.hpp"

namespace slfp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kFTab = (kEncEntries * 8 + 15) & ~15;
constexpr uint32_t kOobF = 0xFFFFFFF0u;

struct DwPwParams {
    const float* x;        // depthwise input, NHWC
    const float* wdw;      // [9][C] quantized depthwise weights (slfp_conv2d_prepare_weights of the dw layer)
    const float* sc1;      // BatchNorm 1 folded: scale / shift [C]
    const float* sh1;
    const _Float16* wpw;   // pointwise blob (hi plane), tile (nt, ks) at (nt * KSb + ks) * 512 halves
    const float* bias2;    // pointwise bias or nullptr
    const float* sc2;      // BatchNorm 2 folded or nullptr
    const float* sh2;
    float* y;              // pointwise output, NHWC
    int N, H, W, C, Ho, Wo, O;   // images, dw input size, channels, dw output size, pointwise channels
    int tiles_h, tiles_w, pad;
    int KSb;               // k-steps per tile in the pointwise blob
    int relu1, relu2;
    float ka1, kw1;        // depthwise scales
    float s1, s2, s1x;     // pointwise epilogue (Ka2, Kw2, Ka2 / 256)
    uint32_t nblocks;
    EncArgsCompact enc1;   // QA(x / Ka_dw) as float32
    EncArgsCompact enc2;   // fp16(16 * QA(. / Ka_pw)), packed pairs (two tables: compact form, 4 KiB of kernel arguments in all)
};

constexpr int kFT = 512;                         // threads: 64 pixel slots x 8 channel quads in phase 1, 8 waves in phase 2

// S: depthwise stride; KS: K / 32 (1, 2, 4); NT: N / 16 channel tiles (4, 8, 16)
template <int S, int KS, int NT>
__global__ __launch_bounds__(kFT, 2) void k_dwpw(const DwPwParams p) {
    constexpr int TH = S == 2 ? 7 : 14, TW = TH;
    constexpr int IH = (TH - 1) * S + 3, IW = IH;
    constexpr int RPL = kFT / 8 / 16;            // halo rows loaded per step (4)
    constexpr int NI = (IH + RPL - 1) / RPL;
    constexpr int ROWB = 16 * 32 * 4;
    constexpr int NPX = TH * TW;                 // depthwise outputs of the tile = pointwise pixels
    constexpr int NU = (NPX + 15) / 16;          // 16-pixel MFMA units (13 / 4)
    constexpr int NWV = kFT / 64;
    constexpr int UW = (NU + NWV - 1) / NWV;     // units per wave
    constexpr int K = KS * 32;
    constexpr int XROW = K * 2 + 16;             // bytes per pixel row of the fp16 image (+16: conflict-free fragment reads)
    constexpr int CW = TW > 8 ? 16 : 8, RPI = (kFT / 8) / CW, NO = (TH + RPI - 1) / RPI;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tab2 = smem + kFTab;
    unsigned char* tile = tab2 + kFTab;                        // [RPL * NI rows][16 slots][32 ch] float32
    unsigned char* x16 = tile + RPL * NI * ROWB;               // [NU * 16 rows][XROW]
    unsigned char* wl = x16 + NU * 16 * XROW;                  // [tile j][k-step][lane] 16 B
    float* ep = reinterpret_cast<float*>(wl + NT * KS * 1024);  // [3][NT * 16]
    unsigned char* stg_all = tile;   // [8 waves][2 KiB]: phase 2 reuses the (then idle) halo tile

    enc_fill_compact<kFT>(reinterpret_cast<uint2*>(smem), p.enc1);
    enc_fill_compact<kFT>(reinterpret_cast<uint2*>(tab2), p.enc2);
    for (int i = threadIdx.x; i < NT * KS * 64; i += kFT) {     // W -> LDS, 16 bytes per thread
        const int l = i & 63, t = i >> 6, ks = t % KS, j = t / KS;
        *reinterpret_cast<u32x4*>(wl + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(p.wpw + ((size_t)j * p.KSb + ks) * 512 + l * 8);
    }
    for (int i = threadIdx.x; i < NT * 16; i += kFT) {
        ep[i] = p.bias2 ? 256.f * ((p.bias2[i] / p.s1) / p.s2) : 0.f;
        ep[NT * 16 + i] = p.sc2 ? p.sc2[i] : 1.f;
        ep[2 * NT * 16 + i] = p.sc2 ? p.sh2[i] : 0.f;
    }
    for (int i = threadIdx.x; i < (NU * 16 - NPX) * (XROW / 8); i += kFT)   // rows beyond the tile's pixels: zeros
        reinterpret_cast<uint2*>(x16 + NPX * XROW)[i] = make_uint2(0u, 0u);

    const uint32_t lb = xcd_remap(blockIdx.x, p.nblocks);
    const uint32_t tiles_per_img = (uint32_t)(p.tiles_h * p.tiles_w);
    const uint32_t n = lb / tiles_per_img, tr = lb - n * tiles_per_img;
    const int th = (int)(tr / (uint32_t)p.tiles_w), tw = (int)(tr - (uint32_t)th * p.tiles_w);

    const int c4 = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const int iw = slot & 15, ihh = slot >> 4;
    const int colslot = (S == 2) ? ((iw & 1) * 8 + (iw >> 1)) : iw;
    const uint32_t lds_w = (uint32_t)((ihh * 16 + colslot) * 128 + c4 * 16);
    const bool col_live = iw < IW;
    const int ow = slot & (CW - 1), ohb = slot / CW;
    const bool ocol_live = ow < TW;
    const uint32_t lds_r0 = (uint32_t)(((ohb * S) * 16 + ow) * 128 + c4 * 16);
    const float r1 = p.enc1.r1, lo1 = p.enc1.lo, hi1 = p.enc1.hi;
    const float r2 = p.enc2.r1, lo2 = p.enc2.lo, hi2 = p.enc2.hi;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (size_t)n * p.H * p.W * p.C, 0,
                                                                         (uint32_t)p.H * p.W * p.C * 4u, 0x00020000);
    const int gw = tw * TW * S - p.pad + iw;
    const int gh0 = th * TH * S - p.pad + ihh;
    const bool w_ok = col_live && (unsigned)gw < (unsigned)p.W;
    const uint32_t off0 = (uint32_t)((gh0 * p.W + gw) * p.C + c4 * 4) * 4u;
    const uint32_t step = (uint32_t)(RPL * p.W * p.C) * 4u;
    uint32_t voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const bool ok = w_ok && (unsigned)(gh0 + RPL * i) < (unsigned)p.H && (RPL * i + ihh) < IH;
        voff[i] = ok ? off0 + (uint32_t)i * step : kOobF;
        asm volatile("" : "+v"(voff[i]));
    }
    f32x4 v[NI];
    auto issue = [&](int cg) {   // halo tile of channel group cg: soffset moves along the channel axis
#pragma unroll
        for (int i = 0; i < NI; ++i)
            v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], (uint32_t)cg * 128u, 0));
    };

    // ================= phase 1: depthwise, one 32-channel group at a time =================
    issue(0);
    __syncthreads();   // tables, W, zero rows visible
#pragma unroll 1
    for (int cg = 0; cg < KS; ++cg) {
        const int c = cg * 32 + c4 * 4;
        f32x4 wt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const f32x4*>(p.wdw + (size_t)k * p.C + c);
        const f32x4 psc = *reinterpret_cast<const f32x4*>(p.sc1 + c);
        const f32x4 psh = *reinterpret_cast<const f32x4*>(p.sh1 + c);
        bool any_nan = false;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 xi = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
            any_nan |= enc_has_nan4(xi);
            *reinterpret_cast<float4*>(tile + lds_w + (uint32_t)i * (uint32_t)(RPL * ROWB)) = enc4_f32_raw(xi, r1, lo1, hi1, smem);
        }
        if (__builtin_expect(any_nan, 0)) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[i][e] != v[i][e]) reinterpret_cast<uint32_t*>(tile + lds_w + (uint32_t)i * (uint32_t)(RPL * ROWB))[e] = kBitsQNaN;
        }
        __syncthreads();
        if (cg + 1 < KS) issue(cg + 1);   // in flight under the convolution below
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
            float4 rr;   // conv_dw2.hip's epilogue, then conv_pw.hip's quantizer
            {
                float u0 = (acc[0] * p.ka1) * p.kw1, u1 = (acc[1] * p.ka1) * p.kw1, u2 = (acc[2] * p.ka1) * p.kw1, u3 = (acc[3] * p.ka1) * p.kw1;
                u0 = __builtin_fmaf(u0, psc[0], psh[0]); u1 = __builtin_fmaf(u1, psc[1], psh[1]);
                u2 = __builtin_fmaf(u2, psc[2], psh[2]); u3 = __builtin_fmaf(u3, psc[3], psh[3]);
                if (p.relu1) { u0 = fmaxf(u0, 0.f); u1 = fmaxf(u1, 0.f); u2 = fmaxf(u2, 0.f); u3 = fmaxf(u3, 0.f); }
                rr = make_float4(u0, u1, u2, u3);
            }
            const uint2 q = enc4_f16(rr, r2, lo2, hi2, tab2);
            const int oh = ohb + j * RPI;
            if (ocol_live && oh < TH)
                *reinterpret_cast<uint2*>(x16 + (uint32_t)(oh * TW + ow) * XROW + (uint32_t)(cg * 64 + c4 * 8)) = q;
        }
        __syncthreads();   // the tile is free for the next group; after the last group: the fp16 image is complete
    }

    // ================= phase 2: pointwise on the matrix cores =================
    const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
    const int wv = threadIdx.x >> 6;
    half8 xh[UW][KS];
#pragma unroll
    for (int ui = 0; ui < UW; ++ui) {
        const int u = wv + NWV * ui;
        const unsigned char* row = x16 + (uint32_t)((u < NU ? u : NU - 1) * 16 + col) * XROW + kq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x2 a = *reinterpret_cast<const u32x2*>(row + ks * 64);
            const u32x2 b = *reinterpret_cast<const u32x2*>(row + ks * 64 + 32);
            xh[ui][ks] = __builtin_bit_cast(half8, u32x4{a[0], a[1], b[0], b[1]});
        }
    }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)n * p.Ho * p.Wo * p.O, 0,
                                                                         (uint32_t)p.Ho * p.Wo * p.O * 4u, 0x00020000);
    unsigned char* stg = stg_all + wv * 2048;
    const int spx = lane >> 3, sch = lane & 7;
    const bool has_vec = p.bias2 != nullptr || p.sc2 != nullptr;
    const int oh0 = th * TH, ow0 = tw * TW;
#pragma unroll 1
    for (int j0 = 0; j0 < NT; j0 += 2) {
#pragma unroll
        for (int ui = 0; ui < UW; ++ui) {
            const int u = wv + NWV * ui;
            if (u < NU) {   // wave-uniform
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = j0 + jj;
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const half8 wf = *reinterpret_cast<const half8*>(wl + (uint32_t)((j * KS + ks) * 1024) + lane * 16);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xh[ui][ks], acc, 0, 0, 0);
                    }
                    const int nl = j * 16 + kq * 4;
                    float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    if (has_vec) {
                        const float4 bq = *reinterpret_cast<const float4*>(ep + nl);
                        r.x += bq.x; r.y += bq.y; r.z += bq.z; r.w += bq.w;
                    }
                    r.x = (r.x * p.s1x) * p.s2; r.y = (r.y * p.s1x) * p.s2; r.z = (r.z * p.s1x) * p.s2; r.w = (r.w * p.s1x) * p.s2;
                    if (p.sc2) {
                        const float4 sc = *reinterpret_cast<const float4*>(ep + NT * 16 + nl);
                        const float4 sh = *reinterpret_cast<const float4*>(ep + 2 * NT * 16 + nl);
                        r.x = __builtin_fmaf(r.x, sc.x, sh.x); r.y = __builtin_fmaf(r.y, sc.y, sh.y);
                        r.z = __builtin_fmaf(r.z, sc.z, sh.z); r.w = __builtin_fmaf(r.w, sc.w, sh.w);
                    }
                    if (p.relu2) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                    *reinterpret_cast<float4*>(stg + col * 128 + jj * 64 + kq * 16) = r;
                }
                // LDS operations of one wave execute in order: reads see the writes, the next writes follow the reads
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 vv = *reinterpret_cast<const u32x4*>(stg + lane * 16 + h * 1024);
                    const int pix = u * 16 + spx + 8 * h;                      // pixel of the tile
                    const int oh = pix / TW, owp = pix - oh * TW;
                    const bool live = pix < NPX && (oh0 + oh) < p.Ho && (ow0 + owp) < p.Wo;
                    uint32_t so = live ? (uint32_t)(((oh0 + oh) * p.Wo + ow0 + owp) * p.O + j0 * 16 + sch * 4) * 4u : kOobF;
                    asm volatile("" : "+v"(so));
                    __builtin_amdgcn_raw_buffer_store_b128(vv, ry, so, 0, 0);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------- host
static bool dwpw_shape(const slfp_conv2d_desc& dw, const slfp_conv2d_desc& pw, int* ks, int* nt) {
    if (dw.groups != dw.c_in || dw.c_out != dw.c_in || dw.kh != 3 || dw.kw != 3 || dw.dil_h != 1 || dw.dil_w != 1) return false;
    if (dw.stride_h != dw.stride_w || (dw.stride_h != 1 && dw.stride_h != 2) || dw.pad_h != dw.pad_w || dw.pad_h > 2) return false;
    if (pw.kh != 1 || pw.kw != 1 || pw.groups != 1 || pw.stride_h != 1 || pw.stride_w != 1 || pw.pad_h || pw.pad_w) return false;
    if (pw.c_in != dw.c_out || dw.qbits != pw.qbits) return false;
    if (dw.x_layout != SLFP_LAYOUT_NHWC || pw.y_layout != SLFP_LAYOUT_NHWC) return false;
    const int K = (int)dw.c_in, N = (int)pw.c_out;
    if (K != 32 && K != 64 && K != 128) return false;
    if (N != 64 && N != 128 && N != 256) return false;
    if ((size_t)K * N * 2 > 64 * 1024) return false;
    if (dw.stride_h == 1 && (dw.h < 14 || dw.w < 14)) return false;
    *ks = K / 32; *nt = N / 16;
    return true;
}

static size_t dwpw_lds(int S, int KS, int NT) {
    const int TH = S == 2 ? 7 : 14, IH = (TH - 1) * S + 3, NI = (IH + 3) / 4, NU = (TH * TH + 15) / 16;
    return (size_t)2 * kFTab + (size_t)4 * NI * 16 * 32 * 4 + (size_t)NU * 16 * (KS * 64 + 16) + (size_t)NT * KS * 1024 +
           (size_t)3 * NT * 16 * 4;
}

}  // namespace slfp

using namespace slfp;

extern "C" int slfp_dwpw_supported(const slfp_conv2d_desc* dw, const slfp_conv2d_desc* pw) {
    if (!dw || !pw || long_encode_forced()) return 0;
    ConvPlan p1, p2;
    if (make_plan(dw, &p1) != SLFP_OK || make_plan(pw, &p2) != SLFP_OK) return 0;
    if (p1.family != kDw3x3 || p2.family != kPointwise || p1.repad || p2.repad || p2.passes != 1) return 0;
    if (pw->n != dw->n || pw->h != p1.h_out || pw->w != p1.w_out) return 0;
    int ks, nt;
    if (!dwpw_shape(*dw, *pw, &ks, &nt)) return 0;
    if (dwpw_lds(dw->stride_h, ks, nt) > 160 * 1024) return 0;
    if (!act_table(dw->ka, p1.fmt_act, kEncF32) || !act_table(pw->ka, p2.fmt_act, kEncF16P)) return 0;
    return 1;
}

extern "C" int slfp_dwpw_fwd(const slfp_conv2d_desc* dw, const slfp_conv2d_desc* pw, const float* x, const void* wprep_dw,
                             const float* post1_scale, const float* post1_shift, int relu1, const void* wprep_pw,
                             const float* bias_pw, const float* post2_scale, const float* post2_shift, int relu2, float* y,
                             void* stream) {
    if (!slfp_dwpw_supported(dw, pw)) return fail(SLFP_ERR_UNSUPPORTED, "slfp_dwpw_fwd: this pair of layers is not fusable (slfp_dwpw_supported)");
    if (!x || !wprep_dw || !wprep_pw || !y || !post1_scale || !post1_shift)
        return fail(SLFP_ERR_BAD_ARG, "slfp_dwpw_fwd: null pointer (the depthwise layer's folded BatchNorm is required)");
    if ((post2_scale == nullptr) != (post2_shift == nullptr)) return fail(SLFP_ERR_BAD_ARG, "slfp_dwpw_fwd: post2_scale and post2_shift go together");
    if (!aligned16(x) || !aligned16(y) || !aligned16(wprep_dw) || !aligned16(wprep_pw) || !aligned16(post1_scale) || !aligned16(post1_shift))
        return fail(SLFP_ERR_ALIGNMENT, "slfp_dwpw_fwd: pointers must be 16-byte aligned");
    ConvPlan p1, p2;
    make_plan(dw, &p1);
    make_plan(pw, &p2);
    int ks, nt;
    dwpw_shape(*dw, *pw, &ks, &nt);
    DwPwParams p;
    p.x = x; p.wdw = reinterpret_cast<const float*>(wprep_dw); p.sc1 = post1_scale; p.sh1 = post1_shift;
    p.wpw = reinterpret_cast<const _Float16*>(wprep_pw); p.bias2 = bias_pw; p.sc2 = post2_scale; p.sh2 = post2_shift; p.y = y;
    p.N = (int)dw->n; p.H = (int)dw->h; p.W = (int)dw->w; p.C = (int)dw->c_in; p.Ho = (int)p1.h_out; p.Wo = (int)p1.w_out; p.O = (int)pw->c_out;
    const int S = dw->stride_h, TH = S == 2 ? 7 : 14;
    p.tiles_h = (int)ceil_div(p.Ho, TH); p.tiles_w = (int)ceil_div(p.Wo, TH); p.pad = dw->pad_h;
    p.KSb = (int)(p2.k_pad / 32);
    p.relu1 = relu1 ? 1 : 0; p.relu2 = relu2 ? 1 : 0;
    p.ka1 = dw->ka; p.kw1 = dw->kw_scale;
    p.s1 = p2.s1; p.s2 = p2.s2; p.s1x = p2.s1 * (1.0f / 256.0f);
    const int64_t nblocks = (int64_t)p.N * p.tiles_h * p.tiles_w;
    if (nblocks > 0x7FFFFFFF || (int64_t)p.H * p.W * p.C >= (1ll << 29) || (int64_t)p.Ho * p.Wo * p.O >= (1ll << 29))
        return fail(SLFP_ERR_UNSUPPORTED, "slfp_dwpw_fwd: tensor too large");
    p.nblocks = (uint32_t)nblocks;
    p.enc1 = enc_compact(*act_table(dw->ka, p1.fmt_act, kEncF32));
    p.enc2 = enc_compact(*act_table(pw->ka, p2.fmt_act, kEncF16P));
    static_assert(sizeof(DwPwParams) <= 4096, "kernel arguments must fit the 4 KiB kernarg segment");
    const size_t lds = dwpw_lds(S, ks, nt);
    hipStream_t st = as_stream(stream);
#define SLFP_F(SS, KK, NN)                                                                                         \
    if (S == SS && ks == KK && nt == NN) {                                                                          \
        auto fn = k_dwpw<SS, KK, NN>;                                                                               \
        const int rc = raise_lds_limit(reinterpret_cast<const void*>(fn), lds);                                     \
        if (rc != SLFP_OK) return rc;                                                                               \
        hipLaunchKernelGGL(fn, dim3(p.nblocks), dim3(kFT), lds, st, p);                                             \
        return check_launch("slfp fused depthwise + pointwise kernel");                                             \
    }
    SLFP_F(1, 1, 4) SLFP_F(2, 2, 8) SLFP_F(1, 4, 8) SLFP_F(2, 4, 16)
    SLFP_F(1, 2, 8) SLFP_F(2, 1, 4) SLFP_F(1, 1, 8) SLFP_F(2, 4, 8) SLFP_F(1, 2, 4) SLFP_F(1, 4, 16) SLFP_F(2, 2, 4) SLFP_F(1, 4, 4) SLFP_F(2, 1, 8) SLFP_F(2, 2, 16) SLFP_F(1, 2, 16) SLFP_F(2, 4, 4)
#undef SLFP_F
    return fail(SLFP_ERR_UNSUPPORTED, "slfp_dwpw_fwd: no instantiation for this shape");
}
