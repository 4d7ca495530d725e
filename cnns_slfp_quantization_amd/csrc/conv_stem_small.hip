// conv_stem_small.hip -- 3x3-class image stems on the matrix cores: VGG-16 3x3 s1 3->64,
// ShuffleNetV2 3x3 s1 3->24 and other small-K first layers (MobileNetV1's own 3x3 s2 3->32 stem keeps its
// specialised fp32 kernel: same speed, exact).
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) for groups == 1, C_in <= 4 and
// K = KH*KW*C_in <= 32: the whole contraction is ONE 32-deep MFMA k-step.  The fp32 stem kernel
// (k_stem, conv_direct.hip) needs ~150 VALU instructions per output float4 and is VALU-issue-bound at
// ~3.7 TB/s (profiles/r01e); here the 27 multiply-adds per output value cost 1/16 of an MFMA and
// the VALU only encodes the input once per element and rescales the output:
//   * workgroup = 8 x 32 output pixels x all output channels, 4 waves; the input halo tile is read with
//     coalesced dword loads, x/Ka + SLFP encode applied once, and parked in LDS as fp16 rows
//     [ih][iw*C + c] (the K = (kh, kw, c) elements of one tap row are contiguous);
//   * a wave's unit = 16 consecutive output pixels of a row: lane (pixel, k-quarter) gathers its 8
//     k-values with ds_read_u16 from per-lane offsets computed once (k >= K reads a zero slot),
//     multiplies with the register-resident W fragments (C_out/16 MFMAs) and stores 16 bytes per tile;
//   * fp16 operands (x16 pre-scale as conv_pw.hip), float32 accumulation, the reference's
//     (out*Ka)*Kw roundings, bias and the optional fused BN/ReLU post-op in the epilogue.
// Precision: single-pass fp16 (SLFP<3,4>) / exact (SFP<3,3>); the float32-equivalent mode keeps k_stem.
#include "slfp_device.hpp"
#include "slfp_codes.hpp"
#include "slfp_host.hpp"

namespace slfp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kSsThreads = 256;
constexpr int kSsTH = 8, kSsTW = 32;   // output tile

struct StemSmallParams {
    const float* x;
    const _Float16* w;   // [NT][64 lanes][8]: A fragments, k = (kh*KW + kw)*C + c
    const float* bias;
    float* y;
    int N, H, W, C, O, KH, KW, S, ph, pw, Ho, Wo;
    int tiles_h, tiles_w;
    int IH, IWC, row_h;  // halo tile: IH rows of IWC = ((TW-1)*S + KW)*C elements, row pitch row_h halfs
    int K;               // KH*KW*C <= 32
    ScaleDiv sd;
    float s1, s2, s1x;
    PostOp post;
    uint32_t nblocks;
    // 1-byte codes out (slfp_codes.hpp; NT == 4, C_out == 64: VGG-16's first layer): the consumer's quantizer in the epilogue
    uint8_t* yc;
    int y_sgn, y_fmt;
    uint32_t enc_off;     // byte offset of the kEncCode table inside the dynamic LDS (behind the halo tile)
    EncArgs enc_out;
};

template <int FMT, int NT>
__global__ __launch_bounds__(kSsThreads) void k_stem_small(const StemSmallParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sT = reinterpret_cast<uint32_t*>(smem);
    _Float16* tile = reinterpret_cast<_Float16*>(smem + 64);   // [IH][row_h] + one zero slot at the end
    lut_fill<FMT>(sT);
    if (p.yc) enc_fill<kSsThreads>(reinterpret_cast<uint2*>(smem + p.enc_off), p.enc_out);   // published by the barrier below

    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int h_in0 = th * kSsTH * p.S - p.ph, w_in0 = tw * kSsTW * p.S - p.pw;
    const int zero_slot = p.IH * p.row_h;

    // W fragments: registers for the whole kernel
    half8 wf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const half8*>(p.w + ((size_t)j * 64 + lane) * 8);

    // this lane's 8 k-values -> halo-tile offsets relative to the pixel's window origin
    int koff[8];
    const int rl = p.KW * p.C;   // elements per tap row
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kq * 8 + j;
        const int kh = k / rl, r = k - kh * rl;
        koff[j] = k < p.K ? kh * p.row_h + r : -1;
    }
    __syncthreads();  // LUT visible

    // ---- halo tile: coalesced dword loads, encode once, fp16 to LDS
    {
        const float* xn = p.x + (size_t)n * p.H * p.W * p.C;
        const int n_el = p.IH * p.IWC;
        const int e_w0 = w_in0 * p.C;                 // element offset of the tile's first column inside an image row
        const int row_el = p.W * p.C;
        if (threadIdx.x == 0) tile[zero_slot] = (_Float16)0.f;
        constexpr int U = 8;
        for (int base = threadIdx.x; base < n_el; base += kSsThreads * U) {
            float v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * kSsThreads;
                const int ih = i / p.IWC, e = i - ih * p.IWC;
                const int gh = h_in0 + ih, ge = e_w0 + e;
                const bool live = i < n_el;
                const bool inb = live && (unsigned)gh < (unsigned)p.H && ge >= 0 && ge < row_el;
                dst[u] = live ? ih * p.row_h + e : -1;
                v[u] = xn[inb ? gh * row_el + ge : 0];   // unconditional (clamped) load
                if (!inb) v[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) tile[dst[u]] = (_Float16)quantize_scaled<FMT, 4>(v[u], p.sd, sT);
        }
    }
    __syncthreads();

    // this lane's channels are the same for every unit: bias / BN vectors once, not per store
    float4 bqv[NT];
    PostVec pvv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ch = j * 16 + kq * 4;
        bqv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        pvv[j] = PostVec{make_float4(1.f, 1.f, 1.f, 1.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        if (ch < p.O) {
            if (p.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + ch);
                bqv[j] = make_float4(256.f * ((bb.x / p.s1) / p.s2), 256.f * ((bb.y / p.s1) / p.s2),
                                     256.f * ((bb.z / p.s1) / p.s2), 256.f * ((bb.w / p.s1) / p.s2));
            }
            pvv[j] = post_load(p.post, ch);
        }
    }
    // ---- units: (output row of the tile, 16-pixel segment)
    for (int u = wave; u < kSsTH * (kSsTW / 16); u += kSsThreads / 64) {
        const int orow = u >> 1, seg = u & 1;
        const int goh = th * kSsTH + orow, gow = tw * kSsTW + seg * 16 + col;
        const int origin = (orow * p.S) * p.row_h + ((seg * 16 + col) * p.S) * p.C;
        half8 xf;
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = tile[koff[j] >= 0 ? origin + koff[j] : zero_slot];
        floatx4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if constexpr (NT == 4) {
            if (p.yc) {   // (wave-uniform) encode the 4 channel tiles, regroup to 16 consecutive channels per lane, one 16-byte store
                const unsigned char* senc = smem + p.enc_off;
                PostOp po = p.post;
                po.relu = 0;   // folded into the quantizer
                uint32_t c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 bq = bqv[j];
                    float4 r;
                    r.x = ((acc[j][0] + bq.x) * p.s1x) * p.s2;
                    r.y = ((acc[j][1] + bq.y) * p.s1x) * p.s2;
                    r.z = ((acc[j][2] + bq.z) * p.s1x) * p.s2;
                    r.w = ((acc[j][3] + bq.w) * p.s1x) * p.s2;
                    r = post_apply_v(r, po, pvv[j]);
                    if (p.y_sgn) c[j] = code_sign4(enc4_code<false>(r, p.enc_out.r1, p.enc_out.lo, p.enc_out.hi, senc), r, p.y_fmt);
                    else c[j] = enc4_code_relu(r, p.enc_out.r1, p.enc_out.lo, p.enc_out.hi, senc);
                }
                rows_transpose4(c[0], c[1], c[2], c[3]);
                if (goh < p.Ho && gow < p.Wo && kq * 16 < p.O) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<u32x4*>(p.yc + (((size_t)n * p.Ho + goh) * p.Wo + gow) * p.O + kq * 16) = u32x4{c[0], c[1], c[2], c[3]};
                }
                continue;
            }
        }
        if (goh < p.Ho && gow < p.Wo) {
            float* yp = p.y + (((size_t)n * p.Ho + goh) * p.Wo + gow) * p.O;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int ch = j * 16 + kq * 4;
                if (ch >= p.O) continue;
                const float4 bq = bqv[j];
                float4 r;
                r.x = ((acc[j][0] + bq.x) * p.s1x) * p.s2;
                r.y = ((acc[j][1] + bq.y) * p.s1x) * p.s2;
                r.z = ((acc[j][2] + bq.z) * p.s1x) * p.s2;
                r.w = ((acc[j][3] + bq.w) * p.s1x) * p.s2;
                st_stream4<SLFP_NT_STEM_MFMA>(yp + ch, post_apply_v(r, p.post, pvv[j]));
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------
bool stem_small_applicable(const slfp_conv2d_desc& d, int passes) {
    if (d.groups != 1 || d.c_in > 4 || d.dil_h != 1 || d.dil_w != 1 || d.stride_h != d.stride_w || d.stride_h > 2) return false;
    if (d.kh * d.kw <= 1 || d.kh * d.kw * d.c_in > 32) return false;
    if (d.c_out % 4 || d.c_out > 64) return false;
    if (d.qbits == 8 && passes == 3) return false;   // the float32-equivalent mode stays on k_stem / k_direct
    if ((int64_t)d.h * d.w * d.c_in >= (1ll << 30)) return false;
    // MobileNetV1's 3x3 s2 3->32 stem has a fully specialised fp32 kernel (k_stem_fixed) that measures the
    // same 0.153 ms at batch 256 (both are bound by the 411 MB output write): keep the exact one there
    if (d.c_in == 3 && d.kh == 3 && d.kw == 3 && d.stride_h == 2 && d.c_out == 32) return false;
    return true;
}

int stem_small_tiles(const slfp_conv2d_desc& d) { return (int)ceil_div(d.c_out, 16); }

template <int FMT>
static int launch_stem_small_f(const StemSmallParams& p, int nt, size_t lds, hipStream_t stream) {
    switch (nt) {
        case 1: hipLaunchKernelGGL((k_stem_small<FMT, 1>), dim3(p.nblocks), dim3(kSsThreads), lds, stream, p); break;
        case 2: hipLaunchKernelGGL((k_stem_small<FMT, 2>), dim3(p.nblocks), dim3(kSsThreads), lds, stream, p); break;
        case 3: hipLaunchKernelGGL((k_stem_small<FMT, 3>), dim3(p.nblocks), dim3(kSsThreads), lds, stream, p); break;
        default: hipLaunchKernelGGL((k_stem_small<FMT, 4>), dim3(p.nblocks), dim3(kSsThreads), lds, stream, p); break;
    }
    return check_launch("slfp small-K MFMA stem kernel");
}

// code output: the 64-channel form only (four channel tiles regroup to 16 consecutive codes per lane)
bool stem_small_codes_applicable(const slfp_conv2d_desc& d, const ConvPlan& plan, int post_flags) {
    return plan.family == kStemSmall && !plan.repad && d.c_out == 64 && !(post_flags & SLFP_POST_LAYEROUT);
}

int launch_stem_small(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const void* wblob,
                      const float* bias, const PostOp& post, float* y, hipStream_t stream) {
    const CodeIo io{false, false, 1.f, kFmtAct8};
    return launch_stem_small_io(d, plan, x, wblob, bias, post, y, io, stream);
}

int launch_stem_small_io(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const void* wblob,
                         const float* bias, const PostOp& post, void* y_any, const CodeIo& io, hipStream_t stream) {
    float* y = reinterpret_cast<float*>(y_any);
    StemSmallParams p;
    p.x = x; p.w = reinterpret_cast<const _Float16*>(wblob); p.bias = bias; p.y = y; p.post = post;
    p.yc = nullptr; p.y_sgn = 0; p.y_fmt = kFmtAct8; p.enc_off = 0; p.enc_out.valid = 0;
    p.N = (int)d.n; p.H = (int)d.h; p.W = (int)d.w; p.C = (int)d.c_in; p.O = (int)d.c_out;
    p.KH = (int)d.kh; p.KW = (int)d.kw; p.S = d.stride_h; p.ph = d.pad_h; p.pw = d.pad_w;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out;
    p.tiles_h = (int)ceil_div(p.Ho, kSsTH); p.tiles_w = (int)ceil_div(p.Wo, kSsTW);
    p.IH = (kSsTH - 1) * p.S + p.KH;
    p.IWC = ((kSsTW - 1) * p.S + p.KW) * p.C;
    p.row_h = (p.IWC + 1) | 1;           // odd pitch (in halfs): consecutive tap rows start on different banks
    p.K = p.KH * p.KW * p.C;
    p.sd = make_scale_div(d.ka, 4);
    p.s1 = plan.s1; p.s2 = plan.s2; p.s1x = plan.s1 * (1.0f / 256.0f);
    const int64_t nblocks = (int64_t)p.N * p.tiles_h * p.tiles_w;
    if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "small stem: grid too large");
    p.nblocks = (uint32_t)nblocks;
    size_t lds = 64 + ((size_t)p.IH * p.row_h + 8) * sizeof(_Float16);
    const int nt = stem_small_tiles(d);
    if (io.y_codes) {
        const EncArgs* t = enc_table(io.y_ka, io.y_fmt, kEncCode);
        if (!t->valid || nt != 4 || d.c_out != 64) return fail(SLFP_ERR_UNSUPPORTED, "small stem: no code output for this layer");
        p.enc_out = *t;
        p.yc = reinterpret_cast<uint8_t*>(y_any);
        p.y = nullptr;
        p.y_sgn = post.relu ? 0 : 1;
        p.y_fmt = io.y_fmt;
        p.enc_off = (uint32_t)((lds + 15) & ~(size_t)15);
        lds = p.enc_off + (size_t)kEncEntries * 8;
    }
    if (lds > 64 * 1024) return fail(SLFP_ERR_UNSUPPORTED, "small stem: halo tile needs %zu B of LDS", lds);
    return plan.fmt_act == kFmtAct8 ? launch_stem_small_f<kFmtAct8>(p, nt, lds, stream)
                                    : launch_stem_small_f<kFmtSfp7>(p, nt, lds, stream);
}

}  // namespace slfp
