// conv_dwc.hip -- SLFP-quantized depthwise 3x3 convolution on 1-byte activation codes, NHWC, gfx950.
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25) for groups == C_in == C_out, 3x3, stride 1 / 2 (the 13 "dw"
// layers of MobileNetV1, nets_imgnet/mobilenetv1.py:27) when the layer sits INSIDE a chain of quantized convolutions
// (csrc/slfp_codes.hpp): its input arrives as the codes the previous layer's epilogue wrote -- input_q = QA(x / Ka) is
// already applied, utils/conv2d_func.py:21 -- and its output leaves as the codes of the NEXT layer's QA(. / Ka_next)
// (or as float32 when it is the last of the chain).  Same arithmetic as conv_dw2.hip step for step: float32 FMAs over
// the decoded values in (kh, kw) order from +0, (acc * Ka) * Kw, fused BatchNorm + ReLU; outputs are bit-identical to the
// float32-interface kernel fed with the decoded tensor.
//
// Structure: no LDS tile, no barrier after the prologue.  At 1 B per element a pixel's channels are too few bytes for the
// (pixel slot, 32-channel group) lanes of conv_dw2.hip (32-byte pieces), so here a lane owns 4 channels of ONE output
// column and walks down TH output rows with a 3-row x 3-column window of decoded values in registers:
//   * lanes of a wave = (consecutive output columns) x (up to 128 consecutive channels): every load / store instruction
//     of a wave covers whole 128-byte lines (C >= 128) or one contiguous 256-byte run (C = 32, 64);
//   * all (TH-1)*S+3 rows x 3 columns of a lane's input codes (27 dwords) are requested back to back through a buffer
//     descriptor (out-of-image taps: out-of-range offset, the returned 0 is replaced by the exact-zero code 0x01), and
//     consumed row by row under counted vmcnt waits while the later rows are still in flight;
//   * decode = 1 VALU + 1 LDS lookup per element (256-entry table), encode = 5 VALU + 1 LDS (slfp_codes.hpp).
// The 3 columns of a lane overlap its neighbours' (they hit the same lines in L1); the redundancy is in decode
// instructions, not in HBM bytes.
#include "slfp_device.hpp"
#include "slfp_enc.hpp"
#include "slfp_codes.hpp"
#include "slfp_host.hpp"

namespace slfp {

constexpr int kDwcThreads = 256;
constexpr uint32_t kDwcOob = 0xFFFFFFF0u;

struct DwcParams {
    int N, H, W, C, Ho, Wo, pad;
    int cgs;              // channel groups of LC * 4 channels
    int row_tiles;        // ceil(Ho / TH)
    int col_tiles;        // ceil(Wo / TW)
    uint32_t ntasks;      // N * row_tiles * col_tiles * cgs
    uint32_t nblocks;
    int fmt_in;           // kFmtAct8 | kFmtSfp7: what the input codes are
    int fmt_out;          // the same for the output codes (YC)
    float ka, kw;
    int relu;
    const float* post_scale;
    const float* post_shift;
    EncArgs enc;          // YC: code table of the consumer's Ka (kEncCode)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// S: stride; LCS: log2(lanes across channels) = 3 / 4 / 5 for C = 32 / 64 / >= 128; TH x TW: output rows x columns per lane;
// YC: output as codes (else float32); POST: fused per-channel scale / shift; SIGNED: no ReLU before the output quantizer.
// VALU-issue-bound (profiles/r03c*): what is counted is instructions per output element.  A lane's TW adjacent columns
// share TW + 2 decoded input columns; row / column validity is one select each on the row and the column part of the
// offset (the parts are >= 2^30 when invalid, so their sum is out of the descriptor's range: tensors are < 1 GiB).
template <int S, int LCS, int TH, int TW, bool YC, bool POST, bool SIGNED>
__global__ __launch_bounds__(kDwcThreads) void k_dwc(const uint8_t* __restrict__ x, const float* __restrict__ wq,
                                                     void* __restrict__ y, const DwcParams p) {
    constexpr int LC = 1 << LCS;
    constexpr int NR = (TH - 1) * S + 3, NC = (TW - 1) * S + 3;
    __shared__ __attribute__((aligned(16))) unsigned char senc[YC ? ((kEncEntries * 8 + 15) & ~15) : 16];
    __shared__ __attribute__((aligned(16))) uint32_t sdec[256];
    if constexpr (YC) enc_fill<kDwcThreads>(reinterpret_cast<uint2*>(senc), p.enc);
    if (p.fmt_in == kFmtSfp7) dec_fill<kFmtSfp7, kDecF32, kDwcThreads>(sdec);
    else dec_fill<kFmtAct8, kDecF32, kDwcThreads>(sdec);
    const unsigned char* dtab = reinterpret_cast<const unsigned char*>(sdec);

    const uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    uint32_t t = b * (kDwcThreads / LC) + (threadIdx.x >> LCS);
    const bool valid = t < p.ntasks;
    t = valid ? t : p.ntasks - 1;
    const int cg = (int)(t % (uint32_t)p.cgs); t /= (uint32_t)p.cgs;
    const int ow0 = (int)(t % (uint32_t)p.col_tiles) * TW; t /= (uint32_t)p.col_tiles;
    const int rt = (int)(t % (uint32_t)p.row_tiles);
    const int n = (int)(t / (uint32_t)p.row_tiles);
    const int c = (cg * LC + (threadIdx.x & (LC - 1))) * 4;

    // ---- every input dword of this lane, back to back
    uint32_t raw[NR][NC];
    {
        const uint32_t in_bytes = (uint32_t)p.N * p.H * p.W * p.C;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(x), 0, in_bytes, 0x00020000);
        const int ih0 = rt * TH * S - p.pad, iw0 = ow0 * S - p.pad;
        const uint32_t rowb = (uint32_t)(n * p.H + ih0) * (uint32_t)(p.W * p.C) + (uint32_t)c;   // wraps for negative rows: only used in range
        const uint32_t rstep = (uint32_t)(p.W * p.C);
        uint32_t roff[NR], coff[NC];
        bool rok[NR], cok[NC];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            rok[j] = valid && (unsigned)(ih0 + j) < (unsigned)p.H;
            roff[j] = rok[j] ? rowb + (uint32_t)j * rstep : 0x80000000u;
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            cok[k] = (unsigned)(iw0 + k) < (unsigned)p.W;
            coff[k] = cok[k] ? (uint32_t)((iw0 + k) * p.C) : 0x40000000u;
        }
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int k = 0; k < NC; ++k) raw[j][k] = __builtin_amdgcn_raw_buffer_load_b32(rs, roff[j] + coff[k], 0, 0);
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int k = 0; k < NC; ++k) raw[j][k] = (rok[j] && cok[k]) ? raw[j][k] : 0x01010101u;   // zero padding = the exact-zero code
    }

    // this lane's 4 channels x 9 taps, fused BN vectors
    f32x4 wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const f32x4*>(wq + (uint32_t)(k * p.C + c));
    f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (POST) {
        psc = *reinterpret_cast<const f32x4*>(p.post_scale + c);
        psh = *reinterpret_cast<const f32x4*>(p.post_shift + c);
    }
    __syncthreads();   // tables visible

    const uint32_t out_bytes = (uint32_t)p.N * p.Ho * p.Wo * p.C * (YC ? 1u : 4u);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, out_bytes, 0x00020000);
    const int oh0 = rt * TH;
    const uint32_t ostep = (uint32_t)(p.Wo * p.C) * (YC ? 1u : 4u);
    uint32_t ocol[TW];
#pragma unroll
    for (int q = 0; q < TW; ++q)
        ocol[q] = (valid && ow0 + q < p.Wo) ? (((uint32_t)(n * p.Ho + oh0) * (uint32_t)p.Wo + (uint32_t)(ow0 + q)) * (uint32_t)p.C + (uint32_t)c) * (YC ? 1u : 4u)
                                            : 0x80000000u;
    const float r1 = p.enc.r1, lo = p.enc.lo, hi = p.enc.hi;

    float4 win[3][NC];   // ring of decoded input rows: row j lives in slot j % 3
#pragma unroll
    for (int j = 0; j < 3 - S; ++j)
#pragma unroll
        for (int k = 0; k < NC; ++k) win[j][k] = dec4_f32(raw[j][k], dtab);
#pragma unroll
    for (int r = 0; r < TH; ++r) {
#pragma unroll
        for (int j = r * S + 3 - S; j < r * S + 3; ++j)   // the S rows this step adds
#pragma unroll
            for (int k = 0; k < NC; ++k) win[j % 3][k] = dec4_f32(raw[j][k], dtab);
        const uint32_t orow = (oh0 + r) < p.Ho ? (uint32_t)r * ostep : 0x40000000u;
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float4 a = win[(r * S + kh) % 3][q * S + kw];
                    const f32x4 w = wt[kh * 3 + kw];
                    acc[0] = fmaf(a.x, w[0], acc[0]); acc[1] = fmaf(a.y, w[1], acc[1]);
                    acc[2] = fmaf(a.z, w[2], acc[2]); acc[3] = fmaf(a.w, w[3], acc[3]);
                }
            }
            f32x4 rr;   // (out * Ka) * Kw: two float32 roundings, as utils/conv2d_func.py:24; then the fused BN / ReLU
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = (acc[e] * p.ka) * p.kw;
                if constexpr (POST) u = __builtin_fmaf(u, psc[e], psh[e]);
                rr[e] = u;
            }
            const uint32_t so = ocol[q] + orow;
            if constexpr (YC) {
                uint32_t code;
                if constexpr (SIGNED) {
                    code = enc4_code<true>(make_float4(rr[0], rr[1], rr[2], rr[3]), r1, lo, hi, senc);
                    if (p.fmt_out == kFmtSfp7) code = (code & 0x3F3F3F3Fu) | ((code & 0x80808080u) >> 1);
                } else {
                    // the ReLU is the quantizer's: a negative value selects the lowest bin and fails its compare = the zero code
                    code = enc4_code_relu(make_float4(rr[0], rr[1], rr[2], rr[3]), r1, lo, hi, senc);
                }
                __builtin_amdgcn_raw_buffer_store_b32(code, ry, so, 0, 0);
            } else {
                if (p.relu) { rr[0] = fmaxf(rr[0], 0.f); rr[1] = fmaxf(rr[1], 0.f); rr[2] = fmaxf(rr[2], 0.f); rr[3] = fmaxf(rr[3], 0.f); }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, rr), ry, so, 0, 0);
            }
        }
    }
}

bool dwc_applicable(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* bias, int post_flags) {
    if (bias || plan.family != kDw3x3 || plan.repad) return false;
    if (d.c_in % 32 != 0 || (post_flags & SLFP_POST_LAYEROUT)) return false;
    if (d.c_in > 64 && d.c_in % 128 != 0) return false;
    if (d.pad_h > 2 || d.pad_h != d.pad_w) return false;
    const uint64_t in_b = (uint64_t)d.n * d.h * d.w * d.c_in, out_b = (uint64_t)d.n * plan.h_out * plan.w_out * d.c_in * 4;
    return in_b < (1ull << 30) && out_b < (1ull << 30);   // offsets: bit 30 / 31 mark an invalid column / row
}

// y_codes: output codes for a consumer with scale y_ka and format y_fmt (kFmtAct8 | kFmtSfp7); else float32
int launch_dwc(const slfp_conv2d_desc& d, const ConvPlan& plan, const uint8_t* x, const float* wq9c, const PostOp& post,
               void* y, bool y_codes, float y_ka, int y_fmt, hipStream_t stream) {
    DwcParams p;
    p.N = (int)d.n; p.H = (int)d.h; p.W = (int)d.w; p.C = (int)d.c_in;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out; p.pad = d.pad_h;
    const int S = d.stride_h;
    const int TH = S == 2 ? 4 : 7, TW = S == 2 ? 1 : 2;
    const int lcs = p.C >= 128 ? 5 : (p.C == 64 ? 4 : 3);
    p.cgs = p.C / (4 << lcs);
    p.row_tiles = (int)ceil_div(p.Ho, TH);
    p.col_tiles = (int)ceil_div(p.Wo, TW);
    const int64_t ntasks = (int64_t)p.N * p.row_tiles * p.col_tiles * p.cgs;
    const int tasks_per_block = kDwcThreads >> lcs;
    const int64_t nblocks = ceil_div(ntasks, tasks_per_block);
    if (ntasks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "dw3x3 (codes): grid too large");
    p.ntasks = (uint32_t)ntasks;
    p.nblocks = (uint32_t)nblocks;
    p.fmt_in = plan.fmt_act;
    p.fmt_out = y_fmt;
    p.ka = d.ka; p.kw = d.kw_scale;
    p.relu = post.relu;
    p.post_scale = post.scale; p.post_shift = post.shift;
    p.enc.valid = 0;
    if (y_codes) {
        const EncArgs* t = enc_table(y_ka, y_fmt, kEncCode);
        if (!t->valid) return fail(SLFP_ERR_UNSUPPORTED, "dw3x3 (codes): no code table for the consumer's scale %g", (double)y_ka);
        p.enc = *t;
    }
    const bool has_post = post.scale != nullptr;
    const bool sgn = !post.relu;
#define SLFP_DWC_L(SS, LL, TT, WW, YY, PP, GG) \
    hipLaunchKernelGGL((k_dwc<SS, LL, TT, WW, YY, PP, GG>), dim3(p.nblocks), dim3(kDwcThreads), 0, stream, x, wq9c, y, p)
#define SLFP_DWC_P(SS, LL, TT, WW) \
    do { if (y_codes) { if (has_post) { if (sgn) SLFP_DWC_L(SS, LL, TT, WW, true, true, true); else SLFP_DWC_L(SS, LL, TT, WW, true, true, false); } \
                        else { if (sgn) SLFP_DWC_L(SS, LL, TT, WW, true, false, true); else SLFP_DWC_L(SS, LL, TT, WW, true, false, false); } } \
         else { if (has_post) SLFP_DWC_L(SS, LL, TT, WW, false, true, false); else SLFP_DWC_L(SS, LL, TT, WW, false, false, false); } } while (0)
#define SLFP_DWC_S(SS, TT, WW) \
    do { if (lcs == 5) SLFP_DWC_P(SS, 5, TT, WW); else if (lcs == 4) SLFP_DWC_P(SS, 4, TT, WW); else SLFP_DWC_P(SS, 3, TT, WW); } while (0)
    if (S == 2) SLFP_DWC_S(2, 4, 1);
    else SLFP_DWC_S(1, 7, 2);
#undef SLFP_DWC_S
#undef SLFP_DWC_P
#undef SLFP_DWC_L
    return check_launch("slfp dw3x3 (codes) kernel");
}

}  // namespace slfp
