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
#include "slfp_enc.hpp"
#include "slfp_codes.hpp"
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
    PostOp post;
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
        for (int r = 0; r < 4; ++r) r4[r] = r < n_o ? post_apply1(((acc[q][r] + bq[r]) * p.s1) * p.s2, p.post, o0 + r) : 0.f;
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


// ======================================================================================
// k_stem: dense convolution with a tiny C_in (the RGB stems: MobileNetV1 3x3 s2,
// ResNet-50 / SqueezeNet 7x7 s2, VGG 3x3 s1).  HBM-bound on the OUTPUT side (32..96
// channels written per 3 read), so the store pattern is what matters:
//   * one workgroup = one image, an 8 x 16 output tile, all C_out (<= 64, power-of-two / 4 lanes);
//   * input halo rows are contiguous in NHWC (IW*C floats): coalesced dword loads, encoded once;
//   * weights ([KH][KW][C][O], the direct-family blob) are copied to LDS once per workgroup;
//   * lane = (pixel column, 4 output channels): for a fixed output row the lanes of a wave
//     cover consecutive pixels x all channels = one contiguous run (1 KiB per store
//     instruction for C_out = 32); each thread walks P rows.
// ======================================================================================
constexpr int kStemTH = 8, kStemTW = 16;

struct StemParams {
    int N, H, W, C, O, KH, KW;
    int s, ph, pw;
    int Ho, Wo, tiles_h, tiles_w;
    int IH, IWC;         // input tile rows, floats per tile row (IW * C)
    int l4_shift;        // log2(O / 4): lanes per pixel
    int P;               // output rows per thread
    int step_h, step_j;  // 256 consecutive floats of the input tile = step_h rows + step_j floats
    ScaleDiv sd;
    float s1, s2;
    PostOp post;
    uint32_t nblocks;
    EncArgs enc;         // threshold table of QA(x / Ka) (k_stem_fixed<.., TAB>; slfp_enc.hpp)
    EncArgsCompact enc_out;  // YC: code table of the consumer's Ka (kEncCode; slfp_codes.hpp)
    int sgn, fmt_out;        // YC: codes carry a sign (no ReLU in front of the quantizer); their format
};

template <int FMT>
__global__ __launch_bounds__(256) void k_stem(const float* __restrict__ x, const float* __restrict__ wq,
                                              const float* __restrict__ bias, float* __restrict__ y, const StemParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sT = reinterpret_cast<uint32_t*>(smem);
    float* sW = reinterpret_cast<float*>(smem + 64);            // [KH*KW*C][O]
    const int n_w = p.KH * p.KW * p.C * p.O;
    float* tile = sW + ((n_w + 3) & ~3);                         // [IH][IWC]
    lut_fill<FMT>(sT);
    for (int i = threadIdx.x * 4; i < n_w; i += 256 * 4)        // O % 4 == 0 -> n_w % 4 == 0, 16-byte aligned blob
        *reinterpret_cast<float4*>(sW + i) = *reinterpret_cast<const float4*>(wq + i);

    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b;
    const int h_in0 = th * kStemTH * p.s - p.ph, w_in0 = tw * kStemTW * p.s - p.pw;
    __syncthreads();

    // ---- load + encode the input halo tile (rows are contiguous in NHWC)
    {
        const int n_in = p.IH * p.IWC;
        int idx = threadIdx.x;
        int ih = idx / p.IWC, j = idx - ih * p.IWC;  // once per thread
        const float* xn = x + (size_t)n * p.H * p.W * p.C;
        const int j_lo = -w_in0 * p.C, j_hi = (p.W - w_in0) * p.C;  // valid float range of a tile row
        constexpr int U = 4;
        while (idx < n_in) {
            float v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int gh = h_in0 + ih;
                dst[u] = idx < n_in ? ih * p.IWC + j : -1;
                v[u] = 0.f;
                if (idx < n_in && gh >= 0 && gh < p.H && j >= j_lo && j < j_hi)
                    v[u] = xn[((size_t)gh * p.W + w_in0) * p.C + j];
                idx += 256;
                ih += p.step_h;
                j += p.step_j;
                if (j >= p.IWC) { j -= p.IWC; ++ih; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) tile[dst[u]] = quantize_scaled<FMT>(v[u], p.sd, sT);
        }
    }
    __syncthreads();

    // ---- compute: lane = (column, 4 channels), P rows per thread
    const int l4 = 1 << p.l4_shift;
    const int c4 = threadIdx.x & (l4 - 1);
    const int grp = threadIdx.x >> p.l4_shift;
    const int col = grp & (kStemTW - 1);
    const int row0 = (grp >> 4) * p.P;
    constexpr int PMAX = 8;
    float4 acc[PMAX];
#pragma unroll
    for (int q = 0; q < PMAX; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int row_stride = p.s * p.IWC;
    const float* tbase = tile + (row0 * p.s) * p.IWC + col * p.s * p.C;
    const float* wbase = sW + c4 * 4;
    for (int kh = 0; kh < p.KH; ++kh) {
        for (int kwc = 0; kwc < p.KW * p.C; ++kwc) {  // (kw, ci) are adjacent both in the tile row and in the blob
            const float4 w = *reinterpret_cast<const float4*>(wbase + (size_t)(kh * p.KW * p.C + kwc) * p.O);
            const float* a = tbase + kh * p.IWC + kwc;
#pragma unroll
            for (int q = 0; q < PMAX; ++q) {
                if (q < p.P) {
                    const float av = a[q * row_stride];
                    acc[q].x = fmaf(av, w.x, acc[q].x);
                    acc[q].y = fmaf(av, w.y, acc[q].y);
                    acc[q].z = fmaf(av, w.z, acc[q].z);
                    acc[q].w = fmaf(av, w.w, acc[q].w);
                }
            }
        }
    }
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + c4 * 4);
        bq = make_float4((bb.x / p.s1) / p.s2, (bb.y / p.s1) / p.s2, (bb.z / p.s1) / p.s2, (bb.w / p.s1) / p.s2);
    }
    const int gow = tw * kStemTW + col;
    if (gow >= p.Wo) return;
    const PostVec pv = post_load(p.post, c4 * 4);  // once per thread, not per stored row
#pragma unroll
    for (int q = 0; q < PMAX; ++q) {
        const int goh = th * kStemTH + row0 + q;
        if (q < p.P && goh < p.Ho) {
            float4 r;
            r.x = ((acc[q].x + bq.x) * p.s1) * p.s2;
            r.y = ((acc[q].y + bq.y) * p.s1) * p.s2;
            r.z = ((acc[q].z + bq.z) * p.s1) * p.s2;
            r.w = ((acc[q].w + bq.w) * p.s1) * p.s2;
            *reinterpret_cast<float4*>(y + (((size_t)n * p.Ho + goh) * p.Wo + gow) * p.O + c4 * 4) = post_apply_v(r, p.post, pv);
        }
    }
}


// Fully specialised stem (compile-time KH, KW, C, stride, O): every LDS address is base +
// immediate, the 27 (or 147) taps unroll into ds_read + packed FMAs with no index arithmetic
// (the generic k_stem was VALU-bound: 1225 instructions per wave, profiles/r01c).
// TAB: threshold-table quantizer (slfp_enc.hpp) and branch-free halo loads: a buffer descriptor over the image makes
// out-of-image elements an out-of-range offset that reads 0, so the 7 loads of a thread issue back to back.
// TH: output rows per workgroup (16 for the table variant: the 2 KiB table + 3.4 KiB of weights a workgroup stages are
// then paid once per 32 KiB of output instead of once per 16 KiB).
// YC: the output leaves as the 1-byte codes of the next layer's QA(. / Ka) (slfp_codes.hpp): a lane's 4 channels are one dword.
template <int FMT, int KH, int KW, int C, int S, int O, bool TAB = false, int TH = kStemTH, bool YC = false>
__global__ __launch_bounds__(256) void k_stem_fixed(const float* __restrict__ x, const float* __restrict__ wq,
                                                    const float* __restrict__ bias, float* __restrict__ y,
                                                    const StemParams p) {
    constexpr int IH = (TH - 1) * S + KH, IWC = ((kStemTW - 1) * S + KW) * C;
    constexpr int NW = KH * KW * C * O;
    constexpr int L4 = O / 4, GROUPS = 256 / L4, P = TH / (GROUPS / kStemTW);
    static_assert(GROUPS % kStemTW == 0 && P >= 1 && NW % 4 == 0, "unsupported stem shape");
    __shared__ __attribute__((aligned(16))) float sW[NW];
    __shared__ __attribute__((aligned(16))) float tile[IH * IWC];
    __shared__ __attribute__((aligned(16))) uint32_t sT[TAB ? 2 * (kEncEntries + 1) : 16];
    __shared__ __attribute__((aligned(16))) unsigned char senc[YC ? ((kEncEntries * 8 + 15) & ~15) : 16];
    if constexpr (TAB) enc_fill<256>(reinterpret_cast<uint2*>(sT), p.enc);
    else lut_fill<FMT>(sT);
    if constexpr (YC) enc_fill_compact<256>(reinterpret_cast<uint2*>(senc), p.enc_out);
    for (int i = threadIdx.x * 4; i < NW; i += 256 * 4)
        *reinterpret_cast<float4*>(sW + i) = *reinterpret_cast<const float4*>(wq + i);

    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b;
    const int h_in0 = th * TH * S - p.ph, w_in0 = tw * kStemTW * S - p.pw;
    __syncthreads();

    {   // load + encode the halo tile: rows are contiguous in NHWC, one dword per lane
        constexpr int N_IN = IH * IWC, U = (N_IN + 255) / 256;
        const float* xn = x + (size_t)n * p.H * p.W * C;
        const int j_lo = -w_in0 * C, j_hi = (p.W - w_in0) * C;
        float v[U];
        if constexpr (TAB) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, (uint32_t)p.H * p.W * C * 4u, 0x00020000);
            uint32_t vo[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = threadIdx.x + u * 256;
                const int ih = idx / IWC, j = idx - ih * IWC;  // compile-time divisor
                const int gh = h_in0 + ih;
                const bool ok = idx < N_IN && (unsigned)gh < (unsigned)p.H && j >= j_lo && j < j_hi;
                vo[u] = ok ? (uint32_t)((gh * p.W + w_in0) * C + j) * 4u : 0xFFFFFFF0u;
                asm volatile("" : "+v"(vo[u]));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo[u], 0, 0));
            const unsigned char* tb = reinterpret_cast<const unsigned char*>(sT);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = threadIdx.x + u * 256;
                float q = enc_f32(v[u], p.enc.r1, p.enc.lo, p.enc.hi, tb);
                q = v[u] != v[u] ? __uint_as_float(kBitsQNaN) : q;   // NaN in -> NaN out
                if (idx < N_IN) tile[idx] = q;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = threadIdx.x + u * 256;
                const int ih = idx / IWC, j = idx - ih * IWC;  // compile-time divisor
                const int gh = h_in0 + ih;
                v[u] = 0.f;
                if (idx < N_IN && (unsigned)gh < (unsigned)p.H && j >= j_lo && j < j_hi)
                    v[u] = xn[(gh * p.W + w_in0) * C + j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = threadIdx.x + u * 256;
                if (idx < N_IN) tile[idx] = quantize_scaled<FMT>(v[u], p.sd, sT);
            }
        }
    }
    __syncthreads();

    const int c4 = threadIdx.x & (L4 - 1);
    const int grp = threadIdx.x / L4;
    const int col = grp & (kStemTW - 1);
    const int row0 = (grp / kStemTW) * P;
    float4 acc[P];
#pragma unroll
    for (int q = 0; q < P; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* tbase = tile + (row0 * S) * IWC + col * S * C;
    const float* wbase = sW + c4 * 4;
#pragma unroll 1  // one kernel row at a time: a full unroll makes hipcc hoist all 135 LDS reads (243 VGPRs)
    for (int kh = 0; kh < KH; ++kh) {
        const float* wrow = wbase + kh * KW * C * O;
        const float* trow = tbase + kh * IWC;
#pragma unroll
        for (int kwc = 0; kwc < KW * C; ++kwc) {
            const float4 w = *reinterpret_cast<const float4*>(wrow + kwc * O);
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const float av = trow[q * S * IWC + kwc];
                acc[q].x = fmaf(av, w.x, acc[q].x);
                acc[q].y = fmaf(av, w.y, acc[q].y);
                acc[q].z = fmaf(av, w.z, acc[q].z);
                acc[q].w = fmaf(av, w.w, acc[q].w);
            }
        }
    }
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + c4 * 4);
        bq = make_float4((bb.x / p.s1) / p.s2, (bb.y / p.s1) / p.s2, (bb.z / p.s1) / p.s2, (bb.w / p.s1) / p.s2);
    }
    const int gow = tw * kStemTW + col;
    if (gow >= p.Wo) return;
    if constexpr (YC) {
        uint8_t* yc = reinterpret_cast<uint8_t*>(y) + (((size_t)n * p.Ho + th * TH + row0) * p.Wo + gow) * O + c4 * 4;
        const PostVec pv = post_load(p.post, c4 * 4);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            if (th * TH + row0 + q < p.Ho) {
                float4 r;
                r.x = ((acc[q].x + bq.x) * p.s1) * p.s2;
                r.y = ((acc[q].y + bq.y) * p.s1) * p.s2;
                r.z = ((acc[q].z + bq.z) * p.s1) * p.s2;
                r.w = ((acc[q].w + bq.w) * p.s1) * p.s2;
                PostOp po = p.post;
                po.relu = 0;   // with code output the ReLU is the quantizer's (enc4_code_relu)
                r = post_apply_v(r, po, pv);
                uint32_t code = p.sgn ? enc4_code<false>(r, p.enc_out.r1, p.enc_out.lo, p.enc_out.hi, senc)
                                      : enc4_code_relu(r, p.enc_out.r1, p.enc_out.lo, p.enc_out.hi, senc);
                if (p.sgn) {
                    const float rs[4] = {r.x, r.y, r.z, r.w};
                    uint32_t out = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint32_t cc = (code >> (8 * k)) & 0xFFu;
                        cc = (cc == 1u || rs[k] != rs[k]) ? cc : (cc | ((__float_as_uint(rs[k]) >> 24) & 0x80u));
                        out |= cc << (8 * k);
                    }
                    code = p.fmt_out == kFmtSfp7 ? ((out & 0x3F3F3F3Fu) | ((out & 0x80808080u) >> 1)) : out;
                }
                *reinterpret_cast<uint32_t*>(yc + (uint32_t)(q * p.Wo * O)) = code;
            }
        }
        return;
    }
    float* yb = y + (((size_t)n * p.Ho + th * TH + row0) * p.Wo + gow) * O + c4 * 4;
    if (p.post.scale) {   // fused BN: the vectors once per thread (its own path: see conv_pw.hip, k_pw_tiled)
        const PostVec pv = post_load(p.post, c4 * 4);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            if (th * TH + row0 + q < p.Ho) {
                float4 r;
                r.x = ((acc[q].x + bq.x) * p.s1) * p.s2;
                r.y = ((acc[q].y + bq.y) * p.s1) * p.s2;
                r.z = ((acc[q].z + bq.z) * p.s1) * p.s2;
                r.w = ((acc[q].w + bq.w) * p.s1) * p.s2;
                st_stream4<SLFP_NT_STEM>(yb + (uint32_t)(q * p.Wo * O), post_apply_v(r, p.post, pv));
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < P; ++q) {
        if (th * TH + row0 + q < p.Ho) {
            float4 r;
            r.x = ((acc[q].x + bq.x) * p.s1) * p.s2;
            r.y = ((acc[q].y + bq.y) * p.s1) * p.s2;
            r.z = ((acc[q].z + bq.z) * p.s1) * p.s2;
            r.w = ((acc[q].w + bq.w) * p.s1) * p.s2;
            st_stream4<SLFP_NT_STEM>(yb + (uint32_t)(q * p.Wo * O), post_apply(r, p.post, c4 * 4));
        }
    }
}

// ======================================================================================
// k_stem_mx: the MobileNetV1 stem (3x3 s2, 3 -> 32, nets_imgnet/mobilenetv1.py:44) on the FLOAT32 matrix cores (round 3).
// k_stem_fixed spends 27 float32 FMAs per output element on the vector ALU and is VALU-issue-bound (33 VALU instructions
// per output, profiles/r03c0: the kernel takes the same 122 us whether it writes float32 or 4x smaller codes).
// v_mfma_f32_16x16x4_f32 multiplies float32 operands exactly and accumulates in float32 in k order, i.e. the same fused
// multiply-add chain (kh, kw, c) the vector kernel runs -- outputs are bit-identical (tests/test_gpu_parity.py) -- at 1024
// multiply-adds per instruction, on a pipe that runs beside the vector ALU:
//   A = W [16 output channels x 4 taps], B = input patches [4 taps x 16 pixels] from the quantized float32 halo tile in LDS
//   (one ds_read_b32 per k-step per lane), 7 k-steps (27 taps + one zero tap) x 2 channel tiles per 16 output pixels.
//   A's rows are channels in the order 8 (r / 4) + 4 t + r % 4 for tile t, so a lane ends up with 8 CONSECUTIVE channels
//   of its pixel: float32 output is staged through LDS into whole 1 KiB runs (8 pixels x 128 B per store instruction),
//   code output (YC) is one 8-byte store per lane (a wave: 512 contiguous bytes).
// One workgroup = one image, a 16 x 16 output tile; a wave owns 4 tile rows = 4 units of 16 pixels.
// ======================================================================================
template <bool YC>
__global__ __launch_bounds__(256) void k_stem_mx(const float* __restrict__ x, const float* __restrict__ wq,
                                                 const float* __restrict__ bias, float* __restrict__ y, const StemParams p) {
    constexpr int KH = 3, KW = 3, C = 3, S = 2, O = 32, TH = 16;
    constexpr int IH = (TH - 1) * S + KH, IWC = ((kStemTW - 1) * S + KW) * C;   // 33 rows x 99 floats
    constexpr int N_IN = IH * IWC, U = (N_IN + 255) / 256;
    typedef float f32x4m __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float tile[N_IN + 1];   // [IH][IWC] + one zero word (the 28th tap)
    __shared__ __attribute__((aligned(16))) uint32_t sT[2 * (kEncEntries + 1)];
    __shared__ __attribute__((aligned(16))) unsigned char senc[YC ? ((kEncEntries * 8 + 15) & ~15) : 16];
    constexpr int SP = 36;   // staging row pitch in floats: 128 B of channels + 16 B, so the 16 pixels' float4 writes spread over the banks
    __shared__ __attribute__((aligned(16))) float stg[YC ? 4 : 4 * 16 * SP];   // float32 output: one 16-pixel unit per wave
    enc_fill<256>(reinterpret_cast<uint2*>(sT), p.enc);
    if constexpr (YC) enc_fill_compact<256>(reinterpret_cast<uint2*>(senc), p.enc_out);
    if (threadIdx.x == 0) tile[N_IN] = 0.f;

    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b;
    const int h_in0 = th * TH * S - p.ph, w_in0 = tw * kStemTW * S - p.pw;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, kq = lane >> 4;
    // A fragments: W[k = 4 ks + kq][channel of row px of tile t], 7 k-steps x 2 tiles, once per workgroup
    float wa[7][2];
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const int k = 4 * ks + kq;
#pragma unroll
        for (int t = 0; t < 2; ++t) wa[ks][t] = k < KH * KW * C ? wq[k * O + 8 * (px >> 2) + 4 * t + (px & 3)] : 0.f;
    }
    // byte offset of tap k = 4 ks + kq inside a pixel's window: (k / 9) rows + k % 9 floats; tap 27 is the zero word
    uint32_t koff[7];
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const int k = 4 * ks + kq;
        koff[ks] = (uint32_t)((k / (KW * C)) * IWC + (k % (KW * C))) * 4u;
    }
    __syncthreads();   // tables visible

    {   // load + quantize the halo tile.  Rows are contiguous in NHWC (99 floats of a 672-float image row, starting 3 floats
        // before a multiple of 32 pixels): each lane loads 16 ALIGNED bytes of the 100-float superset that starts one float
        // earlier -- 4 loads per thread instead of 13 dword loads (the stem kernels took ~120 us whatever they computed or
        // stored: the vector memory pipe was busy with 650 k dword-load instructions per launch, profiles/r03c).
        // Needs W * C to be a multiple of 4 floats (the launcher checks): a float4 is then wholly inside or outside its row.
        constexpr int Q = (IWC + 1 + 3) / 4;            // float4 per tile row (25)
        constexpr int N4 = IH * Q, U4 = (N4 + 255) / 256;
        const float* xn = x + (size_t)n * p.H * p.W * C;
        const int rowf = p.W * C;                        // floats per image row
        const int s0 = w_in0 * C - 1;                    // superset start inside the image row (a multiple of 4, or -4)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, (uint32_t)p.H * p.W * C * 4u, 0x00020000);
        uint32_t vo[U4];
        typedef float f32x4l __attribute__((ext_vector_type(4)));
        f32x4l v[U4];
#pragma unroll
        for (int u = 0; u < U4; ++u) {
            const int idx = threadIdx.x + u * 256;
            const int ih = idx / Q, q = idx - ih * Q;    // compile-time divisor
            const int gh = h_in0 + ih, f0 = s0 + 4 * q;
            const bool ok = idx < N4 && (unsigned)gh < (unsigned)p.H && f0 >= 0 && f0 < rowf;
            vo[u] = ok ? (uint32_t)(gh * rowf + f0) * 4u : 0xFFFFFFF0u;
            asm volatile("" : "+v"(vo[u]));
        }
#pragma unroll
        for (int u = 0; u < U4; ++u) v[u] = __builtin_bit_cast(f32x4l, __builtin_amdgcn_raw_buffer_load_b128(rs, vo[u], 0, 0));
        const unsigned char* tb = reinterpret_cast<const unsigned char*>(sT);
#pragma unroll
        for (int u = 0; u < U4; ++u) {
            const int idx = threadIdx.x + u * 256;
            const int ih = idx / Q, q = idx - ih * Q;
            const float4 xv = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
            const float4 qv = enc4_f32(xv, p.enc.r1, p.enc.lo, p.enc.hi, tb);   // NaN in -> NaN out
            const float qs[4] = {qv.x, qv.y, qv.z, qv.w};
            float* trow = tile + ih * IWC + 4 * q - 1;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (idx < N4 && 4 * q + e >= 1 && 4 * q + e - 1 < IWC) trow[e] = qs[e];
        }
    }
    __syncthreads();

    // per-lane epilogue constants: this lane's 8 consecutive channels
    const int c0 = 8 * kq;
    float bq[8], psc[8], psh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bq[e] = bias ? (bias[c0 + e] / p.s1) / p.s2 : 0.f;
        psc[e] = p.post.scale ? p.post.scale[c0 + e] : 1.f;
        psh[e] = p.post.scale ? p.post.shift[c0 + e] : 0.f;
    }
    const unsigned char* tbytes = reinterpret_cast<const unsigned char*>(tile);
    const int gow = tw * kStemTW + px;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int row = wave * 4 + r4;
        const uint32_t pbase = (uint32_t)((row * S) * IWC + px * S * C) * 4u;
        f32x4m acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            uint32_t a = pbase + koff[ks];
            if (ks == 6) a = kq == 3 ? (uint32_t)N_IN * 4u : a;   // tap 27 does not exist: the zero word
            const float xb = *reinterpret_cast<const float*>(tbytes + a);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks][0], xb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks][1], xb, acc1, 0, 0, 0);
        }
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float u = (((e < 4 ? acc0[e] : acc1[e - 4]) + bq[e]) * p.s1) * p.s2;
            if (p.post.scale) {
                u = __builtin_fmaf(u, psc[e], psh[e]);
                if (p.post.layerout) u = layerout1(u);
            }
            r[e] = u;
        }
        const int goh = th * TH + row;
        const bool live = goh < p.Ho && gow < p.Wo;
        if constexpr (YC) {
            uint32_t cA, cB;
            const float4 ra = make_float4(r[0], r[1], r[2], r[3]), rb = make_float4(r[4], r[5], r[6], r[7]);
            if (p.sgn) {
                cA = enc4_code<true>(ra, p.enc_out.r1, p.enc_out.lo, p.enc_out.hi, senc);
                cB = enc4_code<true>(rb, p.enc_out.r1, p.enc_out.lo, p.enc_out.hi, senc);
                if (p.fmt_out == kFmtSfp7) {
                    cA = (cA & 0x3F3F3F3Fu) | ((cA & 0x80808080u) >> 1);
                    cB = (cB & 0x3F3F3F3Fu) | ((cB & 0x80808080u) >> 1);
                }
            } else {   // the ReLU is the quantizer's
                cA = enc4_code_relu(ra, p.enc_out.r1, p.enc_out.lo, p.enc_out.hi, senc);
                cB = enc4_code_relu(rb, p.enc_out.r1, p.enc_out.lo, p.enc_out.hi, senc);
            }
            if (live) {
                uint8_t* yc = reinterpret_cast<uint8_t*>(y) + (((size_t)n * p.Ho + goh) * p.Wo + gow) * O + c0;
                *reinterpret_cast<uint2*>(yc) = make_uint2(cA, cB);
            }
        } else {
            if (p.post.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = fmaxf(r[e], 0.f);
            }
            // stage the unit (16 pixels x 32 channels = 2 KiB) so that each store instruction writes 8 whole pixels
            float* sw = stg + wave * (16 * SP);
            *reinterpret_cast<float4*>(sw + px * SP + c0) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4*>(sw + px * SP + c0 + 4) = make_float4(r[4], r[5], r[6], r[7]);
            // LDS operations of one wave execute in order: the reads below see the writes above
            float* yrow = y + (((size_t)n * p.Ho + goh) * p.Wo + tw * kStemTW) * O;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int spx = h * 8 + (lane >> 3);
                const float4 vv = *reinterpret_cast<const float4*>(sw + spx * SP + (lane & 7) * 4);
                if (goh < p.Ho && tw * kStemTW + spx < p.Wo) st_stream4<SLFP_NT_STEM>(yrow + spx * O + (lane & 7) * 4, vv);
            }
        }
    }
}

bool stem_applicable(const slfp_conv2d_desc& d) {
    const int O = (int)d.c_out, C = (int)d.c_in;
    if (d.groups != 1 || C > 4 || d.dil_h != 1 || d.dil_w != 1 || d.stride_h != d.stride_w || d.stride_h > 4) return false;
    if (d.kh == 1 && d.kw == 1) return false;
    if (O != 16 && O != 32 && O != 64) return false;  // O/4 lanes per pixel: power of two, 16 columns x (O/4) <= 256
    const size_t n_w = (size_t)d.kh * d.kw * C * O;
    const size_t ih = (size_t)(kStemTH - 1) * d.stride_h + d.kh, iwc = ((size_t)(kStemTW - 1) * d.stride_w + d.kw) * C;
    return 64 + (((n_w + 3) & ~(size_t)3) + ih * iwc) * sizeof(float) <= 64 * 1024;
}

// Returns SLFP_OK if it launched, 1 if this geometry is not a stem (caller falls back).
static int try_launch_stem(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const float* wq_hwio,
                           const float* bias, const PostOp& post, float* y, hipStream_t stream, const CodeIo* io = nullptr) {
    const int O = (int)d.c_out, C = (int)d.c_in;
    if (!stem_applicable(d)) return 1;
    StemParams p;
    p.post = post;
    p.N = (int)d.n; p.H = (int)d.h; p.W = (int)d.w; p.C = C; p.O = O; p.KH = (int)d.kh; p.KW = (int)d.kw;
    p.s = d.stride_h; p.ph = d.pad_h; p.pw = d.pad_w;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out;
    p.tiles_h = (int)ceil_div(p.Ho, kStemTH); p.tiles_w = (int)ceil_div(p.Wo, kStemTW);
    p.IH = (kStemTH - 1) * p.s + p.KH;
    p.IWC = ((kStemTW - 1) * p.s + p.KW) * C;
    p.l4_shift = O == 16 ? 2 : (O == 32 ? 3 : 4);
    const int groups = 256 >> p.l4_shift;           // pixel groups: 16 columns x (groups/16) row sets
    p.P = kStemTH / (groups / kStemTW);             // 2, 4 or 8 rows per thread
    p.step_h = 256 / p.IWC; p.step_j = 256 % p.IWC;
    p.sd = make_scale_div(d.ka); p.s1 = plan.s1; p.s2 = plan.s2;
    const size_t n_w = (size_t)p.KH * p.KW * C * O;
    const size_t lds = 64 + (((n_w + 3) & ~(size_t)3) + (size_t)p.IH * p.IWC) * sizeof(float);
    if (lds > 64 * 1024) return 1;
    const int64_t nblocks = (int64_t)p.N * p.tiles_h * p.tiles_w;
    if (nblocks > 0x7FFFFFFF) return 1;
    p.nblocks = (uint32_t)nblocks;
    if (p.KH == 3 && p.KW == 3 && C == 3 && p.s == 2 && O == 32 && (int64_t)p.H * p.W * C < (1ll << 30)) {
        // the MobileNetV1 stem (nets_imgnet/mobilenetv1.py:44): fully specialised variant
        if (const EncArgs* t = act_table(d.ka, plan.fmt_act, kEncF32)) {
            p.enc = *t;
            constexpr int TH = 16;
            p.tiles_h = (int)ceil_div(p.Ho, TH);
            p.nblocks = (uint32_t)((int64_t)p.N * p.tiles_h * p.tiles_w);
            // float32 matrix-core stem (k_stem_mx) with aligned 16-byte halo loads: measured (round 3, same box, us per launch at
            // batch 256) 111.8 vs 123.0 for the vector kernel with float32 output, 117.8 vs 131.4 with code output ->
            // the default for both; SLFP_STEM_OLD keeps the vector kernel.  Bit-identical either way (tests/test_gpu_parity.py).
            const bool mx = !switches().stem_old && (p.W * C) % 4 == 0;   // k_stem_mx loads aligned float4 pieces of the image rows
            if (mx && !(io && io->y_codes)) {
                hipLaunchKernelGGL((k_stem_mx<false>), dim3(p.nblocks), dim3(256), 0, stream, x, wq_hwio, bias, y, p);
                return check_launch("slfp stem conv kernel (3x3x3 s2 -> 32, float32 MFMA)");
            }
            if (io && io->y_codes) {   // output as the next layer's codes (slfp_conv2d_fwd_codes)
                const EncArgs* tc = enc_table(io->y_ka, io->y_fmt, kEncCode);
                if (!tc->valid) return fail(SLFP_ERR_UNSUPPORTED, "stem (codes): no code table for the consumer's scale %g", (double)io->y_ka);
                p.enc_out = enc_compact(*tc);
                p.sgn = post.relu ? 0 : 1;
                p.fmt_out = io->y_fmt;
                if (mx) hipLaunchKernelGGL((k_stem_mx<true>), dim3(p.nblocks), dim3(256), 0, stream, x, wq_hwio, bias, y, p);
                else hipLaunchKernelGGL((k_stem_fixed<kFmtAct8, 3, 3, 3, 2, 32, true, TH, true>), dim3(p.nblocks), dim3(256), 0, stream, x, wq_hwio, bias, y, p);
                return check_launch("slfp stem conv kernel (3x3x3 s2 -> 32, codes out)");
            }
            hipLaunchKernelGGL((k_stem_fixed<kFmtAct8, 3, 3, 3, 2, 32, true, TH>), dim3(p.nblocks), dim3(256), 0, stream, x, wq_hwio, bias, y, p);
        } else if (plan.fmt_act == kFmtAct8)
            hipLaunchKernelGGL((k_stem_fixed<kFmtAct8, 3, 3, 3, 2, 32>), dim3(p.nblocks), dim3(256), 0, stream, x, wq_hwio, bias, y, p);
        else
            hipLaunchKernelGGL((k_stem_fixed<kFmtSfp7, 3, 3, 3, 2, 32>), dim3(p.nblocks), dim3(256), 0, stream, x, wq_hwio, bias, y, p);
        return check_launch("slfp stem conv kernel (3x3x3 s2 -> 32)");
    }
    if (plan.fmt_act == kFmtAct8)
        hipLaunchKernelGGL((k_stem<kFmtAct8>), dim3(p.nblocks), dim3(256), lds, stream, x, wq_hwio, bias, y, p);
    else
        hipLaunchKernelGGL((k_stem<kFmtSfp7>), dim3(p.nblocks), dim3(256), lds, stream, x, wq_hwio, bias, y, p);
    return check_launch("slfp stem conv kernel");
}

// The MobileNetV1 stem (float32 image in) writing the next layer's codes: only the fully specialised table variant.
bool stem_codes_applicable(const slfp_conv2d_desc& d, const ConvPlan& plan, int post_flags) {
    if (plan.family != kDirect || !stem_applicable(d) || (post_flags & SLFP_POST_LAYEROUT)) return false;
    if (!(d.kh == 3 && d.kw == 3 && d.c_in == 3 && d.stride_h == 2 && d.c_out == 32 && plan.fmt_act == kFmtAct8)) return false;
    if ((int64_t)d.h * d.w * d.c_in >= (1ll << 30)) return false;
    return act_table(d.ka, plan.fmt_act, kEncF32) != nullptr;
}

int launch_stem_codes(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const float* wq_hwio, const float* bias,
                      const PostOp& post, void* y, const CodeIo& io, hipStream_t stream) {
    const int rc = try_launch_stem(d, plan, x, wq_hwio, bias, post, reinterpret_cast<float*>(y), stream, &io);
    return rc == 1 ? fail(SLFP_ERR_UNSUPPORTED, "stem (codes): unsupported geometry") : rc;
}

int launch_direct(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const float* wq_hwio,
                  const float* bias, const PostOp& post, float* y, hipStream_t stream) {
    {
        const int rc = try_launch_stem(d, plan, x, wq_hwio, bias, post, y, stream);
        if (rc <= 0) return rc;  // launched (0) or failed (<0); 1 = not a stem geometry
    }
    DirParams p;
    p.post = post;
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
