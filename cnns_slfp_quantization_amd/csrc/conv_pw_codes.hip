// conv_pw_codes.hip -- SLFP-quantized pointwise (1x1) convolution on 1-byte activation codes, gfx950 matrix cores.
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) for 1x1 kernels, groups == 1 (the 13 "pw" layers of
// MobileNetV1, nets_imgnet/mobilenetv1.py:31) when the layer sits inside a chain of quantized convolutions
// (csrc/slfp_codes.hpp): the input arrives as the codes the previous layer's epilogue wrote (input_q = QA(x / Ka),
// utils/conv2d_func.py:21, is already applied), the output leaves as the NEXT layer's codes (YC) or as float32.
//
// Same contraction as conv_pw.hip -- v_mfma_f32_16x16x32_f16, A = W fragments of the SAME prepared blob, B = fp16(16 * Q),
// one MFMA per 32-deep k-step in k order from +0, the same epilogue arithmetic -- so the outputs are bit-identical to
// the float32-interface kernels fed with the decoded tensor (single-pass mode; SFP<3,3> exact).  What differs is how the
// B fragments are made and how the result leaves:
//   * a lane loads 16 CONSECUTIVE channel codes of its pixel with one 16-byte load (a wave: 16 pixels x 64 contiguous
//     bytes); a 4 x 4 dword transpose across the wave's four 16-lane rows (2 x v_permlane32_swap + 2 x
//     v_permlane16_swap) turns that into the fragment order of the blob (lane-quarter kq owns channels 4kq.. and
//     16 + 4kq.. of every 32), so the weights need no second layout;
//   * byte -> fp16 operand is one SDWA shift + one 256-entry LDS lookup (+ half a v_bfi to pair two): 1.5 VALU per
//     element where the float32 interface spends 6-7 on the encode;
//   * YC: the 4 output channels a lane holds per 16-channel tile become 4 code bytes (5 VALU + 1 LDS each, written in
//     place by the byte select); the dwords of 4 tiles go back through the same transpose and leave as ONE 16-byte store
//     per lane: per pixel the wave writes 64 contiguous bytes.
// Two kernels, as in conv_pw.hip: k_pwc_stream (W resident in LDS, persistent, a wave's unit = 16 pixels) and
// k_pwc_tiled (W fragments streamed from L2, X tile decoded once into the swizzled LDS image).
#include "slfp_device.hpp"
#include "slfp_enc.hpp"
#include "slfp_codes.hpp"
#include "slfp_host.hpp"
#include "conv_pw_params.hpp"

namespace slfp {

struct PwcParams {
    const uint8_t* x;
    const _Float16* whi;
    const float* bias;
    void* y;
    int64_t M;            // output pixels
    int K, N, KS, n_tiles;
    int H, W, Ho, Wo, S;
    float s1, s2, s1x;
    uint32_t m_blocks, n_blocks, nblocks;
    int rb;
    int sgn;              // YC: no ReLU in front of the output quantizer: codes carry a sign
    int fmt_out;          // YC: format of the output codes (kFmtAct8 | kFmtSfp7)
    PostOp post;
    EncArgs enc;          // YC: code table of the consumer's Ka (kEncCode)
};

typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t xc_row_offset(const PwcParams& p, int64_t m) {
    if (p.S == 1) return (size_t)m * p.K;
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    const int64_t img = m / hw, r = m - img * hw;
    const int oh = (int)(r / p.Wo), ow = (int)(r - (int64_t)oh * p.Wo);
    return (size_t)(((img * p.H) + (int64_t)oh * p.S) * p.W + (int64_t)ow * p.S) * p.K;
}

__device__ __forceinline__ half8 dec_frag(uint32_t ca, uint32_t cb, const unsigned char* dtab) {
    const uint2 pa = dec4_f16(ca, dtab), pb = dec4_f16(cb, dtab);
    return __builtin_bit_cast(half8, u32x4c{pa.x, pa.y, pb.x, pb.y});
}

// ======================================================================================
// k_pwc_stream: W resident in LDS, codes straight into MFMA fragments, 16-pixel work units.
// ======================================================================================
constexpr int kPwcThreads = 512;

// KS: 32-deep k-steps (K = 32 * KS exactly); XW: K is a multiple of 64 (16-byte code loads + transpose), else two
// dword loads per k-step; YC: output codes (N a multiple of 16), else float32.
template <int FMT, int KS, bool XW, bool YC>
__global__ __launch_bounds__(kPwcThreads) void k_pwc_stream(const PwcParams p) {
    static_assert(!XW || KS % 2 == 0, "16-byte code loads cover two k-steps");
    __shared__ __attribute__((aligned(16))) uint32_t sdec[256];
    __shared__ __attribute__((aligned(16))) unsigned char senc[YC ? kPwTab : 16];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* wl = reinterpret_cast<_Float16*>(smem);
    const int wfrags = p.n_tiles * p.KS;  // 1 KiB each
    dec_fill<FMT, kDecF16D, kPwcThreads>(sdec);
    if constexpr (YC) enc_fill<kPwcThreads>(reinterpret_cast<uint2*>(senc), p.enc);
    for (int i = threadIdx.x; i < wfrags * 64; i += kPwcThreads)
        reinterpret_cast<half8*>(wl)[i] = reinterpret_cast<const half8*>(p.whi)[i];
    float* ep = reinterpret_cast<float*>(smem + (size_t)wfrags * 1024);   // [256 * bias/s1/s2 | post scale | post shift]
    const int n_pad = p.n_tiles * 16;
    const bool has_vec = p.bias != nullptr || p.post.scale != nullptr;   // wave-uniform
    if (has_vec) {
        for (int i = threadIdx.x; i < n_pad; i += kPwcThreads) {
            const bool in = i < p.N;
            ep[i] = (p.bias && in) ? 256.f * ((p.bias[i] / p.s1) / p.s2) : 0.f;
            ep[n_pad + i] = (p.post.scale && in) ? p.post.scale[i] : 1.f;
            ep[2 * n_pad + i] = (p.post.scale && in) ? p.post.shift[i] : 0.f;
        }
    }
    __syncthreads();
    const unsigned char* dtab = reinterpret_cast<const unsigned char*>(sdec);
    const float r1 = p.enc.r1, lo = p.enc.lo, hi = p.enc.hi;

    const int lane = threadIdx.x & 63;
    const int col = lane & 15, kq = lane >> 4;
    const int64_t n_groups = (p.M + 15) >> 4;
    const int64_t waves_total = (int64_t)gridDim.x * (kPwcThreads / 64);
    const int64_t wave_id = (int64_t)blockIdx.x * (kPwcThreads / 64) + (threadIdx.x >> 6);

    for (int64_t g = wave_id; g < n_groups; g += waves_total) {
        const int64_t m = g * 16 + col;
        const bool live = m < p.M;
        const uint8_t* xr = p.x + xc_row_offset(p, live ? m : p.M - 1);   // rows past the end: clamped, computed, dropped
        uint32_t cw[KS][2];
        if constexpr (XW) {
            u32x4c v[KS / 2];
#pragma unroll
            for (int c2 = 0; c2 < KS / 2; ++c2) v[c2] = *reinterpret_cast<const u32x4c*>(xr + c2 * 64 + kq * 16);
#pragma unroll
            for (int c2 = 0; c2 < KS / 2; ++c2) {
                uint32_t a0 = v[c2][0], a1 = v[c2][1], a2 = v[c2][2], a3 = v[c2][3];
                rows_transpose4(a0, a1, a2, a3);   // lane-quarter kq now holds channels 64 c2 + 16 i + 4 kq
                cw[2 * c2][0] = a0; cw[2 * c2][1] = a1; cw[2 * c2 + 1][0] = a2; cw[2 * c2 + 1][1] = a3;
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                cw[ks][0] = *reinterpret_cast<const uint32_t*>(xr + ks * 32 + kq * 4);
                cw[ks][1] = *reinterpret_cast<const uint32_t*>(xr + ks * 32 + 16 + kq * 4);
            }
        }
        half8 xh[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xh[ks] = dec_frag(cw[ks][0], cw[ks][1], dtab);

        auto tile_out = [&](int j) {
            floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
            const _Float16* wj = wl + ((size_t)j * p.KS) * 512 + lane * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 wh = *reinterpret_cast<const half8*>(wj + ks * 512);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[ks], acc, 0, 0, 0);
            }
            const int n = j * 16 + kq * 4;
            float4 r;
            if (has_vec) {
                const float4 bq = *reinterpret_cast<const float4*>(ep + n);
                const float4 sc = *reinterpret_cast<const float4*>(ep + n_pad + n);
                const float4 sh = *reinterpret_cast<const float4*>(ep + 2 * n_pad + n);
                r = epilogue(acc, bq, p.s1x, p.s2);
                if (p.post.scale) {
                    r.x = __builtin_fmaf(r.x, sc.x, sh.x); r.y = __builtin_fmaf(r.y, sc.y, sh.y);
                    r.z = __builtin_fmaf(r.z, sc.z, sh.z); r.w = __builtin_fmaf(r.w, sc.w, sh.w);
                }
            } else {
                r = epilogue(acc, make_float4(0.f, 0.f, 0.f, 0.f), p.s1x, p.s2);
            }
            if constexpr (!YC) {   // with code output the ReLU is the quantizer's (enc4_code_relu)
                if (p.post.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            }
            return r;
        };
        if constexpr (YC) {
            uint8_t* yr = reinterpret_cast<uint8_t*>(p.y) + (size_t)m * p.N;
            for (int j0 = 0; j0 < p.n_tiles; j0 += 4) {   // n_tiles is a multiple of 4 (the blob is padded to 64 channels)
                uint32_t c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 r = tile_out(j0 + j);
                    if (p.sgn) c[j] = code_sign4(enc4_code<false>(r, r1, lo, hi, senc), r, p.fmt_out);
                    else c[j] = enc4_code_relu(r, r1, lo, hi, senc);
                }
                rows_transpose4(c[0], c[1], c[2], c[3]);   // lane-quarter kq now holds channels 16 (j0 + kq) + 0..15
                const int n = (j0 + kq) * 16;
                if (live && n < p.N) *reinterpret_cast<u32x4c*>(yr + n) = u32x4c{c[0], c[1], c[2], c[3]};
            }
        } else {
            float* yr = reinterpret_cast<float*>(p.y) + (size_t)m * p.N + kq * 4;
            for (int j = 0; j < p.n_tiles; ++j) {
                const float4 r = tile_out(j);
                if (live && j * 16 + kq * 4 < p.N) *reinterpret_cast<float4*>(yr + j * 16) = r;
            }
        }
    }
}

// ======================================================================================
// k_pwc_slice: deep layers (K = 256 ... 1024).  W does not fit LDS, but a SLICE of output channels does: a workgroup keeps
// the fragments of NTS channel tiles x all K resident (<= 128 KiB) and streams pixel units past them, as k_pwc_stream does.
// Every slice reads all of X -- at 1 B per element that costs little (the float32 interface could not afford it: X four
// to sixteen times through the encoder): the n_slices workgroups that work on the same pixels share an XCD (blocks b and
// b + 8), so the re-reads are L2 hits, and a byte decodes in 1.5 VALU instructions.  What it removes is k_pwc_tiled's
// per-workgroup W stream from L2 (512 KiB per 64 pixels: 4x the X + Y bytes of these layers) and its barrier per stage.
// K is swept in chunks of KC k-steps with the accumulators of the slice's NTS tiles in registers; k order and MFMA shape
// are those of the other pointwise kernels (bit-identical results).
// ======================================================================================
// The codes of the NEXT chunk -- or of the next unit's first chunk -- are always in flight under the current chunk's MFMAs.
// Measured (round 3, batch 256, us per layer, k_pwc_stream / k_pwc_tiled -> this kernel): 256->256 @28 73 -> 56, 256->512 @14
// 42 -> 34; but 512->512 @14 54 -> 54-64 and 1024->1024 @7 45 -> 66-76 (512 or 1024 threads): with W in LDS every MFMA needs a
// 1 KiB fragment read, exactly the LDS rate of a CU, and the decode lookups come on top -- at K >= 512 the register-blocked
// k_pwc_tiled (4 x 4 tiles per wave, W straight from L2) stays.  Used for K = 256 only (two workgroups per CU).
constexpr int kSliceThreads = 512;
template <int FMT, int NTS, bool YC>
__global__ __launch_bounds__(kSliceThreads) void k_pwc_slice(const PwcParams p) {
    constexpr int KC = 8;   // k-steps per chunk = 256 channels = 4 x 16-byte code loads per lane
    static_assert(NTS % 4 == 0, "code output leaves in groups of 4 channel tiles");
    __shared__ __attribute__((aligned(16))) uint32_t sdec[256];
    __shared__ __attribute__((aligned(16))) unsigned char senc[YC ? kPwTab : 16];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* wl = reinterpret_cast<_Float16*>(smem);
    const int n_slices = p.n_tiles / NTS;
    // blocks b and b + 8 share an XCD: the slices of one pixel range sit on consecutive (b / 8)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int slice = q % n_slices;
    const int rank = xcd + 8 * (q / n_slices);
    const int n_ranks = 8 * ((int)(gridDim.x >> 3) / n_slices);
    const int wfrags = NTS * p.KS;  // 1 KiB each
    dec_fill<FMT, kDecF16D, kSliceThreads>(sdec);
    if constexpr (YC) enc_fill<kSliceThreads>(reinterpret_cast<uint2*>(senc), p.enc);
    {
        const half8* src = reinterpret_cast<const half8*>(p.whi) + (size_t)slice * wfrags * 64;
        for (int i = threadIdx.x; i < wfrags * 64; i += kSliceThreads) reinterpret_cast<half8*>(wl)[i] = src[i];
    }
    float* ep = reinterpret_cast<float*>(smem + (size_t)wfrags * 1024);   // [256 * bias/s1/s2 | post scale | post shift] of the slice
    constexpr int NCH = NTS * 16;
    const int n_lo = slice * NCH;
    const bool has_vec = p.bias != nullptr || p.post.scale != nullptr;
    if (has_vec) {
        for (int i = threadIdx.x; i < NCH; i += kSliceThreads) {
            const bool in = n_lo + i < p.N;
            ep[i] = (p.bias && in) ? 256.f * ((p.bias[n_lo + i] / p.s1) / p.s2) : 0.f;
            ep[NCH + i] = (p.post.scale && in) ? p.post.scale[n_lo + i] : 1.f;
            ep[2 * NCH + i] = (p.post.scale && in) ? p.post.shift[n_lo + i] : 0.f;
        }
    }
    __syncthreads();
    const unsigned char* dtab = reinterpret_cast<const unsigned char*>(sdec);
    const float r1 = p.enc.r1, lo = p.enc.lo, hi = p.enc.hi;

    const int lane = threadIdx.x & 63;
    const int col = lane & 15, kq = lane >> 4;
    const int64_t n_groups = (p.M + 15) >> 4;
    const int64_t stride = (int64_t)n_ranks * (kSliceThreads / 64);
    const int n_chunks = p.KS / KC;

    u32x4c v[KC / 2];
    {
        const int64_t g0 = (int64_t)rank * (kSliceThreads / 64) + (threadIdx.x >> 6);
        const int64_t m0 = (g0 < n_groups ? g0 : n_groups - 1) * 16 + col;
        const uint8_t* x0 = p.x + xc_row_offset(p, m0 < p.M ? m0 : p.M - 1) + kq * 16;
#pragma unroll
        for (int c2 = 0; c2 < KC / 2; ++c2) v[c2] = *reinterpret_cast<const u32x4c*>(x0 + c2 * 64);
    }
    for (int64_t g = (int64_t)rank * (kSliceThreads / 64) + (threadIdx.x >> 6); g < n_groups; g += stride) {
        const int64_t m = g * 16 + col;
        const bool live = m < p.M;
        const uint8_t* xr = p.x + xc_row_offset(p, live ? m : p.M - 1) + kq * 16;
        // first chunk of the next unit of this wave (its own unit again on the last round: L1 hits, unused)
        const int64_t gn = g + stride < n_groups ? g + stride : g;
        const int64_t mn = gn * 16 + col;
        const uint8_t* xrn = p.x + xc_row_offset(p, mn < p.M ? mn : p.M - 1) + kq * 16;
        floatx4 acc[NTS];
#pragma unroll
        for (int j = 0; j < NTS; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < n_chunks; ++kc) {
            half8 xh[KC];
#pragma unroll
            for (int c2 = 0; c2 < KC / 2; ++c2) {
                uint32_t a0 = v[c2][0], a1 = v[c2][1], a2 = v[c2][2], a3 = v[c2][3];
                rows_transpose4(a0, a1, a2, a3);
                xh[2 * c2] = dec_frag(a0, a1, dtab);
                xh[2 * c2 + 1] = dec_frag(a2, a3, dtab);
            }
            {   // the next chunk's codes -- after the last chunk: the next unit's first -- fly during this chunk's MFMAs
                const uint8_t* nx = kc + 1 < n_chunks ? xr + (kc + 1) * (KC * 32) : xrn;
#pragma unroll
                for (int c2 = 0; c2 < KC / 2; ++c2) v[c2] = *reinterpret_cast<const u32x4c*>(nx + c2 * 64);
            }
            const _Float16* wk = wl + (size_t)(kc * KC) * 512 + lane * 8;
#pragma unroll
            for (int j = 0; j < NTS; ++j) {
#pragma unroll
                for (int ks = 0; ks < KC; ++ks) {
                    const half8 wh = *reinterpret_cast<const half8*>(wk + ((size_t)j * p.KS + ks) * 512);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[ks], acc[j], 0, 0, 0);
                }
            }
        }
        auto finish = [&](int j) {
            const int n = j * 16 + kq * 4;   // channel inside the slice
            float4 r;
            if (has_vec) {
                const float4 bq = *reinterpret_cast<const float4*>(ep + n);
                const float4 sc = *reinterpret_cast<const float4*>(ep + NCH + n);
                const float4 sh = *reinterpret_cast<const float4*>(ep + 2 * NCH + n);
                r = epilogue(acc[j], bq, p.s1x, p.s2);
                if (p.post.scale) {
                    r.x = __builtin_fmaf(r.x, sc.x, sh.x); r.y = __builtin_fmaf(r.y, sc.y, sh.y);
                    r.z = __builtin_fmaf(r.z, sc.z, sh.z); r.w = __builtin_fmaf(r.w, sc.w, sh.w);
                }
            } else {
                r = epilogue(acc[j], make_float4(0.f, 0.f, 0.f, 0.f), p.s1x, p.s2);
            }
            if constexpr (!YC) {
                if (p.post.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            }
            return r;
        };
        if constexpr (YC) {
            uint8_t* yr = reinterpret_cast<uint8_t*>(p.y) + (size_t)m * p.N + n_lo;
#pragma unroll
            for (int j0 = 0; j0 < NTS; j0 += 4) {
                uint32_t c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 r = finish(j0 + j);
                    if (p.sgn) c[j] = code_sign4(enc4_code<false>(r, r1, lo, hi, senc), r, p.fmt_out);
                    else c[j] = enc4_code_relu(r, r1, lo, hi, senc);
                }
                rows_transpose4(c[0], c[1], c[2], c[3]);
                const int n = (j0 + kq) * 16;
                if (live && n_lo + n < p.N) *reinterpret_cast<u32x4c*>(yr + n) = u32x4c{c[0], c[1], c[2], c[3]};
            }
        } else {
            float* yr = reinterpret_cast<float*>(p.y) + (size_t)m * p.N + n_lo + kq * 4;
#pragma unroll
            for (int j = 0; j < NTS; ++j) {
                const float4 r = finish(j);
                if (live && n_lo + j * 16 + kq * 4 < p.N) *reinterpret_cast<float4*>(yr + j * 16) = r;
            }
        }
    }
}

// ======================================================================================
// k_pwc_tiled: codes -> swizzled fp16 LDS tile (decoded once), W fragments straight from L2.
// ======================================================================================
// K a multiple of 64; NT = 4 channel tiles per wave.  STG-style float32 epilogue as in conv_pw.hip's k_pw_tiled.
template <int FMT, int WM, int WN, int MT, bool YC>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) <= 4 ? 2 : 4) void k_pwc_tiled(const PwcParams p) {
    constexpr int NT = 4;
    constexpr int T = 64 * WM * WN;
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int NLD = BM * 16 / T;  // code dwords per thread per 64-deep stage
    static_assert(BM * 16 % T == 0, "staging must divide evenly");
    constexpr int XBYTES = BM * 128;

    __shared__ __attribute__((aligned(16))) uint32_t sdec[256];
    __shared__ __attribute__((aligned(16))) unsigned char senc[YC ? kPwTab : 16];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;  // [2 buffers][BM rows][128 B]
    dec_fill<FMT, kDecF16D, T>(sdec);
    if constexpr (YC) enc_fill<T>(reinterpret_cast<uint2*>(senc), p.enc);
    const unsigned char* dtab = reinterpret_cast<const unsigned char*>(sdec);

    const uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const uint32_t nb = b % p.n_blocks, mb = b / p.n_blocks;
    const int64_t m0 = (int64_t)mb * p.rb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, kq = lane >> 4;

    // staging geometry as in k_pw_tiled: dword #i of a thread = the 4 codes of row (tid>>4) + i*(T/16), k = (tid&15)*4
    const int kc = threadIdx.x & 15;
    const int st_chunk = (kc >> 3) * 4 + (kc & 3);
    const uint32_t st_sub = (uint32_t)((kc & 7) >> 2) * 8u;
    const uint8_t* src[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = (threadIdx.x >> 4) + i * (T / 16);
        int64_t m = m0 + (row < p.rb ? row : p.rb - 1);
        m = m < p.M ? m : p.M - 1;
        src[i] = p.x + xc_row_offset(p, m) + kc * 4;
    }
    uint32_t st[NLD];
    auto load_stage = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) st[i] = *reinterpret_cast<const uint32_t*>(src[i] + t * 64);
    };
    auto decode_store = [&](int buf) {
        unsigned char* hi = xs + (size_t)buf * XBYTES;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = (threadIdx.x >> 4) + i * (T / 16);
            *reinterpret_cast<uint2*>(hi + lds_x_off(row, st_chunk) + st_sub) = dec4_f16(st[i], dtab);
        }
    };

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int KT = p.KS >> 1;  // 64-deep stages
    const int ntile0 = (int)nb * (BN / 16) + wn * NT;
    // W fragments by buffer loads: descriptor + per-tile byte offsets in scalar registers (the wave index is made uniform for
    // the compiler), ONE vector register of address (lane * 16) for all of them -- the double buffer below needs the registers
    const int wn_u = __builtin_amdgcn_readfirstlane(wn);
    const int ntile0u = (int)nb * (BN / 16) + wn_u * NT;
    const uint64_t wbytes = (uint64_t)p.n_tiles * p.KS * 1024;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.whi), 0,
                                                                        (uint32_t)(wbytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : wbytes), 0x00020000);
    uint32_t wsoff[NT];   // scalar
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nt = ntile0u + j < p.n_tiles ? ntile0u + j : p.n_tiles - 1;
        wsoff[j] = (uint32_t)nt * (uint32_t)p.KS * 1024u;
    }
    const uint32_t wvoff = (uint32_t)lane * 16u;
    // W fragments double-buffered by k-step (a stage of codes is 2 registers per lane, so there is room -- the float32-interface
    // kernel has none): W(2t+1) is requested at the top of stage t, BEFORE the X loads of stage t+2, W(2t+2) once the first
    // k-step's MFMAs have read its buffer.  Every vmcnt wait is then for loads issued at least half a stage earlier, and no W
    // wait sits behind X loads that have not been waited for already (loads retire in order through one counter).
    half8 whb[2][NT];
    auto load_w = [&](int kstep, int b) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
            const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, wsoff[j] + (uint32_t)kstep * 1024u, 0);
            whb[b][j] = __builtin_bit_cast(half8, v);
        }
    };
    auto mfma_step = [&](int buf, int ks, int b) {
        const unsigned char* hi = xs + (size_t)buf * XBYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = (wm * MT + i) * 16 + col;
            const half8 xh = *reinterpret_cast<const half8*>(hi + lds_x_off(row, ks * 4 + kq));
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whb[b][j], xh, acc[i][j], 0, 0, 0);
        }
    };

    load_stage(0);
    __syncthreads();  // tables visible
    decode_store(0);
    load_w(0, 0);
    load_stage(KT > 1 ? 1 : 0);
    __syncthreads();
    for (int t = 0; t + 1 < KT; ++t) {  // branch-free body
        const int buf = t & 1;
        load_w(t * 2 + 1, 1);
        decode_store(buf ^ 1);
        load_stage(t + 2 < KT ? t + 2 : KT - 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(buf, 0, 0);
        load_w(t * 2 + 2, 0);
        mfma_step(buf, 1, 1);
        __syncthreads();
    }
    {
        const int t = KT - 1, buf = t & 1;
        load_w(t * 2 + 1, 1);
        mfma_step(buf, 0, 0);
        mfma_step(buf, 1, 1);
    }

    const int n_lo = (int)nb * BN;
    float* lsc = reinterpret_cast<float*>(xs);   // the (now free) X tile
    float* lsh = lsc + BN;
    if (p.post.scale) {
        __syncthreads();
        for (int i = threadIdx.x; i < BN; i += T) {
            const bool in = n_lo + i < p.N;
            lsc[i] = in ? p.post.scale[n_lo + i] : 1.f;
            lsh[i] = in ? p.post.shift[n_lo + i] : 0.f;
        }
        __syncthreads();
    }
    auto out_tile = [&](int i, int j) {
        const int n = (ntile0 + j) * 16 + kq * 4;
        const bool in = n < p.N;
        float4 r = epilogue(acc[i][j], (in && p.bias) ? make_float4(256.f * ((p.bias[n] / p.s1) / p.s2), 256.f * ((p.bias[n + 1] / p.s1) / p.s2),
                                                                     256.f * ((p.bias[n + 2] / p.s1) / p.s2), 256.f * ((p.bias[n + 3] / p.s1) / p.s2))
                                                      : make_float4(0.f, 0.f, 0.f, 0.f), p.s1x, p.s2);
        if (p.post.scale) {
            const float4 sc = *reinterpret_cast<const float4*>(lsc + (in ? n - n_lo : 0));
            const float4 sh = *reinterpret_cast<const float4*>(lsh + (in ? n - n_lo : 0));
            r.x = __builtin_fmaf(r.x, sc.x, sh.x); r.y = __builtin_fmaf(r.y, sc.y, sh.y);
            r.z = __builtin_fmaf(r.z, sc.z, sh.z); r.w = __builtin_fmaf(r.w, sc.w, sh.w);
        }
        if constexpr (!YC) {
            if (p.post.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        }
        return r;
    };
    if constexpr (YC) {
        const float r1 = p.enc.r1, lo = p.enc.lo, hi = p.enc.hi;
        uint8_t* yb = reinterpret_cast<uint8_t*>(p.y);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            uint32_t c[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 r = out_tile(i, j);
                if (p.sgn) c[j] = code_sign4(enc4_code<false>(r, r1, lo, hi, senc), r, p.fmt_out);
                else c[j] = enc4_code_relu(r, r1, lo, hi, senc);
            }
            rows_transpose4(c[0], c[1], c[2], c[3]);   // lane-quarter kq: channels 16 (ntile0 + kq) + 0..15 of pixel row `col`
            const int row = (wm * MT + i) * 16 + col;
            const int n = (ntile0 + kq) * 16;
            if (row < p.rb && m0 + row < p.M && n < p.N)
                *reinterpret_cast<u32x4c*>(yb + (size_t)(m0 + row) * p.N + n) = u32x4c{c[0], c[1], c[2], c[3]};
        }
    } else {
        // float32 out: staged through a per-wave LDS area into 256-byte runs (4 rows x the wave's 64 channels per store)
        unsigned char* stg = xs + 2 * XBYTES + wave * (16 * kStgRow);
        const uint64_t left = (uint64_t)(p.M - m0) * p.N * 4;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(p.y) + (size_t)m0 * p.N, 0,
                                                                             (uint32_t)(left > 0xFFFFFFFFull ? 0xFFFFFFFFull : left), 0x00020000);
        const int srow = lane >> 4, sch = lane & 15;
        const int n_st = ntile0 * 16 + sch * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j) *reinterpret_cast<float4*>(stg + col * kStgRow + j * 64 + kq * 16) = out_tile(i, j);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int row = (wm * MT + i) * 16 + h * 4 + srow;
                const u32x4c v = *reinterpret_cast<const u32x4c*>(stg + (h * 4 + srow) * kStgRow + sch * 16);
                const bool ok = row < p.rb && m0 + row < p.M && n_st < p.N;
                uint32_t so = ok ? (uint32_t)(row * p.N + n_st) * 4u : 0xFFFFFFF0u;
                asm volatile("" : "+v"(so));
                __builtin_amdgcn_raw_buffer_store_b128(v, ry, so, 0, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------- launch
static bool pwc_stream_fits(const ConvPlan& plan) { return pointwise_stream_fits(plan.k_pad, plan.n_pad, 1); }

bool pwc_applicable(const slfp_conv2d_desc& d, const ConvPlan& plan, int post_flags, bool y_codes) {
    if (plan.family != kPointwise || plan.repad || plan.passes != 1) return false;
    if (post_flags & SLFP_POST_LAYEROUT) return false;
    if (d.c_in % 32 != 0) return false;
    if (y_codes ? (d.c_out % 16 != 0) : (d.c_out % 4 != 0)) return false;
    if (pwc_stream_fits(plan)) {
        const int ks = (int)(d.c_in / 32);
        return ks == 1 || ks == 2 || ks == 4 || ks == 8;
    }
    return d.c_in % 64 == 0;
}

template <int FMT, int KS, bool XW>
static int launch_pwc_stream(PwcParams& p, bool y_codes, hipStream_t stream) {
    const size_t lds = (size_t)p.n_tiles * p.KS * 1024 + (size_t)3 * p.n_tiles * 16 * sizeof(float);
    const size_t lds_total = lds + 1024 + (y_codes ? kPwTab : 16);
    auto fn = y_codes ? k_pwc_stream<FMT, KS, XW, true> : k_pwc_stream<FMT, KS, XW, false>;
    int rc = raise_lds_limit(reinterpret_cast<const void*>(fn), lds);
    if (rc != SLFP_OK) return rc;
    int per_cu = resident_blocks_per_cu(reinterpret_cast<const void*>(fn), kPwcThreads, lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    (void)lds_total;
    const int64_t groups = (p.M + 15) / 16;
    int64_t grid = (int64_t)device_cu_count() * per_cu;
    const int64_t need = ceil_div(groups, kPwcThreads / 64);
    if (grid > need) grid = need;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kPwcThreads), lds, stream, p);
    return check_launch("slfp pointwise (codes, stream) kernel");
}

template <int FMT, int WM, int WN, int MT>
static int launch_pwc_tiled(PwcParams& p, bool y_codes, hipStream_t stream) {
    constexpr int BM = WM * MT * 16, BN = WN * 4 * 16, T = 64 * WM * WN;
    p.n_blocks = (uint32_t)ceil_div((int64_t)p.N, BN);
    p.rb = BM;
    p.m_blocks = (uint32_t)ceil_div(p.M, p.rb);
    const int64_t nblocks = (int64_t)p.m_blocks * p.n_blocks;
    if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "pointwise (codes): grid too large");
    p.nblocks = (uint32_t)nblocks;
    const size_t lds = (size_t)2 * BM * 128 + (y_codes ? 0 : (size_t)(T / 64) * 16 * kStgRow);
    auto fn = y_codes ? k_pwc_tiled<FMT, WM, WN, MT, true> : k_pwc_tiled<FMT, WM, WN, MT, false>;
    int rc = raise_lds_limit(reinterpret_cast<const void*>(fn), lds);
    if (rc != SLFP_OK) return rc;
    hipLaunchKernelGGL(fn, dim3(p.nblocks), dim3(T), lds, stream, p);
    return check_launch("slfp pointwise (codes, tiled) kernel");
}

// slice kernel: K a multiple of 256, the slice's W (NTS tiles x K) <= 128 KiB, N a multiple of 16 * NTS
static int pwc_slice_nts(const PwcParams& p) {
    if (p.K != 256 || !switches().pwc_slice) return 0;   // see k_pwc_slice: deeper layers stay on k_pwc_tiled
    for (int nts : {8, 4}) {
        if ((size_t)nts * 16 * p.K * 2 <= 128 * 1024 && p.N % (nts * 16) == 0 && p.n_tiles % nts == 0) return nts;
    }
    return 0;
}

template <int FMT, int NTS>
static int launch_pwc_slice(PwcParams& p, bool y_codes, hipStream_t stream) {
    const size_t lds = (size_t)NTS * p.KS * 1024 + (size_t)3 * NTS * 16 * sizeof(float);
    auto fn = y_codes ? k_pwc_slice<FMT, NTS, true> : k_pwc_slice<FMT, NTS, false>;
    int rc = raise_lds_limit(reinterpret_cast<const void*>(fn), lds);
    if (rc != SLFP_OK) return rc;
    int per_cu = resident_blocks_per_cu(reinterpret_cast<const void*>(fn), kSliceThreads, lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    const int n_slices = p.n_tiles / NTS;
    // grid = 8 (XCDs) x n_slices x ranks-per-XCD: as many whole (slice set) groups as fit the device
    const int64_t slots = (int64_t)device_cu_count() * per_cu;
    int64_t groups = slots / (8 * n_slices);
    if (groups < 1) groups = 1;
    const int64_t units = (p.M + 15) / 16;
    const int64_t need = ceil_div(units, (int64_t)8 * (kSliceThreads / 64));   // ranks needed, in multiples of 8 (one per XCD)
    if (groups > need) groups = need < 1 ? 1 : need;
    const int64_t grid = groups * 8 * n_slices;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kSliceThreads), lds, stream, p);
    return check_launch("slfp pointwise (codes, slice) kernel");
}

template <int FMT>
static int launch_pwc_fmt(PwcParams& p, const ConvPlan& plan, bool y_codes, hipStream_t stream) {
    if (const int nts = pwc_slice_nts(p)) {
        if (nts == 8) return launch_pwc_slice<FMT, 8>(p, y_codes, stream);
        return launch_pwc_slice<FMT, 4>(p, y_codes, stream);
    }
    if (pwc_stream_fits(plan)) {
        switch (p.K / 32) {
            case 1: return launch_pwc_stream<FMT, 1, false>(p, y_codes, stream);
            case 2: return launch_pwc_stream<FMT, 2, true>(p, y_codes, stream);
            case 4: return launch_pwc_stream<FMT, 4, true>(p, y_codes, stream);
            case 8: return launch_pwc_stream<FMT, 8, true>(p, y_codes, stream);
            default: return fail(SLFP_ERR_UNSUPPORTED, "pointwise (codes): unsupported channel count %d", p.K);
        }
    }
    if (p.N > 256) return launch_pwc_tiled<FMT, 1, 8, 4>(p, y_codes, stream);   // 64 px x 512 ch
    if (p.N > 128) return launch_pwc_tiled<FMT, 1, 4, 4>(p, y_codes, stream);   // 64 px x 256 ch
    if (p.N > 64) return launch_pwc_tiled<FMT, 2, 2, 2>(p, y_codes, stream);    // 64 px x 128 ch
    return launch_pwc_tiled<FMT, 4, 1, 1>(p, y_codes, stream);                   // 64 px x  64 ch
}

int launch_pwc(const slfp_conv2d_desc& d, const ConvPlan& plan, const uint8_t* x, const void* wfrag, const float* bias,
               const PostOp& post, void* y, bool y_codes, float y_ka, int y_fmt, hipStream_t stream) {
    PwcParams p;
    p.post = post;
    p.x = x; p.bias = bias; p.y = y;
    p.K = (int)d.c_in; p.N = (int)d.c_out;
    p.KS = (int)(plan.k_pad / 32);
    p.n_tiles = (int)(plan.n_pad / 16);
    p.whi = reinterpret_cast<const _Float16*>(wfrag);
    p.H = (int)d.h; p.W = (int)d.w; p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out; p.S = d.stride_h;
    p.M = d.n * plan.h_out * plan.w_out;
    p.s1 = plan.s1; p.s2 = plan.s2; p.s1x = plan.s1 * (1.0f / 256.0f);
    p.sgn = post.relu ? 0 : 1;
    p.fmt_out = y_fmt;
    p.enc.valid = 0;
    if (y_codes) {
        const EncArgs* t = enc_table(y_ka, y_fmt, kEncCode);
        if (!t->valid) return fail(SLFP_ERR_UNSUPPORTED, "pointwise (codes): no code table for the consumer's scale %g", (double)y_ka);
        p.enc = *t;
    }
    if (plan.fmt_act == kFmtSfp7) return launch_pwc_fmt<kFmtSfp7>(p, plan, y_codes, stream);
    return launch_pwc_fmt<kFmtAct8>(p, plan, y_codes, stream);
}

}  // namespace slfp
