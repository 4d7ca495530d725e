// conv_pw.hip -- SLFP-quantized pointwise (1x1) convolution on the gfx950 matrix cores.
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) for 1x1 kernels,
// groups == 1 (the 13 "pw" layers of MobileNetV1, nets_imgnet/mobilenetv1.py:31; the
// 36 1x1 layers of ResNet-50; Fire squeeze/expand1x1 of SqueezeNet).
//
// NHWC makes this a plain GEMM  Y^T[N x M] = Wq[N x K] . Xq^T[K x M]  (M = pixels) with
// K contiguous on both operands.  58 flop/B at float32 would be compute-bound on gfx950
// (ridge 19.6 flop/B), so the contraction runs on v_mfma_f32_16x16x32_f16 with float32
// accumulation and the kernel stays HBM-bound (4 B in + 4 B out per element):
//   * X: every workgroup streams its BM pixel rows once from HBM (16-byte loads, a
//     pixel's K channels are contiguous), applies x/Ka + the SLFP encode inline, converts
//     the dequantized value to fp16 (hi [+ lo]) and stages it in a swizzled LDS tile that
//     all waves of the workgroup share (encode happens once per element);
//   * W: quantized ONCE per weight update into an MFMA-fragment-ordered fp16 blob
//     (slfp_conv2d_prepare_weights); each wave streams the fragments of its own output
//     channels straight from L2 into registers (1 KiB fully coalesced per fragment);
//   * A operand = W (rows = output channels), B operand = X (columns = pixels), so each
//     lane ends up holding 4 CONSECUTIVE output channels of one pixel = one 16-byte
//     NHWC store.
// Operand precision: SFP<3,3> values are exact in fp16 (1 pass, exact products).
// SLFP<3,4> values are 2^(m/16) multiples; fp16x1 rounds them to 11 bits (~2.5e-4
// tensor-relative error), fp16x3 splits both operands hi+lo (3 MFMA passes,
// float32-equivalent).  Both operands are pre-scaled by 2^4 (range [2, 245]) so hi is
// always a normal fp16 and lo keeps 2^-26 relative precision; the epilogue undoes 2^-8
// exactly before the reference's (out * Ka) * Kw roundings.
#include "slfp_device.hpp"
#include "slfp_host.hpp"

namespace slfp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct PwParams {
    const float* x;
    const _Float16* whi;
    const _Float16* wlo;
    const float* bias;
    float* y;
    int64_t M;        // output pixels = N_img * Ho * Wo
    int K, N;         // input / output channels
    int KS;           // number of 32-deep k-steps in the blob (even)
    int n_tiles;      // 16-row tiles in the blob (n_pad / 16)
    int H, W, Ho, Wo, S;  // strided 1x1: input pixel = (oh*S, ow*S)
    float ka, s1, s2;     // x/ka; out = ((acc + bias/s1/s2) * s1) * s2
    uint32_t m_blocks, n_blocks, nblocks;
};

__device__ __forceinline__ uint32_t lds_x_off(int row, int chunk16) {
    // 128-byte rows (64 fp16); XOR swizzle so that the 16 rows a fragment read touches hit
    // 16 distinct 16-byte slots (conflict-free ds_read_b128; cdna guide T2)
    return (uint32_t)row * 128u + (uint32_t)((chunk16 ^ (row & 7)) << 4);
}

template <int FMT, int PASSES, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN) void k_pw(const PwParams p) {
    constexpr int T = 64 * WM * WN;
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int NLD = BM * 16 / T;  // float4 loads per thread per 64-deep stage
    static_assert(BM * 16 % T == 0, "staging must divide evenly");
    constexpr int XBYTES = BM * 128;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sT = reinterpret_cast<uint32_t*>(smem);
    unsigned char* xs = smem + 64;  // [2 buffers][hi, lo][BM rows][128 B]
    lut_fill(sT);

    const uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const uint32_t nb = b % p.n_blocks, mb = b / p.n_blocks;
    const int64_t m0 = (int64_t)mb * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, kq = lane >> 4;

    // ---- per-thread staging geometry: float4 #i covers row (tid>>4) + i*(T/16), k = (tid&15)*4
    const int kc = threadIdx.x & 15;
    uint32_t src_off[NLD];  // element offset of the row start in x, or 0xFFFFFFFF
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = (threadIdx.x >> 4) + i * (T / 16);
        const int64_t m = m0 + row;
        uint32_t off = 0xFFFFFFFFu;
        if (m < p.M) {
            if (p.S == 1) {
                off = (uint32_t)(m * p.K);
            } else {
                const int64_t hw = (int64_t)p.Ho * p.Wo;
                const int64_t img = m / hw, r = m % hw;
                const int oh = (int)(r / p.Wo), ow = (int)(r % p.Wo);
                off = (uint32_t)((((img * p.H) + (int64_t)oh * p.S) * p.W + (int64_t)ow * p.S) * p.K);
            }
        }
        src_off[i] = off;
    }

    float4 st[NLD];
    auto load_stage = [&](int t) {
        const int k = t * 64 + kc * 4;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src_off[i] != 0xFFFFFFFFu && k < p.K)
                st[i] = *reinterpret_cast<const float4*>(p.x + (size_t)src_off[i] + k);
        }
    };
    auto encode_store = [&](int buf) {
        unsigned char* hi = xs + (size_t)buf * (PASSES == 3 ? 2 : 1) * XBYTES;
        unsigned char* lo = hi + XBYTES;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = (threadIdx.x >> 4) + i * (T / 16);
            const float v0 = 16.0f * quantize_scaled<FMT>(st[i].x, p.ka, sT);
            const float v1 = 16.0f * quantize_scaled<FMT>(st[i].y, p.ka, sT);
            const float v2 = 16.0f * quantize_scaled<FMT>(st[i].z, p.ka, sT);
            const float v3 = 16.0f * quantize_scaled<FMT>(st[i].w, p.ka, sT);
            half4 h;
            h[0] = (_Float16)v0; h[1] = (_Float16)v1; h[2] = (_Float16)v2; h[3] = (_Float16)v3;
            const uint32_t off = lds_x_off(row, kc >> 1) + (uint32_t)(kc & 1) * 8u;
            *reinterpret_cast<half4*>(hi + off) = h;
            if constexpr (PASSES == 3) {
                half4 l;
                l[0] = (_Float16)(v0 - (float)h[0]); l[1] = (_Float16)(v1 - (float)h[1]);
                l[2] = (_Float16)(v2 - (float)h[2]); l[3] = (_Float16)(v3 - (float)h[3]);
                *reinterpret_cast<half4*>(lo + off) = l;
            }
        }
    };

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int KT = p.KS >> 1;  // 64-deep stages
    const int ntile0 = (int)nb * (BN / 16) + wn * NT;
    const bool wave_live = ntile0 < p.n_tiles;  // wave-uniform

    __syncthreads();  // LUT visible
    load_stage(0);
    encode_store(0);
    if (KT > 1) load_stage(1);
    __syncthreads();

    for (int t = 0; t < KT; ++t) {
        const int buf = t & 1;
        half8 wh[NT], wl[NT];
        auto load_w = [&](int ks) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nt = ntile0 + j;
                if (nt < p.n_tiles) {
                    const size_t o = ((size_t)nt * p.KS + (size_t)(t * 2 + ks)) * 512 + (size_t)lane * 8;
                    wh[j] = *reinterpret_cast<const half8*>(p.whi + o);
                    if constexpr (PASSES == 3) wl[j] = *reinterpret_cast<const half8*>(p.wlo + o);
                } else {
                    wh[j] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    if constexpr (PASSES == 3) wl[j] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
        };
        if (wave_live) load_w(0);
        // stage t+1: encode the rows fetched one stage ago into the other buffer, then
        // put stage t+2's loads in flight (they land while this stage's MFMAs run)
        if (t + 1 < KT) {
            encode_store(buf ^ 1);
            if (t + 2 < KT) load_stage(t + 2);
        }
        if (wave_live) {
            const unsigned char* hi = xs + (size_t)buf * (PASSES == 3 ? 2 : 1) * XBYTES;
            const unsigned char* lo = hi + XBYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 1) load_w(1);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = (wm * MT + i) * 16 + col;
                    const uint32_t off = lds_x_off(row, ks * 4 + kq);
                    const half8 xh = *reinterpret_cast<const half8*>(hi + off);
                    half8 xl;
                    if constexpr (PASSES == 3) xl = *reinterpret_cast<const half8*>(lo + off);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (PASSES == 3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl, acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: undo the 2^8 operand pre-scale (exact), bias, (out*s1)*s2, 16-byte stores
    if (!wave_live) return;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (ntile0 + j) * 16 + kq * 4;
        if (n >= p.N) continue;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
            bq = make_float4((bb.x / p.s1) / p.s2, (bb.y / p.s1) / p.s2, (bb.z / p.s1) / p.s2, (bb.w / p.s1) / p.s2);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int64_t m = m0 + (wm * MT + i) * 16 + col;
            if (m >= p.M) continue;
            float4 r;
            r.x = ((acc[i][j][0] * 0.00390625f + bq.x) * p.s1) * p.s2;
            r.y = ((acc[i][j][1] * 0.00390625f + bq.y) * p.s1) * p.s2;
            r.z = ((acc[i][j][2] * 0.00390625f + bq.z) * p.s1) * p.s2;
            r.w = ((acc[i][j][3] * 0.00390625f + bq.w) * p.s1) * p.s2;
            *reinterpret_cast<float4*>(p.y + (size_t)m * p.N + n) = r;
        }
    }
}

template <int FMT, int PASSES, int WM, int WN, int MT, int NT>
static int launch_cfg(PwParams& p, hipStream_t stream) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, T = 64 * WM * WN;
    p.m_blocks = (uint32_t)ceil_div(p.M, BM);
    p.n_blocks = (uint32_t)ceil_div((int64_t)p.N, BN);
    const int64_t nblocks = (int64_t)p.m_blocks * p.n_blocks;
    if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "pointwise: grid too large");
    p.nblocks = (uint32_t)nblocks;
    const size_t lds = 64 + (size_t)2 * (PASSES == 3 ? 2 : 1) * BM * 128;
    hipLaunchKernelGGL((k_pw<FMT, PASSES, WM, WN, MT, NT>), dim3(p.nblocks), dim3(T), lds, stream, p);
    return check_launch("slfp pointwise kernel");
}

template <int FMT, int PASSES>
static int launch_by_n(PwParams& p, hipStream_t stream) {
    if (p.N > 256) return launch_cfg<FMT, PASSES, 1, 8, 8, 4>(p, stream);   // 128 px x 512 ch, 8 waves
    if (p.N > 128) return launch_cfg<FMT, PASSES, 1, 4, 8, 4>(p, stream);   // 128 px x 256 ch
    if (p.N > 64) return launch_cfg<FMT, PASSES, 2, 2, 4, 4>(p, stream);    // 128 px x 128 ch
    return launch_cfg<FMT, PASSES, 4, 1, 2, 4>(p, stream);                   // 128 px x  64 ch
}

int launch_pointwise(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const void* wfrag,
                     const float* bias, float* y, hipStream_t stream) {
    PwParams p;
    p.x = x; p.bias = bias; p.y = y;
    p.K = (int)d.c_in; p.N = (int)d.c_out;
    p.KS = (int)(plan.k_pad / 32);
    p.n_tiles = (int)(plan.n_pad / 16);
    p.whi = reinterpret_cast<const _Float16*>(wfrag);
    p.wlo = p.whi + (size_t)plan.n_pad * plan.k_pad;
    p.H = (int)d.h; p.W = (int)d.w; p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out; p.S = d.stride_h;
    p.M = d.n * plan.h_out * plan.w_out;
    p.ka = d.ka; p.s1 = plan.s1; p.s2 = plan.s2;
    if ((uint64_t)d.n * d.h * d.w * d.c_in >= 0xFFFFFFFFull)
        return fail(SLFP_ERR_UNSUPPORTED, "pointwise: input larger than 2^32 elements");
    if (plan.fmt_act == kFmtSfp7) return launch_by_n<kFmtSfp7, 1>(p, stream);  // exact in fp16
    if (plan.passes == 3) return launch_by_n<kFmtAct8, 3>(p, stream);
    return launch_by_n<kFmtAct8, 1>(p, stream);
}

}  // namespace slfp
