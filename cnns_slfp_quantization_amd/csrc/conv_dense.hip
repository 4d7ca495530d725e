// conv_dense.hip -- SLFP-quantized dense k x k convolution as an implicit GEMM on the gfx950
// matrix cores (VGG-16's 3x3 layers, ResNet-50's 3x3 layers, SqueezeNet's expand3x3, AlexNet 5x5).
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) for groups == 1, KH*KW > 1.
// These layers are compute-bound (VGG-16: 337 flop/B), so the contraction must run on MFMA; the
// structure is the pointwise GEMM of conv_pw.hip with the K dimension = (tap, input channel):
//   * workgroup = TH x 16 output pixels x 256 output channels, 8 waves (2 along rows x 4 along
//     channels), accumulators MT x 4 tiles of 16x16 per wave;
//   * per 64-channel chunk the (TH-1)*S+KH x 15*S+KW input HALO tile is read once from HBM,
//     x/Ka + SLFP encode applied inline, and stored as fp16 in a swizzled LDS tile: every one of
//     the KH*KW taps then reads its shifted 16-pixel fragments from the same tile (the encode
//     is amortised over KH*KW * 256 MACs per element);
//   * per tap the 256 x 64 fp16 weight tile (fragment-ordered blob, tap-major) is staged in a
//     double-buffered LDS tile by all waves (each fragment is used by both row-waves and by MT
//     pixel tiles), one barrier per tap;
//   * fp16 operands (both pre-scaled by 2^4, see conv_pw.hip), float32 accumulation, the
//     reference's (out*Ka)*Kw roundings and the optional fused BN/ReLU post-op in the epilogue.
// Single-pass fp16 (SLFP<3,4>: ~2.5e-4 tensor-relative, the north-star 1e-3 bar) or exact
// (SFP<3,3>).  The float32-equivalent mode of these layers stays on k_direct.
#include "slfp_device.hpp"
#include "slfp_host.hpp"

namespace slfp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kDnThreads = 512;
constexpr int kDnTW = 16;            // output columns per tile = one MFMA pixel tile
constexpr int kDnBN = 256;           // output channels per workgroup
constexpr int kDnWBytes = kDnBN * 64 * 2;  // one tap's weight tile: 256 ch x 64 cin fp16 = 32 KiB

struct DenseParams {
    const float* x;
    const _Float16* w;   // [tap][n_tile][k_step][64 lanes][8]
    const float* bias;
    float* y;
    int N, H, W, C, O, KH, KW, S, ph, pw, Ho, Wo;
    int tiles_h, tiles_w, n_blocks;
    int IH, IW, n_pix;   // halo tile
    int KS;              // 32-deep k-steps per tap (c_pad / 32), even
    int n_tiles;         // 16-channel tiles in the blob (n_pad / 16)
    int x_items_per_thread;  // ceil(n_pix * 16 float4 / 512 threads) per chunk
    ScaleDiv sd;
    float s1, s2, s1x;
    PostOp post;
    uint32_t nblocks;
};

__device__ __forceinline__ uint32_t dn_x_off(int row, int chunk16) {
    // 128-byte rows: a 16-lane fragment read touches 16 consecutive rows; (row & 1) picks the
    // 128-byte half of the 256-byte bank line, (row >> 1) & 7 rotates the 16-byte slot -> 16 distinct slots
    return (uint32_t)row * 128u + (uint32_t)((chunk16 ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ void glds16(const void* g, void* l) {  // 64 lanes x 16 B -> 1 KiB of LDS at (wave-uniform) l
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// MT = output rows per row-wave (TH = 2 * MT).  PER_TAP = halo float4 each thread prefetches per tap.
template <int FMT, int MT, int PER_TAP>
__global__ __launch_bounds__(kDnThreads) void k_dense_mfma(const DenseParams p) {
    constexpr int TH = 2 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sT = reinterpret_cast<uint32_t*>(smem);
    unsigned char* wbuf = smem + 64;                  // [2][kDnWBytes]
    unsigned char* xsb = wbuf + 2 * kDnWBytes;        // [2][n_pix][128 B]
    const uint32_t xbytes = (uint32_t)p.n_pix * 128u;
    lut_fill<FMT>(sT);

    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);   // channel slice slowest: an XCD's L2 holds one W slice
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b % p.N;
    const int nb = b / p.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;          // 2 row-waves x 4 channel-waves
    const int col = lane & 15, kq = lane >> 4;
    const int h_in0 = th * TH * p.S - p.ph, w_in0 = tw * kDnTW * p.S - p.pw;

    // ---- halo staging: float4 item #i of a thread = (pixel (tid >> 4) + 32 i, channel quad kc)
    const int kc = threadIdx.x & 15;
    const uint32_t st_sub = (uint32_t)((kc & 7) >> 2) * 8u;
    const int st_chunk = (kc >> 3) * 4 + (kc & 3);
    const float* xn = p.x + (size_t)n * p.H * p.W * p.C;
    auto halo_load = [&](int item, int chunk, float4& v, uint32_t& dst) {
        const int pix = (threadIdx.x >> 4) + item * (kDnThreads / 16);
        const int ih = pix / p.IW, iw = pix - ih * p.IW;
        const int gh = h_in0 + ih, gw = w_in0 + iw;
        const bool live = pix < p.n_pix;
        const int k = chunk * 64 + kc * 4;
        const bool inb = live && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W && k < p.C;
        // unconditional load (clamped address), zeroed where the conv pads / past C_in
        v = *reinterpret_cast<const float4*>(xn + (inb ? (gh * p.W + gw) * p.C + k : 0));
        if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);
        dst = live ? dn_x_off(pix, st_chunk) + st_sub : 0xFFFFFFFFu;
    };
    auto halo_store = [&](const float4& v, uint32_t dst, unsigned char* xs) {
        if (dst == 0xFFFFFFFFu) return;
        half4 h;
        h[0] = (_Float16)quantize_scaled<FMT, 4>(v.x, p.sd, sT);
        h[1] = (_Float16)quantize_scaled<FMT, 4>(v.y, p.sd, sT);
        h[2] = (_Float16)quantize_scaled<FMT, 4>(v.z, p.sd, sT);
        h[3] = (_Float16)quantize_scaled<FMT, 4>(v.w, p.sd, sT);
        *reinterpret_cast<half4*>(xs + dst) = h;
    };

    // ---- weight tap tile: 32 pieces of 1 KiB (channel tile, k-step), 4 per wave, LDS-DMA
    const int nt0 = nb * (kDnBN / 16);
    auto stage_w = [&](int tap, int chunk, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pc = wave * 4 + j;                      // piece = (channel tile pc >> 1, k-step pc & 1)
            int nt = nt0 + (pc >> 1);
            nt = nt < p.n_tiles ? nt : p.n_tiles - 1;         // tiles past C_out: clamp (results never stored)
            const size_t o = (((size_t)tap * p.n_tiles + nt) * p.KS + (size_t)chunk * 2 + (pc & 1)) * 1024 + (size_t)lane * 16;
            glds16(reinterpret_cast<const unsigned char*>(p.w) + o, wbuf + (size_t)buf * kDnWBytes + (size_t)pc * 1024);
        }
    };

    floatx4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int n_chunks = p.KS >> 1, n_taps = p.KH * p.KW;
    stage_w(0, 0, 0);
    __syncthreads();  // LUT visible
    for (int item = 0; item < p.x_items_per_thread; ++item) {  // chunk 0's halo (the only exposed HBM latency)
        float4 v; uint32_t dst;
        halo_load(item, 0, v, dst);
        halo_store(v, dst, xsb);
    }
    __syncthreads();

    int wb = 0, xb = 0;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const bool more_chunks = chunk + 1 < n_chunks;
        const unsigned char* xs = xsb + (size_t)xb * xbytes;
        unsigned char* xs_next = xsb + (size_t)(xb ^ 1) * xbytes;
        for (int tap = 0; tap < n_taps; ++tap) {
            const bool last_tap = tap + 1 == n_taps;
            // next weight tile (next tap, or tap 0 of the next chunk): LDS-DMA into the other buffer
            if (!last_tap || more_chunks) stage_w(last_tap ? 0 : tap + 1, last_tap ? chunk + 1 : chunk, wb ^ 1);
            // a slice of the next chunk's halo flies behind this tap's MFMAs
            float4 xv[PER_TAP]; uint32_t xd[PER_TAP];
#pragma unroll
            for (int q = 0; q < PER_TAP; ++q) {
                xd[q] = 0xFFFFFFFFu;
                xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int item = tap * PER_TAP + q;
                if (more_chunks && item < p.x_items_per_thread) halo_load(item, chunk + 1, xv[q], xd[q]);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the MFMA block

            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const unsigned char* wt = wbuf + (size_t)wb * kDnWBytes + (size_t)(wn * 4) * 2048 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 wf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const half8*>(wt + j * 2048 + ks * 1024);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = ((wm * MT + i) * p.S + kh) * p.IW + col * p.S + kw;
                    const half8 xf = *reinterpret_cast<const half8*>(xs + dn_x_off(row, ks * 4 + kq));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf, acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PER_TAP; ++q) halo_store(xv[q], xd[q], xs_next);
            __syncthreads();  // drains the LDS-DMA (vmcnt(0)); every wave is done with wbuf[wb] (and, on the last tap, xs)
            wb ^= 1;
        }
        xb ^= 1;
    }

    // ---- epilogue
    const int gow = tw * kDnTW + col;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = (nt0 + wn * 4 + j) * 16 + kq * 4;
        if (ch >= p.O) continue;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            const float4 bb = *reinterpret_cast<const float4*>(p.bias + ch);
            bq = make_float4(256.f * ((bb.x / p.s1) / p.s2), 256.f * ((bb.y / p.s1) / p.s2),
                             256.f * ((bb.z / p.s1) / p.s2), 256.f * ((bb.w / p.s1) / p.s2));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int goh = th * TH + wm * MT + i;
            if (goh >= p.Ho || gow >= p.Wo) continue;
            float4 r;
            r.x = ((acc[i][j][0] + bq.x) * p.s1x) * p.s2;
            r.y = ((acc[i][j][1] + bq.y) * p.s1x) * p.s2;
            r.z = ((acc[i][j][2] + bq.z) * p.s1x) * p.s2;
            r.w = ((acc[i][j][3] + bq.w) * p.s1x) * p.s2;
            *reinterpret_cast<float4*>(p.y + (((size_t)n * p.Ho + goh) * p.Wo + gow) * p.O + ch) = post_apply(r, p.post, ch);
        }
    }
}

// geometry helper shared with make_plan(): tile height and halo size for this descriptor
static bool dense_geometry(const slfp_conv2d_desc& d, int* mt, int* ih, int* iw) {
    const int S = d.stride_h;
    for (int m : {4, 2}) {
        const int th = 2 * m;
        const int IH = (th - 1) * S + (int)d.kh, IW = (kDnTW - 1) * S + (int)d.kw;
        const int items = (IH * IW * 16 + kDnThreads - 1) / kDnThreads;  // halo float4 per thread per chunk
        if ((size_t)IH * IW * 128 <= 44 * 1024 && items <= 3 * (int)(d.kh * d.kw)) {
            *mt = m; *ih = IH; *iw = IW;
            return true;
        }
    }
    return false;
}

bool dense_mfma_applicable(const slfp_conv2d_desc& d, int passes) {
    if (d.groups != 1 || d.kh * d.kw <= 1 || d.dil_h != 1 || d.dil_w != 1) return false;
    if (d.stride_h != d.stride_w || d.stride_h > 2) return false;
    if (d.c_in % 4 || d.c_in < 16 || d.c_out % 4) return false;
    if (d.qbits == 8 && passes == 3) return false;  // the float32-equivalent mode stays on k_direct
    if ((int64_t)d.h * d.w * d.c_in >= (1ll << 30)) return false;
    int mt, ih, iw;
    return dense_geometry(d, &mt, &ih, &iw);
}

template <int FMT, int MT, int PER_TAP>
static int launch_dense_t(DenseParams& p, hipStream_t stream) {
    const size_t lds = 64 + 2 * (size_t)kDnWBytes + 2 * (size_t)p.n_pix * 128;
    auto fn = k_dense_mfma<FMT, MT, PER_TAP>;
    static bool lds_raised = false;  // > 64 KiB of dynamic LDS needs the opt-in once per kernel
    if (!lds_raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return check_launch("hipFuncSetAttribute(dense)");
        lds_raised = true;
    }
    hipLaunchKernelGGL(fn, dim3(p.nblocks), dim3(kDnThreads), lds, stream, p);
    return check_launch("slfp dense MFMA conv kernel");
}

int launch_dense_mfma(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const void* wblob,
                      const float* bias, const PostOp& post, float* y, hipStream_t stream) {
    DenseParams p;
    int mt;
    if (!dense_geometry(d, &mt, &p.IH, &p.IW)) return fail(SLFP_ERR_UNSUPPORTED, "dense MFMA conv: halo tile too large");
    p.x = x; p.w = reinterpret_cast<const _Float16*>(wblob); p.bias = bias; p.y = y; p.post = post;
    p.N = (int)d.n; p.H = (int)d.h; p.W = (int)d.w; p.C = (int)d.c_in; p.O = (int)d.c_out;
    p.KH = (int)d.kh; p.KW = (int)d.kw; p.S = d.stride_h; p.ph = d.pad_h; p.pw = d.pad_w;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out;
    p.tiles_h = (int)ceil_div(p.Ho, 2 * mt); p.tiles_w = (int)ceil_div(p.Wo, kDnTW);
    p.n_blocks = (int)ceil_div((int64_t)p.O, kDnBN);
    p.n_pix = p.IH * p.IW;
    p.KS = (int)(plan.k_pad / 32); p.n_tiles = (int)(plan.n_pad / 16);
    p.x_items_per_thread = (int)ceil_div((int64_t)p.n_pix * 16, kDnThreads);
    p.sd = make_scale_div(d.ka, 4);
    p.s1 = plan.s1; p.s2 = plan.s2; p.s1x = plan.s1 * (1.0f / 256.0f);
    const int64_t nblocks = (int64_t)p.N * p.tiles_h * p.tiles_w * p.n_blocks;
    if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "dense MFMA conv: grid too large");
    p.nblocks = (uint32_t)nblocks;
    const int per_tap = (int)ceil_div(p.x_items_per_thread, (int64_t)p.KH * p.KW);  // 1..3, see dense_geometry()
    const bool a8 = plan.fmt_act == kFmtAct8;
#define SLFP_DN(MTT, PT) (a8 ? launch_dense_t<kFmtAct8, MTT, PT>(p, stream) : launch_dense_t<kFmtSfp7, MTT, PT>(p, stream))
    if (mt == 4) return per_tap == 1 ? SLFP_DN(4, 1) : (per_tap == 2 ? SLFP_DN(4, 2) : SLFP_DN(4, 3));
    return per_tap == 1 ? SLFP_DN(2, 1) : (per_tap == 2 ? SLFP_DN(2, 2) : SLFP_DN(2, 3));
#undef SLFP_DN
}

}  // namespace slfp
