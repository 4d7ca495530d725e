// conv_dense.hip -- SLFP-quantized dense k x k convolution as an implicit GEMM on the gfx950
// matrix cores (VGG-16's 3x3 layers, ResNet-50's 3x3 layers, SqueezeNet's expand3x3, AlexNet 5x5).
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) for groups == 1, KH*KW > 1.
// These layers are compute-bound (VGG-16: 337 flop/B), so the contraction runs on MFMA, in two passes:
//   1. k_dense_encode: x/Ka + SLFP encode applied ONCE per input element (fused into the GEMM it
//      would be redone for every overlapping halo and every output-channel slice: 3-5x, and the
//      encode's ~22 VALU ops per element then rival the MFMA time); the result, fp16(16 * QA(x/Ka)),
//      goes to the caller's workspace in B-fragment chunk order (6 B of HBM traffic per element
//      on layers that do >= 500 flop per element);
//   2. k_dense_mfma: the pointwise GEMM of conv_pw.hip with K = (tap, input channel):
//      * workgroup = TH x 16 output pixels x BN output channels, 8 waves as WM (rows) x WN
//        (channels), BN = 64 WN, TH = WM * MT; accumulators MT x 4 tiles of 16x16 per wave;
//      * per 64-channel chunk the (TH-1)*S+KH x 15*S+KW input HALO tile sits in a swizzled LDS
//        tile and every one of the KH*KW taps reads its shifted 16-pixel fragments from it; the
//        NEXT chunk's halo arrives in 1 KiB LDS-DMA pieces (global_load_lds, 8 pixels each, the
//        swizzle applied on the source address, conv padding served from a zero page) behind
//        the taps' MFMAs, into the second buffer;
//      * per tap the BN x 64 fp16 weight tile (fragment-ordered blob, tap-major) is brought in
//        by LDS-DMA into a double buffer; one barrier per tap; no VALU work beyond addresses;
//      * tilings with <= 80 KiB of LDS run two workgroups per CU (one's barrier / DMA wait is
//        the other's MFMA time); dense_choose() picks the tiling per layer;
//      * fp16 operands (both pre-scaled by 2^4, see conv_pw.hip), float32 accumulation, the
//        reference's (out*Ka)*Kw roundings and the optional fused BN/ReLU post-op in the epilogue.
// Single-pass fp16 (SLFP<3,4>: ~2.5e-4 tensor-relative, the north-star 1e-3 bar), exact (SFP<3,3>), or
// float32-equivalent (SLFP_MFMA_F16X3: hi + lo fp16 planes of both operands, 3 MFMAs per tile; stride-2
// layers, whose halo tiles do not fit twice in LDS, stay on k_direct in that mode).
#include "slfp_device.hpp"
#include "slfp_codes.hpp"
#include "slfp_host.hpp"

namespace slfp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kDnThreads = 512;
constexpr int kDnTW = 16;  // output columns per tile = one MFMA pixel tile
constexpr int kMaxSlots = 9;   // halo pieces a wave stages per chunk = ceil(x_pieces / 8); dense_cfg_geometry enforces it

struct DenseParams {
    const _Float16* xe;  // pre-encoded input, NHWC with C padded to Cp (k_dense_encode)
    const _Float16* xlo; // PASSES == 3: its fp16 residual plane
    const _Float16* wlo; // PASSES == 3: the weights' residual plane (same fragment order as w)
    const unsigned char* zero_page;  // 256 zero bytes: DMA source for padding pixels / channels past Cp
    const _Float16* w;   // [tap][n_tile][k_step][64 lanes][8]
    const float* bias;
    float* y;
    int N, H, W, Cp, O, KH, KW, S, ph, pw, Ho, Wo;
    int tiles_h, tiles_w, n_blocks;
    int IH, IW, n_pix;   // halo tile
    int KS;              // 32-deep k-steps per tap (c_pad / 32), even
    int n_tiles;         // 16-channel tiles in the blob (n_pad / 16)
    int x_pieces;        // 1 KiB DMA pieces (8 halo pixels) per chunk = ceil(n_pix / 8)
    int x_per_tap;       // pieces each wave issues per tap so that the next chunk is in by the last tap
    float s1, s2, s1x;
    PostOp post;
    uint32_t nblocks;
    // 1-byte codes out (slfp_codes.hpp): the epilogue applies the CONSUMER's quantizer QA(. / y_ka) and stores extended codes
    uint8_t* yc;          // != nullptr: code output (y is unused)
    int y_sgn;            // no ReLU in front of the output quantizer: codes carry a sign
    int y_fmt;            // kFmtAct8 | kFmtSfp7
    EncArgs enc_out;      // kEncCode table of (y_ka, y_fmt); filled into LDS after the main loop
    // k_dense3x3_res<ENCX>: the float32 input itself (C_in == 64) is encoded where the halo is staged -- no pre-pass, no fp16 copy
    const float* x32;
    EncArgsCompact enc_in;   // kEncF16P table of (Ka, fmt_act)
};

__device__ __forceinline__ uint32_t dn_x_off(int row, int chunk16) {
    // 128-byte rows; a fragment read touches 16 rows base .. base+15 for ANY base (taps shift it).
    // ds_read_b128 is served in lane groups {0-3,12-15,20-27}, ... = 8 lanes of an even lane-quarter
    // (rows +0..3, +12..15) and 8 of the next odd one (rows +4..11): bit 0 of the chunk (= parity of
    // the lane-quarter) is left alone so the two halves of a group never meet; bits 1-2 are rotated
    // by (row >> 1) & 3, which takes 4 distinct values on each half's 4 same-parity rows; (row & 1)
    // picks the 128-byte half of the 256-byte bank line.  16 distinct slots per group (profiles/
    // r01f_vgg16 measured 18 % conflict cycles with the plain (row >> 1) & 7 rotation).
    return (uint32_t)row * 128u + (uint32_t)((chunk16 ^ (((row >> 1) & 3) << 1)) << 4);
}

__device__ __forceinline__ void glds16(const void* g, void* l) {  // 64 lanes x 16 B -> 1 KiB of LDS at (wave-uniform) l
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Pre-pass: x (float32 NHWC) -> xe = fp16(16 * QA(x / Ka)), NHWC with C padded to a multiple of 64 (whole chunks)
// and, inside every 32-channel group, 16-byte chunk j = channels {4j..4j+3, 16+4j..16+4j+3}: exactly
// the 8 k-values lane-quarter j of a 16x16x32 MFMA B fragment holds.  One thread per chunk.
// lo != nullptr (float32-equivalent mode): also writes the fp16 residual v - fp32(fp16(v)) to a second plane.
template <int FMT>
__global__ __launch_bounds__(256) void k_dense_encode(const float* __restrict__ x, _Float16* __restrict__ xe,
                                                      _Float16* __restrict__ lo,
                                                      unsigned char* __restrict__ zero_page, int64_t n_chunks16,
                                                      int C, int Cp, const ScaleDiv sd) {
    __shared__ uint32_t sT[16];
    lut_fill<FMT>(sT);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 16) reinterpret_cast<uint4*>(zero_page)[threadIdx.x] = make_uint4(0, 0, 0, 0);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_chunks16) return;
    const int per_pix = Cp >> 3;
    const int64_t pix = idx / per_pix;
    const int cj = (int)(idx - pix * per_pix);
    const int c0 = (cj >> 2) * 32 + (cj & 3) * 4;
    const float* xp = x + pix * C;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c0 < C) a = ld_stream4<SLFP_NT_DENSE>(xp + c0);
    if (c0 + 16 < C) b = ld_stream4<SLFP_NT_DENSE>(xp + c0 + 16);
    float v[8] = {quantize_scaled<FMT, 4>(a.x, sd, sT), quantize_scaled<FMT, 4>(a.y, sd, sT),
                  quantize_scaled<FMT, 4>(a.z, sd, sT), quantize_scaled<FMT, 4>(a.w, sd, sT),
                  quantize_scaled<FMT, 4>(b.x, sd, sT), quantize_scaled<FMT, 4>(b.y, sd, sT),
                  quantize_scaled<FMT, 4>(b.z, sd, sT), quantize_scaled<FMT, 4>(b.w, sd, sT)};
    half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool pad = c0 + (j >> 2) * 16 >= C;   // quantize(0) is +0 anyway; keep the pad exact
        h[j] = pad ? (_Float16)0.f : (_Float16)v[j];
        l[j] = pad ? (_Float16)0.f : (_Float16)(v[j] - (float)h[j]);
    }
    if (lo) *reinterpret_cast<half8*>(lo + idx * 8) = l;
    *reinterpret_cast<half8*>(xe + idx * 8) = h;
}

// The same copy from 1-byte codes (slfp_codes.hpp): the producer already applied this layer's QA(. / Ka), so a byte decodes to
// exactly the value k_dense_encode would have computed from the float32 tensor -- the GEMM that follows is bit-identical.
template <int FMT>
__global__ __launch_bounds__(256) void k_dense_decode(const uint8_t* __restrict__ x, _Float16* __restrict__ xe,
                                                      _Float16* __restrict__ lo, unsigned char* __restrict__ zero_page,
                                                      int64_t n_chunks16, int C, int Cp) {
    __shared__ __attribute__((aligned(16))) uint32_t sDec[256];
    dec_fill<FMT, kDecF32, 256>(sDec);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 16) reinterpret_cast<uint4*>(zero_page)[threadIdx.x] = make_uint4(0, 0, 0, 0);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_chunks16) return;
    const int per_pix = Cp >> 3;
    const int64_t pix = idx / per_pix;
    const int cj = (int)(idx - pix * per_pix);
    const int c0 = (cj >> 2) * 32 + (cj & 3) * 4;
    const uint8_t* xp = x + pix * C;
    const unsigned char* dt = reinterpret_cast<const unsigned char*>(sDec);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c0 < C) a = dec4_f32(*reinterpret_cast<const uint32_t*>(xp + c0), dt);
    if (c0 + 16 < C) b = dec4_f32(*reinterpret_cast<const uint32_t*>(xp + c0 + 16), dt);
    const float v[8] = {16.f * a.x, 16.f * a.y, 16.f * a.z, 16.f * a.w, 16.f * b.x, 16.f * b.y, 16.f * b.z, 16.f * b.w};
    half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool pad = c0 + (j >> 2) * 16 >= C;
        h[j] = pad ? (_Float16)0.f : (_Float16)v[j];
        l[j] = pad ? (_Float16)0.f : (_Float16)(v[j] - (float)h[j]);
    }
    if (lo) *reinterpret_cast<half8*>(lo + idx * 8) = l;
    *reinterpret_cast<half8*>(xe + idx * 8) = h;
}

// The per-channel constants of a lane's four channel tiles: quantized-domain bias and the post-op's scale / shift.  The
// persistent kernel loads them ONCE (inside its tile loop every one of these loads was followed by s_waitcnt vmcnt(0), which
// on gfx9 also waits for the tile's stores so far: four exposed round trips per tile).
struct EpiConsts { float4 bq[4]; PostVec pv[4]; };
__device__ __forceinline__ void dense_epilogue_consts(const DenseParams& p, int wn, int nt0, int kq, EpiConsts& k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = (nt0 + wn * 4 + j) * 16 + kq * 4;
        k.bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        k.pv[j].sc = make_float4(1.f, 1.f, 1.f, 1.f);
        k.pv[j].sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch < p.O) {
            if (p.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + ch);
                k.bq[j] = make_float4(256.f * ((bb.x / p.s1) / p.s2), 256.f * ((bb.y / p.s1) / p.s2),
                                      256.f * ((bb.z / p.s1) / p.s2), 256.f * ((bb.w / p.s1) / p.s2));
            }
            k.pv[j] = post_load(p.post, ch);
        }
    }
}

// TAB_READY: the code table already sits at `smem` (k_dense3x3_res keeps it next to its operand tiles); otherwise it is filled
// over the dead operand tiles here.
template <int MT, bool TAB_READY = false>
__device__ __forceinline__ void dense_epilogue_apply(const DenseParams& p, const floatx4 (&acc)[MT][4], unsigned char* smem, int n, int th,
                                                     int tw, int TH, int wm, int wn, int nt0, int col, int kq, const EpiConsts& k) {
    const int gow = tw * kDnTW + col;
    if (p.yc) {
        if constexpr (!TAB_READY) {
            __syncthreads();   // every wave has left the main loop: the operand tiles in LDS are dead
            enc_fill<kDnThreads>(reinterpret_cast<uint2*>(smem), p.enc_out);
            __syncthreads();
        }
        const float r1 = p.enc_out.r1, lo = p.enc_out.lo, hi = p.enc_out.hi;
        const int chw = (nt0 + wn * 4) * 16;   // this wave's 64 channels
        PostOp po = p.post;
        po.relu = 0;   // the ReLU is folded into the quantizer (enc4_code_relu)
        const int chs = chw + kq * 16;   // after the transpose: this lane's 16 channels
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int goh = th * TH + wm * MT + i;
            uint32_t c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 r;
                r.x = ((acc[i][j][0] + k.bq[j].x) * p.s1x) * p.s2;
                r.y = ((acc[i][j][1] + k.bq[j].y) * p.s1x) * p.s2;
                r.z = ((acc[i][j][2] + k.bq[j].z) * p.s1x) * p.s2;
                r.w = ((acc[i][j][3] + k.bq[j].w) * p.s1x) * p.s2;
                r = post_apply_v(r, po, k.pv[j]);
                if (p.y_sgn) c[j] = code_sign4(enc4_code<false>(r, r1, lo, hi, smem), r, p.y_fmt);
                else c[j] = enc4_code_relu(r, r1, lo, hi, smem);
            }
            rows_transpose4(c[0], c[1], c[2], c[3]);
            if (goh < p.Ho && gow < p.Wo && chs < p.O) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<u32x4*>(p.yc + (((size_t)n * p.Ho + goh) * p.Wo + gow) * p.O + chs) = u32x4{c[0], c[1], c[2], c[3]};
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = (nt0 + wn * 4 + j) * 16 + kq * 4;
        if (ch >= p.O) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int goh = th * TH + wm * MT + i;
            if (goh >= p.Ho || gow >= p.Wo) continue;
            float4 r;
            r.x = ((acc[i][j][0] + k.bq[j].x) * p.s1x) * p.s2;
            r.y = ((acc[i][j][1] + k.bq[j].y) * p.s1x) * p.s2;
            r.z = ((acc[i][j][2] + k.bq[j].z) * p.s1x) * p.s2;
            r.w = ((acc[i][j][3] + k.bq[j].w) * p.s1x) * p.s2;
            st_stream4<SLFP_NT_DENSE>(p.y + (((size_t)n * p.Ho + goh) * p.Wo + gow) * p.O + ch, post_apply_v(r, p.post, k.pv[j]));
        }
    }
}

// Epilogue of the per-tile GEMM kernels: Conv2d_Q's (acc * Ka) * Kw with the reference's two roundings, the fused post-op, and either
// float32 stores (16 bytes per lane: 4 consecutive channels of one pixel) or the consumer's 1-byte codes.  Code output: the 4
// channel tiles of a wave are encoded, transposed across the lane quarters (rows_transpose4) so that a lane holds 16
// CONSECUTIVE channels of its pixel, and stored as one 16-byte piece.
template <int MT>
__device__ __forceinline__ void dense_epilogue(const DenseParams& p, const floatx4 (&acc)[MT][4], unsigned char* smem, int n, int th,
                                               int tw, int TH, int wm, int wn, int nt0, int col, int kq) {
    EpiConsts k;
    dense_epilogue_consts(p, wn, nt0, kq, k);
    dense_epilogue_apply<MT, false>(p, acc, smem, n, th, tw, TH, wm, wn, nt0, col, kq, k);
}

// WM x WN = 8 waves; MT = output rows per wave.  Both operands arrive by LDS-DMA: no VALU work
// in the main loop beyond addresses.
// PASSES == 3 (float32-equivalent): both operands carry an fp16 residual plane (hi + lo), staged next to
// the hi planes, and every tile takes 3 MFMAs (lo*hi + hi*lo + hi*hi), as the pointwise kernels do.
// NSLOT: halo pieces a wave stages per chunk (3 covers every 3x3 stride-1 tiling; 9 the rest).
// NWB: weight-tile buffers.  2: the tile two taps ahead is requested into the buffer freed by this tap's barrier and must have
// LANDED by the next tap's barrier (s_waitcnt vmcnt(0)): a one-tap flight window (16-32 MFMAs, ~0.25 us) against an L2 round
// trip of ~1 us -- the matrix pipe idles 50-80 % of the time on the one-workgroup-per-CU tilings (profiles/r02e_vgg16).
// 3 (round 3): the tile THREE taps ahead is requested; a tap's wait leaves the requests of the previous tap in flight
// (counted vmcnt: they are the youngest -- the halo pieces of a tap are issued before its weight pieces), so every request has
// two taps to land.
template <int WM, int WN, int MT, int PASSES, int NSLOT, int NWB = 2>
__global__ __launch_bounds__(kDnThreads, (MT == 4 || PASSES == 3 ? 2 : 4)) void k_dense_mfma(const DenseParams p) {
    static_assert(WM * WN == 8, "8 waves");
    static_assert(NWB == 2 || NWB == 3, "two or three weight buffers");
    constexpr int TH = WM * MT, BN = WN * 64, WT = BN * 128;  // WT: bytes of one tap's weight tile (one plane)
    constexpr int PL = PASSES == 3 ? 2 : 1;                   // operand planes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wbuf = smem;                 // [NWB buffers][PL planes][WT]
    unsigned char* xsb = wbuf + NWB * PL * WT;  // [2 buffers][PL planes][x_pieces * 1 KiB]   (8 halo pixels x 128 B per piece)
    const uint32_t xbytes = (uint32_t)p.x_pieces * 1024u;

    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);   // channel slice slowest: an XCD's L2 holds one W slice
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b % p.N;
    const int nb = b / p.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, kq = lane >> 4;
    const int h_in0 = th * TH * p.S - p.ph, w_in0 = tw * kDnTW * p.S - p.pw;

    // ---- halo tile: piece q of a chunk = 8 consecutive halo pixels x 64 channels (fp16), one DMA.
    // lane -> (pixel pc*8 + lane/8, LDS slot lane%8); the slot holds 16-byte chunk slot ^ swz(pixel).
    // A wave stages pieces wave, wave + 8, ...: the same pieces for every 64-channel chunk, so each piece's source
    // address (pixel -> (ih, iw) -> bounds -> pointer: a division and ~40 VALU instructions) is worked out ONCE, before
    // the loops; per chunk it only moves by 128 bytes.  (Profile r02c_vgg16: 67 VALU instructions per tap and wave,
    // more issue cycles than the 16 MFMAs they surround.)
    const unsigned char* xen = reinterpret_cast<const unsigned char*>(p.xe) + (size_t)n * p.H * p.W * p.Cp * 2;
    const ptrdiff_t x_lo_off = PASSES == 3 ? reinterpret_cast<const unsigned char*>(p.xlo) - reinterpret_cast<const unsigned char*>(p.xe) : 0;
    const unsigned char* zp = p.zero_page + (lane & 7) * 16;
    constexpr uint32_t kPad = 0xFFFFFFFFu;
    uint32_t xoff[NSLOT];   // chunk-0 byte offset (from xen) of this lane's 16 bytes of piece slot j; kPad = padding (zero page)
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int pc = wave + 8 * j;
        const int pix = pc * 8 + (lane >> 3);
        const int ih = pix / p.IW, iw = pix - ih * p.IW;
        const int gh = h_in0 + ih, gw = w_in0 + iw;
        const int c16 = (lane & 7) ^ (((pix >> 1) & 3) << 1);     // source chunk for this slot (dn_x_off's swizzle)
        const bool inb = pc < p.x_pieces && pix < p.n_pix && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
        xoff[j] = inb ? (uint32_t)((gh * p.W + gw) * p.Cp) * 2u + (uint32_t)c16 * 16u : kPad;   // < 2^31 (dense_mfma_applicable)
    }
    auto stage_x = [&](int j, int chunk, int buf) {   // piece slot j (compile-time after unrolling) of this wave
        const int pc = wave + 8 * j;
        const bool live = xoff[j] != kPad;            // Cp is a multiple of 64: every chunk is whole
        const unsigned char* src = live ? xen + (xoff[j] + (uint32_t)chunk * 128u) : zp;
        glds16(src, xsb + (size_t)buf * PL * xbytes + (size_t)pc * 1024);
        if constexpr (PASSES == 3)
            glds16(live ? src + x_lo_off : src, xsb + ((size_t)buf * PL + 1) * xbytes + (size_t)pc * 1024);
    };

    // ---- weight tap tile: BN/8 pieces of 1 KiB (channel tile, k-step), WN per wave.  The (tile, k-step) part of the
    // address is fixed per wave; per (tap, chunk) only a wave-uniform offset moves.
    const int nt0 = nb * (BN / 16);
    const unsigned char* wsrc[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int pc = wave * WN + j;                     // piece = (channel tile pc >> 1, k-step pc & 1)
        int nt = nt0 + (pc >> 1);
        nt = nt < p.n_tiles ? nt : p.n_tiles - 1;         // tiles past C_out: clamp (results never stored)
        wsrc[j] = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)nt * p.KS + (pc & 1)) * 1024 + (size_t)lane * 16;
    }
    const size_t w_tap_stride = (size_t)p.n_tiles * p.KS * 1024;
    const ptrdiff_t w_lo_off = reinterpret_cast<const unsigned char*>(p.wlo) - reinterpret_cast<const unsigned char*>(p.w);
    auto stage_w = [&](int tap, int chunk, int buf) {
        const size_t o = (size_t)tap * w_tap_stride + (size_t)chunk * 2048;   // wave-uniform
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int pc = wave * WN + j;
            glds16(wsrc[j] + o, wbuf + (size_t)buf * PL * WT + (size_t)pc * 1024);
            if constexpr (PASSES == 3)
                glds16(wsrc[j] + o + w_lo_off, wbuf + ((size_t)buf * PL + 1) * WT + (size_t)pc * 1024);
        }
    };

    floatx4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int n_chunks = p.KS >> 1, n_taps = p.KH * p.KW;
    const int n_steps = n_chunks * n_taps;
    stage_w(0, 0, 0);
    if (n_steps > 1) stage_w(n_taps > 1 ? 1 : 0, n_taps > 1 ? 0 : 1, 1);
    if constexpr (NWB == 3) { if (n_steps > 2) stage_w(2 % n_taps, 2 / n_taps, 2); }
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
        if (wave + 8 * j < p.x_pieces) stage_x(j, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Per tap: (1) this tap's fragments LDS -> registers; (2) barrier (every wave has read wbuf[wb],
    // and the DMAs issued one tap ago have landed: vmcnt(0)); (3) the DMA for tap+2 goes into the
    // buffer just freed, plus a slice of the next chunk's halo; (4) the MFMAs run from registers while
    // those DMAs fly.  Two weight buffers give a two-tap flight window because the operands of the
    // tap in progress live in registers.
    int wb = 0, xb = 0;
    bool w_in_flight = false;   // NWB == 3: did the previous tap request a weight tile (wave-uniform)?
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const bool more_chunks = chunk + 1 < n_chunks;
        const unsigned char* xs = xsb + (size_t)xb * PL * xbytes;
        for (int tap = 0; tap < n_taps; ++tap) {
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const unsigned char* wt = wbuf + (size_t)wb * PL * WT + (size_t)(wn * 4) * 2048 + lane * 16;
            half8 wf[2][4], xf[2][MT], wl[PASSES == 3 ? 2 : 1][4], xl[PASSES == 3 ? 2 : 1][MT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wf[ks][j] = *reinterpret_cast<const half8*>(wt + j * 2048 + ks * 1024);
                    if constexpr (PASSES == 3) wl[ks][j] = *reinterpret_cast<const half8*>(wt + WT + j * 2048 + ks * 1024);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = ((wm * MT + i) * p.S + kh) * p.IW + col * p.S + kw;
                    const uint32_t off = dn_x_off(row, ks * 4 + kq);
                    xf[ks][i] = *reinterpret_cast<const half8*>(xs + off);
                    if constexpr (PASSES == 3) xl[ks][i] = *reinterpret_cast<const half8*>(xs + xbytes + off);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // The DMAs issued one tap ago must have landed before the barrier publishes them.  hipcc does
            // NOT count a global_load_lds issued in the previous loop iteration when it lowers
            // __syncthreads() here (it emitted lgkmcnt(0) only: rare stale weight fragments at 12 544
            // workgroups), so the wait is explicit.
            if constexpr (NWB == 3) {
                // everything but the previous tap's weight pieces (the WN * PL youngest requests of this wave) has landed
                if (w_in_flight) {
                    if constexpr (WN * PL == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else if constexpr (WN * PL == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if constexpr (WN * PL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            // a slice of the next chunk's halo; nothing on the last tap: its DMA would still be in flight
            // when the next chunk's first fragments are read.  Issued BEFORE the weight pieces (see NWB).
            if (more_chunks && tap + 1 < n_taps) {
#pragma unroll
                for (int j = 0; j < NSLOT; ++j) {   // slot j goes out with tap j / x_per_tap (wave-uniform)
                    if (j / p.x_per_tap == tap && wave + 8 * j < p.x_pieces) stage_x(j, chunk + 1, xb ^ 1);
                }
            }
            // the weight tile NWB taps ahead -> the buffer every wave has just finished reading
            int tap2 = tap + NWB, chunk2 = chunk;
            while (tap2 >= n_taps) { tap2 -= n_taps; ++chunk2; }
            w_in_flight = chunk2 < n_chunks;
            if (w_in_flight) stage_w(tap2, chunk2, wb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (PASSES == 3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ks][j], xf[ks][i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], xl[ks][i], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], xf[ks][i], acc[i][j], 0, 0, 0);
                    }
            wb = (NWB == 2) ? (wb ^ 1) : (wb == 2 ? 0 : wb + 1);
        }
        xb ^= 1;
    }

    dense_epilogue<MT>(p, acc, smem, n, th, tw, TH, wm, wn, nt0, col, kq);
}

// ======================================================================================
// k_dense3x3: k_dense_mfma specialised for 3x3, stride 1, single pass (every VGG-16 layer, 13 of ResNet-50's 16 3x3 layers,
// SqueezeNet's expand3x3) -- round 3.  The general kernel derives every fragment address inside the tap loop (tap -> (kh, kw)
// by division, halo row, XOR swizzle: 60-100 VALU instructions per tap and wave around 16-32 MFMAs, profiles/r02e_vgg16:
// valu_insts_per_wave 1475-4315, mfma_busy 0.32-0.50) and is VALU-issue-bound on them.  Here the 9 taps are unrolled: a
// lane's fragment offsets depend on the tap only through (halo row i + kh, kw), so the (MT + 2) x 3 swizzled offsets are
// computed ONCE per workgroup; per tap what is left is one add per fragment (the halo buffer of the chunk) and one XOR for
// the second k-step (the swizzle moves bit 6 with it).  Same staging, same waits, same MFMA order: bit-identical results.
// ======================================================================================
template <int WM, int WN, int MT, int NSLOT, int NWB>
__global__ __launch_bounds__(kDnThreads, (MT == 4 ? 2 : 4)) void k_dense3x3(const DenseParams p) {
    static_assert(WM * WN == 8, "8 waves");
    constexpr int TH = WM * MT, BN = WN * 64, WT = BN * 128;
    constexpr int IW = kDnTW + 2;   // halo width of a 16-column tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wbuf = smem;                 // [NWB buffers][WT]
    unsigned char* xsb = wbuf + NWB * WT;       // [2 buffers][x_pieces * 1 KiB]
    const uint32_t xbytes = (uint32_t)p.x_pieces * 1024u;

    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b % p.N;
    const int nb = b / p.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, kq = lane >> 4;
    const int h_in0 = th * TH - p.ph, w_in0 = tw * kDnTW - p.pw;

    // halo pieces: as k_dense_mfma
    const unsigned char* xen = reinterpret_cast<const unsigned char*>(p.xe) + (size_t)n * p.H * p.W * p.Cp * 2;
    const unsigned char* zp = p.zero_page + (lane & 7) * 16;
    constexpr uint32_t kPad = 0xFFFFFFFFu;
    uint32_t xoff[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int pc = wave + 8 * j;
        const int pix = pc * 8 + (lane >> 3);
        const int ih = pix / IW, iw = pix - ih * IW;
        const int gh = h_in0 + ih, gw = w_in0 + iw;
        const int c16 = (lane & 7) ^ (((pix >> 1) & 3) << 1);
        const bool inb = pc < p.x_pieces && pix < p.n_pix && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
        xoff[j] = inb ? (uint32_t)((gh * p.W + gw) * p.Cp) * 2u + (uint32_t)c16 * 16u : kPad;
    }
    auto stage_x = [&](int j, int chunk, int buf) {
        const int pc = wave + 8 * j;
        const bool live = xoff[j] != kPad;
        const unsigned char* src = live ? xen + (xoff[j] + (uint32_t)chunk * 128u) : zp;
        glds16(src, xsb + (size_t)buf * xbytes + (size_t)pc * 1024);
    };
    const int nt0 = nb * (BN / 16);
    const unsigned char* wsrc[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int pc = wave * WN + j;
        int nt = nt0 + (pc >> 1);
        nt = nt < p.n_tiles ? nt : p.n_tiles - 1;
        wsrc[j] = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)nt * p.KS + (pc & 1)) * 1024 + (size_t)lane * 16;
    }
    const size_t w_tap_stride = (size_t)p.n_tiles * p.KS * 1024;
    auto stage_w = [&](int tap, int chunk, int buf) {
        const size_t o = (size_t)tap * w_tap_stride + (size_t)chunk * 2048;
#pragma unroll
        for (int j = 0; j < WN; ++j) glds16(wsrc[j] + o, wbuf + (size_t)buf * WT + (size_t)(wave * WN + j) * 1024);
    };

    // this lane's fragment offsets inside a halo buffer, k-step 0: halo row (wm * MT + r), column col + kw
    uint32_t xo[MT + 2][3];
#pragma unroll
    for (int r = 0; r < MT + 2; ++r)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xo[r][kw] = dn_x_off((wm * MT + r) * IW + col + kw, kq);
    const uint32_t wlane = (uint32_t)(wn * 4) * 2048u + (uint32_t)lane * 16u;

    floatx4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int n_chunks = p.KS >> 1;
    constexpr int n_taps = 9;
    const int n_steps = n_chunks * n_taps;
    stage_w(0, 0, 0);
    stage_w(1, 0, 1);
    if constexpr (NWB == 3) stage_w(2, 0, 2);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
        if (wave + 8 * j < p.x_pieces) stage_x(j, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    (void)n_steps;

    int wb = 0, xb = 0;
    bool w_in_flight = false;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const bool more_chunks = chunk + 1 < n_chunks;
        const uint32_t xs_off = (uint32_t)(NWB * WT) + (uint32_t)xb * xbytes;   // byte offset of this chunk's halo buffer in smem
#pragma unroll
        for (int tap = 0; tap < n_taps; ++tap) {
            const int kh = tap / 3, kw = tap % 3;   // compile-time after unrolling
            const unsigned char* wt = wbuf + (size_t)wb * WT + wlane;
            half8 wf[2][4], xf[2][MT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[ks][j] = *reinterpret_cast<const half8*>(wt + j * 2048 + ks * 1024);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                // the add stays HERE (hipcc otherwise hoists the 9 x MT x 2 addresses of a chunk to its top: 40-70 live VGPRs, spills)
                uint32_t o0 = xs_off + xo[i + kh][kw];
                asm volatile("" : "+v"(o0));
                xf[0][i] = *reinterpret_cast<const half8*>(smem + o0);
                xf[1][i] = *reinterpret_cast<const half8*>(smem + (o0 ^ 64u));   // k-step 1: chunk bit 2 -> byte-offset bit 6
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NWB == 3) {
                if (w_in_flight) {
                    if constexpr (WN == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else if constexpr (WN == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (more_chunks && tap < NSLOT && tap + 1 < n_taps) {   // slot `tap` of the next chunk's halo (x_per_tap == 1)
                if (wave + 8 * tap < p.x_pieces) stage_x(tap, chunk + 1, xb ^ 1);
            }
            int tap2 = tap + NWB, chunk2 = chunk;
            if (tap2 >= n_taps) { tap2 -= n_taps; ++chunk2; }
            w_in_flight = chunk2 < n_chunks;
            if (w_in_flight) stage_w(tap2, chunk2, wb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], xf[ks][i], acc[i][j], 0, 0, 0);
            // the next tap's fragment reads stay behind this tap's MFMAs (hoisted, two taps' fragments are live at once: spills)
            __builtin_amdgcn_sched_barrier(0);
            wb = (NWB == 2) ? (wb ^ 1) : (wb == 2 ? 0 : wb + 1);
        }
        xb ^= 1;
    }

    dense_epilogue<MT>(p, acc, smem, n, th, tw, TH, wm, wn, nt0, col, kq);
}

// ======================================================================================
// k_dense3x3_res: 3x3, stride 1, single pass, C_in <= 64 (ONE 64-channel chunk): VGG-16's conv1_2 / conv2_1, ResNet-50's
// layer1 3x3s, SqueezeNet's expand3x3 -- end of round 3.  With one chunk the nine tap tiles of a 64-channel output slice are
// all the weights there are (72 KiB): k_dense3x3 re-staged them for every 128-pixel tile (72 KiB of W next to 23 KiB of halo
// through the DMA path, one barrier per tap around 8 MFMAs per wave: profiles/r03d_vgg16, 64 -> 64 @224 1.09 ms, mfma_busy
// 0.19).  Here a PERSISTENT workgroup keeps them resident in LDS and walks over its share of the output tiles: per tile one
// halo DMA (into the second buffer, issued before the MFMAs of the current tile), 9 taps of MFMAs without a barrier, ONE
// barrier, then the stores -- which stay in flight under the next tile's MFMAs because the wait that publishes the next
// halo comes before them in the instruction stream.  Same fragment layout, same MFMA order: bit-identical to k_dense3x3.
//   workgroup = 8 waves x MT rows x 16 columns x 64 output channels; LDS = 72 KiB W + 2 halos + the code table.
// ======================================================================================
// ENCX (float32 interface, C_in == 64, one channel slice): the halo is loaded as float32 straight from the layer's input,
// encoded with the threshold table (enc4_f16: the values k_dense_encode writes, bit for bit) and written to the same swizzled
// LDS tile -- the loads of the NEXT tile are issued before the MFMAs of the current one and consumed behind them.  Saves the
// pre-pass (64 -> 64 @224 at batch 128: 1.64 GB read + 0.82 GB written + 1.04 GB read back, 427 us) for ~50 VALU
// instructions per 16-byte chunk.
template <int MT, int NSLOT, bool ENCX>
__global__ __launch_bounds__(kDnThreads, 2) void k_dense3x3_res(const DenseParams p) {
    constexpr int TH = 8 * MT, WT = 64 * 128;
    constexpr int IW = kDnTW + 2, IH = TH + 2, NPIX = IH * IW, PIECES = (NPIX + 7) / 8;
    static_assert((PIECES + 7) / 8 <= NSLOT, "halo slots");
    constexpr uint32_t xbytes = (uint32_t)PIECES * 1024u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wres = smem;                 // [9 taps][WT]
    unsigned char* xsb = smem + 9 * WT;         // [2][PIECES KiB]
    unsigned char* tab = xsb + 2 * xbytes;      // code table of the consumer (kEncEntries * 8 bytes)
    constexpr int kTabBytes = (kEncEntries * 8 + 63) & ~63;
    unsigned char* tabx = tab + kTabBytes;      // ENCX: kEncF16P table of this layer's (Ka, format)
    constexpr int XSL = (NPIX * 8 + kDnThreads - 1) / kDnThreads;   // ENCX: 16-byte chunks of a halo per thread

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    // grid = gs spatial walkers x n_blocks channel slices; a walker's tiles: one of R contiguous ranges (R = 8: neighbouring
    // tiles, which share halo columns / rows, on one XCD's L2 when there is a single channel slice), strided inside it
    const uint32_t nbk = (uint32_t)p.n_blocks;
    const uint32_t gs = gridDim.x / nbk;
    const int nb = (int)(blockIdx.x % nbk);
    const uint32_t sb = blockIdx.x / nbk;
    const uint32_t T = (uint32_t)p.N * p.tiles_h * p.tiles_w;
    const uint32_t R = (gs % 8u == 0u) ? 8u : 1u;
    const uint32_t per_r = (T + R - 1) / R;
    const uint32_t r_end = min(T, (sb % R + 1) * per_r);
    const uint32_t t_step = gs / R;
    uint32_t tile = (sb % R) * per_r + sb / R;
    if (tile >= r_end) return;   // whole workgroup (uniform)

    // halo slots of this lane: pixel -> (ih, iw, source chunk), fixed for every tile
    const unsigned char* zp = p.zero_page + (lane & 7) * 16;
    uint32_t slot[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int pc = wave + 8 * j;
        const int pix = pc * 8 + (lane >> 3);
        const int ih = pix / IW, iw = pix - ih * IW;
        const int c16 = (lane & 7) ^ (((pix >> 1) & 3) << 1);
        slot[j] = (pc < PIECES && pix < NPIX) ? (uint32_t)(ih << 16 | iw << 8 | c16) : 0xFFFFFFFFu;
    }
    auto stage_tile = [&](uint32_t t, int buf) {
        const int tw = (int)(t % (uint32_t)p.tiles_w);
        const uint32_t t2 = t / (uint32_t)p.tiles_w;
        const int th = (int)(t2 % (uint32_t)p.tiles_h);
        const int n = (int)(t2 / (uint32_t)p.tiles_h);
        const int h0 = th * TH - p.ph, w0 = tw * kDnTW - p.pw;
        const unsigned char* xen = reinterpret_cast<const unsigned char*>(p.xe) + (size_t)n * p.H * p.W * 128;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int pc = wave + 8 * j;
            if (pc < PIECES) {   // wave-uniform
                const int gh = h0 + (int)(slot[j] >> 16), gw = w0 + (int)((slot[j] >> 8) & 0xFF);
                const bool inb = slot[j] != 0xFFFFFFFFu && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
                const unsigned char* src = inb ? xen + ((uint32_t)(gh * p.W + gw) * 128u + (slot[j] & 0xFF) * 16u) : zp;
                glds16(src, xsb + (size_t)buf * xbytes + (size_t)pc * 1024);
            }
        }
    };

    // ENCX: chunk task q = pixel * 8 + chunk c (channels {4j..4j+3, 16+4j..16+4j+3} + 32 g, c = 4 g + j) -> thread q % 512
    uint32_t xslot[ENCX ? XSL : 1];
    if constexpr (ENCX) {
#pragma unroll
        for (int j = 0; j < XSL; ++j) {
            const int q = (int)threadIdx.x + kDnThreads * j;
            const int pix = q >> 3, c = q & 7;
            const int ih = pix / IW, iw = pix - ih * IW;
            xslot[j] = pix < NPIX ? (uint32_t)(ih << 16 | iw << 8 | c) : 0xFFFFFFFFu;
        }
    }
    float4 xa[ENCX ? XSL : 1], xb4[ENCX ? XSL : 1];
    uint32_t xlive = 0;   // bit j: slot j of the tile in flight is a pixel of the image (else padding: encodes as +0)
    // Every slot ALWAYS loads (padding slots from the start of the input, discarded when the tile is encoded): a load under a
    // divergent `if` is waited for at the end of its block (hipcc: six serialized round trips per tile, measured +3.8 us).
    auto load_tile = [&](uint32_t t) {   // float32 halo of tile t -> registers
        const int tw = (int)(t % (uint32_t)p.tiles_w);
        const uint32_t t2 = t / (uint32_t)p.tiles_w;
        const int th = (int)(t2 % (uint32_t)p.tiles_h);
        const int n = (int)(t2 / (uint32_t)p.tiles_h);
        const int h0 = th * TH - p.ph, w0 = tw * kDnTW - p.pw;
        const float* xn = p.x32 + (size_t)n * p.H * p.W * 64;
        xlive = 0;
#pragma unroll
        for (int j = 0; j < XSL; ++j) {
            const int gh = h0 + (int)(xslot[j] >> 16), gw = w0 + (int)((xslot[j] >> 8) & 0xFF);
            const int c = (int)(xslot[j] & 7);
            const bool inb = xslot[j] != 0xFFFFFFFFu && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            const uint32_t off = inb ? (uint32_t)(gh * p.W + gw) * 64u + (uint32_t)((c >> 2) * 32 + (c & 3) * 4) : 0u;
            xlive |= inb ? (1u << j) : 0u;
            xa[j] = ld_stream4<SLFP_NT_DENSE>(xn + off);
            xb4[j] = ld_stream4<SLFP_NT_DENSE>(xn + off + 16);
        }
    };
    auto encode_tile = [&](int buf) {    // registers -> encoded, swizzled LDS tile (padding pixels: +0, as the zero page gave)
        const float r1 = p.enc_in.r1, lo = p.enc_in.lo, hi = p.enc_in.hi;
#pragma unroll
        for (int j = 0; j < XSL; ++j) {
            const int q = (int)threadIdx.x + kDnThreads * j;
            const bool live = (xlive >> j) & 1u;
            const uint2 e0 = enc4_f16(xa[j], r1, lo, hi, tabx);
            const uint2 e1 = enc4_f16(xb4[j], r1, lo, hi, tabx);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = live ? u32x4{e0.x, e0.y, e1.x, e1.y} : u32x4{0u, 0u, 0u, 0u};
            if (q < NPIX * 8) *reinterpret_cast<u32x4*>(xsb + (size_t)buf * xbytes + dn_x_off(q >> 3, q & 7)) = v;
        }
    };

    // ---- all nine tap tiles of this channel slice: 72 pieces of 1 KiB, 9 per wave
    const int nt0 = nb * 4;
    {
        const size_t w_tap_stride = (size_t)p.n_tiles * p.KS * 1024;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int pq = wave * 9 + q;
            const int tap = pq >> 3, pc = pq & 7;           // piece = (channel tile pc >> 1, k-step pc & 1)
            int nt = nt0 + (pc >> 1);
            nt = nt < p.n_tiles ? nt : p.n_tiles - 1;       // tiles past C_out: clamp (results never stored)
            glds16(reinterpret_cast<const unsigned char*>(p.w) + (size_t)tap * w_tap_stride + ((size_t)nt * p.KS + (pc & 1)) * 1024 + (size_t)lane * 16,
                   wres + (size_t)pq * 1024);
        }
    }
    if constexpr (ENCX) {
        enc_fill_compact<kDnThreads>(reinterpret_cast<uint2*>(tabx), p.enc_in);
        load_tile(tile);
        __syncthreads();   // the table
        encode_tile(0);
    } else {
        stage_tile(tile, 0);
    }
    if (p.yc) enc_fill<kDnThreads>(reinterpret_cast<uint2*>(tab), p.enc_out);

    uint32_t xo[MT + 2][3];
#pragma unroll
    for (int r = 0; r < MT + 2; ++r)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xo[r][kw] = dn_x_off((wave * MT + r) * IW + col + kw, kq);
    const uint32_t wlane = (uint32_t)lane * 16u;

    EpiConsts epi;
    dense_epilogue_consts(p, 0, nt0, kq, epi);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int xb = 0;
    for (;;) {
        const uint32_t next = tile + t_step;
        const bool has_next = next < r_end;
#ifndef SLFP_RES_NODMA
        // (every wave left that buffer before the last barrier)
        if constexpr (ENCX) load_tile(has_next ? next : tile);   // unconditional: under `if (has_next)` the registers become loop phis, copied -- and waited for -- right here
        else if (has_next) stage_tile(next, xb ^ 1);
#endif
        if constexpr (ENCX) __builtin_amdgcn_sched_barrier(0);   // the loads fly under the MFMAs: their encode stays behind them
        floatx4 acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const uint32_t xs_off = (uint32_t)(9 * WT) + (uint32_t)xb * xbytes;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const unsigned char* wt = wres + (size_t)tap * WT + wlane;
            half8 wf[2][4], xf[2][MT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#ifdef SLFP_RES_NOLDS
                    wf[ks][j] = half8{(_Float16)(float)(tap + j), 1, 2, 3, 4, 5, 6, (_Float16)(float)lane};
#else
                    wf[ks][j] = *reinterpret_cast<const half8*>(wt + j * 2048 + ks * 1024);
#endif
                }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                uint32_t o0 = xs_off + xo[i + kh][kw];
                asm volatile("" : "+v"(o0));
#ifdef SLFP_RES_NOLDS
                xf[0][i] = half8{(_Float16)(float)(o0 & 7), 1, 2, 3, 4, 5, 6, 7};
                xf[1][i] = half8{(_Float16)(float)(o0 & 3), 1, 2, 3, 4, 5, 6, 7};
#else
                xf[0][i] = *reinterpret_cast<const half8*>(smem + o0);
                xf[1][i] = *reinterpret_cast<const half8*>(smem + (o0 ^ 64u));
#endif
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#ifdef SLFP_RES_NOMFMA
                        acc[i][j][0] += (float)(wf[ks][j][0] + xf[ks][i][0]);   // keeps the LDS reads alive
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], xf[ks][i], acc[i][j], 0, 0, 0);
#endif
                    }
        }
        // the next halo has had the whole tile to land; the stores of the previous tile are older still
        if constexpr (ENCX) {
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) encode_tile(xb ^ 1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        {
            const int tw = (int)(tile % (uint32_t)p.tiles_w);
            const uint32_t t2 = tile / (uint32_t)p.tiles_w;
            const int th = (int)(t2 % (uint32_t)p.tiles_h);
            const int n = (int)(t2 / (uint32_t)p.tiles_h);
#ifdef SLFP_RES_NOST
            if (acc[0][0][0] == 12345.678f)   // never true for these inputs: the epilogue and its stores are skipped
#endif
            dense_epilogue_apply<MT, true>(p, acc, tab, n, th, tw, TH, wave, 0, nt0, col, kq, epi);
        }
        if (!has_next) break;
        tile = next;
        xb ^= 1;
    }
}

// ---- host side: pick the tiling ----------------------------------------------------------
struct DenseCfg { int wm, wn, mt; };
// instantiated tilings, widest first
static const DenseCfg kDenseCfgs[] = {{2, 4, 4}, {4, 2, 4}, {4, 2, 2}, {4, 2, 1}, {8, 1, 2}, {8, 1, 1}};

struct DenseGeom { DenseCfg cfg; int ih, iw, pieces, per_tap; size_t lds; int occ; int nwb; };

static bool dense_cfg_geometry(const slfp_conv2d_desc& d, const DenseCfg& c, int planes, DenseGeom* g) {
    const int S = d.stride_h, th = c.wm * c.mt, taps = (int)(d.kh * d.kw);
    g->cfg = c;
    g->ih = (th - 1) * S + (int)d.kh;
    g->iw = (kDnTW - 1) * S + (int)d.kw;
    g->pieces = (g->ih * g->iw + 7) / 8;
    g->per_tap = (int)ceil_div(ceil_div(g->pieces, 8), taps - 1);  // all slices issued before the last tap
    g->nwb = 2;
    g->lds = (size_t)planes * (2 * (size_t)c.wn * 64 * 128 + 2 * (size_t)g->pieces * 1024);
    // MT 4 tilings hold 64 accumulator VGPRs + fragments: compiled for one workgroup per CU; the others for two
    g->occ = (c.mt == 4 || planes == 2) ? 1 : (g->lds <= 80 * 1024 ? 2 : 1);
    // one workgroup per CU: nobody covers its DMA waits -> three weight buffers where they fit (single-plane kernels)
    const size_t lds3 = g->lds + (size_t)planes * c.wn * 64 * 128;
    if (g->occ == 1 && planes == 1 && taps >= 3 && lds3 <= 160 * 1024 && switches().dense_nwb != 2) { g->nwb = 3; g->lds = lds3; }
    return g->lds <= 160 * 1024 && ceil_div(g->pieces, 8) <= kMaxSlots;
}

// Relative cost of a tiling for this layer: workgroups per CU x padded tile MACs x a per-tiling
// factor fitted to measured layer times (VGG-16 / ResNet-50 shapes, profiles/dense_cfg_sweep.sh):
// small wave tiles re-read fragments and hit the barrier more often per MFMA, one-workgroup-per-CU
// tilings have nobody to overlap their DMA waits with.
static bool dense_choose(const slfp_conv2d_desc& d, int planes, int64_t h_out, int64_t w_out, DenseGeom* best) {
    double best_cost = 0;
    bool found = false;
    for (const DenseCfg& c : kDenseCfgs) {
        DenseGeom g;
        if (!dense_cfg_geometry(d, c, planes, &g)) continue;
        if (switches().dense_cfg && switches().dense_cfg == c.wm * 100 + c.wn * 10 + c.mt) { *best = g; return true; }   // sweep tool
        const int th = c.wm * c.mt, bn = c.wn * 64;
        const int64_t blocks = d.n * ceil_div(h_out, th) * ceil_div(w_out, kDnTW) * ceil_div(d.c_out, bn);
        const int64_t per_cu = ceil_div(blocks, 256);
        double f = c.mt == 4 ? (c.wn == 4 ? 1.12 : 1.15) : (c.mt == 2 && c.wn == 2 ? 1.0 : 1.5);
        // round 3: 3x3 stride-1 single-plane layers run the unrolled k_dense3x3 on the MT = 4 / MT = 1 tilings: measured per
        // VGG-16 layer (batch 128, profiles/dense_cfg_sweep3.sh) {4,2,4} 500 / 830 / 324 / 580 / 155 us against the general
        // kernel's {4,2,2} 513 / 845 / 367 / 683 / 182; 64 -> 64 @224: {8,1,1} 1509 against 1628
        if (planes == 1 && d.kh == 3 && d.kw == 3 && d.stride_h == 1 && !switches().dense_generic) {
            if (c.mt == 4 && c.wn == 2) f = 0.92;
            else if (c.mt == 1 && c.wm == 8) f = 1.2;
        }
        if (c.mt != 4 && g.occ < 2) f *= 1.15;   // built for two workgroups per CU but LDS only fits one
        if (per_cu < g.occ) f *= 1.3;            // too few workgroups to pair up
        const double cost = (double)per_cu * th * bn * f;
        if (!found || cost < best_cost) { best_cost = cost; *best = g; found = true; }
    }
    return found;
}

static int dense_planes(const slfp_conv2d_desc& d, int passes) { return (d.qbits == 8 && passes == 3) ? 2 : 1; }

bool dense_mfma_applicable(const slfp_conv2d_desc& d, int passes) {
    if (d.groups != 1 || d.kh * d.kw <= 1 || d.dil_h != 1 || d.dil_w != 1) return false;
    if (d.stride_h != d.stride_w || d.stride_h > 2) return false;
    if (d.c_in % 4 || d.c_in < 16 || d.c_out % 4) return false;
    if ((int64_t)d.h * d.w * (d.c_in + 63) >= (1ll << 30)) return false;   // 32-bit byte offsets inside one image of the fp16 copy
    DenseGeom g;
    for (const DenseCfg& c : kDenseCfgs)
        if (dense_cfg_geometry(d, c, dense_planes(d, passes), &g)) return true;
    return false;  // (float32-equivalent mode: stride-2 halo tiles do not fit twice -> k_direct)
}

// channels of the pre-encoded copy: whole 64-channel chunks (the kernel stages 128-byte pixel rows without a tail case)
static int64_t dense_cp(const slfp_conv2d_desc& d) { return ceil_div(d.c_in, 64) * 64; }

// workspace = [256 B zero page][pre-encoded input: N*H*W*Cp fp16][its residual plane in float32-equivalent mode]
static size_t dense_plane_bytes(const slfp_conv2d_desc& d) {
    return (((size_t)d.n * d.h * d.w * dense_cp(d) * sizeof(_Float16)) + 255) & ~(size_t)255;
}
size_t dense_mfma_workspace_bytes(const slfp_conv2d_desc& d, int passes) {
    return 256 + (size_t)dense_planes(d, passes) * dense_plane_bytes(d);
}

template <int WM, int WN, int MT, int PASSES, int NSLOT>
static int launch_dense_tps(DenseParams& p, size_t lds, hipStream_t stream, int nwb = 2) {
    auto fn = k_dense_mfma<WM, WN, MT, PASSES, NSLOT>;
    if constexpr (PASSES == 1) { if (nwb == 3) fn = k_dense_mfma<WM, WN, MT, PASSES, NSLOT, 3>; }
    if constexpr (PASSES == 1 && NSLOT <= 8 && MT != 2) {
        // 3x3 stride 1 with one halo slice per tap: the unrolled kernel (same staging plan, same results).  Not for the MT = 2
        // tilings: two workgroups per CU cap them at 128 registers and the unrolled body spills there (measured slower).
        if (p.KH == 3 && p.KW == 3 && p.S == 1 && p.x_per_tap == 1 && ceil_div(p.x_pieces, 8) <= NSLOT && !switches().dense_generic) {
            fn = nwb == 3 ? k_dense3x3<WM, WN, MT, NSLOT, 3> : k_dense3x3<WM, WN, MT, NSLOT, 2>;
        }
    }
    const int rc = raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024);  // once per (device, kernel)
    if (rc != SLFP_OK) return rc;
    hipLaunchKernelGGL(fn, dim3(p.nblocks), dim3(kDnThreads), lds, stream, p);
    return check_launch("slfp dense MFMA conv kernel");
}

template <int WM, int WN, int MT, int PASSES>
static int launch_dense_tp(DenseParams& p, size_t lds, hipStream_t stream, int nwb) {
    const int slots = (int)ceil_div(p.x_pieces, 8);
    if (slots <= 3) return launch_dense_tps<WM, WN, MT, PASSES, 3>(p, lds, stream, nwb);
    if (slots <= 6 && p.KH == 3 && p.KW == 3 && p.S == 1) return launch_dense_tps<WM, WN, MT, PASSES, 6>(p, lds, stream, nwb);
    return launch_dense_tps<WM, WN, MT, PASSES, kMaxSlots>(p, lds, stream, nwb);
}

template <int WM, int WN, int MT>
static int launch_dense_t(DenseParams& p, size_t lds, int planes, hipStream_t stream, int nwb) {
    return planes == 2 ? launch_dense_tp<WM, WN, MT, 3>(p, lds, stream, 2) : launch_dense_tp<WM, WN, MT, 1>(p, lds, stream, nwb);
}

// ---- the weights-resident persistent form (k_dense3x3_res)
static bool dense_res_applicable(const slfp_conv2d_desc& d, int planes, int cp) {
    return planes == 1 && cp == 64 && d.kh == 3 && d.kw == 3 && d.stride_h == 1 && d.pad_h <= 1 && d.pad_w <= 1 &&
           !switches().dense_generic && switches().dense_res && switches().dense_cfg == 0;
}

template <int MT, int NSLOT>
static int launch_dense_res_t(DenseParams& p, hipStream_t stream) {
    constexpr int TH = 8 * MT, PIECES = ((TH + 2) * (kDnTW + 2) + 7) / 8;
    const size_t lds = 9 * 8192 + 2 * (size_t)PIECES * 1024 + 2 * ((kEncEntries * 8 + 63) & ~63);
    p.tiles_h = (int)ceil_div(p.Ho, TH);
    p.n_blocks = (int)ceil_div((int64_t)p.O, 64);
    const int64_t T = (int64_t)p.N * p.tiles_h * p.tiles_w;
    if (T > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "dense MFMA conv: grid too large");
    int64_t gs = std::min<int64_t>(T, std::max<int64_t>(1, device_cu_count() / p.n_blocks));   // one workgroup per CU (LDS)
    if (gs >= 8) gs -= gs % 8;
    auto fn = k_dense3x3_res<MT, NSLOT, false>;
    if (p.x32) fn = k_dense3x3_res<MT, NSLOT, true>;
    const int rc = raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024);
    if (rc != SLFP_OK) return rc;
    p.nblocks = (uint32_t)(gs * p.n_blocks);
    hipLaunchKernelGGL(fn, dim3(p.nblocks), dim3(kDnThreads), lds, stream, p);
    return check_launch("slfp dense MFMA conv kernel (resident weights)");
}

static int launch_dense_res(DenseParams& p, hipStream_t stream) {
    // 16-row tiles (two MFMA rows per W fragment read) unless they pad the image noticeably more than 8-row tiles
    const int64_t pad16 = ceil_div(p.Ho, 16) * 16, pad8 = ceil_div(p.Ho, 8) * 8;
    if (pad16 * 100 <= pad8 * 108) return launch_dense_res_t<2, 6>(p, stream);
    return launch_dense_res_t<1, 3>(p, stream);
}

// code interface: is there a dense kernel for this layer with code input / output?  (the float32-interface kernel with a decode
// pre-pass instead of the encode pre-pass and / or the code epilogue: same tilings, same results)
bool dense_codes_applicable(const slfp_conv2d_desc& d, const ConvPlan& plan, int post_flags, bool y_codes) {
    if (plan.family != kDenseMfma || plan.repad) return false;
    if (post_flags & SLFP_POST_LAYEROUT) return false;
    if (y_codes && d.c_out % 16 != 0) return false;   // a lane stores 16 consecutive channel codes
    return true;
}

int launch_dense_mfma(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const void* wblob,
                      const float* bias, const PostOp& post, float* y, void* workspace, hipStream_t stream) {
    const CodeIo io{false, false, 1.f, kFmtAct8};
    return launch_dense_mfma_io(d, plan, x, wblob, bias, post, y, workspace, io, stream);
}

int launch_dense_mfma_io(const slfp_conv2d_desc& d, const ConvPlan& plan, const void* x_any, const void* wblob,
                         const float* bias, const PostOp& post, void* y_any, void* workspace, const CodeIo& io, hipStream_t stream) {
    const float* x = reinterpret_cast<const float*>(x_any);
    float* y = reinterpret_cast<float*>(y_any);
    DenseGeom g;
    const int planes = dense_planes(d, plan.passes);
    if (!dense_choose(d, planes, plan.h_out, plan.w_out, &g)) return fail(SLFP_ERR_UNSUPPORTED, "dense MFMA conv: no tiling fits");
    if (!workspace) return fail(SLFP_ERR_BAD_ARG, "dense MFMA conv: workspace required (slfp_conv2d_workspace_bytes)");
    // k_dense3x3_res<ENCX>: float32 input of exactly one 64-channel chunk, one channel slice -> encoded where the halo is staged
    const EncArgs* tx = nullptr;
    if (!io.x_codes && planes == 1 && d.c_in == 64 && d.c_out <= 64 && switches().dense_encx && dense_res_applicable(d, planes, (int)dense_cp(d)))
        tx = act_table(d.ka, plan.fmt_act, kEncF16P);
    // ---- pass 1: encode the input once (every element is reused KH*KW * C_out times by pass 2)
    unsigned char* zero_page = reinterpret_cast<unsigned char*>(workspace);
    _Float16* xe = reinterpret_cast<_Float16*>(zero_page + 256);
    _Float16* xlo = planes == 2 ? reinterpret_cast<_Float16*>(zero_page + 256 + dense_plane_bytes(d)) : nullptr;
    const int cp = (int)dense_cp(d);
    const int64_t n_chunks16 = d.n * d.h * d.w * (cp / 8);
    const ScaleDiv sd = make_scale_div(d.ka, 4);
    const unsigned egrid = (unsigned)ceil_div(n_chunks16, 256);
    if (tx) {
        // no pre-pass
    } else if (io.x_codes) {
        const uint8_t* xc = reinterpret_cast<const uint8_t*>(x_any);
        if (plan.fmt_act == kFmtAct8)
            hipLaunchKernelGGL((k_dense_decode<kFmtAct8>), dim3(egrid), dim3(256), 0, stream, xc, xe, xlo, zero_page, n_chunks16, (int)d.c_in, cp);
        else
            hipLaunchKernelGGL((k_dense_decode<kFmtSfp7>), dim3(egrid), dim3(256), 0, stream, xc, xe, xlo, zero_page, n_chunks16, (int)d.c_in, cp);
    } else if (plan.fmt_act == kFmtAct8)
        hipLaunchKernelGGL((k_dense_encode<kFmtAct8>), dim3(egrid), dim3(256), 0, stream, x, xe, xlo, zero_page, n_chunks16, (int)d.c_in, cp, sd);
    else
        hipLaunchKernelGGL((k_dense_encode<kFmtSfp7>), dim3(egrid), dim3(256), 0, stream, x, xe, xlo, zero_page, n_chunks16, (int)d.c_in, cp, sd);
    int rc = check_launch("slfp dense encode kernel");
    if (rc != SLFP_OK) return rc;
    // ---- pass 2: implicit GEMM
    DenseParams p;
    p.xe = xe; p.xlo = xlo; p.zero_page = zero_page;
    p.w = reinterpret_cast<const _Float16*>(wblob); p.bias = bias; p.y = y; p.post = post;
    p.yc = nullptr; p.y_sgn = 0; p.y_fmt = kFmtAct8; p.enc_out.valid = 0;
    p.x32 = nullptr; p.enc_in.valid = 0;
    if (tx) { p.x32 = x; p.enc_in = enc_compact(*tx); }
    if (io.y_codes) {
        const EncArgs* t = enc_table(io.y_ka, io.y_fmt, kEncCode);
        if (!t->valid) return fail(SLFP_ERR_UNSUPPORTED, "dense MFMA conv: no code table for the consumer's scale");
        p.enc_out = *t;
        p.yc = reinterpret_cast<uint8_t*>(y_any);
        p.y = nullptr;
        p.y_sgn = post.relu ? 0 : 1;
        p.y_fmt = io.y_fmt;
    }
    p.wlo = p.w + (size_t)d.kh * d.kw * plan.k_pad * plan.n_pad;   // second plane of the blob (float32-equivalent mode)
    p.N = (int)d.n; p.H = (int)d.h; p.W = (int)d.w; p.Cp = cp; p.O = (int)d.c_out;
    p.KH = (int)d.kh; p.KW = (int)d.kw; p.S = d.stride_h; p.ph = d.pad_h; p.pw = d.pad_w;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out;
    p.IH = g.ih; p.IW = g.iw; p.n_pix = g.ih * g.iw; p.x_pieces = g.pieces; p.x_per_tap = g.per_tap;
    p.tiles_h = (int)ceil_div(p.Ho, g.cfg.wm * g.cfg.mt); p.tiles_w = (int)ceil_div(p.Wo, kDnTW);
    p.n_blocks = (int)ceil_div((int64_t)p.O, g.cfg.wn * 64);
    p.KS = (int)(plan.k_pad / 32); p.n_tiles = (int)(plan.n_pad / 16);
    p.s1 = plan.s1; p.s2 = plan.s2; p.s1x = plan.s1 * (1.0f / 256.0f);
    const int64_t nblocks = (int64_t)p.N * p.tiles_h * p.tiles_w * p.n_blocks;
    if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "dense MFMA conv: grid too large");
    p.nblocks = (uint32_t)nblocks;
    if (dense_res_applicable(d, planes, cp)) return launch_dense_res(p, stream);
    switch (g.cfg.wm * 100 + g.cfg.wn * 10 + g.cfg.mt) {
        case 244: return launch_dense_t<2, 4, 4>(p, g.lds, planes, stream, g.nwb);
        case 424: return launch_dense_t<4, 2, 4>(p, g.lds, planes, stream, g.nwb);
        case 422: return launch_dense_t<4, 2, 2>(p, g.lds, planes, stream, g.nwb);
        case 421: return launch_dense_t<4, 2, 1>(p, g.lds, planes, stream, g.nwb);
        case 812: return launch_dense_t<8, 1, 2>(p, g.lds, planes, stream, g.nwb);
        default: return launch_dense_t<8, 1, 1>(p, g.lds, planes, stream, g.nwb);
    }
}

}  // namespace slfp
