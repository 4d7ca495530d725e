// conv_pw.hip -- SLFP-quantized pointwise (1x1) convolution on the gfx950 matrix cores.
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) for 1x1 kernels,
// groups == 1 (the 13 "pw" layers of MobileNetV1, nets_imgnet/mobilenetv1.py:31; the
// 36 1x1 layers of ResNet-50; Fire squeeze/expand1x1 of SqueezeNet) and Linear_Q (:60-65).
//
// NHWC makes this a plain GEMM  Y^T[N x M] = Wq[N x K] . Xq^T[K x M]  (M = pixels) with
// K contiguous on both operands.  58 flop/B at float32 would be compute-bound on gfx950
// (ridge 19.6 flop/B), so the contraction runs on v_mfma_f32_16x16x32_f16 with float32
// accumulation and the kernels stay HBM-bound (4 B in + 4 B out per element).
//   A operand = W (rows = output channels), B operand = X (columns = pixels): each lane
//   ends up holding 4 CONSECUTIVE output channels of one pixel = one 16-byte NHWC store.
//   W is quantized ONCE per weight update into an MFMA-fragment-ordered fp16 blob
//   (slfp_conv2d_prepare_weights): fragment (n-tile, k-step) is 1 KiB, lane-linear.
//   Inside a 32-deep k-step lane (col, kq) holds k = {kq*4..kq*4+3, 16+kq*4..16+kq*4+3}
//   (any k order is legal as long as A and B agree): a lane's two 16-byte X loads then sit
//   in two 64-byte runs that the wave reads whole.
// Two kernels, chosen by the size of W:
//   k_pw_stream  K*N small enough for W to live in LDS (MobileNetV1 pw1-pw5: 71 % of the
//                pointwise bytes).  Persistent workgroups; a wave's unit of work is 16
//                pixels: it loads their K channels straight into MFMA B fragments
//                (x/Ka + SLFP encode inline, no LDS round trip for X, no barriers), sweeps
//                the W fragments out of LDS, and stores each 16-channel tile as it is done.
//   k_pw_tiled   larger W: a workgroup owns BM pixels x BN channels; X rows are encoded
//                once into a swizzled LDS tile shared by its waves (double-buffered over
//                64-deep K stages), each wave streams the W fragments of its own channels
//                straight from L2.  Ablation (512->512, 0.081 ms): X loads 0.026, stores
//                0.021, W 0.014, encode 0.011, MFMA 0: the pieces add up (serial chain per
//                workgroup), so the tile that maximises co-resident waves won (64 px x 512
//                ch, 8 waves, 126 VGPRs).  Tried and dropped: deeper X register rings, W
//                double-buffering (vmcnt retires in order: a W wait drains younger-issued X
//                loads anyway), a warp-specialised producer/consumer variant (no faster;
//                MT = 7 tiles spill at the 168-VGPR cap of a 12-wave workgroup), and an
//                "X-stationary" variant that separates the phases in time (whole 64 px x K tile
//                burst-loaded and parked in LDS as fp16, then an MFMA sweep with only L2 W loads,
//                two workgroups per CU, optionally persistent with the second resident started
//                half a period late): 0.082 ms non-persistent, 0.097 ms persistent+staggered vs
//                0.076 ms for this kernel on 512->512 @14x14 -- all CUs burst, encode, multiply
//                and store in lockstep and the start offset does not survive the first tile.
// Operand precision: SFP<3,3> values are exact in fp16 (1 pass, exact products).
// SLFP<3,4> values are 2^(m/16) multiples; fp16x1 rounds them to 11 bits (~2.5e-4
// tensor-relative error), fp16x3 splits both operands hi+lo (3 MFMA passes,
// float32-equivalent).  Both operands are pre-scaled by 2^4 (range [2, 245]: the scale
// is folded into the divisor, x/(Ka/16)) so hi is always a normal fp16 and lo keeps 2^-26
// relative precision; the 2^-8 is folded into the first epilogue scale (power-of-two
// scaling commutes with rounding), which keeps the reference's (out * Ka) * Kw roundings.
#include <cstdlib>
#include "slfp_device.hpp"
#include "slfp_enc.hpp"
#include "slfp_host.hpp"
#include "conv_pw_params.hpp"

namespace slfp {

// two float4 (k = kq*4.., 16 + kq*4..) -> one MFMA B fragment through the threshold table; NaNs not patched
__device__ __forceinline__ half8 enc_frag(const float4 a, const float4 b, const EncArgs& e, const unsigned char* tb) {
    const uint2 pa = enc4_f16_raw(a, e.r1, e.lo, e.hi, tb), pb = enc4_f16_raw(b, e.r1, e.lo, e.hi, tb);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(half8, u32x4{pa.x, pa.y, pb.x, pb.y});
}
__device__ __forceinline__ half8 enc_frag_nan(const float4 a, const float4 b, const EncArgs& e, const unsigned char* tb) {
    const uint2 pa = enc4_f16(a, e.r1, e.lo, e.hi, tb), pb = enc4_f16(b, e.r1, e.lo, e.hi, tb);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(half8, u32x4{pa.x, pa.y, pb.x, pb.y});
}

// the same for the three-pass mode: hi and lo fragments from the two tables (NaN: patched by the caller's cold branch)
__device__ __forceinline__ void enc_frag_hl(const float4 a, const float4 b, const EncArgs& e, const unsigned char* tb,
                                            const unsigned char* tl, half8& h, half8& l) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    enc2_f16_hl(a.x, a.y, e.r1, e.lo, e.hi, tb, tl, h0, l0);
    enc2_f16_hl(a.z, a.w, e.r1, e.lo, e.hi, tb, tl, h1, l1);
    enc2_f16_hl(b.x, b.y, e.r1, e.lo, e.hi, tb, tl, h2, l2);
    enc2_f16_hl(b.z, b.w, e.r1, e.lo, e.hi, tb, tl, h3, l3);
    h = __builtin_bit_cast(half8, u32x4{h0, h1, h2, h3});
    l = __builtin_bit_cast(half8, u32x4{l0, l1, l2, l3});
}
// NaN inputs of a fragment -> NaN in the hi plane, 0 in the lo plane (what (_Float16)NaN and NaN - NaN... give is NaN either way)
__device__ __forceinline__ void frag_patch_nan(const float4 a, const float4 b, half8& h) {
    const float xs[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (xs[i] != xs[i]) h[i] = (_Float16)__uint_as_float(0x7FC00000u);
}

// element offset of input pixel row m (strided 1x1 reads pixel (oh*S, ow*S))
__device__ __forceinline__ size_t x_row_offset(const PwParams& p, int64_t m) {
    if (p.S == 1) return (size_t)m * p.K;
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    const int64_t img = m / hw, r = m - img * hw;
    const int oh = (int)(r / p.Wo), ow = (int)(r - (int64_t)oh * p.Wo);
    return (size_t)(((img * p.H) + (int64_t)oh * p.S) * p.W + (int64_t)ow * p.S) * p.K;
}

template <int FMT, int PASSES>
__device__ __forceinline__ void encode4(const float4 v, const ScaleDiv sd, const uint32_t* sT, half4& h, half4& l) {
    const float v0 = quantize_scaled<FMT, 4>(v.x, sd, sT);
    const float v1 = quantize_scaled<FMT, 4>(v.y, sd, sT);
    const float v2 = quantize_scaled<FMT, 4>(v.z, sd, sT);
    const float v3 = quantize_scaled<FMT, 4>(v.w, sd, sT);
    h[0] = (_Float16)v0; h[1] = (_Float16)v1; h[2] = (_Float16)v2; h[3] = (_Float16)v3;
    if constexpr (PASSES == 3) {
        l[0] = (_Float16)(v0 - (float)h[0]); l[1] = (_Float16)(v1 - (float)h[1]);
        l[2] = (_Float16)(v2 - (float)h[2]); l[3] = (_Float16)(v3 - (float)h[3]);
    }
}

__device__ __forceinline__ half8 join(const half4 a, const half4 b) {
    return half8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// ======================================================================================
// k_pw_stream: W resident in LDS, X straight into MFMA fragments, 16-pixel work units.
// ======================================================================================
constexpr int kStreamThreads = 512;

// KS = number of 32-deep k-steps actually swept (ceil(K/32)); the blob's stride is p.KS.
// KFULL: K is a multiple of 32.  A8: K and N are even but not both multiples of 4 (ShuffleNetV2's
// 58-channel branches): pixel rows are only 8-byte aligned, so every 16-byte access becomes two
// 8-byte ones and the per-channel vectors are read with bounds.
// TAB (single-pass modes only): the quantizer is the threshold table of slfp_enc.hpp, which yields the packed fp16
// operand directly (6 VALU instructions per element instead of 22 + convert + pack).
// STG (with TAB): the outputs of two channel tiles at a time go through a 2 KiB per-wave LDS buffer and leave as 128-byte
// pieces (8 pixels x 32 channels per store instruction) instead of the accumulator layout's 64-byte pieces (16 pixels x
// 16 channels): HBM writes of 64-byte pieces run at about half the rate (profiles/micro/bw_patterns.hip).
template <int FMT, int PASSES, int KS, bool KFULL, bool A8 = false, bool TAB = false, bool STG = false>
__global__ __launch_bounds__(kStreamThreads) void k_pw_stream(const PwParams p) {
    static_assert(!STG || (TAB && !A8), "staged stores: table kernels with 16-byte channel alignment");
    constexpr int TABB = TAB ? kPwTab : 64;
    constexpr int TABL = (TAB && PASSES == 3) ? kPwTab : 16;   // three-pass table kernels: the residual plane's table
    // The quantizer's table is STATIC LDS: its address is a compile-time constant, so a lookup is `ds_read_b64 v, bin`
    // with no add of the (link-time) dynamic-LDS base -- hipcc emitted one VALU add per lookup (33 in the depthwise kernel).
    __shared__ __attribute__((aligned(16))) unsigned char stab[TABB];
    __shared__ __attribute__((aligned(16))) unsigned char stab_lo[TABL];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // everything else (sizes depend on the layer)
    uint32_t* sT = reinterpret_cast<uint32_t*>(stab);
    _Float16* wl_hi = reinterpret_cast<_Float16*>(smem);
    const int wfrags = p.n_tiles * p.KS;  // 1 KiB each
    _Float16* wl_lo = wl_hi + (size_t)wfrags * 512;
    if constexpr (TAB) enc_fill<kStreamThreads>(reinterpret_cast<uint2*>(stab), p.enc);
    else lut_fill<FMT>(sT);
    if constexpr (TAB && PASSES == 3) enc_fill_compact<kStreamThreads>(reinterpret_cast<uint2*>(stab_lo), p.enc_lo);
    // W blob -> LDS (same fragment order), 16 bytes per thread per step
    for (int i = threadIdx.x; i < wfrags * 64; i += kStreamThreads) {
        reinterpret_cast<half8*>(wl_hi)[i] = reinterpret_cast<const half8*>(p.whi)[i];
        if constexpr (PASSES == 3) reinterpret_cast<half8*>(wl_lo)[i] = reinterpret_cast<const half8*>(p.wlo)[i];
    }
    // per-channel epilogue vectors -> LDS once per workgroup: [256 * bias/s1/s2 | post scale | post shift],
    // padded to the blob's channel count.  Read from global inside the sweep, the loads (and the 12 bias
    // divisions) sit on the critical path of every 16-byte store: +12-38 % on these layers (bench.py --post).
    float* ep = reinterpret_cast<float*>(smem + (size_t)(PASSES == 3 ? 2 : 1) * wfrags * 1024);
    const int n_pad = p.n_tiles * 16;
    unsigned char* stg = reinterpret_cast<unsigned char*>(ep + 3 * n_pad) + (threadIdx.x >> 6) * 2048;   // STG: this wave's 2 KiB
    const bool has_vec = p.bias != nullptr || p.post.scale != nullptr;   // wave-uniform
    if (has_vec) {
        for (int i = threadIdx.x; i < n_pad; i += kStreamThreads) {
            const bool in = i < p.N;
            ep[i] = (p.bias && in) ? 256.f * ((p.bias[i] / p.s1) / p.s2) : 0.f;
            ep[n_pad + i] = (p.post.scale && in) ? p.post.scale[i] : 1.f;
            ep[2 * n_pad + i] = (p.post.scale && in) ? p.post.shift[i] : 0.f;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int col = lane & 15, kq = lane >> 4;
    const int64_t n_groups = (p.M + 15) >> 4;
    const int64_t waves_total = (int64_t)gridDim.x * (kStreamThreads / 64);
    const int64_t wave_id = (int64_t)blockIdx.x * (kStreamThreads / 64) + (threadIdx.x >> 6);

    for (int64_t g = wave_id; g < n_groups; g += waves_total) {
        const int64_t m = g * 16 + col;
        const bool live = m < p.M;
        // unconditional loads (rows past the end are clamped and dropped): hipcc cannot count
        // conditional loads and falls back to vmcnt(0) waits
        const float* xr = p.x + x_row_offset(p, live ? m : p.M - 1) + kq * 4;
        // ---- X: 2 x 16-byte loads per k-step, CH k-steps (up to 8 loads per lane) in flight
        // before their encodes
        constexpr int CH = KS;  // (chunking the loads made hipcc allocate MORE registers, not fewer)
        half8 xh[KS], xl[KS];
#pragma unroll
        for (int c0 = 0; c0 < KS; c0 += CH) {
            float4 raw[CH][2];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if constexpr (A8) {
                        const int k = (c0 + c) * 32 + hf * 16 + kq * 4;   // K is even: each 8-byte half is all real or all padding
                        const float* row = xr - kq * 4;
                        float2 a = *reinterpret_cast<const float2*>(row + (k < p.K ? k : p.K - 2));
                        float2 b = *reinterpret_cast<const float2*>(row + (k + 2 < p.K ? k + 2 : p.K - 2));
                        if (k >= p.K) a = make_float2(0.f, 0.f);
                        if (k + 2 >= p.K) b = make_float2(0.f, 0.f);
                        raw[c][hf] = make_float4(a.x, a.y, b.x, b.y);
                    } else if constexpr (KFULL) {
                        raw[c][hf] = ld_stream4<SLFP_NT_PW>(xr + (c0 + c) * 32 + hf * 16);
                    } else {
                        const int k = (c0 + c) * 32 + hf * 16 + kq * 4;
                        raw[c][hf] = *reinterpret_cast<const float4*>(xr + (k < p.K ? (c0 + c) * 32 + hf * 16 : p.K - 4 - kq * 4));
                        if (k >= p.K) raw[c][hf] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
            if constexpr (TAB && PASSES == 3) {
                bool any_nan = false;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    any_nan |= enc_has_nan4(raw[c][0]) | enc_has_nan4(raw[c][1]);
                    enc_frag_hl(raw[c][0], raw[c][1], p.enc, stab, stab_lo, xh[c0 + c], xl[c0 + c]);
                }
                if (__builtin_expect(any_nan, 0)) {   // NaN in -> NaN out; never taken on real activations
#pragma unroll
                    for (int c = 0; c < CH; ++c) frag_patch_nan(raw[c][0], raw[c][1], xh[c0 + c]);
                }
            } else if constexpr (TAB) {
                bool any_nan = false;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    any_nan |= enc_has_nan4(raw[c][0]) | enc_has_nan4(raw[c][1]);
                    xh[c0 + c] = enc_frag(raw[c][0], raw[c][1], p.enc, stab);
                }
                if (__builtin_expect(any_nan, 0)) {   // NaN in -> NaN out; never taken on real activations
#pragma unroll
                    for (int c = 0; c < CH; ++c) xh[c0 + c] = enc_frag_nan(raw[c][0], raw[c][1], p.enc, stab);
                }
            } else {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    half4 h0, h1, l0, l1;
                    encode4<FMT, PASSES>(raw[c][0], p.sd, sT, h0, l0);
                    encode4<FMT, PASSES>(raw[c][1], p.sd, sT, h1, l1);
                    xh[c0 + c] = join(h0, h1);
                    if constexpr (PASSES == 3) xl[c0 + c] = join(l0, l1);
                }
            }
            // keep the chunks sequential: without this hipcc hoists every load of every chunk
            // to the top and runs out of registers at K = 256
            if constexpr (KS > CH) asm volatile("" ::: "memory");
        }
        // ---- sweep the output-channel tiles
        auto tile_out = [&](int j) {
            floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
            const _Float16* wj = wl_hi + ((size_t)j * p.KS) * 512 + lane * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 wh = *reinterpret_cast<const half8*>(wj + ks * 512);
                if constexpr (PASSES == 3) {
                    const half8 wl = *reinterpret_cast<const half8*>(wj + (size_t)wfrags * 512 + ks * 512);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[ks], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[ks], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[ks], acc, 0, 0, 0);
            }
            const int n = j * 16 + kq * 4;
            float4 r;
            if (has_vec) {   // bias and/or fused BN: all three vectors from LDS
                const float4 bq = *reinterpret_cast<const float4*>(ep + n);
                const float4 sc = *reinterpret_cast<const float4*>(ep + n_pad + n);
                const float4 sh = *reinterpret_cast<const float4*>(ep + 2 * n_pad + n);
                r = epilogue(acc, bq, p.s1x, p.s2);
                if (p.post.scale) {
                    r.x = __builtin_fmaf(r.x, sc.x, sh.x); r.y = __builtin_fmaf(r.y, sc.y, sh.y);
                    r.z = __builtin_fmaf(r.z, sc.z, sh.z); r.w = __builtin_fmaf(r.w, sc.w, sh.w);
                    if (p.post.layerout) r = layerout4(r);
                }
            } else {
                r = epilogue(acc, make_float4(0.f, 0.f, 0.f, 0.f), p.s1x, p.s2);
            }
            if (p.post.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            return r;
        };
        if constexpr (STG) {
            // per-unit descriptor: rows beyond M and channels beyond N get an out-of-range offset (store dropped)
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                p.y + (size_t)g * 16 * p.N, 0, (uint32_t)(((uint64_t)(p.M - g * 16) * p.N * 4) > 0xFFFFFFFFull ? 0xFFFFFFFFull : ((uint64_t)(p.M - g * 16) * p.N * 4)), 0x00020000);
            const int spx = lane >> 3, sch = lane & 7;
            for (int j0 = 0; j0 < p.n_tiles; j0 += 2) {
                const float4 ra = tile_out(j0), rb = tile_out(j0 + 1);
                *reinterpret_cast<float4*>(stg + col * 128 + kq * 16) = ra;
                *reinterpret_cast<float4*>(stg + col * 128 + 64 + kq * 16) = rb;
                // LDS operations of one wave execute in order: the reads see the writes, the next pair's writes follow the reads
                const bool ch_ok = (j0 * 16 + sch * 4) < p.N;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = *reinterpret_cast<const u32x4*>(stg + lane * 16 + h * 1024);
                    uint32_t so = ch_ok ? (uint32_t)((spx + 8 * h) * p.N + j0 * 16 + sch * 4) * 4u : 0xFFFFFFF0u;
                    asm volatile("" : "+v"(so));
                    if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(v, ry, so, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, ry, so, 0, 0);
                }
            }
        } else {
            float* yr = p.y + (size_t)m * p.N + kq * 4;
            for (int j = 0; j < p.n_tiles; ++j) {
                const float4 r = tile_out(j);
                const int n = j * 16 + kq * 4;
                if constexpr (A8) {
                    if (live && n < p.N) *reinterpret_cast<float2*>(yr + j * 16) = make_float2(r.x, r.y);
                    if (live && n + 2 < p.N) *reinterpret_cast<float2*>(yr + j * 16 + 2) = make_float2(r.z, r.w);
                } else {
                    if (live && n < p.N) st_stream4<SLFP_NT_PW>(yr + j * 16, r);
                }
            }
        }
    }
}

// ======================================================================================
// k_pw_tiled: X via a swizzled LDS tile (encoded once), W fragments straight from L2.
// ======================================================================================

// KFULL: K is a multiple of 64 (no masking of the K tail).  Every global load in the main
// loop is UNCONDITIONAL (rows beyond M are clamped to the last row and never stored, padded
// output tiles are clamped and never stored) and the loop body has no branches: with a
// per-load `if` hipcc cannot count the loads in flight and falls back to s_waitcnt vmcnt(0)
// in the middle of the MFMA block, draining the HBM loads it has just issued (r01c ISA).
// STG: the epilogue goes through a per-wave LDS staging area so that every store instruction writes whole 256-byte runs
// (4 pixel rows x the wave's 64 adjacent channels) instead of 64-byte pieces, with the nt hint (SLFP_NT_PW_STG).
template <int FMT, int PASSES, int WM, int WN, int MT, int NT, bool KFULL, bool TAB = false, bool STG = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) <= 4 ? 2 : 4) void k_pw_tiled(const PwParams p) {
    static_assert(!STG || NT == 4, "the staged epilogue stores a wave's 64 channels per row");
    constexpr int TABB = TAB ? kPwTab : 64;
    constexpr int TABL = (TAB && PASSES == 3) ? kPwTab : 16;
    constexpr int T = 64 * WM * WN;
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int NLD = BM * 16 / T;  // float4 loads per thread per 64-deep stage
    static_assert(BM * 16 % T == 0, "staging must divide evenly");
    constexpr int XBYTES = BM * 128;

    __shared__ __attribute__((aligned(16))) unsigned char stab[TABB];      // static: constant address (see k_pw_stream)
    __shared__ __attribute__((aligned(16))) unsigned char stab_lo[TABL];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sT = reinterpret_cast<uint32_t*>(stab);
    unsigned char* xs = smem;  // [2 buffers][hi, lo][BM rows][128 B]
    if constexpr (TAB) enc_fill<T>(reinterpret_cast<uint2*>(stab), p.enc);
    else lut_fill<FMT>(sT);
    if constexpr (TAB && PASSES == 3) enc_fill_compact<T>(reinterpret_cast<uint2*>(stab_lo), p.enc_lo);

    const uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const uint32_t nb = b % p.n_blocks, mb = b / p.n_blocks;
    const int64_t m0 = (int64_t)mb * p.rb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, kq = lane >> 4;

    // ---- staging geometry: float4 #i of a thread covers row (tid>>4) + i*(T/16), k = (tid&15)*4.
    // Inside its 32-deep k-step that float4 is element half (kc&7)>>2 of lane-quarter kc&3.
    const int kc = threadIdx.x & 15;
    const int st_chunk = (kc >> 3) * 4 + (kc & 3);
    const uint32_t st_sub = (uint32_t)((kc & 7) >> 2) * 8u;
    const float* src[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = (threadIdx.x >> 4) + i * (T / 16);
        int64_t m = m0 + (row < p.rb ? row : p.rb - 1);  // padding rows of the tile re-read its last row (L1 hits) ...
        m = m < p.M ? m : p.M - 1;                       // ... and rows past the end the last pixel: computed and dropped
        src[i] = p.x + x_row_offset(p, m) + kc * 4;
    }

    float4 st[NLD];
    auto load_stage = [&](int t) {
#ifdef SLFP_ABL_NOX   // diagnostic builds only (profiles/ablate_pw.sh): wrong results, what does the kernel cost without ...
        if (t >= 2) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) asm volatile("" : "+v"(st[i].x), "+v"(st[i].y), "+v"(st[i].z), "+v"(st[i].w));
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if constexpr (KFULL) {
                st[i] = ld_stream4<SLFP_NT_PWT>(src[i] + t * 64);
            } else {
                const int k = t * 64 + kc * 4;
                const int kk = k < p.K ? t * 64 : p.K - 4 - kc * 4;  // clamp inside the row
                st[i] = *reinterpret_cast<const float4*>(src[i] + kk);
                if (k >= p.K) st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto encode_store = [&](int buf) {
        unsigned char* hi = xs + (size_t)buf * (PASSES == 3 ? 2 : 1) * XBYTES;
        unsigned char* lo = hi + XBYTES;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = (threadIdx.x >> 4) + i * (T / 16);
            const uint32_t off = lds_x_off(row, st_chunk) + st_sub;
            if constexpr (TAB && PASSES == 3) {
                uint2 ph, pl;
                enc2_f16_hl(st[i].x, st[i].y, p.enc.r1, p.enc.lo, p.enc.hi, stab, stab_lo, ph.x, pl.x);
                enc2_f16_hl(st[i].z, st[i].w, p.enc.r1, p.enc.lo, p.enc.hi, stab, stab_lo, ph.y, pl.y);
                if (__builtin_expect(enc_has_nan4(st[i]), 0)) enc_patch_nan4_f16(st[i], ph);
                *reinterpret_cast<uint2*>(hi + off) = ph;
                *reinterpret_cast<uint2*>(lo + off) = pl;
            } else if constexpr (TAB) {
#ifdef SLFP_ABL_NOENC
                uint2 pk;
                pk.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(st[i].x, st[i].y));
                pk.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(st[i].z, st[i].w));
                *reinterpret_cast<uint2*>(hi + off) = pk;
#else
                *reinterpret_cast<uint2*>(hi + off) = enc4_f16(st[i], p.enc.r1, p.enc.lo, p.enc.hi, stab);
#endif
            } else {
                half4 h, l;
                encode4<FMT, PASSES>(st[i], p.sd, sT, h, l);
                *reinterpret_cast<half4*>(hi + off) = h;
                if constexpr (PASSES == 3) *reinterpret_cast<half4*>(lo + off) = l;
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
    const size_t wplane = (size_t)(p.wlo - p.whi);
    // W fragments (fragment (ntile0 + j, k-step): 1 KiB, lane-linear; padded tiles clamp to the last real one) by buffer loads:
    // descriptor + per-tile byte offsets in scalar registers (the wave index is made uniform for the compiler), ONE vector
    // register of address (lane * 16) for all of them (8 address registers less than per-tile pointers).  A second W set
    // (W double-buffered by k-step, as k_pwc_tiled does) then fits without spills, and was measured: 118.19 vs 118.53 k images/s
    // without it (200-step A/B, 9 / 6 runs) -- on the float32 interface the X loads and the encoder set the stage time.
    const int wn_u = __builtin_amdgcn_readfirstlane(wn);
    const int ntile0u = (int)nb * (BN / 16) + wn_u * NT;
    const uint64_t wbytes = (uint64_t)p.n_tiles * p.KS * 1024;
    const uint32_t wrecs = (uint32_t)(wbytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : wbytes);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.whi), 0, wrecs, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(PASSES == 3 ? p.wlo : p.whi), 0, wrecs, 0x00020000);
    uint32_t wsoff[NT];   // scalar
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nt = ntile0u + j < p.n_tiles ? ntile0u + j : p.n_tiles - 1;
        wsoff[j] = (uint32_t)nt * (uint32_t)p.KS * 1024u;
    }
    const uint32_t wvoff = (uint32_t)lane * 16u;
    half8 wh[NT], wl[NT];
    auto load_w = [&](int kstep) {
#ifdef SLFP_ABL_NOW
        if (kstep >= 1) {
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(wh[j]));
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
            const uint32_t so = wsoff[j] + (uint32_t)kstep * 1024u;
            wh[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rwh, wvoff, so, 0));
            if constexpr (PASSES == 3) wl[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rwl, wvoff, so, 0));
        }
    };
    auto mfma_step = [&](int buf, int ks) {
        const unsigned char* hi = xs + (size_t)buf * (PASSES == 3 ? 2 : 1) * XBYTES;
        const unsigned char* lo = hi + XBYTES;
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
#ifdef SLFP_ABL_NOMFMA
                asm volatile("" :: "v"(wh[j]), "v"(xh));
#else
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh, acc[i][j], 0, 0, 0);
#endif
            }
        }
    };

    SLFP_STAMP(0);
    __syncthreads();  // LUT visible
    SLFP_STAMP(1);
    load_stage(0);
    encode_store(0);
    load_stage(KT > 1 ? 1 : 0);
    __syncthreads();
    SLFP_STAMP(2);

    for (int t = 0; t + 1 < KT; ++t) {  // branch-free body
        const int buf = t & 1;
#ifdef SLFP_PW_STAMPS2   // diagnostic builds only: fine-grained stamps of stage 3 in slots 3..7 (profiles/stamps_fine.py)
        if (t == 3) SLFP_STAMP(3);
#endif
        load_w(t * 2);
        encode_store(buf ^ 1);                       // stage t+1 (fetched one stage ago) -> the other LDS buffer
        load_stage(t + 2 < KT ? t + 2 : KT - 1);     // stage t+2's HBM loads fly during the MFMAs (last: harmless re-read)
#ifdef SLFP_PW_STAMPS2
        if (t == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SLFP_STAMP(4); }
#endif
        mfma_step(buf, 0);
#ifdef SLFP_PW_STAMPS2
        if (t == 3) { asm volatile("s_nop 0" : "+v"(acc[0][0]), "+v"(acc[3][3])); SLFP_STAMP(5); }
#endif
        load_w(t * 2 + 1);
        mfma_step(buf, 1);
#ifdef SLFP_PW_STAMPS2
        if (t == 3) { asm volatile("s_nop 0" : "+v"(acc[0][0]), "+v"(acc[3][3])); SLFP_STAMP(6); }
#endif
        __syncthreads();
#ifdef SLFP_PW_STAMPS2
        if (t == 3) SLFP_STAMP(7);
#else
        SLFP_STAMP(3 + (t < 8 ? t : 8));
#endif
    }
    SLFP_STAMP(12);
    {   // last stage: nothing left to prefetch
        const int t = KT - 1, buf = t & 1;
        load_w(t * 2);
        mfma_step(buf, 0);
        load_w(t * 2 + 1);
        mfma_step(buf, 1);
    }

    // fused BN/ReLU: this workgroup's BN-channel slices of scale / shift go through the (now free) X tile
    // in LDS, so the store loop reads them with two ds_read_b128 instead of two global loads per store
    // (those sat on the critical path of every store: +13 % on these layers)
    const int n_lo = (int)nb * BN;
    float* lsc = reinterpret_cast<float*>(xs);   // LDS
    float* lsh = lsc + BN;
    if (p.post.scale) {
        __syncthreads();   // every wave is done with the X tile
        for (int i = threadIdx.x; i < BN; i += T) {
            const bool in = n_lo + i < p.N;
            lsc[i] = in ? p.post.scale[n_lo + i] : 1.f;
            lsh[i] = in ? p.post.shift[n_lo + i] : 0.f;
        }
        __syncthreads();
    }
    SLFP_STAMP(13);
    if constexpr (STG) {
        unsigned char* stg = xs + 2 * (PASSES == 3 ? 2 : 1) * XBYTES + wave * (16 * kStgRow);   // private to this wave
        const uint64_t left = (uint64_t)(p.M - m0) * p.N * 4;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)m0 * p.N, 0,
                                                                             (uint32_t)(left > 0xFFFFFFFFull ? 0xFFFFFFFFull : left), 0x00020000);
        const int srow = lane >> 4, sch = lane & 15;
        const int n_st = ntile0 * 16 + sch * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (ntile0 + j) * 16 + kq * 4;
                const bool in = n < p.N;   // channel tiles past C_out: computed from clamped weights, never stored
                float4 r = epilogue(acc[i][j], in ? bias_q256(p, n) : make_float4(0.f, 0.f, 0.f, 0.f), p.s1x, p.s2);
                if (p.post.scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(lsc + (in ? n - n_lo : 0));
                    const float4 sh = *reinterpret_cast<const float4*>(lsh + (in ? n - n_lo : 0));
                    r.x = __builtin_fmaf(r.x, sc.x, sh.x); r.y = __builtin_fmaf(r.y, sc.y, sh.y);
                    r.z = __builtin_fmaf(r.z, sc.z, sh.z); r.w = __builtin_fmaf(r.w, sc.w, sh.w);
                    if (p.post.layerout) r = layerout4(r);
                }
                if (p.post.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                *reinterpret_cast<float4*>(stg + col * kStgRow + j * 64 + kq * 16) = r;
            }
            // LDS operations of one wave execute in order: the reads see the writes, the next tile's writes follow the reads
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const int row = (wm * MT + i) * 16 + h * 4 + srow;
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (h * 4 + srow) * kStgRow + sch * 16);
                const bool ok = row < p.rb && m0 + row < p.M && n_st < p.N;
                uint32_t so = ok ? (uint32_t)(row * p.N + n_st) * 4u : 0xFFFFFFF0u;
#ifdef SLFP_ABL_NOST
                so = 0xFFFFFFF0u;
#endif
                asm volatile("" : "+v"(so));
                if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(v, ry, so, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, ry, so, 0, 0);
            }
        }
        SLFP_STAMP(14);
#ifdef SLFP_PW_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SLFP_STAMP(15);
#endif
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (ntile0 + j) * 16 + kq * 4;
        if (n >= p.N) continue;
        const float4 bq = bias_q256(p, n);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = (wm * MT + i) * 16 + col;
            const int64_t m = m0 + row;
            if (m >= p.M || row >= p.rb) continue;
            float4 r = epilogue(acc[i][j], bq, p.s1x, p.s2);
            if (p.post.scale) {
                const float4 sc = *reinterpret_cast<const float4*>(lsc + (n - n_lo));
                const float4 sh = *reinterpret_cast<const float4*>(lsh + (n - n_lo));
                r.x = __builtin_fmaf(r.x, sc.x, sh.x); r.y = __builtin_fmaf(r.y, sc.y, sh.y);
                r.z = __builtin_fmaf(r.z, sc.z, sh.z); r.w = __builtin_fmaf(r.w, sc.w, sh.w);
                if (p.post.layerout) r = layerout4(r);
            }
            if (p.post.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            st_stream4<SLFP_NT_PW>(p.y + (size_t)m * p.N + n, r);
        }
    }
}


// ---------------------------------------------------------------------------- launch
// once per (device, kernel), not per launch (round 1 re-armed the attribute on every launch)
static int set_lds_limit(const void* fn, size_t lds) { return raise_lds_limit(fn, lds); }

template <int FMT, int PASSES, int WM, int WN, int MT, int NT, bool KFULL>
static int launch_tiled_k(PwParams& p, hipStream_t stream) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, T = 64 * WM * WN;
    p.n_blocks = (uint32_t)ceil_div((int64_t)p.N, BN);
    // (Workgroups that own fewer rows than their tile -- 49 = a quarter image instead of 64, for an integral number of
    // rounds on the 512 resident slots -- were measured: no gain, 61-64 vs 60 us on 512->512.  A workgroup's time is set
    // by streaming all of W from L2, not by its rows.)
    p.rb = BM;
    p.m_blocks = (uint32_t)ceil_div(p.M, p.rb);
    const int64_t nblocks = (int64_t)p.m_blocks * p.n_blocks;
    if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "pointwise: grid too large");
    p.nblocks = (uint32_t)nblocks;
    {
        if (p.enc.valid && (PASSES == 1 || p.enc_lo.valid)) {
            if constexpr (NT == 4) {
                if (!switches().pw_nostg) {   // experiment switch (slfp_host.hpp), read once at load
                    const size_t lds = (size_t)2 * (PASSES == 3 ? 2 : 1) * BM * 128 + (size_t)(T / 64) * 16 * kStgRow;   // dynamic part (the tables are static LDS)
                    auto fn = k_pw_tiled<FMT, PASSES, WM, WN, MT, NT, KFULL, true, true>;
                    int rc = set_lds_limit(reinterpret_cast<const void*>(fn), lds);
                    if (rc != SLFP_OK) return rc;
                    hipLaunchKernelGGL(fn, dim3(p.nblocks), dim3(T), lds, stream, p);
                    return check_launch("slfp pointwise (tiled) kernel");
                }
            }
            const size_t lds = (size_t)2 * (PASSES == 3 ? 2 : 1) * BM * 128;
            auto fn = k_pw_tiled<FMT, PASSES, WM, WN, MT, NT, KFULL, true>;
            int rc = set_lds_limit(reinterpret_cast<const void*>(fn), lds);
            if (rc != SLFP_OK) return rc;
            hipLaunchKernelGGL(fn, dim3(p.nblocks), dim3(T), lds, stream, p);
            return check_launch("slfp pointwise (tiled) kernel");
        }
    }
    const size_t lds = (size_t)2 * (PASSES == 3 ? 2 : 1) * BM * 128;   // dynamic part (the lookup table is static LDS)
    auto fn = k_pw_tiled<FMT, PASSES, WM, WN, MT, NT, KFULL>;
    int rc = set_lds_limit(reinterpret_cast<const void*>(fn), lds);
    if (rc != SLFP_OK) return rc;
    hipLaunchKernelGGL(fn, dim3(p.nblocks), dim3(T), lds, stream, p);
    return check_launch("slfp pointwise (tiled) kernel");
}

template <int FMT, int PASSES, int WM, int WN, int MT, int NT>
static int launch_tiled(PwParams& p, hipStream_t stream) {
    if (p.K % 64 == 0) return launch_tiled_k<FMT, PASSES, WM, WN, MT, NT, true>(p, stream);
    return launch_tiled_k<FMT, PASSES, WM, WN, MT, NT, false>(p, stream);
}

template <int FMT, int PASSES, int KS>
static int launch_stream_ks(PwParams& p, hipStream_t stream) {
    const bool tab = p.enc.valid != 0 && (PASSES == 1 || p.enc_lo.valid != 0);
    // staged 128-byte stores (with the nt hint) pay where stores dominate and follow each other closely (K <= 64: pw1
    // -14 %, pw2 -13 %) and at K = 256 (256->256 @28: 113 -> 97-105 us).  At K = 128 the layer itself loses 5-8 % (profiles/
    // variants.py --var SLFP_PW_STG_MAXKS=4 / 8) but the depthwise layer that reads its output gains more (whole step, 200-step
    // same-box A/B profiles/ab_env_long.sh, 3 rounds: depthwise family 1.035 -> 0.997 ms, pointwise 1.098 -> 1.112, step -1 %):
    // staged everywhere since round 3.  K = 160 / 192 (KS 5..6: no MobileNetV1 layer) keep the direct stores they were measured with
    const int mk = switches().pw_stg_maxks;   // experiment switch (slfp_host.hpp), read once at load
    const bool stg_ks = mk >= 0 ? KS <= mk : (KS <= 4 || KS == 8);
    const bool stg = tab && stg_ks && !(p.K % 4 || p.N % 4) && !switches().pw_nostg;
    const size_t lds = (size_t)(PASSES == 3 ? 2 : 1) * p.n_tiles * p.KS * 1024 + (size_t)3 * p.n_tiles * 16 * sizeof(float) +
                       (stg ? (size_t)(kStreamThreads / 64) * 2048 : 0);   // dynamic part; the table (2 KiB / 64 B) is static LDS
    const size_t lds_total = lds + (tab ? kPwTab * (PASSES == 3 ? 2 : 1) : 64);
    auto fn = (p.K % 4 || p.N % 4) ? k_pw_stream<FMT, PASSES, KS, false, true>
              : (p.K % 32 == 0)    ? k_pw_stream<FMT, PASSES, KS, true> : k_pw_stream<FMT, PASSES, KS, false>;
    {
        if (tab) fn = (p.K % 4 || p.N % 4) ? k_pw_stream<FMT, PASSES, KS, false, true, true>
                      : (p.K % 32 == 0)    ? k_pw_stream<FMT, PASSES, KS, true, false, true> : k_pw_stream<FMT, PASSES, KS, false, false, true>;
        if (stg) fn = (p.K % 32 == 0) ? k_pw_stream<FMT, PASSES, KS, true, false, true, true> : k_pw_stream<FMT, PASSES, KS, false, false, true, true>;
    }
    int rc = set_lds_limit(reinterpret_cast<const void*>(fn), lds);
    if (rc != SLFP_OK) return rc;
    // persistent grid: as many workgroups per CU as LDS and registers allow (<= 4), on every CU of the device
    int per_cu = resident_blocks_per_cu(reinterpret_cast<const void*>(fn), kStreamThreads, lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    (void)lds_total;
    const int64_t groups = (p.M + 15) / 16;
    int64_t grid = (int64_t)device_cu_count() * per_cu;
    const int64_t need = ceil_div(groups, kStreamThreads / 64);
    if (grid > need) grid = need;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kStreamThreads), lds, stream, p);
    return check_launch("slfp pointwise (stream) kernel");
}

// CAN the LDS-resident stream kernel take this layer (W next to the tables and the staging area)?  Also what the planner asks
// for channel counts that are even but not multiples of 4 (only that kernel has the 8-byte access path).
bool pointwise_stream_fits(int64_t k_pad, int64_t n_pad, int passes) {
    const size_t w = (size_t)k_pad * n_pad * 2 * (passes == 3 ? 2 : 1);
    return k_pad <= 256 && w <= (size_t)128 * 1024;
}
// SHOULD it?  Above SLFP_PW_STREAM_MAX_KB (default 30 KiB of W) the tiled kernel is faster on whole steps (codec.hip: Switches).
static bool stream_fits(const ConvPlan& plan, int passes, const PwParams& p) {
    if (!pointwise_stream_fits(plan.k_pad, plan.n_pad, passes)) return false;
    if (p.K % 4 || p.N % 4) return true;   // 8-byte access path: the stream kernel only
    const size_t w = (size_t)plan.k_pad * plan.n_pad * 2 * (passes == 3 ? 2 : 1);
    // three-pass mode (two planes, 3 MFMAs per tile): the stream kernel stays ahead up to 100 KiB (99.4 vs 95.6 k images/s)
    return w <= (size_t)(passes == 3 ? 100 : switches().pw_stream_max_kb) * 1024;
}

template <int FMT, int PASSES>
static int launch_pw(PwParams& p, const ConvPlan& plan, hipStream_t stream) {
    if (stream_fits(plan, PASSES, p)) {
        switch ((p.K + 31) / 32) {  // k-steps that hold real channels (the blob is zero-padded to p.KS)
            case 1: return launch_stream_ks<FMT, PASSES, 1>(p, stream);
            case 2: return launch_stream_ks<FMT, PASSES, 2>(p, stream);
            case 3: return launch_stream_ks<FMT, PASSES, 3>(p, stream);
            case 4: return launch_stream_ks<FMT, PASSES, 4>(p, stream);
            case 5: case 6: return launch_stream_ks<FMT, PASSES, 6>(p, stream);
            case 7: case 8: return launch_stream_ks<FMT, PASSES, 8>(p, stream);
            default: break;
        }
    }
    if (p.K % 4 || p.N % 4) return fail(SLFP_ERR_UNSUPPORTED, "pointwise: channel counts not a multiple of 4 need the stream kernel");
    if constexpr (PASSES == 3) {
        if (p.N > 128) return launch_tiled<FMT, 3, 1, 4, 4, 4>(p, stream);  // 64 px x 256 ch
        if (p.N > 64) return launch_tiled<FMT, 3, 2, 2, 2, 4>(p, stream);   // 64 px x 128 ch
        return launch_tiled<FMT, 3, 4, 1, 1, 4>(p, stream);                  // 64 px x  64 ch
    } else {
        // widest channel tile: fewest re-reads / re-encodes of X.  (Narrower tiles for the small-M, deep-K
        // layers of ResNet-50's last stages -- 49-196 pixel tiles at batch 64 -- were measured: more
        // workgroups, but the redundant encode costs more than the idle CUs did.)
        if (p.N > 256) return launch_tiled<FMT, 1, 1, 8, 4, 4>(p, stream);  // 64 px x 512 ch, 8 waves, 2 workgroups/CU
        if (p.N > 128) return launch_tiled<FMT, 1, 1, 4, 4, 4>(p, stream);  // 64 px x 256 ch
        if (p.N > 64) return launch_tiled<FMT, 1, 2, 2, 2, 4>(p, stream);   // 64 px x 128 ch
        return launch_tiled<FMT, 1, 4, 1, 1, 4>(p, stream);                  // 64 px x  64 ch
    }
}

int launch_pointwise(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const void* wfrag,
                     const float* bias, const PostOp& post, float* y, hipStream_t stream) {
    PwParams p;
    p.post = post;
    p.x = x; p.bias = bias; p.y = y;
    p.K = (int)d.c_in; p.N = (int)d.c_out;
    p.KS = (int)(plan.k_pad / 32);
    p.n_tiles = (int)(plan.n_pad / 16);
    p.whi = reinterpret_cast<const _Float16*>(wfrag);
    p.wlo = p.whi + (size_t)plan.n_pad * plan.k_pad;
    p.H = (int)d.h; p.W = (int)d.w; p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out; p.S = d.stride_h;
    p.M = d.n * plan.h_out * plan.w_out;
    p.sd = make_scale_div(d.ka, 4);  // x / (Ka/16) == 16 * (x / Ka)
#ifdef SLFP_PW_STAMPS
    { const char* e = getenv("SLFP_PW_DBG"); p.dbg = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 16)) : nullptr; }
#endif
    p.enc.valid = 0;
    p.enc_lo.valid = 0;
    if (!switches().pw_notab) {   // experiment switch (slfp_host.hpp), read once at load
        if (const EncArgs* t = act_table(d.ka, plan.fmt_act, kEncF16P)) p.enc = *t;
        if (plan.passes == 3 && p.enc.valid) {   // three-pass mode: the residual plane's table of the same scale (round 3)
            if (const EncArgs* t = act_table(d.ka, plan.fmt_act, kEncF16LO)) p.enc_lo = enc_compact(*t);
            else p.enc.valid = 0;
        }
    }
    p.s1 = plan.s1; p.s2 = plan.s2; p.s1x = plan.s1 * (1.0f / 256.0f);
    {   // staged stores always carry the nt hint: a size threshold as in conv_dw2.hip (plain stores for outputs that fit the
        // Infinity Cache) measured 1.5 % SLOWER on the whole net here (profiles/ab_env_wn.sh); SLFP_PW_NT_MIN_MB: experiment switch
        const int64_t min_mb = switches().pw_nt_min_mb;
        p.nt_out = ((SLFP_NT_PW_STG & 2) && p.M * p.N * 4 >= (min_mb << 20)) ? 1 : 0;
    }
    if (plan.fmt_act == kFmtSfp7) return launch_pw<kFmtSfp7, 1>(p, plan, stream);  // exact in fp16
    if (plan.passes == 3) return launch_pw<kFmtAct8, 3>(p, plan, stream);
    return launch_pw<kFmtAct8, 1>(p, plan, stream);
}

}  // namespace slfp
