// slfp_device.hpp -- device-side SLFP<3,4> / SFP<3,3> codec for gfx950 (CDNA4).
//
// The reference computes its fake-quantizers with ~25 float32 ATen passes per tensor
// (utils/sfp_quant.py:32-47 weights, :80-96 activations, :14-30/:63-78 SFP<3,3>).  The
// result is a pure function of the float32 bit pattern, so here it is one short integer
// routine that every kernel inlines on its load path (~17 VALU instructions + 1 LDS read):
//   - exponent extract  = the biased-exponent field after RNE of the mantissa to 4 (3) bits
//                         (the rounding carry walks into the exponent field by itself),
//   - mantissa lookup   = a 16-entry LDS table indexed by the rounded mantissa; for
//                         activations the table already holds 2^(L[lin]/16), i.e. the
//                         reference's linear-round-then-log double rounding (sfp_quant.py:88-89),
//   - the ordered overrides (tiny -> +-1e-10, [1/16,1/8) -> 1/8, clamp) as max/selects.
// `x / Ka` (utils/conv2d_func.py:21) is an IEEE float32 division; with Ka a per-layer
// constant it is computed by the same Newton/FMA correction sequence the hardware
// division macro uses, minus the v_rcp and the range scaling (see ScaleDiv below).
// Bit-exactness against the reference is pinned by tests/golden (exhaustive sweeps).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slfp {

// Cache policy of the once-through activation streams, per kernel family (bit 0: loads, bit 1: stores carry the
// `nt` hint).  Measured in the bench's cold order (profiles/ab_nt.sh, ab_nt2.sh; MobileNetV1-224, batch 256):
// depthwise 1.096 -> 1.00 ms per step with nt stores (nt loads as well: the depthwise kernels gain another 3 % but the
// pointwise kernels that run after them lose 4 %, profiles/ab_flags.sh), stem 0.133 -> 0.121; pointwise: the hint pays only where a wave
// stores whole 128-byte lines (the staged stores of k_pw_stream: 240 -> 219 us on 32->64 @112) and costs 5-25 % on the
// 64-byte-piece stores and on the loads; slfp_quantize_f32 4.93 -> 5.59 TB/s with nt stores; dense (VGG-16): no effect.
#ifndef SLFP_NT_DW
#define SLFP_NT_DW 2
#endif
#ifndef SLFP_NT_STEM
#define SLFP_NT_STEM 3
#endif
#ifndef SLFP_NT_PW
#define SLFP_NT_PW 0
#endif
#ifndef SLFP_NT_PWT
#define SLFP_NT_PWT 0
#endif
#ifndef SLFP_NT_PW_STG
#define SLFP_NT_PW_STG 2
#endif
#ifndef SLFP_NT_STEM_MFMA
#define SLFP_NT_STEM_MFMA 0
#endif
#ifndef SLFP_NT_CODEC
#define SLFP_NT_CODEC 2
#endif
#ifndef SLFP_NT_DENSE
#define SLFP_NT_DENSE 0
#endif
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
template <int POLICY>
__device__ __forceinline__ float4 ld_stream4(const float* p) {
    if constexpr (POLICY & 1) {
        const nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}
template <int POLICY>
__device__ __forceinline__ void st_stream4(float* p, const float4 r) {
    if constexpr (POLICY & 2) {
        __builtin_nontemporal_store(nt_f32x4{r.x, r.y, r.z, r.w}, reinterpret_cast<nt_f32x4*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = r;
    }
}


constexpr int kFmtAct8 = 0;  // quantize_act(8)     utils/sfp_quant.py:80-96
constexpr int kFmtW8 = 1;    // quantize_weight(8)  utils/sfp_quant.py:32-47
constexpr int kFmtSfp7 = 2;  // quantize_*(7)       utils/sfp_quant.py:14-30, 63-78
constexpr int kFmtMask = 3;
constexpr int kFmtExt = 4;

constexpr uint32_t kBitsTiny = 0x2EDBE6FFu;    // float32(1e-10)    sfp_quant.py:43,92
constexpr uint32_t kBitsMin = 0x3D800000u;     // 0.0625
constexpr uint32_t kBitsEighth = 0x3E000000u;  // 0.125
constexpr uint32_t kBitsClamp8 = 0x4175257Au;  // float32(15.32165) sfp_quant.py:46,95
constexpr uint32_t kBitsClamp7 = 0x41700000u;  // 15.0              sfp_quant.py:29,77
constexpr uint32_t kBitsQNaN = 0x7FC00000u;

// mantissa field of 2^(k/16) rounded to float32 (what the reference's pow(2, e + k/16) yields)
static __device__ const uint32_t kT16[16] = {
    0x000000u, 0x05AAC3u, 0x0B95C2u, 0x11C3D3u, 0x1837F0u, 0x1EF532u, 0x25FED7u, 0x2D583Fu,
    0x3504F3u, 0x3D08A4u, 0x45672Au, 0x4E248Cu, 0x5744FDu, 0x60CCDFu, 0x6AC0C7u, 0x75257Du};

// LDS image of the lookup table: 16 dwords in 16 distinct banks, so any per-lane index is
// conflict-free.  Activations index it with the LINEAR 4-bit mantissa `lin`; the entry is
// the mantissa of 2^(L[lin]/16) with L = [0,1,3,4,...,14,15,15] (log code 2 unreachable,
// lin 14 and 15 collide -- SURVEY 8a); weights / decode index it with the log code itself.
template <int FMT>
__device__ __forceinline__ void lut_fill(uint32_t* sT) {
    if (threadIdx.x < 16) {
        uint32_t l = threadIdx.x;
        if (FMT == kFmtAct8) l = l + (l >= 2u ? 1u : 0u) - (l >= 15u ? 1u : 0u);
        sT[threadIdx.x] = kT16[l];
    }
}

// Weight quantizer: #thresholds <= mantissa field, thresholds of round(16*log2(m)) as the
// reference's float32 log2 places them (k=5 is 1 ULP below the exact value).
__device__ __forceinline__ uint32_t w8_log_mantissa(uint32_t f) {
    uint32_t m = 0;
    m += f >= 0x02CD87u; m += f >= 0x08980Fu; m += f >= 0x0EA43Au; m += f >= 0x14F4F0u;
    m += f >= 0x1B8D3Au; m += f >= 0x227043u; m += f >= 0x29A15Bu; m += f >= 0x3123F6u;
    m += f >= 0x38FBB0u; m += f >= 0x412C4Du; m += f >= 0x49B9BEu; m += f >= 0x52A81Eu;
    m += f >= 0x5BFBB8u; m += f >= 0x65B907u; m += f >= 0x6FE4BAu; m += f >= 0x7A83B3u;
    return m;
}

// d = (a & mask) | c in one VALU instruction (hipcc emits v_and + v_or for the C expression).
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) {
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(mask), "v"(c));
    return d;
}

// bits [off, off+width) of x in one VALU instruction (hipcc lowers __builtin_amdgcn_ubfe with
// constant operands back to shift + and).
template <int OFF, int WIDTH>
__device__ __forceinline__ uint32_t bfe(uint32_t x) {
    uint32_t d;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "n"(OFF), "n"(WIDTH));
    return d;
}

// float32 bits of 2^ESH * Q_FMT(q), given u = bits(2^ESH * q) and ux = bits(x), the value
// q was computed from (q = x / Ka with Ka > 0, so sign(q) == sign(x)).  The sign and the NaN
// test come from x: an infinite or overflowing x turns into NaN inside the FMA division
// chain (inf - inf), and must still land in the clamp class like the reference's inf / Ka.
// ESH = 0 everywhere except the pointwise MFMA path, which works on 16*q (ESH = 4) so that
// the fp16 operands are always normal; scaling by a power of two commutes with every step.
// `sT` must have been filled by lut_fill<FMT> (lut_fill<kFmtW8> for W8).
// Instruction budget (ACT8): and, bfe, add3, bfe, lshl (LUT address), ds_read, and_or, max,
// 4 x (cmp + cndmask) [clamp, tiny, zero, NaN], bfi = 16 VALU + 1 LDS; the range tests are
// float compares on |q| (the abs modifier is free).
template <int FMT, int ESH = 0>
__device__ __forceinline__ uint32_t quant_bits(uint32_t u, uint32_t ux, const uint32_t* __restrict__ sT) {
    constexpr uint32_t E = (uint32_t)ESH << 23;
    const float aq = fabsf(__uint_as_float(u));  // |q| is a free source modifier on the float compares
    uint32_t v;
    bool clamp;
    if constexpr (FMT == kFmtSfp7) {
        // RNE of the mantissa to 3 bits; (1 + m/8) * 2^E is then just the rounded pattern
        v = (u + 0x7FFFFu + bfe<20, 1>(u)) & 0x7FF00000u;
        clamp = !(aq < __uint_as_float(kBitsClamp7 + E));  // >= 15 (true for NaN as well)
        v = clamp ? kBitsClamp7 + E : v;
    } else if constexpr (FMT == kFmtAct8) {
        // RNE to 4 bits (sfp_quant.py:88), carry -> exponent; done on the SIGNED pattern (the carry
        // cannot reach bit 31 for finite q) and the sign is stripped by the and_or mask
        const uint32_t t = u + 0x3FFFFu + bfe<19, 1>(u);
        v = and_or(t, 0x7F800000u, sT[bfe<19, 4>(t)]);    // log converter folded into the table (:89)
        clamp = !(aq <= __uint_as_float(kBitsClamp8 + E));                  // > 15.32165 (true for NaN as well)
        v = clamp ? kBitsClamp8 + E : v;
    } else {
        const uint32_t a = u & 0x7FFFFFFFu;
        const uint32_t idx = ((a >> 23) << 4) + w8_log_mantissa(a & 0x7FFFFFu);  // m == 16 carries
        v = ((idx >> 4) << 23) | sT[idx & 15u];
        clamp = !(aq <= __uint_as_float(kBitsClamp8 + E));
        v = clamp ? kBitsClamp8 + E : v;
    }
    v = v < kBitsEighth + E ? kBitsEighth + E : v;  // [1/16, 1/8) -> 1/8 (the formula gives <= 1/8 there)
    v = aq < __uint_as_float(kBitsMin + E) ? (ESH == 0 ? kBitsTiny : 0x30DBE6FFu /* 16e-10 */) : v;
    float r = copysignf(__uint_as_float(v), __uint_as_float(ux));  // v_bfi: sign of the input
    r = aq == 0.0f ? 0.0f : r;                                     // torch.sign(+-0) == 0 -> +0 (also on underflow, as x/Ka)
    const float x = __uint_as_float(ux);
    r = x != x ? __uint_as_float(kBitsQNaN) : r;                   // NaN in -> NaN out
    return __float_as_uint(r);
}

// canonical (or extended) code byte of Q_FMT(q); u = bits(q).  Not on the hot path.
template <int FMT>
__device__ __forceinline__ uint32_t quant_code(uint32_t u, uint32_t ux, bool ext) {
    const uint32_t a = u & 0x7FFFFFFFu;
    constexpr int MB = (FMT == kFmtSfp7) ? 3 : 4;
    const uint32_t sc = (ux >> 31) << (MB + 3);
    uint32_t idx;
    if constexpr (FMT == kFmtSfp7) {
        idx = (a + 0x7FFFFu + ((a >> 20) & 1u)) >> 20;
    } else if constexpr (FMT == kFmtAct8) {
        const uint32_t lin = (a + 0x3FFFFu + ((a >> 19) & 1u)) >> 19;
        const uint32_t l = lin & 15u;
        idx = lin + (l >= 2u ? 1u : 0u) - (l >= 15u ? 1u : 0u);
    } else {
        idx = ((a >> 23) << 4) + w8_log_mantissa(a & 0x7FFFFFu);
    }
    uint32_t c = idx - (123u << MB);
    if constexpr (FMT == kFmtSfp7) {
        c = a >= kBitsClamp7 ? 0x3Fu : c;
    } else {
        c = a > kBitsClamp8 ? (ext ? 0x02u : 0x7Fu) : c;
    }
    c = a < kBitsEighth ? (1u << MB) : c;
    c = a < kBitsMin ? 0u : c;
    c |= sc;
    c = a == 0u ? (ext ? 1u : 0u) : c;
    c = (ux & 0x7FFFFFFFu) > 0x7F800000u ? 0u : c;
    return c & 0xFFu;
}

// code byte -> float32 bits; sT filled by lut_fill<kFmtW8> (identity table).
template <int FMT>
__device__ __forceinline__ uint32_t decode_bits(uint32_t code, bool ext, const uint32_t* __restrict__ sT) {
    constexpr int MB = (FMT == kFmtSfp7) ? 3 : 4;
    const uint32_t s = ((code >> (MB + 3)) & 1u) << 31;
    const uint32_t mag = code & ((1u << (MB + 3)) - 1u);
    uint32_t v;
    if constexpr (FMT == kFmtSfp7) {
        v = (mag + (123u << 3)) << 20;
    } else {
        const uint32_t idx = mag + (123u << 4);
        v = ((idx >> 4) << 23) | sT[idx & 15u];
        v = (ext && mag == 2u) ? kBitsClamp8 : v;
    }
    v = mag == 0u ? kBitsTiny : v;
    v |= s;
    v = (ext && code == 1u) ? 0u : v;
    return v;
}

// ---- x / Ka with Ka a per-launch constant -------------------------------------------------
// RN(x / d) for every float32 x whose quotient is a normal number, from the host-computed,
// correctly rounded reciprocal r = RN(1/d):
//     q0 = RN(x*r);  q1 = fma(fma(-d,q0,x), r, q0);  q = fma(fma(-d,q1,x), r, q1)
// This is the residual-correction chain of the IEEE division expansion without v_rcp_f32
// and without the v_div_scale/v_div_fixup range handling.  q0 is within 2 ulp, the first
// correction makes q1 faithful (its exact value q0 + (x - d*q0)*r differs from x/d by
// (1 - d*r)(q0 - x/d), a second-order term), and by Markstein's theorem one more
// correction of a faithful quotient with the correctly rounded reciprocal is the correctly
// rounded quotient.  Quotients outside the normal range only ever land in the quantizer's
// "tiny" or "clamp" classes, whose boundaries (1/16, 15.32165) are far inside it; 0 and NaN
// propagate through the FMAs, inf (or an overflowing x*r) becomes NaN and is classified by
// quant_bits from the ORIGINAL x.  The host (make_scale_div) refuses divisors
// outside [1e-30, 1e30].  Equivalence with IEEE `/` is also checked exhaustively on the
// device over all 2^32 inputs (slfp_debug_div_mismatches; tests/test_gpu_parity.py).
struct ScaleDiv {
    float d;  // the divisor (Ka, or Ka/16 on the MFMA path)
    float r;  // RN(1/d)
};

__device__ __forceinline__ float div_const(float x, const ScaleDiv s) {
    const float q0 = x * s.r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-s.d, q0, x), s.r, q0);
    return __builtin_fmaf(__builtin_fmaf(-s.d, q1, x), s.r, q1);
}

// Optional per-output-channel post-op applied AFTER the reference's (out * Ka) * Kw roundings:
// y = max(r * scale[c] + shift[c], 0).  With scale = gamma / sqrt(var + eps) and
// shift = beta - mean * scale this is the eval-mode BatchNorm2d + ReLU that follows every
// Conv2d_Q in the reference nets (nets_imgnet/mobilenetv1.py:24-41), fused into the epilogue
// (SURVEY 8f rank 1).  scale == nullptr: no affine; relu == 0: no clamp.
struct PostOp {
    const float* scale;
    const float* shift;
    int relu;
    int layerout;  // only with scale/shift: SFP<4,4> layer-output quantizer between the affine and the ReLU
};

// quantize_layerout(k <= 8).forward (utils/sfp_quant.py:108-127): SFP<4,4> layer-output quantizer.
// The reference's `2^(-8)` is integer XOR, so only three things are live: RNE to 5 significant
// bits at every exponent (denormals included), the >= 248 clamp, and NaN for an exact zero.
__device__ __forceinline__ uint32_t layerout_bits(uint32_t u) {
    const uint32_t a = u & 0x7FFFFFFFu, s = u & 0x80000000u;
    uint32_t v = (a + 0x3FFFFu + ((a >> 19) & 1u)) & 0xFFF80000u;
    if (a < 0x00800000u && a != 0u) {  // denormal: keep 5 significant bits (rare path)
        const int p = 31 - __clz((int)a);
        const int sh = p > 4 ? p - 4 : 0;
        v = sh > 0 ? ((a + ((1u << (sh - 1)) - 1u) + ((a >> sh) & 1u)) >> sh) << sh : a;
    }
    v = a >= 0x43780000u ? 0x43780000u : v;  // >= 248 -> 248
    v |= s;
    v = (a == 0u || a > 0x7F800000u) ? kBitsQNaN : v;  // 0 * inf in the reference; NaN in -> NaN out
    return v;
}

__device__ __forceinline__ float layerout1(float r) { return __uint_as_float(layerout_bits(__float_as_uint(r))); }
__device__ __forceinline__ float4 layerout4(float4 r) {
    return make_float4(layerout1(r.x), layerout1(r.y), layerout1(r.z), layerout1(r.w));
}

__device__ __forceinline__ float4 post_apply(float4 r, const PostOp po, int c) {
    if (po.scale) {
        const float4 sc = *reinterpret_cast<const float4*>(po.scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(po.shift + c);
        r.x = __builtin_fmaf(r.x, sc.x, sh.x); r.y = __builtin_fmaf(r.y, sc.y, sh.y);
        r.z = __builtin_fmaf(r.z, sc.z, sh.z); r.w = __builtin_fmaf(r.w, sc.w, sh.w);
    }
    if (po.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
    return r;
}

// The same split in two, so that kernels whose lanes keep their channels for many outputs load the
// per-channel vectors ONCE: fetched inside the store loop, the two loads sit on the critical path of
// every store (measured: +12-38 % on the pointwise kernels, +5-8 % on depthwise, bench.py --post).
struct PostVec {
    float4 sc, sh;
};

__device__ __forceinline__ PostVec post_load(const PostOp po, int c) {
    PostVec v;
    v.sc = make_float4(1.f, 1.f, 1.f, 1.f);
    v.sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (po.scale) {
        v.sc = *reinterpret_cast<const float4*>(po.scale + c);
        v.sh = *reinterpret_cast<const float4*>(po.shift + c);
    }
    return v;
}

__device__ __forceinline__ float4 post_apply_v(float4 r, const PostOp po, const PostVec& v) {
    if (po.scale) {
        r.x = __builtin_fmaf(r.x, v.sc.x, v.sh.x); r.y = __builtin_fmaf(r.y, v.sc.y, v.sh.y);
        r.z = __builtin_fmaf(r.z, v.sc.z, v.sh.z); r.w = __builtin_fmaf(r.w, v.sc.w, v.sh.w);
        if (po.layerout) r = layerout4(r);
    }
    if (po.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
    return r;
}

__device__ __forceinline__ float post_apply1(float r, const PostOp po, int c) {
    if (po.scale) {
        r = __builtin_fmaf(r, po.scale[c], po.shift[c]);
        if (po.layerout) r = layerout1(r);
    }
    if (po.relu) r = fmaxf(r, 0.f);
    return r;
}

// 2^ESH * Q_FMT(x / Ka) as float32, where s divides by Ka / 2^ESH.
template <int FMT, int ESH = 0>
__device__ __forceinline__ float quantize_scaled(float x, const ScaleDiv s, const uint32_t* __restrict__ sT) {
    return __uint_as_float(quant_bits<FMT, ESH>(__float_as_uint(div_const(x, s)), __float_as_uint(x), sT));
}

}  // namespace slfp
