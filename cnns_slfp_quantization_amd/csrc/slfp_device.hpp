// slfp_device.hpp -- device-side SLFP<3,4> / SFP<3,3> codec for gfx950 (CDNA4).
//
// The reference computes its fake-quantizers with ~25 float32 ATen passes per tensor
// (utils/sfp_quant.py:32-47 weights, :80-96 activations, :14-30/:63-78 SFP<3,3>).  The
// result is a pure function of the float32 bit pattern, so here it is one short integer
// routine that every kernel inlines on its load path:
//   - exponent extract     = the biased-exponent field of the scaled input,
//   - mantissa lookup      = RNE of the mantissa to 4 (3) bits with the carry walking into
//                            the exponent, the 17-entry lin->log map folded into two
//                            compares, and a 16-entry 2^(m/16) table held in LDS,
//   - the ordered overrides (tiny -> +-1e-10, [1/16,1/8) -> 1/8, clamp) as selects.
// Bit-exactness against the reference is pinned by tests/golden (exhaustive sweep).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slfp {

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

// 2^(k/16) rounded to float32 (identical to what the reference's pow(2, e + k/16) yields).
static __device__ const uint32_t kT16[16] = {
    0x3F800000u, 0x3F85AAC3u, 0x3F8B95C2u, 0x3F91C3D3u, 0x3F9837F0u, 0x3F9EF532u,
    0x3FA5FED7u, 0x3FAD583Fu, 0x3FB504F3u, 0x3FBD08A4u, 0x3FC5672Au, 0x3FCE248Cu,
    0x3FD744FDu, 0x3FE0CCDFu, 0x3FEAC0C7u, 0x3FF5257Du};

// Copy the 2^(m/16) table into LDS (16 dwords, 16 distinct banks: any per-lane index is
// conflict-free).  Caller must __syncthreads() before the first lookup.
__device__ __forceinline__ void lut_fill(uint32_t* sT) {
    if (threadIdx.x < 16) sT[threadIdx.x] = kT16[threadIdx.x];
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

// (biased_exponent << MBITS) | mantissa_code of the NORMAL path (valid for 1/8 <= |q| <= clamp;
// other inputs are overridden by the callers' selects).
template <int FMT>
__device__ __forceinline__ uint32_t normal_index(uint32_t a) {
    if constexpr (FMT == kFmtSfp7) {
        return (a + 0x7FFFFu + ((a >> 20) & 1u)) >> 20;  // RNE to 3 bits, carry -> exponent
    } else if constexpr (FMT == kFmtAct8) {
        const uint32_t lin = (a + 0x3FFFFu + ((a >> 19) & 1u)) >> 19;  // RNE to 4 bits (sfp_quant.py:88)
        const uint32_t l = lin & 15u;
        // log converter (sfp_quant.py:89): L = [0,1,3,4,...,14,15,15]
        return lin + (l >= 2u ? 1u : 0u) - (l >= 15u ? 1u : 0u);
    } else {
        return ((a >> 23) << 4) + w8_log_mantissa(a & 0x7FFFFFu);  // m == 16 carries
    }
}

// float32 bits of Q_FMT(q) -- bit-identical to the reference's qfn.forward output.
template <int FMT>
__device__ __forceinline__ uint32_t quant_bits(uint32_t u, const uint32_t* __restrict__ sT) {
    const uint32_t a = u & 0x7FFFFFFFu;
    const uint32_t s = u & 0x80000000u;
    const uint32_t idx = normal_index<FMT>(a);
    uint32_t v;
    if constexpr (FMT == kFmtSfp7) {
        v = idx << 20;  // (1 + m/8) * 2^E exactly
        v = a >= kBitsClamp7 ? kBitsClamp7 : v;
    } else {
        v = sT[idx & 15u] + (((idx >> 4) - 127u) << 23);
        v = a > kBitsClamp8 ? kBitsClamp8 : v;
    }
    v = a < kBitsEighth ? kBitsEighth : v;
    v = a < kBitsMin ? kBitsTiny : v;
    v |= s;
    v = a == 0u ? 0u : v;               // torch.sign(+-0) == 0
    v = a > 0x7F800000u ? kBitsQNaN : v;  // NaN in -> NaN out
    return v;
}

// canonical (or extended) code byte of Q_FMT(q).
template <int FMT>
__device__ __forceinline__ uint32_t quant_code(uint32_t u, bool ext) {
    const uint32_t a = u & 0x7FFFFFFFu;
    constexpr int MB = (FMT == kFmtSfp7) ? 3 : 4;
    const uint32_t sc = (u >> 31) << (MB + 3);
    uint32_t c = normal_index<FMT>(a) - (123u << MB);
    if constexpr (FMT == kFmtSfp7) {
        c = a >= kBitsClamp7 ? 0x3Fu : c;
    } else {
        c = a > kBitsClamp8 ? (ext ? 0x02u : 0x7Fu) : c;
    }
    c = a < kBitsEighth ? (1u << MB) : c;
    c = a < kBitsMin ? 0u : c;
    c |= sc;
    c = a == 0u ? (ext ? 1u : 0u) : c;
    c = a > 0x7F800000u ? 0u : c;
    return c & 0xFFu;
}

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
        v = sT[idx & 15u] + (((idx >> 4) - 127u) << 23);
        v = (ext && mag == 2u) ? kBitsClamp8 : v;
    }
    v = mag == 0u ? kBitsTiny : v;
    v |= s;
    v = (ext && code == 1u) ? 0u : v;
    return v;
}

// Q_FMT(x / scale): the scaled fake-quant of one float.  The division is IEEE float32
// (hipcc lowers `/` to v_div_scale/v_div_fmas/v_div_fixup unless fast-math is on, which
// this library never enables): `input/self.Ka` of utils/conv2d_func.py:21.
template <int FMT>
__device__ __forceinline__ float quantize_scaled(float x, float scale_div, const uint32_t* __restrict__ sT) {
    return __uint_as_float(quant_bits<FMT>(__float_as_uint(x / scale_div), sT));
}

}  // namespace slfp
