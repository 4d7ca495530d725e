// slfp_codes.hpp -- activations as 1-byte SLFP<3,4> / SFP<3,3> codes BETWEEN layers (gfx950).
//
// Every Conv2d_Q of the reference nets is followed by BatchNorm2d -> ReLU -> the next Conv2d_Q, whose first step is
// input_q = quantize_act(input / Ka) (utils/conv2d_func.py:21; nets_imgnet/mobilenetv1.py:24-33).  That quantizer is a
// pure function of the producer's output, so the producer's epilogue can apply it and store the 8-bit code instead of
// the float32 value; the consumer turns the byte back into the float32 (depthwise) or fp16 (matrix-core operand) value
// of the class.  Results are bit-identical to the float32 interface (same classes, same values), activations cross HBM
// as 1 B/element instead of 4 (second half).
//
// Code bytes are the extended code points of slfp_encode_f32(.., fmt | SLFP_FMT_EXT) (include/slfp.h):
//   Qbits 8: sign<<7 | (E+4)<<4 | m;  Qbits 7: sign<<6 | (E+4)<<3 | m;  0x00 = +-1e-10 class ("tiny", sign in the top
//   bit), 0x01 = exact zero, sign|0x02 = the clamp literal 15.3216496 of Qbits 8 (utils/sfp_quant.py:95), so that
//   decode(code) == quantize_act(x / Ka) bit for bit.  NaN has no code: like slfp_encode_f32 it becomes 0x00.
//
// Producer side  enc4_code():  the threshold table of slfp_enc.hpp with V = the two classes' code bytes: multiply, clamp,
//   bin, ds_read_b64, ONE exact compare, and a byte select that writes straight into its lane of the packed dword:
//   5 VALU + 1 LDS per element.
// Consumer side  dec_*():  a 256-entry LDS table indexed by the byte (SDWA shift extracts byte k and scales it in one
//   instruction): 1 VALU + 1 LDS per element (the encode-on-load it replaces: 6-7 VALU + 1 LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "slfp_device.hpp"
#include "slfp_enc.hpp"

namespace slfp {

constexpr int kEncCode = 2;   // EncArgs representation: V = code(lower class) | code(upper class) << 8, unsigned codes

constexpr int kDecF32 = 0;    // decode table entry = float32 bits of the class value
constexpr int kDecF16D = 1;   // entry = fp16(16 * value) in BOTH halves (a pair is assembled with one v_bfi / v_perm)
constexpr int kDecBytes = 1024;

// float32 bits of extended code `code` (Qbits 8: FMT = kFmtAct8, Qbits 7: kFmtSfp7), without an LDS table
template <int FMT>
__device__ __forceinline__ uint32_t decode_ext_bits(uint32_t code) {
    constexpr int MB = (FMT == kFmtSfp7) ? 3 : 4;
    const uint32_t s = ((code >> (MB + 3)) & 1u) << 31;
    const uint32_t mag = code & ((1u << (MB + 3)) - 1u);
    uint32_t v;
    if constexpr (FMT == kFmtSfp7) {
        v = (mag + (123u << 3)) << 20;
    } else {
        const uint32_t idx = mag + (123u << 4);
        v = ((idx >> 4) << 23) | kT16[idx & 15u];
        v = mag == 2u ? kBitsClamp8 : v;
    }
    v = mag == 0u ? kBitsTiny : v;
    v |= s;
    v = code == 1u ? 0u : v;
    return v;
}

// Fills the 256-entry decode table in LDS (1 KiB, 16-byte aligned).  The caller's next __syncthreads() publishes it.
template <int FMT, int REP, int NT>
__device__ __forceinline__ void dec_fill(uint32_t* sDec) {
    for (int i = threadIdx.x; i < 256; i += NT) {
        const uint32_t v = decode_ext_bits<FMT>((uint32_t)i);
        if constexpr (REP == kDecF32) {
            sDec[i] = v;
        } else {
            const _Float16 h = (_Float16)(16.0f * __uint_as_float(v));   // the tiny class flushes to 0 as on the float32 path
            uint16_t hb;
            __builtin_memcpy(&hb, &h, 2);
            sDec[i] = (uint32_t)hb | ((uint32_t)hb << 16);
        }
    }
}

// table entry of byte K of `codes`
template <int K>
__device__ __forceinline__ uint32_t dec_entry(uint32_t codes, const unsigned char* __restrict__ sDec) {
    uint32_t off;
    if constexpr (K == 0) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(codes));
    else if constexpr (K == 1) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(codes));
    else if constexpr (K == 2) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(codes));
    else asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(codes));
    return *reinterpret_cast<const uint32_t*>(sDec + off);
}

// 4 codes (one dword, byte k = channel k) -> 4 float32 values (kDecF32 table)
__device__ __forceinline__ float4 dec4_f32(uint32_t codes, const unsigned char* __restrict__ sDec) {
    return make_float4(__uint_as_float(dec_entry<0>(codes, sDec)), __uint_as_float(dec_entry<1>(codes, sDec)),
                       __uint_as_float(dec_entry<2>(codes, sDec)), __uint_as_float(dec_entry<3>(codes, sDec)));
}

// 4 codes -> 4 fp16(16 * value), packed in two registers (kDecF16D table): element 0 in the low half of .x
__device__ __forceinline__ uint2 dec4_f16(uint32_t codes, const unsigned char* __restrict__ sDec) {
    const uint32_t e0 = dec_entry<0>(codes, sDec), e1 = dec_entry<1>(codes, sDec);
    const uint32_t e2 = dec_entry<2>(codes, sDec), e3 = dec_entry<3>(codes, sDec);
    uint2 p;
    p.x = (e0 & 0xFFFFu) | (e1 & 0xFFFF0000u);   // one v_bfi_b32 each
    p.y = (e2 & 0xFFFFu) | (e3 & 0xFFFF0000u);
    return p;
}

// 4 float32 values -> their 4 extended codes for a consumer with scale Ka (kEncCode table of that Ka), packed.
// SIGNED = false: the values are known to be >= 0 (a ReLU ran just before): no sign handling.
// NaN inputs give 0x00 like slfp_encode_f32 (patched on a cold branch).
template <bool SIGNED>
__device__ __forceinline__ uint32_t enc4_code(const float4 x, const float r1, const float lo, const float hi,
                                              const unsigned char* __restrict__ sTab) {
    const float q0 = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.x) * r1, lo, hi);
    const float q1 = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.y) * r1, lo, hi);
    const float q2 = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.z) * r1, lo, hi);
    const float q3 = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.w) * r1, lo, hi);
    const uint2 e0 = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(q0));
    const uint2 e1 = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(q1));
    const uint2 e2 = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(q2));
    const uint2 e3 = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(q3));
    uint32_t d;
    // byte k of d = |x_k| >= X_k ? V_k[15:8] : V_k[7:0].  One asm statement: a VALU result written with dst_sel != DWORD
    // needs one wait state before the next VALU access of that register (gfx940+ forwarding hazard); the interleaved
    // compares provide it, the trailing s_nop covers whatever hipcc schedules next.
    asm("v_cmp_ge_f32_e64 vcc, |%1|, %2\n\t"
        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
        "v_cmp_ge_f32_e64 vcc, |%4|, %5\n\t"
        "v_cndmask_b32_sdwa %0, %6, %6, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
        "v_cmp_ge_f32_e64 vcc, |%7|, %8\n\t"
        "v_cndmask_b32_sdwa %0, %9, %9, vcc dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
        "v_cmp_ge_f32_e64 vcc, |%10|, %11\n\t"
        "v_cndmask_b32_sdwa %0, %12, %12, vcc dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
        "s_nop 0"
        : "=&v"(d)
        : "v"(x.x), "v"(e0.x), "v"(e0.y), "v"(x.y), "v"(e1.x), "v"(e1.y), "v"(x.z), "v"(e2.x), "v"(e2.y),
          "v"(x.w), "v"(e3.x), "v"(e3.y)
        : "vcc");
    if constexpr (SIGNED) {
        // sign(x) goes to the top bit (bit 7 / bit 6 is the caller's: see enc4_code_fmt) of every class but exact zero
        const float xs[4] = {x.x, x.y, x.z, x.w};
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t c = (d >> (8 * k)) & 0xFFu;
            const uint32_t s = (__float_as_uint(xs[k]) >> 24) & 0x80u;
            c = c == 1u ? c : (c | s);
            out |= c << (8 * k);
        }
        d = out;
    }
    if (__builtin_expect(enc_has_nan4(x), 0)) {
        if (x.x != x.x) d &= 0xFFFFFF00u;
        if (x.y != x.y) d &= 0xFFFF00FFu;
        if (x.z != x.z) d &= 0xFF00FFFFu;
        if (x.w != x.w) d &= 0x00FFFFFFu;
    }
    return d;
}

// The same for a producer whose epilogue ends in a ReLU, with the ReLU folded into the quantizer: the bin estimate and the
// compare use x itself instead of |x|, so a negative x (and -0) clamps into the lowest bin and fails its compare = the
// class of exact zero (0x01) -- exactly the code of max(x, 0).  No v_max, and the two multiplies of a pair are one v_pk_mul.
__device__ __forceinline__ uint32_t enc4_code_relu(const float4 x, const float r1, const float lo, const float hi,
                                                   const unsigned char* __restrict__ sTab) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 p01 = f32x2{x.x, x.y} * r1, p23 = f32x2{x.z, x.w} * r1;   // v_pk_mul_f32: the same IEEE products
    const float q0 = __builtin_amdgcn_fmed3f(p01[0], lo, hi);
    const float q1 = __builtin_amdgcn_fmed3f(p01[1], lo, hi);
    const float q2 = __builtin_amdgcn_fmed3f(p23[0], lo, hi);
    const float q3 = __builtin_amdgcn_fmed3f(p23[1], lo, hi);
    const uint2 e0 = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(q0));
    const uint2 e1 = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(q1));
    const uint2 e2 = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(q2));
    const uint2 e3 = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(q3));
    uint32_t d;
    asm("v_cmp_ge_f32_e64 vcc, %1, %2\n\t"
        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
        "v_cmp_ge_f32_e64 vcc, %4, %5\n\t"
        "v_cndmask_b32_sdwa %0, %6, %6, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
        "v_cmp_ge_f32_e64 vcc, %7, %8\n\t"
        "v_cndmask_b32_sdwa %0, %9, %9, vcc dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
        "v_cmp_ge_f32_e64 vcc, %10, %11\n\t"
        "v_cndmask_b32_sdwa %0, %12, %12, vcc dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
        "s_nop 0"
        : "=&v"(d)
        : "v"(x.x), "v"(e0.x), "v"(e0.y), "v"(x.y), "v"(e1.x), "v"(e1.y), "v"(x.z), "v"(e2.x), "v"(e2.y),
          "v"(x.w), "v"(e3.x), "v"(e3.y)
        : "vcc");
    if (__builtin_expect(enc_has_nan4(x), 0)) {
        if (x.x != x.x) d &= 0xFFFFFF00u;
        if (x.y != x.y) d &= 0xFFFF00FFu;
        if (x.z != x.z) d &= 0xFF00FFFFu;
        if (x.w != x.w) d &= 0x00FFFFFFu;
    }
    return d;
}

// SFP<3,3> codes carry the sign in bit 6: enc4_code<true> puts it in bit 7; move it.
template <int FMT, bool SIGNED>
__device__ __forceinline__ uint32_t enc4_code_fmt(const float4 x, const float r1, const float lo, const float hi,
                                                  const unsigned char* __restrict__ sTab) {
    uint32_t d = enc4_code<SIGNED>(x, r1, lo, hi, sTab);
    if constexpr (SIGNED && FMT == kFmtSfp7) d = (d & 0x3F3F3F3Fu) | ((d & 0x80808080u) >> 1);
    return d;
}

// sign handling of the output codes when no ReLU precedes the quantizer: the top bit of every class but exact zero
__device__ __forceinline__ uint32_t code_sign4(uint32_t d, const float4 x, int fmt_out) {
    const float xs[4] = {x.x, x.y, x.z, x.w};
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t c = (d >> (8 * k)) & 0xFFu;
        const uint32_t s = (__float_as_uint(xs[k]) >> 24) & 0x80u;
        c = (c == 1u || xs[k] != xs[k]) ? c : (c | s);
        out |= c << (8 * k);
    }
    if (fmt_out == kFmtSfp7) out = (out & 0x3F3F3F3Fu) | ((out & 0x80808080u) >> 1);
    return out;
}

// ---- 4 x 4 dword transpose across the four 16-lane rows of a wave --------------------------------------------------
// in:  row r (lanes 16r..16r+15) holds a[i] = M[r][i];  out: row r holds a[i] = M[i][r].
// Two v_permlane32_swap (rows {0,1} <-> {2,3}) and two v_permlane16_swap (odd <-> even rows): 4 VALU instructions for
// 16 bytes per lane.  Used to turn "16 consecutive channel codes per lane" (one coalesced 16-byte load / store) into
// the matrix-core fragment order (lane-quarter kq owns channels 4kq.. of every 16) and back.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rows_transpose4(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3) {
    u32x2_t t;
    t = __builtin_amdgcn_permlane32_swap(a0, a2, false, false); a0 = t[0]; a2 = t[1];
    t = __builtin_amdgcn_permlane32_swap(a1, a3, false, false); a1 = t[0]; a3 = t[1];
    t = __builtin_amdgcn_permlane16_swap(a0, a1, false, false); a0 = t[0]; a1 = t[1];
    t = __builtin_amdgcn_permlane16_swap(a2, a3, false, false); a2 = t[0]; a3 = t[1];
}

// ---- host side (enc_table.hip) --------------------------------------------------------------------------------------
// unsigned extended code of the class whose float32 value has bits `vbits` (sign ignored), or -1 if vbits is not a value
// of format fmt (kFmtAct8 | kFmtSfp7)
int host_ext_code(uint32_t vbits, int fmt);

}  // namespace slfp
