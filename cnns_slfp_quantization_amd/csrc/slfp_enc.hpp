// slfp_enc.hpp -- threshold-table form of the activation quantizers (gfx950).
//
// quantize_act(k)(x / Ka) (utils/sfp_quant.py:59-97 behind utils/conv2d_func.py:21) is, for a
// fixed Ka, a step function of |x| with ~110 steps, odd in x.  slfp_device.hpp evaluates it
// the long way (correctly rounded x/Ka by an FMA chain, RNE of the mantissa, table, four
// range overrides: 22 VALU instructions per element), which made every HBM-bound kernel
// VALU-issue-bound (profiles/r01*).  Here the same function costs 6:
//
//     q0  = |x| * r1                      r1 = RN(1/Ka) * (1 + 2^-7): a 2-ulp-accurate quotient,
//                                         nudged so that no step of Q lies near a bin edge
//     q0c = med3(q0, lo, hi)              lo = 2^-5, hi = 16 - ulp: 9 binades = 144 live bins
//     bin = bits(q0c)[19..26]             one SDWA `and`: the byte offset of an 8-byte entry
//     {X, V} = table[bin]                 ds_read_b64
//     value  = |x| >= X ? Vnext : V       ONE exact float compare in x-space
//     sign   = bfi from x
//
// Each bin (2^19 consecutive q0 patterns) holds at most one step of Q; X is the smallest
// float32 |x| whose exactly rounded quotient lands above it.  The table is built on the host
// by bisection against a plain C++ restatement of the quantizer that uses IEEE float32
// division (enc_table.hip: build()), which also PROVES the one-step-per-bin property for the
// given Ka (build() fails otherwise and the caller keeps the long form), is cached per
// (Ka, format, representation), and reaches the kernels as a by-value kernel argument (2 KiB
// of the kernarg segment; copied to LDS by each workgroup) -- no allocation, no extra launch,
// capturable in a HIP graph.  Bit-equality with the long form over ALL 2^32 inputs is checked
// on the device by slfp_debug_enc_mismatches (tests/test_gpu_parity.py).
//
// NaN: bins cannot carry it; callers test four inputs with two unordered compares and patch
// the (never taken) lanes -- see enc_fix_nan4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slfp {

constexpr int kEncEntries = 257;        // 256 bins + a copy of bin 0 behind bin 255 (Vnext of the last bin)
constexpr uint32_t kEncNever = 0x7FC00000u;  // X of a bin without a step: |x| >= NaN is false

// value representations
constexpr int kEncF32 = 0;   // V = float32 bits of Q(x/Ka); Vnext = the next entry's V
constexpr int kEncF16P = 1;  // V = fp16(16 * Q) of the lower class | fp16(16 * Q) of the upper class << 16
constexpr int kEncF16LO = 3; // V = the fp16 RESIDUALS of the same two classes: fp16(16 Q - fp32(fp16(16 Q))): with kEncF16P the hi / lo
                             // operand pair of the three-pass (float32-equivalent) MFMA mode (round 3)

struct EncArgs {
    float r1, lo, hi;
    uint32_t valid;            // 0: build() could not prove the table for this Ka (callers use the long form)
    uint2 e[kEncEntries];      // {X, V}
};

// Copies the table from the kernarg segment into LDS (`sTab`: kEncEntries * 8 bytes, 8-byte aligned).
// The caller's next __syncthreads() publishes it.
template <int NT>
__device__ __forceinline__ void enc_fill(uint2* sTab, const EncArgs& a) {
    for (int i = threadIdx.x; i < kEncEntries; i += NT) sTab[i] = a.e[i];
}

// Compact form for kernels that need two tables in one 4 KiB kernarg segment: only the 146 entries a clamped
// quotient can index are live (bins 160..255 = binades 2^-5 .. 2^0, bins 0..49 = binades 2^1 .. 2^3 and their successors).
constexpr int kEncCompact = 146;
struct EncArgsCompact {
    float r1, lo, hi;
    uint32_t valid;
    uint2 e[kEncCompact];   // [0, 96): bins 160..255; [96, 146): bins 0..49
};

inline EncArgsCompact enc_compact(const EncArgs& a) {
    EncArgsCompact c;
    c.r1 = a.r1; c.lo = a.lo; c.hi = a.hi; c.valid = a.valid;
    for (int i = 0; i < kEncCompact; ++i) c.e[i] = a.e[i < 96 ? 160 + i : i - 96];
    return c;
}

template <int NT>
__device__ __forceinline__ void enc_fill_compact(uint2* sTab, const EncArgsCompact& a) {
    for (int i = threadIdx.x; i < kEncCompact; i += NT) sTab[i < 96 ? 160 + i : i - 96] = a.e[i];
    if (threadIdx.x == 0) sTab[256] = a.e[96];   // successor of bin 255 = bin 0
}

// byte offset of the bin of q0c inside the LDS table: bits 19..26 of the pattern, times 8
__device__ __forceinline__ uint32_t enc_bin_off(float q0c) {
    uint32_t off;
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(off) : "s"(0x7F8u), "v"(__float_as_uint(q0c)));
    return off;
}

// Q(x / Ka) as float32 (kEncF32 table).  NaN inputs give a finite value: see enc_fix_nan4.
__device__ __forceinline__ float enc_f32(float x, const float r1, const float lo, const float hi,
                                         const unsigned char* __restrict__ sTab) {
    const float q0c = __builtin_amdgcn_fmed3f(__builtin_fabsf(x) * r1, lo, hi);
    const uint32_t off = enc_bin_off(q0c);
    const uint2 e = *reinterpret_cast<const uint2*>(sTab + off);
    const uint32_t vn = *reinterpret_cast<const uint32_t*>(sTab + off + 12);
    const uint32_t v = __builtin_fabsf(x) >= __uint_as_float(e.x) ? vn : e.y;
    return __builtin_copysignf(__uint_as_float(v), x);
}

// NaN in -> NaN out for four values at once: two unordered compares, and a branch that is never taken
// on real activations.
__device__ __forceinline__ void enc_fix_nan4(const float4 x, float4& q) {
    if (__builtin_expect(__builtin_isunordered(x.x, x.y) | __builtin_isunordered(x.z, x.w), 0)) {
        const float nan = __uint_as_float(0x7FC00000u);
        q.x = x.x != x.x ? nan : q.x; q.y = x.y != x.y ? nan : q.y;
        q.z = x.z != x.z ? nan : q.z; q.w = x.w != x.w ? nan : q.w;
    }
}

// four values, NaN inputs NOT patched (callers that batch several float4s test enc_has_nan4 once for all of
// them, so that no branch splits the encode stream)
__device__ __forceinline__ float4 enc4_f32_raw(const float4 x, const float r1, const float lo, const float hi,
                                               const unsigned char* __restrict__ sTab) {
    float4 q;
    q.x = enc_f32(x.x, r1, lo, hi, sTab); q.y = enc_f32(x.y, r1, lo, hi, sTab);
    q.z = enc_f32(x.z, r1, lo, hi, sTab); q.w = enc_f32(x.w, r1, lo, hi, sTab);
    return q;
}

__device__ __forceinline__ bool enc_has_nan4(const float4 x) {
    return __builtin_isunordered(x.x, x.y) | __builtin_isunordered(x.z, x.w);
}

__device__ __forceinline__ float4 enc4_f32(const float4 x, const float r1, const float lo, const float hi,
                                           const unsigned char* __restrict__ sTab) {
    float4 q = enc4_f32_raw(x, r1, lo, hi, sTab);
    enc_fix_nan4(x, q);
    return q;
}

// fp16(16 * Q(x / Ka)) for two consecutive values, packed (kEncF16P table): the MFMA B-operand form.
// The select writes its 16-bit result into the low / high half of the destination (SDWA), so the
// pair costs no packing instruction; the two signs are inserted together.
__device__ __forceinline__ uint32_t enc2_f16(float xa, float xb, const float r1, const float lo, const float hi,
                                             const unsigned char* __restrict__ sTab) {
    const float qa = __builtin_amdgcn_fmed3f(__builtin_fabsf(xa) * r1, lo, hi);
    const float qb = __builtin_amdgcn_fmed3f(__builtin_fabsf(xb) * r1, lo, hi);
    const uint2 ea = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(qa));
    const uint2 eb = *reinterpret_cast<const uint2*>(sTab + enc_bin_off(qb));
    uint32_t d, sg, out;
    // d[15:0] = |xa| >= Xa ? ea.y[31:16] : ea.y[15:0];  d[31:16] likewise from eb (low half preserved);
    // sg = {xb[31:16], xa[31:16]}: the two sign bits at 31 and 15.  One asm statement: a VALU result written
    // with dst_sel != DWORD needs one wait state before a VALU read (gfx940+ forwarding hazard), which the
    // interleaved compares / the permute provide; hipcc pads nothing inside asm.
    asm("v_cmp_ge_f32_e64 vcc, |%3|, %4\n\t"
        "v_cndmask_b32_sdwa %0, %5, %5, vcc dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_cmp_ge_f32_e64 vcc, |%6|, %7\n\t"
        "v_cndmask_b32_sdwa %0, %8, %8, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_perm_b32 %1, %6, %3, %9\n\t"
        "v_bfi_b32 %2, %10, %0, %1"
        : "=&v"(d), "=&v"(sg), "=&v"(out)
        : "v"(xa), "v"(ea.x), "v"(ea.y), "v"(xb), "v"(eb.x), "v"(eb.y), "s"(0x07060302u), "s"(0x7FFF7FFFu)
        : "vcc");
    return out;
}

// The hi / lo fp16 operand pair of two consecutive values for the three-pass MFMA mode: `sTab` is the kEncF16P table, `sLo`
// the kEncF16LO table of the same scale (same bins, same thresholds: ONE estimate and ONE compare per value select both
// halves).  hi carries x's sign (v_bfi), lo -- which has a sign of its own -- is flipped where x is negative (v_xor).
__device__ __forceinline__ void enc2_f16_hl(float xa, float xb, const float r1, const float lo, const float hi,
                                            const unsigned char* __restrict__ sTab, const unsigned char* __restrict__ sLo,
                                            uint32_t& out_hi, uint32_t& out_lo) {
    const float qa = __builtin_amdgcn_fmed3f(__builtin_fabsf(xa) * r1, lo, hi);
    const float qb = __builtin_amdgcn_fmed3f(__builtin_fabsf(xb) * r1, lo, hi);
    const uint32_t oa = enc_bin_off(qa), ob = enc_bin_off(qb);
    const uint2 ea = *reinterpret_cast<const uint2*>(sTab + oa);
    const uint2 eb = *reinterpret_cast<const uint2*>(sTab + ob);
    const uint32_t la = *reinterpret_cast<const uint32_t*>(sLo + oa + 4);
    const uint32_t lb = *reinterpret_cast<const uint32_t*>(sLo + ob + 4);
    uint32_t d, l, sg, oh, ol;
    asm("v_cmp_ge_f32_e64 vcc, |%5|, %6\n\t"
        "v_cndmask_b32_sdwa %0, %7, %7, vcc dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_cndmask_b32_sdwa %1, %8, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_cmp_ge_f32_e64 vcc, |%9|, %10\n\t"
        "v_cndmask_b32_sdwa %0, %11, %11, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_cndmask_b32_sdwa %1, %12, %12, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_perm_b32 %2, %9, %5, %13\n\t"
        "v_bfi_b32 %3, %14, %0, %2\n\t"
        "v_and_b32 %2, %15, %2\n\t"
        "v_xor_b32 %4, %1, %2"
        : "=&v"(d), "=&v"(l), "=&v"(sg), "=&v"(oh), "=&v"(ol)
        : "v"(xa), "v"(ea.x), "v"(ea.y), "v"(la), "v"(xb), "v"(eb.x), "v"(eb.y), "v"(lb), "s"(0x07060302u), "s"(0x7FFF7FFFu),
          "s"(0x80008000u)
        : "vcc");
    out_hi = oh;
    out_lo = ol;
}

// four consecutive values -> two packed registers; NaN inputs NOT patched (see enc_has_nan4 / enc_patch_nan4_f16)
__device__ __forceinline__ uint2 enc4_f16_raw(const float4 x, const float r1, const float lo, const float hi,
                                              const unsigned char* __restrict__ sTab) {
    uint2 p;
    p.x = enc2_f16(x.x, x.y, r1, lo, hi, sTab);
    p.y = enc2_f16(x.z, x.w, r1, lo, hi, sTab);
    return p;
}

// NaN in -> fp16 NaN (0x7E00) out
__device__ __forceinline__ void enc_patch_nan4_f16(const float4 x, uint2& p) {
    if (x.x != x.x) p.x = (p.x & 0xFFFF0000u) | 0x7E00u;
    if (x.y != x.y) p.x = (p.x & 0x0000FFFFu) | 0x7E000000u;
    if (x.z != x.z) p.y = (p.y & 0xFFFF0000u) | 0x7E00u;
    if (x.w != x.w) p.y = (p.y & 0x0000FFFFu) | 0x7E000000u;
}

__device__ __forceinline__ uint2 enc4_f16(const float4 x, const float r1, const float lo, const float hi,
                                          const unsigned char* __restrict__ sTab) {
    uint2 p = enc4_f16_raw(x, r1, lo, hi, sTab);
    if (__builtin_expect(enc_has_nan4(x), 0)) enc_patch_nan4_f16(x, p);
    return p;
}

// ---- host side (enc_table.hip) ----------------------------------------------------------------
// The cached table for (ka, fmt in {kFmtAct8, kFmtSfp7}, rep in {kEncF32, kEncF16P}); never null.
// ->valid == 0 when the one-step-per-bin property could not be proven for this Ka.
const EncArgs* enc_table(float ka, int fmt, int rep);
// float32 bits of Q_fmt(x / d) computed with IEEE float32 division on the host (x >= 0 or any sign).
uint32_t host_quant_bits(float x, float d, int fmt);

}  // namespace slfp
