// enc_table.hip -- host-side construction (and proof) of the threshold tables of slfp_enc.hpp,
// their cache, and the exhaustive device self-check slfp_debug_enc_mismatches.
//
// The quantizer restated here for the host (host_quant_bits) follows utils/sfp_quant.py:59-97
// (quantize_act, k = 8 and 7) exactly as slfp_device.hpp: quant_bits does, with the reference's
// `input / self.Ka` (utils/conv2d_func.py:21) as an IEEE float32 division.  It is used ONLY to
// place table thresholds; the kernels' results are then compared with the long-form device
// quantizer over all 2^32 inputs (k_enc_check below), which is itself pinned to the reference by
// tests/golden.
#include <map>
#include <mutex>
#include <memory>
#include <cmath>
#include <cstring>
#include "slfp_device.hpp"
#include "slfp_enc.hpp"
#include "slfp_codes.hpp"
#include "slfp_host.hpp"

namespace slfp {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static const uint32_t kT16h[16] = {0x000000u, 0x05AAC3u, 0x0B95C2u, 0x11C3D3u, 0x1837F0u, 0x1EF532u, 0x25FED7u, 0x2D583Fu,
                                   0x3504F3u, 0x3D08A4u, 0x45672Au, 0x4E248Cu, 0x5744FDu, 0x60CCDFu, 0x6AC0C7u, 0x75257Du};

uint32_t host_quant_bits(float x, float d, int fmt) {
    if (x != x) return kBitsQNaN;
    volatile float qv = x / d;  // IEEE float32 division (volatile: no excess precision, no reciprocal rewrite)
    const uint32_t u = f2u(qv), a = u & 0x7FFFFFFFu;
    if (a == 0u) return 0u;  // torch.sign(+-0) == 0
    uint32_t v;
    if (fmt == kFmtSfp7) {
        v = (a + 0x7FFFFu + ((a >> 20) & 1u)) & 0x7FF00000u;      // RNE to 3 mantissa bits   sfp_quant.py:69-72
        if (a >= kBitsClamp7) v = kBitsClamp7;                    // >= 15 -> 15               sfp_quant.py:77
    } else {
        const uint32_t t = a + 0x3FFFFu + ((a >> 19) & 1u);       // RNE to 4 bits             sfp_quant.py:88
        const uint32_t lin = (t >> 19) & 15u;
        const uint32_t l = lin + (lin >= 2u ? 1u : 0u) - (lin >= 15u ? 1u : 0u);  // round(16*log2(.))  :89
        v = (t & 0x7F800000u) | kT16h[l];
        if (a > kBitsClamp8) v = kBitsClamp8;                     // > 15.32165 -> 15.32165    sfp_quant.py:95
    }
    if (v < kBitsEighth) v = kBitsEighth;                         // [1/16, 1/8) -> 1/8        sfp_quant.py:93
    if (a < kBitsMin) v = kBitsTiny;                              // < 1/16 -> 1e-10           sfp_quant.py:92
    return v | (f2u(x) & 0x80000000u);
}

// unsigned extended code (include/slfp.h: SLFP_FMT_EXT) of the class with float32 value bits `vbits` -- the byte
// slfp_encode_f32(x, Ka, fmt | SLFP_FMT_EXT) yields for every x of that class, minus the sign bit (csrc/slfp_device.hpp:
// quant_code).  -1: not a value of the format.
int host_ext_code(uint32_t vbits, int fmt) {
    const uint32_t a = vbits & 0x7FFFFFFFu;
    if (a == 0u) return 1;                    // exact zero
    if (a == kBitsTiny) return 0;             // the +-1e-10 class
    const int E = (int)(a >> 23) - 127;
    if (E < -3 || E > 3) return -1;
    if (fmt == kFmtSfp7) {
        if (a & 0x000FFFFFu) return -1;
        return ((E + 4) << 3) | (int)((a >> 20) & 7u);   // 15.0 = 1.875 * 2^3 is the canonical top code 0x3F
    }
    if (a == kBitsClamp8) return 2;           // the clamp literal 15.3216496 (the computed top value 15.3216524 is 0x7F)
    for (int m = 0; m < 16; ++m)
        if ((a & 0x7FFFFFu) == kT16h[m]) return ((E + 4) << 4) | m;
    return -1;
}

// float32 -> fp16 bits, round to nearest even (the conversion `(_Float16)v` performs on the device)
static uint32_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}

namespace {

struct Builder {
    float d, r1, lo, hi;
    int fmt, rep;

    uint32_t pos(uint32_t xb) const {   // bin position of |x| = u2f(xb): what the kernel computes
        volatile float q0 = u2f(xb) * r1;
        float q = q0;
        q = q < lo ? lo : (q > hi ? hi : q);
        return f2u(q) >> 19;
    }
    uint32_t val(uint32_t xb) const {   // representation of Q(|x|/Ka)
        const uint32_t v = host_quant_bits(u2f(xb), d, fmt);
        if (rep == kEncF32) return v;
        if (rep == kEncCode) return (uint32_t)host_ext_code(v, fmt);   // 0xFFFFFFFF if v is not a class value: build() then fails
        if (rep == kEncF16LO) {   // the residual plane of the hi / lo split, as the long form computes it (conv_pw.hip: encode4)
            const float v16 = 16.0f * u2f(v);
            const _Float16 h = (_Float16)v16;
            volatile float r = v16 - (float)h;
            return f32_to_f16_bits(r);
        }
        return f32_to_f16_bits(16.0f * u2f(v));
    }
    // smallest pattern in [a, b] with pred true (pred is false..true monotone on [a, b]; pred(b) must hold)
    template <class P>
    static uint32_t first_true(uint32_t a, uint32_t b, P pred) {
        while (a < b) {
            const uint32_t m = a + (b - a) / 2;
            if (pred(m)) b = m; else a = m + 1;
        }
        return a;
    }
};

}  // namespace

static bool build(float ka, int fmt, int rep, EncArgs* out) {
    std::memset(out, 0, sizeof(*out));
    Builder B;
    B.d = ka; B.fmt = fmt; B.rep = rep;
    const float r = (float)(1.0 / (double)ka);
    volatile float r1 = r * (1.0f + 0.0078125f);
    B.r1 = r1;
    B.lo = 0.03125f;              // 2^-5
    B.hi = u2f(0x417FFFFFu);      // 16 - ulp
    out->r1 = B.r1; out->lo = B.lo; out->hi = B.hi;
    for (int i = 0; i < kEncEntries; ++i) out->e[i] = make_uint2(kEncNever, 0u);
    const uint32_t P0 = f2u(B.lo) >> 19, P1 = f2u(B.hi) >> 19;   // 122*16 .. 130*16+15
    const uint32_t XMAX = 0x7F800000u;                            // +inf belongs to the last bin
    if (B.pos(0u) != P0 || B.pos(XMAX) != P1) return false;
    uint32_t xa = 0u;
    uint32_t prev_hi_val = 0u;
    bool have_prev = false;
    for (uint32_t p = P0; p <= P1; ++p) {
        // [xa, xb]: the |x| patterns of bin p
        uint32_t xb;
        if (p == P1) xb = XMAX;
        else xb = Builder::first_true(xa, XMAX, [&](uint32_t m) { return B.pos(m) > p; }) - 1u;
        if (xb < xa || xb == 0xFFFFFFFFu) return false;  // empty bin: cannot happen for a sane Ka
        const uint32_t va = B.val(xa), vb = B.val(xb);
        if (have_prev && prev_hi_val != va) return false;  // a step exactly on a bin edge next to a stepped bin
        uint32_t X = kEncNever;
        if (va != vb) {
            X = Builder::first_true(xa, xb, [&](uint32_t m) { return B.val(m) == vb; });
            if (X == xa || B.val(X - 1u) != va) return false;   // more than one step inside the bin
        }
        // sampled confirmation that the bin really is {va below X, vb from X on}
        const uint32_t span = xb - xa;
        for (int s = 1; s < 32; ++s) {
            const uint32_t m = xa + (uint32_t)(((uint64_t)span * s) / 32);
            const uint32_t expect = (X != kEncNever && m >= X) ? vb : va;
            if (B.val(m) != expect) return false;
        }
        const uint32_t c = p & 0xFFu;
        if (rep == kEncCode && (va > 0x7Fu || vb > 0x7Fu)) return false;
        if (rep == kEncF16LO || rep == kEncF16P) {
            // the hi and the lo table share ONE compare per value, so both take their thresholds from the CLASS steps, also
            // where the two classes happen to have equal halves (the computed top value 15.3216524 and the clamp literal
            // 15.3216496 share their fp16 hi half but not their residual; selecting between equal values is harmless)
            const uint32_t ca = host_quant_bits(u2f(xa), B.d, B.fmt), cb = host_quant_bits(u2f(xb), B.d, B.fmt);
            X = kEncNever;
            if (ca != cb) {
                X = Builder::first_true(xa, xb, [&](uint32_t m) { return host_quant_bits(u2f(m), B.d, B.fmt) == cb; });
                if (X == xa || host_quant_bits(u2f(X - 1u), B.d, B.fmt) != ca) return false;
            }
        }
        if (rep == kEncF32) out->e[c] = make_uint2(X, va);
        else if (rep == kEncF16LO) out->e[c] = make_uint2(X, (va & 0xFFFFu) | (vb << 16));
        else if (rep == kEncCode) out->e[c] = make_uint2(X, va | (vb << 8));
        else out->e[c] = make_uint2(X, va | (vb << 16));
        // with the linear float32 layout the upper class of bin p is read from entry p + 1
        have_prev = (rep == kEncF32) && (X != kEncNever);
        prev_hi_val = vb;
        xa = xb + 1u;
    }
    if (have_prev) return false;                  // the last bin must be flat (everything there is the clamp class)
    out->e[256] = out->e[0];                       // Vnext of bin 255 (binade 127 -> 128)
    out->e[(P1 + 1u) & 0xFFu].y = out->e[P1 & 0xFFu].y;  // never selected; keep it tidy
    out->valid = 1u;
    return true;
}

const EncArgs* enc_table(float ka, int fmt, int rep) {
    static std::mutex mu;
    static std::map<uint64_t, std::unique_ptr<EncArgs>> cache;
    const uint64_t key = ((uint64_t)f2u(ka) << 8) | ((uint64_t)(fmt & 3) << 4) | (uint64_t)(rep & 15);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second.get();
    std::unique_ptr<EncArgs> t(new EncArgs);
    if (!(fmt == kFmtAct8 || fmt == kFmtSfp7) || !scale_div_ok(ka) || !build(ka, fmt, rep, t.get())) t->valid = 0u;
    const EncArgs* p = t.get();
    cache.emplace(key, std::move(t));
    return p;
}

// ---- exhaustive device self-check: table form vs long form over all 2^32 inputs -----------------
// REP == kEncF32:  *out = #x whose float32 result differs from quantize_scaled<FMT> (the long form);
// REP == kEncF16P: *out = #x whose fp16 result differs from fp16(16 * long form), tested in both halves of a pair.
// +0 and -0 count as equal: the long form returns +0 for x = -0 (torch.sign(-0) == 0), the table form -0; every
// consumer multiplies the value into an accumulator that starts at +0, where the two are indistinguishable.
template <int FMT, int REP>
__global__ __launch_bounds__(256) void k_enc_check(const ScaleDiv sd, const EncArgs t, unsigned long long* __restrict__ out) {
    __shared__ uint32_t sT[16];
    __shared__ __attribute__((aligned(16))) uint2 sE[kEncEntries + 1];
    lut_fill<FMT>(sT);
    enc_fill<256>(sE, t);
    __syncthreads();
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(sE);
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < (1ull << 32); i += stride) {
        float4 x;
        x.x = __uint_as_float((uint32_t)i); x.y = __uint_as_float((uint32_t)i + 1u);
        x.z = __uint_as_float((uint32_t)i + 2u); x.w = __uint_as_float((uint32_t)i + 3u);
        const float xs[4] = {x.x, x.y, x.z, x.w};
        uint32_t ref[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ref[e] = __float_as_uint(quantize_scaled<FMT>(xs[e], sd, sT));
        if constexpr (REP == kEncF32) {
            const float4 q = enc4_f32(x, t.r1, t.lo, t.hi, tb);
            const float qs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t g = __float_as_uint(qs[e]);
                const bool both_nan = (g & 0x7FFFFFFFu) > 0x7F800000u && (ref[e] & 0x7FFFFFFFu) > 0x7F800000u;
                const bool both_zero = ((g | ref[e]) & 0x7FFFFFFFu) == 0u;
                if (g != ref[e] && !both_nan && !both_zero) ++bad;
            }
        } else {
            const uint2 p = enc4_f16(x, t.r1, t.lo, t.hi, tb);
            const uint32_t hs[4] = {p.x & 0xFFFFu, p.x >> 16, p.y & 0xFFFFu, p.y >> 16};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 h = (_Float16)(16.0f * __uint_as_float(ref[e]));
                uint16_t hb;
                __builtin_memcpy(&hb, &h, 2);
                const bool both_nan = (hs[e] & 0x7FFFu) > 0x7C00u && (hb & 0x7FFFu) > 0x7C00u;
                const bool both_zero = ((hs[e] | hb) & 0x7FFFu) == 0u;
                if (hs[e] != hb && !both_nan && !both_zero) ++bad;
            }
        }
    }
    if (bad) atomicAdd(out, bad);
}

// ---- exhaustive self-check of the code representation (slfp_codes.hpp): all 2^32 inputs -----------------------------
// out[0] = #x whose byte from enc4_code_fmt<FMT, true> differs from quant_code<FMT>(x / Ka, ext) -- the byte
//          slfp_encode_f32(.., fmt | SLFP_FMT_EXT) stores;  out[1] = the same for the unsigned variant (what a producer
//          with a ReLU epilogue runs) over the inputs it can see: x >= +0 and x == -0;
// out[2] = #codes c (0..255, counted once) whose decode-table entries differ from decode_bits / fp16(16 * decode_bits).
template <int FMT>
__global__ __launch_bounds__(256) void k_code_check(const ScaleDiv sd, const EncArgs t, unsigned long long* __restrict__ out) {
    __shared__ uint32_t sT[16];
    __shared__ __attribute__((aligned(16))) uint2 sE[kEncEntries + 1];
    __shared__ __attribute__((aligned(16))) uint32_t sD32[256], sD16[256];
    lut_fill<kFmtW8>(sT);   // identity table for decode_bits
    enc_fill<256>(sE, t);
    dec_fill<FMT, kDecF32, 256>(sD32);
    dec_fill<FMT, kDecF16D, 256>(sD16);
    __syncthreads();
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(sE);
    unsigned long long bad_s = 0, bad_u = 0, bad_d = 0;
    if (blockIdx.x == 0) {   // decode tables against the long-form decoder (codec.hip uses decode_bits)
        const uint32_t c = threadIdx.x;
        const uint32_t ref = decode_bits<FMT>(c, true, sT);
        const _Float16 h = (_Float16)(16.0f * __uint_as_float(ref));
        uint16_t hb;
        __builtin_memcpy(&hb, &h, 2);
        const uint32_t packed = c | (c << 8) | (c << 16) | (c << 24);
        const float4 f = dec4_f32(packed, reinterpret_cast<const unsigned char*>(sD32));
        const uint2 g = dec4_f16(packed, reinterpret_cast<const unsigned char*>(sD16));
        const bool ok = __float_as_uint(f.x) == ref && __float_as_uint(f.y) == ref && __float_as_uint(f.z) == ref &&
                        __float_as_uint(f.w) == ref && g.x == ((uint32_t)hb | ((uint32_t)hb << 16)) && g.y == g.x;
        if (!ok) ++bad_d;
    }
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < (1ull << 32); i += stride) {
        float4 x;
        x.x = __uint_as_float((uint32_t)i); x.y = __uint_as_float((uint32_t)i + 1u);
        x.z = __uint_as_float((uint32_t)i + 2u); x.w = __uint_as_float((uint32_t)i + 3u);
        const float xs[4] = {x.x, x.y, x.z, x.w};
        const uint32_t cs = enc4_code_fmt<FMT, true>(x, t.r1, t.lo, t.hi, tb);
        const uint32_t cu = enc4_code_fmt<FMT, false>(x, t.r1, t.lo, t.hi, tb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t ux = __float_as_uint(xs[e]);
            const uint32_t ref = quant_code<FMT>(__float_as_uint(div_const(xs[e], sd)), ux, true);
            if (((cs >> (8 * e)) & 0xFFu) != ref) ++bad_s;
            if ((ux < 0x80000000u || ux == 0x80000000u) && ((cu >> (8 * e)) & 0xFFu) != ref) ++bad_u;
        }
    }
    if (bad_s) atomicAdd(out, bad_s);
    if (bad_u) atomicAdd(out + 1, bad_u);
    if (bad_d) atomicAdd(out + 2, bad_d);
}

// ---- exhaustive self-check of the hi / lo pair (three-pass MFMA mode): all 2^32 inputs -------------------------------
// *out = #x whose (hi, lo) fp16 pair from enc2_f16_hl differs from the long form of conv_pw.hip's encode4:
// v = 16 * Q(x / Ka); hi = fp16(v); lo = fp16(v - fp32(hi)).  +0 / -0 count as equal, NaN inputs are the callers' cold branch.
template <int FMT>
__global__ __launch_bounds__(256) void k_enc_hl_check(const ScaleDiv sd16, const EncArgs th, const EncArgsCompact tl,
                                                      unsigned long long* __restrict__ out) {
    __shared__ uint32_t sT[16];
    __shared__ __attribute__((aligned(16))) uint2 sH[kEncEntries + 1], sL[kEncEntries + 1];
    lut_fill<FMT>(sT);
    enc_fill<256>(sH, th);
    enc_fill_compact<256>(sL, tl);
    __syncthreads();
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(sH);
    const unsigned char* tlo = reinterpret_cast<const unsigned char*>(sL);
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 2;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < (1ull << 32); i += stride) {
        const float xa = __uint_as_float((uint32_t)i), xb = __uint_as_float((uint32_t)i + 1u);
        uint32_t hp, lp;
        enc2_f16_hl(xa, xb, th.r1, th.lo, th.hi, tb, tlo, hp, lp);
        const float xs[2] = {xa, xb};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (xs[e] != xs[e]) continue;
            const float v = quantize_scaled<FMT, 4>(xs[e], sd16, sT);
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            uint16_t hb, lb;
            __builtin_memcpy(&hb, &h, 2);
            __builtin_memcpy(&lb, &l, 2);
            const uint32_t gh = (hp >> (16 * e)) & 0xFFFFu, gl = (lp >> (16 * e)) & 0xFFFFu;
            const bool h_ok = gh == hb || ((gh | hb) & 0x7FFFu) == 0u;
            const bool l_ok = gl == lb || ((gl | lb) & 0x7FFFu) == 0u;
            if (!h_ok || !l_ok) ++bad;
        }
    }
    if (bad) atomicAdd(out, bad);
}

}  // namespace slfp

using namespace slfp;

extern "C" int slfp_debug_enc_mismatches(float scale_div, int fmt, unsigned long long* out2, void* stream) {
    if (!out2 || !(scale_div > 0.f)) return fail(SLFP_ERR_BAD_ARG, "slfp_debug_enc_mismatches: bad argument");
    if (fmt != SLFP_FMT_ACT8 && fmt != SLFP_FMT_SFP7) return fail(SLFP_ERR_BAD_ARG, "slfp_debug_enc_mismatches: fmt must be ACT8 or SFP7");
    if (!scale_div_ok(scale_div)) return fail(SLFP_ERR_UNSUPPORTED, "scale must be within [1e-30, 1e30]");
    const EncArgs* t32 = enc_table(scale_div, fmt, kEncF32);
    const EncArgs* t16 = enc_table(scale_div, fmt, kEncF16P);
    if (!t32->valid || !t16->valid) return fail(SLFP_ERR_UNSUPPORTED, "no threshold table for scale %g (one-step-per-bin property not provable)", (double)scale_div);
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(out2, 0, 2 * sizeof(unsigned long long), st) != hipSuccess) return check_launch("hipMemsetAsync");
    const ScaleDiv sd = make_scale_div(scale_div);
    if (fmt == SLFP_FMT_ACT8) {
        hipLaunchKernelGGL((k_enc_check<kFmtAct8, kEncF32>), dim3(256 * 16), dim3(256), 0, st, sd, *t32, out2);
        hipLaunchKernelGGL((k_enc_check<kFmtAct8, kEncF16P>), dim3(256 * 16), dim3(256), 0, st, sd, *t16, out2 + 1);
    } else {
        hipLaunchKernelGGL((k_enc_check<kFmtSfp7, kEncF32>), dim3(256 * 16), dim3(256), 0, st, sd, *t32, out2);
        hipLaunchKernelGGL((k_enc_check<kFmtSfp7, kEncF16P>), dim3(256 * 16), dim3(256), 0, st, sd, *t16, out2 + 1);
    }
    return check_launch("slfp threshold-table self-check kernel");
}

// 1 if a proven threshold table exists for this scale / format (host only; used by tests and logs)
extern "C" int slfp_enc_table_ok(float scale_div, int fmt) {
    if (fmt != SLFP_FMT_ACT8 && fmt != SLFP_FMT_SFP7) return 0;
    if (!(scale_div > 0.f) || !scale_div_ok(scale_div)) return 0;
    return (enc_table(scale_div, fmt, kEncF32)->valid && enc_table(scale_div, fmt, kEncF16P)->valid) ? 1 : 0;
}

// The producer side of the 1-byte inter-layer format (csrc/slfp_codes.hpp): sweeps ALL 2^32 float32 inputs and compares
// the table-driven code with the long form behind slfp_encode_f32(.., fmt | SLFP_FMT_EXT); also checks both decode tables.
extern "C" int slfp_debug_code_mismatches(float scale_div, int fmt, unsigned long long* out3, void* stream) {
    if (!out3 || !(scale_div > 0.f)) return fail(SLFP_ERR_BAD_ARG, "slfp_debug_code_mismatches: bad argument");
    if (fmt != SLFP_FMT_ACT8 && fmt != SLFP_FMT_SFP7) return fail(SLFP_ERR_BAD_ARG, "slfp_debug_code_mismatches: fmt must be ACT8 or SFP7");
    if (!scale_div_ok(scale_div)) return fail(SLFP_ERR_UNSUPPORTED, "scale must be within [1e-30, 1e30]");
    const EncArgs* t = enc_table(scale_div, fmt, kEncCode);
    if (!t->valid) return fail(SLFP_ERR_UNSUPPORTED, "no code table for scale %g (one-step-per-bin property not provable)", (double)scale_div);
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(out3, 0, 3 * sizeof(unsigned long long), st) != hipSuccess) return check_launch("hipMemsetAsync");
    const ScaleDiv sd = make_scale_div(scale_div);
    if (fmt == SLFP_FMT_ACT8) hipLaunchKernelGGL((k_code_check<kFmtAct8>), dim3(256 * 16), dim3(256), 0, st, sd, *t, out3);
    else hipLaunchKernelGGL((k_code_check<kFmtSfp7>), dim3(256 * 16), dim3(256), 0, st, sd, *t, out3);
    return check_launch("slfp code-table self-check kernel");
}

// The three-pass (float32-equivalent) pointwise mode's table encoder (csrc/slfp_enc.hpp: enc2_f16_hl): ALL 2^32 float32
// inputs against the long form.  *out1 must be 0.
extern "C" int slfp_debug_enc_hl_mismatches(float scale_div, int fmt, unsigned long long* out1, void* stream) {
    if (!out1 || !(scale_div > 0.f)) return fail(SLFP_ERR_BAD_ARG, "slfp_debug_enc_hl_mismatches: bad argument");
    if (fmt != SLFP_FMT_ACT8 && fmt != SLFP_FMT_SFP7) return fail(SLFP_ERR_BAD_ARG, "slfp_debug_enc_hl_mismatches: fmt must be ACT8 or SFP7");
    if (!scale_div_ok(scale_div)) return fail(SLFP_ERR_UNSUPPORTED, "scale must be within [1e-30, 1e30]");
    const EncArgs* th = enc_table(scale_div, fmt, kEncF16P);
    const EncArgs* tl = enc_table(scale_div, fmt, kEncF16LO);
    if (!th->valid || !tl->valid) return fail(SLFP_ERR_UNSUPPORTED, "no hi / lo threshold tables for scale %g", (double)scale_div);
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(out1, 0, sizeof(unsigned long long), st) != hipSuccess) return check_launch("hipMemsetAsync");
    const ScaleDiv sd16 = make_scale_div(scale_div, 4);
    const EncArgsCompact tc = enc_compact(*tl);
    if (fmt == SLFP_FMT_ACT8) hipLaunchKernelGGL((k_enc_hl_check<kFmtAct8>), dim3(256 * 16), dim3(256), 0, st, sd16, *th, tc, out1);
    else hipLaunchKernelGGL((k_enc_hl_check<kFmtSfp7>), dim3(256 * 16), dim3(256), 0, st, sd16, *th, tc, out1);
    return check_launch("slfp hi/lo table self-check kernel");
}
