// pool_codes.hip -- MaxPool2d on 1-byte SLFP<3,4> / SFP<3,3> activation codes (gfx950).
//
// The reference nets pool AFTER conv -> BN -> ReLU and BEFORE the next Conv2d_Q, whose first step is
// input_q = quantize_act(input / Ka) (nets_cifar/vgg16.py:30-92 + utils/conv2d_func.py:21).  quantize_act maps intervals of
// x to classes in the order of x, so the CLASS of max(x_i) is the highest class among the x_i: pooling the producer's codes
// (which are the consumer's classes, slfp_codes.hpp) gives exactly the code of the pooled float32 tensor -- provided "highest"
// means the order of the classes' pre-images, not of their float32 values: the clamp literal of Qbits 8 (code 0x02,
// 15.3216496) is the class of the LARGEST inputs but 3 ulp below the top regular value, "tiny" (0x00) sits above
// exact zero (0x01), and codes with the sign bit are ordered by decreasing magnitude.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "slfp_device.hpp"
#include "slfp_host.hpp"

namespace slfp {

// signed rank of an extended code in pre-image order; MB = mantissa bits (4: Qbits 8, 3: Qbits 7)
template <int MB>
__device__ __forceinline__ int code_key(uint32_t c) {
    const uint32_t sbit = 1u << (MB + 3);
    const int mag = (int)(c & (sbit - 1u));
    int r = mag;
    r = mag == 1 ? 0 : r;            // exact zero
    r = mag == 0 ? 1 : r;            // the 1e-10 class
    r = mag == 2 ? (int)sbit : r;    // the clamp literal: above every regular class
    return (c & sbit) ? -r : r;
}
template <int MB>
__device__ __forceinline__ uint32_t key_code(int k) {
    const uint32_t sbit = 1u << (MB + 3);
    const int r = k < 0 ? -k : k;
    uint32_t mag = (uint32_t)r;
    mag = r == 0 ? 1u : mag;
    mag = r == 1 ? 0u : mag;
    mag = r == (int)sbit ? 2u : mag;
    return (k < 0 && mag != 1u) ? (mag | sbit) : mag;
}

// one thread = VEC consecutive channels of one output pixel (NHWC); out-of-image taps are skipped (MaxPool2d pads with -inf)
template <int MB, int VEC>
__global__ __launch_bounds__(256) void k_maxpool_codes(const uint8_t* __restrict__ x, uint8_t* __restrict__ y, int64_t total, int H,
                                                       int W, int C, int Ho, int Wo, int KH, int KW, int SH, int SW, int PH, int PW) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / VEC;
    const int cg = (int)(idx % cv);
    int64_t pix = idx / cv;
    const int ow = (int)(pix % Wo); pix /= Wo;
    const int oh = (int)(pix % Ho);
    const int64_t n = pix / Ho;
    int best[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) best[e] = -(1 << 30);
    for (int kh = 0; kh < KH; ++kh) {
        const int ih = oh * SH - PH + kh;
        if ((unsigned)ih >= (unsigned)H) continue;
        for (int kw = 0; kw < KW; ++kw) {
            const int iw = ow * SW - PW + kw;
            if ((unsigned)iw >= (unsigned)W) continue;
            const uint8_t* px = x + (((size_t)n * H + ih) * W + iw) * C + (size_t)cg * VEC;
            uint32_t v[VEC / 4];
            if constexpr (VEC == 16) {
                const uint4 t = *reinterpret_cast<const uint4*>(px);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                v[0] = *reinterpret_cast<const uint32_t*>(px);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int k = code_key<MB>((v[e >> 2] >> (8 * (e & 3))) & 0xFFu);
                best[e] = k > best[e] ? k : best[e];
            }
        }
    }
    uint32_t o[VEC / 4];
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) o[q] = 0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e >> 2] |= key_code<MB>(best[e]) << (8 * (e & 3));
    uint8_t* py = y + (((size_t)n * Ho + oh) * Wo + ow) * C + (size_t)cg * VEC;
    if constexpr (VEC == 16) *reinterpret_cast<uint4*>(py) = make_uint4(o[0], o[1], o[2], o[3]);
    else *reinterpret_cast<uint32_t*>(py) = o[0];
}

}  // namespace slfp

extern "C" int slfp_maxpool2d_codes(const uint8_t* x, uint8_t* y, int64_t n, int64_t h, int64_t w, int64_t c, int kh, int kw, int sh,
                                    int sw, int ph, int pw, int qbits, void* stream) {
    using namespace slfp;
    if (!x || !y) return fail(SLFP_ERR_BAD_ARG, "slfp_maxpool2d_codes: null pointer");
    if (qbits != 8 && qbits != 7) return fail(SLFP_ERR_BAD_ARG, "slfp_maxpool2d_codes: qbits must be 8 or 7");
    if (n < 0 || h <= 0 || w <= 0 || c <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 || 2 * ph > kh || 2 * pw > kw)
        return fail(SLFP_ERR_SHAPE, "slfp_maxpool2d_codes: bad geometry");
    if (c % 4) return fail(SLFP_ERR_UNSUPPORTED, "slfp_maxpool2d_codes: channel count must be a multiple of 4");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return fail(SLFP_ERR_ALIGNMENT, "slfp_maxpool2d_codes: 16-byte aligned pointers");
    const int64_t ho = (h + 2 * ph - kh) / sh + 1, wo = (w + 2 * pw - kw) / sw + 1;   // floor mode (ceil_mode = False)
    if (ho <= 0 || wo <= 0) return fail(SLFP_ERR_SHAPE, "slfp_maxpool2d_codes: empty output");
    if (n == 0) return SLFP_OK;
    const int vec = (c % 16 == 0) ? 16 : 4;
    const int64_t total = n * ho * wo * (c / vec);
    const int64_t grid = (total + 255) / 256;
    if (grid > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "slfp_maxpool2d_codes: grid too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define SLFP_POOL(MBV, VECV) hipLaunchKernelGGL((k_maxpool_codes<MBV, VECV>), dim3((unsigned)grid), dim3(256), 0, st, x, y, total, (int)h, (int)w, \
                                                (int)c, (int)ho, (int)wo, kh, kw, sh, sw, ph, pw)
    if (qbits == 8) { if (vec == 16) SLFP_POOL(4, 16); else SLFP_POOL(4, 4); }
    else { if (vec == 16) SLFP_POOL(3, 16); else SLFP_POOL(3, 4); }
#undef SLFP_POOL
    return check_launch("slfp maxpool (codes) kernel");
}
