// conv_pw_params.hpp -- what the pointwise (1x1) kernels of conv_pw.hip (and experiments built next to it) share: launch parameters,
// the fp16 tile swizzle and the epilogue arithmetic of Conv2d_Q.forward (utils/conv2d_func.py:23-24).
#pragma once
#include "slfp_device.hpp"
#include "slfp_enc.hpp"
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
    ScaleDiv sd;          // divides by Ka/16
    float s1, s2;         // out = ((acc/256 + bias/s1/s2) * s1) * s2 ; s1x = s1/256
    float s1x;
    uint32_t m_blocks, n_blocks, nblocks;
    int rb;               // k_pw_tiled: pixel rows a workgroup really owns (<= BM; the rest of its tile is padding)
    int nt_out;           // staged (whole-line) stores carry the nt hint: outputs too large for the Infinity Cache
    PostOp post;
    EncArgs enc;          // threshold table of fp16(16 * QA(x / Ka)) (TAB kernels; slfp_enc.hpp)
    EncArgsCompact enc_lo;   // three-pass TAB kernels: the residual plane's table (kEncF16LO) of the same scale
#ifdef SLFP_PW_STAMPS
    unsigned long long* dbg;   // diagnostic builds only (profiles/stamps_tiled.py): 16 x s_memrealtime per workgroup
#endif
};

#ifdef SLFP_PW_STAMPS
#define SLFP_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SLFP_STAMP(i) do { } while (0)
#endif

constexpr int kPwTab = (kEncEntries * 8 + 15) & ~15;   // LDS bytes of the threshold table


// out = ((acc * 2^-8 + bq) * s1) * s2 with the 2^-8 folded: ((acc + 256*bq) * (s1/256)) * s2
__device__ __forceinline__ float4 epilogue(const floatx4 acc, const float4 bq256, const float s1x, const float s2) {
    float4 r;
    r.x = ((acc[0] + bq256.x) * s1x) * s2;
    r.y = ((acc[1] + bq256.y) * s1x) * s2;
    r.z = ((acc[2] + bq256.z) * s1x) * s2;
    r.w = ((acc[3] + bq256.w) * s1x) * s2;
    return r;
}

__device__ __forceinline__ float4 bias_q256(const PwParams& p, int n) {
    if (!p.bias) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
    return make_float4(256.f * ((bb.x / p.s1) / p.s2), 256.f * ((bb.y / p.s1) / p.s2),
                       256.f * ((bb.z / p.s1) / p.s2), 256.f * ((bb.w / p.s1) / p.s2));
}

// 128-byte rows (64 fp16); XOR swizzle so that the 16 rows a fragment read touches hit
// 16 distinct 16-byte slots (conflict-free ds_read_b128; cdna guide T2)
__device__ __forceinline__ uint32_t lds_x_off(int row, int chunk16) {
    // 128-byte rows (64 fp16); XOR swizzle so that the 16 rows a fragment read touches hit
    // 16 distinct 16-byte slots (conflict-free ds_read_b128; cdna guide T2)
    return (uint32_t)row * 128u + (uint32_t)((chunk16 ^ (row & 7)) << 4);
}

constexpr int kStgRow = 272;   // 256 B of a row's 64 channels + 16: the 16 rows' float4 writes spread over the banks

}  // namespace slfp
