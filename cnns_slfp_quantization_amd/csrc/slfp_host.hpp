// slfp_host.hpp -- host-side helpers shared by the translation units of libslfp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include "../../include/slfp.h"
#include "slfp_device.hpp"
#include "slfp_enc.hpp"

namespace slfp {

// thread-local last-error text (slfp_last_error()).
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
// Returns SLFP_OK or SLFP_ERR_HIP (and records the HIP error string) after a launch.
int check_launch(const char* what);

// Kernels that need more than 64 KiB of dynamic LDS must opt in with hipFuncSetAttribute.  The attribute belongs to the
// (device, function) pair, so it is set once per pair, under a lock (any host thread may call the library; one process
// may drive several devices).  Returns SLFP_OK or an error status.
int raise_lds_limit(const void* fn, size_t lds_bytes);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Host side of ScaleDiv (slfp_device.hpp): r = RN(1/d) via double (1/d of a float is never
// within 2^-53 of a float rounding midpoint unless it is exact, so the double rounding is
// harmless).  scale_div_ok() is the range the FMA chain is valid for.
inline bool scale_div_ok(float d) { return std::isfinite(d) && d >= 1e-30f && d <= 1e30f; }

inline ScaleDiv make_scale_div(float d, int esh = 0) {
    ScaleDiv s;
    s.d = std::ldexp(d, -esh);  // dividing by d/2^esh == multiplying the quotient by 2^esh, exactly
    s.r = (float)(1.0 / (double)s.d);
    return s;
}

// Experiment switches (profiles/*.sh, profiles/variants.py).  They are environment variables read ONCE, when the library is
// first used (and again only on slfp_debug_reload_switches()): no getenv() is reachable from a launch.
//   SLFP_LONG_ENCODE    keep every kernel on the long-form quantizer (slfp_device.hpp) instead of the threshold table
//   SLFP_DW_OLD         keep every depthwise layer on conv_dw.hip
//   SLFP_PW_NOSTG       pointwise kernels without the LDS-staged whole-line stores
//   SLFP_PW_NOTAB       pointwise kernels on the long-form quantizer
//   SLFP_PW_STG_MAXKS   stream kernel: staged stores up to this many k-steps (default: 1, 2 and 8)
//   SLFP_PW_NT_MIN_MB / SLFP_DW_NT_MIN_MB   output size from which staged / depthwise stores carry the nt hint
struct Switches {
    bool long_encode, dw_old, pw_nostg, pw_notab;
    int pw_stg_maxks;         // -1: default rule
    long long pw_nt_min_mb;   // default 0
    long long dw_nt_min_mb;   // default 30
    bool stem_old;            // SLFP_STEM_OLD: the MobileNetV1 stem on the vector ALU (k_stem_fixed) instead of the float32 MFMA kernel (k_stem_mx)
    bool pwc_slice;           // SLFP_PWC_NOSLICE unsets it: deep code-path pointwise layers on k_pwc_tiled / k_pwc_stream instead of k_pwc_slice
    bool dense_generic;       // SLFP_DENSE_GENERIC: 3x3 stride-1 layers on the general k_dense_mfma instead of the unrolled k_dense3x3
    int dense_cfg;            // SLFP_DENSE_CFG=<wm><wn><mt> (e.g. 244): force a dense k x k tiling where it fits (sweeps); 0 = cost model
    int dense_nwb;            // SLFP_DENSE_NWB=2: keep two weight buffers everywhere (A/B of the three-buffer pipeline)
    bool dense_encx;          // SLFP_DENSE_NOENCX unsets it: k_dense3x3_res encodes the float32 halo on load (C_in == 64, C_out <= 64) instead of reading the pre-pass's fp16 copy
    bool dense_res;           // SLFP_DENSE_NORES unsets it: 3x3 stride-1 layers with C_in <= 64 on the persistent weights-resident k_dense3x3_res
    bool stem_im2row;         // SLFP_STEM_IM2ROW: large-kernel stems through the im2row workspace (the round-1 form; A/B of k_stem_rows)
    int pw_stream_max_kb;     // SLFP_PW_STREAM_MAX_KB: largest W (KiB, fp16) the float32-interface path gives to the LDS-resident stream kernel (default 30; its capacity is 128)
};
const Switches& switches();
void reload_switches();
inline bool long_encode_forced() { return switches().long_encode; }
inline bool dw_old_forced() { return switches().dw_old; }
// The table for the activation side of a layer, or nullptr when the long form has to be used.
inline const EncArgs* act_table(float ka, int fmt_act, int rep) {
    if (long_encode_forced()) return nullptr;
    const EncArgs* t = enc_table(ka, fmt_act, rep);
    return t->valid ? t : nullptr;
}

// Kernel families (slfp_conv2d_kernel_name reports them).
enum KernelFamily { kDw3x3 = 0, kPointwise = 1, kDirect = 2, kDenseMfma = 3, kStemMfma = 4, kStemSmall = 5 };

struct ConvPlan {
    KernelFamily family;
    int fmt_act;   // kFmtAct8 | kFmtSfp7
    int fmt_w;     // kFmtW8   | kFmtSfp7
    int passes;    // MFMA passes for the pointwise family (1 or 3)
    int64_t h_out, w_out;
    float s1, s2;  // epilogue: out = ((acc + bias/s1/s2) * s1) * s2  (conv: Ka,Kw; linear: Kw,Ka)
    // pointwise: padded K (multiple of 64) and N (multiple of 64) of the fragment-ordered blob
    int64_t k_pad, n_pad;
    size_t wprep_bytes;
    // Channel counts that are not a multiple of 4 (ShuffleNetV2's 58-channel branches) break the
    // 16-byte NHWC access the depthwise / pointwise kernels are built on.  Such layers run those
    // same kernels on channel-padded copies: x -> [..][cpi] (zeros appended: Q(0) = 0 and the
    // padded weights are 0, so results are unchanged), y <- [..][cpo].  `family` etc. describe
    // the inner (padded) launch.
    bool repad;
    int64_t cpi, cpo;
};

// Validates `d` and fills `plan`; returns SLFP_OK or an error status (error text recorded).
int make_plan(const slfp_conv2d_desc* d, ConvPlan* plan);

// ---- launchers implemented next to their kernels (all NHWC, all async on `stream`) ----
int launch_quantize(const float* x, float* y, size_t n, float scale, int fmt, hipStream_t stream);
int launch_dw3x3(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const float* wq9c,
                 const float* bias, const PostOp& post, float* y, hipStream_t stream);
// the straight-line depthwise tile kernel (conv_dw2.hip): C a multiple of 32, threshold table available
bool dw3x3_tile_applicable(const slfp_conv2d_desc& d, const ConvPlan& p, const float* bias, const PostOp& post);
int launch_dw3x3_tile(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const float* wq9c,
                      const PostOp& post, float* y, hipStream_t stream);
int launch_pointwise(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const void* wfrag,
                     const float* bias, const PostOp& post, float* y, hipStream_t stream);
// ---- 1-byte activation codes between layers (slfp_codes.hpp; slfp_conv2d_fwd_codes) ----
// How a layer's input arrives and its output leaves: float32, or the extended codes of QA(. / ka) in format fmt.
struct CodeIo {
    bool x_codes;     // x holds uint8 codes of QA(x / d.ka) (format of d.qbits) instead of float32
    bool y_codes;     // y receives uint8 codes of QA(out / y_ka) in format y_fmt instead of float32
    float y_ka;
    int y_fmt;        // kFmtAct8 | kFmtSfp7
};
// Workgroups of `fn` (block size, dynamic LDS) that fit one CU at once (hipOccupancyMaxActiveBlocksPerMultiprocessor, cached
// per device / function / LDS size): how persistent grids are sized.  >= 1.
int resident_blocks_per_cu(const void* fn, int block_threads, size_t dynamic_lds);
// Number of compute units of the current device (cached per device; 256 on MI355X): persistent grids are sized from it.
int device_cu_count();
// pointwise on codes (conv_pw_codes.hpp): codes in, codes or float32 out; the SAME prepared blob as launch_pointwise
bool pwc_applicable(const slfp_conv2d_desc& d, const ConvPlan& p, int post_flags, bool y_codes);
int launch_pwc(const slfp_conv2d_desc& d, const ConvPlan& p, const uint8_t* x, const void* wfrag, const float* bias,
               const PostOp& post, void* y, bool y_codes, float y_ka, int y_fmt, hipStream_t stream);
// the MobileNetV1 image stem with code output (conv_direct.hip)
bool stem_codes_applicable(const slfp_conv2d_desc& d, const ConvPlan& p, int post_flags);
int launch_stem_codes(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const float* wq_hwio, const float* bias,
                      const PostOp& post, void* y, const CodeIo& io, hipStream_t stream);
// depthwise 3x3 on codes (conv_dwc.hip): codes in, codes or float32 out
bool dwc_applicable(const slfp_conv2d_desc& d, const ConvPlan& p, const float* bias, int post_flags);
int launch_dwc(const slfp_conv2d_desc& d, const ConvPlan& p, const uint8_t* x, const float* wq9c, const PostOp& post,
               void* y, bool y_codes, float y_ka, int y_fmt, hipStream_t stream);
int launch_direct(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const float* wq_hwio,
                  const float* bias, const PostOp& post, float* y, hipStream_t stream);
bool stem_applicable(const slfp_conv2d_desc& d);  // direct family: the small-C_in stem kernel takes it
// pointwise: does W (padded) fit the LDS-resident stream kernel?  (that kernel also takes even channel
// counts that are not a multiple of 4, with 8-byte accesses)
bool pointwise_stream_fits(int64_t k_pad, int64_t n_pad, int passes);
// dense k x k implicit GEMM on MFMA (conv_dense.hip); wblob = [tap][n_tile][k_step][64][8] fp16;
// `workspace` (dense_mfma_workspace_bytes) receives the input encoded once to fp16
bool dense_mfma_applicable(const slfp_conv2d_desc& d, int passes);
size_t dense_mfma_workspace_bytes(const slfp_conv2d_desc& d, int passes);
int launch_dense_mfma(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const void* wblob,
                      const float* bias, const PostOp& post, float* y, void* workspace, hipStream_t stream);
// the same with 1-byte codes in and / or out (x_any: float32 or uint8 codes of QA(x / d.ka); y_any: float32 or uint8 codes)
bool dense_codes_applicable(const slfp_conv2d_desc& d, const ConvPlan& p, int post_flags, bool y_codes);
int launch_dense_mfma_io(const slfp_conv2d_desc& d, const ConvPlan& p, const void* x_any, const void* wblob, const float* bias,
                         const PostOp& post, void* y_any, void* workspace, const CodeIo& io, hipStream_t stream);
// codes: `w_oihw` points at 1-byte extended weight codes (slfp_encode_f32 | SLFP_FMT_EXT) instead of float32 weights
int launch_prepare_weights(const slfp_conv2d_desc& d, const ConvPlan& p, const float* w_oihw, void* wprep,
                           float* weight_q_oihw, hipStream_t stream, bool codes = false);

// large-kernel image stems on MFMA (conv_stem_mfma.hip); wblob = [kh*ksub + sub][nt][64][8] fp16;
// `workspace` receives the im2row'ed, encoded input
bool stem_mfma_applicable(const slfp_conv2d_desc& d, int passes);
void stem_mfma_blob_shape(const slfp_conv2d_desc& d, int* ksub, int* nt);
size_t stem_mfma_workspace_bytes(const slfp_conv2d_desc& d, int64_t w_out);
int launch_stem_mfma(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const void* wblob,
                     const float* bias, const PostOp& post, float* y, void* workspace, hipStream_t stream);

// 3x3-class image stems whose whole contraction is one MFMA k-step (conv_stem_small.hip);
// wblob = [o/16][64][8] fp16 with k = (kh*KW + kw)*C_in + c
bool stem_small_applicable(const slfp_conv2d_desc& d, int passes);
int stem_small_tiles(const slfp_conv2d_desc& d);
int launch_stem_small(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const void* wblob,
                      const float* bias, const PostOp& post, float* y, hipStream_t stream);
bool stem_small_codes_applicable(const slfp_conv2d_desc& d, const ConvPlan& p, int post_flags);   // float32 in -> codes out (C_out == 64)
int launch_stem_small_io(const slfp_conv2d_desc& d, const ConvPlan& p, const float* x, const void* wblob, const float* bias,
                         const PostOp& post, void* y_any, const CodeIo& io, hipStream_t stream);

// XCD-aware block remap (MI355X: 8 XCDs, blocks are dealt round-robin over them, so
// blocks b and b+8 share an L2).  Maps the hardware block id to a logical id such that
// each XCD owns a CONTIGUOUS range of logical ids: neighbouring tiles (shared halos,
// shared X rows of a column-split GEMM) then hit the same L2.  Bijective for any grid.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks) {
    const uint32_t xcd = bid & 7u, slot = bid >> 3;
    const uint32_t q = nblocks >> 3, r = nblocks & 7u;
    const uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + slot;
}

}  // namespace slfp
