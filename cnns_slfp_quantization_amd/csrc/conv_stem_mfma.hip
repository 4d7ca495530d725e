// conv_stem_mfma.hip -- large-kernel image stems on the matrix cores: ResNet-50 7x7 s2 3->64
// (nets_imgnet/resnet.py), SqueezeNet 7x7 s2 3->96 (+bias), AlexNet 11x11 s4 3->64 (+bias).
//
// Replaces Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) for groups == 1, C_in <= 4,
// KH*KW >= 25.  At 147-363 MACs per output value these layers are compute-bound on the fp32
// VALU path (k_stem / k_direct reach ~30 TFLOP/s), yet their HBM floor is the output write alone.
// With NHWC and C_in = 3 the KW*C_in input values one tap ROW contributes to an output pixel are
// CONTIGUOUS in memory, so the contraction is arranged as K = KH x (row run padded to Rp = 32 or 64):
//   1. k_stem_im2row: for every input row ih and output column ow, the run
//      x[n][ih][ow*S-pw .. +KW-1][0..C) is encoded (x/Ka, SLFP) once to fp16 (x16) and written,
//      zero-padded to Rp, to the workspace: xe[n][ih][ow][Rp].  16 consecutive output pixels of
//      a row are then 16*Rp contiguous halfs = exactly the B fragments of Rp/32 k-steps;
//   2. k_stem_mfma: all of W (fragment-ordered [kh*Rp/32 + sub][channel tile], 28-88 KiB) stays in
//      LDS; a wave owns 16 output pixels x all output channels, loads its B fragments straight
//      from xe (1 KiB contiguous per k-step, L2/MALL hits: each run is reused by KH/S output rows),
//      16x16x32 fp16 MFMA, float32 accumulation, the reference's (out*Ka)*Kw roundings, bias and
//      the optional fused BN/ReLU post-op in the epilogue.  Tap rows that fall in the vertical
//      padding are skipped (wave-uniform).
// Precision: single-pass fp16 (SLFP<3,4>) / exact (SFP<3,3>), as conv_dense.hip.
#include "slfp_device.hpp"
#include "slfp_host.hpp"

namespace slfp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kSmThreads = 512;

struct StemMfmaParams {
    const _Float16* xe;  // [N][H][Wo][Rp]
    const _Float16* w;   // [KS][NT][64 lanes][8]
    const float* bias;
    float* y;
    int N, H, O, KH, S, ph, Ho, Wo;
    int rp_shift;        // Rp = 32 << rp_shift... (5 or 6: log2 Rp)
    int KS;              // KH * Rp/32
    int segs;            // ceil(Wo / 16)
    int64_t units;       // N * Ho * segs
    float s1, s2, s1x;
    PostOp post;
};

template <int FMT>
__global__ __launch_bounds__(256) void k_stem_im2row(const float* __restrict__ x, _Float16* __restrict__ xe,
                                                     int64_t n_chunks16, int H, int W, int C, int Wo, int S, int pw,
                                                     int RL, int rp_shift, const ScaleDiv sd) {
    __shared__ uint32_t sT[16];
    lut_fill<FMT>(sT);
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_chunks16) return;
    const int cps = rp_shift - 3;                       // log2(16-byte chunks per run)
    const int cj = (int)(idx & ((1 << cps) - 1));
    const int64_t run = idx >> cps;                     // (n*H + ih)*Wo + ow
    const int ow = (int)(run % Wo);
    const int64_t row = run / Wo;                       // n*H + ih
    const float* xr = x + row * (int64_t)W * C;
    const int e0 = (ow * S - pw) * C + cj * 8;          // element offset inside the input row
    half8 h = half8{0, 0, 0, 0, 0, 0, 0, 0};
    if (cj * 8 < RL) {                                   // chunks that are all padding just store zeros
        const int last = W * C - 1;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                    // unconditional (clamped) loads: all 8 in flight
            const int e = e0 + j;
            v[j] = xr[e < 0 ? 0 : (e > last ? last : e)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = cj * 8 + j, e = e0 + j;
            const float q = quantize_scaled<FMT, 4>(v[j], sd, sT);
            h[j] = (_Float16)((r < RL && e >= 0 && e <= last) ? q : 0.f);
        }
    }
    *reinterpret_cast<half8*>(xe + idx * 8) = h;
}

// NT = 16-channel tiles (4: C_out <= 64, 6: C_out <= 96).  KB = k-steps whose B fragments are in flight.
template <int NT, int KB>
__global__ __launch_bounds__(kSmThreads, 2) void k_stem_mfma(const StemMfmaParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // W: [KS][NT][1 KiB]
    {
        const int n16 = p.KS * NT * 64;
        const uint4* src = reinterpret_cast<const uint4*>(p.w);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < n16; i += kSmThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int ksub = 1 << (p.rp_shift - 5);
    // a workgroup owns a CONTIGUOUS range of (image, output row, 16-pixel segment) units: its 8 waves
    // sweep a row's segments together and move down row by row, so the KH/S re-reads of every
    // im2row line come from this CU's L1 / this XCD's L2 (r01f_resnet50: a strided assignment
    // re-fetched the fp16 image 3.3x through the fabric)
    const int64_t chunk = (p.units + gridDim.x - 1) / gridDim.x;
    const int64_t u_begin = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * chunk;
    const int64_t u_end = u_begin + chunk < p.units ? u_begin + chunk : p.units;
    // this lane's channels are the same for every unit: bias / BN vectors once, not per store
    float4 bqv[NT];
    PostVec pvv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ch = j * 16 + kq * 4;
        bqv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        pvv[j] = PostVec{make_float4(1.f, 1.f, 1.f, 1.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        if (ch < p.O) {
            if (p.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + ch);
                bqv[j] = make_float4(256.f * ((bb.x / p.s1) / p.s2), 256.f * ((bb.y / p.s1) / p.s2),
                                     256.f * ((bb.z / p.s1) / p.s2), 256.f * ((bb.w / p.s1) / p.s2));
            }
            pvv[j] = post_load(p.post, ch);
        }
    }
    for (int64_t u = u_begin + wave; u < u_end; u += kSmThreads / 64) {
        const int seg = (int)(u % p.segs);
        const int64_t t = u / p.segs;
        const int oh = (int)(t % p.Ho);
        const int n = (int)(t / p.Ho);
        const int ow = seg * 16 + col;
        const int owc = ow < p.Wo ? ow : p.Wo - 1;      // clamp the loads of a ragged last segment
        const int ih0 = oh * p.S - p.ph;
        const _Float16* xl = p.xe + ((((int64_t)n * p.H) * p.Wo + owc) << p.rp_shift) + kq * 8;
        floatx4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int ks0 = 0; ks0 < p.KS; ks0 += KB) {
            half8 xf[KB];
#pragma unroll
            for (int q = 0; q < KB; ++q) {
                const int ks = ks0 + q;
                const int kh = ks >> (p.rp_shift - 5), sub = ks & (ksub - 1);
                const int ih = ih0 + kh;
                xf[q] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (ks < p.KS && (unsigned)ih < (unsigned)p.H)   // wave-uniform
                    xf[q] = *reinterpret_cast<const half8*>(xl + (((int64_t)ih * p.Wo) << p.rp_shift) + sub * 32);
            }
#pragma unroll
            for (int q = 0; q < KB; ++q) {
                const int ks = ks0 + q;
                if (ks >= p.KS || (unsigned)(ih0 + (ks >> (p.rp_shift - 5))) >= (unsigned)p.H) continue;
                const unsigned char* wt = smem + ((size_t)ks * NT) * 1024 + lane * 16;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const half8 wf = *reinterpret_cast<const half8*>(wt + j * 1024);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[q], acc[j], 0, 0, 0);
                }
            }
        }
        if (ow < p.Wo) {
            float* yp = p.y + (((int64_t)n * p.Ho + oh) * p.Wo + ow) * p.O;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int ch = j * 16 + kq * 4;
                if (ch >= p.O) continue;
                const float4 bq = bqv[j];
                float4 r;
                r.x = ((acc[j][0] + bq.x) * p.s1x) * p.s2;
                r.y = ((acc[j][1] + bq.y) * p.s1x) * p.s2;
                r.z = ((acc[j][2] + bq.z) * p.s1x) * p.s2;
                r.w = ((acc[j][3] + bq.w) * p.s1x) * p.s2;
                st_stream4<SLFP_NT_STEM_MFMA>(yp + ch, post_apply_v(r, p.post, pvv[j]));
            }
        }
    }
}

// ======================================================================================
// k_stem_rows (round 3): the same contraction WITHOUT the im2row workspace.  A workgroup owns an 8 x 32 tile of output
// pixels; the input rows it needs ((8-1)*S + KH rows x ((32-1)*S + KW)*C elements) are loaded once, encoded to fp16 and kept
// in LDS as plain rows (the encoded image never touches HBM: the im2row copy was 400 MB written + re-read per step on
// SqueezeNet's 7x7 s2 stem at batch 256, next to a 154 MB input and a 1.17 GB output).  With C_in = 3 a tap row's KW*C inputs
// are contiguous in such a row, so lane (pixel, kq) of k-step (kh, sub) reads elements sub*32 + kq*8 .. +7 behind its
// pixel's window origin -- 4-byte aligned when S*C is even: four ds_read_b32.  Elements past the run (r >= KW*C) are masked to
// zero, as the im2row copy stores them (the blob's weights are zero there too, but 0 * NaN must not leak in).  Same blob, same k order, rows in the vertical
// padding contribute exact zeros instead of being skipped: bit-identical to k_stem_im2row + k_stem_mfma.
// ======================================================================================
constexpr int kSrThreads = 256, kSrTH = 8, kSrTW = 32;

struct StemRowsParams {
    const float* x;
    const _Float16* w;   // [KS][NT][64 lanes][8]
    const float* bias;
    float* y;
    int N, H, W, C, O, KH, S, ph, pw, Ho, Wo;
    int tiles_h, tiles_w;
    int IH, IWC, row_h;  // halo tile: IH rows of IWC elements, row pitch row_h halfs (even, >= IWC + Rp - RL; the tail is zero)
    int ksub, KS, RL;    // RL = KW * C: real elements of a tap row's run
    uint32_t w_off;      // byte offset of the W fragments inside the dynamic LDS
    ScaleDiv sd;
    float s1, s2, s1x;
    PostOp post;
    uint32_t nblocks;
};

template <int FMT, int NT>
__global__ __launch_bounds__(kSrThreads) void k_stem_rows(const StemRowsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sT = reinterpret_cast<uint32_t*>(smem);
    _Float16* tile = reinterpret_cast<_Float16*>(smem + 64);   // [IH][row_h]
    lut_fill<FMT>(sT);
    {   // W: all k-steps resident
        const int n16 = p.KS * NT * 64;
        const uint4* src = reinterpret_cast<const uint4*>(p.w);
        uint4* dst = reinterpret_cast<uint4*>(smem + p.w_off);
        for (int i = threadIdx.x; i < n16; i += kSrThreads) dst[i] = src[i];
    }
    uint32_t b = xcd_remap(blockIdx.x, p.nblocks);
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int n = b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int h_in0 = th * kSrTH * p.S - p.ph, w_in0 = tw * kSrTW * p.S - p.pw;
    __syncthreads();  // LUT visible
    {   // halo rows: coalesced dword loads, encode once, fp16 to LDS; the pitch's tail and out-of-image elements are zero
        const float* xn = p.x + (size_t)n * p.H * p.W * p.C;
        const int n_el = p.IH * p.row_h;
        const int e_w0 = w_in0 * p.C;
        const int row_el = p.W * p.C;
        constexpr int U = 8;
        for (int base = threadIdx.x; base < n_el; base += kSrThreads * U) {
            float v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * kSrThreads;
                const int ih = i / p.row_h, e = i - ih * p.row_h;
                const int gh = h_in0 + ih, ge = e_w0 + e;
                const bool live = i < n_el;
                const bool inb = live && e < p.IWC && (unsigned)gh < (unsigned)p.H && ge >= 0 && ge < row_el;
                dst[u] = live ? i : -1;
                v[u] = xn[inb ? gh * row_el + ge : 0];   // unconditional (clamped) load
                if (!inb) v[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) tile[dst[u]] = (_Float16)quantize_scaled<FMT, 4>(v[u], p.sd, sT);
        }
    }
    float4 bqv[NT];
    PostVec pvv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ch = j * 16 + kq * 4;
        bqv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        pvv[j] = PostVec{make_float4(1.f, 1.f, 1.f, 1.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        if (ch < p.O) {
            if (p.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + ch);
                bqv[j] = make_float4(256.f * ((bb.x / p.s1) / p.s2), 256.f * ((bb.y / p.s1) / p.s2),
                                     256.f * ((bb.z / p.s1) / p.s2), 256.f * ((bb.w / p.s1) / p.s2));
            }
            pvv[j] = post_load(p.post, ch);
        }
    }
    __syncthreads();
    const unsigned char* wl = smem + p.w_off + lane * 16;
    // elements past the run (r >= KW*C) are neighbouring pixels' data: masked to +0 so that a NaN / Inf there cannot reach this
    // pixel through 0 * NaN (the im2row form stores zeros there)
    uint32_t rmask[2][4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int dw = 0; dw < 4; ++dw) {
            const int r0 = sub * 32 + kq * 8 + dw * 2;
            rmask[sub][dw] = (r0 < p.RL ? 0x0000FFFFu : 0u) | (r0 + 1 < p.RL ? 0xFFFF0000u : 0u);
        }
    for (int u = wave; u < kSrTH * (kSrTW / 16); u += kSrThreads / 64) {
        const int orow = u >> 1, seg = u & 1;
        const int goh = th * kSrTH + orow, gow = tw * kSrTW + seg * 16 + col;
        const uint32_t* origin = reinterpret_cast<const uint32_t*>(tile + (orow * p.S) * p.row_h + ((seg * 16 + col) * p.S) * p.C + kq * 8);
        floatx4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < p.KH; ++kh) {
            for (int sub = 0; sub < p.ksub; ++sub) {
                const uint32_t* src = origin + ((kh * p.row_h + sub * 32) >> 1);
                typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));
                const half8 xf = __builtin_bit_cast(half8, u32x4r{src[0] & rmask[sub][0], src[1] & rmask[sub][1],
                                                                  src[2] & rmask[sub][2], src[3] & rmask[sub][3]});
                const unsigned char* wt = wl + (size_t)((kh * p.ksub + sub) * NT) * 1024;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const half8 wf = *reinterpret_cast<const half8*>(wt + j * 1024);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[j], 0, 0, 0);
                }
            }
        }
        if (goh < p.Ho && gow < p.Wo) {
            float* yp = p.y + (((size_t)n * p.Ho + goh) * p.Wo + gow) * p.O;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int ch = j * 16 + kq * 4;
                if (ch >= p.O) continue;
                const float4 bq = bqv[j];
                float4 r;
                r.x = ((acc[j][0] + bq.x) * p.s1x) * p.s2;
                r.y = ((acc[j][1] + bq.y) * p.s1x) * p.s2;
                r.z = ((acc[j][2] + bq.z) * p.s1x) * p.s2;
                r.w = ((acc[j][3] + bq.w) * p.s1x) * p.s2;
                st_stream4<SLFP_NT_STEM_MFMA>(yp + ch, post_apply_v(r, p.post, pvv[j]));
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------
static int stem_rp(const slfp_conv2d_desc& d) { return (int)ceil_div(d.kw * d.c_in, 32) * 32; }

bool stem_mfma_applicable(const slfp_conv2d_desc& d, int passes) {
    if (d.groups != 1 || d.c_in > 4 || d.dil_h != 1 || d.dil_w != 1 || d.stride_h != d.stride_w) return false;
    if (d.kh * d.kw < 25) return false;                  // 3x3 / 5x5-ish stems: k_stem is at the HBM floor already
    if (d.c_out % 4 || d.c_out > 96) return false;
    if (d.qbits == 8 && passes == 3) return false;      // the float32-equivalent mode stays on k_stem / k_direct
    const int rp = stem_rp(d);
    if (rp > 64) return false;
    const int nt = d.c_out <= 64 ? 4 : 6;
    return (size_t)d.kh * (rp / 32) * nt * 1024 <= 150 * 1024;
}

void stem_mfma_blob_shape(const slfp_conv2d_desc& d, int* ksub, int* nt) {
    *ksub = stem_rp(d) / 32;
    *nt = d.c_out <= 64 ? 4 : 6;
}

// workspace = the im2row'ed, encoded input: N*H*Wo*Rp fp16
size_t stem_mfma_workspace_bytes(const slfp_conv2d_desc& d, int64_t w_out) {
    return (((size_t)d.n * d.h * w_out * stem_rp(d) * sizeof(_Float16)) + 255) & ~(size_t)255;
}

template <int NT>
static int launch_stem_mfma_t(const StemMfmaParams& p, size_t lds, unsigned grid, hipStream_t stream) {
    auto fn = k_stem_mfma<NT, 8>;
    const int rc = raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024);  // once per (device, kernel)
    if (rc != SLFP_OK) return rc;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kSmThreads), lds, stream, p);
    return check_launch("slfp MFMA stem kernel");
}

// the fused form: even S*C (4-byte aligned fragment reads), everything in one workgroup's LDS
static bool stem_rows_geometry(const slfp_conv2d_desc& d, StemRowsParams* p, size_t* lds, int* nt_out) {
    if ((d.stride_w * d.c_in) % 2) return false;
    int ksub, nt;
    stem_mfma_blob_shape(d, &ksub, &nt);
    const int rp = stem_rp(d), rl = (int)(d.kw * d.c_in);
    p->IH = (kSrTH - 1) * d.stride_h + (int)d.kh;
    p->IWC = ((kSrTW - 1) * d.stride_w + (int)d.kw) * (int)d.c_in;
    p->row_h = (p->IWC + (rp - rl) + 1) & ~1;
    p->ksub = ksub;
    p->KS = (int)d.kh * ksub;
    p->RL = rl;
    p->w_off = (uint32_t)((64 + (size_t)p->IH * p->row_h * sizeof(_Float16) + 15) & ~(size_t)15);
    *lds = p->w_off + (size_t)p->KS * nt * 1024;
    *nt_out = nt;
    return *lds <= 64 * 1024 && (int64_t)d.h * d.w * d.c_in < (1ll << 30);
}

int launch_stem_mfma(const slfp_conv2d_desc& d, const ConvPlan& plan, const float* x, const void* wblob,
                     const float* bias, const PostOp& post, float* y, void* workspace, hipStream_t stream) {
    {
        StemRowsParams q;
        size_t lds;
        int nt;
        if (!switches().stem_im2row && stem_rows_geometry(d, &q, &lds, &nt)) {
            q.x = x; q.w = reinterpret_cast<const _Float16*>(wblob); q.bias = bias; q.y = y; q.post = post;
            q.N = (int)d.n; q.H = (int)d.h; q.W = (int)d.w; q.C = (int)d.c_in; q.O = (int)d.c_out; q.KH = (int)d.kh;
            q.S = d.stride_h; q.ph = d.pad_h; q.pw = d.pad_w; q.Ho = (int)plan.h_out; q.Wo = (int)plan.w_out;
            q.tiles_h = (int)ceil_div(q.Ho, kSrTH); q.tiles_w = (int)ceil_div(q.Wo, kSrTW);
            q.sd = make_scale_div(d.ka, 4);
            q.s1 = plan.s1; q.s2 = plan.s2; q.s1x = plan.s1 * (1.0f / 256.0f);
            const int64_t nblocks = (int64_t)q.N * q.tiles_h * q.tiles_w;
            if (nblocks > 0x7FFFFFFF) return fail(SLFP_ERR_UNSUPPORTED, "MFMA stem: grid too large");
            q.nblocks = (uint32_t)nblocks;
            const bool a8 = plan.fmt_act == kFmtAct8;
            if (nt == 4) {
                if (a8) hipLaunchKernelGGL((k_stem_rows<kFmtAct8, 4>), dim3(q.nblocks), dim3(kSrThreads), lds, stream, q);
                else hipLaunchKernelGGL((k_stem_rows<kFmtSfp7, 4>), dim3(q.nblocks), dim3(kSrThreads), lds, stream, q);
            } else {
                if (a8) hipLaunchKernelGGL((k_stem_rows<kFmtAct8, 6>), dim3(q.nblocks), dim3(kSrThreads), lds, stream, q);
                else hipLaunchKernelGGL((k_stem_rows<kFmtSfp7, 6>), dim3(q.nblocks), dim3(kSrThreads), lds, stream, q);
            }
            return check_launch("slfp MFMA stem (rows in LDS) kernel");
        }
    }
    if (!workspace) return fail(SLFP_ERR_BAD_ARG, "MFMA stem: workspace required (slfp_conv2d_workspace_bytes)");
    const int rp = stem_rp(d), rp_shift = rp == 32 ? 5 : 6;
    int ksub, nt;
    stem_mfma_blob_shape(d, &ksub, &nt);
    _Float16* xe = reinterpret_cast<_Float16*>(workspace);
    const int64_t n_chunks16 = d.n * d.h * plan.w_out * (rp / 8);
    const ScaleDiv sd = make_scale_div(d.ka, 4);
    const unsigned egrid = (unsigned)ceil_div(n_chunks16, 256);
    const int RL = (int)(d.kw * d.c_in);
    if (plan.fmt_act == kFmtAct8)
        hipLaunchKernelGGL((k_stem_im2row<kFmtAct8>), dim3(egrid), dim3(256), 0, stream, x, xe, n_chunks16, (int)d.h, (int)d.w,
                           (int)d.c_in, (int)plan.w_out, d.stride_w, d.pad_w, RL, rp_shift, sd);
    else
        hipLaunchKernelGGL((k_stem_im2row<kFmtSfp7>), dim3(egrid), dim3(256), 0, stream, x, xe, n_chunks16, (int)d.h, (int)d.w,
                           (int)d.c_in, (int)plan.w_out, d.stride_w, d.pad_w, RL, rp_shift, sd);
    int rc = check_launch("slfp stem im2row kernel");
    if (rc != SLFP_OK) return rc;
    StemMfmaParams p;
    p.xe = xe; p.w = reinterpret_cast<const _Float16*>(wblob); p.bias = bias; p.y = y; p.post = post;
    p.N = (int)d.n; p.H = (int)d.h; p.O = (int)d.c_out; p.KH = (int)d.kh; p.S = d.stride_h; p.ph = d.pad_h;
    p.Ho = (int)plan.h_out; p.Wo = (int)plan.w_out;
    p.rp_shift = rp_shift; p.KS = (int)d.kh * ksub;
    p.segs = (int)ceil_div(p.Wo, 16);
    p.units = (int64_t)p.N * p.Ho * p.segs;
    p.s1 = plan.s1; p.s2 = plan.s2; p.s1x = plan.s1 * (1.0f / 256.0f);
    const size_t lds = (size_t)p.KS * nt * 1024;
    const int occ = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds));
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(p.units, kSmThreads / 64), (int64_t)device_cu_count() * occ);
    return nt == 4 ? launch_stem_mfma_t<4>(p, lds, grid, stream) : launch_stem_mfma_t<6>(p, lds, grid, stream);
}

}  // namespace slfp
