// codec.hip -- elementwise SLFP/SFP codec kernels + NCHW<->NHWC transposes (gfx950).
//
// Replaces the reference's ~25-pass ATen fake-quantizers (utils/sfp_quant.py:10-48,
// :59-97) by ONE HBM-bound pass: 16-byte loads, integer encode, 16-byte (or 4-byte code)
// stores.  Roofline: 8 B/element (quantize) or 5 B/element (encode) of HBM traffic.
#include <cstdarg>
#include <mutex>
#include <map>
#include <set>
#include <utility>
#include <cstdio>
#include "slfp_device.hpp"
#include "slfp_enc.hpp"
#include "slfp_host.hpp"

namespace slfp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SLFP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return SLFP_OK;
}

const char* last_error_text() { return g_err; }

int raise_lds_limit(const void* fn, size_t lds_bytes) {
    // lds_bytes: the DYNAMIC LDS of the launch (kernels may add a few KiB of static LDS: the lookup tables)
    if (lds_bytes <= 48 * 1024) return SLFP_OK;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> raised;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return check_launch("hipGetDevice");
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, fn);
    const auto it = raised.find(key);
    if (it != raised.end() && it->second >= lds_bytes) return SLFP_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return check_launch("hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    raised[key] = lds_bytes;
    return SLFP_OK;
}

static Switches read_switches() {
    Switches w;
    w.long_encode = std::getenv("SLFP_LONG_ENCODE") != nullptr;
    w.dw_old = std::getenv("SLFP_DW_OLD") != nullptr;
    w.pw_nostg = std::getenv("SLFP_PW_NOSTG") != nullptr;
    w.pw_notab = std::getenv("SLFP_PW_NOTAB") != nullptr;
    const char* e = std::getenv("SLFP_PW_STG_MAXKS");
    w.pw_stg_maxks = e ? atoi(e) : -1;
    e = std::getenv("SLFP_PW_NT_MIN_MB");
    w.pw_nt_min_mb = e ? atoll(e) : 0;
    e = std::getenv("SLFP_DW_NT_MIN_MB");
    w.dw_nt_min_mb = e ? atoll(e) : 30;   // round 3, 200-step same-box A/B (profiles/ab_env_long.sh): 30 / 0 / 120 / 210 MB -> 118.41 / 118.34 / 118.03 / 117.65 k images/s
    w.stem_old = std::getenv("SLFP_STEM_OLD") != nullptr;
    w.pwc_slice = std::getenv("SLFP_PWC_NOSLICE") == nullptr;
    w.dense_generic = std::getenv("SLFP_DENSE_GENERIC") != nullptr;
    e = std::getenv("SLFP_DENSE_CFG");
    w.dense_cfg = e ? atoi(e) : 0;
    e = std::getenv("SLFP_DENSE_NWB");
    w.dense_nwb = e ? atoi(e) : 0;
    w.dense_res = std::getenv("SLFP_DENSE_NORES") == nullptr;
    w.dense_encx = std::getenv("SLFP_DENSE_NOENCX") == nullptr;
    w.stem_im2row = std::getenv("SLFP_STEM_IM2ROW") != nullptr;
    e = std::getenv("SLFP_PW_STREAM_MAX_KB");
    w.pw_stream_max_kb = e ? atoi(e) : 30;   // round 3 (profiles/ab_env_long.sh): W above 30 KiB runs on the tiled kernel -- MobileNetV1 +0.3-0.4 % (128->128, 128->256, 256->256), ShuffleNetV2 14.8 -> 17.5 k images/s (its 116- / 232-channel layers), ResNet-50 / SqueezeNet unchanged
    return w;
}
static Switches g_switches = read_switches();   // once, at load
const Switches& switches() { return g_switches; }
void reload_switches() { g_switches = read_switches(); }

int device_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    std::lock_guard<std::mutex> lock(mu);
    const auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cus[dev] = n;
    return n;
}

int resident_blocks_per_cu(const void* fn, int block_threads, size_t dynamic_lds) {
    static std::mutex mu;
    static std::map<std::pair<std::pair<int, const void*>, size_t>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 1; }
    const auto key = std::make_pair(std::make_pair(dev, fn), dynamic_lds);
    std::lock_guard<std::mutex> lock(mu);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, block_threads, dynamic_lds) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    cache[key] = n;
    return n;
}

constexpr int kThreads = 256;

// MODE 0: y = Q(x/scale) float32; MODE 1: code byte.
template <int FMT, int MODE>
__global__ __launch_bounds__(kThreads) void k_codec(const float* __restrict__ x, void* __restrict__ out,
                                                    size_t n, const ScaleDiv sd, int ext, int vec_ok) {
    __shared__ uint32_t sT[16];
    lut_fill<FMT>(sT);
    __syncthreads();
    const size_t nvec = vec_ok ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const uint32_t u0 = __float_as_uint(div_const(v.x, sd)), u1 = __float_as_uint(div_const(v.y, sd));
        const uint32_t u2 = __float_as_uint(div_const(v.z, sd)), u3 = __float_as_uint(div_const(v.w, sd));
        if constexpr (MODE == 0) {
            float4 r;
            r.x = __uint_as_float(quant_bits<FMT>(u0, __float_as_uint(v.x), sT));
            r.y = __uint_as_float(quant_bits<FMT>(u1, __float_as_uint(v.y), sT));
            r.z = __uint_as_float(quant_bits<FMT>(u2, __float_as_uint(v.z), sT));
            r.w = __uint_as_float(quant_bits<FMT>(u3, __float_as_uint(v.w), sT));
            st_stream4<SLFP_NT_CODEC>(reinterpret_cast<float*>(out) + 4 * i, r);
        } else {
            const uint32_t c = quant_code<FMT>(u0, __float_as_uint(v.x), ext) | (quant_code<FMT>(u1, __float_as_uint(v.y), ext) << 8) |
                               (quant_code<FMT>(u2, __float_as_uint(v.z), ext) << 16) | (quant_code<FMT>(u3, __float_as_uint(v.w), ext) << 24);
            if constexpr (SLFP_NT_CODEC & 2) __builtin_nontemporal_store(c, reinterpret_cast<uint32_t*>(out) + i);
            else reinterpret_cast<uint32_t*>(out)[i] = c;
        }
    }
    // scalar tail (and the whole array when a pointer is not 16-byte aligned)
    for (size_t i = nvec * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const float xi = x[i];
        const uint32_t u = __float_as_uint(div_const(xi, sd));
        if constexpr (MODE == 0) {
            reinterpret_cast<float*>(out)[i] = __uint_as_float(quant_bits<FMT>(u, __float_as_uint(xi), sT));
        } else {
            reinterpret_cast<uint8_t*>(out)[i] = (uint8_t)quant_code<FMT>(u, __float_as_uint(xi), ext);
        }
    }
}

// y = Q(x/scale) float32 through the threshold table (slfp_enc.hpp): 7 VALU instructions per element instead
// of 22.  `+ 0.0f` turns the table form's -0 (x = -0, or a negative x whose quotient underflows) into the
// reference's +0 (torch.sign(-0) == 0) and changes nothing else.
__global__ __launch_bounds__(kThreads) void k_quantize_tab(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                           const EncArgs t, int vec_ok) {
    __shared__ __attribute__((aligned(16))) uint2 sE[kEncEntries + 1];
    enc_fill<kThreads>(sE, t);
    __syncthreads();
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(sE);
    const float r1 = t.r1, lo = t.lo, hi = t.hi;
    const size_t nvec = vec_ok ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
        const float4 v = ld_stream4<SLFP_NT_CODEC>(x + 4 * i);
        float4 r = enc4_f32(v, r1, lo, hi, tb);
        r.x += 0.0f; r.y += 0.0f; r.z += 0.0f; r.w += 0.0f;
        st_stream4<SLFP_NT_CODEC>(y + 4 * i, r);
    }
    for (size_t i = nvec * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const float xi = x[i];
        float r = enc_f32(xi, r1, lo, hi, tb) + 0.0f;
        y[i] = xi != xi ? __uint_as_float(kBitsQNaN) : r;
    }
}

template <int FMT>
__global__ __launch_bounds__(kThreads) void k_decode(const uint8_t* __restrict__ code, float* __restrict__ y,
                                                     size_t n, int ext, int vec_ok) {
    __shared__ uint32_t sT[16];
    lut_fill<kFmtW8>(sT);  // identity table: decode indexes it with the log code
    __syncthreads();
    const size_t nvec = vec_ok ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
        const uint32_t c = reinterpret_cast<const uint32_t*>(code)[i];
        float4 r;
        r.x = __uint_as_float(decode_bits<FMT>(c & 0xFFu, ext, sT));
        r.y = __uint_as_float(decode_bits<FMT>((c >> 8) & 0xFFu, ext, sT));
        r.z = __uint_as_float(decode_bits<FMT>((c >> 16) & 0xFFu, ext, sT));
        r.w = __uint_as_float(decode_bits<FMT>(c >> 24, ext, sT));
        reinterpret_cast<float4*>(y)[i] = r;
    }
    for (size_t i = nvec * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
        y[i] = __uint_as_float(decode_bits<FMT>(code[i], ext, sT));
}

// quantize_layerout: layerout_bits() lives in slfp_device.hpp (the conv epilogues can fuse it too)
__global__ __launch_bounds__(kThreads) void k_layerout(const float* __restrict__ x, float* __restrict__ y, size_t n, int vec_ok) {
    const size_t nvec = vec_ok ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 r;
        r.x = __uint_as_float(layerout_bits(__float_as_uint(v.x)));
        r.y = __uint_as_float(layerout_bits(__float_as_uint(v.y)));
        r.z = __uint_as_float(layerout_bits(__float_as_uint(v.z)));
        r.w = __uint_as_float(layerout_bits(__float_as_uint(v.w)));
        reinterpret_cast<float4*>(y)[i] = r;
    }
    for (size_t i = nvec * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
        y[i] = __uint_as_float(layerout_bits(__float_as_uint(x[i])));
}

// max |x| over a tensor (the per-layer calibration statistic of get_scale_factor,
// cifar100_train_eval.py:261-271): wave64 shuffle reduction, one atomicMax per wave on the
// float bits (non-negative floats order like unsigned integers).  *out must be zeroed first.
__global__ __launch_bounds__(kThreads) void k_absmax(const float* __restrict__ x, uint32_t* __restrict__ out, size_t n, int vec_ok) {
    const size_t nvec = vec_ok ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * kThreads;
    uint32_t m = 0;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = max(max(m, __float_as_uint(v.x) & 0x7FFFFFFFu), __float_as_uint(v.y) & 0x7FFFFFFFu);
        m = max(max(m, __float_as_uint(v.z) & 0x7FFFFFFFu), __float_as_uint(v.w) & 0x7FFFFFFFu);
    }
    for (size_t i = nvec * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
        m = max(m, __float_as_uint(x[i]) & 0x7FFFFFFFu);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

static int grid_for(size_t n) {
    size_t blocks = (n / 4 + kThreads - 1) / kThreads;
    if (blocks < 1) blocks = 1;
    const size_t cap = (size_t)device_cu_count() * 8;  // 8 blocks per CU, grid-stride the rest
    if (blocks > cap) blocks = cap;
    return (int)blocks;
}

template <int MODE>
static int launch_codec(const float* x, void* out, size_t n, float scale, int fmt, hipStream_t st) {
    const int f = fmt & kFmtMask, ext = (fmt & kFmtExt) ? 1 : 0;
    const int vec_ok = aligned16(x) && ((reinterpret_cast<uintptr_t>(out) & (MODE == 0 ? 15u : 3u)) == 0);
    const int g = grid_for(n);
    if (MODE == 0 && (f == kFmtAct8 || f == kFmtSfp7)) {
        const EncArgs* t = act_table(scale, f, kEncF32);
        if (t) {
            hipLaunchKernelGGL(k_quantize_tab, dim3(g), dim3(kThreads), 0, st, x, reinterpret_cast<float*>(out), n, *t, vec_ok);
            return check_launch("slfp codec kernel (threshold table)");
        }
    }
    const ScaleDiv sd = make_scale_div(scale);
    switch (f) {
        case kFmtAct8: hipLaunchKernelGGL((k_codec<kFmtAct8, MODE>), dim3(g), dim3(kThreads), 0, st, x, out, n, sd, ext, vec_ok); break;
        case kFmtW8: hipLaunchKernelGGL((k_codec<kFmtW8, MODE>), dim3(g), dim3(kThreads), 0, st, x, out, n, sd, ext, vec_ok); break;
        case kFmtSfp7: hipLaunchKernelGGL((k_codec<kFmtSfp7, MODE>), dim3(g), dim3(kThreads), 0, st, x, out, n, sd, ext, vec_ok); break;
        default: return fail(SLFP_ERR_BAD_ARG, "unknown codec format %d", fmt);
    }
    return check_launch("slfp codec kernel");
}

int launch_quantize(const float* x, float* y, size_t n, float scale, int fmt, hipStream_t st) {
    return launch_codec<0>(x, y, n, scale, fmt, st);
}

// ---- NCHW <-> NHWC: per image a [R][Cc] -> [Cc][R] transpose through a padded LDS tile ----
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ x, float* __restrict__ y,
                                                   int64_t rows, int64_t cols) {
    __shared__ float tile[32][33];
    const int64_t img = blockIdx.z;
    const float* xi = x + img * rows * cols;
    float* yi = y + img * rows * cols;
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols) tile[j][tx] = xi[r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) yi[c * rows + r] = tile[tx][j];
    }
}

// Exhaustive self-check of div_const against IEEE `/` over ALL 2^32 float32 patterns:
// out[0] = #x with |x| and |x/d| both within [1e-20, 1e20] (no residual underflow) whose quotient differs bitwise,
// out[1] = #x for which ANY of the three quantizers would return a different value.
__global__ __launch_bounds__(kThreads) void k_div_check(const ScaleDiv sd, unsigned long long* __restrict__ out) {
    __shared__ uint32_t sA[16], sW[16];
    lut_fill<kFmtAct8>(sA);
    lut_fill<kFmtW8>(sW);
    __syncthreads();
    unsigned long long bad_q = 0, bad_code = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const float ref = x / sd.d;
        const float got = div_const(x, sd);
        const uint32_t ur = __float_as_uint(ref), ug = __float_as_uint(got);
        const float mag = fabsf(ref);
        const float mx = fabsf(x);
        if (mag >= 1e-20f && mag <= 1e20f && mx >= 1e-20f && mx <= 1e20f && ur != ug) ++bad_q;
        const uint32_t ux = (uint32_t)i;
        if (quant_bits<kFmtAct8>(ur, ux, sA) != quant_bits<kFmtAct8>(ug, ux, sA)) ++bad_code;
        if (quant_bits<kFmtW8>(ur, ux, sW) != quant_bits<kFmtW8>(ug, ux, sW)) ++bad_code;
        if (quant_bits<kFmtSfp7>(ur, ux, sW) != quant_bits<kFmtSfp7>(ug, ux, sW)) ++bad_code;
        if (quant_bits<kFmtAct8, 4>(__float_as_uint(ref * 16.f), ux, sA) != quant_bits<kFmtAct8, 4>(__float_as_uint(got * 16.f), ux, sA)) ++bad_code;
    }
    if (bad_q) atomicAdd(out, bad_q);
    if (bad_code) atomicAdd(out + 1, bad_code);
}

static int launch_transpose(const float* x, float* y, int64_t n, int64_t rows, int64_t cols, hipStream_t st) {
    if (!x || !y || n <= 0 || rows <= 0 || cols <= 0) return fail(SLFP_ERR_BAD_ARG, "transpose: bad argument");
    if (n > 65535 || ceil_div(rows, 32) > 65535) return fail(SLFP_ERR_UNSUPPORTED, "transpose: grid too large");
    dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32), (unsigned)n);
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, st, x, y, rows, cols);
    return check_launch("slfp transpose kernel");
}

}  // namespace slfp

using namespace slfp;

extern "C" {

int slfp_version(void) { return SLFP_ABI_VERSION; }
void slfp_debug_reload_switches(void) { reload_switches(); }
const char* slfp_last_error(void) { return last_error_text(); }

int slfp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int slfp_encode_f32(const float* x, uint8_t* code, size_t n, float scale_div, int fmt, void* stream) {
    if (n == 0) return SLFP_OK;
    if (!x || !code) return fail(SLFP_ERR_BAD_ARG, "slfp_encode_f32: null pointer");
    if (!(scale_div > 0.f)) return fail(SLFP_ERR_BAD_ARG, "slfp_encode_f32: scale must be > 0");
    if (!scale_div_ok(scale_div)) return fail(SLFP_ERR_UNSUPPORTED, "slfp_encode_f32: scale must be within [1e-30, 1e30]");
    return launch_codec<1>(x, code, n, scale_div, fmt, as_stream(stream));
}

int slfp_quantize_f32(const float* x, float* y, size_t n, float scale_div, int fmt, void* stream) {
    if (n == 0) return SLFP_OK;
    if (!x || !y) return fail(SLFP_ERR_BAD_ARG, "slfp_quantize_f32: null pointer");
    if (!(scale_div > 0.f)) return fail(SLFP_ERR_BAD_ARG, "slfp_quantize_f32: scale must be > 0");
    if (!scale_div_ok(scale_div)) return fail(SLFP_ERR_UNSUPPORTED, "slfp_quantize_f32: scale must be within [1e-30, 1e30]");
    return launch_codec<0>(x, y, n, scale_div, fmt, as_stream(stream));
}

int slfp_decode_f32(const uint8_t* code, float* y, size_t n, int fmt, void* stream) {
    if (n == 0) return SLFP_OK;
    if (!code || !y) return fail(SLFP_ERR_BAD_ARG, "slfp_decode_f32: null pointer");
    const int f = fmt & kFmtMask, ext = (fmt & kFmtExt) ? 1 : 0;
    const int vec_ok = aligned16(y) && ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
    const int g = grid_for(n);
    hipStream_t st = as_stream(stream);
    if (f == kFmtSfp7) hipLaunchKernelGGL((k_decode<kFmtSfp7>), dim3(g), dim3(kThreads), 0, st, code, y, n, ext, vec_ok);
    else if (f == kFmtAct8 || f == kFmtW8) hipLaunchKernelGGL((k_decode<kFmtAct8>), dim3(g), dim3(kThreads), 0, st, code, y, n, ext, vec_ok);
    else return fail(SLFP_ERR_BAD_ARG, "unknown codec format %d", fmt);
    return check_launch("slfp decode kernel");
}

int slfp_quantize_layerout_f32(const float* x, float* y, size_t n, void* stream) {
    if (n == 0) return SLFP_OK;
    if (!x || !y) return fail(SLFP_ERR_BAD_ARG, "slfp_quantize_layerout_f32: null pointer");
    const int vec_ok = aligned16(x) && aligned16(y);
    hipLaunchKernelGGL(k_layerout, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(stream), x, y, n, vec_ok);
    return check_launch("slfp layerout kernel");
}

int slfp_absmax_f32(const float* x, size_t n, float* out, void* stream) {
    if (!out) return fail(SLFP_ERR_BAD_ARG, "slfp_absmax_f32: null pointer");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return check_launch("hipMemsetAsync(absmax)");
    if (n == 0) return SLFP_OK;
    if (!x) return fail(SLFP_ERR_BAD_ARG, "slfp_absmax_f32: null pointer");
    hipLaunchKernelGGL(k_absmax, dim3(grid_for(n)), dim3(kThreads), 0, st, x, reinterpret_cast<uint32_t*>(out), n, (int)aligned16(x));
    return check_launch("slfp absmax kernel");
}

int slfp_debug_div_mismatches(float scale_div, unsigned long long* out2, void* stream) {
    if (!out2 || !(scale_div > 0.f)) return fail(SLFP_ERR_BAD_ARG, "slfp_debug_div_mismatches: bad argument");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(out2, 0, 2 * sizeof(unsigned long long), st) != hipSuccess) return check_launch("hipMemsetAsync");
    if (!scale_div_ok(scale_div)) return fail(SLFP_ERR_UNSUPPORTED, "scale must be within [1e-30, 1e30]");
    ScaleDiv sd = make_scale_div(scale_div);
    hipLaunchKernelGGL(k_div_check, dim3(256 * 16), dim3(kThreads), 0, st, sd, out2);
    return check_launch("slfp division self-check kernel");
}

int slfp_nchw_to_nhwc_f32(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, void* stream) {
    return launch_transpose(x, y, n, c, h * w, as_stream(stream));  // [C][HW] -> [HW][C]
}

int slfp_nhwc_to_nchw_f32(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, void* stream) {
    return launch_transpose(x, y, n, h * w, c, as_stream(stream));  // [HW][C] -> [C][HW]
}

}  // extern "C"
