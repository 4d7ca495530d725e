// Microbenchmark (not product code): HBM read / write / copy rates on gfx950 for the access shapes the conv kernels use.
//   hipcc --offload-arch=gfx950 -O3 -o bw_patterns bw_patterns.hip && ./bw_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: read only (sum), 1: write only, 2: copy.  Each lane moves 16 bytes per access; a wave's access covers
// `seg` contiguous bytes per row and rows are `row_stride` bytes apart (seg == row_stride: fully linear).
template <int MODE>
__global__ __launch_bounds__(256) void k_bw(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16, int unroll_dummy, float* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n16; i += 4 * stride) {
        float4 a, b, c, d;
        if (MODE != 1) { a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride]; }
        else { a = b = c = d = make_float4(1.f, 2.f, 3.f, (float)i); }
        if (MODE == 0) { acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y; acc.z += a.z + b.z; acc.w += c.w + d.w; }
        else { dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d; }
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = acc.x;
}

// One contiguous chunk of U x 4 KiB per 256-thread workgroup, no loop (the shape of torch's elementwise kernels).
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_chunk(const float4* __restrict__ src, float4* __restrict__ dst, float* sink) {
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = MODE != 1 ? src[base + u * 256] : make_float4(1.f, 2.f, 3.f, (float)u);
    if (MODE == 0) {
        float s = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
        if (s == 1.2345f) *sink = s;
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) dst[base + u * 256] = v[u];
    }
}

// Row-segment pattern: tensor [rows][row_bytes]; a wave instruction reads `seg` bytes of each of 1024/seg consecutive rows
// (the pointwise X fragment load: seg = 64; the depthwise pixel load: seg = 128); every lane then walks along its row.
template <int MODE>
__global__ __launch_bounds__(256) void k_rows(const char* __restrict__ src, char* __restrict__ dst, size_t rows, int row_bytes, int seg, float* sink) {
    const int lane = threadIdx.x & 63;
    const int lanes_per_row = seg / 16, rows_per_inst = 64 / lanes_per_row;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t r0 = wave * rows_per_inst; r0 + rows_per_inst <= rows; r0 += nwaves * rows_per_inst) {
        const size_t row = r0 + lane / lanes_per_row;
        const size_t base = row * row_bytes + (size_t)(lane % lanes_per_row) * 16;
        for (int k = 0; k < row_bytes; k += 4 * seg) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (MODE != 1 && k + u * seg < row_bytes) ? *reinterpret_cast<const float4*>(src + base + k + u * seg) : make_float4(1, 2, 3, 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 0) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                else if (k + u * seg < row_bytes) *reinterpret_cast<float4*>(dst + base + k + u * seg) = v[u];
            }
        }
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = acc.x;
}

// The pointwise-conv traffic shape without any arithmetic: a wave's unit is 16 consecutive rows; it reads K floats of every row in
// 64-byte pieces per instruction (lane = (row, 16-byte quarter), as an MFMA B-fragment load) and writes N floats per row in WSEG-byte
// pieces.  PERSIST: 8-wave workgroups loop over the units interleaved (unit = round * waves + wave id); otherwise one unit per wave.
template <int WSEG, bool PERSIST>
__global__ __launch_bounds__(512) void k_pwpat(const char* __restrict__ src, char* __restrict__ dst, size_t units, int K, int N, float* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 512 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 512) >> 6;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t g = wave; g < units; g += nwaves) {
        const char* xr = src + (g * 16 + (lane & 15)) * (size_t)K * 4 + (lane >> 4) * 16;
        for (int k = 0; k < K * 4; k += 64) {
            const float4 v = *reinterpret_cast<const float4*>(xr + k);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (WSEG == 64) {
            char* yr = dst + (g * 16 + (lane & 15)) * (size_t)N * 4 + (lane >> 4) * 16;
            for (int n = 0; n < N * 4; n += 64) *reinterpret_cast<float4*>(yr + n) = acc;
        } else {   // 128-byte pieces: lane = (row of 8, 16-byte eighth), two instructions cover 16 rows
            for (int n = 0; n < N * 4; n += 128)
                for (int h = 0; h < 2; ++h)
                    *reinterpret_cast<float4*>(dst + (g * 16 + (lane >> 3) + 8 * h) * (size_t)N * 4 + n + (lane & 7) * 16) = acc;
        }
        if (!PERSIST) break;
    }
    if (acc.x == 1.2345f) *sink = acc.x;
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB per buffer: far beyond the 256 MiB Infinity Cache
    char *a, *b; float* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, double moved, const char* name) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, moved / best / 1e6);
    };
    for (int blocks : {2048, 8192}) {
        printf("grid %d x 256\n", blocks);
        time([&] { hipLaunchKernelGGL(k_bw<0>, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16, 0, sink); }, (double)bytes, "linear read");
        time([&] { hipLaunchKernelGGL(k_bw<1>, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16, 0, sink); }, (double)bytes, "linear write");
        time([&] { hipLaunchKernelGGL(k_bw<2>, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16, 0, sink); }, 2.0 * bytes, "linear copy");
    }
    for (int row_bytes : {512, 1024, 2048, 4096}) for (int seg : {64, 128}) {
        char name[96];
        const size_t rows = bytes / row_bytes;
        snprintf(name, sizeof name, "rows of %4d B, %3d-B segments: read", row_bytes, seg);
        time([&] { hipLaunchKernelGGL(k_rows<0>, dim3(4096), dim3(256), 0, 0, a, b, rows, row_bytes, seg, sink); }, (double)bytes, name);
        snprintf(name, sizeof name, "rows of %4d B, %3d-B segments: copy", row_bytes, seg);
        time([&] { hipLaunchKernelGGL(k_rows<2>, dim3(4096), dim3(256), 0, 0, a, b, rows, row_bytes, seg, sink); }, 2.0 * bytes, name);
    }
    {
        const size_t n16 = bytes / 16;
        time([&] { hipLaunchKernelGGL((k_chunk<0, 4>), dim3(n16 / 1024), dim3(256), 0, 0, (const float4*)a, (float4*)b, sink); }, (double)bytes, "chunk 16 KiB per workgroup: read");
        time([&] { hipLaunchKernelGGL((k_chunk<1, 4>), dim3(n16 / 1024), dim3(256), 0, 0, (const float4*)a, (float4*)b, sink); }, (double)bytes, "chunk 16 KiB per workgroup: write");
        time([&] { hipLaunchKernelGGL((k_chunk<2, 4>), dim3(n16 / 1024), dim3(256), 0, 0, (const float4*)a, (float4*)b, sink); }, 2.0 * bytes, "chunk 16 KiB per workgroup: copy");
        time([&] { hipLaunchKernelGGL((k_chunk<0, 8>), dim3(n16 / 2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, sink); }, (double)bytes, "chunk 32 KiB per workgroup: read");
        time([&] { hipLaunchKernelGGL((k_chunk<1, 8>), dim3(n16 / 2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, sink); }, (double)bytes, "chunk 32 KiB per workgroup: write");
        time([&] { hipLaunchKernelGGL((k_chunk<2, 8>), dim3(n16 / 2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, sink); }, 2.0 * bytes, "chunk 32 KiB per workgroup: copy");
    }
    {   // pointwise traffic shapes of MobileNetV1 (K -> N, pixels), batch 256
        struct { int K, N; size_t M; } L[] = {{32, 64, 256ull * 12544}, {64, 128, 256ull * 3136}, {128, 128, 256ull * 3136}, {128, 256, 256ull * 784},
                                              {256, 256, 256ull * 784}, {512, 512, 256ull * 196}};
        for (auto& l : L) {
            const size_t units = l.M / 16;
            const double moved = (double)l.M * (l.K + l.N) * 4;
            char name[96];
            snprintf(name, sizeof name, "pw %4d->%4d persistent 512 wg, 64-B stores", l.K, l.N);
            time([&] { hipLaunchKernelGGL((k_pwpat<64, true>), dim3(512), dim3(512), 0, 0, a, b, units, l.K, l.N, sink); }, moved, name);
            snprintf(name, sizeof name, "pw %4d->%4d persistent 512 wg, 128-B stores", l.K, l.N);
            time([&] { hipLaunchKernelGGL((k_pwpat<128, true>), dim3(512), dim3(512), 0, 0, a, b, units, l.K, l.N, sink); }, moved, name);
            snprintf(name, sizeof name, "pw %4d->%4d one unit per wave, 64-B stores", l.K, l.N);
            time([&] { hipLaunchKernelGGL((k_pwpat<64, false>), dim3((unsigned)(units / 8)), dim3(512), 0, 0, a, b, units, l.K, l.N, sink); }, moved, name);
            snprintf(name, sizeof name, "pw %4d->%4d one unit per wave, 128-B stores", l.K, l.N);
            time([&] { hipLaunchKernelGGL((k_pwpat<128, false>), dim3((unsigned)(units / 8)), dim3(512), 0, 0, a, b, units, l.K, l.N, sink); }, moved, name);
        }
    }
    // persistent-grid sensitivity: the same row pattern (1024-B rows, 64-B segments) with fewer, fatter workgroups
    for (int blocks : {256, 512, 1024, 2048, 4096, 16384}) {
        char name[96];
        const int row_bytes = 1024, seg = 64;
        const size_t rows = bytes / row_bytes;
        snprintf(name, sizeof name, "rows 1024 B / 64-B seg, %5d blocks: read", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<0>, dim3(blocks), dim3(256), 0, 0, a, b, rows, row_bytes, seg, sink); }, (double)bytes, name);
        snprintf(name, sizeof name, "rows 1024 B / 64-B seg, %5d blocks: copy", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<2>, dim3(blocks), dim3(256), 0, 0, a, b, rows, row_bytes, seg, sink); }, 2.0 * bytes, name);
    }
    return 0;
}
