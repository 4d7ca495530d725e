// MFMA issue-rate microbenchmark (gfx950): ns and cycles per matrix instruction per SIMD, independent accumulators,
// by instruction shape and by waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    floatx4 c[8];
    floatx16 d[4];
    for (int i = 0; i < 8; ++i) c[i] = floatx4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) d[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u], 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d[u], 0, 0, 0);
        } else if (KIND == 2) {
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[u], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], c[u], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += d[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 256 * 16 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[4] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x4_f32"};
    const double flop[4] = {16. * 16 * 32 * 2, 32. * 32 * 16 * 2, 16. * 16 * 16 * 2, 16. * 16 * 4 * 2};
    const int per_iter[4] = {8, 4, 8, 8};
    for (int kind = 0; kind < 4; ++kind)
        for (int wgs_per_cu : {1, 2, 4}) {   // a 256-thread workgroup = 1 wave per SIMD
            dim3 g(256 * wgs_per_cu);
            auto launch = [&]() {
                if (kind == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, d, iters);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, d, iters);
                else if (kind == 2) hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, d, iters);
                else hipLaunchKernelGGL(k<3>, g, dim3(256), 0, 0, d, iters);
            };
            launch(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double n_per_simd = (double)iters * per_iter[kind] * wgs_per_cu;
            const double ns = ms * 1e6 / n_per_simd;
            printf("%-26s waves/SIMD %d: %7.3f ms  %6.2f ns per instr per SIMD  -> %7.1f TFLOP/s on 1024 SIMDs\n", names[kind], wgs_per_cu, ms, ns,
                   flop[kind] / ns * 1024 / 1e3);
        }
    return 0;
}
