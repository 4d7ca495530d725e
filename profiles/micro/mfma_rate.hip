// MFMA issue-rate microbenchmark (gfx950): ns and cycles per matrix instruction per SIMD, independent accumulators,
// by instruction shape and by waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// inline asm on "+a" accumulators: the builtin form let hipcc shuffle the accumulators between iterations (40 v_accvgpr_mov
// per 8 MFMAs in the first version of this file, which made the 16x16x32 shape look half as fast as it is)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    floatx16 d0, d1, d2, d3;
    for (int j = 0; j < 16; ++j) { d0[j] = 0; d1[j] = 0; d2[j] = 0; d3[j] = 0; }
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    float af = (float)a[0], bf = (float)b[1];
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n"
                         "v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n"
                         "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x32_f16 %7, %8, %9, %7"
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7) : "v"(a), "v"(b));
        } else if (KIND == 1) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3"
                         : "+a"(d0), "+a"(d1), "+a"(d2), "+a"(d3) : "v"(a), "v"(b));
        } else if (KIND == 2) {
            asm volatile("v_mfma_f32_16x16x16_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x16_f16 %1, %8, %9, %1\n v_mfma_f32_16x16x16_f16 %2, %8, %9, %2\n"
                         "v_mfma_f32_16x16x16_f16 %3, %8, %9, %3\n v_mfma_f32_16x16x16_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x16_f16 %5, %8, %9, %5\n"
                         "v_mfma_f32_16x16x16_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x16_f16 %7, %8, %9, %7"
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7) : "v"(a4), "v"(b4));
        } else {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n"
                         "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_mfma_f32_16x16x4_f32 %4, %8, %9, %4\n v_mfma_f32_16x16x4_f32 %5, %8, %9, %5\n"
                         "v_mfma_f32_16x16x4_f32 %6, %8, %9, %6\n v_mfma_f32_16x16x4_f32 %7, %8, %9, %7"
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7) : "v"(af), "v"(bf));
        }
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
    for (int j = 0; j < 16; ++j) s += d0[j] + d1[j] + d2[j] + d3[j];
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
