// VALU issue-rate microbenchmark (gfx950): cycles per wave64 VALU instruction per SIMD, by instruction kind and by
// waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {   // independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 1) {   // v_pk_fma_f32 on pairs
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, aa = {a, a}, bb = {b, b};
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(aa), "v"(bb));
                x0 = p0[0]; x1 = p0[1]; x2 = p1[0]; x3 = p1[1]; x4 = p2[0]; x5 = p2[1]; x6 = p3[0]; x7 = p3[1];
            } else if (KIND == 2) {   // integer / select mix: and, cmp+cndmask, med3
                asm volatile("v_med3_f32 %0, %0, %8, %9\n v_and_b32 %1, %1, %8\n v_cmp_ge_f32 vcc, %2, %9\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_med3_f32 %4, %4, %8, %9\n v_and_b32 %5, %5, %8\n v_cmp_ge_f32 vcc, %6, %9\n v_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            } else {   // SDWA ops
                asm volatile("v_lshlrev_b32_sdwa %0, 2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %1, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %2, 2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %3, 2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %4, 2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %5, 2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %6, 2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %7, 2, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 32 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "med3/and/cmp/cndmask", "sdwa shift"};
    for (int kind = 0; kind < 4; ++kind)
        for (int wgs_per_cu : {1, 2, 4, 8}) {   // 256-thread WG = 1 wave per SIMD
            dim3 g(256 * wgs_per_cu);
            auto launch = [&]() {
                if (kind == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                else if (kind == 2) hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<3>, g, dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_simd = (double)iters * 64 * wgs_per_cu;   // 64 instr per iter per wave, waves per SIMD = wgs_per_cu
            printf("%-22s waves/SIMD %d: %.3f ms -> %.2f ns per instr per SIMD (%.2f cycles at 2.4 GHz)\n", names[kind], wgs_per_cu, ms,
                   ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
        }
    return 0;
}
