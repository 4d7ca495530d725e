// L2 -> CU streaming microbenchmark (gfx950): what rate can the workgroups of a pointwise-GEMM-shaped kernel pull a SHARED
// weight blob (L2-resident, every workgroup reads all of it, 1 KiB fragment per wave-load) through the vector memory path?
// hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream && ./l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// NW waves per workgroup; each wave owns 1/NW of every stage; a stage = NW * DEPTH fragments of 1 KiB; DEPTH loads are in
// flight per wave before the wave consumes them (xor) -- the pattern of k_pw_tiled's W fragments
template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void k_stream(const u32x4* __restrict__ w, int frags_total, int reps, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    const int stages = frags_total / (NW * DEPTH);
    for (int r = 0; r < reps; ++r) {
        for (int s = 0; s < stages; ++s) {
            u32x4 v[DEPTH];
            const u32x4* p = w + ((size_t)(s * NW + wave) * DEPTH) * 64 + lane;
#pragma unroll
            for (int f = 0; f < DEPTH; ++f) v[f] = p[f * 64];
#pragma unroll
            for (int f = 0; f < DEPTH; ++f) acc ^= v[f];
        }
        asm volatile("" : "+v"(acc));
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) out[blockIdx.x] = acc.x;
}

template <int NW, int DEPTH>
static void run(const u32x4* d, int bytes, int wgs_per_cu, unsigned* out, int cus) {
    const int frags = bytes / 1024, reps = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g(cus * wgs_per_cu);
    hipLaunchKernelGGL((k_stream<NW, DEPTH>), g, dim3(NW * 64), 0, 0, d, frags, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_stream<NW, DEPTH>), g, dim3(NW * 64), 0, 0, d, frags, reps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tot = (double)bytes * reps * g.x;
    printf("blob %5d KiB  waves/WG %2d  loads in flight/wave %2d  WGs/CU %d: %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s aggregate\n",
           bytes / 1024, NW, DEPTH, wgs_per_cu, ms, tot / cus / (ms * 1e6), tot / (ms * 1e9));
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    u32x4* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
    unsigned* out; hipMalloc(&out, 1 << 20);
    for (int bytes : {128 << 10, 512 << 10, 2 << 20, 16 << 20}) {
        for (int wg : {1, 2}) {
            run<8, 4>(d, bytes, wg, out, cus);
            run<8, 8>(d, bytes, wg, out, cus);
            run<8, 16>(d, bytes, wg, out, cus);
        }
        run<16, 8>(d, bytes, 1, out, cus);
        run<4, 16>(d, bytes, 4, out, cus);
    }
    return 0;
}
