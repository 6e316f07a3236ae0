// Calibration micro-benchmark: back-to-back v_mfma_f32_32x32x16_bf16 on NACC independent accumulators,
// W waves per workgroup, one workgroup per CU.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC>
__global__ void k(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves, int blocks_per_cu) {
    const int iters = 4000;
    const int blocks = 256 * blocks_per_cu;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * waves * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(waves * 64), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(waves * 64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 16 * (double)NACC * iters * waves * blocks;
    printf("NACC=%d waves/WG=%d WG/CU=%d : %.3f ms  %.0f TFLOP/s\n", NACC, waves, blocks_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    run<6>(4, 1); run<6>(8, 1); run<6>(4, 2); run<6>(16, 1); run<2>(8, 1); run<1>(8, 1); run<12>(8, 1);
    return 0;
}
