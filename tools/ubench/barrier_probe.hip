// What does one "empty" K step of an 8-wave workgroup cost?  (FTC_OP_MBHEAD's K loop took 1.24 k cycles per step with its DMA, MFMAs and
// fragment prefetches ablated.)  Loop of N x { s_barrier ; R x ds_read_b128 + wait } with W waves per workgroup and one workgroup per CU,
// s_memtime ticks of wave 0 per iteration.
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/barrier_probe.hip -o tools/ubench/barrier_probe.bin && tools/ubench/barrier_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int R, bool BAR>
__global__ void k(unsigned* out, int iters, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    for (int i = t; i < 16384; i += blockDim.x) reinterpret_cast<u32x4*>(smem)[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    u32x4 acc = {0u, 0u, 0u, 0u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int off = (t & 63) * 16 + (t >> 6) * 4096;
    for (int it = 0; it < iters; ++it) {
        if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int r = 0; r < R; ++r) acc += *reinterpret_cast<const u32x4*>(smem + ((off + r * 1024 + it * 64) & 0x3fff0));
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + t] = acc[0] + acc[1] + acc[2] + acc[3];
    if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int R, bool BAR>
void run(int waves, size_t lds) {
    unsigned* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ticks, 8);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<R, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<R, BAR>), dim3(256), dim3(waves * 64), lds, 0, out, iters, ticks);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("waves %d  reads/iter %d  barrier %d  lds %zu KB : %.1f ticks per iteration\n", waves, R, (int)BAR, lds >> 10, (double)h / iters);
    hipFree(out); hipFree(ticks);
}

int main() {
    run<0, true>(8, 262144 / 2 + 16384);
    run<0, true>(4, 65536);
    run<7, true>(8, 262144 / 2 + 16384);
    run<7, false>(8, 262144 / 2 + 16384);
    run<7, true>(4, 65536);
    run<16, true>(8, 262144 / 2 + 16384);
    run<0, true>(16, 65536 + 16384);
    return 0;
}
