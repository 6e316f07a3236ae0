// Calibration micro-benchmark: back-to-back v_mfma_f32_32x32x16_bf16 on NACC independent accumulators, W waves per workgroup,
// WG/CU workgroups per CU, with ZERO and with RANDOM operand data.
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Why both data sets: the matrix pipe issues one 32x32x16 MFMA per 32 cycles per SIMD regardless of the data, so at the 2.4 GHz
// maximum clock the chip does 1024 SIMDs x 32768 FLOP / 32 cycles x 2.4 GHz = 2.52 PFLOP/s (the guide's 2495 TF).  What the data
// changes is the CLOCK the part sustains inside its power budget (MI355X_MICROARCH.md, "DVFS give-back": zero inputs 2.30 GHz,
// real data 1.90-1.95 GHz).  The round-1 version of this file only used non-zero data and its 1.74-1.9 PF was read as a "practical
// ceiling of 72 %"; it is the power-limited clock, not an issue-rate limit.  Roofline fractions in this repository are quoted
// against the guide's 2.5 PF; the effective clock printed here is for information.
// Each run also reports s_memtime ticks of one wave: 32.0 ticks per MFMA on its SIMD with zeros AND with random data while the tick
// RATE follows the clock (2.36 G/s vs 1.79 G/s) -- the tick is the shader cycle and the slowdown is frequency, not issue throttling.
// A single accumulator on a single wave per SIMD (NACC=1, 4 waves) shows the dependent-issue latency instead: 44 ticks per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC>
__global__ void k(const unsigned* __restrict__ seed, float* out, int iters, unsigned long long* ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 a, b;
    const unsigned s0 = seed[threadIdx.x & 63];
    for (int e = 0; e < 8; ++e) {
        // seed = 0 -> all operands exactly 0; else pseudo-random values in (-1, 1)
        const unsigned h = s0 * (2654435761u + 40503u * e + threadIdx.x);
        a[e] = (__bf16)(s0 ? (float)((int)(h >> 8) & 0xffff) / 32768.f - 1.f : 0.f);
        b[e] = (__bf16)(s0 ? (float)((int)(h >> 12) & 0xffff) / 32768.f - 1.f : 0.f);
    }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // s_memtime ticks this wave spent (read after the accumulators were consumed, i.e. after the last MFMA retired)
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = __builtin_amdgcn_s_memtime() - t0;
}

template <int NACC>
void run(int waves, int blocks_per_cu, bool zeros) {
    const int iters = 20000;
    const int blocks = 256 * blocks_per_cu;
    float* out;
    unsigned* seed;
    unsigned long long* ticks;
    hipMalloc(&ticks, 8);
    hipMalloc(&out, sizeof(float) * blocks * waves * 64);
    hipMalloc(&seed, 64 * sizeof(unsigned));
    std::vector<unsigned> hs(64);
    for (int i = 0; i < 64; ++i) hs[i] = zeros ? 0u : (unsigned)(rand() | 1);
    hipMemcpy(seed, hs.data(), 64 * sizeof(unsigned), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(waves * 64), 0, 0, seed, out, iters, ticks);   // warm-up / clock ramp
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(waves * 64), 0, 0, seed, out, iters, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double flop = 2.0 * 32 * 32 * 16 * (double)NACC * iters * waves * blocks;
    const double tf = flop / ms / 1e9;
    // at full issue rate the chip retires 1024 SIMDs * 32768 FLOP / 32 cycles = 1,048,576 FLOP per clock
    unsigned long long ht = 0;
    hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
    // MFMAs that went through the SIMD of workgroup 0 / wave 0 while it ran: NACC * iters per wave, waves_per_simd waves share the pipe
    const double waves_per_simd = (double)waves * blocks_per_cu / 4.0;
    const double per_mfma = (double)ht / ((double)NACC * iters * (waves_per_simd < 1 ? 1 : waves_per_simd));
    printf("%-6s NACC=%2d waves/WG=%2d WG/CU=%d : %8.3f ms  %6.0f TFLOP/s  = %4.1f %% of 2.5 PF, effective clock if issue-bound %.2f GHz | s_memtime: %.1f ticks per MFMA on the SIMD, %.2f G ticks/s\n",
           zeros ? "zeros" : "random", NACC, waves, blocks_per_cu, ms, tf, 100.0 * tf / 2500.0, tf * 1e12 / 1048576.0 / 1e9, per_mfma, (double)ht / (ms * 1e-3) / 1e9);
    hipFree(ticks);
    hipFree(out); hipFree(seed);
}

int main() {
    for (int z = 1; z >= 0; --z) {
        run<4>(4, 1, z); run<4>(8, 1, z); run<4>(12, 1, z); run<4>(4, 2, z); run<2>(8, 1, z); run<1>(8, 1, z); run<1>(4, 1, z);
    }
    return 0;
}
