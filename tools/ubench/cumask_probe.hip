// Which compute units does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) run on?  2048 workgroups that each hold their CU for a
// while record (XCC id, SE id, CU id); the host prints, per mask, how many distinct CUs every XCC used.
//     hipcc --offload-arch=gfx950 -O2 tools/ubench/cumask_probe.hip -o tools/ubench/cumask_probe.bin && tools/ubench/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void probe(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}                // hold the CU: later workgroups must find another one
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char* name, const std::vector<int>& bits) {
    unsigned words[8] = {0};
    for (int b : bits) words[b >> 5] |= 1u << (b & 31);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, words) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int n = 2048;
    unsigned* d;
    hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 65536, s, d);          // 64 KB of LDS: at most two workgroups per CU
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::set<unsigned> per[8];
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per[xcc & 7].insert(se * 32 + sh * 16 + cu);
    }
    printf("%-44s (%3zu bits): CUs used per XCC:", name, bits.size());
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", per[x].size()); tot += (int)per[x].size(); }
    printf("  = %d\n", tot);
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    std::vector<int> all, half_mod, q3, lo128, x0;
    for (int i = 0; i < 256; ++i) {
        all.push_back(i);
        if (i % 8 < 4) half_mod.push_back(i);
        if (i < 192) q3.push_back(i);
        if (i < 128) lo128.push_back(i);
        if (i % 8 == 0) x0.push_back(i);
    }
    run("all 256 bits", all);
    run("bits with i % 8 < 4", half_mod);
    run("bits 0..191", q3);
    run("bits 0..127", lo128);
    run("bits with i % 8 == 0", x0);
    return 0;
}
