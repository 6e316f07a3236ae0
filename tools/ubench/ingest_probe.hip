// L2 -> LDS ingest probe (round-5 verdict item 2): how many bytes per clock does a CU take through `buffer_load ... lds` when the data are
// L2-resident, and does the rate depend on the ACCESS PATTERN the project-GEMM kernels use (128-byte row pieces at a row pitch of 2 K bytes)?
//
// The kernels under question (csrc/conv1x1_px144.hip, csrc/mbconv_slice.hip) stream their operands as "stages": TN weight rows + 144 pixel
// rows x 64 K values = 208 rows x 128 B = 26 KB per stage and workgroup, one workgroup per CU, the rows of a stage 2 K bytes apart in memory
// (K = 1536 / 3072 / 3840 -> pitch 3072 / 6144 / 7680 B), all 32 workgroups of an XCD walking K in step.  DESIGN.md appendix A7 measured
// ~30 B/clk/CU for that loop and called it "the ingest cap"; MI355X_MICROARCH.md puts L2 at ~34.5 TB/s = 65-75 B/clk/CU.
//
// Patterns (one kernel name each, so rocprofv3 --pmc separates them):
//   slab    : every workgroup streams ONE contiguous private slab (stage after stage, wrapping inside `slab_bytes`) -- the best case
//   pitch   : the GEMM pattern: operand A [NA rows][K], operand B [NB rows][K] row-major 16-bit; workgroup (nt, mt) reads rows
//             [64 nt, +64) of A and [144 mt, +144) of B, 128 bytes of each row per K step; XCD-aware tile order as in the kernel
//   kblock  : the same tiles from K-blocked operands [K/64][rows][64]: a stage is two contiguous runs (8 KB + 18 KB)
// Each with 4 or 8 issuing waves and a ring of NST stages (NST - 1 in flight).  No MFMA, no LDS reads: the number is the ingest
// rate alone.  Output: cycles per stage (s_memtime of wave 0, median over workgroups), B/clk/CU, and GB/s over the launch by HIP events.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ingest_probe tools/ubench/ingest_probe.hip
// Run:   tools/ubench/ingest_probe            (prints a table; profiles/r06_ingest_probe.txt is its output on MI355X)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                         \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)

using lds_void = __attribute__((address_space(3))) void;

struct Args {
    const unsigned char* a;     // operand A (or the slab)
    const unsigned char* b;     // operand B
    unsigned a_bytes, b_bytes;
    int pitch;                  // bytes between rows (pitch pattern) ; kblock: rows_a * 128 / rows_b * 128 are the K-block strides
    int rows_a_total, rows_b_total;
    int nN;                     // channel tiles per pixel tile (tile order: channel tiles fastest)
    int nblk;
    int ksteps;                 // stages per pass
    int passes;
    unsigned slab_bytes;        // slab pattern: private bytes per workgroup
    unsigned long long* cyc;    // [nblk] cycles of the timed loop (wave 0)
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int ROWS_A = 64, ROWS_B = 144, ROWS = ROWS_A + ROWS_B, STAGE = ROWS * 128, PIECES = STAGE / 1024;     // 26 pieces of 1 KiB

// PATTERN 0 slab, 1 pitch, 2 kblock.  NW issuing waves, NST-stage ring.
template <int PATTERN, int NW, int NST>
__device__ __forceinline__ void ingest_body(const Args& p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int bid = blockIdx.x;
    if ((p.nblk & 7) == 0) bid = (bid & 7) * (p.nblk >> 3) + (bid >> 3);        // XCD x works on the contiguous tile range x
    const int nt = bid % p.nN, mt = bid / p.nN;
    constexpr int NPW = (PIECES + NW - 1) / NW;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.b), 0, p.b_bytes, 0x00020000);
    // piece i = rows [8 i, 8 i + 8) of the stage, a lane = 16 bytes: row 8 i + lane / 8, chunk lane % 8
    int voff[NPW];
    bool isb[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int i = wave + NW * j;
        const int r = i * 8 + (lane >> 3), c = lane & 7;
        isb[j] = r >= ROWS_A;
        if (PATTERN == 0) voff[j] = bid * (int)p.slab_bytes + i * 1024 + lane * 16;
        else if (PATTERN == 1) voff[j] = (r < ROWS_A ? (nt * ROWS_A + r) : (mt * ROWS_B + r - ROWS_A)) * p.pitch + c * 16;
        else voff[j] = (r < ROWS_A ? (nt * ROWS_A + r) : (mt * ROWS_B + r - ROWS_A)) * 128 + c * 16;
    }
    const int kstride_a = PATTERN == 1 ? 128 : PATTERN == 2 ? p.rows_a_total * 128 : STAGE;
    const int kstride_b = PATTERN == 1 ? 128 : PATTERN == 2 ? p.rows_b_total * 128 : STAGE;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int issued = 0;
    for (int pass = 0; pass < p.passes; ++pass) {
        for (int k = 0; k < p.ksteps; ++k) {
            unsigned char* st = smem + (issued % NST) * STAGE;
            int soff_a, soff_b;
            if (PATTERN == 0) {
                soff_a = soff_b = (int)(((unsigned)k * STAGE) % p.slab_bytes);
            } else {
                soff_a = k * kstride_a;
                soff_b = k * kstride_b;
            }
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int i = wave + NW * j;
                if (i < PIECES) {
                    if (PATTERN == 0 || !isb[j])
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(st + i * 1024), 16, voff[j], soff_a, 0, 0);
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(st + i * 1024), 16, voff[j], soff_b, 0, 0);
                }
            }
            ++issued;
            // keep NST - 1 stages in flight: wait for the oldest one (each wave issued <= NPW pieces per stage; the count is per wave)
            if (issued >= NST - 1) {
                if (NST == 2) wait_vmcnt<0>();
                else if (NST == 3) wait_vmcnt<NPW>();
                else if (NST == 4) wait_vmcnt<2 * NPW>();
                else wait_vmcnt<3 * NPW>();
                __syncthreads();        // the consumer's barrier: a stage is used only when every wave's pieces have landed
            }
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (t == 0) p.cyc[blockIdx.x] = t1 - t0;
    if (p.passes < 0) p.cyc[0] = smem[t];            // (keeps the LDS writes observable)
}


// The same slab stream by plain `buffer_load_dwordx4` into registers (no LDS): is ~36 B/clk the LDS-DMA path's ceiling or the CU's?
// NW waves, each thread keeps DEPTH x 7 16-byte loads in flight (a "stage" = the same 26 pieces), values folded into one word.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int NW, int NST>
__device__ __forceinline__ void ingest_regs_body(const Args& p) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int NPW = (PIECES + NW - 1) / NW;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.a), 0, p.a_bytes, 0x00020000);
    int voff[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) voff[j] = blockIdx.x * (int)p.slab_bytes + (wave + NW * j) * 1024 + lane * 16;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    u32x4_t buf[NST - 1][NPW];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int total = p.ksteps * p.passes;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
        const int soff = (int)(((unsigned)(s % p.ksteps) * STAGE) % p.slab_bytes);
#pragma unroll
        for (int j = 0; j < NPW; ++j)
            if (wave + NW * j < PIECES) buf[s][j] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff[j], soff, 0);
    }
    for (int k0 = 0; k0 < total; k0 += NST - 1) {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) {
            const int k = k0 + s;
#pragma unroll
            for (int j = 0; j < NPW; ++j)
                if (wave + NW * j < PIECES) acc ^= buf[s][j];                   // consume stage k (waits for exactly its loads)
            const int kn = k + NST - 1;
            if (kn < total) {
                const int soff = (int)(((unsigned)(kn % p.ksteps) * STAGE) % p.slab_bytes);
#pragma unroll
                for (int j = 0; j < NPW; ++j)
                    if (wave + NW * j < PIECES) buf[s][j] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff[j], soff, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (t == 0) p.cyc[blockIdx.x] = t1 - t0;
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) p.cyc[1] = 1;       // (keeps the loads observable)
}

template <int PATTERN, int NW, int NST> struct Kern;
#define FTC_INGEST_KERNEL(PATTERN, NW, NST)                                                                    \
    __global__ __launch_bounds__(NW * 64, 1) void ingest_p##PATTERN##_w##NW##_s##NST(const Args p) { ingest_body<PATTERN, NW, NST>(p); } \
    template <> struct Kern<PATTERN, NW, NST> { static constexpr auto fn = ingest_p##PATTERN##_w##NW##_s##NST; };
FTC_INGEST_KERNEL(0, 4, 4) FTC_INGEST_KERNEL(0, 8, 4) FTC_INGEST_KERNEL(0, 4, 5)
FTC_INGEST_KERNEL(1, 4, 4) FTC_INGEST_KERNEL(1, 8, 4) FTC_INGEST_KERNEL(1, 4, 5)
FTC_INGEST_KERNEL(2, 4, 4) FTC_INGEST_KERNEL(2, 8, 4) FTC_INGEST_KERNEL(2, 4, 5)
#define FTC_INGEST_REGS(NW, NST)                                                                                 \
    __global__ __launch_bounds__(NW * 64, 1) void ingest_regs_w##NW##_s##NST(const Args p) { ingest_regs_body<NW, NST>(p); } \
    template <> struct Kern<3, NW, NST> { static constexpr auto fn = ingest_regs_w##NW##_s##NST; };
FTC_INGEST_REGS(4, 4) FTC_INGEST_REGS(8, 4) FTC_INGEST_REGS(16, 3)

template <int PATTERN, int NW, int NST>
static void run(const char* name, Args p, int reps, double ghz_hint) {
    const int lds = NST * STAGE;
    CHECK(hipFuncSetAttribute((const void*)Kern<PATTERN, NW, NST>::fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) Kern<PATTERN, NW, NST>::fn<<<p.nblk, NW * 64, lds>>>(p);       // warm L2 / MALL
    CHECK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    std::vector<unsigned long long> cyc(p.nblk), med;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0));
        Kern<PATTERN, NW, NST>::fn<<<p.nblk, NW * 64, lds>>>(p);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[i], e0, e1));
        CHECK(hipMemcpy(cyc.data(), p.cyc, sizeof(unsigned long long) * p.nblk, hipMemcpyDeviceToHost));
        std::sort(cyc.begin(), cyc.end());
        med.push_back(cyc[p.nblk / 2]);
    }
    std::sort(ms.begin(), ms.end());
    std::sort(med.begin(), med.end());
    const double stages = (double)p.ksteps * p.passes;
    const double bytes_wg = stages * STAGE;
    const double c = (double)med[reps / 2];
    // __builtin_readcyclecounter = s_memtime: a constant 100 MHz counter on gfx9 -> convert with the launch's wall time instead:
    // report bytes / wall time per CU and, with the clock hint, B/clk
    const double us = ms[reps / 2] * 1e3;
    const double gbs = bytes_wg * p.nblk / (us * 1e-6) / 1e9;
    std::printf("%-34s %2d waves %d-stage ring  %8.1f us  %7.1f GB/s  %6.1f B/clk/CU @%.2f GHz | %8.1f ticks/stage = %5.1f B/tick/CU\n", name, NW, NST, us, gbs,
                gbs * 1e9 / 256.0 / (ghz_hint * 1e9), ghz_hint, c / stages, (double)STAGE / (c / stages));
}

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? std::atof(argv[1]) : 2.0;       // shader clock to convert GB/s into B/clk (rocprofv3: GRBM_GUI_ACTIVE / wall time)
    const int nblk = 256, nN = 8;                                  // 4608 x 512 outputs on 64 x 144 tiles: 8 channel tiles x 32 pixel tiles
    const int rows_a = 512, rows_b = 4608;
    unsigned long long* cyc;
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * nblk));
    std::printf("# L2 -> LDS ingest by buffer_load..lds, 256 workgroups (one per CU), stage = 208 rows x 128 B = 26,624 B\n");
    for (int K : {1536, 3072, 3840}) {
        const int pitch = K * 2, ksteps = K / 64;
        const size_t a_bytes = (size_t)rows_a * pitch, b_bytes = (size_t)rows_b * pitch;
        unsigned char *a, *b;
        CHECK(hipMalloc(&a, a_bytes));
        CHECK(hipMalloc(&b, b_bytes));
        CHECK(hipMemset(a, 1, a_bytes));
        CHECK(hipMemset(b, 2, b_bytes));
        Args p{};
        p.a = a; p.b = b; p.a_bytes = (unsigned)a_bytes; p.b_bytes = (unsigned)b_bytes; p.pitch = pitch; p.rows_a_total = rows_a; p.rows_b_total = rows_b;
        p.nN = nN; p.nblk = nblk; p.ksteps = ksteps; p.passes = 4; p.cyc = cyc;
        char nm[96];
        std::snprintf(nm, sizeof nm, "pitch  K=%d (row pitch %d B)", K, pitch);
        run<1, 4, 4>(nm, p, 9, ghz);
        run<1, 8, 4>(nm, p, 9, ghz);
        run<1, 4, 5>(nm, p, 9, ghz);
        std::snprintf(nm, sizeof nm, "kblock K=%d ([K/64][rows][64])", K);
        run<2, 4, 4>(nm, p, 9, ghz);
        run<2, 8, 4>(nm, p, 9, ghz);
        run<2, 4, 5>(nm, p, 9, ghz);
        CHECK(hipFree(a));
        CHECK(hipFree(b));
    }
    {   // contiguous private slabs: 96 KB per workgroup (32 per XCD = 3 MB: inside the XCD's 4 MiB L2) and 1 MB per workgroup (256 MB: Infinity Cache / HBM)
        for (unsigned slab : {96u * 1024u, 1024u * 1024u}) {
            const size_t bytes = (size_t)slab * nblk + STAGE;
            unsigned char* a;
            CHECK(hipMalloc(&a, bytes));
            CHECK(hipMemset(a, 3, bytes));
            Args p{};
            p.a = a; p.b = a; p.a_bytes = p.b_bytes = (unsigned)bytes; p.nN = nN; p.nblk = nblk; p.ksteps = 48; p.passes = 4; p.slab_bytes = slab; p.cyc = cyc;
            char nm[96];
            std::snprintf(nm, sizeof nm, "slab   %u KB private per WG", slab / 1024);
            run<0, 4, 4>(nm, p, 9, ghz);
            run<0, 8, 4>(nm, p, 9, ghz);
            run<0, 4, 5>(nm, p, 9, ghz);
            std::snprintf(nm, sizeof nm, "slab   %u KB, loads to REGISTERS", slab / 1024);
            run<3, 4, 4>(nm, p, 9, ghz);
            run<3, 8, 4>(nm, p, 9, ghz);
            run<3, 16, 3>(nm, p, 9, ghz);
            CHECK(hipFree(a));
        }
    }
    CHECK(hipFree(cyc));
    return 0;
}
