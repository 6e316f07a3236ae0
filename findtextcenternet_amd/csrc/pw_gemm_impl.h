// Pointwise (1x1, stride 1) convolution of the MBConv blocks as a plain GEMM with big tiles, a deep DMA ring and an optional K split.
//
// Why a second kernel next to conv_igemm_glds_kernel: the MBConv expand / project GEMMs at batch 8 are SMALL (14.5 GFLOP, 2.4-28 M
// outputs) on a 256-CU part.  With the 4-wave 64x64 / 96x128 tiles that fill the chip, every CU re-fetches its operands from L2 so
// often (885 MB for the stage-6 project conv, 3.5 MB per CU) and keeps so few bytes in flight (one 16 KB stage per workgroup) that the
// L2 -> LDS stream, not the matrix pipe, sets the pace: PMC shows MFMA busy 15-17 %, waves parked 64 % (profiles/r02c_*_pmc_kernels.txt).
// This kernel attacks both terms of  time ~ bytes fetched / (bytes in flight / latency):
//   * tiles up to 256x192 / 192x256 with 8 waves (bytes per FLOP down 2-4x),
//   * an NBUF-deep ring of direct-to-LDS stages (NBUF-1 stages in flight, counted s_waitcnt, one barrier per K step),
//   * K split over S workgroups of the SAME XCD when the big tiles alone would leave CUs idle (project convs: K = 768..3840, only
//     2.4 M outputs): each workgroup parks its fp32 partial tile in the op's `aux` scratch in accumulator-fragment order (every store
//     and load is a coalesced 1 KiB wave access, no LDS transpose), the LAST one to arrive (per-tile arrival counter) sums the S
//     partials in split order 0..S-1 -- bit-identical whatever the arrival order -- and runs the normal epilogue.  The counter is
//     left at zero for the next launch; ftc_plan_run zeroes it once per run for safety.
// Same LDS image as the glds kernel: unpadded 128-byte rows (K step 64), 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the DMA
// source side and on the fragment read (conflict-free ds_read_b128).
#pragma once
#include "conv_igemm_impl.h"

namespace convimpl {

template <int NL, int MAXT>
__device__ __forceinline__ void wait_tiles(int rem) {            // at most min(rem, MAXT) later tiles may stay outstanding
    if constexpr (MAXT == 0) {
        wait_vmcnt<0>();
    } else {
        if (rem >= MAXT) wait_vmcnt<MAXT * NL>();
        else wait_tiles<NL, MAXT - 1>(rem);
    }
}

template <typename OutT, int TN, int TM>
constexpr bool pw_epi_fits() { return (size_t)TM * epi_pitch<OutT>(TN) + (size_t)16 * TN * 4 <= 160 * 1024; }

template <typename WT, typename OutT, int WN, int WM, int SN, int SM, int NBUF>
__global__ __launch_bounds__(64 * WN * WM) void pw_gemm_kernel(const ConvP p_launch) {
    ConvP p = p_launch;
    constexpr int NT = 64 * WN * WM, NW = WN * WM;
    constexpr int E = 8, CPR = 8, ROWB = 128;
    constexpr int TN = WN * SN * 32, TM = WM * SM * 32;
    constexpr int NCH = (TN + TM) * CPR;         // 16-byte chunks per stage
    constexpr int NL = NCH / NT;                 // DMA instructions per thread per stage
    constexpr int ACH = TN * CPR;
    constexpr int BUFB = NCH * 16;
    constexpr int D = NBUF - 1;                  // stages in flight
    static_assert(sizeof(WT) == 2, "16-bit operands");
    static_assert(NCH % NT == 0 && ACH % 64 == 0, "a stage must be a whole number of wave-level DMAs");
    static_assert((NW == 4 || NW == 8) && NBUF >= 2 && (D - 1) * NL <= 63, "");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int half = lane >> 5, l31 = lane & 31;

    // workgroup -> (tile, K split): XCD x (= blockIdx & 7, the hardware's round-robin) owns a contiguous range of tiles, and the S
    // splits of a tile are consecutive workgroups of that XCD, so partial tiles meet in one L2
    const int S = p.pw_split;
    int tile, ks;
    {
        const int ntile = p.nblk_g;
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        ks = k % S;
        const int kt = k / S;
        const int q = ntile >> 3, r = ntile & 7;
        if (kt >= q + (xcd < r ? 1 : 0)) return;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kt;
    }
    const int mt = tile / p.nN, nt = tile - mt * p.nN;
    const int m0 = mt * TM, n0 = nt * TN;
    const int it0 = (int)((long)ks * p.nk / S), it1 = (int)((long)(ks + 1) * p.nk / S);
    const int nsteps = it1 - it0;

    const __amdgpu_buffer_rsrc_t rw = weight_rsrc(p, m0);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);

    int s_off[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int q = i * NT + t;
        const bool isA = (ACH % NT == 0) ? (i < ACH / NT) : (i * NT + wave * 64 < ACH);
        const int qq = isA ? q : q - ACH;
        const int row = qq / CPR, slot = qq % CPR;
        const int kc = slot ^ ((row >> 1) & 7);
        if (isA) {
            const int n = n0 + row;
            s_off[i] = n < p.Cout ? (n * p.Cin + kc * E) * 2 : OOB;
        } else {
            const int m = m0 + row;
            s_off[i] = m < p.M ? (m * p.CinT + p.cin_off + kc * E) * 2 : OOB;
        }
    }
    auto issue = [&](int bufoff, int step) {
        const int soff = step * (64 * 2);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const bool isA = (ACH % NT == 0) ? (i < ACH / NT) : (i * NT + wave * 64 < ACH);
            lds_void_t* dst = (lds_void_t*)(smem_raw + bufoff + (i * NT + wave * 64) * 16);
            glds16(isA ? rw : rin, dst, s_off[i], soff);
        }
    };

    f32x16 acc[SN][SM];
#pragma unroll
    for (int i = 0; i < SN; ++i)
#pragma unroll
        for (int j = 0; j < SM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    using FragT = typename Frag<WT>::type;
    constexpr int G = CPR / 2;
    const int fr = (l31 >> 1) & 7;
    int offA[G], offB[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int sl = ((g * 2 + half) ^ fr) * 16;
        offA[g] = (wn * SN * 32 + l31) * ROWB + sl;
        offB[g] = (TN + wm * SM * 32 + l31) * ROWB + sl;
    }
    auto compute = [&](int bufoff) {
        const unsigned char* base = smem_raw + bufoff;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            FragT af[SN], bf[SM];
#pragma unroll
            for (int i = 0; i < SN; ++i) af[i] = *reinterpret_cast<const FragT*>(base + offA[g] + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < SM; ++j) bf[j] = *reinterpret_cast<const FragT*>(base + offB[g] + j * 32 * ROWB);
#pragma unroll
            for (int i = 0; i < SN; ++i)
#pragma unroll
                for (int j = 0; j < SM; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
    };

    int iss_off = 0, cur_off = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (j < nsteps) issue(iss_off, it0 + j);
        iss_off += BUFB;
    }
    for (int it = 0; it < nsteps; ++it) {
        wait_tiles<NL, D - 1>(nsteps - 1 - it);
        wg_barrier();
        if (it + D < nsteps) issue(iss_off, it0 + it + D);     // that slot was consumed in step it-1
        iss_off = iss_off + BUFB == NBUF * BUFB ? 0 : iss_off + BUFB;
        compute(cur_off);
        cur_off = cur_off + BUFB == NBUF * BUFB ? 0 : cur_off + BUFB;
    }

    if (S > 1) {
        float* slot0 = p.pw_ws + (size_t)tile * S * (TN * TM);
        {
            float* mine = slot0 + (size_t)ks * (TN * TM) + (size_t)wave * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < SN; ++i)
#pragma unroll
                for (int j = 0; j < SM; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(mine + (size_t)(((i * SM + j) * 4 + q) * NW) * 256) = v;
                    }
        }
        __threadfence();
        __syncthreads();                                        // also: every wave is done with the operand ring
        int* flag = reinterpret_cast<int*>(smem_raw);
        if (t == 0) {
            const int old = atomicAdd(p.pw_cnt + tile, 1);
            const int last = old == S - 1;
            if (last) atomicExch(p.pw_cnt + tile, 0);
            *flag = last;
        }
        __syncthreads();
        const int last = *reinterpret_cast<volatile int*>(flag);
        if (!last) return;
        __threadfence();
#pragma unroll
        for (int i = 0; i < SN; ++i)
#pragma unroll
            for (int j = 0; j < SM; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        for (int s = 0; s < S; ++s) {
            const float* src = slot0 + (size_t)s * (TN * TM) + (size_t)wave * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < SN; ++i)
#pragma unroll
                for (int j = 0; j < SM; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)(((i * SM + j) * 4 + q) * NW) * 256);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
                    }
        }
    }

    if constexpr (pw_epi_fits<OutT, TN, TM>()) {
        if (epi_lds_ok<OutT>(p)) {
            conv_epilogue_lds<WT, OutT, SN, SM, NT, TN, TM>(p, acc, smem_raw, n0, wn * SN * 32, wm * SM * 32, half, l31,
                                                            [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; });
            return;
        }
    }
    conv_epilogue<WT, OutT, SN, SM>(p, acc, m0, n0, wn, wm, half, l31);
}

template <typename WT, typename OutT, int WN, int WM, int SN, int SM, int NBUF>
hipError_t launch_pw_variant(ConvP p, const PwVariant& v, hipStream_t s) {
    constexpr int TN = WN * SN * 32, TM = WM * SM * 32;
    if (v.tn != TN || v.tm != TM || v.wn != WN || v.wm != WM || v.nbuf != NBUF) return hipErrorInvalidValue;   // kPw (validation, scratch sizes) out of step with this switch
    constexpr size_t lds_stage = (size_t)NBUF * (TN + TM) * 128;
    constexpr size_t lds_epi = pw_epi_fits<OutT, TN, TM>() ? (size_t)TM * epi_pitch<OutT>(TN) + (size_t)16 * TN * 4 : 0;
    constexpr size_t lds_bytes = lds_stage > lds_epi ? lds_stage : lds_epi;
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = pw_gemm_kernel<WT, OutT, WN, WM, SN, SM, NBUF>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.ncb = p.Cin / 64;
    p.nk = p.ncb;
    p.nN = (p.Cout + TN - 1) / TN;
    p.nblk_g = p.nN * ((p.M + TM - 1) / TM);
    p.nblk = ((p.nblk_g + 7) / 8) * 8 * p.pw_split;
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(64 * WN * WM), lds_bytes, s, p);
    return hipGetLastError();
}

template <typename WT, typename OutT>
hipError_t launch_pw(const ConvP& p, const ftc_op& o, hipStream_t s) {
    const PwVariant& v = kPw[hint_pw(o) <= PW_COUNT ? hint_pw(o) : 0];
    switch (hint_pw(o)) {
    case 1: return launch_pw_variant<WT, OutT, 2, 2, 1, 1, 4>(p, v, s);
    case 2: return launch_pw_variant<WT, OutT, 2, 2, 2, 2, 4>(p, v, s);
    case 3: return launch_pw_variant<WT, OutT, 2, 4, 2, 1, 4>(p, v, s);
    case 4: return launch_pw_variant<WT, OutT, 2, 4, 3, 2, 2>(p, v, s);
    case 5: return launch_pw_variant<WT, OutT, 2, 4, 2, 2, 3>(p, v, s);
    case 6: return launch_pw_variant<WT, OutT, 4, 2, 2, 2, 3>(p, v, s);
    case 7: return launch_pw_variant<WT, OutT, 4, 2, 1, 3, 3>(p, v, s);
    case 8: return launch_pw_variant<WT, OutT, 4, 2, 2, 3, 2>(p, v, s);
    case 9: return launch_pw_variant<WT, OutT, 2, 4, 3, 1, 3>(p, v, s);
    case 10: return launch_pw_variant<WT, OutT, 1, 4, 3, 1, 4>(p, v, s);
    case 11: return launch_pw_variant<WT, OutT, 2, 2, 1, 2, 4>(p, v, s);
    case 12: return launch_pw_variant<WT, OutT, 2, 2, 2, 1, 4>(p, v, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace convimpl
