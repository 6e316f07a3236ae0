// 1x1 convolution on (64 | 80 | 96 | 128)-channel x 144-pixel tiles (tile configs "64x144", "80x144", "128x144", "96x144": ftc_op.aux0 low nibble 8..11):
// the project convolutions of the MBConv blocks of the low-resolution stages (M = B x 576 or B x 2304 pixels, N = 256..640, K = 1536..3840).
//
// Why this shape.  Those GEMMs are small (4608 x 512 x 3072 at batch 8) and their time on the 64x64 / 128x64 tiles follows the bytes a CU
// pulls through L2 -> LDS (a CU takes ~30 B/clk by buffer_load..lds whatever the kernel), not the matrix pipe (DESIGN.md section 5, appendix
// A3/A5).  With one workgroup per CU the tile AREA is given (M N / 256); the operand bytes per output are (TN + TM) / (TN TM): 64x64 = 1/32,
// 64x144 = 1/44, 80x144 = 1/51, 128x144 = 1/68.  576 = 4 x 144: a tile lies in one image (per-image weight sets, FTC_FLAG_W_PER_IMAGE), and
// the tile order gives each XCD whole images -- an image's d and its folded weights cross the fabric once.  The channel width is chosen so
// that the launch has 256 workgroups: 64 for N = 512 at batch 8, 80 for N = 640, 128 for N = 256 and 96 for N = 192 on the 48x48 maps.
//
// One workgroup = 8 waves, one per CU.  A stage = 64 K values of the TN weight rows and the 144 pixel rows (128-byte rows = whole cache
// lines), DMA'd straight to LDS, four stages in a ring (three in flight).  Waves 0..3 multiply: 2 channel halves x 2 K halves of a stage
// (intra-workgroup split-K; NA x 9 accumulator tiles of v_mfma_f32_16x16x32 per wave, NA + 9 fragment reads per 9 NA MFMAs); waves 4..7
// only issue the DMA pieces.  The two K halves meet through LDS in the epilogue, each wave finishing half of the pair's pixels (bias,
// residual, fp32 store and the 16-bit trunk copy).  The K order differs from the other tile configs (two interleaved K chains): results
// agree to fp32 rounding, not bitwise.
#include "conv_igemm_impl.h"

namespace convimpl {

constexpr int PX_TM = 144;
template <int NCT, bool X3 = false> struct PxGeom {
    static constexpr int TN = NCT * 16;                // output channels per workgroup
    static constexpr int NA = (NCT + 1) / 2;           // channel tiles of the first channel half (the second has NCT - NA)
    static constexpr int NPG = NCT > 5 ? 2 : 1;        // pixel groups of the multiplying waves (2: blocks 0..4 / 5..8 -- 80 accumulator registers instead of 144)
    static constexpr int NWC = 4 * NPG;                // multiplying waves: 2 channel halves x 2 K halves x NPG
    static constexpr int NT = (NWC + 4) * 64;          // + 4 loader waves
    static constexpr int JN = NPG == 1 ? 9 : 5;        // pixel blocks per multiplying wave (the second group has 4)
    static constexpr int JH = (JN + 1) / 2;            // of which the kh = 0 wave finishes the first JH, its partner the rest
    static constexpr int ROWS = TN + PX_TM;
    static constexpr int ROWB = X3 ? 256 : 128;        // bytes of K per operand row and stage: 64 K values (16-bit: 128 B; pre-split fp32: 256 B)
    static constexpr int NST = X3 ? (NCT > 4 ? 2 : 3) : 4;    // ring depth (NST - 1 stages in flight; the wide fp16x3 tiles: a double buffer is what fits)
    static constexpr int STAGE = ROWS * ROWB;
    static constexpr int PIECES = STAGE / 1024;        // DMA pieces (64 lanes x 16 B = 8 | 4 rows) per stage: 26 | 28 | 34; fp16x3: 52
    static constexpr int NPW = (PIECES + 3) / 4;       // pieces per loader wave (the last ones may have one less)
    static constexpr int RING = NST * STAGE;           // 106,496 | 114,688 | 139,264 B; fp16x3 64x144: 159,744 B
    static constexpr bool RES_LDS = !X3 && RING + TN * PX_TM * 4 <= 160 * 1024;    // the residual tile travels by DMA too (64 and 80 channels)
    static constexpr int RES_PIECES = TN * PX_TM * 4 / 1024;                // 36 | 45
    static constexpr int LDS = RING + (RES_LDS ? TN * PX_TM * 4 : 0);
    static_assert(STAGE % 1024 == 0 && NPG * 2 * JN * NA * 1024 <= RING && LDS <= 160 * 1024, "");
};

__device__ __forceinline__ f32x4 px_mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 px_mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <typename F, int... Is>
__device__ __forceinline__ void px_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void px_for(F&& f) { px_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// vmcnt wait of a loader wave: n = outstanding DMA pieces allowed (0, one or two stages of 6..9 pieces)
__device__ __forceinline__ void px_wait(int n) {
    switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 12: wait_vmcnt<12>(); break;
    case 13: wait_vmcnt<13>(); break;
    case 14: wait_vmcnt<14>(); break;
    case 16: wait_vmcnt<16>(); break;
    case 18: wait_vmcnt<18>(); break;
    default: wait_vmcnt<0>(); break;
    }
}

// chunk slot of 16-byte chunk c of residual row r in LDS (rows of TN fp32 = 16 | 20 chunks): the 16 rows a finishing wave reads at one
// chunk index land in different bank quads (chunks 16..19 of the 80-channel tile only four ways)
__device__ __forceinline__ int px_res_slot(int c, int r) { return c < 16 ? c ^ (r & 15) : 16 + ((c - 16) ^ (r & 3)); }

// K chunk swizzle of operand row r: chunk slot s of row r holds K chunk s ^ px_g(r).  16-bit operands (128-byte rows, 8 chunks): a lane reads
// chunk 4 kh + lq of its row.  fp16x3 (pre-split fp32: a 16-byte chunk = 4 K values as [hi x4 | lo x4]; 256-byte rows, 16 chunks -- every row
// starts in the same bank quad): a lane reads the chunks 8 kh + 2 lq and + 1.  Both maps put the 16 lanes the LDS serves together
// ({0-3, 12-15, 20-27}, ..) into 16 different bank quads (checked by enumeration: tools/px144_bench.py --swizzle).
template <bool X3> __device__ __forceinline__ int px_g(int r) { return X3 ? (r & 15) : ((r >> 1) & 7); }

// T = __bf16 | _Float16: 16-bit operands, a stage = 64 K values, the K halves of a pair of waves = the halves of a stage.
// T = x3f32 (FTC_FLAG_SPLIT16 + FTC_FLAG_PRESPLIT): both operands pre-split fp32, 256-byte rows, a three-stage ring, a product = three fp16
// MFMAs (the wider tiles: a two-stage ring).  (First version: 32 K values per stage, the two K groups taking the even /
// odd stages -- with a barrier per stage they alternated instead of overlapping: 1574 cycles per stage against 842 of DMA time.)
template <typename T, int NCT>
__global__ __launch_bounds__(PxGeom<NCT>::NT, 1) void conv1x1_px144_kernel(const ConvP p) {
    constexpr bool X3 = is_x3<T>;
    using GM = PxGeom<NCT, X3>;
    constexpr int ROWB = GM::ROWB, CPR = ROWB / 16, TSTR = 16 * ROWB, NST = GM::NST;
    constexpr int ES = X3 ? 4 : 2;
    constexpr int TN = GM::TN, NA = GM::NA, STAGE = GM::STAGE, PIECES = GM::PIECES, NPW = GM::NPW, JN = GM::JN, JH = GM::JH, NWC = GM::NWC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    // workgroup ids go round-robin over the 8 XCDs: XCD x takes the contiguous tile range [x nblk/8, (x+1) nblk/8) -- channel tiles
    // fastest, so at batch 8 an XCD works on ONE image (its d and its weight set stay in that XCD's L2)
    int bid = blockIdx.x;
    if ((p.nblk & 7) == 0) bid = (bid & 7) * (p.nblk >> 3) + (bid >> 3);
    const int nt = bid % p.nN, mt = bid / p.nN;
    const int n0 = nt * TN, m0 = mt * PX_TM;
    const __amdgpu_buffer_rsrc_t rw = weight_rsrc(p, m0);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);

    // Waves 0..NWC-1 multiply (2 channel halves x 2 K halves of a stage x NPG pixel groups); the last four only issue the DMA pieces.
    // (The issues first sat between the MFMAs of four multiplying waves, one per SIMD: same time -- the issue stalls were not the bound --
    // but separate loader waves keep the multiply loop free of vmcnt bookkeeping.)
    const bool loader = wave >= NWC;
    const int cw = wave & 1, kh = (wave >> 1) & 1, pg = GM::NPG == 1 ? 0 : (wave >> 2) & 1;
    const int ct0 = cw * NA, na = (NCT & 1) ? (cw ? NCT - NA : NA) : NA;    // this wave's channel tiles [ct0, ct0 + na)
    const int j0 = pg * 5, jn = GM::NPG == 1 ? 9 : (pg ? 4 : 5);           // and its pixel blocks [j0, j0 + jn)

    // DMA piece i of a stage = LDS chunks [i*64, +64) = 8 rows of 8 chunks; loader wave w issues the pieces i = w + 4 j.  Rows 0..TN-1 =
    // weights, then the 144 pixels.  Chunk slot s of row r holds K chunk s ^ g(r), g(r) = (r >> 1) & 7: the 16-lane groups in which the LDS
    // serves a ds_read_b128 ({0-3, 12-15, 20-27}, ..) then find the 16 rows x one K chunk of a 16x16x32 fragment in 16 different bank
    // quads (rows are 128 B: rows of equal parity share their bank quads).
    const int lw = wave & 3;
    int s_off[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int q = (lw + 4 * j) * 64 + lane;
        const int row = q / CPR;
        const int kc = (q & (CPR - 1)) ^ px_g<X3>(row);
        s_off[j] = row < TN ? ((n0 + row) * p.Cin) * ES + kc * 16 : ((m0 + row - TN) * p.CinT + p.cin_off) * ES + kc * 16;
    }
    auto issue_stage = [&](int step, int bufoff) {
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int i = lw + 4 * j;                                   // wave-uniform
            if (i < PIECES) {
                lds_void_t* dst = (lds_void_t*)(smem_raw + bufoff + i * 1024);
                if (i < TN * ROWB / 1024) glds16(rw, dst, s_off[j], step * ROWB);
                else glds16(rin, dst, s_off[j], step * ROWB);
            }
        }
    };
    const int npw = (PIECES - lw + 3) / 4;                              // pieces this loader issues per stage

    // The fp32 residual tile [144][TN] goes to LDS behind the ring (64- and 80-channel tiles: it fits), requested by the multiplying waves
    // before their first MFMA -- they have no other use for vmcnt and wait for it once, behind the K loop.
    const bool has_res = (p.flags & FTC_FLAG_RESIDUAL) != 0;
    const bool res_lds = GM::RES_LDS && has_res && p.res_dtype == FTC_F32;
    if constexpr (GM::RES_LDS) {
        if (res_lds && !loader) {
            const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(static_cast<const char*>(p.res) + ((size_t)m0 * p.Cout + n0) * 4), 0, (unsigned)((PX_TM - 1) * p.Cout + TN) * 4u, 0x00020000);
            constexpr int CPRW = TN / 4;
#pragma unroll
            for (int j = 0; j < (GM::RES_PIECES + NWC - 1) / NWC; ++j) {
                const int i = wave + NWC * j;
                if (i < GM::RES_PIECES) {
                    const int q = i * 64 + lane;
                    const int row = q / CPRW, slot = q - row * CPRW;
                    // the chunk that belongs in this slot: the slot map is an involution per row
                    glds16(rres, (lds_void_t*)(smem_raw + GM::RING + i * 1024), row * p.Cout * 4 + px_res_slot(slot, row) * 16, 0);
                }
            }
        }
    }

    // accumulators: the K-half-0 waves start at the bias
    f32x4 acc[NA][JN];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const f32x4 bi = (!loader && kh == 0 && i < na) ? *reinterpret_cast<const f32x4*>(p.bias + n0 + (ct0 + i) * 16 + 4 * lq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = bi;
    }

    using FragT = typename Frag<T>::type;
    const int swz = (((X3 ? 8 * kh + 2 * lq : kh * 4 + lq) ^ px_g<X3>(l15)) << 4);
    const int swz1 = (((8 * kh + 2 * lq + 1) ^ px_g<X3>(l15)) << 4);   // fp16x3: the second chunk of a fragment
    const int offA = (ct0 * 16 + l15) * ROWB + swz;                     // + i * TSTR
    const int offB = (TN + j0 * 16 + l15) * ROWB + swz;                 // + j * TSTR
    const int nk = p.Cin >> 6;
    if (loader) {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < nk) issue_stage(s, s * STAGE);
    }
    int cur_off = 0, iss_off = (NST - 1) * STAGE;
    // phase timeline of wave 0 (flag 0x1000, tools/px144_bench.py): start, first stage landed, K loop done, exchange done, end, barrier wait cycles
    unsigned long long* tl = (p.w2 && t == 0) ? reinterpret_cast<unsigned long long*>(const_cast<void*>(p.w2)) + (size_t)blockIdx.x * 8 : nullptr;
    unsigned long long tw = 0;
    if (tl) tl[0] = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < nk; ++it) {
        const unsigned long long ta = tl ? __builtin_amdgcn_s_memtime() : 0;
        if (loader) {
            // stage `it` has landed once at most the later-issued stages remain outstanding (per-wave piece counts; vmcnt is in order)
            const int ahead = nk - 1 - it;
            px_wait((ahead < NST - 2 ? ahead : NST - 2) * npw);
        }
        wg_barrier();
        if (tl) { const unsigned long long tb = __builtin_amdgcn_s_memtime(); tw += tb - ta; if (it == 0) tl[1] = tb; }
        if (loader) {
            // stage it + 3 goes to the slot consumed in step it - 1: every multiplying wave has passed this barrier behind its reads
            if (it + NST - 1 < nk) issue_stage(it + NST - 1, iss_off);
            iss_off = iss_off + STAGE == GM::RING ? 0 : iss_off + STAGE;
        } else {
            const unsigned char* base = smem_raw + cur_off;
            if constexpr (X3) {
                {
                    const int d1 = swz1 - swz;
                    f16x8 ah[NA], al[NA];
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        const unsigned char* a = base + offA + (i < na ? i : 0) * TSTR;
                        frag_hl(*reinterpret_cast<const f32x4*>(a), *reinterpret_cast<const f32x4*>(a + d1), ah[i], al[i]);
                    }
                    f32x4 bq[2][2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bq[j][0] = *reinterpret_cast<const f32x4*>(base + offB + j * TSTR);
                        bq[j][1] = *reinterpret_cast<const f32x4*>(base + offB + j * TSTR + d1);
                    }
#pragma unroll
                    for (int j = 0; j < JN; ++j) {
                        f16x8 bh, bl;
                        frag_hl(bq[j % 2][0], bq[j % 2][1], bh, bl);
                        if (j + 2 < JN) {
                            const unsigned char* b = base + offB + (j + 2 < jn ? j + 2 : 0) * TSTR;
                            bq[j % 2][0] = *reinterpret_cast<const f32x4*>(b);
                            bq[j % 2][1] = *reinterpret_cast<const f32x4*>(b + d1);
                        }
                        if (j < jn) {
#pragma unroll
                            for (int i = 0; i < NA; ++i)
                                if (i < na) {
                                    acc[i][j] = px_mfma(ah[i], bl, acc[i][j]);
                                    acc[i][j] = px_mfma(al[i], bh, acc[i][j]);
                                    acc[i][j] = px_mfma(ah[i], bh, acc[i][j]);
                                }
                        }
                    }
                }
            } else {
                FragT af[NA];
#pragma unroll
                for (int i = 0; i < NA; ++i) af[i] = *reinterpret_cast<const FragT*>(base + offA + (i < na ? i : 0) * TSTR);
                FragT bq[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) bq[j] = *reinterpret_cast<const FragT*>(base + offB + j * TSTR);
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    const FragT bcur = bq[j % 3];
                    if (j + 3 < JN) bq[j % 3] = *reinterpret_cast<const FragT*>(base + offB + (j + 3 < jn ? j + 3 : 0) * TSTR);
                    if (j < jn) {
#pragma unroll
                        for (int i = 0; i < NA; ++i)
                            if (i < na) acc[i][j] = px_mfma(af[i], bcur, acc[i][j]);
                    }
                }
            }
            cur_off = cur_off + STAGE == GM::RING ? 0 : cur_off + STAGE;
        }
    }
    if (tl) { tl[2] = __builtin_amdgcn_s_memtime(); tl[5] = tw; }

    // ---- epilogue.  The two K halves of a (channel half, pixel group) meet: the wave with kh = 0 finishes its first JH pixel blocks, its
    // partner the rest; each sends the other's part to LDS in its own register layout ([tile][lane] x 16 B: conflict-free) and adds what
    // it received; the sum is (bias + kh 0 part) + (kh 1 part) on both sides.  Residual: from LDS (above), else loaded RD tiles ahead of their
    // use, the first RD before the waves meet (one load per tile behind its own s_waitcnt -- first version -- put ten HBM latencies end
    // to end at the tail of every workgroup: 29.6 -> 26.5 us on stage 6).  The barrier between sending and receiving waits for LDS only
    // (__syncthreads would also wait for those loads).
    constexpr int RD = GM::NPG == 1 ? 6 : 4;
    f32x4 rv[RD];
    auto load_res = [&](int i, int jl) __attribute__((always_inline)) -> f32x4 {           // (the slow path: 16-bit residuals, or an fp32 residual beside a 128-channel tile)
        if (!has_res || res_lds || i >= na || jl >= jn) return f32x4{0.f, 0.f, 0.f, 0.f};
        const size_t off = (size_t)(m0 + (j0 + jl) * 16 + l15) * p.Cout + n0 + (ct0 + i) * 16 + 4 * lq;
        if (p.res_dtype == FTC_F32) return load4<float>(reinterpret_cast<const float*>(p.res) + off);
        return load4<typename Half16<T>::type>(reinterpret_cast<const typename Half16<T>::type*>(p.res) + off);
    };
    // tiles of a finishing range [J0, J1) in order idx -> (jl = J0 + idx / NA, i = idx % NA)
    auto prefetch = [&](auto J0c, auto J1c) __attribute__((always_inline)) {
        constexpr int J0 = decltype(J0c)::value, NTL = (decltype(J1c)::value - J0) * NA;
#pragma unroll
        for (int idx = 0; idx < RD && idx < NTL; ++idx) rv[idx] = load_res(idx % NA, J0 + idx / NA);
    };
    // Output addressing without a branch per tile (every decision that depends on a flag is taken once, outside the unrolled loops: with
    // flags tested per tile -- first version -- each tile was its own basic block, LDS read -> wait -> store, 9.1 k cycles for ten tiles):
    // fp32 out[m][cout_off + n]; the 16-bit copy at ibase + r SR + (n >> 5) SP + (n & 31), which is NHWC with (SR, SP) = (Cout, 32) and the
    // 32-channel planes of FTC_FLAG_KBLOCK32 with (SR, SP) = (32, 32 Ho Wo).
    using H16 = typename Half16<typename std::conditional<X3, _Float16, T>::type>::type;      // (fp16x3: the copy is the pre-split fp32 chunk, NHWC)
    float* __restrict__ out_base = reinterpret_cast<float*>(p.out) + (size_t)m0 * p.CoutT + p.cout_off + n0;
    const int hw = p.Ho * p.Wo;
    const bool kb = (p.flags & FTC_FLAG_KBLOCK32) != 0;
    const int img = m0 / hw;
    const int SR = kb ? 32 : p.Cout, SP = kb ? hw * 32 : 32;
    H16* __restrict__ out2_base = reinterpret_cast<H16*>(p.out2) + (kb ? (size_t)img * p.Cout * hw + (size_t)(m0 - img * hw) * 32 : (size_t)m0 * p.Cout);
    const unsigned char* xin = smem_raw + (pg * 2 + cw) * (JN * NA * 1024) + lane * 16;
    // RM: where the residual comes from (0 none, 1 the LDS tile, 2 the rv ring); COPY: the 16-bit copy is written
    auto finish_range = [&](auto J0c, auto J1c, auto K0c, auto RMc, auto COPYc) __attribute__((always_inline)) {
        constexpr int J0 = decltype(J0c)::value, NTL = (decltype(J1c)::value - J0) * NA, RM = decltype(RMc)::value;
        constexpr bool mine_k0 = decltype(K0c)::value, COPY = decltype(COPYc)::value;
        px_for<NTL>([&](auto IDXc) __attribute__((always_inline)) {
            constexpr int idx = decltype(IDXc)::value, i = idx % NA, jl = J0 + idx / NA;
            // a tile this wave does not have (the fifth pixel block of the second pixel group, the third channel tile of the second channel
            // half of an 80-channel tile) repeats its neighbour: the same values to the same addresses instead of a branch
            const bool jok = GM::NPG == 1 || jl < JN - 1 || jl < jn, iok = !(NCT & 1) || i < NA - 1 || i < na;
            const int jv = jok ? jl : jl - 1, iv = iok ? i : i - 1;
            f32x4 a = acc[i][jl];
            if constexpr (GM::NPG == 2) { if (jl == JN - 1) a = jok ? a : acc[i][jl > 0 ? jl - 1 : 0]; }
            if constexpr ((NCT & 1) != 0) { if (i == NA - 1) a = iok ? a : acc[i > 0 ? i - 1 : 0][jl]; }
            f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (RM == 2) {
                r = rv[idx % RD];
                if (idx + RD < NTL) rv[idx % RD] = load_res((idx + RD) % NA, J0 + (idx + RD) / NA);
                if (!(jok && iok)) r = load_res(iv, jv);
            }
            const int row = (j0 + jv) * 16 + l15, ch = (ct0 + iv) * 16 + 4 * lq;
            const f32x4 other = *reinterpret_cast<const f32x4*>(xin + (jv * NA + iv) * 1024);
            if constexpr (RM == 1 && GM::RES_LDS) r = *reinterpret_cast<const f32x4*>(smem_raw + GM::RING + row * (TN * 4) + px_res_slot((ct0 + iv) * 4 + lq, row) * 16);
            f32x4 v = mine_k0 ? a + other : other + a;
            v += r;
            store4<float>(out_base + (size_t)row * p.CoutT + ch, v);
            if constexpr (COPY && X3) {
                *reinterpret_cast<u32x4*>(static_cast<char*>(p.out2) + ((size_t)(m0 + row) * p.Cout + n0 + ch) * 4) = chunk_hl(v);
            } else if constexpr (COPY) {
                const int nn = n0 + ch;
                store4<H16>(out2_base + (size_t)row * SR + (size_t)(nn >> 5) * SP + (nn & 31), v);
            }
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, JH>;
    using IN = std::integral_constant<int, JN>;
    if (!loader) {
        if (kh == 0) prefetch(I0{}, IH{}); else prefetch(IH{}, IN{});
    }
    wg_barrier();                                                       // every wave is done with the ring
    // one exchange area per (pixel group, channel half): JN x NA tiles of 1 KB, tile (jl, i) written by the wave that does not finish it
    unsigned char* xout = smem_raw + (pg * 2 + cw) * (JN * NA * 1024) + lane * 16;
    if (!loader) {
#pragma unroll
        for (int jl = 0; jl < JN; ++jl)
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (i < na && jl < jn && (jl < JH) != (kh == 0)) *reinterpret_cast<f32x4*>(xout + (jl * NA + i) * 1024) = acc[i][jl];
        if (res_lds) wait_vmcnt<0>();                                   // the residual tile has landed (requested before the K loop)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();
    if (loader) return;
    if (tl) tl[3] = __builtin_amdgcn_s_memtime();
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    auto finish = [&](auto RMc, auto COPYc) __attribute__((always_inline)) {
        if (kh == 0) finish_range(I0{}, IH{}, std::true_type{}, RMc, COPYc);
        else finish_range(IH{}, IN{}, std::false_type{}, RMc, COPYc);
    };
    const int rm = !has_res ? 0 : res_lds ? 1 : 2;
    if (p.out2) { if (rm == 0) finish(R0{}, std::true_type{}); else if (rm == 1) finish(R1{}, std::true_type{}); else finish(R2{}, std::true_type{}); }
    else { if (rm == 0) finish(R0{}, std::false_type{}); else if (rm == 1) finish(R1{}, std::false_type{}); else finish(R2{}, std::false_type{}); }
    if (tl) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl[4] = __builtin_amdgcn_s_memtime(); }
}

}  // namespace convimpl

using namespace convimpl;

template <typename T, int NCT>
static hipError_t launch_px(ConvP& p, hipStream_t s) {
    using GM = PxGeom<NCT, is_x3<T>>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_px144_kernel<T, NCT>), hipFuncAttributeMaxDynamicSharedMemorySize, GM::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.nN = p.Cout / GM::TN;
    p.nblk = p.nN * (p.M / PX_TM);
    hipLaunchKernelGGL((conv1x1_px144_kernel<T, NCT>), dim3(p.nblk), dim3(GM::NT), GM::LDS, s, p);
    return hipGetLastError();
}

hipError_t launch_conv1x1_px144(const ConvP& p0, const ftc_op& o, hipStream_t s) {
    ConvP p = p0;
    const int tn = kCfgTN[select_cfg(o)];
    if (o.w_dtype == FTC_F32) return tn == 64 ? launch_px<x3f32, 4>(p, s) : tn == 80 ? launch_px<x3f32, 5>(p, s) : tn == 96 ? launch_px<x3f32, 6>(p, s) : launch_px<x3f32, 8>(p, s);
    if (o.w_dtype == FTC_BF16) return tn == 64 ? launch_px<__bf16, 4>(p, s) : tn == 80 ? launch_px<__bf16, 5>(p, s) : tn == 96 ? launch_px<__bf16, 6>(p, s) : launch_px<__bf16, 8>(p, s);
    return tn == 64 ? launch_px<_Float16, 4>(p, s) : tn == 80 ? launch_px<_Float16, 5>(p, s) : tn == 96 ? launch_px<_Float16, 6>(p, s) : launch_px<_Float16, 8>(p, s);
}
