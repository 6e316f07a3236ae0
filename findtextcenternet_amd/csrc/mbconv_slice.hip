// MBConv head of the low-resolution stages in ONE kernel (FTC_OP_MBHEAD): expand 1x1 convolution (+BN+SiLU) -> depthwise 3x3 stride 1
// (+BN+SiLU) -> SE squeeze (+ this slice's share of the SE fc1 layer).  Reference: torchvision MBConv block[0], block[1] and the
// AdaptiveAvgPool2d + fc1 of block[2] as instantiated by /root/reference/models/detector.py:17-20 (stages 6-7 at 768x768: 24x24 maps,
// 512 -> 3072 and 640 -> 3840 channels, 39 blocks).
//
// Why a third fused form (two were measured and rejected): at batch 8 these blocks are bound by
// the memory system -- the 28-35 MB expanded tensor crosses L2/HBM four times per block -- and both earlier forms tiled the IMAGE
// (8x8 tiles; 12-row bands x 128 channels), which recomputes halo rows of the expand GEMM and leaves partial squeeze sums for a second
// pass.  Here the CHANNELS are sliced instead: a workgroup (8 waves) owns one WHOLE image x 128 expanded channels.
//   1. GEMM  e[576 px][128] = x[576][K] . W_e[128][K]^T : x and W_e stream through a 3-stage direct-to-LDS ring in K steps of 32
//      (64-byte rows, chunk-swizzled; two stages in flight across the one barrier per step, their DMA issues spread between the
//      MFMA groups); 8 waves = 2 channel halves x 4 pixel quarters, 4 x 9 accumulator tiles (v_mfma_f32_16x16x32) per wave: 144 of
//      the 256 registers two waves per SIMD get.  The accumulators of the whole 576 x 128 tile (288 KB) live in the CU's register
//      file; they start at the expand bias.
//   2. SiLU, rounded to the 16-bit type exactly like the expand convolution's epilogue, into an LDS image of the WHOLE map:
//      [pixel slot][128 ch] in 264-byte rows (256 + 8: sixteen consecutive slots' 8-byte stores hit sixteen different bank pairs; a
//      row's 32 chunks are one conflict-free ds_read_b64 of a half-wave), one zero slot between consecutive image rows
//      (slot(y, x) = y (W+1) + x + 1): the left / right zero padding of the depthwise convolution is data, the rows above / below the
//      map read slot 0.  601 x 264 B = 155 KB, aliasing the operand ring: one workgroup per CU.
//   3. depthwise 3x3 out of that image: lane = 4 channels x a vertical strip of 6 outputs (8 x 3 reads of 8 bytes at immediate
//      offsets), fp32 weights and accumulation as dwconv_strip_kernel, + bias + SiLU, 8-byte stores (32 lanes = one pixel's 256 bytes).
//   4. the channel sums of the WHOLE image are complete inside the workgroup: no partial-sum slots, no second reduction pass.  The
//      workgroup also multiplies its 128 means into the SE fc1 layer: hpart[b][slice][s] = sum_c fc1_w[s][c] mean[c], so the SE kernel
//      that follows only adds C/128 partial vectors (fixed order: deterministic) instead of re-reading sums and the fc1 matrix.
// L2 -> LDS traffic bounds the GEMM phase: every workgroup of an image streams the image's whole x (590 KB at 24x24x512).  The first
// version of this kernel (64 channels, 256 threads, two workgroups per CU) moved 655 KB per 64 channels and took 54 us per stage-6
// block -- its K loop ran at the 38 B/clk/CU the L2 delivered; 128 channels halve that traffic and balance it against the MFMA time
// (per K step of 32: 45 KB of DMA against 36 MFMAs per SIMD).  Image b's workgroups are placed on XCD b % 8.
#include "conv_igemm_impl.h"

namespace {
using namespace convimpl;

struct MbsP {
    const void* x;          // [B][H*W][K] 16-bit
    const void* we;         // [C][K] 16-bit, K-major
    const float* be;        // [C] expand bias (folded BN)
    const float* wd;        // [9][C] depthwise weights
    const float* bd;        // [C] depthwise bias
    void* out;              // [B][H*W][C] 16-bit
    float* sums;            // [B][C] channel sums of the output (the P = 1 form of FTC_OP_DWCONV's partial sums)
    const float* w1;        // SE fc1 weight [S][C] (optional)
    float* hpart;           // [B][C/128][S] (optional)
    int B, H, W, K, C, S;
    int R, nb;              // band mode: output rows per band, bands per image (R = H, nb = 1: the whole image)
    int kblock;             // FTC_FLAG_KBLOCK32: x is [B][K/32][H*W][32]
    unsigned img_bytes;
    float inv_hw;
    unsigned long long* tl;  // flags 0x1000: s_memtime of wave 0 at the phase boundaries and K steps, 32 values per workgroup (tools/mbslice_bench.py)
};

constexpr int MS_NT = 512;                  // threads
constexpr int MS_MAXPX = 576;               // pixels per image (36 MFMA pixel blocks of 16)
constexpr int MS_XS = MS_MAXPX * 64;        // bytes of the x part of a stage (64-byte rows: K step 32)
constexpr int MS_NSTAGE = 3;
constexpr int MS_MAXSLOT = 601;             // pixel slots of the expanded image (24 x 25 + 1)
constexpr int MS_R = 6;                     // outputs per depthwise strip (12: 3.5 instead of 4 reads per output, but 44 registers spilled)
// Slice width (round 5): NCT channel tiles of 16 per wave = 32 NCT expanded channels per workgroup.  128 (NCT = 4) is the default; 96 (NCT = 3) for
// blocks whose 128-channel slices leave CUs without a workgroup: stage 6 at batch 8 is 8 x 24 = 192 workgroups on 256 CUs, 8 x 32 = 256 with 96.
template <int NCT> struct MsGeom {
    static constexpr int CC = NCT * 32;                  // expanded channels per workgroup
    static constexpr int CQN = CC / 4;                   // channel quads = lanes per depthwise strip lane group (32 | 24)
    static constexpr int NPL = MS_NT / CQN;              // depthwise strip lanes (16 | 21; threads beyond NPL * CQN idle through that phase)
    static constexpr int STAGE = MS_XS + CC * 64;
    static constexpr int PITCH = CC * 2 + 8;             // bytes per slot
    static constexpr int IMG = MS_MAXSLOT * PITCH, RING = MS_NSTAGE * STAGE;
    static constexpr int CONST = ((IMG > RING ? IMG : RING) + 15) / 16 * 16;      // depthwise weights [9][CC] + bias [CC] fp32, behind image and ring
    static constexpr int LDS = CONST + 10 * CC * 4;
    static constexpr int NPIECE = 36 + CC / 16;          // DMA pieces of a stage: 36 of x, CC / 16 of weights (44 | 42)
    static constexpr int NW6 = NPIECE - 40;              // waves that issue six pieces (the others five)
    static_assert(LDS <= 160 * 1024 && NPIECE > 40 && NPIECE <= 48, "");
};

__device__ __forceinline__ f32x4 mfma16x16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16x16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// FAST: the 24x24 map of the 768x768 plans (H, W compile-time: row offsets of the depthwise window are instruction immediates)
// (Two forms of the SqueezeExcitation INSIDE this launch were built, measured and removed: round 4's FTC_FLAG_SE_INLINE -- gated output, plain
// project GEMM: 60.0 vs 41.7 + 14.9 us -- and round 5's FTC_FLAG_SE_TAIL -- ungated output, the last workgroups of an image fold the project
// weights: 61.9 vs 42.5 + 14.8 us on one stream and waits that run into their bound under two lanes; profiles/r04_mbhead_se_inline_experiment.txt,
// profiles/r05_se_tail_experiment.txt, DESIGN.md appendix.)
template <typename T, bool FAST, int NCT = 4>
__global__ __launch_bounds__(MS_NT, 2) void mbconv_slice_kernel(const MbsP p) {
    using GM = MsGeom<NCT>;
    constexpr int MS_CC = GM::CC, MS_STAGE = GM::STAGE, MS_PITCH = GM::PITCH, MS_CONST = GM::CONST, CQN = GM::CQN, NPL = GM::NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int bid = blockIdx.x;
    const int b = bid % p.B;                               // consecutive workgroup ids = consecutive images: image b on XCD b % 8
    const int rest = bid / p.B;
    const int band = FAST ? 0 : rest % p.nb, sl = FAST ? rest : rest / p.nb;
    const int c0 = sl * MS_CC;
    // Band mode (maps larger than 576 pixels: the 48x48 stages): the workgroup owns the output rows [y0, y1) of its image and holds the
    // expanded rows [ylo, yhi) = one halo row above and below (recomputed by the neighbouring band); H below = the rows it HOLDS.
    const int W = FAST ? 24 : p.W;
    const int y0 = FAST ? 0 : band * p.R, y1 = FAST ? 24 : min(p.H, y0 + p.R);
    const int ylo = FAST ? 0 : max(0, y0 - 1), yhi = FAST ? 24 : min(p.H, y1 + 1);
    const int H = yhi - ylo;
    const int M = H * W;
    const int W1 = W + 1;
    const int Mfull = p.H * W;

    const unsigned xbase = (unsigned)ylo * W * (p.kblock ? 64u : (unsigned)p.K * 2u);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.x) + (size_t)b * p.img_bytes + xbase), 0,
                                                                        p.img_bytes - xbase, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwe = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.we) + (size_t)c0 * p.K * 2), 0,
                                                                         (unsigned)(MS_CC * p.K * 2), 0x00020000);

    // DMA piece i of a wave = LDS chunks [(i*8 + wave)*64, +64) of a stage: chunk q -> LDS byte q*16.  The 576 pixel rows come first
    // (2304 chunks = 36 pieces), then the 128 weight rows (512 chunks = 8 pieces): 44 pieces over 8 waves = 6 for waves 0-3, 5 for 4-7.
    // Chunk slot s of row r holds K chunk s ^ f(r), f(r) = 3 ((r >> 3) & 1): the 16-lane groups in which the LDS serves a ds_read_b128
    // ({0-3, 12-15, 20-27}, ..) then find the 16 rows x one K chunk of a 16x16x32 fragment in 16 different bank quads.
    constexpr int NLMAX = 6;
    int s_off[NLMAX];
#pragma unroll
    for (int i = 0; i < NLMAX; ++i) {
        const int q = (i * 8 + wave) * 64 + lane;
        const bool isx = q < MS_MAXPX * 4;
        const int row = isx ? (q >> 2) : ((q - MS_MAXPX * 4) >> 2);
        const int kc = (q & 3) ^ (((row >> 3) & 1) * 3);
        // x in 32-channel planes (FTC_FLAG_KBLOCK32): a pixel's K step is 64 contiguous bytes and the next pixel follows -- a DMA piece
        // (16 rows) is one contiguous KiB.  From NHWC rows a piece touched 16 cache lines and used half of each; the same lines
        // came again one step later (45 KB per step against a 32 KB L1), and the K loop ran at the L1 fill rate: 1.9 k cycles per
        // step against 1.15 k of MFMA time.
        const int rowb = (isx && p.kblock) ? 64 : p.K * 2;
        s_off[i] = (!isx || row < M) ? row * rowb + kc * 16 : OOB;
    }
    const int xstep = p.kblock ? Mfull * 64 : 64;                        // bytes between consecutive K steps of x
    auto issue_piece = [&](int i, int step, int bufoff) {
        const int q0 = (i * 8 + wave) * 64;                              // wave-uniform
        if (i < 5 || wave < GM::NW6) {
            lds_void_t* dst = (lds_void_t*)(smem_raw + bufoff + q0 * 16);
            if (q0 < MS_MAXPX * 4) glds16(rx, dst, s_off[i], step * xstep);
            else glds16(rwe, dst, s_off[i], step * 64);
        }
    };

    // Waves: 2 channel halves (64) x 4 pixel quarters (144 = 9 blocks of 16): 4 x 9 accumulator tiles of 16x16 per wave (144 registers);
    // per K step 4 + 9 fragment reads feed 36 MFMAs (the 32x32x16 tiling, 1 x 9 tiles per wave, read 20 per 18 and kept the LDS busy
    // for longer than the matrix pipe).  The accumulators start at the expand bias.
    const int chw = wave & 1, pg = wave >> 1;
    f32x4 acc[NCT][9];
#pragma unroll
    for (int i = 0; i < NCT; ++i) {
        const f32x4 bi = *reinterpret_cast<const f32x4*>(p.be + c0 + chw * (MS_CC / 2) + i * 16 + 4 * lq);
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = bi;
    }
    if (t < 10 * CQN) {                                                  // depthwise weights [9][CC] + bias [CC] of this slice -> LDS
        const int row = t / CQN, ch = (t - row * CQN) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(row < 9 ? p.wd + (long)row * p.C + c0 + ch : p.bd + c0 + ch);
        *reinterpret_cast<f32x4*>(smem_raw + MS_CONST + t * 16) = v;
    }

    using FragT = typename Frag<T>::type;
    const int swz = ((lq ^ (((l15 >> 3) & 1) * 3)) << 4);
    const int offA = MS_XS + (chw * (MS_CC / 2) + l15) * 64 + swz;     // + i * 1024
    const int offB = (pg * 144 + l15) * 64 + swz;                       // + j * 1024
    const int nk = p.K >> 5;
    const bool tl_on = p.tl && t == 0;
    unsigned long long* tl = p.tl + (size_t)blockIdx.x * 32;
    if (tl_on) tl[0] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < NLMAX; ++i) issue_piece(i, 0, 0);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < NLMAX; ++i) issue_piece(i, 1, MS_STAGE);
    }
    int cur_off = 0, iss_off = 2 * MS_STAGE;
    unsigned long long tw = 0, tw0 = 0;                         // timeline: cycles wave 0 spent waiting for data + barrier, all steps / the first
    for (int it = 0; it < nk; ++it) {
        const unsigned long long ta = tl_on ? __builtin_amdgcn_s_memtime() : 0;
        // stage `it` has landed once at most the later-issued stage remains outstanding (per-wave piece counts)
        if (it + 1 >= nk) wait_vmcnt<0>();
        else if (wave < GM::NW6) wait_vmcnt<6>();
        else wait_vmcnt<5>();
        wg_barrier();
        if (tl_on) { const unsigned long long d = __builtin_amdgcn_s_memtime() - ta; tw += d; if (it == 0) tw0 = d; if (it < 24) tl[8 + it] = ta; }
        const unsigned char* base = smem_raw + cur_off;
        const bool more = it + 2 < nk;                               // stage it + 2 goes to the slot consumed in step it - 1
        FragT af[NCT];
#pragma unroll
        for (int i = 0; i < NCT; ++i) af[i] = *reinterpret_cast<const FragT*>(base + offA + i * 1024);
        // pixel fragments three groups ahead of their MFMAs (a group of 4 MFMAs lasts 64 cycles, an LDS read takes twice that)
        FragT bq[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) bq[j] = *reinterpret_cast<const FragT*>(base + offB + j * 1024);
        // The DMA pieces of the next-but-one stage are issued BETWEEN the MFMA groups: issued together right behind the barrier (first
        // version) all eight waves sat in their 5-6 DMA issues (~100+ cycles each) at the same time and the matrix pipe idled for it.
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const FragT bcur = bq[j % 3];
            if (j + 3 < 9) bq[j % 3] = *reinterpret_cast<const FragT*>(base + offB + (j + 3) * 1024);
#pragma unroll
            for (int i = 0; i < NCT; ++i) acc[i][j] = mfma16x16(af[i], bcur, acc[i][j]);
            constexpr int piece_after[9] = {0, 1, -1, 2, 3, -1, 4, 5, -1};
            if (piece_after[j] >= 0 && more) issue_piece(piece_after[j], it + 2, iss_off);
            __builtin_amdgcn_sched_barrier(0);
        }
        iss_off = iss_off + MS_STAGE == MS_NSTAGE * MS_STAGE ? 0 : iss_off + MS_STAGE;
        cur_off = cur_off + MS_STAGE == MS_NSTAGE * MS_STAGE ? 0 : cur_off + MS_STAGE;
    }
    wg_barrier();                                               // every wave is done with the operand ring: it becomes the expanded image
    if (tl_on) { tl[1] = __builtin_amdgcn_s_memtime(); tl[5] = tw; tl[6] = tw0; }

    // ---- expanded image: SiLU, 16-bit, slot(y, x) = y (W+1) + x + 1 ----
    for (int idx = t; idx < (H + 1) * CQN; idx += MS_NT) {      // the zero slots between the rows (and before the first / after the last)
        const u32x2 z = {0u, 0u};
        *reinterpret_cast<u32x2*>(smem_raw + ((idx / CQN) * W1) * MS_PITCH + (idx % CQN) * 8) = z;     // (264- / 200-byte rows are 8-byte aligned)
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int m = pg * 144 + j * 16 + l15;
        if (m < M) {
            const int y = m / W;
            unsigned char* row = smem_raw + (m + y + 1) * MS_PITCH + (chw * (MS_CC / 2) + 4 * lq) * 2;       // slot = y (W+1) + x + 1 = m + y + 1
#pragma unroll
            for (int i = 0; i < NCT; ++i) store4<T>(reinterpret_cast<T*>(row + i * 32), act_silu_fast4(acc[i][j]));
        }
    }
    __syncthreads();
    if (tl_on) tl[2] = __builtin_amdgcn_s_memtime();

    // ---- depthwise 3x3 + bias + SiLU + channel sums ----
    const int pl = t / CQN, cq = t - pl * CQN;                 // 4 channels cq*4.., strip lane 0..NPL-1 (96-channel slices: threads 504.. have no lane)
    const bool dw_on = pl < NPL;
    const int c = c0 + cq * 4;
    // this slice's columns of the SE fc1 matrix, requested now: they arrive under the depthwise phase (behind its stores they would
    // wait for every store to drain: vmcnt is in order).  Lane = 4 channels, 16 units s apart per pass.
    constexpr int NU = (FTC_MBHEAD_MAX_SQUEEZE + NPL - 1) / NPL;      // S <= 160 (validated: ftc_api.hip)
    f32x4 w1r[NU];
    auto load_w1 = [&]() {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int su = pl + NPL * i;
            w1r[i] = (dw_on && su < p.S) ? *reinterpret_cast<const f32x4*>(p.w1 + (size_t)su * p.C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (p.hpart) load_w1();
    f32x4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(smem_raw + MS_CONST + k * (MS_CC * 4) + cq * 16);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(smem_raw + MS_CONST + 9 * (MS_CC * 4) + cq * 16);
    const int yo = y0 - ylo, yend = y1 - ylo;                 // the output rows, as rows of the image held
    const int nsr = (yend - yo + MS_R - 1) / MS_R;
    const int nstrips = nsr * W;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    T* outp = reinterpret_cast<T*>(p.out) + ((size_t)b * Mfull + (size_t)ylo * W) * p.C + c;
    const unsigned char* zslot = smem_raw + cq * 8;             // slot 0: zeros
    for (int s = dw_on ? pl : nstrips; s < nstrips; s += NPL) {
        const int sr = s / W, x = s - sr * W;
        const int oy0 = yo + sr * MS_R;
        // window rows 0..6 from base0, 7.. from base1: the immediate offset of a DS instruction is 16 bits
        const unsigned char* base0 = smem_raw + ((oy0 - 1) * W1 + x) * MS_PITCH + cq * 8;      // slot of (oy0 - 1, x - 1)
        const unsigned char* base1 = base0 + 7 * W1 * MS_PITCH;
        f32x4 a[MS_R];
#pragma unroll
        for (int oo = 0; oo < MS_R; ++oo) a[oo] = bv;
#pragma unroll
        for (int r = 0; r < MS_R + 2; ++r) {
            bool rok;                                           // FAST: rows 1..R of a strip are inside the 24-row map by construction
            if (FAST) rok = r == 0 ? oy0 > 0 : r == MS_R + 1 ? oy0 + MS_R < 24 : true;
            else rok = (unsigned)(oy0 - 1 + r) < (unsigned)H;
            const unsigned char* rp = (r < 7 ? base0 + r * W1 * MS_PITCH : base1 + (r - 7) * W1 * MS_PITCH);
            f32x4 xin[3];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) xin[s2] = load4<T>(reinterpret_cast<const T*>(rok ? rp + s2 * MS_PITCH : zslot));
#pragma unroll
            for (int oo = 0; oo < MS_R; ++oo) {
                const int kr = r - oo;
                if (kr >= 0 && kr < 3) {
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[oo][e] = fmaf(wv[kr * 3 + s2][e], xin[s2][e], a[oo][e]);
                }
            }
        }
#pragma unroll
        for (int oo = 0; oo < MS_R; ++oo) {
            const int oy = oy0 + oo;
            if (FAST || oy < yend) {
                a[oo] = act_silu_fast4(a[oo]);
                store4<T>(outp + (oy * W + x) * p.C, a[oo]);
                sum += a[oo];
            }
        }
    }
    if (tl_on) tl[3] = __builtin_amdgcn_s_memtime();

    // ---- squeeze: the image's channel sums, complete in this workgroup (the expanded image is dead: its memory holds the scratch) ----
    __syncthreads();
    constexpr int NRED = NCT == 4 ? 8 : NPL;                   // partial sums per channel: one per wave (128 channels: two strip lanes per wave) | per strip lane
    float* red = reinterpret_cast<float*>(smem_raw);           // [NRED][CC]
    float* lmean = red + NRED * MS_CC;
    if constexpr (NCT == 4) {
        f32x4 v = sum;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], 32, 64);
        if (lane < 32) *reinterpret_cast<f32x4*>(red + wave * MS_CC + lane * 4) = v;
    } else {
        if (dw_on) *reinterpret_cast<f32x4*>(red + pl * MS_CC + cq * 4) = sum;
    }
    __syncthreads();
    if (t < MS_CC) {
        float tot = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < NRED; ++w8) tot += red[w8 * MS_CC + t];
        p.sums[((size_t)b * p.nb + band) * p.C + c0 + t] = tot;
        lmean[t] = tot * p.inv_hw;
    }
    if (p.hpart) {
        __syncthreads();
        // lane (4 channels) x unit products -> LDS [unit][CQN + 1], then one thread per unit adds the channel quads in order
        float* fcb = lmean + MS_CC;
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(lmean + (dw_on ? cq : 0) * 4);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const f32x4 pr = w1r[i] * m4;
            const int su = pl + NPL * i;
            if (dw_on && su < p.S) fcb[su * (CQN + 1) + cq] = (pr[0] + pr[1]) + (pr[2] + pr[3]);
        }
        __syncthreads();
        if (t < p.S) {
            float d = 0.f;
#pragma unroll
            for (int q = 0; q < CQN; ++q) d += fcb[t * (CQN + 1) + q];
            float* dst = p.hpart + (((size_t)b * p.nb + band) * (p.C / MS_CC) + sl) * p.S + t;
            *dst = d;
        }
    }
    if (tl_on) tl[4] = __builtin_amdgcn_s_memtime();
}

}  // namespace

// rows of the expanded image a workgroup holds: the whole map, or a band of aux1 output rows plus a halo row above and below
static int mbhead_rows(const ftc_op& o) { return o.aux1 > 0 ? (o.aux1 + 2 < o.H ? o.aux1 + 2 : o.H) : o.H; }

bool ftc_mbhead_legal(const ftc_op& o) {
    const int rows = mbhead_rows(o);
    const bool x3 = o.in_dtype == FTC_F32 && (o.flags & FTC_FLAG_SPLIT16);          // the fp32-tensor form (csrc/mbconv_slice_x3.hip): 64-channel slices
    const int slice = ftc_mbhead_slice(o);
    if (x3 ? slice != FTC_MBHEAD_SLICE_F32 : (slice != 128 && slice != 96)) return false;
    return (ftc_is16(o.in_dtype) || x3) && o.in_dtype == o.out_dtype && o.in_dtype == o.w_dtype && o.stride == 1 && o.ksize == 3 && o.Ho == o.H &&
           o.Wo == o.W && o.aux1 >= 0 && rows * o.W <= MS_MAXPX && rows * (o.W + 1) + 1 <= MS_MAXSLOT && o.Cout > 0 &&
           o.Cout % slice == 0 && o.Cin > 0 && o.Cin % 32 == 0;
}

// expanded channels per workgroup: ftc_op.Cout_total when given (128 | 96; 64 in the fp32-tensor form), else the default of the form
int ftc_mbhead_slice(const ftc_op& o) { return o.Cout_total > 0 ? o.Cout_total : (o.in_dtype == FTC_F32 ? FTC_MBHEAD_SLICE_F32 : FTC_MBHEAD_SLICE); }

int ftc_mbhead_bands(const ftc_op& o) { return o.aux1 > 0 ? (o.H + o.aux1 - 1) / o.aux1 : 1; }

// Band height for a map that does not fit a workgroup whole: as many output rows as leave room for the two halo rows (0 = the map fits)
int ftc_mbhead_band_rows(int H, int W) {
    if (H * W <= MS_MAXPX && H * (W + 1) + 1 <= MS_MAXSLOT) return 0;
    int r = MS_MAXPX / W;
    while (r > 2 && r * (W + 1) + 1 > MS_MAXSLOT) --r;
    return r - 2 > 0 ? r - 2 : -1;
}

hipError_t launch_mbhead_x3(const OpArgs& a, hipStream_t s);

hipError_t launch_mbhead(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    if (!ftc_mbhead_legal(o)) return hipErrorInvalidValue;
    if (o.in_dtype == FTC_F32) return launch_mbhead_x3(a, s);
    MbsP p;
    p.x = a.in; p.we = a.w2; p.be = a.bias2; p.wd = static_cast<const float*>(a.w); p.bd = a.bias; p.out = a.out; p.sums = a.aux;
    p.w1 = a.scale; p.hpart = a.scale ? static_cast<float*>(a.out2) : nullptr;
    p.B = o.B; p.H = o.H; p.W = o.W; p.K = o.Cin; p.C = o.Cout; p.S = o.aux0;
    p.kblock = (o.flags & FTC_FLAG_KBLOCK32) ? 1 : 0;
    p.R = o.aux1 > 0 ? o.aux1 : o.H; p.nb = ftc_mbhead_bands(o);
    p.img_bytes = (unsigned)((long)o.H * o.W * o.Cin * 2);
    p.inv_hw = 1.0f / (float)(o.H * o.W);
    p.tl = (o.flags & 0x1000) ? reinterpret_cast<unsigned long long*>(const_cast<void*>(a.in2)) : nullptr;      // phase timeline (tools/mbslice_bench.py)
    const int slice = ftc_mbhead_slice(o);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipSuccess;
        for (const void* f : {reinterpret_cast<const void*>(mbconv_slice_kernel<__bf16, true, 4>), reinterpret_cast<const void*>(mbconv_slice_kernel<__bf16, false, 4>),
                              reinterpret_cast<const void*>(mbconv_slice_kernel<_Float16, true, 4>), reinterpret_cast<const void*>(mbconv_slice_kernel<_Float16, false, 4>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MsGeom<4>::LDS);
        for (const void* f : {reinterpret_cast<const void*>(mbconv_slice_kernel<__bf16, true, 3>), reinterpret_cast<const void*>(mbconv_slice_kernel<__bf16, false, 3>),
                              reinterpret_cast<const void*>(mbconv_slice_kernel<_Float16, true, 3>), reinterpret_cast<const void*>(mbconv_slice_kernel<_Float16, false, 3>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MsGeom<3>::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nblk = o.B * p.nb * (o.Cout / slice);
    const bool fast = o.H == 24 && o.W == 24 && p.nb == 1 && !(o.flags & 0x100);          // 0x100: the general kernel (tests)
#define MBS_LAUNCH(T, F, N) hipLaunchKernelGGL((mbconv_slice_kernel<T, F, N>), dim3(nblk), dim3(MS_NT), MsGeom<N>::LDS, s, p)
#define MBS_PICK(T, N) do { if (fast) MBS_LAUNCH(T, true, N); else MBS_LAUNCH(T, false, N); } while (0)
    if (o.in_dtype == FTC_F16) { if (slice == 96) MBS_PICK(_Float16, 3); else MBS_PICK(_Float16, 4); }
    else { if (slice == 96) MBS_PICK(__bf16, 3); else MBS_PICK(__bf16, 4); }
#undef MBS_PICK
#undef MBS_LAUNCH
    return hipGetLastError();
}
