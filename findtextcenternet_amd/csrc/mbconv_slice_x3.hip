// FTC_OP_MBHEAD for the fp32-tensor plans (FTC_FLAG_SPLIT16, "fp16x3": the contract-grade mode -- fp32 tensors, weights, epilogues and
// accumulation, every product on the 16-bit matrix pipe as three fp16 MFMAs of hi / lo split operands).  Round 5: the fused MBConv head
// (expand 1x1 + BN + SiLU -> depthwise 3x3 + BN + SiLU -> channel sums + this slice's share of the SE fc1 layer) that the 16-bit plans have
// had since round 4 (csrc/mbconv_slice.hip), for the plan whose numbers meet BASELINE.json's tolerance.  Reference: torchvision MBConv
// block[0], block[1] and the squeeze of block[2] as instantiated by /root/reference/models/detector.py:17-20.
//
// Same decomposition -- a workgroup (8 waves) owns one image (or a band of its rows) x a slice of the expanded channels, the expanded tensor
// only ever exists in LDS -- with what the 4-byte elements change:
//   * the slice is 64 channels: the LDS image of the expanded map is [601 slots][64 ch fp32] in the same 264-byte rows;
//   * x arrives PRE-SPLIT: every 16-byte chunk of four fp32 values as [hi x4 | lo x4] IEEE halves (conv_igemm_impl.h chunk_hl), written by the
//     producing convolution's epilogue as a second copy of the trunk (FTC_OP_CONV out2 with FTC_FLAG_SPLIT16), exactly like the weights.  A
//     fragment of either operand then is two 16-byte LDS reads + register renaming: splitting x while it is consumed would cost ~35 VALU
//     instructions per fragment that feeds only three MFMAs (the depthwise-free GEMM kernels amortise it over 2-6 channel tiles);
//   * a K step of 32 channels is 128 bytes per row: a stage of the whole 576-pixel map would be 80 KB.  The map streams in two HALVES of 288
//     pixels per K step (36 KB each, three ring slots), the 64 weight rows of the step (8 KB) in a ring of their own: 132 KB, aliased by the image;
//   * waves = 4 channel tiles (16) x 2 pixel groups (9 blocks of 16 of the current half): 2 x 9 accumulator tiles per wave, 27 MFMAs
//     (v_mfma_f32_16x16x32_f16: Ah.Bl, Al.Bh, Ah.Bh) per sub-step against 2 + 18 fragment reads;
//   * 128-byte rows: chunk slot s of row r holds K chunk s ^ g(r), g(r) = ((r >> 1) & 7) ^ 2 (((r >> 2) ^ (r >> 3)) & 1) -- the 16-lane groups
//     in which the LDS serves a ds_read_b128 ({0-3, 12-15, 20-27}, ..) find their 16 rows x {chunk c, chunk c ^ 2} in 16 different bank quads.
// Activations: expf SiLU (the fp32 plans' epilogues), depthwise in fp32 FMA as dwconv_strip_kernel<float>.
#include "conv_igemm_impl.h"

namespace {
using namespace convimpl;

struct MbxP {
    const void* x;          // [B][H*W][K] pre-split fp32 chunks
    const void* we;         // [C][K] pre-split fp32 chunks, K-major
    const float* be;        // [C] expand bias (folded BN)
    const float* wd;        // [9][C] depthwise weights
    const float* bd;        // [C] depthwise bias
    float* out;             // [B][H*W][C] fp32
    float* sums;            // [B][nb][C] channel sums of the output
    const float* w1;        // SE fc1 weight [S][C] (optional)
    float* hpart;           // [B][nb * C/64][S] (optional)
    int B, H, W, K, C, S;
    int R, nb;              // band mode: output rows per band, bands per image (R = H, nb = 1: the whole image)
    int presplit_out;       // FTC_FLAG_PRESPLIT: `out` stored as [hi x4 | lo x4] chunks (what the project convolution's three-MFMA product consumes)
    unsigned img_bytes;
    float inv_hw;
};

constexpr int X3_CC = FTC_MBHEAD_SLICE_F32;  // expanded channels per workgroup (64)
constexpr int X3_NT = 512;
constexpr int X3_HALF = 288;                 // pixels per sub-step (18 MFMA blocks of 16)
constexpr int X3_XS = X3_HALF * 128;         // bytes of an x ring slot (128-byte rows: K step 32)
constexpr int X3_WS = X3_CC * 128;           // bytes of a weight ring slot
constexpr int X3_NSTAGE = 3;
constexpr int X3_WRING = X3_NSTAGE * X3_XS;  // the weight ring sits behind the x ring
constexpr int X3_MAXSLOT = 601;
constexpr int X3_PITCH = X3_CC * 4 + 8;      // 264
constexpr int X3_CONST = (X3_MAXSLOT * X3_PITCH + 15) / 16 * 16;
constexpr int X3_LDS = X3_CONST + 10 * X3_CC * 4;
static_assert(X3_WRING + X3_NSTAGE * X3_WS <= X3_MAXSLOT * X3_PITCH && X3_LDS <= 160 * 1024, "");
constexpr int X3_R = 6;

__device__ __forceinline__ int x3_g(int r) { return ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1); }
__device__ __forceinline__ f32x4 silu4(f32x4 v) { return f32x4{act_silu_precise(v[0]), act_silu_precise(v[1]), act_silu_precise(v[2]), act_silu_precise(v[3])}; }

template <bool FAST>
__global__ __launch_bounds__(X3_NT, 2) void mbconv_slice_x3_kernel(const MbxP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int bid = blockIdx.x;
    const int b = bid % p.B;                               // image b on XCD b % 8
    const int rest = bid / p.B;
    const int band = FAST ? 0 : rest % p.nb, sl = FAST ? rest : rest / p.nb;
    const int c0 = sl * X3_CC;
    const int W = FAST ? 24 : p.W;
    const int y0 = FAST ? 0 : band * p.R, y1 = FAST ? 24 : min(p.H, y0 + p.R);
    const int ylo = FAST ? 0 : max(0, y0 - 1), yhi = FAST ? 24 : min(p.H, y1 + 1);
    const int H = yhi - ylo;
    const int M = H * W;
    const int W1 = W + 1;
    const int Mfull = p.H * W;
    const int rowb = p.K * 4;

    const unsigned xbase = (unsigned)ylo * W * (unsigned)rowb;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.x) + (size_t)b * p.img_bytes + xbase), 0,
                                                                        p.img_bytes - xbase, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwe = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.we) + (size_t)c0 * rowb), 0,
                                                                         (unsigned)(X3_CC * rowb), 0x00020000);

    // DMA pieces (64 consecutive 16-byte LDS chunks = 8 rows each).  x half: 2304 chunks = 36 pieces, piece i*8 + wave: 5 for waves 0-3, 4 for
    // waves 4-7; weights: 512 chunks = 8 pieces, one per wave, with the first half of a K step only.
    constexpr int NXP = 5;
    int xo[NXP], xrow[NXP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int q = (i * 8 + wave) * 64 + lane;
        const int row = q >> 3;
        xrow[i] = row;
        xo[i] = row * rowb + (((q & 7) ^ x3_g(row)) << 4);
    }
    int wo;
    {
        const int q = wave * 64 + lane;
        const int row = q >> 3;
        wo = row * rowb + (((q & 7) ^ x3_g(row)) << 4);
    }
    auto issue_x = [&](int i, int u, int slot_off) {               // piece i of sub-step u = (K step u >> 1, half u & 1)
        if (i < 4 || wave < 4) {
            const int half = u & 1;
            const int row = xrow[i] + half * X3_HALF;
            lds_void_t* dst = (lds_void_t*)(smem_raw + slot_off + (i * 8 + wave) * 1024);
            glds16(rx, dst, row < M ? xo[i] : OOB, half * X3_HALF * rowb + (u >> 1) * 128);
        }
    };
    auto issue_w = [&](int k, int wslot_off) {
        lds_void_t* dst = (lds_void_t*)(smem_raw + wslot_off + wave * 1024);
        glds16(rwe, dst, wo, k * 128);
    };

    // waves: channel tile ct (16 channels) x pixel group pg (9 blocks of 16 of the current half)
    const int ct = wave & 3, pg = wave >> 2;
    f32x4 acc[2][9];
    {
        const f32x4 bi = *reinterpret_cast<const f32x4*>(p.be + c0 + ct * 16 + 4 * lq);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int j = 0; j < 9; ++j) acc[h2][j] = bi;
    }
    if (t < 10 * (X3_CC / 4)) {                                      // depthwise weights [9][64] + bias [64] of this slice -> LDS
        const int row = t >> 4, ch = (t & 15) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(row < 9 ? p.wd + (long)row * p.C + c0 + ch : p.bd + c0 + ch);
        *reinterpret_cast<f32x4*>(smem_raw + X3_CONST + t * 16) = v;
    }

    const int gl = x3_g(l15);
    const int sw0 = (((2 * lq) ^ gl) << 4), sw1 = (((2 * lq + 1) ^ gl) << 4);
    const int offA = (ct * 16 + l15) * 128;                          // + wslot
    const int offB = (pg * 144 + l15) * 128;                         // + xslot + j * 2048
    const int nk = p.K >> 5, nu = 2 * nk;
    // prologue: sub-steps 0 and 1 (both halves of K step 0)
    issue_w(0, X3_WRING);
#pragma unroll
    for (int i = 0; i < NXP; ++i) issue_x(i, 0, 0);
#pragma unroll
    for (int i = 0; i < NXP; ++i) issue_x(i, 1, X3_XS);
    int cur = 0, iss = 2 * X3_XS;                                    // x ring slots: consumed / issued into
    int wcur = X3_WRING, wiss = X3_WRING + X3_WS;
    // one sub-step = one half of the map x one K step; H2 compile-time so that the accumulator index is (the loop body is instantiated twice)
    auto substep = [&](auto h2c, int u) {
        constexpr int h2 = decltype(h2c)::value;
        // stage u has landed once at most the later-issued stage u + 1 remains outstanding (per-wave piece counts; a half-0 stage carries the weight piece)
        if (u + 1 >= nu) wait_vmcnt<0>();
        else if (h2 == 0) { if (wave < 4) wait_vmcnt<5>(); else wait_vmcnt<4>(); }
        else { if (wave < 4) wait_vmcnt<6>(); else wait_vmcnt<5>(); }
        wg_barrier();
        const unsigned char* xb = smem_raw + cur;
        const unsigned char* wb = smem_raw + wcur;
        const bool more = u + 2 < nu;
        f16x8 ah, al;
        frag_hl(*reinterpret_cast<const f32x4*>(wb + offA + sw0), *reinterpret_cast<const f32x4*>(wb + offA + sw1), ah, al);
        f32x4 q0[3], q1[3];                                          // pixel fragments three blocks ahead of their MFMAs
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            q0[j] = *reinterpret_cast<const f32x4*>(xb + offB + j * 2048 + sw0);
            q1[j] = *reinterpret_cast<const f32x4*>(xb + offB + j * 2048 + sw1);
        }
        // blocks in pairs (4 pairs + 1): the three MFMAs of a block form a dependent chain on its accumulator, two blocks interleaved
        // keep the matrix pipe from waiting for its own result
        constexpr int piece_after[9] = {0, 1, -1, 2, 3, -1, 4, 5, -1};
        auto pieces = [&](int j) {
            if (more && piece_after[j] >= 0) {
                if (piece_after[j] < NXP) issue_x(piece_after[j], u + 2, iss);
                else if (h2 == 0) issue_w((u + 2) >> 1, wiss);      // u + 2 is a half-0 sub-step too: the next K step's weight rows
            }
        };
#pragma unroll
        for (int j = 0; j < 9; j += 2) {
            const bool two = j + 1 < 9;
            f16x8 bh0, bl0, bh1, bl1;
            frag_hl(q0[j % 3], q1[j % 3], bh0, bl0);
            if (two) frag_hl(q0[(j + 1) % 3], q1[(j + 1) % 3], bh1, bl1);
            if (j + 3 < 9) {
                q0[j % 3] = *reinterpret_cast<const f32x4*>(xb + offB + (j + 3) * 2048 + sw0);
                q1[j % 3] = *reinterpret_cast<const f32x4*>(xb + offB + (j + 3) * 2048 + sw1);
            }
            if (two && j + 4 < 9) {
                q0[(j + 1) % 3] = *reinterpret_cast<const f32x4*>(xb + offB + (j + 4) * 2048 + sw0);
                q1[(j + 1) % 3] = *reinterpret_cast<const f32x4*>(xb + offB + (j + 4) * 2048 + sw1);
            }
            f32x4 c0_ = acc[h2][j], c1_ = two ? acc[h2][j + 1] : acc[h2][j];
            c0_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl0, c0_, 0, 0, 0);
            if (two) c1_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl1, c1_, 0, 0, 0);
            c0_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh0, c0_, 0, 0, 0);
            if (two) c1_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh1, c1_, 0, 0, 0);
            c0_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, c0_, 0, 0, 0);
            if (two) c1_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, c1_, 0, 0, 0);
            acc[h2][j] = c0_;
            if (two) acc[h2][j + 1] = c1_;
            // the DMA pieces of sub-step u + 2 between the MFMA groups (issued together behind the barrier they stall every wave at once)
            pieces(j);
            if (two) pieces(j + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        iss = iss + X3_XS == X3_NSTAGE * X3_XS ? 0 : iss + X3_XS;
        cur = cur + X3_XS == X3_NSTAGE * X3_XS ? 0 : cur + X3_XS;
        if (h2 == 0) wiss = wiss + X3_WS == X3_WRING + X3_NSTAGE * X3_WS ? X3_WRING : wiss + X3_WS;
        else wcur = wcur + X3_WS == X3_WRING + X3_NSTAGE * X3_WS ? X3_WRING : wcur + X3_WS;
    };
    for (int k = 0; k < nk; ++k) {
        substep(std::integral_constant<int, 0>{}, 2 * k);
        substep(std::integral_constant<int, 1>{}, 2 * k + 1);
    }
    wg_barrier();                                               // every wave is done with the rings: they become the expanded image

    // ---- expanded image: SiLU, fp32, slot(y, x) = y (W+1) + x + 1 ----
    for (int idx = t; idx < (H + 1) * 32; idx += X3_NT) {       // the zero slots between the rows (and before the first / after the last)
        const u32x2 z = {0u, 0u};
        *reinterpret_cast<u32x2*>(smem_raw + ((idx >> 5) * W1) * X3_PITCH + (idx & 31) * 8) = z;
    }
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int m = h2 * X3_HALF + pg * 144 + j * 16 + l15;
            if (m < M) {
                const int y = m / W;
                unsigned char* row = smem_raw + (m + y + 1) * X3_PITCH + (ct * 16 + 4 * lq) * 4;
                const u32x4 e = __builtin_bit_cast(u32x4, silu4(acc[h2][j]));
                *reinterpret_cast<u32x2*>(row) = u32x2{e[0], e[1]};                                         // (264-byte rows are 8-byte aligned only)
                *reinterpret_cast<u32x2*>(row + 8) = u32x2{e[2], e[3]};
            }
        }
    __syncthreads();

    // ---- depthwise 3x3 + bias + SiLU + channel sums: lane = 4 channels x a vertical strip of 6 outputs ----
    const int cq = t & 15, pl = t >> 4;                         // 4 channels cq*4.., strip lane 0..31
    const int c = c0 + cq * 4;
    constexpr int NU = FTC_MBHEAD_MAX_SQUEEZE / 32;             // S <= 160
    f32x4 w1r[NU];
    if (p.hpart) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int su = pl + 32 * i;
            w1r[i] = su < p.S ? *reinterpret_cast<const f32x4*>(p.w1 + (size_t)su * p.C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(smem_raw + X3_CONST + k * (X3_CC * 4) + cq * 16);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(smem_raw + X3_CONST + 9 * (X3_CC * 4) + cq * 16);
    const int yo = y0 - ylo, yend = y1 - ylo;
    const int nsr = (yend - yo + X3_R - 1) / X3_R;
    const int nstrips = nsr * W;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    float* outp = p.out + ((size_t)b * Mfull + (size_t)ylo * W) * p.C + c;
    const unsigned char* zslot = smem_raw + cq * 16;            // slot 0: zeros
    auto ld = [](const unsigned char* q) {                       // 16 bytes at an 8-byte aligned LDS address
        const u32x2 a = *reinterpret_cast<const u32x2*>(q), b2 = *reinterpret_cast<const u32x2*>(q + 8);
        return __builtin_bit_cast(f32x4, u32x4{a[0], a[1], b2[0], b2[1]});
    };
    for (int s = pl; s < nstrips; s += 32) {
        const int sr = s / W, x = s - sr * W;
        const int oy0 = yo + sr * X3_R;
        const unsigned char* base0 = smem_raw + ((oy0 - 1) * W1 + x) * X3_PITCH + cq * 16;      // slot of (oy0 - 1, x - 1)
        const unsigned char* base1 = base0 + 7 * W1 * X3_PITCH;
        f32x4 a[X3_R];
#pragma unroll
        for (int oo = 0; oo < X3_R; ++oo) a[oo] = bv;
#pragma unroll
        for (int r = 0; r < X3_R + 2; ++r) {
            bool rok;
            if (FAST) rok = r == 0 ? oy0 > 0 : r == X3_R + 1 ? oy0 + X3_R < 24 : true;
            else rok = (unsigned)(oy0 - 1 + r) < (unsigned)H;
            const unsigned char* rp = (r < 7 ? base0 + r * W1 * X3_PITCH : base1 + (r - 7) * W1 * X3_PITCH);
            f32x4 xin[3];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) xin[s2] = ld(rok ? rp + s2 * X3_PITCH : zslot);
#pragma unroll
            for (int oo = 0; oo < X3_R; ++oo) {
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
        for (int oo = 0; oo < X3_R; ++oo) {
            const int oy = oy0 + oo;
            if (FAST || oy < yend) {
                a[oo] = silu4(a[oo]);
                if (p.presplit_out) *reinterpret_cast<u32x4*>(outp + (size_t)(oy * W + x) * p.C) = chunk_hl(a[oo]);
                else *reinterpret_cast<f32x4*>(outp + (size_t)(oy * W + x) * p.C) = a[oo];
                sum += a[oo];
            }
        }
    }

    // ---- squeeze: the channel sums of the rows this workgroup owns ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);           // [8 waves][64]
    float* lmean = red + 8 * X3_CC;
    {
        f32x4 v = sum;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += __shfl_xor(v[e], 16, 64); v[e] += __shfl_xor(v[e], 32, 64); }
        if (lane < 16) *reinterpret_cast<f32x4*>(red + wave * X3_CC + lane * 4) = v;
    }
    __syncthreads();
    if (t < X3_CC) {
        float tot = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) tot += red[w8 * X3_CC + t];
        p.sums[((size_t)b * p.nb + band) * p.C + c0 + t] = tot;
        lmean[t] = tot * p.inv_hw;
    }
    if (p.hpart) {
        __syncthreads();
        float* fcb = lmean + X3_CC;                             // [unit][16 + 1]
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(lmean + cq * 4);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const f32x4 pr = w1r[i] * m4;
            const int su = pl + 32 * i;
            if (su < p.S) fcb[su * 17 + cq] = (pr[0] + pr[1]) + (pr[2] + pr[3]);
        }
        __syncthreads();
        if (t < p.S) {
            float d = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) d += fcb[t * 17 + q];
            p.hpart[(((size_t)b * p.nb + band) * (p.C / X3_CC) + sl) * p.S + t] = d;
        }
    }
}

}  // namespace

hipError_t launch_mbhead_x3(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    MbxP p;
    p.x = a.in; p.we = a.w2; p.be = a.bias2; p.wd = static_cast<const float*>(a.w); p.bd = a.bias; p.out = static_cast<float*>(a.out); p.sums = a.aux;
    p.w1 = a.scale; p.hpart = a.scale ? static_cast<float*>(a.out2) : nullptr;
    p.B = o.B; p.H = o.H; p.W = o.W; p.K = o.Cin; p.C = o.Cout; p.S = o.aux0;
    p.R = o.aux1 > 0 ? o.aux1 : o.H; p.nb = ftc_mbhead_bands(o);
    p.presplit_out = (o.flags & FTC_FLAG_PRESPLIT) ? 1 : 0;
    p.img_bytes = (unsigned)((long)o.H * o.W * o.Cin * 4);
    p.inv_hw = 1.0f / (float)(o.H * o.W);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mbconv_slice_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mbconv_slice_x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nblk = o.B * p.nb * (o.Cout / X3_CC);
    const bool fast = o.H == 24 && o.W == 24 && p.nb == 1 && !(o.flags & 0x100);
    if (fast) hipLaunchKernelGGL(mbconv_slice_x3_kernel<true>, dim3(nblk), dim3(X3_NT), X3_LDS, s, p);
    else hipLaunchKernelGGL(mbconv_slice_x3_kernel<false>, dim3(nblk), dim3(X3_NT), X3_LDS, s, p);
    return hipGetLastError();
}
