// HBM-bound backbone kernels: stem conv, LDS-tiled depthwise 3x3 (+SE partial sums) and the SE
// excitation (wave-shuffle reductions).  NHWC throughout, 4 channels (16 B fp32 / 8 B bf16) per lane.
//
// Reference ops: torchvision Conv2dNormActivation / MBConv / SqueezeExcitation as instantiated by
// /root/reference/models/detector.py:12-28 (structure restated in SURVEY.md Appendix D), and the
// input scaling of CenterNetDetection.forward (/root/reference/models/detector.py:218).
#include "ftc_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// Stem: y = SiLU(conv3x3_s2(x*2-1, W') + b'), W'/b' = BN-folded.  Cin = 3, so K = 27: direct
// convolution on the VALU; lanes = (pixel, channel quad) so a wave writes 1 KiB contiguous.
// ------------------------------------------------------------------------------------------
// NQ = channel quads per lane: with 4 (C0 % 16 == 0) a lane loads the 27 input values once for 16 output channels instead of once per
// quad (the kernel is load-instruction bound: 8 lanes per pixel re-read the same taps through L1).  FAST = the exp2/rcp SiLU of the 16-bit
// modes (3e-7 relative); the fp32 parity mode keeps libm's expf.  The per-channel order of the 27 multiply-adds is the same in all forms.
template <typename OutT, typename CopyT, int NQ, bool FAST>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, OutT* __restrict__ out, CopyT* __restrict__ out2,
                                                   int B, int H, int W, int Ho, int Wo, int C0, int nchw, int act) {
    extern __shared__ __attribute__((aligned(16))) float sw[];   // [27][C0]
    for (int i = threadIdx.x; i < 27 * C0; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    const int CG = C0 / (4 * NQ);
    const unsigned total = (unsigned)B * Ho * Wo * CG;          // 32-bit index arithmetic (validated)
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned pix = idx / (unsigned)CG;
        const int c0 = (int)(idx - pix * CG) * 4 * NQ;
        const unsigned row = pix / (unsigned)Wo;
        const int ox = (int)(pix - row * Wo);
        const int b = (int)(row / (unsigned)Ho);
        const int oy = (int)(row - (unsigned)b * Ho);
        f32x4 acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = *reinterpret_cast<const f32x4*>(bias + c0 + 4 * q);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * 2 - 1 + r;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = ox * 2 - 1 + s;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = nchw ? in[(((long)b * 3 + c) * H + iy) * W + ix]
                                             : in[(((long)b * H + iy) * W + ix) * 3 + c];
                        const float xv = v * 2.0f - 1.0f;                       // detector.py:218
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[q] += xv * *reinterpret_cast<const f32x4*>(sw + ((r * 3 + s) * 3 + c) * C0 + c0 + 4 * q);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (act) {                                      // (act 0: the raw convolution, for the training-mode forward)
                if constexpr (FAST) acc[q] = act_silu_fast4(acc[q]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q][e] = act_silu_precise(acc[q][e]);
                }
            }
            store4<OutT>(out + (long)pix * C0 + c0 + 4 * q, acc[q]);
            if (out2) store4<CopyT>(out2 + (long)pix * C0 + c0 + 4 * q, acc[q]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Depthwise 3x3 (stride 1|2, pad 1) + folded BN + SiLU, and per-tile channel sums of the
// OUTPUT for the SE squeeze (deterministic: one partial per (image, tile, channel), reduced in a
// fixed order by se_kernel).
// Workgroup = 64 channels x (TH x TW) output pixels; the input halo tile is staged once in LDS
// (16 lanes x 16 B = one 256-B pixel row, coalesced), then every output reads its 9 taps from LDS.
// ------------------------------------------------------------------------------------------
template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, T* __restrict__ out,
                                                     float* __restrict__ partial, int H, int W, int Ho, int Wo,
                                                     int C, int tilesX, int P, int tiles_per_wg, int act) {
    constexpr int V = 16 / (int)sizeof(T);      // channels per lane = one 16-byte access (4 fp32 | 8 bf16)
    constexpr int LPP = 64 / V;                 // lanes per pixel of the 64-channel slab (16 | 8)
    constexpr int SLOTS = 256 / LPP;            // pixel slots per workgroup (16 | 32)
    constexpr int TH = STRIDE == 1 ? 8 : 4;
    constexpr int TW = 8;
    constexpr int IH = (TH - 1) * STRIDE + 3;
    constexpr int IW = (TW - 1) * STRIDE + 3;
    constexpr int NPX = IH * IW;
    constexpr int NLD = (NPX + SLOTS - 1) / SLOTS;                       // halo pixels fetched per lane per tile
    __shared__ __attribute__((aligned(16))) T tile[2][NPX * 64];         // double-buffered halo tile (storage dtype;
                                                                         // widening it to fp32 at staging measured slower)
    __shared__ __attribute__((aligned(16))) float red[SLOTS * 64];

    // Element index of halo pixel i.  16-bit, stride 2: a 16-lane group of a tap read covers two output pixels whose input pixels are
    // 2 apart = 256 B = the same 32 banks twice (SQ_LDS_BANK_CONFLICT 32.5 %, profiles/r03b); swapping the two 128-byte halves of
    // every second pixel PAIR puts them on the other 32 banks.
    auto swz = [](int i) { return (STRIDE == 2 && sizeof(T) == 2) ? (i * 64) ^ (((i >> 1) & 1) << 6) : i * 64; };
    const int t = threadIdx.x;
    const int cq = t % LPP;         // channel group inside the 64-channel slab
    const int pt = t / LPP;         // pixel slot
    const int c = blockIdx.x * 64 + cq * V;
    const int b = blockIdx.z;
    const bool cok = c < C;         // C % V == 0 (validated), so a lane is entirely in or out
    const int tile0 = blockIdx.y * tiles_per_wg;
    const int ntile = min(tiles_per_wg, P - tile0);

    // The workgroup walks `tiles_per_wg` consecutive spatial tiles of its 64-channel slab: the halo of
    // tile k+1 is fetched into registers while tile k is convolved out of LDS, and the 9x(4|8) filter
    // taps are loaded once.
    u32x4 stage[NLD];
    auto fetch = [&](int tileId) {
        const int ty = tileId / tilesX, tx = tileId - ty * tilesX;
        const int iy0 = ty * TH * STRIDE - 1, ix0 = tx * TW * STRIDE - 1;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = pt + j * SLOTS;
            const int ry = i / IW, rx = i - ry * IW;
            const int iy = iy0 + ry, ix = ix0 + rx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (cok && i < NPX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const u32x4*>(in + (((long)b * H + iy) * W + ix) * C + c);
            stage[j] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = pt + j * SLOTS;
            if (i < NPX) *reinterpret_cast<u32x4*>(&tile[buf][swz(i) + cq * V]) = stage[j];
        }
    };

    fetch(tile0);
    float wv[9][V], bv[V];
#pragma unroll
    for (int e = 0; e < V; ++e) bv[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) wv[k][e] = 0.f;
    if (cok) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int q = 0; q < V / 4; ++q) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(w + (long)k * C + c + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) wv[k][4 * q + e] = x[e];
            }
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(bias + c + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[4 * q + e] = x[e];
        }
    }
    commit(0);
    __syncthreads();

    for (int kt = 0; kt < ntile; ++kt) {
        const int tileId = tile0 + kt;
        const int buf = kt & 1;
        if (kt + 1 < ntile) fetch(tileId + 1);                 // in flight during the convolution below
        const int ty = tileId / tilesX, tx = tileId - ty * tilesX;
        const int oy0 = ty * TH, ox0 = tx * TW;
        float sum[V];
#pragma unroll
        for (int e = 0; e < V; ++e) sum[e] = 0.f;
#pragma unroll
        for (int k = 0; k < TH * TW / SLOTS; ++k) {
            const int o = pt + SLOTS * k;
            const int ly = o / TW, lx = o - ly * TW;
            const int oy = oy0 + ly, ox = ox0 + lx;
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = bv[e];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    float x[V];
                    load16<T>(&tile[buf][swz((ly * STRIDE + r) * IW + lx * STRIDE + s) + cq * V], x);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = fmaf(wv[r * 3 + s][e], x[e], acc[e]);
                }
            if (cok && oy < Ho && ox < Wo) {
                if (act) {
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = sizeof(T) == 2 ? act_silu_fast(acc[e]) : act_silu_precise(acc[e]);
                }
                store16<T>(out + (((long)b * Ho + oy) * Wo + ox) * C + c, acc);
#pragma unroll
                for (int e = 0; e < V; ++e) sum[e] += acc[e];
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) red[pt * 64 + cq * V + e] = sum[e];
        if (kt + 1 < ntile) commit(buf ^ 1);                    // the other buffer was last read one iteration ago
        __syncthreads();
        if (t < 64) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) s += red[k * 64 + t];
            const int cc = blockIdx.x * 64 + t;
            if (cc < C) partial[((long)b * P + tileId) * C + cc] = s;
        }
        __syncthreads();                                         // red[] is reused by the next tile
    }
}

// ------------------------------------------------------------------------------------------
// bf16 stride-1 variant (every MBConv block but the first of stages 4 and 6).  Same tiling, LDS
// image, partial-sum contract and tap order as dwconv_kernel, but a lane owns 4 channels and a
// vertical strip of 4 outputs: the 6x3 input window is read and widened once for the 4 outputs
// (18 ds_read_b64 instead of 36), and the 36 filter taps + 16 accumulators fit in < 128 VGPRs,
// i.e. 4 waves per SIMD instead of the 2 the 8-channel kernel gets at 196 VGPRs.  The kernel is
// VALU-bound (9 FMAs + widen + SiLU per output) before it is HBM-bound, so occupancy and the
// shared window are what move it.
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_ptr_t;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_ptr_t* dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, 0, 0, 0);     // LDS[dst + lane*16] = 16 bytes at r[voff]; zeros when voff is out of range
}

// (round 3: also for fp32 tensors -- the fp32 / fp16x3 parity modes and the train step's forward: a 16-byte DMA lane then carries 4 channels,
//  16 lanes per pixel, 7 passes per tile, a 51 KB LDS image; PRECISE = libm expf as the fp32 kernels use; act 0 = none (train mode: BatchNorm
//  follows as its own op))
template <typename T, bool PRECISE>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ out,
                                                           float* __restrict__ partial, int H, int W, int C, int tilesX,
                                                           int P, int tiles_per_wg, int act) {
    constexpr int TH = 8, TW = 8, IW = 10, NPX = 100;
    constexpr int CHS = 16 / (int)sizeof(T);                             // channels per 16-byte DMA lane: 8 | 4
    constexpr int LPS = 64 / CHS;                                        // lanes per pixel: 8 | 16
    constexpr int PPW = 64 / LPS;                                        // pixels per wave-level DMA: 8 | 4
    constexpr int PPP = 4 * PPW;                                         // pixels per pass of the four waves: 32 | 16
    constexpr int NLD = (NPX + PPP - 1) / PPP;                           // DMA passes: 4 | 7
    constexpr int OOB = 0x7ffffff0;
    __shared__ __attribute__((aligned(16))) T tile[2][NLD * PPP * 64];
    __shared__ __attribute__((aligned(16))) float red[16 * 64];

    const int t = threadIdx.x;
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * 64;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in + (long)b * H * W * C), 0,
                                                                         H * W * C * (int)sizeof(T), 0x00020000);
    // staging role: pixel wave*PPW + lane/LPS of each pass, CHS channels (16 B)
    const int s_px = wave * PPW + lane / LPS;
    const int s_coff = (c0 + (lane % LPS) * CHS < C) ? (c0 + (lane % LPS) * CHS) * (int)sizeof(T) : OOB;
    // compute role: 4 channels, column t/16 % 8, rows (t/128)*4 .. +3
    const int cq = t & 15, col = (t >> 4) & 7, rh = t >> 7;
    const int c = c0 + cq * 4;
    const bool cok = c < C;
    const int tile0 = blockIdx.y * tiles_per_wg;
    const int ntile = min(tiles_per_wg, P - tile0);

    auto issue = [&](int tileId, int buf) {
        const int ty = tileId / tilesX, tx = tileId - ty * tilesX;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = s_px + j * PPP;
            const int ry = i / IW, rx = i - ry * IW;
            const int iy = iy0 + ry, ix = ix0 + rx;
            const bool ok = i < NPX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            dma16(rin, (lds_ptr_t*)&tile[buf][(j * PPP + wave * PPW) * 64], ok ? (iy * W + ix) * C * (int)sizeof(T) + s_coff : OOB);
        }
    };

    issue(tile0, 0);
    f32x4 wv[9], bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = bv;
    if (cok) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(w + (long)k * C + c);
        bv = *reinterpret_cast<const f32x4*>(bias + c);
    }

    // Channel sums of the workgroup's outputs: accumulated in registers over all its tiles and reduced once;
    // the total goes to the slot of the first tile, the other slots get 0 (se_fc1 sums the P slots of an image).
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < ntile; ++kt) {
        const int tileId = tile0 + kt;
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this tile's halo has landed (and the taps, first time round)
        __syncthreads();                                                // ... for every wave; buf^1 is free again
        if (kt + 1 < ntile) issue(tileId + 1, buf ^ 1);
        const int ty = tileId / tilesX, tx = tileId - ty * tilesX;
        const int oy0 = ty * TH + rh * 4, ox = tx * TW + col;
        f32x4 acc[4] = {bv, bv, bv, bv};
        const T* tp = &tile[buf][((rh * 4) * IW + col) * 64 + cq * 4];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            f32x4 x[3];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) x[s2] = load4<T>(tp + (r * IW + s2) * 64);
#pragma unroll
            for (int oo = 0; oo < 4; ++oo) {
                const int kr = r - oo;
                if (kr >= 0 && kr < 3) {
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[oo][e] = fmaf(wv[kr * 3 + s2][e], x[s2][e], acc[oo][e]);
                }
            }
        }
#pragma unroll
        for (int oo = 0; oo < 4; ++oo) {
            const int oy = oy0 + oo;
            if (cok && oy < H && ox < W) {
                if (!PRECISE || act) {                                   // (16-bit tensors: always -- launch_dwconv sends act = none to the general kernel)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[oo][e] = PRECISE ? act_silu_precise(acc[oo][e]) : act_silu_fast(acc[oo][e]);
                }
                store4<T>(out + (((long)b * H + oy) * W + ox) * C + c, acc[oo]);
                sum += acc[oo];
            }
        }
    }
    *reinterpret_cast<f32x4*>(&red[(t >> 4) * 64 + cq * 4]) = sum;
    __syncthreads();
    if (t < 64 && c0 + t < C) {
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s2 += red[k * 64 + t];
        float* pp = partial + ((long)b * P + tile0) * C + c0 + t;
        pp[0] = s2;
        for (int kt = 1; kt < ntile; ++kt) pp[(long)kt * C] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// SE excitation: scale[b,c] = sigmoid(fc2(SiLU(fc1(mean_hw(x))))), as two short kernels that
// keep every load independent (the first version chained ~1200 dependent L2 round trips per
// wave and cost 200-450 us per layer):
//   se_fc1: grid (ceil(S/8), B), 8 waves: the workgroup reduces the P per-tile partial sums to
//           mean[C] in LDS (float4, fixed order -> deterministic), then each wave computes ONE
//           hidden unit as a float4 dot product + wave64 shuffle reduction.
//   se_fc2: grid (ceil(C/256), B): one lane per output channel, fc2 stored transposed [S][C]
//           so the S loads of a lane are coalesced across the wave and independent of each other.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void se_fc1_kernel(const float* __restrict__ partial, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, float* __restrict__ hidden,
                                                     int C, int S, int P, float inv_hw) {
    extern __shared__ __attribute__((aligned(16))) float mean[];   // [C]
    const int b = blockIdx.y;
    const int t = threadIdx.x;
    const int CQ = C >> 2;
    const int lane = t & 63, wave = t >> 6;
    const int sidx = blockIdx.x * 8 + wave;
    // This kernel is a chain of dependent memory round trips (partials -> LDS -> weights -> store), not bandwidth: the weight row
    // of this wave's hidden unit does not depend on the partial sums, so it is requested FIRST and is in flight while the sums are
    // reduced (C <= 3840: at most 15 float4 per lane).
    constexpr int WMAX = 15;
    f32x4 wv[WMAX];
    const f32x4* wr = reinterpret_cast<const f32x4*>(w1 + (long)(sidx < S ? sidx : 0) * C);
#pragma unroll
    for (int i = 0; i < WMAX; ++i) {
        const int q = lane + 64 * i;
        wv[i] = q < CQ ? wr[q] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4* pb = reinterpret_cast<const f32x4*>(partial + (long)b * P * C);
    for (int q = t; q < CQ; q += 512) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int p = 0; p < P; ++p) s += pb[(long)p * CQ + q];
        reinterpret_cast<f32x4*>(mean)[q] = s * inv_hw;
    }
    __syncthreads();
    if (sidx >= S) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WMAX; ++i) {
        const int q = lane + 64 * i;
        if (q < CQ) acc += wv[i] * reinterpret_cast<const f32x4*>(mean)[q];
    }
    for (int q = lane + 64 * WMAX; q < CQ; q += 64) acc += wr[q] * reinterpret_cast<const f32x4*>(mean)[q];     // (C > 3840: not in this network)
    float a = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane == 0) hidden[(long)b * S + sidx] = act_silu_precise(a + b1[sidx]);
}

// Hidden vector of the SE MLP into LDS: what se_fc1_kernel stored, or (FTC_FLAG_SE_HPART) SiLU(b1 + the C/64 per-slice partial products
// FTC_OP_MBHEAD wrote, added in slice order: deterministic) -- the fc1 launch and its re-read of the channel sums disappear.
__device__ __forceinline__ void se_load_hidden(float* hid, const float* __restrict__ hidden, const float* __restrict__ hpart,
                                               const float* __restrict__ b1, int b, int S, int NS, int t, int nthreads) {
    if (hpart) {
        const float* hp = hpart + (long)b * NS * S;
        for (int s = t; s < S; s += nthreads) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int j = 0;
            for (; j + 4 <= NS; j += 4) {                      // four independent loads in flight; the association is fixed
                a0 += hp[(long)j * S + s];
                a1 += hp[(long)(j + 1) * S + s];
                a2 += hp[(long)(j + 2) * S + s];
                a3 += hp[(long)(j + 3) * S + s];
            }
            for (; j < NS; ++j) a0 += hp[(long)j * S + s];
            hid[s] = act_silu_precise(((a0 + a1) + (a2 + a3)) + b1[s]);
        }
    } else {
        for (int s = t; s < S; s += nthreads) hid[s] = hidden[(long)b * S + s];
    }
}

// FOLD: additionally write the project weights scaled by this image's excitation, wb[b][n][c] = bf16(wp[n][c] * scale[b,c])
// (FTC_FLAG_SE_FOLD).  The N rows are split over gridDim.z so that ~2 workgroups per CU share the copy; every z-slice
// recomputes its 256 scale values (S coalesced loads per lane), slice 0 stores them.
template <bool FOLD, typename T>
__global__ __launch_bounds__(256) void se_fc2_kernel(const float* __restrict__ hidden, const float* __restrict__ w2t,
                                                     const float* __restrict__ b2, float* __restrict__ scale, int C, int S,
                                                     const T* __restrict__ wp, T* __restrict__ wb, int N,
                                                     const float* __restrict__ hpart, const float* __restrict__ b1, int NS) {
    extern __shared__ __attribute__((aligned(16))) float hid[];    // [S] (+ [256] scale values when FOLD)
    const int b = blockIdx.y;
    se_load_hidden(hid, hidden, hpart, b1, b, S, NS, threadIdx.x, 256);
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    float sc = 0.f;
    if (c < C) {
        float acc = b2[c];
#pragma unroll 8
        for (int s = 0; s < S; ++s) acc += hid[s] * w2t[(long)s * C + c];
        sc = sigmoid_precise(acc);
        if (!FOLD || blockIdx.z == 0) scale[(long)b * C + c] = sc;
    }
    if constexpr (FOLD) {
        float* lsc = hid + S;
        lsc[threadIdx.x] = sc;
        __syncthreads();
        const int chunk = threadIdx.x & 31, r0 = threadIdx.x >> 5;
        const int cc = blockIdx.x * 256 + chunk * 8;
        if (cc >= C) return;                                       // C % 8 == 0 (validated)
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = lsc[chunk * 8 + e];
        const int rows = (N + gridDim.z - 1) / gridDim.z;
        const int n_lo = blockIdx.z * rows, n_hi = min(N, n_lo + rows);
        T* dst = wb + (long)b * N * C;
#pragma unroll 4
        for (int n = n_lo + r0; n < n_hi; n += 8) {
            float x[8];
            load16<T>(wp + (long)n * C + cc, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= f[e];
            store16<T>(dst + (long)n * C + cc, x);
        }
    }
}


// FOLD, second version (bf16 speed mode).  The first version spent ~5 us of each launch in its prologue: every lane walked the S
// rows of fc2 for ITS channel with only 8 loads in flight.  Here a workgroup owns 64 channels: the 256 lanes split S four ways
// (all of a lane's S/4 loads independent and unrolled), the four partial dots meet in LDS in fixed order, and the 64 scale values
// then stream the [N][64] column block of the project weights: 8 lanes x 16 B per row, 32 rows per pass, rows split over gridDim.z.
// Same arithmetic per element as se_fc2_kernel<true> except for the association of the S-sum (4 partial sums of S/4 terms).
template <typename T>
__global__ __launch_bounds__(256) void se_fc2_fold64_kernel(const float* __restrict__ hidden, const float* __restrict__ w2t,
                                                            const float* __restrict__ b2, float* __restrict__ scale, int C, int S,
                                                            const T* __restrict__ wp, T* __restrict__ wb, int N,
                                                            const float* __restrict__ hpart, const float* __restrict__ b1, int NS) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];      // [S] hidden | [4][64] partial dots | [64] scale
    float* hid = lds_f;
    float* part = lds_f + ((S + 3) & ~3);
    float* lsc = part + 256;
    const int b = blockIdx.y, t = threadIdx.x;
    const int cl = t & 63, sg = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int s_lo = sg * ((S + 3) / 4), s_hi = min(S, s_lo + (S + 3) / 4);
    // Dependent memory round trips, not bytes, bound this kernel (it takes ~9 us even when the fold writes 2 MB): the fc2 column of
    // this lane does not depend on the hidden vector, so its S/4 loads are issued before the hidden vector is even requested
    // (S <= 160: at most 40 per lane) and everything is in flight together.
    constexpr int SMAX = 40;
    float wreg[SMAX];
#pragma unroll
    for (int i = 0; i < SMAX; ++i) wreg[i] = (c < C && s_lo + i < s_hi) ? w2t[(long)(s_lo + i) * C + c] : 0.f;
    se_load_hidden(hid, hidden, hpart, b1, b, S, NS, t, 256);
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < SMAX; ++i)
        if (s_lo + i < s_hi) acc += hid[s_lo + i] * wreg[i];
    for (int s = s_lo + SMAX; s < s_hi; ++s) acc += hid[s] * w2t[(long)s * C + c];      // (S > 160: not in this network)
    part[sg * 64 + cl] = acc;
    __syncthreads();
    if (t < 64) {
        float sc = 0.f;
        if (c < C) {
            sc = sigmoid_precise(b2[c] + ((part[cl] + part[64 + cl]) + (part[128 + cl] + part[192 + cl])));
            if (blockIdx.z == 0) scale[(long)b * C + c] = sc;
        }
        lsc[cl] = sc;
    }
    __syncthreads();
    const int chunk = t & 7, r0 = t >> 3;                         // 8 lanes x 8 channels = the 64-channel block, 32 rows per pass
    const int cc = blockIdx.x * 64 + chunk * 8;
    if (cc >= C) return;                                           // C % 8 == 0 (validated)
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = lsc[chunk * 8 + e];
    const int rows = (N + gridDim.z - 1) / gridDim.z;
    const int n_lo = blockIdx.z * rows, n_hi = min(N, n_lo + rows);
    T* dst = wb + (long)b * N * C;
#pragma unroll 4
    for (int n = n_lo + r0; n < n_hi; n += 32) {
        float x[8];
        load16<T>(wp + (long)n * C + cc, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] *= f[e];
        store16<T>(dst + (long)n * C + cc, x);
    }
}

// FOLD for the fp16x3 mode (FTC_FLAG_SPLIT16; the fp32 plan): the project weights are stored PRE-SPLIT -- every 16-byte chunk of four
// fp32 weights as [hi x4 | lo x4] IEEE halves (model.hip add_compute) -- so the folded per-image copy is too: w = hi + lo (exact in fp32),
// w * scale, split again (same split as conv_igemm_impl.h chunk_hl).  With it the project convolution of the fp16x3 plan streams both
// operands by DMA like the 16-bit plans do, instead of gating every activation element while staging it (87 us per stage-6 block).
// Same prologue as se_fc2_fold64_kernel (64 channels per workgroup); 16 lanes x one chunk = the 64 channels, 16 rows per pass.
__global__ __launch_bounds__(256) void se_fc2_foldx3_kernel(const float* __restrict__ hidden, const float* __restrict__ w2t,
                                                            const float* __restrict__ b2, float* __restrict__ scale, int C, int S,
                                                            const u32x4* __restrict__ wp, u32x4* __restrict__ wb, int N,
                                                            const float* __restrict__ hpart, const float* __restrict__ b1, int NS) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];      // [S] hidden | [4][64] partial dots | [64] scale
    float* hid = lds_f;
    float* part = lds_f + ((S + 3) & ~3);
    float* lsc = part + 256;
    const int b = blockIdx.y, t = threadIdx.x;
    const int cl = t & 63, sg = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int s_lo = sg * ((S + 3) / 4), s_hi = min(S, s_lo + (S + 3) / 4);
    constexpr int SMAX = 40;
    float wreg[SMAX];
#pragma unroll
    for (int i = 0; i < SMAX; ++i) wreg[i] = (c < C && s_lo + i < s_hi) ? w2t[(long)(s_lo + i) * C + c] : 0.f;
    se_load_hidden(hid, hidden, hpart, b1, b, S, NS, t, 256);          // (round 5: FTC_FLAG_SE_HPART in the fp16x3 plan too -- the fused head's partial products)
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < SMAX; ++i)
        if (s_lo + i < s_hi) acc += hid[s_lo + i] * wreg[i];
    for (int s = s_lo + SMAX; s < s_hi; ++s) acc += hid[s] * w2t[(long)s * C + c];
    part[sg * 64 + cl] = acc;
    __syncthreads();
    if (t < 64) {
        float sc = 0.f;
        if (c < C) {
            sc = sigmoid_precise(b2[c] + ((part[cl] + part[64 + cl]) + (part[128 + cl] + part[192 + cl])));
            if (blockIdx.z == 0) scale[(long)b * C + c] = sc;
        }
        lsc[cl] = sc;
    }
    __syncthreads();
    const int chunk = t & 15, r0 = t >> 4;
    const int cc = blockIdx.x * 64 + chunk * 4;
    if (cc >= C) return;                                           // C % 4 == 0 (validated)
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = lsc[chunk * 4 + e];
    const int rows = (N + gridDim.z - 1) / gridDim.z;
    const int n_lo = blockIdx.z * rows, n_hi = min(N, n_lo + rows);
    u32x4* dst = wb + (long)b * N * (C >> 2);
#pragma unroll 4
    for (int n = n_lo + r0; n < n_hi; n += 16) {
        const u32x4 v = wp[(long)n * (C >> 2) + (cc >> 2)];
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const u32x2 hu = {v[0], v[1]}, lu = {v[2], v[3]};
        const h4 hi = __builtin_bit_cast(h4, hu), lo = __builtin_bit_cast(h4, lu);
        h4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = ((float)hi[e] + (float)lo[e]) * f[e];
            const _Float16 hh = (_Float16)f16_sat(x);
            oh[e] = hh;
            ol[e] = (_Float16)(x - (float)hh);
        }
        const u32x2 ohu = __builtin_bit_cast(u32x2, oh), olu = __builtin_bit_cast(u32x2, ol);
        dst[(long)n * (C >> 2) + (cc >> 2)] = u32x4{ohu[0], ohu[1], olu[0], olu[1]};
    }
}

}  // namespace

hipError_t launch_stem(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const bool wide = o.Cout % 16 == 0;
    const long total = (long)o.B * o.Ho * o.Wo * (o.Cout / (wide ? 16 : 4));
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const size_t lds = (size_t)27 * o.Cout * sizeof(float);
    const int nchw = (o.flags & FTC_FLAG_IN_NCHW) ? 1 : 0;
    const int act = o.act != FTC_ACT_NONE ? 1 : 0;
#define STEM_LAUNCH(OT, CT, NQ, FAST)                                                                                                        \
    hipLaunchKernelGGL((stem_kernel<OT, CT, NQ, FAST>), dim3(blocks), dim3(256), lds, s, (const float*)a.in, (const float*)a.w, a.bias, (OT*)a.out, \
                       (CT*)a.out2, o.B, o.H, o.W, o.Ho, o.Wo, o.Cout, nchw, act)
#define STEM_WIDTH(OT, CT, FAST) do { if (wide) STEM_LAUNCH(OT, CT, 4, FAST); else STEM_LAUNCH(OT, CT, 1, FAST); } while (0)
    // fp32 trunk output with an optional 16-bit copy in the plan's compute type (ftc_op.w_dtype: bf16 unless FTC_F16); a plan without
    // a 16-bit copy or type is the fp32 parity mode (or the training-mode forward): libm's expf
    if (o.out_dtype == FTC_F32 && o.w_dtype == FTC_F16) STEM_WIDTH(float, _Float16, true);
    else if (o.out_dtype == FTC_F32 && a.out2) STEM_WIDTH(float, __bf16, true);
    else if (o.out_dtype == FTC_F32) STEM_WIDTH(float, __bf16, false);
    else if (o.out_dtype == FTC_F16) STEM_WIDTH(_Float16, _Float16, true);
    else STEM_WIDTH(__bf16, __bf16, true);
#undef STEM_WIDTH
#undef STEM_LAUNCH
    return hipGetLastError();
}

hipError_t launch_dwconv(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int TH = o.stride == 1 ? 8 : 4, TW = 8;
    const int tilesX = (o.Wo + TW - 1) / TW, tilesY = (o.Ho + TH - 1) / TH;
    const int P = tilesX * tilesY;
    if (P != o.aux0) return hipErrorInvalidValue;
    // Consecutive tiles per workgroup: the grid runs in ceil(workgroups / resident slots) rounds of `tpw` tile
    // times each; take the tpw that minimises rounds * tpw (ties: the larger, it amortises the tap loads).
    // stride 1: the strip kernel (16-bit: fast SiLU only; fp32: with or without activation).  0x100: the general kernel (A/B measurements, tests)
    const bool strip = o.stride == 1 && !(o.flags & 0x100) && (ftc_is16(o.in_dtype) ? o.act != FTC_ACT_NONE : (o.Cin % 4 == 0));
    const long slabs = (long)((o.Cin + 63) / 64) * o.B;
    const long slots = 256L * (strip ? (o.in_dtype == FTC_F32 ? 2 : 4) : o.in_dtype == FTC_F32 ? 3 : 2);     // workgroups resident on 256 CUs (LDS / VGPR-limited)
    int tpw = 1;
    long best = -1;
    for (int cand = 1; cand <= (P < 16 ? P : 16); ++cand) {
        const long wgs = slabs * ((P + cand - 1) / cand);
        const long cost = ((wgs + slots - 1) / slots) * cand;
        if (best < 0 || cost <= best) { best = cost; tpw = cand; }
    }
    dim3 grid((o.Cin + 63) / 64, (P + tpw - 1) / tpw, o.B);
#define DW_LAUNCH(T, ST)                                                                                      \
    hipLaunchKernelGGL((dwconv_kernel<T, ST>), grid, dim3(256), 0, s, (const T*)a.in, (const float*)a.w, a.bias, \
                       (T*)a.out, a.aux, o.H, o.W, o.Ho, o.Wo, o.Cin, tilesX, P, tpw, o.act != FTC_ACT_NONE ? 1 : 0)
    const int act = o.act != FTC_ACT_NONE ? 1 : 0;
    if (strip && o.in_dtype == FTC_F32)
        hipLaunchKernelGGL((dwconv_strip_kernel<float, true>), grid, dim3(256), 0, s, (const float*)a.in, (const float*)a.w, a.bias,
                           (float*)a.out, a.aux, o.H, o.W, o.Cin, tilesX, P, tpw, act);
    else if (o.in_dtype == FTC_F32) { if (o.stride == 1) DW_LAUNCH(float, 1); else DW_LAUNCH(float, 2); }
    else if (strip && o.in_dtype == FTC_F16)
        hipLaunchKernelGGL((dwconv_strip_kernel<_Float16, false>), grid, dim3(256), 0, s, (const _Float16*)a.in, (const float*)a.w, a.bias,
                           (_Float16*)a.out, a.aux, o.H, o.W, o.Cin, tilesX, P, tpw, act);
    else if (strip)
        hipLaunchKernelGGL((dwconv_strip_kernel<__bf16, false>), grid, dim3(256), 0, s, (const __bf16*)a.in, (const float*)a.w, a.bias,
                           (__bf16*)a.out, a.aux, o.H, o.W, o.Cin, tilesX, P, tpw, act);
    else if (o.in_dtype == FTC_F16) { if (o.stride == 1) DW_LAUNCH(_Float16, 1); else DW_LAUNCH(_Float16, 2); }
    else { if (o.stride == 1) DW_LAUNCH(__bf16, 1); else DW_LAUNCH(__bf16, 2); }
#undef DW_LAUNCH
    return hipGetLastError();
}

hipError_t launch_se(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int C = o.Cin, S = o.aux0, P = o.aux1;
    float* hidden = const_cast<float*>(static_cast<const float*>(a.in2));      // [B,S] scratch
    // FTC_FLAG_SE_HPART: `aux` = the per-slice fc1 partial products of FTC_OP_MBHEAD, [B][P slices][S]: no fc1 launch
    const bool hp = (o.flags & FTC_FLAG_SE_HPART) != 0;
    const float* hpart = hp ? a.aux : nullptr;
    if (!hp) {
        hipLaunchKernelGGL(se_fc1_kernel, dim3((S + 7) / 8, o.B), dim3(512), (size_t)C * sizeof(float), s, (const float*)a.aux,
                           (const float*)a.w, a.bias, hidden, C, S, P, 1.0f / (float)(o.H * o.W));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (hp && (o.flags & FTC_FLAG_SE_FOLD) && (o.flags & 0x100)) return hipErrorInvalidValue;    // (validated: fold64 / foldx3 or no fold)
    if ((o.flags & FTC_FLAG_SE_FOLD) && o.w_dtype == FTC_F32) {    // fp16x3 plan: pre-split fp32 chunks in, pre-split per-image copies out
        const int cb = (C + 63) / 64;
        int nz = (768 + cb * o.B - 1) / (cb * o.B);
        const int max_nz = (o.Cout_total + 15) / 16;
        nz = nz < 1 ? 1 : nz > max_nz ? max_nz : nz;
        hipLaunchKernelGGL(se_fc2_foldx3_kernel, dim3(cb, o.B, nz), dim3(256), (size_t)(((S + 3) & ~3) + 256 + 64) * sizeof(float), s, hidden,
                           (const float*)a.w2, a.bias2, (float*)a.out, C, S, (const u32x4*)a.in, (u32x4*)a.out2, o.Cout_total, hpart, a.bias, P);
    } else if ((o.flags & FTC_FLAG_SE_FOLD) && !(o.flags & 0x100)) {
        const int cb = (C + 63) / 64;
        int nz = (768 + cb * o.B - 1) / (cb * o.B);                  // ~3 workgroups per CU
        const int max_nz = (o.Cout_total + 31) / 32;
        nz = nz < 1 ? 1 : nz > max_nz ? max_nz : nz;
        if (o.w_dtype == FTC_F16)
            hipLaunchKernelGGL(se_fc2_fold64_kernel<_Float16>, dim3(cb, o.B, nz), dim3(256), (size_t)(((S + 3) & ~3) + 256 + 64) * sizeof(float), s, hidden,
                               (const float*)a.w2, a.bias2, (float*)a.out, C, S, (const _Float16*)a.in, (_Float16*)a.out2, o.Cout_total, hpart, a.bias, P);
        else
            hipLaunchKernelGGL(se_fc2_fold64_kernel<__bf16>, dim3(cb, o.B, nz), dim3(256), (size_t)(((S + 3) & ~3) + 256 + 64) * sizeof(float), s, hidden,
                               (const float*)a.w2, a.bias2, (float*)a.out, C, S, (const __bf16*)a.in, (__bf16*)a.out2, o.Cout_total, hpart, a.bias, P);
    } else if (o.flags & FTC_FLAG_SE_FOLD) {                         // 0x100: the first version (kept for A/B measurements)
        const int cb = (C + 255) / 256;
        int nz = 512 / (cb * o.B);
        nz = nz < 1 ? 1 : nz > 8 ? 8 : nz;
        if (o.w_dtype == FTC_F16)
            hipLaunchKernelGGL((se_fc2_kernel<true, _Float16>), dim3(cb, o.B, nz), dim3(256), (size_t)(S + 256) * sizeof(float), s, hidden,
                               (const float*)a.w2, a.bias2, (float*)a.out, C, S, (const _Float16*)a.in, (_Float16*)a.out2, o.Cout_total, nullptr, nullptr, 0);
        else
            hipLaunchKernelGGL((se_fc2_kernel<true, __bf16>), dim3(cb, o.B, nz), dim3(256), (size_t)(S + 256) * sizeof(float), s, hidden,
                               (const float*)a.w2, a.bias2, (float*)a.out, C, S, (const __bf16*)a.in, (__bf16*)a.out2, o.Cout_total, nullptr, nullptr, 0);
    } else if (hp && C % 64 == 0 && S <= 160) {
        // gates only, from FTC_OP_MBHEAD's partial products (round 4: the project convolution applies them to its weight fragments): the
        // 64-channel kernel with an EMPTY fold -- every load of its prologue in flight at once (the plain kernel below walks S dependent loads)
        hipLaunchKernelGGL(se_fc2_fold64_kernel<__bf16>, dim3(C / 64, o.B, 1), dim3(256), (size_t)(((S + 3) & ~3) + 256 + 64) * sizeof(float), s, hidden,
                           (const float*)a.w2, a.bias2, (float*)a.out, C, S, (const __bf16*)nullptr, (__bf16*)nullptr, 0, hpart, a.bias, P);
    } else {
        hipLaunchKernelGGL((se_fc2_kernel<false, __bf16>), dim3((C + 255) / 256, o.B), dim3(256), (size_t)S * sizeof(float), s, hidden,
                           (const float*)a.w2, a.bias2, (float*)a.out, C, S, (const __bf16*)nullptr, (__bf16*)nullptr, 0, hpart, a.bias, P);
    }
    return hipGetLastError();
}
