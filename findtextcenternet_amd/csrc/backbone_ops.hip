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
template <typename OutT>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, OutT* __restrict__ out, __bf16* __restrict__ out2,
                                                   int B, int H, int W, int Ho, int Wo, int C0, int nchw) {
    extern __shared__ __attribute__((aligned(16))) float sw[];   // [27][C0]
    for (int i = threadIdx.x; i < 27 * C0; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    const int CQ = C0 >> 2;
    const long total = (long)B * Ho * Wo * CQ;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cq = (int)(idx % CQ);
        const long pix = idx / CQ;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const int b = (int)(pix / ((long)Wo * Ho));
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + cq * 4);
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
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(sw + ((r * 3 + s) * 3 + c) * C0 + cq * 4);
                        acc += xv * wv;
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = act_silu_precise(acc[e]);
        store4<OutT>(out + pix * C0 + cq * 4, acc);
        if (out2) store4<__bf16>(out2 + pix * C0 + cq * 4, acc);
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
                                                     int C, int tilesX, int P, int tiles_per_wg) {
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
            if (i < NPX) *reinterpret_cast<u32x4*>(&tile[buf][i * 64 + cq * V]) = stage[j];
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
                    load16<T>(&tile[buf][((ly * STRIDE + r) * IW + lx * STRIDE + s) * 64 + cq * V], x);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = fmaf(wv[r * 3 + s][e], x[e], acc[e]);
                }
            if (cok && oy < Ho && ox < Wo) {
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = sizeof(T) == 2 ? act_silu_fast(acc[e]) : act_silu_precise(acc[e]);
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
    const f32x4* pb = reinterpret_cast<const f32x4*>(partial + (long)b * P * C);
    for (int q = t; q < CQ; q += 512) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int p = 0; p < P; ++p) s += pb[(long)p * CQ + q];
        reinterpret_cast<f32x4*>(mean)[q] = s * inv_hw;
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    const int sidx = blockIdx.x * 8 + wave;
    if (sidx >= S) return;
    const f32x4* wr = reinterpret_cast<const f32x4*>(w1 + (long)sidx * C);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int q = lane; q < CQ; q += 64) acc += wr[q] * reinterpret_cast<const f32x4*>(mean)[q];
    float a = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane == 0) hidden[(long)b * S + sidx] = act_silu_precise(a + b1[sidx]);
}

__global__ __launch_bounds__(256) void se_fc2_kernel(const float* __restrict__ hidden, const float* __restrict__ w2t,
                                                     const float* __restrict__ b2, float* __restrict__ scale, int C, int S) {
    extern __shared__ __attribute__((aligned(16))) float hid[];    // [S]
    const int b = blockIdx.y;
    for (int s = threadIdx.x; s < S; s += 256) hid[s] = hidden[(long)b * S + s];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float acc = b2[c];
#pragma unroll 8
    for (int s = 0; s < S; ++s) acc += hid[s] * w2t[(long)s * C + c];
    scale[(long)b * C + c] = sigmoid_precise(acc);
}

}  // namespace

hipError_t launch_stem(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const long total = (long)o.B * o.Ho * o.Wo * (o.Cout / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    const size_t lds = (size_t)27 * o.Cout * sizeof(float);
    const int nchw = (o.flags & FTC_FLAG_IN_NCHW) ? 1 : 0;
    if (o.out_dtype == FTC_F32)
        hipLaunchKernelGGL(stem_kernel<float>, dim3(blocks), dim3(256), lds, s, (const float*)a.in, (const float*)a.w,
                           a.bias, (float*)a.out, (__bf16*)a.out2, o.B, o.H, o.W, o.Ho, o.Wo, o.Cout, nchw);
    else
        hipLaunchKernelGGL(stem_kernel<__bf16>, dim3(blocks), dim3(256), lds, s, (const float*)a.in, (const float*)a.w,
                           a.bias, (__bf16*)a.out, (__bf16*)nullptr, o.B, o.H, o.W, o.Ho, o.Wo, o.Cout, nchw);
    return hipGetLastError();
}

hipError_t launch_dwconv(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int TH = o.stride == 1 ? 8 : 4, TW = 8;
    const int tilesX = (o.Wo + TW - 1) / TW, tilesY = (o.Ho + TH - 1) / TH;
    const int P = tilesX * tilesY;
    if (P != o.aux0) return hipErrorInvalidValue;
    // consecutive tiles per workgroup: as many as still leave >= 4 workgroups per CU
    const long slabs = (long)((o.Cin + 63) / 64) * o.B;
    int tpw = (int)((slabs * P) / 1024);
    if (tpw < 1) tpw = 1;
    if (tpw > 8) tpw = 8;
    dim3 grid((o.Cin + 63) / 64, (P + tpw - 1) / tpw, o.B);
#define DW_LAUNCH(T, ST)                                                                                      \
    hipLaunchKernelGGL((dwconv_kernel<T, ST>), grid, dim3(256), 0, s, (const T*)a.in, (const float*)a.w, a.bias, \
                       (T*)a.out, a.aux, o.H, o.W, o.Ho, o.Wo, o.Cin, tilesX, P, tpw)
    if (o.in_dtype == FTC_F32) { if (o.stride == 1) DW_LAUNCH(float, 1); else DW_LAUNCH(float, 2); }
    else { if (o.stride == 1) DW_LAUNCH(__bf16, 1); else DW_LAUNCH(__bf16, 2); }
#undef DW_LAUNCH
    return hipGetLastError();
}

hipError_t launch_se(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int C = o.Cin, S = o.aux0, P = o.aux1;
    float* hidden = const_cast<float*>(static_cast<const float*>(a.in2));      // [B,S] scratch
    hipLaunchKernelGGL(se_fc1_kernel, dim3((S + 7) / 8, o.B), dim3(512), (size_t)C * sizeof(float), s, (const float*)a.aux,
                       (const float*)a.w, a.bias, hidden, C, S, P, 1.0f / (float)(o.H * o.W));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(se_fc2_kernel, dim3((C + 255) / 256, o.B), dim3(256), (size_t)S * sizeof(float), s, hidden,
                       (const float*)a.w2, a.bias2, (float*)a.out, C, S);
    return hipGetLastError();
}
