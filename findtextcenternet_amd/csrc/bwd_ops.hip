// Backward kernels of the train step (BASELINE configs[4]) other than the MFMA weight gradient (wgrad.hip): everything here is
// HBM- or latency-bound.  What the reference gets from autograd for
//   loss.backward()   /root/reference/train1.py:170-179 (train_step :125-131)
// through TextDetectorModel.forward (/root/reference/models/detector.py:262-268), the torchvision MBConv / FusedMBConv blocks,
// Leafmap (:148-201), SimpleDecoder (:232-254) and loss_function / CoVWeightingLoss (/root/reference/loss_func.py:94-177, 24-72).
//
// Conventions: activations fp32 NHWC; reductions are two-stage with float64 partials summed in a fixed order (deterministic, no
// floating-point atomics); parameter gradients are ADDED to the flat gradient buffer in the PyTorch parameter layout.
#include <cstring>

#include "ftc_common.h"

hipError_t launch_gather_rows(const float* feat, const int32_t* sel_index, const int32_t* count, long cap, int C, int Cpad, void* rows, int out_dtype,
                              hipStream_t s);

namespace {

// d act(t) / dt.  SiLU: s (1 + t (1 - s)); exact GELU: Phi(t) + t phi(t).
__device__ __forceinline__ float dact(float t, int act) {
    if (act == FTC_ACT_SILU) {
        const float s = 1.0f / (1.0f + expf(-t));
        return s * (1.0f + t * (1.0f - s));
    }
    if (act == FTC_ACT_GELU) return 0.5f * (1.0f + erff(t * 0.70710678118654752440f)) + t * 0.39894228040143267794f * expf(-0.5f * t * t);
    return 1.0f;
}

// ------------------------------------------------------------------------------------------------------------------------
// FTC_OP_BNBWD: batch-statistics BatchNorm + activation backward (torch.nn.BatchNorm2d in train(), native_batch_norm_backward).
// Both passes are HBM streams: a workgroup owns Q channel quads x (256 / Q) row lanes, every access is one 16-byte lane (Q is the
// largest power of two dividing C / 4, so no channel guards), the per-(image, channel) operands of the MBConv blocks are reloaded only
// when the row crosses into the next image.  FAST: the exp2 / rcp forms of the 16-bit modes; the fp32 parity mode keeps libm.
// ------------------------------------------------------------------------------------------------------------------------
struct BnBwdP {
    const float* gy; int gs, goff;      // incoming gradient, row stride / channel offset
    const void* z;                      // BN input [M][C]: fp32, or 16-bit in the plan's compute type (the template parameter ZT of the kernels)
    const float* ss;                    // [4][C] scale, shift, mean, invstd
    const float* keep;                  // [B] or null
    const float* ga;                    // [B][C] or null
    const float* gb;                    // [B][C] or null
    int HW; int M; int C; int act;
};

template <bool FAST> __device__ __forceinline__ f32x4 dact4(f32x4 t, int act) {
    f32x4 r;
    if constexpr (FAST) {
        if (act == FTC_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t[e]));
                r[e] = s * (1.0f + t[e] * (1.0f - s));
            }
            return r;
        }
        if (act == FTC_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                      // Phi(t) + t phi(t); erf by Abramowitz-Stegun 7.1.26, sharing exp(-t^2/2) with phi
                const float a = fabsf(t[e]), zz = a * 0.70710678118654752440f;
                const float k = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * zz);
                float poly = 1.061405429f;
                poly = poly * k - 1.453152027f;
                poly = poly * k + 1.421413741f;
                poly = poly * k - 0.284496736f;
                poly = poly * k + 0.254829592f;
                const float ex = __builtin_amdgcn_exp2f(-0.72134752044448170368f * t[e] * t[e]);
                const float erfa = 1.0f - poly * k * ex;
                r[e] = 0.5f + copysignf(0.5f * erfa, t[e]) + t[e] * 0.39894228040143267794f * ex;
            }
            return r;
        }
    } else {
        if (act != FTC_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = dact(t[e], act);
            return r;
        }
    }
    r = f32x4{1.f, 1.f, 1.f, 1.f};
    return r;
}

template <int Q, bool FAST, typename ZT>
__global__ __launch_bounds__(256) void bnbwd_partial_kernel(BnBwdP p, double* __restrict__ part, int nchunk) {
    constexpr int RL = 256 / Q;
    __shared__ double red[2][RL][Q * 4];
    const int t = threadIdx.x, cq = t % Q, rl = t / Q;
    const int c = (blockIdx.x * Q + cq) * 4;
    const int chunk = blockIdx.y;
    const int rows = (p.M + nchunk - 1) / nchunk;
    const int r0 = chunk * rows, r1 = min(p.M, r0 + rows);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.ss + c), sh = *reinterpret_cast<const f32x4*>(p.ss + p.C + c);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(p.ss + 2 * p.C + c), istd = *reinterpret_cast<const f32x4*>(p.ss + 3 * p.C + c);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    int bcur = -1;
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
    float kp = 1.0f;
    // four rows' loads (gradient + pre-activation) in flight before the first is consumed; same row order of the sums
    constexpr int U = 4;
    for (int r = r0 + rl; r < r1; r += RL * U) {
        f32x4 gv[U], zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * RL;
            const bool ok = rr < r1;
            gv[u] = ok ? *reinterpret_cast<const f32x4*>(p.gy + (long)rr * p.gs + p.goff + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            zv[u] = ok ? load4<ZT>(static_cast<const ZT*>(p.z) + (long)rr * p.C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * RL;
            if (rr >= r1) break;
            const int b = rr / p.HW;
            if (b != bcur) {
                bcur = b;
                if (p.ga) ga = *reinterpret_cast<const f32x4*>(p.ga + (long)b * p.C + c);
                if (p.gb) gb = *reinterpret_cast<const f32x4*>(p.gb + (long)b * p.C + c);
                if (p.keep) kp = p.keep[b];
            }
            const f32x4 g = (gv[u] * ga + gb) * kp;
            const f32x4 dt = g * dact4<FAST>(zv[u] * sc + sh, p.act);
            const f32x4 zh = (zv[u] - mean) * istd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] += (double)dt[e];
                s2[e] += (double)dt[e] * (double)zh[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][rl][cq * 4 + e] = s1[e];
        red[1][rl][cq * 4 + e] = s2[e];
    }
    __syncthreads();
    for (int i = t; i < 2 * Q * 4; i += 256) {
        const int which = i / (Q * 4), l = i - which * (Q * 4);
        double a = 0.0;
#pragma unroll 4
        for (int k = 0; k < RL; ++k) a += red[which][k][l];
        part[((long)chunk * 2 + which) * p.C + blockIdx.x * Q * 4 + l] = a;
    }
}

__global__ __launch_bounds__(256) void bnbwd_final_kernel(const double* __restrict__ part, float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                          float* __restrict__ coef, long M, int C, int nchunk) {
    __shared__ double red[2][16][16];                           // 16 channels x 16 chunk lanes (as bnstat_final_kernel)
    const int t = threadIdx.x, cl = t & 15, kl = t >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int k = kl; k < nchunk; k += 16) {
            s1 += part[((long)k * 2 + 0) * C + c];
            s2 += part[((long)k * 2 + 1) * C + c];
        }
    red[0][kl][cl] = s1;
    red[1][kl][cl] = s2;
    __syncthreads();
    if (t >= 16 || c >= C) return;
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        s1 += red[0][k][t];
        s2 += red[1][k][t];
    }
    if (gbeta) gbeta[c] += (float)s1;
    if (ggamma) ggamma[c] += (float)s2;
    coef[c] = (float)(s1 / (double)M);
    coef[C + c] = (float)(s2 / (double)M);
}

// out (fp32, optional) and / or out2 (16-bit copy in the plan's compute type, optional): the gradient of a convolution's output is only
// ever read by that convolution's data- and weight-gradient GEMMs, which narrow it anyway -- writing it once in 16 bits halves their
// operand streams and lets them run on the DMA-staged kernels
template <int Q, bool FAST, typename TC, typename ZT>
__global__ __launch_bounds__(256) void bnbwd_apply_kernel(BnBwdP p, const float* __restrict__ coef, float* __restrict__ out, TC* __restrict__ out2, int accum,
                                                          int nchunk) {
    constexpr int RL = 256 / Q;
    const int t = threadIdx.x, cq = t % Q, rl = t / Q;
    const int c = (blockIdx.x * Q + cq) * 4;
    const int chunk = blockIdx.y;
    const int rows = (p.M + nchunk - 1) / nchunk;
    const int r0 = chunk * rows, r1 = min(p.M, r0 + rows);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.ss + c), sh = *reinterpret_cast<const f32x4*>(p.ss + p.C + c);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(p.ss + 2 * p.C + c), istd = *reinterpret_cast<const f32x4*>(p.ss + 3 * p.C + c);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(coef + c), c2 = *reinterpret_cast<const f32x4*>(coef + p.C + c);
    int bcur = -1;
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
    float kp = 1.0f;
    // (round 6) U rows' loads (gradient, pre-activation and, when accumulating, the old output) in flight before the first is consumed: the plain loop
    // issued one row's loads and waited (40 us per launch at 2.8 TB/s, 356 launches on the step's critical stream); same arithmetic per element
    constexpr int U = 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int r0u = r0 + rl; r0u < r1; r0u += RL * U) {
        f32x4 gv[U], zvv[U], ov[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0u + u * RL;
            const bool ok = r < r1;
            const int rr = ok ? r : r0u;
            gv[u] = ok ? *reinterpret_cast<const f32x4*>(p.gy + (long)rr * p.gs + p.goff + c) : zero;
            zvv[u] = ok ? load4<ZT>(static_cast<const ZT*>(p.z) + (long)rr * p.C + c) : zero;
            ov[u] = (ok && out && accum) ? *reinterpret_cast<const f32x4*>(out + (long)rr * p.C + c) : zero;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0u + u * RL;
            if (r >= r1) break;
            const int b = r / p.HW;
            if (b != bcur) {
                bcur = b;
                if (p.ga) ga = *reinterpret_cast<const f32x4*>(p.ga + (long)b * p.C + c);
                if (p.gb) gb = *reinterpret_cast<const f32x4*>(p.gb + (long)b * p.C + c);
                if (p.keep) kp = p.keep[b];
            }
            const f32x4 g = (gv[u] * ga + gb) * kp;
            const f32x4 zv = zvv[u];
            const f32x4 dt = g * dact4<FAST>(zv * sc + sh, p.act);
            f32x4 o = sc * (dt - c1 - (zv - mean) * istd * c2);
            if (out) {
                float* op = out + (long)r * p.C + c;
                if (accum) o += ov[u];
                *reinterpret_cast<f32x4*>(op) = o;
            }
            if constexpr (sizeof(TC) == 2) {
                if (out2) store4<TC>(out2 + (long)r * p.C + c, o);
            }
        }
    }
}

inline int pick_quads(int C) {
    const int q = C / 4;
    for (int t : {64, 32, 16, 8, 4, 2}) if (q % t == 0) return t;
    return 1;
}

// ------------------------------------------------------------------------------------------------------------------------
// FTC_OP_DWBWD: depthwise 3x3 (pad 1, stride 1|2) backward.  32-bit index arithmetic throughout (validated: < 2^31 quads).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwbwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ out, int B, int H,
                                                         int W, int Ho, int Wo, int C, int stride) {
    const unsigned Q = (unsigned)C >> 2;
    const unsigned total = (unsigned)B * H * W * Q;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned pix = idx / Q;
        const int c = (int)(idx - pix * Q) * 4;
        const unsigned row = pix / (unsigned)W;
        const int ix = (int)(pix - row * W);
        const int b = (int)(row / (unsigned)H);
        const int iy = (int)(row - (unsigned)b * H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ty = iy + 1 - r;
            if (ty < 0 || (stride == 2 && (ty & 1))) continue;
            const int oy = stride == 2 ? ty >> 1 : ty;
            if (oy >= Ho) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int tx = ix + 1 - s;
                if (tx < 0 || (stride == 2 && (tx & 1))) continue;
                const int ox = stride == 2 ? tx >> 1 : tx;
                if (ox >= Wo) continue;
                const f32x4 g = *reinterpret_cast<const f32x4*>(dz + (((long)b * Ho + oy) * Wo + ox) * C + c);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (r * 3 + s) * C + c);
                acc += g * wv;
            }
        }
        *reinterpret_cast<f32x4*>(out + (long)pix * C + c) = acc;
    }
}

// (round 6) The same data gradient with a lane = 4 channels walking its pixels: the nine weight quads live in registers (the kernel above re-loads them
// for every pixel), the pixel coordinates advance by carries instead of three integer divisions per output, and two pixels' taps (18 loads) are in flight.
// Same taps in the same order: bit-identical.  68 us per launch at 1.6 TB/s before (80 launches on the step's critical stream).
template <int Q>
__global__ __launch_bounds__(256) void dwbwd_data_q_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ out, int B, int H, int W,
                                                           int Ho, int Wo, int C, int stride, int nchunk) {
    constexpr int RL = 256 / Q, U = 2;
    const int t = threadIdx.x, cq = t % Q, rl = t / Q;
    const int c = (blockIdx.x * Q + cq) * 4;
    const int M = B * H * W;
    const int rows = (M + nchunk - 1) / nchunk;
    const int r0 = blockIdx.y * rows, r1 = min(M, r0 + rows);
    f32x4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(w + k * C + c);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int p = r0 + rl;
    if (p >= r1) return;
    int row = p / W, ix = p - row * W, b = row / H, iy = row - b * H;
    for (; p < r1; p += RL * U) {
        f32x4 g[U][9];
        int pu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pu[u] = p + u * RL;
            const bool pok = pu[u] < r1;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ty = iy + 1 - r;
                const int oy = stride == 2 ? ty >> 1 : ty;
                const bool yok = pok && ty >= 0 && !(stride == 2 && (ty & 1)) && oy < Ho;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int tx = ix + 1 - s;
                    const int ox = stride == 2 ? tx >> 1 : tx;
                    const bool ok = yok && tx >= 0 && !(stride == 2 && (tx & 1)) && ox < Wo;
                    g[u][r * 3 + s] = ok ? *reinterpret_cast<const f32x4*>(dz + (((long)b * Ho + oy) * Wo + ox) * C + c) : zero;
                }
            }
            ix += RL;                                            // the next pixel of this lane (RL <= 256 pixels on: a few carries)
            while (ix >= W) { ix -= W; if (++iy == H) { iy = 0; ++b; } }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pu[u] >= r1) break;
            f32x4 acc = zero;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += g[u][k] * wv[k];
            *reinterpret_cast<f32x4*>(out + (long)pu[u] * C + c) = acc;
        }
    }
}

// Q channel quads x (256 / Q) pixel lanes; every lane keeps 9 taps x 4 channels of partial sums over its pixels of the chunk
template <int Q>
__global__ __launch_bounds__(256) void dwbwd_weight_partial_kernel(const float* __restrict__ x, const float* __restrict__ dz, double* __restrict__ part, int B,
                                                                   int H, int W, int Ho, int Wo, int C, int stride, int nchunk) {
    constexpr int RL = 256 / Q;
    __shared__ float red[RL][9][Q * 4];
    const int t = threadIdx.x, cq = t % Q, rl = t / Q;
    const int c = (blockIdx.x * Q + cq) * 4;
    const int chunk = blockIdx.y;
    const int M = B * Ho * Wo;
    const int rows = (M + nchunk - 1) / nchunk;
    const int r0 = chunk * rows, r1 = min(M, r0 + rows);
    f32x4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (round 6) U pixels per trip with all of their 10 loads in flight before the first product: the plain loop issued a pixel's ten 16-byte loads and
    // waited for them, one memory round trip per pixel and lane -- 142 us per launch for 113 MB (0.8 TB/s; profiles/r06b_train_bf16_b8_kernel_stats.txt).
    // Out-of-image taps read nothing and add 0; the sums are taken in the same pixel and tap order as before.
    constexpr int U = 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = r0 + rl; p0 < r1; p0 += RL * U) {
        f32x4 g[U], xv[U][9];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * RL;
            const bool pok = p < r1;
            const int pp = pok ? p : r0;
            const int row = pp / Wo;
            const int ox = pp - row * Wo;
            const int b = row / Ho;
            const int oy = row - b * Ho;
            g[u] = pok ? *reinterpret_cast<const f32x4*>(dz + (long)pp * C + c) : zero;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oy * stride + r - 1;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int ix = ox * stride + s - 1;
                    const bool ok = pok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    xv[u][r * 3 + s] = ok ? *reinterpret_cast<const f32x4*>(x + (((long)b * H + iy) * W + ix) * C + c) : zero;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] += g[u] * xv[u][k];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) *reinterpret_cast<f32x4*>(&red[rl][k][cq * 4]) = acc[k];
    __syncthreads();
    for (int i = t; i < 9 * Q * 4; i += 256) {
        const int k = i / (Q * 4), l = i - k * (Q * 4);
        double a = 0.0;
#pragma unroll 4
        for (int j = 0; j < RL; ++j) a += (double)red[j][k][l];
        part[((long)chunk * 9 + k) * C + blockIdx.x * Q * 4 + l] = a;
    }
}

__global__ __launch_bounds__(256) void dwbwd_weight_final_kernel(const double* __restrict__ part, float* __restrict__ gw, int C, int nchunk) {
    const int i = blockIdx.x * 256 + threadIdx.x;         // i = k * C + c
    if (i >= 9 * C) return;
    const int k = i / C, c = i - k * C;
    double s = 0.0;
    for (int j = 0; j < nchunk; ++j) s += part[((long)j * 9 + k) * C + c];
    gw[(long)c * 9 + k] += (float)s;                       // parameter layout [C][1][3][3]
}

// ------------------------------------------------------------------------------------------------------------------------
// FTC_OP_SEBWD: y*s with s = sigmoid(fc2(SiLU(fc1(mean_hw y)))).  Scratch layout (floats): ds [B][C] | du2 [B][C] | mean [B][C] |
// dmean/HW [B][C] | da1 [B][S] | h [B][S] | ds partial sums [B][SE_PCH][C].
// ------------------------------------------------------------------------------------------------------------------------
constexpr int SE_PCH = 32;

template <int Q>
__global__ __launch_bounds__(256) void sebwd_ds_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ dsp, int HW, int C, int pch) {
    constexpr int RL = 256 / Q;
    __shared__ float red[RL][Q * 4];
    const int t = threadIdx.x, cq = t % Q, rl = t / Q;
    const int c = (blockIdx.x * Q + cq) * 4, chunk = blockIdx.y, b = blockIdx.z;
    const int rows = (HW + pch - 1) / pch;
    const int r0 = chunk * rows, r1 = min(HW, r0 + rows);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // (round 6: four rows' loads in flight before the first product -- the plain loop waited for each pair; same order of the sums)
    constexpr int U = 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + rl; r < r1; r += RL * U) {
        f32x4 gv[U], yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * RL;
            const bool ok = rr < r1;
            const long i = ((long)b * HW + (ok ? rr : r)) * C + c;
            gv[u] = ok ? *reinterpret_cast<const f32x4*>(g + i) : zero;
            yv[u] = ok ? *reinterpret_cast<const f32x4*>(y + i) : zero;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u * RL < r1) acc += gv[u] * yv[u];
    }
    *reinterpret_cast<f32x4*>(&red[rl][cq * 4]) = acc;
    __syncthreads();
    for (int l = t; l < Q * 4; l += 256) {
        double a = 0.0;
#pragma unroll 4
        for (int j = 0; j < RL; ++j) a += (double)red[j][l];
        dsp[((long)b * pch + chunk) * C + blockIdx.x * Q * 4 + l] = (float)a;
    }
}

// (1) grid (ceil(C / 256), B): mean of y, total of the ds partial sums, d(pre-sigmoid)
__global__ __launch_bounds__(256) void sebwd_prep_kernel(const float* __restrict__ sums, const float* __restrict__ scale, float* __restrict__ scratch, int B, int C,
                                                         int S, int P, float inv_hw, int pch) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    float* o_ds = scratch;
    float* o_du2 = scratch + (long)B * C;
    float* o_mean = scratch + (long)2 * B * C;
    const float* dsp = scratch + (long)4 * B * C + (long)2 * B * S;
    float m = 0.f;
    for (int p = 0; p < P; ++p) m += sums[((long)b * P + p) * C + c];
    o_mean[(long)b * C + c] = m * inv_hw;
    double dsum = 0.0;
    for (int p = 0; p < pch; ++p) dsum += (double)dsp[((long)b * pch + p) * C + c];
    const float dsv = (float)dsum;
    o_ds[(long)b * C + c] = dsv;
    const float sv = scale[(long)b * C + c];
    o_du2[(long)b * C + c] = dsv * sv * (1.0f - sv);
}

// (2) grid (ceil(S / 4), B), one wave per hidden unit j: a1 = fc1 . mean + b1, dh = fc2^T . du2 -> da1, h
__global__ __launch_bounds__(256) void sebwd_hidden_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2t,
                                                           float* __restrict__ scratch, int B, int C, int S) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (j >= S) return;
    const float* du2 = scratch + (long)B * C + (long)b * C;
    const float* mean = scratch + (long)2 * B * C + (long)b * C;
    float* o_da1 = scratch + (long)4 * B * C;
    float* o_h = o_da1 + (long)B * S;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dh = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane * 4; c < C; c += 256) {
        acc += *reinterpret_cast<const f32x4*>(w1 + (long)j * C + c) * *reinterpret_cast<const f32x4*>(mean + c);
        dh += *reinterpret_cast<const f32x4*>(w2t + (long)j * C + c) * *reinterpret_cast<const f32x4*>(du2 + c);
    }
    const float a = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3])) + b1[j];
    const float d = wave_sum((dh[0] + dh[1]) + (dh[2] + dh[3])) * dact(a, FTC_ACT_SILU);
    if (lane == 0) {
        o_da1[(long)b * S + j] = d;
        o_h[(long)b * S + j] = a / (1.0f + expf(-a));
    }
}

// (3) grid (ceil(C / 256), B): d mean / HW = fc1^T . da1 / HW
__global__ __launch_bounds__(256) void sebwd_dmean_kernel(const float* __restrict__ w1, float* __restrict__ scratch, int B, int C, int S, float inv_hw) {
    extern __shared__ float da1[];
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const float* o_da1 = scratch + (long)4 * B * C + (long)b * S;
    for (int j = threadIdx.x; j < S; j += 256) da1[j] = o_da1[j];
    __syncthreads();
    if (c >= C) return;
    float acc = 0.f;
#pragma unroll 8
    for (int j = 0; j < S; ++j) acc += da1[j] * w1[(long)j * C + c];
    scratch[(long)3 * B * C + (long)b * C + c] = acc * inv_hw;
}

// grads: fc1.weight [S][C] | fc1.bias [S] | fc2.weight [C][S] | fc2.bias [C].  grid (ceil(C / 256), S): one (j, c) pair per thread
__global__ __launch_bounds__(256) void sebwd_w_kernel(const float* __restrict__ scratch, float* __restrict__ grads, int B, int C, int S) {
    const float* du2 = scratch + (long)B * C;
    const float* mean = scratch + (long)2 * B * C;
    const float* da1 = scratch + (long)4 * B * C;
    const float* h = da1 + (long)B * S;
    float* gw1 = grads;
    float* gb1 = gw1 + (long)S * C;
    float* gw2 = gb1 + S;
    float* gb2 = gw2 + (long)C * S;
    const int c = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += da1[(long)b * S + j];
        gb1[j] += a;
    }
    if (c >= C) return;
    float a1 = 0.f, a2 = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = du2[(long)b * C + c];
        a1 += da1[(long)b * S + j] * mean[(long)b * C + c];
        a2 += d * h[(long)b * S + j];
        sb += d;
    }
    gw1[(long)j * C + c] += a1;
    gw2[(long)c * S + j] += a2;
    if (j == 0) gb2[c] += sb;
}

// ------------------------------------------------------------------------------------------------------------------------
// FTC_OP_UPCATBWD: transpose of the x2 bilinear upsample (align_corners=True) in gather form, with the forward's own index arithmetic
// (upcat_kernel, fpn_ops.hip): low-resolution pixel (yi, xi) collects w_y(Y, yi) * w_x(X, xi) * g[Y, X].
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float up_weight(int Y, int yi, int Hi, float ry) {
    const float sy = ry * (float)Y;
    const int y0 = (int)sy;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
    const float l1 = sy - (float)y0, l0 = 1.0f - l1;
    return (y0 == yi ? l0 : 0.f) + (y1 == yi ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void upcatbwd_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int Hi, int Wi, int Ho, int Wo, int Cy,
                                                       int Ctot, float ry, float rx) {
    const int Q = Cy >> 2;
    const long total = (long)B * Hi * Wi * Q;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pix = idx / Q;
        const int c = (int)(idx - pix * Q) * 4;
        const long row = pix / Wi;
        const int xi = (int)(pix - row * Wi);
        const int b = (int)(row / Hi);
        const int yi = (int)(row - (long)b * Hi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int Ylo = max(0, 2 * yi - 3), Yhi = min(Ho - 1, 2 * yi + 3);
        const int Xlo = max(0, 2 * xi - 3), Xhi = min(Wo - 1, 2 * xi + 3);
        for (int Y = Ylo; Y <= Yhi; ++Y) {
            const float wy = up_weight(Y, yi, Hi, ry);
            if (wy == 0.f) continue;
            for (int X = Xlo; X <= Xhi; ++X) {
                const float wx = up_weight(X, xi, Wi, rx);
                if (wx == 0.f) continue;
                acc += (wy * wx) * *reinterpret_cast<const f32x4*>(g + (((long)b * Ho + Y) * Wo + X) * Ctot + c);
            }
        }
        *reinterpret_cast<f32x4*>(out + pix * Cy + c) = acc;
    }
}

// FTC_OP_DILATE
__global__ __launch_bounds__(256) void dilate_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int Ho, int Wo, int C) {
    const int Q = C >> 2;
    const long total = (long)B * Ho * Wo * Q;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pix = idx / Q;
        const int c = (int)(idx - pix * Q) * 4;
        const long row = pix / Wo;
        const int x = (int)(pix - row * Wo);
        const int b = (int)(row / Ho);
        const int y = (int)(row - (long)b * Ho);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(x & 1) && !(y & 1) && (y >> 1) < H && (x >> 1) < W) v = *reinterpret_cast<const f32x4*>(in + (((long)b * H + (y >> 1)) * W + (x >> 1)) * C + c);
        *reinterpret_cast<f32x4*>(out + pix * C + c) = v;
    }
}

// FTC_OP_TOPDGRAD: dY[p][ci] = sum_{co, r, s} g[p - (r-1, s-1)][off + co] * W[co][r*3+s][ci]
template <typename WT>
__global__ __launch_bounds__(256) void topdgrad_kernel(const float* __restrict__ g, const WT* __restrict__ w, float* __restrict__ out, int B, int H, int W,
                                                       int Co, int CoT, int off, int Ci) {
    const int Q = Ci >> 2;
    const long total = (long)B * H * W * Q;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pix = idx / Q;
        const int c = (int)(idx - pix * Q) * 4;
        const long row = pix / W;
        const int x = (int)(pix - row * W);
        const int b = (int)(row / H);
        const int y = (int)(row - (long)b * H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int qy = y - (r - 1);
            if ((unsigned)qy >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int qx = x - (s - 1);
                if ((unsigned)qx >= (unsigned)W) continue;
                const float* gp = g + (((long)b * H + qy) * W + qx) * CoT + off;
                for (int co = 0; co < Co; ++co) acc += gp[co] * load4<WT>(w + ((long)co * 9 + r * 3 + s) * Ci + c);
            }
        }
        *reinterpret_cast<f32x4*>(out + pix * Ci + c) = acc;
    }
}

// FTC_OP_COLSUM
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, double* __restrict__ part, long M, int C, int CT, int off, int nchunk) {
    __shared__ double red[4][64];
    const int t = threadIdx.x, cl = t & 63, rl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int chunk = blockIdx.y;
    const long rows = (M + nchunk - 1) / nchunk;
    const long r0 = (long)chunk * rows, r1 = r0 + rows < M ? r0 + rows : M;
    double s = 0.0;
    if (c < C)
        for (long r = r0 + rl; r < r1; r += 4) s += (double)x[r * CT + off + c];
    red[rl][cl] = s;
    __syncthreads();
    if (t < 64 && c < C) part[(long)chunk * C + c] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, float* __restrict__ out, int C, int nchunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += part[(long)k * C + c];
    out[c] += (float)s;
}

// FTC_OP_STEMWGRAD: gW[co][ci][r][s] += sum_p dz[p][co] * (2 img - 1)[p*2 + (r-1, s-1)][ci]; thread = (co, tap group), 4 taps each.
__global__ __launch_bounds__(256) void stemwgrad_partial_kernel(const float* __restrict__ img, const float* __restrict__ dz, double* __restrict__ part, int B,
                                                                int H, int W, int Ho, int Wo, int Co, int nchunk) {
    const int t = threadIdx.x;
    const int co = t % Co, kg = t / Co;                    // kg in 0..7 (Co <= 32)
    const int chunk = blockIdx.x;
    const long M = (long)B * Ho * Wo;
    const long rows = (M + nchunk - 1) / nchunk;
    const long r0 = (long)chunk * rows, r1 = r0 + rows < M ? r0 + rows : M;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (kg < 8)
        for (long p = r0; p < r1; ++p) {
            const long row = p / Wo;
            const int ox = (int)(p - row * Wo);
            const int b = (int)(row / Ho);
            const int oy = (int)(row - (long)b * Ho);
            const float g = dz[p * Co + co];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kg + 8 * i;                   // k = (r*3+s)*3 + ci
                if (k >= 27) break;
                const int ci = k % 3, rs = k / 3, r = rs / 3, s = rs - r * 3;
                const int iy = oy * 2 - 1 + r, ix = ox * 2 - 1 + s;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    acc[i] += (double)(g * (img[(((long)b * H + iy) * W + ix) * 3 + ci] * 2.0f - 1.0f));
            }
        }
    if (kg < 8)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kg + 8 * i;
            if (k < 27) part[((long)chunk * 27 + k) * Co + co] = acc[i];
        }
}
__global__ __launch_bounds__(256) void stemwgrad_final_kernel(const double* __restrict__ part, float* __restrict__ gw, int Co, int nchunk) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // i = k * Co + co
    if (i >= 27 * Co) return;
    const int k = i / Co, co = i - k * Co;
    double s = 0.0;
    for (int j = 0; j < nchunk; ++j) s += part[((long)j * 27 + k) * Co + co];
    const int ci = k % 3, rs = k / 3;
    gw[((long)co * 3 + ci) * 9 + rs] += (float)s;          // [Co][3][3][3]
}

// FTC_OP_SCATTER_ROWS (after a memset of `out`)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ rows, const int32_t* __restrict__ sel, float* __restrict__ out, long n, int C) {
    const int Q = C >> 2;
    const long total = n * Q;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / Q;
        const int c = (int)(idx - r * Q) * 4;
        *reinterpret_cast<f32x4*>(out + (long)sel[r] * C + c) = *reinterpret_cast<const f32x4*>(rows + r * C + c);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// FTC_OP_LOSS_BWD: d (sum_i alpha_i loss_i) * loss_scale (loss_func.py:94-177 differentiated by hand; CoVWeightingLoss multiplies the
// raw losses by DETACHED alphas, loss_func.py:69-71).
// ------------------------------------------------------------------------------------------------------------------------
struct LossBwdP {
    const float* heat; const float* label; const int32_t* idmap; const float* alphas; const float* lossvec;
    float* gheat; int B, h, w; float lscale;
};

__global__ __launch_bounds__(256) void maploss_bwd_kernel(LossBwdP p) {
    const long hw = (long)p.h * p.w, n = (long)p.B * hw;
    const float inv_n = 1.0f / (float)n;
    const float w1c = p.lossvec[12];
    const float a_key = p.alphas[0] * p.lscale, a_size = p.alphas[1] * p.lscale, a_line = p.alphas[2] * p.lscale, a_sep = p.alphas[3] * p.lscale;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / hw, r = i - b * hw;
        const float* hp = p.heat + i * 9;
        const float* lp = p.label + b * 5 * hw + r;
        float* gp = p.gheat + i * 9;
        const float key = lp[0];
        const float x = hp[0];
        const float pr = 1.0f / (1.0f + expf(-x));
        float d;
        if (key >= 1.0f) {
            const float logsig = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
            const float om = 1.0f - pr;
            d = -om * om * om + logsig * 2.0f * pr * om * om;
        } else {
            const float om = 1.0f - key;
            const float sp = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));          // softplus(x) = x + softplus(-x)
            d = (om * om) * (om * om) * pr * pr * (pr + 2.0f * sp * (1.0f - pr));
        }
        gp[0] = a_key * 10.0f * inv_n * d;
        const float w2 = fmaxf(key - 0.85f, 0.f) / (1.f - 0.85f);
        if (key > 0.85f) {
            const float d1 = hp[1] - lp[1 * hw], d2 = hp[2] - lp[2 * hw];
            gp[1] = a_size * w2 / w1c * fminf(fmaxf(d1, -1.0f), 1.0f);
            gp[2] = a_size * w2 / w1c * fminf(fmaxf(d2, -1.0f), 1.0f);
        } else {
            gp[1] = 0.f;
            gp[2] = 0.f;
        }
        gp[3] = a_line * inv_n * (1.0f / (1.0f + expf(-hp[3])) - lp[3 * hw]);
        gp[4] = a_sep * inv_n * (1.0f / (1.0f + expf(-hp[4])) - lp[4 * hw]);
        const int code = p.idmap[(b * 2 + 1) * hw + r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float bit = (code & (1 << k)) ? 1.f : 0.f;
            gp[5 + k] = p.alphas[5 + k] * p.lscale * inv_n * (1.f + bit * w2 + w2) * (1.0f / (1.0f + expf(-hp[5 + k])) - bit);
        }
    }
}

struct IdBwdP {
    const float* dec[3]; int mod[3];
    const int32_t* sel_index; long n; const float* label; const int32_t* idmap; long hw;
    const float* alphas; const float* lossvec; float* gdec; int pad; float lscale;
};

// one wave per selected row: d id_loss / d logits = w3 / w3c * (softmax - onehot) on the rows of mask3, 0 elsewhere
__global__ __launch_bounds__(256) void idloss_bwd_kernel(IdBwdP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float a_id = p.alphas[4] * p.lscale;
    const float w3c = p.lossvec[13];
    for (long r = (long)blockIdx.x * 4 + wave; r < p.n; r += (long)gridDim.x * 4) {
        const long px = p.sel_index[r];
        const long b = px / p.hw, q = px - b * p.hw;
        const float key = p.label[b * 5 * p.hw + q];
        const int id = p.idmap[b * 2 * p.hw + q];
        const bool m3 = key > 0.99f && id > 0;
        const float coef = m3 ? a_id * (fmaxf(key - 0.99f, 0.f) / (1.f - 0.99f)) / w3c : 0.f;
        for (int hd = 0; hd < 3; ++hd) {
            const int m = p.mod[hd];
            float* go = p.gdec + ((long)hd * p.n + r) * p.pad;
            if (!m3) {
                for (int c = lane; c < p.pad; c += 64) go[c] = 0.f;
                continue;
            }
            const float* row = p.dec[hd] + r * m;
            float mx = -INFINITY;
            for (int c = lane; c < m; c += 64) mx = fmaxf(mx, row[c]);
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float se = 0.f;
            for (int c = lane; c < m; c += 64) se += expf(row[c] - mx);
            se = wave_sum(se);
            const int tgt = id % m;
            const float inv = 1.0f / se;
            for (int c = lane; c < p.pad; c += 64) go[c] = c < m ? coef * (expf(row[c] - mx) * inv - (c == tgt ? 1.f : 0.f)) : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// ftc_pack_train_weights: fp32 OIHW -> K-major forward weights and flipped / transposed data-gradient weights, one launch for
// every convolution and Linear layer.  grid (ceil(max_elems / 256), n_entries).
// ------------------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void put(void* base, long i, float v) { reinterpret_cast<T*>(base)[i] = from_f32<T>(v); }
__device__ __forceinline__ void put_dt(void* base, long i, float v, int dt) {
    if (dt == FTC_F32) put<float>(base, i, v);
    else if (dt == FTC_F16) reinterpret_cast<_Float16*>(base)[i] = (_Float16)v;      // weights: plain conversion (no saturation needed)
    else put<__bf16>(base, i, v);
}
__global__ __launch_bounds__(256) void pack_train_kernel(const ftc_pack_entry* __restrict__ ent) {
    const ftc_pack_entry e = ent[blockIdx.y];
    const int kk = e.kk;
    const long nf = (long)e.Cout * kk * e.cin_pad, nd = e.dgrad ? (long)e.Cin * kk * e.cout_pad : 0;
    const float* src = static_cast<const float*>(e.src);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += (long)gridDim.x * 256) {
        if (i < nf) {
            if (!e.fwd) continue;
            const int ci = (int)(i % e.cin_pad);
            const long t = i / e.cin_pad;
            const int tap = (int)(t % kk), co = (int)(t / kk);
            put_dt(e.fwd, i, ci < e.Cin ? src[((long)co * e.Cin + ci) * kk + tap] : 0.f, e.dtype);
        } else {
            const long j = i - nf;
            const int co = (int)(j % e.cout_pad);
            const long t = j / e.cout_pad;
            const int tap = (int)(t % kk), ci = (int)(t / kk);
            put_dt(e.dgrad, j, co < e.Cout ? src[((long)co * e.Cin + ci) * kk + (kk - 1 - tap)] : 0.f, e.dtype);
        }
    }
}

inline int nblocks(long total, int cap = 16384) { const long nb = (total + 255) / 256; return (int)(nb < 1 ? 1 : nb > cap ? cap : nb); }

}  // namespace

hipError_t launch_pack_train(const ftc_pack_entry* entries, int n, long max_elems, hipStream_t s) {
    // (round 4: up to 512 workgroups per entry -- with 64 the 20 M-element entries of the merged FPN level ran on a quarter of the GPU and the
    //  launch took 3.3 ms for 2 GB of traffic; workgroups beyond a small entry's size fall through their loop)
    hipLaunchKernelGGL(pack_train_kernel, dim3(nblocks(max_elems, 512), n), dim3(256), 0, s, entries);
    return hipGetLastError();
}

namespace {
template <int Q, bool FAST, typename TC, typename ZT>
hipError_t bnbwd_run(const BnBwdP& p, double* part, float* coef, float* ggamma, float* gbeta, float* out, TC* out2, int accum, int nchunk, hipStream_t s) {
    const dim3 grid(p.C / (4 * Q), nchunk);
    hipLaunchKernelGGL((bnbwd_partial_kernel<Q, FAST, ZT>), grid, dim3(256), 0, s, p, part, nchunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bnbwd_final_kernel, dim3((p.C + 15) / 16), dim3(256), 0, s, part, ggamma, gbeta, coef, (long)p.M, p.C, nchunk);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the apply pass streams: enough row chunks to fill the GPU (the partial pass is bound to the scratch layout's chunk count)
    int ach = (int)((4096L * 4 * Q) / p.C);
    ach = ach < 1 ? 1 : ach > p.M / 8 + 1 ? p.M / 8 + 1 : ach;
    hipLaunchKernelGGL((bnbwd_apply_kernel<Q, FAST, TC, ZT>), dim3(p.C / (4 * Q), ach), dim3(256), 0, s, p, coef, out, out2, accum, ach);
    return hipGetLastError();
}
template <bool FAST, typename TC, typename ZT>
hipError_t bnbwd_q(int q, const BnBwdP& p, double* part, float* coef, float* gg, float* gb, float* out, TC* out2, int accum, int nchunk, hipStream_t s) {
    switch (q) {
    case 64: return bnbwd_run<64, FAST, TC, ZT>(p, part, coef, gg, gb, out, out2, accum, nchunk, s);
    case 32: return bnbwd_run<32, FAST, TC, ZT>(p, part, coef, gg, gb, out, out2, accum, nchunk, s);
    case 16: return bnbwd_run<16, FAST, TC, ZT>(p, part, coef, gg, gb, out, out2, accum, nchunk, s);
    case 8: return bnbwd_run<8, FAST, TC, ZT>(p, part, coef, gg, gb, out, out2, accum, nchunk, s);
    case 4: return bnbwd_run<4, FAST, TC, ZT>(p, part, coef, gg, gb, out, out2, accum, nchunk, s);
    case 2: return bnbwd_run<2, FAST, TC, ZT>(p, part, coef, gg, gb, out, out2, accum, nchunk, s);
    default: return bnbwd_run<1, FAST, TC, ZT>(p, part, coef, gg, gb, out, out2, accum, nchunk, s);
    }
}
}  // namespace

hipError_t launch_bnbwd(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    BnBwdP p;
    p.gy = (const float*)a.in; p.gs = o.Cin_total > 0 ? o.Cin_total : o.Cin; p.goff = o.cin_off;
    p.z = a.in2; p.ss = a.scale; p.keep = (const float*)a.w2; p.ga = a.bias; p.gb = a.bias2;
    p.HW = o.H * o.W; p.M = o.B * o.H * o.W; p.C = o.Cin; p.act = o.act;
    const int nchunk = ftc_bnstat_chunks(p.M);
    double* part = reinterpret_cast<double*>(a.aux);
    float* coef = reinterpret_cast<float*>(part + (long)nchunk * 2 * p.C);
    const int q = pick_quads(p.C);
    float* gg = (float*)const_cast<void*>(a.w);
    float* gb = const_cast<float*>(a.shift);
    const int accum = (o.flags & FTC_FLAG_ACCUM) ? 1 : 0;
    // in_dtype = the type z (in2) is STORED in: fp32, or the plan's 16-bit compute type (what the reference's autocast stores: the
    // convolution wrote it in 16 bits, and the three BatchNorm passes -- statistics, normalise, backward -- stream half the bytes)
    const bool z16 = ftc_is16(o.in_dtype);
    if (o.w_dtype == FTC_F32) return bnbwd_q<false, float, float>(q, p, part, coef, gg, gb, (float*)a.out, (float*)nullptr, accum, nchunk, s);
    if (o.w_dtype == FTC_F16)
        return z16 ? bnbwd_q<true, _Float16, _Float16>(q, p, part, coef, gg, gb, (float*)a.out, (_Float16*)a.out2, accum, nchunk, s)
                   : bnbwd_q<true, _Float16, float>(q, p, part, coef, gg, gb, (float*)a.out, (_Float16*)a.out2, accum, nchunk, s);
    return z16 ? bnbwd_q<true, __bf16, __bf16>(q, p, part, coef, gg, gb, (float*)a.out, (__bf16*)a.out2, accum, nchunk, s)
               : bnbwd_q<true, __bf16, float>(q, p, part, coef, gg, gb, (float*)a.out, (__bf16*)a.out2, accum, nchunk, s);
}

hipError_t launch_dwbwd(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const long M = (long)o.B * o.Ho * o.Wo;
    const int nchunk = ftc_chunks256(M), C = o.Cin;
    // `out` (data gradient) and `out2` + `aux` (weight gradient) are each optional (round 4): the train plan emits the weight half as its own op on
    // the side stream -- nothing on the backward chain reads a depthwise weight gradient (7.4 ms of the step's main stream)
    hipError_t e = hipSuccess;
    if (a.out) {
        static const bool old_form = [] { const char* e2 = std::getenv("FTC_DWBWD_DATA_OLD"); return e2 && *e2 && *e2 != '0'; }();
        const long Mi = (long)o.B * o.H * o.W;
        if (!old_form && Mi < 0x7fffffffL) {
            const int Qd = pick_quads(C);
            const int RLd = 256 / Qd;
            // row chunks: ~8 workgroups per CU over all channel groups, at least 2 trips of 2 pixels per lane
            long want = (2048L * 4 * Qd) / C;
            const long maxc = Mi / (4L * RLd) + 1;
            want = want < 1 ? 1 : want > maxc ? maxc : want;
            const dim3 gd(C / (4 * Qd), (unsigned)want);
#define DWD(QQ) hipLaunchKernelGGL(dwbwd_data_q_kernel<QQ>, gd, dim3(256), 0, s, (const float*)a.in2, (const float*)a.w, (float*)a.out, o.B, o.H, o.W, o.Ho, o.Wo, C, \
                                   o.stride, (int)want)
            switch (Qd) { case 64: DWD(64); break; case 32: DWD(32); break; case 16: DWD(16); break; case 8: DWD(8); break; case 4: DWD(4); break; case 2: DWD(2); break;
                          default: DWD(1); }
#undef DWD
        } else
        hipLaunchKernelGGL(dwbwd_data_kernel, dim3(nblocks((long)o.B * o.H * o.W * (C / 4))), dim3(256), 0, s, (const float*)a.in2, (const float*)a.w, (float*)a.out,
                           o.B, o.H, o.W, o.Ho, o.Wo, C, o.stride);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (!a.out2) return hipSuccess;
    double* part = reinterpret_cast<double*>(a.aux);
    const int Q = pick_quads(C) > 16 ? 16 : pick_quads(C);        // 9 x 4 accumulators per lane: keep the LDS image at 36 KB
#define DWW(QQ) hipLaunchKernelGGL(dwbwd_weight_partial_kernel<QQ>, dim3(C / (4 * QQ), nchunk), dim3(256), 0, s, (const float*)a.in, (const float*)a.in2, part, \
                                   o.B, o.H, o.W, o.Ho, o.Wo, C, o.stride, nchunk)
    switch (Q) { case 16: DWW(16); break; case 8: DWW(8); break; case 4: DWW(4); break; case 2: DWW(2); break; default: DWW(1); }
#undef DWW
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dwbwd_weight_final_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, s, part, (float*)a.out2, C, nchunk);
    return hipGetLastError();
}

hipError_t launch_sebwd(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int C = o.Cin, S = o.aux0, P = o.aux1, HW = o.H * o.W;
    float* scratch = (float*)a.out;
    float* dsp = scratch + (long)4 * o.B * C + (long)2 * o.B * S;
    const int Q = pick_quads(C);
    int pch = 1024 / ((C / (4 * Q)) * o.B);
    pch = pch < 1 ? 1 : pch > SE_PCH ? SE_PCH : pch;
    if (pch > HW / 16) pch = HW / 16 > 0 ? HW / 16 : 1;
#define SEDS(QQ) hipLaunchKernelGGL(sebwd_ds_kernel<QQ>, dim3(C / (4 * QQ), pch, o.B), dim3(256), 0, s, (const float*)a.in, (const float*)a.in2, dsp, HW, C, pch)
    switch (Q) { case 64: SEDS(64); break; case 32: SEDS(32); break; case 16: SEDS(16); break; case 8: SEDS(8); break; case 4: SEDS(4); break;
                 case 2: SEDS(2); break; default: SEDS(1); }
#undef SEDS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const float inv_hw = 1.0f / (float)HW;
    hipLaunchKernelGGL(sebwd_prep_kernel, dim3((C + 255) / 256, o.B), dim3(256), 0, s, (const float*)a.aux, a.scale, scratch, o.B, C, S, P, inv_hw, pch);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sebwd_hidden_kernel, dim3((S + 3) / 4, o.B), dim3(256), 0, s, (const float*)a.w, a.bias, (const float*)a.w2, scratch, o.B, C, S);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sebwd_dmean_kernel, dim3((C + 255) / 256, o.B), dim3(256), (size_t)S * sizeof(float), s, (const float*)a.w, scratch, o.B, C, S, inv_hw);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sebwd_w_kernel, dim3((C + 255) / 256, S), dim3(256), 0, s, scratch, (float*)a.out2, o.B, C, S);
    return hipGetLastError();
}

hipError_t launch_upcatbwd(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const float ry = o.Ho > 1 ? (float)(o.H - 1) / (float)(o.Ho - 1) : 0.f;
    const float rx = o.Wo > 1 ? (float)(o.W - 1) / (float)(o.Wo - 1) : 0.f;
    hipLaunchKernelGGL(upcatbwd_kernel, dim3(nblocks((long)o.B * o.H * o.W * (o.aux0 / 4))), dim3(256), 0, s, (const float*)a.in, (float*)a.out, o.B, o.H, o.W, o.Ho,
                       o.Wo, o.aux0, o.Cin_total, ry, rx);
    return hipGetLastError();
}

hipError_t launch_dilate(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    hipLaunchKernelGGL(dilate_kernel, dim3(nblocks((long)o.B * o.Ho * o.Wo * (o.Cin / 4))), dim3(256), 0, s, (const float*)a.in, (float*)a.out, o.B, o.H, o.W, o.Ho,
                       o.Wo, o.Cin);
    return hipGetLastError();
}

hipError_t launch_topdgrad(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const dim3 grid(nblocks((long)o.B * o.H * o.W * (o.Cout / 4)));
    if (o.w_dtype == FTC_F32)
        hipLaunchKernelGGL(topdgrad_kernel<float>, grid, dim3(256), 0, s, (const float*)a.in, (const float*)a.w, (float*)a.out, o.B, o.H, o.W, o.Cin, o.Cin_total,
                           o.cin_off, o.Cout);
    else if (o.w_dtype == FTC_F16)
        hipLaunchKernelGGL(topdgrad_kernel<_Float16>, grid, dim3(256), 0, s, (const float*)a.in, (const _Float16*)a.w, (float*)a.out, o.B, o.H, o.W, o.Cin,
                           o.Cin_total, o.cin_off, o.Cout);
    else
        hipLaunchKernelGGL(topdgrad_kernel<__bf16>, grid, dim3(256), 0, s, (const float*)a.in, (const __bf16*)a.w, (float*)a.out, o.B, o.H, o.W, o.Cin, o.Cin_total,
                           o.cin_off, o.Cout);
    return hipGetLastError();
}

hipError_t launch_colsum(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const long M = (long)o.B * o.H * o.W;
    const int nchunk = ftc_chunks256(M), C = o.Cin;
    double* part = reinterpret_cast<double*>(a.aux);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, nchunk), dim3(256), 0, s, (const float*)a.in, part, M, C, o.Cin_total > 0 ? o.Cin_total : C, o.cin_off,
                       nchunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, (float*)a.out, C, nchunk);
    return hipGetLastError();
}

hipError_t launch_stemwgrad(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const long M = (long)o.B * o.Ho * o.Wo;
    const int nchunk = ftc_stemwgrad_chunks(M);
    double* part = reinterpret_cast<double*>(a.aux);
    hipLaunchKernelGGL(stemwgrad_partial_kernel, dim3(nchunk), dim3(256), 0, s, (const float*)a.in, (const float*)a.in2, part, o.B, o.H, o.W, o.Ho, o.Wo, o.Cout,
                       nchunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stemwgrad_final_kernel, dim3((27 * o.Cout + 255) / 256), dim3(256), 0, s, part, (float*)a.out, o.Cout, nchunk);
    return hipGetLastError();
}

hipError_t launch_fill(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    return hipMemsetAsync(a.out, o.aux0 & 255, (size_t)o.B * o.H * o.W * o.Cin * 4, s);      // aux0 = the byte (0 unless the plan says otherwise)
}

hipError_t launch_gather_rows_op(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    return launch_gather_rows((const float*)a.in, (const int32_t*)a.in2, nullptr, o.aux0, o.Cin, o.Cout_total, a.out, o.out_dtype, s);
}

hipError_t launch_scatter_rows(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    hipError_t e = hipMemsetAsync(a.out, 0, (size_t)o.B * o.H * o.W * o.Cout_total * 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(nblocks((long)o.aux0 * (o.Cout_total / 4))), dim3(256), 0, s, (const float*)a.in, (const int32_t*)a.in2,
                       (float*)a.out, (long)o.aux0, o.Cout_total);
    return hipGetLastError();
}

hipError_t launch_loss_bwd(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    float lscale;
    memcpy(&lscale, &o.Cout, 4);
    LossBwdP p{(const float*)a.in, (const float*)a.in2, (const int32_t*)a.w, a.shift, a.aux, (float*)a.out, o.B, o.H, o.W, lscale};
    hipLaunchKernelGGL(maploss_bwd_kernel, dim3(nblocks((long)o.B * o.H * o.W, 2048)), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !a.out2 || o.aux0 <= 0) return e;
    IdBwdP q{{(const float*)a.w2, a.bias, a.bias2}, {1091, 1093, 1097}, (const int32_t*)a.scale, (long)o.aux0, (const float*)a.in2, (const int32_t*)a.w,
             (long)o.H * o.W, a.shift, a.aux, (float*)a.out2, o.aux1, lscale};
    hipLaunchKernelGGL(idloss_bwd_kernel, dim3((o.aux0 + 3) / 4 < 2048 ? (o.aux0 + 3) / 4 : 2048), dim3(256), 0, s, q);
    return hipGetLastError();
}
