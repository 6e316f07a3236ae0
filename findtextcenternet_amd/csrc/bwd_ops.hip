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
// ------------------------------------------------------------------------------------------------------------------------
struct BnBwdP {
    const float* gy; int gs, goff;      // incoming gradient, row stride / channel offset
    const float* z;                     // BN input [M][C]
    const float* ss;                    // [4][C] scale, shift, mean, invstd
    const float* keep;                  // [B] or null
    const float* ga;                    // [B][C] or null
    const float* gb;                    // [B][C] or null
    int HW; long M; int C; int act;
};

__device__ __forceinline__ float bn_dt(const BnBwdP& p, long r, int c, int b, float zv, float sc, float sh) {
    float g = p.gy[r * p.gs + p.goff + c];
    if (p.ga) g *= p.ga[(long)b * p.C + c];
    if (p.gb) g += p.gb[(long)b * p.C + c];
    if (p.keep) g *= p.keep[b];
    return g * dact(zv * sc + sh, p.act);
}

__global__ __launch_bounds__(256) void bnbwd_partial_kernel(BnBwdP p, double* __restrict__ part, int nchunk) {
    __shared__ double red[2][4][64];
    const int t = threadIdx.x, cl = t & 63, rl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int chunk = blockIdx.y;
    const long rows = (p.M + nchunk - 1) / nchunk;
    const long r0 = (long)chunk * rows, r1 = r0 + rows < p.M ? r0 + rows : p.M;
    double s1 = 0.0, s2 = 0.0;
    if (c < p.C) {
        const float sc = p.ss[c], sh = p.ss[p.C + c], mean = p.ss[2 * p.C + c], istd = p.ss[3 * p.C + c];
        for (long r = r0 + rl; r < r1; r += 4) {
            const int b = (int)(r / p.HW);
            const float zv = p.z[r * p.C + c];
            const float dt = bn_dt(p, r, c, b, zv, sc, sh);
            s1 += (double)dt;
            s2 += (double)dt * (double)((zv - mean) * istd);
        }
    }
    red[0][rl][cl] = s1;
    red[1][rl][cl] = s2;
    __syncthreads();
    if (t < 64 && c < p.C) {
        part[((long)chunk * 2 + 0) * p.C + c] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
        part[((long)chunk * 2 + 1) * p.C + c] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
    }
}

__global__ __launch_bounds__(256) void bnbwd_final_kernel(const double* __restrict__ part, float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                          float* __restrict__ coef, long M, int C, int nchunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        s1 += part[((long)k * 2 + 0) * C + c];
        s2 += part[((long)k * 2 + 1) * C + c];
    }
    if (gbeta) gbeta[c] += (float)s1;
    if (ggamma) ggamma[c] += (float)s2;
    coef[c] = (float)(s1 / (double)M);
    coef[C + c] = (float)(s2 / (double)M);
}

__global__ __launch_bounds__(256) void bnbwd_apply_kernel(BnBwdP p, const float* __restrict__ coef, float* __restrict__ out, int accum) {
    const int Q = p.C >> 2;
    const long total = p.M * Q;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / Q;
        const int c = (int)(idx - r * Q) * 4;
        const int b = (int)(r / p.HW);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(p.z + r * p.C + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = p.ss[c + e], sh = p.ss[p.C + c + e], mean = p.ss[2 * p.C + c + e], istd = p.ss[3 * p.C + c + e];
            const float dt = bn_dt(p, r, c + e, b, zv[e], sc, sh);
            o[e] = sc * (dt - coef[c + e] - (zv[e] - mean) * istd * coef[p.C + c + e]);
        }
        float* op = out + r * p.C + c;
        if (accum) o += *reinterpret_cast<const f32x4*>(op);
        *reinterpret_cast<f32x4*>(op) = o;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// FTC_OP_DWBWD: depthwise 3x3 (pad 1, stride 1|2) backward.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwbwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ out, int B, int H,
                                                         int W, int Ho, int Wo, int C, int stride) {
    const int Q = C >> 2;
    const long total = (long)B * H * W * Q;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pix = idx / Q;
        const int c = (int)(idx - pix * Q) * 4;
        const long row = pix / W;
        const int ix = (int)(pix - row * W);
        const int b = (int)(row / H);
        const int iy = (int)(row - (long)b * H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ty = iy + 1 - r;
            if (ty < 0 || (ty % stride) != 0) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int tx = ix + 1 - s;
                if (tx < 0 || (tx % stride) != 0) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const f32x4 g = *reinterpret_cast<const f32x4*>(dz + (((long)b * Ho + oy) * Wo + ox) * C + c);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (r * 3 + s) * C + c);
                acc += g * wv;
            }
        }
        *reinterpret_cast<f32x4*>(out + pix * C + c) = acc;
    }
}

__global__ __launch_bounds__(256) void dwbwd_weight_partial_kernel(const float* __restrict__ x, const float* __restrict__ dz, double* __restrict__ part, int B,
                                                                   int H, int W, int Ho, int Wo, int C, int stride, int nchunk) {
    __shared__ float red[4][9][64];
    const int t = threadIdx.x, cl = t & 63, rl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int chunk = blockIdx.y;
    const long M = (long)B * Ho * Wo;
    const long rows = (M + nchunk - 1) / nchunk;
    const long r0 = (long)chunk * rows, r1 = r0 + rows < M ? r0 + rows : M;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    if (c < C)
        for (long p = r0 + rl; p < r1; p += 4) {
            const long row = p / Wo;
            const int ox = (int)(p - row * Wo);
            const int b = (int)(row / Ho);
            const int oy = (int)(row - (long)b * Ho);
            const float g = dz[p * C + c];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oy * stride + r - 1;
                if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int ix = ox * stride + s - 1;
                    if ((unsigned)ix >= (unsigned)W) continue;
                    acc[r * 3 + s] += g * x[(((long)b * H + iy) * W + ix) * C + c];
                }
            }
        }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[rl][k][cl] = acc[k];
    __syncthreads();
    for (int i = t; i < 9 * 64; i += 256) {
        const int k = i >> 6, l = i & 63;
        if (blockIdx.x * 64 + l < C)
            part[((long)chunk * 9 + k) * C + blockIdx.x * 64 + l] = ((double)red[0][k][l] + (double)red[1][k][l]) + ((double)red[2][k][l] + (double)red[3][k][l]);
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
// dmean/HW [B][C] | da1 [B][S] | h [B][S].
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sebwd_ds_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ ds, int HW, int C) {
    __shared__ double red[4][64];
    const int t = threadIdx.x, cl = t & 63, rl = t >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
    double acc = 0.0;
    if (c < C)
        for (int r = rl; r < HW; r += 4) {
            const long i = ((long)b * HW + r) * C + c;
            acc += (double)(g[i] * y[i]);
        }
    red[rl][cl] = acc;
    __syncthreads();
    if (t < 64 && c < C) ds[(long)b * C + c] = (float)((red[0][t] + red[1][t]) + (red[2][t] + red[3][t]));
}

__global__ __launch_bounds__(512) void sebwd_mlp_kernel(const float* __restrict__ sums, const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2t, const float* __restrict__ scale, float* __restrict__ scratch,
                                                        int B, int C, int S, int P, float inv_hw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // mean [C] | du2 [C] | a1 [S] | da1 [S]
    float* mean = lds;
    float* du2 = lds + C;
    float* a1 = du2 + C;
    float* da1 = a1 + S;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float* o_ds = scratch;
    float* o_du2 = scratch + (long)B * C;
    float* o_mean = scratch + (long)2 * B * C;
    float* o_gb = scratch + (long)3 * B * C;
    float* o_da1 = scratch + (long)4 * B * C;
    float* o_h = o_da1 + (long)B * S;
    for (int c = t; c < C; c += 512) {
        float m = 0.f;
        for (int p = 0; p < P; ++p) m += sums[((long)b * P + p) * C + c];
        m *= inv_hw;
        mean[c] = m;
        o_mean[(long)b * C + c] = m;
        const float sv = scale[(long)b * C + c];
        const float d = o_ds[(long)b * C + c] * sv * (1.0f - sv);
        du2[c] = d;
        o_du2[(long)b * C + c] = d;
    }
    __syncthreads();
    for (int j = wave; j < S; j += 8) {
        float acc = 0.f, dh = 0.f;
        for (int c = lane; c < C; c += 64) {
            acc += w1[(long)j * C + c] * mean[c];
            dh += w2t[(long)j * C + c] * du2[c];
        }
        acc = wave_sum(acc);
        dh = wave_sum(dh);
        if (lane == 0) {
            const float a = acc + b1[j];
            a1[j] = a;
            const float d = dh * dact(a, FTC_ACT_SILU);
            da1[j] = d;
            o_da1[(long)b * S + j] = d;
            o_h[(long)b * S + j] = a / (1.0f + expf(-a));
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += 512) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc += da1[j] * w1[(long)j * C + c];
        o_gb[(long)b * C + c] = acc * inv_hw;
    }
}

// grads: fc1.weight [S][C] | fc1.bias [S] | fc2.weight [C][S] | fc2.bias [C]
__global__ __launch_bounds__(256) void sebwd_w_kernel(const float* __restrict__ scratch, float* __restrict__ grads, int B, int C, int S) {
    const float* du2 = scratch + (long)B * C;
    const float* mean = scratch + (long)2 * B * C;
    const float* da1 = scratch + (long)4 * B * C;
    const float* h = da1 + (long)B * S;
    float* gw1 = grads;
    float* gb1 = gw1 + (long)S * C;
    float* gw2 = gb1 + S;
    float* gb2 = gw2 + (long)C * S;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; j < S; j += 256) {
            float a = 0.f;
            for (int b = 0; b < B; ++b) a += da1[(long)b * S + j];
            gb1[j] += a;
        }
    if (c >= C) return;
    float sb = 0.f;
    for (int b = 0; b < B; ++b) sb += du2[(long)b * C + c];
    gb2[c] += sb;
    for (int j = 0; j < S; ++j) {
        float a1 = 0.f, a2 = 0.f;
        for (int b = 0; b < B; ++b) {
            a1 += da1[(long)b * S + j] * mean[(long)b * C + c];
            a2 += du2[(long)b * C + c] * h[(long)b * S + j];
        }
        gw1[(long)j * C + c] += a1;
        gw2[(long)c * S + j] += a2;
    }
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
    hipLaunchKernelGGL(pack_train_kernel, dim3(nblocks(max_elems, 64), n), dim3(256), 0, s, entries);
    return hipGetLastError();
}

hipError_t launch_bnbwd(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    BnBwdP p;
    p.gy = (const float*)a.in; p.gs = o.Cin_total > 0 ? o.Cin_total : o.Cin; p.goff = o.cin_off;
    p.z = (const float*)a.in2; p.ss = a.scale; p.keep = (const float*)a.w2; p.ga = a.bias; p.gb = a.bias2;
    p.HW = o.H * o.W; p.M = (long)o.B * o.H * o.W; p.C = o.Cin; p.act = o.act;
    const int nchunk = ftc_bnstat_chunks(p.M);
    double* part = reinterpret_cast<double*>(a.aux);
    float* coef = reinterpret_cast<float*>(part + (long)nchunk * 2 * p.C);
    hipLaunchKernelGGL(bnbwd_partial_kernel, dim3((p.C + 63) / 64, nchunk), dim3(256), 0, s, p, part, nchunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bnbwd_final_kernel, dim3((p.C + 255) / 256), dim3(256), 0, s, part, (float*)const_cast<void*>(a.w), const_cast<float*>(a.shift), coef,
                       p.M, p.C, nchunk);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bnbwd_apply_kernel, dim3(nblocks(p.M * (p.C / 4))), dim3(256), 0, s, p, coef, (float*)a.out, (o.flags & FTC_FLAG_ACCUM) ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_dwbwd(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const long M = (long)o.B * o.Ho * o.Wo;
    const int nchunk = ftc_bnstat_chunks(M), C = o.Cin;
    hipLaunchKernelGGL(dwbwd_data_kernel, dim3(nblocks((long)o.B * o.H * o.W * (C / 4))), dim3(256), 0, s, (const float*)a.in2, (const float*)a.w, (float*)a.out,
                       o.B, o.H, o.W, o.Ho, o.Wo, C, o.stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    double* part = reinterpret_cast<double*>(a.aux);
    hipLaunchKernelGGL(dwbwd_weight_partial_kernel, dim3((C + 63) / 64, nchunk), dim3(256), 0, s, (const float*)a.in, (const float*)a.in2, part, o.B, o.H, o.W,
                       o.Ho, o.Wo, C, o.stride, nchunk);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dwbwd_weight_final_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, s, part, (float*)a.out2, C, nchunk);
    return hipGetLastError();
}

hipError_t launch_sebwd(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int C = o.Cin, S = o.aux0, P = o.aux1, HW = o.H * o.W;
    float* scratch = (float*)a.out;
    hipLaunchKernelGGL(sebwd_ds_kernel, dim3((C + 63) / 64, o.B), dim3(256), 0, s, (const float*)a.in, (const float*)a.in2, scratch, HW, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sebwd_mlp_kernel, dim3(o.B), dim3(512), (size_t)(2 * C + 2 * S) * sizeof(float), s, (const float*)a.aux, (const float*)a.w, a.bias,
                       (const float*)a.w2, a.scale, scratch, o.B, C, S, P, 1.0f / (float)HW);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sebwd_w_kernel, dim3((C + 255) / 256), dim3(256), 0, s, scratch, (float*)a.out2, o.B, C, S);
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
    const int nchunk = ftc_bnstat_chunks(M), C = o.Cin;
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
    return hipMemsetAsync(a.out, 0, (size_t)o.B * o.H * o.W * o.Cin * 4, s);
}

hipError_t launch_gather_rows_op(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    return launch_gather_rows((const float*)a.in, (const int32_t*)a.in2, nullptr, o.aux0, o.Cin, o.Cout_total, a.out, FTC_F32, s);
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
