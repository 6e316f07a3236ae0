// Schedule-Free AdamW update as ONE multi-tensor kernel (BASELINE config 5).
//
// The reference's optimizer (/root/reference/models/adamw_schedulefree.py:157-184, the `foreach` branch) walks the
// parameters in ten torch._foreach_* passes -- mul, addcmul, div, sqrt, add, div, add, lerp, add, sub -- i.e. about
// 27 streams of 4 bytes per parameter through HBM for 262 M parameters.  The update is elementwise, so here every
// element is read once (y, grad, exp_avg_sq, z) and written once (y, exp_avg_sq, z [, the normalised grad]): 7-8 streams.
//
//   v  <- v*beta2 + ((1-beta2)*g)*g
//   gn <- g / (sqrt(v / bias_correction2) + eps)  [+ decay*y]
//   y  <- lerp(y, z, ckp1) + y_alpha*gn            y_alpha = lr*(beta1*(1-ckp1)-1)
//   z  <- z - lr*gn
//
// Operation order and the lerp formula are ATen's (weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)), contraction is off, so
// the result differs from the ten-pass form by rounding of the last bit at most.  The per-step scalars are computed on the
// host in float64 (findtextcenternet_amd/optim.py) and arrive as the fp32 values ATen would use.
#include "ftc_common.h"

#pragma clang fp contract(off)

namespace {

struct Scal { float beta2, omb2, bc2, eps, decay, ckp1, y_alpha, lr; int write_grad; };

__device__ __forceinline__ void upd(float& y, float& g, float& v, float& z, const Scal& c) {
    v = v * c.beta2;
    v = v + (c.omb2 * g) * g;
    const float denom = sqrtf(v / c.bc2) + c.eps;
    float gn = g / denom;
    if (c.decay != 0.0f) gn = gn + c.decay * y;
    const float d = z - y;
    const float yl = c.ckp1 < 0.5f ? y + c.ckp1 * d : z - d * (1.0f - c.ckp1);
    y = yl + c.y_alpha * gn;
    z = z - c.lr * gn;
    g = gn;
}

// one workgroup per chunk (<= 4096 elements of one parameter tensor, 16-byte aligned start)
__global__ __launch_bounds__(256) void adamw_sf_kernel(const ftc_mt_chunk* __restrict__ chunks, const Scal c) {
    const ftc_mt_chunk ch = chunks[blockIdx.x];
    float* y = static_cast<float*>(ch.y);
    float* g = static_cast<float*>(ch.g);
    float* v = static_cast<float*>(ch.v);
    float* z = static_cast<float*>(ch.z);
    const int n4 = ch.n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
        f32x4 Y = reinterpret_cast<f32x4*>(y)[i], G = reinterpret_cast<f32x4*>(g)[i], V = reinterpret_cast<f32x4*>(v)[i],
              Z = reinterpret_cast<f32x4*>(z)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = Y[e], b = G[e], d = V[e], f = Z[e];
            upd(a, b, d, f, c);
            Y[e] = a; G[e] = b; V[e] = d; Z[e] = f;
        }
        reinterpret_cast<f32x4*>(y)[i] = Y;
        reinterpret_cast<f32x4*>(v)[i] = V;
        reinterpret_cast<f32x4*>(z)[i] = Z;
        if (c.write_grad) reinterpret_cast<f32x4*>(g)[i] = G;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < ch.n; i += 256) {
        float Y = y[i], G = g[i], V = v[i], Z = z[i];
        upd(Y, G, V, Z, c);
        y[i] = Y; v[i] = V; z[i] = Z;
        if (c.write_grad) g[i] = G;
    }
}

}  // namespace

hipError_t launch_adamw_sf(const ftc_mt_chunk* chunks, int n_chunks, float beta2, float one_minus_beta2, float bias_correction2, float eps,
                           float decay, float ckp1, float y_alpha, float lr, int write_grad, hipStream_t s) {
    Scal c{beta2, one_minus_beta2, bias_correction2, eps, decay, ckp1, y_alpha, lr, write_grad};
    hipLaunchKernelGGL(adamw_sf_kernel, dim3(n_chunks), dim3(256), 0, s, chunks, c);
    return hipGetLastError();
}
