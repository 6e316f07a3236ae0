// Shared device helpers for the gfx950 kernels (wave64, MFMA, NHWC).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ftc.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// 16-bit storage / MFMA operand types: bf16 (speed mode) and IEEE half (fp16 mode).  Host-side helpers on ftc_dtype values.
inline bool ftc_is16(int dt) { return dt == FTC_BF16 || dt == FTC_F16; }
inline int ftc_esize(int dt) { return dt == FTC_F32 ? 4 : 2; }
inline const char* ftc_dtname(int dt) { return dt == FTC_F32 ? "f32" : dt == FTC_BF16 ? "bf16" : "f16"; }
// The 16-bit type that goes with a compute type: trunk copies (`out2`) and 16-bit residuals use it.
template <typename WT> struct Half16 { using type = __bf16; };
template <> struct Half16<_Float16> { using type = _Float16; };
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// Resolved (absolute-pointer) form of an ftc_op, what the launchers consume.
struct OpArgs {
    const ftc_op* op;
    const void* in;
    const void* in2;
    void* out;
    const void* w;
    const void* w2;
    const float* bias;
    const float* bias2;
    const float* scale;
    const float* shift;
    float* aux;
    void* out2;
};

__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f32_to_bf16(float v) { return (__bf16)v; }   // RNE (v_cvt_pk_bf16_f32)

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ float to_f32<_Float16>(_Float16 v) { return (float)v; }
// fp16 stores saturate (the format tops out at 65504; bf16 / fp32 activations of the same network do not overflow)
__device__ __forceinline__ float f16_sat(float v) { return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
template <> __device__ __forceinline__ _Float16 from_f32<_Float16>(float v) { return (_Float16)f16_sat(v); }

// 4 consecutive elements <-> float4
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<__bf16>(const __bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
template <> __device__ __forceinline__ f32x4 load4<_Float16>(const _Float16* p) {
    f16x4 v = *reinterpret_cast<const f16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, f32x4 v) {
    bf16x4 r = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = r;
}

template <> __device__ __forceinline__ void store4<_Float16>(_Float16* p, f32x4 v) {
    f16x4 r = {(_Float16)f16_sat(v[0]), (_Float16)f16_sat(v[1]), (_Float16)f16_sat(v[2]), (_Float16)f16_sat(v[3])};
    *reinterpret_cast<f16x4*>(p) = r;
}

// V consecutive elements (16 bytes of storage: V = 4 fp32 | 8 bf16) <-> V floats
template <typename T> struct vec16 { static constexpr int V = 16 / (int)sizeof(T); };
template <typename T> __device__ __forceinline__ void load16(const T* p, float (&f)[16 / sizeof(T)]);
template <> __device__ __forceinline__ void load16<float>(const float* p, float (&f)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = v[e];
}
template <> __device__ __forceinline__ void load16<__bf16>(const __bf16* p, float (&f)[8]) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(v[e] << 16);
        f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void load16<_Float16>(const _Float16* p, float (&f)[8]) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
template <typename T> __device__ __forceinline__ void store16(T* p, const float (&f)[16 / sizeof(T)]);
template <> __device__ __forceinline__ void store16<float>(float* p, const float (&f)[4]) {
    f32x4 v = {f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(p) = v;
}
template <> __device__ __forceinline__ void store16<__bf16>(__bf16* p, const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)f[e];
    *reinterpret_cast<bf16x8*>(p) = v;
}

template <> __device__ __forceinline__ void store16<_Float16>(_Float16* p, const float (&f)[8]) {
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)f16_sat(f[e]);
    *reinterpret_cast<f16x8*>(p) = v;
}

// Activations.  SiLU = x*sigmoid(x); GELU = exact erf form (nn.GELU default, detector.py:169).
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float act_silu_precise(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoid_precise(float x) { return 1.0f / (1.0f + expf(-x)); }

// Fast forms for the bf16 speed mode (a conv epilogue applies the activation to 64-96 values per
// lane; libm expf/erff made the epilogue cost as much VALU time as ~200 MFMAs):
//   SiLU  = x * rcp(1 + exp2(-x*log2 e))                       (v_exp_f32 + v_rcp_f32, ~3e-7 rel)
//   GELU  = 0.5 x (1 + erf(x/sqrt2)), erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 abs)
__device__ __forceinline__ float act_silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float act_gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = 1.0f - poly * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // erf(|x|/sqrt2)
    return 0.5f * x + 0.5f * fabsf(x) * e;                     // x*erf(x/sqrt2) = |x|*erf(|x|/sqrt2)
}
template <bool FAST> __device__ __forceinline__ float apply_act_sel(float x, int act) {
    if (act == FTC_ACT_SILU) return FAST ? act_silu_fast(x) : act_silu_precise(x);
    if (act == FTC_ACT_GELU) return FAST ? act_gelu_fast(x) : act_gelu(x);
    return x;
}

// Four values at a time, written on vectors so that the full-rate part of the polynomial becomes packed fp32 instructions
// (v_pk_mul_f32 / v_pk_fma_f32: two values per issue slot); the rcp / exp2 stay one lane-op per value.  A conv epilogue
// evaluates 64-96 activations per lane and is VALU-bound on them.
__device__ __forceinline__ f32x4 act_gelu_fast4(f32x4 x) {
    f32x4 ax, t, ex;
#pragma unroll
    for (int e = 0; e < 4; ++e) ax[e] = fabsf(x[e]);
    const f32x4 z = ax * 0.70710678118654752440f;
    const f32x4 d = z * 0.3275911f + 1.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_rcpf(d[e]);
    f32x4 poly = t * 1.061405429f - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const f32x4 zz = (z * -1.4426950408889634f) * z;
#pragma unroll
    for (int e = 0; e < 4; ++e) ex[e] = __builtin_amdgcn_exp2f(zz[e]);
    const f32x4 er = 1.0f - (poly * t) * ex;
    return x * 0.5f + (ax * 0.5f) * er;
}
__device__ __forceinline__ f32x4 act_silu_fast4(f32x4 x) {
    const f32x4 a = x * -1.4426950408889634f;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a[e]));
    return x * r;
}
template <bool FAST> __device__ __forceinline__ f32x4 apply_act4(f32x4 v, int act) {
    if constexpr (FAST) {
        if (act == FTC_ACT_SILU) return act_silu_fast4(v);
        if (act == FTC_ACT_GELU) return act_gelu_fast4(v);
        return v;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act_sel<false>(v[e], act);
        return v;
    }
}

template <int ACT> __device__ __forceinline__ float apply_act(float x) {
    if constexpr (ACT == FTC_ACT_SILU) return act_silu_precise(x);
    else if constexpr (ACT == FTC_ACT_GELU) return act_gelu(x);
    else return x;
}
__device__ __forceinline__ float apply_act_rt(float x, int act) {
    if (act == FTC_ACT_SILU) return act_silu_precise(x);
    if (act == FTC_ACT_GELU) return act_gelu(x);
    return x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Host-side launch entry points (one per .hip file); return hipError_t of the launch.
hipError_t launch_stem(const OpArgs& a, hipStream_t s);
hipError_t launch_conv(const OpArgs& a, hipStream_t s);
hipError_t launch_dwconv(const OpArgs& a, hipStream_t s);
hipError_t launch_bnstat(const OpArgs& a, hipStream_t s);
hipError_t launch_bnact(const OpArgs& a, hipStream_t s);
// fpn_ops.hip: 3x3 convolutions with 1..4 output channels on fp32 tensors (vector units; see thin_conv3x3_kernel)
bool ftc_thin_conv_legal(const ftc_op& o);
hipError_t launch_thin_conv(const OpArgs& a, hipStream_t s);
// FTC_OP_BNSTAT: number of row chunks the partial sums are split into
// (64 rows per chunk up to 512 chunks: a 24x24-map layer has 4608 rows -- with 256-row chunks its partial pass was 200 workgroups of 64
//  serial loads each, slower than the streaming it does)
inline int ftc_bnstat_chunks(long M) { const long n = (M + 63) / 64; return (int)(n < 1 ? 1 : n > 512 ? 512 : n); }
// row chunks of the depthwise weight-gradient and column-sum partial passes (256 rows each)
inline int ftc_chunks256(long M) { const long n = (M + 255) / 256; return (int)(n < 1 ? 1 : n > 512 ? 512 : n); }
inline int ftc_stemwgrad_chunks(long M) { const long n = (M + 255) / 256; return (int)(n < 1 ? 1 : n > 2048 ? 2048 : n); }
// train step (bwd_ops.hip, wgrad.hip)
hipError_t launch_bnbwd(const OpArgs& a, hipStream_t s);
hipError_t launch_wgrad(const OpArgs& a, hipStream_t s);
hipError_t launch_dwbwd(const OpArgs& a, hipStream_t s);
hipError_t launch_sebwd(const OpArgs& a, hipStream_t s);
hipError_t launch_upcatbwd(const OpArgs& a, hipStream_t s);
hipError_t launch_dilate(const OpArgs& a, hipStream_t s);
hipError_t launch_topdgrad(const OpArgs& a, hipStream_t s);
hipError_t launch_colsum(const OpArgs& a, hipStream_t s);
hipError_t launch_stemwgrad(const OpArgs& a, hipStream_t s);
hipError_t launch_fill(const OpArgs& a, hipStream_t s);
hipError_t launch_gather_rows_op(const OpArgs& a, hipStream_t s);
hipError_t launch_scatter_rows(const OpArgs& a, hipStream_t s);
hipError_t launch_loss_bwd(const OpArgs& a, hipStream_t s);
hipError_t launch_pack_train(const ftc_pack_entry* entries, int n, long max_elems, hipStream_t s);
int ftc_wgrad_splits_impl(int B, int Ho, int Wo, int Cout, int Cin, int ksize);
hipError_t launch_se(const OpArgs& a, hipStream_t s);
// mbconv_slice.hip: FTC_OP_MBHEAD (expand 1x1 + depthwise 3x3 + squeeze, one image x 64 channels per workgroup)
bool ftc_mbhead_legal(const ftc_op& o);
int ftc_mbhead_bands(const ftc_op& o);
int ftc_mbhead_band_rows(int H, int W);
int ftc_mbhead_slice(const ftc_op& o);         // expanded channels per workgroup (ftc_op.Cout_total, or the form's default)
hipError_t launch_mbhead(const OpArgs& a, hipStream_t s);
hipError_t launch_fmbconv(const OpArgs& a, hipStream_t s);      // fused_mbconv.hip (FTC_OP_FMBCONV)
const char* ftc_fmbconv_label(const ftc_op& o, char* buf, int len);
hipError_t launch_upcat(const OpArgs& a, hipStream_t s);
hipError_t launch_nms(const OpArgs& a, hipStream_t s);
hipError_t launch_tapsum(const OpArgs& a, hipStream_t s);
const char* conv_validate(const ftc_op& op);   // NULL if supported, else reason
void conv_kernel_label(const ftc_op& op, char* buf, int len);
