// Dense 1x1 / 3x3 convolution as an im2col-free implicit GEMM on the gfx950 matrix cores.
//
// Covers every dense convolution on the detector path: the Fused-MBConv
// 3x3 and 1x1 convs, the MBConv expand / project 1x1 convs, the backbone head conv, the FPN 3x3
// convs and the `top_conv`s (reference: torchvision blocks instantiated by
// /root/reference/models/detector.py:12-28, Leafmap layers :164-190).
//
//   D[n][m] = sum_{tap,c} W[n][tap][c] * X[pixel(m) shifted by tap][c]
//
// GEMM view: rows n = output channels (MFMA "A" operand = weights, K-major [Cout][k*k][Cin]),
// columns m = output pixels (MFMA "B" operand = NHWC activations, K-contiguous per pixel), so the
// MFMA C/D layout gives every lane 4 CONSECUTIVE output channels of one pixel per register quad:
// the NHWC epilogue (bias, activation, residual, store) is vectorised 4 wide with no shuffles.
//
// fp32 mode  : v_mfma_f32_32x32x2_f32  (exact f32 FMA chain, 157 TF peak) -- the parity mode.
// bf16 mode  : v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 2.5 PF peak)    -- the speed mode.
// Staging is global -> registers -> LDS (so out-of-image taps are zero-filled, the SE scale of
// the MBConv project conv is applied on the fly and fp32 trunk activations can be narrowed to
// bf16), double-buffered in LDS with one barrier per 32-deep K step; LDS rows are padded by one
// 16-byte chunk which makes both the ds_write_b128 and the fragment ds_read_b128 conflict-free.
#include <cstdio>

#include "ftc_common.h"

namespace {

struct ConvP {
    const void* in;
    const void* w;
    const float* bias;
    const void* res;
    void* out;
    const float* se;
    int B, H, W, Ho, Wo;
    int Cin, CinT, cin_off;
    int Cout, CoutT, cout_off;
    int KS, stride, pad;
    int act, flags, res_dtype;
    int M;      // B*Ho*Wo
    int ncb;    // ceil(Cin / 32)
    int nk;     // KS*KS*ncb
    int nN;     // channel tiles
    int nblk;   // total workgroups
};

constexpr int BK = 32;

template <typename WT> struct Frag;
template <> struct Frag<float> { using type = f32x4; };
template <> struct Frag<__bf16> { using type = bf16x8; };

__device__ __forceinline__ u32x4 zero16() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

// Load one 16-byte LDS chunk worth of K (E elements of WT) for a pixel row from `src` (InT).
template <typename WT, typename InT>
__device__ __forceinline__ u32x4 load_chunk(const InT* src, const float* se) {
    if constexpr (sizeof(WT) == 4) {
        static_assert(sizeof(InT) == 4, "fp32 compute takes fp32 activations");
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
        if (se) { f32x4 s = *reinterpret_cast<const f32x4*>(se); v *= s; }
        return __builtin_bit_cast(u32x4, v);
    } else {
        float f[8];
        if constexpr (sizeof(InT) == 4) {
            f32x4 lo = reinterpret_cast<const f32x4*>(src)[0];
            f32x4 hi = reinterpret_cast<const f32x4*>(src)[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
        } else {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
            if (!se) return __builtin_bit_cast(u32x4, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
        }
        if (se) {
            f32x4 s0 = reinterpret_cast<const f32x4*>(se)[0];
            f32x4 s1 = reinterpret_cast<const f32x4*>(se)[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] *= s0[e]; f[4 + e] *= s1[e]; }
        }
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (__bf16)f[e];
        return __builtin_bit_cast(u32x4, r);
    }
}

template <typename WT, typename InT, typename OutT, int WN, int WM, int SN, int SM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
    constexpr int E = 16 / (int)sizeof(WT);      // elements per 16-byte chunk (4 fp32 | 8 bf16)
    constexpr int CPR = BK / E;                  // chunks per LDS row (8 | 4)
    constexpr int ROW = BK + E;                  // padded LDS row, elements (144 B | 80 B)
    constexpr int TN = WN * SN * 32;             // output channels per workgroup
    constexpr int TM = WM * SM * 32;             // output pixels per workgroup
    constexpr int NA = (TN * CPR + 255) / 256;
    constexpr int NB = (TM * CPR + 255) / 256;
    constexpr int RPP = 256 / CPR;               // rows covered per staging pass
    static_assert(WN * WM == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    WT* lds = reinterpret_cast<WT*>(smem_raw);
    constexpr int BUF = (TN + TM) * ROW;         // elements per LDS buffer

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const int half = lane >> 5, l31 = lane & 31;

    // XCD-aware remap (block b runs on XCD b % 8): give each XCD a contiguous run of tiles so
    // that the 9 taps / the channel tiles of neighbouring pixel tiles hit the same private L2.
    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int mt = bid / p.nN, nt = bid - mt * p.nN;
    const int m0 = mt * TM, n0 = nt * TN;

    const int kc = t % CPR;                      // this thread's 16-byte chunk inside a K row
    const int row0 = t / CPR;
    const int HoWo = p.Ho * p.Wo;

    int b_iy0[NB], b_ix0[NB], b_pix[NB], b_img[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = row0 + i * RPP;
        const int m = m0 + row;
        const bool ok = (row < TM) && (m < p.M);
        const int mm = ok ? m : 0;
        const int img = mm / HoWo;
        const int rem = mm - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        b_iy0[i] = ok ? oy * p.stride - p.pad : -100000;   // makes every tap out of range
        b_ix0[i] = ox * p.stride - p.pad;
        b_pix[i] = img * p.H * p.W;
        b_img[i] = img;
    }

    const WT* __restrict__ wgt = reinterpret_cast<const WT*>(p.w);
    const InT* __restrict__ inp = reinterpret_cast<const InT*>(p.in);
    const int KK = p.KS * p.KS;
    const bool use_se = (p.flags & FTC_FLAG_SE_SCALE) != 0;

    u32x4 ra[NA], rb[NB];

    auto gload = [&](int it) {
        const int tap = it / p.ncb;
        const int cb = it - tap * p.ncb;
        const int r = tap / p.KS, s = tap - r * p.KS;
        const int c = cb * BK + kc * E;
        const bool cok = c < p.Cin;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = row0 + i * RPP;
            const int n = n0 + row;
            const bool ok = cok && (row < TN) && (n < p.Cout);
            ra[i] = ok ? *reinterpret_cast<const u32x4*>(wgt + ((size_t)(n * KK + tap) * p.Cin + c)) : zero16();
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int iy = b_iy0[i] + r, ix = b_ix0[i] + s;
            const bool ok = cok && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
            if (ok) {
                const InT* src = inp + ((size_t)(b_pix[i] + iy * p.W + ix) * p.CinT + p.cin_off + c);
                const float* se = use_se ? p.se + (size_t)b_img[i] * p.Cin + c : nullptr;
                rb[i] = load_chunk<WT, InT>(src, se);
            } else {
                rb[i] = zero16();
            }
        }
    };
    auto lds_write = [&](int buf) {
        WT* A = lds + buf * BUF;
        WT* Bm = A + TN * ROW;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = row0 + i * RPP;
            if (row < TN) *reinterpret_cast<u32x4*>(A + row * ROW + kc * E) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = row0 + i * RPP;
            if (row < TM) *reinterpret_cast<u32x4*>(Bm + row * ROW + kc * E) = rb[i];
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
    auto compute = [&](int buf) {
        const WT* A = lds + buf * BUF + (wn * SN * 32 + l31) * ROW;
        const WT* Bm = lds + buf * BUF + (TN + wm * SM * 32 + l31) * ROW;
        if constexpr (sizeof(WT) == 4) {
            // 8 k per group: lanes 0-31 hold k = 0..3, lanes 32-63 hold k = 4..7 of the group;
            // step tt feeds A[:,k=tt | 4+tt], B likewise -- same permutation on both operands.
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                FragT af[SN], bf[SM];
#pragma unroll
                for (int i = 0; i < SN; ++i) af[i] = *reinterpret_cast<const FragT*>(A + i * 32 * ROW + g * 8 + half * 4);
#pragma unroll
                for (int j = 0; j < SM; ++j) bf[j] = *reinterpret_cast<const FragT*>(Bm + j * 32 * ROW + g * 8 + half * 4);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < SN; ++i)
#pragma unroll
                        for (int j = 0; j < SM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][tt], bf[j][tt], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < BK / 16; ++g) {
                FragT af[SN], bf[SM];
#pragma unroll
                for (int i = 0; i < SN; ++i) af[i] = *reinterpret_cast<const FragT*>(A + i * 32 * ROW + g * 16 + half * 8);
#pragma unroll
                for (int j = 0; j < SM; ++j) bf[j] = *reinterpret_cast<const FragT*>(Bm + j * 32 * ROW + g * 16 + half * 8);
#pragma unroll
                for (int i = 0; i < SN; ++i)
#pragma unroll
                    for (int j = 0; j < SM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    gload(0);
    lds_write(0);
    __syncthreads();
    for (int it = 0; it < p.nk; ++it) {
        const bool more = it + 1 < p.nk;
        if (more) gload(it + 1);
        compute(it & 1);
        if (more) lds_write((it + 1) & 1);
        __syncthreads();
    }

    // Epilogue: lane owns pixel (l31) of each 32-pixel sub-tile and, per register quad q,
    // channels 8q + 4*half .. +3 of each 32-channel sub-tile (C/D layout of the 32x32 MFMA).
    OutT* __restrict__ outp = reinterpret_cast<OutT*>(p.out);
    const bool has_res = (p.flags & FTC_FLAG_RESIDUAL) != 0;
    const bool vec_ok = ((p.Cout | p.CoutT | p.cout_off) & 3) == 0;
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int m = m0 + wm * SM * 32 + j * 32 + l31;
        if (m >= p.M) continue;
        OutT* orow = outp + (size_t)m * p.CoutT + p.cout_off;
#pragma unroll
        for (int i = 0; i < SN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * SN * 32 + i * 32 + 8 * q + 4 * half;
                if (n >= p.Cout) continue;
                if (vec_ok) {
                    f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    v += *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act_rt(v[e], p.act);
                    if (has_res) {
                        if (p.res_dtype == FTC_F32) v += load4<float>(reinterpret_cast<const float*>(p.res) + (size_t)m * p.Cout + n);
                        else v += load4<__bf16>(reinterpret_cast<const __bf16*>(p.res) + (size_t)m * p.Cout + n);
                    }
                    store4<OutT>(orow + n, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e >= p.Cout) continue;
                        float v = acc[i][j][4 * q + e] + p.bias[n + e];
                        v = apply_act_rt(v, p.act);
                        if (has_res) {
                            if (p.res_dtype == FTC_F32) v += reinterpret_cast<const float*>(p.res)[(size_t)m * p.Cout + n + e];
                            else v += (float)reinterpret_cast<const __bf16*>(p.res)[(size_t)m * p.Cout + n + e];
                        }
                        orow[n + e] = from_f32<OutT>(v);
                    }
                }
            }
        }
    }
}

template <typename WT, typename InT, typename OutT, int WN, int WM, int SN, int SM>
hipError_t launch_cfg(ConvP p, hipStream_t s) {
    constexpr int E = 16 / (int)sizeof(WT);
    constexpr int ROW = BK + E;
    constexpr int TN = WN * SN * 32, TM = WM * SM * 32;
    constexpr size_t lds_bytes = (size_t)2 * (TN + TM) * ROW * sizeof(WT);
    auto kern = conv_igemm_kernel<WT, InT, OutT, WN, WM, SN, SM>;
    static bool attr_set = false;     // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.nN = (p.Cout + TN - 1) / TN;
    const int nM = (p.M + TM - 1) / TM;
    p.nblk = p.nN * nM;
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(256), lds_bytes, s, p);
    return hipGetLastError();
}

// Tile configuration by output-channel count and problem size (channels x pixels per workgroup).
enum { CFG_32x256 = 0, CFG_64x128, CFG_96x128, CFG_192x128, CFG_128x64, CFG_128x128 };
const char* const kCfgName[] = {"32x256", "64x128", "96x128", "192x128", "128x64", "128x128"};

int select_cfg(int n, int M) {
    if (n <= 32) return CFG_32x256;
    if (n <= 64) return CFG_64x128;
    if (n <= 96) return CFG_96x128;
    if (n % 192 == 0 && n % 128 != 0) return CFG_192x128;
    // 128-channel tiles; shrink the pixel tile when the grid would not fill the 256 CUs twice
    const long tiles128 = (long)((n + 127) / 128) * ((M + 127) / 128);
    return tiles128 < 512 ? CFG_128x64 : CFG_128x128;
}

template <typename WT, typename InT, typename OutT>
hipError_t launch_types(const ConvP& p, hipStream_t s) {
    switch (select_cfg(p.Cout, p.M)) {
    case CFG_32x256: return launch_cfg<WT, InT, OutT, 1, 4, 1, 2>(p, s);
    case CFG_64x128: return launch_cfg<WT, InT, OutT, 2, 2, 1, 2>(p, s);
    case CFG_96x128: return launch_cfg<WT, InT, OutT, 1, 4, 3, 1>(p, s);
    case CFG_192x128: return launch_cfg<WT, InT, OutT, 2, 2, 3, 2>(p, s);
    case CFG_128x64: return launch_cfg<WT, InT, OutT, 2, 2, 2, 1>(p, s);
    default: return launch_cfg<WT, InT, OutT, 2, 2, 2, 2>(p, s);
    }
}

}  // namespace

void conv_kernel_label(const ftc_op& op, char* buf, int len) {
    const char* dt[] = {"f32", "bf16"};
    snprintf(buf, len, "conv_igemm<%s,in=%s,out=%s,tile=%s>", dt[op.w_dtype & 1], dt[op.in_dtype & 1], dt[op.out_dtype & 1],
             kCfgName[select_cfg(op.Cout, op.B * op.Ho * op.Wo)]);
}

const char* conv_validate(const ftc_op& op) {
    if (op.ksize != 1 && op.ksize != 3) return "conv: ksize must be 1 or 3";
    if (op.stride != 1 && op.stride != 2) return "conv: stride must be 1 or 2";
    const int E = op.w_dtype == FTC_F32 ? 4 : 8;
    const int Ein = op.in_dtype == FTC_F32 ? 4 : 8;
    if (op.w_dtype == FTC_F32 && (op.in_dtype != FTC_F32)) return "conv: fp32 compute needs fp32 input";
    if (op.Cin % E) return "conv: Cin must be a multiple of the 16-byte chunk";
    if (op.Cin_total % Ein || op.cin_off % E) return "conv: input channel stride/offset not 16-byte aligned";
    if (op.cin_off + op.Cin > op.Cin_total) return "conv: input channel slice out of range";
    if (op.cout_off + op.Cout > op.Cout_total) return "conv: output channel slice out of range";
    const int pad = (op.ksize - 1) / 2;
    if (op.Ho != (op.H + 2 * pad - op.ksize) / op.stride + 1 || op.Wo != (op.W + 2 * pad - op.ksize) / op.stride + 1)
        return "conv: Ho/Wo inconsistent with H/W/ksize/stride";
    if ((op.flags & FTC_FLAG_SE_SCALE) && op.ksize != 1) return "conv: SE scale only on 1x1";
    if ((long)op.B * op.Ho * op.Wo > 0x7fffffffL / 4) return "conv: too many output pixels";
    return nullptr;
}

hipError_t launch_conv(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    ConvP p;
    p.in = a.in; p.w = a.w; p.bias = a.bias; p.res = a.in2; p.out = a.out; p.se = a.scale;
    p.B = o.B; p.H = o.H; p.W = o.W; p.Ho = o.Ho; p.Wo = o.Wo;
    p.Cin = o.Cin; p.CinT = o.Cin_total; p.cin_off = o.cin_off;
    p.Cout = o.Cout; p.CoutT = o.Cout_total; p.cout_off = o.cout_off;
    p.KS = o.ksize; p.stride = o.stride; p.pad = (o.ksize - 1) / 2;
    p.act = o.act; p.flags = o.flags; p.res_dtype = o.res_dtype;
    p.M = o.B * o.Ho * o.Wo;
    p.ncb = (o.Cin + BK - 1) / BK;
    p.nk = o.ksize * o.ksize * p.ncb;
    p.nN = 0; p.nblk = 0;
    if (o.w_dtype == FTC_F32) {
        if (o.out_dtype == FTC_F32) return launch_types<float, float, float>(p, s);
        return hipErrorInvalidValue;
    }
    if (o.in_dtype == FTC_BF16 && o.out_dtype == FTC_BF16) return launch_types<__bf16, __bf16, __bf16>(p, s);
    if (o.in_dtype == FTC_F32 && o.out_dtype == FTC_BF16) return launch_types<__bf16, float, __bf16>(p, s);
    if (o.in_dtype == FTC_BF16 && o.out_dtype == FTC_F32) return launch_types<__bf16, __bf16, float>(p, s);
    return launch_types<__bf16, float, float>(p, s);
}
