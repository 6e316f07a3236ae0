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
//
// Staging is global -> registers -> LDS (out-of-image taps zero-filled, the SE scale of the MBConv
// project conv applied on the fly, fp32 trunk activations narrowed to bf16 on the way).  All
// global reads are raw buffer loads: the per-row byte offset lives in one VGPR, the K position
// (tap, channel block) is wave-uniform and goes in the SGPR offset / one scalar add, and rows or
// taps outside the image are redirected to an out-of-range offset which the buffer unit answers
// with zeros -- about 4 VALU instructions per 16-byte load instead of a 64-bit address chain
// (the first version of this kernel issued 13.7 VALU per MFMA; PMC in profiles/).
// LDS rows are padded by one 16-byte chunk: conflict-free ds_read_b128 fragments.
#pragma once
#include <cstdio>
#include <type_traits>

#include "ftc_common.h"

namespace convimpl {

struct ConvP {
    const void* in;
    const void* w;
    const float* bias;
    const void* res;
    void* out;
    void* out2;
    const float* se;
    unsigned in_bytes, w_bytes, se_bytes;
    int B, H, W, Ho, Wo;
    int Cin, CinT, cin_off;
    int Cout, CoutT, cout_off;
    int KS, stride, pad;
    int act, flags, res_dtype;
    int M;      // B*Ho*Wo
    int ncb;    // ceil(Cin / BK)
    int nk;     // KS*KS*ncb
    int nN;     // channel tiles
    int nblk;   // total workgroups
    int use_glds;   // direct-to-LDS kernel selected (uses_glds)
    int glds_nbuf;  // tuning: LDS ring depth of the DMA kernel (2 | 3)
    int split_k;    // tuning: K groups per workgroup of the register-staged kernel (1 | 2 | 4)
    int wset_bytes; // FTC_FLAG_W_PER_IMAGE: bytes between the weight sets of consecutive images (0 = one shared set)
    // grouped launch (ftc_op.groups): G independent instances, workgroups [g*nblk_g, (g+1)*nblk_g) belong to instance g
    int groups, nblk_g;
    long in_gs, w_gs, out_gs, out2_gs;   // bytes between the instances' operands
    int bias_gs, cout_gs;                // floats between bias tables; channel-slice step of the output (OUT_SLICE)
    // FTC_FLAG_TOP_FUSE: the 32 x Cout tap matrix of the following top convolution and the width of its output rows
    const void* w2;
    long w2_gs;
    int Tw;
    // FTC_FLAG_UPCAT_IN: channels [0, Cy) of the input are the x2 bilinear upsample (align_corners) of `in`
    // [B,Hi,Wi,Cy], computed while the halo is staged; channels [Cy, Cin) come from in2u [B,H,W,Cin-Cy]
    const void* in2u;
    unsigned in2u_bytes;
    long in2u_gs;
    int Cy, Hi, Wi;
    float ry, rx;
    int epi_off;    // weights-through-L1 kernel: LDS byte offset of the epilogue image (behind halo buffer 0 when both fit, else 0)
};

// Turns the launch-wide parameter block into the one of the group that owns workgroup `bid`; returns the
// workgroup index inside the group.
__device__ __forceinline__ int enter_group(ConvP& p, int bid) {
    if (p.groups <= 1) return bid;
    const int g = bid / p.nblk_g;
    p.in = static_cast<const char*>(p.in) + g * p.in_gs;
    p.w = static_cast<const char*>(p.w) + g * p.w_gs;
    p.bias += (long)g * p.bias_gs;
    p.out = static_cast<char*>(p.out) + g * p.out_gs;
    if (p.out2) p.out2 = static_cast<char*>(p.out2) + g * p.out2_gs;
    p.cout_off += g * p.cout_gs;
    if (p.w2) p.w2 = static_cast<const char*>(p.w2) + g * p.w2_gs;
    if (p.in2u) p.in2u = static_cast<const char*>(p.in2u) + g * p.in2u_gs;      // (stride 0: one tensor shared by the groups)
    return bid - g * p.nblk_g;
}

// Weight descriptor of the workgroup whose first output row is m0 (all its rows are in one image when wset_bytes != 0).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const ConvP& p, int m0) {
    const char* w = static_cast<const char*>(p.w);
    if (p.wset_bytes) w += (long)(m0 / (p.Ho * p.Wo)) * p.wset_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, p.w_bytes, 0x00020000);
}

// Compute-type tag of the fp16x3 mode (FTC_FLAG_SPLIT16): staged, stored and addressed exactly like fp32 (4-byte elements), multiplied
// as hi / lo IEEE halves.  Its own tag = its own kernel instantiations: the exact-fp32 kernels keep their register budget.
struct x3f32 { float v; };
template <typename WT> constexpr bool is_x3 = std::is_same<WT, x3f32>::value;

template <typename WT> struct Frag;
template <> struct Frag<float> { using type = f32x4; };
template <> struct Frag<x3f32> { using type = f32x4; };
template <> struct Frag<__bf16> { using type = bf16x8; };
template <> struct Frag<_Float16> { using type = f16x8; };

// 32x32x16 MFMA on the two 16-bit operand formats (same rate, fp32 accumulation)
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// FTC_FLAG_SPLIT16 ("fp16x3": fp32 tensors, weights and accumulation; the products on the 16-bit matrix pipe).  Two fp32 fragments of
// four K values (two K groups of the fp32 kernels: the lower half-wave holds k = 0..3 | 8..11, the upper 4..7 | 12..15 -- the same
// permutation on both operands) become the hi / lo halves of ONE 32x32x16 operand: hi = fp16(x) (clamped to the format's range),
// lo = fp16(x - hi): 22 significand bits.  A.B ~= Ahi.Blo + Alo.Bhi + Ahi.Bhi (the dropped lo.lo term is 2^-24 relative): 3 MFMAs of
// 32 cycles for 16 K values instead of 8 fp32 MFMAs of 64 cycles.
__device__ __forceinline__ void split16(const f32x4& a0, const f32x4& a1, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h0 = (_Float16)f16_sat(a0[e]), h1 = (_Float16)f16_sat(a1[e]);
        hi[e] = h0;
        hi[4 + e] = h1;
        lo[e] = (_Float16)(a0[e] - (float)h0);
        lo[4 + e] = (_Float16)(a1[e] - (float)h1);
    }
}
// One 16-byte chunk of four fp32 values <-> the same 16 bytes as [hi x4 | lo x4] IEEE halves ("pre-split" chunk).  The WEIGHTS of an
// fp16x3 convolution are stored pre-split (model.hip packs them so; the K order is untouched), activations are converted where they pass
// through registers anyway (register-staged kernel) or once per LDS-resident halo (3x3 halo kernel); only the DMA-staged activation
// tiles of conv_igemm_glds_kernel are still split per fragment read.
__device__ __forceinline__ u32x4 chunk_hl(const f32x4& v) {
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 hh = (_Float16)f16_sat(v[e]);
        h[e] = hh;
        l[e] = (_Float16)(v[e] - (float)hh);
    }
    const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
    return u32x4{hu[0], hu[1], lu[0], lu[1]};
}
// two pre-split chunks (K groups g, g+1 of the fp32 kernels) -> the hi / lo operands of one 32x32x16 MFMA: register renaming only
__device__ __forceinline__ void frag_hl(const f32x4& c0, const f32x4& c1, f16x8& hi, f16x8& lo) {
    const u32x4 a = __builtin_bit_cast(u32x4, c0), b = __builtin_bit_cast(u32x4, c1);
    hi = __builtin_bit_cast(f16x8, u32x4{a[0], a[1], b[0], b[1]});
    lo = __builtin_bit_cast(f16x8, u32x4{a[2], a[3], b[2], b[3]});
}
__device__ __forceinline__ f32x16 mfma_split(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
}

constexpr int OOB = 0x7ffffff0;      // byte offset beyond any buffer: the load returns zeros

__device__ __forceinline__ u32x4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// One 16-byte LDS chunk worth of K (E elements of WT) of an activation row, read as InT.
template <typename WT, typename InT>
__device__ __forceinline__ u32x4 load_act(__amdgpu_buffer_rsrc_t rin, int voff, bool use_se, __amdgpu_buffer_rsrc_t rse, int seoff, bool presplit = false) {
    if constexpr (sizeof(WT) == 4) {
        static_assert(sizeof(InT) == 4, "fp32 compute takes fp32 activations");
        u32x4 raw = bload(rin, voff, 0);
        if (is_x3<WT> && presplit) return raw;                         // FTC_FLAG_PRESPLIT: the producer stored [hi x4 | lo x4] already (validated: no SE scale)
        if (use_se) {
            f32x4 v = __builtin_bit_cast(f32x4, raw) * __builtin_bit_cast(f32x4, bload(rse, seoff, 0));
            raw = __builtin_bit_cast(u32x4, v);
        }
        if constexpr (is_x3<WT>) raw = chunk_hl(__builtin_bit_cast(f32x4, raw));      // fp16x3: split once, on the way to LDS
        return raw;
    } else {
        using FragW = typename Frag<WT>::type;
        float f[8];
        if constexpr (sizeof(InT) == 4) {
            const f32x4 lo = __builtin_bit_cast(f32x4, bload(rin, voff, 0));
            const f32x4 hi = __builtin_bit_cast(f32x4, bload(rin, voff + 16, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
        } else {
            static_assert(std::is_same<InT, WT>::value, "16-bit activations come in the compute type");
            const u32x4 raw = bload(rin, voff, 0);
            if (!use_se) return raw;
            const FragW v = __builtin_bit_cast(FragW, raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
        }
        if (use_se) {
            const f32x4 s0 = __builtin_bit_cast(f32x4, bload(rse, seoff, 0));
            const f32x4 s1 = __builtin_bit_cast(f32x4, bload(rse, seoff + 16, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] *= s0[e]; f[4 + e] *= s1[e]; }
        }
        FragW r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = from_f32<WT>(f[e]);
        return __builtin_bit_cast(u32x4, r);
    }
}

// Element index of channel n (a multiple of 4) of output pixel m in the 16-bit trunk copy `out2`: NHWC, or 32-channel planes per image
// (FTC_FLAG_KBLOCK32: [B][Cout/32][Ho*Wo][32], what FTC_OP_MBHEAD streams as whole cache lines).
__device__ __forceinline__ size_t out2_index(const ConvP& p, int m, int n) {
    if (p.flags & FTC_FLAG_KBLOCK32) {
        const int hw = p.Ho * p.Wo;
        const int img = m / hw, r = m - img * hw;
        return (((size_t)img * (p.Cout >> 5) + (n >> 5)) * hw + r) * 32 + (n & 31);
    }
    return (size_t)m * p.Cout + n;
}

// The second copy of an fp32 output: the 16-bit trunk copy the 16-bit plans' next GEMM reads, or (fp16x3 plans, FTC_FLAG_SPLIT16) the
// PRE-SPLIT copy FTC_OP_MBHEAD streams by DMA -- the four fp32 values of a 16-byte chunk as [hi x4 | lo x4] IEEE halves, NHWC.
template <typename WT>
__device__ __forceinline__ void store_out2(const ConvP& p, int m, int n, const f32x4& v) {
    if constexpr (is_x3<WT>) {
        *reinterpret_cast<u32x4*>(static_cast<char*>(p.out2) + ((size_t)m * p.Cout + n) * 4) = chunk_hl(v);
    } else {
        store4<typename Half16<WT>::type>(reinterpret_cast<typename Half16<WT>::type*>(p.out2) + out2_index(p, m, n), v);
    }
}

// Epilogue shared by both kernels: lane owns pixel (l31) of each 32-pixel sub-tile and, per register
// quad q, channels 8q + 4*half .. +3 of each 32-channel sub-tile (C/D layout of the 32x32 MFMA).
template <typename WT, typename OutT, int SN, int SM>
__device__ __forceinline__ void conv_epilogue_rows(const ConvP& p, f32x16 (&acc)[SN][SM], const int (&mrow)[SM], int nbase, int half);

template <typename WT, typename OutT, int SN, int SM>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[SN][SM], int m0, int n0, int wn, int wm, int half, int l31) {
    int mrow[SM];
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int m = m0 + wm * SM * 32 + j * 32 + l31;
        mrow[j] = m < p.M ? m : -1;
    }
    conv_epilogue_rows<WT, OutT, SN, SM>(p, acc, mrow, n0 + wn * SN * 32, half);
}

// mrow[j] = flat output pixel index (b*Ho*Wo + oy*Wo + ox) this lane owns in sub-tile j, or -1.
template <typename WT, typename OutT, int SN, int SM>
__device__ __forceinline__ void conv_epilogue_rows(const ConvP& p, f32x16 (&acc)[SN][SM], const int (&mrow)[SM], int nbase, int half) {
    OutT* __restrict__ outp = reinterpret_cast<OutT*>(p.out);
    const bool has_res = (p.flags & FTC_FLAG_RESIDUAL) != 0;
    const bool vec_ok = ((p.Cout | p.CoutT | p.cout_off) & 3) == 0;
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int m = mrow[j];
        if (m < 0) continue;
        OutT* orow = outp + (size_t)m * p.CoutT + p.cout_off;
        const float* brow = p.bias;
        if (p.flags & FTC_FLAG_BORDER_BIAS) {
            const int rem = m % (p.Ho * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int idx = (oy == 0 ? 1 : 0) | (oy == p.Ho - 1 ? 2 : 0) | (ox == 0 ? 4 : 0) | (ox == p.Wo - 1 ? 8 : 0);
            brow += idx * p.Cout;
        }
#pragma unroll
        for (int i = 0; i < SN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nbase + i * 32 + 8 * q + 4 * half;
                if (n >= p.Cout) continue;
                if (vec_ok) {
                    f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    v += *reinterpret_cast<const f32x4*>(brow + n);
                    v = apply_act4<sizeof(WT) == 2>(v, p.act);
                    if (has_res) {
                        if (p.res_dtype == FTC_F32) v += load4<float>(reinterpret_cast<const float*>(p.res) + (size_t)m * p.Cout + n);
                        else v += load4<typename Half16<WT>::type>(reinterpret_cast<const typename Half16<WT>::type*>(p.res) + (size_t)m * p.Cout + n);
                    }
                    store4<OutT>(orow + n, v);
                    if constexpr (sizeof(OutT) == 4) {
                        if (p.out2) store_out2<WT>(p, m, n, v);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e >= p.Cout) continue;
                        float v = acc[i][j][4 * q + e] + brow[n + e];
                        v = apply_act_sel<sizeof(WT) == 2>(v, p.act);
                        if (has_res) {
                            if (p.res_dtype == FTC_F32) v += reinterpret_cast<const float*>(p.res)[(size_t)m * p.Cout + n + e];
                            else v += (float)reinterpret_cast<const typename Half16<WT>::type*>(p.res)[(size_t)m * p.Cout + n + e];
                        }
                        orow[n + e] = from_f32<OutT>(v);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged epilogue.  The MFMA C/D layout hands every lane 4-channel quads of 32 DIFFERENT pixels, so
// storing straight from the accumulators issues 64 scattered 8-byte (bf16) writes per instruction: the
// s_memtime timeline of a 192x256 tile showed 23k of 131k cycles spent in that store tail (store-issue
// bound, nothing overlaps it).  Instead each lane drops its biased/activated quads into an LDS image
// of the output tile ([pixel][channel], 16-byte chunks XOR-swizzled by pixel&7, free of the operand
// buffers after the K loop) and the workgroup then writes whole NHWC rows with 16-byte lanes; the
// residual add and the bf16 trunk copy ride on the same coalesced pass.
// Used when rows are 16-byte aligned in the output (else the direct path below).
// ------------------------------------------------------------------------------------------------
template <typename OutT> __host__ __device__ constexpr int epi_pitch(int tn) { return (tn * (int)sizeof(OutT) + 127) / 128 * 128; }

template <typename OutT> __device__ __forceinline__ bool epi_lds_ok(const ConvP& p) {
    constexpr int V = 16 / (int)sizeof(OutT);
    return ((p.Cout | p.CoutT | p.cout_off) % V) == 0 && (p.Cout % 8) == 0;
}

template <typename WT, typename OutT, int SN, int SM, int NTHREADS, int TN, int TM, typename RowFn>
__device__ __forceinline__ void conv_epilogue_lds(const ConvP& p, f32x16 (&acc)[SN][SM], unsigned char* smem, int n0, int nw0, int pw0,
                                                  int half, int l31, RowFn row_to_m) {
    constexpr int PITCH = epi_pitch<OutT>(TN);
    constexpr int V = 16 / (int)sizeof(OutT);               // channels per 16-byte chunk (4 fp32 | 8 bf16)
    constexpr int CH = TN / V;                              // chunks per pixel row
    __syncthreads();                                        // every wave is done with the operand buffers
    // bias rows of this channel tile -> LDS behind the output image (1 row, or the 16 border cases):
    // read from global lane by lane they were 4*SN serialized L1 round trips per pixel sub-tile.
    float* lbias = reinterpret_cast<float*>(smem + TM * PITCH);
    const int nrows = (p.flags & FTC_FLAG_BORDER_BIAS) ? 16 : 1;
    for (int c = threadIdx.x; c < nrows * (TN / 4); c += NTHREADS) {
        const int r = c / (TN / 4), q = c - r * (TN / 4);
        const int n = n0 + 4 * q;
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (n < p.Cout) b = *reinterpret_cast<const f32x4*>(p.bias + (size_t)r * p.Cout + n);
        *reinterpret_cast<f32x4*>(lbias + r * TN + 4 * q) = b;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int prow = pw0 + j * 32 + l31;                // pixel row inside the tile
        const float* brow = lbias;
        if (p.flags & FTC_FLAG_BORDER_BIAS) {
            const int m = row_to_m(prow);
            if (m >= 0) {
                const int rem = m % (p.Ho * p.Wo);
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                brow += ((oy == 0 ? 1 : 0) | (oy == p.Ho - 1 ? 2 : 0) | (ox == 0 ? 4 : 0) | (ox == p.Wo - 1 ? 8 : 0)) * TN;
            }
        }
        unsigned char* lrow = smem + prow * PITCH;
#pragma unroll
        for (int i = 0; i < SN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = nw0 + i * 32 + 8 * q + 4 * half;          // channel inside the tile
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(brow + nl);
                v = apply_act4<sizeof(WT) == 2>(v, p.act);
                const int chunk = (nl / V) ^ (prow & 7);
                store4<OutT>(reinterpret_cast<OutT*>(lrow + chunk * 16) + (nl % V), v);
            }
        }
    }
    __syncthreads();
    OutT* __restrict__ outp = reinterpret_cast<OutT*>(p.out);
    const bool has_res = (p.flags & FTC_FLAG_RESIDUAL) != 0;
    for (int c = threadIdx.x; c < TM * CH; c += NTHREADS) {
        const int prow = c / CH, cc = c - prow * CH;
        const int m = row_to_m(prow);
        const int n = n0 + cc * V;
        if (m < 0 || n >= p.Cout) continue;
        const u32x4 raw = *reinterpret_cast<const u32x4*>(smem + prow * PITCH + ((cc ^ (prow & 7)) * 16));
        if constexpr (sizeof(OutT) == 4) {
            f32x4 v = __builtin_bit_cast(f32x4, raw);
            if (has_res) {
                if (p.res_dtype == FTC_F32) v += load4<float>(reinterpret_cast<const float*>(p.res) + (size_t)m * p.Cout + n);
                else v += load4<typename Half16<WT>::type>(reinterpret_cast<const typename Half16<WT>::type*>(p.res) + (size_t)m * p.Cout + n);
            }
            *reinterpret_cast<f32x4*>(outp + (size_t)m * p.CoutT + p.cout_off + n) = v;
            if (p.out2) store_out2<WT>(p, m, n, v);
        } else {
            if (has_res) {
                float f[8], r[8];
                load16<OutT>(reinterpret_cast<const OutT*>(&raw), f);
                if (p.res_dtype == FTC_F32) {
                    const f32x4 r0 = load4<float>(reinterpret_cast<const float*>(p.res) + (size_t)m * p.Cout + n);
                    const f32x4 r1 = load4<float>(reinterpret_cast<const float*>(p.res) + (size_t)m * p.Cout + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { r[e] = r0[e]; r[4 + e] = r1[e]; }
                } else {
                    load16<OutT>(reinterpret_cast<const OutT*>(p.res) + (size_t)m * p.Cout + n, r);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += r[e];
                store16<OutT>(outp + (size_t)m * p.CoutT + p.cout_off + n, f);
            } else {
                *reinterpret_cast<u32x4*>(outp + (size_t)m * p.CoutT + p.cout_off + n) = raw;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FTC_FLAG_TOP_FUSE epilogue (bf16, the tile holds ALL Cout = TN channels of its pixels): the last FPN level of
// a map head is consumed only by that head's `top_conv` (3x3, 1-2 output channels).  A 3x3 convolution with tiny
// Cout is a per-pixel linear map followed by a 9-point sum:  T[p][tap*Co+o] = sum_c y[p][c] * Wtop[o][tap][c],
// out[p][o] = bias[o] + sum_tap T[p + d(tap)][tap*Co+o].  The first half is one more MFMA GEMM on the output tile
// while it sits in LDS (K = TN, 32 rows = 9*Co padded), so the 192-channel tensor (384 B/pixel) is never written:
// only T (aux1 floats per pixel) leaves the CU, and FTC_OP_TAPSUM does the 9-point sum.
// ------------------------------------------------------------------------------------------------
template <typename WT, int SN, int SM, int NTHREADS, int TN, int TM, typename RowFn>
__device__ __forceinline__ void conv_epilogue_topfuse(const ConvP& p, f32x16 (&acc)[SN][SM], unsigned char* smem, int nw0, int pw0,
                                                      int half, int l31, int lpix, int wave, RowFn row_to_m) {
    constexpr int PITCH = TN * 2;                           // 16-bit image of the tile, [pixel][channel]
    constexpr int CH = TN / 8;                              // 16-byte chunks per row
    static_assert(PITCH % 128 == 0 && TM <= 32 * (NTHREADS / 64), "one 32-pixel MFMA column block per wave (surplus waves idle)");
    __syncthreads();
    const int nrows = (p.flags & FTC_FLAG_BORDER_BIAS) ? 16 : 1;   // 16 border cases when a BatchNorm of the input is folded in
    // tap matrix [wrows][TN] (same swizzle as the image), then the bias table.  Only the Tw rows that produce stored outputs are staged:
    // the MFMA below reads 32 rows, the surplus ones come out of the bias table's bytes and only feed output rows nobody stores.
    const int wrows = (p.Tw + 3) & ~3;
    unsigned char* lwt = smem + TM * PITCH;
    float* lbias = reinterpret_cast<float*>(lwt + wrows * PITCH);
    for (int c = threadIdx.x; c < nrows * (TN / 4); c += NTHREADS)
        *reinterpret_cast<f32x4*>(lbias + 4 * c) = *reinterpret_cast<const f32x4*>(p.bias + 4 * c);
    for (int c = threadIdx.x; c < wrows * CH; c += NTHREADS) {
        const int r = c / CH, cc = c - r * CH;
        *reinterpret_cast<u32x4*>(lwt + r * PITCH + ((cc ^ (r & 7)) * 16)) =
            *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.w2) + ((size_t)r * TN + cc * 8) * 2);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int prow = pw0 + j * 32 + lpix;           // pixel slot of this lane's accumulator column (see halo_pixel_slot)
        unsigned char* lrow = smem + prow * PITCH;
        const float* brow = lbias;
        if (p.flags & FTC_FLAG_BORDER_BIAS) {
            const int m = row_to_m(prow);
            if (m >= 0) {
                const int rem = m % (p.Ho * p.Wo);
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                brow += ((oy == 0 ? 1 : 0) | (oy == p.Ho - 1 ? 2 : 0) | (ox == 0 ? 4 : 0) | (ox == p.Wo - 1 ? 8 : 0)) * TN;
            }
        }
#pragma unroll
        for (int i = 0; i < SN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = nw0 + i * 32 + 8 * q + 4 * half;
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(brow + nl);
                v = apply_act4<true>(v, p.act);
                const int chunk = (nl / 8) ^ (prow & 7);
                store4<WT>(reinterpret_cast<WT*>(lrow + chunk * 16) + (nl % 8), v);
            }
        }
    }
    __syncthreads();
    // wave w: pixels 32w .. 32w+31 of the tile;  A = tap matrix rows, B = image rows, K = TN in steps of 16
    if (wave * 32 >= TM) return;
    const int prow = wave * 32 + l31;
    f32x16 t;
#pragma unroll
    for (int e = 0; e < 16; ++e) t[e] = 0.0f;
    const unsigned char* arow = lwt + l31 * PITCH;
    const unsigned char* brow = smem + prow * PITCH;
#pragma unroll
    for (int g = 0; g < TN / 16; ++g) {
        const int c = g * 2 + half;
        using FragW = typename Frag<WT>::type;
        const FragW a = *reinterpret_cast<const FragW*>(arow + ((c ^ (l31 & 7)) * 16));
        const FragW b = *reinterpret_cast<const FragW*>(brow + ((c ^ (prow & 7)) * 16));
        t = mfma16(a, b, t);
    }
    const int m = row_to_m(prow);
    if (m >= 0) {
        float* dst = reinterpret_cast<float*>(p.out) + (size_t)m * p.Tw;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nq = 8 * q + 4 * half;                // rows (r&3) + 8*(r>>2) + 4*half of the C layout
            if (nq < p.Tw) {
                const f32x4 v = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
                *reinterpret_cast<f32x4*>(dst + nq) = v;
            }
        }
    }
}

// FTC_FLAG_TOP_FUSE for the fp32-tensor plans (fp32 and fp16x3; round 5).  The same decomposition as above -- T[p][tap*Co+o] = sum_c y[p][c]
// Wtop[o][tap][c], FTC_OP_TAPSUM does the 9-point sum -- but the per-pixel map is done in fp32 FMA straight from the accumulators: a lane
// holds 48 of its pixels' 192 activated channels (the 32x32 C layout: 4 consecutive channels per register quad), multiplies them by the tap
// matrix rows in LDS (wave-wide broadcast reads: only `half` differs between the lanes of a read) and the four partial sums of a pixel
// (2 channel halves of the tile x 2 lane halves) meet in LDS in a fixed order.  Per tile 2 x 960 FMA per lane -- ~1.5 % of the tile's K loop
// in the three-MFMA plan -- and the 192-channel fp32 tensor of the eight map heads (1.8 GB at batch 8) is neither written nor read back by
// thin_conv3x3_kernel.  `w2` = fp32 [32][TN] per group (rows >= Tw unused), T fp32 [pixels][Tw].
template <int SN, int SM, int NTHREADS, int TN, int TM, typename RowFn>
__device__ __forceinline__ void conv_epilogue_topfuse_f32(const ConvP& p, f32x16 (&acc)[SN][SM], unsigned char* smem, int nw0, int pw0,
                                                          int half, int lpix, RowFn row_to_m) {
    constexpr int TWMAX = 20;                               // floats per pixel of T (9 taps x 2 outputs, padded)
    __syncthreads();                                        // every wave is done with the operand buffers
    const int nrows = (p.flags & FTC_FLAG_BORDER_BIAS) ? 16 : 1;
    float* lw = reinterpret_cast<float*>(smem);                             // [TWMAX][TN] tap matrix
    float* lbias = lw + TWMAX * TN;                                         // [nrows][TN]
    float* part = lbias + 16 * TN;                                          // [4][TM][TWMAX] partial sums
    const int Tw = p.Tw < TWMAX ? p.Tw : TWMAX;
    for (int c = threadIdx.x; c < TWMAX * (TN / 4); c += NTHREADS) {
        const int r = c / (TN / 4);
        *reinterpret_cast<f32x4*>(lw + 4 * c) = r < Tw ? *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.w2) + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c = threadIdx.x; c < nrows * (TN / 4); c += NTHREADS)
        *reinterpret_cast<f32x4*>(lbias + 4 * c) = *reinterpret_cast<const f32x4*>(p.bias + 4 * c);
    __syncthreads();
    const float* brow[SM];
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        brow[j] = lbias;
        if (p.flags & FTC_FLAG_BORDER_BIAS) {
            const int m = row_to_m(pw0 + j * 32 + lpix);
            if (m >= 0) {
                const int rem = m % (p.Ho * p.Wo);
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                brow[j] += ((oy == 0 ? 1 : 0) | (oy == p.Ho - 1 ? 2 : 0) | (ox == 0 ? 4 : 0) | (ox == p.Wo - 1 ? 8 : 0)) * TN;
            }
        }
    }
    // bias + activation in place (the accumulators become y), then one pass per output row of the tap matrix: 12 broadcast reads of 16 bytes,
    // 48 FMA per pixel, one partial sum per pixel stored -- a loop over o that is NOT unrolled (unrolled, 240 bodies x 2 pixels, the
    // register allocator gave up: 3.4 KB of scratch)
#pragma unroll
    for (int i = 0; i < SN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl = nw0 + i * 32 + 8 * q + 4 * half;
#pragma unroll
            for (int j = 0; j < SM; ++j) {
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(brow[j] + nl);
                v = apply_act4<false>(v, p.act);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = v[e];
            }
        }
    const int pidx = (nw0 ? 2 : 0) + half;                  // which of the pixel's four partial sums (nw0 = 0 | TN / 2)
    float* pdst = part + ((size_t)pidx * TM + pw0 + lpix) * TWMAX;
#pragma unroll 1
    for (int o = 0; o < TWMAX; ++o) {
        float sj[SM];
#pragma unroll
        for (int j = 0; j < SM; ++j) sj[j] = 0.f;
        if (o < Tw) {
#pragma unroll
            for (int i = 0; i < SN; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(lw + o * TN + nw0 + i * 32 + 8 * q + 4 * half);
#pragma unroll
                    for (int j = 0; j < SM; ++j)
                        sj[j] += (acc[i][j][4 * q] * w[0] + acc[i][j][4 * q + 1] * w[1]) + (acc[i][j][4 * q + 2] * w[2] + acc[i][j][4 * q + 3] * w[3]);
                }
        }
#pragma unroll
        for (int j = 0; j < SM; ++j) pdst[(size_t)j * 32 * TWMAX + o] = sj[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < TM * (TWMAX / 4); c += NTHREADS) {
        const int prow = c / (TWMAX / 4), o4 = (c - prow * (TWMAX / 4)) * 4;
        const int m = row_to_m(prow);
        if (m < 0 || o4 >= p.Tw) continue;
        const float* s0 = part + (size_t)prow * TWMAX + o4;
        const f32x4 v = (*reinterpret_cast<const f32x4*>(s0) + *reinterpret_cast<const f32x4*>(s0 + TM * TWMAX)) +
                        (*reinterpret_cast<const f32x4*>(s0 + 2 * TM * TWMAX) + *reinterpret_cast<const f32x4*>(s0 + 3 * TM * TWMAX));
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.Tw + o4) = v;
    }
}

// Epilogue of the intra-workgroup split-K variant: every K group parks its raw fp32 partial tile in its own
// LDS image, then all 256*KG threads sum the KG images chunk by chunk (group 0 first: fixed order) and
// apply bias / activation / residual / bf16 copy on the way out.
template <typename WT, typename OutT, int SN, int SM, int KG, int TN, int TM, typename RowFn>
__device__ __forceinline__ void conv_epilogue_splitk(const ConvP& p, f32x16 (&acc)[SN][SM], unsigned char* smem, int kg, int n0, int nw0,
                                                     int pw0, int half, int l31, RowFn row_to_m) {
    constexpr int PITCH = epi_pitch<float>(TN);
    constexpr int CH = TN / 4;
    __syncthreads();
    unsigned char* img = smem + (size_t)kg * TM * PITCH;
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int prow = pw0 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < SN; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = nw0 + i * 32 + 8 * q + 4 * half;
                const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *reinterpret_cast<f32x4*>(img + prow * PITCH + (((nl >> 2) ^ (prow & 7)) << 4)) = v;
            }
    }
    __syncthreads();
    OutT* __restrict__ outp = reinterpret_cast<OutT*>(p.out);
    const bool has_res = (p.flags & FTC_FLAG_RESIDUAL) != 0;
    const bool vec_ok = ((p.Cout | p.CoutT | p.cout_off) & 3) == 0;
    for (int c = threadIdx.x; c < TM * CH; c += 256 * KG) {
        const int prow = c / CH, cc = c - prow * CH;
        const int m = row_to_m(prow);
        const int n = n0 + cc * 4;
        if (m < 0 || n >= p.Cout) continue;
        const int loff = prow * PITCH + ((cc ^ (prow & 7)) << 4);
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + loff);
#pragma unroll
        for (int g = 1; g < KG; ++g) v += *reinterpret_cast<const f32x4*>(smem + (size_t)g * TM * PITCH + loff);
        const float* brow = p.bias;
        if (p.flags & FTC_FLAG_BORDER_BIAS) {
            const int rem = m % (p.Ho * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            brow += ((oy == 0 ? 1 : 0) | (oy == p.Ho - 1 ? 2 : 0) | (ox == 0 ? 4 : 0) | (ox == p.Wo - 1 ? 8 : 0)) * p.Cout;
        }
        if (vec_ok) {
            v += *reinterpret_cast<const f32x4*>(brow + n);
            v = apply_act4<sizeof(WT) == 2>(v, p.act);
            if (has_res) {
                if (p.res_dtype == FTC_F32) v += load4<float>(reinterpret_cast<const float*>(p.res) + (size_t)m * p.Cout + n);
                else v += load4<typename Half16<WT>::type>(reinterpret_cast<const typename Half16<WT>::type*>(p.res) + (size_t)m * p.Cout + n);
            }
            store4<OutT>(outp + (size_t)m * p.CoutT + p.cout_off + n, v);
            if constexpr (sizeof(OutT) == 4) {
                if (p.out2) store_out2<WT>(p, m, n, v);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= p.Cout) continue;
                float x = apply_act_sel<sizeof(WT) == 2>(v[e] + brow[n + e], p.act);
                if (has_res) {
                    if (p.res_dtype == FTC_F32) x += reinterpret_cast<const float*>(p.res)[(size_t)m * p.Cout + n + e];
                    else x += (float)reinterpret_cast<const typename Half16<WT>::type*>(p.res)[(size_t)m * p.Cout + n + e];
                }
                outp[(size_t)m * p.CoutT + p.cout_off + n + e] = from_f32<OutT>(x);
            }
        }
    }
}

// KG > 1 = intra-workgroup split-K: KG groups of 4 waves each stage and multiply their own 1/KG of the K
// range of the SAME output tile (own LDS buffers), and the partial tiles are summed in fixed group order
// during the coalesced copy-out.  It multiplies the loads in flight per CU for the long-K, small-M MBConv
// project convs (M = 4608 at batch 8 gives only 2 workgroups of 64x64 per CU) without any inter-workgroup
// protocol and stays deterministic.
template <typename WT, typename InT, typename OutT, int BK, int WN, int WM, int SN, int SM, int NBUF, bool SE, int KG = 1>
__global__ __launch_bounds__(256 * KG) void conv_igemm_kernel(const ConvP p_launch) {
    ConvP p = p_launch;
    constexpr int E = 16 / (int)sizeof(WT);      // elements per 16-byte chunk (4 fp32 | 8 bf16)
    constexpr int CPR = BK / E;                  // chunks per LDS row
    constexpr int ROW = BK + E;                  // padded LDS row, elements
    constexpr int TN = WN * SN * 32;             // output channels per workgroup
    constexpr int TM = WM * SM * 32;             // output pixels per workgroup
    constexpr int NA = (TN * CPR + 255) / 256;
    constexpr int NB = (TM * CPR + 255) / 256;
    constexpr int RPP = 256 / CPR;               // rows covered per staging pass
    constexpr int BUF = (TN + TM) * ROW;         // elements per LDS buffer
    static_assert(WN * WM == 4, "4 waves per workgroup");
    static_assert(NBUF == 1 || NBUF == 2, "");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);     // K group of this wave
    WT* lds = reinterpret_cast<WT*>(smem_raw) + kg * NBUF * BUF;

    const int t = threadIdx.x & 255;
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
    bid = enter_group(p, bid);
    const int mt = bid / p.nN, nt = bid - mt * p.nN;
    const int m0 = mt * TM, n0 = nt * TN;

    const __amdgpu_buffer_rsrc_t rw = weight_rsrc(p, m0);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.se), 0, p.se_bytes, 0x00020000);
    constexpr bool use_se = SE;
    const int kc = t % CPR;                      // this thread's 16-byte chunk inside a K row
    const int row0 = t / CPR;
    const int HoWo = p.Ho * p.Wo;
    const int KK = p.KS * p.KS;

    // per-row byte offsets (one VGPR each) and the 9-bit "tap lands inside the image" masks
    int a_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = row0 + i * RPP;
        const int n = n0 + row;
        a_off[i] = (row < TN && n < p.Cout) ? (n * KK * p.Cin + kc * E) * (int)sizeof(WT) : OOB;
    }
    int b_off[NB], b_mask[NB], b_se[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = row0 + i * RPP;
        const int m = m0 + row;
        const bool ok = (row < TM) && (m < p.M);
        const int mm = ok ? m : 0;
        const int img = mm / HoWo;
        const int rem = mm - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        b_off[i] = (((img * p.H + iy0) * p.W + ix0) * p.CinT + p.cin_off + kc * E) * (int)sizeof(InT);
        int mask = 0;
        for (int r = 0; r < p.KS; ++r)
            for (int s = 0; s < p.KS; ++s)
                if (ok && (unsigned)(iy0 + r) < (unsigned)p.H && (unsigned)(ix0 + s) < (unsigned)p.W) mask |= 1 << (r * p.KS + s);
        b_mask[i] = mask;
        b_se[i] = (img * p.Cin + kc * E) * 4;
    }

    u32x4 ra[NA], rb[NB];
    // K position of the NEXT tile to fetch (all wave-uniform -> SALU); K group kg starts at step kg*nk/KG
    const int nk_g = p.nk / KG;
    int ld_tap = (kg * nk_g) / p.ncb, ld_cb = (kg * nk_g) % p.ncb;
    int ld_r = ld_tap / p.KS, ld_s = ld_tap % p.KS;

    auto gload = [&]() {
        const int c0 = ld_cb * BK;
        const int w_soff = (ld_tap * p.Cin + c0) * (int)sizeof(WT);
        const int in_toff = ((ld_r * p.W + ld_s) * p.CinT + c0) * (int)sizeof(InT);
        // partial last channel block: only possible when Cin % BK != 0 (never for BK >= 64, see select_bk)
        const bool cok = BK >= 64 ? true : (c0 + kc * E) < p.Cin;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload(rw, cok ? a_off[i] : OOB, w_soff);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const bool ok = cok && ((b_mask[i] >> ld_tap) & 1);
            rb[i] = load_act<WT, InT>(rin, ok ? b_off[i] + in_toff : OOB, use_se, rse, b_se[i] + c0 * 4, (p.flags & FTC_FLAG_PRESPLIT) != 0);
        }
        if (++ld_cb == p.ncb) {
            ld_cb = 0;
            ++ld_tap;
            if (++ld_s == p.KS) { ld_s = 0; ++ld_r; }
        }
    };

    WT* const wA = lds + row0 * ROW + kc * E;                     // staging write base (weights)
    WT* const wB = lds + (TN + row0) * ROW + kc * E;              //                     (pixels)
    auto lds_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if ((TN * CPR) % 256 == 0 || row0 + i * RPP < TN) *reinterpret_cast<u32x4*>(wA + buf * BUF + i * RPP * ROW) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if ((TM * CPR) % 256 == 0 || row0 + i * RPP < TM) *reinterpret_cast<u32x4*>(wB + buf * BUF + i * RPP * ROW) = rb[i];
    };

    f32x16 acc[SN][SM];
#pragma unroll
    for (int i = 0; i < SN; ++i)
#pragma unroll
        for (int j = 0; j < SM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // One 32x32 tile per wave means every MFMA of a K step depends on the previous one (the s_memtime timeline of the
    // 64x64 tile showed ~775 cycles of "compute" per step for 4 MFMAs); odd K groups go to a second accumulator that is
    // added back once, after the K loop -- two independent chains (bf16 only; changes fp32 association, not determinism).
    constexpr bool DUAL = sizeof(WT) == 2 && SN * SM == 1;
    f32x16 accB[DUAL ? SN : 1][DUAL ? SM : 1];
    if constexpr (DUAL) {
#pragma unroll
        for (int e = 0; e < 16; ++e) accB[0][0][e] = 0.0f;
    }

    using FragT = typename Frag<WT>::type;
    const WT* const fA = lds + (wn * SN * 32 + l31) * ROW + half * E;
    const WT* const fB = lds + (TN + wm * SM * 32 + l31) * ROW + half * E;
    auto compute = [&](int buf) {
        const WT* A = fA + buf * BUF;
        const WT* Bm = fB + buf * BUF;
        if constexpr (sizeof(WT) == 4) {
            if constexpr (is_x3<WT>) {
                static_assert((BK / 8) % 2 == 0, "fp16x3 pairs the K groups of the fp32 kernel");
#pragma unroll
                for (int g = 0; g < BK / 8; g += 2) {
                    f16x8 ah[SN], al[SN];
#pragma unroll
                    for (int i = 0; i < SN; ++i)
                        frag_hl(*reinterpret_cast<const FragT*>(A + i * 32 * ROW + g * 8), *reinterpret_cast<const FragT*>(A + i * 32 * ROW + g * 8 + 8), ah[i], al[i]);
#pragma unroll
                    for (int j = 0; j < SM; ++j) {
                        f16x8 bh, bl;
                        frag_hl(*reinterpret_cast<const FragT*>(Bm + j * 32 * ROW + g * 8), *reinterpret_cast<const FragT*>(Bm + j * 32 * ROW + g * 8 + 8), bh, bl);
#pragma unroll
                        for (int i = 0; i < SN; ++i) acc[i][j] = mfma_split(ah[i], al[i], bh, bl, acc[i][j]);
                    }
                }
                return;
            }
            // 8 k per group: lanes 0-31 hold k = 0..3, lanes 32-63 hold k = 4..7 of the group;
            // step tt feeds A[:,k=tt | 4+tt], B likewise -- same permutation on both operands.
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                FragT af[SN], bf[SM];
#pragma unroll
                for (int i = 0; i < SN; ++i) af[i] = *reinterpret_cast<const FragT*>(A + i * 32 * ROW + g * 8);
#pragma unroll
                for (int j = 0; j < SM; ++j) bf[j] = *reinterpret_cast<const FragT*>(Bm + j * 32 * ROW + g * 8);
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
                for (int i = 0; i < SN; ++i) af[i] = *reinterpret_cast<const FragT*>(A + i * 32 * ROW + g * 16);
#pragma unroll
                for (int j = 0; j < SM; ++j) bf[j] = *reinterpret_cast<const FragT*>(Bm + j * 32 * ROW + g * 16);
#pragma unroll
                for (int i = 0; i < SN; ++i)
#pragma unroll
                    for (int j = 0; j < SM; ++j) {
                        if (DUAL && (g & 1)) accB[i][j] = mfma16(af[i], bf[j], accB[i][j]);
                        else acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
                    }
            }
        }
    };

    gload();
    if constexpr (NBUF == 2) {
        lds_write(0);
        __syncthreads();
        for (int it = 0; it < nk_g; it += 2) {
            if (it + 1 < nk_g) gload();
            compute(0);
            if (it + 1 < nk_g) lds_write(1);
            __syncthreads();
            if (it + 1 >= nk_g) break;
            if (it + 2 < nk_g) gload();
            compute(1);
            if (it + 2 < nk_g) lds_write(0);
            __syncthreads();
        }
    } else {
        for (int it = 0; it < nk_g; ++it) {
            lds_write(0);
            __syncthreads();
            if (it + 1 < nk_g) gload();          // in flight while this tile is multiplied
            compute(0);
            __syncthreads();
        }
    }
    if constexpr (DUAL) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] += accB[0][0][e];
    }
    if constexpr (KG > 1) {
        conv_epilogue_splitk<WT, OutT, SN, SM, KG, TN, TM>(p, acc, smem_raw, kg, n0, wn * SN * 32, wm * SM * 32, half, l31,
                                                           [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; });
        return;
    }

    if (epi_lds_ok<OutT>(p)) {
        conv_epilogue_lds<WT, OutT, SN, SM, 256, TN, TM>(p, acc, smem_raw, n0, wn * SN * 32, wm * SM * 32, half, l31,
                                                         [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; });
    } else {
        conv_epilogue<WT, OutT, SN, SM>(p, acc, m0, n0, wn, wm, half, l31);
    }
}

// ------------------------------------------------------------------------------------------------
// Direct-to-LDS variant (buffer_load ... lds): used whenever the activation dtype equals the MFMA
// compute dtype and no SE scale is applied, i.e. for ~95 % of the FLOPs.  The staging pass of the
// register kernel (ds_write_b128 at ~79 B/clk/CU) made the LDS pipe as busy as the MFMA pipe; here
// the tiles go HBM/L2 -> LDS by DMA: no staging VGPRs, no ds_write, and NBUF-1 tiles stay in
// flight across the single barrier per K step (counted s_waitcnt vmcnt).
// LDS image: unpadded rows of CPR 16-byte chunks; a wave-level DMA writes 64 consecutive chunks,
// so the bank-conflict-avoiding XOR swizzle is applied on the SOURCE side (lane -> which global
// chunk it fetches) and again on the fragment read: slot = chunk ^ f(row), f(row) = (row>>1)&7 for
// 128-byte rows, (row>>2)&3 for 64-byte rows (conflict-free ds_read_b128).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;

// (kept in a __device__ function: called straight from a lambda the builtin makes the host pass drop the kernel stub)
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, lds_void_t* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void wg_barrier() { __builtin_amdgcn_s_barrier(); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// (Round 4 measured a GATE variant -- the SE gate multiplied into the weight fragments as they leave LDS instead of a per-image folded
// weight copy: 49.0 vs 33.3 us on the stage-6 project GEMM, bf16 has no packed multiply on gfx950 -- and round 5 removed it: DESIGN.md appendix.)
template <typename WT, typename OutT, int BK, int WN, int WM, int SN, int SM, int NBUF>
__global__ __launch_bounds__(256) void conv_igemm_glds_kernel(const ConvP p_launch) {
    ConvP p = p_launch;
    constexpr int E = 16 / (int)sizeof(WT);
    constexpr int CPR = BK / E;                  // 16-byte chunks per row: 8 (128-byte rows) or 4
    constexpr int ROWB = CPR * 16;
    constexpr int TN = WN * SN * 32, TM = WM * SM * 32;
    constexpr int NCH = (TN + TM) * CPR;         // chunks per tile
    constexpr int NL = NCH / 256;                // DMA instructions per thread per tile
    constexpr int ACH = TN * CPR;                // chunks of the weight part
    constexpr int BUFB = NCH * 16;               // bytes per LDS buffer
    constexpr int D = NBUF - 1;                  // tiles in flight
    static_assert(CPR == 8 || CPR == 4, "");
    static_assert(NCH % 256 == 0 && ACH % 64 == 0, "tile must be a whole number of wave-level DMAs");
    static_assert(WN * WM == 4 && (NBUF == 2 || NBUF == 3), "");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int half = lane >> 5, l31 = lane & 31;

    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    bid = enter_group(p, bid);
    const int mt = bid / p.nN, nt = bid - mt * p.nN;
    const int m0 = mt * TM, n0 = nt * TN;

    const __amdgpu_buffer_rsrc_t rw = weight_rsrc(p, m0);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
    const int HoWo = p.Ho * p.Wo;
    const int KK = p.KS * p.KS;

    // DMA slot q = i*256 + t of a tile: LDS byte q*16; rows of the weight part first, pixels after.
    int s_off[NL], s_mask[NL];                   // byte offset (global), tap mask | (chunk index << 16)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int q = i * 256 + t;
        const bool isA = (ACH % 256 == 0) ? (i < ACH / 256) : (i * 256 + wave * 64 < ACH);
        const int qq = isA ? q : q - ACH;
        const int row = qq / CPR, slot = qq % CPR;
        const int kc = slot ^ (CPR == 8 ? (row >> 1) & 7 : (row >> 2) & 3);
        if (isA) {
            const int n = n0 + row;
            s_off[i] = n < p.Cout ? (n * KK * p.Cin + kc * E) * (int)sizeof(WT) : OOB;
            s_mask[i] = 0x1ff | (kc << 16);
        } else {
            const int m = m0 + row;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int img = mm / HoWo;
            const int rem = mm - img * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            s_off[i] = (((img * p.H + iy0) * p.W + ix0) * p.CinT + p.cin_off + kc * E) * (int)sizeof(WT);
            int mask = 0;
            for (int r = 0; r < p.KS; ++r)
                for (int s = 0; s < p.KS; ++s)
                    if (ok && (unsigned)(iy0 + r) < (unsigned)p.H && (unsigned)(ix0 + s) < (unsigned)p.W) mask |= 1 << (r * p.KS + s);
            s_mask[i] = mask | (kc << 16);
        }
    }

    int ld_tap = 0, ld_r = 0, ld_s = 0, ld_cb = 0;
    // `bufoff` = byte offset of the ring slot (wave-uniform, lives in an SGPR): one straight-line loop
    // body, so the accumulators stay in AGPRs (an unrolled-by-NBUF body made hipcc shuttle all 96 of
    // them through VGPRs every K step).
    auto issue = [&](int bufoff) {
        const int c0 = ld_cb * BK;
        const int w_soff = (ld_tap * p.Cin + c0) * (int)sizeof(WT);
        const int in_toff = ((ld_r * p.W + ld_s) * p.CinT + c0) * (int)sizeof(WT);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const bool isA = (ACH % 256 == 0) ? (i < ACH / 256) : (i * 256 + wave * 64 < ACH);
            const bool cok = BK == 64 ? true : (c0 + (s_mask[i] >> 16) * E) < p.Cin;
            lds_void_t* dst = (lds_void_t*)(smem_raw + bufoff + (i * 256 + wave * 64) * 16);
            if (isA) {
                glds16(rw, dst, cok ? s_off[i] : OOB, w_soff);
            } else {
                const bool ok = cok && ((s_mask[i] >> ld_tap) & 1);
                glds16(rin, dst, ok ? s_off[i] + in_toff : OOB, 0);
            }
        }
        if (++ld_cb == p.ncb) {
            ld_cb = 0;
            ++ld_tap;
            if (++ld_s == p.KS) { ld_s = 0; ++ld_r; }
        }
    };

    f32x16 acc[SN][SM];
#pragma unroll
    for (int i = 0; i < SN; ++i)
#pragma unroll
        for (int j = 0; j < SM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // One 32x32 tile per wave means every MFMA of a K step depends on the previous one (the s_memtime timeline of the
    // 64x64 tile showed ~775 cycles of "compute" per step for 4 MFMAs); odd K groups go to a second accumulator that is
    // added back once, after the K loop -- two independent chains (bf16 only; changes fp32 association, not determinism).
    constexpr bool DUAL = sizeof(WT) == 2 && SN * SM == 1;
    f32x16 accB[DUAL ? SN : 1][DUAL ? SM : 1];
    if constexpr (DUAL) {
#pragma unroll
        for (int e = 0; e < 16; ++e) accB[0][0][e] = 0.0f;
    }

    using FragT = typename Frag<WT>::type;
    constexpr int G = CPR / 2;                   // MFMA K groups per tile (two chunks each: lower / upper half-wave)
    const int fr = CPR == 8 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;
    int offA[G], offB[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int sl = ((g * 2 + half) ^ fr) * 16;
        offA[g] = (wn * SN * 32 + l31) * ROWB + sl;
        offB[g] = (TN + wm * SM * 32 + l31) * ROWB + sl;
    }
    const bool presplit = (p.flags & FTC_FLAG_PRESPLIT) != 0;
    auto compute = [&](int bufoff) {
        const unsigned char* base = smem_raw + bufoff;
        if constexpr (is_x3<WT>) {
            static_assert(!is_x3<WT> || G % 2 == 0, "fp16x3 pairs the K groups of the fp32 kernel");
#pragma unroll
            for (int g = 0; g < G; g += 2) {
                f16x8 ah[SN], al[SN];
#pragma unroll
                for (int i = 0; i < SN; ++i)
                    frag_hl(*reinterpret_cast<const f32x4*>(base + offA[g] + i * 32 * ROWB), *reinterpret_cast<const f32x4*>(base + offA[g + 1] + i * 32 * ROWB), ah[i], al[i]);
#pragma unroll
                for (int j = 0; j < SM; ++j) {
                    f16x8 bh, bl;
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(base + offB[g] + j * 32 * ROWB), b1 = *reinterpret_cast<const f32x4*>(base + offB[g + 1] + j * 32 * ROWB);
                    if (presplit) frag_hl(b0, b1, bh, bl);             // FTC_FLAG_PRESPLIT: register renaming instead of ~35 VALU instructions per fragment
                    else split16(b0, b1, bh, bl);
#pragma unroll
                    for (int i = 0; i < SN; ++i) acc[i][j] = mfma_split(ah[i], al[i], bh, bl, acc[i][j]);
                }
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            FragT af[SN], bf[SM];
#pragma unroll
            for (int i = 0; i < SN; ++i) af[i] = *reinterpret_cast<const FragT*>(base + offA[g] + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < SM; ++j) bf[j] = *reinterpret_cast<const FragT*>(base + offB[g] + j * 32 * ROWB);
            if constexpr (sizeof(WT) == 4) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < SN; ++i)
#pragma unroll
                        for (int j = 0; j < SM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][tt], bf[j][tt], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < SN; ++i)
#pragma unroll
                    for (int j = 0; j < SM; ++j) {
                        if (DUAL && (g & 1)) accB[i][j] = mfma16(af[i], bf[j], accB[i][j]);
                        else acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
                    }
            }
        }
    };

    // prologue: D tiles in flight
    int iss_off = 0, cur_off = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (j < p.nk) issue(iss_off);
        iss_off += BUFB;
    }
    for (int it = 0; it < p.nk; ++it) {
        // tile `it` has landed once at most the later-issued tiles remain outstanding
        if (D == 1 || it + 1 >= p.nk) wait_vmcnt<0>(); else wait_vmcnt<NL>();
        wg_barrier();
        if (it + D < p.nk) issue(iss_off);                   // that slot was consumed in step it-1
        iss_off = iss_off + BUFB == NBUF * BUFB ? 0 : iss_off + BUFB;
        compute(cur_off);
        cur_off = cur_off + BUFB == NBUF * BUFB ? 0 : cur_off + BUFB;
    }
    if constexpr (DUAL) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] += accB[0][0][e];
    }
    if (epi_lds_ok<OutT>(p)) {
        conv_epilogue_lds<WT, OutT, SN, SM, 256, TN, TM>(p, acc, smem_raw, n0, wn * SN * 32, wm * SM * 32, half, l31,
                                                         [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; });
    } else {
        conv_epilogue<WT, OutT, SN, SM>(p, acc, m0, n0, wn, wm, half, l31);
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with the activation HALO resident in LDS (8 waves, 2-D pixel tiles).
//
// The per-tap kernels above re-fetch every activation row 9 times and every weight row once per
// 128-pixel tile: 78 FLOP per byte moved L2 -> LDS, which (PMC + Little's law) caps
// the matrix pipe near 40 %.  Here a workgroup owns a 16x16-pixel output tile of one image:
//   * per 64-(or 32-)channel block the 18x18 halo of input pixels is DMA'd ONCE (double buffered) and
//     all nine taps read their shifted windows out of it;
//   * the weights of one (tap, channel block) form a K step and stream through a 3-slot DMA ring;
//   * 8 waves (2 channel halves x 4 pixel quarters) share both, so a 192-channel tile moves
//     ~14 KB per 192x128x64 MACs instead of 40 KB.
// LDS: 3 x TN x ROWB (weights) + 2 x 324 x ROWB (halo) = 153 KB for TN = 192: one workgroup per CU.
// Same swizzled, unpadded row image and counted-vmcnt protocol as conv_igemm_glds_kernel; the number of
// DMA instructions a wave issues per step depends on the wave (partial last pass), so the wait counts
// are per-wave values.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    default: wait_vmcnt<9>(); break;
    }
}

// Dynamic LDS of the halo kernel: 3 weight slots + 2 halo buffers; the half-height form also sizes for its epilogue images (the fused
// tap epilogue needs tile image + 16 bias rows + tap matrix, more than its small operand buffers).
template <typename WT, int CPR, int SN, bool TOPF, int WMQ>
__host__ __device__ constexpr size_t halo_lds_bytes() {
    constexpr int TN = 2 * SN * 32, TY = 4 * WMQ, NH = (TY + 2) * 18;
    constexpr size_t ops = (size_t)3 * TN * CPR * 16 + (size_t)2 * NH * CPR * 16;
    constexpr size_t topf = (size_t)TY * 16 * TN * 2 + (size_t)16 * TN * 4 + (size_t)32 * TN * 2;
    constexpr size_t epi = (size_t)TY * 16 * TN * 2 + (size_t)16 * TN * 4;          // LDS-staged bf16 / fp16 output image + bias rows
    return WMQ == 4 ? ops : (TOPF ? (ops > topf ? ops : topf) : (ops > epi ? ops : epi));
}

// (second launch bound = waves per SIMD: with 64-byte rows two workgroups fit the LDS of a CU, which needs <= 128 VGPRs)
// WMQ = pixel quarters (waves along the pixel axis): 4 = the 16x16-pixel tile of 8 waves described above.  WMQ = 2 -- a HALF-HEIGHT
// tile (8 x 16 pixels, 4 waves) whose 64-byte-row buffers + epilogue image take 72 KB so that TWO workgroups share a CU with 256 VGPRs
// each -- was built and measured in round 2 for the fused last-level kernels (parity-green on every halo / UPCAT_IN / TOP_FUSE case):
// 2.87 ms against 2.55 ms for the 16x16 form on the 8-head launch, 384 vs 346 us on the feature head.  The K step of 32 doubles the
// barriers per FLOP and every workgroup streams its own copy of the weight slices (2x the weight bytes L2 -> LDS per CU), which costs
// more than the inter-workgroup overlap returns.  The parameter stays; the variant is not instantiated.
template <typename WT, typename OutT, int CPR, int SN, bool TOPF = false, bool UPIN = false, int WMQ = 4>
__global__ __launch_bounds__(128 * WMQ, (WMQ == 2) ? 2 : (CPR == 4 && !UPIN && !TOPF) ? 4 : 2) void conv3x3_halo_kernel(const ConvP p_launch) {
    ConvP p = p_launch;
    constexpr int E = 16 / (int)sizeof(WT);
    constexpr int BK = CPR * E;
    constexpr int ROWB = CPR * 16;
    constexpr int WN = 2, WM = WMQ, SM = 2;
    constexpr int NT = 64 * WN * WM;                                         // threads per workgroup (512 | 256)
    constexpr int TN = WN * SN * 32;
    constexpr int TY = 4 * WM, TX = 16, HW = TX + 2, NH = (TY + 2) * HW;    // 324 (180) halo pixels
    constexpr int WCH = TN * CPR, HCH = NH * CPR;                            // 16-byte chunks per weight step / halo block
    constexpr int NLW = (WCH + NT - 1) / NT, NLH = (HCH + NT - 1) / NT;      // DMA passes of the workgroup's threads
    constexpr int WSLOT = TN * ROWB, HBUF = NH * ROWB;
    static_assert(CPR == 8 || CPR == 4, "");
    static_assert(NLW + NLH <= 9 && NLH <= 8, "");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const wbase = smem_raw;                   // 3 weight slots
    unsigned char* const hbase = smem_raw + 3 * WSLOT;       // 2 halo buffers

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int half = lane >> 5, l31 = lane & 31;

    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    bid = enter_group(p, bid);
    const int nt = bid % p.nN;
    int sp = bid / p.nN;
    const int tilesX = (p.Wo + TX - 1) / TX, tilesY = (p.Ho + TY - 1) / TY;
    const int img = sp / (tilesX * tilesY);
    sp -= img * tilesX * tilesY;
    const int ty0 = (sp / tilesX) * TY, tx0 = (sp % tilesX) * TX;
    const int n0 = nt * TN;

    const __amdgpu_buffer_rsrc_t rw = weight_rsrc(p, img * p.Ho * p.Wo);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);

    // per-lane global byte offsets of the DMA slots (fixed for the whole tile)
    int w_off[NLW], h_off[NLH];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int q = i * NT + t;
        const int row = q / CPR, kc = (q % CPR) ^ (CPR == 8 ? (row >> 1) & 7 : (row >> 2) & 3);
        const int n = n0 + row;
        w_off[i] = (q < WCH && n < p.Cout) ? (n * 9 * p.Cin + kc * E) * (int)sizeof(WT) : OOB;
    }
#pragma unroll
    for (int i = 0; i < NLH; ++i) {
        const int q = i * NT + t;
        const int hr = q / CPR, kc = (q % CPR) ^ (CPR == 8 ? (hr >> 1) & 7 : (hr >> 2) & 3);
        const int iy = ty0 - 1 + hr / HW, ix = tx0 - 1 + hr % HW;
        const bool ok = q < HCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        if constexpr (UPIN) h_off[i] = ok ? (((img * p.H + iy) * p.W + ix) * (p.Cin - p.Cy) + kc * E) * (int)sizeof(WT) : OOB;
        else h_off[i] = ok ? (((img * p.H + iy) * p.W + ix) * p.CinT + p.cin_off + kc * E) * (int)sizeof(WT) : OOB;
    }
    // UPIN: the channel blocks are walked tap-source first (plain DMA, so the prologue needs no arithmetic), then the
    // upsampled ones, each produced by the VALU during the nine taps of the block before it.
    const int ncb_up = UPIN ? p.Cy / BK : 0;
    const int nb_dma = p.ncb - ncb_up;
    const __amdgpu_buffer_rsrc_t rin2 = UPIN ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in2u), 0, p.in2u_bytes, 0x00020000) : rin;
    auto phys_cb = [&](int j) { return UPIN ? (j < nb_dma ? ncb_up + j : j - nb_dma) : j; };
    // DMA instructions THIS wave issues per weight step / halo block (the last pass may cover fewer waves)
    int nlw = 0, nlh = 0;
#pragma unroll
    for (int i = 0; i < NLW; ++i) nlw += (i * NT + wave * 64 < WCH) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < NLH; ++i) nlh += (i * NT + wave * 64 < HCH) ? 1 : 0;

    auto issue_w = [&](int k) {                              // weights of K step k -> ring slot k % 3
        const int cb = k / 9, tap = k - cb * 9;
        const int soff = (tap * p.Cin + phys_cb(cb) * BK) * (int)sizeof(WT);
        unsigned char* slot = wbase + (k % 3) * WSLOT;
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            if (i * NT + wave * 64 < WCH) {                 // wave-uniform
                lds_void_t* dst = (lds_void_t*)(slot + (i * NT + wave * 64) * 16);
                if (i * NT + t < WCH) glds16(rw, dst, w_off[i], soff);
            }
        }
    };
    auto issue_h = [&](int cb) {                             // halo of channel block cb -> buffer cb & 1
        const int soff = cb * BK * (int)sizeof(WT);
        unsigned char* buf = hbase + (cb & 1) * HBUF;
#pragma unroll
        for (int i = 0; i < NLH; ++i) {
            if (i * NT + wave * 64 < HCH) {
                lds_void_t* dst = (lds_void_t*)(buf + (i * NT + wave * 64) * 16);
                if (i * NT + t < HCH) glds16(UPIN ? rin2 : rin, dst, h_off[i], soff);
            }
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
    constexpr int G = CPR / 2;
    const int frA = CPR == 8 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;
    int offA[G];
#pragma unroll
    for (int g = 0; g < G; ++g) offA[g] = (wn * SN * 32 + l31) * ROWB + (((g * 2 + half) ^ frA) << 4);
    // Which of its sub-tile's 32 pixels (2 tile rows x 16) a lane owns.  ds_read_b128 services a wave in four groups of 16 lanes --
    // {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32 (MI355X_MICROARCH.md, LDS) -- and the swizzled halo image is
    // conflict-free for 16 CONSECUTIVE halo pixels (slot = halo row mod 16).  With the natural mapping (lane = pixel) a group mixes
    // pixels of two tile rows whose halo rows are 18 apart, two of them collide mod 16 and every B-fragment read takes 8 LDS cycles
    // instead of 4 (measured: SQ_LDS_BANK_CONFLICT = 29 % of the LDS cycles of this kernel, profiles/r02a).  So each lane group gets
    // ONE tile row: group parity = popcount(lane >> 2) & 1, rank inside the group = (lane >> 3) * 4 + (lane & 3).
    const int lpix = ((__builtin_popcount(l31 >> 2) & 1) << 4) | ((l31 >> 3) << 2) | (l31 & 3);
    int hr0[SM];                                             // halo row of this lane's pixel (tap 0,0) per sub-tile
#pragma unroll
    for (int j = 0; j < SM; ++j) hr0[j] = (wm * 4 + j * 2 + (lpix >> 4)) * HW + (lpix & 15);

    auto compute = [&](int k) {
        const int cb = k / 9, tap = k - cb * 9;
        const int d = (tap / 3) * HW + (tap % 3);
        const unsigned char* wa = wbase + (k % 3) * WSLOT;
        const unsigned char* hb = hbase + (cb & 1) * HBUF;
        int rowB[SM], f4[SM];
#pragma unroll
        for (int j = 0; j < SM; ++j) {
            const int hr = hr0[j] + d;
            rowB[j] = hr * ROWB;
            f4[j] = (CPR == 8 ? (hr >> 1) & 7 : (hr >> 2) & 3) << 4;
        }
        // fragments of K group g+1 are requested from LDS before the MFMAs of group g are issued
        FragT af[2][SN], bf[2][SM];
        auto ldfrag = [&](int g, int s) {
#pragma unroll
            for (int i = 0; i < SN; ++i) af[s][i] = *reinterpret_cast<const FragT*>(wa + offA[g] + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < SM; ++j) bf[s][j] = *reinterpret_cast<const FragT*>(hb + rowB[j] + ((((g * 2 + half) << 4)) ^ f4[j]));
        };
        if constexpr (is_x3<WT>) {
            static_assert(!is_x3<WT> || G % 2 == 0, "fp16x3 pairs the K groups of the fp32 kernel");
#pragma unroll
            for (int g = 0; g < G; g += 2) {
                f16x8 ah[SN], al[SN];
#pragma unroll
                for (int i = 0; i < SN; ++i)
                    frag_hl(*reinterpret_cast<const FragT*>(wa + offA[g] + i * 32 * ROWB), *reinterpret_cast<const FragT*>(wa + offA[g + 1] + i * 32 * ROWB), ah[i], al[i]);
#pragma unroll
                for (int j = 0; j < SM; ++j) {
                    f16x8 bh, bl;                                    // the halo image was split in place (halo_split below)
                    frag_hl(*reinterpret_cast<const FragT*>(hb + rowB[j] + ((((g * 2 + half) << 4)) ^ f4[j])),
                            *reinterpret_cast<const FragT*>(hb + rowB[j] + (((((g + 1) * 2 + half) << 4)) ^ f4[j])), bh, bl);
#pragma unroll
                    for (int i = 0; i < SN; ++i) acc[i][j] = mfma_split(ah[i], al[i], bh, bl, acc[i][j]);
                }
            }
            return;
        }
        ldfrag(0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int s = g & 1;
            if (g + 1 < G) ldfrag(g + 1, s ^ 1);
            if constexpr (sizeof(WT) == 4) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < SN; ++i)
#pragma unroll
                        for (int j = 0; j < SM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][i][tt], bf[s][j][tt], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < SN; ++i)
#pragma unroll
                    for (int j = 0; j < SM; ++j)
                        acc[i][j] = mfma16(af[s][i], bf[s][j], acc[i][j]);
            }
        }
    };

    // ---- UPIN: one pass = one 16-byte chunk (E channels of one halo pixel) per thread: four buffer loads at the bilinear
    // corners (issued in one K step), interpolated and written to the halo image in the next (ATen upsample_bilinear2d,
    // align_corners=True, same expression as upcat_kernel; halo pixels outside the image are the conv's zero padding).
    // The corner offsets and weights of a pass do not depend on the channel block: computed once per tile
    // (halo pixels outside the image get out-of-range offsets -> zeros -> the conv's zero padding).
    constexpr int NUP = UPIN ? NLH : 1;
    int up_off[NUP][4];
    float up_ly[NUP], up_lx[NUP];
    if constexpr (UPIN) {
#pragma unroll
        for (int i = 0; i < NLH; ++i) {
            const int q = i * NT + t;
            const int hr = q / CPR, kc = (q % CPR) ^ (CPR == 8 ? (hr >> 1) & 7 : (hr >> 2) & 3);
            const int iy = ty0 - 1 + hr / HW, ix = tx0 - 1 + hr % HW;
            const bool ok = q < HCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float sy = p.ry * (float)iy, sx = p.rx * (float)ix;
            const int y0 = ok ? (int)sy : 0, x0 = ok ? (int)sx : 0;
            const int y1 = y0 + (y0 < p.Hi - 1 ? 1 : 0), x1 = x0 + (x0 < p.Wi - 1 ? 1 : 0);
            up_ly[i] = sy - (float)y0;
            up_lx[i] = sx - (float)x0;
            const int r0 = (img * p.Hi + y0) * p.Wi, r1 = (img * p.Hi + y1) * p.Wi;
            const int pb = p.Cy * (int)sizeof(WT), cbase = kc * E * (int)sizeof(WT);
            up_off[i][0] = ok ? (r0 + x0) * pb + cbase : OOB;
            up_off[i][1] = ok ? (r0 + x1) * pb + cbase : OOB;
            up_off[i][2] = ok ? (r1 + x0) * pb + cbase : OOB;
            up_off[i][3] = ok ? (r1 + x1) * pb + cbase : OOB;
        }
    }
    u32x4 st[4] = {};
    auto up_load = [&](int pass, int ub) {
        const int soff = ub * BK * (int)sizeof(WT);
#pragma unroll
        for (int i = 0; i < NUP; ++i)
            if (i == pass) {
#pragma unroll
                for (int c = 0; c < 4; ++c) st[c] = bload(rin, up_off[i][c], soff);
            }
    };
    auto up_store = [&](int pass, int bufidx) {
        const int q = pass * NT + t;
        if (q >= HCH) return;
        float ly1 = 0.f, lx1 = 0.f;
#pragma unroll
        for (int i = 0; i < NUP; ++i)
            if (i == pass) { ly1 = up_ly[i]; lx1 = up_lx[i]; }
        const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
        if constexpr (sizeof(WT) == 2) {
            float v[8];
            const FragT c0 = __builtin_bit_cast(FragT, st[0]), c1 = __builtin_bit_cast(FragT, st[1]);
            const FragT c2 = __builtin_bit_cast(FragT, st[2]), c3 = __builtin_bit_cast(FragT, st[3]);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = ly0 * (lx0 * (float)c0[e] + lx1 * (float)c1[e]) + ly1 * (lx0 * (float)c2[e] + lx1 * (float)c3[e]);
            store16<WT>(reinterpret_cast<WT*>(hbase + bufidx * HBUF + q * 16), v);
        } else {
            // fp32 tensors (the fp32 and fp16x3 plans, round 3): four channels per chunk; the fp16x3 halo image holds pre-split chunks,
            // so an upsampled chunk is written split right away (halo_split only walks the DMA-sourced blocks)
            const f32x4 c0 = __builtin_bit_cast(f32x4, st[0]), c1 = __builtin_bit_cast(f32x4, st[1]);
            const f32x4 c2 = __builtin_bit_cast(f32x4, st[2]), c3 = __builtin_bit_cast(f32x4, st[3]);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ly0 * (lx0 * c0[e] + lx1 * c1[e]) + ly1 * (lx0 * c2[e] + lx1 * c3[e]);
            if constexpr (is_x3<WT>) *reinterpret_cast<u32x4*>(hbase + bufidx * HBUF + q * 16) = chunk_hl(v);
            else *reinterpret_cast<f32x4*>(hbase + bufidx * HBUF + q * 16) = v;
        }
    };

    const int nk = 9 * p.ncb;
    // 0x1000: timeline of wave 0 of the first 512 workgroups into p.res (tools/conv_bench.py --timeline)
    const bool tl_on = (p.flags & 0x1000) && blockIdx.x < 512 && t == 0;
    unsigned long long* tl = reinterpret_cast<unsigned long long*>(const_cast<void*>(p.res)) + (size_t)blockIdx.x * 64;
    if (tl_on) tl[0] = __builtin_amdgcn_s_memtime();
    // fp16x3: the fp32 halo image is split IN PLACE into [hi x4 | lo x4] chunks, once per channel block, by the thread that DMA'd the
    // chunk (its own vmcnt wait is all the ordering that needs; the next workgroup barrier publishes the result) -- instead of once per
    // fragment read, which repeated the conversion for every tap and every channel-block wave (18x).  Pass i of block nxt runs in tap
    // 2 + i of the block before it: the DMA issued in tap 0 is complete by the wait at the top of tap 2.
    static_assert(!is_x3<WT> || NLH <= 7, "one in-place split pass per tap 2..8");
    auto halo_split = [&](int pass, int bufidx) {
#pragma unroll
        for (int i = 0; i < NLH; ++i)
            if (i == pass && i * NT + t < HCH) {
                f32x4* q = reinterpret_cast<f32x4*>(hbase + bufidx * HBUF + (i * NT + t) * 16);
                *reinterpret_cast<u32x4*>(q) = chunk_hl(*q);
            }
    };
    issue_h(0);
    issue_w(0);
    if (nk > 1) issue_w(1);
    if constexpr (is_x3<WT>) {
        wait_vmcnt<0>();                                                 // (prologue only) own halo chunks of block 0 have landed
#pragma unroll
        for (int i = 0; i < NLH; ++i) halo_split(i, 0);
    }
    if (tl_on) tl[1] = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < nk; ++k) {
        if (tl_on && k < 40) tl[2 + k] = __builtin_amdgcn_s_memtime();
        // weights(k) (and, at tap 0, halo(cb)) have landed once only the later-issued DMAs remain outstanding:
        // weights(k+1), and the halo prefetch issued in step k-1 when that step was a tap 0
        const int kp = k - 1;
        // a halo DMA was issued in step k-1 (tap 0 of a block whose successor is DMA-sourced)
        const bool halo_prev = kp >= 0 && (kp % 9) == 0 && (kp / 9 + 1) < (UPIN ? nb_dma : p.ncb);
        if (p.flags & 0x100) wait_vmcnt<0>(); else wait_vmcnt_n((k + 1 < nk ? nlw : 0) + (halo_prev ? nlh : 0));
        if (!(p.flags & 0x800)) wg_barrier();                        // 0x800: ablation, no barrier
        auto produce = [&]() {
            const int cbj = k / 9, tapj = k - cbj * 9, nxt = cbj + 1;
            if (tapj == 0 && nxt < (UPIN ? nb_dma : p.ncb)) issue_h(nxt);   // other halo buffer: last read 9 steps ago
            if constexpr (UPIN) {
                if (nxt < p.ncb && nxt >= nb_dma) {                  // next block is upsampled: produce its halo pass by pass
                    if (tapj >= 1 && tapj - 1 < NLH) up_store(tapj - 1, nxt & 1);    // loads of step k-1 are complete (in-order, older than weights(k+1))
                    if (tapj < NLH) up_load(tapj, nxt - nb_dma);                     // issued BEFORE weights(k+2): the next wait covers them
                }
            }
            if (k + 2 < nk) issue_w(k + 2);                              // slot (k+2)%3 was read in step k-1
            if constexpr (is_x3<WT>) {
                if (tapj >= 2 && tapj - 2 < NLH && nxt < (UPIN ? nb_dma : p.ncb)) halo_split(tapj - 2, nxt & 1);
            }
        };
        // Round-2 timeline of this loop (tools/conv_bench.py --timeline --ablate N, s_memtime per K step, last FPN level, 1536 = the
        // MFMA-bound step): MFMA + LDS reads alone 1510-1565 | + barrier 2064 | DMA + barrier alone 1229 | everything 2697.  The
        // compute part alone runs AT the matrix-pipe limit; the per-step barrier (pipe drains, every wave re-reads LDS before its first
        // MFMA) costs ~550 and the concurrent DMA stream ~630 more.  WHERE in the step a wave issues its DMA pieces does not matter:
        // before / after its MFMAs staggered between the two waves of a SIMD, after K group 0 / 1 / 2 -- all 2660-2840 (measured, removed).
        if (!(p.flags & 0x100)) produce();                            // 0x100/0x200: ablation switches of tools/conv_bench.py
        if (!(p.flags & 0x200)) compute(k);
    }

    if (tl_on) tl[42] = __builtin_amdgcn_s_memtime();
    int mrow[SM];
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int oy = ty0 + wm * 4 + j * 2 + (lpix >> 4), ox = tx0 + (lpix & 15);
        mrow[j] = (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
    }
    if constexpr (TOPF && sizeof(WT) == 4) {
        static_assert(sizeof(OutT) == 4 && SN == 3 && WMQ == 4, "one 192-channel tile, 2 channel halves x 4 pixel quarters");
        conv_epilogue_topfuse_f32<SN, SM, NT, TN, TY * TX>(p, acc, smem_raw, wn * SN * 32, wm * SM * 32, half, lpix, [&](int row) {
            const int oy = ty0 + (row >> 4), ox = tx0 + (row & 15);
            return (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
        });
    } else if constexpr (TOPF) {
        static_assert(sizeof(WT) == 2 && sizeof(OutT) == 2, "");
        conv_epilogue_topfuse<WT, SN, SM, NT, TN, TY * TX>(p, acc, smem_raw, wn * SN * 32, wm * SM * 32, half, l31, lpix, wave, [&](int row) {
            const int oy = ty0 + (row >> 4), ox = tx0 + (row & 15);
            return (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
        });
    } else if (epi_lds_ok<OutT>(p) && (size_t)TY * TX * epi_pitch<OutT>(TN) + (size_t)16 * TN * 4 <= halo_lds_bytes<WT, CPR, SN, TOPF, WMQ>()) {
        conv_epilogue_lds<WT, OutT, SN, SM, NT, TN, TY * TX>(p, acc, smem_raw, n0, wn * SN * 32, wm * SM * 32, half, lpix, [&](int row) {
            const int oy = ty0 + (row >> 4), ox = tx0 + (row & 15);
            return (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
        });
    } else {
        conv_epilogue_rows<WT, OutT, SN, SM>(p, acc, mrow, n0 + wn * SN * 32, half);
    }
    if (tl_on) tl[43] = __builtin_amdgcn_s_memtime();
}

template <typename WT, typename OutT, int CPR, int SN, bool TOPF = false, bool UPIN = false, int WMQ = 4>
hipError_t launch_halo(ConvP p, hipStream_t s) {
    constexpr int E = 16 / (int)sizeof(WT);
    constexpr int TN = 2 * SN * 32;
    constexpr int TY = 4 * WMQ;
    constexpr size_t lds_bytes = halo_lds_bytes<WT, CPR, SN, TOPF, WMQ>();
    static_assert(!TOPF || (size_t)TY * 16 * TN * 2 + 16 * TN * 4 + 32 * TN * 2 <= lds_bytes, "image + bias rows + tap matrix must fit");
    auto kern = conv3x3_halo_kernel<WT, OutT, CPR, SN, TOPF, UPIN, WMQ>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.ncb = p.Cin / (CPR * E);
    p.nk = 9 * p.ncb;
    p.nN = (p.Cout + TN - 1) / TN;
    p.nblk_g = p.nN * p.B * ((p.Ho + TY - 1) / TY) * ((p.Wo + 15) / 16);
    p.nblk = p.nblk_g * (p.groups > 1 ? p.groups : 1);
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(128 * WMQ), lds_bytes, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, 192-channel tile, WEIGHTS THROUGH L1 ("wl1"): round-2 successor of conv3x3_halo_kernel for the
// fused last FPN level.
//
// The s_memtime timeline of conv3x3_halo_kernel showed its MFMA + LDS-read part running AT the matrix-pipe limit (1540 cycles per
// K step) and the rest of the 2700 going to the per-step barrier (weights ring: every wave must see every wave's DMA) and to the
// weight DMA stream competing with the fragment reads for LDS.  Here the weights never touch LDS:
//   * 12 waves = 6 channel blocks of 32 x 2 pixel halves of 128; a wave's A operand (its 32 channels x the step's 64 K) is 4
//     register fragments per step, fetched straight from global memory one full step ahead (double-buffered in VGPRs).  The two
//     pixel halves request the same lines (L1 hits); the weights are packed FRAGMENT-MAJOR (FTC_FLAG_W_FRAG: one fully
//     coalesced 1 KiB per wave-load: [row block][tap][channel block][K group][lane][8]) so a load touches 8 lines, not 32;
//   * only the activation halo (18x18 pixels x 64 channels, shared by all 12 waves, reused by the 9 taps) lives in LDS, double
//     buffered and refilled by DMA once per channel block -- so the workgroup synchronises once per NINE K steps instead of
//     every step, and in between the waves drift freely (one's fragment reads overlap another's MFMAs);
//   * LDS traffic per step: 12 waves x 16 B-fragment reads (192 KB) instead of 160 KB of A + B, no DMA writes except the halo.
// Same UPCAT_IN loader, TOP_FUSE / LDS-staged epilogues, pixel-slot mapping and output layout as conv3x3_halo_kernel.
// ------------------------------------------------------------------------------------------------
// WMH = pixel halves per workgroup.  2 (used): 12 waves, 16x16-pixel tile, one workgroup per CU (the epilogue image needs 120 KB).
// 1: 6 waves, 8x16-pixel tile, 72 KB -> TWO workgroups per CU with the same fragment loads per MFMA -- built and measured
// (parity-green): a workgroup then takes 96 k cycles for HALF the pixels against 107 k for the full tile, i.e. the CU is already
// saturated by 12 waves either way and the extra halo rows and L1 misses make it slower end to end (497 vs 532 img/s).  Not instantiated.
template <typename WT, typename OutT, bool TOPF = false, bool UPIN = false, int WMH = 2>
__global__ __launch_bounds__(384 * WMH, 3) void conv3x3_wl1_kernel(const ConvP p_launch) {
    ConvP p = p_launch;
    static_assert(sizeof(WT) == 2 && sizeof(OutT) == 2, "16-bit operands");
    constexpr int E = 8, CPR = 8, BK = 64, ROWB = 128, G = 4;
    constexpr int SM = 4, NT = 384 * WMH, TN = 192;
    constexpr int TY = 8 * WMH, TX = 16, HW = TX + 2, NH = (TY + 2) * HW;     // 324 | 180 halo pixels
    constexpr int HCH = NH * CPR, NLH = (HCH + NT - 1) / NT;                  // 2592 | 1440 chunks, 4 DMA passes
    constexpr int HBUF = NH * ROWB;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const hbase = smem_raw;                   // 2 halo buffers

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave / WMH, wm = wave % WMH;              // (the pixel halves of a channel block are neighbours: same lines, same time)
    const int half = lane >> 5, l31 = lane & 31;

    // PERSISTENT workgroups: XCD x (= blockIdx & 7, the hardware's round-robin) owns a contiguous eighth of the tiles (all heads of a
    // grouped launch laid end to end) and its gridDim/8 workgroups walk that range with stride gridDim/8.  While a workgroup runs the LAST
    // channel block of a tile, halo buffer 0 is already free, so the first halo of its NEXT tile is DMA'd then -- nine K steps plus the
    // epilogue ahead of its use -- and the epilogue stages its image behind buffer 0 (p.epi_off): the 4-10 k cycles a fresh workgroup
    // spent waiting for its first halo are gone, and so is the workgroup turnaround.  (Prefetch needs an even number of channel blocks --
    // the last one then sits in buffer 1 -- a next tile of the same head, and p.epi_off != 0, i.e. enough LDS for both.)
    const int tilesX = (p.Wo + TX - 1) / TX, tilesY = (p.Ho + TY - 1) / TY;
    int gt, gt_end, gstep;
    {
        const int total = p.nblk, q8 = total >> 3, r8 = total & 7, xcd = blockIdx.x & 7;
        const int cstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        gt = cstart + (int)(blockIdx.x >> 3);
        gt_end = cstart + q8 + (xcd < r8 ? 1 : 0);
        gstep = (int)(gridDim.x >> 3);
    }
    if (gt >= gt_end) return;
    constexpr int n0 = 0;
    const bool pf_ok = p.epi_off != 0 && !(p_launch.Cin / BK & 1);
    unsigned char* const epi_smem = smem_raw + p.epi_off;

    int grp = gt / p_launch.nblk_g;
    int img, ty0, tx0;
    auto set_tile = [&](int g_tile, int& im, int& y0, int& x0) {
        int sp = g_tile % p_launch.nblk_g;
        im = sp / (tilesX * tilesY);
        sp -= im * tilesX * tilesY;
        y0 = (sp / tilesX) * TY;
        x0 = (sp % tilesX) * TX;
    };
    enter_group(p, gt);
    set_tile(gt, img, ty0, tx0);

    __amdgpu_buffer_rsrc_t rw = weight_rsrc(p, 0);
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);

    const int ncb_up = UPIN ? p.Cy / BK : 0;
    const int nb_dma = p.ncb - ncb_up;
    __amdgpu_buffer_rsrc_t rin2 = UPIN ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in2u), 0, p.in2u_bytes, 0x00020000) : rin;
    auto phys_cb = [&](int j) { return UPIN ? (j < nb_dma ? ncb_up + j : j - nb_dma) : j; };

    int h_off[NLH];
    auto calc_hoff = [&](int im, int y0, int x0) {
        int tt = t;
        asm volatile("" : "+v"(tt));                          // (opaque: keeps the per-thread halo geometry from being hoisted out of the
                                                              //  tile loop, where it would pin ~30 registers through the K loop)
#pragma unroll
        for (int i = 0; i < NLH; ++i) {
            const int q = i * NT + tt;
            const int hr = q / CPR, kc = (q % CPR) ^ ((hr >> 1) & 7);
            const int iy = y0 - 1 + hr / HW, ix = x0 - 1 + hr % HW;
            const bool ok = q < HCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            if constexpr (UPIN) h_off[i] = ok ? (((im * p.H + iy) * p.W + ix) * (p.Cin - p.Cy) + kc * E) * 2 : OOB;
            else h_off[i] = ok ? (((im * p.H + iy) * p.W + ix) * p.CinT + p.cin_off + kc * E) * 2 : OOB;
        }
    };

    auto issue_h = [&](int cb) {                             // halo of channel block cb -> buffer cb & 1
        const int soff = cb * BK * 2;
        unsigned char* buf = hbase + (cb & 1) * HBUF;
#pragma unroll
        for (int i = 0; i < NLH; ++i) {
            if (i * NT + wave * 64 < HCH) {
                lds_void_t* dst = (lds_void_t*)(buf + (i * NT + wave * 64) * 16);
                if (i * NT + t < HCH) glds16(UPIN ? rin2 : rin, dst, h_off[i], soff);
            }
        }
    };

    using FragT = typename Frag<WT>::type;
    // A operand: fragment-major weights.  Fragment (row block rb, tap, channel block cb, K group g) = 1 KiB at
    // (((rb * 9 + tap) * ncb + cb) * G + g) * 1024; lane L reads its 16 bytes at L * 16.
    const int a_voff = lane * 16;
    auto loadA = [&](int k, FragT (&dst)[G]) {
        const int cb = k / 9, tap = k - cb * 9;
        const int soff = (((wn * 9 + tap) * p.ncb + phys_cb(cb)) * G) * 1024;
#pragma unroll
        for (int g = 0; g < G; ++g) dst[g] = __builtin_bit_cast(FragT, bload(rw, a_voff, soff + g * 1024));
    };

    f32x16 acc[1][SM];

    const int lpix = ((__builtin_popcount(l31 >> 2) & 1) << 4) | ((l31 >> 3) << 2) | (l31 & 3);      // see conv3x3_halo_kernel
    int hr0[SM];
#pragma unroll
    for (int j = 0; j < SM; ++j) hr0[j] = (wm * 8 + j * 2 + (lpix >> 4)) * HW + (lpix & 15);

    const int nk = 9 * p.ncb;
    // K step: the B fragments of a K group are read from LDS as ONE batch, a group ahead of the MFMAs that use them, and the A fragment of
    // group g is reloaded IN PLACE for the next step right after its MFMAs (a full step of prefetch distance without a second register
    // set; after the last step of a tile that is step 0 again -- the next tile of the same head starts with its weights in registers).
    // The scheduling barriers pin that order: left alone under the 168-register cap of three waves per SIMD, the compiler sank every
    // ds_read to just above its MFMA -- read, wait, multiply, sixteen times per step.  BB = B fragments per batch (the upsampling
    // variant has registers for two).
    FragT a_frag[G];
    constexpr int BB = UPIN ? 2 : SM;
    static_assert(SM % BB == 0, "");
    auto compute = [&](int k) {
        const int cb = k / 9, tap = k - cb * 9;
        const int d = (tap / 3) * HW + (tap % 3);
        const unsigned char* hb = hbase + (cb & 1) * HBUF;
        int rowB[SM], f4[SM];
#pragma unroll
        for (int j = 0; j < SM; ++j) {
            const int hr = hr0[j] + d;
            rowB[j] = hr * ROWB;
            f4[j] = ((hr >> 1) & 7) << 4;
        }
        const int k1 = k + 1 == nk ? 0 : k + 1, cb1 = k1 / 9, tap1 = k1 - cb1 * 9;
        const int a_soff = (((wn * 9 + tap1) * p.ncb + phys_cb(cb1)) * G) * 1024;
        constexpr int NB = G * (SM / BB);                   // batches per step
        FragT bf[2][BB];
        auto ldbatch = [&](int b, int s) {
            const int g = b / (SM / BB), j0 = (b % (SM / BB)) * BB;
#pragma unroll
            for (int j = 0; j < BB; ++j) bf[s][j] = *reinterpret_cast<const FragT*>(hb + rowB[j0 + j] + ((((g * 2 + half) << 4)) ^ f4[j0 + j]));
        };
        // Timing ablations of this step on MI355X (single head, 1152 one-shot workgroups, kernel time; switches since removed): full 311 us;
        // no A reloads 311; no B reads 289; neither 274; neither and a quarter of the MFMAs 168; no block barrier 318 (the wait moves to the
        // epilogue's first barrier).  I.e. the operand traffic costs 12 %, the rest is the matrix pipe at the 1.75-1.8 GHz the part
        // sustains on real data (tools/ubench/mfma_peak.hip: 32.0 ticks per MFMA, 1.79 G ticks/s) plus the per-block barrier skew.
        ldbatch(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int s = b & 1, g = b / (SM / BB), j0 = (b % (SM / BB)) * BB;
            if (b + 1 < NB) ldbatch(b + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < BB; ++j) acc[0][j0 + j] = mfma16(a_frag[g], bf[s][j], acc[0][j0 + j]);
            if (j0 + BB == SM) a_frag[g] = __builtin_bit_cast(FragT, bload(rw, a_voff, a_soff + g * 1024));
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- UPIN: identical to conv3x3_halo_kernel (one 16-byte chunk of the halo image per thread and pass) ----
    constexpr int NUP = UPIN ? NLH : 1;
    int up_off[NUP][4];
    float up_ly[NUP], up_lx[NUP];
    auto calc_up = [&]() {
        if constexpr (UPIN) {
            int tt = t;
            asm volatile("" : "+v"(tt));
#pragma unroll
            for (int i = 0; i < NLH; ++i) {
                const int q = i * NT + tt;
                const int hr = q / CPR, kc = (q % CPR) ^ ((hr >> 1) & 7);
                const int iy = ty0 - 1 + hr / HW, ix = tx0 - 1 + hr % HW;
                const bool ok = q < HCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const float sy = p.ry * (float)iy, sx = p.rx * (float)ix;
                const int y0 = ok ? (int)sy : 0, x0 = ok ? (int)sx : 0;
                const int y1 = y0 + (y0 < p.Hi - 1 ? 1 : 0), x1 = x0 + (x0 < p.Wi - 1 ? 1 : 0);
                up_ly[i] = sy - (float)y0;
                up_lx[i] = sx - (float)x0;
                const int r0 = (img * p.Hi + y0) * p.Wi, r1 = (img * p.Hi + y1) * p.Wi;
                const int pb = p.Cy * 2, cbase = kc * E * 2;
                up_off[i][0] = ok ? (r0 + x0) * pb + cbase : OOB;
                up_off[i][1] = ok ? (r0 + x1) * pb + cbase : OOB;
                up_off[i][2] = ok ? (r1 + x0) * pb + cbase : OOB;
                up_off[i][3] = ok ? (r1 + x1) * pb + cbase : OOB;
            }
        }
    };
    u32x4 st[4] = {};
    auto up_load = [&](int pass, int ub) {
        const int soff = ub * BK * 2;
#pragma unroll
        for (int i = 0; i < NUP; ++i)
            if (i == pass) {
#pragma unroll
                for (int c = 0; c < 4; ++c) st[c] = bload(rin, up_off[i][c], soff);
            }
    };
    auto up_store = [&](int pass, int bufidx) {
        const int q = pass * NT + t;
        if (q >= HCH) return;
        float ly1 = 0.f, lx1 = 0.f;
#pragma unroll
        for (int i = 0; i < NUP; ++i)
            if (i == pass) { ly1 = up_ly[i]; lx1 = up_lx[i]; }
        const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
        float v[8];
        const FragT c0 = __builtin_bit_cast(FragT, st[0]), c1 = __builtin_bit_cast(FragT, st[1]);
        const FragT c2 = __builtin_bit_cast(FragT, st[2]), c3 = __builtin_bit_cast(FragT, st[3]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = ly0 * (lx0 * (float)c0[e] + lx1 * (float)c1[e]) + ly1 * (lx0 * (float)c2[e] + lx1 * (float)c3[e]);
        store16<WT>(reinterpret_cast<WT*>(hbase + bufidx * HBUF + q * 16), v);
    };

    calc_hoff(img, ty0, tx0);
    issue_h(0);
    loadA(0, a_frag);
    for (int it = 0;; ++it) {
        // 0x1000: timeline of wave 0 of every workgroup's SECOND tile (the steady state) into p.res (tools/conv_bench.py --timeline)
        const bool tl_on = (p.flags & 0x1000) && blockIdx.x < 512 && t == 0 && it == 1;
        unsigned long long* tl = reinterpret_cast<unsigned long long*>(const_cast<void*>(p_launch.res)) + (size_t)blockIdx.x * 64;
        if (tl_on) tl[0] = tl[1] = __builtin_amdgcn_s_memtime();
        calc_up();
#pragma unroll
        for (int j = 0; j < SM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.0f;
        const int gnext = gt + gstep;
        const bool has_next = gnext < gt_end;
        const bool pf = pf_ok && has_next && gnext / p_launch.nblk_g == grp;
        for (int k = 0; k < nk; ++k) {
            if (tl_on && k < 40) tl[2 + k] = __builtin_amdgcn_s_memtime();
            const int cbj = k / 9, tapj = k - cbj * 9, nxt = cbj + 1;
            if (tapj == 0) {
                // channel-block boundary: this block's halo has landed (this wave's pieces: vmcnt; everyone's: barrier) and every wave is
                // done reading the other buffer, which the next block's halo is about to overwrite.  LDS writes of the in-loader upsample
                // retire in order with the fragment reads that followed them.
                wait_vmcnt<0>();
                wg_barrier();
                if (nxt < (UPIN ? nb_dma : p.ncb)) {
                    issue_h(nxt);
                } else if (nxt == p.ncb && pf) {                     // last block: buffer 0 is free -> first halo of the next tile
                    int im2, y2, x2;
                    set_tile(gnext, im2, y2, x2);
                    calc_hoff(im2, y2, x2);
                    issue_h(0);
                }
            }
            if constexpr (UPIN) {
                if (nxt < p.ncb && nxt >= nb_dma) {                  // next block is upsampled: produce its halo pass by pass
                    if (tapj >= 1 && tapj - 1 < NLH) up_store(tapj - 1, nxt & 1);
                    if (tapj < NLH) up_load(tapj, nxt - nb_dma);
                }
            }
            compute(k);
        }

        if (tl_on) tl[42] = __builtin_amdgcn_s_memtime();
        auto row_to_m = [&](int row) {
            const int oy = ty0 + (row >> 4), ox = tx0 + (row & 15);
            return (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
        };
        // (opaque copies: the epilogue's per-lane LDS addresses are tile-invariant, and hoisted out of the tile loop they were spilled
        //  to scratch and reloaded one by one inside the epilogue -- ~45 dependent scratch loads per tile)
        int e_half = half, e_l31 = l31, e_lpix = lpix;
        asm volatile("" : "+v"(e_half), "+v"(e_l31), "+v"(e_lpix));
        if constexpr (TOPF) {
            conv_epilogue_topfuse<WT, 1, SM, NT, TN, TY * TX>(p, acc, epi_smem, wn * 32, wm * SM * 32, e_half, e_l31, e_lpix, wave, row_to_m);
        } else if (epi_lds_ok<OutT>(p)) {
            conv_epilogue_lds<WT, OutT, 1, SM, NT, TN, TY * TX>(p, acc, epi_smem, n0, wn * 32, wm * SM * 32, e_half, e_lpix, row_to_m);
        } else {
            int mrow[SM];
#pragma unroll
            for (int j = 0; j < SM; ++j) {
                const int oy = ty0 + wm * 8 + j * 2 + (e_lpix >> 4), ox = tx0 + (e_lpix & 15);
                mrow[j] = (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
            }
            conv_epilogue_rows<WT, OutT, 1, SM>(p, acc, mrow, n0 + wn * 32, e_half);
        }
        if (tl_on) tl[43] = __builtin_amdgcn_s_memtime();
        if (!has_next) break;

        gt = gnext;
        const int g2 = gt / p_launch.nblk_g;
        const bool new_group = g2 != grp;
        if (new_group) {                                             // next head of a grouped launch: its operands
            p = p_launch;
            enter_group(p, gt);
            grp = g2;
            rw = weight_rsrc(p, 0);
            rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
            rin2 = UPIN ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in2u), 0, p.in2u_bytes, 0x00020000) : rin;
        }
        set_tile(gt, img, ty0, tx0);
        if (!pf) {
            __syncthreads();                                         // everyone is done with the epilogue image (it may overlap buffer 0)
            calc_hoff(img, ty0, tx0);
            issue_h(0);
        }
        if (new_group) loadA(0, a_frag);                             // (same head: step 0's fragments were reloaded by the last K step)
    }
}

template <typename WT, typename OutT, bool TOPF = false, bool UPIN = false, int WMH = 2>
hipError_t launch_wl1(ConvP p, hipStream_t s) {
    constexpr int TY = 8 * WMH;
    constexpr size_t HBUF = (size_t)(TY + 2) * 18 * 128, LDS_MAX = 160 * 1024;
    auto kern = conv3x3_wl1_kernel<WT, OutT, TOPF, UPIN, WMH>;
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
        if (e != hipSuccess) return e;
        int dev = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        attr_set = true;
    }
    p.ncb = p.Cin / 64;
    p.nk = 9 * p.ncb;
    p.nN = 1;
    p.nblk_g = p.B * ((p.Ho + TY - 1) / TY) * ((p.Wo + 15) / 16);
    p.nblk = p.nblk_g * (p.groups > 1 ? p.groups : 1);
    // epilogue image (+ tap matrix rows in use) + bias table behind halo buffer 0 when the LDS holds both: lets a workgroup fetch its next
    // tile's first halo while it finishes the current tile
    const size_t epi = (size_t)TY * 16 * 192 * 2 + (size_t)16 * 192 * 4 + (TOPF ? (size_t)((p.Tw + 3) & ~3) * 192 * 2 : 0);
    p.epi_off = HBUF + epi <= LDS_MAX ? (int)HBUF : 0;
    size_t lds_bytes = p.epi_off + epi;
    if (lds_bytes < 2 * HBUF) lds_bytes = 2 * HBUF;
    if (lds_bytes > LDS_MAX) return hipErrorInvalidValue;
    // persistent grid: one workgroup per CU (the LDS admits no second one), a multiple of 8 so that every XCD gets the same number
    int grid = ((n_cu > 0 ? n_cu : 256) + 7) / 8 * 8;
    // (measured on the 8-head launch, 36 tiles per CU: one-shot workgroups 2114 us, persistent 2065, persistent + halo prefetch 2054)
    if (grid > (p.nblk + 7) / 8 * 8) grid = (p.nblk + 7) / 8 * 8;
    // test hook: a smaller grid makes every workgroup walk several tiles (and heads) even on the small shapes of the unit tests
    if (const char* e = getenv("FTC_WL1_GRID_CAP")) {
        const int c = atoi(e) / 8 * 8;
        if (c >= 8 && c < grid) grid = c;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(384 * WMH), lds_bytes, s, p);
    return hipGetLastError();
}

template <typename WT, typename InT, typename OutT, int BK, int WN, int WM, int SN, int SM, int NBUF, bool SE, int KG = 1>
hipError_t launch_cfg2(ConvP p, hipStream_t s) {
    constexpr int E = 16 / (int)sizeof(WT);
    constexpr int ROW = BK + E;
    constexpr int TN = WN * SN * 32, TM = WM * SM * 32;
    constexpr size_t lds_stage = (size_t)KG * NBUF * (TN + TM) * ROW * sizeof(WT);
    constexpr size_t lds_epi = KG == 1 ? (size_t)TM * epi_pitch<OutT>(TN) + (size_t)16 * TN * 4 : (size_t)KG * TM * epi_pitch<float>(TN);
    constexpr size_t lds_bytes = lds_stage > lds_epi ? lds_stage : lds_epi;
    auto kern = conv_igemm_kernel<WT, InT, OutT, BK, WN, WM, SN, SM, NBUF, SE, KG>;
    static bool attr_set = false;     // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.ncb = (p.Cin + BK - 1) / BK;
    p.nk = p.KS * p.KS * p.ncb;
    p.nN = (p.Cout + TN - 1) / TN;
    const int nM = (p.M + TM - 1) / TM;
    p.nblk_g = p.nN * nM;
    p.nblk = p.nblk_g * (p.groups > 1 ? p.groups : 1);
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(256 * KG), lds_bytes, s, p);
    return hipGetLastError();
}

template <typename WT, typename OutT, int BK, int WN, int WM, int SN, int SM, int NBUF>
hipError_t launch_glds(ConvP p, hipStream_t s) {
    constexpr int E = 16 / (int)sizeof(WT);
    constexpr int TN = WN * SN * 32, TM = WM * SM * 32;
    constexpr size_t lds_stage = (size_t)NBUF * (TN + TM) * (BK / E) * 16;
    constexpr size_t lds_epi = (size_t)TM * epi_pitch<OutT>(TN) + (size_t)16 * TN * 4;
    constexpr size_t lds_bytes = lds_stage > lds_epi ? lds_stage : lds_epi;
    auto kern = conv_igemm_glds_kernel<WT, OutT, BK, WN, WM, SN, SM, NBUF>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.ncb = (p.Cin + BK - 1) / BK;
    p.nk = p.KS * p.KS * p.ncb;
    p.nN = (p.Cout + TN - 1) / TN;
    p.nblk_g = p.nN * ((p.M + TM - 1) / TM);
    p.nblk = p.nblk_g * (p.groups > 1 ? p.groups : 1);
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(256), lds_bytes, s, p);
    return hipGetLastError();
}

template <typename WT, typename InT, typename OutT, int BK, int WN, int WM, int SN, int SM, int NBUF>
hipError_t launch_cfg(const ConvP& p, hipStream_t s) {
    constexpr int E = 16 / (int)sizeof(WT);
    constexpr bool glds_ok = sizeof(WT) == sizeof(InT) && (BK / E == 8 || BK / E == 4) &&
                             (((WN * SN + WM * SM) * 32 * (BK / E)) % 256 == 0) && ((WN * SN * 32 * (BK / E)) % 64 == 0);
    if constexpr (glds_ok) {
        if (p.use_glds) {
            if (p.flags & FTC_FLAG_SE_SCALE) return hipErrorInvalidValue;      // (glds_legal keeps such ops away: the DMA cannot rescale)
            if (p.glds_nbuf == 3) return launch_glds<WT, OutT, BK, WN, WM, SN, SM, 3>(p, s);
            return launch_glds<WT, OutT, BK, WN, WM, SN, SM, 2>(p, s);
        }
    }
    // intra-workgroup split-K (tuned per layer): long-K 1x1 convs on small tiles
    if constexpr (sizeof(WT) == 2 && sizeof(InT) == 2 && WN * SN * WM * SM <= 8 && BK >= 64) {
        if (p.split_k == 2) {
            if (p.flags & FTC_FLAG_SE_SCALE) return launch_cfg2<WT, InT, OutT, BK, WN, WM, SN, SM, NBUF, true, 2>(p, s);
            return launch_cfg2<WT, InT, OutT, BK, WN, WM, SN, SM, NBUF, false, 2>(p, s);
        }
        if (p.split_k == 4) {
            if (p.flags & FTC_FLAG_SE_SCALE) return launch_cfg2<WT, InT, OutT, BK, WN, WM, SN, SM, NBUF, true, 4>(p, s);
            return launch_cfg2<WT, InT, OutT, BK, WN, WM, SN, SM, NBUF, false, 4>(p, s);
        }
    }
    // the SE-scaled variant exists only where the network uses it: 1x1 project convs
    if (p.flags & FTC_FLAG_SE_SCALE) return launch_cfg2<WT, InT, OutT, BK, WN, WM, SN, SM, NBUF, true>(p, s);
    return launch_cfg2<WT, InT, OutT, BK, WN, WM, SN, SM, NBUF, false>(p, s);
}

// Tile configurations (output channels x output pixels per workgroup), in order of preference.
// (tried in round 2 and removed: a 256x64 tile (4 waves side by side over N) for the wide MBConv expand GEMMs -- 0.6x the L2->LDS bytes
//  per FLOP of the 64x64 tile -- never won in the tuner: gpurun_out/tuning_r2_1x1.log)
// The x144 configs are a kernel of their own (conv1x1_px144.hip: 1x1, 16-bit operands, fp32 output), chosen only by hint.
enum { CFG_192x128 = 0, CFG_128x128, CFG_96x128, CFG_64x128, CFG_128x64, CFG_32x256, CFG_64x64, CFG_64x144, CFG_80x144, CFG_128x144, CFG_96x144, CFG_COUNT };
static const char* const kCfgName[] = {"192x128", "128x128", "96x128", "64x128", "128x64", "32x256", "64x64", "64x144", "80x144", "128x144", "96x144"};
static const int kCfgTN[] = {192, 128, 96, 64, 128, 32, 64, 64, 80, 128, 96};
static const int kCfgTM[] = {128, 128, 128, 128, 64, 256, 64, 144, 144, 144, 144};
inline bool cfg_px144(int cfg) { return cfg >= CFG_64x144 && cfg <= CFG_96x144; }

// ftc_op.aux0 carries the tuned kernel choice (0 = heuristics below): bits 0-3 tile config + 1,
// bits 4-5 staging (1 = register-staged, 2 = direct-to-LDS 2-slot ring, 3 = 3-slot ring), bits 8-9 K step
// (1 = 32, 2 = 64, 3 = 128).  The Python side fills it from a table measured on MI355X
// (findtextcenternet_amd/tuning.py); the choices compute the same convolution -- bit-identical among the tile configs and stagings (same K
// order), in another fp32 summation order with split-K and on the 144-pixel tiles.
inline int hint_cfg(const ftc_op& o) { return (o.aux0 & 15) - 1; }
inline bool hint_halo(const ftc_op& o) { return (o.aux0 & 64) != 0; }         // bit 6: LDS-halo 3x3 kernel
inline bool hint_wl1(const ftc_op& o) { return (o.aux0 & 192) == 192; }       // bits 6+7: its weights-through-L1 successor (needs FTC_FLAG_W_FRAG weights)
inline int hint_splitk(const ftc_op& o) { const int c = (o.aux0 >> 10) & 3; return c == 1 ? 2 : c == 2 ? 4 : 1; }   // bits 10-11
inline int hint_stage(const ftc_op& o) { return (o.aux0 >> 4) & 3; }
inline int hint_bk(const ftc_op& o) { const int b = (o.aux0 >> 8) & 3; return b == 1 ? 32 : b == 2 ? 64 : b == 3 ? 128 : 0; }

inline int default_cfg(int n, int M) {
    if (n <= 32) return CFG_32x256;
    if (n <= 64) return CFG_64x128;
    if (n <= 96) return CFG_96x128;
    const long t192 = (long)((n + 191) / 192) * ((M + 127) / 128);
    if (n % 192 == 0 && n % 128 != 0) return t192 >= 256 ? CFG_192x128 : CFG_64x64;
    // 128-channel tiles; shrink the pixel tile when the grid would not fill the 256 CUs twice
    const long tiles128 = (long)((n + 127) / 128) * ((M + 127) / 128);
    return tiles128 < 512 ? CFG_128x64 : CFG_128x128;
}
inline int select_cfg(const ftc_op& o) {
    const int h = hint_cfg(o);
    if (h >= 0 && h < CFG_COUNT) return h;
    const int d = default_cfg(o.Cout, o.B * o.Ho * o.Wo * (o.groups > 1 ? o.groups : 1));
    // per-image weight sets: the pixel tile must divide the image
    if ((o.flags & FTC_FLAG_W_PER_IMAGE) && (o.Ho * o.Wo) % kCfgTM[d]) return o.Cout > 64 ? CFG_128x64 : CFG_64x64;
    return d;
}
inline bool px144_legal(const ftc_op& o, int cfg) {
    // 16-bit operands, or the fp16x3 form with BOTH operands pre-split (fp32 tensors, FTC_FLAG_SPLIT16 | FTC_FLAG_PRESPLIT); K step 64
    const bool x3 = o.w_dtype == FTC_F32 && (o.flags & FTC_FLAG_SPLIT16) && (o.flags & FTC_FLAG_PRESPLIT) && !(o.flags & FTC_FLAG_KBLOCK32);
    const bool h16 = ftc_is16(o.w_dtype) && !(o.flags & FTC_FLAG_PRESPLIT);
    const int ks = 64;
    return o.ksize == 1 && o.stride == 1 && o.act == FTC_ACT_NONE && (x3 || h16) && o.in_dtype == o.w_dtype && o.out_dtype == FTC_F32 && o.Cin >= ks && o.Cin % ks == 0 &&
           o.Cout % kCfgTN[cfg] == 0 && ((o.Cout_total | o.cout_off | o.Cin_total | o.cin_off) & 7) == 0 && (o.Ho * o.Wo) % 144 == 0 && o.groups <= 1 &&
           !(o.flags & (FTC_FLAG_SE_SCALE | FTC_FLAG_BORDER_BIAS | FTC_FLAG_UPCAT_IN | FTC_FLAG_TOP_FUSE | FTC_FLAG_GROUP_OUT_SLICE));
}
inline bool wset_legal(const ftc_op& o) {
    if (!(o.flags & FTC_FLAG_W_PER_IMAGE)) return true;
    if (hint_halo(o)) return true;                                   // the halo kernel tiles each image separately
    return (o.Ho * o.Wo) % kCfgTM[select_cfg(o)] == 0;
}

// K step: 64 for bf16 when the channel count allows (half the barriers per FLOP), else 32; 128 only by hint.
inline int select_bk(const ftc_op& o) {
    if (!ftc_is16(o.w_dtype)) return 32;
    const int h = hint_bk(o);
    if (h) return h;
    return o.Cin % 64 == 0 ? 64 : 32;
}
inline bool glds_legal(const ftc_op& o) {
    if (o.in_dtype != o.w_dtype) return false;
    if (cfg_px144(select_cfg(o))) return false;
    const int bk = select_bk(o);
    if (bk == 128) return false;
    const int cpr = bk / (ftc_is16(o.w_dtype) ? 8 : 4);
    const int cfg = select_cfg(o);
    if (o.flags & FTC_FLAG_SE_SCALE) return false;      // the SE scale is applied while staging through registers
    // tiles must be a whole number of workgroup-level DMA passes
    return ((kCfgTN[cfg] + kCfgTM[cfg]) * cpr) % 256 == 0 && (kCfgTN[cfg] * cpr) % 64 == 0;
}
inline bool uses_glds(const ftc_op& o) {
    if (!glds_legal(o) || hint_splitk(o) > 1) return false;
    const int st = hint_stage(o);
    if (st) return st >= 2;
    // Untuned default (tools/conv_bench.py, MI355X): the 2-slot DMA ring wins on the 192x128 and 64x128
    // tiles (FPN: 819 vs 746 TF); on 128-channel tiles the register-staged kernel keeps 3-4 workgroups
    // per CU with its single 36 KB buffer and is faster (stage2 3x3: 504 vs 439 TF).  fp32: DMA everywhere.
    if (o.w_dtype == FTC_F32) return true;
    const int cfg = select_cfg(o);
    return cfg == CFG_192x128 || cfg == CFG_64x128;
}
inline int glds_ring(const ftc_op& o) { return hint_stage(o) == 3 ? 3 : 2; }
// intra-workgroup split-K: register-staged kernel, bf16 activations, K step >= 64, tiles of <= 8 MFMA sub-tiles
// (64x64, 64x128, 128x64), and a K loop that divides evenly
inline bool splitk_legal(const ftc_op& o, int kg) {
    if (kg == 1) return true;
    if (!ftc_is16(o.w_dtype) || o.in_dtype != o.w_dtype || select_bk(o) < 64) return false;
    const int cfg = select_cfg(o);
    if (!(cfg == CFG_64x64 || cfg == CFG_64x128 || cfg == CFG_128x64)) return false;
    const int bk = select_bk(o);
    const long lds = (long)kg * (kCfgTN[cfg] + kCfgTM[cfg]) * (bk + 8) * 2;           // KG staging buffers (bf16, padded rows)
    if (lds > 160 * 1024) return false;
    const int nk = o.ksize * o.ksize * ((o.Cin + bk - 1) / bk);
    return nk % kg == 0 && nk / kg >= 2;
}
// LDS-halo kernel: 3x3 stride 1, activations in the compute dtype, whole channel blocks, tile = 64/128/192 channels
inline int halo_sn(const ftc_op& o) { const int c = select_cfg(o); return c == CFG_192x128 ? 3 : c == CFG_128x128 ? 2 : c == CFG_64x128 ? 1 : 0; }
inline int halo_cpr(const ftc_op& o) {
    if (o.w_dtype == FTC_F32) return o.Cin % 32 == 0 ? 8 : 0;
    // 128-byte rows (K step 64) unless the channel count or the tuning hint (bk = 32) asks for 64-byte rows: those halve
    // the LDS footprint, so two workgroups share a CU and one's epilogue overlaps the other's K loop
    if (hint_bk(o) == 32 && hint_halo(o) && !(o.flags & (FTC_FLAG_TOP_FUSE | FTC_FLAG_UPCAT_IN))) return o.Cin % 32 == 0 ? 4 : 0;
    return o.Cin % 64 == 0 ? 8 : (o.Cin % 32 == 0 ? 4 : 0);
}
inline bool halo_legal(const ftc_op& o) {
    return o.ksize == 3 && o.stride == 1 && o.in_dtype == o.w_dtype && !(o.flags & FTC_FLAG_SE_SCALE) && halo_sn(o) > 0 && halo_cpr(o) > 0;
}
inline bool uses_halo(const ftc_op& o) { return hint_halo(o) && halo_legal(o); }

template <typename WT, typename InT, typename OutT, int BK>
hipError_t launch_tiles(const ConvP& p, int cfg, hipStream_t s) {
    switch (cfg) {
    case CFG_192x128: return launch_cfg<WT, InT, OutT, BK, 2, 2, 3, 2, 1>(p, s);
    case CFG_128x128: return launch_cfg<WT, InT, OutT, BK, 2, 2, 2, 2, 1>(p, s);
    case CFG_96x128: return launch_cfg<WT, InT, OutT, BK, 1, 4, 3, 1, 1>(p, s);
    case CFG_64x128: return launch_cfg<WT, InT, OutT, BK, 2, 2, 1, 2, 1>(p, s);
    case CFG_128x64: return launch_cfg<WT, InT, OutT, BK, 2, 2, 2, 1, 1>(p, s);
    case CFG_32x256: return launch_cfg<WT, InT, OutT, BK, 1, 4, 1, 2, 1>(p, s);
    default: return launch_cfg<WT, InT, OutT, BK, 2, 2, 1, 1, 1>(p, s);
    }
}

template <typename WT, typename OutT>
hipError_t launch_halo_dispatch(const ConvP& p, const ftc_op& o, hipStream_t s) {
    const int sn = halo_sn(o), cpr = halo_cpr(o);
    if constexpr (sizeof(WT) == 2 && sizeof(OutT) == 2) {
        if (hint_wl1(o) && (o.flags & FTC_FLAG_W_FRAG) && cpr == 8 && sn == 3) {
            const bool up = (o.flags & FTC_FLAG_UPCAT_IN) != 0;
            if (o.flags & FTC_FLAG_TOP_FUSE) return up ? launch_wl1<WT, OutT, true, true>(p, s) : launch_wl1<WT, OutT, true, false>(p, s);
            return up ? launch_wl1<WT, OutT, false, true>(p, s) : launch_wl1<WT, OutT, false, false>(p, s);
        }
    }
    if (o.flags & FTC_FLAG_TOP_FUSE) {
        if constexpr (sizeof(WT) == 2 && sizeof(OutT) == 2) {
            if (cpr == 8 && sn == 3) return (o.flags & FTC_FLAG_UPCAT_IN) ? launch_halo<WT, OutT, 8, 3, true, true>(p, s) : launch_halo<WT, OutT, 8, 3, true>(p, s);
        }
        if constexpr (sizeof(WT) == 4 && sizeof(OutT) == 4) {                  // fp32 / fp16x3: the FMA epilogue (conv_epilogue_topfuse_f32)
            if (cpr == 8 && sn == 3) return (o.flags & FTC_FLAG_UPCAT_IN) ? launch_halo<WT, OutT, 8, 3, true, true>(p, s) : launch_halo<WT, OutT, 8, 3, true>(p, s);
        }
        return hipErrorInvalidValue;
    }
    if (o.flags & FTC_FLAG_UPCAT_IN) {
        if constexpr (sizeof(WT) == 2 && sizeof(OutT) == 2) {
            if (sn == 3) return cpr == 8 ? launch_halo<WT, OutT, 8, 3, false, true>(p, s) : launch_halo<WT, OutT, 4, 3, false, true>(p, s);
        }
        if constexpr (sizeof(WT) == 4 && sizeof(OutT) == 4) {
            if (sn == 3 && cpr == 8) return launch_halo<WT, OutT, 8, 3, false, true>(p, s);
        }
        return hipErrorInvalidValue;
    }
    if (cpr == 8) {
        if (sn == 3) return launch_halo<WT, OutT, 8, 3>(p, s);
        if (sn == 2) return launch_halo<WT, OutT, 8, 2>(p, s);
        return launch_halo<WT, OutT, 8, 1>(p, s);
    }
    if constexpr (sizeof(WT) == 2) {
        if (sn == 3) return launch_halo<WT, OutT, 4, 3>(p, s);
        if (sn == 2) return launch_halo<WT, OutT, 4, 2>(p, s);
        return launch_halo<WT, OutT, 4, 1>(p, s);
    }
    return hipErrorInvalidValue;
}

// The instantiations of one type combination fall into four independent parts (halo kernels; tiles with K step 32 / 64 / 128),
// so that the two heavy combinations can be compiled as four translation units each (conv_igemm_part.hip).
enum { PART_HALO = 0, PART_BK32 = 1, PART_BK64 = 2, PART_BK128 = 3 };

template <typename WT, typename InT>
int conv_part(const ftc_op& o) {
    if constexpr (sizeof(WT) == sizeof(InT)) {
        if (uses_halo(o)) return PART_HALO;
    }
    if constexpr (sizeof(WT) == 2) {
        const int bk = select_bk(o);
        if constexpr (sizeof(InT) == 2) {
            if (bk == 128) return PART_BK128;
        }
        if (bk >= 64) return PART_BK64;
    }
    return PART_BK32;
}

template <typename WT, typename InT, typename OutT, int PART>
hipError_t launch_part(const ConvP& p, const ftc_op& o, hipStream_t s) {
    if constexpr (PART == PART_HALO) {
        if constexpr (sizeof(WT) == sizeof(InT)) return launch_halo_dispatch<WT, OutT>(p, o, s);
        else return hipErrorInvalidValue;
    } else if constexpr (PART == PART_BK128) {
        if constexpr (sizeof(WT) == 2 && sizeof(InT) == 2) return launch_tiles<WT, InT, OutT, 128>(p, select_cfg(o), s);
        else return hipErrorInvalidValue;
    } else if constexpr (PART == PART_BK64) {
        if constexpr (sizeof(WT) == 2) return launch_tiles<WT, InT, OutT, 64>(p, select_cfg(o), s);
        else return hipErrorInvalidValue;
    } else {
        return launch_tiles<WT, InT, OutT, 32>(p, select_cfg(o), s);
    }
}

template <typename WT, typename InT, typename OutT>
hipError_t launch_types(const ConvP& p, const ftc_op& o, hipStream_t s) {
    switch (conv_part<WT, InT>(o)) {
    case PART_HALO: return launch_part<WT, InT, OutT, PART_HALO>(p, o, s);
    case PART_BK128: return launch_part<WT, InT, OutT, PART_BK128>(p, o, s);
    case PART_BK64: return launch_part<WT, InT, OutT, PART_BK64>(p, o, s);
    default: return launch_part<WT, InT, OutT, PART_BK32>(p, o, s);
    }
}

}  // namespace convimpl
