// FTC_OP_WGRAD: weight gradient of a dense 1x1 / 3x3 convolution (stride 1 | 2) on the matrix cores -- the half of
// `loss.backward()` (/root/reference/train1.py:170-179) that autograd spends in conv2d's weight gradient.
//
//   dW[co][tap][ci] = sum over output pixels p of  dZ[p][co] * X[p @ tap][ci]          (X zero outside the image)
//
// A GEMM whose CONTRACTION runs over pixels, i.e. over the slow axis of both NHWC operands ("TN" form).  The operands are fp32 in
// HBM (the training-mode plan keeps activations fp32), so they are register-staged anyway: a lane loads 4 consecutive channels of 8
// consecutive pixels, narrows them to the MFMA operand type and writes four 16-byte, pixel-contiguous rows into LDS -- the transpose
// happens in the registers of the staging pass, and the MFMA fragments are plain ds_read_b128 of [channel][8 pixels].
// Grid = (Cout tiles, Cin tiles x taps, pixel splits): every workgroup reduces its pixel range for one (tile, tap) and writes an
// fp32 partial tile; wgrad_reduce_kernel sums the splits in a fixed order (deterministic) and ADDS the result to the gradient buffer
// in the PyTorch layout [Cout][Cin][k][k].
#include "ftc_common.h"

namespace {

template <typename WT> struct WFrag;
template <> struct WFrag<__bf16> { using type = bf16x8; };
template <> struct WFrag<_Float16> { using type = f16x8; };

// gfx950 LDS transpose read: inside every 16-lane group, lane 4*j + q supplies the address of 4 consecutive 16-bit elements = columns
// 4q..4q+3 of row j of a 4 x 16 block, and lane i receives column i of the four rows (measured: tools/ubench/tr16_probe.hip).  With it an
// MFMA fragment along the PIXEL axis comes straight out of the [pixel][channel] image the operands have in memory: no register transpose.
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
template <typename WT>
__device__ __forceinline__ typename WFrag<WT>::type tr_frag(const WT* lo_rows, int hi_off) {
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    const v4i16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)lo_rows);
    const v4i16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(lo_rows + hi_off));
    const v8i16 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(typename WFrag<WT>::type, v);
}
// 16-byte chunk swizzle of row `row` of a [pixel][channel] image with `pitch_bytes` per row: the four rows a 16-lane group reads land in
// four different 64-byte bank quadrants
__device__ __forceinline__ int tr_swz(int row, int pitch_bytes) {
    return pitch_bytes % 256 == 0 ? (row & 3) << 2 : pitch_bytes % 128 == 0 ? ((row >> 1) & 1) << 2 : 0;
}

// Workgroup -> (x, y, z) of the launch grid such that the pixel split z is the SLOWEST index of what one XCD runs: the hardware hands
// consecutive workgroups to the 8 XCDs round-robin, so with the plain blockIdx decode every XCD saw every pixel split and each operand
// slice crossed HBM once per tile of the OTHER operand (stage-6 expand: 226 MB for 33 MB of operands).  Here XCD k owns the k-th
// eighth of the split-major order: a split's slices of both operands (a few MB) are fetched into that XCD's L2 once and re-read there.
__device__ __forceinline__ void xcd_major_block(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
    const int L = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
    const int q = total >> 3, r = total & 7, xcd = L & 7, k = L >> 3;
    const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    bz = Lp / (gx * gy);
    const int rem = Lp - bz * gx * gy;
    by = rem / gx;
    bx = rem - by * gx;
}

struct WgradP {
    const void* x; const void* dz; const float* se; float* part;
    int B, H, W, Ho, Wo, Cin, CinT, cin_off, Cout, CoutT, cout_off, KS, stride, pad;
    long P;          // B * Ho * Wo
    long chunk;      // pixels per split (multiple of BK)
    int se_epi;      // the SE gate multiplies the partial tile's columns (every split lies inside one image) instead of every staged element
};

template <typename WT> __device__ __forceinline__ WT narrow(float v);
template <> __device__ __forceinline__ float narrow<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 narrow<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ _Float16 narrow<_Float16>(float v) { return (_Float16)f16_sat(v); }

// 4 consecutive channels c..c+3 of row `row` (channel stride CT, offset off), zero beyond C; T = fp32 or a 16-bit copy
template <typename T>
__device__ __forceinline__ f32x4 load_ch4(const T* base, long row, int CT, int off, int c, int C, bool vec_ok) {
    const T* q = base + row * CT + off + c;
    if (vec_ok && c + 3 < C) return load4<T>(q);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c + e < C) v[e] = to_f32<T>(q[e]);
    return v;
}

// component j of 8 staged pixels -> 8 consecutive K elements of one LDS row (16-byte stores)
__device__ __forceinline__ void put8(float* dst, const f32x4 (&r)[8], int j) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{r[0][j], r[1][j], r[2][j], r[3][j]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{r[4][j], r[5][j], r[6][j], r[7][j]};
}
__device__ __forceinline__ void put8(__bf16* dst, const f32x4 (&r)[8], int j) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)r[i][j];
    *reinterpret_cast<bf16x8*>(dst) = v;
}
__device__ __forceinline__ void put8(_Float16* dst, const f32x4 (&r)[8], int j) {
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)f16_sat(r[i][j]);
    *reinterpret_cast<f16x8*>(dst) = v;
}

// One staging task of an operand stored as T: CH consecutive channels (one 16-byte access: 4 fp32 | 8 halves) of 8 consecutive pixels,
// transposed in registers into CH pixel-contiguous 16-byte LDS rows.  16-bit copies (T == WT) move without a conversion.
template <typename WT, typename T>
struct Stager {
    static constexpr int CH = sizeof(T) == 2 ? 8 : 4;
    u32x4 r[8];
    __device__ __forceinline__ void zero(int i) { r[i] = u32x4{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ void load(int i, const T* base, long row, int CT, int off, int c, int C, bool vec_ok) {
        const T* q = base + row * CT + off + c;
        if (vec_ok && c + CH - 1 < C) { r[i] = *reinterpret_cast<const u32x4*>(q); return; }
        if constexpr (sizeof(T) == 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < C) v[e] = q[e];
            r[i] = __builtin_bit_cast(u32x4, v);
        } else {
            typedef __attribute__((ext_vector_type(8))) T t8;
            t8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = c + e < C ? q[e] : (T)0.0f;
            r[i] = __builtin_bit_cast(u32x4, v);
        }
    }
    // r[i][e] *= se[e] (fp32 gate per channel, zero beyond C)
    __device__ __forceinline__ void scale(int i, const float* se, int c, int C) {
        if constexpr (sizeof(T) == 4) {
            f32x4 v = __builtin_bit_cast(f32x4, r[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= c + e < C ? se[e] : 0.f;
            r[i] = __builtin_bit_cast(u32x4, v);
        } else {
            typedef __attribute__((ext_vector_type(8))) T t8;
            t8 v = __builtin_bit_cast(t8, r[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = from_f32<T>((float)v[e] * (c + e < C ? se[e] : 0.f));
            r[i] = __builtin_bit_cast(u32x4, v);
        }
    }
    // channel j of the 8 staged pixels -> 8 consecutive K elements of one LDS row
    __device__ __forceinline__ void put(WT* dst, int j) const {
        if constexpr (sizeof(T) == 4) {
            typedef __attribute__((ext_vector_type(8))) WT w8;
            if constexpr (sizeof(WT) == 4) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{__uint_as_float(r[0][j]), __uint_as_float(r[1][j]), __uint_as_float(r[2][j]), __uint_as_float(r[3][j])};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{__uint_as_float(r[4][j]), __uint_as_float(r[5][j]), __uint_as_float(r[6][j]), __uint_as_float(r[7][j])};
            } else {
                w8 v;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = narrow<WT>(__uint_as_float(r[i][j]));
                *reinterpret_cast<w8*>(dst) = v;
            }
        } else {
            static_assert(sizeof(T) != 2 || sizeof(WT) == 2, "16-bit copies come in the compute type");
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t lo = (j & 1) ? (r[2 * w][j >> 1] >> 16) : (r[2 * w][j >> 1] & 0xffffu);
                const uint32_t hi = (j & 1) ? (r[2 * w + 1][j >> 1] & 0xffff0000u) : (r[2 * w + 1][j >> 1] << 16);
                o[w] = lo | hi;
            }
            *reinterpret_cast<u32x4*>(dst) = o;
        }
    }
};

// XT / DT: storage types of the layer input and of the output gradient (fp32, or the 16-bit copies the BatchNorm passes write)
template <typename WT, typename XT, typename DT, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradP p) {
    constexpr int BK = sizeof(WT) == 4 ? 32 : 64;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LD = BK + 16 / (int)sizeof(WT);             // row stride in elements: 144 bytes -> conflict-free ds_read_b128 fragments
    constexpr int OCT = BK / 8;                                // pixel octets per K step
    using SA = Stager<WT, DT>;
    using SB = Stager<WT, XT>;
    constexpr int CHA = SA::CH, CHB = SB::CH;
    // Both operands are 16-bit copies in the compute type: the LDS images stay [pixel][channel] (a staging task = eight plain 16-byte
    // stores, no v_perm transposes -- they cost as many VALU cycles as the step's MFMAs), and the fragments are transpose reads.
    constexpr bool TR = sizeof(WT) == 2 && sizeof(XT) == 2 && sizeof(DT) == 2;
    constexpr int NTA = (BM / CHA) * OCT, NTB = (BN / CHB) * OCT;       // staging tasks per operand (<= 256 each: launch_wgrad picks the tile accordingly)
    __shared__ __attribute__((aligned(16))) WT As[BM * LD];
    __shared__ __attribute__((aligned(16))) WT Bs[BN * LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int KK = p.KS * p.KS;
    int bx, by, split;
    xcd_major_block(bx, by, split);
    const int mt = bx, nt = by / KK, tap = by - nt * KK;
    const int tr = tap / p.KS, ts = tap - tr * p.KS;
    const long k0 = (long)split * p.chunk, k1 = k0 + p.chunk < p.P ? k0 + p.chunk : p.P;
    const bool a_vec = (p.CoutT % CHA == 0) && (p.cout_off % CHA == 0);
    const bool b_vec = (p.CinT % CHB == 0) && (p.cin_off % CHB == 0);
    const bool plain = p.KS == 1 && p.stride == 1;
    const int HoWo = p.Ho * p.Wo;
    // staging tasks: (channel group, pixel octet); the output-gradient tasks fill the threads from 0 up, the input tasks from 128 up, so that
    // with 16-bit copies (128 tasks each at 128 x 128) every thread has exactly one
    const int tb = (t + 128) & 255;
    const bool a_on = t < NTA, b_on = tb < NTB;
    const int a_cq = t % (BM / CHA), a_po = t / (BM / CHA);
    const int b_cq = tb % (BN / CHB), b_po = tb / (BN / CHB);
    const int m0 = mt * BM + a_cq * CHA, n0 = nt * BN + b_cq * CHB;
    SA sa;
    SB sb;

    auto fetch = [&](long kb) {
        if (a_on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long pix = kb + a_po * 8 + i;
                if (pix < k1 && m0 < p.Cout) sa.load(i, static_cast<const DT*>(p.dz), pix, p.CoutT, p.cout_off, m0, p.Cout, a_vec);
                else sa.zero(i);
            }
        }
        if (b_on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long pix = kb + b_po * 8 + i;
                sb.zero(i);
                if (pix < k1 && n0 < p.Cin) {
                    long row;
                    int b;
                    bool ok = true;
                    if (plain) {
                        row = pix;
                        b = (int)(pix / HoWo);
                    } else {
                        b = (int)(pix / HoWo);
                        const int rem = (int)(pix - (long)b * HoWo);
                        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                        const int iy = oy * p.stride + tr - p.pad, ix = ox * p.stride + ts - p.pad;
                        ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                        row = ((long)b * p.H + iy) * p.W + ix;
                    }
                    if (ok) {
                        sb.load(i, static_cast<const XT*>(p.x), row, p.CinT, p.cin_off, n0, p.Cin, b_vec);
                        if (p.se && !p.se_epi) sb.scale(i, p.se + (long)b * p.Cin + n0, n0, p.Cin);
                    }
                }
            }
        }
    };
    auto stage = [&]() {
        if constexpr (TR) {
            if (a_on) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = a_po * 8 + i;
                    *reinterpret_cast<u32x4*>(As + row * BM + ((a_cq ^ tr_swz(row, BM * 2)) << 3)) = sa.r[i];
                }
            }
            if (b_on) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = b_po * 8 + i;
                    *reinterpret_cast<u32x4*>(Bs + row * BN + ((b_cq ^ tr_swz(row, BN * 2)) << 3)) = sb.r[i];
                }
            }
            return;
        }
        if (a_on) {
#pragma unroll
            for (int j = 0; j < CHA; ++j) sa.put(As + (a_cq * CHA + j) * LD + a_po * 8, j);
        }
        if (b_on) {
#pragma unroll
            for (int j = 0; j < CHB; ++j) sb.put(Bs + (b_cq * CHB + j) * LD + b_po * 8, j);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fk = (lane >> 5) * 8;
    // transpose-read addresses (elements): lane = 16 * g + 4 * j + q reads row (g >> 1) * 8 + j (+ 4 for the upper half of the fragment),
    // columns (g & 1) * 16 + 4 * q .. + 3 of its 32-channel block
    int tra[TM], trb[TN];
    if constexpr (TR) {
        const int g = lane >> 4, rowl = (g >> 1) * 8 + ((lane >> 2) & 3), coll = (g & 1) * 16 + (lane & 3) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int col = (wm * TM + i) * 32 + coll;
            tra[i] = rowl * BM + (((col >> 3) ^ tr_swz(rowl, BM * 2)) << 3) + (col & 7);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = (wn * TN + j) * 32 + coll;
            trb[j] = rowl * BN + (((col >> 3) ^ tr_swz(rowl, BN * 2)) << 3) + (col & 7);
        }
    }
    // (a second staging set so that the loads run two K steps ahead was measured: 242 + 64 registers = one wave per SIMD instead of two,
    //  1x1 layers 15.1 -> 22.9 ms per step)
    if (k0 < k1) fetch(k0);
    for (long kb = k0; kb < k1; kb += BK) {
        stage();
        __syncthreads();
        if (kb + BK < k1) fetch(kb + BK);                       // in flight behind the MFMAs of this step
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if constexpr (sizeof(WT) == 4) {
                f32x4 af[TM][2], bf[TN][2];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float* q = As + ((wm * TM + i) * 32 + fr) * LD + kk * 16 + fk;
                    af[i][0] = *reinterpret_cast<const f32x4*>(q);
                    af[i][1] = *reinterpret_cast<const f32x4*>(q + 4);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float* q = Bs + ((wn * TN + j) * 32 + fr) * LD + kk * 16 + fk;
                    bf[j][0] = *reinterpret_cast<const f32x4*>(q);
                    bf[j][1] = *reinterpret_cast<const f32x4*>(q + 4);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e >> 2][e & 3], bf[j][e >> 2][e & 3], acc[i][j], 0, 0, 0);
            } else {
                using F = typename WFrag<WT>::type;
                F af[TM], bf[TN];
                if constexpr (TR) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = tr_frag<WT>(As + tra[i] + kk * 16 * BM, 4 * BM);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[j] = tr_frag<WT>(Bs + trb[j] + kk * 16 * BN, 4 * BN);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const F*>(As + ((wm * TM + i) * 32 + fr) * LD + kk * 16 + fk);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const F*>(Bs + ((wn * TN + j) * 32 + fr) * LD + kk * 16 + fk);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (sizeof(WT) == 2 && __is_same(WT, __bf16)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
    // partial tile -> part[((split * KK + tap) * Cout + m) * Cin + n]; accumulator element e of a lane: row 8*(e/4) + 4*(lane/32) + e%4, column lane%32
    float* pp = p.part + ((long)split * KK + tap) * p.Cout * p.Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nt * BN + (wn * TN + j) * 32 + (lane & 31);
            if (n >= p.Cin) continue;
            // sum_p dz[p][m] * (x[p][n] * g[b][n]) = g[b][n] * sum_p dz[p][m] * x[p][n] when all of this split's pixels are in image b
            const float gate = p.se_epi ? p.se[(k0 / HoWo) * p.Cin + n] : 1.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mt * BM + (wm * TM + i) * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
                if (m < p.Cout) pp[(long)m * p.Cin + n] = acc[i][j][e] * gate;
            }
        }
}


// ------------------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 weight gradient with the NINE TAPS IN ONE WORKGROUP (16-bit copies of both operands, Wo % 32 == 0).
// The generic kernel above gives every (Cout tile, Cin tile, tap) its own workgroup, so both operand streams are re-read per tap and per
// tile of the other operand: the last FPN level moved 4.7 GB per launch and ran at the speed of those re-reads.  Here a workgroup owns a
// 64 x 64 (Cout x Cin) tile for all nine taps: a K step is 32 consecutive output pixels of one image row; the output-gradient tile is
// staged ONCE and its fragments feed nine MFMAs each, the activation rows y-1, y, y+1 are staged as nine pre-shifted 64 x 32 tiles (one
// per tap: a shift by one pixel is 2 bytes, which a 16-byte fragment read cannot absorb; the shifted loads of a step hit L1).
// Each wave keeps the nine taps' accumulators of its 32 x 32 sub-tile (144 registers).  Same partial-sum layout and reduction.
// ------------------------------------------------------------------------------------------------------------------------
template <typename WT>
__global__ __launch_bounds__(256) void wgrad3_kernel(WgradP p, int nseg, long steps_per_split) {
    constexpr int LD = 40;                                    // 32 pixels + 8: 80-byte rows, conflict-free ds_read_b128 fragments
    using ST = Stager<WT, WT>;
    // tile 0 = output gradient; tiles 1 + 3 * slot + sx = activation row `slot` (a ring of three input rows) shifted by sx - 1 pixels
    __shared__ __attribute__((aligned(16))) WT Ls[10 * 64 * LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int mt, nt, split;
    xcd_major_block(mt, nt, split);
    // K steps walk DOWN an image column segment (step = (image, x segment) * Ho + y): consecutive steps share two of their three input
    // rows, which stay in the LDS ring -- only row y + 1 is staged per step (3 shifted tiles + the gradient tile = 128 tasks instead of 320)
    const long total = (long)p.B * nseg * p.Ho;
    const long s0 = (long)split * steps_per_split, s1 = s0 + steps_per_split < total ? s0 + steps_per_split : total;
    const bool a_vec = (p.CoutT % 8 == 0) && (p.cout_off % 8 == 0), b_vec = (p.CinT % 8 == 0) && (p.cin_off % 8 == 0);
    ST st[2];
    auto decode = [&](long step, int& b, int& x0, int& y) {
        const long col = step / p.Ho;
        y = (int)(step - col * p.Ho);
        const int seg = (int)(col % nseg);
        b = (int)(col / nseg);
        x0 = seg * 32;
    };
    // task q of a step: q < 32 the gradient tile; then activation tiles -- all nine (full: first step of a split or of a column) or the three of row y + 1
    auto task = [&](int q, bool full, int y, int& tile, int& iy, int& sx) {
        if (q < 32) { tile = 0; iy = 0; sx = 0; return; }
        const int k = (q - 32) >> 5;
        const int r = full ? k / 3 : 2;
        sx = full ? k - r * 3 : k;
        iy = y + r - 1;
        tile = 1 + ((iy + 3) % 3) * 3 + sx;
    };
    auto fetch = [&](long step, bool full) {
        int b, x0, y;
        decode(step, b, x0, y);
        const int ntask = full ? 320 : 128;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = t + 256 * u;
            if (q >= ntask) break;
            int tile, iy, sx;
            task(q, full, y, tile, iy, sx);
            const int w = q & 31, cq = w & 7, po = w >> 3;
            if (tile == 0) {
                const int m0 = mt * 64 + cq * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const long row = ((long)b * p.Ho + y) * p.Wo + x0 + po * 8 + i;
                    if (m0 < p.Cout) st[u].load(i, static_cast<const WT*>(p.dz), row, p.CoutT, p.cout_off, m0, p.Cout, a_vec);
                    else st[u].zero(i);
                }
            } else {
                const int n0 = nt * 64 + cq * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ix = x0 + po * 8 + i + sx - 1;
                    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && n0 < p.Cin)
                        st[u].load(i, static_cast<const WT*>(p.x), ((long)b * p.H + iy) * p.W + ix, p.CinT, p.cin_off, n0, p.Cin, b_vec);
                    else st[u].zero(i);
                }
            }
        }
    };
    auto stage = [&](long step, bool full) {
        int b, x0, y;
        decode(step, b, x0, y);
        const int ntask = full ? 320 : 128;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = t + 256 * u;
            if (q >= ntask) break;
            int tile, iy, sx;
            task(q, full, y, tile, iy, sx);
            const int w = q & 31, cq = w & 7, po = w >> 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) st[u].put(Ls + (tile * 64 + cq * 8 + j) * LD + po * 8, j);
        }
    };
    auto is_full = [&](long step) { return step == s0 || (step % p.Ho) == 0; };
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    using F = typename WFrag<WT>::type;
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    if (s0 < s1) fetch(s0, true);
    for (long step = s0; step < s1; ++step) {
        stage(step, is_full(step));
        __syncthreads();
        if (step + 1 < s1) fetch(step + 1, is_full(step + 1));  // in flight behind the 18 MFMAs of this step
        const int y = (int)(step % p.Ho);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const F af = *reinterpret_cast<const F*>(Ls + (wm * 32 + fr) * LD + kk * 16 + fk);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int slot = (y + r + 2) % 3;                   // row y + r - 1
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const F bf = *reinterpret_cast<const F*>(Ls + ((1 + slot * 3 + sx) * 64 + wn * 32 + fr) * LD + kk * 16 + fk);
                    if constexpr (__is_same(WT, __bf16)) acc[r * 3 + sx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[r * 3 + sx], 0, 0, 0);
                    else acc[r * 3 + sx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[r * 3 + sx], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    const int n = nt * 64 + wn * 32 + (lane & 31);
    if (n >= p.Cin) return;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float* pp = p.part + ((long)split * 9 + k) * p.Cout * p.Cin;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = mt * 64 + wm * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
            if (m < p.Cout) pp[(long)m * p.Cin + n] = acc[k][e];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The nine-tap kernel on TRANSPOSE READS (everything 8-channel aligned): the LDS images are [pixel][64 channels] exactly as in memory,
// so (1) a tap's one-pixel shift is a ROW offset of the fragment read -- no pre-shifted copies: a step stages ONE activation row of 34
// pixels and one gradient row of 32 (528 plain 16-byte stores per workgroup instead of 128 tasks of 8 loads + 64 v_perm + 8 stores),
// (2) the staging registers of a step are two or three 16-byte values per thread, so the loads run TWO steps ahead of their use.
// Ring of four activation rows (three live, one being filled) and two gradient rows: one barrier per step.
// ------------------------------------------------------------------------------------------------------------------------
template <typename WT>
__global__ __launch_bounds__(256) void wgrad3t_kernel(WgradP p, int nseg, long steps_per_split) {
    constexpr int XROW = 34 * 64, DROW = 32 * 64;              // elements per ring row
    __shared__ __attribute__((aligned(16))) WT XR[4 * XROW];
    __shared__ __attribute__((aligned(16))) WT DR[2 * DROW];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int mt, nt, split;
    xcd_major_block(mt, nt, split);
    const long total = (long)p.B * nseg * p.Ho;
    const long s0 = (long)split * steps_per_split, s1 = s0 + steps_per_split < total ? s0 + steps_per_split : total;
    auto decode = [&](long step, int& b, int& x0, int& y) {
        const long col = step / p.Ho;
        y = (int)(step - col * p.Ho);
        const int seg = (int)(col % nseg);
        b = (int)(col / nseg);
        x0 = seg * 32;
    };
    // item i of a row set: i < 272 activation (pixel i / 8 of 34, 16-byte chunk i % 8), then 256 of the gradient row
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto load_x = [&](int b, int iy, int x0, int i) -> u32x4 {
        const int px = i >> 3, ch = (i & 7) * 8, ix = x0 - 1 + px, n = nt * 64 + ch;
        if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W || n >= p.Cin) return zero4;
        return *reinterpret_cast<const u32x4*>(static_cast<const WT*>(p.x) + (((long)b * p.H + iy) * p.W + ix) * p.CinT + p.cin_off + n);
    };
    auto load_d = [&](int b, int y, int x0, int i) -> u32x4 {
        const int px = i >> 3, ch = (i & 7) * 8, m = mt * 64 + ch;
        if (m >= p.Cout) return zero4;
        return *reinterpret_cast<const u32x4*>(static_cast<const WT*>(p.dz) + (((long)b * p.Ho + y) * p.Wo + x0 + px) * p.CoutT + p.cout_off + m);
    };
    auto put = [&](WT* row, int i, u32x4 v) {                   // item i -> pixel i / 8, chunk (i % 8) ^ swizzle(pixel)
        const int px = i >> 3;
        *reinterpret_cast<u32x4*>(row + px * 64 + (((i & 7) ^ tr_swz(px, 128)) << 3)) = v;
    };
    // the row set of step s = activation row y + 1 and gradient row y of its column
    struct Set { u32x4 v[3]; };
    auto fetch = [&](long step, Set& st) {
        int b, x0, y;
        decode(step, b, x0, y);
        st.v[0] = load_x(b, y + 1, x0, t);
        st.v[1] = t < 16 ? load_x(b, y + 1, x0, 256 + t) : load_d(b, y, x0, t - 16);
        st.v[2] = t < 16 ? load_d(b, y, x0, 240 + t) : zero4;
    };
    auto store = [&](long step, const Set& st) {
        const int y = (int)(step % p.Ho);
        WT* xr = XR + ((y + 1) & 3) * XROW;
        WT* dr = DR + (y & 1) * DROW;
        put(xr, t, st.v[0]);
        if (t < 16) { put(xr, 256 + t, st.v[1]); put(dr, 240 + t, st.v[2]); }
        else put(dr, t - 16, st.v[1]);
    };
    // first step of a split or of a column: rows y - 1 and y are not in the ring yet
    auto fill = [&](long step) {
        int b, x0, y;
        decode(step, b, x0, y);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int iy = y - 1 + r;
            WT* xr = XR + (iy & 3) * XROW;
            put(xr, t, load_x(b, iy, x0, t));
            if (t < 16) put(xr, 256 + t, load_x(b, iy, x0, 256 + t));
        }
    };
    auto is_full = [&](long step) { return step == s0 || (step % p.Ho) == 0; };

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    // transpose-read addresses: lane = 16 g + 4 j + q reads pixel row (g >> 1) * 8 + j (+ 4: upper half), channels (g & 1) * 16 + 4 q .. + 3 of
    // its 32-channel block; the activation fragment of tap column sx starts sx pixel rows lower, which moves the row's swizzle with it
    const int g = lane >> 4, rowl = (g >> 1) * 8 + ((lane >> 2) & 3), coll = (g & 1) * 16 + (lane & 3) * 4;
    const int cola = wm * 32 + coll, colb = wn * 32 + coll;
    const int tra = rowl * 64 + (((cola >> 3) ^ tr_swz(rowl, 128)) << 3) + (cola & 7);
    int trb[3];
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) trb[sx] = (rowl + sx) * 64 + (((colb >> 3) ^ tr_swz(rowl + sx, 128)) << 3) + (colb & 7);

    Set sa, sb;                                                  // sets of steps s and s + 1
    if (s0 < s1) fetch(s0, sa);
    if (s0 + 1 < s1) fetch(s0 + 1, sb);
    for (long step = s0; step < s1; ++step) {
        if (is_full(step)) {
            __syncthreads();                                     // (the previous column's last step is still being read)
            fill(step);
        }
        store(step, sa);
        __syncthreads();
        sa = sb;
        if (step + 2 < s1) fetch(step + 2, sb);                  // two steps (36 MFMAs per wave) ahead of its use
        const int y = (int)(step % p.Ho);
        const WT* dr = DR + (y & 1) * DROW;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const auto af = tr_frag<WT>(dr + tra + kk * 16 * 64, 4 * 64);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const WT* xr = XR + ((y + r - 1) & 3) * XROW;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const auto bf = tr_frag<WT>(xr + trb[sx] + kk * 16 * 64, 4 * 64);
                    if constexpr (__is_same(WT, __bf16)) acc[r * 3 + sx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[r * 3 + sx], 0, 0, 0);
                    else acc[r * 3 + sx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[r * 3 + sx], 0, 0, 0);
                }
            }
        }
    }
    const int n = nt * 64 + wn * 32 + (lane & 31);
    if (n >= p.Cin) return;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float* pp = p.part + ((long)split * 9 + k) * p.Cout * p.Cin;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = mt * 64 + wm * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
            if (m < p.Cout) pp[(long)m * p.Cin + n] = acc[k][e];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// 1x1 stride-1 weight gradient, both operands 16-bit and 8-channel aligned (every MBConv expand / project convolution): DMA ring.
// The generic kernel spends ~4.5 us per K step of 64 pixels on 0.3 us of MFMAs: its operands travel through 208 staging / address
// registers (two waves per SIMD) one step ahead.  With transpose reads the LDS image IS the memory layout, so the tiles go
// HBM -> LDS by buffer_load ... lds: no staging registers, three stages of 32 pixels in flight per workgroup, a 128 x 128 tile
// (Cout x Cin), four waves of 64 x 64.  LDS slot (row, c') of a stage holds global chunk c' ^ swizzle(row) (the DMA writes lanes
// linearly, so the bank swizzle is applied on the source side).  The SE gate, if any, multiplies the partial tile's columns (se_epi).
// ------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_w;
template <int N> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename WT>
__global__ __launch_bounds__(256, 2) void wgrad1t_kernel(WgradP p) {
    constexpr int BK = 32, BM = 128, BN = 128, NST = 4, D = NST - 1;
    constexpr int STAGE = BK * (BM + BN);                       // elements per stage: [32][128] gradient rows, then [32][128] activation rows
    constexpr int OOBW = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    WT* const ring = reinterpret_cast<WT*>(wg_smem);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int mt, nt, split;
    xcd_major_block(mt, nt, split);
    const long k0 = (long)split * p.chunk, k1 = k0 + p.chunk < p.P ? k0 + p.chunk : p.P;
    const int nk = k1 > k0 ? (int)((k1 - k0 + BK - 1) / BK) : 0;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dz), 0, (unsigned)(p.P * p.CoutT * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (unsigned)(p.P * p.CinT * 2), 0x00020000);
    // DMA d (0, 1) of an operand: this wave's 64 lanes fill rows (d * 4 + wave) * 4 .. + 3, 16 chunks each
    int rowd[2], offa[2], offb[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int row = (d * 4 + wave) * 4 + (lane >> 4), c = (lane & 15) ^ tr_swz(row, 256);
        rowd[d] = row;
        const int m = mt * BM + c * 8, n = nt * BN + c * 8;
        offa[d] = m < p.Cout ? (p.cout_off + m) * 2 : OOBW;      // (+ pixel * CoutT * 2 per stage)
        offb[d] = n < p.Cin ? (p.cin_off + n) * 2 : OOBW;
    }
    int ld = 0;
    auto issue = [&](int slot) {
        const long pix0 = k0 + (long)ld * BK;
        WT* st = ring + slot * STAGE;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const long pix = pix0 + rowd[d];
            const bool ok = pix < k1;
            const int va = (ok && offa[d] != OOBW) ? (int)(pix * p.CoutT * 2) + offa[d] : OOBW;
            const int vb = (ok && offb[d] != OOBW) ? (int)(pix * p.CinT * 2) + offb[d] : OOBW;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_w*)(st + (d * 4 + wave) * 4 * BM), 16, va, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_w*)(st + BK * BM + (d * 4 + wave) * 4 * BN), 16, vb, 0, 0, 0);
        }
        ++ld;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int g = lane >> 4, rowl = (g >> 1) * 8 + ((lane >> 2) & 3), coll = (g & 1) * 16 + (lane & 3) * 4;
    int tra[2], trb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ca = (wm * 2 + i) * 32 + coll, cb = (wn * 2 + i) * 32 + coll;
        tra[i] = rowl * BM + (((ca >> 3) ^ tr_swz(rowl, 256)) << 3) + (ca & 7);
        trb[i] = BK * BM + rowl * BN + (((cb >> 3) ^ tr_swz(rowl, 256)) << 3) + (cb & 7);
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (j < nk) issue(j);
    int cur = 0, nxt = D;
    for (int it = 0; it < nk; ++it) {
        const int later = nk - 1 - it < D - 1 ? nk - 1 - it : D - 1;     // stages requested after this one: 4 DMAs each
        if (later == 0) wg_wait_vmcnt<0>(); else if (later == 1) wg_wait_vmcnt<4>(); else wg_wait_vmcnt<8>();
        __builtin_amdgcn_s_barrier();                                    // everyone's pieces of stage `it` are in; slot `nxt` was read in step it - 1
        if (it + D < nk) issue(nxt);
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
        const WT* st = ring + cur * STAGE;
        cur = cur + 1 == NST ? 0 : cur + 1;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            typename WFrag<WT>::type af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = tr_frag<WT>(st + tra[i] + kk * 16 * BM, 4 * BM);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = tr_frag<WT>(st + trb[j] + kk * 16 * BN, 4 * BN);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (__is_same(WT, __bf16)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    float* pp = p.part + (long)split * p.Cout * p.Cin;
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nt * BN + (wn * 2 + j) * 32 + (lane & 31);
            if (n >= p.Cin) continue;
            const float gate = p.se_epi ? p.se[(k0 / HoWo) * p.Cin + n] : 1.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mt * BM + (wm * 2 + i) * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
                if (m < p.Cout) pp[(long)m * p.Cin + n] = acc[i][j][e] * gate;
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------
// THIN weight gradient: a 3x3 (or 1x1) stride-1 convolution with ONE or TWO output channels (the map heads' top convolutions: 192 -> 1 | 2).
// On the matrix cores 31 of a tile's 32 rows were padding and each of the nine taps re-read the activation: 550 us per head at 1.8 TFLOP/s
// for what is a weighted column sum.  Here every INPUT pixel q is read once: lane = (pixel slot, CH-channel group) keeps
// acc[co][tap][CH] += dz[q - tap offset][co] * x[q][ch] in registers (the nine dz values of a pixel are broadcast loads), the slots of a
// workgroup meet in LDS in a fixed order, and the workgroup writes one partial block in the layout wgrad_reduce_kernel sums.
// ------------------------------------------------------------------------------------------------------------------------
template <typename XT, int CO, int KS>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(WgradP p) {
    constexpr int CH = 16 / (int)sizeof(XT);
    __shared__ float red[256 * CH];
    const int t = threadIdx.x;
    const int NQ = p.Cin / CH;                                 // lanes per pixel (Cin % CH == 0, NQ <= 256: launch_wgrad checks)
    const int slots = 256 / NQ;
    const int slot = t / NQ, cq = t - slot * NQ;
    constexpr int KK = KS * KS, pad = (KS - 1) / 2;
    const long k0 = (long)blockIdx.x * p.chunk, k1 = k0 + p.chunk < p.P ? k0 + p.chunk : p.P;      // (P = B*H*W: stride 1, same padding)
    float acc[CO][KK][CH];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int k = 0; k < KK; ++k)
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[c][k][e] = 0.f;
    const XT* xb = static_cast<const XT*>(p.x);
    const float* dzb = static_cast<const float*>(p.dz);
    if (slot < slots) {
        // U pixels per trip: their activation loads and 9 * CO gradient loads are all issued before the first product (one pixel per trip
        // made this a chain of ten dependent round trips per pixel: 390 us per head)
        constexpr int U = 4;
        for (long q0 = k0 + slot; q0 < k1; q0 += (long)slots * U) {
            float xv[U][CH], dv[U][CO][KK];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long q = q0 + (long)u * slots;
                const bool qok = q < k1;
                const long qq = qok ? q : k0;
                const int b = (int)(qq / ((long)p.H * p.W));
                const int rem = (int)(qq - (long)b * p.H * p.W);
                const int iy = rem / p.W, ix = rem - iy * p.W;
                load16<XT>(xb + qq * p.CinT + p.cin_off + cq * CH, xv[u]);
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const int r = k / KS, s2 = k - r * KS;
                    const int oy = iy - r + pad, ox = ix - s2 + pad;             // the output pixel whose tap (r, s2) lands on q
                    const bool ok = qok && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
                    const float* dq = dzb + (((long)b * p.Ho + (ok ? oy : 0)) * p.Wo + (ok ? ox : 0)) * p.CoutT + p.cout_off;
#pragma unroll
                    for (int c = 0; c < CO; ++c) dv[u][c][k] = (ok && c < p.Cout) ? dq[c] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < KK; ++k)
#pragma unroll
                    for (int c = 0; c < CO; ++c)
#pragma unroll
                        for (int e = 0; e < CH; ++e) acc[c][k][e] = fmaf(dv[u][c][k], xv[u][e], acc[c][k][e]);
        }
    }
    float* pp = p.part + (long)blockIdx.x * KK * p.Cout * p.Cin;
#pragma unroll
    for (int c = 0; c < CO; ++c)
        for (int k = 0; k < KK; ++k) {
            if (c >= p.Cout) break;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                float v = 0.f;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
                    if (kk == k) v = acc[c][kk][e];
                red[t * CH + e] = slot < slots ? v : 0.f;
            }
            __syncthreads();
            for (int n = t; n < p.Cin; n += 256) {
                float v = 0.f;
                for (int sl = 0; sl < slots; ++sl) v += red[(sl * NQ + n / CH) * CH + n % CH];
                pp[((long)k * p.Cout + c) * p.Cin + n] = v;
            }
        }
}

inline bool wgrad_thin_shape(int Cout, int Cin, int ksize, int stride) { return Cout <= 2 && stride == 1 && (ksize == 3 || ksize == 1) && Cin % 8 == 0 && Cin <= 1024; }

inline bool wgrad3_shape(int Wo, int Cout, int Cin, int ksize) { return ksize == 3 && Wo % 32 == 0 && Cout >= 32 && Cin >= 32; }

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int S, int KK, int Cout, int Cin) {
    const long per = (long)KK * Cout * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += part[(long)k * per + i];
        const int ci = (int)(i % Cin);
        const long t = i / Cin;
        const int co = (int)(t % Cout), tap = (int)(t / Cout);
        out[((long)co * Cin + ci) * KK + tap] += s;
    }
}
// many splits of a small gradient (the thin kernel: 288 splits x 1728 weights): 16 elements x 16 split lanes per workgroup, lane k sums
// splits k, k + 16, ... and the 16 lanes meet in LDS in a fixed order (one thread per element walked 288 dependent loads)
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ part, float* __restrict__ out, int S, int KK, int Cout, int Cin) {
    __shared__ float red[16][16];
    const long per = (long)KK * Cout * Cin;
    const int el = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + el;
    float s = 0.f;
    if (i < per)
        for (int k = kl; k < S; k += 16) s += part[(long)k * per + i];
    red[kl][el] = s;
    __syncthreads();
    if (kl != 0 || i >= per) return;
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][el];
    const int ci = (int)(i % Cin);
    const long t = i / Cin;
    const int co = (int)(t % Cout), tap = (int)(t / Cout);
    out[((long)co * Cin + ci) * KK + tap] += s;
}

inline hipError_t launch_wgrad_reduce(const void* part, float* out, int S, int KK, int Cout, int Cin, hipStream_t s) {
    const long per = (long)KK * Cout * Cin;
    if (S >= 48) {
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((per + 15) / 16)), dim3(256), 0, s, (const float*)part, out, S, KK, Cout, Cin);
    } else {
        const long nb = (per + 255) / 256;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, s, (const float*)part, out, S, KK, Cout, Cin);
    }
    return hipGetLastError();
}

struct WgCfg { int id, BM, BN; };
inline WgCfg wgrad_cfg(int Cout, int Cin, int ksize) {
    if (Cout <= 32) return {0, 32, 128};
    if (Cout <= 64 || Cin <= 64) return {1, 64, 64};
    // every (Cout tile, Cin tile, tap) workgroup re-reads its two operand streams: the kernel is bound by those re-reads (last FPN level:
    // 4.7 GB per launch at 128 x 128), so wide layers take the 192 x 256 tile (12 accumulators per wave, one workgroup per CU)
    if (ksize == 3 && Cout >= 192 && Cin >= 256) return {3, 192, 256};        // (1x1 layers: measured slower, 25.5 -> 29.6 ms per step)
    return {2, 128, 128};
}

template <typename WT, typename XT, typename DT>
void launch_cfg(const WgradP& p, int cfg, dim3 grid, hipStream_t s) {
    if (cfg == 0) hipLaunchKernelGGL((wgrad_kernel<WT, XT, DT, 1, 1, 1, 4>), grid, dim3(256), 0, s, p);
    else if (cfg == 1) hipLaunchKernelGGL((wgrad_kernel<WT, XT, DT, 1, 1, 2, 2>), grid, dim3(256), 0, s, p);
    else if (cfg == 3) hipLaunchKernelGGL((wgrad_kernel<WT, XT, DT, 3, 4, 2, 2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<WT, XT, DT, 2, 2, 2, 2>), grid, dim3(256), 0, s, p);
}
// 16-bit compute type: every combination of fp32 / 16-bit storage of the two operands
template <typename WT>
void launch_io(const WgradP& p, int cfg, dim3 grid, bool x16, bool d16, hipStream_t s) {
    if (x16 && d16) launch_cfg<WT, WT, WT>(p, cfg, grid, s);
    else if (x16) launch_cfg<WT, WT, float>(p, cfg, grid, s);
    else if (d16) launch_cfg<WT, float, WT>(p, cfg, grid, s);
    else launch_cfg<WT, float, float>(p, cfg, grid, s);
}

}  // namespace

int ftc_wgrad_splits_impl(int B, int Ho, int Wo, int Cout, int Cin, int ksize) {
    const WgCfg c = wgrad_cfg(Cout, Cin, ksize);
    long tiles = (long)((Cout + c.BM - 1) / c.BM) * ((Cin + c.BN - 1) / c.BN) * ksize * ksize;
    if (wgrad3_shape(Wo, Cout, Cin, ksize)) tiles = (long)((Cout + 63) / 64) * ((Cin + 63) / 64);      // the nine-tap kernel: 64 x 64 tiles, taps inside
    const long P = (long)B * Ho * Wo;
    if (wgrad_thin_shape(Cout, Cin, ksize, 1)) {                  // (the thin kernel: one workgroup per ~1024 pixels, at most 512)
        const long st = (P + 1023) / 1024;
        return (int)(st < 1 ? 1 : st > 512 ? 512 : st);
    }
    long S = (1536 + tiles - 1) / tiles;                          // ~6 workgroups per CU in flight
    // ... but every split writes (and the reduction re-reads) a full fp32 copy of the gradient: on the deep stages (4608 pixels, 1.5 M
    // weights) 16 splits moved 200 MB of partial sums around 33 MB of operands.  At least 1024 pixels (16 K steps) per split.
    const long smax = (P + 1023) / 1024;
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    if (S > 4096) S = 4096;
    return (int)S;
}

hipError_t launch_wgrad(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    WgradP p;
    p.x = a.in; p.dz = a.in2; p.se = (o.flags & FTC_FLAG_SE_SCALE) ? a.scale : nullptr; p.part = a.aux;
    p.B = o.B; p.H = o.H; p.W = o.W; p.Ho = o.Ho; p.Wo = o.Wo;
    p.Cin = o.Cin; p.CinT = o.Cin_total > 0 ? o.Cin_total : o.Cin; p.cin_off = o.cin_off;
    p.Cout = o.Cout; p.CoutT = o.Cout_total > 0 ? o.Cout_total : o.Cout; p.cout_off = o.cout_off;
    p.KS = o.ksize; p.stride = o.stride; p.pad = (o.ksize - 1) / 2;
    p.P = (long)o.B * o.Ho * o.Wo;
    const int S = o.aux0 > 0 ? o.aux0 : 1;
    const int BK = o.w_dtype == FTC_F32 ? 32 : 64;
    p.chunk = ((p.P + S - 1) / S + BK - 1) / BK * BK;
    const long HoWo = (long)o.Ho * o.Wo;
    p.se_epi = (p.se && o.ksize == 1 && o.stride == 1 && HoWo % p.chunk == 0) ? 1 : 0;
    const int KK = o.ksize * o.ksize;
    if (wgrad_thin_shape(o.Cout, o.Cin, o.ksize, o.stride) && o.res_dtype == FTC_F32 && !p.se && o.H == o.Ho && o.W == o.Wo && !(o.flags & 0x100) &&
        p.CinT % 8 == 0 && p.cin_off % 8 == 0) {
        p.chunk = (p.P + S - 1) / S;
#define THIN(XT) do { if (o.ksize == 3) { if (o.Cout == 1) hipLaunchKernelGGL((wgrad_thin_kernel<XT, 1, 3>), dim3(S), dim3(256), 0, s, p); \
                                           else hipLaunchKernelGGL((wgrad_thin_kernel<XT, 2, 3>), dim3(S), dim3(256), 0, s, p); } \
                      else { if (o.Cout == 1) hipLaunchKernelGGL((wgrad_thin_kernel<XT, 1, 1>), dim3(S), dim3(256), 0, s, p); \
                             else hipLaunchKernelGGL((wgrad_thin_kernel<XT, 2, 1>), dim3(S), dim3(256), 0, s, p); } } while (0)
        if (o.in_dtype == FTC_F32) THIN(float); else if (o.in_dtype == FTC_F16) THIN(_Float16); else THIN(__bf16);
#undef THIN
        hipError_t et = hipGetLastError();
        if (et != hipSuccess) return et;
        return launch_wgrad_reduce(a.aux, (float*)a.out, S, KK, o.Cout, o.Cin, s);
    }
    if (wgrad3_shape(o.Wo, o.Cout, o.Cin, o.ksize) && o.stride == 1 && ftc_is16(o.w_dtype) && o.in_dtype == o.w_dtype && o.res_dtype == o.w_dtype && !p.se &&
        !(o.flags & 0x100)) {                                    // (0x100: the generic kernel, for A/B measurements)
        const int nseg = o.Wo / 32;
        const long total = (long)o.B * nseg * o.Ho;
        const long sps = (total + S - 1) / S;
        const dim3 g3((o.Cout + 63) / 64, (o.Cin + 63) / 64, S);
        // everything 8-channel (16-byte) aligned: the transpose-read kernel; 0x200 forces the staged one (A/B measurements, tests)
        const bool al8 = o.Cout % 8 == 0 && o.Cin % 8 == 0 && p.CoutT % 8 == 0 && p.CinT % 8 == 0 && p.cout_off % 8 == 0 && p.cin_off % 8 == 0 && !(o.flags & 0x200);
        if (al8 && o.w_dtype == FTC_F16) hipLaunchKernelGGL(wgrad3t_kernel<_Float16>, g3, dim3(256), 0, s, p, nseg, sps);
        else if (al8) hipLaunchKernelGGL(wgrad3t_kernel<__bf16>, g3, dim3(256), 0, s, p, nseg, sps);
        else if (o.w_dtype == FTC_F16) hipLaunchKernelGGL(wgrad3_kernel<_Float16>, g3, dim3(256), 0, s, p, nseg, sps);
        else hipLaunchKernelGGL(wgrad3_kernel<__bf16>, g3, dim3(256), 0, s, p, nseg, sps);
        hipError_t e3 = hipGetLastError();
        if (e3 != hipSuccess) return e3;
        return launch_wgrad_reduce(a.aux, (float*)a.out, S, KK, o.Cout, o.Cin, s);
    }
    // 1x1 stride 1, 16-bit copies of both operands, 8-channel aligned, no per-element gate: the DMA-ring kernel (0x200: the generic one)
    if (o.ksize == 1 && o.stride == 1 && ftc_is16(o.w_dtype) && o.in_dtype == o.w_dtype && o.res_dtype == o.w_dtype && (!p.se || p.se_epi) && !(o.flags & 0x300) &&
        o.Cout % 8 == 0 && o.Cin % 8 == 0 && p.CoutT % 8 == 0 && p.CinT % 8 == 0 && p.cout_off % 8 == 0 && p.cin_off % 8 == 0 && o.Cout >= 64 && o.Cin >= 64 &&
        p.P * p.CoutT * 2 < 0x7fffffffL && p.P * p.CinT * 2 < 0x7fffffffL && p.chunk % 32 == 0) {
        const dim3 g1((o.Cout + 127) / 128, (o.Cin + 127) / 128, S);
        const size_t lds = (size_t)4 * 32 * 256 * 2;
        if (o.w_dtype == FTC_F16) {
            static bool set16 = false;
            if (!set16) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1t_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set16 = true; }
            hipLaunchKernelGGL(wgrad1t_kernel<_Float16>, g1, dim3(256), lds, s, p);
        } else {
            static bool setb = false;
            if (!setb) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1t_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); setb = true; }
            hipLaunchKernelGGL(wgrad1t_kernel<__bf16>, g1, dim3(256), lds, s, p);
        }
        hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return e1;
        return launch_wgrad_reduce(a.aux, (float*)a.out, S, 1, o.Cout, o.Cin, s);
    }
    WgCfg c = wgrad_cfg(o.Cout, o.Cin, o.ksize);
    // the 192 x 256 tile has one staging task per thread only with 16-byte accesses of 8 channels (16-bit copies) or the fp32 K step of 32
    if (c.id == 3 && ftc_is16(o.w_dtype) && !(o.in_dtype == o.w_dtype && o.res_dtype == o.w_dtype)) c = WgCfg{2, 128, 128};
    const dim3 grid((o.Cout + c.BM - 1) / c.BM, ((o.Cin + c.BN - 1) / c.BN) * KK, S);
    // in_dtype / res_dtype: storage of the layer input / of the output gradient (fp32, or a 16-bit copy in the compute type)
    if (o.w_dtype == FTC_F32) launch_cfg<float, float, float>(p, c.id, grid, s);
    else if (o.w_dtype == FTC_F16) launch_io<_Float16>(p, c.id, grid, o.in_dtype == FTC_F16, o.res_dtype == FTC_F16, s);
    else launch_io<__bf16>(p, c.id, grid, o.in_dtype == FTC_BF16, o.res_dtype == FTC_BF16, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_wgrad_reduce(a.aux, (float*)a.out, S, KK, o.Cout, o.Cin, s);
}
