// MBConv head fused: expand 1x1 convolution (+BN+SiLU) -> depthwise 3x3 stride 1 (+BN+SiLU) -> SE squeeze sums, in ONE kernel for
// the low-resolution stages (FTC_OP_DWCONV + FTC_FLAG_EXPAND_IN; reference: torchvision MBConv block[0..1], models/detector.py:17-20).
//
// Why: at batch 8 the MBConv chain is bound by the memory system, not by the matrix pipes or by launch latency,
// and the expanded tensor (28-35 MB per block at 24x24) crosses HBM four times: written by the expand conv, read and re-written by the
// depthwise conv, read by the project conv.  Here it never leaves the CU between the first two.
//
// Work split: a workgroup (8 waves) owns one image's row band [y0, y0 + TY) (TY = 12 of the 24 rows at 24x24: the band plus one halo
// row above and below, full width, is at most 352 pixels) and 128 expanded channels.
//   1. GEMM  e[pixel][128] = W_e[128][Cin] . x[pixel][Cin] over the band + halo rows: rows of x and of W_e stream through a 2-stage
//      direct-to-LDS ring in K steps of 32 (64-byte rows, XOR-swizzled as in conv_igemm_glds_kernel); wave w multiplies channel blocks
//      {2(w&1), 2(w&1)+1} x pixel blocks {w>>1, (w>>1)+4, (w>>1)+8}: 5 fragment reads per 6 MFMAs.
//   2. bias + SiLU, rounded to the 16-bit type exactly like the expand conv's epilogue, into an LDS image [pixel][128] (256-byte rows,
//      16-byte chunks XOR-swizzled by pixel so the MFMA layout's 16-lane store groups hit 32 distinct banks).
//   3. depthwise 3x3 out of that image: lane = 4 channels x 6 vertically adjacent outputs of one column (8 input rows x 3 columns
//      = 24 LDS reads for 6 outputs), the same accumulation order as dwconv_strip_kernel -- the result is bit-identical to the
//      two-kernel path -- bias + SiLU, 8-byte stores (32 lanes = one pixel's 256 bytes), channel sums for the SE squeeze.
// The halo rows are recomputed by the neighbouring band (14/12 of the GEMM work); zero padding is by predication, never stored.
#include "conv_igemm_impl.h"

namespace {
using namespace convimpl;

struct MbP {
    const void* x;
    const void* we;
    const float* be;
    const float* wd;
    const float* bd;
    void* out;
    float* partial;
    int B, H, W, Cin, C;
    int TY, P, nchunk, nblk;
    unsigned x_bytes, we_bytes;
    unsigned long long* tl;      // flags 0x1000: s_memtime of wave 0 at the phase boundaries, 8 values per workgroup (tools/mbfused_bench.py)
};

constexpr int MB_CC = 128;                 // expanded channels per workgroup
constexpr int MB_NPX = 352;                // pixels (band + halo rows) per workgroup, 11 MFMA pixel blocks
constexpr int MB_XS = MB_NPX * 64;         // bytes of the x part of a stage (64-byte rows: K step 32)
constexpr int MB_STAGE = MB_XS + MB_CC * 64;
constexpr int MB_EOFF = 2 * MB_STAGE;      // LDS offset of the expanded image
constexpr int MB_LDS = MB_EOFF + MB_NPX * 256;

template <typename T>
__global__ __launch_bounds__(512) void mb_expand_dw_kernel(const MbP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 512, NL = 4, G = 2;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int cc = bid % p.nchunk;
    const int bt = bid / p.nchunk;
    const int tile = bt % p.P, b = bt / p.P;
    const int y0 = tile * p.TY;
    const int y1 = min(p.H, y0 + p.TY);                    // output rows [y0, y1)
    const int ylo = max(0, y0 - 1), yhi = min(p.H - 1, y1);   // expanded rows held: [ylo, yhi]
    const int npx = (yhi - ylo + 1) * p.W;
    const int npb = (npx + 31) >> 5;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwe = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.we), 0, p.we_bytes, 0x00020000);

    // DMA slot q = i*512 + t of a stage -> LDS byte q*16: pixel rows first (4 chunks each), then the 128 weight rows
    int s_off[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int q = i * NT + t;
        if (q < MB_NPX * 4) {
            const int row = q >> 2, kc = (q & 3) ^ ((row >> 2) & 3);
            s_off[i] = row < npx ? (((b * p.H + ylo) * p.W + row) * p.Cin + kc * 8) * 2 : OOB;
        } else {
            const int row = (q - MB_NPX * 4) >> 2, kc = (q & 3) ^ ((row >> 2) & 3);
            s_off[i] = row < MB_CC ? ((cc * MB_CC + row) * p.Cin + kc * 8) * 2 : OOB;
        }
    }
    auto issue = [&](int step, int st) {
        const int soff = step * 64;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q0 = i * NT + wave * 64;                          // wave-uniform
            if (q0 < MB_NPX * 4 + MB_CC * 4) {
                lds_void_t* dst = (lds_void_t*)(smem_raw + st * MB_STAGE + q0 * 16);
                glds16(q0 < MB_NPX * 4 ? rx : rwe, dst, s_off[i], soff);
            }
        }
    };

    const int chp = wave & 1, pq = wave >> 1;
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    using FragT = typename Frag<T>::type;
    int offA[2][G], offB[3][G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (chp * 2 + i) * 32 + l31;
            offA[i][g] = MB_XS + row * 64 + (((g * 2 + half) ^ ((row >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int row = (pq + 4 * j) * 32 + l31;
            offB[j][g] = row * 64 + (((g * 2 + half) ^ ((row >> 2) & 3)) << 4);
        }
    }
    const int nk = p.Cin >> 5;
    const bool tl_on = p.tl && t == 0;
    unsigned long long* tl = p.tl + (size_t)blockIdx.x * 8;
    if (tl_on) tl[0] = __builtin_amdgcn_s_memtime();
    issue(0, 0);
    for (int it = 0; it < nk; ++it) {
        wait_vmcnt<0>();
        wg_barrier();
        if (it + 1 < nk) issue(it + 1, (it + 1) & 1);
        const unsigned char* base = smem_raw + (it & 1) * MB_STAGE;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            FragT af[2], bf[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const FragT*>(base + offA[i][g]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (pq + 4 * j < npb) bf[j] = *reinterpret_cast<const FragT*>(base + offB[j][g]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (pq + 4 * j < npb) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
                }
        }
    }

    if (tl_on) tl[1] = __builtin_amdgcn_s_memtime();
    // ---- expanded image: bias + SiLU, 16-bit, [pixel][128] with chunk c16 of pixel m at ((c16 ^ (m & 15)) << 4) ----
    unsigned char* eimg = smem_raw + MB_EOFF;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int m = (pq + 4 * j) * 32 + l31;
        if (pq + 4 * j < npb && m < npx) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = (chp * 2 + i) * 32 + 8 * q + 4 * half;          // channel inside the chunk of 128
                    f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    v += *reinterpret_cast<const f32x4*>(p.be + cc * MB_CC + cl);
                    v = apply_act4<true>(v, FTC_ACT_SILU);
                    store4<T>(reinterpret_cast<T*>(eimg + m * 256 + ((((cl >> 3) ^ (m & 15))) << 4)) + (cl & 7), v);
                }
        }
    }
    __syncthreads();
    if (tl_on) tl[2] = __builtin_amdgcn_s_memtime();

    // ---- depthwise 3x3 + bias + SiLU + channel sums ----
    const int cq = t & 31, pl = t >> 5;                     // 4 channels cq*4.., pixel lane 0..15
    const int c = cc * MB_CC + cq * 4;
    f32x4 wv[9], bv;
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(p.wd + (long)k * p.C + c);
    bv = *reinterpret_cast<const f32x4*>(p.bd + c);
    constexpr int R = 6;
    const int nsr = (y1 - y0 + R - 1) / R;
    const int nstrips = nsr * p.W;
    const int csub = (cq & 1) * 8;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    T* outp = reinterpret_cast<T*>(p.out);
    for (int s = pl; s < nstrips; s += 16) {
        const int sr = s / p.W, x = s - sr * p.W;
        const int oy0 = y0 + sr * R;
        f32x4 a[R];
#pragma unroll
        for (int oo = 0; oo < R; ++oo) a[oo] = bv;
#pragma unroll
        for (int r = 0; r < R + 2; ++r) {
            const int yy = oy0 - 1 + r;
            f32x4 xin[3];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const int xx = x - 1 + s2;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (yy >= ylo && yy <= yhi && (unsigned)xx < (unsigned)p.W) {
                    const int m = (yy - ylo) * p.W + xx;
                    z = load4<T>(reinterpret_cast<const T*>(eimg + m * 256 + ((((cq >> 1) ^ (m & 15))) << 4) + csub));
                }
                xin[s2] = z;
            }
#pragma unroll
            for (int oo = 0; oo < R; ++oo) {
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
        for (int oo = 0; oo < R; ++oo) {
            const int oy = oy0 + oo;
            if (oy < y1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[oo][e] = act_silu_fast(a[oo][e]);
                store4<T>(outp + (((long)b * p.H + oy) * p.W + x) * p.C + c, a[oo]);
                sum += a[oo];
            }
        }
    }
    if (tl_on) tl[3] = __builtin_amdgcn_s_memtime();
    float* red = reinterpret_cast<float*>(smem_raw);       // [16][128]: the operand ring is free
    *reinterpret_cast<f32x4*>(red + pl * MB_CC + cq * 4) = sum;
    __syncthreads();
    if (t < MB_CC) {
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s2 += red[k * MB_CC + t];
        p.partial[((long)b * p.P + tile) * p.C + cc * MB_CC + t] = s2;
    }
    if (tl_on) tl[4] = __builtin_amdgcn_s_memtime();
}

}  // namespace

hipError_t launch_mbfused(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    int P = 0;
    const int TY = ftc_mbfused_rows(o.H, o.W, &P);
    if (TY <= 0 || P != o.aux0 || o.Cin % MB_CC || o.Cin_total % 32 || o.stride != 1) return hipErrorInvalidValue;
    MbP p;
    p.x = a.in; p.we = a.w2; p.be = a.bias2; p.wd = static_cast<const float*>(a.w); p.bd = a.bias; p.out = a.out; p.partial = a.aux;
    p.B = o.B; p.H = o.H; p.W = o.W; p.Cin = o.Cin_total; p.C = o.Cin;
    p.TY = TY; p.P = P; p.nchunk = o.Cin / MB_CC; p.nblk = o.B * P * p.nchunk;
    p.x_bytes = (unsigned)((long)o.B * o.H * o.W * o.Cin_total * 2);
    p.we_bytes = (unsigned)((long)o.Cin * o.Cin_total * 2);
    p.tl = (o.flags & 0x1000) ? reinterpret_cast<unsigned long long*>(const_cast<void*>(a.in2)) : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mb_expand_dw_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, MB_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mb_expand_dw_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, MB_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (o.in_dtype == FTC_F16) hipLaunchKernelGGL(mb_expand_dw_kernel<_Float16>, dim3(p.nblk), dim3(512), MB_LDS, s, p);
    else hipLaunchKernelGGL(mb_expand_dw_kernel<__bf16>, dim3(p.nblk), dim3(512), MB_LDS, s, p);
    return hipGetLastError();
}
