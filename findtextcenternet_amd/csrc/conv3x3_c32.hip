// 3x3 stride-1 convolution 32 -> 32 channels with EVERYTHING resident: the Fused-MBConv blocks of stage 1 (expand_ratio 1: one 3x3 convolution + BN + SiLU +
// residual per block, /root/reference/models/detector.py:14; 384x384 maps at 768x768, 4 blocks).
//
// Why its own kernel (round 6).  The implicit-GEMM kernel runs this shape on 32x256 tiles with K = 288 in nine steps of 32, each step a global -> register ->
// LDS round trip that feeds FOUR MFMAs per wave: profiles/r05e: 131 us per block for 453 MB (3.45 TB/s), MFMA busy 6.7 %, 64 % of the wave cycles parked,
// 32.6 % of the LDS cycles lost to write conflicts of its 80-byte padded rows.  But the whole problem of a 16x16-pixel output tile fits a corner of the LDS:
// the 18x18x32 halo is 20.7 KB, ALL the weights (32 x 288) are 18.4 KB.  So: one round of DMA (halo + weights, every piece in flight at once, out-of-image
// pixels answered with zeros by the buffer unit), one barrier, 36 MFMAs per wave with nothing to wait for, the LDS-staged coalesced epilogue of the other
// kernels (bias, SiLU, fp32 residual, fp32 trunk + 16-bit copy).  39 KB of LDS: four workgroups per CU hide each other's single load latency.
// Both LDS images are unpadded 64-byte rows with the 16-byte chunks XOR-swizzled by (row >> 2) & 3: the 16 lanes the LDS serves together in a
// ds_read_b128 -- 16 consecutive halo pixels, or 16 weight rows -- differ in (row & 3, (row >> 2) & 3) and land in 16 different bank quads; DMA writes are
// whole 1 KiB pieces.  No conflicts on either side (the verdict's "stage-1 LDS conflicts" item).
// Same K order as the implicit-GEMM kernel (tap-major, then the two 16-channel groups): bit-identical results.
#include "conv_igemm_impl.h"

namespace convimpl {

constexpr int C32_NT = 256, C32_HALO = 324, C32_HB = C32_HALO * 64, C32_WB = 9 * 32 * 64;      // threads; halo pixels; bytes of the two images
constexpr int C32_LDS = C32_HB + C32_WB;                                                         // 39,168 B (the epilogue image + bias rows alias it: 34.8 KB)

template <typename T>
__global__ __launch_bounds__(C32_NT, 4) void conv3x3_c32_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tilesX = (p.W + 15) / 16, tilesY = (p.H + 15) / 16;
    const int img = bid / (tilesX * tilesY);
    const int sp = bid - img * tilesX * tilesY;
    const int ty0 = (sp / tilesX) * 16, tx0 = (sp % tilesX) * 16;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);

    // ---- one round of DMA: 1296 halo chunks + 1152 weight chunks of 16 bytes, 64 per wave-instruction ----
    constexpr int HCH = C32_HALO * 4, WCH = 9 * 32 * 4;
#pragma unroll
    for (int i = 0; i < (HCH + C32_NT - 1) / C32_NT; ++i) {
        const int q = i * C32_NT + t;
        if (i * C32_NT + wave * 64 < HCH) {                                         // wave-uniform
            const int hr = q >> 2, kc = (q & 3) ^ ((hr >> 2) & 3);
            const int iy = ty0 - 1 + hr / 18, ix = tx0 - 1 + hr % 18;
            const bool ok = q < HCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int off = ok ? (((img * p.H + iy) * p.W + ix) * 32 + kc * 8) * 2 : OOB;
            if (q < HCH) glds16(rin, (lds_void_t*)(smem_raw + (i * C32_NT + wave * 64) * 16), off, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < (WCH + C32_NT - 1) / C32_NT; ++i) {
        const int q = i * C32_NT + t;
        if (i * C32_NT + wave * 64 < WCH) {
            const int row = q >> 2, kc = (q & 3) ^ ((row >> 2) & 3);                // row = tap * 32 + n
            const int tap = row >> 5, n = row & 31;
            const int off = ((n * 9 + tap) * 32 + kc * 8) * 2;                      // weights [Cout][9][Cin]
            if (q < WCH) glds16(rw, (lds_void_t*)(smem_raw + C32_HB + (i * C32_NT + wave * 64) * 16), off, 0);
        }
    }

    f32x16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.0f;
    // a lane's pixel inside a 32-pixel sub-tile (2 tile rows x 16): every 16-lane group of a ds_read_b128 gets ONE tile row (see conv3x3_halo_kernel)
    const int lpix = ((__builtin_popcount(l31 >> 2) & 1) << 4) | ((l31 >> 3) << 2) | (l31 & 3);
    int hr0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) hr0[j] = (wave * 4 + j * 2 + (lpix >> 4)) * 18 + (lpix & 15);
    wait_vmcnt<0>();
    __syncthreads();

    using FragT = typename Frag<T>::type;
    const unsigned char* const wbase = smem_raw + C32_HB + l31 * 64;
    const int fa = (l31 >> 2) & 3;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int d = (tap / 3) * 18 + tap % 3;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const FragT a = *reinterpret_cast<const FragT*>(wbase + tap * 2048 + (((g * 2 + half) ^ fa) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hr = hr0[j] + d;
                const FragT b = *reinterpret_cast<const FragT*>(smem_raw + hr * 64 + (((g * 2 + half) ^ ((hr >> 2) & 3)) << 4));
                acc[0][j] = mfma16(a, b, acc[0][j]);
            }
        }
    }
    conv_epilogue_lds<T, float, 1, 2, C32_NT, 32, 256>(p, acc, smem_raw, 0, 0, wave * 64, half, lpix, [&](int row) {
        const int oy = ty0 + (row >> 4), ox = tx0 + (row & 15);
        return (oy < p.H && ox < p.W) ? (img * p.H + oy) * p.W + ox : -1;
    });
}

template <typename T>
static hipError_t launch_c32_t(ConvP p, hipStream_t s) {
    auto kern = conv3x3_c32_kernel<T>;
    p.nN = 1;
    p.nblk = p.B * ((p.H + 15) / 16) * ((p.W + 15) / 16);
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(C32_NT), C32_LDS, s, p);
    return hipGetLastError();
}

// 16-bit operands of one type, fp32 output (+ optional 16-bit NHWC copy), 3x3 stride 1 "same", exactly 32 -> 32 whole-tensor channels, one group, no SE scale /
// per-image weights / border bias / fused forms; fp32 residual
bool conv3x3_c32_legal(const ftc_op& o) {
    static const bool off = [] { const char* e = std::getenv("FTC_NO_C32"); return e && *e && *e != '0'; }();
    if (off) return false;
    return ftc_is16(o.w_dtype) && o.in_dtype == o.w_dtype && o.out_dtype == FTC_F32 && o.ksize == 3 && o.stride == 1 && o.Ho == o.H && o.Wo == o.W && o.Cin == 32 &&
           o.Cin_total == 32 && o.cin_off == 0 && o.Cout == 32 && o.Cout_total == 32 && o.cout_off == 0 && o.groups <= 1 && (o.flags & ~FTC_FLAG_RESIDUAL) == 0 &&
           (!(o.flags & FTC_FLAG_RESIDUAL) || o.res_dtype == FTC_F32);
}

hipError_t launch_conv3x3_c32(const ConvP& p, const ftc_op& o, hipStream_t s) {
    return o.w_dtype == FTC_F16 ? launch_c32_t<_Float16>(p, s) : launch_c32_t<__bf16>(p, s);
}

}  // namespace convimpl
