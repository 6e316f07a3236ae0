// Fused-MBConv block with expansion (torchvision FusedMBConv, expand_ratio 4: stages 2-3 of EfficientNetV2-XL as instantiated by
// /root/reference/models/detector.py:14-16) in ONE kernel -- FTC_OP_FMBCONV:
//     e   = SiLU(BN(conv3x3(x)))          Cin -> E = 4 Cin       (block[0]; BN folded into weights / bias)
//     out = BN(conv1x1(e)) + x             E   -> Cout = Cin     (block[1] + the residual add)
//
// Why (round-5 verdict item 4, asked since round 2): as two launches the expanded tensor e makes a round trip through HBM -- at batch 8
// stage 2 writes and re-reads 151 MB per block, and the 1x1 projection that reads it is a pure HBM-bound pass (15 launches, 46-65 us at
// 5.0-5.2 TB/s: profiles/r05e_bf16_b8_kernel_stats.txt, `conv_igemm<..out=f32,tile=64x64>`).  Here a workgroup owns 128 output pixels and ALL E
// expanded channels of them: the 3x3 implicit GEMM leaves the E x 128 tile in the accumulators, bias + SiLU + the rounding to the 16-bit
// type happen in registers (the same rounding point as the two-launch form), the tile goes to LDS as the B operand of a second MFMA GEMM
// against the projection weights (read as fragments straight from L2: 32-74 KB, shared by every workgroup), and only the Cout-channel result
// (+ bias + residual) is written: fp32 trunk + its 16-bit copy.  e never exists in memory.
//
// Shape of the first GEMM: rows = E output channels (MFMA A operand = weights, K-major [E][9][Cin]), columns = 128 pixels (B operand =
// NHWC activations gathered per tap, out-of-image taps answered with zeros by the buffer unit), K = 9 Cin in steps of BK = 64 (Cin % 64 == 0)
// or 32.  8 waves = 4 channel quarters x 2 pixel halves, (E/128) x 2 accumulator tiles of 32x32 per wave; operands are staged global ->
// registers -> LDS (padded rows, conflict-free ds_read_b128): the loads of step k+1 are in flight under the MFMAs of step k.  The adopted form
// keeps ONE operand buffer (65 KB of LDS, 122 VGPRs): two workgroups share a CU and one's SiLU / projection phases run under the other's 3x3
// GEMM (the first version had two buffers and one barrier per step but only one workgroup per CU: no faster than the pair of launches it
// replaced; all forms measured: DESIGN.md appendix A9, profiles/r06_fmbconv_forms.txt).  E x 128 instead of the 128 x 128 tiles the tuner
// picks for the stand-alone 3x3: 0.75x the operand bytes per FLOP through L2 -> LDS.
// Second GEMM: rows = Cout channels, columns = the same 128 pixels, K = E; wave = (pixel block of 32) x (channel tiles t, t + 2).
//
// Numerics: identical rounding points AND K order to conv3x3 (+SiLU, 16-bit store) followed by conv1x1 (fp32 accumulate, + bias, + fp32 residual):
// measured bit-identical to the two-launch form in its default tile configurations (tests/test_gpu_ops.py::test_fused_mbconv_block_in_one_launch).
#include "conv_igemm_impl.h"
#include "ftc_host.h"

namespace convimpl {

struct FmbP {
    const void* x;        // [B][H][W][Cin] 16-bit
    const void* w1;       // [E][9][Cin] 16-bit
    const float* b1;      // [E]
    const void* w2;       // [Cout][E] 16-bit
    const float* b2;      // [Cout]
    const float* res;     // fp32 [M][Cout] or null
    float* out;           // fp32 [M][Cout]
    void* out2;           // 16-bit [M][Cout] or null
    unsigned x_bytes, w1_bytes, w2_bytes;
    int B, H, W, Cin, E, Cout, M, ncb, nk, nblk;
};

// Geometry.  The template keeps the parameters of the forms that were measured against each other (WM = pixel halves of the tile and waves along the pixel
// axis, NBUF operand buffers, SM pixel sub-tiles per wave: DESIGN.md appendix A9, profiles/r06_fmbconv_forms.txt); the library instantiates the adopted one --
// WM = 2 (128 pixels, 8 waves), ONE operand buffer, SM = 2: 65 KB of LDS and 122 VGPRs for E = 256, i.e. two workgroups per CU.
template <int BK, int SN, int WM, int NBUF_, int SM_ = 2> struct FmbGeom {
    static constexpr int E = SN * 128, SM = SM_, TM = 32 * SM_ * WM, NT = 256 * WM, NBUF = NBUF_;
    static constexpr int CPR = BK / 8, ROW = BK + 8;                    // 16-byte chunks per K row; padded LDS row (elements)
    static constexpr int RPP = NT / CPR;                                // rows one staging pass covers
    static constexpr int NA = E / RPP, NB = TM / RPP;
    static constexpr int BUF = (E + TM) * ROW * 2;                      // bytes per operand buffer
    static constexpr int PITCH = E * 2;                                 // bytes per pixel row of the activated tile (chunks XOR-swizzled by row & 15)
    static constexpr int TILE = TM * PITCH;
    static constexpr int LDS = (NBUF * BUF > TILE + E * 4 ? NBUF * BUF : TILE + E * 4);
    static_assert(E % RPP == 0 && TM % RPP == 0 && LDS <= 160 * 1024, "");
};

template <typename T, int BK, int SN, int WM, int NBUF, int SMT = 2>
__global__ __launch_bounds__(256 * WM, SMT == 4 ? 2 : WM == 1 ? 3 : (NBUF == 1 && SN == 2) ? 4 : 2) void fmbconv_fused_kernel(const FmbP p) {
    using GM = FmbGeom<BK, SN, WM, NBUF, SMT>;
    constexpr int E = GM::E, CPR = GM::CPR, ROW = GM::ROW, RPP = GM::RPP, NA = GM::NA, NB = GM::NB, BUF = GM::BUF, PITCH = GM::PITCH;
    constexpr int SM = SMT, TM = GM::TM, FMB_NT = GM::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* const lds = reinterpret_cast<T*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int half = lane >> 5, l31 = lane & 31;

    // XCD-aware remap (workgroup b runs on XCD b % 8): each XCD takes a contiguous run of pixel tiles -- the halo rows neighbouring tiles share
    // are fetched into ONE private L2
    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int m0 = bid * TM;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const int kc = t % CPR, row0 = t / CPR;
    const int HW = p.H * p.W;

    const int a_off0 = (row0 * 9 * p.Cin + kc * 8) * 2;               // staging pass i adds a wave-uniform i * RPP rows (scalar offset)
    const int a_pass = RPP * 9 * p.Cin * 2;
    int b_off[NB], b_mask[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int m = m0 + row0 + i * RPP;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int img = mm / HW, rem = mm - img * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        b_off[i] = (((img * p.H + oy - 1) * p.W + ox - 1) * p.Cin + kc * 8) * 2;
        int mask = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (ok && (unsigned)(oy - 1 + r) < (unsigned)p.H && (unsigned)(ox - 1 + s) < (unsigned)p.W) mask |= 1 << (r * 3 + s);
        b_mask[i] = mask;
    }

    u32x4 ra[NA], rb[NB];
    int ld_tap = 0, ld_cb = 0, ld_r = 0, ld_s = 0;                     // K position of the next tile to fetch (wave-uniform)
    auto gload = [&]() {
        const int c0 = ld_cb * BK;
        const int w_soff = (ld_tap * p.Cin + c0) * 2;
        const int in_toff = ((ld_r * p.W + ld_s) * p.Cin + c0) * 2;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload(rw, a_off0, w_soff + i * a_pass);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload(rin, ((b_mask[i] >> ld_tap) & 1) ? b_off[i] + in_toff : OOB, 0);
        if (++ld_cb == p.ncb) {
            ld_cb = 0;
            ++ld_tap;
            if (++ld_s == 3) { ld_s = 0; ++ld_r; }
        }
    };
    T* const wA = lds + row0 * ROW + kc * 8;
    T* const wB = lds + (E + row0) * ROW + kc * 8;
    auto lds_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(wA + buf * (BUF / 2) + i * RPP * ROW) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(wB + buf * (BUF / 2) + i * RPP * ROW) = rb[i];
    };

    f32x16 acc[SN][SM];
#pragma unroll
    for (int i = 0; i < SN; ++i)
#pragma unroll
        for (int j = 0; j < SM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    using FragT = typename Frag<T>::type;
    const T* const fA = lds + (wn * SN * 32 + l31) * ROW + half * 8;
    const T* const fB = lds + (E + wm * SM * 32 + l31) * ROW + half * 8;
    auto compute = [&](int buf) {
        const T* A = fA + buf * (BUF / 2);
        const T* Bm = fB + buf * (BUF / 2);
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
                for (int j = 0; j < SM; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
    };

    // ---- GEMM 1: the 3x3 convolution, two operand buffers, one barrier per K step ----
    gload();
    if constexpr (GM::NBUF == 2) {
        lds_write(0);
        __syncthreads();
        for (int it = 0; it < p.nk; ++it) {
            const int cur = it & 1;
            if (it + 1 < p.nk) gload();                     // in flight while this step is multiplied
            compute(cur);
            if (it + 1 < p.nk) lds_write(cur ^ 1);          // the other buffer: last read in step it - 1, behind the barrier that ended it
            __syncthreads();
        }
    } else {
        for (int it = 0; it < p.nk; ++it) {
            lds_write(0);
            __syncthreads();
            if (it + 1 < p.nk) gload();                     // in flight while this step is multiplied
            compute(0);
            __syncthreads();
        }
    }

    // ---- projection weights of this wave's output tiles: the first K groups are requested now, they arrive under the SiLU pass ----
    constexpr int NPB = TM / 32;                                        // pixel blocks of 32 (2 | 4)
    constexpr int NCG = (FMB_NT / 64) / NPB;                            // channel-tile groups of the waves (2; 1 in the 4-wave 128-pixel form)
    constexpr int NT2 = 4 / NCG;                                        // channel tiles per wave: wn2, wn2 + NCG, .. (those below Cout / 32)
    const int wm2 = wave % NPB, wn2 = wave / NPB;                       // pixel block; first channel tile
    const int nt2 = p.Cout >> 5;
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, p.w2_bytes, 0x00020000);
    constexpr int NG2 = E / 16, PF = 4;                                 // K groups of GEMM 2; groups per prefetch batch
    int w2off[NT2];
#pragma unroll
    for (int u = 0; u < NT2; ++u) {
        const int tile = wn2 + NCG * u;
        w2off[u] = tile < nt2 ? ((tile * 32 + l31) * E + half * 8) * 2 : OOB;
    }
    u32x4 wq[NT2][PF];
#pragma unroll
    for (int g = 0; g < PF; ++g)
#pragma unroll
        for (int u = 0; u < NT2; ++u) wq[u][g] = bload(rw2, w2off[u], g * 32);

    // ---- bias + SiLU + rounding to the 16-bit type -> the tile [128 px][E] in LDS (the operand buffers are dead: the K loop ended on a barrier) ----
    float* const lbias = reinterpret_cast<float*>(smem_raw + GM::TILE);
    for (int c = t; c < E / 4; c += FMB_NT) *reinterpret_cast<f32x4*>(lbias + 4 * c) = *reinterpret_cast<const f32x4*>(p.b1 + 4 * c);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int prow = wm * SM * 32 + j * 32 + l31;
        unsigned char* lrow = smem_raw + prow * PITCH;
#pragma unroll
        for (int i = 0; i < SN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * SN * 32 + i * 32 + 8 * q + 4 * half;
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(lbias + nl);
                v = act_silu_fast4(v);
                const int chunk = (nl >> 3) ^ (prow & 15);
                store4<T>(reinterpret_cast<T*>(lrow + chunk * 16) + (nl & 7), v);
            }
        }
    }
    __syncthreads();

    // ---- GEMM 2: out[Cout][128 px] = W2[Cout][E] . tile^T ----
    f32x16 acc2[NT2];
#pragma unroll
    for (int u = 0; u < NT2; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[u][e] = 0.0f;
    const int prow2 = wm2 * 32 + l31;
    const unsigned char* trow = smem_raw + prow2 * PITCH;
#pragma unroll
    for (int g0 = 0; g0 < NG2; g0 += PF) {
        u32x4 wn_[NT2][PF];
        if (g0 + PF < NG2) {
#pragma unroll
            for (int g = 0; g < PF; ++g)
#pragma unroll
                for (int u = 0; u < NT2; ++u) wn_[u][g] = bload(rw2, w2off[u], (g0 + PF + g) * 32);
        }
#pragma unroll
        for (int g = 0; g < PF; ++g) {
            const int chunk = (2 * (g0 + g) + half) ^ (prow2 & 15);
            const FragT bf = *reinterpret_cast<const FragT*>(trow + chunk * 16);
#pragma unroll
            for (int u = 0; u < NT2; ++u)
                if (u == 0 || wn2 + NCG * u < nt2) acc2[u] = mfma16(__builtin_bit_cast(FragT, wq[u][g]), bf, acc2[u]);      // (wave-uniform; tile 0 of a group beyond Cout multiplies zeros)
        }
        if (g0 + PF < NG2) {
#pragma unroll
            for (int g = 0; g < PF; ++g)
#pragma unroll
                for (int u = 0; u < NT2; ++u) wq[u][g] = wn_[u][g];
        }
    }

    // ---- bias + residual, fp32 trunk + 16-bit copy: a lane owns 4 consecutive channels of its pixel per register quad ----
    const int m = m0 + prow2;
    if (m < p.M) {
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            const int tile = wn2 + NCG * u;
            if (tile >= nt2) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tile * 32 + 8 * q + 4 * half;
                f32x4 v = {acc2[u][4 * q], acc2[u][4 * q + 1], acc2[u][4 * q + 2], acc2[u][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(p.b2 + n);
                if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.Cout + n);
                *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.Cout + n) = v;
                if (p.out2) store4<T>(reinterpret_cast<T*>(p.out2) + (size_t)m * p.Cout + n, v);
            }
        }
    }
}

template <typename T, int BK, int SN, int WM, int NBUF, int SMT = 2>
hipError_t launch_fmb(FmbP p, hipStream_t s) {
    using GM = FmbGeom<BK, SN, WM, NBUF, SMT>;
    auto kern = fmbconv_fused_kernel<T, BK, SN, WM, NBUF, SMT>;
    p.nblk = (p.M + GM::TM - 1) / GM::TM;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GM::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(GM::NT), GM::LDS, s, p);
    return hipGetLastError();
}


// ---- fp16x3 form (fp32 tensors, FTC_FLAG_SPLIT16: the contract-grade plan; stage 2 there is 7 x (388 + 104) us with a 302 MB fp32 expanded tensor making the
// round trip).  Same structure as the 8-wave form above, E = 256 only: K steps of 32 (128-byte rows of pre-split chunks: four fp32 values as [hi x4 | lo x4]
// IEEE halves), weights arrive pre-split from the blob, activation chunks are split on the way to LDS (as the register-staged x3 kernel does), a product =
// three fp16 MFMAs; exact SiLU; the activated tile goes to LDS as pre-split chunks (1 KiB rows, XOR-swizzled by row & 15: 128 KB -- one workgroup per CU) --
// the very values the stand-alone 1x1 forms when it stages the fp32 tensor the 3x3 wrote, so the result is bit-identical to the two-launch form.
struct FmbX3 {
    static constexpr int E = 256, SN = 2, SM = 2, TM = 128, NT = 512, BK = 32, CPR = 8, ROW = 36, RPP = NT / CPR, NA = E / RPP, NB = TM / RPP;
    static constexpr int BUF = (E + TM) * ROW * 4, PITCH = E * 4, TILE = TM * PITCH, LDS = TILE + E * 4;
    static_assert(BUF <= TILE && LDS <= 160 * 1024, "");
};

__global__ __launch_bounds__(FmbX3::NT, 2) void fmbconv_fused_x3_kernel(const FmbP p) {
    using GM = FmbX3;
    constexpr int E = GM::E, SN = GM::SN, SM = GM::SM, TM = GM::TM, NTH = GM::NT, BK = GM::BK, CPR = GM::CPR, ROW = GM::ROW, RPP = GM::RPP, NA = GM::NA, NB = GM::NB,
                  PITCH = GM::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const lds = reinterpret_cast<float*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    int bid = blockIdx.x;
    {
        const int q = p.nblk >> 3, r = p.nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int m0 = bid * TM;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const int kc = t % CPR, row0 = t / CPR;
    const int HW = p.H * p.W;
    const int a_off0 = (row0 * 9 * p.Cin + kc * 4) * 4;
    const int a_pass = RPP * 9 * p.Cin * 4;
    int b_off[NB], b_mask[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int m = m0 + row0 + i * RPP;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int img = mm / HW, rem = mm - img * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        b_off[i] = (((img * p.H + oy - 1) * p.W + ox - 1) * p.Cin + kc * 4) * 4;
        int mask = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (ok && (unsigned)(oy - 1 + r) < (unsigned)p.H && (unsigned)(ox - 1 + s) < (unsigned)p.W) mask |= 1 << (r * 3 + s);
        b_mask[i] = mask;
    }
    u32x4 ra[NA], rb[NB];
    int ld_tap = 0, ld_cb = 0, ld_r = 0, ld_s = 0;
    auto gload = [&]() {
        const int c0 = ld_cb * BK;
        const int w_soff = (ld_tap * p.Cin + c0) * 4;
        const int in_toff = ((ld_r * p.W + ld_s) * p.Cin + c0) * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload(rw, a_off0, w_soff + i * a_pass);                       // weights: pre-split in the blob
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload(rin, ((b_mask[i] >> ld_tap) & 1) ? b_off[i] + in_toff : OOB, 0);
        if (++ld_cb == p.ncb) {
            ld_cb = 0;
            ++ld_tap;
            if (++ld_s == 3) { ld_s = 0; ++ld_r; }
        }
    };
    float* const wA = lds + row0 * ROW + kc * 4;
    float* const wB = lds + (E + row0) * ROW + kc * 4;
    // TWO operand buffers (2 x 55 KB: inside the 129 KB the activated tile needs anyway, so they cost no occupancy): the loads of step k + 1 are issued before
    // the MFMAs of step k, their LDS writes go to the other buffer after them, one barrier per step
    constexpr int BUFE = GM::BUF / 4;                                    // floats per buffer
    static_assert(2 * GM::BUF <= GM::LDS, "");
    auto lds_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(wA + buf * BUFE + i * RPP * ROW) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(wB + buf * BUFE + i * RPP * ROW) = chunk_hl(__builtin_bit_cast(f32x4, rb[i]));      // activations: split once, on the way to LDS
    };
    f32x16 acc[SN][SM];
#pragma unroll
    for (int i = 0; i < SN; ++i)
#pragma unroll
        for (int j = 0; j < SM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    const float* const fA = lds + (wn * SN * 32 + l31) * ROW + half * 4;
    const float* const fB = lds + (E + wm * SM * 32 + l31) * ROW + half * 4;
    gload();
    lds_write(0);
    __syncthreads();
    for (int it = 0; it < p.nk; ++it) {
        const int cur = it & 1;
        if (it + 1 < p.nk) gload();
        const float* A = fA + cur * BUFE;
        const float* Bm = fB + cur * BUFE;
#pragma unroll
        for (int g = 0; g < BK / 8; g += 2) {
            f16x8 ah[SN], al[SN];
#pragma unroll
            for (int i = 0; i < SN; ++i)
                frag_hl(*reinterpret_cast<const f32x4*>(A + i * 32 * ROW + g * 8), *reinterpret_cast<const f32x4*>(A + i * 32 * ROW + g * 8 + 8), ah[i], al[i]);
#pragma unroll
            for (int j = 0; j < SM; ++j) {
                f16x8 bh, bl;
                frag_hl(*reinterpret_cast<const f32x4*>(Bm + j * 32 * ROW + g * 8), *reinterpret_cast<const f32x4*>(Bm + j * 32 * ROW + g * 8 + 8), bh, bl);
#pragma unroll
                for (int i = 0; i < SN; ++i) acc[i][j] = mfma_split(ah[i], al[i], bh, bl, acc[i][j]);
            }
        }
        if (it + 1 < p.nk) lds_write(cur ^ 1);                            // last read in step it - 1, behind the barrier that ended it
        __syncthreads();
    }

    // ---- projection weights (pre-split): K block b of 16 = the chunks 4 b + half and 4 b + 2 + half of a row ----
    const int wm2 = wave & 3, wn2 = wave >> 2;
    const int nt2 = p.Cout >> 5;
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, p.w2_bytes, 0x00020000);
    constexpr int NG2 = E / 16, PF = 4, NT2 = 2;
    int w2off[NT2];
#pragma unroll
    for (int u = 0; u < NT2; ++u) {
        const int tile = wn2 + 2 * u;
        w2off[u] = tile < nt2 ? (tile * 32 + l31) * E * 4 + half * 16 : OOB;
    }
    u32x4 wq[NT2][PF][2];
#pragma unroll
    for (int g = 0; g < PF; ++g)
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            wq[u][g][0] = bload(rw2, w2off[u], g * 64);
            wq[u][g][1] = bload(rw2, w2off[u], g * 64 + 32);
        }

    // ---- bias + exact SiLU -> the tile [128 px][E] in LDS as pre-split chunks ----
    float* const lbias = reinterpret_cast<float*>(smem_raw + GM::TILE);
    for (int c = t; c < E / 4; c += NTH) *reinterpret_cast<f32x4*>(lbias + 4 * c) = *reinterpret_cast<const f32x4*>(p.b1 + 4 * c);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        const int prow = wm * SM * 32 + j * 32 + l31;
        unsigned char* lrow = smem_raw + prow * PITCH;
#pragma unroll
        for (int i = 0; i < SN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * SN * 32 + i * 32 + 8 * q + 4 * half;
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(lbias + nl);
                v = apply_act4<false>(v, FTC_ACT_SILU);
                *reinterpret_cast<u32x4*>(lrow + (((nl >> 2) ^ (prow & 15)) << 4)) = chunk_hl(v);
            }
        }
    }
    __syncthreads();

    // ---- GEMM 2 ----
    f32x16 acc2[NT2];
#pragma unroll
    for (int u = 0; u < NT2; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[u][e] = 0.0f;
    const int prow2 = wm2 * 32 + l31;
    const unsigned char* trow = smem_raw + prow2 * PITCH;
    const int fx = prow2 & 15;
#pragma unroll
    for (int g0 = 0; g0 < NG2; g0 += PF) {
        u32x4 wn_[NT2][PF][2];
        if (g0 + PF < NG2) {
#pragma unroll
            for (int g = 0; g < PF; ++g)
#pragma unroll
                for (int u = 0; u < NT2; ++u) {
                    wn_[u][g][0] = bload(rw2, w2off[u], (g0 + PF + g) * 64);
                    wn_[u][g][1] = bload(rw2, w2off[u], (g0 + PF + g) * 64 + 32);
                }
        }
#pragma unroll
        for (int g = 0; g < PF; ++g) {
            const int b16 = g0 + g;
            f16x8 bh, bl;
            frag_hl(*reinterpret_cast<const f32x4*>(trow + (((4 * b16 + half) ^ fx) << 4)), *reinterpret_cast<const f32x4*>(trow + (((4 * b16 + 2 + half) ^ fx) << 4)), bh, bl);
#pragma unroll
            for (int u = 0; u < NT2; ++u)
                if (u == 0 || wn2 + 2 * u < nt2) {
                    f16x8 ah, al;
                    frag_hl(__builtin_bit_cast(f32x4, wq[u][g][0]), __builtin_bit_cast(f32x4, wq[u][g][1]), ah, al);
                    acc2[u] = mfma_split(ah, al, bh, bl, acc2[u]);
                }
        }
        if (g0 + PF < NG2) {
#pragma unroll
            for (int g = 0; g < PF; ++g)
#pragma unroll
                for (int u = 0; u < NT2; ++u) { wq[u][g][0] = wn_[u][g][0]; wq[u][g][1] = wn_[u][g][1]; }
        }
    }

    const int m = m0 + prow2;
    if (m < p.M) {
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            const int tile = wn2 + 2 * u;
            if (tile >= nt2) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tile * 32 + 8 * q + 4 * half;
                f32x4 v = {acc2[u][4 * q], acc2[u][4 * q + 1], acc2[u][4 * q + 2], acc2[u][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(p.b2 + n);
                if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.Cout + n);
                *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.Cout + n) = v;
                if (p.out2) *reinterpret_cast<u32x4*>(static_cast<char*>(p.out2) + ((size_t)m * p.Cout + n) * 4) = chunk_hl(v);      // the pre-split copy
            }
        }
    }
}

static hipError_t launch_fmb_x3(FmbP p, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fmbconv_fused_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FmbX3::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.nblk = (p.M + FmbX3::TM - 1) / FmbX3::TM;
    hipLaunchKernelGGL(fmbconv_fused_x3_kernel, dim3(p.nblk), dim3(FmbX3::NT), FmbX3::LDS, s, p);
    return hipGetLastError();
}

}  // namespace convimpl

// Shapes FTC_OP_FMBCONV accepts (the plan builder asks before it emits one): 16-bit operands of one type, fp32 output (+ optional 16-bit copy), 3x3
// stride 1 "same", Cin % 32 == 0, E = aux1 in {256, 384}, Cout % 32 == 0 and <= 128; every byte offset inside one 2 GiB buffer resource.
bool ftc_fmbconv_legal(const ftc_op& o) {
    const int E = o.aux1;
    const long px = (long)o.B * o.H * o.W;
    if (o.w_dtype == FTC_F32)               // the fp16x3 form: fp32 tensors with FTC_FLAG_SPLIT16, E = 256, out2 = the optional PRE-SPLIT copy
        return (o.flags & FTC_FLAG_SPLIT16) && o.in_dtype == FTC_F32 && o.out_dtype == FTC_F32 && o.ksize == 3 && o.stride == 1 && o.Ho == o.H && o.Wo == o.W && o.Cin > 0 &&
               o.Cin % 32 == 0 && E == 256 && o.Cout > 0 && o.Cout % 32 == 0 && o.Cout <= 128 && o.Cin_total == o.Cin && o.cin_off == 0 && o.Cout_total == o.Cout &&
               o.cout_off == 0 && o.groups <= 1 && o.act == FTC_ACT_SILU && (o.flags & ~(FTC_FLAG_RESIDUAL | FTC_FLAG_SPLIT16)) == 0 &&
               (!(o.flags & FTC_FLAG_RESIDUAL) || o.res_dtype == FTC_F32) && px * o.Cin * 4 < 0x7fffffffL && px > 0 && px < 0x7fffffffL / 128;
    return ftc_is16(o.w_dtype) && o.in_dtype == o.w_dtype && o.out_dtype == FTC_F32 && o.ksize == 3 && o.stride == 1 && o.Ho == o.H && o.Wo == o.W &&
           o.Cin > 0 && o.Cin % 32 == 0 && (E == 256 || E == 384) && o.Cout > 0 && o.Cout % 32 == 0 && o.Cout <= 128 && o.Cin_total == o.Cin && o.cin_off == 0 &&
           o.Cout_total == o.Cout && o.cout_off == 0 && o.groups <= 1 && o.act == FTC_ACT_SILU && (o.flags & ~(FTC_FLAG_RESIDUAL)) == 0 &&
           (!(o.flags & FTC_FLAG_RESIDUAL) || o.res_dtype == FTC_F32) && px * o.Cin * 2 < 0x7fffffffL && px > 0 && px < 0x7fffffffL / 128;
}

const char* ftc_fmbconv_label(const ftc_op& o, char* buf, int len) {
    if (o.w_dtype == FTC_F32) std::snprintf(buf, len, "fmbconv_fused<f16x3,e=%d,bk=32>", o.aux1);
    else std::snprintf(buf, len, "fmbconv_fused<%s,e=%d,bk=%d>", o.w_dtype == FTC_F16 ? "f16" : "bf16", o.aux1, o.Cin % 64 == 0 ? 64 : 32);
    return buf;
}

hipError_t launch_fmbconv(const OpArgs& a, hipStream_t s) {
    using namespace convimpl;
    const ftc_op& o = *a.op;
    if (!ftc_fmbconv_legal(o)) return hipErrorInvalidValue;
    FmbP p;
    p.x = a.in; p.w1 = a.w2; p.b1 = a.bias2; p.w2 = a.w; p.b2 = a.bias;
    p.res = (o.flags & FTC_FLAG_RESIDUAL) ? static_cast<const float*>(a.in2) : nullptr;
    p.out = static_cast<float*>(a.out); p.out2 = a.out2;
    p.B = o.B; p.H = o.H; p.W = o.W; p.Cin = o.Cin; p.E = o.aux1; p.Cout = o.Cout;
    p.M = o.B * o.H * o.W;
    const long es = o.w_dtype == FTC_F32 ? 4 : 2;
    p.x_bytes = (unsigned)((long)p.M * o.Cin * es);
    p.w1_bytes = (unsigned)((long)p.E * 9 * o.Cin * es);
    p.w2_bytes = (unsigned)((long)o.Cout * p.E * es);
    if (o.w_dtype == FTC_F32) {
        p.ncb = o.Cin / 32;
        p.nk = 9 * p.ncb;
        return launch_fmb_x3(p, s);
    }
    const int bk = o.Cin % 64 == 0 ? 64 : 32;
    p.ncb = o.Cin / bk;
    p.nk = 9 * p.ncb;
    p.nblk = 0;
    const bool h = o.w_dtype == FTC_F16;
#define FMB_GO(BK_, SN_) return h ? launch_fmb<_Float16, BK_, SN_, 2, 1>(p, s) : launch_fmb<__bf16, BK_, SN_, 2, 1>(p, s)
    if (p.E == 256) { if (bk == 64) FMB_GO(64, 2); FMB_GO(32, 2); }
    if (bk == 64) FMB_GO(64, 3);
    FMB_GO(32, 3);
#undef FMB_GO
}
