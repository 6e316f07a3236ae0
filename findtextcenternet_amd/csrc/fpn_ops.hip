// FPN-side bandwidth kernels: bilinear x2 upsample (align_corners=True) + concat + per-head input
// BatchNorm, and the 3x3 max-pool NMS of the key heat-map.
//
// Reference: Leafmap.forward (/root/reference/models/detector.py:192-201; nn.UpsamplingBilinear2d
// at :170/:177 == F.interpolate(mode='bilinear', align_corners=True)) and
// CenterNetDetector.forward (/root/reference/models/detector.py:289-296).
#include "ftc_common.h"

namespace {

// out[b,y,x, 0:Cy]      = bilinear_x2(prev[b,:,:,0:Cy])           (Cy = 0: absent)
// out[b,y,x, Cy:Cy+Ct]  = tap[b,y,x,:] * scale + shift            (eval-mode BatchNorm2d)
// One lane = one pixel x 4 channels; consecutive lanes walk the channel axis (coalesced).
template <typename T, typename TapT>
__global__ __launch_bounds__(256) void upcat_kernel(const T* __restrict__ prev, const TapT* __restrict__ tap,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    T* __restrict__ out, int B, int Hi, int Wi, int Ho, int Wo,
                                                    int Cy, int Ct, int CyT, float ry, float rx, long prev_gs, long out_gs) {
    constexpr int V = 16 / (int)sizeof(T);         // channels per lane = one 16-byte store (4 fp32 | 8 bf16)
    const int Ctot = Cy + Ct;
    // grouped launch (ftc_op.groups): blockIdx.y = instance; the backbone tap is shared, everything else is per instance
    prev += blockIdx.y * prev_gs;
    out += blockIdx.y * out_gs;
    scale += blockIdx.y * Ct;
    shift += blockIdx.y * Ct;
    const int Q = Ctot / V;
    // 32-bit index arithmetic (validated: B*Ho*Wo*Q < 2^31): the 64-bit divisions of the first version cost more VALU time
    // than the interpolation itself
    const unsigned total = (unsigned)B * Ho * Wo * Q;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned pix = idx / (unsigned)Q;
        const int q = (int)(idx - pix * Q);
        const int c = q * V;
        float v[V];
        if (c < Cy) {
            const unsigned row = pix / (unsigned)Wo;
            const int x = (int)(pix - row * Wo);
            const int b = (int)(row / (unsigned)Ho);
            const int y = (int)(row - (unsigned)b * Ho);
            // ATen upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1)
            const float sy = ry * (float)y, sx = rx * (float)x;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
            const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
            const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
            const T* base = prev + (long)b * Hi * Wi * CyT + c;      // prev already points at its channel slice
            float v00[V], v01[V], v10[V], v11[V];
            load16<T>(base + ((long)y0 * Wi + x0) * CyT, v00);
            load16<T>(base + ((long)y0 * Wi + x1) * CyT, v01);
            load16<T>(base + ((long)y1 * Wi + x0) * CyT, v10);
            load16<T>(base + ((long)y1 * Wi + x1) * CyT, v11);
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
        } else {
            const int ct = c - Cy;
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const f32x4 xv = load4<TapT>(tap + (long)pix * Ct + ct + 4 * h);
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ct + 4 * h);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ct + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * h + e] = xv[e] * sc[e] + sh[e];
            }
        }
        store16<T>(out + (long)pix * Ctot + c, v);
    }
}

// heat[b,y,x,1] = heat[b,y,x,0] < max3x3(heat[..,0]) ? -inf : heat[b,y,x,0]   (pad = -inf)
// 32x32 pixel tile per workgroup, 34x34 halo of channel 0 staged in LDS.
__global__ __launch_bounds__(256) void nms_kernel(float* __restrict__ heat, int h, int w, int CH) {
    __shared__ float tile[34][35];
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * 32, x0 = blockIdx.x * 32;
    float* hb = heat + (long)b * h * w * CH;
    const float ninf = -__builtin_huge_valf();
    for (int i = threadIdx.x; i < 34 * 34; i += 256) {
        const int ry = i / 34, rx = i - ry * 34;
        const int y = y0 + ry - 1, x = x0 + rx - 1;
        tile[ry][rx] = ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) ? hb[((long)y * w + x) * CH] : ninf;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int ly = i >> 5, lx = i & 31;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= h || x >= w) continue;
        const float k = tile[ly + 1][lx + 1];
        float m = ninf;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, tile[ly + dy][lx + dx]);
        hb[((long)y * w + x) * CH + 1] = (k < m) ? ninf : k;
    }
}

// 9-point sum that finishes a top convolution whose per-pixel taps T were produced by the FTC_FLAG_TOP_FUSE epilogue.
// Workgroup = a 16x16 pixel tile of one image; for each head in turn the 18x18 halo of its tap rows is staged in LDS with
// coalesced 16-byte loads (T is read once, sequentially), then lane = pixel adds the nine taps of the head's outputs.
__global__ __launch_bounds__(256) void tapsum_kernel(const float* __restrict__ T, const int* __restrict__ map, const float* __restrict__ bias,
                                                     float* __restrict__ out, int H, int W, int Tw, int nout, int CH, long gs, int G) {
    // [18*18][Tw + 1]: with the natural pitch (Tw = 20 floats) every address of a wave's tap read is a multiple of 4 floats, i.e. the 64
    // lanes share 16 of the 64 banks (SQ_LDS_BANK_CONFLICT 45.7 % of the LDS cycles, profiles/r03b); an odd pitch spreads the 64
    // pixels of a wave over 64 banks (pixel index * 21 mod 64 is a permutation; six lanes of the fourth row wrap -> 2-way).
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int t = threadIdx.x;
    const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
    int bid = blockIdx.x;
    const int b = bid / (tilesX * tilesY);
    bid -= b * tilesX * tilesY;
    const int y0 = (bid / tilesX) * 16, x0 = (bid % tilesX) * 16;
    const int ly = t >> 4, lx = t & 15;
    const int y = y0 + ly, x = x0 + lx;
    const bool inside = y < H && x < W;
    const long pix = ((long)b * H + y) * W + x;
    const int Q = Tw >> 2;                                                // float4 per pixel row
    const int TP = Tw + 1;
    for (int g = 0; g < G; ++g) {
        __syncthreads();                                                  // previous head's rows are consumed
        const f32x4* Tg = reinterpret_cast<const f32x4*>(T + g * gs + (long)b * H * W * Tw);
        for (int c = t; c < 324 * Q; c += 256) {
            const int hp = c / Q, q = c - hp * Q;
            const int yy = y0 - 1 + hp / 18, xx = x0 - 1 + hp % 18;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};                               // zero padding of the top convolution
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = Tg[((long)yy * W + xx) * Q + q];
            float* dst = rows + hp * TP + 4 * q;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        __syncthreads();
        if (!inside) continue;
        for (int j = 0; j < nout; ++j) {
            if (map[4 * j] != g) continue;
            const int o = map[4 * j + 1], co = map[4 * j + 2], ch = map[4 * j + 3];
            float acc = bias[j];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) acc += rows[((ly + r) * 18 + lx + s2) * TP + (r * 3 + s2) * co + o];
            out[pix * CH + ch] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// THIN 3x3 convolution: fp32 tensors, ONE to FOUR output channels (the map heads' top convolutions in the fp32 / fp16x3 plans, where they
// are not fused into the last FPN level's epilogue).  On the matrix cores the 1-2 output channels were padded to a 32-row tile: 1.5 ms for
// the six single-channel heads at 4 TFLOP/s.  Here thread = pixel of a 16x16 tile, the 18x18 halo of 32 input channels is DMA'd into LDS
// per channel block (16-byte chunk slot = chunk ^ (pixel & 7): a wave's reads spread over the banks), the block's 9 x 32 x CO weights sit
// in LDS and are read as broadcasts, the sums are plain fp32 FMAs.  fp16x3 plans store their weights pre-split ([hi x4 | lo x4] halves per
// four fp32): w = hi + lo.  grid = (tiles, groups); a group's operands as in the MFMA kernels (in / w / bias strides, output channel slice).
// ------------------------------------------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256) void thin_conv3x3_kernel(const float* __restrict__ in, const void* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ out, int B, int H, int W, int Cin, int Cout, int CoutT, int cout_off,
                                                           int cout_gs, long in_gs, long w_gs, int split) {
    constexpr int NH = 18 * 18, CB = 32, CPRW = CB / 4;          // halo pixels, channels per block, 16-byte chunks per pixel
    __shared__ __attribute__((aligned(16))) float halo[(NH * CPRW + 63) / 64 * 256];      // whole wave-level DMAs (the last one is half padding)
    __shared__ __attribute__((aligned(16))) float wl[9 * CB * CO];                        // [tap][4-channel group][co][4]
    const int t = threadIdx.x, g = blockIdx.y;
    const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
    int bid = blockIdx.x;
    const int b = bid / (tilesX * tilesY);
    bid -= b * tilesX * tilesY;
    const int y0 = (bid / tilesX) * 16, x0 = (bid % tilesX) * 16;
    const int ly = t >> 4, lx = t & 15;
    const float* inb = in + g * (in_gs >> 2) + (long)b * H * W * Cin;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb), 0, H * W * Cin * 4, 0x00020000);
    const char* wg = static_cast<const char*>(w) + g * w_gs;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    for (int cb = 0; cb < Cin; cb += CB) {
        __syncthreads();                                             // the previous block's halo and weights are consumed
        // halo: NH * 8 chunks; wave-level DMA k covers chunks [64 k, 64 k + 64); LDS slot (pixel, c') holds the pixel's chunk c' ^ (pixel & 7)
        for (int k = wave; k * 64 < NH * CPRW; k += 4) {
            const int q = k * 64 + lane, hp = q / CPRW, cs = (q % CPRW) ^ (hp & 7);
            const int yy = y0 - 1 + hp / 18, xx = x0 - 1 + hp % 18;
            const bool ok = q < NH * CPRW && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(halo + k * 256), 16,
                                                     ok ? ((yy * W + xx) * Cin + cb + cs * 4) * 4 : 0x7ffffff0, 0, 0, 0);
        }
        // weights of this block: wl[((tap * 8 + c4) * CO + co) * 4 + e] = W[co][tap][cb + 4 c4 + e]
        for (int i = t; i < 9 * (CB / 4) * CO; i += 256) {
            const int co = i % CO, r = i / CO, c4 = r % (CB / 4), tap = r / (CB / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (co < Cout) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(wg + (((long)co * 9 + tap) * Cin + cb + c4 * 4) * 4);
                if (split) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    const u32x2 hu = {raw[0], raw[1]}, lu = {raw[2], raw[3]};
                    const h4 hi = __builtin_bit_cast(h4, hu), lo = __builtin_bit_cast(h4, lu);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)hi[e] + (float)lo[e];
                } else {
                    v = __builtin_bit_cast(f32x4, raw);
                }
            }
            *reinterpret_cast<f32x4*>(wl + ((tap * (CB / 4) + c4) * CO + co) * 4) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const int hp = (ly + r) * 18 + lx + s2;
                const float* hx = halo + hp * CB;
                const float* wt = wl + (r * 3 + s2) * CB * CO;
#pragma unroll
                for (int c4 = 0; c4 < CB / 4; ++c4) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(hx + ((c4 ^ (hp & 7)) << 2));
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + ((c4 * CO + c) << 2));      // same address in every lane: a broadcast
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[c] = fmaf(xv[e], wv[e], acc[c]);
                    }
                }
            }
    }
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) return;
    float* op = out + (((long)b * H + y) * W + x) * CoutT + cout_off + g * cout_gs;
#pragma unroll
    for (int c = 0; c < CO; ++c)
        if (c < Cout) op[c] = acc[c] + bias[g * Cout + c];
}

}  // namespace

hipError_t launch_tapsum(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int nb = o.B * ((o.H + 15) / 16) * ((o.W + 15) / 16);
    hipLaunchKernelGGL(tapsum_kernel, dim3(nb), dim3(256), (size_t)324 * (o.aux0 + 1) * sizeof(float), s, (const float*)a.in, (const int*)a.w, a.bias,
                       (float*)a.out, o.H, o.W, o.aux0, o.aux1, o.Cout_total, (long)o.B * o.H * o.W * o.aux0, o.groups > 1 ? o.groups : 1);
    return hipGetLastError();
}

hipError_t launch_upcat(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int Cy = o.aux0, Ct = o.aux1;
    const int CyT = o.Cin_total > 0 ? o.Cin_total : Cy;     // channel stride of the upsampled tensor
    const int V = o.in_dtype == FTC_F32 ? 4 : 8;
    const long total = (long)o.B * o.Ho * o.Wo * ((Cy + Ct) / V);
    long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    const int G = o.groups > 1 ? o.groups : 1;
    const long prev_gs = G == 1 ? 0 : (o.flags & FTC_FLAG_GROUP_IN_SLICE) ? Cy : (long)o.B * o.H * o.W * CyT;
    const long out_gs = G == 1 ? 0 : (long)o.B * o.Ho * o.Wo * (Cy + Ct);
    const float ry = o.Ho > 1 ? (float)(o.H - 1) / (float)(o.Ho - 1) : 0.f;
    const float rx = o.Wo > 1 ? (float)(o.W - 1) / (float)(o.Wo - 1) : 0.f;
#define UPCAT_LAUNCH(T, TT)                                                                                       \
    hipLaunchKernelGGL((upcat_kernel<T, TT>), dim3((unsigned)nb, G), dim3(256), 0, s, (const T*)a.in + o.cin_off, (const TT*)a.in2, \
                       a.scale, a.shift, (T*)a.out, o.B, o.H, o.W, o.Ho, o.Wo, Cy, Ct, CyT, ry, rx, prev_gs, out_gs)
    // res_dtype = dtype of the backbone tap (the trunk stays fp32 in bf16 mode)
    if (o.in_dtype == FTC_F32) UPCAT_LAUNCH(float, float);
    else if (o.in_dtype == FTC_F16) { if (o.res_dtype == FTC_F32) UPCAT_LAUNCH(_Float16, float); else UPCAT_LAUNCH(_Float16, _Float16); }
    else if (o.res_dtype == FTC_F32) UPCAT_LAUNCH(__bf16, float);
    else UPCAT_LAUNCH(__bf16, __bf16);
#undef UPCAT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_nms(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    hipLaunchKernelGGL(nms_kernel, dim3((o.W + 31) / 32, (o.H + 31) / 32, o.B), dim3(256), 0, s, (float*)a.out, o.H, o.W,
                       o.Cout_total);
    return hipGetLastError();
}

// FTC_OP_CONV, 3x3 stride 1, fp32 in / out / weights (plain or fp16x3 pre-split), 1..4 output channels, no activation / residual / gate
bool ftc_thin_conv_legal(const ftc_op& o) {
    return o.ksize == 3 && o.stride == 1 && o.Cout >= 1 && o.Cout <= 4 && o.in_dtype == FTC_F32 && o.out_dtype == FTC_F32 && o.w_dtype == FTC_F32 &&
           o.act == FTC_ACT_NONE && o.Cin % 32 == 0 && o.Cin_total == o.Cin && o.cin_off == 0 && o.H == o.Ho && o.W == o.Wo &&
           !(o.flags & (FTC_FLAG_RESIDUAL | FTC_FLAG_SE_SCALE | FTC_FLAG_BORDER_BIAS | FTC_FLAG_W_PER_IMAGE | FTC_FLAG_UPCAT_IN | FTC_FLAG_TOP_FUSE | 0x100)) &&
           (long)o.H * o.W * o.Cin * 4 < 0x7ff00000L;
}

hipError_t launch_thin_conv(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int G = o.groups > 1 ? o.groups : 1;
    const bool oslice = (o.flags & FTC_FLAG_GROUP_OUT_SLICE) != 0;
    if (G > 1 && !oslice) return hipErrorInvalidValue;               // (stacked group outputs: not needed by any plan)
    const dim3 grid((unsigned)(o.B * ((o.H + 15) / 16) * ((o.W + 15) / 16)), (unsigned)G);
    const long in_gs = (long)o.B * o.H * o.W * o.Cin * 4, w_gs = (long)o.Cout * 9 * o.Cin * 4;
    const int split = (o.flags & FTC_FLAG_SPLIT16) ? 1 : 0;
#define THIN(CO) hipLaunchKernelGGL(thin_conv3x3_kernel<CO>, grid, dim3(256), 0, s, (const float*)a.in, a.w, a.bias, (float*)a.out, o.B, o.H, o.W, o.Cin, \
                                    o.Cout, o.Cout_total, o.cout_off, oslice ? o.Cout : 0, in_gs, w_gs, split)
    if (o.Cout == 1) THIN(1); else if (o.Cout == 2) THIN(2); else THIN(4);
#undef THIN
    return hipGetLastError();
}

