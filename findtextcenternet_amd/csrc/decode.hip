// GPU peak decode + 100-d feature gather: the replacement of the reference's per-tile host loop
// (/root/reference/process_ocr_base.py:518-538 == /root/reference/test_image1_torch.py:123-143;
// sigmoid = /root/reference/util_func.py:14-15).
//
// The reference copies the whole [10,h,w] + [100,h,w] maps to the host, argsorts 36 864 scores and
// walks them in a Python loop.  Here only the kept peaks leave the GPU:
//   1. select : every pixel of the trusted rectangle whose NMS'd key logit passes the cut-off and
//               whose decoded box passes the reference's w/h checks gets a 64-bit sort key
//               (orderable logit bits << 32 | ~pixel index)  -> per-image candidate list;
//   2. rank   : rank[i] = #candidates with a larger key (keys staged through LDS in 256-wide
//               chunks) -- a total order (score desc, pixel index asc), deterministic whatever
//               order step 1's atomics produced;
//   3. gather : the first max_boxes ranks write their 9-float box record and copy their
//               contiguous NHWC feature row (100 floats = 25 lanes x 16 B).
#include "ftc_common.h"

namespace {

__device__ __forceinline__ float ref_sigmoid(float x) { return (tanhf(x * 0.5f) + 1.0f) * 0.5f; }   // util_func.py:15

__device__ __forceinline__ uint32_t orderable(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Round 6.  The select pass used to take 50 us for 12 MB of reads (profiles/r05e_bf16_b8_kernel_stats.txt): one pixel per thread, and even with
// one atomicAdd per WAVE (round 5) an image's counter saw ~540 same-address atomics, which the L2 serialises at ~90 ns each.  Now a workgroup
// owns a contiguous 1/SEL_WG of the trusted rectangle: every thread has all its SEL_PT key logits in flight at once (one round of loads), the
// candidates' w/h checks are a second, sparse round, the slots inside the workgroup come from ballots + an LDS scan over the waves, and the
// image's counter sees ONE atomicAdd per workgroup (16 per image).  Slot order is arbitrary as before -- the rank pass imposes the total order.
constexpr int SEL_WG = 16;            // workgroups per image
constexpr int SEL_PT = 12;            // positions per thread and pass (256 x 12 x 16 = 49,152 >= 192 x 192; larger maps take more passes)

__global__ __launch_bounds__(256) void decode_select_kernel(const float* __restrict__ heat, const ftc_tile* __restrict__ tiles,
                                                            int h, int w, float logit_cut, unsigned long long* __restrict__ cand,
                                                            int32_t* __restrict__ counts) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int b = blockIdx.y;
    const ftc_tile tl = tiles[b];
    const int rw = tl.x_max - tl.x_min, rh = tl.y_max - tl.y_min;
    const int n = rw * rh;
    const float* hb = heat + (long)b * h * w * 10;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per_wg = (n + SEL_WG - 1) / SEL_WG;
    const int lo = blockIdx.x * per_wg, hi = min(n, lo + per_wg);
    for (int p0 = lo; p0 < hi; p0 += 256 * SEL_PT) {              // (one pass for maps up to 192 x 192)
        float v[SEL_PT];
        int idx[SEL_PT];
#pragma unroll
        for (int u = 0; u < SEL_PT; ++u) {                         // position = p0 + u * 256 + t: a wave reads 64 consecutive pixels
            const int i = p0 + u * 256 + t;
            idx[u] = -1;
            v[u] = -__builtin_huge_valf();
            if (i < hi) {
                const int y = tl.y_min + i / rw, x = tl.x_min + i % rw;
                idx[u] = y * w + x;
                v[u] = hb[(long)idx[u] * 10 + 1];
            }
        }
        unsigned keepm = 0;
        int mine = 0;
#pragma unroll
        for (int u = 0; u < SEL_PT; ++u) {
            if (idx[u] >= 0 && v[u] >= logit_cut) {                // -inf (suppressed) and NaN fail
                const float bw = expf(hb[(long)idx[u] * 10 + 2] - 3.0f) * 1024.0f;   // process_ocr_base.py:523-524
                const float bh = expf(hb[(long)idx[u] * 10 + 3] - 3.0f) * 1024.0f;
                if (!(bw <= 0.f || bh <= 0.f) &&                   // :525
                    !(bw > (float)tl.page_w || bh > (float)tl.page_h)) {   // :527
                    keepm |= 1u << u;
                    ++mine;
                }
            }
        }
        // exclusive prefix of `mine` over the workgroup: lanes by shuffles, waves through LDS
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        const int w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
        const int total = w0 + w1 + w2 + w3;
        if (t == 0) s_base = total ? atomicAdd(&counts[b], total) : 0;
        __syncthreads();
        int slot = s_base + (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0) + incl - mine;
#pragma unroll
        for (int u = 0; u < SEL_PT; ++u)
            if (keepm & (1u << u))
                cand[(long)b * h * w + slot++] = ((unsigned long long)orderable(v[u]) << 32) | (uint32_t)(~(uint32_t)idx[u]);
        __syncthreads();                                             // s_wave / s_base are reused by the next pass
    }
}

__global__ __launch_bounds__(256) void decode_rank_gather_kernel(const float* __restrict__ heat, const float* __restrict__ feat,
                                                                 const ftc_tile* __restrict__ tiles, int h, int w, int C, int scale,
                                                                 const unsigned long long* __restrict__ cand,
                                                                 const int32_t* __restrict__ counts, int max_boxes,
                                                                 float* __restrict__ boxes, int box_stride, float* __restrict__ feats,
                                                                 int feat_stride, int32_t* __restrict__ index) {
    // (round 5) A workgroup owns DG = 64 candidates (it was 256: the ~1600 candidates of a tile kept 7 workgroups per image busy and each of their
    // waves copied 64 feature rows one after the other -- 62 us).  Its 256 threads split the comparison range four ways (thread = candidate x
    // quarter of the list), the four partial ranks meet in LDS, and every wave then copies 16 rows, four at a time.
    constexpr int DG = 64;
    __shared__ unsigned long long keys[256];
    __shared__ int s_part[4][DG];
    __shared__ int s_rank[DG];
    __shared__ int s_idx[DG];
    const int b = blockIdx.y;
    const int n = counts[b];
    const int base = blockIdx.x * DG;
    if (base >= n) return;
    const unsigned long long* cb = cand + (long)b * h * w;
    const int t = threadIdx.x;
    const int cl = t & (DG - 1), qd = t >> 6;                       // candidate of the block, quarter of each 256-key chunk
    const int me = base + cl;
    const unsigned long long mykey = me < n ? cb[me] : 0ull;
    int part = 0;
    for (int c0 = 0; c0 < n; c0 += 256) {
        __syncthreads();
        keys[t] = (c0 + t < n) ? cb[c0 + t] : 0ull;
        __syncthreads();
        const int lim = min(64, n - c0 - qd * 64);
        for (int j = 0; j < lim; ++j) part += keys[qd * 64 + j] > mykey ? 1 : 0;
    }
    s_part[qd][cl] = part;
    __syncthreads();
    const int rank = (s_part[0][cl] + s_part[1][cl]) + (s_part[2][cl] + s_part[3][cl]);
    const int idx = (int)(~(uint32_t)(mykey & 0xffffffffull));
    if (t < DG) {
        s_rank[t] = (me < n && rank < max_boxes) ? rank : -1;
        s_idx[t] = idx;
    }
    __syncthreads();

    const ftc_tile tl = tiles[b];
    const float* hb = heat + (long)b * h * w * 10;
    if (t < DG && s_rank[t] >= 0) {
        const float* px = hb + (long)idx * 10;
        const int y = idx / w, x = idx - y * w;
        float* o = boxes + ((long)b * max_boxes + rank) * box_stride;
        o[0] = ref_sigmoid(px[1]);
        o[1] = (float)(x * scale + tl.offset_x);
        o[2] = (float)(y * scale + tl.offset_y);
        o[3] = expf(px[2] - 3.0f) * 1024.0f;
        o[4] = expf(px[3] - 3.0f) * 1024.0f;
        o[5] = ref_sigmoid(px[6]);
        o[6] = ref_sigmoid(px[7]);
        o[7] = ref_sigmoid(px[8]);
        o[8] = ref_sigmoid(px[9]);
        index[(long)b * max_boxes + rank] = idx;
    }
    // feature rows: one wave per candidate, 16 B per lane
    const int lane = t & 63, wave = t >> 6;
    const int CQ = C >> 2;
    const float* fb = feat + (long)b * h * w * C;
    if (CQ <= 64) {                                                   // (C <= 256: one 16-byte access per lane and row -- four rows in flight)
        // (round 6) all 16 rows of the wave in flight at once: the four dependent rounds of four rows cost ~1.5 us each
        const int k0 = wave * 16;
        f32x4 v[16];
        int r[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            r[u] = s_rank[k0 + u];
            if (r[u] >= 0 && lane < CQ) v[u] = reinterpret_cast<const f32x4*>(fb + (long)s_idx[k0 + u] * C)[lane];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (r[u] >= 0 && lane < CQ) reinterpret_cast<f32x4*>(feats + ((long)b * max_boxes + r[u]) * feat_stride)[lane] = v[u];
    } else {
        for (int k = wave * 16; k < wave * 16 + 16; ++k) {
            const int r = s_rank[k];
            if (r < 0) continue;
            const float* src = fb + (long)s_idx[k] * C;
            float* dst = feats + ((long)b * max_boxes + r) * feat_stride;
            for (int q = lane; q < CQ; q += 64) reinterpret_cast<f32x4*>(dst)[q] = reinterpret_cast<const f32x4*>(src)[q];
        }
    }
}

}  // namespace

extern "C" int64_t ftc_decode_scratch_bytes(int B, int h, int w) { return (int64_t)B * h * w * 8; }

hipError_t launch_decode(const float* heat, const float* feat, int B, int h, int w, int C, const ftc_tile* tiles,
                         float logit_cut, int scale, int max_boxes, float* boxes, int box_stride, float* feats, int feat_stride,
                         int32_t* index, int32_t* counts, void* scratch, hipStream_t s) {
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * B, s);
    if (e != hipSuccess) return e;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(scratch);
    hipLaunchKernelGGL(decode_select_kernel, dim3(SEL_WG, B), dim3(256), 0, s, heat, tiles, h, w, logit_cut, cand, counts);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(decode_rank_gather_kernel, dim3((h * w + 63) / 64, B), dim3(256), 0, s, heat, feat, tiles, h, w, C, scale, cand, counts,
                       max_boxes, boxes, box_stride, feats, feat_stride, index);
    return hipGetLastError();
}
