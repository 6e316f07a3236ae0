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

__global__ __launch_bounds__(256) void decode_select_kernel(const float* __restrict__ heat, const ftc_tile* __restrict__ tiles,
                                                            int h, int w, float logit_cut, unsigned long long* __restrict__ cand,
                                                            int32_t* __restrict__ counts) {
    const int b = blockIdx.y;
    const ftc_tile tl = tiles[b];
    const int rw = tl.x_max - tl.x_min, rh = tl.y_max - tl.y_min;
    const int n = rw * rh;
    const float* hb = heat + (long)b * h * w * 10;
    // (round 5) One atomicAdd per WAVE and pass instead of one per candidate: a random-init tile has ~1600 candidates, and their atomics on the
    // image's one counter serialised at the L2 (50 us for 12 MB of reads).  The slot order is arbitrary either way -- the rank pass below
    // imposes the total order.  The loop bound is wave-uniform so that the ballot sees every lane.
    const int lane = threadIdx.x & 63;
    const int n_up = (n + 63) & ~63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_up; i += gridDim.x * blockDim.x) {
        bool keep = false;
        int idx = 0;
        float v = 0.f;
        if (i < n) {
            const int y = tl.y_min + i / rw, x = tl.x_min + i % rw;
            idx = y * w + x;
            v = hb[(long)idx * 10 + 1];
            if (v >= logit_cut) {                                     // -inf (suppressed) and NaN fail
                const float bw = expf(hb[(long)idx * 10 + 2] - 3.0f) * 1024.0f;   // process_ocr_base.py:523-524
                const float bh = expf(hb[(long)idx * 10 + 3] - 3.0f) * 1024.0f;
                keep = !(bw <= 0.f || bh <= 0.f) &&                   // :525
                       !(bw > (float)tl.page_w || bh > (float)tl.page_h);   // :527
            }
        }
        const unsigned long long m = __ballot(keep);
        if (m == 0ull) continue;
        int base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&counts[b], __popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, 64);
        if (keep) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            cand[(long)b * h * w + slot] = ((unsigned long long)orderable(v) << 32) | (uint32_t)(~(uint32_t)idx);
        }
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
        for (int k0 = wave * 16; k0 < wave * 16 + 16; k0 += 4) {
            f32x4 v[4];
            int r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                r[u] = s_rank[k0 + u];
                if (r[u] >= 0 && lane < CQ) v[u] = reinterpret_cast<const f32x4*>(fb + (long)s_idx[k0 + u] * C)[lane];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r[u] >= 0 && lane < CQ) reinterpret_cast<f32x4*>(feats + ((long)b * max_boxes + r[u]) * feat_stride)[lane] = v[u];
        }
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
    const int nb = (h * w + 255) / 256;
    hipLaunchKernelGGL(decode_select_kernel, dim3(nb, B), dim3(256), 0, s, heat, tiles, h, w, logit_cut, cand, counts);      // one pixel per thread: a single round of loads (36 workgroups per image walked four dependent rounds: 50 us)
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(decode_rank_gather_kernel, dim3((h * w + 63) / 64, B), dim3(256), 0, s, heat, feat, tiles, h, w, C, scale, cand, counts,
                       max_boxes, boxes, box_stride, feats, feat_stride, index);
    return hipGetLastError();
}
