// Page-level box selection on the GPU: the contrast filter, the sequential greedy suppression with
// its coverage rule, the separator filter and the 3x3 code maximum of OCR_Processer.run_detector
// (/root/reference/process_ocr_base.py:540-650, imageHist :652-693; restated and pinned bit-exactly against the
// reference's own outputs in oracle/decode_oracle.py:page_merge).
//
// The reference runs this in NumPy float64 on the host, O(N^2) over the page's boxes.  The semantics are sequential
// (a box is judged against the boxes KEPT so far), so the greedy pass stays sequential here too: ONE workgroup walks the
// score-sorted boxes and parallelises inside a step -- the comparison against the kept list (1024 lanes), the coverage
// bitmap (LDS bit image + popcount).  All arithmetic is IEEE float64 with contraction off, integer histogram sums are
// exact, so the results are bit-identical to the reference's.
#include "ftc_common.h"

#pragma clang fp contract(off)

namespace {

// ---- imageHist (process_ocr_base.py:652-693): distance between the two 1-D 2-means centres of a 256-bin histogram ----
__device__ double cluster_dist(const unsigned int* hist) {
    long long tot = 0, sv = 0;
    for (int i = 0; i < 256; ++i) { tot += hist[i]; sv += (long long)hist[i] * i; }
    if (tot == 0) return 0.0;
    const int cut = (int)((double)sv / (double)tot + 0.5);
    long long s1 = 0, s2 = 0, v1 = 0, v2 = 0;
    for (int i = 0; i < 256; ++i) {
        if (i < cut) { s1 += hist[i]; v1 += (long long)hist[i] * i; }
        else { s2 += hist[i]; v2 += (long long)hist[i] * i; }
    }
    if (s1 == 0 || s2 == 0) return 0.0;
    double k1 = (double)v1 / (double)s1, k2 = (double)v2 / (double)s2;
    double prev = 256.0, cur = fabs(k1 - k2);
    while (prev != cur) {
        prev = cur;
        s1 = s2 = v1 = v2 = 0;
        for (int i = 0; i < 256; ++i) {
            const bool near1 = fabs((double)i - k1) < fabs((double)i - k2);
            if (near1) { s1 += hist[i]; v1 += (long long)hist[i] * i; }
            else { s2 += hist[i]; v2 += (long long)hist[i] * i; }
        }
        if (s1 == 0 || s2 == 0) return 0.0;
        k1 = (double)v1 / (double)s1;
        k2 = (double)v2 / (double)s2;
        cur = fabs(k1 - k2);
    }
    return prev;
}

// Python slice bounds [a:b] on an axis of length n
__device__ __forceinline__ void py_slice(long a, long b, int n, int* lo, int* hi) {
    if (a < 0) a += n;
    if (a < 0) a = 0;
    if (a > n) a = n;
    if (b < 0) b += n;
    if (b < 0) b = 0;
    if (b > n) b = n;
    *lo = (int)a;
    *hi = (int)(b > a ? b : a);
}

// blockIdx.x = box, blockIdx.y = variant: 0 = the threshold sample (:563-571, raw Python slices around the box),
// 1 = the crop tested in the greedy loop (:579-582, clamped to the page).  out[variant][box].
__global__ __launch_bounds__(256) void box_hist_kernel(const float* __restrict__ loc, int N, const float* __restrict__ page, int PH, int PW,
                                                       float cut_off, double* __restrict__ out) {
    __shared__ unsigned int hist[3][256];
    __shared__ double res[3];
    const int i = blockIdx.x, variant = blockIdx.y, t = threadIdx.x;
    const double p = loc[i * 9], cx = loc[i * 9 + 1], cy = loc[i * 9 + 2], w = loc[i * 9 + 3], h = loc[i * 9 + 4];
    if (p < (double)cut_off) {
        if (t == 0) out[(long)variant * N + i] = 0.0;
        return;
    }
    int x0, x1, y0, y1;
    if (variant == 0) {
        py_slice((long)(cy - h / 2) - 1, (long)(cy + h / 2) + 2, PH, &y0, &y1);
        py_slice((long)(cx - w / 2) - 1, (long)(cx + w / 2) + 2, PW, &x0, &x1);
    } else {
        const long bx0 = max(0L, (long)(cx - w / 2)), bx1 = min((long)PW - 1, (long)(cx + w / 2) + 1);
        const long by0 = max(0L, (long)(cy - h / 2)), by1 = min((long)PH - 1, (long)(cy + h / 2) + 1);
        py_slice(by0, by1, PH, &y0, &y1);
        py_slice(bx0, bx1, PW, &x0, &x1);
    }
    for (int k = t; k < 3 * 256; k += 256) (&hist[0][0])[k] = 0;
    __syncthreads();
    const int cw = x1 - x0, ch = y1 - y0;
    for (long k = t; k < (long)cw * ch * 3; k += 256) {
        const int c = (int)(k % 3);
        const long px = k / 3;
        const int x = x0 + (int)(px % cw), y = y0 + (int)(px / cw);
        const float v = page[((long)y * PW + x) * 3 + c];
        if (v >= 0.0f && v <= 256.0f) {                         // np.histogram(bins=256, range=(0, 256)): last bin closed
            int b = (int)v;
            if (b > 255) b = 255;
            atomicAdd(&hist[c][b], 1u);
        }
    }
    __syncthreads();
    if (t < 3) res[t] = cluster_dist(hist[t]);
    __syncthreads();
    if (t == 0) {
        double best = -1.0;
        for (int c = 0; c < 3; ++c) best = res[c] > best ? res[c] : best;
        out[(long)variant * N + i] = best;
    }
}

// ---- the greedy pass: one workgroup, boxes in score order ----
constexpr int GT = 1024;
constexpr int FILL_WORDS = 8192;          // LDS bit image of the candidate box: up to 262144 cells (512 x 512)

__device__ __forceinline__ double block_max(double v, double* red, int t) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    double m = red[0];
    for (int k = 1; k < GT / 64; ++k) m = red[k] > m ? red[k] : m;
    __syncthreads();
    return m;
}

__global__ __launch_bounds__(GT) void greedy_kernel(const float* __restrict__ loc, const int* __restrict__ order, int N,
                                                    const double* __restrict__ hist1, const double* __restrict__ th_ptr, float cut_off,
                                                    double* kept /*[N][4]: written by lane 0, read by all after a barrier*/, int* keep_idx,
                                                    int* __restrict__ n_keep, unsigned int* fill_big, long fill_big_words) {
    __shared__ double red[GT / 64];
    __shared__ unsigned int fill[FILL_WORDS];
    __shared__ long long s_cnt;
    const int t = threadIdx.x;
    const double th = *th_ptr;
    int nk = 0;
    for (int oi = 0; oi < N; ++oi) {
        const int i = order[oi];
        const double p = loc[i * 9], cx = loc[i * 9 + 1], cy = loc[i * 9 + 2], w = loc[i * 9 + 3], h = loc[i * 9 + 4];
        if (p < (double)cut_off) break;
        if (hist1[i] < th) continue;                             // NaN threshold (no sample): never true, as in NumPy
        const double a0 = w * h;
        const double bx0 = cx - w / 2, bx1 = cx + w / 2, by0 = cy - h / 2, by1 = cy + h / 2;
        bool drop = false;
        if (nk > 0) {
            double m_iou = 0.0, m_inter = 0.0;
            for (int j = t; j < nk; j += GT) {
                const double dcx = kept[4 * j], dcy = kept[4 * j + 1], dw = kept[4 * j + 2], dh = kept[4 * j + 3];
                const double a1 = dw * dh;
                const double ix0 = fmax(bx0, dcx - dw / 2), iy0 = fmax(by0, dcy - dh / 2);
                const double ix1 = fmin(bx1, dcx + dw / 2), iy1 = fmin(by1, dcy + dh / 2);
                const double inter = fmax(ix1 - ix0, 0.0) * fmax(iy1 - iy0, 0.0);
                const double uni = a0 + a1 - inter;
                const double iou = uni > 0.0 ? inter / uni : 0.0;
                m_iou = iou > m_iou ? iou : m_iou;
                m_inter = inter > m_inter ? inter : m_inter;
            }
            m_iou = block_max(m_iou, red, t);
            m_inter = block_max(m_inter, red, t);
            if (m_iou > 0.5 || m_inter > a0 * 0.75) drop = true;
            else if (m_iou > 0.0) {
                // coverage rule (:602-613): cells of the candidate's int(w) x int(h) grid covered by kept boxes with iou > 0
                const long fw = (long)w, fh = (long)h;
                const long cells = fw * fh;
                if (cells > 0) {
                    const long words = (cells + 31) / 32;
                    unsigned int* bits = words <= FILL_WORDS ? fill : fill_big;
                    if (words > FILL_WORDS && words > fill_big_words) { if (t == 0) *n_keep = -1; return; }   // scratch too small: reported to the host
                    for (long k = t; k < words; k += GT) bits[k] = 0u;
                    __syncthreads();
                    for (int j = t; j < nk; j += GT) {
                        const double dcx = kept[4 * j], dcy = kept[4 * j + 1], dw = kept[4 * j + 2], dh = kept[4 * j + 3];
                        const double a1 = dw * dh;
                        const double ix0 = fmax(bx0, dcx - dw / 2), iy0 = fmax(by0, dcy - dh / 2);
                        const double ix1 = fmin(bx1, dcx + dw / 2), iy1 = fmin(by1, dcy + dh / 2);
                        const double inter = fmax(ix1 - ix0, 0.0) * fmax(iy1 - iy0, 0.0);
                        const double uni = a0 + a1 - inter;
                        const double iou = uni > 0.0 ? inter / uni : 0.0;
                        if (!(iou > 0.0)) continue;
                        long p1x = (long)(fmax(dcx - dw / 2, bx0) - bx0), p2x = (long)(fmin(dcx + dw / 2, bx1) - bx0) + 1;
                        long p1y = (long)(fmax(dcy - dh / 2, by0) - by0), p2y = (long)(fmin(dcy + dh / 2, by1) - by0) + 1;
                        if (p2x > fw) p2x = fw;
                        if (p2y > fh) p2y = fh;
                        for (long x = p1x; x < p2x; ++x)
                            for (long y = p1y; y < p2y; ++y) {
                                const long c = x * fh + y;
                                atomicOr(&bits[c >> 5], 1u << (c & 31));
                            }
                    }
                    __syncthreads();
                    long cnt = 0;
                    for (long k = t; k < words; k += GT) cnt += __popc(bits[k]);
                    if (t == 0) s_cnt = 0;                                // integer sum over the workgroup
                    __syncthreads();
                    atomicAdd(reinterpret_cast<unsigned long long*>(&s_cnt), (unsigned long long)cnt);
                    __syncthreads();
                    if ((double)s_cnt / (double)cells > 0.5) drop = true;
                    __syncthreads();
                }
            }
        }
        if (!drop) {
            if (t == 0) {
                kept[4 * nk] = cx; kept[4 * nk + 1] = cy; kept[4 * nk + 2] = w; kept[4 * nk + 3] = h;
                keep_idx[nk] = i;
            }
            ++nk;
            __syncthreads();                                     // the new entry is visible to the whole workgroup
        }
    }
    if (t == 0) *n_keep = nk;
}

// ---- separator filter (:636-643) and 3x3 maximum of the code maps (:644-650); one lane per kept box, order preserved ----
__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ loc, const int* __restrict__ keep_idx, const int* __restrict__ n_keep,
                                                     const float* __restrict__ seps, const float* __restrict__ codes, int mh, int mw, int scale,
                                                     float* __restrict__ out_loc, int* __restrict__ out_idx, int* __restrict__ out_n) {
    // a single workgroup keeps the output order with a prefix count
    __shared__ int s_base;
    __shared__ int s_scan[256];
    const int t = threadIdx.x;
    const int nk = *n_keep;
    if (t == 0) s_base = 0;
    __syncthreads();
    if (nk < 0) { if (t == 0) *out_n = -1; return; }
    for (int k0 = 0; k0 < nk; k0 += 256) {
        const int k = k0 + t;
        int i = -1;
        bool ok = false;
        if (k < nk) {
            i = keep_idx[k];
            const double cx = loc[i * 9 + 1], cy = loc[i * 9 + 2];
            const long x = (long)(cx / scale), y = (long)(cy / scale);
            ok = !(x >= 0 && x < mw && y >= 0 && y < mh && seps[y * mw + x] > 0.5f);
        }
        s_scan[t] = ok ? 1 : 0;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {                       // inclusive scan
            const int v = t >= o ? s_scan[t - o] : 0;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const int pos = s_base + s_scan[t] - 1;
        if (ok) {
            const double cx = loc[i * 9 + 1], cy = loc[i * 9 + 2];
            float r[9];
            for (int e = 0; e < 9; ++e) r[e] = loc[i * 9 + e];
            const long x = (long)(cx / scale), y = (long)(cy / scale);
            if (x >= 0 && x < mw && y >= 0 && y < mh) {
                const long x0 = max(0L, (long)(cx / scale - 1)), y0 = max(0L, (long)(cy / scale - 1));
                const long x1 = min((long)mw, (long)(cx / scale + 1) + 1), y1 = min((long)mh, (long)(cy / scale + 1) + 1);
                for (int c = 0; c < 4; ++c) {
                    const float* cm = codes + (long)c * mh * mw;
                    float m = r[5 + c];
                    bool any = false;
                    float mx = 0.f;
                    for (long yy = y0; yy < y1; ++yy)
                        for (long xx = x0; xx < x1; ++xx) {
                            const float v = cm[yy * mw + xx];
                            mx = any ? (v > mx ? v : mx) : v;
                            any = true;
                        }
                    if (any) m = mx > m ? mx : m;
                    r[5 + c] = m;
                }
            }
            for (int e = 0; e < 9; ++e) out_loc[(long)pos * 9 + e] = r[e];
            out_idx[pos] = i;
        }
        __syncthreads();
        if (t == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (t == 0) *out_n = s_base;
}

}  // namespace

hipError_t launch_box_hists(const float* loc, int N, const float* page, int PH, int PW, float cut_off, double* out, hipStream_t s) {
    hipLaunchKernelGGL(box_hist_kernel, dim3(N, 2), dim3(256), 0, s, loc, N, page, PH, PW, cut_off, out);
    return hipGetLastError();
}

hipError_t launch_greedy(const float* loc, const int* order, int N, const double* hist1, const double* th, float cut_off, double* kept,
                         int* keep_idx, int* n_keep, unsigned int* fill_big, long fill_big_words, const float* seps, const float* codes, int mh,
                         int mw, int scale, float* out_loc, int* out_idx, int* out_n, hipStream_t s) {
    hipLaunchKernelGGL(greedy_kernel, dim3(1), dim3(GT), 0, s, loc, order, N, hist1, th, cut_off, kept, keep_idx, n_keep, fill_big, fill_big_words);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, s, loc, keep_idx, n_keep, seps, codes, mh, mw, scale, out_loc, out_idx, out_n);
    return hipGetLastError();
}
