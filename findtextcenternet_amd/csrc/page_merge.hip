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
// One WAVE per histogram (round 4; one thread walked the 256 bins through every 2-means iteration before: ~100 us per box, 5 ms for the
// 20 k boxes of a dense page): lane = 4 bins, the four sums of an iteration meet by xor shuffles.  The sums are integers (exact in any
// order) and every lane ends up with the same centres, so control flow is wave-uniform and the result is the serial loop's.
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ double cluster_dist(const unsigned int* hist, int lane) {
    long long h[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = hist[lane * 4 + e];
    long long tot = 0, sv = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { tot += h[e]; sv += h[e] * (lane * 4 + e); }
    tot = wave_sum_ll(tot);
    sv = wave_sum_ll(sv);
    if (tot == 0) return 0.0;
    const int cut = (int)((double)sv / (double)tot + 0.5);
    long long s1 = 0, s2 = 0, v1 = 0, v2 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = lane * 4 + e;
        if (i < cut) { s1 += h[e]; v1 += h[e] * i; }
        else { s2 += h[e]; v2 += h[e] * i; }
    }
    s1 = wave_sum_ll(s1); s2 = wave_sum_ll(s2); v1 = wave_sum_ll(v1); v2 = wave_sum_ll(v2);
    if (s1 == 0 || s2 == 0) return 0.0;
    double k1 = (double)v1 / (double)s1, k2 = (double)v2 / (double)s2;
    double prev = 256.0, cur = fabs(k1 - k2);
    while (prev != cur) {
        prev = cur;
        s1 = s2 = v1 = v2 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = lane * 4 + e;
            const bool near1 = fabs((double)i - k1) < fabs((double)i - k2);
            if (near1) { s1 += h[e]; v1 += h[e] * i; }
            else { s2 += h[e]; v2 += h[e] * i; }
        }
        s1 = wave_sum_ll(s1); s2 = wave_sum_ll(s2); v1 = wave_sum_ll(v1); v2 = wave_sum_ll(v2);
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
    if (t < 192) {                                                // waves 0..2: one colour channel each
        const double d = cluster_dist(hist[t >> 6], t & 63);
        if ((t & 63) == 0) res[t >> 6] = d;
    }
    __syncthreads();
    if (t == 0) {
        double best = -1.0;
        for (int c = 0; c < 3; ++c) best = res[c] > best ? res[c] : best;
        out[(long)variant * N + i] = best;
    }
}

// ---- variants of the selection -------------------------------------------------------------------------------------
// demo = 0: OCR_Processer.run_detector (process_ocr_base.py:559-650), what the rest of this file describes.
// demo = 1: the demo script's eval() (/root/reference/test_image1_torch.py:152-240): NO contrast filter; the coverage image is filled with
//           different offsets (p2x WITHOUT the +1, p1y WITH a +1: :196-200); and rows [seed_start, N) are the boxes a coarse first pass found
//           on the page shrunk by `seed_scale` (twopass, :313-332: locations0[:,1:] * s) -- their columns 1..8 are multiplied by seed_scale
//           in float64 before anything looks at them (the fp32 rows hold the unscaled values, so that the products are NumPy's).
struct PmVar { int demo; int seed_start; double seed_scale; };
__device__ __forceinline__ double pm_col(const float* __restrict__ loc, int i, int c, const PmVar& v) {
    const double x = loc[(long)i * 9 + c];
    return (c > 0 && i >= v.seed_start) ? x * v.seed_scale : x;
}
// coverage run of a kept box (edges d[0..3] = x0, x1, y0, y1) inside the candidate (edges c[0..3]), Python slice semantics on an fw x fh image
__device__ __forceinline__ void pm_cover(const double* d, const double* c, long fw, long fh, int demo, long* p1x, long* p2x, long* p1y, long* p2y) {
    long a = (long)(fmax(d[0], c[0]) - c[0]), b = (long)(fmin(d[1], c[1]) - c[0]) + (demo ? 0 : 1);
    long e = (long)(fmax(d[2], c[2]) - c[2]) + (demo ? 1 : 0), f = (long)(fmin(d[3], c[3]) - c[2]) + 1;
    if (a < 0) a = 0;                                               // (cannot happen: d >= c edge after fmax; kept for the cast of a NaN)
    if (e < 0) e = 0;
    if (b > fw) b = fw;
    if (f > fh) f = fh;
    *p1x = a; *p2x = b; *p1y = e; *p2y = f;
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
                                                    int* __restrict__ n_keep, unsigned int* fill_big, long fill_big_words, const int* use_seq, const PmVar var) {
    __shared__ double red[GT / 64];
    __shared__ unsigned int fill[FILL_WORDS];
    __shared__ long long s_cnt;
    if (use_seq && !*use_seq) return;                            // (round 4) the parallel selection handled this page
    const int t = threadIdx.x;
    const double th = var.demo ? 0.0 : *th_ptr;
    int nk = 0;
    for (int oi = 0; oi < N; ++oi) {
        const int i = order[oi];
        const double p = loc[i * 9], cx = pm_col(loc, i, 1, var), cy = pm_col(loc, i, 2, var), w = pm_col(loc, i, 3, var), h = pm_col(loc, i, 4, var);
        if (p < (double)cut_off) break;
        if (!var.demo && hist1[i] < th) continue;                // NaN threshold (no sample): never true, as in NumPy
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
                        const double de[4] = {dcx - dw / 2, dcx + dw / 2, dcy - dh / 2, dcy + dh / 2}, ce[4] = {bx0, bx1, by0, by1};
                        long p1x, p2x, p1y, p2y;
                        pm_cover(de, ce, fw, fh, var.demo, &p1x, &p2x, &p1y, &p2y);
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

// ------------------------------------------------------------------------------------------------------------------
// Round 4: the same selection, parallel.  The greedy pass above costs ~1 us per candidate (two workgroup reductions over the
// kept list per box): 53 ms for the 56 k candidates of a random-init A4 page.  A box's fate depends only on the KEPT boxes
// among its earlier-ranked OVERLAPPING candidates (IoU > 0: max_iou, max_inter and the coverage image all come from those), so:
//   pm_prep      rank-ordered edge table (x0, x1, y0, y1, w, h in float64, exactly the expressions of the greedy kernel),
//                eligibility (p >= cut_off, contrast filter) -> status 0 (undecided) | 2 (never kept);
//   pm_pairs     2-D tiled all-pairs pass, twice: count the earlier eligible neighbours with IoU > 0 of every candidate, then
//                (after pm_scan's prefix sum) fill the neighbour lists;
//   pm_resolve   persistent waves take candidates IN RANK ORDER from a ticket counter; a wave waits (spins on the status words)
//                until every earlier neighbour is decided, applies the three rules to the kept ones (coverage image: a bit image
//                in the wave's LDS share, filled by row runs, popcount) and publishes 1 (kept) | 2 (dropped).  A candidate only
//                waits for lower ranks, whose tickets were drawn earlier by waves that are running: no deadlock whatever the
//                dispatch order; the dependency chains are as long as the overlap graph is deep (tens of boxes), not N.
//   pm_compact   kept ranks -> keep_idx in rank order (what the greedy kernel produced), then finish_kernel as before.
// Every comparison is the greedy kernel's own float64 expression, so the result is bit-identical (tests/test_gpu_page.py).  If the
// neighbour lists do not fit the scratch the caller gave, a device flag routes the page through the sequential kernel instead.
// ------------------------------------------------------------------------------------------------------------------
struct PmHdr { int n_keep, ticket, use_seq, big_lock, total_edges, stall_r, stall_j, stall_n; };      // stall_*: the first wait that ran into PM_SPIN_LIMIT
constexpr int PM_T = 256;                  // candidates per tile of the all-pairs pass
constexpr int PM_SPIN_LIMIT = 1 << 19;      // polls (~130 cycles apart) a wave waits for ONE neighbour before it hands the page to the sequential kernel
constexpr int PM_WAVE_WORDS = 2048;        // LDS coverage image per wave: 65536 cells (larger boxes: one shared global image behind a lock)

struct PmPair { double inter, iou; };
// the greedy kernel's expressions (candidate = the later rank: a0; kept = the earlier one)
__device__ __forceinline__ PmPair pm_pair(const double* c, const double* k) {
    const double a0 = c[4] * c[5], a1 = k[4] * k[5];
    const double ix0 = fmax(c[0], k[0]), iy0 = fmax(c[2], k[2]);
    const double ix1 = fmin(c[1], k[1]), iy1 = fmin(c[3], k[3]);
    const double inter = fmax(ix1 - ix0, 0.0) * fmax(iy1 - iy0, 0.0);
    const double uni = a0 + a1 - inter;
    return {inter, uni > 0.0 ? inter / uni : 0.0};
}

__global__ __launch_bounds__(256) void pm_prep_kernel(const float* __restrict__ loc, const int* __restrict__ order, int N, const double* __restrict__ hist1,
                                                      const double* __restrict__ th_ptr, float cut_off, double* __restrict__ rb, int* __restrict__ status,
                                                      int* __restrict__ cnt, PmHdr* hdr, int force_seq, const PmVar var) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) { hdr->n_keep = 0; hdr->ticket = 0; hdr->use_seq = force_seq; hdr->big_lock = 0; hdr->total_edges = 0; hdr->stall_r = hdr->stall_j = -1; hdr->stall_n = 0; }
    if (r > N) return;
    if (r == N) { cnt[N] = 0; return; }
    const int i = order[r];
    const double p = loc[i * 9], cx = pm_col(loc, i, 1, var), cy = pm_col(loc, i, 2, var), w = pm_col(loc, i, 3, var), h = pm_col(loc, i, 4, var);
    const bool elig = p >= (double)cut_off && (var.demo || !(hist1[i] < *th_ptr));
    double* o = rb + (long)r * 6;
    o[0] = cx - w / 2; o[1] = cx + w / 2; o[2] = cy - h / 2; o[3] = cy + h / 2; o[4] = w; o[5] = h;
    status[r] = elig ? 0 : 2;
    cnt[r] = 0;
}

// FILL = false: cnt[r] += neighbours of r inside tile column blockIdx.x;  FILL = true: nbr[cursor[r]++] = j
template <bool FILL>
__global__ __launch_bounds__(PM_T) void pm_pairs_kernel(const double* __restrict__ rb, const int* __restrict__ status0, int N, int* cnt_or_cursor,
                                                        int* __restrict__ nbr, const PmHdr* hdr) {
    const int tj = blockIdx.x, tr = blockIdx.y;
    if (tj > tr) return;
    if (FILL && hdr->use_seq) return;
    __shared__ double sx0[PM_T], sx1[PM_T], sy0[PM_T], sy1[PM_T], sw[PM_T], sh[PM_T];
    __shared__ int sel[PM_T];
    const int t = threadIdx.x;
    const int j0 = tj * PM_T, r = tr * PM_T + t;
    {
        const int j = j0 + t;
        const bool ok = j < N;
        const double* b = rb + (long)(ok ? j : 0) * 6;
        sx0[t] = b[0]; sx1[t] = b[1]; sy0[t] = b[2]; sy1[t] = b[3]; sw[t] = b[4]; sh[t] = b[5];
        sel[t] = ok && status0[j] != 2;                 // eligibility: status is 0 | 2 until pm_resolve runs
    }
    __syncthreads();
    if (r >= N || status0[r] == 2) return;
    double c[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) c[e] = rb[(long)r * 6 + e];
    const int jn = min(PM_T, r - j0);                    // only earlier ranks
    int n = 0;
    for (int k = 0; k < jn; ++k) {
        if (!sel[k]) continue;
        if (!(sx0[k] < c[1] && c[0] < sx1[k] && sy0[k] < c[3] && c[2] < sy1[k])) continue;       // (necessary for inter > 0)
        const double kb[6] = {sx0[k], sx1[k], sy0[k], sy1[k], sw[k], sh[k]};
        if (pm_pair(c, kb).iou > 0.0) {
            if (FILL) nbr[atomicAdd(&cnt_or_cursor[r], 1)] = j0 + k;
            else ++n;
        }
    }
    if (!FILL && n) atomicAdd(&cnt_or_cursor[r], n);
}

// exclusive prefix sum of cnt[0..N] in place -> offsets (cnt[N] = total); cursor = a copy for the fill pass
__global__ __launch_bounds__(1024) void pm_scan_kernel(int* cnt, int* cursor, int N, long cap, PmHdr* hdr) {
    __shared__ int wsum[16];
    __shared__ long long s_base;           // 64-bit running total: 2^20 candidates can have more than 2^31 overlapping pairs, and a wrapped
    __shared__ int s_over;                 // 32-bit total could pass the `total > cap` test with wrapped offsets in between (round-4 advisor finding)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) { s_base = 0; s_over = 0; }
    __syncthreads();
    const long long lim = cap < 0x7fffffffL ? (long long)cap : 0x7fffffffLL;      // offsets are ints: more edges than that go to the sequential kernel
    for (int k0 = 0; k0 <= N; k0 += 1024) {
        const int k = k0 + t;
        const int v = k < N ? cnt[k] : 0;                       // <= N <= 2^20: a 1024-row chunk sums to < 2^31
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        long long woff = 0;
        for (int q = 0; q < wave; ++q) woff += wsum[q];
        const long long base = s_base;
        const long long excl = base + woff + inc - v;
        if (k <= N) { const int e = (int)(excl < lim ? excl : lim); cnt[k] = e; cursor[k] = e; }      // saturated: never a negative or wrapped offset
        __syncthreads();
        if (t == 1023) { s_base = base + woff + inc; if (base + woff + inc > lim) s_over = 1; }
        __syncthreads();
    }
    if (t == 0) {
        const long long total = s_base;
        hdr->total_edges = (int)(total < 0x7fffffffLL ? total : 0x7fffffffLL);
        if (total > lim || s_over) hdr->use_seq = 1;
    }
}

// bits [start, end) of a bit image, word by word
template <typename OrFn>
__device__ __forceinline__ void pm_set_run(long start, long end, OrFn&& orf) {
    if (end <= start) return;
    const long w0 = start >> 5, w1 = (end - 1) >> 5;
    const unsigned lo = 0xffffffffu << (start & 31), hi = 0xffffffffu >> (31 - (int)((end - 1) & 31));
    if (w0 == w1) { orf(w0, lo & hi); return; }
    orf(w0, lo);
    for (long w = w0 + 1; w < w1; ++w) orf(w, 0xffffffffu);
    orf(w1, hi);
}

__global__ __launch_bounds__(256) void pm_resolve_kernel(const double* __restrict__ rb, int* status, const int* __restrict__ off, const int* __restrict__ nbr,
                                                         int N, PmHdr* hdr, unsigned int* big_bits, long big_words, int demo) {
    __shared__ unsigned int lbits[4][PM_WAVE_WORDS];
    if (hdr->use_seq) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int* mybits = lbits[wave];
    for (;;) {
        int r = 0;
        if (lane == 0) r = atomicAdd(&hdr->ticket, 1);
        r = __shfl(r, 0, 64);
        if (r >= N) break;
        if ((r & 63) == 0 && __hip_atomic_load(&hdr->use_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (status[r] == 2) continue;                                // not eligible (set by pm_prep; an eligible box is 0 until THIS wave decides it)
        const int beg = off[r], end = off[r + 1];
        double c[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) c[e] = rb[(long)r * 6 + e];
        const double a0 = c[4] * c[5];
        bool drop = false;
        int nkept = 0;
        for (int base = beg; base < end && !drop; base += 64) {
            const int q = base + lane;
            const int j = q < end ? nbr[q] : -1;
            // everything that does not depend on the neighbour's fate first: its edges and whether it would suppress this box outright
            bool hard = false;
            if (j >= 0) {
                double kb[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) kb[e] = rb[(long)j * 6 + e];
                const PmPair pr = pm_pair(c, kb);
                hard = pr.iou > 0.5 || pr.inter > a0 * 0.75;
            }
            // Poll the neighbours' status words (relaxed: the word is the only thing communicated).  ONE kept `hard` neighbour settles the
            // box (dropped) whatever the others turn out to be; otherwise every neighbour has to be decided.  Bounded: a wait that does
            // not end (it cannot, by the ticket order -- but a hang would cost the whole process) flips the device flag instead, and the
            // sequential kernel launched behind this one redoes the page.
            int st = j >= 0 ? 0 : 2;
            for (int polls = 0;; ++polls) {
                if (st == 0) st = __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__ballot(st == 1 && hard)) { drop = true; break; }
                if (!__ballot(st == 0)) break;
                __builtin_amdgcn_s_sleep(1);
                if (polls > PM_SPIN_LIMIT || ((polls & 1023) == 1023 && __hip_atomic_load(&hdr->use_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    if (lane == 0) {
                        if (atomicAdd(&hdr->stall_n, 1) == 0) { hdr->stall_r = r; hdr->stall_j = polls; }
                        atomicExch(&hdr->use_seq, 1);
                    }
                    return;
                }
            }
            nkept += __popcll(__ballot(st == 1));
        }
        if (!drop && nkept > 0) {
            // coverage rule: cells of the candidate's int(w) x int(h) grid covered by the kept neighbours (all have IoU > 0)
            const long fw = (long)c[4], fh = (long)c[5];
            const long cells = fw * fh;
            if (cells > 0) {
                const long words = (cells + 31) / 32;
                const bool big = words > PM_WAVE_WORDS;
                if (big && words > big_words) {                      // scratch too small for this box: reported to the host as before
                    if (lane == 0) hdr->n_keep = -1;
                    drop = true;
                } else {
                    unsigned int* bits = big ? big_bits : mybits;
                    if (big) {
                        if (lane == 0) {
                            int polls = 0;
                            while (atomicCAS(&hdr->big_lock, 0, 1) != 0) {
                                __builtin_amdgcn_s_sleep(8);
                                if (++polls > PM_SPIN_LIMIT) { atomicExch(&hdr->use_seq, 1); break; }      // (as above: never a hang)
                            }
                        }
                        __threadfence();
                    }
                    for (long k = lane; k < words; k += 64) bits[k] = 0u;
                    if (big) __threadfence();
                    for (int base = beg; base < end; base += 64) {
                        const int q = base + lane;
                        const int j = q < end ? nbr[q] : -1;
                        const bool kept = j >= 0 && __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1;
                        double kb[4] = {0, 0, 0, 0};
                        if (kept) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) kb[e] = rb[(long)j * 6 + e];
                        }
                        unsigned long long m = __ballot(kept);
                        while (m) {
                            const int src = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            double d[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[e] = __shfl(kb[e], src, 64);
                            long p1x, p2x, p1y, p2y;
                            pm_cover(d, c, fw, fh, demo, &p1x, &p2x, &p1y, &p2y);
                            for (long x = p1x + lane; x < p2x; x += 64)
                                pm_set_run(x * fh + p1y, x * fh + p2y, [&](long w, unsigned v) { atomicOr(&bits[w], v); });
                        }
                    }
                    if (big) __threadfence();
                    long cntb = 0;
                    for (long k = lane; k < words; k += 64) cntb += __popc(big ? __hip_atomic_load(&bits[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bits[k]);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) cntb += __shfl_xor(cntb, o, 64);
                    if (big) {
                        __threadfence();
                        if (lane == 0) atomicExch(&hdr->big_lock, 0);
                    }
                    if ((double)cntb / (double)cells > 0.5) drop = true;
                }
            }
        }
        if (lane == 0) __hip_atomic_store(status + r, drop ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// kept ranks, in rank order -> keep_idx (source rows), n_keep
__global__ __launch_bounds__(1024) void pm_compact_kernel(const int* __restrict__ status, const int* __restrict__ order, int N, int* keep_idx, PmHdr* hdr) {
    __shared__ int wsum[16];
    __shared__ int s_base;
    if (hdr->use_seq) return;                                          // the sequential kernel wrote keep_idx / n_keep itself
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < N; k0 += 1024) {
        const int k = k0 + t;
        const int v = (k < N && status[k] == 1) ? 1 : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < wave; ++q) woff += wsum[q];
        const int base = s_base;
        if (v) keep_idx[base + woff + inc - 1] = order[k];
        __syncthreads();
        if (t == 1023) s_base = base + woff + inc;
        __syncthreads();
    }
    if (t == 0 && hdr->n_keep >= 0) hdr->n_keep = s_base;
}

// ---- in-tree replacement of the two library sorts in front of the selection (round 4) ----
// order = stable argsort of -p (ties: lower row first): rank by counting over 64-bit keys (score bits, ~row) -- all keys distinct
__device__ __forceinline__ unsigned long long pm_key(float p, int i) {
    unsigned int u = __float_as_uint(p);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                    // order-preserving map of a float to an unsigned
    return ((unsigned long long)u << 32) | (unsigned int)(~i);
}
// Rows with p >= cut_off first (the only ones the selection looks at: it stops at the first score below the cut-off), compacted by one
// workgroup-wide scan: their 64-bit keys and contrasts become dense arrays, so that the rank and median kernels walk M entries, not the N rows of
// the record blocks (an A4 page: 56 k of 143 k).  order[M + k] = the k-th row below the cut-off, in row order.
struct PoHdr { int M, pad[3]; };
__global__ __launch_bounds__(1024) void pm_select_kernel(const float* __restrict__ loc, int N, const double* __restrict__ hist0, float cut_off,
                                                         unsigned long long* __restrict__ keys, double* __restrict__ hv, int* __restrict__ order, PoHdr* hdr) {
    __shared__ int wsum[16];
    __shared__ int s_base, s_M;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_M = 0;
    __syncthreads();
    int m = 0;
    for (int i = t; i < N; i += 1024) m += (double)loc[(long)i * 9] >= (double)cut_off ? 1 : 0;
    atomicAdd(&s_M, m);
    __syncthreads();
    const int M = s_M;
    if (t == 0) { s_base = 0; hdr->M = M; }
    __syncthreads();
    for (int k0 = 0; k0 < N; k0 += 1024) {
        const int k = k0 + t;
        const float p = k < N ? loc[(long)k * 9] : 0.f;
        const int v = (k < N && (double)p >= (double)cut_off) ? 1 : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < wave; ++q) woff += wsum[q];
        const int base = s_base;
        const int pos = base + woff + inc - v;                       // eligible rows before this one
        if (k < N) {
            if (v) { keys[pos] = pm_key(p, k); hv[pos] = hist0[k]; }
            else order[M + (k - pos)] = k;
        }
        __syncthreads();
        if (t == 1023) s_base = base + woff + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pm_rank_kernel(const unsigned long long* __restrict__ keys, const PoHdr* __restrict__ hdr, int* __restrict__ order) {
    __shared__ unsigned long long sk[1024];
    const int M = hdr->M;
    if ((int)blockIdx.x * 256 >= M) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long mine = i < M ? keys[i] : 0ull;
    int rank = 0;
    for (int j0 = 0; j0 < M; j0 += 1024) {
        for (int q = threadIdx.x; q < 1024; q += 256) sk[q] = j0 + q < M ? keys[j0 + q] : 0ull;
        __syncthreads();
        const int jn = min(1024, M - j0);
        for (int q = 0; q < jn; ++q) rank += sk[q] > mine ? 1 : 0;
        __syncthreads();
    }
    if (i < M) order[rank] = (int)~(unsigned int)(mine & 0xffffffffull);          // the row index sits (complemented) in the key's low word
}

// threshold = median(contrasts of the rows with p >= cut_off) / 5 (NaN without rows): the two middle order statistics by an MSB-first
// radix select over the float64 bit patterns (contrasts are >= 0: the patterns order like the values)
__global__ __launch_bounds__(1024) void pm_median_kernel(const double* __restrict__ hv, const PoHdr* __restrict__ hdr, double* th_out) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ long long s_k;
    const int t = threadIdx.x;
    const int M = hdr->M;
    if (M == 0) { if (t == 0) *th_out = __longlong_as_double(0x7ff8000000000000ll); return; }
    double v[2];
    for (int which = 0; which < 2; ++which) {
        if (t == 0) { s_prefix = 0ull; s_k = which == 0 ? (M - 1) / 2 : M / 2; }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (t < 256) hist[t] = 0u;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            for (int i = t; i < M; i += 1024) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(hv[i]);
                if (shift == 56 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
            }
            __syncthreads();
            if (t == 0) {
                long long k = s_k;
                int d = 0;
                for (; d < 255; ++d) { if (k < (long long)hist[d]) break; k -= hist[d]; }
                s_k = k;
                s_prefix = (prefix << 8) | (unsigned long long)d;
            }
            __syncthreads();
        }
        v[which] = __longlong_as_double((long long)s_prefix);
        __syncthreads();
    }
    if (t == 0) *th_out = (v[0] + v[1]) / 2 / 5;
}

// ---- separator filter (:636-643) and 3x3 maximum of the code maps (:644-650); one lane per kept box, order preserved ----
__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ loc, const int* __restrict__ keep_idx, const int* __restrict__ n_keep,
                                                     const float* __restrict__ seps, const float* __restrict__ codes, int mh, int mw, int scale,
                                                     float* __restrict__ out_loc, int* __restrict__ out_idx, int* __restrict__ out_n, const PmVar var,
                                                     float* __restrict__ out_cmax) {
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
            const double cx = pm_col(loc, i, 1, var), cy = pm_col(loc, i, 2, var);
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
            const double cx = pm_col(loc, i, 1, var), cy = pm_col(loc, i, 2, var);
            float r[9];
            for (int e = 0; e < 9; ++e) r[e] = loc[i * 9 + e];
            float cm4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};    // out_cmax: the 3x3 maxima themselves (-inf: centre outside the page, no update)
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
                    if (any) { m = mx > m ? mx : m; cm4[c] = mx; }
                    r[5 + c] = m;
                }
            }
            for (int e = 0; e < 9; ++e) out_loc[(long)pos * 9 + e] = r[e];
            if (out_cmax)
                for (int c = 0; c < 4; ++c) out_cmax[(long)pos * 4 + c] = cm4[c];
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

// scratch layout of ftc_page_merge (ftc_api.hip passes the pieces): hdr | kept [N][4] f64 | keep_idx [N] | rb [N][6] f64 | status [N] | cnt [N+1] | cursor [N+1] |
// nbr [edge_cap] | coverage bit image of the page
hipError_t launch_greedy(const float* loc, const int* order, int N, const double* hist1, const double* th, float cut_off, double* kept,
                         int* keep_idx, int* hdr_, double* rb, int* status, int* cnt, int* cursor, int* nbr, long edge_cap, unsigned int* fill_big,
                         long fill_big_words, int force_seq, const float* seps, const float* codes, int mh, int mw, int scale, float* out_loc,
                         int* out_idx, int* out_n, int variant, int seed_start, double seed_scale, float* out_cmax, hipStream_t s) {
    PmHdr* hdr = reinterpret_cast<PmHdr*>(hdr_);
    const PmVar var{variant ? 1 : 0, (variant && seed_start >= 0) ? seed_start : N, seed_scale};
    const int T = (N + PM_T - 1) / PM_T;
    hipLaunchKernelGGL(pm_prep_kernel, dim3((N + 256) / 256), dim3(256), 0, s, loc, order, N, hist1, th, cut_off, rb, status, cnt, hdr, force_seq, var);
    hipLaunchKernelGGL(pm_pairs_kernel<false>, dim3(T, T), dim3(PM_T), 0, s, rb, status, N, cnt, nbr, hdr);
    hipLaunchKernelGGL(pm_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, cursor, N, edge_cap, hdr);
    hipLaunchKernelGGL(pm_pairs_kernel<true>, dim3(T, T), dim3(PM_T), 0, s, rb, status, N, cursor, nbr, hdr);
    int ncu = 256;
    hipLaunchKernelGGL(pm_resolve_kernel, dim3(ncu * 2), dim3(256), 0, s, rb, status, cnt, nbr, N, hdr, fill_big, fill_big_words, var.demo);
    hipLaunchKernelGGL(greedy_kernel, dim3(1), dim3(GT), 0, s, loc, order, N, hist1, th, cut_off, kept, keep_idx, &hdr->n_keep, fill_big, fill_big_words,
                       (const int*)&hdr->use_seq, var);
    hipLaunchKernelGGL(pm_compact_kernel, dim3(1), dim3(1024), 0, s, status, order, N, keep_idx, hdr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, s, loc, keep_idx, (const int*)&hdr->n_keep, seps, codes, mh, mw, scale, out_loc, out_idx, out_n, var, out_cmax);
    return hipGetLastError();
}

hipError_t launch_page_order(const float* loc, int N, const double* hist0, float cut_off, int* order, double* th, void* scratch, hipStream_t s) {
    PoHdr* hdr = static_cast<PoHdr*>(scratch);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch) + 256);
    double* hv = reinterpret_cast<double*>(keys + N);
    hipLaunchKernelGGL(pm_select_kernel, dim3(1), dim3(1024), 0, s, loc, N, hist0, cut_off, keys, hv, order, hdr);
    hipLaunchKernelGGL(pm_rank_kernel, dim3((N + 255) / 256), dim3(256), 0, s, keys, hdr, order);
    hipLaunchKernelGGL(pm_median_kernel, dim3(1), dim3(1024), 0, s, hv, hdr, th);
    return hipGetLastError();
}
