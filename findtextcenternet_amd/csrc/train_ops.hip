// Validation / training-step adjuncts of the detector (BASELINE config 5), forward only:
//
//   ftc_topk_mask      TextDetectorModel.get_fmask      /root/reference/models/detector.py:270-281
//   ftc_mask_compact   `features[fmask]` index list      /root/reference/models/detector.py:265-266
//   ftc_gather_rows    the boolean-mask gather itself    (rows of the NHWC feature map -> decoder input rows)
//   ftc_detector_losses / ftc_id_losses   loss_function /root/reference/loss_func.py:94-177 (heatmap_loss :74-92)
//   ftc_cov_weighting_step                CoVWeightingLoss.forward /root/reference/loss_func.py:24-72
//
// Everything here is bandwidth-trivial (a few MB per step); the kernels are written for determinism: integer histograms,
// fixed-order two-stage reductions (per-workgroup partials summed in index order in float64), no floating-point atomics.
#include <cstring>

#include "ftc_common.h"

namespace {

__device__ __forceinline__ uint32_t orderable(float v) {          // larger float <=> larger unsigned key
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ------------------------------------------------------------------------------------------------------------------------
// Top-k selection mask: mask[i] = 1 for the k largest values (ties at the k-th value: lowest index first, which is what the
// reference's stable CPU sort yields), plus the selected indices in ascending order (the row order of `x[mask]`).
// ONE 1024-thread workgroup (n <= a few 10^5): 4 radix passes of 8 bits over LDS histograms find the k-th key exactly, then
// every thread walks ITS contiguous chunk of indices, so a block-wide exclusive scan of the per-thread counts gives both the
// tie cut and the compaction offsets.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* lds /* [1024 + 1] */) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = t >= off ? lds[t - off] : 0;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const int incl = lds[t];
    if (t == 1023) lds[1024] = incl;
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(1024) void topk_mask_kernel(const float* __restrict__ vals, long n, long k, unsigned char* __restrict__ mask,
                                                         int32_t* __restrict__ sel_index, int32_t* __restrict__ count_out) {
    __shared__ int hist[256];
    __shared__ int scan[1025];
    __shared__ uint32_t s_prefix;
    __shared__ long s_need;
    const int t = threadIdx.x;
    if (k > n) k = n;
    if (t == 0) { s_prefix = 0u; s_need = k; }
    __syncthreads();
    for (int pass = 0; pass < 4 && k > 0; ++pass) {
        const int shift = 24 - 8 * pass;
        if (t < 256) hist[t] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        for (long i = t; i < n; i += 1024) {
            const uint32_t key = orderable(vals[i]);
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
        __syncthreads();
        if (t == 0) {
            long need = s_need;
            int b = 255;
            for (; b > 0; --b) {                 // buckets in DEscending key order
                if (hist[b] >= need) break;
                need -= hist[b];
            }
            s_prefix = prefix | ((uint32_t)b << shift);
            s_need = need;                       // how many of the elements that match the new prefix are still needed
        }
        __syncthreads();
    }
    const uint32_t T = s_prefix;                 // the k-th largest key
    const long need_eq = s_need;                 // how many elements equal to it belong to the top k
    const long chunk = (n + 1023) / 1024;
    const long lo = (long)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    int n_eq = 0;
    for (long i = lo; i < hi; ++i) n_eq += (k > 0 && orderable(vals[i]) == T) ? 1 : 0;
    const int eq_before = block_exclusive_scan_1024(n_eq, scan);
    __syncthreads();
    int n_sel = 0, e = eq_before;
    for (long i = lo; i < hi; ++i) {
        const uint32_t key = orderable(vals[i]);
        bool s = k > 0 && key > T;
        if (k > 0 && key == T) { s = e < need_eq; ++e; }
        mask[i] = s ? 1 : 0;
        n_sel += s ? 1 : 0;
    }
    const int sel_before = block_exclusive_scan_1024(n_sel, scan);
    if (sel_index) {
        int o = sel_before;
        for (long i = lo; i < hi; ++i)
            if (mask[i]) sel_index[o++] = (int32_t)i;
    }
    if (t == 1023 && count_out) *count_out = sel_before + n_sel;
}

__global__ __launch_bounds__(1024) void mask_compact_kernel(const unsigned char* __restrict__ mask, long n, int32_t* __restrict__ sel_index,
                                                            long cap, int32_t* __restrict__ count_out) {
    __shared__ int scan[1025];
    const int t = threadIdx.x;
    const long chunk = (n + 1023) / 1024;
    const long lo = (long)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    int c = 0;
    for (long i = lo; i < hi; ++i) c += mask[i] ? 1 : 0;
    int o = block_exclusive_scan_1024(c, scan);
    for (long i = lo; i < hi; ++i)
        if (mask[i]) { if (o < cap) sel_index[o] = (int32_t)i; ++o; }
    if (t == 1023) *count_out = o;
}

// rows[i][0..C) = feat[sel_index[i]][0..C), rows[i][C..Cpad) = 0; 16-byte lanes (C % 4 == 0).
template <typename OutT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ feat, const int32_t* __restrict__ sel_index,
                                                          const int32_t* __restrict__ count, long cap, int C, int Cpad, OutT* __restrict__ rows) {
    const long n = count ? (*count < cap ? *count : cap) : cap;
    const int Q = Cpad / 4;
    const long total = cap * Q;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / Q;
        const int c = (int)(idx - r * Q) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < n && c < C) v = *reinterpret_cast<const f32x4*>(feat + (long)sel_index[r] * C + c);
        store4<OutT>(rows + r * Cpad + c, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// loss_function, map part (loss_func.py:94-133): one pass over the B*h*w pixels, 12 partial sums per workgroup.
//   0 keymap focal sum        heatmap_loss :74-92 (alpha 2, beta 4, pos_th 1)       -> mean * 10
//   1 size numerator          sum over keylabel > 0.85 of (huber(x) + huber(y)) * weight1      2 weight1 sum
//   3 textline BCE sum   4 separator BCE sum   5..8 weighted code BCE sums (weight = 1 + bit*w2 + w2)
// Strided views: heat (b, c, y, x) strides in elements for the NINE reference channels, so both the NHWC block the detector
// writes and a plain NCHW tensor are accepted.
// ------------------------------------------------------------------------------------------------------------------------
struct MapLossP {
    const float* heat; long hs_b, hs_c, hs_y, hs_x;
    const float* label;       // [B,5,h,w] contiguous
    const int32_t* idmap;     // [B,2,h,w] contiguous
    int B, h, w;
};
constexpr int NMAP = 9;

__device__ __forceinline__ float bce_logits(float x, float y) {          // max(x,0) - x*y + log1p(exp(-|x|))
    return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float huber1(float a, float b) {
    const float d = fabsf(a - b);
    return d < 1.0f ? 0.5f * d * d : d - 0.5f;
}
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // torch threshold = 20

__global__ __launch_bounds__(256) void map_loss_kernel(MapLossP p, double* __restrict__ partial /* [gridDim.x][NMAP] */) {
    __shared__ float red[NMAP][256];
    const long hw = (long)p.h * p.w, n = (long)p.B * hw;
    float acc[NMAP];
#pragma unroll
    for (int j = 0; j < NMAP; ++j) acc[j] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / hw, r = i - b * hw;
        const int y = (int)(r / p.w), x = (int)(r - (long)y * p.w);
        const float* hp = p.heat + b * p.hs_b + y * p.hs_y + x * p.hs_x;
        const float* lp = p.label + b * 5 * hw + r;
        const float key = lp[0];
        // heatmap_loss
        const float lg = hp[0];
        const float pr = 1.0f / (1.0f + expf(-lg));
        if (key >= 1.0f) {
            const float logsig = fminf(lg, 0.f) - log1pf(expf(-fabsf(lg)));
            acc[0] += -logsig * (1.f - pr) * (1.f - pr);
        } else {
            const float om = 1.f - key;
            acc[0] += (lg + softplusf_(-lg)) * pr * pr * (om * om * om * om);
        }
        const float w2 = fmaxf(key - 0.85f, 0.f) / (1.f - 0.85f);
        if (key > 0.85f) {
            acc[1] += (huber1(hp[1 * p.hs_c], lp[1 * hw]) + huber1(hp[2 * p.hs_c], lp[2 * hw])) * w2;
            acc[2] += w2;
        }
        acc[3] += bce_logits(hp[3 * p.hs_c], lp[3 * hw]);
        acc[4] += bce_logits(hp[4 * p.hs_c], lp[4 * hw]);
        const int code = p.idmap[(b * 2 + 1) * hw + r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float bit = (code & (1 << k)) ? 1.f : 0.f;
            acc[5 + k] += (1.f + bit * w2 + w2) * bce_logits(hp[(5 + k) * p.hs_c], bit);
        }
    }
#pragma unroll
    for (int j = 0; j < NMAP; ++j) red[j][threadIdx.x] = acc[j];
    __syncthreads();
    if (threadIdx.x < NMAP) {
        double s = 0.0;
        for (int k = 0; k < 256; ++k) s += (double)red[threadIdx.x][k];
        partial[(long)blockIdx.x * NMAP + threadIdx.x] = s;
    }
}

// id part (loss_func.py:135-163): one wave per selected pixel; three decoder heads.
//   partial[.][0] = sum of weight3 * (CE0 + CE1 + CE2) over mask3, [1] = weight3 sum, [2] = rows with all three arg-maxes right
//   (over mask4), [3] = rows in mask4
struct IdLossP {
    const float* dec[3]; int mod[3];
    const int32_t* sel_index; const int32_t* count; long cap;
    const float* label; const int32_t* idmap; long hw;
};

__global__ __launch_bounds__(256) void id_loss_kernel(IdLossP p, double* __restrict__ partial /* [gridDim.x][4] */) {
    __shared__ double red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n = p.count ? (*p.count < p.cap ? *p.count : p.cap) : p.cap;       // count == NULL: all `cap` rows are selected
    double a_ce = 0.0, a_w = 0.0, a_ok = 0.0, a_tot = 0.0;
    for (long r = (long)blockIdx.x * 4 + wave; r < n; r += (long)gridDim.x * 4) {
        const long px = p.sel_index[r];
        const long b = px / p.hw, q = px - b * p.hw;
        const float key = p.label[b * 5 * p.hw + q];
        const int id = p.idmap[b * 2 * p.hw + q];
        const bool m3 = key > 0.99f && id > 0, m4 = key == 1.0f && id > 0;
        if (!m3 && !m4) continue;                                   // wave-uniform
        float ce_sum = 0.f;
        int right = 0;
        for (int hd = 0; hd < 3; ++hd) {
            const int m = p.mod[hd];
            const float* row = p.dec[hd] + r * m;
            float mx = -INFINITY;
            int am = 0x7fffffff;
            for (int c = lane; c < m; c += 64) {
                const float v = row[c];
                if (v > mx) { mx = v; am = c; }                      // first maximum within the lane's strided walk
            }
            for (int o = 32; o > 0; o >>= 1) {                       // (max, lowest index) over the wave: torch.argmax returns the first
                const float omx = __shfl_xor(mx, o, 64);
                const int oam = __shfl_xor(am, o, 64);
                if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
            }
            float se = 0.f;
            for (int c = lane; c < m; c += 64) se += expf(row[c] - mx);
            se = wave_sum(se);
            const int tgt = id % m;
            ce_sum += (mx + logf(se)) - row[tgt];                    // -log_softmax[target]
            right += (am == tgt) ? 1 : 0;
        }
        if (lane == 0) {
            if (m3) { const float w3 = fmaxf(key - 0.99f, 0.f) / (1.f - 0.99f); a_ce += (double)(ce_sum * w3); a_w += (double)w3; }
            if (m4) { a_tot += 1.0; a_ok += right == 3 ? 1.0 : 0.0; }
        }
    }
    if (lane == 0) { red[0][wave] = a_ce; red[1][wave] = a_w; red[2][wave] = a_ok; red[3][wave] = a_tot; }
    __syncthreads();
    if (threadIdx.x < 4) partial[(long)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// out[0..14) = loss, keymap, size, textline, separator, id, code1, code2, code4, code8, correct, total, max(1, sum w1), max(1, sum w3)
__global__ void finish_losses_kernel(const double* __restrict__ map_partial, int n_map, const double* __restrict__ id_partial, int n_id,
                                     double n_pixels, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[NMAP] = {0}, d[4] = {0};
    for (int i = 0; i < n_map; ++i)
        for (int j = 0; j < NMAP; ++j) s[j] += map_partial[(long)i * NMAP + j];
    for (int i = 0; i < n_id; ++i)
        for (int j = 0; j < 4; ++j) d[j] += id_partial[(long)i * 4 + j];
    const float keymap = (float)(s[0] / n_pixels) * 10.f;
    const float size = (float)(s[1] / fmax(1.0, s[2]));
    const float textline = (float)(s[3] / n_pixels), sep = (float)(s[4] / n_pixels);
    const float idl = n_id > 0 ? (float)(d[0] / fmax(1.0, d[1])) : 0.f;
    float total = keymap + size + textline + sep + idl;
    for (int k = 0; k < 4; ++k) { out[6 + k] = (float)(s[5 + k] / n_pixels); total += out[6 + k]; }
    out[0] = total; out[1] = keymap; out[2] = size; out[3] = textline; out[4] = sep; out[5] = idl;
    out[10] = (float)d[2]; out[11] = (float)d[3];
    out[12] = (float)fmax(1.0, s[2]);                               // the two normalisers, for FTC_OP_LOSS_BWD
    out[13] = n_id > 0 ? (float)fmax(1.0, d[1]) : 1.0f;
}

// CoVWeightingLoss.forward (loss_func.py:24-72) for n <= 16 losses; state = [mean_L, mean_l, S_l, std_l][16] floats + alphas[16].
// (The reference's `if not self.train:` tests a bound method and is never true: the weighted form is used in validation too.)
__global__ void cov_step_kernel(const float* __restrict__ L, int n, int iter, float* __restrict__ state, float* __restrict__ out_loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float* mean_L = state; float* mean_l = state + 16; float* S_l = state + 32; float* std_l = state + 48; float* alphas = state + 64;
    float l[16];
    for (int i = 0; i < n; ++i) l[i] = L[i] / (iter == 0 ? L[i] : mean_L[i]);
    if (iter <= 1) {
        for (int i = 0; i < n; ++i) alphas[i] = 1.0f / (float)n;
    } else {
        float ls[16], tot = 0.f;
        for (int i = 0; i < n; ++i) { ls[i] = std_l[i] / mean_l[i]; tot += ls[i]; }
        for (int i = 0; i < n; ++i) alphas[i] = ls[i] / tot;
    }
    // Python computes mean_param and (1 - mean_param) in float64 and hands each to a float32 tensor op
    const double mpd = iter == 0 ? 0.0 : (1.0 - 1.0 / (double)(iter + 1));
    const float mp = (float)mpd, omp = (float)(1.0 - mpd);
    float loss = 0.f;
    for (int i = 0; i < n; ++i) {
        const float new_mean = mp * mean_l[i] + omp * l[i];
        S_l[i] += (l[i] - mean_l[i]) * (l[i] - new_mean);
        mean_l[i] = new_mean;
        std_l[i] = sqrtf(fmaxf(S_l[i] / (float)(iter + 1), 1e-16f));
        mean_L[i] = mp * mean_L[i] + omp * L[i];
    }
    for (int i = 0; i < n; ++i) loss += alphas[i] * L[i];          // sum(weighted_losses) in key order, as the reference's Python sum
    *out_loss = loss;
}

// ------------------------------------------------------------------------------------------------------------------------
// Training-mode BatchNorm (FTC_OP_BNSTAT / FTC_OP_BNACT): the BN-refresh pass of the reference (train1.py:203-211) runs the model in
// train() mode without gradients, so that every BatchNorm normalises with the statistics of the batch and moves its running
// statistics.  Statistics are summed in float64 in a fixed order: per (row chunk, channel) partial sums, then one thread per channel.
// ------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bnstat_partial_kernel(const T* __restrict__ x, double* __restrict__ part, long M, int C, int nchunk) {
    __shared__ double red[2][4][64];
    const int t = threadIdx.x, cl = t & 63, rl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int chunk = blockIdx.y;
    const long rows = (M + nchunk - 1) / nchunk;
    const long r0 = (long)chunk * rows, r1 = r0 + rows < M ? r0 + rows : M;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (long r = r0 + rl; r < r1; r += 4) {
            const double v = (double)to_f32<T>(x[r * C + c]);
            s1 += v;
            s2 += v * v;
        }
    red[0][rl][cl] = s1;
    red[1][rl][cl] = s2;
    __syncthreads();
    if (t < 64 && c < C) {
        part[((long)chunk * 2 + 0) * C + c] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
        part[((long)chunk * 2 + 1) * C + c] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
    }
}

// fp32 input with C % 4 == 0 (the train step): 16-byte lanes, Q channel quads x (256 / Q) row lanes per workgroup (Q = the largest
// power of two dividing C / 4); same sums as above up to the association of the per-lane partial sums
template <int Q>
__global__ __launch_bounds__(256) void bnstat_partial_q_kernel(const float* __restrict__ x, double* __restrict__ part, int M, int C, int nchunk) {
    constexpr int RL = 256 / Q;
    __shared__ double red[2][RL][Q * 4];
    const int t = threadIdx.x, cq = t % Q, rl = t / Q;
    const int c = (blockIdx.x * Q + cq) * 4;
    const int chunk = blockIdx.y;
    const int rows = (M + nchunk - 1) / nchunk;
    const int r0 = chunk * rows, r1 = min(M, r0 + rows);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    // eight rows' loads in flight before the first is consumed (the plain loop issued one 16-byte load per trip and waited for it);
    // the sums are taken in the same row order
    constexpr int U = 8;
    for (int r = r0 + rl; r < r1; r += RL * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * RL;
            v[u] = rr < r1 ? *reinterpret_cast<const f32x4*>(x + (long)rr * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d = (double)v[u][e];
                s1[e] += d;
                s2[e] += d * d;
            }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][rl][cq * 4 + e] = s1[e];
        red[1][rl][cq * 4 + e] = s2[e];
    }
    __syncthreads();
    for (int i = t; i < 2 * Q * 4; i += 256) {
        const int which = i / (Q * 4), l = i - which * (Q * 4);
        double a = 0.0;
#pragma unroll 4
        for (int k = 0; k < RL; ++k) a += red[which][k][l];
        part[((long)chunk * 2 + which) * C + blockIdx.x * Q * 4 + l] = a;
    }
}

// 16 channels x 16 chunk lanes per workgroup: lane k sums chunks k, k+16, ... in a fixed order (one thread per channel walking up to
// 512 chunks made this kernel, not the streaming pass, the larger half of a BNSTAT op)
__global__ __launch_bounds__(256) void bnstat_final_kernel(const double* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running, float* __restrict__ out, long M, int C, int nchunk, float eps,
                                                           float momentum) {
    __shared__ double red[2][16][16];
    const int t = threadIdx.x, cl = t & 15, kl = t >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int k = kl; k < nchunk; k += 16) {
            s1 += part[((long)k * 2 + 0) * C + c];
            s2 += part[((long)k * 2 + 1) * C + c];
        }
    red[0][kl][cl] = s1;
    red[1][kl][cl] = s2;
    __syncthreads();
    if (t >= 16 || c >= C) return;
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        s1 += red[0][k][t];
        s2 += red[1][k][t];
    }
    const double mean = s1 / (double)M;
    double var = s2 / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const double sc = (double)gamma[c] / sqrt(var + (double)eps);
    out[c] = (float)sc;
    out[C + c] = (float)((double)beta[c] - mean * sc);
    out[2 * C + c] = (float)mean;                                   // for FTC_OP_BNBWD
    out[3 * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running[c] = (float)((1.0 - (double)momentum) * (double)running[c] + (double)momentum * mean);
        running[C + c] = (float)((1.0 - (double)momentum) * (double)running[C + c] + (double)momentum * unb);
    }
}

// grid (C/64, row chunks P, B): 64 channels x 4 row lanes; the activation in fp32 exactly as the inference epilogues apply it
template <typename TI, typename TO, typename TC>
__global__ __launch_bounds__(256) void bnact_kernel(const TI* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const void* __restrict__ res, int res_dtype, const float* __restrict__ keep, TO* __restrict__ out,
                                                    TC* __restrict__ out2, float* __restrict__ sums, int HW, int C, int P, int act) {
    __shared__ float red[4][64];
    const int t = threadIdx.x, cl = t & 63, rl = t >> 6;
    const int c = blockIdx.x * 64 + cl, p = blockIdx.y, b = blockIdx.z;
    const int rows = (HW + P - 1) / P;
    const int r0 = p * rows, r1 = min(HW, r0 + rows);
    float acc = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c];
        const float kp = keep ? keep[b] : 1.0f;
        for (int r = r0 + rl; r < r1; r += 4) {
            const long i = ((long)b * HW + r) * C + c;
            float v = to_f32<TI>(x[i]) * sc + sh;
            v = apply_act_rt(v, act);
            if (res) v = v * kp + (res_dtype == FTC_F32 ? reinterpret_cast<const float*>(res)[i] : to_f32<TC>(reinterpret_cast<const TC*>(res)[i]));
            out[i] = from_f32<TO>(v);
            if (out2) out2[i] = from_f32<TC>(v);
            acc += v;
        }
    }
    if (sums) {
        red[rl][cl] = acc;
        __syncthreads();
        if (t < 64 && c < C) sums[((long)b * P + p) * C + c] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
}

// (round 6) The same pass with 16-byte lanes and U rows in flight: the kernel above moves 4 bytes per lane and trip and waits for each load before it
// issues the next (12 ms of the train step's forward over 320 launches, profiles/r06b_train_bf16_b8_kernel_stats.txt).  Q channel quads x (256 / Q) row
// lanes per workgroup (Q = the largest power of two <= 64 dividing C / 4); same arithmetic per element; with Q = 64 the same four row lanes and the same
// association of the channel sums as the scalar kernel.
__device__ __forceinline__ f32x4 act4_rt(f32x4 v, int act) { return f32x4{apply_act_rt(v[0], act), apply_act_rt(v[1], act), apply_act_rt(v[2], act), apply_act_rt(v[3], act)}; }

template <typename TI, typename TO, typename TC, int Q>
__global__ __launch_bounds__(256) void bnact_q_kernel(const TI* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const void* __restrict__ res, int res_dtype, const float* __restrict__ keep, TO* __restrict__ out,
                                                      TC* __restrict__ out2, float* __restrict__ sums, int HW, int C, int P, int act) {
    constexpr int RL = 256 / Q, U = 4;
    __shared__ __attribute__((aligned(16))) float red[RL][Q * 4];
    const int t = threadIdx.x, cq = t % Q, rl = t / Q;
    const int c = (blockIdx.x * Q + cq) * 4, p = blockIdx.y, b = blockIdx.z;
    const int rows = (HW + P - 1) / P;
    const int r0 = p * rows, r1 = min(HW, r0 + rows);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    const float kp = keep ? keep[b] : 1.0f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc = zero;
    for (int r = r0 + rl; r < r1; r += RL * U) {
        f32x4 xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * RL;
            const bool ok = rr < r1;
            const long i = ((long)b * HW + (ok ? rr : r)) * C + c;
            xv[u] = ok ? load4<TI>(x + i) : zero;
            rv[u] = zero;
            if (res && ok) rv[u] = res_dtype == FTC_F32 ? load4<float>(reinterpret_cast<const float*>(res) + i) : load4<TC>(reinterpret_cast<const TC*>(res) + i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * RL;
            if (rr >= r1) break;
            const long i = ((long)b * HW + rr) * C + c;
            f32x4 v = act4_rt(xv[u] * sc + sh, act);
            if (res) v = v * kp + rv[u];
            store4<TO>(out + i, v);
            if (out2) store4<TC>(out2 + i, v);
            acc += v;
        }
    }
    if (sums) {
        *reinterpret_cast<f32x4*>(&red[rl][cq * 4]) = acc;
        __syncthreads();
        if (t < Q * 4) {
            float tot;
            if constexpr (RL == 4) tot = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
            else {
                tot = 0.f;
#pragma unroll
                for (int k = 0; k < RL; ++k) tot += red[k][t];
            }
            sums[((long)b * P + p) * C + blockIdx.x * Q * 4 + t] = tot;
        }
    }
}

}  // namespace

hipError_t launch_bnstat(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const long M = (long)o.B * o.H * o.W;
    const int C = o.Cin, nchunk = ftc_bnstat_chunks(M);
    double* part = reinterpret_cast<double*>(const_cast<void*>(a.in2));
    float eps, mom;
    memcpy(&eps, &o.aux0, 4);
    memcpy(&mom, &o.aux1, 4);
    const dim3 grid((C + 63) / 64, nchunk);
    if (o.in_dtype == FTC_F32 && C % 4 == 0 && M < 0x7fffffffL) {
        const int q = C / 4;
        const int Q = q % 64 == 0 ? 64 : q % 32 == 0 ? 32 : q % 16 == 0 ? 16 : q % 8 == 0 ? 8 : q % 4 == 0 ? 4 : q % 2 == 0 ? 2 : 1;
        const dim3 gq(C / (4 * Q), nchunk);
#define BNSTAT_Q(QQ) hipLaunchKernelGGL(bnstat_partial_q_kernel<QQ>, gq, dim3(256), 0, s, (const float*)a.in, part, (int)M, C, nchunk)
        switch (Q) { case 64: BNSTAT_Q(64); break; case 32: BNSTAT_Q(32); break; case 16: BNSTAT_Q(16); break; case 8: BNSTAT_Q(8); break;
                     case 4: BNSTAT_Q(4); break; case 2: BNSTAT_Q(2); break; default: BNSTAT_Q(1); }
#undef BNSTAT_Q
    }
    else if (o.in_dtype == FTC_F32) hipLaunchKernelGGL(bnstat_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)a.in, part, M, C, nchunk);
    else if (o.in_dtype == FTC_F16) hipLaunchKernelGGL(bnstat_partial_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)a.in, part, M, C, nchunk);
    else hipLaunchKernelGGL(bnstat_partial_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)a.in, part, M, C, nchunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bnstat_final_kernel, dim3((C + 15) / 16), dim3(256), 0, s, part, (const float*)a.w, a.bias, a.aux, (float*)a.out, M, C, nchunk, eps, mom);
    return hipGetLastError();
}

namespace {
template <typename TI, typename TO, typename TC>
hipError_t launch_bnact_t(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const int HW = o.H * o.W, C = o.Cin, P = o.aux0 > 0 ? o.aux0 : 1;
    const bool has_res = (o.flags & FTC_FLAG_RESIDUAL) != 0;
    static const bool scalar_only = [] { const char* e = std::getenv("FTC_BNACT_SCALAR"); return e && *e && *e != '0'; }();
    if (C % 4 == 0 && !scalar_only) {
        const int q = C / 4;
        const int Q = q % 64 == 0 ? 64 : q % 32 == 0 ? 32 : q % 16 == 0 ? 16 : q % 8 == 0 ? 8 : q % 4 == 0 ? 4 : q % 2 == 0 ? 2 : 1;
        const dim3 grid(C / (4 * Q), P, o.B);
#define BNACT_Q(QQ) hipLaunchKernelGGL((bnact_q_kernel<TI, TO, TC, QQ>), grid, dim3(256), 0, s, (const TI*)a.in, a.scale, a.shift, has_res ? a.in2 : nullptr, o.res_dtype, \
                                       has_res ? static_cast<const float*>(a.w2) : nullptr, (TO*)a.out, (TC*)a.out2, a.aux, HW, C, P, o.act)
        switch (Q) { case 64: BNACT_Q(64); break; case 32: BNACT_Q(32); break; case 16: BNACT_Q(16); break; case 8: BNACT_Q(8); break; case 4: BNACT_Q(4); break;
                     case 2: BNACT_Q(2); break; default: BNACT_Q(1); }
#undef BNACT_Q
        return hipGetLastError();
    }
    hipLaunchKernelGGL((bnact_kernel<TI, TO, TC>), dim3((C + 63) / 64, P, o.B), dim3(256), 0, s, (const TI*)a.in, a.scale, a.shift, has_res ? a.in2 : nullptr,
                       o.res_dtype, has_res ? static_cast<const float*>(a.w2) : nullptr, (TO*)a.out, (TC*)a.out2, a.aux, HW, C, P, o.act);
    return hipGetLastError();
}
template <typename TI, typename TC>
hipError_t launch_bnact_i(const OpArgs& a, hipStream_t s) {
    if (a.op->out_dtype == FTC_F32) return launch_bnact_t<TI, float, TC>(a, s);
    return launch_bnact_t<TI, TC, TC>(a, s);
}
}  // namespace

hipError_t launch_bnact(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    // TC = the plan's 16-bit type (w_dtype: bf16 unless fp16): 16-bit outputs, copies and residuals come in it
    if (o.w_dtype == FTC_F16) {
        if (o.in_dtype == FTC_F32) return launch_bnact_i<float, _Float16>(a, s);
        return launch_bnact_i<_Float16, _Float16>(a, s);
    }
    if (o.in_dtype == FTC_F32) return launch_bnact_i<float, __bf16>(a, s);
    return launch_bnact_i<__bf16, __bf16>(a, s);
}

hipError_t launch_topk_mask(const float* vals, long n, long k, unsigned char* mask, int32_t* sel_index, int32_t* count, hipStream_t s) {
    hipLaunchKernelGGL(topk_mask_kernel, dim3(1), dim3(1024), 0, s, vals, n, k, mask, sel_index, count);
    return hipGetLastError();
}
hipError_t launch_mask_compact(const unsigned char* mask, long n, int32_t* sel_index, long cap, int32_t* count, hipStream_t s) {
    hipLaunchKernelGGL(mask_compact_kernel, dim3(1), dim3(1024), 0, s, mask, n, sel_index, cap, count);
    return hipGetLastError();
}
hipError_t launch_gather_rows(const float* feat, const int32_t* sel_index, const int32_t* count, long cap, int C, int Cpad, void* rows, int out_dtype,
                              hipStream_t s) {
    const long total = cap * (Cpad / 4);
    const int nb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (out_dtype == FTC_F32) hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(nb), dim3(256), 0, s, feat, sel_index, count, cap, C, Cpad, (float*)rows);
    else if (out_dtype == FTC_F16) hipLaunchKernelGGL(gather_rows_kernel<_Float16>, dim3(nb), dim3(256), 0, s, feat, sel_index, count, cap, C, Cpad, (_Float16*)rows);
    else hipLaunchKernelGGL(gather_rows_kernel<__bf16>, dim3(nb), dim3(256), 0, s, feat, sel_index, count, cap, C, Cpad, (__bf16*)rows);
    return hipGetLastError();
}
hipError_t launch_losses(const float* heat, const long* hstrides, const float* label, const int32_t* idmap, int B, int h, int w, const float* const* dec,
                         const int* mod, const int32_t* sel_index, const int32_t* count, long cap, float* out, void* scratch, hipStream_t s) {
    MapLossP mp{heat, hstrides[0], hstrides[1], hstrides[2], hstrides[3], label, idmap, B, h, w};
    const long n = (long)B * h * w;
    const int nb_map = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    double* map_partial = static_cast<double*>(scratch);
    double* id_partial = map_partial + (long)512 * NMAP;
    hipLaunchKernelGGL(map_loss_kernel, dim3(nb_map), dim3(256), 0, s, mp, map_partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int nb_id = 0;
    if (dec && sel_index && cap > 0) {
        IdLossP ip{{dec[0], dec[1], dec[2]}, {mod[0], mod[1], mod[2]}, sel_index, count, cap, label, idmap, (long)h * w};
        nb_id = (int)((cap + 3) / 4 < 512 ? (cap + 3) / 4 : 512);
        hipLaunchKernelGGL(id_loss_kernel, dim3(nb_id), dim3(256), 0, s, ip, id_partial);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(finish_losses_kernel, dim3(1), dim3(64), 0, s, map_partial, nb_map, id_partial, nb_id, (double)n, out);
    return hipGetLastError();
}
hipError_t launch_cov_step(const float* L, int n, int iter, float* state, float* out_loss, hipStream_t s) {
    hipLaunchKernelGGL(cov_step_kernel, dim3(1), dim3(64), 0, s, L, n, iter, state, out_loss);
    return hipGetLastError();
}
