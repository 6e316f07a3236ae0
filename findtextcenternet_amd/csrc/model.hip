// Self-contained model entry points of the C ABI (include/ftc.h): ftc_create / ftc_forward / ftc_destroy.
//
// Everything a host needs to run the detector path lives behind them, in the library: the network
// description (what the reference expresses as nn.Module composition -- CenterNetDetection.forward,
// /root/reference/models/detector.py:217-230 = stem + 100 Fused-MBConv / MBConv blocks with taps
// (BackboneModel.forward :139-146, config rows :12-28) + nine Leafmap heads (:148-201) -- followed by the NMS of
// CenterNetDetector.forward (:289-296)), the weight packing (eval-mode BatchNorm folded into the preceding convolution
// in float64, K-major [Cout][kh*kw][Cin] re-layout, conversion to the MFMA compute type, one blob), the per-shape op
// list with its liveness-based activation arena, and the measured kernel selection (tuning_table.inc).
// Host-only code: no kernels here.  Compiled with -ffp-contract=off so the float64 folding is the plain
// multiply / subtract sequence (no fused rounding differences between builds).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "ftc_host.h"

namespace {

constexpr int64_t ALIGN = 256;
inline int64_t align_up(int64_t n, int64_t a = ALIGN) { return (n + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------
// network description
// ------------------------------------------------------------------------------------------------
struct StageRow { bool fused; int expand, kernel, stride, cin, cout, layers; };
struct BlockSpec {
    bool fused;
    std::string prefix;      // "backbone.features.4.0"
    int cin, cout, exp, stride, squeeze;
    bool residual;
};
struct HeadSpec { const char* name; int out_dim; int ch0; };

// efficientnet_v2_xl (models/detector.py:12-28) and torchvision's published s/m/l tables (models/detector.py:131-136)
const std::vector<StageRow>& stage_rows(const std::string& size) {
    static const std::map<std::string, std::vector<StageRow>> t = {
        {"xl", {{true, 1, 3, 1, 32, 32, 4}, {true, 4, 3, 2, 32, 64, 8}, {true, 4, 3, 2, 64, 96, 8}, {false, 4, 3, 2, 96, 192, 16},
                {false, 6, 3, 1, 192, 256, 24}, {false, 6, 3, 2, 256, 512, 32}, {false, 6, 3, 1, 512, 640, 8}}},
        {"l", {{true, 1, 3, 1, 32, 32, 4}, {true, 4, 3, 2, 32, 64, 7}, {true, 4, 3, 2, 64, 96, 7}, {false, 4, 3, 2, 96, 192, 10},
               {false, 6, 3, 1, 192, 224, 19}, {false, 6, 3, 2, 224, 384, 25}, {false, 6, 3, 1, 384, 640, 7}}},
        {"m", {{true, 1, 3, 1, 24, 24, 3}, {true, 4, 3, 2, 24, 48, 5}, {true, 4, 3, 2, 48, 80, 5}, {false, 4, 3, 2, 80, 160, 7},
               {false, 6, 3, 1, 160, 176, 14}, {false, 6, 3, 2, 176, 304, 18}, {false, 6, 3, 1, 304, 512, 5}}},
        {"s", {{true, 1, 3, 1, 24, 24, 2}, {true, 4, 3, 2, 24, 48, 4}, {true, 4, 3, 2, 48, 64, 4}, {false, 4, 3, 2, 64, 128, 6},
               {false, 6, 3, 1, 128, 160, 9}, {false, 6, 3, 2, 160, 256, 15}}},
    };
    static const std::vector<StageRow> none;
    auto it = t.find(size);
    return it == t.end() ? none : it->second;
}
std::vector<int> tap_dims(const std::string& size) {
    if (size == "xl") return {64, 96, 256, 1280};
    if (size == "l") return {64, 96, 224, 1280};
    if (size == "m") return {48, 80, 176, 1280};
    return {48, 64, 160, 1280};
}
constexpr int LAST_CHANNEL = 1280, FPN_DIM = 192, FEATURE_DIM = 100;
constexpr double BACKBONE_BN_EPS = 1e-3, HEAD_BN_EPS = 1e-5;     // models/detector.py:27; nn.BatchNorm2d default (:161-184)
// CenterNetDetection heads in forward order (models/detector.py:207-230; the reference's spelling "sepatator")
const HeadSpec HEADS[9] = {{"keyheatmap", 1, 0}, {"sizes", 2, 1}, {"textline", 1, 3}, {"sepatator", 1, 4}, {"code1", 1, 5},
                           {"code2", 1, 6}, {"code4", 1, 7}, {"code8", 1, 8}, {"feature", FEATURE_DIM, -1}};
constexpr int NHEADS = 9;
constexpr int DECODER_MID = 2048, DECODER_KPAD = 128;
const int DECODER_MODULO[3] = {1091, 1093, 1097};          // util_func.py:5 modulo_list

int make_divisible(double v, int d = 8) {        // torchvision _make_divisible
    int nv = std::max(d, (int)(v + d / 2.0) / d * d);
    if (nv < 0.9 * v) nv += d;
    return nv;
}

std::vector<std::vector<BlockSpec>> backbone_blocks(const std::string& size) {
    std::vector<std::vector<BlockSpec>> out;
    const auto& rows = stage_rows(size);
    for (size_t si = 0; si < rows.size(); ++si) {
        const StageRow& r = rows[si];
        std::vector<BlockSpec> st;
        for (int j = 0; j < r.layers; ++j) {
            const int bcin = j == 0 ? r.cin : r.cout, bstride = j == 0 ? r.stride : 1;
            st.push_back({r.fused, "backbone.features." + std::to_string(si + 1) + "." + std::to_string(j), bcin, r.cout,
                          make_divisible((double)bcin * r.expand), bstride, std::max(1, bcin / 4), bstride == 1 && bcin == r.cout});
        }
        out.push_back(st);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
struct TensorView {
    const float* data = nullptr;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Weights {
    std::map<std::string, TensorView> t;
    std::string missing;
    const TensorView* get(const std::string& k, std::initializer_list<int64_t> shape) {
        auto it = t.find(k);
        if (it == t.end()) { if (missing.empty()) missing = "missing tensor '" + k + "'"; return nullptr; }
        if (it->second.shape != std::vector<int64_t>(shape)) {
            if (missing.empty()) {
                missing = "tensor '" + k + "' has shape [";
                for (auto s : it->second.shape) missing += std::to_string(s) + ",";
                missing += "], expected [";
                for (auto s : shape) missing += std::to_string(s) + ",";
                missing += "]";
            }
            return nullptr;
        }
        return &it->second;
    }
};

inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

inline float f16_to_f32(uint16_t u) {
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return (float)h;
}

inline uint16_t f32_to_f16_rne(float f) {           // IEEE half, round-to-nearest-even, saturating like the device stores
    if (f > 65504.0f) f = 65504.0f;
    if (f < -65504.0f) f = -65504.0f;
    const _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

// The fold / re-layout / conversion loops below touch every one of the 262 M weights in float64: on one thread ftc_create took 5-9 s.
// par_for splits an index range over a few threads when it is large (results do not depend on the split: every index is independent).
template <typename F>
void par_for(int64_t n, int64_t grain, F&& f) {          // f(begin, end); begin is a multiple of `grain`
    const int64_t chunks = (n + grain - 1) / grain;
    int nt = (int)std::min<int64_t>(std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency())), chunks);
    if (n < (int64_t)1 << 18 || nt <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    const int64_t per = (chunks + nt - 1) / nt * grain;
    for (int t = 0; t < nt; ++t) {
        const int64_t b = t * per, e = std::min(n, b + per);
        if (b < e) th.emplace_back([&f, b, e] { f(b, e); });
    }
    for (auto& x : th) x.join();
}

struct Blob {
    std::vector<uint8_t> bytes;
    std::map<std::string, int64_t> table;
    uint8_t* add(const std::string& name, int64_t nbytes) {
        const int64_t off = (int64_t)bytes.size();
        table[name] = off;
        bytes.resize((size_t)align_up(off + nbytes), 0);
        return bytes.data() + off;
    }
    void add_f32(const std::string& name, const double* v, int64_t n) {
        float* d = reinterpret_cast<float*>(add(name, n * 4));
        par_for(n, 1024, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) d[i] = (float)v[i]; });
    }
    void add_f32(const std::string& name, const float* v, int64_t n) { std::memcpy(add(name, n * 4), v, (size_t)n * 4); }
    // MFMA compute type: fp32, or bf16 / fp16 (double -> float -> 16 bit, both steps round-to-nearest-even)
    void add_compute(const std::string& name, const double* v, int64_t n, int dt) {
        if (dt == FTC_F32) { add_f32(name, v, n); return; }
        if (dt == FTC_PRECISION_F16X3) {                    // fp16x3: every 16-byte chunk of four fp32 weights becomes [hi x4 | lo x4] IEEE halves
            uint16_t* d = reinterpret_cast<uint16_t*>(add(name, n * 4));
            par_for(n, 1024, [&](int64_t b, int64_t en) {
                for (int64_t i = b; i + 3 < en; i += 4)
                    for (int e = 0; e < 4; ++e) {
                        float x = (float)v[i + e];
                        const float xs = x > 65504.0f ? 65504.0f : x < -65504.0f ? -65504.0f : x;
                        const uint16_t h = f32_to_f16_rne(xs);
                        d[2 * i + e] = h;
                        d[2 * i + 4 + e] = f32_to_f16_rne(x - f16_to_f32(h));
                    }
            });
            return;
        }
        uint16_t* d = reinterpret_cast<uint16_t*>(add(name, n * 2));
        if (dt == FTC_BF16) par_for(n, 1024, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) d[i] = f32_to_bf16_rne((float)v[i]); });
        else par_for(n, 1024, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) d[i] = f32_to_f16_rne((float)v[i]); });
    }
};

struct BnAffine { std::vector<double> s, t; };     // y = x*s + t  (eval-mode BatchNorm)

bool bn_affine(Weights& w, const std::string& p, int c, double eps, BnAffine* out) {
    const TensorView *g = w.get(p + ".weight", {c}), *b = w.get(p + ".bias", {c}), *m = w.get(p + ".running_mean", {c}),
                     *v = w.get(p + ".running_var", {c});
    if (!g || !b || !m || !v) return false;
    out->s.resize(c);
    out->t.resize(c);
    for (int i = 0; i < c; ++i) {
        const double s = (double)g->data[i] / std::sqrt((double)v->data[i] + eps);
        out->s[i] = s;
        out->t[i] = (double)b->data[i] - (double)m->data[i] * s;
    }
    return true;
}

// conv weight [O,I,kh,kw] followed by an eval BN -> (W*s in OIHW order, float64; bias float64)
bool fold(Weights& w, const std::string& conv_key, const std::string& bn_prefix, double eps, int O, int I, int k, std::vector<double>* wf,
          std::vector<double>* bias) {
    const TensorView* cw = w.get(conv_key, {O, I, k, k});
    BnAffine a;
    if (!cw || !bn_affine(w, bn_prefix, O, eps, &a)) return false;
    const int64_t per = (int64_t)I * k * k;
    wf->resize((size_t)O * per);
    par_for((int64_t)O * per, per, [&](int64_t b, int64_t e) {
        for (int64_t o = b / per; o * per < e; ++o) {
            const double s = a.s[o];
            const float* src = cw->data + o * per;
            double* dst = wf->data() + o * per;
            for (int64_t i = 0; i < per; ++i) dst[i] = (double)src[i] * s;
        }
    });
    *bias = a.t;
    return true;
}

// [O,I,kh,kw] -> [O, kh*kw, I]
std::vector<double> kmajor(const std::vector<double>& w, int O, int I, int k) {
    std::vector<double> out(w.size());
    const int kk = k * k;
    const int64_t per = (int64_t)I * kk;
    par_for((int64_t)O * per, per, [&](int64_t b, int64_t e) {
        for (int64_t o = b / per; o * per < e; ++o)
            for (int i = 0; i < I; ++i)
                for (int t = 0; t < kk; ++t) out[((size_t)o * kk + t) * I + i] = w[((size_t)o * I + i) * kk + t];
    });
    return out;
}

// Bias table of a 3x3 convolution whose INPUT carries a folded per-channel shift t (a BatchNorm in front of a zero-padded
// convolution: the shift does not see the padding ring): entry idx = top | bottom<<1 | left<<2 | right<<3 sums the shift
// contribution of the taps that fall inside the image.  wf = [N][C][3][3] (already scaled by the output BN), cs = first
// input channel the shift applies to, ti = shift per channel.
void border_bias16(const std::vector<double>& wf, int N, int C, int cs, const std::vector<double>& ti, const std::vector<double>& bo,
                   std::vector<double>* b16 /* [16][N] */) {
    std::vector<double> tmap((size_t)N * 9, 0.0);
    const int nt = (int)ti.size();
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < nt; ++c) {
            const double* p = wf.data() + ((size_t)n * C + cs + c) * 9;
            for (int t = 0; t < 9; ++t) tmap[(size_t)n * 9 + t] += p[t] * ti[c];
        }
    b16->assign((size_t)16 * N, 0.0);
    for (int idx = 0; idx < 16; ++idx)
        for (int n = 0; n < N; ++n) {
            double acc = 0.0;
            for (int r = 0; r < 3; ++r) {
                if ((r == 0 && (idx & 1)) || (r == 2 && (idx & 2))) continue;
                for (int c = 0; c < 3; ++c) {
                    if ((c == 0 && (idx & 4)) || (c == 2 && (idx & 8))) continue;
                    acc += tmap[(size_t)n * 9 + r * 3 + c];
                }
            }
            (*b16)[(size_t)idx * N + n] = bo[n] + acc;
        }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// model object
// ------------------------------------------------------------------------------------------------
struct OpMeta { std::string name, kind; double flops = 0, bytes = 0; };
struct ModelPlan {
    ftc_plan plan;
    std::vector<OpMeta> meta;
    int B, H, W, h, w;
    int64_t peak_live_bytes = 0, total_buffer_bytes = 0;
};

struct ftc_model {
    std::string size;
    int precision;                          // FTC_F32 | FTC_BF16 | FTC_F16
    int split16 = 0;                        // FTC_PRECISION_F16X3: the fp32 plan with FTC_FLAG_SPLIT16 on every convolution
    Blob blob;
    bool has_decoder = false;               // the checkpoint carried the "decoder.*" tensors (SimpleDecoder)
    std::mutex mu;
    std::map<std::tuple<int, int, int, int>, std::unique_ptr<ModelPlan>> plans;
    // by number of rows.  The row count follows the data (peaks per page), so this cache is bounded: at most kMaxDecoderPlans entries,
    // the least recently used one is dropped (shared_ptr: a caller that is still running the evicted plan keeps it alive)
    std::map<int, std::shared_ptr<ModelPlan>> decoder_plans;
    std::map<int, uint64_t> decoder_plan_use;
    uint64_t decoder_clock = 0;
    static constexpr size_t kMaxDecoderPlans = 16;
};

namespace {

bool env_on(const char* k) { const char* v = std::getenv(k); return v && *v && std::strcmp(v, "0") != 0; }

// ---- weight packing (once per checkpoint) -------------------------------------------------------
int pack_weights(ftc_model* m, Weights& w) {
    const bool bf = m->precision != FTC_F32;      // a 16-bit speed mode (bf16 or fp16 operands): the fused / folded head variants exist
    const int cdt = m->split16 ? FTC_PRECISION_F16X3 : m->precision;      // storage of the MFMA weight operands (fp16x3: pre-split fp32 chunks)
    Blob& bl = m->blob;
    bool ok = true;
    std::vector<double> wf, b;
    auto conv_bn = [&](const std::string& name, const std::string& conv_key, const std::string& bn, double eps, int O, int I, int k) {
        if (!fold(w, conv_key, bn, eps, O, I, k, &wf, &b)) { ok = false; return; }
        const std::vector<double> km = kmajor(wf, O, I, k);
        bl.add_compute(name + ".w", km.data(), (int64_t)km.size(), cdt);
        bl.add_f32(name + ".b", b.data(), O);
    };
    const auto stages = backbone_blocks(m->size);
    const int c0 = stage_rows(m->size)[0].cin;
    // stem: [C0,3,3,3] -> [(r*3+s)*3+c][C0] fp32 (VALU kernel, always fp32)
    if (fold(w, "backbone.features.0.0.weight", "backbone.features.0.1", BACKBONE_BN_EPS, c0, 3, 3, &wf, &b)) {
        std::vector<double> sw((size_t)27 * c0);
        for (int o = 0; o < c0; ++o)
            for (int c = 0; c < 3; ++c)
                for (int t = 0; t < 9; ++t) sw[((size_t)t * 3 + c) * c0 + o] = wf[((size_t)o * 3 + c) * 9 + t];
        bl.add_f32("stem.w", sw.data(), (int64_t)sw.size());
        bl.add_f32("stem.b", b.data(), c0);
    } else ok = false;
    for (const auto& st : stages)
        for (const BlockSpec& blk : st) {
            const std::string p = blk.prefix + ".block";
            if (blk.fused) {
                if (blk.exp != blk.cin) {
                    conv_bn(p + ".0", p + ".0.0.weight", p + ".0.1", BACKBONE_BN_EPS, blk.exp, blk.cin, 3);
                    conv_bn(p + ".1", p + ".1.0.weight", p + ".1.1", BACKBONE_BN_EPS, blk.cout, blk.exp, 1);
                } else {
                    conv_bn(p + ".0", p + ".0.0.weight", p + ".0.1", BACKBONE_BN_EPS, blk.cout, blk.cin, 3);
                }
            } else {
                conv_bn(p + ".0", p + ".0.0.weight", p + ".0.1", BACKBONE_BN_EPS, blk.exp, blk.cin, 1);
                if (fold(w, p + ".1.0.weight", p + ".1.1", BACKBONE_BN_EPS, blk.exp, 1, 3, &wf, &b)) {          // depthwise [C,1,3,3] -> [9][C]
                    std::vector<double> dw((size_t)9 * blk.exp);
                    for (int c = 0; c < blk.exp; ++c)
                        for (int t = 0; t < 9; ++t) dw[(size_t)t * blk.exp + c] = wf[(size_t)c * 9 + t];
                    bl.add_f32(p + ".1.w", dw.data(), (int64_t)dw.size());
                    bl.add_f32(p + ".1.b", b.data(), blk.exp);
                } else ok = false;
                const TensorView *w1 = w.get(p + ".2.fc1.weight", {blk.squeeze, blk.exp, 1, 1}), *b1 = w.get(p + ".2.fc1.bias", {blk.squeeze}),
                                 *w2 = w.get(p + ".2.fc2.weight", {blk.exp, blk.squeeze, 1, 1}), *b2 = w.get(p + ".2.fc2.bias", {blk.exp});
                if (w1 && b1 && w2 && b2) {
                    bl.add_f32(p + ".2.w1", w1->data, (int64_t)blk.squeeze * blk.exp);                         // [S][C]
                    bl.add_f32(p + ".2.b1", b1->data, blk.squeeze);
                    std::vector<float> w2t((size_t)blk.squeeze * blk.exp);                                     // fc2 transposed [S][C]
                    for (int c = 0; c < blk.exp; ++c)
                        for (int s = 0; s < blk.squeeze; ++s) w2t[(size_t)s * blk.exp + c] = w2->data[(size_t)c * blk.squeeze + s];
                    bl.add_f32(p + ".2.w2t", w2t.data(), (int64_t)w2t.size());
                    bl.add_f32(p + ".2.b2", b2->data, blk.exp);
                } else ok = false;
                conv_bn(p + ".3", p + ".3.0.weight", p + ".3.1", BACKBONE_BN_EPS, blk.cout, blk.exp, 1);
            }
        }
    const int nfeat = (int)stages.size() + 1;
    const std::string hp = "backbone.features." + std::to_string(nfeat);
    const int clast = stages.back().back().cout;
    conv_bn(hp, hp + ".0.weight", hp + ".1", BACKBONE_BN_EPS, LAST_CHANNEL, clast, 1);
    const std::vector<int> taps = tap_dims(m->size);
    const int ntap = (int)taps.size();
    // FPN level 0 of all nine heads as ONE convolution over the shared 1/32 tap: each head's input BatchNorm is folded in
    // exactly -- scale into the weights, shift into a 16-entry border-case bias table.  Leafmap.forward i=0, detector.py:194-197.
    {
        const int C4 = taps[ntap - 1];
        std::vector<double> wm_all, b16_all((size_t)16 * NHEADS * FPN_DIM);
        wm_all.reserve((size_t)NHEADS * FPN_DIM * 9 * C4);
        for (int hi = 0; hi < NHEADS; ++hi) {
            const std::string name = HEADS[hi].name;
            BnAffine in;
            if (!bn_affine(w, name + ".in_bn." + std::to_string(ntap - 1), C4, HEAD_BN_EPS, &in) ||
                !fold(w, name + ".upsamplers.0.0.weight", name + ".upsamplers.0.1", HEAD_BN_EPS, FPN_DIM, C4, 3, &wf, &b)) { ok = false; break; }
            std::vector<double> b16;
            border_bias16(wf, FPN_DIM, C4, 0, in.t, b, &b16);
            for (int n = 0; n < FPN_DIM; ++n)
                for (int c = 0; c < C4; ++c) {
                    double* q = wf.data() + ((size_t)n * C4 + c) * 9;
                    for (int t = 0; t < 9; ++t) q[t] *= in.s[c];
                }
            const std::vector<double> km = kmajor(wf, FPN_DIM, C4, 3);
            wm_all.insert(wm_all.end(), km.begin(), km.end());
            for (int idx = 0; idx < 16; ++idx)
                for (int n = 0; n < FPN_DIM; ++n) b16_all[(size_t)idx * NHEADS * FPN_DIM + hi * FPN_DIM + n] = b16[(size_t)idx * FPN_DIM + n];
        }
        if (ok) {
            bl.add_compute("heads.L0.w", wm_all.data(), (int64_t)wm_all.size(), cdt);
            bl.add_f32("heads.L0.b", b16_all.data(), (int64_t)b16_all.size());                         // [16][9*192]
        }
    }
    // FPN levels 1.. and the input BatchNorms of the nine heads are stored head-major ([9][...]) so that one grouped launch
    // (ftc_op.groups = 9) covers all heads of a level.
    for (int i = 0; i < ntap - 1 && ok; ++i) {
        std::vector<float> sc((size_t)NHEADS * taps[i]), sh((size_t)NHEADS * taps[i]);
        for (int hi = 0; hi < NHEADS; ++hi) {
            BnAffine a;
            if (!bn_affine(w, std::string(HEADS[hi].name) + ".in_bn." + std::to_string(i), taps[i], HEAD_BN_EPS, &a)) { ok = false; break; }
            for (int c = 0; c < taps[i]; ++c) { sc[(size_t)hi * taps[i] + c] = (float)a.s[c]; sh[(size_t)hi * taps[i] + c] = (float)a.t[c]; }
        }
        bl.add_f32("heads.in_bn." + std::to_string(i) + ".scale", sc.data(), (int64_t)sc.size());
        bl.add_f32("heads.in_bn." + std::to_string(i) + ".shift", sh.data(), (int64_t)sh.size());
    }
    for (int i = 1; i < ntap && ok; ++i) {
        const int cin = FPN_DIM + taps[ntap - 1 - i];
        std::vector<double> wall, ball;
        for (int hi = 0; hi < NHEADS; ++hi) {
            const std::string name = HEADS[hi].name;
            if (!fold(w, name + ".upsamplers." + std::to_string(i) + ".0.weight", name + ".upsamplers." + std::to_string(i) + ".1", HEAD_BN_EPS,
                      FPN_DIM, cin, 3, &wf, &b)) { ok = false; break; }
            const std::vector<double> km = kmajor(wf, FPN_DIM, cin, 3);
            wall.insert(wall.end(), km.begin(), km.end());
            ball.insert(ball.end(), b.begin(), b.end());
        }
        if (!ok) break;
        bl.add_compute("heads.L" + std::to_string(i) + ".w", wall.data(), (int64_t)wall.size(), cdt);
        bl.add_f32("heads.L" + std::to_string(i) + ".b", ball.data(), (int64_t)ball.size());
    }
    if (ntap >= 2 && ok) {                  // (round 5: in the fp32 / fp16x3 plans too; the convolution then reads the fp32 tap itself)
        // Last level with the input BatchNorm of the backbone tap folded in exactly (as level 0 above): scale into the tap columns
        // of the weights, shift into a 16-case border bias table -- the convolution then reads the shared bf16 trunk copy of the
        // tap instead of nine batch-normed copies (FTC_FLAG_GROUP_IN2_SHARED + FTC_FLAG_BORDER_BIAS).
        const int i = ntap - 1, tc = taps[0], cin = FPN_DIM + tc;
        std::vector<double> wall, ball((size_t)NHEADS * 16 * FPN_DIM);
        for (int hi = 0; hi < NHEADS; ++hi) {
            const std::string name = HEADS[hi].name;
            BnAffine in;
            if (!bn_affine(w, name + ".in_bn.0", tc, HEAD_BN_EPS, &in) ||
                !fold(w, name + ".upsamplers." + std::to_string(i) + ".0.weight", name + ".upsamplers." + std::to_string(i) + ".1", HEAD_BN_EPS,
                      FPN_DIM, cin, 3, &wf, &b)) { ok = false; break; }
            std::vector<double> b16;
            border_bias16(wf, FPN_DIM, cin, FPN_DIM, in.t, b, &b16);
            for (int n = 0; n < FPN_DIM; ++n)
                for (int c = 0; c < tc; ++c) {
                    double* q = wf.data() + ((size_t)n * cin + FPN_DIM + c) * 9;
                    for (int t = 0; t < 9; ++t) q[t] *= in.s[c];
                }
            const std::vector<double> km = kmajor(wf, FPN_DIM, cin, 3);
            wall.insert(wall.end(), km.begin(), km.end());
            std::copy(b16.begin(), b16.end(), ball.begin() + (size_t)hi * 16 * FPN_DIM);
        }
        if (ok) {
            bl.add_compute("heads.L" + std::to_string(i) + "f.w", wall.data(), (int64_t)wall.size(), cdt);
            bl.add_f32("heads.L" + std::to_string(i) + "f.b", ball.data(), (int64_t)ball.size());         // [9][16][192]
            if (bf && cin % 64 == 0) {
                // the same weights FRAGMENT-MAJOR for the weights-through-L1 kernel (FTC_FLAG_W_FRAG): per head
                // [6 row blocks][9 taps][cin/64][4 K groups][64 lanes][8]: lane L, element e = W[32 rb + (L & 31)][tap][64 cb + 16 g + 8 (L >> 5) + e]
                const int ncb = cin / 64;
                std::vector<double> wf2(wall.size());
                const size_t per_head = (size_t)FPN_DIM * 9 * cin;
                for (int hi = 0; hi < NHEADS; ++hi) {
                    const double* km = wall.data() + hi * per_head;
                    double* dst = wf2.data() + hi * per_head;
                    for (int rb = 0; rb < 6; ++rb)
                        for (int t = 0; t < 9; ++t)
                            for (int cb = 0; cb < ncb; ++cb)
                                for (int g = 0; g < 4; ++g)
                                    for (int L = 0; L < 64; ++L) {
                                        const size_t src = ((size_t)(rb * 32 + (L & 31)) * 9 + t) * cin + cb * 64 + g * 16 + (L >> 5) * 8;
                                        const size_t d = (((((size_t)rb * 9 + t) * ncb + cb) * 4 + g) * 64 + L) * 8;
                                        for (int e = 0; e < 8; ++e) dst[d + e] = km[src + e];
                                    }
                }
                bl.add_compute("heads.L" + std::to_string(i) + "f.wfrag", wf2.data(), (int64_t)wf2.size(), cdt);
            }
        }
    }
    // top convolutions (3x3, with bias, no BN): K-major [co][9][192]
    auto top = [&](const std::string& name, int co, std::vector<double>* km, std::vector<float>* bias) -> bool {
        const TensorView *tw = w.get(name + ".top_conv.0.weight", {co, FPN_DIM, 3, 3}), *tb = w.get(name + ".top_conv.0.bias", {co});
        if (!tw || !tb) return false;
        std::vector<double> wd((size_t)co * FPN_DIM * 9);
        for (size_t i = 0; i < wd.size(); ++i) wd[i] = (double)tw->data[i];
        *km = kmajor(wd, co, FPN_DIM, 3);
        bias->assign(tb->data, tb->data + co);
        return true;
    };
    std::vector<double> km;
    std::vector<float> tb;
    for (int hi : {0, 1, 8}) {
        if (!ok || !top(HEADS[hi].name, HEADS[hi].out_dim, &km, &tb)) { ok = false; break; }
        bl.add_compute(std::string(HEADS[hi].name) + ".top_conv.w", km.data(), (int64_t)km.size(), cdt);
        bl.add_f32(std::string(HEADS[hi].name) + ".top_conv.b", tb.data(), (int64_t)tb.size());
    }
    if (ok) {   // the six one-channel heads whose heat-map channels are consecutive (textline, separator, code1/2/4/8 -> channels 4..9)
        std::vector<double> w6;
        std::vector<float> b6;
        for (int hi = 2; hi < 8; ++hi) {
            if (!top(HEADS[hi].name, 1, &km, &tb)) { ok = false; break; }
            w6.insert(w6.end(), km.begin(), km.end());
            b6.push_back(tb[0]);
        }
        if (ok) {
            bl.add_compute("heads.top6.w", w6.data(), (int64_t)w6.size(), cdt);
            bl.add_f32("heads.top6.b", b6.data(), (int64_t)b6.size());
        }
    }
    if (ok) {
        // The eight map heads' top convolutions as per-pixel tap matrices for the fused last-level epilogue (FTC_FLAG_TOP_FUSE +
        // FTC_OP_TAPSUM): row tap*Co + o of head g = top_conv weight [o, :, r, s], 32 rows zero padded.
        std::vector<double> wt((size_t)(NHEADS - 1) * 32 * FPN_DIM, 0.0);
        std::vector<float> bias;
        std::vector<int32_t> omap;
        for (int g = 0; g < NHEADS - 1; ++g) {
            const int co = HEADS[g].out_dim, ch0 = HEADS[g].ch0;
            if (!top(HEADS[g].name, co, &km, &tb)) { ok = false; break; }
            for (int o = 0; o < co; ++o) {
                for (int t = 0; t < 9; ++t)
                    for (int c = 0; c < FPN_DIM; ++c) wt[((size_t)g * 32 + t * co + o) * FPN_DIM + c] = km[((size_t)o * 9 + t) * FPN_DIM + c];
                bias.push_back(tb[o]);
                omap.insert(omap.end(), {g, o, co, (ch0 == 0 ? 0 : ch0 + 1) + o});
            }
        }
        if (ok) {
            if (!bf) {                           // fp32 / fp16x3 plans: the epilogue multiplies in fp32 FMA (conv_epilogue_topfuse_f32): a plain fp32 matrix
                std::vector<float> wf(wt.begin(), wt.end());
                bl.add_f32("heads.top8.wt", wf.data(), (int64_t)wf.size());
            } else {
                bl.add_compute("heads.top8.wt", wt.data(), (int64_t)wt.size(), cdt);
            }
            bl.add_f32("heads.top8.b", bias.data(), (int64_t)bias.size());
            std::memcpy(bl.add("heads.top8.map", (int64_t)omap.size() * 4), omap.data(), omap.size() * 4);
        }
    }
    if (!ok) return ftc_set_error(FTC_ERR_INVALID, "ftc_create: " + (w.missing.empty() ? std::string("weight packing failed") : w.missing));
    // SimpleDecoder (models/detector.py:232-254), optional: three MLPs Linear(100,2048,no bias) -> BatchNorm1d -> GELU -> Linear(2048,2048,
    // no bias) -> BatchNorm1d -> GELU -> Linear(2048, modulo).  Eval-mode BatchNorm1d (eps 1e-5) folds into the Linear before it; a Linear
    // weight [out][in] already is the K-major layout of a 1x1 convolution.  The first layer's K is zero-padded 100 -> 128.
    if (w.t.count("decoder.blocks.0.0.weight")) {
        for (int i = 0; i < 3 && ok; ++i) {
            const std::string p = "decoder.blocks." + std::to_string(i), q = "decoder." + std::to_string(i);
            const int mod = DECODER_MODULO[i];
            const TensorView *w0 = w.get(p + ".0.weight", {DECODER_MID, FEATURE_DIM}), *w1 = w.get(p + ".3.weight", {DECODER_MID, DECODER_MID}),
                             *w2 = w.get(p + ".6.weight", {mod, DECODER_MID}), *b2 = w.get(p + ".6.bias", {mod});
            BnAffine a0, a1;
            if (!w0 || !w1 || !w2 || !b2 || !bn_affine(w, p + ".1", DECODER_MID, HEAD_BN_EPS, &a0) || !bn_affine(w, p + ".4", DECODER_MID, HEAD_BN_EPS, &a1)) { ok = false; break; }
            std::vector<double> l0((size_t)DECODER_MID * DECODER_KPAD, 0.0), l1((size_t)DECODER_MID * DECODER_MID), l2((size_t)mod * DECODER_MID);
            for (int o = 0; o < DECODER_MID; ++o) {
                for (int c = 0; c < FEATURE_DIM; ++c) l0[(size_t)o * DECODER_KPAD + c] = (double)w0->data[(size_t)o * FEATURE_DIM + c] * a0.s[o];
                for (int c = 0; c < DECODER_MID; ++c) l1[(size_t)o * DECODER_MID + c] = (double)w1->data[(size_t)o * DECODER_MID + c] * a1.s[o];
            }
            for (size_t j = 0; j < l2.size(); ++j) l2[j] = (double)w2->data[j];
            bl.add_compute(q + ".l0.w", l0.data(), (int64_t)l0.size(), cdt);
            bl.add_f32(q + ".l0.b", a0.t.data(), DECODER_MID);
            bl.add_compute(q + ".l1.w", l1.data(), (int64_t)l1.size(), cdt);
            bl.add_f32(q + ".l1.b", a1.t.data(), DECODER_MID);
            bl.add_compute(q + ".l2.w", l2.data(), (int64_t)l2.size(), cdt);
            bl.add_f32(q + ".l2.b", b2->data, mod);
        }
        if (!ok) return ftc_set_error(FTC_ERR_INVALID, "ftc_create: decoder: " + (w.missing.empty() ? std::string("weight packing failed") : w.missing));
        m->has_decoder = true;
    }
    return FTC_OK;
}

// ---- measured kernel selection ----------------------------------------------------------------------
struct TuneEntry { const char* sig; int aux0; };
const TuneEntry kTuning[] = {
#include "tuning_table.inc"
    {nullptr, 0}};

std::string conv_signature(const ftc_op& o, bool strip_split = false) {
    char buf[192];
    // fp16 operands run the same kernels at the same rate as bf16: they share the measured table (dtype 2 looks up as 1)
    auto d = [](int dt) { return dt == FTC_F16 ? (int)FTC_BF16 : dt; };
    int n = std::snprintf(buf, sizeof buf, "w%di%do%d_B%d_%dx%d_c%dof%d_n%dof%d_k%ds%d_f%d_a%d", d(o.w_dtype), d(o.in_dtype), d(o.out_dtype), o.B, o.H, o.W,
                          o.Cin, o.Cin_total, o.Cout, o.Cout_total, o.ksize, o.stride, (strip_split ? (o.flags & ~FTC_FLAG_SPLIT16) : o.flags) & ~(FTC_FLAG_KBLOCK32 | FTC_FLAG_PRESPLIT), o.act);      // (KBLOCK32: where the 16-bit copy goes, not which kernel is fastest)
    if (o.groups > 1) std::snprintf(buf + n, sizeof buf - n, "_g%d", o.groups);
    return buf;
}

void apply_tuning(std::vector<ftc_op>& ops) {
    if (env_on("FTC_NO_TUNING")) return;
    static std::map<std::string, int> table;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const TuneEntry* e = kTuning; e->sig; ++e) table[e->sig] = e->aux0;
        // FTC_TUNING_OVERRIDE=<file of "signature aux0" lines>: entries replacing / extending the compiled-in table (tuning experiments without
        // a rebuild, e.g. tools/lanes_tuning.py: choices measured for the two-lane steady state instead of the kernel alone)
        if (const char* path = std::getenv("FTC_TUNING_OVERRIDE")) {
            if (FILE* f = std::fopen(path, "r")) {
                char sig[192];
                int v;
                int n = 0;
                while (std::fscanf(f, "%191s %d", sig, &v) == 2) { table[sig] = v; ++n; }
                std::fclose(f);
                std::fprintf(stderr, "[ftc] FTC_TUNING_OVERRIDE: %d entries from %s\n", n, path);
            }
        }
    });
    for (ftc_op& o : ops) {
        if (o.kind != FTC_OP_CONV) continue;
        auto it = table.find(conv_signature(o));
        if (it == table.end() && (o.flags & FTC_FLAG_SPLIT16)) it = table.find(conv_signature(o, true));     // fp16x3 without its own measurement: the fp32 choice
        if (it == table.end() || !it->second) continue;
        // A table entry is a HINT measured for one flag combination; the signature drops flags that do not change which kernel is fastest
        // (KBLOCK32, PRESPLIT) and fp16x3 falls back to the fp32 entry, so an entry can name a kernel that is not legal for THIS op
        // (e.g. a 144-pixel fp16x3 tile without pre-split operands under FTC_NO_PRESPLIT / FTC_NO_MBSLICE_X3).  Adopt it only if the op
        // validates with it; otherwise the default selection stands.
        const int keep = o.aux0;
        o.aux0 = it->second;
        if (conv_validate(o) != nullptr) o.aux0 = keep;
    }
}

// ---- plan builder (once per input shape) ---------------------------------------------------------------
struct Buf { int64_t nbytes; int first = 1 << 30, last = -1; int64_t offset = -1; };
struct R {                    // symbolic operand: arena buffer (+ byte offset inside it) | weights | input | outputs
    int kind = 0;             // 0 none, 1 buf, 2 weights, 3 input, 4 heatmap, 5 features
    int64_t v = 0, extra = 0;
    explicit operator bool() const { return kind != 0; }
};
struct SymOp {
    ftc_op o{};
    R in, in2, out, w, w2, bias, bias2, scale, shift, aux, out2;
};

struct ConvOpt {
    int cout_total = 0, cout_off = 0;
    R residual, se, out2, wsets;
    int res_dt = 0, extra_flags = 0, groups = 1;
    int64_t w_off = 0, b_off = 0;
};

class Builder {
public:
    Builder(ftc_model* m, int B, int H, int W, bool nchw) : m_(m), B(B), H(H), W(W), nchw_(nchw) {
        bf_ = m->precision != FTC_F32;            // 16-bit speed mode (bf16 or fp16 operands)
        act_ = m->precision;
        cdt_ = act_;
    }
    int build(ModelPlan* out);
    int build_decoder(ModelPlan* out);          // constructed with B = 1, H = rows, W = 1

private:
    ftc_model* m_;
    int B, H, W;
    bool nchw_, bf_;
    int act_, cdt_;
    const int trunk_ = FTC_F32;                // residual trunk + taps stay fp32
    std::vector<SymOp> ops_;
    std::vector<OpMeta> meta_;
    std::vector<Buf> bufs_;
    std::string err_;

    static int esize(int dt) { return dt == FTC_F32 ? 4 : 2; }
    R buf(int64_t nelem, int dt) { bufs_.push_back({align_up(nelem * esize(dt))}); return {1, (int64_t)bufs_.size() - 1, 0}; }
    R sub(const R& b, int64_t off) const { return {1, b.v, b.extra + off}; }
    R wref(const std::string& name, int64_t off = 0) {
        auto it = m_->blob.table.find(name);
        if (it == m_->blob.table.end()) { if (err_.empty()) err_ = "packed weight '" + name + "' missing"; return {}; }
        return {2, it->second + off, 0};
    }
    bool has_w(const std::string& name) const { return m_->blob.table.count(name) != 0; }
    void emit(const OpMeta& meta, const SymOp& s) {
        const int idx = (int)ops_.size();
        for (const R* r : {&s.in, &s.in2, &s.out, &s.aux, &s.scale, &s.out2, &s.w, &s.w2})
            if (r->kind == 1) { Buf& b = bufs_[r->v]; b.first = std::min(b.first, idx); b.last = std::max(b.last, idx); }
        ops_.push_back(s);
        meta_.push_back(meta);
    }
    void conv(const std::string& name, R x, int xdt, int h, int w, int cin, int cin_total, int cin_off, const std::string& wname, int cout, int k,
              int stride, int act, R out, int odt, const ConvOpt& c = ConvOpt()) {
        const int Ho = (h - 1) / stride + 1, Wo = (w - 1) / stride + 1;
        int flags = (c.residual ? FTC_FLAG_RESIDUAL : 0) | (c.se ? FTC_FLAG_SE_SCALE : 0) | c.extra_flags | (m_->split16 && cdt_ == FTC_F32 ? FTC_FLAG_SPLIT16 : 0);
        const double macs = (double)c.groups * B * Ho * Wo * cout * cin * k * k;
        double byt = (double)c.groups * ((double)B * h * w * cin * esize(xdt) + (double)B * Ho * Wo * cout * esize(odt) + (double)cout * cin * k * k * esize(cdt_));
        if (c.residual) byt += (double)B * Ho * Wo * cout * esize(c.res_dt);
        if (c.out2) byt += (double)B * Ho * Wo * cout * 2;
        if (c.wsets) { flags |= FTC_FLAG_W_PER_IMAGE; byt += (double)(B - 1) * cout * cin * k * k * esize(cdt_); }
        SymOp s;
        ftc_op& o = s.o;
        o.kind = FTC_OP_CONV; o.flags = flags; o.act = act; o.in_dtype = xdt; o.out_dtype = odt; o.w_dtype = cdt_;
        o.B = B; o.H = h; o.W = w; o.Ho = Ho; o.Wo = Wo; o.Cin = cin; o.Cin_total = cin_total; o.cin_off = cin_off;
        o.Cout = cout; o.Cout_total = c.cout_total ? c.cout_total : cout; o.cout_off = c.cout_off; o.ksize = k; o.stride = stride;
        o.res_dtype = c.res_dt; o.groups = c.groups > 1 ? c.groups : 0;
        s.in = x; s.in2 = c.residual; s.out = out; s.w = c.wsets ? c.wsets : wref(wname + ".w", c.w_off); s.bias = wref(wname + ".b", c.b_off);
        s.scale = c.se; s.out2 = c.out2;
        emit({name, "conv" + std::to_string(k) + "x" + std::to_string(k), 2.0 * macs, byt}, s);
    }
    int finish(ModelPlan* out, int mh, int mw);
};

int Builder::build(ModelPlan* out) {
    const std::string& ms = m_->size;
    const auto stages = backbone_blocks(ms);
    const int c0 = stage_rows(ms)[0].cin;
    const int T = trunk_, A = act_;
    // In bf16 mode every trunk tensor (fp32, feeds the residual adds and the FPN taps) is written together with a bf16 copy by
    // the producing epilogue; the next GEMM reads the copy.
    const bool dual = bf_;
    const int G = dual ? A : T;                // dtype the GEMMs read the trunk in
    // fp16x3 plan (round 5): a block whose output feeds a fused MBConv head gets a second, PRE-SPLIT copy of its fp32 trunk tensor (hi | lo halves
    // per 16-byte chunk: what csrc/mbconv_slice_x3.hip streams by DMA) -- `want_copy`
    const bool x3 = m_->split16 && cdt_ == FTC_F32;
    auto trunk = [&](int64_t nelem, R* t, R* tb, bool want_copy = false) { *t = buf(nelem, T); *tb = dual ? buf(nelem, A) : (x3 && want_copy) ? buf(nelem, FTC_F32) : R(); };

    int h = (H - 1) / 2 + 1, w = (W - 1) / 2 + 1;
    R x, xb;
    trunk((int64_t)B * h * w * c0, &x, &xb);
    {
        SymOp s;
        ftc_op& o = s.o;
        o.kind = FTC_OP_STEM; o.flags = nchw_ ? FTC_FLAG_IN_NCHW : 0; o.act = FTC_ACT_SILU; o.in_dtype = FTC_F32; o.out_dtype = T;
        o.w_dtype = dual ? A : 0;                  // STEM: dtype of the 16-bit trunk copy (out2)
        o.B = B; o.H = H; o.W = W; o.Ho = h; o.Wo = w; o.Cin = 3; o.Cout = c0; o.ksize = 3; o.stride = 2;
        s.in = {3, 0, 0}; s.out = x; s.out2 = xb; s.w = wref("stem.w"); s.bias = wref("stem.b");
        emit({"backbone.features.0", "stem", 2.0 * B * h * w * c0 * 27, (double)B * H * W * 3 * 4 + (double)B * h * w * c0 * (esize(T) + (dual ? 2 : 0))}, s);
    }
    struct Tap { R buf; int c, h, w, dt; };
    std::vector<Tap> taps;
    std::vector<R> tap_copies;                  // bf16 trunk copies of the backbone taps (bf16 mode), same order
    // Low-resolution MBConv stages (24x24 maps at 768x768: stages 6-7), 16-bit plans: expand + depthwise + squeeze + the block's share of
    // the SE fc1 layer in ONE launch, a workgroup per (image, 128 expanded channels) -- csrc/mbconv_slice.hip; the expanded tensor never
    // leaves the CU.  FTC_NO_MBSLICE=1: the three-kernel form; FTC_MBSLICE_MINWG: workgroups below which the three-kernel form is kept
    // (small batches leave most CUs without a slice).  bh, bw = the block's INPUT map.
    // (maps of more than 576 pixels -- the 48x48 stages 4-5 -- run in bands of R output rows: returns R, 0 = the whole map, -1 = not sliced)
    // Slice width of a block's fused head: 64 (fp32 tensors), 128, or -- whole-map blocks whose 128-channel slices leave more than a fifth of the 256
    // CUs without a workgroup while 96-channel slices still fit one round (stage 6 at batch 8: 192 -> 256 workgroups) -- 96.  FTC_MBSLICE_96=0: never.
    auto mb_slice_of = [&](const BlockSpec& blk, int bh, int bw) -> int {
        if (x3) return FTC_MBHEAD_SLICE_F32;
        if (ftc_mbhead_band_rows(bh, bw) != 0 || blk.exp % 96 != 0) return FTC_MBHEAD_SLICE;
        const char* e96 = std::getenv("FTC_MBSLICE_96");
        if (e96 && e96[0] == '0') return FTC_MBHEAD_SLICE;
        const int w128 = B * (blk.exp / 128), w96 = B * (blk.exp / 96);
        return (w128 <= 204 && w96 <= 256) ? 96 : FTC_MBHEAD_SLICE;
    };
    auto sliced = [&](const BlockSpec& blk, int bh, int bw) -> int {
        if (blk.fused || !(dual || x3) || blk.stride != 1 || blk.squeeze > FTC_MBHEAD_MAX_SQUEEZE || env_on("FTC_NO_MBSLICE")) return -1;
        if (x3 && env_on("FTC_NO_MBSLICE_X3")) return -1;
        const int R = ftc_mbhead_band_rows(bh, bw);
        if (R < 0 || (R > 0 && env_on("FTC_NO_MBBAND"))) return -1;
        ftc_op t{};
        t.in_dtype = t.out_dtype = t.w_dtype = A; t.stride = blk.stride; t.ksize = 3; t.H = t.Ho = bh; t.W = t.Wo = bw;
        t.Cin = blk.cin; t.Cout = blk.exp; t.aux1 = R; t.flags = x3 ? FTC_FLAG_SPLIT16 : 0;
        const int mb_slice = mb_slice_of(blk, bh, bw);
        t.Cout_total = mb_slice;
        const char* mw_env = std::getenv("FTC_MBSLICE_MINWG");
        const int min_wg = mw_env ? std::atoi(mw_env) : 128;
        // (fp16x3, stage 5 at batch 8 -- 960 workgroups, every 64-channel slice re-streams its image's x: 143 us against 80 + 57 for the two kernels it
        //  replaces; kept all the same: its pre-split output saves the project convolution 10 us and the pair moves 113 MB less through HBM)
        return ftc_mbhead_legal(t) && B * ftc_mbhead_bands(t) * (blk.exp / mb_slice) >= min_wg ? R : -1;
    };
    // Fused-MBConv blocks with expansion, 16-bit plans, stride 1: one launch (FTC_OP_FMBCONV) where the shape is one the kernel holds
    auto fmb_fused = [&](const BlockSpec& blk, int bh, int bw) -> bool {
        if (!blk.fused || blk.exp == blk.cin || !(dual || x3) || blk.stride != 1 || env_on("FTC_NO_FMBFUSE")) return false;
        if (x3 && env_on("FTC_NO_FMBFUSE_X3")) return false;
        // Measured (tools/fmbconv_bench.py, batch 8): the one-launch form wins where the 3x3 runs K steps of 64 (Cin % 64 == 0: stage 2, 174-182 us
        // against 126 + 65) and loses on stage 3 (Cin = 96: K steps of 32, twice the barriers per FLOP: 116-135 us against 75 + 28.5).
        // FTC_FMBFUSE_ALL=1: every shape the kernel holds.
        if (blk.cin % 64 != 0 && !env_on("FTC_FMBFUSE_ALL")) return false;
        ftc_op t{};
        t.in_dtype = t.w_dtype = x3 ? FTC_F32 : A; t.out_dtype = T; t.res_dtype = T; t.ksize = 3; t.stride = 1; t.H = t.Ho = bh; t.W = t.Wo = bw; t.B = B;
        t.Cin = t.Cin_total = blk.cin; t.Cout = t.Cout_total = blk.cout; t.aux1 = blk.exp; t.act = FTC_ACT_SILU;
        t.flags = (blk.residual ? FTC_FLAG_RESIDUAL : 0) | (x3 ? FTC_FLAG_SPLIT16 : 0);
        return ftc_fmbconv_legal(t);
    };
    std::vector<const BlockSpec*> flat;
    for (const auto& st : stages)
        for (const BlockSpec& blk : st) flat.push_back(&blk);
    size_t bi = 0;
    bool in_blocked = false;                    // the 16-bit trunk copy feeding the current block is in 32-channel planes (FTC_FLAG_KBLOCK32)
    for (size_t si = 0; si < stages.size(); ++si) {
        for (const BlockSpec& blk : stages[si]) {
            const std::string p = blk.prefix + ".block";
            const int ho = (h - 1) / blk.stride + 1, wo = (w - 1) / blk.stride + 1;
            const R res = blk.residual ? x : R();
            const R gin = dual ? xb : x;        // GEMM-side view of the block input
            R y, yb;
            ++bi;
            const bool next_sliced = bi < flat.size() && sliced(*flat[bi], ho, wo) >= 0;
            trunk((int64_t)B * ho * wo * blk.cout, &y, &yb, next_sliced);
            ConvOpt tail;
            tail.residual = res; tail.res_dt = T; tail.out2 = yb;
            // the consumer of this block's 16-bit copy is the next block's expand GEMM: FTC_OP_MBHEAD streams it in 32-channel planes
            const bool out_blocked = dual && next_sliced && blk.cout % 32 == 0 && !env_on("FTC_NO_KBLOCK");
            if (out_blocked) tail.extra_flags |= FTC_FLAG_KBLOCK32;
            if (blk.fused && blk.exp == blk.cin) {
                conv(p + ".0", gin, G, h, w, blk.cin, blk.cin, 0, p + ".0", blk.cout, 3, blk.stride, FTC_ACT_SILU, y, T, tail);
            } else if (blk.fused && fmb_fused(blk, h, w)) {
                // Round 6: the whole block in one launch (csrc/fused_mbconv.hip): the expanded tensor (151 MB per stage-2 block at batch 8) is
                // neither written nor read back.  FTC_NO_FMBFUSE=1: the two-launch form below.
                SymOp s;
                ftc_op& o = s.o;
                o.kind = FTC_OP_FMBCONV; o.act = FTC_ACT_SILU; o.in_dtype = G; o.out_dtype = T; o.w_dtype = cdt_; o.res_dtype = T;
                o.flags = (res ? FTC_FLAG_RESIDUAL : 0) | (x3 ? FTC_FLAG_SPLIT16 : 0);
                o.B = B; o.H = h; o.W = w; o.Ho = ho; o.Wo = wo; o.Cin = blk.cin; o.Cin_total = blk.cin; o.Cout = blk.cout; o.Cout_total = blk.cout;
                o.ksize = 3; o.stride = 1; o.aux1 = blk.exp;
                s.in = gin; s.in2 = res; s.out = y; s.out2 = yb; s.w2 = wref(p + ".0.w"); s.bias2 = wref(p + ".0.b"); s.w = wref(p + ".1.w"); s.bias = wref(p + ".1.b");
                const double px = (double)B * h * w;
                emit({p + ".0+1", "conv3x3+conv1x1", 2.0 * px * blk.exp * (9.0 * blk.cin + blk.cout),
                      px * blk.cin * esize(G) + px * blk.cout * (esize(T) * (res ? 2 : 1) + (yb ? 2 : 0)) + (double)blk.exp * (9.0 * blk.cin + blk.cout) * esize(cdt_)}, s);
            } else if (blk.fused) {
                const R e = buf((int64_t)B * ho * wo * blk.exp, A);
                conv(p + ".0", gin, G, h, w, blk.cin, blk.cin, 0, p + ".0", blk.exp, 3, blk.stride, FTC_ACT_SILU, e, A);
                conv(p + ".1", e, A, ho, wo, blk.exp, blk.exp, 0, p + ".1", blk.cout, 1, 1, FTC_ACT_NONE, y, T, tail);
            } else {
                const int band_rows = (x3 && !xb) ? -1 : sliced(blk, h, w);      // (fp16x3: the head streams the PRE-SPLIT copy of its input)
                const bool slice = band_rows >= 0;
                const int nbands = band_rows > 0 ? (h + band_rows - 1) / band_rows : 1;
                const int th = blk.stride == 1 ? 8 : 4;
                const int mb_slice = mb_slice_of(blk, h, w);
                const int P = slice ? nbands * (blk.exp / mb_slice) : ((ho + th - 1) / th) * ((wo + 7) / 8);
                const R d = buf((int64_t)B * ho * wo * blk.exp, A);
                const R part = buf((int64_t)B * P * (slice ? blk.squeeze : blk.exp), FTC_F32);
                if (slice) {
                    const R sums = buf((int64_t)B * nbands * blk.exp, FTC_F32);
                    SymOp s;
                    ftc_op& o = s.o;
                    o.kind = FTC_OP_MBHEAD; o.act = FTC_ACT_SILU; o.in_dtype = A; o.out_dtype = A; o.w_dtype = A; o.B = B; o.H = h; o.W = w; o.Ho = ho; o.Wo = wo;
                    o.Cin = blk.cin; o.Cout = blk.exp; o.Cout_total = mb_slice; o.ksize = 3; o.stride = 1; o.aux0 = blk.squeeze; o.aux1 = band_rows;
                    // fp16x3: the head writes d PRE-SPLIT when its only reader, the project convolution, runs on folded weights (no SE scale on the activations)
                    const bool d_presplit = x3 && !env_on("FTC_NO_X3FOLD") && (ho * wo) % 64 == 0 && blk.exp % 8 == 0 && !env_on("FTC_NO_PRESPLIT");
                    o.flags = (in_blocked ? FTC_FLAG_KBLOCK32 : 0) | (x3 ? FTC_FLAG_SPLIT16 : 0) | (d_presplit ? FTC_FLAG_PRESPLIT : 0);
                    s.in = x3 ? xb : gin; s.out = d; s.w2 = wref(p + ".0.w"); s.bias2 = wref(p + ".0.b"); s.w = wref(p + ".1.w"); s.bias = wref(p + ".1.b"); s.aux = sums;
                    s.scale = wref(p + ".2.w1"); s.out2 = part;
                    emit({p + ".0+1", "conv1x1+dw3x3", 2.0 * B * h * w * blk.exp * (blk.cin + 9),
                          (double)B * h * w * (blk.cin + blk.exp) * esize(A) + (double)blk.exp * blk.cin * esize(A) + blk.exp * 44.0 + 4.0 * blk.exp * blk.squeeze}, s);
                } else {
                const R e = buf((int64_t)B * h * w * blk.exp, A);
                conv(p + ".0", gin, G, h, w, blk.cin, blk.cin, 0, p + ".0", blk.exp, 1, 1, FTC_ACT_SILU, e, A);
                {
                    SymOp s;
                    ftc_op& o = s.o;
                    o.kind = FTC_OP_DWCONV; o.act = FTC_ACT_SILU; o.in_dtype = A; o.out_dtype = A; o.B = B; o.H = h; o.W = w; o.Ho = ho; o.Wo = wo;
                    o.Cin = blk.exp; o.Cout = blk.exp; o.ksize = 3; o.stride = blk.stride; o.aux0 = P;
                    s.in = e; s.out = d; s.w = wref(p + ".1.w"); s.bias = wref(p + ".1.b"); s.aux = part;
                    emit({p + ".1", "dwconv3x3", 2.0 * B * ho * wo * blk.exp * 9, (double)B * ((double)h * w + (double)ho * wo) * blk.exp * esize(A) + blk.exp * 40.0}, s);
                }
                }
                const R sc = buf((int64_t)B * blk.exp, FTC_F32);
                const R hid = buf((int64_t)B * blk.squeeze, FTC_F32);
                // bf16 mode: the SE op also writes the project weights scaled per image, so that the project convolution streams both
                // operands by DMA instead of rescaling activations while staging them.  Needs a 64-pixel tile that divides the image.
                // (fp16x3 plan: the same with pre-split fp32 chunks -- FTC_NO_X3FOLD=1 keeps the gate in the project convolution's staging)
                const bool x3fold = m_->split16 && cdt_ == FTC_F32 && !env_on("FTC_NO_X3FOLD");
                const bool foldse = (dual || x3fold) && (ho * wo) % 64 == 0 && blk.exp % 8 == 0;
                const int fdt = dual ? A : FTC_F32;
                const R wb = foldse ? buf((int64_t)B * blk.cout * blk.exp, fdt) : R();
                {
                    SymOp s;
                    ftc_op& o = s.o;
                    o.kind = FTC_OP_SE; o.flags = (foldse ? FTC_FLAG_SE_FOLD | (dual ? 0 : FTC_FLAG_SPLIT16) : 0) | (slice ? FTC_FLAG_SE_HPART : 0); o.w_dtype = foldse ? fdt : 0; o.B = B; o.H = ho; o.W = wo;
                    o.Cin = blk.exp; o.Cout = blk.exp; o.Cout_total = foldse ? blk.cout : 0; o.aux0 = blk.squeeze; o.aux1 = P;
                    s.aux = part; s.out = sc; s.in2 = hid; s.w = wref(p + ".2.w1"); s.w2 = wref(p + ".2.w2t"); s.bias = wref(p + ".2.b1");
                    s.bias2 = wref(p + ".2.b2"); s.in = foldse ? wref(p + ".3.w") : R(); s.out2 = wb;
                    const double se_bytes = (slice ? 4.0 : 8.0) * blk.exp * blk.squeeze + (double)B * P * (slice ? blk.squeeze : blk.exp) * 4 +
                                            (foldse ? (double)(B + 1) * blk.cout * blk.exp * esize(fdt) : 0.0);
                    emit({p + ".2", "se", 4.0 * B * blk.exp * blk.squeeze, se_bytes}, s);
                }
                ConvOpt pj = tail;
                pj.se = foldse ? R() : sc;
                pj.wsets = wb;
                if (x3 && slice && foldse && !env_on("FTC_NO_PRESPLIT")) pj.extra_flags |= FTC_FLAG_PRESPLIT;      // d was written pre-split by the fused head
                conv(p + ".3", d, A, ho, wo, blk.exp, blk.exp, 0, p + ".3", blk.cout, 1, 1, FTC_ACT_NONE, y, T, pj);
            }
            x = y; xb = yb; h = ho; w = wo;
            in_blocked = out_blocked;
        }
        if (si + 1 == 2 || si + 1 == 3 || si + 1 == 5) {          // BackboneModel.forward taps (models/detector.py:143)
            taps.push_back({x, stages[si].back().cout, h, w, T});
            tap_copies.push_back(xb);
        }
    }
    const int nfeat = (int)stages.size() + 1;
    const std::string hp = "backbone.features." + std::to_string(nfeat);
    const int clast = stages.back().back().cout;
    const R x4 = buf((int64_t)B * h * w * LAST_CHANNEL, A);
    conv(hp, dual ? xb : x, G, h, w, clast, clast, 0, hp, LAST_CHANNEL, 1, 1, FTC_ACT_SILU, x4, A);
    taps.push_back({x4, LAST_CHANNEL, h, w, A});
    const int mh = taps[0].h, mw = taps[0].w;
    // heads.  Level 0 of all nine heads is one convolution (see pack_weights); levels 1.. are ONE grouped launch each (upsample+concat,
    // then the 3x3 convolution) over head-major stacked tensors [9][B,h,w,C].
    const int ntap = (int)taps.size(), nh = NHEADS;
    const Tap t4 = taps[ntap - 1];
    const R y0 = buf((int64_t)B * t4.h * t4.w * nh * FPN_DIM, A);
    {
        ConvOpt c;
        c.extra_flags = FTC_FLAG_BORDER_BIAS;
        conv("heads.upsamplers.0", t4.buf, t4.dt, t4.h, t4.w, t4.c, t4.c, 0, "heads.L0", nh * FPN_DIM, 3, 1, FTC_ACT_GELU, y0, A, c);
    }
    R y = y0;
    int yh = t4.h, yw = t4.w;
    const int nmap = nh - 1;                   // the map heads (all but `feature`)
    // (round 5: in the fp32 / fp16x3 plans too -- fp32 FMA epilogue, conv_epilogue_topfuse_f32; FTC_NO_TOPFUSE32=1: the two-kernel form)
    const bool fuse_top = (dual || (cdt_ == FTC_F32 && !env_on("FTC_NO_TOPFUSE32"))) && taps[0].c + FPN_DIM == 256 && !env_on("FTC_NO_TOPFUSE");
    const int TW = 20;                         // floats per pixel of the tap tensor T (9 * 2 outputs, padded)
    // (round 3: also in the fp32 / fp16x3 plans, whose last-level concatenated input is 2.7 GB: FTC_NO_UPFUSE32=1 materialises it)
    const bool fuse_up = (dual || (cdt_ == FTC_F32 && !env_on("FTC_NO_UPFUSE32"))) && !env_on("FTC_NO_UPFUSE");
    bool heads_done = false;
    for (int i = 1; i < ntap; ++i) {
        const Tap& tp = taps[ntap - 1 - i];
        const int tc = tp.c, th_ = tp.h, tw_ = tp.w, tdt = tp.dt;
        const int cy = FPN_DIM, cin = cy + tc;
        const int64_t M = (int64_t)B * th_ * tw_;
        const bool last = i == ntap - 1;
        const std::string bi = std::to_string(ntap - 1 - i);
        const R bn_s = wref("heads.in_bn." + bi + ".scale"), bn_t = wref("heads.in_bn." + bi + ".shift");
        const int64_t wsz = (int64_t)FPN_DIM * cin * 9 * esize(cdt_);
        // bf16 mode, levels whose upsampled source is a stacked tensor (2..): the concatenated input is never materialised -- the
        // convolution upsamples while it stages its halo (FTC_FLAG_UPCAT_IN).  (measured: with 32-channel K blocks the per-block
        // upsampling work outweighs the saved pass, so Cin 288 keeps the two-kernel form)
        // (fp32 / fp16x3 plans: the halo kernel's K block is 32 channels of 4 bytes, and the three-MFMA product makes the upsampling work per block
        // a smaller share: level 2 -- 192 + 96 channels -- goes too; FTC_NO_UPFUSE32_L2=1: as before)
        const int kblk = (dual || env_on("FTC_NO_UPFUSE32_L2")) ? 64 : 32;
        const bool up_in = fuse_up && i >= 2 && th_ == 2 * yh && tw_ == 2 * yw && cy % kblk == 0 && tc % kblk == 0;
        // ... and on the last level the tap's BatchNorm is folded into the weights + a border bias table, so that all heads read the ONE
        // bf16 trunk copy of the tap
        // (fp32 / fp16x3 plans: the tap itself -- fp32 NHWC, what the halo loader reads)
        const R tap_copy = !dual ? ((tdt == A && !env_on("FTC_NO_BNFOLD32")) ? tp.buf : R()) : (ntap - 1 - i) < (int)tap_copies.size() ? tap_copies[ntap - 1 - i] : R();
        const std::string lf = "heads.L" + std::to_string(i) + "f";
        const bool bn_fold = up_in && last && tap_copy && has_w(lf + ".w") && !env_on("FTC_NO_BNFOLD");
        R tapbn, cat;
        double src_bytes;
        if (bn_fold) {
            tapbn = tap_copy;
            src_bytes = (double)B * yh * yw * cy * esize(A) + (double)M * tc * esize(A) / nh;
        } else if (up_in) {
            tapbn = buf((int64_t)nh * M * tc, A);
            SymOp s;
            ftc_op& o = s.o;
            o.kind = FTC_OP_UPCAT; o.in_dtype = A; o.out_dtype = A; o.res_dtype = tdt; o.B = B; o.H = th_; o.W = tw_; o.Ho = th_; o.Wo = tw_;
            o.Cin = tc; o.Cout = tc; o.aux0 = 0; o.aux1 = tc; o.groups = nh;
            s.in2 = tp.buf; s.out = tapbn; s.scale = bn_s; s.shift = bn_t;
            emit({"heads.tapbn" + std::to_string(i), "upcat", 0.0, (double)nh * M * tc * esize(A) + (double)M * tc * esize(tdt)}, s);
            src_bytes = (double)B * yh * yw * cy * esize(A) + (double)M * tc * esize(A);
        } else {
            cat = buf((int64_t)nh * M * cin, A);
            SymOp s;
            ftc_op& o = s.o;
            o.kind = FTC_OP_UPCAT; o.flags = i == 1 ? FTC_FLAG_GROUP_IN_SLICE : 0; o.in_dtype = A; o.out_dtype = A; o.res_dtype = tdt; o.B = B;
            o.H = yh; o.W = yw; o.Ho = th_; o.Wo = tw_; o.Cin = cin; o.Cin_total = i == 1 ? nh * FPN_DIM : FPN_DIM; o.cin_off = 0; o.Cout = cin;
            o.aux0 = cy; o.aux1 = tc; o.groups = nh;
            s.in = y; s.in2 = tp.buf; s.out = cat; s.scale = bn_s; s.shift = bn_t;
            emit({"heads.cat" + std::to_string(i), "upcat", 0.0,
                  (double)nh * ((double)M * ((double)cin * esize(A) + (double)tc * esize(tdt)) + (double)B * yh * yw * cy * esize(A))}, s);
            src_bytes = (double)M * cin * 2;
        }
        // groups [g0, g0+ng) of level i; `top`: fused top convolution (out = T) instead of the 192-channel output
        auto level_conv = [&](const std::string& name, int g0, int ng, R outr, bool top) {
            const std::string wname = bn_fold ? lf : "heads.L" + std::to_string(i);
            const int brows = bn_fold ? 16 : 1;
            SymOp s;
            ftc_op& o = s.o;
            o.kind = FTC_OP_CONV; o.act = FTC_ACT_GELU; o.in_dtype = A; o.out_dtype = A; o.w_dtype = cdt_; o.B = B; o.H = th_; o.W = tw_;
            o.Ho = th_; o.Wo = tw_; o.Cin = cin; o.Cout = FPN_DIM; o.Cout_total = FPN_DIM; o.ksize = 3; o.stride = 1; o.groups = ng > 1 ? ng : 0;
            s.out = outr;
            s.w = wref(wname + ".w", (int64_t)g0 * wsz);
            s.bias = wref(wname + ".b", (int64_t)g0 * brows * FPN_DIM * 4);
            int flags = (m_->split16 && cdt_ == FTC_F32) ? FTC_FLAG_SPLIT16 : 0;
            // weights-through-L1 kernel for the fused last level (fragment-major copy of the folded weights); FTC_NO_WL1=1: the LDS-ring halo kernel
            const bool wl1 = bn_fold && has_w(lf + ".wfrag") && !env_on("FTC_NO_WL1");
            if (bn_fold) {
                flags |= FTC_FLAG_UPCAT_IN | FTC_FLAG_BORDER_BIAS | FTC_FLAG_GROUP_IN2_SHARED;
                o.Cin_total = cy; o.aux0 = 65;
                s.in = sub(y, (int64_t)g0 * B * yh * yw * cy * esize(A)); s.in2 = tapbn;
                if (wl1) { flags |= FTC_FLAG_W_FRAG; s.w = wref(lf + ".wfrag", (int64_t)g0 * wsz); }
            } else if (up_in) {
                flags |= FTC_FLAG_UPCAT_IN;
                o.Cin_total = cy; o.aux0 = 65;
                s.in = sub(y, (int64_t)g0 * B * yh * yw * cy * esize(A)); s.in2 = sub(tapbn, (int64_t)g0 * M * tc * esize(A));
            } else {
                o.Cin_total = cin;
                s.in = sub(cat, (int64_t)g0 * M * cin * esize(A));
            }
            double flops = 2.0 * ng * M * FPN_DIM * cin * 9;
            double byt = ng * (src_bytes + (double)FPN_DIM * cin * 9 * esize(cdt_));
            if (top) {
                flags |= FTC_FLAG_TOP_FUSE;
                int nout = 0;
                for (int g = 0; g < nmap; ++g) nout += HEADS[g].out_dim;
                o.aux0 = 65; o.aux1 = TW;
                s.w2 = wref("heads.top8.wt");
                flops += 2.0 * M * FPN_DIM * 9 * nout;
                byt += (double)ng * M * TW * 4;
            } else {
                byt += (double)ng * M * FPN_DIM * esize(A);
            }
            o.flags = flags;
            if (wl1) o.aux0 |= 128;
            emit({name, "conv3x3", flops, byt}, s);
        };
        if (last && fuse_top) {
            // Last level, bf16: the eight map heads never store their 192-channel output -- the epilogue multiplies the tile by the head's
            // top-convolution taps and stores 20 floats per pixel; TAPSUM does the 9-point sum into the heat-map channels.  The feature
            // head (100 output channels) keeps the two-kernel form.
            const R Tt = buf((int64_t)nmap * M * TW, FTC_F32);
            int nout = 0;
            for (int g = 0; g < nmap; ++g) nout += HEADS[g].out_dim;
            level_conv("heads.upsamplers." + std::to_string(i) + "+top", 0, nmap, Tt, true);
            {
                SymOp s;
                ftc_op& o = s.o;
                o.kind = FTC_OP_TAPSUM; o.B = B; o.H = th_; o.W = tw_; o.Ho = th_; o.Wo = tw_; o.Cout_total = 10; o.aux0 = TW; o.aux1 = nout; o.groups = nmap;
                s.in = Tt; s.out = {4, 0, 0}; s.w = wref("heads.top8.map"); s.bias = wref("heads.top8.b");
                emit({"heads.top8.tapsum", "tapsum", 0.0, (double)nmap * M * TW * 4 + (double)M * nout * 4}, s);
            }
            const R yf = buf(M * FPN_DIM, A);
            level_conv("feature.upsamplers." + std::to_string(i), nh - 1, 1, yf, false);
            ConvOpt c;
            c.cout_total = FEATURE_DIM;
            conv("feature.top_conv", yf, A, th_, tw_, FPN_DIM, FPN_DIM, 0, "feature.top_conv", FEATURE_DIM, 3, 1, FTC_ACT_NONE, {5, 0, 0}, FTC_F32, c);
            heads_done = true;
            break;
        }
        const R ynew = buf((int64_t)nh * M * FPN_DIM, A);
        level_conv("heads.upsamplers." + std::to_string(i), 0, nh, ynew, false);
        y = ynew;
        yh = th_; yw = tw_;
    }
    if (!heads_done) {
        const int64_t gs = (int64_t)B * yh * yw * FPN_DIM * esize(A);          // bytes between the heads' last-level tensors
        for (int hi = 0; hi < nh; ++hi) {
            const R yi = sub(y, hi * gs);
            const std::string name = HEADS[hi].name;
            ConvOpt c;
            if (hi >= 2 && hi < 8) {
                if (hi != 2) continue;                                       // covered by the grouped launch
                c.cout_total = 10; c.cout_off = HEADS[hi].ch0 + 1; c.groups = 6; c.extra_flags = FTC_FLAG_GROUP_OUT_SLICE;
                conv("heads.top6", yi, A, yh, yw, FPN_DIM, FPN_DIM, 0, "heads.top6", 1, 3, 1, FTC_ACT_NONE, {4, 0, 0}, FTC_F32, c);
            } else if (HEADS[hi].ch0 >= 0) {                                 // map heads write straight into their channel slice; channel 1 is the NMS slot
                c.cout_total = 10; c.cout_off = HEADS[hi].ch0 == 0 ? 0 : HEADS[hi].ch0 + 1;
                conv(name + ".top_conv", yi, A, yh, yw, FPN_DIM, FPN_DIM, 0, name + ".top_conv", HEADS[hi].out_dim, 3, 1, FTC_ACT_NONE, {4, 0, 0}, FTC_F32, c);
            } else {
                c.cout_total = FEATURE_DIM;
                conv(name + ".top_conv", yi, A, yh, yw, FPN_DIM, FPN_DIM, 0, name + ".top_conv", HEADS[hi].out_dim, 3, 1, FTC_ACT_NONE, {5, 0, 0}, FTC_F32, c);
            }
        }
    }
    {
        SymOp s;
        ftc_op& o = s.o;
        o.kind = FTC_OP_NMS; o.B = B; o.H = mh; o.W = mw; o.Ho = mh; o.Wo = mw; o.Cout_total = 10;
        s.out = {4, 0, 0};
        emit({"nms", "nms", 0.0, (double)B * mh * mw * 8.0}, s);
    }
    if (!err_.empty()) return ftc_set_error(FTC_ERR_INVALID, "ftc model plan: " + err_);
    return finish(out, mh, mw);
}

// SimpleDecoder.forward (models/detector.py:249-254) on `H` gathered feature rows: ops 3i .. 3i+2 = head i; the head's output is
// addressed through FTC_BASE_HEATMAP so that ftc_decoder_forward runs each op range with that base pointing at its own buffer.
int Builder::build_decoder(ModelPlan* out) {
    const int A = act_;
    for (int i = 0; i < 3; ++i) {
        const std::string q = "decoder." + std::to_string(i), name = "decoder.blocks." + std::to_string(i);
        const R a = buf((int64_t)H * DECODER_MID, A), b = buf((int64_t)H * DECODER_MID, A);
        conv(name + ".0", {3, 0, 0}, A, H, 1, DECODER_KPAD, DECODER_KPAD, 0, q + ".l0", DECODER_MID, 1, 1, FTC_ACT_GELU, a, A);
        conv(name + ".3", a, A, H, 1, DECODER_MID, DECODER_MID, 0, q + ".l1", DECODER_MID, 1, 1, FTC_ACT_GELU, b, A);
        conv(name + ".6", b, A, H, 1, DECODER_MID, DECODER_MID, 0, q + ".l2", DECODER_MODULO[i], 1, 1, FTC_ACT_NONE, {4, 0, 0}, FTC_F32);
    }
    if (!err_.empty()) return ftc_set_error(FTC_ERR_INVALID, "ftc decoder plan: " + err_);
    return finish(out, H, 1);
}

// liveness-based first-fit arena + resolution of the symbolic operands
int Builder::finish(ModelPlan* out, int mh, int mw) {
    std::vector<int> order(bufs_.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bufs_[a].first < bufs_[b].first; });
    struct Iv { int64_t off, end; int last; };
    std::vector<Iv> live;
    int64_t top = 0, peak = 0;
    for (int bi : order) {
        Buf& b = bufs_[bi];
        if (b.last < 0) return ftc_set_error(FTC_ERR_INVALID, "ftc model plan: buffer never used");
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Iv& iv) { return iv.last < b.first; }), live.end());
        std::sort(live.begin(), live.end(), [](const Iv& a, const Iv& c) { return a.off < c.off || (a.off == c.off && a.end < c.end); });
        int64_t off = 0;
        for (const Iv& iv : live) {
            if (off + b.nbytes <= iv.off) break;
            off = std::max(off, iv.end);
        }
        b.offset = off;
        live.push_back({off, off + b.nbytes, b.last});
        top = std::max(top, off + b.nbytes);
        int64_t sum = 0;
        for (const Iv& iv : live) sum += iv.end - iv.off;
        peak = std::max(peak, sum);
    }
    auto res = [&](const R& r) -> ftc_ref {
        ftc_ref f{};
        switch (r.kind) {
        case 1: f.base = FTC_BASE_WORKSPACE; f.offset = bufs_[r.v].offset + r.extra; break;
        case 2: f.base = FTC_BASE_WEIGHTS; f.offset = r.v; break;
        case 3: f.base = FTC_BASE_INPUT; break;
        case 4: f.base = FTC_BASE_HEATMAP; break;
        case 5: f.base = FTC_BASE_FEATURES; break;
        default: break;
        }
        return f;
    };
    out->plan.ops.clear();
    for (const SymOp& s : ops_) {
        ftc_op o = s.o;
        o.in = res(s.in); o.in2 = res(s.in2); o.out = res(s.out); o.w = res(s.w); o.w2 = res(s.w2); o.bias = res(s.bias); o.bias2 = res(s.bias2);
        o.scale = res(s.scale); o.shift = res(s.shift); o.aux = res(s.aux); o.out2 = res(s.out2);
        out->plan.ops.push_back(o);
    }
    apply_tuning(out->plan.ops);                // measured kernel choice per conv shape (ftc_op.aux0)
    out->plan.workspace_bytes = align_up(top);
    out->plan.weights_bytes = (int64_t)m_->blob.bytes.size();
    out->meta = meta_;
    out->B = B; out->H = H; out->W = W; out->h = mh; out->w = mw;
    out->peak_live_bytes = peak;
    out->total_buffer_bytes = 0;
    for (const Buf& b : bufs_) out->total_buffer_bytes += b.nbytes;
    return FTC_OK;
}

int get_plan(ftc_model* m, int B, int H, int W, int nchw, ModelPlan** out) {
    if (!m) return ftc_set_error(FTC_ERR_INVALID, "ftc model: null handle");
    if (B <= 0 || H <= 0 || W <= 0 || (H % 32) || (W % 32))
        return ftc_set_error(FTC_ERR_INVALID, "ftc model: B must be positive and H, W positive multiples of 32 (the reference always uses 768)");
    std::lock_guard<std::mutex> lk(m->mu);
    const auto key = std::make_tuple(B, H, W, nchw ? 1 : 0);
    auto it = m->plans.find(key);
    if (it == m->plans.end()) {
        std::unique_ptr<ModelPlan> mp(new (std::nothrow) ModelPlan());
        if (!mp) return ftc_set_error(FTC_ERR_NOMEM, "ftc model: out of host memory");
        Builder b(m, B, H, W, nchw != 0);
        int rc = b.build(mp.get());
        if (rc != FTC_OK) return rc;
        // every op goes through the same validation as a caller-supplied op list
        ftc_plan* checked = nullptr;
        rc = ftc_plan_create(mp->plan.ops.data(), (int)mp->plan.ops.size(), mp->plan.workspace_bytes, mp->plan.weights_bytes, &checked);
        if (rc != FTC_OK) return rc;
        ftc_plan_destroy(checked);
        it = m->plans.emplace(key, std::move(mp)).first;
    }
    *out = it->second.get();
    return FTC_OK;
}

}  // namespace

extern "C" {

int ftc_create(const ftc_tensor* tensors, int n_tensors, const char* model_size, int precision, ftc_model** out) {
    if (!tensors || n_tensors <= 0 || !out) return ftc_set_error(FTC_ERR_INVALID, "ftc_create: null/empty arguments");
    const std::string size = model_size && *model_size ? model_size : "xl";
    if (stage_rows(size).empty()) return ftc_set_error(FTC_ERR_INVALID, "ftc_create: model_size must be one of xl, l, m, s");
    if (precision != FTC_F32 && precision != FTC_BF16 && precision != FTC_F16 && precision != FTC_PRECISION_F16X3)
        return ftc_set_error(FTC_ERR_INVALID, "ftc_create: precision must be FTC_F32, FTC_BF16, FTC_F16 or FTC_PRECISION_F16X3");
    const int split16 = precision == FTC_PRECISION_F16X3 ? 1 : 0;
    if (split16) precision = FTC_F32;
    Weights w;
    for (int i = 0; i < n_tensors; ++i) {
        const ftc_tensor& t = tensors[i];
        if (!t.name || !t.data) return ftc_set_error(FTC_ERR_INVALID, "ftc_create: tensor " + std::to_string(i) + " has a null name or data pointer");
        if (t.dtype != FTC_F32) continue;                               // e.g. num_batches_tracked (int64): not used by the forward pass
        if (t.ndim < 0 || t.ndim > 4) return ftc_set_error(FTC_ERR_INVALID, std::string("ftc_create: tensor '") + t.name + "' has more than 4 dimensions");
        std::string name = t.name;
        if (name.rfind("detector.", 0) == 0) name = name.substr(9);     // TextDetectorModel keys ("decoder.*" keep their prefix)
        TensorView v;
        v.data = static_cast<const float*>(t.data);
        v.shape.assign(t.shape, t.shape + t.ndim);
        w.t[name] = v;
    }
    ftc_model* m = new (std::nothrow) ftc_model();
    if (!m) return ftc_set_error(FTC_ERR_NOMEM, "ftc_create: out of host memory");
    m->size = size;
    m->precision = precision;
    m->split16 = split16;
    const int rc = pack_weights(m, w);
    if (rc != FTC_OK) { delete m; return rc; }
    *out = m;
    return FTC_OK;
}

void ftc_destroy(ftc_model* model) { delete model; }

int64_t ftc_weights_bytes(const ftc_model* model) { return model ? (int64_t)model->blob.bytes.size() : 0; }

const void* ftc_weights_host(const ftc_model* model) { return model ? model->blob.bytes.data() : nullptr; }

int64_t ftc_weights_offset(const ftc_model* model, const char* name) {
    if (!model || !name) return -1;
    auto it = model->blob.table.find(name);
    return it == model->blob.table.end() ? -1 : it->second;
}

int64_t ftc_workspace_bytes(ftc_model* model, int B, int H, int W) {
    ModelPlan* mp = nullptr;
    if (get_plan(model, B, H, W, 0, &mp) != FTC_OK) return -1;
    return mp->plan.workspace_bytes;
}

int ftc_forward(ftc_model* model, const void* weights_dev, const void* image, int B, int H, int W, int nchw, int with_nms, void* heatmap,
                void* features, void* workspace, void* stream) {
    if (!weights_dev || !image || !heatmap || !features || !workspace) return ftc_set_error(FTC_ERR_INVALID, "ftc_forward: null pointer argument");
    ModelPlan* mp = nullptr;
    int rc = get_plan(model, B, H, W, nchw, &mp);
    if (rc != FTC_OK) return rc;
    void* bases[FTC_NUM_BASES] = {nullptr, workspace, const_cast<void*>(weights_dev), const_cast<void*>(image), heatmap, features};
    const int n = (int)mp->plan.ops.size();
    return ftc_plan_run(&mp->plan, bases, stream, 0, with_nms ? n - 1 : n - 2);
}

static int get_decoder_plan(ftc_model* m, int n_rows, std::shared_ptr<ModelPlan>* out) {
    if (!m) return ftc_set_error(FTC_ERR_INVALID, "ftc decoder: null model");
    if (!m->has_decoder) return ftc_set_error(FTC_ERR_INVALID, "ftc decoder: the model was created from a checkpoint without decoder.* tensors");
    if (n_rows <= 0) return ftc_set_error(FTC_ERR_INVALID, "ftc decoder: n_rows must be positive");
    std::lock_guard<std::mutex> lk(m->mu);
    auto it = m->decoder_plans.find(n_rows);
    if (it == m->decoder_plans.end()) {
        std::shared_ptr<ModelPlan> mp(new (std::nothrow) ModelPlan());
        if (!mp) return ftc_set_error(FTC_ERR_NOMEM, "ftc decoder: out of host memory");
        if (m->decoder_plans.size() >= ftc_model::kMaxDecoderPlans) {
            auto lru = m->decoder_plan_use.begin();
            for (auto u = m->decoder_plan_use.begin(); u != m->decoder_plan_use.end(); ++u)
                if (u->second < lru->second) lru = u;
            m->decoder_plans.erase(lru->first);
            m->decoder_plan_use.erase(lru);
        }
        Builder b(m, 1, n_rows, 1, false);
        int rc = b.build_decoder(mp.get());
        if (rc != FTC_OK) return rc;
        ftc_plan* checked = nullptr;
        rc = ftc_plan_create(mp->plan.ops.data(), (int)mp->plan.ops.size(), mp->plan.workspace_bytes, mp->plan.weights_bytes, &checked);
        if (rc != FTC_OK) return rc;
        ftc_plan_destroy(checked);
        it = m->decoder_plans.emplace(n_rows, std::move(mp)).first;
    }
    m->decoder_plan_use[n_rows] = ++m->decoder_clock;
    *out = it->second;
    return FTC_OK;
}

int64_t ftc_decoder_workspace_bytes(ftc_model* model, int n_rows) {
    std::shared_ptr<ModelPlan> mp;
    if (get_decoder_plan(model, n_rows, &mp) != FTC_OK) return -1;
    return mp->plan.workspace_bytes;
}

int ftc_decoder_forward(ftc_model* model, const void* weights_dev, const void* rows, int n_rows, float* out0, float* out1, float* out2,
                        void* workspace, void* stream) {
    if (!weights_dev || !rows || !out0 || !out1 || !out2 || !workspace) return ftc_set_error(FTC_ERR_INVALID, "ftc_decoder_forward: null pointer argument");
    std::shared_ptr<ModelPlan> mp;
    int rc = get_decoder_plan(model, n_rows, &mp);
    if (rc != FTC_OK) return rc;
    float* outs[3] = {out0, out1, out2};
    for (int i = 0; i < 3; ++i) {
        void* bases[FTC_NUM_BASES] = {nullptr, workspace, const_cast<void*>(weights_dev), const_cast<void*>(rows), outs[i], nullptr};
        rc = ftc_plan_run(&mp->plan, bases, stream, 3 * i, 3 * i + 2);
        if (rc != FTC_OK) return rc;
    }
    return FTC_OK;
}

int ftc_model_plan(ftc_model* model, int B, int H, int W, int nchw, const ftc_plan** plan, ftc_plan_info* info) {
    ModelPlan* mp = nullptr;
    int rc = get_plan(model, B, H, W, nchw, &mp);
    if (rc != FTC_OK) return rc;
    if (plan) *plan = &mp->plan;
    if (info) {
        info->n_ops = (int)mp->plan.ops.size();
        info->map_h = mp->h; info->map_w = mp->w;
        info->reserved = 0;
        info->workspace_bytes = mp->plan.workspace_bytes;
        info->weights_bytes = mp->plan.weights_bytes;
        info->peak_live_bytes = mp->peak_live_bytes;
        info->total_buffer_bytes = mp->total_buffer_bytes;
    }
    return FTC_OK;
}

int ftc_model_op_info(ftc_model* model, int B, int H, int W, int nchw, int index, ftc_op_info* out) {
    ModelPlan* mp = nullptr;
    int rc = get_plan(model, B, H, W, nchw, &mp);
    if (rc != FTC_OK) return rc;
    if (!out || index < 0 || index >= (int)mp->meta.size()) return ftc_set_error(FTC_ERR_INVALID, "ftc_model_op_info: index out of range");
    const OpMeta& me = mp->meta[index];
    std::memset(out, 0, sizeof *out);
    std::strncpy(out->name, me.name.c_str(), sizeof out->name - 1);
    std::strncpy(out->kind, me.kind.c_str(), sizeof out->kind - 1);
    out->flops = me.flops;
    out->bytes = me.bytes;
    return FTC_OK;
}

int ftc_plan_op(const ftc_plan* plan, int index, ftc_op* out) {
    if (!plan || !out || index < 0 || index >= (int)plan->ops.size()) return ftc_set_error(FTC_ERR_INVALID, "ftc_plan_op: bad arguments");
    *out = plan->ops[index];
    return FTC_OK;
}

}  // extern "C"
