// C ABI (include/ftc.h): plan validation / execution and the decode entry point.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ftc_common.h"
#include "ftc_host.h"

hipError_t launch_decode(const float* heat, const float* feat, int B, int h, int w, int C, const ftc_tile* tiles,
                         float logit_cut, int scale, int max_boxes, float* boxes, int box_stride, float* feats, int feat_stride,
                         int32_t* index, int32_t* counts, void* scratch, hipStream_t s);

hipError_t launch_tile_gather(const unsigned char* page, int PH, int PW, const int* origins, int B, int th, int tw, float* out,
                              hipStream_t s);
hipError_t launch_adamw_sf(const ftc_mt_chunk* chunks, int n_chunks, float beta2, float one_minus_beta2, float bias_correction2, float eps,
                           float decay, float ckp1, float y_alpha, float lr, int write_grad, hipStream_t s);
hipError_t launch_box_hists(const float* loc, int N, const float* page, int PH, int PW, float cut_off, double* out, hipStream_t s);
hipError_t launch_greedy(const float* loc, const int* order, int N, const double* hist1, const double* th, float cut_off, double* kept,
                         int* keep_idx, int* hdr, double* rb, int* status, int* cnt, int* cursor, int* nbr, long edge_cap, unsigned int* fill_big,
                         long fill_big_words, int force_seq, const float* seps, const float* codes, int mh, int mw, int scale, float* out_loc,
                         int* out_idx, int* out_n, int variant, int seed_start, double seed_scale, float* out_cmax, hipStream_t s);
hipError_t launch_page_order(const float* loc, int N, const double* hist0, float cut_off, int* order, double* th, void* scratch, hipStream_t s);
hipError_t launch_paste_maps(const float* heat, const ftc_tile* tiles, int B, int h, int w, int scale, float* canv, int ph, int pw,
                             hipStream_t s);

hipError_t launch_topk_mask(const float* vals, long n, long k, unsigned char* mask, int32_t* sel_index, int32_t* count, hipStream_t s);
hipError_t launch_mask_compact(const unsigned char* mask, long n, int32_t* sel_index, long cap, int32_t* count, hipStream_t s);
hipError_t launch_gather_rows(const float* feat, const int32_t* sel_index, const int32_t* count, long cap, int C, int Cpad, void* rows, int out_dtype,
                              hipStream_t s);
hipError_t launch_losses(const float* heat, const long* hstrides, const float* label, const int32_t* idmap, int B, int h, int w, const float* const* dec,
                         const int* mod, const int32_t* sel_index, const int32_t* count, long cap, float* out, void* scratch, hipStream_t s);
hipError_t launch_cov_step(const float* L, int n, int iter, float* state, float* out_loss, hipStream_t s);

namespace {
// FTC_OP_LOSSES: ftc_losses on the NHWC [B,H,W,9] map block of the training plan
hipError_t launch_losses_op(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    const long hs[4] = {(long)o.H * o.W * 9, 1, (long)o.W * 9, 9};
    static const int mod[3] = {1091, 1093, 1097};                   // util_func.py:5 modulo_list
    const float* dec[3] = {(const float*)a.w2, a.bias, a.bias2};
    const bool has_dec = a.w2 && a.bias && a.bias2 && o.aux0 > 0;
    return launch_losses((const float*)a.in, hs, (const float*)a.in2, (const int32_t*)a.w, o.B, o.H, o.W, has_dec ? dec : nullptr, mod,
                         (const int32_t*)a.scale, nullptr, (long)o.aux0, (float*)a.out, a.aux, s);
}
}  // namespace

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return FTC_ERR_HIP;
}

// `extent` = bytes the op reads / writes from the operand's start (0 = unknown: only the start is checked).  The input / heat-map /
// feature bases are caller buffers whose sizes the plan does not know (ftc_forward derives them from B, H, W).
bool ref_ok(const ftc_ref& r, const ftc_plan* pl, bool required, int64_t extent, std::string* why) {
    if (r.base == FTC_BASE_NULL) {
        if (required) { *why = "missing required operand"; return false; }
        return true;
    }
    if (r.base < 0 || r.base >= FTC_NUM_BASES) { *why = "bad base id"; return false; }
    if (r.offset < 0 || (r.offset & 15)) { *why = "operand offset must be >= 0 and 16-byte aligned"; return false; }
    if (extent < 0) { *why = "operand size overflows"; return false; }
    const int64_t end = r.offset + (extent > 0 ? extent : 1);
    if (r.base == FTC_BASE_WORKSPACE && end > pl->workspace_bytes) { *why = "workspace operand out of range (offset + extent > workspace_bytes)"; return false; }
    if (r.base == FTC_BASE_WEIGHTS && end > pl->weights_bytes) { *why = "weights operand out of range (offset + extent > weights_bytes)"; return false; }
    return true;
}

const char* validate_op(const ftc_op& o, const ftc_plan* pl, std::string* why) {
    auto need = [&](const ftc_ref& r, bool req, const char* name, int64_t extent = 0) -> bool {
        std::string w;
        if (!ref_ok(r, pl, req, extent, &w)) { *why = std::string(name) + ": " + w; return false; }
        return true;
    };
    auto es = [](int dt) -> int64_t { return dt == FTC_F32 ? 4 : 2; };
    const int64_t G = o.groups > 1 ? o.groups : 1;
    const int64_t pin = (int64_t)o.B * o.H * o.W, pout = (int64_t)o.B * o.Ho * o.Wo;
    if (o.B <= 0 || o.H <= 0 || o.W <= 0) return "B/H/W must be positive";
    switch (o.kind) {
    case FTC_OP_STEM:
        if (!need(o.in, true, "in") || !need(o.out, true, "out", pout * o.Cout * es(o.out_dtype)) || !need(o.w, true, "w", (int64_t)27 * o.Cout * 4) ||
            !need(o.bias, true, "bias", (int64_t)o.Cout * 4)) return why->c_str();
        if (o.Cout % 4 || o.Cout > 256) return "stem: Cout must be a multiple of 4 (<= 256)";
        if ((long)o.B * o.Ho * o.Wo * (o.Cout / 4) >= 0x7fffffffL) return "stem: more than 2^31 output quads";
        if (!need(o.out2, false, "out2", pout * o.Cout * 2)) return why->c_str();
        if (o.out2.base != FTC_BASE_NULL && o.out_dtype != FTC_F32) return "stem: out2 (16-bit copy) needs an fp32 primary output";
        if (o.Ho != (o.H - 1) / 2 + 1 || o.Wo != (o.W - 1) / 2 + 1) return "stem: Ho/Wo inconsistent";
        return nullptr;
    case FTC_OP_CONV: {
        if (o.Cin <= 0 || o.Cout <= 0 || o.Cin_total <= 0 || o.Cout_total <= 0 || o.Ho <= 0 || o.Wo <= 0 || o.ksize <= 0) return "conv: sizes must be positive";
        const bool upin = (o.flags & FTC_FLAG_UPCAT_IN) != 0, topf = (o.flags & FTC_FLAG_TOP_FUSE) != 0;
        const int64_t kk = (int64_t)o.ksize * o.ksize;
        const int64_t in_ext = upin ? G * o.B * (o.H / 2) * (o.W / 2) * o.Cin_total * es(o.in_dtype) : G * pin * o.Cin_total * es(o.in_dtype);
        const int64_t in2_ext = upin ? ((o.flags & FTC_FLAG_GROUP_IN2_SHARED) ? 1 : G) * pin * (o.Cin - o.Cin_total) * es(o.in_dtype)
                                     : pout * o.Cout * es(o.res_dtype);
        const int64_t w_ext = G * ((o.flags & FTC_FLAG_W_PER_IMAGE) ? o.B : 1) * o.Cout * kk * o.Cin * es(o.w_dtype);
        const int64_t out_ext = topf ? G * pout * o.aux1 * 4 : ((o.flags & FTC_FLAG_GROUP_OUT_SLICE) ? 1 : G) * pout * o.Cout_total * es(o.out_dtype);
        if (!need(o.in, true, "in", in_ext) || !need(o.out, true, "out", out_ext) || !need(o.w, true, "w", w_ext) ||
            !need(o.bias, true, "bias", G * ((o.flags & FTC_FLAG_BORDER_BIAS) ? 16 : 1) * o.Cout * 4)) return why->c_str();
        if (!need(o.in2, (o.flags & (FTC_FLAG_RESIDUAL | FTC_FLAG_UPCAT_IN)) != 0, "in2", in2_ext) ||
            !need(o.scale, (o.flags & FTC_FLAG_SE_SCALE) != 0, "scale", (int64_t)o.B * o.Cin * 4)) return why->c_str();
        const bool x3conv = o.w_dtype == FTC_F32 && (o.flags & FTC_FLAG_SPLIT16);      // out2 = the pre-split copy (4 bytes per element)
        if (!need(o.out2, false, "out2", pout * o.Cout * (x3conv ? 4 : 2))) return why->c_str();
        if (!need(o.w2, topf, "w2", G * 32 * o.Cout * (o.w_dtype == FTC_F32 ? 4 : 2))) return why->c_str();
        if (o.out2.base != FTC_BASE_NULL && (o.out_dtype != FTC_F32 || o.Cout % 4)) return "conv: out2 (bf16 copy) needs an fp32 primary output and Cout % 4 == 0";
        if ((o.flags & FTC_FLAG_PRESPLIT) && (!x3conv || o.ksize != 1 || (o.flags & (FTC_FLAG_SE_SCALE | FTC_FLAG_UPCAT_IN)) || (o.Cin | o.Cin_total | o.cin_off) % 4))
            return "conv: FTC_FLAG_PRESPLIT (pre-split input) needs an fp16x3 1x1 convolution without SE scale, channel counts and offsets in whole chunks of 4";
        if (o.out2.base != FTC_BASE_NULL && o.w_dtype == FTC_F32 && (!x3conv || o.Cout != o.Cout_total || o.cout_off != 0 || G > 1))
            return "conv: an fp32 convolution writes out2 only as the pre-split copy of an fp16x3 plan (FTC_FLAG_SPLIT16; whole rows, one group)";
        // out2_index (conv_igemm_impl.h) lays the planes out from Cout alone: no channel slice, no groups, a 16-bit compute type
        if ((o.flags & FTC_FLAG_KBLOCK32) && (o.out2.base == FTC_BASE_NULL || !ftc_is16(o.w_dtype) || o.Cout % 32 || o.Cout != o.Cout_total || o.cout_off != 0 || G > 1))
            return "conv: KBLOCK32 describes out2 (the 16-bit copy) and needs a 16-bit w_dtype, Cout % 32 == 0, Cout == Cout_total, cout_off == 0, one group";
        return conv_validate(o);
    }
    case FTC_OP_DWCONV:
        if (o.in_dtype != FTC_F32 && !ftc_is16(o.in_dtype)) return "dwconv: unknown dtype";
        if (o.Cin <= 0 || o.aux0 <= 0 || o.Ho <= 0 || o.Wo <= 0) return "dwconv: sizes must be positive";
        if (!need(o.in, true, "in", pin * o.Cin * es(o.in_dtype)) || !need(o.out, true, "out", pout * o.Cin * es(o.in_dtype)) ||
            !need(o.w, true, "w", (int64_t)9 * o.Cin * 4) || !need(o.bias, true, "bias", (int64_t)o.Cin * 4) ||
            !need(o.aux, true, "aux", (int64_t)o.B * o.aux0 * o.Cin * 4)) return why->c_str();
        // the bf16 / fp16 stride-1 strip kernel builds a 32-bit buffer resource per image
        if (ftc_is16(o.in_dtype) && (int64_t)o.H * o.W * o.Cin * 2 >= 0x7fffffffLL) return "dwconv: one image exceeds the 2 GiB buffer-resource limit";
        if (o.Cin % (o.in_dtype == FTC_F32 ? 4 : 8) || o.Cin != o.Cout) return "dwconv: C must be a multiple of one 16-byte access (4 fp32 / 8 bf16) and Cin == Cout";
        if (o.stride != 1 && o.stride != 2) return "dwconv: stride must be 1 or 2";
        if (o.Ho != (o.H - 1) / o.stride + 1 || o.Wo != (o.W - 1) / o.stride + 1) return "dwconv: Ho/Wo inconsistent";
        if (o.in_dtype != o.out_dtype) return "dwconv: in/out dtype must match";
        return nullptr;
    case FTC_OP_MBHEAD: {
        if (o.Cin <= 0 || o.Cout <= 0) return "mbhead: sizes must be positive";
        if ((o.flags & FTC_FLAG_PRESPLIT) && o.in_dtype != FTC_F32) return "mbhead: FTC_FLAG_PRESPLIT applies to the fp32-tensor form";
        if (!ftc_mbhead_legal(o))
            return "mbhead: needs 16-bit in/out/w of one type (slice width Cout_total = 0 | 128 | 96 dividing Cout) or fp32 with FTC_FLAG_SPLIT16 (Cout % 64 == 0), stride 1, ksize 3, Ho = H, Wo = W, Cin % 32 == 0 and a map (or, with aux1 = "
                   "output rows per band, a band + 2 halo rows) of <= 576 pixels and < 601 row-separated slots";
        const int64_t esz = o.in_dtype == FTC_F32 ? 4 : 2;
        const int64_t nb = ftc_mbhead_bands(o), ns = o.Cout / ftc_mbhead_slice(o);
        if (!need(o.in, true, "in", pin * o.Cin * esz) || !need(o.out, true, "out", pin * o.Cout * esz) || !need(o.w2, true, "w2", (int64_t)o.Cout * o.Cin * esz) ||
            !need(o.bias2, true, "bias2", (int64_t)o.Cout * 4) || !need(o.w, true, "w", (int64_t)9 * o.Cout * 4) ||
            !need(o.bias, true, "bias", (int64_t)o.Cout * 4) || !need(o.aux, true, "aux", (int64_t)o.B * nb * o.Cout * 4)) return why->c_str();
        if ((o.scale.base != FTC_BASE_NULL) != (o.out2.base != FTC_BASE_NULL)) return "mbhead: scale (fc1 weight) and out2 (fc1 partial products) come together";
        if (o.scale.base != FTC_BASE_NULL) {
            // the kernel forms the partial products of at most 10 passes x 16 = 160 squeeze units (w1r[NU], mbconv_slice.hip); units beyond that were never written
            if (o.aux0 <= 0 || o.aux0 > FTC_MBHEAD_MAX_SQUEEZE) return "mbhead: aux0 (squeeze width S) must be in 1..160 when the fc1 partial products are requested";
            if (!need(o.scale, true, "scale", (int64_t)o.aux0 * o.Cout * 4) || !need(o.out2, true, "out2", (int64_t)o.B * nb * ns * o.aux0 * 4)) return why->c_str();
        }
        if (o.flags & 0x1000) { if (!need(o.in2, true, "in2", (int64_t)o.B * nb * ns * 256)) return why->c_str(); }      // phase timeline (tools/mbslice_bench.py)
        return nullptr;
    }
    case FTC_OP_FMBCONV: {
        if (!ftc_fmbconv_legal(o))
            return "fmbconv: needs 16-bit in / w of one type (or fp32 in / w with FTC_FLAG_SPLIT16 and aux1 = 256), fp32 out, 3x3 stride 1 (Ho = H, Wo = W), Cin % 32 == 0, aux1 (expanded channels) 256 or 384, Cout % 32 == 0 and "
                   "<= 128, whole tensors (no channel slices, no groups), act = SiLU, flags = RESIDUAL or none";
        const int64_t E = o.aux1, esz = o.w_dtype == FTC_F32 ? 4 : 2;      // (fp16x3 form: fp32 tensors, pre-split weights, out2 = the pre-split copy)
        if (!need(o.in, true, "in", pin * o.Cin * esz) || !need(o.out, true, "out", pin * o.Cout * 4) || !need(o.w2, true, "w2", E * 9 * o.Cin * esz) ||
            !need(o.bias2, true, "bias2", E * 4) || !need(o.w, true, "w", (int64_t)o.Cout * E * esz) || !need(o.bias, true, "bias", (int64_t)o.Cout * 4) ||
            !need(o.in2, (o.flags & FTC_FLAG_RESIDUAL) != 0, "in2", pin * o.Cout * 4) || !need(o.out2, false, "out2", pin * o.Cout * esz)) return why->c_str();
        return nullptr;
    }
    case FTC_OP_SE:
        if (o.aux0 <= 0 || o.aux1 <= 0 || o.Cin <= 0) return "se: C, S, P must be positive";
        if (!need(o.aux, true, "aux", (int64_t)o.B * o.aux1 * ((o.flags & FTC_FLAG_SE_HPART) ? o.aux0 : o.Cin) * 4) || !need(o.out, true, "out", (int64_t)o.B * o.Cin * 4) ||
            !need(o.in2, true, "in2", (int64_t)o.B * o.aux0 * 4) || !need(o.w, !(o.flags & FTC_FLAG_SE_HPART), "w", (int64_t)o.aux0 * o.Cin * 4) ||
            !need(o.w2, true, "w2", (int64_t)o.aux0 * o.Cin * 4) || !need(o.bias, true, "bias", (int64_t)o.aux0 * 4) ||
            !need(o.bias2, true, "bias2", (int64_t)o.Cin * 4)) return why->c_str();
        if (o.Cin % 4) return "se: C must be a multiple of 4";
        if ((size_t)o.Cin * 4 > 64000 || (size_t)o.aux0 * 4 > 64000) return "se: C or S too large for LDS";
        if (o.flags & FTC_FLAG_SE_FOLD) {
            if (o.Cout_total <= 0) return "se: SE_FOLD needs Cout_total (rows of the folded matrix)";
            const bool x3 = o.w_dtype == FTC_F32 && (o.flags & FTC_FLAG_SPLIT16);      // fp16x3 plan: pre-split fp32 chunks
            const int64_t eb = x3 ? 4 : 2;
            if (!need(o.in, true, "in", (int64_t)o.Cout_total * o.Cin * eb) || !need(o.out2, true, "out2", (int64_t)o.B * o.Cout_total * o.Cin * eb)) return why->c_str();
            if (o.Cin % 8 || o.Cout_total <= 0 || !(ftc_is16(o.w_dtype) || x3))
                return "se: SE_FOLD needs a 16-bit (or, with FTC_FLAG_SPLIT16, pre-split fp32) [Cout_total][C] matrix with C % 8 == 0";
        }
        return nullptr;
    case FTC_OP_UPCAT:
        if (o.aux0 < 0 || o.aux1 <= 0 || o.Ho <= 0 || o.Wo <= 0) return "upcat: sizes must be positive";
        {
            const int64_t cyt = o.Cin_total > 0 ? o.Cin_total : o.aux0;
            const int64_t in_ext = ((o.flags & FTC_FLAG_GROUP_IN_SLICE) ? 1 : G) * pin * cyt * es(o.in_dtype);
            if (!need(o.in, o.aux0 > 0, "in", in_ext) || !need(o.in2, true, "in2", pout * o.aux1 * es(o.res_dtype)) ||
                !need(o.out, true, "out", G * pout * (o.aux0 + o.aux1) * es(o.in_dtype)) || !need(o.scale, true, "scale", G * o.aux1 * 4) ||
                !need(o.shift, true, "shift", G * o.aux1 * 4)) return why->c_str();
        }
        // 16-byte lanes: 4 fp32 / 8 16-bit channels per access, for the channel counts AND the slice of a wider upsampled tensor
        if (o.aux0 % (o.in_dtype == FTC_F32 ? 4 : 8) || o.aux1 % (o.in_dtype == FTC_F32 ? 4 : 8) || o.aux1 <= 0) return "upcat: channel counts must be multiples of one 16-byte access (4 fp32 / 8 bf16)";
        if (o.aux0 > 0 && o.Cin_total > 0 && (o.Cin_total % (o.in_dtype == FTC_F32 ? 4 : 8) || o.cin_off % (o.in_dtype == FTC_F32 ? 4 : 8))) return "upcat: channel slice of the upsampled tensor is not 16-byte aligned";
        if (o.aux0 > 0 && o.Cin_total > 0 && (o.Cin_total % 4 || o.cin_off % 4 || o.cin_off + o.aux0 > o.Cin_total)) return "upcat: bad channel slice of the upsampled tensor";
        if (o.in_dtype != o.out_dtype) return "upcat: in/out dtype must match";
        if ((long)o.B * o.Ho * o.Wo * (o.aux0 + o.aux1) / 4 >= 0x7fffffffL) return "upcat: more than 2^31 output chunks per group";
        if (o.groups < 0 || o.groups > 64 || o.reserved0 != 0) return "upcat: groups must be in 0..64 and reserved0 zero";
        if ((o.flags & FTC_FLAG_GROUP_IN_SLICE) && (o.groups <= 1 || o.cin_off + o.groups * o.aux0 > o.Cin_total)) return "upcat: GROUP_IN_SLICE channel slices out of range";
        return nullptr;
    case FTC_OP_TAPSUM:
        if (o.aux0 <= 0 || o.aux1 <= 0) return "tapsum: bad aux0 / aux1";
        if (!need(o.in, true, "in", G * pin * o.aux0 * 4) || !need(o.out, true, "out") || !need(o.w, true, "w", (int64_t)o.aux1 * 16) ||
            !need(o.bias, true, "bias", (int64_t)o.aux1 * 4)) return why->c_str();
        if (o.aux0 < 4 || o.aux0 > 32 || o.aux0 % 4 || o.aux1 < 1 || o.aux1 > 64 || o.Cout_total < 1 || o.groups < 1) return "tapsum: bad aux0 / aux1 / Cout_total / groups";
        return nullptr;
    case FTC_OP_BNSTAT: {
        if (o.Cin <= 0) return "bnstat: Cin must be positive";
        if (o.in_dtype != FTC_F32 && !ftc_is16(o.in_dtype)) return "bnstat: unknown dtype";
        const int64_t M = pin;
        if (!need(o.in, true, "in", M * o.Cin * es(o.in_dtype)) || !need(o.out, true, "out", (int64_t)4 * o.Cin * 4) || !need(o.w, true, "w", (int64_t)o.Cin * 4) ||
            !need(o.bias, true, "bias", (int64_t)o.Cin * 4) || !need(o.aux, false, "aux", (int64_t)2 * o.Cin * 4) ||
            !need(o.in2, true, "in2", (int64_t)ftc_bnstat_chunks(M) * 2 * o.Cin * 8)) return why->c_str();
        return nullptr;
    }
    case FTC_OP_BNACT: {
        if (o.Cin <= 0 || o.aux0 < 0 || o.aux0 > 65535) return "bnact: bad Cin / aux0";
        if ((o.in_dtype != FTC_F32 && !ftc_is16(o.in_dtype)) || (o.out_dtype != FTC_F32 && !ftc_is16(o.out_dtype))) return "bnact: unknown dtype";
        const int tc = o.w_dtype == FTC_F16 ? FTC_F16 : FTC_BF16;
        if ((ftc_is16(o.in_dtype) && o.in_dtype != tc) || (ftc_is16(o.out_dtype) && o.out_dtype != tc)) return "bnact: 16-bit tensors must be in the plan's 16-bit type (w_dtype)";
        if ((o.flags & FTC_FLAG_RESIDUAL) && o.res_dtype != FTC_F32 && o.res_dtype != tc) return "bnact: residual must be fp32 or the plan's 16-bit type";
        if (o.act != FTC_ACT_NONE && o.act != FTC_ACT_SILU && o.act != FTC_ACT_GELU) return "bnact: unknown activation";
        if (!need(o.in, true, "in", pin * o.Cin * es(o.in_dtype)) || !need(o.out, true, "out", pin * o.Cin * es(o.out_dtype)) ||
            !need(o.scale, true, "scale", (int64_t)o.Cin * 4) || !need(o.shift, true, "shift", (int64_t)o.Cin * 4) ||
            !need(o.in2, (o.flags & FTC_FLAG_RESIDUAL) != 0, "in2", pin * o.Cin * es(o.res_dtype)) || !need(o.w2, false, "w2", (int64_t)o.B * 4) ||
            !need(o.out2, false, "out2", pin * o.Cin * 2) || !need(o.aux, false, "aux", (int64_t)o.B * (o.aux0 > 0 ? o.aux0 : 1) * o.Cin * 4)) return why->c_str();
        if (o.out2.base != FTC_BASE_NULL && o.out_dtype != FTC_F32) return "bnact: out2 (16-bit copy) needs an fp32 primary output";
        return nullptr;
    }
    case FTC_OP_NMS:
        if (!need(o.out, true, "out")) return why->c_str();
        if (o.Cout_total < 2) return "nms: heat-map needs >= 2 channels";
        return nullptr;
    case FTC_OP_GATHER_ROWS:
        if (o.aux0 <= 0 || o.Cin <= 0 || (o.Cin & 3) || o.Cout_total < o.Cin || (o.Cout_total & 7)) return "gather_rows: need aux0 > 0, Cin % 4 == 0, Cout_total >= Cin, Cout_total % 8 == 0";
        if (o.out_dtype != FTC_F32 && !ftc_is16(o.out_dtype)) return "gather_rows: unknown out_dtype";
        if (!need(o.in, true, "in", pin * o.Cin * 4) || !need(o.in2, true, "in2", (int64_t)o.aux0 * 4) || !need(o.out, true, "out", (int64_t)o.aux0 * o.Cout_total * es(o.out_dtype))) return why->c_str();
        return nullptr;
    case FTC_OP_LOSSES:
    case FTC_OP_LOSS_BWD: {
        const int64_t hw = (int64_t)o.H * o.W;
        if (!need(o.in, true, "in", pin * 9 * 4) || !need(o.in2, true, "in2", o.B * 5 * hw * 4) || !need(o.w, true, "w", o.B * 2 * hw * 4)) return why->c_str();
        if (o.aux0 < 0) return "losses: aux0 (selected rows) must be >= 0";
        if (o.aux0 > 0 && (!need(o.w2, true, "w2", (int64_t)o.aux0 * 1091 * 4) || !need(o.bias, true, "bias", (int64_t)o.aux0 * 1093 * 4) ||
                           !need(o.bias2, true, "bias2", (int64_t)o.aux0 * 1097 * 4) || !need(o.scale, true, "scale", (int64_t)o.aux0 * 4))) return why->c_str();
        if (o.kind == FTC_OP_LOSSES) {
            if (!need(o.out, true, "out", 64) || !need(o.aux, true, "aux", ftc_losses_scratch_bytes())) return why->c_str();
        } else {
            if (!need(o.out, true, "out", pin * 9 * 4) || !need(o.shift, true, "shift", 36) || !need(o.aux, true, "aux", 64)) return why->c_str();
            if (o.aux0 > 0 && (o.aux1 < 1097 || (o.aux1 & 3) || !need(o.out2, true, "out2", (int64_t)3 * o.aux0 * o.aux1 * 4))) return "loss_bwd: out2 / aux1 (padded logit row >= 1097, % 4 == 0)";
        }
        return nullptr;
    }
    case FTC_OP_SCATTER_ROWS:
        if (o.aux0 <= 0 || o.Cout_total <= 0 || (o.Cout_total & 3)) return "scatter_rows: need aux0 > 0 and Cout_total % 4 == 0";
        if (!need(o.in, true, "in", (int64_t)o.aux0 * o.Cout_total * 4) || !need(o.in2, true, "in2", (int64_t)o.aux0 * 4) || !need(o.out, true, "out", pin * o.Cout_total * 4)) return why->c_str();
        return nullptr;
    case FTC_OP_BNBWD: {
        if (o.Cin <= 0 || (o.Cin & 3)) return "bnbwd: Cin must be a positive multiple of 4";
        const int64_t gs = o.Cin_total > 0 ? o.Cin_total : o.Cin;
        if ((gs & 3) || (o.cin_off & 3) || o.cin_off + o.Cin > gs) return "bnbwd: bad channel slice of the incoming gradient";
        if (o.act != FTC_ACT_NONE && o.act != FTC_ACT_SILU && o.act != FTC_ACT_GELU) return "bnbwd: unknown activation";
        if (o.out.base == FTC_BASE_NULL && o.out2.base == FTC_BASE_NULL) return "bnbwd: needs out (fp32) and / or out2 (16-bit copy)";
        if (o.out2.base != FTC_BASE_NULL && !ftc_is16(o.w_dtype)) return "bnbwd: out2 is a 16-bit copy in the plan's compute type (w_dtype)";
        if (o.in_dtype != FTC_F32 && !(ftc_is16(o.in_dtype) && o.in_dtype == o.w_dtype)) return "bnbwd: in_dtype (the type z is stored in) is fp32 or the plan's 16-bit compute type (w_dtype)";
        if (!need(o.in, true, "in", pin * gs * 4) || !need(o.in2, true, "in2", pin * o.Cin * es(o.in_dtype)) || !need(o.scale, true, "scale", (int64_t)4 * o.Cin * 4) ||
            !need(o.out, false, "out", pin * o.Cin * 4) || !need(o.out2, false, "out2", pin * o.Cin * 2) || !need(o.w, false, "w", (int64_t)o.Cin * 4) || !need(o.shift, false, "shift", (int64_t)o.Cin * 4) ||
            !need(o.w2, false, "w2", (int64_t)o.B * 4) || !need(o.bias, false, "bias", (int64_t)o.B * o.Cin * 4) || !need(o.bias2, false, "bias2", (int64_t)o.B * o.Cin * 4) ||
            !need(o.aux, true, "aux", (int64_t)ftc_bnstat_chunks(pin) * 2 * o.Cin * 8 + (int64_t)2 * o.Cin * 4)) return why->c_str();
        return nullptr;
    }
    case FTC_OP_WGRAD: {
        if (o.Cin <= 0 || o.Cout <= 0 || o.Ho <= 0 || o.Wo <= 0 || (o.ksize != 1 && o.ksize != 3) || (o.stride != 1 && o.stride != 2)) return "wgrad: bad sizes";
        if (o.w_dtype != FTC_F32 && !ftc_is16(o.w_dtype)) return "wgrad: unknown compute type";
        if ((o.in_dtype != FTC_F32 && o.in_dtype != o.w_dtype) || (o.res_dtype != FTC_F32 && o.res_dtype != o.w_dtype))
            return "wgrad: operands are fp32 or 16-bit copies in the compute type (in_dtype: layer input, res_dtype: output gradient)";
        const int pad = (o.ksize - 1) / 2;
        if (o.Ho != (o.H + 2 * pad - o.ksize) / o.stride + 1 || o.Wo != (o.W + 2 * pad - o.ksize) / o.stride + 1) return "wgrad: Ho/Wo inconsistent";
        const int64_t cit = o.Cin_total > 0 ? o.Cin_total : o.Cin, cot = o.Cout_total > 0 ? o.Cout_total : o.Cout;
        if (o.cin_off + o.Cin > cit || o.cout_off + o.Cout > cot || o.aux0 < 1 || o.aux0 > 4096) return "wgrad: channel slices / splits out of range";
        const int64_t kk = (int64_t)o.ksize * o.ksize;
        if (!need(o.in, true, "in", pin * cit * es(o.in_dtype)) || !need(o.in2, true, "in2", pout * cot * es(o.res_dtype)) || !need(o.out, true, "out", kk * o.Cout * o.Cin * 4) ||
            !need(o.aux, true, "aux", (int64_t)o.aux0 * kk * o.Cout * o.Cin * 4) || !need(o.scale, (o.flags & FTC_FLAG_SE_SCALE) != 0, "scale", (int64_t)o.B * o.Cin * 4)) return why->c_str();
        return nullptr;
    }
    case FTC_OP_DWBWD:
        if (o.Cin <= 0 || (o.Cin & 3) || (o.stride != 1 && o.stride != 2)) return "dwbwd: Cin % 4 == 0, stride 1 | 2";
        if (o.Ho != (o.H - 1) / o.stride + 1 || o.Wo != (o.W - 1) / o.stride + 1) return "dwbwd: Ho/Wo inconsistent";
        // data gradient (out; needs w) and weight gradient (out2 + aux; needs in) are each optional, at least one is there
        if (o.out.base == FTC_BASE_NULL && o.out2.base == FTC_BASE_NULL) return "dwbwd: neither out (data gradient) nor out2 (weight gradient)";
        if (!need(o.in2, true, "in2", pout * o.Cin * 4)) return why->c_str();
        if (o.out.base != FTC_BASE_NULL && (!need(o.out, true, "out", pin * o.Cin * 4) || !need(o.w, true, "w", (int64_t)9 * o.Cin * 4))) return why->c_str();
        if (o.out2.base != FTC_BASE_NULL && (!need(o.in, true, "in", pin * o.Cin * 4) || !need(o.out2, true, "out2", (int64_t)9 * o.Cin * 4) ||
                                             !need(o.aux, true, "aux", (int64_t)ftc_chunks256(pout) * 9 * o.Cin * 8))) return why->c_str();
        return nullptr;
    case FTC_OP_SEBWD:
        if (o.Cin <= 0 || (o.Cin & 3) || o.aux0 <= 0 || o.aux1 <= 0) return "sebwd: C (% 4 == 0), S, P must be positive";
        if ((size_t)(2 * o.Cin + 2 * o.aux0) * 4 > 64000) return "sebwd: C / S too large for LDS";
        if (!need(o.in, true, "in", pin * o.Cin * 4) || !need(o.in2, true, "in2", pin * o.Cin * 4) || !need(o.scale, true, "scale", (int64_t)o.B * o.Cin * 4) ||
            !need(o.aux, true, "aux", (int64_t)o.B * o.aux1 * o.Cin * 4) || !need(o.w, true, "w", (int64_t)o.aux0 * o.Cin * 4) || !need(o.w2, true, "w2", (int64_t)o.aux0 * o.Cin * 4) ||
            !need(o.bias, true, "bias", (int64_t)o.aux0 * 4) || !need(o.out, true, "out", ((int64_t)(4 + 32) * o.B * o.Cin + (int64_t)2 * o.B * o.aux0) * 4) ||
            !need(o.out2, true, "out2", ((int64_t)2 * o.aux0 * o.Cin + o.aux0 + o.Cin) * 4)) return why->c_str();
        return nullptr;
    case FTC_OP_UPCATBWD:
        if (o.aux0 <= 0 || (o.aux0 & 3) || o.Cin_total < o.aux0 || (o.Cin_total & 3) || o.Ho <= 0 || o.Wo <= 0) return "upcatbwd: bad channel counts";
        if (!need(o.in, true, "in", pout * o.Cin_total * 4) || !need(o.out, true, "out", pin * o.aux0 * 4)) return why->c_str();
        return nullptr;
    case FTC_OP_DILATE:
        if (o.Cin <= 0 || (o.Cin & 3) || o.Ho != 2 * o.H || o.Wo != 2 * o.W) return "dilate: Cin % 4 == 0, Ho = 2H, Wo = 2W";
        if (!need(o.in, true, "in", pin * o.Cin * 4) || !need(o.out, true, "out", pout * o.Cin * 4)) return why->c_str();
        return nullptr;
    case FTC_OP_TOPDGRAD:
        if (o.Cin <= 0 || o.Cin > 8 || o.Cout <= 0 || (o.Cout & 3) || o.cin_off + o.Cin > o.Cin_total) return "topdgrad: 1..8 gradient channels, Cout % 4 == 0";
        if (!need(o.in, true, "in", pin * o.Cin_total * 4) || !need(o.w, true, "w", (int64_t)o.Cin * 9 * o.Cout * es(o.w_dtype)) || !need(o.out, true, "out", pin * o.Cout * 4)) return why->c_str();
        return nullptr;
    case FTC_OP_COLSUM: {
        const int64_t ct = o.Cin_total > 0 ? o.Cin_total : o.Cin;
        if (o.Cin <= 0 || o.cin_off + o.Cin > ct) return "colsum: bad column slice";
        if (!need(o.in, true, "in", pin * ct * 4) || !need(o.out, true, "out", (int64_t)o.Cin * 4) || !need(o.aux, true, "aux", (int64_t)ftc_chunks256(pin) * o.Cin * 8)) return why->c_str();
        return nullptr;
    }
    case FTC_OP_STEMWGRAD:
        if (o.Cout <= 0 || o.Cout > 32 || o.Ho != (o.H - 1) / 2 + 1 || o.Wo != (o.W - 1) / 2 + 1) return "stemwgrad: Cout <= 32, Ho/Wo of a stride-2 3x3";
        if (!need(o.in, true, "in") || !need(o.in2, true, "in2", pout * o.Cout * 4) || !need(o.out, true, "out", (int64_t)27 * o.Cout * 4) ||
            !need(o.aux, true, "aux", (int64_t)ftc_stemwgrad_chunks(pout) * 27 * o.Cout * 8)) return why->c_str();
        return nullptr;
    case FTC_OP_FILL:
        if (o.Cin <= 0) return "fill: Cin must be positive";
        if (!need(o.out, true, "out", pin * o.Cin * 4)) return why->c_str();
        return nullptr;
    case FTC_OP_JOIN:
        return nullptr;
    default:
        return "unknown op kind";
    }
}

inline void* resolve(const ftc_ref& r, void* const bases[FTC_NUM_BASES]) {
    if (r.base == FTC_BASE_NULL) return nullptr;
    return static_cast<char*>(bases[r.base]) + r.offset;
}

hipError_t run_one(const ftc_op& o, void* const bases[FTC_NUM_BASES], hipStream_t s) {
    OpArgs a;
    a.op = &o;
    a.in = resolve(o.in, bases);
    a.in2 = resolve(o.in2, bases);
    a.out = resolve(o.out, bases);
    a.w = resolve(o.w, bases);
    a.w2 = resolve(o.w2, bases);
    a.bias = static_cast<const float*>(resolve(o.bias, bases));
    a.bias2 = static_cast<const float*>(resolve(o.bias2, bases));
    a.scale = static_cast<const float*>(resolve(o.scale, bases));
    a.shift = static_cast<const float*>(resolve(o.shift, bases));
    a.aux = static_cast<float*>(resolve(o.aux, bases));
    a.out2 = resolve(o.out2, bases);
    switch (o.kind) {
    case FTC_OP_STEM: return launch_stem(a, s);
    case FTC_OP_CONV: return launch_conv(a, s);
    case FTC_OP_DWCONV: return launch_dwconv(a, s);
    case FTC_OP_SE: return launch_se(a, s);
    case FTC_OP_MBHEAD: return launch_mbhead(a, s);
    case FTC_OP_FMBCONV: return launch_fmbconv(a, s);
    case FTC_OP_UPCAT: return launch_upcat(a, s);
    case FTC_OP_NMS: return launch_nms(a, s);
    case FTC_OP_TAPSUM: return launch_tapsum(a, s);
    case FTC_OP_BNSTAT: return launch_bnstat(a, s);
    case FTC_OP_BNACT: return launch_bnact(a, s);
    case FTC_OP_GATHER_ROWS: return launch_gather_rows_op(a, s);
    case FTC_OP_LOSSES: return launch_losses_op(a, s);
    case FTC_OP_LOSS_BWD: return launch_loss_bwd(a, s);
    case FTC_OP_SCATTER_ROWS: return launch_scatter_rows(a, s);
    case FTC_OP_BNBWD: return launch_bnbwd(a, s);
    case FTC_OP_WGRAD: return launch_wgrad(a, s);
    case FTC_OP_DWBWD: return launch_dwbwd(a, s);
    case FTC_OP_SEBWD: return launch_sebwd(a, s);
    case FTC_OP_UPCATBWD: return launch_upcatbwd(a, s);
    case FTC_OP_DILATE: return launch_dilate(a, s);
    case FTC_OP_TOPDGRAD: return launch_topdgrad(a, s);
    case FTC_OP_COLSUM: return launch_colsum(a, s);
    case FTC_OP_STEMWGRAD: return launch_stemwgrad(a, s);
    case FTC_OP_FILL: return launch_fill(a, s);
    case FTC_OP_JOIN: return hipSuccess;
    default: return hipErrorInvalidValue;
    }
}

int check_bases(const ftc_plan* plan, void* const bases[FTC_NUM_BASES], int first, int last) {
    for (int i = first; i <= last; ++i) {
        const ftc_op& o = plan->ops[i];
        const ftc_ref* refs[] = {&o.in, &o.in2, &o.out, &o.w, &o.w2, &o.bias, &o.bias2, &o.scale, &o.shift, &o.aux, &o.out2};
        for (const ftc_ref* r : refs)
            if (r->base != FTC_BASE_NULL && bases[r->base] == nullptr)
                return fail(FTC_ERR_INVALID, "ftc_plan_run: op " + std::to_string(i) + " needs base " + std::to_string(r->base) + " which is NULL");
    }
    return FTC_OK;
}

}  // namespace

int ftc_set_error(int code, const std::string& msg) { return fail(code, msg); }

extern "C" {

int ftc_abi_version(void) { return FTC_ABI_VERSION; }

const char* ftc_last_error(void) { return g_err.c_str(); }

int ftc_device_info(int* n_cu, char* name, int name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(FTC_ERR_NO_DEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fail(FTC_ERR_NO_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (name && name_len > 0) { std::strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FTC_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    return FTC_OK;
}

int ftc_plan_create(const ftc_op* ops, int n_ops, int64_t workspace_bytes, int64_t weights_bytes, ftc_plan** out) {
    if (!ops || n_ops <= 0 || !out) return fail(FTC_ERR_INVALID, "ftc_plan_create: null/empty arguments");
    ftc_plan* pl = new (std::nothrow) ftc_plan();
    if (!pl) return fail(FTC_ERR_NOMEM, "ftc_plan_create: out of host memory");
    pl->workspace_bytes = workspace_bytes;
    pl->weights_bytes = weights_bytes;
    pl->ops.assign(ops, ops + n_ops);
    for (int i = 0; i < n_ops; ++i) {
        std::string why;
        const char* bad = validate_op(pl->ops[i], pl, &why);
        if (bad) {
            std::string msg = "ftc_plan_create: op " + std::to_string(i) + " (kind " + std::to_string(pl->ops[i].kind) + "): " + bad;
            delete pl;
            return fail(FTC_ERR_INVALID, msg);
        }
    }
    *out = pl;
    return FTC_OK;
}

void ftc_plan_destroy(ftc_plan* plan) { delete plan; }

int ftc_plan_num_ops(const ftc_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

int ftc_plan_run(const ftc_plan* plan, void* const bases[FTC_NUM_BASES], void* stream, int first_op, int last_op) {
    if (!plan || !bases) return fail(FTC_ERR_INVALID, "ftc_plan_run: null arguments");
    const int n = (int)plan->ops.size();
    if (last_op < 0 || last_op >= n) last_op = n - 1;
    if (first_op < 0) first_op = 0;
    if (first_op > last_op) return fail(FTC_ERR_INVALID, "ftc_plan_run: empty op range");
    int rc = check_bases(plan, bases, first_op, last_op);
    if (rc != FTC_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = first_op; i <= last_op; ++i) {
        hipError_t e = run_one(plan->ops[i], bases, s);
        if (e != hipSuccess) {
            char buf[64];
            std::snprintf(buf, sizeof buf, "ftc_plan_run: op %d (kind %d)", i, plan->ops[i].kind);
            return fail_hip(e, buf);
        }
    }
    return FTC_OK;
}

int ftc_plan_run_streams(const ftc_plan* plan, void* const bases[FTC_NUM_BASES], void* stream, void* side_stream, int first_op, int last_op) {
    if (!side_stream) return ftc_plan_run(plan, bases, stream, first_op, last_op);
    if (!plan || !bases) return fail(FTC_ERR_INVALID, "ftc_plan_run_streams: null arguments");
    const int n = (int)plan->ops.size();
    if (last_op < 0 || last_op >= n) last_op = n - 1;
    if (first_op < 0) first_op = 0;
    if (first_op > last_op) return fail(FTC_ERR_INVALID, "ftc_plan_run_streams: empty op range");
    int rc = check_bases(plan, bases, first_op, last_op);
    if (rc != FTC_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream), s2 = static_cast<hipStream_t>(side_stream);
    // the two events belong to this call (a plan may be run from several host threads); destroying an event with work pending is legal
    hipEvent_t fork = nullptr, join = nullptr;
    hipError_t e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&join, hipEventDisableTiming);
    bool pending = false;
    int i = first_op;
    auto do_join = [&]() {
        if (!pending || e != hipSuccess) return;
        e = hipEventRecord(join, s2);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, join, 0);
        pending = false;
    };
    bool main_dirty = true;                                     // work issued on the main stream since the last fork
    for (; i <= last_op && e == hipSuccess; ++i) {
        const ftc_op& o = plan->ops[i];
        if (o.kind == FTC_OP_JOIN) { do_join(); continue; }
        if (o.flags & FTC_FLAG_SIDE_STREAM) {
            if (main_dirty) {
                e = hipEventRecord(fork, s);
                if (e == hipSuccess) e = hipStreamWaitEvent(s2, fork, 0);
                main_dirty = false;
            }
            if (e == hipSuccess) e = run_one(o, bases, s2);
            pending = true;
        } else {
            e = run_one(o, bases, s);
            main_dirty = true;
        }
    }
    const int failed = i - 1;
    hipError_t e_op = e;
    e = hipSuccess;
    do_join();
    if (fork) (void)hipEventDestroy(fork);
    if (join) (void)hipEventDestroy(join);
    if (e_op != hipSuccess || e != hipSuccess) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "ftc_plan_run_streams: op %d", failed);
        return fail_hip(e_op != hipSuccess ? e_op : e, buf);
    }
    return FTC_OK;
}

int ftc_op_kernel_label(const ftc_op* op, char* buf, int len) {
    if (!op || !buf || len <= 0) return fail(FTC_ERR_INVALID, "ftc_op_kernel_label: null arguments");
    switch (op->kind) {
    case FTC_OP_STEM: std::snprintf(buf, len, "stem_kernel"); break;
    case FTC_OP_CONV: conv_kernel_label(*op, buf, len); break;
    case FTC_OP_DWCONV:
        if (op->stride == 1 && !(op->flags & 0x100) && (ftc_is16(op->in_dtype) ? op->act != FTC_ACT_NONE : op->Cin % 4 == 0))
            std::snprintf(buf, len, "dwconv_strip_kernel<%s,s1>", ftc_dtname(op->in_dtype));
        else std::snprintf(buf, len, "dwconv_kernel<%s,s%d>", ftc_dtname(op->in_dtype), op->stride);
        break;
    case FTC_OP_SE: std::snprintf(buf, len, (op->flags & FTC_FLAG_SE_HPART) ? "se_gate" : "se_fc1+se_fc2"); break;
    case FTC_OP_MBHEAD:     // two instantiations, as the profiler sees them: the whole 24x24 map (FAST) / the general kernel (bands of rows)
        std::snprintf(buf, len, "mbconv_slice<%s,%dch,%s>", op->in_dtype == FTC_F32 ? "f16x3" : ftc_dtname(op->in_dtype), ftc_mbhead_slice(*op),
                      op->H == 24 && op->W == 24 && op->aux1 == 0 && !(op->flags & 0x100) ? "24x24" : "bands");
        break;
    case FTC_OP_FMBCONV: ftc_fmbconv_label(*op, buf, len); break;
    case FTC_OP_UPCAT: std::snprintf(buf, len, "upcat_kernel<%s>", ftc_dtname(op->in_dtype)); break;
    case FTC_OP_NMS: std::snprintf(buf, len, "nms_kernel"); break;
    case FTC_OP_TAPSUM: std::snprintf(buf, len, "tapsum_kernel"); break;
    case FTC_OP_BNSTAT: std::snprintf(buf, len, "bnstat_partial+final<%s>", ftc_dtname(op->in_dtype)); break;
    case FTC_OP_BNACT: std::snprintf(buf, len, "bnact_kernel<%s,%s>", ftc_dtname(op->in_dtype), ftc_dtname(op->out_dtype)); break;
    case FTC_OP_GATHER_ROWS: std::snprintf(buf, len, "gather_rows_kernel"); break;
    case FTC_OP_LOSSES: std::snprintf(buf, len, "map_loss+id_loss+finish"); break;
    case FTC_OP_LOSS_BWD: std::snprintf(buf, len, "maploss_bwd+idloss_bwd"); break;
    case FTC_OP_SCATTER_ROWS: std::snprintf(buf, len, "scatter_rows_kernel"); break;
    case FTC_OP_BNBWD: std::snprintf(buf, len, "bnbwd_partial+final+apply"); break;
    case FTC_OP_WGRAD: std::snprintf(buf, len, "wgrad_kernel<%s,k%d,s%d>+reduce", ftc_dtname(op->w_dtype), op->ksize, op->stride); break;
    case FTC_OP_DWBWD: std::snprintf(buf, len, "dwbwd_data+weight<s%d>", op->stride); break;
    case FTC_OP_SEBWD: std::snprintf(buf, len, "sebwd_ds+mlp+w"); break;
    case FTC_OP_UPCATBWD: std::snprintf(buf, len, "upcatbwd_kernel"); break;
    case FTC_OP_DILATE: std::snprintf(buf, len, "dilate_kernel"); break;
    case FTC_OP_TOPDGRAD: std::snprintf(buf, len, "topdgrad_kernel<%s>", ftc_dtname(op->w_dtype)); break;
    case FTC_OP_COLSUM: std::snprintf(buf, len, "colsum_partial+final"); break;
    case FTC_OP_STEMWGRAD: std::snprintf(buf, len, "stemwgrad_partial+final"); break;
    case FTC_OP_FILL: std::snprintf(buf, len, "memset"); break;
    case FTC_OP_JOIN: std::snprintf(buf, len, "join"); break;
    default: return fail(FTC_ERR_INVALID, "ftc_op_kernel_label: unknown op kind");
    }
    return FTC_OK;
}

int ftc_plan_profile(const ftc_plan* plan, void* const bases[FTC_NUM_BASES], void* stream, float* ms_out) {
    if (!plan || !bases || !ms_out) return fail(FTC_ERR_INVALID, "ftc_plan_profile: null arguments");
    const int n = (int)plan->ops.size();
    int rc = check_bases(plan, bases, 0, n - 1);
    if (rc != FTC_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) return fail(FTC_ERR_HIP, "ftc_plan_profile: hipEventCreate failed");
    (void)hipEventRecord(ev[0], s);
    int ret = FTC_OK;
    for (int i = 0; i < n; ++i) {
        hipError_t e = run_one(plan->ops[i], bases, s);
        if (e != hipSuccess) { ret = fail_hip(e, "ftc_plan_profile: launch"); break; }
        (void)hipEventRecord(ev[i + 1], s);
    }
    hipError_t se = hipStreamSynchronize(s);
    if (ret == FTC_OK && se != hipSuccess) ret = fail_hip(se, "ftc_plan_profile: sync");
    if (ret == FTC_OK)
        for (int i = 0; i < n; ++i) (void)hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return ret;
}

int ftc_decode(const float* heatmap, const float* features, int B, int h, int w, int C, const ftc_tile* tiles_dev,
               float logit_cut, int scale, int max_boxes, float* boxes, int box_stride, float* feats, int feat_stride,
               int32_t* index, int32_t* counts, void* scratch_dev, void* stream) {
    if (!heatmap || !features || !tiles_dev || !boxes || !feats || !index || !counts || !scratch_dev)
        return fail(FTC_ERR_INVALID, "ftc_decode: null pointer argument");
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3) || max_boxes <= 0 || scale <= 0)
        return fail(FTC_ERR_INVALID, "ftc_decode: bad sizes (C must be a multiple of 4)");
    if ((long)h * w > 0x7fffffffL / 16) return fail(FTC_ERR_INVALID, "ftc_decode: map too large");
    if (box_stride < 9 || feat_stride < C || (feat_stride & 3) || (reinterpret_cast<uintptr_t>(feats) & 15) || (reinterpret_cast<uintptr_t>(features) & 15))
        return fail(FTC_ERR_INVALID, "ftc_decode: box_stride >= 9, feat_stride >= C and a multiple of 4 floats, feature rows 16-byte aligned");
    hipError_t e = launch_decode(heatmap, features, B, h, w, C, tiles_dev, logit_cut, scale, max_boxes, boxes, box_stride, feats,
                                 feat_stride, index, counts, scratch_dev, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_decode");
    return FTC_OK;
}

int ftc_tile_gather(const unsigned char* page_u8, int page_h, int page_w, const int32_t* origins_yx_dev, int B, int tile_h,
                    int tile_w, float* tiles_out, void* stream) {
    if (!page_u8 || !origins_yx_dev || !tiles_out) return fail(FTC_ERR_INVALID, "ftc_tile_gather: null pointer argument");
    if (page_h <= 0 || page_w <= 0 || B <= 0 || tile_h <= 0 || tile_w <= 0) return fail(FTC_ERR_INVALID, "ftc_tile_gather: bad sizes");
    hipError_t e = launch_tile_gather(page_u8, page_h, page_w, origins_yx_dev, B, tile_h, tile_w, tiles_out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_tile_gather");
    return FTC_OK;
}

int ftc_paste_maps(const float* heatmap, const ftc_tile* tiles_dev, int B, int h, int w, int scale, float* canvases, int page_mh,
                   int page_mw, void* stream) {
    if (!heatmap || !tiles_dev || !canvases) return fail(FTC_ERR_INVALID, "ftc_paste_maps: null pointer argument");
    if (B <= 0 || h <= 0 || w <= 0 || scale <= 0 || page_mh <= 0 || page_mw <= 0) return fail(FTC_ERR_INVALID, "ftc_paste_maps: bad sizes");
    hipError_t e = launch_paste_maps(heatmap, tiles_dev, B, h, w, scale, canvases, page_mh, page_mw, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_paste_maps");
    return FTC_OK;
}

namespace {
// scratch of ftc_page_merge: header | kept boxes [N][4] f64 + their source rows (sequential fallback, result list) | rank-ordered edge table
// [N][6] f64 | status, neighbour counts / offsets, fill cursors | neighbour lists | a coverage bit image as large as the page
struct PageScratch { int64_t hdr, kept, keep_idx, rb, status, cnt, cursor, fill, fill_words, nbr, nbr_cap, total; };
PageScratch page_scratch_layout(int64_t n, int64_t page_h, int64_t page_w) {
    PageScratch L{};
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    int64_t o = 0;
    L.hdr = o; o += 256;
    L.kept = o; o = up(o + n * 32);
    L.keep_idx = o; o = up(o + n * 4);
    L.rb = o; o = up(o + n * 48);
    L.status = o; o = up(o + n * 4);
    L.cnt = o; o = up(o + (n + 1) * 4);
    L.cursor = o; o = up(o + (n + 1) * 4);
    L.fill_words = page_h * page_w / 32 + 64;
    L.fill = o; o = up(o + L.fill_words * 4);
    L.nbr = o;                                                   // the neighbour lists take the rest of the block
    int64_t cap = n * 256;                                       // default: room for 256 earlier overlapping candidates per box on average
    if (cap < (1 << 20)) cap = 1 << 20;
    if (cap > (1ll << 28)) cap = 1ll << 28;
    L.nbr_cap = cap;
    o = up(o + L.nbr_cap * 4);
    L.total = o;
    return L;
}
}  // namespace

int64_t ftc_page_merge_scratch_bytes(int n_boxes, int page_h, int page_w) {
    if (n_boxes <= 0 || page_h <= 0 || page_w <= 0) return 0;
    return page_scratch_layout(n_boxes, page_h, page_w).total;
}

int64_t ftc_page_order_scratch_bytes(int n_boxes) { return n_boxes > 0 ? 256 + (int64_t)n_boxes * 16 : 0; }

int ftc_page_order(const float* locations, int n_boxes, const double* hist0, float cut_off, int32_t* order_out, double* threshold_out, void* scratch,
                   int64_t scratch_bytes, void* stream) {
    if (!locations || !hist0 || !order_out || !threshold_out || !scratch) return fail(FTC_ERR_INVALID, "ftc_page_order: null pointer argument");
    if (n_boxes <= 0 || n_boxes > (1 << 20)) return fail(FTC_ERR_INVALID, "ftc_page_order: bad sizes");
    if (scratch_bytes < ftc_page_order_scratch_bytes(n_boxes)) return fail(FTC_ERR_INVALID, "ftc_page_order: scratch smaller than ftc_page_order_scratch_bytes");
    hipError_t e = launch_page_order(locations, n_boxes, hist0, cut_off, order_out, threshold_out, scratch, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_page_order");
    return FTC_OK;
}

int ftc_box_hists(const float* locations, int n_boxes, const float* page, int page_h, int page_w, float cut_off, double* hist_out,
                  void* stream) {
    if (!locations || !page || !hist_out) return fail(FTC_ERR_INVALID, "ftc_box_hists: null pointer argument");
    if (n_boxes <= 0 || page_h <= 0 || page_w <= 0) return fail(FTC_ERR_INVALID, "ftc_box_hists: bad sizes");
    hipError_t e = launch_box_hists(locations, n_boxes, page, page_h, page_w, cut_off, hist_out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_box_hists");
    return FTC_OK;
}

int ftc_page_merge_variant(const float* locations, const int32_t* order, int n_boxes, const double* hist1, const double* threshold_dev,
                           float cut_off, const float* seps, const float* codes, int mh, int mw, int scale, int page_h, int page_w, int variant,
                           int seed_start, double seed_scale, float* out_locations, int32_t* out_index, float* out_code_max, int32_t* out_count,
                           void* scratch, int64_t scratch_bytes, void* stream) {
    if (!locations || !order || !seps || !codes || !out_locations || !out_index || !out_count || !scratch)
        return fail(FTC_ERR_INVALID, "ftc_page_merge: null pointer argument");
    if (variant != FTC_PAGE_MERGE_PRODUCTION && variant != FTC_PAGE_MERGE_DEMO) return fail(FTC_ERR_INVALID, "ftc_page_merge: unknown variant");
    if (variant == FTC_PAGE_MERGE_PRODUCTION && (!hist1 || !threshold_dev)) return fail(FTC_ERR_INVALID, "ftc_page_merge: the production variant needs hist1 and threshold_dev");
    if (variant == FTC_PAGE_MERGE_PRODUCTION && seed_start >= 0 && seed_start < n_boxes) return fail(FTC_ERR_INVALID, "ftc_page_merge: seed rows exist only in the demo variant");
    if (n_boxes <= 0 || mh <= 0 || mw <= 0 || scale <= 0 || page_h <= 0 || page_w <= 0) return fail(FTC_ERR_INVALID, "ftc_page_merge: bad sizes");
    if (n_boxes > (1 << 20)) return fail(FTC_ERR_INVALID, "ftc_page_merge: more than 2^20 boxes");
    // fixed part | coverage image of the page | neighbour lists: whatever the block has behind the image (ftc_page_merge_scratch_bytes leaves
    // room for 256 per box).  A page whose lists do not fit goes through the sequential kernel on the device: same result, slower.
    const PageScratch lay = page_scratch_layout(n_boxes, page_h, page_w);
    if (scratch_bytes < lay.nbr + 4096) return fail(FTC_ERR_INVALID, "ftc_page_merge: scratch smaller than the fixed part of ftc_page_merge_scratch_bytes");
    const int64_t nbr_cap = (scratch_bytes - lay.nbr) / 4;
    char* sp = static_cast<char*>(scratch);
    const char* fs_env = std::getenv("FTC_PAGE_MERGE_SEQ");                  // A/B and tests: "1" = the sequential (round-3) kernel
    const bool force_seq = fs_env && fs_env[0] == '1';
    hipError_t e = launch_greedy(locations, order, n_boxes, hist1, threshold_dev, cut_off, reinterpret_cast<double*>(sp + lay.kept),
                                 reinterpret_cast<int*>(sp + lay.keep_idx), reinterpret_cast<int*>(sp + lay.hdr), reinterpret_cast<double*>(sp + lay.rb),
                                 reinterpret_cast<int*>(sp + lay.status), reinterpret_cast<int*>(sp + lay.cnt), reinterpret_cast<int*>(sp + lay.cursor),
                                 reinterpret_cast<int*>(sp + lay.nbr), (long)nbr_cap, reinterpret_cast<unsigned int*>(sp + lay.fill), (long)lay.fill_words,
                                 force_seq ? 1 : 0, seps, codes, mh, mw, scale, out_locations, out_index, out_count, variant, seed_start, seed_scale,
                                 out_code_max, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_page_merge");
    return FTC_OK;
}

int ftc_page_merge(const float* locations, const int32_t* order, int n_boxes, const double* hist1, const double* threshold_dev,
                   float cut_off, const float* seps, const float* codes, int mh, int mw, int scale, int page_h, int page_w, float* out_locations,
                   int32_t* out_index, int32_t* out_count, void* scratch, int64_t scratch_bytes, void* stream) {
    return ftc_page_merge_variant(locations, order, n_boxes, hist1, threshold_dev, cut_off, seps, codes, mh, mw, scale, page_h, page_w,
                                  FTC_PAGE_MERGE_PRODUCTION, -1, 1.0, out_locations, out_index, nullptr, out_count, scratch, scratch_bytes, stream);
}

int ftc_adamw_schedulefree_step(const ftc_mt_chunk* chunks_dev, int n_chunks, float beta2, float one_minus_beta2, float bias_correction2,
                                float eps, float weight_decay, float ckp1, float y_alpha, float lr, int write_grad, void* stream) {
    if (!chunks_dev || n_chunks <= 0) return fail(FTC_ERR_INVALID, "ftc_adamw_schedulefree_step: empty chunk table");
    if (!(bias_correction2 > 0.0f)) return fail(FTC_ERR_INVALID, "ftc_adamw_schedulefree_step: bias_correction2 must be positive");
    hipError_t e = launch_adamw_sf(chunks_dev, n_chunks, beta2, one_minus_beta2, bias_correction2, eps, weight_decay, ckp1, y_alpha, lr,
                                   write_grad, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_adamw_schedulefree_step");
    return FTC_OK;
}

int ftc_topk_mask(const float* values, int64_t n, int64_t k, unsigned char* mask, int32_t* sel_index, int32_t* count, void* stream) {
    if (!values || !mask) return fail(FTC_ERR_INVALID, "ftc_topk_mask: null pointer argument");
    if (n <= 0 || n >= 0x7fffffffL || k < 0) return fail(FTC_ERR_INVALID, "ftc_topk_mask: need 0 < n < 2^31 and k >= 0");
    hipError_t e = launch_topk_mask(values, (long)n, (long)k, mask, sel_index, count, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_topk_mask");
    return FTC_OK;
}

int ftc_mask_compact(const unsigned char* mask, int64_t n, int32_t* sel_index, int64_t cap, int32_t* count, void* stream) {
    if (!mask || !sel_index || !count) return fail(FTC_ERR_INVALID, "ftc_mask_compact: null pointer argument");
    if (n <= 0 || n >= 0x7fffffffL || cap < 0) return fail(FTC_ERR_INVALID, "ftc_mask_compact: need 0 < n < 2^31 and cap >= 0");
    hipError_t e = launch_mask_compact(mask, (long)n, sel_index, (long)cap, count, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_mask_compact");
    return FTC_OK;
}

int ftc_gather_rows(const float* features, const int32_t* sel_index, const int32_t* count, int64_t cap, int C, int c_pad, void* rows,
                    int out_dtype, void* stream) {
    if (!features || !sel_index || !rows) return fail(FTC_ERR_INVALID, "ftc_gather_rows: null pointer argument");
    if (cap <= 0 || C <= 0 || (C & 3) || c_pad < C || (c_pad & 7) || (out_dtype != FTC_F32 && !ftc_is16(out_dtype)))
        return fail(FTC_ERR_INVALID, "ftc_gather_rows: need cap > 0, C % 4 == 0, c_pad >= C, c_pad % 8 == 0, out_dtype fp32 | bf16 | fp16");
    hipError_t e = launch_gather_rows(features, sel_index, count, (long)cap, C, c_pad, rows, out_dtype, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_gather_rows");
    return FTC_OK;
}

int ftc_pack_train_weights(const ftc_pack_entry* entries_dev, int n_entries, int64_t max_elems, void* stream) {
    if (!entries_dev || n_entries <= 0 || n_entries > 65535 || max_elems <= 0) return fail(FTC_ERR_INVALID, "ftc_pack_train_weights: need 1..65535 entries and max_elems > 0");
    hipError_t e = launch_pack_train(entries_dev, n_entries, (long)max_elems, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_pack_train_weights");
    return FTC_OK;
}

int ftc_wgrad_splits(int B, int Ho, int Wo, int Cout, int Cin, int ksize) {
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return 1;
    return ftc_wgrad_splits_impl(B, Ho, Wo, Cout, Cin, ksize);
}

int64_t ftc_losses_scratch_bytes(void) { return (int64_t)(512 * 9 + 512 * 4) * 8; }

int ftc_losses(const float* heatmap, const int64_t heat_strides[4], const float* labelmap, const int32_t* idmap, int B, int h, int w,
               const float* dec0, const float* dec1, const float* dec2, const int32_t* sel_index, const int32_t* count, int64_t cap,
               float* out, void* scratch, void* stream) {
    if (!heatmap || !heat_strides || !labelmap || !idmap || !out || !scratch) return fail(FTC_ERR_INVALID, "ftc_losses: null pointer argument");
    if (B <= 0 || h <= 0 || w <= 0) return fail(FTC_ERR_INVALID, "ftc_losses: bad sizes");
    const bool has_dec = dec0 || dec1 || dec2;
    if (has_dec && (!dec0 || !dec1 || !dec2 || !sel_index || !count || cap <= 0)) return fail(FTC_ERR_INVALID, "ftc_losses: decoder outputs need all three heads, sel_index, count and cap > 0");
    const long hs[4] = {(long)heat_strides[0], (long)heat_strides[1], (long)heat_strides[2], (long)heat_strides[3]};
    const float* dec[3] = {dec0, dec1, dec2};
    static const int mod[3] = {1091, 1093, 1097};                   // util_func.py:5 modulo_list
    hipError_t e = launch_losses(heatmap, hs, labelmap, idmap, B, h, w, has_dec ? dec : nullptr, mod, sel_index, count, (long)cap, out, scratch,
                                 static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_losses");
    return FTC_OK;
}

int ftc_cov_weighting_step(const float* losses, int n, int iteration, float* state, float* out_loss, void* stream) {
    if (!losses || !state || !out_loss) return fail(FTC_ERR_INVALID, "ftc_cov_weighting_step: null pointer argument");
    if (n <= 0 || n > 16 || iteration < 0) return fail(FTC_ERR_INVALID, "ftc_cov_weighting_step: need 0 < n <= 16 and iteration >= 0");
    hipError_t e = launch_cov_step(losses, n, iteration, state, out_loss, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "ftc_cov_weighting_step");
    return FTC_OK;
}

}  // extern "C"
