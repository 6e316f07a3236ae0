// Dispatcher of the implicit-GEMM convolution (kernel in conv_igemm_impl.h, one translation unit per
// type combination): validation, kernel label and the ftc_op -> ConvP lowering.
#include <cstring>

#include "conv_igemm_impl.h"

using namespace convimpl;

hipError_t launch_conv_f32(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_x3(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_bf16_bb(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_bf16_fb(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv1x1_px144(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_bf16_bf(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_bf16_ff(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_f16_hh(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_f16_fh(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_f16_hf(const ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_f16_ff(const ConvP& p, const ftc_op& o, hipStream_t s);
namespace convimpl {
bool conv3x3_c32_legal(const ftc_op& o);                               // conv3x3_c32.hip: the resident 32 -> 32 kernel (stage 1)
hipError_t launch_conv3x3_c32(const ConvP& p, const ftc_op& o, hipStream_t s);
}

void conv_kernel_label(const ftc_op& op, char* buf, int len) {
    const char* dt[] = {"f32", "bf16", "f16", "?"};
    if (ftc_thin_conv_legal(op)) {
        snprintf(buf, len, op.groups > 1 ? "thin_conv3x3<%s,co=%d,groups=%d>" : "thin_conv3x3<%s,co=%d>", (op.flags & FTC_FLAG_SPLIT16) ? "f16x3" : "f32", op.Cout, op.groups);
        return;
    }
    if (conv3x3_c32_legal(op)) {
        snprintf(buf, len, "conv3x3_c32<%s,tile=32x16x16,resident>", (op.flags & FTC_FLAG_SPLIT16) ? "f16x3" : dt[op.w_dtype & 3]);
        return;
    }
    if (uses_halo(op) && hint_wl1(op) && (op.flags & FTC_FLAG_W_FRAG)) {
        snprintf(buf, len, (op.flags & FTC_FLAG_TOP_FUSE) ? "conv3x3_wl1+top<%s,tile=192x16x16,bk=64>" : "conv3x3_wl1<%s,tile=192x16x16,bk=64>", dt[op.w_dtype & 3]);
        if (op.groups > 1) snprintf(buf + strlen(buf) - 1, len - strlen(buf) + 1, ",groups=%d>", op.groups);
        return;
    }
    if (uses_halo(op)) {
        snprintf(buf, len, (op.flags & FTC_FLAG_TOP_FUSE) ? "conv3x3_halo+top<%s,out=%s,tile=%dx16x16,bk=%d>" : "conv3x3_halo<%s,out=%s,tile=%dx16x16,bk=%d>", dt[op.w_dtype & 3],
                 dt[op.out_dtype & 3], halo_sn(op) * 64, halo_cpr(op) * (ftc_is16(op.w_dtype) ? 8 : 4));
        if (op.groups > 1) snprintf(buf + strlen(buf) - 1, len - strlen(buf) + 1, ",groups=%d>", op.groups);
        return;
    }
    if (cfg_px144(select_cfg(op))) {
        snprintf(buf, len, "conv1x1_px144<%s,tile=%s,bk=%d,nbuf=4>", op.w_dtype == FTC_F32 ? "f16x3" : dt[op.w_dtype & 3], kCfgName[select_cfg(op)], 64);
        return;
    }
    const bool dma = uses_glds(op);
    snprintf(buf, len, "conv_igemm%s<%s,in=%s,out=%s,tile=%s,bk=%d,nbuf=%d>", dma ? "_glds" : "", (op.flags & FTC_FLAG_SPLIT16) ? "f16x3" : dt[op.w_dtype & 3], dt[op.in_dtype & 3],
             dt[op.out_dtype & 3], kCfgName[select_cfg(op)], select_bk(op), dma ? glds_ring(op) : 1);
    if (!dma && hint_splitk(op) > 1) snprintf(buf + strlen(buf) - 1, len - strlen(buf) + 1, ",splitk=%d>", hint_splitk(op));
    if (op.groups > 1) snprintf(buf + strlen(buf) - 1, len - strlen(buf) + 1, ",groups=%d>", op.groups);
}

const char* conv_validate(const ftc_op& op) {
    if (op.ksize != 1 && op.ksize != 3) return "conv: ksize must be 1 or 3";
    if (op.stride != 1 && op.stride != 2) return "conv: stride must be 1 or 2";
    for (int dt : {op.w_dtype, op.in_dtype, op.out_dtype})
        if (dt != FTC_F32 && dt != FTC_BF16 && dt != FTC_F16) return "conv: unknown dtype";
    if (ftc_is16(op.w_dtype) && ((ftc_is16(op.in_dtype) && op.in_dtype != op.w_dtype) || (ftc_is16(op.out_dtype) && op.out_dtype != op.w_dtype)))
        return "conv: 16-bit input / output must be in the compute type (bf16 and fp16 do not mix)";
    if ((op.flags & FTC_FLAG_RESIDUAL) && ftc_is16(op.res_dtype) && op.res_dtype != (op.w_dtype == FTC_F16 ? FTC_F16 : FTC_BF16))
        return "conv: a 16-bit residual must be in the compute type";
    const int E = op.w_dtype == FTC_F32 ? 4 : 8;
    const int Ein = op.in_dtype == FTC_F32 ? 4 : 8;
    if (op.w_dtype == FTC_F32 && (op.in_dtype != FTC_F32)) return "conv: fp32 compute needs fp32 input";
    if (op.w_dtype == FTC_F32 && (op.out_dtype != FTC_F32)) return "conv: fp32 compute needs fp32 output";
    if (op.Cin % E) return "conv: Cin must be a multiple of the 16-byte chunk";
    if (op.Cin_total % Ein || op.cin_off % E) return "conv: input channel stride/offset not 16-byte aligned";
    if (!(op.flags & FTC_FLAG_UPCAT_IN) && op.cin_off + op.Cin > op.Cin_total) return "conv: input channel slice out of range";
    if (op.cout_off + op.Cout > op.Cout_total) return "conv: output channel slice out of range";
    const int pad = (op.ksize - 1) / 2;
    if (op.Ho != (op.H + 2 * pad - op.ksize) / op.stride + 1 || op.Wo != (op.W + 2 * pad - op.ksize) / op.stride + 1)
        return "conv: Ho/Wo inconsistent with H/W/ksize/stride";
    if ((op.flags & FTC_FLAG_SE_SCALE) && op.ksize != 1) return "conv: SE scale only on 1x1";
    if ((op.flags & FTC_FLAG_SPLIT16) && op.w_dtype != FTC_F32) return "conv: SPLIT16 (fp16x3) applies to fp32 operands";
    if ((op.flags & FTC_FLAG_BORDER_BIAS) && (op.ksize != 3 || op.stride != 1)) return "conv: border-bias table only for 3x3 stride 1";
    if ((long)op.B * op.Ho * op.Wo > 0x7fffffffL / 4) return "conv: too many output pixels";
    // buffer addressing is 32-bit: keep every operand below 2 GiB
    const long in_bytes = (long)op.B * op.H * op.W * op.Cin_total * (op.in_dtype == FTC_F32 ? 4 : 2);
    const long w_bytes = (long)op.Cout * op.ksize * op.ksize * op.Cin * (op.w_dtype == FTC_F32 ? 4 : 2);
    if (in_bytes >= 0x7ff00000L || w_bytes >= 0x7ff00000L) return "conv: operand larger than 2 GiB (split the batch)";
    if ((op.flags & FTC_FLAG_W_PER_IMAGE) && (op.flags & (FTC_FLAG_SE_SCALE | FTC_FLAG_BORDER_BIAS))) return "conv: per-image weight sets exclude SE_SCALE / BORDER_BIAS";
    if (!wset_legal(op)) return "conv: per-image weight sets need a pixel tile that divides Ho*Wo";
    if (op.flags & FTC_FLAG_UPCAT_IN) {
        const int bk = halo_cpr(op) * (ftc_is16(op.w_dtype) ? 8 : 4);
        if (!uses_halo(op) || halo_sn(op) != 3 || op.in_dtype != op.w_dtype || op.out_dtype != op.w_dtype)
            return "conv: UPCAT_IN needs the LDS-halo kernel with 192-channel tiles (aux0 = 65), input and output in the compute type";
        if ((op.H | op.W) & 1 || op.cin_off != 0 || op.Cin_total <= 0 || op.Cin_total >= op.Cin || op.Cin_total % bk || (op.Cin - op.Cin_total) % bk)
            return "conv: UPCAT_IN needs even H, W and both channel parts multiples of the K block";
        if (op.flags & (FTC_FLAG_RESIDUAL | FTC_FLAG_SE_SCALE | FTC_FLAG_W_PER_IMAGE)) return "conv: UPCAT_IN excludes RESIDUAL / SE_SCALE / W_PER_IMAGE";
    }
    if (op.flags & FTC_FLAG_TOP_FUSE) {
        if (!uses_halo(op) || halo_sn(op) != 3 || halo_cpr(op) != 8 || op.Cout != 192 || op.Cout_total != 192 || op.cout_off != 0 ||
            op.in_dtype != op.w_dtype || op.out_dtype != op.w_dtype)
            return "conv: TOP_FUSE needs the LDS-halo kernel with one 192-channel tile (aux0 = 65, Cin % 64 == 0 in 16 bits / % 32 in fp32, Cout = 192), tensors in the compute type";
        if (op.w_dtype == FTC_F32 && op.aux1 > 20) return "conv: TOP_FUSE in fp32 holds at most 20 outputs per pixel";
        if (op.flags & (FTC_FLAG_RESIDUAL | FTC_FLAG_GROUP_OUT_SLICE)) return "conv: TOP_FUSE excludes RESIDUAL / GROUP_OUT_SLICE";
        if (op.aux1 < 4 || op.aux1 > 32 || op.aux1 % 4) return "conv: TOP_FUSE output row width (aux1) must be a multiple of 4 in 4..32";
    }
    if (op.groups < 0 || op.groups > 64 || op.reserved0 != 0) return "conv: groups must be in 0..64 and reserved0 zero";
    if (op.groups > 1 && (op.flags & (FTC_FLAG_RESIDUAL | FTC_FLAG_SE_SCALE | FTC_FLAG_W_PER_IMAGE))) return "conv: grouped launches exclude RESIDUAL / SE_SCALE / W_PER_IMAGE";
    if ((op.flags & FTC_FLAG_GROUP_OUT_SLICE) && (op.groups <= 1 || op.cout_off + op.groups * op.Cout > op.Cout_total)) return "conv: GROUP_OUT_SLICE channel slices out of range";
    if (op.groups > 1 && (long)op.groups * op.B * op.Ho * op.Wo > 0x7fffffffL / 4) return "conv: too many output pixels over all groups";
    if (cfg_px144(hint_cfg(op)) && (!px144_legal(op, hint_cfg(op)) || hint_halo(op) || hint_splitk(op) > 1))
        return "conv: the x144 tiles are the 1x1 kernel for 16-bit operands or pre-split fp16x3 operands, Cin % 64 == 0, fp32 output, no activation, Cout % (64 | 80 | 128) == 0, Ho*Wo % 144 == 0";
    if (hint_halo(op) && !halo_legal(op)) return "conv: LDS-halo kernel is not legal for this op/tile";
    if (hint_splitk(op) > 1 && !splitk_legal(op, hint_splitk(op))) return "conv: split-K variant is not legal for this op/tile";
    if (op.aux0 < 0 || op.aux0 > 0xfff || hint_cfg(op) >= CFG_COUNT) return "conv: aux0 (tuned kernel choice) out of range";
    if ((op.aux0 & 128) || (op.flags & FTC_FLAG_W_FRAG)) {
        if (!hint_wl1(op) || !(op.flags & FTC_FLAG_W_FRAG)) return "conv: aux0 bit 7 (weights-through-L1 kernel) and FTC_FLAG_W_FRAG go together (with bit 6)";
        if (op.ksize != 3 || op.stride != 1 || !ftc_is16(op.w_dtype) || op.in_dtype != op.w_dtype || op.out_dtype != op.w_dtype || op.Cout != 192 ||
            op.Cout_total != 192 || op.cout_off != 0 || op.Cin % 64 || halo_sn(op) != 3 || halo_cpr(op) != 8 || (op.flags & (FTC_FLAG_RESIDUAL | FTC_FLAG_SE_SCALE | FTC_FLAG_W_PER_IMAGE)))
            return "conv: the weights-through-L1 kernel needs 3x3 stride 1, 16-bit operands, one 192-channel tile, Cin % 64 == 0 (aux0 = 193)";
        if (!(op.flags & FTC_FLAG_UPCAT_IN) && (op.Cin_total != op.Cin || op.cin_off != 0)) return "conv: the weights-through-L1 kernel reads whole input tensors";
    }
    if (hint_bk(op) && ftc_is16(op.w_dtype) && (op.Cin % hint_bk(op)) && hint_bk(op) != 32) return "conv: tuned K step does not divide Cin";
    if (hint_bk(op) == 128 && !ftc_is16(op.in_dtype)) return "conv: K step 128 needs 16-bit activations";
    if (hint_stage(op) >= 2 && !glds_legal(op)) return "conv: direct-to-LDS kernel is not legal for this op/tile";
    return nullptr;
}

hipError_t launch_conv(const OpArgs& a, hipStream_t s) {
    const ftc_op& o = *a.op;
    if (ftc_thin_conv_legal(o)) return launch_thin_conv(a, s);
    ConvP p;
    p.in = a.in; p.w = a.w; p.bias = a.bias; p.res = a.in2; p.out = a.out; p.out2 = a.out2; p.se = a.scale;
    p.in_bytes = (unsigned)((long)o.B * o.H * o.W * o.Cin_total * (o.in_dtype == FTC_F32 ? 4 : 2));
    p.w_bytes = (unsigned)((long)o.Cout * o.ksize * o.ksize * o.Cin * (o.w_dtype == FTC_F32 ? 4 : 2));
    p.se_bytes = (unsigned)((long)o.B * o.Cin * 4);
    p.B = o.B; p.H = o.H; p.W = o.W; p.Ho = o.Ho; p.Wo = o.Wo;
    p.Cin = o.Cin; p.CinT = o.Cin_total; p.cin_off = o.cin_off;
    p.Cout = o.Cout; p.CoutT = o.Cout_total; p.cout_off = o.cout_off;
    p.KS = o.ksize; p.stride = o.stride; p.pad = (o.ksize - 1) / 2;
    p.act = o.act; p.flags = o.flags; p.res_dtype = o.res_dtype;
    p.M = o.B * o.Ho * o.Wo;
    p.ncb = p.nk = p.nN = p.nblk = 0;
    p.use_glds = uses_glds(o) ? 1 : 0;
    p.glds_nbuf = glds_ring(o);
    p.split_k = (!uses_halo(o) && !uses_glds(o)) ? hint_splitk(o) : 1;
    p.wset_bytes = (o.flags & FTC_FLAG_W_PER_IMAGE) ? (int)p.w_bytes : 0;
    p.groups = o.groups > 1 ? o.groups : 1;
    p.nblk_g = 0;
    const long osz = o.out_dtype == FTC_F32 ? 4 : 2;
    const bool oslice = (o.flags & FTC_FLAG_GROUP_OUT_SLICE) != 0;
    p.in_gs = (long)p.in_bytes;
    p.w_gs = (long)p.w_bytes;
    p.bias_gs = ((o.flags & FTC_FLAG_BORDER_BIAS) ? 16 : 1) * o.Cout;
    p.out_gs = oslice ? 0 : (long)o.B * o.Ho * o.Wo * o.Cout_total * osz;
    p.out2_gs = oslice ? 0 : (long)o.B * o.Ho * o.Wo * o.Cout_total * (o.w_dtype == FTC_F32 ? 4 : 2);
    p.cout_gs = oslice ? o.Cout : 0;
    p.w2 = nullptr; p.w2_gs = 0; p.Tw = 0;
    p.in2u = nullptr; p.in2u_bytes = 0; p.in2u_gs = 0; p.Cy = 0; p.Hi = p.Wi = 0; p.ry = p.rx = 0.f;
    p.epi_off = 0;
    if (o.flags & FTC_FLAG_UPCAT_IN) {
        p.Cy = o.Cin_total; p.Hi = o.H / 2; p.Wi = o.W / 2;
        p.ry = o.H > 1 ? (float)(p.Hi - 1) / (float)(o.H - 1) : 0.f;
        p.rx = o.W > 1 ? (float)(p.Wi - 1) / (float)(o.W - 1) : 0.f;
        const long esz = ftc_is16(o.w_dtype) ? 2 : 4;
        p.in_bytes = (unsigned)((long)o.B * p.Hi * p.Wi * p.Cy * esz);
        p.in_gs = (long)p.in_bytes;
        p.in2u = a.in2;
        p.in2u_bytes = (unsigned)((long)o.B * o.H * o.W * (o.Cin - p.Cy) * esz);
        p.in2u_gs = (o.flags & FTC_FLAG_GROUP_IN2_SHARED) ? 0 : (long)p.in2u_bytes;
        p.res = nullptr;
    }
    if (o.flags & FTC_FLAG_TOP_FUSE) {
        p.w2 = a.w2; p.w2_gs = (long)32 * o.Cout * (o.w_dtype == FTC_F32 ? 4 : 2); p.Tw = o.aux1;      // (fp32 plans: an fp32 tap matrix)
        p.out_gs = (long)o.B * o.Ho * o.Wo * o.aux1 * 4;       // `out` holds T [G][B,Ho,Wo][aux1] fp32
        p.out2 = nullptr;
    }
    if (conv3x3_c32_legal(o)) return launch_conv3x3_c32(p, o, s);
    if (cfg_px144(select_cfg(o))) {
        if (o.flags & 0x1000) p.w2 = a.w2;                                  // phase timeline (tools/px144_bench.py)
        return launch_conv1x1_px144(p, o, s);
    }
    if (o.w_dtype == FTC_F32) return (o.flags & FTC_FLAG_SPLIT16) ? launch_conv_x3(p, o, s) : launch_conv_f32(p, o, s);
    if (o.w_dtype == FTC_F16) {
        if (o.in_dtype == FTC_F16 && o.out_dtype == FTC_F16) return launch_conv_f16_hh(p, o, s);
        if (o.in_dtype == FTC_F32 && o.out_dtype == FTC_F16) return launch_conv_f16_fh(p, o, s);
        if (o.in_dtype == FTC_F16 && o.out_dtype == FTC_F32) return launch_conv_f16_hf(p, o, s);
        return launch_conv_f16_ff(p, o, s);
    }
    if (o.in_dtype == FTC_BF16 && o.out_dtype == FTC_BF16) return launch_conv_bf16_bb(p, o, s);
    if (o.in_dtype == FTC_F32 && o.out_dtype == FTC_BF16) return launch_conv_bf16_fb(p, o, s);
    if (o.in_dtype == FTC_BF16 && o.out_dtype == FTC_F32) return launch_conv_bf16_bf(p, o, s);
    return launch_conv_bf16_ff(p, o, s);
}
