/*
 * ftc.h -- C ABI of the MI355X (gfx950) detector hot path of findtextCenterNet.
 *
 * The reference has NO foreign-function interface for this path: its detector is a Python
 * nn.Module (`CenterNetDetector`, /root/reference/models/detector.py:283-296) called through the
 * plug-in point `OCR_Processer.call_detector` (/root/reference/process_ocr_base.py:49-51, torch
 * implementation /root/reference/process_ocr_torch.py:43-49), and its peak decode is inline host
 * NumPy (/root/reference/process_ocr_base.py:496-538).  This header is therefore the boundary a
 * maintainer would bind from that Python (ctypes stub in INTEGRATION.md): plain pointers, sizes
 * and a hipStream_t -- no torch types.
 *
 * Conventions
 *   - every entry point returns 0 on success or a negative ftc_status; nothing throws across the
 *     ABI; ftc_last_error() gives a thread-local message for the last failure;
 *   - all device buffers are owned by the caller (PyTorch allocates them); the library allocates
 *     no device memory and never synchronises -- work is enqueued on the stream passed in;
 *   - a plan is immutable after ftc_plan_create, so one plan may be run from several host
 *     threads as long as each uses its own workspace and stream;
 *   - activations are NHWC; `heatmap` is [B,h,w,10] fp32 and `features` [B,h,w,100] fp32 in
 *     memory (the Python side returns them as NCHW *views*, which is what PyTorch itself
 *     produces for the channels_last-strided input the reference's callers pass,
 *     process_ocr_torch.py:44).
 */
#ifndef FTC_H_
#define FTC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: FTC_OP_BNSTAT / FTC_OP_BNACT (training-mode BatchNorm); FTC_OP_STEM and FTC_OP_DWCONV honour act = FTC_ACT_NONE (they applied SiLU
   unconditionally before); FTC_FLAG_W_FRAG
   5: the train step (BASELINE configs[4]): FTC_BASE_GRADS, FTC_OP_GATHER_ROWS .. FTC_OP_FILL (backward kernels), FTC_OP_BNSTAT writes
      [4][Cin] (scale, shift, mean, 1/std), ftc_losses out[12..13] = the two weight normalisers, ftc_pack_train_weights
   6: ftc_plan_run_streams / FTC_FLAG_SIDE_STREAM / FTC_OP_JOIN (ops with no consumer on the main chain -- the weight gradients -- on a
      second stream)
   7: FTC_OP_MBHEAD (expand 1x1 + depthwise 3x3 + SE squeeze of an MBConv block in one launch), FTC_FLAG_SE_HPART
   8: ftc_page_order, page_h / page_w arguments of ftc_page_merge (parallel page-level selection); FTC_FLAG_SE_INLINE
   9: FTC_FLAG_SE_INLINE removed (flag bit 0x20000000 is free again); FTC_MBHEAD_MAX_SQUEEZE; KBLOCK32 validation on CONV;
      ftc_page_merge_variant (the demo script's selection + two-pass seed rows)
   10: FTC_OP_FMBCONV (Fused-MBConv block with expansion in one launch: 3x3 expand + SiLU + 1x1 project + residual) */
#define FTC_ABI_VERSION 10

typedef enum ftc_status {
    FTC_OK = 0,
    FTC_ERR_INVALID = -1,      /* bad argument / unsupported shape */
    FTC_ERR_HIP = -2,          /* a HIP runtime call or launch failed */
    FTC_ERR_NO_DEVICE = -3,    /* no gfx950 device visible */
    FTC_ERR_NOMEM = -4
} ftc_status;

typedef enum ftc_dtype { FTC_F32 = 0, FTC_BF16 = 1, FTC_F16 = 2 } ftc_dtype;     /* FTC_F16: IEEE half (same MFMA rate as bf16, 3 more mantissa bits) */
/* ftc_create precision only: the fp32 plan (fp32 tensors and weights) with FTC_FLAG_SPLIT16 on every convolution */
#define FTC_PRECISION_F16X3 3

/* Address bases an op operand can be relative to; resolved at ftc_plan_run time. */
typedef enum ftc_base {
    FTC_BASE_NULL = 0,
    FTC_BASE_WORKSPACE = 1,    /* activation arena (caller-allocated, ftc_plan workspace bytes) */
    FTC_BASE_WEIGHTS = 2,      /* packed weight blob (caller-allocated, uploaded once) */
    FTC_BASE_INPUT = 3,        /* image batch */
    FTC_BASE_HEATMAP = 4,      /* [B,h,w,10] fp32 output */
    FTC_BASE_FEATURES = 5,     /* [B,h,w,100] fp32 output */
    FTC_BASE_GRADS = 6,        /* train step: the flat fp32 gradient buffer (every parameter's .grad is a view into it) */
    FTC_NUM_BASES = 7
} ftc_base;

typedef struct ftc_ref {
    int32_t base;              /* ftc_base */
    int32_t reserved;
    int64_t offset;            /* bytes from the base */
} ftc_ref;

#define FTC_MBHEAD_SLICE 128   /* expanded channels one workgroup of FTC_OP_MBHEAD owns */
#define FTC_MBHEAD_SLICE_F32 64   /* ... in the fp32-tensor form (in_dtype = FTC_F32 with FTC_FLAG_SPLIT16, ABI 9) */
#define FTC_MBHEAD_MAX_SQUEEZE 160   /* largest SE squeeze width (aux0) for which FTC_OP_MBHEAD forms the fc1 partial products */

typedef enum ftc_op_kind {
    /* conv3x3 stride 2 on the 3-channel image with x*2-1 fused (CenterNetDetection.forward,
       detector.py:218) + folded BN + SiLU  -- features[0] */
    FTC_OP_STEM = 1,
    /* dense 1x1 / 3x3 convolution as im2col-free implicit GEMM on MFMA, NHWC, with fused
       per-channel bias (folded BN), activation, residual add, optional per-(image,channel)
       SE scale on the input, channel-sliced input and output (free concat) */
    FTC_OP_CONV = 2,
    /* depthwise 3x3 (stride 1|2) + folded BN + SiLU, plus per-(image,channel) partial sums of
       the output for the SE squeeze */
    FTC_OP_DWCONV = 3,
    /* SE excitation: mean of the partial sums -> fc1+bias -> SiLU -> fc2+bias -> sigmoid */
    FTC_OP_SE = 4,
    /* FPN level input: bilinear x2 (align_corners=True) of the previous level, concatenated with
       the per-head BatchNorm of the backbone tap (Leafmap.forward, detector.py:192-201) */
    FTC_OP_UPCAT = 5,
    /* 3x3 max-pool NMS on heatmap channel 0 -> channel 1 (CenterNetDetector.forward,
       detector.py:291-296) */
    FTC_OP_NMS = 6,
    /* Training-mode BatchNorm, first half (BN-refresh pass, train1.py:203-211): per-channel batch statistics of `in` [B*H*W][Cin]
       (fp32 or 16-bit; float64 sums, fixed order) -> out = fp32 [4][Cin] (scale = gamma / sqrt(var + eps), shift = beta - mean * scale
       for FTC_OP_BNACT; mean and 1 / sqrt(var + eps) for FTC_OP_BNBWD), and the running statistics aux = fp32 [2][Cin] (mean | var) updated in place:
       r = (1 - momentum) * r + momentum * batch value (the variance unbiased, as torch.nn.BatchNorm2d does).
       w = gamma, bias = beta (fp32 [Cin]); aux0 / aux1 = the bit patterns of the floats eps / momentum; in2 = scratch for the partial sums,
       float64 [chunks = min(512, ceil(B*H*W / 256))][2][Cin] (2 launches: partial sums, finalize) */
    FTC_OP_BNSTAT = 8,
    /* Training-mode BatchNorm, second half + activation (+ stochastic depth + residual): out[b,y,x,c] = act(in * scale[c] + shift[c]);
       with FTC_FLAG_RESIDUAL: out = out * w2[b] + in2 (w2 = fp32 [B] keep-scales of torchvision's StochasticDepth "row" mode, may be NULL = 1).
       in: fp32 or 16-bit [B,H,W,Cin]; scale / shift: fp32 [Cin] (the two halves of BNSTAT's output); out: out_dtype; out2: optional 16-bit
       copy (w_dtype) when out is fp32; aux0 = row chunks per image (the launch is B x aux0 x Cin/64 workgroups); aux: optional
       fp32 [B][aux0][Cin] per-image partial channel sums of the result for the SE squeeze */
    FTC_OP_BNACT = 9,
    /* ---- train step (BASELINE configs[4]; /root/reference/train1.py:125-131, 170-179): the backward kernels.  All tensors fp32 NHWC;
       parameter gradients are ACCUMULATED (+=) into the FTC_BASE_GRADS buffer in the PyTorch parameter layout (conv OIHW), so that
       `optimizer.zero_grad()` = one memset and gradient accumulation over micro-batches (train1.py:176-179) needs nothing extra. */
    /* rows[i][0..Cin) = in[in2[i]][0..Cin), zero-padded to Cout_total columns, i < aux0 (`features[fmask]`, models/detector.py:265-266):
       in = features [P][Cin] fp32, in2 = int32 [aux0] ascending pixel indices, out = fp32 [aux0][Cout_total] */
    FTC_OP_GATHER_ROWS = 10,
    /* loss_function (loss_func.py:94-177) as a plan step = ftc_losses: in = maps [B,H,W,9], in2 = labelmap [B,5,H,W], w = idmap int32
       [B,2,H,W], w2 / bias / bias2 = decoder logits [aux0][1091 | 1093 | 1097], scale = int32 [aux0] selected pixels, out = fp32 [16],
       aux = scratch (ftc_losses_scratch_bytes) */
    FTC_OP_LOSSES = 11,
    /* d(sum_i alpha_i * loss_i) * loss_scale / d(maps, decoder logits): operands as FTC_OP_LOSSES plus shift = alphas fp32 [9] in the
       reference's key order (keymap, size, textline, separator, id, code1, code2, code4, code8: train1.py:107-114), aux = the fp32 [16]
       vector FTC_OP_LOSSES wrote (out[12], out[13] = the clamped weight sums), out = d maps [B,H,W,9], out2 = d logits, three blocks
       [aux0][aux1] (rows zero-padded to aux1 columns), Cout = the bit pattern of the float loss_scale (1 / iters_to_accumulate) */
    FTC_OP_LOSS_BWD = 12,
    /* out = zeros [B*H*W][Cout_total]; out[in2[i]][0..Cin_total) = in[i][0..Cin_total), i < aux0 (backward of GATHER_ROWS) */
    FTC_OP_SCATTER_ROWS = 13,
    /* Backward of BNSTAT + BNACT: incoming gradient g = in[r][cin_off + c] (row stride Cin_total, 0 = Cin) * bias[b][c] + bias2[b][c]
       (both optional: the SE gate and the squeeze-mean gradient of an MBConv block), * w2[b] (optional StochasticDepth keep-scale);
       t = z * scale + shift; dt = g * act'(t); out (+= with FTC_FLAG_ACCUM) = scale * (dt - mean(dt) - zhat * mean(dt * zhat));
       w (gamma grad) += sum dt * zhat, shift (beta grad) += sum dt.  in2 = z [B*H*W][Cin] (in_dtype: fp32, or stored in the 16-bit
       compute type w_dtype as the reference's autocast stores convolution outputs), scale = the [4][Cin] block of BNSTAT,
       aux = scratch float64 [chunks][2][Cin] + fp32 [2][Cin] (chunks as BNSTAT) */
    FTC_OP_BNBWD = 14,
    /* Weight gradient of a dense 1x1 / 3x3 convolution on MFMA: out[co][ci][r][s] += sum_p in2[p][cout_off + co] * in[p @ (r,s)][cin_off + ci]
       (* scale[b][ci] with FTC_FLAG_SE_SCALE).  in = the layer input [B,H,W,Cin_total], in2 = d output [B,Ho,Wo,Cout_total], w_dtype = the
       MFMA operand type (operands narrowed while they are staged), aux = fp32 partial sums [aux0 pixel splits][k*k][Cout][Cin], aux0 >= 1 */
    FTC_OP_WGRAD = 15,
    /* Depthwise 3x3 backward: in = layer input [B,H,W,Cin], in2 = d output [B,Ho,Wo,Cin], w = fp32 [9][Cin] -> out = d input,
       out2 ([Cin][1][3][3], grads) += d weight; aux = float64 [chunks][9][Cin] (chunks as BNSTAT over B*Ho*Wo) */
    FTC_OP_DWBWD = 16,
    /* SqueezeExcitation backward: in = d(y * s) [B,H*W,Cin], in2 = y, scale = s [B][Cin], aux = the forward's partial channel sums
       [B][aux1][Cin], w / w2 / bias / bias2 as FTC_OP_SE (aux0 = squeeze width S) -> out = fp32 scratch [4][B][Cin] + [2][B][S] + [32][B][Cin]:
       block 3 = d mean / (H*W) (the `bias2` of the following FTC_OP_BNBWD, whose `bias` is s); out2 = grads of fc1.weight [S][Cin],
       fc1.bias [S], fc2.weight [Cin][S], fc2.bias [Cin], consecutive, += */
    FTC_OP_SEBWD = 17,
    /* Backward of the upsampled part of UPCAT: out[b,y,x,0..aux0) = sum over the x2 bilinear (align_corners) footprint of
       in[b,Y,X,0..aux0) (row stride Cin_total); in = d cat [B,Ho,Wo,Cin_total], out = [B,H,W,aux0] */
    FTC_OP_UPCATBWD = 18,
    /* out [B,Ho,Wo,Cin] = in [B,H,W,Cin] with a zero between the pixels (Ho = 2H, Wo = 2W): turns the data gradient of a stride-2
       convolution into a stride-1 convolution with the flipped kernel */
    FTC_OP_DILATE = 19,
    /* Data gradient of a 3x3 convolution with very few output channels (the map heads' top_conv, 1 or 2): in = d maps [B,H,W,Cin_total]
       (channels cin_off .. cin_off + Cin), w = [Cin][9][Cout] (w_dtype), out = [B,H,W,Cout] fp32 */
    FTC_OP_TOPDGRAD = 20,
    /* out[c] += sum_r in[r][cin_off + c], c < Cin (bias gradients); in = [B*H*W][Cin_total]; aux = float64 [chunks][Cin] */
    FTC_OP_COLSUM = 21,
    /* Weight gradient of the stem (3 -> Cout <= 32, stride 2, input x*2-1): in = image [B,H,W,3], in2 = d output [B,Ho,Wo,Cout],
       out [Cout][3][3][3] +=; aux = float64 [chunks][27][Cout] */
    FTC_OP_STEMWGRAD = 22,
    /* out[0 .. B*H*W*Cin) fp32 = 0 */
    FTC_OP_FILL = 23,
    /* no kernel: ftc_plan_run_streams makes the main stream wait for everything issued on the side stream so far (a no-op in ftc_plan_run) */
    FTC_OP_JOIN = 24,
    /* MBConv head of a low-resolution stage in one launch (torchvision MBConv block[0], block[1] and the squeeze of block[2];
       /root/reference/models/detector.py:17-20): e = SiLU(in . w2^T + bias2) rounded to the 16-bit type -- the expand 1x1 convolution with
       its folded BatchNorm --, out = SiLU(depthwise3x3(e; w) + bias) (stride 1, zero padding), aux[b][c] = sum_{y,x} out[b,y,x,c] (fp32,
       summed before the result is narrowed: the P = 1 form of FTC_OP_DWCONV's partial sums, consumed by FTC_OP_SE with aux1 = 1).
       A workgroup owns one image x FTC_MBHEAD_SLICE expanded channels; the expanded tensor only ever exists in LDS (csrc/mbconv_slice.hip).
       in [B,H,W,Cin] 16-bit, w2 [Cout][Cin] (K-major, same type), bias2 fp32 [Cout], w fp32 [9][Cout], bias fp32 [Cout], out [B,H,W,Cout]
       16-bit; in_dtype == out_dtype == w_dtype; Cin % 32 == 0, Cout % FTC_MBHEAD_SLICE == 0, H*W <= 576 and H*(W+1) < 601 (a 24x24 map).
       Optional: scale = the SE fc1 weight fp32 [aux0][Cout] and out2 = fp32 [B][Cout/FTC_MBHEAD_SLICE][aux0]: out2[b][j][s] = sum over the
       channels c of slice j of scale[s][c] * mean_hw(out[b,:,:,c]) -- FTC_OP_SE with FTC_FLAG_SE_HPART adds the slices' vectors.
       Band mode for larger maps (the 48x48 stages), aux1 = R > 0: a workgroup owns R output rows of an image (nb = ceil(H / R) bands) and
       recomputes one expanded halo row above and below; (R + 2) * W <= 576 and (R + 2) * (W + 1) < 601 then replace the whole-map limits,
       aux = [B][nb][Cout] per-band channel sums (FTC_OP_SE: aux1 = nb) and out2 = [B][nb * Cout/FTC_MBHEAD_SLICE][aux0].
       Round 5 (ABI 9): Cout_total = the slice width (0 = the default: 128; 96 where 128-channel slices leave CUs idle), Cout % slice == 0, out2 and
       FTC_OP_SE's aux1 count Cout / slice slices; the fp32-tensor form (in_dtype = out_dtype = w_dtype = FTC_F32 with FTC_FLAG_SPLIT16, csrc/mbconv_slice_x3.hip):
       `in` and w2 PRE-SPLIT (see FTC_FLAG_PRESPLIT), 64-channel slices, fp32 `out` (pre-split with FTC_FLAG_PRESPLIT) */
    FTC_OP_MBHEAD = 25,
    /* Fused-MBConv block with expansion (torchvision FusedMBConv, expand_ratio != 1; /root/reference/models/detector.py:14-16) in one launch
       (csrc/fused_mbconv.hip):  e = act(conv3x3(in) + bias2)  [Cin -> E = aux1, 3x3 stride 1 "same"],  out = conv1x1(e) + bias (+ in2)  [E -> Cout].
       in [B,H,W,Cin] 16-bit, w2 [E][9][Cin] (K-major, same type), bias2 fp32 [E], w [Cout][E] (same type), bias fp32 [Cout], in2 fp32 [B,H,W,Cout]
       with FTC_FLAG_RESIDUAL, out fp32 [B,H,W,Cout], out2 = optional 16-bit copy of out (NHWC).  e is rounded to the 16-bit type exactly as
       the two-launch form stores it, but never leaves the CU.  Cin % 32 == 0, aux1 in {256, 384}, Cout % 32 == 0, Cout <= 128, act = FTC_ACT_SILU. */
    FTC_OP_FMBCONV = 26,
    FTC_OP_TAPSUM = 7          /* second half of a 3x3 convolution split as per-pixel taps + 9-point sum (FTC_FLAG_TOP_FUSE):
                                  out[b,y,x,ch_j] = bias[j] + sum_{r,s} in[g_j][b,y+r-1,x+s-1][(3r+s)*co_j + o_j] (zero outside),
                                  for the aux1 outputs j listed in `w` as int32 quadruples (g_j, o_j, co_j, ch_j);
                                  in = T [groups][B,H,W][aux0] fp32, out = [B,H,W,Cout_total] fp32 */
} ftc_op_kind;

enum {
    FTC_ACT_NONE = 0, FTC_ACT_SILU = 1, FTC_ACT_GELU = 2
};

enum {
    FTC_FLAG_RESIDUAL = 1,     /* out += in2 (after activation) */
    FTC_FLAG_SE_SCALE = 2,     /* input multiplied by scale[b, cin] while staging (CONV) */
    FTC_FLAG_IN_NCHW = 4,      /* STEM: input is [B,3,H,W]-contiguous instead of NHWC */
    FTC_FLAG_BORDER_BIAS = 8,  /* CONV 3x3 s1: `bias` is a [16][Cout] table indexed by which image borders the
                                  output pixel touches (top | bottom<<1 | left<<2 | right<<3): lets a per-channel
                                  affine (BatchNorm) that PRECEDES a zero-padded conv be folded into it exactly */
    FTC_FLAG_W_PER_IMAGE = 16, /* CONV: `w` holds B weight sets [B][Cout][k*k][Cin], image b uses set b (the SE
                                  excitation folded into the project weights, see FTC_FLAG_SE_FOLD).  The pixel
                                  tile of a workgroup must not straddle images: Ho*Wo % tile rows == 0 */
    FTC_FLAG_GROUP_IN_SLICE = 64,  /* UPCAT with groups > 1: the upsampled inputs of the groups are channel slices of ONE
                                  tensor (group g reads channels cin_off + g*aux0 ..) instead of stacked tensors */
    FTC_FLAG_GROUP_OUT_SLICE = 128, /* CONV with groups > 1: the groups write channel slices of ONE tensor (group g writes
                                  channels cout_off + g*Cout ..) instead of stacked tensors */
    FTC_FLAG_TOP_FUSE = 0x10000, /* CONV 3x3 (LDS-halo kernel, one 192-channel tile): the activated output tile is not
                                  stored; instead T[p][0..32) = tile[p][:] . w2[0..32)[:] is computed on it and its first
                                  aux1 values per pixel go to `out` = T [groups][B,Ho,Wo][aux1] fp32 (w2 = [groups][32][Cout] in
                                  the compute type: bf16 / fp16 -- one more MFMA GEMM on the tile's 16-bit LDS image -- or, with
                                  fp32 tensors (fp32 and fp16x3 plans; aux1 <= 20), plain fp32 applied in fp32 FMA);
                                  FTC_OP_TAPSUM finishes the following top convolution */
    FTC_FLAG_UPCAT_IN = 0x20000, /* CONV 3x3 stride 1 (bf16 LDS-halo kernel, 192-channel tiles): the input is the concatenation the
                                  reference builds with UpsamplingBilinear2d + cat (models/detector.py:192-201), formed while the
                                  halo is staged: channels [0, Cin_total) = x2 bilinear upsample (align_corners) of `in`
                                  [groups][B,H/2,W/2,Cin_total], channels [Cin_total, Cin) = in2 [groups][B,H,W,Cin-Cin_total];
                                  both channel counts multiples of the kernel's K block (64, or 32 when Cin % 64 != 0) */
    FTC_FLAG_GROUP_IN2_SHARED = 0x200000, /* CONV + UPCAT_IN with groups > 1: in2 is ONE tensor [B,H,W,Cin-Cin_total] read by every group
                                  (the backbone tap; its per-head BatchNorm folded into the weights and a BORDER_BIAS table) */
    FTC_FLAG_W_FRAG = 0x400000, /* CONV 3x3 stride 1, 16-bit, Cout = 192, Cin % 64 == 0, aux0 bits 6+7 (weights-through-L1 kernel): `w` is packed
                                  FRAGMENT-MAJOR -- [groups][6 row blocks of 32][9 taps][Cin/64][4 K groups of 16][64 lanes][8]: element e of
                                  lane L = W[32*rb + (L & 31)][tap][64*cb + 16*g + 8*(L >> 5) + e] -- so that a wave reads an MFMA A fragment
                                  as one coalesced 1 KiB load straight from global memory (the weights never touch LDS) */
    FTC_FLAG_SPLIT16 = 0x2000000, /* CONV with fp32 operands (w_dtype = in_dtype = out_dtype = FTC_F32): "fp16x3" arithmetic -- every fp32 operand is
                                  split into hi + lo IEEE halves while the MFMA fragments are read from LDS and a product becomes three
                                  v_mfma_f32_32x32x16_f16 (hi.lo + lo.hi + hi.hi, fp32 accumulation) instead of eight v_mfma_f32_32x32x2_f32:
                                  22-bit operands at up to 5.3x the fp32 matrix rate.  Tensors, weights, epilogues: those of the fp32 mode */
    FTC_FLAG_ACCUM = 0x800000, /* BNBWD / CONV-as-dgrad helpers: the data-gradient output is added to what `out` holds */
    FTC_FLAG_SIDE_STREAM = 0x4000000, /* any op: ftc_plan_run_streams enqueues it on the side stream (after everything issued on the main stream so
                                  far); whoever builds the plan keeps the op's operands alive and unwritten until the next FTC_OP_JOIN */
    FTC_FLAG_SE_HPART = 0x8000000, /* SE: `aux` holds per-slice partial products of the fc1 layer, fp32 [B][aux1][aux0] (FTC_OP_MBHEAD's out2, aux1 =
                                  Cout/FTC_MBHEAD_SLICE slices), instead of partial channel sums: hidden = SiLU(bias + sum_j aux[b][j][:]); `w` (fc1 weight) unused */
    FTC_FLAG_KBLOCK32 = 0x10000000, /* a 16-bit activation tensor stored in 32-channel planes, [B][C/32][H*W][32] instead of NHWC [B][H*W][C]: a pixel's 32
                                  channels of one plane are 64 contiguous bytes and consecutive pixels follow each other, so the K step of 32 that
                                  FTC_OP_MBHEAD streams per stage is whole cache lines (from NHWC it fetched half of every 128-byte line per step and
                                  ran at the L1 fill rate).  On FTC_OP_CONV: the layout of `out2` (the 16-bit trunk copy; Cout % 32 == 0); on
                                  FTC_OP_MBHEAD: the layout of `in` */
    FTC_FLAG_PRESPLIT = 0x20000000, /* fp16x3 plans (with FTC_FLAG_SPLIT16; ABI 9): an fp32 activation tensor stored PRE-SPLIT -- every 16-byte chunk of four values as
                                  [hi x4 | lo x4] IEEE halves, what the three-MFMA product consumes, so that the consumer's fragments cost no VALU work and the
                                  22 bits it would have used are exactly the ones stored.  CONV 1x1 without SE scale: the layout of `in`; MBHEAD: the layout of `out`
                                  (its `in` always is pre-split); CONV out2 of an fp16x3 convolution always is */
    FTC_FLAG_SE_FOLD = 32      /* SE: besides scale[b,c], write out2[b][n][c] = bf16(in[n][c] * scale[b,c]) for the
                                  bf16 matrix `in` [Cout_total][C] -- the following 1x1 convolution then runs with
                                  FTC_FLAG_W_PER_IMAGE on unscaled activations (both operands by DMA) */
};

/* One step of a plan.  Fields that an op kind does not use must be zero. */
typedef struct ftc_op {
    int32_t kind;              /* ftc_op_kind */
    int32_t flags;
    int32_t act;               /* FTC_ACT_* */
    int32_t in_dtype;          /* ftc_dtype of `in` (and `in2` for UPCAT) */
    int32_t out_dtype;         /* ftc_dtype of `out` */
    int32_t w_dtype;           /* ftc_dtype of `w` = MFMA compute type (CONV) */
    int32_t B, H, W;           /* input batch / height / width */
    int32_t Ho, Wo;            /* output height / width */
    int32_t Cin;               /* input channels consumed */
    int32_t Cin_total;         /* channel stride of the input buffer (>= cin_off + Cin);
                                  UPCAT: channel stride of `in` (the tensor being upsampled) */
    int32_t cin_off;
    int32_t Cout;              /* output channels produced */
    int32_t Cout_total;        /* channel stride of the output buffer */
    int32_t cout_off;
    int32_t ksize;             /* 1 or 3 */
    int32_t stride;            /* 1 or 2 */
    int32_t aux0;              /* DWCONV: number of row-strips P;  SE: squeeze channels;
                                  UPCAT: channels of the upsampled part (0 = none);
                                  CONV: tuned kernel choice, 0 = the library's heuristics (bits 0-3 tile config + 1 -- 8 | 9 | 10 = the
                                  64 | 80 | 128-channel x 144-pixel 1x1 kernel for 16-bit operands and fp32 output, Cin % 64 == 0,
                                  Cout % tile == 0, Ho*Wo % 144 == 0 --, bits 4-5 staging, 6-7 LDS-halo kernels, 8-9 K step, 10-11 split-K:
                                  csrc/conv_igemm_impl.h; every choice computes the same convolution, the 144-pixel tiles and
                                  split-K in another summation order) */
    int32_t aux1;              /* SE: number of partial sums P;  UPCAT: tap channels;  CONV+TOP_FUSE: floats per pixel of T;
                                  TAPSUM: number of outputs (aux0 = floats per pixel of T) */
    int32_t res_dtype;         /* ftc_dtype of in2 (CONV residual / UPCAT tap) */
    int32_t groups;            /* CONV / UPCAT: G > 1 runs G independent instances of the op in ONE launch (the nine FPN
                                  heads share every shape).  Operands of instance g are stacked, g-major: CONV in
                                  [G][B,H,W,Cin_total], w [G][Cout][k*k][Cin], bias [G][rows][Cout], out [G][B,Ho,Wo,Cout_total];
                                  UPCAT in [G][B,H,W,Cin_total], scale/shift [G][aux1], out [G][B,Ho,Wo,C], in2 (the backbone
                                  tap) shared.  See FTC_FLAG_GROUP_IN_SLICE / _OUT_SLICE.  0 or 1 = a single instance */
    int32_t reserved0;         /* must be 0 */
    ftc_ref in;                /* main input */
    ftc_ref in2;               /* CONV: residual [B,Ho,Wo,Cout];  UPCAT: backbone tap [B,Ho,Wo,aux1];
                                  SE: hidden-unit scratch fp32 [B,aux0] */
    ftc_ref out;
    ftc_ref w;                 /* CONV: [Cout][k*k][Cin] (K-major);  DWCONV: [9][C] fp32;
                                  STEM: [27][Cout] fp32;  SE: fc1 [S][C] fp32 */
    ftc_ref w2;                /* SE: fc2 transposed [S][C] fp32 */
    ftc_ref bias;              /* fp32 [Cout] (SE: fc1 bias [S]) */
    ftc_ref bias2;             /* SE: fc2 bias [C] */
    ftc_ref scale;             /* CONV+SE_SCALE: fp32 [B,Cin];  UPCAT: BN scale fp32 [aux1] */
    ftc_ref shift;             /* UPCAT: BN shift fp32 [aux1] */
    ftc_ref aux;               /* DWCONV: partial sums out fp32 [B,P,C];  SE: partial sums in */
    ftc_ref out2;              /* CONV / STEM with fp32 `out`: optional bf16 copy [B,Ho,Wo,Cout] of the same
                                  values (the fp32 tensor feeds the residual adds, the copy feeds the
                                  next bf16 GEMM without a conversion pass);  SE+SE_FOLD: the B scaled
                                  weight sets bf16 [B][Cout_total][C] */
} ftc_op;

typedef struct ftc_plan ftc_plan;

/* Library / device ------------------------------------------------------------------------- */
int ftc_abi_version(void);
/* Thread-local message of the last failing call ("" if none). */
const char* ftc_last_error(void);
/* FTC_OK iff the current HIP device is a gfx950 part; writes CU count and name when non-NULL. */
int ftc_device_info(int* n_cu, char* name, int name_len);

/* Plan ------------------------------------------------------------------------------------- */
/* Validates and copies `ops`; `workspace_bytes` is recorded for bounds checks of workspace refs. */
int ftc_plan_create(const ftc_op* ops, int n_ops, int64_t workspace_bytes, int64_t weights_bytes,
                    ftc_plan** out);
void ftc_plan_destroy(ftc_plan* plan);
int ftc_plan_num_ops(const ftc_plan* plan);
/* Enqueues every op on `stream` (a hipStream_t; NULL = the default stream).  `bases[i]` is the
   device address for ftc_base i (bases[0] ignored).  `first_op..last_op` (inclusive, -1 = end)
   selects a sub-range, used by the per-op parity tests and the profiler harness. */
int ftc_plan_run(const ftc_plan* plan, void* const bases[FTC_NUM_BASES], void* stream,
                 int first_op, int last_op);
/* ftc_plan_run on two streams: ops carrying FTC_FLAG_SIDE_STREAM go to `side_stream` (which first waits for the main stream's work so
   far), FTC_OP_JOIN and the end of the range make `stream` wait for the side stream.  side_stream == NULL: exactly ftc_plan_run.
   The train step's weight gradients -- a quarter of its time, nothing on the backward chain reads them -- overlap the HBM-bound
   BatchNorm / depthwise passes this way (tools/train_two_stream_experiment.py: 126 -> 116 ms). */
int ftc_plan_run_streams(const ftc_plan* plan, void* const bases[FTC_NUM_BASES], void* stream, void* side_stream,
                         int first_op, int last_op);
/* Same as ftc_plan_run, bracketing every op with HIP events on `stream` and returning the
   per-op elapsed milliseconds in ms_out[n_ops] (synchronises; measurement only). */
int ftc_plan_profile(const ftc_plan* plan, void* const bases[FTC_NUM_BASES], void* stream,
                     float* ms_out);

/* Label of the kernel instantiation an op dispatches to (dtype + tile configuration), e.g.
   "conv_igemm<bf16,in=bf16,out=bf16,tile=192x128>"; used to attribute rocprof / HIP-event time. */
int ftc_op_kernel_label(const ftc_op* op, char* buf, int len);

/* Model ------------------------------------------------------------------------------------ */
/*
 * The self-contained entry points a non-Python host binds: the network graph, BatchNorm folding, K-major weight packing,
 * the activation arena and the measured kernel selection all live in the library.  They replace, for the detector path,
 *   TextDetectorModel(...).load_state_dict(...) + CenterNetDetector(model.detector)   (/root/reference/process_ocr_torch.py:12-27)
 *   heatmap, features = detector(images)                                              (/root/reference/process_ocr_torch.py:43-49,
 *                                                                                       /root/reference/models/detector.py:289-296)
 */
typedef struct ftc_tensor {
    const char* name;          /* state_dict key of the reference checkpoint, with or without the "detector." prefix
                                  (e.g. "detector.backbone.features.4.0.block.2.fc1.weight"); "decoder.*" keys are ignored */
    const void* data;          /* HOST pointer, contiguous, row-major in the PyTorch shape (conv weights OIHW) */
    int32_t dtype;             /* FTC_F32; tensors of any other dtype (num_batches_tracked) are skipped */
    int32_t ndim;              /* 0..4 */
    int64_t shape[4];
} ftc_tensor;

typedef struct ftc_model ftc_model;

/* Folds and packs the checkpoint (host only; a few seconds for the 242 M detector parameters).  model_size: "xl" (default when
   NULL), "l", "m", "s" (models/detector.py:131-136); precision: FTC_F32 = parity mode (what the reference computes), FTC_BF16 =
   speed mode (bf16 MFMA, fp32 accumulation / residual trunk / outputs), FTC_F16 = the same plan with IEEE-half operands (same
   matrix rate, 11-bit significands: ~8x closer to the fp32 result than bf16; activations saturate at +-65504).  Fails with FTC_ERR_INVALID naming the first missing
   or mis-shaped tensor.  The tensors may be freed after the call. */
int ftc_create(const ftc_tensor* tensors, int n_tensors, const char* model_size, int precision, ftc_model** out);
/* precision may also be FTC_PRECISION_F16X3: the parity-grade fast mode (fp32 everywhere except the multiplier: see FTC_FLAG_SPLIT16) */
void ftc_destroy(ftc_model* model);
/* The packed weight blob: the caller copies ftc_weights_bytes() bytes from ftc_weights_host() into device memory
   (256-byte aligned) once and passes that address to every ftc_forward. */
int64_t ftc_weights_bytes(const ftc_model* model);
const void* ftc_weights_host(const ftc_model* model);
/* Byte offset of a packed tensor inside the blob ("<layer>.w" / "<layer>.b", e.g. "heads.L0.w"), -1 if absent (tests). */
int64_t ftc_weights_offset(const ftc_model* model, const char* name);
/* Activation arena the caller must provide for this input shape (builds and caches the plan); -1 on error. */
int64_t ftc_workspace_bytes(ftc_model* model, int B, int H, int W);
/*
 * image: [B,H,W,3] fp32 in 0..1 (NHWC; nchw != 0: [B,3,H,W]) in device memory, H and W multiples of 32;
 * heatmap [B,H/4,W/4,10] fp32, features [B,H/4,W/4,100] fp32 (device, caller-owned, NHWC);  with_nms = 0 leaves heat-map
 * channel 1 untouched (CenterNetDetection.forward), != 0 fills it (CenterNetDetector.forward).  Enqueues on `stream`, never
 * synchronises, allocates nothing.  One model may be used from several host threads with separate workspaces and streams.
 */
int ftc_forward(ftc_model* model, const void* weights_dev, const void* image, int B, int H, int W, int nchw, int with_nms,
                void* heatmap, void* features, void* workspace, void* stream);

/* Introspection (parity tests, profiling, the tuner): the plan ftc_forward runs for a shape -- borrowed, owned by the model. */
typedef struct ftc_plan_info {
    int32_t n_ops, map_h, map_w, reserved;
    int64_t workspace_bytes, weights_bytes, peak_live_bytes, total_buffer_bytes;
} ftc_plan_info;
typedef struct ftc_op_info {
    char name[64];             /* reference module path of the op, e.g. "backbone.features.4.0.block.3" */
    char kind[16];             /* stem | conv1x1 | conv3x3 | dwconv3x3 | se | upcat | tapsum | nms */
    double flops;              /* 2 * MACs of the convolution (bias / activation excluded) */
    double bytes;              /* algorithmic bytes: inputs + outputs + weights, each once */
} ftc_op_info;
int ftc_model_plan(ftc_model* model, int B, int H, int W, int nchw, const ftc_plan** plan, ftc_plan_info* info);
int ftc_model_op_info(ftc_model* model, int B, int H, int W, int nchw, int index, ftc_op_info* out);
int ftc_plan_op(const ftc_plan* plan, int index, ftc_op* out);

/* Peak decode ------------------------------------------------------------------------------ */
/* Per-image geometry of the tile being decoded (process_ocr_base.py:487-503). */
typedef struct ftc_tile {
    int32_t offset_x, offset_y;        /* tile origin on the page, input pixels */
    int32_t page_w, page_h;            /* page size, input pixels (boxes wider/taller are dropped) */
    int32_t x_min, x_max, y_min, y_max;/* trusted rectangle in map pixels, max exclusive */
} ftc_tile;

/*
 * GPU replacement of the per-tile host loop (process_ocr_base.py:518-538 == test_image1_torch.py:
 * 123-143): keeps pixels of heatmap channel 1 (NMS'd key logit) inside the tile's trusted
 * rectangle whose logit >= logit_cut, drops boxes with w,h <= 0 or larger than the page, orders
 * them by (score desc, pixel index asc) and writes for the first `max_boxes` of them
 *   boxes[(b*max_boxes + i)*box_stride + 0..8]  = p, ix, iy, w, h, code1, code2, code4, code8   (fp32)
 *   feats[(b*max_boxes + i)*feat_stride + 0..C) = features[b, y, x, :]                           (fp32)
 *   index[b, i]                                 = y * w + x                                      (int32)
 * Row strides are in floats: box_stride = 9, feat_stride = C give two dense arrays; box_stride = feat_stride = 112 with
 * feats = boxes + 12 gives ONE record block [B, max_boxes, 112] (box, 3 pad, feature row 16-byte aligned) -- the message
 * of the multi-GPU box gather, written without a concatenation pass.
 * counts[b] receives the TOTAL number of kept peaks (may exceed max_boxes: caller detects
 * truncation).  heatmap [B,h,w,10] fp32 NHWC, features [B,h,w,C] fp32 NHWC, tiles_dev = B
 * ftc_tile records in DEVICE memory, scratch_dev >= ftc_decode_scratch_bytes(B,h,w) bytes.
 */
int64_t ftc_decode_scratch_bytes(int B, int h, int w);
int ftc_decode(const float* heatmap, const float* features, int B, int h, int w, int C,
               const ftc_tile* tiles_dev, float logit_cut, int scale, int max_boxes,
               float* boxes, int box_stride, float* feats, int feat_stride, int32_t* index, int32_t* counts,
               void* scratch_dev, void* stream);

/* Page front / back end ("next" rows of SURVEY.md 8f) ----------------------------------------- */
/*
 * Tiling front-end of OCR_Processer.call_OCR (process_ocr_base.py:67-76): cuts B tiles of
 * tile_h x tile_w out of a uint8 RGB page resident in device memory and writes them as
 * [B,tile_h,tile_w,3] fp32 = pixel / 255 (process_ocr_torch.py:44).  origins_yx_dev = B (y, x) int32
 * pairs in device memory; pixels beyond the page read as 255 (the reference's white padding, :63-65).
 */
int ftc_tile_gather(const unsigned char* page_u8, int page_h, int page_w, const int32_t* origins_yx_dev, int B, int tile_h,
                    int tile_w, float* tiles_out, void* stream);
/*
 * np.maximum paste of the masked sigmoid maps of B tiles into page canvases (process_ocr_base.py:505-516).
 * canvases = [7][page_mh][page_mw] fp32 (key, textline, separator, code1, code2, code4, code8), zeroed by
 * the caller before the first batch of a page; merges with atomic max (values >= 0), so the result does
 * not depend on tile order.
 */
int ftc_paste_maps(const float* heatmap, const ftc_tile* tiles_dev, int B, int h, int w, int scale, float* canvases, int page_mh,
                   int page_mw, void* stream);

/* Page-level box selection ----------------------------------------------------
 * Replaces the host loop of OCR_Processer.run_detector, /root/reference/process_ocr_base.py:559-650: contrast filter
 * (imageHist :652-693), greedy suppression in score order (IoU > 0.5, intersection > 0.75 of the box, > 50 % of the box
 * covered by kept boxes), separator filter, 3x3 maximum of the code maps.  Float64 arithmetic, bit-identical results.
 *
 * locations  [N,9] fp32 rows (p, cx, cy, w, h, c1, c2, c4, c8) in the reference's concatenation order (its leading all-zero row
 *            included or not: a row with p < cut_off is inert);  page [page_h,page_w,3] fp32 0..255;
 * ftc_box_hists   -> hist_out [2][N] float64: row 0 = the contrast of the threshold sample (:563-571), row 1 = of the crop
 *                    tested in the loop (:579-582).  The caller takes threshold = median(row 0 over p >= cut_off) / 5.
 * ftc_page_order  -> order_out [N] int32 = the rows with p >= cut_off in stable score order (= the front of the stable argsort of -p; ties:
 *                    lower row first), then the rows below the cut-off in row order (the selection never reaches them), and threshold_out [1] float64 =
 *                    median(row 0 over p >= cut_off) / 5 (NaN without such rows), both on the device: rank by counting and a radix
 *                    select, no library sort (round 4)
 * ftc_page_merge  <- order [N] int32 = stable argsort of -p;  hist1 = row 1 above;  threshold_dev = 1 float64 on the device
 *                    (NaN = no sample: nothing is dropped, as NumPy's comparison with NaN);  seps [mh,mw], codes [4][mh,mw]
 *                    fp32 page canvases (ftc_paste_maps rows 2 and 3..6)
 *                 -> out_locations [<=N,9] fp32 (codes updated), out_index [<=N] int32 source rows, out_count [1] int32
 *                    (-1: scratch too small for a box's coverage bitmap), all on the device; kept order = score order.
 *                    Round 4: the suppression runs in parallel -- neighbour lists of overlapping candidates, then persistent waves
 *                    resolve the candidates in rank order, each waiting only for its earlier overlapping neighbours; every comparison
 *                    is the sequential loop's own float64 expression (bit-identical).  Neighbour lists that outgrow the scratch block
 *                    route the page through the sequential kernel on the device (FTC_PAGE_MERGE_SEQ=1 forces it).  page_h, page_w
 *                    (input pixels; ABI 8) size the coverage image inside `scratch`; the lists take the rest of the block. */
int64_t ftc_page_merge_scratch_bytes(int n_boxes, int page_h, int page_w);
int ftc_box_hists(const float* locations, int n_boxes, const float* page, int page_h, int page_w, float cut_off, double* hist_out,
                  void* stream);
int64_t ftc_page_order_scratch_bytes(int n_boxes);
int ftc_page_order(const float* locations, int n_boxes, const double* hist0, float cut_off, int32_t* order_out, double* threshold_out,
                   void* scratch, int64_t scratch_bytes, void* stream);
int ftc_page_merge(const float* locations, const int32_t* order, int n_boxes, const double* hist1, const double* threshold_dev,
                   float cut_off, const float* seps, const float* codes, int mh, int mw, int scale, int page_h, int page_w,
                   float* out_locations, int32_t* out_index, int32_t* out_count, void* scratch, int64_t scratch_bytes, void* stream);
/* The same selection in the variant of the reference's demo script (ABI 9): eval() of /root/reference/test_image1_torch.py:152-240 differs from
 * OCR_Processer.run_detector in three ways -- no contrast filter (hist1 / threshold_dev unused, may be NULL); the coverage image is filled with
 * the offsets of :196-200 (p2x without the +1, p1y with a +1); and its two-pass mode (:313-332) appends the boxes of a coarse first pass,
 * multiplied by the shrink factor in float64 (`locations0[:,1:] * s`): rows [seed_start, n_boxes) are such seed rows, given UNSCALED in
 * fp32, and their columns 1..8 are multiplied by seed_scale in float64 wherever the selection reads them (seed_start < 0 or >= n_boxes: none).
 * out_locations holds the selected fp32 rows as given (codes updated as in the production variant, from the unscaled values); out_code_max
 * (optional, [<=N,4] fp32) the 3x3 code-map maxima themselves (-inf where the centre lies outside the page), so that a caller can form
 * eval()'s float64 result rows: max(code maximum, code column * seed_scale).  variant = FTC_PAGE_MERGE_PRODUCTION: exactly ftc_page_merge. */
#define FTC_PAGE_MERGE_PRODUCTION 0
#define FTC_PAGE_MERGE_DEMO 1
int ftc_page_merge_variant(const float* locations, const int32_t* order, int n_boxes, const double* hist1, const double* threshold_dev,
                           float cut_off, const float* seps, const float* codes, int mh, int mw, int scale, int page_h, int page_w, int variant,
                           int seed_start, double seed_scale, float* out_locations, int32_t* out_index, float* out_code_max, int32_t* out_count,
                           void* scratch, int64_t scratch_bytes, void* stream);

/* Validation / training-step adjuncts (forward only) --------------------------------------------
 * The reference's validation step (train1.py:133-139 test_step, eval mode): fmask = model.get_fmask(labelmap) ->
 * heatmap, decoder_outputs = model(image, fmask) -> loss_function(...) -> CoVWeightingLoss(...).  All pointers are device memory. */

/* TextDetectorModel.get_fmask (models/detector.py:270-281): mask[i] = 1 for the k largest of values[0..n) (ties at the k-th value:
   lowest index first, as the reference's stable sort), sel_index[0..k) = the selected indices ascending (= row order of `x[mask]`),
   count[0] = number selected (min(k, n)).  sel_index / count may be NULL. */
int ftc_topk_mask(const float* values, int64_t n, int64_t k, unsigned char* mask, int32_t* sel_index, int32_t* count, void* stream);
/* Index list of an arbitrary boolean mask (`features[fmask]`, models/detector.py:265-266): sel_index[0..min(count, cap)) ascending. */
int ftc_mask_compact(const unsigned char* mask, int64_t n, int32_t* sel_index, int64_t cap, int32_t* count, void* stream);
/* rows[i][0..C) = features[sel_index[i]][0..C), zero-padded to c_pad columns (c_pad % 8 == 0), for i < min(count, cap); rows at and
   beyond count are zero.  features = the NHWC block [P, C] fp32 the detector wrote; rows in `out_dtype` (FTC_F32 | FTC_BF16). */
int ftc_gather_rows(const float* features, const int32_t* sel_index, const int32_t* count, int64_t cap, int C, int c_pad, void* rows,
                    int out_dtype, void* stream);
/* SimpleDecoder.forward in eval mode (models/detector.py:232-254): three MLPs 100 -> 2048 -> 2048 -> {1091, 1093, 1097} with
   BatchNorm1d folded into the Linear layers and exact GELU, as 1x1 implicit GEMMs on the conv kernel.  The model must have been
   created from a checkpoint that contains the "decoder.*" tensors.  rows [n_rows, 128] in the model's compute dtype
   (ftc_gather_rows with c_pad = 128), out[j] [n_rows, modulo_j] fp32. */
int64_t ftc_decoder_workspace_bytes(ftc_model* model, int n_rows);
int ftc_decoder_forward(ftc_model* model, const void* weights_dev, const void* rows, int n_rows, float* out0, float* out1, float* out2,
                        void* workspace, void* stream);
/* loss_function (loss_func.py:94-177, heatmap_loss :74-92).  heatmap = the NINE reference channels addressed through element strides
   (batch, channel, y, x) so that NHWC and NCHW memory are both accepted; labelmap [B,5,h,w] fp32 and idmap [B,2,h,w] int32 contiguous;
   dec0..2 [cap, 1091 | 1093 | 1097] fp32 decoder outputs of the pixels sel_index[0..count) (may all be NULL: id_loss = 0).
   out[0..14) = loss, keymap, size, textline, separator, id, code1, code2, code4, code8, correct, total, max(1, sum weight1),
   max(1, sum weight3) (fp32; the last two are the normalisers of size_loss / id_loss, consumed by FTC_OP_LOSS_BWD).
   scratch >= ftc_losses_scratch_bytes() bytes. */
int64_t ftc_losses_scratch_bytes(void);
int ftc_losses(const float* heatmap, const int64_t heat_strides[4], const float* labelmap, const int32_t* idmap, int B, int h, int w,
               const float* dec0, const float* dec1, const float* dec2, const int32_t* sel_index, const int32_t* count, int64_t cap,
               float* out, void* scratch, void* stream);
/* CoVWeightingLoss.forward (loss_func.py:24-72): one step for n <= 16 losses.  state = 80 floats (zero-initialised before iteration
   0): running mean of L, mean of l, S_l, std_l, alphas (16 each).  out_loss[0] = sum(alphas * losses). */
int ftc_cov_weighting_step(const float* losses, int n, int iteration, float* state, float* out_loss, void* stream);

/* Schedule-Free AdamW step as one multi-tensor kernel ------------------------------
 * Replaces the ten torch._foreach_* passes of AdamWScheduleFree.step, /root/reference/models/adamw_schedulefree.py:157-184.
 * chunks_dev: device array; every entry is a run of <= 4096 fp32 elements of one parameter (16-byte aligned) with its
 * gradient, exp_avg_sq and z.  The scalars are the fp32 values the reference passes to ATen for this step
 * (findtextcenternet_amd/optim.py computes them in float64 as the reference does):
 *   one_minus_beta2 = 1-beta2, bias_correction2 = 1-beta2^(k+1), ckp1 = weight/weight_sum, y_alpha = lr*(beta1*(1-ckp1)-1).
 * write_grad != 0 also stores the normalised gradient back, as the reference does in place. */
typedef struct ftc_mt_chunk {
    void* y;                   /* parameter (the optimizer's y iterate while training) */
    void* g;                   /* gradient */
    void* v;                   /* exp_avg_sq */
    void* z;                   /* z iterate */
    int32_t n;                 /* elements in this chunk */
    int32_t reserved;
} ftc_mt_chunk;
int ftc_adamw_schedulefree_step(const ftc_mt_chunk* chunks_dev, int n_chunks, float beta2, float one_minus_beta2, float bias_correction2,
                                float eps, float weight_decay, float ckp1, float y_alpha, float lr, int write_grad, void* stream);

/* Train step: one multi-tensor launch that re-packs the raw parameters an optimizer step changed into the layouts the kernels read
 * (replaces ~1500 small permute / cast launches per step).  Every entry converts one fp32 OIHW convolution / Linear weight
 * [Cout][Cin][k][k] into   fwd  = [Cout][k*k][cin_pad]            (K-major, `dtype`; columns >= Cin zero)   and, if dgrad != NULL,
 *                          dgrad = [Cin][k*k flipped][cout_pad]    (the data-gradient convolution's weights; columns >= Cout zero).
 * entries_dev: device array. */
typedef struct ftc_pack_entry {
    const void* src;           /* fp32 [Cout][Cin][k][k] */
    void* fwd;                 /* may be NULL */
    void* dgrad;               /* may be NULL */
    int32_t Cout, Cin, kk, cin_pad, cout_pad, dtype, reserved0, reserved1;
} ftc_pack_entry;
int ftc_pack_train_weights(const ftc_pack_entry* entries_dev, int n_entries, int64_t max_elems, void* stream);

/* Pixel splits FTC_OP_WGRAD should use for a layer (fills the GPU without oversizing the partial-sum scratch). */
int ftc_wgrad_splits(int B, int Ho, int Wo, int Cout, int Cin, int ksize);

#ifdef __cplusplus
}
#endif
#endif /* FTC_H_ */
