"""ctypes binding of include/ftc.h.  The HIP library is REQUIRED: there is no CPU or eager-PyTorch
fallback anywhere in the product path -- if ``libftc_hip.so`` is missing or fails to load, importing
callers get a loud ``FtcLibraryError``."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libftc_hip.so")

FTC_ABI_VERSION = 10
F32, BF16, F16 = 0, 1, 2
(BASE_NULL, BASE_WORKSPACE, BASE_WEIGHTS, BASE_INPUT, BASE_HEATMAP, BASE_FEATURES, BASE_GRADS, NUM_BASES) = range(8)
OP_STEM, OP_CONV, OP_DWCONV, OP_SE, OP_UPCAT, OP_NMS, OP_TAPSUM, OP_BNSTAT, OP_BNACT = 1, 2, 3, 4, 5, 6, 7, 8, 9
(OP_GATHER_ROWS, OP_LOSSES, OP_LOSS_BWD, OP_SCATTER_ROWS, OP_BNBWD, OP_WGRAD, OP_DWBWD, OP_SEBWD, OP_UPCATBWD, OP_DILATE, OP_TOPDGRAD, OP_COLSUM,
 OP_STEMWGRAD, OP_FILL, OP_JOIN, OP_MBHEAD, OP_FMBCONV) = range(10, 27)
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
FLAG_RESIDUAL, FLAG_SE_SCALE, FLAG_IN_NCHW, FLAG_BORDER_BIAS, FLAG_W_PER_IMAGE, FLAG_SE_FOLD = 1, 2, 4, 8, 16, 32
FLAG_GROUP_IN_SLICE, FLAG_GROUP_OUT_SLICE = 64, 128
FLAG_TOP_FUSE, FLAG_UPCAT_IN, FLAG_GROUP_IN2_SHARED, FLAG_W_FRAG, FLAG_ACCUM, FLAG_SPLIT16 = 0x10000, 0x20000, 0x200000, 0x400000, 0x800000, 0x2000000
FLAG_SIDE_STREAM = 0x4000000
FLAG_SE_HPART = 0x8000000
FLAG_KBLOCK32 = 0x10000000
FLAG_PRESPLIT = 0x20000000
MBHEAD_SLICE = 128

EXPORTS = ["ftc_abi_version", "ftc_last_error", "ftc_device_info", "ftc_plan_create", "ftc_plan_destroy",
           "ftc_plan_num_ops", "ftc_plan_run", "ftc_plan_run_streams", "ftc_plan_profile", "ftc_op_kernel_label", "ftc_decode_scratch_bytes", "ftc_decode", "ftc_tile_gather", "ftc_paste_maps",
           "ftc_page_merge_scratch_bytes", "ftc_box_hists", "ftc_page_order_scratch_bytes", "ftc_page_order", "ftc_page_merge", "ftc_page_merge_variant", "ftc_adamw_schedulefree_step",
           "ftc_create", "ftc_destroy", "ftc_weights_bytes", "ftc_weights_host", "ftc_weights_offset", "ftc_workspace_bytes", "ftc_forward",
           "ftc_model_plan", "ftc_model_op_info", "ftc_plan_op",
           "ftc_topk_mask", "ftc_mask_compact", "ftc_gather_rows", "ftc_decoder_workspace_bytes", "ftc_decoder_forward",
           "ftc_losses_scratch_bytes", "ftc_losses", "ftc_cov_weighting_step", "ftc_pack_train_weights", "ftc_wgrad_splits"]


class FtcLibraryError(RuntimeError):
    pass


class FtcError(RuntimeError):
    pass


class Ref(C.Structure):
    _fields_ = [("base", C.c_int32), ("reserved", C.c_int32), ("offset", C.c_int64)]


class Op(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "kind", "flags", "act", "in_dtype", "out_dtype", "w_dtype", "B", "H", "W", "Ho", "Wo", "Cin", "Cin_total",
        "cin_off", "Cout", "Cout_total", "cout_off", "ksize", "stride", "aux0", "aux1", "res_dtype", "groups", "reserved0")] + [
        (n, Ref) for n in ("in_", "in2", "out", "w", "w2", "bias", "bias2", "scale", "shift", "aux", "out2")]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class PlanInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_ops", "map_h", "map_w", "reserved")] + [
        (n, C.c_int64) for n in ("workspace_bytes", "weights_bytes", "peak_live_bytes", "total_buffer_bytes")]


class OpInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("kind", C.c_char * 16), ("flops", C.c_double), ("bytes", C.c_double)]


class MtChunk(C.Structure):
    _fields_ = [("y", C.c_void_p), ("g", C.c_void_p), ("v", C.c_void_p), ("z", C.c_void_p), ("n", C.c_int32), ("reserved", C.c_int32)]


class PackEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("fwd", C.c_void_p), ("dgrad", C.c_void_p)] + [
        (n, C.c_int32) for n in ("Cout", "Cin", "kk", "cin_pad", "cout_pad", "dtype", "reserved0", "reserved1")]


class Tile(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("offset_x", "offset_y", "page_w", "page_h", "x_min", "x_max", "y_min", "y_max")]


_lib = None


def load():
    """Loads (once) and returns the ctypes handle; raises FtcLibraryError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FtcLibraryError(
            f"{LIB_PATH} not found: build it with `python -m findtextcenternet_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the detector path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise FtcLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.ftc_abi_version.restype = i32
    lib.ftc_last_error.restype = C.c_char_p
    lib.ftc_device_info.argtypes = [C.POINTER(i32), C.c_char_p, i32]
    lib.ftc_plan_create.argtypes = [C.POINTER(Op), i32, i64, i64, C.POINTER(vp)]
    lib.ftc_plan_destroy.argtypes = [vp]
    lib.ftc_plan_destroy.restype = None
    lib.ftc_plan_num_ops.argtypes = [vp]
    lib.ftc_plan_run.argtypes = [vp, C.POINTER(vp), vp, i32, i32]
    lib.ftc_plan_run_streams.argtypes = [vp, C.POINTER(vp), vp, vp, i32, i32]
    lib.ftc_plan_profile.argtypes = [vp, C.POINTER(vp), vp, C.POINTER(C.c_float)]
    lib.ftc_op_kernel_label.argtypes = [C.POINTER(Op), C.c_char_p, i32]
    lib.ftc_decode_scratch_bytes.argtypes = [i32, i32, i32]
    lib.ftc_decode_scratch_bytes.restype = i64
    lib.ftc_decode.argtypes = [vp, vp, i32, i32, i32, i32, vp, C.c_float, i32, i32, vp, i32, vp, i32, vp, vp, vp, vp]
    lib.ftc_tile_gather.argtypes = [vp, i32, i32, vp, i32, i32, i32, vp, vp]
    lib.ftc_paste_maps.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, i32, vp]
    lib.ftc_page_merge_scratch_bytes.argtypes = [i32, i32, i32]
    lib.ftc_page_merge_scratch_bytes.restype = i64
    lib.ftc_box_hists.argtypes = [vp, i32, vp, i32, i32, C.c_float, vp, vp]
    lib.ftc_adamw_schedulefree_step.argtypes = [vp, i32] + [C.c_float] * 8 + [i32, vp]
    lib.ftc_page_order_scratch_bytes.argtypes = [i32]
    lib.ftc_page_order_scratch_bytes.restype = i64
    lib.ftc_page_order.argtypes = [vp, i32, vp, C.c_float, vp, vp, vp, i64, vp]
    lib.ftc_page_merge.argtypes = [vp, vp, i32, vp, vp, C.c_float, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i64, vp]
    lib.ftc_page_merge_variant.argtypes = [vp, vp, i32, vp, vp, C.c_float, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.c_double, vp, vp, vp, vp, vp, i64, vp]
    lib.ftc_create.argtypes = [C.POINTER(Tensor), i32, C.c_char_p, i32, C.POINTER(vp)]
    lib.ftc_destroy.argtypes = [vp]
    lib.ftc_destroy.restype = None
    lib.ftc_weights_bytes.argtypes = [vp]
    lib.ftc_weights_bytes.restype = i64
    lib.ftc_weights_host.argtypes = [vp]
    lib.ftc_weights_host.restype = vp
    lib.ftc_weights_offset.argtypes = [vp, C.c_char_p]
    lib.ftc_weights_offset.restype = i64
    lib.ftc_workspace_bytes.argtypes = [vp, i32, i32, i32]
    lib.ftc_workspace_bytes.restype = i64
    lib.ftc_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.ftc_model_plan.argtypes = [vp, i32, i32, i32, i32, C.POINTER(vp), C.POINTER(PlanInfo)]
    lib.ftc_model_op_info.argtypes = [vp, i32, i32, i32, i32, i32, C.POINTER(OpInfo)]
    lib.ftc_plan_op.argtypes = [vp, i32, C.POINTER(Op)]
    lib.ftc_topk_mask.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    lib.ftc_mask_compact.argtypes = [vp, i64, vp, i64, vp, vp]
    lib.ftc_gather_rows.argtypes = [vp, vp, vp, i64, i32, i32, vp, i32, vp]
    lib.ftc_decoder_workspace_bytes.argtypes = [vp, i32]
    lib.ftc_decoder_workspace_bytes.restype = i64
    lib.ftc_decoder_forward.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.ftc_losses_scratch_bytes.restype = i64
    lib.ftc_losses.argtypes = [vp, C.POINTER(i64), vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp]
    lib.ftc_cov_weighting_step.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.ftc_pack_train_weights.argtypes = [vp, i32, i64, vp]
    lib.ftc_wgrad_splits.argtypes = [i32] * 6
    if lib.ftc_abi_version() != FTC_ABI_VERSION:
        raise FtcLibraryError(f"ABI mismatch: library {lib.ftc_abi_version()} vs binding {FTC_ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ftc_last_error().decode(errors="replace")
        raise FtcError(f"{what} failed (status {rc}): {msg}")
