"""CPU (no GPU): the C-ABI library loads and exports every symbol include/ftc.h declares, plan
validation works through ftc_plan_create, and the host-side plan / weight-packing logic is sound."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from findtextcenternet_amd import _lib as L
from findtextcenternet_amd.model import FtcModel
from findtextcenternet_amd import tuning
from findtextcenternet_amd.weights import deterministic_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "ftc.h")).read()
    declared = sorted(set(re.findall(r"\b(ftc_[a-z_0-9]+)\s*\(", hdr)))
    assert set(declared) == set(L.EXPORTS), (declared, L.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ftc_abi_version() == L.FTC_ABI_VERSION
    assert C.sizeof(L.Op) == 24 * 4 + 11 * 16 and C.sizeof(L.Ref) == 16 and C.sizeof(L.Tile) == 32
    assert C.sizeof(L.Tensor) == 56 and C.sizeof(L.PlanInfo) == 48 and C.sizeof(L.OpInfo) == 96


def test_device_info_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = L.load()
    n = C.c_int()
    buf = C.create_string_buffer(64)
    assert lib.ftc_device_info(C.byref(n), buf, 64) == -3           # FTC_ERR_NO_DEVICE
    assert b"device" in lib.ftc_last_error().lower() or b"hip" in lib.ftc_last_error().lower()


def _op(**f):
    op = (L.Op * 1)()
    for k, v in f.items():
        if k in ("in_", "in2", "out", "w", "w2", "bias", "bias2", "scale", "shift", "aux", "out2"):
            r = getattr(op[0], k)
            r.base, r.offset = L.BASE_WORKSPACE, v
        else:
            setattr(op[0], k, v)
    return op


def test_plan_create_validates_ops():
    lib = L.load()
    h = C.c_void_p()
    good = dict(kind=L.OP_CONV, in_dtype=L.F32, out_dtype=L.F32, w_dtype=L.F32, B=1, H=8, W=8, Ho=8, Wo=8, Cin=32, Cin_total=32,
                Cout=64, Cout_total=64, ksize=3, stride=1, in_=0, out=65536, w=131072, bias=262144)
    assert lib.ftc_plan_create(_op(**good), 1, 1 << 20, 0, C.byref(h)) == 0
    assert lib.ftc_plan_num_ops(h) == 1
    lib.ftc_plan_destroy(h)
    for bad, msg in [(dict(good, Cin=30), b"Cin"), (dict(good, Ho=7), b"Ho/Wo"), (dict(good, ksize=5), b"ksize"),
                     (dict(good, in_=8), b"16-byte"), (dict(good, out=1 << 21), b"out of range"),
                     (dict(good, kind=99), b"unknown op"), (dict(good, flags=L.FLAG_RESIDUAL), b"in2")]:
        assert lib.ftc_plan_create(_op(**bad), 1, 1 << 20, 0, C.byref(h)) == -1
        assert msg in lib.ftc_last_error(), (msg, lib.ftc_last_error())
    # operand EXTENTS are checked, not just start offsets: an output that starts inside the workspace but runs past its end,
    # a weight matrix larger than the blob, a fused top convolution without its tap matrix
    ws = 1 << 20
    out_bytes = 1 * 8 * 8 * 64 * 4
    assert lib.ftc_plan_create(_op(**dict(good, out=ws - out_bytes)), 1, ws, 0, C.byref(h)) == 0
    lib.ftc_plan_destroy(h)
    assert lib.ftc_plan_create(_op(**dict(good, out=ws - out_bytes + 16)), 1, ws, 0, C.byref(h)) == -1 and b"offset + extent" in lib.ftc_last_error()
    wop = _op(**good)
    wop[0].w.base, wop[0].w.offset = L.BASE_WEIGHTS, 0
    assert lib.ftc_plan_create(wop, 1, ws, 64 * 9 * 32 * 4, C.byref(h)) == 0
    lib.ftc_plan_destroy(h)
    assert lib.ftc_plan_create(wop, 1, ws, 64 * 9 * 32 * 4 - 16, C.byref(h)) == -1 and b"weights operand out of range" in lib.ftc_last_error()
    top = dict(kind=L.OP_CONV, flags=L.FLAG_TOP_FUSE, act=L.ACT_GELU, in_dtype=L.BF16, out_dtype=L.BF16, w_dtype=L.BF16, B=1, H=16, W=16, Ho=16, Wo=16,
               Cin=256, Cin_total=256, Cout=192, Cout_total=192, ksize=3, stride=1, aux0=65, aux1=12, in_=0, out=1 << 19, w=0, bias=0)
    assert lib.ftc_plan_create(_op(**top), 1, 1 << 21, 0, C.byref(h)) == -1 and b"w2" in lib.ftc_last_error()
    dw = dict(kind=L.OP_DWCONV, act=L.ACT_SILU, in_dtype=L.BF16, out_dtype=L.BF16, B=1, H=8, W=8, Ho=8, Wo=8, Cin=64, Cout=64, ksize=3, stride=1, aux0=1,
              in_=0, out=8192, w=16384, bias=32768, aux=ws - 64 * 4 + 16)
    assert lib.ftc_plan_create(_op(**dw), 1, ws, 0, C.byref(h)) == -1 and b"aux" in lib.ftc_last_error()
    # running without a device / with NULL bases is an error, not a crash
    assert lib.ftc_plan_create(_op(**good), 1, 1 << 20, 0, C.byref(h)) == 0
    bases = (C.c_void_p * L.NUM_BASES)()
    assert lib.ftc_plan_run(h, bases, None, 0, -1) == -1 and b"NULL" in lib.ftc_last_error()
    lib.ftc_plan_destroy(h)


@pytest.fixture(scope="module")
def sd():
    return deterministic_state_dict(0, prefix_detector=False)


@pytest.fixture(scope="module")
def models(sd):
    """Library-side models (ftc_create: host-only, works without a GPU) of the seeded checkpoint, one per numeric mode."""
    return {mode: FtcModel(sd, mode) for mode in ("fp32", "bf16")}


def _blob(model, name, dtype, shape):
    off = model.offset(name)
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return model.weights_host()[off:off + n].view(dtype).reshape(shape).copy()


def test_create_rejects_bad_checkpoints(sd):
    bad = dict(sd)
    del bad["keyheatmap.top_conv.0.bias"]
    with pytest.raises(L.FtcError, match="keyheatmap.top_conv.0.bias"):
        FtcModel(bad, "fp32")
    bad = dict(sd)
    bad["backbone.features.4.0.block.2.fc1.weight"] = torch.zeros(24, 383, 1, 1)
    with pytest.raises(L.FtcError, match="shape"):
        FtcModel(bad, "bf16")
    with pytest.raises(L.FtcError, match="model_size"):
        FtcModel(sd, "fp32", "xxl")
    m = FtcModel(sd, "bf16")
    with pytest.raises(L.FtcError, match="multiples of 32"):
        m.plan(1, 100, 768)
    # TextDetectorModel keys (detector.* + decoder.*) are accepted as they come out of model.pt
    full = deterministic_state_dict(0)
    mf = FtcModel(full, "bf16")
    lib = L.load()
    # ... and then the SimpleDecoder is packed too (3 x (2048x128 + 2048x2048 + modulo x 2048) bf16 + fp32 biases)
    assert 38_000_000 < mf.weights_bytes - m.weights_bytes < 42_000_000 and mf.offset("decoder.0.l1.w") > 0
    assert lib.ftc_decoder_workspace_bytes(mf.handle, 2048) > 0
    assert lib.ftc_decoder_workspace_bytes(m.handle, 2048) == -1 and b"decoder" in lib.ftc_last_error()


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_plan_structure_flops_and_arena(models, mode):
    pl = models[mode].plan(2, 768, 768)                  # built AND validated (ftc_plan_create rules) inside the library
    # forward work of the reference network: 432.50 GMAC = 865.0 GFLOP per image
    assert abs(sum(m.flops for m in pl.meta) / 2 / 1e9 - 865.0006) < 0.01
    kinds = [m.kind for m in pl.meta]
    assert kinds.count("dwconv3x3") == 80 and kinds.count("se") == 80 and kinds.count("stem") == 1 and kinds[-1] == "nms"
    # nine heads x three levels of upsample+concat and conv, nine top convs: grouped launches count once per instance
    inst = [max(1, pl.ops[i].groups) for i in range(len(kinds))]
    # (the last level forms its input in the convolution's loader and reads the shared tap, no upcat launch; round 5: in the fp32 / fp16x3 plans too)
    n_up = 2
    assert sum(n for k, n in zip(kinds, inst) if k == "upcat") == 9 * n_up and kinds.count("upcat") == n_up
    # the eight map heads' top convolutions live in the epilogue of the last FPN level (+ one TAPSUM); round 5: in the fp32 / fp16x3 plans too
    fused_top = 8
    # round 6, 16-bit plans: the 7 stride-1 Fused-MBConv blocks of stage 2 are ONE launch each (FTC_OP_FMBCONV: 3x3 expand + 1x1 project; stage 3's
    # Cin = 96 runs K steps of 32 and measured slower fused: FTC_FMBFUSE_ALL=1)
    fmb = kinds.count("conv3x3+conv1x1")
    assert fmb == (7 if mode == "bf16" else 0) and sum(pl.ops[i].kind == L.OP_FMBCONV for i in range(len(kinds))) == fmb
    assert sum(n for k, n in zip(kinds, inst) if k.startswith("conv")) == 20 + 16 + 160 + 1 + 1 + 27 + 9 - fused_top - fmb
    assert kinds.count("tapsum") == (1 if fused_top else 0)
    # arena: no two simultaneously-live buffers overlap
    live = {}
    ops = pl.ops
    spans = {}
    for i in range(len(ops)):
        for fld in ("in_", "in2", "out", "aux", "scale", "out2"):
            r = getattr(ops[i], fld)
            if r.base == L.BASE_WORKSPACE:
                s = spans.setdefault(r.offset, [i, i])
                s[1] = i
    assert max(spans) < pl.workspace_bytes and pl.workspace_bytes < 0.5 * pl.info.total_buffer_bytes
    assert pl.h == 192 and pl.w == 192 and pl.info.weights_bytes == models[mode].weights_bytes
    # the same op list is accepted by the op-list entry point
    h = C.c_void_p()
    assert L.load().ftc_plan_create(pl.ops, len(pl.ops), pl.workspace_bytes, pl.info.weights_bytes, C.byref(h)) == 0
    L.load().ftc_plan_destroy(h)


def test_bn_fold_and_kmajor_layout_match_torch(sd, models):
    """ftc_create: conv + eval BatchNorm == conv with folded weights + bias, in the K-major layout."""
    name = "backbone.features.2.1.block.0"
    w = sd[name + ".0.weight"]
    cout, cin, k, _ = w.shape
    wk = torch.from_numpy(_blob(models["fp32"], name + ".w", np.float32, (cout, k * k, cin)))
    b = torch.from_numpy(_blob(models["fp32"], name + ".b", np.float32, (cout,)))
    x = torch.randn(1, cin, 9, 9)
    ref = F.batch_norm(F.conv2d(x, w, None, 1, 1), sd[name + ".1.running_mean"], sd[name + ".1.running_var"], sd[name + ".1.weight"],
                       sd[name + ".1.bias"], False, 0.0, 1e-3)
    mine = F.conv2d(x, wk.reshape(cout, k, k, cin).permute(0, 3, 1, 2), b, 1, 1)
    assert float((ref - mine).abs().max()) < 1e-5


def test_merged_fpn_level0_border_bias_is_exact(sd, models):
    """The nine per-head (in_bn -> conv3x3 -> BN) at the 1/32 tap packed as ONE conv with a
    16-case border bias equals the reference composition (Leafmap.forward i=0, detector.py:194-197)."""
    C4, N = 1280, 9 * 192
    wk = torch.from_numpy(_blob(models["fp32"], "heads.L0.w", np.float32, (N, 3, 3, C4))).permute(0, 3, 1, 2)
    b16 = torch.from_numpy(_blob(models["fp32"], "heads.L0.b", np.float32, (16, N)))
    x = torch.randn(1, C4, 4, 5)
    y = F.conv2d(x, wk, None, 1, 1)
    H, W = 4, 5
    for oy in range(H):
        for ox in range(W):
            idx = (oy == 0) | ((oy == H - 1) << 1) | ((ox == 0) << 2) | ((ox == W - 1) << 3)
            y[0, :, oy, ox] += b16[idx]
    for hi, name in enumerate(["keyheatmap", "feature"]):
        h = {"keyheatmap": 0, "feature": 8}[name]
        xin = F.batch_norm(x, sd[f"{name}.in_bn.3.running_mean"], sd[f"{name}.in_bn.3.running_var"], sd[f"{name}.in_bn.3.weight"],
                           sd[f"{name}.in_bn.3.bias"], False, 0.0, 1e-5)
        ref = F.batch_norm(F.conv2d(xin, sd[f"{name}.upsamplers.0.0.weight"], None, 1, 1), sd[f"{name}.upsamplers.0.1.running_mean"],
                           sd[f"{name}.upsamplers.0.1.running_var"], sd[f"{name}.upsamplers.0.1.weight"],
                           sd[f"{name}.upsamplers.0.1.bias"], False, 0.0, 1e-5)
        assert float((ref - y[:, h * 192:(h + 1) * 192]).abs().max()) < 2e-4


def test_tuning_table_entries_are_legal(models):
    tab = tuning.load_table()
    assert isinstance(tab, dict)
    for k, v in tab.items():
        assert 0 <= v <= 0xfff and (v & 15) - 1 < len(tuning.CFG_NAMES), (k, v)
        assert tuning.describe(v)
    # the table compiled into the library (csrc/tuning_table.inc) is the committed JSON: the bench plan carries its choices
    pl = models["bf16"].plan(8, 768, 768)
    hits = 0
    for i in range(len(pl.ops)):
        if pl.ops[i].kind == L.OP_CONV:
            want = tab.get(tuning.signature(pl.ops[i]))
            if want:
                hits += 1
                assert pl.ops[i].aux0 == want, (pl.meta[i].name, pl.ops[i].aux0, want)
    assert hits > 80                     # (round 6: the 28 convolutions of the stride-1 Fused-MBConv blocks became 14 FTC_OP_FMBCONV launches: 122 -> 94)


def test_drop_in_module_schema_and_loud_failures():
    from findtextcenternet_amd import CenterNetDetector, TextDetectorModel
    from findtextcenternet_amd.schema import text_detector_schema
    m = TextDetectorModel(pre_weights=False)
    keys = list(m.state_dict().keys())
    assert keys == list(text_detector_schema("xl").keys()) and len(keys) == 2444
    m.load_state_dict(deterministic_state_dict(0))               # strict load of the reference key set
    det = CenterNetDetector(m.detector).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        det(torch.zeros(1, 3, 64, 64))
    with pytest.raises(TypeError):
        CenterNetDetector(torch.nn.Identity())


def test_tf_npz_importer_matches_reference_load_weight(tmp_path):
    """Our TF-checkpoint importer against the reference's own load_weight (models/detector.py:30-121),
    run in the build container on a synthetic npz: per-tensor digests in tests/golden/g5_tf_import_digest.json.gz."""
    import gzip
    import hashlib
    import json
    from findtextcenternet_amd import CenterNetDetection, load_tf_efficientnetv2_npz
    from findtextcenternet_amd.weights import tf_efficientnetv2_npz_names
    with gzip.open(os.path.join(ROOT, "tests", "golden", "g5_tf_import_digest.json.gz"), "rt") as f:
        ref = json.load(f)
    det = CenterNetDetection(pre_weights=False)
    sd = det.state_dict()
    rng = np.random.Generator(np.random.PCG64(ref["seed"]))
    arrs = {}
    for key, name, perm in tf_efficientnetv2_npz_names("xl"):
        shape = tuple(sd[key].shape)
        if perm is not None:
            shape = tuple(np.array(shape)[np.argsort(perm)])
        arrs[name] = rng.standard_normal(shape).astype(np.float32)
    path = str(tmp_path / "synthetic-xl.npz")
    np.savez(path, **arrs)
    bias_before = sd["backbone.features.4.0.block.2.fc1.bias"].clone()
    assert load_tf_efficientnetv2_npz(det, path) is True
    after = det.state_dict()
    assert len(ref["digest"]) == 1550
    for key, d in ref["digest"].items():
        assert hashlib.sha1(after[key].contiguous().numpy().tobytes()).hexdigest()[:16] == d, key
    assert torch.equal(after["backbone.features.4.0.block.2.fc1.bias"], bias_before)     # SE biases are not imported (reference quirk)
    assert load_tf_efficientnetv2_npz(det, str(tmp_path / "missing.npz")) is False


def test_adamw_schedulefree_host_schedule_matches_reference_fixture():
    """The float64 per-step scalars of findtextcenternet_amd.optim.step_scalars against what the reference's own optimizer
    recorded in its param_group (scheduled_lr, lr_max, weight_sum) -- tests/golden/g6_adamw_schedulefree.npz."""
    import synth
    from findtextcenternet_amd.optim import step_scalars
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_adamw_schedulefree.npz"))
    for ci, cfg in enumerate(synth.ADAMW_CASES):
        kw = dict(lr=0.0025, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, warmup_steps=0, r=0.0, weight_lr_power=2.0)
        kw.update(cfg["kwargs"])
        group = dict(kw, k=0, weight_sum=0.0, lr_max=-1.0, scheduled_lr=0.0)
        for step in range(cfg["steps"]):
            sc = step_scalars(group)
            group["k"] += 1
            want = gold[f"c{ci}_sched"][step]
            assert [group["scheduled_lr"], group["lr_max"], group["weight_sum"]] == list(want)
            assert 0 < sc["bias_correction2"] <= 1 and sc["lr"] == group["scheduled_lr"]


@pytest.mark.parametrize("size", ["s", "m", "l"])
def test_library_builds_plans_for_the_other_model_sizes(size):
    """ftc_create + plan construction for the EfficientNetV2-S / -M / -L variants the reference can instantiate
    (models/detector.py:131-136): other stage depths, tap widths (Leafmap.in_dims :151-158) and, for 's', one stage fewer."""
    from findtextcenternet_amd.schema import STAGES
    sd_ = deterministic_state_dict(0, model_size=size, prefix_detector=False, with_decoder=False)
    for mode in ("fp32", "bf16"):
        m = FtcModel(sd_, mode, size)
        pl = m.plan(2, 128, 160)
        kinds = [x.kind for x in pl.meta]
        n_mb = sum(r[6] for r in STAGES[size] if r[0] == "mb")
        assert kinds.count("dwconv3x3") == n_mb == kinds.count("se") and kinds[0] == "stem" and kinds[-1] == "nms"
        assert pl.h == 32 and pl.w == 40
    with pytest.raises(L.FtcError, match="shape|missing"):
        FtcModel(sd_, "fp32", "xl")                          # a checkpoint of another size is rejected, not mis-read


def test_mbconv_slice_rejects_shapes_it_cannot_hold():
    lib = L.load()
    base = dict(kind=L.OP_MBHEAD, act=L.ACT_SILU, in_dtype=L.BF16, out_dtype=L.BF16, w_dtype=L.BF16, B=1, H=24, W=24, Ho=24, Wo=24, Cin=64, Cout=128,
                ksize=3, stride=1, aux0=0)
    for bad in (dict(H=25, Ho=25), dict(Cout=192), dict(Cin=48), dict(in_dtype=L.F32), dict(stride=2), dict(H=12, Ho=12, W=50, Wo=50)):
        op = (L.Op * 1)()
        for k, v in dict(base, **bad).items():
            setattr(op[0], k, int(v))
        for f in ("in_", "w2", "bias2", "w", "bias", "out", "aux"):
            r = getattr(op[0], f)
            r.base, r.offset = L.BASE_WORKSPACE, 0
        h = C.c_void_p()
        assert lib.ftc_plan_create(op, 1, 1 << 30, 0, C.byref(h)) != 0 and b"mbhead" in lib.ftc_last_error()


def test_px144_tile_configs_are_validated_and_planned(models):
    """The 144-pixel 1x1 kernel (aux0 low nibble 8..11, csrc/conv1x1_px144.hip) is refused where its tiles do not divide the op, and the bf16 batch-8
    plan runs the MBConv project convolutions of stages 4-7 on it (the tuning table's choice)."""
    lib = L.load()
    h = C.c_void_p()
    ok = dict(kind=L.OP_CONV, flags=L.FLAG_RESIDUAL, in_dtype=L.BF16, out_dtype=L.F32, w_dtype=L.BF16, res_dtype=L.F32, B=2, H=24, W=24, Ho=24, Wo=24,
              Cin=512, Cin_total=512, Cout=640, Cout_total=640, ksize=1, stride=1, in_=0, in2=1 << 22, out=1 << 23, w=1 << 21, bias=1 << 20)
    for aux0 in (8, 9, 10):                     # 640 = 10 x 64 = 8 x 80 = 5 x 128
        assert lib.ftc_plan_create(_op(**dict(ok, aux0=aux0)), 1, 1 << 25, 0, C.byref(h)) == 0, lib.ftc_last_error()
        lib.ftc_plan_destroy(h)
    for bad in (dict(ok, aux0=11),                               # 640 % 96
                dict(ok, aux0=8, H=16, W=16, Ho=16, Wo=16),      # 256 pixels per image: 144 does not divide
                dict(ok, aux0=8, Cin=480, Cin_total=480),        # K step 64
                dict(ok, aux0=8, out_dtype=L.BF16),              # fp32 output only
                dict(ok, aux0=8, act=L.ACT_SILU),
                dict(ok, aux0=8, ksize=3),
                dict(ok, aux0=8, in_dtype=L.F32, w_dtype=L.F32, flags=L.FLAG_RESIDUAL | L.FLAG_SPLIT16)):      # fp16x3 needs BOTH operands pre-split
        assert lib.ftc_plan_create(_op(**bad), 1, 1 << 26, 0, C.byref(h)) == -1, bad
        assert b"x144" in lib.ftc_last_error() or b"conv" in lib.ftc_last_error()
    x3 = dict(ok, in_dtype=L.F32, w_dtype=L.F32, flags=L.FLAG_RESIDUAL | L.FLAG_SPLIT16 | L.FLAG_PRESPLIT, w=1 << 24, in_=1 << 22, in2=0)
    assert lib.ftc_plan_create(_op(**dict(x3, aux0=8)), 1, 1 << 26, 0, C.byref(h)) == 0, lib.ftc_last_error()
    lib.ftc_plan_destroy(h)
    pl = models["bf16"].plan(8, 768, 768)
    px = [pl.meta[i].name for i in range(len(pl.ops)) if pl.ops[i].kind == L.OP_CONV and 8 <= (pl.ops[i].aux0 & 15) <= 11 and not pl.ops[i].aux0 & 64]
    assert len(px) >= 70 and all(".block.3" in n or ".block.2" in n for n in px), (len(px), px[:4])


def test_fp16x3_plans_build_with_every_documented_switch(sd, monkeypatch):
    """Round-5 advisor (medium): the tuning table's signature drops FTC_FLAG_PRESPLIT, so an fp16x3 project convolution built WITHOUT
    pre-split operands (any of the switches below) used to inherit a 144-pixel-tile hint that is only legal with them, and
    ftc_plan_create refused the whole batch-8 / batch-32 plan.  apply_tuning now adopts a hint only if the op validates with it."""
    m = FtcModel(sd, "fp16x3")
    cases = [({"FTC_NO_PRESPLIT": "1"}, 8, False), ({"FTC_NO_MBSLICE_X3": "1"}, 8, True), ({"FTC_NO_MBSLICE": "1"}, 32, False),
             ({"FTC_MBSLICE_MINWG": "1000000"}, 32, True)]
    for env, B, nchw in cases:                   # (plans are cached per (B, H, W, layout): one key per switch)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl = m.plan(B, 768, 768, nchw=nchw)
        for k in env:
            monkeypatch.delenv(k)
        px = [i for i in range(len(pl.ops)) if pl.ops[i].kind == L.OP_CONV and 8 <= (pl.ops[i].aux0 & 15) <= 11 and not pl.ops[i].aux0 & 64]
        assert all(pl.ops[i].flags & L.FLAG_PRESPLIT for i in px), env
        if "FTC_NO_PRESPLIT" in env:
            assert not px
    # and the default plan does run the fused heads and the 144-pixel tiles
    pl = m.plan(16, 768, 768)
    n_head = sum(pl.ops[i].kind == L.OP_MBHEAD for i in range(len(pl.ops)))
    px = [i for i in range(len(pl.ops)) if pl.ops[i].kind == L.OP_CONV and 8 <= (pl.ops[i].aux0 & 15) <= 11 and not pl.ops[i].aux0 & 64]
    assert n_head >= 70, n_head


def test_round6_kernels_are_selected_by_the_plans_that_bench_runs(sd, models):
    """The batch-8 plans of the two modes bench.py reports -- bf16 (`value`) and fp16x3 (`config.contract_mode`) -- run stage 1 on the resident 32 -> 32 kernel
    (csrc/conv3x3_c32.hip: 4 launches) and the stride-1 blocks of stage 2 as one launch each (FTC_OP_FMBCONV: 7); the exact-fp32 plan keeps the generic kernels."""
    lib = L.load()
    buf = C.create_string_buffer(160)

    def labels(m):
        pl = m.plan(8, 768, 768)
        out = []
        for i in range(len(pl.ops)):
            L.check(lib.ftc_op_kernel_label(C.byref(pl.ops[i]), buf, 160), "ftc_op_kernel_label")
            out.append(buf.value.decode())
        return out
    for mode, m in (("bf16", models["bf16"]), ("fp16x3", FtcModel(sd, "fp16x3")), ("fp32", models["fp32"])):
        lab = labels(m)
        n_c32 = sum(l.startswith("conv3x3_c32<") for l in lab)
        n_fmb = sum(l.startswith("fmbconv_fused<") for l in lab)
        assert (n_c32, n_fmb) == ((4, 7) if mode != "fp32" else (0, 0)), (mode, n_c32, n_fmb)
        if mode == "fp16x3":
            assert all("f16x3" in l for l in lab if l.startswith(("conv3x3_c32<", "fmbconv_fused<")))


def test_fused_block_switches_change_the_plan_and_it_still_validates(models, monkeypatch):
    """FTC_NO_FMBFUSE=1 keeps the two-launch form of the Fused-MBConv blocks, FTC_FMBFUSE_ALL=1 also fuses stage 3 (measured slower):
    both plans build and pass ftc_plan_create's validation (plans are cached per (B, H, W, layout): one key per switch)."""
    m = models["bf16"]
    for env, key, want in (({"FTC_NO_FMBFUSE": "1"}, (4, 768, 768, False), 0), ({"FTC_FMBFUSE_ALL": "1"}, (4, 768, 768, True), 14), ({}, (5, 768, 768, False), 7)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl = m.plan(key[0], key[1], key[2], nchw=key[3])
        for k in env:
            monkeypatch.delenv(k)
        assert sum(pl.ops[i].kind == L.OP_FMBCONV for i in range(len(pl.ops))) == want, env
        assert abs(sum(mm.flops for mm in pl.meta) / key[0] / 1e9 - 865.0006) < 0.01          # the same network either way
