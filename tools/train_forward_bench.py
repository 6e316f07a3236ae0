import time, torch, sys
sys.path.insert(0, '.')
from findtextcenternet_amd import TextDetectorModel, deterministic_state_dict, synth
for prec in ("bf16", "fp32"):
    m = TextDetectorModel(pre_weights=False, precision=prec); m.load_state_dict(deterministic_state_dict(0)); m = m.to("cuda").train()
    B = 8
    x = torch.rand(B, 768, 768, 3, device="cuda").permute(0, 3, 1, 2)
    label, _ = synth.train_labels(1, B, 192, 192)
    with torch.no_grad():
        fmask = m.get_fmask(torch.from_numpy(label).cuda(), None)
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            maps, dec = m(x, fmask)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            tf = m.__dict__["_train_forward"]
            print(prec, "iter", it, f"{dt*1e3:.1f} ms", "workspace GB", tf.workspace.numel()/2**30, "ops", tf.plans[(B,768,768)]["n_ops"], "finite", bool(torch.isfinite(maps).all()), flush=True)
    del m; torch.cuda.empty_cache()

# where the time goes: the op list alone vs the Python bookkeeping around it
import ctypes as C
import numpy as np
from findtextcenternet_amd import _lib as L
m = TextDetectorModel(pre_weights=False, precision="bf16"); m.load_state_dict(deterministic_state_dict(0)); m = m.to("cuda").train()
with torch.no_grad():
    fmask = m.get_fmask(torch.from_numpy(label).cuda(), None)
    m(x, fmask)
    tf = m.__dict__["_train_forward"]
    plan = tf.plans[(8, 768, 768)]
    tf.workspace = torch.empty(plan["workspace_bytes"], dtype=torch.uint8, device="cuda")
    xn = x.permute(0, 2, 3, 1).contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter(); tf._run(plan, xn.data_ptr()); torch.cuda.synchronize(); print("detector op list alone", (time.perf_counter() - t0) * 1e3, "ms; workspace", plan["workspace_bytes"] / 2**30, "GB")
    t0 = time.perf_counter(); tf._unpack_running_stats(); torch.cuda.synchronize(); print("running-stat write-back", (time.perf_counter() - t0) * 1e3, "ms")
    lib = L.load()
    ms = (C.c_float * plan["n_ops"])()
    bases = (C.c_void_p * L.NUM_BASES)(None, tf.workspace.data_ptr(), tf.wdev.data_ptr(), xn.data_ptr(), None, None)
    L.check(lib.ftc_plan_profile(plan["handle"], bases, C.c_void_p(torch.cuda.current_stream().cuda_stream), ms), "profile")
    a = np.array(list(ms))
    print("sum of op times", a.sum(), "ms; slowest ops:", sorted(((round(float(v), 2), i) for i, v in enumerate(a)), reverse=True)[:8])
