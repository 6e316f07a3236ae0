"""Times the train step (forward + loss + backward) at BASELINE configs[4]'s per-GPU shape (batch 8, 768x768) and attributes the time to
kernel labels (HIP events per op).  `python tools/train_step_bench.py [precision] [batch] [size]`"""
import collections
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from findtextcenternet_amd import TextDetectorModel, deterministic_state_dict, synth  # noqa: E402
from findtextcenternet_amd import _lib as L  # noqa: E402
from findtextcenternet_amd.train_step import TrainStep  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 768
m = TextDetectorModel(pre_weights=False, precision=prec)
m.load_state_dict(deterministic_state_dict(0))
m = m.to("cuda").train()
ts = TrainStep(m)
x = torch.rand(B, S, S, 3, device="cuda").permute(0, 3, 1, 2)
label, idmap = synth.train_labels(1, B, S // 4, S // 4)
label, idmap = torch.from_numpy(label).cuda(), torch.from_numpy(idmap).cuda()
fmask = m.get_fmask(label, None)
for it in range(4):
    ts.zero_grad()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, raw = ts.forward_backward(x, label, idmap, fmask)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(prec, "iter", it, f"{dt * 1e3:.1f} ms  loss {float(loss):.4f}  images/s {B / dt:.1f}", flush=True)
plan = ts.plan_for(B, S, S)
print("ops", plan["n_ops"], "fwd", plan["n_fwd"], "workspace GB", plan["workspace_bytes"] / 2**30)
lib = L.load()
ms = (C.c_float * plan["n_ops"])()
xn = x.permute(0, 2, 3, 1).contiguous()
bases = (C.c_void_p * L.NUM_BASES)(None, ts.workspace.data_ptr(), ts.blob.data_ptr(), xn.data_ptr(), None, None, ts.grads.data_ptr())
ts.zero_grad()
L.check(lib.ftc_plan_profile(plan["handle"], bases, C.c_void_p(torch.cuda.current_stream().cuda_stream), ms), "profile")
a = np.array(list(ms))
print(f"sum of op times {a.sum():.1f} ms: forward {a[:plan['n_fwd']].sum():.1f}, backward {a[plan['n_fwd']:].sum():.1f}")
agg = collections.defaultdict(lambda: [0.0, 0])
buf = C.create_string_buffer(256)
for i in range(plan["n_ops"]):
    lib.ftc_op_kernel_label(C.byref(plan["ops"][i]), buf, 256)
    lab = buf.value.decode()
    if plan["ops"][i].kind == L.OP_CONV:
        lab = ("dgrad:" if plan["names"][i].startswith("dgrad:") else "fwd:") + lab.split("<")[0]
    agg[lab][0] += float(a[i])
    agg[lab][1] += 1
for lab, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {t:8.2f} ms  {n:5d}  {lab}")
slow = sorted(((float(v), i) for i, v in enumerate(a)), reverse=True)[:25]
for v, i in slow:
    print(f"  op {i:5d} {v:7.3f} ms  {plan['names'][i]}")
# weight-gradient ops by shape: time and TFLOP/s
wg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for i in range(plan["n_ops"]):
    o = plan["ops"][i]
    if o.kind != L.OP_WGRAD:
        continue
    P = o.B * o.Ho * o.Wo
    key = (o.Cout, o.Cin, o.ksize, o.stride, P, o.aux0, o.in_dtype, o.res_dtype, bool(o.flags & L.FLAG_SE_SCALE))
    wg[key][0] += float(a[i])
    wg[key][1] += 1
    wg[key][2] += 2.0 * P * o.Cout * o.Cin * o.ksize * o.ksize
print("wgrad by shape (Cout, Cin, k, stride, pixels, splits, x dtype, dz dtype, se): ms total, n, TFLOP/s")
for key, (t, n, fl) in sorted(wg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {t:7.2f} ms {n:3d}  {fl / (t * 1e-3) / 1e12:7.1f} TF  {key}")
