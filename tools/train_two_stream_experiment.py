"""What-if: the weight-gradient ops of the train step's backward on a SECOND stream (they depend on the main chain, nothing on the main chain
depends on them).  Timing only: buffer lifetimes are not extended here, so the gradients of this run are garbage.
    python tools/train_two_stream_experiment.py [precision]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from findtextcenternet_amd import TextDetectorModel, deterministic_state_dict, synth  # noqa: E402
from findtextcenternet_amd import _lib as L  # noqa: E402
from findtextcenternet_amd.train_step import TrainStep  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, S = 8, 768
m = TextDetectorModel(pre_weights=False, precision=prec)
m.load_state_dict(deterministic_state_dict(0))
m = m.to("cuda").train()
ts = TrainStep(m)
x = torch.rand(B, S, S, 3, device="cuda").permute(0, 3, 1, 2)
label, idmap = synth.train_labels(1, B, S // 4, S // 4)
label, idmap = torch.from_numpy(label).cuda(), torch.from_numpy(idmap).cuda()
fmask = m.get_fmask(label, None)
for _ in range(2):
    ts.zero_grad()
    ts.forward_backward(x, label, idmap, fmask)
torch.cuda.synchronize()
plan = ts.plan_for(B, S, S)
lib = L.load()
xn = x.permute(0, 2, 3, 1).contiguous()
bases = (C.c_void_p * L.NUM_BASES)(None, ts.workspace.data_ptr(), ts.blob.data_ptr(), xn.data_ptr(), None, None, ts.grads.data_ptr())
n, nf = plan["n_ops"], plan["n_fwd"]
wg = [i for i in range(nf, n) if plan["ops"][i].kind == L.OP_WGRAD]
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
h = plan["handle"]


def run_single():
    L.check(lib.ftc_plan_run(h, bases, C.c_void_p(main.cuda_stream), 0, n - 1), "run")


def run_two():
    L.check(lib.ftc_plan_run(h, bases, C.c_void_p(main.cuda_stream), 0, nf - 1), "fwd")
    pos = nf
    for i in wg:
        if i > pos:
            L.check(lib.ftc_plan_run(h, bases, C.c_void_p(main.cuda_stream), pos, i - 1), "main")
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        L.check(lib.ftc_plan_run(h, bases, C.c_void_p(side.cuda_stream), i, i), "side")
        pos = i + 1
    if pos < n:
        L.check(lib.ftc_plan_run(h, bases, C.c_void_p(main.cuda_stream), pos, n - 1), "tail")
    main.wait_stream(side)


for name, fn in (("one stream", run_single), ("wgrad on a second stream", run_two), ("one stream", run_single)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms per forward+backward ({len(wg)} weight-gradient ops)", flush=True)
