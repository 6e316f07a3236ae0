"""What-if: each lane's forward captured in a HIP graph and the two graphs replayed alternately -- does removing the host's 368 launches per
batch change the two-lane rate?   python tools/lanes_graph_experiment.py"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict  # noqa: E402
from findtextcenternet_amd import _lib as L  # noqa: E402

lib = L.load()
m = TextDetectorModel(pre_weights=False, precision="bf16")
m.load_state_dict(deterministic_state_dict(0))
det = CenterNetDetector(m.detector).to("cuda").eval()
B = 8
x = torch.rand((B, 768, 768, 3), device="cuda")
with torch.no_grad():
    det(x.permute(0, 3, 1, 2))
eng = m.detector._engine
wsb = eng.model.workspace_bytes(B, 768, 768)
S = 2
streams = [torch.cuda.Stream() for _ in range(S)]
ws = [torch.empty(wsb, dtype=torch.uint8, device="cuda") for _ in range(S)]
heat = [torch.empty((B, 192, 192, 10), device="cuda") for _ in range(S)]
feat = [torch.empty((B, 192, 192, 100), device="cuda") for _ in range(S)]


def fwd(i, stream):
    L.check(lib.ftc_forward(eng.handle, eng.wdev.data_ptr(), x.data_ptr(), B, 768, 768, 0, 1, heat[i].data_ptr(), feat[i].data_ptr(), ws[i].data_ptr(),
                            C.c_void_p(stream.cuda_stream)), "ftc_forward")


def run_eager(n):
    for k in range(n):
        fwd(k % S, streams[k % S])


graphs = []
for i in range(S):
    fwd(i, streams[i])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[i]):
        fwd(i, torch.cuda.current_stream())
    graphs.append(g)


def run_graph(n):
    for k in range(n):
        with torch.cuda.stream(streams[k % S]):
            graphs[k % S].replay()


for name, fn in (("eager, 2 lanes", run_eager), ("graph replay, 2 lanes", run_graph), ("eager, 2 lanes", run_eager), ("graph replay, 2 lanes", run_graph)):
    fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(60)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name:24s} {1e3 * el / 60:7.3f} ms / batch  {B * 60 / el:7.1f} img/s", flush=True)
