"""-m gpu: the fused Schedule-Free AdamW step (ftc_adamw_schedulefree_step through findtextcenternet_amd.optim.AdamWScheduleFree)
against the reference's own optimizer run on CPU (tests/golden/g6_adamw_schedulefree.npz, written by gen_golden.py from
/root/reference/models/adamw_schedulefree.py).  fp32; tolerance 2e-6 relative + 1e-7 absolute per step: ATen's ten passes and
the single fused pass round identically except where the CPU build contracts a*b+c (documented in csrc/optim.hip)."""
import os

import numpy as np
import pytest
import torch

import synth
from findtextcenternet_amd.optim import AdamWScheduleFree

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "g6_adamw_schedulefree.npz")


def _close(a, b):
    return np.allclose(a, b, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("ci", range(len(synth.ADAMW_CASES)))
def test_fused_step_tracks_reference_optimizer(ci):
    gold = np.load(GOLD)
    cfg = synth.ADAMW_CASES[ci]
    params0, grads = synth.adamw_case(ci)
    ps = [torch.nn.Parameter(torch.from_numpy(a.copy()).cuda()) for a in params0]
    opt = AdamWScheduleFree(ps, **cfg["kwargs"])
    with pytest.raises(Exception, match="not in train mode"):
        opt.step()
    opt.train()
    for step, gs in enumerate(grads):
        for p, g in zip(ps, gs):
            p.grad = torch.from_numpy(g.copy()).cuda()
        opt.step()
        for pi, p in enumerate(ps):
            assert _close(p.detach().cpu().numpy(), gold[f"c{ci}_step{step}_p{pi}"]), (ci, step, pi)
    for pi, p in enumerate(ps):
        assert _close(opt.state[p]["z"].cpu().numpy(), gold[f"c{ci}_z{pi}"])
        assert _close(opt.state[p]["exp_avg_sq"].cpu().numpy(), gold[f"c{ci}_v{pi}"])
    opt.eval()
    for pi, p in enumerate(ps):
        assert _close(p.detach().cpu().numpy(), gold[f"c{ci}_eval_p{pi}"])
    opt.train()                                   # and back: y again
    for pi, p in enumerate(ps):
        assert _close(p.detach().cpu().numpy(), gold[f"c{ci}_step{len(grads) - 1}_p{pi}"])


def test_cpu_parameters_fail_loudly():
    p = torch.nn.Parameter(torch.zeros(8))
    opt = AdamWScheduleFree([p])
    opt.train()
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_throughput_of_the_fused_step_is_hbm_bound():
    """262 M parameters (the detector's size): 8 fp32 streams per element in one pass."""
    n = 262_000_000
    p = torch.nn.Parameter(torch.randn(n, device="cuda"))
    opt = AdamWScheduleFree([p], weight_decay=0.01)
    opt.train()
    p.grad = torch.randn(n, device="cuda")
    opt.step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        opt.step()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    gbs = n * 4 * 8 / (ms * 1e-3) / 1e9           # y, g, v, z read + y, g, v, z written
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/test_ops.log", "a") as f:
        f.write(f"adamw_schedulefree fused step: {n / 1e6:.0f} M params {ms:.3f} ms = {gbs:.0f} GB/s\n")
    assert gbs > 1500


def test_state_reload_rebuilds_the_pointer_table():
    """load_state_dict() replaces the state tensors while parameter and gradient keep their addresses: the cached chunk table
    (raw device pointers) must be rebuilt, and a deep-copied optimizer must work (its table cache is not restored)."""
    import copy
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(5000, device="cuda"))
    opt = AdamWScheduleFree([p], lr=0.01, warmup_steps=0)
    opt.train()
    g = torch.randn(5000, device="cuda")
    p.grad = g.clone()
    opt.step()
    saved = copy.deepcopy(opt.state_dict())
    snap = p.detach().clone()
    p.grad = g.clone()
    opt.step()                                   # step 2 from the live state
    after_live = p.detach().clone()
    with torch.no_grad():
        p.copy_(snap)                            # back to the state after step 1
    opt.load_state_dict(saved)                   # NEW z / exp_avg_sq tensors, same parameter and gradient addresses
    p.grad.copy_(g)
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(p.detach(), after_live)   # the restored state was used (and nothing was written through stale pointers)
    opt2 = copy.deepcopy(opt)
    assert "_tables" not in opt2.__dict__
    p2 = opt2.param_groups[0]["params"][0]
    p2.grad = g.clone()
    opt2.step()
    torch.cuda.synchronize()
    assert torch.isfinite(p2).all()
