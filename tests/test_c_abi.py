"""The C ABI from a host with no Python in it: tests/c_abi/ftc_c_smoke.c (C11, gcc) binds include/ftc.h exactly as a non-Python
maintainer of the reference would -- ftc_create on a state_dict read from a file, hipMalloc/hipMemcpy of the packed blob,
ftc_forward on its own stream -- and compares with the reference's own golden outputs (tests/golden/g1_fwd128.npz)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import synth
from findtextcenternet_amd.weights import deterministic_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c_abi", "ftc_c_smoke")


def _exe():
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ftc.h")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c_abi", "ftc_c_smoke.c")
    if not os.path.exists(EXE) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(EXE) for f in (hdr, src)):
        import __graft_entry__
        __graft_entry__.build_c_client()
    return EXE


def test_c_client_builds_and_fails_loudly_without_a_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([_exe(), str(tmp_path / "none.bin"), str(tmp_path / "none.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "ftc error -3" in r.stderr              # FTC_ERR_NO_DEVICE through ftc_last_error()


def _write_weights(wpath):
    sd = deterministic_state_dict(0)                                     # TextDetectorModel keys, as in model.pt
    with open(wpath, "wb") as f:
        items = [(k, v) for k, v in sd.items() if v.is_floating_point()]
        f.write(b"FTCW" + struct.pack("<I", len(items)))
        for k, v in items:
            a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
            shape = list(a.shape) + [0] * (4 - a.ndim)
            f.write(struct.pack("<I", len(k)) + k.encode() + struct.pack("<I", a.ndim) + struct.pack("<4q", *shape) + struct.pack("<Q", a.nbytes))
            f.write(a.tobytes())


def test_host_side_of_the_abi_under_asan_ubsan(tmp_path):
    """the host side of the shim -- checkpoint folding / packing for all four precisions, plan building and
    validation for three shapes, op introspection, the bounded decoder-plan cache (eviction), error paths -- built with
    -fsanitize=address,undefined (tools/build_asan.sh: model.hip + ftc_api.hip instrumented, device code not) and driven by the
    host-only C client tests/c_abi/ftc_c_host_check.c.  No device is touched: runs in the build container."""
    import shutil
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_asan.sh")], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = r.stdout.strip().splitlines()[-1]
    wpath = str(tmp_path / "weights.bin")
    _write_weights(wpath)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, wpath], capture_output=True, text=True, timeout=1800, env=env)
    os.remove(wpath)
    assert r.returncode == 0 and r.stdout.startswith("OK host-side ABI walk"), (r.returncode, r.stdout[-500:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_c_client_reproduces_the_reference_golden(tmp_path, golden_dir, mode):
    wpath, cpath = str(tmp_path / "weights.bin"), str(tmp_path / "case.bin")
    _write_weights(wpath)
    g = np.load(os.path.join(golden_dir, "g1_fwd128.npz"))
    x = np.concatenate([synth.noise_images(1234, 1, 128, 128), synth.page_images(77, 1, 128, 128)])
    with open(cpath, "wb") as f:
        f.write(b"FTCC" + struct.pack("<3I", 2, 128, 128))
        f.write(np.ascontiguousarray(x, np.float32).tobytes() + np.ascontiguousarray(g["heatmap"], np.float32).tobytes()
                + np.ascontiguousarray(g["features"], np.float32).tobytes())
    r = subprocess.run([_exe(), wpath, cpath] + (["bf16"] if mode == "bf16" else []), capture_output=True, text=True, timeout=600)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/test_c_abi.log", "a") as f:
        f.write(r.stdout + r.stderr)
    assert r.returncode == 0 and r.stdout.startswith("OK gfx950"), (r.returncode, r.stdout, r.stderr)
