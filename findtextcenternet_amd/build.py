"""Builds the in-tree HIP library (gfx950 only) with hipcc; no JIT cache, no site-packages install."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libftc_hip.so")
SOURCES = ["conv_igemm_bf16_bb.hip", "conv_igemm_bf16_fb.hip", "conv_igemm_bf16_bf.hip", "conv_igemm_bf16_ff.hip",
           "conv_igemm_f32.hip", "conv_igemm.hip", "backbone_ops.hip", "fpn_ops.hip", "decode.hip", "page_ops.hip", "page_merge.hip", "optim.hip", "ftc_api.hip"]
HEADERS = [os.path.join(CSRC, "ftc_common.h"), os.path.join(CSRC, "conv_igemm_impl.h"), os.path.join(os.path.dirname(HERE), "include", "ftc.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function"]
# conv_igemm_part.hip is compiled once per (combination, part): (object name, -D flags)
PARTS = [(f"conv_igemm_bf16_{name}_p{part}.o", [f"-DFTC_PART_FN=launch_conv_bf16_{name}_p{part}", f"-DFTC_PART_OUT={outt}", f"-DFTC_PART={part}"])
         for name, outt in (("bb", "__bf16"), ("bf", "float")) for part in range(4)]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o, []))
    part_src = os.path.join(CSRC, "conv_igemm_part.hip")
    for obj, defs in PARTS:
        o = os.path.join(objdir, obj)
        if force or _stale(o, [part_src] + HEADERS):
            jobs.append((part_src, o, defs))
    jobs.sort(key=lambda j: 0 if "conv_igemm" in j[1] else 1)          # the long compiles first

    def cc(job):
        s, o, defs = job
        cmd = [hipcc] + FLAGS + defs + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return o

    if jobs:
        if verbose:
            print(f"[ftc build] compiling {len(jobs)} HIP source(s) for gfx950 ...", flush=True)
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES] + [os.path.join(objdir, obj) for obj, _ in PARTS]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose:
            print("[ftc build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
