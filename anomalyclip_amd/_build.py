"""Builds libacx.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.  In-tree output so the
binary travels with the repository snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["acx_api.hip", "acx_gemm.hip", "acx_norm.hip", "acx_attn.hip", "acx_head.hip", "acx_train.hip", "acx_metrics.hip", "acx_probe.hip", "acx_step.hip"]
LIB = os.path.join(CSRC, "libacx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "acx.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC, *FLAGS, "-c", path, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    tmp = f"{LIB}.{os.getpid()}.tmp"
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, LIB)          # atomic: concurrent importers never see a half-written library
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
