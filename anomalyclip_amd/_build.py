"""Builds libacx.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.  In-tree output so the
binary travels with the repository snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["acx_api.hip", "acx_gemm.hip", "acx_norm.hip", "acx_attn.hip", "acx_head.hip", "acx_train.hip", "acx_metrics.hip", "acx_probe.hip", "acx_step.hip", "acx_comm.hip"]
LIB = os.path.join(CSRC, "libacx.so")
RESOURCES = os.path.join(CSRC, "libacx.resources.tsv")    # per-kernel registers / scratch / occupancy of the build (kernel_resources())
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]       # (remarks only: the table below; no effect on the code)
_RES_KEYS = ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def _parse_resources(src: str, text: str, rows: dict) -> str:
    """hipcc's kernel-resource-usage remarks of one source -> rows[mangled kernel name]; returns the output without them"""
    import re
    keep, cur = [], None
    for line in text.split("\n"):
        if "remark:" in line and "[-Rpass-analysis=kernel-resource-usage]" in line:
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = rows.setdefault(m.group(1), {"file": src})
                continue
            for k in _RES_KEYS:
                m = re.search(re.escape(k) + r": (\d+)", line)
                if m and cur is not None:
                    cur[k] = int(m.group(1))
            continue
        keep.append(line)
    text = "\n".join(keep)
    # (each remark drags its source-context lines along: without a real diagnostic in the output there is nothing to show)
    return text if ("warning:" in text or "error:" in text) else ""


def kernel_resources() -> dict:
    """{mangled kernel name: {"file", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "Occupancy [waves/SIMD]", ...}} of the
    library as built (written by build(); tests/test_cpu_host.py holds every kernel to zero scratch)."""
    out = {}
    with open(RESOURCES) as fh:
        head = fh.readline().rstrip("\n").split("\t")
        for line in fh:
            v = line.rstrip("\n").split("\t")
            out[v[0]] = {h: (int(x) if x.lstrip("-").isdigit() else x) for h, x in zip(head[1:], v[1:])}
    return out


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
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
    rows: dict = {}
    for src, p in procs:
        out, _ = p.communicate()
        out = _parse_resources(src, out, rows)
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
    head = ["kernel", "file"] + list(_RES_KEYS)
    rtmp = f"{RESOURCES}.{os.getpid()}.tmp"
    with open(rtmp, "w") as fh:
        fh.write("\t".join(head) + "\n")
        for k in sorted(rows):
            fh.write("\t".join([k] + [str(rows[k].get(h, "")) for h in head[1:]]) + "\n")
    os.replace(rtmp, RESOURCES)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
