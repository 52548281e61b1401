#!/usr/bin/env python
"""Per-kernel register / occupancy / spill table of libacx (hipcc -Rpass-analysis=kernel-resource-usage over every .hip source;
runs without a GPU).  A kernel that crosses a VGPR step (64 / 72 / 80 / 96 / 128 / 168 / 256 allocated registers) loses a wave
per SIMD -- round 4's in-kernel split-K reduction took gemm_f32_w8_kernel from 108 to 178 VGPRs (one workgroup per CU instead of
two) and nothing but this report shows that before a GPU run.

    python tools/kernel_resources.py --out profiles/r04_kernel_resources.tsv [--against profiles/r03_kernel_resources.tsv]
    python tools/kernel_resources.py --src-root /tmp/r3src --out /tmp/r3.tsv        # another checkout of the sources
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("(anonymous namespace)::", "") for o in out[:len(names)]]


def report(src_root):
    csrc = os.path.join(src_root, "anomalyclip_amd", "csrc")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rows = {}
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        with tempfile.NamedTemporaryFile(suffix=".o") as obj:
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f, "-o", obj.name,
                                "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True)
        if r.returncode:
            sys.exit(f"hipcc failed on {f}:\n{r.stderr[-2000:]}")
        cur = None
        for line in r.stderr.split("\n"):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = rows.setdefault(m.group(1), {"file": f})
                continue
            for k in KEYS:
                m = re.search(re.escape(k) + r": (\d+)", line)
                if m and cur is not None:
                    cur[k] = int(m.group(1))
    names = list(rows)
    return {d: rows[n] for n, d in zip(names, demangle(names))}


def load(path):
    out = {}
    with open(path) as fh:
        head = fh.readline().rstrip("\n").split("\t")
        for line in fh:
            v = line.rstrip("\n").split("\t")
            out[v[0]] = {h: (int(x) if x.lstrip("-").isdigit() else x) for h, x in zip(head[1:], v[1:])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src-root", default=ROOT)
    ap.add_argument("--out", default=None)
    ap.add_argument("--against", default=None, help="an earlier table: report kernels whose occupancy / spills / scratch changed")
    a = ap.parse_args()
    rows = report(a.src_root)
    head = ["kernel", "file"] + list(KEYS)
    text = "\t".join(head) + "\n" + "".join(
        "\t".join([k] + [str(v.get(h, "")) for h in head[1:]]) + "\n" for k, v in sorted(rows.items()))
    if a.out:
        open(a.out, "w").write(text)
    else:
        sys.stdout.write(text)
    if a.against:
        old = load(a.against)
        bad = 0
        for k, v in sorted(rows.items()):
            o = old.get(k)
            if o is None:
                continue
            for key in ("Occupancy [waves/SIMD]", "VGPRs Spill", "ScratchSize [bytes/lane]"):
                if o.get(key) != v.get(key):
                    print(f"CHANGED {k}: {key} {o.get(key)} -> {v.get(key)} (VGPRs {o.get('VGPRs')} -> {v.get('VGPRs')})")
                    bad += 1
        print(f"{bad} change(s) against {a.against}; {len(rows)} kernels")


if __name__ == "__main__":
    main()
