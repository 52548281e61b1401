"""Development probe: the 512-frame ViT encode as TWO 256-frame half batches on two HIP streams (own workspaces), with the persistent
bf16 x 6 GEMMs capped at fewer workgroups than CUs (ACX_OPT_X6_CUS) so that one half's memory-bound launches (LayerNorm, attention)
find CUs beside the other half's GEMMs -- against the one-launch and the sequential two-launch forms.  frames/s, HIP events."""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from anomalyclip_amd import ops

dev = torch.device("cuda:0")
net, sd, eot, hc = B.build_net("auto", dev, 512)
vit = net.image_encoder
vit2 = copy.deepcopy(vit)
vit2._ws = None
vit2._wcache = None
COPIES = [copy.deepcopy(vit) for _ in range(7)]            # (before the first call: the weight caches hold ctypes tables)
frames = torch.randn(512, 3, 224, 224, device=dev)
fa, fb = frames[:256].contiguous(), frames[256:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, n=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def one():
    vit.chunk = 512
    vit(frames)


def seq():
    vit.chunk = 256
    vit(frames)


def two():
    vit.chunk = 256
    vit2.chunk = 256
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        vit(fa)
    with torch.cuda.stream(s2):
        vit2(fb)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def multi(n):
    vits = [vit] + COPIES[: n - 1]
    per = 512 // n
    parts = [frames[i * per:(i + 1) * per].contiguous() for i in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]

    def run():
        cur = torch.cuda.current_stream()
        for v, f, st in zip(vits, parts, streams):
            v.chunk = per
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                v(f)
        for st in streams:
            cur.wait_stream(st)
    return run


t1 = timeit(one)
print(f"one 512-frame launch {512 / t1:7.0f} frames/s", flush=True)
for n in (2, 3, 4, 8):
    if 512 % n:
        per = 512 // n
    tn = timeit(multi(n))
    print(f"{n} streams x {512 // n} frames: {(512 // n) * n / tn:7.0f} frames/s", flush=True)
t1 = timeit(one)
print(f"one 512-frame launch {512 / t1:7.0f} frames/s", flush=True)
