"""Probe: the headline step (ViT encode + head) against its two halves timed alone, same process, same clip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
dev = torch.device("cuda", 0)
net, sd, eot, hc = B.build_net("f32", dev, 512)
frames = torch.randn(1, 512, 3, 224, 224, device=dev)
nc = torch.zeros(512, device=dev)
def t(fn, n=8, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    full = t(lambda: net(frames, None, nc, 1, True))
    enc = t(lambda: net.image_encoder(frames.view(-1, 3, 224, 224)))
    f = net.image_encoder(frames.view(-1, 3, 224, 224)).view(1, 1, 512, -1).clone()
    net.load_from_features = True
    head = t(lambda: net(f, None, nc, 1, True), n=50, w=5)
    net.load_from_features = False
    full2 = t(lambda: net(frames, None, nc, 1, True))
    prof = B.Prof(0)
    prof.start(); full_prof = t(lambda: net(frames, None, nc, 1, True)); prof.stop(); prof.collect()
print(f"with per-launch event pairs armed {full_prof:.3f} ms")
print(f"full {full:.3f} ms  encode {enc:.3f} ms  head alone {head:.3f} ms  sum {enc + head:.3f}  full again {full2:.3f}")
