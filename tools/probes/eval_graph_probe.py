"""Probe: can the whole test-mode head forward (features in, similarity + scores out) be captured in ONE HIP graph per tile
count S, and what does a replay cost against the eager launch sequence?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
dev = torch.device("cuda", 0)
net, sd, eot, hc = B.build_net("f32", dev)
net.load_from_features = True
nc = torch.zeros(512, device=dev)
for S in (1, 4, 12):
    feats = torch.randn(1, 1, 512 * S, 512, device=dev) * 0.3
    with torch.no_grad():
        ref_sim, ref_sc = net(feats, None, nc, S, True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                net(feats, None, nc, S, True)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        static_in = feats.clone()
        with torch.cuda.graph(g):
            out_sim, out_sc = net(static_in, None, nc, S, True)
        static_in.copy_(feats)
        g.replay()
        torch.cuda.synchronize()
        print("S", S, "graph vs eager max diff", (out_sim - ref_sim).abs().max().item(), (out_sc - ref_sc).abs().max().item())
        def t(fn, n=50):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
        print("  eager ms", round(t(lambda: net(feats, None, nc, S, True)), 3), " replay ms", round(t(lambda: (static_in.copy_(feats), g.replay())), 3))
