"""Kernel-time breakdown of one head training step (UCF config, B=64) using libacx's launch timer."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import init_weights as IW, _lib as L
from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP, lookup_prompts
from anomalyclip_amd.components.loss import ComputeLoss
dev = torch.device("cuda", 0)
hc = IW.UCF_HEAD
toks = torch.tensor(lookup_prompts(key="ucf")["tokenized_prompts"], dtype=torch.int32)
net = AnomalyCLIP(arch="ViT-B/16", labels_key="ucf", emb_size=256, depth=1, heads=8, dim_heads=None, num_segments=32, seg_length=16,
                  concat_features=False, normal_id=7, stride=1, load_from_features=True, select_idx_dropout_topk=0.7,
                  select_idx_dropout_bottomk=0.7, ncrops=1, num_topk=3, num_bottomk=3)
net.load_state_dict(IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, seed=0), strict=True)
mod = AnomalyCLIPModule(net, None, None, ComputeLoss(7, 3, 1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3, 16, 32), num_classes=14, solver={"lr": 1e-5}).to(dev)
mod.ncentroid = torch.zeros(512, device=dev)
opt = mod.configure_optimizers()["optimizer"]
B = 64
g = torch.Generator().manual_seed(1)
feats = (torch.randn(B, 1, 512, 512, generator=g) * 0.3).to(dev)
labels = torch.tensor([i % 13 + (1 if i % 13 >= 7 else 0) for i in range(B // 2)] + [7] * (B // 2)).to(dev)
batch = ((feats[B // 2:], labels[B // 2:]), (feats[:B // 2], labels[:B // 2]))
net.train()
for _ in range(3):
    mod.train_batch(batch, opt)
torch.cuda.synchronize()
h = L.ctx(0); lib = L.lib()
lib.acx_prof_enable(h, 1)
t0 = time.perf_counter()
N = 5
for _ in range(N):
    mod.train_batch(batch, opt)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N * 1e3
lib.acx_prof_enable(h, 0)
counts = (ctypes.c_int32 * 4)(); tot = (ctypes.c_double * 4)()
lib.acx_prof_collect(h, counts, tot)
fl = ctypes.c_double(0); lib.acx_prof_gemm_flops(h, ctypes.byref(fl))
names = ["gemm", "attention", "norm", "other"]
print(f"wall {dt:.2f} ms/step (with event recording)")
for n, c, t in zip(names, counts, tot):
    print(f"  {n:10s} launches/step {c / N:7.1f}  kernel ms/step {t / N:8.3f}")
print(f"  sum kernel ms/step {sum(tot) / N:.3f}   gemm TFLOP/s {fl.value / N / 1e9 / (tot[0] / N):.1f}")
