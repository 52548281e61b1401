"""Probe: which PyTorch glue kernels (copies, adds, fills ... everything that is not a libacx launch) one EAGER training step of the
UCF head issues, grouped by the Python line that calls them.  usage: torch_glue_prof.py [emulated world] (default 8 -> 8 videos)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import bench_head as BH
from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
from anomalyclip_amd.components.loss import ComputeLoss

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
net, sd, eot, hc = B.build_net("f32", dev)
net.load_from_features = True
net.text_class_parallel = True
if len(sys.argv) > 2 and sys.argv[2] == "graphs":
    net.text_graph = True
    net.temporal_model.graph = True
crit = ComputeLoss(7, 3, 1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3, 16, 32)
mod = AnomalyCLIPModule(net, None, None, crit, num_classes=14, solver={"lr": 1e-5}).to(dev)
mod.ncentroid = torch.zeros(512, device=dev)
opt = mod.configure_optimizers()["optimizer"]
batch, idx = B.head_batch(64, world, 0, dev)
if world > 1:
    BH.stub_collectives(world, net)
net.train()
def step(i):
    torch.manual_seed(i)
    mt, mb = type(net.selector_model).generate_mask(net.selector_model, 64)
    net.selector_model.generate_mask = lambda b, mt=mt[idx], mb=mb[idx]: (mt, mb)
    mod.train_batch(batch, opt)
for i in range(4):
    step(i)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNEL_OPS = ("add", "copy_", "fill_", "cat", "div", "sum", "mul", "clone", "zeros", "zero_", "sub", "_to_copy", "contiguous", "index",
              "stack", "ones", "full", "masked", "where", "repeat", "expand_copy", "gather", "scatter", "arange", "clamp", "pow")
agg = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        base = str(func).split(".")[1] if "." in str(func) else name
        if any(base.startswith(k) for k in KERNEL_OPS):
            dev_args = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            if any(a.is_cuda for a in dev_args) or base in ("zeros", "ones", "full", "arange"):
                where = "autograd engine / no package frame"
                for fr in reversed(traceback.extract_stack()):
                    if "anomalyclip_amd" in fr.filename and "site-packages" not in fr.filename:
                        where = f"{fr.filename.replace(ROOT + '/', '')}:{fr.lineno} {fr.line[:70] if fr.line else ''}"
                        break
                agg[(base, where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    for i in range(3, 6):
        step(i)
torch.cuda.synchronize()
print(f"aten ops that launch (or may launch) a kernel: {sum(agg.values()) / 3:.1f} per step")
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n / 3:6.1f}  {name:14s} {where}")
